import os, sys, torch, torch.distributed as dist
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from vla_adapter_amd import ddp, engine as E, synthetic as S
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = "cuda:0"; cfg = E.tiny_config(); W = S.make_weights(cfg, dev, seed=3, std=0.05); batch = S.make_batch(cfg, 8, dev, seed=100, P=40)
def run(mode):
    eng = E.VLAEngine(cfg, W, dev)
    if mode:
        eng.reducer = ddp.FlatGradReducer()
        eng.reducer.world = 2
        type(eng.reducer).grad_scale = property(lambda self: 1.0)
        if mode == 2:     # same stream/event choreography, collectives replaced by nothing
            import types
            def fake(self, flat, start=0, end=None, after_event=None):
                if after_event is not None: self.stream.wait_event(after_event)
                else: self.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.stream):
                    done = torch.cuda.Event(); done.record()
                self._pending = True
                return done
            eng.reducer.reduce_async = types.MethodType(fake, eng.reducer)
    eng.capture({k: v.clone() for k, v in batch.items()}, None)
    out = []
    for _ in range(4):
        l = eng.train_step_graphed(1e-3)[0].item()
        torch.cuda.synchronize()
        out.append((round(l, 5), round(eng.head.P.grad.float().norm().item(), 5), round(eng.head.P.data.float().norm().item(), 4), eng.step_count))
    eng.flush(); torch.cuda.synchronize()
    return out
for m in (0, 1, 2, 1, 0):
    print("mode", m, run(m))
dist.destroy_process_group()
