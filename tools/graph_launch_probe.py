"""How does hipGraphLaunch behave when graphs are queued back-to-back?  Measures host time per replay() call and the
GPU-side gaps for (a) the same stream, (b) alternating streams chained by events."""
import sys
import time

import torch

sys.path.insert(0, ".")
dev = "cuda"
x = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
w = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)


def body(n):
    y = x
    for _ in range(n):
        y = torch.mm(y, w)
    return y


def cap(n, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        body(n)
    return g


for nk in (20, 100, 300):
    s0, s1, cs = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    body(3)
    torch.cuda.synchronize()
    gs = [cap(nk, cs) for _ in range(6)]
    # single graph duration
    for g in gs:
        with torch.cuda.stream(s0):
            g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s0):
        e0.record()
        gs[0].replay()
        e1.record()
    torch.cuda.synchronize()
    one = e0.elapsed_time(e1)
    # (a) same stream
    host = []
    with torch.cuda.stream(s0):
        e0.record()
        for g in gs:
            t = time.perf_counter()
            g.replay()
            host.append((time.perf_counter() - t) * 1e3)
        e1.record()
    torch.cuda.synchronize()
    tot_a = e0.elapsed_time(e1)
    # (b) alternating streams chained by events
    host_b = []
    prev = torch.cuda.Event()
    e0.record(s0)
    prev.record(s0)
    for k, g in enumerate(gs):
        st = (s0, s1)[k % 2]
        with torch.cuda.stream(st):
            st.wait_event(prev)
            t = time.perf_counter()
            g.replay()
            host_b.append((time.perf_counter() - t) * 1e3)
            prev = torch.cuda.Event()
            prev.record(st)
    s0.wait_event(prev)
    e1.record(s0)
    torch.cuda.synchronize()
    tot_b = e0.elapsed_time(e1)
    print(f"nodes {nk}: one graph {one:.3f} ms | same stream x6: {tot_a:.3f} ms (ideal {6 * one:.3f}), host per replay {['%.3f' % h for h in host]}"
          f" | alternating: {tot_b:.3f} ms, host {['%.3f' % h for h in host_b]}", flush=True)
