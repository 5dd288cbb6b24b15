"""GPU input stage.  The reference's ActionTokenizer module cannot be imported here (it imports
transformers.models.qwen2.tokenization_qwen2_fast, absent from the installed transformers) and torchvision is not
installed, so these checks restate the two formulas from the reference text - action_tokenizer.py:60-66 is np.clip +
np.digitize, processing_prismatic.py:128-145 is ToTensor + Normalize - with numpy / torch on the CPU: parity unpinned by
reference execution, bit-exact against the restatement."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_action_tokenize_matches_numpy_digitize():
    from vla_adapter_amd.input_stage import GPUInputStage
    st = GPUInputStage(DEV)
    bins = np.linspace(-1, 1, 256)
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(-1.3, 1.3, 5000).astype(np.float32), bins.astype(np.float32), np.float32([-1, 1, 0, -5, 5]),
                        np.nextafter(bins.astype(np.float32), np.float32(2)), np.nextafter(bins.astype(np.float32), np.float32(-2))])
    ref = 151643 - np.digitize(np.clip(a, a_min=-1.0, a_max=1.0), bins)           # action_tokenizer.py:62-66
    got = st.tokenize_actions(torch.from_numpy(a)).cpu().numpy()
    assert np.array_equal(got, ref)
    assert got.min() >= 151387 and got.max() <= 151642                            # the 256 action tokens above ACTION_TOKEN_BEGIN_IDX


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_image_normalize_matches_totensor_normalize(dtype):
    from vla_adapter_amd.input_stage import GPUInputStage, IMAGENET_MEAN, IMAGENET_STD, SIGLIP_MEAN, SIGLIP_STD
    st = GPUInputStage(DEV, backbones=("dino", "siglip"), out_dtype=dtype)
    g = torch.Generator().manual_seed(1)
    prim, wrist = (torch.randint(0, 256, (3, 224, 224, 3), generator=g, dtype=torch.uint8) for _ in range(2))
    out = st.pixels([prim, wrist])
    assert tuple(out.shape) == (3, 12, 224, 224)

    def ref_one(img, mean, std):      # TVF.to_tensor: u8 HWC -> f32 CHW / 255 ; TVF.normalize: (t - mean) / std
        t = img.permute(0, 3, 1, 2).to(torch.float32).div(255)
        return t.sub(torch.tensor(mean).view(1, 3, 1, 1)).div(torch.tensor(std).view(1, 3, 1, 1))
    ref = torch.cat([ref_one(prim, IMAGENET_MEAN, IMAGENET_STD), ref_one(prim, SIGLIP_MEAN, SIGLIP_STD),
                     ref_one(wrist, IMAGENET_MEAN, IMAGENET_STD), ref_one(wrist, SIGLIP_MEAN, SIGLIP_STD)], dim=1).to(dtype)
    assert torch.equal(out.cpu(), ref)


def test_build_batch_follows_transform_and_collator():
    from vla_adapter_amd import engine as E, ops
    from vla_adapter_amd.constants import IGNORE_INDEX, NUM_TOKENS
    from vla_adapter_amd.input_stage import GPUInputStage
    st = GPUInputStage(DEV)
    B = 3
    g = torch.Generator().manual_seed(2)
    frames = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)]
    prompts = [list(range(100, 100 + n)) for n in (51, 44, 48)]                   # tokenizer output incl. the 3 trailing ids
    actions = torch.rand(B, 8, 7, generator=g) * 2.4 - 1.2
    batch = st.build(frames, prompts, actions, proprio=torch.rand(B, 8, generator=g), rng=random.Random(7))
    L = 48 + NUM_TOKENS
    assert tuple(batch["input_ids"].shape) == (B, L) and batch["pixel_values"].shape[1] == 3
    bins = np.linspace(-1, 1, 256)
    rr = random.Random(7)
    for b, n in enumerate((48, 41, 45)):
        flat = (151643 - np.digitize(np.clip(actions[b].numpy().reshape(-1), -1.0, 1.0), bins)).tolist()
        want = prompts[b][:-3] + flat + rr.choices(flat, k=NUM_TOKENS - len(flat))
        ids, lab, am = (batch[k][b].cpu().tolist() for k in ("input_ids", "labels", "attention_mask"))
        assert ids[:len(want)] == want and all(t == 151643 for t in ids[len(want):])
        assert am == [True] * len(want) + [False] * (L - len(want))
        assert lab[:n - 1] == [IGNORE_INDEX] * (n - 1) and lab[n - 1:len(want)] == want[n - 1:] and all(t == IGNORE_INDEX for t in lab[len(want):])
    # the masks the model derives from these labels select exactly the 64 action positions of every row
    _, _, cnt = ops.action_mask(batch["labels"], 0)
    assert cnt.cpu().tolist() == [NUM_TOKENS] * B
    # and the batch runs through the native step
    cfg = E.config2()
    assert batch["pixel_values"].dtype == torch.bfloat16 and batch["input_ids"].max().item() < cfg.llm.vocab


@pytest.mark.parametrize("H,W", [(256, 256), (300, 200), (128, 160), (480, 640)])
def test_resize_is_bit_exact_against_pillow(H, W):
    """PrismaticImageProcessor.apply_transform (processing_prismatic.py:128-145): TVF.resize(PIL image, (224, 224), BICUBIC,
    antialias=True) = PIL.Image.resize.  The two device passes reproduce Pillow's uint8 output bit for bit, and the pixel tensor
    built from the resized frames equals ToTensor + Normalize of Pillow's result."""
    import numpy as np
    from PIL import Image
    from vla_adapter_amd.input_stage import GPUInputStage, pil_bicubic_coeffs
    from oracle import vla_oracle as O
    rng = np.random.default_rng(H + W)
    imgs = rng.integers(0, 256, size=(3, H, W, 3), dtype=np.uint8)
    imgs[0, : H // 2] = np.linspace(0, 255, W, dtype=np.uint8)[None, :, None]
    st = GPUInputStage("cuda", backbones=("siglip",))
    got = st.resize(torch.from_numpy(imgs)).cpu().numpy()
    ref = np.stack([np.asarray(Image.fromarray(im).resize((224, 224), resample=Image.BICUBIC)) for im in imgs])
    assert got.shape == ref.shape and np.array_equal(got, ref), f"max |diff| {np.abs(got.astype(int) - ref.astype(int)).max()}"
    b1, c1 = pil_bicubic_coeffs(W, 224)
    b2, c2 = O.pil_bicubic_coeffs(W, 224)
    assert np.array_equal(b1, b2) and np.array_equal(c1, c2)              # product taps == oracle taps (pinned against Pillow on CPU)
    px = st.pixels([torch.from_numpy(imgs)]).float().cpu()
    want = ((torch.from_numpy(ref).permute(0, 3, 1, 2).float() / 255.0) - 0.5) / 0.5
    assert (px - want.to(torch.bfloat16).float()).abs().max().item() == 0.0
