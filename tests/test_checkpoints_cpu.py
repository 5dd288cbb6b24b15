"""Checkpoint interop: key maps between the reference's state-dict layouts and the engine's weight dict."""
import torch

from vla_adapter_amd import checkpoints as C


class _V:          # minimal stand-ins for engine.VLACfg on a CPU-only box (no GPU needed for key handling)
    def __init__(self, fused):
        self.fused = fused


def _hf_sd(fused):
    sd = {"vision_backbone.featurizer.blocks.0.attn.qkv.weight": torch.ones(2), "vision_backbone.featurizer.pos_embed": torch.zeros(3),
          "vision_backbone.featurizer.blocks.0.ls1.scale_factor": torch.ones(1),
          "projector.fc1.weight": torch.ones(4), "projector.fc2.bias": torch.ones(5),
          "language_model.model.embed_tokens.weight": torch.ones(6), "language_model.model.norm.weight": torch.ones(7),
          "language_model.model.layers.0.self_attn.q_proj.bias": torch.ones(8), "language_model.lm_head.weight": torch.ones(9),
          "action_queries.weight": torch.ones(10)}
    if fused:
        sd["vision_backbone.fused_featurizer.blocks.0.mlp.fc1.weight"] = torch.ones(11)
    return sd


def test_split_and_merge_roundtrip():
    for fused in (False, True):
        sd = _hf_sd(fused)
        W = C.split_reference_state_dict(sd, _V(fused))
        assert len(W["vit"]) == (2 if fused else 1)
        assert set(W["vit"][0]) == {"blocks.0.attn.qkv.weight", "pos_embed", "blocks.0.ls1.scale_factor"}
        assert set(W["proj"]) == {"fc1.weight", "fc2.bias"}
        assert set(W["llm"]) == {"embed_tokens.weight", "norm.weight", "layers.0.self_attn.q_proj.bias"}      # lm_head is not part of the path
        assert W["action_queries"].numel() == 10
        back = C.merge_reference_state_dict(W, _V(fused))
        assert set(back) == set(sd) - {"language_model.lm_head.weight"}
        assert all(torch.equal(back[k], sd[k]) for k in back)


def test_native_prismatic_keys_follow_the_reference_rename_map():
    native = {"vision_backbone.dino_featurizer.blocks.1.ls2.gamma": torch.ones(1), "vision_backbone.siglip_featurizer.pos_embed": torch.ones(2),
              "llm_backbone.llm.model.norm.weight": torch.ones(3), "llm_backbone.llm.model.embed_tokens.weight": torch.ones(3),
              "projector.projector.0.weight": torch.ones(4), "projector.projector.2.bias": torch.ones(5), "projector.projector.4.weight": torch.ones(6)}
    r = C.rename_prismatic_keys(native)
    assert set(r) == {"vision_backbone.featurizer.blocks.1.ls2.scale_factor", "vision_backbone.fused_featurizer.pos_embed",
                      "language_model.model.norm.weight", "language_model.model.embed_tokens.weight",
                      "projector.fc1.weight", "projector.fc2.bias", "projector.fc3.weight"}
    W = C.split_reference_state_dict({"module." + k: v for k, v in native.items()}, _V(True))     # DDP prefix + native names
    assert "blocks.1.ls2.scale_factor" in W["vit"][0] and "pos_embed" in W["vit"][1] and "fc3.weight" in W["proj"]


def test_run_dir_files_roundtrip(tmp_path):
    head = {"module.model.fc1.weight": torch.ones(3), "module.model.mlp_resnet_blocks.0.q_proj.bias": torch.zeros(2)}
    pp = {"fc1.weight": torch.ones(4)}
    torch.save(head, tmp_path / "action_head--150_checkpoint.pt")
    torch.save(pp, tmp_path / "proprio_projector--150_checkpoint.pt")
    h, p = C.load_run_dir(str(tmp_path), 150)
    assert set(h) == {"model.fc1.weight", "model.mlp_resnet_blocks.0.q_proj.bias"} and set(p) == {"fc1.weight"}
    from safetensors.torch import save_file
    save_file({"a.b": torch.arange(3.0)}, str(tmp_path / "m.safetensors"))
    assert torch.equal(C.load_file(str(tmp_path / "m.safetensors"))["a.b"], torch.arange(3.0))


def test_infer_config_reads_the_geometry_off_a_state_dict():
    """finetune.py picks the model from the checkpoint (VERDICT r2 #1c): synthetic CPU weights of every named geometry, exported
    under the reference's HF key names, give the same geometry back - incl. the DINOv2 prefix tokens / LayerScale, the fused
    3-layer projector and the 1.5B head dim."""
    from vla_adapter_amd import engine as E, synthetic as S
    for name in ("tiny", "tiny_fused"):
        cfg = E.NAMED_CONFIGS[name]()
        W = S.make_weights(cfg, "cpu", seed=1)
        sd = C.merge_reference_state_dict(W, cfg)
        got = C.infer_config({"module." + k: v for k, v in sd.items()})
        assert [v.as_oracle() for v in got.vit] == [v.as_oracle() for v in cfg.vit] and [v.img for v in got.vit] == [v.img for v in cfg.vit]
        assert got.llm == cfg.llm and got.num_blocks == cfg.num_blocks and got.fused == cfg.fused
    # full-size geometries from shapes alone (meta tensors: no memory)
    real_rn = S._rn
    S._rn = lambda gen, shape, std, device: torch.empty(*shape, device="meta", dtype=torch.bfloat16)
    try:
        sds = {name: C.merge_reference_state_dict(S.make_weights(E.NAMED_CONFIGS[name](), "cpu", seed=0), E.NAMED_CONFIGS[name]())
               for name in ("config2", "dinosiglip-0_5b", "config5")}
    finally:
        S._rn = real_rn
    for name, sd in sds.items():
        cfg = E.NAMED_CONFIGS[name]()
        got = C.infer_config(sd)
        assert [v.as_oracle() for v in got.vit] == [v.as_oracle() for v in cfg.vit] and got.llm == cfg.llm and got.num_blocks == 24


def test_offline_lora_merge_matches_w_plus_scaled_ba():
    """merge_lora_weights_and_save.py:44-103: W + (alpha / r) B A for every adapted Linear, everything else untouched."""
    g = torch.Generator().manual_seed(0)
    base = {"language_model.model.layers.0.self_attn.q_proj.weight": torch.randn(16, 8, generator=g).to(torch.bfloat16),
            "language_model.model.layers.0.self_attn.q_proj.bias": torch.randn(16, generator=g).to(torch.bfloat16),
            "projector.fc1.weight": torch.randn(4, 8, generator=g).to(torch.bfloat16)}
    A, B = torch.randn(2, 8, generator=g).to(torch.bfloat16), torch.randn(16, 2, generator=g).to(torch.bfloat16)
    pre = "base_model.model.language_model.model.layers.0.self_attn.q_proj."
    merged = C.merge_lora_into_state_dict(base, {pre + "lora_A.weight": A, pre + "lora_B.weight": B})
    k = "language_model.model.layers.0.self_attn.q_proj.weight"
    assert torch.equal(merged[k], (base[k].float() + 2.0 * (B.float() @ A.float())).to(torch.bfloat16)) and merged[k].dtype == torch.bfloat16
    assert all(torch.equal(merged[n], base[n]) for n in base if n != k)
    import pytest
    with pytest.raises(KeyError):
        C.merge_lora_into_state_dict(base, {"base_model.model.nope.lora_A.weight": A, "base_model.model.nope.lora_B.weight": B})
    # the scale is lora_alpha / r of adapter_config.json: alpha 3, r 2 -> 1.5; peft's in-memory key form carries the adapter name
    live = {pre + "lora_A.default.weight": A, pre + "lora_B.default.weight": B}
    m2 = C.merge_lora_into_state_dict(base, live, adapter_config=dict(r=2, lora_alpha=3))
    assert torch.equal(m2[k], (base[k].float() + 1.5 * (B.float() @ A.float())).to(torch.bfloat16))
    with pytest.raises(ValueError):
        C.merge_lora_into_state_dict(base, live, scaling=2.0, adapter_config=dict(r=2, lora_alpha=3))
    assert C.lora_scaling(None) == 2.0 and C.lora_scaling(dict(r=64, lora_alpha=128)) == 2.0 and C.lora_scaling({}, 4.0) == 4.0


def test_load_lora_adapter_returns_the_config(tmp_path):
    import json
    from safetensors.torch import save_file
    A = torch.ones(2, 8, dtype=torch.bfloat16)
    save_file({"base_model.model.x.lora_A.weight": A}, str(tmp_path / "adapter_model.safetensors"))
    sd, cfg = C.load_lora_adapter(str(tmp_path), with_config=True)
    assert cfg == {} and torch.equal(sd["base_model.model.x.lora_A.weight"], A)
    json.dump(dict(r=8, lora_alpha=4), open(tmp_path / "adapter_config.json", "w"))
    _, cfg = C.load_lora_adapter(str(tmp_path), with_config=True)
    assert C.lora_scaling(cfg) == 0.5
