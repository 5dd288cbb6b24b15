"""CPU checks of the host-side mirror of the reference API (no GPU, no compute calls)."""
import dataclasses
import inspect
import re

import torch

from vla_adapter_amd import constants, finetune as F, train_utils
from oracle import vla_oracle as O

REF_FIELDS = """config_file_path vlm_path use_minivlm resum_vla_path data_root_dir dataset_name run_root_dir shuffle_buffer_size
use_l1_regression use_diffusion num_diffusion_steps use_film num_images_in_input use_proprio phase1_path batch_size learning_rate
lr_warmup_steps num_steps_before_decay grad_accumulation_steps max_steps use_val_set val_freq val_time_limit save_freq
save_latest_checkpoint_only resume resume_step image_aug diffusion_sample_freq use_lora lora_rank lora_dropout
merge_lora_during_training use_fz wandb_entity wandb_project run_id_note run_id_override wandb_log_freq use_pro_version phase""".split()


def test_finetune_config_keeps_every_reference_flag():
    names = [f.name for f in dataclasses.fields(F.FinetuneConfig)]
    for n in REF_FIELDS:                       # vla-scripts/finetune.py:66-128
        assert n in names, n
    cfg = F.parse_args(["--batch_size", "32", "--learning_rate", "2e-4", "--use_pro_version", "True", "--phase", "Inference"])
    assert cfg.batch_size == 32 and cfg.learning_rate == 2e-4 and cfg.use_pro_version is True and cfg.phase == "Inference"
    assert cfg.lr_warmup_steps == 0.1 and cfg.num_steps_before_decay == 100000      # reference defaults


def test_lr_schedule_matches_oracle():
    cfg = F.FinetuneConfig()
    for step in (0, 1, 99999, 100000, 150000):
        assert F.lr_at(step, cfg) == O.lr_at(step, cfg.learning_rate, cfg.lr_warmup_steps, cfg.num_steps_before_decay)


def test_constants_and_masks_match_reference_semantics():
    assert (constants.IGNORE_INDEX, constants.ACTION_TOKEN_BEGIN_IDX, constants.NUM_TOKENS) == (-100, 151386, 64)
    assert constants.detect_robot_platform(["x", "--dataset", "aloha_thing"]) == "ALOHA"
    assert constants.detect_robot_platform(["finetune.py"]) == "LIBERO"
    import numpy as np, os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "masks.npz"))
    lab = torch.from_numpy(z["labels"])
    assert torch.equal(train_utils.get_current_action_mask(lab), torch.from_numpy(z["cur"]).bool())
    assert torch.equal(train_utils.get_next_actions_mask(lab), torch.from_numpy(z["nxt"]).bool())


def test_forward_signatures_match_reference():
    from vla_adapter_amd import modeling_prismatic as M
    ref = ("input_ids attention_mask pixel_values labels inputs_embeds past_key_values use_cache output_attentions "
           "output_hidden_states output_projector_features return_dict proprio proprio_projector noisy_actions "
           "noisy_action_projector diffusion_timestep_embeddings use_film").split()        # modeling_prismatic.py:525-544
    assert list(inspect.signature(M.OpenVLAForActionPrediction.forward).parameters)[1:] == ref
    ref2 = ("input_ids attention_mask pixel_values labels inputs_embeds past_key_values use_cache output_attentions "
            "output_hidden_states return_dict multimodal_indices").split()                 # vlms/prismatic.py:312-325
    assert list(inspect.signature(M.PrismaticVLM.forward).parameters)[1:] == ref2
    from vla_adapter_amd.action_heads import L1RegressionActionHead
    sig = list(inspect.signature(L1RegressionActionHead.predict_action).parameters)[1:5]
    assert sig == ["actions_hidden_states", "proprio", "proprio_projector", "phase"]       # action_heads.py:43-49


def test_product_package_never_imports_the_oracle():
    import glob, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in glob.glob(os.path.join(root, "vla_adapter_amd", "*.py")):
        src = open(f).read()
        if os.path.basename(f) == "smoke.py":
            continue                            # smoke() is an allowed checker (called only by __graft_entry__.smoke)
        assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
