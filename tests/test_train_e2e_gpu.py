"""End-to-end on the GPU: the reference's entry point (`train_halva.train` with the flag set of src/hallava_7b.sh) on a tiny
on-disk checkpoint + JSON dataset + PNG images: model / tokenizer / dataset construction, sampler, engine steps with gradient
accumulation, AdamW + cosine schedule, PEFT-format outputs."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import e2e_util  # noqa: E402


def test_train_entry_point(tmp_path, monkeypatch):
    import llava.train.train_halva as TH
    paths = e2e_util.build(str(tmp_path))
    e2e_util.patch_tokenizer(monkeypatch, paths["vocab_size"])
    out = os.path.join(str(tmp_path), "out")
    argv = ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --deepspeed src/json/zero3.json --loss_alpha 0.4 "
            "--model_name_or_path %s --version v1 --data_path %s --ref_data_path %s --image_folder %s --vision_tower %s "
            "--mm_projector_type mlp2x_gelu --mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False "
            "--image_aspect_ratio pad --group_by_modality_length True --bf16 True --output_dir %s --num_train_epochs 2 "
            "--per_device_train_batch_size 2 --per_device_eval_batch_size 4 --gradient_accumulation_steps 2 --evaluation_strategy no "
            "--save_strategy steps --save_steps 50000 --learning_rate 1e-3 --weight_decay 0. --warmup_ratio 0.03 "
            "--lr_scheduler_type cosine --logging_steps 1 --tf32 True --model_max_length 64 --gradient_checkpointing True "
            "--dataloader_num_workers 0 --lazy_preprocess True --report_to wandb --save_total_limit 1 --run_name e2e"
            % (paths["ckpt"], paths["data"], paths["ref"], paths["images"], paths["vision"], out)).split()
    TH.train(argv)
    state = json.load(open(os.path.join(out, "trainer_state.json")))
    losses = [r["loss"] for r in state["log_history"]]
    assert state["global_step"] == len(losses) >= 2 and all(math.isfinite(l) for l in losses)
    assert losses[-1] < losses[0], losses            # lr 1e-3 on 6 samples: the DPA loss must go down
    adapter = torch.load(os.path.join(out, "adapter_model.bin"))
    k = "base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"
    assert k in adapter and adapter[k].shape == (8, 128)
    assert adapter["base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"].shape == (128, 8)
    assert float(adapter["base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight"].abs().sum()) > 0   # B moved off zero
    nl = torch.load(os.path.join(out, "non_lora_trainables.bin"))
    assert "base_model.model.model.mm_projector.0.weight" in nl
    cfg = json.load(open(os.path.join(out, "adapter_config.json")))
    assert cfg["r"] == 8 and cfg["lora_alpha"] == 16 and set(cfg["target_modules"]) == {"q_proj", "k_proj", "v_proj", "o_proj",
                                                                                           "gate_proj", "up_proj", "down_proj"}
    assert os.path.exists(os.path.join(out, "config.json"))


def test_vila_train_entry_point(tmp_path, monkeypatch):
    """vila.train.train_halva.train with the flag set of src_vila/halva_vila_13b.sh on a VILA-layout tiny checkpoint."""
    import vila.train.train_halva as TV
    paths = e2e_util.build_vila(str(tmp_path))
    e2e_util.patch_tokenizer(monkeypatch, paths["vocab_size"])
    out = os.path.join(str(tmp_path), "out_vila")
    argv = ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --deepspeed src/json/zero3.json --loss_alpha 0.2 "
            "--model_name_or_path %s --version v1 --data_path %s --ref_data_path %s --image_folder %s "
            "--vision_tower google/siglip-so400m-patch14-384 --mm_vision_select_feature cls_patch --mm_projector mlp_downsample "
            "--tune_vision_tower False --tune_mm_projector True --tune_language_model False --mm_vision_select_layer -2 "
            "--mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio resize --bf16 True --output_dir %s "
            "--num_train_epochs 2 --per_device_train_batch_size 2 --per_device_eval_batch_size 4 --gradient_accumulation_steps 2 "
            "--evaluation_strategy no --save_strategy steps --save_steps 50000 --learning_rate 1e-3 --weight_decay 0. "
            "--warmup_ratio 0.03 --lr_scheduler_type cosine --logging_steps 1 --tf32 True --model_max_length 64 "
            "--gradient_checkpointing True --dataloader_num_workers 0 --lazy_preprocess True --report_to wandb "
            "--save_total_limit 1 --vflan_no_system_prompt True --run_name e2e-vila"
            % (paths["ckpt"], paths["data"], paths["ref"], paths["images"], out)).split()
    from unittest import mock
    from vila.train.transformer_normalize_monkey_patch import patched_normalize
    with mock.patch("transformers.image_transforms.normalize", new=patched_normalize):
        TV.train(argv)
    state = json.load(open(os.path.join(out, "trainer_state.json")))
    losses = [r["loss"] for r in state["log_history"]]
    assert state["global_step"] == len(losses) >= 2 and all(math.isfinite(l) for l in losses)
    assert losses[-1] < losses[0], losses
    adapter = torch.load(os.path.join(out, "adapter_model.bin"))
    k = "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.weight"
    assert k in adapter and adapter[k].shape == (8, 128)
    assert float(adapter["llm.base_model.model.model.layers.0.self_attn.q_proj.lora_B.weight"].abs().sum()) > 0
    nl = torch.load(os.path.join(out, "non_lora_trainables.bin"))
    assert set(nl) == {"mm_projector.layers.%s" % s for s in ("1.weight", "1.bias", "2.weight", "2.bias", "4.weight", "4.bias")}
    # --mm_projector_lr 0: the projector receives gradients but must not move (reference optimizer groups)
    from safetensors.torch import load_file
    before = load_file(os.path.join(paths["ckpt"], "mm_projector", "model.safetensors"))
    for kk, v in nl.items():
        assert torch.equal(v, before[kk[len("mm_projector."):]]), kk
    assert os.path.exists(os.path.join(out, "config.json"))


def test_gpu_image_pipeline_matches_cpu_dataset(tmp_path, monkeypatch):
    """--gpu_image_pipeline True: the dataset hands over decoded uint8 images and the trainer's GPU preprocessing yields exactly
    the tensors the reference's CPU path (expand2square + CLIPImageProcessor) produces, bf16-rounded; then a full train() run."""
    import types
    from transformers import CLIPImageProcessor
    import llava.train.train_halva as TH
    from llava import conversation as conv_lib
    from llava.train.halva_trainer import HalvaTrainer
    paths = e2e_util.build(str(tmp_path))
    tok = e2e_util._Tok(model_max_length=64)
    tok.pad_token = tok.unk_token
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    proc = CLIPImageProcessor.from_pretrained(paths["vision"])
    batches = {}
    for flag in (False, True):
        args = types.SimpleNamespace(data_path=paths["data"], ref_data_path=paths["ref"], image_folder=paths["images"],
                                     image_aspect_ratio="pad", image_processor=proc, is_multimodal=True, mm_use_im_start_end=False,
                                     gpu_image_pipeline=flag)
        mod = TH.make_supervised_data_module(tok, args)
        ds = mod["train_dataset"]
        batch = mod["data_collator"]([ds[i] for i in range(4)])
        stub = types.SimpleNamespace(model=types.SimpleNamespace(device=torch.device("cuda")), train_dataset=ds, _pipe=None)
        stub._image_pipeline = types.MethodType(HalvaTrainer._image_pipeline, stub)
        stub._is_raw = HalvaTrainer._is_raw
        batches[flag] = HalvaTrainer._to_device(stub, batch)
    for k in ("images", "ref_images"):
        assert batches[True][k].dtype == torch.bfloat16 and batches[True][k].shape == batches[False][k].shape
        assert torch.equal(batches[True][k].cpu(), batches[False][k].cpu()), k
    e2e_util.patch_tokenizer(monkeypatch, paths["vocab_size"])
    out = os.path.join(str(tmp_path), "out_gpu_pipe")
    argv = ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --loss_alpha 0.4 --model_name_or_path %s --version v1 "
            "--data_path %s --ref_data_path %s --image_folder %s --vision_tower %s --mm_projector_type mlp2x_gelu "
            "--mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad "
            "--group_by_modality_length True --bf16 True --output_dir %s "
            "--num_train_epochs 1 --per_device_train_batch_size 2 --gradient_accumulation_steps 1 --learning_rate 1e-3 "
            "--logging_steps 1 --model_max_length 64 --gpu_image_pipeline True"
            % (paths["ckpt"], paths["data"], paths["ref"], paths["images"], paths["vision"], out)).split()
    TH.train(argv)
    state = json.load(open(os.path.join(out, "trainer_state.json")))
    assert state["global_step"] == 3 and all(math.isfinite(r["loss"]) for r in state["log_history"])


def _argv(paths, out, extra):
    return ("--lora_enable True --lora_r 8 --lora_alpha 16 --mm_projector_lr 0 --loss_alpha 0.4 --model_name_or_path %s --version v1 "
            "--data_path %s --ref_data_path %s --image_folder %s --vision_tower %s --mm_projector_type mlp2x_gelu "
            "--mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad "
            "--group_by_modality_length True --bf16 True --output_dir %s --num_train_epochs 2 --per_device_train_batch_size 2 "
            "--gradient_accumulation_steps 1 --learning_rate 1e-3 --warmup_ratio 0.03 --lr_scheduler_type cosine --logging_steps 1 "
            "--model_max_length 64 %s" % (paths["ckpt"], paths["data"], paths["ref"], paths["images"], paths["vision"], out, extra)).split()


def test_checkpoints_and_resume_continue_the_run(tmp_path, monkeypatch):
    """--save_strategy steps --save_steps 2 writes checkpoint-2/4/6 (adapter in the output format + fp32 master weights, AdamW
    moments, position in the epoch, RNG state); --save_total_limit keeps the newest; a run that finds checkpoint-4 in its
    output_dir (reference train_halva.py:1222-1225 -> trainer.train(resume_from_checkpoint=True)) continues from it: same batches
    in the same order, same schedule, and the tensors of the uninterrupted run (the host logic is checked bit for bit on the CPU,
    tests/test_trainer_loop_cpu.py; here the two GPU runs are compared to bf16 precision)."""
    import shutil
    import llava.train.train_halva as TH
    paths = e2e_util.build(str(tmp_path))
    e2e_util.patch_tokenizer(monkeypatch, paths["vocab_size"], warm_paths=paths)      # token ids independent of the visiting order
    out_a = os.path.join(str(tmp_path), "run_a")
    TH.train(_argv(paths, out_a, "--save_strategy steps --save_steps 2"))
    state_a = json.load(open(os.path.join(out_a, "trainer_state.json")))
    assert state_a["global_step"] == 6                                   # 6 samples / bs 2 = 3 micro-batches x 2 epochs, accum 1
    assert sorted(d for d in os.listdir(out_a) if d.startswith("checkpoint-")) == ["checkpoint-2", "checkpoint-4", "checkpoint-6"]
    for f in ("adapter_model.bin", "non_lora_trainables.bin", "config.json", "halva_state.pt", "trainer_state.json"):
        assert os.path.exists(os.path.join(out_a, "checkpoint-4", f)), f
    # the final artefacts equal the last checkpoint's adapter
    fin = torch.load(os.path.join(out_a, "adapter_model.bin"))
    ck6 = torch.load(os.path.join(out_a, "checkpoint-6", "adapter_model.bin"))
    assert all(torch.equal(fin[k], ck6[k]) for k in fin)
    # interrupted run: only checkpoint-4 survives
    out_b = os.path.join(str(tmp_path), "run_b")
    os.makedirs(out_b)
    shutil.copytree(os.path.join(out_a, "checkpoint-4"), os.path.join(out_b, "checkpoint-4"))
    TH.train(_argv(paths, out_b, "--save_strategy steps --save_steps 2 --save_total_limit 1"))
    state_b = json.load(open(os.path.join(out_b, "trainer_state.json")))
    assert state_b["global_step"] == 6
    assert any("resumed_from" in r for r in state_b["log_history"])
    la, lb = ([r["loss"] for r in st["log_history"] if "loss" in r] for st in (state_a, state_b))
    assert lb[:4] == la[:4] and len(lb) == 6                               # the history up to the checkpoint travels with it
    assert max(abs(x - y) for x, y in zip(la, lb)) < 2e-3, (la, lb)
    assert [r["learning_rate"] for r in state_b["log_history"] if "loss" in r] == [r["learning_rate"] for r in state_a["log_history"]]
    assert sorted(d for d in os.listdir(out_b) if d.startswith("checkpoint-")) == ["checkpoint-6"]          # --save_total_limit 1
    got = torch.load(os.path.join(out_b, "adapter_model.bin"))
    assert set(got) == set(fin)
    print("resumed run bitwise equal to the uninterrupted one:", all(torch.equal(got[k], fin[k]) for k in fin))
    for k in fin:
        assert float((got[k].float() - fin[k].float()).norm()) <= 2e-2 * float(fin[k].float().norm()) + 1e-6, k
    nl_a = torch.load(os.path.join(out_a, "non_lora_trainables.bin"))
    nl_b = torch.load(os.path.join(out_b, "non_lora_trainables.bin"))
    assert all(torch.equal(nl_a[k], nl_b[k]) for k in nl_a)                # --mm_projector_lr 0: the projector never moves
    # resume_from_checkpoint with nothing to resume from fails loudly instead of silently restarting
    from llava.train.halva_trainer import HalvaTrainer
    t = HalvaTrainer.__new__(HalvaTrainer)
    t.args = type("A", (), dict(output_dir=os.path.join(str(tmp_path), "empty")))()
    assert t._checkpoint_dirs() == []
