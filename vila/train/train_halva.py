"""train() of the VILA twin (reference vila/train/train_halva.py) on the MI355X DPA path.

Keeps the reference's flags (src_vila/halva_vila_13b.sh:30-68), dataset semantics and output artefacts; the model is
halva_amd.vila_model.VilaLlavaLlamaModel (no peft / DeepSpeed / bitsandbytes / flash-attn), the step is
halva_amd.dpa.DPAEngine, gradients are all-reduced over RCCL (halva_amd/dp.py).
"""
import copy
import json
import os
import pathlib
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch

from llava import conversation as conversation_lib
from llava.constants import IGNORE_INDEX
from llava.train import train_halva as _L
from llava.train.train_halva import (DataCollatorForHallDataset, _str2bool, parse_args_into_dataclasses,  # noqa: F401
                                     preprocess_multimodal, preprocess_v1_ref, split_string_by_mask_and_tokenize,
                                     tokenizer_image_token_masked)
from vila.mm_utils import is_gemma_tokenizer, process_image
from vila.model import LlavaLlamaConfig, LlavaLlamaModel
from vila.train.halva_trainer import HalvaTrainer

local_rank = None


def mprint(*args, **kw):
    if local_rank in (0, -1, None):
        print(*args, **kw)


@dataclass
class DataArguments:
    data_path: str = field(default=None)
    ref_data_path: str = field(default=None)
    lazy_preprocess: bool = False
    is_multimodal: bool = False
    image_folder: Optional[str] = field(default=None)
    image_aspect_ratio: str = "square"
    data_mixture: str = "llava_1_5_mm_align"
    eval_data_mixture: Optional[str] = None
    vflan_no_system_prompt: bool = False
    downsample_video: bool = False
    num_video_frames: int = 8
    gpu_image_pipeline: bool = field(default_factory=lambda: os.environ.get("HALVA_GPU_IMAGE_PIPELINE", "0") == "1")   # see llava twin


@dataclass
class ModelArguments:
    version: Optional[str] = field(default="v0")
    model_name_or_path: Optional[str] = field(default="facebook/opt-125m")
    vision_tower: Optional[str] = field(default="google/siglip-so400m-patch14-384")
    mm_projector: Optional[str] = field(default="mlp2x_gelu")
    mm_use_im_start_end: bool = field(default=False)
    mm_use_im_patch_token: bool = field(default=True)
    mm_vision_select_layer: Optional[int] = field(default=-1)
    mm_vision_select_feature: Optional[str] = field(default="patch")
    vision_resolution: Optional[int] = field(default=-1)
    interpolate_mode: Optional[str] = field(default="linear")
    drop_path_rate: Optional[float] = field(default=0.0)
    s2: bool = field(default=False)
    s2_scales: Optional[str] = field(default="336,672,1008")
    s2_max_split_size: int = field(default=336)
    loss_alpha: Optional[float] = field(default=0.0)


@dataclass
class TrainingArguments(_L.TrainingArguments):
    """reference :99-144: the LLaVA arguments plus the tune_* switches, model_dtype and the SLURM time limits."""
    tune_vision_tower: bool = field(default=False)
    tune_language_model: bool = field(default=False)
    tune_mm_projector: bool = field(default=False)
    model_dtype: str = field(default="torch.bfloat16")
    total_time_limit: int = field(default=-1)
    pre_terminate_time: int = field(default=10)
    data_seed: Optional[int] = None


def find_all_linear_names(model):
    """reference :212-225 over the Llama linears (vision tower / projector excluded, lm_head removed)."""
    return ["q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"]


# ------------------------------------------------------------------------------------------------
def preprocess_v1(sources, tokenizer, has_image: bool = False, no_system_prompt: bool = False) -> Dict:
    """reference :623-747; identical to the LLaVA twin except that rounds after the first shrink by one token for
    non-gemma tokenizers (:717-726) - a no-op for HALVA's single-turn samples."""
    return _L.preprocess_v1(sources, tokenizer, has_image=has_image, no_system_prompt=no_system_prompt,
                            later_round_shrink=0 if is_gemma_tokenizer(tokenizer) else 1)


def preprocess_v1_ref_vila(sources, tokenizer, has_image: bool = False, no_system_prompt: bool = False) -> Dict:
    return preprocess_v1_ref(sources, tokenizer, has_image=has_image, no_system_prompt=no_system_prompt,
                             later_round_shrink=0 if is_gemma_tokenizer(tokenizer) else 1)


class HallDataset(_L.HallDataset):
    """reference :867-1155.  Differences from the LLaVA dataset, kept: images go through process_image (aspect ratio
    'resize' for SigLIP) and come back [n, 3, H, W] (n = 1 unless the sample lists several files); the reference sample's
    image is read from the TRAINING sample of the same index (:1111, `self.list_data_dict[i]["image"]`)."""

    def _images(self, image_file):
        if getattr(self.data_args, "gpu_image_pipeline", False):
            import numpy as np
            from PIL import Image
            if isinstance(image_file, list):
                raise NotImplementedError("gpu_image_pipeline with several images per sample")
            img = Image.open(self.get_image_file_path(image_file)).convert("RGB")
            return torch.from_numpy(np.asarray(img).copy())               # [H, W, 3] uint8 -> GPU pipeline in the trainer
        if isinstance(image_file, list):
            return torch.stack([process_image(self.get_image_file_path(f), self.data_args, self.data_args.image_folder)
                                for f in image_file])
        return process_image(self.get_image_file_path(image_file), self.data_args, self.data_args.image_folder)

    def _blank(self):
        proc = self.data_args.image_processor
        cs = proc.crop_size if getattr(proc, "crop_size", None) else proc.size
        return torch.zeros(1, 3, cs["height"], cs["width"])

    @property
    def lengths(self):
        return [sum(len(c["value"].split()) for c in s["conversations"]) + (128 if "image" in s else 0)
                for s in self.list_data_dict]

    def __getitem__(self, i) -> Dict[str, torch.Tensor]:
        pos, neg = self.list_data_dict[i], self.neg_list_data_dict[i]
        assert pos["id"] == neg["id"]
        has_image = "image" in pos
        if has_image:
            image = self._images(pos["image"])
            p_src = preprocess_multimodal(copy.deepcopy([pos["conversations"]]), self.data_args)
            n_src = preprocess_multimodal(copy.deepcopy([neg["conversations"]]), self.data_args)
        else:
            p_src, n_src = copy.deepcopy([pos["conversations"]]), copy.deepcopy([neg["conversations"]])
        p = preprocess_v1(p_src, self.tokenizer, has_image=has_image)
        n = preprocess_v1(n_src, self.tokenizer, has_image=("image" in neg))
        item = dict(input_ids=p["input_ids"][0], labels=p["labels"][0], neg_input_ids=n["input_ids"][0], neg_labels=n["labels"][0],
                    pos_signs=p["signs"][0], neg_signs=n["signs"][0])
        raw = getattr(self.data_args, "gpu_image_pipeline", False)
        item["image"] = (image if (raw or image.ndim == 4) else image.unsqueeze(0)) if has_image else self._blank()
        if self.ref_data_dict is not None:
            r = self.ref_getitem(i)
            item["ref_input_ids"], item["ref_labels"], item["ref_image"] = r["input_ids"], r["labels"], r["image"]
        else:
            item["ref_input_ids"], item["ref_labels"], item["ref_image"] = item["input_ids"], item["labels"], item["image"]
        return item

    def ref_getitem(self, i) -> Dict[str, torch.Tensor]:
        s = self.ref_data_dict[i]
        has_image = "image" in s
        if has_image:
            image = self._images(self.list_data_dict[i]["image"])          # sic: the training sample's image
            src = preprocess_multimodal(copy.deepcopy([s["conversations"]]), self.data_args)
        else:
            src = copy.deepcopy([s["conversations"]])
        d = preprocess_v1_ref_vila(src, self.tokenizer, has_image=has_image)
        out = dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
        raw = getattr(self.data_args, "gpu_image_pipeline", False)
        out["image"] = (image if (raw or image.ndim == 4) else image.unsqueeze(0)) if has_image else self._blank()
        return out


def make_supervised_data_module(tokenizer, data_args) -> Dict:
    ds = HallDataset(tokenizer=tokenizer, data_path=data_args.data_path, ref_data_path=data_args.ref_data_path, data_args=data_args)
    return dict(train_dataset=ds, eval_dataset=None, data_collator=DataCollatorForHallDataset(tokenizer=tokenizer))


# ------------------------------------------------------------------------------------------------
# output artefacts (reference :1356-1373): config.json, the LoRA adapter of `model.llm`, non_lora_trainables.bin
# ------------------------------------------------------------------------------------------------
def get_peft_state_maybe_zero_3(model, bias="none"):
    """{`llm.base_model.model.<hf name>.lora_{A,B}.weight`} - named_parameters() of the VILA wrapper whose `.llm` is a
    PeftModel (reference :165-188 called with model.named_parameters(), :1357-1359), adapter name stripped by peft on save."""
    out = {}
    for i, layer in enumerate(model.get_llm().model.layers):
        for sub, grp in layer.groups():
            for k, v in grp.lora_state().items():
                out["llm.base_model.model.model.layers.%d.%s.%s" % (i, sub, k.replace(".default", ""))] = v.detach().cpu().clone()
    return out


def get_peft_state_non_lora_maybe_zero_3(model, require_grad_only=True):
    return {"mm_projector." + k: v.detach().cpu().clone() for k, v in model.get_mm_projector().named_parameters()
            if v.requires_grad or not require_grad_only}


def save_lora_outputs(model, training_args):
    out = training_args.output_dir
    os.makedirs(out, exist_ok=True)
    model.config.save_pretrained(out)
    torch.save(get_peft_state_maybe_zero_3(model, training_args.lora_bias), os.path.join(out, "adapter_model.bin"))
    with open(os.path.join(out, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": training_args.lora_r, "lora_alpha": training_args.lora_alpha,
                   "lora_dropout": training_args.lora_dropout, "bias": training_args.lora_bias, "fan_in_fan_out": False,
                   "target_modules": find_all_linear_names(model), "inference_mode": True, "modules_to_save": None,
                   "base_model_name_or_path": getattr(model.config, "_name_or_path", None), "init_lora_weights": True}, f, indent=2)
    torch.save(get_peft_state_non_lora_maybe_zero_3(model), os.path.join(out, "non_lora_trainables.bin"))


# ------------------------------------------------------------------------------------------------
def prepare_config_for_training(config, model_args, training_args, data_args):
    """reference vila/train/utils.py:65-95"""
    assert model_args.vision_tower is not None, "requires vision tower"
    if getattr(config, "llm_cfg", None) is None:
        config.llm_cfg = model_args.model_name_or_path
    if getattr(config, "vision_tower_cfg", None) is None:
        config.vision_tower_cfg = model_args.vision_tower
    if getattr(config, "mm_projector_cfg", None) is None:
        config.mm_projector_cfg = model_args.mm_projector
    config.model_dtype = "torch.bfloat16" if training_args.bf16 else "torch.float16"
    config.tune_language_model = training_args.tune_language_model
    config.tune_vision_tower = training_args.tune_vision_tower
    config.tune_mm_projector = training_args.tune_mm_projector
    config.image_aspect_ratio = data_args.image_aspect_ratio
    config.mm_vision_select_layer = model_args.mm_vision_select_layer
    config.mm_vision_select_feature = model_args.mm_vision_select_feature
    config.vision_resolution, config.interpolate_mode = model_args.vision_resolution, model_args.interpolate_mode
    config.drop_path_rate, config.s2 = model_args.drop_path_rate, model_args.s2
    config.s2_scales, config.s2_max_split_size = model_args.s2_scales, model_args.s2_max_split_size


def setup_model(model_args, data_args, training_args):
    """reference :268-485 without bitsandbytes / peft / DeepSpeed."""
    from halva_amd.llama import add_lora
    if training_args.bits != 16:
        raise NotImplementedError("4/8-bit loading (bitsandbytes) is not part of the MI355X DPA path; use --bits 16")
    if not training_args.bf16:
        raise NotImplementedError("the MI355X DPA path computes in bf16 (--bf16 True, as src_vila/halva_vila_13b.sh:52)")
    name = model_args.model_name_or_path
    if any(k in name.lower() for k in ("mpt", "mistral", "mixtral", "gemma")):
        raise NotImplementedError(name)
    if training_args.tune_vision_tower or training_args.tune_language_model:
        raise NotImplementedError("the HALVA recipe freezes the vision tower and the language model (LoRA only): "
                                  "--tune_vision_tower / --tune_language_model True have no backward on this path")
    if model_args.vision_resolution not in (-1, None):
        raise NotImplementedError("--vision_resolution (position-embedding interpolation) is not on the HALVA path")
    config = LlavaLlamaConfig.from_pretrained(name)
    if getattr(config, "resume_path", None) is not None:
        config.resume_path = name
    prepare_config_for_training(config, model_args, training_args, data_args)
    from halva_amd.dp import local_device_index
    dev = torch.device("cuda", local_device_index())
    model = LlavaLlamaModel(config=config, attn_implementation="flash_attention_2", model_max_length=training_args.model_max_length,
                            cache_dir=training_args.cache_dir, device=dev)
    model.llm.config.use_cache = False
    model.get_llm().requires_grad_(training_args.tune_language_model)
    model.get_vision_tower().requires_grad_(training_args.tune_vision_tower)
    model.get_mm_projector().requires_grad_(training_args.tune_mm_projector)
    mprint(f"Tunable parameters:\nlanguage model {training_args.tune_language_model}\nvision tower "
           f"{training_args.tune_vision_tower}\nmm projector {training_args.tune_mm_projector}")
    if training_args.lora_enable:
        mprint("Adding LoRA adapters...")
        proj_grad = [p.requires_grad for p in model.get_mm_projector().parameters()]
        add_lora(model.llm, training_args.lora_r, training_args.lora_alpha)       # peft on model.llm only (:403-405)
        for p, g in zip(model.get_mm_projector().parameters(), proj_grad):
            p.requires_grad_(g)
    else:
        model._use_lora = False
    tokenizer = model.tokenizer
    if tokenizer is None:
        raise FileNotFoundError("no tokenizer under %s/llm" % name)
    if model_args.version == "v0":
        raise NotImplementedError("version v0 (pad-token embedding resize) is not on the HALVA path")
    tokenizer.pad_token = tokenizer.unk_token
    conversation_lib.default_conversation = conversation_lib.conv_templates.get(model_args.version,
                                                                                conversation_lib.conv_templates["vicuna_v1"])
    model.llm.pad_token_id = tokenizer.pad_token_id
    model.llm.config.tokenizer_padding_side = tokenizer.padding_side
    model.llm.config.tokenizer_model_max_length = tokenizer.model_max_length
    vt = model.get_vision_tower()
    if vt is not None:
        data_args.image_processor = vt.image_processor
        data_args.is_multimodal = True
        model.config.num_video_frames = data_args.num_video_frames
        model.config.image_aspect_ratio = data_args.image_aspect_ratio
        model.config.mm_use_im_start_end = data_args.mm_use_im_start_end = model_args.mm_use_im_start_end
        model.config.mm_projector_lr = training_args.mm_projector_lr
        training_args.use_im_start_end = model_args.mm_use_im_start_end
        model.config.mm_use_im_patch_token = model_args.mm_use_im_patch_token
        model.initialize_vision_tokenizer(model_args, tokenizer=tokenizer)
    return model, tokenizer


def train(argv=None):
    global local_rank
    model_args, data_args, training_args = parse_args_into_dataclasses((ModelArguments, DataArguments, TrainingArguments), argv)
    training_args.run_name = training_args.output_dir.split("/")[-1]
    local_rank = training_args.local_rank if training_args.local_rank >= 0 else int(os.environ.get("LOCAL_RANK", "-1"))
    assert model_args.version in ["v1"], "This code supports llama3 conversation template."
    ref_model_args, ref_data_args, ref_training_args = (copy.deepcopy(x) for x in (model_args, data_args, training_args))
    torch.manual_seed(training_args.seed)
    mprint("Loading online model")
    model, tokenizer = setup_model(model_args, data_args, training_args)
    mprint(f"Loading reference model: {ref_model_args.model_name_or_path}")
    ref_training_args.lora_enable = False
    ref_model, _ = setup_model(ref_model_args, ref_data_args, ref_training_args)
    for p in ref_model.parameters():
        p.requires_grad = False
    data_module = make_supervised_data_module(tokenizer=tokenizer, data_args=data_args)
    trainer = HalvaTrainer(model=model, tokenizer=tokenizer, args=training_args, **data_module)
    trainer.custom_setup(model=model, ref_model=ref_model, label_pad_token_id=IGNORE_INDEX, padding_value=tokenizer.pad_token_id,
                         loss_alpha=model_args.loss_alpha)
    print("length of dataloader:", len(trainer.get_train_dataloader()), len(trainer.train_dataset), flush=True)
    print("[GPU memory] before trainer", torch.cuda.memory_allocated() / 1024 / 1024 / 1024, flush=True)
    trainer.train(resume_from_checkpoint=False)
    trainer.save_state()
    model.llm.config.use_cache = True
    if training_args.lora_enable and trainer.dist.rank == 0:
        save_lora_outputs(model, training_args)


if __name__ == "__main__":
    train()
