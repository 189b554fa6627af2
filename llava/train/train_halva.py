"""Training driver of the HALVA DPA path with the reference's surface (llava/train/train_halva.py) on the MI355X-native
engine: the same argument dataclasses / flags (every flag of src/hallava_7b.sh:30-69 is accepted), the masked
tokenisation walk, label masking, HallDataset, DataCollatorForHallDataset, setup_llava and train().

Host-side integer work (ids / labels / signs / masks) is bit-exact with the reference (tests/test_host_logic.py replays
the golden vectors the reference produced).  ZeRO / DeepSpeed / wandb / tf32 flags are accepted and degrade to the
replica + single-all-reduce design; nothing here touches CUDA-only APIs.
"""
import argparse
import copy
import dataclasses
import json
import os
import pathlib
import random
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch
from torch.utils.data import Dataset

from llava import conversation as conversation_lib
from llava.constants import (DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IGNORE_INDEX, IMAGE_TOKEN_INDEX)
from llava.mm_utils import tokenizer_image_token
from llava.model import LlavaLlamaForCausalLM
from llava.train.halva_trainer import HalvaTrainer

MASK_PLACEHOLDER_START = "<MASK>"
MASK_PLACEHOLDER_END = "</MASK>"
local_rank = None


def rank0_print(*args):
    if local_rank in (0, -1, None):
        print(*args)


# ------------------------------------------------------------------------------------------------
# arguments (reference train_halva.py:41-100; TrainingArguments no longer inherits transformers', whose 5.x version
# dropped --evaluation_strategy / --warmup_ratio and validates --tf32 against CUDA: SURVEY.md 8b "arg-parsing trap")
# ------------------------------------------------------------------------------------------------
@dataclass
class ModelArguments:
    model_name_or_path: Optional[str] = field(default="facebook/opt-125m")
    version: Optional[str] = field(default="v0")
    freeze_backbone: bool = field(default=False)
    tune_mm_mlp_adapter: bool = field(default=False)
    vision_tower: Optional[str] = field(default=None)
    mm_vision_select_layer: Optional[int] = field(default=-1)
    pretrain_mm_mlp_adapter: Optional[str] = field(default=None)
    mm_projector_type: Optional[str] = field(default="linear")
    mm_use_im_start_end: bool = field(default=False)
    mm_use_im_patch_token: bool = field(default=True)
    mm_vision_select_feature: Optional[str] = field(default="patch")
    loss_alpha: Optional[float] = field(default=0.0)


@dataclass
class DataArguments:
    data_path: str = field(default=None)
    ref_data_path: str = field(default=None)
    lazy_preprocess: bool = False
    is_multimodal: bool = False
    image_folder: Optional[str] = field(default=None)
    image_aspect_ratio: str = "square"
    # MI355X addition (default off = the reference's CPU preprocessing): the dataset hands over DECODED uint8 images and the
    # trainer runs expand2square / bicubic resize / crop / normalise on the GPU (halva_amd/image_pipeline.py, bit-exact)
    gpu_image_pipeline: bool = field(default_factory=lambda: os.environ.get("HALVA_GPU_IMAGE_PIPELINE", "0") == "1")


@dataclass
class TrainingArguments:
    output_dir: str = field(default="./output")
    cache_dir: Optional[str] = field(default=None)
    optim: str = field(default="adamw_torch")
    remove_unused_columns: bool = field(default=False)
    freeze_mm_mlp_adapter: bool = field(default=False)
    mpt_attn_impl: Optional[str] = field(default="triton")
    model_max_length: int = field(default=512)
    double_quant: bool = field(default=True)
    quant_type: str = field(default="nf4")
    bits: int = field(default=16)
    lora_enable: bool = False
    lora_r: int = 64
    lora_alpha: int = 16
    lora_dropout: float = 0.05
    lora_weight_path: str = ""
    lora_bias: str = "none"
    mm_projector_lr: Optional[float] = None
    group_by_modality_length: bool = field(default=False)
    # the subset of transformers.TrainingArguments the launch scripts use
    num_train_epochs: float = 3.0
    max_steps: int = -1
    per_device_train_batch_size: int = 8
    per_device_eval_batch_size: int = 8
    gradient_accumulation_steps: int = 1
    evaluation_strategy: str = "no"
    save_strategy: str = "steps"
    save_steps: int = 500
    save_total_limit: Optional[int] = None
    learning_rate: float = 5e-5
    weight_decay: float = 0.0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 1.0          # accepted; effectively unused by the reference recipe
    warmup_ratio: float = 0.0
    lr_scheduler_type: str = "linear"
    logging_steps: int = 500
    bf16: bool = False
    fp16: bool = False
    tf32: Optional[bool] = None         # accept-and-ignore (Ampere TF32 switch)
    gradient_checkpointing: bool = False
    dataloader_num_workers: int = 0
    dataloader_drop_last: bool = False
    report_to: str = "none"             # accept-and-degrade: metrics are printed as JSON lines on rank 0
    run_name: Optional[str] = None
    deepspeed: Optional[str] = None     # JSON accepted; ZeRO stage ignored (full replicas + one all-reduce)
    local_rank: int = -1
    seed: int = 42
    fsdp: str = ""
    device: str = "cuda"

    @property
    def world_size(self):
        return int(os.environ.get("WORLD_SIZE", "1"))

    @property
    def train_batch_size(self):
        return self.per_device_train_batch_size


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("true", "1", "yes", "y", "t"):
        return True
    if v.lower() in ("false", "0", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError("expected a boolean, got %r" % v)


def parse_args_into_dataclasses(classes, argv=None):
    """HfArgumentParser.parse_args_into_dataclasses for plain dataclasses (`--flag True` booleans included)."""
    parser = argparse.ArgumentParser(allow_abbrev=False)
    owners = {}
    for cls in classes:
        for f in dataclasses.fields(cls):
            default = None if f.default is dataclasses.MISSING else f.default
            if f.default is dataclasses.MISSING and f.default_factory is not dataclasses.MISSING:
                default = f.default_factory()
            base = f.type
            if getattr(base, "__origin__", None) is not None:               # Optional[X]
                base = [a for a in base.__args__ if a is not type(None)][0]
            kind = _str2bool if base is bool else (base if base in (int, float, str) else str)
            kw = dict(type=kind, default=default)
            if base is bool:
                kw.update(nargs="?", const=True)
            parser.add_argument("--" + f.name, **kw)
            owners[f.name] = cls
    ns, unknown = parser.parse_known_args(argv)
    if unknown:
        raise ValueError("Some specified arguments are not used by the argument parser: %s" % unknown)
    out = []
    for cls in classes:
        out.append(cls(**{f.name: getattr(ns, f.name) for f in dataclasses.fields(cls)}))
    return tuple(out)


def find_all_linear_names(model):
    """LoRA targets = every Llama linear except lm_head / vision / projector (reference train_halva.py:156-169)."""
    from halva_amd.llama import LORA_TARGETS
    return list(LORA_TARGETS)


# ------------------------------------------------------------------------------------------------
# masked tokenisation (reference train_halva.py:236-363)
# ------------------------------------------------------------------------------------------------
def preprocess_multimodal(sources: Sequence[str], data_args) -> Dict:
    if not data_args.is_multimodal:
        return sources
    for source in sources:
        for turn in source:
            text = turn["value"]
            if DEFAULT_IMAGE_TOKEN in text:
                text = (DEFAULT_IMAGE_TOKEN + "\n" + text.replace(DEFAULT_IMAGE_TOKEN, "").strip()).strip()
            token = DEFAULT_IMAGE_TOKEN
            if getattr(data_args, "mm_use_im_start_end", False):
                token = DEFAULT_IM_START_TOKEN + token + DEFAULT_IM_END_TOKEN
            turn["value"] = text.replace(DEFAULT_IMAGE_TOKEN, token)
    return sources


def _tail(tokenizer, text, drop_lead):
    """ids of `text` without BOS (and, if drop_lead == 2, without the lone word-boundary piece) and without the last piece."""
    return tokenizer(text).input_ids[drop_lead:-1]


def split_string_by_mask_and_tokenize(string, tokenizer):
    """Token ids and phrase ids for the part of the prompt after `<image>`.  Tagged phrases are numbered 1, 2, ...;
    everything else - including the '.', ',' or "'s" glued behind a phrase - is 0."""
    ids, signs = [], []
    pos, phrase = 0, 1
    n_open, n_close = len(MASK_PLACEHOLDER_START), len(MASK_PLACEHOLDER_END)
    while True:
        a = string.find(MASK_PLACEHOLDER_START, pos)
        if a == -1:
            ids += _tail(tokenizer, string[pos:], 2)
            signs += [0] * (len(ids) - len(signs))
            return ids, signs
        b = string.find(MASK_PLACEHOLDER_END, a + n_open)
        ids += _tail(tokenizer, string[pos:a], 1 if pos == 0 else 2)
        signs += [0] * (len(ids) - len(signs))
        body, nxt = string[a + n_open:b], b + n_close
        one, two = string[nxt:nxt + 1], string[nxt:nxt + 2]
        if one in ".,":                       # also true for the empty string at end-of-text, as in the reference
            ids += _tail(tokenizer, (body + one).replace(" .", ". ").replace(" ,", ", "), 2)
            signs += [phrase] * (len(ids) - len(signs) - 1) + [0]
            pos = nxt + 1
        elif two == "'s":
            ids += _tail(tokenizer, (body + two).replace(" 's", "'s "), 2)
            signs += [phrase] * (len(ids) - len(signs) - 1) + [0]
            pos = nxt + 2
        else:
            ids += _tail(tokenizer, body, 2)
            signs += [phrase] * (len(ids) - len(signs))
            pos = nxt
        phrase += 1


def tokenizer_image_token_masked(prompt, tokenizer, image_token_index=IMAGE_TOKEN_INDEX, return_tensors=None):
    parts = prompt.split("<image>")
    assert len(parts) == 2, "assuming the only users give image, and it is a single turn conversation"
    head, rest = parts
    assert MASK_PLACEHOLDER_START not in head
    ids = list(tokenizer(head).input_ids) + [image_token_index]
    signs = [0] * len(ids)
    r_ids, r_signs = split_string_by_mask_and_tokenize(rest, tokenizer)
    ids += r_ids + [tokenizer.eos_token_id]
    signs += r_signs + [0]
    if return_tensors is None:
        return ids, signs
    if return_tensors == "pt":
        return torch.tensor(ids, dtype=torch.long), torch.tensor(signs, dtype=torch.long)
    raise ValueError(f"Unsupported tensor type: {return_tensors}")


def _render(conv, roles, source):
    if roles[source[0]["from"]] != conv.roles[0]:
        source = source[1:]
    conv.messages = []
    for j, turn in enumerate(source):
        role = roles[turn["from"]]
        assert role == conv.roles[j % 2]
        conv.append_message(role, turn["value"])
    return conv.get_prompt()


def _mask_targets(targets, conversations, conv, tokenizer, has_image, later_round_shrink=0):
    """labels := ids with BOS and every round's instruction part set to IGNORE_INDEX (reference :432-473).
    later_round_shrink: VILA's twin takes 1 token off the round / instruction length of every round after the first
    for non-gemma tokenizers (vila/train/train_halva.py:717-726)."""
    sep = conv.sep + conv.roles[1] + ": "
    for conversation, target in zip(conversations, targets):
        total_len = int(target.ne(tokenizer.pad_token_id).sum())
        cur = 1
        target[:cur] = IGNORE_INDEX
        for i_round, rou in enumerate(conversation.split(conv.sep2)):
            if rou == "":
                break
            parts = rou.split(sep)
            if len(parts) != 2:
                break
            instr = parts[0] + sep
            if has_image:
                round_len = len(tokenizer_image_token(rou, tokenizer))
                instr_len = len(tokenizer_image_token(instr, tokenizer)) - 2
            else:
                round_len = len(tokenizer(rou).input_ids)
                instr_len = len(tokenizer(instr).input_ids) - 2
            if i_round > 0:
                round_len, instr_len = round_len - later_round_shrink, instr_len - later_round_shrink
            target[cur:cur + instr_len] = IGNORE_INDEX
            cur += round_len
        target[cur:] = IGNORE_INDEX
        if cur < tokenizer.model_max_length and cur != total_len:
            target[:] = IGNORE_INDEX
            print(f"WARNING: tokenization mismatch: {cur} vs. {total_len}. (ignored)")


def preprocess_v1(sources, tokenizer, has_image: bool = False, no_system_prompt: bool = False, later_round_shrink: int = 0) -> Dict:
    """sources[0] = [human, gpt (masked answer), gpt-ref (plain answer)].  Returns input_ids / labels / signs [1, L],
    or None when the masked tokenisation differs from the plain one (the caller then fails, as in the reference).
    no_system_prompt / later_round_shrink: the VILA twin's extras (vila/train/train_halva.py:623-634,717-726)."""
    assert has_image, "this code may not be ready to handle non image setup"
    conv = conversation_lib.default_conversation.copy()
    if no_system_prompt:
        conv.system = ""
    roles = {"human": conv.roles[0], "gpt": conv.roles[1]}
    turns = copy.deepcopy(sources[0])
    assert turns[2]["from"] == "gpt-ref"
    plain = [turns[0], dict(turns[2], **{"from": "gpt"})]
    masked = turns[:2]
    plain_prompt = _render(conv, roles, plain)
    ref_ids = torch.stack([tokenizer_image_token(plain_prompt, tokenizer, return_tensors="pt")], dim=0)
    masked_prompt = _render(conv, roles, masked)
    ids, signs = tokenizer_image_token_masked(masked_prompt, tokenizer, return_tensors="pt")
    ids, signs = ids.unsqueeze(0), signs.unsqueeze(0)
    if (ids != ref_ids).sum() > 0:          # raises on a length mismatch exactly like the reference's comparison
        print("conversations: ", [masked_prompt])
        print("ref_conversations", [plain_prompt])
        print(f"[Error in tokenization] input_ids: {ids}, ref_input_ids: {ref_ids}")
        return None
    targets = ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    _mask_targets(targets, [plain_prompt], conv, tokenizer, has_image, later_round_shrink)
    return dict(input_ids=ids, labels=targets, signs=signs)


def preprocess_v1_ref(sources, tokenizer, has_image: bool = False, no_system_prompt: bool = False,
                      later_round_shrink: int = 0) -> Dict:
    conv = conversation_lib.default_conversation.copy()
    if no_system_prompt:
        conv.system = ""
    roles = {"human": conv.roles[0], "gpt": conv.roles[1]}
    prompts = [_render(conv, roles, s) for s in sources]
    if has_image:
        ids = torch.stack([tokenizer_image_token(p, tokenizer, return_tensors="pt") for p in prompts], dim=0)
    else:
        ids = tokenizer(prompts, return_tensors="pt", padding="longest", max_length=tokenizer.model_max_length,
                        truncation=True).input_ids
    targets = ids.clone()
    assert conv.sep_style == conversation_lib.SeparatorStyle.TWO
    _mask_targets(targets, prompts, conv, tokenizer, has_image, later_round_shrink)
    return dict(input_ids=ids, labels=targets)


# ------------------------------------------------------------------------------------------------
# dataset + collator (reference train_halva.py:565-993)
# ------------------------------------------------------------------------------------------------
def _expand2square(img, fill):
    from PIL import Image
    w, h = img.size
    if w == h:
        return img
    side = max(w, h)
    canvas = Image.new(img.mode, (side, side), fill)
    canvas.paste(img, ((side - w) // 2, (side - h) // 2))
    return canvas


class HallDataset(Dataset):
    """Pairs of (correct, hallucinated) conversations + an independent reference sample per index."""

    def __init__(self, data_path, ref_data_path, tokenizer, data_args):
        super().__init__()
        self.data_args, self.tokenizer = data_args, tokenizer
        self.list_data_dict, self.neg_list_data_dict = self.prepare_data_dict(data_path)
        if ref_data_path in (None, "none"):
            rank0_print(f"we will use {data_path} as reference")
            self.ref_data_dict = None
        else:
            rank0_print(f"we will use {ref_data_path} as reference")
            self.ref_data_dict = self.get_ref_data_dict(ref_data_path, len(self.list_data_dict))
            assert len(self.list_data_dict) == len(self.neg_list_data_dict) == len(self.ref_data_dict)
        srcs = ("textvqa", "gqa", "vg", "coco", "ocr_vqa")
        if data_args.image_folder == "default":         # the reference's hard-coded cluster layout (:591-598)
            base = {"textvqa": "/h/anonymous/anonymous_ssd004/datasets/textvqa", "gqa": "/h/anonymous/anonymous_ssd004/datasets/gqa",
                    "vg": "/h/anonymous/anonymous_ssd004/datasets/vg/images", "coco": "/scratch/ssd004/datasets/MSCOCO2017",
                    "ocr_vqa": "/h/anonymous/anonymous_ssd004/datasets/ocr_vqa"}
            self.IMAGE_DIRS = base
        else:
            self.IMAGE_DIRS = {s: data_args.image_folder + s for s in srcs}

    def get_ref_data_dict(self, data_path, num_samples):
        with open(data_path) as f:
            data = json.load(f)
        assert len(data) > num_samples
        return data[:num_samples]

    def get_image_file_path(self, image_file):
        src, _, rest = image_file.partition("/")
        return os.path.join(self.IMAGE_DIRS[src], rest)

    def prepare_data_dict(self, data_path):
        with open(data_path) as f:
            data = json.load(f)
        closed = [s for s in data if s["tag"] == "closed"]
        opened = [s for s in data if s["tag"] == "open"]
        qa = [s for s in data if s["tag"] == "qa"]
        rank0_print(f"number of closed set samples {len(closed)}")
        rank0_print(f"number of open set samples {len(opened)}")
        random.seed(42)                                   # yes/no balancing (:647-657)
        random.shuffle(qa)
        yes = [k for k in qa if k["raw_answer"].lower() == "yes"]
        no = [k for k in qa if k["raw_answer"].lower() == "no"]
        n = min(len(yes), len(no))
        qa = yes[:n] + no[:n]
        rank0_print(f"number of qa samples {len(qa)}")
        data = closed + opened + qa
        random.seed(42)
        random.shuffle(data)
        rank0_print(f"current data size {len(data)}")
        pos, neg = [], []
        for s in data:
            def conv(masked, plain):
                return [{"from": "human", "value": s["question"]}, {"from": "gpt", "value": masked},
                        {"from": "gpt-ref", "value": plain}]
            pos.append({"conversations": conv(s["correct_answer_masked"], s["correct_answer"]), "id": s["id"], "image": s["image"]})
            neg.append({"conversations": conv(s["hallucinated_answer_masked"], s["hallucinated_answer"]), "id": s["id"],
                        "image": s["image"]})
        return pos, neg

    def __len__(self):
        return len(self.list_data_dict)

    @property
    def lengths(self):
        return [sum(len(c["value"].split()) for c in s["conversations"]) + (128 if "image" in s else 0) for s in self.list_data_dict]

    @property
    def modality_lengths(self):
        out = []
        for s in self.list_data_dict:
            n = sum(len(c["value"].split()) for c in s["conversations"])
            out.append(n if "image" in s else -n)
        return out

    def _load_image(self, rel):
        from PIL import Image
        proc = self.data_args.image_processor
        img = Image.open(self.get_image_file_path(rel)).convert("RGB")
        if getattr(self.data_args, "gpu_image_pipeline", False):
            import numpy as np
            return torch.from_numpy(np.asarray(img).copy())           # [H, W, 3] uint8; preprocessed on the GPU by the trainer
        if self.data_args.image_aspect_ratio == "pad":
            img = _expand2square(img, tuple(int(x * 255) for x in proc.image_mean))
        return proc.preprocess(img, return_tensors="pt")["pixel_values"][0]

    def __getitem__(self, i) -> Dict[str, torch.Tensor]:
        pos, neg = self.list_data_dict[i], self.neg_list_data_dict[i]
        assert pos["id"] == neg["id"]
        has_image = "image" in pos
        if has_image:
            image = self._load_image(pos["image"])
            p_src = preprocess_multimodal(copy.deepcopy([pos["conversations"]]), self.data_args)
            n_src = preprocess_multimodal(copy.deepcopy([neg["conversations"]]), self.data_args)
        else:
            p_src, n_src = copy.deepcopy([pos["conversations"]]), copy.deepcopy([neg["conversations"]])
        p = preprocess_v1(p_src, self.tokenizer, has_image=has_image)
        n = preprocess_v1(n_src, self.tokenizer, has_image=has_image)
        item = dict(input_ids=p["input_ids"][0], labels=p["labels"][0], neg_input_ids=n["input_ids"][0], neg_labels=n["labels"][0],
                    pos_signs=p["signs"][0], neg_signs=n["signs"][0])
        if has_image:
            item["image"] = image
        elif self.data_args.is_multimodal:
            cs = self.data_args.image_processor.crop_size
            item["image"] = torch.zeros(3, cs["height"], cs["width"])
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
            image = self._load_image(s["image"])
            src = preprocess_multimodal(copy.deepcopy([s["conversations"]]), self.data_args)
        else:
            src = copy.deepcopy([s["conversations"]])
        d = preprocess_v1_ref(src, self.tokenizer, has_image=has_image)
        out = dict(input_ids=d["input_ids"][0], labels=d["labels"][0])
        if has_image:
            out["image"] = image
        elif self.data_args.is_multimodal:
            cs = self.data_args.image_processor.crop_size
            out["image"] = torch.zeros(3, cs["height"], cs["width"])
        return out


@dataclass
class DataCollatorForHallDataset(object):
    """Right-pad the eight id/label/sign lists, truncate to model_max_length, build the three attention masks
    (ids != pad_token_id) and stack images (reference train_halva.py:896-993)."""
    tokenizer: object

    def __call__(self, instances: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        pad = self.tokenizer.pad_token_id
        limit = self.tokenizer.model_max_length
        fills = (("input_ids", pad), ("labels", IGNORE_INDEX), ("neg_input_ids", pad), ("neg_labels", IGNORE_INDEX),
                 ("pos_signs", 0), ("neg_signs", 0), ("ref_input_ids", pad), ("ref_labels", IGNORE_INDEX))
        out = {}
        for key, fill in fills:
            seqs = [torch.as_tensor(x[key]) for x in instances]
            out[key] = torch.nn.utils.rnn.pad_sequence(seqs, batch_first=True, padding_value=fill)[:, :limit]
        batch = dict(input_ids=out["input_ids"], labels=out["labels"], attention_mask=out["input_ids"].ne(pad),
                     neg_input_ids=out["neg_input_ids"], neg_labels=out["neg_labels"],
                     neg_attention_mask=out["neg_input_ids"].ne(pad), pos_signs=out["pos_signs"], neg_signs=out["neg_signs"],
                     ref_input_ids=out["ref_input_ids"], ref_labels=out["ref_labels"],
                     ref_attention_mask=out["ref_input_ids"].ne(pad))
        for src, dst in (("image", "images"), ("ref_image", "ref_images")):
            if src in instances[0]:
                ims = [x[src] for x in instances]
                same = all(im is not None and im.shape == ims[0].shape for im in ims)
                batch[dst] = torch.stack([torch.as_tensor(im) for im in ims]) if same else ims
        return batch


def make_supervised_data_module(tokenizer, data_args) -> Dict:
    ds = HallDataset(tokenizer=tokenizer, data_path=data_args.data_path, ref_data_path=data_args.ref_data_path, data_args=data_args)
    return dict(train_dataset=ds, eval_dataset=None, data_collator=DataCollatorForHallDataset(tokenizer=tokenizer))


# ------------------------------------------------------------------------------------------------
# output artefacts (reference train_halva.py:1011-1027,1230-1240): PEFT adapter + non_lora_trainables.bin + config.json
# ------------------------------------------------------------------------------------------------
def get_peft_state_maybe_zero_3(model, bias="none"):
    """{`base_model.model.<hf name>.lora_{A,B}.weight`: tensor} in PEFT's adapter_model.bin naming."""
    out = {}
    for i, layer in enumerate(model.get_model().layers):
        for sub, grp in layer.groups():
            for k, v in grp.lora_state().items():
                out["base_model.model.model.layers.%d.%s.%s" % (i, sub, k.replace(".default", ""))] = v.detach().cpu().clone()
    return out


def get_peft_state_non_lora_maybe_zero_3(model, require_grad_only=True):
    proj = model.get_model().mm_projector
    return {"base_model.model.model.mm_projector." + k: v.detach().cpu().clone() for k, v in proj.state_dict().items()}


def save_lora_outputs(model, training_args):
    os.makedirs(training_args.output_dir, exist_ok=True)
    model.config.save_pretrained(training_args.output_dir)
    torch.save(get_peft_state_maybe_zero_3(model, training_args.lora_bias), os.path.join(training_args.output_dir, "adapter_model.bin"))
    with open(os.path.join(training_args.output_dir, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": training_args.lora_r, "lora_alpha": training_args.lora_alpha,
                   "lora_dropout": training_args.lora_dropout, "bias": training_args.lora_bias, "fan_in_fan_out": False,
                   "target_modules": find_all_linear_names(model), "base_model_name_or_path": getattr(model.config, "_name_or_path", None),
                   "inference_mode": True, "modules_to_save": None, "init_lora_weights": True}, f, indent=2)
    torch.save(get_peft_state_non_lora_maybe_zero_3(model), os.path.join(training_args.output_dir, "non_lora_trainables.bin"))


class SaverCallback:
    def on_train_end(self, args, state, control, **kwargs):
        if getattr(args, "lora_enable", False) and int(os.environ.get("RANK", "0")) == 0:
            save_lora_outputs(kwargs["model"], args)


# ------------------------------------------------------------------------------------------------
def setup_llava(model_args, data_args, training_args):
    """Model + tokenizer construction (reference train_halva.py:1029-1176) without bitsandbytes / peft / DeepSpeed."""
    from halva_amd.llama import add_lora
    if training_args.bits != 16:
        raise NotImplementedError("4/8-bit loading (bitsandbytes) is not part of the MI355X DPA path; use --bits 16")
    if not training_args.bf16:
        raise NotImplementedError("the MI355X DPA path computes in bf16 (--bf16 True, as src/hallava_7b.sh:48)")
    from halva_amd.dp import local_device_index
    dev = torch.device("cuda", local_device_index())
    model = LlavaLlamaForCausalLM.from_pretrained(model_args.model_name_or_path, cache_dir=training_args.cache_dir, device=dev)
    model.config._name_or_path = model_args.model_name_or_path
    model.config.use_cache = False
    if training_args.gradient_checkpointing:
        model.enable_input_require_grads()
    if training_args.lora_enable:
        rank0_print("Adding LoRA adapters...")
        add_lora(model, training_args.lora_r, training_args.lora_alpha)
    import transformers
    tokenizer = transformers.AutoTokenizer.from_pretrained(model_args.model_name_or_path, cache_dir=training_args.cache_dir,
                                                           model_max_length=training_args.model_max_length, padding_side="right",
                                                           use_fast=False)
    tokenizer.pad_token = tokenizer.unk_token
    conversation_lib.default_conversation = conversation_lib.conv_templates.get(model_args.version,
                                                                                conversation_lib.conv_templates["vicuna_v1"])
    if model_args.vision_tower is not None:
        model.get_model().initialize_vision_modules(model_args=model_args, fsdp=training_args.fsdp)
        vt = model.get_vision_tower()
        data_args.image_processor = vt.image_processor
        data_args.is_multimodal = True
        cfg = model.config
        cfg.image_aspect_ratio = data_args.image_aspect_ratio
        cfg.tokenizer_padding_side = tokenizer.padding_side
        cfg.tokenizer_model_max_length = tokenizer.model_max_length
        cfg.tune_mm_mlp_adapter = training_args.tune_mm_mlp_adapter = model_args.tune_mm_mlp_adapter
        cfg.freeze_mm_mlp_adapter = training_args.freeze_mm_mlp_adapter
        if training_args.freeze_mm_mlp_adapter:
            for p in model.get_model().mm_projector.parameters():
                p.requires_grad = False
        cfg.mm_use_im_start_end = data_args.mm_use_im_start_end = model_args.mm_use_im_start_end
        cfg.mm_projector_lr = training_args.mm_projector_lr
        training_args.use_im_start_end = model_args.mm_use_im_start_end
        cfg.mm_use_im_patch_token = model_args.mm_use_im_patch_token
        model.initialize_vision_tokenizer(model_args, tokenizer=tokenizer)
    return model, tokenizer


def train(argv=None):
    global local_rank
    model_args, data_args, training_args = parse_args_into_dataclasses((ModelArguments, DataArguments, TrainingArguments), argv)
    local_rank = training_args.local_rank if training_args.local_rank >= 0 else int(os.environ.get("LOCAL_RANK", "-1"))
    assert model_args.version in ["v1", "vicuna_v1"], "This code supports v1 and vicuna_v1 conversation template."
    torch.manual_seed(training_args.seed)      # the adapter's random initialisation follows --seed (and is rank 0's on every replica)
    ref_model_args, ref_data_args, ref_training_args = (copy.deepcopy(x) for x in (model_args, data_args, training_args))
    rank0_print("Loading online model")
    model, tokenizer = setup_llava(model_args, data_args, training_args)
    rank0_print(f"Loading reference model: {ref_model_args.model_name_or_path}")
    ref_training_args.lora_enable = False
    ref_model, _ = setup_llava(ref_model_args, ref_data_args, ref_training_args)
    for p in ref_model.parameters():
        p.requires_grad = False
    ref_model._use_lora = False
    data_module = make_supervised_data_module(tokenizer=tokenizer, data_args=data_args)
    trainer = HalvaTrainer(model=model, ref_model=ref_model, tokenizer=tokenizer, loss_alpha=model_args.loss_alpha,
                           args=training_args, label_pad_token_id=IGNORE_INDEX, padding_value=tokenizer.pad_token_id, **data_module)
    trainer.add_callback(SaverCallback())
    resume = bool(list(pathlib.Path(training_args.output_dir).glob("checkpoint-*")))
    trainer.train(resume_from_checkpoint=resume)
    trainer.save_state()
    model.config.use_cache = True
    if training_args.lora_enable and trainer.dist.rank == 0:
        save_lora_outputs(model, training_args)


if __name__ == "__main__":
    train()
