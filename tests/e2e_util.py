"""Builds a tiny on-disk LLaVA checkpoint + dataset for end-to-end tests of the train() entry point."""
import json
import os

import numpy as np
import torch

from fake_tokenizer import FakeLlamaTokenizer
from golden_util import load_npz, meta_of, tensors

SAMPLES = [
    ("closed", "What is in the picture?", "There is a <MASK> dog </MASK> sitting on <MASK> the grass </MASK>.",
     "There is a <MASK> cat </MASK> sitting on <MASK> the sand </MASK>."),
    ("open", "Describe the scene.", "A man rides a <MASK> brown horse </MASK> near <MASK> two </MASK> trees.",
     "A man rides a <MASK> white horse </MASK> near <MASK> three </MASK> trees."),
    ("qa", "Is there a cat?", "<MASK> Yes </MASK>, there is a cat.", "<MASK> No </MASK>, there is a cat."),
    ("qa", "Is the door open?", "<MASK> No </MASK>, the door is closed.", "<MASK> Yes </MASK>, the door is closed."),
    ("closed", "What colour is the bus?", "The bus is <MASK> blue </MASK>.", "The bus is <MASK> green </MASK>."),
    ("open", "What is he doing?", "He is <MASK> surfing </MASK> on a <MASK> big wave </MASK>.",
     "He is <MASK> skiing </MASK> on a <MASK> big hill </MASK>."),
]
REF = [("What is shown here?", "A plate of food with rice."), ("Is it raining?", "No"), ("Describe the image.", "Two children play football."),
       ("What is this?", "A red car on a road."), ("Who is there?", "A man and a dog."), ("What colour?", "Blue and green."),
       ("Anything else?", "Nothing else."), ("Where?", "On the left side.")]


def plain(masked):
    s = masked.replace(" </MASK> ", " ").replace(" </MASK>", "").replace(" <MASK> ", " ")
    return s[len("<MASK> "):] if s.startswith("<MASK> ") else s


def build(root, n_samples=None):
    """n_samples: more than len(SAMPLES) repeats them under new ids / images (the 8-rank test needs 8 micro-batches per optimizer step)"""
    from PIL import Image
    from safetensors.torch import save_file
    z = load_npz("dpa_step_d64_init.npz")
    cfg, ccfg = meta_of(z, "llama_cfg"), meta_of(z, "clip_cfg")
    ck, vt, data, img = (os.path.join(root, d) for d in ("ckpt", "vision", "data", "images/"))
    for d in (ck, vt, data, os.path.join(img, "coco")):
        os.makedirs(d, exist_ok=True)
    conf = dict(cfg, model_type="llava", mm_vision_tower=vt, mm_projector_type="mlp2x_gelu", mm_hidden_size=ccfg["hidden_size"],
                mm_vision_select_layer=-2, mm_vision_select_feature="patch")
    json.dump(conf, open(os.path.join(ck, "config.json"), "w"))
    save_file({k: v.to(torch.bfloat16).contiguous() for k, v in tensors(z, "base.").items()}, os.path.join(ck, "model.safetensors"))
    json.dump({"vision_config": ccfg}, open(os.path.join(vt, "config.json"), "w"))
    save_file({"vision_model." + k: v.to(torch.bfloat16).contiguous() for k, v in tensors(z, "clip.").items()},
              os.path.join(vt, "model.safetensors"))
    json.dump({"image_processor_type": "CLIPImageProcessor", "do_resize": True, "size": {"shortest_edge": ccfg["image_size"]},
               "do_center_crop": True, "crop_size": {"height": ccfg["image_size"], "width": ccfg["image_size"]}, "do_normalize": True,
               "do_rescale": True, "do_convert_rgb": True, "image_mean": [0.48145466, 0.4578275, 0.40821073],
               "image_std": [0.26862954, 0.26130258, 0.27577711], "resample": 3},
              open(os.path.join(vt, "preprocessor_config.json"), "w"))
    rng = np.random.RandomState(0)
    rows, refs = [], []
    samples = SAMPLES if n_samples is None else [SAMPLES[i % len(SAMPLES)] for i in range(n_samples)]
    for i, (tag, q, pos, neg) in enumerate(samples):
        name = "coco/im%d.png" % i
        Image.fromarray(rng.randint(0, 255, (24 + 3 * i, 30, 3), dtype=np.uint8)).save(os.path.join(img, name))
        rows.append({"id": i, "image": name, "tag": tag, "raw_answer": "yes" if "Yes" in pos else "no", "question": "<image>\n" + q,
                     "correct_answer": plain(pos), "correct_answer_masked": pos, "hallucinated_answer": plain(neg),
                     "hallucinated_answer_masked": neg})
    ref_rows = REF if n_samples is None else [REF[i % len(REF)] for i in range(max(len(REF), n_samples + 2))]      # (the reference asserts len(ref) > len(data))
    for i, (q, a) in enumerate(ref_rows):
        refs.append({"id": "r%d" % i, "image": "coco/im%d.png" % (i % len(SAMPLES)),
                     "conversations": [{"from": "human", "value": "<image>\n" + q}, {"from": "gpt", "value": a}]})
    json.dump(rows, open(os.path.join(data, "data.json"), "w"))
    json.dump(refs, open(os.path.join(data, "ref_data.json"), "w"))
    return dict(ckpt=ck, vision=vt, data=os.path.join(data, "data.json"), ref=os.path.join(data, "ref_data.json"), images=img,
                vocab_size=cfg["vocab_size"])


def warm_vocab(tok, paths):
    """The stand-in tokenizer numbers pieces in first-seen order: fix that order by walking the fixture dataset once in index
    order, so that token ids do not depend on which sample a run visits first - two runs, or a resumed run, then see the same ids.
    The vocabulary is frozen afterwards: an unseen piece raises."""
    import llava.train.train_halva as TH
    from llava import conversation as conv_lib
    from transformers import CLIPImageProcessor
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    args = TH.DataArguments(data_path=paths["data"], ref_data_path=paths["ref"], image_folder=paths["images"], image_aspect_ratio="pad")
    args.image_processor = CLIPImageProcessor.from_pretrained(paths["vision"])
    args.is_multimodal, args.mm_use_im_start_end = True, False
    keep, tok.pad_token = tok.pad_token, tok.unk_token
    ds = TH.make_supervised_data_module(tok, args)["train_dataset"]
    for i in range(len(ds)):
        ds[i]
    tok.pad_token = keep
    tok.frozen = True


class _Tok(FakeLlamaTokenizer):
    unk_token = "<unk>"
    pad_token = None


def patch_tokenizer(monkeypatch, max_vocab, warm_paths=None):
    """transformers 5.x no longer ships the legacy slow Llama tokenizer the reference's span walk relies on: use the
    deterministic stand-in (tests/golden/fake_tokenizer.py) for the end-to-end tests."""
    import transformers

    def from_pretrained(path, cache_dir=None, model_max_length=2048, padding_side="right", use_fast=False, **kw):
        t = _Tok(model_max_length=model_max_length)
        t.padding_side = padding_side
        t.max_vocab = max_vocab
        if warm_paths is not None:
            warm_vocab(t, warm_paths)
        return t
    monkeypatch.setattr(transformers.AutoTokenizer, "from_pretrained", staticmethod(from_pretrained))


def build_vila(root, multi_image=False):
    """A VILA-layout checkpoint (root/{config.json, llm/, vision_tower/, mm_projector/}) from the vila_step_init fixture
    + the same JSON dataset / PNG images as build()."""
    from PIL import Image
    from safetensors.torch import save_file
    z = load_npz("vila_step_init.npz")
    cfg, vcfg = meta_of(z, "llama_cfg"), meta_of(z, "vis_cfg")
    ck, data, img = (os.path.join(root, d) for d in ("vila_ckpt", "data", "images/"))
    llm, vt, mp = (os.path.join(ck, d) for d in ("llm", "vision_tower", "mm_projector"))
    for d in (llm, vt, mp, data, os.path.join(img, "coco")):
        os.makedirs(d, exist_ok=True)
    bf = lambda d: {k: v.to(torch.bfloat16).contiguous() for k, v in d.items()}
    json.dump(dict(cfg, model_type="llama", architectures=["LlamaForCausalLM"]), open(os.path.join(llm, "config.json"), "w"))
    save_file(bf(tensors(z, "llm.")), os.path.join(llm, "model.safetensors"))
    json.dump(dict(vcfg, model_type="siglip_vision_model", architectures=["SiglipVisionModel"]),
              open(os.path.join(vt, "config.json"), "w"))
    save_file(bf(tensors(z, "vis.")), os.path.join(vt, "model.safetensors"))          # keys already carry `vision_model.`
    json.dump({"image_processor_type": "SiglipImageProcessor", "do_resize": True,
               "size": {"height": vcfg["image_size"], "width": vcfg["image_size"]}, "do_normalize": True, "do_rescale": True,
               "rescale_factor": 1 / 255, "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5], "resample": 3},
              open(os.path.join(vt, "preprocessor_config.json"), "w"))
    json.dump({"mm_projector_type": "mlp_downsample", "model_type": "v2l_projector", "architectures": ["MultimodalProjector"]},
              open(os.path.join(mp, "config.json"), "w"))
    save_file(bf(tensors(z, "proj.")), os.path.join(mp, "model.safetensors"))
    json.dump({"model_type": "llava_llama", "architectures": ["LlavaLlamaModel"], "llm_cfg": dict(cfg, model_type="llama"),
               "vision_tower_cfg": dict(vcfg, model_type="siglip_vision_model"),
               "mm_projector_cfg": {"mm_projector_type": "mlp_downsample"}, "hidden_size": cfg["hidden_size"],
               "mm_hidden_size": vcfg["hidden_size"], "model_dtype": "torch.bfloat16", "resume_path": ck},
              open(os.path.join(ck, "config.json"), "w"))
    rng = np.random.RandomState(0)
    rows, refs = [], []
    for i, (tag, q, pos, neg) in enumerate(SAMPLES):
        name = "coco/im%d.png" % i
        shape = (24 + 3 * i, 30, 3) if i != 2 else (26, 26)              # one single-channel image: patched_normalize path
        Image.fromarray(rng.randint(0, 255, shape, dtype=np.uint8)).save(os.path.join(img, name))
        rows.append({"id": i, "image": name, "tag": tag, "raw_answer": "yes" if "Yes" in pos else "no", "question": "<image>\n" + q,
                     "correct_answer": plain(pos), "correct_answer_masked": pos, "hallucinated_answer": plain(neg),
                     "hallucinated_answer_masked": neg})
    for i, (q, a) in enumerate(REF):
        refs.append({"id": "r%d" % i, "image": "coco/im%d.png" % (i % len(SAMPLES)),
                     "conversations": [{"from": "human", "value": "<image>\n" + q}, {"from": "gpt", "value": a}]})
    json.dump(rows, open(os.path.join(data, "data.json"), "w"))
    json.dump(refs, open(os.path.join(data, "ref_data.json"), "w"))
    return dict(ckpt=ck, data=os.path.join(data, "data.json"), ref=os.path.join(data, "ref_data.json"), images=img,
                vocab_size=cfg["vocab_size"], image_size=vcfg["image_size"])
