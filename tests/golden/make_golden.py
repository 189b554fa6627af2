#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (CPU, this container only).

Test infrastructure.  Runs only where /root/reference is mounted; the fixtures it writes (inputs +
the reference's own outputs) are committed, the reference never travels.  Recipe for importing the
reference under transformers 5.x follows SURVEY.md Appendix A:
  1. neutralise the reference's AutoConfig/AutoModel registration (model_type "llava" is taken),
  2. re-create two names transformers.trainer no longer exports,
  3. stub `peft` (not installed; only names are needed),
  4. call HalvaTrainer's loss methods unbound on a SimpleNamespace (no HF Trainer construction).

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz / *.json)
"""
import copy
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HALVA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

from fake_tokenizer import FakeLlamaTokenizer  # noqa: E402


def import_reference():
    from transformers import AutoConfig, AutoModelForCausalLM
    AutoConfig.register = staticmethod(lambda *a, **k: None)
    AutoModelForCausalLM.register = classmethod(lambda cls, *a, **k: None)
    import transformers.trainer as T
    for n in ("ALL_LAYERNORM_LAYERS", "ShardedDDPOption"):
        if not hasattr(T, n):
            setattr(T, n, object())
    peft, pm = types.ModuleType("peft"), types.ModuleType("peft.peft_model")

    class PeftModelForCausalLM:
        pass
    pm.PeftModelForCausalLM = peft.PeftModel = PeftModelForCausalLM
    peft.peft_model, peft.get_peft_model, peft.prepare_model_for_kbit_training = pm, None, None
    peft.__spec__ = importlib.machinery.ModuleSpec("peft", None)
    sys.modules["peft"], sys.modules["peft.peft_model"] = peft, pm
    import llava.train.halva_trainer as H
    import llava.train.train_halva as TH
    from llava import conversation as conv_lib
    conv_lib.default_conversation = conv_lib.conv_templates["v1"]
    return H, TH


H, TH = import_reference()
from llava.model import LlavaConfig, LlavaLlamaForCausalLM  # noqa: E402
from llava.model.language_model import modelling_llama as V431  # noqa: E402
from llava.model.multimodal_encoder.clip_encoder import CLIPVisionTower  # noqa: E402


def t2n(t):
    t = t.detach()
    if t.dtype == torch.bfloat16:
        t = t.float()
    return t.cpu().numpy()


def save_npz(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


def save_json(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, ensure_ascii=False, indent=0)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------------------
# G2 / G3: masked tokenisation + label masking  (train_halva.py:263-561)
# ----------------------------------------------------------------------------------------------
def plain_of(masked):
    s = masked.replace(" </MASK> ", " ").replace(" </MASK>", "").replace(" <MASK> ", " ")
    if s.startswith("<MASK> "):
        s = s[len("<MASK> "):]
    return s


TOK_CASES = [
    ("What is in the picture?",
     "There is a <MASK> dog </MASK> sitting on <MASK> the grass </MASK>. It is the <MASK> dog </MASK>'s toy, "
     "<MASK> red </MASK>, nearby."),
    ("Is there a cat?", "<MASK> Yes </MASK>, there is a cat."),
    ("Is there a cat?", "<MASK> No </MASK>, there is no cat in the image."),
    ("Describe the scene.", "A man rides a <MASK> brown horse </MASK> near <MASK> two </MASK> trees."),
    ("Describe the scene.", "A man rides a <MASK> white horse </MASK> near <MASK> three </MASK> trees."),
    ("What colour is the bus?", "The bus is <MASK> blue </MASK>"),
    ("What colour is the bus?", "The bus is <MASK> green and yellow </MASK>"),
    ("How many people?", "I can see <MASK> 4 </MASK> people and <MASK> 2 </MASK> dogs."),
    ("What is the woman holding?",
     "The woman's hand holds an <MASK> umbrella </MASK>, and the <MASK> child </MASK>'s hat is <MASK> small </MASK>."),
    ("Anything else?", "Nothing else is visible in this image."),
    ("List objects.", "<MASK> chair </MASK>, <MASK> table </MASK>, <MASK> lamp </MASK>."),
    ("Where is it?", "It is on the <MASK> left side </MASK> of the <MASK> wooden table </MASK>"),
    ("Two lines?", "First line has a <MASK> bird </MASK>.\nSecond line has <MASK> no bird </MASK>."),
    ("Is the door open?", "<MASK> Yes </MASK>"),
    ("Is the door open?", "<MASK> No </MASK>"),
    ("What is he doing?", "He is <MASK> surfing </MASK> on a <MASK> big wave </MASK>, wearing a <MASK> black </MASK> wetsuit."),
    # malformed for Llama-style tokenisation (no space inside the tags): the reference's own sanity check rejects it
    ("Bad form?", "There is a <MASK>dog</MASK> here."),
]

REF_CASES = [
    ("What is shown here?", "A plate of food with rice and vegetables."),
    ("Describe the image in detail.", "Two children play football on a green field. One wears a red shirt."),
    ("Is it raining?", "No"),
]


def gen_tokenize():
    tok = FakeLlamaTokenizer(model_max_length=2048)
    out = []
    for q, masked in TOK_CASES:
        plain = plain_of(masked)
        sources = [[{"from": "human", "value": "<image>\n" + q},
                    {"from": "gpt", "value": masked},
                    {"from": "gpt-ref", "value": plain}]]
        rec = {"question": q, "answer_masked": masked, "answer": plain}
        try:
            d = TH.preprocess_v1(copy.deepcopy(sources), tok, has_image=True)
            if d is None:
                rec["result"] = "none"
            else:
                rec["result"] = "ok"
                rec["input_ids"] = d["input_ids"][0].tolist()
                rec["labels"] = d["labels"][0].tolist()
                rec["signs"] = d["signs"][0].tolist()
        except Exception as e:  # the reference raises on length mismatch (train_halva.py:426)
            rec["result"] = "raise:" + type(e).__name__
        out.append(rec)
    refs = []
    for q, a in REF_CASES:
        sources = [[{"from": "human", "value": "<image>\n" + q}, {"from": "gpt", "value": a}]]
        d = TH.preprocess_v1_ref(copy.deepcopy(sources), tok, has_image=True)
        refs.append({"question": q, "answer": a, "input_ids": d["input_ids"][0].tolist(),
                     "labels": d["labels"][0].tolist()})
    # direct calls of the span walker on post-image prompt fragments
    walk = []
    for s in [" a <MASK> dog </MASK> and <MASK> a cat </MASK>. done",
              " plain text without tags",
              " <MASK> Yes </MASK>, ok",
              " the <MASK> man </MASK>'s hat"]:
        ids, signs = TH.split_string_by_mask_and_tokenize(s, tok)
        walk.append({"string": s, "ids": ids, "signs": signs})
    save_json("tokenize_masks.json", {"vocab": tok.vocab, "model_max_length": tok.model_max_length,
                                      "cases": out, "ref_cases": refs, "walk": walk})
    return tok, out, refs


# ----------------------------------------------------------------------------------------------
# G1: collator (train_halva.py:902-993)
# ----------------------------------------------------------------------------------------------
def gen_collator(tok_cases, ref_cases):
    ok = [c for c in tok_cases if c["result"] == "ok"]
    g = torch.Generator().manual_seed(7)
    packs = {}
    meta = []
    for ci, (idx, max_len) in enumerate([((0, 1, 3), 2048), ((3, 4, 5, 7), 40), ((9,), 2048)]):
        tok = FakeLlamaTokenizer(model_max_length=max_len)
        inst = []
        for k, i in enumerate(idx):
            pos, neg = ok[i], ok[(i + 1) % len(ok)]
            r = ref_cases[k % len(ref_cases)]
            inst.append(dict(
                input_ids=torch.tensor(pos["input_ids"]), labels=torch.tensor(pos["labels"]),
                neg_input_ids=torch.tensor(neg["input_ids"]), neg_labels=torch.tensor(neg["labels"]),
                pos_signs=torch.tensor(pos["signs"]), neg_signs=torch.tensor(neg["signs"]),
                ref_input_ids=torch.tensor(r["input_ids"]), ref_labels=torch.tensor(r["labels"]),
                image=torch.randn(3, 4, 4, generator=g), ref_image=torch.randn(3, 4, 4, generator=g)))
        batch = TH.DataCollatorForHallDataset(tokenizer=tok)(inst)
        meta.append({"max_len": max_len, "n": len(inst)})
        for k, inst_k in enumerate(inst):
            for key, v in inst_k.items():
                packs["c%d_in%d_%s" % (ci, k, key)] = t2n(v)
        for key, v in batch.items():
            packs["c%d_out_%s" % (ci, key)] = t2n(v)
    packs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save_npz("collator.npz", **packs)


# ----------------------------------------------------------------------------------------------
# G6: sampler (halva_trainer.py:60-152)
# ----------------------------------------------------------------------------------------------
def gen_sampler():
    rng = np.random.RandomState(3)
    cases = []
    for n, bs, ws, seed, mixed in [(37, 4, 2, 0, False), (64, 4, 4, 1, False), (50, 3, 2, 2, True), (16, 4, 16, 5, False)]:
        lengths = rng.randint(5, 200, size=n).tolist()
        if mixed:
            lengths = [l if i % 3 else -l for i, l in enumerate(lengths)]
        g = torch.Generator().manual_seed(seed)
        torch.manual_seed(1000 + seed)      # the mixed-modality branch draws from the GLOBAL torch RNG (generator=None)
        s = H.LengthGroupedSampler(bs, ws, lengths=lengths, generator=g, group_by_modality=True)
        idx = list(iter(s))
        g = torch.Generator().manual_seed(seed)
        idx_plain = H.get_length_grouped_indices([abs(l) for l in lengths], bs, ws, generator=g)
        cases.append({"lengths": lengths, "batch_size": bs, "world_size": ws, "seed": seed, "global_seed": 1000 + seed,
                      "modality_indices": idx, "length_indices": idx_plain})
    chunks = []
    for n, k in [(12, 3), (10, 3), (8, 4)]:
        lengths = rng.randint(1, 50, size=n).tolist()
        ind = sorted(range(n), key=lambda i: -lengths[i])
        chunks.append({"indices": ind, "lengths": lengths, "num_chunks": k,
                       "out": H.split_to_even_chunks(ind, lengths, k)})
    save_json("sampler.json", {"cases": cases, "chunks": chunks})


# ----------------------------------------------------------------------------------------------
# tiny models
# ----------------------------------------------------------------------------------------------
TINY = dict(vocab_size=160, hidden_size=64, intermediate_size=96, num_hidden_layers=2, num_attention_heads=4,
            num_key_value_heads=4, max_position_embeddings=128, rms_norm_eps=1e-5, pad_token_id=0)
VIS = dict(hidden_size=32, intermediate_size=48, num_hidden_layers=3, num_attention_heads=2, image_size=28,
           patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3)


def build_vision_tower(seed, vis=None, bf16_round=False):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(**(vis or VIS))
    orig = CLIPVisionConfig.from_pretrained
    CLIPVisionConfig.from_pretrained = classmethod(lambda cls, *a, **k: cfg)
    try:
        tower = CLIPVisionTower("tiny-clip", types.SimpleNamespace(mm_vision_select_layer=-2,
                                                                  mm_vision_select_feature="patch"), delay_load=True)
    finally:
        CLIPVisionConfig.from_pretrained = orig
    torch.manual_seed(seed)
    tower.vision_tower = CLIPVisionModel(cfg)
    with torch.no_grad():
        for p in tower.vision_tower.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
        for n, p in tower.vision_tower.named_parameters():
            if "layer_norm" in n or "layrnorm" in n:
                if n.endswith("weight"):
                    p.add_(1.0)
        if bf16_round:
            for p in tower.vision_tower.parameters():
                p.copy_(p.bfloat16().float())
    tower.vision_tower.requires_grad_(False)
    tower.is_loaded = True
    return tower


def build_llava(seed, tower, max_len=64, padding_side="right", tiny=None, vis=None, bf16_round=False, std=0.06):
    tiny, vis = tiny or TINY, vis or VIS
    torch.manual_seed(seed)
    cfg = LlavaConfig(**tiny)
    cfg._attn_implementation = "eager"
    m = LlavaLlamaForCausalLM(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
            else:
                p.copy_(torch.randn_like(p) * std)
    m.model.vision_tower = tower
    m.model.mm_projector = torch.nn.Sequential(torch.nn.Linear(vis["hidden_size"], tiny["hidden_size"]), torch.nn.GELU(),
                                               torch.nn.Linear(tiny["hidden_size"], tiny["hidden_size"]))
    with torch.no_grad():
        for p in m.model.mm_projector.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
        if bf16_round:
            for n, p in m.named_parameters():
                if "vision_tower" not in n:
                    p.copy_(p.bfloat16().float())
    m.config.tokenizer_model_max_length = max_len
    m.config.tokenizer_padding_side = padding_side
    m.eval()
    return m


def export_llava(m, prefix):
    out = {}
    for n, p in m.state_dict().items():
        if "rotary_emb" in n or "vision_tower" in n:
            continue
        out[prefix + n] = t2n(p)
    return out


def make_batch(B, seed, n_patch, long_resp=False, vis=None, vocab=None, resp_base=None):
    """Synthetic collated batch in the reference's key layout (train_halva.py:963-989)."""
    vis = vis or VIS
    g = torch.Generator().manual_seed(seed)
    V = vocab or TINY["vocab_size"]

    def seq(resp_len, phrases):
        pre = torch.randint(3, V, (4,), generator=g).tolist()
        q = torch.randint(3, V, (5,), generator=g).tolist()
        resp = torch.randint(3, V, (resp_len,), generator=g).tolist()
        ids = [1] + pre + [-200] + q + resp + [2]
        labels = [-100] * (1 + len(pre) + 1 + len(q)) + resp + [2]
        signs = [0] * len(ids)
        off = 1 + len(pre) + 1 + len(q)
        for k, (s, l) in enumerate(phrases):
            for t in range(s, s + l):
                signs[off + t] = k + 1
        return ids, labels, signs

    inst = []
    for b in range(B):
        rl = ((14 if long_resp else 8) if resp_base is None else resp_base) + 3 * b
        nph = 2 if b % 2 == 0 else 1            # unequal phrase counts -> batch-global slots with log2 filler
        phrases = [(1 + 4 * k, 2) for k in range(nph)]
        ids, labels, signs = seq(rl, phrases)
        nids, nlabels = list(ids), list(labels)
        off = len(ids) - 1 - rl
        for s, l in phrases:                    # neg differs from pos only inside phrase spans
            for t in range(s, s + l):
                v = int(torch.randint(3, V, (1,), generator=g))
                nids[off + t] = v
                nlabels[off + t] = v
        if b == 1:                              # unequal pos/neg lengths
            nids = nids[:-1] + [7, 2]
            nlabels = nlabels[:-1] + [7, 2]
            nsigns = signs[:-1] + [0, 0]
        else:
            nsigns = list(signs)
        rids, rlabels, _ = seq(6 + 2 * b, [])
        inst.append(dict(input_ids=torch.tensor(ids), labels=torch.tensor(labels), neg_input_ids=torch.tensor(nids),
                         neg_labels=torch.tensor(nlabels), pos_signs=torch.tensor(signs), neg_signs=torch.tensor(nsigns),
                         ref_input_ids=torch.tensor(rids), ref_labels=torch.tensor(rlabels),
                         image=torch.randn(3, vis["image_size"], vis["image_size"], generator=g).bfloat16().float(),
                         ref_image=torch.randn(3, vis["image_size"], vis["image_size"], generator=g).bfloat16().float()))
    tok = FakeLlamaTokenizer(model_max_length=2048)
    return TH.DataCollatorForHallDataset(tokenizer=tok)(inst)


def trainer_stub(policy, ref, alpha):
    stub = types.SimpleNamespace(model=policy, ref_model=ref, loss_alpha=alpha, label_pad_token_id=-100,
                                 is_encoder_decoder=False)
    for n in ("cal_batch_logp", "accumulate_logps", "concatenated_forward", "reference_forward", "compute_loss"):
        setattr(stub, n, types.MethodType(getattr(H.HalvaTrainer, n), stub))
    return stub


# ----------------------------------------------------------------------------------------------
# G7: CLIP tower + projector (clip_encoder.py:27-49, multimodal_projector/builder.py:39-46)
# ----------------------------------------------------------------------------------------------
def gen_clip(tower, model):
    g = torch.Generator().manual_seed(11)
    images = torch.randn(3, 3, VIS["image_size"], VIS["image_size"], generator=g)
    with torch.no_grad():
        feats = tower(images)
        proj = model.encode_images(images)
    packs = {"images": t2n(images), "features": t2n(feats), "projected": t2n(proj)}
    for n, p in tower.vision_tower.state_dict().items():
        if "position_ids" in n:
            continue
        packs["clip." + n] = t2n(p)
    for n, p in model.model.mm_projector.state_dict().items():
        packs["proj." + n] = t2n(p)
    packs["cfg"] = np.frombuffer(json.dumps(VIS).encode(), dtype=np.uint8)
    save_npz("clip_tower.npz", **packs)


# ----------------------------------------------------------------------------------------------
# G3: splice (llava_arch.py:85-394)
# ----------------------------------------------------------------------------------------------
def gen_splice(tower):
    packs = {}
    meta = []
    n_patch = (VIS["image_size"] // VIS["patch_size"]) ** 2
    for ci, (B, seed, max_len, side, long_resp) in enumerate([(3, 21, 64, "right", False), (3, 22, 20, "right", True),
                                                               (2, 23, 64, "left", False)]):
        m = build_llava(100 + ci, tower, max_len=max_len, padding_side=side)
        batch = make_batch(B, seed, n_patch, long_resp)
        stub = trainer_stub(m, m, 0.4)
        # the pos||neg assembly is part of concatenated_forward; restate its inputs via the same code path
        ids, neg = batch["input_ids"], batch["neg_input_ids"]
        T0 = max(ids.shape[1], neg.shape[1])
        cat_ids = torch.zeros(2 * B, T0, dtype=torch.long)
        cat_lab = torch.full((2 * B, T0), -100, dtype=torch.long)
        cat_att = torch.zeros(2 * B, T0, dtype=torch.bool)
        cat_sig = torch.zeros(2 * B, T0, dtype=torch.long)
        cat_ids[:B, :ids.shape[1]] = ids
        cat_ids[B:, :neg.shape[1]] = neg
        cat_lab[:B, :ids.shape[1]] = batch["labels"]
        cat_lab[B:, :neg.shape[1]] = batch["neg_labels"]
        cat_att[:B, :ids.shape[1]] = batch["attention_mask"]
        cat_att[B:, :neg.shape[1]] = batch["neg_attention_mask"]
        cat_sig[:B, :ids.shape[1]] = batch["pos_signs"]
        cat_sig[B:, :neg.shape[1]] = batch["neg_signs"]
        images = torch.cat([batch["images"], batch["images"]], 0)
        with torch.no_grad():
            feats = m.encode_images(images)
            r = m.prepare_inputs_labels_for_multimodal_signed(cat_ids, None, cat_att, None, cat_lab, images, cat_sig)
            r2 = m.prepare_inputs_labels_for_multimodal(batch["ref_input_ids"], None, batch["ref_attention_mask"], None,
                                                        batch["ref_labels"], batch["ref_images"])
            ref_feats = m.encode_images(batch["ref_images"])
        assert r[0] is None and r[1] is None and r[3] is None
        p = "s%d_" % ci
        packs.update({p + "ids": t2n(cat_ids), p + "labels": t2n(cat_lab), p + "mask": t2n(cat_att), p + "signs": t2n(cat_sig),
                      p + "features": t2n(feats), p + "embed_tokens": t2n(m.model.embed_tokens.weight),
                      p + "out_mask": t2n(r[2]), p + "out_embeds": t2n(r[4]), p + "out_labels": t2n(r[5]),
                      p + "out_signs": t2n(r[6]),
                      p + "ref_ids": t2n(batch["ref_input_ids"]), p + "ref_labels": t2n(batch["ref_labels"]),
                      p + "ref_mask": t2n(batch["ref_attention_mask"]), p + "ref_features": t2n(ref_feats),
                      p + "ref_out_mask": t2n(r2[2]), p + "ref_out_embeds": t2n(r2[4]), p + "ref_out_labels": t2n(r2[5])})
        meta.append({"B": B, "max_len": max_len, "padding_side": side})
    packs["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save_npz("splice.npz", **packs)


# ----------------------------------------------------------------------------------------------
# G4: cal_batch_logp / accumulate_logps (halva_trainer.py:392-419)
# ----------------------------------------------------------------------------------------------
def gen_loss_small():
    g = torch.Generator().manual_seed(5)
    stub = trainer_stub(None, None, 0.4)
    S, T, Vv = 4, 12, 50
    logits = torch.randn(S, T, Vv, generator=g) * 3
    labels = torch.randint(0, Vv, (S, T), generator=g)
    labels[:, :4] = -100
    labels[2, 9:] = -100
    logps = stub.cal_batch_logp(logits, labels)
    signs = torch.zeros(S, T - 1, dtype=torch.long)
    signs[0, 4:6] = 1
    signs[0, 8:9] = 2
    signs[1, 5:7] = 1
    signs[2, 4:5] = 1
    signs[2, 6:8] = 3        # non-contiguous ids: slots follow torch.unique over the whole half-batch
    acc = stub.accumulate_logps(logps, signs)
    save_npz("loss_small.npz", logits=t2n(logits), labels=t2n(labels), logps=t2n(logps), signs=t2n(signs), acc=t2n(acc))


# ----------------------------------------------------------------------------------------------
# G5: full compute_loss on tiny models, fp32 (halva_trainer.py:534-592)
# ----------------------------------------------------------------------------------------------
def lora_merge(model, seed, r=4, alpha=8.0, bf16_round=False, std=0.05):
    """Emulate peft-0.4 LoRA (W x + (alpha/r) B A x) by merging into the dense reference weights;
    returns the A/B factors so the build's explicit-LoRA model can be checked against the merge."""
    g = torch.Generator().manual_seed(seed)
    fac = {}
    with torch.no_grad():
        for n, mod in model.named_modules():
            if isinstance(mod, torch.nn.Linear) and "mm_projector" not in n and "vision_tower" not in n and "lm_head" not in n:
                A = torch.randn(r, mod.in_features, generator=g) * std
                Bm = torch.randn(mod.out_features, r, generator=g) * std
                if bf16_round:
                    A, Bm = A.bfloat16().float(), Bm.bfloat16().float()
                mod.weight.add_((alpha / r) * (Bm @ A))
                fac["lora." + n + ".A"] = t2n(A)
                fac["lora." + n + ".B"] = t2n(Bm)
    return fac


def gen_dpa_step(tower):
    n_patch = (VIS["image_size"] // VIS["patch_size"]) ** 2
    for name, B, seed, max_len, same in [("dpa_step_a", 3, 31, 64, False), ("dpa_step_trunc", 2, 32, 24, False),
                                         ("dpa_step_identity", 2, 33, 64, True)]:
        ref = build_llava(200 + seed, tower, max_len=max_len)
        policy = copy.deepcopy(ref)
        policy.model.vision_tower = tower
        ref.requires_grad_(False)
        packs = export_llava(ref, "base.")
        if not same:
            packs.update(lora_merge(policy, seed + 1))
        packs["lora_cfg"] = np.array([4, 8.0])
        batch = make_batch(B, seed, n_patch, long_resp=(max_len < 40))
        alpha = 0.4
        stub = trainer_stub(policy, ref, alpha)
        policy.zero_grad()
        pos_logps, neg_logps, batch_labels, all_logits, batch_signs = stub.concatenated_forward(policy, batch)
        loss = stub.compute_loss(policy, batch)
        loss.backward()
        # components, recomputed the reference's way for the fixture
        with torch.no_grad():
            mask = (batch_labels != -100)
            half = pos_logps.shape[0]
            sg = batch_signs.masked_fill(batch_signs == -100, 0)
            pa = stub.accumulate_logps(pos_logps * mask[:half].float(), sg[:half])
            na = stub.accumulate_logps(neg_logps * mask[half:].float(), sg[half:])
            align = torch.log(1 + torch.exp(na - pa)).mean()
            div = (loss - align) / alpha
        for k, v in batch.items():
            packs["batch." + k] = t2n(v)
        packs.update({"out.loss": t2n(loss), "out.alignment": t2n(align), "out.divergence": t2n(div),
                      "out.pos_logps": t2n(pos_logps), "out.neg_logps": t2n(neg_logps), "out.batch_labels": t2n(batch_labels),
                      "out.batch_signs": t2n(batch_signs), "out.pos_acc": t2n(pa), "out.neg_acc": t2n(na),
                      "out.all_logits": t2n(all_logits),
                      "alpha": np.array(alpha), "max_len": np.array(max_len)})
        # gradients w.r.t. merged dense weights (dL/dW); LoRA grads follow by the chain rule dA = s B^T dW, dB = s dW A^T
        for n, p in policy.named_parameters():
            if p.grad is not None and ("layers.0.self_attn.q_proj" in n or "layers.1.mlp.down_proj" in n
                                       or "mm_projector" in n or "layers.1.self_attn.v_proj" in n
                                       or "layers.0.mlp.gate_proj" in n):
                packs["grad." + n] = t2n(p.grad)
        for n, p in tower.vision_tower.state_dict().items():
            if "position_ids" not in n:
                packs["clip." + n] = t2n(p)
        packs["clip_cfg"] = np.frombuffer(json.dumps(VIS).encode(), dtype=np.uint8)
        packs["llama_cfg"] = np.frombuffer(json.dumps(TINY).encode(), dtype=np.uint8)
        save_npz(name + ".npz", **packs)
        print("   ", name, "loss", float(loss), "align", float(align), "div", float(div))


# ----------------------------------------------------------------------------------------------
# G8: one decoder layer of the vendored transformers-4.31 spec, fwd/bwd (modelling_llama.py:56-420)
# ----------------------------------------------------------------------------------------------
TINY64 = dict(vocab_size=160, hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2,
              num_key_value_heads=2, max_position_embeddings=128, rms_norm_eps=1e-5, pad_token_id=0)
VIS64 = dict(hidden_size=128, intermediate_size=192, num_hidden_layers=3, num_attention_heads=2, image_size=28,
             patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5, num_channels=3)


# head_dim 128 = the headline attention instantiation (LLaVA-1.5: 32 x 128): 2 heads x 128, responses of ~120 tokens so
# that the causal kernel crosses 64-key tile boundaries (T ~ 140 after the splice)
TINY128 = dict(vocab_size=160, hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=2, max_position_embeddings=256, rms_norm_eps=1e-5, pad_token_id=0)


# Multi-block length (round 4): post-splice rows of 300-1024 tokens, so that a row spans >= 4 row blocks of 256 and >= 8 key blocks
# of 128, packed rows [prefix | correct | pad | hallucinated] cross 256-row blocks INSIDE branch B, the correct and the hallucinated
# answer differ in length (phrases of different lengths), and one sample is cut at tokenizer_model_max_length.
TINY128L = dict(TINY128, max_position_embeddings=2048)


def make_batch_long(B, seed, n_patch, vis=None, vocab=None, resp_base=None):
    """Collated batch (train_halva.py:963-989) of LONG responses: sample b's correct answer has RESP[b] tokens; its phrases sit at
    PHR[b] = [(offset in the correct answer, correct length, hallucinated length)], so the hallucinated answer is the correct one
    with every phrase replaced by a phrase of another length - everything behind the first phrase is shifted."""
    vis = vis or VIS
    g = torch.Generator().manual_seed(seed)
    V = vocab or TINY["vocab_size"]
    RESP = [300, 517, 806, 1100][:B]
    PHR = [[(40, 3, 5), (200, 2, 2)], [(9, 2, 6), (260, 4, 1), (400, 3, 3)], [(150, 5, 2), (700, 2, 4)],
           [(64, 3, 3), (500, 2, 7), (900, 4, 2)]][:B]
    inst = []
    for b in range(B):
        pre = torch.randint(3, V, (4,), generator=g).tolist()
        q = torch.randint(3, V, (5,), generator=g).tolist()
        resp = torch.randint(3, V, (RESP[b],), generator=g).tolist()
        head = [1] + pre + [-200] + q
        ids, signs = list(head), [0] * len(head)
        nids, nsigns = list(head), [0] * len(head)
        at = 0
        for k, (off, lp, ln) in enumerate(PHR[b]):
            ids += resp[at:off + lp]
            signs += [0] * (off - at) + [k + 1] * lp
            nids += resp[at:off] + torch.randint(3, V, (ln,), generator=g).tolist()
            nsigns += [0] * (off - at) + [k + 1] * ln
            at = off + lp
        ids += resp[at:] + [2]
        signs += [0] * (len(resp) - at + 1)
        nids += resp[at:] + [2]
        nsigns += [0] * (len(resp) - at + 1)
        labels = [-100] * len(head) + ids[len(head):]
        nlabels = [-100] * len(head) + nids[len(head):]
        rresp = torch.randint(3, V, (190 + 240 * b,), generator=g).tolist()
        rids = [1] + torch.randint(3, V, (4,), generator=g).tolist() + [-200] + torch.randint(3, V, (5,), generator=g).tolist() + rresp + [2]
        rlabels = [-100] * 11 + rresp + [2]
        inst.append(dict(input_ids=torch.tensor(ids), labels=torch.tensor(labels), neg_input_ids=torch.tensor(nids),
                         neg_labels=torch.tensor(nlabels), pos_signs=torch.tensor(signs), neg_signs=torch.tensor(nsigns),
                         ref_input_ids=torch.tensor(rids), ref_labels=torch.tensor(rlabels),
                         image=torch.randn(3, vis["image_size"], vis["image_size"], generator=g).bfloat16().float(),
                         ref_image=torch.randn(3, vis["image_size"], vis["image_size"], generator=g).bfloat16().float()))
    tok = FakeLlamaTokenizer(model_max_length=2048)
    return TH.DataCollatorForHallDataset(tokenizer=tok)(inst)


def gen_dpa_step_d64(name="dpa_step_d64", std=0.06, lora_std=0.05, TINY64=TINY64, B=3, seed=51, max_len=64, resp_base=None,
                     model_seed=300, make_batch=None):
    """Geometry the HIP kernels support (head_dim 64; TINY128: head_dim 128) with bf16-representable weights/images, so the GPU
    path (bf16) and the reference (fp32 here) start from identical numbers.  Weights are stored as raw bf16 bits (uint16)."""
    def bits(t):
        return t.detach().bfloat16().view(torch.int16).numpy().view(np.uint16)
    tower = build_vision_tower(41, VIS64, bf16_round=True)
    n_patch = (VIS64["image_size"] // VIS64["patch_size"]) ** 2
    alpha = 0.4
    ref = build_llava(model_seed, tower, max_len=max_len, tiny=TINY64, vis=VIS64, bf16_round=True, std=std)
    policy = copy.deepcopy(ref)
    policy.model.vision_tower = tower
    ref.requires_grad_(False)
    packs = {}
    for n, p in ref.state_dict().items():
        if "rotary_emb" in n or "vision_tower" in n:
            continue
        packs["base." + n] = bits(p)
    fac = lora_merge(policy, seed + 1, bf16_round=True, std=lora_std)
    for k, v in fac.items():
        packs[k] = bits(torch.from_numpy(v))
    packs["lora_cfg"] = np.array([4, 8.0])
    batch = (make_batch or globals()["make_batch"])(B, seed, n_patch, vis=VIS64, vocab=TINY64["vocab_size"], resp_base=resp_base)
    stub = trainer_stub(policy, ref, alpha)
    policy.zero_grad()
    pos_logps, neg_logps, batch_labels, all_logits, batch_signs = stub.concatenated_forward(policy, batch)
    loss = stub.compute_loss(policy, batch)
    loss.backward()
    with torch.no_grad():
        mask = (batch_labels != -100)
        half = pos_logps.shape[0]
        sg = batch_signs.masked_fill(batch_signs == -100, 0)
        pa = stub.accumulate_logps(pos_logps * mask[:half].float(), sg[:half])
        na = stub.accumulate_logps(neg_logps * mask[half:].float(), sg[half:])
        align = torch.log(1 + torch.exp(na - pa)).mean()
        div = (loss - align) / alpha
    for k, v in batch.items():
        packs["batch." + k] = t2n(v)
    packs.update({"out.loss": t2n(loss), "out.alignment": t2n(align), "out.divergence": t2n(div),
                  "out.pos_logps": t2n(pos_logps), "out.neg_logps": t2n(neg_logps), "out.batch_labels": t2n(batch_labels),
                  "out.batch_signs": t2n(batch_signs), "out.pos_acc": t2n(pa), "out.neg_acc": t2n(na),
                  "alpha": np.array(alpha), "max_len": np.array(max_len)})
    for n, p in policy.named_parameters():
        if p.grad is not None and ("layers.0.self_attn.q_proj" in n or "layers.1.mlp.down_proj" in n or "mm_projector" in n
                                   or "layers.1.self_attn.v_proj" in n or "layers.0.mlp.gate_proj" in n
                                   or "layers.1.self_attn.o_proj" in n or "layers.0.self_attn.k_proj" in n):
            packs["grad." + n] = t2n(p.grad)
    for n, p in tower.vision_tower.state_dict().items():
        if "position_ids" not in n:
            packs["clip." + n] = bits(p)
    packs["clip_cfg"] = np.frombuffer(json.dumps(VIS64).encode(), dtype=np.uint8)
    packs["llama_cfg"] = np.frombuffer(json.dumps(TINY64).encode(), dtype=np.uint8)
    save_npz(name + ".npz", **packs)
    print("   ", name, "loss", float(loss), "align", float(align), "div", float(div))


def gen_clip_d64():
    """CLIP tower + projector at a geometry the HIP kernels run (2 heads x 64, 56 px -> 16 patches + CLS), bf16-exact weights and
    images: CLIPVisionTower.forward (clip_encoder.py:37-56: hidden_states[-2], CLS dropped) and encode_images (llava_arch.py:80-83)."""
    def bits(t):
        return t.detach().bfloat16().view(torch.int16).numpy().view(np.uint16)
    vis = dict(VIS64, image_size=56)
    tower = build_vision_tower(43, vis, bf16_round=True)
    m = build_llava(320, tower, tiny=TINY64, vis=vis, bf16_round=True, std=0.02)
    g = torch.Generator().manual_seed(12)
    images = torch.randn(3, 3, 56, 56, generator=g).bfloat16().float()
    with torch.no_grad():
        feats = tower(images)
        proj = m.encode_images(images)
        hs = tower.vision_tower(images, output_hidden_states=True).hidden_states
    packs = {"images": t2n(images), "features": t2n(feats), "projected": t2n(proj), "hidden_first": t2n(hs[0]),
             "hidden_m2": t2n(hs[-2])}
    for n, p in tower.vision_tower.state_dict().items():
        if "position_ids" not in n:
            packs["clip." + n] = bits(p)
    for n, p in m.model.mm_projector.state_dict().items():
        packs["proj." + n] = bits(p)
    packs["cfg"] = np.frombuffer(json.dumps(vis).encode(), dtype=np.uint8)
    save_npz("clip_tower_d64.npz", **packs)


def gen_llama_layer():
    from transformers import LlamaConfig
    torch.manual_seed(77)
    cfg = LlamaConfig(**TINY)
    cfg.rope_scaling = None
    cfg.pretraining_tp = 1
    layer = V431.LlamaDecoderLayer(cfg)
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(1.0 + 0.1 * torch.randn_like(p) if "norm" in n else torch.randn_like(p) * 0.08)
    Bs, T = 2, 20
    x = torch.randn(Bs, T, TINY["hidden_size"], requires_grad=True)
    keep = torch.ones(Bs, T, dtype=torch.bool)
    keep[1, 13:] = False
    dtype = torch.float32
    causal = V431._make_causal_mask((Bs, T), dtype, x.device)
    mask = V431._expand_mask(keep, dtype, tgt_len=T) + causal
    pos = torch.arange(T)[None]
    y = layer(x, attention_mask=mask, position_ids=pos)[0]
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    gy = gy * keep[..., None]
    y.backward(gy)
    packs = {"x": t2n(x), "keep": t2n(keep), "y": t2n(y), "gy": t2n(gy), "gx": t2n(x.grad)}
    for n, p in layer.named_parameters():
        packs["w." + n] = t2n(p)
        packs["g." + n] = t2n(p.grad)
    # attention internals of the same layer (eager fp32): q/k after RoPE, softmax output
    with torch.no_grad():
        att = layer.self_attn
        h = layer.input_layernorm(x)
        q = att.q_proj(h).view(Bs, T, 4, 16).transpose(1, 2)
        k = att.k_proj(h).view(Bs, T, 4, 16).transpose(1, 2)
        v = att.v_proj(h).view(Bs, T, 4, 16).transpose(1, 2)
        cos, sin = att.rotary_emb(v, seq_len=T)
        qr, kr = V431.apply_rotary_pos_emb(q, k, cos, sin, pos)
        packs.update({"rms1": t2n(h), "q": t2n(q), "k": t2n(k), "v": t2n(v), "q_rope": t2n(qr), "k_rope": t2n(kr),
                      "cos": t2n(cos[0, 0]), "sin": t2n(sin[0, 0])})
    packs["llama_cfg"] = np.frombuffer(json.dumps(TINY).encode(), dtype=np.uint8)
    save_npz("llama_layer.npz", **packs)


def gen_peft_names():
    """The key sets of the run's output files as the REFERENCE's own helpers select them (train_halva.py:116-153 called exactly as
    train() does at :1230-1240), on the reference's tiny LLaVA wrapped the way peft 0.4.0 wraps it.  peft is not installed here, so
    the WRAPPING is emulated from its documented module layout (PeftModel.base_model = LoraModel, LoraModel.model = the wrapped
    network; every target nn.Linear keeps `.weight` and gains `.lora_A.default` / `.lora_B.default` Linear sub-modules) - the
    selection (which names are LoRA state, which are "non-LoRA trainables", which modules are targets) is the reference's code.
    `adapter_model_bin` = the names peft's save_pretrained writes for that state dict (adapter name stripped: `.default` removed)."""
    ds = types.ModuleType("deepspeed")
    ds.zero = types.SimpleNamespace(GatheredParameters=None)
    pp = types.ModuleType("deepspeed.runtime.zero.partition_parameters")
    pp.ZeroParamStatus = types.SimpleNamespace(NOT_AVAILABLE=0)
    for name, mod in (("deepspeed", ds), ("deepspeed.runtime", types.ModuleType("deepspeed.runtime")),
                      ("deepspeed.runtime.zero", types.ModuleType("deepspeed.runtime.zero")),
                      ("deepspeed.runtime.zero.partition_parameters", pp)):
        sys.modules.setdefault(name, mod)
    tower = build_vision_tower(9)
    m = build_llava(10, tower)
    targets = sorted(TH.find_all_linear_names(m))
    r = 4

    class LoraLinear(torch.nn.Linear):
        def __init__(self, base):
            super().__init__(base.in_features, base.out_features, bias=base.bias is not None)
            self.weight = base.weight
            self.lora_A = torch.nn.ModuleDict({"default": torch.nn.Linear(base.in_features, r, bias=False)})
            self.lora_B = torch.nn.ModuleDict({"default": torch.nn.Linear(r, base.out_features, bias=False)})

    for p in m.parameters():
        p.requires_grad = False
    for name, mod in list(m.named_modules()):
        if isinstance(mod, torch.nn.Linear) and name.split(".")[-1] in targets and not any(
                k in name for k in ("mm_projector", "vision_tower", "vision_resampler")):
            parent = m.get_submodule(".".join(name.split(".")[:-1]))
            setattr(parent, name.split(".")[-1], LoraLinear(mod))
    for p in m.get_model().mm_projector.parameters():      # llava_arch.py:58-61 (initialize_vision_modules re-enables them)
        p.requires_grad = True

    class LoraModel(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model

    class Peft(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.base_model = LoraModel(model)
    pm = Peft(m)
    lora = TH.get_peft_state_maybe_zero_3(pm.named_parameters(), "none")
    non_lora = TH.get_peft_state_non_lora_maybe_zero_3(pm.named_parameters())
    save_json("peft_state_names.json", {
        "llama_cfg": TINY, "lora_r": r, "target_modules": targets,
        "lora_state": {k: list(v.shape) for k, v in lora.items()},
        "adapter_model_bin": {k.replace(".default", ""): list(v.shape) for k, v in lora.items()},
        "non_lora_trainables": {k: list(v.shape) for k, v in non_lora.items()}})


def main():
    torch.set_num_threads(4)
    torch.manual_seed(0)
    _, cases, refs = gen_tokenize()
    gen_collator(cases, refs)
    gen_sampler()
    tower = build_vision_tower(9)
    m = build_llava(10, tower)
    gen_clip(tower, m)
    gen_splice(tower)
    gen_loss_small()
    gen_dpa_step(tower)
    gen_dpa_step_d64()
    gen_dpa_step_d64("dpa_step_d64_init", std=0.02, lora_std=0.02)
    gen_dpa_step_d64("dpa_step_d128_init", std=0.02, lora_std=0.02, TINY64=TINY128, B=4, seed=61, max_len=192, resp_base=118,
                     model_seed=310)
    gen_dpa_step_long()
    gen_llama_layer()
    gen_clip_d64()
    gen_peft_names()


def gen_dpa_step_long():
    gen_dpa_step_d64("dpa_step_d128_long", std=0.02, lora_std=0.02, TINY64=TINY128L, B=4, seed=71, max_len=1024, model_seed=320,
                     make_batch=make_batch_long)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "long":           # only the fixture added in round 4
        torch.set_num_threads(4)
        gen_dpa_step_long()
    elif len(sys.argv) > 1 and sys.argv[1] == "peft":          # only the fixture added in round 3
        gen_peft_names()
    elif len(sys.argv) > 1 and sys.argv[1] == "d128":          # only the fixtures added in round 2 (the others reproduce bit for bit)
        torch.set_num_threads(4)
        gen_clip_d64()
        gen_dpa_step_d64("dpa_step_d128_init", std=0.02, lora_std=0.02, TINY64=TINY128, B=4, seed=61, max_len=192, resp_base=118,
                         model_seed=310)
    else:
        main()
