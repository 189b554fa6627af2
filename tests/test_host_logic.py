"""CPU tests of the PRODUCT's host-side integer logic (llava.train.*, halva_amd.splice / dpa planning) against the
golden vectors produced by the reference.  Bit-exact."""
import copy

import numpy as np
import pytest
import torch

from fake_tokenizer import FakeLlamaTokenizer
from golden_util import load_json, load_npz, meta_of


@pytest.fixture(scope="module")
def TH():
    import llava.train.train_halva as th
    return th


def test_masked_tokenisation_matches_reference(TH):
    fx = load_json("tokenize_masks.json")
    tok = FakeLlamaTokenizer(fx["model_max_length"], vocab=fx["vocab"], frozen=True)
    for c in fx["cases"]:
        src = [[{"from": "human", "value": "<image>\n" + c["question"]}, {"from": "gpt", "value": c["answer_masked"]},
                {"from": "gpt-ref", "value": c["answer"]}]]
        if c["result"].startswith("raise"):
            with pytest.raises(RuntimeError):
                TH.preprocess_v1(copy.deepcopy(src), tok, has_image=True)
            continue
        out = TH.preprocess_v1(copy.deepcopy(src), tok, has_image=True)
        if c["result"] == "none":
            assert out is None
            continue
        assert out["input_ids"][0].tolist() == c["input_ids"]
        assert out["signs"][0].tolist() == c["signs"]
        assert out["labels"][0].tolist() == c["labels"]
    for c in fx["ref_cases"]:
        src = [[{"from": "human", "value": "<image>\n" + c["question"]}, {"from": "gpt", "value": c["answer"]}]]
        out = TH.preprocess_v1_ref(src, tok, has_image=True)
        assert out["input_ids"][0].tolist() == c["input_ids"] and out["labels"][0].tolist() == c["labels"]
    for w in fx["walk"]:
        ids, signs = TH.split_string_by_mask_and_tokenize(w["string"], tok)
        assert ids == w["ids"] and signs == w["signs"]


def test_collator_matches_reference(TH):
    z = load_npz("collator.npz")
    for ci, m in enumerate(meta_of(z)):
        inst = []
        for k in range(m["n"]):
            p = "c%d_in%d_" % (ci, k)
            inst.append({key[len(p):]: torch.from_numpy(z[key]) for key in z.files if key.startswith(p)})
        batch = TH.DataCollatorForHallDataset(tokenizer=FakeLlamaTokenizer(model_max_length=m["max_len"]))(inst)
        p = "c%d_out_" % ci
        keys = [key[len(p):] for key in z.files if key.startswith(p)]
        assert sorted(keys) == sorted(batch.keys())
        for key in keys:
            assert batch[key].dtype == torch.from_numpy(z[p + key]).dtype, key
            np.testing.assert_array_equal(batch[key].numpy(), z[p + key], err_msg=key)


def test_sampler_matches_reference():
    from llava.train import halva_trainer as H
    fx = load_json("sampler.json")
    for c in fx["cases"]:
        g = torch.Generator().manual_seed(c["seed"])
        torch.manual_seed(c["global_seed"])
        s = H.LengthGroupedSampler(c["batch_size"], c["world_size"], lengths=c["lengths"], generator=g, group_by_modality=True)
        assert list(iter(s)) == c["modality_indices"]
        g = torch.Generator().manual_seed(c["seed"])
        assert H.get_length_grouped_indices([abs(l) for l in c["lengths"]], c["batch_size"], c["world_size"], generator=g) == c["length_indices"]
    for c in fx["chunks"]:
        assert H.split_to_even_chunks(c["indices"], c["lengths"], c["num_chunks"]) == c["out"]


def test_splice_plan_matches_reference():
    from halva_amd import splice as sp
    z = load_npz("splice.npz")
    for ci, m in enumerate(meta_of(z)):
        p = "s%d_" % ci
        n_patch = z[p + "features"].shape[1]
        plan = sp.plan_splice(z[p + "ids"], z[p + "mask"], z[p + "labels"], z[p + "signs"], n_patch, m["max_len"], m["padding_side"])
        np.testing.assert_array_equal(plan.labels.numpy(), z[p + "out_labels"])
        np.testing.assert_array_equal(plan.signs.numpy(), z[p + "out_signs"])
        np.testing.assert_array_equal(plan.mask.numpy(), z[p + "out_mask"])
        emb, feats = z[p + "embed_tokens"], z[p + "features"].reshape(-1, z[p + "embed_tokens"].shape[1])
        src = plan.src.numpy()
        out = np.zeros((len(src), emb.shape[1]), np.float32)          # CPU emulation of the gather kernel's contract
        out[src >= 0] = emb[src[src >= 0]]
        out[src <= -2] = feats[-src[src <= -2] - 2]
        np.testing.assert_array_equal(out.reshape(plan.S, plan.T, -1), z[p + "out_embeds"])
        start, length = sp.spans_from_mask(plan.mask)
        assert torch.equal(start, plan.seq_start) and torch.equal(length, plan.seq_len)
        rp = sp.plan_splice(z[p + "ref_ids"], z[p + "ref_mask"], z[p + "ref_labels"], None, n_patch, m["max_len"], m["padding_side"])
        np.testing.assert_array_equal(rp.labels.numpy(), z[p + "ref_out_labels"])
        np.testing.assert_array_equal(rp.mask.numpy(), z[p + "ref_out_mask"])
    with pytest.raises(ValueError):
        sp.spans_from_mask(np.array([[True, False, True]]))


def test_step_plan_slots_and_shared_image_map():
    """Batch-global phrase slots come out of the host plan; pos/neg rows share one image slot."""
    from halva_amd import dpa
    z = load_npz("dpa_step_a.npz")
    batch = {k[len("batch."):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("batch.")}
    plan = dpa.DPAStepPlan(batch, n_patch=4, max_len=int(z["max_len"]))
    sg = z["out.batch_signs"]
    B = plan.B
    assert plan.pos_slots.tolist() == np.unique(np.where(sg[:B] == -100, 0, sg[:B]))[1:].tolist()
    assert plan.P == z["out.pos_acc"].shape[1]
    g = plan.pair_group([1, 2])
    full_lab = z["out.batch_labels"]                      # shifted labels of the whole batch from the reference
    for row, b in enumerate([1, 2, B + 1, B + 2]):
        n = int(g.seq_len[row])
        np.testing.assert_array_equal(g.labels.numpy()[row, 1:n], full_lab[b, :n - 1])
    feat = -g.src.view(g.S, g.T)[:, :].numpy() - 2
    img_rows = [set((feat[r][feat[r] >= 0] // 4).tolist()) for r in range(4)]
    assert img_rows == [{0}, {1}, {0}, {1}]


def test_argument_parser_accepts_reference_flags(TH):
    argv = ("--lora_enable True --lora_r 128 --lora_alpha 256 --mm_projector_lr 0 --deepspeed src/json/zero3.json --loss_alpha 0.4 "
            "--model_name_or_path liuhaotian/llava-v1.5-7b --version v1 --data_path data/data.json --ref_data_path data/ref_data.json "
            "--image_folder default --vision_tower openai/clip-vit-large-patch14-336 --mm_projector_type mlp2x_gelu "
            "--mm_vision_select_layer -2 --mm_use_im_start_end False --mm_use_im_patch_token False --image_aspect_ratio pad "
            "--group_by_modality_length True --bf16 True --output_dir /tmp/o --num_train_epochs 1 --per_device_train_batch_size 4 "
            "--per_device_eval_batch_size 4 --gradient_accumulation_steps 4 --evaluation_strategy no --save_strategy steps "
            "--save_steps 50000 --learning_rate 5e-6 --weight_decay 0. --warmup_ratio 0.03 --lr_scheduler_type cosine "
            "--logging_steps 1 --tf32 True --model_max_length 2048 --gradient_checkpointing True --dataloader_num_workers 8 "
            "--lazy_preprocess True --report_to wandb --save_total_limit 1 --run_name halva-7b-lora --local_rank=0").split()
    m, d, t = TH.parse_args_into_dataclasses((TH.ModelArguments, TH.DataArguments, TH.TrainingArguments), argv)
    assert t.lora_enable is True and t.lora_r == 128 and t.mm_projector_lr == 0.0 and m.loss_alpha == 0.4
    assert t.warmup_ratio == 0.03 and t.bf16 and t.gradient_accumulation_steps == 4 and d.image_aspect_ratio == "pad"
    with pytest.raises(ValueError):
        TH.parse_args_into_dataclasses((TH.ModelArguments,), ["--no_such_flag", "1"])


def test_cosine_schedule():
    from halva_amd.dpa import cosine_with_warmup
    assert cosine_with_warmup(0, 100, 0.03) == 0.0
    assert cosine_with_warmup(3, 100, 0.03) == 1.0
    assert abs(cosine_with_warmup(100, 100, 0.03)) < 1e-12
    assert abs(cosine_with_warmup(51, 100, 0.03) - 0.5 * (1 + np.cos(np.pi * 48 / 97))) < 1e-12


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: kernels reject host tensors; a missing library raises."""
    from halva_amd import hip, kernels
    if not torch.cuda.is_available():
        with pytest.raises(hip.HalvaHipError):
            kernels.rmsnorm(torch.zeros(2, 64, dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16), 1e-5)
    import os
    old = hip.LIB_PATH
    hip.LIB_PATH, saved = "/nonexistent/libhalva_hip.so", hip._lib
    hip._lib = None
    try:
        with pytest.raises(hip.HalvaHipError):
            hip.load()
    finally:
        hip.LIB_PATH, hip._lib = old, saved


def test_bench_synthetic_batch_follows_baseline_layout():
    """[BOS, 34 prompt, <image>, 12 question, 5 'ASSISTANT:', R response, EOS], T = 2048 after the splice,
    six 3-token phrases at 40 + 200k, neg == pos outside the phrases - and what the prefix-sharing plan makes of it."""
    import numpy as np
    import bench
    from halva_amd import splice as SP
    from halva_amd.dpa import concat_pos_neg, phrase_slots
    b = bench.synthetic_batch(2, 7)
    ids, neg = b["input_ids"].numpy(), b["neg_input_ids"].numpy()
    assert ids.shape == (2, 1 + 34 + 1 + 12 + 5 + 1419 + 1) and (ids[:, 0] == 1).all() and (ids[:, 35] == -200).all() and (ids[:, -1] == 2).all()
    off = 1 + 34 + 1 + 12 + 5
    diff = ids != neg
    spans = np.zeros_like(diff)
    for k in range(6):
        spans[:, off + 40 + 200 * k: off + 43 + 200 * k] = True
    assert not (diff & ~spans).any() and (b["pos_signs"].numpy() > 0).sum() == 2 * 18
    assert (b["labels"].numpy()[:, :off] == -100).all() and (b["labels"].numpy()[:, off:] == ids[:, off:]).all()
    c_ids, c_lab, c_att, c_sig = concat_pos_neg(b)
    plan = SP.plan_splice(c_ids, c_att, c_lab, c_sig, 576, 2048, "right", image_map=[0, 1, 0, 1])
    assert plan.T == 2048 and plan.seq_len.tolist() == [2048] * 4
    assert len(phrase_slots(plan.signs.numpy()[:2, 1:])) == 6
    pk = SP.pack_pairs(plan)
    assert pk.br_a.tolist() == [628 + 40] * 2 and pk.br_b.tolist() == [2048] * 2       # shared: prefix + the first 40 response tokens
    assert pk.rows_packed == 2 * (2048 + 2048 - 668) and pk.rows_unpacked == 4 * 2048


def test_bench_visible_pairs_counts_what_the_branch_mask_lets_through():
    """bench.py's FLOP accounting for packed [prefix | A | pad | B] rows against a brute-force count of the attention mask
    (rows >= br_b do not see [br_a, br_b); include/halva_hip.h)."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for T, a, b, n in [(200, 37, 128, 200), (192, 60, 128, 150), (128, 5, 64, 64), (100, 0, 64, 100)]:
        brute = sum(1 for q in range(n) for k in range(q + 1) if not (q >= b and a <= k < b) and (k < min(n, b) or k >= b))
        # keys in [min(n, b), b) are padding of the A part: never valid
        assert bench.visible_pairs(T, a, b, n) == brute, (T, a, b, n)
    assert bench.visible_pairs(2048) == 2048 * 2049 // 2


def test_tile_dma_source_swizzle_inverts_the_lds_tile_layout():
    """csrc/sdpa.hip: an LDS-DMA request writes 1 KiB at (wave-uniform base + 16 * lane), so lane l of chunk c must FETCH the 16 bytes
    whose tile_off is 1024 c + 16 l (stage_tile_dma / TileDma::init).  Integer mirror of both formulas: the source map is the exact
    inverse of tile_off, a bijection onto the tile, and a wave's chunks differ by whole rows (what lets one lane offset serve them)."""
    for D in (128, 64):
        subrow = (D // 32) * 512

        def tile_off(row, ch):
            return subrow * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3))

        def source_of(c, lane):
            o = 1024 * c + 16 * lane
            band, rem = o // subrow, o % subrow
            row = 8 * band + ((rem % 512) >> 6)
            ch = 4 * (rem // 512) + (((rem >> 4) & 3) ^ ((row >> 2) & 3))
            return row, ch

        for rows in (64, 128):
            chunks = rows * D * 2 // 1024
            seen = set()
            for c in range(chunks):
                for lane in range(64):
                    row, ch = source_of(c, lane)
                    assert 0 <= row < rows and 0 <= ch < D // 8
                    assert tile_off(row, ch) == 1024 * c + 16 * lane
                    seen.add((row, ch))
            assert len(seen) == rows * D // 8
            for nw in (4, 8):                       # waves sharing a tile: chunk (wave + nw i) = chunk `wave` shifted by i * rows_per_i rows
                if (1024 * nw) % subrow:
                    continue
                rows_per_i = 8 * (1024 * nw // subrow)
                for wave in range(nw):
                    for i in range(chunks // nw):
                        for lane in (0, 17, 63):
                            r0, c0 = source_of(wave, lane)
                            assert source_of(wave + nw * i, lane) == (r0 + i * rows_per_i, c0)


def test_projector_builder_accepts_every_reference_type():
    """reference llava/model/multimodal_projector/builder.py:33-51: 'linear', 'mlp<N>x_gelu', 'identity', else ValueError; parameter
    names as the reference's nn.Linear / nn.Sequential give them (they are the keys of non_lora_trainables.bin)."""
    import types
    from halva_amd import clip
    def names(kind):
        m = clip.build_vision_projector(types.SimpleNamespace(mm_projector_type=kind, mm_hidden_size=8, hidden_size=16), device="cpu")
        return [n for n, _ in m.named_parameters()], m
    assert names("linear")[0] == ["weight", "bias"]
    assert names("mlp1x_gelu")[0] == ["0.weight", "0.bias"]
    assert names("mlp2x_gelu")[0] == ["0.weight", "0.bias", "2.weight", "2.bias"]
    assert names("mlp3x_gelu")[0] == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]
    ident = names("identity")[1]
    x = torch.randn(2, 3)
    assert ident(x) is x and ident.config == {"mm_projector_type": "identity"}
    with pytest.raises(ValueError):
        names("resblock")


def test_initialize_vision_tokenizer_resizes_and_mean_initialises():
    """reference llava/model/llava_arch.py:398-440 on a stand-in model: <im_patch> then <im_start>/<im_end> are appended, both
    vocabulary-sized matrices grow, the two start/end rows become the mean of the rows before them; both flags False = no-op."""
    import types
    from halva_amd.llava_model import LlavaMetaForCausalLM

    class Tok:
        def __init__(self):
            self.v = ["a%d" % i for i in range(10)]
        def add_tokens(self, toks, special_tokens=False):
            new = [t for t in toks if t not in self.v]
            self.v += new
            return len(new)
        def __len__(self):
            return len(self.v)

    class M(LlavaMetaForCausalLM):
        def __init__(self):
            g = torch.Generator().manual_seed(0)
            self.emb = torch.nn.Embedding(10, 4)
            self.head = torch.nn.Linear(4, 10, bias=False)
            self.emb.weight.data = torch.randn(10, 4, generator=g)
            self.head.weight.data = torch.randn(10, 4, generator=g)
            self.config = types.SimpleNamespace(vocab_size=10)
        def get_input_embeddings(self):
            return self.emb
        def get_output_embeddings(self):
            return self.head

    m, tok = M(), Tok()
    e0, h0 = m.emb.weight.data.clone(), m.head.weight.data.clone()
    m.initialize_vision_tokenizer(types.SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=False), tok)
    assert len(tok) == 10 and m.emb.weight.shape[0] == 10
    m.initialize_vision_tokenizer(types.SimpleNamespace(mm_use_im_patch_token=True, mm_use_im_start_end=True, tune_mm_mlp_adapter=False,
                                                        pretrain_mm_mlp_adapter=None), tok)
    assert len(tok) == 13 and m.emb.weight.shape == (13, 4) and m.head.weight.shape == (13, 4) and m.config.vocab_size == 13
    assert torch.equal(m.emb.weight.data[:10], e0) and torch.equal(m.head.weight.data[:10], h0)
    for w in (m.emb.weight.data, m.head.weight.data):
        assert torch.allclose(w[-2:], w[:-2].mean(0, keepdim=True).expand(2, 4), atol=1e-6)
    with pytest.raises(NotImplementedError):
        M().initialize_vision_tokenizer(types.SimpleNamespace(mm_use_im_patch_token=False, mm_use_im_start_end=True,
                                                              tune_mm_mlp_adapter=True), Tok())


def test_output_file_keys_match_the_reference_helpers():
    """f2 (PEFT-format outputs): the key sets and shapes of adapter_model.bin / non_lora_trainables.bin and the adapter's
    target_modules, against tests/golden/peft_state_names.json - produced by the REFERENCE's get_peft_state_maybe_zero_3 /
    get_peft_state_non_lora_maybe_zero_3 / find_all_linear_names (llava/train/train_halva.py:116-169, called as train() does at
    :1230-1240) on the reference's tiny LLaVA (tests/golden/make_golden.py:gen_peft_names).  peft itself is absent offline: its module
    wrapping is emulated there, the selection logic is the reference's."""
    from halva_amd.llava_model import build_random_llava
    import llava.train.train_halva as TH
    d = load_json("peft_state_names.json")
    vis = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=28, patch_size=14, layer_norm_eps=1e-5)
    m = build_random_llava(d["llama_cfg"], vis, lora_r=d["lora_r"], lora_alpha=8, seed=1, device="cpu", max_len=64)
    adapter = TH.get_peft_state_maybe_zero_3(m, "none")
    assert {k: list(v.shape) for k, v in adapter.items()} == d["adapter_model_bin"]
    assert list(adapter) == list(d["adapter_model_bin"])                       # the same order, too
    non_lora = TH.get_peft_state_non_lora_maybe_zero_3(m)
    assert {k: list(v.shape) for k, v in non_lora.items()} == d["non_lora_trainables"]
    assert sorted(TH.find_all_linear_names(m)) == d["target_modules"]
    assert {k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight") for k in adapter} == set(d["lora_state"])
