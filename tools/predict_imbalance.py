#!/usr/bin/env python3
"""Predicted rank imbalance of the 8-GPU recipe from ONE GPU's timings.  NOT a scaling measurement.

The reference deals its batches with `LengthGroupedSampler(train_batch_size, world_size * gradient_accumulation_steps, lengths,
group_by_modality=True)` (reference llava/train/halva_trainer.py:60-152,261-274) and every rank then takes its share of each
megabatch; the ranks meet at the gradient exchange, so an optimizer step lasts as long as its SLOWEST rank.  This tool
  1. draws response lengths from a HALVA-like mixture (an ASSUMPTION - data/data.json is not in the reference tree: yes/no answers with a
     short explanation, one-sentence and detailed image descriptions; reference samples from a LLaVA-instruct-like, broader distribution),
  2. builds the global order with the PRODUCT's sampler (llava/train/halva_trainer.py: bit-exact with the reference's, tests/test_host_logic.py)
     and deals the micro-batches to `world` ranks exactly as the trainer does (halva_amd/dp.py:shard_batches),
  3. runs every rank's micro-batches of the first `--steps` optimizer steps SEQUENTIALLY on this GPU through the real engine (forward +
     backward, 7B geometry; each micro-batch once untimed, then timed with HIP events),
  4. reports per step mean / max over the ranks; sum(mean) / sum(max) = the predicted parallel efficiency of the compute part (no
     communication, no host effects: the exchange overlaps with the last backward, halva_amd/dp.py).
Output: one JSON document (profiles/r05_natural_length.json is made from it)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def halva_like_lengths(n, g):
    """(response tokens of the training pair, response tokens of the reference sample) per index.  Mixture (assumed): 40 % yes/no + short
    explanation (8-40 tokens), 30 % one-sentence descriptions (15-60), 30 % detailed descriptions (80-260); reference samples 10-400."""
    kind = torch.multinomial(torch.tensor([0.4, 0.3, 0.3]), n, replacement=True, generator=g)
    lo = torch.tensor([8, 15, 80])[kind]
    hi = torch.tensor([40, 60, 260])[kind]
    u = torch.rand(n, generator=g)
    resp = (lo + (hi - lo) * u).long()
    ref = (10 + 390 * torch.rand(n, generator=g) ** 2).long()      # skewed to short
    return resp.tolist(), ref.tolist()


def ragged_batch(idx, resp, ref, seed, vocab=32000, image=336):
    """bench.synthetic_batch's layout ([BOS, 34 prompt, <image>, 12 question, 5 'ASSISTANT:', R response, EOS]) with a response length per
    sample, right-padded (attention_mask False, labels -100 on the padding); one phrase per ~60 response tokens (at least one), 3 tokens each."""
    g = torch.Generator().manual_seed(seed)
    pre, post = 1 + 34, 12 + 5
    off = pre + 1 + post
    B = len(idx)

    def rows(lengths):
        L = off + max(lengths) + 1
        ids = torch.zeros(B, L, dtype=torch.long)
        mask = torch.zeros(B, L, dtype=torch.bool)
        labels = torch.full((B, L), -100, dtype=torch.long)
        for b, R in enumerate(lengths):
            n = off + R + 1
            x = torch.randint(3, vocab, (n,), generator=g)
            x[0], x[pre], x[-1] = 1, -200, 2
            ids[b, :n], mask[b, :n] = x, True
            labels[b, off:n] = x[off:]
        return ids, mask, labels
    r_pos = [resp[i] for i in idx]
    ids, mask, labels = rows(r_pos)
    neg, neg_labels = ids.clone(), labels.clone()
    signs = torch.zeros_like(ids)
    for b, R in enumerate(r_pos):
        for k in range(max(1, R // 60)):
            s = off + 2 + 60 * k
            if s + 3 <= off + R:
                signs[b, s:s + 3] = k + 1
                neg[b, s:s + 3] = torch.randint(3, vocab, (3,), generator=g)
                neg_labels[b, s:s + 3] = neg[b, s:s + 3]
    rids, rmask, rlabels = rows([ref[i] for i in idx])
    return dict(input_ids=ids, labels=labels, attention_mask=mask, neg_input_ids=neg, neg_labels=neg_labels, neg_attention_mask=mask.clone(),
                pos_signs=signs, neg_signs=signs.clone(), ref_input_ids=rids, ref_labels=rlabels, ref_attention_mask=rmask,
                images=torch.randn(B, 3, image, image, generator=g), ref_images=torch.randn(B, 3, image, image, generator=g))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--per-device", type=int, default=8, help="pairs per micro-batch (reference recipe on 4 GPUs: 4 x 4 accumulation; 8 x 1 on 8 GPUs keeps its global 64)")
    ap.add_argument("--accum", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--samples", type=int, default=4096)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--layers", type=int, default=0)
    ap.add_argument("--no-grouping", action="store_true", help="plain random order instead of the length-grouped sampler (for comparison)")
    args = ap.parse_args()
    import bench
    from halva_amd import dp, dpa, hip
    from halva_amd.llava_model import build_random_llava
    from llava.train.halva_trainer import LengthGroupedSampler
    hip.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(args.seed)
    resp, ref = halva_like_lengths(args.samples, g)
    if args.no_grouping:
        order = torch.randperm(args.samples, generator=g).tolist()
    else:      # the trainer's call: world_size * gradient_accumulation_steps chunks per megabatch, `lengths` = the training samples' own (all multimodal: > 0)
        order = list(LengthGroupedSampler(args.per_device, args.world * args.accum, lengths=resp, generator=g, group_by_modality=True))
    micro = [order[i:i + args.per_device] for i in range(0, len(order) - args.per_device + 1, args.per_device)]
    geo = dict(bench.LLAMA_7B)
    if args.layers:
        geo["num_hidden_layers"] = args.layers
    policy = build_random_llava(geo, bench.CLIP_L_336, lora_r=128, lora_alpha=256, seed=1234, device=dev, max_len=2048)
    refm = build_random_llava(geo, bench.CLIP_L_336, seed=1234, device=dev, max_len=2048, share_base_from=policy)
    with torch.no_grad():
        gB = torch.Generator(device=dev).manual_seed(99)
        for layer in policy.model.layers:
            for _, grp in layer.groups():
                for n in grp.names:
                    getattr(grp, n).lora_B["default"].weight.normal_(0.0, 0.01, generator=gB)
    flat = dpa.FlatTrainables(dpa.trainable_named_parameters(policy))
    dpa.bind_model(flat, policy)
    dpa.set_grad_sink(policy, True)
    eng = dpa.DPAEngine(policy, refm, 0.4, pairs_per_group=args.per_device, ref_rows_per_group=2 * args.per_device)

    def run(batch):
        flat.zero_grad()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.loss(batch, backward=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    steps = []
    for k in range(args.steps):
        per_rank, tok = [], []
        for r in range(args.world):
            t, n_tok = 0.0, 0
            for j in range(args.accum):
                gi = (k * args.accum + j) * args.world + r      # halva_amd/dp.py:shard_batches - rank r takes global micro-batches r, r + world, ...
                idx = micro[gi]
                batch = ragged_batch(idx, resp, ref, 7 + gi)
                batch["images"] = batch["images"].to(dev, torch.bfloat16)
                batch["ref_images"] = batch["ref_images"].to(dev, torch.bfloat16)
                run(batch)              # untimed: library heuristics, allocator growth for a new shape
                t += run(batch)
                n_tok += sum(2 * (53 + 576 + resp[i]) + 53 + 576 + ref[i] for i in idx)
            per_rank.append(t)
            tok.append(n_tok)
        steps.append({"ms_per_rank": [round(x, 2) for x in per_rank], "tokens_per_rank": tok, "mean_ms": sum(per_rank) / len(per_rank), "max_ms": max(per_rank)})
        print("step %d: per-rank ms %s  mean %.1f max %.1f  eff %.3f" % (k, steps[-1]["ms_per_rank"], steps[-1]["mean_ms"], steps[-1]["max_ms"],
                                                                        steps[-1]["mean_ms"] / steps[-1]["max_ms"]), file=sys.stderr, flush=True)
    eff = sum(s["mean_ms"] for s in steps) / sum(s["max_ms"] for s in steps)
    pairs = args.world * args.per_device * args.accum
    out = {"what": "PREDICTED from 1-GPU timings of every rank's micro-batches run sequentially; no scaling curve was measured",
           "world": args.world, "pairs_per_micro_batch": args.per_device, "accumulation": args.accum, "global_pairs_per_step": pairs,
           "sampler": "plain random order" if args.no_grouping else "LengthGroupedSampler(per_device, world * accum, group_by_modality=True) - the product's, bit-exact with the reference's",
           "length_model": "ASSUMED HALVA-like mixture (tools/predict_imbalance.py:halva_like_lengths): responses 8-260 tokens (mean %.0f), reference samples 10-400 (mean %.0f)"
                           % (sum(resp) / len(resp), sum(ref) / len(ref)),
           "steps": steps, "predicted_compute_efficiency": round(eff, 4),
           "predicted_pairs_per_s_at_world": round(pairs / (sum(s["max_ms"] for s in steps) / len(steps) / 1e3), 2),
           "one_gpu_pairs_per_s_same_work": round(pairs / (sum(sum(s["ms_per_rank"]) for s in steps) / len(steps) / 1e3), 2),
           "layers": geo["num_hidden_layers"]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
