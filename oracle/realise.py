"""Oracle (test infrastructure): several bf16 REALISATIONS of the reference arithmetic on the CPU.

The reference trains with `--bf16 True` (reference src/hallava_7b.sh:34): its own numbers on a GPU are ONE draw of the rounding noise a bf16
residual stream carries - which draw depends on the order every contraction is summed in (the GEMM library's tiling, split-K, the hardware).
`oracle.dpa.compute_loss(dtype=bf16)` on the CPU is another draw; ONE such draw is not a distribution ("weak": the long step
fixture's margin bound was 2.5 x a single CPU realisation).  This module produces more of them without changing the mathematics: inside the
context every `torch.nn.functional.linear` the oracle issues (oracle/nets.py: lora_linear, the projector, lm_head; reference
llava/model/language_model/modelling_llama.py:185-420 via nn.Linear)
  * sums its contraction in another ORDER (a fixed permutation of the input features, applied to the activation and the weight alike), and / or
  * is computed as `chunks` partial products over slices of the contraction, each rounded to the tensor dtype and added in that dtype -
    what a split-K GEMM with bf16 partial results does.
In fp32 every realisation equals the plain oracle to ~1e-6 (tests/test_oracle_vs_golden.py); in bf16 they spread, and the spread - not one
sample - is the floor the product's own bf16 execution is held to (tests/test_dpa_step_gpu.py).
"""
import contextlib

import torch
import torch.nn.functional as F

# the set used by the tests: name -> (permutation seed or None, chunks).  FROZEN (round 6): a bound defined as "the largest of the set"
# loosens whenever a realisation is added, so nothing is added or removed any more; tests/test_oracle_vs_golden.py pins the names.
REALISATIONS = {
    "plain": (None, 1),
    "perm1": (1, 1),
    "perm2": (2, 1),
    "chunk2": (None, 2),
    "chunk4": (None, 4),
    "perm3_chunk2": (3, 2),
    "perm4_chunk3": (4, 3),
    "perm5": (5, 1),
    "perm6": (6, 1),
    "perm7": (7, 1),
    "perm8": (8, 1),
    "chunk3": (None, 3),
}


# The realisations the test BOUNDS are taken over (round 6): the permutation-only ones.  The chunked ones round split-K partial products
# to bf16 and add them in bf16 - no GEMM the reference runs does that (cuBLAS, hipBLASLt and flash-attn accumulate in fp32 and round once) - so they
# stay in the table as information, not as draws of the reference.  (On the smallest fixture the six permutations give the SAME numbers: with K = 64
# and fp32 accumulation the order of a sum does not reach the bf16 result - the legitimate spread there is two draws wide, and that is the finding.)
BOUND_SET = tuple(n for n, (_, chunks) in REALISATIONS.items() if chunks == 1)


@contextlib.contextmanager
def realisation(name):
    """Context: F.linear sums in the order / with the partial roundings of REALISATIONS[name]."""
    seed, chunks = REALISATIONS[name]
    if seed is None and chunks == 1:
        yield
        return
    orig = F.linear
    perms = {}

    def linear(x, w, b=None):
        K = x.shape[-1]
        if seed is not None:
            if K not in perms:
                perms[K] = torch.randperm(K, generator=torch.Generator().manual_seed(1000 * seed + K % 997))
            p = perms[K]
            x, w = x[..., p], w[:, p]
        if chunks > 1 and K >= 2 * chunks:
            edges = [K * i // chunks for i in range(chunks + 1)]
            y = orig(x[..., edges[0]:edges[1]], w[:, edges[0]:edges[1]])
            for a, e in zip(edges[1:-1], edges[2:]):
                y = y + orig(x[..., a:e], w[:, a:e])
        else:
            y = orig(x, w)
        return y if b is None else y + b

    F.linear = linear
    try:
        yield
    finally:
        F.linear = orig
