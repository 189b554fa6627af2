"""Host-side prefix sharing plan (halva_amd/splice.py:pack_pairs): integer bookkeeping only."""
import numpy as np
import torch

from halva_amd import splice as SP


def _plan(rows, n_patch=3, max_len=None):
    L = max(len(r) for r in rows)
    ids = np.zeros((len(rows), L), dtype=np.int64)
    att = np.zeros((len(rows), L), dtype=bool)
    for i, r in enumerate(rows):
        ids[i, :len(r)] = r
        att[i, :len(r)] = True
    lab = np.where(att, ids, -100)
    g = len(rows) // 2
    return SP.plan_splice(ids, att, lab, np.zeros_like(ids), n_patch, max_len, "right", image_map=list(range(g)) * 2)


def test_pack_pairs_layout_and_maps():
    pos = [[1, 5, -200, 6, 7, 8, 9, 2], [1, 4, -200, 3, 2]]
    neg = [[1, 5, -200, 6, 7, 11, 12, 13, 2], [1, 4, -200, 3, 2]]          # pair 0 differs from token 5 on; pair 1 identical
    plan = _plan(pos + neg)
    pk = SP.pack_pairs(plan, align=8)
    src = plan.src.numpy().reshape(plan.S, plan.T)
    lens = plan.seq_len.numpy()
    assert pk.br_a.tolist() == [7, 7]                 # 2 text + 3 patch rows + 2 text rows shared; pair 1: whole row (len 7)
    assert pk.br_b.tolist() == [16, 8]                # ceil8(len(pos row))
    assert pk.seq_len.tolist() == [16 + (lens[2] - 7), 8 + 0]
    P = pk.src.numpy().reshape(2, pk.T)
    np.testing.assert_array_equal(P[0, :lens[0]], src[0, :lens[0]])                       # prefix + correct rest
    assert (P[0, lens[0]:16] == -1).all()                                                 # padding rows are zero vectors
    np.testing.assert_array_equal(P[0, 16:16 + lens[2] - 7], src[2, 7:lens[2]])           # hallucinated rest
    pp = pk.pos.numpy().reshape(2, pk.T)
    np.testing.assert_array_equal(pp[0, :lens[0]], np.arange(lens[0]))
    np.testing.assert_array_equal(pp[0, 16:16 + lens[2] - 7], np.arange(7, lens[2]))      # RoPE positions continue from the prefix
    # every (row, position) of the un-packed batch finds its hidden state; shared rows map to the same packed row
    for r in range(plan.S):
        assert (pk.row_of[r, :lens[r]] >= 0).all() and (pk.row_of[r, lens[r]:] == -1).all()
    np.testing.assert_array_equal(pk.row_of[0, :7], pk.row_of[2, :7])
    np.testing.assert_array_equal(pk.row_of[1, :lens[1]], pk.row_of[3, :lens[3]])         # identical pair: one set of rows
    assert pk.row_of[2, 7] == 16 and pk.row_of[0, 7] == 7
    assert pk.rows_unpacked == int(lens.sum()) and pk.rows_packed == int(pk.seq_len.sum())


def test_pack_pairs_truncated_rows():
    """Rows cut at tokenizer_model_max_length after the splice (reference llava_arch.py:334-339) pack like any other."""
    pos = [[1, -200, 5, 6, 7, 8, 9, 10, 11, 2]]
    neg = [[1, -200, 5, 6, 20, 21, 22, 23, 24, 2]]
    plan = _plan(pos + neg, n_patch=4, max_len=9)
    assert plan.T == 9
    pk = SP.pack_pairs(plan, align=4)
    assert pk.br_a.tolist() == [7] and pk.br_b.tolist() == [12] and pk.seq_len.tolist() == [14]      # 1 + 4 patches + "5 6" shared


def test_positions_follow_from_the_branch_points():
    """halva_sdpa_branch_bwd_rope / halva_rope_qk_branch (round 5) take no position table: row t of a packed row sits at position t, and at
    br_a + (t - br_b) once t >= br_b (csrc/sdpa.hip:rope_position).  That must be what pack_pairs writes into `pos` for EVERY row that carries a
    token (the padding rows between the correct rest and br_b carry zero vectors: their gradients are exactly zero, their position is immaterial)."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        g = int(rng.integers(1, 4))
        pos, neg = [], []
        for _ in range(g):
            n_pre = int(rng.integers(2, 9))
            pre = [1] + list(rng.integers(3, 50, n_pre)) + [-200] + list(rng.integers(3, 50, int(rng.integers(1, 6))))
            pos.append(pre + list(rng.integers(3, 50, int(rng.integers(1, 40)))) + [2])
            neg.append(pre + list(rng.integers(50, 99, int(rng.integers(1, 40)))) + [2])
        plan = _plan(pos + neg, n_patch=int(rng.integers(1, 6)))
        pk = SP.pack_pairs(plan, align=64)
        P = pk.src.numpy().reshape(g, pk.T)
        pp = pk.pos.numpy().reshape(g, pk.T)
        for i in range(g):
            a, b, n = int(pk.br_a[i]), int(pk.br_b[i]), int(pk.seq_len[i])
            t = np.arange(pk.T)
            formula = np.where(t >= b, a + (t - b), t)
            carries = (t < n) & (P[i] != -1)
            np.testing.assert_array_equal(pp[i][carries], formula[carries])
            assert b % 64 == 0 and a <= b
