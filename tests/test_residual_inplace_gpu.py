"""The decoder block's residual adds accumulate onto a buffer of the block's own (the copy the fork kernel writes; without autograd
the previous block's output) instead of copying the residual into a fresh GEMM output first.  Reference semantics:
`hidden_states = residual + hidden_states` (modelling_llama.py:395-417).  Checked here: same hidden states and the same LoRA
gradients as the path that leaves the residual to autograd (HALVA_NORM_FORK=0: out-of-place addmm), and the caller's input is
never written to."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)
from golden_util import load_npz  # noqa: E402
from model_util import build_product_models  # noqa: E402


def _run(lm, x0, ss, sl, fork, grad):
    os.environ["HALVA_NORM_FORK"] = fork
    try:
        from halva_amd.llama import lora_named_parameters
        for _, p in lora_named_parameters(lm):
            p.grad = None
        x = x0.clone().requires_grad_(grad)
        keep = x.detach().clone()
        with torch.set_grad_enabled(grad):
            h = lm.model.run_layers(x, ss, sl)
        assert torch.equal(x.detach(), keep), "the caller's hidden states were written to"
        grads = None
        if grad:
            (h.float() * torch.linspace(-1, 1, h.shape[-1], device=h.device)).sum().backward()
            grads = [p.grad.clone() for _, p in lora_named_parameters(lm)] + [x.grad.clone()]
        return h.detach().clone(), grads
    finally:
        os.environ["HALVA_NORM_FORK"] = "1"


def test_inplace_residual_matches_the_out_of_place_path():
    z = load_npz("dpa_step_d128_init.npz")
    pol, _, _ = build_product_models(z, device="cuda:0")
    from halva_amd import dpa
    dpa.set_grad_sink(pol, False)
    d = pol.config.hidden_size
    S, T = 3, 96
    g = torch.Generator().manual_seed(7)
    x0 = (torch.randn(S, T, d, generator=g) * 0.5).to(torch.bfloat16).cuda()
    ss = torch.tensor([0, 5, 0], dtype=torch.int32, device="cuda")
    sl = torch.tensor([96, 80, 33], dtype=torch.int32, device="cuda")
    h1, g1 = _run(pol, x0, ss, sl, "1", True)       # fork kernel + in-place accumulate
    h0, g0 = _run(pol, x0, ss, sl, "0", True)       # plain norm, out-of-place addmm, autograd sums the residual gradient
    hn, _ = _run(pol, x0, ss, sl, "1", False)       # no autograd: in place on the previous block's output
    # the forward arithmetic is the same GEMM with beta = 1 on the same values: bitwise equal
    assert torch.equal(h1, h0) and torch.equal(hn, h0)
    for a, b in zip(g1, g0):
        # (the fork kernel adds the two gradients in one rounding where autograd rounds the norm's dx first: bf16 noise)
        assert float((a.float() - b.float()).abs().max()) <= 2e-2 * float(b.float().abs().max()) + 1e-6


@pytest.mark.parametrize("grad", [True, False])
def test_top_layer_on_the_kept_rows_only_matches_the_full_top_layer(grad):
    """run_layers(rows=...): the top decoder layer does its row-wise part (o projection, post-attention norm, MLP, final norm) for the
    rows the loss reads only.  Same rows as gathering from the full [S, T, d] result (the reference computes every row,
    modelling_llama.py:580-705, and the loss reads logits[labels != -100], halva_trainer.py:522-537), same LoRA and input gradients
    up to the bf16 noise of GEMMs that see another row count."""
    from halva_amd import dpa
    from halva_amd.llama import lora_named_parameters
    z = load_npz("dpa_step_d128_init.npz")
    pol, _, _ = build_product_models(z, device="cuda:0")
    dpa.set_grad_sink(pol, False)
    d = pol.config.hidden_size
    S, T = 3, 96
    g = torch.Generator().manual_seed(11)
    x0 = (torch.randn(S, T, d, generator=g) * 0.5).to(torch.bfloat16).cuda()
    ss = torch.tensor([0, 5, 0], dtype=torch.int32, device="cuda")
    sl = torch.tensor([96, 80, 33], dtype=torch.int32, device="cuda")
    rows = torch.tensor([3, 17, 18, 95, 96 + 5, 96 + 60, 96 + 84, 2 * 96, 2 * 96 + 32], dtype=torch.int64, device="cuda")
    w = torch.linspace(-1, 1, d, device="cuda")

    def run(pruned):
        for _, p in lora_named_parameters(pol):
            p.grad = None
        x = x0.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            if pruned:
                h = pol.model.run_layers(x, ss, sl, rows=rows)
                assert h.shape == (rows.numel(), d)
            else:
                h = pol.model.run_layers(x, ss, sl).view(-1, d).index_select(0, rows)
        gr = None
        if grad:
            (h.float() * w).sum().backward()
            gr = [p.grad.clone() for _, p in lora_named_parameters(pol)] + [x.grad.clone()]
        return h.detach().float(), gr

    h1, g1 = run(True)
    h0, g0 = run(False)
    assert float((h1 - h0).abs().max()) <= 2e-2 * float(h0.abs().max())
    if grad:
        assert all(torch.isfinite(a).all() for a in g1)
        for a, b in zip(g1, g0):
            assert float((a.float() - b.float()).abs().max()) <= 2e-2 * float(b.float().abs().max()) + 1e-6
