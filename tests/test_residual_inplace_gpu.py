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
