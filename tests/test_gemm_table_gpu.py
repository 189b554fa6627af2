"""The measured GEMM solution table on the GPU: it loads against this stack's validators, nothing is tuned at run time, and a tabled
shape still computes the product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_table_loads_and_tabled_gemm_is_right(monkeypatch):
    from torch.cuda import tunable
    from halva_amd import gemm_tuning as G
    monkeypatch.setitem(G._state, "loaded", None)
    path = G.enable_tuned_gemms()
    assert path and path.endswith(".csv"), "the shipped table was refused (validators of another ROCm / library build?)"
    assert tunable.is_enabled() and not tunable.tuning_is_enabled()
    assert len(tunable.get_results()) >= 40
    torch.manual_seed(0)
    x = torch.randn(27424, 4480, device="cuda").to(torch.bfloat16)      # packed rows of the 7B step x [hidden | LoRA-A columns]
    w = torch.randn(12288, 4480, device="cuda").to(torch.bfloat16)      # fused q/k/v: a tabled (tn_12288_27424_4480) shape
    y = torch.mm(x, w.t())
    rows = torch.arange(0, 27424, 997, device="cuda")
    ref = x[rows].float() @ w.float().t()
    err = float((y[rows].float() - ref).norm() / ref.norm())
    assert err < 4e-3, err
