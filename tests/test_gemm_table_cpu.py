"""The shipped library-GEMM solution table (halva_amd/gemm_tuning.py): well-formed, for gfx950, and a no-op without a GPU."""
import os

import torch

from halva_amd import gemm_tuning as G


def test_shipped_table_is_well_formed():
    validators, rows = G.table_entries()
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "ROCBLAS_VERSION"} <= set(validators)
    assert len(rows) >= 40
    for op, key, sol, ms in rows:
        assert op.split("_")[0] in ("GemmTunableOp", "GemmAndBiasTunableOp", "GemmStridedBatchedTunableOp") and "BFloat16" in op
        assert key[:2] in ("tn", "nt", "nn", "tt") and sol.split("_")[0] in ("Gemm", "Default") and ms > 0
    # the table covers the headline workload: the fused q/k/v + LoRA-A projection of the packed 7B step
    assert any("_12288_27424_4480_" in key for _, key, _, _ in rows)


def test_loader_is_a_noop_without_gpu(monkeypatch):
    monkeypatch.setitem(G._state, "loaded", None)
    if not torch.cuda.is_available():
        assert G.enable_tuned_gemms() is None
    monkeypatch.setenv("HALVA_GEMM_TABLE", "0")
    assert G.table_path() is None
    monkeypatch.delenv("HALVA_GEMM_TABLE")
    assert os.path.exists(G.table_path())
