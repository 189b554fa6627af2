"""Library-GEMM solution tables.  The step's large matmuls are plain hipBLASLt / rocBLAS calls issued by PyTorch-ROCm; which of a
library's several hundred kernels serves a given (layout, m, n, k, ld) is the library's heuristic unless PyTorch's TunableOp has a
measured answer.  A table measured on MI355X for the shapes of the bench workloads (the headline one - BASELINE.json configs[1]:
LLaVA-1.5-7B, T=2048, 16 pairs/GPU, prefix sharing on - and the 13B / VILA-13B extras) is shipped under halva_amd/tuned/ and loaded
when the engine is built: +3.5 % step throughput on the headline workload,
results within the bf16 noise of the default kernels (same arithmetic, different tiling).  Shapes not in the table - other models,
other batch layouts - keep the library default; nothing is tuned at run time unless HALVA_GEMM_TUNE=1 asks for it.

  HALVA_GEMM_TABLE=<csv>   use this table instead of the shipped one ("0": none)
  HALVA_GEMM_TUNE=1        measure missing shapes on the fly (slow first steps) and write them back to HALVA_GEMM_TABLE
  tools/tune_gemms.sh      regenerates the shipped table

The table carries the PyTorch / ROCm / hipBLASLt / rocBLAS versions and the GPU arch it was measured with; PyTorch refuses a table
whose validators do not match the running stack, and the defaults stay in force."""
import os

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gfx950_tunableop.csv")
_state = {"loaded": None}


def table_path():
    p = os.environ.get("HALVA_GEMM_TABLE", SHIPPED)
    return None if p in ("0", "") else p


def enable_tuned_gemms(path=None):
    """Load the solution table once per process (no-op without a GPU or a table).  Returns the path in force, or None."""
    if _state["loaded"] is not None:
        return _state["loaded"] or None
    _state["loaded"] = ""
    path = path or table_path()
    tune = os.environ.get("HALVA_GEMM_TUNE", "0") == "1"
    if not torch.cuda.is_available() or (path is None and not tune):
        return None
    from torch.cuda import tunable
    if path is not None and os.path.exists(path):
        tunable.enable(True)
        tunable.tuning_enable(tune)
        ok = tunable.read_file(path)
        if not ok and not tune:              # validators differ (another ROCm / library build): keep the library defaults
            tunable.enable(False)
            return None
        _state["loaded"] = path
    elif tune:
        tunable.enable(True)
        tunable.tuning_enable(True)
        _state["loaded"] = path or ""
    if tune and path is not None:
        tunable.set_filename(path, insert_device_ordinal=False)
    return _state["loaded"] or None


def table_entries(path=None):
    """(validators, rows) of a table file: rows are (op, shape key, solution, measured ms)."""
    validators, rows = {}, []
    with open(path or SHIPPED) as f:
        for line in f:
            parts = line.rstrip("\n").split(",")
            if parts[0] == "Validator":
                validators[parts[1]] = ",".join(parts[2:])
            elif len(parts) >= 4:
                rows.append((parts[0], parts[1], parts[2], float(parts[3])))
    return validators, rows
