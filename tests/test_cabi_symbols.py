"""The C-ABI library loads on a GPU-less box and exports every symbol include/halva_hip.h declares (no compute)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    lib = os.path.join(ROOT, "halva_amd", "libhalva_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    return lib


def test_header_symbols_exported_and_bound():
    lib = ctypes.CDLL(_ensure_built())
    header = open(os.path.join(ROOT, "include", "halva_hip.h")).read()
    names = re.findall(r"^(?:int|const char\*)\s+(halva_[a-z0-9_]+)\s*\(", header, flags=re.M)
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/halva_hip.h is not exported" % n
    lib.halva_abi_version.restype = ctypes.c_int
    assert lib.halva_abi_version() == 1
    from halva_amd import hip
    assert set(hip.SIGNATURES) | {"halva_last_error", "halva_abi_version"} == set(names)
    hip.load()                                  # argtypes bind for every entry point


def test_argument_validation_without_gpu():
    """Entry points validate before launching: bad arguments return HALVA_ERR_INVALID_ARG with a message."""
    from halva_amd import hip
    lib = hip.load()
    rc = lib.halva_rmsnorm_fwd(None, None, None, None, 4, 64, 1e-5, None)
    assert rc == -1 and b"null pointer" in lib.halva_last_error()
    rc = lib.halva_sdpa_causal_fwd(1, 1, 1, None, None, 1, 16, 2, 48, 0.0, None)
    assert rc == -1 and b"head_dim" in lib.halva_last_error()
