"""ctypes binding of libhalva_hip.so (include/halva_hip.h) - the only way the product reaches the GPU kernels.

There is NO fallback: if the shared library is missing or an entry point fails, this module raises.
PyTorch supplies device memory and the HIP stream only (raw pointers cross the boundary).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HALVA_HIP_LIB", os.path.join(_HERE, "libhalva_hip.so"))
ABI_VERSION = 1
BF16, F32 = 0, 1

_P, _I, _L, _F = c_void_p, c_int, c_int64, c_float

# name -> argtypes (all return int), mirroring include/halva_hip.h
SIGNATURES = {
    "halva_rmsnorm_fwd": [_P, _P, _P, _P, _L, _I, _F, _P],
    "halva_rmsnorm_bwd": [_P, _P, _P, _P, _P, _L, _I, _P],
    "halva_rmsnorm_fwd_ld": [_P, _P, _P, _L, _P, _L, _I, _F, _P],
    "halva_rmsnorm_bwd_ld": [_P, _L, _P, _P, _P, _P, _L, _I, _P],
    "halva_rmsnorm_fwd_fork_ld": [_P, _P, _P, _L, _P, _P, _L, _I, _F, _P],
    "halva_rmsnorm_bwd_res_ld": [_P, _L, _P, _P, _P, _P, _P, _L, _I, _P],
    "halva_swiglu_fwd_ld": [_P, _P, _L, _L, _I, _P],
    "halva_swiglu_bwd_ld": [_P, _L, _P, _P, _L, _I, _P],
    "halva_sdpa_causal_fwd_ld": [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_causal_bwd_ld": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_branch_fwd": [_P, _P, _L, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_branch_bwd": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_branch_bwd_ws": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_branch_bwd_rope": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    "halva_rope_qk": [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "halva_rope_qk_branch": [_P, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "halva_swiglu_fwd": [_P, _P, _L, _I, _P],
    "halva_swiglu_bwd": [_P, _P, _P, _L, _I, _P],
    "halva_sdpa_causal_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_causal_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "halva_sdpa_full_fwd": [_P, _P, _I, _I, _I, _I, _F, _P],
    "halva_gemm_bf16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "halva_wgrad_accumulate": [_P, _L, _P, _L, _P, _I, _I, _L, _F, _P, _L, _P],
    "halva_wgrad_accumulate_batch": [_I, _P, _P, _L, _P],
    "halva_clip_patch_embed": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "halva_vit_patch_embed": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "halva_layernorm_fwd": [_P, _P, _P, _P, _P, _L, _I, _F, _P],
    "halva_layernorm_bwd_params": [_P, _P, _P, _P, _P, _L, _I, _P],
    "halva_downsample2x2": [_P, _P, _I, _I, _I, _P],
    "halva_image_preprocess": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "halva_gelu_bwd": [_P, _P, _P, _L, _I, _P],
    "halva_colsum": [_P, _P, _L, _I, _P],
    "halva_quick_gelu": [_P, _P, _L, _P],
    "halva_transpose_bf16": [_P, _L, _P, _L, _I, _I, _P],
    "halva_splice_rows": [_P, _P, _P, _P, _L, _I, _P],
    "halva_token_logp_fwd": [_P, _I, _L, _P, _P, _P, _L, _I, _P],
    "halva_token_logp_bwd": [_P, _I, _L, _P, _P, _P, _P, _L, _I, _P],
    "halva_kl_rows": [_P, _P, _I, _L, _P, _P, _P, _F, _L, _I, _P],
    "halva_phrase_sum_fwd": [_P, _P, _P, _P, _I, _P, _I, _I, _P],
    "halva_phrase_sum_bwd": [_P, _P, _P, _P, _I, _P, _I, _I, _P],
    "halva_probe_layouts": [_P, _I, _P],
    "halva_clock_probe": [_P, _I, _I, _P],
    "halva_sdpa_block_pairs": [_I, _I, _I, _I, _P],
}



class WgradItem(ctypes.Structure):
    """halva_wgrad_item (include/halva_hip.h)"""
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64), ("C", c_void_p), ("M", ctypes.c_int32), ("N", ctypes.c_int32),
                ("rows", c_int64), ("alpha", c_float), ("reserved", ctypes.c_int32)]


_lib = None


class HalvaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library; raises HalvaHipError if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HalvaHipError(
            "libhalva_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C halva_amd/csrc`). There is no CPU fallback for the DPA hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.halva_last_error.restype = c_char_p
    lib.halva_last_error.argtypes = []
    lib.halva_abi_version.restype = c_int
    lib.halva_abi_version.argtypes = []
    if lib.halva_abi_version() != ABI_VERSION:
        raise HalvaHipError("libhalva_hip.so ABI %d != expected %d" % (lib.halva_abi_version(), ABI_VERSION))
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = c_int
        fn.argtypes = args
    lib.halva_sdpa_bwd_ws_bytes.restype = c_int64
    lib.halva_sdpa_bwd_ws_bytes.argtypes = [c_int, c_int, c_int, c_int]
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise HalvaHipError("%s failed (%d): %s" % (name, rc, lib.halva_last_error().decode()))


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a CUDA/HIP tensor (None -> NULL).  CPU tensors are rejected: the kernels are GPU-only."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HalvaHipError("halva_amd kernels need device tensors; got a %s tensor (no CPU fallback)" % t.device)
    return t.data_ptr()
