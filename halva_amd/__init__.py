"""halva_amd - MI355X-native implementation of the HALVA DPA training step.

Only what the hot path needs lives here: `csrc/` (HIP kernels + the C ABI of include/halva_hip.h), the ctypes
binding (`hip`), autograd wrappers (`kernels`), the host-side splice plan (`splice`), the Llama/CLIP/LoRA modules
that call the kernels (`llama`, `clip`), the DPA loss + step engine (`dpa`) and data-parallel gradient exchange (`dp`).
"""
__version__ = "0.1.0"
