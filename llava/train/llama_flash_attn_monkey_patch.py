"""The reference's one optimisation seam (llava/train/llama_flash_attn_monkey_patch.py:105-115) replaced
LlamaAttention.forward with a CUDA flash-attn call.  Here attention already IS the hand-written gfx950 kernel
(halva_sdpa_causal_fwd/bwd behind halva_amd.kernels.attention), so the entry point train_halva.py calls only has to make
sure the HIP library is present - loudly, with no CUDA capability query and no fallback."""


def replace_llama_attn_with_flash_attn():
    from halva_amd import hip
    hip.load()          # raises HalvaHipError if libhalva_hip.so is missing or its ABI does not match
