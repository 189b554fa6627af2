"""reference vila/model/__init__.py: LlavaLlamaModel / LlavaLlamaConfig (the only architecture setup_model accepts,
vila/train/train_halva.py:296-310)."""
from halva_amd.vila_model import (DownSampleBlock, LlamaForCausalLM, MultimodalProjector,  # noqa: F401
                                  VilaConfig as LlavaLlamaConfig, VilaLlavaLlamaModel as LlavaLlamaModel)
from halva_amd.siglip import SiglipVisionTower  # noqa: F401
