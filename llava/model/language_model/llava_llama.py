"""reference llava/model/language_model/llava_llama.py -> MI355X-native classes (halva_amd/llava_model.py)."""
from halva_amd.llava_model import LlavaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
