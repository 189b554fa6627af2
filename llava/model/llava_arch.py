"""reference llava/model/llava_arch.py -> MI355X-native mixins (halva_amd/llava_model.py)."""
from halva_amd.llava_model import LlavaMetaForCausalLM, LlavaMetaModel  # noqa: F401
