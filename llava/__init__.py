"""`llava` import surface of the reference (pritamqu/HALVA), backed by the MI355X-native implementation in halva_amd."""
from .model import LlavaLlamaForCausalLM  # noqa: F401
