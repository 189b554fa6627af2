from .language_model.llava_llama import LlavaLlamaForCausalLM, LlavaConfig  # noqa: F401
