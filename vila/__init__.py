"""Reference-shaped import surface of the VILA twin (reference vila/...): thin modules over halva_amd."""
from .model import LlavaLlamaConfig, LlavaLlamaModel  # noqa: F401
