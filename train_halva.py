"""Entry point named by the reference launch scripts (`deepspeed train_halva.py <flags>`, src/hallava_7b.sh:30).
Same two steps as the reference's file: install the attention seam, then run llava.train.train_halva.train()."""
import os

os.environ.setdefault("WANDB_PROJECT", "HALVA")      # kept for parity; metrics are printed as JSON lines (no wandb offline)

from llava.train.llama_flash_attn_monkey_patch import replace_llama_attn_with_flash_attn

replace_llama_attn_with_flash_attn()                 # = "libhalva_hip.so is present and ABI-compatible" on MI355X
from llava.train.train_halva import train

if __name__ == "__main__":
    train()
