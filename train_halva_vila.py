"""Entry point of the VILA twin (reference train_halva_vila.py): `deepspeed train_halva_vila.py <flags of
src_vila/halva_vila_13b.sh>`.  The reference patches transformers' image normalisation so single-channel images
survive; the same patch is applied here."""
import os

os.environ.setdefault("WANDB_PROJECT", "HALVA")

from unittest import mock  # noqa: E402

from vila.train.train_halva import train  # noqa: E402
from vila.train.transformer_normalize_monkey_patch import patched_normalize  # noqa: E402

if __name__ == "__main__":
    with mock.patch("transformers.image_transforms.normalize", new=patched_normalize):
        train()
