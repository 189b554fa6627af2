"""Helpers to read the committed golden fixtures (tests/golden/*.npz|json)."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLD, name))


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def meta_of(npz, key="meta"):
    return json.loads(bytes(npz[key]).decode())


def tensors(npz, prefix):
    """{name-without-prefix: torch tensor} for every key starting with prefix."""
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}
