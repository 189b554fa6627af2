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


def to_tensor(a):
    """numpy -> torch; uint16 arrays hold raw bf16 bits (weights stored bf16-exact) and come back as fp32."""
    if a.dtype == np.uint16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).float()
    return torch.from_numpy(a)


def tensors(npz, prefix):
    """{name-without-prefix: torch tensor} for every key starting with prefix."""
    return {k[len(prefix):]: to_tensor(npz[k]) for k in npz.files if k.startswith(prefix)}
