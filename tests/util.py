"""shared helpers for the test-suite (fixtures, manifests, seeded weights)."""
import json
import os

import numpy as np
import torch

from lidarseg3d_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def manifest(key):
    with open(os.path.join(GOLDEN, "manifests.json")) as f:
        return {k: tuple(v) for k, v in json.load(f)[key].items()}


def seeded_sd(key, seed, prefix=""):
    """state_dict (torch CPU tensors) for reference module `key` exactly as make_golden.py loaded it."""
    sd = synth.random_state_dict(manifest(key), int(seed))
    return {prefix + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
