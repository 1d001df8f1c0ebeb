"""Shared helpers for tests (fixture loading, seeded synthetic weights)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def tt(a):
    return torch.from_numpy(np.asarray(a))


def sub(npz, prefix):
    """{'a.b.c': arr} with prefix 'a.' -> {'b.c': tensor}."""
    return {k[len(prefix):]: tt(npz[k]) for k in npz.files if k.startswith(prefix)}


def enc_layers(flat, L):
    """'layers.N.self_attn.q_proj.weight' state dict -> list of per-layer dicts in oracle naming."""
    out = []
    for i in range(L):
        p = f"layers.{i}."
        d = {}
        for k, v in flat.items():
            if k.startswith(p):
                k2 = k[len(p):].replace("self_attn.", "").replace("mlp.", "")
                d[k2] = v
        out.append(d)
    return out


def cfg_of(npz, key):
    return json.loads(str(npz[key]))
