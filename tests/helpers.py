"""Shared helpers for the test-suite (fixture loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def t(a):
    return torch.from_numpy(np.array(a, copy=True))


def sub(d, prefix):
    """{'sd0/foo': x} -> {'foo': tensor}"""
    p = prefix + "/"
    return {k[len(p):]: t(v) for k, v in d.items() if k.startswith(p)}


def meta_of(d):
    return json.loads(bytes(d["meta"]).decode())


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the 'relative fp32 tolerance' of the north star, scale-normalised."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    denom = b.abs().max().clamp_min(1e-30)
    return float((a - b).abs().max() / denom)


def rel_l2(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
