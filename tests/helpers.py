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


def ssim_value_torch(x, y, data_range=1.0, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """The same quantity spelled with torch ops (the checker of the HIP kernel in tests/): per-image SSIM averaged over
    channels (piq.ssim defaults, reduction='none')."""
    with torch.no_grad():
        x, y = x / data_range, y / data_range
        f = max(1, round(min(x.shape[-2:]) / 256))
        if f > 1:
            x, y = torch.nn.functional.avg_pool2d(x, f), torch.nn.functional.avg_pool2d(y, f)
        c = torch.arange(kernel_size, dtype=x.dtype, device=x.device) - (kernel_size - 1) / 2.0
        g1 = torch.exp(-(c ** 2) / (2 * sigma ** 2))
        k = (g1[:, None] * g1[None, :])
        k = (k / k.sum())[None, None].repeat(x.shape[1], 1, 1, 1)
        C = x.shape[1]
        mu_x, mu_y = torch.nn.functional.conv2d(x, k, groups=C), torch.nn.functional.conv2d(y, k, groups=C)
        sxx = torch.nn.functional.conv2d(x * x, k, groups=C) - mu_x ** 2
        syy = torch.nn.functional.conv2d(y * y, k, groups=C) - mu_y ** 2
        sxy = torch.nn.functional.conv2d(x * y, k, groups=C) - mu_x * mu_y
        c1, c2 = k1 ** 2, k2 ** 2
        cs = (2 * sxy + c2) / (sxx + syy + c2)
        ss = (2 * mu_x * mu_y + c1) / (mu_x ** 2 + mu_y ** 2 + c1) * cs
        return ss.mean(dim=(-1, -2)).mean(dim=1)
