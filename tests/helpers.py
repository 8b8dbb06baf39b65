"""Shared test helpers: fixture loading and seeded synthetic graphs."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.p, self.i, self.o, self.g, self.m = {}, {}, {}, {}, {}
        for k in z.files:
            tag, key = k.split("/", 1)
            arr = z[k]
            getattr(self, tag)[key] = torch.from_numpy(arr) if tag != "m" else arr


def fixture_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def close(a, b, atol=1e-6, rtol=1e-5):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs()
    tol = atol + rtol * b.double().abs()
    worst = (err - tol).max().item() if err.numel() else -1.0
    assert worst <= 0, f"max abs err {err.max().item():.3e} (scale {b.abs().max().item():.3e})"


def rand_graph(n, e, seed, sort_by_col=False):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(0, n, (e,), generator=g)
    col = torch.randint(0, n, (e,), generator=g)
    col = torch.where(col == row, (col + 1) % n, col)
    if sort_by_col:
        order = torch.argsort(col, stable=True)
        row, col = row[order], col[order]
    x = torch.randn(n, 3, generator=g)
    return torch.stack((row, col)), x
