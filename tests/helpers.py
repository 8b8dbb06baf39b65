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
    rep = os.environ.get("GCPNET_PARITY_REPORT")  # (tools/parity_table.py: what every comparison of a run actually achieved)
    if rep and err.numel():
        unit = (err / (1e-5 + 1e-5 * b.double().abs())).max().item()  # in units of the north_star tolerance (1e-5 abs + 1e-5 rel)
        with open(rep, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0]}\t{tuple(a.shape)}\t{err.max().item():.3e}\t"
                    f"{b.abs().max().item():.3e}\t{atol:g}\t{rtol:g}\t{unit:.3f}\n")
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


def as_accurate(got, cpu32, cpu64, name="", factor=4.0, floor=2e-3, abs_floor=0.0):
    """`got` (the HIP path, fp32) is as close to the float64 result as the reference's own fp32 CPU arithmetic is, up to
    `factor` (L2 over the tensor), with a floor of `floor` x the tensor's norm (a handful of flips more or fewer on one side
    moves a small weight gradient by a few 1e-4 of its norm; the smooth-activation variants of the same tests, which run the
    same kernels, are held to 1e-4 element-wise, and the small ReLU fixtures to 1e-4 .. 1e-5).

    Why not element-wise at 1e-5: with ReLU a pre-activation within round-off of zero takes a different branch in ANY two fp32
    evaluations (different summation orders); each flip moves an isolated row of a gradient by O(1/s).  Both fp32 results then
    differ from the exact one by the same kind of sparse O(1e-3) perturbations; what can be demanded is that the HIP path is not
    worse than the CPU path the reference itself runs.  Smooth activations are checked element-wise by the callers."""
    got, cpu32, cpu64 = (torch.as_tensor(t).double() for t in (got, cpu32, cpu64))
    assert got.shape == cpu64.shape, (name, got.shape, cpu64.shape)
    ref = float(cpu64.norm())
    e_got, e_cpu = float((got - cpu64).norm()), float((cpu32 - cpu64).norm())
    bound = max(factor * e_cpu, floor * max(ref, 1e-30), abs_floor)  # abs_floor: for tensors whose gradient is ~0 overall
    assert e_got <= bound, f"{name}: |hip - f64| = {e_got:.3e} > max({factor} x |cpu32 - f64| ({e_cpu:.3e}), {floor} x |f64| ({ref:.3e}))"


def poison_allocations():
    """Debugging aid for the GPU sweeps (SWEEP_POISON=1): every fresh float32 device buffer from torch.empty / empty_like starts
    as NaN, so a kernel that reads memory nobody has written shows it deterministically instead of depending on what the caching
    allocator hands back.  Returns a function that restores the originals."""
    import torch

    orig_empty, orig_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.dtype == torch.float32 and t.numel():
            t.fill_(float("nan"))
        return t

    def empty(*a, **k):
        return fill(orig_empty(*a, **k))

    def empty_like(*a, **k):
        return fill(orig_like(*a, **k))

    torch.empty, torch.empty_like = empty, empty_like

    def restore():
        torch.empty, torch.empty_like = orig_empty, orig_like

    return restore


def bit_identical(runs):
    """`runs`: list of dicts name -> tensor from repeated executions of the same step; returns the names that differ bitwise
    from the first run (NaN == NaN counts as equal)."""
    import torch

    bad = []
    for r in runs[1:]:
        for k, t in r.items():
            a, b = runs[0][k], t
            if a.shape != b.shape or not torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                                                     b.view(torch.int32) if b.dtype == torch.float32 else b):
                bad.append(k)
    return sorted(set(bad))
