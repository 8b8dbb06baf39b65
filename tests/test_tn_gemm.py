"""The weight-gradient GEMM out[m, n] = sum_r A[r, m] B[r, n] (gcpnet_tn_gemm) against float64 matmul: the 128 x 160 block kernel,
the big-block kernel (one workgroup owns an output of up to 256 x 320: the (256,32) message GCPs of BASELINE configs[4]) and the
register-staged generic kernel, at row counts that end inside a 32-row chunk; the bias column (`ones`) through _Linear's call."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,M,N", [
    (70001, 256, 284),    # the (256,32) ResGCP weight gradient without its ones column: big-block kernel, all eight waves
    (4133, 200, 300),     # partial last m-tile and n-tile
    (999, 128, 320),      # M fits one block, N does not: four of the eight waves idle
    (33, 256, 160),       # two chunks, the second with one valid row
    (50000, 128, 144),    # the (128,16) shape: the 128 x 160 kernel as before
    (777, 36, 12),        # tiny widths (still multiples of 4): 128 x 160 kernel
])
def test_tn_weight_grad_vs_float64(rows, M, N):
    from gcpnet_amd import ops

    g = torch.Generator().manual_seed(rows + M)
    a = torch.randn(rows, M, generator=g)
    b = torch.randn(rows, N, generator=g)
    want = a.double().t() @ b.double()
    got = ops._tn_weight_grad(a.cuda(), b.cuda())
    again = ops._tn_weight_grad(a.cuda(), b.cuda())
    assert torch.equal(got, again), "the reduction order is fixed: two runs must agree bitwise"
    err = (got.cpu().double() - want).abs().max().item()
    assert err <= 2e-6 * float(want.abs().max()) + 1e-5 * (rows ** 0.5), f"{err:.3e} (scale {float(want.abs().max()):.3e})"


def test_linear_weight_and_bias_gradient_wide():
    """ops.linear's backward: dW = g^T x and db through the `ones` column, at a width that takes the big-block kernel."""
    from gcpnet_amd import ops

    g = torch.Generator().manual_seed(3)
    n, din, dout = 5000, 300, 200
    x = torch.randn(n, din, generator=g)
    w = (torch.randn(dout, din, generator=g) * 0.05)
    b = torch.randn(dout, generator=g)
    lw = torch.randn(n, dout, generator=g)
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ((xd @ wd.t() + bd) * lw.double()).sum().backward()
    xg, wg, bg = x.cuda().requires_grad_(), w.cuda().requires_grad_(), b.cuda().requires_grad_()
    (ops.linear(xg, wg, bg) * lw.cuda()).sum().backward()
    for got, want, name in ((xg.grad, xd.grad, "dx"), (wg.grad, wd.grad, "dW"), (bg.grad, bd.grad, "db")):
        err = (got.cpu().double() - want).abs().max().item()
        assert err <= 1e-5 * float(want.abs().max()) + 1e-5, f"{name}: {err:.3e}"
