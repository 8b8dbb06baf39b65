"""The weight-gradient GEMM out[m, n] = sum_r A[r, m] B[r, n] (gcpnet_tn_gemm) against float64 matmul: the pipelined kernels
(128 x 160 and 256 x 288 blocks: the (128,16) and (256,32) message GCPs of BASELINE configs[1] / [4]), the earlier four-wave DMA
kernel and the register-staged generic kernel behind them, at row counts that end inside a 16-row chunk; the bias column (`ones`)
through _Linear's call."""
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
    (5000, 64, 160),      # two m-tiles: waves laid out 2 along m x 2 along n (n-tiles 0,2,4 / 1,3)
    (5000, 16, 144),      # one m-tile: four wave groups along n, the fifth n-tile back on group 0
    (3000, 96, 100),      # three m-tiles (the fourth wave idles), last n-tile four columns wide
    (2000, 128, 16),      # one n-tile
    (2000, 160, 128),     # M over one block: two blocks along m in the 128 x 160 kernel when the big-block kernel is off
    (97, 300, 53),        # second m-block of the 256 x 288 kernel with two m-tiles; six full 16-row chunks + a one-row chunk; N not a multiple of 4
    (1500, 320, 100),     # m-blocks of eight and two m-tiles
    (15, 64, 40),         # fewer rows than one chunk
    (3001, 132, 600),     # M one column over a block (wide kernel, second block one m-tile), three column blocks
    (2048, 32, 257),      # the gate problem of the (256,32) blocks: narrow kernel, two column blocks
    (50001, 144, 157),    # [ds_pre | dgate]^T [s | norms | frames | 1] of a (128,16) block: the five-wave 160 x 160 kernel, ragged last chunk
    (37, 160, 160),       # the five-wave kernel's full block, two chunks + a ragged one
    (900, 132, 36),       # five m-tiles (the last four rows wide), two n-tiles
])
@pytest.mark.parametrize("form", ["default", "earlier", "fp32", "five-wave"])
def test_tn_weight_grad_vs_float64(rows, M, N, form, monkeypatch):
    """`form`: the switches of gcpnet_tn_gemm that select another kernel or row distribution for the same product."""
    from gcpnet_amd import ops

    if form == "fp32":       # plain fp32 MFMA arithmetic (the four-wave DMA kernel's MODE 0)
        monkeypatch.setenv("GCPNET_TN_FP32", "1")
    elif form == "earlier":  # the kernels the pipelined form took over from (they still serve gathered / activated operands)
        monkeypatch.setenv("GCPNET_TN_PIPE", "0")
    elif form == "five-wave":  # the opt-in 160 x 160 form for 128 < M <= 160, N <= 160 (other shapes: unchanged routing)
        monkeypatch.setenv("GCPNET_TN_MID", "1")

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


@pytest.mark.parametrize("earlier", [False, True], ids=["pipelined", "earlier"])
@pytest.mark.parametrize("rows,M,N", [(40000, 128, 144), (20000, 256, 284)])
def test_bf16_three_term_products_are_fp32_accurate(rows, M, N, earlier, monkeypatch):
    """The default kernels multiply on the bf16 pipe with three-term operand splits (six MFMAs per product block); the fp32-MFMA
    form of the same kernels stays behind GCPNET_TN_FP32.  Both against float64: the split form must not be less accurate than
    fp32 arithmetic itself (bound: twice the fp32 form's own error + one ulp of the result scale)."""
    from gcpnet_amd import ops

    if earlier:
        monkeypatch.setenv("GCPNET_TN_PIPE", "0")
    g = torch.Generator().manual_seed(11)
    a = torch.randn(rows, M, generator=g) * torch.logspace(-3, 3, M)[None, :]   # columns of very different magnitude
    b = torch.randn(rows, N, generator=g)
    want = a.double().t() @ b.double()
    scale = (a.double().abs().t() @ b.double().abs())                          # sum |a b| per output: the natural error scale
    x3 = ops._tn_weight_grad(a.cuda(), b.cuda()).cpu().double()
    monkeypatch.setenv("GCPNET_TN_FP32", "1")
    f32 = ops._tn_weight_grad(a.cuda(), b.cuda()).cpu().double()
    monkeypatch.delenv("GCPNET_TN_FP32")
    e_x3 = ((x3 - want).abs() / scale).max().item()
    e_f32 = ((f32 - want).abs() / scale).max().item()
    assert not torch.equal(x3, f32), "the switch must select a different kernel"
    assert e_x3 <= 2 * e_f32 + 2 ** -22, f"three-term {e_x3:.3e} vs fp32 {e_f32:.3e} (of sum |a b|)"


@pytest.mark.parametrize("splits", [None, 2, 10, 7, 200])
@pytest.mark.parametrize("rows", [5007, 64, 33])
def test_tn_segments_tile_blocked_ones_and_callers_split_count(rows, splits):
    """One gcpnet_tn_gemm problem through the C ABI with everything the chain's weight gradients use at once: a tile-blocked A
    operand, a B operand of three segments (tile-blocked | row-major with a row stride | the ones column -> the bias gradient in
    `out2`), an output narrower than the product (`out_n`), and a split count of the caller's choice (gcpnet_tn_splits only
    recommends; an odd count is served by the earlier kernels).  Against float64."""
    import ctypes as C

    from gcpnet_amd import _lib, ops
    from gcpnet_amd._lib import Operand, TnProblem, check

    lib = _lib.load()
    g = torch.Generator().manual_seed(rows)
    M, N1, N2, ld2 = 128, 96, 12, 20
    a = torch.randn(rows, M, generator=g).cuda()
    b1 = torch.randn(rows, N1, generator=g).cuda()
    b2w = torch.randn(rows, ld2, generator=g).cuda()           # the segment is its first N2 columns
    a_tb, b1_tb = ops.TileBlocked.from_rows(a), ops.TileBlocked.from_rows(b1)
    N = N1 + N2 + 1
    pr = TnProblem()
    pr.rows = rows
    pr.a.n = 1
    pr.a.ptr[0], pr.a.dim[0], pr.a.ld[0], pr.a.tb[0] = a_tb.data_ptr(), M, M, 1
    pr.b.n, pr.b.ones = 2, 1
    pr.b.ptr[0], pr.b.dim[0], pr.b.ld[0], pr.b.tb[0] = b1_tb.data_ptr(), N1, N1, 1
    pr.b.ptr[1], pr.b.dim[1], pr.b.ld[1], pr.b.tb[1] = b2w.data_ptr(), N2, ld2, 0
    out = torch.full((M, N - 1), float("nan"), device="cuda")
    bias = torch.full((M,), float("nan"), device="cuda")
    pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = out.data_ptr(), N - 1, 1, M, N - 1
    pr.out2, pr.out2_n = bias.data_ptr(), N - 1
    pr.splits = splits if splits is not None else lib.gcpnet_tn_splits(rows, M, N)
    part = torch.empty((pr.splits, M, N), device="cuda")
    pr.partial = part.data_ptr()
    check(lib.gcpnet_tn_gemm(1, C.byref(pr), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tn_gemm")
    torch.cuda.synchronize()
    bfull = torch.cat([b1, b2w[:, :N2]], dim=1).cpu().double()
    want = a.cpu().double().t() @ bfull
    want_bias = a.cpu().double().sum(0)
    tol = 2e-6 * float(want.abs().max()) + 1e-5 * rows ** 0.5
    assert (out.cpu().double() - want).abs().max().item() <= tol
    assert (bias.cpu().double() - want_bias).abs().max().item() <= tol
    bad = TnProblem.from_buffer_copy(pr)
    bad.splits = 0
    assert lib.gcpnet_tn_gemm(1, C.byref(bad), None) != 0, "a split count below 1 is refused"


@pytest.mark.parametrize("mid", ["1", "0"], ids=["five-wave", "wide"])  # (GCPNET_TN_MID: the five-wave form is opt-in)
@pytest.mark.parametrize("rows,so,vo", [(5007, 128, 16), (6000, 128, 10), (333, 100, 16), (4000, 96, 16)])
def test_tn_two_gradients_from_one_product(rows, so, vo, mid, monkeypatch):
    """`m_split` (ABI 4): the rows m >= m_split of a product leave into a second matrix / bias vector -- d scalar_out.weight | bias and,
    below it, G | d gate bias of gcp2_wgrad_job_t.gate_lin with their common second operand read once.  A = [tile-blocked so | row-major
    vo padded to a multiple of 4], B = [tile-blocked 128 | row-major 28 | ones]; both halves against float64, with the five-wave
    kernel (GCPNET_TN_MID=1: M = 144 <= 160) and without it (the 256-row kernel or, for M <= 128, the 128-row one either way)."""
    monkeypatch.setenv("GCPNET_TN_MID", mid)
    import ctypes as C

    from gcpnet_amd import _lib, ops
    from gcpnet_amd._lib import TnProblem, check

    lib = _lib.load()
    g = torch.Generator().manual_seed(rows + so)
    vop, N1, N2 = (vo + 3) // 4 * 4, 128, 28
    a1 = torch.randn(rows, so, generator=g).cuda()
    a2 = torch.randn(rows, vop, generator=g).cuda()
    b1 = torch.randn(rows, N1, generator=g).cuda()
    b2 = torch.randn(rows, N2, generator=g).cuda()
    a1_tb, b1_tb = ops.TileBlocked.from_rows(a1), ops.TileBlocked.from_rows(b1)
    K, N = N1 + N2, N1 + N2 + 1
    pr = TnProblem()
    pr.rows = rows
    pr.a.n = 2
    pr.a.ptr[0], pr.a.dim[0], pr.a.ld[0], pr.a.tb[0] = a1_tb.data_ptr(), so, so, 1
    pr.a.ptr[1], pr.a.dim[1], pr.a.ld[1], pr.a.tb[1] = a2.data_ptr(), vop, vop, 0
    pr.b.n, pr.b.ones = 2, 1
    pr.b.ptr[0], pr.b.dim[0], pr.b.ld[0], pr.b.tb[0] = b1_tb.data_ptr(), N1, N1, 1
    pr.b.ptr[1], pr.b.dim[1], pr.b.ld[1], pr.b.tb[1] = b2.data_ptr(), N2, N2, 0
    nan = float("nan")
    out, bias = torch.full((so, K), nan, device="cuda"), torch.full((so,), nan, device="cuda")
    out_b, bias_b = torch.full((vo, K), nan, device="cuda"), torch.full((vo,), nan, device="cuda")
    pr.out, pr.out_sm, pr.out_sn, pr.out_m, pr.out_n = out.data_ptr(), K, 1, so + vo, K
    pr.out2, pr.out2_n = bias.data_ptr(), K
    pr.m_split, pr.out_b, pr.out_b_sm, pr.out2_b = so, out_b.data_ptr(), K, bias_b.data_ptr()
    M = so + vop
    pr.splits = lib.gcpnet_tn_splits(rows, M, N)
    part = torch.empty((pr.splits, M, N), device="cuda")
    pr.partial = part.data_ptr()
    check(lib.gcpnet_tn_gemm(1, C.byref(pr), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tn_gemm")
    torch.cuda.synchronize()
    bfull = torch.cat([b1, b2], dim=1).cpu().double()
    tol = 2e-6 * float((a1.cpu().double().t() @ bfull).abs().max()) + 1e-5 * rows ** 0.5
    for got, want, name in ((out, a1.cpu().double().t() @ bfull, "out"), (bias, a1.cpu().double().sum(0), "out2"),
                            (out_b, a2[:, :vo].cpu().double().t() @ bfull, "out_b"), (bias_b, a2[:, :vo].cpu().double().sum(0), "out2_b")):
        assert (got.cpu().double() - want).abs().max().item() <= tol, name
    bad = TnProblem.from_buffer_copy(pr)
    bad.out_b = None
    assert lib.gcpnet_tn_gemm(1, C.byref(bad), None) != 0, "m_split without a second destination is refused"


@pytest.mark.parametrize("pattern", ["growing", "shrinking", "zeros-first", "spikes-2^12", "spikes-2^25", "spikes-2^25-bf16"])
def test_wide_form_running_exponents_follow_the_data(pattern, monkeypatch):
    """The 256 x 288 form multiplies two-term fp16 operands under a RUNNING power-of-two exponent per 32-column fragment (tn_gemm.hip,
    GCP_TN_F16X2): set by the first non-zero 16-row chunk, lowered -- with the accumulator tiles rescaled -- when a later chunk would
    overflow.  Row magnitudes that grow by 2^60 over the rows (a rescale every few chunks), shrink by as much (the exponent stays: small
    rows lose relative precision only), start with all-zero chunks (unset -> set), or spike in single rows of single columns: each
    against float64, error relative to sum |a b| per output (the bound of gcp_f16x2.h; 2e-6 leaves room for the fp32 accumulation of
    40 000 rows).  The limit of a shared exponent, stated and held here: a spike of 2^25 in ONE element costs the other elements of its
    16-row x 32-column chunk their low bits (the exponent comes back up behind it) -- invisible in a sum over 40 000 rows UNLESS the
    same row also carries a spike in the other operand, which multiplies exactly those coarsened elements: the `spikes-2^25` case puts
    both spikes in the same rows and the affected outputs keep ~8 bits (bound 2^-7 of sum |a b|); spikes of 2^12 stay at round-off.
    GCPNET_TN_F16=0 selects the six-product bf16 form, which has fp32's exponent range and no such coupling at all."""
    from gcpnet_amd import ops

    rows, M, N = 40000, 256, 284
    g = torch.Generator().manual_seed(7)
    a = torch.randn(rows, M, generator=g)
    b = torch.randn(rows, N, generator=g)
    t = torch.linspace(0, 1, rows)[:, None]
    bound = 2e-6
    if pattern == "growing":
        a = a * torch.exp2(60 * t - 30)
        b = b * torch.exp2(40 * t - 20)
    elif pattern == "shrinking":
        a = a * torch.exp2(30 - 60 * t)
        b = b * torch.exp2(-20 * t)
    elif pattern == "zeros-first":
        a[:4096] = 0
        b[:1000] = 0
        a[:, 40:72] = 0          # a fragment that never sees data
    else:
        k = 12 if "2^12" in pattern else 25
        idx = torch.randint(0, rows, (200,), generator=g)
        a[idx, torch.randint(0, M, (200,), generator=g)] *= 2.0 ** k
        b[idx, torch.randint(0, N, (200,), generator=g)] *= 2.0 ** (k - 7)
        if pattern.endswith("bf16"):
            monkeypatch.setenv("GCPNET_TN_F16", "0")
        elif k == 25:
            bound = 2.0 ** -7
    want = a.double().t() @ b.double()
    scale = a.double().abs().t() @ b.double().abs()
    got = ops._tn_weight_grad(a.cuda(), b.cuda()).cpu().double()
    assert torch.isfinite(got).all()
    ok = scale > 0
    err = ((got - want).abs()[ok] / scale[ok]).max().item()
    assert err <= bound, f"{pattern}: {err:.3e} of sum |a b|"
    assert torch.equal(got[~ok], torch.zeros_like(got[~ok]))
