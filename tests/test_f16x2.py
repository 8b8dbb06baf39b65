"""fp32 products on the fp16 matrix pipe (gcpnet_amd/csrc/gcp_f16x2.h): two fp16 terms per operand, three products, a power-of-two
scale per activation ROW and a fixed one for the weights.  CPU: the scheme restated in numpy -- the bound the header states
(3 * 2^-22 of sum |a b| for operands within the scales' comfortable range; the documented floor below it), at the magnitudes the chain
kernels meet: O(1) activations, gradients of 1e-6, weights from 1e-4 to 10.  The GPU side of the same arithmetic is tests/test_bf16x3.py
(the kernels' default form against their fp32-MFMA form) and every parity test of the suite."""
import numpy as np
import pytest

WEXP = 6  # GCP_F16_WEXP


def _row_exp(m):
    """gcp_f16_row_exp: 14 - floor(log2 m), clamped to [-60, 60]; zero / subnormal rows: 0."""
    eb = ((m.view(np.uint32) >> 23) & 0xFF).astype(np.int64)
    return np.where(eb == 0, 0, np.clip(14 + 127 - eb, -60, 60))


def _split2(x):
    """x (float32, already scaled) -> (h, l) as float16, round to nearest; the residual x - h is exact in fp32."""
    h = x.astype(np.float16)
    r = (x - h.astype(np.float32)).astype(np.float32)
    assert np.array_equal(r.astype(np.float64), x.astype(np.float64) - h.astype(np.float64)), "the residual is exact in fp32"
    return h, r.astype(np.float16)


def _three_products(a, w):
    """out[i, n] = sum_k a[i, k] w[n, k] the kernels' way (float64 stands in for the exact fp16 products and the fp32 accumulate)."""
    pa = _row_exp(np.abs(a).max(axis=1))
    sa = np.ldexp(np.float32(1), pa.astype(np.int32)).astype(np.float32)
    ah, al = _split2(a * sa[:, None])
    ws = np.clip(w * np.float32(2 ** WEXP), -65504, 65504).astype(np.float32)
    wh, wl = _split2(ws)
    f = lambda p, q: p.astype(np.float64) @ q.astype(np.float64).T
    kept = f(al, wh) + f(ah, wl) + f(ah, wh)
    return kept * np.ldexp(1.0, -(pa + WEXP))[:, None]


@pytest.mark.parametrize("a_scale", [1.0, 1e-6, 3e4, 1e-13])
@pytest.mark.parametrize("w_scale", [0.1, 1e-3, 8.0])
def test_three_products_are_within_the_stated_bound(a_scale, w_scale):
    rng = np.random.default_rng(3)
    K = 160
    a = (rng.standard_normal((256, K)) * a_scale * np.where(rng.random((256, K)) < 0.3, 1e-3, 1.0)).astype(np.float32)
    a[:, ::7] = 0.0       # relu-dead columns
    a[5] = 0.0            # an all-zero row (scale clamped, nothing to multiply)
    w = (rng.standard_normal((128, K)) * w_scale).astype(np.float32)
    got = _three_products(a, w)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    ok = scale > 0
    assert np.max(np.abs(got - exact)[ok] / scale[ok]) <= 3 * 2.0 ** -22
    assert np.all(got[5] == 0)


def test_wide_dynamic_range_inside_a_row_degrades_gracefully():
    """Elements far below their row's maximum lose RELATIVE precision (their fp16 terms run into the subnormal range) but the error
    stays below 2^-38 of (row maximum x |w|) per term -- the floor the header documents; small weights likewise below 2^-31 absolute."""
    rng = np.random.default_rng(4)
    K = 160
    a = (rng.standard_normal((64, K)) * np.exp(rng.uniform(-25, 0, (64, K)))).astype(np.float32)  # eleven decades inside a row
    w = (rng.standard_normal((128, K)) * np.exp(rng.uniform(-14, 0, (128, K)))).astype(np.float32)
    got = _three_products(a, w)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    bound = (3 * 2.0 ** -22 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T)
             + K * 2.0 ** -38 * np.abs(a).max(axis=1)[:, None] * np.abs(w).max(axis=1)[None, :]
             + 2.0 ** -31 * np.abs(a).sum(axis=1)[:, None])
    assert np.all(np.abs(got - exact) <= bound)


def test_rows_below_the_scale_clamp_keep_an_absolute_floor():
    """Rows whose largest magnitude is below 2^-46 are scaled by the clamp's 2^60 only: relative precision goes, the absolute error per
    term stays below 2^-25 (half an fp16 subnormal step) / 2^60 x |w| -- nothing a sum that also holds a bias or a residual state can see."""
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((32, 160)) * 1e-20).astype(np.float32)
    w = (rng.standard_normal((128, 160)) * 0.1).astype(np.float32)
    got = _three_products(a, w)
    exact = a.astype(np.float64) @ w.astype(np.float64).T
    floor = 2.0 ** -85 * np.abs(w).astype(np.float64).sum(axis=1)[None, :]
    assert np.all(np.abs(got - exact) <= floor + 3 * 2.0 ** -22 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T))


def test_weight_terms_saturate_instead_of_overflowing():
    w = np.array([[2000.0, -5000.0, 1.0]], dtype=np.float32)   # 2^6 w beyond 65504
    ws = np.clip(w * np.float32(2 ** WEXP), -65504, 65504).astype(np.float32)
    h, l = _split2(ws)
    assert np.all(np.isfinite(h.astype(np.float32))) and np.all(np.isfinite(l.astype(np.float32)))


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(128, 16), (256, 32)], ids=["wave-per-tile", "workgroup"])
def test_row_scales_follow_rows_of_very_different_magnitude(dims):
    """The kernels' side of the per-row scales: a 3-block residual chain whose input rows span seven decades (1e-6 ... 10) and whose
    incoming gradient rows span ten more (1e-8 ... 100).  Two-term fp16 form and fp32-MFMA form of the same kernels (gcpnet_debug_set_fp32_mfma),
    both against the oracle in float64, ROW by row: the fp16 form may be at most 16 x as far from float64 as the fp32 form is (its
    round-off bound is 3 * 2^-22 against 2^-24 per product: 12 x) plus 1e-5 of the row's own scale -- a scale shared between rows, or chosen from
    the wrong row, would drown the small rows (errors of order 1).  (Input rows are kept below 10: at 1e4 the pre-activations in front of
    the gate's sigmoid are of that size and EITHER form's round-off shows as 1e-4 ... 1e-3 of the row -- the fp32 form is the worse one
    on as many rows as the fp16 form, `tools`-level check at the end of round 6 -- which says nothing about scales.)  silu: no
    activation kinks to flip."""
    import torch

    import gcpnet_amd as G
    from gcpnet_amd import _lib, ops
    from oracle import gcp_oracle as O

    lib = _lib.load()
    torch.manual_seed(2)
    rows, nblk = 2500, 3
    S, V = dims
    mods = [G.GCP2((S, V), (S, V), nonlinearities=("silu", None), bottleneck=4).cuda() for _ in range(nblk)]
    specs = [m.make_spec([None], [None], residual=True) for m in mods]
    g = torch.Generator(device="cuda").manual_seed(0)
    rs = torch.pow(10.0, torch.rand(rows, 1, device="cuda", generator=g) * 7 - 6)
    gs = torch.pow(10.0, torch.rand(rows, 1, device="cuda", generator=g) * 10 - 8)
    s0 = torch.randn(rows, S, device="cuda", generator=g) * rs
    v0 = torch.randn(rows, V, 3, device="cuda", generator=g) * rs[:, :, None]
    fr = torch.randn(rows, 3, 3, device="cuda", generator=g)
    ds = torch.randn(rows, S, device="cuda", generator=g) * gs
    dv = torch.randn(rows, V, 3, device="cuda", generator=g) * gs[:, :, None]

    def run(fp32_mfma):
        prev = lib.gcpnet_debug_set_fp32_mfma(int(fp32_mfma))
        saved = ops.CHAIN_SKIP_S_PRE
        ops.CHAIN_SKIP_S_PRE = False
        try:
            s = s0.clone().requires_grad_()
            v = v0.clone().requires_grad_()
            o_s, o_v = ops.gcp2_chain(specs, s, v, fr, [m._weights() for m in mods])
            torch.autograd.backward([o_s, o_v], [ds, dv])
            torch.cuda.synchronize()
            return dict(o_s=o_s.detach().clone(), o_v=o_v.detach().flatten(1).clone(), d_s=s.grad.clone(), d_v=v.grad.flatten(1).clone())
        finally:
            ops.CHAIN_SKIP_S_PRE = saved
            lib.gcpnet_debug_set_fp32_mfma(prev)

    f32 = {k: t.cpu().double() for k, t in run(True).items()}
    f16 = {k: t.cpu().double() for k, t in run(False).items()}
    assert any(not torch.equal(f32[k], f16[k]) for k in f32), "the switch did not change the arithmetic"
    # float64 on the CPU: x <- x + GCP2(x), three times (components/gcpnet.py:921-924)
    ei = torch.stack((torch.arange(rows), torch.arange(rows)))
    s = s0.cpu().double().requires_grad_()
    v = v0.cpu().double().requires_grad_()
    xs, xv = s, v
    for m in mods:
        P = {k: t.detach().cpu().double() for k, t in m.state_dict().items()}
        ys, yv = O.gcp2(P, "", xs, xv, ei, fr.cpu().double(), nonlinearities=("silu", None), vector_output_dim=V)
        xs, xv = xs + ys, xv + yv
    torch.autograd.backward([xs, xv], [ds.cpu().double(), dv.cpu().double()])
    ref = dict(o_s=xs.detach(), o_v=xv.detach().flatten(1), d_s=s.grad, d_v=v.grad.flatten(1))
    for k in ref:
        pair = {"o_s": "o_v", "o_v": "o_s", "d_s": "d_v", "d_v": "d_s"}[k]  # (a row's scalar and vector parts mix: one scale per row)
        row_scale = torch.maximum(ref[k].abs().amax(1), ref[pair].abs().amax(1)).clamp_min(1e-300)
        e32 = (f32[k] - ref[k]).abs().amax(1) / row_scale
        e16 = (f16[k] - ref[k]).abs().amax(1) / row_scale
        worst = (e16 - 16 * e32).max().item()
        assert worst <= 1e-5, f"{k}: a row is {worst:.3e} (of its own scale) further from float64 than 16 x the fp32-MFMA form's distance"


def test_check_weight_range_reports_weights_the_fp16_images_cannot_hold():
    import torch

    import gcpnet_amd as G
    from gcpnet_amd import _lib

    m = G.GCP2((16, 4), (16, 4), nonlinearities=("relu", None), bottleneck=4)
    assert 0 < G.check_weight_range(m) < 10
    with torch.no_grad():
        m.scalar_out.weight[0, 0] = 2000.0
    with pytest.raises(_lib.GcpnetHipError):
        G.check_weight_range(m)
