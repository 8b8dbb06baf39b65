"""fp32 products on the bf16 matrix pipe (gcpnet_amd/csrc/gcp_bf16x3.h): three-term split, six products.

CPU: the split itself, restated in numpy (x = h + m + l exactly; the three dropped products are below 3 * 2^-24 |a b|).
GPU: the chain kernels with the bf16 form switched on and off in one process (gcpnet_debug_set_fp32_mfma) -- both forms
are the same fp32 computation up to summation order, and neither is further from a float64 evaluation than the other."""
import numpy as np
import pytest
import torch


def _trunc_bf16(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _split3(x):
    h = _trunc_bf16(x)
    r = (x - h).astype(np.float32)
    m = _trunc_bf16(r)
    s = (r - m).astype(np.float32)
    return h, m, _trunc_bf16(s), s


def test_three_term_split_is_exact():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32),
                        (rng.standard_normal(50000) * 1e-12).astype(np.float32),
                        (rng.standard_normal(50000) * 1e12).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.17549435e-38], dtype=np.float32)])
    h, m, l, s = _split3(x)
    assert np.array_equal(l, s), "the third term holds the whole second residual (24 significant bits = 3 x 8)"
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)).astype(np.float32), x)
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -7) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -15)


def test_six_products_match_fp32_roundoff():
    rng = np.random.default_rng(1)
    K = 128
    a = rng.standard_normal((64, K)).astype(np.float32)
    b = (rng.standard_normal((K, 64)) * np.where(rng.random((K, 64)) < 0.15, 1e-3, 1.0)).astype(np.float32)
    ah, am, al, _ = _split3(a)
    bh, bm, bl, _ = _split3(b)
    f = lambda p, q: p.astype(np.float64) @ q.astype(np.float64)
    kept = f(al, bh) + f(ah, bl) + f(am, bm) + f(am, bh) + f(ah, bm) + f(ah, bh)
    exact = f(a, b)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(kept - exact) / scale) <= 3 * 2.0 ** -24  # the dropped ml, lm, ll


@pytest.mark.gpu
@pytest.mark.parametrize("dims,act", [((128, 16), "relu"), ((128, 16), "silu"), ((64, 16), "relu"), ((100, 16), "silu"),
                                      ((256, 32), "silu")])  # (256,32): the workgroup kernels' bf16 forms (8 waves)
def test_chain_kernels_bf16_and_fp32_forms_agree(dims, act):
    import gcpnet_amd as G
    from gcpnet_amd import _lib, ops

    lib = _lib.load()
    torch.manual_seed(3)
    rows, nblk = 3000, 7
    S, V = dims
    mods = [G.GCP2((S, V), (S, V), nonlinearities=(act, None), bottleneck=4).cuda() for _ in range(nblk)]
    specs = [m.make_spec([None], [None], residual=True) for m in mods]
    g = torch.Generator(device="cuda").manual_seed(0)
    s0 = torch.randn(rows, S, device="cuda", generator=g)
    v0 = torch.randn(rows, V, 3, device="cuda", generator=g)
    fr = torch.randn(rows, 3, 3, device="cuda", generator=g)
    ds = torch.randn(rows, S, device="cuda", generator=g)
    dv = torch.randn(rows, V, 3, device="cuda", generator=g)

    def run(fp32_mfma, wg_forward, dtype=torch.float32):
        prev = lib.gcpnet_debug_set_fp32_mfma(int(fp32_mfma))
        saved, saved_skip = ops.PREFER_WAVE_CHAIN_FORWARD, ops.CHAIN_SKIP_S_PRE
        ops.PREFER_WAVE_CHAIN_FORWARD = not wg_forward
        if fp32_mfma:  # (the fp32 form of the chain backward reads s_pre: it has no sign-mask instantiation)
            ops.CHAIN_SKIP_S_PRE = False
        try:
            s = s0.clone().requires_grad_()
            v = v0.clone().requires_grad_()
            ws = [m._weights() for m in mods]
            for m in mods:
                m.zero_grad(set_to_none=True)
            o_s, o_v = ops.gcp2_chain(specs, s, v, fr, ws)
            torch.autograd.backward([o_s, o_v], [ds, dv])
            torch.cuda.synchronize()
            out = dict(o_s=o_s.detach(), o_v=o_v.detach(), d_s=s.grad, d_v=v.grad)
            out.update({f"w{k}.{n}": q.grad for k, m in enumerate(mods) for n, q in m.named_parameters()})
            return {k: t.clone() for k, t in out.items()}
        finally:
            ops.PREFER_WAVE_CHAIN_FORWARD, ops.CHAIN_SKIP_S_PRE = saved, saved_skip
            lib.gcpnet_debug_set_fp32_mfma(prev)

    # so <= 128: wave-per-tile forward + chain backward; wider: workgroup forward + workgroup backward block by block
    fp32 = run(True, wg_forward=False)       # v_mfma_f32_32x32x2_f32 everywhere
    bf16 = run(False, wg_forward=False)      # the same kernels, bf16 x 6
    wg = run(False, wg_forward=True)         # the default route (so <= 128: workgroup forward with fp32 MFMA + chain backward)
    assert any(not torch.equal(fp32[k], bf16[k]) for k in fp32), "the switch did not change the arithmetic"
    for k in fp32:
        scale = max(float(fp32[k].abs().max()), 1e-6)
        tol = (2e-5 if act == "silu" else 1e-3) * scale  # (relu: a pre-activation within round-off of 0 may flip, helpers.as_accurate)
        err = float((fp32[k] - bf16[k]).abs().max())
        assert err <= tol, f"{k}: bf16 x 6 and fp32 MFMA forms differ by {err:.3e} (scale {scale:.3e})"
        err = float((wg[k] - bf16[k]).abs().max())
        assert err <= tol, f"{k}: workgroup / wave forward routes differ by {err:.3e} (scale {scale:.3e})"


@pytest.mark.gpu
def test_fp32_form_refuses_a_chain_that_kept_only_sign_masks():
    """A chain forward with piecewise-linear activations keeps sign masks INSTEAD of s_pre (ops.CHAIN_SKIP_S_PRE); only the sign-mask
    instantiations of the chain backward (bf16 form) can run it.  With the fp32-MFMA A/B form forced the library must refuse the
    launch -- never read the s_pre that is not there."""
    import gcpnet_amd as G
    from gcpnet_amd import _lib, ops

    if not (ops.CHAIN_SKIP_S_PRE and ops.CHAIN_SIGN_MASKS and ops.GATE_GRADS_FROM_INPUTS):
        pytest.skip("the forward stores s_pre under these switches")
    lib = _lib.load()
    torch.manual_seed(0)
    rows = 500
    mods = [G.GCP2((128, 16), (128, 16), nonlinearities=("relu", None), bottleneck=4).cuda() for _ in range(3)]
    specs = [m.make_spec([None], [None], residual=True) for m in mods]
    s = torch.randn(rows, 128, device="cuda").requires_grad_()
    v = torch.randn(rows, 16, 3, device="cuda").requires_grad_()
    fr = torch.randn(rows, 3, 3, device="cuda")
    saved = ops.PREFER_WAVE_CHAIN_FORWARD
    ops.PREFER_WAVE_CHAIN_FORWARD = True
    try:
        o_s, o_v = ops.gcp2_chain(specs, s, v, fr, [m._weights() for m in mods])
        prev = lib.gcpnet_debug_set_fp32_mfma(1)
        try:
            with pytest.raises(_lib.GcpnetHipError):
                torch.autograd.backward([o_s, o_v], [torch.ones_like(o_s), torch.ones_like(o_v)])
        finally:
            lib.gcpnet_debug_set_fp32_mfma(prev)
        torch.cuda.synchronize()
    finally:
        ops.PREFER_WAVE_CHAIN_FORWARD = saved
