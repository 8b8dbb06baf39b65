"""Sign masks of s_pre (include/gcpnet_hip.h, gcp2_chain_item_t.s_sign): the chain backward kernel reading one bit per element --
where s_pre is positive -- instead of s_pre itself must give bit-identical results whenever the activations are piecewise linear, and
the mask the forward writes must be exactly (s_pre > 0) in the tile-blocked register order."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_chain_split import G, _chain, _flatten  # noqa: E402,F401


@pytest.mark.parametrize("dims,act", [((128, 16), "relu"), ((64, 8), "relu"), ((100, 16), "relu"), ((128, 16), "leakyrelu"), ((64, 16), "relu")],
                         ids=["128x16", "64x8", "100x16-padded", "128x16-leaky", "64x16"])
@pytest.mark.parametrize("rows", [1013, 64, 7])
def test_backward_from_sign_masks_is_bit_identical(G, dims, act, rows):
    from gcpnet_amd import _lib

    ops, specs, ins, outs, frames, ws_, packs, n, g, _keep = _chain(G, dims, rows, act)
    lib = _lib.load()
    words = int(lib.gcpnet_tb_sign_words(rows, dims[0]))
    assert words == ((rows + 31) // 32) * ((dims[0] + 31) // 32 // 2) * 64 > 0
    for k in range(n):  # the mask itself: bit 16 t + r of lane (e, hi) of a tile = (s_pre > 0) of register r of accumulator tile t
        sp = outs[k][2]
        assert isinstance(sp, ops.TileBlocked) and sp.sign is not None
        off = (sp.sign - sp._owner.data_ptr()) // 4
        got = sp._owner[off:off + words].view(torch.int32).cpu()
        wp = (dims[0] + 31) // 32 * 32
        tiles = (rows + 31) // 32
        tb = sp.data.view(tiles, wp // 32, 4, 2, 32, 4).cpu()  # [tile, t, q, hi, e, i]: register r = 4 q + i
        pos = (tb > 0).permute(0, 3, 4, 1, 2, 5).reshape(tiles, 64, wp // 32 * 16)  # [tile, lane = 32 hi + e, 16 t + r]
        want = torch.zeros(tiles, wp // 64, 64, dtype=torch.int64)
        for b in range(wp // 32 * 16):
            want[:, b // 32, :] |= pos[:, :, b].to(torch.int64) << (b % 32)
        want = torch.where(want >= 2 ** 31, want - 2 ** 32, want).to(torch.int32)
        assert torch.equal(got.view(tiles, wp // 64, 64), want), f"block {k}: sign mask differs from (s_pre > 0)"
    ds = torch.randn(rows, dims[0], device="cuda", generator=g)
    dv = torch.randn(rows, dims[1], 3, device="cuda", generator=g)

    def run():
        with torch.no_grad():
            res = ops.gcp2_chain_backward_data(specs, rows, ins, outs, frames, ws_, packs, ds, dv, [True] * n)
        torch.cuda.synchronize()
        assert res is not None
        return _flatten(res)

    saved = ops.CHAIN_SIGN_MASKS
    try:
        ops.CHAIN_SIGN_MASKS = False
        want = run()
        ops.CHAIN_SIGN_MASKS = True
        got = run()
    finally:
        ops.CHAIN_SIGN_MASKS = saved
    for i, (a, b) in enumerate(zip(got, want)):
        assert torch.equal(a, b), f"tensor {i} differs by {float((a - b).abs().max()):.3e}"


def test_no_sign_masks_for_smooth_activations(G):
    ops, specs, ins, outs, *_ = _chain(G, (128, 16), 200, "silu")
    assert all(o[2].sign is None for o in outs if isinstance(o[2], ops.TileBlocked))
