"""How do the wave-per-tile chain kernels scale with the number of 32-row tiles?  Times the forward and the backward (data) launch of
one layer's 7-block (s,V) chain for a list of tile counts -- the question behind it: 2 waves per SIMD x 1024 SIMDs = 2048 tile slots,
so 4998 tiles (configs[1]) are 2.44 "rounds"; does the partial last round cost a whole one?

  python tools/chain_rows_sweep.py [--sdim 128 --vdim 16] [--tiles 512,1024,2048,3072,4096,4998,6144]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gcpnet_amd as G  # noqa: E402
from gcpnet_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sdim", type=int, default=128)
    ap.add_argument("--vdim", type=int, default=16)
    ap.add_argument("--tiles", default="256,512,1024,1536,2048,3072,4096,4998,5120,6144,8192")
    ap.add_argument("--iters", type=int, default=15)
    args = ap.parse_args()
    torch.manual_seed(0)
    layer = G.GCPInteractions((args.sdim, args.vdim), (32, 4), cfg=G.default_module_cfg(), layer_cfg=G.default_layer_cfg(),
                              dropout=0.0).cuda()
    blocks = list(layer.interaction.message_fusion[1:])
    n = len(blocks)
    specs = [b.make_spec([None], [None], residual=True) for b in blocks]
    ws = [tuple(None if t is None else t.detach().requires_grad_() for t in b._weights()) for b in blocks]
    out = []
    for tiles in [int(t) for t in args.tiles.split(",")]:
        rows = tiles * 32 if tiles != 4998 else 159913
        g = torch.Generator(device="cuda").manual_seed(0)
        s = torch.randn(rows, args.sdim, device="cuda", generator=g).requires_grad_()
        v = torch.randn(rows, args.vdim, 3, device="cuda", generator=g).requires_grad_()
        ds = torch.randn(rows, args.sdim, device="cuda", generator=g)
        dv = torch.randn(rows, args.vdim, 3, device="cuda", generator=g)
        x = torch.randn(rows, 3, 3, device="cuda", generator=g)
        frames = x / x.norm(dim=-1, keepdim=True)
        keep = {}

        def timeit(f):
            for _ in range(3):
                f()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.iters)]
            torch.cuda.synchronize()
            for a, b in ev:
                a.record()
                f()
                b.record()
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in ev)
            return ts[len(ts) // 2]

        def fwd():
            keep.pop("out", None)
            keep["out"] = ops.gcp2_chain(specs, s, v, frames, ws)

        t_f = timeit(fwd)
        s0, v0, ws_, packs, outs = keep["out"][0].grad_fn.state
        ins = [(s0, v0) if k == 0 else (outs[k - 1][0], outs[k - 1][1]) for k in range(n)]
        with torch.no_grad():
            def bwd():
                keep["bwd"] = ops.gcp2_chain_backward_data(specs, rows, ins, outs, frames, ws_, packs, ds, dv, [True] * n)

            t_b = timeit(bwd)
        rec = dict(tiles=tiles, rows=rows, fwd_ms=round(t_f, 4), bwd_ms=round(t_b, 4), fwd_us_per_ktile=round(1e3 * t_f / tiles * 1e3, 2),
                   bwd_us_per_ktile=round(1e3 * t_b / tiles * 1e3, 2))
        print(json.dumps(rec), flush=True)
        out.append(rec)
        keep.clear()
        del s, v, ds, dv
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
