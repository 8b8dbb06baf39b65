"""The RCCL branches of gcpnet_amd.parallel, executed for real: the GPU boxes of this build have ONE GPU, and RCCL refuses two
ranks on one device, so the multi-rank tests run over gloo (tests/test_parallel_gloo.py, tests/test_sharded_gpu.py).  What those
cannot cover is that the `backend == "nccl"` code paths exist in this stack at all -- `ReduceOp.AVG` on the flat gradient bucket,
`all_gather_into_tensor` / `reduce_scatter_tensor` inside the all_gather_rows autograd Function, `all_to_all_single` of the halo
mode, a device-bound process group.
A process group of size 1 over RCCL runs exactly those calls (a subprocess, so that the suite's process keeps no group)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

JOB = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GCP_REPO"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import gcpnet_amd as G
from gcpnet_amd import ops
from gcpnet_amd.parallel import GradAllReducer, ShardedGraph, sharded_interactions_forward
from tests.helpers import rand_graph
assert dist.get_backend() == "nccl"
torch.manual_seed(0)
n, e, dims = 300, 3000, (128, 16)
layer = G.GCPInteractions(dims, (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu"), layer_cfg=G.default_layer_cfg(),
                          dropout=0.0).cuda().eval()
ei, x = rand_graph(n, e, 3, sort_by_col=True)
g = torch.Generator().manual_seed(1)
ins = {k: v.cuda().requires_grad_() for k, v in dict(h=torch.randn(n, 128, generator=g), chi=torch.randn(n, 16, 3, generator=g),
                                                     e=torch.randn(e, 32, generator=g), xi=torch.randn(e, 4, 3, generator=g)).items()}
lw = dict(h=torch.randn(n, 128, generator=g).cuda(), chi=torch.randn(n, 16, 3, generator=g).cuda())
fr = G.localize(x.cuda(), ei.cuda())
# unsharded reference on the same GPU
(h, chi) = layer((ins["h"], ins["chi"]), (ins["e"], ins["xi"]), ei.cuda(), fr)
((h * lw["h"]).sum() + (chi * lw["chi"]).sum()).backward()
ref = dict(h=h.detach().clone(), chi=chi.detach().clone(), **{"d" + k: v.grad.clone() for k, v in ins.items()})
wref = [p.grad.clone() for p in layer.parameters()]
for v in ins.values():
    v.grad = None
for p in layer.parameters():
    p.grad = None
# the sharded path with ONE rank over RCCL: all_gather_into_tensor / reduce_scatter_tensor / ReduceOp.AVG and SUM on the bucket
sg = ShardedGraph(ei, n, 0, 1)
sg.to("cuda")
fr_l = G.localize(x.cuda(), sg.edge_index_global)
fr_out = G.localize(x.cuda(), sg.out_edge_index_global)
node_frames = ops.segment_reduce(fr_out.reshape(-1, 9), ops.GatherPlan(sg.out_row_local, sg.n_local), mean=True).reshape(sg.n_local, 3, 3)
red = GradAllReducer(list(layer.parameters()))
e_l, xi_l = sg.local_edges(ins["e"]), sg.local_edges(ins["xi"])
h2, chi2 = sharded_interactions_forward(layer, (ins["h"], ins["chi"]), (e_l, xi_l), sg, fr_l, node_frames)
((h2 * lw["h"]).sum() + (chi2 * lw["chi"]).sum()).backward()
red.all_reduce_sum()
torch.cuda.synchronize()
def close(a, b, name):
    err = float((a - b).abs().max()); scale = float(b.abs().max())
    assert err <= 2e-5 * max(scale, 1.0), f"{name}: {err:.3e} (scale {scale:.3e})"
close(h2, ref["h"], "h"); close(chi2, ref["chi"], "chi")
for k in ("h", "chi"):
    close(ins[k].grad, ref["d" + k], "d" + k)
for p, w in zip(layer.parameters(), wref):
    close(p.grad, w, "weight gradient (bucket, SUM)")
red.all_reduce_mean()  # ReduceOp.AVG inside RCCL: the identity for one rank
torch.cuda.synchronize()
for p, w in zip(layer.parameters(), wref):
    close(p.grad, w, "weight gradient (bucket, AVG)")
# the halo-exchange mode: all_to_all_single over RCCL (one rank: every split is empty, the call itself and its autograd pairing run)
for v in ins.values():
    v.grad = None
for p in layer.parameters():
    p.grad = None
sgh = ShardedGraph(ei, n, 0, 1, halo=True).to("cuda")
assert sgh.table_rows == n and sum(sgh.send_counts) == 0
h3, chi3 = sharded_interactions_forward(layer, (ins["h"], ins["chi"]), (e_l, xi_l), sgh, fr_l, node_frames)
((h3 * lw["h"]).sum() + (chi3 * lw["chi"]).sum()).backward()
torch.cuda.synchronize()
close(h3, ref["h"], "h (halo)"); close(chi3, ref["chi"], "chi (halo)")
for k in ("h", "chi"):
    close(ins[k].grad, ref["d" + k], "d" + k + " (halo)")
# ... and with real traffic: one rank sending rows to itself through the same call
send = torch.arange(12, dtype=torch.float32, device="cuda").reshape(4, 3)
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send, output_split_sizes=[4], input_split_sizes=[4])
torch.cuda.synchronize()
assert torch.equal(recv, send)
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def test_rccl_branches_with_one_rank():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket

    with socket.socket() as sock:  # a free port (a fixed one collides with anything else rendezvousing on this box)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, GCP_REPO=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", JOB], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


DDP_JOB = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["GCP_REPO"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import gcpnet_amd as G
from gcpnet_amd import ops
from tests.helpers import rand_graph
from torch.nn.parallel import DistributedDataParallel as DDP

class Stack(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = torch.nn.ModuleList(G.GCPInteractions((64, 16), (32, 4), cfg=G.default_module_cfg(scalar_nonlinearity="silu"),
                                                            layer_cfg=G.default_layer_cfg(), dropout=0.0) for _ in range(2))
    def forward(self, h, chi, e, xi, ei, fr):
        for l in self.layers:
            h, chi = l((h, chi), (e, xi), ei, fr)
        return h, chi

torch.manual_seed(0)
n, e = 200, 2400
model = Stack().cuda().train()
ei, x = rand_graph(n, e, 3, sort_by_col=True)
g = torch.Generator().manual_seed(1)
ins = [torch.randn(n, 64, generator=g).cuda(), torch.randn(n, 16, 3, generator=g).cuda(), torch.randn(e, 32, generator=g).cuda(),
       torch.randn(e, 4, 3, generator=g).cuda()]
lw = [torch.randn(n, 64, generator=g).cuda(), torch.randn(n, 16, 3, generator=g).cuda()]
fr = G.localize(x.cuda(), ei.cuda())
def run(m):
    for p in model.parameters():
        p.grad = None
    h, chi = m(*ins, ei.cuda(), fr)
    ((h * lw[0]).sum() + (chi * lw[1]).sum()).backward()
    torch.cuda.synchronize()
    return [p.grad.clone() for p in model.parameters()]
want = run(model)  # unwrapped (no process-group consumer of the gradients: the weight-gradient stream may be used)
ddp = DDP(model, device_ids=[0])  # what the reference's trainer does (configs/trainer/default.yaml:8, ddp.yaml)
for side in (True, False):
    ops.WEIGHT_GRADS_ON_SIDE_STREAM = side
    got = run(ddp)
    for (name, _), a, b in zip(model.named_parameters(), got, want):
        err, scale = float((a - b).abs().max()), float(b.abs().max())
        assert err <= 1e-6 * max(scale, 1.0), f"side stream {side}: {name}: {err:.3e} (scale {scale:.3e})"
dist.barrier()
dist.destroy_process_group()
print("DDP_WORLD1_OK")
"""


def test_distributed_data_parallel_wrapper_gets_the_unwrapped_gradients():
    """INTEGRATION.md says the mirror works under torch's DistributedDataParallel, which is what the reference's trainer uses
    (configs/trainer/default.yaml:8): DDP's bucket hooks read every weight gradient INSIDE the backward pass, so the weight-gradient
    stream must stand down by itself (ops._side_stream_ok).  One rank over RCCL; gradients must equal the unwrapped model's with the
    side-stream switch on and off."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, GCP_REPO=root, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", DDP_JOB], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DDP_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
