"""N > 1 path on CPU: two gloo ranks.  Checks (1) the flat-bucket gradient all-reduce against a single-process mean,
(2) that sharding a collated batch by whole graphs + all-reducing a readout reproduces the unsharded result (graphs are
block-diagonal, so no data-path collective is needed -- SURVEY.md section 8e), using the oracle as the per-rank model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gcpnet_amd.parallel import GradAllReducer, shard_graph_batch
from tests.helpers import Fixture


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return dict(ret)


def _grad_job(rank, world):
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    x = torch.arange(20, dtype=torch.float32).reshape(4, 5) * (rank + 1)
    lin(x).square().mean().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    GradAllReducer(lin.parameters()).all_reduce_mean()
    return [p.grad.clone() for p in lin.parameters()], local


def test_grad_all_reducer_matches_mean_of_ranks():
    out = _run(_grad_job)
    mean = [(a + b) / 2 for a, b in zip(out[0][1], out[1][1])]
    for r in (0, 1):
        for g, m in zip(out[r][0], mean):
            assert torch.allclose(g, m, atol=1e-6)


def _shard_job(rank, world):
    from oracle import gcp_oracle as O

    f = Fixture("model_lba_small")
    batch = {k: v for k, v in f.i.items()}
    part = shard_graph_batch(batch, rank, world)
    cfg, lc = O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4)
    n_local = int(part["batch"].max()) + 1 if part["batch"].numel() else 0
    pred = O.lba_forward(f.p, part, cfg, lc, 2)["pred"].reshape(-1) if n_local else torch.zeros(0)
    # readout all-reduce: sum of per-graph predictions and graph count (what a loss/metric reduction needs)
    t = torch.tensor([pred.sum().item(), float(n_local)], dtype=torch.float64)
    dist.all_reduce(t)
    return pred, part["edge_index"].shape[1], t.tolist()


def test_graph_sharding_reproduces_unsharded_readout():
    from oracle import gcp_oracle as O

    f = Fixture("model_lba_small")
    full = O.lba_forward(f.p, f.i, O.default_module_cfg(), O.default_layer_cfg(num_message_layers=4), 2)["pred"]
    out = _run(_shard_job)
    joined = torch.cat((out[0][0], out[1][0]))
    assert joined.shape == full.shape
    assert torch.allclose(joined, full, atol=1e-5, rtol=1e-4)
    assert out[0][1] + out[1][1] == f.i["edge_index"].shape[1]  # every edge lives on exactly one rank
    assert abs(out[0][1] - out[1][1]) <= f.i["edge_index"].shape[1] // 2  # balanced by edges
    assert out[0][2] == out[1][2] and abs(out[0][2][0] - full.sum().item()) < 1e-4 and out[0][2][1] == full.numel()


def test_shard_graph_batch_single_process():
    f = Fixture("model_nms_small")
    b = f.i
    parts = [shard_graph_batch(b, r, 3) for r in range(3)]
    assert sum(p["h"].shape[0] for p in parts) == b["h"].shape[0]
    for p in parts:
        assert int(p["edge_index"].max()) < p["h"].shape[0] and int(p["edge_index"].min()) >= 0
        assert torch.equal(torch.unique(p["batch"]), torch.arange(int(p["batch"].max()) + 1))
