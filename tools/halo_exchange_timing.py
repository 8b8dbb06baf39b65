"""Times ShardedGraph's two exchange modes (whole-table all-gather / reduce-scatter vs. halo all-to-all) on the device, two ranks
sharing cuda:0 over gloo (a one-GPU box; RCCL needs one device per rank).  python tools/halo_exchange_timing.py [nodes] [s_dim]"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def job(rank, world, nodes, width):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gcpnet_amd.parallel import ShardedGraph, spatial_order
    from gcpnet_amd.synthetic import make_inputs, reorder_nodes

    torch.cuda.set_device(0)
    host = make_inputs(nodes, 16, (width, 16), (32, 4), seed=0)
    host = reorder_nodes(host, spatial_order(host["x"]))
    for halo in (False, True):
        sg = ShardedGraph(host["edge_index"], nodes, rank, world, halo=halo).to("cuda")
        loc = sg.local_nodes(host["h"]).contiguous().cuda()
        for _ in range(3):
            t = sg._gather(loc)
        torch.cuda.synchronize(); dist.barrier(); t0 = time.time()
        for _ in range(20):
            t = sg._gather(loc)
        torch.cuda.synchronize(); dist.barrier(); t1 = time.time()
        for _ in range(20):
            sg._scatter_sum(t)
        torch.cuda.synchronize(); dist.barrier(); t2 = time.time()
        if rank == 0:
            print(f"halo={halo}: table rows {sg.table_rows} (local {sg.n_local}); forward exchange {(t1 - t0) / 20 * 1e3:.2f} ms, "
                  f"backward {(t2 - t1) / 20 * 1e3:.2f} ms", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    mp.spawn(job, args=(2, n, w), nprocs=2)
