"""hipGraph capture of a whole training step (forward + loss + backward [+ optimizer]) for launch-bound configurations.  With
`optimizer=FusedAdam(..., capturable=True)` the Adam update is part of the capture (its step count lives in device memory and is
advanced by the update itself); an optimizer with a host-side step count stays outside (its step() refuses to run under capture).

A GCPNet step on the n-body batches of the NMS task is ~700 kernel launches for 2 000 - 38 000 edges: the GPU idles between launches
while Python and the HIP runtime enqueue them.  Captured once into a hipGraph (torch.cuda.CUDAGraph; the ctypes launches of this
package go to torch's current stream, so they are recorded like any other kernel) the step replays from a single call.

Requirements (the usual ones of CUDA/HIP graphs): static shapes and static input tensors (copy new data INTO them), no host
synchronisation inside the step (index plans are cached on their tensors: GraphPlan.get / GatherPlan.get; run a few eager warm-up
steps first, which `GraphedStep` does), dropout masks would be frozen (capture with dropout 0 / eval dropout)."""
from __future__ import annotations

import os
from typing import Callable, Optional

import torch

from . import ops


class GraphedStep:
    def __init__(self, step_fn: Callable[[], torch.Tensor], warmup: int = 3, optimizer=None, side_stream: Optional[bool] = None):
        """step_fn: zeroes the gradients, runs forward + loss + backward, returns the loss.  optimizer (optional, capturable):
        optimizer.step() runs behind step_fn in every warm-up step and inside the capture -- the `warmup` eager steps DO update the
        parameters, the capture pass itself executes nothing.
        side_stream: keep the weight-gradient stream inside the capture -- its launches fork from the capturing stream
        (`side.wait_stream(main)`) and rejoin it in the end-of-backward callback, i.e. they become a parallel branch of the graph.
        None = the GCPNET_GRAPH_SIDE_STREAM environment variable (default off: one stream inside the capture)."""
        if side_stream is None:
            side_stream = os.environ.get("GCPNET_GRAPH_SIDE_STREAM", "0") == "1"
        def full_step():
            loss = step_fn()
            if optimizer is not None:
                optimizer.step()
            return loss

        self._saved_side = ops.WEIGHT_GRADS_ON_SIDE_STREAM
        if not side_stream:
            ops.WEIGHT_GRADS_ON_SIDE_STREAM = False  # one stream inside the capture
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    full_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ops.invalidate_packs()  # every packed-weight image is rebuilt INSIDE the graph, i.e. at every replay
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.loss = full_step()
        finally:
            ops.WEIGHT_GRADS_ON_SIDE_STREAM = self._saved_side
        ops.invalidate_packs()  # (eager calls after this must not trust images that live in the graph's private pool)

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss
