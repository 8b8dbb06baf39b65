"""Trainer-side glue of the hot path: the reference trains with torch.optim.Adam over ~90 small parameter
tensors per GCPInteractions layer (configs/model/gcpnet_nms.yaml:8-12, gcpnet_lba.yaml); FusedAdam applies the same update to all
of them in one HIP launch per 96 tensors (gcpnet_adam_step) instead of several launches per tensor."""
from __future__ import annotations

import ctypes as C
from typing import Iterable

import torch

from . import _lib, ops
from ._lib import AdamTensor, check


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay) semantics (amsgrad=False, maximize=False, L2 weight decay added to
    the gradient), state keys `step`, `exp_avg`, `exp_avg_sq` as in torch (state_dicts interchange).  fp32 CUDA parameters."""

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            # the step count and the bias corrections derived from it are host values passed as kernel arguments: a captured
            # step() would replay the capture-time corrections for ever and state["step"] would stop advancing
            raise RuntimeError("FusedAdam.step() cannot be captured into a hipGraph: call it eagerly after GraphedStep's replay")
        lib = _lib.load()
        for group in self.param_groups:
            items, step = [], None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise _lib.GcpnetHipError("FusedAdam: fp32 parameters on the GPU only (gcpnet_amd has no CPU path)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                if step is None:
                    step = st["step"]
                if st["step"] != step:  # (parameters that joined later: their own launch)
                    self._launch(lib, group, [(p, st)], st["step"])
                    continue
                items.append((p, st))
            if items:
                self._launch(lib, group, items, step)
        ops.invalidate_packs()  # (p.data was written through a raw pointer: the packed-weight caches must not outlive it)
        return loss

    @staticmethod
    def _launch(lib, group, items, step):
        arr = (AdamTensor * len(items))()
        keep = []
        for k, (p, st) in enumerate(items):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            keep.append(g)
            arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
            arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
        b1, b2 = group["betas"]
        check(lib.gcpnet_adam_step(len(items), arr, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                   float(group["weight_decay"]), int(step), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "adam_step")
