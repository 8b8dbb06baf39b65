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

    def __init__(self, params: Iterable, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 capturable: bool = False):
        """capturable: the step count lives in device memory (one int64 per parameter group, advanced by the update itself:
        gcpnet_adam_step_dev), state["step"] is that tensor as in torch.optim.Adam(capturable=True), and step() may be captured
        into a hipGraph (gcpnet_amd.graphs.GraphedStep(step_fn, optimizer=...)).  lr / betas / eps / weight_decay are frozen into
        a capture; every parameter of a group must take part from the first step on."""
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=capturable))
        # capturable groups: the device step counter and the parameters that take part, by group index.  NOT in param_groups:
        # torch serialises every key of a group, and a counter that came back from torch.load(map_location="cpu") would hand a
        # host pointer to the kernel; load_state_dict drops them and the next step rebuilds them from state["step"]
        self._step_dev = {}
        self._members = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_dev, self._members = {}, {}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._step_dev, self._members = {}, {}
        for g in self.param_groups:
            g.pop("_step_dev", None)  # (pickles of the earlier layout)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            if group.get("capturable"):
                self._step_capturable(lib, group, gi)
                continue
            if capturing:
                # the step count and the bias corrections derived from it are host values passed as kernel arguments: a captured
                # step() would replay the capture-time corrections for ever and state["step"] would stop advancing
                raise RuntimeError("FusedAdam.step() with a host-side step count cannot be captured into a hipGraph: construct the "
                                   "optimizer with capturable=True, or call it eagerly after GraphedStep's replay")
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

    def _step_capturable(self, lib, group, gi):
        """One update of a capturable group: the step count is read and advanced on the device."""
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return
        # one counter per group: a parameter that skipped a step would get the bias corrections of the wrong step count
        # (torch.optim.Adam counts per parameter), so the set that takes part is fixed by the first step
        members = tuple(id(p) for p in ps)
        if self._members.setdefault(gi, members) != members:
            raise RuntimeError("FusedAdam(capturable=True): the parameters with a gradient changed between two steps of a group "
                               "(one device step counter per group: every parameter must take part in every step)")
        dev_step = self._step_dev.get(gi)
        if dev_step is None or dev_step.device != ps[0].device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdam(capturable=True): run at least one eager step before capturing (state is created there)")
            known = [self.state[p]["step"] for p in ps if self.state.get(p) and "step" in self.state[p]]  # (a loaded state_dict: per-parameter values)
            start = int(round(float(known[0].item() if torch.is_tensor(known[0]) else known[0]))) if known else 0
            dev_step = self._step_dev[gi] = torch.full((1,), start, dtype=torch.int64, device=ps[0].device)
        items = []
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32:
                raise _lib.GcpnetHipError("FusedAdam: fp32 parameters on the GPU only (gcpnet_amd has no CPU path)")
            st = self.state[p]
            if "exp_avg" not in st:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("FusedAdam(capturable=True): a parameter without state inside a capture")
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["step"] = dev_step  # (one counter per group: every parameter takes part in every step)
            items.append((p, st))
        arr, keep = self._tensors(items)
        b1, b2 = group["betas"]
        check(lib.gcpnet_adam_step_dev(len(items), arr, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                       float(group["weight_decay"]), C.c_void_p(dev_step.data_ptr()),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "adam_step_dev")
        del keep

    @staticmethod
    def _tensors(items):
        arr = (AdamTensor * len(items))()
        keep = []
        for k, (p, st) in enumerate(items):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            keep.append(g)
            arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
            arr[k].exp_avg, arr[k].exp_avg_sq, arr[k].n = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
        return arr, keep

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
