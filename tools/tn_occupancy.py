"""Resident workgroups per CU of the pipelined weight-gradient GEMM kernels, as hipOccupancyMaxActiveBlocksPerMultiprocessor reports them."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from gcpnet_amd import _lib  # noqa: E402

torch.zeros(1, device="cuda")
lib = _lib.load()
print("narrow (4 waves, 27 KB):", lib.gcpnet_debug_tn_occupancy(0), " wide (8 waves, 102 KB):", lib.gcpnet_debug_tn_occupancy(1))
