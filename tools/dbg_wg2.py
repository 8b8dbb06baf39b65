import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
from gcpnet_amd import ops
T = ["-q", "-m", "gpu", "--tb=line", "tests/test_gpu_parity.py::test_interactions2_large_vs_oracle"]
for side in (False, True):
    ops.WEIGHT_GRADS_ON_SIDE_STREAM = side
    print("=== side stream", side, flush=True)
    pytest.main(T)
    print("=== again", flush=True)
    pytest.main(T)
