"""The randomised parity sweeps (tests/sweep_gcp2.py: single GCP2 blocks of random dims / gating / activations / options;
tests/sweep_layers.py: whole GCPInteractions layers) as collected GPU tests, a short run of each with a seed of its own -- the
full-length sweeps found the shapes behind the regression cases of tests/test_wg_kernels.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,n,seed", [("sweep_gcp2.py", 30, 11), ("sweep_layers.py", 16, 11)])
def test_random_shapes_against_the_oracle(script, n, seed):
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", script), str(n), str(seed)], cwd=ROOT, capture_output=True,
                       text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "mismatches: 0" in r.stdout, tail
