"""Builds libgcpnet_hip.so (gfx950) in-tree with hipcc.  Usage: python -m gcpnet_amd.csrc.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gcp2_fwd.hip", "gcp2_chain_fwd.hip", "gcp2_bwd.hip", "gcp2_chain_bwd.hip", "tn_gemm.hip", "graph_ops.hip", "gcp_wg_fwd.hip", "gcp_wg_bwd.hip", "misc_ops.hip", "featurize.hip"]
HEADERS = ["common.h", "tile_io.h", "vec_mfma.h", "gcp_wg.h", "gcp_bf16x3.h", os.path.join("..", "..", "include", "gcpnet_hip.h")]
LIB = os.path.join(HERE, "libgcpnet_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    extra = os.environ.get("GCPNET_HIPCC_EXTRA", "")
    flag_file = os.path.join(HERE, ".build_flags")
    flags_same = os.path.exists(flag_file) and open(flag_file).read() == extra
    dep_t = max(os.path.getmtime(os.path.join(HERE, f)) for f in HEADERS + ["build.py"])
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(obj)
        # incremental: an object newer than its source, every header and this script (built with the same extra flags) is kept
        if (not force and flags_same and os.path.exists(obj)
                and os.path.getmtime(obj) > max(dep_t, os.path.getmtime(os.path.join(HERE, src)))):
            continue
        cmd = [hipcc] + FLAGS + os.environ.get("GCPNET_HIPCC_EXTRA", "").split() + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    with open(flag_file, "w") as f:
        f.write(extra)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
