"""Builds libgcpnet_hip.so (gfx950) in-tree with hipcc.  Usage: python -m gcpnet_amd.csrc.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gcp2_fwd.hip", "gcp2_chain_fwd.hip", "gcp2_bwd.hip", "gcp2_chain_bwd.hip", "tn_gemm.hip", "graph_ops.hip", "gcp_wg_fwd.hip", "gcp_wg_bwd.hip", "misc_ops.hip", "featurize.hip"]
HEADERS = ["common.h", "tile_io.h", "vec_mfma.h", "gcp_wg.h", "gcp_bf16x3.h", "gcp_f16x2.h", os.path.join("..", "..", "include", "gcpnet_hip.h")]
LIB = os.path.join(HERE, "libgcpnet_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _extra():
    return os.environ.get("GCPNET_HIPCC_EXTRA", "")


def _stamp(obj):
    """Per-object record of the flags the object was compiled with (written right after THAT compile succeeds, so a failed or
    partial build with measurement flags cannot leave objects that a later plain build links)."""
    return obj + ".flags"


def _obj_fresh(src, obj, dep_t, flags):
    st = _stamp(obj)
    return (os.path.exists(obj) and os.path.exists(st) and open(st).read() == flags
            and os.path.getmtime(obj) > max(dep_t, os.path.getmtime(os.path.join(HERE, src))))


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS + ["build.py"]):
        return True
    flag_file = os.path.join(HERE, ".build_flags")  # flags of the last successful LINK
    return not (os.path.exists(flag_file) and open(flag_file).read() == _extra())


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    extra = _extra()
    flags = " ".join(FLAGS) + " | " + extra
    flag_file = os.path.join(HERE, ".build_flags")
    dep_t = max(os.path.getmtime(os.path.join(HERE, f)) for f in HEADERS + ["build.py"])
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(obj)
        # incremental: an object newer than its source, every header and this script, stamped with the SAME flags, is kept
        if not force and _obj_fresh(src, obj, dep_t, flags):
            continue
        if os.path.exists(_stamp(obj)):
            os.remove(_stamp(obj))
        cmd = [hipcc] + FLAGS + extra.split() + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = []
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed.append(f"hipcc failed on {src}:\n{out.decode()}")
            continue
        with open(_stamp(obj), "w") as f:
            f.write(flags)
        if verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.run(cmd, check=True)
    with open(flag_file, "w") as f:
        f.write(extra)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
