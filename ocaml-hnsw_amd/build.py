#!/usr/bin/env python3
"""Build libhnsw_mi355x.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libhnsw_mi355x.so")
SOURCES = ["hnsw_capi.hip"]
DEPS = ["hnsw_capi.hip", "hnsw_device.hip.h", os.path.join(ROOT, "include", "hnsw_mi355x.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    for d in DEPS + [os.path.abspath(__file__)]:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.getmtime(p) > t:
            return False
    return True


def build(force=False, verbose=False, resource_log=None):
    if not force and up_to_date():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include")]
    if resource_log:
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    if resource_log:
        with open(resource_log, "w") as f:
            subprocess.check_call(cmd, stderr=f)
    else:
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          resource_log=(sys.argv[sys.argv.index("--resources") + 1] if "--resources" in sys.argv else None))
    print(LIB)
