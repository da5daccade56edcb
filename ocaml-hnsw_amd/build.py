#!/usr/bin/env python3
"""Build libhnsw_mi355x.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libhnsw_mi355x.so")
SOURCES = ["hnsw_capi.hip", "hnsw_build.hip", "hnsw_layer_ops.hip", "hnsw_multi.hip", "hnsw_order.hip"]
DEPS = SOURCES + ["hnsw_device.hip.h", "hnsw_build_device.hip.h", "hnsw_internal.h",
        os.path.join(ROOT, "include", "hnsw_mi355x.h")]


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
    if not force and up_to_date() and not os.environ.get("HNSW_LIB_OUT"):
        return LIB
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include")]
    if os.environ.get("HNSW_RB_NCH2"):
        base += ["-DHNSW_RB_NCH2=" + os.environ["HNSW_RB_NCH2"]]
    if os.environ.get("HNSW_SEARCH_MIN_WAVES"):
        base += ["-DHNSW_SEARCH_MIN_WAVES=" + os.environ["HNSW_SEARCH_MIN_WAVES"]]
    extra = os.environ.get("HNSW_EXTRA_CFLAGS", "").split()   # experiments: variant builds
    base += extra
    lib_out = os.environ.get("HNSW_LIB_OUT") or LIB
    if resource_log:
        base += ["-Rpass-analysis=kernel-resource-usage"]
    objdir = os.path.join(HERE, "build") if lib_out == LIB else lib_out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s_ in SOURCES:  # one hipcc per translation unit, in parallel
        obj = os.path.join(objdir, s_ + ".o")
        cmd = base + ["-c", os.path.join(CSRC, s_), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        log = open(resource_log + "." + s_, "w") if resource_log else None
        procs.append((subprocess.Popen(cmd, stderr=log), obj, cmd, log))
    objs = []
    for pr, obj, cmd, log in procs:
        rc = pr.wait()
        if log:
            log.close()
        if rc != 0:
            raise subprocess.CalledProcessError(rc, cmd)
        objs.append(obj)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_out]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    if resource_log:
        with open(resource_log, "w") as f:
            for s_ in SOURCES:
                f.write(open(resource_log + "." + s_).read())
    return lib_out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          resource_log=(sys.argv[sys.argv.index("--resources") + 1] if "--resources" in sys.argv else None))
    print(LIB)
