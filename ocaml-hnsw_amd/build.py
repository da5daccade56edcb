#!/usr/bin/env python3
"""Build libhnsw_mi355x.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libhnsw_mi355x.so")
SOURCES = ["hnsw_capi.hip", "hnsw_build.hip", "hnsw_layer_ops.hip", "hnsw_multi.hip", "hnsw_order.hip", "hnsw_rows8.hip",
           "hnsw_rows_split.hip", "hnsw_locality.hip"]
# the knn kernel's variants: one object per (metric, accept rule, row shape), see hnsw_search_variants.hip
VARIANT_SOURCE = "hnsw_search_variants.hip"
VARIANTS = [(m, s, f) for m in (0, 1) for s in (0, 1) for f in (0, 1, 2, 3)]   # f: rows ragged fp32 / full fp32 / bytes / split fp32
DEPS = SOURCES + [VARIANT_SOURCE, "hnsw_device.hip.h", "hnsw_hop_asm.hip.h", "hnsw_hop_loop.inc", "hnsw_hop_slots.inc", "hnsw_hop_instances.inc", "hnsw_build_device.hip.h", "hnsw_internal.h",
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
        if verbose:
            print("build_mode: reused %s (newer than every source; --force recompiles)" % LIB, flush=True)
        return LIB
    # -fno-slp-vectorize: the SLP vectorizer pairs the fp32 chains of two row batches into v_pk_fma_f32,
    # which costs extra register moves and is no faster than two v_fma_f32 on this SIMD
    base = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
            "-Wall", "-Wextra", "-Wno-unused-parameter",        # the sources are clean under these: keep them so
            "-I", os.path.join(ROOT, "include")]
    if os.environ.get("HNSW_RB_NCH2"):
        base += ["-DHNSW_RB_NCH2=" + os.environ["HNSW_RB_NCH2"]]
    if os.environ.get("HNSW_SEARCH_MIN_WAVES"):
        base += ["-DHNSW_SEARCH_MIN_WAVES(NCH,NSLOT,METRIC,FULL,SEMF)=" + os.environ["HNSW_SEARCH_MIN_WAVES"]]
    extra = os.environ.get("HNSW_EXTRA_CFLAGS", "").split()   # experiments: variant builds
    base += extra
    lib_out = os.environ.get("HNSW_LIB_OUT") or LIB
    if resource_log:
        base += ["-Rpass-analysis=kernel-resource-usage"]
    objdir = os.path.join(HERE, "build") if lib_out == LIB else lib_out + ".obj"
    os.makedirs(objdir, exist_ok=True)
    units = [(s_, s_, []) for s_ in SOURCES]
    units += [("hnsw_search_variants_%d_%d_%d" % v, VARIANT_SOURCE,
               ["-DHNSW_V_METRIC=%d" % v[0], "-DHNSW_V_SEMF=%d" % v[1], "-DHNSW_V_FULL=%d" % v[2]]) for v in VARIANTS]
    jobs = max(1, int(os.environ.get("HNSW_BUILD_JOBS", os.cpu_count() or 4)))
    pending = list(units)
    running, objs, logs = [], [], []
    n_compiled = [0]
    headers = [os.path.join(CSRC, h) for h in DEPS if isinstance(h, str) and h.endswith((".h", ".inc")) and not os.path.isabs(h)]
    headers += [h for h in DEPS if os.path.isabs(h)]
    newest_header = max(os.path.getmtime(h) for h in headers)

    def fresh(obj, src, cmd):
        """an object is kept when it is newer than its source and every header and was made by the same command"""
        try:
            return (not force and not resource_log and os.path.getmtime(obj) >= max(newest_header, os.path.getmtime(src))
                    and open(obj + ".cmd").read() == " ".join(cmd))
        except OSError:
            return False
    while pending or running:   # one hipcc per translation unit, at most `jobs` at a time
        while pending and len(running) < jobs:
            name, src, defs = pending.pop(0)
            obj = os.path.join(objdir, name + ".o")
            cmd = base + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
            if fresh(obj, os.path.join(CSRC, src), cmd):
                objs.append(obj)
                continue
            if os.path.exists(obj + ".cmd"):
                os.remove(obj + ".cmd")
            n_compiled[0] += 1
            if verbose:
                print(" ".join(cmd), flush=True)
            log = open(resource_log + "." + name, "w") if resource_log else None
            if log:
                logs.append(resource_log + "." + name)
            running.append((subprocess.Popen(cmd, stderr=log), obj, cmd, log))
        if not running:
            continue
        pr, obj, cmd, log = running.pop(0)
        rc = pr.wait()
        if log:
            log.close()
        if rc != 0:
            for other in running:
                other[0].kill()
            raise subprocess.CalledProcessError(rc, cmd)
        with open(obj + ".cmd", "w") as f:
            f.write(" ".join(cmd))
        objs.append(obj)
    if verbose:
        print("build_mode: compiled %d of %d translation units for gfx950 (the others were up to date), linking" % (n_compiled[0], len(units)), flush=True)
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_out]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    if resource_log:
        with open(resource_log, "w") as f:
            for part in logs:
                f.write(open(part).read())
                os.remove(part)
    return lib_out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          resource_log=(sys.argv[sys.argv.index("--resources") + 1] if "--resources" in sys.argv else None))
    print(LIB)
