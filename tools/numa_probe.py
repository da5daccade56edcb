#!/usr/bin/env python3
"""Where do the library's page-locked matrices live, and does the NUMA node of the calling thread matter?  (GPU box)

Prints the node of hnsw_host_alloc'd pages (/proc/self/numa_maps) with the process bound to the GPU's node and to the other one,
and the time of the host-protocol call on C2's shape either way."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H


def cpus_of(node):
    out = []
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def node_of(arr):
    addr = arr.ctypes.data
    for line in open("/proc/self/numa_maps"):
        f = line.split()
        if int(f[0], 16) <= addr < int(f[0], 16) + (1 << 40) and any(x.startswith("N") and "=" in x for x in f):
            if int(f[0], 16) == addr & ~0xFFF or int(f[0], 16) == addr:
                return " ".join(x for x in f if x[0] == "N" and "=" in x)
    return "?"


p = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
gnode = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
print("GPU 0 at %s, NUMA node %d" % (bus, gnode), flush=True)
n, d, nq, k, ef = int(os.environ.get("N", 200000)), 128, 10000, 10, 128
rng = np.random.default_rng(1)
X = rng.integers(0, 256, size=(n, d)).astype(np.float32)
hg = H.Ohnsw.build_batch_bigarray(X, 16, 100, seed=1, metric=0)
nodes = sorted(int(x[4:]) for x in os.listdir("/sys/devices/system/node") if x.startswith("node"))
for node in [gnode] + [x for x in nodes if x != gnode] + [gnode]:
    os.sched_setaffinity(0, cpus_of(node))
    Q = H.host_empty((nq, d), np.float32)
    Q[:] = rng.integers(0, 256, size=(nq, d)).astype(np.float32)
    oi = H.host_empty((nq, k), np.int32); od = H.host_empty((nq, k), np.float32)
    for _ in range(5):
        H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, out=(oi, od))
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); H.Ohnsw.knn_batch_bigarray(hg, k, Q, ef=ef, out=(oi, od)); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("thread on node %d: queries' pages %s; host call median %.4f ms, min %.4f" % (node, node_of(Q), 1e3 * ts[15], 1e3 * ts[0]), flush=True)
    # a pageable matrix first touched on this node and then registered
    Qn = np.empty((nq, d), np.float32); Qn[:] = Q
    H.pin(Qn)
    for _ in range(5):
        H.Ohnsw.knn_batch_bigarray(hg, k, Qn, ef=ef, out=(oi, od))
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); H.Ohnsw.knn_batch_bigarray(hg, k, Qn, ef=ef, out=(oi, od)); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("   registered numpy matrix touched on node %d: host call median %.4f ms, min %.4f" % (node, 1e3 * ts[15], 1e3 * ts[0]), flush=True)
    H.unpin(Qn)
