#!/usr/bin/env python3
"""Regenerates tests/golden/c1_oracle_checksum.json: BASELINE.json configs[0] ("C1": 10 k random fp32
vectors, d = 32, M = 8, efConstruction = 100, ef = 32, k = 10, 1000 queries) run through the CPU
oracle.  The fixture is a REGRESSION pin of this repository's own restatement (own seeded RNG for the
level draws), not a vector of the reference: it keeps the checker from drifting unnoticed."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run():
    from oracle import oracle as o
    o.build()
    rng = np.random.default_rng(0)
    X = rng.uniform(-1, 1, size=(10000, 32)).astype(np.float32)      # Lacaml Mat.random range, benchmark/dataset.ml:48
    Q = rng.uniform(-1, 1, size=(1000, 32)).astype(np.float32)
    out = {}
    for name, arith in (("seq_f32", o.SEQ_F32), ("tree16", o.TREE16)):
        sp = o.Space.l2(X, arith=arith)
        g = o.build_ohnsw(sp, 8, 100, seed=0)
        ids, dist = o.Ohnsw.knn_batch_bigarray(g, sp, Q, k=10, ef=32, ties=o.TIES_CANONICAL)
        gt_ids, gt_dist = o.brute_force_knn(o.Space.l2(X, arith=o.F64), Q, 10)
        rec_ids = float(np.mean([len(set(a) & set(b)) / 10 for a, b in zip(ids.tolist(), gt_ids.tolist())]))
        out[name] = {
            "max_layer": int(g.max_layer), "entry_point": int(g.entry_point),
            "deg0_sum": int(g.deg0.sum()),
            "ids_sha256": hashlib.sha256(np.ascontiguousarray(ids, np.int32).tobytes()).hexdigest(),
            "dist_sha256": hashlib.sha256(np.ascontiguousarray(dist, np.float32).tobytes()).hexdigest(),
            "recall_at_10": round(rec_ids, 4),
            "recall_distance_threshold": round(float(o.recall_distance_threshold(gt_dist, dist)), 4),
        }
    return out


if __name__ == "__main__":
    res = run()
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c1_oracle_checksum.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
