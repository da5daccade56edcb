#!/usr/bin/env python3
"""Which visited-cache POLICY would stop the re-evaluations on clustered data?  (run on the GPU box)

Companion of tools/visited_cache_sim.py (which varies sets x ways of the round-4 policy).  The exact layer-0 walk
(lib/ohnsw.ml:543-588) of a few queries is recorded once as a trace of ENCOUNTERS -- one per (expanded node, neighbour)
pair, in adjacency-row order -- with what an exact Visited (lib/ohnsw.ml:256-268) would answer, the neighbour's
distance, whether W accepted it and whether it sits in W at that moment.  Every policy then replays the same trace; a
policy can only change how often a row is fetched again, never a result (a forgotten node cannot re-enter W).

    N=10000000 D=96 M=32 EF=512 METRIC=0 python tools/visited_policy_sim.py [queries]      # C5_clustered's shape
    python tools/visited_policy_sim.py [queries]                                           # C3_clustered's shape

Policies (capacity = sets x ways tags, as the LDS of a wave holds them):
  fifo         round 4: insert at the filter (before the evaluation), newest tag in way 0, the others move down
  fifo+W       the same behind a membership test against the current W (what would a W lookup save?)
  keep         insert AFTER the evaluation; a neighbour that W accepted enters way 0 (the others move down), a rejected
               one only replaces the LAST way: tags of nodes close to the query -- the ones many expansions meet again --
               outlive the far ones
  keep1        as keep, but rejected neighbours are not remembered at all when the last way holds an accepted tag
  opt          Belady's optimum for a fully associative cache of the same capacity with bypass: the bound for ANY policy
"""
import heapq
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ocaml_hnsw_amd as H

N = int(os.environ.get("N", 1183514)); D = int(os.environ.get("D", 100)); M = int(os.environ.get("M", 32))
EF = int(os.environ.get("EF", 256)); METRIC = int(os.environ.get("METRIC", 1)); NQ = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)


CLUSTER_OF = {}


def clustered(n, seed, centres=256, spread=1.5):
    g = torch.Generator(device=dev); g.manual_seed(4321)
    cen = torch.randn((centres, D), generator=g, device=dev); cen = cen / cen.norm(dim=1, keepdim=True)
    g.manual_seed(seed)
    out = np.empty((n, D), np.float32)
    for s in range(0, n, 1 << 20):
        m = min(1 << 20, n - s)
        idx = torch.randint(0, centres, (m,), generator=g, device=dev)
        x = cen[idx] + spread * torch.randn((m, D), generator=g, device=dev) / (D ** 0.5)
        out[s:s + m] = (x / x.norm(dim=1, keepdim=True)).cpu().numpy()
        CLUSTER_OF.setdefault(seed, []).append(idx.cpu().numpy())
    return out


if os.environ.get("KIND", "clustered") == "sift":      # bench.py's harder SIFT-like set (256 blobs, sigma 40): byte-valued, M 16
    import bench
    X = bench.make_sift_like(N, D, 1, dev, 256, 40.0).cpu().numpy()
    Q = bench.make_sift_like(NQ, D, 2, dev, 256, 40.0).cpu().numpy()
    os.environ["BLOCKS_IDEAL"] = "0"
else:
    X = clustered(N, 12)
    Q = clustered(NQ, 112)
hg = H.Ohnsw.build_batch_bigarray(X, M, 200, seed=1, metric=METRIC)
hg.export()
deg0, nbr0 = hg.deg0, hg.nbr0
ids, dist, nd, nh = H.Ohnsw.knn_batch_bigarray(hg, 10, Q, ef=EF, counters=True)
print("n %d d %d M %d ef %d metric %d; device: %.0f evaluations, %.0f hops per query" % (N, D, M, EF, METRIC, nd.mean(), nh.mean()), flush=True)


def nearest_hub(nodes, hubs):
    """index into hubs of the nearest hub of every node (exact, on the GPU)"""
    Hm = torch.from_numpy(X[hubs]).to(dev)
    hn = (Hm * Hm).sum(1)
    out = np.empty(len(nodes), np.int64)
    for s0 in range(0, len(nodes), 1 << 18):
        xs = torch.from_numpy(X[nodes[s0:s0 + (1 << 18)]]).to(dev)
        sc = xs @ Hm.T
        key = sc if METRIC else (2.0 * sc - hn[None, :])
        out[s0:s0 + (1 << 18)] = key.argmax(1).cpu().numpy()
    return out


def locality_codes():
    """L(v): a bijection node -> position in an order that keeps graph-close nodes together, derived from the index's OWN
    upper layers: nodes ordered by their nearest layer-2 node, those by their nearest layer-3 node, and so on"""
    hg.export()
    layers = [np.asarray(u[0], np.int64) for u in hg.upper]          # layers[l - 1] = nodes of layer l
    top = len(layers)
    lo = int(os.environ.get("HUB_LAYER", 2))
    rank_of = {int(v): i for i, v in enumerate(layers[top - 1])}       # top layer: as listed
    for l in range(top - 1, lo - 1, -1):                               # order layer l's nodes by their nearest node of layer l + 1
        nodes, hubs = layers[l - 1], layers[l]
        a = nearest_hub(nodes, hubs)
        key = np.array([rank_of[int(hubs[i])] for i in a], np.int64)
        order = np.lexsort((nodes, key))
        rank_of = {int(nodes[i]): r for r, i in enumerate(order)}
    hubs = layers[lo - 1]
    alln = np.arange(N, dtype=np.int64)
    a = nearest_hub(alln, hubs)
    key = np.array([rank_of[int(h)] for h in hubs], np.int64)[a]
    order = np.lexsort((alln, key))
    L = np.empty(N, np.int64)
    L[order] = np.arange(N)
    return L


class Blocks:
    """visited as a cache of BITMAP BLOCKS over a locality code: block = L >> bs (tag), bit = L & (2^bs - 1); nb blocks
    (LDS: nb * (2^bs / 8 + 2) bytes), least recently used block replaced -- exact inside the cached blocks"""
    def __init__(self, L, bs, nb, lru=True):
        self.L, self.bs, self.nb, self.lru = L, bs, nb, lru
        self.blocks = {}          # insertion-ordered: first = oldest
        self.evals = 0

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        c = int(self.L[node])
        b, bit = c >> self.bs, c & ((1 << self.bs) - 1)
        blk = self.blocks.get(b)
        if blk is not None:
            if self.lru:
                del self.blocks[b]
                self.blocks[b] = blk
            if bit in blk:
                return
            blk.add(bit)
            self.evals += 1
            return
        self.evals += 1
        if len(self.blocks) >= self.nb:
            del self.blocks[next(iter(self.blocks))]
        self.blocks[b] = {bit}


class BlocksDM:
    """the same with a DIRECT-MAPPED directory: block b lives in slot b mod nb only (what one LDS lookup per lane can do);
    ways = 2: two slots per set, the older one replaced"""
    def __init__(self, L, bs, nb, ways=1, lru=False):
        self.L, self.bs, self.nb, self.ways, self.lru = L, bs, nb // ways, ways, lru
        self.slots = {}
        self.evals = 0
        self.touched = set()

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        c = int(self.L[node])
        b, bit = c >> self.bs, c & ((1 << self.bs) - 1)
        ws = self.slots.setdefault(b % self.nb, [])
        self.touched.add(b)
        for e in ws:
            if e[0] == b:
                if self.lru and ws[0] is not e:
                    ws.remove(e); ws.insert(0, e)
                if bit in e[1]:
                    return
                e[1].add(bit)
                self.evals += 1
                return
        self.evals += 1
        ws.insert(0, (b, {bit}))
        del ws[self.ways:]


class HW:
    """what the kernel would do: `sets` sets x `ways` ways of (block tag, 8-bit stamp) + one bitmap of 2^bs codes per way; the 64 lanes
    of a hop look up the state the hop started with, a lane that misses claims the set's least recently touched way
    (stamp = hop >> 1, ages modulo 256), the last writer of a way owns it, owners clear the bitmap, then bits are set"""
    def __init__(self, L, bs, sets, ways, shift=1, alt=False):
        self.L, self.bs, self.ns, self.ways, self.shift, self.alt = L, bs, sets, ways, shift, alt
        self.tag = [[-1] * ways for _ in range(sets)]
        self.stamp = [[0] * ways for _ in range(sets)]
        self.bits = [[set() for _ in range(ways)] for _ in range(sets)]
        self.evals = 0
        self.cur = None
        self.buf = []

    def flush(self):
        if not self.buf:
            return
        now = (self.cur >> self.shift) & 255
        lanes = []
        for node in self.buf:
            c = int(self.L[node]); b, bit = c >> self.bs, c & ((1 << self.bs) - 1)
            s_ = b % self.ns
            w = self.tag[s_].index(b) if b in self.tag[s_] else -1
            vis = w >= 0 and bit in self.bits[s_][w]
            if w < 0:
                ages = [((now - st) & 255) if t >= 0 else 1000 for t, st in zip(self.tag[s_], self.stamp[s_])]
                w2 = ages.index(max(ages))
                if self.alt and (b >> 5) & 1:            # every other block claims the SECOND oldest way: two new blocks of one set in one hop both get a slot
                    a2 = list(ages); a2[w2] = -1
                    w2 = a2.index(max(a2))
            else:
                w2 = w
            lanes.append((b, bit, s_, w, w2, vis))
        for b, bit, s_, w, w2, vis in lanes:          # one write instruction: the last lane wins a contested way
            newtag = (self.tag[s_][w2] != b)
            self.tag[s_][w2] = b; self.stamp[s_][w2] = now
        cleared = set()
        for b, bit, s_, w, w2, vis in lanes:
            if self.tag[s_][w2] == b and w < 0 and (s_, w2) not in cleared:
                self.bits[s_][w2] = set(); cleared.add((s_, w2))
        for b, bit, s_, w, w2, vis in lanes:
            if not vis:
                self.evals += 1
                if self.tag[s_][w2] == b:
                    self.bits[s_][w2].add(bit)
        self.buf = []

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        if hop != self.cur:
            self.flush()
            self.cur = hop
        self.buf.append(node)


def dist_to(q, rows):
    v = X[rows] @ q
    return (1.0 - v) if METRIC else ((X[rows] - q) ** 2).sum(1)


def record(q, start):
    """the exact walk -> list of encounters (node, fresh, accepted, in_w)"""
    d0 = float(dist_to(q, np.array([start]))[0])
    visited = {start: d0}
    inw = {start}
    cand = [(d0, start)]
    W = [(-d0, start)]
    tr = [(start, True, True, False, d0, d0, 0)]
    hop = 0
    while cand:
        dc, c_ = heapq.heappop(cand)
        if len(W) >= EF and dc > -W[0][0]:
            break
        row = nbr0[c_, :deg0[c_]]
        row = [int(x) for x in row if x >= 0]
        hop += 1
        fresh = [nb for nb in row if nb not in visited]
        dd = dist_to(q, np.array(fresh)) if fresh else []
        dmap = {nb: float(dn) for nb, dn in zip(fresh, dd)}
        for nb in row:
            if nb in visited:
                tr.append((nb, False, False, nb in inw, visited[nb], -W[0][0] if len(W) >= EF else float("inf"), hop))
                continue
            dn = dmap[nb]
            visited[nb] = dn
            acc = len(W) < EF or dn < -W[0][0]
            tr.append((nb, True, acc, False, dn, -W[0][0] if len(W) >= EF else float("inf"), hop))
            if acc:
                heapq.heappush(cand, (dn, nb))
                heapq.heappush(W, (-dn, nb))
                inw.add(nb)
                if len(W) > EF:
                    inw.discard(heapq.heappop(W)[1])
    return tr, visited


class Fifo:
    def __init__(self, sb, ways, wcheck=False):
        self.sb, self.ways, self.wcheck = sb, ways, wcheck
        self.sets = {}
        self.evals = 0

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        if self.wcheck and in_w:
            return
        ws = self.sets.setdefault(node & ((1 << self.sb) - 1), [])
        tag = node >> self.sb
        if tag in ws:
            return
        self.evals += 1
        ws.insert(0, tag)
        del ws[self.ways:]


class Keep:
    """insert after the evaluation: accepted -> way 0, rejected -> last way only"""
    def __init__(self, sb, ways, strict=False, promote=False):
        self.sb, self.ways, self.strict, self.promote = sb, ways, strict, promote
        self.sets = {}
        self.evals = 0

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        ws = self.sets.setdefault(node & ((1 << self.sb) - 1), [])
        tag = node >> self.sb
        for e in ws:
            if e[0] == tag:
                return
        self.evals += 1
        # a re-evaluated node: "accepted" is what the kernel would see -- d below the current max(W) (it is then found in W
        # by the rank step) -- the trace only says so for first evaluations; for a repeat use membership of W
        a = acc if fresh else in_w
        if a:
            ws.insert(0, (tag, True))
            del ws[self.ways:]
        else:
            if len(ws) < self.ways:
                ws.append((tag, False))
            elif not (self.strict and ws[-1][1]):
                ws[-1] = (tag, False)


class Thresh:
    """insert after the evaluation, and only nodes nearer than max(W) + alpha * (max(W) - min seen): far nodes are never
    remembered (they are met again less often), so the tags of the near ones live longer; fifo among what is inserted"""
    def __init__(self, sb, ways, alpha):
        self.sb, self.ways, self.alpha = sb, ways, alpha
        self.sets = {}
        self.evals = 0
        self.dmin = float("inf")

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        ws = self.sets.setdefault(node & ((1 << self.sb) - 1), [])
        tag = node >> self.sb
        if tag in ws:
            return
        self.evals += 1
        self.dmin = min(self.dmin, d)
        if wmax == float("inf") or d < wmax + self.alpha * (wmax - self.dmin):
            ws.insert(0, tag)
            del ws[self.ways:]


class Opt:
    """Belady with bypass, fully associative"""
    def __init__(self, cap, nxt):
        self.cap, self.nxt = cap, nxt
        self.inc = {}        # node -> next use
        self.heap = []       # (-next use, node)
        self.evals = 0
        self.i = 0

    def see(self, node, fresh, acc, in_w, d, wmax, hop=0):
        i = self.i
        self.i += 1
        nu = self.nxt[i]
        if node in self.inc:
            self.inc[node] = nu
            heapq.heappush(self.heap, (-nu, node))
            return
        self.evals += 1
        if nu >= 1 << 60:
            return                                   # never met again: bypass
        while len(self.inc) >= self.cap:
            far, v = self.heap[0]
            if self.inc.get(v) != -far:
                heapq.heappop(self.heap)
                continue
            if -far <= nu:
                return                               # everything cached is needed sooner: bypass
            heapq.heappop(self.heap)
            del self.inc[v]
        self.inc[node] = nu
        heapq.heappush(self.heap, (-nu, node))


tag_bits = max(1, (N - 1).bit_length())
SB = [int(x) for x in os.environ.get("SB", "11,12").split(",")]
names = []
for sb in SB:
    w = 3 if tag_bits - sb <= 10 else 2
    for ways in sorted({2, w}):
        names += [("fifo", sb, ways), ("fifo+W", sb, ways), ("keep", sb, ways), ("keep1", sb, ways), ("opt", sb, ways)]
        names += [("thr%.2f" % a, sb, ways) for a in (0.0, 0.1, 0.2, 0.3, 0.5, 0.8, 1.2)]
LCODES = {}
if os.environ.get("BLOCKS", "1") != "0":
    if os.environ.get("BLOCKS_IDEAL", "1") != "0":
        cl = np.concatenate(CLUSTER_OF[12])
        order = np.lexsort((np.arange(N), cl))
        Lc = np.empty(N, np.int64); Lc[order] = np.arange(N)
        LCODES["ideal"] = Lc                              # the generator's own cluster index: what a perfect order could do
    if os.environ.get("BLOCKS_HUBS", "1") != "0":
        LCODES["hubs"] = locality_codes()
    LCODES["device"] = hg.locality_codes().astype(np.int64)        # what the library derives (greedy descents instead of exact nearest hubs)
    assert np.array_equal(np.sort(LCODES["device"]), np.arange(N)), "the device's codes are not a permutation"
    LCODES["ident"] = np.arange(N, dtype=np.int64)        # the ids as they are (insertion order: no locality)
    for lname in LCODES:
        for bs, nb in ((7, 448), (8, 240), (9, 124), (8, 480)):
            names.append(("blk-" + lname, bs, nb))
        if lname in ("hubs", "device"):
            for bs, nb in ((7, 409), (8, 227)):
                for w in (8, 16):
                    names.append(("dl%d-" % w + lname, bs, nb))
            for bs, sets, ways in ((8, 16, 16), (8, 32, 8), (7, 25, 16), (7, 32, 16), (8, 8, 16), (8, 8, 32), (7, 16, 16), (8, 8, 8), (8, 16, 8), (7, 16, 8), (6, 32, 8)):
                names.append(("hw%d-" % sets + lname, bs, ways))
            names.append(("hx32-" + lname, 8, 8))        # 32 sets x 8 ways with a stamp per FOUR hops (1024 hops before it wraps)
            names.append(("hy32-" + lname, 8, 8))        # 32 sets x 8 ways, every other block claims the second oldest way
tot = {nm: 0 for nm in names}
exact = 0
enc = 0
rep_in_w = 0
rep = 0
n_acc = 0
rank_hist = [0] * 40
again_hist = [0] * 40
all_hist = [0] * 40
for qi in range(NQ):
    tr, vis = record(Q[qi], int(ids[qi, 0]))
    last = {}
    nxt = [1 << 60] * len(tr)
    for i in range(len(tr) - 1, -1, -1):
        nxt[i] = last.get(tr[i][0], 1 << 60)
        last[tr[i][0]] = i
    pol = {}
    for nm in names:
        kind, sb, ways = nm
        pol[nm] = (Fifo(sb, ways) if kind == "fifo" else Fifo(sb, ways, True) if kind == "fifo+W" else
                   Keep(sb, ways) if kind == "keep" else Keep(sb, ways, strict=True) if kind == "keep1" else
                   Thresh(sb, ways, float(kind[3:])) if kind.startswith("thr") else
                   Blocks(LCODES[kind[4:]], sb, ways) if kind.startswith("blk-") else
                   BlocksDM(LCODES[kind.split("-")[1]], sb, ways, int(kind.split("-")[0][2:]), kind[1] == "l") if kind[0] == "d" else
                   HW(LCODES[kind.split("-")[1]], sb, int(kind.split("-")[0][2:]), ways) if kind.startswith("hw") else
                   HW(LCODES[kind.split("-")[1]], sb, 32, ways, shift=2) if kind.startswith("hx") else
                   HW(LCODES[kind.split("-")[1]], sb, 32, ways, alt=True) if kind.startswith("hy") else Opt(ways << sb, nxt))
    for e in tr:
        for p in pol.values():
            p.see(*e)
    for nm in names:
        if nm[0][:2] in ("hw", "hx", "hy"):
            pol[nm].flush()
        tot[nm] += pol[nm].evals
    exact += sum(1 for e in tr if e[1])
    order = sorted(vis, key=lambda v: vis[v])
    rank = {v: i for i, v in enumerate(order)}
    for e in tr:
        if not e[1]:
            rank_hist[min(rank[e[0]] // 1000, 39)] += 1
    for v in set(e[0] for e in tr if not e[1]):
        again_hist[min(rank[v] // 1000, 39)] += 1
    for v in vis:
        all_hist[min(rank[v] // 1000, 39)] += 1
    enc += len(tr)
    rep += sum(1 for e in tr if not e[1])
    rep_in_w += sum(1 for e in tr if not e[1] and e[3])
    n_acc += sum(1 for e in tr if e[2])
    print("query %d: %d encounters, %d nodes" % (qi, len(tr), len(vis)), flush=True)
print("exact Visited: %.0f evaluations per query; %.0f encounters, %.0f of them repeats (%.0f %% of the repeats are members of W at that moment); "
      "%.0f accepted by W" % (exact / NQ, enc / NQ, rep / NQ, 100.0 * rep_in_w / max(rep, 1), n_acc / NQ))
print("by final distance rank of the node (thousands): repeats per query / nodes met again / nodes")
for i in range(40):
    if all_hist[i]:
        print("  rank %2dk: %7.0f %7.0f %7.0f" % (i, rank_hist[i] / NQ, again_hist[i] / NQ, all_hist[i] / NQ))
for nm in names:
    if nm[0][:2] in ("hw", "hx", "hy"):
        sets = int(nm[0].split("-")[0][2:])
        print("%-10s %d sets x %d ways, blocks of 2^%d codes (%d B): %.0f evaluations per query (+%.1f %%)" % (nm[0], sets, nm[2], nm[1], sets * nm[2] * ((1 << nm[1]) // 8 + 4), tot[nm] / NQ, 100.0 * (tot[nm] / max(exact, 1) - 1)))
        continue
    if nm[0].startswith("blk-") or nm[0][0] == "d":
        print("%-10s blocks of 2^%d codes x %d blocks (%d B): %.0f evaluations per query (+%.1f %%)" % (nm[0], nm[1], nm[2], nm[2] * ((1 << nm[1]) // 8 + 2), tot[nm] / NQ, 100.0 * (tot[nm] / max(exact, 1) - 1)))
        continue
    print("%-7s 2^%d sets x %d ways: %.0f evaluations per query (+%.1f %%)" % (nm[0], nm[1], nm[2], tot[nm] / NQ, 100.0 * (tot[nm] / max(exact, 1) - 1)))
