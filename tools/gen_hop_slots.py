#!/usr/bin/env python3
"""Writes ocaml-hnsw_amd/csrc/hnsw_hop_slots8.inc: the insertion of the hand-scheduled layer-0 loop for W in EIGHT key
registers per lane (ef 257..512), i.e. HNSW_INSERT_LOOP8 / HNSW_INSERT_RARE8 -- the text of HNSW_INSERT_LOOP4 /
HNSW_INSERT_RARE4 (csrc/hnsw_hop_asm.hip.h, written by hand) with the slot-dependent parts repeated per slot: the chain over
the slots' maxima that finds the rank's slot, one block per slot (tie check, position, the cascade of every slot above it,
the shift, the maxima that moved), the general rank over all slots and one eviction-tie block per way back.

    python tools/gen_hop_slots.py            # rewrite the file
    python tools/gen_hop_slots.py --check    # exit 1 if the committed file differs from what this script writes
    python tools/gen_hop_slots.py --n 4      # print the same text for four slots (compared with the hand-written macros by
                                             # tests/test_asm_hazards.py::test_generated_insertion_equals_the_hand_written_one)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "ocaml-hnsw_amd", "csrc", "hnsw_hop_slots8.inc")
DPP = '" HNSW_DPP_ALL "'


def lines_loop(n):
    t = n - 1
    L = []
    a = L.append
    a('10:')
    a('\ts_cmp_eq_u64 %[fresh], 0')
    a('\ts_cbranch_scc1 19f')
    a('110:')
    a('\ts_ff1_i32_b64 %[i], %[fresh]')
    a('\tv_readlane_b32 %[kd], %[ckey], %[i]')
    a('\tv_readlane_b32 %[klo], %[cid], %[i]')
    a('\tv_readlane_b32 %%[nw], %%[h%d], 62' % t)
    a('\ts_cmp_ge_u32 %[kd], %[wmax]')
    a('\ts_cbranch_scc1 @REJECT@')                       # no longer below max(W): rejected, :574
    a('\ts_cmp_lt_u32 %%[mx%d], %%[kd]' % (t - 1))       # the rank's slot, from the slots' maxima
    a('\ts_cbranch_scc0 8%df' % (t - 1))
    L += slot_head(t)
    a('8%d1:' % t)
    L += slot_shift(t, t)
    a('\ts_mov_b32 %[wmax], %[nw]')
    a('18:')
    a('\ts_bitset0_b64 %[fresh], %[i]')
    a('\ts_cmp_lg_u64 %[fresh], 0')
    a('\ts_cbranch_scc1 110b')                           # the next accepted candidate
    a('19:')
    return L


def slot_head(k):
    """tie check and position inside slot k, the new maximum, the eviction tie"""
    return ['\tv_cmp_eq_u32_e32 vcc, %%[kd], %%[h%d]' % k,
            '\tv_cmp_gt_u32_e64 %%[um0], %%[kd], %%[h%d]' % k,
            '\ts_cmp_lg_u64 vcc, 0',
            '\ts_cbranch_scc1 14f',                      # a member of W at this very distance
            '\ts_bcnt1_i32_b64 m0, %[um0]',              # rank inside the slot
            '\ts_max_u32 %[nw], %[nw], %[kd]',           # the new max(W).d (this key if it ranks last)
            '\ts_cmp_eq_u32 %[nw], %[wmax]',
            '\ts_cbranch_scc1 15%df' % k]                # the entry falling off ties with it


def slot_shift(k, t):
    """slots above k move whole and take their lower neighbour's last key; slot k moves from m0 on; the key lands"""
    L = []
    for j in range(t - 1, k - 1, -1):
        L += ['\tv_readlane_b32 %%[sx], %%[h%d], 63' % j,
              '\tv_readlane_b32 %%[tmp], %%[l%d], 63' % j,
              '\tv_mov_b32_dpp %%[h%d], %%[h%d] wave_shr:1%s' % (j + 1, j + 1, DPP),
              '\tv_mov_b32_dpp %%[l%d], %%[l%d] wave_shr:1%s' % (j + 1, j + 1, DPP),
              '\tv_writelane_b32 %%[h%d], %%[sx], 0' % (j + 1),
              '\tv_writelane_b32 %%[l%d], %%[tmp], 0' % (j + 1)]
    L += ['\ts_bfm_b64 vcc, m0, 0',                      # lanes below m0 keep their keys
          '\tv_cndmask_b32_dpp %%[h%d], %%[h%d], %%[h%d], vcc wave_shr:1%s' % (k, k, k, DPP),
          '\tv_cndmask_b32_dpp %%[l%d], %%[l%d], %%[l%d], vcc wave_shr:1%s' % (k, k, k, DPP),
          '\tv_writelane_b32 %%[h%d], %%[kd], m0' % k,
          '\tv_writelane_b32 %%[l%d], %%[klo], m0' % k]
    return L


def lines_rare(n):
    t = n - 1
    L = []
    a = L.append
    for k in range(t - 1, -1, -1):
        a('8%d:' % k)
        if k > 0:
            a('\ts_cmp_lt_u32 %%[mx%d], %%[kd]' % (k - 1))
            a('\ts_cbranch_scc0 8%df' % (k - 1))
        L += slot_head(k)
        a('8%d1:' % k)
        L += slot_shift(k, t)
        for j in range(k, t):
            a('\tv_readlane_b32 %%[mx%d], %%[h%d], 63' % (j, j))
        a('\ts_mov_b32 %[wmax], %[nw]')
        a('\ts_bitset0_b64 %[fresh], %[i]')
        a('\ts_cmp_lg_u64 %[fresh], 0')
        a('\ts_cbranch_scc1 110b')
        a('\ts_branch 19b')
    # a member of W at exactly this distance: rank over all slots, ids decide; the node itself in W: ignored
    a('14:')
    a('\ts_mov_b32 %[P], 0')
    a('\ts_mov_b64 %[um1], 0')
    for s in range(n):
        a('@RANK%d@' % s)
    a('\ts_cmp_lg_u64 %[um1], 0')
    a('\ts_cbranch_scc1 18b')                            # already in W
    a('\ts_max_u32 %[nw], %[nw], %[kd]')
    a('\ts_cmp_eq_u32 %[nw], %[wmax]')
    a('\ts_cbranch_scc1 15%df' % n)
    a('141:')
    a('\ts_and_b32 m0, %[P], 63')
    for k in range(t, 0, -1):
        a('\ts_cmp_ge_u32 %%[P], %d' % (64 * k))
        a('\ts_cbranch_scc1 8%d1b' % k)
    a('\ts_branch 801b')
    for k in range(t, -1, -1):
        a('@EVICT:15%d:8%d1b:16%d@' % (k, k, k))
    a('@EVICT:15%d:141b:16%d@' % (n, n))
    return L


def emit(name, L, n):
    out = ['#define %s \\' % name]
    for i, ln in enumerate(L):
        last = i == len(L) - 1
        if ln.startswith('@RANK'):
            s = int(ln[5:-1])
            txt = '    HNSW_RANK_GENERAL_SLOT("%%[h%d]", "%%[l%d]")' % (s, s)
        elif ln.startswith('@EVICT'):
            _, lbl, back, full = ln.strip('@').split(':')
            txt = '    HNSW_EVICT_TIE("%s", "%s", "%%[l%d]", "%s")' % (lbl, back, n - 1, full)
        elif ln.endswith(':'):
            txt = '    "%s\\n\\t"' % ln if False else '    "%s\\n"' % ln
        else:
            body = ln.strip('\t').replace('@REJECT@', '" HNSW_SEM_REJECT "')
            # a label follows: the line ends with \n only (the hand-written text does the same; cosmetic)
            nxt = L[i + 1] if not last else ''
            end = '\\n' if (nxt.endswith(':') and not nxt.startswith('@')) else '\\n\\t'
            txt = '    "%s%s"' % (body, end)
        out.append(txt + ('' if last else ' \\'))
    return '\n'.join(out)


def text(n):
    head = ('// hnsw_hop_slots%d.inc -- GENERATED by tools/gen_hop_slots.py (do not edit; `python tools/gen_hop_slots.py` rewrites it,\n'
            '// tests/test_asm_hazards.py checks that it is current): HNSW_INSERT_LOOP%d / HNSW_INSERT_RARE%d, the insertion of the\n'
            '// hand-scheduled layer-0 loop for W in %d key registers per lane.  Same structure as the four-slot text in\n'
            '// hnsw_hop_asm.hip.h (which explains it): the slot of the rank from the slots\' maxima (mx0..mx%d, scalar registers), one block\n'
            '// per slot, the cascade of the slots above it, the general rank for distance ties, one eviction-tie block per way back.\n' % (n, n, n, n, n - 2))
    return head + emit('HNSW_INSERT_LOOP%d' % n, lines_loop(n), n) + '\n\n' + emit('HNSW_INSERT_RARE%d' % n, lines_rare(n), n) + '\n'


def main(argv):
    n = 8
    if '--n' in argv:
        n = int(argv[argv.index('--n') + 1])
        sys.stdout.write(text(n))
        return 0
    t = text(8)
    if '--check' in argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ''
        if cur != t:
            print('%s is not what tools/gen_hop_slots.py writes' % OUT)
            return 1
        return 0
    with open(OUT, 'w') as f:
        f.write(t)
    print('wrote', OUT)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
