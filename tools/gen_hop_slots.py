#!/usr/bin/env python3
"""Writes the two GENERATED includes of the hand-scheduled layer-0 loop (ocaml-hnsw_amd/csrc/hnsw_hop_asm.hip.h):

  hnsw_hop_slots.inc      everything in hnsw_hop_loop.inc that depends on HOW MANY key registers per lane hold W, as one macro set
                          per slot count N in SLOTS = 3, 4, 6, 8 (ef 129..192 / 193..256 / 257..384 / 385..512): the insertion
                          (HNSW_INSERT_LOOP<N> / HNSW_INSERT_RARE<N> -- the text of the hand-written four-slot macros of
                          hnsw_hop_asm.hip.h with the slot-dependent parts repeated per slot: the chain over the slots' maxima that
                          finds the rank's slot, one block per slot (tie check, position, the cascade of every slot above it, the
                          shift, the maxima that moved), the general rank over all slots and one eviction-tie block per way back),
                          the pop / peek chains over the slots' unexpanded masks, the register declarations and operand lists.
                          hnsw_hop_loop.inc has ONE body for all of them (HNSW_NS(...) picks the set of HNSW_LOOP_NSLOT); the
                          four-slot insertion stays the hand-written, commented text and the generator must reproduce it.
  hnsw_hop_instances.inc  the table of instantiations: one `#define HNSW_LOOP_* ... #include "hnsw_hop_loop.inc"` stanza per
                          (row family, metric, accept rule, slot count, visited structure), each under the feature switches of
                          hnsw_hop_asm.hip.h and -- in the translation units of hnsw_search_variants.hip, which are compiled per
                          (metric, rule, row format) -- only where the unit can reach it (rounds 2-5 wrote 184 stanzas by hand and
                          every unit parsed all of them).

    python tools/gen_hop_slots.py            # rewrite both files
    python tools/gen_hop_slots.py --check    # exit 1 if a committed file differs from what this script writes
    python tools/gen_hop_slots.py --n 4      # print the insertion for four slots (compared with the hand-written macros by
                                             # tests/test_asm_hazards.py::test_generated_insertion_is_current_and_equals_the_hand_written_one)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ocaml-hnsw_amd", "csrc")
OUT_SLOTS = os.path.join(CSRC, "hnsw_hop_slots.inc")
OUT_INST = os.path.join(CSRC, "hnsw_hop_instances.inc")
SLOTS = (3, 4, 6, 8)            # slot counts served by the generic body of hnsw_hop_loop.inc (1 and 2 have bodies of their own)
HAND_WRITTEN = (4,)             # ... whose insertion is the hand-written text of hnsw_hop_asm.hip.h
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
    """HNSW_INSERT_LOOP<n> / HNSW_INSERT_RARE<n>"""
    return emit('HNSW_INSERT_LOOP%d' % n, lines_loop(n), n) + '\n\n' + emit('HNSW_INSERT_RARE%d' % n, lines_rare(n), n) + '\n'


def macro(name, lines):
    """a multi-line #define from already quoted / C text lines"""
    if not lines:
        return '#define %s' % name
    return '#define %s \\\n' % name + ' \\\n'.join('    ' + ln for ln in lines)


def slot_set(n):
    """the macro set HNSW_NSX_<WHAT>_<n> that hnsw_hop_loop.inc's generic body is made of"""
    t = n - 1
    out = []
    q = lambda txt: '"%s"' % txt
    out.append(macro('HNSW_NSX_DECL_%d' % n,
                     ['uint64_t %s;' % ', '.join('um%d' % j for j in range(n)),
                      'uint32_t %s;' % ', '.join('mx%d' % j for j in range(t))]))
    out.append(macro('HNSW_NSX_ALIGN_IN_%d' % n, [q('v_alignbit_b32 %%[l%d], %%[l%d], %%[l%d], 1\\n\\t' % (j, j, j)) for j in range(n)]))
    out.append(macro('HNSW_NSX_ALIGN_OUT_%d' % n, [q('\\n\\tv_alignbit_b32 %%[l%d], %%[l%d], %%[l%d], 31' % (j, j, j)) for j in range(n)]))
    out.append(macro('HNSW_NSX_MX_INIT_%d' % n, [q('v_readlane_b32 %%[mx%d], %%[h%d], 63\\n\\t' % (j, j)) for j in range(t)]))
    out.append(macro('HNSW_NSX_UM_INIT_%d' % n, [q('v_cmp_lt_i32_e64 %%[um%d], -1, %%[l%d]\\n%s' % (j, j, '' if j == t else '\\t')) for j in range(n)]))
    out.append(macro('HNSW_NSX_POP_CHAIN_%d' % n,
                     ['HNSW_POP_SLOT("6%d", "%%[um%d]", "%%[l%d]", "%s")' % (j, j, j, '90f' if j == t else '6%df' % (j + 1)) for j in range(1, n)]))
    out.append(macro('HNSW_NSX_PEEK_CHAIN_%d' % n,
                     ['HNSW_PEEK_SLOT("7%d", "%%[um%d]", "%%[l%d]", "%s")' % (j, j, j, '9b' if j == t else '7%df' % (j + 1)) for j in range(1, n)]))
    out.append(macro('HNSW_NSX_WOUT_%d' % n,
                     [', '.join('[h%d] "+&v"(w.hi[%d])' % (j, j) for j in range(n)) + ',',
                      ', '.join('[l%d] "+&v"(w.lo[%d])' % (j, j) for j in range(n)) + ',']))
    out.append(macro('HNSW_NSX_UMOUT_%d' % n, [', '.join('[um%d] "=&s"(um%d)' % (j, j) for j in range(n)) + ',']))
    out.append(macro('HNSW_NSX_MXOUT_%d' % n, [', ' + ', '.join('[mx%d] "=&s"(mx%d)' % (j, j) for j in range(t))]))
    out.append('#define HNSW_NSX_INSERT_LOOP_%d HNSW_INSERT_LOOP%d' % (n, n))
    out.append('#define HNSW_NSX_INSERT_RARE_%d HNSW_INSERT_RARE%d' % (n, n))
    out.append('#define HNSW_NSX_ALIGN_PAD_%d HNSW_ASM_ALIGN_PAD%d' % (n, n))
    return '\n'.join(out) + '\n'


def slots_text():
    head = ('// hnsw_hop_slots.inc -- GENERATED by tools/gen_hop_slots.py (do not edit; `python tools/gen_hop_slots.py` rewrites it,\n'
            '// tests/test_asm_hazards.py checks that it is current).  For every slot count N in %s -- key registers per lane that hold W --\n'
            '// the parts of the hand-scheduled layer-0 loop that depend on N: HNSW_INSERT_LOOP<N> / HNSW_INSERT_RARE<N> (same structure as\n'
            '// the four-slot text in hnsw_hop_asm.hip.h, which explains it: the slot of the rank from the slots\' maxima mx0..mx<N-2>, one\n'
            '// block per slot, the cascade of the slots above it, the general rank for distance ties, one eviction-tie block per way back;\n'
            '// N = %s: hand-written there) and the macro set HNSW_NSX_*_<N> the generic body of hnsw_hop_loop.inc is made of.\n'
            % (', '.join(map(str, SLOTS)), ', '.join(map(str, HAND_WRITTEN))))
    parts = [head]
    for n in SLOTS:
        parts.append('// ---- W in %d key registers per lane ----' % n)
        if n not in HAND_WRITTEN:
            parts.append(text(n))
        parts.append(slot_set(n))
    return '\n'.join(parts)


# ---- the instantiation table ----------------------------------------------------------------------------------------------
# row families: (name, NCH, ROWS values, feature guard)
FAMILIES = (("bytes", 2, (2,), "1"),
            ("bytes4", 4, (2,), "HNSW_ASM_LOOP_BYTES4"),
            ("f32", 2, (1, 0, 3), "HNSW_ASM_LOOP_F32"),
            ("f32n4", 4, (1, 0, 3), "HNSW_ASM_LOOP_F32 && HNSW_ASM_LOOP_F32N4"))
NSLOTS = (1, 2, 3, 4, 6, 8)


def instances():
    """(nch, rows, metric, sem, nslot, blk, guard) of every instantiation"""
    out = []
    for fam, nch, rows_list, fguard in FAMILIES:
        for rows in rows_list:
            for metric in (0, 1):
                for sem in (0, 1):
                    for nslot in NSLOTS:
                        for blk in (0, 1):
                            if blk and (nslot < 3 or fam == "bytes4"):
                                continue            # bitmap blocks: W in three or more registers; not for byte rows of 129..256 dimensions
                            g = [fguard]
                            if rows == 3:
                                g.append("HNSW_ASM_LOOP_SPLIT")
                            if sem:
                                g.append("HNSW_ASM_LOOP_SEM1")
                            if nslot > 4:
                                g.append("HNSW_ASM_LOOP_8SLOTS")
                            g = [x for x in g if x != "1"]
                            out.append((nch, rows, metric, sem, nslot, blk, " && ".join(g) if g else "1"))
    return out


def instances_text():
    L = ['// hnsw_hop_instances.inc -- GENERATED by tools/gen_hop_slots.py (do not edit; `python tools/gen_hop_slots.py` rewrites it,',
         '// tests/test_asm_hazards.py checks that it is current): the instantiations HopLoop<NCH, NSLOT, METRIC, ROWS, SEM, BLK> of the',
         '// hand-scheduled layer-0 loop, one stanza each.  A stanza is compiled when the feature switches of hnsw_hop_asm.hip.h allow it',
         '// and the translation unit can reach it: the units of hnsw_search_variants.hip are compiled per (metric, accept rule, row',
         '// format) and define HNSW_V_METRIC / HNSW_V_SEMF / HNSW_V_FULL; every other unit (the builder, the layer operators: row format',
         '// decided at run time) takes search_layer\'s C++ loop and instantiates nothing.',
         '#if defined(HNSW_HOP_ALL_INSTANCES)      /* (tests: every instantiation in one preprocessed unit) */',
         '#define HNSW_HOP_UNIT(M, S, R) 1',
         '#elif defined(HNSW_V_METRIC)',
         '#define HNSW_HOP_UNIT(M, S, R) (HNSW_V_METRIC == (M) && HNSW_V_SEMF == (S) && HNSW_V_FULL == (R))',
         '#else',
         '#define HNSW_HOP_UNIT(M, S, R) 0',
         '#endif']
    for nch, rows, metric, sem, nslot, blk, guard in instances():
        cond = 'HNSW_HOP_UNIT(%d, %d, %d)' % (metric, sem, rows) + ('' if guard == '1' else ' && ' + guard)
        L.append('#if %s' % cond)
        L.append('#define HNSW_LOOP_NCH %d' % nch)
        L.append('#define HNSW_LOOP_NSLOT %d' % nslot)
        L.append('#define HNSW_LOOP_ROWS %d' % rows)
        L.append('#define HNSW_LOOP_METRIC %d' % metric)
        L.append('#define HNSW_LOOP_SEM %d' % sem)
        L.append('#define HNSW_LOOP_BLK %d' % blk)
        L.append('#include "hnsw_hop_loop.inc"')
        L.append('#endif')
    L.append('#undef HNSW_HOP_UNIT')
    return '\n'.join(L) + '\n'


def main(argv):
    if '--n' in argv:
        n = int(argv[argv.index('--n') + 1])
        sys.stdout.write(text(n))
        return 0
    want = {OUT_SLOTS: slots_text(), OUT_INST: instances_text()}
    if '--check' in argv:
        rc = 0
        for path, t in want.items():
            cur = open(path).read() if os.path.exists(path) else ''
            if cur != t:
                print('%s is not what tools/gen_hop_slots.py writes' % path)
                rc = 1
        return rc
    for path, t in want.items():
        with open(path, 'w') as f:
            f.write(t)
        print('wrote', path)
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv[1:]))
