"""Static hazard check of the hand-scheduled gfx950 code (tools/check_asm_hazards.py).

csrc/hnsw_hop_asm.hip.h + hnsw_hop_loop.inc (the layer-0 loops over byte rows and over float32 rows, the descent of the
headline shape) and insert_island2 are inline assembly:
nothing inserts the wait states gfx950 needs between, say, a v_dot4 and a different vector instruction that reads its
result.  The checker re-derives them from the disassembly of the built code objects.  Here: its rules on hand-made
listings, a clean pass over the objects of the in-tree build (hipcc cross-compiles without a GPU), and the mutation test --
every single s_nop of the hand-scheduled kernels deleted in turn from a copy of the listing: the checker must notice."""
import importlib.util
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    if "check_asm_hazards" in sys.modules:                   # (one module object per process: its classes travel between the worker
        return sys.modules["check_asm_hazards"]              # processes of the `objects` fixture and the test process by name)
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["check_asm_hazards"] = m
    spec.loader.exec_module(m)
    return m


def _listing(lines):
    """lines of assembly -> text in llvm-objdump's format (4-byte addresses are enough for the parser)"""
    out = ["0000000000001000 <k>:"]
    for i, ln in enumerate(lines):
        out.append("\t%-60s// %012X: 00000000" % (ln, 0x1000 + 4 * i))
    return "\n".join(out) + "\n"


def _violations(hz, lines):
    funcs = hz.parse(_listing(lines))
    return hz.check(funcs["k"])


def test_rules_on_hand_made_listings():
    hz = _tool()
    dot = "v_dot4_u32_u8 v1, v2, v3, 0"
    # R1: a different VALU reads a dot result: 3 wait states
    assert _violations(hz, [dot, "s_nop 1", "v_add_u32_e32 v4, v1, v5", "s_endpgm"])
    assert not _violations(hz, [dot, "s_nop 2", "v_add_u32_e32 v4, v1, v5", "s_endpgm"])
    assert not _violations(hz, [dot, "s_mov_b32 s0, 0", "s_mov_b32 s1, 0", "v_mov_b32_e32 v9, 0", "v_add_u32_e32 v4, v1, v5", "s_endpgm"])
    # ... the same dot opcode taking it as its accumulator needs none, as a factor it needs 3
    assert not _violations(hz, [dot, "v_dot4_u32_u8 v1, v6, v7, v1", "s_endpgm"])
    assert _violations(hz, [dot, "v_dot4_u32_u8 v8, v1, v7, 0", "s_endpgm"])
    # R2: overwritten by another VALU
    assert _violations(hz, [dot, "s_nop 1", "v_mov_b32_e32 v1, 0", "s_endpgm"])
    # R3 / R4: VALU-written SGPR or VCC read by VALU (2), used as lane select (4)
    assert _violations(hz, ["v_cmp_lt_u32_e32 vcc, v1, v2", "s_nop 0", "v_cndmask_b32_e32 v3, v4, v5, vcc", "s_endpgm"])
    assert not _violations(hz, ["v_cmp_lt_u32_e32 vcc, v1, v2", "s_nop 1", "v_cndmask_b32_e32 v3, v4, v5, vcc", "s_endpgm"])
    assert _violations(hz, ["v_readlane_b32 s4, v1, 3", "s_nop 2", "v_writelane_b32 v2, s5, s4", "s_endpgm"])
    assert not _violations(hz, ["v_readlane_b32 s4, v1, 3", "s_nop 3", "v_writelane_b32 v2, s5, s4", "s_endpgm"])
    assert not _violations(hz, ["s_mov_b32 m0, s4", "v_writelane_b32 v2, s5, m0", "s_endpgm"])          # SALU write of m0: interlocked
    # R5: DPP source written by VALU: 2
    assert _violations(hz, ["v_add_u32_e32 v1, v2, v3", "s_nop 0", "v_mov_b32_dpp v4, v1 wave_shr:1 row_mask:0xf bank_mask:0xf", "s_endpgm"])
    assert not _violations(hz, ["v_add_u32_e32 v1, v2, v3", "s_nop 1", "v_mov_b32_dpp v4, v1 wave_shr:1 row_mask:0xf bank_mask:0xf", "s_endpgm"])
    # R8: v_readlane of a VGPR just written: 1
    assert _violations(hz, ["v_add_u32_e32 v1, v2, v3", "v_readlane_b32 s0, v1, 5", "s_endpgm"])
    # R9: VALU-written SGPR as the scalar base of a vector-memory instruction: 5
    assert _violations(hz, ["v_readlane_b32 s4, v1, 3", "v_readlane_b32 s5, v1, 4", "s_nop 2", "global_load_dword v2, v3, s[4:5]", "s_endpgm"])
    assert not _violations(hz, ["v_readlane_b32 s4, v1, 3", "v_readlane_b32 s5, v1, 4", "s_nop 4", "global_load_dword v2, v3, s[4:5]", "s_endpgm"])
    # across a branch: the hazard is followed along the taken edge and along the fall-through
    loop = [dot,                                              # 0x1000
            "s_cbranch_scc1 3 <k+0x14>",                      # 0x1004 -> 0x1014
            "s_nop 3",                                        # 0x1008
            "s_branch 1 <k+0x14>",                            # 0x100c
            "s_nop 0",                                        # 0x1010 (never executed)
            "v_add_u32_e32 v4, v1, v5",                       # 0x1014: reached from 0x1004 with one state (the branch)
            "s_endpgm"]
    assert _violations(hz, loop)
    loop[1] = "s_nop 0"
    assert not _violations(hz, loop)


def _wanted():
    """(object, mangled-name fragment) of the kernels that contain hand-scheduled code: the layer-0 loops (W in 1 / 2 / 3 / 4 / 6 / 8 key
    registers) over byte rows and over float32 rows (L2 and inner product; full, ragged and split rows), with the tag cache and
    with the bitmap blocks, and the descent kernel with the hand-scheduled descent"""
    want = [("hnsw_search_variants_0_0_2.o", "hnsw_search_kernelILi2ELi4ELi%dELi0ELi0ELi2ELi0EE" % s) for s in (1, 2, 3, 4, 6)]     # (3, 6: the generated insertions)
    for obj, metric, rows in (("hnsw_search_variants_0_0_1.o", 0, 1), ("hnsw_search_variants_0_0_0.o", 0, 0),
                              ("hnsw_search_variants_1_0_1.o", 1, 1), ("hnsw_search_variants_1_0_0.o", 1, 0)):
        want += [(obj, "hnsw_search_kernelILi2ELi4ELi%dELi%dELi0ELi%dELi0EE" % (s, metric, rows)) for s in (1, 2, 4)]
    want += [("hnsw_search_variants_0_1_2.o", "hnsw_search_kernelILi2ELi4ELi%dELi0ELi1ELi2ELi0EE" % s) for s in (1, 2, 4)]     # the functor rule's instantiations
    want += [("hnsw_search_variants_1_1_0.o", "hnsw_search_kernelILi2ELi4ELi%dELi1ELi1ELi0ELi0EE" % s) for s in (2,)]
    want += [("hnsw_search_variants_1_0_2.o", "hnsw_search_kernelILi2ELi4ELi%dELi1ELi0ELi2ELi0EE" % s) for s in (2, 4)]              # byte rows, inner product
    want += [("hnsw_search_variants_1_0_3.o", "hnsw_search_kernelILi2ELi4ELi%dELi1ELi0ELi3ELi0EE" % s) for s in (4,)]               # split rows (C3's kernel)
    want += [("hnsw_search_variants_0_0_0.o", "hnsw_search_kernelILi2ELi4ELi8ELi0ELi0ELi0ELi0EE")]                                    # eight slots (C5's kernel)
    want += [("hnsw_search_variants_0_0_0.o", "hnsw_search_kernelILi2ELi4ELi%dELi0ELi0ELi0ELi1EE" % s) for s in (3, 6)]                  # three / six slots with the bitmap blocks
    # Visited as bitmap blocks (BLK = 1): the block filter inside the eight-slot ragged-row loop (C5's kernel when its data is
    # clustered), inside the four-slot split-row inner-product loop (C3's) and inside the four-slot byte-row loop
    want += [("hnsw_search_variants_0_0_0.o", "hnsw_search_kernelILi2ELi4ELi8ELi0ELi0ELi0ELi1EE"),
             ("hnsw_search_variants_1_0_3.o", "hnsw_search_kernelILi2ELi4ELi4ELi1ELi0ELi3ELi1EE"),
             ("hnsw_search_variants_0_0_2.o", "hnsw_search_kernelILi2ELi4ELi4ELi0ELi0ELi2ELi1EE")]
    want += [("hnsw_search_variants_0_0_1.o", "hnsw_search_kernelILi4ELi2ELi4ELi0ELi0ELi1ELi1EE")]      # ... and in a loop over rows of 129..256 dimensions
    want += [("hnsw_order.hip.o", "hnsw_descent_kernelILi2ELi8ELi0ELi2E")]
    return want


def _scan_object(path):
    """(worker process) one code object: every kernel checked, the hand-scheduled ones handed back parsed"""
    hz = _tool()
    funcs = hz.parse(hz.disassemble(path))
    name = os.path.basename(path)
    bad = [(k, str(v)) for k, body in funcs.items() for v in hz.check(body)]
    frags = [f for o, f in _wanted() if o == name]
    keep = {k: b for k, b in funcs.items() if any(f in k for f in frags)}
    return name, len(funcs), bad, keep


@pytest.fixture(scope="module")
def objects():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") and not shutil.which("llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    hz = _tool()
    paths = hz.default_objects()
    if not all(os.path.exists(p) for p in paths):            # a tree that was never built here: build it (hipcc cross-compiles)
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g._load_build_module().build(force=True)
    # seventeen objects of some 200 000 instructions each: disassembled, parsed and checked in worker processes; only the
    # violations and the hand-scheduled kernels (the mutation test's material) come back
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
        scanned = list(ex.map(_scan_object, paths))
    return hz, {name: (nfuncs, bad, keep) for name, nfuncs, bad, keep in scanned}


def test_built_code_objects_have_no_unpadded_hazard(objects):
    hz, objs = objects
    for name, (nfuncs, bad, _) in objs.items():
        assert nfuncs >= 1, name
        assert not bad, "%s: %d hazards, first: %s\n%s" % (name, len(bad), bad[0][0], bad[0][1])


def _asm_kernels(objs):
    out = []
    for obj, frag in _wanted():
        hits = [(k, b) for k, b in objs[obj][2].items() if frag in k]
        assert len(hits) == 1, (obj, frag, [k for k, _ in hits])
        out.append((obj, hits[0][0], hits[0][1]))
    return out


def _owned_nops(body):
    """the s_nop instructions that pad a hazard (the checker's business): not the runs that align a loop head
    (HNSW_ASM_ALIGN_K: three or more in a row), not the single ones the compiler puts between two memory instructions of
    its own code (they end a "soft clause" whose later load overwrites an earlier one's address registers: an XNACK replay
    rule, not a wait-state rule)"""
    mem = ("global_", "buffer_", "flat_", "scratch_", "s_load", "s_buffer_load")
    live = [i for i in body if not getattr(i, "dead", False)]
    out = []
    for n, i in enumerate(live):
        if i.mn != "s_nop":
            continue
        a = n
        while a > 0 and live[a - 1].mn == "s_nop":
            a -= 1
        b = n
        while b + 1 < len(live) and live[b + 1].mn == "s_nop":
            b += 1
        if b - a + 1 >= 3:
            continue
        prev, nxt = (live[a - 1].mn if a > 0 else ""), (live[b + 1].mn if b + 1 < len(live) else "")
        if prev.startswith(mem) and nxt.startswith(mem):
            continue
        out.append(i)
    return out


def test_labels_are_unique_within_every_block(tmp_path):
    """The blocks use numeric local labels and assume each is defined once per asm statement ("66f" must mean THE 66): the
    preprocessed text of one translation unit built with HNSW_HOP_ALL_INSTANCES holds every instantiation of the generated
    table (a unit normally holds its own (metric, rule, row format) only); a label defined twice in one statement -- an
    eight-slot pop label reused by a later addition, say -- fails here, not on the GPU."""
    import re
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not available")
    out = str(tmp_path / "pp.ii")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-DHNSW_V_METRIC=0", "-DHNSW_V_SEMF=0",
                    "-DHNSW_V_FULL=0", "-DHNSW_HOP_ALL_INSTANCES", "--cuda-device-only", "-E", "-P", os.path.join(ROOT, "ocaml-hnsw_amd", "csrc", "hnsw_search_variants.hip"),
                    "-o", out], check=True, capture_output=True)
    text = open(out).read()
    blocks = []
    lit = re.compile(r'\s*"((?:[^"\\]|\\.)*)"')
    for m in re.finditer(r"asm volatile\(", text):
        j, parts = m.end(), []
        while True:
            mm = lit.match(text, j)                    # (no slicing: the preprocessed unit is tens of megabytes)
            if not mm:
                break
            parts.append(bytes(mm.group(1), "utf-8").decode("unicode_escape"))
            j = mm.end()
        body = "".join(parts)
        if re.search(r"^\s*\d+:", body, flags=re.M):
            blocks.append(body)
    spec = importlib.util.spec_from_file_location("gen_hop_slots", os.path.join(ROOT, "tools", "gen_hop_slots.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    n_inst = len(gen.instances())
    assert n_inst == 304                             # 4 row families x rows x 2 metrics x 2 rules x 6 slot counts (+ 4 with bitmap blocks)
    assert len(blocks) >= n_inst + 2, len(blocks)    # ... the descent, the island
    for body in blocks:
        labels = re.findall(r"^\s*(\d+):", body, flags=re.M)
        dup = sorted({x for x in labels if labels.count(x) > 1})
        assert not dup, "labels defined twice in one block: %s" % dup


def test_generated_insertion_is_current_and_equals_the_hand_written_one():
    """csrc/hnsw_hop_slots.inc (what depends on the slot count: W in 3, 4, 6, 8 registers) and csrc/hnsw_hop_instances.inc (the
    table of instantiations) are what tools/gen_hop_slots.py writes, and the generator, asked for FOUR slots, reproduces the
    hand-written HNSW_INSERT_LOOP4 / HNSW_INSERT_RARE4 of hnsw_hop_asm.hip.h instruction for instruction"""
    import re
    spec = importlib.util.spec_from_file_location("gen_hop_slots", os.path.join(ROOT, "tools", "gen_hop_slots.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    assert gen.main(["--check"]) == 0

    def body(text, name):
        i = text.index("#define %s " % name)
        out = []
        for ln in text[i:].split("\n"):
            out.append(ln)
            if not ln.rstrip().endswith("\\"):
                break
        b = "\n".join(out)
        b = re.sub(r"/\*.*?\*/", "", b[b.index(name) + len(name):], flags=re.S)
        return b.replace("\\\n", "\n")

    def instructions(b):
        text = ""
        for t in re.findall(r'"(?:[^"\\]|\\.)*"|HNSW_[A-Z0-9_]+(?:\([^)]*\))?', b):
            if t.startswith('"'):
                text += bytes(t[1:-1], "utf-8").decode("unicode_escape")
            elif t == "HNSW_DPP_ALL":
                text += " row_mask:0xf bank_mask:0xf"
            elif t == "HNSW_SEM_REJECT":
                text += "18f"
            else:
                text += "\n" + re.sub(r"\s+", " ", t) + "\n"
        return [x.strip() for x in text.split("\n") if x.strip()]

    hand = open(os.path.join(ROOT, "ocaml-hnsw_amd", "csrc", "hnsw_hop_asm.hip.h")).read()
    four = gen.text(4)
    for name in ("HNSW_INSERT_LOOP4", "HNSW_INSERT_RARE4"):
        a, b = instructions(body(hand, name)), instructions(body(four, name))
        assert len(a) > 20 and a == b, name


def test_a_deleted_wait_state_is_noticed(objects):
    """the deliberately broken copy: each hazard-padding s_nop of the hand-scheduled kernels deleted in turn from the parsed listing"""
    hz, objs = objects
    for obj, name, body in _asm_kernels(objs):
        nops = _owned_nops(body)
        caught = 0
        for i in nops:
            i.dead = True
            if hz.check(body, report_limit=1):
                caught += 1
            i.dead = False
        print("%s %s: %d instructions, %d of %d single s_nop deletions caught" % (obj, name[:60], len(body), caught, len(nops)))
        assert nops and caught >= 0.9 * len(nops), (name, caught, len(nops))
        assert not hz.check(body)                             # and the untouched listing is clean again


def test_hand_scheduled_blocks_hand_m0_back():
    """the blocks use m0 as the lane select of v_writelane.  m0 is a reserved register: naming it as a clobber
    draws `inline asm clobber list contains reserved registers` (128 warnings in round 4) and promises nothing.  Every block now
    saves m0 on entry and restores it at its single exit, and no clobber list names it -- checked on the source text (the build
    prints no warning: ocaml-hnsw_amd/build.py)."""
    import re
    csrc = os.path.join(ROOT, "ocaml-hnsw_amd", "csrc")
    inc = open(os.path.join(csrc, "hnsw_hop_loop.inc")).read()
    blocks = inc.split("asm volatile(")[1:]
    assert len(blocks) == 3                                   # W in 2 / three and more / 1 registers
    for b in blocks:
        body = b.split("    w.wmax = wmax;")[0]
        assert body.count('"s_mov_b32 %[sm0], m0\\n\\t"') == 1 and body.count('s_mov_b32 m0, %[sm0]') == 1
        first = "v_alignbit_b32" if "v_alignbit_b32" in body else "HNSW_NS(ALIGN_IN)"      # (the generic body: the slot set's macro)
        assert body.index("s_mov_b32 %[sm0], m0") < body.index(first) < body.rindex("s_mov_b32 m0, %[sm0]")
        assert '[sm0] "=&s"(sm0)' in body
    for f in os.listdir(csrc):
        txt = open(os.path.join(csrc, f)).read()
        for clob in re.findall(r'\n\s*: ((?:"[a-z0-9]+",? ?)+)(?:HNSW_[A-Z0-9_]+ ?)*\);', txt):
            assert '"m0"' not in clob, (f, clob)
