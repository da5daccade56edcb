#!/usr/bin/env python3
"""Static check of the gfx950 data hazards that hardware does not interlock, on the DISASSEMBLED code objects.

Why: csrc/hnsw_hop_asm.hip.h and insert_island2 are inline assembly.  LLVM's hazard recognizer (GCNHazardRecognizer)
pads compiler-scheduled code with s_nop where gfx940-family hardware needs wait states, but it does not look inside an
`asm` string -- there the wait states are hand-counted.  This tool re-derives them from `llvm-objdump -d` of the gfx950
code object and fails on any pair that is too close.  It checks EVERY instruction of every kernel in the object (compiler
code passes by construction, which also calibrates the rules: a rule stricter than LLVM's would flag compiler code).

Rules (wait states = instructions issued in between, `s_nop N` counting N + 1; LLVM names in brackets):
  R1  v_dot* writes a VGPR -> a DIFFERENT VALU opcode reads it: 3; the same dot opcode reading it as src A / B: 3, as
      its accumulator (src C): 0                                        [DotWriteDifferentVALURead, DotWriteSameDotReadSrcAB]
  R2  v_dot* writes a VGPR -> a different VALU opcode overwrites it without reading it: 3  [DotWriteDifferentVALUWrite;
      calibrated on this toolchain: hipcc 7.2 pads  v_dot4 v4 .. ; v_cndmask_b32 v4  to exactly 3 states]
  R3  VALU writes an SGPR / VCC -> VALU reads that SGPR / VCC: 2                          [VALUWriteSGPRVALURead]
  R4  VALU writes an SGPR / VCC -> v_readlane / v_writelane uses it as lane select: 4     [RWLaneWaitStates]
  R5  VALU writes a VGPR -> a DPP instruction reads it as its DPP source or keeps it as
      the old value of its destination: 2                                                 [DppVgprWaitStates]
  R6  VALU writes EXEC (v_cmpx) -> DPP instruction: 5                                     [DppExecWaitStates]
  R7  VALU writes EXEC (v_cmpx) -> v_readlane / v_readfirstlane / v_writelane: 4          [VALUWriteEXECRWLane]
  R8  VALU writes a VGPR -> v_readlane / v_readfirstlane reads it: 1                      [VALUWriteVGPRReadlaneRead]
  R9  VALU writes an SGPR -> a vector-memory instruction reads that SGPR: 5               [VmemSgprWaitStates]
  R10 transcendental VALU writes a VGPR -> non-transcendental VALU reads it: 1            [TransDefWaitstates]
(SALU writes, including s_mov exec / s_mov m0, are interlocked for these consumers and only count as wait states.)

Hazards are followed across branches: every path that reaches the consumer within the rule's distance is examined.

    python tools/check_asm_hazards.py [object.o | code-object | disassembly.txt] ...   (default: the in-tree build's
                                       hnsw_search_variants_0_0_{1,2}.o and hnsw_order.hip.o)
    --mutate   self-test: delete each s_nop in turn and report how many deletions the checker catches
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")

DOT_SAME_AB, DOT_DIFF_READ, DOT_DIFF_WRITE = 3, 3, 3
VALU_SGPR_VALU, RWLANE_SELECT, DPP_VGPR, DPP_EXEC, EXEC_RWLANE, VGPR_READLANE, VMEM_SGPR, TRANS_USE = 2, 4, 2, 5, 4, 1, 5, 1

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
VMEM_PREFIX = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")


def disassemble(path):
    """-> text of `llvm-objdump -d` for a host object with an offload bundle, a bare code object, or a saved listing"""
    if path.endswith(".txt"):
        return open(path).read()
    with open(path, "rb") as f:
        head = f.read(20)
    tmp = tempfile.mkdtemp(prefix="hazards_")
    target = path
    if head[:4] == b"\x7fELF" and head[18:20] != b"\xe0\x00":      # not EM_AMDGPU (224): a host object, unbundle
        local = os.path.join(tmp, os.path.basename(path))
        with open(path, "rb") as src, open(local, "wb") as dst:
            dst.write(src.read())
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        if not cos:
            raise RuntimeError("no amdgcn bundle in " + path)
        target = cos[0]
    return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", target], check=True, capture_output=True, text=True).stdout


def split_operands(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


REG = re.compile(r"^(?:-|\||abs\(|neg\()?([vsa])(\d+)\|?\)?$")
REGRANGE = re.compile(r"^(?:-|\||abs\(|neg\()?([vsa])\[(\d+):(\d+)\]\|?\)?$")


def regs_of(tok):
    """register names an operand token stands for: {'v5'}, {'s4','s5'}, {'vcc'}, {'exec'}, {'m0'} ..."""
    tok = tok.strip()
    tok = re.sub(r"\s+(?:row_|quad_perm|wave_|bank_mask|bound_ctrl|dst_sel|dst_unused|src0_sel|src1_sel|clamp|op_sel|neg_|offset|sc0|sc1|nt|glc|slc|lds).*$", "", tok)
    m = REG.match(tok)
    if m:
        return {m.group(1) + m.group(2)}
    m = REGRANGE.match(tok)
    if m:
        return {m.group(1) + str(i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    base = tok.split()[0] if tok else ""
    if base in ("vcc", "vcc_lo", "vcc_hi"):
        return {"vcc"}
    if base in ("exec", "exec_lo", "exec_hi"):
        return {"exec"}
    if base == "m0":
        return {"m0"}
    return set()


class Ins:
    __slots__ = ("addr", "text", "mn", "ops", "ws", "valu", "dot", "dpp", "trans", "vmem", "defs", "uses", "dpp_src", "lane_sel",
                 "dot_c", "rlane_src", "target", "ends", "func", "dead")

    def __init__(self, addr, text, func):
        self.addr, self.text, self.func = addr, text, func
        parts = text.split(None, 1)
        self.mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.ops = split_operands(rest)
        mn = self.mn
        self.dead = False
        self.ws = 1
        if mn == "s_nop":
            self.ws = int(self.ops[0], 0) + 1
        self.valu = mn.startswith("v_") and not mn.startswith("v_nop")
        self.dot = mn.startswith("v_dot")
        self.dpp = "_dpp" in mn or bool(re.search(r"\b(row_|quad_perm|wave_sh|wave_ro|row_bcast|row_mirror|row_half_mirror)", rest))
        self.trans = mn.startswith(TRANS)
        self.vmem = mn.startswith(VMEM_PREFIX)
        self.defs, self.uses = set(), set()
        self.dpp_src, self.lane_sel, self.dot_c, self.rlane_src = set(), set(), set(), set()
        self.target, self.ends = None, mn in ("s_endpgm", "s_branch", "s_setpc_b64", "s_swappc_b64", "s_trap")
        ops = [regs_of(o) for o in self.ops]
        if self.valu:
            if mn.startswith("v_cmpx_"):
                self.defs |= {"exec"}
                start = 0
                if ops and (ops[0] & {"vcc"} or any(r.startswith("s") for r in ops[0])) and not mn.endswith("_e32"):
                    self.defs |= ops[0]
                    start = 1
                for o in ops[start:]:
                    self.uses |= o
            elif mn.startswith("v_cmp_"):
                if mn.endswith("_e32"):
                    self.defs |= {"vcc"}
                    for o in ops:
                        self.uses |= o
                    # the disassembler prints the implicit vcc destination first
                    if ops and ops[0] == {"vcc"}:
                        self.uses -= {"vcc"} if not any(o == {"vcc"} for o in ops[1:]) else set()
                else:
                    self.defs |= ops[0] if ops else set()
                    for o in ops[1:]:
                        self.uses |= o
            elif mn.startswith(("v_readlane_b32", "v_readfirstlane_b32")):
                self.defs |= ops[0]
                self.uses |= ops[1]
                self.rlane_src |= ops[1]
                if len(ops) > 2:
                    self.uses |= ops[2]
                    self.lane_sel |= {r for r in ops[2] if r.startswith("s") or r == "vcc"}
            elif mn.startswith("v_writelane_b32"):
                self.defs |= ops[0]
                self.uses |= ops[1]
                if len(ops) > 2:
                    self.uses |= ops[2]
                    self.lane_sel |= {r for r in ops[2] if r.startswith("s") or r == "vcc"}
            else:
                if ops:
                    self.defs |= ops[0]
                rest_ops = ops[1:]
                if ("_co_" in mn or mn.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_div_scale"))) and rest_ops:
                    self.defs |= rest_ops[0]                       # carry-out / scalar destination
                    rest_ops = rest_ops[1:]
                for o in rest_ops:
                    self.uses |= o
                if self.dpp and rest_ops:
                    # the DPP source, and the destination too: lanes the DPP control switches off keep the OLD value, which
                    # the instruction therefore reads (LLVM: the tied `old` operand is one of the uses it checks)
                    self.dpp_src |= {r for r in rest_ops[0] if r.startswith("v")} | {r for r in ops[0] if r.startswith("v")}
                if self.dot and len(rest_ops) >= 3:
                    self.dot_c |= {r for r in rest_ops[2] if r.startswith("v")}
                if mn.startswith(("v_cndmask_b32_e32", "v_cndmask_b32_dpp", "v_cndmask_b32_sdwa", "v_addc_", "v_subb_", "v_subbrev_", "v_div_fmas")) and \
                        not any(o == {"vcc"} or any(r.startswith("s") for r in o) for o in rest_ops[2:]):
                    self.uses |= {"vcc"}                           # implicit carry / select input
        elif self.vmem or mn.startswith("ds_"):
            for o in ops:
                self.uses |= o                                     # (destinations of loads are not VALU writes: no rule looks at them)
        # SALU / SMEM: neither producer nor consumer of any rule here
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", text)
            m0 = re.search(r"<([^>+]+)>", text)
            if m:
                self.target = ("rel", int(m.group(1), 16))
            elif m0:
                self.target = ("rel", 0)


def parse(text):
    """-> {function name: [Ins]} with branch targets resolved to indices"""
    funcs, cur, name, start = {}, None, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            name, start = m.group(2), int(m.group(1), 16)
            cur = funcs.setdefault(name, [])
            continue
        m = re.match(r"^\t(.+?)\s*// ([0-9A-F]+): ", line)
        if m and cur is not None:
            ins = Ins(int(m.group(2), 16), m.group(1).strip(), name)
            if ins.target:
                ins.target = start + ins.target[1]
            cur.append(ins)
    for body in funcs.values():
        at = {i.addr: k for k, i in enumerate(body)}
        for i in body:
            if i.target is not None:
                i.target = at.get(i.target)
    return funcs


def check(body, report_limit=50):
    """-> list of violation strings for one function"""
    preds = [[] for _ in body]
    for k, i in enumerate(body):
        if k + 1 < len(body) and not i.ends:
            preds[k + 1].append(k)
        if i.target is not None:
            preds[i.target].append(k)
    out = []

    def producers(k, need, match):
        """walk back from instruction k over every path; yield producers closer than `need` wait states"""
        stack, seen = [(p, 0) for p in preds[k]], set()
        while stack:
            j, c = stack.pop()
            if c >= need or (j, c) in seen:
                continue
            seen.add((j, c))
            pj = body[j]
            if not pj.dead and match(pj):
                yield pj, c
                continue                       # the nearest producer on this path decides
            c2 = c + (0 if pj.dead else pj.ws)
            for p in preds[j]:
                stack.append((p, c2))

    def flag(rule, need, cons, prod, have):
        out.append("%s: needs %d wait states, has %d\n      producer %06X  %s\n      consumer %06X  %s" %
                   (rule, need, have, prod.addr, prod.text, cons.addr, cons.text))

    for k, c in enumerate(body):
        if c.dead or len(out) >= report_limit:
            continue
        if c.valu:
            vuse = {r for r in c.uses if r.startswith("v")}
            suse = {r for r in c.uses if r.startswith("s") or r == "vcc"}
            vdef = {r for r in c.defs if r.startswith("v")}
            if vuse:
                for p, have in producers(k, DOT_DIFF_READ, lambda p: p.dot and p.defs & vuse):
                    hit = p.defs & vuse
                    if p.mn == c.mn and hit <= c.dot_c and not (hit & ((c.uses - c.dot_c) | set())):
                        continue                                   # the same dot opcode accumulating into it
                    flag("R1 dot result read by another VALU" if p.mn != c.mn else "R1 dot result read as src A/B by the same dot", DOT_DIFF_READ, c, p, have)
                if not c.trans:
                    for p, have in producers(k, TRANS_USE, lambda p: p.trans and p.defs & vuse):
                        flag("R10 transcendental result used", TRANS_USE, c, p, have)
            if vdef:
                # (an instruction that also READS the register is covered by R1: hipcc itself pads such pairs to 3, e.g.
                # v_dot4 v27 ... v_lshlrev_b32 v27, 1, v27; the 4th state is for an overwrite that does not wait on the value)
                pure = vdef - vuse
                if pure:
                    for p, have in producers(k, DOT_DIFF_WRITE, lambda p: p.dot and p.mn != c.mn and p.defs & pure):
                        flag("R2 dot result overwritten by another VALU", DOT_DIFF_WRITE, c, p, have)
            if suse:
                for p, have in producers(k, VALU_SGPR_VALU, lambda p: p.valu and p.defs & suse):
                    flag("R3 VALU-written SGPR/VCC read by VALU", VALU_SGPR_VALU, c, p, have)
            if c.lane_sel:
                for p, have in producers(k, RWLANE_SELECT, lambda p: p.valu and p.defs & c.lane_sel):
                    flag("R4 VALU-written lane select", RWLANE_SELECT, c, p, have)
            if c.dpp:
                if c.dpp_src:
                    for p, have in producers(k, DPP_VGPR, lambda p: p.valu and p.defs & c.dpp_src):
                        flag("R5 DPP source written by VALU", DPP_VGPR, c, p, have)
                for p, have in producers(k, DPP_EXEC, lambda p: p.valu and "exec" in p.defs):
                    flag("R6 VALU write of EXEC before DPP", DPP_EXEC, c, p, have)
            if c.mn.startswith(("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32")):
                for p, have in producers(k, EXEC_RWLANE, lambda p: p.valu and "exec" in p.defs):
                    flag("R7 VALU write of EXEC before a lane access", EXEC_RWLANE, c, p, have)
            if c.rlane_src:
                for p, have in producers(k, VGPR_READLANE, lambda p: p.valu and p.defs & c.rlane_src):
                    flag("R8 v_readlane of a VGPR just written", VGPR_READLANE, c, p, have)
        elif c.vmem:
            suse = {r for r in c.uses if r.startswith("s") or r == "vcc"}
            if suse:
                for p, have in producers(k, VMEM_SGPR, lambda p: p.valu and p.defs & suse):
                    flag("R9 VALU-written SGPR read by vector memory", VMEM_SGPR, c, p, have)
    return out


def default_objects():
    b = os.path.join(ROOT, "ocaml-hnsw_amd", "build")
    # the translation units with hand-scheduled code: per accept rule (0 Ohnsw, 1 functor) byte rows L2 / inner product (0_s_2, 1_s_2), float32 rows L2
    # full / ragged / split (0_s_1, 0_s_0, 0_s_3) and inner product (1_s_1, 1_s_0, 1_s_3); the descent pre-pass
    names = ["hnsw_search_variants_%d_%d_%d.o" % (m, sem, r) for sem in (0, 1) for (m, r) in ((0, 2), (1, 2), (0, 1), (0, 0), (0, 3), (1, 1), (1, 0), (1, 3))]
    return [os.path.join(b, f) for f in names + ["hnsw_order.hip.o"]]


def main(argv):
    mutate = "--mutate" in argv
    paths = [a for a in argv if not a.startswith("--")] or default_objects()
    bad = 0
    for path in paths:
        funcs = parse(disassemble(path))
        n_ins = sum(len(b) for b in funcs.values())
        n_nop = sum(1 for b in funcs.values() for i in b if i.mn == "s_nop")
        v = []
        for name, body in funcs.items():
            for msg in check(body):
                v.append("%s\n    %s" % (name, msg))
        print("%s: %d kernels, %d instructions, %d s_nop, %d hazard violations" % (os.path.basename(path), len(funcs), n_ins, n_nop, len(v)))
        for msg in v[:40]:
            print("  " + msg)
        bad += len(v)
        if mutate:
            caught = total = 0
            for name, body in funcs.items():
                for i in body:
                    if i.mn != "s_nop":
                        continue
                    total += 1
                    i.dead = True
                    if check(body, report_limit=1):
                        caught += 1
                    i.dead = False
            print("  mutation self-test: %d of %d single s_nop deletions are caught (the others pad for something no rule here covers, "
                  "or are the compiler's alignment / scheduling nops)" % (caught, total))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
