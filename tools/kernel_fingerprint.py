#!/usr/bin/env python3
"""Fingerprint of every gfx950 kernel of the in-tree build: one line per kernel symbol with the number of instructions and a
hash of its instruction stream (mnemonics and operands as llvm-objdump prints them; addresses and encodings dropped, branch
targets reduced to their distance).  Two builds whose lists are equal run the same code -- what a refactoring of the
hand-scheduled loops' instantiation (hnsw_hop_loop.inc, tools/gen_hop_slots.py) is checked with.

    python tools/kernel_fingerprint.py [objects ...] > before.txt      (default: every ocaml-hnsw_amd/build/*.o)
    python tools/kernel_fingerprint.py --diff before.txt after.txt     kernels added / removed / changed
"""
import glob
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from check_asm_hazards import disassemble, LLVM  # noqa: E402

SYM = re.compile(r"^[0-9a-f]+ <(.+)>:$")
INS = re.compile(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):")


def demangle(names):
    tool = os.path.join(LLVM, "llvm-cxxfilt")
    out = subprocess.run([tool if os.path.exists(tool) else "c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def fingerprints(path):
    try:
        txt = disassemble(path)
    except (RuntimeError, subprocess.CalledProcessError):
        return []                     # a host-only object: no device code in it
    kernels, cur, body = [], None, []
    for line in txt.splitlines():
        m = SYM.match(line)
        if m:
            if cur is not None:
                kernels.append((cur, body))
            cur, body = m.group(1), []
            continue
        m = INS.match(line)
        if m and cur is not None:
            body.append(m.group(1))          # (a branch prints its relative offset: nothing position dependent is left)
    if cur is not None:
        kernels.append((cur, body))
    names = demangle([k for k, _ in kernels])
    out = []
    for (_, body), name in zip(kernels, names):
        if not body:
            continue
        h = hashlib.sha256("\n".join(body).encode()).hexdigest()[:16]
        out.append("%s\t%d\t%s" % (name.split("(")[0].replace("void ", "").replace(" ", ""), len(body), h))
    return out


def main(argv):
    if argv and argv[0] == "--diff":
        def load(p):
            d = {}
            for ln in open(p):
                f = ln.rstrip("\n").split("\t")
                if len(f) == 4:
                    d[(f[0], f[1])] = (f[2], f[3])
            return d
        a, b = load(argv[1]), load(argv[2])
        gone, new = sorted(set(a) - set(b)), sorted(set(b) - set(a))
        changed = sorted(k for k in set(a) & set(b) if a[k] != b[k])
        for k in gone:
            print("removed ", k[0], k[1])
        for k in new:
            print("added   ", k[0], k[1], b[k][0])
        for k in changed:
            print("changed ", k[0], k[1], "%s -> %s instructions" % (a[k][0], b[k][0]))
        print("%d kernels before, %d after: %d removed, %d added, %d changed, %d identical" %
              (len(a), len(b), len(gone), len(new), len(changed), len(set(a) & set(b)) - len(changed)))
        return 1 if (gone or changed) else 0
    objs = argv or sorted(glob.glob(os.path.join(ROOT, "ocaml-hnsw_amd", "build", "*.o")))
    for o in objs:
        for ln in fingerprints(o):
            print("%s\t%s" % (os.path.basename(o), ln))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
