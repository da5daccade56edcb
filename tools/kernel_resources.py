#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log) per kernel."""
import re
import subprocess
import sys

log = open(sys.argv[1]).read()
for b in log.split('Function Name: ')[1:]:
    name = b.split(' ')[0].split('[')[0].strip()

    def g(k):
        m = re.search(re.escape(k) + r': (\S+)', b)
        return m.group(1) if m else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r'hnsw_dev::', '', dn)
    dn = re.sub(r'\(.*', '', dn)[:64]
    print("%-64s VGPR=%4s AGPR=%3s SGPR=%4s scratch=%4s occ=%s" % (
        dn, g('VGPRs'), g('AGPRs'), g('SGPRs'), g('ScratchSize [bytes/lane]'), g('Occupancy [waves/SIMD]')))
