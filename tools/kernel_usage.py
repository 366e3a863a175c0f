"""Register / scratch usage of the kernels of one source file, from hipcc -Rpass-analysis=kernel-resource-usage output.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2> usage.txt; python tools/kernel_usage.py usage.txt [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].strip()
    try:
        name = subprocess.check_output(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name]).decode().strip()
    except Exception:
        pass
    if flt not in name:
        continue
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    print('%-120s VGPR %s AGPR %s spill %s scratch %s sgpr-spill %s occ %s' % (
        name[:120], g('VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g('SGPRs Spill'), g(r'Occupancy \[waves/SIMD\]')))
