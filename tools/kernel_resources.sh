#!/bin/bash
# VGPRs / scratch / occupancy of the kernels of one source file (hipcc -Rpass-analysis=kernel-resource-usage, device only):
#   bash tools/kernel_resources.sh gemm2.hip [name-filter]
F=${1:-gemm2.hip}; PAT=${2:-.}
cd /tmp && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffast-math -fno-finite-math-only -Wno-unused-value \
  --offload-device-only -Rpass-analysis=kernel-resource-usage -c /root/repo/kalle_audio_amd/csrc/$F -o /tmp/_res.o 2> /tmp/_res.txt
python3 - "$PAT" <<'PY'
import re, sys, subprocess
pat = sys.argv[1]
txt = open('/tmp/_res.txt').read()
for blk in txt.split('Function Name: ')[1:]:
    name = blk.split()[0]
    try:
        name = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt', name], capture_output=True, text=True).stdout.strip()
    except OSError:
        pass
    if not re.search(pat, name):
        continue
    g = lambda k: (re.search(k + r': (\d+)', blk) or [None, '?'])[1]
    scr, occ = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
    print(f"{name[:110]:110s} VGPR {g('VGPRs'):>3} AGPR {g('AGPRs'):>3} scratch {scr:>4} occ {occ} spillV {g('VGPRs Spill')}")
PY
