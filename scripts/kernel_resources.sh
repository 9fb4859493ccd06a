#!/bin/bash
# Register / scratch / LDS use of every kernel of one instantiation file (compile-only, no GPU needed):
#   scripts/kernel_resources.sh inst_unicycle_f64.hip
cd "$(dirname "$0")/../altro-cpp_amd/csrc" || exit 1
FLAGS=$(grep '^CXXFLAGS' Makefile | sed 's/^CXXFLAGS := //; s/\$(ARCH)/gfx950/')
/opt/rocm/bin/hipcc $FLAGS $EXTRA --cuda-device-only -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  python3 -c "
import sys, re, subprocess
name = None; rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'altro_hip::', '', name); rows[name] = {}
    for key in ('VGPRs', 'AGPRs', 'SGPRs', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]', 'LDS Size \[bytes/block\]'):
        m = re.search(r'remark: .*?\s' + key + r': (\d+)', line)
        if m and name: rows[name][key.split(' ')[0]] = m.group(1)
for n, r in rows.items():
    print(f\"{n[:100]:100s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('SGPRs','?'):>4s} scratch {r.get('ScratchSize','?'):>5s} occ {r.get('Occupancy','?'):>2s} lds {r.get('LDS','?'):>6s}\")
"
