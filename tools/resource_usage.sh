#!/bin/bash
# usage: tools/resource_usage.sh [extra hipcc flags]  -- registers / spills / LDS / occupancy of every
# env_kernel instantiation as the compiler reports them (-Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 \
  -Rpass-analysis=kernel-resource-usage "$@" -o /tmp/rvo3d_ru.so \
  3drvo-marl-collisionavoidance_amd/csrc/rvo3d_capi.hip 2>&1 | python3 -c "
import re, sys
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m: cur = m.group(1); rows[cur] = {}
    for key in ('VGPRs:', 'AGPRs', 'SGPRs:', 'ScratchSize', 'Occupancy', 'SGPRs Spill', 'VGPRs Spill', 'LDS Size'):
        m = re.search(key + r'[^:]*:\s*(\d+)', line) if not key.endswith(':') else re.search(key + r'\s*(\d+)', line)
        if m and cur: rows[cur][key.strip(':')] = int(m.group(1))
import subprocess
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()
    if 'env_kernel' in name or 'kernel' in name:
        print(f'{name[:70]:70s}', ' '.join(f'{a}={b}' for a, b in v.items()))
"
