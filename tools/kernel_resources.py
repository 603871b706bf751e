#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of a csrc/*.hip file as the compiler reports it (-Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py mmgen_kernels.hip [-DNAME=VALUE ...]"""
import re, subprocess, sys, os
src = sys.argv[1]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'mega-minecraft_amd', 'csrc')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-fast-math', '-fno-slp-vectorize',
       '--offload-device-only', '-c', os.path.join(csrc, src), '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.split('\n'):
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        cur = {'name': re.sub(r'\(.*', '', d)}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'sSpill':>7s} {'vSpill':>7s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:44]:44s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>8s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('SGPRs Spill','?'):>7s} {r.get('VGPRs Spill','?'):>7s} {r.get('LDS Size [bytes/block]','?'):>7s}")
