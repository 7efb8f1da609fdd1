"""Register / LDS / occupancy table of every kernel of libfokl_hip.so (development aid): parses the remarks of
`make -C fokl_gpy_amd/csrc asm` (-Rpass-analysis=kernel-resource-usage).  tools/kernel_resources.py > profiles/kernel_resources_rNN.txt"""
import os, re, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
out = subprocess.run(['make', '-C', os.path.join(ROOT, 'fokl_gpy_amd', 'csrc'), 'asm'], capture_output=True, text=True)
text = out.stdout + out.stderr
try:
    os.remove(os.path.join(ROOT, 'fokl_gpy_amd', 'csrc', 'fokl_hip.gfx950.s'))
except OSError:
    pass
rows, cur = [], None
for line in text.split('\n'):
    m = re.search(r'remark: .*?Name: (\S+)', line)
    if m:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r'\(.*', '', name).replace('void ', '').replace('fokl::', '')
        cur = dict(name=name)
        rows.append(cur)
        continue
    for key, pat in (('sgpr', r'TotalSGPRs: (\d+)'), ('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'),
                     ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                     ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'LDS(static)':>11s} {'waves/SIMD':>10s}")
for r in rows:
    print(f"{r['name'][:78]:78s} {r.get('vgpr', 0):5d} {r.get('agpr', 0):5d} {r.get('sgpr', 0):5d} {r.get('scratch', 0):7d} "
          f"{r.get('lds', 0):11d} {r.get('occ', 0):10d}")
