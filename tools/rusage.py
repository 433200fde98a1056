"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output read from stdin."""
import re, subprocess, sys
rows = []
cur = None
for line in sys.stdin:
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}
        rows.append(cur)
    for k, pat in (('v', 'VGPRs'), ('a', 'AGPRs'), ('scr', r'ScratchSize \[bytes/lane\]'), ('occ', r'Occupancy \[waves/SIMD\]'), ('lds', r'LDS Size \[bytes/block\]')):
        m = re.search(r'remark:\s+' + pat + r': (\d+)', line)
        if m and cur is not None:
            cur[k] = int(m.group(1))
    if 'error' in line:
        print(line.rstrip())
flt = sys.argv[1] if len(sys.argv) > 1 else ''
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    if re.search(flt, n):
        print(n[:84].ljust(86), 'v', r.get('v'), 'a', r.get('a'), 'scr', r.get('scr'), 'occ', r.get('occ'))
