#!/bin/bash
# development aid: sensitivity of the memory-bound level-0 kernels to the segment length near the default (the tile stride of the
# packed arrays moves with it): alternating runs.  Usage: tools/r0_sweep.sh 98 100 98 100 ...
for r0 in "$@"; do
  echo -n "R0=$r0: "
  MFGM_R0=$r0 python bench.py --no-cpu-baseline --no-vdp --no-other-configs --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'fwd', round(r['kernel_ms'],3), [(o['kernel'][11:25], round(o['kernel_ms'],3)) for o in r['other_kernels']])"
done
