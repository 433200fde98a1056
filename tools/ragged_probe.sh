#!/bin/bash
# development aid: is the sensitivity of the memory-bound sweeps to the level-0 segment length about ragged last segments or about the
# tile stride?  (T, R0) pairs with and without a ragged segment, then a finer sweep of R0.
run() { echo -n "T=$1 R0=$2: "; MFGM_R0=$2 python bench.py --T $1 --no-cpu-baseline --no-vdp --no-other-configs --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'lanes', d['config']['partition']['lanes'], 'fwd', round(r['kernel_ms'],3), [(o['kernel'][11:25], round(o['kernel_ms'],3)) for o in r['other_kernels']])"; }
for r0 in "$@"; do run 100000 $r0; done
