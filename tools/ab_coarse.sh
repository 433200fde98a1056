#!/bin/bash
# development aid: headline step with the fused coarse-level kernels on / off and a few coarse-level shapes
run() { python bench.py --no-cpu-baseline --no-vdp --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', d['config']['partition'], 'elbo', d['elbo_last'])"; }
echo -n "fused default: "; run
echo -n "per-level (MFGM_COARSE_FUSED=0): "; MFGM_COARSE_FUSED=0 run
for cfg in "2 4" "2 8" "3 8" "3 12" "4 4" "4 12" "6 12" "8 16" "16 16"; do
  set -- $cfg
  echo -n "fused Rup=$1 top=$2: "; MFGM_RUP=$1 MFGM_TOP=$2 run
done
echo -n "fused default again: "; run
