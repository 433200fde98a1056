#!/bin/bash
# development aid: headline step time against the partition (MFGM_R0 level-0 segment length, MFGM_RUP segment length above level 0,
# MFGM_TOP sequential-top size); the default first and last (drift of the box).  Run through gpurun from the repo root.
run() {
  echo -n "R0=${1:-auto} Rup=$2 top=$3: "
  env ${1:+MFGM_R0=$1} MFGM_RUP=$2 MFGM_TOP=$3 python bench.py --no-cpu-baseline --no-vdp --no-other-configs --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['roofline']['step']; print(round(d['ms_per_step'],3), 'ms/step  coarse f/b', round(s['coarse_factor_ms_per_refresh'],3), round(s['coarse_backward_ms_per_refresh'],3), d['config'].get('partition'))"
}
run "" 4 12
for cfg in "3 12" "5 12" "6 12" "8 12" "4 4" "4 8" "4 16" "4 24" "3 8" "2 8" "8 48"; do set -- $cfg; run "" $1 $2; done
for r0 in 64 80 112 128 160; do run $r0 4 12; done
run "" 4 12
