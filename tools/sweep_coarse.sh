#!/bin/bash
# development aid: headline step time against the coarse-level shape (segment length above level 0, sequential top size)
for cfg in "8 48" "4 12" "4 8" "3 8" "2 8" "5 16" "4 16" "3 12" "4 4" "8 48"; do
  set -- $cfg
  echo -n "Rup=$1 top=$2: "
  MFGM_RUP=$1 MFGM_TOP=$2 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), 'ms/step', d['config']['partition'])"
done
