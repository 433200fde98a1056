#!/bin/bash
# A/B of the in-tree library against MFGM_LIB=$1 on the bench configurations $2..., one box, two alternating rounds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ALT=$PWD/$1; shift
for c in "$@"; do
  A="--config $c --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-vdp"
  [ "$c" = "dense" ] && A="--dense --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-vdp"
  for r in 1 2; do
    for v in default alt; do
      if [ $v = alt ]; then export MFGM_LIB=$ALT; else unset MFGM_LIB; fi
      python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$c $v', round(d['ms_per_step'],4))"
    done
  done
done
