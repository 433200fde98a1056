#!/bin/bash
# A/B of one environment knob on one box: alternating bench runs.   usage (through gpurun): bash tools/ab_env.sh NAME VALUE_A VALUE_B [VALUE_C] [-- bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
name=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
A="--steps 30 --warmup 5 --no-cpu-baseline --no-vdp --no-other-configs $*"
for round in 1 2; do
  for v in "${vals[@]}"; do
    env $name=$v python bench.py $A > gpurun_out/ab_${name}_${v}_$round.json 2> gpurun_out/ab_${name}_${v}_$round.err || { tail -5 gpurun_out/ab_${name}_${v}_$round.err; exit 1; }
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_${name}_${v}_$round.json").read().strip().splitlines()[-1])
print("$name=$v round $round: %.4f ms/step" % d["ms_per_step"])
PY
  done
done
