#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the headline loop under each value of one environment knob, one box.
# usage (through gpurun): bash tools/stats_env.sh NAME VALUE_A VALUE_B ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; name=$1; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export $name=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stats_${name}_$v -o p -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vdp --no-other-configs > $R/gpurun_out/stats_${name}_$v.json 2> $R/gpurun_out/stats_${name}_$v.err
  f=$(find $R/gpurun_out/stats_${name}_$v -name '*kernel_stats.csv' | head -1)
  echo "== $name=$v"; python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-60s calls %5s avg %9.1f us  total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
