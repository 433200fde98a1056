#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of one bench configuration.   usage (through gpurun): bash tools/stats_cfg.sh CONFIG [N]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; C=${1:-c3}; N=${2:-12}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sc_$C -o p -- python3 $R/bench.py --config $C --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/sc_$C.json 2> $R/gpurun_out/sc_$C.err
f=$(find $R/gpurun_out/sc_$C -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$N" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2])]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $R/gpurun_out/sc_$C
