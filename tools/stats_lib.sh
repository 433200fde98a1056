#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of one bench configuration under the in-tree library and under an alternative
# build (MFGM_LIB), one box.   usage (through gpurun): bash tools/stats_lib.sh CONFIG path/to/alt.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; C=$1; ALT=$R/$2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in default alt; do
  if [ $v = alt ]; then export MFGM_LIB=$ALT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sl_$v -o p -- python3 $R/bench.py --config $C --steps 20 --warmup 3 --no-cpu-baseline --no-vdp --no-other-configs > $R/gpurun_out/sl_$v.json 2> $R/gpurun_out/sl_$v.err
  echo "== $v (exit $?)"
  f=$(find $R/gpurun_out/sl_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $R/gpurun_out/sl_$v
done
