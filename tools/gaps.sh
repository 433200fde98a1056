#!/bin/bash
# GPU idle time inside the timed steps of a bench configuration: rocprofv3 kernel trace, then the gaps between consecutive kernels in
# the middle of the trace (tools/trace_window.py) and their sum.  Usage: tools/gaps.sh c5 [dispatches in the window]
R=/root/repo
c=${1:-c5}; w=${2:-160}
O=$R/gpurun_out/gaps_$c
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/tr
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/bench.json 2>/dev/null || \
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2>/dev/null
f="$(find $O/tr -name '*kernel_trace.csv' | head -1)"
python3 $R/tools/trace_window.py "$f" $w > $O/window.txt
python3 - "$O/window.txt" <<'PY'
import sys
rows = [l.split() for l in open(sys.argv[1])]
dur = sum(float(r[3]) for r in rows); gap = sum(float(r[5]) for r in rows[1:])
big = sorted(((float(r[5]), " ".join(r[6:])[:70]) for r in rows[1:]), reverse=True)[:8]
print(f"window: {len(rows)} kernels, busy {dur:.0f} us, idle {gap:.0f} us ({100 * gap / (dur + gap):.1f} %)")
for g, n in big: print(f"  gap {g:7.1f} us before {n}")
PY
rm -rf $O/tr
