#!/bin/bash
# A/B of the coarse-level kernels on one box: rocprofv3 kernel statistics of the headline bench with the 16-lanes-per-segment row bodies
# (mfgm_rows.h) at each of the thresholds given (MFGM_COARSE_ROWS: 0 = lane-per-segment bodies only).  Usage: tools/coarse_ab.sh 16 4 1 0
R=/root/repo
O=$R/gpurun_out/coarse_ab
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
    rm -rf $O/st_$v
    MFGM_COARSE_ROWS=$v rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$v -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-vdp --no-other-configs > $O/bench_$v.json 2> /dev/null
    cp "$(find $O/st_$v -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_$v.csv
    rm -rf $O/st_$v
    echo "MFGM_COARSE_ROWS=$v"; grep -i "coarse" $O/kernel_stats_$v.csv | cut -c1-200
done
