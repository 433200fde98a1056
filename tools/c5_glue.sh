#!/bin/bash
# which torch (non-library) kernels run inside the config-5 step: rocprofv3 kernel statistics of bench.py --config c5, at::native rows only
R=/root/repo
O=$R/gpurun_out/c5_glue
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/st
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench.json 2>/dev/null
cp "$(find $O/st -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
rm -rf $O/st
python3 - <<'PY'
import csv
rows = list(csv.reader(open('/root/repo/gpurun_out/c5_glue/kernel_stats.csv')))
for r in rows[1:40]:
    if int(r[1]) >= 20:
        print(r[0][:150].ljust(150), r[1], round(float(r[3]) / 1e3, 1), 'us')
PY
