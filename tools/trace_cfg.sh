#!/bin/bash
# kernel trace of one bench configuration: bash tools/trace_cfg.sh c5  (through gpurun); leaves gpurun_out/trace_<cfg>/p_results.db
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; C=${1:-c5}
mkdir -p gpurun_out
python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trace_$C.json 2> gpurun_out/trace_$C.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$C -o p -- python3 $R/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
ls $R/gpurun_out/trace_$C
