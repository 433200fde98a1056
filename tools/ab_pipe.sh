#!/bin/bash
# A/B of the cross-step pipelining on one box: pipelined vs VIDP_PIPELINE=0 bench lines + a kernel trace of the pipelined run.
# usage (through gpurun, from the repo root): bash tools/ab_pipe.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_api.py -x -q -k "pipelined or full_size_steps" > gpurun_out/r04_t2.log 2>&1; tail -3 gpurun_out/r04_t2.log
python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r04_b2.json 2> gpurun_out/r04_b2.err
VIDP_PIPELINE=0 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/r04_b2_nopipe.json 2>> gpurun_out/r04_b2.err
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_prof2 -o p -- python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
ls $R/gpurun_out/r04_prof2
