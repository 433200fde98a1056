#!/bin/bash
# A/B of the cross-step pipelining on one box: the two-wavefront kernel (default), the two-stream form (VIDP_PIPE_STREAMS=1) and no
# pipelining (VIDP_PIPELINE=0), bench lines + a kernel trace of the default run.   usage (through gpurun): bash tools/ab_pipe.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
A="--steps 20 --warmup 10 --no-cpu-baseline --no-other-configs --no-vdp"
for round in 1 2; do
python bench.py $A > gpurun_out/r04_ab_fused_$round.json 2> gpurun_out/r04_ab.err
VIDP_PIPE_STREAMS=1 python bench.py $A > gpurun_out/r04_ab_streams_$round.json 2>> gpurun_out/r04_ab.err
VIDP_PIPELINE=0 python bench.py $A > gpurun_out/r04_ab_nopipe_$round.json 2>> gpurun_out/r04_ab.err
done
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/r04_prof3 -o p -- python3 $R/bench.py $A > /dev/null 2>&1
ls $R/gpurun_out/r04_prof3
