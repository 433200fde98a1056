#!/bin/bash
# A/B of two builds of the library on one box: the in-tree one vs MFGM_LIB=$1 (a path below the repo root, e.g. a variant built with other -D flags) (alternating, 2 rounds), + fused-kernel averages under the profiler
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A="--steps 20 --warmup 10 --no-cpu-baseline --no-other-configs --no-vdp"
ALT=$PWD/$1
run() { python bench.py $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', round(d['ms_per_step'],4))"; }
for r in 1 2; do run default; MFGM_LIB=$ALT run alt; done
R=$PWD; cd /tmp && export TMPDIR=/tmp
for v in default alt; do
  if [ $v = alt ]; then export MFGM_LIB=$ALT; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$v -o p -- python3 $R/bench.py $A > /dev/null 2>&1
  echo $v; grep -h "k_forward_reduce_cq\|k_forward_cq" $(find $R/gpurun_out/abl_$v -name '*kernel_stats.csv') | cut -d, -f1-4 | cut -c1-90
  rm -rf $R/gpurun_out/abl_$v
done
