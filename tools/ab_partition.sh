#!/bin/bash
# development aid: headline step (cq kernels) against the level-0 segment length and the coarse-level shape
run() { python bench.py --no-cpu-baseline --no-vdp --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['ms_per_step'],3), 'ms/step', d['config']['partition'], [(k['kernel'][12:24], round(k['kernel_ms'],3)) for k in [r]+r['other_kernels']])"; }
echo -n "default: "; run
for r0 in 49 64 80 128 196; do echo -n "R0=$r0: "; MFGM_R0=$r0 run; done
for cfg in "3 8" "4 8" "5 12" "6 12" "4 16"; do set -- $cfg; echo -n "Rup=$1 top=$2: "; MFGM_RUP=$1 MFGM_TOP=$2 run; done
for fp in 64 1024; do echo -n "FUSE_P=$fp: "; MFGM_FUSE_P=$fp run; done
echo -n "per-level coarse: "; MFGM_COARSE_FUSED=0 run
echo -n "default: "; run
