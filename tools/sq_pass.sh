#!/bin/bash
# SQ activity counters of the headline bench kernels (one --pmc pass); bash tools/sq_pass.sh  -> gpurun_out/sq_pass.json
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; O=$R/gpurun_out; mkdir -p $O
V="$(python -c "import vidp_amd; print(vidp_amd._lib.load().mfgm_version().decode())" 2>/dev/null)"
H="--steps 3 --warmup 2 --no-cpu-baseline --no-vdp --no-other-configs"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sqp -- python3 $R/bench.py $H > $O/sqp.log 2>&1
cp "$(find $O/sqp -name '*counter_collection.csv' | head -1)" $O/sq_counter_collection.csv; rm -rf $O/sqp
python3 $R/tools/pmc_counters.py "sq pass" 6400000 "$V" $O/sq_counter_collection.csv > $O/sq_pass.json
python3 - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/sq_pass.json"))
for k,v in d.get("kernels",d).items():
    if not isinstance(v,dict) or "SQ_WAVE_CYCLES" not in v: continue
    if not any(x in k for x in ("_cq<6","forward_reduce")): continue
    w=v["SQ_WAVE_CYCLES"]
    print(k[:48].ljust(48), "active %.2f wait_inst %.2f wait_any %.2f  valu/wave-cycle %.3f  gui %.0f" % (v["SQ_ACTIVE_INST_ANY"]/w, v["SQ_WAIT_INST_ANY"]/w, v["SQ_WAIT_ANY"]/w, v["SQ_INSTS_VALU"]/w, v["GRBM_GUI_ACTIVE"]))
PY
