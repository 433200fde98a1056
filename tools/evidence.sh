#!/bin/bash
# End-of-round evidence pass on the GPU box (run through gpurun from the repo root): full GPU test suite, the default bench line (every
# BASELINE configuration in it), rocprofv3 kernel statistics of the headline / c2 / c3 / c5 benches, the FETCH_SIZE / WRITE_SIZE passes
# behind roofline.traffic (headline, c3, c5) and the MFMA-utilisation pass of the config-5 sweeps.  Everything lands under
# gpurun_out/ev/; copy what is to be judged into profiles/ (names per round).
set -o pipefail
ROUND=${ROUND:-r04}
R=/root/repo
O=$R/gpurun_out/ev
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee $O/smoke_rc.txt
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt
python -c "import vidp_amd; print(vidp_amd._lib.load().mfgm_version().decode())" 2>/dev/null > $O/version.txt
V="$(cat $O/version.txt)"
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters, bench arguments...
    local name=$1 ctr=$2; shift 2
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -- python3 $R/bench.py "$@" > $O/pmc_$name.log 2>&1
    cp "$(find $O/pmc_$name -name '*counter_collection.csv' | head -1)" $O/${name}_counter_collection.csv
    rm -rf $O/pmc_$name
}
H="--steps 2 --warmup 1 --no-cpu-baseline --no-vdp --no-other-configs"
pmc h_fetch FETCH_SIZE $H
pmc h_write WRITE_SIZE $H
python3 $R/tools/pmc_summarize.py $O/h_fetch_counter_collection.csv $O/h_write_counter_collection.csv 64 100000 6 "$V" > $O/pmc_traffic.json
mkdir -p $R/profiles/${ROUND}_pmc && cp $O/pmc_traffic.json $R/profiles/${ROUND}_pmc/pmc_traffic.json   # bench.py reads roofline.traffic from it
for c in c3 c5; do
    pmc ${c}_fetch FETCH_SIZE --config $c --steps 2 --warmup 1 --no-cpu-baseline
    pmc ${c}_write WRITE_SIZE --config $c --steps 2 --warmup 1 --no-cpu-baseline
done
python3 $R/tools/pmc_counters.py "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline" 3200000 "$V" $O/c3_fetch_counter_collection.csv $O/c3_write_counter_collection.csv > $O/pmc_traffic_c3.json
python3 $R/tools/pmc_counters.py "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline" 200000 "$V" $O/c5_fetch_counter_collection.csv $O/c5_write_counter_collection.csv > $O/pmc_traffic_c5.json
pmc c5_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" --config c5 --steps 2 --warmup 1 --no-cpu-baseline
python3 $R/tools/pmc_counters.py "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --config c5 --steps 2 --warmup 1 --no-cpu-baseline" 200000 "$V" $O/c5_mfma_counter_collection.csv > $O/pmc_mfma_c5.json
pmc h_sq "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $H
python3 $R/tools/pmc_counters.py "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py $H" 6400000 "$V" $O/h_sq_counter_collection.csv > $O/pmc_sq_headline.json
# kernel statistics (and the line each profiled run prints)
stats() {  # name, bench arguments...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$name -- python3 $R/bench.py "$@" > $O/bench_${name}_profiled.json 2> /dev/null
    cp "$(find $O/st_$name -name '*kernel_stats.csv' | head -1)" $O/kernel_stats_$name.csv
    rm -rf $O/st_$name
}
stats headline --steps 20 --warmup 3 --no-cpu-baseline --no-vdp --no-other-configs
for c in c2 c3 c5; do stats $c --config $c --steps 20 --warmup 3 --no-cpu-baseline; done
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -m pytest tests/test_gpu_accuracy.py -m gpu -s -q 2>&1 | grep -v amdgpu.ids > $O/accuracy_tables.txt
tail -3 $O/pytest_gpu.log; cut -c1-400 $O/bench.json
