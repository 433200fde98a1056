#!/bin/bash
# End-of-round evidence pass on the GPU box (run through gpurun from the repo root): full GPU test suite, the default bench line, the
# other configurations, rocprofv3 kernel statistics and the two PMC passes behind roofline.traffic.  Everything lands under gpurun_out/ev/.
set -o pipefail
R=/root/repo
O=$R/gpurun_out/ev
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $O/pytest_rc.txt
python -c "import vidp_amd; print(vidp_amd._lib.load().mfgm_version().decode())" 2>/dev/null > $O/version.txt
cd /tmp && export TMPDIR=/tmp
# PMC passes first (their summary must be in profiles/ when the bench line is printed)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vdp > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vdp > $O/pmc_write.log 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
mkdir -p $R/profiles/r02_pmc
python3 $R/tools/pmc_summarize.py "$F" "$W" 64 100000 6 "$(cat $O/version.txt)" > $R/profiles/r02_pmc/pmc_traffic.json
cp "$F" $O/fetch_counter_collection.csv; cp "$W" $O/write_counter_collection.csv; cp $R/profiles/r02_pmc/pmc_traffic.json $O/pmc_traffic.json
# kernel statistics of the headline bench and the line it prints
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-vdp > $O/bench_profiled.json 2> $O/bench_profiled.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for c in c1 c2 c3 c5; do python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --config c5 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c5_profiled.json 2> /dev/null
cp $(find $O/stats_c5 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c5.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $R/bench.py --config c3 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_c3_profiled.json 2> /dev/null
cp $(find $O/stats_c3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_c3.csv
cd $R
{ for M in 0 1; do echo "== moments_only=$M (d=16)"; PROBE_MOMENTS_ONLY=$M python tools/perf_probe_wide.py 1 200000 16; echo "== moments_only=$M (d=30)"; PROBE_MOMENTS_ONLY=$M python tools/perf_probe_wide.py 1 20000 30; done; } 2>&1 | grep -v amdgpu.ids > $O/mfma_forms.txt
python -m pytest tests/test_gpu_accuracy.py -m gpu -s -q 2>&1 | grep -v amdgpu.ids > $O/accuracy_tables.txt
rm -rf $O/pmc_fetch $O/pmc_write $O/stats $O/stats_c5 $O/stats_c3
tail -3 $O/pytest_gpu.log; cat $O/bench.json | cut -c1-400
