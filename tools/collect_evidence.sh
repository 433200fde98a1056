#!/bin/bash
# Copy what tools/evidence.sh left under gpurun_out/ev/ into profiles/ under this round's names (run here, after the gpurun call).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/ev
P=$R/profiles
cp $E/bench.json $P/r03_a_bench.json
cp $E/bench_headline_profiled.json $P/r03_a_bench_profiled.json
cp $E/kernel_stats_headline.csv $P/r03_a_kernel_stats.csv
for c in c2 c3 c5; do
    cp $E/bench_${c}_profiled.json $P/r03_${c}_bench_profiled.json
    cp $E/kernel_stats_$c.csv $P/r03_${c}_kernel_stats.csv
done
mkdir -p $P/r03_pmc $P/r03_mfma
cp $E/pmc_traffic.json $E/pmc_traffic_c3.json $E/pmc_traffic_c5.json $E/pmc_sq_headline.json $P/r03_pmc/
cp $E/h_fetch_counter_collection.csv $P/r03_pmc/fetch_counter_collection.csv
cp $E/h_write_counter_collection.csv $P/r03_pmc/write_counter_collection.csv
cp $E/pmc_mfma_c5.json $E/accuracy_tables.txt $P/r03_mfma/
cat $E/version.txt
