#!/bin/bash
# Copy what tools/evidence.sh left under gpurun_out/ev/ into profiles/ under this round's names (run here, after the gpurun call).
set -e
ROUND=${ROUND:-r04}
R=$(cd "$(dirname "$0")/.." && pwd)
E=$R/gpurun_out/ev
P=$R/profiles
cp $E/bench.json $P/${ROUND}_a_bench.json
cp $E/bench_headline_profiled.json $P/${ROUND}_a_bench_profiled.json
cp $E/kernel_stats_headline.csv $P/${ROUND}_a_kernel_stats.csv
for c in c2 c3 c5; do
    cp $E/bench_${c}_profiled.json $P/${ROUND}_${c}_bench_profiled.json
    cp $E/kernel_stats_$c.csv $P/${ROUND}_${c}_kernel_stats.csv
done
mkdir -p $P/${ROUND}_pmc $P/${ROUND}_mfma
cp $E/pmc_traffic.json $E/pmc_traffic_c3.json $E/pmc_traffic_c5.json $E/pmc_sq_headline.json $P/${ROUND}_pmc/
cp $E/h_fetch_counter_collection.csv $P/${ROUND}_pmc/fetch_counter_collection.csv
cp $E/h_write_counter_collection.csv $P/${ROUND}_pmc/write_counter_collection.csv
cp $E/pmc_mfma_c5.json $E/accuracy_tables.txt $P/${ROUND}_mfma/
cat $E/version.txt
