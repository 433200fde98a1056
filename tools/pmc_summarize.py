"""
Per-kernel HBM traffic from two rocprofv3 counter passes of bench.py (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE):

    python tools/pmc_summarize.py FETCH.csv WRITE.csv B T d "<mfgm_version() of the library that ran>" > profiles/r02_pmc/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are KB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so HBM read bytes are
2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section).  Medians over the dispatches of each kernel.
"""
import csv
import json
import statistics
import sys


def medians(path, counter):
    per = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == counter and "mfgm::" in row["Kernel_Name"]:
                per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: (statistics.median(v), len(v)) for k, v in per.items()}


def main():
    fetch, write, B, T, d = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    build = sys.argv[6] if len(sys.argv) > 6 else None
    fr, wr = medians(fetch, "FETCH_SIZE"), medians(write, "WRITE_SIZE")
    kernels = {}
    for k, (f_kb, n) in fr.items():
        w_kb = wr.get(k, (0.0, 0))[0]
        kernels[k] = {"dispatches": n, "FETCH_SIZE_KB": f_kb, "WRITE_SIZE_KB": w_kb,
                      "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
                      "read_bytes_per_node": 2.0 * f_kb * 1024.0 / (B * T), "write_bytes_per_node": w_kb * 1024.0 / (B * T)}
    print(json.dumps({
        "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                   "(and a second pass with --pmc WRITE_SIZE); summarised by tools/pmc_summarize.py",
        "workload": {"B": B, "T": T, "d": d},
        "library_build": build,
        "units": "FETCH_SIZE / WRITE_SIZE are KB per dispatch (median over dispatches); on gfx950 FETCH_SIZE counts 64 B per 128-B "
                 "request, so HBM read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section)",
        "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
