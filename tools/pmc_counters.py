"""
Per-kernel medians of the counters in one or more rocprofv3 counter_collection.csv files (development / evidence aid):

    python tools/pmc_counters.py "<command that was profiled>" nodes_per_launch "<mfgm_version()>" A_counter_collection.csv [B_...csv ...]

Prints JSON: for every `mfgm::` kernel the median of each counter over its dispatches, plus, when present,
  hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024   (KB counters; gfx950 read-side doubling, MI355X_MICROARCH.md)
  read_bytes_per_node / write_bytes_per_node                         (nodes_per_launch > 0)
  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)  (GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs)
"""
import csv
import json
import statistics
import sys


def main():
    command, nodes, build, files = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4:]
    per = {}
    for path in files:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if "mfgm::" in row["Kernel_Name"]:
                    per.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out = {}
    for k, cs in sorted(per.items()):
        o = {c: statistics.median(v) for c, v in cs.items()}
        o["dispatches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in o or "WRITE_SIZE" in o:
            rd, wr = 2.0 * o.get("FETCH_SIZE", 0.0) * 1024.0, o.get("WRITE_SIZE", 0.0) * 1024.0
            o["hbm_bytes_per_launch"] = rd + wr
            if nodes > 0:
                o["read_bytes_per_node"], o["write_bytes_per_node"] = rd / nodes, wr / nodes
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o.get("GRBM_GUI_ACTIVE", 0) > 0:
            o["mfma_util"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        out[k] = o
    print(json.dumps({"command": command, "nodes_per_launch": nodes, "library_build": build,
                      "units": "FETCH_SIZE / WRITE_SIZE in KB per dispatch, read side doubled on gfx950; SQ_* summed over the chip; "
                               "GRBM_GUI_ACTIVE summed over the 8 XCDs", "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
