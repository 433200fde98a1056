"""Kernel timeline around one occurrence of a kernel, from a rocprofv3 --kernel-trace database (rocpd).
usage: python tools/trace_show.py results.db <kernel substring> [occurrence] [before] [count]"""
import re
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    key = sys.argv[2]
    occ = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    before = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    count = int(sys.argv[5]) if len(sys.argv) > 5 else 30
    rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if key in r[0]]
    i0 = max(0, idx[min(occ, len(idx) - 1)] - before)
    t0 = rows[i0][1]
    for r in rows[i0:i0 + count]:
        nm = re.sub(r"\(.*", "", r[0]).replace("void mfgm::", "")[:48]
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}us  s{r[3]} {nm}")


main()
