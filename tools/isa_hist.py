"""Instruction histogram per basic block of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: python tools/isa_hist.py file.s <mangled-name substring> [min instructions per block]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    start = next(i for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l) and key in l)
    end = next(i for i in range(start, len(txt)) if ".end_amdhsa_kernel" in txt[i] or txt[i].startswith("\t.section") and i > start + 5)
    lab, per, order = "entry", collections.defaultdict(collections.Counter), ["entry"]
    for l in txt[start + 1:end]:
        t = l.strip()
        if re.match(r"\.LBB\d+_\d+:", t):
            lab = t
            order.append(lab)
            continue
        if not t or t.startswith((";", ".")):
            continue
        op = t.split()[0]
        grp = ("fp64" if re.match(r"v_(fma|mul|add|rcp|rsq|div|max|min|fmac)_f64", op) else
               "accvgpr" if op.startswith("v_accvgpr") else
               "scratch" if op.startswith("scratch_") else
               "gload" if op.startswith("global_load") else "gstore" if op.startswith("global_store") else
               "lds" if op.startswith("ds_") else "waitcnt" if op == "s_waitcnt" else
               "salu" if op.startswith("s_") else "valu_other")
        per[lab][grp] += 1
    tot = collections.Counter()
    for lab in order:
        c = per[lab]
        n = sum(c.values())
        tot.update(c)
        if n >= floor:
            print(f"{lab:14s} {n:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("total", sum(tot.values()), dict(tot))


main()
