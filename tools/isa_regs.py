"""Register / scratch usage per kernel from a `hipcc -S --cuda-device-only` listing (the amdhsa.kernels metadata at its end).
usage: python tools/isa_regs.py file.s [substring ...]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    pats = sys.argv[2:]
    meta = txt[txt.rfind("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void mfgm::", "")
        if pats and not any(p in dem for p in pats):
            continue
        agpr = blk.split("\n")[0].strip()
        print(f"{dem:60s} vgpr {g('vgpr_count'):>4s} (agpr {agpr:>3s}) sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} "
              f"spill v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}")


main()
