#!/bin/bash
# Same-box A/B of two builds of the library (MFGM_LIB): alternating headline bench runs.  Usage: tools/layout_ab.sh libA.so libB.so
R=/root/repo
O=$R/gpurun_out/layout_ab
mkdir -p $O
for rep in 1 2 3; do
    for lib in "$@"; do
        n=$(basename $lib .so)
        MFGM_LIB=$R/$lib python3 $R/bench.py --no-cpu-baseline --no-vdp --no-other-configs > $O/${n}_$rep.json 2>/dev/null
        python3 - "$O/${n}_$rep.json" "$n" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
r = d["roofline"]
ks = [(r["kernel"].split("(")[0][-16:], r["kernel_ms"])] + [(o["kernel"].split("(")[0][-16:], o["kernel_ms"]) for o in r["other_kernels"]]
st = r["step"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], " ".join("%s %.3f" % (k.split("::")[-1], v) for k, v in ks),
      "coarse f %.3f b %.3f" % (st["coarse_factor_ms_per_refresh"], st["coarse_backward_ms_per_refresh"]))
PY
    done
done
