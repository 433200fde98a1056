for cfg in "0 0" "16 8" "32 8" "64 8" "16 16" "8 8" "32 16" "24 6"; do
  set -- $cfg
  echo "== R0=$1 Rup=$2"
  timeout -k 10 120 python tools/perf_probe.py 1 100000 3 $1 $2 2>/dev/null | tail -3
done
