for cfg in "0 0" "98 8" "98 16" "64 8" "64 16" "128 16" "128 8" "196 14" "49 8"; do
  set -- $cfg
  echo "== R0=$1 Rup=$2"
  timeout -k 10 120 python tools/perf_probe.py 64 100000 6 $1 $2 2>/dev/null | tail -3
done
