#!/bin/bash
# A/B of environment hooks on the bench workload, on the MI355X box: bash tools/exp/bench_env.sh VAR=value [VAR=value ...] ...
# (one bench run per argument group separated by "--"; without arguments: the default build only)
set -e
run() { echo "== $*"; ( [ $# -gt 0 ] && export "$@"; timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['refine'], d['roofline']['avg_launch_ms'], d['roofline']['kernels_ms_per_step'])" ); }
group=()
for a in "$@"; do
  if [ "$a" = "--" ]; then run "${group[@]}"; group=(); else group+=("$a"); fi
done
run "${group[@]}"
