#!/bin/bash
# Bench several builds of the library side by side, on the MI355X box: bash tools/exp/bench_libs.sh lib1.so lib2.so ...
# (python tools/exp/build_variant.py <tag> -DSOMETHING=1  ->  tools/exp/lib_<tag>.so)
set -e
for lib in "$@"; do
  echo "== $lib"
  timeout -k 10 120 python tools/exp/bench_lib.py $lib --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['refine'], d['roofline']['avg_launch_ms'], d['roofline']['kernels_ms_per_step'])"
done
