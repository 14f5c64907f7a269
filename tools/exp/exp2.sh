set -e
run() { echo "== $*"; ( export "$@"; timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['refine'], d['roofline']['avg_launch_ms'], d['roofline']['kernels_ms_per_step'])" ); }
run X=1
