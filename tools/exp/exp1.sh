set -e
run() { echo "== $*"; ( export "$@"; timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['refine'], d['roofline']['avg_launch_ms'])" ); }
run X=1
run MINRMS_COOP_SEG_PB=16
run MINRMS_COOP_SEG_TASKS=5
run MINRMS_COOP_SEG_TASKS=3
run MINRMS_REFINE_SPEC_MAX=8
run MINRMS_REFINE_SPEC_MAX=16
