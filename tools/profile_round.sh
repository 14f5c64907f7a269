#!/bin/bash
# Measurement set of one round, run on the MI355X box from the repo root:
#   bash tools/profile_round.sh <tag>      (writes gpurun_out/<tag>_*)
# One bench line, one kernel-trace summary, the two HBM counter passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass) and one pass of SQ counters
# (vector instructions issued, cycles the vector ALU was busy, wave lifetimes and
# the share of them spent in s_waitcnt), each step under its own limit.
set -e -o pipefail
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out
mkdir -p "$out"
timeout -k 10 300 python3 "$root/bench.py" > "$out/${tag}_bench.json"
cat "$out/${tag}_bench.json"
cd /tmp && export TMPDIR=/tmp
# the profiled runs sweep on ONE stream: with the two streams of the default the launches of the two
# halves of a batch overlap, and a kernel's duration in the trace is no longer its own
export MINRMS_SWEEP_STREAMS=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/${tag}_trace" -- \
  python3 "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$out/${tag}_trace_bench.json"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${tag}_fetch" -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${tag}_write" -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES \
  --output-format csv -d "$out/${tag}_sq" -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > /dev/null
cd "$root"
python3 tools/pmc_sum.py "$out/${tag}_fetch" FETCH_SIZE > "$out/${tag}_pmc_fetch_size.csv"
python3 tools/pmc_sum.py "$out/${tag}_write" WRITE_SIZE > "$out/${tag}_pmc_write_size.csv"
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES; do
  python3 tools/pmc_sum.py "$out/${tag}_sq" $c > "$out/${tag}_pmc_$c.csv"
done
python3 tools/pmc_block0.py "$out" "$tag" > "$out/${tag}_pmc_dp_block0.json"
find "$out/${tag}_trace" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \;
rm -rf "$out/${tag}_fetch" "$out/${tag}_write" "$out/${tag}_sq"
echo "profile set $tag done"
