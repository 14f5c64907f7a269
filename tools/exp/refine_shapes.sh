#!/bin/bash
# Refinement time of one pair under different shapes of the cooperative kernel's segments, on the MI355X box:
#   bash tools/exp/refine_shapes.sh N1 N2 "VAR=value ..." "VAR=value ..." ...      (one run per quoted group; "" = default)
set -e
n1=$1; n2=$2; shift 2
for g in "$@"; do
  echo "== $g"
  ( [ -n "$g" ] && export $g; timeout -k 10 300 python tools/time_search.py $n1 $n2 2>&1 | tail -1 )
done
