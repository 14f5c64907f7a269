#!/bin/bash
# run the drop-in binary on the 41x13 fixture pair under several engine settings
set -u
T=$(mktemp -d); cd $T
python3 - <<PY
import json
g=json.load(open("$GRAFT_REPO_ROOT/tests/golden/msf_ship41x13.json"))
open("a.pdb","w").write(g["a_pdb"]); open("b.pdb","w").write(g["b_pdb"])
open("want.plot","w").write(g["files"]["align_chimera.plot"])
PY
for mode in default default default nocoop seqrefine; do
  rm -f align*
  case $mode in
    nocoop) export MINRMS_NO_COOP=1;;
    seqrefine) unset MINRMS_NO_COOP; export MINRMS_REFINE_SEQUENTIAL=1;;
  esac
  $GRAFT_REPO_ROOT/oracle/_ref/minrms_hip -fm 4 -HS -ir -r a.pdb b.pdb > log.txt 2>&1
  echo "mode=$mode rc=$? plot_equal=$(cmp -s align_chimera.plot want.plot && echo yes || echo NO)  $(grep -c . align_chimera.plot) lines; $(grep 'During' log.txt)"
done
diff align_chimera.plot want.plot | head -5
