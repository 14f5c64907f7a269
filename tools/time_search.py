#!/usr/bin/env python3
"""Time Search (the orientation sweep) and Refine of one synthetic pair.

usage: time_search.py [n1 [n2 [0|1 refine_while_searching [workspace GiB [minN [max_rmsd [max_dist]]]]]]]
minN as the reference's -minN (a fraction < 1 of the shorter structure, or a count); 0 = off for the cutoffs.
MINRMS_NO_EARLY_EXIT=1 in the environment gives the sweep without early termination for comparison.
"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from minrms_amd import engine as eng          # noqa: E402
from minrms_amd.search_or import SearchOrHip  # noqa: E402
from minrms_amd.synth import synth_pair       # noqa: E402


def main():
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    n2 = int(sys.argv[2]) if len(sys.argv) > 2 else n1
    rws = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    a, b = synth_pair(n1, n2)
    kw = {"workspace_bytes": int(float(sys.argv[4]) * 2**30)} if len(sys.argv) > 4 and float(sys.argv[4]) > 0 else {}
    if len(sys.argv) > 5 and float(sys.argv[5]) > 0:
        v = float(sys.argv[5])
        kw["n_min"] = int(v * min(n1, n2) + 0.5) if v < 1 else int(v)      # floor(f * N + 0.5), minrms_parser.cc:53-62
    if len(sys.argv) > 6 and float(sys.argv[6]) > 0:
        kw["max_rmsd"] = float(sys.argv[6])
    if len(sys.argv) > 7 and float(sys.argv[7]) > 0:
        kw["max_dist"] = float(sys.argv[7])
    cfg = eng.make_cfg(n1, n2, refine_while_searching=rws, **kw)
    for rep in range(2):
        so = SearchOrHip(cfg, a, b)
        og = so.generate_orientations(frag_len=4)
        t0 = time.perf_counter()
        so.Search(og)
        t1 = time.perf_counter()
        it = so.Refine().num_iterations
        t2 = time.perf_counter()
        ssd = so.tables()[0]
        so.close()
        print(f"{n1}x{n2} rws={int(rws)} {kw} search {1e3 * (t1 - t0):.1f} ms  refine {1e3 * (t2 - t1):.1f} ms "
              f"({it} iterations)  checksum {float(np.sum(ssd[ssd < eng.UPPER_BOUND])):.4f}", flush=True)


if __name__ == "__main__":
    main()
