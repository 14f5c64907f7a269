#!/usr/bin/env python3
"""GPU time of one job by kernel class (HIP events on the engine's stream).

usage: kernel_times.py [n1 [n2 [minN [max_rmsd [reps]]]]]   (minN: fraction < 1 or count, 0 = off; max_rmsd 0 = off)
Level-2 timing puts an event between the kernels of every refinement round and lengthens the rounds by a
few per cent; the wall times printed beside are taken in a pass without events."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minrms_amd import engine as eng          # noqa: E402
from minrms_amd.synth import synth_pair       # noqa: E402

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else n1
kw = {}
if len(sys.argv) > 3 and float(sys.argv[3]) > 0:
    v = float(sys.argv[3])
    kw["n_min"] = int(v * min(n1, n2) + 0.5) if v < 1 else int(v)
if len(sys.argv) > 4 and float(sys.argv[4]) > 0:
    kw["max_rmsd"] = float(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
a, b = synth_pair(n1, n2)
E = eng.Engine
names = [("k_cij sweep", E.KC_CIJ), ("k_trace_fit sweep", E.KC_TRACE), ("k_reduce", E.KC_REDUCE),
         ("k_cij small", E.KC_COOP_CIJ), ("k_dp_coop", E.KC_COOP), ("k_trace_fit_wave", E.KC_TRACE_WAVE),
         ("k_refine_round", E.KC_ROUND), ("other", E.KC_OTHER)]
with eng.Engine(eng.make_cfg(n1, n2, **kw), a, b) as e:
    og = e.gen_orientations()
    for level in (0, 2):
        e.enable_timing(level)
        wall = [0.0, 0.0, 0.0]
        for rep in range(reps + 1):
            e.reset()
            t0 = time.perf_counter()
            e.search(og)
            t1 = time.perf_counter()
            e.finalize_search()
            t2 = time.perf_counter()
            st = e.refine()
            t3 = time.perf_counter()
            if rep == 0:                       # warm-up
                for _, c in names:
                    e.kernel_time(c)
                for blk in range(e.num_blocks()):
                    e.dp_block_time(blk)
                continue
            wall = [wall[0] + t1 - t0, wall[1] + t2 - t1, wall[2] + t3 - t2]
        print("%dx%d %s, %d orientations, timing level %d: search %.2f ms, winners %.2f ms, refine %.2f ms (%d iterations, %d rounds, %d tables)"
              % (n1, n2, kw, og.shape[0], level, wall[0] / reps * 1e3, wall[1] / reps * 1e3, wall[2] / reps * 1e3,
                 st.num_iterations, st.rounds, st.dp_runs), flush=True)
    e.enable_timing(0)
    for blk in range(e.num_blocks()):
        ms, n, o, c = e.dp_block_time(blk)
        if n:
            print("  k_dp_panel block %d: %.3f ms per job, %.1f launches, %.1f us each" % (blk, ms / reps, n / reps, ms / n * 1e3))
    for name, c in names:
        ms, n = e.kernel_time(c)
        if n:
            print("  %-18s %.3f ms per job, %.1f launches, %.1f us each" % (name + ":", ms / reps, n / reps, ms / n * 1e3))
