#!/usr/bin/env python3
"""How the sweep's traceback + refit kernel divides: with refine_while_searching off the kernel only
walks (and, to make it run at all, the per-orientation update counts are switched on)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from minrms_amd import engine as eng          # noqa: E402
from minrms_amd.synth import synth_pair       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
if len(sys.argv) > 2:
    eng.LIB_PATH = os.path.abspath(sys.argv[2])      # another build of the library
a, b = synth_pair(n, n)
for rws in (True, False):
    with eng.Engine(eng.make_cfg(n, n, refine_while_searching=rws), a, b) as e:
        og = e.gen_orientations()
        e.count_updates()
        e.enable_timing(2)
        for rep in range(3):
            e.reset()
            e.search(og)
            if rep == 0:
                e.kernel_time(e.KC_TRACE)
        ms, cnt = e.kernel_time(e.KC_TRACE)
        print("%dx%d refit %s: k_trace_fit %.3f ms per sweep (%d launches)" % (n, n, "on" if rws else "off (walk only)", ms / 2, cnt // 2), flush=True)
