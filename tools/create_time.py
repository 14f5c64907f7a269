#!/usr/bin/env python3
"""Time of engine creation (device allocations) for several workspace sizes.
usage: create_time.py n1 n2 gib [gib ...]"""
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from minrms_amd import engine as eng          # noqa: E402
from minrms_amd.synth import synth_pair       # noqa: E402

n1, n2 = int(sys.argv[1]), int(sys.argv[2])
a, b = synth_pair(n1, n2)
e = eng.Engine(eng.make_cfg(8, 8), a[:8], b[:8])      # device start-up outside the timings
e.close()
for g in sys.argv[3:]:
    for rep in range(2):
        t0 = time.perf_counter()
        e = eng.Engine(eng.make_cfg(n1, n2, workspace_bytes=int(float(g) * 2**30)), a, b)
        t1 = time.perf_counter()
        e.close()
        t2 = time.perf_counter()
        print(f"{n1}x{n2} workspace {g} GiB: create {1e3 * (t1 - t0):.1f} ms, close {1e3 * (t2 - t1):.1f} ms", flush=True)
    eng.trim()
