import sys, time, os
sys.path.insert(0, "/root/repo")
import numpy as np
from minrms_amd import engine as eng
from minrms_amd.synth import synth_pair
a, b = synth_pair(150, 150)
for mode in ("device", "host"):
    if mode == "host": os.environ["MINRMS_ALT_HOST_FILTER"] = "1"
    with eng.Engine(eng.make_cfg(150, 150, num_solutions=3, alt_min_3d_difference=3.0, alt_rmsd_tolerance=1.6), a, b) as e:
        og = e.gen_orientations()
        for rep in range(3):
            e.reset()
            t0 = time.perf_counter()
            e.search(og)
            t1 = time.perf_counter()
            print(mode, "search with -alt-o 3,3.0,1.6: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
