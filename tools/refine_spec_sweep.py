"""Refinement time against the number of entries prepared per round (MINRMS_REFINE_SPEC)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minrms_amd import engine as eng
from minrms_amd.synth import synth_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
a, b = synth_pair(n, n)
e = eng.Engine(eng.make_cfg(n, n), a, b)
og = e.gen_orientations()
for spec in ["adaptive", "4", "8", "16", "32", "64", "100000"]:
    if spec == "adaptive":
        os.environ.pop("MINRMS_REFINE_SPEC", None)
    else:
        os.environ["MINRMS_REFINE_SPEC"] = spec
    for rep in range(2):
        e.reset(); e.search(og); e.finalize_search()
        t0 = time.perf_counter(); st = e.refine(); t1 = time.perf_counter()
    print("spec %-8s refine %.1f ms  iters %d rounds %d dp_runs %d" % (spec, (t1 - t0) * 1e3, st.num_iterations, st.rounds, st.dp_runs), flush=True)
