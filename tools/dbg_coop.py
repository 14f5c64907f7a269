import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from minrms_amd import engine as eng
from minrms_amd.synth import synth_pair
n1, n2 = 70, 80
a, b = synth_pair(n1, n2, 101, 202)
def run():
    e = eng.Engine(eng.make_cfg(n1, n2), a, b)
    og = e.gen_orientations()
    e.search(og)
    s0 = e.results()[0].copy()
    st = e.refine()
    s1, xf, pairs, win = e.results()
    e.close()
    return s0, s1.copy(), st.num_iterations, st.rounds
os.environ["MINRMS_NO_COOP"] = "1"
ref = run()
del os.environ["MINRMS_NO_COOP"]
for rep in range(5):
    r = run()
    print("rep", rep, "search equal", np.array_equal(ref[0].view(np.uint32), r[0].view(np.uint32)),
          "refine equal", np.array_equal(ref[1].view(np.uint32), r[1].view(np.uint32)),
          "iters", ref[2], r[2], "rounds", ref[3], r[3],
          "first diff n", (np.nonzero(ref[1].view(np.uint32) != r[1].view(np.uint32))[0][:5] + 1))
