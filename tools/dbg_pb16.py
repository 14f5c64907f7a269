import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from golden_util import load, cfg_kwargs
from minrms_amd import engine as eng
g = load("syn60x50_altr2IR")
k = cfg_kwargs(g)
def run():
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"], max_rmsd=k["max_rmsd"],
                       max_dist=k["max_dist"], refine_while_searching=k["refine_while_searching"])
    with eng.Engine(cfg, g["xyza"], g["xyzb"]) as e:
        e.search(g["og"][:50])
        e.finalize_search()
        return e.results()
os.environ["MINRMS_COOP_PB"] = "32"; a = run()
os.environ["MINRMS_COOP_PB"] = "16"; b = run()
print("rws", k["refine_while_searching"])
bad = np.nonzero(a[0].view(np.uint32) != b[0].view(np.uint32))[0]
print("ssd differ at n =", bad + k["n_min"])
for kk in bad[:4]:
    print(kk + k["n_min"], a[0][kk], b[0][kk], "winner", a[3][kk], b[3][kk])
    n = kk + k["n_min"]
    print(" pairs32", a[2][kk, :n].tolist()); print(" pairs16", b[2][kk, :n].tolist())
