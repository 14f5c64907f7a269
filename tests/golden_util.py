"""Helpers shared by the parity tests (test infrastructure)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def cfg_kwargs(g):
    """Arguments for oracle_py.make_cfg / minrms_amd engine from a golden."""
    return dict(N1=int(g["N1"]), N2=int(g["N2"]), nbb=int(g["nbb"]),
                n_min=int(g["n_min"]), n_max=int(g["n_max"]),
                max_rmsd=float(g["max_rmsd"]), max_dist=float(g["max_dist"]),
                refine_while_searching=bool(int(g["rws"])),
                of_enable=bool(int(g["of"])), of_max_dist=float(g["of_dist"]),
                of_min_matches=int(g["of_min"]))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_tables_equal(ssd, xform, pairs, g, tag, n_min):
    """Bit-exact comparison of result tables with golden tables `tag`."""
    gs, gx, gp = g[tag + "_ssd"], g[tag + "_xform"], g[tag + "_pairs"]
    assert np.array_equal(bits(ssd), bits(gs)), "ssd bits differ: %s" % (
        np.nonzero(bits(ssd) != bits(gs))[0] + n_min)
    ok = gs < np.float32(1e26)
    assert np.array_equal(bits(xform)[ok], bits(gx)[ok]), "transform bits differ"
    for k in np.nonzero(ok)[0]:
        n = n_min + k
        assert np.array_equal(pairs[k, :n], gp[k, :n]), "pairs differ at n=%d" % n
