"""HIP engine (through the C ABI) against the golden vectors of the reference
and against the CPU oracle.  Bit-exact: every float is compared by bits."""
import numpy as np
import pytest

from golden_util import load, cfg_kwargs, bits, assert_tables_equal

pytestmark = pytest.mark.gpu

SMALL = ["ship41x13_full", "ship41x13_noir", "ship41x13_minN5", "syn60x50_full",
         "syn60x50_c1", "syn60x50_maxdist", "syn50x70_maxN"]


def _engine(g, **kw):
    from minrms_amd import engine as eng
    k = cfg_kwargs(g)
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                       max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                       refine_while_searching=k["refine_while_searching"], **kw)
    return eng.Engine(cfg, g["xyza"], g["xyzb"])


@pytest.mark.parametrize("name", SMALL)
def test_orientations(name):
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        og = e.gen_orientations(frag_len=int(g["fm"]))
    assert og.shape[0] == int(g["og_count"])
    assert np.array_equal(bits(og), bits(g["og"]))


@pytest.mark.parametrize("name", SMALL)
def test_search_refine(name):
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        e.search(g["og"])
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", e.cfg.n_min)
        ok = g["search_ssd"] < np.float32(1e26)
        assert np.array_equal(win[ok], g["winner_oid"][ok])
        st = e.refine()
        assert (st.num_iterations, st.num_new_solutions) == (int(g["refine_iters"]), int(g["refine_new"]))
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "refine", e.cfg.n_min)
