"""Edge cases of the HIP engine against the CPU oracle (which is pinned to the
reference by tests/test_oracle_golden.py): sizes at the 63-layer block and
16/32-column panel boundaries, tiny and maximum-size structures, ragged shapes,
empty orientation ranges, searches that find nothing."""
import numpy as np
import pytest

from golden_util import bits

pytestmark = pytest.mark.gpu


def _pair(n1, n2, seed=7):
    from minrms_amd.synth import synth_ca
    return synth_ca(n1, 1000 + seed), synth_ca(n2, 2000 + seed)


def _compare(n1, n2, n_orient=None, refine=True, seed=7, max_iters=-1, **cfg_kw):
    from oracle import oracle_py as orc
    from minrms_amd import engine as eng
    a, b = _pair(n1, n2, seed)
    ocfg = orc.make_cfg(n1, n2, **cfg_kw)
    og = orc.gen_orientations(ocfg, a, b, frag_len=4)
    if n_orient is not None:
        og = og[:n_orient]
    print("case %dx%d %s orientations=%d" % (n1, n2, cfg_kw, og.shape[0]), flush=True)
    t = orc.search(ocfg, a, b, og)
    ecfg = eng.make_cfg(n1, n2, **{("num_bb" if k == "nbb" else k): v for k, v in cfg_kw.items()})
    with eng.Engine(ecfg, a, b) as e:
        og_e = e.gen_orientations(frag_len=4)
        if n_orient is None:
            assert np.array_equal(bits(og_e), bits(og))
        e.search(og)
        assert np.array_equal(e.filter_verdicts(), t.pass_flags)
        ok = t.ssd < np.float32(1e26)
        if not ok.any():
            with pytest.raises(eng.MinrmsError) as ei:
                e.finalize_search()
            assert ei.value.code == eng.E_NOSOLUTION
            return
        ssd, xf, pairs, win = e.results()
        _same(ssd, xf, pairs, t, ocfg)
        assert np.array_equal(win[ok], t.winner[ok])
        if refine:
            iters, new = orc.refine(ocfg, a, b, t, max_iters=max_iters)
            st = e.refine(max_iters)
            assert (st.num_iterations, st.num_new_solutions) == (iters, new)
            ssd, xf, pairs, win = e.results()
            _same(ssd, xf, pairs, t, ocfg)


def _same(ssd, xf, pairs, t, cfg):
    assert np.array_equal(bits(ssd), bits(t.ssd))
    ok = t.ssd < np.float32(1e26)
    assert np.array_equal(bits(xf)[ok], bits(t.xform)[ok])
    for k in np.nonzero(ok)[0]:
        n = cfg.n_min + k
        assert np.array_equal(pairs[k, :n], t.pairs[k, :n]), "pairs differ at n=%d" % n


# Small batches go through the cooperative kernel (one workgroup per
# orientation, or several that hand planes to each other when an orientation has
# more than 16 tasks), large ones through the panel kernels of the sweep: every
# shape is run through all three (MINRMS_NO_COOP=1 forces the sweep kernels,
# MINRMS_COOP_GROUP_TASKS=3 several workgroups of at most 3 tasks or one block).
# "compiled": segments of 16-column tasks with the compiled row loop instead of
# the hand-written one; "wide": the hand-written loop with tasks as wide as the
# layout's panels ("coop" runs it with 8-column tasks).
@pytest.fixture(params=["coop", "sweep", "groups", "compiled", "wide"])
def dp_path(request, monkeypatch):
    if request.param == "wide":
        monkeypatch.setenv("MINRMS_COOP_SEG_HALF", "0")
    if request.param == "sweep":
        monkeypatch.setenv("MINRMS_NO_COOP", "1")
    if request.param == "groups":
        monkeypatch.setenv("MINRMS_COOP_GROUP_TASKS", "3")
    if request.param == "compiled":
        monkeypatch.setenv("MINRMS_COOP_ASM", "0")
    return request.param


@pytest.mark.parametrize("n1,n2", [(4, 4), (5, 9), (12, 4), (17, 16), (31, 33)])
def test_tiny_and_ragged(n1, n2, dp_path):
    _compare(n1, n2)


@pytest.mark.parametrize("n1,n2", [(63, 63), (64, 64), (65, 63), (63, 70), (64, 127), (126, 126),
                                   (127, 127), (128, 65), (129, 190), (70, 64), (150, 128), (140, 139)])
def test_layer_block_boundaries(n1, n2, dp_path):
    """n_max = 63, 64, 126, 127, ...: the last layer block is full, has one layer,
    has a single column, ..."""
    _compare(n1, n2, n_orient=40)


@pytest.mark.parametrize("n1,n2", [(80, 97), (100, 161), (70, 193), (200, 96)])
def test_panel_boundaries(n1, n2, dp_path):
    """Column counts just past a multiple of the panel width."""
    _compare(n1, n2, n_orient=24)


def test_segmented_grid_larger_than_the_gpu(monkeypatch):
    """300 tables x 9 workgroups of one CU each, ten times the CUs there are: the
    workgroups of a table's later segments and layer blocks start long after its
    first ones have finished.  Nothing in the kernel waits for a workgroup that
    is dispatched after it (producers come first in the grid, and what crosses
    workgroups goes through arrays in memory that hold every row, so a producer
    never waits for its consumer)."""
    monkeypatch.setenv("MINRMS_COOP_SEG_MAX_BATCH", "400")
    _compare(150, 150, n_orient=300, refine=False)


def test_sweep_with_a_narrow_last_block():
    """More orientations than the cooperative kernel takes, last layer block 7 columns wide."""
    _compare(40, 70, refine=False)


@pytest.mark.parametrize("kw", [dict(n_min=30, n_max=30), dict(n_min=2, n_max=2), dict(n_min=1, n_max=1),
                                dict(n_min=40), dict(n_max=10), dict(n_min=25, n_max=38, max_rmsd=2.0),
                                dict(max_dist=3.0), dict(max_rmsd=0.5),
                                dict(refine_while_searching=False),
                                dict(of_enable=True, of_max_dist=6.0, of_min_matches=40),
                                dict(of_enable=True, of_max_dist=0.5, of_min_matches=30)])
def test_settings_corners(kw):
    """Single n, windows, cutoffs that leave few or no solutions, no -ir,
    a filter that rejects (almost) everything."""
    _compare(45, 40, n_orient=120, **kw)


@pytest.mark.parametrize("dp_path", ["coop", "sweep"])
def test_maximum_size(dp_path, monkeypatch):
    """1008 x 1008 residues (16 layer blocks, the engine's limit): two orientations,
    through the cooperative kernel (the default for so few) and the sweep kernels."""
    if dp_path == "sweep":
        monkeypatch.setenv("MINRMS_NO_COOP", "1")
    _compare(1008, 1008, n_orient=2, refine=False)


@pytest.mark.parametrize("n1,n2,dp_path", [(1500, 1300, "sweep"), (1500, 1000, "coop"), (1100, 1500, "sweep")])
def test_structures_above_a_thousand_residues(n1, n2, dp_path):
    """Up to 33 layer blocks (2 079 residues).  1500 x 1300: 21 blocks, more
    columns than the cooperative kernel's widest layout takes, so every batch
    goes through the sweep kernels (refinement rounds planned by the host);
    1500 x 1000: the cooperative kernel with 1500 rows.  Two orientations and
    two refinement iterations against the oracle."""
    _compare(n1, n2, n_orient=2, refine=True, max_iters=2)


def test_limits_are_enforced():
    from minrms_amd import engine as eng
    a = np.zeros((2080, 3), np.float32)
    with pytest.raises(eng.MinrmsError) as ei:
        eng.Engine(eng.make_cfg(2080, 2080), a, a)
    assert ei.value.code == eng.E_ARG


def test_empty_range_and_call_order():
    from minrms_amd import engine as eng
    a, b = _pair(30, 28)
    with eng.Engine(eng.make_cfg(30, 28), a, b) as e:
        with pytest.raises(eng.MinrmsError) as ei:
            e.refine()
        assert ei.value.code == eng.E_STATE
        og = e.gen_orientations()
        e.search(og, 5, 5)                       # empty o_id range
        with pytest.raises(eng.MinrmsError) as ei:
            e.finalize_search()
        assert ei.value.code == eng.E_NOSOLUTION
        with pytest.raises(eng.MinrmsError) as ei:
            e.search(og, 3, og.shape[0] + 1)
        assert ei.value.code == eng.E_ARG
        # the engine is still usable, and two partial searches equal one full search
        e.reset()
        e.search(og, 0, 100)
        e.search(og, 100, og.shape[0])
        e.finalize_search()
        r1 = e.results()
        e.reset()
        e.search(og)
        r2 = e.results()
        for x, y in zip(r1, r2):
            assert np.array_equal(x, y)
