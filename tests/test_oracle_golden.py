"""The CPU restatement (oracle/) against the golden vectors taken from the
real reference (strict build).  Bit-exact on every float and index."""
import numpy as np
import pytest

from oracle import oracle_py as orc
from golden_util import load, cfg_kwargs, bits, assert_tables_equal

SMALL = ["ship41x13_full", "ship41x13_noir", "ship41x13_minN5", "syn60x50_full",
         "syn60x50_c1", "syn60x50_maxdist", "syn50x70_maxN",
         # -of orientation filter; three atoms per residue
         "ship41x13_of", "syn60x50_of", "syn60x50_of_noir", "syn60x50_of_md", "bb3_40x45_full", "bb3_40x45_of_md",
         # fragment matching restricted to interval pairs (-hs, -fm-interval)
         "ss60x50_hs", "ss60x50_hs_minN", "syn60x50_fmint", "syn60x50_fmcut"]


def test_glibc_rand_shuffle_first_values():
    # rand() after srand(1): 1804289383, 846930886, ... => perm of 3 elements
    p = orc.shuffle_perm(3)
    # i=1: j = 1804289383 % 2 = 1 (no swap); i=2: j = 846930886 % 3 = 1 -> swap(2,1)
    assert list(p) == [0, 2, 1]


@pytest.mark.parametrize("name", SMALL)
def test_orientations_match_reference(name):
    g = load(name)
    cfg = orc.make_cfg(**cfg_kwargs(g))
    og = orc.gen_orientations(cfg, g["xyza"], g["xyzb"], frag_len=int(g["fm"]),
                              cutoff_rmsd=float(g.get("fm_cut", -1.0)), intervals=g.get("intervals"))
    assert og.shape[0] == int(g["og_count"])
    assert np.array_equal(bits(og), bits(g["og"]))


@pytest.mark.parametrize("name", SMALL)
def test_search_and_refine_match_reference(name):
    g = load(name)
    cfg = orc.make_cfg(**cfg_kwargs(g))
    t = orc.search(cfg, g["xyza"], g["xyzb"], g["og"], per_on=True)
    assert_tables_equal(t.ssd, t.xform, t.pairs, g, "search", cfg.n_min)
    ok = g["search_ssd"] < np.float32(1e26)
    assert np.array_equal(t.winner[ok], g["winner_oid"][ok])
    assert t.passed == int(g["passed"])
    rejected = np.nonzero(t.pass_flags == 0)[0]
    assert np.array_equal(rejected, g.get("filtered_out", np.zeros(0, np.int64)))
    if "clip_after_search" in g:
        assert bits(np.float32(t.clip)) == bits(g["clip_after_search"])
    if "trace_o" in g:   # every (o_id, n) the reference evaluated
        o, n = g["trace_o"], g["trace_n"].astype(np.int64)
        assert np.array_equal(bits(t.per_on[o, n - cfg.n_min]), bits(g["trace_ssd"]))
        assert np.isnan(t.per_on).sum() == t.per_on.size - o.size
    iters, new = orc.refine(cfg, g["xyza"], g["xyzb"], t)
    assert (iters, new) == (int(g["refine_iters"]), int(g["refine_new"]))
    assert_tables_equal(t.ssd, t.xform, t.pairs, g, "refine", cfg.n_min)


def test_trace_pairs_and_costs():
    """Per-(o_id,n) DP cost and traceback against the reference trace."""
    g = load("ship41x13_full")
    cfg = orc.make_cfg(**cfg_kwargs(g))
    for o in range(0, int(g["og_count"]), 7):
        rot = orc.apply_transform(g["og"][o], g["xyzb"])
        cost, bp = orc.dp(cfg, g["xyza"], rot)
        rows = np.nonzero(g["trace_o"] == o)[0]
        for r in rows:
            n = int(g["trace_n"][r])
            assert bits(cost[n - 1:n])[0] == bits(g["trace_ssd_orig"][r:r + 1])[0]
            assert np.array_equal(orc.traceback(cfg, bp, n), g["trace_pairs"][r, :n])


def test_helix_sheet_interval_pairs_match_reference():
    """Host mirror of the -hs interval pairing against the reference's list."""
    from minrms_amd.synth import synth_pdb_text
    from minrms_amd.intervals import helix_sheet_interval_pairs, extend_interval
    a = synth_pdb_text(60, 1234, with_secondary_structure=True)
    b = synth_pdb_text(50, 5678, with_secondary_structure=True)
    for name in ("ss60x50_hs", "ss60x50_hs_minN"):
        g = load(name)
        iv = helix_sheet_interval_pairs(a, b, fragment_size=int(g["fm"]))
        assert np.array_equal(iv, g["intervals"])
    # widening of short intervals (lib/biopolymer/intervals.cc:248-317)
    assert extend_interval(10, 12, 50, 3) == (10, 12)
    assert extend_interval(10, 11, 50, 4) == (9, 12)       # 2 needed: 1 left, 1 right
    assert extend_interval(10, 11, 50, 5) == (8, 13)       # 3 needed: 2 and 2
    assert extend_interval(0, 1, 50, 6) == (0, 5)          # clipped at the start, shifted right
    assert extend_interval(48, 49, 50, 6) == (44, 49)      # clipped at the end, shifted left
    assert extend_interval(1, 2, 4, 8) == (0, 3)           # sequence shorter than the fragment


ALT = ["ship41x13_alt3", "syn60x50_alt3", "syn60x50_alt4all", "syn60x50_alt2nr",
       "ship41x13_alto3", "syn60x50_alto3all",
       "ship41x13_altr3", "syn60x50_altr3", "syn60x50_altr3tol", "syn60x50_altr2IR"]


def alt_cfg(g):
    return orc.AltCfg(int(g["num_solutions"]), float(g["alt_min_3d"]), float(g["alt_tol"]),
                      int(int(g["refine_method"]) == 2), int(int(g["alt_method"]) == 2),
                      int(g.get("alt_pairs", 0)))


def assert_alt_tables_equal(ssd, xform, pairs, g, stage, n_min):
    """All ranks of the result tables against the golden tables of `stage`."""
    for r in range(ssd.shape[0]):
        tag = stage if r == 0 else "%s_alt%d" % (stage, r)
        assert_tables_equal(ssd[r], xform[r], pairs[r], g, tag, n_min)


def test_principal_axes_match_reference():
    g = load("syn60x50_alt3")
    eig, axes = orc.principal_axes(g["xyzb"])
    assert np.array_equal(bits(eig), bits(g["paxis_eigenvalues"]))
    assert np.array_equal(bits(axes), bits(g["paxis_matrix"]))


@pytest.mark.parametrize("name", ALT)
def test_alternate_solutions_match_reference(name):
    g = load(name)
    cfg = orc.make_cfg(**cfg_kwargs(g))
    alt = alt_cfg(g)
    t = orc.search_alt(cfg, alt, g["xyza"], g["xyzb"], g["og"])
    assert t.found
    assert_alt_tables_equal(t.ssd, t.xform, t.pairs, g, "search", cfg.n_min)
    if int(g["refine_method"]) != 0:
        iters, new = orc.refine_alt(cfg, alt, g["xyza"], g["xyzb"], t)
        assert (iters, new) == (int(g["refine_iters"]), int(g["refine_new"]))
        assert_alt_tables_equal(t.ssd, t.xform, t.pairs, g, "refine", cfg.n_min)
