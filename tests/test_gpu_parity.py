"""HIP engine (through the C ABI) against the golden vectors of the reference
and against the CPU oracle.  Bit-exact: every float is compared by bits."""
import numpy as np
import pytest

from golden_util import load, cfg_kwargs, bits, assert_tables_equal

pytestmark = pytest.mark.gpu

SMALL = ["ship41x13_full", "ship41x13_noir", "ship41x13_minN5", "syn60x50_full",
         "syn60x50_c1", "syn60x50_maxdist", "syn50x70_maxN",
         # -of orientation filter; three atoms per residue
         "ship41x13_of", "syn60x50_of", "syn60x50_of_noir", "syn60x50_of_md", "bb3_40x45_full", "bb3_40x45_of_md",
         # fragment matching restricted to interval pairs (-hs, -fm-interval)
         "ss60x50_hs", "ss60x50_hs_minN", "syn60x50_fmint", "syn60x50_fmcut"]


def _check_filter(e, g):
    """Verdict of the -of filter for every orientation against the reference's."""
    v = e.filter_verdicts()
    assert v.size == int(g.get("searched", g["og_count"]))
    assert np.array_equal(np.nonzero(v == 0)[0], g.get("filtered_out", np.zeros(0, np.int64)))
    assert int(v.sum()) == int(g["passed"])


def _engine(g, **kw):
    from minrms_amd import engine as eng
    k = cfg_kwargs(g)
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                       max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                       refine_while_searching=k["refine_while_searching"],
                       of_enable=k["of_enable"], of_max_dist=k["of_max_dist"],
                       of_min_matches=k["of_min_matches"], **kw)
    return eng.Engine(cfg, g["xyza"], g["xyzb"])


@pytest.mark.parametrize("name", SMALL)
def test_orientations(name):
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        og = e.gen_orientations(frag_len=int(g["fm"]), cutoff_rmsd=float(g.get("fm_cut", -1.0)),
                                  intervals=g.get("intervals"))
    assert og.shape[0] == int(g["og_count"])
    assert np.array_equal(bits(og), bits(g["og"]))


@pytest.mark.parametrize("name", SMALL)
def test_search_refine(name):
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        e.search(g["og"])
        _check_filter(e, g)
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", e.cfg.n_min)
        ok = g["search_ssd"] < np.float32(1e26)
        assert np.array_equal(win[ok], g["winner_oid"][ok])
        st = e.refine()
        assert (st.num_iterations, st.num_new_solutions) == (int(g["refine_iters"]), int(g["refine_new"]))
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "refine", e.cfg.n_min)


BIG = ["syn90x100_full", "syn90x100_noir", "syn90x100_of", "syn150x150_c1", "syn150x150_full"]


def _run_big(name, monkeypatch=None, **engine_kw):
    import hashlib
    g = load(name)
    print("case", name, flush=True)
    with _engine(g, **engine_kw) as e:
        og = e.gen_orientations(frag_len=int(g["fm"]), cutoff_rmsd=float(g.get("fm_cut", -1.0)),
                                  intervals=g.get("intervals"))
        assert og.shape[0] == int(g["og_count"])
        assert hashlib.sha256(np.ascontiguousarray(og).tobytes()).hexdigest() == str(g["og_sha256"])
        e.search(og, 0, int(g["searched"]) if "searched" in g else None)
        _check_filter(e, g)
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", e.cfg.n_min)
        ok = g["search_ssd"] < np.float32(1e26)
        assert np.array_equal(win[ok], g["winner_oid"][ok])
        st = e.refine()
        assert (st.num_iterations, st.num_new_solutions) == (int(g["refine_iters"]), int(g["refine_new"]))
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "refine", e.cfg.n_min)
        # The reference's own recipe (-O3 -ffast-math) is not bit-reproducible
        # against its strict build: the two reference builds differ from each
        # other by up to ~1.5e-4 A (3e-3 A where the RMSD itself is ~0) and pick
        # different alignments for a few n.  The engine is pinned to the strict
        # build (exact, above); against the fast build only that noise floor
        # can be asserted.
        n = np.arange(e.cfg.n_min, e.cfg.n_max + 1)
        okr = g["fast_refine_ssd"] < np.float32(1e26)
        assert np.array_equal(okr, ssd < np.float32(1e26))
        rm = np.sqrt(ssd[okr].astype(np.float64) / n[okr])
        rf = np.sqrt(g["fast_refine_ssd"][okr].astype(np.float64) / n[okr])
        # (an exact fit of 2 or 3 pairs has RMSD ~ 0: there sqrt() turns rounding
        # noise of the sum of squares into up to ~1e-2 A)
        tol = np.where(rf > 0.1, 5e-3, 2e-2)
        assert np.all(np.abs(rm - rf) < tol)


@pytest.mark.parametrize("name", BIG)
def test_full_size(name):
    """Several layer blocks (n_max > 63), full benchmark size included."""
    _run_big(name)


@pytest.mark.parametrize("name", ["syn90x100_full", "syn90x100_noir", "syn90x100_of", "syn150x150_c1"])
def test_sweep_in_several_batches(name):
    """A workspace that holds a fraction of the orientations: the sweep runs in
    several (equal) batches and finds the same winners."""
    _run_big(name, workspace_bytes=600 << 20)


@pytest.mark.parametrize("env", [{"MINRMS_SWEEP_STREAMS": "1"},
                                 {"MINRMS_SWEEP_STREAMS": "2", "MINRMS_SWEEP_PIECES": "3"},
                                 {"MINRMS_NO_TAIL8": "1"},
                                 {"MINRMS_NO_TAIL8": "1", "MINRMS_SWEEP_STREAMS": "1"}],
                         ids=["one_stream", "two_streams_three_pieces", "whole_last_panel", "whole_last_panel_one_stream"])
@pytest.mark.parametrize("name", ["syn150x150_full", "syn150x150_c1"])
def test_sweep_forms(name, env, monkeypatch):
    """The sweep of a large batch on one stream and as two halves on two, and
    with the last eight-cell group of a block's last panel computed or skipped:
    all against the reference's tables (the default -- two streams, skipped --
    is test_full_size)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _run_big(name)


FULL_SWEEPS = ["syn250x250_full", "syn400x400_fullsearch"]


@pytest.mark.parametrize("name", FULL_SWEEPS)
def test_full_sweeps_of_the_large_configs(name):
    """EVERY orientation of the 250x250 pair (61 009; Search + Refine, one
    reference process) and of the 400x400 pair (157 609; Search by several
    reference processes over o_id ranges, merged by the rule of FirstPass;
    Refine -- 912 iterations -- by one reference process that was handed the
    merged tables), through the default multi-batch sweep."""
    import hashlib
    import os
    from golden_util import GOLD
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        pytest.skip("golden %s not generated (hours of reference CPU time)" % name)
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        og = e.gen_orientations(frag_len=int(g["fm"]))
        assert og.shape[0] == int(g["og_count"])
        assert hashlib.sha256(np.ascontiguousarray(og).tobytes()).hexdigest() == str(g["og_sha256"])
        e.search(og)
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", e.cfg.n_min)
        # the reference does not say which orientation won, but the merge knows which range did
        ok = g["search_ssd"] < np.float32(1e26)
        lo, hi = g["range_lo"][g["search_range"]], g["range_hi"][g["search_range"]]
        assert np.all((win[ok] >= lo[ok]) & (win[ok] < hi[ok]))
        if "refine_ssd" in g:
            st = e.refine()
            assert (st.num_iterations, st.num_new_solutions) == (int(g["refine_iters"]), int(g["refine_new"]))
            ssd, xf, pairs, win = e.results()
            assert_tables_equal(ssd, xf, pairs, g, "refine", e.cfg.n_min)


LARGE = ["syn250x250_first400", "syn300x220_first200", "syn400x400_first150"]


@pytest.mark.parametrize("dp_path", ["coop", "sweep"])
@pytest.mark.parametrize("name", LARGE)
def test_large_structures(name, dp_path, monkeypatch):
    """The larger benchmark sizes (several column panels, up to 7 layer blocks):
    the first few hundred orientations of the shuffled list, then Refine.  A
    few hundred orientations go through the cooperative kernel by default;
    "sweep" sends them through the sweep kernels instead (panels of 128-160
    columns, three or more per row, path strings in memory), so that those are
    pinned to the reference at these sizes too."""
    if dp_path == "sweep":
        monkeypatch.setenv("MINRMS_NO_COOP", "1")
    _run_big(name)


def test_cooperative_kernel_gives_up_cleanly(monkeypatch):
    """A bounded wait of the cooperative DP kernel that runs out must surface as
    MINRMS_E_DEVICE from every call that would publish results, and stick until
    reset().  The host forces the condition (the workgroups start in the state a
    wait that ran out leaves behind); every wave still runs to its normal end."""
    from minrms_amd import engine as eng
    g = load("syn60x50_full")
    monkeypatch.setenv("MINRMS_COOP_SPIN_LIMIT", "-1")
    with _engine(g) as bad:
        with pytest.raises(eng.MinrmsError) as ei:
            bad.search(g["og"][:40])          # a small batch runs the cooperative kernel
            bad.results()
        assert ei.value.code == eng.E_DEVICE
        for call in (bad.get_keys, bad.keys_device_ptr, bad.results):
            with pytest.raises(eng.MinrmsError) as ei:
                call()
            assert ei.value.code == eng.E_DEVICE
    # the sweep kernels are not affected; winners and refinement (cooperative) are
    with _engine(g) as bad:
        bad.search(g["og"])                     # 2679 orientations: sweep kernels
        for call in (bad.results, bad.refine, bad.finalize_search, bad.get_keys, bad.results):
            with pytest.raises(eng.MinrmsError) as ei:
                call()                          # the first re-solves the winners with the cooperative kernel
            assert ei.value.code == eng.E_DEVICE
        monkeypatch.delenv("MINRMS_COOP_SPIN_LIMIT")
        # reset() clears the failure; the engine keeps the give-up setting it was created with
        bad.reset()
        bad.search(g["og"])
        assert bad.get_keys().size == bad.range_n
    with _engine(g) as good:                    # same process, same pooled buffers: clean again
        good.search(g["og"])
        ssd, xf, pairs, win = good.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", good.cfg.n_min)


def test_launch_tags_of_the_cooperative_kernel_wrap(monkeypatch):
    """Row counters and granules that cross workgroups carry the launch number;
    when it runs out of bits both arrays are cleared and the count starts again.
    The engine is created a few launches before that point, so the wrap happens
    in the middle of the refinement rounds (segmented launches) of the full
    150x150 job."""
    monkeypatch.setenv("MINRMS_COOP_LAUNCHES_START", str((1 << 20) - 40))
    _run_big("syn150x150_full")
    monkeypatch.setenv("MINRMS_COOP_SEG_TASKS", "0")     # whole blocks per workgroup: the row counters
    monkeypatch.setenv("MINRMS_COOP_GROUP_TASKS", "5")
    _run_big("syn90x100_full")


@pytest.mark.parametrize("name,where", [("syn60x50_full", "memory"), ("syn90x100_full", "memory"),
                                        ("bb3_40x45_full", "memory"), ("syn250x250_first400", "lds")])
def test_path_strings_in_lds_and_in_memory(name, where, monkeypatch):
    """The sweep's thread-per-chain traceback keeps its path strings in LDS or,
    when they are long, in memory: both give the same bits."""
    monkeypatch.setenv("MINRMS_TRACE_PATH", where)
    if name in SMALL:
        test_search_refine(name)
    else:
        _run_big(name)


@pytest.mark.parametrize("name,limit", [("syn60x50_c1", "1000000000"), ("ship41x13_full", "0"),
                                        ("bb3_40x45_full", "1000000000")])
def test_both_traceback_kernels(name, limit, monkeypatch):
    """Wave-per-chain and thread-per-chain traceback/refit give the same bits."""
    monkeypatch.setenv("MINRMS_WAVE_TRACE_MAX", limit)
    test_search_refine(name)


@pytest.mark.parametrize("name", ["syn60x50_full", "bb3_40x45_of_md", "syn90x100_of"])
def test_refine_rounds_planned_by_host(name, monkeypatch):
    """Refinement with the host choosing each round's entries (the fallback for
    structures too large for the cooperative DP kernel)."""
    monkeypatch.setenv("MINRMS_REFINE_HOST_ROUNDS", "1")
    if name in BIG:
        _run_big(name)
    else:
        test_search_refine(name)


@pytest.mark.parametrize("host_rounds", [False, True])
@pytest.mark.parametrize("max_iters", [1, 2, 7, 40])
def test_refine_iteration_limit(max_iters, host_rounds, monkeypatch):
    """refine_max_iters (search_or_dyn3d.cc:2372) against the CPU oracle."""
    from oracle import oracle_py as orc
    if host_rounds:
        monkeypatch.setenv("MINRMS_REFINE_HOST_ROUNDS", "1")
    g = load("syn60x50_full")
    cfg = orc.make_cfg(**cfg_kwargs(g))
    t = orc.search(cfg, g["xyza"], g["xyzb"], g["og"])
    iters, new = orc.refine(cfg, g["xyza"], g["xyzb"], t, max_iters=max_iters)
    with _engine(g) as e:
        e.search(g["og"])
        st = e.refine(max_iters)
        assert (st.num_iterations, st.num_new_solutions) == (iters, new)
        ssd, xf, pairs, win = e.results()
        assert np.array_equal(bits(ssd), bits(t.ssd))
        ok = t.ssd < np.float32(1e26)
        assert np.array_equal(bits(xf)[ok], bits(t.xform)[ok])
        for k in np.nonzero(ok)[0]:
            assert np.array_equal(pairs[k, :cfg.n_min + k], t.pairs[k, :cfg.n_min + k])


@pytest.mark.parametrize("name", ["syn90x100_full", "ship41x13_full"])
def test_narrow_panels(name, monkeypatch):
    """Force 32-column panels so rows are handed over between panels."""
    monkeypatch.setenv("MINRMS_MAX_PB", "32")
    if name in BIG:
        _run_big(name)
    else:
        test_search_refine(name)


@pytest.mark.parametrize("name", ["syn150x150_full", "syn90x100_of", "syn60x50_c1", "bb3_40x45_full"])
def test_segment_tasks_with_the_compiled_row_loop(name, monkeypatch):
    """The tasks of 16-column segments run a hand-written row loop by default
    (every other cooperative test goes through it); the compiled loop it
    replaces must give the same bits."""
    monkeypatch.setenv("MINRMS_COOP_ASM", "0")
    if name in BIG:
        _run_big(name)
    else:
        test_search_refine(name)


@pytest.mark.parametrize("name,cap", [("syn150x150_full", "5"), ("syn150x150_full", "1"), ("syn90x100_of", "2")])
def test_cooperative_dp_over_several_workgroups(name, cap, monkeypatch):
    """Winners and every refinement round with the tasks of an orientation
    spread over workgroups that hand planes to each other (the layout large
    structures get by default)."""
    monkeypatch.setenv("MINRMS_COOP_GROUP_TASKS", cap)
    _run_big(name)


ALT = ["ship41x13_alt3", "syn60x50_alt3", "syn60x50_alt4all", "syn60x50_alt2nr",
       "ship41x13_alto3", "syn60x50_alto3all",
       "ship41x13_altr3", "syn60x50_altr3", "syn60x50_altr3tol", "syn60x50_altr2IR"]


@pytest.mark.parametrize("name", ALT)
def test_alternate_solutions(name):
    """-alt-o: 2nd, 3rd, ... best alignments per n, after the search and after
    refinement (-r refines the best only, -r-all every rank), against the
    reference's tables of every rank."""
    g = load(name)
    print("case", name, flush=True)
    S = int(g["num_solutions"])
    with _engine(g, num_solutions=S, alt_min_3d_difference=float(g["alt_min_3d"]),
                 alt_rmsd_tolerance=float(g["alt_tol"]), refine_all=int(g["refine_method"]) == 2,
                 alt_by_pairs=int(g["alt_method"]) == 2, alt_max_pairs_common=int(g.get("alt_pairs", 0))) as e:
        e.search(g["og"])
        for r in range(S):
            ssd, xf, pairs = e.results_rank(r)
            assert_tables_equal(ssd, xf, pairs, g, "search" if r == 0 else "search_alt%d" % r, e.cfg.n_min)
        if int(g["refine_method"]) != 0:
            st = e.refine()
            assert (st.num_iterations, st.num_new_solutions) == (int(g["refine_iters"]), int(g["refine_new"]))
            for r in range(S):
                ssd, xf, pairs = e.results_rank(r)
                assert_tables_equal(ssd, xf, pairs, g, "refine" if r == 0 else "refine_alt%d" % r, e.cfg.n_min)


@pytest.mark.parametrize("name", ["syn60x50_alt3", "ship41x13_alto3", "syn60x50_altr3tol"])
def test_alternate_solutions_with_the_candidates_filtered_on_the_host(name, monkeypatch):
    """The records that may become alternates are filtered on the device and only
    the survivors copied out (every other test); copying all records and
    filtering on the host must give the same tables."""
    monkeypatch.setenv("MINRMS_ALT_HOST_FILTER", "1")
    test_alternate_solutions(name)


@pytest.mark.parametrize("name", ["syn150x150_full", "syn90x100_of", "bb3_40x45_full"])
@pytest.mark.parametrize("compiled", [False, True])
def test_segment_tasks_of_32_columns(name, compiled, monkeypatch):
    """Structures above 256 residues get 32-column panels in the cooperative
    kernel; forced here on the small goldens, with the hand-written row loop
    and with the compiled one."""
    monkeypatch.setenv("MINRMS_COOP_PB", "32")
    monkeypatch.setenv("MINRMS_COOP_SEG_HALF", "0")      # tasks as wide as the panels
    if compiled:
        monkeypatch.setenv("MINRMS_COOP_ASM", "0")
    if name in BIG:
        _run_big(name)
    else:
        test_search_refine(name)


# (400 columns do not fit the 16 panels of a 16-column layout)
WIDTH_CASES = [(name, layout, task)
               for name in ["syn150x150_full", "syn90x100_of", "syn60x50_c1", "bb3_40x45_full",
                            "ss60x50_hs_minN", "syn250x250_first400", "syn400x400_first150"]
               for layout, task in [(16, 8), (16, 16), (32, 8), (32, 16), (32, 32), (64, 8), (64, 16), (64, 32), (64, 64)]
               if not (name.startswith("syn400") and layout == 16)]


@pytest.mark.parametrize("name,layout,task", WIDTH_CASES)
def test_segment_task_widths(name, layout, task, monkeypatch):
    """The tasks of a refinement round are narrower than the panels of the
    cooperative layout they work on (8 columns by default, 16 or 32 for the
    largest structures): every combination of layout and task width gives the
    reference's bits.  An 8-column task writes half a back-pointer word per row."""
    monkeypatch.setenv("MINRMS_COOP_PB", str(layout))
    if task == layout:
        monkeypatch.setenv("MINRMS_COOP_SEG_HALF", "0")
    else:
        monkeypatch.setenv("MINRMS_COOP_SEG_PB", str(task))
        monkeypatch.setenv("MINRMS_COOP_SEG_HALF_MAX_TASKS", "64")
    if name in BIG or name in LARGE:
        _run_big(name)
    else:
        test_search_refine(name)


@pytest.mark.parametrize("task", [8, 16])
@pytest.mark.parametrize("per_group", [1, 2, 3, 6, 8, 11])
@pytest.mark.parametrize("name", ["syn150x150_full", "syn250x250_first400"])
def test_segment_sizes(name, per_group, task, monkeypatch):
    """Segments of 1 ... 11 tasks per workgroup (4 by default): the extra waves
    of a segment fetch and publish planes of 9 ... 177 columns (one, two or
    three loads per row and lane)."""
    if task * per_group > 176:
        pytest.skip("a segment's plane fetch takes at most 192 columns")
    monkeypatch.setenv("MINRMS_COOP_SEG_TASKS", str(per_group))
    monkeypatch.setenv("MINRMS_COOP_SEG_PB", str(task))
    monkeypatch.setenv("MINRMS_COOP_SEG_HALF_MAX_TASKS", "64")
    _run_big(name)


@pytest.mark.parametrize("name", ["syn90x100_full", "syn90x100_of", "syn150x150_c1", "syn90x100_noir"])
def test_update_counts_do_not_depend_on_how_the_sweep_is_cut(name):
    """How many entries of the per-N table each orientation improved when it was
    evaluated (what the reference prints as "<count> updates for orientation
    <o>"; the numbers themselves are checked against the stock binary's output
    in test_gpu_dropin_binary.py): the same whether the list is searched in one
    call, in two calls one after the other, or in several batches; and asking
    for them changes no result."""
    g = load(name)
    print("case", name, flush=True)
    with _engine(g) as e:
        og = e.gen_orientations(frag_len=int(g["fm"]), cutoff_rmsd=float(g.get("fm_cut", -1.0)),
                                  intervals=g.get("intervals"))
        n_og = int(g["searched"]) if "searched" in g else og.shape[0]
        e.count_updates()
        e.search(og, 0, n_og)
        whole = e.update_counts(n_og)
        ssd, xf, pairs, win = e.results()
        assert_tables_equal(ssd, xf, pairs, g, "search", e.cfg.n_min)
        assert whole.min() >= 0 and whole.sum() >= int((ssd < np.float32(1e26)).sum())
        # every winner improved its entry when it came
        assert np.all(whole[win[win >= 0]] > 0)
        e.reset()
        half = n_og // 3
        e.search(og, 0, half)
        first = e.update_counts(half)
        e.search(og, half, n_og)
        second = e.update_counts(n_og - half)
        assert np.array_equal(np.concatenate([first, second]), whole)
    with _engine(g, workspace_bytes=600 << 20) as e:
        e.count_updates()
        e.search(og, 0, n_og)
        assert np.array_equal(e.update_counts(n_og), whole)
