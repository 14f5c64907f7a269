#!/usr/bin/env python3
"""Random-shape parity sweep: HIP engine against the CPU oracle (tests only use
a fixed list of shapes).  Every case: random sizes and settings, the first K
orientations of the shuffled list, Search + Refine, all tables compared bit for
bit; each case through the cooperative kernels and through the sweep kernels.
Then half as many cases with alternate solutions (all ranks, all three methods).

  python tools/stress_parity.py [cases] [seed]
"""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc          # noqa: E402
from minrms_amd import engine as eng         # noqa: E402
from minrms_amd.synth import synth_ca        # noqa: E402


def bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def one_case(rng, idx, sizes=None, k_or=None, max_iters=-1):
    n1, n2 = sizes if sizes else (rng.randint(8, 230), rng.randint(8, 230))
    kw = {}
    mm = min(n1, n2)
    if rng.random() < 0.4:
        # now and then close to the shorter chain: the layers below n_min fill a narrow band only in the reference
        kw["n_min"] = rng.randint(1, max(1, mm // 2)) if rng.random() < 0.6 else rng.randint(max(1, mm // 2), mm)
    if rng.random() < 0.3:
        kw["n_max"] = rng.randint(kw.get("n_min", 1), mm)
    if rng.random() < 0.3:
        kw["max_rmsd"] = rng.choice([2.0, 3.5, 6.0])
    if rng.random() < 0.3:
        kw["max_dist"] = rng.choice([4.0, 6.5, 9.0])
    if rng.random() < 0.2:
        kw["refine_while_searching"] = False
    if rng.random() < 0.2:
        kw["of_enable"] = True
        kw["of_max_dist"] = rng.choice([5.0, 7.0, 10.0])
        kw["of_min_matches"] = rng.randint(1, max(1, min(kw.get("n_max", mm), mm) // 3))
    if k_or is None:
        k_or = rng.choice([40, 150, 300])
    a, b = synth_ca(n1, 7000 + idx), synth_ca(n2, 8000 + idx)
    okw = {("nbb" if k == "num_bb" else k): v for k, v in kw.items()}
    ocfg = orc.make_cfg(n1, n2, **okw)
    og = orc.gen_orientations(ocfg, a, b)[:k_or]
    t = orc.search(ocfg, a, b, og)
    found = bool((t.ssd < np.float32(1e26)).any())
    s_ssd, s_xf, s_pairs = t.ssd.copy(), t.xform.copy(), t.pairs.copy()
    if found:
        iters, new = orc.refine(ocfg, a, b, t, max_iters)
    for path in ("coop", "sweep"):
        if path == "sweep":
            os.environ["MINRMS_NO_COOP"] = "1"
        else:
            os.environ.pop("MINRMS_NO_COOP", None)
        with eng.Engine(eng.make_cfg(n1, n2, workspace_bytes=2 << 30, **kw), a, b) as e:
            e.search(og)
            if not found:
                try:
                    e.finalize_search()
                    raise AssertionError("engine found solutions, oracle none")
                except eng.MinrmsError as err:
                    assert err.code == eng.E_NOSOLUTION
                continue
            ssd, xf, pairs, _ = e.results()
            ok = s_ssd < np.float32(1e26)
            assert np.array_equal(bits(ssd), bits(s_ssd)), "search ssd"
            assert np.array_equal(bits(xf)[ok], bits(s_xf)[ok]), "search transforms"
            for k in np.nonzero(ok)[0]:
                n = ocfg.n_min + k
                assert np.array_equal(pairs[k, :n], s_pairs[k, :n]), "search pairs n=%d" % n
            st = e.refine(max_iters)
            assert (st.num_iterations, st.num_new_solutions) == (iters, new), "refine counts"
            ssd, xf, pairs, _ = e.results()
            ok = t.ssd < np.float32(1e26)
            assert np.array_equal(bits(ssd), bits(t.ssd)), "refine ssd"
            assert np.array_equal(bits(xf)[ok], bits(t.xform)[ok]), "refine transforms"
            for k in np.nonzero(ok)[0]:
                n = ocfg.n_min + k
                assert np.array_equal(pairs[k, :n], t.pairs[k, :n]), "refine pairs n=%d" % n
    os.environ.pop("MINRMS_NO_COOP", None)
    return n1, n2, kw, og.shape[0], found


def alt_case(rng, idx):
    """Alternate solutions (-alt-o, -alt-o-orig, -alt-r) with random settings, all ranks."""
    n1, n2 = rng.randint(12, 75), rng.randint(12, 75)
    method = rng.choice(["o", "orig", "r"])
    S = rng.randint(2, 4)
    tol = rng.choice([-1.0, 1.3, 1.8, 2.5])
    refine_all = rng.random() < 0.5
    kw = {}
    if rng.random() < 0.3:
        kw["n_min"] = rng.randint(1, min(n1, n2) // 2)
    if rng.random() < 0.3:
        kw["max_rmsd"] = rng.choice([3.5, 6.0])
    rws = method == "o" or (method == "r" and rng.random() < 0.7)
    kw["refine_while_searching"] = rws
    alt = orc.AltCfg(S, rng.choice([1.5, 3.0, 5.0]), tol, int(refine_all), int(method == "r"),
                     rng.randint(0, min(n1, n2) // 2))
    a, b = synth_ca(n1, 9000 + idx), synth_ca(n2, 9500 + idx)
    ocfg = orc.make_cfg(n1, n2, **kw)
    og = orc.gen_orientations(ocfg, a, b)[:rng.choice([60, 400, 1500])]
    t = orc.search_alt(ocfg, alt, a, b, og)
    if not t.found:
        return n1, n2, method, S, og.shape[0], False
    s_ssd, s_xf, s_pairs = t.ssd.copy(), t.xform.copy(), t.pairs.copy()
    iters, new = orc.refine_alt(ocfg, alt, a, b, t)
    cfg = eng.make_cfg(n1, n2, workspace_bytes=2 << 30, num_solutions=S,
                       alt_min_3d_difference=float(alt.min_3d_difference), alt_rmsd_tolerance=tol,
                       refine_all=refine_all, alt_by_pairs=method == "r",
                       alt_max_pairs_common=int(alt.max_pairs_common), **kw)
    with eng.Engine(cfg, a, b) as e:
        e.search(og)
        e.finalize_search()

        def check(want_ssd, want_xf, want_pairs, stage):
            for r in range(S):
                ssd, xf, pairs = e.results_rank(r)
                assert np.array_equal(bits(ssd), bits(want_ssd[r])), "%s ssd rank %d" % (stage, r)
                ok = want_ssd[r] < np.float32(1e26)
                assert np.array_equal(bits(xf)[ok], bits(want_xf[r])[ok]), "%s transforms rank %d" % (stage, r)
                for k in np.nonzero(ok)[0]:
                    n = ocfg.n_min + k
                    assert np.array_equal(pairs[k, :n], want_pairs[r][k, :n]), "%s pairs rank %d n=%d" % (stage, r, n)

        check(s_ssd, s_xf, s_pairs, "search")
        st = e.refine()
        assert (st.num_iterations, st.num_new_solutions) == (iters, new), "refine counts"
        check(t.ssd, t.xform, t.pairs, "refine")
    return n1, n2, method, S, og.shape[0], True


def big_cases(seed):
    """Sizes at which the cooperative layout changes panel width or grows a
    workgroup (208/209, 256/257, 416/417 columns; 252, 315, 378 rows), few orientations,
    a dozen refinement iterations (the oracle needs seconds per table here)."""
    rng = random.Random(seed)
    t0 = time.time()
    shapes = [(206, 208), (208, 209), (215, 230), (240, 250), (250, 256), (256, 257), (252, 253), (260, 316),
              (300, 300), (379, 380), (330, 416), (417, 417), (440, 300), (505, 130), (130, 505)]
    for idx, sz in enumerate(shapes):
        n1, n2, kw, k_or, found = one_case(rng, 500 + idx, sizes=sz, k_or=12, max_iters=12)
        print("big %2d ok: %dx%d %s  (%.0f s)" % (idx, n1, n2, kw, time.time() - t0), flush=True)
    print("all %d large cases identical" % len(shapes))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        return big_cases(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    t0 = time.time()
    for idx in range(cases):
        state = rng.getstate()
        try:
            n1, n2, kw, k_or, found = one_case(rng, idx)
        except Exception:
            rng.setstate(state)
            print("FAILED case %d of seed %d" % (idx, seed), flush=True)
            raise
        print("case %3d ok: %dx%d %s orientations=%d%s  (%.0f s)" % (
            idx, n1, n2, kw, k_or, "" if found else " [no solution]", time.time() - t0), flush=True)
    print("all %d cases identical" % cases)
    alt_cases = cases // 2
    for idx in range(alt_cases):
        state = rng.getstate()
        try:
            n1, n2, method, S, k_or, found = alt_case(rng, idx)
        except Exception:
            rng.setstate(state)
            print("FAILED alternates case %d of seed %d" % (idx, seed), flush=True)
            raise
        print("alt %3d ok: %dx%d -alt-%s x%d orientations=%d%s  (%.0f s)" % (
            idx, n1, n2, method, S, k_or, "" if found else " [no solution]", time.time() - t0), flush=True)
    print("all %d alternate-solution cases identical" % alt_cases)


if __name__ == "__main__":
    main()
