#!/usr/bin/env python3
"""Random-shape parity sweep: HIP engine against the CPU oracle (tests only use
a fixed list of shapes).  Every case: random sizes and settings, the first K
orientations of the shuffled list, Search + Refine, all tables compared bit for
bit; each case through the cooperative kernels and through the sweep kernels.

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


def one_case(rng, idx):
    n1, n2 = rng.randint(8, 230), rng.randint(8, 230)
    kw = {}
    mm = min(n1, n2)
    if rng.random() < 0.4:
        kw["n_min"] = rng.randint(1, max(1, mm // 2))
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
    k_or = rng.choice([40, 150, 300])
    a, b = synth_ca(n1, 7000 + idx), synth_ca(n2, 8000 + idx)
    okw = {("nbb" if k == "num_bb" else k): v for k, v in kw.items()}
    ocfg = orc.make_cfg(n1, n2, **okw)
    og = orc.gen_orientations(ocfg, a, b)[:k_or]
    t = orc.search(ocfg, a, b, og)
    found = bool((t.ssd < np.float32(1e26)).any())
    s_ssd, s_xf, s_pairs = t.ssd.copy(), t.xform.copy(), t.pairs.copy()
    if found:
        iters, new = orc.refine(ocfg, a, b, t)
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
            st = e.refine()
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


def main():
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


if __name__ == "__main__":
    main()
