"""The C++ host class (minrms_amd/host/search_or_hip.h) through the program
built from it: Search + Refine of goldens of every kind (plain, -of, the three
alternate-solution methods) must give the reference's tables, and the update
counts those of the Python binding."""
import os

import numpy as np
import pytest

from golden_util import assert_tables_equal, cfg_kwargs, load
from host_job_util import HOST_JOB, read_result, run_job, write_job

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["syn60x50_c1", "syn50x70_maxN", "bb3_40x45_of_md", "ship41x13_alt3",
                                  "syn60x50_alto3all", "syn60x50_altr3"])
def test_cpp_host_class_gives_the_reference_tables(name, tmp_path):
    from minrms_amd.build import build_host_job
    assert build_host_job() == HOST_JOB and os.path.exists(HOST_JOB)     # a second of g++ if it is not there yet
    g = load(name)
    assert int(g["refine_max_iters"]) < 0
    h = write_job(tmp_path / "job.bin", g)
    print("host job", name, flush=True)
    p = run_job(tmp_path / "job.bin", tmp_path / "res.bin")
    assert p.returncode == 0, p.stderr
    z = read_result(tmp_path / "res.bin", h)
    n_min = int(g["n_min"])
    for r, (ssd, xf, pairs) in enumerate(z["search"]):
        assert_tables_equal(ssd, xf, pairs, g, "search" if r == 0 else "search_alt%d" % r, n_min)
    if h["refine"]:
        assert (z["iters"], z["new"]) == (int(g["refine_iters"]), int(g["refine_new"]))
        for r, (ssd, xf, pairs) in enumerate(z["refine"]):
            assert_tables_equal(ssd, xf, pairs, g, "refine" if r == 0 else "refine_alt%d" % r, n_min)


def test_cpp_host_class_update_counts_and_a_range(tmp_path):
    from minrms_amd import engine as eng
    from minrms_amd.build import build_host_job
    build_host_job()
    g = load("syn60x50_c1")
    n = g["og"].reshape(-1, 12).shape[0]
    start, stop = n // 5, n - n // 7
    h = write_job(tmp_path / "job.bin", g, count_updates=True, start=start, stop=stop, refine=0)
    p = run_job(tmp_path / "job.bin", tmp_path / "res.bin")
    assert p.returncode == 0, p.stderr
    z = read_result(tmp_path / "res.bin", h)
    k = cfg_kwargs(g)
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"], max_rmsd=k["max_rmsd"],
                       max_dist=k["max_dist"], refine_while_searching=k["refine_while_searching"], device=0,
                       workspace_bytes=1 << 30)
    e = eng.Engine(cfg, g["xyza"], g["xyzb"])
    e.count_updates()
    e.search(g["og"], start, stop)
    want = e.update_counts(stop - start)
    e.finalize_search()
    ssd, xf, pairs, _ = e.results()
    e.close()
    assert np.array_equal(z["updates"], want) and int(want.sum()) > 0
    s, x, pr = z["search"][0]
    assert np.array_equal(s.view(np.uint32), ssd.view(np.uint32))
    ok = ssd < eng.UPPER_BOUND
    assert np.array_equal(x.view(np.uint32)[ok], xf.reshape(-1, 12).view(np.uint32)[ok])
    assert np.array_equal(pr[ok], pairs.reshape(pr.shape)[ok])
