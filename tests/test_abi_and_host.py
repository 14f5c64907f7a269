"""CPU-only checks: the C-ABI library loads and exports what include/minrms_engine.h
declares, argument validation works without a GPU, and the host-side sharding /
key-reduction logic is right.  No compute call needs a GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "minrms_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(minrms_(?:engine|search|noalloc)_\w+)\s*\(", text)))


def test_header_symbols_are_exported():
    from minrms_amd import engine as eng
    lib = eng.load_library()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libminrms_hip.so does not export %s" % n
    # and the binding knows every one of them
    assert sorted(eng.SYMBOLS) == names


def test_cells_per_orientation_matches_baseline_table():
    from minrms_amd import engine as eng
    # DP cells per orientation with n_min = 1: 150x150 -> 1 136 275, 250x250 -> 5 239 625, 400x400 -> 21 413 400
    for n, cells in ((150, 1136275), (250, 5239625), (400, 21413400)):
        assert eng.cells_per_orientation(eng.make_cfg(n, n)) == cells
    # windowed (n_min = 50): layer n is (min(N, n+w-1)-n+1)^2 with w = N-n_min+1
    c = eng.make_cfg(150, 150, n_min=50)
    w = 101
    assert eng.cells_per_orientation(c) == sum((min(150, n + w - 1) - n + 1) ** 2 for n in range(1, 151))


def test_argument_validation_needs_no_gpu():
    from minrms_amd import engine as eng
    a = np.zeros((10, 3), np.float32)
    with pytest.raises(eng.MinrmsError) as ei:
        eng.Engine(eng.make_cfg(10, 10, n_min=5, n_max=3), a, a)
    assert ei.value.code == eng.E_ARG and "does not make sense" in str(ei.value)
    with pytest.raises(eng.MinrmsError) as ei:
        eng.Engine(eng.make_cfg(10, 10, n_min=1, n_max=11), a, a)
    assert ei.value.code == eng.E_ARG
    with pytest.raises(ValueError):
        eng.Engine(eng.make_cfg(12, 10), a, a)
    # -alt-o: minrms_parser.cc:812-836
    for bad in (dict(alt_min_3d_difference=-1.0), dict(alt_rmsd_tolerance=0.9), dict(alt_rmsd_tolerance=1.0)):
        kw = dict(num_solutions=3, alt_min_3d_difference=2.0, alt_rmsd_tolerance=1.5)
        kw.update(bad)
        with pytest.raises(eng.MinrmsError) as ei:
            eng.Engine(eng.make_cfg(10, 10, **kw), a, a)
        assert ei.value.code == eng.E_ARG
    # -alt-r: minrms_parser.cc:911-925
    for bad in (dict(alt_max_pairs_common=-1), dict(alt_rmsd_tolerance=1.0)):
        kw = dict(num_solutions=2, alt_by_pairs=True, alt_max_pairs_common=3)
        kw.update(bad)
        with pytest.raises(eng.MinrmsError) as ei:
            eng.Engine(eng.make_cfg(10, 10, **kw), a, a)
        assert ei.value.code == eng.E_ARG
    # -of: minrms_parser.cc:1089-1123
    for bad in (dict(of_min_matches=0), dict(of_min_matches=11), dict(of_min_matches=5, n_max=4),
                dict(of_min_matches=3, of_max_dist=0.0)):
        kw = dict(of_enable=True, of_max_dist=6.0, of_min_matches=3)
        kw.update(bad)
        with pytest.raises(eng.MinrmsError) as ei:
            eng.Engine(eng.make_cfg(10, 10, **kw), a, a)
        assert ei.value.code == eng.E_ARG


def test_fails_loudly_without_gpu():
    """No silent CPU fallback: without a usable GPU construction raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from minrms_amd import engine as eng
    a = np.zeros((10, 3), np.float32)
    with pytest.raises(eng.MinrmsError) as ei:
        eng.Engine(eng.make_cfg(10, 10), a, a)
    assert ei.value.code == eng.E_DEVICE


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "minrms_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f


def test_shard_range_partitions():
    from minrms_amd.search_or import shard_range
    for n in (0, 1, 7, 21609):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def test_key_order_survives_int64_round_trip():
    from minrms_amd.search_or import keys_to_i64, i64_to_keys
    ssd = np.array([0.0, 1e-30, 1.5, 1.5, 3.0e25, 9.9e25], np.float32)
    oid = np.array([7, 0, 5, 4, 2**32 - 1, 1], np.uint64)
    keys = (ssd.view(np.uint32).astype(np.uint64) << np.uint64(32)) | oid
    keys = np.append(keys, np.uint64(0xFFFFFFFFFFFFFFFF))
    i = keys_to_i64(keys)
    assert np.array_equal(np.argsort(i, kind="stable"), np.argsort(keys, kind="stable"))
    assert i[-1] == np.iinfo(np.int64).max
    assert np.array_equal(i64_to_keys(i), keys)


def test_cpp_host_program_builds_and_reports_errors_like_the_reference(tmp_path):
    """minrms_amd/host/search_or_hip.h compiles against the header alone and ends
    the program with a message and exit(-1) on an engine error, as ERR() does."""
    import subprocess
    from minrms_amd.build import build_host_job
    from golden_util import load
    from host_job_util import run_job, write_job
    exe = build_host_job()
    assert os.path.exists(exe)
    write_job(tmp_path / "job.bin", load("syn60x50_c1"), n_min=0)     # n_min must be at least 1
    p = run_job(tmp_path / "job.bin", tmp_path / "res.bin")
    assert p.returncode == 255 and "min_n = 0" in p.stderr, (p.returncode, p.stderr)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 2 and "usage" in p.stderr
