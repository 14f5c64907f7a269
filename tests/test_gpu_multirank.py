"""Two ranks sharing the one GPU of the test box: each sweeps its own o_id range
with the HIP engine, keys are MIN-all-reduced over gloo (host tensors), and both
ranks must end with the reference's tables."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from golden_util import load, cfg_kwargs, assert_tables_equal

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir, on_device=False, workspace=1 << 30):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.init()                       # the engine then shares the HIP runtime torch has loaded
    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip

    def all_reduce_min(keys_i64):           # gloo gets CPU tensors only
        t = torch.from_numpy(np.array(keys_i64, copy=True))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.numpy()

    def device_all_reduce_min(t):
        # Stands in for the RCCL all-reduce (two ranks cannot share one GPU under
        # RCCL): the tensor IS the engine's key table in device memory; it makes a
        # synchronous trip through host memory for gloo and is written back in place.
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN)
        t.copy_(h)

    g = load(name)
    k = cfg_kwargs(g)
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                       max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                       refine_while_searching=k["refine_while_searching"],
                       of_enable=k["of_enable"], of_max_dist=k["of_max_dist"],
                       of_min_matches=k["of_min_matches"], device=0,
                       workspace_bytes=workspace)
    if on_device:
        so = SearchOrHip(cfg, g["xyza"], g["xyzb"], rank=rank, world_size=world,
                         device_all_reduce_min=device_all_reduce_min)
    else:
        so = SearchOrHip(cfg, g["xyza"], g["xyzb"], rank=rank, world_size=world,
                         all_reduce_min=all_reduce_min)
    print("rank", rank, "case", name, "keys reduced", "on the device table" if on_device else "through the host", flush=True)
    og = g["og"] if "og" in g else so.generate_orientations(frag_len=int(g["fm"]))
    so.Search(og)
    ssd, xf, pairs, win = so.tables()
    np.savez(os.path.join(outdir, "search_%d.npz" % rank), ssd=ssd, xf=xf, pairs=pairs, win=win)
    st = so.Refine()
    ssd, xf, pairs, win = so.tables()
    np.savez(os.path.join(outdir, "refine_%d.npz" % rank), ssd=ssd, xf=xf, pairs=pairs,
             iters=st.num_iterations)
    so.close()
    dist.barrier()
    dist.destroy_process_group()


def _check(name, tmp_path):
    g = load(name)
    n_min = int(g["n_min"])
    for rank in range(2):
        s = np.load(tmp_path / ("search_%d.npz" % rank))
        assert_tables_equal(s["ssd"], s["xf"], s["pairs"], g, "search", n_min)
        ok = g["search_ssd"] < np.float32(1e26)
        if "winner_oid" in g:
            assert np.array_equal(s["win"][ok], g["winner_oid"][ok])
        else:
            # a sweep the reference ran as several processes: the merge knows which o_id range won
            lo, hi = g["range_lo"][g["search_range"]], g["range_hi"][g["search_range"]]
            assert np.all((s["win"][ok] >= lo[ok]) & (s["win"][ok] < hi[ok]))
        r = np.load(tmp_path / ("refine_%d.npz" % rank))
        assert int(r["iters"]) == int(g["refine_iters"])
        assert_tables_equal(r["ssd"], r["xf"], r["pairs"], g, "refine", n_min)


@pytest.mark.parametrize("name", ["syn60x50_c1", "syn50x70_maxN", "bb3_40x45_of_md"])
def test_two_ranks_one_gpu(name, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    _check(name, tmp_path)


@pytest.mark.parametrize("name", ["syn60x50_c1", "bb3_40x45_of_md"])
def test_keys_reduced_in_place_on_the_device_table(name, tmp_path):
    """The production path of SearchOrHip.Search for world_size > 1: a torch
    tensor over minrms_engine_keys_device_ptr, "no solution" re-coded for a
    signed MIN, reduced in place (the all-reduce itself is stood in for)."""
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path), True), nprocs=2, join=True)
    _check(name, tmp_path)


@pytest.mark.parametrize("name,workspace", [("syn150x150_full", 8 << 30), ("syn90x100_of", 8 << 30),
                                            ("syn250x250_full", 40 << 30), ("syn400x400_fullsearch", 40 << 30)])
def test_two_ranks_at_full_size(name, workspace, tmp_path):
    """The benchmark pair (21 609 orientations, 328 refinement iterations), a
    filtered search, and the complete jobs of the two large configurations
    (250x250: 61 009 orientations, 641 refinement iterations; 400x400, the one
    that is cut over GPUs: 157 609 orientations, 912 iterations), split over two
    ranks."""
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path), True, workspace), nprocs=2, join=True)
    _check(name, tmp_path)


def _alt_worker(rank, world, port, name, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.init()
    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip

    def all_reduce_min(keys_i64):
        t = torch.from_numpy(np.array(keys_i64, copy=True))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.numpy()

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)          # gloo: host objects
        return out

    g = load(name)
    k = cfg_kwargs(g)
    S = int(g["num_solutions"])
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                       max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                       refine_while_searching=k["refine_while_searching"], device=0, workspace_bytes=1 << 30,
                       of_enable=k["of_enable"], of_max_dist=k["of_max_dist"], of_min_matches=k["of_min_matches"],
                       num_solutions=S, alt_min_3d_difference=float(g["alt_min_3d"]),
                       alt_rmsd_tolerance=float(g["alt_tol"]), refine_all=int(g["refine_method"]) == 2,
                       alt_by_pairs=int(g["alt_method"]) == 2, alt_max_pairs_common=int(g.get("alt_pairs", 0)))
    so = SearchOrHip(cfg, g["xyza"], g["xyzb"], rank=rank, world_size=world, all_reduce_min=all_reduce_min,
                     all_gather=all_gather)
    print("rank", rank, "case", name, flush=True)
    so.Search(g["og"])
    out = {}
    for r in range(S):
        ssd, xf, pairs = so.engine.results_rank(r)
        out["s_ssd%d" % r], out["s_xf%d" % r], out["s_pairs%d" % r] = ssd, xf, pairs
    if int(g["refine_method"]) != 0:
        st = so.Refine()
        out["iters"], out["new"] = st.num_iterations, st.num_new_solutions
        for r in range(S):
            ssd, xf, pairs = so.engine.results_rank(r)
            out["r_ssd%d" % r], out["r_xf%d" % r], out["r_pairs%d" % r] = ssd, xf, pairs
    np.savez(os.path.join(outdir, "alt_%d.npz" % rank), **out)
    so.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ship41x13_alt3", "syn60x50_alt3", "syn60x50_alt4all", "syn60x50_alto3all",
                                  "ship41x13_altr3", "syn60x50_altr3", "syn60x50_altr3tol", "syn60x50_altr2IR"])
def test_alternates_with_a_split_search(name, tmp_path):
    """-alt-o / -alt-o-orig with the orientation list split over two ranks: the
    ranks exchange their candidate records, rebuild the one-process lists and
    must end with the reference's tables of every rank (2nd, 3rd best ...).
    -alt-r: every further pass sweeps each rank's part of the orientations it
    revisits and ends in one MIN all-reduce of the pass's keys."""
    mp.spawn(_alt_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    g = load(name)
    S, n_min = int(g["num_solutions"]), int(g["n_min"])
    for rank in range(2):
        z = np.load(tmp_path / ("alt_%d.npz" % rank))
        for r in range(S):
            assert_tables_equal(z["s_ssd%d" % r], z["s_xf%d" % r], z["s_pairs%d" % r], g,
                                "search" if r == 0 else "search_alt%d" % r, n_min)
        if int(g["refine_method"]) != 0:
            assert (int(z["iters"]), int(z["new"])) == (int(g["refine_iters"]), int(g["refine_new"]))
            for r in range(S):
                assert_tables_equal(z["r_ssd%d" % r], z["r_xf%d" % r], z["r_pairs%d" % r], g,
                                    "refine" if r == 0 else "refine_alt%d" % r, n_min)


def _altr_of_worker(rank, world, port, outdir):
    """-alt-r with the -of filter and a distance cutoff whose sqrt/square round trip is not the
    identity: the cutoff of an orientation depends on its place in the revisited list of ALL ranks."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.init()
    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip

    def all_reduce_min(keys_i64):
        t = torch.from_numpy(np.array(keys_i64, copy=True))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.numpy()

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    g = load("bb3_40x45_of_md")
    k = cfg_kwargs(g)
    S = 3
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"], max_rmsd=k["max_rmsd"],
                       max_dist=k["max_dist"], refine_while_searching=True, device=0, workspace_bytes=1 << 30,
                       of_enable=k["of_enable"], of_max_dist=k["of_max_dist"], of_min_matches=k["of_min_matches"],
                       num_solutions=S, alt_rmsd_tolerance=1.8, alt_by_pairs=True, alt_max_pairs_common=5)
    out = {}
    for label, w in (("split", world), ("whole", 1)):       # the same job by this rank alone: the oracle-checked path
        so = SearchOrHip(cfg, g["xyza"], g["xyzb"], rank=rank if w > 1 else 0, world_size=w,
                         all_reduce_min=all_reduce_min, all_gather=all_gather)
        print("rank", rank, "-alt-r with -of,", label, flush=True)
        so.Search(g["og"])
        st = so.Refine()
        out[label + "_iters"] = st.num_iterations
        for r in range(S):
            ssd, xf, pairs = so.engine.results_rank(r)
            out["%s_ssd%d" % (label, r)], out["%s_xf%d" % (label, r)], out["%s_pairs%d" % (label, r)] = ssd, xf, pairs
        so.close()
    np.savez(os.path.join(outdir, "altr_of_%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_alternates_by_pairs_with_the_filter_and_a_split_search(tmp_path):
    mp.spawn(_altr_of_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        z = np.load(tmp_path / ("altr_of_%d.npz" % rank))
        assert int(z["split_iters"]) == int(z["whole_iters"])
        found = 0
        for r in range(3):
            for what in ("ssd", "xf", "pairs"):
                a, b = z["split_%s%d" % (what, r)], z["whole_%s%d" % (what, r)]
                assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (rank, r, what)
            found += int((z["whole_ssd%d" % r] < np.float32(1e26)).sum())
        assert found > 20
