"""The complete job of bench.py (reset -> orientations -> Search with one o_id
range per rank and the MIN exchange of keys -> winners -> Refine -> tables) run
by two gloo ranks on the CPU, the oracle standing in for the GPU engine under
the real SearchOrHip.  Both ranks must end with the reference's tables."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import load, cfg_kwargs, assert_tables_equal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle import oracle_py as orc
    from oracle_engine import OracleEngine
    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip
    g = load(name)
    k = cfg_kwargs(g)
    ocfg = orc.make_cfg(**k)
    ecfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                        max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                        refine_while_searching=k["refine_while_searching"], of_enable=k["of_enable"],
                        of_max_dist=k["of_max_dist"], of_min_matches=k["of_min_matches"])

    def all_reduce_min(keys_i64):
        t = torch.from_numpy(np.array(keys_i64, copy=True))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.numpy()

    so = SearchOrHip(ecfg, g["xyza"], g["xyzb"], rank=rank, world_size=world, all_reduce_min=all_reduce_min,
                     engine=OracleEngine(ocfg, g["xyza"], g["xyzb"]))
    n_or, dp_runs, iters = bench.run_job(so)
    ssd, xf, pairs, win = so.tables()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), ssd=ssd, xf=xf, pairs=pairs, win=win, n_or=n_or, iters=iters)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ship41x13_full", "ship41x13_minN5", "ship41x13_of"])
def test_bench_job_on_two_ranks(name, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
    g = load(name)
    for rank in range(2):
        r = np.load(tmp_path / ("rank%d.npz" % rank))
        assert int(r["n_or"]) == int(g["og_count"])
        assert int(r["iters"]) == int(g["refine_iters"])
        assert_tables_equal(r["ssd"], r["xf"], r["pairs"], g, "refine", int(g["n_min"]))
