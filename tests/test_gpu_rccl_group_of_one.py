"""The exchange steps of a split search over RCCL itself.  RCCL refuses two
ranks on one GPU, so the two-rank GPU tests reduce over gloo; here ONE rank forms
an `nccl` process group and SearchOrHip runs every collective of the N > 1 path
with no stand-in: the MIN all-reduce in place on the engine's device key table,
the all-gather of the candidate records, the per-pass collectives of -alt-r.
With one rank they must change nothing: the tables are the reference's."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from golden_util import assert_tables_equal, cfg_kwargs, load

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, name, outdir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip

    g = load(name)
    k = cfg_kwargs(g)
    S = int(g["num_solutions"]) if "num_solutions" in g else 1
    alt = {}
    if S > 1:
        alt = dict(num_solutions=S, alt_min_3d_difference=float(g["alt_min_3d"]),
                   alt_rmsd_tolerance=float(g["alt_tol"]), refine_all=int(g["refine_method"]) == 2,
                   alt_by_pairs=int(g["alt_method"]) == 2, alt_max_pairs_common=int(g.get("alt_pairs", 0)))
    cfg = eng.make_cfg(k["N1"], k["N2"], num_bb=k["nbb"], n_min=k["n_min"], n_max=k["n_max"],
                       max_rmsd=k["max_rmsd"], max_dist=k["max_dist"],
                       refine_while_searching=k["refine_while_searching"], device=0, workspace_bytes=1 << 30,
                       of_enable=k["of_enable"], of_max_dist=k["of_max_dist"], of_min_matches=k["of_min_matches"],
                       **alt)
    so = SearchOrHip(cfg, g["xyza"], g["xyzb"], rank=0, world_size=1, always_reduce=True)
    print("nccl group of one, case", name, flush=True)
    og = g["og"] if "og" in g else so.generate_orientations(frag_len=int(g["fm"]))
    so.Search(og)
    out = {}
    for r in range(S):
        out["s_ssd%d" % r], out["s_xf%d" % r], out["s_pairs%d" % r] = so.engine.results_rank(r) if S > 1 else so.tables()[:3]
    refine = S == 1 or int(g["refine_method"]) != 0
    if refine:
        st = so.Refine()
        out["iters"] = st.num_iterations
        for r in range(S):
            out["r_ssd%d" % r], out["r_xf%d" % r], out["r_pairs%d" % r] = so.engine.results_rank(r) if S > 1 else so.tables()[:3]
    np.savez(os.path.join(outdir, "one.npz"), **out)
    so.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["syn60x50_c1", "bb3_40x45_of_md", "syn60x50_alt3", "syn60x50_altr3"])
def test_collectives_over_rccl_with_one_rank(name, tmp_path):
    mp.spawn(_worker, args=(_free_port(), name, str(tmp_path)), nprocs=1, join=True)
    g = load(name)
    S = int(g["num_solutions"]) if "num_solutions" in g else 1
    n_min = int(g["n_min"])
    z = np.load(tmp_path / "one.npz")
    for r in range(S):
        assert_tables_equal(z["s_ssd%d" % r], z["s_xf%d" % r], z["s_pairs%d" % r], g,
                            "search" if r == 0 else "search_alt%d" % r, n_min)
    if "iters" in z.files:
        assert int(z["iters"]) == int(g["refine_iters"])
        for r in range(S):
            assert_tables_equal(z["r_ssd%d" % r], z["r_xf%d" % r], z["r_pairs%d" % r], g,
                                "refine" if r == 0 else "refine_alt%d" % r, n_min)


def test_bench_with_a_group_of_one():
    """bench.py's N > 1 path (process group over RCCL, barriers, the maximum over
    the ranks, the second pair's sweep) with one rank: the same job as without."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    lines = {}
    for label, extra in (("plain", []), ("group", ["--group-of-one", "--extra-n", "250"])):
        print("bench.py", label, flush=True)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline"] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        js = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(js) == 1, p.stdout
        lines[label] = json.loads(js[0])
    a, b = lines["plain"], lines["group"]
    assert b["n_gpus"] == 1
    for k in ("orientations", "refine_iterations", "refine_dp_runs_executed"):
        assert a["config"][k] == b["config"][k]
    assert b["extra"]["orientations"] == 247 * 247 and b["extra"]["value"] > 0
    assert b["phases_ms"]["all_reduce"] > 0.0
    # one small all-reduce per search: the step must not get noticeably longer
    assert b["ms_per_step"] < 1.5 * a["ms_per_step"] + 5.0
