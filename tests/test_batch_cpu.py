"""Host logic of the all-against-all batch on CPU: the deal of pairs to ranks,
the cost model, and a world_size-2 gloo run in which each rank solves its share
with the CPU oracle standing in for the GPU job; the gathered RMSD-vs-N curves
must be the reference's (one reference run per pair, tests/golden/batch_*.npz)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def load_batch(name):
    g = np.load(os.path.join(HERE, "golden", "batch_%s.npz" % name))
    structures = [g["xyz%d" % k] for k in range(len(g["lengths"]))]
    pairs = list(zip(g["pair_i"].tolist(), g["pair_j"].tolist()))
    return g, structures, pairs


def test_deal_is_a_balanced_partition():
    from minrms_amd.batch import all_pairs, deal_pairs, pair_cost
    rng = np.random.RandomState(3)
    sizes = rng.randint(80, 201, size=64).tolist()
    pairs = all_pairs(64)
    assert len(pairs) == 2016
    costs = [pair_cost(sizes[i], sizes[j]) for i, j in pairs]
    for world in (1, 2, 8):
        share = deal_pairs(costs, world)
        assert sorted(q for s in share for q in s) == list(range(len(pairs)))
        load = [sum(costs[q] for q in s) for s in share]
        assert max(load) - min(load) <= max(costs)
        assert max(load) <= 1.01 * sum(costs) / world + (max(costs) if world > 1 else 0)
    assert deal_pairs(costs, 8) == deal_pairs(costs, 8)
    assert deal_pairs([], 4) == [[], [], [], []]
    assert deal_pairs([5], 4) == [[0], [], [], []]


def test_pair_cost_counts_orientations_times_cells():
    from minrms_amd.batch import pair_cost
    # 150x150: (N1-3)(N2-3) orientations, sum_n (N1-n+1)(N2-n+1) cells
    assert pair_cost(150, 150) == 21609 * 1136275
    assert pair_cost(3, 50) == 0
    g, _, _ = load_batch("doms6")
    L = g["lengths"]
    for p, (i, j) in enumerate(zip(g["pair_i"], g["pair_j"])):
        cells = sum((L[i] - n + 1) * (L[j] - n + 1) for n in range(1, min(L[i], L[j]) + 1))
        assert pair_cost(int(L[i]), int(L[j])) == int(g["og_count"][p]) * cells


def _solve_with_oracle(i, j, a, b, frag_len=4, **kw):
    from oracle import oracle_py as orc
    from minrms_amd.batch import PairResult
    cfg = orc.make_cfg(a.shape[0], b.shape[0])
    og = orc.gen_orientations(cfg, a, b, frag_len=frag_len)
    t = orc.search(cfg, a, b, og)
    iters, _ = orc.refine(cfg, a, b, t)
    return PairResult(i, j, cfg.n_min, t.ssd.copy(), t.xform.copy(), None, iters, og.shape[0])


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minrms_amd.batch import align_pairs

    def all_reduce_sum(x):
        t = torch.from_numpy(np.array(x, copy=True))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    _, structures, pairs = load_batch("doms6")
    curves, mine = align_pairs(structures, pairs, rank=rank, world_size=world,
                               all_reduce_sum=all_reduce_sum, solve=_solve_with_oracle,
                               split_above=float("inf"))          # whole pairs only
    np.savez(out % rank, curves=curves, mine=np.array(sorted(mine)),
             iters=np.array([mine[p].refine_iterations for p in sorted(mine)]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_the_reference_curves(tmp_path):
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g, _, pairs = load_batch("doms6")
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert sorted(r0["mine"].tolist() + r1["mine"].tolist()) == list(range(len(pairs)))
    assert len(r0["mine"]) > 0 and len(r1["mine"]) > 0
    want = g["refine_ssd"][:, :r0["curves"].shape[1]]
    for r in (r0, r1):
        assert np.array_equal(r["curves"].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(r["iters"], g["refine_iters"][r["mine"]])


def test_plan_cuts_large_pairs_into_ranges():
    from minrms_amd.batch import plan_units
    units, costs = plan_units([100, 5, 5, 5], 4)
    assert units[:4] == [(0, 0, 4), (0, 1, 4), (0, 2, 4), (0, 3, 4)] and units[4:] == [(1, 0, 1), (2, 0, 1), (3, 0, 1)]
    assert sum(costs) <= 115 and max(costs) == 25
    assert plan_units([100, 5], 1)[0] == [(0, 0, 1), (1, 0, 1)]              # one rank: nothing to cut
    units, _ = plan_units([7, 7, 7, 7, 7, 7, 7, 7], 2)                       # equal pairs of a large batch stay whole
    assert all(u[2] == 1 for u in units)


def _make_oracle_engine(a, b, **kw):
    from oracle import oracle_py as orc
    from oracle_engine import OracleEngine
    return OracleEngine(orc.make_cfg(a.shape[0], b.shape[0]), a, b)


def _worker_cut(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from minrms_amd.batch import align_pairs, pair_cost

    def all_reduce(op):
        def f(x):
            t = torch.from_numpy(np.array(x, copy=True))
            dist.all_reduce(t, op=op)
            return t.numpy()
        return f

    g, structures, pairs = load_batch("doms6")
    L = g["lengths"]
    costs = sorted(pair_cost(int(L[i]), int(L[j])) for i, j in pairs)
    curves, mine = align_pairs(structures, pairs, rank=rank, world_size=world, solve=_solve_with_oracle,
                               all_reduce_sum=all_reduce(dist.ReduceOp.SUM), all_reduce_min=all_reduce(dist.ReduceOp.MIN),
                               make_engine=_make_oracle_engine, split_above=costs[len(costs) // 2])
    np.savez(out % rank, curves=curves, mine=np.array(sorted(mine)),
             iters=np.array([mine[p].refine_iterations for p in sorted(mine)]))
    dist.barrier()
    dist.destroy_process_group()


def test_pairs_cut_into_orientation_ranges_over_two_ranks(tmp_path):
    """The larger half of the pairs is cut into two o_id ranges each, one per
    rank; their keys meet in one MIN all-reduce and the first range's rank
    finishes the pair.  Curves and iteration counts are the reference's."""
    out = str(tmp_path / "c%d.npz")
    mp.spawn(_worker_cut, args=(2, _free_port(), out), nprocs=2, join=True)
    g, _, pairs = load_batch("doms6")
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert sorted(r0["mine"].tolist() + r1["mine"].tolist()) == list(range(len(pairs)))
    want = g["refine_ssd"][:, :r0["curves"].shape[1]]
    for r in (r0, r1):
        assert np.array_equal(r["curves"].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(r["iters"], g["refine_iters"][r["mine"]])
