"""Batches of structure pairs (all against all), pairs dealt over the GPUs.

The reference aligns one pair per program run (bin/minrms/main.cc:59-176:
parse two structures, generate orientations, Search, Refine, write); a set of
domains is aligned by running it once per pair.  Every pair is an independent
job, so a batch shards by pair: each rank (one process per GPU) takes a share
of the pairs balanced by their estimated DP work and runs the complete job for
each of them on its own GPU.  The data path has no collective; the RMSD-vs-N
curves are collected at the end with one all-reduce in which every element has
exactly one non-zero contributor (x + 0 == x, so the sum is exact).
"""
import time

import numpy as np

from . import engine as eng
from .search_or import SearchOrHip


def all_pairs(count):
    """(i, j), i < j, in row-major order."""
    return [(i, j) for i in range(count) for j in range(i + 1, count)]


def pair_cost(n1, n2, frag_len=4, n_min=1):
    """DP cells of the full orientation sweep of one pair: one orientation per
    pair of fragments (lib/or_gen/or_gen.cc:507-509) times the cells of the
    3-index table (lib/dyn/dyn3d.cc:188-189)."""
    n_max = min(n1, n2)
    wa, wb = n1 - n_min + 1, n2 - n_min + 1
    n = np.arange(1, n_max + 1, dtype=np.int64)
    rows = np.minimum(n1, n + wa - 1) - n + 1
    cols = np.minimum(n2, n + wb - 1) - n + 1
    orientations = max(0, n1 - frag_len + 1) * max(0, n2 - frag_len + 1)
    return int(orientations) * int((rows * cols).sum())


def deal_pairs(costs, world_size):
    """Longest-job-first assignment of jobs to ranks: jobs in order of falling
    cost (ties: lower index first), each to the least loaded rank (ties: lower
    rank).  Returns one ascending index list per rank.  Deterministic, so every
    rank computes the same assignment without communication."""
    order = sorted(range(len(costs)), key=lambda q: (-costs[q], q))
    load = [0] * world_size
    share = [[] for _ in range(world_size)]
    for q in order:
        r = min(range(world_size), key=lambda x: (load[x], x))
        share[r].append(q)
        load[r] += costs[q]
    return [sorted(s) for s in share]


class PairResult:
    """Result tables of one pair (what SearchOrDyn3d holds after Refine,
    search_or_dyn3d.h:293-310)."""

    def __init__(self, i, j, n_min, ssd, transforms, alignments, refine_iterations, orientations):
        self.i, self.j, self.n_min = i, j, n_min
        self.ssd, self.transforms, self.alignments = ssd, transforms, alignments
        self.refine_iterations, self.orientations = refine_iterations, orientations


def solve_pair_hip(i, j, xyz_i, xyz_j, device=0, refine=True, frag_len=4, **cfg_kwargs):
    """The complete job for one pair on the HIP engine."""
    nbb = int(cfg_kwargs.get("num_bb", 1))
    n1 = int(np.asarray(xyz_i).reshape(-1, 3).shape[0]) // nbb
    n2 = int(np.asarray(xyz_j).reshape(-1, 3).shape[0]) // nbb
    cfg = eng.make_cfg(n1, n2, device=device, **cfg_kwargs)
    so = SearchOrHip(cfg, xyz_i, xyz_j)
    try:
        og = so.generate_orientations(frag_len=frag_len)
        so.Search(og)
        iters = so.Refine().num_iterations if refine else 0
        ssd, xf, _, _ = so.tables()
        return PairResult(i, j, cfg.n_min, ssd.copy(), xf.reshape(-1, 3, 4).copy(),
                          [a.copy() if a is not None else None for a in so.aaAlignments],
                          iters, og.shape[0])
    finally:
        so.close()


def align_pairs(structures, pairs=None, rank=0, world_size=1, all_reduce_sum=None, solve=solve_pair_hip,
                frag_len=4, concurrency=1, **solve_kwargs):
    """Run the complete job (orientations, Search, Refine) for every pair.

    structures: list of (N_k * num_bb, 3) float32 backbone coordinates.
    pairs: list of (i, j); default all_pairs(len(structures)).
    Returns (curves, mine): curves[p] = float32 sums of squared distances per n
    (column n - 1; eng.UPPER_BOUND = no solution) of every pair of the batch,
    identical on all ranks; mine = {p: PairResult} for the pairs this rank
    computed.
    all_reduce_sum: callable(float32 ndarray) -> ndarray, used instead of
    torch.distributed (tests inject a gloo/CPU version).
    solve: the per-pair job; tests of the host logic inject a CPU stand-in.
    concurrency: pairs in flight on this rank's GPU, each with its own engine
    and stream on a host thread of its own (the engine calls release the GIL).
    The refinement of one pair is a chain of small launches that leaves most of
    the GPU idle; the sweep of another pair fills it.
    """
    if pairs is None:
        pairs = all_pairs(len(structures))
    nbb = int(solve_kwargs.get("num_bb", 1))
    sizes = [int(np.asarray(s).reshape(-1, 3).shape[0]) // nbb for s in structures]
    n_min = int(solve_kwargs.get("n_min", 1))
    costs = [pair_cost(sizes[i], sizes[j], frag_len, n_min) for i, j in pairs]
    share = deal_pairs(costs, world_size)[rank]
    width = max([min(sizes[i], sizes[j]) for i, j in pairs] + [1])
    curves = np.zeros((len(pairs), width), np.float32)
    mine = {}

    def job(p):
        i, j = pairs[p]
        t0 = time.perf_counter()
        res = solve(i, j, structures[i], structures[j], frag_len=frag_len, **solve_kwargs)
        res.seconds = time.perf_counter() - t0
        return p, res

    if concurrency > 1 and solve is solve_pair_hip and "workspace_bytes" not in solve_kwargs:
        solve_kwargs["workspace_bytes"] = (96 << 30) // concurrency     # the engines share the GPU's memory
    if concurrency > 1 and len(share) > 1:
        from concurrent.futures import ThreadPoolExecutor
        # largest jobs first, so that the tail of the batch is made of small ones
        order = sorted(share, key=lambda q: (-costs[q], q))
        with ThreadPoolExecutor(max_workers=concurrency) as ex:
            done = list(ex.map(job, order))
    else:
        done = [job(p) for p in share]
    for p, res in done:
        mine[p] = res
        curves[p, :] = eng.UPPER_BOUND
        curves[p, res.n_min - 1:res.n_min - 1 + res.ssd.shape[0]] = res.ssd
    if world_size > 1:
        if all_reduce_sum is not None:
            curves = all_reduce_sum(curves)
        else:
            import torch
            import torch.distributed as dist
            t = torch.from_numpy(curves).cuda()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)      # RCCL; one contributor per element
            curves = t.cpu().numpy()
    return curves, mine
