"""Batches of structure pairs (all against all), pairs and orientations dealt
over the GPUs.

The reference aligns one pair per program run (bin/minrms/main.cc:59-176:
parse two structures, generate orientations, Search, Refine, write); a set of
domains is aligned by running it once per pair.  Every pair is an independent
job, and inside a pair the orientations are independent (Search(og, start,
stop), bin/minrms/search_or.h:83-103), so the batch is one flat list of work
units (pair, o_id range): a pair whose sweep is small against a rank's share of
the batch is one unit, a larger one is cut into several o_id ranges.  Each rank
(one process per GPU) takes a share of the units balanced by their estimated DP
work.  Whole pairs need no collective at all; the pieces of a cut pair meet in
ONE all-reduce(MIN) of their per-N keys (all cut pairs in one call), after which
the rank that swept the pair's first range re-solves the winners and refines.
The RMSD-vs-N curves are collected at the end with one all-reduce in which every
element has exactly one non-zero contributor (x + 0 == x, so the sum is exact).
"""
import time

import numpy as np

from . import engine as eng
from .search_or import SearchOrHip, shard_range, keys_to_i64, i64_to_keys

_I64_MAX = np.iinfo(np.int64).max


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


def plan_units(costs, world_size, split_above=None):
    """Work units of a batch: [(pair index, part, parts)], and their costs.

    A pair is cut into o_id ranges when its sweep exceeds `split_above` DP cells
    (default: a quarter of one rank's share of the batch, so that no single pair
    can unbalance the deal; never with one rank), into as many ranges as keep
    each below that size, at most one per rank."""
    total = sum(costs)
    if split_above is None:
        split_above = total / (4.0 * world_size) if world_size > 1 else float("inf")
    units, unit_costs = [], []
    for p, c in enumerate(costs):
        parts = 1
        if world_size > 1 and c > split_above > 0:
            parts = int(min(world_size, -(-c // max(1, int(split_above)))))
        for q in range(parts):
            units.append((p, q, parts))
            unit_costs.append(c // parts)
    return units, unit_costs


def make_engine_hip(xyz_i, xyz_j, device=0, **cfg_kwargs):
    """The engine of one pair (tests of the host logic inject a CPU stand-in with the same methods)."""
    nbb = int(cfg_kwargs.get("num_bb", 1))
    n1 = int(np.asarray(xyz_i).reshape(-1, 3).shape[0]) // nbb
    n2 = int(np.asarray(xyz_j).reshape(-1, 3).shape[0]) // nbb
    return eng.Engine(eng.make_cfg(n1, n2, device=device, **cfg_kwargs), xyz_i, xyz_j)


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
                frag_len=4, concurrency=1, split_above=None, all_reduce_min=None, make_engine=make_engine_hip,
                refine=True, **solve_kwargs):
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
    split_above: pairs whose sweep has more DP cells than this are cut into o_id
    ranges dealt to different ranks (plan_units); all_reduce_min: callable(int64
    ndarray) -> ndarray for their keys; make_engine: callable(xyz_i, xyz_j,
    **cfg) -> engine for the pieces of cut pairs (tests inject a CPU stand-in).
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
    units, unit_costs = plan_units(costs, world_size, split_above)
    share = [units[u] for u in deal_pairs(unit_costs, world_size)[rank]]
    cost_of_unit = {u: c for u, c in zip(units, unit_costs)}
    width = max([min(sizes[i], sizes[j]) for i, j in pairs] + [1])
    curves = np.zeros((len(pairs), width), np.float32)
    mine = {}

    def job(unit):
        p = unit[0]
        i, j = pairs[p]
        t0 = time.perf_counter()
        res = solve(i, j, structures[i], structures[j], frag_len=frag_len, **solve_kwargs)
        res.seconds = time.perf_counter() - t0
        return p, res

    whole = [u for u in share if u[2] == 1]
    pieces = [u for u in share if u[2] > 1]
    if concurrency > 1 and solve is solve_pair_hip and "workspace_bytes" not in solve_kwargs:
        solve_kwargs["workspace_bytes"] = (96 << 30) // concurrency     # the engines share the GPU's memory
    if concurrency > 1 and len(whole) > 1:
        from concurrent.futures import ThreadPoolExecutor
        # largest jobs first, so that the tail of the batch is made of small ones
        order = sorted(whole, key=lambda u: (-cost_of_unit[u], u))
        with ThreadPoolExecutor(max_workers=concurrency) as ex:
            done = list(ex.map(job, order))
    else:
        done = [job(u) for u in whole]

    # ---- pairs cut into o_id ranges -------------------------------------------
    cut = sorted({u[0] for u in units if u[2] > 1})           # the same list on every rank
    if cut:
        row = {p: r for r, p in enumerate(cut)}
        keys = np.full((len(cut), width), _I64_MAX, np.int64)
        og_of_pair = {}                                         # orientation lists of the pairs this rank will finish
        for p, q, parts in pieces:                              # this rank's ranges: sweep, keep the keys
            i, j = pairs[p]
            e = make_engine(structures[i], structures[j], **solve_kwargs)
            try:
                og = e.gen_orientations(frag_len=frag_len)
                if q == 0:
                    og_of_pair[p] = og
                lo, hi = shard_range(og.shape[0], q, parts)
                e.set_filter_calls(lo)                          # -of: the calls that precede this range
                e.search(og, lo, hi)
                k = keys_to_i64(e.get_keys())
                keys[row[p], :k.size] = np.minimum(keys[row[p], :k.size], k)
            finally:
                e.close()
        if all_reduce_min is not None:
            keys = all_reduce_min(keys)
        else:
            import torch
            import torch.distributed as dist
            dev = torch.device("cuda", int(solve_kwargs.get("device", 0)))   # the GPU the engines of this rank use
            t = torch.from_numpy(keys).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)            # RCCL; all cut pairs in one call
            keys = t.cpu().numpy()
        for p, q, parts in pieces:                              # the rank of a pair's first range finishes it
            if q != 0:
                continue
            i, j = pairs[p]
            t0 = time.perf_counter()
            # a new engine (the one that swept did not sit on its share of the GPU's memory across the
            # all-reduce; its blocks are handed back by the pool), the orientation list kept from the sweep
            e = make_engine(structures[i], structures[j], **solve_kwargs)
            try:
                og = og_of_pair.pop(p)
                e.search(og, 0, 0)                              # registers the orientation list, sweeps nothing
                rn = e.range_n
                e.set_keys(i64_to_keys(keys[row[p], :rn]))
                e.set_filter_calls(og.shape[0])
                e.finalize_search()
                iters = e.refine().num_iterations if refine else 0
                ssd, xf, prs, _ = e.results()
                nm = e.cfg.n_min
                al = [prs[k, :nm + k].copy() if ssd[k] < eng.UPPER_BOUND else None for k in range(rn)]
                res = PairResult(i, j, nm, ssd.copy(), xf.reshape(-1, 3, 4).copy(), al, iters, og.shape[0])
                res.seconds = time.perf_counter() - t0
                done.append((p, res))
            finally:
                e.close()

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
            t = torch.from_numpy(curves).to(torch.device("cuda", int(solve_kwargs.get("device", 0))))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)      # RCCL; one contributor per element
            curves = t.cpu().numpy()
    return curves, mine
