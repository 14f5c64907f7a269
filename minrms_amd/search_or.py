"""Host-side mirror of the reference's search plug-in interface.

`SearchOrHip` has the shape of `class SearchOr` (bin/minrms/search_or.h:57-317
of the reference): it is constructed from the settings, `Search(og, start,
stop)` sweeps an o_id range, `Refine()` runs iterative refinement, and the three
result tables that the reference's writers read are exposed as
`aaSumSqdDists`, `aaTransforms`, `aaAlignments` (search_or_dyn3d.h:293-310,
r = 0).  All work happens in the HIP engine behind the C ABI.

With `world_size > 1` (one process per GPU, torch.distributed initialised with
the `nccl` backend = RCCL) `Search(og)` takes this rank's contiguous share of
the o_id range and finishes with one MIN all-reduce of the per-N keys, the seam
search_or.h:92-103 describes.
"""
import time

import numpy as np

from . import engine as eng

_NONE_U64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_I64_MAX = np.iinfo(np.int64).max


def shard_range(n, rank, world_size):
    """Contiguous o_id range [start, stop) of `rank` among `world_size` ranks."""
    base, rem = divmod(n, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def keys_to_i64(keys_u64):
    """uint64 keys -> int64 values with the same order ("none" = int64 max)."""
    k = np.asarray(keys_u64, np.uint64)
    out = k.astype(np.int64)          # real keys are < 2^63 (ssd bits < 0x7f800000)
    out[k == _NONE_U64] = _I64_MAX
    return out


def i64_to_keys(vals_i64):
    v = np.asarray(vals_i64, np.int64)
    out = v.astype(np.uint64)
    out[v == _I64_MAX] = _NONE_U64
    return out


class _DeviceArray:
    """range_n int64 at a raw device address, for torch.as_tensor (no copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class SearchOrHip:
    def __init__(self, cfg, xyzA, xyzB, rank=0, world_size=1, all_reduce_min=None,
                 device_all_reduce_min=None, engine=None, all_gather=None, always_reduce=False):
        """cfg: engine.Cfg.
        all_reduce_min: callable(int64 ndarray) -> int64 ndarray: the keys make a
        round trip through host memory and this callable (CPU tests: gloo).
        device_all_reduce_min: callable(int64 CUDA tensor), in place: stands in for
        torch.distributed.all_reduce on the engine's own key table (tests on one GPU).
        Neither given: torch.distributed.all_reduce(MIN) over RCCL, in place on the
        engine's key table in device memory.
        engine: an object with Engine's methods (tests run this class on the CPU
        with a stand-in); default: the HIP engine.
        all_gather: callable(object) -> list with every rank's object, in rank order
        (alternate solutions: the candidate records of all ranks); default
        torch.distributed.all_gather_object.
        always_reduce: run the exchange steps of a split search with one rank too (a
        process group of one: the collectives execute and change nothing)."""
        self.cfg = cfg
        self.engine = engine if engine is not None else eng.Engine(cfg, xyzA, xyzB)
        self.rank, self.world_size = rank, world_size
        self._all_reduce_min = all_reduce_min
        self._device_all_reduce_min = device_all_reduce_min
        self._all_gather = all_gather
        self._split = world_size > 1 or always_reduce
        self.n_min, self.n_max = cfg.n_min, cfg.n_max
        self.range_of_n = cfg.n_max - cfg.n_min + 1
        self._tables = None
        self._filter_calls = 0     # GoodOrientation calls of the whole job so far
        # host seconds spent in the stages of Search(), summed over calls: "sweep" (this rank's o_id
        # range), "all_reduce" (exchange of candidates and keys), "winners" (finalize_search)
        self.phase_s = {}

    # -- orientation list (lib/or_gen) ------------------------------------
    def generate_orientations(self, frag_len=4, cutoff_rmsd=-1.0, intervals=None):
        return self.engine.gen_orientations(frag_len, cutoff_rmsd, intervals, shuffle=True)

    def reset(self):
        self.engine.reset()
        self._tables = None
        self._filter_calls = 0

    # -- SearchOr::Search ---------------------------------------------------
    def Search(self, og, start=None, stop=None):
        og = np.ascontiguousarray(og, np.float32).reshape(-1, 12)
        if start is None:
            start, stop = 0, og.shape[0]
        if self._split and self.cfg.num_solutions > 1 and self.cfg.alt_by_pairs:
            # MultiPass (search_or_dyn3d.cc:1969-2250): every further pass sweeps this rank's part of the
            # orientations it revisits and ends in one MIN all-reduce of the pass's keys
            self.engine.set_collectives(self._allreduce_min_u64, self._allgather_i64)
        t0 = t1 = time.perf_counter()
        if self._split:
            s, e = shard_range(stop - start, self.rank, self.world_size)
            # the -of filter changes process-wide state in the reference with
            # every call: tell the engine how many calls precede its share
            self.engine.set_filter_calls(self._filter_calls + s)
            self.engine.search(og, start + s, start + e)
            t1 = time.perf_counter()
            if self.cfg.num_solutions > 1:
                # every rank kept a superset of its part of the one-process candidate lists:
                # all ranks get the union and rebuild the lists from it
                parts = self._gather(self.engine.alt_candidates())
                self.engine.alt_set_candidates(*[np.concatenate([p[q] for p in parts]) for q in range(5)])
            if self._all_reduce_min is not None:
                keys = keys_to_i64(self.engine.get_keys())
                self.engine.set_keys(i64_to_keys(self._all_reduce_min(keys)))
            else:
                self._reduce_on_device()
            self.engine.set_filter_calls(self._filter_calls + (stop - start))
        else:
            self.engine.search(og, start, stop)
            t1 = time.perf_counter()
        if self.cfg.of_enable:
            self._filter_calls += stop - start
        t2 = time.perf_counter()
        self.engine.finalize_search()
        t3 = time.perf_counter()
        for name, dt in (("sweep", t1 - t0), ("all_reduce", t2 - t1), ("winners", t3 - t2)):
            self.phase_s[name] = self.phase_s.get(name, 0.0) + dt
        self._tables = None

    def _gather(self, obj):
        if self._all_gather is not None:
            return self._all_gather(obj)
        import torch.distributed as dist
        out = [None] * self.world_size
        dist.all_gather_object(out, obj)
        return out

    def _allreduce_min_u64(self, keys_u64):
        """MIN over the ranks of range_n uint64 keys held in host memory (the keys of one pass of -alt-r)."""
        k = keys_to_i64(keys_u64)
        if self._all_reduce_min is not None:
            return i64_to_keys(self._all_reduce_min(k))
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(k).to(torch.device("cuda", self.cfg.device))
        if self._device_all_reduce_min is not None:
            self._device_all_reduce_min(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)  # RCCL
        return i64_to_keys(t.cpu().numpy())

    def _allgather_i64(self, mine):
        parts = self._gather(np.asarray(mine, np.int64))
        return np.concatenate([np.asarray(p, np.int64).reshape(-1) for p in parts]) if parts else np.zeros(0, np.int64)

    def _reduce_on_device(self):
        """MIN all-reduce of the per-N keys in place on the engine's device table
        (range_n x 8 bytes: one small RCCL call, no host hop).  The table holds
        uint64 with all-ones = "no solution"; as int64 that is -1, which MIN would
        prefer, so it becomes int64 max for the reduction and -1 again after."""
        import torch
        import torch.distributed as dist
        dev = torch.device("cuda", self.cfg.device)
        ptr = self.engine.keys_device_ptr()          # waits for the engine's stream
        t = torch.as_tensor(_DeviceArray(ptr, self.range_of_n), device=dev)
        t.masked_fill_(t == -1, _I64_MAX)
        if self._device_all_reduce_min is not None:
            self._device_all_reduce_min(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MIN)  # RCCL over xGMI
        t.masked_fill_(t == _I64_MAX, -1)
        torch.cuda.current_stream(dev).synchronize()  # the engine works on a stream of its own

    # -- SearchOr::Refine ---------------------------------------------------
    def Refine(self, max_iters=-1):
        st = self.engine.refine(max_iters)
        self._tables = None
        return st

    # -- result tables --------------------------------------------------------
    def tables(self):
        if self._tables is None:
            self._tables = self.engine.results()
        return self._tables

    @property
    def aaSumSqdDists(self):
        return self.tables()[0]

    @property
    def aaTransforms(self):
        return self.tables()[1].reshape(-1, 3, 4)

    @property
    def aaAlignments(self):
        ssd, _, pairs, _ = self.tables()
        return [pairs[k, :self.n_min + k] if ssd[k] < eng.UPPER_BOUND else None
                for k in range(self.range_of_n)]

    def rank_tables(self, rank):
        """(aaSumSqdDists[rank], aaTransforms[rank], aaAlignments[rank]) for
        cfg.num_solutions > 1: rank 0 = best, 1 = 2nd best, ..."""
        ssd, xf, pairs = self.engine.results_rank(rank)
        al = [pairs[k, :self.n_min + k] if ssd[k] < eng.UPPER_BOUND else None
              for k in range(self.range_of_n)]
        return ssd, xf.reshape(-1, 3, 4), al

    @property
    def winner_oid(self):
        return self.tables()[3]

    def rmsd(self):
        """sqrt(ssd / (n * num_bb)) per n, NaN where there is no solution
        (search_or_dyn3d.cc:2867-2869)."""
        ssd = self.aaSumSqdDists.astype(np.float32)
        n = np.arange(self.n_min, self.n_max + 1, dtype=np.float32) * np.float32(self.cfg.num_bb)
        out = np.sqrt(ssd / n).astype(np.float32)
        out[ssd >= eng.UPPER_BOUND] = np.nan
        return out

    def close(self):
        self.engine.close()
