"""CPU stand-in for minrms_amd.engine.Engine built on the oracle -- TEST
INFRASTRUCTURE.  It lets the host logic above the C ABI (SearchOrHip: o_id
ranges per rank, exchange of keys, winners, refinement; bench.run_job) run in
the CPU suite, several ranks over gloo, with no GPU."""
import ctypes
import types

import numpy as np

from oracle import oracle_py as orc

NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


class OracleEngine:
    def __init__(self, ocfg, xyzA, xyzB, frag_len=4):
        self.cfg = ocfg
        self.a = np.ascontiguousarray(xyzA, np.float32)
        self.b = np.ascontiguousarray(xyzB, np.float32)
        self.range_n = ocfg.n_max - ocfg.n_min + 1
        self.frag_len = frag_len
        self.reset()

    # -- what SearchOrHip calls ------------------------------------------------
    def reset(self):
        self.keys = np.full(self.range_n, NONE, np.uint64)
        self.calls = 0                      # GoodOrientation calls so far (DistanceMetric's cutoff follows them)
        self.og = None
        self.tables = None
        self.origin = (0, 0)

    def gen_orientations(self, frag_len=4, cutoff_rmsd=-1.0, intervals=None, shuffle=True):
        return orc.gen_orientations(self.cfg, self.a, self.b, frag_len, cutoff_rmsd, intervals, shuffle)

    def _clip_after(self, calls):
        clip = orc.initial_clip(self.cfg)
        if self.cfg.of_enable:
            for _ in range(calls):
                clip = orc.lib().orc_clip_after_filter_call(ctypes.byref(self.cfg), clip)
        return clip

    def set_filter_calls(self, calls):
        self.calls = int(calls)

    def search(self, og, start=0, stop=None):
        og = np.ascontiguousarray(og, np.float32).reshape(-1, 12)
        stop = og.shape[0] if stop is None else stop
        self.og = og
        self.origin = (self.calls, start)
        t = orc.ResultTables(self.cfg)
        t.clip = self._clip_after(self.calls)
        t = orc.search(self.cfg, self.a, self.b, og, start, stop, tables=t, per_on=True)
        po = t.per_on
        if po.shape[0]:
            bits = np.where(np.isnan(po), np.float32(0), po).astype(np.float32).view(np.uint32).astype(np.uint64)
            oid = (np.arange(po.shape[0], dtype=np.uint64) + np.uint64(start))[:, None]
            keys = (bits << np.uint64(32)) | oid
            keys[np.isnan(po)] = NONE
            self.keys = np.minimum(self.keys, keys.min(axis=0))
        if self.cfg.of_enable:
            self.calls += stop - start
        self.tables = None

    def get_keys(self):
        return self.keys.copy()

    def set_keys(self, keys):
        self.keys = np.asarray(keys, np.uint64).copy()
        self.tables = None

    def finalize_search(self, allow_no_solution=False):
        t = orc.ResultTables(self.cfg)
        found = False
        calls0, start0 = self.origin
        for k in range(self.range_n):
            if self.keys[k] == NONE:
                continue
            found = True
            o = int(self.keys[k] & np.uint64(0xFFFFFFFF))
            one = orc.ResultTables(self.cfg)
            one.clip = self._clip_after(calls0 + (o - start0)) if self.cfg.of_enable else orc.initial_clip(self.cfg)
            one = orc.search(self.cfg, self.a, self.b, self.og, o, o + 1, tables=one)
            t.ssd[k], t.xform[k], t.pairs[k], t.winner[k] = one.ssd[k], one.xform[k], one.pairs[k], o
        t.clip = self._clip_after(self.calls)
        self.tables = t
        if not found and not allow_no_solution:
            raise RuntimeError("No solutions were found satisfying your criteria")

    def refine(self, max_iters=-1):
        if self.tables is None:
            self.finalize_search()
        iters, new = orc.refine(self.cfg, self.a, self.b, self.tables, max_iters)
        return types.SimpleNamespace(num_iterations=iters, num_new_solutions=new, dp_runs=iters - 1, rounds=iters - 1)

    def results(self):
        if self.tables is None:
            self.finalize_search(allow_no_solution=True)
        t = self.tables
        return t.ssd.copy(), t.xform.copy(), t.pairs.copy(), t.winner.copy()

    def close(self):
        pass


def make_for_bench(cfg, xyzA, xyzB):
    """Factory for bench.py's test hook: an OracleEngine for the engine configuration `cfg`.
    ORACLE_ENGINE_FAIL_RANK=r makes rank r raise (the launcher must then fail the run)."""
    import os
    if os.environ.get("ORACLE_ENGINE_FAIL_RANK") == os.environ.get("RANK", "0"):
        raise RuntimeError("rank %s fails on purpose (test of the launcher)" % os.environ.get("RANK", "0"))
    return OracleEngine(orc.make_cfg(cfg.N1, cfg.N2), xyzA, xyzB)
