"""ctypes binding of libminrms_hip.so, the C-ABI engine (include/minrms_engine.h).

There is no CPU fallback: if the HIP library is missing or no GPU is usable the
calls raise.  Build the library first with `python -m minrms_amd.build`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libminrms_hip.so")

UPPER_BOUND = np.float32(1.0e26)

OK, E_ARG, E_DEVICE, E_NOSOLUTION, E_STATE = 0, -1, -2, -3, -4

# every symbol include/minrms_engine.h declares
SYMBOLS = [
    "minrms_engine_create", "minrms_engine_destroy", "minrms_engine_reset", "minrms_engine_gen_orientations",
    "minrms_engine_search", "minrms_engine_get_keys", "minrms_engine_set_keys",
    "minrms_engine_keys_device_ptr", "minrms_engine_finalize_search", "minrms_engine_refine",
    "minrms_engine_results", "minrms_engine_cells_per_orientation",
    "minrms_engine_enable_timing", "minrms_engine_dp_block_time", "minrms_engine_num_blocks",
    "minrms_engine_filter_verdicts", "minrms_engine_set_filter_calls", "minrms_engine_results_rank",
    "minrms_engine_count_updates", "minrms_engine_get_update_counts",
    "minrms_engine_last_error", "minrms_engine_trim", "minrms_engine_dp_block_shape",
    "minrms_search_noalloc", "minrms_noalloc_prepare", "minrms_engine_kernel_time", "minrms_engine_coop_shape",
    "minrms_engine_dp_block_columns",
    "minrms_engine_alt_num_candidates", "minrms_engine_alt_get_candidates", "minrms_engine_alt_set_candidates",
    "minrms_engine_set_collectives",
]

# callbacks of minrms_engine_set_collectives
ALLREDUCE_MIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_int64)
ALLGATHER_I64_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64), C.c_int64,
                               C.POINTER(C.c_int64))


def trim():
    """Return the device memory kept from closed engines to the system."""
    load_library().minrms_engine_trim()


class MinrmsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("minrms engine error %d: %s" % (code, msg))
        self.code = code


class Cfg(C.Structure):
    _fields_ = [("N1", C.c_int32), ("N2", C.c_int32), ("num_bb", C.c_int32),
                ("n_min", C.c_int32), ("n_max", C.c_int32),
                ("max_rmsd", C.c_float), ("max_dist", C.c_float),
                ("refine_while_searching", C.c_int32),
                ("of_enable", C.c_int32), ("of_max_dist", C.c_float),
                ("of_min_matches", C.c_int32),
                ("device", C.c_int32), ("reserved", C.c_int32),
                ("workspace_bytes", C.c_int64),
                ("num_solutions", C.c_int32), ("alt_min_3d_difference", C.c_float),
                ("alt_rmsd_tolerance", C.c_float), ("refine_all", C.c_int32),
                ("alt_by_pairs", C.c_int32), ("alt_max_pairs_common", C.c_int32)]


class RefineStats(C.Structure):
    _fields_ = [("num_iterations", C.c_int64), ("num_new_solutions", C.c_int64),
                ("dp_runs", C.c_int64), ("rounds", C.c_int64)]


_lib = None


def load_library():
    """dlopen the engine and declare its prototypes.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("HIP engine not built: %s is missing (run `python -m minrms_amd.build`)"
                          % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.minrms_engine_create.restype = C.c_int
    L.minrms_engine_create.argtypes = [C.POINTER(Cfg), fp, fp, C.POINTER(vp)]
    L.minrms_engine_destroy.restype = None
    L.minrms_engine_destroy.argtypes = [vp]
    L.minrms_engine_trim.restype = None
    L.minrms_engine_trim.argtypes = []
    L.minrms_engine_reset.restype = C.c_int
    L.minrms_engine_reset.argtypes = [vp]
    L.minrms_engine_gen_orientations.restype = C.c_int
    L.minrms_engine_gen_orientations.argtypes = [vp, C.c_int32, C.c_float, C.c_int32,
                                                 C.POINTER(C.c_int32), C.c_int32, fp, C.c_int64,
                                                 C.POINTER(C.c_int64)]
    L.minrms_engine_search.restype = C.c_int
    L.minrms_engine_search.argtypes = [vp, fp, C.c_int64, C.c_int64, C.c_int64]
    L.minrms_engine_get_keys.restype = C.c_int
    L.minrms_engine_get_keys.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.minrms_engine_set_keys.restype = C.c_int
    L.minrms_engine_set_keys.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.minrms_engine_keys_device_ptr.restype = C.c_int
    L.minrms_engine_keys_device_ptr.argtypes = [vp, C.POINTER(vp)]
    L.minrms_engine_finalize_search.restype = C.c_int
    L.minrms_engine_finalize_search.argtypes = [vp]
    L.minrms_engine_refine.restype = C.c_int
    L.minrms_engine_refine.argtypes = [vp, C.c_int64, C.POINTER(RefineStats)]
    L.minrms_engine_results.restype = C.c_int
    L.minrms_engine_results.argtypes = [vp, fp, fp, C.POINTER(C.c_int16), C.POINTER(C.c_int64)]
    L.minrms_engine_cells_per_orientation.restype = C.c_int64
    L.minrms_engine_cells_per_orientation.argtypes = [C.POINTER(Cfg)]
    L.minrms_engine_enable_timing.restype = C.c_int
    L.minrms_engine_enable_timing.argtypes = [vp, C.c_int32]
    L.minrms_engine_dp_block_time.restype = C.c_int
    L.minrms_engine_dp_block_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double),
                                              C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                              C.POINTER(C.c_int64)]
    L.minrms_engine_dp_block_shape.restype = C.c_int
    L.minrms_engine_dp_block_shape.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32)]
    L.minrms_engine_dp_block_columns.restype = C.c_int
    L.minrms_engine_dp_block_columns.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
    L.minrms_engine_num_blocks.restype = C.c_int
    L.minrms_engine_num_blocks.argtypes = [vp]
    L.minrms_engine_count_updates.restype = C.c_int
    L.minrms_engine_count_updates.argtypes = [vp, C.c_int32]
    L.minrms_engine_get_update_counts.restype = C.c_int
    L.minrms_engine_get_update_counts.argtypes = [vp, C.POINTER(C.c_int32), C.c_int64]
    L.minrms_engine_filter_verdicts.restype = C.c_int
    L.minrms_engine_filter_verdicts.argtypes = [vp, C.POINTER(C.c_uint8), C.c_int64,
                                                C.POINTER(C.c_int64)]
    L.minrms_engine_set_filter_calls.restype = C.c_int
    L.minrms_engine_set_filter_calls.argtypes = [vp, C.c_int64]
    L.minrms_engine_results_rank.restype = C.c_int
    L.minrms_engine_results_rank.argtypes = [vp, C.c_int32, fp, fp, C.POINTER(C.c_int16)]
    L.minrms_search_noalloc.restype = C.c_int
    L.minrms_search_noalloc.argtypes = [C.POINTER(Cfg), vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    L.minrms_noalloc_prepare.restype = C.c_int
    L.minrms_noalloc_prepare.argtypes = [C.c_int32]
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    L.minrms_engine_kernel_time.restype = C.c_int
    L.minrms_engine_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), i64p]
    L.minrms_engine_coop_shape.restype = C.c_int
    L.minrms_engine_coop_shape.argtypes = [vp, i32p, i32p, i64p, i32p, i32p, i32p]
    L.minrms_engine_alt_num_candidates.restype = C.c_int
    L.minrms_engine_alt_num_candidates.argtypes = [vp, i64p]
    L.minrms_engine_alt_get_candidates.restype = C.c_int
    L.minrms_engine_alt_get_candidates.argtypes = [vp, i32p, i64p, fp, fp, fp]
    L.minrms_engine_alt_set_candidates.restype = C.c_int
    L.minrms_engine_alt_set_candidates.argtypes = [vp, C.c_int64, i32p, i64p, fp, fp, fp]
    L.minrms_engine_set_collectives.restype = C.c_int
    L.minrms_engine_set_collectives.argtypes = [vp, ALLREDUCE_MIN_FN, ALLGATHER_I64_FN, vp]
    L.minrms_engine_last_error.restype = C.c_char_p
    L.minrms_engine_last_error.argtypes = []
    _lib = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _check(rc, allow=()):
    if rc != OK and rc not in allow:
        raise MinrmsError(rc, load_library().minrms_engine_last_error().decode(errors="replace"))
    return rc


def make_cfg(N1, N2, num_bb=1, n_min=1, n_max=-1, max_rmsd=-1.0, max_dist=-1.0,
             refine_while_searching=True, of_enable=False, of_max_dist=-1.0, of_min_matches=0,
             device=0, workspace_bytes=0, num_solutions=1, alt_min_3d_difference=0.0,
             alt_rmsd_tolerance=-1.0, refine_all=False, alt_by_pairs=False, alt_max_pairs_common=0):
    if n_max is None or n_max < 0:
        n_max = min(N1, N2)
    return Cfg(N1, N2, num_bb, n_min, n_max, max_rmsd, max_dist, int(refine_while_searching),
               int(of_enable), of_max_dist, of_min_matches, device, 0, workspace_bytes,
               num_solutions, alt_min_3d_difference, alt_rmsd_tolerance, int(refine_all),
               int(alt_by_pairs), alt_max_pairs_common)


def search_noalloc(cfg, d_xyzA, d_xyzB, d_og, n_og, d_ssd, d_xforms, d_pairs, d_winner, stream=0):
    """minrms_search_noalloc: every pointer is a DEVICE address (int), stream a hipStream_t (int, 0 = default).
    Only enqueues work; synchronise the stream before reading the outputs."""
    _check(load_library().minrms_search_noalloc(C.byref(cfg), d_xyzA, d_xyzB, d_og, n_og, d_ssd, d_xforms,
                                                d_pairs, d_winner, stream))


def noalloc_prepare(device=0):
    """Set-up of minrms_search_noalloc on one device (address of its static array, code objects): call it
    once, outside stream capture, before the first search_noalloc on that device."""
    _check(load_library().minrms_noalloc_prepare(device))


def cells_per_orientation(cfg):
    return int(load_library().minrms_engine_cells_per_orientation(C.byref(cfg)))


class Engine:
    """One engine = one SearchOrDyn3d-equivalent object bound to one GPU."""

    def __init__(self, cfg, xyzA, xyzB):
        self._lib = load_library()
        self.cfg = cfg
        self.range_n = cfg.n_max - cfg.n_min + 1
        xyzA = np.ascontiguousarray(xyzA, np.float32)
        xyzB = np.ascontiguousarray(xyzB, np.float32)
        if xyzA.size != cfg.N1 * cfg.num_bb * 3 or xyzB.size != cfg.N2 * cfg.num_bb * 3:
            raise ValueError("coordinate arrays do not match N1/N2/num_bb")
        h = C.c_void_p()
        _check(self._lib.minrms_engine_create(C.byref(cfg), _fp(xyzA), _fp(xyzB), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.minrms_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reset(self):
        _check(self._lib.minrms_engine_reset(self._h))

    def gen_orientations(self, frag_len=4, cutoff_rmsd=-1.0, intervals=None, shuffle=True):
        if intervals is None:
            intervals = [[0, self.cfg.N1 - 1, 0, self.cfg.N2 - 1]]
        iv = np.ascontiguousarray(intervals, np.int32).reshape(-1, 4)
        cap = self.cfg.N1 * self.cfg.N2
        out = np.empty((cap, 12), np.float32)
        cnt = C.c_int64(0)
        _check(self._lib.minrms_engine_gen_orientations(
            self._h, frag_len, cutoff_rmsd, iv.shape[0], iv.ctypes.data_as(C.POINTER(C.c_int32)),
            int(shuffle), _fp(out), cap, C.byref(cnt)))
        return out[:cnt.value].copy()

    def search(self, og, start=0, stop=None):
        og = np.ascontiguousarray(og, np.float32).reshape(-1, 12)
        if stop is None:
            stop = og.shape[0]
        _check(self._lib.minrms_engine_search(self._h, _fp(og), og.shape[0], start, stop))

    def count_updates(self, enable=True):
        """Make the next searches keep, per orientation, how many entries of the
        per-N table it improved when it was evaluated (FirstPass's count_num_updates)."""
        _check(self._lib.minrms_engine_count_updates(self._h, 1 if enable else 0))

    def update_counts(self, n):
        """int32[n]: the counts of the n orientations of the last search() range."""
        out = np.zeros(int(n), np.int32)
        _check(self._lib.minrms_engine_get_update_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_int32)), int(n)))
        return out

    def filter_verdicts(self):
        """uint8 per orientation of the last search() range: 1 = passed the -of filter."""
        n = C.c_int64(0)
        _check(self._lib.minrms_engine_filter_verdicts(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, np.uint8)
        _check(self._lib.minrms_engine_filter_verdicts(
            self._h, out.ctypes.data_as(C.POINTER(C.c_uint8)), n.value, None))
        return out

    def set_filter_calls(self, calls):
        _check(self._lib.minrms_engine_set_filter_calls(self._h, int(calls)))

    def get_keys(self):
        k = np.empty(self.range_n, np.uint64)
        _check(self._lib.minrms_engine_get_keys(self._h, k.ctypes.data_as(C.POINTER(C.c_uint64))))
        return k

    def set_keys(self, keys):
        k = np.ascontiguousarray(keys, np.uint64)
        assert k.size == self.range_n
        _check(self._lib.minrms_engine_set_keys(self._h, k.ctypes.data_as(C.POINTER(C.c_uint64))))

    def keys_device_ptr(self):
        p = C.c_void_p()
        _check(self._lib.minrms_engine_keys_device_ptr(self._h, C.byref(p)))
        return p.value

    def alt_candidates(self):
        """(k, o_id, ssd, rmsd, xf[., 12]) of every candidate record this engine kept during search()."""
        n = C.c_int64(0)
        _check(self._lib.minrms_engine_alt_num_candidates(self._h, C.byref(n)))
        k = np.empty(n.value, np.int32)
        o = np.empty(n.value, np.int64)
        ssd, rmsd, xf = np.empty(n.value, np.float32), np.empty(n.value, np.float32), np.empty((n.value, 12), np.float32)
        if n.value:
            _check(self._lib.minrms_engine_alt_get_candidates(
                self._h, k.ctypes.data_as(C.POINTER(C.c_int32)), o.ctypes.data_as(C.POINTER(C.c_int64)),
                _fp(ssd), _fp(rmsd), _fp(xf)))
        return k, o, ssd, rmsd, xf

    def alt_set_candidates(self, k, o_id, ssd, rmsd, xf):
        """The union of all engines' candidate records (any order)."""
        k = np.ascontiguousarray(k, np.int32)
        o = np.ascontiguousarray(o_id, np.int64)
        ssd, rmsd = np.ascontiguousarray(ssd, np.float32), np.ascontiguousarray(rmsd, np.float32)
        xf = np.ascontiguousarray(xf, np.float32).reshape(-1, 12)
        _check(self._lib.minrms_engine_alt_set_candidates(
            self._h, k.size, k.ctypes.data_as(C.POINTER(C.c_int32)), o.ctypes.data_as(C.POINTER(C.c_int64)),
            _fp(ssd), _fp(rmsd), _fp(xf)))

    def set_collectives(self, allreduce_min=None, allgather=None):
        """Alternates by matched pairs with the search split over several engines
        (minrms_engine_set_collectives).  allreduce_min(uint64 ndarray) -> the
        element-wise minimum over all engines; allgather(int64 ndarray) -> the
        concatenation of all engines' arrays.  None, None removes them."""
        if allreduce_min is None and allgather is None:
            self._callbacks = None
            _check(self._lib.minrms_engine_set_collectives(self._h, ALLREDUCE_MIN_FN(0), ALLGATHER_I64_FN(0), None))
            return

        def c_allreduce(user, keys, n):
            try:
                view = np.ctypeslib.as_array(keys, shape=(n,))
                view[:] = np.asarray(allreduce_min(view.copy()), np.uint64)
                return 0
            except Exception:                     # an exception must not travel through the C frames
                import traceback
                traceback.print_exc()
                return 1

        def c_allgather(user, mine, n_mine, out, cap, n_out):
            try:
                own = np.ctypeslib.as_array(mine, shape=(n_mine,)).copy() if n_mine else np.zeros(0, np.int64)
                every = np.asarray(allgather(own), np.int64).reshape(-1)
                if every.size > cap:
                    return 2
                if every.size:
                    np.ctypeslib.as_array(out, shape=(cap,))[:every.size] = every
                n_out[0] = every.size
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        self._callbacks = (ALLREDUCE_MIN_FN(c_allreduce), ALLGATHER_I64_FN(c_allgather))   # kept alive with the engine
        _check(self._lib.minrms_engine_set_collectives(self._h, self._callbacks[0], self._callbacks[1], None))

    def finalize_search(self, allow_no_solution=False):
        return _check(self._lib.minrms_engine_finalize_search(self._h),
                      allow=(E_NOSOLUTION,) if allow_no_solution else ())

    def refine(self, max_iters=-1):
        st = RefineStats()
        _check(self._lib.minrms_engine_refine(self._h, max_iters, C.byref(st)))
        return st

    def results_rank(self, rank):
        """(ssd, xforms, pairs) of rank `rank` (0 = best, 1 = 2nd best, ...)."""
        rn, nm = self.range_n, self.cfg.n_max
        ssd = np.empty(rn, np.float32)
        xf = np.empty((rn, 12), np.float32)
        pairs = np.empty((rn, nm, 2), np.int16)
        _check(self._lib.minrms_engine_results_rank(self._h, rank, _fp(ssd), _fp(xf),
                                                    pairs.ctypes.data_as(C.POINTER(C.c_int16))),
               allow=(E_NOSOLUTION,))
        return ssd, xf, pairs

    def results(self):
        rn, nm = self.range_n, self.cfg.n_max
        ssd = np.empty(rn, np.float32)
        xf = np.empty((rn, 12), np.float32)
        pairs = np.empty((rn, nm, 2), np.int16)
        win = np.empty(rn, np.int64)
        _check(self._lib.minrms_engine_results(self._h, _fp(ssd), _fp(xf),
                                               pairs.ctypes.data_as(C.POINTER(C.c_int16)),
                                               win.ctypes.data_as(C.POINTER(C.c_int64))))
        return ssd, xf, pairs, win

    # kernel classes of kernel_time() (include/minrms_engine.h)
    KC_CIJ, KC_TRACE, KC_REDUCE, KC_COOP_CIJ, KC_COOP, KC_TRACE_WAVE, KC_ROUND, KC_OTHER, KC_DP_BLOCK0 = range(9)

    def enable_timing(self, level=1):
        """0 off; 1 the kernels of the sweep and of the winners' batch; 2 also those of the refinement rounds."""
        _check(self._lib.minrms_engine_enable_timing(self._h, int(level)))

    def kernel_time(self, kernel_class):
        """(GPU ms, launches) of one kernel class since the last call for it."""
        ms, n = C.c_double(0), C.c_int64(0)
        _check(self._lib.minrms_engine_kernel_time(self._h, kernel_class, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def coop_shape(self):
        """dict: how the cooperative kernel works on one table in a refinement round."""
        t, w, r0, t0, g = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
        cells = C.c_int64(0)
        _check(self._lib.minrms_engine_coop_shape(self._h, C.byref(t), C.byref(w), C.byref(cells), C.byref(r0),
                                                  C.byref(t0), C.byref(g)))
        return {"tasks": t.value, "task_width": w.value, "cells_visited": cells.value, "rows_first_block": r0.value,
                "tasks_first_block": t0.value, "workgroups": g.value}

    def num_blocks(self):
        return int(self._lib.minrms_engine_num_blocks(self._h))

    def dp_block_shape(self, block):
        """(rows, panels, panel width) the sweep kernels visit for one layer block."""
        r, p, w = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(self._lib.minrms_engine_dp_block_shape(self._h, block, C.byref(r), C.byref(p), C.byref(w)))
        return r.value, p.value, w.value

    def dp_block_columns(self, block):
        """Columns the sweep kernels compute per row of one layer block."""
        c = C.c_int32(0)
        _check(self._lib.minrms_engine_dp_block_columns(self._h, block, C.byref(c)))
        return c.value

    def dp_block_time(self, block):
        """(ms, launches, orientations, reference cells per orientation) of one layer block."""
        ms, n, o, c = C.c_double(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(self._lib.minrms_engine_dp_block_time(self._h, block, C.byref(ms), C.byref(n),
                                                     C.byref(o), C.byref(c)))
        return ms.value, n.value, o.value, c.value
