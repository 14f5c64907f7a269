"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (minrms_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

UPPER_BOUND = np.float32(1.0e26)


class Cfg(C.Structure):
    _fields_ = [("N1", C.c_int32), ("N2", C.c_int32), ("nbb", C.c_int32),
                ("n_min", C.c_int32), ("n_max", C.c_int32),
                ("max_rmsd", C.c_float), ("max_dist", C.c_float),
                ("refine_while_searching", C.c_int32),
                ("of_enable", C.c_int32), ("of_max_dist", C.c_float),
                ("of_min_matches", C.c_int32)]


class AltCfg(C.Structure):
    _fields_ = [("num_solutions", C.c_int32), ("min_3d_difference", C.c_float),
                ("rmsd_tolerance", C.c_float), ("refine_all", C.c_int32),
                ("by_pairs", C.c_int32), ("max_pairs_common", C.c_int32)]


class Tables(C.Structure):
    _fields_ = [("ssd", C.POINTER(C.c_float)), ("xform", C.POINTER(C.c_float)),
                ("pairs", C.POINTER(C.c_int16)), ("winner", C.POINTER(C.c_int64))]


def build():
    """Compile liboracle.so (gcc) if it is missing or stale."""
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("minrms_oracle.c", "minrms_oracle.h", "minrms_oracle_alt.cc")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.orc_jacobi.restype = C.c_int
        L.orc_jacobi.argtypes = [C.c_int, fp, fp, fp]
        L.orc_superimpose.restype = C.c_float
        L.orc_superimpose.argtypes = [C.c_int, fp, fp, fp]
        L.orc_apply_transform.restype = None
        L.orc_apply_transform.argtypes = [fp, fp, C.c_int, fp]
        L.orc_shuffle_perm.restype = None
        L.orc_shuffle_perm.argtypes = [C.c_int64, C.POINTER(C.c_int64)]
        L.orc_gen_orientations.restype = C.c_int64
        L.orc_gen_orientations.argtypes = [C.POINTER(Cfg), fp, fp, C.c_int, C.c_float,
                                           C.c_int, ip, fp, C.c_int64]
        L.orc_dp.restype = None
        L.orc_dp.argtypes = [C.POINTER(Cfg), fp, fp, fp, C.POINTER(C.c_uint8)]
        L.orc_traceback.restype = None
        L.orc_traceback.argtypes = [C.POINTER(Cfg), C.POINTER(C.c_uint8), C.c_int,
                                    C.POINTER(C.c_int16)]
        L.orc_calc_min_rmsd.restype = C.c_float
        L.orc_calc_min_rmsd.argtypes = [C.POINTER(Cfg), fp, fp, C.c_int,
                                        C.POINTER(C.c_int16), fp, fp]
        L.orc_tables_init.restype = None
        L.orc_tables_init.argtypes = [C.POINTER(Cfg), C.POINTER(Tables)]
        L.orc_search.restype = C.c_int64
        L.orc_search.argtypes = [C.POINTER(Cfg), fp, fp, fp, C.c_int64, C.c_int64,
                                 C.POINTER(Tables), fp]
        L.orc_search_filtered.restype = C.c_int64
        L.orc_search_filtered.argtypes = [C.POINTER(Cfg), fp, fp, fp, C.c_int64, C.c_int64,
                                          C.POINTER(Tables), fp, C.POINTER(C.c_uint8), fp]
        L.orc_refine.restype = C.c_int64
        L.orc_refine.argtypes = [C.POINTER(Cfg), fp, fp, C.POINTER(Tables), C.c_int64,
                                 C.POINTER(C.c_int64)]
        L.orc_refine_clip.restype = C.c_int64
        L.orc_refine_clip.argtypes = [C.POINTER(Cfg), fp, fp, C.POINTER(Tables), C.c_int64,
                                      C.POINTER(C.c_int64), C.c_float]
        L.orc_nw_num_matches.restype = C.c_int
        L.orc_nw_num_matches.argtypes = [C.POINTER(Cfg), fp, fp, C.c_float, C.c_float, C.c_int]
        L.orc_clip_after_filter_call.restype = C.c_float
        L.orc_clip_after_filter_call.argtypes = [C.POINTER(Cfg), C.c_float]
        i16 = C.POINTER(C.c_int16)
        L.orc_principal_axes.restype = None
        L.orc_principal_axes.argtypes = [C.c_int, fp, fp, fp]
        L.orc_fast_rot_ssd.restype = C.c_float
        L.orc_fast_rot_ssd.argtypes = [fp, fp, fp, fp, C.c_int]
        L.orc_slow_rot_ssd.restype = C.c_float
        L.orc_slow_rot_ssd.argtypes = [fp, fp, fp, C.c_int]
        L.orc_search_alt.restype = C.c_int
        L.orc_search_alt.argtypes = [C.POINTER(Cfg), C.POINTER(AltCfg), fp, fp, fp, C.c_int64,
                                     fp, fp, i16, fp]
        L.orc_search_alt_pairs.restype = C.c_int
        L.orc_search_alt_pairs.argtypes = L.orc_search_alt.argtypes
        L.orc_refine_alt.restype = C.c_int64
        L.orc_refine_alt.argtypes = [C.POINTER(Cfg), C.POINTER(AltCfg), fp, fp, fp, fp, i16,
                                     C.c_int64, C.POINTER(C.c_int64), C.c_float]
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_cfg(N1, N2, nbb=1, n_min=1, n_max=None, max_rmsd=-1.0, max_dist=-1.0,
             refine_while_searching=True, of_enable=False, of_max_dist=-1.0,
             of_min_matches=0):
    if n_max is None or n_max < 0:
        n_max = min(N1, N2)
    return Cfg(N1, N2, nbb, n_min, n_max, max_rmsd, max_dist,
               int(refine_while_searching), int(of_enable), of_max_dist, of_min_matches)


class ResultTables:
    """numpy-backed orc_tables."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.range_n = cfg.n_max - cfg.n_min + 1
        self.ssd = np.empty(self.range_n, np.float32)
        self.xform = np.empty((self.range_n, 12), np.float32)
        self.pairs = np.empty((self.range_n, cfg.n_max, 2), np.int16)
        self.winner = np.empty(self.range_n, np.int64)
        self.c = Tables(_fp(self.ssd), _fp(self.xform),
                        self.pairs.ctypes.data_as(C.POINTER(C.c_int16)),
                        self.winner.ctypes.data_as(C.POINTER(C.c_int64)))
        lib().orc_tables_init(C.byref(cfg), C.byref(self.c))


def jacobi(a):
    a = np.array(a, np.float32, copy=True)
    n = a.shape[0]
    d = np.zeros(n, np.float32)
    v = np.zeros((n, n), np.float32)
    sweeps = lib().orc_jacobi(n, _fp(a), _fp(d), _fp(v))
    return d, v, sweeps


def superimpose(fixed, moveable):
    fixed = np.ascontiguousarray(fixed, np.float32)
    moveable = np.ascontiguousarray(moveable, np.float32)
    xf = np.zeros(12, np.float32)
    r = lib().orc_superimpose(fixed.shape[0], _fp(fixed), _fp(moveable), _fp(xf))
    return np.float32(r), xf


def apply_transform(xf, xyz):
    xf = np.ascontiguousarray(xf, np.float32).reshape(12)
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.empty_like(xyz)
    lib().orc_apply_transform(_fp(xf), _fp(xyz), xyz.size // 3, _fp(out))
    return out


def shuffle_perm(count):
    perm = np.empty(count, np.int64)
    lib().orc_shuffle_perm(count, perm.ctypes.data_as(C.POINTER(C.c_int64)))
    return perm


def gen_orientations(cfg, xyzA, xyzB, frag_len=4, cutoff_rmsd=-1.0, intervals=None,
                     shuffle=True):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    if intervals is None:
        intervals = [[0, cfg.N1 - 1, 0, cfg.N2 - 1]]
    iv = np.ascontiguousarray(intervals, np.int32).reshape(-1, 4)
    cap = cfg.N1 * cfg.N2
    out = np.empty((cap, 12), np.float32)
    cnt = lib().orc_gen_orientations(C.byref(cfg), _fp(xyzA), _fp(xyzB), frag_len,
                                     cutoff_rmsd, iv.shape[0],
                                     iv.ctypes.data_as(C.POINTER(C.c_int32)), _fp(out), cap)
    out = out[:cnt].copy()
    if shuffle:
        out = out[shuffle_perm(cnt)]
    return out


def dp(cfg, xyzA, xyzB_rot, want_bp=True):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB_rot = np.ascontiguousarray(xyzB_rot, np.float32)
    cost = np.empty(cfg.n_max, np.float32)
    bp = np.empty((cfg.n_max, cfg.N1, cfg.N2), np.uint8) if want_bp else None
    lib().orc_dp(C.byref(cfg), _fp(xyzA), _fp(xyzB_rot), _fp(cost),
                 bp.ctypes.data_as(C.POINTER(C.c_uint8)) if want_bp else None)
    return cost, bp


def traceback(cfg, bp, n):
    pairs = np.empty((n, 2), np.int16)
    lib().orc_traceback(C.byref(cfg), bp.ctypes.data_as(C.POINTER(C.c_uint8)), n,
                        pairs.ctypes.data_as(C.POINTER(C.c_int16)))
    return pairs


def calc_min_rmsd(cfg, xyzA, xyzB, pairs):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    pairs = np.ascontiguousarray(pairs, np.int16)
    xf = np.zeros(12, np.float32)
    ssd = C.c_float(0)
    r = lib().orc_calc_min_rmsd(C.byref(cfg), _fp(xyzA), _fp(xyzB), pairs.shape[0],
                                pairs.ctypes.data_as(C.POINTER(C.c_int16)), _fp(xf),
                                C.byref(ssd))
    return np.float32(r), xf, np.float32(ssd.value)


def search(cfg, xyzA, xyzB, og, start=0, stop=None, tables=None, per_on=False):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    og = np.ascontiguousarray(og, np.float32).reshape(-1, 12)
    if stop is None:
        stop = og.shape[0]
    if tables is None:
        tables = ResultTables(cfg)
    po = np.empty((stop - start, tables.range_n), np.float32) if per_on else None
    flags = np.zeros(stop - start, np.uint8)
    clip = getattr(tables, "clip", None)
    clip = C.c_float(initial_clip(cfg) if clip is None else clip)
    passed = lib().orc_search_filtered(C.byref(cfg), _fp(xyzA), _fp(xyzB), _fp(og), start, stop,
                                       C.byref(tables.c), _fp(po) if per_on else None,
                                       flags.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(clip))
    tables.passed = passed
    tables.per_on = po
    tables.pass_flags = flags
    tables.clip = clip.value        # what Refine will see
    return tables


def initial_clip(cfg):
    """DistanceMetric's cutoff as SetMaxPhysicalDistance leaves it (distance_metric.cc:76-87)."""
    if cfg.max_dist == -1.0:
        return float(np.float32(1.0e26))
    d = np.float32(cfg.max_dist)
    return float(np.float32(d * d) * np.float32(cfg.nbb))


def refine(cfg, xyzA, xyzB, tables, max_iters=-1):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    num_new = C.c_int64(0)
    clip = getattr(tables, "clip", None)
    if clip is None:
        clip = initial_clip(cfg)
    iters = lib().orc_refine_clip(C.byref(cfg), _fp(xyzA), _fp(xyzB), C.byref(tables.c),
                                  max_iters, C.byref(num_new), clip)
    return iters, num_new.value


# ---- alternate solutions by orientation (-alt-o) -------------------------------
class AltTables:
    """ssd[S][range_n], xform[S][range_n][12], pairs[S][range_n][n_max][2]."""

    def __init__(self, cfg, num_solutions):
        rn = cfg.n_max - cfg.n_min + 1
        self.ssd = np.empty((num_solutions, rn), np.float32)
        self.xform = np.empty((num_solutions, rn, 12), np.float32)
        self.pairs = np.empty((num_solutions, rn, cfg.n_max, 2), np.int16)
        self.clip = initial_clip(cfg)


def principal_axes(xyz):
    xyz = np.ascontiguousarray(xyz, np.float32)
    eig, axes = np.zeros(3, np.float32), np.zeros(12, np.float32)
    lib().orc_principal_axes(xyz.size // 3, _fp(xyz), _fp(eig), _fp(axes))
    return eig, axes


def search_alt(cfg, alt, xyzA, xyzB, og):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    og = np.ascontiguousarray(og, np.float32).reshape(-1, 12)
    t = AltTables(cfg, alt.num_solutions)
    clip = C.c_float(t.clip)
    fn = lib().orc_search_alt_pairs if alt.by_pairs else lib().orc_search_alt
    rc = fn(C.byref(cfg), C.byref(alt), _fp(xyzA), _fp(xyzB), _fp(og), og.shape[0],
                              _fp(t.ssd), _fp(t.xform), t.pairs.ctypes.data_as(C.POINTER(C.c_int16)),
                              C.byref(clip))
    t.clip = clip.value
    t.found = rc == 0
    return t


def refine_alt(cfg, alt, xyzA, xyzB, t, max_iters=-1):
    xyzA = np.ascontiguousarray(xyzA, np.float32)
    xyzB = np.ascontiguousarray(xyzB, np.float32)
    num_new = C.c_int64(0)
    iters = lib().orc_refine_alt(C.byref(cfg), C.byref(alt), _fp(xyzA), _fp(xyzB), _fp(t.ssd),
                                 _fp(t.xform), t.pairs.ctypes.data_as(C.POINTER(C.c_int16)),
                                 max_iters, C.byref(num_new), t.clip)
    return iters, num_new.value
