"""Writes job files for minrms_amd/host/minrms_host_job (the C++ host class as a
program) and reads its result files (test infrastructure)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_JOB = os.path.join(ROOT, "minrms_amd", "host", "minrms_host_job")


def write_job(path, g, count_updates=False, start=0, stop=None, **over):
    """g: a golden (golden_util.load); over: header fields to overwrite."""
    og = np.ascontiguousarray(g["og"], np.float32).reshape(-1, 12)
    S = int(g["num_solutions"]) if "num_solutions" in g else 1
    h = dict(N1=int(g["N1"]), N2=int(g["N2"]), nbb=int(g["nbb"]), n_min=int(g["n_min"]), n_max=int(g["n_max"]),
             rws=int(g["rws"]), of=int(g["of"]), of_min=int(g["of_min"]), S=S,
             alt_method=int(g["alt_method"]) if "alt_method" in g else 0,
             alt_pairs=int(g["alt_pairs"]) if "alt_pairs" in g else 0,
             refine=int(g["refine_method"]), n_og=og.shape[0], count=int(count_updates), start=start,
             stop=og.shape[0] if stop is None else stop)
    h.update(over)
    x = [float(g["max_rmsd"]), float(g["max_dist"]), float(g["of_dist"]),
         float(g["alt_min_3d"]) if "alt_min_3d" in g else 0.0, float(g["alt_tol"]) if "alt_tol" in g else -1.0, 0.0]
    with open(path, "wb") as f:
        f.write(np.array(list(h.values()), np.int32).tobytes())
        f.write(np.array(x, np.float32).tobytes())
        for a in (g["xyza"], g["xyzb"], og):
            f.write(np.ascontiguousarray(a, np.float32).tobytes())
    return h


def run_job(job, result):
    return subprocess.run([HOST_JOB, str(job), str(result)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=300)


def read_result(path, h):
    """{"search": [(ssd, xf, pairs) per rank], "iters", "new", "refine": [...], "updates"}"""
    buf = open(path, "rb").read()
    rn, n_max, pos = h["n_max"] - h["n_min"] + 1, h["n_max"], 0

    def take(dtype, n):
        nonlocal pos
        a = np.frombuffer(buf, dtype, n, pos)
        pos += a.nbytes
        return a

    def tables():
        return [(take(np.float32, rn), take(np.float32, rn * 12).reshape(rn, 12),
                 take(np.int16, rn * n_max * 2).reshape(rn, n_max, 2)) for _ in range(max(1, h["S"]))]

    out = {"search": tables()}
    if h["refine"]:
        out["iters"], out["new"] = (int(v) for v in take(np.int64, 2))
        out["refine"] = tables()
    if h["count"]:
        out["updates"] = take(np.int32, h["stop"] - h["start"])
    assert pos == len(buf), (pos, len(buf))
    return out
