"""When each task of one cooperative DP launch starts, reaches row 63 and ends
(diagnostic build of the engine with -DMINRMS_COOP_STAMPS; tools only).
usage: python tools/coop_stamps.py build [-D...]        (here: builds tools/ubench/libminrms_hip_stamps.so)
       python tools/coop_stamps.py [N1 [N2 [orientations]]]      (on the GPU box)
The library is listed in .gpurunignore (no test or bench loads it): take that
line out for the runs that use this tool.  MINRMS_TRACE_ALL=1 prints every
chain of the wave tracer."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minrms_amd import build, engine as eng  # noqa: E402

LIB = os.path.join(ROOT, "tools", "ubench", "libminrms_hip_stamps.so")


def build_stamps_lib(extra=(), out=LIB):
    build.build_engine(force=True, extra_flags=["-DMINRMS_COOP_STAMPS"] + list(extra), lib=out, guard=False)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        # extra arguments: -D switches of an experiment, e.g. build -DMINRMS_EXP_NOBP -> ..._stamps_NOBP.so
        extra = sys.argv[2:]
        tag = "".join("_" + x.split("MINRMS_EXP_")[-1] for x in extra)
        build_stamps_lib(extra, LIB.replace(".so", tag + ".so"))
        sys.exit(0)
    if os.environ.get("STAMPS_LIB_TAG"):
        LIB = LIB.replace(".so", "_" + os.environ["STAMPS_LIB_TAG"] + ".so")
    import numpy as np
    from minrms_amd.synth import synth_pair
    n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    n2 = int(sys.argv[2]) if len(sys.argv) > 2 else n1
    cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    eng.LIB_PATH = LIB
    lib = eng.load_library()
    a, b = synth_pair(n1, n2)
    with eng.Engine(eng.make_cfg(n1, n2), a, b) as e:
        og = e.gen_orientations()
        for rep in range(3):
            e.reset()
            e.search(og[:cnt])
            ts = (C.c_ulonglong * 8)()
            lib.minrms_engine_debug_read_trace_stamps(e._h, ts)
            print("rep %d: tracer, longest chain: %d cycles = %d before the refit (%d fetches, %d of them ahead, took %d, %d walk steps %d) + refit %d"
                  % (rep, ts[0], ts[6], ts[4] & 0xffffffff, ts[4] >> 32, ts[1], ts[5], ts[2], ts[0] - ts[6])
                  + " (pass 1 %d, pass 2 %d, eigen-solve and transform %d)" % (ts[3] >> 32, ts[3] & 0xffffffff, ts[7]))
            out = (C.c_int * 64)()
            lib.minrms_engine_debug_read_err(e._h, out)
            st = np.array(out[8:62], np.int64).reshape(9, 6)
            print("rep %d: cycles of the row loop = wait/inject + cells + hand-over + progress; where the wave ran" % rep)
            for t in range(9):
                if st[t, 1] == 0:
                    continue
                hw = int(st[t, 0]) & 0xffffffff
                print("  task %2d: %8d = %7d + %7d + %7d + %7d   simd %d cu %2d se %d" % (
                    t, st[t, 1], st[t, 2], st[t, 3], st[t, 4], st[t, 5], (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7))

        # time line of the last launch: every task and helper wave of orientation 0
        tm = (C.c_uint * 1536)()
        lib.minrms_engine_debug_read_coop_times(e._h, tm)
        tm = np.array(tm[:], np.int64).reshape(384, 4)
        t0 = min(int(x) for x in tm[:, 0] if x)
        print("time line (x 10 ns after the first wave entered): entered, rows begun, rows ended")
        for t in range(384):
            if tm[t, 0] == 0:
                continue
            name = "task %2d" % t if t < 128 else "group %d helper %d" % ((t - 128) // 4, (t - 128) % 4)
            sc = int(tm[t, 3])
            if t >= 128:
                if tm[t, 1]:
                    print("  %-18s %6d        %6d   fetch wave: %d polls, %d of them empty; waited %d for loads, %d for room in the ring" % (
                        name, tm[t, 0] - t0, tm[t, 2] - t0, tm[t, 1] & 0xffff, tm[t, 1] >> 16, sc & 0xffff, sc >> 16))
                else:
                    print("  %-18s %6d        %6d" % (name, tm[t, 0] - t0, (tm[t, 2] - t0) if tm[t, 2] else -1))
                continue
            print("  %-18s %6d %6d %6d   waits off the straight path: left %3d plane %3d right %3d publish %3d" % (
                name, tm[t, 0] - t0, (tm[t, 1] - t0) if tm[t, 1] else -1, (tm[t, 2] - t0) if tm[t, 2] else -1,
                sc & 255, (sc >> 8) & 255, (sc >> 16) & 255, sc >> 24))

        ft = (C.c_uint * 1536)()
        lib.minrms_engine_debug_read_fetch_trace(e._h, ft)
        ft = np.array(ft[:], np.int64).reshape(64, 8, 3)
        print("edge-fetch waves: (time, rows handed on, rows the consumer has finished) every 32 rows")
        for g in range(64):
            if ft[g, 0, 0]:
                print("  group %2d: " % g + "  ".join("%5d %3d %3d" % (ft[g, k, 0] - t0, ft[g, k, 1], ft[g, k, 2])
                                                     for k in range(8) if ft[g, k, 0]))

        # the round kernel of the refinement: cycles per launch of its three parts (printed by the library)
        e.reset()
        e.search(og)
        e.finalize_search()
        e.refine()
        ts = (C.c_ulonglong * 8)()
        lib.minrms_engine_debug_read_trace_stamps(e._h, ts)
