#!/usr/bin/env python3
"""Search time of one synthetic pair for several widest-panel settings.

usage: panel_sweep.py n1 n2 pb [pb ...]      (pb = 0: the default policy)
"""
import os
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from minrms_amd import engine as eng          # noqa: E402
from minrms_amd.search_or import SearchOrHip  # noqa: E402
from minrms_amd.synth import synth_pair       # noqa: E402


def main():
    n1, n2 = int(sys.argv[1]), int(sys.argv[2])
    a, b = synth_pair(n1, n2)
    for pb in [int(x) for x in sys.argv[3:]]:
        if pb:
            os.environ["MINRMS_MAX_PB"] = str(pb)
        else:
            os.environ.pop("MINRMS_MAX_PB", None)
        cfg = eng.make_cfg(n1, n2)
        best = 1e9
        shapes = []
        for rep in range(2):
            so = SearchOrHip(cfg, a, b)
            og = so.generate_orientations(frag_len=4)
            if not shapes:
                blk = 0
                while True:
                    try:
                        shapes.append(so.engine.dp_block_shape(blk))
                    except Exception:
                        break
                    blk += 1
            t0 = time.perf_counter()
            so.Search(og)
            best = min(best, time.perf_counter() - t0)
            so.close()
        print(f"{n1}x{n2} max_pb={pb or 'default'} search {1e3 * best:.1f} ms  blocks (rows, panels, width): {shapes}",
              flush=True)


if __name__ == "__main__":
    main()
