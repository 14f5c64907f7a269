"""Wall time of each phase of one job (orientation generation, sweep, winners, refinement)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from minrms_amd import engine as eng
from minrms_amd.synth import synth_pair

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n2 = int(sys.argv[2]) if len(sys.argv) > 2 else n1
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # only the first K orientations of the shuffled list
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
a, b = synth_pair(n1, n2)
e = eng.Engine(eng.make_cfg(n1, n2), a, b)
for rep in range(reps):
    e.reset()
    e.enable_timing(True)
    for blk in range(e.num_blocks()):
        e.dp_block_time(blk)
    t0 = time.perf_counter(); og = e.gen_orientations(); t1 = time.perf_counter()
    if limit:
        og = og[:limit]
    e.search(og); t2 = time.perf_counter()
    bt = [e.dp_block_time(blk) for blk in range(e.num_blocks())]
    e.finalize_search(); t3 = time.perf_counter()
    st = e.refine(); t4 = time.perf_counter()
    e.results(); t5 = time.perf_counter()
    print("rep %d: gen %.1f ms | search %.1f ms (dp blocks: %s) | winners %.1f ms | refine %.1f ms (%d iters) | results %.1f ms"
          % (rep, (t1-t0)*1e3, (t2-t1)*1e3, ", ".join("%.2f" % x[0] for x in bt), (t3-t2)*1e3, (t4-t3)*1e3,
             st.num_iterations, (t5-t4)*1e3), flush=True)
cells = eng.cells_per_orientation(e.cfg) * og.shape[0]
print("search: %d orientations, %.3e cells" % (og.shape[0], cells))
