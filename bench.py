#!/usr/bin/env python3
"""Benchmark of the MINRMS RMSD-vs-N search on MI355X.

One step = the complete job for one structure pair: fragment-pair orientation
generation, the exhaustive orientation sweep (3-index DP + traceback + refit per
orientation and N), the per-N minimum, and iterative refinement -- i.e. what
`minrms -fm 4 -HS -ir -r A.pdb B.pdb` computes between reading the PDB files and
writing the MSF files.  Inputs (coordinates) are resident in HBM before the timed
region.  With N GPUs the orientation list is cut into N contiguous o_id ranges,
one per rank, the per-N keys are all-reduced with MIN over RCCL, and every rank
finishes with the full tables (strong scaling: the job is fixed).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the program starts its N ranks
itself (one child process per GPU; the parent touches no GPU); under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is
one of the ranks.  Fewer visible GPUs than N, a WORLD_SIZE that is not N, or a
rank that fails end the run with a non-zero exit code and no JSON line.

Metric: reference DP cells per second (one cell = one (n,i,j) update of
Dyn3d::Solve_IncrN / InitTable_IncrN, lib/dyn/dyn3d.cc:135-387 of the
reference): (orientations searched + DP tables evaluated by refinement) x cells
per orientation / wall time.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_CELL = 9.0           # 4 B cost read + 4 B cost write + 1 B back-pointer
# What bounds the DP kernels is the issue of vector instructions: 7 per cell and
# wave (add, min with DPP shift, 2 x compare + add-with-carry for the two tie
# flags, min).  One wave64 instruction occupies a SIMD's vector issue for 4
# cycles: measured with tools/ubench/dp_cell_ubench (profiles/r02_ubench_dp_cell.txt:
# 3.6-3.8 cycles per instruction and SIMD for the cell sequence and 4.4-4.7 for
# independent v_add_f32 / v_fma_f32 at 2-5 waves per SIMD; two waves on one SIMD
# each take 2 x the cycles of a wave alone).  256 CUs x 4 SIMDs at 2.4 GHz.
VALU_PER_CELL, CYCLES_PER_VALU, N_SIMD, CLOCK_HZ = 7, 4, 1024, 2.4e9
PMC_FILE = "r03_d_pmc_dp_block0.json"      # FETCH_SIZE / WRITE_SIZE passes (tools/profile_round.sh)
PEAK_ISSUE = N_SIMD * CLOCK_HZ / CYCLES_PER_VALU          # wave instructions per second, whole GPU
PEAK_BASIS = ("1 wave64 vector instruction per %d cycles per SIMD (measured, profiles/r02_ubench_dp_cell.txt), "
              "%d SIMDs, %.1f GHz" % (CYCLES_PER_VALU, N_SIMD, CLOCK_HZ / 1e9))


def host_cpu():
    """(model name, cores of this process's CPU share)."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    # a GPU box hands each job a share of the host; its size is exported as the thread count of the usual pools
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            cores = min(cores, int(v))
            break
    return model, cores


def _ref_search(exe, tmp, start, count):
    """One reference process: SearchOrDyn3d::Search over o_ids [start, start + count)."""
    env = dict(os.environ, REF_DUMP_BENCH=str(count), REF_DUMP_START=str(start),
               REF_DUMP_OUT=os.path.join(tmp, "d%d.txt" % start))
    return subprocess.Popen([exe, "-fm", "4", "-HS", "-ir", "-r", "a.pdb", "b.pdb"], cwd=tmp, env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _ref_result(proc):
    out = proc.communicate()[0]
    for line in out.splitlines():
        if line.startswith("REF_BENCH"):
            t = line.split()
            return int(t[2]), int(t[4]), float(t[6])
    return None


def cpu_baseline(n1, n2, cells_per_o, seconds_target=12.0):
    """Reference CPU path on this host's cores, bounded sample (rank 0 only).

    Row 1: the reference as it is, one thread.  Row 2: all cores of this job's
    CPU share, the only way the reference offers: one process per core, each
    with its own o_id range through Search(og, start, stop)
    (bin/minrms/search_or.h:83-103); wall time = the slowest process."""
    from minrms_amd.synth import synth_pdb_text
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_dump_fast")
    model, cores = host_cpu()
    est_rate = 5.0e8          # reference on one core of the GPU box host (measured)
    k = max(8, int(seconds_target * est_rate / cells_per_o))
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as tmp:
            open(os.path.join(tmp, "a.pdb"), "w").write(synth_pdb_text(n1, 1234))
            open(os.path.join(tmp, "b.pdb"), "w").write(synth_pdb_text(n2, 5678))
            one = _ref_result(_ref_search(exe, tmp, 0, k))
            if one:
                ko, n_og, sec = one
                base = {"value": ko * cells_per_o / sec, "unit": "DP-cells/s", "cores": 1, "kind": "reference",
                        "host_cpu_model": model, "host_cores": cores,
                        "sample": "reference SearchOrDyn3d::Search (its own -O3 -ffast-math recipe) "
                                  "over the first %d of %d shuffled orientations of the same pair, "
                                  "%.1f s, 1 thread" % (ko, n_og, sec)}
                # all cores: one range of k orientations per process
                per = max(1, min(k, n_og // max(1, cores)))
                t0 = time.time()
                procs = [_ref_search(exe, tmp, q * per, per) for q in range(cores)]
                res = [_ref_result(p) for p in procs]
                wall = time.time() - t0
                if all(res):
                    slowest = max(r[2] for r in res)
                    base["all_cores"] = {
                        "value": cores * per * cells_per_o / slowest, "unit": "DP-cells/s", "cores": cores,
                        "sample": "%d reference processes, one o_id range of %d orientations each "
                                  "(Search(og, start, stop)), slowest %.1f s (wall %.1f s with start-up)"
                                  % (cores, per, slowest, wall)}
                return base
    # fall back to our own C restatement of the same loop
    from oracle import oracle_py as orc
    from minrms_amd.synth import synth_pair
    a, b = synth_pair(n1, n2)
    cfg = orc.make_cfg(n1, n2)
    og = orc.gen_orientations(cfg, a, b)
    k = min(k, og.shape[0])
    t0 = time.time()
    orc.search(cfg, a, b, og, 0, k)
    sec = time.time() - t0
    return {"value": k * cells_per_o / sec, "unit": "DP-cells/s", "cores": 1, "kind": "port",
            "host_cpu_model": model, "host_cores": cores,
            "sample": "C restatement (oracle/) of the search loop over the first %d of %d shuffled "
                      "orientations of the same pair, %.1f s, 1 thread" % (k, og.shape[0], sec)}


def run_job(so, refine=True):
    """One step: the complete job through the SearchOr-shaped interface.
    Returns (orientations, DP tables refinement evaluated, refinement iterations).
    Host wall time per phase is added to so.phase_s (every phase ends with the
    engine's stream drained, so the sum is the step)."""
    ph = so.phase_s
    t0 = time.perf_counter()
    so.reset()
    og = so.generate_orientations()
    t1 = time.perf_counter()
    ph["generate"] = ph.get("generate", 0.0) + (t1 - t0)
    so.Search(og)                 # this rank's o_id range + MIN all-reduce of the keys + winners (timed inside)
    t2 = time.perf_counter()
    if not refine:
        return og.shape[0], 0, 1
    st = so.Refine()
    so.tables()
    ph["refine"] = ph.get("refine", 0.0) + (time.perf_counter() - t2)
    return og.shape[0], st.dp_runs, st.num_iterations


# ---------------------------------------------------------------------------
# launcher: --gpus N without a rendezvous in the environment
# ---------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, backend):
    """Start n ranks of this program (fresh processes; this one touches no GPU),
    wait for all; a rank that fails ends the others.  Returns the exit code."""
    if backend == "nccl":
        import torch
        visible = torch.cuda.device_count()          # counting devices does not initialise the GPU
        if visible < n:
            print("bench.py: --gpus %d but only %d GPU(s) visible" % (n, visible), file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    pending = list(procs)
    while pending:
        time.sleep(0.05)
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in pending:                    # the exact processes started above
                    q.terminate()
    if rc != 0:
        print("bench.py: a rank failed (exit code %d): no result" % rc, file=sys.stderr, flush=True)
    return rc


# ---------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------
def _kernel_table(engine, steps):
    """Per kernel class: GPU ms per step and launches per step (HIP events on the engine's stream)."""
    E = type(engine)
    names = [("k_cij (sweep)", E.KC_CIJ), ("k_trace_fit (sweep)", E.KC_TRACE), ("k_reduce", E.KC_REDUCE),
             ("k_cij (small batches)", E.KC_COOP_CIJ), ("dp::k_dp_coop", E.KC_COOP),
             ("k_trace_fit_wave", E.KC_TRACE_WAVE), ("k_refine_round", E.KC_ROUND), ("other", E.KC_OTHER)]
    out = {}
    for name, cls in names:
        ms, n = engine.kernel_time(cls)
        if n:
            out[name] = {"ms_per_step": ms / steps, "launches_per_step": n / steps, "avg_launch_ms": ms / n}
    return out


def rank_main(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    gpu = args.backend == "nccl"
    # with --group-of-one a single rank forms the process group too and runs every collective
    # of the N > 1 path (a rehearsal of that path over RCCL on a one-GPU box)
    grouped = world > 1 or args.group_of_one
    if grouped and "MASTER_ADDR" not in os.environ:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1")
    if gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit("bench.py: rank %d has no GPU %d" % (rank, local_rank))
        torch.cuda.set_device(local_rank)
    if grouped:
        if gpu:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip
    from minrms_amd.synth import synth_pair

    def make_search(n1, n2):
        xyzA, xyzB = synth_pair(n1, n2)
        cfg = eng.make_cfg(n1, n2, device=local_rank if gpu else 0)
        kw = {}
        if args.engine_factory:
            # Test hook: the CPU suite drives the launcher and the rank logic with a
            # stand-in engine ("module:function"(cfg, xyzA, xyzB) -> object with
            # Engine's methods) over gloo.  The product path is the HIP engine.
            mod, fn = args.engine_factory.split(":")
            kw["engine"] = getattr(importlib.import_module(mod), fn)(cfg, xyzA, xyzB)
        if not gpu and grouped:
            def all_reduce_min(keys_i64):
                t = torch.from_numpy(np.array(keys_i64, copy=True))
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return t.numpy()
            kw["all_reduce_min"] = all_reduce_min
        return cfg, SearchOrHip(cfg, xyzA, xyzB, rank=rank, world_size=world, always_reduce=grouped, **kw)

    def sync():
        if gpu:
            torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    def reduce_max(x):
        if not grouped:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ranks that take part: a sum of ones over the process group, not an environment variable
    n_ranks = 1
    if grouped:
        t = torch.ones(1, dtype=torch.int64, device="cuda" if gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_ranks = int(t.item())

    cfg, so = make_search(args.n1, args.n2)
    cells_per_o = eng.cells_per_orientation(cfg)
    timing = gpu and not args.engine_factory

    for _ in range(args.warmup):
        run_job(so)
    so.phase_s.clear()
    sync()
    t0 = time.perf_counter()
    n_or = n_ref = iters = 0
    for _ in range(args.steps):
        o, r, it = run_job(so)
        n_or += o
        n_ref += r
        iters = it
    sync()
    dt = reduce_max(time.perf_counter() - t0)
    phases_ms = {k: reduce_max(v) / args.steps * 1e3 for k, v in sorted(so.phase_s.items())}
    for k in ("generate", "sweep", "all_reduce", "winners", "refine"):
        phases_ms.setdefault(k, 0.0)

    roofline = None
    if timing:
        roofline = measure_roofline(args, so, world, cells_per_o, sync)

    extra = None
    if (grouped or args.extra_sweep) and args.extra_n > 0:
        # the configuration that is sharded over GPUs in BASELINE.json: the sweep of the 400x400 pair
        so.close()
        cfg4, so4 = make_search(args.extra_n, args.extra_n)
        run_job(so4, refine=False)
        so4.phase_s.clear()
        sync()
        t0 = time.perf_counter()
        reps, n_or4 = 2, 0
        for _ in range(reps):
            n_or4 += run_job(so4, refine=False)[0]
        sync()
        dt4 = reduce_max(time.perf_counter() - t0)
        cells4 = eng.cells_per_orientation(cfg4)
        extra = {"workload": "synthetic CA-only %dx%d pair (seeds 1234/5678), -fm 4 -HS -ir: orientations, sweep, "
                             "MIN all-reduce, winners; no refinement" % (args.extra_n, args.extra_n),
                 "orientations": n_or4 // reps, "cells_per_orientation": cells4, "steps": reps,
                 "ms_per_step": dt4 / reps * 1e3, "value": n_or4 * cells4 / dt4, "unit": "DP-cells/s",
                 "phases_ms": {k: reduce_max(v) / reps * 1e3 for k, v in sorted(so4.phase_s.items())}}
        so4.close()

    if rank == 0:
        # work = the DP tables the REFERENCE evaluates for this job: one per
        # orientation and one per refinement iteration (num_iterations counts
        # from 1).  Tables the engine evaluates speculatively are not counted.
        total_cells = (n_or + (iters - 1) * args.steps) * cells_per_o
        out = {
            "metric": "DP-cells/s, full RMSD-vs-N curve (search + refine), %dx%d pair" % (args.n1, args.n2),
            "value": total_cells / dt, "unit": "DP-cells/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic CA-only %dx%d pair (seeds 1234/5678), -fm 4 -HS -ir -r: "
                                   "full orientation sweep, n_min=1" % (args.n1, args.n2),
                       "orientations": n_or // args.steps, "refine_dp_runs_executed": n_ref // args.steps,
                       "refine_iterations": iters, "cells_per_orientation": cells_per_o,
                       "parallelism": "o_id range per GPU + min-all-reduce" if world > 1 else "1 GPU"},
            # host wall time per phase of a step, maximum over the ranks; only the sweep is cut over
            # the ranks, so T(N) ~ generate + sweep(1)/N + all_reduce + winners + refine
            "phases_ms": phases_ms,
            "roofline": roofline,
        }
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1 and gpu:
            out["cpu_baseline"] = cpu_baseline(args.n1, args.n2, cells_per_o)
        print(json.dumps(out), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


def measure_roofline(args, so, world, cells_per_o, sync):
    """Dominant kernel = the kernel with the most GPU time per step among ALL
    kernels of the job, each bracketed by HIP events on the stream it runs on.
    The brackets are taken in extra steps right after the timed region, not in
    it: the timed sweep runs as two batches on two streams whose launches overlap
    (a kernel's own duration cannot be told from events then), and an event
    between the short dependent launches of a refinement round lengthens the
    chain the step is made of.  The instrumented steps run the sweep on one
    stream; their wall time is reported beside the timed one."""
    e = so.engine
    steps = max(1, min(3, args.steps))
    os.environ["MINRMS_SWEEP_STREAMS"] = "1"
    e.enable_timing(2)
    for b in range(e.num_blocks()):
        e.dp_block_time(b)                                   # drain
    _kernel_table(e, 1)
    sync()
    t0 = time.perf_counter()
    tables = rounds = 0
    for _ in range(steps):
        so.reset()
        og = so.generate_orientations()
        so.Search(og)
        st = so.Refine()
        tables += st.dp_runs
        rounds += st.rounds
    sync()
    instrumented_ms = (time.perf_counter() - t0) / steps * 1e3
    e.enable_timing(0)
    del os.environ["MINRMS_SWEEP_STREAMS"]
    measured_in = ("%d instrumented steps after the timed region (sweep on one stream, an event in front of every "
                   "launch: %.2f ms per step)" % (steps, instrumented_ms))
    blocks = []
    for b in range(e.num_blocks()):
        ms, launches, orients, cells_b = e.dp_block_time(b)
        rows, panels, width = e.dp_block_shape(b)
        blocks.append({"block": b, "ms": ms, "launches": launches, "orientations": orients,
                       "cells_per_orientation": cells_b, "rows": rows, "panels": panels, "panel_width": width,
                       "columns": e.dp_block_columns(b)})
    kernels = _kernel_table(e, steps)
    for blk in blocks:
        if blk["launches"]:
            kernels["dp::k_dp_panel (layer block %d)" % blk["block"]] = {
                "ms_per_step": blk["ms"] / steps, "launches_per_step": blk["launches"] / steps,
                "avg_launch_ms": blk["ms"] / blk["launches"]}
    if not kernels:
        return None
    dominant = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
    extra_steps = steps

    # ---- the sweep's DP, layer block with the most time
    sweep = None
    dom = max(blocks, key=lambda x: x["ms"]) if blocks else None
    if dom and dom["ms"] > 0:
        launches = max(1, dom["launches"])
        sec = dom["ms"] * 1e-3
        traffic = None
        pmc = os.path.join(ROOT, "profiles", PMC_FILE)
        if os.path.exists(pmc) and (args.n1, args.n2, world) == (150, 150, 1) and dom["block"] == 0:
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
        # wave instructions the launches execute: every wave (= orientation) visits
        # rows x columns cells with 7 vector instructions each
        wave_instr = dom["orientations"] * dom["rows"] * dom["columns"] * VALU_PER_CELL
        achieved = wave_instr / sec
        alg_bytes = BYTES_PER_CELL * dom["cells_per_orientation"] * dom["orientations"]
        sweep = {
            "bound": "valu_issue", "achieved": achieved / 1e9, "peak": PEAK_ISSUE / 1e9, "unit": "G wave-instr/s",
            "frac": achieved / PEAK_ISSUE, "peak_basis": PEAK_BASIS,
            # the same launches against the 2 cycles the hardware guide lists for >= 2 waves per SIMD
            "frac_if_2_cycles_per_instr": achieved / (N_SIMD * CLOCK_HZ / 2),
            "kernel": "dp::k_dp_panel (layer block %d)" % dom["block"],
            "launches": dom["launches"], "avg_launch_ms": dom["ms"] / launches,
            # HBM bytes per launch: FETCH_SIZE + WRITE_SIZE of the committed PMC passes, NOT measured in this run
            "traffic": traffic,
            "traffic_source": ("profiles/%s (separate rocprofv3 --pmc passes of the same workload), "
                               "not measured in this run" % PMC_FILE) if traffic else None,
            "hbm_frac_measured": (traffic / (sec / launches) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            # the 9 B/cell model of the reference's own table traffic (4 B cost read + 4 B cost
            # write + 1 B back-pointer per cell): not a bound here, the cost layers never leave registers
            "algorithmic_hbm": {"bytes_per_launch": alg_bytes / launches, "achieved_GBs": alg_bytes / sec / 1e9,
                                "peak_GBs": HBM_PEAK_GBS, "frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBS},
            "executed_cells_over_reference_cells": (dom["rows"] * dom["columns"] * 63.0) /
                                                   max(1, dom["cells_per_orientation"]),
            "blocks": blocks}

    # ---- the refinement's DP: a chain of short launches, a few tables each
    refine_dp = None
    coop = kernels.get("dp::k_dp_coop")
    if coop and tables:
        shape = e.coop_shape()
        n_launch = coop["launches_per_step"] * extra_steps
        sec = coop["ms_per_step"] * extra_steps * 1e-3
        wave_instr = tables * shape["cells_visited"] * VALU_PER_CELL      # every cell by the 64 lanes of one wave
        achieved = wave_instr / sec
        us_launch = sec / n_launch * 1e6
        refine_dp = {
            "bound": "valu_issue", "achieved": achieved / 1e9, "peak": PEAK_ISSUE / 1e9, "unit": "G wave-instr/s",
            "frac": achieved / PEAK_ISSUE, "peak_basis": PEAK_BASIS, "kernel": "dp::k_dp_coop (refinement rounds)",
            "launches": int(n_launch), "avg_launch_ms": us_launch * 1e-3,
            "tables_per_launch": tables / max(1, rounds), "rounds_per_step": rounds / extra_steps,
            # what a launch is: the tasks of a table are a pipeline of waves, one SIMD each; the first
            # layer block's first task walks its rows_first_block rows, the last task follows tasks_first_block - 1
            # hand-overs later
            "latency_chain": dict(shape, us_per_launch=us_launch,
                                  cycles_per_row_of_the_chain=us_launch * 1e-6 * CLOCK_HZ / max(1, shape["rows_first_block"])),
            "cus_holding_a_wave_frac": min(1.0, tables / max(1, rounds) * shape["workgroups"] / 256.0),
            "traffic": None}
        pmc = os.path.join(ROOT, "profiles", PMC_FILE)
        if os.path.exists(pmc) and (args.n1, args.n2, world) == (150, 150, 1):
            c = json.load(open(pmc)).get("refinement_dp_kernel")
            if c:
                refine_dp["traffic"] = c["hbm_bytes_per_launch"]
                refine_dp["traffic_source"] = ("profiles/%s (separate rocprofv3 --pmc passes of the same workload, average over "
                                               "the launches of a step), not measured in this run" % PMC_FILE)
                refine_dp["hbm_frac_measured"] = c["hbm_bytes_per_launch"] / (us_launch * 1e-6) / 1e9 / HBM_PEAK_GBS

    primary = refine_dp if (dominant == "dp::k_dp_coop" and refine_dp) else sweep
    if primary is None:
        return None
    out = dict(primary)
    out["measured_in"] = measured_in
    out["dominant_kernel"] = dominant
    out["kernels_ms_per_step"] = {k: round(v["ms_per_step"], 4) for k, v in sorted(kernels.items())}
    out["kernel_launches"] = kernels
    if primary is not sweep and sweep:
        out["sweep"] = sweep
    if primary is not refine_dp and refine_dp:
        out["refine_dp"] = refine_dp
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n1", type=int, default=150)
    ap.add_argument("--n2", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-n", type=int, default=400, help="size of the second pair whose sweep is timed with N > 1 (0 = off)")
    ap.add_argument("--extra-sweep", action="store_true", help="time the second pair's sweep on one GPU too")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: CPU ranks (tests, with --engine-factory)")
    ap.add_argument("--group-of-one", action="store_true",
                    help="with --gpus 1: form the process group anyway and run the collectives of the N > 1 path")
    ap.add_argument("--engine-factory", default="", help="test hook: module:function that builds a stand-in engine")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if args.backend == "gloo" and not args.engine_factory:
        raise SystemExit("bench.py: the gloo backend is for tests with a stand-in engine (--engine-factory)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.backend))
    rank_main(args)


if __name__ == "__main__":
    main()
