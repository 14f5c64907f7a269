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

Metric: reference DP cells per second (one cell = one (n,i,j) update of
Dyn3d::Solve_IncrN / InitTable_IncrN, lib/dyn/dyn3d.cc:135-387 of the
reference): (orientations searched + DP tables evaluated by refinement) x cells
per orientation / wall time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_CELL = 9.0           # 4 B cost read + 4 B cost write + 1 B back-pointer
# What bounds the DP kernel is the issue of vector instructions: 7 per cell and
# wave (add, min with DPP shift, 2 x compare + add-with-carry for the two tie
# flags, min).  One wave64 instruction occupies a SIMD's vector issue for 4
# cycles: measured with tools/ubench/dp_cell_ubench (profiles/r02_ubench_dp_cell.txt:
# 3.6-3.8 cycles per instruction and SIMD for the cell sequence and 4.4-4.7 for
# independent v_add_f32 / v_fma_f32 at 2-5 waves per SIMD; two waves on one SIMD
# each take 2 x the cycles of a wave alone).  256 CUs x 4 SIMDs at 2.4 GHz.
VALU_PER_CELL, CYCLES_PER_VALU, N_SIMD, CLOCK_HZ = 7, 4, 1024, 2.4e9
PMC_FILE = "r02_pmc_dp_block0.json"      # FETCH_SIZE / WRITE_SIZE passes of this round (tools/profile_round.sh)


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


def run_job(so):
    """One step: the complete job through the SearchOr-shaped interface.
    Returns (orientations, DP tables refinement evaluated, refinement iterations)."""
    so.reset()
    og = so.generate_orientations()
    so.Search(og)                 # this rank's o_id range + MIN all-reduce of the keys
    st = so.Refine()
    so.tables()
    return og.shape[0], st.dp_runs, st.num_iterations


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n1", type=int, default=150)
    ap.add_argument("--n2", type=int, default=150)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from minrms_amd import engine as eng
    from minrms_amd.search_or import SearchOrHip
    from minrms_amd.synth import synth_pair

    xyzA, xyzB = synth_pair(args.n1, args.n2)
    cfg = eng.make_cfg(args.n1, args.n2, device=local_rank)
    cells_per_o = eng.cells_per_orientation(cfg)
    so = SearchOrHip(cfg, xyzA, xyzB, rank=rank, world_size=world)

    def step():
        return run_job(so)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    so.engine.enable_timing(True)
    for b in range(so.engine.num_blocks()):
        so.engine.dp_block_time(b)    # drain
    sync()
    t0 = time.perf_counter()
    n_or = n_ref = iters = 0
    for _ in range(args.steps):
        o, r, it = step()
        n_or += o
        n_ref += r
        iters = it
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    so.engine.enable_timing(False)

    # dominant kernel = the layer block with the most GPU time (this rank)
    blocks = []
    for b in range(so.engine.num_blocks()):
        ms, launches, orients, cells_b = so.engine.dp_block_time(b)
        rows, panels, width = so.engine.dp_block_shape(b)
        blocks.append({"block": b, "ms": ms, "launches": launches, "orientations": orients,
                       "cells_per_orientation": cells_b, "rows": rows, "panels": panels, "panel_width": width})
    dom = max(blocks, key=lambda x: x["ms"])
    # HBM bytes of that launch from the PMC passes committed under profiles/
    # (FETCH_SIZE and WRITE_SIZE, one rocprofv3 --pmc run each); only valid for
    # the default workload they were taken on.
    traffic = None
    pmc = os.path.join(ROOT, "profiles", PMC_FILE)
    if os.path.exists(pmc) and (args.n1, args.n2, world) == (150, 150, 1) and dom["block"] == 0:
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]
    roofline = None
    if dom["ms"] > 0:
        launches = max(1, dom["launches"])
        sec = dom["ms"] * 1e-3
        # wave instructions the launches execute: every wave (= orientation) visits
        # rows x panels x panel_width cells with 7 vector instructions each
        wave_instr = dom["orientations"] * dom["rows"] * dom["panels"] * dom["panel_width"] * VALU_PER_CELL
        peak_issue = N_SIMD * CLOCK_HZ / CYCLES_PER_VALU          # wave instructions per second, whole GPU
        achieved_issue = wave_instr / sec
        alg_bytes = BYTES_PER_CELL * dom["cells_per_orientation"] * dom["orientations"]
        roofline = {
            # the bound that binds: vector-instruction issue (not HBM, not MFMA: nothing here is a contraction)
            "bound": "valu_issue",
            "achieved": achieved_issue / 1e9, "peak": peak_issue / 1e9, "unit": "G wave-instr/s",
            "frac": achieved_issue / peak_issue,
            "peak_basis": "1 wave64 vector instruction per %d cycles per SIMD (measured, profiles/r02_ubench_dp_cell.txt), "
                          "%d SIMDs, %.1f GHz" % (CYCLES_PER_VALU, N_SIMD, CLOCK_HZ / 1e9),
            # the same launches against the 2 cycles the hardware guide lists for >= 2 waves per SIMD
            "frac_if_2_cycles_per_instr": achieved_issue / (N_SIMD * CLOCK_HZ / 2),
            "kernel": "dp::k_dp_panel (layer block %d)" % dom["block"],
            "launches": dom["launches"],
            "avg_launch_ms": dom["ms"] / launches,
            # HBM bytes per launch: FETCH_SIZE + WRITE_SIZE of the committed PMC passes, NOT measured in this run
            "traffic": traffic,
            "traffic_source": ("profiles/%s (separate rocprofv3 --pmc passes of the same workload), "
                               "not measured in this run" % PMC_FILE) if traffic else None,
            "hbm_frac_measured": (traffic / (sec / launches) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            # the 9 B/cell model of the reference's own table traffic (4 B cost read + 4 B cost
            # write + 1 B back-pointer per cell): not a bound here, the cost layers never leave registers
            "algorithmic_hbm": {"bytes_per_launch": alg_bytes / launches, "achieved_GBs": alg_bytes / sec / 1e9,
                                "peak_GBs": HBM_PEAK_GBS, "frac": alg_bytes / sec / 1e9 / HBM_PEAK_GBS},
            "executed_cells_over_reference_cells": (dom["rows"] * dom["panels"] * dom["panel_width"] * 63.0) /
                                                   max(1, dom["cells_per_orientation"]),
            "blocks": blocks}

    if rank == 0:
        # work = the DP tables the REFERENCE evaluates for this job: one per
        # orientation and one per refinement iteration (num_iterations counts
        # from 1).  Tables the engine evaluates speculatively are not counted.
        total_cells = (n_or + (iters - 1) * args.steps) * cells_per_o
        out = {
            "metric": "DP-cells/s, full RMSD-vs-N curve (search + refine), %dx%d pair" % (args.n1, args.n2),
            "value": total_cells / dt, "unit": "DP-cells/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic CA-only %dx%d pair (seeds 1234/5678), -fm 4 -HS -ir -r: "
                                   "full orientation sweep, n_min=1" % (args.n1, args.n2),
                       "orientations": n_or // args.steps, "refine_dp_runs_executed": n_ref // args.steps,
                       "refine_iterations": iters, "cells_per_orientation": cells_per_o,
                       "parallelism": "o_id range per GPU + min-all-reduce" if world > 1 else "1 GPU"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n1, args.n2, cells_per_o)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
