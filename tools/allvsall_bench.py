#!/usr/bin/env python3
"""Time an all-against-all batch of synthetic domains (lengths U[80,200]) on the
GPUs of one node: pairs dealt over the ranks, the complete job per pair.

  python tools/allvsall_bench.py --domains 64                  # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
         --master-port P tools/allvsall_bench.py --domains 64  # N GPUs

Prints one JSON line on rank 0: pairs/s and reference DP cells/s (one table per
orientation and per refinement iteration, as in bench.py)."""
import argparse
import json
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--domains", type=int, default=64)
    ap.add_argument("--max-pairs", type=int, default=0, help="only the first K pairs (0 = all)")
    ap.add_argument("--seed", type=int, default=64)
    ap.add_argument("--concurrency", type=int, default=4, help="pairs in flight per GPU (the box has 4 hardware queues)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from minrms_amd import engine as eng
    from minrms_amd.batch import align_pairs, all_pairs
    from minrms_amd.synth import synth_ca
    rng = random.Random(args.seed)
    lengths = [rng.randint(80, 200) for _ in range(args.domains)]
    structures = [synth_ca(n, 2000 + k) for k, n in enumerate(lengths)]
    pairs = all_pairs(args.domains)
    if args.max_pairs:
        pairs = pairs[:args.max_pairs]
    # warm-up: one small pair (module load, first allocations)
    align_pairs(structures[:2], [(0, 1)], device=local_rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    curves, mine = align_pairs(structures, pairs, rank=rank, world_size=world, device=local_rank,
                               concurrency=args.concurrency)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cells = 0
    for p, res in mine.items():
        i, j = pairs[p]
        cfg = eng.make_cfg(lengths[i], lengths[j])
        cells += (res.orientations + max(0, res.refine_iterations - 1)) * eng.cells_per_orientation(cfg)
    if world > 1:
        t = torch.tensor([float(cells), dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t[:1], op=dist.ReduceOp.SUM)
        dist.all_reduce(t[1:], op=dist.ReduceOp.MAX)
        cells, dt = float(t[0].item()), float(t[1].item())
    slow = sorted(mine.items(), key=lambda kv: -kv[1].seconds)[:5]
    print("rank %d: %d pairs, slowest: %s" % (rank, len(mine), ", ".join(
        "%dx%d %.0f ms" % (lengths[r.i], lengths[r.j], r.seconds * 1e3) for _, r in slow)), flush=True)
    if rank == 0:
        print(json.dumps({"workload": "all-vs-all, %d synthetic domains of 80..200 residues, -fm 4 -HS -ir -r"
                                      % args.domains, "pairs": len(pairs), "n_gpus": world, "concurrency": args.concurrency,
                          "seconds": dt,
                          "pairs_per_s": len(pairs) / dt, "dp_cells_per_s": cells / dt,
                          "checksum": float(np.where(curves < 1e25, curves, 0).sum(dtype=np.float64))}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
