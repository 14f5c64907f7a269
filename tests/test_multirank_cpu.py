"""world_size-2 test of the multi-GPU seam on CPU (gloo): each rank reduces its own
o_id range to per-N keys, the keys are MIN-all-reduced, and the merged winners
must be the single-process winners (and the reference's, from the golden)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import load, cfg_kwargs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _keys_from_per_on(per_on, o_base):
    """(bits(ssd) << 32 | o_id) minimum over the rows of per_on (NaN = not evaluated)."""
    ssd = np.where(np.isnan(per_on), np.float32(np.inf), per_on).astype(np.float32)
    bits = ssd.view(np.uint32).astype(np.uint64)
    oid = (np.arange(per_on.shape[0], dtype=np.uint64) + np.uint64(o_base))[:, None]
    keys = (bits << np.uint64(32)) | oid
    keys[np.isnan(per_on)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return keys.min(axis=0) if per_on.shape[0] else np.full(per_on.shape[1], 0xFFFFFFFFFFFFFFFF, np.uint64)


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_py as orc
    from minrms_amd.search_or import shard_range, keys_to_i64, i64_to_keys
    g = load(name)
    cfg = orc.make_cfg(**cfg_kwargs(g))
    og = g["og"]
    s, e = shard_range(og.shape[0], rank, world)
    # DistanceMetric's cutoff as the s filter calls that precede this share leave it
    t = orc.ResultTables(cfg)
    t.clip = orc.initial_clip(cfg)
    if cfg.of_enable:
        import ctypes
        for _ in range(s):
            t.clip = orc.lib().orc_clip_after_filter_call(ctypes.byref(cfg), t.clip)
    t = orc.search(cfg, g["xyza"], g["xyzb"], og, s, e, tables=t, per_on=True)   # CPU stand-in for the GPU sweep
    mine = keys_to_i64(_keys_from_per_on(t.per_on, s))
    tt = torch.from_numpy(mine.copy())
    dist.all_reduce(tt, op=dist.ReduceOp.MIN)
    merged = i64_to_keys(tt.numpy())
    if rank == 0:
        np.save(out, merged)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("name", ["ship41x13_full", "bb3_40x45_of_md"])
def test_two_rank_min_allreduce_reproduces_winners(tmp_path, name):
    out = str(tmp_path / "merged.npy")
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    merged = np.load(out)
    g = load(name)
    ok = g["search_ssd"] < np.float32(1e26)
    assert np.array_equal((merged != np.uint64(0xFFFFFFFFFFFFFFFF)), ok)
    oid = (merged & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ssd_bits = (merged >> np.uint64(32)).astype(np.uint32)
    assert np.array_equal(oid[ok], g["winner_oid"][ok])
    assert np.array_equal(ssd_bits[ok], g["search_ssd"].view(np.uint32)[ok])
