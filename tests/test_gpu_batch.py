"""All-against-all batch on the GPU: every pair of a set of domains goes through
the HIP engine (orientations, Search, Refine); curves, alignments and iteration
counts must be the reference's (one reference run per pair)."""
import numpy as np
import pytest

from test_batch_cpu import load_batch

pytestmark = pytest.mark.gpu


# doms6: 6 domains of 27-55 residues; doms12_80to200: the first 12 of the 64 domains of 80-200 residues of the
# all-against-all benchmark (tools/allvsall_bench.py), 66 pairs, one reference run each
@pytest.mark.parametrize("name,concurrency,workspace", [("doms6", 1, 1 << 30), ("doms6", 3, 1 << 30),
                                                        ("doms12_80to200", 4, 0)])
def test_batch_matches_reference_pair_by_pair(name, concurrency, workspace):
    from minrms_amd.batch import align_pairs
    g, structures, pairs = load_batch(name)
    print("case batch", name, len(pairs), "pairs, in flight:", concurrency, flush=True)
    kw = dict(workspace_bytes=workspace) if workspace else {}
    curves, mine = align_pairs(structures, pairs, concurrency=concurrency, **kw)
    assert sorted(mine) == list(range(len(pairs)))
    want = g["refine_ssd"][:, :curves.shape[1]]
    assert np.array_equal(curves.view(np.uint32), want.view(np.uint32))
    want_pairs, want_iters, want_og = g["refine_pairs"], g["refine_iters"], g["og_count"]   # (an archive member is read anew at every access)
    for p, res in mine.items():
        assert res.refine_iterations == int(want_iters[p])
        assert res.orientations == int(want_og[p])
        for k, al in enumerate(res.alignments):
            n = res.n_min + k
            if al is None:
                assert want[p, n - 1] >= np.float32(1e26)
            else:
                assert np.array_equal(al, want_pairs[p, n - 1, :n]), (p, n)


def test_engines_in_a_row_reuse_device_memory():
    """Engines of changing sizes, one after another: results stay right when the
    buffers of a closed engine are handed to the next one, and trim() gives the
    memory back."""
    import torch
    from oracle import oracle_py as orc
    from minrms_amd import engine as eng
    from minrms_amd.synth import synth_ca
    sizes = [(40, 52), (64, 30), (33, 33), (70, 45), (40, 52)]
    for n1, n2 in sizes:
        print("case reuse %dx%d" % (n1, n2), flush=True)
        a, b = synth_ca(n1, 501), synth_ca(n2, 502)
        ocfg = orc.make_cfg(n1, n2)
        og = orc.gen_orientations(ocfg, a, b)
        t = orc.search(ocfg, a, b, og)
        orc.refine(ocfg, a, b, t)
        with eng.Engine(eng.make_cfg(n1, n2, workspace_bytes=1 << 30), a, b) as e:
            assert np.array_equal(e.gen_orientations().view(np.uint32), og.view(np.uint32))
            e.search(og)
            e.finalize_search()
            e.refine()
            ssd, xf, pairs, _ = e.results()
        assert np.array_equal(ssd.view(np.uint32), t.ssd.view(np.uint32))
        ok = t.ssd < np.float32(1e26)
        assert np.array_equal(xf.view(np.uint32)[ok], t.xform.view(np.uint32)[ok])
    free_before = torch.cuda.mem_get_info()[0]
    eng.trim()
    assert torch.cuda.mem_get_info()[0] > free_before


def _worker_cut_gpu(rank, world, port, out, name="doms6", workspace=1 << 30):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.init()
    from minrms_amd.batch import align_pairs, pair_cost

    def all_reduce(op):
        def f(x):                                   # gloo gets CPU tensors only
            t = torch.from_numpy(np.array(x, copy=True))
            dist.all_reduce(t, op=op)
            return t.numpy()
        return f

    g, structures, pairs = load_batch(name)
    L = g["lengths"]
    costs = sorted(pair_cost(int(L[i]), int(L[j])) for i, j in pairs)
    print("rank", rank, "batch", name, "with the larger pairs cut into two o_id ranges", flush=True)
    curves, mine = align_pairs(structures, pairs, rank=rank, world_size=world, workspace_bytes=workspace,
                               all_reduce_sum=all_reduce(dist.ReduceOp.SUM), all_reduce_min=all_reduce(dist.ReduceOp.MIN),
                               split_above=costs[len(costs) // 2])
    np.savez(out % rank, curves=curves, mine=np.array(sorted(mine)),
             iters=np.array([mine[p].refine_iterations for p in sorted(mine)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,workspace", [("doms6", 1 << 30), ("doms12_80to200", 24 << 30)])
def test_pairs_and_orientations_co_sharded_over_two_ranks(name, workspace, tmp_path):
    """Two ranks on the one GPU: whole pairs and o_id ranges of the larger pairs
    dealt as one list of work units; curves and iteration counts must be the
    reference's on both ranks (6 small domains; 12 domains of the benchmark's sizes)."""
    import torch.multiprocessing as mp
    from test_batch_cpu import _free_port
    out = str(tmp_path / "g%d.npz")
    mp.spawn(_worker_cut_gpu, args=(2, _free_port(), out, name, workspace), nprocs=2, join=True)
    g, _, pairs = load_batch(name)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert sorted(r0["mine"].tolist() + r1["mine"].tolist()) == list(range(len(pairs)))
    want = g["refine_ssd"][:, :r0["curves"].shape[1]]
    for r in (r0, r1):
        assert np.array_equal(r["curves"].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(r["iters"], g["refine_iters"][r["mine"]])


def test_all_64_domains_against_all():
    """The all-against-all benchmark itself: 64 synthetic domains of 80-200
    residues, 2 016 pairs, four in flight.  The 66 pairs among the first twelve
    domains are checked against the reference (curves, alignments, iteration
    counts: one reference run per pair); every other pair must have a solution for
    every n >= 2 (a full sweep without cutoffs always has), and its curve must be
    the one a lone engine computes for it (a fixed sample of 12 pairs)."""
    import random
    from minrms_amd.batch import align_pairs, all_pairs, solve_pair_hip
    from minrms_amd.synth import synth_ca
    g, gold_structures, gold_pairs = load_batch("doms12_80to200")
    rng = random.Random(64)
    lengths = [rng.randint(80, 200) for _ in range(64)]
    structures = [synth_ca(n, 2000 + k) for k, n in enumerate(lengths)]
    for k, sref in enumerate(gold_structures):              # the golden's domains ARE the benchmark's first twelve
        assert np.array_equal(structures[k].view(np.uint32), sref.view(np.uint32))
    pairs = all_pairs(64)
    print("case all-against-all,", len(pairs), "pairs", flush=True)
    curves, mine = align_pairs(structures, pairs, concurrency=4)
    assert sorted(mine) == list(range(len(pairs)))
    index = {ij: p for p, ij in enumerate(pairs)}
    want_ssd, want_pairs, want_iters = g["refine_ssd"], g["refine_pairs"], g["refine_iters"]
    for q, ij in enumerate(gold_pairs):
        p = index[tuple(ij)]
        n_max = min(lengths[ij[0]], lengths[ij[1]])
        assert np.array_equal(curves[p, :n_max].view(np.uint32), want_ssd[q, :n_max].view(np.uint32)), ij
        assert mine[p].refine_iterations == int(want_iters[q])
        for k, al in enumerate(mine[p].alignments):
            n = mine[p].n_min + k
            if al is not None:
                assert np.array_equal(al, want_pairs[q, n - 1, :n]), (ij, n)
    for p, (i, j) in enumerate(pairs):
        n_max = min(lengths[i], lengths[j])
        assert np.all(curves[p, 1:n_max] < np.float32(1e26)) and curves[p, 0] >= np.float32(1e26), (i, j)
    sample = random.Random(7).sample([p for p in range(len(pairs)) if max(pairs[p]) >= 12], 12)
    for p in sample:
        i, j = pairs[p]
        print("case lone engine for pair", i, j, flush=True)
        alone = solve_pair_hip(i, j, structures[i], structures[j])
        assert np.array_equal(alone.ssd.view(np.uint32), mine[p].ssd.view(np.uint32))
        assert alone.refine_iterations == mine[p].refine_iterations
