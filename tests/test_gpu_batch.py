"""All-against-all batch on the GPU: every pair of a set of domains goes through
the HIP engine (orientations, Search, Refine); curves, alignments and iteration
counts must be the reference's (one reference run per pair)."""
import numpy as np
import pytest

from test_batch_cpu import load_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("concurrency", [1, 3])
def test_batch_matches_reference_pair_by_pair(concurrency):
    from minrms_amd.batch import align_pairs
    g, structures, pairs = load_batch("doms6")
    print("case batch doms6,", len(pairs), "pairs, in flight:", concurrency, flush=True)
    curves, mine = align_pairs(structures, pairs, workspace_bytes=1 << 30, concurrency=concurrency)
    assert sorted(mine) == list(range(len(pairs)))
    want = g["refine_ssd"][:, :curves.shape[1]]
    assert np.array_equal(curves.view(np.uint32), want.view(np.uint32))
    for p, res in mine.items():
        assert res.refine_iterations == int(g["refine_iters"][p])
        assert res.orientations == int(g["og_count"][p])
        for k, al in enumerate(res.alignments):
            n = res.n_min + k
            if al is None:
                assert want[p, n - 1] >= np.float32(1e26)
            else:
                assert np.array_equal(al, g["refine_pairs"][p, n - 1, :n]), (p, n)


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
