"""minrms_search_noalloc (the entry point that allocates nothing) against
minrms_engine_search + minrms_engine_results, bit for bit, through the C ABI.
Successive calls share the static device array; each is of a different size,
the second smaller than the first.

Device buffers come straight from the HIP runtime the engine itself is bound to
(ctypes), so the test does not depend on which library initialised the GPU
first in this process."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H2D, D2H = 1, 2


class Hip:
    def __init__(self):
        from minrms_amd import engine as eng
        eng.load_library()                      # pulls in the HIP runtime
        lib = None
        for name in ("libamdhip64.so.7", "libamdhip64.so"):
            try:
                lib = C.CDLL(name, mode=os.RTLD_NOLOAD | os.RTLD_NOW)     # the copy that is already loaded
                break
            except OSError:
                continue
        if lib is None:
            lib = C.CDLL("/opt/rocm/lib/libamdhip64.so.7")
        self.lib = lib
        lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        lib.hipFree.argtypes = [C.c_void_p]
        lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.live = []

    def ok(self, rc):
        assert rc == 0, "HIP error %d" % rc

    def empty(self, nbytes):
        p = C.c_void_p()
        self.ok(self.lib.hipMalloc(C.byref(p), max(nbytes, 4)))
        self.live.append(p)
        return p

    def put(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.empty(arr.nbytes)
        self.ok(self.lib.hipMemcpy(p, arr.ctypes.data_as(C.c_void_p), arr.nbytes, H2D))
        return p

    def get(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        self.ok(self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, D2H))     # waits for the null stream
        return out

    def free_bytes(self):
        self.ok(self.lib.hipDeviceSynchronize())
        f, t = C.c_size_t(0), C.c_size_t(0)
        self.ok(self.lib.hipMemGetInfo(C.byref(f), C.byref(t)))
        return f.value

    def release(self):
        for p in self.live:
            self.lib.hipFree(p)
        self.live = []


def _existing(cfg_kw, a, b, og):
    from minrms_amd import engine as eng
    with eng.Engine(eng.make_cfg(**cfg_kw), a, b) as e:
        e.search(og)
        return e.results()


def _noalloc(hip, cfg_kw, a, b, og):
    from minrms_amd import engine as eng
    cfg = eng.make_cfg(**cfg_kw)
    rn, nm = cfg.n_max - cfg.n_min + 1, cfg.n_max
    da, db, dog = hip.put(np.asarray(a, np.float32)), hip.put(np.asarray(b, np.float32)), hip.put(np.asarray(og, np.float32))
    ssd, xf, pairs, win = hip.empty(4 * rn), hip.empty(48 * rn), hip.empty(4 * rn * nm), hip.empty(8 * rn)
    eng.noalloc_prepare(cfg.device)            # the documented set-up call (a no-op after the first time)
    free0 = hip.free_bytes()
    eng.search_noalloc(cfg, da, db, dog, og.shape[0], ssd, xf, pairs, win, None)      # null stream
    free1 = hip.free_bytes()
    assert free0 - free1 < (1 << 20), "the call took %d bytes of device memory" % (free0 - free1)
    out = (hip.get(ssd, rn, np.float32), hip.get(xf, (rn, 12), np.float32),
           hip.get(pairs, (rn, nm, 2), np.int16), hip.get(win, rn, np.int64))
    hip.release()
    return out


def _same(x, y):
    for p, q in zip(x, y):
        assert p.shape == q.shape
        assert np.array_equal(np.ascontiguousarray(p).view(np.uint8), np.ascontiguousarray(q).view(np.uint8))


CASES = [
    # (N1, N2, orientations kept, extra settings); each call follows the one before it in one process
    (150, 150, 30, {}),
    (70, 80, 12, {}),
    (90, 100, 20, dict(n_min=30, max_rmsd=3.5)),
    (60, 50, 7, dict(refine_while_searching=False, max_dist=6.0)),
    (150, 150, 5, {}),
]


def test_no_alloc_matches_engine_two_calls_in_a_row():
    from minrms_amd import engine as eng
    from minrms_amd.synth import synth_pair
    hip = Hip()
    for n1, n2, keep, extra in CASES:
        print("case", n1, n2, keep, extra, flush=True)
        a, b = synth_pair(n1, n2, 1234 + n1, 5678 + n2)
        kw = dict(N1=n1, N2=n2, **extra)
        with eng.Engine(eng.make_cfg(**kw), a, b) as e:
            og = e.gen_orientations()[:keep]
        want = _existing(kw, a, b, og)
        got = _noalloc(hip, kw, a, b, og)
        _same(got, want)
        assert (want[0] < np.float32(1e26)).sum() > 0


def test_no_alloc_rejects_what_does_not_fit():
    from minrms_amd import engine as eng
    from minrms_amd.synth import synth_pair
    hip = Hip()
    a, b = synth_pair(150, 150)
    cfg = eng.make_cfg(150, 150)
    da, db = hip.put(a), hip.put(b)
    og = hip.put(np.zeros((200, 12), np.float32))
    with pytest.raises(eng.MinrmsError) as ei:
        eng.search_noalloc(cfg, da, db, og, 200, None, None, None, None, None)
    assert ei.value.code == eng.E_ARG and "bytes of scratch" in str(ei.value)
    with pytest.raises(eng.MinrmsError):
        eng.search_noalloc(cfg, da, db, og, 0, None, None, None, None, None)
    with pytest.raises(eng.MinrmsError):
        eng.search_noalloc(eng.make_cfg(150, 150, of_enable=True, of_max_dist=8.0, of_min_matches=50),
                           da, db, og, 4, None, None, None, None, None)
    hip.release()


def test_no_alloc_gives_up_cleanly_and_recovers(monkeypatch):
    """The give-up contract: if a bounded wait of the DP kernel runs out (forced
    from the host: every workgroup starts in the given-up state and still runs to
    its normal end), every sum is NaN and every winner -3; the next call, without
    the hook, is bit-exact again.  Also: any output pointer may be NULL."""
    from minrms_amd import engine as eng
    from minrms_amd.synth import synth_pair
    hip = Hip()
    n1, n2, keep = 150, 150, 6
    a, b = synth_pair(n1, n2)
    kw = dict(N1=n1, N2=n2)
    with eng.Engine(eng.make_cfg(**kw), a, b) as e:
        og = e.gen_orientations()[:keep]
    want = _existing(kw, a, b, og)
    print("case give-up", flush=True)
    monkeypatch.setenv("MINRMS_COOP_SPIN_LIMIT", "-1")
    ssd, xf, pairs, win = _noalloc(hip, kw, a, b, og)
    assert np.all(np.isnan(ssd)) and np.all(win == -3)
    monkeypatch.delenv("MINRMS_COOP_SPIN_LIMIT")
    print("case after the give-up", flush=True)
    _same(_noalloc(hip, kw, a, b, og), want)
    print("case null outputs", flush=True)
    cfg = eng.make_cfg(**kw)
    rn = cfg.n_max - cfg.n_min + 1
    da, db, dog = hip.put(a), hip.put(b), hip.put(og)
    d_ssd, d_win = hip.empty(4 * rn), hip.empty(8 * rn)
    eng.search_noalloc(cfg, da, db, dog, keep, d_ssd, None, None, d_win, None)
    _same((hip.get(d_ssd, rn, np.float32), hip.get(d_win, rn, np.int64)), (want[0], want[3]))
    eng.search_noalloc(cfg, da, db, dog, keep, None, None, None, None, None)
    hip.ok(hip.lib.hipDeviceSynchronize())
    hip.release()


def test_no_alloc_under_stream_capture():
    """The call captured into a graph (after the set-up call; nothing it records is
    a memset node), replayed three times with eager work and synchronisations in
    between: every replay gives the bits of the existing entry point."""
    import torch
    from minrms_amd import engine as eng
    from minrms_amd.synth import synth_pair
    n1, n2, keep = 90, 100, 9
    a, b = synth_pair(n1, n2, 77, 88)
    kw = dict(N1=n1, N2=n2)
    with eng.Engine(eng.make_cfg(**kw), a, b) as e:
        og_all = e.gen_orientations()
    og = og_all[:keep]
    want = _existing(kw, a, b, og)
    cfg = eng.make_cfg(**kw)
    rn, nm = cfg.n_max - cfg.n_min + 1, cfg.n_max
    dev = torch.device("cuda", 0)
    ta, tb = torch.from_numpy(np.ascontiguousarray(a)).to(dev), torch.from_numpy(np.ascontiguousarray(b)).to(dev)
    tog = torch.from_numpy(np.ascontiguousarray(og)).to(dev)
    t_ssd = torch.zeros(rn, dtype=torch.float32, device=dev)
    t_xf = torch.zeros((rn, 12), dtype=torch.float32, device=dev)
    t_pairs = torch.zeros((rn, nm, 2), dtype=torch.int16, device=dev)
    t_win = torch.zeros(rn, dtype=torch.int64, device=dev)
    eng.noalloc_prepare(0)
    stream = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    print("case capture", flush=True)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=stream):
        eng.search_noalloc(cfg, ta.data_ptr(), tb.data_ptr(), tog.data_ptr(), keep, t_ssd.data_ptr(), t_xf.data_ptr(),
                           t_pairs.data_ptr(), t_win.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for rep in range(3):
        print("case replay", rep, flush=True)
        for t in (t_ssd, t_xf, t_pairs, t_win):
            t.zero_()
        # eager work of the program itself between the replays: another list through the same static array
        tog2 = torch.from_numpy(np.ascontiguousarray(og_all[keep + rep:keep + rep + 4])).to(dev)
        t_ssd2 = torch.zeros(rn, dtype=torch.float32, device=dev)
        eng.search_noalloc(cfg, ta.data_ptr(), tb.data_ptr(), tog2.data_ptr(), 4, t_ssd2.data_ptr(), None, None, None,
                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert float(t_ssd2.min()) < 1e26
        graph.replay()
        torch.cuda.synchronize()
        _same((t_ssd.cpu().numpy(), t_xf.cpu().numpy(), t_pairs.cpu().numpy(), t_win.cpu().numpy()), want)
