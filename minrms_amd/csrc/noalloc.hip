// noalloc.hip -- minrms_search_noalloc: the search of a short orientation list
// with every intermediate buffer in one static device array.
#include "engine_internal.h"

// ---------------------------------------------------------------------------
// Search of a short orientation list without any allocation
// (minrms_search_noalloc): the work of minrms_engine_search + finalize_search +
// results for at most a few dozen orientations, with every intermediate buffer
// -- distance tables, back-pointers, hand-over planes, costs, per-(slot, n)
// alignments / sums / transforms, keys, row counters, error word -- carved out
// of one static device array instead of hipMalloc blocks.
// ---------------------------------------------------------------------------
#define MINRMS_NOALLOC_SCRATCH_BYTES (32u << 20)
#define MINRMS_NOALLOC_MAX_ORIENTATIONS 1024
static __device__ __align__(256) unsigned char scratch[MINRMS_NOALLOC_SCRATCH_BYTES];

namespace noalloc {
static std::mutex mu;
static unsigned char *ptr_of_device[64];        // device address of `scratch` on each GPU (every GPU holds its own copy)

// Everything the entry point needs that may allocate or load code, once per
// device, `device` current: the device address of this GPU's copy of the array
// (in host code the array's name is only a handle for this lookup, never an
// address; the lookup loads this unit's code object, which is what reserves the
// array's memory), the larger dynamic LDS of the cooperative kernels, and the
// code objects of the other units whose kernels the call launches (the runtime
// loads a code object when something of it is first used -- not something to
// leave to a call that may run under stream capture).
static int prepare(int device, unsigned char **out)
{
  if (device < 0 || device >= 64) return fail(MINRMS_E_DEVICE, "no HIP device %d", device);
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr_of_device[device]) {
    void *p = nullptr;
    const hipError_t r = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (r != hipSuccess || !p)
      return fail(MINRMS_E_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(r));
    { int rc = allow_large_lds(); if (rc) return rc; }
    { int rc = load_code_objects(); if (rc) return rc; }
    ptr_of_device[device] = (unsigned char *)p;
  }
  if (out) *out = ptr_of_device[device];
  return 0;
}

// the caller's current device comes back when the call returns
struct DeviceScope {
  int prev = -1;
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace noalloc

extern "C" int minrms_noalloc_prepare(int32_t device)
{
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MINRMS_E_DEVICE, "no HIP device %d", device);
  noalloc::DeviceScope scope;
  if (hipGetDevice(&scope.prev) != hipSuccess) scope.prev = -1;
  HIP_TRY(hipSetDevice(device));
  return noalloc::prepare(device, nullptr);
}

// keys = "none", zero everything else that must start clean (a fill kernel, not
// hipMemsetAsync: the caller may be capturing the stream into a graph)
__global__ void __launch_bounds__(256)
k_noalloc_clear(unsigned long long *keys, const int n_keys, uint32_t *zero_a, const int64_t n_a,
                uint32_t *zero_b, const int64_t n_b, uint32_t *zero_c, const int64_t n_c)
{
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t q = t; q < n_keys; q += stride) keys[q] = ~0ull;
  for (int64_t q = t; q < n_a; q += stride) zero_a[q] = 0u;
  for (int64_t q = t; q < n_b; q += stride) zero_b[q] = 0u;
  for (int64_t q = t; q < n_c; q += stride) zero_c[q] = 0u;
}

// Result tables straight from the batch records: the winner of entry k is slot
// (key & 0xffffffff) of the batch that is still in the scratch buffer.  Entries
// without a solution get what minrms_engine_reset leaves in the tables (1e26f,
// identity, -1).  If the cooperative DP kernel gave up on a wait (*err), nothing
// of the batch can be trusted: every sum becomes NaN and every winner -3.
__global__ void __launch_bounds__(64)
k_pick_direct(const SearchParams sp, const int n_max, const unsigned long long *__restrict__ keys,
              const int *__restrict__ err, const float *__restrict__ ssd_all, const float *__restrict__ xf_all,
              const short2 *__restrict__ pairs_all, float *__restrict__ best_ssd, float *__restrict__ best_xf,
              short2 *__restrict__ best_pairs, long long *__restrict__ winner)
{
  const int k = blockIdx.x, tid = threadIdx.x;
  const unsigned long long key = keys[k];
  const bool failed = *err != 0;
  const bool none = failed || key == ~0ull;
  const int64_t slot = (int64_t)(key & 0xffffffffull);
  const int64_t rec = none ? 0 : slot * sp.range_n + k;
  if (tid == 0) {
    if (best_ssd) best_ssd[k] = failed ? __int_as_float(0x7fc00000) : none ? MINRMS_UPPER_BOUND : ssd_all[rec];
    if (winner) winner[k] = failed ? -3 : none ? -1 : (long long)slot;
  }
  if (best_xf && tid < 12)
    best_xf[12 * k + tid] = none ? ((tid == 0 || tid == 5 || tid == 10) ? 1.0f : 0.0f) : xf_all[12 * rec + tid];
  if (best_pairs) {
    const int n = sp.n_min + k;
    for (int m = tid; m < n_max; m += 64)
      best_pairs[(int64_t)k * n_max + m] = (!none && m < n) ? pairs_all[rec * n_max + m] : make_short2(-1, -1);
  }
}

extern "C" int minrms_search_noalloc(const minrms_cfg *cfg, const float *d_xyzA, const float *d_xyzB,
                                     const float *d_og, int32_t n_og, float *d_ssd, float *d_xforms,
                                     int16_t *d_pairs, int64_t *d_winner_oid, void *stream)
{
  if (!cfg || !d_xyzA || !d_xyzB || !d_og) return fail(MINRMS_E_ARG, "null argument");
  if (n_og < 1 || n_og > MINRMS_NOALLOC_MAX_ORIENTATIONS)
    return fail(MINRMS_E_ARG, "minrms_search_noalloc takes 1..%d orientations, not %d",
                MINRMS_NOALLOC_MAX_ORIENTATIONS, n_og);
  if (cfg->of_enable || cfg->num_solutions > 1)
    return fail(MINRMS_E_ARG, "minrms_search_noalloc: the -of filter and alternate solutions need an engine");
  // an engine-shaped view of the scratch buffer: it owns nothing, and nothing of
  // it outlives this call
  minrms_engine v;
  { int rc = configure(&v, *cfg); if (rc) return rc; }
  const minrms_cfg &c = v.cfg;
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (c.device < 0 || c.device >= ndev) return fail(MINRMS_E_DEVICE, "no HIP device %d", c.device);
  noalloc::DeviceScope scope;
  if (hipGetDevice(&scope.prev) != hipSuccess) scope.prev = -1;
  HIP_TRY(hipSetDevice(c.device));
  unsigned char *base = nullptr;
  { int rc = noalloc::prepare(c.device, &base); if (rc) return rc; }      // a table look-up after the first call on a device

  const DpLayout &L = v.L;
  const int rn = v.sp.range_n;
  const int64_t count = n_og;
  const int64_t bp_words = std::max(L.bp_words, v.coop_pb ? v.Lc.bp_words : 0);
  const int64_t plane_floats = std::max(L.plane_floats, v.coop_pb ? v.Lc.plane_floats : 0);
  int64_t used = 0;
  auto carve = [&](int64_t bytes) -> unsigned char * {
    unsigned char *p = base + used;
    used += round_up(std::max<int64_t>(bytes, 4), 256);
    return p;
  };
  v.d_og = const_cast<float *>(d_og);
  v.d_xyzA = const_cast<float *>(d_xyzA);
  v.d_xyzB = const_cast<float *>(d_xyzB);
  v.d_C = (float *)carve(4 * L.c_floats * count);
  v.d_planes = (float *)carve(4 * plane_floats * count);
  const int64_t gprog_ints = (v.coop_pb != 0 && v.coop_groups.n_groups > 1) ? count * v.coop_groups.total_tasks : 0;
  v.d_gprog = (int *)carve(4 * gprog_ints);
  v.d_err = (int *)carve(256);
  v.d_keys = (unsigned long long *)carve(8 * (int64_t)rn);
  v.d_bp = (uint32_t *)carve(4 * bp_words * count);
  v.d_edges = (float *)carve(4 * L.edge_floats * count);
  v.d_cost = (float *)carve(4 * (int64_t)L.n_max * count);
  v.d_stop = (int32_t *)carve(4 * count);
  v.d_ssd = (float *)carve(4 * (int64_t)rn * count);
  v.d_xf = (float *)carve(48 * (int64_t)rn * count);
  v.d_pairs = (short2 *)carve((int64_t)sizeof(short2) * rn * L.n_max * count);
  if (used > (int64_t)MINRMS_NOALLOC_SCRATCH_BYTES)
    return fail(MINRMS_E_ARG,
                "minrms_search_noalloc: %d orientations of a %dx%d pair need %lld bytes of scratch, the static "
                "buffer holds %u: pass fewer orientations per call or use an engine",
                n_og, c.N1, c.N2, (long long)used, (unsigned)MINRMS_NOALLOC_SCRATCH_BYTES);
  v.stream = (hipStream_t)stream;
  v.batch_cap = count;
  v.coop_max_batch = MINRMS_NOALLOC_MAX_ORIENTATIONS;
  v.timing = 0;

  // The array keeps what the last call left in it, carved up for that call's
  // sizes: the row counters, the error word and the keys must start clean, and
  // the distance tables and planes must hold floats (their padding is read).
  hipLaunchKernelGGL(k_noalloc_clear, dim3(256), dim3(256), 0, v.stream, v.d_keys, rn,
                     (uint32_t *)v.d_C, L.c_floats * count, (uint32_t *)v.d_planes, plane_floats * count,
                     (uint32_t *)v.d_gprog, (int64_t)((unsigned char *)v.d_keys - (unsigned char *)v.d_gprog) / 4);
  int rc = run_batch(&v, count, true, nullptr, nullptr, nullptr, true, false);
  v.stream = nullptr;                        // the caller's stream: not ours to destroy
  if (rc) return rc;
  launch_reduce(v.sp, count, 0, nullptr, v.d_cost, v.d_ssd, rn, v.d_keys, v.d_stop, (hipStream_t)stream);
  hipLaunchKernelGGL(k_pick_direct, dim3(rn), dim3(64), 0, (hipStream_t)stream, v.sp, c.n_max, v.d_keys, v.d_err,
                     v.d_ssd, v.d_xf, v.d_pairs, d_ssd, d_xforms, (short2 *)d_pairs, (long long *)d_winner_oid);
  HIP_TRY(hipGetLastError());
  return MINRMS_OK;
}

