// minrms_engine.hip -- MI355X (gfx950) engine for the MINRMS exhaustive
// RMSD-vs-N search: kernels around the DP (dp_kernel.h) and the C ABI declared
// in include/minrms_engine.h.
//
// Reference behaviour restated here (paths relative to the reference tree):
//   bin/minrms/search_or_dyn3d.cc:1643-1946  FirstPass  -> minrms_engine_search
//   bin/minrms/search_or_dyn3d.cc:2261-2658  Refine     -> minrms_engine_refine
//   lib/biopolymer/apply_trans.cc:72-108 + lib/dyn/dyn3d.cc:664-701
//                                            -> k_cij
//   lib/dyn/dyn3d.cc:391-431 + lib/pair_alignment/pair_alignment.cc:714-820
//                                            -> k_trace_fit
//   lib/or_gen/or_gen.cc:237-587,179         -> minrms_engine_gen_orientations
//   lib/or_gen/orientation_filter.cc:42-130 + lib/dyn/dynNW.cc:147-531
//                                            -> nwf::k_nw_filter (nw_filter.h)
//   bin/minrms/search_or_dyn3d.cc:421-1640   alternate solutions by orientation
//                                            -> alt_solutions.h + the alt_* functions below
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fused
// multiply-add anywhere: float results must carry the reference's roundings).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include <algorithm>
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/minrms_engine.h"
#include "layout.h"
#include "dp_kernel.h"
#include "fit_device.h"
#include "nw_filter.h"
#include "alt_solutions.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(MINRMS_E_DEVICE, "%s failed: %s (%s:%d)", #expr,                 \
                  hipGetErrorString(e_), __FILE__, __LINE__);                      \
  } while (0)

extern "C" const char *minrms_engine_last_error(void) { return g_last_error.c_str(); }

// ---------------------------------------------------------------------------
// Device memory of destroyed engines is kept for the next engine of the process.
// A batch of pairs creates and destroys one engine per pair with buffers of a
// few GB; after some tens of hipMalloc/hipFree cycles of that size the runtime
// takes 0.3-1 s per call (measured), ten times the job itself.  Blocks are
// parked per device and per purpose (the name of the engine member) and handed
// out again when large enough; minrms_engine_trim() returns them to the system.
// ---------------------------------------------------------------------------
namespace pool {
struct Parked { void *ptr; size_t bytes; size_t clean; };   // clean: leading bytes that hold floats a kernel or a memset wrote (no NaN)
static std::mutex mu;
// pinned host blocks with their two events (the refinement's control snapshots), reused by later engines
struct HostSlot { int device; void *ptr; hipEvent_t ev[2]; };   // the events belong to a device
static std::vector<HostSlot> host_slots;
static std::map<std::pair<int, std::string>, std::vector<Parked>> parked;

// size classes 1/8 of a power of two apart: growing requests settle quickly
static size_t size_class(size_t bytes)
{
  size_t step = 256;
  while (step * 8 < bytes) step *= 2;
  return (bytes + step - 1) / step * step;
}

static size_t parked_bytes(int device)
{
  std::lock_guard<std::mutex> lock(mu);
  size_t total = 0;
  for (auto &kv : parked)
    if (kv.first.first == device)
      for (const Parked &b : kv.second) total += b.bytes;
  return total;
}

// device must be current
static hipError_t take(int device, const char *tag, size_t bytes, void **out, size_t *got, size_t *clean)
{
  *clean = 0;
  std::vector<void *> to_free;
  {
    std::lock_guard<std::mutex> lock(mu);
    std::vector<Parked> &v = parked[std::make_pair(device, std::string(tag))];
    size_t best = v.size();
    for (size_t q = 0; q < v.size(); ++q)
      if (v[q].bytes >= bytes && (best == v.size() || v[q].bytes < v[best].bytes)) best = q;
    if (best != v.size()) {
      *out = v[best].ptr;
      *got = v[best].bytes;
      *clean = v[best].clean;
      v.erase(v.begin() + (long)best);
      return hipSuccess;
    }
    // everything parked for this purpose is too small: make room for the larger block
    for (const Parked &b : v) to_free.push_back(b.ptr);
    v.clear();
  }
  for (void *q : to_free) (void)hipFree(q);     // outside the lock: hipFree can take long
  *got = size_class(bytes);
  hipError_t r = hipMalloc(out, *got);
  if (r == hipSuccess) return r;
  // Out of memory while blocks parked for OTHER purposes sit idle (the batch
  // budget counts them as available): give those of this device back and try
  // once more.
  (void)hipGetLastError();
  to_free.clear();
  {
    std::lock_guard<std::mutex> lock(mu);
    for (auto &kv : parked) {
      if (kv.first.first != device) continue;
      for (const Parked &b : kv.second) to_free.push_back(b.ptr);
      kv.second.clear();
    }
  }
  if (to_free.empty()) return r;
  for (void *q : to_free) (void)hipFree(q);
  return hipMalloc(out, *got);
}

static void park(int device, const char *tag, void *ptr, size_t bytes, size_t clean)
{
  std::lock_guard<std::mutex> lock(mu);
  parked[std::make_pair(device, std::string(tag))].push_back(Parked{ptr, bytes, clean});
}

static void trim()
{
  std::lock_guard<std::mutex> lock(mu);
  int current = 0;
  const bool have = hipGetDevice(&current) == hipSuccess;
  for (auto &kv : parked) {
    if (kv.second.empty()) continue;
    (void)hipSetDevice(kv.first.first);
    for (const Parked &b : kv.second) (void)hipFree(b.ptr);
    kv.second.clear();
  }
  for (const HostSlot &hs : host_slots) {
    (void)hipSetDevice(hs.device);
    (void)hipEventDestroy(hs.ev[0]);
    (void)hipEventDestroy(hs.ev[1]);
    (void)hipHostFree(hs.ptr);
  }
  host_slots.clear();
  if (have) (void)hipSetDevice(current);
}
}  // namespace pool

extern "C" void minrms_engine_trim(void) { pool::trim(); }

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
struct SearchParams {
  int32_t N1, N2, nbb, n_min, n_max, range_n;
  float max_rmsd;          // -1 = off
  float clip;              // DistanceMetric::max_allowed_dist_metric as the constructor sets it
  int32_t rws;             // refine_while_searching
};

// lib/vect_3d/vect_3d.h:145-155: ((m0*x + m1*y) + m2*z) + m3
__device__ __forceinline__ void xform_point(const float *xf, const float *p, float *out)
{
#pragma unroll
  for (int r = 0; r < 3; ++r)
    out[r] = xf[4 * r] * p[0] + xf[4 * r + 1] * p[1] + xf[4 * r + 2] * p[2] + xf[4 * r + 3];
}

// Squared-distance table of one orientation (lib/dyn/dyn3d.cc:664-701,
// lib/dyn/distance_metric.h:104-130).  Every thread re-derives the moved
// coordinates of its own residue j: 18 flops against one table entry.
__global__ void __launch_bounds__(256)
k_cij(const SearchParams sp, const int64_t c_floats, const float *__restrict__ xyzA,
      const float *__restrict__ xyzB, const float *__restrict__ og, float *__restrict__ C,
      const float uniform_clip, const float *__restrict__ clip_of_slot,
      const int32_t *__restrict__ og_row_of_slot, const int *__restrict__ live_count)
{
  const int cell = blockIdx.x * 256 + threadIdx.x;
  const size_t o = blockIdx.y;
  if (live_count && (int)o >= *live_count) return;
  if (cell >= sp.N1 * sp.N2) return;
  const int i = cell / sp.N2, j = cell - i * sp.N2;
  const float *xf = og + 12 * (og_row_of_slot ? (size_t)og_row_of_slot[o] : o);
  float out = 0.0f;
  for (int q = 0; q < sp.nbb; ++q) {
    const float *a = xyzA + 3 * ((size_t)i * sp.nbb + q);
    float bm[3];
    xform_point(xf, xyzB + 3 * ((size_t)j * sp.nbb + q), bm);
    const float dx = a[0] - bm[0], dy = a[1] - bm[1], dz = a[2] - bm[2];
    out += dx * dx + dy * dy + dz * dz;
  }
  const float clip = clip_of_slot ? clip_of_slot[o] : uniform_clip;
  if (out > clip) out = MINRMS_UPPER_BOUND;
  C[o * c_floats + cell] = out;
}

// The same table for one atom per residue (the usual case), written to be
// short on vector instructions: the sweep's launch of k_cij is bound by them
// (counters: the vector ALUs are busy 0.99 of its 1.09 ms; 80 instructions per
// cell, a third of them the division cell / N2 and 64-bit address arithmetic).
// One thread per column j moves its point of structure 2 once and walks down
// MINRMS_CIJ_ROWS rows; the point of structure 1 is the same for the whole
// wave and arrives through scalar loads.  Same operations in the same order per
// cell as k_cij.
#define MINRMS_CIJ_ROWS 16
__global__ void __launch_bounds__(256)
k_cij_one_atom(const SearchParams sp, const int64_t c_floats, const float *__restrict__ xyzA,
               const float *__restrict__ xyzB, const float *__restrict__ og, float *__restrict__ C,
               const float uniform_clip, const float *__restrict__ clip_of_slot,
               const int32_t *__restrict__ og_row_of_slot, const int *__restrict__ live_count)
{
  const size_t o = blockIdx.z;
  if (live_count && (int)o >= *live_count) return;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= sp.N2) return;
  const int i0 = blockIdx.y * MINRMS_CIJ_ROWS;
  const int i1 = min(i0 + MINRMS_CIJ_ROWS, sp.N1);
  const float *xf = og + 12 * (og_row_of_slot ? (size_t)og_row_of_slot[o] : o);
  float bm[3];
  xform_point(xf, xyzB + 3 * (size_t)j, bm);
  const float clip = clip_of_slot ? clip_of_slot[o] : uniform_clip;
  float *out = C + o * c_floats + (size_t)i0 * sp.N2 + j;
  for (int i = i0; i < i1; ++i, out += sp.N2) {
    const float *a = xyzA + 3 * (size_t)i;
    const float dx = a[0] - bm[0], dy = a[1] - bm[1], dz = a[2] - bm[2];
    float v = 0.0f;
    v += dx * dx + dy * dy + dz * dz;
    if (v > clip) v = MINRMS_UPPER_BOUND;
    *out = v;
  }
}

static void launch_cij(const SearchParams &sp, int64_t count, hipStream_t st, int64_t c_floats, const float *xyzA,
                       const float *xyzB, const float *og, float *C, float uniform_clip, const float *clip_of_slot,
                       const int32_t *og_row_of_slot, const int *live_count);

// Where the back-pointer words of one layer block start and how wide its rows
// are: copied from the launch arguments into LDS once per workgroup, because
// the tracebacks index it with a per-thread block number.
struct BlockRef { long long bp_off; int W; int j0; };

__device__ __forceinline__ void load_block_refs(const DpLayout &L, BlockRef *refs)
{
  if ((int)threadIdx.x < L.NB) {
    const BlockDesc &bd = L.blk[threadIdx.x];
    refs[threadIdx.x].bp_off = bd.bp_off;
    refs[threadIdx.x].W = bd.W;
    refs[threadIdx.x].j0 = bd.j0;
  }
}

// The word of 16 two-bit codes that holds cell (n, i, j) of block b = (n-1)/63 is
//   bp[blk.bp_off + ((i - blk.j0) * blk.W + (j - blk.j0) / 16) * 64 + (n - 63 b)];
// cell p = (j - blk.j0) % 16 sits at bits [31-2p : 30-2p]; code 0 = match,
// 2 = from (n,i-1,j), odd = from (n,i,j-1).

// Replays, last move first, a path recorded by k_trace_fit: 2 bits per move in
// mine[64 * word] (the words of 64 threads interleaved), 0 = left (j-1), 1 = up (i-1), 2 = match.  (ci, cj) = the cell
// the walk ended in.  Undoing the moves in reverse yields the matched pairs in
// ascending order, the order the reference sums them in; f(i, j) gets the
// 0-based residue indices of each.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// word w of this thread's path string (the words of 64 threads interleaved); PL: the strings are in LDS
template <bool PL>
__device__ __forceinline__ uint32_t path_word(const uint32_t *mine, const int w)
{
  return PL ? ((const lds_u32 *)mine)[w * 64] : mine[(size_t)w * 64];
}
template <bool PL>
__device__ __forceinline__ void set_path_word(uint32_t *mine, const int w, const uint32_t v)
{
  if (PL) ((lds_u32 *)mine)[w * 64] = v;
  else mine[(size_t)w * 64] = v;
}

template <bool PL, class F>
__device__ __forceinline__ void replay_path(const uint32_t *mine, int n_moves, int ci, int cj, F &&f)
{
  int q = n_moves - 1;
  while (q >= 0) {
    const uint32_t word = path_word<PL>(mine, q >> 4);
    const int t = q & 15;
    const uint32_t fields = t == 15 ? word : (word & ((1u << (2 * t + 2)) - 1u));
    if (fields == 0u) { cj += t + 1; q -= t + 1; continue; }     // only left moves below q in this word
    const int top = (31 - __builtin_clz(fields)) >> 1;
    cj += t - top;
    q -= t - top;
    ++ci;
    if (((word >> (2 * top)) & 3u) != 1u) { ++cj; f(ci - 1, cj - 1); }
    --q;
  }
}

// Traceback (Dyn3d::ExportSolution, lib/dyn/dyn3d.cc:391-431) and the
// per-(orientation, n) least-squares refit (PairwiseAlignment::CalcMinRMSD,
// lib/pair_alignment/pair_alignment.cc:714-820; ssd = SQR(rmsd)*n*num_bb,
// bin/minrms/search_or_dyn3d.cc:587-588).  One thread per (slot, k).
// The walk records its moves as a bit string in LDS (at most N1+N2 moves, 2
// bits each) instead of writing the matched pairs to memory; the refit reads the
// pairs back from there, twice (centroids, then the correlation sums).  The
// alignments are only written out when pairs_all is given: during the sweep
// nothing reads them (the winners are solved again), and as 4-byte stores
// scattered over one row per lane they were most of the kernel's memory traffic.
// Dynamic LDS: 16 BlockRef, the paths [path_words][64], then (if coords_in_lds)
// both coordinate sets.
typedef __attribute__((address_space(3))) float lds_float;

// One atom's coordinates, from LDS (a typed LDS read: three ds_read, 32-bit
// address) or from memory; LDSC says where at compile time, because through a
// generic pointer every read is a flat load behind 64-bit address arithmetic,
// and the refit of the sweep is bound by its vector instructions.
template <bool LDSC>
__device__ __forceinline__ void load_atom(const float *base, const int atom, float (&out)[3])
{
  if (LDSC) {
    const lds_float *p = (const lds_float *)base + 3 * atom;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
  } else {
    const float *p = base + 3 * (size_t)atom;
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
  }
}

// LDSC: both coordinate sets are in LDS; ONE: one atom per residue; PL: the path strings are in LDS.
template <bool LDSC, bool ONE, bool PL>
__global__ void __launch_bounds__(64)
k_trace_fit(const DpLayout L, const SearchParams sp, const int64_t n_slots,
            const uint32_t *__restrict__ bp_all, const float *__restrict__ cost_all,
            const float *__restrict__ xyzA_g, const float *__restrict__ xyzB_g,
            const float *__restrict__ og, short2 *__restrict__ pairs_all,
            float *__restrict__ ssd_all, float *__restrict__ xf_all, const int path_words,
            const int coords_in_lds, uint32_t *__restrict__ path_global,
            const int k_lo, const int k_cnt, const int32_t *__restrict__ stop_n)
{
  // k_lo, k_cnt: the entries k_lo .. k_lo+k_cnt-1 of every slot (the layers of one
  // layer block, when the sweep evaluates block by block); stop_n: slots whose
  // n-loop has already ended below these layers are left alone
  extern __shared__ __attribute__((aligned(16))) unsigned char tf_lds[];
  BlockRef *refs = (BlockRef *)tf_lds;
  uint32_t *lds_path = (uint32_t *)(tf_lds + MINRMS_MAX_BLOCKS * sizeof(BlockRef));
  load_block_refs(L, refs);
  const float *xyzA = xyzA_g, *xyzB = xyzB_g;
  if (LDSC) {
    float *la = (float *)(lds_path + (path_global ? 0 : (size_t)path_words * 64));
    float *lb = la + 3 * sp.N1 * sp.nbb;
    for (int q = threadIdx.x; q < 3 * sp.N1 * sp.nbb; q += 64) la[q] = xyzA_g[q];
    for (int q = threadIdx.x; q < 3 * sp.N2 * sp.nbb; q += 64) lb[q] = xyzB_g[q];
    xyzA = la; xyzB = lb;
  }
  const int lane = threadIdx.x;
  // this thread's path words: mine[64 * w], in LDS or (long paths) in memory
  uint32_t *mine = (path_global ? path_global + (size_t)blockIdx.x * path_words * 64 : lds_path) + lane;
  __syncthreads();
  const int64_t tid = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (tid >= n_slots * k_cnt) return;
  const int64_t slot = tid / k_cnt;
  const int k = k_lo + (int)(tid - slot * k_cnt);
  const int n = sp.n_min + k;
  float *ssd_out = ssd_all + slot * sp.range_n + k;
  if (stop_n && stop_n[slot] < n) return;
  if (n < 2) { *ssd_out = MINRMS_UPPER_BOUND; return; }   // n = 1 is never recorded (:1714)
  const float cost = cost_all[slot * L.n_max + (n - 1)];
  if (cost >= MINRMS_UPPER_BOUND) { *ssd_out = cost; return; }

  const uint32_t *bp = bp_all + slot * L.bp_words;
  int cn = n, ci = L.N1, cj = L.N2, pos = 0;
  uint32_t acc = 0u;                  // the word of the path string that is being filled
  int cur_b = -1;                     // layer block whose description is held in registers
  BlockRef bd = {0, 0, 0};
  while (cn > 0) {
    // a well-formed table never leaves the cells i >= n, j >= n; refuse to
    // follow a pointer that would (the result is then marked unusable)
    if (ci < cn || cj < cn) { *ssd_out = __int_as_float(0x7fc00000); return; }
    const int b = (cn - 1) / MINRMS_LAYERS_PER_BLOCK;
    if (b != cur_b) { bd = refs[b]; cur_b = b; }      // changes 0-2 times per walk: no LDS read per step
    const int c = cj - bd.j0;
    const int p = c & 15;
    const uint32_t word =
        bp[bd.bp_off + ((int64_t)(ci - bd.j0) * bd.W + (c >> 4)) * 64 + (cn - MINRMS_LAYERS_PER_BLOCK * b)];
    // cells p, p-1, ... of the word, lowest bits first: skip the run of "left" moves
    const uint32_t not_left = (~word & 0x55555555u) >> (30 - 2 * p);
    // moves to skip (left moves are zeros in the string), then one move to record:
    // at most 16 positions, so at most one word of the string is completed
    const bool only_left = not_left == 0u;
    const int run = only_left ? p + 1 : (__builtin_ctz(not_left) >> 1);
    cj -= run;
    if (((pos + run) >> 4) != (pos >> 4)) { set_path_word<PL>(mine, pos >> 4, acc); acc = 0u; }
    pos += run;
    if (only_left) continue;
    const uint32_t code = (word >> (30 - 2 * (p - run))) & 3u;
    if (code == 2u) { acc |= 1u << (2 * (pos & 15)); --ci; }
    else { acc |= 2u << (2 * (pos & 15)); --cn; --ci; --cj; }
    if ((pos & 15) == 15) { set_path_word<PL>(mine, pos >> 4, acc); acc = 0u; }
    ++pos;
  }
  set_path_word<PL>(mine, pos >> 4, acc);       // the word in progress (pos <= N1+N2 < 16 * path_words)

  if (pairs_all) {
    short2 *pairs = pairs_all + (slot * sp.range_n + k) * (int64_t)L.n_max;
    int m = 0;
    replay_path<PL>(mine, pos, ci, cj, [&](int i, int j) { pairs[m++] = make_short2((short)i, (short)j); });
  }
  float *xf_out = xf_all ? xf_all + (slot * sp.range_n + k) * 12 : nullptr;
  if (!sp.rws) {
    *ssd_out = cost;
    if (xf_out)
      for (int q = 0; q < 12; ++q) xf_out[q] = og[12 * slot + q];
    return;
  }
  fit::Superimpose S;
  S.begin();
  const int nbb = ONE ? 1 : sp.nbb;
  replay_path<PL>(mine, pos, ci, cj, [&](int i, int j) {
    for (int q = 0; q < nbb; ++q) {
      float f[3], m[3];
      load_atom<LDSC>(xyzA, i * nbb + q, f);
      load_atom<LDSC>(xyzB, j * nbb + q, m);
      S.sum(f, m);
    }
  });
  const int npts = n * nbb;
  S.centre(npts);
  replay_path<PL>(mine, pos, ci, cj, [&](int i, int j) {
    for (int q = 0; q < nbb; ++q) {
      float f[3], m[3];
      load_atom<LDSC>(xyzA, i * nbb + q, f);
      load_atom<LDSC>(xyzB, j * nbb + q, m);
      S.accumulate(f, m);
    }
  });
  float xf[12];
  const float rmsd = S.finish(npts, xf);
  *ssd_out = (rmsd * rmsd) * n * nbb;
  if (xf_out)
    for (int q = 0; q < 12; ++q) xf_out[q] = xf[q];
}

// The same work with one WAVE per (slot, k): for small batches (refinement
// rounds, the winners of a search) the thread-per-chain kernel above is bound
// by the latency of its dependent loads, one per row of the path.  Here the
// lanes fetch, two words each and all at once, the back-pointer words of every
// cell the path can reach within the next 9 row moves without leaving its layer
// block (a row move goes to (n, i-1) or to (n-1, i-1): 55 cells, two 16-column
// words each); the walk then runs on cross-lane reads, in scalar registers,
// until it leaves that cone.  The walk is what the kernel's time is made of
// (about 170 steps for a 150x150 pair), so its state is kept relative to the
// cone's origin and every step is a few scalar instructions:
//   ddi, ddn  rows and layers below the origin; tri = ddi(ddi+1)/2
//   c         column of the current cell relative to the block's first column
//   g         the same for the column the cone was fetched around (origin's
//             column minus ddn); g>>4 is the word lane tri+ddn holds in w0,
//             g>>4 - 1 the one it holds in w1
// A well-formed table never leaves the cells i >= n, j >= n, and i - n and j - n
// never grow along a walk: a walk that ends with n = 0 and i, j >= 0 has
// stayed inside.  Words outside the table are never loaded (they read as 0, a
// diagonal move), so a malformed table still ends the walk, and is marked.
// The refit is fit::wave_superimpose.
#define MINRMS_CONE_ROWS 9
#define MINRMS_CONE_CELLS 55          // (ROWS+1)(ROWS+2)/2 <= 64

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// lane 0 stores one word to LDS if `on` (0 or 1, wave-uniform); no branch around it
__device__ __forceinline__ void lds_store_lane0_if(const uint32_t on, const uint32_t byte_addr, const uint32_t value)
{
  asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, 0\n\tds_write_b32 %0, %1\n\ts_mov_b64 exec, -1"
               :: "v"(byte_addr), "v"(value), "s"(on) : "memory");
}

#ifdef MINRMS_COOP_STAMPS
__device__ unsigned long long g_round_stamps[8];    // diagnostic build: cycles of the round kernel's three parts, summed
__device__ unsigned long long g_trace_stamps[8];    // diagnostic build: cycles of the longest chain of slot 0
#define MINRMS_TSTAMP() __builtin_readcyclecounter()
#endif

__global__ void __launch_bounds__(64)
k_trace_fit_wave(const DpLayout L, const SearchParams sp, const int64_t n_slots,
                 const uint32_t *__restrict__ bp_all, const float *__restrict__ cost_all,
                 const float *__restrict__ xyzA, const float *__restrict__ xyzB,
                 const float *__restrict__ og, short2 *__restrict__ pairs_all,
                 float *__restrict__ ssd_all, float *__restrict__ xf_all,
                 const int *__restrict__ live_count)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char tfw_lds[];
  if (live_count && (int64_t)blockIdx.x >= (int64_t)*live_count * sp.range_n) return;
  BlockRef *refs = (BlockRef *)tfw_lds;
  float *chunk = (float *)(tfw_lds + MINRMS_MAX_BLOCKS * sizeof(BlockRef));   // [fit::WAVE_LDS_FLOATS]
  short2 *lpairs = (short2 *)(chunk + fit::WAVE_LDS_FLOATS);                  // [n_max]
  const uint32_t lpairs_at = (uint32_t)(uintptr_t)(lds_byte *)tfw_lds +
                             (uint32_t)(MINRMS_MAX_BLOCKS * sizeof(BlockRef) + fit::WAVE_LDS_FLOATS * 4);
  const int lane = threadIdx.x;
  const int64_t rec = blockIdx.x;
  if (rec >= n_slots * sp.range_n) return;
  const int64_t slot = rec / sp.range_n;
  const int k = (int)(rec - slot * sp.range_n);
  const int n = sp.n_min + k;
  float *ssd_out = ssd_all + rec;
  if (n < 2) { if (lane == 0) *ssd_out = MINRMS_UPPER_BOUND; return; }
  const float cost = cost_all[slot * L.n_max + (n - 1)];
  if (cost >= MINRMS_UPPER_BOUND) { if (lane == 0) *ssd_out = cost; return; }
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long ts_begin = MINRMS_TSTAMP();
  unsigned long long ts_fetch = 0, ts_walk = 0, ts_fetches = 0, ts_steps = 0;
#endif
  load_block_refs(L, refs);
  __syncthreads();

  const uint32_t *bp = bp_all + slot * L.bp_words;
  // this lane's cell of the cone: lane = di(di+1)/2 + dn, 0 <= dn <= di
  int e_di = 0;
  while ((e_di + 1) * (e_di + 2) / 2 <= lane) ++e_di;
  const int e_dn = lane - e_di * (e_di + 1) / 2;
  int cn = n, ci = L.N1, cj = L.N2;
  while (cn > 0) {
    if (ci < cn || cj < cn) break;                      // left the table: marked below
    const int b = (cn - 1) / MINRMS_LAYERS_PER_BLOCK;
    const BlockRef bd = refs[b];
    const int n_lo = MINRMS_LAYERS_PER_BLOCK * b;       // the block holds layers n_lo+1 .. n_lo+63
    const int j0 = n_lo + 1;                            // = bd.j0, but known to be the same in every lane
    const int on = cn, oi = ci, oc = cj - j0;
    uint32_t w0 = 0u, w1 = 0u;
#ifdef MINRMS_COOP_STAMPS
    const unsigned long long ts_a = MINRMS_TSTAMP();
#endif
    {
      const int ln = on - e_dn, li = oi - e_di, lg = oc - e_dn;
      if (lane < MINRMS_CONE_CELLS && ln > n_lo && li >= ln && lg >= 0) {
        const uint32_t *at = bp + bd.bp_off + ((int64_t)(li - j0) * bd.W + (lg >> 4)) * 64 + (ln - n_lo);
        w0 = at[0];
        if (lg >= 16) w1 = at[-64];
      }
    }
    // One pass of the loop = one row move of the walk, preceded by its run of
    // left moves.  A lone wave pays for every instruction (about 6 cycles each)
    // and dearly for every taken branch, so the pass is straight-line scalar
    // code.  The two words a lane holds are taken as ONE string of 32 cells,
    // V = w1:w0: the code of the cell q columns right of w1's first column sits
    // at bit 62 - 2q, for both words alike.  With g the column the cone was
    // fetched around at the current layer (lane idx holds the words g>>4 and
    // g>>4 - 1) and dd/2 the number of columns the walk is left of g,
    // q = 16 + (g & 15) - dd/2 and the bit position is sh = (30 & ~2g) + dd.
#ifdef MINRMS_COOP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts_b = MINRMS_TSTAMP();
    ts_fetch += ts_b - ts_a; ++ts_fetches;
#endif
    // Written out as one block of scalar instructions (36 per row move, one
    // taken branch); the compiler's version of the same loop, with its three
    // exits, came to about 60.  s[90:96] and m0 (the lane to write: a second SGPR source is not allowed) are its temporaries.  The matched pairs
    // of the cone are collected in the lanes of one vector register
    // (v_writelane, like v_readlane, does not look at exec) and go to LDS in one
    // store after the loop.
    int g = oc, ddi = 0, idx = 0, dd = 0;
    int row = oi;                                       // = oi - ddi
    int lay = on - n_lo;                                // layers left in this block
    uint32_t found = 0u;                                // lane q: the q-th match of this cone, (j - 1) << 16 | (i - 1)
    {
      asm volatile(
          "s_mov_b32 m0, 0\n"
          "Lwalk_%=:\n\t"
          "v_readlane_b32 s90, %[w0], %[idx]\n\t"
          "v_readlane_b32 s91, %[w1], %[idx]\n\t"
          "s_lshl_b32 s94, %[g], 1\n\t"
          "s_andn2_b32 s94, 30, s94\n\t"
          "s_add_i32 s96, s94, %[dd]\n\t"            // sh
          "s_cmp_gt_i32 s96, 62\n\t"
          "s_cbranch_scc1 Lwalk_out_%=\n\t"          // left of both words: fetch again
          "s_andn2_b64 s[92:93], %[k5], s[90:91]\n\t"
          "s_lshr_b64 s[92:93], s[92:93], s96\n\t"
          "s_cmp_eq_u64 s[92:93], 0\n\t"
          "s_cbranch_scc1 Lwalk_edge_%=\n\t"         // the run of left moves leaves both words
          "s_ff1_i32_b64 s94, s[92:93]\n\t"          // twice the length of the run
          "s_add_i32 %[dd], %[dd], s94\n\t"
          "s_add_i32 s96, s96, s94\n\t"
          "s_lshr_b64 s[92:93], s[90:91], s96\n\t"
          "s_and_b32 s92, s92, 3\n\t"                // code of the cell the run ends in
          "s_add_i32 %[ddi], %[ddi], 1\n\t"
          "s_add_i32 %[row], %[row], -1\n\t"
          "s_add_i32 %[idx], %[idx], %[ddi]\n\t"
          "s_cmp_lg_u32 s92, 2\n\t"
          "s_cselect_b32 s95, 1, 0\n\t"              // match: a row move that is not \"up\" (code 2)
          "s_lshr_b32 s94, %[dd], 1\n\t"
          "s_sub_i32 s94, %[g], s94\n\t"
          "s_add_i32 s94, s94, %[j0m1]\n\t"
          "s_lshl_b32 s94, s94, 16\n\t"
          "s_or_b32 s94, s94, %[row]\n\t"            // residues (row, column + j0 - 1)
          "v_writelane_b32 %[acc], s94, m0\n\t"     // lane = matches found so far in this cone; a move that is
          "s_add_i32 m0, m0, s95\n\t"                // not a match leaves the lane to the next one
          "s_add_i32 %[idx], %[idx], s95\n\t"
          "s_sub_i32 %[g], %[g], s95\n\t"
          "s_sub_i32 %[lay], %[lay], s95\n\t"
          "s_cmp_eq_u32 %[lay], 0\n\t"
          "s_cbranch_scc1 Lwalk_out_%=\n\t"          // the next layer is in the block below (or the walk is over)
          "s_cmp_le_i32 %[ddi], %[rows]\n\t"
          "s_cbranch_scc1 Lwalk_%=\n\t"
          "s_branch Lwalk_out_%=\n"                   // out of the cone
          "Lwalk_edge_%=:\n\t"
          "s_sub_i32 s94, 64, s96\n\t"
          "s_add_i32 %[dd], %[dd], s94\n"
          "Lwalk_out_%=:\n\t"
          // the cone's matches, last one first: the (on - q)-th pair of the alignment is in lane q
          "s_bfm_b64 exec, m0, 0\n\t"
          "ds_write_b32 %[at], %[acc]\n\t"
          "s_mov_b64 exec, -1"
          : [g] "+s"(g), [dd] "+s"(dd), [ddi] "+s"(ddi), [idx] "+s"(idx), [row] "+s"(row), [lay] "+s"(lay),
            [acc] "+v"(found)
          : [w0] "v"(w0), [w1] "v"(w1), [k5] "s"(0x5555555555555555ull), [j0m1] "s"(j0 - 1),
            [at] "v"(lpairs_at + 4u * (uint32_t)(on - 1 - lane)),
            [rows] "n"(MINRMS_CONE_ROWS)
          : "s90", "s91", "s92", "s93", "s94", "s95", "s96", "m0", "scc", "memory");
    }
    const int ddn = (on - n_lo) - lay;
#ifdef MINRMS_COOP_STAMPS
    ts_walk += MINRMS_TSTAMP() - ts_b;
    ts_steps += ddi;
#endif
    const int c = g - (dd >> 1);
    cn = on - ddn; ci = oi - ddi; cj = c + j0;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the pairs are in LDS
  const bool bad = cn != 0 || ci < 0 || cj < 0;
  if (bad) { if (lane == 0) *ssd_out = __int_as_float(0x7fc00000); return; }
  __syncthreads();
  short2 *pairs = pairs_all + rec * (int64_t)L.n_max;
  for (int m = lane; m < n; m += 64) pairs[m] = lpairs[m];

  float *xf_out = xf_all ? xf_all + rec * 12 : nullptr;
  if (!sp.rws) {
    if (lane == 0) *ssd_out = cost;
    if (xf_out && lane < 12) xf_out[lane] = og[12 * slot + lane];
    return;
  }
  float xf[12];
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long ts_c = MINRMS_TSTAMP();
#endif
  const float rmsd = fit::wave_superimpose(n, sp.nbb, lpairs, chunk, xyzA, xyzB, xf);
  if (lane == 0) {
    *ssd_out = (rmsd * rmsd) * n * sp.nbb;
    if (xf_out)
      for (int q = 0; q < 12; ++q) xf_out[q] = xf[q];
  }
#ifdef MINRMS_COOP_STAMPS
  if (lane == 0 && slot == 0 && k == sp.range_n - 1) {
    const unsigned long long ts_d = MINRMS_TSTAMP();
    g_trace_stamps[0] = ts_d - ts_begin; g_trace_stamps[1] = ts_fetch; g_trace_stamps[2] = ts_walk;
    g_trace_stamps[3] = ts_d - ts_c; g_trace_stamps[4] = ts_fetches; g_trace_stamps[5] = ts_steps;
    g_trace_stamps[6] = ts_c - ts_begin;
  }
#endif
}

// First n at which the reference's n-loop leaves an orientation
// (MaxDistanceExceeded, search_or_dyn3d.cc:1720; MaxRMSDExceeded, :1810);
// n_max+1 if it runs to the end.
__device__ __forceinline__ int first_break(const SearchParams &sp, const float *cost, const float *ssd)
{
  for (int n = 2; n <= sp.n_max; ++n) {
    if (cost[n - 1] >= MINRMS_UPPER_BOUND) return n;
    if (n < sp.n_min) continue;
    const float s = ssd[n - sp.n_min];
    if (sp.max_rmsd != -1.0f && s > (float)(n * sp.nbb) * (sp.max_rmsd * sp.max_rmsd)) return n;
  }
  return sp.n_max + 1;
}

// Sweep evaluated block by block (early termination): after the layers
// n_lo .. n_hi have their costs and sums, find whether the n-loop leaves the
// orientation among them (the rule of first_break, restricted to these n).
__global__ void __launch_bounds__(256)
k_update_stop(const SearchParams sp, const int64_t n_slots, const int n_lo, const int n_hi,
              const float *__restrict__ cost_all, const float *__restrict__ ssd_all, const int ssd_stride,
              int32_t *__restrict__ stop_n)
{
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= n_slots) return;
  if (stop_n[slot] <= sp.n_max) return;                  // already left
  const float *cost = cost_all + slot * sp.n_max;
  const float *ssd = ssd_all + slot * ssd_stride;
  for (int n = n_lo < 2 ? 2 : n_lo; n <= n_hi; ++n) {
    bool brk = cost[n - 1] >= MINRMS_UPPER_BOUND;
    if (!brk && n >= sp.n_min && sp.max_rmsd != -1.0f)
      brk = ssd[n - sp.n_min] > (float)(n * sp.nbb) * (sp.max_rmsd * sp.max_rmsd);
    if (brk) { stop_n[slot] = n; return; }
  }
}

__global__ void __launch_bounds__(256) k_fill_i32(int32_t *p, const int64_t n, const int32_t v)
{
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) p[t] = v;
}

// Alternate solutions by orientation keep, per n, every (orientation, n) record
// whose sum is within a tolerance of the best so far in o_id order
// (KeepTrackOfAltSolutions, search_or_dyn3d.cc:421-470).  Nearly all records of
// a sweep fail that test, so it is applied here first, with a little slack, and
// only the survivors travel to the host, which applies the exact rule to them
// (a record that lowers the running best always survives, so the host sees the
// same sequence of bests).
struct AltRecord { int32_t slot, k; float ssd; int32_t pad; float xf[12]; };

__global__ void __launch_bounds__(256)
k_first_break(const SearchParams sp, const int64_t n_slots, const float *__restrict__ cost_all,
              const float *__restrict__ ssd_all, int32_t *__restrict__ stop)
{
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot < n_slots) stop[slot] = first_break(sp, cost_all + slot * sp.n_max, ssd_all + slot * sp.range_n);
}

// The running best is a prefix minimum over the slots in o_id order.  The slots
// are cut into chunks of MINRMS_ALT_CHUNK: (1) the minimum of every chunk, per
// n; (2) per n, the best before each chunk (one thread walks the chunks); (3)
// one thread per (chunk, n) walks its chunk with the running best and appends
// the survivors.
#define MINRMS_ALT_CHUNK 128
__global__ void __launch_bounds__(64)
k_alt_chunk_min(const SearchParams sp, const int64_t n_slots, const float *__restrict__ ssd_all,
                const int32_t *__restrict__ stop, float *__restrict__ chunk_min)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= sp.range_n) return;
  const int n = sp.n_min + k;
  const int64_t s0 = (int64_t)blockIdx.y * MINRMS_ALT_CHUNK, s1 = min(s0 + MINRMS_ALT_CHUNK, n_slots);
  float best = MINRMS_UPPER_BOUND;
  if (n >= 2)
    for (int64_t s = s0; s < s1; ++s)
      if (n < stop[s]) best = fminf(best, ssd_all[s * sp.range_n + k]);
  chunk_min[(int64_t)blockIdx.y * sp.range_n + k] = best;
}

// chunk_min[c][k] becomes the best before chunk c (starting from best_in)
__global__ void __launch_bounds__(64)
k_alt_chunk_prefix(const SearchParams sp, const int n_chunks, const float *__restrict__ best_in,
                   float *__restrict__ chunk_min)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= sp.range_n) return;
  float best = best_in[k];
  for (int c = 0; c < n_chunks; ++c) {
    const float m = chunk_min[(int64_t)c * sp.range_n + k];
    chunk_min[(int64_t)c * sp.range_n + k] = best;
    best = fminf(best, m);
  }
}

// WRITE = false: counts the survivors of every (chunk, n) into counts[n][chunk];
// WRITE = true: writes them to rec, those of (chunk, n) from offsets[n][chunk] on:
// the records of one n end up together, in o_id order.
template <bool WRITE>
__global__ void __launch_bounds__(64)
k_alt_filter(const SearchParams sp, const int64_t n_slots, const int n_chunks, const float *__restrict__ ssd_all,
             const float *__restrict__ xf_all, const int32_t *__restrict__ stop,
             const float *__restrict__ best_before, const float tol2_with_slack, AltRecord *__restrict__ rec,
             int32_t *__restrict__ counts_or_offsets)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= sp.range_n) return;
  const int n = sp.n_min + k;
  const int64_t s0 = (int64_t)blockIdx.y * MINRMS_ALT_CHUNK, s1 = min(s0 + MINRMS_ALT_CHUNK, n_slots);
  float best = best_before[(int64_t)blockIdx.y * sp.range_n + k];
  int32_t at = WRITE ? counts_or_offsets[(int64_t)k * n_chunks + blockIdx.y] : 0;
  if (n >= 2)
    for (int64_t s = s0; s < s1; ++s) {
      if (n >= stop[s]) continue;
      const float v = ssd_all[s * sp.range_n + k];
      if (v < best) best = v;
      if (v <= best * tol2_with_slack) {
        if (WRITE) {
          AltRecord r;
          r.slot = (int32_t)s; r.k = k; r.ssd = v; r.pad = 0;
          for (int q = 0; q < 12; ++q) r.xf[q] = xf_all ? xf_all[(s * sp.range_n + k) * 12 + q] : 0.0f;
          rec[at] = r;
        }
        ++at;
      }
    }
  if (!WRITE) counts_or_offsets[(int64_t)k * n_chunks + blockIdx.y] = at;
}

// counts -> offsets (exclusive prefix sum, one workgroup); total[0] = number of survivors
__global__ void __launch_bounds__(256)
k_alt_offsets(const int64_t n, int32_t *__restrict__ counts, int32_t *__restrict__ total)
{
  __shared__ int32_t part[256];
  const int t = threadIdx.x;
  const int64_t per = (n + 255) / 256, lo = t * per, hi = min(lo + per, n);
  int32_t sum = 0;
  for (int64_t q = lo; q < hi; ++q) sum += counts[q];
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int32_t run = 0;
    for (int q = 0; q < 256; ++q) { const int32_t c = part[q]; part[q] = run; run += c; }
    total[0] = run;
  }
  __syncthreads();
  int32_t run = part[t];
  for (int64_t q = lo; q < hi; ++q) { const int32_t c = counts[q]; counts[q] = run; run += c; }
}

// Running per-N minimum (search_or_dyn3d.cc:1825-1835): strict "<" in o_id
// order == lexicographic minimum of (ssd, o_id) == unsigned min of
// (float bits << 32 | o_id) because ssd >= +0.
__global__ void __launch_bounds__(256)
k_reduce(const SearchParams sp, const int64_t n_slots, const int64_t o_base,
         const uint32_t *__restrict__ oid_of_slot,
         const float *__restrict__ cost_all, const float *__restrict__ ssd_all, const int ssd_stride,
         unsigned long long *__restrict__ keys)
{
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = slot < n_slots;
  const float *cost = cost_all + (live ? slot : 0) * sp.n_max;
  const float *ssd = ssd_all + (live ? slot : 0) * ssd_stride;
  const int stop = live ? first_break(sp, cost, ssd) : 0;
  const unsigned long long oid =
      !live ? 0ull : oid_of_slot ? (unsigned long long)oid_of_slot[slot] : (unsigned long long)(o_base + slot);
  // minimum over the 64 slots of a wave first, then one atomic per wave and n
  for (int n = (sp.n_min > 2 ? sp.n_min : 2); n <= sp.n_max; ++n) {
    unsigned long long key = ~0ull;
    if (n < stop) key = ((unsigned long long)__float_as_uint(ssd[n - sp.n_min]) << 32) | oid;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long other = __shfl_xor(key, d);
      key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0 && key != ~0ull) atomicMin(&keys[n - sp.n_min], key);
  }
}

// One more pass over the orientations for the alignments of rank r
// (`-alt-r`, MultiPass, search_or_dyn3d.cc:2095-2150): the running per-N minimum
// again, but only over alignments that share at most max_common matched pairs
// (NumMatchesInBothAssumingMonotonicity, lib/pair_alignment/pair_alignment.cc:
// 2529-2555) with the alignment every better rank holds for that N.
// rank_pairs = [r][range_n][n_max].  One workgroup per slot, one thread per N.
__global__ void __launch_bounds__(256)
k_reduce_dissimilar(const SearchParams sp, const int n_max, const int64_t n_slots,
                    const uint32_t *__restrict__ oid_of_slot, const float *__restrict__ cost_all,
                    const float *__restrict__ ssd_all, const short2 *__restrict__ pairs_all, const int r,
                    const short2 *__restrict__ rank_pairs, const int max_common,
                    unsigned long long *__restrict__ keys)
{
  __shared__ int s_stop;
  const int64_t slot = blockIdx.x;
  if (slot >= n_slots) return;
  if (threadIdx.x == 0) s_stop = first_break(sp, cost_all + slot * sp.n_max, ssd_all + slot * sp.range_n);
  __syncthreads();
  const int stop = s_stop;
  const unsigned long long oid = oid_of_slot[slot];
  for (int k = threadIdx.x; k < sp.range_n; k += 256) {
    const int n = sp.n_min + k;
    if (n < 2 || n >= stop) continue;
    const unsigned long long key =
        ((unsigned long long)__float_as_uint(ssd_all[slot * sp.range_n + k]) << 32) | oid;
    // keys only ever decrease: a candidate that does not beat the value seen
    // now cannot win later either (:2104 tests the RMSD before the pairs)
    if (key >= __atomic_load_n(&keys[k], __ATOMIC_RELAXED)) continue;
    const short2 *mine = pairs_all + (slot * sp.range_n + k) * n_max;
    bool dissimilar = true;
    for (int ra = 0; ra < r && dissimilar; ++ra) {
      const short2 *theirs = rank_pairs + ((int64_t)ra * sp.range_n + k) * n_max;
      int common = 0, n1 = 0;
      for (int n2 = 0; n2 < n; ++n2) {
        const short2 b = theirs[n2];
        while (n1 < n && mine[n1].x < b.x) ++n1;
        if (n1 < n && mine[n1].x == b.x && mine[n1].y == b.y) ++common;
      }
      dissimilar = common <= max_common;
    }
    if (dissimilar) atomicMin(&keys[k], key);
  }
}

// Copy the winners' records into the result tables.  slot_of_k[k] = slot of
// the export batch that holds entry k's winning orientation, -1 = leave alone.
__global__ void __launch_bounds__(64)
k_pick(const SearchParams sp, const int n_max, const int32_t *__restrict__ slot_of_k,
       const unsigned long long *__restrict__ keys, const float *__restrict__ ssd_all,
       const float *__restrict__ xf_all, const short2 *__restrict__ pairs_all,
       float *__restrict__ best_ssd, float *__restrict__ best_xf,
       short2 *__restrict__ best_pairs, long long *__restrict__ winner)
{
  const int k = blockIdx.x;
  const int slot = slot_of_k[k];
  if (slot < 0) return;
  const int64_t rec = (int64_t)slot * sp.range_n + k;
  if (threadIdx.x == 0) {
    best_ssd[k] = ssd_all[rec];
    winner[k] = (long long)(keys[k] & 0xffffffffull);
  }
  if (threadIdx.x < 12) best_xf[12 * k + threadIdx.x] = xf_all[12 * rec + threadIdx.x];
  const int n = sp.n_min + k;
  for (int m = threadIdx.x; m < n; m += 64)
    best_pairs[(int64_t)k * n_max + m] = pairs_all[rec * n_max + m];
}

// Refine: transform that minimises the RMSD of the current best alignment with
// n = n_min + k matches (search_or_dyn3d.cc:2401-2407).  One thread.
__global__ void k_refit_entry(const SearchParams sp, const int n_max, const int k,
                              const float *__restrict__ xyzA, const float *__restrict__ xyzB,
                              const short2 *__restrict__ best_pairs, float *__restrict__ og_out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int n = sp.n_min + k;
  const short2 *pairs = best_pairs + (int64_t)k * n_max;
  fit::Superimpose S;
  S.begin();
  for (int m = 0; m < n; ++m)
    for (int q = 0; q < sp.nbb; ++q)
      S.sum(xyzA + 3 * ((size_t)pairs[m].x * sp.nbb + q), xyzB + 3 * ((size_t)pairs[m].y * sp.nbb + q));
  S.centre(n * sp.nbb);
  for (int m = 0; m < n; ++m)
    for (int q = 0; q < sp.nbb; ++q)
      S.accumulate(xyzA + 3 * ((size_t)pairs[m].x * sp.nbb + q), xyzB + 3 * ((size_t)pairs[m].y * sp.nbb + q));
  float xf[12];
  S.finish(n * sp.nbb, xf);
  for (int q = 0; q < 12; ++q) og_out[q] = xf[q];
}

// Refine: fold the alignments found at the refined orientation (slot 0 of the
// batch buffers) into the result tables (search_or_dyn3d.cc:2432-2629).
// One workgroup.  need[] is aNeed2Refine; counters[0] += record updates.
__global__ void __launch_bounds__(256)
k_refine_apply(const SearchParams sp, const int n_max, const int k_self, const long long tag,
               const float *__restrict__ cost, const float *__restrict__ ssd,
               const float *__restrict__ xf, const short2 *__restrict__ pairs,
               float *__restrict__ best_ssd, float *__restrict__ best_xf,
               short2 *__restrict__ best_pairs, long long *__restrict__ winner,
               unsigned char *__restrict__ need, long long *__restrict__ counters)
{
  __shared__ int s_stop, s_changed, s_updates;
  if (threadIdx.x == 0) {
    s_stop = first_break(sp, cost, ssd);
    s_changed = 0;
    s_updates = 0;
  }
  __syncthreads();
  const int stop = s_stop;
  for (int k = threadIdx.x; k < sp.range_n; k += 256) {
    const int m = sp.n_min + k;
    if (m < 2 || m >= stop) continue;
    if (!(ssd[k] < best_ssd[k])) continue;                 // :2515
    const short2 *src = pairs + (int64_t)k * n_max;
    short2 *dst = best_pairs + (int64_t)k * n_max;
    if (k == k_self) {                                     // :2619 oldNthAlignment == new?
      int diff = 0;
      for (int q = 0; q < m; ++q) diff |= (src[q].x != dst[q].x) | (src[q].y != dst[q].y);
      if (diff) s_changed = 1;
    } else {
      need[k] = 1;                                         // :2544
    }
    best_ssd[k] = ssd[k];
    winner[k] = tag;
    for (int q = 0; q < 12; ++q) best_xf[12 * k + q] = xf[12 * k + q];
    for (int q = 0; q < m; ++q) dst[q] = src[q];
    atomicAdd(&s_updates, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    need[k_self] = (unsigned char)s_changed;               // :2619-2629
    counters[0] += s_updates;
  }
}

// ---- refinement in rounds ---------------------------------------------------
// The refinement loop is serial: iteration t re-solves the DP at the orientation
// that best fits the CURRENT best alignment of n_t = median(need).  What an
// iteration computes, though, depends only on that alignment.  A round therefore
// evaluates, in one batch, the DP of every n that still needs refining (slot s
// holds entry klist[s]), and k_refine_round then replays the reference's loop on
// the device, in the reference's order, consuming batch results for as long as
// the entry it has to refine next has not been changed since the batch was
// computed ("stale").  Results are identical to the one-at-a-time loop.

struct RefineCtrl {
  long long it;          // the reference's num_iterations (starts at 1)
  long long updates;     // num_new_solutions
  long long max_iters;   // -1 = until convergence
  int done;              // loop has ended (2 = a round made no progress: internal error)
  int applied;           // iterations consumed in the last round
  // device-driven rounds: the round kernel itself chooses what the next round prepares
  int count;             // entries prepared for the coming round (klist[0..count))
  int spec, spec_min, spec_max;   // how many to prepare: adapts to how many get used
  int device_plan;       // 0: the host fills klist / slot_of_k
  int pad;
  long long dp_runs, rounds;
  // why rounds ended (diagnostics): the next entry was not among the prepared
  // ones / had been changed by the round / everything prepared was used
  long long stop_unprepared, stop_stale, stop_all_used;
  long long applied_hist[16];
};

__global__ void __launch_bounds__(64)
k_refit_entries(const SearchParams sp, const int n_max, const int n_list,
                const int32_t *__restrict__ klist, const float *__restrict__ xyzA,
                const float *__restrict__ xyzB, const short2 *__restrict__ best_pairs,
                float *__restrict__ og_out, const int *__restrict__ live_count)
{
  // one wave per entry (fit::wave_superimpose)
  if (live_count && (int)blockIdx.x >= *live_count) return;
  extern __shared__ __attribute__((aligned(16))) float rfe_lds[];
  float *chunk = rfe_lds;                               // [fit::WAVE_LDS_FLOATS]
  short2 *lpairs = (short2 *)(rfe_lds + fit::WAVE_LDS_FLOATS);   // [n_max]
  const int s = blockIdx.x;
  if (s >= n_list) return;
  const int lane = threadIdx.x;
  const int k = klist[s];
  const int n = sp.n_min + k;
  const short2 *pairs = best_pairs + (int64_t)k * n_max;
  for (int m = lane; m < n; m += 64) lpairs[m] = pairs[m];
  __syncthreads();
  float xf[12];
  fit::wave_superimpose(n, sp.nbb, lpairs, chunk, xyzA, xyzB, xf);
  if (lane < 12) {
    float v = xf[0];
#pragma unroll
    for (int q = 1; q < 12; ++q) v = (lane == q) ? xf[q] : v;
    og_out[12 * s + lane] = v;
  }
}

#define MINRMS_NEED_WORDS ((MINRMS_MAX_RES + 63) / 64)

// FindMedian over a set of integers kept as a bit set
// (lib/global_utils/simple_numeric_utils.h:350-376: the ceil(size/2)-th smallest
// member), by one wave: lane w < MINRMS_NEED_WORDS brings word w of the set (the
// other lanes 0).  Returns the member, -1 for the empty set; the same value in
// every lane.  No memory access: a lone thread walking the words in LDS took
// longer than everything else the round kernel does.
__device__ __forceinline__ int wave_median_of_set(const unsigned long long my_word, const int lane)
{
  static_assert(MINRMS_NEED_WORDS <= 16, "prefix sum within one row of 16 lanes");
  const int cnt = __popcll(my_word);
  int upto = cnt;                                       // members in words 0..lane: prefix sum by row shifts
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x111, 0xf, 0xf, true);   // row_shr:1, lanes without a source add 0
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x112, 0xf, 0xf, true);
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x114, 0xf, 0xf, true);
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x118, 0xf, 0xf, true);
  const int total = __builtin_amdgcn_readlane(upto, 15);
  if (total == 0) return -1;
  const int half = (total + 1) / 2;
  const unsigned long long reached = __ballot(lane < 16 && upto >= half);
  const int w = __ffsll((long long)reached) - 1;        // the word that holds the half-th member
  const int r = half - (__builtin_amdgcn_readlane(upto, w) - __builtin_amdgcn_readlane(cnt, w));   // its r-th member
  const unsigned long long word = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_word >> 32), w) << 32) |
                                  (unsigned)__builtin_amdgcn_readlane((int)my_word, w);
  const unsigned long long upto_bit = word & ((2ull << lane) - 1ull);          // bits 0..lane
  const unsigned long long hit = __ballot(((word >> lane) & 1ull) && __popcll(upto_bit) == r);
  return 64 * w + (__ffsll((long long)hit) - 1);
}

__global__ void __launch_bounds__(256)
k_refine_round(const SearchParams sp, const int n_max, const int32_t *slot_of_k,
               const float *__restrict__ cost_all, const float *__restrict__ ssd_all,
               const float *__restrict__ xf_all, const short2 *__restrict__ pairs_all,
               float *__restrict__ best_ssd, float *__restrict__ best_xf,
               short2 *__restrict__ best_pairs, long long *__restrict__ winner,
               unsigned char *__restrict__ need_g, RefineCtrl *__restrict__ ctrl,
               int32_t *slot_of_k_next, int32_t *klist_next)
{
  // slot_of_k and slot_of_k_next are the same array in the device-planned
  // rounds (the next plan overwrites the one just replayed, after a barrier):
  // neither may be declared __restrict__.
  //
  // One workgroup, many short phases: what it waits for is the latency of its
  // global loads, one after the other.  So the control block and the slot
  // table are copied to LDS once, every field is read and updated there and the
  // block is written back at the end; and the values a phase needs are loaded
  // together with those of the phase before.
  __shared__ unsigned long long need[MINRMS_NEED_WORDS], stale[MINRMS_NEED_WORDS];
  __shared__ short upd_list[MINRMS_MAX_RES];          // entries the table at hand improves
  __shared__ int32_t slot_lds[MINRMS_MAX_RES];
  __shared__ float best_lds[MINRMS_MAX_RES];
  __shared__ RefineCtrl c;
  __shared__ int s_k, s_stop, s_changed, s_updates, s_go, s_nupd;
  __shared__ long long s_it;
  const int tid = threadIdx.x;
  constexpr int PER_THREAD = (MINRMS_MAX_RES + 255) / 256;
  static_assert(sizeof(RefineCtrl) % 4 == 0, "copied as words");
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long rs0 = MINRMS_TSTAMP();
#endif
  unsigned char need_in[PER_THREAD];
#pragma unroll
  for (int u = 0; u < PER_THREAD; ++u) {
    const int k = tid + 256 * u;
    need_in[u] = k < sp.range_n ? need_g[k] : (unsigned char)0;
  }
  for (int q = tid; q < (int)(sizeof(RefineCtrl) / 4); q += 256) ((int *)&c)[q] = ((const int *)ctrl)[q];
  for (int k = tid; k < sp.range_n; k += 256) { slot_lds[k] = slot_of_k[k]; best_lds[k] = best_ssd[k]; }
  if (tid < MINRMS_NEED_WORDS) { need[tid] = 0ull; stale[tid] = 0ull; }
  __syncthreads();
  if (c.done) return;
#pragma unroll
  for (int u = 0; u < PER_THREAD; ++u)
    if (need_in[u]) atomicOr(&need[(tid + 256 * u) >> 6], 1ull << (tid & 63));
  if (tid == 0) { s_it = c.it; s_updates = 0; }
  __syncthreads();
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long rs1 = MINRMS_TSTAMP();
#endif
  int applied = 0;
  for (;;) {
    if (tid < 64) {
      // the loop's next entry: FindMedian of the set still to refine
      const int k = wave_median_of_set(tid < MINRMS_NEED_WORDS ? need[tid] : 0ull, tid);
      if (tid == 0) {
        int go = 1;
        if (!(s_it <= c.max_iters || c.max_iters == -1)) { go = 0; c.done = 1; }
        else if (k < 0) { go = 0; c.done = 1; }
        else if (slot_lds[k] < 0) { go = 0; if (c.count > 0) { if (applied >= c.count) c.stop_all_used++; else c.stop_unprepared++; } }
        else if ((stale[k >> 6] >> (k & 63)) & 1ull) { go = 0; c.stop_stale++; }   // needs a new batch
        s_go = go; s_k = k; s_stop = sp.n_max + 1; s_changed = 0; s_nupd = 0;
      }
    }
    __syncthreads();
    if (!s_go) break;
    const int k_self = s_k;
    const int slot = slot_lds[k_self];
    const float *cost = cost_all + (int64_t)slot * sp.n_max;
    const float *ssd = ssd_all + (int64_t)slot * sp.range_n;
    const float *xf = xf_all + (int64_t)slot * sp.range_n * 12;
    const short2 *pairs = pairs_all + (int64_t)slot * sp.range_n * n_max;
    // this thread's entries, and the sums they hold now (used after the barrier)
    float new_ssd[PER_THREAD], old_ssd[PER_THREAD];
#pragma unroll
    for (int u = 0; u < PER_THREAD; ++u) {
      const int k = tid + 256 * u;
      new_ssd[u] = old_ssd[u] = 0.0f;
      if (k < sp.range_n) { new_ssd[u] = ssd[k]; old_ssd[u] = best_lds[k]; }
    }
    // first n at which the reference's m-loop breaks (:2448, :2502)
    for (int n = 2 + tid; n <= sp.n_max; n += 256) {
      bool brk = cost[n - 1] >= MINRMS_UPPER_BOUND;
      if (!brk && n >= sp.n_min && sp.max_rmsd != -1.0f)
        brk = ssd[n - sp.n_min] > (float)(n * sp.nbb) * (sp.max_rmsd * sp.max_rmsd);
      if (brk) atomicMin(&s_stop, n);
    }
    __syncthreads();
    const int stop = s_stop;
    const long long tag = -(2 + s_it);
    // which entries this table improves (:2515), one thread per entry
#pragma unroll
    for (int u = 0; u < PER_THREAD; ++u) {
      const int k = tid + 256 * u;
      const int m = sp.n_min + k;
      if (k >= sp.range_n || m < 2 || m >= stop) continue;
      if (!(new_ssd[u] < old_ssd[u])) continue;
      upd_list[atomicAdd(&s_nupd, 1)] = (short)k;
      if (k != k_self) atomicOr(&need[k >> 6], 1ull << (k & 63));   // :2544
      atomicOr(&stale[k >> 6], 1ull << (k & 63));
      best_ssd[k] = new_ssd[u];
      best_lds[k] = new_ssd[u];
      winner[k] = tag;
    }
    __syncthreads();
    // their transforms and alignments: all threads over the elements of the improved entries
    const int n_upd = s_nupd;
    for (int idx = tid; idx < n_upd * 12; idx += 256) {
      const int k = upd_list[idx / 12], q = idx % 12;
      best_xf[12 * k + q] = xf[12 * k + q];
    }
    for (int idx = tid; idx < n_upd * n_max; idx += 256) {
      const int e = idx / n_max, q = idx - e * n_max;
      const int k = upd_list[e];
      if (q >= sp.n_min + k) continue;
      const int64_t at = (int64_t)k * n_max + q;
      const short2 v = pairs[at], old = best_pairs[at];
      if (k == k_self && (v.x != old.x || v.y != old.y)) s_changed = 1;   // :2619
      best_pairs[at] = v;
    }
    if (tid == 0) s_updates += n_upd;
    __syncthreads();
    if (tid == 0) {
      if (s_changed) need[k_self >> 6] |= 1ull << (k_self & 63);   // :2619-2629
      else need[k_self >> 6] &= ~(1ull << (k_self & 63));
      s_it += 1;
    }
    ++applied;
    __threadfence_block();
    __syncthreads();
  }
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long rs2 = MINRMS_TSTAMP();
#endif
  for (int k = tid; k < sp.range_n; k += 256) need_g[k] = (unsigned char)((need[k >> 6] >> (k & 63)) & 1ull);
  if (tid == 0) {
    c.it = s_it;
    c.updates += s_updates;
    c.applied = applied;
    c.applied_hist[applied < 15 ? applied : 15]++;
  }
  if (c.device_plan) {                 // uniform: device_plan is never written here
    // Plan the next round: the entries the loop visits next if no table entry
    // changes any more (repeated median of the set still to refine).
    __syncthreads();
    for (int k = tid; k < sp.range_n; k += 256) slot_of_k_next[k] = -1;
    __syncthreads();
    if (tid < 64) {
      const int prepared = c.count;
      int spec = c.spec;
      int done = c.done;
      if (prepared > 0) {
        if (applied == 0 && !done) done = 2;
        if (applied >= prepared) spec = min(c.spec_max, spec * 2);
        else if (4 * applied < prepared) spec = max(c.spec_min, spec / 2);
      }
      int count = 0;
      if (!done) {
        unsigned long long my_word = tid < MINRMS_NEED_WORDS ? need[tid] : 0ull;
        while (count < spec) {
          const int k = wave_median_of_set(my_word, tid);
          if (k < 0) break;
          if (tid == 0) { slot_of_k_next[k] = count; klist_next[count] = k; }
          ++count;
          if (tid == (k >> 6)) my_word &= ~(1ull << (k & 63));
        }
      }
      if (tid == 0) {
        c.done = done;
        c.spec = spec;
        c.count = count;
        c.dp_runs += count;
        c.rounds += 1;
      }
    }
  }
  __syncthreads();
  for (int q = tid; q < (int)(sizeof(RefineCtrl) / 4); q += 256) ((int *)ctrl)[q] = ((const int *)&c)[q];
#ifdef MINRMS_COOP_STAMPS
  if (tid == 0 && applied > 0) {       // sums over the launches that replayed something
    const unsigned long long rs3 = MINRMS_TSTAMP();
    g_round_stamps[0] += rs1 - rs0; g_round_stamps[1] += rs2 - rs1; g_round_stamps[2] += rs3 - rs2;
    g_round_stamps[3] += 1; g_round_stamps[4] += applied; g_round_stamps[5] += c.count;
  }
#endif
}

// Fragment-pair superpositions (lib/or_gen/or_gen.cc:503-544): one thread per
// allowed (I,J); fixed = fragment of structure 1, moved = fragment of 2.
__global__ void __launch_bounds__(64)
k_fragment_fit(const int nbb, const int frag_len, const int64_t n_pairs,
               const int32_t *__restrict__ IJ, const float *__restrict__ xyzA,
               const float *__restrict__ xyzB, float *__restrict__ og, float *__restrict__ rmsd)
{
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= n_pairs) return;
  const float *fa = xyzA + 3 * (size_t)IJ[2 * t] * nbb;
  const float *fb = xyzB + 3 * (size_t)IJ[2 * t + 1] * nbb;
  const int npts = frag_len * nbb;
  fit::Superimpose S;
  S.begin();
  for (int a = 0; a < npts; ++a) S.sum(fa + 3 * a, fb + 3 * a);
  S.centre(npts);
  for (int a = 0; a < npts; ++a) S.accumulate(fa + 3 * a, fb + 3 * a);
  float xf[12];
  rmsd[t] = S.finish(npts, xf);
  for (int q = 0; q < 12; ++q) og[12 * t + q] = xf[q];
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
typedef void (*dp_kernel_fn)(const DpLayout, const float *, uint32_t *, float *, float *, float *,
                             const int, const int, const int32_t *);

template <int PB, bool BP>
static dp_kernel_fn pick_flags2(bool inject, bool has_next)
{
  if (inject) return has_next ? dp::k_dp_panel<PB, true, true, BP> : dp::k_dp_panel<PB, true, false, BP>;
  return has_next ? dp::k_dp_panel<PB, false, true, BP> : dp::k_dp_panel<PB, false, false, BP>;
}

template <int PB>
static dp_kernel_fn pick_flags(bool inject, bool has_next, bool bp)
{
  return bp ? pick_flags2<PB, true>(inject, has_next) : pick_flags2<PB, false>(inject, has_next);
}

static dp_kernel_fn pick_dp_kernel(int PB, bool inject, bool has_next, bool bp = true)
{
  switch (PB) {
    case 32: return pick_flags<32>(inject, has_next, bp);
    case 48: return pick_flags<48>(inject, has_next, bp);
    case 64: return pick_flags<64>(inject, has_next, bp);
    case 80: return pick_flags<80>(inject, has_next, bp);
    case 96: return pick_flags<96>(inject, has_next, bp);
    case 112: return pick_flags<112>(inject, has_next, bp);
    case 128: return pick_flags<128>(inject, has_next, bp);
    case 144: return pick_flags<144>(inject, has_next, bp);
    case 160: return pick_flags<160>(inject, has_next, bp);
    case 176: return pick_flags<176>(inject, has_next, bp);
    case 192: return pick_flags<192>(inject, has_next, bp);
  }
  return nullptr;
}

static void launch_cij(const SearchParams &sp, int64_t count, hipStream_t st, int64_t c_floats, const float *xyzA,
                       const float *xyzB, const float *og, float *C, float uniform_clip, const float *clip_of_slot,
                       const int32_t *og_row_of_slot, const int *live_count)
{
  if (sp.nbb == 1) {
    // a workgroup writes whole rows where they fit (consecutive rows are consecutive in memory)
    const int threads = std::min(256, (sp.N2 + 63) / 64 * 64);
    const dim3 grid((unsigned)((sp.N2 + threads - 1) / threads), (unsigned)((sp.N1 + MINRMS_CIJ_ROWS - 1) / MINRMS_CIJ_ROWS),
                    (unsigned)count);
    hipLaunchKernelGGL(k_cij_one_atom, grid, dim3(threads), 0, st, sp, c_floats, xyzA, xyzB, og, C, uniform_clip,
                       clip_of_slot, og_row_of_slot, live_count);
  } else {
    const dim3 grid((unsigned)((sp.N1 * sp.N2 + 255) / 256), (unsigned)count);
    hipLaunchKernelGGL(k_cij, grid, dim3(256), 0, st, sp, c_floats, xyzA, xyzB, og, C, uniform_clip, clip_of_slot,
                       og_row_of_slot, live_count);
  }
}

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

static int build_layout(const minrms_cfg &c, int max_pb, bool uniform_pb, DpLayout *out, int min_panels = 1)
{
  DpLayout L;
  memset(&L, 0, sizeof L);
  L.N1 = c.N1; L.N2 = c.N2; L.n_max = c.n_max;
  L.NB = (c.n_max + MINRMS_LAYERS_PER_BLOCK - 1) / MINRMS_LAYERS_PER_BLOCK;
  if (L.NB > MINRMS_MAX_BLOCKS) return -1;
  int64_t bp = 0, plane = 0, edge = 0;
  int prev_cols = 0;
  for (int b = 0; b < L.NB; ++b) {
    BlockDesc &d = L.blk[b];
    d.j0 = MINRMS_LAYERS_PER_BLOCK * b + 1;
    const int ncols = c.N2 - MINRMS_LAYERS_PER_BLOCK * b;
    d.nrows = c.N1 - MINRMS_LAYERS_PER_BLOCK * b;
    d.P = (ncols + max_pb - 1) / max_pb;
    if (!uniform_pb) {
      // Two panels of <= 96 columns run at 4-5 waves per SIMD instead of 2-3
      // (one row of costs per lane lives in PB registers): measured 8 % faster
      // at 150 columns.  Only when the split wastes at most one 16-column word.
      const int half = (int)round_up((ncols + 1) / 2, 16);
      if (d.P == 1 && ncols > 96 && 2 * half - ncols <= 16) d.P = 2;
      if (min_panels > d.P && ncols >= 32 * min_panels) d.P = min_panels;
    }
    d.PB = uniform_pb ? max_pb : std::max(32, (int)round_up((ncols + d.P - 1) / d.P, 16));   // narrowest kernel: 32
    d.W = d.P * d.PB / 16;
    d.bp_off = bp;
    bp += (int64_t)d.nrows * d.W * 64;
    if (b > 0) {
      // rows 0..nrows (producer writes one more than the consumer reads);
      // columns = producer cell index: consumer reads 62 .. P*PB+62
      d.plane_stride = (int)round_up(std::max(prev_cols, d.P * d.PB + 63), 4);
      d.plane_off = plane;
      plane += (int64_t)(d.nrows + 1) * d.plane_stride;
      plane = round_up(plane, 4);
    }
    if (d.P > 1) edge = std::max<int64_t>(edge, (int64_t)d.nrows * 64);
    prev_cols = d.P * d.PB;
  }
  L.bp_words = bp;
  L.plane_floats = plane;
  L.edge_floats = edge;
  // the scalar C stream of a row reads up to PB+1 floats past column j0 and
  // one row ahead, also of the last row: pad the table's tail
  L.c_floats = round_up((int64_t)c.N1 * c.N2 + c.N2 + 2 * MINRMS_MAX_PB + 64, 64);
  *out = L;
  return 0;
}

// engines alive in this process (a batch of pairs runs several at a time, one host thread each)
static std::atomic<int> live_engines{0};

struct minrms_engine {
  minrms_cfg cfg;
  SearchParams sp;
  DpLayout L;
  DpLayout Lc;                      // cooperative layout (valid if coop_pb != 0)
  DpLayout Lc_seg;                  // the same tables cut into the panels of the segment tasks (coop_seg_pb columns)
  int coop_seg_pb = 0;              // = coop_pb, or 8: 16-column tables worked on by 8-column tasks
  int coop_pb = 0;
  dp::CoopGroups coop_groups;       // how its tasks are spread over workgroups: whole layer blocks (throughput)
  dp::CoopGroups coop_seg;          // ... segments of blocks, a SIMD per task (latency); n_groups = 0: not used
  int64_t coop_seg_batch = 0;       // largest batch the segmented spread is used for
  unsigned long long *d_xedge = nullptr;   // [coop_seg_batch][coop_seg.xedge_granules] hand-over columns between segments
  int *d_gprog = nullptr;           // [coop_max_batch][total_tasks] row counters (several groups only)
  int coop_launches = 0;            // tags the row counters of each cooperative launch
  RefineCtrl *h_ctrl = nullptr;     // pinned: two snapshots of the refinement's control block
  hipEvent_t ctrl_event[2] = {nullptr, nullptr};
  int64_t coop_max_batch = 1024;
  int64_t wave_trace_max_chains = 32768;   // test hook: MINRMS_WAVE_TRACE_MAX
  int *d_err = nullptr;               // raised by a cooperative DP kernel whose bounded wait gave up
  bool device_failed = false;         // ... seen by the host: results are unusable until reset()
  unsigned char *d_gen = nullptr;     // staging of gen_orientations (allocated on first use)
  struct DevBlock { void *ptr; size_t bytes; const char *tag; size_t clean; };
  std::vector<DevBlock> dev_blocks;   // every device allocation of this engine
  hipStream_t stream = nullptr;
  int64_t batch_cap = 0;
  // inputs
  float *d_xyzA = nullptr, *d_xyzB = nullptr;
  // per-batch workspace
  float *d_og = nullptr, *d_C = nullptr, *d_planes = nullptr, *d_edges = nullptr;
  float *d_cost = nullptr, *d_ssd = nullptr, *d_xf = nullptr;
  uint32_t *d_bp = nullptr;
  short2 *d_pairs = nullptr;
  // results
  unsigned long long *d_keys = nullptr;
  float *d_best_ssd = nullptr, *d_best_xf = nullptr;
  short2 *d_best_pairs = nullptr;
  long long *d_winner = nullptr;
  int32_t *d_slot_of_k = nullptr;
  unsigned char *d_need = nullptr;
  long long *d_counters = nullptr;
  int32_t *d_klist = nullptr;
  RefineCtrl *d_ctrl = nullptr;
  // -of orientation filter.  DistanceMetric's cutoff is process-wide state in
  // the reference and every GoodOrientation call re-installs it through a
  // sqrt/square round trip (orientation_filter.cc:96,127), so the cutoff an
  // orientation is evaluated with depends on how many filter calls came first.
  bool of = false;
  nwf::Params nwp;
  int64_t filter_cap = 0;           // orientations per filter launch
  float *d_fog = nullptr;           // [filter_cap][12]
  unsigned char *d_pass = nullptr;  // [filter_cap]
  float *d_clip = nullptr;          // [batch_cap] cutoff of each slot
  uint32_t *d_oids = nullptr;       // [batch_cap] o_id of each slot
  int32_t *d_stop = nullptr;        // [batch_cap] n at which the n-loop leaves each slot (sweep with early termination)
  std::vector<float> clip_seq;      // cutoff after 0,1,2,... filter calls
  int64_t clip_cycle_from = 0, clip_cycle_len = 1;
  int64_t of_calls = 0;             // filter calls made so far
  int64_t of_origin_calls = 0, of_origin_start = 0;   // of the last search()
  float clip_now = 0.0f;            // cutoff in force (what Refine sees)
  std::vector<unsigned char> h_pass;   // filter verdicts of the last search() range
  // alternate solutions by orientation (cfg.num_solutions > 1)
  int num_solutions = 1;
  std::vector<float> h_xyzA, h_xyzB;
  std::vector<std::vector<alt::Candidate>> alt_files;   // per n: what the reference spills to "<n>.tmp"
  std::vector<float> alt_running_best;                   // best ssd so far per n, as FirstPass sees it
  AltRecord *d_alt_rec = nullptr;                        // survivors of k_alt_filter, [alt_rec_cap]
  int64_t alt_rec_cap = 0;
  float *d_alt_best = nullptr;                           // [range_n] running best handed to the filter
  float *d_alt_chunk = nullptr;                          // [chunks of slots][range_n] chunk minima, then the best before each chunk
  int32_t *d_alt_offsets = nullptr;                      // [range_n][chunks] survivors per (n, chunk), then where they go
  int *d_alt_count = nullptr;
  // alternates by matched pairs (`-alt-r`): one more sweep per rank
  bool alt_by_pairs = false;
  std::vector<unsigned char> alt_seen;                   // per o_id: 0 not searched, 1 failed the filter, 2 evaluated
  short2 *d_rank_pairs = nullptr;                        // [num_solutions-1][range_n][n_max]
  unsigned long long *d_rank_keys = nullptr;             // [range_n] keys of the rank being searched
  alt::RankTables ranks;                                 // all ranks on the host (rank 0 mirrors the device)
  bool ranks_current = false;
  std::vector<float> h_og;          // host copy of the orientation list seen by search()
  bool tables_current = false;      // result tables reflect the keys
  bool searched = false;
  // timing of the DP launches: events bracket every (layer block, panel)
  // launch; they are read back lazily, after the stream has been synchronised
  bool timing = false;
  std::vector<hipEvent_t> ev_pool;          // all events ever created
  size_t ev_used = 0;                       // events handed out since the last drain
  struct Stamp { size_t first; int nb; int blocks[MINRMS_MAX_BLOCKS * 8]; int64_t count; };
  std::vector<Stamp> stamps;
  double blk_ms[MINRMS_MAX_BLOCKS] = {0};
  int64_t blk_launches[MINRMS_MAX_BLOCKS] = {0};
  int64_t blk_orients[MINRMS_MAX_BLOCKS] = {0};
};

static void free_all(minrms_engine *e)
{
  for (const minrms_engine::DevBlock &b : e->dev_blocks) pool::park(e->cfg.device, b.tag, b.ptr, b.bytes, b.clean);
  e->dev_blocks.clear();
  if (e->d_alt_best) { (void)hipFree(e->d_alt_rec); (void)hipFree(e->d_alt_best); (void)hipFree(e->d_alt_chunk); (void)hipFree(e->d_alt_offsets); (void)hipFree(e->d_alt_count); e->d_alt_best = nullptr; }
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->h_ctrl) {
    std::lock_guard<std::mutex> lock(pool::mu);
    pool::host_slots.push_back(pool::HostSlot{e->cfg.device, e->h_ctrl, {e->ctrl_event[0], e->ctrl_event[1]}});
    e->h_ctrl = nullptr;
  }
  if (e->stream) (void)hipStreamDestroy(e->stream);
}

extern "C" int64_t minrms_engine_cells_per_orientation(const minrms_cfg *c)
{
  if (!c) return 0;
  const int wA = c->N1 - c->n_min + 1, wB = c->N2 - c->n_min + 1;
  int64_t cells = 0;
  for (int n = 1; n <= c->n_max; ++n) {
    const int imax = std::min(c->N1, n + wA - 1), jmax = std::min(c->N2, n + wB - 1);
    cells += (int64_t)(imax - n + 1) * (jmax - n + 1);
  }
  return cells;
}

// Cutoff after `calls` GoodOrientation calls.
static float clip_after(const minrms_engine *e, int64_t calls)
{
  if (calls < 0) calls = 0;
  const int64_t n = (int64_t)e->clip_seq.size();
  if (calls < n) return e->clip_seq[(size_t)calls];
  return e->clip_seq[(size_t)(e->clip_cycle_from + (calls - e->clip_cycle_from) % e->clip_cycle_len)];
}

// distance_metric.cc:76-100: Get (sqrt) then Set (square) of the cutoff
static float clip_round_trip(float clip, int nbb)
{
  if (clip == MINRMS_UPPER_BOUND) return clip;
  const float phys = sqrtf(clip / (float)nbb);
  if (phys == -1.0f) return MINRMS_UPPER_BOUND;
  return (phys * phys) * (float)nbb;
}

static void build_clip_sequence(minrms_engine *e)
{
  e->clip_seq.assign(1, e->sp.clip);
  e->clip_cycle_from = 0;
  e->clip_cycle_len = 1;
  if (!e->of) return;
  for (int it = 0; it < 64; ++it) {
    const float next = clip_round_trip(e->clip_seq.back(), e->sp.nbb);
    for (size_t q = 0; q < e->clip_seq.size(); ++q)
      if (e->clip_seq[q] == next) {           // entered a cycle (normally a fixed point)
        e->clip_cycle_from = (int64_t)q;
        e->clip_cycle_len = (int64_t)(e->clip_seq.size() - q);
        return;
      }
    e->clip_seq.push_back(next);
  }
  // no repeat within 64 steps (not seen in practice): freeze the last value
  e->clip_cycle_from = (int64_t)e->clip_seq.size() - 1;
  e->clip_cycle_len = 1;
}

static int reset_results(minrms_engine *e)
{
  const int rn = e->sp.range_n;
  HIP_TRY(hipMemsetAsync(e->d_keys, 0xFF, sizeof(unsigned long long) * rn, e->stream));
  std::vector<float> ssd(rn, MINRMS_UPPER_BOUND), xf((size_t)rn * 12, 0.0f);
  std::vector<long long> win(rn, -1);
  for (int k = 0; k < rn; ++k) xf[12 * k] = xf[12 * k + 5] = xf[12 * k + 10] = 1.0f;
  HIP_TRY(hipMemcpyAsync(e->d_best_ssd, ssd.data(), 4 * rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_best_xf, xf.data(), 48 * rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_winner, win.data(), 8 * rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_best_pairs, 0xFF, sizeof(short2) * (size_t)rn * e->cfg.n_max, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return 0;
}

static void coop_block_first_tasks(const DpLayout &Lc, dp::CoopGroups *G)
{
  int t = 0;
  for (int b = 0; b < Lc.NB; ++b) { G->block_first_task[b] = (int16_t)t; t += Lc.blk[b].P; }
  G->block_first_task[Lc.NB] = (int16_t)t;
  G->total_tasks = t;
}

// Workgroups of whole layer blocks, at most `cap` tasks each.
static bool groups_of_whole_blocks(const DpLayout &Lc, int cap, dp::CoopGroups *G)
{
  memset(G, 0, sizeof *G);
  coop_block_first_tasks(Lc, G);
  int g = 0, in_group = 0;
  for (int b = 0; b < Lc.NB; ++b) {
    if (b > 0 && in_group + Lc.blk[b].P > cap) { G->max_tasks = std::max(G->max_tasks, in_group); ++g; in_group = 0; }
    if (g + 1 >= MINRMS_COOP_MAX_GROUPS) return false;
    if (in_group == 0) {
      G->blk[g] = (int16_t)b; G->pan[g] = 0; G->whole[g] = 1;
      G->first_task[g] = G->block_first_task[b];
      G->xin_off[g] = G->xout_off[g] = -1;
    }
    in_group += Lc.blk[b].P;
    G->ntask[g] = (int16_t)in_group;
  }
  G->max_tasks = std::max(G->max_tasks, in_group);
  G->n_groups = g + 1;
  G->blk[G->n_groups] = (int16_t)Lc.NB;              // sentinel: "first block of the group after the last"
  return G->max_tasks <= MINRMS_COOP_MAX_TASKS;
}

// Every layer block cut into segments of at most `seg` consecutive panels (sizes
// as equal as possible), each a workgroup of its own.
static bool groups_of_segments(const DpLayout &Lc, int seg, dp::CoopGroups *G)
{
  memset(G, 0, sizeof *G);
  coop_block_first_tasks(Lc, G);
  const int pb = Lc.blk[0].PB;
  seg = std::min(seg, 191 / pb);              // the plane-fetch wave takes at most 192 columns
  if (seg < 1) return false;
  int g = 0;
  int64_t granules = 0;
  // planes between blocks: [rows of block b + 1][columns of block b-1]
  for (int b = 0; b <= Lc.NB; ++b) { G->plane_goff[b] = -1; G->plane_gstride[b] = 0; }
  for (int b = 1; b < Lc.NB; ++b) {
    G->plane_goff[b] = (int32_t)granules;
    G->plane_gstride[b] = Lc.blk[b - 1].P * pb;
    granules += (int64_t)(Lc.blk[b].nrows + 1) * G->plane_gstride[b];
  }
  for (int b = 0; b < Lc.NB; ++b) {
    const int P = Lc.blk[b].P, n_seg = (P + seg - 1) / seg;
    int p = 0;
    for (int q = 0; q < n_seg; ++q) {
      if (g + 1 >= MINRMS_COOP_MAX_GROUPS) return false;
      const int size = P / n_seg + (q < P % n_seg ? 1 : 0);
      G->blk[g] = (int16_t)b; G->pan[g] = (int16_t)p; G->ntask[g] = (int16_t)size;
      G->whole[g] = n_seg == 1;
      G->first_task[g] = (int16_t)(G->block_first_task[b] + p);
      G->xin_off[g] = -1; G->xout_off[g] = -1;
      if (q > 0) G->xin_off[g] = G->xout_off[g - 1];
      if (q + 1 < n_seg) { G->xout_off[g] = (int32_t)granules; granules += (int64_t)Lc.blk[b].nrows * 64; }
      G->max_tasks = std::max(G->max_tasks, size);
      p += size;
      ++g;
    }
  }
  G->n_groups = g;
  G->blk[g] = (int16_t)Lc.NB;
  G->xedge_granules = granules;
  G->fetch_waves = 3;                         // the edge-fetch, the plane-fetch and the plane-publish wave
  // 16-column tasks run the hand-written row loop; MINRMS_COOP_ASM=0 (test hook) keeps the compiled one
  G->asm_rows = 1;
  if (const char *s = getenv("MINRMS_COOP_ASM")) G->asm_rows = atoi(s) != 0;
  return G->max_tasks + G->fetch_waves <= MINRMS_COOP_MAX_TASKS && granules < (1ll << 31);
}

// Everything of an engine that is decided on the host and needs no device
// memory: argument checks, search parameters, the two DP layouts.  `c` is the
// caller's configuration with n_max resolved.
static int configure(minrms_engine *e, minrms_cfg c)
{
  const int max_matches = std::min(c.N1, c.N2);
  if (c.N1 < 1 || c.N2 < 1 || c.num_bb < 1)
    return fail(MINRMS_E_ARG, "structures must have at least one residue and one backbone atom");
  if (c.n_max < 0) c.n_max = max_matches;    // NO_MAX_NUM_MATCHES
  // same test as Dyn3d::Alloc (lib/dyn/dyn3d.cc:459-467)
  if (c.n_min < 1 || c.n_min > max_matches || c.n_max < 1 || c.n_max > max_matches || c.n_max < c.n_min)
    return fail(MINRMS_E_ARG,
                "Minimum and/or maximum number of matches allowed does not make sense. The range of "
                "allowed matches must be >=1, and <=%d. Presently min_n = %d, max_n = %d",
                max_matches, c.n_min, c.n_max);
  if (c.N1 > MINRMS_MAX_RES || c.N2 > MINRMS_MAX_RES)
    return fail(MINRMS_E_ARG, "structures longer than %d residues are not supported", MINRMS_MAX_RES);
  if (c.N1 > 32767 || c.N2 > 32767) return fail(MINRMS_E_ARG, "residue index exceeds int16");
  if (c.of_enable) {
    // DynNW::Alloc (lib/dyn/dynNW.cc:79-85), minrms_parser.cc:1089-1123
    if (c.of_min_matches < 1 || c.of_min_matches > max_matches)
      return fail(MINRMS_E_ARG,
                  "-of's second parameter, (%d), must be > 0 and must not exceed the number of residues "
                  "in the smaller structure, (%d).", c.of_min_matches, max_matches);
    if (c.n_max < c.of_min_matches)
      return fail(MINRMS_E_ARG, "-of's second parameter, (%d), must not exceed n_max (%d).",
                  c.of_min_matches, c.n_max);
    // DynNW::Solve refuses a gap penalty <= 0 (lib/dyn/dynNW.cc:257-276)
    const float gap = (float)(0.5 * (double)(c.of_max_dist * c.of_max_dist) * (double)max_matches * (double)c.num_bb);
    if (!(gap > 0.0f))
      return fail(MINRMS_E_ARG, "the -of distance gives a gap penalty of %g: it must be positive", (double)gap);
  }

  if (c.num_solutions < 1) c.num_solutions = 1;
  if (c.num_solutions > 1) {
    // minrms_parser.cc:812-836.  With refine_while_searching the candidates are
    // compared by their refit transforms (-alt-o, ALT_METHOD_3D), without it by
    // the orientations they were found at (-alt-o-orig, ALT_METHOD_3D_ORIG): the
    // parser ties the two settings together (:946-984).
    if (!c.alt_by_pairs && c.alt_min_3d_difference < 0.0f)
      return fail(MINRMS_E_ARG, "The \"-alt-o\" second parameter (min_delta_orientation) must be at least 0.");
    if (c.alt_by_pairs && c.alt_max_pairs_common < 0)       // minrms_parser.cc:911-913
      return fail(MINRMS_E_ARG, "The \"-alt-r\" second parameter (max_pairs_in_common) must be at least 0.");
    if (c.alt_rmsd_tolerance <= 1.0f && c.alt_rmsd_tolerance != -1.0f)
      return fail(MINRMS_E_ARG, "The \"-alt-o\" third parameter (RMSD_tolerance) must be greater than 1.0.");
  }

  e->cfg = c;
  SearchParams &sp = e->sp;
  sp.N1 = c.N1; sp.N2 = c.N2; sp.nbb = c.num_bb; sp.n_min = c.n_min; sp.n_max = c.n_max;
  sp.range_n = c.n_max - c.n_min + 1;
  sp.max_rmsd = c.max_rmsd;
  sp.clip = (c.max_dist == -1.0f) ? MINRMS_UPPER_BOUND : (c.max_dist * c.max_dist) * (float)c.num_bb;
  sp.rws = c.refine_while_searching ? 1 : 0;
  e->of = c.of_enable != 0;
  if (e->of) {
    nwf::Params &fp = e->nwp;
    const float d = c.of_max_dist;
    fp.N1 = c.N1; fp.N2 = c.N2; fp.nbb = c.num_bb;
    fp.win_a = c.N1 - c.of_min_matches + 1;
    fp.win_b = c.N2 - c.of_min_matches + 1;
    fp.min_matches = c.of_min_matches;
    fp.clip = (d == -1.0f) ? MINRMS_UPPER_BOUND : (d * d) * (float)c.num_bb;
    fp.gap = (float)(0.5 * (double)(d * d) * (double)max_matches * (double)c.num_bb);
    fp.upper_bound = MINRMS_UPPER_BOUND;
  }
  build_clip_sequence(e);
  e->clip_now = sp.clip;
  e->num_solutions = c.num_solutions;
  e->alt_by_pairs = c.num_solutions > 1 && c.alt_by_pairs != 0;

  // panels wider than 160 columns leave two waves per SIMD: measured 20 % slower
  // per cell than three panels of 112-128 (block of 337 columns, 400x400)
  int max_pb = 160;
  if (const char *s = getenv("MINRMS_MAX_PB")) {    // test hook: force narrow panels
    const int v = atoi(s);
    if (v >= 32 && v <= MINRMS_MAX_PB && v % 32 == 0) max_pb = v;
  }
  if (const char *s = getenv("MINRMS_WAVE_TRACE_MAX")) e->wave_trace_max_chains = atoll(s);
  int min_panels = 1;
  if (const char *s = getenv("MINRMS_SWEEP_PANELS")) min_panels = std::max(1, atoi(s));
  if (build_layout(c, max_pb, false, &e->L, min_panels) != 0) return fail(MINRMS_E_ARG, "layout failed");
  // cooperative (low-latency) layout: uniform narrow panels, the tasks of an
  // orientation in a few workgroups of at most 16 waves
  e->coop_pb = 0;
  memset(&e->coop_groups, 0, sizeof e->coop_groups);
  if (!getenv("MINRMS_NO_COOP")) {
    // Panel width and workgroups.  Narrow panels shorten the row of cells each
    // wave has to finish before its neighbour can go on, and the first layer
    // block, the longest chain of rows, does best in a workgroup (= on a CU) of
    // its own: measured refinement times, one workgroup of 32-column panels vs.
    // this rule: 37 -> 34 ms at 150x150, 94 -> 82 ms at 200x200, and 201 (64
    // columns) -> 140 ms at 250x250.  So: the narrowest width that leaves the
    // first block at most 13 tasks (16 at width 64, and with segments, below),
    // the first block alone, the following blocks together up to as many tasks
    // as the first has.
    int forced_pb = 0, forced_cap = 0;                  // test hooks
    if (const char *s = getenv("MINRMS_COOP_PB")) forced_pb = atoi(s);
    if (const char *s = getenv("MINRMS_COOP_GROUP_TASKS")) forced_cap = std::max(1, atoi(s));
    for (int pb : {16, 32, 64}) {
      if (forced_pb && pb != forced_pb) continue;
      DpLayout Lc;
      if (build_layout(c, pb, true, &Lc) != 0) continue;
      int tasks = 0, widest = 0;
      for (int b = 0; b < Lc.NB; ++b) { tasks += Lc.blk[b].P; widest = std::max(widest, Lc.blk[b].P); }
      if (widest > MINRMS_COOP_MAX_TASKS) continue;
      // With segments (the refinement rounds: every task a SIMD of its own, at most 4 tasks per
      // workgroup) the narrowest panels win as long as the block fits the 16 progress counters:
      // refinement at 250x250 94 -> 60 ms with 16 tasks of 16 columns instead of 8 of 32.  Only the
      // few large batches (the winners of a search) then run 16 tasks on one CU.
      const char *seg_env = getenv("MINRMS_COOP_SEG_TASKS");
      const bool segments = !seg_env || atoi(seg_env) > 0;
      if (!forced_pb && widest > ((pb == 64 || segments) ? MINRMS_COOP_MAX_TASKS : 13)) continue;
      const int cap = forced_cap ? std::min(forced_cap, MINRMS_COOP_MAX_TASKS) : std::max(widest, 8);
      if (!groups_of_whole_blocks(Lc, cap, &e->coop_groups)) continue;
      // A SIMD per task: segments of at most 4 panels, for batches small enough
      // that the extra workgroups find idle CUs (the rounds of a refinement).
      int seg_tasks = 4;
      e->coop_seg_batch = 40;
      if (const char *s = getenv("MINRMS_COOP_SEG_TASKS")) seg_tasks = atoi(s);          // test hooks; 0 = off
      if (const char *s = getenv("MINRMS_COOP_SEG_MAX_BATCH")) e->coop_seg_batch = std::max(0, atoi(s));
      memset(&e->coop_seg, 0, sizeof e->coop_seg);
      // A task's row is paced by the number of its instructions (a lone wave issues one per 4-5
      // cycles): segment tasks narrower than the layout's panels have fewer cells per row.  They
      // work on the same layout (same back-pointer words -- an 8-column task writes half a word per
      // row -- and the same distance tables), so everything downstream is unchanged.  Measured
      // refinement times, narrow against full-width tasks: 17.0 -> 16.4 ms at 150x150, 57.8 ->
      // 51.6 ms at 250x250.  MINRMS_COOP_SEG_HALF=0 (test hook): full-width tasks.
      e->Lc_seg = Lc;
      e->coop_seg_pb = pb;
      {
        const char *sh = getenv("MINRMS_COOP_SEG_HALF");
        const bool half = !(sh && atoi(sh) == 0);
        int most = 40;                                   // tasks of the first block
        if (const char *sm = getenv("MINRMS_COOP_SEG_HALF_MAX_TASKS")) most = atoi(sm);   // tuning hook
        // the narrowest of 8, 16 and 32 columns that keeps the first block within `most` tasks (320, 640 columns)
        // (refinement at 300x300: 91.1 ms with 32-column tasks, 70.4 with 16, 67.1 with 8; at 400x400 163.6, 131.4, 132.1)
        int w = (c.N2 + 7) / 8 <= most ? 8 : (c.N2 + 15) / 16 <= most ? 16 : 32;
        if (const char *sw = getenv("MINRMS_COOP_SEG_PB")) w = atoi(sw);                  // tuning hook: 8, 16 or 32
        if (half && (w == 8 || w == 16 || w == 32) && w < pb && (c.N2 + w - 1) / w <= most) {
          for (int b = 0; b < Lc.NB; ++b) {
            e->Lc_seg.blk[b].PB = w;
            // never more columns than the full-width panels have
            e->Lc_seg.blk[b].P = (c.N2 - MINRMS_LAYERS_PER_BLOCK * b + w - 1) / w;
          }
          e->coop_seg_pb = w;
        }
      }
      if (seg_tasks < 1 || seg_tasks >= MINRMS_COOP_MAX_TASKS || e->coop_seg_batch < 1 ||
          !groups_of_segments(e->Lc_seg, seg_tasks, &e->coop_seg))
        e->coop_seg.n_groups = 0;
      if (e->coop_seg.n_groups == 0 || !e->coop_seg.asm_rows) {
        // no segments, or the compiled row loop (test hook): back to full-width tasks (8-column
        // tasks exist in assembly only, and the two loops are compared on the same tasks)
        if (e->coop_seg_pb != pb) {
          e->Lc_seg = Lc;
          e->coop_seg_pb = pb;
          memset(&e->coop_seg, 0, sizeof e->coop_seg);
          if (!(seg_tasks < 1 || seg_tasks >= MINRMS_COOP_MAX_TASKS || e->coop_seg_batch < 1) &&
              !groups_of_segments(Lc, seg_tasks, &e->coop_seg))
            e->coop_seg.n_groups = 0;
        }
      }
      for (dp::CoopGroups *G : {&e->coop_groups, &e->coop_seg}) {
        G->spin_limit = MINRMS_COOP_SPIN_LIMIT;
        // test hook: a tiny limit makes waits give up early, a negative one starts
        // every workgroup in the given-up state (every wave still runs to the end
        // of its rows; the host has to notice and discard the batch)
        if (const char *s = getenv("MINRMS_COOP_SPIN_LIMIT")) G->spin_limit = atoi(s);
      }
      e->Lc = Lc;
      e->coop_pb = pb;
      break;
    }
  }
  return 0;
}

// The cooperative kernel of several workgroups asks for more LDS than half a CU
// holds (one workgroup per CU): raise the limit of its instantiations.  Device
// must be current.
static int allow_large_lds()
{
  const void *kernels[7] = {(const void *)dp::k_dp_coop<16, false>, (const void *)dp::k_dp_coop<32, false>,
                            (const void *)dp::k_dp_coop<64, false>, (const void *)dp::k_dp_coop<16, true>,
                            (const void *)dp::k_dp_coop<32, true>, (const void *)dp::k_dp_coop<64, true>,
                            (const void *)dp::k_dp_coop<8, true>};
  for (const void *k : kernels)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return fail(MINRMS_E_DEVICE, "hipFuncSetAttribute(dynamic LDS) failed");
  return 0;
}

// Device bytes one orientation of a batch needs (both layouts share the buffers).
static int64_t workspace_per_orientation(const minrms_engine *e)
{
  const DpLayout &L = e->L;
  const int rn = e->sp.range_n;
  const int64_t bp_words = std::max(L.bp_words, e->coop_pb ? e->Lc.bp_words : 0);
  const int64_t plane_floats = std::max(L.plane_floats, e->coop_pb ? e->Lc.plane_floats : 0);
  return 4 * (L.c_floats + bp_words + plane_floats + L.edge_floats + L.n_max + rn + 12 * (int64_t)rn + 12) +
         (int64_t)sizeof(short2) * rn * L.n_max;
}

extern "C" int minrms_engine_create(const minrms_cfg *cfg, const float *xyzA, const float *xyzB,
                                    minrms_engine **out)
{
  if (!cfg || !xyzA || !xyzB || !out) return fail(MINRMS_E_ARG, "null argument");
  *out = nullptr;
  minrms_engine *e = new minrms_engine();
  { int rc = configure(e, *cfg); if (rc) { delete e; return rc; } }
  const minrms_cfg c = e->cfg;
  SearchParams &sp = e->sp;

  int ndev = 0;
  { hipError_t he0 = hipGetDeviceCount(&ndev); if (he0 != hipSuccess) { delete e; return fail(MINRMS_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(he0)); } }
  if (c.device < 0 || c.device >= ndev) { delete e; return fail(MINRMS_E_DEVICE, "no HIP device %d", c.device); }
  { hipError_t he0 = hipSetDevice(c.device); if (he0 != hipSuccess) { delete e; return fail(MINRMS_E_DEVICE, "hipSetDevice: %s", hipGetErrorString(he0)); } }

  e->h_xyzA.assign(xyzA, xyzA + (size_t)c.N1 * c.num_bb * 3);
  e->h_xyzB.assign(xyzB, xyzB + (size_t)c.N2 * c.num_bb * 3);
  if (e->num_solutions > 1) {
    e->alt_files.assign(sp.range_n, std::vector<alt::Candidate>());
    e->alt_running_best.assign(sp.range_n, MINRMS_UPPER_BOUND);
  }
  if (e->coop_pb != 0 && (e->coop_groups.n_groups > 1 || e->coop_seg.n_groups > 1)) {
    int rc = allow_large_lds();
    if (rc) { delete e; return rc; }
  }
  const DpLayout &L = e->L;
  const int rn = sp.range_n;

  // workspace per orientation
  const int64_t bp_words = std::max(L.bp_words, e->coop_pb ? e->Lc.bp_words : 0);
  const int64_t plane_floats = std::max(L.plane_floats, e->coop_pb ? e->Lc.plane_floats : 0);
  const int64_t per_o = workspace_per_orientation(e);
  // Default: a third of the 288 GB of an MI355X.  Larger batches pay off up to
  // there: every launch of a batch ends in a tail of partly filled CUs, and a
  // batch is ~10 ordered launches (400x400: sweep 1.75 s at 24 GB, 1.60 s at 96 GB).
  int64_t budget = c.workspace_bytes > 0 ? c.workspace_bytes : (int64_t)96 << 30;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
    budget = std::min<int64_t>(budget, (int64_t)((free_b + pool::parked_bytes(c.device)) * 0.8));
  int64_t cap = budget / per_o;
  cap = std::min<int64_t>(cap, 32768);
  if (cap < 1) { delete e; return fail(MINRMS_E_DEVICE, "workspace budget too small: %lld bytes per orientation", (long long)per_o); }
  e->batch_cap = cap;

#define ALLOC(ptr, bytes)                                                              \
  do {                                                                                 \
    size_t got_ = 0, clean_ = 0;                                                       \
    hipError_t e_ = pool::take(c.device, #ptr, (size_t)std::max<int64_t>((bytes), 256), (void **)&(ptr), &got_, &clean_); \
    if (e_ != hipSuccess) {                                                            \
      (ptr) = nullptr;                                                                 \
      free_all(e); delete e;                                                           \
      return fail(MINRMS_E_DEVICE, "hipMalloc(%lld) failed: %s", (long long)(bytes), hipGetErrorString(e_)); \
    }                                                                                  \
    e->dev_blocks.push_back(minrms_engine::DevBlock{(void *)(ptr), got_, #ptr, clean_});       \
  } while (0)
  const size_t nA = (size_t)c.N1 * c.num_bb * 3, nB = (size_t)c.N2 * c.num_bb * 3;
  ALLOC(e->d_xyzA, 4 * nA);
  ALLOC(e->d_xyzB, 4 * nB);
  ALLOC(e->d_og, 48 * cap);
  ALLOC(e->d_C, 4 * L.c_floats * cap);
  ALLOC(e->d_bp, 4 * bp_words * cap);
  ALLOC(e->d_planes, 4 * plane_floats * cap);
  ALLOC(e->d_err, 256);
  ALLOC(e->d_edges, 4 * L.edge_floats * cap);
  ALLOC(e->d_cost, 4 * (int64_t)L.n_max * cap);
  ALLOC(e->d_ssd, 4 * (int64_t)rn * cap);
  ALLOC(e->d_xf, 48 * (int64_t)rn * cap);
  ALLOC(e->d_pairs, (int64_t)sizeof(short2) * rn * L.n_max * cap);
  ALLOC(e->d_keys, 8 * rn);
  ALLOC(e->d_best_ssd, 4 * rn);
  ALLOC(e->d_best_xf, 48 * rn);
  ALLOC(e->d_best_pairs, (int64_t)sizeof(short2) * rn * L.n_max);
  ALLOC(e->d_winner, 8 * rn);
  ALLOC(e->d_slot_of_k, 4 * rn);
  ALLOC(e->d_need, rn);
  ALLOC(e->d_counters, 64);
  ALLOC(e->d_klist, 4 * rn);
  ALLOC(e->d_ctrl, sizeof(RefineCtrl));
  ALLOC(e->d_clip, 4 * cap);
  ALLOC(e->d_oids, 4 * cap);
  ALLOC(e->d_stop, 4 * cap);
  if (e->coop_pb != 0 && (e->coop_groups.n_groups > 1 || e->coop_seg.n_groups > 1))
    ALLOC(e->d_gprog, 4 * e->coop_max_batch * e->coop_groups.total_tasks);
  if (e->coop_pb != 0 && e->coop_seg.n_groups > 1 && e->coop_seg.xedge_granules > 0)
    ALLOC(e->d_xedge, 8 * e->coop_seg_batch * e->coop_seg.xedge_granules);
  if (e->alt_by_pairs) {
    ALLOC(e->d_rank_pairs, (int64_t)sizeof(short2) * (c.num_solutions - 1) * rn * L.n_max);
    ALLOC(e->d_rank_keys, 8 * rn);
  }
  if (e->of) {
    e->filter_cap = 262144;
    ALLOC(e->d_fog, 48 * e->filter_cap);
    ALLOC(e->d_pass, e->filter_cap);
  }
#undef ALLOC
  hipError_t he = hipStreamCreate(&e->stream);
  if (he != hipSuccess) { free_all(e); delete e; return fail(MINRMS_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(he)); }
  if (e->d_gprog) {
    // the block may come from an earlier engine: its row counters must not read as tags of this one's launches
    he = hipMemsetAsync(e->d_gprog, 0, 4 * (size_t)e->coop_max_batch * e->coop_groups.total_tasks, e->stream);
    if (he != hipSuccess) { free_all(e); delete e; return fail(MINRMS_E_DEVICE, "hipMemsetAsync: %s", hipGetErrorString(he)); }
  }
  if (e->d_xedge) {
    // ... and neither must the tags of its granules
    he = hipMemsetAsync(e->d_xedge, 0, 8 * (size_t)e->coop_seg_batch * e->coop_seg.xedge_granules, e->stream);
    if (he != hipSuccess) { free_all(e); delete e; return fail(MINRMS_E_DEVICE, "hipMemsetAsync: %s", hipGetErrorString(he)); }
  }
  // Tail padding of the distance tables and the planes must hold floats, not
  // whatever bit patterns the memory was left with (NaN).  A block that an
  // earlier engine used for the same purpose holds what that engine's memset and
  // kernels wrote (finite values and +inf) as far as it used it: only the rest
  // is cleared.
  auto clear_rest = [&](void *ptr, size_t bytes) -> hipError_t {
    for (minrms_engine::DevBlock &b : e->dev_blocks) {
      if (b.ptr != ptr) continue;
      hipError_t r = hipSuccess;
      if (b.clean < bytes) r = hipMemsetAsync((char *)ptr + b.clean, 0, bytes - b.clean, e->stream);
      b.clean = std::max(b.clean, bytes);
      return r;
    }
    return hipErrorInvalidValue;
  };
  if (clear_rest(e->d_C, (size_t)(4 * L.c_floats * cap)) != hipSuccess ||
      clear_rest(e->d_planes, (size_t)std::max<int64_t>(4 * plane_floats * cap, 4)) != hipSuccess ||
      hipMemsetAsync(e->d_err, 0, 256, e->stream) != hipSuccess ||
      hipMemcpyAsync(e->d_xyzA, xyzA, 4 * nA, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
      hipMemcpyAsync(e->d_xyzB, xyzB, 4 * nB, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
    free_all(e); delete e;
    return fail(MINRMS_E_DEVICE, "device initialisation failed");
  }
  int rc = reset_results(e);
  if (rc) { free_all(e); delete e; return rc; }
  live_engines.fetch_add(1);
  *out = e;
  return MINRMS_OK;
}

extern "C" int minrms_engine_reset(minrms_engine *e)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->searched = false;
  e->tables_current = false;
  HIP_TRY(hipMemsetAsync(e->d_err, 0, 256, e->stream));
  e->device_failed = false;
  e->of_calls = 0;
  e->of_origin_calls = e->of_origin_start = 0;
  e->clip_now = e->sp.clip;
  e->h_pass.clear();
  if (e->num_solutions > 1) {
    e->alt_files.assign(e->sp.range_n, std::vector<alt::Candidate>());
    e->alt_running_best.assign(e->sp.range_n, MINRMS_UPPER_BOUND);
    e->alt_seen.clear();
  }
  e->ranks_current = false;
  return reset_results(e);
}

extern "C" void minrms_engine_destroy(minrms_engine *e)
{
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  // everything of this engine runs on its stream; other engines of the process are not waited for
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  else (void)hipDeviceSynchronize();
  free_all(e);
  delete e;
  live_engines.fetch_sub(1);
}

static int check_device_error(minrms_engine *e);

// Cooperative DP launch for small batches (see dp::k_dp_coop).
static int launch_coop(minrms_engine *e, int64_t count, const int *live_count = nullptr)
{
  // few orientations: a SIMD per task (segments), the extra workgroups run on CUs that would be idle
  const bool seg = e->coop_seg.n_groups > 1 && count <= e->coop_seg_batch && e->d_xedge != nullptr;
  const dp::CoopGroups &G = seg ? e->coop_seg : e->coop_groups;
  const int waves = G.max_tasks + G.fetch_waves;
  const int pbw = seg ? e->coop_seg_pb : e->coop_pb;         // columns per task
  const DpLayout &Lk = seg ? e->Lc_seg : e->Lc;
  size_t shmem = 512 + (size_t)waves * MINRMS_COOP_RING * 64 * 4 + (size_t)G.max_tasks * (pbw + 8) * 4;
  if (seg) shmem += (size_t)MINRMS_COOP_PRING * (G.max_tasks * pbw + 4) * 4 +     // the plane ring
                    (size_t)G.max_tasks * MINRMS_COOP_ORING * pbw * 4;             // the out ring
  // workgroups that wait for each other across CUs: one per CU (more LDS than
  // half a CU has), the setting the hand-over was measured in
  if (G.n_groups > 1) shmem = std::max<size_t>(shmem, 81 * 1024);
  const dim3 block(64 * waves), grid((unsigned)(count * G.n_groups));
  int gbase = 0;
  if (G.n_groups > 1) {
    // the launch number tags the row counters and granules; when it runs out of bits the arrays start from zero again
    if (++e->coop_launches >= (1 << (31 - MINRMS_COOP_ROW_BITS))) {
      HIP_TRY(hipMemsetAsync(e->d_gprog, 0, 4 * (size_t)e->coop_max_batch * G.total_tasks, e->stream));
      if (e->d_xedge)
        HIP_TRY(hipMemsetAsync(e->d_xedge, 0, 8 * (size_t)e->coop_seg_batch * e->coop_seg.xedge_granules, e->stream));
      e->coop_launches = 1;
    }
    gbase = e->coop_launches << MINRMS_COOP_ROW_BITS;
  }
#define MINRMS_LAUNCH_COOP(PBV, SEGV)                                                              \
  hipLaunchKernelGGL((dp::k_dp_coop<PBV, SEGV>), grid, block, shmem, e->stream, Lk, G, e->d_C, e->d_bp,  \
                     e->d_planes, e->d_cost, e->d_err, e->d_gprog, e->d_xedge, gbase, live_count)
  if (seg) {
    if (pbw == 8) MINRMS_LAUNCH_COOP(8, true);
    else if (pbw == 16) MINRMS_LAUNCH_COOP(16, true);
    else if (pbw == 32) MINRMS_LAUNCH_COOP(32, true);
    else MINRMS_LAUNCH_COOP(64, true);
  } else {
    if (e->coop_pb == 16) MINRMS_LAUNCH_COOP(16, false);
    else if (e->coop_pb == 32) MINRMS_LAUNCH_COOP(32, false);
    else MINRMS_LAUNCH_COOP(64, false);
  }
#undef MINRMS_LAUNCH_COOP
  return 0;
}

// Traceback + refit of every (slot, n): one wave per chain while the batch is
// small enough to be latency-bound, one thread per chain otherwise.
// k_lo/k_cnt (thread-per-chain kernel only): the entries of one layer block;
// stop: slots that have left the n-loop are skipped.
static void launch_trace_fit(minrms_engine *e, const DpLayout &L, int64_t count, bool want_xf, bool want_pairs,
                             int k_lo = 0, int k_cnt = -1, const int32_t *stop = nullptr)
{
  const SearchParams &sp = e->sp;
  if (k_cnt < 0) k_cnt = sp.range_n;
  const int64_t chains = count * k_cnt;
  float *xf = want_xf ? e->d_xf : nullptr;
  if (!stop && chains <= e->wave_trace_max_chains) {
    const size_t lds = MINRMS_MAX_BLOCKS * sizeof(BlockRef) + fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)L.n_max;
    hipLaunchKernelGGL(k_trace_fit_wave, dim3((unsigned)chains), dim3(64), lds, e->stream, L, sp, count,
                       e->d_bp, e->d_cost, e->d_xyzA, e->d_xyzB, e->d_og, e->d_pairs, e->d_ssd, xf, nullptr);
  } else {
    // The walk waits for one dependent load per row of the path, so the kernel
    // lives on occupancy: 82 VGPRs allow 5 waves per SIMD = 20 single-wave
    // workgroups per CU, which 160 KB of LDS sustain up to 8 KB each.  The path
    // strings (2 bits per move, N1+N2 moves at most) stay in LDS while they fit
    // in that; longer ones go to memory -- into the buffer of the alignments,
    // which a launch that does not write alignments leaves unused.  Both
    // coordinate sets join them in LDS only while that costs next to no
    // occupancy (250x250: 27 instead of 36 ms per sweep with the coordinates
    // read from memory).
    const size_t coord_bytes = 12 * (size_t)(sp.N1 + sp.N2) * sp.nbb;
    const int path_words = (sp.N1 + sp.N2 + 16) / 16;        // 16 moves per word
    const size_t refs_bytes = MINRMS_MAX_BLOCKS * sizeof(BlockRef);
    const int64_t blocks = (chains + 63) / 64;
    uint32_t *path_global = nullptr;
    const char *where = getenv("MINRMS_TRACE_PATH");        // test hook: "lds" / "memory"
    const bool long_path = refs_bytes + (size_t)path_words * 256 > 10240;   // fewer than 16 workgroups per CU
    if (!want_pairs && (where ? !strcmp(where, "memory") : long_path) &&
        blocks * 64 * path_words * 4 <= (int64_t)sizeof(short2) * sp.range_n * L.n_max * e->batch_cap)
      path_global = (uint32_t *)e->d_pairs;
    const size_t base_lds = refs_bytes + (path_global ? 0 : (size_t)path_words * 256);
    const int in_lds = base_lds + coord_bytes <= 9216;
    const size_t lds = base_lds + (in_lds ? coord_bytes : 0);
#define MINRMS_LAUNCH_TRACE(LDSC, ONE, PL)                                                         \
    hipLaunchKernelGGL((k_trace_fit<LDSC, ONE, PL>), dim3((unsigned)blocks), dim3(64), lds, e->stream, L, sp, \
                       count, e->d_bp, e->d_cost, e->d_xyzA, e->d_xyzB, e->d_og,                  \
                       want_pairs ? e->d_pairs : nullptr, e->d_ssd, xf, path_words, in_lds, path_global, \
                       k_lo, k_cnt, stop)
#define MINRMS_LAUNCH_TRACE2(LDSC, ONE) do { if (path_global) MINRMS_LAUNCH_TRACE(LDSC, ONE, false); else MINRMS_LAUNCH_TRACE(LDSC, ONE, true); } while (0)
    if (in_lds) { if (sp.nbb == 1) MINRMS_LAUNCH_TRACE2(true, true); else MINRMS_LAUNCH_TRACE2(true, false); }
    else        { if (sp.nbb == 1) MINRMS_LAUNCH_TRACE2(false, true); else MINRMS_LAUNCH_TRACE2(false, false); }
#undef MINRMS_LAUNCH_TRACE2
#undef MINRMS_LAUNCH_TRACE
  }
}

// Distance tables, DP, traceback + refit for the first `count` slots of d_og.
// slot_clip: device array with the distance cutoff of every slot, or NULL for
// the cutoff currently in force.
// og_table/og_rows: if given, slot s is evaluated at transform og_table[og_rows[s]]
// instead of d_og[s] (only valid with refine_while_searching, where nothing
// else reads the transform a slot was evaluated at).
static int run_batch(minrms_engine *e, int64_t count, bool want_xf, const float *slot_clip = nullptr,
                     const float *og_table = nullptr, const int32_t *og_rows = nullptr, bool want_pairs = true,
                     bool cost_only = false)
{
  const bool coop = e->coop_pb != 0 && count <= e->coop_max_batch;
  const DpLayout &L = coop ? e->Lc : e->L;
  const SearchParams &sp = e->sp;
  hipStream_t st = e->stream;
  launch_cij(sp, count, st, L.c_floats, e->d_xyzA, e->d_xyzB, og_table ? og_table : e->d_og, e->d_C, e->clip_now,
             slot_clip, og_rows, nullptr);
  if (coop) {
    int rc = launch_coop(e, count);
    if (rc) return rc;
    if (!cost_only) launch_trace_fit(e, L, count, want_xf, want_pairs);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  minrms_engine::Stamp stamp;
  stamp.first = e->ev_used; stamp.nb = 0; stamp.count = count;
  auto next_event = [&](hipEvent_t *ev) -> hipError_t {
    if (e->ev_used == e->ev_pool.size()) {
      hipEvent_t n;
      hipError_t r = hipEventCreate(&n);
      if (r != hipSuccess) return r;
      e->ev_pool.push_back(n);
    }
    *ev = e->ev_pool[e->ev_used++];
    return hipSuccess;
  };
  // Early termination (search_or_dyn3d.cc:1720-1721, 1810-1811): with a distance
  // or RMSD cutoff the reference leaves the n-loop of most orientations long
  // before n_max.  The sweep then goes layer block by layer block: after the DP
  // of a block, the alignments and sums of ITS layers are evaluated and the
  // orientations whose loop ends among them are marked; their workgroups in the
  // launches of the blocks above leave at once, and so do their chains in the
  // tracebacks above.
  const bool stepwise = L.NB > 1 && (sp.max_rmsd != -1.0f || slot_clip != nullptr || e->clip_now != MINRMS_UPPER_BOUND) &&
                        !getenv("MINRMS_NO_EARLY_EXIT");
  const int32_t *stop = nullptr;
  if (stepwise) {
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, e->d_stop, count,
                       (int32_t)(sp.n_max + 1));
    stop = e->d_stop;
  }
  for (int b = 0; b < L.NB; ++b) {
    const BlockDesc &bd = L.blk[b];
    dp_kernel_fn fn = pick_dp_kernel(bd.PB, b > 0, b + 1 < L.NB, !cost_only);
    if (!fn) return fail(MINRMS_E_ARG, "no DP kernel for panel width %d", bd.PB);
    for (int p = 0; p < bd.P; ++p) {
      if (e->timing && stamp.nb < MINRMS_MAX_BLOCKS * 8) {
        hipEvent_t ev;
        HIP_TRY(next_event(&ev));
        HIP_TRY(hipEventRecord(ev, st));
        stamp.blocks[stamp.nb++] = b;
      }
      hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(64), 0, st, L, e->d_C, e->d_bp, e->d_planes,
                         e->d_edges, e->d_cost, b, p, stop);
    }
    if (stepwise) {
      if (e->timing && stamp.nb > 0 && stamp.nb < MINRMS_MAX_BLOCKS * 8) {
        // the stamp after the block's last panel: what follows is not DP time
        hipEvent_t ev;
        HIP_TRY(next_event(&ev));
        HIP_TRY(hipEventRecord(ev, st));
        stamp.blocks[stamp.nb++] = -1;
      }
      const int n_lo = MINRMS_LAYERS_PER_BLOCK * b + 1, n_hi = std::min(sp.n_max, MINRMS_LAYERS_PER_BLOCK * (b + 1));
      const int k_lo = std::max(n_lo, sp.n_min) - sp.n_min, k_hi = n_hi - sp.n_min;
      if (!cost_only && k_hi >= k_lo) launch_trace_fit(e, L, count, want_xf, want_pairs, k_lo, k_hi - k_lo + 1, stop);
      if (b + 1 < L.NB)
        hipLaunchKernelGGL(k_update_stop, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, sp, count, n_lo, n_hi,
                           e->d_cost, cost_only ? e->d_cost + (sp.n_min - 1) : e->d_ssd,
                           cost_only ? sp.n_max : sp.range_n, e->d_stop);
    }
  }
  if (e->timing && stamp.nb > 0) {
    hipEvent_t ev;
    HIP_TRY(next_event(&ev));
    HIP_TRY(hipEventRecord(ev, st));
    e->stamps.push_back(stamp);
  }
  if (stepwise) {
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (!cost_only) launch_trace_fit(e, L, count, want_xf, want_pairs);
  HIP_TRY(hipGetLastError());
  return 0;
}

// First n at which the n-loop leaves an orientation: host twin of first_break().
static int first_break_host(const SearchParams &sp, const float *cost, const float *ssd)
{
  for (int n = 2; n <= sp.n_max; ++n) {
    if (cost[n - 1] >= MINRMS_UPPER_BOUND) return n;
    if (n < sp.n_min) continue;
    const float s = ssd[n - sp.n_min];
    if (sp.max_rmsd != -1.0f && s > (float)(n * sp.nbb) * (sp.max_rmsd * sp.max_rmsd)) return n;
  }
  return sp.n_max + 1;
}

// KeepTrackOfAltSolutions (search_or_dyn3d.cc:477-542) for the batch that has
// just been evaluated (with transforms): slot s holds orientation
// oids ? oids[s] : o_base + s; slots are in ascending o_id order.
static int alt_collect_batch_on_host(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids);

// The records of one batch that may become alternate solutions: filtered on the
// device (MINRMS_ALT_HOST_FILTER=1, a test hook, copies everything instead).
static int alt_collect_batch(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n;
  const bool with_xf = !e->alt_by_pairs;
  const float tol = e->cfg.alt_rmsd_tolerance;
  // without a tolerance every record is kept; a tolerance below 1 would make "lowers the best => survives" false
  if (tol == -1.0f || tol < 1.0f || getenv("MINRMS_ALT_HOST_FILTER")) return alt_collect_batch_on_host(e, count, o_base, oids);
  const int n_chunks = (int)((count + MINRMS_ALT_CHUNK - 1) / MINRMS_ALT_CHUNK);
  const int max_chunks = (int)((e->batch_cap + MINRMS_ALT_CHUNK - 1) / MINRMS_ALT_CHUNK);
  if (!e->d_alt_best) {
    HIP_TRY(hipMalloc((void **)&e->d_alt_best, 4 * (size_t)rn));
    HIP_TRY(hipMalloc((void **)&e->d_alt_chunk, 4 * (size_t)rn * max_chunks));
    HIP_TRY(hipMalloc((void **)&e->d_alt_offsets, 4 * (size_t)rn * max_chunks));
    HIP_TRY(hipMalloc((void **)&e->d_alt_count, 4));
  }
  hipStream_t st = e->stream;
  HIP_TRY(hipMemcpyAsync(e->d_alt_best, e->alt_running_best.data(), 4 * (size_t)rn, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_first_break, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, sp, count, e->d_cost,
                     e->d_ssd, e->d_stop);
  const dim3 chunk_grid((unsigned)((rn + 63) / 64), (unsigned)n_chunks);
  hipLaunchKernelGGL(k_alt_chunk_min, chunk_grid, dim3(64), 0, st, sp, count, e->d_ssd, e->d_stop, e->d_alt_chunk);
  hipLaunchKernelGGL(k_alt_chunk_prefix, dim3((unsigned)((rn + 63) / 64)), dim3(64), 0, st, sp, n_chunks, e->d_alt_best,
                     e->d_alt_chunk);
  const float tol2 = (tol * tol) * 1.0001f;
  hipLaunchKernelGGL(k_alt_filter<false>, chunk_grid, dim3(64), 0, st, sp, count, n_chunks, e->d_ssd, nullptr, e->d_stop,
                     e->d_alt_chunk, tol2, nullptr, e->d_alt_offsets);
  hipLaunchKernelGGL(k_alt_offsets, dim3(1), dim3(256), 0, st, (int64_t)rn * n_chunks, e->d_alt_offsets, e->d_alt_count);
  HIP_TRY(hipGetLastError());
  int n_rec = 0;
  HIP_TRY(hipMemcpyAsync(&n_rec, e->d_alt_count, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (getenv("MINRMS_ALT_DEBUG"))
    fprintf(stderr, "alt filter: %d survivors of %lld records\n", n_rec, (long long)(count * rn));
  if (n_rec == 0) return 0;
  if (n_rec > e->alt_rec_cap) {                         // room for the survivors, grown as needed
    if (e->d_alt_rec) HIP_TRY(hipFree(e->d_alt_rec));
    e->d_alt_rec = nullptr;
    e->alt_rec_cap = (int64_t)n_rec + n_rec / 4;
    HIP_TRY(hipMalloc((void **)&e->d_alt_rec, sizeof(AltRecord) * (size_t)e->alt_rec_cap));
  }
  hipLaunchKernelGGL(k_alt_filter<true>, chunk_grid, dim3(64), 0, st, sp, count, n_chunks, e->d_ssd,
                     with_xf ? e->d_xf : nullptr, e->d_stop, e->d_alt_chunk, tol2, e->d_alt_rec, e->d_alt_offsets);
  HIP_TRY(hipGetLastError());
  std::vector<AltRecord> rec((size_t)n_rec);
  HIP_TRY(hipMemcpyAsync(rec.data(), e->d_alt_rec, sizeof(AltRecord) * (size_t)n_rec, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // the records of one n are together and in o_id order, as the one-pass loop would meet them
  for (const AltRecord &r : rec) {
    const int k = r.k, n = sp.n_min + k;
    const float v = r.ssd;
    if (v < e->alt_running_best[k]) e->alt_running_best[k] = v;        // :1825-1835 comes first
    if (v <= e->alt_running_best[k] * (tol * tol)) {
      alt::Candidate cand;
      cand.o_id = oids ? (int64_t)oids[r.slot] : o_base + r.slot;
      cand.rmsd = sqrtf(v / (float)(n * sp.nbb));
      cand.ssd = v;
      memcpy(cand.xf, r.xf, 48);
      e->alt_files[k].push_back(cand);
    }
  }
  return 0;
}

static int alt_collect_batch_on_host(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n;
  // by matched pairs only (o_id, rmsd) is kept, and only under an RMSD tolerance (:2179-2186)
  const bool with_xf = !e->alt_by_pairs;
  std::vector<float> ssd((size_t)count * rn), cost((size_t)count * sp.n_max), xf(with_xf ? (size_t)count * rn * 12 : 0);
  HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_ssd, 4 * ssd.size(), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(cost.data(), e->d_cost, 4 * cost.size(), hipMemcpyDeviceToHost, e->stream));
  if (with_xf) HIP_TRY(hipMemcpyAsync(xf.data(), e->d_xf, 4 * xf.size(), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const float tol = e->cfg.alt_rmsd_tolerance;
  for (int64_t s = 0; s < count; ++s) {
    const float *c = &cost[(size_t)s * sp.n_max], *sd = &ssd[(size_t)s * rn];
    const int stop = first_break_host(sp, c, sd);
    for (int n = std::max(2, sp.n_min); n < stop; ++n) {
      const int k = n - sp.n_min;
      const float v = sd[k];
      if (v < e->alt_running_best[k]) e->alt_running_best[k] = v;        // :1825-1835 comes first
      if (tol == -1.0f || v <= e->alt_running_best[k] * (tol * tol)) {
        alt::Candidate cand;
        cand.o_id = oids ? (int64_t)oids[s] : o_base + s;
        cand.rmsd = sqrtf(v / (float)(n * sp.nbb));
        cand.ssd = v;
        if (with_xf) memcpy(cand.xf, &xf[((size_t)s * rn + k) * 12], 48);
        else memset(cand.xf, 0, 48);
        e->alt_files[k].push_back(cand);
      }
    }
  }
  return 0;
}

// FirstPass with the orientation filter on (search_or_dyn3d.cc:1696-1698): the
// filter runs over the whole range first, the survivors are packed into dense
// batches for the 3-index DP.
static int search_filtered(minrms_engine *e, const float *og, int64_t start, int64_t stop)
{
  e->h_pass.assign((size_t)(stop - start), 0);
  e->of_origin_calls = e->of_calls;
  e->of_origin_start = start;
  const size_t lds = 2 * (size_t)(e->sp.N2 + 1) * 4;
  std::vector<uint32_t> oids;
  std::vector<float> h_og, h_clip;
  for (int64_t off = start; off < stop; off += e->filter_cap) {
    const int64_t cnt = std::min(e->filter_cap, stop - off);
    HIP_TRY(hipMemcpyAsync(e->d_fog, og + 12 * off, 48 * cnt, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(nwf::k_nw_filter, dim3((unsigned)cnt), dim3(64), lds, e->stream, e->nwp, cnt,
                       e->d_xyzA, e->d_xyzB, e->d_fog, e->d_pass);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&e->h_pass[(size_t)(off - start)], e->d_pass, cnt, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    oids.clear();
    for (int64_t q = 0; q < cnt; ++q)
      if (e->h_pass[(size_t)(off - start + q)]) oids.push_back((uint32_t)(off + q));
    for (size_t base = 0; base < oids.size(); base += (size_t)e->batch_cap) {
      const int64_t count = std::min<int64_t>(e->batch_cap, (int64_t)(oids.size() - base));
      h_og.resize((size_t)count * 12);
      h_clip.resize((size_t)count);
      for (int64_t s = 0; s < count; ++s) {
        const int64_t o = oids[base + s];
        memcpy(&h_og[12 * (size_t)s], og + 12 * o, 48);
        h_clip[(size_t)s] = clip_after(e, e->of_origin_calls + (o - start) + 1);
      }
      HIP_TRY(hipMemcpyAsync(e->d_og, h_og.data(), 48 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_clip, h_clip.data(), 4 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_oids, &oids[base], 4 * count, hipMemcpyHostToDevice, e->stream));
      const bool alts = e->num_solutions > 1 && (!e->alt_by_pairs || e->cfg.alt_rmsd_tolerance != -1.0f);
      const bool cost_only = !e->sp.rws && !alts;
      int rc = run_batch(e, count, alts && !e->alt_by_pairs, e->d_clip, nullptr, nullptr, false, cost_only);
      if (rc) return rc;
      hipLaunchKernelGGL(k_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream, e->sp,
                         count, 0, e->d_oids, e->d_cost, cost_only ? e->d_cost + (e->sp.n_min - 1) : e->d_ssd,
                         cost_only ? e->sp.n_max : e->sp.range_n, e->d_keys);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(e->stream));   // the staging vectors are reused
      if (alts) { rc = alt_collect_batch(e, count, 0, &oids[base]); if (rc) return rc; }
    }
  }
  e->of_calls += stop - start;
  e->clip_now = clip_after(e, e->of_calls);
  return 0;
}

extern "C" int minrms_engine_set_filter_calls(minrms_engine *e, int64_t calls)
{
  if (!e || calls < 0) return fail(MINRMS_E_ARG, "bad argument");
  e->of_calls = calls;
  e->clip_now = clip_after(e, calls);
  return MINRMS_OK;
}

extern "C" int minrms_engine_filter_verdicts(minrms_engine *e, uint8_t *pass, int64_t cap, int64_t *count)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (!e->searched) return fail(MINRMS_E_STATE, "filter_verdicts before search");
  if (count) *count = (int64_t)e->h_pass.size();
  if (pass)
    memcpy(pass, e->h_pass.data(), (size_t)std::min<int64_t>(cap, (int64_t)e->h_pass.size()));
  return MINRMS_OK;
}

extern "C" int minrms_engine_search(minrms_engine *e, const float *og, int64_t n_og, int64_t start,
                                    int64_t stop)
{
  if (!e || !og) return fail(MINRMS_E_ARG, "null argument");
  if (start < 0 || stop > n_og || stop < start) return fail(MINRMS_E_ARG, "bad orientation range [%lld,%lld) of %lld", (long long)start, (long long)stop, (long long)n_og);
  if (n_og > 0xffffffffll) return fail(MINRMS_E_ARG, "more than 2^32 orientations");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->h_og.assign(og, og + 12 * n_og);
  if (e->of) {
    int rc = search_filtered(e, og, start, stop);
    if (rc) return rc;
  } else {
    // batches of equal size: a short last batch would leave most of the GPU idle
    const int64_t n_batches = (stop - start + e->batch_cap - 1) / e->batch_cap;
    const int64_t per_batch = n_batches ? (stop - start + n_batches - 1) / n_batches : 0;
    for (int64_t off = start; off < stop; off += per_batch) {
      const int64_t count = std::min(per_batch, stop - off);
      HIP_TRY(hipMemcpyAsync(e->d_og, og + 12 * off, 48 * count, hipMemcpyHostToDevice, e->stream));
      const bool alts = e->num_solutions > 1 && (!e->alt_by_pairs || e->cfg.alt_rmsd_tolerance != -1.0f);
      // Without refinement while searching the sum of squared distances of an
      // orientation is the DP cost itself (search_or_dyn3d.cc:1783-1790 not
      // taken): the running minimum needs neither traceback nor back-pointers.
      const bool cost_only = !e->sp.rws && !alts;
      int rc = run_batch(e, count, alts && !e->alt_by_pairs, nullptr, nullptr, nullptr, false, cost_only);
      if (rc) return rc;
      hipLaunchKernelGGL(k_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, e->stream, e->sp,
                         count, off, nullptr, e->d_cost, cost_only ? e->d_cost + (e->sp.n_min - 1) : e->d_ssd,
                         cost_only ? e->sp.n_max : e->sp.range_n, e->d_keys);
      HIP_TRY(hipGetLastError());
      if (alts) { rc = alt_collect_batch(e, count, off, nullptr); if (rc) return rc; }
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    e->h_pass.assign((size_t)(stop - start), 1);
  }
  { int rc = check_device_error(e); if (rc) return rc; }     // small batches run the cooperative kernel
  if (e->alt_by_pairs) {
    e->alt_seen.resize((size_t)n_og, 0);
    for (int64_t o = start; o < stop; ++o) e->alt_seen[(size_t)o] = e->h_pass[(size_t)(o - start)] ? 2 : 1;
  }
  e->searched = true;
  e->tables_current = false;
  e->ranks_current = false;
  return MINRMS_OK;
}

extern "C" int minrms_engine_get_keys(minrms_engine *e, uint64_t *keys)
{
  if (!e || !keys) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  { int rc = check_device_error(e); if (rc) return rc; }
  HIP_TRY(hipMemcpyAsync(keys, e->d_keys, 8 * e->sp.range_n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MINRMS_OK;
}

extern "C" int minrms_engine_set_keys(minrms_engine *e, const uint64_t *keys)
{
  if (!e || !keys) return fail(MINRMS_E_ARG, "null argument");
  if (e->num_solutions > 1 && e->alt_by_pairs)
    return fail(MINRMS_E_ARG, "alternate solutions by matched pairs need the whole orientation range in one engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipMemcpyAsync(e->d_keys, keys, 8 * e->sp.range_n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->tables_current = false;
  return MINRMS_OK;
}

extern "C" int minrms_engine_keys_device_ptr(minrms_engine *e, void **ptr)
{
  if (!e || !ptr) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  { int rc = check_device_error(e); if (rc) return rc; }
  *ptr = e->d_keys;
  e->tables_current = false;
  return MINRMS_OK;
}

// Alternates by orientation with the search split over several engines.  Every
// engine keeps the records KeepTrackOfAltSolutions (search_or_dyn3d.cc:477-542)
// would keep if the sweep began at its own first orientation: its running
// minimum starts later than the one-process minimum, so it keeps a superset.
// The union of all engines' records, put in o_id order per n and filtered again
// with the running minimum rebuilt from the records themselves (an orientation
// that lowers the minimum always passes its own test, so it is among them),
// is exactly the one-process list.
extern "C" int minrms_engine_alt_num_candidates(minrms_engine *e, int64_t *count)
{
  if (!e || !count) return fail(MINRMS_E_ARG, "null argument");
  int64_t n = 0;
  for (const std::vector<alt::Candidate> &v : e->alt_files) n += (int64_t)v.size();
  *count = n;
  return MINRMS_OK;
}

extern "C" int minrms_engine_alt_get_candidates(minrms_engine *e, int32_t *k, int64_t *o_id, float *ssd,
                                                float *rmsd, float *xf)
{
  if (!e || !k || !o_id || !ssd || !rmsd || !xf) return fail(MINRMS_E_ARG, "null argument");
  int64_t q = 0;
  for (size_t kk = 0; kk < e->alt_files.size(); ++kk)
    for (const alt::Candidate &c : e->alt_files[kk]) {
      k[q] = (int32_t)kk; o_id[q] = c.o_id; ssd[q] = c.ssd; rmsd[q] = c.rmsd;
      memcpy(xf + 12 * q, c.xf, 48);
      ++q;
    }
  return MINRMS_OK;
}

extern "C" int minrms_engine_alt_set_candidates(minrms_engine *e, int64_t count, const int32_t *k, const int64_t *o_id,
                                                const float *ssd, const float *rmsd, const float *xf)
{
  if (!e || (count > 0 && (!k || !o_id || !ssd || !rmsd || !xf))) return fail(MINRMS_E_ARG, "null argument");
  if (e->num_solutions <= 1 || e->alt_by_pairs)
    return fail(MINRMS_E_ARG, "candidate lists exist for alternate solutions by orientation only");
  const int rn = e->sp.range_n;
  std::vector<std::vector<alt::Candidate>> merged(rn);
  for (int64_t q = 0; q < count; ++q) {
    if (k[q] < 0 || k[q] >= rn) return fail(MINRMS_E_ARG, "candidate %lld: entry %d outside 0..%d", (long long)q, k[q], rn - 1);
    alt::Candidate c;
    c.o_id = o_id[q]; c.ssd = ssd[q]; c.rmsd = rmsd[q];
    memcpy(c.xf, xf + 12 * q, 48);
    merged[k[q]].push_back(c);
  }
  const float tol = e->cfg.alt_rmsd_tolerance;
  for (int kk = 0; kk < rn; ++kk) {
    std::vector<alt::Candidate> &v = merged[kk];
    std::stable_sort(v.begin(), v.end(), [](const alt::Candidate &a, const alt::Candidate &b) { return a.o_id < b.o_id; });
    float best = MINRMS_UPPER_BOUND;
    std::vector<alt::Candidate> kept;
    for (size_t q = 0; q < v.size(); ++q) {
      if (q > 0 && v[q].o_id == v[q - 1].o_id) continue;                 // the same record from two engines
      if (v[q].ssd < best) best = v[q].ssd;                               // :1825-1835 comes first
      if (tol == -1.0f || v[q].ssd <= best * (tol * tol)) kept.push_back(v[q]);
    }
    e->alt_files[kk].swap(kept);
    e->alt_running_best[kk] = best;
  }
  e->tables_current = false;
  e->ranks_current = false;
  return MINRMS_OK;
}

// Rank 0 of the host tables <- device tables.
static int alt_pull_rank0(minrms_engine *e)
{
  const int rn = e->sp.range_n;
  alt::RankTables &t = e->ranks;
  HIP_TRY(hipMemcpyAsync(t.ssd.data(), e->d_best_ssd, 4 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(t.xf.data(), e->d_best_xf, 48 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(t.pairs.data(), e->d_best_pairs, sizeof(short2) * (size_t)rn * e->cfg.n_max,
                         hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return 0;
}

// Device tables <- rank 0 of the host tables (the ranks may have been re-ordered).
static int alt_push_rank0(minrms_engine *e)
{
  const int rn = e->sp.range_n;
  alt::RankTables &t = e->ranks;
  HIP_TRY(hipMemcpyAsync(e->d_best_ssd, t.ssd.data(), 4 * rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_best_xf, t.xf.data(), 48 * rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_best_pairs, t.pairs.data(), sizeof(short2) * (size_t)rn * e->cfg.n_max,
                         hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return 0;
}

// What Search() does after FirstPass when alternates were asked for
// (CalcAltSolutionsUsingOrientation, search_or_dyn3d.cc:421-450): choose, per n,
// the orientations of the 2nd, 3rd, ... best alignments among the candidates
// kept during the sweep, then re-solve those orientations and store the
// alignments under their rank.
static int alt_fill_ranks(minrms_engine *e)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n, S = e->num_solutions, n_max = e->cfg.n_max;
  alt::RankTables &t = e->ranks;
  t.init(S, rn, n_max, MINRMS_UPPER_BOUND);
  int rc = alt_pull_rank0(e);
  if (rc) return rc;

  alt::OrientationMetric metric;
  metric.init(sp.N2 * sp.nbb, e->h_xyzB.data());
  const float min_delta = (e->cfg.alt_min_3d_difference * e->cfg.alt_min_3d_difference) * sp.N2 * sp.nbb;
  std::multimap<int64_t, alt::Owner> owners;
  for (int k = 0; k < rn; ++k) {
    if (t.s(0, k) == MINRMS_UPPER_BOUND) continue;
    alt::choose_alternates(sp.n_min + k, e->alt_files[k], t.x(0, k), S, e->cfg.alt_rmsd_tolerance, min_delta,
                           metric, owners);
  }
  // distinct orientations, ascending
  std::vector<int64_t> oids;
  for (auto it = owners.begin(); it != owners.end(); ++it)
    if (oids.empty() || oids.back() != it->first) oids.push_back(it->first);
  const int64_t n_og = (int64_t)e->h_og.size() / 12;
  std::vector<float> og, ssd, xf;
  std::vector<short2> pairs;
  for (size_t base = 0; base < oids.size(); base += (size_t)e->batch_cap) {
    const int64_t count = std::min<int64_t>(e->batch_cap, (int64_t)(oids.size() - base));
    og.resize((size_t)count * 12);
    for (int64_t s = 0; s < count; ++s) {
      if (oids[base + s] >= n_og) return fail(MINRMS_E_STATE, "alternate o_id outside the orientation list");
      memcpy(&og[12 * (size_t)s], &e->h_og[12 * (size_t)oids[base + s]], 48);
    }
    HIP_TRY(hipMemcpyAsync(e->d_og, og.data(), 48 * count, hipMemcpyHostToDevice, e->stream));
    rc = run_batch(e, count, true);            // the cutoff now in force (search_or_dyn3d.cc:1293)
    if (rc) return rc;
    ssd.resize((size_t)count * rn);
    xf.resize((size_t)count * rn * 12);
    pairs.resize((size_t)count * rn * n_max);
    HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_ssd, 4 * ssd.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(xf.data(), e->d_xf, 4 * xf.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(pairs.data(), e->d_pairs, sizeof(short2) * pairs.size(), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (int64_t s = 0; s < count; ++s) {
      auto range = owners.equal_range(oids[base + s]);
      for (auto it = range.first; it != range.second; ++it) {
        const int k = it->second.n - sp.n_min, r = it->second.r;
        const size_t rec = (size_t)s * rn + k;
        t.s(r, k) = ssd[rec];
        memcpy(t.x(r, k), &xf[rec * 12], 48);
        memcpy(t.p(r, k), &pairs[rec * n_max], sizeof(short2) * (size_t)it->second.n);
      }
    }
  }
  e->ranks_current = true;
  return 0;
}

// Alternates by matched pairs (CalcAltSolutionsUsingResiduePairs,
// search_or_dyn3d.cc:342-413): for rank 1, 2, ... sweep the orientations again
// (all of them, or those that came within the RMSD tolerance of the best for
// some n) and keep, per n, the best alignment that shares few enough pairs with
// every better rank.
static int alt_fill_ranks_by_pairs(minrms_engine *e)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n, S = e->num_solutions, n_max = e->cfg.n_max;
  const float tol = e->cfg.alt_rmsd_tolerance;
  alt::RankTables &t = e->ranks;
  t.init(S, rn, n_max, MINRMS_UPPER_BOUND);
  int rc = alt_pull_rank0(e);
  if (rc) return rc;
  const int64_t n_og = (int64_t)e->h_og.size() / 12;

  // use_these_orientations (:385-399, 2219-2247); every member calls the filter again
  std::vector<int64_t> use;
  {
    std::vector<unsigned char> want((size_t)n_og, 0);
    if (tol == -1.0f) {
      for (int64_t o = 0; o < n_og && o < (int64_t)e->alt_seen.size(); ++o) want[(size_t)o] = e->alt_seen[(size_t)o] != 0;
    } else {
      for (int k = 0; k < rn; ++k) {
        if (t.s(0, k) == MINRMS_UPPER_BOUND) continue;
        const float best_rmsd = sqrtf(t.s(0, k) / (float)((sp.n_min + k) * sp.nbb));
        const float tolerable = tol * best_rmsd;
        for (const alt::Candidate &c : e->alt_files[k])
          if (tolerable >= c.rmsd && c.o_id < n_og) want[(size_t)c.o_id] = 1;
      }
    }
    for (int64_t o = 0; o < n_og; ++o)
      if (want[(size_t)o]) use.push_back(o);
  }

  std::vector<uint32_t> oids;
  std::vector<float> clips, og, ssd, xf;
  std::vector<short2> pairs;
  std::vector<uint64_t> keys(rn);
  for (int r = 1; r < S; ++r) {
    HIP_TRY(hipMemcpyAsync(e->d_rank_pairs, t.pairs.data(), sizeof(short2) * (size_t)r * rn * n_max,
                           hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(e->d_rank_keys, 0xFF, 8 * rn, e->stream));
    // the orientations that pass the filter, each with the cutoff in force
    // after its own filter call
    oids.clear();
    clips.clear();
    for (size_t q = 0; q < use.size(); ++q) {
      const int64_t o = use[q];
      if (e->of && e->alt_seen[(size_t)o] != 2) continue;
      oids.push_back((uint32_t)o);
      clips.push_back(e->of ? clip_after(e, e->of_calls + (int64_t)q + 1) : e->clip_now);
    }
    if (e->of) e->of_calls += (int64_t)use.size();
    for (size_t base = 0; base < oids.size(); base += (size_t)e->batch_cap) {
      const int64_t count = std::min<int64_t>(e->batch_cap, (int64_t)(oids.size() - base));
      og.resize((size_t)count * 12);
      for (int64_t s = 0; s < count; ++s) memcpy(&og[12 * (size_t)s], &e->h_og[12 * (size_t)oids[base + s]], 48);
      HIP_TRY(hipMemcpyAsync(e->d_og, og.data(), 48 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_clip, &clips[base], 4 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_oids, &oids[base], 4 * count, hipMemcpyHostToDevice, e->stream));
      rc = run_batch(e, count, false, e->d_clip);
      if (rc) return rc;
      hipLaunchKernelGGL(k_reduce_dissimilar, dim3((unsigned)count), dim3(256), 0, e->stream, sp, n_max, count,
                         e->d_oids, e->d_cost, e->d_ssd, e->d_pairs, r, e->d_rank_pairs,
                         e->cfg.alt_max_pairs_common, e->d_rank_keys);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(e->stream));      // og is reused
    }
    HIP_TRY(hipMemcpyAsync(keys.data(), e->d_rank_keys, 8 * rn, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // re-solve the winners of this pass for their alignments and transforms
    std::vector<uint32_t> win;
    for (int k = 0; k < rn; ++k)
      if (keys[k] != ~0ull) win.push_back((uint32_t)(keys[k] & 0xffffffffull));
    std::sort(win.begin(), win.end());
    win.erase(std::unique(win.begin(), win.end()), win.end());
    for (size_t base = 0; base < win.size(); base += (size_t)e->batch_cap) {
      const int64_t count = std::min<int64_t>(e->batch_cap, (int64_t)(win.size() - base));
      og.resize((size_t)count * 12);
      std::vector<float> wclip((size_t)count);
      for (int64_t s = 0; s < count; ++s) {
        memcpy(&og[12 * (size_t)s], &e->h_og[12 * (size_t)win[base + s]], 48);
        const size_t at = (size_t)(std::lower_bound(oids.begin(), oids.end(), win[base + s]) - oids.begin());
        wclip[(size_t)s] = clips[at];
      }
      HIP_TRY(hipMemcpyAsync(e->d_og, og.data(), 48 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_clip, wclip.data(), 4 * count, hipMemcpyHostToDevice, e->stream));
      rc = run_batch(e, count, true, e->d_clip);
      if (rc) return rc;
      ssd.resize((size_t)count * rn);
      xf.resize((size_t)count * rn * 12);
      pairs.resize((size_t)count * rn * n_max);
      HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_ssd, 4 * ssd.size(), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(xf.data(), e->d_xf, 4 * xf.size(), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipMemcpyAsync(pairs.data(), e->d_pairs, sizeof(short2) * pairs.size(), hipMemcpyDeviceToHost, e->stream));
      HIP_TRY(hipStreamSynchronize(e->stream));
      for (int k = 0; k < rn; ++k) {
        if (keys[k] == ~0ull) continue;
        const uint32_t o = (uint32_t)(keys[k] & 0xffffffffull);
        const auto it = std::lower_bound(win.begin() + base, win.begin() + base + count, o);
        if (it == win.begin() + base + count || *it != o) continue;
        const size_t rec = (size_t)(it - (win.begin() + base)) * rn + k;
        t.s(r, k) = ssd[rec];
        memcpy(t.x(r, k), &xf[rec * 12], 48);
        memcpy(t.p(r, k), &pairs[rec * n_max], sizeof(short2) * (size_t)(sp.n_min + k));
      }
    }
  }
  if (e->of) e->clip_now = clip_after(e, e->of_calls);
  e->ranks_current = true;
  return 0;
}

extern "C" int minrms_engine_finalize_search(minrms_engine *e)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (!e->searched) return fail(MINRMS_E_STATE, "finalize_search before search");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->device_failed) return check_device_error(e);
  const int rn = e->sp.range_n;
  std::vector<uint64_t> keys(rn);
  HIP_TRY(hipMemcpyAsync(keys.data(), e->d_keys, 8 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  // distinct winning orientations
  std::vector<uint32_t> oids;
  for (int k = 0; k < rn; ++k)
    if (keys[k] != ~0ull) oids.push_back((uint32_t)(keys[k] & 0xffffffffull));
  std::sort(oids.begin(), oids.end());
  oids.erase(std::unique(oids.begin(), oids.end()), oids.end());
  const int64_t n_og = (int64_t)e->h_og.size() / 12;
  int found = 0;
  for (size_t base = 0; base < oids.size(); base += (size_t)e->batch_cap) {
    const int64_t count = std::min<int64_t>(e->batch_cap, (int64_t)oids.size() - (int64_t)base);
    std::vector<float> og((size_t)count * 12), clips((size_t)count);
    for (int64_t s = 0; s < count; ++s) {
      if ((int64_t)oids[base + s] >= n_og) return fail(MINRMS_E_STATE, "winner o_id %u outside the orientation list", oids[base + s]);
      memcpy(&og[12 * s], &e->h_og[12 * (size_t)oids[base + s]], 48);
      // the cutoff the orientation was searched with
      clips[(size_t)s] = e->of ? clip_after(e, e->of_origin_calls + ((int64_t)oids[base + s] - e->of_origin_start) + 1)
                               : e->sp.clip;
    }
    std::vector<int32_t> slot_of_k(rn, -1);
    for (int k = 0; k < rn; ++k) {
      if (keys[k] == ~0ull) continue;
      const uint32_t o = (uint32_t)(keys[k] & 0xffffffffull);
      const auto it = std::lower_bound(oids.begin() + base, oids.begin() + base + count, o);
      if (it != oids.begin() + base + count && *it == o) { slot_of_k[k] = (int32_t)(it - (oids.begin() + base)); ++found; }
    }
    HIP_TRY(hipMemcpyAsync(e->d_og, og.data(), 48 * count, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_slot_of_k, slot_of_k.data(), 4 * rn, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipMemcpyAsync(e->d_clip, clips.data(), 4 * count, hipMemcpyHostToDevice, e->stream));
    int rc = run_batch(e, count, true, e->d_clip);
    if (rc) return rc;
    hipLaunchKernelGGL(k_pick, dim3(rn), dim3(64), 0, e->stream, e->sp, e->cfg.n_max, e->d_slot_of_k,
                       e->d_keys, e->d_ssd, e->d_xf, e->d_pairs, e->d_best_ssd, e->d_best_xf,
                       e->d_best_pairs, e->d_winner);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  e->tables_current = true;
  if (found != 0 && e->num_solutions > 1) {
    int rc = e->alt_by_pairs ? alt_fill_ranks_by_pairs(e) : alt_fill_ranks(e);
    if (rc) return rc;
  }
  { int rc = check_device_error(e); if (rc) return rc; }     // the winners were re-solved by the cooperative kernel
  if (found == 0)
    return fail(MINRMS_E_NOSOLUTION,
                "Error: No solutions were found satisfying your criteria.\n"
                "       The restrictions you placed on accepting solutions were\n"
                "       too strict.  Try using a lower minimum-number-of-matches,\n"
                "       a larger maximum distance cutoff, for example.");
  return MINRMS_OK;
}

// One DP table per host round trip: the literal loop.  Used when the batch
// buffers cannot hold one slot per n.
static int refine_one_at_a_time(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats)
{
  const int rn = e->sp.range_n;
  std::vector<float> ssd(rn);
  std::vector<unsigned char> need(rn);
  HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_best_ssd, 4 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemsetAsync(e->d_counters, 0, 64, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int k = 0; k < rn; ++k) need[k] = ssd[k] < MINRMS_UPPER_BOUND;   // :2348-2358
  HIP_TRY(hipMemcpyAsync(e->d_need, need.data(), rn, hipMemcpyHostToDevice, e->stream));
  int64_t it, dp_runs = 0;
  for (it = 1; it <= max_iters || max_iters == -1; ++it) {
    // FindMedian, lib/global_utils/simple_numeric_utils.h:350-376
    int in_set = 0;
    for (int k = 0; k < rn; ++k) in_set += need[k];
    if (in_set == 0) break;
    const int half = (in_set + 1) / 2;
    int k = -1, seen = 0;
    do { ++k; seen += need[k]; } while (seen < half);
    hipLaunchKernelGGL(k_refit_entry, dim3(1), dim3(64), 0, e->stream, e->sp, e->cfg.n_max, k, e->d_xyzA,
                       e->d_xyzB, e->d_best_pairs, e->d_og);
    int rc = run_batch(e, 1, true);
    if (rc) return rc;
    ++dp_runs;
    hipLaunchKernelGGL(k_refine_apply, dim3(1), dim3(256), 0, e->stream, e->sp, e->cfg.n_max, k,
                       (long long)(-(2 + it)), e->d_cost, e->d_ssd, e->d_xf, e->d_pairs, e->d_best_ssd,
                       e->d_best_xf, e->d_best_pairs, e->d_winner, e->d_need, e->d_counters);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(need.data(), e->d_need, rn, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  long long counters[8] = {0};
  HIP_TRY(hipMemcpyAsync(counters, e->d_counters, 64, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  { int rc = check_device_error(e); if (rc) return rc; }
  if (stats) {
    stats->num_iterations = it;
    stats->num_new_solutions = counters[0];
    stats->dp_runs = dp_runs;
    stats->rounds = dp_runs;
  }
  return MINRMS_OK;
}

// Refinement with the rounds chained on the device: the round kernel replays
// what was prepared and itself picks the entries of the next round, so the
// host only enqueues rounds (grids sized for the largest round, surplus
// workgroups leave at once) and looks at the control block every few rounds.
static int refine_device_rounds(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats)
{
  const int rn = e->sp.range_n;
  const SearchParams &sp = e->sp;
  hipStream_t st = e->stream;
  std::vector<float> ssd(rn);
  std::vector<unsigned char> need(rn);
  HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_best_ssd, 4 * rn, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int k = 0; k < rn; ++k) need[k] = ssd[k] < MINRMS_UPPER_BOUND;   // :2348-2358
  // Most rounds use two or three of the tables they prepare, and the grids of a
  // round are sized for the largest number a round may prepare (the host does
  // not know the count when it enqueues): a limit of 12 instead of 64 measured
  // 6 % faster at 150x150 and no slower from 100 to 400 residues.
  int spec_limit = 12;
  if (const char *v = getenv("MINRMS_REFINE_SPEC_MAX")) spec_limit = std::max(1, atoi(v));   // tuning hook
  const int spec_cap = (int)std::min<int64_t>(std::min<int64_t>(spec_limit, e->batch_cap), e->coop_max_batch);
  RefineCtrl ctrl;
  memset(&ctrl, 0, sizeof ctrl);
  ctrl.it = 1;
  ctrl.max_iters = max_iters;
  ctrl.spec = std::min(16, spec_cap);
  ctrl.spec_min = std::min(4, spec_cap);
  ctrl.spec_max = spec_cap;
  if (const char *v = getenv("MINRMS_REFINE_SPEC"))
    ctrl.spec = ctrl.spec_min = ctrl.spec_max = (int)std::max<int64_t>(1, std::min<int64_t>(spec_cap, atoll(v)));
  ctrl.device_plan = 1;
  HIP_TRY(hipMemcpyAsync(e->d_need, need.data(), rn, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(e->d_ctrl, &ctrl, sizeof ctrl, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(e->d_slot_of_k, 0xFF, 4 * (size_t)rn, st));     // nothing prepared yet
  auto round_kernel = [&]() {
    hipLaunchKernelGGL(k_refine_round, dim3(1), dim3(256), 0, st, sp, e->cfg.n_max, e->d_slot_of_k, e->d_cost,
                       e->d_ssd, e->d_xf, e->d_pairs, e->d_best_ssd, e->d_best_xf, e->d_best_pairs,
                       e->d_winner, e->d_need, e->d_ctrl, e->d_slot_of_k, e->d_klist);
  };
  round_kernel();                       // plans the first round
  const int *live = &e->d_ctrl->count;
  const DpLayout &L = e->Lc;
  const size_t trace_lds = MINRMS_MAX_BLOCKS * sizeof(BlockRef) + fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)L.n_max;
  // The host looks at the control block after every few rounds.  So that the
  // GPU does not sit idle while it does, the next few rounds are enqueued
  // BEFORE the host waits for the previous ones; when the loop has ended, the
  // rounds enqueued in vain find ctrl.done / a live count of 0 and leave at once.
  auto enqueue_rounds = [&](int rounds, RefineCtrl *snapshot, hipEvent_t done_event) -> int {
    for (int q = 0; q < rounds; ++q) {
      const float *og = e->d_best_xf;
      const int32_t *rows = e->d_klist;
      if (!sp.rws) {
        hipLaunchKernelGGL(k_refit_entries, dim3(spec_cap), dim3(64), fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)e->cfg.n_max, st,
                           sp, e->cfg.n_max, spec_cap, e->d_klist, e->d_xyzA, e->d_xyzB, e->d_best_pairs,
                           e->d_og, live);
        og = e->d_og;
        rows = nullptr;
      }
      launch_cij(sp, spec_cap, st, L.c_floats, e->d_xyzA, e->d_xyzB, og, e->d_C, e->clip_now, nullptr, rows, live);
      int rc = launch_coop(e, spec_cap, live);
      if (rc) return rc;
      hipLaunchKernelGGL(k_trace_fit_wave, dim3((unsigned)(spec_cap * rn)), dim3(64), trace_lds, st, L, sp,
                         (int64_t)spec_cap, e->d_bp, e->d_cost, e->d_xyzA, e->d_xyzB, e->d_og, e->d_pairs,
                         e->d_ssd, e->d_xf, live);
      round_kernel();
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(snapshot, e->d_ctrl, sizeof(RefineCtrl), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(done_event, st));
    return 0;
  };
  if (!e->h_ctrl) {
    std::lock_guard<std::mutex> lock(pool::mu);
    for (size_t q = 0; q < pool::host_slots.size(); ++q) {
      if (pool::host_slots[q].device != e->cfg.device) continue;
      const pool::HostSlot hs = pool::host_slots[q];
      pool::host_slots.erase(pool::host_slots.begin() + (long)q);
      e->h_ctrl = (RefineCtrl *)hs.ptr;
      e->ctrl_event[0] = hs.ev[0];
      e->ctrl_event[1] = hs.ev[1];
      break;
    }
  }
  if (!e->h_ctrl) {
    HIP_TRY(hipHostMalloc((void **)&e->h_ctrl, 2 * sizeof(RefineCtrl), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&e->ctrl_event[0], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&e->ctrl_event[1], hipEventDisableTiming));
  }
  // With other engines at work on the GPU the gaps are theirs, and rounds
  // enqueued in vain would only be in their way: then look after every 8 rounds
  // and enqueue nothing ahead.
  const bool ahead = live_engines.load() <= 1;
  const int rounds_per_look = ahead ? 4 : 8;
  int slot = 0;
  { int rc = enqueue_rounds(rounds_per_look, &e->h_ctrl[0], e->ctrl_event[0]); if (rc) return rc; }
  for (;;) {
    if (ahead) { int rc = enqueue_rounds(rounds_per_look, &e->h_ctrl[slot ^ 1], e->ctrl_event[slot ^ 1]); if (rc) return rc; }
    HIP_TRY(hipEventSynchronize(e->ctrl_event[slot]));
    if (e->h_ctrl[slot].done) break;
    if (ahead) slot ^= 1;
    else { int rc = enqueue_rounds(rounds_per_look, &e->h_ctrl[slot], e->ctrl_event[slot]); if (rc) return rc; }
  }
  HIP_TRY(hipMemcpyAsync(&ctrl, e->d_ctrl, sizeof ctrl, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  int rc = check_device_error(e);
  if (rc) return rc;
  if (ctrl.done == 2) return fail(MINRMS_E_STATE, "refinement made no progress in a round (internal error)");
  if (stats) {
    stats->num_iterations = ctrl.it;
    stats->num_new_solutions = ctrl.updates;
    stats->dp_runs = ctrl.dp_runs;
    stats->rounds = ctrl.rounds;
  }
  if (getenv("MINRMS_REFINE_DEBUG")) {
    fprintf(stderr, "refine rounds %lld: ended because next entry unprepared %lld, changed in the round %lld, all used %lld; used per round:",
            ctrl.rounds, ctrl.stop_unprepared, ctrl.stop_stale, ctrl.stop_all_used);
    for (int q = 0; q < 16; ++q) fprintf(stderr, " %lld", ctrl.applied_hist[q]);
    fprintf(stderr, "\n");
  }
  return MINRMS_OK;
}

static int refine_best(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats);

// Refinement of rank r >= 1 (REFINE_ALL, search_or_dyn3d.cc:2344-2640): one
// alignment at a time, and an iteration only ever replaces the alignment it
// started from (m == n, :2596-2613).
static int alt_refine_rank(minrms_engine *e, int r, int64_t max_iters, int64_t *iterations, int64_t *updates)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n, n_max = e->cfg.n_max;
  alt::RankTables &t = e->ranks;
  std::vector<unsigned char> need(rn);
  for (int k = 0; k < rn; ++k) need[k] = t.s(r, k) < MINRMS_UPPER_BOUND;
  std::vector<float> cost(n_max), ssd(rn), xf(12);
  std::vector<short2> pairs(n_max);
  int64_t it;
  for (it = 1; it <= max_iters || max_iters == -1; ++it) {
    int in_set = 0;
    for (int k = 0; k < rn; ++k) in_set += need[k];
    if (in_set == 0) break;
    const int half = (in_set + 1) / 2;
    int k = -1, seen = 0;
    do { ++k; seen += need[k]; } while (seen < half);
    const int n = sp.n_min + k;
    if (sp.rws) {
      // the stored transform is the least-squares transform of the stored alignment
      HIP_TRY(hipMemcpyAsync(e->d_og, t.x(r, k), 48, hipMemcpyHostToDevice, e->stream));
    } else {
      // :2401-2407.  Rank 0 lives on the host while the other ranks are
      // refined, so its device table serves as the staging area for the refit.
      const int32_t kk = k;
      HIP_TRY(hipMemcpyAsync(e->d_best_pairs + (size_t)k * n_max, t.p(r, k), sizeof(short2) * (size_t)n,
                             hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_klist, &kk, 4, hipMemcpyHostToDevice, e->stream));
      hipLaunchKernelGGL(k_refit_entries, dim3(1), dim3(64), fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)n_max, e->stream, sp, n_max,
                         1, e->d_klist, e->d_xyzA, e->d_xyzB, e->d_best_pairs, e->d_og, nullptr);
    }
    int rc = run_batch(e, 1, true);
    if (rc) return rc;
    const size_t rec = (size_t)k;
    HIP_TRY(hipMemcpyAsync(cost.data(), e->d_cost, 4 * n_max, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_ssd, 4 * rn, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(xf.data(), e->d_xf + rec * 12, 48, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(pairs.data(), e->d_pairs + rec * n_max, sizeof(short2) * n, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    bool changed = false;
    const int stop = first_break_host(sp, cost.data(), ssd.data());
    if (n >= 2 && n < stop && ssd[k] < t.s(r, k)) {
      changed = memcmp(t.p(r, k), pairs.data(), sizeof(short2) * (size_t)n) != 0;
      t.s(r, k) = ssd[k];
      memcpy(t.x(r, k), xf.data(), 48);
      memcpy(t.p(r, k), pairs.data(), sizeof(short2) * (size_t)n);
      ++*updates;
    }
    need[k] = changed;
  }
  { int rc = check_device_error(e); if (rc) return rc; }
  *iterations = it;
  return 0;
}

extern "C" int minrms_engine_refine(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (e->num_solutions <= 1) return refine_best(e, max_iters, stats);
  // Refine() with alternates: rank 0 as always; the other ranks if asked
  // (REFINE_ALL); then the consistency check over all ranks (:2648-2654).
  minrms_refine_stats st;
  memset(&st, 0, sizeof st);
  int rc = refine_best(e, max_iters, &st);
  if (rc) return rc;
  if (!e->ranks_current) return fail(MINRMS_E_STATE, "alternate tables are not up to date");
  rc = alt_pull_rank0(e);
  if (rc) return rc;
  if (e->cfg.refine_all) {
    for (int r = 1; r < e->num_solutions; ++r) {
      int64_t its = 0, ups = 0;
      rc = alt_refine_rank(e, r, max_iters, &its, &ups);
      if (rc) return rc;
      st.num_iterations = its;                 // the reference reports the last rank's count
      st.num_new_solutions += ups;
      st.dp_runs += its - 1;
      st.rounds += its - 1;
    }
  }
  alt::check_consistency(e->ranks, e->sp.n_min, e->sp.nbb, e->h_xyzB.data(), e->sp.N2 * e->sp.nbb,
                         e->cfg.alt_min_3d_difference, e->cfg.alt_rmsd_tolerance, MINRMS_UPPER_BOUND,
                         e->alt_by_pairs, e->cfg.alt_max_pairs_common);
  rc = alt_push_rank0(e);
  if (rc) return rc;
  if (stats) *stats = st;
  return MINRMS_OK;
}

static int refine_best(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (!e->searched) return fail(MINRMS_E_STATE, "refine before search");
  HIP_TRY(hipSetDevice(e->cfg.device));
  { int rc = check_device_error(e); if (rc) return rc; }
  if (!e->tables_current) {
    int rc = minrms_engine_finalize_search(e);
    if (rc) return rc;
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int rn = e->sp.range_n;
  if (rn > e->batch_cap || getenv("MINRMS_REFINE_SEQUENTIAL")) return refine_one_at_a_time(e, max_iters, stats);
  if (e->coop_pb != 0 && !getenv("MINRMS_REFINE_HOST_ROUNDS")) return refine_device_rounds(e, max_iters, stats);

  std::vector<float> ssd(rn);
  std::vector<unsigned char> need(rn);
  HIP_TRY(hipMemcpyAsync(ssd.data(), e->d_best_ssd, 4 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int k = 0; k < rn; ++k) need[k] = ssd[k] < MINRMS_UPPER_BOUND;   // :2348-2358
  RefineCtrl ctrl;
  memset(&ctrl, 0, sizeof ctrl);
  ctrl.it = 1;
  ctrl.max_iters = max_iters;
  HIP_TRY(hipMemcpyAsync(e->d_need, need.data(), rn, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->d_ctrl, &ctrl, sizeof ctrl, hipMemcpyHostToDevice, e->stream));
  int64_t dp_runs = 0, rounds = 0;
  std::vector<int32_t> klist, slot_of_k(rn);
  std::vector<unsigned char> pending;
  // entries prepared per round: grows while rounds use up what was prepared,
  // shrinks while most of it is thrown away
  int64_t spec = 16, spec_min = 4, spec_max = e->batch_cap;
  if (const char *v = getenv("MINRMS_REFINE_SPEC")) { spec = spec_min = spec_max = std::max<int64_t>(1, atoll(v)); }
  spec = std::min(spec, e->batch_cap);
  for (;;) {
    // The entries a round prepares: the next `spec` entries in the order the
    // loop will visit them if no table entry changes any more (repeated
    // median of the set still to refine).  An entry the replay reaches but
    // finds unprepared, or changed since, ends the round.
    klist.clear();
    std::fill(slot_of_k.begin(), slot_of_k.end(), -1);
    {
      pending = need;
      int in_set = 0;
      for (int k = 0; k < rn; ++k) in_set += pending[k];
      while (in_set > 0 && (int64_t)klist.size() < spec) {
        const int half = (in_set + 1) / 2;
        int k = -1, seen = 0;
        do { ++k; seen += pending[k]; } while (seen < half);
        slot_of_k[k] = (int32_t)klist.size();
        klist.push_back(k);
        pending[k] = 0;
        --in_set;
      }
    }
    const int count = (int)klist.size();
    if (count > 0 && (ctrl.it <= max_iters || max_iters == -1)) {
      HIP_TRY(hipMemcpyAsync(e->d_klist, klist.data(), 4 * count, hipMemcpyHostToDevice, e->stream));
      HIP_TRY(hipMemcpyAsync(e->d_slot_of_k, slot_of_k.data(), 4 * rn, hipMemcpyHostToDevice, e->stream));
      int rc;
      if (e->sp.rws) {
        // With refine_while_searching every table entry already holds the
        // least-squares transform of its own alignment (it was produced by the
        // same CalcMinRMSD that :2401-2407 would repeat): no refit needed.
        rc = run_batch(e, count, true, nullptr, e->d_best_xf, e->d_klist);
      } else {
        hipLaunchKernelGGL(k_refit_entries, dim3(count), dim3(64), fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)e->cfg.n_max,
                           e->stream, e->sp, e->cfg.n_max, count, e->d_klist, e->d_xyzA, e->d_xyzB,
                           e->d_best_pairs, e->d_og, nullptr);
        rc = run_batch(e, count, true);
      }
      if (rc) return rc;
      dp_runs += count;
    } else {
      HIP_TRY(hipMemcpyAsync(e->d_slot_of_k, slot_of_k.data(), 4 * rn, hipMemcpyHostToDevice, e->stream));
    }
    ++rounds;
    hipLaunchKernelGGL(k_refine_round, dim3(1), dim3(256), 0, e->stream, e->sp, e->cfg.n_max, e->d_slot_of_k,
                       e->d_cost, e->d_ssd, e->d_xf, e->d_pairs, e->d_best_ssd, e->d_best_xf,
                       e->d_best_pairs, e->d_winner, e->d_need, e->d_ctrl, nullptr, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(need.data(), e->d_need, rn, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(&ctrl, e->d_ctrl, sizeof ctrl, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (ctrl.done) break;
    if (ctrl.applied == 0 && count > 0)
      return fail(MINRMS_E_STATE, "refinement made no progress in a round (internal error)");
    if (ctrl.applied >= count) spec = std::min(spec_max, spec * 2);
    else if (4 * ctrl.applied < count) spec = std::max(spec_min, spec / 2);
  }
  { int rc = check_device_error(e); if (rc) return rc; }
  if (stats) {
    stats->num_iterations = ctrl.it;
    stats->num_new_solutions = ctrl.updates;
    stats->dp_runs = dp_runs;
    stats->rounds = rounds;
  }
  return MINRMS_OK;
}

// A cooperative DP kernel whose bounded wait for a neighbour task ran out
// raises d_err and finishes on whatever data it has: everything computed since
// is unusable.  Called (device current) wherever the host has just waited for
// the stream and is about to read or publish results; the failure sticks until
// minrms_engine_reset().
static int check_device_error(minrms_engine *e)
{
  if (!e->device_failed) {
    int err = 0;
    HIP_TRY(hipMemcpyAsync(&err, e->d_err, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (err) e->device_failed = true;
  }
  if (e->device_failed) {
    e->tables_current = false;
    e->ranks_current = false;
    return fail(MINRMS_E_DEVICE, "cooperative DP kernel gave up waiting for a neighbour task: the tables and keys "
                                 "of this engine are invalid until minrms_engine_reset()");
  }
  return 0;
}

extern "C" int minrms_engine_results(minrms_engine *e, float *ssd, float *xforms, int16_t *pairs,
                                     int64_t *winner_oid)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->searched && !e->tables_current && !e->device_failed) {
    int rc = minrms_engine_finalize_search(e);
    if (rc && rc != MINRMS_E_NOSOLUTION) return rc;
  }
  { int rc0 = check_device_error(e); if (rc0) return rc0; }
  const int rn = e->sp.range_n;
  if (ssd) HIP_TRY(hipMemcpyAsync(ssd, e->d_best_ssd, 4 * rn, hipMemcpyDeviceToHost, e->stream));
  if (xforms) HIP_TRY(hipMemcpyAsync(xforms, e->d_best_xf, 48 * rn, hipMemcpyDeviceToHost, e->stream));
  if (pairs) HIP_TRY(hipMemcpyAsync(pairs, e->d_best_pairs, sizeof(short2) * (size_t)rn * e->cfg.n_max, hipMemcpyDeviceToHost, e->stream));
  if (winner_oid) HIP_TRY(hipMemcpyAsync(winner_oid, e->d_winner, 8 * rn, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MINRMS_OK;
}

extern "C" int minrms_engine_results_rank(minrms_engine *e, int32_t rank, float *ssd, float *xforms,
                                          int16_t *pairs)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (rank < 0 || rank >= e->num_solutions)
    return fail(MINRMS_E_ARG, "rank %d outside 0..%d", rank, e->num_solutions - 1);
  if (rank == 0) return minrms_engine_results(e, ssd, xforms, pairs, nullptr);
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->searched && !e->tables_current && !e->device_failed) {
    int rc = minrms_engine_finalize_search(e);
    if (rc) return rc;
  }
  { int rc0 = check_device_error(e); if (rc0) return rc0; }
  if (!e->ranks_current) return fail(MINRMS_E_STATE, "results_rank before search");
  alt::RankTables &t = e->ranks;
  const int rn = e->sp.range_n;
  if (ssd) memcpy(ssd, &t.s(rank, 0), 4 * (size_t)rn);
  if (xforms) memcpy(xforms, t.x(rank, 0), 48 * (size_t)rn);
  if (pairs) memcpy(pairs, t.p(rank, 0), 4 * (size_t)rn * e->cfg.n_max);
  return MINRMS_OK;
}

#ifdef MINRMS_COOP_STAMPS
// diagnostic build only: the 64 words of the error block (word 0 = give-up flag, 8.. = task time stamps)
extern "C" int minrms_engine_debug_read_err(minrms_engine *e, int *out64)
{
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipMemcpyAsync(out64, e->d_err, 256, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return 0;
}
// cycles of the wave-per-chain tracer for the longest chain of slot 0:
// total, waiting for cone fetches, walking, refit, number of fetches, of walk steps, up to the refit
extern "C" int minrms_engine_debug_read_trace_stamps(minrms_engine *e, unsigned long long *out8)
{
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_trace_stamps), 64));
  unsigned long long f[4];
  HIP_TRY(hipMemcpyFromSymbol(f, HIP_SYMBOL(fit::g_fit_stamps), 32));
  out8[7] = f[2];                      // the eigen-solve and what follows it
  unsigned long long r[8];
  HIP_TRY(hipMemcpyFromSymbol(r, HIP_SYMBOL(g_round_stamps), 64));
  fprintf(stderr, "round kernel, %llu launches: start %llu, replay %llu (%llu iterations), plan and write-back %llu cycles per launch; %llu tables planned\n",
          r[3], r[3] ? r[0] / r[3] : 0, r[3] ? r[1] / r[3] : 0, r[4], r[3] ? r[2] / r[3] : 0, r[5]);
  out8[3] = f[0] << 32 | (f[1] & 0xffffffffull);   // pass 1, pass 2 of the superposition
  return 0;
}
#endif

extern "C" int minrms_engine_enable_timing(minrms_engine *e, int32_t on)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  e->timing = on != 0;
  return MINRMS_OK;
}

extern "C" int minrms_engine_dp_block_time(minrms_engine *e, int32_t block, double *ms,
                                           int64_t *launches, int64_t *orientations,
                                           int64_t *cells_per_orientation)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (block < 0 || block >= e->L.NB) return fail(MINRMS_E_ARG, "no layer block %d", block);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (const minrms_engine::Stamp &s : e->stamps) {     // drain pending event pairs
    for (int q = 0; q < s.nb; ++q) {
      float t = 0.f;
      if (s.blocks[q] < 0) continue;                      // between two layer blocks: traceback, not DP
      HIP_TRY(hipEventElapsedTime(&t, e->ev_pool[s.first + q], e->ev_pool[s.first + q + 1]));
      e->blk_ms[s.blocks[q]] += t;
      e->blk_launches[s.blocks[q]] += 1;
    }
    for (int b = 0; b < e->L.NB; ++b) e->blk_orients[b] += s.count;
  }
  e->stamps.clear();
  e->ev_used = 0;
  if (ms) *ms = e->blk_ms[block];
  if (launches) *launches = e->blk_launches[block];
  if (orientations) *orientations = e->blk_orients[block];
  if (cells_per_orientation) {
    // cells of the reference's table that live in this block's layers
    const minrms_cfg &c = e->cfg;
    const int wA = c.N1 - c.n_min + 1, wB = c.N2 - c.n_min + 1;
    int64_t cells = 0;
    for (int n = MINRMS_LAYERS_PER_BLOCK * block + 1;
         n <= std::min(c.n_max, MINRMS_LAYERS_PER_BLOCK * (block + 1)); ++n)
      cells += (int64_t)(std::min(c.N1, n + wA - 1) - n + 1) * (std::min(c.N2, n + wB - 1) - n + 1);
    *cells_per_orientation = cells;
  }
  e->blk_ms[block] = 0.0; e->blk_launches[block] = 0; e->blk_orients[block] = 0;
  return MINRMS_OK;
}

extern "C" int minrms_engine_num_blocks(minrms_engine *e)
{
  return e ? e->L.NB : 0;
}

extern "C" int minrms_engine_dp_block_shape(minrms_engine *e, int32_t block, int32_t *rows, int32_t *panels,
                                            int32_t *panel_width)
{
  if (!e || block < 0 || block >= e->L.NB) return fail(MINRMS_E_ARG, "bad argument");
  const BlockDesc &bd = e->L.blk[block];
  if (rows) *rows = bd.nrows;
  if (panels) *panels = bd.P;
  if (panel_width) *panel_width = bd.PB;
  return MINRMS_OK;
}

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
static bool lds_allowed[64];

// Device address of this GPU's copy of the array; `device` must be current.
// In host code the array's name is only a handle for this lookup, never an address.
static int scratch_of(int device, unsigned char **out)
{
  if (device < 0 || device >= 64) return fail(MINRMS_E_DEVICE, "no HIP device %d", device);
  std::lock_guard<std::mutex> lock(mu);
  if (!ptr_of_device[device]) {
    void *p = nullptr;
    const hipError_t r = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (r != hipSuccess || !p)
      return fail(MINRMS_E_DEVICE, "hipGetSymbolAddress(scratch) failed on device %d: %s", device, hipGetErrorString(r));
    ptr_of_device[device] = (unsigned char *)p;
  }
  *out = ptr_of_device[device];
  return 0;
}
}  // namespace noalloc

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
  HIP_TRY(hipSetDevice(c.device));
  unsigned char *base = nullptr;
  { int rc = noalloc::scratch_of(c.device, &base); if (rc) return rc; }
  if (v.coop_pb != 0 && v.coop_groups.n_groups > 1 && !noalloc::lds_allowed[c.device]) {
    int rc = allow_large_lds();
    if (rc) return rc;
    noalloc::lds_allowed[c.device] = true;
  }

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
  v.timing = false;

  // The array keeps what the last call left in it, carved up for that call's
  // sizes: the row counters, the error word and the keys must start clean, and
  // the distance tables and planes must hold floats (their padding is read).
  hipLaunchKernelGGL(k_noalloc_clear, dim3(256), dim3(256), 0, v.stream, v.d_keys, rn,
                     (uint32_t *)v.d_C, L.c_floats * count, (uint32_t *)v.d_planes, plane_floats * count,
                     (uint32_t *)v.d_gprog, (int64_t)((unsigned char *)v.d_keys - (unsigned char *)v.d_gprog) / 4);
  int rc = run_batch(&v, count, true, nullptr, nullptr, nullptr, true, false);
  v.stream = nullptr;                        // the caller's stream: not ours to destroy
  if (rc) return rc;
  hipLaunchKernelGGL(k_reduce, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, v.sp, count,
                     (int64_t)0, (const uint32_t *)nullptr, v.d_cost, v.d_ssd, rn, v.d_keys);
  hipLaunchKernelGGL(k_pick_direct, dim3(rn), dim3(64), 0, (hipStream_t)stream, v.sp, c.n_max, v.d_keys, v.d_err,
                     v.d_ssd, v.d_xf, v.d_pairs, d_ssd, d_xforms, (short2 *)d_pairs, (long long *)d_winner_oid);
  HIP_TRY(hipGetLastError());
  return MINRMS_OK;
}

// ---------------------------------------------------------------------------
// orientation generation
// ---------------------------------------------------------------------------
namespace {
// glibc rand() in its default state (srand(1)): additive feedback generator of
// degree 31, separation 3, 310 values discarded after seeding.
struct GlibcRand {
  int32_t r[31];
  int f = 3, b = 0;
  GlibcRand()
  {
    int32_t word = 1;
    r[0] = word;
    for (int i = 1; i < 31; ++i) {
      const int64_t hi = word / 127773, lo = word % 127773;
      int64_t w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      r[i] = word;
    }
    for (int i = 0; i < 310; ++i) next();
  }
  int32_t next()
  {
    const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    f = (f + 1) % 31;
    b = (b + 1) % 31;
    return (int32_t)(v >> 1);
  }
};
}  // namespace

extern "C" int minrms_engine_gen_orientations(minrms_engine *e, int32_t frag_len, float cutoff_rmsd,
                                              int32_t n_int, const int32_t *ip, int32_t shuffle,
                                              float *out_og, int64_t cap, int64_t *count)
{
  if (!e || !count || (n_int > 0 && !ip)) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int N1 = e->cfg.N1, N2 = e->cfg.N2, nbb = e->cfg.num_bb, L = frag_len;
  if (L < 1 || L > N1 || L > N2) return fail(MINRMS_E_ARG, "ERROR: fragment-size cannot exceed length of sequence!");
  const int w1 = N1 - e->cfg.n_min + 1, w2 = N2 - e->cfg.n_min + 1;   // or_gen.cc:253-254
  std::vector<unsigned char> mask((size_t)N1 * N2, 0);
  for (int p = 0; p < n_int; ++p) {
    const int s1 = ip[4 * p], e1 = ip[4 * p + 1], s2 = ip[4 * p + 2], e2 = ip[4 * p + 3];
    if (e1 < s1 || e2 < s2) return fail(MINRMS_E_ARG, "Subset-Match Interval cannot have start > end.");
    if (s1 < 0 || s2 < 0) return fail(MINRMS_E_ARG, "Subset-Match Interval cannot begin before start of sequence.");
    if (e1 >= N1 || e2 >= N2) return fail(MINRMS_E_ARG, "ERROR: fragment-size cannot exceed length of sequence!");
    if (s1 > e1 - L + 1 || s2 > e2 - L + 1) continue;                 // interval too small: skipped
    for (int I = s1; I <= e1 - L + 1; ++I) {
      const int jlo = std::max(s2, I - (w1 - 1));                     // or_gen.cc:451-464
      const int jhi = std::min(I + (w2 - 1), e2 - L + 1);
      for (int J = jlo; J <= jhi; ++J) mask[(size_t)I * N2 + J] = 1;
    }
  }
  std::vector<int32_t> IJ;
  for (int I = 0; I < N1; ++I)
    for (int J = 0; J < N2; ++J)
      if (mask[(size_t)I * N2 + J]) { IJ.push_back(I); IJ.push_back(J); }
  const int64_t np = (int64_t)IJ.size() / 2;
  std::vector<float> og((size_t)np * 12), rmsd((size_t)np);
  if (np > 0) {
    // staging for the largest possible list (every (I,J)), taken once per engine
    if (!e->d_gen) {
      const size_t want = (size_t)64 * N1 * N2 + 256;
      size_t got = 0, clean = 0;
      void *ptr = nullptr;
      HIP_TRY(pool::take(e->cfg.device, "d_gen", want, &ptr, &got, &clean));
      e->dev_blocks.push_back(minrms_engine::DevBlock{ptr, got, "d_gen", clean});
      e->d_gen = (unsigned char *)ptr;
    }
    const size_t cells = (size_t)N1 * N2;
    float *d_o = (float *)e->d_gen;                          // [cells][12]
    int32_t *d_IJ = (int32_t *)(e->d_gen + 48 * cells);      // [cells][2]
    float *d_r = (float *)(e->d_gen + 56 * cells);           // [cells]
    HIP_TRY(hipMemcpyAsync(d_IJ, IJ.data(), 8 * np, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_fragment_fit, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, e->stream, nbb, L, np,
                       d_IJ, e->d_xyzA, e->d_xyzB, d_o, d_r);
    HIP_TRY(hipMemcpyAsync(og.data(), d_o, 48 * np, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(rmsd.data(), d_r, 4 * np, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  std::vector<int64_t> keep;
  for (int64_t t = 0; t < np; ++t)
    if (cutoff_rmsd < 0.0f || rmsd[t] <= cutoff_rmsd) keep.push_back(t);   // or_gen.cc:537-541
  if (keep.empty())
    return fail(MINRMS_E_NOSOLUTION,
                "Error: You have requested to generate orientations by fitting together blocks of %d "
                "residues each, however there are no valid regions from which to draw them.", L);
  if (shuffle) {
    // std::random_shuffle as libstdc++ implements it, driven by glibc rand()
    GlibcRand rng;
    for (int64_t i = 1; i < (int64_t)keep.size(); ++i) {
      const int64_t j = (int64_t)rng.next() % (i + 1);
      if (i != j) std::swap(keep[i], keep[j]);
    }
  }
  *count = (int64_t)keep.size();
  if (out_og)
    for (int64_t t = 0; t < std::min<int64_t>(cap, (int64_t)keep.size()); ++t)
      memcpy(out_og + 12 * t, &og[12 * (size_t)keep[t]], 48);
  return MINRMS_OK;
}
