// search.hip -- the orientation sweep around the DP:
//   bin/minrms/search_or_dyn3d.cc:1643-1946  FirstPass  -> minrms_engine_search
//   lib/biopolymer/apply_trans.cc:72-108 + lib/dyn/dyn3d.cc:664-701
//                                            -> k_cij
//   lib/dyn/dyn3d.cc:391-431 + lib/pair_alignment/pair_alignment.cc:714-820
//                                            -> k_trace_fit, k_trace_fit_wave
//   lib/or_gen/orientation_filter.cc:42-130 + lib/dyn/dynNW.cc:147-531
//                                            -> nwf::k_nw_filter (nw_filter.h)
//   bin/minrms/search_or_dyn3d.cc:1825-1835  running per-N minimum -> k_reduce, k_pick
#include "engine_internal.h"
#include "fit_device.h"
#include "nw_filter.h"

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
// mine[TB * word] (the words of a workgroup's TB threads interleaved), 0 = left (j-1), 1 = up (i-1), 2 = match.  (ci, cj) = the cell
// the walk ended in.  Undoing the moves in reverse yields the matched pairs in
// ascending order, the order the reference sums them in; f(i, j) gets the
// 0-based residue indices of each.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// word w of this thread's path string (the words of the workgroup's TB threads interleaved); PL: the strings are in LDS
template <int TB, bool PL>
__device__ __forceinline__ uint32_t path_word(const uint32_t *mine, const int w)
{
  return PL ? ((const lds_u32 *)mine)[w * TB] : mine[(size_t)w * TB];
}
template <int TB, bool PL>
__device__ __forceinline__ void set_path_word(uint32_t *mine, const int w, const uint32_t v)
{
  if (PL) ((lds_u32 *)mine)[w * TB] = v;
  else mine[(size_t)w * TB] = v;
}

template <int TB, bool PL, class F>
__device__ __forceinline__ void replay_path(const uint32_t *mine, int n_moves, int ci, int cj, F &&f)
{
  int q = n_moves - 1;
  while (q >= 0) {
    const uint32_t word = path_word<TB, PL>(mine, q >> 4);
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
// Dynamic LDS: one BlockRef per layer block, the paths [path_words][TB], then (if coords_in_lds)
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

// TB: threads of a workgroup (256: four waves share one copy of the coordinates in LDS, which leaves room
// for 24 waves per CU at 150x150 where workgroups of one wave have room for 18; 64: the finer grain when
// long path strings must stay in LDS); LDSC: both coordinate sets are in LDS; ONE: one atom per
// residue; PL: the path strings are in LDS.
#ifndef MINRMS_TRACE_WAVES
#define MINRMS_TRACE_WAVES 8     // waves per SIMD the compiler fits k_trace_fit's registers to
#endif
template <int TB, bool LDSC, bool ONE, bool PL>
__global__ void __launch_bounds__(TB) __attribute__((amdgpu_waves_per_eu(MINRMS_TRACE_WAVES, MINRMS_TRACE_WAVES)))
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
  uint32_t *lds_path = (uint32_t *)(tf_lds + L.NB * sizeof(BlockRef));
  load_block_refs(L, refs);
  const float *xyzA = xyzA_g, *xyzB = xyzB_g;
  if (LDSC) {
    float *la = (float *)(lds_path + (path_global ? 0 : (size_t)path_words * TB));
    float *lb = la + 3 * sp.N1 * sp.nbb;
    for (int q = threadIdx.x; q < 3 * sp.N1 * sp.nbb; q += TB) la[q] = xyzA_g[q];
    for (int q = threadIdx.x; q < 3 * sp.N2 * sp.nbb; q += TB) lb[q] = xyzB_g[q];
    xyzA = la; xyzB = lb;
  }
  const int lane = threadIdx.x;
  // this thread's path words: mine[TB * w], in LDS or (long paths) in memory
  uint32_t *mine = (path_global ? path_global + (size_t)blockIdx.x * path_words * TB : lds_path) + lane;
  __syncthreads();
  const int64_t tid = (int64_t)blockIdx.x * TB + threadIdx.x;
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
  // The walk is bound by its instruction count (one dependent load per row, but thousands of
  // walks in flight): the cell's place in the table is kept as running 32-bit word offsets (an
  // orientation's table is shorter than 2^30 words) that a move adjusts, instead of being worked
  // out from (n, i, j) in every step.
  int b = (n - 1) / MINRMS_LAYERS_PER_BLOCK;            // layer block of the current cell
  BlockRef bd = refs[b];                                // changes 0-2 times per walk: no LDS read per step
  int ln = n - MINRMS_LAYERS_PER_BLOCK * b;             // lane of layer cn in its block, 1..63
  uint32_t rowstep = (uint32_t)bd.W * 64u;              // words from one row of the block to the next
  uint32_t rowbase = (uint32_t)bd.bp_off + (uint32_t)(ci - bd.j0) * rowstep + (uint32_t)ln;   // word 0 of (row ci, lane ln)
  while (cn > 0) {
    // a well-formed table never leaves the cells i >= n, j >= n; refuse to
    // follow a pointer that would (the result is then marked unusable)
    if (ci < cn || cj < cn) { *ssd_out = __int_as_float(0x7fc00000); return; }
    const int c = cj - bd.j0;
    const int p = c & 15;
    const uint32_t word = bp[rowbase + ((uint32_t)(c >> 4) << 6)];
    // cells p, p-1, ... of the word, lowest bits first: skip the run of "left" moves
    const uint32_t not_left = (~word & 0x55555555u) >> (30 - 2 * p);
    // moves to skip (left moves are zeros in the string), then one move to record:
    // at most 16 positions, so at most one word of the string is completed
    const bool only_left = not_left == 0u;
    const int run = only_left ? p + 1 : (__builtin_ctz(not_left) >> 1);
    cj -= run;
    if (((pos + run) >> 4) != (pos >> 4)) { set_path_word<TB, PL>(mine, pos >> 4, acc); acc = 0u; }
    pos += run;
    if (only_left) continue;
    const uint32_t code = (word >> (30 - 2 * (p - run))) & 3u;
    --ci;
    if (code == 2u) {
      acc |= 1u << (2 * (pos & 15));
      rowbase -= rowstep;
    } else {
      acc |= 2u << (2 * (pos & 15));
      --cn; --cj; --ln;
      rowbase -= rowstep + 1u;
      if (ln == 0 && cn > 0) {
        // layer 63 b is lane 63 of the block below
        --b;
        bd = refs[b];
        ln = MINRMS_LAYERS_PER_BLOCK;
        rowstep = (uint32_t)bd.W * 64u;
        rowbase = (uint32_t)bd.bp_off + (uint32_t)(ci - bd.j0) * rowstep + (uint32_t)ln;
      }
    }
    if ((pos & 15) == 15) { set_path_word<TB, PL>(mine, pos >> 4, acc); acc = 0u; }
    ++pos;
  }
  set_path_word<TB, PL>(mine, pos >> 4, acc);       // the word in progress (pos <= N1+N2 < 16 * path_words)

  if (pairs_all) {
    short2 *pairs = pairs_all + (slot * sp.range_n + k) * (int64_t)L.n_max;
    int m = 0;
    replay_path<TB, PL>(mine, pos, ci, cj, [&](int i, int j) { pairs[m++] = make_short2((short)i, (short)j); });
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
  replay_path<TB, PL>(mine, pos, ci, cj, [&](int i, int j) {
    for (int q = 0; q < nbb; ++q) {
      float f[3], m[3];
      load_atom<LDSC>(xyzA, i * nbb + q, f);
      load_atom<LDSC>(xyzB, j * nbb + q, m);
      S.sum(f, m);
    }
  });
  const int npts = n * nbb;
  S.centre(npts);
  replay_path<TB, PL>(mine, pos, ci, cj, [&](int i, int j) {
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
__device__ unsigned long long g_trace_stamps[8];    // diagnostic build: cycles of the longest chain of slot 0
__device__ unsigned int g_trace_all[2080 * 6];       // ... per chain of slot 0 (by n): total, up to the refit, fetches | hits << 16, pass 1, pass 2, eigen-solve
__device__ unsigned long long g_trace_slowest[2];   // ... and the chain of slot 0 that took longest: cycles << 32 | n; the same with the number of cone fetches
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
  // (both loads at once: each is a round trip to memory, the data comes from the kernels before this one)
  const float cost_early = blockIdx.x < n_slots * sp.range_n && sp.n_min + (int)(blockIdx.x % sp.range_n) >= 2
                               ? cost_all[(blockIdx.x / sp.range_n) * L.n_max + (sp.n_min + (int)(blockIdx.x % sp.range_n) - 1)]
                               : 0.0f;
  if (live_count && (int64_t)blockIdx.x >= (int64_t)*live_count * sp.range_n) return;
  BlockRef *refs = (BlockRef *)tfw_lds;
  float *chunk = (float *)(tfw_lds + L.NB * sizeof(BlockRef));   // [fit::WAVE_LDS_FLOATS]
  short2 *lpairs = (short2 *)(chunk + fit::WAVE_LDS_FLOATS);                  // [n_max]
  const uint32_t lpairs_at = (uint32_t)(uintptr_t)(lds_byte *)tfw_lds +
                             (uint32_t)(L.NB * sizeof(BlockRef) + fit::WAVE_LDS_FLOATS * 4);
  const int lane = threadIdx.x;
  const int64_t rec = blockIdx.x;
  if (rec >= n_slots * sp.range_n) return;
  const int64_t slot = rec / sp.range_n;
  const int k = (int)(rec - slot * sp.range_n);
  const int n = sp.n_min + k;
  float *ssd_out = ssd_all + rec;
  if (n < 2) { if (lane == 0) *ssd_out = MINRMS_UPPER_BOUND; return; }
  const float cost = cost_early;                       // = cost_all[slot * L.n_max + (n - 1)]
  if (cost >= MINRMS_UPPER_BOUND) { if (lane == 0) *ssd_out = cost; return; }
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long ts_begin = MINRMS_TSTAMP();
  unsigned long long ts_fetch = 0, ts_walk = 0, ts_fetches = 0, ts_steps = 0, ts_hits = 0;
#endif
  load_block_refs(L, refs);
  __syncthreads();

  const uint32_t *bp = bp_all + slot * L.bp_words;
  // this lane's cell of the cone: lane = di(di+1)/2 + dn, 0 <= dn <= di
  int e_di = 0;
  while ((e_di + 1) * (e_di + 2) / 2 <= lane) ++e_di;
  const int e_dn = lane - e_di * (e_di + 1) / 2;
  int cn = n, ci = L.N1, cj = L.N2;
  // This lane's two words of the cone with its origin at layer `on`, row `oi`, column `oc` of the
  // block: where the first one is (bytes from bp, below 4 GB for every table the engine accepts;
  // the second is 256 bytes before it), whether the lane has a cell of the cone at all, and
  // whether it has the second word.
  auto cone_word = [&](const BlockRef &bd, const int n_lo, const int j0, const int on, const int oi, const int oc,
                       bool &has, bool &has_two) -> uint32_t {
    const int ln = on - e_dn, li = oi - e_di, lg = oc - e_dn;
    has = lane < MINRMS_CONE_CELLS && ln > n_lo && li >= ln && lg >= 0;
    has_two = has && lg >= 16;
    const uint32_t word = (uint32_t)bd.bp_off + (__umul24((uint32_t)(li - j0), (uint32_t)bd.W) + (uint32_t)(lg >> 4)) * 64u +
                          (uint32_t)(ln - n_lo);
    return has ? 4u * word : 0u;
  };
  // While the walk crosses a cone, the cone it will reach if it goes on with nothing but matches
  // (MINRMS_CONE_ROWS + 1 of them, down the diagonal) is already on its way: the chains that take
  // longest, those of the largest n, have few other moves, and a fetch takes about the time of
  // four row moves.  Those words travel straight into LDS (no register waits for them, and the
  // compiler, which does not know of the loads, waits for none of them); if the walk arrives where
  // they were fetched for, it picks them up there.
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  const uint32_t ahead_at = lpairs_at + 4u * (uint32_t)L.n_max;
  volatile lds_u32 *ahead = (volatile lds_u32 *)((lds_byte *)tfw_lds + (ahead_at - (uint32_t)(uintptr_t)(lds_byte *)tfw_lds));   // [2][64]
  int ahead_n = -1, ahead_i = -1, ahead_j = -1;             // where that cone has its origin; -1: none
  bool ahead_has = false, ahead_two = false;                // which of its words this lane asked for
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
    if (cn == ahead_n && ci == ahead_i && cj == ahead_j) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the words fetched ahead have landed
#ifdef MINRMS_COOP_STAMPS
      ++ts_hits;
#endif
      if (ahead_has) w0 = ahead[lane];
      if (ahead_two) w1 = ahead[64 + lane];
    } else {
      bool has, has_two;
      const uint32_t off = cone_word(bd, n_lo, j0, on, oi, oc, has, has_two);
      if (has) w0 = *(const uint32_t *)((const unsigned char *)bp + off);
      if (has_two) w1 = *(const uint32_t *)((const unsigned char *)bp + off - 256);
    }
    // the words are needed HERE as far as the compiler is concerned: its wait for the loads above must
    // come before the loads below are issued, not after them (a wait counts every load in flight)
    asm volatile("" : "+v"(w0), "+v"(w1));
    ahead_n = on - (MINRMS_CONE_ROWS + 1); ahead_i = oi - (MINRMS_CONE_ROWS + 1); ahead_j = cj - (MINRMS_CONE_ROWS + 1);
    if (ahead_n > n_lo) {                                   // (the same block: the same bd)
      const uint32_t off = cone_word(bd, n_lo, j0, ahead_n, ahead_i, ahead_j - j0, ahead_has, ahead_two);
      if (ahead_has)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 offset:0"
                     :: "v"(off), "s"(bp), "s"(ahead_at) : "memory", "m0");
      if (ahead_two)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 offset:0"
                     :: "v"(off - 256u), "s"(bp), "s"(ahead_at + 256u) : "memory", "m0");
    } else {
      ahead_n = -1;
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
    const unsigned long long ts_b = MINRMS_TSTAMP();       // (the wait for the words is part of the walk's time: a cone may be in flight)
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
  // the pairs are in LDS; and whatever cone is still on its way has landed: no load may write to
  // LDS after the wave has ended and the workgroup has given its LDS back
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
  if (lane == 0 && slot == 0) {
    const unsigned long long took = MINRMS_TSTAMP() - ts_begin;
    atomicMax(&g_trace_slowest[0], took << 32 | (unsigned long long)n);
    atomicMax(&g_trace_slowest[1], ts_fetches << 32 | (unsigned long long)n);
    unsigned int *rec6 = g_trace_all + 6 * n;
    rec6[0] = (unsigned int)took; rec6[1] = (unsigned int)(ts_c - ts_begin); rec6[2] = (unsigned int)(ts_fetches | ts_hits << 16);
  }
  if (lane == 0 && slot == 0 && k == sp.range_n - 1) {
    const unsigned long long ts_d = MINRMS_TSTAMP();
    g_trace_stamps[0] = ts_d - ts_begin; g_trace_stamps[1] = ts_fetch; g_trace_stamps[2] = ts_walk;
    g_trace_stamps[3] = ts_d - ts_c; g_trace_stamps[4] = ts_fetches | ts_hits << 32; g_trace_stamps[5] = ts_steps;
    g_trace_stamps[6] = ts_c - ts_begin;
  }
#endif
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

// Running per-N minimum (search_or_dyn3d.cc:1825-1835): strict "<" in o_id
// order == lexicographic minimum of (ssd, o_id) == unsigned min of
// (float bits << 32 | o_id) because ssd >= +0.
// Two kernels, both with the match count n across the lanes, so that a wave
// reads a slot's costs and sums as whole rows: (1) the first n at which the
// n-loop leaves each slot -- the smallest n whose test fails; what lies behind
// it (left over from earlier batches when the sweep stops evaluating an
// orientation early) cannot lower that minimum; (2) per n, the minimum key over
// a chunk of slots, then one atomic per chunk and n.  (A thread per slot that
// walked its n one dependent load at a time took 0.23 ms for 21 609 slots.)
#define MINRMS_REDUCE_CHUNK 128
__global__ void __launch_bounds__(256)
k_first_break_rows(const SearchParams sp, const int64_t n_slots, const float *__restrict__ cost_all,
                   const float *__restrict__ ssd_all, const int ssd_stride, int32_t *__restrict__ stop)
{
  const int64_t slot = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (slot >= n_slots) return;
  const float *cost = cost_all + slot * sp.n_max;
  const float *ssd = ssd_all + slot * ssd_stride;
  const float limit = sp.max_rmsd * sp.max_rmsd;
  int first = sp.n_max + 1;
  for (int n0 = 2; n0 <= sp.n_max; n0 += 64) {
    const int n = n0 + lane;
    bool brk = false;
    if (n <= sp.n_max) {
      brk = cost[n - 1] >= MINRMS_UPPER_BOUND;
      if (!brk && n >= sp.n_min && sp.max_rmsd != -1.0f) brk = ssd[n - sp.n_min] > (float)(n * sp.nbb) * limit;
    }
    const unsigned long long any = __ballot(brk);
    if (any) { first = n0 + __ffsll((long long)any) - 1; break; }
  }
  if (lane == 0) stop[slot] = first;
}

__global__ void __launch_bounds__(64)
k_reduce(const SearchParams sp, const int64_t n_slots, const int64_t o_base,
         const uint32_t *__restrict__ oid_of_slot, const int32_t *__restrict__ stop,
         const float *__restrict__ ssd_all, const int ssd_stride,
         unsigned long long *__restrict__ keys)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  const int n = sp.n_min + k;
  if (k >= sp.range_n || n < 2) return;
  const int64_t s0 = (int64_t)blockIdx.y * MINRMS_REDUCE_CHUNK, s1 = min(s0 + MINRMS_REDUCE_CHUNK, n_slots);
  unsigned long long key = ~0ull;
#pragma unroll 4
  for (int64_t s = s0; s < s1; ++s) {
    if (n >= stop[s]) continue;
    const unsigned long long oid = oid_of_slot ? (unsigned long long)oid_of_slot[s] : (unsigned long long)(o_base + s);
    const unsigned long long mine = ((unsigned long long)__float_as_uint(ssd_all[s * ssd_stride + k]) << 32) | oid;
    key = mine < key ? mine : key;
  }
  if (key != ~0ull) atomicMin(&keys[k], key);
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

void launch_cij(const SearchParams &sp, int64_t count, hipStream_t st, int64_t c_floats, const float *xyzA,
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

// Traceback + refit of every (slot, n): one wave per chain while the batch is
// small enough to be latency-bound, one thread per chain otherwise.
// k_lo/k_cnt (thread-per-chain kernel only): the entries of one layer block;
// stop: slots that have left the n-loop are skipped.
void launch_trace_fit_wave(minrms_engine *e, const DpLayout &L, int64_t count, bool want_xf, const int *live_count)
{
  MINRMS_MARK(e, live_count ? 2 : 1, MINRMS_KC_TRACE_WAVE);     // with a live count: a refinement round
  const size_t lds = L.NB * sizeof(BlockRef) + fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)L.n_max + 512;   // + the cone fetched ahead
  hipLaunchKernelGGL(k_trace_fit_wave, dim3((unsigned)(count * e->sp.range_n)), dim3(64), lds, e->stream, L, e->sp, count,
                     e->d_bp, e->d_cost, e->d_xyzA, e->d_xyzB, e->d_og, e->d_pairs, e->d_ssd,
                     want_xf ? e->d_xf : nullptr, live_count);
}

void launch_first_break(const SearchParams &sp, int64_t count, const float *cost_all, const float *ssd_all, int ssd_stride,
                        int32_t *stop, hipStream_t st)
{
  hipLaunchKernelGGL(k_first_break_rows, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, st, sp, count, cost_all, ssd_all,
                     ssd_stride, stop);
}

// stop: room for one int per slot (overwritten)
void launch_reduce(const SearchParams &sp, int64_t count, int64_t o_base, const uint32_t *oid_of_slot,
                   const float *cost_all, const float *ssd_all, int ssd_stride, unsigned long long *keys,
                   int32_t *stop, hipStream_t st)
{
  launch_first_break(sp, count, cost_all, ssd_all, ssd_stride, stop, st);
  const dim3 grid((unsigned)((sp.range_n + 63) / 64), (unsigned)((count + MINRMS_REDUCE_CHUNK - 1) / MINRMS_REDUCE_CHUNK));
  hipLaunchKernelGGL(k_reduce, grid, dim3(64), 0, st, sp, count, o_base, oid_of_slot, stop, ssd_all, ssd_stride, keys);
}

int launch_trace_fit(minrms_engine *e, const DpLayout &L, int64_t count, bool want_xf, bool want_pairs,
                     int k_lo, int k_cnt, const int32_t *stop)
{
  const SearchParams &sp = e->sp;
  if (k_cnt < 0) k_cnt = sp.range_n;
  const int64_t chains = count * k_cnt;
  float *xf = want_xf ? e->d_xf : nullptr;
  if (!stop && chains <= e->wave_trace_max_chains) {
    launch_trace_fit_wave(e, L, count, want_xf, nullptr);
  } else {
    // The walk waits for one dependent load per row of the path and the refit for
    // its LDS reads, so the kernel lives on occupancy (with a quarter of the waves it
    // takes 3.7 times as long).  The compiler is held to 64 registers, 8 waves per
    // SIMD, although that parks up to 25 values in scratch memory (250x250, strings
    // in memory: 18.6 ms at 6 waves per SIMD, 16.7 at 7, 15.3 at 8).  A workgroup
    // is four waves that share one copy of both coordinate sets.  The path strings
    // (2 bits per move, N1+N2 moves at most) stay in LDS as long as five
    // workgroups (20 waves) still fit a CU; longer ones go to memory -- into the
    // buffer of the alignments, which a launch that does not write alignments
    // leaves unused -- where the refit reads them more slowly, but all 32 waves
    // run (k_trace_fit of a whole sweep at 6 waves per SIMD, strings in LDS / in
    // memory: 200x200 6.4 / 7.8 ms with 20 waves in LDS, 250x250 24.3 / 18.5 ms
    // with 16, 300x300 49.5 / 38.2 ms).  The coordinates stay in LDS as long as a
    // CU then still holds 16 waves (250x250: 27 instead of 36 ms per sweep with
    // the coordinates read from memory).  A launch that writes alignments has
    // nowhere else to put the strings: it takes workgroups of one wave when those
    // hold more waves on a CU (long strings: the finer grain).
    MINRMS_MARK(e, 1, MINRMS_KC_TRACE);
    constexpr size_t CU_LDS = 160 * 1024, MAX_LDS = 64 * 1024;
    const size_t coord_bytes = 12 * (size_t)(sp.N1 + sp.N2) * sp.nbb;
    const int path_words = (sp.N1 + sp.N2 + 16) / 16;        // 16 moves per word
    const size_t refs_bytes = L.NB * sizeof(BlockRef);      // (16 bytes each: what follows stays 16-byte aligned)
    struct Shape { int tb; bool path_mem; int in_lds; size_t lds; int waves; };
    auto shape_of = [&](int tb, bool path_mem) {
      const size_t base = refs_bytes + (path_mem ? 0 : (size_t)path_words * 4 * tb);
      auto waves_of = [&](size_t bytes) { return bytes > MAX_LDS ? 0 : (int)std::min<size_t>(4 * MINRMS_TRACE_WAVES, CU_LDS / bytes * (tb / 64)); };
      const int in = waves_of(base + coord_bytes) >= 16;
      const size_t bytes = base + (in ? coord_bytes : 0);
      return Shape{tb, path_mem, in, bytes, waves_of(bytes)};
    };
    const char *where = getenv("MINRMS_TRACE_PATH");        // test hook: "lds" / "memory"
    const bool mem_ok = !want_pairs && (chains + 255) / 256 * 256 * path_words * 4 <=
                                           (int64_t)sizeof(short2) * sp.range_n * L.n_max * e->batch_cap;
    Shape sh = shape_of(256, false);
    const bool good = sh.in_lds && sh.waves >= 20;
    if (mem_ok && (where ? !strcmp(where, "memory") : !good)) {
      sh = shape_of(256, true);
    } else if (!good) {
      const Shape narrow = shape_of(64, false);
      if (narrow.waves > sh.waves) sh = narrow;
    }
    if (sh.waves == 0) return fail(MINRMS_E_ARG, "the path strings of %d x %d residues do not fit LDS", sp.N1, sp.N2);
    const int in_lds = sh.in_lds;
    const size_t lds = sh.lds;
    const int64_t blocks = (chains + sh.tb - 1) / sh.tb;
    uint32_t *path_global = sh.path_mem ? (uint32_t *)e->d_pairs : nullptr;
#define MINRMS_LAUNCH_TRACE(TBV, LDSC, ONE, PL)                                                    \
    hipLaunchKernelGGL((k_trace_fit<TBV, LDSC, ONE, PL>), dim3((unsigned)blocks), dim3(TBV), lds, e->stream, L, sp, \
                       count, e->d_bp, e->d_cost, e->d_xyzA, e->d_xyzB, e->d_og,                  \
                       want_pairs ? e->d_pairs : nullptr, e->d_ssd, xf, path_words, in_lds, path_global, \
                       k_lo, k_cnt, stop)
#define MINRMS_LAUNCH_TRACE2(LDSC, ONE) do {                                                       \
      if (path_global) MINRMS_LAUNCH_TRACE(256, LDSC, ONE, false);                                 \
      else if (sh.tb == 256) MINRMS_LAUNCH_TRACE(256, LDSC, ONE, true);                            \
      else MINRMS_LAUNCH_TRACE(64, LDSC, ONE, true); } while (0)
    if (in_lds) { if (sp.nbb == 1) MINRMS_LAUNCH_TRACE2(true, true); else MINRMS_LAUNCH_TRACE2(true, false); }
    else        { if (sp.nbb == 1) MINRMS_LAUNCH_TRACE2(false, true); else MINRMS_LAUNCH_TRACE2(false, false); }
#undef MINRMS_LAUNCH_TRACE2
#undef MINRMS_LAUNCH_TRACE
  }
  return 0;
}

// Distance tables, DP, traceback + refit for the first `count` slots of d_og.
// slot_clip: device array with the distance cutoff of every slot, or NULL for
// the cutoff currently in force.
// og_table/og_rows: if given, slot s is evaluated at transform og_table[og_rows[s]]
// instead of d_og[s] (only valid with refine_while_searching, where nothing
// else reads the transform a slot was evaluated at).
int run_batch(minrms_engine *e, int64_t count, bool want_xf, const float *slot_clip, const float *og_table,
              const int32_t *og_rows, bool want_pairs, bool cost_only)
{
  const bool coop = e->coop_pb != 0 && count <= e->coop_max_batch;
  const DpLayout &L = coop ? e->Lc : e->L;
  const SearchParams &sp = e->sp;
  hipStream_t st = e->stream;
  MINRMS_MARK(e, 1, coop ? MINRMS_KC_COOP_CIJ : MINRMS_KC_CIJ);
  launch_cij(sp, count, st, L.c_floats, e->d_xyzA, e->d_xyzB, og_table ? og_table : e->d_og, e->d_C, e->clip_now,
             slot_clip, og_rows, nullptr);
  if (coop) {
    int rc = launch_coop(e, count);
    if (rc) return rc;
    if (!cost_only) { rc = launch_trace_fit(e, L, count, want_xf, want_pairs); if (rc) return rc; }
    MINRMS_MARK(e, 1, -1);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (e->timing)
    for (int b = 0; b < L.NB; ++b) e->blk_orients[b] += count;
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
      MINRMS_MARK(e, 1, MINRMS_KC_DP_BLOCK0 + b);
      hipLaunchKernelGGL(fn, dim3((unsigned)count), dim3(64), 0, st, L, e->d_C, e->d_bp, e->d_planes,
                         e->d_edges, e->d_cost, b, p, stop);
    }
    if (stepwise) {
      MINRMS_MARK(e, 1, MINRMS_KC_OTHER);         // after the block's last panel: what follows is not DP time
      const int n_lo = MINRMS_LAYERS_PER_BLOCK * b + 1, n_hi = std::min(sp.n_max, MINRMS_LAYERS_PER_BLOCK * (b + 1));
      const int k_lo = std::max(n_lo, sp.n_min) - sp.n_min, k_hi = n_hi - sp.n_min;
      if (!cost_only && k_hi >= k_lo) {
        const int rc = launch_trace_fit(e, L, count, want_xf, want_pairs, k_lo, k_hi - k_lo + 1, stop);
        if (rc) return rc;
      }
      if (b + 1 < L.NB)
        hipLaunchKernelGGL(k_update_stop, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, sp, count, n_lo, n_hi,
                           e->d_cost, cost_only ? e->d_cost + (sp.n_min - 1) : e->d_ssd,
                           cost_only ? sp.n_max : sp.range_n, e->d_stop);
    }
  }
  if (!stepwise && !cost_only) {
    const int rc = launch_trace_fit(e, L, count, want_xf, want_pairs);
    if (rc) return rc;
  }
  MINRMS_MARK(e, 1, -1);
  HIP_TRY(hipGetLastError());
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
      const bool cost_only = !e->sp.rws && !alts && !e->count_updates;
      int rc = run_batch(e, count, alts && !e->alt_by_pairs, e->d_clip, nullptr, nullptr, false, cost_only);
      if (rc) return rc;
      if (e->count_updates) { rc = update_counts_batch(e, count, 0, e->d_oids); if (rc) return rc; }
      MINRMS_MARK(e, 1, MINRMS_KC_REDUCE);
      launch_reduce(e->sp, count, 0, e->d_oids, e->d_cost, cost_only ? e->d_cost + (e->sp.n_min - 1) : e->d_ssd,
                    cost_only ? e->sp.n_max : e->sp.range_n, e->d_keys, e->d_stop, e->stream);
      MINRMS_MARK(e, 1, -1);
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
  if (e->count_updates) { int rc = update_counts_begin(e, start, stop - start); if (rc) return rc; }
  if (e->of) {
    int rc = search_filtered(e, og, start, stop);
    if (rc) return rc;
  } else {
    const bool alts = e->num_solutions > 1 && (!e->alt_by_pairs || e->cfg.alt_rmsd_tolerance != -1.0f);
    // Every launch of a batch ends in a tail: its last waves run on SIMDs that
    // are emptying, and the next launch of the chain (same stream) cannot start.
    // A sweep of many orientations is therefore cut into an even number of
    // batches that alternate between the two halves of the workspace and between
    // two streams: the launches of one batch fill the tails of the other's.
    // (Not while alternates are collected: that reads every batch back.)
    int halves = 1;
    if (!alts && !e->count_updates && e->stream_b && stop - start >= 8192 && e->batch_cap >= 2) {   // (update counts need the batches in o_id order)
      halves = 2;
      if (const char *s = getenv("MINRMS_SWEEP_STREAMS")) halves = atoi(s) >= 2 ? 2 : 1;   // 1: the whole workspace, one stream
    }
    int64_t region_cap = e->batch_cap / halves;
    // batches of equal size: a short last batch would leave most of the GPU idle
    int64_t n_batches = (stop - start + region_cap - 1) / region_cap;
    if (halves == 2) {
      int pieces = 2;
      if (const char *s = getenv("MINRMS_SWEEP_PIECES")) pieces = std::max(2, atoi(s));    // tuning hook
      n_batches = std::max<int64_t>(n_batches, pieces);
      n_batches += n_batches & 1;
    }
    int64_t per_batch = n_batches ? (stop - start + n_batches - 1) / n_batches : 0;
    if (halves == 2 && e->coop_pb != 0 && per_batch - n_batches <= e->coop_max_batch) {
      // batches this small (the last one is the smallest, by less than the number of batches) go
      // through the cooperative kernel, whose row counters and granules are per engine, not per
      // stream: one stream, the whole workspace
      halves = 1;
      n_batches = (stop - start + e->batch_cap - 1) / e->batch_cap;
      per_batch = n_batches ? (stop - start + n_batches - 1) / n_batches : 0;
    }
    // the engine's buffers as the launchers see them; a batch in the second half works on a view of them
    struct View { float *og, *C, *planes, *edges, *cost, *ssd, *xf; uint32_t *bp; short2 *pairs; int32_t *stop; hipStream_t stream; int64_t cap; };
    const View whole = {e->d_og, e->d_C, e->d_planes, e->d_edges, e->d_cost, e->d_ssd, e->d_xf, e->d_bp, e->d_pairs, e->d_stop, e->stream, e->batch_cap};
    auto set_view = [&](const View &v) {
      e->d_og = v.og; e->d_C = v.C; e->d_planes = v.planes; e->d_edges = v.edges; e->d_cost = v.cost; e->d_ssd = v.ssd;
      e->d_xf = v.xf; e->d_bp = v.bp; e->d_pairs = v.pairs; e->d_stop = v.stop; e->stream = v.stream; e->batch_cap = v.cap;
    };
    if (halves == 1) region_cap = e->batch_cap;
    View second = whole;
    if (halves == 2) {
      const int64_t at = region_cap, rn = e->sp.range_n;
      const int64_t bp_words = std::max(e->L.bp_words, e->coop_pb ? e->Lc.bp_words : 0);
      const int64_t plane_floats = std::max(e->L.plane_floats, e->coop_pb ? e->Lc.plane_floats : 0);
      second.og += 12 * at; second.C += e->L.c_floats * at; second.planes += plane_floats * at;
      second.edges += e->L.edge_floats * at; second.cost += (int64_t)e->L.n_max * at; second.ssd += rn * at;
      second.xf += 12 * rn * at; second.bp += bp_words * at; second.pairs += rn * e->L.n_max * at; second.stop += at;
      second.stream = e->stream_b;
      second.cap = e->batch_cap - region_cap;
    }
    View first = whole;
    first.cap = region_cap;
    int rc_loop = 0;
    int64_t batch_no = 0;
    for (int64_t off = start; off < stop && rc_loop == 0; off += per_batch, ++batch_no) {
      const int64_t count = std::min(per_batch, stop - off);
      set_view((halves == 2 && (batch_no & 1)) ? second : first);
      if (hipMemcpyAsync(e->d_og, og + 12 * off, 48 * count, hipMemcpyHostToDevice, e->stream) != hipSuccess) {
        rc_loop = fail(MINRMS_E_DEVICE, "copy of the orientations failed");
        break;
      }
      // Without refinement while searching the sum of squared distances of an
      // orientation is the DP cost itself (search_or_dyn3d.cc:1783-1790 not
      // taken): the running minimum needs neither traceback nor back-pointers.
      const bool cost_only = !e->sp.rws && !alts && !e->count_updates;
      rc_loop = run_batch(e, count, alts && !e->alt_by_pairs, nullptr, nullptr, nullptr, false, cost_only);
      if (rc_loop) break;
      if (e->count_updates) { rc_loop = update_counts_batch(e, count, off, nullptr); if (rc_loop) break; }
      MINRMS_MARK(e, 1, MINRMS_KC_REDUCE);
      launch_reduce(e->sp, count, off, nullptr, e->d_cost, cost_only ? e->d_cost + (e->sp.n_min - 1) : e->d_ssd,
                    cost_only ? e->sp.n_max : e->sp.range_n, e->d_keys, e->d_stop, e->stream);
      MINRMS_MARK(e, 1, -1);
      if (hipGetLastError() != hipSuccess) { rc_loop = fail(MINRMS_E_DEVICE, "a launch of the sweep failed"); break; }
      if (alts) rc_loop = alt_collect_batch(e, count, off, nullptr);
    }
    set_view(whole);
    if (halves == 2) HIP_TRY(hipStreamSynchronize(e->stream_b));
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (rc_loop) return rc_loop;
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
  if (getenv("MINRMS_TRACE_ALL")) {
    static unsigned int all[2080 * 6];
    HIP_TRY(hipMemcpyFromSymbol(all, HIP_SYMBOL(g_trace_all), sizeof all));
    for (int n = 2; n <= e->cfg.n_max && n < 2080; ++n)
      fprintf(stderr, "chain n=%d: %u cycles, %u before the refit, %u fetches (%u ahead)\n", n, all[6 * n], all[6 * n + 1],
              all[6 * n + 2] & 0xffff, all[6 * n + 2] >> 16);
  }
  {
    unsigned long long slow[2], zero[2] = {0, 0};
    HIP_TRY(hipMemcpyFromSymbol(slow, HIP_SYMBOL(g_trace_slowest), 16));
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_slowest), zero, 16));
    fprintf(stderr, "tracer: the slowest chain of slot 0 since the last call: n = %llu, %llu cycles; most cone fetches: n = %llu, %llu\n",
            slow[0] & 0xffffffffull, slow[0] >> 32, slow[1] & 0xffffffffull, slow[1] >> 32);
  }
  unsigned long long f[4];
  HIP_TRY(hipMemcpyFromSymbol(f, HIP_SYMBOL(fit::g_fit_stamps), 32));
  out8[7] = f[2];                      // the eigen-solve and what follows it
  refine_debug_print_round_stamps();
  out8[3] = f[0] << 32 | (f[1] & 0xffffffffull);   // pass 1, pass 2 of the superposition
  return 0;
}
#endif

hipError_t touch_unit_search()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)k_fill_i32);
}
