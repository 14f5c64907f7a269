// refine.hip -- iterative refinement of the per-N best alignments:
//   bin/minrms/search_or_dyn3d.cc:2261-2658  Refine  -> minrms_engine_refine
//   lib/global_utils/simple_numeric_utils.h:350-376  FindMedian -> wave_median_of_set
#include "engine_internal.h"
#include "fit_device.h"

#ifdef MINRMS_COOP_STAMPS
__device__ unsigned long long g_round_stamps[8];    // diagnostic build: cycles of the round kernel's three parts, summed
#define MINRMS_TSTAMP() __builtin_readcyclecounter()
#endif

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
  static_assert(MINRMS_NEED_WORDS <= 64, "one word of the set per lane");
  const int cnt = __popcll(my_word);
  int upto = cnt;                                       // members in words 0..lane: prefix sum by row shifts ...
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x111, 0xf, 0xf, true);   // row_shr:1, lanes without a source add 0
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x112, 0xf, 0xf, true);
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x114, 0xf, 0xf, true);
  upto += __builtin_amdgcn_update_dpp(0, upto, 0x118, 0xf, 0xf, true);
  // ... inside each row of 16 lanes; then the totals of the rows before
  const int t0 = __builtin_amdgcn_readlane(upto, 15), t1 = __builtin_amdgcn_readlane(upto, 31);
  const int t2 = __builtin_amdgcn_readlane(upto, 47), t3 = __builtin_amdgcn_readlane(upto, 63);
  const int row = lane >> 4;
  upto += (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
  const int total = t0 + t1 + t2 + t3;
  if (total == 0) return -1;
  const int half = (total + 1) / 2;
  const unsigned long long reached = __ballot(upto >= half);
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

void launch_refit_entries(minrms_engine *e, int n_list, const int32_t *klist, float *og_out, const int *live_count)
{
  hipLaunchKernelGGL(k_refit_entries, dim3((unsigned)n_list), dim3(64), fit::WAVE_LDS_FLOATS * 4 + 4 * (size_t)e->cfg.n_max,
                     e->stream, e->sp, e->cfg.n_max, n_list, klist, e->d_xyzA, e->d_xyzB, e->d_best_pairs, og_out,
                     live_count);
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
    MINRMS_MARK(e, 2, MINRMS_KC_ROUND);
    hipLaunchKernelGGL(k_refine_round, dim3(1), dim3(256), 0, st, sp, e->cfg.n_max, e->d_slot_of_k, e->d_cost,
                       e->d_ssd, e->d_xf, e->d_pairs, e->d_best_ssd, e->d_best_xf, e->d_best_pairs,
                       e->d_winner, e->d_need, e->d_ctrl, e->d_slot_of_k, e->d_klist);
  };
  round_kernel();                       // plans the first round
  MINRMS_MARK(e, 2, -1);
  const int *live = &e->d_ctrl->count;
  const DpLayout &L = e->Lc;
  // The host looks at the control block after every few rounds.  So that the
  // GPU does not sit idle while it does, the next few rounds are enqueued
  // BEFORE the host waits for the previous ones; when the loop has ended, the
  // rounds enqueued in vain find ctrl.done / a live count of 0 and leave at once.
  auto enqueue_rounds = [&](int rounds, RefineCtrl *snapshot, hipEvent_t done_event) -> int {
    for (int q = 0; q < rounds; ++q) {
      const float *og = e->d_best_xf;
      const int32_t *rows = e->d_klist;
      if (!sp.rws) {
        launch_refit_entries(e, spec_cap, e->d_klist, e->d_og, live);
        og = e->d_og;
        rows = nullptr;
      }
      MINRMS_MARK(e, 2, MINRMS_KC_COOP_CIJ);
      launch_cij(sp, spec_cap, st, L.c_floats, e->d_xyzA, e->d_xyzB, og, e->d_C, e->clip_now, nullptr, rows, live);
      int rc = launch_coop(e, spec_cap, live);
      if (rc) return rc;
      launch_trace_fit_wave(e, L, spec_cap, true, live);
      round_kernel();
    }
    MINRMS_MARK(e, 2, -1);
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

int refine_best(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats)
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
        launch_refit_entries(e, count, e->d_klist, e->d_og, nullptr);
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

#ifdef MINRMS_COOP_STAMPS
// diagnostic build only: what the round kernel spent where (the stream must have been synchronised)
void refine_debug_print_round_stamps()
{
  unsigned long long r[8];
  if (hipMemcpyFromSymbol(r, HIP_SYMBOL(g_round_stamps), 64) != hipSuccess) return;
  fprintf(stderr, "round kernel, %llu launches: start %llu, replay %llu (%llu iterations), plan and write-back %llu cycles per launch; %llu tables planned\n",
          r[3], r[3] ? r[0] / r[3] : 0, r[3] ? r[1] / r[3] : 0, r[4], r[3] ? r[2] / r[3] : 0, r[5]);
}
#endif

hipError_t touch_unit_refine()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)k_refit_entry);
}
