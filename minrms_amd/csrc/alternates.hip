// alternates.hip -- 2nd, 3rd, ... best alignments per N:
//   bin/minrms/search_or_dyn3d.cc:421-1640   by orientation (-alt-o, -alt-o-orig)
//   bin/minrms/search_or_dyn3d.cc:1969-2250  by matched pairs (-alt-r, MultiPass)
// The selection itself is host code (alt_solutions.h).
#include "engine_internal.h"

// Alternate solutions by orientation keep, per n, every (orientation, n) record
// whose sum is within a tolerance of the best so far in o_id order
// (KeepTrackOfAltSolutions, search_or_dyn3d.cc:421-470).  Nearly all records of
// a sweep fail that test, so it is applied here first, with a little slack, and
// only the survivors travel to the host, which applies the exact rule to them
// (a record that lowers the running best always survives, so the host sees the
// same sequence of bests).
struct AltRecord { int32_t slot, k; float ssd; int32_t pad; float xf[12]; };

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
k_alt_chunk_prefix(const SearchParams sp, const int n_chunks, const float *best_in,
                   float *__restrict__ chunk_min, float *best_out)
{
  // best_out (may be best_in, may be NULL): the best after the last chunk
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= sp.range_n) return;
  float best = best_in[k];
  for (int c = 0; c < n_chunks; ++c) {
    const float m = chunk_min[(int64_t)c * sp.range_n + k];
    chunk_min[(int64_t)c * sp.range_n + k] = best;
    best = fminf(best, m);
  }
  if (best_out) best_out[k] = best;
}

// How many entries of the per-N table every orientation of a batch improves when
// it is evaluated, in o_id order (count_num_updates of FirstPass,
// search_or_dyn3d.cc:1712,1825-1834): one thread per (chunk, n) walks its chunk
// with the running best; counts[o_id - first] is raised for every strict improvement.
__global__ void __launch_bounds__(64)
k_count_updates(const SearchParams sp, const int64_t n_slots, const float *__restrict__ ssd_all,
                const int32_t *__restrict__ stop, const float *__restrict__ best_before,
                const uint32_t *__restrict__ oid_of_slot, const int64_t o_base, const int64_t first,
                int32_t *__restrict__ counts)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= sp.range_n) return;
  const int n = sp.n_min + k;
  if (n < 2) return;
  const int64_t s0 = (int64_t)blockIdx.y * MINRMS_ALT_CHUNK, s1 = min(s0 + MINRMS_ALT_CHUNK, n_slots);
  float best = best_before[(int64_t)blockIdx.y * sp.range_n + k];
  for (int64_t s = s0; s < s1; ++s) {
    if (n >= stop[s]) continue;
    const float v = ssd_all[s * sp.range_n + k];
    if (v < best) {
      best = v;
      atomicAdd(&counts[(oid_of_slot ? (int64_t)oid_of_slot[s] : o_base + s) - first], 1);
    }
  }
}

// best[k] = the sum of squared distances in key k (1e26 where there is none)
__global__ void __launch_bounds__(64)
k_best_from_keys(const int rn, const unsigned long long *__restrict__ keys, float *__restrict__ best)
{
  const int k = blockIdx.x * 64 + threadIdx.x;
  if (k >= rn) return;
  const unsigned long long key = keys[k];
  best[k] = key == ~0ull ? MINRMS_UPPER_BOUND : __uint_as_float((uint32_t)(key >> 32));
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

// KeepTrackOfAltSolutions (search_or_dyn3d.cc:477-542) for the batch that has
// just been evaluated (with transforms): slot s holds orientation
// oids ? oids[s] : o_base + s; slots are in ascending o_id order.
static int alt_collect_batch_on_host(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids);

// Update counts (minrms_engine_count_updates): room for the orientations [first, first + n)
// of a search, the running best taken from the keys the engine holds.
int update_counts_begin(minrms_engine *e, int64_t first, int64_t n)
{
  const int rn = e->sp.range_n;
  const int max_chunks = (int)((e->batch_cap + MINRMS_ALT_CHUNK - 1) / MINRMS_ALT_CHUNK);
  if (!e->d_upd_best) {
    HIP_TRY(hipMalloc((void **)&e->d_upd_best, 4 * (size_t)rn));
    HIP_TRY(hipMalloc((void **)&e->d_upd_chunk, 4 * (size_t)rn * max_chunks));
  }
  if (n > e->upd_cap) {
    if (e->d_upd_counts) HIP_TRY(hipFree(e->d_upd_counts));
    e->d_upd_counts = nullptr;
    e->upd_cap = 0;
    HIP_TRY(hipMalloc((void **)&e->d_upd_counts, 4 * (size_t)std::max<int64_t>(n, 1)));
    e->upd_cap = n;
  }
  e->upd_first = first;
  e->upd_n = n;
  HIP_TRY(hipMemsetAsync(e->d_upd_counts, 0, 4 * (size_t)std::max<int64_t>(n, 1), e->stream));
  hipLaunchKernelGGL(k_best_from_keys, dim3((unsigned)((rn + 63) / 64)), dim3(64), 0, e->stream, rn, e->d_keys, e->d_upd_best);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ... and the counts of the batch that has just been evaluated (slots in ascending o_id order:
// slot s holds orientation d_oids[s], or o_base + s)
int update_counts_batch(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *d_oids)
{
  const SearchParams &sp = e->sp;
  const int rn = sp.range_n;
  const int n_chunks = (int)((count + MINRMS_ALT_CHUNK - 1) / MINRMS_ALT_CHUNK);
  hipStream_t st = e->stream;
  launch_first_break(sp, count, e->d_cost, e->d_ssd, rn, e->d_stop, st);
  const dim3 chunk_grid((unsigned)((rn + 63) / 64), (unsigned)n_chunks);
  hipLaunchKernelGGL(k_alt_chunk_min, chunk_grid, dim3(64), 0, st, sp, count, e->d_ssd, e->d_stop, e->d_upd_chunk);
  hipLaunchKernelGGL(k_alt_chunk_prefix, dim3((unsigned)((rn + 63) / 64)), dim3(64), 0, st, sp, n_chunks, e->d_upd_best,
                     e->d_upd_chunk, e->d_upd_best);
  hipLaunchKernelGGL(k_count_updates, chunk_grid, dim3(64), 0, st, sp, count, e->d_ssd, e->d_stop, e->d_upd_chunk, d_oids,
                     o_base, e->upd_first, e->d_upd_counts);
  HIP_TRY(hipGetLastError());
  return 0;
}

// The records of one batch that may become alternate solutions: filtered on the
// device (MINRMS_ALT_HOST_FILTER=1, a test hook, copies everything instead).
int alt_collect_batch(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids)
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
  launch_first_break(sp, count, e->d_cost, e->d_ssd, rn, e->d_stop, st);
  const dim3 chunk_grid((unsigned)((rn + 63) / 64), (unsigned)n_chunks);
  hipLaunchKernelGGL(k_alt_chunk_min, chunk_grid, dim3(64), 0, st, sp, count, e->d_ssd, e->d_stop, e->d_alt_chunk);
  hipLaunchKernelGGL(k_alt_chunk_prefix, dim3((unsigned)((rn + 63) / 64)), dim3(64), 0, st, sp, n_chunks, e->d_alt_best,
                     e->d_alt_chunk, (float *)nullptr);
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
  if (e->num_solutions <= 1) return fail(MINRMS_E_ARG, "candidate lists exist with alternate solutions only");
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
int alt_pull_rank0(minrms_engine *e)
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
int alt_push_rank0(minrms_engine *e)
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
int alt_fill_ranks(minrms_engine *e)
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
int alt_fill_ranks_by_pairs(minrms_engine *e)
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
  // Search split over several engines: each knows the orientations of its own
  // range (alt_seen) -- the passes revisit the union, every engine its own part.
  // An orientation's place in the union decides the cutoff it is evaluated with
  // (one more filter call per orientation revisited).
  const bool split = e->comm_allgather != nullptr;
  if ((int64_t)e->alt_seen.size() < n_og) e->alt_seen.resize((size_t)n_og, 0);
  if (split) {
    // this engine's contribution: those of its own range (after an exchange of the candidate records every
    // engine already holds the whole list; without an RMSD tolerance only its own part)
    std::vector<int64_t> own;
    for (int64_t o : use)
      if (e->alt_seen[(size_t)o] != 0) own.push_back(o);
    std::vector<int64_t> all((size_t)std::max<int64_t>(n_og, 1));
    int64_t n_all = 0;
    if (e->comm_allgather(e->comm_user, own.data(), (int64_t)own.size(), all.data(), n_og, &n_all) != 0 || n_all < 0 || n_all > n_og)
      return fail(MINRMS_E_STATE, "the caller's all-gather of the orientations to revisit failed");
    all.resize((size_t)n_all);
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (int64_t o : all)
      if (o < 0 || o >= n_og) return fail(MINRMS_E_STATE, "all-gather returned o_id %lld outside the orientation list", (long long)o);
    use.swap(all);
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
    const int64_t calls_before = e->of_calls;
    for (size_t q = 0; q < use.size(); ++q) {
      const int64_t o = use[q];
      if (split && e->alt_seen[(size_t)o] == 0) continue;       // another engine's orientation
      if (e->of && e->alt_seen[(size_t)o] != 2) continue;
      oids.push_back((uint32_t)o);
      clips.push_back(e->of ? clip_after(e, calls_before + (int64_t)q + 1) : e->clip_now);
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
    if (split && e->comm_allreduce_min(e->comm_user, keys.data(), rn) != 0)
      return fail(MINRMS_E_STATE, "the caller's all-reduce of the keys of pass %d failed", r);
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
        // the cutoff the pass evaluated the winner with (on whichever engine): its place among the revisited
        const int64_t q = (int64_t)(std::lower_bound(use.begin(), use.end(), (int64_t)win[base + s]) - use.begin());
        wclip[(size_t)s] = e->of ? clip_after(e, calls_before + q + 1) : e->clip_now;
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

// Refinement of rank r >= 1 (REFINE_ALL, search_or_dyn3d.cc:2344-2640): one
// alignment at a time, and an iteration only ever replaces the alignment it
// started from (m == n, :2596-2613).
int alt_refine_rank(minrms_engine *e, int r, int64_t max_iters, int64_t *iterations, int64_t *updates)
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
      launch_refit_entries(e, 1, e->d_klist, e->d_og, nullptr);
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

hipError_t touch_unit_alternates()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)k_alt_offsets);
}

// ---------------------------------------------------------------------------
// "<count> updates for orientation <o_id + 1>": what FirstPass prints for every
// orientation that improved the table (search_or_dyn3d.cc:1885-1889).
// ---------------------------------------------------------------------------
extern "C" int minrms_engine_count_updates(minrms_engine *e, int32_t enable)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  e->count_updates = enable != 0;
  return MINRMS_OK;
}

extern "C" int minrms_engine_get_update_counts(minrms_engine *e, int32_t *counts, int64_t n)
{
  if (!e || !counts) return fail(MINRMS_E_ARG, "null argument");
  if (!e->searched || !e->count_updates || !e->d_upd_counts) return fail(MINRMS_E_STATE, "update counts were not kept by the last search");
  if (n != e->upd_n) return fail(MINRMS_E_ARG, "the last search covered %lld orientations, not %lld", (long long)e->upd_n, (long long)n);
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (n > 0) HIP_TRY(hipMemcpyAsync(counts, e->d_upd_counts, 4 * (size_t)n, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MINRMS_OK;
}
