// engine_internal.h -- what the translation units of libminrms_hip.so share:
// the engine object, the search parameters handed to the kernels, error
// plumbing, the pool of device memory, and the host functions one unit calls in
// another.  Every kernel is launched from the unit that defines it.
#pragma once
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
#include "nw_filter_params.h"
#include "alt_solutions.h"

// ---------------------------------------------------------------------------
// error plumbing (engine_core.hip)
// ---------------------------------------------------------------------------
int fail(int code, const char *fmt, ...);

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(MINRMS_E_DEVICE, "%s failed: %s (%s:%d)", #expr,                 \
                  hipGetErrorString(e_), __FILE__, __LINE__);                      \
  } while (0)

// ---------------------------------------------------------------------------
// device memory of destroyed engines, kept for the next engine (engine_core.hip)
// ---------------------------------------------------------------------------
namespace pool {
struct HostSlot { int device; void *ptr; hipEvent_t ev[2]; };   // the events belong to a device
extern std::mutex mu;
extern std::vector<HostSlot> host_slots;   // pinned host blocks with their two events (the refinement's control snapshots)
size_t parked_bytes(int device);
// device must be current
hipError_t take(int device, const char *tag, size_t bytes, void **out, size_t *got, size_t *clean);
void park(int device, const char *tag, void *ptr, size_t bytes, size_t clean);
void trim();
}  // namespace pool

// ---------------------------------------------------------------------------
// what the kernels are told about the search
// ---------------------------------------------------------------------------
struct SearchParams {
  int32_t N1, N2, nbb, n_min, n_max, range_n;
  float max_rmsd;          // -1 = off
  float clip;              // DistanceMetric::max_allowed_dist_metric as the constructor sets it
  int32_t rws;             // refine_while_searching
};


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


struct AltRecord;        // alternates.hip
// control block of the refinement rounds (refine.hip), one per engine in device memory
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

typedef void (*dp_kernel_fn)(const DpLayout, const float *, uint32_t *, float *, float *, float *,
                             const int, const int, const int32_t *);
// dp_sweep_kernels.hip: the instantiation of dp::k_dp_panel for one panel width, or NULL
dp_kernel_fn pick_dp_kernel(int PB, bool inject, bool has_next, bool bp = true);

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// kernel classes of the timing facility (include/minrms_engine.h: MINRMS_KC_*); layer block b of the sweep is MINRMS_KC_DP_BLOCK0 + b
#define MINRMS_KC_COUNT (MINRMS_KC_DP_BLOCK0 + MINRMS_MAX_BLOCKS)

// engines alive in this process (a batch of pairs runs several at a time, one host thread each)
extern std::atomic<int> live_engines;

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
  int *d_gprog = nullptr;           // [coop_max_batch][coop_groups.total_tasks] row counters (several groups only)
  size_t gprog_bytes = 0, xedge_bytes = 0;   // what was asked for d_gprog / d_xedge (cleared as a whole when the tags wrap)
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
  hipStream_t stream_b = nullptr;     // the sweep's second stream (batches alternate between two halves of the workspace)
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
  // update counts (minrms_engine_count_updates)
  bool count_updates = false;
  int32_t *d_upd_counts = nullptr;                       // [upd_cap] per orientation of the last search
  int64_t upd_cap = 0, upd_first = 0, upd_n = 0;
  float *d_upd_best = nullptr, *d_upd_chunk = nullptr;   // running best; chunk minima / best before each chunk
  float *d_alt_best = nullptr;                           // [range_n] running best handed to the filter
  float *d_alt_chunk = nullptr;                          // [chunks of slots][range_n] chunk minima, then the best before each chunk
  int32_t *d_alt_offsets = nullptr;                      // [range_n][chunks] survivors per (n, chunk), then where they go
  int *d_alt_count = nullptr;
  // alternates by matched pairs (`-alt-r`): one more sweep per rank
  bool alt_by_pairs = false;
  std::vector<unsigned char> alt_seen;                   // per o_id: 0 not searched, 1 failed the filter, 2 evaluated
  short2 *d_rank_pairs = nullptr;                        // [num_solutions-1][range_n][n_max]
  unsigned long long *d_rank_keys = nullptr;             // [range_n] keys of the rank being searched
  // ... with the search split over several engines: the caller's collectives (minrms_engine_set_collectives)
  minrms_allreduce_min_fn comm_allreduce_min = nullptr;
  minrms_allgather_i64_fn comm_allgather = nullptr;
  void *comm_user = nullptr;
  alt::RankTables ranks;                                 // all ranks on the host (rank 0 mirrors the device)
  bool ranks_current = false;
  std::vector<float> h_og;          // host copy of the orientation list seen by search()
  bool tables_current = false;      // result tables reflect the keys
  bool searched = false;
  // Timing of the kernels by class (MINRMS_KC_*): an event is recorded on the
  // engine's stream in front of a launch, and the time up to the next event
  // belongs to the class named with it.  Level 1: the kernels of the sweep and of
  // the winners' batch; level 2: also every kernel of the refinement rounds (an
  // event between the kernels of a chain that is bound by latency lengthens it:
  // not for timed runs).  Read back lazily, after the stream has been synchronised.
  int timing = 0;
  std::vector<hipEvent_t> ev_pool;          // all events ever created
  std::vector<int> ev_class;                // class of what runs between event q and event q + 1, -1: nothing timed
  size_t ev_used = 0;                       // events handed out since the last drain
  double kc_ms[MINRMS_KC_COUNT] = {0};
  int64_t kc_launches[MINRMS_KC_COUNT] = {0};
  int64_t blk_orients[MINRMS_MAX_BLOCKS] = {0};
};

// engine_core.hip
int build_layout(const minrms_cfg &c, int max_pb, bool uniform_pb, DpLayout *out, int min_panels = 1);
int configure(minrms_engine *e, minrms_cfg c);
float clip_after(const minrms_engine *e, int64_t calls);
int check_device_error(minrms_engine *e);
// timing: what is enqueued from here to the next mark belongs to kernel class `cls` (-1: not timed)
void time_mark(minrms_engine *e, int cls);
#define MINRMS_MARK(E, LEVEL, CLS) do { if ((E)->timing >= (LEVEL)) time_mark((E), (CLS)); } while (0)
// Loads, on the current device, the code object of every unit (the runtime would
// otherwise load each when a kernel of it is first launched).
int load_code_objects();
hipError_t touch_unit_search();        // each: hipFuncGetAttributes of one kernel of the unit
hipError_t touch_unit_refine();
hipError_t touch_unit_alternates();
hipError_t touch_unit_orientations();
hipError_t touch_unit_dp_sweep();
hipError_t touch_unit_dp_coop();
// dp_coop_kernels.hip
int allow_large_lds();
int launch_coop(minrms_engine *e, int64_t count, const int *live_count = nullptr);
// search.hip
void launch_cij(const SearchParams &sp, int64_t count, hipStream_t st, int64_t c_floats, const float *xyzA,
                const float *xyzB, const float *og, float *C, float uniform_clip, const float *clip_of_slot,
                const int32_t *og_row_of_slot, const int *live_count);
int launch_trace_fit(minrms_engine *e, const DpLayout &L, int64_t count, bool want_xf, bool want_pairs,
                     int k_lo = 0, int k_cnt = -1, const int32_t *stop = nullptr);
// wave-per-chain traceback + refit of every (slot, n) of `count` slots (the live ones, if live_count is given)
void launch_trace_fit_wave(minrms_engine *e, const DpLayout &L, int64_t count, bool want_xf, const int *live_count);
int run_batch(minrms_engine *e, int64_t count, bool want_xf, const float *slot_clip = nullptr,
              const float *og_table = nullptr, const int32_t *og_rows = nullptr, bool want_pairs = true,
              bool cost_only = false);
// running per-N minimum of `count` slots into keys (k_reduce)
void launch_first_break(const SearchParams &sp, int64_t count, const float *cost_all, const float *ssd_all, int ssd_stride,
                        int32_t *stop, hipStream_t st);
void launch_reduce(const SearchParams &sp, int64_t count, int64_t o_base, const uint32_t *oid_of_slot,
                   const float *cost_all, const float *ssd_all, int ssd_stride, unsigned long long *keys,
                   int32_t *stop, hipStream_t st);
// refine.hip
// least-squares transforms of the best alignments of entries klist[0..n_list) -> og_out[n_list][12]
void launch_refit_entries(minrms_engine *e, int n_list, const int32_t *klist, float *og_out, const int *live_count);
int refine_best(minrms_engine *e, int64_t max_iters, minrms_refine_stats *stats);
#ifdef MINRMS_COOP_STAMPS
void refine_debug_print_round_stamps();
#endif
// alternates.hip
int alt_collect_batch(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *oids);
int update_counts_begin(minrms_engine *e, int64_t first, int64_t n);
int update_counts_batch(minrms_engine *e, int64_t count, int64_t o_base, const uint32_t *d_oids);
int alt_pull_rank0(minrms_engine *e);
int alt_push_rank0(minrms_engine *e);
int alt_fill_ranks(minrms_engine *e);
int alt_fill_ranks_by_pairs(minrms_engine *e);
int alt_refine_rank(minrms_engine *e, int r, int64_t max_iters, int64_t *iterations, int64_t *updates);
