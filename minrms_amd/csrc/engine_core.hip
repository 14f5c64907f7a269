// engine_core.hip -- MI355X (gfx950) engine for the MINRMS exhaustive RMSD-vs-N
// search: life cycle of an engine (configuration, layouts, device memory), the
// key and result accessors of the C ABI declared in include/minrms_engine.h.
// The kernels live in search.hip, refine.hip, alternates.hip, orientations.hip,
// noalloc.hip (around the DP) and dp_sweep_kernels*.hip / dp_coop_kernels.hip (the DP).
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
#include "engine_internal.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}


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
std::mutex mu;
std::vector<HostSlot> host_slots;
static std::map<std::pair<int, std::string>, std::vector<Parked>> parked;

// size classes 1/8 of a power of two apart: growing requests settle quickly
static size_t size_class(size_t bytes)
{
  size_t step = 256;
  while (step * 8 < bytes) step *= 2;
  return (bytes + step - 1) / step * step;
}

size_t parked_bytes(int device)
{
  std::lock_guard<std::mutex> lock(mu);
  size_t total = 0;
  for (auto &kv : parked)
    if (kv.first.first == device)
      for (const Parked &b : kv.second) total += b.bytes;
  return total;
}

// device must be current
hipError_t take(int device, const char *tag, size_t bytes, void **out, size_t *got, size_t *clean)
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

void park(int device, const char *tag, void *ptr, size_t bytes, size_t clean)
{
  std::lock_guard<std::mutex> lock(mu);
  parked[std::make_pair(device, std::string(tag))].push_back(Parked{ptr, bytes, clean});
}

void trim()
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

int build_layout(const minrms_cfg &c, int max_pb, bool uniform_pb, DpLayout *out, int min_panels)
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
    // the sweep kernels leave out the last 8 columns of a row when they lie outside the table (MINRMS_NO_TAIL8=1: test hook)
    d.tail8 = (!uniform_pb && d.P * d.PB - ncols >= 8 && !getenv("MINRMS_NO_TAIL8")) ? 1 : 0;
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
std::atomic<int> live_engines{0};

static void free_all(minrms_engine *e)
{
  for (const minrms_engine::DevBlock &b : e->dev_blocks) pool::park(e->cfg.device, b.tag, b.ptr, b.bytes, b.clean);
  e->dev_blocks.clear();
  if (e->d_upd_best) { (void)hipFree(e->d_upd_best); (void)hipFree(e->d_upd_chunk); e->d_upd_best = nullptr; }
  if (e->d_upd_counts) { (void)hipFree(e->d_upd_counts); e->d_upd_counts = nullptr; }
  if (e->d_alt_best) { (void)hipFree(e->d_alt_rec); (void)hipFree(e->d_alt_best); (void)hipFree(e->d_alt_chunk); (void)hipFree(e->d_alt_offsets); (void)hipFree(e->d_alt_count); e->d_alt_best = nullptr; }
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->h_ctrl) {
    std::lock_guard<std::mutex> lock(pool::mu);
    pool::host_slots.push_back(pool::HostSlot{e->cfg.device, e->h_ctrl, {e->ctrl_event[0], e->ctrl_event[1]}});
    e->h_ctrl = nullptr;
  }
  if (e->stream) (void)hipStreamDestroy(e->stream);
  if (e->stream_b) (void)hipStreamDestroy(e->stream_b);
}

int load_code_objects()
{
  HIP_TRY(touch_unit_search());
  HIP_TRY(touch_unit_refine());
  HIP_TRY(touch_unit_alternates());
  HIP_TRY(touch_unit_orientations());
  HIP_TRY(touch_unit_dp_sweep());
  HIP_TRY(touch_unit_dp_coop());
  return 0;
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
float clip_after(const minrms_engine *e, int64_t calls)
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
int configure(minrms_engine *e, minrms_cfg c)
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
  // (the tracebacks address a table's back-pointer words with 32-bit byte offsets)
  if (e->L.bp_words >= ((int64_t)1 << 30)) return fail(MINRMS_E_ARG, "the back-pointers of one table exceed 4 GB");
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
        int most = 50;                                   // tasks of the first block
        if (const char *sm = getenv("MINRMS_COOP_SEG_HALF_MAX_TASKS")) most = atoi(sm);   // tuning hook
        // the narrowest of 8, 16 and 32 columns that keeps the first block within `most` tasks (400, 800 columns)
        // (refinement at 300x300: 91.1 ms with 32-column tasks, 70.4 with 16, 67.1 with 8; at 400x400 163.6, 131.4, 132.1;
        // with the short poll loops of the fetch waves, 16 against 8 columns: 78.0 / 73.3 ms at 330x330,
        // 90.4 / 87.3 at 360x360, 129.5 / 127.1 at 400x400, no difference at 450x450; 32 against 16 columns at
        // 700x700: 660 / 585 ms)
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
      if (Lc.bp_words >= ((int64_t)1 << 30)) return fail(MINRMS_E_ARG, "the back-pointers of one table exceed 4 GB");
      e->Lc = Lc;
      e->coop_pb = pb;
      break;
    }
  }
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
  if (e->coop_pb != 0 && (e->coop_groups.n_groups > 1 || e->coop_seg.n_groups > 1)) {
    e->gprog_bytes = (size_t)(4 * e->coop_max_batch * e->coop_groups.total_tasks);
    ALLOC(e->d_gprog, (int64_t)e->gprog_bytes);
  }
  if (e->coop_pb != 0 && e->coop_seg.n_groups > 1 && e->coop_seg.xedge_granules > 0) {
    e->xedge_bytes = (size_t)(8 * e->coop_seg_batch * e->coop_seg.xedge_granules);
    ALLOC(e->d_xedge, (int64_t)e->xedge_bytes);
  }
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
  he = hipStreamCreate(&e->stream_b);
  if (he != hipSuccess) { free_all(e); delete e; return fail(MINRMS_E_DEVICE, "hipStreamCreate: %s", hipGetErrorString(he)); }
  if (e->d_gprog) {
    // the block may come from an earlier engine: its row counters must not read as tags of this one's launches
    he = hipMemsetAsync(e->d_gprog, 0, e->gprog_bytes, e->stream);
    if (he != hipSuccess) { free_all(e); delete e; return fail(MINRMS_E_DEVICE, "hipMemsetAsync: %s", hipGetErrorString(he)); }
  }
  if (e->d_xedge) {
    // ... and neither must the tags of its granules
    he = hipMemsetAsync(e->d_xedge, 0, e->xedge_bytes, e->stream);
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
  // test hook: start the launch counter close to the value at which the tags wrap
  if (const char *v = getenv("MINRMS_COOP_LAUNCHES_START")) e->coop_launches = std::max(0, atoi(v));
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
  if (e->stream_b) (void)hipStreamSynchronize(e->stream_b);
  free_all(e);
  delete e;
  live_engines.fetch_sub(1);
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
  if (e->num_solutions > 1 && e->alt_by_pairs && !e->comm_allreduce_min)
    return fail(MINRMS_E_ARG, "alternate solutions by matched pairs with a split search need minrms_engine_set_collectives");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipMemcpyAsync(e->d_keys, keys, 8 * e->sp.range_n, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->tables_current = false;
  return MINRMS_OK;
}

extern "C" int minrms_engine_set_collectives(minrms_engine *e, minrms_allreduce_min_fn allreduce_min,
                                             minrms_allgather_i64_fn allgather, void *user)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if ((allreduce_min == nullptr) != (allgather == nullptr)) return fail(MINRMS_E_ARG, "both collectives or none");
  e->comm_allreduce_min = allreduce_min;
  e->comm_allgather = allgather;
  e->comm_user = user;
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

// A cooperative DP kernel whose bounded wait for a neighbour task ran out
// raises d_err and finishes on whatever data it has: everything computed since
// is unusable.  Called (device current) wherever the host has just waited for
// the stream and is about to read or publish results; the failure sticks until
// minrms_engine_reset().
int check_device_error(minrms_engine *e)
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

void time_mark(minrms_engine *e, int cls)
{
  if (e->ev_used == e->ev_pool.size()) {
    hipEvent_t n;
    if (hipEventCreate(&n) != hipSuccess) { e->timing = 0; return; }
    e->ev_pool.push_back(n);
    e->ev_class.push_back(-1);
  }
  if (hipEventRecord(e->ev_pool[e->ev_used], e->stream) != hipSuccess) { e->timing = 0; return; }
  e->ev_class[e->ev_used++] = cls;
}

// Waits for the stream and adds the pending event pairs to the totals.
static int drain_marks(minrms_engine *e)
{
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->stream_b) HIP_TRY(hipStreamSynchronize(e->stream_b));
  for (size_t q = 0; q + 1 < e->ev_used; ++q) {
    const int cls = e->ev_class[q];
    if (cls < 0 || cls >= MINRMS_KC_COUNT) continue;
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e->ev_pool[q], e->ev_pool[q + 1]));
    e->kc_ms[cls] += t;
    e->kc_launches[cls] += 1;
  }
  e->ev_used = 0;
  return 0;
}

extern "C" int minrms_engine_enable_timing(minrms_engine *e, int32_t level)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (e->timing && !level) time_mark(e, -1);       // closes what the last mark opened
  e->timing = level < 0 ? 0 : level;
  return MINRMS_OK;
}

extern "C" int minrms_engine_kernel_time(minrms_engine *e, int32_t kernel_class, double *ms, int64_t *launches)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (kernel_class < 0 || kernel_class >= MINRMS_KC_COUNT) return fail(MINRMS_E_ARG, "no kernel class %d", kernel_class);
  { int rc = drain_marks(e); if (rc) return rc; }
  if (ms) *ms = e->kc_ms[kernel_class];
  if (launches) *launches = e->kc_launches[kernel_class];
  e->kc_ms[kernel_class] = 0.0; e->kc_launches[kernel_class] = 0;
  return MINRMS_OK;
}

extern "C" int minrms_engine_dp_block_time(minrms_engine *e, int32_t block, double *ms,
                                           int64_t *launches, int64_t *orientations,
                                           int64_t *cells_per_orientation)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (block < 0 || block >= e->L.NB) return fail(MINRMS_E_ARG, "no layer block %d", block);
  { int rc = minrms_engine_kernel_time(e, MINRMS_KC_DP_BLOCK0 + block, ms, launches); if (rc) return rc; }
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
  e->blk_orients[block] = 0;
  return MINRMS_OK;
}

extern "C" int minrms_engine_dp_block_columns(minrms_engine *e, int32_t block, int32_t *columns)
{
  if (!e || !columns || block < 0 || block >= e->L.NB) return fail(MINRMS_E_ARG, "bad argument");
  const BlockDesc &bd = e->L.blk[block];
  *columns = bd.P * bd.PB - (bd.tail8 ? 8 : 0);
  return MINRMS_OK;
}

extern "C" int minrms_engine_coop_shape(minrms_engine *e, int32_t *tasks, int32_t *task_width, int64_t *cells_visited,
                                        int32_t *rows_first_block, int32_t *tasks_first_block, int32_t *workgroups)
{
  if (!e) return fail(MINRMS_E_ARG, "null argument");
  if (e->coop_pb == 0) return fail(MINRMS_E_STATE, "this engine has no cooperative layout");
  // the spread the refinement rounds use: segments if the engine has them, else whole blocks
  const bool seg = e->coop_seg.n_groups > 1 && e->d_xedge != nullptr;
  const DpLayout &Lk = seg ? e->Lc_seg : e->Lc;
  const dp::CoopGroups &G = seg ? e->coop_seg : e->coop_groups;
  int64_t cells = 0;
  for (int b = 0; b < Lk.NB; ++b) cells += (int64_t)Lk.blk[b].nrows * Lk.blk[b].P * Lk.blk[b].PB;
  if (tasks) *tasks = G.total_tasks;
  if (task_width) *task_width = seg ? e->coop_seg_pb : e->coop_pb;
  if (cells_visited) *cells_visited = cells;
  if (rows_first_block) *rows_first_block = Lk.blk[0].nrows;
  if (tasks_first_block) *tasks_first_block = Lk.blk[0].P;
  if (workgroups) *workgroups = G.n_groups;
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

