// search_or_hip.h -- C++ host side above the C ABI, shaped like the reference's
// search plug-in interface (class SearchOr, bin/minrms/search_or.h:57-317):
// construct from the settings, Search(og, start, stop), Refine(), then read the
// three result tables (search_or_dyn3d.h:293-310, r = 0).  Header-only; link
// with libminrms_hip.so.  Errors behave like the reference's ERR() macro
// (lib/global_utils/ERR.h:20-21): message to stderr, exit(-1).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/minrms_engine.h"

namespace minrms_hip {

class SearchOrHip {
 public:
  // The members of SearchOrDyn3d::Settings / SearchOr::Settings the path reads.
  struct Settings {
    int N1 = 0, N2 = 0, num_bb = 1;
    int n_min = 1, n_max = -1;                 // -1 = NO_MAX_NUM_MATCHES
    float dyn3d_max_rmsd = -1.0f;              // dyn3d_NO_MAX_RMSD
    float dyn3d_max_dist = -1.0f;              // NO_MAX_PHYS_DISTANCE
    bool refine_while_searching = true;
    long refine_max_iters = -1;                // ITERATE_UNTIL_CONVERGENCE
    bool orientation_filter_nw = false;        // -of (bin/minrms/search_or_dyn3d.h:131-140)
    float orientation_filter_nw_max_dist = 0.0f;
    int orientation_filter_nw_min_num_matches = 0;
    // alternate solutions (bin/minrms/search_or.h:154-244)
    int num_solutions = 1;
    enum AltMethod { ALT_METHOD_3D, ALT_METHOD_3D_ORIG, ALT_METHOD_PAIRS } alt_method = ALT_METHOD_3D;
    float alt_min_3d_difference = 0.0f;
    float alt_rmsd_tolerance = -1.0f;          // ALTERNATES_NOT_LIMITED_BY_RMSD
    int alt_max_pairs_common = 0;
    bool refine_all = false;                   // refine_method == REFINE_ALL
    bool count_updates = false;                // keep the "<count> updates for orientation <o>" numbers of a search
    int device = 0;
    long long workspace_bytes = 0;             // 0 = the engine's default
  };

  SearchOrHip(const Settings &s, const float *xyzA, const float *xyzB) : settings(s)
  {
    minrms_cfg c = {};
    c.N1 = s.N1; c.N2 = s.N2; c.num_bb = s.num_bb;
    c.n_min = s.n_min;
    c.n_max = s.n_max < 0 ? (s.N1 < s.N2 ? s.N1 : s.N2) : s.n_max;
    c.max_rmsd = s.dyn3d_max_rmsd; c.max_dist = s.dyn3d_max_dist;
    c.refine_while_searching = s.refine_while_searching ? 1 : 0;
    c.of_enable = s.orientation_filter_nw ? 1 : 0;
    c.of_max_dist = s.orientation_filter_nw_max_dist;
    c.of_min_matches = s.orientation_filter_nw_min_num_matches;
    c.num_solutions = s.num_solutions;
    c.alt_min_3d_difference = s.alt_min_3d_difference;
    c.alt_rmsd_tolerance = s.alt_rmsd_tolerance;
    c.alt_by_pairs = s.alt_method == Settings::ALT_METHOD_PAIRS ? 1 : 0;
    c.alt_max_pairs_common = s.alt_max_pairs_common;
    c.refine_all = s.refine_all ? 1 : 0;
    c.device = s.device;
    c.workspace_bytes = s.workspace_bytes;
    n_min = c.n_min; n_max = c.n_max; range_of_n = n_max - n_min + 1;
    check(minrms_engine_create(&c, xyzA, xyzB, &engine));
    aaSumSqdDists.assign(range_of_n, MINRMS_UPPER_BOUND);
    aaTransforms.assign((size_t)range_of_n * 12, 0.0f);
    aaAlignments.assign((size_t)range_of_n * n_max * 2, -1);
    winner_oid.assign(range_of_n, -1);
  }
  ~SearchOrHip() { minrms_engine_destroy(engine); }
  SearchOrHip(const SearchOrHip &) = delete;
  SearchOrHip &operator=(const SearchOrHip &) = delete;

  // SearchOr::Search(og, start, stop); og = og[i].transform for every i.
  void Search(const float *og, long n_og, long start, long stop)
  {
    if (settings.count_updates) check(minrms_engine_count_updates(engine, 1));
    check(minrms_engine_search(engine, og, n_og, start, stop));
    if (settings.count_updates) {
      // what FirstPass prints per orientation while it runs (search_or_dyn3d.cc:1885-1889)
      updates.assign(stop > start ? stop - start : 0, 0);
      if (!updates.empty()) check(minrms_engine_get_update_counts(engine, updates.data(), stop - start));
    }
    check(minrms_engine_finalize_search(engine));
    fetch();
  }
  // SearchOr::Refine()
  void Refine()
  {
    check(minrms_engine_refine(engine, settings.refine_max_iters, &refine_stats));
    fetch();
  }

  // aaAlignments[k] = n_min+k rows of {i, j}
  const int16_t *alignment(int k) const { return &aaAlignments[(size_t)k * n_max * 2]; }
  const float *transform(int k) const { return &aaTransforms[(size_t)k * 12]; }

  // Tables of rank `rank` (0 = best, 1 = 2nd best, ...) when settings.num_solutions > 1:
  // aaSumSqdDists[rank], aaTransforms[rank], aaAlignments[rank] of the reference.
  void rank_tables(int rank, std::vector<float> &ssd, std::vector<float> &xforms,
                   std::vector<int16_t> &alignments)
  {
    ssd.resize(range_of_n);
    xforms.resize((size_t)range_of_n * 12);
    alignments.resize((size_t)range_of_n * n_max * 2);
    check(minrms_engine_results_rank(engine, rank, ssd.data(), xforms.data(), alignments.data()));
  }

  Settings settings;
  int n_min, n_max, range_of_n;
  std::vector<float> aaSumSqdDists, aaTransforms;
  std::vector<int16_t> aaAlignments;
  std::vector<int64_t> winner_oid;
  std::vector<int32_t> updates;               // per orientation of the last Search(), with settings.count_updates
  minrms_refine_stats refine_stats = {};
  minrms_engine *engine = nullptr;

 private:
  void fetch()
  {
    check(minrms_engine_results(engine, aaSumSqdDists.data(), aaTransforms.data(),
                                aaAlignments.data(), winner_oid.data()));
  }
  static void check(int rc)
  {
    if (rc == MINRMS_OK) return;
    std::fprintf(stderr, "%s\n", minrms_engine_last_error());
    std::exit(-1);
  }
};

}  // namespace minrms_hip
