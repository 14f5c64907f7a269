// See search_or_hip_minrms.h.  Build: oracle/Makefile target `hipbin`.
#include <vector>
#include <iostream>
using namespace std;

#include <global_utils.h>
#include <simple_numeric_utils.h>
#include <distance_metric.h>
#include "search_or_hip_minrms.h"
#include "../include/minrms_engine.h"

namespace minrms {

// Flat copy of the backbone atoms, the way the reference makes its own
// (bin/minrms/search_or_dyn3d.cc:186-224).
static void Flatten(Biopolymer const &m, vector<float> &xyz)
{
  xyz.clear();
  for (Biopolymer::const_iterator ps = m.begin(); ps < m.end(); ++ps)
    for (int q = 0; q < Biopolymer::Residue::NumBackboneAtoms(); ++q) {
      Biopolymer::Residue::const_iterator pa = ps->GetBackboneAtom(q);
      for (int d = 0; d < 3; ++d) xyz.push_back((*pa).second.xyz[d]);
    }
}

static void Check(int rc)
{
  if (rc != MINRMS_OK) ERR(minrms_engine_last_error());
}

SearchOrHipMinrms::SearchOrHipMinrms(SearchOrDyn3d::Settings const &s)
    : SearchOrDyn3d(s), engine(NULL), refine_max_iters(s.refine_max_iters)
{
  vector<float> a, b;
  Flatten(s.aMol_f[0], a);
  Flatten(s.aMol_f[1], b);
  minrms_cfg c = minrms_cfg();
  c.N1 = s.aMol_f[0].size();
  c.N2 = s.aMol_f[1].size();
  c.num_bb = Biopolymer::Residue::NumBackboneAtoms();
  c.n_min = s.n_min;
  c.n_max = s.n_max;
  c.max_rmsd = s.dyn3d_max_rmsd;
  c.max_dist = s.dyn3d_max_dist;
  c.refine_while_searching = s.refine_while_searching;
  c.of_enable = s.orientation_filter_nw;
  c.of_max_dist = s.orientation_filter_nw_max_dist;
  c.of_min_matches = s.orientation_filter_nw_min_num_matches;
  c.num_solutions = s.num_solutions;
  c.alt_min_3d_difference = s.alt_min_3d_difference;
  c.alt_rmsd_tolerance = s.alt_rmsd_tolerance;
  c.alt_by_pairs = s.alt_method == SearchOr::Settings::ALT_METHOD_PAIRS;
  c.alt_max_pairs_common = s.alt_max_pairs_common;
  c.refine_all = (s.refine_method == SearchOr::Settings::REFINE_ALL);
  Check(minrms_engine_create(&c, &a[0], &b[0], &engine));
}

SearchOrHipMinrms::~SearchOrHipMinrms() { minrms_engine_destroy(engine); }

void SearchOrHipMinrms::Search(OrientationGenerator const &og, long start, long stop)
{
  vector<float> flat(12 * (size_t)og.size());
  for (long o = 0; o < (long)og.size(); ++o)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) flat[12 * o + 4 * r + c] = og[o].transform[r][c];
  Check(minrms_engine_count_updates(engine, 1));
  Check(minrms_engine_search(engine, &flat[0], og.size(), start, stop));
  // What FirstPass reports while it runs (search_or_dyn3d.cc:1885-1889), after the fact: the sweep
  // is over in a fraction of the second below which the reference adds no time estimate either.
  {
    vector<int32_t> updates(stop > start ? stop - start : 0);
    if (!updates.empty()) Check(minrms_engine_get_update_counts(engine, &updates[0], stop - start));
    for (long o = start; o < stop; ++o)
      if (updates[o - start] > 0) cout << updates[o - start] << " updates for orientation " << o + 1 << endl;
  }
  Check(minrms_engine_finalize_search(engine));
  FetchTables();
}

void SearchOrHipMinrms::Refine()
{
  cout << "Entering refinement stage." << endl;
  minrms_refine_stats st;
  Check(minrms_engine_refine(engine, refine_max_iters, &st));
  FetchTables();
  cout << "(During refinement, " << st.num_iterations << " new orientations were tested,\n"
          " and " << st.num_new_solutions << " instances of better alignments were found.)" << endl;
}

void SearchOrHipMinrms::FetchTables()
{
  vector<float> ssd(range_of_n), xf(12 * (size_t)range_of_n);
  vector<int16_t> pairs(2 * (size_t)range_of_n * n_max);
  for (int rank = 0; rank < settings.num_solutions; ++rank) {
    Check(minrms_engine_results_rank(engine, rank, &ssd[0], &xf[0], &pairs[0]));
    for (int k = 0; k < range_of_n; ++k) {
      int n = n_min + k;
      aaSumSqdDists[rank][k] = ssd[k];
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) aaTransforms[rank][k][r][c] = xf[12 * k + 4 * r + c];
      if (ssd[k] < DistanceMetric::UPPER_BOUND) {
        aaAlignments[rank][k].resize(n);
        for (int m = 0; m < n; ++m) {
          aaAlignments[rank][k][m][0] = pairs[2 * ((size_t)k * n_max + m)];
          aaAlignments[rank][k][m][1] = pairs[2 * ((size_t)k * n_max + m) + 1];
        }
      }
    }
  }
}

}  // namespace minrms
