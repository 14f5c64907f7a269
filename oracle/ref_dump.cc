// ref_dump -- golden-vector generator.  TEST INFRASTRUCTURE ONLY.
//
// Our own driver around the *real* reference objects (compiled in place from
// /root/reference by oracle/Makefile, never copied).  It accepts the minrms
// command line, runs the reference's own OrientationGenerator / SearchOrDyn3d
// exactly as bin/minrms/main.cc:106-167 does, and writes what the stock binary
// never prints: the shuffled orientation list, and the per-N result tables
// (sum-squared-distance bits, transform, residue pairs) after Search() and after
// Refine().  With REF_DUMP_TRACE=1 it additionally replays the orientation
// sweep through the reference's public Dyn3d / CalcMinRMSD primitives (the
// sequence of calls of search_or_dyn3d.cc:1676-1851) to record per-(o_id, n)
// values and the winning o_id per n, and cross-checks that replay against the
// tables the real Search() produced.
//
// Output: a line-oriented text file ($REF_DUMP_OUT, default "ref_dump.txt");
// every float is written as the hex of its IEEE-754 bits.

#include <vector>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cassert>
using namespace std;

#include <sys/time.h>
#include <global_utils.h>
#include <simple_numeric_utils.h>
#include <parse_utils.h>
// principal axes of OrientationGenerator are private too (or_gen.h:226-227)
#define private public
#include <or_gen.h>
#undef private
#include <orientation_filter.h>
#include <apply_trans.h>
#include <distance_metric.h>
#include <dyn3d.h>
#include "search_or.h"
// The result tables are private members of SearchOrDyn3d
// (search_or_dyn3d.h:293-310); open the class up for the dump.
#define private public
#define protected public
#include "search_or_dyn3d.h"
#undef private
#undef protected
#include "minrms_parser.h"

using namespace minrms;

// main.cc:51-52 defines these two globals; other translation units print them.
const char g_version_string[] = "1.0.1";
const char g_date_string[]    = "<ref_dump>";

extern "C" void srand48(long);

static unsigned fbits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

// rank 0 (the best alignments) under `tag`; the alternates (rank 1, 2, ...),
// if any were asked for, under "<tag>_ALT <rank>"
static void dump_tables(FILE *fp, const char *tag, SearchOrDyn3d &s, int num_solutions)
{
  for (int rank = 0; rank < num_solutions; ++rank) {
    if (rank == 0) fprintf(fp, "%s %d\n", tag, s.range_of_n);
    else fprintf(fp, "%s_ALT %d %d\n", tag, rank, s.range_of_n);
    for (int n = s.n_min; n <= s.n_max; ++n) {
      int k = n - s.n_min;
      fprintf(fp, "%d %08x", n, fbits(s.aaSumSqdDists[rank][k]));
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
          fprintf(fp, " %08x", fbits(s.aaTransforms[rank][k][r][c]));
      PairwiseAlignment const &a = s.aaAlignments[rank][k];
      bool have = s.aaSumSqdDists[rank][k] < DistanceMetric::UPPER_BOUND;
      fprintf(fp, " %d", have ? a.NumMatches() : 0);
      if (have)
        for (int m = 0; m < a.NumMatches(); ++m)
          fprintf(fp, " %d %d", (int)a[m][0], (int)a[m][1]);
      fprintf(fp, "\n");
    }
  }
}

int main(int argc, char **argv)
{
  srand48(42);   // main.cc:194
  const char *out_name = getenv("REF_DUMP_OUT");
  if (!out_name) out_name = "ref_dump.txt";
  bool trace = getenv("REF_DUMP_TRACE") && atoi(getenv("REF_DUMP_TRACE")) != 0;
  // REF_DUMP_TRACE=2 also writes every (o_id, n) record, not only the winners.
  bool trace_all = getenv("REF_DUMP_TRACE") && atoi(getenv("REF_DUMP_TRACE")) >= 2;

  vector<string> vOrigArgs, vArgs;
  ParseUtils::ConvertArgvToVectorOfStrings(argc, argv, vOrigArgs);
  char back_slash[2] = {'\\', '\0'};
  ParseUtils::HandleArgFiles(vOrigArgs, vArgs, "-cfg", "#", back_slash, string());
  MinrmsParser settings(vArgs);
  if (vArgs.size() != 1)
    ERR("Error: Unrecognized command line option: \"" << vArgs[1] << "\"\n");

  FILE *fp = fopen(out_name, "w");
  if (!fp) { perror(out_name); return 1; }

  int const nbb = Biopolymer::Residue::NumBackboneAtoms();
  int const N1 = settings.aMol_f[0].size(), N2 = settings.aMol_f[1].size();
  fprintf(fp, "DIMS %d %d %d\n", N1, N2, nbb);
  fprintf(fp, "SETTINGS n_min %d n_max %d max_rmsd %08x max_dist %08x rws %d "
              "refine_method %d refine_max_iters %ld num_solutions %d "
              "of %d of_dist %08x of_min %d fm %d fm_cut %08x "
              "alt_method %d alt_min_3d %08x alt_tol %08x alt_pairs %d\n",
          settings.n_min, settings.n_max,
          fbits(settings.dyn3d_max_rmsd), fbits(settings.dyn3d_max_dist),
          (int)settings.refine_while_searching, (int)settings.refine_method,
          (long)settings.refine_max_iters, settings.num_solutions,
          (int)settings.orientation_filter_nw,
          fbits(settings.orientation_filter_nw_max_dist),
          settings.orientation_filter_nw_min_num_matches,
          settings.fm_fragment_size, fbits(settings.fm_cutoff_rmsd),
          (int)settings.alt_method, fbits(settings.alt_min_3d_difference),
          fbits(settings.alt_rmsd_tolerance), settings.alt_max_pairs_common);
  for (int m = 0; m < 2; ++m) {
    fprintf(fp, "XYZ%c", 'A' + m);
    for (Biopolymer::const_iterator ps = settings.aMol_f[m].begin();
         ps < settings.aMol_f[m].end(); ++ps)
      for (int q = 0; q < nbb; ++q) {
        Biopolymer::Residue::const_iterator pa = ps->GetBackboneAtom(q);
        for (int d = 0; d < 3; ++d)
          fprintf(fp, " %08x", fbits((*pa).second.xyz[d]));
      }
    fprintf(fp, "\n");
  }

  OrientationGenerator og(settings);
  // the interval pairs fragment matching was restricted to, as 0-based
  // inclusive residue indices (or_gen.cc:359-384)
  fprintf(fp, "IV %ld\n", (long)settings.fm_interval_pair_list.size());
  for (size_t q = 0; q < settings.fm_interval_pair_list.size(); ++q) {
    Biopolymer const &mA = settings.aMol_f[0], &mB = settings.aMol_f[1];
    fprintf(fp, "%ld %ld %ld %ld\n",
            (long)(mA.find(settings.fm_interval_pair_list[q][0].first) - mA.begin()),
            (long)(mA.find(settings.fm_interval_pair_list[q][0].last) - mA.begin()),
            (long)(mB.find(settings.fm_interval_pair_list[q][1].first) - mB.begin()),
            (long)(mB.find(settings.fm_interval_pair_list[q][1].last) - mB.begin()));
  }
  // what OrientationGenerator::SumSqdDistBetweenOrientations works with
  // (or_gen.cc:94-110,158-165): eigenvalues and axes+centroid of structure 2
  fprintf(fp, "PAXIS");
  for (int d = 0; d < 3; ++d) fprintf(fp, " %08x", fbits(og.aPrincipleAxisEigenvalues[1][d]));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) fprintf(fp, " %08x", fbits(og.aPrincipleAxisAndCentroid[1][r][c]));
  fprintf(fp, "\n");
  fprintf(fp, "OG %ld\n", (long)og.size());
  for (long o = 0; o < (long)og.size(); ++o) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c)
        fprintf(fp, "%08x%c", fbits(og[o].transform[r][c]), (r == 2 && c == 3) ? '\n' : ' ');
  }

  SearchOrDyn3d *pS = new SearchOrDyn3d(settings);
  if (getenv("REF_DUMP_BENCH")) {
    // CPU-baseline mode: time the reference's own Search() over the first K
    // orientations of the shuffled list and stop (no dump, no refinement).
    // REF_DUMP_START=S moves the timed range to [S, S+K): one process per host
    // core, each with its own range, is the all-cores baseline.
    long K = atol(getenv("REF_DUMP_BENCH"));
    long S = getenv("REF_DUMP_START") ? atol(getenv("REF_DUMP_START")) : 0;
    if (S < 0 || S >= (long)og.size()) S = 0;
    if (K < 1 || S + K > (long)og.size()) K = og.size() - S;
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    pS->Search(og, S, S + K);
    gettimeofday(&t1, NULL);
    double sec = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);
    printf("REF_BENCH orientations %ld of %ld seconds %.6f\n", K, (long)og.size(), sec);
    fclose(fp);
    return 0;
  }
  // REF_DUMP_LIMIT=K: sweep only the orientations below K of the shuffled list
  // (large structures, where the full sweep takes the reference hours);
  // REF_DUMP_START=S: begin at orientation S -- Search(og, start, stop) is the
  // reference's own seam for splitting a sweep (search_or.h:92-103), so several
  // processes can each take one range of a long sweep.
  long n_search = og.size(), first = 0;
  if (getenv("REF_DUMP_LIMIT")) {
    long K = atol(getenv("REF_DUMP_LIMIT"));
    if (K >= 1 && K < n_search) n_search = K;
  }
  if (getenv("REF_DUMP_START")) {
    first = atol(getenv("REF_DUMP_START"));
    if (first < 0 || first > n_search) first = n_search;
    if (trace) { fprintf(stderr, "ref_dump: REF_DUMP_START cannot be combined with REF_DUMP_TRACE\n"); return 2; }
  }
  fprintf(fp, "SEARCHED %ld %ld\n", n_search, first);
  if (const char *load = getenv("REF_DUMP_LOAD_SEARCH")) {
    // Refine() after a sweep that was split over several processes: the merged
    // per-N tables of Search() (one line per n: n, ssd bits, 12 transform words,
    // number of pairs, the pairs -- the format dump_tables writes) are put back
    // into the reference's own result tables, and the reference's own Refine()
    // runs on them below.  Refine() reads nothing else that Search() leaves
    // behind when neither -of nor -max-dist is given (search_or_dyn3d.cc:2348-2629).
    if (settings.num_solutions != 1 || settings.orientation_filter_nw) {
      fprintf(stderr, "ref_dump: REF_DUMP_LOAD_SEARCH takes neither alternates nor -of\n");
      return 2;
    }
    FILE *in = fopen(load, "r");
    if (!in) { perror(load); return 1; }
    for (int n = pS->n_min; n <= pS->n_max; ++n) {
      int k = n - pS->n_min, n_read = 0, n_pairs = 0;
      unsigned bits = 0;
      if (fscanf(in, "%d %x", &n_read, &bits) != 2 || n_read != n) { fprintf(stderr, "ref_dump: bad table line for n=%d\n", n); return 2; }
      memcpy(&pS->aaSumSqdDists[0][k], &bits, 4);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
          if (fscanf(in, "%x", &bits) != 1) return 2;
          memcpy(&pS->aaTransforms[0][k][r][c], &bits, 4);
        }
      if (fscanf(in, "%d", &n_pairs) != 1) return 2;
      if (n_pairs > 0) {
        PairwiseAlignment al(n_pairs);
        for (int m = 0; m < n_pairs; ++m) {
          int i = 0, j = 0;
          if (fscanf(in, "%d %d", &i, &j) != 2) return 2;
          al[m][0] = (short)i; al[m][1] = (short)j;
        }
        pS->aaAlignments[0][k] = al;
      }
    }
    fclose(in);
  } else
  pS->Search(og, first, n_search);
  dump_tables(fp, "SEARCH", *pS, settings.num_solutions);
  if (getenv("REF_DUMP_SEARCH_ONLY")) {   // one range of a split sweep: the caller merges
    fclose(fp);
    return 0;
  }

  if (trace) {
    // Replay of search_or_dyn3d.cc:1676-1851 with the reference's primitives.
    int const n_min = settings.n_min, n_max = settings.n_max;
    int const range_n = n_max - n_min + 1;
    Dyn3d dyn(N1, N2, n_min, n_max);
    dyn.pCostFunc = &DistanceMetric::SquaredDistanceBetweenIandJ;
    // the replay must leave the process-wide cutoff as Search() left it, so
    // that Refine() below sees what it would see without the replay
    float const metric_after_search = DistanceMetric::GetMaxMetricDistance();
    DistanceMetric::SetMaxPhysicalDistance(settings.dyn3d_max_dist);
    Biopolymer molA = settings.aMol_f[0], molB = settings.aMol_f[1];
    Superimpose sup(MIN(N1, N2) * nbb);
    Vect3 *w1 = new Vect3[N1 * nbb], *w2 = new Vect3[N2 * nbb];
    OrientationFilter *pFilter = NULL;
    if (settings.orientation_filter_nw)
      pFilter = new OrientationFilter(settings.orientation_filter_nw_max_dist,
                                      settings.orientation_filter_nw_min_num_matches,
                                      N1, N2);
    vector<float> best(range_n, DistanceMetric::UPPER_BOUND);
    vector<long> best_o(range_n, -1);
    vector<PairwiseAlignment> best_al(range_n);
    PairwiseAlignment al(MIN(N1, N2));
    long n_pass = 0;
    fprintf(fp, "TRACE %d\n", trace_all ? 1 : 0);
    for (long o = 0; o < n_search; ++o) {
      ApplyTransformBackbone(og[o].transform, settings.aMol_f[1], molB);
      if (pFilter && !pFilter->GoodOrientation(molA, molB)) {
        fprintf(fp, "F %ld\n", o);
        continue;
      }
      ++n_pass;
      dyn.FillLookupTable(molA, molB);
      dyn.InitTable_IncrN();
      for (int n = 2; n <= n_max; ++n) {
        dyn.Solve_IncrN(n);
        if (dyn.MaxDistanceExceeded(n)) break;
        if (n < n_min) continue;
        al.resize(n);
        dyn.ExportSolution(n, al);
        float ssd_orig = dyn.GetCost(n), ssd = ssd_orig;
        Matrix3x4 xf;
        if (settings.refine_while_searching) {
          float rmsd = al.CalcMinRMSD(settings.aMol_f[0], settings.aMol_f[1], xf, &sup, w1, w2);
          ssd = SQR(rmsd) * n * nbb;
        }
        if ((settings.dyn3d_max_rmsd != settings.dyn3d_NO_MAX_RMSD) &&
            (ssd > n * nbb * SQR(settings.dyn3d_max_rmsd)))
          break;
        if (trace_all) {
          fprintf(fp, "T %ld %d %08x %08x", o, n, fbits(ssd_orig), fbits(ssd));
          for (int m = 0; m < n; ++m) fprintf(fp, " %d %d", (int)al[m][0], (int)al[m][1]);
          fprintf(fp, "\n");
        }
        if (ssd < best[n - n_min]) {
          best[n - n_min] = ssd; best_o[n - n_min] = o; best_al[n - n_min] = al;
        }
      }
    }
    fprintf(fp, "WINNERS %d passed %ld\n", range_n, n_pass);
    for (int n = n_min; n <= n_max; ++n) {
      int k = n - n_min;
      fprintf(fp, "%d %ld %08x\n", n, best_o[k], fbits(best[k]));
      // the replay must agree with what the real Search() recorded
      if (fbits(best[k]) != fbits(pS->aaSumSqdDists[0][k]) ||
          (best_o[k] >= 0 && !(best_al[k] == pS->aaAlignments[0][k]))) {
        fprintf(stderr, "ref_dump: replay disagrees with Search() at n=%d\n", n);
        return 2;
      }
    }
    fprintf(fp, "CLIP %08x %08x\n", fbits(DistanceMetric::GetMaxMetricDistance()),
            fbits(metric_after_search));
    DistanceMetric::SetMaxMetricDistance(metric_after_search);
  }

  if (settings.refine_method != SearchOr::Settings::NO_REFINE) {
    pS->Refine();
    dump_tables(fp, "REFINE", *pS, settings.num_solutions);
  }
  fclose(fp);
  // Also let the reference write its own files (MSF, plot) into the cwd.
  pS->WriteMSF();
  pS->WriteChimeraInfo();
  return 0;
}
