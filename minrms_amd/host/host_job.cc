// host_job.cc -- one search (+ refinement) through the C++ host class of
// search_or_hip.h, for callers and tests that have no minrms source tree:
//   minrms_host_job <job file> <result file>
// Job file (native byte order): 16 x int32 {N1, N2, num_bb, n_min, n_max,
// refine_while_searching, orientation_filter_nw, of_min_num_matches, num_solutions,
// alt_method (0 = 3D, 1 = 3D_ORIG, 2 = PAIRS), alt_max_pairs_common,
// refine (0 = none, 1 = best, 2 = all ranks), n_orientations, count_updates, start, stop},
// 6 x float {max_rmsd, max_dist, of_max_dist, alt_min_3d_difference, alt_rmsd_tolerance, 0},
// then xyzA [N1*num_bb*3], xyzB [N2*num_bb*3], the orientations [n*12] as float.
// Result file: per rank the tables after Search (ssd [range_n] float, transforms
// [range_n*12] float, alignments [range_n*n_max*2] int16); with refinement
// 2 x int64 {iterations, new solutions} and the tables again; with
// count_updates the per-orientation counts [stop-start] int32 at the end.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "search_or_hip.h"

template <class T>
static void get(std::FILE *f, T *p, size_t n)
{
  if (n && std::fread(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "job file too short\n"); std::exit(2); }
}
template <class T>
static void put(std::FILE *f, const T *p, size_t n)
{
  if (n && std::fwrite(p, sizeof(T), n, f) != n) { std::fprintf(stderr, "cannot write the result file\n"); std::exit(2); }
}

int main(int argc, char **argv)
{
  if (argc != 3) { std::fprintf(stderr, "usage: %s <job file> <result file>\n", argv[0]); return 2; }
  std::FILE *in = std::fopen(argv[1], "rb");
  if (!in) { std::fprintf(stderr, "cannot open %s\n", argv[1]); return 2; }
  int32_t h[16];
  float x[6];
  get(in, h, 16);
  get(in, x, 6);
  minrms_hip::SearchOrHip::Settings s;
  s.N1 = h[0]; s.N2 = h[1]; s.num_bb = h[2]; s.n_min = h[3]; s.n_max = h[4];
  s.refine_while_searching = h[5] != 0;
  s.orientation_filter_nw = h[6] != 0;
  s.orientation_filter_nw_min_num_matches = h[7];
  s.num_solutions = h[8];
  s.alt_method = (minrms_hip::SearchOrHip::Settings::AltMethod)h[9];
  s.alt_max_pairs_common = h[10];
  const int refine = h[11];
  s.refine_all = refine == 2;
  const long n_og = h[12];
  s.count_updates = h[13] != 0;
  const long start = h[14], stop = h[15];
  s.dyn3d_max_rmsd = x[0]; s.dyn3d_max_dist = x[1];
  s.orientation_filter_nw_max_dist = x[2];
  s.alt_min_3d_difference = x[3]; s.alt_rmsd_tolerance = x[4];
  s.workspace_bytes = 1ll << 30;
  if (s.N1 < 0 || s.N2 < 0 || s.num_bb < 0 || n_og < 0) { std::fprintf(stderr, "bad job header\n"); return 2; }
  std::vector<float> a((size_t)s.N1 * s.num_bb * 3), b((size_t)s.N2 * s.num_bb * 3), og((size_t)n_og * 12);
  get(in, a.data(), a.size());
  get(in, b.data(), b.size());
  get(in, og.data(), og.size());
  std::fclose(in);

  minrms_hip::SearchOrHip so(s, a.data(), b.data());
  std::FILE *out = std::fopen(argv[2], "wb");
  if (!out) { std::fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
  std::vector<float> ssd, xf;
  std::vector<int16_t> al;
  const int ranks = s.num_solutions > 1 ? s.num_solutions : 1;
  auto dump = [&]() {
    for (int r = 0; r < ranks; ++r) {
      if (ranks > 1) so.rank_tables(r, ssd, xf, al);
      else { ssd = so.aaSumSqdDists; xf = so.aaTransforms; al = so.aaAlignments; }
      put(out, ssd.data(), ssd.size());
      put(out, xf.data(), xf.size());
      put(out, al.data(), al.size());
    }
  };
  so.Search(og.data(), n_og, start, stop);
  dump();
  if (refine) {
    so.Refine();
    const int64_t st[2] = {so.refine_stats.num_iterations, so.refine_stats.num_new_solutions};
    put(out, st, 2);
    dump();
  }
  if (s.count_updates) put(out, so.updates.data(), so.updates.size());
  std::fclose(out);
  return 0;
}
