// alt_solutions.h -- host side of the alternate solutions by orientation
// (`-alt-o count,min_distance[,rmsd_ratio]`, SearchOr::Settings::ALT_METHOD_3D).
//
// The GPU produces, for every (orientation, n) the sweep evaluates, the sum of
// squared distances and the least-squares transform; everything below works on
// those records on the host, in the reference's order of operations:
//   candidates     KeepTrackOfAltSolutions    bin/minrms/search_or_dyn3d.cc:477-542
//   selection      ReadAltSolutionFile        :662-921 (+ :928-962)
//   consistency    CheckAlternatesForConsistency :1099-1247, SortAltList :1585-1640
//   metric         OrientationGenerator::SumSqdDistBetweenOrientations
//                  lib/or_gen/or_gen.cc:93-110, FastRotSumSqdDist
//                  lib/fast_rot_metric/fast_rot_metric.h:96-132, SlowRotSumSqdDist
//                  fast_rot_metric.cc:13-28, FindPrincipleAxis principle_axis.cc:30-160
//
// The reference orders candidates with std::sort, which is not stable: entries
// of equal RMSD come out in whatever order libstdc++'s introsort leaves them.
// The same calls on the same sequences are made here for that reason.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "jacobi_sym.h"

namespace alt {

struct Candidate {            // one record of the reference's per-n ".tmp" file
  int64_t o_id;
  float rmsd;
  float xf[12];
  float ssd;                  // not in the reference's record: the sum the RMSD was made from (lists of several engines are merged on it)
  bool operator<(const Candidate &b) const { return rmsd < b.rmsd; }
};

struct Owner {                // an orientation that has to be re-solved for (n, rank)
  int n, r;
  bool operator<(const Owner &b) const { return n < b.n || (n == b.n && r < b.r); }
};

struct Ranked {               // one entry of a rank table while it is being sorted
  float ssd;
  float xf[12];
  std::vector<int16_t> pairs;
  bool operator<(const Ranked &b) const { return ssd < b.ssd; }
};

inline bool same_matrix(const float *a, const float *b)          // vect_3d.h:403-411
{
  for (int q = 0; q < 12; ++q)
    if (a[q] != b[q]) return false;
  return true;
}

// x' = ((m0 x + m1 y) + m2 z) + m3, lib/vect_3d/vect_3d.h:145-155
inline void apply(const float *xf, const float *p, float *out)
{
  for (int r = 0; r < 3; ++r)
    out[r] = xf[4 * r] * p[0] + xf[4 * r + 1] * p[1] + xf[4 * r + 2] * p[2] + xf[4 * r + 3];
}

// A * [B; 0 0 0 1], lib/vect_3d/vect_3d.h:199-224
inline void mat_times_mat(const float *A, const float *B, float *dest)
{
  float t[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      t[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    t[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
  }
  memcpy(dest, t, sizeof t);
}

// Orientation metric of the moved structure: variance of its points along the
// principal axes (eigenvalues) and the axes + centroid as a 3x4 matrix.
struct OrientationMetric {
  float eig[3];
  float axes[12];
  int npts;

  void init(int n_points, const float *xyz)
  {
    npts = n_points;
    float c[3] = {0.0f, 0.0f, 0.0f};
    const float inv = 1.0f / (float)npts;
    for (int a = 0; a < npts; ++a)
      for (int k = 0; k < 3; ++k) c[k] += xyz[3 * a + k];
    for (int k = 0; k < 3; ++k) c[k] *= inv;
    float q[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = 0; a < npts; ++a) {
      const float r[3] = {xyz[3 * a] - c[0], xyz[3 * a + 1] - c[1], xyz[3 * a + 2] - c[2]};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) q[i][j] += r[i] * r[j];
    }
    float d[3], v[3][3];
    jac::jacobi_sym<3>(q, d, v);     // behaviour of lib/eigen/jacobi.c:7-85, float storage / double intermediates
    for (int i = 0; i < 3; ++i) {
      eig[i] = d[i];
      const float row[3] = {v[i][0], v[i][1], v[i][2]};
      const float len = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);
      const float il = (float)(1.0 / (double)len);
      for (int j = 0; j < 3; ++j) axes[4 * i + j] = row[j] * il;
      axes[4 * i + 3] = c[i];
    }
  }

  // sum of squared distances between the positions the two transforms give
  // the structure, from the axes alone
  float fast(const float *o1, const float *o2) const
  {
    float a[12], b[12];
    mat_times_mat(o1, axes, a);
    mat_times_mat(o2, axes, b);
    float s = 0.0f;
    for (int i = 0; i < 3; ++i)
      s += eig[i] * (1 - a[i] * b[i] - a[4 + i] * b[4 + i] - a[8 + i] * b[8 + i]);
    s *= 2.0f;
    const float dx = b[3] - a[3], dy = b[7] - a[7], dz = b[11] - a[11];
    s += npts * (dx * dx + dy * dy + dz * dz);
    if (s < 0.0f) s = 0.0f;
    return s;
  }
};

// the same quantity point by point
inline float slow_rot_ssd(const float *o1, const float *o2, const float *xyz, int npts)
{
  float s = 0.0f;
  for (int i = 0; i < npts; ++i) {
    float p1[3], p2[3];
    apply(o1, xyz + 3 * i, p1);
    apply(o2, xyz + 3 * i, p2);
    const float dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = p2[2] - p1[2];
    s += dx * dx + dy * dy + dz * dz;
  }
  return s;
}

// Which orientations supply the 2nd, 3rd, ... best alignment with n matches:
// candidates in file order, best_xf = transform of the best alignment.
// Appends (o_id -> {n, rank}) to owners.
inline void choose_alternates(int n, std::vector<Candidate> &file, const float *best_xf, int num_solutions,
                              float rmsd_tolerance, float min_delta_ssd, const OrientationMetric &metric,
                              std::multimap<int64_t, Owner> &owners)
{
  std::vector<Candidate> v;
  v.reserve(file.size());
  bool best_found = false;
  Candidate best;
  memset(&best, 0, sizeof best);
  for (const Candidate &c : file) {
    if (!best_found && same_matrix(best_xf, c.xf)) { best = c; best_found = true; }
    else v.push_back(c);
  }
  std::sort(v.begin(), v.end());
  v.insert(v.begin(), best);
  std::vector<const float *> chosen(1, v[0].xf);
  const float best_rmsd = v[0].rmsd;
  int alt_count = 1;
  for (size_t q = 1; q < v.size() && alt_count < num_solutions &&
                     (v[q].rmsd <= best_rmsd * rmsd_tolerance || rmsd_tolerance == -1.0f); ++q) {
    int r = 0;
    while (r < alt_count && metric.fast(v[q].xf, chosen[r]) >= min_delta_ssd) ++r;
    if (r == alt_count) {
      const Owner ow = {n, alt_count};
      owners.insert(std::make_pair(v[q].o_id, ow));
      chosen.push_back(v[q].xf);
      ++alt_count;
    }
  }
}

// Matched pairs two alignments share (both sorted by residue), the way
// NumMatchesInBothAssumingMonotonicity counts them
// (lib/pair_alignment/pair_alignment.cc:2529-2555).
inline int pairs_in_common(int len1, const int16_t *a1, int len2, const int16_t *a2)
{
  int count = 0, n1 = 0;
  for (int n2 = 0; n2 < len2; ++n2) {
    while (n1 < len1 && a1[2 * n1] < a2[2 * n2]) ++n1;
    if (n1 < len1 && a1[2 * n1] == a2[2 * n2] && a1[2 * n1 + 1] == a2[2 * n2 + 1]) ++count;
  }
  return count;
}

// Result tables of all ranks on the host: ssd[S][rn], xf[S][rn][12], pairs[S][rn][n_max][2].
struct RankTables {
  int S = 0, rn = 0, n_max = 0;
  std::vector<float> ssd, xf;
  std::vector<int16_t> pairs;

  void init(int num_solutions, int range_n, int nmax, float upper_bound)
  {
    S = num_solutions; rn = range_n; n_max = nmax;
    ssd.assign((size_t)S * rn, upper_bound);
    xf.assign((size_t)S * rn * 12, 0.0f);
    for (size_t q = 0; q < (size_t)S * rn; ++q) xf[12 * q] = xf[12 * q + 5] = xf[12 * q + 10] = 1.0f;
    pairs.assign((size_t)S * rn * n_max * 2, (int16_t)-1);
  }
  float &s(int r, int k) { return ssd[(size_t)r * rn + k]; }
  float *x(int r, int k) { return &xf[((size_t)r * rn + k) * 12]; }
  int16_t *p(int r, int k) { return &pairs[((size_t)r * rn + k) * n_max * 2]; }
};

// After refinement: per n, order the ranks by sum of squared distances, drop
// alternates that came too close to a better one or left the RMSD tolerance.
inline void check_consistency(RankTables &t, int n_min, int nbb, const float *xyzB, int num_points,
                              float min_3d_difference, float rmsd_tolerance, float upper_bound,
                              bool by_pairs, int max_pairs_common)
{
  static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const float min_delta = (min_3d_difference * min_3d_difference) * num_points;
  const size_t pr = (size_t)t.n_max * 2;
  for (int k = 0; k < t.rn; ++k) {
    const int n = n_min + k;
    std::vector<Ranked> v;
    for (int r = 0; r < t.S && t.s(r, k) != upper_bound; ++r) {
      Ranked e;
      e.ssd = t.s(r, k);
      memcpy(e.xf, t.x(r, k), 48);
      e.pairs.assign(t.p(r, k), t.p(r, k) + pr);
      v.push_back(e);
    }
    std::sort(v.begin(), v.end());
    for (size_t r = 0; r < v.size(); ++r) {
      t.s((int)r, k) = v[r].ssd;
      memcpy(t.x((int)r, k), v[r].xf, 48);
      memcpy(t.p((int)r, k), v[r].pairs.data(), pr * 2);
    }
    for (int r = 1; r < t.S && t.s(r, k) < upper_bound; ++r) {
      bool drop = false;
      int rp = 0;
      if (by_pairs) {          // AltSolutionsSufficientlyDissimilar, search_or_dyn3d.cc:1087-1094
        while (rp < r && pairs_in_common(n, t.p(rp, k), n, t.p(r, k)) <= max_pairs_common) ++rp;
      } else {
        while (rp < r && slow_rot_ssd(t.x(rp, k), t.x(r, k), xyzB, num_points) >= min_delta) ++rp;
      }
      if (rp < r) drop = true;
      if (rmsd_tolerance != -1.0f) {
        const float rmsd = sqrtf(t.s(r, k) / (float)(n * nbb));
        const float best = sqrtf(t.s(0, k) / (float)(n * nbb));
        if (rmsd > best * rmsd_tolerance) drop = true;
      }
      if (drop) {
        for (int rc = r; rc + 1 < t.S; ++rc) {
          t.s(rc, k) = t.s(rc + 1, k);
          memcpy(t.p(rc, k), t.p(rc + 1, k), pr * 2);
          memcpy(t.x(rc, k), t.x(rc + 1, k), 48);
        }
        t.s(t.S - 1, k) = upper_bound;
        memcpy(t.x(t.S - 1, k), ident, 48);
        --r;
      }
    }
  }
}

}  // namespace alt
