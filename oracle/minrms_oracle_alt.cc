/* minrms_oracle_alt.cc -- CPU restatement of the reference's alternate
 * solutions by orientation (`-alt-o count,min_distance[,rmsd_ratio]`,
 * SearchOr::Settings::ALT_METHOD_3D).  TEST INFRASTRUCTURE ONLY: linked into
 * liboracle.so beside minrms_oracle.c, used by tests/ to check the HIP engine.
 * Parity PINNED: tests/test_oracle_golden.py compares it bit for bit with the
 * tables the compiled reference produced (the "alt" cases under tests/golden).
 *
 * C++ rather than C because the reference orders its candidates with
 * std::sort (bin/minrms/search_or_dyn3d.cc:756,1333,1621), which is not
 * stable: candidates of equal RMSD end up in the order libstdc++'s introsort
 * leaves them in, and the same call on the same sequence is the only
 * practical way to get that order.
 *
 * Follows, paths relative to the reference tree:
 *   bin/minrms/search_or_dyn3d.cc:283-325    Search() with num_solutions > 1
 *   :477-542   KeepTrackOfAltSolutions       :626-921  ReadAltSolutionFile(s)
 *   :928-962   SufficientlyDifferent...      :1045-1096 AltSolutionsSufficientlyDissimilar
 *   :1099-1247 CheckAlternatesForConsistency :1252-1452 CalcAltSolutionsFromAltSolutionTable
 *   :1585-1640 SortAltList                   :2261-2658 Refine() over the ranks
 *   :367-414 CalcAltSolutionsUsing_ALT_METHOD_PAIRS, :965-1040 GetOrientationsOfLowRmsdFromAltTable,
 *   :1969-2250 MultiPass; lib/pair_alignment/pair_alignment.cc:2529-2555
 *   lib/or_gen/or_gen.cc:93-110,158-165      SumSqdDistBetweenOrientations
 *   lib/fast_rot_metric/fast_rot_metric.h:96-132, fast_rot_metric.cc:13-28
 *   lib/fast_rot_metric/principle_axis.cc:30-160
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "minrms_oracle.h"

namespace {

struct Candidate {                       /* OrientationUsageInstance, search_or_dyn3d.h:473-492 */
  int64_t o_id;
  float rmsd;
  float xf[12];
  bool operator<(const Candidate &b) const { return rmsd < b.rmsd; }
};

struct Solution {                        /* SolutionInfo, search_or_dyn3d.h:500-507 */
  float ssd;
  float xf[12];
  std::vector<int16_t> pairs;
  bool operator<(const Solution &b) const { return ssd < b.ssd; }
};

struct Owner {                           /* OrientationOwner, search_or_dyn3d.h:527-534 */
  int n, r;
  bool operator<(const Owner &b) const { return n < b.n || (n == b.n && r < b.r); }
};

bool same_matrix(const float *a, const float *b)      /* vect_3d.h:403-411 */
{
  for (int q = 0; q < 12; ++q)
    if (a[q] != b[q]) return false;
  return true;
}

/* vect_3d.h:199-224 */
void mat_times_mat(const float *A, const float *B, float *dest)
{
  float t[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j)
      t[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    t[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
  }
  memcpy(dest, t, sizeof t);
}

}  // namespace

extern "C" {

/* FindPrincipleAxis, lib/fast_rot_metric/principle_axis.cc:30-160 */
void orc_principal_axes(int npts, const float *xyz, float eig[3], float axes[12])
{
  float c[3] = {0.0f, 0.0f, 0.0f};                     /* FindCentroid, vect_3d.cc:73-94 */
  const float inv = 1.0f / (float)npts;
  for (int a = 0; a < npts; ++a)
    for (int k = 0; k < 3; ++k) c[k] += xyz[3 * a + k];
  for (int k = 0; k < 3; ++k) c[k] *= inv;
  float q[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = 0; a < npts; ++a) {
    const float r[3] = {xyz[3 * a] - c[0], xyz[3 * a + 1] - c[1], xyz[3 * a + 2] - c[2]};
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) q[3 * i + j] += r[i] * r[j];
  }
  float d[3], v[9];
  orc_jacobi(3, q, d, v);
  for (int i = 0; i < 3; ++i) {
    eig[i] = d[i];
    float row[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
    const float len = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);   /* Normalize3 */
    const float il = (float)(1.0 / (double)len);
    for (int j = 0; j < 3; ++j) axes[4 * i + j] = row[j] * il;
    axes[4 * i + 3] = c[i];
  }
}

/* OrientationGenerator::SumSqdDistBetweenOrientations + FastRotSumSqdDist */
float orc_fast_rot_ssd(const float o1[12], const float o2[12], const float eig[3],
                       const float axes[12], int npts)
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

/* SlowRotSumSqdDist, lib/fast_rot_metric/fast_rot_metric.cc:13-28 */
float orc_slow_rot_ssd(const float o1[12], const float o2[12], const float *xyz, int npts)
{
  float s = 0.0f;
  float *p1 = (float *)malloc(12 * (size_t)npts), *p2 = (float *)malloc(12 * (size_t)npts);
  orc_apply_transform(o1, xyz, npts, p1);
  orc_apply_transform(o2, xyz, npts, p2);
  for (int i = 0; i < npts; ++i) {
    const float dx = p2[3 * i] - p1[3 * i], dy = p2[3 * i + 1] - p1[3 * i + 1], dz = p2[3 * i + 2] - p1[3 * i + 2];
    s += dx * dx + dy * dy + dz * dz;
  }
  free(p1); free(p2);
  return s;
}

static void init_rank(const orc_cfg *cfg, int rn, float *ssd, float *xform, int16_t *pairs)
{
  static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  for (int k = 0; k < rn; ++k) {
    ssd[k] = ORC_UPPER_BOUND;
    memcpy(xform + 12 * k, ident, sizeof ident);
  }
  memset(pairs, 0xFF, (size_t)rn * cfg->n_max * 2 * sizeof(int16_t));
}

static bool max_rmsd_exceeded(const orc_cfg *cfg, float ssd, int n)
{
  return cfg->max_rmsd != -1.0f && ssd > (float)(n * cfg->nbb) * (cfg->max_rmsd * cfg->max_rmsd);
}

/* Search() with num_solutions > 1 and ALT_METHOD_3D.  Tables are
 * [num_solutions][range_n](...); *io_clip as in orc_search_filtered.
 * Returns 0, or -1 if no n has a solution (search_or_dyn3d.cc:1929-1941). */
int orc_search_alt(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA, const float *xyzB,
                   const float *og, int64_t n_og, float *ssd, float *xform, int16_t *pairs,
                   float *io_clip)
{
  const int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb, n_max = cfg->n_max, n_min = cfg->n_min;
  const int rn = n_max - n_min + 1, S = alt->num_solutions;
  const size_t plane = (size_t)N1 * N2;
  const size_t pr_stride = (size_t)n_max * 2;
  for (int r = 0; r < S; ++r)
    init_rank(cfg, rn, ssd + (size_t)r * rn, xform + (size_t)r * rn * 12, pairs + (size_t)r * rn * pr_stride);

  std::vector<uint8_t> bp((size_t)n_max * plane);
  std::vector<float> cost(n_max), rot((size_t)N2 * nbb * 3);
  std::vector<int16_t> tmp(pr_stride);
  std::vector<std::vector<Candidate>> files(rn);        /* the ".tmp" files, one per n */
  float clip = io_clip ? *io_clip : orc_initial_clip(cfg);
  const float tol = alt->rmsd_tolerance;

  for (int64_t o = 0; o < n_og; ++o) {                  /* FirstPass, :1676-1925 */
    orc_apply_transform(og + 12 * o, xyzB, N2 * nbb, rot.data());
    if (cfg->of_enable && !orc_good_orientation(cfg, xyzA, rot.data(), &clip)) continue;
    orc_dp_clip(cfg, xyzA, rot.data(), clip, cost.data(), bp.data());
    for (int n = 2; n <= n_max; ++n) {
      if (cost[n - 1] >= ORC_UPPER_BOUND) break;
      if (n < n_min) continue;
      const int k = n - n_min;
      orc_traceback(cfg, bp.data(), n, tmp.data());
      float s = cost[n - 1], xf[12];
      memcpy(xf, og + 12 * o, sizeof xf);
      if (cfg->refine_while_searching) orc_calc_min_rmsd(cfg, xyzA, xyzB, n, tmp.data(), xf, &s);
      if (max_rmsd_exceeded(cfg, s, n)) break;
      if (s < ssd[k]) {
        ssd[k] = s;
        memcpy(xform + 12 * k, xf, sizeof xf);
        memcpy(pairs + k * pr_stride, tmp.data(), (size_t)n * 4);
      }
      /* KeepTrackOfAltSolutions, :477-542: after the update, against the best so far */
      if (tol == -1.0f || s <= ssd[k] * (tol * tol)) {
        Candidate c;
        c.o_id = o;
        c.rmsd = sqrtf(s / (float)(n * nbb));
        memcpy(c.xf, xf, sizeof xf);
        files[k].push_back(c);
      }
    }
  }
  if (io_clip) *io_clip = clip;
  bool any = false;
  for (int k = 0; k < rn; ++k) any = any || ssd[k] != ORC_UPPER_BOUND;
  if (!any) return -1;

  /* ReadAltSolutionFiles, :626-921 */
  float eig[3], axes[12];
  orc_principal_axes(N2 * nbb, xyzB, eig, axes);
  const float min_delta = (alt->min_3d_difference * alt->min_3d_difference) * N2 * nbb;   /* :935-938 */
  std::multimap<int64_t, Owner> owners;
  for (int n = n_min; n <= n_max; ++n) {
    const int k = n - n_min;
    if (ssd[k] == ORC_UPPER_BOUND) continue;
    std::vector<Candidate> v;
    v.reserve(files[k].size());
    bool best_found = false;
    Candidate best;
    for (const Candidate &c : files[k]) {
      if (!best_found && same_matrix(xform + 12 * k, c.xf)) { best = c; best_found = true; }
      else v.push_back(c);
    }
    std::sort(v.begin(), v.end());
    v.insert(v.begin(), best);
    std::vector<const float *> chosen(1, v[0].xf);
    const float best_rmsd = v[0].rmsd;
    int alt_count = 1;
    for (size_t q = 1; q < v.size() && alt_count < S && (v[q].rmsd <= best_rmsd * tol || tol == -1.0f); ++q) {
      int r = 0;
      while (r < alt_count && orc_fast_rot_ssd(v[q].xf, chosen[r], eig, axes, N2 * nbb) >= min_delta) ++r;
      if (r == alt_count) {
        Owner ow = {n, alt_count};
        owners.insert(std::make_pair(v[q].o_id, ow));
        chosen.push_back(v[q].xf);
        ++alt_count;
      }
    }
  }

  /* CalcAltSolutionsFromAltSolutionTable, :1252-1452 */
  for (auto it = owners.begin(); it != owners.end();) {
    const int64_t o = it->first;
    std::vector<Owner> mine;
    for (; it != owners.end() && it->first == o; ++it) mine.push_back(it->second);
    std::sort(mine.begin(), mine.end());
    orc_apply_transform(og + 12 * o, xyzB, N2 * nbb, rot.data());
    orc_dp_clip(cfg, xyzA, rot.data(), clip, cost.data(), bp.data());
    for (const Owner &ow : mine) {
      const int k = ow.n - n_min;
      float *s_out = ssd + (size_t)ow.r * rn + k;
      float *x_out = xform + ((size_t)ow.r * rn + k) * 12;
      int16_t *p_out = pairs + ((size_t)ow.r * rn + k) * pr_stride;
      orc_traceback(cfg, bp.data(), ow.n, p_out);
      if (cfg->refine_while_searching) {
        orc_calc_min_rmsd(cfg, xyzA, xyzB, ow.n, p_out, x_out, s_out);
      } else {
        *s_out = cost[ow.n - 1];
        memcpy(x_out, og + 12 * o, 48);
      }
    }
  }
  return 0;
}

/* NumMatchesInBothAssumingMonotonicity, lib/pair_alignment/pair_alignment.cc:2529-2555 */
int orc_pairs_in_common(int len1, const int16_t *a1, int len2, const int16_t *a2)
{
  int count = 0, n1 = 0;
  for (int n2 = 0; n2 < len2; ++n2) {
    while (n1 < len1 && a1[2 * n1] < a2[2 * n2]) ++n1;
    if (n1 < len1 && a1[2 * n1] == a2[2 * n2] && a1[2 * n1 + 1] == a2[2 * n2 + 1]) ++count;
  }
  return count;
}

int orc_search_alt_pairs(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA, const float *xyzB,
                         const float *og, int64_t n_og, float *ssd, float *xform, int16_t *pairs,
                         float *io_clip)
{
  const int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb, n_max = cfg->n_max, n_min = cfg->n_min;
  const int rn = n_max - n_min + 1, S = alt->num_solutions;
  const size_t plane = (size_t)N1 * N2, pr_stride = (size_t)n_max * 2;
  for (int r = 0; r < S; ++r)
    init_rank(cfg, rn, ssd + (size_t)r * rn, xform + (size_t)r * rn * 12, pairs + (size_t)r * rn * pr_stride);
  std::vector<uint8_t> bp((size_t)n_max * plane);
  std::vector<float> cost(n_max), rot((size_t)N2 * nbb * 3);
  std::vector<int16_t> tmp(pr_stride);
  struct Rec { int64_t o_id; float rmsd; };
  std::vector<std::vector<Rec>> files(rn);
  float clip = io_clip ? *io_clip : orc_initial_clip(cfg);
  const float tol = alt->rmsd_tolerance;

  /* one pass over a set of orientations for rank r (r = 0: FirstPass) */
  auto pass = [&](int r, const std::vector<int64_t> &set) {
    float *r_ssd = ssd + (size_t)r * rn, *r_xf = xform + (size_t)r * rn * 12;
    int16_t *r_pr = pairs + (size_t)r * rn * pr_stride;
    for (int64_t o : set) {
      orc_apply_transform(og + 12 * o, xyzB, N2 * nbb, rot.data());
      if (cfg->of_enable && !orc_good_orientation(cfg, xyzA, rot.data(), &clip)) continue;
      orc_dp_clip(cfg, xyzA, rot.data(), clip, cost.data(), bp.data());
      for (int n = 2; n <= n_max; ++n) {
        if (cost[n - 1] >= ORC_UPPER_BOUND) break;
        if (n < n_min) continue;
        const int k = n - n_min;
        orc_traceback(cfg, bp.data(), n, tmp.data());
        float s = cost[n - 1], xf[12];
        memcpy(xf, og + 12 * o, sizeof xf);
        if (cfg->refine_while_searching) orc_calc_min_rmsd(cfg, xyzA, xyzB, n, tmp.data(), xf, &s);
        if (max_rmsd_exceeded(cfg, s, n)) break;
        if (s < r_ssd[k]) {
          int ra = 0;                                    /* :2107-2143 */
          while (ra < r && orc_pairs_in_common(n, tmp.data(), n, pairs + ((size_t)ra * rn + k) * pr_stride)
                               <= alt->max_pairs_common) ++ra;
          if (ra == r) {
            r_ssd[k] = s;
            memcpy(r_xf + 12 * k, xf, sizeof xf);
            memcpy(r_pr + k * pr_stride, tmp.data(), (size_t)n * 4);
          }
        }
        if (r == 0 && tol != -1.0f && s <= r_ssd[k] * (tol * tol)) {   /* :493-541, table only with a tolerance */
          Rec c = {o, sqrtf(s / (float)(n * nbb))};
          files[k].push_back(c);
        }
      }
    }
  };
  std::vector<int64_t> all((size_t)n_og);
  for (int64_t o = 0; o < n_og; ++o) all[(size_t)o] = o;
  pass(0, all);
  bool any = false;
  for (int k = 0; k < rn; ++k) any = any || ssd[k] != ORC_UPPER_BOUND;
  if (!any) { if (io_clip) *io_clip = clip; return -1; }

  std::vector<int64_t> use = all;                          /* :385-399 */
  if (tol != -1.0f) {
    std::vector<char> want((size_t)n_og, 0);
    for (int k = 0; k < rn; ++k) {
      if (ssd[k] == ORC_UPPER_BOUND) continue;
      const float best_rmsd = sqrtf(ssd[k] / (float)((n_min + k) * nbb));
      const float tolerable = tol * best_rmsd;
      for (const Rec &c : files[k])
        if (tolerable >= c.rmsd) want[(size_t)c.o_id] = 1;
    }
    use.clear();
    for (int64_t o = 0; o < n_og; ++o)
      if (want[(size_t)o]) use.push_back(o);
  }
  for (int r = 1; r < S; ++r) pass(r, use);
  if (io_clip) *io_clip = clip;
  return 0;
}

/* CheckAlternatesForConsistency, :1099-1247 (with SortAltList, :1585-1640) */
static void check_alternates(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzB, float *ssd,
                             float *xform, int16_t *pairs)
{
  const int n_max = cfg->n_max, n_min = cfg->n_min, rn = n_max - n_min + 1, S = alt->num_solutions;
  const size_t pr_stride = (size_t)n_max * 2;
  const int num_points = cfg->N2 * cfg->nbb;
  const float min_delta = (alt->min_3d_difference * alt->min_3d_difference) * num_points;
  static const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  auto S_ssd = [&](int r, int k) -> float & { return ssd[(size_t)r * rn + k]; };
  auto S_xf = [&](int r, int k) { return xform + ((size_t)r * rn + k) * 12; };
  auto S_pr = [&](int r, int k) { return pairs + ((size_t)r * rn + k) * pr_stride; };
  for (int n = n_min; n <= n_max; ++n) {
    const int k = n - n_min;
    std::vector<Solution> v;                            /* SortAltList */
    for (int r = 0; r < S && S_ssd(r, k) != ORC_UPPER_BOUND; ++r) {
      Solution s;
      s.ssd = S_ssd(r, k);
      memcpy(s.xf, S_xf(r, k), 48);
      s.pairs.assign(S_pr(r, k), S_pr(r, k) + pr_stride);
      v.push_back(s);
    }
    std::sort(v.begin(), v.end());
    for (size_t r = 0; r < v.size(); ++r) {
      S_ssd((int)r, k) = v[r].ssd;
      memcpy(S_xf((int)r, k), v[r].xf, 48);
      memcpy(S_pr((int)r, k), v[r].pairs.data(), pr_stride * 2);
    }
    for (int r = 1; r < S && S_ssd(r, k) < ORC_UPPER_BOUND; ++r) {
      bool del = false;
      int rp = 0;
      if (alt->by_pairs) {                               /* :1087-1094 */
        while (rp < r && orc_pairs_in_common(n, S_pr(rp, k), n, S_pr(r, k)) <= alt->max_pairs_common) ++rp;
      } else {
        while (rp < r && orc_slow_rot_ssd(S_xf(rp, k), S_xf(r, k), xyzB, num_points) >= min_delta) ++rp;
      }
      if (rp < r) del = true;
      if (alt->rmsd_tolerance != -1.0f) {
        const float rmsd = sqrtf(S_ssd(r, k) / (float)(n * cfg->nbb));
        const float best = sqrtf(S_ssd(0, k) / (float)(n * cfg->nbb));
        if (rmsd > best * alt->rmsd_tolerance) del = true;
      }
      if (del) {
        for (int rc = r; rc + 1 < S; ++rc) {
          S_ssd(rc, k) = S_ssd(rc + 1, k);
          memcpy(S_pr(rc, k), S_pr(rc + 1, k), pr_stride * 2);
          memcpy(S_xf(rc, k), S_xf(rc + 1, k), 48);
        }
        S_ssd(S - 1, k) = ORC_UPPER_BOUND;
        memcpy(S_xf(S - 1, k), ident, 48);
        --r;
      }
    }
  }
}

/* Refine() over the ranks, :2261-2658.  Returns the iteration count the
 * reference prints (that of the last rank refined); *num_new = updates. */
int64_t orc_refine_alt(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA, const float *xyzB,
                       float *ssd, float *xform, int16_t *pairs, int64_t max_iters, int64_t *num_new,
                       float clip)
{
  const int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb, n_max = cfg->n_max, n_min = cfg->n_min;
  const int rn = n_max - n_min + 1, S = alt->num_solutions;
  const size_t plane = (size_t)N1 * N2, pr_stride = (size_t)n_max * 2;
  std::vector<uint8_t> bp((size_t)n_max * plane);
  std::vector<float> cost(n_max), rot((size_t)N2 * nbb * 3);
  std::vector<int16_t> tmp(pr_stride), old(pr_stride);
  std::vector<unsigned char> need(rn);
  int64_t updates = 0, it = 1;
  const int ranks = alt->refine_all ? S : 1;
  for (int r = 0; r < ranks; ++r) {
    float *r_ssd = ssd + (size_t)r * rn, *r_xf = xform + (size_t)r * rn * 12;
    int16_t *r_pr = pairs + (size_t)r * rn * pr_stride;
    for (int k = 0; k < rn; ++k) need[k] = r_ssd[k] < ORC_UPPER_BOUND;
    for (it = 1; it <= max_iters || max_iters == -1; ++it) {
      int in_set = 0;
      for (int k = 0; k < rn; ++k) in_set += need[k];
      if (in_set == 0) break;
      int half = (in_set + 1) / 2, k = -1, seen = 0;
      do { ++k; seen += need[k]; } while (seen < half);
      const int n = n_min + k;
      float xf[12];
      orc_calc_min_rmsd(cfg, xyzA, xyzB, n, r_pr + k * pr_stride, xf, NULL);
      orc_apply_transform(xf, xyzB, N2 * nbb, rot.data());
      memcpy(old.data(), r_pr + k * pr_stride, (size_t)n * 4);
      orc_dp_clip(cfg, xyzA, rot.data(), clip, cost.data(), bp.data());
      for (int m = 2; m <= n_max; ++m) {
        if (cost[m - 1] >= ORC_UPPER_BOUND) break;
        if (m < n_min) continue;
        const int km = m - n_min;
        orc_traceback(cfg, bp.data(), m, tmp.data());
        float s = cost[m - 1], mxf[12];
        memcpy(mxf, xf, sizeof mxf);
        if (cfg->refine_while_searching) orc_calc_min_rmsd(cfg, xyzA, xyzB, m, tmp.data(), mxf, &s);
        if (max_rmsd_exceeded(cfg, s, m)) break;
        if (s < r_ssd[km] && (r == 0 || m == n)) {      /* :2515-2614 */
          r_ssd[km] = s;
          memcpy(r_xf + 12 * km, mxf, sizeof mxf);
          memcpy(r_pr + km * pr_stride, tmp.data(), (size_t)m * 4);
          if (r == 0 && m != n) need[km] = 1;
          ++updates;
        }
      }
      need[k] = memcmp(old.data(), r_pr + k * pr_stride, (size_t)n * 4) != 0;
    }
  }
  if (num_new) *num_new = updates;
  if (S > 1) check_alternates(cfg, alt, xyzB, ssd, xform, pairs);
  return it;
}

}  /* extern "C" */
