/* minrms_oracle.c -- CPU restatement of the MINRMS RMSD-vs-N search.
 *
 * TEST INFRASTRUCTURE ONLY (see minrms_oracle.h).  Plain C99, scalar, one
 * thread.  Compile with -O2 -ffp-contract=off and without -ffast-math: the
 * float operation order below is the contract.
 *
 * Every function cites the reference file:line whose behaviour it restates.
 * The eigen-solver is our own implementation of Rutishauser's cyclic Jacobi
 * method (H. Rutishauser, "The Jacobi method for real symmetric matrices",
 * Numer. Math. 9, 1-10, 1966) arranged so that its rounding sequence is the one
 * the reference's solver (lib/eigen/jacobi.c:7-85) produces.
 */
#include "minrms_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* Eigen-solver                                                             */
/* ------------------------------------------------------------------------ */

/* |x| + g evaluated in double and then narrowed, compared with |x|: the
 * "g is negligible beside x" test (jacobi.c:41-42,46, where fabs() is the
 * double libm function because that file is compiled as C). */
static int negligible_beside(float x, float g)
{
  double ax = fabs((double)x);
  return (float)(ax + (double)g) == (float)ax;
}

/* Plane rotation of the element pair (*p,*q) (jacobi.c:4-5). */
static void rot_pair(float *p, float *q, float s, float tau)
{
  float g = *p, h = *q;
  *p = g - s * (h + g * tau);
  *q = h + s * (g - h * tau);
}

int orc_jacobi(int n, float *a, float *d, float *v)
{
  float b[8], z[8];
  if (n > 8) return -1;
  for (int r = 0; r < n; ++r) {
    for (int c = 0; c < n; ++c) v[r * n + c] = (r == c) ? 1.0f : 0.0f;
    b[r] = d[r] = a[r * n + r];
    z[r] = 0.0f;
  }
  for (int sweep = 1; sweep <= 50; ++sweep) {
    /* off-diagonal magnitude, accumulated through double (jacobi.c:24-28) */
    float sm = 0.0f;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q)
        sm = (float)((double)sm + fabs((double)a[p * n + q]));
    if (sm == 0.0f) return sweep - 1;
    float tresh = (sweep < 4) ? (float)(0.2 * (double)sm / (double)(n * n)) : 0.0f;

    for (int p = 0; p < n - 1; ++p) {
      for (int q = p + 1; q < n; ++q) {
        float apq = a[p * n + q];
        float g = (float)(100.0 * fabs((double)apq));
        if (sweep > 4 && negligible_beside(d[p], g) && negligible_beside(d[q], g)) {
          a[p * n + q] = 0.0f;
          continue;
        }
        if (!(fabs((double)apq) > (double)tresh)) continue;

        float h = d[q] - d[p];
        float t;
        if (negligible_beside(h, g)) {
          t = apq / h;
        } else {
          float theta = (float)(0.5 * (double)h / (double)apq);
          float th2 = theta * theta;
          t = (float)(1.0 / (fabs((double)theta) + sqrt(1.0 + (double)th2)));
          if (theta < 0.0f) t = -t;
        }
        float one_tt = 1.0f + t * t;
        float c = (float)(1.0 / sqrt((double)one_tt));
        float s = t * c;
        float tau = (float)((double)s / (1.0 + (double)c));
        h = t * apq;
        z[p] -= h; z[q] += h;
        d[p] -= h; d[q] += h;
        a[p * n + q] = 0.0f;
        for (int j = 0; j < p; ++j)      rot_pair(&a[j * n + p], &a[j * n + q], s, tau);
        for (int j = p + 1; j < q; ++j)  rot_pair(&a[p * n + j], &a[j * n + q], s, tau);
        for (int j = q + 1; j < n; ++j)  rot_pair(&a[p * n + j], &a[q * n + j], s, tau);
        for (int j = 0; j < n; ++j)      rot_pair(&v[j * n + p], &v[j * n + q], s, tau);
      }
    }
    for (int r = 0; r < n; ++r) {
      b[r] += z[r];
      d[r] = b[r];
      z[r] = 0.0f;
    }
  }
  return -1;
}

/* ------------------------------------------------------------------------ */
/* Least-squares superposition (Diamond 1988 quaternion method)             */
/* ------------------------------------------------------------------------ */

/* lib/vect_3d/vect_3d.cc:73-94: plain running sum, then times 1/n */
static void centroid(int npts, const float *p, float c[3])
{
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  float inv = 1.0f / (float)npts;
  for (int a = 0; a < npts; ++a) {
    sx += p[3 * a + 0]; sy += p[3 * a + 1]; sz += p[3 * a + 2];
  }
  c[0] = sx * inv; c[1] = sy * inv; c[2] = sz * inv;
}

float orc_superimpose(int npts, const float *fixed, const float *moveable,
                      float xform[12])
{
  float cf[3], cm[3];
  centroid(npts, fixed, cf);      /* superimpose.cc:35-37 */
  centroid(npts, moveable, cm);

  /* correlation matrix of the centred sets (superimpose.cc:49-68);
   * M[i][j] = sum_a mov[a][i] * fix[a][j], accumulated in point order */
  float M[3][3] = {{0}};
  float E0 = 0.0f;
  for (int a = 0; a < npts; ++a) {
    float f[3], m[3];
    for (int k = 0; k < 3; ++k) {
      f[k] = fixed[3 * a + k] - cf[k];
      m[k] = moveable[3 * a + k] - cm[k];
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        M[i][j] += m[i] * f[j];
  }
  /* residual before rotation (superimpose.cc:258-263); a separate pass in the
   * reference, but each term depends only on the centred coordinates */
  for (int a = 0; a < npts; ++a)
    for (int k = 0; k < 3; ++k) {
      float df = (fixed[3 * a + k] - cf[k]) - (moveable[3 * a + k] - cm[k]);
      E0 += df * df;
    }

  float tr = M[0][0] + M[1][1] + M[2][2];           /* vect_3d.h:359 */
  float P[16];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)                      /* superimpose.cc:78-82 */
      P[i * 4 + j] = (M[i][j] + M[j][i]) - ((i == j) ? (2 * tr) : 0.0f);
  float V[3] = { M[1][2] - M[2][1], M[2][0] - M[0][2], M[0][1] - M[1][0] };
  for (int k = 0; k < 3; ++k) P[k * 4 + 3] = P[12 + k] = V[k];
  P[15] = 0.0f;

  float d[4], ev[16];
  orc_jacobi(4, P, d, ev);

  int big = 0;                                       /* superimpose.cc:187-196 */
  for (int k = 1; k < 4; ++k) if (d[k] > d[big]) big = k;
  float lam = d[big];
  float q[4] = { ev[0 * 4 + big], ev[1 * 4 + big], ev[2 * 4 + big], ev[3 * 4 + big] };

  /* vect_3d.cc:61-71 */
  float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  float inv = (float)(1.0 / (double)len);
  for (int k = 0; k < 4; ++k) q[k] *= inv;

  /* vect_3d.cc:97-117 */
  float R[3][3];
  R[0][0] = +(q[0]*q[0]) - (q[1]*q[1]) - (q[2]*q[2]) + (q[3]*q[3]);
  R[1][1] = -(q[0]*q[0]) + (q[1]*q[1]) - (q[2]*q[2]) + (q[3]*q[3]);
  R[2][2] = -(q[0]*q[0]) - (q[1]*q[1]) + (q[2]*q[2]) + (q[3]*q[3]);
  R[0][1] = 2 * (q[0]*q[1] - q[2]*q[3]);
  R[1][0] = 2 * (q[0]*q[1] + q[2]*q[3]);
  R[1][2] = 2 * (q[1]*q[2] - q[0]*q[3]);
  R[2][1] = 2 * (q[1]*q[2] + q[0]*q[3]);
  R[0][2] = 2 * (q[0]*q[2] + q[1]*q[3]);
  R[2][0] = 2 * (q[0]*q[2] - q[1]*q[3]);

  for (int i = 0; i < 3; ++i) {                      /* superimpose.cc:245-255 */
    float rc = R[i][0] * cm[0] + R[i][1] * cm[1] + R[i][2] * cm[2];
    xform[4 * i + 0] = R[i][0];
    xform[4 * i + 1] = R[i][1];
    xform[4 * i + 2] = R[i][2];
    xform[4 * i + 3] = cf[i] - rc;
  }

  float min_ssd = E0 - 2 * lam;                      /* superimpose.cc:266-277 */
  if (min_ssd < 0.0f) min_ssd = 0.0f;
  return sqrtf(min_ssd / (float)npts);               /* superimpose.cc:285 */
}

void orc_apply_transform(const float xf[12], const float *xyz, int npts, float *out)
{
  for (int a = 0; a < npts; ++a) {
    float x = xyz[3 * a], y = xyz[3 * a + 1], z = xyz[3 * a + 2];
    for (int r = 0; r < 3; ++r)
      out[3 * a + r] = xf[4 * r] * x + xf[4 * r + 1] * y + xf[4 * r + 2] * z + xf[4 * r + 3];
  }
}

/* ------------------------------------------------------------------------ */
/* Orientation generation                                                   */
/* ------------------------------------------------------------------------ */

/* glibc rand() (TYPE_3 additive feedback, degree 31, separation 3) in its
 * never-seeded state, i.e. srand(1). */
typedef struct { int32_t r[34]; int f, b; } glibc_rand;

static void glibc_rand_init(glibc_rand *g)
{
  int32_t st[31];
  int32_t word = 1;
  st[0] = word;
  for (int i = 1; i < 31; ++i) {
    int64_t hi = word / 127773, lo = word % 127773;
    int64_t w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    st[i] = word;
  }
  memcpy(g->r, st, sizeof st);
  g->f = 3; g->b = 0;
  for (int i = 0; i < 310; ++i) {
    g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
    g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
  }
}

static int32_t glibc_rand_next(glibc_rand *g)
{
  uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
  g->r[g->f] = (int32_t)v;
  g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
  return (int32_t)(v >> 1);
}

void orc_shuffle_perm(int64_t count, int64_t *perm)
{
  glibc_rand g;
  glibc_rand_init(&g);
  for (int64_t k = 0; k < count; ++k) perm[k] = k;
  /* libstdc++ bits/stl_algo.h random_shuffle(first,last):
   *   for i in 1..count-1: j = rand() % (i+1); swap(v[i], v[j]) */
  for (int64_t i = 1; i < count; ++i) {
    int64_t j = (int64_t)glibc_rand_next(&g) % (i + 1);
    if (i != j) { int64_t t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  }
}

int64_t orc_gen_orientations(const orc_cfg *cfg, const float *xyzA,
                             const float *xyzB, int L, float cutoff_rmsd,
                             int n_int, const int32_t *ip, float *out, int64_t cap)
{
  int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb;
  int w1 = N1 - cfg->n_min + 1, w2 = N2 - cfg->n_min + 1;   /* or_gen.cc:253-254 */
  unsigned char *mask = calloc((size_t)N1 * N2, 1);
  for (int p = 0; p < n_int; ++p) {
    int s1 = ip[4 * p], e1 = ip[4 * p + 1], s2 = ip[4 * p + 2], e2 = ip[4 * p + 3];
    if (s1 > e1 - L + 1 || s2 > e2 - L + 1) continue;          /* or_gen.cc:404-429 */
    for (int I = s1; I <= e1 - L + 1; ++I) {
      int jlo = I - (w1 - 1); if (jlo < s2) jlo = s2;          /* or_gen.cc:451-464 */
      int jhi = I + (w2 - 1); if (jhi > e2 - L + 1) jhi = e2 - L + 1;
      for (int J = jlo; J <= jhi; ++J) mask[(size_t)I * N2 + J] = 1;
    }
  }
  int64_t count = 0;
  for (int I = 0; I < N1; ++I)
    for (int J = 0; J < N2; ++J) {
      if (!mask[(size_t)I * N2 + J]) continue;
      float xf[12];
      float rmsd = orc_superimpose(L * nbb, xyzA + 3 * (size_t)I * nbb,
                                   xyzB + 3 * (size_t)J * nbb, xf);
      if (cutoff_rmsd >= 0.0f && !(rmsd <= cutoff_rmsd)) continue; /* or_gen.cc:537-541 */
      if (count < cap) memcpy(out + 12 * count, xf, sizeof xf);
      ++count;
    }
  free(mask);
  return count;
}

/* ------------------------------------------------------------------------ */
/* The 3-index dynamic program                                              */
/* ------------------------------------------------------------------------ */

static float clip_metric(const orc_cfg *cfg)
{                                                   /* distance_metric.cc:76-87 */
  if (cfg->max_dist == -1.0f) return ORC_UPPER_BOUND;
  return (cfg->max_dist * cfg->max_dist) * (float)cfg->nbb;
}

/* distance_metric.h:104-130 */
static float pair_cost(const float *a, const float *b, int nbb, float clip)
{
  float out = 0.0f;
  for (int q = 0; q < nbb; ++q) {
    float dx = a[3 * q] - b[3 * q], dy = a[3 * q + 1] - b[3 * q + 1], dz = a[3 * q + 2] - b[3 * q + 2];
    out += dx * dx + dy * dy + dz * dz;
  }
  return (out > clip) ? ORC_UPPER_BOUND : out;
}

static void dp_with_clip(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot,
                         float clip, float *cost, uint8_t *bp)
{
  int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb;
  int wA = N1 - cfg->n_min + 1, wB = N2 - cfg->n_min + 1;   /* dyn3d.cc:472-473 */
  size_t plane = (size_t)N1 * N2;
  float *C = malloc(plane * sizeof(float));
  float *lay = malloc(2 * plane * sizeof(float));
  for (size_t k = 0; k < plane; ++k) C[k] = NAN;
  for (size_t k = 0; k < 2 * plane; ++k) lay[k] = NAN;
  if (bp) memset(bp, 0xFF, (size_t)cfg->n_max * plane);

  /* dyn3d.cc:664-701: only the diagonal band is filled */
  for (int i = 0; i < N1; ++i) {
    int jlo = (i < wA) ? 0 : i - wA + 1;
    int jhi = (i + wB < N2) ? i + wB : N2;
    for (int j = jlo; j < jhi; ++j)
      C[(size_t)i * N2 + j] = pair_cost(xyzA + 3 * (size_t)i * nbb,
                                        xyzB_rot + 3 * (size_t)j * nbb, nbb, clip);
  }

  for (int n = 1; n <= cfg->n_max; ++n) {
    float *cur = lay + (size_t)(n & 1) * plane;
    const float *prv = lay + (size_t)((n - 1) & 1) * plane;
    int imax = (n + wA - 1 < N1) ? n + wA - 1 : N1;        /* dyn3d.cc:188-189 */
    int jmax = (n + wB - 1 < N2) ? n + wB - 1 : N2;
    for (int i = n; i <= imax; ++i)
      for (int j = n; j <= jmax; ++j) {
        size_t at = (size_t)(i - 1) * N2 + (j - 1);
        float best = C[at];
        if (n > 1) best = prv[at - N2 - 1] + C[at];       /* dyn3d.cc:200,227,298 */
        uint8_t code = 0;
        if (i > n && cur[at - N2] <= best) { best = cur[at - N2]; code = 1; }
        if (j > n && cur[at - 1]  <= best) { best = cur[at - 1];  code = 2; }
        cur[at] = best;
        if (bp) bp[(size_t)(n - 1) * plane + at] = code;
      }
    cost[n - 1] = cur[(size_t)(imax - 1) * N2 + (jmax - 1)]; /* dyn3d.cc:176,385 */
  }
  free(C); free(lay);
}

float orc_initial_clip(const orc_cfg *cfg) { return clip_metric(cfg); }

void orc_dp_clip(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot, float clip,
                 float *cost, uint8_t *bp)
{
  dp_with_clip(cfg, xyzA, xyzB_rot, clip, cost, bp);
}

void orc_dp(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot,
            float *cost, uint8_t *bp)
{
  dp_with_clip(cfg, xyzA, xyzB_rot, clip_metric(cfg), cost, bp);
}

void orc_traceback(const orc_cfg *cfg, const uint8_t *bp, int n, int16_t *pairs)
{
  size_t plane = (size_t)cfg->N1 * cfg->N2;
  int cn = n, ci = cfg->N1, cj = cfg->N2;                  /* dyn3d.cc:398-400 */
  while (cn > 0) {
    uint8_t code = bp[(size_t)(cn - 1) * plane + (size_t)(ci - 1) * cfg->N2 + (cj - 1)];
    if (code == 0) {
      pairs[2 * (cn - 1)] = (int16_t)(ci - 1);
      pairs[2 * (cn - 1) + 1] = (int16_t)(cj - 1);
      --cn; --ci; --cj;
    } else if (code == 1) --ci;
    else --cj;
  }
}

float orc_calc_min_rmsd(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                        int n, const int16_t *pairs, float xform[12], float *ssd_reduced)
{
  int nbb = cfg->nbb;
  float *fa = malloc((size_t)n * nbb * 3 * sizeof(float));
  float *fb = malloc((size_t)n * nbb * 3 * sizeof(float));
  for (int m = 0; m < n; ++m) {                     /* pair_alignment.cc:754-804 */
    memcpy(fa + (size_t)m * nbb * 3, xyzA + (size_t)pairs[2 * m] * nbb * 3, (size_t)nbb * 12);
    memcpy(fb + (size_t)m * nbb * 3, xyzB + (size_t)pairs[2 * m + 1] * nbb * 3, (size_t)nbb * 12);
  }
  float rmsd = orc_superimpose(n * nbb, fa, fb, xform);
  free(fa); free(fb);
  if (ssd_reduced)                                  /* search_or_dyn3d.cc:587-588 */
    *ssd_reduced = (rmsd * rmsd) * n * nbb;
  return rmsd;
}

/* ------------------------------------------------------------------------ */
/* Needleman-Wunsch orientation filter                                      */
/* ------------------------------------------------------------------------ */

/* Number of matched pairs in the alignment DynNW::Solve + ExportSolution
 * produce (lib/dyn/dynNW.cc:243-395,480-531) for the band InitTable lays out
 * (:147-236).  The table is rebuilt from scratch here; the reference keeps it
 * between calls, but every cell a call reads is either a border cell (constant)
 * or was written earlier in the same call. */
int orc_nw_num_matches(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot,
                       float clip, float gap, int min_matches)
{
  int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb;
  int a = N1 - min_matches + 1, b = N2 - min_matches + 1;    /* dynNW.cc:104-105 */
  int T2 = N2 + 1;
  size_t cells = (size_t)(N1 + 1) * T2;
  float *A = malloc(cells * sizeof(float));
  uint8_t *prev = malloc(cells);             /* 0 match, 1 i-1, 2 j-1, 3 uninitialised */
  for (size_t k = 0; k < cells; ++k) { A[k] = -1.0f; prev[k] = 3; }
  A[0] = 0.0f;                                                /* :201 */
  for (int i = 1; i <= N1; ++i) {                             /* :205-217 */
    int j = i - a; if (j < 0) j = 0;
    A[(size_t)i * T2 + j] = (float)(i + j) * gap; prev[(size_t)i * T2 + j] = 3;
  }
  for (int j = 1; j <= N2; ++j) {                             /* :221-234 */
    int i = j - b; if (i < 0) i = 0;
    A[(size_t)i * T2 + j] = (float)(i + j) * gap; prev[(size_t)i * T2 + j] = 3;
  }
  float G = A[1];                                             /* :255 */
  for (int i = 1; i <= N1; ++i) {                             /* :293-391 */
    int jlo = (i <= a) ? 1 : i - a + 1;
    int jhi = (i + b <= N2) ? i + b - 1 : N2;
    for (int j = jlo; j <= jhi; ++j) {
      size_t at = (size_t)i * T2 + j;
      float Cij = pair_cost(xyzA + 3 * (size_t)(i - 1) * nbb, xyzB_rot + 3 * (size_t)(j - 1) * nbb, nbb, clip);
      float m = A[at - T2 - 1] + Cij, up = A[at - T2] + G, left = A[at - 1] + G;
      A[at] = up; prev[at] = 1;
      if (left < A[at]) { A[at] = left; prev[at] = 2; }
      if (m < A[at]) { A[at] = m; prev[at] = 0; }
    }
  }
  int n = 0, ci = N1, cj = N2;                                /* :484-498 */
  while (ci > 0 && cj > 0) {
    uint8_t p = prev[(size_t)ci * T2 + cj];
    if (p == 0) { ++n; --ci; --cj; }
    else if (p == 1) --ci;
    else if (p == 2) --cj;
    else { ci = 0; cj = 0; }                                  /* dynNW.h:158-161 */
  }
  free(A); free(prev);
  return n;
}

/* The clip DistanceMetric holds after one GoodOrientation call: the filter
 * reads the current clip back as a physical distance and re-installs it from
 * that (orientation_filter.cc:96,127; distance_metric.cc:76-100), which is not
 * always the identity in float arithmetic. */
float orc_clip_after_filter_call(const orc_cfg *cfg, float clip)
{
  if (clip == ORC_UPPER_BOUND) return clip;
  float phys = sqrtf(clip / (float)cfg->nbb);
  if (phys == -1.0f) return ORC_UPPER_BOUND;
  return (phys * phys) * (float)cfg->nbb;
}

int orc_good_orientation(const orc_cfg *cfg, const float *xyzA,
                         const float *xyzB_rot, float *io_max_dist_metric)
{
  /* orientation_filter.cc:42-130 */
  int N1 = cfg->N1, N2 = cfg->N2, nbb = cfg->nbb;
  int minN = (N1 < N2) ? N1 : N2;
  float d = cfg->of_max_dist;
  float clip = (d == -1.0f) ? ORC_UPPER_BOUND : (d * d) * (float)nbb;      /* :97 */
  float gap = (float)(0.5 * (double)(d * d) * (double)minN * (double)nbb); /* :77-79 */
  int n = orc_nw_num_matches(cfg, xyzA, xyzB_rot, clip, gap, cfg->of_min_matches);
  *io_max_dist_metric = orc_clip_after_filter_call(cfg, *io_max_dist_metric);
  return n >= cfg->of_min_matches;                                          /* :129 */
}

/* ------------------------------------------------------------------------ */
/* Search over orientations and iterative refinement                        */
/* ------------------------------------------------------------------------ */

void orc_tables_init(const orc_cfg *cfg, orc_tables *t)
{
  int range_n = cfg->n_max - cfg->n_min + 1;
  static const float ident[12] = {1,0,0,0, 0,1,0,0, 0,0,1,0};
  for (int k = 0; k < range_n; ++k) {               /* search_or_dyn3d.cc:177-183 */
    t->ssd[k] = ORC_UPPER_BOUND;
    memcpy(t->xform + 12 * k, ident, sizeof ident);
    t->winner[k] = -1;
  }
  memset(t->pairs, 0xFF, (size_t)range_n * cfg->n_max * 2 * sizeof(int16_t));
}

static int max_rmsd_exceeded(const orc_cfg *cfg, float ssd, int n)
{                                                   /* search_or_dyn3d.cc:463-473 */
  return cfg->max_rmsd != -1.0f &&
         ssd > (float)(n * cfg->nbb) * (cfg->max_rmsd * cfg->max_rmsd);
}

/* One pass of the n-loop shared by FirstPass (search_or_dyn3d.cc:1703-1851)
 * and Refine (:2417-2617): returns the number of table updates.  `tag` is
 * stored in winner[]; need2refine/self_n implement :2544. */
static int sweep_n(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                   const float *xyzB_rot, const float *orient, int64_t tag,
                   orc_tables *t, float *row_ssd, unsigned char *need2refine,
                   int self_n, uint8_t *bp, float *cost, int16_t *tmp_pairs, float clip)
{
  int updates = 0;
  dp_with_clip(cfg, xyzA, xyzB_rot, clip, cost, bp);
  for (int n = 2; n <= cfg->n_max; ++n) {
    if (cost[n - 1] >= ORC_UPPER_BOUND) break;      /* :1720, dyn3d.h:407-408 */
    if (n < cfg->n_min) continue;
    int k = n - cfg->n_min;
    orc_traceback(cfg, bp, n, tmp_pairs);
    float ssd = cost[n - 1];
    float xf[12];
    memcpy(xf, orient, sizeof xf);
    if (cfg->refine_while_searching)
      orc_calc_min_rmsd(cfg, xyzA, xyzB, n, tmp_pairs, xf, &ssd);
    if (max_rmsd_exceeded(cfg, ssd, n)) break;      /* :1810 */
    if (row_ssd) row_ssd[k] = ssd;
    if (ssd < t->ssd[k]) {                          /* :1825-1835 */
      t->ssd[k] = ssd;
      memcpy(t->xform + 12 * k, xf, sizeof xf);
      memcpy(t->pairs + (size_t)k * cfg->n_max * 2, tmp_pairs, (size_t)n * 4);
      t->winner[k] = tag;
      if (need2refine && n != self_n) need2refine[k] = 1;
      ++updates;
    }
  }
  return updates;
}

int64_t orc_search(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                   const float *og, int64_t start, int64_t stop, orc_tables *t,
                   float *per_on_ssd)
{
  return orc_search_filtered(cfg, xyzA, xyzB, og, start, stop, t, per_on_ssd, NULL, NULL);
}

int64_t orc_search_filtered(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                            const float *og, int64_t start, int64_t stop, orc_tables *t,
                            float *per_on_ssd, uint8_t *pass_flags, float *io_clip)
{
  int range_n = cfg->n_max - cfg->n_min + 1;
  size_t plane = (size_t)cfg->N1 * cfg->N2;
  uint8_t *bp = malloc((size_t)cfg->n_max * plane);
  float *cost = malloc((size_t)cfg->n_max * sizeof(float));
  int16_t *tmp = malloc((size_t)cfg->n_max * 4);
  float *rot = malloc((size_t)cfg->N2 * cfg->nbb * 12);
  float clip_state = io_clip ? *io_clip : clip_metric(cfg);
  int64_t passed = 0;
  for (int64_t o = start; o < stop; ++o) {
    float *row = per_on_ssd ? per_on_ssd + (size_t)(o - start) * range_n : NULL;
    if (row) for (int k = 0; k < range_n; ++k) row[k] = NAN;
    orc_apply_transform(og + 12 * o, xyzB, cfg->N2 * cfg->nbb, rot); /* :1691 */
    int good = !cfg->of_enable || orc_good_orientation(cfg, xyzA, rot, &clip_state);
    if (pass_flags) pass_flags[o - start] = (uint8_t)good;
    if (!good) continue;
    ++passed;
    sweep_n(cfg, xyzA, xyzB, rot, og + 12 * o, o, t, row, NULL, -1, bp, cost, tmp, clip_state);
  }
  if (io_clip) *io_clip = clip_state;
  free(bp); free(cost); free(tmp); free(rot);
  return passed;
}

int64_t orc_refine(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                   orc_tables *t, int64_t max_iters, int64_t *num_new)
{
  return orc_refine_clip(cfg, xyzA, xyzB, t, max_iters, num_new, clip_metric(cfg));
}

/* `clip`: the DistanceMetric cutoff in force when Refine runs, i.e. what the
 * filter calls of the search left behind. */
int64_t orc_refine_clip(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                        orc_tables *t, int64_t max_iters, int64_t *num_new, float clip)
{
  int range_n = cfg->n_max - cfg->n_min + 1;
  size_t plane = (size_t)cfg->N1 * cfg->N2;
  uint8_t *bp = malloc((size_t)cfg->n_max * plane);
  float *cost = malloc((size_t)cfg->n_max * sizeof(float));
  int16_t *tmp = malloc((size_t)cfg->n_max * 4);
  int16_t *old = malloc((size_t)cfg->n_max * 4);
  float *rot = malloc((size_t)cfg->N2 * cfg->nbb * 12);
  unsigned char *need = malloc((size_t)range_n);
  for (int k = 0; k < range_n; ++k)                 /* :2348-2358 */
    need[k] = t->ssd[k] < ORC_UPPER_BOUND;
  int64_t updates = 0, it;
  FILE *log = getenv("ORC_REFINE_LOG") ? fopen(getenv("ORC_REFINE_LOG"), "w") : NULL;
  float *ssd_before = log ? malloc((size_t)range_n * sizeof(float)) : NULL;
  for (it = 1; it <= max_iters || max_iters == -1; ++it) {
    /* median of the set still to refine, rounded up
     * (lib/global_utils/simple_numeric_utils.h:350-376) */
    int in_set = 0;
    for (int k = 0; k < range_n; ++k) in_set += need[k];
    if (in_set == 0) break;                         /* :2377-2385 */
    int half = (in_set + 1) / 2, k = -1, seen = 0;
    do { ++k; seen += need[k]; } while (seen < half);
    int n = cfg->n_min + k;

    float xf[12];                                   /* :2401-2407 */
    int16_t *best_pairs = t->pairs + (size_t)k * cfg->n_max * 2;
    orc_calc_min_rmsd(cfg, xyzA, xyzB, n, best_pairs, xf, NULL);
    orc_apply_transform(xf, xyzB, cfg->N2 * cfg->nbb, rot);  /* :2411 */
    memcpy(old, best_pairs, (size_t)n * 4);         /* :2424-2425 */
    if (log) memcpy(ssd_before, t->ssd, (size_t)range_n * sizeof(float));
    updates += sweep_n(cfg, xyzA, xyzB, rot, xf, -(2 + it), t, NULL, need, n,
                       bp, cost, tmp, clip);
    need[k] = memcmp(old, best_pairs, (size_t)n * 4) != 0;   /* :2619-2629 */
    if (log) {                                      /* study of the iteration order: k, stays in the set, entries improved */
      fprintf(log, "%lld %d %d", (long long)it, k, (int)need[k]);
      for (int q = 0; q < range_n; ++q)
        if (t->ssd[q] != ssd_before[q]) fprintf(log, " %d", q);
      fprintf(log, "\n");
    }
  }
  if (log) { fclose(log); free(ssd_before); }
  if (num_new) *num_new = updates;
  free(bp); free(cost); free(tmp); free(old); free(rot); free(need);
  return it;
}
