// jacobi_sym.h -- the one eigen-solver of the engine, for host and device code.
//
// Cyclic Jacobi method for a real symmetric N x N matrix (Rutishauser, Numer.
// Math. 9, 1966), N a compile-time constant (4: the quaternion matrix of a
// superposition; 3: principal axes of a structure).  It reproduces, bit for
// bit, what the reference gets from its eigen-solver in a C build (behaviour of
// lib/eigen/jacobi.c:7-85): float storage, and double arithmetic exactly where
// that build promotes to double -- every expression that contains a double
// literal or a libm call (:35,40-42,46,49-50,53,55) -- with the results
// narrowed to float where they are stored.  Compile with -ffp-contract=off.
//
// All loop bounds are compile-time constants, so on the GPU the matrices stay
// in registers.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define MINRMS_HD __host__ __device__ __forceinline__
#else
#define MINRMS_HD inline
#endif

namespace jac {

// |x| + g, evaluated in double and narrowed, equals |x| narrowed
MINRMS_HD bool negligible_beside(float x, float g)
{
  const double ax = fabs((double)x);
  return (float)(ax + (double)g) == (float)ax;
}

MINRMS_HD void rot_pair(float &x, float &y, float sine, float tan_half)
{
  const float x0 = x, y0 = y;
  x = x0 - sine * (y0 + x0 * tan_half);
  y = y0 + sine * (x0 - y0 * tan_half);
}

// a: upper triangle used, destroyed.  d: eigenvalues.  v: columns are the eigenvectors.
template <int N>
MINRMS_HD void jacobi_sym(float (&a)[N][N], float (&d)[N], float (&v)[N][N])
{
  float diag0[N], shift[N];   // the diagonal at the start of a sweep, and what the sweep adds to it
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int c = 0; c < N; ++c) v[r][c] = (r == c) ? 1.0f : 0.0f;
    diag0[r] = d[r] = a[r][r];
    shift[r] = 0.0f;
  }
  for (int sweep = 1; sweep <= 50; ++sweep) {
    float off_sum = 0.0f;
#pragma unroll
    for (int p = 0; p < N - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < N; ++q)
        off_sum = (float)((double)off_sum + fabs((double)a[p][q]));
    if (off_sum == 0.0f) return;
    const float skip_below = (sweep < 4) ? (float)(0.2 * (double)off_sum / (double)(N * N)) : 0.0f;
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        const float off = a[p][q];
        const float tiny = (float)(100.0 * fabs((double)off));
        if (sweep > 4 && negligible_beside(d[p], tiny) && negligible_beside(d[q], tiny)) {
          a[p][q] = 0.0f;
        } else if (fabs((double)off) > (double)skip_below) {
          float gap = d[q] - d[p];
          float t;
          if (negligible_beside(gap, tiny)) {
            t = off / gap;
          } else {
            const float cot_2phi = (float)(0.5 * (double)gap / (double)off);
            const float cot_sq = cot_2phi * cot_2phi;
            t = (float)(1.0 / (fabs((double)cot_2phi) + sqrt(1.0 + (double)cot_sq)));
            if (cot_2phi < 0.0f) t = -t;
          }
          const float sec_sq = 1.0f + t * t;
          const float cosine = (float)(1.0 / sqrt((double)sec_sq));
          const float sine = t * cosine;
          const float tan_half = (float)((double)sine / (1.0 + (double)cosine));
          const float moved = t * off;
          shift[p] -= moved; shift[q] += moved;
          d[p] -= moved; d[q] += moved;
          a[p][q] = 0.0f;
#pragma unroll
          for (int j = 0; j < N; ++j) {
            if (j < p)               rot_pair(a[j][p], a[j][q], sine, tan_half);
            else if (j > p && j < q) rot_pair(a[p][j], a[j][q], sine, tan_half);
            else if (j > q)          rot_pair(a[p][j], a[q][j], sine, tan_half);
          }
#pragma unroll
          for (int j = 0; j < N; ++j) rot_pair(v[j][p], v[j][q], sine, tan_half);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      diag0[r] += shift[r];
      d[r] = diag0[r];
      shift[r] = 0.0f;
    }
  }
}

}  // namespace jac
