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

#if defined(__HIPCC__) || defined(__CUDACC__)
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

MINRMS_HD void rot_pair(float &p, float &q, float s, float tau)
{
  const float g = p, h = q;
  p = g - s * (h + g * tau);
  q = h + s * (g - h * tau);
}

// a: upper triangle used, destroyed.  d: eigenvalues.  v: columns are the eigenvectors.
template <int N>
MINRMS_HD void jacobi_sym(float (&a)[N][N], float (&d)[N], float (&v)[N][N])
{
  float b[N], z[N];
#pragma unroll
  for (int r = 0; r < N; ++r) {
#pragma unroll
    for (int c = 0; c < N; ++c) v[r][c] = (r == c) ? 1.0f : 0.0f;
    b[r] = d[r] = a[r][r];
    z[r] = 0.0f;
  }
  for (int sweep = 1; sweep <= 50; ++sweep) {
    float sm = 0.0f;
#pragma unroll
    for (int p = 0; p < N - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < N; ++q)
        sm = (float)((double)sm + fabs((double)a[p][q]));
    if (sm == 0.0f) return;
    const float tresh = (sweep < 4) ? (float)(0.2 * (double)sm / (double)(N * N)) : 0.0f;
#pragma unroll
    for (int p = 0; p < N - 1; ++p) {
#pragma unroll
      for (int q = p + 1; q < N; ++q) {
        const float apq = a[p][q];
        const float g = (float)(100.0 * fabs((double)apq));
        if (sweep > 4 && negligible_beside(d[p], g) && negligible_beside(d[q], g)) {
          a[p][q] = 0.0f;
        } else if (fabs((double)apq) > (double)tresh) {
          float h = d[q] - d[p];
          float t;
          if (negligible_beside(h, g)) {
            t = apq / h;
          } else {
            const float theta = (float)(0.5 * (double)h / (double)apq);
            const float th2 = theta * theta;
            t = (float)(1.0 / (fabs((double)theta) + sqrt(1.0 + (double)th2)));
            if (theta < 0.0f) t = -t;
          }
          const float one_tt = 1.0f + t * t;
          const float c = (float)(1.0 / sqrt((double)one_tt));
          const float s = t * c;
          const float tau = (float)((double)s / (1.0 + (double)c));
          h = t * apq;
          z[p] -= h; z[q] += h;
          d[p] -= h; d[q] += h;
          a[p][q] = 0.0f;
#pragma unroll
          for (int j = 0; j < N; ++j) {
            if (j < p)               rot_pair(a[j][p], a[j][q], s, tau);
            else if (j > p && j < q) rot_pair(a[p][j], a[j][q], s, tau);
            else if (j > q)          rot_pair(a[p][j], a[q][j], s, tau);
          }
#pragma unroll
          for (int j = 0; j < N; ++j) rot_pair(v[j][p], v[j][q], s, tau);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      b[r] += z[r];
      d[r] = b[r];
      z[r] = 0.0f;
    }
  }
}

}  // namespace jac
