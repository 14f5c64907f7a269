// fit_device.h -- per-thread least-squares superposition for gfx950.
//
// Device restatement of Superimpose::FindTransformMinRMSD
// (lib/superimpose/superimpose.cc:17-287 of the reference); the 4x4 symmetric
// eigen-solve it calls is jac::jacobi_sym<4> (jacobi_sym.h), held in registers.
//
// Build with -ffp-contract=off: no product may be fused into a following sum.
#pragma once
#include <hip/hip_runtime.h>
#include "jacobi_sym.h"

namespace fit {

// Accumulators of one superposition problem.  Points are fed twice, in the
// same order: first to sum(), then (after centre()) to accumulate().
struct Superimpose {
  float sf[3], sm[3];   // running coordinate sums -> centroids
  float M[3][3];
  float E0;

  __device__ __forceinline__ void begin()
  {
#pragma unroll
    for (int k = 0; k < 3; ++k) { sf[k] = 0.0f; sm[k] = 0.0f; }
  }
  // vect_3d.cc:73-94
  __device__ __forceinline__ void sum(const float f[3], const float m[3])
  {
#pragma unroll
    for (int k = 0; k < 3; ++k) { sf[k] += f[k]; sm[k] += m[k]; }
  }
  __device__ __forceinline__ void centre(int npts)
  {
    const float inv = 1.0f / (float)npts;
#pragma unroll
    for (int k = 0; k < 3; ++k) { sf[k] *= inv; sm[k] *= inv; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[i][j] = 0.0f;
    E0 = 0.0f;
  }
  // superimpose.cc:49-68 and :258-263
  __device__ __forceinline__ void accumulate(const float f[3], const float m[3])
  {
    float fc[3], mc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { fc[k] = f[k] - sf[k]; mc[k] = m[k] - sm[k]; }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[i][j] += mc[i] * fc[j];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float df = fc[k] - mc[k]; E0 += df * df; }
  }
  // superimpose.cc:75-285; returns the rmsd, writes the 3x4 transform
  __device__ inline float finish(int npts, float xf[12])
  {
    const float tr = M[0][0] + M[1][1] + M[2][2];
    float P[4][4], d[4], ev[4][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        P[i][j] = (M[i][j] + M[j][i]) - ((i == j) ? (2 * tr) : 0.0f);
    const float V0 = M[1][2] - M[2][1], V1 = M[2][0] - M[0][2], V2 = M[0][1] - M[1][0];
    P[0][3] = P[3][0] = V0; P[1][3] = P[3][1] = V1; P[2][3] = P[3][2] = V2;
    P[3][3] = 0.0f;
    jac::jacobi_sym<4>(P, d, ev);
    // first strictly-largest eigenvalue (superimpose.cc:187-196)
    float lam = d[0];
    float q[4] = { ev[0][0], ev[1][0], ev[2][0], ev[3][0] };
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (d[k] > lam) { lam = d[k]; q[0] = ev[0][k]; q[1] = ev[1][k]; q[2] = ev[2][k]; q[3] = ev[3][k]; }
    const float len = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const float inv = (float)(1.0 / (double)len);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] *= inv;
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
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float rc = R[i][0] * sm[0] + R[i][1] * sm[1] + R[i][2] * sm[2];
      xf[4 * i + 0] = R[i][0]; xf[4 * i + 1] = R[i][1]; xf[4 * i + 2] = R[i][2];
      xf[4 * i + 3] = sf[i] - rc;
    }
    float min_ssd = E0 - 2 * lam;
    if (min_ssd < 0.0f) min_ssd = 0.0f;
    return sqrtf(min_ssd / (float)npts);
  }
};

// ---------------------------------------------------------------------------
// One wave, one superposition.  The sums of FindTransformMinRMSD are sixteen
// independent running sums (6 centroid coordinates, 9 entries of M, E0), each
// of which must see the points in the reference's order; they do not depend on
// each other, so each gets its own lane.  The points are gathered 64 at a time
// into LDS by all lanes.  The eigen-solve then runs redundantly on every lane.
//
// pairs: n matched pairs (LDS or global), point p = pair p / nbb, atom p % nbb
//        (PairwiseAlignment::CalcMinRMSD, pair_alignment.cc:754-804);
// chunk: 64*6 floats of LDS owned by this wave.
// Must be called by all 64 lanes of a single-wave workgroup.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float lane_value(float v, int src_lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

__device__ __forceinline__ void gather_point(int p, int nbb, const short2 *pairs,
                                             const float *__restrict__ xyzA,
                                             const float *__restrict__ xyzB, float *dst)
{
  const int m = p / nbb, q = p - m * nbb;
  const short2 pr = pairs[m];
  const float *a = xyzA + 3 * ((size_t)pr.x * nbb + q);
  const float *b = xyzB + 3 * ((size_t)pr.y * nbb + q);
  dst[0] = a[0]; dst[1] = a[1]; dst[2] = a[2];
  dst[3] = b[0]; dst[4] = b[1]; dst[5] = b[2];
}

__device__ inline float wave_superimpose(const int n, const int nbb, const short2 *pairs,
                                         float *chunk, const float *__restrict__ xyzA,
                                         const float *__restrict__ xyzB, float (&xf)[12])
{
  const int lane = threadIdx.x & 63;
  const int npts = n * nbb;
  // pass 1 (vect_3d.cc:73-94): lane k < 3 sums fixed[.][k], lane 3+k sums moveable[.][k]
  float acc = 0.0f;
  for (int base = 0; base < npts; base += 64) {
    const int cnt = (npts - base < 64) ? npts - base : 64;
    __syncthreads();
    if (lane < cnt) gather_point(base + lane, nbb, pairs, xyzA, xyzB, chunk + 6 * lane);
    __syncthreads();
    if (lane < 6) {
#pragma unroll 8
      for (int q = 0; q < cnt; ++q) acc += chunk[6 * q + lane];
    }
  }
  acc *= 1.0f / (float)npts;
  Superimpose S;
#pragma unroll
  for (int k = 0; k < 3; ++k) { S.sf[k] = lane_value(acc, k); S.sm[k] = lane_value(acc, 3 + k); }

  // pass 2 (superimpose.cc:49-68,258-263): lane 3i+j owns M[i][j] = sum mc[i]*fc[j];
  // lanes 9..11 produce the three squared differences of a point, which are
  // added to E0 one after the other on every lane
  const int role = lane < 12 ? lane : 0;
  const bool is_e = role >= 9;
  const int ri = is_e ? role - 9 : role / 3;
  const int rj = is_e ? role - 9 : role - 3 * (role / 3);
  const float cm = ri == 0 ? S.sm[0] : ri == 1 ? S.sm[1] : S.sm[2];
  const float cf = rj == 0 ? S.sf[0] : rj == 1 ? S.sf[1] : S.sf[2];
  float m_acc = 0.0f, e0 = 0.0f;
  for (int base = 0; base < npts; base += 64) {
    const int cnt = (npts - base < 64) ? npts - base : 64;
    __syncthreads();
    if (lane < cnt) gather_point(base + lane, nbb, pairs, xyzA, xyzB, chunk + 6 * lane);
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < cnt; ++q) {
      const float mc = chunk[6 * q + 3 + ri] - cm;
      const float fc = chunk[6 * q + rj] - cf;
      const float df = fc - mc;
      const float u = is_e ? df : mc, w = is_e ? df : fc;
      const float term = u * w;
      m_acc += term;
      e0 += lane_value(term, 9);
      e0 += lane_value(term, 10);
      e0 += lane_value(term, 11);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S.M[i][j] = lane_value(m_acc, 3 * i + j);
  S.E0 = e0;
  return S.finish(npts, xf);
}

}  // namespace fit
