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
// each other, so each gets its own lane.  What is NOT a running sum is done
// with one lane per point, 64 points at a time: gathering the coordinates
// (the next 64 are under way while the sums of the current 64 are taken) and,
// in the second pass, the twelve products of a point (9 for M, 3 for E0), which
// go to LDS as twelve rows of 64 floats, one per summed quantity.  The serial
// part of the second pass is then three adds per point on every lane (and three
// 16-byte LDS reads per four points): lane 3i+j adds M's product and two zeros
// (x + 0 is x: a sum that starts at +0 is never -0), lane 9 adds e0, e1, e2 in turn.
// The eigen-solve runs redundantly on every lane.
//
// pairs: n matched pairs (LDS or global), point p = pair p / nbb, atom p % nbb
//        (PairwiseAlignment::CalcMinRMSD, pair_alignment.cc:754-804);
// chunk: WAVE_LDS_FLOATS floats of LDS owned by this wave, 16-byte aligned.
// Must be called by all 64 lanes of a single-wave workgroup.
// ---------------------------------------------------------------------------
constexpr int WAVE_LDS_FLOATS = 64 * 16;

__device__ __forceinline__ float lane_value(float v, int src_lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

struct PointPair { float f[3], m[3]; };

__device__ __forceinline__ PointPair gather_point(int p, int nbb, const short2 *pairs,
                                                  const float *__restrict__ xyzA,
                                                  const float *__restrict__ xyzB)
{
  const int m = p / nbb, q = p - m * nbb;
  const short2 pr = pairs[m];
  const float *a = xyzA + 3 * ((size_t)pr.x * nbb + q);
  const float *b = xyzB + 3 * ((size_t)pr.y * nbb + q);
  PointPair pt;
  pt.f[0] = a[0]; pt.f[1] = a[1]; pt.f[2] = a[2];
  pt.m[0] = b[0]; pt.m[1] = b[1]; pt.m[2] = b[2];
  return pt;
}

#ifdef MINRMS_COOP_STAMPS
static __device__ unsigned long long g_fit_stamps[4];      // diagnostic build: cycle counter after pass 1, pass 2, the eigen-solve
#endif

__device__ inline float wave_superimpose(const int n, const int nbb, const short2 *pairs,
                                         float *chunk, const float *__restrict__ xyzA,
                                         const float *__restrict__ xyzB, float (&xf)[12])
{
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long fs0 = __builtin_readcyclecounter();
#endif
  const int lane = threadIdx.x & 63;
  const int npts = n * nbb;
  PointPair pt;
  // pass 1 (vect_3d.cc:73-94): lane k < 3 sums fixed[.][k], lane 3+k sums moveable[.][k].
  // Both passes keep the values of 64 points in LDS as one row of 64 floats per summed quantity,
  // so a summing lane takes in four points with one 16-byte read; the lanes behind the last point
  // store zeros, which change no sum (x + 0 is x: a sum that starts at +0 is never -0), and the
  // loops need no test for the end.
  float acc = 0.0f;
  if (lane < npts) pt = gather_point(lane, nbb, pairs, xyzA, xyzB);
  for (int base = 0; base < npts; base += 64) {
    const int cnt = (npts - base < 64) ? npts - base : 64;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      chunk[64 * k + lane] = lane < cnt ? pt.f[k] : 0.0f;
      chunk[64 * (3 + k) + lane] = lane < cnt ? pt.m[k] : 0.0f;
    }
    __syncthreads();
    if (base + 64 + lane < npts) pt = gather_point(base + 64 + lane, nbb, pairs, xyzA, xyzB);
    if (lane < 6) {
      const float4 *mine = (const float4 *)(chunk + 64 * lane);
#pragma unroll 4
      for (int q = 0; 4 * q < cnt; ++q) {
        const float4 v = mine[q];
        acc += v.x; acc += v.y; acc += v.z; acc += v.w;
      }
    }
  }
  acc *= 1.0f / (float)npts;
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long fs1 = __builtin_readcyclecounter();
#endif
  Superimpose S;
#pragma unroll
  for (int k = 0; k < 3; ++k) { S.sf[k] = lane_value(acc, k); S.sm[k] = lane_value(acc, 3 + k); }

  // pass 2 (superimpose.cc:49-68,258-263): rows 0..8 the products for M, 9..11 the three squares for
  // E0, 12 and 13 zeros.  Lane 3i+j adds M's product and two of the zeros, lane 9 adds e0, e1, e2
  // in turn: three adds per point on every lane, as the order of E0's sum demands.
  const int row0 = lane < 9 ? lane : 9, row1 = lane < 9 ? 12 : 10, row2 = lane < 9 ? 13 : 11;
  const float4 *r0 = (const float4 *)(chunk + 64 * row0), *r1 = (const float4 *)(chunk + 64 * row1);
  const float4 *r2 = (const float4 *)(chunk + 64 * row2);
  acc = 0.0f;
  if (lane < npts) pt = gather_point(lane, nbb, pairs, xyzA, xyzB);
  __syncthreads();                                             // pass 1 has read its last rows
  chunk[64 * 12 + lane] = 0.0f;
  chunk[64 * 13 + lane] = 0.0f;
  for (int base = 0; base < npts; base += 64) {
    const int cnt = (npts - base < 64) ? npts - base : 64;
    __syncthreads();
    {
      float fc[3], mc[3], e[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        fc[k] = pt.f[k] - S.sf[k];
        mc[k] = pt.m[k] - S.sm[k];
        const float df = fc[k] - mc[k];
        e[k] = df * df;
      }
      const bool live = lane < cnt;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) chunk[64 * (3 * i + j) + lane] = live ? mc[i] * fc[j] : 0.0f;
        chunk[64 * (9 + i) + lane] = live ? e[i] : 0.0f;
      }
    }
    __syncthreads();
    if (base + 64 + lane < npts) pt = gather_point(base + 64 + lane, nbb, pairs, xyzA, xyzB);
#pragma unroll 2
    for (int q = 0; 4 * q < cnt; ++q) {
      const float4 a = r0[q], b = r1[q], c = r2[q];
      acc += a.x; acc += b.x; acc += c.x;
      acc += a.y; acc += b.y; acc += c.y;
      acc += a.z; acc += b.z; acc += c.z;
      acc += a.w; acc += b.w; acc += c.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S.M[i][j] = lane_value(acc, 3 * i + j);
  S.E0 = lane_value(acc, 9);
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long fs2 = __builtin_readcyclecounter();
  const float r = S.finish(npts, xf);
  if (lane == 0 && n == 150) {
    g_fit_stamps[0] = fs1 - fs0; g_fit_stamps[1] = fs2 - fs1; g_fit_stamps[2] = __builtin_readcyclecounter() - fs2;
  }
  return r;
#else
  return S.finish(npts, xf);
#endif
}

}  // namespace fit
