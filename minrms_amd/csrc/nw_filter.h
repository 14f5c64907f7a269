// Needleman-Wunsch orientation filter (-of) on the GPU.
//
// What it decides (lib/or_gen/orientation_filter.cc:88-130): align the two
// structures, at the orientation under test, with a 2-index DP whose match
// cost is the clipped squared distance and whose gap penalty is so large that
// gaps are only taken when unavoidable (lib/dyn/dynNW.cc:243-395); the
// orientation passes if the resulting alignment has at least `min_matches`
// matched pairs (dynNW.cc:480-531 counts them on the traceback).
//
// Mapping: one wave per orientation.  Lane l owns row i0+l of a 64-row strip and
// walks it left to right one cell per step, one step behind lane l-1 (a skewed
// wavefront), so the three predecessors of a cell are: its own previous value
// (left), the value lane l-1 produced one step ago (up) and two steps ago
// (diagonal).  The last row of a strip is parked in LDS for lane 0 of the next
// strip.  Instead of storing back-pointers and walking them, every cell carries
// the number of matches on its own traceback path: the path is unique, so the
// count at the corner cell is what the reference counts.
//
// The band (dynNW.cc:104-105,298-306) and its border cells (:201-234) are not
// special-cased in the sweep: every cell outside the band takes the border
// value (i+j)*G with a match count of 0.  Cells inside the band only ever read
// in-band cells or border cells, for which that is the reference's value.
#ifndef MINRMS_NW_FILTER_H
#define MINRMS_NW_FILTER_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nw_filter_params.h"

namespace nwf {

__global__ void __launch_bounds__(64)
k_nw_filter(const Params P, const int64_t count, const float *__restrict__ xyzA,
            const float *__restrict__ xyzB, const float *__restrict__ og,
            unsigned char *__restrict__ pass)
{
  extern __shared__ float nw_lds[];
  float *rowA = nw_lds;                         // [N2+1] last row of the previous strip
  int *rowC = (int *)(nw_lds + P.N2 + 1);       // [N2+1] its match counts
  const int lane = threadIdx.x;
  const int64_t o = blockIdx.x;
  if (o >= count) return;
  float xf[12];
#pragma unroll
  for (int q = 0; q < 12; ++q) xf[q] = og[12 * o + q];
  const float G = P.gap;

  // row 0 lies outside the band everywhere
  for (int j = lane; j <= P.N2; j += 64) { rowA[j] = (float)j * G; rowC[j] = 0; }
  __syncthreads();

  const int steps = P.N2 + 64;
  for (int i0 = 1; i0 <= P.N1; i0 += 64) {
    const int i = i0 + lane;
    const int jlo = (i <= P.win_a) ? 1 : i - P.win_a + 1;                 // dynNW.cc:298-301
    const int jhi = (i + P.win_b <= P.N2) ? i + P.win_b - 1 : P.N2;       // :303-306
    const bool row_ok = i <= P.N1;
    float curA = 0.0f, diagA = 0.0f;
    int curC = 0, diagC = 0;
    for (int t = 0; t < steps; ++t) {
      const int j = t - lane;
      float upA = __shfl_up(curA, 1);
      int upC = __shfl_up(curC, 1);
      if (lane == 0) {
        const int jj = j <= P.N2 ? j : P.N2;      // j >= 0 for lane 0
        upA = rowA[jj];
        upC = rowC[jj];
      }
      float best;
      int cnt;
      if (row_ok && j >= jlo && j <= jhi) {
        // DistanceMetric::InlinedSquaredDistanceBetweenIandJ (distance_metric.h:104-130)
        // on the moved copy of structure 2 (ApplyTransformBackbone, search_or_dyn3d.cc:1691)
        float c = 0.0f;
        for (int q = 0; q < P.nbb; ++q) {
          const float *a = xyzA + 3 * ((size_t)(i - 1) * P.nbb + q);
          const float *b = xyzB + 3 * ((size_t)(j - 1) * P.nbb + q);
          const float bx = xf[0] * b[0] + xf[1] * b[1] + xf[2] * b[2] + xf[3];
          const float by = xf[4] * b[0] + xf[5] * b[1] + xf[6] * b[2] + xf[7];
          const float bz = xf[8] * b[0] + xf[9] * b[1] + xf[10] * b[2] + xf[11];
          const float dx = a[0] - bx, dy = a[1] - by, dz = a[2] - bz;
          c += dx * dx + dy * dy + dz * dz;
        }
        if (c > P.clip) c = P.upper_bound;
        const float m = diagA + c, u = upA + G, l = curA + G;             // dynNW.cc:350-352
        best = u; cnt = upC;                                              // :354-355
        if (l < best) { best = l; cnt = curC; }                           // :356-359
        if (m < best) { best = m; cnt = diagC + 1; }                      // :360-363
      } else {
        best = (float)(i + j) * G;                                        // :211,227
        cnt = 0;
      }
      diagA = upA; diagC = upC;
      curA = best; curC = cnt;
      if (lane == 63 && j >= 0 && j <= P.N2) { rowA[j] = best; rowC[j] = cnt; }
      if (i == P.N1 && j == P.N2)
        pass[o] = (unsigned char)(cnt >= P.min_matches);                  // orientation_filter.cc:129
      __syncthreads();
    }
  }
}

}  // namespace nwf
#endif
