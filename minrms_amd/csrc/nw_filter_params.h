// nw_filter_params.h -- parameters of the Needleman-Wunsch orientation filter (nw_filter.h),
// in a header of their own so that host-only units can hold them.
#pragma once
#include <stdint.h>

namespace nwf {

struct Params {
  int32_t N1, N2, nbb;
  int32_t win_a, win_b;    // aWindowSize[0], aWindowSize[1] (dynNW.cc:104-105)
  int32_t min_matches;
  float clip;              // SQR(max_phys_distance) * nbb (distance_metric.cc:84-85)
  float gap;               // orientation_filter.cc:77-79
  float upper_bound;       // DistanceMetric::UPPER_BOUND
};

}  // namespace nwf
