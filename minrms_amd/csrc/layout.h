// layout.h -- how one orientation's 3-index DP is cut into wave-sized pieces,
// shared by host and device code.
//
// Layers (n = number of matched pairs) go across the 64 lanes of a wave, cells
// (i,j) of all those layers are visited together, row by row.  A *layer block*
// b holds layers 63b .. 63b+63: lane 0 is a passive copy of layer 63b (all
// zeros for b = 0, otherwise the values lane 63 of block b-1 produced), lanes
// 1..63 compute layers 63b+1 .. 63b+63.  Rows and columns of block b start at
// 63b+1.  A row is cut into P panels of PB columns (PB a multiple of 16, at
// most 192) so that one row of every layer of the block fits in registers.
#pragma once
#include <stdint.h>

#define MINRMS_LAYERS_PER_BLOCK 63
#define MINRMS_MAX_BLOCKS 16
#define MINRMS_MAX_PB 192
#define MINRMS_MAX_RES (MINRMS_LAYERS_PER_BLOCK * MINRMS_MAX_BLOCKS)

struct BlockDesc {
  int32_t j0;            // first row/column (1-based) of the block = 63b+1
  int32_t PB;            // panel width
  int32_t P;             // panels per row
  int32_t nrows;         // N1 - 63b
  int32_t W;             // 32-bit back-pointer words per row and lane = P*PB/16
  int32_t plane_stride;  // floats per row of the plane that feeds this block (b>=1)
  int64_t bp_off;        // offset (in 32-bit words) of the block's back-pointers
  int64_t plane_off;     // offset (in floats) of the plane that feeds this block
};

struct DpLayout {
  int32_t N1, N2, n_max, NB;
  int64_t c_floats;      // per orientation: N1*N2 + tail padding
  int64_t bp_words;      // per orientation
  int64_t plane_floats;  // per orientation
  int64_t edge_floats;   // per orientation (panel hand-over, 0 if P == 1 everywhere)
  BlockDesc blk[MINRMS_MAX_BLOCKS];
};

// Back-pointer of cell (n,i,j), 1-based, inside one orientation's table:
//   b = (n-1)/63, lane = n - 63b, r = i - j0, c = j - j0,
//   word = bp[blk.bp_off + (r*W + c/16)*64 + lane]
//   two bits at position 30 - 2*(c%16): 0 = match (n-1,i-1,j-1),
//   2 or 3 = (n,i,j-1), 1 = (n,i-1,j)      (dyn3d.cc:53-58 uses 0/1/2)
