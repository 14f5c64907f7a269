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
#define MINRMS_MAX_BLOCKS 33
#define MINRMS_MAX_PB 192
#define MINRMS_MAX_RES (MINRMS_LAYERS_PER_BLOCK * MINRMS_MAX_BLOCKS)

struct BlockDesc {
  int32_t j0;            // first row/column (1-based) of the block = 63b+1
  int32_t PB;            // panel width
  int32_t P;             // panels per row
  int32_t nrows;         // N1 - 63b
  int32_t W;             // 32-bit back-pointer words per row and lane = P*PB/16
  int32_t plane_stride;  // floats per row of the plane that feeds this block (b>=1)
  int32_t tail8;         // 1: the last 8 columns of the last panel lie outside the table and are not computed (sweep kernels)
  int32_t pad_;
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

// ---- the cooperative kernel (dp::k_dp_coop, dp_kernel.h): sizes and the spread of tasks over workgroups ----
namespace dp {
#define MINRMS_COOP_RING 8
#define MINRMS_COOP_ORING 16       // rows of the LDS out ring (plane rows on their way to the publish wave, which follows the segment's LAST task)
#define MINRMS_COOP_PRING 16        // rows of the LDS ring that holds the plane rows a segment is fed
#define MINRMS_COOP_MAX_TASKS 16
#define MINRMS_COOP_MAX_GROUPS 96
#define MINRMS_COOP_SPIN_LIMIT (1 << 22)
#define MINRMS_COOP_ROW_BITS 12     // rows of a layer block < 4096
// How the tasks of one orientation are spread over workgroups.  A group is
// either a run of whole layer blocks, or a *segment*: some consecutive panels of
// ONE block.  Segments put the panels of a block on several CUs so that every
// task has a SIMD to itself (a row of a block is then paced by one panel, not
// by the three that share the fullest SIMD).  The column a segment's first task
// needs from the segment to its left travels through global memory as 8-byte
// granules {value, tag} per lane and row -- tag = launch tag + row + 1, written
// through (sc1) in one store, so data and "it is there" arrive together -- and
// is brought into the LDS edge ring by one extra wave of the consuming
// workgroup (fetch_edge), for which the first task then waits exactly as it
// would for a left neighbour in its own workgroup.  That wave only loads and
// the producing task only stores: vector-memory operations of a wave complete
// in issue order, so a wave that did both per row would wait for its own
// write-through stores (1-2 us each) before it could take delivery of any load.
struct CoopGroups {
  int32_t n_groups;                 // workgroups per orientation
  int32_t max_tasks;                // compute waves per workgroup = largest group
  int32_t total_tasks;
  int32_t spin_limit;               // polls after which a wait gives up (MINRMS_COOP_SPIN_LIMIT; tests force a tiny one)
  int32_t fetch_waves;              // 1: some group is a segment that starts inside a block (one more wave per workgroup)
  int32_t asm_rows;                 // 1: tasks of 16-column segments run the hand-written row loop (dp_task_asm.h)
  int64_t xedge_granules;           // per orientation: granules of all segment boundaries
  int16_t blk[MINRMS_COOP_MAX_GROUPS];          // first block of the group
  int16_t pan[MINRMS_COOP_MAX_GROUPS];          // first panel of that block (0 for groups of whole blocks)
  int16_t ntask[MINRMS_COOP_MAX_GROUPS];        // compute tasks of the group
  int16_t first_task[MINRMS_COOP_MAX_GROUPS];   // block-major number of the group's first task
  int16_t whole[MINRMS_COOP_MAX_GROUPS];        // 1: whole blocks, 0: a segment of one block
  int32_t xin_off[MINRMS_COOP_MAX_GROUPS];      // granule offset of the boundary to the left of the group, -1: none
  int32_t xout_off[MINRMS_COOP_MAX_GROUPS];     // granule offset of the boundary to its right, -1: none
  int16_t block_first_task[MINRMS_MAX_BLOCKS + 1];   // block-major number of the first task of each block
  // segments only: the plane that feeds block b (lane 63's rows of block b-1) as granules,
  // [rows of block b + 1][columns of block b-1]
  int32_t plane_goff[MINRMS_MAX_BLOCKS + 1];    // granule offset, -1: none
  int32_t plane_gstride[MINRMS_MAX_BLOCKS + 1]; // granules per row = columns the producers write
};
}  // namespace dp

// Back-pointer of cell (n,i,j), 1-based, inside one orientation's table:
//   b = (n-1)/63, lane = n - 63b, r = i - j0, c = j - j0,
//   word = bp[blk.bp_off + (r*W + c/16)*64 + lane]
//   two bits at position 30 - 2*(c%16): 0 = match (n-1,i-1,j-1),
//   2 or 3 = (n,i,j-1), 1 = (n,i-1,j)      (dyn3d.cc:53-58 uses 0/1/2)
