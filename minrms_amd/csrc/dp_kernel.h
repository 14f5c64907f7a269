// dp_kernel.h -- the 3-index dynamic program of one orientation on one wave.
//
// What it computes (reference: Dyn3d::InitTable_IncrN / Solve_IncrN,
// lib/dyn/dyn3d.cc:135-179,183-387):
//     cost(n,i,j) = min( cost(n-1,i-1,j-1) + C(i,j),   -> back-pointer 0
//                        cost(n,i-1,j),                 -> back-pointer 1 if <=
//                        cost(n,i,j-1) )                -> back-pointer 2 if <=
// with cost(0,.,.) = 0, and cost(n,i,j) undefined (+inf here) for i<n or j<n.
// min is exact and the only rounding is the single float add, so any evaluation
// order gives the reference's bits; the tie rule "later candidate wins on <="
// is recovered from equalities: with t = min(m,up) and v = min(t,left),
//     (t == up)   <=>  up   <= m            (reference picks pointer 1)
//     (v == left) <=>  left <= min(m,up)    (reference picks pointer 2)
//
// How it is mapped (see layout.h): lane = layer.  All lanes visit cell (i,j)
// together, so C(i,j) is wave-uniform and arrives through scalar loads; the
// value a lane needs from layer n-1 is its left neighbour's previous-row value
// one column to the left: a single DPP wave_shr:1.  One row of cost values per
// lane lives in PB registers; nothing but back-pointer bits (2 per cell), the
// per-block hand-over plane and the final costs ever leave the wave.
//
// Seven VALU instructions per cell:
//     v_add_f32      mp'  = C(i,j+1) + up(j)             (prepares next cell)
//     v_min_f32_dpp  t    = min(shr1(mp), up)
//     v_cmp_eq_f32   vcc  = (t == up)
//     v_addc_co_u32  bits = 2*bits + vcc
//     v_min_f32      v    = min(t, left)                 (written over up(j))
//     v_cmp_eq_f32   vcc  = (v == left)
//     v_addc_co_u32  bits = 2*bits + vcc
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.h"

namespace dp {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
// LDS pointers carry their address space so that every access is a DS
// instruction (in-order with the hand-written ds_read below), never FLAT.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) volatile int lds_vint;

struct OrientationBuffers {
  const float *C;     // [N1][N2] squared distances (+ tail padding)
  uint32_t *bp;       // back-pointer words
  float *planes;      // hand-over planes between layer blocks
  float *edges;       // [nrows][64] hand-over column between panels
  float *cost;        // [n_max] cost(n,N1,N2)
};

// Cooperative mode: all (layer block, panel) tasks of ONE orientation run as
// the waves of one workgroup and hand rows to each other while they run.
// prog[t] = rows task t has finished.  Row r of task (b,p) needs row r of its
// left neighbour (b,p-1) (through an LDS ring) and, for b > 0, row r+62 of the
// block above (through the plane in global memory, read past the L1).
//
// An orientation with more than 16 tasks is spread over several workgroups
// (CoopGroups): each takes a run of whole layer blocks, so panels still talk
// through LDS and only the plane between the last block of one group and the
// first block of the next crosses workgroups.  That hand-over follows the
// measured recipe for data passed between CUs inside one launch: every plane
// word is written by an sc1 (write-through) store, the storing wave drains its
// stores (s_waitcnt vmcnt(0)) and then raises its row counter in global memory
// with an sc1 store; the consumer polls the counter with sc1 loads and only then
// issues its sc1 loads of the plane row, from the same wave.  Producer groups
// come first in the grid, so a workgroup never waits for one that is dispatched
// after it.
#define MINRMS_COOP_RING 8
#define MINRMS_COOP_MAX_TASKS 16
#define MINRMS_COOP_SPIN_LIMIT (1 << 22)
#define MINRMS_COOP_ROW_BITS 11     // rows of a layer block < 2048
struct CoopGroups {
  int32_t n_groups;                 // workgroups per orientation
  int32_t max_tasks;                // waves per workgroup = largest group
  int32_t total_tasks;
  int32_t spin_limit;               // polls after which a wait gives up (MINRMS_COOP_SPIN_LIMIT; tests force a tiny one)
  int32_t first_block[MINRMS_MAX_BLOCKS + 1];   // group g = blocks first_block[g] .. first_block[g+1]-1
  int32_t first_task[MINRMS_MAX_BLOCKS + 1];    // block-major task number of the group's first task
};
struct CoopCtx {
  lds_vint *prog;
  lds_vint *abort_flag;
  lds_f32 *edge_ring;               // [task][MINRMS_COOP_RING][64]
  int task, left_task, right_task;
  int prod_first, prod_last;        // tasks of block b-1 whose columns this panel reads
  int prod_nrows;
  int prod_global;                  // the producers run in another workgroup: their ids index gprog
  int gtask;                        // this task's index in gprog
  int *gprog;                       // global: gbase + rows finished, per task of this orientation
  int gbase;                        // tag of this launch; values below it are left over from earlier launches
  int *err;                         // global: set when a wait gave up
  int spin_limit;
};

// Every wave reaches an exit: a wait that is not satisfied within the spin
// limit raises the abort flag (and the global error word); from then on no wait
// of the workgroup blocks any more, every task runs to the end of its rows on
// whatever data it has (all addresses are data-independent) and the host
// discards the batch.
//
// The waits remember the last progress value seen: a task that runs a few rows
// behind its neighbour then polls once per several rows, not every row.  In step
// with its neighbour a task waits a few hundred cycles in every row, so the
// first looks at the counter come back to back; the abort flag is only
// consulted, and the wave only put to sleep, once that did not help.
__device__ __forceinline__ void coop_wait_cached(const CoopCtx &cx, const int task, const int need, int &known)
{
  int spins = 0;
  while (known < need) {
    known = __builtin_amdgcn_readfirstlane(cx.prog[task]);
    if (known >= need) break;
    if (++spins < 8) continue;
    if (__builtin_amdgcn_readfirstlane(*cx.abort_flag)) break;
    __builtin_amdgcn_s_sleep(1);
    if (spins > cx.spin_limit) {
      *cx.abort_flag = 1;
      *cx.err = 1;
    }
  }
}

// Progress of a task of another workgroup.
__device__ __forceinline__ void coop_wait_global(const CoopCtx &cx, const int gtask, const int need, int &known)
{
  int spins = 0;
  while (known < need) {
    known = __builtin_amdgcn_readfirstlane(
        __hip_atomic_load(cx.gprog + gtask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - cx.gbase;
    if (known >= need || __builtin_amdgcn_readfirstlane(*cx.abort_flag)) break;
    __builtin_amdgcn_s_sleep(2);
    if (++spins > cx.spin_limit) {
      *cx.abort_flag = 1;
      *cx.err = 1;
    }
  }
}

__device__ __forceinline__ void coop_wait_producer(const CoopCtx &cx, const int task, const int need, int &known)
{
  if (cx.prod_global) coop_wait_global(cx, task, need, known);
  else coop_wait_cached(cx, task, need, known);
}

#define MINRMS_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// The scalar loads of the distance stream are issued from inline asm and land
// some time later; until the wait nothing may read their destination, and that
// includes copies the compiler makes on its own (a register-to-register move of
// a loop-carried value was once scheduled between a load and its wait and
// carried the registers' previous contents on).  The destinations are therefore
// fixed physical registers, named in the load, which defines a value living
// there, and in the wait, which consumes and redefines it in place: there is no
// register for the compiler to move the value to before the wait, and whatever
// it does with the value afterwards reads registers that have landed.
#define MINRMS_SREG_CN "{s[84:99]}"
#define MINRMS_SREG_C0 "{s83}"
__device__ __forceinline__ void sload16(f16v &v, const float *p)
{
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=" MINRMS_SREG_CN(v) : "s"(p) : "memory");
}
__device__ __forceinline__ void sload1(float &v, const float *p)
{
  asm volatile("s_load_dword %0, %1, 0x0" : "=" MINRMS_SREG_C0(v) : "s"(p) : "memory");
}
#define MINRMS_WAIT_SLOADS(A, B) \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+" MINRMS_SREG_CN(A), "+" MINRMS_SREG_C0(B) : : "memory")

// Four consecutive cells.  MP holds shr-source "C(i,j) + up(j-1)" of the first
// cell on entry and of the cell after the fourth on exit.
#define MINRMS_DP_ONE(MPI, MPO, CN, RJ, RJM1)                                         \
      "v_add_f32 " MPO ", " CN ", " RJ "\n\t"                                          \
      "v_min_f32_dpp %[t], " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_cmp_eq_f32 vcc, %[t], " RJ "\n\t"                                            \
      "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t"                         \
      "v_min_f32 " RJ ", %[t], " RJM1 "\n\t"                                          \
      "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t"                                        \
      "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t"

#define MINRMS_DP_CELL4(MP, C0, C1, C2, C3, R0, R1, R2, R3, RM1)                      \
  asm volatile(                                                                       \
      MINRMS_DP_ONE("%[mp]", "%[mq]", "%[c0]", "%[r0]", "%[rm1]")                     \
      MINRMS_DP_ONE("%[mq]", "%[mp]", "%[c1]", "%[r1]", "%[r0]")                      \
      MINRMS_DP_ONE("%[mp]", "%[mq]", "%[c2]", "%[r2]", "%[r1]")                      \
      MINRMS_DP_ONE("%[mq]", "%[mp]", "%[c3]", "%[r3]", "%[r2]")                      \
      : [mp] "+v"(MP), [mq] "=&v"(mq), [t] "=&v"(t), [bits] "+v"(bits),               \
        [r0] "+v"(R0), [r1] "+v"(R1), [r2] "+v"(R2), [r3] "+v"(R3)                    \
      : [rm1] "v"(RM1), [c0] "s"(C0), [c1] "s"(C1), [c2] "s"(C2), [c3] "s"(C3)        \
      : "vcc")

// The same four cells without back-pointers: a search that keeps only the
// costs (no refinement while searching, search_or_dyn3d.cc:1783-1790 not taken)
// needs no traceback of the orientations that do not win.
#define MINRMS_DP_ONE_NB(MPI, MPO, CN, RJ, RJM1)                                      \
      "v_add_f32 " MPO ", " CN ", " RJ "\n\t"                                          \
      "v_min_f32_dpp %[t], " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_min_f32 " RJ ", %[t], " RJM1 "\n\t"

#define MINRMS_DP_CELL4_NB(MP, C0, C1, C2, C3, R0, R1, R2, R3, RM1)                   \
  asm volatile(                                                                       \
      MINRMS_DP_ONE_NB("%[mp]", "%[mq]", "%[c0]", "%[r0]", "%[rm1]")                  \
      MINRMS_DP_ONE_NB("%[mq]", "%[mp]", "%[c1]", "%[r1]", "%[r0]")                   \
      MINRMS_DP_ONE_NB("%[mp]", "%[mq]", "%[c2]", "%[r2]", "%[r1]")                   \
      MINRMS_DP_ONE_NB("%[mq]", "%[mp]", "%[c3]", "%[r3]", "%[r2]")                   \
      : [mp] "+v"(MP), [mq] "=&v"(mq), [t] "=&v"(t),                                  \
        [r0] "+v"(R0), [r1] "+v"(R1), [r2] "+v"(R2), [r3] "+v"(R3)                    \
      : [rm1] "v"(RM1), [c0] "s"(C0), [c1] "s"(C1), [c2] "s"(C2), [c3] "s"(C3))

#define MINRMS_ROW(j) row[j]
#ifndef MINRMS_C_PREFETCH
#define MINRMS_C_PREFETCH 1
#endif

// One panel (PB columns) of one layer block, all rows.
// PUBLISH: the plane this task writes is read by another workgroup.
// BP: write the back-pointers.
template <int PB, bool INJECT, bool HAS_NEXT, bool COOP, bool PUBLISH = false, bool BP = true>
__device__ __forceinline__ void dp_panel(const DpLayout &L, const OrientationBuffers &ob,
                                         const int b, const int p, const int lane,
                                         lds_f32 *lds, const CoopCtx &cx)
{
  const BlockDesc &bd = L.blk[b];
  const float INF = __builtin_inff();
  const int N2 = L.N2;
  const int jstart = bd.j0 + p * PB;            // 1-based column of cell 0
  const int nrows = bd.nrows;
  const int bp_row_words = bd.W;                // kept in a register: the layout lives in memory, and every
                                                // asm statement with a memory clobber makes the compiler read it again
  constexpr bool inject = INJECT;      // b > 0
  constexpr bool has_next = HAS_NEXT;  // b + 1 < NB
  const bool has_left = (p > 0);
  const bool has_right = (p + 1 < bd.P);
  const float base_init = (b == 0 && lane == 0) ? 0.0f : INF;

  float row[PB];
#pragma unroll
  for (int jj = 0; jj < PB; ++jj) row[jj] = base_init;

  // plane that feeds this block (rows r, columns = producer cell index)
  const float *plane_in = ob.planes + bd.plane_off;
  const int si = bd.plane_stride;
  const int kin = p * PB + 62;                  // plane column of "one left of cell 0"
  float *plane_out = has_next ? ob.planes + L.blk[b + 1].plane_off : nullptr;
  const int so = has_next ? L.blk[b + 1].plane_stride : 0;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;   // LDS byte address of the staging row
  constexpr int U = (PB + 1 + 63) / 64;         // injection loads per lane and row
  float inj[U];
  int known_pa = 0, known_pb = 0, known_left = 0, known_right = 0;   // cached progress of the neighbours
  if (inject) {
    if (COOP) {
      const int need = (63 < cx.prod_nrows) ? 63 : cx.prod_nrows;
      coop_wait_producer(cx, cx.prod_first, need, known_pa);
      coop_wait_producer(cx, cx.prod_last, need, known_pb);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = lane + 64 * u;
      if (COOP) inj[u] = (idx <= PB) ? __hip_atomic_load(plane_in + kin + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
      else inj[u] = (idx <= PB) ? plane_in[kin + idx] : 0.0f;
    }
  }

  const float *Crow = ob.C + (size_t)(bd.j0 - 1) * N2 + (jstart - 1);
  uint32_t *bprow = ob.bp + bd.bp_off + ((size_t)p * (PB / 16)) * 64 + lane;
  float edge_prev = base_init;                  // left panel's value of the previous row

  // current and next 16 distances of the row, and the row's first one
  float c0 = 0.0f, c0n = 0.0f;
  f16v cn = {}, cnn = {};
  sload1(c0n, Crow);
  sload16(cnn, Crow + 1);
  MINRMS_WAIT_SLOADS(cnn, c0n);
  c0 = c0n;
  cn = cnn;

  for (int r = 0; r < nrows; ++r) {
    if constexpr (!COOP && MINRMS_C_PREFETCH) {
      // Warm the L2 with the next row's distances.  The scalar loads below run
      // one 16-column group ahead and cannot run further ahead (they return out
      // of order, so the only wait is for all of them); every line they fetch is
      // used once, so without help each comes from HBM, and a wave of a wide
      // panel (3 per SIMD) spent half its time waiting for them.  The lines
      // arrive through a vector load that writes to a piece of LDS nobody reads:
      // no register, no wait.
      if (r + 1 < nrows) {
        const float *nextrow = Crow + N2;
        uint32_t m0_saved;
        const uint32_t sink = lds_base + 4u * (uint32_t)(2 * (MINRMS_MAX_PB + 8));
#define MINRMS_PREFETCH_ROW(LOADS)                                                    \
        asm volatile("s_mov_b32 %0, m0\n\t"                                            \
                     "s_mov_b32 m0, %3\n\t"                                            \
                     "s_nop 0\n\t"                                                    \
                     LOADS                                                            \
                     "s_mov_b32 m0, %0"                                               \
                     : "=&s"(m0_saved) : "v"(lane * 4), "s"(nextrow), "s"(sink) : "memory")
        if constexpr (PB + 1 <= 64)
          MINRMS_PREFETCH_ROW("global_load_lds_dword %1, %2 offset:0\n\t");
        else if constexpr (PB + 1 <= 128)
          MINRMS_PREFETCH_ROW("global_load_lds_dword %1, %2 offset:0\n\t"
                              "global_load_lds_dword %1, %2 offset:256\n\t");
        else
          MINRMS_PREFETCH_ROW("global_load_lds_dword %1, %2 offset:0\n\t"
                              "global_load_lds_dword %1, %2 offset:256\n\t"
                              "global_load_lds_dword %1, %2 offset:512\n\t");
#undef MINRMS_PREFETCH_ROW
      }
    }
    float linit = base_init, prev_init = base_init;
    if (has_left) {
      if (COOP) {
        coop_wait_cached(cx, cx.left_task, r + 1, known_left);
        linit = cx.edge_ring[(cx.left_task * MINRMS_COOP_RING + (r % MINRMS_COOP_RING)) * 64 + lane];
      } else {
        linit = ob.edges[(size_t)r * 64 + lane];
      }
      prev_init = edge_prev;
      edge_prev = linit;
    }
    if (inject) {
      // previous block's lane-63 row -> LDS -> lane 0's row registers
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = lane + 64 * u;
        if (idx <= PB) lds[idx + 3] = inj[u];
      }
      const int rn = (r + 1 < nrows) ? r + 1 : r;
      if (COOP) {
        // plane row rn is the producers' block row rn + 62
        const int need = (rn + 63 < cx.prod_nrows) ? rn + 63 : cx.prod_nrows;
        coop_wait_producer(cx, cx.prod_first, need, known_pa);
        coop_wait_producer(cx, cx.prod_last, need, known_pb);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = lane + 64 * u;
        if (COOP) inj[u] = (idx <= PB) ? __hip_atomic_load(plane_in + (size_t)rn * si + kin + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
        else inj[u] = (idx <= PB) ? plane_in[(size_t)rn * si + kin + idx] : 0.0f;
      }
      {
        const float bnd_left = lds[3];
        prev_init = (lane == 0) ? bnd_left : prev_init;
      }
      // lane 0 only: LDS -> its row registers
#pragma unroll
      for (int g = 0; g < PB; g += 8)
        asm volatile(
            "s_mov_b64 exec, 1\n\t"
            "ds_read_b32 %0, %8 offset:%9\n\t"
            "ds_read_b32 %1, %8 offset:%10\n\t"
            "ds_read_b32 %2, %8 offset:%11\n\t"
            "ds_read_b32 %3, %8 offset:%12\n\t"
            "ds_read_b32 %4, %8 offset:%13\n\t"
            "ds_read_b32 %5, %8 offset:%14\n\t"
            "ds_read_b32 %6, %8 offset:%15\n\t"
            "ds_read_b32 %7, %8 offset:%16\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(row[g]), "+v"(row[g + 1]), "+v"(row[g + 2]), "+v"(row[g + 3]),
              "+v"(row[g + 4]), "+v"(row[g + 5]), "+v"(row[g + 6]), "+v"(row[g + 7])
            : "v"(lds_base), "n"(16 + 4 * g), "n"(20 + 4 * g), "n"(24 + 4 * g), "n"(28 + 4 * g),
              "n"(32 + 4 * g), "n"(36 + 4 * g), "n"(40 + 4 * g), "n"(44 + 4 * g)
            : "memory");
    }
    MINRMS_WAIT_LGKM();                         // the staged plane row is in LDS
    float t, mpa, mq;
    uint32_t bits = 0;
    mpa = c0 + prev_init;
    asm volatile("s_nop 1");                    // VALU write -> DPP read distance
    const float *Cnext = Crow + ((r + 1 < nrows) ? N2 : 0);
#pragma unroll
    for (int w = 0; w < PB / 16; ++w) {
      if (w + 1 < PB / 16) {
        sload16(cnn, Crow + 1 + 16 * (w + 1));
      } else {
        sload16(cnn, Cnext + 1);
        sload1(c0n, Cnext);
      }
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const int jj = 16 * w + q;
        if constexpr (BP) {
          if (jj == 0) {
            MINRMS_DP_CELL4(mpa, cn[0], cn[1], cn[2], cn[3], row[0], row[1], row[2], row[3], linit);
          } else {
            MINRMS_DP_CELL4(mpa, cn[q], cn[q + 1], cn[q + 2], cn[q + 3], row[jj], row[jj + 1],
                            row[jj + 2], row[jj + 3], row[jj - 1]);
          }
        } else {
          if (jj == 0) {
            MINRMS_DP_CELL4_NB(mpa, cn[0], cn[1], cn[2], cn[3], row[0], row[1], row[2], row[3], linit);
          } else {
            MINRMS_DP_CELL4_NB(mpa, cn[q], cn[q + 1], cn[q + 2], cn[q + 3], row[jj], row[jj + 1],
                               row[jj + 2], row[jj + 3], row[jj - 1]);
          }
        }
      }
      if constexpr (BP) {
        bprow[((size_t)r * bp_row_words + w) * 64] = bits;
        bits = 0;
      }
      MINRMS_WAIT_SLOADS(cnn, c0n);
      cn = cnn;
    }
    c0 = c0n;
    if (has_right) {
      if (COOP) {
        // ring slot r % RING was last read by the right neighbour for row r - RING
        if (r >= MINRMS_COOP_RING) coop_wait_cached(cx, cx.right_task, r - MINRMS_COOP_RING + 1, known_right);
        cx.edge_ring[(cx.task * MINRMS_COOP_RING + (r % MINRMS_COOP_RING)) * 64 + lane] = MINRMS_ROW(PB - 1);
      } else {
        ob.edges[(size_t)r * 64 + lane] = MINRMS_ROW(PB - 1);
      }
    }
    if (!COOP && has_next && r >= 62) {
      // Lane 63's finished row is row r-62 of the next block's plane.  PB stores
      // of one lane each kept the vector-memory pipe of the CU as busy as the
      // cells keep the VALUs (13-18 % of the sweep, measured by leaving them
      // out), so the row goes through LDS -- lane 63 writes it there, two
      // values per instruction -- and leaves in stores of 64 lanes.
      lds_f32 *out = lds + (PB + 8);
      const uint32_t out_base = (uint32_t)(uintptr_t)out;
#pragma unroll
      for (int g = 0; g < PB; g += 8)
        asm volatile(
            "s_mov_b32 exec_lo, 0\n\t"
            "s_mov_b32 exec_hi, 0x80000000\n\t"
            "ds_write2_b32 %8, %0, %1 offset0:%9 offset1:%10\n\t"
            "ds_write2_b32 %8, %2, %3 offset0:%11 offset1:%12\n\t"
            "ds_write2_b32 %8, %4, %5 offset0:%13 offset1:%14\n\t"
            "ds_write2_b32 %8, %6, %7 offset0:%15 offset1:%16\n\t"
            "s_mov_b64 exec, -1"
            :
            : "v"(row[g]), "v"(row[g + 1]), "v"(row[g + 2]), "v"(row[g + 3]),
              "v"(row[g + 4]), "v"(row[g + 5]), "v"(row[g + 6]), "v"(row[g + 7]), "v"(out_base),
              "n"(g), "n"(g + 1), "n"(g + 2), "n"(g + 3), "n"(g + 4), "n"(g + 5), "n"(g + 6), "n"(g + 7)
            : "memory");
      float *dst = plane_out + (size_t)(r - 62) * so + p * PB;
#pragma unroll
      for (int u = 0; u < (PB + 63) / 64; ++u) {
        const int idx = lane + 64 * u;
        if (idx < PB) dst[idx] = out[idx];      // LDS operations of a wave execute in order
      }
    }
    if (COOP && has_next && r >= 62) {
      // lane 63 only: its finished row is row r-62 of the next block's plane
      const float *dst = plane_out + (size_t)(r - 62) * so + p * PB;
#define MINRMS_PLANE_STORE8(SC)                                                       \
        asm volatile(                                                                 \
            "s_mov_b32 exec_lo, 0\n\t"                                                \
            "s_mov_b32 exec_hi, 0x80000000\n\t"                                       \
            "global_store_dword %8, %0, off offset:%9" SC "\n\t"                      \
            "global_store_dword %8, %1, off offset:%10" SC "\n\t"                     \
            "global_store_dword %8, %2, off offset:%11" SC "\n\t"                     \
            "global_store_dword %8, %3, off offset:%12" SC "\n\t"                     \
            "global_store_dword %8, %4, off offset:%13" SC "\n\t"                     \
            "global_store_dword %8, %5, off offset:%14" SC "\n\t"                     \
            "global_store_dword %8, %6, off offset:%15" SC "\n\t"                     \
            "global_store_dword %8, %7, off offset:%16" SC "\n\t"                     \
            "s_mov_b64 exec, -1"                                                      \
            :                                                                         \
            : "v"(row[g]), "v"(row[g + 1]), "v"(row[g + 2]), "v"(row[g + 3]),         \
              "v"(row[g + 4]), "v"(row[g + 5]), "v"(row[g + 6]), "v"(row[g + 7]), "v"(dst), \
              "n"(4 * g), "n"(4 + 4 * g), "n"(8 + 4 * g), "n"(12 + 4 * g),            \
              "n"(16 + 4 * g), "n"(20 + 4 * g), "n"(24 + 4 * g), "n"(28 + 4 * g)      \
            : "memory")
      if (PUBLISH) {
        // write-through 16-byte stores: a 4-byte sc1 store is a fabric write of its own
#pragma unroll
        for (int g = 0; g < PB; g += 4) {
          f4v q4 = {row[g], row[g + 1], row[g + 2], row[g + 3]};
          asm volatile(
              "s_mov_b32 exec_lo, 0\n\t"
              "s_mov_b32 exec_hi, 0x80000000\n\t"
              "global_store_dwordx4 %1, %0, off offset:%2 sc1\n\t"
              "s_mov_b64 exec, -1"
              :
              : "v"(q4), "v"(dst), "n"(4 * g)
              : "memory");
        }
      } else {
#pragma unroll
        for (int g = 0; g < PB; g += 8) MINRMS_PLANE_STORE8("");
      }
#undef MINRMS_PLANE_STORE8
      if (COOP && !PUBLISH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // plane row is in L2 before it is announced
    }
    if (COOP) {
      MINRMS_WAIT_LGKM();                       // edge value is in LDS before it is announced
      if (lane == 0) cx.prog[cx.task] = r + 1;
      if constexpr (PUBLISH) {
        // Announce to the other workgroup one row late instead of draining the
        // write-through stores of every row: vector-memory operations complete in
        // issue order, and a row issues U loads (if it injects), PB/16 back-pointer
        // stores and PB/4 plane stores, so once all but that many are done
        // everything rows 0..r-1 stored is done.  (Before row 62 a row issues
        // fewer and the wait covers less, but there is no plane row to announce
        // yet: the consumer's first need is 63.)  The last row drains.
        constexpr int ops_row = (INJECT ? U : 0) + PB / 16 + PB / 4;
        static_assert(ops_row < 63, "vmcnt is a 6-bit counter");
        const bool last = r + 1 == nrows;
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_row) : "memory");
        if (lane == 0)
          __hip_atomic_store(cx.gprog + cx.gtask, cx.gbase + (last ? nrows : r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    Crow = Cnext;
  }

  if constexpr (!COOP && MINRMS_C_PREFETCH)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no load may write to LDS after the workgroup has given it back
  // cost(n, N1, N2) sits in the last row, column N2 (dyn3d.cc:385)
  const int jl = N2 - jstart;
  if (jl >= 0 && jl < PB) {
    // row[jl] with a wave-uniform jl: one scalar compare and one select per
    // column, through vcc (written as a compare of every jj with jl the
    // compiler first builds all PB lane masks in scalar registers, more than
    // there are, and parks them in the lanes of a vector register)
    float cv = 0.0f;
#pragma unroll
    for (int jj = 0; jj < PB; ++jj)
      asm volatile("s_cmp_eq_u32 %2, %3\n\t"
                   "s_cselect_b64 vcc, -1, 0\n\t"
                   "v_cndmask_b32 %0, %0, %1, vcc"
                   : "+v"(cv) : "v"(MINRMS_ROW(jj)), "s"(jl), "n"(jj) : "vcc", "scc");
    const int n = MINRMS_LAYERS_PER_BLOCK * b + lane;
    if (lane > 0 && n <= L.n_max) ob.cost[n - 1] = cv;
  }
}

// One launch = one (layer block, panel) of every orientation of the batch:
// one wave (= one 64-thread workgroup) per orientation.  Layer blocks and
// panels are separate, ordered launches on one stream, so the hand-over plane
// and edge column written by one launch are visible to the next.
template <int PB, bool INJECT, bool HAS_NEXT, bool BP = true>
__global__ void __launch_bounds__(64)
k_dp_panel(const DpLayout L, const float *C, uint32_t *bp, float *planes, float *edges,
           float *cost, const int b, const int p)
{
  __shared__ __attribute__((aligned(16))) float lds_mem[2 * (MINRMS_MAX_PB + 8) + 192];   // plane row in, plane row out, prefetch sink
  lds_f32 *lds = (lds_f32 *)lds_mem;
  const int lane = threadIdx.x;
  const size_t o = blockIdx.x;
  OrientationBuffers ob;
  ob.C = C + o * L.c_floats;
  ob.bp = bp + o * L.bp_words;
  ob.planes = planes + o * L.plane_floats;
  ob.edges = edges + o * L.edge_floats;
  ob.cost = cost + o * L.n_max;
  CoopCtx none = {};
  dp_panel<PB, INJECT, HAS_NEXT, false, false, BP>(L, ob, b, p, lane, lds, none);
}

// Cooperative launch: one wave per (block, panel) task, tasks numbered
// block-major; the tasks of one orientation form one workgroup, or G.n_groups
// workgroups of whole layer blocks.  Grid = count * n_groups, the workgroups of
// group 0 first.  Latency of one orientation drops from the sum of all panels to
// about one panel column plus pipeline fill; used for the small batches of the
// refinement rounds and for re-solving the winners.
// gprog: [count][G.total_tasks] row counters (n_groups > 1 only).  A counter
// holds gbase + rows; gbase = (launch number) << MINRMS_COOP_ROW_BITS grows from
// launch to launch, so whatever earlier launches left there reads as "no row
// yet" and nothing has to clear the array in between.
template <int PB>
__global__ void __launch_bounds__(1024)
k_dp_coop(const DpLayout L, const CoopGroups G, const float *C, uint32_t *bp, float *planes, float *cost,
          int *err, int *gprog, const int gbase, const int *live_count)
{
  const int count_max = gridDim.x / G.n_groups;
  const int o_idx = blockIdx.x % count_max, grp = blockIdx.x / count_max;
  // live_count: number of orientations that are live in this launch when the
  // grid was sized for the maximum (device-driven refinement rounds)
  if (live_count && o_idx >= *live_count) return;
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int n_tasks = blockDim.x >> 6;
  lds_vint *prog = (lds_vint *)smem;                         // [16] + abort flag at [16]
  lds_f32 *edge_ring = (lds_f32 *)(smem + 256);
  lds_f32 *stage = edge_ring + n_tasks * MINRMS_COOP_RING * 64;
  const int lane = threadIdx.x & 63;
  const int task = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, local to the group
  if (threadIdx.x <= MINRMS_COOP_MAX_TASKS) prog[threadIdx.x] = 0;
  // A negative limit (test hook) starts the workgroup in the state a wait that
  // ran out leaves behind -- abort flag and error word raised -- so that the
  // give-up path is taken deterministically: no wait blocks, every task runs to
  // the end of its rows, the host must discard the batch.
  if (G.spin_limit < 0 && threadIdx.x == MINRMS_COOP_MAX_TASKS) { prog[MINRMS_COOP_MAX_TASKS] = 1; *err = 1; }
  __syncthreads();

  const int b_first = G.first_block[grp], b_end = G.first_block[grp + 1];
  int b = b_first, first = 0;                                 // first task of block b
  while (b + 1 < b_end && task >= first + L.blk[b].P) { first += L.blk[b].P; ++b; }
  const int p = task - first;
  if (task >= first + L.blk[b].P) return;                     // spare wave

  const size_t o = (size_t)o_idx;
  OrientationBuffers ob;
  ob.C = C + o * L.c_floats;
  ob.bp = bp + o * L.bp_words;
  ob.planes = planes + o * L.plane_floats;
  ob.edges = nullptr;
  ob.cost = cost + o * L.n_max;

  CoopCtx cx;
  cx.prog = prog;
  cx.abort_flag = prog + MINRMS_COOP_MAX_TASKS;
  cx.edge_ring = edge_ring;
  cx.task = task;
  cx.left_task = task - 1;
  cx.right_task = task + 1;
  cx.err = err;
  cx.gprog = gprog + o * G.total_tasks;
  cx.gtask = G.first_task[grp] + task;
  cx.gbase = gbase;
  cx.spin_limit = G.spin_limit;
  cx.prod_first = cx.prod_last = -1;
  cx.prod_nrows = 0;
  cx.prod_global = 0;
  if (b > 0) {
    const int Pp = L.blk[b - 1].P;
    cx.prod_global = (b == b_first);
    // the block above: local task numbers, or numbers in gprog if it belongs to the group before
    const int prev_first = cx.prod_global ? G.first_task[grp] - Pp : first - Pp;
    int pa = (62 + p * PB) / PB, pb = (62 + p * PB + PB) / PB;
    if (pa > Pp - 1) pa = Pp - 1;
    if (pb > Pp - 1) pb = Pp - 1;
    cx.prod_first = prev_first + pa;
    cx.prod_last = prev_first + pb;
    cx.prod_nrows = L.blk[b - 1].nrows;
  }
  lds_f32 *lds = stage + task * (PB + 8);
  const bool inject = b > 0, has_next = b + 1 < L.NB;
  const bool publish = has_next && b + 1 == b_end;            // the next block runs in another workgroup
  // Waves of one SIMD (w, w+4, w+8, ...) are served oldest first: left alone,
  // the first panels of a block run ahead as far as the edge ring lets them and
  // finish early, and the last panels, which everything downstream waits for,
  // do the rest of their rows with the SIMDs nearly empty.  A later panel
  // therefore gets the higher issue priority: it runs whenever its left
  // neighbour has a row for it, and the earlier panels fill the gaps.
  {
    const int pr = task >> 2;
    if (pr == 1) __builtin_amdgcn_s_setprio(1);
    else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    else if (pr >= 3) __builtin_amdgcn_s_setprio(3);
  }
  if (inject) {
    if (publish)       dp_panel<PB, true, true, true, true>(L, ob, b, p, lane, lds, cx);
    else if (has_next) dp_panel<PB, true, true, true>(L, ob, b, p, lane, lds, cx);
    else               dp_panel<PB, true, false, true>(L, ob, b, p, lane, lds, cx);
  } else {
    if (publish)       dp_panel<PB, false, true, true, true>(L, ob, b, p, lane, lds, cx);
    else if (has_next) dp_panel<PB, false, true, true>(L, ob, b, p, lane, lds, cx);
    else               dp_panel<PB, false, false, true>(L, ob, b, p, lane, lds, cx);
  }
}

}  // namespace dp
