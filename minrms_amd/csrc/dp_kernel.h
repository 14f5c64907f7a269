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
#include "dp_task_asm.h"

namespace dp {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
// LDS pointers carry their address space so that every access is a DS
// instruction (in-order with the hand-written ds_read below), never FLAT.
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) volatile int lds_vint;
typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

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
struct CoopCtx {
  lds_vint *prog;
  lds_vint *abort_flag;
  lds_f32 *edge_ring;               // [wave][MINRMS_COOP_RING][64]
  int task, left_task, right_task;
  int prod_first, prod_last;        // tasks of block b-1 whose columns this panel reads
  int prod_nrows;
  int prod_global;                  // the producers run in another workgroup: their ids index gprog
  int gtask;                        // this task's index in gprog
  int *gprog;                       // global: gbase + rows finished, per task of this orientation
  int gbase;                        // tag of this launch; values below it are left over from earlier launches
  int *err;                         // global: set when a wait gave up
  int spin_limit;
  uint32_t sink;                    // LDS byte address of 256 bytes nobody reads (destination of warm-up loads)
  // segments: the plane rows arrive in an LDS ring filled by the workgroup's
  // plane-fetch wave (prog slot plane_task), and leave as granules
  int plane_task;
  lds_f32 *pring;                   // [MINRMS_COOP_PRING][pring_rs]; this task's "one left of cell 0" is column task*PB
  int pring_rs;
  int pub_task;                     // prog slot of the publish wave: plane rows it has stored
  lds_f32 *oring;                   // the out ring [task][MINRMS_COOP_ORING][PB]
  unsigned long long *xout;         // the segment to the right runs in another workgroup: this task's last column goes here
#ifdef MINRMS_COOP_STAMPS
  int *stamp;                       // diagnostic build: six words of this task (where it ran, cycles per section), or NULL
  unsigned int *times;              // diagnostic build: this wave's entry of g_coop_times, or NULL
  int trace_group;                  // diagnostic build: the workgroup's number
#endif
};
#ifdef MINRMS_COOP_STAMPS
// diagnostic build: per task of orientation 0 (and per helper wave: 128 + 4 * group + role), the
// 100-MHz clock when the wave entered the kernel, [1] began its rows, [2] ended them; [3] tasks: how
// often each wait left the straight path; fetch waves: [1] polls | empty polls << 16, [3] time spent
// waiting for the loads | for room in the ring << 16
static __device__ unsigned int g_coop_times[4 * 384];
// ... and of every edge-fetch wave of orientation 0, each time 32 more rows have come in: the clock, the rows it has
// handed on, the rows its consumer has finished
static __device__ unsigned int g_fetch_trace[64 * 8 * 3];
#define MINRMS_STAMP(CX, K) do { if ((CX).stamp && lane == 0) (CX).stamp[K] = (int)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MINRMS_STAMP(CX, K) do { } while (0)
#endif

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
// GRAN (cooperative, segments): planes come from the LDS ring and go out as granules.
template <int PB, bool INJECT, bool HAS_NEXT, bool COOP, bool PUBLISH = false, bool BP = true, bool GRAN = false>
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
  // sweep kernels: a last panel whose last 8 columns lie outside the table stops 8 cells early in every row
  const bool tail8 = !COOP && bd.tail8 != 0 && !has_right;
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
  if (inject && !GRAN) {
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

  if constexpr (COOP) {
    // Warm the L2 with this task's columns of the distance table, all rows at
    // once: the table was written by another kernel, on any XCD, so every row's
    // scalar load would otherwise come from memory, and with one group of 16
    // columns per row (a task that has a SIMD to itself) there is nothing to
    // hide that behind.  One lane per 64 bytes touched; the data goes to a piece
    // of LDS nobody reads (no register, no wait).
    constexpr int row_bytes = (PB + 1) * 4, per_row = (row_bytes + 63) / 64 + 1;
    const int total = nrows * per_row;
    uint32_t m0_saved;
    for (int base = 0; base < total; base += 64) {
      int idx = base + lane;
      idx = idx < total ? idx : total - 1;
      const int prow = idx / per_row, k = idx - prow * per_row;
      const uint32_t off = (uint32_t)prow * (uint32_t)N2 * 4u + (uint32_t)(64 * k < row_bytes - 4 ? 64 * k : row_bytes - 4);
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %3\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dword %1, %2 offset:0\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "v"(off), "s"(Crow), "s"(cx.sink) : "memory");
    }
  }
  // What a row needs from its neighbours (the left panel's column value, the
  // plane row) sits in LDS behind a progress counter.  Asking at the top of the
  // row costs two LDS latencies in a row of which nothing can be hidden, so
  // counter and data of the NEXT row are read right after the cells of this one
  // (LDS operations of a wave execute in order: a counter that says "there"
  // vouches for the data read after it); if the neighbour was not that far yet,
  // the top of the next row asks again the slow way.
  int spec_left_prog = 0, spec_plane_prog = 0;
  float spec_linit = 0.0f, spec_bnd = 0.0f;
#ifdef MINRMS_COOP_STAMPS
  const unsigned long long stamp_cycles0 = __builtin_amdgcn_s_memtime();
  unsigned long long sec_t = stamp_cycles0;
  unsigned int sec_acc[4] = {0, 0, 0, 0};
#define MINRMS_SECTION(K) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sec_acc[K] += (unsigned int)(now_ - sec_t); sec_t = now_; } while (0)
#else
#define MINRMS_SECTION(K) do { } while (0)
#endif
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
        const int seen = r > 0 ? __builtin_amdgcn_readfirstlane(spec_left_prog) : 0;
        if (seen >= r + 1) {
          known_left = seen;
          linit = spec_linit;
        } else {
          coop_wait_cached(cx, cx.left_task, r + 1, known_left);
          linit = cx.edge_ring[(cx.left_task * MINRMS_COOP_RING + (r % MINRMS_COOP_RING)) * 64 + lane];
        }
      } else {
        linit = ob.edges[(size_t)r * 64 + lane];
      }
      prev_init = edge_prev;
      edge_prev = linit;
    }
    // lane 0 only: LDS -> its row registers; BASE = byte address such that word 3 is "one left of cell 0"
#define MINRMS_INJECT_READS(BASE)                                                       \
    _Pragma("unroll")                                                                   \
    for (int g = 0; g < PB; g += 8)                                                     \
      asm volatile(                                                                     \
          "s_mov_b64 exec, 1\n\t"                                                       \
          "ds_read_b32 %0, %8 offset:%9\n\t"                                            \
          "ds_read_b32 %1, %8 offset:%10\n\t"                                           \
          "ds_read_b32 %2, %8 offset:%11\n\t"                                           \
          "ds_read_b32 %3, %8 offset:%12\n\t"                                           \
          "ds_read_b32 %4, %8 offset:%13\n\t"                                           \
          "ds_read_b32 %5, %8 offset:%14\n\t"                                           \
          "ds_read_b32 %6, %8 offset:%15\n\t"                                           \
          "ds_read_b32 %7, %8 offset:%16\n\t"                                           \
          "s_mov_b64 exec, -1"                                                          \
          : "+v"(row[g]), "+v"(row[g + 1]), "+v"(row[g + 2]), "+v"(row[g + 3]),         \
            "+v"(row[g + 4]), "+v"(row[g + 5]), "+v"(row[g + 6]), "+v"(row[g + 7])      \
          : "v"(BASE), "n"(16 + 4 * g), "n"(20 + 4 * g), "n"(24 + 4 * g), "n"(28 + 4 * g), \
            "n"(32 + 4 * g), "n"(36 + 4 * g), "n"(40 + 4 * g), "n"(44 + 4 * g)          \
          : "memory")
    if (inject) {
      // previous block's lane-63 row -> LDS -> lane 0's row registers
      float bnd_left;
      if constexpr (GRAN) {
        const int seen = r > 0 ? __builtin_amdgcn_readfirstlane(spec_plane_prog) : 0;
        if (seen >= r + 1) {
          known_pa = seen;                        // the reads issued after the previous row's cells were good
          bnd_left = spec_bnd;
        } else {
          coop_wait_cached(cx, cx.plane_task, r + 1, known_pa);
          lds_f32 *in = cx.pring + ((r % MINRMS_COOP_PRING) * cx.pring_rs + cx.task * PB);
          const uint32_t in_base = (uint32_t)(uintptr_t)in - 12u;
          bnd_left = in[0];
          MINRMS_INJECT_READS(in_base);
        }
      } else {
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
        bnd_left = lds[3];
        MINRMS_INJECT_READS(lds_base);
      }
      prev_init = (lane == 0) ? bnd_left : prev_init;
    }
    MINRMS_WAIT_LGKM();                         // the staged plane row is in LDS
    if (COOP) MINRMS_SECTION(0);                // waits for the left neighbour and the plane, injection
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
        // (wave-uniform: one scalar branch per row around the last 8 cells)
        if (!COOP && w + 1 == PB / 16 && q >= 8 && tail8) continue;
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
        // a word of 8 cells: the first cell of a word owns its highest bits
        if (!COOP && w + 1 == PB / 16 && tail8) bits <<= 16;
        bprow[((size_t)r * bp_row_words + w) * 64] = bits;
        bits = 0;
      }
      MINRMS_WAIT_SLOADS(cnn, c0n);
      cn = cnn;
    }
    c0 = c0n;
    if (COOP && r + 1 < nrows) {
      // next row's inputs, if the neighbours already have them (checked at the top of the next row)
      if (has_left) {
        spec_left_prog = cx.prog[cx.left_task];
        asm volatile("" ::: "memory");            // compiler only: the counter is read before the data it vouches for
        spec_linit = cx.edge_ring[(cx.left_task * MINRMS_COOP_RING + ((r + 1) % MINRMS_COOP_RING)) * 64 + lane];
      }
      if constexpr (GRAN && INJECT) {
        spec_plane_prog = cx.prog[cx.plane_task];
        asm volatile("" ::: "memory");            // compiler only, as above
        lds_f32 *in = cx.pring + (((r + 1) % MINRMS_COOP_PRING) * cx.pring_rs + cx.task * PB);
        const uint32_t in_base = (uint32_t)(uintptr_t)in - 12u;
        spec_bnd = in[0];
        MINRMS_INJECT_READS(in_base);
      }
    }
    if (COOP) MINRMS_SECTION(1);                // the cells, back-pointer store, wait for the next row's distances
    if (has_right) {
      if (COOP) {
        if (cx.xout) {
          // the right neighbour is the first task of another workgroup: one
          // write-through store of {value, tag}; nothing to wait for
          const unsigned long long granule =
              ((unsigned long long)(uint32_t)(cx.gbase + r + 1) << 32) | (unsigned long long)__float_as_uint(MINRMS_ROW(PB - 1));
          __hip_atomic_store(cx.xout + (size_t)r * 64 + lane, granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          // ring slot r % RING was last read by the right neighbour for row r - RING
          if (r >= MINRMS_COOP_RING) coop_wait_cached(cx, cx.right_task, r - MINRMS_COOP_RING + 1, known_right);
          cx.edge_ring[(cx.task * MINRMS_COOP_RING + (r % MINRMS_COOP_RING)) * 64 + lane] = MINRMS_ROW(PB - 1);
        }
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
    if (COOP && GRAN && has_next && r >= 62) {
      // Lane 63's finished row is row r-62 of the next block's plane, which
      // another workgroup consumes.  The task only drops the row into its slot
      // of the LDS out ring (lane 63 writes it, two values per instruction);
      // the workgroup's publish wave turns it into granules {value, tag} and
      // stores them, so the task neither waits for LDS nor issues the store.
      if (r - 62 >= MINRMS_COOP_ORING) coop_wait_cached(cx, cx.pub_task, r - 62 - MINRMS_COOP_ORING + 1, known_pb);
      const uint32_t out_base = (uint32_t)(uintptr_t)(cx.oring + (cx.task * MINRMS_COOP_ORING + (r % MINRMS_COOP_ORING)) * PB);
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
    }
    if (COOP && !GRAN && has_next && r >= 62) {
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
    if (COOP) MINRMS_SECTION(2);                // hand-over to the right, plane row out
    if (COOP) {
      // No wait between the row's LDS writes (edge value, plane row) and the
      // counter that announces them: the LDS operations of a wave are carried
      // out in the order they were issued.
      asm volatile("" ::: "memory");              // compiler only: the row's LDS writes stay in front of the counter
      if (lane == 0) cx.prog[cx.task] = r + 1;
      if constexpr (PUBLISH && !GRAN) {
        // Announce to the other workgroup LATE rows late instead of draining the
        // write-through stores of every row (one takes 1-2 us to be
        // acknowledged, several rows' worth): vector-memory operations complete in
        // issue order, and a row issues at least U loads (if it injects), PB/16
        // back-pointer stores and PB/4 plane stores, so once all but LATE times
        // that many are done everything rows 0..r-LATE stored is done.  (Before
        // row 62 a row issues fewer and the wait covers less, but there is no
        // plane row to announce yet: the consumer's first need is 63.)  The last
        // row drains.
        constexpr int ops_row = (INJECT ? U : 0) + PB / 16 + PB / 4;
        constexpr int LATE = (63 / ops_row) < 1 ? 1 : ((63 / ops_row) > 6 ? 6 : (63 / ops_row));
        static_assert(ops_row * LATE <= 63, "vmcnt is a 6-bit counter");
        const bool last = r + 1 == nrows;
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ops_row * LATE) : "memory");
        const int done = last ? nrows : (r + 1 - LATE > 0 ? r + 1 - LATE : 0);
        if (lane == 0)
          __hip_atomic_store(cx.gprog + cx.gtask, cx.gbase + done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    Crow = Cnext;
    if (COOP) MINRMS_SECTION(3);                // progress counter
  }
#undef MINRMS_INJECT_READS

#ifdef MINRMS_COOP_STAMPS
  if (COOP && cx.stamp && lane == 0) {
    cx.stamp[1] = (int)(__builtin_amdgcn_s_memtime() - stamp_cycles0);   // shader cycles of the row loop
    cx.stamp[0] = (int)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID: which SIMD / CU the wave runs on
    for (int q = 0; q < 4; ++q) cx.stamp[2 + q] = (int)sec_acc[q];
  }
#endif
  if constexpr (COOP || MINRMS_C_PREFETCH)
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
// stop_n (or NULL): per orientation, the n at which the reference's n-loop leaves
// it as far as the layers computed so far tell (n_max + 1: still running); an
// orientation that has left needs no layer block above the one that holds n.
template <int PB, bool INJECT, bool HAS_NEXT, bool BP = true>
__global__ void __launch_bounds__(64)
k_dp_panel(const DpLayout L, const float *C, uint32_t *bp, float *planes, float *edges,
           float *cost, const int b, const int p, const int32_t *stop_n)
{
  __shared__ __attribute__((aligned(16))) float lds_mem[2 * (MINRMS_MAX_PB + 8) + 192];   // plane row in, plane row out, prefetch sink
  lds_f32 *lds = (lds_f32 *)lds_mem;
  const int lane = threadIdx.x;
  const size_t o = blockIdx.x;
  if (stop_n && __builtin_amdgcn_readfirstlane(stop_n[o]) <= MINRMS_LAYERS_PER_BLOCK * b) return;
  OrientationBuffers ob;
  ob.C = C + o * L.c_floats;
  ob.bp = bp + o * L.bp_words;
  ob.planes = planes + o * L.plane_floats;
  ob.edges = edges + o * L.edge_floats;
  ob.cost = cost + o * L.n_max;
  CoopCtx none = {};
  dp_panel<PB, INJECT, HAS_NEXT, false, false, BP>(L, ob, b, p, lane, lds, none);
}

// One task of the segmented cooperative kernel, 16 or 32 columns wide, through
// the hand-written row loop of dp_task_asm.h: the same panel, the same protocol
// and the same results as dp_panel<PB, inject, has_next, true, has_next, true, true>.
template <int PB>
__device__ __forceinline__ void dp_panel_asm(const DpLayout &L, const OrientationBuffers &ob,
                                             const int b, const int p, const int lane,
                                             const CoopCtx &cx, const bool inject, const bool has_next)
{
  static_assert(PB == 8 || PB == 16 || PB == 32, "the hand-written row loop exists for 8, 16 and 32 columns");
  static_assert(MINRMS_COOP_RING == 8 && MINRMS_COOP_PRING == 16 && MINRMS_COOP_ORING == 16,
                "the ring depths are written into the masks and shifts of dp_task_asm.h");
  const BlockDesc &bd = L.blk[b];
  const int N2 = L.N2;
  const int jstart = bd.j0 + p * PB;
  const int nrows = bd.nrows;
  const bool has_left = p > 0, has_right = p + 1 < bd.P;
  const float *Crow = ob.C + (size_t)(bd.j0 - 1) * N2 + (jstart - 1);
  {
    // warm the L2 with this task's columns of the distance table (see dp_panel)
    constexpr int row_bytes = (PB + 1) * 4, per_row = (row_bytes + 63) / 64 + 1;
    const int total = nrows * per_row;
    uint32_t m0_saved;
    for (int base = 0; base < total; base += 64) {
      int idx = base + lane;
      idx = idx < total ? idx : total - 1;
      const int prow = idx / per_row, k = idx - prow * per_row;
      const uint32_t off = (uint32_t)prow * (uint32_t)N2 * 4u + (uint32_t)(64 * k < row_bytes - 4 ? 64 * k : row_bytes - 4);
      asm volatile("s_mov_b32 %0, m0\n\t"
                   "s_mov_b32 m0, %3\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dword %1, %2 offset:0\n\t"
                   "s_mov_b32 m0, %0"
                   : "=&s"(m0_saved) : "v"(off), "s"(Crow), "s"(cx.sink) : "memory");
    }
  }
  TaskAsmArgs a;
  a.crow = Crow;
  a.cstride_bytes = (uint32_t)N2 * 4u;
  a.nrows = nrows;
  a.flags = (has_left ? 1u : 0u) | (inject ? 2u : 0u) | ((has_right && !cx.xout) ? 4u : 0u) |
            ((has_right && cx.xout) ? 8u : 0u) | (has_next ? 16u : 0u);
  a.bp_addr = (uint64_t)(uintptr_t)(ob.bp + bd.bp_off + (size_t)p * (PB / 16) * 64 + lane);
  if (PB == 8)   // half a word per row: the first cell of a word has its highest bits, so even panels write the upper half
    a.bp_addr = (uint64_t)(uintptr_t)(ob.bp + bd.bp_off + (size_t)(p >> 1) * 64 + lane) + ((p & 1) ? 0u : 2u);
  a.bp_stride_bytes = (uint64_t)bd.W * 256u;
  a.xout_addr = (uint64_t)(uintptr_t)(cx.xout + lane);
  a.gbase = cx.gbase;
  const uint32_t prog_at = (uint32_t)(uintptr_t)cx.prog;
  a.p_my = prog_at + 4u * (uint32_t)cx.task;
  a.p_left = prog_at + 4u * (uint32_t)cx.left_task;
  a.p_right = prog_at + 4u * (uint32_t)cx.right_task;
  a.p_plane = prog_at + 4u * (uint32_t)cx.plane_task;
  a.p_pub = prog_at + 4u * (uint32_t)cx.pub_task;
  a.p_abort = (uint32_t)(uintptr_t)cx.abort_flag;
  const uint32_t ring_at = (uint32_t)(uintptr_t)cx.edge_ring + 4u * (uint32_t)lane;
  a.ring_in = ring_at + (uint32_t)(cx.left_task * MINRMS_COOP_RING * 256);
  a.ring_out = ring_at + (uint32_t)(cx.task * MINRMS_COOP_RING * 256);
  a.pring_at = (uint32_t)(uintptr_t)(cx.pring + cx.task * PB);
  a.pring_row_bytes = (uint32_t)cx.pring_rs * 4u;
  a.oring_at = (uint32_t)(uintptr_t)(cx.oring + cx.task * MINRMS_COOP_ORING * PB);
  a.base_init = (b == 0 && lane == 0) ? 0.0f : __builtin_inff();
  a.err = cx.err;
  a.spin_limit = cx.spin_limit;
  a.jl = N2 - jstart;
#ifdef MINRMS_COOP_STAMPS
  uint32_t slow_counts = 0;
  if (cx.times && lane == 0) cx.times[1] = (unsigned int)__builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const float cv = PB == 8 ? dp_task_rows_asm8(a, &slow_counts)
                 : PB == 16 ? dp_task_rows_asm16(a, &slow_counts) : dp_task_rows_asm32(a, &slow_counts);
  if (cx.times && lane == 0) { cx.times[2] = (unsigned int)__builtin_amdgcn_s_memrealtime(); cx.times[3] = slow_counts; }
  if (cx.stamp && lane == 0) {
    // cycles of the row loop; polls that left the straight path: left | plane << 8 | right << 16 | publish << 24
    cx.stamp[1] = (int)(__builtin_readcyclecounter() - t0);
    cx.stamp[0] = (int)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    cx.stamp[2] = (int)(slow_counts & 0xff); cx.stamp[3] = (int)((slow_counts >> 8) & 0xff);
    cx.stamp[4] = (int)((slow_counts >> 16) & 0xff); cx.stamp[5] = (int)(slow_counts >> 24);
  }
#else
  const float cv = PB == 8 ? dp_task_rows_asm8(a) : PB == 16 ? dp_task_rows_asm16(a) : dp_task_rows_asm32(a);
#endif
  // cost(n, N1, N2) sits in the last row, column N2 (dyn3d.cc:385)
  if (a.jl >= 0 && a.jl < PB) {
    const int n = MINRMS_LAYERS_PER_BLOCK * b + lane;
    if (lane > 0 && n <= L.n_max) ob.cost[n - 1] = cv;
  }
}

// The extra waves of a segment's workgroup poll granules in global memory.  Like
// every lone wave they issue about one instruction per five cycles, so what a
// poll costs beyond its round trip to memory (about 1 000 cycles) is the number
// of its instructions: the loops below are written to be short -- every load
// unconditional (lanes and rows that have nothing to fetch read a valid
// neighbour and are masked out of the test), addresses advanced by additions, a
// row's arrival tested with one compare and one mask operation.  (The first
// version of the plane fetch, with a test around every load and three loads per
// row whatever the width, came to about 400 instructions per poll of four rows:
// the tasks of the blocks above the first got their plane rows a third slower
// than they could use them, and every block ended that much later.)
#ifdef MINRMS_COOP_STAMPS
#define MINRMS_FETCH_STAMP_DECL unsigned int st_polls = 0, st_empty = 0, st_load = 0, st_ring = 0
#define MINRMS_FETCH_STAMP_A const unsigned int st_a = (unsigned int)__builtin_amdgcn_s_memrealtime()
#define MINRMS_FETCH_STAMP_B                                                          \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
  st_load += (unsigned int)__builtin_amdgcn_s_memrealtime() - st_a;                   \
  ++st_polls
#define MINRMS_FETCH_STAMP_EMPTY ++st_empty
#define MINRMS_FETCH_STAMP_RING(CALL)                                                 \
  { const unsigned int st_b = (unsigned int)__builtin_amdgcn_s_memrealtime(); CALL;   \
    st_ring += (unsigned int)__builtin_amdgcn_s_memrealtime() - st_b; }
#define MINRMS_FETCH_STAMP_OUT                                                        \
  if (cx.times && lane == 0) { cx.times[1] = st_polls | st_empty << 16; cx.times[3] = st_load | st_ring << 16; }
#else
#define MINRMS_FETCH_STAMP_DECL
#define MINRMS_FETCH_STAMP_A
#define MINRMS_FETCH_STAMP_B
#define MINRMS_FETCH_STAMP_EMPTY
#define MINRMS_FETCH_STAMP_RING(CALL) CALL
#define MINRMS_FETCH_STAMP_OUT
#endif

// how many of the four leading rows of a poll have arrived (a, b, c, d: row 0 .. 3 has)
__device__ __forceinline__ int leading_rows(const bool a, const bool b, const bool c, const bool d)
{
  return a ? (b ? (c ? (d ? 4 : 3) : 2) : 1) : 0;
}

// The extra wave of a workgroup whose first task continues a block that another
// workgroup started: brings that block's hand-over column, row by row, from the
// granules in global memory into edge-ring slot `slot`, and raises prog[slot]
// like a task that has finished rows.  One poll asks for the next four rows at
// once (one round trip to memory for up to four rows: a poll per row would cost
// more than the producer needs for a row); a granule whose tag is not this
// launch's tag for its row has not been written yet, and that row and the ones
// after it are asked for again (bounded, like every wait here).
#ifndef MINRMS_EDGE_ROWS
#define MINRMS_EDGE_ROWS 4
#endif
__device__ __forceinline__ void fetch_edge(const CoopCtx &cx, const unsigned long long *src, const int nrows,
                                           const int slot, const int consumer_task, const int lane)
{
  constexpr int NR = MINRMS_EDGE_ROWS;
  int r = 0, known = 0, spins = 0;
  const unsigned long long *at = src + lane;                  // this lane's granule of row r
  lds_f32 *ring = cx.edge_ring + slot * MINRMS_COOP_RING * 64 + lane;
  MINRMS_FETCH_STAMP_DECL;
  while (r < nrows) {
    MINRMS_FETCH_STAMP_A;
    const int last = nrows - 1 - r;                           // rows behind the last read the last row again
    unsigned long long v[NR];
#pragma unroll
    for (int d = 0; d < NR; ++d)
      v[d] = __hip_atomic_load(at + 64 * (d < last ? d : last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    MINRMS_FETCH_STAMP_B;
    const uint32_t want = (uint32_t)(cx.gbase + r + 1);
    int ready = 0;                                            // leading rows of the poll that have arrived
#pragma unroll
    for (int d = 0; d < NR; ++d)
      if (ready == d && __all((uint32_t)(v[d] >> 32) == want + d)) ready = d + 1;   // (a repeated last row never matches)
    if (ready == 0) {
      MINRMS_FETCH_STAMP_EMPTY;
      if (__builtin_amdgcn_readfirstlane(*cx.abort_flag)) {
        ready = last < NR - 1 ? last + 1 : NR;                      // given up: run through the rows on whatever is there
      } else {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > cx.spin_limit) {
          *cx.abort_flag = 1;
          *cx.err = 1;
        }
        continue;
      }
    }
    spins = 0;
    // ring slot (r + d) % RING was last read by the consumer for row r + d - RING
    if (r + ready > MINRMS_COOP_RING && known < r + ready - MINRMS_COOP_RING)
      MINRMS_FETCH_STAMP_RING(coop_wait_cached(cx, consumer_task, r + ready - MINRMS_COOP_RING, known));
#pragma unroll
    for (int d = 0; d < NR; ++d)
      if (d < ready) ring[((r + d) % MINRMS_COOP_RING) * 64] = __uint_as_float((uint32_t)v[d]);
    r += ready;
    at += 64 * ready;
    MINRMS_WAIT_LGKM();
    if (lane == 0) cx.prog[slot] = r;
#ifdef MINRMS_COOP_STAMPS
    if (cx.times && lane == 0 && (r >> 5) != ((r - ready) >> 5) && (r >> 5) <= 8 && cx.trace_group < 64) {
      unsigned int *tr = g_fetch_trace + (cx.trace_group * 8 + (r >> 5) - 1) * 3;
      tr[0] = (unsigned int)__builtin_amdgcn_s_memrealtime(); tr[1] = (unsigned int)r; tr[2] = (unsigned int)cx.prog[consumer_task];
    }
#endif
  }
  MINRMS_FETCH_STAMP_OUT;
}

// The second extra wave of a segment's workgroup (blocks above the first): brings
// the rows of the plane that feeds the segment's panels -- columns col0 ..
// col0+ncols-1 of the granule array the producers of block b-1 write, ncols <=
// 64 CH -- into the LDS plane ring, four rows per poll like fetch_edge, and raises
// prog[slot].  Columns the producers never write (beyond their last panel; they
// only feed cells outside the table) read as 0.
template <int CH>
__device__ __forceinline__ void fetch_plane(const CoopCtx &cx, const unsigned long long *src, const int gstride,
                                            const int prev_cols, const int col0, const int ncols, const int nrows,
                                            const int slot, const int last_task, const int lane)
{
  // lane l takes columns l, l + 64, ...; a lane with nothing to fetch reads column col0 (which exists: the
  // first column of a panel always has a cell of the block above diagonally behind it) and is left out of the test
  unsigned long long need_mask[CH];
  bool need[CH], writes[CH];
  uint32_t col_bytes[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = lane + 64 * c;
    need[c] = col < ncols && col0 + col < prev_cols;
    writes[c] = col < ncols;
    need_mask[c] = __ballot(need[c]);
    col_bytes[c] = need[c] ? 8u * (uint32_t)col : 0u;
  }
  int r = 0, known = 0, spins = 0;
  const unsigned char *at = (const unsigned char *)(src + col0);           // row r, column col0
  const size_t row_bytes = 8 * (size_t)gstride;
  lds_f32 *ring = cx.pring + lane;
  MINRMS_FETCH_STAMP_DECL;
  while (r < nrows) {
    MINRMS_FETCH_STAMP_A;
    const int last = nrows - 1 - r;
    unsigned long long v[4][CH];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const unsigned char *row = at + row_bytes * (size_t)(d < last ? d : last);
#pragma unroll
      for (int c = 0; c < CH; ++c)
        v[d][c] = __hip_atomic_load((const unsigned long long *)(row + col_bytes[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    MINRMS_FETCH_STAMP_B;
    const uint32_t want = (uint32_t)(cx.gbase + r + 1);
    bool ok[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      ok[d] = true;
#pragma unroll
      for (int c = 0; c < CH; ++c)
        ok[d] = ok[d] && (__ballot((uint32_t)(v[d][c] >> 32) == want + d) & need_mask[c]) == need_mask[c];
    }
    int ready = leading_rows(ok[0], ok[1], ok[2], ok[3]);
    if (ready == 0) {
      MINRMS_FETCH_STAMP_EMPTY;
      if (__builtin_amdgcn_readfirstlane(*cx.abort_flag)) {
        ready = last < 3 ? last + 1 : 4;
      } else {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > cx.spin_limit) {
          *cx.abort_flag = 1;
          *cx.err = 1;
        }
        continue;
      }
    }
    spins = 0;
    // ring row (r + d) % PRING was last read, by the segment's last task, for row r + d - PRING
    if (r + ready > MINRMS_COOP_PRING && known < r + ready - MINRMS_COOP_PRING)
      MINRMS_FETCH_STAMP_RING(coop_wait_cached(cx, last_task, r + ready - MINRMS_COOP_PRING, known));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d < ready) {
        lds_f32 *dst = ring + ((r + d) % MINRMS_COOP_PRING) * cx.pring_rs;
#pragma unroll
        for (int c = 0; c < CH; ++c)
          if (writes[c]) dst[64 * c] = need[c] ? __uint_as_float((uint32_t)v[d][c]) : 0.0f;
      }
    }
    r += ready;
    at += row_bytes * (size_t)ready;
    MINRMS_WAIT_LGKM();
    if (lane == 0) cx.prog[slot] = r;
  }
  MINRMS_FETCH_STAMP_OUT;
}

// The third extra wave of a segment's workgroup (all blocks but the last): takes
// the plane rows the workgroup's tasks drop into the LDS out ring and stores
// them as granules {value, tag = launch tag + plane row + 1}, one store per
// 64 columns, written through (sc1) so that the consuming workgroup, on any CU,
// sees value and tag together.  The segment's last task is the slowest: when it
// has finished a row, all have.
// A helper wave's wait for a task's counter: it shares its SIMD with a task, and
// every instruction it issues while it waits is one the task does not issue, so
// it sleeps between its looks at the counter from the first one on.
__device__ __forceinline__ void coop_wait_sleeping(const CoopCtx &cx, const int task, const int need, int &known)
{
  int spins = 0;
  while (known < need) {
    known = __builtin_amdgcn_readfirstlane(cx.prog[task]);
    if (known >= need) break;
    __builtin_amdgcn_s_sleep(2);
    if ((++spins & 7) == 0 && __builtin_amdgcn_readfirstlane(*cx.abort_flag)) break;
    if (spins > cx.spin_limit) {
      *cx.abort_flag = 1;
      *cx.err = 1;
    }
  }
}

template <int PB>
__device__ __forceinline__ void publish_planes(const CoopCtx &cx, unsigned long long *dst, const int gstride,
                                               const int col0, const int my_tasks, const int nrows,
                                               const int slot, const int lane)
{
  const int ncols = my_tasks * PB;
  const int chunks = (ncols + 63) >> 6;
  int known = 0;
  unsigned long long *out = dst + col0 + lane;                // plane row 0, this lane's first column
  for (int r = 62; r < nrows; ++r) {
    coop_wait_sleeping(cx, my_tasks - 1, r + 1, known);
    const unsigned long long tag = (unsigned long long)(uint32_t)(cx.gbase + (r - 62) + 1) << 32;
    const lds_f32 *in = cx.oring + (r % MINRMS_COOP_ORING) * PB;
    for (int c = 0; c < chunks; ++c) {
      const int col = lane + 64 * c;
      if (col < ncols) {
        const int t = col / PB, j = col - t * PB;
        const float v = in[t * MINRMS_COOP_ORING * PB + j];
        __hip_atomic_store(out + 64 * c, tag | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    out += gstride;
    MINRMS_WAIT_LGKM();                          // the ring slots have been read
    if (lane == 0) cx.prog[slot] = r - 62 + 1;
  }
}

// Cooperative launch: one wave per (block, panel) task, tasks numbered
// block-major; the tasks of one orientation run in G.n_groups workgroups (see
// CoopGroups).  Grid = count * n_groups, all workgroups of group 0 first, so a
// workgroup only ever waits for workgroups dispatched before it.  Latency of one
// orientation drops from the sum of all panels to about one panel column plus
// pipeline fill; used for the small batches of the refinement rounds and for
// re-solving the winners.
//
// SEG = false: groups of whole layer blocks.  Planes between blocks of different
// workgroups are plain arrays announced through gprog: [count][G.total_tasks]
// row counters, a counter holds gbase + rows.
// SEG = true: every group is a segment of one block (or one whole block): all
// traffic between workgroups is granules in xedge (hand-over columns and planes),
// moved between global memory and LDS by three extra waves (edge fetch, plane
// fetch, plane publish); no counters in global memory, no wait on vector memory
// in any task.
// gbase = (launch number) << MINRMS_COOP_ROW_BITS grows from launch to launch, so
// whatever earlier launches left in gprog and in the granules' tags reads as
// "nothing yet" and nothing has to clear the arrays in between.
// ASM (segments of 8, 16 or 32 columns): the tasks run the hand-written row loop of
// dp_task_asm.h and the kernel contains nothing else; without it, the compiled
// loop (dp_panel) -- two kernels, so that the registers the compiler needs for
// the one do not weigh on the other.
template <int PB, bool SEG, bool ASM = false>
__global__ void __launch_bounds__(1024)
k_dp_coop(const DpLayout L, const CoopGroups G, const float *C, uint32_t *bp, float *planes, float *cost,
          int *err, int *gprog, unsigned long long *xedge, const int gbase, const int *live_count)
{
  const int count_max = gridDim.x / G.n_groups;
  const int o_idx = blockIdx.x % count_max, grp = blockIdx.x / count_max;
  // live_count: number of orientations that are live in this launch when the
  // grid was sized for the maximum (device-driven refinement rounds)
  if (live_count && o_idx >= *live_count) return;
  extern __shared__ __attribute__((aligned(256))) unsigned char smem[];
  const int n_waves = blockDim.x >> 6;
  lds_vint *prog = (lds_vint *)smem;                         // [16] + abort flag at [16]; smem + 256: sink of the warm-up loads
  lds_f32 *edge_ring = (lds_f32 *)(smem + 512);
  lds_f32 *stage = edge_ring + n_waves * MINRMS_COOP_RING * 64;
  const int lane = threadIdx.x & 63;
  const int task = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, local to the group
  if (threadIdx.x <= MINRMS_COOP_MAX_TASKS) prog[threadIdx.x] = 0;
  // A negative limit (test hook) starts the workgroup in the state a wait that
  // ran out leaves behind -- abort flag and error word raised -- so that the
  // give-up path is taken deterministically: no wait blocks, every task runs to
  // the end of its rows, the host must discard the batch.
  if (G.spin_limit < 0 && threadIdx.x == MINRMS_COOP_MAX_TASKS) { prog[MINRMS_COOP_MAX_TASKS] = 1; *err = 1; }
  __syncthreads();

  const int my_tasks = G.ntask[grp];
  const bool whole = G.whole[grp] != 0;
  const bool fetches = SEG && G.xin_off[grp] >= 0;
  const size_t o = (size_t)o_idx;
  unsigned long long *granules = SEG ? xedge + o * G.xedge_granules : nullptr;

  CoopCtx cx;
  cx.prog = prog;
  cx.abort_flag = prog + MINRMS_COOP_MAX_TASKS;
  cx.edge_ring = edge_ring;
  cx.task = task;
  cx.left_task = (task == 0 && fetches) ? my_tasks : task - 1;   // the fetch wave stands in for the left neighbour
  cx.right_task = task + 1;
  cx.err = err;
  cx.gprog = gprog + o * G.total_tasks;
  cx.gtask = G.first_task[grp] + task;
  cx.gbase = gbase;
  cx.spin_limit = G.spin_limit;
  cx.sink = (uint32_t)(uintptr_t)(lds_byte_t *)smem + 256u;
  cx.prod_first = cx.prod_last = -1;
  cx.prod_nrows = 0;
  cx.prod_global = 0;
  cx.xout = nullptr;
  cx.plane_task = my_tasks + 1;
  cx.pring = stage + (n_waves - 3) * (PB + 8);               // behind the staging rows of the compute waves
  cx.pring_rs = (n_waves - 3) * PB + 4;
  cx.pub_task = my_tasks + 2;
  cx.oring = cx.pring + MINRMS_COOP_PRING * cx.pring_rs;     // behind the plane ring
#ifdef MINRMS_COOP_STAMPS
  cx.stamp = (o_idx == 0 && cx.gtask < 9) ? err + 8 + 6 * cx.gtask : nullptr;
  cx.times = nullptr;
  cx.trace_group = grp;
  if (o_idx == 0) {
    const int at = task < my_tasks ? (cx.gtask < 128 ? cx.gtask : 384) : 128 + 4 * grp + (task - my_tasks);
    if (at < 384) cx.times = g_coop_times + 4 * at;
  }
  if (cx.times && lane == 0) { cx.times[0] = (unsigned int)__builtin_amdgcn_s_memrealtime(); cx.times[1] = cx.times[2] = cx.times[3] = 0; }
#endif

  // block and panel of this task; `first` = local number of the first task of its block
  int b = G.blk[grp], p = G.pan[grp] + task, first = -G.pan[grp];

  if (task >= my_tasks) {
    if constexpr (SEG) {
      if (task == my_tasks && fetches) {
        __builtin_amdgcn_s_setprio(3);                         // it sleeps most of the time; when it has rows, it goes first
        fetch_edge(cx, granules + G.xin_off[grp], L.blk[b].nrows, my_tasks, 0, lane);
      } else if (task == my_tasks + 1 && b > 0) {
        __builtin_amdgcn_s_setprio(3);
        const unsigned long long *plane = granules + G.plane_goff[b];
        const int gs = G.plane_gstride[b], col0 = G.pan[grp] * PB + 62, ncols = my_tasks * PB + 1;   // the tasks' columns and the one left of them
        if (ncols <= 64) fetch_plane<1>(cx, plane, gs, gs, col0, ncols, L.blk[b].nrows, my_tasks + 1, my_tasks - 1, lane);
        else if (ncols <= 128) fetch_plane<2>(cx, plane, gs, gs, col0, ncols, L.blk[b].nrows, my_tasks + 1, my_tasks - 1, lane);
        else fetch_plane<3>(cx, plane, gs, gs, col0, ncols, L.blk[b].nrows, my_tasks + 1, my_tasks - 1, lane);
      } else if (task == my_tasks + 2 && b + 1 < L.NB) {
        __builtin_amdgcn_s_setprio(3);
        publish_planes<PB>(cx, granules + G.plane_goff[b + 1], G.plane_gstride[b + 1], G.pan[grp] * PB, my_tasks,
                           L.blk[b].nrows, my_tasks + 2, lane);
      }
    }
#ifdef MINRMS_COOP_STAMPS
    if (cx.times && lane == 0) cx.times[2] = (unsigned int)__builtin_amdgcn_s_memrealtime();
#endif
    return;                                                   // spare wave
  }

  if (whole)
    while (p >= L.blk[b].P) { p -= L.blk[b].P; first += L.blk[b].P; ++b; }

  OrientationBuffers ob;
  ob.C = C + o * L.c_floats;
  ob.bp = bp + o * L.bp_words;
  ob.planes = planes + o * L.plane_floats;
  ob.edges = nullptr;
  ob.cost = cost + o * L.n_max;

  lds_f32 *lds = stage + task * (PB + 8);
  const bool inject = b > 0, has_next = b + 1 < L.NB;
  if constexpr (SEG) {
    if (task == my_tasks - 1 && G.xout_off[grp] >= 0) cx.xout = granules + G.xout_off[grp];
    static_assert(!ASM || PB == 8 || PB == 16 || PB == 32, "the hand-written row loop exists for 8, 16 and 32 columns");
    static_assert(ASM || PB != 8, "8-column tasks exist with the hand-written row loop only");
    if constexpr (ASM) {
      dp_panel_asm<PB>(L, ob, b, p, lane, cx, inject, has_next);
      return;
    } else {
    if (inject) {
      if (has_next) dp_panel<PB, true, true, true, true, true, true>(L, ob, b, p, lane, lds, cx);
      else          dp_panel<PB, true, false, true, false, true, true>(L, ob, b, p, lane, lds, cx);
    } else {
      if (has_next) dp_panel<PB, false, true, true, true, true, true>(L, ob, b, p, lane, lds, cx);
      else          dp_panel<PB, false, false, true, false, true, true>(L, ob, b, p, lane, lds, cx);
    }
    return;
    }
  } else {

  if (b > 0) {
    const int Pp = L.blk[b - 1].P;
    // the block above runs in this workgroup (local task numbers) or in another (numbers in gprog)
    cx.prod_global = b == G.blk[grp];
    const int prev_first = cx.prod_global ? G.block_first_task[b - 1] : first - Pp;
    int pa = (62 + p * PB) / PB, pb = (62 + p * PB + PB) / PB;
    if (pa > Pp - 1) pa = Pp - 1;
    if (pb > Pp - 1) pb = Pp - 1;
    cx.prod_first = prev_first + pa;
    cx.prod_last = prev_first + pb;
    cx.prod_nrows = L.blk[b - 1].nrows;
  }
  const bool publish = has_next && b + 1 == G.blk[grp + 1];   // the next block runs in another workgroup
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
}

}  // namespace dp
