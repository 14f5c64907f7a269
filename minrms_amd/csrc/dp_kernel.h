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

struct OrientationBuffers {
  const float *C;     // [N1][N2] squared distances (+ tail padding)
  uint32_t *bp;       // back-pointer words
  float *planes;      // hand-over planes between layer blocks
  float *edges;       // [nrows][64] hand-over column between panels
  float *cost;        // [n_max] cost(n,N1,N2)
};

#define MINRMS_WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ f16v sload16(const float *p)
{
  f16v v;
  asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ float sload1(const float *p)
{
  float v;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory");
  return v;
}

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

#define MINRMS_ROW(j) row[j]

// One panel (PB columns) of one layer block, all rows.
template <int PB, bool INJECT, bool HAS_NEXT>
__device__ __forceinline__ void dp_panel(const DpLayout &L, const OrientationBuffers &ob,
                                         const int b, const int p, const int lane,
                                         float *lds)
{
  const BlockDesc &bd = L.blk[b];
  const float INF = __builtin_inff();
  const int N2 = L.N2;
  const int jstart = bd.j0 + p * PB;            // 1-based column of cell 0
  const int nrows = bd.nrows;
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
  if (inject) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = lane + 64 * u;
      inj[u] = (idx <= PB) ? plane_in[kin + idx] : 0.0f;
    }
  }

  const float *Crow = ob.C + (size_t)(bd.j0 - 1) * N2 + (jstart - 1);
  uint32_t *bprow = ob.bp + bd.bp_off + ((size_t)p * (PB / 16)) * 64 + lane;
  float edge_prev = base_init;                  // left panel's value of the previous row

  float c0 = sload1(Crow);
  f16v cn = sload16(Crow + 1);

  for (int r = 0; r < nrows; ++r) {
    float linit = base_init, prev_init = base_init;
    if (has_left) {
      linit = ob.edges[(size_t)r * 64 + lane];
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
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = lane + 64 * u;
        inj[u] = (idx <= PB) ? plane_in[(size_t)rn * si + kin + idx] : 0.0f;
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
    MINRMS_WAIT_LGKM();                         // c0, cn (and LDS) have landed
    float t, mpa, mq;
    uint32_t bits = 0;
    mpa = c0 + prev_init;
    asm volatile("s_nop 1");                    // VALU write -> DPP read distance
    const float *Cnext = Crow + ((r + 1 < nrows) ? N2 : 0);
#pragma unroll
    for (int w = 0; w < PB / 16; ++w) {
      f16v cnn;
      if (w + 1 < PB / 16) {
        cnn = sload16(Crow + 1 + 16 * (w + 1));
      } else {
        cnn = sload16(Cnext + 1);
        c0 = sload1(Cnext);
      }
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const int jj = 16 * w + q;
        if (jj == 0) {
          MINRMS_DP_CELL4(mpa, cn[0], cn[1], cn[2], cn[3], row[0], row[1], row[2], row[3], linit);
        } else {
          MINRMS_DP_CELL4(mpa, cn[q], cn[q + 1], cn[q + 2], cn[q + 3], row[jj], row[jj + 1],
                          row[jj + 2], row[jj + 3], row[jj - 1]);
        }
      }
      bprow[((size_t)r * bd.W + w) * 64] = bits;
      bits = 0;
      MINRMS_WAIT_LGKM();
      cn = cnn;
    }
    if (has_right) ob.edges[(size_t)r * 64 + lane] = MINRMS_ROW(PB - 1);
    if (has_next && r >= 62) {
      // lane 63 only: its finished row is row r-62 of the next block's plane
      const float *dst = plane_out + (size_t)(r - 62) * so + p * PB;
#pragma unroll
      for (int g = 0; g < PB; g += 8)
        asm volatile(
            "s_mov_b32 exec_lo, 0\n\t"
            "s_mov_b32 exec_hi, 0x80000000\n\t"
            "global_store_dword %8, %0, off offset:%9\n\t"
            "global_store_dword %8, %1, off offset:%10\n\t"
            "global_store_dword %8, %2, off offset:%11\n\t"
            "global_store_dword %8, %3, off offset:%12\n\t"
            "global_store_dword %8, %4, off offset:%13\n\t"
            "global_store_dword %8, %5, off offset:%14\n\t"
            "global_store_dword %8, %6, off offset:%15\n\t"
            "global_store_dword %8, %7, off offset:%16\n\t"
            "s_mov_b64 exec, -1"
            :
            : "v"(row[g]), "v"(row[g + 1]), "v"(row[g + 2]), "v"(row[g + 3]),
              "v"(row[g + 4]), "v"(row[g + 5]), "v"(row[g + 6]), "v"(row[g + 7]), "v"(dst),
              "n"(4 * g), "n"(4 + 4 * g), "n"(8 + 4 * g), "n"(12 + 4 * g),
              "n"(16 + 4 * g), "n"(20 + 4 * g), "n"(24 + 4 * g), "n"(28 + 4 * g)
            : "memory");
    }
    Crow = Cnext;
  }

  // cost(n, N1, N2) sits in the last row, column N2 (dyn3d.cc:385)
  const int jl = N2 - jstart;
  if (jl >= 0 && jl < PB) {
    float cv = 0.0f;
#pragma unroll
    for (int jj = 0; jj < PB; ++jj)
      if (jj == jl) cv = MINRMS_ROW(jj);
    const int n = MINRMS_LAYERS_PER_BLOCK * b + lane;
    if (lane > 0 && n <= L.n_max) ob.cost[n - 1] = cv;
  }
}

// One launch = one (layer block, panel) of every orientation of the batch:
// one wave (= one 64-thread workgroup) per orientation.  Layer blocks and
// panels are separate, ordered launches on one stream, so the hand-over plane
// and edge column written by one launch are visible to the next.
template <int PB, bool INJECT, bool HAS_NEXT>
__global__ void __launch_bounds__(64)
k_dp_panel(const DpLayout L, const float *C, uint32_t *bp, float *planes, float *edges,
           float *cost, const int b, const int p)
{
  __shared__ __attribute__((aligned(16))) float lds[MINRMS_MAX_PB + 8];
  const int lane = threadIdx.x;
  const size_t o = blockIdx.x;
  OrientationBuffers ob;
  ob.C = C + o * L.c_floats;
  ob.bp = bp + o * L.bp_words;
  ob.planes = planes + o * L.plane_floats;
  ob.edges = edges + o * L.edge_floats;
  ob.cost = cost + o * L.n_max;
  dp_panel<PB, INJECT, HAS_NEXT>(L, ob, b, p, lane, lds);
}

}  // namespace dp
