// Micro-benchmark of the DP cell instruction stream (tools only, not product).
// Variants isolate what each instruction class costs per wave-instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef float f16v __attribute__((ext_vector_type(16)));

#define ONE_FULL(MPI, MPO, CN, RJ, RJM1) \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t" \
  "v_min_f32_dpp %[t], " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  "v_cmp_eq_f32 vcc, %[t], " RJ "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t" \
  "v_min_f32 " RJ ", %[t], " RJM1 "\n\t" \
  "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t"
#define ONE_NODPP(MPI, MPO, CN, RJ, RJM1) \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t" \
  "v_min_f32 %[t], " MPI ", " RJ "\n\t" \
  "v_cmp_eq_f32 vcc, %[t], " RJ "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t" \
  "v_min_f32 " RJ ", %[t], " RJM1 "\n\t" \
  "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t"
#define ONE_NOBITS(MPI, MPO, CN, RJ, RJM1) \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t" \
  "v_min_f32_dpp %[t], " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  "v_min_f32 " RJ ", %[t], " RJM1 "\n\t"
#define ONE_ROWSHR(MPI, MPO, CN, RJ, RJM1) \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t" \
  "v_min_f32_dpp %[t], " MPI ", " RJ " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  "v_cmp_eq_f32 vcc, %[t], " RJ "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t" \
  "v_min_f32 " RJ ", %[t], " RJM1 "\n\t" \
  "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t" \
  "v_addc_co_u32 %[bits], vcc, %[bits], %[bits], vcc\n\t"
// compare into SGPR pairs + cndmask packing instead of vcc/addc
#define ONE_NOADDC(MPI, MPO, CN, RJ, RJM1) \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t" \
  "v_min_f32_dpp %[t], " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  "v_cmp_eq_f32 vcc, %[t], " RJ "\n\t" \
  "v_min_f32 " RJ ", %[t], " RJM1 "\n\t" \
  "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t"

#define CELL4_(ONE, MP, C0, C1, C2, C3, R0, R1, R2, R3, RM1) \
  asm volatile( \
      ONE("%[mp]", "%[mq]", "%[c0]", "%[r0]", "%[rm1]") \
      ONE("%[mq]", "%[mp]", "%[c1]", "%[r1]", "%[r0]") \
      ONE("%[mp]", "%[mq]", "%[c2]", "%[r2]", "%[r1]") \
      ONE("%[mq]", "%[mp]", "%[c3]", "%[r3]", "%[r2]") \
      : [mp] "+v"(MP), [mq] "=&v"(mq), [t] "=&v"(t), [bits] "+v"(bits), \
        [r0] "+v"(R0), [r1] "+v"(R1), [r2] "+v"(R2), [r3] "+v"(R3) \
      : [rm1] "v"(RM1), [c0] "s"(C0), [c1] "s"(C1), [c2] "s"(C2), [c3] "s"(C3) \
      : "vcc")

// Four cells with the instructions of different cells interleaved so that no
// instruction uses the result of the one before it: compares go to SGPR pairs,
// the two flags of a cell to separate accumulators (bA: up <= match, bB: left <= both).
#define CELL4_SCHED(MP, C0, C1, C2, C3, R0, R1, R2, R3, RM1) \
  asm volatile( \
      "v_min_f32_dpp %[t0], %[mp], %[r0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_add_f32 %[m1], %[c0], %[r0]\n\t" \
      "v_add_f32 %[m2], %[c1], %[r1]\n\t" \
      "v_add_f32 %[m3], %[c2], %[r2]\n\t" \
      "v_add_f32 %[mp], %[c3], %[r3]\n\t" \
      "v_min_f32_dpp %[t1], %[m1], %[r1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_min_f32_dpp %[t2], %[m2], %[r2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_min_f32_dpp %[t3], %[m3], %[r3] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
      "v_cmp_eq_f32_e64 %[p0], %[t0], %[r0]\n\t" \
      "v_cmp_eq_f32_e64 %[p1], %[t1], %[r1]\n\t" \
      "v_cmp_eq_f32_e64 %[p2], %[t2], %[r2]\n\t" \
      "v_cmp_eq_f32_e64 %[p3], %[t3], %[r3]\n\t" \
      "v_min_f32 %[r0], %[t0], %[rm1]\n\t" \
      "v_addc_co_u32_e64 %[ba], vcc, %[ba], %[ba], %[p0]\n\t" \
      "v_min_f32 %[r1], %[t1], %[r0]\n\t" \
      "v_addc_co_u32_e64 %[ba], vcc, %[ba], %[ba], %[p1]\n\t" \
      "v_min_f32 %[r2], %[t2], %[r1]\n\t" \
      "v_addc_co_u32_e64 %[ba], vcc, %[ba], %[ba], %[p2]\n\t" \
      "v_min_f32 %[r3], %[t3], %[r2]\n\t" \
      "v_addc_co_u32_e64 %[ba], vcc, %[ba], %[ba], %[p3]\n\t" \
      "v_cmp_eq_f32_e64 %[p0], %[r0], %[rm1]\n\t" \
      "v_cmp_eq_f32_e64 %[p1], %[r1], %[r0]\n\t" \
      "v_addc_co_u32_e64 %[bb], vcc, %[bb], %[bb], %[p0]\n\t" \
      "v_cmp_eq_f32_e64 %[p2], %[r2], %[r1]\n\t" \
      "v_addc_co_u32_e64 %[bb], vcc, %[bb], %[bb], %[p1]\n\t" \
      "v_cmp_eq_f32_e64 %[p3], %[r3], %[r2]\n\t" \
      "v_addc_co_u32_e64 %[bb], vcc, %[bb], %[bb], %[p2]\n\t" \
      "v_addc_co_u32_e64 %[bb], vcc, %[bb], %[bb], %[p3]\n\t" \
      : [mp] "+v"(MP), [m1] "=&v"(m1), [m2] "=&v"(m2), [m3] "=&v"(m3), [t0] "=&v"(t0), [t1] "=&v"(t1), \
        [t2] "=&v"(t2), [t3] "=&v"(t3), [ba] "+v"(bitsA), [bb] "+v"(bitsB), \
        [r0] "+v"(R0), [r1] "+v"(R1), [r2] "+v"(R2), [r3] "+v"(R3), \
        [p0] "=&s"(p0), [p1] "=&s"(p1), [p2] "=&s"(p2), [p3] "=&s"(p3) \
      : [rm1] "v"(RM1), [c0] "s"(C0), [c1] "s"(C1), [c2] "s"(C2), [c3] "s"(C3) \
      : "vcc")

template <int PB, int VAR, bool SLOAD>
__global__ void __launch_bounds__(64) k(const float* C, int N2, int nrows, uint32_t* out, float* fout)
{
  const int lane = threadIdx.x;
  float row[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) row[j] = lane == 0 ? 0.f : __builtin_inff();
  const float* Crow = C + (size_t)blockIdx.x * 64;
  uint32_t acc = 0;
  f16v cn;
  for (int q = 0; q < 16; ++q) cn[q] = 1.0f + q;
  for (int r = 0; r < nrows; ++r) {
    float t, mq, mpa = 1.0f; uint32_t bits = 0;
    float m1, m2, m3, t0, t1, t2, t3; uint32_t bitsA = 0, bitsB = 0; uint64_t p0, p1, p2, p3;
    const float linit = row[PB-1];
#pragma unroll
    for (int w = 0; w < PB / 16; ++w) {
      if (SLOAD) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(cn) : "s"(Crow + 16 * w) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int q = 0; q < 16; q += 4) {
        const int jj = 16 * w + q;
#define CELL4(...) CELL4_(__VA_ARGS__)
#define ARGS mpa, cn[q], cn[q+1], cn[q+2], cn[q+3], row[jj], row[jj+1], row[jj+2], row[jj+3]
        if (VAR == 0) { if (jj == 0) CELL4(ONE_FULL, ARGS, linit); else CELL4(ONE_FULL, ARGS, row[jj-1]); }
        if (VAR == 1) { if (jj == 0) CELL4(ONE_NODPP, ARGS, linit); else CELL4(ONE_NODPP, ARGS, row[jj-1]); }
        if (VAR == 2) { if (jj == 0) CELL4(ONE_NOBITS, ARGS, linit); else CELL4(ONE_NOBITS, ARGS, row[jj-1]); }
        if (VAR == 3) { if (jj == 0) CELL4(ONE_ROWSHR, ARGS, linit); else CELL4(ONE_ROWSHR, ARGS, row[jj-1]); }
        if (VAR == 5) {
          if (jj == 0) CELL4_SCHED(mpa, cn[q], cn[q+1], cn[q+2], cn[q+3], row[jj], row[jj+1], row[jj+2], row[jj+3], linit);
          else CELL4_SCHED(mpa, cn[q], cn[q+1], cn[q+2], cn[q+3], row[jj], row[jj+1], row[jj+2], row[jj+3], row[jj-1]);
        }
        if (VAR == 4) { if (jj == 0) CELL4(ONE_NOADDC, ARGS, linit); else CELL4(ONE_NOADDC, ARGS, row[jj-1]); }
      }
      if (VAR == 5) { bits = (bitsA << 16) | bitsB; bitsA = bitsB = 0; }
      acc ^= bits;
    }
    Crow += N2;
  }
  float s = 0; 
#pragma unroll
  for (int j = 0; j < PB; ++j) s += row[j];
  out[blockIdx.x * 64 + lane] = acc; fout[blockIdx.x * 64 + lane] = s;
}

template <int PB, int VAR, bool SLOAD>
void run(const char* name, int instr_per_cell, const float* C, int N2, uint32_t* out, float* fout)
{
  const int nrows = 2000;
  for (int waves : {256, 256 * 4, 256 * 8, 256 * 16}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PB, VAR, SLOAD><<<waves, 64>>>(C, N2, nrows, out, fout);
    hipEventRecord(a);
    k<PB, VAR, SLOAD><<<waves, 64>>>(C, N2, nrows, out, fout);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr = (double)nrows * PB * instr_per_cell;      // per wave
    double wps = waves / 1024.0;                               // waves per SIMD
    double cyc = ms * 1e-3 * 2.4e9;                            // at nominal 2.4 GHz
    printf("%-34s PB=%d waves/SIMD=%.0f  %.3f ms  %.2f cyc/instr/wave  %.2f cyc/instr/SIMD\n", name, PB, wps, ms,
           cyc / instr, cyc / (instr * wps));
  }
}

int main()
{
  const int N2 = 4096;
  float* C; uint32_t* out; float* fout;
  hipMalloc(&C, (size_t)N2 * 2100 * 4 + (1 << 22)); hipMemset(C, 0, (size_t)N2 * 2100 * 4 + (1 << 22));
  hipMalloc(&out, 4 << 20); hipMalloc(&fout, 4 << 20);
  run<32, 0, false>("full, PB=32", 7, C, N2, out, fout);
  run<32, 5, false>("interleaved schedule, PB=32", 7, C, N2, out, fout);
  run<16, 0, false>("full, PB=16", 7, C, N2, out, fout);
  run<16, 5, false>("interleaved schedule, PB=16", 7, C, N2, out, fout);
  run<80, 0, false>("full, PB=80", 7, C, N2, out, fout);
  run<80, 5, false>("interleaved schedule, PB=80", 7, C, N2, out, fout);
  run<160, 0, false>("full, no sload", 7, C, N2, out, fout);
  run<160, 0, true>("full, sload+wait each chunk", 7, C, N2, out, fout);
  run<160, 1, false>("no dpp", 7, C, N2, out, fout);
  run<160, 3, false>("row_shr instead of wave_shr", 7, C, N2, out, fout);
  run<160, 2, false>("add+min_dpp+min only", 3, C, N2, out, fout);
  run<160, 4, false>("no addc (5 instr)", 5, C, N2, out, fout);
  run<96, 0, false>("full, PB=96", 7, C, N2, out, fout);
  return 0;
}
