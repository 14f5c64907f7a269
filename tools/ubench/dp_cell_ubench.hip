// Micro-benchmark of the DP cell instruction stream (tools only, not product).
// Variants isolate what each instruction class costs per wave-instruction.
// Every wave stamps its row loop with s_memtime (shader cycles) and
// s_memrealtime (100 MHz): "cyc/instr/wave" is the median wave's own cycle
// count per instruction, independent of the clock the chip holds; "clock" is
// the clock that median wave saw.  cyc/instr/SIMD = cyc/instr/wave divided by
// the waves that share a SIMD: the issue rate of the SIMD as a whole.
// Build: hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -packed-fp32-ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

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

// eight independent v_add_f32 / v_fma_f32 per "cell" (the guide's plain-VALU row)
#define EIGHT_ADD(A) \
  asm volatile("v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %8, %1\n\tv_add_f32 %2, %8, %2\n\tv_add_f32 %3, %8, %3\n\t" \
               "v_add_f32 %4, %8, %4\n\tv_add_f32 %5, %8, %5\n\tv_add_f32 %6, %8, %6\n\tv_add_f32 %7, %8, %7" \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]) : "s"(cn[0]))
#define EIGHT_FMA(A) \
  asm volatile("v_fma_f32 %0, %8, %0, %0\n\tv_fma_f32 %1, %8, %1, %1\n\tv_fma_f32 %2, %8, %2, %2\n\tv_fma_f32 %3, %8, %3, %3\n\t" \
               "v_fma_f32 %4, %8, %4, %4\n\tv_fma_f32 %5, %8, %5, %5\n\tv_fma_f32 %6, %8, %6, %6\n\tv_fma_f32 %7, %8, %7, %7" \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]) : "s"(cn[0]))
// eight independent v_min_f32_dpp wave_shr:1 / v_cmp_eq_f32 (vcc) / v_addc_co_u32 (vcc in and out): one class each
#define EIGHT_DPP(A) \
  asm volatile("v_min_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %1, %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %2, %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %3, %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %4, %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %5, %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %6, %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
               "v_min_f32_dpp %7, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), "+v"(A[7]))
#define EIGHT_CMP(A) \
  asm volatile("v_cmp_eq_f32 vcc, %0, %1\n\tv_cmp_eq_f32 vcc, %1, %2\n\tv_cmp_eq_f32 vcc, %2, %3\n\tv_cmp_eq_f32 vcc, %3, %4\n\t" \
               "v_cmp_eq_f32 vcc, %4, %5\n\tv_cmp_eq_f32 vcc, %5, %6\n\tv_cmp_eq_f32 vcc, %6, %7\n\tv_cmp_eq_f32 vcc, %7, %0" \
               :: "v"(A[0]), "v"(A[1]), "v"(A[2]), "v"(A[3]), "v"(A[4]), "v"(A[5]), "v"(A[6]), "v"(A[7]) : "vcc")
#define EIGHT_ADDC(B) \
  asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc\n\t" \
               "v_addc_co_u32 %2, vcc, %2, %2, vcc\n\tv_addc_co_u32 %3, vcc, %3, %3, vcc\n\t" \
               "v_addc_co_u32 %4, vcc, %4, %4, vcc\n\tv_addc_co_u32 %5, vcc, %5, %5, vcc\n\t" \
               "v_addc_co_u32 %6, vcc, %6, %6, vcc\n\tv_addc_co_u32 %7, vcc, %7, %7, vcc" \
               : "+v"(B[0]), "+v"(B[1]), "+v"(B[2]), "+v"(B[3]), "+v"(B[4]), "+v"(B[5]), "+v"(B[6]), "+v"(B[7]) :: "vcc")

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
__global__ void __launch_bounds__(64) k(const float* C, int N2, int nrows, uint32_t* out, float* fout, uint64_t* stamps)
{
  const int lane = threadIdx.x;
  float row[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) row[j] = lane == 0 ? 0.f : __builtin_inff();
  const float* Crow = C + (size_t)blockIdx.x * 64;
  uint32_t acc = 0;
  f16v cn;
  for (int q = 0; q < 16; ++q) cn[q] = 1.0f + q;
  float ind[8]; uint32_t indb[8];
  for (int q = 0; q < 8; ++q) { ind[q] = (float)(lane + q); indb[q] = lane + q; }
  const uint64_t t_start = __builtin_amdgcn_s_memtime(), rt_start = __builtin_amdgcn_s_memrealtime();
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
        if (VAR == 6) { EIGHT_ADD(ind); }
        if (VAR == 7) { EIGHT_FMA(ind); }
        if (VAR == 8) { EIGHT_DPP(ind); }
        if (VAR == 9) { EIGHT_CMP(ind); }
        if (VAR == 10) { EIGHT_ADDC(indb); }
      }
      if (VAR == 5) { bits = (bitsA << 16) | bitsB; bitsA = bitsB = 0; }
      acc ^= bits;
    }
    Crow += N2;
  }
  const uint64_t t_end = __builtin_amdgcn_s_memtime(), rt_end = __builtin_amdgcn_s_memrealtime();
  if (lane == 0) { stamps[2 * blockIdx.x] = t_end - t_start; stamps[2 * blockIdx.x + 1] = rt_end - rt_start; }
  float s = 0; 
  for (int q = 0; q < 8; ++q) { s += ind[q]; acc ^= indb[q]; }
#pragma unroll
  for (int j = 0; j < PB; ++j) s += row[j];
  out[blockIdx.x * 64 + lane] = acc; fout[blockIdx.x * 64 + lane] = s;
}

static uint64_t* g_stamps;

template <int PB, int VAR, bool SLOAD>
void run(const char* name, double instr_per_cell, const float* C, int N2, uint32_t* out, float* fout)
{
  const int nrows = 2000;
  for (int waves : {256, 256 * 4, 256 * 8, 256 * 16, 256 * 20}) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<PB, VAR, SLOAD><<<waves, 64>>>(C, N2, nrows, out, fout, g_stamps);
    hipEventRecord(a);
    k<PB, VAR, SLOAD><<<waves, 64>>>(C, N2, nrows, out, fout, g_stamps);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<uint64_t> st(2 * (size_t)waves);
    hipMemcpy(st.data(), g_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<std::pair<uint64_t, uint64_t>> w(waves);
    for (int q = 0; q < waves; ++q) w[q] = {st[2 * q], st[2 * q + 1]};
    std::sort(w.begin(), w.end());
    const auto med = w[waves / 2];
    double instr = (double)nrows * PB * instr_per_cell;      // per wave
    double wps = waves / 1024.0;                               // waves per SIMD
    double clock_ghz = med.second ? (double)med.first / ((double)med.second * 10.0) : 0.0;   // realtime ticks are 10 ns
    printf("%-34s PB=%-3d waves/SIMD=%-4.2f %8.3f ms  in-kernel: %6.2f cyc/instr/wave %6.2f cyc/instr/SIMD  clock %.2f GHz"
           "  | wall at 2.4 GHz: %6.2f cyc/instr/SIMD\n", name, PB, wps, ms,
           med.first / instr, med.first / instr / (wps < 1 ? 1 : wps), clock_ghz,
           ms * 1e-3 * 2.4e9 / (instr * (wps < 1 ? 1 : wps)));
  }
}

int main()
{
  const int N2 = 4096;
  float* C; uint32_t* out; float* fout;
  hipMalloc(&C, (size_t)N2 * 2100 * 4 + (1 << 22)); hipMemset(C, 0, (size_t)N2 * 2100 * 4 + (1 << 22));
  hipMalloc(&out, 4 << 20); hipMalloc(&fout, 4 << 20); hipMalloc(&g_stamps, 1 << 20);
  // plain VALU classes, independent instructions: what one wave-instruction costs the SIMD
  run<16, 6, false>("8 independent v_add_f32 (s,v)", 2.0, C, N2, out, fout);
  run<16, 7, false>("8 independent v_fma_f32", 2.0, C, N2, out, fout);
  run<16, 8, false>("8 v_min_f32_dpp wave_shr:1", 2.0, C, N2, out, fout);
  run<16, 9, false>("8 v_cmp_eq_f32 -> vcc", 2.0, C, N2, out, fout);
  run<16, 10, false>("8 v_addc_co_u32 vcc->vcc", 2.0, C, N2, out, fout);
  // the DP cell sequence
  run<80, 0, false>("cell (7 instr), PB=80", 7, C, N2, out, fout);
  run<80, 5, false>("cell, 4 cells interleaved, PB=80", 7, C, N2, out, fout);
  run<80, 1, false>("cell, no dpp, PB=80", 7, C, N2, out, fout);
  run<80, 3, false>("cell, row_shr for wave_shr, PB=80", 7, C, N2, out, fout);
  run<80, 2, false>("add+min_dpp+min only, PB=80", 3, C, N2, out, fout);
  run<80, 4, false>("cell without the 2 addc, PB=80", 5, C, N2, out, fout);
  run<80, 0, true>("cell, sload+wait per 16 cols, PB=80", 7, C, N2, out, fout);
  run<16, 0, false>("cell, PB=16", 7, C, N2, out, fout);
  run<32, 0, false>("cell, PB=32", 7, C, N2, out, fout);
  run<160, 0, false>("cell, PB=160", 7, C, N2, out, fout);
  return 0;
}
