// dp_coop_kernels.hip -- the instantiations of dp::k_dp_coop (dp_kernel.h,
// dp_task_asm.h): the same DP (lib/dyn/dyn3d.cc:135-387) for small batches, all
// tasks of an orientation running at once and handing rows to each other.
#include "engine_internal.h"
#include "dp_kernel.h"

// The cooperative kernel of several workgroups asks for more LDS than half a CU
// holds (one workgroup per CU): raise the limit of its instantiations.  Device
// must be current.
int allow_large_lds()
{
  const void *kernels[9] = {(const void *)dp::k_dp_coop<16, false>, (const void *)dp::k_dp_coop<32, false>,
                            (const void *)dp::k_dp_coop<64, false>, (const void *)dp::k_dp_coop<16, true>,
                            (const void *)dp::k_dp_coop<32, true>, (const void *)dp::k_dp_coop<64, true>,
                            (const void *)dp::k_dp_coop<8, true, true>, (const void *)dp::k_dp_coop<16, true, true>,
                            (const void *)dp::k_dp_coop<32, true, true>};
  for (const void *k : kernels)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return fail(MINRMS_E_DEVICE, "hipFuncSetAttribute(dynamic LDS) failed");
  return 0;
}

// Cooperative DP launch for small batches (see dp::k_dp_coop).
int launch_coop(minrms_engine *e, int64_t count, const int *live_count)
{
  // few orientations: a SIMD per task (segments), the extra workgroups run on CUs that would be idle
  const bool seg = e->coop_seg.n_groups > 1 && count <= e->coop_seg_batch && e->d_xedge != nullptr;
  const dp::CoopGroups &G = seg ? e->coop_seg : e->coop_groups;
  const int waves = G.max_tasks + G.fetch_waves;
  const int pbw = seg ? e->coop_seg_pb : e->coop_pb;         // columns per task
  const DpLayout &Lk = seg ? e->Lc_seg : e->Lc;
  size_t shmem = 512 + (size_t)waves * MINRMS_COOP_RING * 64 * 4 + (size_t)G.max_tasks * (pbw + 8) * 4;
  if (seg) shmem += (size_t)MINRMS_COOP_PRING * (G.max_tasks * pbw + 4) * 4 +     // the plane ring
                    (size_t)G.max_tasks * MINRMS_COOP_ORING * pbw * 4;             // the out ring
  // workgroups that wait for each other across CUs: one per CU (more LDS than
  // half a CU has), the setting the hand-over was measured in
  if (G.n_groups > 1) shmem = std::max<size_t>(shmem, 81 * 1024);
  const dim3 block(64 * waves), grid((unsigned)(count * G.n_groups));
  MINRMS_MARK(e, live_count ? 2 : 1, MINRMS_KC_COOP);        // with a live count: a refinement round
  int gbase = 0;
  if (G.n_groups > 1) {
    // the launch number tags the row counters and granules; when it runs out of bits the arrays start from zero again
    if (++e->coop_launches >= (1 << (31 - MINRMS_COOP_ROW_BITS))) {
      // both arrays as a whole, whatever spread this launch uses: the row counters belong to the
      // spread over whole blocks (coop_groups), the granules to the segments (coop_seg)
      if (e->d_gprog && e->gprog_bytes) HIP_TRY(hipMemsetAsync(e->d_gprog, 0, e->gprog_bytes, e->stream));
      if (e->d_xedge && e->xedge_bytes) HIP_TRY(hipMemsetAsync(e->d_xedge, 0, e->xedge_bytes, e->stream));
      e->coop_launches = 1;
    }
    gbase = e->coop_launches << MINRMS_COOP_ROW_BITS;
  }
#define MINRMS_LAUNCH_COOP(PBV, SEGV, ASMV)                                                        \
  hipLaunchKernelGGL((dp::k_dp_coop<PBV, SEGV, ASMV>), grid, block, shmem, e->stream, Lk, G, e->d_C, e->d_bp,  \
                     e->d_planes, e->d_cost, e->d_err, e->d_gprog, e->d_xedge, gbase, live_count)
  if (seg) {
    // the hand-written row loop (8-column tasks have no other), or the compiled one (MINRMS_COOP_ASM=0, 64 columns)
    const bool hand = G.asm_rows != 0 || pbw == 8;
    if (pbw == 8) MINRMS_LAUNCH_COOP(8, true, true);
    else if (pbw == 16) { if (hand) MINRMS_LAUNCH_COOP(16, true, true); else MINRMS_LAUNCH_COOP(16, true, false); }
    else if (pbw == 32) { if (hand) MINRMS_LAUNCH_COOP(32, true, true); else MINRMS_LAUNCH_COOP(32, true, false); }
    else MINRMS_LAUNCH_COOP(64, true, false);
  } else {
    if (e->coop_pb == 16) MINRMS_LAUNCH_COOP(16, false, false);
    else if (e->coop_pb == 32) MINRMS_LAUNCH_COOP(32, false, false);
    else MINRMS_LAUNCH_COOP(64, false, false);
  }
#undef MINRMS_LAUNCH_COOP
  return 0;
}

hipError_t touch_unit_dp_coop()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)dp::k_dp_coop<16, false>);
}

#ifdef MINRMS_COOP_STAMPS
// diagnostic build only: the time stamps of the last cooperative launch (dp::g_coop_times)
extern "C" int minrms_engine_debug_read_coop_times(minrms_engine *e, unsigned int *out1536)
{
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpyFromSymbol(out1536, HIP_SYMBOL(dp::g_coop_times), 4 * 4 * 384));
  return 0;
}
extern "C" int minrms_engine_debug_read_fetch_trace(minrms_engine *e, unsigned int *out1536)
{
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpyFromSymbol(out1536, HIP_SYMBOL(dp::g_fetch_trace), 4 * 64 * 8 * 3));
  return 0;
}
#endif
