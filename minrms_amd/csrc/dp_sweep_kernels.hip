// dp_sweep_kernels.hip -- the instantiations of dp::k_dp_panel (dp_kernel.h):
// one (layer block, panel) of every orientation of a batch per launch, the
// sweep's form of Dyn3d::InitTable_IncrN / Solve_IncrN (lib/dyn/dyn3d.cc:135-387).
//
// Compiled three times (-DMINRMS_SWEEP_PART=0, 1, 2), each time with a third of
// the panel widths: the unrolled rows of up to 192 cells make this by far the
// longest compile of the library, and the parts build side by side.  Part 0
// also holds the dispatch.
#include "engine_internal.h"
#include "dp_kernel.h"

#ifndef MINRMS_SWEEP_PART
#error "compile with -DMINRMS_SWEEP_PART=0, 1 or 2 (minrms_amd/build.py does)"
#endif

template <int PB, bool BP>
static dp_kernel_fn pick_flags2(bool inject, bool has_next)
{
  if (inject) return has_next ? dp::k_dp_panel<PB, true, true, BP> : dp::k_dp_panel<PB, true, false, BP>;
  return has_next ? dp::k_dp_panel<PB, false, true, BP> : dp::k_dp_panel<PB, false, false, BP>;
}

template <int PB>
static dp_kernel_fn pick_flags(bool inject, bool has_next, bool bp)
{
  return bp ? pick_flags2<PB, true>(inject, has_next) : pick_flags2<PB, false>(inject, has_next);
}

dp_kernel_fn pick_dp_kernel_narrow(int PB, bool inject, bool has_next, bool bp);   // part 0: 32 .. 96 columns
dp_kernel_fn pick_dp_kernel_medium(int PB, bool inject, bool has_next, bool bp);   // part 1: 112 .. 144
dp_kernel_fn pick_dp_kernel_wide(int PB, bool inject, bool has_next, bool bp);     // part 2: 160 .. 192
hipError_t touch_sweep_medium();
hipError_t touch_sweep_wide();

#if MINRMS_SWEEP_PART == 0
dp_kernel_fn pick_dp_kernel_narrow(int PB, bool inject, bool has_next, bool bp)
{
  switch (PB) {
    case 32: return pick_flags<32>(inject, has_next, bp);
    case 48: return pick_flags<48>(inject, has_next, bp);
    case 64: return pick_flags<64>(inject, has_next, bp);
    case 80: return pick_flags<80>(inject, has_next, bp);
    case 96: return pick_flags<96>(inject, has_next, bp);
  }
  return nullptr;
}

dp_kernel_fn pick_dp_kernel(int PB, bool inject, bool has_next, bool bp)
{
  if (PB <= 96) return pick_dp_kernel_narrow(PB, inject, has_next, bp);
  if (PB <= 144) return pick_dp_kernel_medium(PB, inject, has_next, bp);
  return pick_dp_kernel_wide(PB, inject, has_next, bp);
}

hipError_t touch_unit_dp_sweep()
{
  hipFuncAttributes attr;
  hipError_t r = hipFuncGetAttributes(&attr, (const void *)dp::k_dp_panel<32, false, false, true>);
  if (r == hipSuccess) r = touch_sweep_medium();
  if (r == hipSuccess) r = touch_sweep_wide();
  return r;
}
#elif MINRMS_SWEEP_PART == 1
dp_kernel_fn pick_dp_kernel_medium(int PB, bool inject, bool has_next, bool bp)
{
  switch (PB) {
    case 112: return pick_flags<112>(inject, has_next, bp);
    case 128: return pick_flags<128>(inject, has_next, bp);
    case 144: return pick_flags<144>(inject, has_next, bp);
  }
  return nullptr;
}

hipError_t touch_sweep_medium()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)dp::k_dp_panel<112, false, false, true>);
}
#else
dp_kernel_fn pick_dp_kernel_wide(int PB, bool inject, bool has_next, bool bp)
{
  switch (PB) {
    case 160: return pick_flags<160>(inject, has_next, bp);
    case 176: return pick_flags<176>(inject, has_next, bp);
    case 192: return pick_flags<192>(inject, has_next, bp);
  }
  return nullptr;
}

hipError_t touch_sweep_wide()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)dp::k_dp_panel<160, false, false, true>);
}
#endif
