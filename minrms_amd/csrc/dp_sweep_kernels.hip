// dp_sweep_kernels.hip -- the instantiations of dp::k_dp_panel (dp_kernel.h):
// one (layer block, panel) of every orientation of a batch per launch, the
// sweep's form of Dyn3d::InitTable_IncrN / Solve_IncrN (lib/dyn/dyn3d.cc:135-387).
#include "engine_internal.h"
#include "dp_kernel.h"

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

dp_kernel_fn pick_dp_kernel(int PB, bool inject, bool has_next, bool bp)
{
  switch (PB) {
    case 32: return pick_flags<32>(inject, has_next, bp);
    case 48: return pick_flags<48>(inject, has_next, bp);
    case 64: return pick_flags<64>(inject, has_next, bp);
    case 80: return pick_flags<80>(inject, has_next, bp);
    case 96: return pick_flags<96>(inject, has_next, bp);
    case 112: return pick_flags<112>(inject, has_next, bp);
    case 128: return pick_flags<128>(inject, has_next, bp);
    case 144: return pick_flags<144>(inject, has_next, bp);
    case 160: return pick_flags<160>(inject, has_next, bp);
    case 176: return pick_flags<176>(inject, has_next, bp);
    case 192: return pick_flags<192>(inject, has_next, bp);
  }
  return nullptr;
}

hipError_t touch_unit_dp_sweep()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)dp::k_dp_panel<32, false, false, true>);
}
