// orientations.hip -- the orientation list of fragment matching
// (lib/or_gen/or_gen.cc:237-587, shuffled as :179) -> minrms_engine_gen_orientations.
#include "engine_internal.h"
#include "fit_device.h"

// Fragment-pair superpositions (lib/or_gen/or_gen.cc:503-544): one thread per
// allowed (I,J); fixed = fragment of structure 1, moved = fragment of 2.
__global__ void __launch_bounds__(64)
k_fragment_fit(const int nbb, const int frag_len, const int64_t n_pairs,
               const int32_t *__restrict__ IJ, const float *__restrict__ xyzA,
               const float *__restrict__ xyzB, float *__restrict__ og, float *__restrict__ rmsd)
{
  const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (t >= n_pairs) return;
  const float *fa = xyzA + 3 * (size_t)IJ[2 * t] * nbb;
  const float *fb = xyzB + 3 * (size_t)IJ[2 * t + 1] * nbb;
  const int npts = frag_len * nbb;
  fit::Superimpose S;
  S.begin();
  for (int a = 0; a < npts; ++a) S.sum(fa + 3 * a, fb + 3 * a);
  S.centre(npts);
  for (int a = 0; a < npts; ++a) S.accumulate(fa + 3 * a, fb + 3 * a);
  float xf[12];
  rmsd[t] = S.finish(npts, xf);
  for (int q = 0; q < 12; ++q) og[12 * t + q] = xf[q];
}

// ---------------------------------------------------------------------------
// orientation generation
// ---------------------------------------------------------------------------
namespace {
// glibc rand() in its default state (srand(1)): additive feedback generator of
// degree 31, separation 3, 310 values discarded after seeding.
struct GlibcRand {
  int32_t r[31];
  int f = 3, b = 0;
  GlibcRand()
  {
    int32_t word = 1;
    r[0] = word;
    for (int i = 1; i < 31; ++i) {
      const int64_t hi = word / 127773, lo = word % 127773;
      int64_t w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      r[i] = word;
    }
    for (int i = 0; i < 310; ++i) next();
  }
  int32_t next()
  {
    const uint32_t v = (uint32_t)r[f] + (uint32_t)r[b];
    r[f] = (int32_t)v;
    f = (f + 1) % 31;
    b = (b + 1) % 31;
    return (int32_t)(v >> 1);
  }
};
}  // namespace

extern "C" int minrms_engine_gen_orientations(minrms_engine *e, int32_t frag_len, float cutoff_rmsd,
                                              int32_t n_int, const int32_t *ip, int32_t shuffle,
                                              float *out_og, int64_t cap, int64_t *count)
{
  if (!e || !count || (n_int > 0 && !ip)) return fail(MINRMS_E_ARG, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int N1 = e->cfg.N1, N2 = e->cfg.N2, nbb = e->cfg.num_bb, L = frag_len;
  if (L < 1 || L > N1 || L > N2) return fail(MINRMS_E_ARG, "ERROR: fragment-size cannot exceed length of sequence!");
  const int w1 = N1 - e->cfg.n_min + 1, w2 = N2 - e->cfg.n_min + 1;   // or_gen.cc:253-254
  std::vector<unsigned char> mask((size_t)N1 * N2, 0);
  for (int p = 0; p < n_int; ++p) {
    const int s1 = ip[4 * p], e1 = ip[4 * p + 1], s2 = ip[4 * p + 2], e2 = ip[4 * p + 3];
    if (e1 < s1 || e2 < s2) return fail(MINRMS_E_ARG, "Subset-Match Interval cannot have start > end.");
    if (s1 < 0 || s2 < 0) return fail(MINRMS_E_ARG, "Subset-Match Interval cannot begin before start of sequence.");
    if (e1 >= N1 || e2 >= N2) return fail(MINRMS_E_ARG, "ERROR: fragment-size cannot exceed length of sequence!");
    if (s1 > e1 - L + 1 || s2 > e2 - L + 1) continue;                 // interval too small: skipped
    for (int I = s1; I <= e1 - L + 1; ++I) {
      const int jlo = std::max(s2, I - (w1 - 1));                     // or_gen.cc:451-464
      const int jhi = std::min(I + (w2 - 1), e2 - L + 1);
      for (int J = jlo; J <= jhi; ++J) mask[(size_t)I * N2 + J] = 1;
    }
  }
  std::vector<int32_t> IJ;
  for (int I = 0; I < N1; ++I)
    for (int J = 0; J < N2; ++J)
      if (mask[(size_t)I * N2 + J]) { IJ.push_back(I); IJ.push_back(J); }
  const int64_t np = (int64_t)IJ.size() / 2;
  std::vector<float> og((size_t)np * 12), rmsd((size_t)np);
  if (np > 0) {
    // staging for the largest possible list (every (I,J)), taken once per engine
    if (!e->d_gen) {
      const size_t want = (size_t)64 * N1 * N2 + 256;
      size_t got = 0, clean = 0;
      void *ptr = nullptr;
      HIP_TRY(pool::take(e->cfg.device, "d_gen", want, &ptr, &got, &clean));
      e->dev_blocks.push_back(minrms_engine::DevBlock{ptr, got, "d_gen", clean});
      e->d_gen = (unsigned char *)ptr;
    }
    const size_t cells = (size_t)N1 * N2;
    float *d_o = (float *)e->d_gen;                          // [cells][12]
    int32_t *d_IJ = (int32_t *)(e->d_gen + 48 * cells);      // [cells][2]
    float *d_r = (float *)(e->d_gen + 56 * cells);           // [cells]
    HIP_TRY(hipMemcpyAsync(d_IJ, IJ.data(), 8 * np, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_fragment_fit, dim3((unsigned)((np + 63) / 64)), dim3(64), 0, e->stream, nbb, L, np,
                       d_IJ, e->d_xyzA, e->d_xyzB, d_o, d_r);
    HIP_TRY(hipMemcpyAsync(og.data(), d_o, 48 * np, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(rmsd.data(), d_r, 4 * np, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  std::vector<int64_t> keep;
  for (int64_t t = 0; t < np; ++t)
    if (cutoff_rmsd < 0.0f || rmsd[t] <= cutoff_rmsd) keep.push_back(t);   // or_gen.cc:537-541
  if (keep.empty())
    return fail(MINRMS_E_NOSOLUTION,
                "Error: You have requested to generate orientations by fitting together blocks of %d "
                "residues each, however there are no valid regions from which to draw them.", L);
  if (shuffle) {
    // std::random_shuffle as libstdc++ implements it, driven by glibc rand()
    GlibcRand rng;
    for (int64_t i = 1; i < (int64_t)keep.size(); ++i) {
      const int64_t j = (int64_t)rng.next() % (i + 1);
      if (i != j) std::swap(keep[i], keep[j]);
    }
  }
  *count = (int64_t)keep.size();
  if (out_og)
    for (int64_t t = 0; t < std::min<int64_t>(cap, (int64_t)keep.size()); ++t)
      memcpy(out_og + 12 * t, &og[12 * (size_t)keep[t]], 48);
  return MINRMS_OK;
}

hipError_t touch_unit_orientations()
{
  hipFuncAttributes attr;
  return hipFuncGetAttributes(&attr, (const void *)k_fragment_fit);
}
