/* minrms_oracle.h -- CPU restatement of the MINRMS RMSD-vs-N search.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP engine is compared
 * against; it is never linked into, loaded by, or shipped with the product
 * library.  Parity status: PINNED -- every function below is checked bit for
 * bit against the real reference compiled from /root/reference with
 * `-O2 -ffp-contract=off` (oracle/Makefile, target `ref`), through the golden
 * vectors in tests/golden/ that oracle/make_golden.py generated with
 * oracle/ref_dump.cc.
 *
 * All citations are relative to the reference tree.
 */
#ifndef MINRMS_ORACLE_H
#define MINRMS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_UPPER_BOUND 1.0e26f /* lib/dyn/distance_metric.cc:49 */

typedef struct {
  int32_t N1, N2;        /* residues in structure 1 (fixed) and 2 (moved)      */
  int32_t nbb;           /* backbone atoms per residue (1 = CA only)            */
  int32_t n_min, n_max;  /* range of match counts that is recorded              */
  float   max_rmsd;      /* -1 = off   (search_or_dyn3d.cc:463-473)             */
  float   max_dist;      /* -1 = off   (distance_metric.cc:76-87)               */
  int32_t refine_while_searching; /* rank by RMSD after optimal superposition  */
  int32_t of_enable;     /* Needleman-Wunsch orientation filter (-of)           */
  float   of_max_dist;
  int32_t of_min_matches;
} orc_cfg;

/* Result tables, the restatement of aaSumSqdDists/aaTransforms/aaAlignments
 * (bin/minrms/search_or_dyn3d.h:293-310) for r = 0.  Entry k holds n = n_min+k. */
typedef struct {
  float   *ssd;        /* [range_n]              1e26f = none                   */
  float   *xform;      /* [range_n][12]          row-major 3x4                  */
  int16_t *pairs;      /* [range_n][n_max][2]    first n entries valid          */
  int64_t *winner;     /* [range_n]  o_id that set the entry; -1 none;
                                     -(2+iter) when set by refinement iter     */
} orc_tables;

/* lib/eigen/jacobi.c:7-85 (behaviour); a is destroyed; v columns = vectors.
 * returns number of sweeps used, or -1 if 50 sweeps did not converge. */
int orc_jacobi(int n, float *a /*n*n*/, float *d /*n*/, float *v /*n*n*/);

/* lib/superimpose/superimpose.cc:17-287 */
float orc_superimpose(int npts, const float *fixed, const float *moveable,
                      float xform[12]);

/* lib/vect_3d/vect_3d.h:145-155 applied to npts points */
void orc_apply_transform(const float xform[12], const float *xyz, int npts,
                         float *out);

/* libstdc++ std::random_shuffle driven by glibc rand() seeded with 1
 * (lib/or_gen/or_gen.cc:179).  perm[k] = index (in generation order) of the
 * orientation that ends up at position k. */
void orc_shuffle_perm(int64_t count, int64_t *perm);

/* lib/or_gen/or_gen.cc:237-587.  interval_pairs: n_int rows of
 * {first1,last1,first2,last2} (0-based, inclusive residue indices).
 * Writes at most cap transforms (generation order: I outer, J inner) and
 * returns the number generated (may exceed cap: call again with more room). */
int64_t orc_gen_orientations(const orc_cfg *cfg, const float *xyzA,
                             const float *xyzB, int frag_len, float cutoff_rmsd,
                             int n_int, const int32_t *interval_pairs,
                             float *out, int64_t cap);

/* One orientation: lib/dyn/dyn3d.cc:664-701 + 135-179 + 183-387.
 * xyzB_rot is structure 2 already transformed.  cost[n-1] = aOutcomes[n-1].cost
 * for n = 1..n_max.  bp (optional) is [n_max][N1][N2] uint8, 0xFF = not computed. */
void orc_dp(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot,
            float *cost, uint8_t *bp);

/* lib/dyn/dyn3d.cc:391-431 on a bp table from orc_dp */
void orc_traceback(const orc_cfg *cfg, const uint8_t *bp, int n, int16_t *pairs);

/* lib/pair_alignment/pair_alignment.cc:714-820 + search_or_dyn3d.cc:587-588:
 * returns rmsd; *ssd_reduced = SQR(rmsd)*n*nbb */
float orc_calc_min_rmsd(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                        int n, const int16_t *pairs, float xform[12],
                        float *ssd_reduced);

/* lib/or_gen/orientation_filter.cc:42-130 + lib/dyn/dynNW.cc:147-531.
 * *io_max_dist_metric is DistanceMetric's cutoff, which every call reads back
 * and re-installs (orc_clip_after_filter_call). */
int orc_good_orientation(const orc_cfg *cfg, const float *xyzA,
                         const float *xyzB_rot, float *io_max_dist_metric);
int orc_nw_num_matches(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot,
                       float clip, float gap, int min_matches);
float orc_clip_after_filter_call(const orc_cfg *cfg, float clip);

void orc_tables_init(const orc_cfg *cfg, orc_tables *t);

/* bin/minrms/search_or_dyn3d.cc:1643-1946 for o_id in [start, stop).
 * og = orientation list [.][12].  If per_on_ssd != NULL it receives, for every
 * (o_id - start, n - n_min), the ssd that was compared (NaN where the loop
 * had already broken out).  Returns the number of orientations that passed the
 * filter. */
int64_t orc_search(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                   const float *og, int64_t start, int64_t stop, orc_tables *t,
                   float *per_on_ssd);

/* orc_search plus: pass_flags[o_id - start] = 1 where the -of filter let the
 * orientation through (NULL: not wanted); *io_clip = DistanceMetric's cutoff
 * before the first / after the last orientation (NULL: start from cfg). */
int64_t orc_search_filtered(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                            const float *og, int64_t start, int64_t stop, orc_tables *t,
                            float *per_on_ssd, uint8_t *pass_flags, float *io_clip);

/* bin/minrms/search_or_dyn3d.cc:2261-2658 (refine_method = REFINE_BEST, r = 0).
 * max_iters = -1: until convergence.  Returns num_iterations as the reference
 * prints it; *num_new = number of record updates. */
int64_t orc_refine(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                   orc_tables *t, int64_t max_iters, int64_t *num_new);
int64_t orc_refine_clip(const orc_cfg *cfg, const float *xyzA, const float *xyzB,
                        orc_tables *t, int64_t max_iters, int64_t *num_new, float clip);

/* DistanceMetric's cutoff as the constructor installs it (distance_metric.cc:76-87)
 * and orc_dp with an explicit cutoff. */
float orc_initial_clip(const orc_cfg *cfg);
void orc_dp_clip(const orc_cfg *cfg, const float *xyzA, const float *xyzB_rot, float clip,
                 float *cost, uint8_t *bp);

/* ---- alternate solutions by orientation (-alt-o), minrms_oracle_alt.cc ---- */
typedef struct {
  int32_t num_solutions;       /* SearchOr::Settings::num_solutions (>= 2)            */
  float   min_3d_difference;   /* alt_min_3d_difference                               */
  float   rmsd_tolerance;      /* alt_rmsd_tolerance, -1 = ALTERNATES_NOT_LIMITED_BY_RMSD */
  int32_t refine_all;          /* refine_method == REFINE_ALL                         */
  int32_t by_pairs;            /* alt_method == ALT_METHOD_PAIRS (-alt-r)             */
  int32_t max_pairs_common;    /* alt_max_pairs_common                                */
} orc_alt_cfg;

void orc_principal_axes(int npts, const float *xyz, float eig[3], float axes[12]);
float orc_fast_rot_ssd(const float o1[12], const float o2[12], const float eig[3],
                       const float axes[12], int npts);
float orc_slow_rot_ssd(const float o1[12], const float o2[12], const float *xyz, int npts);
/* tables: ssd[S][range_n], xform[S][range_n][12], pairs[S][range_n][n_max][2] */
int orc_search_alt(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA, const float *xyzB,
                   const float *og, int64_t n_og, float *ssd, float *xform, int16_t *pairs,
                   float *io_clip);
/* Search() with ALT_METHOD_PAIRS: FirstPass, then one more pass over the
 * orientations per rank (MultiPass, search_or_dyn3d.cc:367-414,1969-2250). */
int orc_search_alt_pairs(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA,
                         const float *xyzB, const float *og, int64_t n_og, float *ssd,
                         float *xform, int16_t *pairs, float *io_clip);
int orc_pairs_in_common(int n1, const int16_t *a1, int n2, const int16_t *a2);
int64_t orc_refine_alt(const orc_cfg *cfg, const orc_alt_cfg *alt, const float *xyzA,
                       const float *xyzB, float *ssd, float *xform, int16_t *pairs,
                       int64_t max_iters, int64_t *num_new, float clip);

#ifdef __cplusplus
}
#endif
#endif
