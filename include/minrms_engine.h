/* minrms_engine.h -- C ABI of the MI355X engine for the MINRMS exhaustive
 * RMSD-vs-N search.
 *
 * The library (libminrms_hip.so) sits behind the reference's search plug-in
 * interface `class SearchOr` (bin/minrms/search_or.h:57-317): one engine object
 * does what one `SearchOrDyn3d` object does between construction and the
 * writers -- Search(og,start,stop), Refine() -- and hands back the three result
 * tables the reference's writers read (aaSumSqdDists / aaTransforms /
 * aaAlignments, bin/minrms/search_or_dyn3d.h:293-310).
 *
 * Conventions
 *   - plain C, caller-owned host buffers, engine-owned device memory;
 *   - every function returns 0 on success or a negative MINRMS_E_* code and
 *     leaves a message for minrms_engine_last_error(); nothing calls exit()
 *     (the reference's ERR() macro, lib/global_utils/ERR.h:20-21, prints and
 *     exits: an adapter turns the error code back into ERR());
 *   - one host thread per engine; an engine is bound to one GPU;
 *   - float = IEEE binary32 everywhere (`typedef float Real`,
 *     lib/vect_3d/vect_3d.h:66); a transform is a row-major 3x4 matrix
 *     (`Matrix3x4`, lib/vect_3d/vect_3d.h:91) applied as x' = R x + t;
 *   - residue indices are 0-based, a match is a pair of int16 (`PairwiseMatch`,
 *     lib/pair_alignment/pair_alignment.h:80-108);
 *   - result entry k describes the alignment with n = n_min + k matched pairs.
 */
#ifndef MINRMS_ENGINE_H
#define MINRMS_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MINRMS_UPPER_BOUND 1.0e26f   /* DistanceMetric::UPPER_BOUND, lib/dyn/distance_metric.cc:49 */

enum {
  MINRMS_OK           =  0,
  MINRMS_E_ARG        = -1,   /* bad argument / unsupported size                */
  MINRMS_E_DEVICE     = -2,   /* HIP runtime error (no GPU, out of memory, ...) */
  MINRMS_E_NOSOLUTION = -3,   /* "No solutions were found satisfying your
                                 criteria" (search_or_dyn3d.cc:1929-1941)       */
  MINRMS_E_STATE      = -4    /* call order violated                            */
};

/* Replaces SearchOrDyn3d::Settings (bin/minrms/search_or_dyn3d.h:159-215) plus
 * the few SearchOr::Settings / PairAlignSettings members the hot path reads
 * (bin/minrms/search_or.h:149-308). */
typedef struct minrms_cfg {
  int32_t N1, N2;          /* residues of structure 1 (fixed) / 2 (moved)          */
  int32_t num_bb;          /* Biopolymer::Residue::NumBackboneAtoms(), 1 = CA only */
  int32_t n_min, n_max;    /* PairAlignSettings::n_min, Settings::n_max            */
  float   max_rmsd;        /* dyn3d_max_rmsd, -1 = dyn3d_NO_MAX_RMSD               */
  float   max_dist;        /* dyn3d_max_dist, -1 = NO_MAX_PHYS_DISTANCE            */
  int32_t refine_while_searching;   /* SearchOr::Settings::refine_while_searching  */
  int32_t of_enable;       /* orientation_filter_nw                                */
  float   of_max_dist;     /* orientation_filter_nw_max_dist                       */
  int32_t of_min_matches;  /* orientation_filter_nw_min_num_matches                */
  int32_t device;          /* HIP device ordinal                                   */
  int32_t reserved;
  int64_t workspace_bytes; /* cap on per-batch device workspace; 0 = 96 GiB, never
                              more than 80 % of the free device memory            */
  /* Alternate solutions by orientation (SearchOr::Settings,
   * bin/minrms/search_or.h:154-244): `-alt-o count,min_distance[,rmsd_ratio]`
   * = ALT_METHOD_3D with refine_while_searching, `-alt-o-orig ...` =
   * ALT_METHOD_3D_ORIG without it (the parser ties the two together,
   * minrms_parser.cc:946-984).  0 or 1 = best only.                           */
  int32_t num_solutions;          /* SearchOr::Settings::num_solutions             */
  float   alt_min_3d_difference;  /* alt_min_3d_difference (Angstrom, RMS)         */
  float   alt_rmsd_tolerance;     /* alt_rmsd_tolerance, -1 = ALTERNATES_NOT_LIMITED_BY_RMSD */
  int32_t refine_all;             /* refine_method == REFINE_ALL (`-r-all`)        */
  /* `-alt-r count,max_pairs_in_common[,rmsd_ratio]` = ALT_METHOD_PAIRS: alternates
   * must share at most alt_max_pairs_common matched pairs with every better
   * rank; found by one more pass over the orientations per rank (MultiPass,
   * search_or_dyn3d.cc:1969-2250).  alt_min_3d_difference is then unused.      */
  int32_t alt_by_pairs;
  int32_t alt_max_pairs_common;
} minrms_cfg;

typedef struct minrms_refine_stats {
  int64_t num_iterations;  /* the count Refine() prints (search_or_dyn3d.cc:2643)  */
  int64_t num_new_solutions;
  int64_t dp_runs;         /* DP tables actually evaluated on the GPU              */
  int64_t rounds;          /* host<->device round trips                            */
} minrms_refine_stats;

typedef struct minrms_engine minrms_engine;

/* Replaces SearchOrDyn3d::SearchOrDyn3d(Settings const&)
 * (bin/minrms/search_or_dyn3d.cc:109-234).  xyzA/xyzB: the backbone-atom
 * coordinates of settings.aMol_f[0]/[1], residue-major, num_bb atoms per
 * residue, 3 floats per atom -- the flat copies the reference itself makes at
 * search_or_dyn3d.cc:186-224. */
int minrms_engine_create(const minrms_cfg *cfg, const float *xyzA,
                         const float *xyzB, minrms_engine **out);

/* Device memory of a destroyed engine is not returned to the system but kept
 * for the next engine this process creates on the same device (a batch of pairs
 * creates one engine per pair; repeated multi-GB hipMalloc/hipFree cycles cost
 * far more than the jobs). */
void minrms_engine_destroy(minrms_engine *e);

/* Return the memory kept from destroyed engines to the system.  No counterpart
 * in the reference. */
void minrms_engine_trim(void);

/* Back to the state right after construction (tables = 1e26f / identity, as
 * search_or_dyn3d.cc:177-183 initialises them); device buffers are kept. */
int minrms_engine_reset(minrms_engine *e);

/* Replaces OrientationGenerator::GenerateOrientationsUsingFragmentMatching +
 * the random_shuffle of the constructor (lib/or_gen/or_gen.cc:237-587, :179).
 * interval_pairs: n_intervals rows {first1,last1,first2,last2}, inclusive
 * 0-based residue indices (settings.fm_interval_pair_list after the parser has
 * resolved residue ids; one row {0,N1-1,0,N2-1} is the "-HS" case).
 * cutoff_rmsd < 0 = fm_NO_CUTOFF_RMSD.  shuffle != 0 applies the reference's
 * shuffle order.  Writes min(count,cap) transforms to out_og (host, [cap][12])
 * and the total to *count. */
int minrms_engine_gen_orientations(minrms_engine *e, int32_t frag_len,
                                   float cutoff_rmsd, int32_t n_intervals,
                                   const int32_t *interval_pairs, int32_t shuffle,
                                   float *out_og, int64_t cap, int64_t *count);

/* Replaces SearchOr::Search(og, start, stop) (bin/minrms/search_or.h:83-85,
 * SearchOrDyn3d::FirstPass, search_or_dyn3d.cc:1643-1946).  og: host array of
 * all n_og orientations (og[i].transform, lib/or_gen/or_gen.h:73-95); the
 * orientations start <= o_id < stop are evaluated.  May be called repeatedly
 * (e.g. one o_id range per call); the per-N minimum is kept across calls with
 * the reference's rule: smaller sum-squared distance wins, equal values go to
 * the smaller o_id. */
int minrms_engine_search(minrms_engine *e, const float *og, int64_t n_og,
                         int64_t start, int64_t stop);

/* With cfg.of_enable the Needleman-Wunsch orientation filter runs in front of
 * the search exactly as in FirstPass (search_or_dyn3d.cc:1667-1673,1696-1698;
 * OrientationFilter::GoodOrientation, lib/or_gen/orientation_filter.cc:88-130;
 * DynNW, lib/dyn/dynNW.cc:147-531).  filter_verdicts copies the verdict of
 * every orientation of the last search() range (1 = passed; all 1 with the
 * filter off) to pass[0..min(cap,count)) and the range length to *count.
 *
 * Every GoodOrientation call re-installs DistanceMetric's process-wide cutoff
 * through a sqrt/square round trip (orientation_filter.cc:96,127;
 * lib/dyn/distance_metric.cc:76-100), so with -of and -max-dist together the
 * cutoff an orientation (and later Refine) sees depends on the number of
 * filter calls made before it.  The engine counts them across search() calls
 * (reset() zeroes the count).  A caller that splits one reference run over
 * several engines sets the count an engine starts from -- the number of
 * orientations that precede its range -- with set_filter_calls before search(),
 * and the total after the exchange of keys; then every engine reproduces the
 * one-process result bit for bit. */
int minrms_engine_filter_verdicts(minrms_engine *e, uint8_t *pass, int64_t cap,
                                  int64_t *count);
int minrms_engine_set_filter_calls(minrms_engine *e, int64_t calls);

/* Multi-GPU seam (documented intent of search_or.h:92-103).  After each rank has
 * searched its own o_id range, the per-N keys
 *     key[k] = (bits(ssd) << 32) | o_id        (all-ones = none)
 * are all-reduced with MIN across ranks by the caller (one RCCL call on
 * range_n uint64) and pushed back; every rank then materialises the winners. */
int minrms_engine_get_keys(minrms_engine *e, uint64_t *keys /*[range_n]*/);
int minrms_engine_set_keys(minrms_engine *e, const uint64_t *keys /*[range_n]*/);

/* Device address of the key table (range_n uint64) for callers that run the
 * all-reduce directly on device memory. */
int minrms_engine_keys_device_ptr(minrms_engine *e, void **ptr);

/* Alternate solutions by orientation (-alt-o, -alt-o-orig) with the search
 * split over several engines.  The reference keeps, per n, a file of the
 * orientations that came within the RMSD tolerance of the best so far
 * (KeepTrackOfAltSolutions, search_or_dyn3d.cc:477-542); an engine that starts
 * in the middle of the orientation list keeps a superset.  After search(), the
 * caller collects every engine's records (entry k = n - n_min, o_id, sum of
 * squared distances, RMSD, transform[12]), concatenates them in any order and
 * hands the union to every engine, which rebuilds the one-process lists from it;
 * then get_keys / all-reduce / set_keys and finalize_search as usual.
 * Alternates by matched pairs (-alt-r): the same exchange of records (transforms
 * unused), and minrms_engine_set_collectives below. */
int minrms_engine_alt_num_candidates(minrms_engine *e, int64_t *count);
int minrms_engine_alt_get_candidates(minrms_engine *e, int32_t *k, int64_t *o_id,
                                     float *ssd, float *rmsd, float *xf);
int minrms_engine_alt_set_candidates(minrms_engine *e, int64_t count,
                                     const int32_t *k, const int64_t *o_id,
                                     const float *ssd, const float *rmsd,
                                     const float *xf);

/* Alternates by matched pairs (-alt-r) with the search split over several
 * engines.  MultiPass (search_or_dyn3d.cc:1969-2250) makes one more pass over
 * the orientations per rank of alternates, each pass depending on the ranks
 * before it.  With collectives installed, finalize_search runs every pass over
 * the engine's own o_id range only and calls back
 *   allgather (once): the o_ids this engine would revisit; the caller returns
 *              the o_ids of ALL engines, in any order, in out[0..*n_out), at
 *              most cap (= the number of orientations);
 *   allreduce_min (once per pass): the per-N keys of the pass, range_n uint64,
 *              replaced in place by the element-wise minimum over all engines;
 * every engine then re-solves the winners of the pass itself, so all engines
 * hold the same tables of every rank.  Both return 0 on success.  Every engine
 * of the job must make the same calls in the same order (they are collectives).
 * NULL, NULL removes them (one engine sweeps the whole range). */
typedef int (*minrms_allreduce_min_fn)(void *user, uint64_t *keys, int64_t n);
typedef int (*minrms_allgather_i64_fn)(void *user, const int64_t *mine, int64_t n_mine,
                                       int64_t *out, int64_t cap, int64_t *n_out);
int minrms_engine_set_collectives(minrms_engine *e, minrms_allreduce_min_fn allreduce_min,
                                  minrms_allgather_i64_fn allgather, void *user);

/* Fills the result tables from the current keys (re-solves the winning
 * orientations).  Returns MINRMS_E_NOSOLUTION if no n has a solution. */
/* "<count> updates for orientation <o_id + 1>": FirstPass prints such a line for
 * every orientation that improved an entry of the per-N table when it was
 * evaluated (count_num_updates, search_or_dyn3d.cc:1712,1825-1834,1885-1889).
 * count_updates(e, 1) before minrms_engine_search makes the search keep these
 * counts (it then sweeps its batches one after the other on one stream, in
 * o_id order); get_update_counts copies them out, counts[i] for orientation
 * start + i of the last search, n = stop - start.  The counts are relative to
 * the table the engine held when the search began, as the reference's are.
 * Passes after the first (-alt-r) are not counted. */
int minrms_engine_count_updates(minrms_engine *e, int32_t enable);
int minrms_engine_get_update_counts(minrms_engine *e, int32_t *counts /*[n]*/, int64_t n);

int minrms_engine_finalize_search(minrms_engine *e);

/* Replaces SearchOr::Refine() (bin/minrms/search_or.h:116,
 * SearchOrDyn3d::Refine, search_or_dyn3d.cc:2261-2658) for
 * refine_method = REFINE_BEST.  max_iters = -1: ITERATE_UNTIL_CONVERGENCE. */
int minrms_engine_refine(minrms_engine *e, int64_t max_iters,
                         minrms_refine_stats *stats);

/* The tables the reference's writers read (search_or_dyn3d.h:293-310), r = 0:
 *   ssd[range_n]               aaSumSqdDists[0][k]   (1e26f = no solution)
 *   xforms[range_n][12]        aaTransforms[0][k]
 *   pairs[range_n][n_max][2]   aaAlignments[0][k]    (first n_min+k rows valid)
 *   winner_oid[range_n]        o_id that produced the entry, -1 = none,
 *                              <= -2 = produced by refinement
 * Any pointer may be NULL. */
int minrms_engine_results(minrms_engine *e, float *ssd, float *xforms,
                          int16_t *pairs, int64_t *winner_oid);

/* The tables of rank `rank` (0 = best, 1 = 2nd best, ...; aaSumSqdDists[rank]
 * etc., search_or_dyn3d.h:293-310) when cfg.num_solutions > 1.  After
 * finalize_search they hold what Search() leaves there
 * (CalcAltSolutionsUsingOrientation, search_or_dyn3d.cc:421-450); after refine
 * what Refine() leaves (ranks refined per cfg.refine_all, then
 * CheckAlternatesForConsistency, :1099-1247).  With the search split over
 * several engines see minrms_engine_alt_set_candidates (alternates by
 * orientation) and minrms_engine_set_collectives (by matched pairs). */
int minrms_engine_results_rank(minrms_engine *e, int32_t rank, float *ssd,
                               float *xforms, int16_t *pairs);

/* No counterpart in the reference.  For callers that cannot allocate device
 * memory -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc: what minrms_engine_search + minrms_engine_results do for a
 * SHORT orientation list (the per-N best alignments among n_og orientations of
 * one pair; no refinement), with no hipMalloc / hipFree or other allocating
 * call, no engine object and no host synchronisation.
 *   cfg            as for minrms_engine_create (of_enable = 0, num_solutions <= 1)
 *   d_xyzA, d_xyzB DEVICE pointers, coordinates laid out as for create
 *   d_og           DEVICE pointer, n_og transforms [n_og][12]; o_id = index
 *   d_ssd, d_xforms, d_pairs, d_winner_oid
 *                  DEVICE pointers, the tables of minrms_engine_results
 *                  ([range_n], [range_n][12], [range_n][n_max][2], [range_n]);
 *                  any may be NULL
 *   stream         hipStream_t (NULL = the default stream).  The call only
 *                  enqueues kernels; results are ready when the stream has run.
 * All intermediate buffers live in one static 32 MiB device array per GPU, which
 * bounds the input: the call fails with MINRMS_E_ARG (and says how many bytes
 * it would need) beyond that -- about 36 orientations of a 150x150 pair, 4 of a
 * 400x400 pair -- and beyond 1024 orientations.  Calls on one GPU share that
 * array: two calls in flight on the same GPU at the same time race (calls on
 * one stream are ordered and safe).  If the DP kernel's bounded wait gives up
 * (an internal error) every ssd is NaN and every winner -3.
 * The FIRST call on a device looks up the array's address and loads the
 * library's code objects (which reserves the array's memory): make it outside
 * stream capture -- or call minrms_noalloc_prepare(device) at set-up, which does
 * exactly that and nothing else.  The caller's current device is restored. */
int minrms_search_noalloc(const minrms_cfg *cfg, const float *d_xyzA,
                          const float *d_xyzB, const float *d_og, int32_t n_og,
                          float *d_ssd, float *d_xforms, int16_t *d_pairs,
                          int64_t *d_winner_oid, void *stream);
int minrms_noalloc_prepare(int32_t device);

/* Number of DP cells (one (n,i,j) update of Dyn3d::Solve_IncrN /
 * InitTable_IncrN, lib/dyn/dyn3d.cc:135-387) the reference evaluates for one
 * orientation with these settings and no early termination. */
int64_t minrms_engine_cells_per_orientation(const minrms_cfg *cfg);

/* Measurement hooks (no counterpart in the reference).  The 3-index DP of a
 * batch is one kernel launch per layer block (63 layers) and column panel.
 * With timing enabled every such launch is bracketed by HIP events on the
 * engine's stream; minrms_engine_dp_block_time() synchronises the stream and
 * returns, for one layer block, the GPU milliseconds, launches and orientations
 * accumulated since the previous call for that block, and the number of
 * reference DP cells of one orientation that fall into the block. */
int minrms_engine_enable_timing(minrms_engine *e, int32_t level);
int minrms_engine_dp_block_time(minrms_engine *e, int32_t block, double *ms,
                                int64_t *launches, int64_t *orientations,
                                int64_t *cells_per_orientation);
int minrms_engine_num_blocks(minrms_engine *e);
/* How the sweep kernels visit one layer block: `rows` rows of `panels` panels
 * of `panel_width` columns, every cell by all 64 lanes of a wave (the cells a
 * launch executes, as opposed to the reference cells it covers). */
int minrms_engine_dp_block_shape(minrms_engine *e, int32_t block, int32_t *rows,
                                 int32_t *panels, int32_t *panel_width);
/* Columns the sweep kernels compute per row of one layer block: panels x
 * panel_width, less the last 8 when those lie outside the table. */
int minrms_engine_dp_block_columns(minrms_engine *e, int32_t block, int32_t *columns);

/* The same for the other kernels, by class.  enable_timing(1) brackets the
 * kernels of the sweep and of the batch that re-solves the winners, (2) also
 * every kernel of the refinement rounds (the events sit inside a chain of short
 * dependent launches and lengthen it: for profiling passes, not for timed runs).
 * kernel_time synchronises the stream and returns the GPU milliseconds and
 * launches of one class accumulated since the previous call for that class. */
enum {
  MINRMS_KC_CIJ = 0,        /* distance tables of a sweep batch (k_cij)                  */
  MINRMS_KC_TRACE = 1,      /* traceback + refit of a sweep batch, one thread per chain  */
  MINRMS_KC_REDUCE = 2,     /* running per-N minimum                                     */
  MINRMS_KC_COOP_CIJ = 3,   /* distance tables of a small batch (refinement, winners)    */
  MINRMS_KC_COOP = 4,       /* the DP of a small batch (cooperative kernel)              */
  MINRMS_KC_TRACE_WAVE = 5, /* traceback + refit of a small batch, one wave per chain    */
  MINRMS_KC_ROUND = 6,      /* the refinement's round kernel (replay + plan)             */
  MINRMS_KC_OTHER = 7,
  MINRMS_KC_DP_BLOCK0 = 8   /* + b: layer block b of the sweep (= dp_block_time)         */
};
int minrms_engine_kernel_time(minrms_engine *e, int32_t kernel_class, double *ms,
                              int64_t *launches);
/* What one table costs the cooperative kernel as the refinement rounds launch
 * it: compute waves (tasks) per table, columns per task, cells all tasks of a
 * table visit together (rows x columns of every layer block, each by the 64
 * lanes of a wave), rows of the first layer block (its longest chain of rows),
 * tasks of the first layer block, workgroups per table. */
int minrms_engine_coop_shape(minrms_engine *e, int32_t *tasks, int32_t *task_width,
                             int64_t *cells_visited, int32_t *rows_first_block,
                             int32_t *tasks_first_block, int32_t *workgroups);

const char *minrms_engine_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MINRMS_ENGINE_H */
