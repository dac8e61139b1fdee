/*
 * libcaretta_hip -- C ABI of the MI355X-native pairwise structural-alignment path.
 *
 * Drop-in boundary for the numba-compiled functions of TurtleTools/caretta v0.2.0 that form the
 * all-vs-all pairwise alignment path.  The reference has no FFI (it is pure Python + @nb.njit), so
 * each entry point names the reference function it replaces (paths relative to the reference
 * root).  Plain pointers and sizes only; every array is C-contiguous; all floating point is FP64.
 * Unless a parameter is documented as a device pointer, buffers are host memory owned by the
 * caller, never retained after the call returns.
 *
 * Every function returns 0 on success or a negative cr_status; cr_last_error() gives the message
 * (thread-local).  Per-pair soft conditions are reported in `flags`, not as errors.
 * All compute entry points need a gfx950 device: there is no CPU fallback.
 */
#ifndef CARETTA_HIP_H
#define CARETTA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CR_ABI_VERSION 1

typedef enum {
    CR_OK = 0,
    CR_ERR_ARGUMENT = -1,     /* bad pointer / size / unsupported width   -> ValueError  */
    CR_ERR_HIP = -2,          /* HIP runtime failure (incl. no device)    -> RuntimeError */
    CR_ERR_MEMORY = -3,       /* host or device allocation failed         -> MemoryError */
    CR_ERR_STATE = -4         /* call order violated (e.g. fetch before run) */
} cr_status;

/* per-pair flag bits */
#define CR_FLAG_SEED_SKIPPED 1u     /* <=3 seed positions: superposition skipped (multiple_alignment.py:337-342) */
#define CR_FLAG_METRICS_SKIPPED 2u  /* <3 aligned positions: RMSD/coverage/TM not computed (assert at :1034)     */
#define CR_FLAG_SEED_ALL_ZERO 4u    /* tensor Smith-Waterman matrix all zero (the reference raises TypeError)    */
#define CR_FLAG_MEAN_UNSUPERPOSED 8u /* progressive node: <=3 aligned positions, mean taken on raw coordinates (:364-368) */

typedef struct cr_context cr_context;   /* device + stream + profiling events */
typedef struct cr_batch cr_batch;       /* structures resident in HBM + per-pair scratch */

/* Score/alignment parameters; defaults are the reference's hard-coded values
 * (multiple_alignment.py:490-492, bin/caretta-cli:39-44, dynamic_time_warping.py:205,226). */
typedef struct {
    double gamma_tensor;   /* 7.0  */
    double gamma_coords;   /* 0.03 */
    double gap_open;       /* 1.0  */
    double gap_extend;     /* 0.01 */
    double sw_gap;         /* 0.0  */
} cr_params;

/* Per-pair scalar results of the batched pipeline (host layout == device layout). */
typedef struct {
    double sw;             /* smith_waterman_score of the coordinate score matrix: the P x P matrix entry
                              (multiple_alignment.py:164) */
    double dtw_score;      /* score returned by dtw_align (dynamic_time_warping.py:184) */
    double R[9];           /* rotation, row-major; X_j @ R + t ~ X_i on the aligned positions
                              (superposition_functions.py:33) */
    double t[3];           /* translation (superposition_functions.py:34) */
    double rmsd;           /* score_functions.py:15-19 */
    double coverage;       /* multiple_alignment.py:1046-1048 */
    double tm;             /* multiple_alignment.py:59-70 (the reference's own formula) */
    double seed_score;     /* smith_waterman score on the tensor score matrix (multiple_alignment.py:332) */
    int32_t aln_len;       /* length of the dtw_align alignment rows */
    int32_t aln_start;     /* internal: first valid element in the device alignment rows */
    int32_t seed_len;      /* length of the seed smith_waterman alignment */
    uint32_t flags;        /* CR_FLAG_* */
} cr_pair_result;

#define CR_NUM_STAGES 2    /* k_seed: tensor SW fill + traceback + Kabsch; k_align: SW score + DTW fill + traceback + metrics */

const char *cr_last_error(void);
int cr_abi_version(void);
int cr_device_count(int *count);

/* ---- context ---------------------------------------------------------------------------- */
/* Device scratch released by batches / drop-ins is kept for reuse (hipMalloc / hipFree dominate short calls);
 * this returns all of it to the driver.  CARETTA_NO_CACHE=1 in the environment disables the reuse altogether. */
int cr_device_trim(int device);
/* `stream` is a hipStream_t to launch on or NULL to create a private stream.  The private stream is a BLOCKING
 * stream (hipStreamDefault): it is implicitly ordered with the legacy default stream, on which PyTorch and RCCL
 * work unless told otherwise, so a collective queued after cr_batch_run sees the finished scores. */
int cr_context_create(int device, void *stream, cr_context **out);
/* Borrow `stream` as it is, INCLUDING the legacy default stream (hipStream_t 0 -- what
 * torch.cuda.current_stream().cuda_stream returns by default, and which cr_context_create cannot tell from
 * "no stream"). */
int cr_context_create_on_stream(int device, void *stream, cr_context **out);
int cr_context_destroy(cr_context *ctx);
int cr_context_synchronize(cr_context *ctx);
int cr_context_stream(cr_context *ctx, void **stream_out);
/* Per-stage HIP events on the launch stream: a ring of `slots` event sets, one per cr_batch_run
 * (0 disables; calling it again resets the ring). */
int cr_context_set_profiling(cr_context *ctx, int slots);

/* ---- batched all-vs-all pipeline -------------------------------------------------------- */
/* Replaces the pair loop MultipleAlignment.make_pairwise_matrix (multiple_alignment.py:158-170)
 * together with Protein.score_function (:321-349), the 2-sequence dtw_align (:263-275) and the
 * per-pair body of make_rmsd_coverage_tm_matrix (:1028-1054, superpose_first=False).
 *
 * coords  f64[total, 3], tensors f64[total, d], offsets i64[P+1] (structure s = rows
 * offsets[s]..offsets[s+1]).  Uploads the structures to HBM. */
int cr_batch_create(cr_context *ctx, const double *coords, const double *tensors, const int64_t *offsets,
                    int64_t num_structures, int64_t d, cr_batch **out);
/* pairs i32[npairs, 2]: ordered (i, j) = (rows, columns); structure j is superposed onto i. */
int cr_batch_set_pairs(cr_batch *b, const int32_t *pairs, int64_t npairs);
/* Enqueue the two kernels on the context's stream (asynchronous).  `d_sw_out`, if not NULL, is a
 * DEVICE pointer to f64[npairs] that also receives the `sw` scores (e.g. a torch tensor that is
 * then all-gathered over RCCL). */
int cr_batch_run(cr_batch *b, const cr_params *params, double *d_sw_out);
/* cr_batch_run with the download folded in: `results` (npairs records) and / or `aln` (int32 [npairs][2][aln_stride]:
 * cr_batch_fetch_i32's layout, except that only the first results[p].aln_len entries of a row are written -- what lies
 * behind them in the caller's array is left as it was) are PAGE-LOCKED host arrays (cr_host_alloc) that the alignment kernel itself writes
 * into -- every wave stores its pair's rows and record over PCIe when it has finished the pair, under the fills of the
 * other waves -- so nothing is left to copy after the last kernel.  Asynchronous like cr_batch_run: the arrays are
 * complete after cr_context_synchronize.  (The device-side copies stay valid: cr_batch_fetch* still work.)
 * This is how the reference's per-pair results (multiple_alignment.py:263-275: every alignment goes back to the
 * caller) leave the device without a serial download phase. */
int cr_batch_run_stream_i32(cr_batch *b, const cr_params *params, cr_pair_result *results, int32_t *aln,
                            int64_t aln_stride, double *d_sw_out);
/* The same for callers that only want the P x P matrix entries (MultipleAlignment.make_pairwise_matrix,
 * multiple_alignment.py:158-170: smith_waterman_score of Protein.score_function per pair): the seed kernel, then the
 * coordinate score matrix + smith_waterman_score WITHOUT the pairwise dtw_align, its traceback and metrics.
 * Afterwards only cr_batch_fetch_scores (and d_sw_out) have results; the flags carry the seed conditions
 * (CR_FLAG_SEED_SKIPPED, CR_FLAG_SEED_ALL_ZERO) only.  With sw_gap != 0 it runs the full pipeline. */
int cr_batch_run_scores(cr_batch *b, const cr_params *params, double *d_sw_out);
/* flexible=True: Protein.score_function short-circuits to the tensor score matrix (multiple_alignment.py:323-326), whose
 * smith_waterman_score (gap 0, dynamic_time_warping.py:205-222) is the P x P matrix entry (:164).  One launch over the pair
 * list; only gamma_tensor of `params` is read; afterwards cr_batch_fetch_scores has the entries (flags are 0: this path of
 * the reference raises nothing).  d_sw_out as for cr_batch_run. */
int cr_batch_run_tensor_scores(cr_batch *b, const cr_params *params, double *d_sw_out);
/* Synchronise and copy results to host.  Any pointer may be NULL.  results[npairs];
 * aln i64[npairs, 2, aln_stride] (rows padded with -2 after aln_len; aln_stride >= max(n+m)). */
int cr_batch_fetch(cr_batch *b, cr_pair_result *results, int64_t *aln, int64_t aln_stride);
/* The same with int32 alignment rows (half the bytes over PCIe).  Both variants lay the rows out on the device and
 * copy straight into the caller's arrays: at DMA speed when those are page-locked (cr_host_alloc below). */
int cr_batch_fetch_i32(cr_batch *b, cr_pair_result *results, int32_t *aln, int64_t aln_stride);
/* page-locked host memory for result arrays */
int cr_host_alloc(size_t bytes, void **out);
int cr_host_free(void *p);
/* Only what make_pairwise_matrix needs (multiple_alignment.py:158-170): sw f64[npairs] = the smith_waterman_score of
 * every pair, flags u32[npairs].  Either pointer may be NULL. */
int cr_batch_fetch_scores(cr_batch *b, double *sw, uint32_t *flags);
int cr_batch_max_aln_len(cr_batch *b, int64_t *out);
/* per-stage device time in ms, averaged over the recorded runs (at most `slots` of them) */
int cr_batch_stage_ms(cr_batch *b, float ms[CR_NUM_STAGES], int *runs_averaged);
/* algorithmic HBM bytes of one run of the current pair list (SURVEY.md 8(d) B_alg) and DP cells */
int cr_batch_work(cr_batch *b, double *alg_bytes, double *cells);
/* Which kernel family cr_batch_set_pairs chose for the current pair list (the results do not depend on it; the reference's
 * pair loop, multiple_alignment.py:158-170, knows nothing of it): family = CR_LAYOUT_* -- one wave per pair; four-wave teams;
 * one workgroup per pair with a wave per strip ("wide"); scores formed by their own launches ("staged"); mid-size lists with a
 * pair's rows over 2 .. 8 waves ("duo") or one wave of recurrences + waves of scores ("trio") --, rows per lane of the first
 * `strips_a` strips and of the others (equal when the layout has one kind of strip).  Any pointer may be NULL.
 * CR_LAYOUT_CLASSES: a ragged list that cr_batch_set_pairs split into size classes (at most three: longest structure of a
 * pair <= 320 rows / <= 1 472 rows / longer), each laid out as a list of its own; rows_a then holds their number and
 * cr_batch_part_layout reports every class (part 0 .. count - 1; a list that was not split has the one part 0). */
enum { CR_LAYOUT_SINGLE = 0, CR_LAYOUT_TEAM = 1, CR_LAYOUT_WIDE = 2, CR_LAYOUT_STAGED = 3, CR_LAYOUT_DUO = 4, CR_LAYOUT_TRIO = 5,
       CR_LAYOUT_CLASSES = 6 };
int cr_batch_layout(cr_batch *b, int *family, int *rows_a, int *rows_b, int *strips_a);
int cr_batch_part_layout(cr_batch *b, int part, int *family, int *rows_a, int *rows_b, int *strips_a, int64_t *npairs);
/* The calibration switches (CARETTA_* environment variables: caretta_amd/csrc/cr_config.h) are read ONCE, when the library is
 * loaded; a measurement tool that changes its environment afterwards calls this to have the change seen.  NOT thread-safe:
 * call it only while no other library call is in flight on any thread (the struct it replaces is read without a lock by
 * every call, cr_multi's host threads included).  No reference counterpart (the reference has no tuning knobs on this path). */
int cr_config_reload(void);
/* What cr_batch_set_pairs would decide for this pair list, WITHOUT a device or a context (host only): whether the list is split
 * into size classes and the kernel family of every part.  offsets i64[P + 1]; parts i32[3][5] = (family CR_LAYOUT_*, rows per
 * lane A, rows per lane B, strips with A, pairs) of every part, *nparts of them (1: one list); class_of_pair i32[npairs] (may be
 * NULL): the part every pair goes to.  No reference counterpart (the reference's pair loop, multiple_alignment.py:158-170, has
 * one code path). */
int cr_plan_layout(const int64_t *offsets, int64_t num_structures, int64_t d, const int32_t *pairs, int64_t npairs,
                   int32_t *class_of_pair, int32_t *parts, int *nparts);
int cr_batch_destroy(cr_batch *b);

/* ---- the same matrix over several GPUs of one node, from ONE process ---------------------- */
/* The reference calls make_pairwise_matrix from a single process (multiple_alignment.py:497-501 inside
 * align_from_structure_files); its pair loop (:158-170) has no cross-pair dependency.  A cr_multi owns one context
 * (device + stream) per listed GPU.  cr_multi_pairwise_scores replicates the structures on every GPU, deals the pair
 * set with cr_partition_pairs, runs every GPU's share from its own host thread (cr_batch_run_scores: no data-path
 * collective), and assembles the score vector with ONE grouped RCCL all-gather over xGMI (ncclCommInitAll +
 * ncclGroupStart / ncclAllGather per device / ncclGroupEnd; librccl is bound at run time).  The result does not
 * depend on the number of devices, bit for bit.
 * devices == NULL or ndev <= 0: every visible device.  cr_multi_create opens the contexts, starts one host thread per device
 * and creates the communicators (a box without a usable RCCL fails here); batches, pair lists and buffers of a call are
 * kept for the next call with the same layout (num_structures, d, offsets), which only re-uploads coordinates and tensors. */
typedef struct cr_multi cr_multi;
int cr_multi_create(const int *devices, int ndev, cr_multi **out);
int cr_multi_device_count(cr_multi *m, int *ndev);
/* scores f64[P(P-1)/2] = smith_waterman_score of Protein.score_function per pair, flags u32[P(P-1)/2] (may be NULL),
 * both in the row-major i < j order of multiple_alignment.py:162-163 (what cr_assemble_matrix takes). */
int cr_multi_pairwise_scores(cr_multi *m, const double *coords, const double *tensors, const int64_t *offsets,
                             int64_t num_structures, int64_t d, const cr_params *params, double *scores, uint32_t *flags);
/* ms of the last call's phases: [0] upload + kernels of the slowest device, [1] the all-gather, [2] device 0's copy to the
 * host + the scatter to pair order.  [0], [1] and the copy are read from events on the devices' streams (no host wait
 * separates the phases), the scatter is host time. */
int cr_multi_last_ms(cr_multi *m, float ms[3]);
/* nodes[device_count]: the NUMA node each device's host thread was pinned to (CARETTA_MULTI_NUMA=1; default off: all -1).
 * No reference counterpart (the reference is single-device). */
int cr_multi_numa_nodes(cr_multi *m, int *nodes);
int cr_multi_destroy(cr_multi *m);
/* The deal of the pair set over `world` devices or ranks (host only): indices into the row-major i < j pair list owned
 * by `rank`, ascending -- pairs sorted by DP cell count (descending, stable on the index) and dealt round robin; equal
 * lengths: index % world == rank.  idx_out (may be NULL to query the count) needs ceil(npairs / world) entries. */
int cr_partition_pairs(const int64_t *lengths, int64_t num_structures, int world, int rank, int64_t *idx_out,
                       int64_t *count_out);

/* ---- single-call drop-ins (host buffers in, host buffers out) --------------------------- */
/* score_functions.make_score_matrix(a, b, get_gaussian_score, gamma)   score_functions.py:23-51 */
int cr_make_score_matrix(cr_context *ctx, const double *a, int64_t n, const double *b, int64_t m, int64_t k,
                         double gamma, double *S);
/* Protein.score_function(other, flexible=False, ...)                   multiple_alignment.py:321-349 */
int cr_protein_score_function(cr_context *ctx, const double *coords_i, const double *tensors_i, int64_t n,
                              const double *coords_j, const double *tensors_j, int64_t m, int64_t d,
                              double gamma_tensor, double gamma_coords, double *S, uint32_t *flags);
/* One node of MultipleAlignment.progressive_align for two Proteins: make_intermediate_node
 * (multiple_alignment.py:193-234) = score_function (:204) + consensus-weight RBF (:207-210) -> dtw_align
 * (:211-214) -> Protein.mean_function (:351-381) + get_mean_weights (:73-82).  mult1/mult2 are the
 * multipliers of :199-202.  aln1/aln2: n+m entries; coords_out (n+m,3), tensors_out (n+m,d),
 * weights_out (n+m): the first *aln_len rows are the new node.  flags: CR_FLAG_* of the seed, bit 3 set when
 * the mean was taken without superposition (<=3 aligned positions, :364-368). */
int cr_progressive_node(cr_context *ctx, const double *coords_1, const double *tensors_1, const double *weights_1,
                        int64_t n, const double *coords_2, const double *tensors_2, const double *weights_2,
                        int64_t m, int64_t d, double mult1, double mult2, const cr_params *params,
                        double gamma_weight, int64_t *aln1, int64_t *aln2, int64_t *aln_len,
                        double *coords_out, double *tensors_out, double *weights_out, uint32_t *flags);
/* ---- whole guide tree: MultipleAlignment.progressive_align           multiple_alignment.py:172-253 --------
 * All leaves are uploaded once; every node of the tree (make_intermediate_node, :193-234) stays resident in
 * HBM, and nodes whose children are complete (one LEVEL of the tree) run as one launch pair, one wave per node.
 * tree: uint64 (tree_rows = 2P-3, 2) exactly as neighbor_joining returns it (neighbor_joining.py:19-95): rows
 * 2x, 2x+1 = (child, P + x) for x = 0 .. P-3, last row = the two nodes joined by the final node (:244-247).
 * Leaf weights are `consensus_weight` everywhere (:189-193).  Internal nodes are numbered 0 .. P-2 in creation
 * order (node id P + k; the final node is k = P-2). */
typedef struct cr_progressive cr_progressive;
int cr_progressive_align(cr_context *ctx, const double *coords, const double *tensors, const int64_t *offsets,
                         int64_t num_structures, int64_t d, const uint64_t *tree, int64_t tree_rows,
                         const cr_params *params, double consensus_weight, double gamma_weight,
                         cr_progressive **out);
/* The same with flexible=True in the score function AND the mean function (multiple_alignment.py:323-326: the score matrix of
 * two nodes is the tensor RBF alone; :351-362: a node is its mean tensors, no coordinates): tensors f64[total][d] only.
 * cr_progressive_fetch_nodes then takes coords == NULL.  CR_ERR_STATE when a node outgrows the launch bound (1.5 x the
 * longest structure, at most 2048 columns): the caller walks the tree itself (cr_dtw_align per node). */
int cr_progressive_align_flexible(cr_context *ctx, const double *tensors, const int64_t *offsets, int64_t num_structures, int64_t d,
                                  const uint64_t *tree, int64_t tree_rows, const cr_params *params, double consensus_weight,
                                  double gamma_weight, cr_progressive **out);
/* sizes[0] = columns of the final alignment, [1] = internal nodes (P-1), [2] = sum of their lengths,
 * [3] = levels (= launch pairs), [4] = OR of all node flags (CR_FLAG_*) */
int cr_progressive_sizes(cr_progressive *h, int64_t sizes[5]);
/* msa: int64 (P, columns): residue index of structure s in every column, -1 = gap (the dict that
 * progressive_align returns, :248-251, as a matrix in input order) */
int cr_progressive_fetch_msa(cr_progressive *h, int64_t *msa);
/* table: int64 (P-1, 6) = child id 1, child id 2, length, level, flags, number of member structures */
int cr_progressive_node_table(cr_progressive *h, int64_t *table);
/* Internal nodes concatenated in creation order; any pointer may be NULL.  aln: for node k the two dtw_align rows
 * (2 * length entries, row 1 then row 2); coords (sum,3); tensors (sum,d); weights (sum). */
int cr_progressive_fetch_nodes(cr_progressive *h, int64_t *aln, double *coords, double *tensors, double *weights);
int cr_progressive_destroy(cr_progressive *h);
/* dynamic_time_warping.dtw_align / dtw_align_score (aln1 == NULL)      dynamic_time_warping.py:148-201
 * S f64[s_rows, s_cols] indexed S[seq1[i], seq2[j]]; aln1/aln2 need n+m entries. */
int cr_dtw_align(cr_context *ctx, const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                 const double *S, int64_t s_rows, int64_t s_cols, double gap_open, double gap_extend,
                 int64_t *aln1, int64_t *aln2, int64_t *aln_len, double *score);
/* dynamic_time_warping.smith_waterman_score                            dynamic_time_warping.py:205-222 */
int cr_smith_waterman_score(cr_context *ctx, const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                            const double *S, int64_t s_rows, int64_t s_cols, double gap, double *score);
/* ---- many explicit score matrices per launch --------------------------------------------------------
 * MultipleAlignment.make_pairwise_matrix calls smith_waterman_score once per pair on the matrix a SequenceBase
 * plugin's score_function returned (multiple_alignment.py:158-170), progressive_align calls dtw_align once per tree
 * node (:204-217).  A cr_explicit_batch holds a list of such problems -- matrices packed in S, index sequences packed
 * in seqs (int64, as the reference's np.arange) -- resident in HBM; the two entry points below run
 * dynamic_time_warping.py:205-222 / :148-184 over the whole list in one launch sequence. */
typedef struct {
    int64_t s_off;              /* element offset of the s_rows x s_cols matrix in S */
    int64_t seq1_off, seq2_off; /* element offsets of the index sequences in seqs */
    int32_t s_rows, s_cols;
    int32_t n, m;               /* sequence lengths */
} cr_explicit_problem;
typedef struct cr_explicit_batch cr_explicit_batch;
int cr_explicit_batch_create(cr_context *ctx, const double *S, int64_t s_elems, const int64_t *seqs, int64_t seq_elems,
                             const cr_explicit_problem *problems, int64_t count, cr_explicit_batch **out);
int cr_explicit_batch_destroy(cr_explicit_batch *b);
/* device time (ms, HIP events on the context's stream) of the kernels of the last batch call */
int cr_explicit_batch_last_ms(cr_explicit_batch *b, float *ms);
/* scores[count] = smith_waterman_score(seq1_k, seq2_k, S_k, gap)      dynamic_time_warping.py:205-222 */
int cr_smith_waterman_score_batch(cr_explicit_batch *b, double gap, double *scores);
/* smith_waterman WITH its traceback (dynamic_time_warping.py:226-278) of every problem: aln i64[count][2][aln_stride] = the
 * two index rows (-1 = gap; padded with -2 behind aln_len[p]), scores[p] = the maximum, all_zero[p] != 0 where the matrix has no
 * positive cell (the reference then fails to unpack max_pos = None).  aln_stride >= the longest n + m.  scores, all_zero may
 * be NULL.  A -1 in seq2 is rejected (only smith_waterman_score gives it a meaning). */
int cr_smith_waterman_batch(cr_explicit_batch *b, double gap, int64_t *aln, int64_t aln_stride, int64_t *aln_len, double *scores,
                            int32_t *all_zero);
/* dtw_align(seq1_k, seq2_k, S_k, gap_open, gap_extend) for every k     dynamic_time_warping.py:148-184
 * aln: int64 [count][2][aln_stride] (aln_stride >= longest n + m), rows padded with -2; aln may be NULL (scores only). */
int cr_dtw_align_batch(cr_explicit_batch *b, double gap_open, double gap_extend, int64_t *aln, int64_t aln_stride,
                       int64_t *aln_len, double *scores);
/* dynamic_time_warping.smith_waterman; *all_zero = 1 where the reference raises (no maximum)
 *                                                                      dynamic_time_warping.py:226-278 */
int cr_smith_waterman(cr_context *ctx, const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                      const double *S, int64_t s_rows, int64_t s_cols, double gap,
                      int64_t *aln1, int64_t *aln2, int64_t *aln_len, double *score, int *all_zero);
/* superposition_functions.paired_svd_superpose                         superposition_functions.py:7-35 */
int cr_paired_svd_superpose(cr_context *ctx, const double *x1, const double *x2, int64_t k, double *R, double *t);
/* superposition_functions.paired_svd_superpose_with_subset             superposition_functions.py:39-60 */
int cr_paired_svd_superpose_with_subset(cr_context *ctx, const double *c1, int64_t n, const double *c2, int64_t m,
                                        const double *s1, const double *s2, int64_t k,
                                        double *o1, double *o2, double *o3);
/* superposition_functions.apply_rotran                                 superposition_functions.py:64-80 */
int cr_apply_rotran(cr_context *ctx, const double *x, int64_t k, const double *R, const double *t, double *out);
/* score_functions.get_rmsd                                             score_functions.py:15-19 */
int cr_get_rmsd(cr_context *ctx, const double *x1, const double *x2, int64_t k, double *out);
/* multiple_alignment.tm_score                                          multiple_alignment.py:59-70 */
int cr_tm_score(cr_context *ctx, const double *x1, const double *x2, int64_t k, int64_t l1, int64_t l2, double *out);

/* make_rmsd_coverage_tm_matrix over a finished multiple alignment       multiple_alignment.py:1000-1055
 * coords f64[total,3] + offsets i64[P+1] as in cr_batch_create; msa i32[P, W] residue index per column, -1 = gap.
 * superpose != 0: per-pair Kabsch on the common positions first (superpose_first=False, what the reference's
 * --matrix output uses, :571); 0: coordinates compared as given (after the caller's own superpose()).
 * rmsd/coverage/tm: f64[P, P], diagonals 0 / 1 / 1 as the reference initialises them.  Pairs with fewer than 3
 * common positions (the reference asserts) get rmsd = tm = NaN. */
int cr_msa_metrics(cr_context *ctx, const double *coords, const int64_t *offsets, int64_t P, const int32_t *msa,
                   int64_t W, int superpose, double *rmsd, double *coverage, double *tm);
/* superpose_core (multiple_alignment.py:914-950): all P structures fitted onto structure `ref` over the gap-free
 * alignment columns core[ncore] (msa int32 [P, W], -1 = gap): one launch, one wave per structure.  coords_out
 * (sum of lengths, 3): the moved coordinates; the reference is shifted by the centroid of its core positions. */
int cr_superpose_core(cr_context *ctx, const double *coords, const int64_t *offsets, int64_t num_structures,
                      const int32_t *msa, int64_t width, const int32_t *core, int64_t ncore, int64_t ref,
                      double *coords_out);
/* superpose_reference (multiple_alignment.py:953-972): every structure fitted onto structure `ref` over the columns
 * the two share, in list order -- the reference is refitted onto itself when its turn comes and later structures are
 * fitted onto that copy, as in the reference's loop.  Three launches (before / the reference / after), one wave per
 * structure.  CR_ERR_ARGUMENT if a structure shares 3 or fewer columns with the reference (:965). */
int cr_superpose_reference(cr_context *ctx, const double *coords, const int64_t *offsets, int64_t num_structures,
                           const int32_t *msa, int64_t width, int64_t ref, double *coords_out);
/* One group of superpose_references (multiple_alignment.py:975-997): the structures which[nwhich] fitted onto
 * structure `ref` as it currently is in coords (in place).  All listed structures see the same copy of `ref`: where the
 * reference's loop refits `ref` onto itself in the middle of a group, the caller splits the list there and lists `ref`
 * alone.  One wave per listed structure. */
int cr_superpose_members(cr_context *ctx, double *coords, const int64_t *offsets, int64_t num_structures,
                         const int32_t *msa, int64_t width, int64_t ref, const int32_t *which, int64_t nwhich);
/* helper.nb_mean_axis_0 (sequential column means, as numba computes them)   helper.py:46-53 */
int cr_mean_axis0(const double *x, int64_t rows, int64_t cols, double *out);

/* ---- host-side integer / tree work ------------------------------------------------------ */
/* helper.get_common_positions                                          helper.py:13-42 */
int cr_get_common_positions(const int64_t *a1, const int64_t *a2, int64_t len, int64_t *p1, int64_t *p2, int64_t *k);
/* neighbor_joining.neighbor_joining; tree u64[2P-3, 2], branch_lengths f64[2P-3]
 *                                                                      neighbor_joining.py:19-157 */
int cr_neighbor_joining(const double *D, int64_t P, uint64_t *tree, double *branch_lengths);
/* the same on the device: ONE persistent launch of up to 64 workgroups (4 waves each, a wave owns up to 8 rows), no
 * barrier in the join loop; every sum and tie in the reference's order (results equal cr_neighbor_joining's bit for
 * bit).  Symmetric finite matrices of up to 2048 nodes; anything else -- and a launch whose workgroups could not all
 * be resident (busy or partitioned device) -- is handed to cr_neighbor_joining.
 *                                                                      neighbor_joining.py:19-157 */
int cr_neighbor_joining_device(cr_context *ctx, const double *D, int64_t P, uint64_t *tree, double *branch_lengths);
/* scatter per-pair scores into the symmetric P x P matrix (multiple_alignment.py:161-169) */
int cr_assemble_matrix(const int32_t *pairs, const double *scores, int64_t npairs, int64_t P, double *M);

#ifdef __cplusplus
}
#endif
#endif
