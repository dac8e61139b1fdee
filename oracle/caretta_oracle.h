/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * Plain-C restatement of the reference's (TurtleTools/caretta v0.2.0) pairwise
 * structural-alignment hot path, used as the checker by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing under
 * caretta_amd/ may include, link, import or execute it.
 *
 * Parity status: PINNED.  Every function here is checked against golden vectors
 * produced by executing the reference's own source (tests/golden/_gen/
 * generate_golden.py; stand-in `numba.njit` = identity) -- see tests/test_oracle_golden.py.
 *
 * Semantics restated are those of the reference's numba path: FP64 everywhere,
 * sequential summation (numba's np.sum/np.mean), no FMA contraction (build with
 * -ffp-contract=off), first-maximum tie-breaks of np.argmax.  cro_set_sum_mode(1)
 * switches every np.sum/np.mean restatement to numpy's pairwise order, which is what
 * the CPython-executed golden vectors contain (used for byte-exact NJ trees).
 *
 * exp(): cro_exp() is a table-driven FP64 exp (<0.52 ulp) written twice, here and in
 * caretta_amd/csrc/, with explicit fma() only, so that the CPU checker and the GPU
 * kernels produce bit-identical score matrices.  Building with -DCRO_LIBM_EXP uses libm's
 * exp (what numba lowers np.exp to) instead; tests quantify the difference.
 */
#ifndef CARETTA_ORACLE_H
#define CARETTA_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double gamma_tensor;   /* multiple_alignment.py:491  (7.0)  */
    double gamma_coords;   /* multiple_alignment.py:492  (0.03) */
    double gap_open;       /* bin/caretta-cli:39-41      (1.0)  */
    double gap_extend;     /* bin/caretta-cli:42-44      (0.01) */
    double sw_gap;         /* multiple_alignment.py:335  (0.0)  */
} cro_params;

/* per-pair outputs of pipeline H */
typedef struct {
    double sw;          /* smith_waterman_score on the coordinate score matrix -> NJ matrix entry */
    double dtw_score;
    double R[9];
    double t[3];
    double rmsd, coverage, tm;
    double seed_score;  /* smith_waterman score on the tensor score matrix */
    int64_t aln_len;
    int64_t seed_len;
    uint32_t flags;     /* bit0: seed superposition skipped (k<=3); bit1: metrics skipped (k<3); bit2: tensor SW all zero */
} cro_pair_out;

void cro_set_sum_mode(int mode);           /* 0 = sequential (numba), 1 = numpy pairwise */
double cro_exp(double x);

/* score_functions.py:7-11, 23-51 */
void cro_make_score_matrix(const double *a, int64_t n, const double *b, int64_t m, int64_t k,
                           double gamma, double *S);
/* score_functions.py:15-19 */
double cro_get_rmsd(const double *x1, const double *x2, int64_t k);
/* multiple_alignment.py:59-70 */
double cro_tm_score(const double *x1, const double *x2, int64_t k, int64_t l1, int64_t l2);

/* dynamic_time_warping.py:8-86, 90-144, 148-184.  S is indexed S[seq1[i]*s_cols + seq2[j]].
 * aln1/aln2 need room for n+m entries.  matrix_out/backtrack_out (nullable) receive the
 * (n+1)(m+1)3 arrays of _make_dtw_matrix. */
int cro_dtw_align(const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                  const double *S, int64_t s_cols, double gap_open, double gap_extend,
                  int64_t *aln1, int64_t *aln2, int64_t *aln_len, double *score,
                  double *matrix_out, int64_t *backtrack_out);
/* dynamic_time_warping.py:205-222 */
double cro_smith_waterman_score(const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                                const double *S, int64_t s_cols, double gap);
/* dynamic_time_warping.py:226-278.  Returns 1 when H is all zero (the reference raises). */
int cro_smith_waterman(const int64_t *seq1, int64_t n, const int64_t *seq2, int64_t m,
                       const double *S, int64_t s_cols, double gap,
                       int64_t *aln1, int64_t *aln2, int64_t *aln_len, double *score);
/* helper.py:13-42 */
int64_t cro_get_common_positions(const int64_t *a1, const int64_t *a2, int64_t len,
                                 int64_t *p1, int64_t *p2);
/* superposition_functions.py:7-35, 39-60, 64-80 */
void cro_paired_svd_superpose(const double *x1, const double *x2, int64_t k, double R[9], double t[3]);
void cro_paired_svd_superpose_with_subset(const double *c1, int64_t n, const double *c2, int64_t m,
                                          const double *s1, const double *s2, int64_t k,
                                          double *o1, double *o2, double *o3);
void cro_apply_rotran(const double *x, int64_t k, const double R[9], const double t[3], double *out);
void cro_svd3(const double C[9], double U[9], double S[3], double Vt[9]);

/* multiple_alignment.py:321-349 (Protein.score_function, flexible=False) */
uint32_t cro_protein_score_function(const double *Xi, const double *Ti, int64_t n,
                                    const double *Xj, const double *Tj, int64_t m, int64_t d,
                                    double gamma_tensor, double gamma_coords, double sw_gap,
                                    double *S, int64_t *seed1, int64_t *seed2, int64_t *seed_len,
                                    double *seed_score);
/* pipeline H for one ordered pair; aln1/aln2 need n+m entries, seed arrays (nullable) too */
void cro_pipeline_pair(const double *Xi, const double *Ti, int64_t n,
                       const double *Xj, const double *Tj, int64_t m, int64_t d,
                       const cro_params *prm, cro_pair_out *out,
                       int64_t *aln1, int64_t *aln2, int64_t *seed1, int64_t *seed2);
/* batch over a pair list on packed inputs; aln buffers are [npairs][2][aln_stride] (nullable).
 * nthreads<=1: serial (faithful to multiple_alignment.py:162-169); >1: OpenMP over pairs. */
int cro_pairwise_batch(const double *coords, const double *tensors, const int64_t *offsets,
                       int64_t d, const int32_t *pairs, int64_t npairs, const cro_params *prm,
                       cro_pair_out *outs, int64_t *aln, int64_t aln_stride, int nthreads);

/* One node of progressive_align (multiple_alignment.py:193-234: make_intermediate_node) for two Proteins:
 * score_function (:204) + consensus-weight RBF (:207-210) -> dtw_align (:211-214) -> Protein.mean_function
 * (:351-381) and get_mean_weights (:73-82).  mult1/mult2 are the multipliers of :199-202.
 * Outputs: alignment rows (n+m entries), the node's coordinates (len,3), tensors (len,d), weights (len). */
uint32_t cro_progressive_node(const double *X1, const double *T1, const double *W1, int64_t n,
                              const double *X2, const double *T2, const double *W2, int64_t m, int64_t d,
                              double mult1, double mult2, const cro_params *prm, double gamma_weight,
                              int64_t *aln1, int64_t *aln2, int64_t *aln_len,
                              double *Xn, double *Tn, double *Wn);
/* ... with flexible=True in score and mean function: tensors and consensus weights only (multiple_alignment.py:323-326, :351-362) */
void cro_progressive_node_flexible(const double *T1, const double *W1, int64_t n, const double *T2, const double *W2, int64_t m,
                                   int64_t d, double mult1, double mult2, double gamma_tensor, double gamma_weight,
                                   double gap_open, double gap_extend, int64_t *aln1, int64_t *aln2, int64_t *aln_len,
                                   double *Tn, double *Wn);

/* neighbor_joining.py:19-157.  tree: (2P-3, 2) uint64, branch_lengths: (2P-3).
 * hoist=0 recomputes the row sums inside the double loop exactly as written (O(P^4));
 * hoist=1 computes each row sum once per iteration (identical values, O(P^3)). */
int cro_neighbor_joining(const double *D, int64_t P, int hoist, uint64_t *tree, double *branch_lengths);

int cro_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
