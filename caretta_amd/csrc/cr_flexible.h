// flexible=True: the P x P matrix entry of a pair is smith_waterman_score (gap 0) of the TENSOR score matrix alone --
// Protein.score_function short-circuits to make_score_matrix(tensors, gamma_tensor) (multiple_alignment.py:323-326), and
// make_pairwise_matrix takes smith_waterman_score(arange, arange, that matrix) (:164, dynamic_time_warping.py:205-222).
// One launch over the pair list, one wave per pair: the column sweep without decisions (sweep_cols_score) over the tensor
// RBF provider.  No seed walk, no superposition, no coordinates.  Included by cr_api.hip.
#pragma once

namespace cr {

template <int R, int D>
__global__ __launch_bounds__(kWave) void k_tensor_score(const PairDesc* __restrict__ pairs, const double* __restrict__ tensors, int d,
                                                       double gamma, double* __restrict__ hand, PairResult* __restrict__ res) {
    extern __shared__ double lds[];
    const PairDesc pd = pairs[blockIdx.x];
    RbfTensor<R, D> src;
    src.rows_g = tensors + pd.off_i * d;
    src.cols_g = tensors + pd.off_j * d;
    src.d = d;
    src.neg_gamma = -gamma;
    const double sw = sweep_cols_score<R>(src, pd.n, pd.m, lds, hand + pd.hand_off);
    if (threadIdx.x == 0) {
        PairResult r;
        r.sw = sw;
        r.dtw_score = 0.0;
#pragma unroll
        for (int x = 0; x < 9; x++) r.R[x] = 0.0;
#pragma unroll
        for (int x = 0; x < 3; x++) r.t[x] = 0.0;
        r.rmsd = r.coverage = r.tm = 0.0;
        r.seed_score = sw;
        r.aln_len = r.aln_start = 0;
        r.seed_len = 0;
        r.flags = 0;
        res[blockIdx.x] = r;
    }
}

}  // namespace cr
