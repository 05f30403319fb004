// Fully-connected layers on v_mfma_f32_32x32x2_f32: one strided GEMM kernel
//   C[M][N] = sum_k A(m,k) * B(k,n)       A(m,k) = a[m*sam + k*sak],  B(k,n) = b[k*sbk + n*sbn]
// covers forward (x . w^T), backward-data (dy . w) and backward-weight (dy^T . x) of nn.Linear
// (models/VGGSlim.py:68-74).  Batch is only 200, so the forward/backward-data shapes have few
// output tiles and a long K: K is split across blocks into workspace slabs that a second kernel
// sums in fixed order (deterministic) and finishes with the bias / ReLU / ReLU-mask epilogue.
#include "common.hpp"
#include "gemm_body.hpp"

namespace {

template <bool AK, bool BKc, int BKT = BK>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(clhip_gemm_args g) { gemm_tile<AK, BKc, BKT>(g, blockIdx.x); }

__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                 size_t mn, int N, int splits,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ mask_src, int relu) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < mn; e += stride) {
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += part[(size_t)s * mn + e];
        if (bias) v += bias[e % N];
        if (relu) v = fmaxf(v, 0.f);
        if (mask_src) v = mask_src[e] > 0.f ? v : 0.f;
        out[e] = v;
    }
}

// db[o] = sum_m dy[m][o]: 16 columns x 16 row-groups per block; each thread sums its rows in order
// (double accumulator), the 16 groups are combined in fixed order through LDS.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int M, int O) {
    __shared__ double part[16][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + c;
    const int per = (M + 15) / 16;
    const int m0 = g * per, m1 = min(M, m0 + per);
    double s = 0.0;
    if (o < O)
        for (int m = m0; m < m1; ++m) s += (double)dy[(size_t)m * O + o];
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && o < O) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += part[i][c];
        db[o] = (float)t;
    }
}

int choose_splits(int M, int N, int K) {
    int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    int s = 512 / tiles;
    int max_by_k = K / 96;           // keep >= 96 of K per split (a split costs an extra reduce launch)
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
}

template <bool AK, bool BKc>
int gemm_launch(const float* a, const float* b, float* out, int M, int N, int K, long sam, long sak, long sbk,
                long sbn, const float* bias, const float* mask_src, int relu, void* ws, size_t ws_bytes,
                hipStream_t s) {
    int splits = choose_splits(M, N, K);
    size_t mn = (size_t)M * N;
    if (splits > 1 && (!ws || ws_bytes < mn * splits * sizeof(float))) splits = 1;
    int k_per_split = (((K + splits - 1) / splits) + BK - 1) / BK * BK;
    int m_tiles = (M + TM - 1) / TM, n_tiles = (N + TN - 1) / TN;
    float* dst = splits == 1 ? out : static_cast<float*>(ws);
    const clhip_gemm_args g{a, b, dst, M, N, K, sam, sak, sbk, sbn, n_tiles, splits, k_per_split, bias, mask_src, relu};
    hipLaunchKernelGGL((gemm_mfma_kernel<AK, BKc>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, g);
    CLHIP_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(ew_grid(mn, 256)), dim3(256), 0, s,
                           static_cast<const float*>(ws), out, mn, N, splits, bias, mask_src, relu);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

// Forward GEMM of a Linear layer WITHOUT the split-K reduction: the slabs stay in ws as [split][M][O] and the caller
// (fc_tail_kernel) sums the *live slabs in ascending order, adds the bias and applies the ReLU while it stages its row
// block — exactly gemm_splitk_reduce_kernel's arithmetic (slabs past *live hold zeros).  *live = 0: the shape runs
// unsplit, nothing was launched, use clhip_fc_fwd.
int clhip_internal_fc_fwd_partial(const float* x, const float* w, int M, int I, int O, void* ws, size_t ws_bytes, int* live,
                                  hipStream_t s) {
    *live = 0;
    const int splits = choose_splits(M, O, I);
    const size_t mn = (size_t)M * O;
    if (splits <= 1 || !ws || ws_bytes < mn * splits * sizeof(float)) return 0;
    const int k_per_split = (((I + splits - 1) / splits) + BK - 1) / BK * BK;
    const int m_tiles = (M + TM - 1) / TM, n_tiles = (O + TN - 1) / TN;
    const clhip_gemm_args g{x, w, static_cast<float*>(ws), M, O, I, (long)I, 1L, 1L, (long)I, n_tiles, splits, k_per_split,
                     nullptr, nullptr, 0};
    hipLaunchKernelGGL((gemm_mfma_kernel<true, true>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, g);
    CLHIP_LAUNCH_CHECK();
    *live = (I + k_per_split - 1) / k_per_split;
    return 0;
}

// Backward-data GEMM of a Linear layer as clhip_gemm_args for a combined launch (fc_chain.hip); 0 blocks: the shape would be split
// over K, use clhip_fc_bwd_data.
int clhip_internal_fc_bwd_data_args(const float* dy, const float* w, const float* relu_src, float* dx, int M, int I, int O,
                                    clhip_gemm_args* args_out) {
    if (choose_splits(M, I, O) != 1) return 0;
    const int m_tiles = (M + TM - 1) / TM, n_tiles = (I + TN - 1) / TN;
    const int k_per_split = (O + BK - 1) / BK * BK;
    *args_out = clhip_gemm_args{dy, w, dx, M, I, O, (long)O, 1L, (long)I, 1L, n_tiles, 1, k_per_split, nullptr, relu_src, 0};
    return m_tiles * n_tiles;
}

extern "C" {

size_t clhip_fc_ws(int M, int I, int O) {
    if (M <= 0 || I <= 0 || O <= 0) return 0;
    size_t a = (size_t)M * O * choose_splits(M, O, I);   // forward
    size_t b = (size_t)M * I * choose_splits(M, I, O);   // backward-data
    size_t c = (size_t)O * I * choose_splits(O, I, M);   // backward-weight
    size_t m = a > b ? a : b;
    return (m > c ? m : c) * sizeof(float);
}

int clhip_fc_fwd(const float* x, const float* w, const float* b, float* y, int M, int I, int O, int relu,
                 void* ws, size_t ws_bytes, void* stream) {
    if (!x || !w || !y || M <= 0 || I <= 0 || O <= 0) return CLHIP_EINVAL;
    // C[M][O]: A = x (k contiguous), B(k,n) = w[n][k] (k contiguous)
    return gemm_launch<true, true>(x, w, y, M, O, I, I, 1, 1, I, b, nullptr, relu, ws, ws_bytes, as_stream(stream));
}

int clhip_fc_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int M, int I, int O,
                      void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !w || !dx || M <= 0 || I <= 0 || O <= 0) return CLHIP_EINVAL;
    // C[M][I]: A = dy (k = o contiguous), B(k = o, n = i) = w[o][i] (n contiguous)
    return gemm_launch<true, false>(dy, w, dx, M, I, O, O, 1, I, 1, nullptr, relu_src, 0, ws, ws_bytes, as_stream(stream));
}

int clhip_fc_bwd_weight(const float* x, const float* dy, float* dw, float* db, int M, int I, int O,
                        void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !dw || M <= 0 || I <= 0 || O <= 0) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    // C[O][I]: A(m = o, k = batch) = dy[k][m] (m contiguous), B(k = batch, n = i) = x[k][n] (n contiguous)
    int rc = gemm_launch<false, false>(dy, x, dw, O, I, M, 1, O, I, 1, nullptr, nullptr, 0, ws, ws_bytes, s);
    if (rc) return rc;
    if (db) {
        hipLaunchKernelGGL(colsum_kernel, dim3((O + 15) / 16), dim3(256), 0, s, dy, db, M, O);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
