// Fully-connected layers on v_mfma_f32_32x32x2_f32: one strided GEMM kernel
//   C[M][N] = sum_k A(m,k) * B(k,n)       A(m,k) = a[m*sam + k*sak],  B(k,n) = b[k*sbk + n*sbn]
// covers forward (x . w^T), backward-data (dy . w) and backward-weight (dy^T . x) of nn.Linear
// (models/VGGSlim.py:68-74).  Batch is only 200, so the forward/backward-data shapes have few
// output tiles and a long K: K is split across blocks into workspace slabs that a second kernel
// sums in fixed order (deterministic) and finishes with the bias / ReLU / ReLU-mask epilogue.
#include "common.hpp"

namespace {

constexpr int TM = 64, TN = 64, BK = 32, LD = 65;

// AK: A is contiguous along k (sak == 1); BKc: B is contiguous along k (sbk == 1)
template <bool AK, bool BKc>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(
    const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
    int M, int N, int K, long sam, long sak, long sbk, long sbn, int n_tiles, int splits, int k_per_split,
    const float* __restrict__ bias, const float* __restrict__ mask_src, int relu) {
    __shared__ float as[BK * LD];
    __shared__ float bs[BK * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int tile = blockIdx.x / splits;
    const int tn = tile % n_tiles, tm = tile / n_tiles;
    const int m0 = tm * TM, n0 = tn * TN;
    const int k_begin = split * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // register-staged software pipeline: chunk t+1 is in flight while chunk t feeds the MFMAs
    constexpr int A_IT = (TM * BK) / 256, B_IT = (TN * BK) / 256;
    float ar[A_IT], br[B_IT];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int e = tid + 256 * j;
            int ml, kl;
            if (AK) { ml = e / BK; kl = e - ml * BK; } else { kl = e / TM; ml = e - kl * TM; }
            int m = m0 + ml, k = k0 + kl;
            ar[j] = (m < M && k < k_end) ? a[(long)m * sam + (long)k * sak] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int e = tid + 256 * j;
            int nl, kl;
            if (BKc) { nl = e / BK; kl = e - nl * BK; } else { kl = e / TN; nl = e - kl * TN; }
            int n = n0 + nl, k = k0 + kl;
            br[j] = (n < N && k < k_end) ? b[(long)k * sbk + (long)n * sbn] : 0.f;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int e = tid + 256 * j;
            int ml, kl;
            if (AK) { ml = e / BK; kl = e - ml * BK; } else { kl = e / TM; ml = e - kl * TM; }
            as[kl * LD + ml] = ar[j];
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int e = tid + 256 * j;
            int nl, kl;
            if (BKc) { nl = e / BK; kl = e - nl * BK; } else { kl = e / TN; nl = e - kl * TN; }
            bs[kl * LD + nl] = br[j];
        }
    };
    if (k_begin < k_end) load_chunk(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (k0 + BK < k_end) load_chunk(k0 + BK);
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
            float av = as[(k2 + kk) * LD + wm * 32 + li];
            float bv = bs[(k2 + kk) * LD + wn * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }

    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 32 + mfma32_row(r, lane);
        if (m < M && n < N) {
            float v = acc[r];
            if (splits == 1) {
                if (bias) v += bias[n];
                if (relu) v = fmaxf(v, 0.f);
                if (mask_src) v = mask_src[(size_t)m * N + n] > 0.f ? v : 0.f;
                out[(size_t)m * N + n] = v;
            } else {
                out[((size_t)split * M + m) * N + n] = v;
            }
        }
    }
}

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
    hipLaunchKernelGGL((gemm_mfma_kernel<AK, BKc>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s,
                       a, b, dst, M, N, K, sam, sak, sbk, sbn, n_tiles, splits, k_per_split, bias, mask_src, relu);
    CLHIP_LAUNCH_CHECK();
    if (splits > 1) {
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(ew_grid(mn, 256)), dim3(256), 0, s,
                           static_cast<const float*>(ws), out, mn, N, splits, bias, mask_src, relu);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

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
