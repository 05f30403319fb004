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

// ---------------------------------------------------------------------------------------------- wide layers
// 128x128 output tile for the big classifiers (AlexNet: 9216 -> 4096 -> 4096 at batch 128, models/net.py:96-125): a wave
// owns 2x2 accumulators (one LDS read per MFMA instead of two), a block moves half the operand bytes per flop of the 64x64
// tile (with 128 rows the batch is ONE row tile: the 151 MB weight matrix is read once per pass, not twice), and the
// operand that is contiguous in memory is fetched with 16-byte loads.  Same k order per output element inside a split; the
// split partition differs from the 64x64 kernel's (deterministic either way).
constexpr int WT = 128, WLD = WT + 1;

template <bool AK, bool BKc>
__global__ __launch_bounds__(256) void gemm_wide_kernel(clhip_gemm_args g) {
    const float* __restrict__ a = g.a; const float* __restrict__ b = g.b; float* __restrict__ out = g.out;
    const int M = g.M, N = g.N, K = g.K, n_tiles = g.n_tiles, splits = g.splits, k_per_split = g.k_per_split, relu = g.relu;
    const long sam = g.sam, sak = g.sak, sbk = g.sbk, sbn = g.sbn;
    const float* __restrict__ bias = g.bias; const float* __restrict__ mask_src = g.mask_src;
    __shared__ float as[BK * WLD];
    __shared__ float bs[BK * WLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int tile = blockIdx.x / splits;
    const int tn = tile % n_tiles, tm = tile / n_tiles;
    const int m0 = tm * WT, n0 = tn * WT;
    const int k_begin = split * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);

    floatx16 acc[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][c][r] = 0.f;

    // 128 x 32 floats per operand and chunk = 1024 float4 = 4 per thread.  K-contiguous operand: thread -> (row e / 8,
    // k4 = 4 (e % 8)); row-contiguous operand: thread -> (k = e / 32, row4 = 4 (e % 32)).
    float4 ar[4], br[4];
    const float4* zero4 = reinterpret_cast<const float4*>(clhip_zero16);      // out-of-range elements are loaded FROM 16 zero bytes (address select)
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            if (AK) {
                const int ml = e >> 3, k = k0 + 4 * (e & 7), m = m0 + ml;
                ar[j] = *((m < M && k < k_end) ? reinterpret_cast<const float4*>(a + (long)m * sam + k) : zero4);    // k_end % 4 == 0
            } else {
                const int kl = e >> 5, m = m0 + 4 * (e & 31), k = k0 + kl;
                ar[j] = *((m < M && k < k_end) ? reinterpret_cast<const float4*>(a + (long)k * sak + m) : zero4);    // M % 4 == 0
            }
            if (BKc) {
                const int nl = e >> 3, k = k0 + 4 * (e & 7), n = n0 + nl;
                br[j] = *((n < N && k < k_end) ? reinterpret_cast<const float4*>(b + (long)n * sbn + k) : zero4);
            } else {
                const int kl = e >> 5, n = n0 + 4 * (e & 31), k = k0 + kl;
                br[j] = *((n < N && k < k_end) ? reinterpret_cast<const float4*>(b + (long)k * sbk + n) : zero4);    // N % 4 == 0
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            if (AK) {
                float* d = as + (4 * (e & 7)) * WLD + (e >> 3);
                d[0] = ar[j].x; d[WLD] = ar[j].y; d[2 * WLD] = ar[j].z; d[3 * WLD] = ar[j].w;
            } else {
                float* d = as + (e >> 5) * WLD + 4 * (e & 31);
                d[0] = ar[j].x; d[1] = ar[j].y; d[2] = ar[j].z; d[3] = ar[j].w;
            }
            if (BKc) {
                float* d = bs + (4 * (e & 7)) * WLD + (e >> 3);
                d[0] = br[j].x; d[WLD] = br[j].y; d[2 * WLD] = br[j].z; d[3 * WLD] = br[j].w;
            } else {
                float* d = bs + (e >> 5) * WLD + 4 * (e & 31);
                d[0] = br[j].x; d[1] = br[j].y; d[2] = br[j].z; d[3] = br[j].w;
            }
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
            float av[2], bv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) av[q] = as[(k2 + kk) * WLD + q * 64 + wm * 32 + li];
#pragma unroll
            for (int c = 0; c < 2; ++c) bv[c] = bs[(k2 + kk) * WLD + c * 64 + wn * 32 + li];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[c], acc[q][c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = n0 + c * 64 + wn * 32 + li;
        if (n >= N) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + q * 64 + wm * 32 + mfma32_row(r, lane);
                if (m < M) {
                    float v = acc[q][c][r];
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
}

// wide tiles: at least one full row tile and eight column tiles, 16-byte loads possible on both operands
bool wide_ok(const float* a, const float* b, int M, int N, int K, long sam, long sak, long sbk, long sbn, bool AK, bool BKc) {
#ifdef CLHIP_NO_WIDE_GEMM
    return false;
#endif
    if (M < WT || N < 8 * WT || (K & 3)) return false;
    if (!aligned16(a) || !aligned16(b)) return false;
    if (AK ? (sam & 3) : ((sak & 3) || (M & 3))) return false;
    if (BKc ? (sbn & 3) : ((sbk & 3) || (N & 3))) return false;
    return true;
}

int choose_splits_wide(int M, int N, int K) {
    int tiles = ((M + WT - 1) / WT) * ((N + WT - 1) / WT);
    int s = 512 / tiles;
    int max_by_k = K / 128;
    if (s > max_by_k) s = max_by_k;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    return s;
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
    const bool wide = wide_ok(a, b, M, N, K, sam, sak, sbk, sbn, AK, BKc);
    int splits = wide ? choose_splits_wide(M, N, K) : choose_splits(M, N, K);
    size_t mn = (size_t)M * N;
    if (splits > 1 && (!ws || ws_bytes < mn * splits * sizeof(float))) splits = 1;
    int k_per_split = (((K + splits - 1) / splits) + BK - 1) / BK * BK;
    const int T = wide ? WT : TM;
    int m_tiles = (M + T - 1) / T, n_tiles = (N + T - 1) / T;
    float* dst = splits == 1 ? out : static_cast<float*>(ws);
    const clhip_gemm_args g{a, b, dst, M, N, K, sam, sak, sbk, sbn, n_tiles, splits, k_per_split, bias, mask_src, relu};
    if (wide) hipLaunchKernelGGL((gemm_wide_kernel<AK, BKc>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((gemm_mfma_kernel<AK, BKc>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, g);
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
    auto sp = [](int m, int n, int k) { const int u = choose_splits(m, n, k), v = choose_splits_wide(m, n, k); return u > v ? u : v; };
    size_t a = (size_t)M * O * sp(M, O, I);   // forward
    size_t b = (size_t)M * I * sp(M, I, O);   // backward-data
    size_t c = (size_t)O * I * sp(O, I, M);   // backward-weight
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
