// General 2-D convolution (any kernel size / stride / zero padding) for the AlexNet layers of BASELINE config 4
// (torchvision alexnet as pickled by models/net.py:96-125: 11x11 stride 4, 5x5 pad 2, 3x3 pad 1) — forward,
// backward-data and backward-weight as gather-GEMMs on v_mfma_f32_32x32x2_f32, NCHW in and out, no im2col buffer.
//
//   forward   y[n][k][oh][ow] = b[k] + sum_{c,r,s} w[k][c][r][s] x[n][c][oh*st - pad + r][ow*st - pad + s]
//             rows = k, cols = output pixels (n, oh, ow), reduction = (c, r, s)
//   bwd-data  dx[n][c][h][w]  = sum_{k,r,s : (h + pad - r) % st == 0, ...} w[k][c][r][s] dy[n][k][(h+pad-r)/st][(w+pad-s)/st]
//             rows = c, cols = input pixels, reduction = (k, r, s)
//   bwd-weight dw[k][c][r][s] = sum_{n,oh,ow} dy[n][k][oh][ow] x[n][c][oh*st - pad + r][ow*st - pad + s]
//             rows = k, cols = (c, r, s), reduction = output pixels, split over blocks + fixed-order reduction
//
// 64x64 output tile per block (4 waves x one 32x32 accumulator), 32-deep reduction chunks staged through LDS with
// register prefetch.  Operands are fetched with raw buffer loads: a tap that falls into the padding (or past a tile
// edge) is simply an out-of-range offset and reads 0.  The reduction index (channel, tap) of an element is carried in
// registers and stepped by 32 per chunk (no divisions in the loop), pixel decodes are done once per thread.
// The 3x3 pad-1 stride-1 VGG layers never come here (conv3x3.hip is 2-3x faster on them).
#include "common.hpp"

namespace {

constexpr int TM = 64, TN = 64, BK = 32, LD = 65;
constexpr int TAB_MAX = 256;           // taps per kernel window (R*S): AlexNet max 11*11 = 121

struct ConvP { int N, C, H, W, K, R, S, st, pad, OH, OW; };

enum { OP_FWD = 0, OP_DGRAD = 1, OP_WGRAD = 2 };

template <int OP>
__global__ __launch_bounds__(256) void conv2d_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dy, float* __restrict__ out, ConvP p,
                                                         int M, int Nn, int Kd, int n_tiles, int splits, int k_per_split,
                                                         const float* __restrict__ bias, const float* __restrict__ mask_src,
                                                         int relu) {
    __shared__ float as[BK * LD];
    __shared__ float bs[BK * LD];
    __shared__ int tab_rs[OP == OP_WGRAD ? 1 : TAB_MAX];      // tap index rs -> (r << 8) | s
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int tile = blockIdx.x / splits;
    const int tn = tile % n_tiles, tm = tile / n_tiles;
    const int m0 = tm * TM, n0 = tn * TN;
    const int k_begin = split * k_per_split;
    const int k_end = min(Kd, k_begin + k_per_split);
    const int RS = p.R * p.S, HW = p.H * p.W, OHW = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(x, (size_t)p.N * p.C * HW * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_w = clhip_rsrc(w, (size_t)p.K * p.C * RS * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_dy = clhip_rsrc(dy, (size_t)p.N * p.K * OHW * sizeof(float));

    if (OP != OP_WGRAD) {
        for (int e = tid; e < RS; e += 256) { const int r = e / p.S; tab_rs[e] = (r << 8) | (e - r * p.S); }
    }

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    constexpr int IT = (TM * BK) / 256;                       // 8 elements of each tile per thread
    float ar[IT], br[IT];

    // ---- per-thread constants of the gather
    // FWD / DGRAD: B element j = (kd = k0 + tid/64 + 4j, column n0 + tid%64): the column (a pixel) is fixed
    // WGRAD:       A / B element j = (row or column = tid/32 + 8j, kd = k0 + tid%32): the rows / columns are fixed
    int pix_base = CLHIP_OOB, pix_h = 0, pix_w = 0;          // FWD: x offset of (img, c=0, ih0, iw0); DGRAD: dy image base
    // FWD / DGRAD: reduction index of B element j as (channel, tap), stepped by BK = 32 from chunk to chunk
    int bch[IT], brs[IT];
    const int step_q = BK / RS, step_r = BK - step_q * RS;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int k = k_begin + (tid >> 6) + 4 * j;
        bch[j] = k / RS; brs[j] = k - bch[j] * RS;
    }
    int wcol_off[IT], wcol_r[IT], wcol_s[IT];                 // WGRAD: (c, r, s) of this thread's 8 columns
    if (OP == OP_FWD) {
        const int n = n0 + (tid & 63);
        if (n < Nn) {
            const int img = n / OHW, pp = n - img * OHW, oh = pp / p.OW, ow = pp - oh * p.OW;
            pix_h = oh * p.st - p.pad; pix_w = ow * p.st - p.pad;
            pix_base = img * p.C * HW + pix_h * p.W + pix_w;
        }
    } else if (OP == OP_DGRAD) {
        const int n = n0 + (tid & 63);
        if (n < Nn) {
            const int img = n / HW, pp = n - img * HW, h = pp / p.W, ww = pp - h * p.W;
            pix_h = h + p.pad; pix_w = ww + p.pad;
            pix_base = img * p.K * OHW;
        }
    } else {
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int n = n0 + (tid >> 5) + 8 * j;
            wcol_off[j] = CLHIP_OOB; wcol_r[j] = 0; wcol_s[j] = 0;
            if (n < Nn) {
                const int c = n / RS, rs = n - c * RS, r = rs / p.S, s = rs - r * p.S;
                wcol_off[j] = c * HW + r * p.W + s; wcol_r[j] = r; wcol_s[j] = s;
            }
        }
    }
    __syncthreads();                                          // tables

    auto load_chunk = [&](int k0) {
        if (OP == OP_FWD) {
            // A(m = k, kd = crs) = w[m][kd]: kd contiguous
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int e = tid + 256 * j, ml = e / BK, kl = e - ml * BK;
                const int m = m0 + ml, k = k0 + kl;
                ar[j] = clhip_buf_load(rs_w, (m < M && k < k_end) ? (m * Kd + k) * 4 : CLHIP_OOB, 0);
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int k = k0 + (tid >> 6) + 4 * j;
                int off = CLHIP_OOB;
                if (k < k_end && pix_base != CLHIP_OOB) {
                    const int rs = tab_rs[brs[j]], r = rs >> 8, sx = rs & 255;
                    const int h = pix_h + r, ww = pix_w + sx;
                    if ((unsigned)h < (unsigned)p.H && (unsigned)ww < (unsigned)p.W) off = (pix_base + bch[j] * HW + r * p.W + sx) * 4;
                }
                br[j] = clhip_buf_load(rs_x, off, 0);
                brs[j] += step_r; bch[j] += step_q;
                if (brs[j] >= RS) { brs[j] -= RS; ++bch[j]; }
            }
        } else if (OP == OP_DGRAD) {
            // A(m = c, kd = (k, rs)) = w[k][m][rs]
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int e = tid + 256 * j, kl = e / TM, ml = e - kl * TM;
                const int m = m0 + ml, k = k0 + kl;
                // kl = tid / 64 + 4 j: the same reduction index as B element j, whose (channel, tap) is carried in registers
                const int off = (m < M && k < k_end) ? ((bch[j] * p.C + m) * RS + brs[j]) * 4 : CLHIP_OOB;
                ar[j] = clhip_buf_load(rs_w, off, 0);
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int k = k0 + (tid >> 6) + 4 * j;
                int off = CLHIP_OOB;
                if (k < k_end && pix_base != CLHIP_OOB) {
                    const int rs = tab_rs[brs[j]];
                    const int th = pix_h - (rs >> 8), tw = pix_w - (rs & 255);
                    if (p.st == 1) {                 // uniform branch: no integer divisions on the stride-1 layers
                        if ((unsigned)th < (unsigned)p.OH && (unsigned)tw < (unsigned)p.OW)
                            off = (pix_base + bch[j] * OHW + th * p.OW + tw) * 4;
                    } else if (th >= 0 && tw >= 0) {
                        const int oh = th / p.st, ow = tw / p.st;
                        if (oh * p.st == th && ow * p.st == tw && oh < p.OH && ow < p.OW)
                            off = (pix_base + bch[j] * OHW + oh * p.OW + ow) * 4;
                    }
                }
                br[j] = clhip_buf_load(rs_dy, off, 0);
                brs[j] += step_r; bch[j] += step_q;
                if (brs[j] >= RS) { brs[j] -= RS; ++bch[j]; }
            }
        } else {
            // reduction index = output pixel (img, oh, ow): one decode per thread and chunk
            const int k = k0 + (tid & 31);
            int a_base = CLHIP_OOB, b_base = CLHIP_OOB, ih0 = 0, iw0 = 0;
            if (k < k_end) {
                const int img = k / OHW, pp = k - img * OHW, oh = pp / p.OW, ow = pp - oh * p.OW;
                a_base = img * p.K * OHW + pp;
                ih0 = oh * p.st - p.pad; iw0 = ow * p.st - p.pad;
                b_base = img * p.C * HW + ih0 * p.W + iw0;
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int m = m0 + (tid >> 5) + 8 * j;
                ar[j] = clhip_buf_load(rs_dy, (a_base != CLHIP_OOB && m < M) ? (a_base + m * OHW) * 4 : CLHIP_OOB, 0);
                int off = CLHIP_OOB;
                if (b_base != CLHIP_OOB && wcol_off[j] != CLHIP_OOB) {
                    const int h = ih0 + wcol_r[j], ww = iw0 + wcol_s[j];
                    if ((unsigned)h < (unsigned)p.H && (unsigned)ww < (unsigned)p.W) off = (b_base + wcol_off[j]) * 4;
                }
                br[j] = clhip_buf_load(rs_x, off, 0);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int e = tid + 256 * j;
            if (OP == OP_FWD) {
                const int ml = e / BK, kl = e - ml * BK;
                as[kl * LD + ml] = ar[j];
                bs[((tid >> 6) + 4 * j) * LD + (tid & 63)] = br[j];
            } else if (OP == OP_DGRAD) {
                const int kl = e / TM, ml = e - kl * TM;
                as[kl * LD + ml] = ar[j];
                bs[((tid >> 6) + 4 * j) * LD + (tid & 63)] = br[j];
            } else {
                as[(tid & 31) * LD + (tid >> 5) + 8 * j] = ar[j];
                bs[(tid & 31) * LD + (tid >> 5) + 8 * j] = br[j];
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
            const float av = as[(k2 + kk) * LD + wm * 32 + li];
            const float bv = bs[(k2 + kk) * LD + wn * 32 + li];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }

    // ---- epilogue: reg r of lane l = C[m0 + wm*32 + row(r)][n0 + wn*32 + li]
    const int n = n0 + wn * 32 + li;
    if (n >= Nn) return;
    size_t col_off;                                            // offset of (row 0, column n) in the output tensor
    size_t row_stride;
    if (OP == OP_FWD) {
        const int img = n / OHW, pp = n - img * OHW;
        col_off = (size_t)img * p.K * OHW + pp; row_stride = OHW;
    } else if (OP == OP_DGRAD) {
        const int img = n / HW, pp = n - img * HW;
        col_off = (size_t)img * p.C * HW + pp; row_stride = HW;
    } else {
        col_off = (size_t)split * M * Nn + n; row_stride = Nn;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + mfma32_row(r, lane);
        if (m < M) {
            float v = acc[r];
            const size_t o = col_off + (size_t)m * row_stride;
            if (OP == OP_FWD) {
                if (bias) v += bias[m];
                if (relu) v = fmaxf(v, 0.f);
            } else if (OP == OP_DGRAD) {
                if (mask_src) v = mask_src[o] > 0.f ? v : 0.f;
            }
            out[o] = v;
        }
    }
}

// dw = sum of the split slabs (fixed order); db[k] = sum_{n, pixels} dy[n][k][.] (f64, fixed order)
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  size_t mn, int splits) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < mn; e += stride) {
        double v = 0.0;
        for (int s = 0; s < splits; ++s) v += (double)part[(size_t)s * mn + e];
        dw[e] = (float)v;
    }
}

// db[k] = sum over images and pixels of dy: BIAS_SPLIT blocks per channel (image ranges) write f64 partials, a second
// tiny launch adds them in order (deterministic).  One block per channel was 340 us on AlexNet's conv1 (64 blocks, one
// load in flight per thread).
constexpr int BIAS_SPLIT = 16;

__global__ __launch_bounds__(256) void conv2d_bias_grad_kernel(const float* __restrict__ dy, double* __restrict__ part,
                                                               int N, int K, int OHW) {
    __shared__ double red[256];
    const int k = blockIdx.x, sp = blockIdx.y;
    const int n0 = (int)((long)N * sp / BIAS_SPLIT), n1 = (int)((long)N * (sp + 1) / BIAS_SPLIT);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int img = n0; img < n1; ++img) {
        const float* row = dy + ((size_t)img * K + k) * OHW;
        int e = threadIdx.x;
        for (; e + 768 < OHW; e += 1024) {
            const float a = row[e], b = row[e + 256], c = row[e + 512], d = row[e + 768];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; e < OHW; e += 256) s0 += row[e];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)k * BIAS_SPLIT + sp] = red[0];
}

__global__ void conv2d_bias_finish_kernel(const double* __restrict__ part, float* __restrict__ db, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (int i = 0; i < BIAS_SPLIT; ++i) s += part[(size_t)k * BIAS_SPLIT + i];
    db[k] = (float)s;
}

bool conv_ok(int N, int C, int H, int W, int K, int R, int S, int st, int pad, int& OH, int& OW) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || R <= 0 || S <= 0 || st <= 0 || pad < 0) return false;
    if (R > 255 || S > 255) return false;
    OH = (H + 2 * pad - R) / st + 1; OW = (W + 2 * pad - S) / st + 1;
    if (OH <= 0 || OW <= 0) return false;
    if ((long long)N * C * H * W > 0x1fffffffLL || (long long)N * K * OH * OW > 0x1fffffffLL) return false;   // 32-bit byte offsets
    return true;
}

int wgrad_splits(int M, int Nn, long Kd) {
    const int tiles = ((M + TM - 1) / TM) * ((Nn + TN - 1) / TN);
    long s = 1024 / tiles;
    const long by_k = Kd / 256;
    if (s > by_k) s = by_k;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

extern "C" {

size_t clhip_conv2d_bwd_weight_ws(int N, int C, int H, int W, int K, int R, int S, int stride, int pad) {
    int OH, OW;
    if (!conv_ok(N, C, H, W, K, R, S, stride, pad, OH, OW)) return 0;
    const size_t slabs = (size_t)K * C * R * S * wgrad_splits(K, C * R * S, (long)N * OH * OW) * sizeof(float);
    const size_t bias = (size_t)K * BIAS_SPLIT * sizeof(double);     // reused after the slab reduction (stream order)
    return slabs > bias ? slabs : bias;
}

int clhip_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R, int S,
                     int stride, int pad, int relu, void* stream) {
    int OH, OW;
    if (!x || !w || !y || !conv_ok(N, C, H, W, K, R, S, stride, pad, OH, OW)) return CLHIP_EINVAL;
    if (R * S > TAB_MAX) return CLHIP_ENOTSUP;
    const ConvP p{N, C, H, W, K, R, S, stride, pad, OH, OW};
    const int M = K, Nn = N * OH * OW, Kd = C * R * S;
    const int n_tiles = (Nn + TN - 1) / TN, m_tiles = (M + TM - 1) / TM;
    hipLaunchKernelGGL((conv2d_gemm_kernel<OP_FWD>), dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, as_stream(stream), x, w,
                       nullptr, y, p, M, Nn, Kd, n_tiles, 1, (Kd + BK - 1) / BK * BK, b, nullptr, relu);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_conv2d_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int H, int W, int K,
                          int R, int S, int stride, int pad, void* stream) {
    int OH, OW;
    if (!dy || !w || !dx || !conv_ok(N, C, H, W, K, R, S, stride, pad, OH, OW)) return CLHIP_EINVAL;
    if (R * S > TAB_MAX) return CLHIP_ENOTSUP;
    const ConvP p{N, C, H, W, K, R, S, stride, pad, OH, OW};
    const int M = C, Nn = N * H * W, Kd = K * R * S;
    const int n_tiles = (Nn + TN - 1) / TN, m_tiles = (M + TM - 1) / TM;
    hipLaunchKernelGGL((conv2d_gemm_kernel<OP_DGRAD>), dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, as_stream(stream), nullptr,
                       w, dy, dx, p, M, Nn, Kd, n_tiles, 1, (Kd + BK - 1) / BK * BK, nullptr, relu_src, 0);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int H, int W, int K, int R,
                            int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    int OH, OW;
    if (!x || !dy || !dw || !ws || !conv_ok(N, C, H, W, K, R, S, stride, pad, OH, OW)) return CLHIP_EINVAL;
    const ConvP p{N, C, H, W, K, R, S, stride, pad, OH, OW};
    const int M = K, Nn = C * R * S;
    const long Kd = (long)N * OH * OW;
    const int splits = wgrad_splits(M, Nn, Kd);
    const size_t mn = (size_t)M * Nn;
    if (ws_bytes < mn * splits * sizeof(float) || (db && ws_bytes < (size_t)K * BIAS_SPLIT * sizeof(double))) return CLHIP_ENOSPC;
    const int k_per_split = (int)(((Kd + splits - 1) / splits + BK - 1) / BK * BK);
    const int n_tiles = (Nn + TN - 1) / TN, m_tiles = (M + TM - 1) / TM;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL((conv2d_gemm_kernel<OP_WGRAD>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, x, nullptr, dy,
                       static_cast<float*>(ws), p, M, Nn, (int)Kd, n_tiles, splits, k_per_split, nullptr, nullptr, 0);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3(ew_grid(mn, 256)), dim3(256), 0, s, static_cast<const float*>(ws), dw, mn, splits);
    CLHIP_LAUNCH_CHECK();
    if (db) {
        double* part = static_cast<double*>(ws);       // the slabs are consumed: same stream, after the reduction
        hipLaunchKernelGGL(conv2d_bias_grad_kernel, dim3(K, BIAS_SPLIT), dim3(256), 0, s, dy, part, N, K, OH * OW);
        CLHIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(conv2d_bias_finish_kernel, dim3((K + 255) / 256), dim3(256), 0, s, part, db, K);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

}  // extern "C"
