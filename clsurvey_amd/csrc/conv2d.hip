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
#include <cstdlib>

#ifdef CLHIP_TRACE
__device__ unsigned long long* g_c2trace = nullptr;      // tuning aid: per-wave cycle sums of the loop phases
#define C2_NOW() __builtin_amdgcn_s_memtime()
#endif

namespace {

constexpr int TM = 64, TN = 64, BK = 32;
constexpr int TAB_MAX = 256;           // taps per kernel window (R*S): AlexNet max 11*11 = 121

struct ConvP { int N, C, H, W, K, R, S, st, pad, OH, OW; };

enum { OP_FWD = 0, OP_DGRAD = 1, OP_WGRAD = 2 };

// RT, CT: the block's output tile is 64 RT rows x 64 CT columns; a wave owns RT x CT 32x32 accumulators (rows wm*32 + 64 q,
// columns wn*32 + 64 c) that share RT A fragments and CT B fragments per k-pair: LDS reads per MFMA 2 -> (RT + CT) / (RT CT),
// operand bytes moved from L2 per flop likewise, and the gathered B operand (taps, bounds tests, address arithmetic: 10-18
// VALU instructions per element, issued from the same pipe as the fp32 MFMAs) is amortised over RT times the matrix work.
// Ablation of the 192 x 64 tile on AlexNet's conv2 forward (N = 128): MFMA phase alone 486 us, gathers alone 405 us
// (5.9 TB/s of 4-byte requests out of L2), together 751 us.  The k order of every output element is the same for every tile.
template <int OP, int RT, int CT>
__global__ __launch_bounds__(256, 2) void conv2d_gemm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ dy, float* __restrict__ out, ConvP p,
                                                         int M, int Nn, int Kd, int n_tiles, int splits, int k_per_split,
                                                         const float* __restrict__ bias, const float* __restrict__ mask_src,
                                                         int relu) {
    constexpr int TMR = TM * RT, LDA = TMR + 1, TNC = TN * CT, LDB = TNC + 1;
    __shared__ float as[BK * LDA];
    __shared__ float bs[BK * LDB];
    __shared__ int tab_rs[OP == OP_WGRAD ? 1 : TAB_MAX];      // tap index rs -> (r << 8) | s
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int tile = blockIdx.x / splits;
    const int tn = tile % n_tiles, tm = tile / n_tiles;
    const int m0 = tm * TMR, n0 = tn * TNC;
    const int k_begin = split * k_per_split;
    const int k_end = min(Kd, k_begin + k_per_split);
    const int RS = p.R * p.S, HW = p.H * p.W, OHW = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(x, (size_t)p.N * p.C * HW * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_w = clhip_rsrc(w, (size_t)p.K * p.C * RS * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_dy = clhip_rsrc(dy, (size_t)p.N * p.K * OHW * sizeof(float));

    if (OP != OP_WGRAD) {
        for (int e = tid; e < RS; e += 256) { const int r = e / p.S; tab_rs[e] = (r << 8) | (e - r * p.S); }
    }

    floatx16 acc[RT][CT];
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][c][r] = 0.f;

    constexpr int IT = (TN * BK) / 256;                       // 8 elements per thread and 64 rows / columns
    constexpr int ITA = IT * RT;                              // A tile
    constexpr int ITB = IT * CT;                              // B tile
    float ar[ITA], br[ITB];

    // ---- per-thread constants of the gather
    // FWD / DGRAD: B element j = (kd = k0 + tid/64 + 4j, column n0 + tid%64): the column (a pixel) is fixed
    // WGRAD:       A / B element j = (row or column = tid/32 + 8j, kd = k0 + tid%32): the rows / columns are fixed
    // (FWD / DGRAD with CT > 1: the thread owns the columns n0 + tid%64 + 64 c, B element (c, j) sits in br[c * IT + j])
    int pix_base[CT], pix_h[CT], pix_w[CT];                   // FWD: x offset of (img, c=0, ih0, iw0); DGRAD: dy image base
#pragma unroll
    for (int c = 0; c < CT; ++c) { pix_base[c] = CLHIP_OOB; pix_h[c] = 0; pix_w[c] = 0; }
    // FWD / DGRAD: reduction index of B element j as (channel, tap), stepped by BK = 32 from chunk to chunk
    int bch[IT], brs[IT];
    const int step_q = BK / RS, step_r = BK - step_q * RS;
#pragma unroll
    for (int j = 0; j < IT; ++j) {
        const int k = k_begin + (tid >> 6) + 4 * j;
        bch[j] = k / RS; brs[j] = k - bch[j] * RS;
    }
    // DGRAD: A element j = (row m0 + tid/32 + 8j, kd = k0 + tid%32): ONE reduction index per thread, so that the 32 lanes of
    // a half-wave read 32 consecutive reduction indices of one weight row — w[k][m][rs .. ] is contiguous in rs, i.e. 128-byte
    // runs instead of 64 lanes x 100-byte strides per load (the texture path, not the matrix pipe, bounded this kernel:
    // 1000 us for AlexNet's conv2 at N = 128)
    int ach = 0, ars = 0;
    if (OP == OP_DGRAD) { const int k = k_begin + (tid & 31); ach = k / RS; ars = k - ach * RS; }
    int wcol_off[ITB], wcol_rs[ITB];                          // WGRAD: (c, r, s) of this thread's 8 CT columns (r | s << 8)
    if (OP == OP_FWD) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int n = n0 + (tid & 63) + 64 * c;
            if (n < Nn) {
                const int img = n / OHW, pp = n - img * OHW, oh = pp / p.OW, ow = pp - oh * p.OW;
                pix_h[c] = oh * p.st - p.pad; pix_w[c] = ow * p.st - p.pad;
                pix_base[c] = img * p.C * HW + pix_h[c] * p.W + pix_w[c];
            }
        }
    } else if (OP == OP_DGRAD) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int n = n0 + (tid & 63) + 64 * c;
            if (n < Nn) {
                const int img = n / HW, pp = n - img * HW, h = pp / p.W, ww = pp - h * p.W;
                pix_h[c] = h + p.pad; pix_w[c] = ww + p.pad;
                pix_base[c] = img * p.K * OHW;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < ITB; ++j) {
            const int n = n0 + (tid >> 5) + 8 * j;
            wcol_off[j] = CLHIP_OOB; wcol_rs[j] = 0;
            if (n < Nn) {
                const int c = n / RS, rs = n - c * RS, r = rs / p.S, s = rs - r * p.S;
                wcol_off[j] = c * HW + r * p.W + s; wcol_rs[j] = r | (s << 8);
            }
        }
    }
    // WGRAD: the reduction index of a thread is the output pixel k0 + tid%32, advanced by 32 per chunk: (img, oh, ow) are
    // carried and stepped with one carry each (32 = qo * OW + ro) instead of two integer divisions per thread and chunk
    int w_img = 0, w_oh = 0, w_ow = 0;
    const int qo = BK / p.OW, ro = BK - qo * p.OW;
    const bool w_inc = qo + 1 <= p.OH;                        // uniform; tiny planes keep the divisions
    if (OP == OP_WGRAD) {
        const int k = k_begin + (tid & 31);
        w_img = k / OHW; const int pp = k - w_img * OHW; w_oh = pp / p.OW; w_ow = pp - w_oh * p.OW;
    }
    __syncthreads();                                          // tables

    auto load_chunk = [&](int k0) {
        if (OP == OP_FWD) {
            // A(m = k, kd = crs) = w[m][kd]: kd contiguous
#pragma unroll
            for (int j = 0; j < ITA; ++j) {
                const int e = tid + 256 * j, ml = e / BK, kl = e - ml * BK;
                const int m = m0 + ml, k = k0 + kl;
                ar[j] = clhip_buf_load(rs_w, ((m < M) & (k < k_end)) ? (m * Kd + k) * 4 : CLHIP_OOB, 0);
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int k = k0 + (tid >> 6) + 4 * j;
                const int rs = tab_rs[brs[j]], r = rs >> 8, sx = rs & 255;
                const int tap_off = bch[j] * HW + r * p.W + sx;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    // predicates combined with & (no short circuit): v_cmp + s_and + ONE v_cndmask on the offset instead of an
                    // exec-mask branch per element — the branchy form made ISSUING a chunk's gathers cost as much as its MFMAs
                    const int h = pix_h[c] + r, ww = pix_w[c] + sx;
                    const bool ok = (k < k_end) & (pix_base[c] != CLHIP_OOB) & ((unsigned)h < (unsigned)p.H) & ((unsigned)ww < (unsigned)p.W);
                    br[c * IT + j] = clhip_buf_load(rs_x, ok ? (pix_base[c] + tap_off) * 4 : CLHIP_OOB, 0);
                }
                brs[j] += step_r; bch[j] += step_q;
                if (brs[j] >= RS) { brs[j] -= RS; ++bch[j]; }
            }
        } else if (OP == OP_DGRAD) {
            // A(m = c, kd = (k, rs)) = w[k][m][rs]
            {
                const bool kok = k0 + (tid & 31) < k_end;
                const int base = (ach * p.C + m0 + (tid >> 5)) * RS + ars;
#pragma unroll
                for (int j = 0; j < ITA; ++j) {
                    const int m = m0 + (tid >> 5) + 8 * j;
                    ar[j] = clhip_buf_load(rs_w, ((m < M) & kok) ? (base + 8 * j * RS) * 4 : CLHIP_OOB, 0);
                }
                ars += step_r; ach += step_q;
                if (ars >= RS) { ars -= RS; ++ach; }
            }
#pragma unroll
            for (int j = 0; j < IT; ++j) {
                const int k = k0 + (tid >> 6) + 4 * j;
                const int rs = tab_rs[brs[j]];
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    const int th = pix_h[c] - (rs >> 8), tw = pix_w[c] - (rs & 255);
                    const bool live = (k < k_end) & (pix_base[c] != CLHIP_OOB);
                    int off;
                    if (p.st == 1) {                 // uniform branch: no integer divisions on the stride-1 layers
                        const bool ok = live & ((unsigned)th < (unsigned)p.OH) & ((unsigned)tw < (unsigned)p.OW);
                        off = ok ? (pix_base[c] + bch[j] * OHW + th * p.OW + tw) * 4 : CLHIP_OOB;
                    } else {
                        const int ths = th < 0 ? 0 : th, tws = tw < 0 ? 0 : tw;
                        const int oh = ths / p.st, ow = tws / p.st;
                        const bool ok = live & (th >= 0) & (tw >= 0) & (oh * p.st == th) & (ow * p.st == tw) & (oh < p.OH) & (ow < p.OW);
                        off = ok ? (pix_base[c] + bch[j] * OHW + oh * p.OW + ow) * 4 : CLHIP_OOB;
                    }
                    br[c * IT + j] = clhip_buf_load(rs_dy, off, 0);
                }
                brs[j] += step_r; bch[j] += step_q;
                if (brs[j] >= RS) { brs[j] -= RS; ++bch[j]; }
            }
        } else {
            // reduction index = output pixel (img, oh, ow): one decode per thread and chunk
            const int k = k0 + (tid & 31);
            int a_base = CLHIP_OOB, b_base = CLHIP_OOB, ih0 = 0, iw0 = 0;
            if (k < k_end) {
                int img, oh, ow;
                if (w_inc) { img = w_img; oh = w_oh; ow = w_ow; }
                else { img = k / OHW; const int pq = k - img * OHW; oh = pq / p.OW; ow = pq - oh * p.OW; }
                const int pp = oh * p.OW + ow;
                a_base = img * p.K * OHW + pp;
                ih0 = oh * p.st - p.pad; iw0 = ow * p.st - p.pad;
                b_base = img * p.C * HW + ih0 * p.W + iw0;
            }
            if (w_inc) {
                w_ow += ro; w_oh += qo;
                if (w_ow >= p.OW) { w_ow -= p.OW; ++w_oh; }
                if (w_oh >= p.OH) { w_oh -= p.OH; ++w_img; }
            }
#pragma unroll
            for (int j = 0; j < ITA; ++j) {
                const int m = m0 + (tid >> 5) + 8 * j;
                ar[j] = clhip_buf_load(rs_dy, ((a_base != CLHIP_OOB) & (m < M)) ? (a_base + m * OHW) * 4 : CLHIP_OOB, 0);
            }
#pragma unroll
            for (int j = 0; j < ITB; ++j) {
                const int h = ih0 + (wcol_rs[j] & 255), ww = iw0 + (wcol_rs[j] >> 8);
                const bool ok = (b_base != CLHIP_OOB) & (wcol_off[j] != CLHIP_OOB) & ((unsigned)h < (unsigned)p.H) & ((unsigned)ww < (unsigned)p.W);
                br[j] = clhip_buf_load(rs_x, ok ? (b_base + wcol_off[j]) * 4 : CLHIP_OOB, 0);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < ITA; ++j) {
            if (OP == OP_FWD) {
                const int e = tid + 256 * j, ml = e / BK, kl = e - ml * BK;
                as[kl * LDA + ml] = ar[j];
            } else {
                as[(tid & 31) * LDA + (tid >> 5) + 8 * j] = ar[j];
            }
        }
#pragma unroll
        for (int j = 0; j < ITB; ++j) {
            if (OP == OP_WGRAD) bs[(tid & 31) * LDB + (tid >> 5) + 8 * j] = br[j];
            else bs[((tid >> 6) + 4 * (j % IT)) * LDB + (tid & 63) + 64 * (j / IT)] = br[j];
        }
    };

#ifdef CLHIP_TRACE
    unsigned long long t_b1 = 0, t_st = 0, t_b2 = 0, t_ld = 0, t_mf = 0;
    const unsigned long long t_begin = C2_NOW();
#define C2_T(acc_) do { __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long n__ = C2_NOW(); acc_ += n__ - t_last; t_last = n__; } while (0)
    unsigned long long t_last = t_begin;
#else
#define C2_T(acc_) do {} while (0)
#endif
    if (k_begin < k_end) load_chunk(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        C2_T(t_mf);
        __syncthreads();
        C2_T(t_b1);
        store_chunk();
        C2_T(t_st);
        __syncthreads();
        C2_T(t_b2);
        if (k0 + BK < k_end) load_chunk(k0 + BK);
        C2_T(t_ld);
#pragma unroll
        for (int k2 = 0; k2 < BK; k2 += 2) {
            float bv[CT], av[RT];
#pragma unroll
            for (int c = 0; c < CT; ++c) bv[c] = bs[(k2 + kk) * LDB + c * 64 + wn * 32 + li];
#pragma unroll
            for (int q = 0; q < RT; ++q) av[q] = as[(k2 + kk) * LDA + q * 64 + wm * 32 + li];
#pragma unroll
            for (int q = 0; q < RT; ++q)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[q][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[c], acc[q][c], 0, 0, 0);
        }
    }

#ifdef CLHIP_TRACE
    C2_T(t_mf);
    if (g_c2trace && lane == 0 && blockIdx.x < 4096) {
        unsigned long long* t = g_c2trace + ((size_t)blockIdx.x * 4 + wave) * 8;
        t[0] = t_begin; t[1] = t_last; t[2] = t_b1; t[3] = t_st; t[4] = t_b2; t[5] = t_ld; t[6] = t_mf; t[7] = (k_end - k_begin) / BK;
    }
#endif
    // ---- epilogue: reg r of lane l = C[m0 + 64 q + wm*32 + row(r)][n0 + 64 c + wn*32 + li]
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int n = n0 + c * 64 + wn * 32 + li;
        if (n >= Nn) continue;
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
        for (int q = 0; q < RT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + q * 64 + wm * 32 + mfma32_row(r, lane);
                if (m < M) {
                    float v = acc[q][c][r];
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

// rows per block tile in units of 64: the choice that pads M least, larger on ties (192 -> 3, 256 -> 2, 384 -> 3, 64 -> 1)
int row_tiles(int M) {
    int best = 1; long best_pad = -1;
    for (int rt = 1; rt <= 3; ++rt) {
        const long pad = (long)((M + 64 * rt - 1) / (64 * rt)) * 64 * rt;
        if (best_pad < 0 || pad <= best_pad) { best = rt; best_pad = pad; }
    }
    return best;
}
// two column tiles pay off next to two or three row tiles (measured on AlexNet's layers at N = 128: weight gradients
// 5-17 % faster) and, once the gather predicates were branch-free, for forward / backward-data with one row tile too
// (conv1 forward 302 -> 274 us, conv2 backward-data 797 -> 772 us; conv1's weight gradient loses: 300 -> 326 us)
int col_tiles(int M, int Nn, bool wgrad = false) { return ((row_tiles(M) >= 2 || !wgrad) && Nn > 64) ? 2 : 1; }

int wgrad_splits(int M, int Nn, long Kd) {
    const int rt = row_tiles(M), ct = col_tiles(M, Nn, true);
    const int tiles = ((M + TM * rt - 1) / (TM * rt)) * ((Nn + TN * ct - 1) / (TN * ct));
    long s = (rt * ct > 1 ? 512 : 1024) / tiles;
    const long by_k = Kd / 256;
    if (s > by_k) s = by_k;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
}

// (the LDS-halo kernel of convkk.hip wherever its shape domain allows; the gather-GEMM serves the rest)
bool halo_kernel() { return true; }

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
    if (halo_kernel() && clhip_internal_convkk_ok(N, C, K, H, W, R, S, stride, pad))        // 5x5 on small maps: plane staged in LDS
        return clhip_internal_convkk(0, x, w, b, nullptr, y, N, C, K, H, W, relu, as_stream(stream));
    if (R * S > TAB_MAX) return CLHIP_ENOTSUP;
    const ConvP p{N, C, H, W, K, R, S, stride, pad, OH, OW};
    const int M = K, Nn = N * OH * OW, Kd = C * R * S;
    const int rt = row_tiles(M), ct = col_tiles(M, Nn);
    const int n_tiles = (Nn + TN * ct - 1) / (TN * ct), m_tiles = (M + TM * rt - 1) / (TM * rt);
#define CLHIP_C2_GO(RT_, CT_) hipLaunchKernelGGL((conv2d_gemm_kernel<OP_FWD, RT_, CT_>), dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, as_stream(stream), x, w, \
                                                 nullptr, y, p, M, Nn, Kd, n_tiles, 1, (Kd + BK - 1) / BK * BK, b, nullptr, relu)
    if (ct == 2) { if (rt == 3) CLHIP_C2_GO(3, 2); else if (rt == 2) CLHIP_C2_GO(2, 2); else CLHIP_C2_GO(1, 2); }
    else { if (rt == 3) CLHIP_C2_GO(3, 1); else if (rt == 2) CLHIP_C2_GO(2, 1); else CLHIP_C2_GO(1, 1); }
#undef CLHIP_C2_GO
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_conv2d_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int H, int W, int K,
                          int R, int S, int stride, int pad, void* stream) {
    int OH, OW;
    if (!dy || !w || !dx || !conv_ok(N, C, H, W, K, R, S, stride, pad, OH, OW)) return CLHIP_EINVAL;
    if (halo_kernel() && clhip_internal_convkk_ok(N, K, C, H, W, R, S, stride, pad))        // (stride 1, pad K/2: OH = H, OW = W)
        return clhip_internal_convkk(1, dy, w, nullptr, relu_src, dx, N, K, C, H, W, 0, as_stream(stream));
    if (R * S > TAB_MAX) return CLHIP_ENOTSUP;
    const ConvP p{N, C, H, W, K, R, S, stride, pad, OH, OW};
    const int M = C, Nn = N * H * W, Kd = K * R * S;
    const int rt = row_tiles(M), ct = col_tiles(M, Nn);
    const int n_tiles = (Nn + TN * ct - 1) / (TN * ct), m_tiles = (M + TM * rt - 1) / (TM * rt);
#define CLHIP_C2_GO(RT_, CT_) hipLaunchKernelGGL((conv2d_gemm_kernel<OP_DGRAD, RT_, CT_>), dim3((unsigned)(m_tiles * n_tiles)), dim3(256), 0, as_stream(stream), nullptr, \
                                                 w, dy, dx, p, M, Nn, Kd, n_tiles, 1, (Kd + BK - 1) / BK * BK, nullptr, relu_src, 0)
    if (ct == 2) { if (rt == 3) CLHIP_C2_GO(3, 2); else if (rt == 2) CLHIP_C2_GO(2, 2); else CLHIP_C2_GO(1, 2); }
    else { if (rt == 3) CLHIP_C2_GO(3, 1); else if (rt == 2) CLHIP_C2_GO(2, 1); else CLHIP_C2_GO(1, 1); }
#undef CLHIP_C2_GO
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
    const int rt = row_tiles(M), ct = col_tiles(M, Nn, true);
    const int n_tiles = (Nn + TN * ct - 1) / (TN * ct), m_tiles = (M + TM * rt - 1) / (TM * rt);
    hipStream_t s = as_stream(stream);
#define CLHIP_C2_GO(RT_, CT_) hipLaunchKernelGGL((conv2d_gemm_kernel<OP_WGRAD, RT_, CT_>), dim3((unsigned)(m_tiles * n_tiles * splits)), dim3(256), 0, s, x, nullptr, dy, \
                                                 static_cast<float*>(ws), p, M, Nn, (int)Kd, n_tiles, splits, k_per_split, nullptr, nullptr, 0)
    if (ct == 2) { if (rt == 3) CLHIP_C2_GO(3, 2); else if (rt == 2) CLHIP_C2_GO(2, 2); else CLHIP_C2_GO(1, 2); }
    else { if (rt == 3) CLHIP_C2_GO(3, 1); else if (rt == 2) CLHIP_C2_GO(2, 1); else CLHIP_C2_GO(1, 1); }
#undef CLHIP_C2_GO
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

#ifdef CLHIP_TRACE
int clhip_debug_set_conv2d_trace(void* p) {
    unsigned long long* q = static_cast<unsigned long long*>(p);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_c2trace), &q, sizeof(q));
}
#endif

}  // extern "C"
