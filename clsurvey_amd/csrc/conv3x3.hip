// 3x3 pad-1 stride-1 convolution for gfx950: implicit GEMM on v_mfma_f32_32x32x2_f32.
//
//   forward      y[n][k][h][w]  = relu?(b[k] + sum_{c,r,s} x[n][c][h+r-1][w+s-1] * wt[k][c][r][s])
//   bwd (data)   dx[n][c][h][w] = mask * sum_{k,r',s'} dy[n][k][h+r'-1][w+s'-1] * wt[k][c][2-r'][2-s']
//   bwd (weight) dw[k][c][r][s] = sum_{n,h,w} dy[n][k][h][w] * x[n][c][h+r-1][w+s-1]   (conv3x3_wgrad.hip)
//
// Layout is torch's own NCHW / KCRS: no repacking passes, and an NCHW plane row is what a
// wave stores (32 consecutive pixels of one channel per accumulator register => 128 B lines).
//
// GEMM view (D = A.B per MFMA, 32x32 output, K = 2):
//   D rows  (A operand, from LDS weight tile)      : 32 output channels
//   D cols  (B operand, from LDS activation tile)  : 32 pixels of the block's spatial tile
//   K pair                                         : two input channels (c, c+1), same tap (r,s)
// The activation tile is staged ONCE per channel chunk with its 1-pixel halo, so the nine taps
// read the same LDS plane at nine immediate offsets (no im2col buffer, 9x less HBM/L2 traffic).
//
// Block = 256 threads = 4 waves = (2 halves of 64 output channels) x (2 halves of BP pixels).
// LDS per buffer: weights [CK*9][65] + activations [CK][PLANE]; two buffers, register-staged
// prefetch of chunk t+1 while chunk t feeds the matrix pipe (one barrier per chunk).
#include "common.hpp"

namespace {

#ifdef CLHIP_TRACE
// tools/trace_conv.py only (never in the product build): per-wave cycle stamps of the chunked kernel
__device__ unsigned long long* g_trace = nullptr;
#define TR_NOW() __builtin_amdgcn_s_memtime()
#endif

constexpr int KT = 64;     // output channels per block
constexpr int LDW = 65;    // weight-tile row stride (odd: conflict-free transposing ds_write)

template <int TW, int TH, int NB>
struct Geo {
    static constexpr int BP = TW * TH * NB;          // pixels per block tile
    static constexpr int TWP = TW + 2;               // halo row length
    static constexpr int PLANE = NB * (TH + 2) * TWP;  // floats per staged channel plane
    static constexpr int NT = (BP + 63) / 64;        // 32-pixel subtiles per wave
    // DENSE: the tile is one whole TW x TH plane whose pixels are dealt to the lanes in linear order (pixel q = 32 s + li,
    // slots past TW*TH idle) instead of as rows of a power-of-two rectangle: a 13x13 plane (AlexNet's 3x3 layers,
    // models/net.py:96-125) fills 169 of 192 slots; 16x8 tiles covered 169 of 256.
    static constexpr bool DENSE = (32 % TW != 0 && TW != 32);
    static_assert(DENSE || BP == 64 || BP == 128, "block tile must hold 64 or 128 pixels");
    static_assert(!DENSE || (NB == 1 && BP <= 192), "dense tiles: one plane of at most 192 pixels");
};

// Subtile s (32 pixels) / lane li -> pixel of the block tile.  With a fused 2x2 max-pool every 2x2 window must
// sit inside ONE 32-pixel subtile so the four candidates are in four lanes of the same accumulator register:
// the natural mapping already does that for TW = 16 (2 rows x 16) and TW = 8 (4 rows x 8); for TW = 32 the
// pooled mapping uses 2 rows x 16 columns per subtile instead of 1 row x 32.
template <int TW, int TH>
__device__ __forceinline__ void tile_pixel(int s, int li, bool pool, int& nb, int& th, int& tw) {
    if (pool && TW == 32) {
        nb = 0; th = 2 * (s >> 1) + (li >> 4); tw = 16 * (s & 1) + (li & 15);
    } else {
        const int q = 32 * s + li;
        tw = q % TW; th = (q / TW) % TH; nb = q / (TW * TH);
        if ((32 % TW != 0 && TW != 32) && q >= TW * TH) { nb = 0; th = TH; tw = 0; }     // idle slot of a dense tile: row TH is outside the image (never stored) but inside the halo plane
    }
}

// MODE 0: forward (wt used as is, bias+relu epilogue)
// MODE 1: backward-data (in = dy with Kw channels, out = dx with Cw channels, taps flipped)
#ifndef CLHIP_CONV_MIN_WAVES
#define CLHIP_CONV_MIN_WAVES 1
#endif
// UNPOOL (MODE 1, VEC only): `in` is the gradient w.r.t. the 2x2-max-POOLED output [N][Cin][H/2][W/2] and `pool_idx` the
// 2-bit arg-max codes of the forward pass; the un-pooled gradient tile is rebuilt while it is staged into LDS (fused
// max_pool2d backward: the 4x larger tensor is never written nor read, and its kernel launch disappears).
template <int TW, int TH, int NB, int CK>
struct ConvLds {
    static constexpr int BUF_FLOATS = CK * 9 * LDW + CK * Geo<TW, TH, NB>::PLANE;
    static constexpr int FLOATS = 2 * BUF_FLOATS;
};

// The kernel body for block `bid` of a launch whose images start at `img0` (see conv3x3_mfma_mixed_kernel); `lds` and
// `bias_s` are the launching kernel's shared arrays.
template <int TW, int TH, int NB, int CK, int MODE, bool VEC, bool UNPOOL = false>
__device__ __forceinline__ void conv3x3_mfma_body(
    const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out,
    int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu,
    int tiles_w, int tiles_h, int n_pix_tiles, uint8_t* __restrict__ pool_idx,
    const int bid, const int img0, float* __restrict__ lds, float* __restrict__ bias_s) {
    using G = Geo<TW, TH, NB>;
    static_assert(!UNPOOL || (MODE == 1 && VEC), "the fused un-pool lives in the 16-byte staging of backward-data");
    const bool pool = !UNPOOL && pool_idx != nullptr;     // MODE 0 only: out = 2x2-max-pooled relu(conv), idx = argmax
    constexpr int WS_FLOATS = CK * 9 * LDW;
    constexpr int XS_FLOATS = CK * G::PLANE;
    constexpr int BUF_FLOATS = WS_FLOATS + XS_FLOATS;
    static_assert(BUF_FLOATS == ConvLds<TW, TH, NB, CK>::BUF_FLOATS, "shared-array size of the launching kernel");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
#ifdef CLHIP_TRACE
    const unsigned long long tr_start = TR_NOW();
#endif
    const int wk = wave & 1;        // which 32 output channels
    const int wp = wave >> 1;       // which half of the pixels
    const int li = lane & 31;
    const int kk = lane >> 5;

    // block -> (output-channel tile, pixel tile); pixel tile fastest so concurrently resident
    // blocks share one weight slice in L2.
    const int kt = bid / n_pix_tiles;
    const int pt = bid - kt * n_pix_tiles;
    const int tw_i = pt % tiles_w;
    const int th_i = (pt / tiles_w) % tiles_h;
    const int ng = pt / (tiles_w * tiles_h);
    const int n0 = img0 + ng * NB, h0 = th_i * TH, w0 = tw_i * TW;
    const int ko0 = kt * KT;        // first output channel of this block
    if (MODE == 0 && threadIdx.x < KT) bias_s[threadIdx.x] = (bias && ko0 + (int)threadIdx.x < Cout) ? bias[ko0 + threadIdx.x] : 0.f;

    constexpr int W_ELEMS = KT * CK * 9;
    const size_t plane_hw = (size_t)H * W;
    const int n_chunks = (Cin + CK - 1) / CK;
    const int Wp = W >> 1;
    const size_t plane_in = UNPOOL ? (size_t)(H >> 1) * Wp : plane_hw;       // plane of the tensor `in` points at
    const float* in_blk = in + (size_t)n0 * Cin * plane_in;

    // ------------------------------------------------------------------ staging
    // All index math is done ONCE per block; per chunk only base pointers move.  Every load is
    // unconditional + a select on the OFFSET (no exec-mask branches, nothing waits on a load).
    //  VEC (aligned shapes: Cin % 8 == 0, W % 4 == 0, W % TW == 0, Cw % 4 == 0): 16-byte global loads
    //      — weights 4.5 / activations 1.5-2 (+halo columns) instructions per thread per chunk;
    //  scalar path: first layer (C = 3) and odd shapes.
    // A chunk's staging work is cut into UNITS (one global load + the LDS writes of its result); the main loop
    // issues them one at a time in the shadow of the MFMAs (see "pipeline" below).
    constexpr int W_IT = VEC ? (W_ELEMS / 4 + 255) / 256 : (W_ELEMS + 255) / 256;
    constexpr int XROWS = CK * NB * (TH + 2);                 // halo-plane rows per chunk
    constexpr int XV_ELEMS = XROWS * (TW / 4);                // interior float4 per chunk
    constexpr int X_IT = VEC ? (XV_ELEMS + 255) / 256 : (XS_FLOATS + 255) / 256;
    constexpr int H_IT = VEC ? (XROWS * 2 + 255) / 256 : 0;   // halo-column scalars
    constexpr int NUNITS = W_IT + X_IT + H_IT;
    float4 wv[VEC ? W_IT : 1];
    float4 xv[VEC ? X_IT : 1];
    float hv[VEC ? (H_IT > 0 ? H_IT : 1) : 1];
    float wreg[VEC ? 1 : W_IT];
    float xreg[VEC ? 1 : X_IT];
    int woff[W_IT], wdst[VEC ? W_IT * (MODE == 0 ? 1 : 4) : W_IT];
    int xoff[X_IT], xdst[VEC ? X_IT : 1];
    int hoff[H_IT > 0 ? H_IT : 1], hdst[H_IT > 0 ? H_IT : 1];
    int wch[VEC ? 1 : W_IT], xch[VEC ? 1 : X_IT];             // scalar path: channel-in-chunk (tail chunks)
    // UNPOOL: per unit the pooled pair / element, its arg-max byte(s), the byte offset into the idx tensor and the code
    // (row parity * 2 [+ column parity]) an element must carry to receive the gradient
    float2 xp[UNPOOL ? X_IT : 1];
    unsigned xi[UNPOOL ? X_IT : 1], hi[UNPOOL && H_IT > 0 ? H_IT : 1];
    int xioff[UNPOOL ? X_IT : 1], xcode[UNPOOL ? X_IT : 1], hioff[UNPOOL && H_IT > 0 ? H_IT : 1], hcode[UNPOOL && H_IT > 0 ? H_IT : 1];

    // ---- part A: global offsets only, so that chunk 0 is in flight while the rest of the index math runs.
    // Raw buffer loads: out-of-range / halo elements get voffset = CLHIP_OOB and read back as 0 from the hardware
    // range check (no select on the loaded value), the per-chunk base is the scalar offset of the instruction.
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int e = tid + 256 * j;
            woff[j] = CLHIP_OOB;
            if (MODE == 0) {
                const int kl = e / (CK * 9 / 4), f = e - kl * (CK * 9 / 4);
                if (e < W_ELEMS / 4 && ko0 + kl < Kw) woff[j] = ((ko0 + kl) * Cw * 9 + 4 * f) * 4;
            } else {
                const int kl = e / (KT * 9 / 4), f = e - kl * (KT * 9 / 4);
                if (e < W_ELEMS / 4 && 4 * f < (Cw - ko0) * 9) woff[j] = ((kl * Cw + ko0) * 9 + 4 * f) * 4;
            }
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e / (TW / 4), f = e - rowid * (TW / 4);
            const int cl = rowid / (NB * (TH + 2)), rr = rowid - cl * (NB * (TH + 2));
            const int nb = rr / (TH + 2), row = rr - nb * (TH + 2);
            const int n = n0 + nb, h = h0 - 1 + row;
            xoff[j] = CLHIP_OOB;
            if constexpr (UNPOOL) {
                xioff[j] = CLHIP_OOB;
                xcode[j] = (h & 1) << 1;
                if (e < XV_ELEMS && n < N && h >= 0 && h < H) {
                    xioff[j] = (int)(((size_t)nb * Cin + cl) * plane_in) + (h >> 1) * Wp + ((w0 + 4 * f) >> 1);
                    xoff[j] = xioff[j] * 4;
                }
            } else if (e < XV_ELEMS && n < N && h >= 0 && h < H) {
                xoff[j] = ((int)(((size_t)nb * Cin + cl) * plane_hw) + h * W + w0 + 4 * f) * 4;
            }
        }
#pragma unroll
        for (int j = 0; j < H_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e >> 1, side = e & 1;
            const int cl = rowid / (NB * (TH + 2)), rr = rowid - cl * (NB * (TH + 2));
            const int nb = rr / (TH + 2), row = rr - nb * (TH + 2);
            const int n = n0 + nb, h = h0 - 1 + row, w = side ? w0 + TW : w0 - 1;
            hoff[j] = CLHIP_OOB;
            if constexpr (UNPOOL) {
                hioff[j] = CLHIP_OOB;
                hcode[j] = ((h & 1) << 1) | (w & 1);
                if (e < XROWS * 2 && n < N && h >= 0 && h < H && w >= 0 && w < W) {
                    hioff[j] = (int)(((size_t)nb * Cin + cl) * plane_in) + (h >> 1) * Wp + (w >> 1);
                    hoff[j] = hioff[j] * 4;
                }
            } else if (e < XROWS * 2 && n < N && h >= 0 && h < H && w >= 0 && w < W) {
                hoff[j] = ((int)(((size_t)nb * Cin + cl) * plane_hw) + h * W + w) * 4;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int e = tid + 256 * j;
            woff[j] = CLHIP_OOB; wch[j] = 0;
            if (e < W_ELEMS) {
                if (MODE == 0) {
                    const int kl = e / (CK * 9), kidx = e - kl * (CK * 9);
                    wch[j] = kidx / 9;
                    if (ko0 + kl < Kw) woff[j] = ((ko0 + kl) * Cw * 9 + kidx) * 4;
                } else {
                    const int kl = e / (KT * 9), rem = e - kl * (KT * 9);
                    const int cl = rem / 9;
                    wch[j] = kl;
                    if (ko0 + cl < Cw) woff[j] = ((kl * Cw + ko0) * 9 + rem) * 4;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int e = tid + 256 * j;
            xoff[j] = CLHIP_OOB; xch[j] = 0;
            if (e < XS_FLOATS) {
                const int cl = e / G::PLANE, rem = e - cl * G::PLANE;
                const int col = rem % G::TWP, rr = rem / G::TWP;
                const int row = rr % (TH + 2), nb = rr / (TH + 2);
                const int n = n0 + nb, h = h0 - 1 + row, w = w0 - 1 + col;
                xch[j] = cl;
                if (n < N && h >= 0 && h < H && w >= 0 && w < W)
                    xoff[j] = ((int)(((size_t)nb * Cin + cl) * plane_hw) + h * W + w) * 4;
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rs_w = clhip_rsrc(wt, (size_t)Kw * Cw * 9 * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_blk, (size_t)(N - n0) * Cin * plane_in * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? pool_idx + (size_t)n0 * Cin * plane_in : pool_idx,
                                                   UNPOOL ? (size_t)(N - n0) * Cin * plane_in : 0);

    // unit u of chunk `chunk` -> staging registers.  Chunks past the end load nothing (every voffset OOB).
    auto load_unit = [&](int u, int chunk) {
#ifdef CLHIP_ABL_NOLOAD
        if (chunk >= 2) return;
#endif
        const int c0 = chunk * CK;
        const bool live = chunk < n_chunks;                                   // wave-uniform
        const int wb = (MODE == 0 ? c0 * 9 : c0 * Cw * 9) * (int)sizeof(float);
        const int xb = c0 * (int)plane_in * (int)sizeof(float);
        if constexpr (VEC) {
            if (u < W_IT) wv[u] = clhip_buf_load4(rs_w, live ? woff[u] : CLHIP_OOB, live ? wb : 0);
            else if (u < W_IT + X_IT) {
                const int j = u - W_IT;
                if constexpr (UNPOOL) {
                    xp[j] = clhip_buf_load2(rs_x, live ? xoff[j] : CLHIP_OOB, live ? xb : 0);
                    xi[j] = clhip_buf_load_u16(rs_i, live ? xioff[j] : CLHIP_OOB, live ? xb / 4 : 0);
                } else {
                    xv[j] = clhip_buf_load4(rs_x, live ? xoff[j] : CLHIP_OOB, live ? xb : 0);
                }
            } else {
                const int j = u - W_IT - X_IT;
                hv[j] = clhip_buf_load(rs_x, live ? hoff[j] : CLHIP_OOB, live ? xb : 0);
                if constexpr (UNPOOL) hi[j] = clhip_buf_load_u8(rs_i, live ? hioff[j] : CLHIP_OOB, live ? xb / 4 : 0);
            }
        } else {
            const int cleft = live ? Cin - c0 : 0;               // channels left (>= CK except in the tail chunk)
            if (u < W_IT) wreg[u] = clhip_buf_load(rs_w, wch[u] < cleft ? woff[u] : CLHIP_OOB, live ? wb : 0);
            else xreg[u - W_IT] = clhip_buf_load(rs_x, xch[u - W_IT] < cleft ? xoff[u - W_IT] : CLHIP_OOB, live ? xb : 0);
        }
    };
#pragma unroll
    for (int u = 0; u < NUNITS; ++u) load_unit(u, 0);
#ifdef CLHIP_TRACE
    const unsigned long long tr_idx = TR_NOW();
#endif

    // ---- part B: LDS destinations and fragment addresses (chunk 0 is on its way)
    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int e = tid + 256 * j;
            if (MODE == 0) {
                // row kl: CK*9 contiguous floats = CK*9/4 float4; LDS dst of float t: (4f+t)*LDW + kl
                const int kl = e / (CK * 9 / 4), f = e - kl * (CK * 9 / 4);
                wdst[j] = (4 * f) * LDW + kl;
            } else {
                // in-channel row kl: KT*9 contiguous floats (c = ko0.., rs); dst of float t individually
                const int kl = e / (KT * 9 / 4), f = e - kl * (KT * 9 / 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int rem = 4 * f + t, cl = rem / 9, rs = rem - cl * 9;
                    wdst[j * 4 + t] = (kl * 9 + (8 - rs)) * LDW + cl;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e / (TW / 4), f = e - rowid * (TW / 4);
            const int cl = rowid / (NB * (TH + 2)), rr = rowid - cl * (NB * (TH + 2));
            const int nb = rr / (TH + 2), row = rr - nb * (TH + 2);
            xdst[j] = cl * G::PLANE + (nb * (TH + 2) + row) * G::TWP + 1 + 4 * f;
        }
#pragma unroll
        for (int j = 0; j < H_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e >> 1, side = e & 1;
            const int cl = rowid / (NB * (TH + 2)), rr = rowid - cl * (NB * (TH + 2));
            const int nb = rr / (TH + 2), row = rr - nb * (TH + 2);
            hdst[j] = cl * G::PLANE + (nb * (TH + 2) + row) * G::TWP + (side ? TW + 1 : 0);
        }
    } else {
#pragma unroll
        for (int j = 0; j < W_IT; ++j) {
            const int e = tid + 256 * j;
            wdst[j] = 0;
            if (e < W_ELEMS) {
                if (MODE == 0) {
                    const int kl = e / (CK * 9), kidx = e - kl * (CK * 9);
                    wdst[j] = kidx * LDW + kl;
                } else {
                    const int kl = e / (KT * 9), rem = e - kl * (KT * 9);
                    const int cl = rem / 9, rs = rem - cl * 9;
                    wdst[j] = (kl * 9 + (8 - rs)) * LDW + cl;
                }
            }
        }
    }

    // staging registers of unit u -> LDS buffer at float offset bo
    auto store_unit = [&](int u, int bo) {
#ifdef CLHIP_ABL_NOSTORE
        if (bo >= 0 && n_chunks > 0) return;
#endif
        float* ws = lds + bo;
        float* xs = ws + WS_FLOATS;
        if constexpr (VEC) {
            if (u < W_IT) {
                const int j = u;
                if (256 * (j + 1) <= W_ELEMS / 4 || tid + 256 * j < W_ELEMS / 4) {
                    if (MODE == 0) {
                        float* d = ws + wdst[j];
                        d[0] = wv[j].x; d[LDW] = wv[j].y; d[2 * LDW] = wv[j].z; d[3 * LDW] = wv[j].w;
                    } else {
                        ws[wdst[4 * j]] = wv[j].x; ws[wdst[4 * j + 1]] = wv[j].y;
                        ws[wdst[4 * j + 2]] = wv[j].z; ws[wdst[4 * j + 3]] = wv[j].w;
                    }
                }
            } else if (u < W_IT + X_IT) {
                const int j = u - W_IT;
                if (256 * (j + 1) <= XV_ELEMS || tid + 256 * j < XV_ELEMS) {
                    float* d = xs + xdst[j];
                    if constexpr (UNPOOL) {
                        // four columns of one row = two pooling windows: the element whose position code the forward
                        // pass recorded gets the window's gradient, the others 0 (max_pool2d backward)
                        const int c = xcode[j], i0 = (int)(xi[j] & 0xffu), i1 = (int)((xi[j] >> 8) & 0xffu);
                        d[0] = i0 == c ? xp[j].x : 0.f; d[1] = i0 == c + 1 ? xp[j].x : 0.f;
                        d[2] = i1 == c ? xp[j].y : 0.f; d[3] = i1 == c + 1 ? xp[j].y : 0.f;
                    } else {
                        d[0] = xv[j].x; d[1] = xv[j].y; d[2] = xv[j].z; d[3] = xv[j].w;
                    }
                }
            } else {
                const int j = u - W_IT - X_IT;
                if (256 * (j + 1) <= XROWS * 2 || tid + 256 * j < XROWS * 2)
                    xs[hdst[j]] = (!UNPOOL || (int)hi[j] == hcode[j]) ? hv[j] : 0.f;
            }
        } else {
            if (u < W_IT) {
                if (256 * (u + 1) <= W_ELEMS || tid + 256 * u < W_ELEMS) ws[wdst[u]] = wreg[u];
            } else {
                const int j = u - W_IT;
                if (256 * (j + 1) <= XS_FLOATS || tid + 256 * j < XS_FLOATS) xs[tid + 256 * j] = xreg[j];
            }
        }
    };

    // per-lane LDS offsets of this wave's pixel subtiles (pixel q -> halo-plane address)
    int pixoff[G::NT];
#pragma unroll
    for (int t = 0; t < G::NT; ++t) {
        int nb, th, tw;
        tile_pixel<TW, TH>(wp * G::NT + t, li, pool, nb, th, tw);
        pixoff[t] = (nb * (TH + 2) + th) * G::TWP + tw;
    }
    const int a_lane = kk * 9 * LDW + wk * 32 + li;   // weight-tile read offset
    const int b_lane = kk * G::PLANE;                 // activation-plane read offset

    // One accumulator per 32-pixel subtile, k-ordered: the channel reduction of an output element is ONE fmaf chain in
    // (channel, tap) order, the order the reference's CPU kernels use per output element — a two-accumulator split of
    // the 64-pixel tiles measured no faster and let chaotic end-to-end trajectories drift from the reference's (G10).
    constexpr int NACC = G::NT;
    floatx16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ------------------------------------------------------------------ pipeline
    // Two LDS buffers, ONE barrier per chunk, and no phase in which the matrix pipe waits for staging: an MFMA keeps
    // the pipe busy for 64 cycles, in whose shadow this wave issues one piece of other work per MFMA "slot":
    //   channel pair 0          : LDS writes of chunk c+1 (loaded during chunk c-1) into the other buffer
    //   channel pair 1          : global loads of chunk c+2 into the staging registers just freed
    //   every pair              : the LDS operand reads of the NEXT pair (register double buffering)
    //   last pair, after barrier: the operand reads of chunk c+1's first pair from the other buffer
    // The barrier sits in front of the last pair: by then every wave has issued (and waited for) all its reads of the
    // current buffer — the last pair's operands are already in registers — and all writes of chunk c+1.
    // (Measured on the phase-separated form of this loop, one wave per SIMD: 80 % of the MFMA phase busy and
    // load-issue + store + barrier phases worth another 45 % of it with the pipe idle.)
    constexpr int CP = CK / 2;                         // channel pairs per chunk
    constexpr int SL = 9 * G::NT;                      // MFMA slots per pair
    constexpr int UPS = (NUNITS + SL - 1) / SL;        // staging units per slot
    constexpr int CP_LD = CP > 2 ? 1 : CP - 1;
    float af[2][9], bf[2][G::NT][9];

#pragma unroll
    for (int u = 0; u < NUNITS; ++u) store_unit(u, 0);
#pragma unroll
    for (int u = 0; u < NUNITS; ++u) load_unit(u, 1);
    __syncthreads();
#pragma unroll
    for (int rs = 0; rs < 9; ++rs) {
        const int r = rs / 3, s3 = rs - 3 * (rs / 3);
        af[0][rs] = lds[a_lane + rs * LDW];
#pragma unroll
        for (int t = 0; t < G::NT; ++t) bf[0][t][rs] = lds[WS_FLOATS + b_lane + pixoff[t] + r * G::TWP + s3];
    }
#ifdef CLHIP_TRACE
    const unsigned long long tr_pro = TR_NOW();
#endif

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int bo = (chunk & 1) * BUF_FLOATS, bn = BUF_FLOATS - bo;
        const float* wsc = lds + bo + a_lane;
        const float* xsc = lds + bo + WS_FLOATS + b_lane;
        const float* wsn = lds + bn + a_lane;
        const float* xsn = lds + bn + WS_FLOATS + b_lane;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
#pragma unroll
            for (int rs = 0; rs < 9; ++rs) {
                const int r = rs / 3, s3 = rs - 3 * (rs / 3);
#pragma unroll
                for (int t = 0; t < G::NT; ++t) {
                    const int slot = rs * G::NT + t;
                    const int ai = t;
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cp & 1][rs], bf[cp & 1][t][rs], acc[ai], 0, 0, 0);
#ifndef CLHIP_ABL_NOBAR
                    if (cp == CP - 1 && slot == 0) __syncthreads();
#endif
#ifndef CLHIP_ABL_NOFRAG
                    if (cp + 1 < CP) {
                        if (t == 0) af[(cp + 1) & 1][rs] = wsc[((2 * (cp + 1)) * 9 + rs) * LDW];
                        bf[(cp + 1) & 1][t][rs] = xsc[(2 * (cp + 1)) * G::PLANE + pixoff[t] + r * G::TWP + s3];
                    } else {
                        if (t == 0) af[0][rs] = wsn[rs * LDW];
                        bf[0][t][rs] = xsn[pixoff[t] + r * G::TWP + s3];
                    }
#endif
                    if (cp == 0) {
#pragma unroll
                        for (int u = slot * UPS; u < (slot + 1) * UPS && u < NUNITS; ++u) {
#ifdef CLHIP_ABL_NOWSTORE          // tuning aid: weight tiles stay those of chunks 0 / 1 (valid data, wrong results)
                            if (u < W_IT && chunk >= 1) continue;
#endif
#ifdef CLHIP_ABL_NOXSTORE
                            if (u >= W_IT && chunk >= 1) continue;
#endif
                            store_unit(u, bn);
                        }
                    }
                    if (cp == CP_LD) {
#pragma unroll
                        for (int u = slot * UPS; u < (slot + 1) * UPS && u < NUNITS; ++u) load_unit(u, chunk + 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#ifdef CLHIP_TRACE
    const unsigned long long tr_loop = TR_NOW();
    auto tr_finish = [&]() {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tr_end = TR_NOW();
        if (g_trace && lane == 0) {
            unsigned long long* t = g_trace + ((size_t)blockIdx.x * 4 + wave) * 16;
            t[0] = tr_start; t[1] = tr_idx; t[2] = tr_pro; t[3] = tr_loop; t[4] = tr_end;
            t[5] = 0; t[6] = 0; t[7] = 0; t[8] = 0;
            t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            t[10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
            t[11] = n_chunks;
        }
    };
#endif

    // ---- epilogue: reg r of lane l = D[row = out-channel][col = pixel li]
    // Buffer stores: the lane's pixel/channel-base offset is one VGPR per subtile (CLHIP_OOB drops the store), the
    // channel of register r is a scalar offset.  Bias comes from LDS (a global load per register, each waited for,
    // used to cost ~10 % of the kernel).
    const int kb = ko0 + wk * 32 + 4 * kk;               // channel of register r: kb + (r & 3) + 8 * (r >> 2)
    const bool kfull = ko0 + KT <= Cout;                 // uniform: no per-register channel test
    auto rch = [](int r) { return (r & 3) + 8 * (r >> 2); };
    if (MODE == 0 && pool) {
        // fused ReLU + 2x2/2 max-pool (VGGSlim.py:32,38): the window's candidates are lanes li, li^1 (right),
        // li^VX (below), li^1^VX; the top-left lane writes the maximum and the 2-bit argmax (first maximum in
        // ATen's scan order wins).  The pre-pool activation never goes to HBM.
        constexpr int VX = TW == 8 ? 8 : 16;
        const int OH = H >> 1, OW = W >> 1;
        const int chw = OH * OW;
        const __amdgpu_buffer_rsrc_t rs_o = clhip_rsrc(out + (size_t)n0 * Cout * chw, (size_t)(N - n0) * Cout * chw * sizeof(float));
        const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(pool_idx + (size_t)n0 * Cout * chw, (size_t)(N - n0) * Cout * chw);
        const bool writer = !(li & 1) && !(li & VX);
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {
            int nb, th, tw;
            tile_pixel<TW, TH>(wp * G::NT + t, li, true, nb, th, tw);
            const int h = h0 + th, w = w0 + tw;
            const bool ok = writer && (n0 + nb < N) && (h < H) && (w < W);
            const int eoff = (nb * Cout + kb) * chw + (h >> 1) * OW + (w >> 1);      // element offset of channel kb
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaxf(acc[t][r] + bias_s[wk * 32 + 4 * kk + rch(r)], 0.f);
                const float tr = __shfl_xor(v, 1, 64);
                const float bl = __shfl_xor(v, VX, 64);
                const float br = __shfl_xor(tr, VX, 64);
                float m = v; int a = 0;
                if (tr > m) { m = tr; a = 1; }
                if (bl > m) { m = bl; a = 2; }
                if (br > m) { m = br; a = 3; }
                if (!(m > 0.f)) a = CLHIP_POOL_DEAD;      // ReLU folded into the code: no window position takes a gradient
                const bool okr = ok && (kfull || kb + rch(r) < Cout);
                const int eo = eoff + rch(r) * chw;
                clhip_buf_store(m, rs_o, okr ? eo * 4 : CLHIP_OOB, 0);
                clhip_buf_store_u8((uint8_t)a, rs_i, okr ? eo : CLHIP_OOB, 0);
            }
        }
#ifdef CLHIP_TRACE
        tr_finish();
#endif
        return;
    }
    {
        const int chw = (int)plane_hw;
        const __amdgpu_buffer_rsrc_t rs_o = clhip_rsrc(out + (size_t)n0 * out_img_of(Cout, chw), (size_t)(N - n0) * Cout * chw * sizeof(float));
        const __amdgpu_buffer_rsrc_t rs_m = clhip_rsrc(MODE == 1 && mask_src ? mask_src + (size_t)n0 * out_img_of(Cout, chw) : out,
                                                       (size_t)(N - n0) * Cout * chw * sizeof(float));
#pragma unroll
        for (int t = 0; t < G::NT; ++t) {
            int nb, th, tw;
            tile_pixel<TW, TH>(wp * G::NT + t, li, false, nb, th, tw);
            const int h = h0 + th, w = w0 + tw;
            const bool ok = (n0 + nb < N) && (h < H) && (w < W);
            // CLHIP_OOB (0x80000000) + a channel offset < 2^31 is still out of range
            const int boff = ok ? ((nb * Cout + kb) * chw + h * W + w) * 4 : CLHIP_OOB;
            float mk[16];
            if (MODE == 1 && mask_src) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mk[r] = clhip_buf_load(rs_m, (kfull || kb + rch(r) < Cout) ? boff + rch(r) * chw * 4 : CLHIP_OOB, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r];
                if (MODE == 0) {
                    if (bias) v += bias_s[wk * 32 + 4 * kk + rch(r)];
                    if (relu) v = fmaxf(v, 0.f);
                } else {
                    if (mask_src) v = mk[r] > 0.f ? v : 0.f;
                }
                clhip_buf_store(v, rs_o, (kfull || kb + rch(r) < Cout) ? boff + rch(r) * chw * 4 : CLHIP_OOB, 0);
            }
        }
    }
#ifdef CLHIP_TRACE
    tr_finish();
#endif
}

template <int TW, int TH, int NB, int CK, int MODE, bool VEC, bool UNPOOL = false>
__global__ __launch_bounds__(256, CLHIP_CONV_MIN_WAVES) void conv3x3_mfma_kernel(
    const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out,
    int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu,
    int tiles_w, int tiles_h, int n_pix_tiles, uint8_t* __restrict__ pool_idx) {
    __shared__ float lds[ConvLds<TW, TH, NB, CK>::FLOATS];
    __shared__ float bias_s[KT];
    conv3x3_mfma_body<TW, TH, NB, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tiles_w, tiles_h,
                                                         n_pix_tiles, pool_idx, (int)blockIdx.x, 0, lds, bias_s);
}

// Two tile geometries in ONE launch: blocks [0, blocks_a) cover images [0, n_a) with TWxTHa tiles, the rest cover images
// [n_a, N) with TWxTHb tiles (half the pixels).  1600 equal blocks on 256 CUs leave a quarter of the chip a seventh block
// while the rest idles (layer 2 of small_VGG9 at N = 200 ran at 94 TFLOP/s, at N = 192 — whole rounds — at 104): the odd
// images go out as half-size tiles that fill the last round evenly.  Same per-element arithmetic in either geometry.
template <int TW, int THa, int THb, int CK, int MODE, bool VEC, bool UNPOOL = false>
__global__ __launch_bounds__(256, CLHIP_CONV_MIN_WAVES) void conv3x3_mfma_mixed_kernel(
    const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out,
    int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu,
    int tiles_w, int tiles_ha, int tiles_hb, int n_a, int blocks_a, uint8_t* __restrict__ pool_idx) {
    __shared__ float lds[ConvLds<TW, THa, 1, CK>::FLOATS];
    __shared__ float bias_s[KT];
    static_assert(ConvLds<TW, THb, 1, CK>::FLOATS <= ConvLds<TW, THa, 1, CK>::FLOATS, "the larger geometry sizes the LDS");
    if ((int)blockIdx.x < blocks_a)
        conv3x3_mfma_body<TW, THa, 1, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, n_a, Cin, Cout, H, W, Kw, Cw, relu, tiles_w,
                                                              tiles_ha, tiles_w * tiles_ha * n_a, pool_idx, (int)blockIdx.x, 0, lds, bias_s);
    else
        conv3x3_mfma_body<TW, THb, 1, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tiles_w,
                                                              tiles_hb, tiles_w * tiles_hb * (N - n_a), pool_idx,
                                                              (int)blockIdx.x - blocks_a, n_a, lds, bias_s);
}

// The same for the deeper layers: 16-wide planes (16x8 + 16x4 tiles) and 8x8 planes (two images per tile + one image per
// tile).  base_VGG9 / wide_VGG9 at N = 200 launch 800 or 1600 of the large tiles there: 3.125 or 6.25 per CU.
template <int TW, int THa, int NBa, int THb, int NBb, int CK, int MODE, bool VEC, bool UNPOOL = false>
__global__ __launch_bounds__(256, CLHIP_CONV_MIN_WAVES) void conv3x3_mfma_mixed2_kernel(
    const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out,
    int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu,
    int tiles_w, int tiles_ha, int tiles_hb, int n_a, int blocks_a, uint8_t* __restrict__ pool_idx) {
    constexpr int FA = ConvLds<TW, THa, NBa, CK>::FLOATS, FB = ConvLds<TW, THb, NBb, CK>::FLOATS;
    __shared__ float lds[FA > FB ? FA : FB];
    __shared__ float bias_s[KT];
    if ((int)blockIdx.x < blocks_a)
        conv3x3_mfma_body<TW, THa, NBa, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, n_a, Cin, Cout, H, W, Kw, Cw, relu, tiles_w,
                                                                tiles_ha, tiles_w * tiles_ha * ((n_a + NBa - 1) / NBa), pool_idx,
                                                                (int)blockIdx.x, 0, lds, bias_s);
    else
        conv3x3_mfma_body<TW, THb, NBb, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tiles_w,
                                                                tiles_hb, tiles_w * tiles_hb * ((N - n_a + NBb - 1) / NBb), pool_idx,
                                                                (int)blockIdx.x - blocks_a, n_a, lds, bias_s);
}

template <int TW, int TH, int NB, int CK, int MODE, bool VEC, bool UNPOOL = false>
int launch_geo(const float* in, const float* wt, const float* bias, const float* mask_src, float* out,
               int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu, hipStream_t s, uint8_t* pool_idx) {
    int tiles_w = (W + TW - 1) / TW, tiles_h = (H + TH - 1) / TH, ngrp = (N + NB - 1) / NB;
    int n_pix_tiles = tiles_w * tiles_h * ngrp;
    int kts = (Cout + KT - 1) / KT;
    long long blocks = (long long)n_pix_tiles * kts;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;
    hipLaunchKernelGGL((conv3x3_mfma_kernel<TW, TH, NB, CK, MODE, VEC, UNPOOL>), dim3((unsigned)blocks), dim3(256), 0, s,
                       in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tiles_w, tiles_h, n_pix_tiles, pool_idx);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// Tile geometry choice: full 128-pixel tiles while they still give more blocks than CUs,
// otherwise 64-pixel tiles (deep layers at 8x8 have few pixels).
template <int CK, int MODE, bool VEC, bool UNPOOL = false>
int launch_conv(const float* in, const float* wt, const float* bias, const float* mask_src, float* out,
                int N, int Cin, int Cout, int H, int W, int Kw, int Cw, int relu, hipStream_t s,
                uint8_t* pool_idx = nullptr) {
    const int kts = (Cout + KT - 1) / KT;
    const long long pix = (long long)N * H * W;
    // measured on small_VGG9 at N=200: 400 blocks of 128 pixels beat 800 of 64 (47 vs 51 us at 16x16: one A fragment
    // feeds two MFMAs), 200 of 128 lose to 400 of 64 at 8x8 (too few blocks for 256 CUs)
#ifndef CLHIP_BIG_MIN
#define CLHIP_BIG_MIN 300
#endif
#ifndef CLHIP_MIXED_TILES
#define CLHIP_MIXED_TILES 1
#endif
    const bool big = (pix / 128) * kts >= CLHIP_BIG_MIN;
#define GEO(TW_, TH_, NB_) launch_geo<TW_, TH_, NB_, CK, MODE, VEC, UNPOOL>(in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, s, pool_idx)
    if constexpr (!VEC && !UNPOOL) {
        if (H == 13 && W == 13 && !pool_idx) return GEO(13, 13, 1);      // whole-plane dense tile
    }
    if constexpr (VEC) {
        if (W > 16 && big && CLHIP_MIXED_TILES) {
            // whole launch rounds of 128-pixel tiles, the remaining images as 64-pixel tiles (conv3x3_mfma_mixed_kernel)
            const int tw_n = (W + 31) / 32, tha = (H + 3) / 4, thb = (H + 1) / 2;
            const long long per_img = (long long)tw_n * tha * kts, total = per_img * N, rem = total % 256;
            if (total > 256 && rem != 0 && rem <= 192) {
                const int imgs_b = (int)((rem + per_img - 1) / per_img), n_a = N - imgs_b;
                const long long blocks_a = per_img * n_a, blocks_b = (long long)tw_n * thb * kts * imgs_b;
                if (n_a > 0 && blocks_a + blocks_b <= 0x7fffffffLL) {
                    hipLaunchKernelGGL((conv3x3_mfma_mixed_kernel<32, 4, 2, CK, MODE, VEC, UNPOOL>), dim3((unsigned)(blocks_a + blocks_b)),
                                       dim3(256), 0, s, in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tw_n, tha, thb, n_a,
                                       (int)blocks_a, pool_idx);
                    CLHIP_LAUNCH_CHECK();
                    return 0;
                }
            }
        }
    }
    if constexpr (VEC) {
        if (W <= 16 && big && H > 4 && CLHIP_MIXED_TILES) {
            // (a launch of fewer than two full rounds gains nothing: the remainder is then most of the work)
            const bool w16 = W > 8;
            const int TWx = w16 ? 16 : 8, tw_n = (W + TWx - 1) / TWx, nba = w16 ? 1 : 2;
            const int tha = w16 ? (H + 7) / 8 : (H + 7) / 8, thb = w16 ? (H + 3) / 4 : (H + 7) / 8;
            const long long per_unit = (long long)tw_n * tha * kts;               // blocks per nba images
            const long long units = (N + nba - 1) / nba, total = per_unit * units, rem = total % 256;
            if (total >= 512 && rem != 0 && rem <= 192 && N % nba == 0) {
                const long long units_b = (rem + per_unit - 1) / per_unit;
                const int n_a = (int)((units - units_b) * nba), imgs_b = N - n_a;
                const long long blocks_a = per_unit * (units - units_b), blocks_b = (long long)tw_n * thb * kts * imgs_b;
                if (n_a > 0 && blocks_a + blocks_b <= 0x7fffffffLL) {
                    if (w16)
                        hipLaunchKernelGGL((conv3x3_mfma_mixed2_kernel<16, 8, 1, 4, 1, CK, MODE, VEC, UNPOOL>), dim3((unsigned)(blocks_a + blocks_b)),
                                           dim3(256), 0, s, in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tw_n, tha, thb, n_a,
                                           (int)blocks_a, pool_idx);
                    else
                        hipLaunchKernelGGL((conv3x3_mfma_mixed2_kernel<8, 8, 2, 8, 1, CK, MODE, VEC, UNPOOL>), dim3((unsigned)(blocks_a + blocks_b)),
                                           dim3(256), 0, s, in, wt, bias, mask_src, out, N, Cin, Cout, H, W, Kw, Cw, relu, tw_n, tha, thb, n_a,
                                           (int)blocks_a, pool_idx);
                    CLHIP_LAUNCH_CHECK();
                    return 0;
                }
            }
        }
    }
    if (W > 16) return big ? GEO(32, 4, 1) : GEO(32, 2, 1);
    if (W > 8) return big ? GEO(16, 8, 1) : GEO(16, 4, 1);
    return (big && H > 4) ? GEO(8, 8, 2) : GEO(8, 8, 1);
#undef GEO
}

// ---------------------------------------------------------------------------------------------------------
// First layer (C = 3) with fused bias + ReLU + 2x2 max-pool (VGGSlim.py:27-40: conv(3->64), ReLU, 'M').
//
// K = 27 (padded to 28 = 14 MFMA k-pairs) is far too short for the chunked kernel above: one chunk, no
// pipelining, a block lives ~10 us and most of that is prologue / epilogue latency (measured 24 TFLOP/s).
// Here the block is persistent: the 64x27 weight slice lives in 28 VGPRs per lane for the whole kernel (A
// operand straight from registers), tiles of 8 rows x 32 columns stream through two 4 KB LDS halo buffers
// (register-staged prefetch of tile i+1 during tile i), and each wave owns two image rows h, h+1 so the
// vertical pool partner is the SAME lane of the other accumulator and the horizontal one is lane^1 (one DPP
// quad_perm) — no cross-row shuffles, the pre-pool activation never leaves registers.
//   per wave and tile: 28 ds_read_b32, 56 MFMA (2 rows x 2 channel halves x 14 k-pairs).
constexpr int C3_TW = 32, C3_TH = 8, C3_TWP = C3_TW + 2;
constexpr int C3_WROWS = 4;                         // halo rows a wave stages for ITS two image rows
constexpr int C3_PLANE = C3_WROWS * C3_TWP;         // 136 floats per channel
constexpr int C3_HALO = 3 * C3_PLANE;               // 408 floats per wave and buffer
constexpr int C3_PLD = 20;     // pooled-slab row stride: 16-byte aligned rows, kk halves (4 rows apart) on disjoint banks

__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

// Every wave is an independent worker: it stages the 4-row halo of its own two image rows into a private,
// double-buffered LDS slab, so the tile loop has NO block barrier and the waves of a SIMD drift apart —
// one wave's MFMA phase overlaps another's epilogue (VALU + LDS) instead of all four marching in lock step.
__global__ __launch_bounds__(256, 2) void conv3x3_c3_relu_pool_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    float* __restrict__ out, uint8_t* __restrict__ pool_idx, int N, int Cout, int H, int W,
    int tiles_w, int tiles_h, int ntiles) {
    __shared__ float halo_s[4 * 2 * C3_HALO];
    __shared__ float bias_s[KT];
    __shared__ __attribute__((aligned(16))) float pool_s[4 * KT * C3_PLD];     // per wave: [64 channels][16 pooled px] (+4 pad)
    __shared__ __attribute__((aligned(16))) uint8_t idx_s[4 * KT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kk = lane >> 5;
    const int ko0 = blockIdx.y * KT;

    if (tid < KT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;

    // K order: the 27 taps (+1 zero pad) are paired so that the two taps of a pair differ by a FIXED LDS offset
    // (kk selects the tap): 9 pairs (c, r=0, s)/(c, r=1, s) [+1 halo row], 3 pairs (c, 2, 0)/(c, 2, 1) [+1 column],
    // (0, 2, 2)/(1, 2, 2) [+1 channel plane] and (2, 2, 2)/pad.  Three per-lane base registers + immediates address
    // every B operand; the A fragments (weights, in registers for the whole kernel) follow the same order.
    float a[2][14];
    auto tap = [&](int j, int& c, int& r, int& s) -> bool {        // tap of k-pair j for this lane's kk; false = zero pad
        if (j < 9) { c = j / 3; s = j - 3 * c; r = kk; return true; }
        if (j < 12) { c = j - 9; r = 2; s = kk; return true; }
        if (j == 12) { c = kk; r = 2; s = 2; return true; }
        c = 2; r = 2; s = 2; return kk == 0;
    };
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        int c, r, s2;
        const bool real = tap(j, c, r, s2);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int ch = ko0 + 32 * half + li;
            a[half][j] = (real && ch < Cout) ? wt[(size_t)ch * 27 + c * 9 + r * 3 + s2] : 0.f;
        }
    }
    const int b_row = li + kk * C3_TWP, b_col = li + kk, b_pln = li + kk * C3_PLANE;
    auto b_addr = [&](int j) -> int {               // LDS offset of tap (j, kk) for image row 0 of the wave
        if (j < 9) return b_row + (j / 3) * C3_PLANE + (j % 3);
        if (j < 12) return b_col + (j - 9) * C3_PLANE + 2 * C3_TWP;
        if (j == 12) return b_pln + 2 * C3_TWP + 2;
        return li + 2 * C3_PLANE + 2 * C3_TWP + 2;                 // kk = 1 reads the same finite value; its A is 0
    };

    // staging of this wave's [3][4][34] halo: six loads cover columns 1..32 of two rows each (lanes 0-31 row 2i,
    // lanes 32-63 row 2i+1 of the 12 (channel, row) lines), a seventh the two border columns of all 12 lines.
    float sv[7];
    const size_t plane_hw = (size_t)H * W;
    const int st_src = kk * W + li;                                 // + (c*plane_hw + (rr-1)*W) per load
    const int st_dst = kk * C3_TWP + li + 1;
    const int e_line = lane >> 1, e_side = lane & 1;                // border columns: 24 lanes
    const int e_c = e_line >> 2, e_rr = e_line & 3;
    const int st_esrc = e_c * (int)plane_hw + (e_rr - 1) * W + (e_side ? C3_TW : -1);
    const int st_edst = e_c * C3_PLANE + e_rr * C3_TWP + (e_side ? C3_TW + 1 : 0);
    auto tile_coords = [&](int t, int& n, int& h, int& w0) {
        const int tw = t % tiles_w, q = t / tiles_w;
        const int th = q % tiles_h;
        n = q / tiles_h; h = th * C3_TH + 2 * wave; w0 = tw * C3_TW;
    };
    auto load_tile = [&](int t) {
        int n, h, w0;
        tile_coords(t, n, h, w0);
        const float* xb = x + (size_t)n * 3 * plane_hw + (size_t)h * W + w0;
        bool rok[2];                                 // halo rows h-1+kk and h+1+kk
        rok[0] = (unsigned)(h - 1 + kk) < (unsigned)H;
        rok[1] = (unsigned)(h + 1 + kk) < (unsigned)H;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float* p = xb + (i >> 1) * (ptrdiff_t)plane_hw + (2 * (i & 1) - 1) * W + st_src;
            sv[i] = *(rok[i & 1] ? p : clhip_zero16);
        }
        const bool eok = lane < 24 && (unsigned)(h - 1 + e_rr) < (unsigned)H &&
                         (unsigned)(w0 + (e_side ? C3_TW : -1)) < (unsigned)W;
        const float* pe = xb + st_esrc;
        sv[6] = *(eok ? pe : clhip_zero16);
    };
    float* hw_s = halo_s + wave * (2 * C3_HALO);
    auto store_tile = [&](int buf) {
        float* d = hw_s + buf * C3_HALO;
#pragma unroll
        for (int i = 0; i < 6; ++i) d[(i >> 1) * C3_PLANE + 2 * (i & 1) * C3_TWP + st_dst] = sv[i];
        if (lane < 24) d[st_edst] = sv[6];
    };

    const int OH = H >> 1, OW = W >> 1;          // OW % 16 == 0 (W % 32 == 0): every 16-pixel pooled run is 64-byte aligned
    int tile = blockIdx.x, buf = 0;
    if (tile < ntiles) { load_tile(tile); store_tile(0); }
    // the weight fragments must be complete BEFORE the loop: otherwise hipcc's waitcnt pass, merging the loop
    // back-edge with this preheader, puts s_waitcnt vmcnt(0) in front of the first MFMA of every tile and the
    // prefetch of tile i+1 stops overlapping the MFMAs of tile i
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();                                // bias_s
    float* pw = pool_s + wave * (KT * C3_PLD);
    uint8_t* iw = idx_s + wave * (KT * 16);
    // one lane-dependent base per array, everything else is an immediate offset of the LDS instruction
    float* pw_l = pw + 4 * kk * C3_PLD + (li >> 1);
    uint8_t* iw_l = iw + 4 * kk * 16 + (li >> 1);
    const float* bias_k = bias_s + 4 * kk;
    for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const int next = tile + gridDim.x;
        if (next < ntiles) load_tile(next);

        floatx16 acc[2][2];
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[half][row][r] = 0.f;
        const float* xs = hw_s + buf * C3_HALO;
        float b0[14], b1[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) { b0[j] = xs[b_addr(j)]; b1[j] = xs[b_addr(j) + C3_TWP]; }
        __builtin_amdgcn_sched_barrier(0);          // all 28 LDS reads in flight before the first MFMA
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][j], b0[j], acc[0][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][j], b0[j], acc[1][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][j], b1[j], acc[0][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1][j], b1[j], acc[1][1], 0, 0, 0);
        }

        // stage tile i+1 now: its loads had the whole MFMA phase to land, and doing it BEFORE the global stores keeps
        // the vmcnt wait from also draining those. The other buffer was last read one iteration ago by this same wave.
        if (next < ntiles) store_tile(buf ^ 1);

        // epilogue: window = {(h, w), (h, w+1), (h+1, w), (h+1, w+1)} = {acc row 0 lane, lane^1, acc row 1 lane, lane^1};
        // first maximum in that scan order wins (ATen max_pool2d).  The even lanes put value and 2-bit argmax into this
        // wave's LDS slab [64 channels][16 pooled pixels]; the slab then leaves as 16-byte stores (4 + 1 store
        // instructions per wave and tile instead of 64 quarter-filled ones — the VMEM issue rate, not the matrix
        // pipe, was the limiter with direct stores: measured 61 us for the layer).
        int n, h, w0;
        tile_coords(tile, n, h, w0);
        asm volatile("" ::: "memory");              // keep the 32 bias reads in the loop (hoisted they cost 32 VGPRs -> scratch spills)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bias_k[32 * half + (r & 3) + 8 * (r >> 2)];
                const float tl = fmaxf(acc[half][0][r] + bv, 0.f), bl = fmaxf(acc[half][1][r] + bv, 0.f);
                const float tr = dpp_xor1(tl), br = dpp_xor1(bl);
                float m = tl; int am = 0;
                if (tr > m) { m = tr; am = 1; }
                if (bl > m) { m = bl; am = 2; }
                if (br > m) { m = br; am = 3; }
                if (!(m > 0.f)) am = CLHIP_POOL_DEAD;
                acc[half][0][r] = m;                           // results stay in the accumulator registers ...
                acc[half][1][r] = __int_as_float(am);
            }
        if (!(li & 1)) {                                        // ... and ONE exec-masked region writes them out
#pragma unroll
            for (int half = 0; half < 2; ++half)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c0 = 32 * half + (r & 3) + 8 * (r >> 2);      // + 4*kk: in the lane bases
                    pw_l[c0 * C3_PLD] = acc[half][0][r];
                    iw_l[c0 * 16] = (uint8_t)__float_as_int(acc[half][1][r]);
                }
        }
        __builtin_amdgcn_wave_barrier();            // LDS ops of one wave execute in order; keep the compiler's order too
        if (h < H) {
            const size_t chw = (size_t)OH * OW;
            const size_t obase = (size_t)n * Cout * chw + (size_t)(h >> 1) * OW + (w0 >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = lane + 64 * i, cl = q >> 2, part = q & 3;
                const float4 v = *reinterpret_cast<const float4*>(pw + cl * C3_PLD + 4 * part);
                if (ko0 + cl < Cout) *reinterpret_cast<float4*>(out + obase + (size_t)(ko0 + cl) * chw + 4 * part) = v;
            }
            const uint4 iv = *reinterpret_cast<const uint4*>(iw + lane * 16);
            if (ko0 + lane < Cout) *reinterpret_cast<uint4*>(pool_idx + obase + (size_t)(ko0 + lane) * chw) = iv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// First layer for rows of whole 64-pixel runs (Tiny-ImageNet: W = 64): the pool window in ONE lane.
//
// In the kernel above a lane owns pixel column li, so the horizontal pool partner lives in lane^1: two DPP moves (+ their
// hazard nops) per accumulator register, both lanes of a pair compute the same window, and the results take a detour
// through an LDS slab to leave as 16-byte stores.  Measured by ablation (N = 200): that epilogue is ~25 us of VALU / LDS
// issue next to ~21 us of MFMA, and the two do not overlap well on one SIMD.
// Here lane li owns columns 2 li and 2 li + 1 of image rows h and h + 1 as FOUR accumulators (row x column parity), i.e. a
// wave covers 2 rows x 64 columns x 32 output channels per unit: the 2x2 window is acc[0][0], acc[0][1], acc[1][0],
// acc[1][1] of the same register and lane — no cross-lane traffic, no redundant lanes (half the VALU work per pooled
// pixel), and register r of the 32 lanes of a half-wave IS a 128-byte run of the pooled row: it is stored directly.
// Same arithmetic and scan order as above => the same bits (values and arg-max codes).
//   unit = (row pair, 32-channel half): 56 MFMAs.  Units are dealt to the 1024 SIMDs of the chip as contiguous ranges (12 or
//   13 each at N = 200), split over the two resident waves of a SIMD (512-thread blocks, one per CU): the balance of the
//   old kernel, and consecutive units of a wave share their staged halo (two channel halves per row pair).
#ifndef CLHIP_C3W64
#define CLHIP_C3W64 1
#endif
// (Measured and removed, round 6: the B operands of a unit as 12 ds_read2_b64 — the four floats X[c][halo row][2 li .. 2 li + 3] around a
// lane's two pixel columns, even row stride, operands picked by name + 20 v_cndmask — instead of 22 two-dword reads: 43.0 against
// 43.1 us; the 13.5 us the LDS reads cost by ablation are not a matter of the instruction count.  tools/experiments/r06_b9.sh)
constexpr int W6_TWP = 67;                          // odd row stride: the two taps of a k-pair sit on different banks
constexpr int W6_PLANE = 4 * W6_TWP;                // 4 halo rows per channel
constexpr int W6_HALO = 3 * W6_PLANE;               // 804 floats per wave and buffer

// Timing-only ablations (tools/experiments; results wrong by design; the product is built with 0): 1 no stores, 2 no pooling
// arithmetic, 4 no LDS reads of the B operands, 8 no MFMAs
#ifndef C3W64_ABL
#define C3W64_ABL 0
#endif
// 1: the bias rides in the 28th slot of K (second half of the last k-pair; A = bias, B = 1.0 from ones_s) and arrives inside the last
// MFMA of the chain instead of through 64 v_add_f32 per unit in the epilogue (measured ~ -0.8 us of 44).  The sum then differs from
// "(sum over 27 taps) + bias" by the rounding of one addition — fp32-grade either way, but ANY change of the last bit re-draws the
// outcome of the bench's ill-conditioned 10-task sweep, and the draw of the bit pattern of rounds 3 - 6 is the recorded one:
// default 0, the product adds the bias behind the chain as every other conv kernel of the library does.
#ifndef CLHIP_C3W64_BIAS_IN_K
#define CLHIP_C3W64_BIAS_IN_K 0
#endif
#if C3W64_ABL & 8
__device__ __forceinline__ floatx16 c3w64_fake_mfma(float a, float b, floatx16 c) { c[0] += a * b; return c; }
#define C3W64_MFMA(a, b, c, x, y, z) c3w64_fake_mfma(a, b, c)
#else
#define C3W64_MFMA(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z)
#endif
template <bool FULL>       // FULL: every block owns 64 real output channels (no per-store channel predicate)
__global__ __launch_bounds__(512) void conv3x3_c3w64_relu_pool_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    float* __restrict__ out, uint8_t* __restrict__ pool_idx, int N, int Cout, int H, int W,
    int tiles_w, int row_pairs, int ntiles) {
    __shared__ __attribute__((aligned(16))) float halo_s[8 * 2 * W6_HALO];
#if CLHIP_C3W64_BIAS_IN_K
    __shared__ float ones_s[W6_TWP + 5];
#else
    __shared__ float bias_s[KT];
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kk = lane >> 5;
    const int ko0 = blockIdx.y * KT;
#if CLHIP_C3W64_BIAS_IN_K
    if (tid < W6_TWP + 5) ones_s[tid] = 1.f;
#else
    if (tid < KT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;
#endif

    // K order and pairing: as in conv3x3_c3_relu_pool_kernel (the two taps of a pair differ by a fixed LDS offset)
    float a[2][14];
    auto tap = [&](int j, int& c, int& r, int& s) -> bool {
        if (j < 9) { c = j / 3; s = j - 3 * c; r = kk; return true; }
        if (j < 12) { c = j - 9; r = 2; s = kk; return true; }
        if (j == 12) { c = kk; r = 2; s = 2; return true; }
        c = 2; r = 2; s = 2; return kk == 0;
    };
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        int c, r, s2;
        const bool real = tap(j, c, r, s2);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int ch = ko0 + 32 * half + li;
            a[half][j] = (real && ch < Cout) ? wt[(size_t)ch * 27 + c * 9 + r * 3 + s2] : 0.f;
#if CLHIP_C3W64_BIAS_IN_K
            if (j == 13 && kk == 1) a[half][j] = (bias && ch < Cout) ? bias[ch] : 0.f;
#endif
        }
    }
    const int b_row = 2 * li + kk * W6_TWP, b_col = 2 * li + kk, b_pln = 2 * li + kk * W6_PLANE;
    auto b_addr = [&](int j) -> int {               // LDS offset of tap (j, kk) for image row 0, even column of the lane
        if (j < 9) return b_row + (j / 3) * W6_PLANE + (j % 3);
        if (j < 12) return b_col + (j - 9) * W6_PLANE + 2 * W6_TWP;
        if (j == 12) return b_pln + 2 * W6_TWP + 2;
        return 2 * li + 2 * W6_PLANE + 2 * W6_TWP + 2;              // kk = 1 reads the same finite value; its A is 0 (or: the bias slot, reads ones_s)
    };

    // units of this wave
    const int simd = blockIdx.x * 4 + (wave & 3), nsimd = gridDim.x * 4;
    const int units = 2 * ntiles;
    const int per = units / nsimd, rem = units - per * nsimd;
    const int s_start = simd * per + min(simd, rem), s_cnt = per + (simd < rem ? 1 : 0);
    const int first = (s_cnt + 1) >> 1;
    const int u0 = __builtin_amdgcn_readfirstlane((wave >> 2) ? s_start + first : s_start);
    const int u1 = __builtin_amdgcn_readfirstlane((wave >> 2) ? s_start + s_cnt : s_start + first);

    // staging of a [3][4][66] halo (rows h-1 .. h+2, columns w0-1 .. w0+64): lane l loads column w0 + l of the 12
    // (channel, row) lines, lanes 0..23 the two border columns; out-of-image elements read as zero (buffer range check)
    const size_t plane_hw = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t r_x = clhip_rsrc(x, (size_t)N * 3 * plane_hw * 4);
    float sv[13];
    const int e_line = lane >> 1, e_side = lane & 1, e_c = e_line >> 2, e_rr = e_line & 3;
    auto tile_coords = [&](int t, int& n, int& h, int& w0) {
        const int tw = t % tiles_w, q = t / tiles_w;
        const int rp = q % row_pairs;
        n = q / row_pairs; h = 2 * rp; w0 = tw * 64;
    };
    auto load_tile = [&](int t) {
        int n, h, w0;
        tile_coords(t, n, h, w0);
        const int base = (n * 3 * H + h - 1) * W + w0;          // element (n, c = 0, row h - 1, column w0); may be negative
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int c = i >> 2, rr = i & 3;
            const bool ok = (unsigned)(h - 1 + rr) < (unsigned)H;
            sv[i] = clhip_buf_load(r_x, ok ? (base + (c * H + rr) * W + lane) * 4 : CLHIP_OOB, 0);
        }
        const int ecol = w0 + (e_side ? 64 : -1);
        const bool eok = (lane < 24) & ((unsigned)(h - 1 + e_rr) < (unsigned)H) & ((unsigned)ecol < (unsigned)W);      // & : no exec-mask branch
        sv[12] = clhip_buf_load(r_x, eok ? ((n * 3 * H + h - 1) * W + (e_c * H + e_rr) * W + ecol) * 4 : CLHIP_OOB, 0);
    };
    float* hw_s = halo_s + wave * (2 * W6_HALO);
    auto store_tile = [&](int buf) {
        float* d = hw_s + buf * W6_HALO;
#pragma unroll
        for (int i = 0; i < 12; ++i) d[(i >> 2) * W6_PLANE + (i & 3) * W6_TWP + 1 + lane] = sv[i];
        if (lane < 24) d[e_c * W6_PLANE + e_rr * W6_TWP + (e_side ? 65 : 0)] = sv[12];
    };

    const int OH = H >> 1, OW = W >> 1;
    const int chw = OH * OW;
    const __amdgpu_buffer_rsrc_t r_o = clhip_rsrc(out, (size_t)N * Cout * chw * 4);
    const __amdgpu_buffer_rsrc_t r_i = clhip_rsrc(pool_idx, (size_t)N * Cout * chw);
    __syncthreads();                                // ones_s / bias_s
    if (u0 >= u1) return;
    int t_cur = u0 >> 1;
    const int t_last = (u1 - 1) >> 1;
    int buf = 0;
    load_tile(t_cur);
    store_tile(0);
    if (t_cur < t_last) load_tile(t_cur + 1);
    for (int u = u0; u < u1; ++u) {
        const int t = u >> 1, half = u & 1;
        float ac[14];
#pragma unroll
        for (int j = 0; j < 14; ++j) ac[j] = half ? a[1][j] : a[0][j];
        floatx16 acc[2][2];
        const float* xs = hw_s + buf * W6_HALO;
        float b[2][2][14];
#pragma unroll
        for (int j = 0; j < 14; ++j)
#pragma unroll
            for (int row = 0; row < 2; ++row)
#pragma unroll
                for (int par = 0; par < 2; ++par)
#if C3W64_ABL & 4          // timing only: no LDS reads of the B operands
                    b[row][par][j] = __int_as_float(0x3f800000 + j + row + par + lane);
#else
#if CLHIP_C3W64_BIAS_IN_K
                    b[row][par][j] = (j == 13 && kk == 1) ? ones_s[row * W6_TWP + par] : xs[b_addr(j) + row * W6_TWP + par];
#else
                    b[row][par][j] = xs[b_addr(j) + row * W6_TWP + par];
#endif
#endif
        __builtin_amdgcn_sched_barrier(0);          // all LDS reads in flight before the first MFMA
        {   // first k-pair: C operand is the constant 0 (no 64 v_mov to clear the accumulators)
            const floatx16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0][0] = C3W64_MFMA(ac[0], b[0][0][0], zero, 0, 0, 0);
            acc[0][1] = C3W64_MFMA(ac[0], b[0][1][0], zero, 0, 0, 0);
            acc[1][0] = C3W64_MFMA(ac[0], b[1][0][0], zero, 0, 0, 0);
            acc[1][1] = C3W64_MFMA(ac[0], b[1][1][0], zero, 0, 0, 0);
        }
#pragma unroll
        for (int j = 1; j < 14; ++j) {
            acc[0][0] = C3W64_MFMA(ac[j], b[0][0][j], acc[0][0], 0, 0, 0);
            acc[0][1] = C3W64_MFMA(ac[j], b[0][1][j], acc[0][1], 0, 0, 0);
            acc[1][0] = C3W64_MFMA(ac[j], b[1][0][j], acc[1][0], 0, 0, 0);
            acc[1][1] = C3W64_MFMA(ac[j], b[1][1][j], acc[1][1], 0, 0, 0);
        }
        // The next row pair's halo (loaded while this one computed) goes to the other LDS buffer HERE, between the MFMAs and the
        // output stores, and the loads of the row pair after it are issued here too.  Vector-memory operations of a wave complete in
        // order and loads and stores share one counter: with the staging at the top of the next unit the s_waitcnt vmcnt(0) in front of
        // it also waited for the 32 stores the wave had issued a moment before — a full store round trip per unit with nothing else
        // to do (round 6, second session: ablations in profiles/r06b_c3w64_ablations.txt).  Here the youngest stores in front of the
        // loads are a whole MFMA phase old.
        const bool last_of_tile = u + 1 < u1 && ((u + 1) >> 1) != t;
        if (last_of_tile) {
            store_tile(buf ^ 1);
            t_cur = t + 1;
            if (t_cur < t_last) load_tile(t_cur + 1);
        }
        // epilogue: window = {(h, w), (h, w+1), (h+1, w), (h+1, w+1)}, first maximum in that scan order wins (ATen
        // max_pool2d); lanes 0-31 store the pooled run of channel c, lanes 32-63 that of channel c + 4
        int n, h, w0;
        tile_coords(t, n, h, w0);
        const int cbase = ko0 + 32 * half;
        const int obase = ((n * Cout + cbase) * OH + (h >> 1)) * OW + (w0 >> 1);      // scalar part of the output offset
        const int ovoff = (4 * kk * chw + li);                                        // lane part (elements)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c0 = (r & 3) + 8 * (r >> 2);
#if CLHIP_C3W64_BIAS_IN_K
            const float bv = 0.f;                   // (already in the accumulators: see a[][13])
#else
            const float bv = bias_s[32 * half + 4 * kk + c0];
#endif
            // max(relu(.)) = relu(max(.)), and while the maximum is positive the first window position that holds it is the same before
            // and after the ReLU (smaller positions are <= it either way); a non-positive maximum is a dead window: same values and
            // codes as the scan over the four ReLU outputs, 15 instead of 21 VALU instructions per pooled value (they add to the
            // f32-MFMA time of the SIMD's other wave)
#if C3W64_ABL & 2          // timing only: no pooling arithmetic
            const float m2 = acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
            const bool cok2 = FULL || cbase + c0 + 4 * kk < Cout;
            clhip_buf_store(m2, r_o, cok2 ? ovoff * 4 : CLHIP_OOB, (obase + c0 * chw) * 4);
            clhip_buf_store_u8((uint8_t)r, r_i, cok2 ? ovoff : CLHIP_OOB, obase + c0 * chw);
            continue;
#endif
            const float tl = acc[0][0][r] + bv, tr = acc[0][1][r] + bv, bl = acc[1][0][r] + bv, br = acc[1][1][r] + bv;
            // (v_max3 / v_max by name: fmaxf() on an MFMA result first canonicalises it with a v_max_f32 x, x of its own)
            float mx, m;
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(tl), "v"(tr), "v"(bl));
            asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(br));
            int am = 3;
            am = bl == mx ? 2 : am;
            am = tr == mx ? 1 : am;
            am = tl == mx ? 0 : am;
            if (!(mx > 0.f)) am = CLHIP_POOL_DEAD;
            asm("v_max_f32 %0, 0, %1" : "=v"(m) : "v"(mx));
            const bool cok = FULL || cbase + c0 + 4 * kk < Cout;
#if C3W64_ABL & 1          // timing only: no stores (one lane keeps the values alive)
            if (m == 12345.678f) {
#endif
            clhip_buf_store(m, r_o, cok ? ovoff * 4 : CLHIP_OOB, (obase + c0 * chw) * 4);
            clhip_buf_store_u8((uint8_t)am, r_i, cok ? ovoff : CLHIP_OOB, obase + c0 * chw);
#if C3W64_ABL & 1
            }
#endif
        }
        if (last_of_tile) buf ^= 1;
    }
}

int launch_c3w64_pool(const float* x, const float* wt, const float* bias, float* out, uint8_t* idx,
                      int N, int Cout, int H, int W, hipStream_t s) {
    const int tiles_w = W / 64, row_pairs = H / 2;
    const long long ntiles = (long long)tiles_w * row_pairs * N;
    const int kts = (Cout + KT - 1) / KT;
    long long gx = 256;                             // one 8-wave block per CU
    if (gx * 8 > 2 * ntiles) gx = (2 * ntiles + 7) / 8;
    if (Cout % KT == 0)
        hipLaunchKernelGGL(conv3x3_c3w64_relu_pool_kernel<true>, dim3((unsigned)gx, (unsigned)kts), dim3(512), 0, s,
                           x, wt, bias, out, idx, N, Cout, H, W, tiles_w, row_pairs, (int)ntiles);
    else
        hipLaunchKernelGGL(conv3x3_c3w64_relu_pool_kernel<false>, dim3((unsigned)gx, (unsigned)kts), dim3(512), 0, s,
                           x, wt, bias, out, idx, N, Cout, H, W, tiles_w, row_pairs, (int)ntiles);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int launch_c3_pool(const float* x, const float* wt, const float* bias, float* out, uint8_t* idx,
                   int N, int Cout, int H, int W, hipStream_t s) {
    const int tiles_w = W / C3_TW, tiles_h = (H + C3_TH - 1) / C3_TH;
    const long long ntiles = (long long)tiles_w * tiles_h * N;
    const int kts = (Cout + KT - 1) / KT;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return CLHIP_EINVAL;
    long long gx = 512 / kts;                       // 2 resident blocks per CU over all channel tiles
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    hipLaunchKernelGGL(conv3x3_c3_relu_pool_kernel, dim3((unsigned)gx, (unsigned)kts), dim3(256), 0, s,
                       x, wt, bias, out, idx, N, Cout, H, W, tiles_w, tiles_h, (int)ntiles);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// 16-byte staging needs aligned rows and whole tiles along w.
bool vec_ok(const float* in, const float* wt, int Cin, int H, int W, int Cw) {
    const int TW = W > 16 ? 32 : (W > 8 ? 16 : 8);
    return (Cin % 8 == 0) && (Cw % 4 == 0) && (W % 4 == 0) && (W % TW == 0) && aligned16(in) && aligned16(wt);
}

}  // namespace

extern "C" {

#ifdef CLHIP_TRACE
int clhip_debug_set_conv_trace(void* p) {
    unsigned long long* q = static_cast<unsigned long long*>(p);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &q, sizeof(q));
}
#endif

int clhip_conv3x3_fwd(const float* x, const float* w, const float* b, float* y,
                      int N, int C, int K, int H, int W, int relu, void* stream) {
    if (!x || !w || !y || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    if (C <= 4) return launch_conv<4, 0, false>(x, w, b, nullptr, y, N, C, K, H, W, K, C, relu, s);
    if (vec_ok(x, w, C, H, W, C)) return launch_conv<8, 0, true>(x, w, b, nullptr, y, N, C, K, H, W, K, C, relu, s);
    return launch_conv<8, 0, false>(x, w, b, nullptr, y, N, C, K, H, W, K, C, relu, s);
}

int clhip_conv3x3_relu_pool_fwd(const float* x, const float* w, const float* b, float* y_pool, uint8_t* idx_u8,
                                int N, int C, int K, int H, int W, void* stream) {
    if (!x || !w || !y_pool || !idx_u8 || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1))
        return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    if (CLHIP_C3W64 && C == 3 && W % 64 == 0 && H % 2 == 0 && (size_t)N * 3 * H * W < ((size_t)1 << 29) &&
        (size_t)N * K * (H / 2) * (W / 2) < ((size_t)1 << 29))
        return launch_c3w64_pool(x, w, b, y_pool, idx_u8, N, K, H, W, s);
    if (C == 3 && W % C3_TW == 0) return launch_c3_pool(x, w, b, y_pool, idx_u8, N, K, H, W, s);

    if (C <= 4) return launch_conv<4, 0, false>(x, w, b, nullptr, y_pool, N, C, K, H, W, K, C, 1, s, idx_u8);
    if (vec_ok(x, w, C, H, W, C)) return launch_conv<8, 0, true>(x, w, b, nullptr, y_pool, N, C, K, H, W, K, C, 1, s, idx_u8);
    return launch_conv<8, 0, false>(x, w, b, nullptr, y_pool, N, C, K, H, W, K, C, 1, s, idx_u8);
}

int clhip_conv3x3_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx,
                           int N, int C, int K, int H, int W, void* stream) {
    if (!dy || !w || !dx || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return CLHIP_EINVAL;
    // in = dy (K channels), out = dx (C channels)
    if (vec_ok(dy, w, K, H, W, C))
        return launch_conv<8, 1, true>(dy, w, nullptr, relu_src, dx, N, K, C, H, W, K, C, 0, as_stream(stream));
    return launch_conv<8, 1, false>(dy, w, nullptr, relu_src, dx, N, K, C, H, W, K, C, 0, as_stream(stream));
}

// Backward-data of a conv whose ReLU output was 2x2-max-pooled, straight from the gradient w.r.t. the POOLED output and
// the arg-max codes (no un-pooled gradient tensor, no clhip_maxpool2_bwd launch).  Aligned shapes only (the 16-byte
// staging path): CLHIP_ENOTSUP otherwise, callers then un-pool first.
int clhip_conv3x3_bwd_data_unpool(const float* dy_pool, const uint8_t* idx_u8, const float* w, const float* relu_src, float* dx,
                                  int N, int C, int K, int H, int W, void* stream) {
    if (!dy_pool || !idx_u8 || !w || !dx || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return CLHIP_EINVAL;
    if ((H & 1) || (W & 1) || !vec_ok(dy_pool, w, K, H, W, C) || (reinterpret_cast<uintptr_t>(idx_u8) & 1u)) return CLHIP_ENOTSUP;
    return launch_conv<8, 1, true, true>(dy_pool, w, nullptr, relu_src, dx, N, K, C, H, W, K, C, 0, as_stream(stream),
                                         const_cast<uint8_t*>(idx_u8));
}

}  // extern "C"
