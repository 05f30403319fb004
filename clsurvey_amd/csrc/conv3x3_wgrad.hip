// Weight/bias gradient of the 3x3 pad-1 convolution on v_mfma_f32_32x32x2_f32.
//
//   dw[k][c][r][s] = sum_{n,h,w} dy[n][k][h][w] * x[n][c][h+r-1][w+s-1]     db[k] = sum dy[n][k][h][w]
//
// GEMM view: D rows = 32 output channels k (A operand = dy), D cols = 32 input channels c
// (B operand = x shifted by the tap), reduction (MFMA K) = pixels, two per instruction.
// One A fragment feeds NINE MFMAs (one accumulator per tap): 10 ds_read_b32 per 9 MFMAs.
//
// The reduction dimension (N*H*W, up to 819 200) is split across blocks (about one block per
// CU: the kernel is MFMA-bound and software-pipelined — the next 64-pixel stage is prefetched
// into registers while the current one feeds the matrix pipe).  Every block writes a partial
// [9][K][C](+[K]) slab into the caller's workspace; two small kernels then sum the slabs in a
// fixed order (grouped, then across groups) => bitwise run-to-run deterministic, as the
// reference promises by seeding everything and setting cudnn.deterministic
// (utilities/utils.py:52-58).
//
// First layer (C*9 <= 32): the 27 (c,r,s) combinations become the D columns of ONE accumulator
// (84 % of the MFMA columns useful instead of 9 %); that kernel is HBM-bound on reading dy.
#include <cstdlib>
#include "common.hpp"

namespace {

constexpr int KT = 64, CT = 64;

#ifdef CLHIP_TRACE
// tools/trace_conv.py only (never in the product build): per-wave cycle stamps
__device__ unsigned long long* g_wtrace = nullptr;
#define TR_NOW() __builtin_amdgcn_s_memtime()
#endif

template <int TW, int TH>
struct WGeo {
    static constexpr int BP = TW * TH;                 // 64 pixels per stage, one image
    static constexpr int TWP = TW + 2;
    static constexpr int PLANE = (TH + 2) * TWP;
    static constexpr int PLANEP = PLANE | 1;           // odd stride: conflict-free lane-per-channel reads
    static constexpr int LDP = BP + 1;                 // dy tile row stride (odd)
    static_assert(BP == 64, "stage is 64 pixels");
};

__device__ __forceinline__ void decode_stage(int st, int tiles_w, int tiles_h, int TW, int TH,
                                             int& n, int& h0, int& w0) {
    int tw_i = st % tiles_w;
    int t = st / tiles_w;
    int th_i = t % tiles_h;
    n = t / tiles_h;
    h0 = th_i * TH;
    w0 = tw_i * TW;
}

// grid.x = n_tiles(k,c) * splits ; block 256 = 4 waves.
//   PS = false: 64x64 (k,c) tile, waves = (2 k-halves) x (2 c-halves), every wave sees all 64 pixels of a stage;
//   PS = true ("pixel split", deep layers with few pixels): 32x32 tile, the four waves take a quarter of each stage's
//        pixels and their accumulators are added through LDS in wave order at the end.  A 4x smaller slab per block and
//        4x more stages per block: at 16x16 / 8x8 the 64x64 form left 1.5-3 stages per block (20-30 % lost to the
//        whole-stage quantisation) and moved 38 MB of partial slabs per layer.
//
// Pipeline: two LDS stage buffers, ONE barrier per stage, and all staging work issued one piece per MFMA "slot" in
// the shadow of the 64-cycle MFMAs of the current stage (the phase-separated form of this loop — barrier, LDS writes,
// barrier, load issue, MFMAs — measured 71 % of its time in the MFMA phase with the pixel split, 86 % without):
//   slots 0 .. NU-1      : LDS writes of stage st+1 (loaded during stage st-1) into the other buffer
//   slots NU .. 2 NU-1   : global (raw buffer) loads of stage st+2 into the staging registers just freed
//   every pixel pair     : the LDS operand reads of the NEXT pair (register double buffering)
//   last pair, after the barrier: the operand reads of stage st+1's first pair from the other buffer
// The barrier sits behind the first MFMA of the last pixel pair: every read of the current buffer has been issued and
// waited for by then (the last pair's operands are in registers), and so have the writes of stage st+1.
#ifndef CLHIP_WGRAD_PS_WAVES
#define CLHIP_WGRAD_PS_WAVES 1
#endif
// UNPOOL (VEC only): `dy` is the gradient w.r.t. the 2x2-max-POOLED output [N][K][H/2][W/2] and `unpool_idx` the arg-max
// codes of the forward pass; the un-pooled dy tile is rebuilt while it is staged (fused max_pool2d backward).
template <int TW, int TH, bool VEC, bool PS, bool UNPOOL = false>
__global__ __launch_bounds__(256, CLHIP_WGRAD_PS_WAVES) void conv3x3_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h,
    int total_stages, int splits, int c_tiles, size_t slab_stride, const uint8_t* __restrict__ unpool_idx) {
    static_assert(!UNPOOL || VEC, "the fused un-pool lives in the 16-byte staging path");
    using G = WGeo<TW, TH>;
    constexpr int KTt = PS ? 32 : 64, CTt = PS ? 32 : 64;
    constexpr int DYS_FLOATS = KTt * G::LDP, XS_FLOATS = CTt * G::PLANEP;
    constexpr int BUF_FLOATS = DYS_FLOATS + XS_FLOATS;
    static_assert(!PS || 2 * BUF_FLOATS >= 9 * 1024 + 3 * 128, "the pixel-split reduction pad lives in the stage buffers");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CLHIP_TRACE
    const unsigned long long tr_start = TR_NOW();
#endif
    const int wk = PS ? 0 : (wave & 1), wc = PS ? 0 : (wave >> 1);
    const int li = lane & 31, kk = lane >> 5;

    const int split = blockIdx.x % splits;
    const int tile = blockIdx.x / splits;
    const int ct = tile % c_tiles, kt = tile / c_tiles;
    const int k0 = kt * KTt, c0 = ct * CTt;

    // contiguous, balanced stage range of this split
    const int per = total_stages / splits, extra = total_stages % splits;
    const int st_begin = split * per + min(split, extra);
    const int st_end = st_begin + per + (split < extra ? 1 : 0);

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    double bsum = 0.0;      // bias sums cancel heavily: accumulate in f64 (VALU has slack)

    const int plane_hw = H * W;
    const int Wp = W >> 1, plane_dy = UNPOOL ? (H >> 1) * Wp : plane_hw;       // plane of the tensor `dy` points at

    // ------------------------------------------------------------------ staging units
    // Every unit is ONE raw buffer load per thread (+ the LDS writes of its result).  The buffer of a stage starts at
    // the top-left HALO element of the (image, channel block) the stage reads, so all per-thread offsets are constants
    // >= 0 and the per-stage part is the descriptor; the hardware range check masks the channel tail (k >= K, c >= C),
    // the halo rows / columns outside the image get CLHIP_OOB by predicate (one v_cndmask, nothing waits on a load).
    constexpr int XROWS = CTt * (TH + 2);
    constexpr int DY_IT = VEC ? (KTt * G::BP / 4) / 256 : KTt * G::BP / 256;        // 4 | 16 (PS: 2 | 8)
    constexpr int X_ELEMS = CTt * G::PLANE;
    constexpr int XV_ELEMS = XROWS * (TW / 4);
    constexpr int X_IT = VEC ? (XV_ELEMS + 255) / 256 : (X_ELEMS + 255) / 256;
    constexpr int H_IT = VEC ? (XROWS * 2 + 255) / 256 : 0;
    constexpr int NU = DY_IT + X_IT + H_IT;
    float4 dyv[VEC ? DY_IT : 1], xv[VEC ? X_IT : 1];
    float hv[H_IT > 0 ? H_IT : 1];
    float dyr[VEC ? 1 : DY_IT], xr[VEC ? 1 : X_IT];
    int dyoff[DY_IT], dydst[DY_IT], dyrow[DY_IT];      // byte offset | LDS float offset | tile row (scalar path: | col << 8)
    float2 dyp[UNPOOL ? DY_IT : 1];                    // UNPOOL: pooled pair + its two arg-max bytes per unit
    unsigned dyi[UNPOOL ? DY_IT : 1];
    int xoff[X_IT], xdst[X_IT], xrc[X_IT];             // byte offset from the halo origin | LDS float offset | halo row | col << 8
    int hoff[H_IT > 0 ? H_IT : 1], hdst[H_IT > 0 ? H_IT : 1], hrc[H_IT > 0 ? H_IT : 1];

    if constexpr (VEC) {
#pragma unroll
        for (int j = 0; j < DY_IT; ++j) {
            const int e = tid + 256 * j;                     // float4 index: (kl, 16 float4 per k)
            const int kl = e / (G::BP / 4), f = e - kl * (G::BP / 4);
            const int q = 4 * f, th = q / TW, tw = q - th * TW;
            dyoff[j] = UNPOOL ? (kl * plane_dy + (th >> 1) * Wp + (tw >> 1)) * 4 : (kl * plane_hw + th * W + tw) * 4;
            dydst[j] = kl * G::LDP + q;
            dyrow[j] = th;
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e / (TW / 4), f = e - rowid * (TW / 4);
            const int cl = rowid / (TH + 2), row = rowid - cl * (TH + 2);
            xoff[j] = e < XV_ELEMS ? (cl * plane_hw + row * W + 1 + 4 * f) * 4 : CLHIP_OOB;
            xdst[j] = cl * G::PLANEP + row * G::TWP + 1 + 4 * f;
            xrc[j] = row;
        }
#pragma unroll
        for (int j = 0; j < H_IT; ++j) {
            const int e = tid + 256 * j;
            const int rowid = e >> 1, side = e & 1;
            const int cl = rowid / (TH + 2), row = rowid - cl * (TH + 2);
            hoff[j] = e < XROWS * 2 ? (cl * plane_hw + row * W + (side ? TW + 1 : 0)) * 4 : CLHIP_OOB;
            hdst[j] = cl * G::PLANEP + row * G::TWP + (side ? TW + 1 : 0);
            hrc[j] = row | (side << 8);
        }
    } else {
        const int q_t = tid & 63;
        const int th_t = q_t / TW, tw_t = q_t - th_t * TW;
#pragma unroll
        for (int j = 0; j < DY_IT; ++j) {
            const int kl = wave + 4 * j;
            dyoff[j] = (kl * plane_hw + th_t * W + tw_t) * 4;
            dydst[j] = kl * G::LDP + q_t;
            dyrow[j] = th_t | (tw_t << 8);
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int e = tid + 256 * j;
            xoff[j] = CLHIP_OOB; xdst[j] = 0; xrc[j] = 0;
            if (e < X_ELEMS) {
                const int cl = e / G::PLANE, rem = e - cl * G::PLANE;
                const int row = rem / G::TWP, col = rem - row * G::TWP;
                xoff[j] = (cl * plane_hw + row * W + col) * 4;
                xdst[j] = cl * G::PLANEP + rem;
                xrc[j] = row | (col << 8);
            }
        }
    }

    // stages are visited in order: decode the first one with divisions, then step (tile column, tile row, image)
    int cur_n, cur_th, cur_tw;
    {
        const int st0 = st_begin < st_end ? st_begin : 0;
        cur_tw = st0 % tiles_w;
        const int t0 = st0 / tiles_w;
        cur_th = t0 % tiles_h;
        cur_n = t0 / tiles_h;
    }
    // descriptors + tile origin of the stage whose loads are being issued (set by begin_stage, used by load_unit).
    // The (image, channel block) base pointers step by one image when the stage walk wraps; per stage only the tile
    // origin inside the plane changes (32-bit), so a stage costs ~25 scalar instructions, not two 64-bit multiplies.
    __amdgpu_buffer_rsrc_t rs_dy = clhip_rsrc(dy, 0), rs_x = clhip_rsrc(x, 0), rs_di = clhip_rsrc(x, 0);
    int ld_h0 = 0, ld_w0 = 0;
    const float* dy_img = dy + ((size_t)cur_n * K + k0) * plane_dy;
    const uint8_t* di_img = UNPOOL ? unpool_idx + ((size_t)cur_n * K + k0) * plane_dy : nullptr;
    const float* x_img = x + ((size_t)cur_n * C + c0) * plane_hw;
    const long long dy_blk = (long long)(K - k0) * plane_dy, x_blk = (long long)(C - c0) * plane_hw;   // floats to the end of the block
    auto begin_stage = [&](bool live) {
        const int h0 = cur_th * TH, w0 = cur_tw * TW;
        ld_h0 = h0; ld_w0 = w0;
        const int org = h0 * W + w0;                                         // tile origin inside a plane
        const int org_dy = UNPOOL ? (h0 >> 1) * Wp + (w0 >> 1) : org;         // ... of the (pooled) dy plane
        const long long dy_left = dy_blk - org_dy, x_left = x_blk - (org - W - 1);
        rs_dy = clhip_rsrc(dy_img + org_dy, live && dy_left > 0 ? (size_t)dy_left * 4 : 0);
        if constexpr (UNPOOL) rs_di = clhip_rsrc(di_img + org_dy, live && dy_left > 0 ? (size_t)dy_left : 0);
        rs_x = clhip_rsrc(x_img + org - W - 1, live && x_left > 0 ? (size_t)x_left * 4 : 0);
        if (++cur_tw == tiles_w) {
            cur_tw = 0;
            if (++cur_th == tiles_h) {
                cur_th = 0; ++cur_n; dy_img += (size_t)K * plane_dy; x_img += (size_t)C * plane_hw;
                if constexpr (UNPOOL) di_img += (size_t)K * plane_dy;
            }
        }
    };
    auto load_unit = [&](int u) {
#ifdef CLHIP_ABL_NOLOAD
        if (ld_h0 >= 0) return;
#endif
        const int h0 = ld_h0, w0 = ld_w0;
        if constexpr (VEC) {
            if (u < DY_IT) {
                if constexpr (UNPOOL) {
                    dyp[u] = clhip_buf_load2(rs_dy, h0 + dyrow[u] < H ? dyoff[u] : CLHIP_OOB, 0);
                    dyi[u] = clhip_buf_load_u16(rs_di, h0 + dyrow[u] < H ? dyoff[u] >> 2 : CLHIP_OOB, 0);
                } else {
                    dyv[u] = clhip_buf_load4(rs_dy, h0 + dyrow[u] < H ? dyoff[u] : CLHIP_OOB, 0);
                }
            } else if (u < DY_IT + X_IT) {
                const int j = u - DY_IT;
                xv[j] = clhip_buf_load4(rs_x, (unsigned)(h0 - 1 + xrc[j]) < (unsigned)H ? xoff[j] : CLHIP_OOB, 0);
            } else {
                const int j = u - DY_IT - X_IT;
                const int h = h0 - 1 + (hrc[j] & 255), w = (hrc[j] >> 8) ? w0 + TW : w0 - 1;
                hv[j] = clhip_buf_load(rs_x, ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? hoff[j] : CLHIP_OOB, 0);
            }
        } else {
            if (u < DY_IT) {
                const bool ok = (h0 + (dyrow[u] & 255) < H) && (w0 + (dyrow[u] >> 8) < W);
                dyr[u] = clhip_buf_load(rs_dy, ok ? dyoff[u] : CLHIP_OOB, 0);
            } else {
                const int j = u - DY_IT;
                const int h = h0 - 1 + (xrc[j] & 255), w = w0 - 1 + (xrc[j] >> 8);
                xr[j] = clhip_buf_load(rs_x, ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? xoff[j] : CLHIP_OOB, 0);
            }
        }
    };
    auto store_unit = [&](int u, int bo) {
#ifdef CLHIP_ABL_NOSTORE
        if (bo >= 0) return;
#endif
        float* dys = lds + bo;
        float* xs = dys + DYS_FLOATS;
        if constexpr (VEC) {
            if (u < DY_IT) {
                float* d = dys + dydst[u];
                if constexpr (UNPOOL) {
                    // four columns of one row = two pooling windows (tile rows start at even image rows)
                    const int c = (dyrow[u] & 1) << 1, i0 = (int)(dyi[u] & 0xffu), i1 = (int)((dyi[u] >> 8) & 0xffu);
                    d[0] = i0 == c ? dyp[u].x : 0.f; d[1] = i0 == c + 1 ? dyp[u].x : 0.f;
                    d[2] = i1 == c ? dyp[u].y : 0.f; d[3] = i1 == c + 1 ? dyp[u].y : 0.f;
                } else {
                    d[0] = dyv[u].x; d[1] = dyv[u].y; d[2] = dyv[u].z; d[3] = dyv[u].w;
                }
            } else if (u < DY_IT + X_IT) {
                const int j = u - DY_IT;
                if (256 * (j + 1) <= XV_ELEMS || tid + 256 * j < XV_ELEMS) {
                    float* d = xs + xdst[j];
                    d[0] = xv[j].x; d[1] = xv[j].y; d[2] = xv[j].z; d[3] = xv[j].w;
                }
            } else {
                const int j = u - DY_IT - X_IT;
                if (256 * (j + 1) <= XROWS * 2 || tid + 256 * j < XROWS * 2) xs[hdst[j]] = hv[j];
            }
        } else {
            if (u < DY_IT) dys[dydst[u]] = dyr[u];
            else {
                const int j = u - DY_IT;
                if (256 * (j + 1) <= X_ELEMS || tid + 256 * j < X_ELEMS) xs[xdst[j]] = xr[j];
            }
        }
    };

    // ------------------------------------------------------------------ main loop
    // PS: this wave's quarter of the stage starts at pixel 16*wave (whole rows or half rows for every TW)
    constexpr int NPP = PS ? G::BP / 8 : G::BP / 2;
    constexpr int SLOTS = NPP * 9;
    constexpr int UPS = (2 * NU + SLOTS - 10) / (SLOTS - 9);     // units per slot so that stores + loads end before the last pair
    const int pb = PS ? 16 * wave : 0;
    const int a_off = (wk * 32 + li) * G::LDP + kk + pb;
    const int b_off = DYS_FLOATS + (wc * 32 + li) * G::PLANEP + kk + (pb / TW) * G::TWP + (pb % TW);
    float af[2], bf[2][9];
    auto frag = [&](const float* base, int pp, int slot, int rs) {      // operand rs of pixel pair pp (rs = 0 also loads A)
        const int q0 = 2 * pp;
        const int th = q0 / TW, tw = q0 - (q0 / TW) * TW;
        const int r = rs / 3, s3 = rs - 3 * (rs / 3);
        if (rs == 0) af[slot] = base[a_off + q0];
        bf[slot][rs] = base[b_off + (th + r) * G::TWP + tw + s3];
    };

#ifdef CLHIP_TRACE
    const unsigned long long tr_idx = TR_NOW();
#endif
    if (st_begin < st_end) {
        begin_stage(true);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
#pragma unroll
        for (int u = 0; u < NU; ++u) store_unit(u, 0);
        begin_stage(st_begin + 1 < st_end);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        __syncthreads();
#pragma unroll
        for (int rs = 0; rs < 9; ++rs) frag(lds, 0, 0, rs);
    }
    for (int st = st_begin; st < st_end; ++st) {
        const int bo = ((st - st_begin) & 1) * BUF_FLOATS, bn = BUF_FLOATS - bo;
        const float* cur = lds + bo;
        const float* nxt = lds + bn;
#pragma unroll
        for (int pp = 0; pp < NPP; ++pp) {
            bsum += (double)af[pp & 1];      // f64: 819 200 cancelling terms per channel on the first layers
#pragma unroll
            for (int rs = 0; rs < 9; ++rs) {
                const int slot = pp * 9 + rs;
                acc[rs] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp & 1], bf[pp & 1][rs], acc[rs], 0, 0, 0);
#ifndef CLHIP_ABL_NOBAR
                if (pp == NPP - 1 && rs == 0) __syncthreads();
#endif
#ifndef CLHIP_ABL_NOFRAG
                if (pp + 1 < NPP) frag(cur, pp + 1, (pp + 1) & 1, rs);
                else frag(nxt, 0, 0, rs);
#endif
#pragma unroll
                for (int u = slot * UPS; u < (slot + 1) * UPS; ++u) {
                    if (u < NU) store_unit(u, bn);
                    else if (u < 2 * NU) {
#ifndef CLHIP_ABL_NOBEGIN
                        if (u == NU) begin_stage(st + 2 < st_end);
#endif
                        load_unit(u - NU);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef CLHIP_TRACE
    const unsigned long long tr_loop = TR_NOW();
    auto tr_finish = [&]() {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long tr_end = TR_NOW();
        if (g_wtrace && lane == 0) {
            unsigned long long* t = g_wtrace + ((size_t)blockIdx.x * 4 + wave) * 16;
            t[0] = tr_start; t[1] = tr_idx; t[2] = tr_idx; t[3] = tr_loop; t[4] = tr_end;
            t[5] = 0; t[6] = 0; t[7] = 0; t[8] = 0;
            t[9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            t[10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
            t[11] = st_end - st_begin;
        }
    };
#endif

    float* slab = part + (size_t)split * slab_stride;
    if constexpr (PS) {
        // add the four pixel quarters in wave order ((0 + 1) + 2) + 3 (fixed => deterministic).  Three rounds of three
        // taps through a 36 KB pad at the start of the stage buffers: every wave parks the taps it does not own, then
        // wave w (< 3) sums tap 3*round + w in that order — its own quarter from registers — and writes it to the slab,
        // so three waves share the additions and the 144 slab stores per lane that one wave used to do alone.
        __syncthreads();                                   // the last stage's LDS reads are done
        static_assert(2 * BUF_FLOATS >= 9 * 1024 + 3 * 128, "pad [3 taps][3 other waves][16 regs][64 lanes] + bias pad");
        float* pad = lds;
        double* bpad = reinterpret_cast<double*>(lds + 9 * 1024);     // [3 waves][64 lanes]
        bsum += __shfl_xor(bsum, 32, 64);                  // even + odd pixels of this wave
        if (wave > 0) bpad[(wave - 1) * 64 + lane] = bsum;
        const int c = c0 + li;
#pragma unroll
        for (int round = 0; round < 3; ++round) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                // slot of wave `wave` in tap t's pad: the three waves other than the owner t, in ascending order
                if (wave != t) {
                    const int sl = wave - (wave > t ? 1 : 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) pad[((t * 3 + sl) * 16 + r) * 64 + lane] = acc[3 * round + t][r];
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (wave == t) {
                    const int rs = 3 * round + t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // contributions in wave order 0, 1, 2, 3; this wave's own sits in registers at position t
                        float v = 0.f;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const float contrib = (w == t) ? acc[rs][r] : pad[((t * 3 + (w - (w > t ? 1 : 0))) * 16 + r) * 64 + lane];
                            v = (w == 0) ? contrib : v + contrib;
                        }
                        const int k = k0 + mfma32_row(r, lane);
                        if (k < K && c < C) slab[((size_t)rs * K + k) * C + c] = v;
                    }
                }
            }
            if (round < 2) __syncthreads();
        }
        if (wave == 0 && ct == 0) {
            const double tot = ((bsum + bpad[0 * 64 + lane]) + bpad[1 * 64 + lane]) + bpad[2 * 64 + lane];
            const int k = k0 + li;
            if (kk == 0 && k < K) slab[(size_t)9 * K * C + k] = (float)tot;
        }
#ifdef CLHIP_TRACE
        tr_finish();
#endif
        return;
    }

    // ---- partial slab [split]{[9][K][C], [K]}: reg r of lane l = D[row = k][col = c]
    const int c = c0 + wc * 32 + li;
#pragma unroll
    for (int rs = 0; rs < 9; ++rs)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int k = k0 + wk * 32 + mfma32_row(r, lane);
            if (k < K && c < C) slab[((size_t)rs * K + k) * C + c] = acc[rs][r];
        }
    if (ct == 0 && wc == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        int k = k0 + wk * 32 + li;
        if (kk == 0 && k < K) slab[(size_t)9 * K * C + k] = (float)bsum;
    }
#ifdef CLHIP_TRACE
    tr_finish();
#endif
}

// First-layer variant: C*9 <= 32 columns in one accumulator. Block = 256 threads = 4 waves:
// (2 k-halves of a 64-channel tile) x (2 halves of the stage's pixel pairs); the two pixel
// halves are summed through LDS at the end.  HBM-bound on dy: stages are prefetched.
template <int TW, int TH>
__global__ __launch_bounds__(256) void conv3x3_wgrad_smallc_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h,
    int total_stages, int splits, size_t slab_stride, const uint8_t* __restrict__ unpool_idx) {
    // unpool_idx != NULL: `dy` is the gradient w.r.t. the 2x2-max-POOLED output [N][K][H/2][W/2] and the
    // un-pooled gradient tile is rebuilt on the fly from the 2-bit argmax (fused maxpool backward: the
    // 4x larger dy tensor of the first layer is never written nor read).
    using G = WGeo<TW, TH>;
    __shared__ float dys[KT * G::LDP];
    __shared__ float xs[3 * G::PLANE + 8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wh = wave >> 1;
    const int li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int kt = blockIdx.x / splits;
    const int k0 = kt * KT;
    const int per = total_stages / splits, extra = total_stages % splits;
    const int st_begin = split * per + min(split, extra);
    const int st_end = st_begin + per + (split < extra ? 1 : 0);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    double bsum = 0.0;      // bias sums cancel heavily: accumulate in f64 (VALU has slack)

    // column li -> (c, r, s); columns >= C*9 read a valid dummy address, never stored
    const int ncol = C * 9;
    int col_off = 0;
    if (li < ncol) {
        int cc = li / 9, rs = li - cc * 9;
        col_off = cc * G::PLANE + (rs / 3) * G::TWP + (rs % 3);
    }
    const float* a_ptr = dys + (wk * 32 + li) * G::LDP + kk + wh * (G::BP / 2);
    const float* b_ptr = xs + col_off + kk;
    const size_t plane_hw = (size_t)H * W;

    constexpr int DY_ITERS = KT * G::BP / 256;                 // 16
    constexpr int X_ELEMS = 3 * G::PLANE;
    constexpr int X_ITERS = (X_ELEMS + 255) / 256;             // 2
    float dyr[DY_ITERS];
    float xr[X_ITERS];
    int upa[4];

    const int q_t = tid & 63;
    const int th_t = q_t / TW, tw_t = q_t - th_t * TW;
    const int dy_lds = wave * G::LDP + q_t;
    int xoff[X_ITERS], xmeta[X_ITERS];
#pragma unroll
    for (int j = 0; j < X_ITERS; ++j) {
        int e = tid + 256 * j;
        xoff[j] = 0; xmeta[j] = 0;
        if (e < X_ELEMS) {
            int cl = e / G::PLANE, rem = e - cl * G::PLANE;
            int row = rem / G::TWP, col = rem - row * G::TWP;
            xoff[j] = cl * (int)plane_hw + (row - 1) * W + (col - 1);
            xmeta[j] = row | (col << 4) | (cl << 10);
        }
    }

    // stages are visited in order: decode the first one with divisions, then step (tile column, tile row, image) —
    // the ~135 scalar instructions of a full decode per stage sat in front of every stage's MFMAs of this lone wave
    int cur_n, cur_th, cur_tw;
    {
        const int st0 = st_begin < st_end ? st_begin : 0;
        cur_tw = st0 % tiles_w;
        const int t0 = st0 / tiles_w;
        cur_th = t0 % tiles_h;
        cur_n = t0 / tiles_h;
    }
    auto load_stage = [&](int) {
        const int n = cur_n, h0 = cur_th * TH, w0 = cur_tw * TW;
        if (++cur_tw == tiles_w) { cur_tw = 0; if (++cur_th == tiles_h) { cur_th = 0; ++cur_n; } }
        const bool pix_ok = (h0 + th_t < H) && (w0 + tw_t < W);
        const float* dyp = dy + ((size_t)n * K + k0 + wave) * plane_hw + (size_t)(h0 + th_t) * W + (w0 + tw_t);
        const float* xp = x + (size_t)n * C * plane_hw + (size_t)h0 * W + w0;
        if (unpool_idx) {
            const int OH = H >> 1, OW = W >> 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = tid + 256 * j;                          // 64 k x 16 pooled windows
                const int kl = e >> 4, pwin = e & 15;
                const int ph = pwin / (TW / 2), pw = pwin - ph * (TW / 2);
                const int k = k0 + kl, hh = (h0 >> 1) + ph, ww = (w0 >> 1) + pw;
                const bool ok = k < K && hh < OH && ww < OW;
                const size_t o = (((size_t)n * K + k) * OH + hh) * OW + ww;
                dyr[j] = *(ok ? dy + o : clhip_zero16);              // zero gradient => all four window slots 0
                upa[j] = (int)*(ok ? unpool_idx + o : reinterpret_cast<const uint8_t*>(clhip_zero16));
            }
        } else {
#pragma unroll
            for (int j = 0; j < DY_ITERS; ++j)
                dyr[j] = *((pix_ok && (k0 + wave + 4 * j < K)) ? dyp + (size_t)(4 * j) * plane_hw : clhip_zero16);
        }
#pragma unroll
        for (int j = 0; j < X_ITERS; ++j) {
            const int mt = xmeta[j];
            const int h = h0 - 1 + (mt & 15), w = w0 - 1 + ((mt >> 4) & 63);
            const bool ok = (tid + 256 * j < X_ELEMS) && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W &&
                            (((mt >> 10) & 127) < C);
            xr[j] = *(ok ? xp + xoff[j] : clhip_zero16);
        }
    };
    auto store_stage = [&]() {
        if (unpool_idx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = tid + 256 * j;
                const int kl = e >> 4, pwin = e & 15;
                const int ph = pwin / (TW / 2), pw = pwin - ph * (TW / 2);
                float* d = dys + kl * G::LDP + (2 * ph) * TW + 2 * pw;
                const float g = dyr[j];
                const int a = upa[j];
                d[0] = a == 0 ? g : 0.f; d[1] = a == 1 ? g : 0.f;
                d[TW] = a == 2 ? g : 0.f; d[TW + 1] = a == 3 ? g : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < DY_ITERS; ++j) dys[dy_lds + 4 * j * G::LDP] = dyr[j];
        }
#pragma unroll
        for (int j = 0; j < X_ITERS; ++j)
            if (tid + 256 * j < X_ELEMS) xs[tid + 256 * j] = xr[j];
    };

    if (st_begin < st_end) load_stage(st_begin);
    for (int st = st_begin; st < st_end; ++st) {
        __syncthreads();
        store_stage();
        __syncthreads();
        if (st + 1 < st_end) load_stage(st + 1);
#pragma unroll
        for (int pp = 0; pp < G::BP / 4; ++pp) {
            // this wave's half of the stage: pixel = wh*32 + ql (+kk); 32 is a multiple of TW, so the
            // half offset is whole rows
            const int ql = 2 * pp;
            float a = a_ptr[ql];
            bsum += (double)a;
            const int th = ql / TW, tw = ql - (ql / TW) * TW;
            float b = b_ptr[(th + wh * (32 / TW)) * G::TWP + tw];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // combine the two pixel halves through LDS (reuse dys), fixed order: half 0 + half 1
    bsum += __shfl_xor(bsum, 32, 64);
    __syncthreads();                                             // last stage's LDS reads are done
    float* red = dys;                                            // [2 k-halves][16 regs][64 lanes] + [64] bias
    if (wh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wk * 16 + r) * 64 + lane] = acc[r];
        if (kk == 0) red[2048 + wk * 32 + li] = (float)bsum;
    }
    __syncthreads();
    if (wh == 0) {
        float* slab = part + (size_t)split * slab_stride;
        if (li < ncol) {
            int cc = li / 9, rs = li - cc * 9;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int k = k0 + wk * 32 + mfma32_row(r, lane);
                if (k < K) slab[((size_t)rs * K + k) * C + cc] = acc[r] + red[(wk * 16 + r) * 64 + lane];
            }
        }
        int k = k0 + wk * 32 + li;
        if (kk == 0 && k < K) slab[(size_t)9 * K * C + k] = (float)(bsum + (double)red[2048 + wk * 32 + li]);
    }
}

// First layer with the fused max-pool backward, rows of whole 32-pixel tiles: the un-pooled gradient is never built.
//
// The kernel above expands every pooled gradient into its 2x2 window in LDS (4 compares + 4 selects + 4 LDS writes per
// pooled element, two block barriers per 64-pixel stage) before 16 MFMAs per wave can run: 0.44 MFMA-busy.  Here the A
// operand of an MFMA (32 channels x the two pixels 2wp, 2wp+1 of image row h0 + wh) is formed in registers: lane
// (channel, kk) holds the pooled gradient g and arg-max code of window wp of ITS channel (16 windows of a stage = four
// 16-byte loads + one 16-byte load of codes per lane) and supplies  a = (code == 2 wh + kk) ? g : 0.  The x halo of the
// wave's image row (3 channels x 3 rows x 34 columns) is wave-private in LDS, double buffered: no block barrier in the
// stage loop, the four waves drift apart.  Stage order, pixel-pair order and the final addition of the two row halves are those of
// conv3x3_wgrad_smallc_kernel<32, 2> => the weight-gradient part of every slab holds the same bits (the bias sums: see
// CLHIP_U3_BSUM_MFMA below).
#ifndef CLHIP_U3
#define CLHIP_U3 1
#endif
// 1: the bias gradient (sum of the un-pooled gradient over the pixels) rides on the matrix pipe — column C * 9 of the B operand, one of
// the 32 - 27 spare ones, reads 1.0, so accumulator column C * 9 IS the sum of the A operands: no v_cvt_f64_f32 + v_add_f64 per MFMA
// in the stage loop (32 of its ~80 VALU instructions per 16 MFMAs).  The block's partial sum is then an fp32 MFMA chain (a few hundred
// terms) instead of an f64 sum; the slabs are still added in f64 (wgrad_reduce_block).  0: f64 sums as conv3x3_wgrad_smallc_kernel.
// Measured on the LDS-staged build (N = 200, tools/experiments/r06_b6.sh): 45.0 -> 43.3 us; the bias gradient of a 200-image batch
// differs from the f64 sum by 1.9e-4 at |db| = 1e3 (2e-7 relative; the f64 form: 8e-5), tools/experiments/u3_dump.py.
// Default 0 all the same: any change of the last bit of a gradient re-draws the outcome of the bench's ill-conditioned 10-task sweep
// (with this form on, the sweep's task 2 is accepted above the stability limit and the sequence ends without a model at
// task 4), and the recorded draw is that of the f64 sums.
#ifndef CLHIP_U3_BSUM_MFMA
#define CLHIP_U3_BSUM_MFMA 0
#endif
// 1: the pooled gradient and its arg-max codes reach the lanes through a wave-private LDS image instead of straight from memory.  A
// lane of the A operand is a CHANNEL: loading "the 16 windows of my channel" puts every lane of a load instruction on a cache line of
// its own (channel planes are 4 KB apart) — 5 instructions x 64 lines per stage and wave, ~400 tag look-ups of the CU's vector cache
// per 1024 matrix cycles, 16 waves per CU: the cache's address pipe, not the matrix pipe, set the pace (removing 75 % of the loop's
// VALU instructions moved the launch by 4 %).  Here four lanes share a channel's 64-byte run (16 channels per instruction: 2 x 16 lines
// for the gradient, 32 half-lines for the codes), write it to LDS rows of 80 bytes, and lane (channel, half) reads its row back with
// five conflict-free ds_read_b128.  Same values into the same MFMAs: the slabs hold the same bits.
#ifndef CLHIP_U3_LDSDY
#define CLHIP_U3_LDSDY 1
#endif
constexpr int U3_DYROW = 20, U3_DY = 32 * U3_DYROW;          // dwords: 16 windows + 16 code bytes per channel row (80 bytes), 32 rows
constexpr int U3_TWP = 35, U3_PLANE = 3 * U3_TWP, U3_HALO = 3 * U3_PLANE + 5;      // 320 floats per wave and buffer

__global__ __launch_bounds__(256) void conv3x3_wgrad_c3_unpool_kernel(
    const float* __restrict__ x, const float* __restrict__ dyp, float* __restrict__ part,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h, int total_stages, int splits, size_t slab_stride,
    const uint8_t* __restrict__ pool_idx) {
    __shared__ float halo[4 * 2 * U3_HALO];
    __shared__ float red[2 * 16 * 64 + 64];
#if CLHIP_U3_LDSDY
    __shared__ __attribute__((aligned(16))) float dyl[4 * 2 * U3_DY];
#endif
#if CLHIP_U3_BSUM_MFMA
    __shared__ float ones_s[64];
    ones_s[threadIdx.x & 63] = 1.f;          // (every wave writes the same values; read after the first stage's wave-private LDS traffic)
    __syncthreads();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wh = wave >> 1;
    const int li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits;
    const int kt = blockIdx.x / splits;
    const int k0 = kt * KT;
    const int per = total_stages / splits, extra = total_stages % splits;
    const int st_begin = split * per + min(split, extra);
    const int st_end = st_begin + per + (split < extra ? 1 : 0);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    double bsum = 0.0;

    const int ncol = C * 9;
    int col_off = 0;
    if (li < ncol) {
        const int cc = li / 9, rs = li - cc * 9;
        col_off = cc * U3_PLANE + (rs / 3) * U3_TWP + (rs % 3);
    }
    float* xs_w = halo + wave * (2 * U3_HALO);
    const int OH = H >> 1, OW = W >> 1;
    const int k = k0 + wk * 32 + li;
    [[maybe_unused]] const bool kok = k < K;
    const unsigned want = 2 * wh + kk;
    // Addresses of a stage = a per-lane constant (vector offset, computed once) + a wave-uniform part (scalar offset, SALU): no vector
    // multiplies and no per-element coordinate compares in the stage loop (they were ~65 of its ~190 VALU instructions per 16 MFMAs,
    // and VALU issue adds to the f32-MFMA time on this chip: round 6, second session).
    const int chw_p = OH * OW;
    const __amdgpu_buffer_rsrc_t r_dy = clhip_rsrc(dyp, (size_t)N * K * chw_p * 4);
    const __amdgpu_buffer_rsrc_t r_ix = clhip_rsrc(pool_idx, (size_t)N * K * chw_p);
#if CLHIP_U3_LDSDY
    // loads: lane = (channel cA = lane / 4 [+ 16], quarter q of its 64-byte run); codes: lane = (channel lane / 2, half of its 16 bytes)
    float* dy_w = dyl + wave * (2 * U3_DY);
    const int cA = lane >> 2, q4 = lane & 3, c2 = lane >> 1, h2 = lane & 1;
    const int kA = k0 + wk * 32 + cA, kB = kA + 16, k2 = k0 + wk * 32 + c2;
    const int dyA_voff = kA < K ? (kA * chw_p + 4 * q4) * 4 : CLHIP_OOB, dyB_voff = kB < K ? (kB * chw_p + 4 * q4) * 4 : CLHIP_OOB;
    const int ix8_voff = k2 < K ? k2 * chw_p + 8 * h2 : CLHIP_OOB;
    const int dstA = cA * U3_DYROW + 4 * q4, dstB = dstA + 16 * U3_DYROW, dst8 = c2 * U3_DYROW + 16 + 2 * h2;
#else
    const int dy_voff = kok ? k * chw_p * 4 : CLHIP_OOB, ix_voff = kok ? k * chw_p : CLHIP_OOB;
#endif
    // x: the descriptor starts (W + 1) elements BEFORE the tensor so that the scalar part (image, row h - 1, column w0 - 1) is never
    // negative; elements outside the image are masked by lane (vector offset CLHIP_OOB reads 0), nothing in front of x is touched
    const __amdgpu_buffer_rsrc_t r_x = clhip_rsrc(x - (W + 1), ((size_t)N * C * H * W + (size_t)(W + 1)) * 4);
    // halo element e = lane + 64 j of [3 channels][3 rows][34 columns]
    int xe_voff[5], xe_dst[5];
    unsigned edge_bits = 0;         // nibble j: element j sits in halo row 0 / row 2 / column 0 / column 33
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int e = lane + 64 * j;
        const int c = e / 102, rem = e - c * 102, row = rem / 34, col = rem - row * 34;
        xe_voff[j] = (e < 306 && c < C) ? ((c * H + row) * W + col) * 4 : CLHIP_OOB;
        edge_bits |= ((row == 0 ? 1u : 0u) | (row == 2 ? 2u : 0u) | (col == 0 ? 4u : 0u) | (col == 33 ? 8u : 0u)) << (4 * j);
        xe_dst[j] = e < 306 ? c * U3_PLANE + row * U3_TWP + col : 3 * U3_PLANE + (lane & 3);       // spare words behind the planes
    }
#if CLHIP_U3_LDSDY
    struct Stage { float4 gA, gB; float2 code8; float xr[5]; };
#else
    struct Stage { float4 g[4]; clhip_u32x4 code; float xr[5]; };
#endif

    int cur_n, cur_th, cur_tw;
    {
        const int st0 = st_begin < st_end ? st_begin : 0;
        cur_tw = st0 % tiles_w;
        const int t0 = st0 / tiles_w;
        cur_th = t0 % tiles_h;
        cur_n = t0 / tiles_h;
    }
    // (live == false: a stage past this block's range — every vector offset out of range, the loads return zeros and touch nothing)
    auto load_stage = [&](Stage& S, bool live) {
        const int n = cur_n, h0 = cur_th * 2, w0 = cur_tw * 32;
        if (++cur_tw == tiles_w) { cur_tw = 0; if (++cur_th == tiles_h) { cur_th = 0; ++cur_n; } }
        const int o = __builtin_amdgcn_readfirstlane(live ? (n * K * OH + (h0 >> 1)) * OW + (w0 >> 1) : 0);      // 16 pooled windows of every channel
#if CLHIP_U3_LDSDY
        S.gA = clhip_buf_load4(r_dy, live ? dyA_voff : CLHIP_OOB, o * 4);
        S.gB = clhip_buf_load4(r_dy, live ? dyB_voff : CLHIP_OOB, o * 4);
        S.code8 = clhip_buf_load2(r_ix, live ? ix8_voff : CLHIP_OOB, o);
#else
        const int dv = live ? dy_voff : CLHIP_OOB, iv = live ? ix_voff : CLHIP_OOB;
#pragma unroll
        for (int j = 0; j < 4; ++j) S.g[j] = clhip_buf_load4(r_dy, dv, (o + 4 * j) * 4);
        S.code = __builtin_amdgcn_raw_buffer_load_b128(r_ix, iv, o, 0);
#endif
        const int hrow = h0 + wh;
        // halo row 0 is image row hrow - 1, row 2 is hrow + 1; column 0 is w0 - 1, column 33 is w0 + 32
        const unsigned edges = (hrow == 0 ? 1u : 0u) | (hrow + 1 >= H ? 2u : 0u) | (w0 == 0 ? 4u : 0u) | (w0 + 32 >= W ? 8u : 0u);
        const unsigned bad = live ? edge_bits & (unsigned)__builtin_amdgcn_readfirstlane((int)(edges * 0x11111u)) : 0xfffffu;
        const int xo = __builtin_amdgcn_readfirstlane(live ? ((n * C * H + hrow) * W + w0) * 4 : 0);       // (+ (W + 1) - (W + 1): see r_x)
#pragma unroll
        for (int j = 0; j < 5; ++j)
            S.xr[j] = clhip_buf_load(r_x, (bad & (0xfu << (4 * j))) ? CLHIP_OOB : xe_voff[j], xo);
    };
    auto store_x = [&](const Stage& S, int buf) {
        float* d = xs_w + buf * U3_HALO;
#pragma unroll
        for (int j = 0; j < 5; ++j) d[xe_dst[j]] = S.xr[j];
#if CLHIP_U3_LDSDY
        float* dd = dy_w + buf * U3_DY;
        *reinterpret_cast<float4*>(dd + dstA) = S.gA;
        *reinterpret_cast<float4*>(dd + dstB) = S.gB;
        *reinterpret_cast<float2*>(dd + dst8) = S.code8;
#endif
    };
    auto compute = [&](const Stage& S, int buf) {
#if CLHIP_U3_LDSDY
        // this lane's channel row: 16 windows + 16 codes (LDS operations of one wave execute in order: the row is complete)
        const float* row = dy_w + buf * U3_DY + li * U3_DYROW;
        float4 Sg[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) Sg[j] = *reinterpret_cast<const float4*>(row + 4 * j);
        const clhip_u32x4 Scode = *reinterpret_cast<const clhip_u32x4*>(row + 16);
#else
        const float4* Sg = S.g;
        const clhip_u32x4 Scode = S.code;
#endif
#if CLHIP_U3_BSUM_MFMA
        const float* bp = li == ncol ? ones_s + kk : xs_w + buf * U3_HALO + col_off + kk;
#else
        const float* bp = xs_w + buf * U3_HALO + col_off + kk;
#endif
        // (the accumulator lives in AGPRs across the whole stage loop: without the pin the loop-carried copy sat in VGPRs and every
        //  trip paid 16 v_accvgpr_write + 16 v_accvgpr_read)
        asm volatile("" : "+a"(acc));
#pragma unroll
        for (int wp = 0; wp < 16; ++wp) {
            const float g = wp & 2 ? (wp & 1 ? Sg[wp >> 2].w : Sg[wp >> 2].z) : (wp & 1 ? Sg[wp >> 2].y : Sg[wp >> 2].x);
            const unsigned word = Scode[wp >> 2];
            const unsigned code = (word >> (8 * (wp & 3))) & 0xffu;
            const float a = code == want ? g : 0.f;
#if !CLHIP_U3_BSUM_MFMA
            bsum += (double)a;
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[2 * wp], acc, 0, 0, 0);
        }
        asm volatile("" : "+a"(acc));
    };

    // Stage pairs in ONE basic block (loads, LDS writes and both stages' MFMAs unconditional; the load behind the block's last stage is
    // dead: see load_stage), the odd last stage behind the loop: with branches inside the loop the loop-carried accumulator was copied
    // AGPR -> VGPR -> AGPR on every trip (16 + 16 VALU instructions per two stages).
    const int n_st = st_end - st_begin;
    if (n_st > 0) {
        Stage SA, SB;
        load_stage(SA, true);
        store_x(SA, 0);
        const int pairs = n_st >> 1;
        for (int i = 0; i < pairs; ++i) {
            load_stage(SB, true);                    // stage 2 i + 1 < n_st
            compute(SA, 0);
            store_x(SB, 1);
            load_stage(SA, 2 * i + 2 < n_st);
            compute(SB, 1);
            store_x(SA, 0);
        }
        if (n_st & 1) compute(SA, 0);
    }
    // combine the two row halves through LDS, fixed order: half 0 + half 1 (as conv3x3_wgrad_smallc_kernel)
    bsum += __shfl_xor(bsum, 32, 64);
    if (wh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wk * 16 + r) * 64 + lane] = acc[r];
        if (kk == 0) red[2048 + wk * 32 + li] = (float)bsum;
    }
    __syncthreads();
    if (wh == 0) {
        float* slab = part + (size_t)split * slab_stride;
        if (li < ncol) {
            const int cc = li / 9, rs = li - cc * 9;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = k0 + wk * 32 + mfma32_row(r, lane);
                if (kr < K) slab[((size_t)rs * K + kr) * C + cc] = acc[r] + red[(wk * 16 + r) * 64 + lane];
            }
        }
#if CLHIP_U3_BSUM_MFMA
        if (li == ncol) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kr = k0 + wk * 32 + mfma32_row(r, lane);
                if (kr < K) slab[(size_t)9 * K * C + kr] = acc[r] + red[(wk * 16 + r) * 64 + lane];
            }
        }
#else
        if (kk == 0 && kok) slab[(size_t)9 * K * C + k] = (float)(bsum + (double)red[2048 + wk * 32 + li]);
#endif
    }
}

// Fixed-order reduction of the partial slabs in ONE launch (the two tiny launches it replaces were launch-latency
// bound: 12 us per layer and pass).  Block = 64 consecutive elements x 16 split-lanes: lane j sums its contiguous
// range of splits in order (f64), the 16 partials meet in LDS and are added in order j = 0..15 — the same two-level
// association for every run, every grid size: bitwise run-to-run deterministic.
//   dw[k][c][rs] = sum_s part[s][rs][k][c] ; db[k] = sum_s part[s][9KC + k]
constexpr int RED_EL = 64, RED_J = 16, RED_V = 4;          // 64 elements x 16 split-lanes per block, 4 elements per thread
constexpr int RED_THREADS = RED_EL / RED_V * RED_J;
// (Measured and removed in round 5: 256 elements x 4 split-lanes — 1 KB of ONE slab per wave and load instruction — ran 39.1 us
// against 22.7 us for the slabs of a small_VGG9 pass, profiles/r05_wgred_wide.txt: fewer splits in flight per element.)

// One block: elements [e0, e0 + 64) of the (9KC + K)-element slab.  Thread (j, v) sums splits [j q, (j + 1) q) of the 4
// consecutive elements v (16-byte loads when the slab rows are 16-byte aligned), then lane j = 0 adds the 16 partial
// sums in order: the order per element does not depend on the vector width.
template <int EL, int J>
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                                   int K, int C, int splits, size_t e0, double (*partial)[EL], int T = 9) {
    const size_t kc = (size_t)K * C, nw = (size_t)T * kc, total = nw + K;      // T taps per (k, c): 9, or 25 for the 5x5 slabs of bswgrad5.hip
    const int v = threadIdx.x % (EL / RED_V), j = threadIdx.x / (EL / RED_V);
    const size_t e = e0 + (size_t)v * RED_V;
    const int q = (splits + J - 1) / J;
    const int s0 = j * q, s1 = min(splits, s0 + q);
    double s[RED_V] = {0.0, 0.0, 0.0, 0.0};
    if ((total & 3) == 0 && e + RED_V <= total) {
        // (four slabs in flight per thread; eight were measured SLOWER — 36.7 us against 24.5 for the slabs of a small_VGG9 pass)
#pragma unroll 4
        for (int sp = s0; sp < s1; ++sp) {
            const float4 t = *reinterpret_cast<const float4*>(part + (size_t)sp * total + e);
            s[0] += (double)t.x; s[1] += (double)t.y; s[2] += (double)t.z; s[3] += (double)t.w;
        }
    } else {
        for (int sp = s0; sp < s1; ++sp)
#pragma unroll
            for (int t = 0; t < RED_V; ++t)
                if (e + t < total) s[t] += (double)part[(size_t)sp * total + e + t];
    }
#pragma unroll
    for (int t = 0; t < RED_V; ++t) partial[j][v * RED_V + t] = s[t];
    __syncthreads();
    if (threadIdx.x < EL) {
        const int el = threadIdx.x;
        const size_t ee = e0 + el;
        if (ee < total) {
            double t = 0.0;
#pragma unroll
            for (int jj = 0; jj < J; ++jj) t += partial[jj][el];
            if (ee < nw) {
                const size_t rs = ee / kc, rem = ee - rs * kc;     // rem = k*C + c
                dw[rem * T + rs] = (float)t;
            } else if (db) {
                db[ee - nw] = (float)t;
            }
        }
    }
}

template <int EL, int J>
__global__ __launch_bounds__(RED_THREADS) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                   float* __restrict__ db, int K, int C, int splits) {
    __shared__ double partial[J][EL];
    wgrad_reduce_block<EL, J>(part, dw, db, K, C, splits, (size_t)blockIdx.x * EL, partial);
}

// The same reduction for the slabs of SEVERAL layers in one launch (the plan executor defers them to the end of the
// backward pass).  Per job the arithmetic and its order are those of wgrad_reduce_kernel.
struct RedJobs {
    clhip_wgrad_job job[CLHIP_WGRAD_JOBS_MAX];
    unsigned first[CLHIP_WGRAD_JOBS_MAX + 1];       // prefix sums of blocks
    int n;
};

template <int EL, int JL>
__global__ __launch_bounds__(RED_THREADS) void wgrad_reduce_multi_kernel(RedJobs jobs) {
    __shared__ double partial[JL][EL];
    int ji = 0;
    while (ji + 1 < jobs.n && blockIdx.x >= jobs.first[ji + 1]) ++ji;
    const clhip_wgrad_job J = jobs.job[ji];
    wgrad_reduce_block<EL, JL>(J.part, J.dw, J.db, J.K, J.C, J.splits, (size_t)(blockIdx.x - jobs.first[ji]) * EL, partial, J.taps ? J.taps : 9);
}

struct WPlan {
    int TW, TH, tiles_w, tiles_h, total_stages, splits, k_tiles, c_tiles, group, groups, ps;
    size_t slab, ws_floats;
};

WPlan make_plan(int N, int C, int K, int H, int W) {
    WPlan p;
    if (W > 16) { p.TW = 32; p.TH = 2; }
    else if (W > 8) { p.TW = 16; p.TH = 4; }
    else { p.TW = 8; p.TH = 8; }
    p.tiles_w = (W + p.TW - 1) / p.TW;
    p.tiles_h = (H + p.TH - 1) / p.TH;
    p.total_stages = p.tiles_w * p.tiles_h * N;
    const bool smallc = (C * 9 <= 32);
    p.k_tiles = (K + KT - 1) / KT;
    p.c_tiles = smallc ? 1 : (C + CT - 1) / CT;
    p.ps = 0;
    if (!smallc) {
        // few stages per 64x64 block (deep layers): 32x32 tiles with the pixels of a stage split over the waves
        const int t64 = p.k_tiles * p.c_tiles;
        const int splits64 = (256 + t64 - 1) / t64;
        if (p.total_stages / (splits64 > 0 ? splits64 : 1) < 8) {
            p.ps = 1;
            p.k_tiles = (K + 31) / 32;
            p.c_tiles = (C + 31) / 32;
        }
    }
    int tiles = p.k_tiles * p.c_tiles;
    // MFMA-bound general kernel: ~1 block per CU. HBM-bound first-layer kernel: several per CU.
    int target = smallc ? 1024 : 256;       // (512 PS blocks, two per CU, measured slower: 44 vs 39 us at 8x8)
    if (smallc) {                           // tuning switch (tools/experiments): blocks of the first-layer launch
        static const int t_env = [] { const char* e = getenv("CLHIP_WG_SMALLC_BLOCKS"); return e && e[0] ? atoi(e) : 0; }();
        if (t_env > 0) target = t_env;
    }
    int splits = (target + tiles - 1) / tiles;
    if (splits > p.total_stages) splits = p.total_stages;
    p.slab = (size_t)9 * K * C + K;
    size_t max_splits = ((size_t)64 << 20) / (p.slab * sizeof(float));
    if (max_splits < 1) max_splits = 1;
    if ((size_t)splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    p.splits = splits;
    // two-level reduction: groups of ~sqrt(splits)
    int group = 1;
    while (group * group < splits) ++group;
    p.group = group;
    p.groups = (splits + group - 1) / group;
    p.ws_floats = p.slab * (size_t)(splits + p.groups);
    return p;
}

}  // namespace

extern "C" {

#ifdef CLHIP_TRACE
int clhip_debug_set_wgrad_trace(void* p) {
    unsigned long long* q = static_cast<unsigned long long*>(p);
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wtrace), &q, sizeof(q));
}
#endif

size_t clhip_conv3x3_bwd_weight_ws(int N, int C, int K, int H, int W) {
    if (N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return 0;
    return make_plan(N, C, K, H, W).ws_floats * sizeof(float);
}

static int bwd_weight_impl(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db,
                           int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream,
                           clhip_wgrad_job* defer = nullptr) {
    if (!x || !dy || !dw || !ws || N <= 0 || C <= 0 || K <= 0 || H <= 0 || W <= 0) return CLHIP_EINVAL;
    if (unpool_idx && ((H & 1) || (W & 1))) return CLHIP_ENOTSUP;
    WPlan p = make_plan(N, C, K, H, W);
    if (ws_bytes < p.ws_floats * sizeof(float)) return CLHIP_ENOSPC;
    hipStream_t s = as_stream(stream);
    float* part = static_cast<float*>(ws);
    const bool smallc = (C * 9 <= 32);
    unsigned grid = (unsigned)(p.k_tiles * p.c_tiles * p.splits);
#define WG_ARGS x, dy, part, N, C, K, H, W, p.tiles_w, p.tiles_h, p.total_stages, p.splits
    if (CLHIP_U3 && smallc && unpool_idx && p.TW == 32 && W % 32 == 0 && aligned16(dy) && aligned16(unpool_idx) &&
        (size_t)N * K * H * W < ((size_t)1 << 31) && (size_t)N * C * H * W < ((size_t)1 << 29)) {
        hipLaunchKernelGGL(conv3x3_wgrad_c3_unpool_kernel, dim3(grid), dim3(256), 0, s, WG_ARGS, p.slab, unpool_idx);
    } else if (smallc) {
        if (p.TW == 32) hipLaunchKernelGGL((conv3x3_wgrad_smallc_kernel<32, 2>), dim3(grid), dim3(256), 0, s, WG_ARGS, p.slab, unpool_idx);
        else if (p.TW == 16) hipLaunchKernelGGL((conv3x3_wgrad_smallc_kernel<16, 4>), dim3(grid), dim3(256), 0, s, WG_ARGS, p.slab, unpool_idx);
        else hipLaunchKernelGGL((conv3x3_wgrad_smallc_kernel<8, 8>), dim3(grid), dim3(256), 0, s, WG_ARGS, p.slab, unpool_idx);
    } else {
        // 16-byte staging needs aligned rows and whole tiles along w
        const bool vec = (W % 4 == 0) && (W % p.TW == 0) && aligned16(x) && aligned16(dy);
        if (unpool_idx && (!vec || (reinterpret_cast<uintptr_t>(unpool_idx) & 1u))) return CLHIP_ENOTSUP;
#define WG_K(TW_, TH_, VEC_, PS_, UNP_) hipLaunchKernelGGL((conv3x3_wgrad_kernel<TW_, TH_, VEC_, PS_, UNP_>), dim3(grid), dim3(256), 0, s, \
                                                           WG_ARGS, p.c_tiles, p.slab, unpool_idx)
#define WG_LAUNCH1(TW_, TH_, PS_) do { if (unpool_idx) WG_K(TW_, TH_, true, PS_, true); else if (vec) WG_K(TW_, TH_, true, PS_, false); \
                                      else WG_K(TW_, TH_, false, PS_, false); } while (0)
#define WG_LAUNCH(TW_, TH_) do { if (p.ps) WG_LAUNCH1(TW_, TH_, true); else WG_LAUNCH1(TW_, TH_, false); } while (0)
        if (p.TW == 32) WG_LAUNCH(32, 2);
        else if (p.TW == 16) WG_LAUNCH(16, 4);
        else WG_LAUNCH(8, 8);
#undef WG_LAUNCH
#undef WG_LAUNCH1
#undef WG_K
    }
#undef WG_ARGS
    CLHIP_LAUNCH_CHECK();
    if (defer) {
        *defer = clhip_wgrad_job{part, dw, db, K, C, p.splits};
        return 0;
    }
    const unsigned bx = (unsigned)((p.slab + RED_EL - 1) / RED_EL);
    hipLaunchKernelGGL((wgrad_reduce_kernel<RED_EL, RED_J>), dim3(bx), dim3(RED_THREADS), 0, s, part, dw, db, K, C, p.splits);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                             int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    return bwd_weight_impl(x, dy, nullptr, dw, db, N, C, K, H, W, ws, ws_bytes, stream);
}

// dy_pool[N][K][H/2][W/2] + idx: weight gradient of a conv whose ReLU output was 2x2-max-pooled, without
// materialising the un-pooled gradient.  Supported for the first-layer kernel (C*9 <= 32); others: ENOTSUP.
int clhip_conv3x3_bwd_weight_unpool(const float* x, const float* dy_pool, const uint8_t* idx_u8, float* dw, float* db,
                                    int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!idx_u8) return CLHIP_EINVAL;
    return bwd_weight_impl(x, dy_pool, idx_u8, dw, db, N, C, K, H, W, ws, ws_bytes, stream);
}

// The two halves of clhip_conv3x3_bwd_weight for callers that defer the reduction (the plan executor reduces the
// slabs of all layers in one launch at the end of backward): slabs only, then the fixed-order reduction.
int clhip_conv3x3_bwd_weight_slabs(const float* x, const float* dy, const uint8_t* idx_u8_or_null, int N, int C, int K, int H,
                                   int W, void* ws, size_t ws_bytes, int* splits_out, void* stream) {
    if (!splits_out || !ws) return CLHIP_EINVAL;
    clhip_wgrad_job job;
    const int rc = bwd_weight_impl(x, dy, idx_u8_or_null, static_cast<float*>(ws), nullptr, N, C, K, H, W, ws, ws_bytes, stream, &job);
    if (rc == 0) *splits_out = job.splits;
    return rc;
}

int clhip_conv3x3_bwd_weight_reduce(const void* ws, float* dw, float* db, int K, int C, int splits, void* stream) {
    if (!ws || !dw || K <= 0 || C <= 0 || splits <= 0) return CLHIP_EINVAL;
    const size_t slab = (size_t)9 * K * C + K;
    const unsigned bx = (unsigned)((slab + RED_EL - 1) / RED_EL);
    hipLaunchKernelGGL((wgrad_reduce_kernel<RED_EL, RED_J>), dim3(bx), dim3(RED_THREADS), 0, as_stream(stream),
                            static_cast<const float*>(ws), dw, db, K, C, splits);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// Partial slabs only; the caller reduces several layers at once with clhip_internal_wgrad_reduce_multi (the slabs in
// ws must stay untouched until then).  unpool_idx as in clhip_conv3x3_bwd_weight_unpool (may be NULL).
int clhip_internal_conv3x3_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N,
                                         int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream, clhip_wgrad_job* job) {
    if (!job) return CLHIP_EINVAL;
    return bwd_weight_impl(x, dy, unpool_idx, dw, db, N, C, K, H, W, ws, ws_bytes, stream, job);
}

int clhip_internal_wgrad_reduce_multi(const clhip_wgrad_job* jobs, int n, hipStream_t s) {
    if (n <= 0) return 0;
    if (!jobs || n > CLHIP_WGRAD_JOBS_MAX) return CLHIP_EINVAL;
    RedJobs r;
    r.n = n;
    r.first[0] = 0;
    for (int i = 0; i < n; ++i) {
        r.job[i] = jobs[i];
        const size_t total = (size_t)(jobs[i].taps ? jobs[i].taps : 9) * jobs[i].K * jobs[i].C + jobs[i].K;
        r.first[i + 1] = r.first[i] + (unsigned)((total + RED_EL - 1) / RED_EL);
    }
    for (int i = n; i < CLHIP_WGRAD_JOBS_MAX; ++i) { r.job[i] = clhip_wgrad_job{nullptr, nullptr, nullptr, 0, 0, 0, 0}; r.first[i + 1] = r.first[n]; }
    hipLaunchKernelGGL((wgrad_reduce_multi_kernel<RED_EL, RED_J>), dim3(r.first[n]), dim3(RED_THREADS), 0, s, r);
    CLHIP_LAUNCH_CHECK();
    return 0;
}
