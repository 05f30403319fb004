// bsconv.hip — 3x3 convolution (stride 1, padding 1) forward / backward-data on the bf16 matrix cores with fp32 operands split into
// three bf16 pieces each: the convolutions of VGGSlim.forward (models/VGGSlim.py:27-40) and their autograd backward w.r.t. the
// input, same operators and fusions as wino.hip's entry points (bias + ReLU + optional 2x2 max-pool with arg-max codes on the
// forward; gradient of the POOLED output + codes as input, (mask_src > 0) on the output of backward-data).
//
// Why.  On gfx950 the f32-input MFMA runs at the f32 VECTOR rate and, measured (tools/micro/bf16_split_dot.hip,
// profiles/r05_bf16_split_dot.txt), it shares the vector lanes: ONE v_fma per MFMA in the same wave costs x1.48, a VALU-only
// sibling wave adds its full time — the "MFMA time + everything else" sum that kept the Winograd kernels at 0.45 of the pipe.
// v_mfma_f32_32x32x16_bf16 is 16x that rate and hides 4 VALU instructions per MFMA for free.  An fp32 value is EXACTLY
// a0 + a1 + a2 with three bf16 pieces (8 + 8 + 8 significand bits, round-to-nearest residuals); every bf16 x bf16 product is exact
// in fp32; the six products ai * bj with i + j <= 2 carry a * b to 2^-26 relative, the accumulation is fp32 inside the matrix core.
// Measured against fp64 on 576 .. 4608-term dot products: max / rms error 1.1 - 2.2e-7 / 2.3 - 3.3e-8 of sum|a b| — the same as
// the f32-input MFMA chain (1.2 - 2.1e-7 / 2.7 - 2.9e-8) and as a host fmaf chain; 4 - 8e-8 / 1e-8 with a0 b0 in its own accumulator.  Six bf16 MFMAs per 16-deep k-step = 2.67x the
// f32 matrix rate on the DIRECT form (no Winograd transforms: their VALU work was the problem, and their conditioning is gone too).
//
// Data flow of one block (256 threads = 4 waves as 2 x 2, each 64 pixels x 32 channels; 128 output pixels x 64 output channels;
// three blocks per CU):
//   * per 16-channel chunk the (RH + 2) x (RW + 2) halo tile of the input is loaded NCHW -> registers (dword buffer loads, lanes
//     along x; one chunk ahead), split into 3 bf16 pieces and written to LDS as [piece][k half][halo row][pitch P] x 16 bytes
//     (8 channels of one pixel, the A operand of one lane).  P = 8 mod 16 slots: the 16 lanes of every ds_read_b128 lane group
//     hit 16 different 4-bank groups for every tap (MI355X_MICROARCH.md, LDS table; measured 1.7 % conflict cycles).
//     Double-buffered, one barrier per chunk; the split / LDS writes of chunk c + 1 and the loads of chunk c + 2 are issued INSIDE
//     the wave's MFMA stream of chunk c (one basic block per chunk).
//   * weights: a lane-ordered image [n tile][chunk][tap][piece][lane] x 16 bytes (bs_weight_multi_kernel, once per pass) is read
//     straight from L2 / L1 with one coalesced 1 KB buffer_load_dwordx4 per operand, one tap ahead.  (Copying the chunk's slices to
//     LDS once per block instead — three different block structures, see the notes in front of the kernel — was slower every time.)
//   * an MFMA tile is 32 pixels x 32 channels; the 32 pixels are 8 pooling windows x 4 positions (m = 4 w + q) so that the four
//     accumulator registers r = 4 g + q of a lane are one 2x2 window: ReLU + max-pool + arg-max code are lane-local.
//   * per (tap, k-step) a wave issues 6 MFMAs per tile: a0 b2, a0 b0, a2 b0, a1 b1, a0 b1, a1 b0 — a0 b0 into an accumulator of
//     its own (added in the epilogue): a third of the error of one accumulator.
//   * outputs leave through an LDS transposition (the staging buffers are free by then) as whole rows: float4 stores of 32 .. 128
//     contiguous bytes per channel instead of 64 scattered 8-byte pieces per store instruction.
//   * the same kernel with 25 taps (KS = 5, halo 2) is AlexNet's second convolution.
//
// Algorithmic FLOPs 2 * 9 * Cin * Cout * H * W * N; issued on the matrix pipe: 6x that, against the 2.5 PFLOP/s dense bf16 peak.
#include "common.hpp"
#include "bs_weight.hpp"
#include <cstdlib>

// Timing-only ablations (tools/experiments; results wrong by design; the product is built with 0): 1 no MFMAs, 2 no weight-operand
// loads, 4 no split / LDS writes, 8 no activation loads, 16 no output stores, 32 no barriers, 64 per-lane (scattered) output stores
#ifndef BS_ABL
#define BS_ABL 0
#endif

namespace {

// ---------------------------------------------------------------------------------------------------- weight image
// img[nt][chunk][tap][piece][lane] (16 bytes each): lane l of the B operand of n tile nt holds output channel ko = 32 nt + (l & 31)
// and input channels ci = 16 chunk + 8 (l >> 5) + e, e = 0..7, of tap (r, s) — MODE 0: w[ko][ci][r][s]; MODE 1 (backward-data: the
// kernel's input channels are the layer's output channels): w[ci][ko][ks - 1 - r][ks - 1 - s].  Ko / Ci: channel counts as the KERNEL
// sees them; ks x ks taps (3 x 3, or 5 x 5: AlexNet's second convolution, models/net.py:96-125).  The body (bs_weight_block) lives in
// bs_weight.hpp: the plan executor builds these images and the Winograd U images of a pass in ONE launch (wino.hip,
// clhip_internal_weight_images); this kernel serves the single-layer entry points.
constexpr int BS_WT_JOBS = 24;
struct BsWtJobs { int n; int pad; clhip_wino_wt j[BS_WT_JOBS]; int first[BS_WT_JOBS + 1]; };

__global__ __launch_bounds__(256) void bs_weight_multi_kernel(BsWtJobs J) {
    int jb = 0;
    for (int i = 1; i < J.n; ++i) jb = ((int)blockIdx.x >= J.first[i]) ? i : jb;
    bs_weight_block(J.j[jb], (int)blockIdx.x - J.first[jb], (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------- the convolution
// Geometry of a block (256 threads = 4 waves as WAVES_M x WAVES_N, each WM x WN MFMA tiles): NI images x RH rows x RW columns of
// output pixels (= BM) x one 64-channel group of output channels.
//   RW = 32: one image, RH = BM / 32;  RW = 16: RH = BM / 16;  RW = 8: whole 8-row images, two side by side per LDS row.
//   M tiles: 2 rows x 16 columns (RW >= 16) or 4 rows x 8 columns.
template <int RW_, int RH_, int NI_, int WAVES_M_, int WM_, int WN_, int KS_ = 3>
struct BsGeo {
    static constexpr int RW = RW_, RH = RH_, NI = NI_, WAVES_M = WAVES_M_, WAVES_N = 4 / WAVES_M_, WM = WM_, WN = WN_;
    static constexpr int KS = KS_, HP = KS_ / 2, TAPS = KS_ * KS_;         // kernel size (3 or 5, stride 1, padding KS / 2)
    static constexpr int BM = RW * RH * NI;
    static_assert(BM == 32 * WAVES_M * WM, "pixels per block = M tiles of the waves");
    static_assert(32 * WAVES_N * WN == BS_BN, "64 output channels per block");
    static constexpr int MTW = RW >= 16 ? 16 : 8, MTH = 32 / MTW;          // M tile: MTH rows x MTW columns
    static_assert(RW % MTW == 0 && RH % MTH == 0, "whole M tiles");
    static constexpr int MT_PER_ROW = RW / MTW, MT_PER_IMG = MT_PER_ROW * (RH / MTH);
    static constexpr int IPR = RW == 8 ? 2 : 1;                            // images per LDS row
    static_assert(NI % IPR == 0, "image pairs");
    static constexpr int HW_ = RW + 2 * HP, HR = RH + 2 * HP;              // halo tile of one image
    static constexpr int P = (IPR * HW_ + 7) / 16 * 16 + 8;                // slots per LDS row, = 8 mod 16
    static_assert(P >= IPR * HW_ && P % 16 == 8, "pitch");
    static constexpr int ROWS = (NI / IPR) * HR;
    static constexpr int PLANE_SLOTS = (ROWS - 1) * P + IPR * HW_;         // (the last row needs no padding)
    static constexpr int PLANE_BYTES = PLANE_SLOTS * 16;
    static constexpr int BUF_BYTES = 6 * PLANE_BYTES;                      // 3 pieces x 2 k halves
    static constexpr int NHALO = NI * HR * HW_;
    static constexpr int ITEMS = 2 * NHALO;                                // (halo pixel, k half): 8 channels each
    static constexpr int ROUNDS = (ITEMS + 255) / 256;
    static_assert(ROUNDS * 256 - ITEMS <= ITEMS, "the last round wraps at most once");
    // output region of one wave (its WM M tiles): RGH rows x RGW columns of one image, TPR tiles per region row
    static constexpr int TPR = WM < MT_PER_ROW ? WM : MT_PER_ROW;
    static_assert(WM % TPR == 0 && MT_PER_ROW % TPR == 0 && MT_PER_IMG % WM == 0, "a wave's tiles form a rectangle inside one image");
    static constexpr int RGW = TPR * MTW, RGH = 32 * WM / RGW;
    static constexpr int TS = 32 * WM + 4;                                 // floats per channel in the transposition buffer (16-byte rows)
    static_assert(4 * 32 * TS * 4 <= 2 * BUF_BYTES, "transposition buffers fit the staging LDS");
};

// The product of the leading pieces a0 b0 and the five small products are summed in accumulators of their own (added once, in the
// epilogue): measured 3x less error than one accumulator, i.e. 3x less than an fp32 fmaf chain (profiles/r05_bf16_split_dot.txt), at
// the same speed (profiles/r05_bs_v2_per_layer.txt and the single-accumulator run of the same session).
//
// What was measured on the way here (layer 2 of the bench net, 64 -> 64 @ 32 x 32, N = 200; matrix floor 36 us at 2.5 PF, ~43 us at the
// clock the pipe sustains; Winograd f32 kernel 95 - 100 us in the same sessions):
//   v1  activations double-buffered in LDS and staged INSIDE the wave's MFMA stream, weight operands straight from L2 (one 1 KB
//       coalesced load per operand): 92 us with 64 x 32 wave tiles, 102 us with 64 x 64 (profiles/r05_bs_v1_per_layer.txt);
//   v2  weights through LDS once per block, single-buffered, two blocks per CU taking turns: 98 us — the two blocks run in lockstep,
//       matrix 43 + staging 47 + stores add up (profiles/r05_bs_v2_ablations.txt; matrix + LDS reads alone: 51 us);
//   v3  producer / consumer waves, everything double-buffered, one block per CU: 114 us, either role alone 80 us;
//   v4  persistent one-wave-per-SIMD blocks, one in-wave pipeline over (tile, chunk) stages: 121 us, matrix + reads 76, staging + reads 76.
// In every form the non-matrix work ADDS to the matrix time; what differs is how much of it there is per MFMA.  This form keeps v1's
// structure (no weight bytes through VALU / LDS-write at all) and halves its load on the vector-memory path with 128 x 32 wave tiles.
// The work of block b of a launch (a device function: bs_conv_kernel runs it for the whole grid, bs_conv_mixed_kernel for two geometries
// in one grid); lds: 2 * G::BUF_BYTES bytes of the block's LDS.
template <class G, int MODE, bool UNPOOL>
__device__ __forceinline__ void bs_conv_body(
    unsigned char* __restrict__ lds, const int b, const float* __restrict__ in, const clhip_u32x4* __restrict__ wimg,
    const float* __restrict__ bias, const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx, int N,
    int Cin, int Cout, int H, int W, int relu, int tiles_x, int tiles_y, int npb) {
    constexpr int RW = G::RW, RH = G::RH, NI = G::NI, WM = G::WM, WN = G::WN, P = G::P, HR = G::HR, HW_ = G::HW_;
    constexpr int ROUNDS = G::ROUNDS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kts = (Cout + BS_BN - 1) / BS_BN;
    // blocks of one pixel tile (all channel groups) follow each other on ONE XCD (block b runs on XCD b % 8): its input tile is read
    // from HBM once per XCD L2
    const int kt = (b >> 3) % kts, pb = (b / (8 * kts)) * 8 + (b & 7);
    if (pb >= npb) return;
    const int tx = pb % tiles_x, ty = (pb / tiles_x) % tiles_y, grp = pb / (tiles_x * tiles_y);
    const int n0 = grp * NI, y0 = ty * RH, x0 = tx * RW;
    const int n_chunks = Cin / BS_CK;
    const int IH = UNPOOL ? H >> 1 : H, IW = UNPOOL ? W >> 1 : W;          // the input tensor's own plane
    const int plane_in = IH * IW;

    const float* in_blk = in + (size_t)n0 * Cin * plane_in;
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_blk, (size_t)(N - n0) * Cin * plane_in * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? pool_idx + (size_t)n0 * Cin * plane_in : pool_idx,
                                                   UNPOOL ? (size_t)(N - n0) * Cin * plane_in : 0);
    const int n_nt = (Cout + 31) / 32;
    constexpr int TAPS = G::TAPS, KS = G::KS;
    const __amdgpu_buffer_rsrc_t rs_w = clhip_rsrc(wimg, (size_t)n_nt * n_chunks * 3 * TAPS * 1024);

    // ---- staging items of this thread: (halo pixel, k half) -> element offset of channel 8 h of the chunk, LDS slot
    int xoff[ROUNDS], lw[ROUNDS], pos[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // (the last round wraps around: its spare threads stage the first items a second time — same data to the same slots —
        // so that no load / LDS write of the loop sits under a branch)
        const int it_ = r * 256 + tid, it = it_ < G::ITEMS ? it_ : it_ - G::ITEMS;
        const int h = it >= G::NHALO ? 1 : 0, p = it - h * G::NHALO;
        const int ni = p / (HR * HW_), rem = p - ni * (HR * HW_), hy = rem / HW_, hx = rem - hy * HW_;
        const int gy = y0 + hy - G::HP, gx = x0 + hx - G::HP, n = n0 + ni;
        const bool ok = n < N && gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int e = UNPOOL ? (gy >> 1) * IW + (gx >> 1) : gy * IW + gx;
        xoff[r] = ok ? (ni * Cin + 8 * h) * plane_in + e : CLHIP_OOB;
        pos[r] = ((gy & 1) << 1) | (gx & 1);
        lw[r] = (h * G::PLANE_SLOTS + ((ni / G::IPR) * HR + hy) * P + (ni % G::IPR) * HW_ + hx) * 16;
    }
    float xr[ROUNDS][8];
    unsigned xi[UNPOOL ? ROUNDS : 1][8];
    auto load_chunk = [&](int c) {
        if (BS_ABL & 8) return;
        const int cb = c * BS_CK * plane_in;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int so = cb + e * plane_in;
                xr[r][e] = clhip_buf_load(rs_x, xoff[r] != CLHIP_OOB ? xoff[r] * 4 : CLHIP_OOB, so * 4);
                if constexpr (UNPOOL) xi[r][e] = clhip_buf_load_u8(rs_i, xoff[r], so);
            }
    };
    auto store_chunk = [&](int buf) {
        if (BS_ABL & 4) return;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (UNPOOL) v[e] = xi[r][e] == (unsigned)pos[r] ? xr[r][e] : 0.f;     // max_pool2d backward (+ ReLU: dead code 4)
                else v[e] = xr[r][e];
            }
            clhip_u32x4 q0, q1, q2;
            bs_split8(v, q0, q1, q2);
            unsigned char* d = lds + buf * G::BUF_BYTES + lw[r];
            *reinterpret_cast<clhip_u32x4*>(d) = q0;
            *reinterpret_cast<clhip_u32x4*>(d + 2 * G::PLANE_BYTES) = q1;
            *reinterpret_cast<clhip_u32x4*>(d + 4 * G::PLANE_BYTES) = q2;
        }
    };

    // ---- this wave's M tiles / N tiles
    const int wm = wave % G::WAVES_M, wn = wave / G::WAVES_M;
    const int m = lane & 31, kh = lane >> 5;
    const int mw = m >> 2, mq = m & 3;
    const int prow = 2 * (mw / (G::MTW / 2)) + (mq >> 1), pcol = 2 * (mw % (G::MTW / 2)) + (mq & 1);     // pixel of lane m inside its M tile
    int abase[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mt = wm * WM + i;
        const int ni = mt / G::MT_PER_IMG, rem = mt - ni * G::MT_PER_IMG, tr = rem / G::MT_PER_ROW, tc = rem - tr * G::MT_PER_ROW;
        abase[i] = (kh * G::PLANE_SLOTS + ((ni / G::IPR) * HR + tr * G::MTH + prow) * P + (ni % G::IPR) * HW_ + tc * G::MTW + pcol) * 16;
    }
    const int nt0 = kt * (BS_BN / 32) + wn * WN;
    int wvoff[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) wvoff[j] = (nt0 + j < n_nt) ? ((nt0 + j) * n_chunks * 3 * TAPS * 64 + lane) * 16 : CLHIP_OOB;
    auto load_b = [&](clhip_u32x4 (&bq)[WN][3], int c, int tap) {
        if (BS_ABL & 2) return;
        const int so = (c * 3 * TAPS + tap * 3) * 1024;
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int s = 0; s < 3; ++s) bq[j][s] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvoff[j], so + s * 1024, 0);
    };

    floatx16 acc[WM][WN], accl[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                accl[i][j][r] = 0.f;
            }

    // ---- prologue: chunk 0 in LDS, chunk 1 in flight, first weight operands in flight
    clhip_u32x4 bcur[WN][3];
    load_chunk(0);
    load_b(bcur, 0, 0);
    store_chunk(0);
    load_chunk(n_chunks > 1 ? 1 : 0);
    if (!(BS_ABL & 32)) __syncthreads();

    for (int c = 0; c < n_chunks; ++c) {
        const unsigned char* lb = lds + (c & 1) * G::BUF_BYTES;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dr = tap / KS, ds = tap - dr * KS;
            // next tap's weight operands (the first tap of the next chunk after the last one; past the end: any valid address)
            clhip_u32x4 bnext[WN][3];
            if (tap < TAPS - 1) load_b(bnext, c, tap + 1);
            else load_b(bnext, c + 1 < n_chunks ? c + 1 : c, 0);
            clhip_u32x4 a[WM][3];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    a[i][s] = *reinterpret_cast<const clhip_u32x4*>(lb + abase[i] + (dr * P + ds) * 16 + s * 2 * G::PLANE_BYTES);
            if (tap == TAPS / 3) {
                // staging of the next chunk inside this chunk's matrix stream, unconditionally (one basic block per chunk: the
                // scheduler may place these VALU / LDS / load instructions between the MFMAs): after the last chunk the spare
                // buffer takes a second copy of it, which nobody reads
                store_chunk((c + 1) & 1);
                load_chunk(c + 2 < n_chunks ? c + 2 : n_chunks - 1);
            }
            // six products per tile pair, small ones first; consecutive MFMAs go to different accumulators
#if BS_ABL & 1
#define BS_TERM(ACC, PA, PB)                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j)                         \
                asm volatile("" : "+v"(ACC[i][j]) : "v"(a[i][PA]), "v"(bcur[j][PB]));
#else
#define BS_TERM(ACC, PA, PB)                                                                                                      \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) _Pragma("unroll") for (int j = 0; j < WN; ++j)                         \
                ACC[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bs_bf16x8, a[i][PA]),                      \
                                                                   __builtin_bit_cast(bs_bf16x8, bcur[j][PB]), ACC[i][j], 0, 0, 0);
#endif
            BS_TERM(accl, 0, 2) BS_TERM(acc, 0, 0) BS_TERM(accl, 2, 0) BS_TERM(accl, 1, 1) BS_TERM(accl, 0, 1) BS_TERM(accl, 1, 0)
#undef BS_TERM
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int s = 0; s < 3; ++s) bcur[j][s] = bnext[j][s];
        }
        if (!(BS_ABL & 32)) __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += accl[i][j][r];
    const int cw = wave;
    unsigned char* const lds_free = lds;                 // (the last barrier has passed: both staging buffers are free)

    // ---- epilogue: acc[i][j][4 g + q] = (pixel = window 2 g + kh of M tile i, position q; channel 32 (nt0 + j) + (lane & 31))
    const bool pool = MODE == 0 && pool_idx != nullptr;
    const bool odd = (H | W) & 1;
    const size_t chw = (size_t)H * W;
    const int OH = H >> 1, OW = W >> 1;
    // the wave's region: RGH x RGW pixels of image n_w from (oh0, ow0)
    const int mt0 = wm * WM;
    const int ni_w = mt0 / G::MT_PER_IMG, rem_w = mt0 - ni_w * G::MT_PER_IMG, tr_w = rem_w / G::MT_PER_ROW, tc_w = rem_w - tr_w * G::MT_PER_ROW;
    const int n_w = n0 + ni_w, oh0 = y0 + tr_w * G::MTH, ow0 = x0 + tc_w * G::MTW;
    constexpr int RGW = G::RGW, RGH = G::RGH, TS = G::TS;
    // Fast path (rows of whole float4s: W % 4 == 0, pooled rows likewise): the tile goes through LDS — every lane writes its channel's
    // 2x2 windows, then reads rows back as float4s — so that a store instruction covers whole 32 .. 128-byte row pieces per channel
    // instead of 64 scattered 8-byte pieces (the scattered form took 24 us of a 102 us launch on its own, 77 us with the mask reads of
    // backward-data: profiles/r05_bs_v2_ablations.txt).  All staging buffers are free here (last barrier passed); each wave uses its own piece.
    const bool fast = !odd && (W & 3) == 0 && (!pool || (OW & 3) == 0) && !(BS_ABL & 64);
    if (fast) {
        // (output stores: buffer instructions with the output cache policy of common.hpp, descriptors at the wave's image)
        const size_t img_elems = (size_t)Cout * (pool ? (size_t)OH * OW : chw);
        const __amdgpu_buffer_rsrc_t rs_out = clhip_out_rsrc(out + (size_t)n_w * img_elems);
        const __amdgpu_buffer_rsrc_t rs_code = clhip_out_rsrc(pool ? pool_idx + (size_t)n_w * img_elems : nullptr);
        float* const T = reinterpret_cast<float*>(lds_free) + cw * (32 * TS);
        // arg-max codes [pooled row][pooled column] of a channel: behind its 8 WM pooled values, inside its own row of T
        uint8_t* const Cb = reinterpret_cast<uint8_t*>(T + 16 * WM);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int k = (nt0 + j) * 32 + m;
            const float bv = (MODE == 0 && bias != nullptr && k < Cout) ? bias[k] : 0.f;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int di = (i / G::TPR) * G::MTH, dj = (i % G::TPR) * G::MTW;                            // M tile i inside the region
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int w8 = 2 * g + kh;
                    const int rr = di + 2 * (w8 / (G::MTW / 2)), cc = dj + 2 * (w8 % (G::MTW / 2));
                    float y00 = acc[i][j][4 * g], y01 = acc[i][j][4 * g + 1], y10 = acc[i][j][4 * g + 2], y11 = acc[i][j][4 * g + 3];
                    if (MODE == 0) {
                        y00 += bv; y01 += bv; y10 += bv; y11 += bv;
                        if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                    }
                    if (pool) {
                        float mx = y00; int am = 0;
                        if (y01 > mx) { mx = y01; am = 1; }
                        if (y10 > mx) { mx = y10; am = 2; }
                        if (y11 > mx) { mx = y11; am = 3; }
                        if (relu && !(mx > 0.f)) am = CLHIP_POOL_DEAD;       // ReLU folded into the code (common.hpp)
                        T[m * TS + (rr >> 1) * (RGW / 2) + (cc >> 1)] = mx;
                        Cb[m * (TS * 4) + (rr >> 1) * (RGW / 2) + (cc >> 1)] = (uint8_t)am;
                    } else {
                        *reinterpret_cast<float2*>(T + m * TS + rr * RGW + cc) = make_float2(y00, y01);
                        *reinterpret_cast<float2*>(T + m * TS + (rr + 1) * RGW + cc) = make_float2(y10, y11);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int kb = (nt0 + j) * 32;
            if (pool) {
                constexpr int PW = RGW / 2, PH = RGH / 2, F4 = PW * PH / 4;                 // pooled region, float4s per channel
#pragma unroll
                for (int t = 0; t < (32 * F4 + 63) / 64; ++t) {
                    const int f = t * 64 + lane, oc = f / F4, r4 = f - oc * F4, pr = r4 / (PW / 4), c4 = r4 - pr * (PW / 4);
                    const int k = kb + oc, ph = (oh0 >> 1) + pr, pw = (ow0 >> 1) + 4 * c4;
                    if (f < 32 * F4 && k < Cout && n_w < N && ph < OH && pw < OW && !(BS_ABL & 16)) {
                        const float4 v = *reinterpret_cast<const float4*>(T + oc * TS + pr * PW + 4 * c4);
                        clhip_buf_store4(v, rs_out, ((k * OH + ph) * OW + pw) * 4, 0);
                    }
                }
                // codes: one (channel, pooled row) per lane and step, PW bytes each
#pragma unroll
                for (int t = 0; t < (32 * PH + 63) / 64; ++t) {
                    const int f = t * 64 + lane, oc = f / PH, pr = f - oc * PH;
                    const int k = kb + oc, ph = (oh0 >> 1) + pr, pw = ow0 >> 1;
                    if (f < 32 * PH && k < Cout && n_w < N && ph < OH && !(BS_ABL & 16)) {
                        const uint8_t* src = Cb + oc * (TS * 4) + pr * PW;
                        uint8_t* dst = pool_idx + ((size_t)n_w * Cout + k) * OH * OW + (size_t)ph * OW + pw;
                        if (pw + PW <= OW) {
                            const int co = (k * OH + ph) * OW + pw;
                            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                            if constexpr (PW == 16) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const clhip_u32x4*>(src), rs_code, co, 0, 0);
                            else if constexpr (PW == 8) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2*>(src), rs_code, co, 0, 0);
                            else __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned*>(src), rs_code, co, 0, 0);
                        } else {
                            for (int q = 0; q < PW && pw + q < OW; ++q) dst[q] = src[q];
                        }
                    }
                }
            } else {
                constexpr int F4 = RGW * RGH / 4;                                           // float4s per channel (16 WM)
#pragma unroll
                for (int t = 0; t < 32 * F4 / 64; ++t) {
                    const int f = t * 64 + lane, oc = f / F4, r4 = f - oc * F4, row = r4 / (RGW / 4), c4 = r4 - row * (RGW / 4);
                    const int k = kb + oc, oh = oh0 + row, ow = ow0 + 4 * c4;
                    if (k < Cout && n_w < N && oh < H && ow < W && !(BS_ABL & 16)) {
                        float4 v = *reinterpret_cast<const float4*>(T + oc * TS + row * RGW + 4 * c4);
                        const size_t o = ((size_t)n_w * Cout + k) * chw + (size_t)oh * W + ow;
                        if (MODE == 1 && mask_src) {
                            const float4 ms = *reinterpret_cast<const float4*>(mask_src + o);
                            v.x = ms.x > 0.f ? v.x : 0.f; v.y = ms.y > 0.f ? v.y : 0.f; v.z = ms.z > 0.f ? v.z : 0.f; v.w = ms.w > 0.f ? v.w : 0.f;
                        }
                        clhip_buf_store4(v, rs_out, ((k * H + oh) * W + ow) * 4, 0);
                    }
                }
            }
            if (j + 1 < WN) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
        }
    } else {
    // general path (odd maps, rows that are not whole float4s): every lane stores its own 2x2 windows
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int k = (nt0 + j) * 32 + m;
        const bool kok = k < Cout;
        const float bv = (MODE == 0 && bias != nullptr && kok) ? bias[k] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int mt = wm * WM + i;
            const int ni = mt / G::MT_PER_IMG, rem = mt - ni * G::MT_PER_IMG, tr = rem / G::MT_PER_ROW, tc = rem - tr * G::MT_PER_ROW;
            const int n = n0 + ni;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int w8 = 2 * g + kh;                                               // window of the M tile
                const int oh = y0 + tr * G::MTH + 2 * (w8 / (G::MTW / 2)), ow = x0 + tc * G::MTW + 2 * (w8 % (G::MTW / 2));
                const bool ok = kok && n < N && oh < H && ow < W;
                float y00 = acc[i][j][4 * g], y01 = acc[i][j][4 * g + 1], y10 = acc[i][j][4 * g + 2], y11 = acc[i][j][4 * g + 3];
                if (MODE == 0) {
                    y00 += bv; y01 += bv; y10 += bv; y11 += bv;
                    if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
                    if (pool) {
                        float mx = y00; int am = 0;
                        if (y01 > mx) { mx = y01; am = 1; }
                        if (y10 > mx) { mx = y10; am = 2; }
                        if (y11 > mx) { mx = y11; am = 3; }
                        if (relu && !(mx > 0.f)) am = CLHIP_POOL_DEAD;       // ReLU folded into the code (common.hpp)
                        if (ok && (!(BS_ABL & 16) || mx == 12345.678f)) {
                            const size_t o = ((size_t)n * Cout + k) * OH * OW + (size_t)(oh >> 1) * OW + (ow >> 1);
                            out[o] = mx;
                            pool_idx[o] = (uint8_t)am;
                        }
                        continue;
                    }
                }
                if (!ok || ((BS_ABL & 16) && y00 != 12345.678f)) continue;
                const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
                if (odd) {
                    const bool row1 = oh + 1 < H, col1 = ow + 1 < W;
                    if (MODE == 1 && mask_src) {
                        y00 = mask_src[o] > 0.f ? y00 : 0.f;
                        if (col1) y01 = mask_src[o + 1] > 0.f ? y01 : 0.f;
                        if (row1) y10 = mask_src[o + W] > 0.f ? y10 : 0.f;
                        if (row1 && col1) y11 = mask_src[o + W + 1] > 0.f ? y11 : 0.f;
                    }
                    out[o] = y00;
                    if (col1) out[o + 1] = y01;
                    if (row1) out[o + W] = y10;
                    if (row1 && col1) out[o + W + 1] = y11;
                } else {
                    if (MODE == 1 && mask_src) {
                        const float2 m0 = *reinterpret_cast<const float2*>(mask_src + o), m1 = *reinterpret_cast<const float2*>(mask_src + o + W);
                        y00 = m0.x > 0.f ? y00 : 0.f; y01 = m0.y > 0.f ? y01 : 0.f;
                        y10 = m1.x > 0.f ? y10 : 0.f; y11 = m1.y > 0.f ? y11 : 0.f;
                    }
                    *reinterpret_cast<float2*>(out + o) = make_float2(y00, y01);
                    *reinterpret_cast<float2*>(out + o + W) = make_float2(y10, y11);
                }
            }
        }
    }
    }
}

// (three blocks per CU — 45 KB of LDS each — where the registers allow: every forward / plain backward-data instance at <= 162, the
// un-pooling instance of the 32-wide geometry at 168 without spills; its narrower geometries would spill 11 - 12 registers)
template <class G, bool UNPOOL>
constexpr int bs_blocks_per_cu() {
    // (the 25-tap instances of the narrow geometries would spill 56 - 64 bytes at three blocks per CU)
    return (3 * 2 * G::BUF_BYTES <= 160 * 1024 && (G::RW == 32 || (!UNPOOL && G::KS == 3))) ? 3 : 2;
}

template <class G, int MODE, bool UNPOOL>
__global__ __launch_bounds__(256, (bs_blocks_per_cu<G, UNPOOL>())) void bs_conv_kernel(
    const float* __restrict__ in, const clhip_u32x4* __restrict__ wimg, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx, int N, int Cin, int Cout, int H,
    int W, int relu, int tiles_x, int tiles_y, int npb) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * G::BUF_BYTES];
    bs_conv_body<G, MODE, UNPOOL>(lds, (int)blockIdx.x, in, wimg, bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, tiles_x, tiles_y, npb);
}

// Two geometries in ONE grid.  1 600 equal blocks on 3 x 256 slots leave the last 64 of them a round of their own: measured on layer 2 of
// the bench model (64 -> 64 @32x32; tools/experiments/r06_m.sh), N = 192 (two whole rounds) 75.0 us, N = 200 83.2 — the last 8 images cost
// 8.2 us instead of 3.1.  So the images that fill whole rounds of the launch go out as GA tiles, the rest as the smaller GB tiles
// (twice as many blocks of half the length behind them): blocks [0, blocks_a) run GA on images [0, NA), the others GB on [NA, N).
// An output pixel's sum does not depend on the tile it falls in (same chunk / tap / product order): results are bitwise those of
// the one-geometry launch (tests/test_gpu_bs.py::test_two_geometry_launch_is_bitwise).
template <class GA, class GB, int MODE, bool UNPOOL>
__global__ __launch_bounds__(256, (bs_blocks_per_cu<GA, UNPOOL>() < bs_blocks_per_cu<GB, UNPOOL>() ? bs_blocks_per_cu<GA, UNPOOL>()
                                                                                                  : bs_blocks_per_cu<GB, UNPOOL>()))
void bs_conv_mixed_kernel(const float* __restrict__ in, const clhip_u32x4* __restrict__ wimg, const float* __restrict__ bias,
                          const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx, int N, int Cin,
                          int Cout, int H, int W, int relu, int blocks_a, int NA, int tiles_xa, int tiles_ya, int npb_a, int tiles_xb,
                          int tiles_yb, int npb_b) {
    constexpr int LB = 2 * (GA::BUF_BYTES > GB::BUF_BYTES ? GA::BUF_BYTES : GB::BUF_BYTES);
    __shared__ __attribute__((aligned(16))) unsigned char lds[LB];
    const int b = (int)blockIdx.x;
    if (b < blocks_a) {
        bs_conv_body<GA, MODE, UNPOOL>(lds, b, in, wimg, bias, mask_src, out, pool_idx, NA, Cin, Cout, H, W, relu, tiles_xa, tiles_ya, npb_a);
    } else {
        // the tensors of images NA .. N - 1: the input ([Cin] planes of H x W, or of the pooled size when un-pooling, with its arg-max
        // bytes), the output ([Cout] planes; pooled + arg-max bytes when the forward pools), the ReLU mask source of backward-data
        const size_t in_img = (size_t)Cin * (UNPOOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W);
        const bool pool = MODE == 0 && pool_idx != nullptr;
        const size_t out_img = (size_t)Cout * (pool ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W);
        uint8_t* idx_b = pool_idx ? pool_idx + (size_t)NA * (UNPOOL ? in_img : out_img) : nullptr;
        bs_conv_body<GB, MODE, UNPOOL>(lds, b - blocks_a, in + (size_t)NA * in_img, wimg, bias,
                                       mask_src ? mask_src + (size_t)NA * Cout * H * W : nullptr, out + (size_t)NA * out_img, idx_b, N - NA,
                                       Cin, Cout, H, W, relu, tiles_xb, tiles_yb, npb_b);
    }
}

// (A first-layer form of this scheme — 3 -> K, K dimension 27 -> 32, the A operand gathered from a raw fp32 halo tile in LDS and split
// in registers, weights split once per persistent wave — was built, passed the parity suite and ran 50.9 us against 42 - 45 us for
// conv3x3_c3w64_relu_pool_kernel on the f32 pipe (profiles/r05_j_conv_layers_small_with_c3.txt): 27 scalar LDS gathers + 88 split
// instructions per 24 MFMAs.  Removed; the first layer stays on the f32 MFMA kernels of conv3x3.hip.)

static int bs_env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && e[0] ? atoi(e) : dflt;
}

template <class G, int MODE, bool UNPOOL>
int bs_launch_geo(const float* in, const clhip_u32x4* wimg, const float* bias, const float* mask_src, float* out, uint8_t* pool_idx,
                  int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    const int tiles_x = (W + G::RW - 1) / G::RW, tiles_y = (H + G::RH - 1) / G::RH, groups = (N + G::NI - 1) / G::NI;
    const long long npb = (long long)tiles_x * tiles_y * groups;
    const int kts = (Cout + BS_BN - 1) / BS_BN;
    const long long blocks = (npb + 7) / 8 * 8 * kts;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;
    hipLaunchKernelGGL((bs_conv_kernel<G, MODE, UNPOOL>), dim3((unsigned)blocks), dim3(256), 0, s, in, wimg, bias, mask_src, out,
                       pool_idx, N, Cin, Cout, H, W, relu, tiles_x, tiles_y, (int)npb);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// GA = 128-pixel tiles of the 32-wide geometry, GB = its 64-pixel tiles; taken when the launch has whole rounds of GA blocks plus a
// remainder of at most half a round (CLHIP_BS_MIXED=0: never)
template <int MODE, bool UNPOOL>
int bs_launch_mixed32(const float* in, const clhip_u32x4* wimg, const float* bias, const float* mask_src, float* out, uint8_t* pool_idx,
                      int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s, bool* taken) {
    using GA = BsGeo<32, 4, 1, 2, 2, 1>;
    using GB = BsGeo<32, 2, 1, 2, 1, 1>;
    *taken = false;
    static const int on = bs_env_int("CLHIP_BS_MIXED", 1);
    if (!on) return 0;
    const int kts = (Cout + BS_BN - 1) / BS_BN;
    const long long per_img_a = (long long)((W + GA::RW - 1) / GA::RW) * ((H + GA::RH - 1) / GA::RH);
    const long long slots = 256LL * bs_blocks_per_cu<GA, UNPOOL>();
    const long long total = per_img_a * N * kts, rounds = total / slots, rem = total - rounds * slots;
    if (rounds < 1 || rem == 0 || 2 * rem > slots) return 0;
    const long long na = rounds * slots / (per_img_a * kts);            // images that fill the whole rounds
    if (na < 1 || na >= N || na % 8) return 0;                             // (whole groups of 8 pixel tiles: the XCD order of the blocks)
    const int NA = (int)na;
    const int txa = (W + GA::RW - 1) / GA::RW, tya = (H + GA::RH - 1) / GA::RH, txb = (W + GB::RW - 1) / GB::RW, tyb = (H + GB::RH - 1) / GB::RH;
    const long long npb_a = (long long)txa * tya * NA, npb_b = (long long)txb * tyb * (N - NA);
    const long long blocks_a = (npb_a + 7) / 8 * 8 * kts, blocks_b = (npb_b + 7) / 8 * 8 * kts;
    if (blocks_a + blocks_b > 0x7fffffffLL) return 0;
    hipLaunchKernelGGL((bs_conv_mixed_kernel<GA, GB, MODE, UNPOOL>), dim3((unsigned)(blocks_a + blocks_b)), dim3(256), 0, s, in, wimg, bias,
                       mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, (int)blocks_a, NA, txa, tya, (int)npb_a, txb, tyb, (int)npb_b);
    CLHIP_LAUNCH_CHECK();
    *taken = true;
    return 0;
}

template <int MODE, bool UNPOOL>
int bs_launch(const float* in, const clhip_u32x4* wimg, const float* bias, const float* mask_src, float* out, uint8_t* pool_idx,
              int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    if (W > 16 && !((H | W) & 1)) {
        bool taken = false;
        const int rc = bs_launch_mixed32<MODE, UNPOOL>(in, wimg, bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, s, &taken);
        if (rc || taken) return rc;
    }
    // 128-pixel blocks of 64 x 32 wave tiles everywhere.  Measured and dropped (profiles/r05_bs_v5_per_layer_cfg*.txt, layer 2 forward /
    // backward-data at N = 200): 256-pixel blocks of 128 x 32 wave tiles (half the weight-operand bytes per MFMA, no room for the
    // separate accumulators at two blocks per CU) 92 / 90 us, of 64 x 64 wave tiles 95 / 96 us, against 83 / 92 us for this shape.
#define BS_GO(...) return bs_launch_geo<BsGeo<__VA_ARGS__>, MODE, UNPOOL>(in, wimg, bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, s)
    if (W > 16) BS_GO(32, 4, 1, 2, 2, 1);
    if (W > 8) BS_GO(16, 8, 1, 2, 2, 1);
    BS_GO(8, 8, 2, 2, 2, 1);
#undef BS_GO
}

// 5 x 5 taps (stride 1, padding 2; no fused pooling: AlexNet pools 3 x 3 / 2)
template <int MODE>
int bs_launch5(const float* in, const clhip_u32x4* wimg, const float* bias, const float* mask_src, float* out, int N, int Cin, int Cout,
               int H, int W, int relu, hipStream_t s) {
    if (W > 16) return bs_launch_geo<BsGeo<32, 4, 1, 2, 2, 1, 5>, MODE, false>(in, wimg, bias, mask_src, out, nullptr, N, Cin, Cout, H, W, relu, s);
    return bs_launch_geo<BsGeo<16, 8, 1, 2, 2, 1, 5>, MODE, false>(in, wimg, bias, mask_src, out, nullptr, N, Cin, Cout, H, W, relu, s);
}

}  // namespace

// shapes this path takes: whole 32-channel k pairs on the input side, whole 64-channel groups on the output side; any H, W >= 4
// (tiles past the edge stage zeros and store nothing); fused pooling / un-pooling only on even maps
bool clhip_internal_bs_ok(int Cin, int Cout, int H, int W) {
    return Cin >= 32 && Cin % 32 == 0 && Cout % 64 == 0 && H >= 4 && W >= 4;
}

// ... and where the plan executor prefers it to the Winograd f32 kernels (CLHIP_BS=0: nowhere; CLHIP_BS=2: wherever it can run).
// Every block re-reads the weight operands of its 64 output channels (221 KB per 64 input channels) for its 128 pixels, so the path
// pays where a launch has many pixels per weight: measured at N = 200 (profiles/r05_bs_v5_per_layer_cfg0.txt, us, Winograd / this):
// 64 -> 64 @32x32 forward 100 / 83, backward-data 102 / 92; 64 -> 128 @32x32 158 / 144, 163 / 156; at 16 x 16 within 1 - 2 us either
// way; at 8 x 8 and from 256 channels on backward-data is 5 - 20 % slower.
// At 224 x 224 inputs (N = 50, profiles/r05_k_bs_shapes.txt): 256 -> 512 and 512 -> 512 @28x28 513 / 441 and 985 / 856 forward, 554 / 475
// and 1007 / 899 backward-data; @56x56 within 2 %; 64 -> 128 @112x112 476 / 487 (3.5 tiles of 32 columns per row).
bool clhip_internal_bs_preferred(int Cin, int Cout, int H, int W) {
    static const int mode = bs_env_int("CLHIP_BS", 1);
    if (mode == 0 || !clhip_internal_bs_ok(Cin, Cout, H, W)) return false;
    if (mode == 2) return true;
    const long long px = (long long)H * W;
    return (px >= 1024 && px <= 4096 && Cin <= 128 && Cout <= 128) || (px >= 512 && px < 1024 && Cin >= 256 && Cout >= 256);
}
// 5 x 5 layers (AlexNet's second convolution: 64 -> 192 on 27 x 27): against the f32 LDS-halo kernel of convkk.hip
bool clhip_internal_bs5_preferred(int Cin, int Cout, int H, int W) {
    static const int mode = bs_env_int("CLHIP_BS", 1);
    return mode != 0 && clhip_internal_bs_ok(Cin, Cout, H, W) && W > 8;
}

size_t clhip_internal_bs_ws(int Cin, int Cout) {
    return (size_t)((Cout + 31) / 32) * ((Cin + BS_CK - 1) / BS_CK) * 27 * 1024;
}
size_t clhip_internal_bs5_ws(int Cin, int Cout) {
    return (size_t)((Cout + 31) / 32) * ((Cin + BS_CK - 1) / BS_CK) * 75 * 1024;
}

int clhip_internal_bs_weights(const clhip_wino_wt* jobs, int n, hipStream_t s) {
    if (n <= 0) return 0;
    if (!jobs) return CLHIP_EINVAL;
    for (int base = 0; base < n; base += BS_WT_JOBS) {
        BsWtJobs J;
        J.n = n - base < BS_WT_JOBS ? n - base : BS_WT_JOBS;
        J.pad = 0;
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) {
            const clhip_wino_wt& q = jobs[base + i];
            if (!q.w || !q.U || q.Ko <= 0 || q.Ci <= 0) return CLHIP_EINVAL;
            J.j[i] = q;
            J.first[i] = blocks;
            blocks += bs_weight_blocks(q);
        }
        J.first[J.n] = blocks;
        hipLaunchKernelGGL(bs_weight_multi_kernel, dim3(blocks), dim3(256), 0, s, J);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

// forward (mode 0) / backward-data (mode 1) on a weight image made by clhip_internal_bs_weights; arguments as
// clhip_internal_wino_conv_u (backward-data: Cin = the layer's OUT channels, Cout = its IN channels)
int clhip_internal_bs_conv_u(int mode, const float* in, const void* wimg, const float* bias, const float* mask_src, float* out,
                             uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    if (!in || !wimg || !out || N <= 0 || !clhip_internal_bs_ok(Cin, Cout, H, W)) return CLHIP_EINVAL;
    if (((H | W) & 1) && (pool_idx || unpool)) return CLHIP_ENOTSUP;
    const clhip_u32x4* img = static_cast<const clhip_u32x4*>(wimg);
    if (mode == 0) return bs_launch<0, false>(in, img, bias, nullptr, out, pool_idx, N, Cin, Cout, H, W, relu, s);
    if (unpool) return bs_launch<1, true>(in, img, nullptr, mask_src, out, pool_idx, N, Cin, Cout, H, W, 0, s);
    return bs_launch<1, false>(in, img, nullptr, mask_src, out, nullptr, N, Cin, Cout, H, W, 0, s);
}

// the same for 5 x 5 taps (image made by clhip_internal_bs_weights from a job with pad = 5)
int clhip_internal_bs5_conv_u(int mode, const float* in, const void* wimg, const float* bias, const float* mask_src, float* out, int N,
                              int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    if (!in || !wimg || !out || N <= 0 || !clhip_internal_bs_ok(Cin, Cout, H, W) || W <= 8) return CLHIP_EINVAL;
    const clhip_u32x4* img = static_cast<const clhip_u32x4*>(wimg);
    if (mode == 0) return bs_launch5<0>(in, img, bias, nullptr, out, N, Cin, Cout, H, W, relu, s);
    return bs_launch5<1>(in, img, nullptr, mask_src, out, N, Cin, Cout, H, W, 0, s);
}

extern "C" {

size_t clhip_conv5x5_bs_ws(int C, int K) {
    const size_t a = clhip_internal_bs5_ws(C, K), b = clhip_internal_bs5_ws(K, C);
    return a > b ? a : b;
}

int clhip_conv5x5_bs_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int K, int H, int W, int relu, void* ws,
                         size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_bs_ok(C, K, H, W) || W <= 8) return CLHIP_ENOTSUP;
    if (!x || !w || !y || !ws || ws_bytes < clhip_internal_bs5_ws(C, K)) return CLHIP_EINVAL;
    const clhip_wino_wt job{w, static_cast<float*>(ws), K, C, 0, 5};
    const int rc = clhip_internal_bs_weights(&job, 1, as_stream(stream));
    if (rc) return rc;
    return clhip_internal_bs5_conv_u(0, x, ws, b, nullptr, y, N, C, K, H, W, relu, as_stream(stream));
}

int clhip_conv5x5_bs_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int K, int H, int W,
                              void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_bs_ok(K, C, H, W) || W <= 8) return CLHIP_ENOTSUP;
    if (!dy || !w || !dx || !ws || ws_bytes < clhip_internal_bs5_ws(K, C)) return CLHIP_EINVAL;
    const clhip_wino_wt job{w, static_cast<float*>(ws), C, K, 1, 5};
    const int rc = clhip_internal_bs_weights(&job, 1, as_stream(stream));
    if (rc) return rc;
    return clhip_internal_bs5_conv_u(1, dy, ws, nullptr, relu_src, dx, N, K, C, H, W, 0, as_stream(stream));
}

size_t clhip_conv3x3_bs_ws(int C, int K) {
    const size_t a = clhip_internal_bs_ws(C, K), b = clhip_internal_bs_ws(K, C);
    return a > b ? a : b;
}

int clhip_conv3x3_bs_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx_u8_or_null, int N, int C, int K, int H,
                         int W, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_bs_ok(C, K, H, W)) return CLHIP_ENOTSUP;
    if (!x || !w || !y || !ws || ws_bytes < clhip_internal_bs_ws(C, K)) return CLHIP_EINVAL;
    const clhip_wino_wt job{w, static_cast<float*>(ws), K, C, 0, 0};
    const int rc = clhip_internal_bs_weights(&job, 1, as_stream(stream));
    if (rc) return rc;
    return clhip_internal_bs_conv_u(0, x, ws, b, nullptr, y, idx_u8_or_null, 0, N, C, K, H, W, relu, as_stream(stream));
}

int clhip_conv3x3_bs_bwd_data(const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src, float* dx, int N,
                              int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_bs_ok(K, C, H, W)) return CLHIP_ENOTSUP;
    if (!dy || !w || !dx || !ws || ws_bytes < clhip_internal_bs_ws(K, C)) return CLHIP_EINVAL;
    const clhip_wino_wt job{w, static_cast<float*>(ws), C, K, 1, 0};
    const int rc = clhip_internal_bs_weights(&job, 1, as_stream(stream));
    if (rc) return rc;
    return clhip_internal_bs_conv_u(1, dy, ws, nullptr, relu_src, dx, const_cast<uint8_t*>(idx_u8_or_null), idx_u8_or_null != nullptr,
                                    N, K, C, H, W, 0, as_stream(stream));
}

}  // extern "C"
