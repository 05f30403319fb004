// 3x3 pad-1 convolution by Winograd F(2x2, 3x3) on v_mfma_f32_32x32x2_f32  (forward and backward-data).
//
// The direct kernels of conv3x3.hip spend 9 MFMAs per (32 out-channels x 32 pixels x 2 in-channels); they are bound by
// the matrix pipe (0.7-0.8 of the fp32-MFMA peak on the large layers), so the only way to make those layers much faster is
// to issue fewer MFMAs.  F(2x2, 3x3) computes a 2x2 output tile from a 4x4 input tile with 16 multiplies instead of 36:
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A            (Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks")
// Summed over input channels the element-wise product becomes 16 independent GEMMs  M_f[k][tile] = sum_c U_f[k][c] V_f[c][tile],
// i.e. 16 MFMAs per (32 out-channels x 32 TILES = 128 pixels x 2 in-channels): 2.25x fewer matrix instructions per pixel.
// Everything else is fused into the one kernel (nothing but x, the pre-transformed weights and y touches HBM):
//   * U = G g G^T is formed once per call by a small kernel into [k-tile][chunk][c][k][f] (the LDS image of a chunk, so
//     that staging a chunk's weights is a straight 40 KB copy and a lane reads its 16 frequencies with 4 ds_read_b128),
//   * the input transform B^T d B runs in registers: a lane owns one tile of one channel (MFMA B operand: lane = tile,
//     k-pair half = channel), reads its 4x4 window from the LDS halo plane (8 ds_read_b64) and forms the 16 frequency
//     values with 32 adds — next to 16 MFMAs (1024 matrix cycles) that is ~12 % extra issue time,
//   * the output transform A^T M A runs in registers too: accumulator register r of the 16 frequency accumulators of a
//     lane is (out-channel r, this lane's tile) in every one of them, so the 2x2 outputs come from 24 adds per register,
//     and the 2x2 max-pool of VGGSlim.py:32 is a maximum over four values the lane already holds (no shuffles).
// fp32 throughout; transform constants are 0, +-1, +-0.5 (exact), so the result differs from a direct fp32 convolution by
// rounding only (measured ~1e-6 of the output scale; north_star: 1e-3).
//
// Block = 4 waves = (2 halves of 64 out-channels) x (2 groups of 32 tiles); a wave holds 16 x 16 = 256 accumulator
// registers (one wave per SIMD, like the weight-gradient kernel).  Tiles of a block: TCB x TRB tiles of NIMG images.
//
// Kernels of this file, in the order they were built (all share the U layout and the weight-transform launch):
//   wino_weight_kernel / wino_weight_multi_kernel   U = G g G^T (one layer / every Winograd layer of a pass in one launch)
//   wino_conv_kernel        forward / backward-data, 32-tile waves on 32x32x2 MFMAs, slot-pinned pipeline — the A/B reference
//                           kept for the shapes the 16-tile kernel below does not take (odd maps wider than 8 tiles, maps narrower than 16 other than 8 x 8)
//   wino_wgrad_kernel       weight gradient, 64 x 64 (k, c) tiles, 256 accumulators, slot-pinned — layers with >= 8 stages per block
//   wino_conv16_kernel      forward / backward-data of 8 x 8 maps with few units: one image per wave on 16x16x4 MFMAs
//   wino_conv16g_kernel     forward / backward-data, 16-tile waves on 16x16x4 MFMAs, two blocks per CU — the DEFAULT path
//                           (even maps, 8 x 8 image pairs, odd maps 9..16 wide through row-packed tile rows)
//   wino_wgrad_ps_kernel    weight gradient of layers with few stages per block: 32 x 32 tiles, pixel split, two blocks per CU
#include "common.hpp"
#include "bs_weight.hpp"
#include <cstdlib>

namespace {

#ifndef CLHIP_W16G_ABL
#define CLHIP_W16G_ABL 0     // TIMING-ONLY ablations of wino_conv16g_kernel's loop (wrong results; tools/gpu_r04_k.sh): bit 0 no input
#endif                       // transform, 1 A operand read once, 2 no barrier, 3 no LDS stores, 4 no global loads, 5 no MFMAs
#ifndef CLHIP_W16G_PRIO
#define CLHIP_W16G_PRIO 2    // MFMA burst of a chunk fenced off and run at raised wave priority, see compute() in wino_conv16g_kernel
#endif                       // (0: off — the A/B reference; 1: priority ramps up through the burst; 3: raised during staging instead)
#ifndef CLHIP_W16_ADIRECT
#define CLHIP_W16_ADIRECT 1  // wino_conv16_kernel (8 x 8 maps with few units): its A operands from the lane-ordered image too, except the
#endif                       // instance that would spill (un-pooling, 32-channel waves).  0: through LDS (the A/B reference;
                             // profiles/r04_w16_adirect.txt: 21.1 -> 19.8, 24.9 -> 20.9, 32.6 -> 30.4 us on small_VGG9's 8 x 8 launches)
#ifndef CLHIP_W16G_PRIO_UNPOOL
#define CLHIP_W16G_PRIO_UNPOOL 0   // the same switch for the instances that rebuild the un-pooled gradient while staging (measured slower with 2)
#endif
#ifndef CLHIP_WGPS_FENCE
#ifndef CLHIP_WGPS_PITCH
#define CLHIP_WGPS_PITCH 0   // wino_wgrad_ps_kernel: 1 = LDS row pitches of 4 mod 64 floats (no bank conflict on the operand reads).  Measured
#endif                       // in round 6 (profiles/r06_j_wgps_pitch_ab.txt): 99.8 / 37.5 / 39.2 us against 97.7 / 36.7 / 38.7 with the old pitches —
                             // the conflicts (0.38 of the LDS cycles) are not what the kernel waits for; not adopted.
#define CLHIP_WGPS_FENCE 1   // wino_wgrad_ps_kernel: MFMAs of a step fenced off behind both dy transforms (0: the A/B reference)
#endif
#ifndef CLHIP_W16G_BURST
#define CLHIP_W16G_BURST 0   // 1: no VALU between the MFMAs of a chunk (A/B experiment, see compute() in wino_conv16g_kernel)
#endif
#ifndef CLHIP_W16G_PF
#define CLHIP_W16G_PF 2      // staging pipeline of wino_conv16g_kernel, see there
#endif
#ifndef CLHIP_W16G_ADIRECT
#define CLHIP_W16G_ADIRECT 1 // 1: wino_conv16g_kernel loads its A operands (transformed weights) from L2 straight into registers, from a
#endif                       // second, lane-ordered image of U behind the LDS image; LDS then holds the halo planes only (see there).
                             // 0: through LDS (the A/B reference; profiles/r04_w16g_adirect.txt: conv time of a pass -1.4 / -2.4 / -3.6 %
                             // on small / base / wide_VGG9, the instances that un-pool while staging -7 .. -13 %)

constexpr int WKT = 64;      // out channels per block
constexpr int WCK = 8;       // in channels per chunk
constexpr int WFP = 20;      // floats per (channel, out-channel) in the U tile: 16 frequencies + 4 pad — an 80-byte stride makes
                             // the 16-byte reads of 16 consecutive lanes land on 64 distinct LDS banks
constexpr int W_FLOATS = WCK * WKT * WFP;         // 10240 floats = 40 KB: U tile of one chunk [c][k][f]
// The lane-ordered image (CLHIP_W16G_ADIRECT): per (k-tile, 4-channel chunk) 16 pieces of 1 KB,
//   [out-channel half wk][row tile r][frequency quad fq][lane = 16 * (channel of the quad) + (out channel & 15)][4 frequencies]
// = what ONE buffer_load_dwordx4 of a wave of wino_conv16g_kernel wants as its A operands of 4 MFMAs: contiguous, whole lines.
constexpr int WD_FLOATS = 4 * WKT * 16;           // 4096 floats = 16 KB per (k-tile, 4-channel chunk)
#if CLHIP_W16G_ADIRECT
constexpr int WU_FLOATS = W_FLOATS + 2 * WD_FLOATS;   // both images of an 8-channel chunk
#else
constexpr int WU_FLOATS = W_FLOATS;
#endif

// the four float4 of one (channel, out-channel) pair -> the LDS image (and the lane-ordered image behind it)
__device__ __forceinline__ void wino_u_store(float* __restrict__ U, int kts, int n_chunks, int kt, int chunk, int c_l, int k_l,
                                             const float4 (&u)[4]) {
    float4* dst = reinterpret_cast<float4*>(U + (((size_t)(kt * n_chunks + chunk) * WCK + c_l) * WKT + k_l) * WFP);
#pragma unroll
    for (int a = 0; a < 4; ++a) dst[a] = u[a];
    dst[4] = make_float4(0.f, 0.f, 0.f, 0.f);
#if CLHIP_W16G_ADIRECT
    float* Ud = U + (size_t)kts * n_chunks * W_FLOATS;
    const int chunk4 = 2 * chunk + (c_l >> 2), q = c_l & 3, wk = k_l >> 5, r = (k_l >> 4) & 1, ti = k_l & 15;
    float4* dd = reinterpret_cast<float4*>(Ud + ((((size_t)(kt * 2 * n_chunks + chunk4) * 2 + wk) * 2 + r) * 4) * 256 + (q * 16 + ti) * 4);
#pragma unroll
    for (int a = 0; a < 4; ++a) dd[a * 64] = u[a];
#else
    (void)kts;
#endif
}

template <int TCB, int TRB, int NIMG>
struct WGeoW {
    static_assert(TCB * TRB * NIMG == 64, "64 tiles per block");
    static constexpr int PW = 2 * TCB + 2;                 // halo plane row length (even)
    static constexpr int PR = 2 * TRB + 2;                 // halo plane rows per image
    static constexpr int PLANE = NIMG * PR * PW;           // floats per channel
    static constexpr int X_FLOATS = WCK * PLANE;
    static constexpr int BUF = W_FLOATS + X_FLOATS;
};

// U[kt][chunk][c][k][f (16 of WFP)] = (G g G^T)[f] of g = w[k][c] (MODE 0) or of the 180-degree-rotated w[c][k] (MODE 1: backward-data,
// where the kernel's "input channels" are the convolution's output channels).  Ko / Ci: channel counts as the KERNEL sees them.
__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int Ko, int Ci,
                                                          int mode, int n_chunks) {
    const int total = ((Ko + WKT - 1) / WKT) * n_chunks * WCK * WKT;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int k_l = i % WKT, c_l = (i / WKT) % WCK, chunk = (i / (WKT * WCK)) % n_chunks, kt = i / (WKT * WCK * n_chunks);
        const int k = kt * WKT + k_l, c = chunk * WCK + c_l;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                float v = 0.f;
                if (k < Ko && c < Ci)
                    v = mode == 0 ? w[((size_t)k * Ci + c) * 9 + r * 3 + s]
                                  : w[((size_t)c * Ko + k) * 9 + (2 - r) * 3 + (2 - s)];     // w[c_out = c][c_in = k], flipped
                g[r][s] = v;
            }
        // t = G g  (4x3), u = t G^T (4x4);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
        float4 u[4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
            u[a] = make_float4(t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]);
        wino_u_store(U, (Ko + WKT - 1) / WKT, n_chunks, kt, chunk, c_l, k_l, u);
    }
}

// The same for several (layer, mode) pairs in ONE launch: the plan executor transforms the weights of every Winograd layer of a
// pass at its start (six ~5 us launches per pass of small_VGG9 were 3 % of the step).
constexpr int WT_JOBS = 24;
struct WtJobs { int n; int pad; clhip_wino_wt j[WT_JOBS]; int first[WT_JOBS + 1]; };      // pad: bit i = job i is a bf16-split image (bs_weight.hpp)

__device__ __forceinline__ void wino_weight_one(const float* __restrict__ w, float* __restrict__ U, int Ko, int Ci, int mode,
                                                int n_chunks, int i) {
    const int k_l = i % WKT, c_l = (i / WKT) % WCK, chunk = (i / (WKT * WCK)) % n_chunks, kt = i / (WKT * WCK * n_chunks);
    const int k = kt * WKT + k_l, c = chunk * WCK + c_l;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            float v = 0.f;
            if (k < Ko && c < Ci)
                v = mode == 0 ? w[((size_t)k * Ci + c) * 9 + r * 3 + s] : w[((size_t)c * Ko + k) * 9 + (2 - r) * 3 + (2 - s)];
            g[r][s] = v;
        }
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        t[0][s] = g[0][s];
        t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
        t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
        t[3][s] = g[2][s];
    }
    float4 u[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
        u[a] = make_float4(t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]);
    wino_u_store(U, (Ko + WKT - 1) / WKT, n_chunks, kt, chunk, c_l, k_l, u);
}

__global__ __launch_bounds__(256) void wino_weight_multi_kernel(WtJobs J) {
    int jb = 0;
    for (int i = 1; i < J.n; ++i) jb = ((int)blockIdx.x >= J.first[i]) ? i : jb;
    const clhip_wino_wt& q = J.j[jb];
    if ((J.pad >> jb) & 1) {                 // (uniform per block) a bf16-split image of the same pass
        bs_weight_block(q, (int)blockIdx.x - J.first[jb], (int)threadIdx.x);
        return;
    }
    const int n_chunks = (q.Ci + WCK - 1) / WCK;
    const int total = ((q.Ko + WKT - 1) / WKT) * n_chunks * WCK * WKT;
    const int i = ((int)blockIdx.x - J.first[jb]) * 256 + threadIdx.x;
    if (i < total) wino_weight_one(q.w, q.U, q.Ko, q.Ci, q.mode, n_chunks, i);
}

// MODE 0: forward   — out = [relu](conv(in, w) + bias), optionally 2x2-max-pooled with arg-max codes (pool_idx != NULL)
// MODE 1: backward-data — in = dy (Cin = the layer's out channels), out = dx (* (mask_src > 0) when mask_src != NULL);
//         UNPOOL: `in` is the gradient w.r.t. the POOLED output + the forward's arg-max codes (fused max_pool2d backward)
//         ROWPACK (TRB = 1): the block's NIMG "images" are NIMG consecutive TILE ROWS of the (image, tile row) list — odd maps
//         (AlexNet's 13 x 13: 7 x 7 tiles) fill 49 of 56 tile slots instead of 49 of 64; odd H / W: the last tile row / column
//         is half outside (its inputs read 0, its outputs are not stored; 4-byte stores, rows of odd length are not 8-byte aligned)
template <int TCB, int TRB, int NIMG, int MODE, bool UNPOOL, bool ROWPACK = false>
__global__ __launch_bounds__(256, 1) void wino_conv_kernel(
    const float* __restrict__ in, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int H, int W, int relu, int tiles_w, int tiles_h, int n_pix_blocks) {
    using G = WGeoW<TCB, TRB, NIMG>;
    static_assert(!ROWPACK || (TRB == 1 && !UNPOOL), "row packing: one tile row per entity, no fused un-pooling");
    __shared__ __attribute__((aligned(16))) float lds[2 * G::BUF];
    __shared__ float bias_s[WKT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wp = wave >> 1;
    const int li = lane & 31, kk = lane >> 5;

    const int kt = blockIdx.x / n_pix_blocks, pb = blockIdx.x - kt * n_pix_blocks;
    const int bw = pb % tiles_w, bh = (pb / tiles_w) % tiles_h, ng = pb / (tiles_w * tiles_h);
    // ROWPACK: tiles_h = tile rows per image, entity v = v0 + nb = (image v / tiles_h, tile row v % tiles_h); offsets relative to image n0
    const int v0 = ROWPACK ? pb * NIMG : 0;
    const int n0 = ROWPACK ? v0 / tiles_h : ng * NIMG, h0 = ROWPACK ? 0 : bh * 2 * TRB, w0 = ROWPACK ? 0 : bw * 2 * TCB;
    const int ko0 = kt * WKT;
    const int n_chunks = (Cin + WCK - 1) / WCK;
    if (MODE == 0 && tid < WKT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;

    const int Hi = UNPOOL ? H >> 1 : H, Wi = UNPOOL ? W >> 1 : W;          // geometry of the tensor `in` points at
    const size_t plane_in = (size_t)Hi * Wi;
    const float* in_blk = in + (size_t)n0 * Cin * plane_in;
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_blk, (size_t)(N - n0) * Cin * plane_in * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? pool_idx + (size_t)n0 * Cin * plane_in : pool_idx,
                                                   UNPOOL ? (size_t)(N - n0) * Cin * plane_in : 0);
    const __amdgpu_buffer_rsrc_t rs_u = clhip_rsrc(U + (size_t)kt * n_chunks * W_FLOATS, (size_t)n_chunks * W_FLOATS * sizeof(float));

    // ------------------------------------------------------------------ staging (registers -> LDS, one chunk ahead)
    // weights: straight copy of 2048 float4; activations: one scalar per halo-plane element (raw buffer loads: padding,
    // image tails and channel tails read 0 through the OFFSET predicate — never a branch, never a select on a loaded value).
    // A chunk's staging is cut into UNITS of one instruction each, issued one per MFMA slot (see the pipeline below).
    constexpr int W_IT = W_FLOATS / 4 / 256;                         // 10
    constexpr int X_IT = (G::X_FLOATS + 255) / 256;
    constexpr int NU = W_IT + X_IT;
    float4 wv[W_IT];
    float xr[X_IT];
    unsigned xi[UNPOOL ? X_IT : 1];
    int xoff[X_IT], xcode[UNPOOL ? X_IT : 1];
#pragma unroll
    for (int j = 0; j < X_IT; ++j) {
        const int e = tid + 256 * j;
        xoff[j] = CLHIP_OOB;
        if (e < G::X_FLOATS) {
            const int cl = e / G::PLANE, rem = e - cl * G::PLANE;
            const int col = rem % G::PW, rr = rem / G::PW;
            const int row = rr % G::PR, nb = rr / G::PR;
            int n = n0 + nb, h = h0 - 1 + row;
            const int w = w0 - 1 + col;
            if constexpr (ROWPACK) { const int v = v0 + nb; n = v / tiles_h; h = 2 * (v - n * tiles_h) - 1 + row; }
            const int nr = n - n0;                                       // image relative to the block's base pointer
            if (n < N && h >= 0 && h < H && w >= 0 && w < W) {
                if constexpr (UNPOOL) {
                    xoff[j] = (int)(((size_t)nr * Cin + cl) * plane_in) + (h >> 1) * Wi + (w >> 1);     // ELEMENT offset
                    xcode[j] = ((h & 1) << 1) | (w & 1);
                } else {
                    xoff[j] = ((int)(((size_t)nr * Cin + cl) * plane_in) + h * W + w) * 4;
                }
            }
        }
    }
    // unit u of chunk `chunk` -> staging registers.  Cin is a whole number of chunks (clhip_internal_wino_ok), so a live
    // chunk needs no channel predicate; the two prefetches past the last chunk re-read it (never used).
    auto load_unit = [&](int u, int chunk) {
        const int cw = chunk < n_chunks ? chunk : n_chunks - 1;
        if (u < W_IT) {
            wv[u] = clhip_buf_load4(rs_u, (tid + 256 * u) * 16, cw * W_FLOATS * 4);
        } else {
            const int j = u - W_IT;
            const int xb = cw * WCK * (int)plane_in;                 // wave-uniform scalar offset
            if constexpr (UNPOOL) {
                xr[j] = clhip_buf_load(rs_x, xoff[j] != CLHIP_OOB ? xoff[j] * 4 : CLHIP_OOB, xb * 4);
                xi[j] = clhip_buf_load_u8(rs_i, xoff[j], xb);
            } else {
                xr[j] = clhip_buf_load(rs_x, xoff[j], xb * 4);
            }
        }
    };
    auto store_unit = [&](int u, int bo) {
        if (u < W_IT) {
            float* wd = lds + bo + 4 * (tid + 256 * u);
            *reinterpret_cast<floatx4*>(wd) = floatx4{wv[u].x, wv[u].y, wv[u].z, wv[u].w};
        } else {
            const int j = u - W_IT;
            float* xs = lds + bo + W_FLOATS;
            if (256 * (j + 1) <= G::X_FLOATS || tid + 256 * j < G::X_FLOATS) {
                if constexpr (UNPOOL) xs[tid + 256 * j] = ((int)xi[j] == xcode[j]) ? xr[j] : 0.f;
                else xs[tid + 256 * j] = xr[j];
            }
        }
    };

    // this lane's tile: index t = 32 * wp + li over (image, tile row, tile column)
    const int t_idx = 32 * wp + li;
    const int t_img = t_idx / (TRB * TCB), t_row = (t_idx / TCB) % TRB, t_col = t_idx % TCB;
    const int d_off = W_FLOATS + kk * G::PLANE + (t_img * G::PR + 2 * t_row) * G::PW + 2 * t_col;     // even: 8-byte aligned
    const int a_off = (kk * WKT + wk * 32 + li) * WFP;

    floatx16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    // ------------------------------------------------------------------ pipeline
    // One wave per SIMD: nothing hides a stall, and on this part every VALU / LDS instruction issues from the port the
    // fp32 MFMAs use (ablation on 512->512 @8x8: MFMAs alone 246 us, + transforms 63, + LDS reads 70, + staging 40), so the
    // loop is written to need FEW instructions per MFMA and each of them rides behind one MFMA "slot" (order pinned by
    // sched_barrier): while pair p's 16 MFMAs issue, slot f
    //   * f < 4:      reads one row of pair p + 1's 4x4 window                       (2 ds_read_b64),
    //   * f = 4..7:   reads four weight frequencies of pair p + 1                    (1 ds_read_b128),
    //   * f = 8..11 / 12..15: row / column half of the input transform B^T d B of pair p + 1 (packed fp32 adds),
    //   * pair 0: LDS writes of chunk c + 1, pairs 1-2: global loads of chunk c + 2    (one or two units per slot),
    // with ONE barrier per chunk behind the first MFMA of the last pair (every read of the current buffer has been issued
    // and waited for by then; the writes of chunk c + 1 were issued during pair 0).
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NP = WCK / 2;
    floatx4 av[2][4];            // [pipeline slot][frequency quad]
    float vv[2][16];
    f2 dlo[4], dhi[4], tlo[4], thi[4];       // window rows / row-transformed rows as (col 0, col 1) and (col 2, col 3)
    auto rd_a = [&](const float* abase, int pair, int q, int slot) {
        av[slot][q] = *reinterpret_cast<const floatx4*>(abase + 2 * pair * WKT * WFP + 4 * q);
    };
    auto rd_d = [&](const float* dbase, int pair, int r) {
        dlo[r] = *reinterpret_cast<const f2*>(dbase + 2 * pair * G::PLANE + r * G::PW);
        dhi[r] = *reinterpret_cast<const f2*>(dbase + 2 * pair * G::PLANE + r * G::PW + 2);
    };
    auto row_tf = [&](int i) {           // row i of t = B^T d, both column pairs
        if (i == 0) { tlo[0] = dlo[0] - dlo[2]; thi[0] = dhi[0] - dhi[2]; }
        if (i == 1) { tlo[1] = dlo[1] + dlo[2]; thi[1] = dhi[1] + dhi[2]; }
        if (i == 2) { tlo[2] = dlo[2] - dlo[1]; thi[2] = dhi[2] - dhi[1]; }
        if (i == 3) { tlo[3] = dlo[1] - dlo[3]; thi[3] = dhi[1] - dhi[3]; }
    };
    auto col_tf = [&](int i, int slot) { // row i of V = t B:  (V0, V3) = (t0, t1) - (t2, t3);  (V1, V2) = (t1 + t2, t2 - t1)
        const f2 o = tlo[i] - thi[i];
        vv[slot][4 * i + 0] = o.x;
        vv[slot][4 * i + 3] = o.y;
        vv[slot][4 * i + 1] = tlo[i].y + thi[i].x;
        vv[slot][4 * i + 2] = thi[i].x - tlo[i].y;
    };

#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) store_unit(u, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, 1);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) rd_a(lds + a_off, 0, q, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) rd_d(lds + d_off, 0, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_tf(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_tf(i, 0);

    static_assert(NU <= 32, "two store units per slot of pair 0, one load unit per slot of pairs 1-2");
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int bo = (chunk & 1) * G::BUF, bn = G::BUF - bo;
        const float* a_cur = lds + bo + a_off;
        const float* d_cur = lds + bo + d_off;
        const float* a_nxt = lds + bn + a_off;
        const float* d_nxt = lds + bn + d_off;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const float* ab = (p + 1 < NP) ? a_cur : a_nxt;
            const float* db = (p + 1 < NP) ? d_cur : d_nxt;
            const int np = (p + 1) % NP, ns = (p + 1) & 1, cs = p & 1;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][f >> 2][f & 3], vv[cs][f], acc[f], 0, 0, 0);
                if (p == NP - 1 && f == 0) __syncthreads();
                if (f < 4) rd_d(db, np, f);                 // (window first: the transform at slot 8 needs it landed)
                else if (f < 8) rd_a(ab, np, f - 4, ns);
                else if (f < 12) row_tf(f - 8);
                else col_tf(f - 12, ns);
                // (no inner loops over units: every register-array index must be a constant once p and f are unrolled,
                // otherwise the staging arrays are demoted to LDS-backed storage and each load is waited for at once)
                if (p == 0) {
                    if (2 * f < NU) store_unit(2 * f, bn);
                    if (2 * f + 1 < NU) store_unit(2 * f + 1, bn);
                } else if (p <= 2) {
                    if ((p - 1) * 16 + f < NU) load_unit((p - 1) * 16 + f, chunk + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ------------------------------------------------------------------ epilogue: Y = A^T M A per accumulator register
    // register r of lane l = (out channel kb + rch(r), tile l) in every frequency accumulator
    const int kb = ko0 + wk * 32 + 4 * kk;
    auto rch = [](int r) { return (r & 3) + 8 * (r >> 2); };
    int n = n0 + t_img, oh = h0 + 2 * t_row;
    const int ow = w0 + 2 * t_col;
    if constexpr (ROWPACK) { const int v = v0 + t_img; n = v / tiles_h; oh = 2 * (v - n * tiles_h); }
    const bool tile_ok = n < N && oh < H && ow < W;                 // H, W even: a tile is inside or outside as a whole
    const bool odd = (H | W) & 1;                                   // uniform; the last tile row / column may be half outside
    const bool row1 = oh + 1 < H, col1 = ow + 1 < W;
    const bool pool = MODE == 0 && pool_idx != nullptr;
    // output stores with the output cache policy (common.hpp) while 32-bit byte offsets reach the whole tensor, plain stores beyond
    const bool st32 = (size_t)N * Cout * (pool ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * sizeof(float) <= 0x7fffffffull;
    const __amdgpu_buffer_rsrc_t rs_out = clhip_out_rsrc(out), rs_code = clhip_out_rsrc(pool_idx);
    auto st1 = [&](size_t o, float v) { if (st32) clhip_buf_store(v, rs_out, (int)o * 4, 0); else out[o] = v; };
    auto st2 = [&](size_t o, float a, float b) {
        if (st32) clhip_buf_store2(make_float2(a, b), rs_out, (int)o * 4, 0); else *reinterpret_cast<float2*>(out + o) = make_float2(a, b);
    };
    auto stc = [&](size_t o, int a) { if (st32) clhip_buf_store_u8((uint8_t)a, rs_code, (int)o, 0); else pool_idx[o] = (uint8_t)a; };
    const size_t chw = (size_t)H * W;
    const int OH = H >> 1, OW = W >> 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float u0[4], u1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u0[j] = acc[j][r] + acc[4 + j][r] + acc[8 + j][r];
            u1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
        }
        float y00 = u0[0] + u0[1] + u0[2], y01 = u0[1] - u0[2] - u0[3];
        float y10 = u1[0] + u1[1] + u1[2], y11 = u1[1] - u1[2] - u1[3];
        const int k = kb + rch(r);
        const bool ok = tile_ok && k < Cout;
        if (MODE == 0) {
            const float b = bias_s[wk * 32 + 4 * kk + rch(r)];
            y00 += b; y01 += b; y10 += b; y11 += b;
            if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            if (pool) {
                float m = y00; int a = 0;              // first maximum in ATen's scan order wins
                if (y01 > m) { m = y01; a = 1; }
                if (y10 > m) { m = y10; a = 2; }
                if (y11 > m) { m = y11; a = 3; }
                if (relu && !(m > 0.f)) a = CLHIP_POOL_DEAD;   // ReLU folded into the code (see common.hpp)
                if (ok) {
                    const size_t o = ((size_t)n * Cout + k) * OH * OW + (size_t)(oh >> 1) * OW + (ow >> 1);
                    st1(o, m);
                    stc(o, a);
                }
                continue;
            }
        }
        if (ok && odd) {
            const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
            if (MODE == 1 && mask_src) {
                y00 = mask_src[o] > 0.f ? y00 : 0.f;
                if (col1) y01 = mask_src[o + 1] > 0.f ? y01 : 0.f;
                if (row1) y10 = mask_src[o + W] > 0.f ? y10 : 0.f;
                if (row1 && col1) y11 = mask_src[o + W + 1] > 0.f ? y11 : 0.f;
            }
            st1(o, y00);
            if (col1) st1(o + 1, y01);
            if (row1) st1(o + W, y10);
            if (row1 && col1) st1(o + W + 1, y11);
        } else if (ok) {
            const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
            if (MODE == 1 && mask_src) {
                const float2 m0 = *reinterpret_cast<const float2*>(mask_src + o), m1 = *reinterpret_cast<const float2*>(mask_src + o + W);
                y00 = m0.x > 0.f ? y00 : 0.f; y01 = m0.y > 0.f ? y01 : 0.f;
                y10 = m1.x > 0.f ? y10 : 0.f; y11 = m1.y > 0.f ? y11 : 0.f;
            }
            st2(o, y00, y01);
            st2(o + W, y10, y11);
        }
    }
}


// ------------------------------------------------------------------------------------------------ weight gradient
// dL/dg = G^T [ (A dY A^T) .* (B^T d B) ] G  summed over tiles and images: 16 GEMMs  M_f[k][c] = sum_tiles DY_f[k][tile] V_f[c][tile]
// with the TILES as the reduction dimension (two per MFMA): D rows = 32 out-channels (A operand: lane = k, half = tile of the
// pair), D cols = 32 in-channels (B operand: lane = c).  Both operands are transformed in registers from raw LDS tiles —
// a lane reads the 2x2 dy tile of its out-channel (2 ds_read_b64) and the 4x4 x window of its in-channel (8 ds_read_b64)
// and forms 16 + 16 frequency values with 12 + 32 adds next to 16 MFMAs; the direct kernel (conv3x3_wgrad.hip) issues 36
// MFMAs for the same two tiles.  The reduction over (image, tile) is split across blocks exactly like the direct kernel's:
// every block applies the output transform G^T M G in registers and writes a partial [9][K][C] (+ [K] bias sums) slab
// of the SAME format, so the fixed-order slab reduction (wgrad_reduce_multi) serves both.
// Signs: A dY A^T has rows / columns 3 negated (A = [[1,0],[1,1],[1,-1],[0,-1]]); the kernel accumulates with the un-negated
// values and flips M_f for f in row 3 xor column 3 inside the output transform.
template <int TCS, int TRS, int NIS>
struct WGeoG {
    static constexpr int TILES = TCS * TRS * NIS;          // 16 tiles per stage
    static_assert(TILES == 16, "16 tiles per stage");
    static constexpr int DW = 2 * TCS, DR = 2 * TRS;       // dy tile of one image: DR rows x DW cols
    static constexpr int DPIX = NIS * DR * DW;             // 64 pixels
    static constexpr int LDP = DPIX + 2;                   // dy row stride per out-channel (8-byte units odd => conflict-free b64)
    static constexpr int PW = DW + 2, PR = DR + 2;
    static constexpr int PLANE = NIS * PR * PW;
    static constexpr int PLANEP = (PLANE % 4 == 2) ? PLANE : PLANE + 2;      // even, PLANEP / 2 odd
    static_assert((LDP / 2) % 2 == 1 && (PLANEP / 2) % 2 == 1, "lane strides must be odd in 8-byte units");
    static constexpr int DY_FLOATS = WKT * LDP, X_FLOATS = WKT * PLANEP;
    static constexpr int BUF = DY_FLOATS + X_FLOATS;
};

// VEC (one image per stage, maps of whole tiles in width, 16-byte-aligned tensors): the staging units are 16-byte pieces as in
// wino_wgrad_ps_kernel below — interior float4s of the halo-plane rows + the two halo columns, dy rows as float4s: 15 - 16 units per
// stage where the scalar form has 48 (27 - 28 against 48 when dy is the pooled gradient, which stays one element + code per unit).
template <int TCS, int TRS, int NIS, bool UNPOOL, bool VEC>
__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, const uint8_t* __restrict__ unpool_idx,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h, int total_stages, int splits, int c_tiles, size_t slab_stride) {
    using G = WGeoG<TCS, TRS, NIS>;
    __shared__ __attribute__((aligned(16))) float lds[2 * G::BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wc = wave >> 1;
    const int li = lane & 31, kk = lane >> 5;
    const int split = blockIdx.x % splits, tile = blockIdx.x / splits;
    const int ct = tile % c_tiles, kt = tile / c_tiles;
    const int k0 = kt * WKT, c0 = ct * WKT;
    const int per = total_stages / splits, extra = total_stages % splits;
    const int st_begin = split * per + min(split, extra);
    const int st_end = st_begin + per + (split < extra ? 1 : 0);
    const int Hd = UNPOOL ? H >> 1 : H, Wd = UNPOOL ? W >> 1 : W;          // geometry of the tensor `dy` points at
    const int plane_hw = H * W, plane_dy = Hd * Wd;

    // ------------------------------------------------------------------ staging units (one scalar raw buffer load each)
    // dy: thread -> pixel q = tid & 63 of out-channel (tid >> 6) + 4 j;  x: thread -> halo-plane position tid % PLANE of
    // in-channel (tid / PLANE) + CPT j  (threads past CPT * PLANE idle): every per-unit index is base + j * constant, so the
    // staging needs a handful of registers instead of six index arrays
    constexpr int DY_IT = WKT * G::DPIX / 256;                        // 16
    constexpr int CPT = 256 / G::PLANE;                               // channels staged per pass
    static_assert(CPT >= 1 && WKT % CPT == 0, "whole passes over the 64 in-channels");
    constexpr int X_IT = WKT / CPT;
    static_assert(!VEC || NIS == 1, "16-byte staging: one image per stage");
    constexpr int RSEG = G::DW / 4;
    constexpr int XU = G::PR * RSEG, XCPT = 256 / XU, XV_IT = (WKT + XCPT - 1) / XCPT;              // 24, 10, 7  |  20, 12, 6
    constexpr int HU = G::PR * 2, HCPT = 256 / HU, XH_IT = (WKT + HCPT - 1) / HCPT;                 // 12, 21, 4  |  20, 12, 6
    constexpr int DQV = G::DR * RSEG, DKSV = 256 / DQV, DYV_IT = WKT / DKSV;                         // 16, 16, 4
    static_assert(!VEC || (DQV * DKSV == 256 && DYV_IT * DKSV == WKT), "dy float4 units fill the block exactly");
    constexpr bool DYV = VEC && !UNPOOL;
    constexpr int DYU = DYV ? DYV_IT : DY_IT;                         // dy units of a stage
    constexpr int NU = VEC ? DYU + XV_IT + XH_IT : DY_IT + X_IT;
    float dyr[DYV ? 1 : DY_IT], xr[VEC ? 1 : X_IT];
    float4 dyv[DYV ? DYV_IT : 1], xv[VEC ? XV_IT : 1];
    float xh[VEC ? XH_IT : 1];
    const int v_cl = tid / XU, v_rem = tid - v_cl * XU, v_row = v_rem / RSEG, v_seg = v_rem - v_row * RSEG;
    const bool v_thr = tid < XCPT * XU;
    const int v_e0 = v_cl * plane_hw + v_row * W + 1 + 4 * v_seg;                   // relative to the halo origin (h0 - 1, w0 - 1)
    const int v_dst0 = G::DY_FLOATS + v_cl * G::PLANEP + v_row * G::PW + 1 + 4 * v_seg;
    const int h_cl = tid / HU, h_rem = tid - h_cl * HU, h_row = h_rem >> 1, h_col = (h_rem & 1) ? G::DW + 1 : 0;
    const bool h_thr = tid < HCPT * HU;
    const int h_e0 = h_cl * plane_hw + h_row * W + h_col;
    const int h_dst0 = G::DY_FLOATS + h_cl * G::PLANEP + h_row * G::PW + h_col;
    const int dv_kl = tid / DQV, dv_q = tid - dv_kl * DQV, dv_r = dv_q / RSEG, dv_seg = dv_q - dv_r * RSEG;
    const int dv_e0 = dv_kl * plane_dy + dv_r * Wd + 4 * dv_seg;
    const int dv_dst0 = dv_kl * G::LDP + dv_r * G::DW + 4 * dv_seg;                 // even: 8-byte aligned
    bool xv_ok = false, xh_ok = false, dv_ok = false;
    unsigned dyi[UNPOOL ? DY_IT : 1];
    const int dq = tid & 63, dkl = tid >> 6;
    const int d_nb = dq / (G::DR * G::DW), d_r = (dq / G::DW) % G::DR, d_c = dq % G::DW;
    const int dy_e0 = UNPOOL ? ((d_nb * K + dkl) * plane_dy + (d_r >> 1) * Wd + (d_c >> 1)) : ((d_nb * K + dkl) * plane_dy + d_r * W + d_c);
    const int dy_dst0 = dkl * G::LDP + dq;
    const int d_code = ((d_r & 1) << 1) | (d_c & 1);
    const bool x_thr = tid < CPT * G::PLANE;
    const int x_cl = tid / G::PLANE, x_rem = tid - x_cl * G::PLANE;
    const int x_col = x_rem % G::PW, x_rr = x_rem / G::PW, x_row = x_rr % G::PR, x_nb = x_rr / G::PR;
    const int x_e0 = (x_nb * C + x_cl) * plane_hw + x_row * W + x_col;     // relative to the (image, halo origin) base
    const int x_dst0 = G::DY_FLOATS + x_cl * G::PLANEP + x_rem;
    // stage -> (image group, tile-row block, tile-col block)
    int ld_n0 = 0, ld_h0 = 0, ld_w0 = 0;
    bool dy_ok = false, x_ok = false;
    __amdgpu_buffer_rsrc_t rs_dy = clhip_rsrc(dy, 0), rs_x = clhip_rsrc(x, 0), rs_di = clhip_rsrc(x, 0);
    auto begin_stage = [&](int st) {
        const bool live = st < st_end;
        const int s = live ? st : st_begin;
        const int bw = s % tiles_w, bh = (s / tiles_w) % tiles_h, ng = s / (tiles_w * tiles_h);
        ld_n0 = ng * NIS; ld_h0 = bh * G::DR; ld_w0 = bw * G::DW;
        const int org_dy = UNPOOL ? (ld_h0 >> 1) * Wd + (ld_w0 >> 1) : ld_h0 * W + ld_w0;
        const float* dyb = dy + ((size_t)ld_n0 * K + k0) * plane_dy + org_dy;
        const long long dy_left = ((long long)(N - ld_n0) * K - k0) * plane_dy - org_dy;
        rs_dy = clhip_rsrc(dyb, live && dy_left > 0 ? (size_t)dy_left * 4 : 0);
        if constexpr (UNPOOL) rs_di = clhip_rsrc(unpool_idx + ((size_t)ld_n0 * K + k0) * plane_dy + org_dy, live && dy_left > 0 ? (size_t)dy_left : 0);
        const long long org_x = (long long)ld_h0 * W + ld_w0 - W - 1;
        const float* xb = x + ((size_t)ld_n0 * C + c0) * plane_hw + org_x;
        const long long x_left = ((long long)(N - ld_n0) * C - c0) * plane_hw - org_x;
        rs_x = clhip_rsrc(xb, live && x_left > 0 ? (size_t)x_left * 4 : 0);
        dy_ok = ld_n0 + d_nb < N && ld_h0 + d_r < H && ld_w0 + d_c < W;
        const int h = ld_h0 - 1 + x_row, w = ld_w0 - 1 + x_col;
        x_ok = x_thr && ld_n0 + x_nb < N && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        if constexpr (VEC) {                      // whole tiles in width: the interior columns of a row are all inside the image
            xv_ok = v_thr && ld_n0 < N && (unsigned)(ld_h0 - 1 + v_row) < (unsigned)H;
            xh_ok = h_thr && ld_n0 < N && (unsigned)(ld_h0 - 1 + h_row) < (unsigned)H && (unsigned)(ld_w0 - 1 + h_col) < (unsigned)W;
            dv_ok = ld_n0 < N && ld_h0 + dv_r < H;
        }
    };
    auto load_unit = [&](int u) {
        if constexpr (VEC) {
            if (u < DYU) {
                if constexpr (DYV) {
                    dyv[u] = clhip_buf_load4(rs_dy, dv_ok ? (dv_e0 + u * DKSV * plane_dy) * 4 : CLHIP_OOB, 0);
                } else {
                    const int e = dy_e0 + u * 4 * plane_dy;
                    dyr[u] = clhip_buf_load(rs_dy, dy_ok ? e * 4 : CLHIP_OOB, 0);
                    dyi[u] = clhip_buf_load_u8(rs_di, dy_ok ? e : CLHIP_OOB, 0);
                }
            } else if (u < DYU + XV_IT) {
                const int j = u - DYU;
                const bool ok = xv_ok && ((j + 1) * XCPT <= WKT || v_cl + j * XCPT < WKT);
                xv[j] = clhip_buf_load4(rs_x, ok ? (v_e0 + j * XCPT * plane_hw) * 4 : CLHIP_OOB, 0);
            } else {
                const int j = u - DYU - XV_IT;
                const bool ok = xh_ok && ((j + 1) * HCPT <= WKT || h_cl + j * HCPT < WKT);
                xh[j] = clhip_buf_load(rs_x, ok ? (h_e0 + j * HCPT * plane_hw) * 4 : CLHIP_OOB, 0);
            }
            return;
        }
        if (u < DY_IT) {
            const int e = dy_e0 + u * 4 * plane_dy;
            dyr[u] = clhip_buf_load(rs_dy, dy_ok ? e * 4 : CLHIP_OOB, 0);
            if constexpr (UNPOOL) dyi[u] = clhip_buf_load_u8(rs_di, dy_ok ? e : CLHIP_OOB, 0);
        } else {
            const int j = u - DY_IT;
            xr[j] = clhip_buf_load(rs_x, x_ok ? (x_e0 + j * CPT * plane_hw) * 4 : CLHIP_OOB, 0);
        }
    };
    auto store_unit = [&](int u, int bo) {
        if constexpr (VEC) {
            typedef float f2s __attribute__((ext_vector_type(2)));
            if (u < DYU) {
                if constexpr (DYV) {
                    float* d = lds + bo + dv_dst0 + u * DKSV * G::LDP;
                    *reinterpret_cast<f2s*>(d) = f2s{dyv[u].x, dyv[u].y};
                    *reinterpret_cast<f2s*>(d + 2) = f2s{dyv[u].z, dyv[u].w};
                } else {
                    lds[bo + dy_dst0 + u * 4 * G::LDP] = ((int)dyi[u] == d_code) ? dyr[u] : 0.f;
                }
            } else if (u < DYU + XV_IT) {
                const int j = u - DYU;
                if (v_thr && ((j + 1) * XCPT <= WKT || v_cl + j * XCPT < WKT)) {
                    float* d = lds + bo + v_dst0 + j * XCPT * G::PLANEP;
                    d[0] = xv[j].x; d[1] = xv[j].y; d[2] = xv[j].z; d[3] = xv[j].w;
                }
            } else {
                const int j = u - DYU - XV_IT;
                if (h_thr && ((j + 1) * HCPT <= WKT || h_cl + j * HCPT < WKT)) lds[bo + h_dst0 + j * HCPT * G::PLANEP] = xh[j];
            }
            return;
        }
        if (u < DY_IT) {
            float v = dyr[u];
            if constexpr (UNPOOL) v = ((int)dyi[u] == d_code) ? v : 0.f;
            lds[bo + dy_dst0 + u * 4 * G::LDP] = v;
        } else {
            const int j = u - DY_IT;
            if (x_thr) lds[bo + x_dst0 + j * CPT * G::PLANEP] = xr[j];
        }
    };

    floatx16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    float bsum = 0.f;

    // lane operands: A = dy tile of out-channel wk*32 + li, B = x window of in-channel wc*32 + li; tile t = 2 * step + kk
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NSTEP = G::TILES / 2;
    const int a_lane = (wk * 32 + li) * G::LDP;
    const int b_lane = G::DY_FLOATS + (wc * 32 + li) * G::PLANEP;
    auto tile_off = [&](int step, int& doff, int& xo) {
        const int t = 2 * step + kk;
        const int ti = t / (TRS * TCS), tr = (t / TCS) % TRS, tc = t % TCS;
        doff = (ti * G::DR + 2 * tr) * G::DW + 2 * tc;
        xo = (ti * G::PR + 2 * tr) * G::PW + 2 * tc;
    };
    int doffs[NSTEP], xoffs[NSTEP];
#pragma unroll
    for (int sidx = 0; sidx < NSTEP; ++sidx) tile_off(sidx, doffs[sidx], xoffs[sidx]);
    f2 ya[2], xlo[4], xhi[4];            // raw dy tile rows / x window rows of the NEXT step
    float av[2][16], vv[2][16];
    auto rd = [&](const float* buf, int step, int which) {       // which: 0 dy rows, 1..4 x rows
        if (which == 0) {
            ya[0] = *reinterpret_cast<const f2*>(buf + a_lane + doffs[step]);
            ya[1] = *reinterpret_cast<const f2*>(buf + a_lane + doffs[step] + G::DW);
        } else {
            const int r = which - 1;
            xlo[r] = *reinterpret_cast<const f2*>(buf + b_lane + xoffs[step] + r * G::PW);
            xhi[r] = *reinterpret_cast<const f2*>(buf + b_lane + xoffs[step] + r * G::PW + 2);
        }
    };
    f2 tlo[4], thi[4];
    auto tf = [&](int part, int slot) {
        if (part == 0) {                                            // A dY A^T without the negations (see header)
            const float a = ya[0].x, b = ya[0].y, c = ya[1].x, d = ya[1].y;
            bsum += (a + b) + (c + d);
            const float r1p = a + c, r1q = b + d, r2p = a - c, r2q = b - d;
            float* o = av[slot];
            o[0] = a;   o[1] = a + b;     o[2] = a - b;     o[3] = b;
            o[4] = r1p; o[5] = r1p + r1q; o[6] = r1p - r1q; o[7] = r1q;
            o[8] = r2p; o[9] = r2p + r2q; o[10] = r2p - r2q; o[11] = r2q;
            o[12] = c;  o[13] = c + d;    o[14] = c - d;    o[15] = d;
        } else if (part == 1) {                                     // rows of B^T d
            tlo[0] = xlo[0] - xlo[2]; thi[0] = xhi[0] - xhi[2];
            tlo[1] = xlo[1] + xlo[2]; thi[1] = xhi[1] + xhi[2];
        } else if (part == 2) {
            tlo[2] = xlo[2] - xlo[1]; thi[2] = xhi[2] - xhi[1];
            tlo[3] = xlo[1] - xlo[3]; thi[3] = xhi[1] - xhi[3];
        } else {                                                     // rows part - 3 .. of (B^T d) B
            const int i = part - 3;
            const f2 o = tlo[i] - thi[i];
            vv[slot][4 * i + 0] = o.x;
            vv[slot][4 * i + 3] = o.y;
            vv[slot][4 * i + 1] = tlo[i].y + thi[i].x;
            vv[slot][4 * i + 2] = thi[i].x - tlo[i].y;
        }
    };

    if (st_begin < st_end) {
        begin_stage(st_begin);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
#pragma unroll
        for (int u = 0; u < NU; ++u) store_unit(u, 0);
        begin_stage(st_begin + 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) load_unit(u);
        __syncthreads();
#pragma unroll
        for (int w5 = 0; w5 < 5; ++w5) rd(lds, 0, w5);
#pragma unroll
        for (int part = 0; part < 7; ++part) tf(part, 0);
    }
    static_assert(NU <= 48, "staging units fit the slots of three steps");
    for (int st = st_begin; st < st_end; ++st) {
        const int bo = ((st - st_begin) & 1) * G::BUF, bn = G::BUF - bo;
        const float* cur = lds + bo;
        const float* nxt = lds + bn;
#pragma unroll
        for (int step = 0; step < NSTEP; ++step) {
            const float* nb = (step + 1 < NSTEP) ? cur : nxt;
            const int nstep = (step + 1) % NSTEP, ns = (step + 1) & 1, cs = step & 1;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][f], vv[cs][f], acc[f], 0, 0, 0);
                if (step == NSTEP - 1 && f == 0) __syncthreads();
                if (f < 5) rd(nb, nstep, f);
                else if (f >= 8 && f < 15) tf(f - 8, ns);
                // staging: LDS writes of stage st + 1 during steps 0-2, global loads of stage st + 2 during steps 3-5
                if (step < 3) { if (step * 16 + f < NU) store_unit(step * 16 + f, bn); }
                else if (step < 6) {
                    if (step == 3 && f == 0) begin_stage(st + 2);
                    if ((step - 3) * 16 + f < NU) load_unit((step - 3) * 16 + f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ------------------------------------------------------------------ output transform  dW = G^T M' G  per accumulator register
    // register r of lane l = (k = k0 + wk*32 + row(r, l), c = c0 + wc*32 + li); M'_f carries the sign (-1)^{[i == 3] + [j == 3]}
    float* slab = part + (size_t)split * slab_stride;
    const __amdgpu_buffer_rsrc_t rs_slab = clhip_out_rsrc(slab);          // (output cache policy: common.hpp)
    const int cidx = c0 + wc * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = ((i == 3) != (j == 3)) ? -acc[4 * i + j][r] : acc[4 * i + j][r];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s12 = 0.5f * (m[1][j] + m[2][j]);
            t[0][j] = m[0][j] + s12;
            t[1][j] = 0.5f * (m[1][j] - m[2][j]);
            t[2][j] = s12 + m[3][j];
        }
        const int k = k0 + wk * 32 + mfma32_row(r, lane);
        if (k < K && cidx < C) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float s12 = 0.5f * (t[a][1] + t[a][2]);
                clhip_buf_store(t[a][0] + s12, rs_slab, (int)(((3 * a + 0) * K + k) * C + cidx) * 4, 0);
                clhip_buf_store(0.5f * (t[a][1] - t[a][2]), rs_slab, (int)(((3 * a + 1) * K + k) * C + cidx) * 4, 0);
                clhip_buf_store(s12 + t[a][3], rs_slab, (int)(((3 * a + 2) * K + k) * C + cidx) * 4, 0);
            }
        }
    }
    if (ct == 0 && wc == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        const int k = k0 + wk * 32 + li;
        if (kk == 0 && k < K) clhip_buf_store(bsum, rs_slab, (int)(9 * K * C + k) * 4, 0);
    }
}


// ------------------------------------------------------------------------------------------------ 8 x 8 maps, few units
// The kernel above gives a wave 32 out-channels x 32 TILES (256 accumulators): a 128 -> 128 layer on 8 x 8 maps at N = 200
// has only 400 such units for 1024 SIMDs and ran at 0.75x the direct kernel.  This variant computes the same F(2x2, 3x3)
// product on v_mfma_f32_16x16x4_f32 (16 out-channels x 16 tiles x 4 in-channels, 32 cycles): a wave owns ONE image (its
// 4 x 4 tiles) x 32 out-channels (two 16-row MFMA tiles sharing the transformed input), 128 accumulator registers, twice as
// many units.  A lane is (tile i = lane & 15, channel q = lane >> 4 of the current channel quad): it reads its 4x4 window
// (8 ds_read_b64), forms 16 frequency values (32 adds) and feeds them to 2 x 16 MFMAs; A operands are its 16 frequencies of
// U[c = q][k = 16 kt + i] (4 ds_read_b128 per row tile) — the U layout, the weight-transform kernels and the epilogue
// arithmetic are the ones of the kernel above.  Same slot pipeline: 16 slots of two MFMAs per channel quad, each followed by
// one piece of the next quad's operand work; LDS writes of chunk c + 1 during quad 0, global loads of chunk c + 2 during
// quad 1, one barrier per 8-channel chunk.  D layout of the 16x16x4 MFMA: register r of lane l = D[row 4 (l >> 4) + r][col l & 15].
typedef float floatx4v __attribute__((ext_vector_type(4)));

// KT2 = 16-row MFMA tiles per wave: 2 -> block = 2 images x 64 out-channels (wave = image wp, channel half wk);
//       1 -> block = 1 image x 64 out-channels (wave = channel quarter): twice the waves for layers with few out-channels
// (LDS floats of a block: the two chunk buffers + the 64 bias values; body = device function over (LDS base, block index), as
// wino_conv16g_body below: the kernel of its own and one half of wino_pair_kernel are the same code)
template <bool UNPOOL, int KT2>
__host__ __device__ constexpr int wino16_lds_floats() {
    return 2 * ((CLHIP_W16_ADIRECT != 0 && !(UNPOOL && KT2 == 2) ? 0 : W_FLOATS) + WCK * KT2 * 100) + WKT;
}

template <int MODE, bool UNPOOL, int KT2>
__device__ __forceinline__ void wino_conv16_body(
    float* __restrict__ lds, const int bid,
    const float* __restrict__ in, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int relu) {
    constexpr int H = 8, W = 8, NIMG = KT2;
    constexpr int PW = 10, PR = 10, PLANE = NIMG * PR * PW;           // halo planes of the block's two images, per channel
    // CLHIP_W16_ADIRECT: A operands from the lane-ordered image of U straight into registers (as wino_conv16g_kernel); LDS = halo planes
    // (two blocks per CU by registers then, which keeps the operands out of the AGPR shuffle a 512-register budget invites; the
    // un-pooling instance with 32-channel waves needs 256 + registers that way and stays on the LDS path: measured 36.2 vs 37.4 us)
    constexpr bool ADIRECT = CLHIP_W16_ADIRECT != 0 && !(UNPOOL && KT2 == 2);
    static_assert(!ADIRECT || CLHIP_W16G_ADIRECT != 0, "the lane-ordered image of U is written only with CLHIP_W16G_ADIRECT");
    constexpr int WOFF = ADIRECT ? 0 : W_FLOATS;
    constexpr int X_FLOATS = WCK * PLANE, BUF = WOFF + X_FLOATS;
    static_assert(wino16_lds_floats<UNPOOL, KT2>() == 2 * BUF + WKT, "LDS size of the wrappers");
    float* const bias_s = lds + 2 * BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = KT2 == 2 ? wave & 1 : wave, wp = KT2 == 2 ? wave >> 1 : 0;      // wk: 32- (KT2 = 2) or 16-channel slice of the block's 64
    constexpr int KW = 16 * KT2;                                                     // out channels per wave
    const int ti = lane & 15, q = lane >> 4;
    const int n_grp = (N + NIMG - 1) / NIMG;
    const int kt = bid / n_grp, ng = bid - kt * n_grp;
    const int n0 = ng * NIMG, ko0 = kt * WKT;
    const int n_chunks = (Cin + WCK - 1) / WCK;
    if (MODE == 0 && tid < WKT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;

    constexpr int Hi = UNPOOL ? H >> 1 : H, Wi = UNPOOL ? W >> 1 : W;
    constexpr int plane_in = Hi * Wi;
    const float* in_blk = in + (size_t)n0 * Cin * plane_in;
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_blk, (size_t)(N - n0) * Cin * plane_in * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? pool_idx + (size_t)n0 * Cin * plane_in : pool_idx,
                                                   UNPOOL ? (size_t)(N - n0) * Cin * plane_in : 0);
    const __amdgpu_buffer_rsrc_t rs_u = ADIRECT
        ? clhip_rsrc(U + (size_t)((Cout + WKT - 1) / WKT) * n_chunks * W_FLOATS + (size_t)kt * 2 * n_chunks * WD_FLOATS,
                     (size_t)2 * n_chunks * WD_FLOATS * sizeof(float))
        : clhip_rsrc(U + (size_t)kt * n_chunks * W_FLOATS, (size_t)n_chunks * W_FLOATS * sizeof(float));

    // ---- staging units (as above: one instruction each, weights = straight 16-byte copy, activations = one scalar per halo element)
    constexpr int W_IT = ADIRECT ? 0 : W_FLOATS / 4 / 256;           // 10
    constexpr int X_IT = (X_FLOATS + 255) / 256;                     // 7
    constexpr int NU = W_IT + X_IT;
    float4 wv[W_IT ? W_IT : 1];
    float xr[X_IT];
    unsigned xi[UNPOOL ? X_IT : 1];
    int xoff[X_IT], xcode[UNPOOL ? X_IT : 1];
#pragma unroll
    for (int j = 0; j < X_IT; ++j) {
        const int e = tid + 256 * j;
        xoff[j] = CLHIP_OOB;
        if (e < X_FLOATS) {
            const int cl = e / PLANE, rem = e - cl * PLANE;
            const int col = rem % PW, rr = rem / PW;
            const int row = rr % PR, nb = rr / PR;
            const int h = row - 1, w = col - 1;
            if (n0 + nb < N && h >= 0 && h < H && w >= 0 && w < W) {
                if constexpr (UNPOOL) {
                    xoff[j] = (nb * Cin + cl) * plane_in + (h >> 1) * Wi + (w >> 1);     // ELEMENT offset
                    xcode[j] = ((h & 1) << 1) | (w & 1);
                } else {
                    xoff[j] = ((nb * Cin + cl) * plane_in + h * W + w) * 4;
                }
            }
        }
    }
    auto load_unit = [&](int u, int chunk) {
        const int cw = chunk < n_chunks ? chunk : n_chunks - 1;
        if (u < W_IT) {
            wv[u] = clhip_buf_load4(rs_u, (tid + 256 * u) * 16, cw * W_FLOATS * 4);
        } else {
            const int j = u - W_IT;
            const int xb = cw * WCK * plane_in;
            if constexpr (UNPOOL) {
                xr[j] = clhip_buf_load(rs_x, xoff[j] != CLHIP_OOB ? xoff[j] * 4 : CLHIP_OOB, xb * 4);
                xi[j] = clhip_buf_load_u8(rs_i, xoff[j], xb);
            } else {
                xr[j] = clhip_buf_load(rs_x, xoff[j], xb * 4);
            }
        }
    };
    auto store_unit = [&](int u, int bo) {
        if (u < W_IT) {
            float* wd = lds + bo + 4 * (tid + 256 * u);
            *reinterpret_cast<floatx4*>(wd) = floatx4{wv[u].x, wv[u].y, wv[u].z, wv[u].w};
        } else {
            const int j = u - W_IT;
            float* xs = lds + bo + WOFF;
            if (256 * (j + 1) <= X_FLOATS || tid + 256 * j < X_FLOATS) {
                if constexpr (UNPOOL) xs[tid + 256 * j] = ((int)xi[j] == xcode[j]) ? xr[j] : 0.f;
                else xs[tid + 256 * j] = xr[j];
            }
        }
    };

    // this lane's tile: image wp of the block, tile (t_row, t_col) = (ti >> 2, ti & 3); channel q of a quad
    const int t_row = ti >> 2, t_col = ti & 3;
    const int d_off = WOFF + q * PLANE + (wp * PR + 2 * t_row) * PW + 2 * t_col;            // even: 8-byte aligned
    const int a_off = (q * WKT + wk * KW + ti) * WFP;                                      // + 16 * WFP: second row tile

    floatx4v acc[KT2][16];
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[k2][f] = floatx4v{0.f, 0.f, 0.f, 0.f};

    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NQ = WCK / 4;                                      // channel quads per chunk
    floatx4 av[2][KT2][4];         // [pipeline slot][row tile][frequency quad]
    float vv[2][16];
    f2 dlo[4], dhi[4], tlo[4], thi[4];
    // c4 = the 4-channel chunk the operands belong to (ADIRECT: piece (out-channel sixteen, frequency quad) of that chunk, 1 KB per wave)
    auto rd_a = [&](const float* abase, int quad, int k2, int fq, int slot, int c4) {
        if constexpr (ADIRECT) {
            const int cc = c4 < 2 * n_chunks ? c4 : 2 * n_chunks - 1;
            const clhip_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (((wk * KW) >> 4) + k2) * 4096 + fq * 1024 + lane * 16,
                                                                        cc * (WD_FLOATS * 4), 0);
            av[slot][k2][fq] = floatx4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        } else {
            av[slot][k2][fq] = *reinterpret_cast<const floatx4*>(abase + 4 * quad * WKT * WFP + k2 * 16 * WFP + 4 * fq);
        }
    };
    auto rd_d = [&](const float* dbase, int quad, int r) {
        dlo[r] = *reinterpret_cast<const f2*>(dbase + 4 * quad * PLANE + r * PW);
        dhi[r] = *reinterpret_cast<const f2*>(dbase + 4 * quad * PLANE + r * PW + 2);
    };
    auto row_tf = [&](int i) {
        if (i == 0) { tlo[0] = dlo[0] - dlo[2]; thi[0] = dhi[0] - dhi[2]; }
        if (i == 1) { tlo[1] = dlo[1] + dlo[2]; thi[1] = dhi[1] + dhi[2]; }
        if (i == 2) { tlo[2] = dlo[2] - dlo[1]; thi[2] = dhi[2] - dhi[1]; }
        if (i == 3) { tlo[3] = dlo[1] - dlo[3]; thi[3] = dhi[1] - dhi[3]; }
    };
    auto col_tf = [&](int i, int slot) {
        const f2 o = tlo[i] - thi[i];
        vv[slot][4 * i + 0] = o.x;
        vv[slot][4 * i + 3] = o.y;
        vv[slot][4 * i + 1] = tlo[i].y + thi[i].x;
        vv[slot][4 * i + 2] = thi[i].x - tlo[i].y;
    };

#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) store_unit(u, 0);
#pragma unroll
    for (int u = 0; u < NU; ++u) load_unit(u, 1);
    __syncthreads();
#pragma unroll
    for (int fq = 0; fq < 4; ++fq) { rd_a(lds + a_off, 0, 0, fq, 0, 0); if (KT2 == 2) rd_a(lds + a_off, 0, KT2 - 1, fq, 0, 0); }
#pragma unroll
    for (int r = 0; r < 4; ++r) rd_d(lds + d_off, 0, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) row_tf(i);
#pragma unroll
    for (int i = 0; i < 4; ++i) col_tf(i, 0);

    static_assert(NQ == 2 && NU <= 32 && (KT2 == 1 || KT2 == 2), "stores: two units per slot of quad 0; loads: up to two units per slot of quad 1");
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int bo = (chunk & 1) * BUF, bn = BUF - bo;
        const float* a_cur = lds + bo + a_off;
        const float* d_cur = lds + bo + d_off;
        const float* a_nxt = lds + bn + a_off;
        const float* d_nxt = lds + bn + d_off;
#pragma unroll
        for (int p = 0; p < NQ; ++p) {
            const float* ab = (p + 1 < NQ) ? a_cur : a_nxt;
            const float* db = (p + 1 < NQ) ? d_cur : d_nxt;
            const int np = (p + 1) % NQ, ns = (p + 1) & 1, cs = p & 1;
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cs][0][f >> 2][f & 3], vv[cs][f], acc[0][f], 0, 0, 0);
                if (KT2 == 2) acc[KT2 - 1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cs][KT2 - 1][f >> 2][f & 3], vv[cs][f], acc[KT2 - 1][f], 0, 0, 0);
                if (p == NQ - 1 && f == 0) __syncthreads();
                if (f < 4) { rd_d(db, np, f); rd_a(ab, np, 0, f, ns, 2 * chunk + p + 1); }
                else if (f < 8) { if (KT2 == 2) rd_a(ab, np, KT2 - 1, f - 4, ns, 2 * chunk + p + 1); }
                else if (f < 12) row_tf(f - 8);
                else col_tf(f - 12, ns);
                if (p == 0) {
                    if (2 * f < NU) store_unit(2 * f, bn);
                    if (2 * f + 1 < NU) store_unit(2 * f + 1, bn);
                } else {
                    if (f < NU) load_unit(f, chunk + 2);
                    if (16 + f < NU) load_unit(16 + f, chunk + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: register r of acc[k2][f] = (out channel ko0 + 32 wk + 16 k2 + 4 q + r, this lane's tile)
    const int n = n0 + wp, oh = 2 * t_row, ow = 2 * t_col;
    const bool tile_ok = n < N;
    const bool pool = MODE == 0 && pool_idx != nullptr;
    // output stores with the output cache policy (common.hpp) while 32-bit byte offsets reach the whole tensor, plain stores beyond
    const bool st32 = (size_t)N * Cout * (pool ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * sizeof(float) <= 0x7fffffffull;
    const __amdgpu_buffer_rsrc_t rs_out = clhip_out_rsrc(out), rs_code = clhip_out_rsrc(pool_idx);
    auto st1 = [&](size_t o, float v) { if (st32) clhip_buf_store(v, rs_out, (int)o * 4, 0); else out[o] = v; };
    auto st2 = [&](size_t o, float a, float b) {
        if (st32) clhip_buf_store2(make_float2(a, b), rs_out, (int)o * 4, 0); else *reinterpret_cast<float2*>(out + o) = make_float2(a, b);
    };
    auto stc = [&](size_t o, int a) { if (st32) clhip_buf_store_u8((uint8_t)a, rs_code, (int)o, 0); else pool_idx[o] = (uint8_t)a; };
    constexpr int chw = H * W, OH = H >> 1, OW = W >> 1;
#pragma unroll
    for (int k2 = 0; k2 < KT2; ++k2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u0[4], u1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u0[j] = acc[k2][j][r] + acc[k2][4 + j][r] + acc[k2][8 + j][r];
            u1[j] = acc[k2][4 + j][r] - acc[k2][8 + j][r] - acc[k2][12 + j][r];
        }
        float y00 = u0[0] + u0[1] + u0[2], y01 = u0[1] - u0[2] - u0[3];
        float y10 = u1[0] + u1[1] + u1[2], y11 = u1[1] - u1[2] - u1[3];
        const int kl = wk * KW + k2 * 16 + 4 * q + r;
        const int k = ko0 + kl;
        const bool ok = tile_ok && k < Cout;
        if (MODE == 0) {
            const float b = bias_s[kl];
            y00 += b; y01 += b; y10 += b; y11 += b;
            if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            if (pool) {
                float m = y00; int a = 0;
                if (y01 > m) { m = y01; a = 1; }
                if (y10 > m) { m = y10; a = 2; }
                if (y11 > m) { m = y11; a = 3; }
                if (relu && !(m > 0.f)) a = CLHIP_POOL_DEAD;
                if (ok) {
                    const size_t o = ((size_t)n * Cout + k) * OH * OW + (size_t)(oh >> 1) * OW + (ow >> 1);
                    st1(o, m);
                    stc(o, a);
                }
                continue;
            }
        }
        if (ok) {
            const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
            if (MODE == 1 && mask_src) {
                const float2 m0 = *reinterpret_cast<const float2*>(mask_src + o), m1 = *reinterpret_cast<const float2*>(mask_src + o + W);
                y00 = m0.x > 0.f ? y00 : 0.f; y01 = m0.y > 0.f ? y01 : 0.f;
                y10 = m1.x > 0.f ? y10 : 0.f; y11 = m1.y > 0.f ? y11 : 0.f;
            }
            st2(o, y00, y01);
            st2(o + W, y10, y11);
        }
    }
}

template <int MODE, bool UNPOOL, int KT2>
__global__ __launch_bounds__(256, (CLHIP_W16_ADIRECT && !(UNPOOL && KT2 == 2)) ? 2 : 1) void wino_conv16_kernel(
    const float* __restrict__ in, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int relu) {
    __shared__ __attribute__((aligned(16))) float lds[wino16_lds_floats<UNPOOL, KT2>()];
    wino_conv16_body<MODE, UNPOOL, KT2>(lds, (int)blockIdx.x, in, U, bias, mask_src, out, pool_idx, N, Cin, Cout, relu);
}


// ------------------------------------------------------------------------------------------------ 16-tile waves, two blocks per CU
// The 32-tile kernel keeps 256 accumulators per wave, so ONE wave runs per SIMD and nothing covers its prologue (first
// global loads), its barriers and its epilogue (output transform, bias / ReLU / pool, stores): on a 64-channel layer (8 chunks)
// those are about a third of a unit's time.  This kernel has the 16x16x4-MFMA wave of wino_conv16_kernel on ANY even map: a wave
// is TC x TR = 16 tiles x 32 out-channels (128 accumulators), a block 64 out-channels x (TC x 2 TR) tiles of one image, LDS
// holds 4-channel chunks (23 KB, double-buffered) and the kernel stays under 256 registers — two blocks share a CU, each
// SIMD has two waves, one issues MFMAs while the other waits, transforms or stores.  No hand-placed slots: the work of a
// chunk is read operands -> transform -> 32 MFMAs, the compiler orders it, the second wave fills the gaps.
// Geometry: the block's TC x 2 TR tiles are 2 TR / EROWS ENTITIES of EROWS tile rows each, every entity with its own halo band
// of 2 EROWS + 2 rows; entities are numbered over (image, row group of the image):
//   EROWS = 2 TR      one entity = a (TC x 2 TR)-tile region of one image                     (even maps >= 16 wide: <8, 2, 4>)
//   EROWS = TR = 4    two entities = two whole 8 x 8 images                                   (<4, 4, 4>)
//   EROWS = 1         four consecutive tile rows of the (image, tile row) list, odd maps 9..16 wide (AlexNet's 13 x 13: <8, 2, 1>;
//                     49 of 56 tile slots busy; half-outside tiles read 0 and store nothing, 4-byte stores)
// LDS floats of a block: the two chunk buffers, then the 64 bias values.  The kernel body is a device function over (LDS base, block
// index) so that the SAME code is both the launch of its own (wino_conv16g_kernel) and one half of a merged grid (wino_pair_kernel,
// below): the __global__ wrappers own the LDS array.
template <int TC, int TR, int EROWS>
__host__ __device__ constexpr int wino16g_lds_floats() {
    return 2 * ((CLHIP_W16G_ADIRECT != 0 ? 0 : 4 * WKT * WFP) + 4 * (2 * TR / EROWS) * (2 * EROWS + 2) * (2 * TC + 2)) + WKT;
}

template <int TC, int TR, int EROWS, int MODE, bool UNPOOL>
__device__ __forceinline__ void wino_conv16g_body(
    float* __restrict__ lds, const int bid,
    const float* __restrict__ in, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int H, int W, int relu, int tiles_w, int tiles_h, int n_pix_blocks) {
    static_assert(TC * TR == 16, "16 tiles per wave");
    constexpr int CQ = 4;                                             // in channels per chunk = one MFMA k-quad
    constexpr int NE = 2 * TR / EROWS;                                // entities per block
    static_assert(NE * EROWS == 2 * TR && (NE == 1 || !UNPOOL || EROWS == 4), "whole entities");
    constexpr int PW = 2 * TC + 2, PRE = 2 * EROWS + 2, PLANE = NE * PRE * PW;
    constexpr int WQ_FLOATS = CQ * WKT * WFP;                         // 5120 floats: half of a U chunk (channels are its outer index)
    constexpr bool ADIRECT = CLHIP_W16G_ADIRECT != 0;                 // A operands from L2 straight into registers: LDS holds the halo planes only
    constexpr int WOFF = ADIRECT ? 0 : WQ_FLOATS;
    constexpr int X_FLOATS = CQ * PLANE, BUF = WOFF + X_FLOATS;
    static_assert(wino16g_lds_floats<TC, TR, EROWS>() == 2 * BUF + WKT, "LDS size of the wrappers");
    float* const bias_s = lds + 2 * BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave & 1, wp = wave >> 1;
    const int ti = lane & 15, q = lane >> 4;
    const int kt = bid / n_pix_blocks, pb = bid - kt * n_pix_blocks;
    // tiles_h = row groups (entities) per image; entity v = (image v / tiles_h, group v % tiles_h); offsets relative to image nb0
    const int bw = pb % tiles_w, v0 = (pb / tiles_w) * NE;
    const int nb0 = v0 / tiles_h, w0 = bw * 2 * TC;
    const int ko0 = kt * WKT;
    const int n_chunks = Cin / CQ;                                    // Cin is a whole number of 8-channel U chunks
    if (MODE == 0 && tid < WKT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;

    const int Hi = UNPOOL ? H >> 1 : H, Wi = UNPOOL ? W >> 1 : W;
    const int plane_in = Hi * Wi;
    const float* in_img = in + (size_t)nb0 * Cin * plane_in;
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_img, (size_t)(N - nb0) * Cin * plane_in * sizeof(float));
    const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? pool_idx + (size_t)nb0 * Cin * plane_in : pool_idx,
                                                   UNPOOL ? (size_t)(N - nb0) * Cin * plane_in : 0);
    // LDS image of this k-tile's chunks, or (ADIRECT) its lane-ordered image behind the LDS images of the whole layer
    const __amdgpu_buffer_rsrc_t rs_u = ADIRECT
        ? clhip_rsrc(U + (size_t)((Cout + WKT - 1) / WKT) * (Cin / WCK) * W_FLOATS + (size_t)kt * n_chunks * WD_FLOATS,
                     (size_t)n_chunks * WD_FLOATS * sizeof(float))
        : clhip_rsrc(U + (size_t)kt * n_chunks * WQ_FLOATS, (size_t)n_chunks * WQ_FLOATS * sizeof(float));

    constexpr int W_IT = ADIRECT ? 1 : WQ_FLOATS / 4 / 256;           // 5 (ADIRECT: unused)
    constexpr int X_IT = (X_FLOATS + 255) / 256;
    struct Stage {                                                    // one chunk on its way from global memory to LDS
        float4 wv[W_IT];
        float xr[X_IT];
        unsigned xi[UNPOOL ? X_IT : 1];
    };
    int xoff[X_IT], xcode[UNPOOL ? X_IT : 1];
#pragma unroll
    for (int j = 0; j < X_IT; ++j) {
        const int e = tid + 256 * j;
        xoff[j] = CLHIP_OOB;
        if (e < X_FLOATS) {
            const int cl = e / PLANE, rem = e - cl * PLANE;
            const int col = rem % PW, rr = rem / PW;
            const int row = rr % PRE, v = v0 + rr / PRE;
            const int nn = v / tiles_h, g = v - nn * tiles_h;
            const int h = g * 2 * EROWS - 1 + row, w = w0 - 1 + col;
            if (nn < N && h >= 0 && h < H && w >= 0 && w < W) {
                if constexpr (UNPOOL) {
                    xoff[j] = ((nn - nb0) * Cin + cl) * plane_in + (h >> 1) * Wi + (w >> 1);       // ELEMENT offset
                    xcode[j] = ((h & 1) << 1) | (w & 1);
                } else {
                    xoff[j] = (((nn - nb0) * Cin + cl) * plane_in + h * W + w) * 4;
                }
            }
        }
    }
    auto load_chunk = [&](int chunk, Stage& st) {
        const int cw = chunk < n_chunks ? chunk : n_chunks - 1;
        if constexpr (!ADIRECT) {
#pragma unroll
            for (int u = 0; u < W_IT; ++u) st.wv[u] = clhip_buf_load4(rs_u, (tid + 256 * u) * 16, cw * WQ_FLOATS * 4);
        }
        const int xb = cw * CQ * plane_in;
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            if constexpr (UNPOOL) {
                st.xr[j] = clhip_buf_load(rs_x, xoff[j] != CLHIP_OOB ? xoff[j] * 4 : CLHIP_OOB, xb * 4);
                st.xi[j] = clhip_buf_load_u8(rs_i, xoff[j], xb);
            } else {
                st.xr[j] = clhip_buf_load(rs_x, xoff[j], xb * 4);
            }
        }
    };
    auto store_chunk = [&](int bo, const Stage& st) {
        if constexpr (!ADIRECT) {
#pragma unroll
            for (int u = 0; u < W_IT; ++u)
                *reinterpret_cast<floatx4*>(lds + bo + 4 * (tid + 256 * u)) = floatx4{st.wv[u].x, st.wv[u].y, st.wv[u].z, st.wv[u].w};
        }
        float* xs = lds + bo + WOFF;
#pragma unroll
        for (int j = 0; j < X_IT; ++j)
            if (256 * (j + 1) <= X_FLOATS || tid + 256 * j < X_FLOATS) {
                if constexpr (UNPOOL) xs[tid + 256 * j] = ((int)st.xi[j] == xcode[j]) ? st.xr[j] : 0.f;
                else xs[tid + 256 * j] = st.xr[j];
            }
    };

    const int t_rowb = wp * TR + ti / TC, t_col = ti % TC;           // this lane's tile inside the block's TC x 2 TR tiles
    const int t_ent = t_rowb / EROWS, t_row = t_rowb % EROWS;        // its entity and its row inside the entity
    const int d_off = WOFF + q * PLANE + (t_ent * PRE + 2 * t_row) * PW + 2 * t_col;           // even: 8-byte aligned
    const int a_off = (q * WKT + wk * 32 + ti) * WFP;

    floatx4v acc[2][16];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[k2][f] = floatx4v{0.f, 0.f, 0.f, 0.f};

    typedef float f2 __attribute__((ext_vector_type(2)));
#if CLHIP_W16G_ABL & 2
    floatx4 abl_a0[4], abl_a1[4];
#pragma unroll
    for (int fq = 0; fq < 4; ++fq) {
        abl_a0[fq] = floatx4{0.5f + fq, 0.25f, -0.5f, 1.f} * (float)(lane + 1);
        abl_a1[fq] = floatx4{0.75f, 0.5f + fq, 1.5f, -1.f} * (float)(lane + 2);
    }
#endif
    // ADIRECT: the A operands of a chunk = 8 x 1 KB pieces of the lane-ordered image (this wave's out-channel half: 2 row tiles x 4
    // frequency quads), each ONE coalesced buffer_load_dwordx4; issued a whole chunk ahead into the register set the previous
    // burst has just released, so that L2 latency (~200-500 cycles) lies under the other set's chunk.
    struct AReg { floatx4 a0[4], a1[4]; };
    const int a_voff = wk * (2 * 4 * 256 * 4) + lane * 16;
    auto load_a = [&](int chunk, AReg& A) {
        const int ca = (chunk < n_chunks ? chunk : n_chunks - 1) * (WD_FLOATS * 4);
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) {
            const clhip_u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_u, a_voff + fq * 1024, ca, 0);
            const clhip_u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(rs_u, a_voff + 4096 + fq * 1024, ca, 0);
            A.a0[fq] = floatx4{__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w)};
            A.a1[fq] = floatx4{__uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
        }
    };
    // one chunk from LDS buffer `bo`: operands, input transform, 32 MFMAs
    auto compute = [&](int bo, const AReg* areg) {
        const float* ab = lds + bo + a_off;
        const float* db = lds + bo + d_off;
        f2 dlo[4], dhi[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dlo[r] = *reinterpret_cast<const f2*>(db + r * PW);
            dhi[r] = *reinterpret_cast<const f2*>(db + r * PW + 2);
        }
        floatx4 a0[4], a1[4];
#pragma unroll
        for (int fq = 0; fq < 4; ++fq) {
#if CLHIP_W16G_ABL & 2
            a0[fq] = abl_a0[fq]; a1[fq] = abl_a1[fq];
#else
            if constexpr (ADIRECT) {
                a0[fq] = areg->a0[fq]; a1[fq] = areg->a1[fq];
            } else {
                a0[fq] = *reinterpret_cast<const floatx4*>(ab + 4 * fq);
                a1[fq] = *reinterpret_cast<const floatx4*>(ab + 16 * WFP + 4 * fq);
            }
#endif
        }
        // V = B^T d B
        f2 tlo[4], thi[4];
        tlo[0] = dlo[0] - dlo[2]; thi[0] = dhi[0] - dhi[2];
        tlo[1] = dlo[1] + dlo[2]; thi[1] = dhi[1] + dhi[2];
        tlo[2] = dlo[2] - dlo[1]; thi[2] = dhi[2] - dhi[1];
        tlo[3] = dlo[1] - dlo[3]; thi[3] = dhi[1] - dhi[3];
        float vv[16];
#if CLHIP_W16G_ABL & 1
#pragma unroll
        for (int i = 0; i < 4; ++i) { vv[4 * i] = dlo[i].x; vv[4 * i + 1] = dlo[i].y; vv[4 * i + 2] = dhi[i].x; vv[4 * i + 3] = dhi[i].y; }
        (void)tlo; (void)thi;
#else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f2 o = tlo[i] - thi[i];
            vv[4 * i + 0] = o.x;
            vv[4 * i + 3] = o.y;
            vv[4 * i + 1] = tlo[i].y + thi[i].x;
            vv[4 * i + 2] = thi[i].x - tlo[i].y;
        }
#endif
#if CLHIP_W16G_BURST
        __builtin_amdgcn_sched_barrier(0);      // A/B: all 16 transformed values in their own registers first, then 32 MFMAs back to back
#endif
        // Round 4, measured per instance (profiles/r04_w16g_priority_variants.txt): fencing the 32 MFMAs of a chunk off from the
        // operand reads / transform in front of them and running them at raised wave priority is 4 - 13 % FASTER on every instance
        // that stages plain activations (forward, backward-data without a fused un-pool: wide_VGG9 359 -> 311, 188 -> 166 us) and
        // 4 - 5 % SLOWER on the instances that rebuild the un-pooled gradient while staging (more VALU in front of the burst) —
        // so it is on for the former only.  The three priority patterns tried (ramp, high in the burst, high while staging) time
        // alike: what matters is the fence, the s_setprio pair costs nothing.
        constexpr int PRIO = UNPOOL ? CLHIP_W16G_PRIO_UNPOOL : CLHIP_W16G_PRIO;
        if constexpr (PRIO == 2) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (PRIO == 3) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if constexpr (PRIO == 1)
            if ((f & 3) == 0) {                   // the further a wave is into its burst, the higher its claim on the matrix pipe
                __builtin_amdgcn_sched_barrier(0);
                if (f == 0) __builtin_amdgcn_s_setprio(0);
                else if (f == 4) __builtin_amdgcn_s_setprio(1);
                else if (f == 8) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
                __builtin_amdgcn_sched_barrier(0);
            }
#if CLHIP_W16G_ABL & 32
            acc[0][f][0] += a0[f >> 2][f & 3] * vv[f];          // (one VALU op instead of the MFMA: keeps operands alive)
            acc[1][f][0] += a1[f >> 2][f & 3] * vv[f];
#else
            acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[f >> 2][f & 3], vv[f], acc[0][f], 0, 0, 0);
            acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[f >> 2][f & 3], vv[f], acc[1][f], 0, 0, 0);
#endif
        }
#if CLHIP_W16G_BURST
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (PRIO == 1 || PRIO == 2) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (PRIO == 3) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);        // staging of the next chunk (LDS writes, global loads, operand reads) goes first
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#if CLHIP_W16G_ABL & 4
#define W16G_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define W16G_SYNC() __syncthreads()
#endif
#if CLHIP_W16G_ABL & 8
#define W16G_STORE(bo, st) ((void)0)
#else
#define W16G_STORE(bo, st) store_chunk(bo, st)
#endif
#if CLHIP_W16G_ABL & 16
#define W16G_LOAD(c, st) ((void)0)
#else
#define W16G_LOAD(c, st) load_chunk(c, st)
#endif
    // Staging pipeline.  The loads of a later chunk must be ISSUED before this chunk's MFMAs and WAITED FOR after them.
    // Left to itself the scheduler (it minimises register pressure, 180 of the 256 registers two waves per SIMD allow) sinks
    // the loads below the MFMAs — issued right before the barrier, waited for right after it: every chunk then pays the whole
    // L2 / HBM latency with nothing of its own wave in flight (the loop ran at 0.42-0.55 of the matrix pipe on every layer
    // width, round 3).  __builtin_amdgcn_sched_barrier(0) behind the issue pins the order.
    //   CLHIP_W16G_PF = 0  the round-3 schedule (A/B reference)
    //                   1  one register set: chunk + 2 is issued at the top of chunk's iteration, stored at the top of the next
    //                   2  two register sets: chunk + 3 issued at the top of chunk's iteration (two iterations in flight)
#if CLHIP_W16G_PF == 2
    Stage sa, sb;
    AReg ra, rb;                                                       // (ADIRECT) A operands of the even / odd chunks
    load_chunk(0, sa);
    store_chunk(0, sa);
    load_chunk(1, sa);
    load_chunk(2, sb);
    if constexpr (ADIRECT) { load_a(0, ra); load_a(1, rb); }
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; chunk += 2) {               // n_chunks is even (Cin is a multiple of 8)
        W16G_STORE(BUF, sa);                                           // chunk + 1 -> buffer 1 (read last in iteration chunk - 1)
        W16G_LOAD(chunk + 3, sa);
        __builtin_amdgcn_sched_barrier(0);
        compute(0, &ra);
        if constexpr (ADIRECT) { load_a(chunk + 2, ra); __builtin_amdgcn_sched_barrier(0); }     // behind the burst that read `ra`
        W16G_SYNC();
        W16G_STORE(0, sb);                                             // chunk + 2 -> buffer 0
        W16G_LOAD(chunk + 4, sb);
        __builtin_amdgcn_sched_barrier(0);
        compute(BUF, &rb);
        if constexpr (ADIRECT) { load_a(chunk + 3, rb); __builtin_amdgcn_sched_barrier(0); }
        W16G_SYNC();
    }
#else
    static_assert(!ADIRECT, "CLHIP_W16G_ADIRECT needs the two-set pipeline (CLHIP_W16G_PF = 2)");
    Stage sa;
    load_chunk(0, sa);
    store_chunk(0, sa);
    load_chunk(1, sa);
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int bo = (chunk & 1) * BUF;
        // chunk + 1 (in registers since the previous iteration) -> the other buffer (its readers finished before the last barrier)
        store_chunk(BUF - bo, sa);
        load_chunk(chunk + 2, sa);
#if CLHIP_W16G_PF == 1
        __builtin_amdgcn_sched_barrier(0);
#endif
        compute(bo, nullptr);
        __syncthreads();
    }
#endif

    // ---- epilogue: register r of acc[k2][f] = (out channel ko0 + 32 wk + 16 k2 + 4 q + r, this lane's tile)
    const int tv = v0 + t_ent, n = tv / tiles_h;
    const int oh = (tv - n * tiles_h) * 2 * EROWS + 2 * t_row, ow = w0 + 2 * t_col;
    const bool tile_ok = n < N && oh < H && ow < W;
    const bool odd = (H | W) & 1;                                   // uniform; the last tile row / column may be half outside
    const bool row1 = oh + 1 < H, col1 = ow + 1 < W;
    const bool pool = MODE == 0 && pool_idx != nullptr;
    // output stores with the output cache policy (common.hpp) while 32-bit byte offsets reach the whole tensor, plain stores beyond
    const bool st32 = (size_t)N * Cout * (pool ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * sizeof(float) <= 0x7fffffffull;
    const __amdgpu_buffer_rsrc_t rs_out = clhip_out_rsrc(out), rs_code = clhip_out_rsrc(pool_idx);
    auto st1 = [&](size_t o, float v) { if (st32) clhip_buf_store(v, rs_out, (int)o * 4, 0); else out[o] = v; };
    auto st2 = [&](size_t o, float a, float b) {
        if (st32) clhip_buf_store2(make_float2(a, b), rs_out, (int)o * 4, 0); else *reinterpret_cast<float2*>(out + o) = make_float2(a, b);
    };
    auto stc = [&](size_t o, int a) { if (st32) clhip_buf_store_u8((uint8_t)a, rs_code, (int)o, 0); else pool_idx[o] = (uint8_t)a; };
    const size_t chw = (size_t)H * W;
    const int OH = H >> 1, OW = W >> 1;
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float u0[4], u1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u0[j] = acc[k2][j][r] + acc[k2][4 + j][r] + acc[k2][8 + j][r];
            u1[j] = acc[k2][4 + j][r] - acc[k2][8 + j][r] - acc[k2][12 + j][r];
        }
        float y00 = u0[0] + u0[1] + u0[2], y01 = u0[1] - u0[2] - u0[3];
        float y10 = u1[0] + u1[1] + u1[2], y11 = u1[1] - u1[2] - u1[3];
        const int kl = wk * 32 + k2 * 16 + 4 * q + r;
        const int k = ko0 + kl;
        const bool ok = tile_ok && k < Cout;
        if (MODE == 0) {
            const float b = bias_s[kl];
            y00 += b; y01 += b; y10 += b; y11 += b;
            if (relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            if (pool) {
                float m = y00; int a = 0;
                if (y01 > m) { m = y01; a = 1; }
                if (y10 > m) { m = y10; a = 2; }
                if (y11 > m) { m = y11; a = 3; }
                if (relu && !(m > 0.f)) a = CLHIP_POOL_DEAD;
                if (ok) {
                    const size_t o = ((size_t)n * Cout + k) * OH * OW + (size_t)(oh >> 1) * OW + (ow >> 1);
                    st1(o, m);
                    stc(o, a);
                }
                continue;
            }
        }
        if (ok && odd) {
            const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
            if (MODE == 1 && mask_src) {
                y00 = mask_src[o] > 0.f ? y00 : 0.f;
                if (col1) y01 = mask_src[o + 1] > 0.f ? y01 : 0.f;
                if (row1) y10 = mask_src[o + W] > 0.f ? y10 : 0.f;
                if (row1 && col1) y11 = mask_src[o + W + 1] > 0.f ? y11 : 0.f;
            }
            st1(o, y00);
            if (col1) st1(o + 1, y01);
            if (row1) st1(o + W, y10);
            if (row1 && col1) st1(o + W + 1, y11);
        } else if (ok) {
            const size_t o = ((size_t)n * Cout + k) * chw + (size_t)oh * W + ow;
            if (MODE == 1 && mask_src) {
                const float2 m0 = *reinterpret_cast<const float2*>(mask_src + o), m1 = *reinterpret_cast<const float2*>(mask_src + o + W);
                y00 = m0.x > 0.f ? y00 : 0.f; y01 = m0.y > 0.f ? y01 : 0.f;
                y10 = m1.x > 0.f ? y10 : 0.f; y11 = m1.y > 0.f ? y11 : 0.f;
            }
            st2(o, y00, y01);
            st2(o + W, y10, y11);
        }
    }
}

template <int TC, int TR, int EROWS, int MODE, bool UNPOOL>
__global__ __launch_bounds__(256, 2) void wino_conv16g_kernel(
    const float* __restrict__ in, const float* __restrict__ U, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int H, int W, int relu, int tiles_w, int tiles_h, int n_pix_blocks) {
    __shared__ __attribute__((aligned(16))) float lds[wino16g_lds_floats<TC, TR, EROWS>()];
    wino_conv16g_body<TC, TR, EROWS, MODE, UNPOOL>(lds, (int)blockIdx.x, in, U, bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu,
                                                   tiles_w, tiles_h, n_pix_blocks);
}

// (measured against the 32-tile kernel on even maps >= 16 wide in round 3: N = 200: 64->64 @32x32 forward 126 -> 109 us,
// backward-data 136 -> 110; 256->256 @16x16 357 -> 317 = 190 TFLOP/s algorithmic)
static bool wino16g_on() { return true; }

// units of the 32-tile kernel a layer has; below this the 8 x 8 variant takes 8 x 8 maps
#ifndef CLHIP_WINO16_BELOW_UNITS
#define CLHIP_WINO16_BELOW_UNITS 640
#endif
constexpr long long WINO16_BELOW_UNITS = CLHIP_WINO16_BELOW_UNITS;


// ------------------------------------------------------------------------------------------------ weight gradient of small layers
// wino_wgrad_kernel gives a block a 64 x 64 (k, c) tile and 256 accumulators per wave: a 64 -> 64 layer on 16 x 16 maps has
// ONE such tile, so all parallelism must come from splitting the tile list 256 ways — 3 stages and one 147 KB slab per block,
// which is why those layers stayed on the direct kernel (whose pixel-split form has the same shape as this one).  Here a block
// owns a 32 x 32 (k, c) tile; its four waves are (2 in-channel halves of 16) x (2 halves of every 16-tile stage); a wave
// reduces its 8 tiles of the stage with two v_mfma_f32_16x16x4_f32 steps into 32 k x 16 c x 16 frequencies = 128 accumulators.
// Four times as many (k, c) tiles -> a quarter of the splits, 4x the stages per block and a 36 KB slab; under 256 registers and
// 64 KB of LDS two blocks share a CU.  At the end the two tile-halves exchange half of their accumulators through LDS (the wave
// of half p finishes row tile p: a two-term sum, order-free), transform G^T M G and write the slab — the [9][K][C] (+ [K])
// format of conv3x3_wgrad.hip, reduced by the same fixed-order launch.
template <int TCS, int TRS>
struct WGeoP {
    static_assert(TCS * TRS == 16 && (TCS & (TCS - 1)) == 0, "16 tiles per stage, tile of a step by shifts");
    static constexpr int KB = 32, CB = 32;                 // (k, c) tile of a block
    static constexpr int DW = 2 * TCS, DR = 2 * TRS;
    static constexpr int DPIX = DR * DW;                   // 64 pixels
    static constexpr int PW = DW + 2, PR = DR + 2;
    static constexpr int PLANE = PR * PW;
#if CLHIP_WGPS_PITCH
    // The operand reads of a step are ds_read_b64 whose 32 lanes of a group are 16 channels (lane stride = the row pitch) x 2
    // neighbouring tiles (2 floats apart); ds_read_b64 banks are (float index) mod 64, so the group is conflict-free when the pitch
    // is 4 mod 64 floats: slot (8 bytes) = 2 * channel + tile, all 32 distinct.  (Round 5's pitches, 66 and PLANE | 2, were odd in
    // 8-byte units — conflict-free over the channels alone, two-way between (channel, tile + 1) and (channel + k, tile): 0.38 of
    // the kernel's LDS cycles were bank conflicts, profiles/r05_i_pmc_summary.txt.)
    static constexpr int LDP = (DPIX + 63 - 4) / 64 * 64 + 4;          // 68
    static constexpr int PLANEP = (PLANE + 63 - 4) / 64 * 64 + 4;      // 132
    static_assert(LDP >= DPIX && PLANEP >= PLANE && LDP % 64 == 4 && PLANEP % 64 == 4, "row pitches of 4 mod 64 floats");
#else
    static constexpr int LDP = DPIX + 2;
    static constexpr int PLANEP = (PLANE % 4 == 2) ? PLANE : PLANE + 2;
    static_assert((LDP / 2) % 2 == 1 && (PLANEP / 2) % 2 == 1, "lane strides must be odd in 8-byte units");
#endif
    static constexpr int DY_FLOATS = KB * LDP, X_FLOATS = CB * PLANEP;
    static constexpr int BUF = DY_FLOATS + X_FLOATS;
    static constexpr int XCH = 4 * 65 * 64;                // exchange area: per wave 64 accumulators + 1 bias sum, 64 lanes
    static constexpr int LDS_FLOATS = 2 * BUF > XCH ? 2 * BUF : XCH;
};

// VEC (maps of whole tiles in width: W % (2 TCS) == 0, 16-byte-aligned tensors): a stage is staged in 16-byte pieces — every halo-plane
// row as its interior float4s (aligned in global memory; their LDS home starts one float behind the halo column, hence four 4-byte
// LDS writes the compiler pairs into ds_write2_b32) plus the two halo columns as scalars, dy rows as float4s (two ds_write_b64) —
// 6 + 2 vector loads and 10 + 4 LDS writes per thread and stage where the scalar form has 16 + 8 and 16 + 8.
template <int TCS, int TRS, bool UNPOOL, bool VEC>
__device__ __forceinline__ void wino_wgrad_ps_body(
    float* __restrict__ lds, const int bid,
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, const uint8_t* __restrict__ unpool_idx,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h, int total_stages, int splits, int c_tiles, size_t slab_stride) {
    using G = WGeoP<TCS, TRS>;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave & 1, wp = wave >> 1;
    const int li = lane & 15, q = lane >> 4;
    const int split = bid % splits, tile = bid / splits;
    const int ct = tile % c_tiles, kt = tile / c_tiles;
    const int k0 = kt * G::KB, c0 = ct * G::CB;
    const int per = total_stages / splits, extra = total_stages % splits;
    const int st_begin = split * per + min(split, extra);
    const int st_end = st_begin + per + (split < extra ? 1 : 0);
    const int Hd = UNPOOL ? H >> 1 : H, Wd = UNPOOL ? W >> 1 : W;
    const int plane_hw = H * W, plane_dy = Hd * Wd;

    // ---- staging: dy 32 k x 64 pixels (pooled: 32 k x 16 pooled pixels + codes, 2x2 windows rebuilt on the way into LDS),
    //      x 32 c x halo plane; every per-unit index is base + j * constant
    constexpr int DQ = UNPOOL ? G::DPIX / 4 : G::DPIX, DKS = 256 / DQ;
    constexpr int DY_IT = G::KB / DKS;                                  // 8 : 2
    constexpr int DWp = UNPOOL ? G::DW / 2 : G::DW;
    constexpr int CPT = 256 / G::PLANE;
    static_assert(CPT >= 1 && G::CB % CPT == 0, "whole passes over the 32 in-channels");
    constexpr int X_IT = G::CB / CPT;
    float dyr[DY_IT], xr[X_IT];
    unsigned dyi[UNPOOL ? DY_IT : 1];
    const int dq = tid % DQ, dkl = tid / DQ;
    const int d_r = dq / DWp, d_c = dq % DWp;
    const int dy_e0 = dkl * plane_dy + d_r * Wd + d_c;
    const int dy_dst0 = UNPOOL ? dkl * G::LDP + 2 * d_r * G::DW + 2 * d_c : dkl * G::LDP + dq;
    const bool x_thr = tid < CPT * G::PLANE;
    const int x_cl = tid / G::PLANE, x_rem = tid - x_cl * G::PLANE;
    const int x_col = x_rem % G::PW, x_row = x_rem / G::PW;
    const int x_e0 = x_cl * plane_hw + x_row * W + x_col;
    const int x_dst0 = G::DY_FLOATS + x_cl * G::PLANEP + x_rem;
    bool dy_ok = false, x_ok = false;
    // VEC units: x float4s — XU per plane, XCPT planes per pass; x halo scalars — HU per plane, HCPT planes per pass; dy float4s
    constexpr int RSEG = G::DW / 4;
    constexpr int XU = G::PR * RSEG, XCPT = 256 / XU, XV_IT = (G::CB + XCPT - 1) / XCPT;            // 24, 10, 4  |  20, 12, 3
    constexpr int HU = G::PR * 2, HCPT = 256 / HU, XH_IT = (G::CB + HCPT - 1) / HCPT;               // 12, 21, 2  |  20, 12, 3
    constexpr int DQV = G::DR * RSEG, DKSV = 256 / DQV, DYV_IT = G::KB / DKSV;                       // 16, 16, 2
    static_assert(!VEC || (DQV * DKSV == 256 && DYV_IT * DKSV == G::KB), "dy float4 units fill the block exactly");
    float4 xv[VEC ? XV_IT : 1], dyv[VEC && !UNPOOL ? DYV_IT : 1];
    float xh[VEC ? XH_IT : 1];
    const int v_cl = tid / XU, v_rem = tid - v_cl * XU, v_row = v_rem / RSEG, v_seg = v_rem - v_row * RSEG;
    const bool v_thr = tid < XCPT * XU;
    const int v_e0 = v_cl * plane_hw + v_row * W + 1 + 4 * v_seg;                   // relative to the halo origin (h0 - 1, w0 - 1)
    const int v_dst0 = G::DY_FLOATS + v_cl * G::PLANEP + v_row * G::PW + 1 + 4 * v_seg;
    const int h_cl = tid / HU, h_rem = tid - h_cl * HU, h_row = h_rem >> 1, h_col = (h_rem & 1) ? G::DW + 1 : 0;
    const bool h_thr = tid < HCPT * HU;
    const int h_e0 = h_cl * plane_hw + h_row * W + h_col;
    const int h_dst0 = G::DY_FLOATS + h_cl * G::PLANEP + h_row * G::PW + h_col;
    const int dv_kl = tid / DQV, dv_q = tid - dv_kl * DQV, dv_r = dv_q / RSEG, dv_seg = dv_q - dv_r * RSEG;
    const int dv_e0 = dv_kl * plane_dy + dv_r * Wd + 4 * dv_seg;
    const int dv_dst0 = dv_kl * G::LDP + dv_r * G::DW + 4 * dv_seg;                 // even: 8-byte aligned
    bool xv_ok = false, xh_ok = false, dv_ok = false;
    __amdgpu_buffer_rsrc_t rs_dy = clhip_rsrc(dy, 0), rs_x = clhip_rsrc(x, 0), rs_di = clhip_rsrc(x, 0);
    auto begin_stage = [&](int st) {
        const bool live = st < st_end;
        const int s = live ? st : st_begin;
        const int bw = s % tiles_w, bh = (s / tiles_w) % tiles_h, n = s / (tiles_w * tiles_h);
        const int h0 = bh * G::DR, w0 = bw * G::DW;
        const int org_dy = UNPOOL ? (h0 >> 1) * Wd + (w0 >> 1) : h0 * W + w0;
        const float* dyb = dy + ((size_t)n * K + k0) * plane_dy + org_dy;
        const long long dy_left = ((long long)(N - n) * K - k0) * plane_dy - org_dy;
        rs_dy = clhip_rsrc(dyb, live && dy_left > 0 ? (size_t)dy_left * 4 : 0);
        if constexpr (UNPOOL) rs_di = clhip_rsrc(unpool_idx + ((size_t)n * K + k0) * plane_dy + org_dy, live && dy_left > 0 ? (size_t)dy_left : 0);
        const long long org_x = (long long)h0 * W + w0 - W - 1;
        const float* xb = x + ((size_t)n * C + c0) * plane_hw + org_x;
        const long long x_left = ((long long)(N - n) * C - c0) * plane_hw - org_x;
        rs_x = clhip_rsrc(xb, live && x_left > 0 ? (size_t)x_left * 4 : 0);
        dy_ok = h0 + (UNPOOL ? 2 * d_r : d_r) < H && w0 + (UNPOOL ? 2 * d_c : d_c) < W;
        const int h = h0 - 1 + x_row, w = w0 - 1 + x_col;
        x_ok = x_thr && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        if constexpr (VEC) {                      // whole tiles in width: the interior columns of a row are all inside the image
            xv_ok = v_thr && (unsigned)(h0 - 1 + v_row) < (unsigned)H;
            xh_ok = h_thr && (unsigned)(h0 - 1 + h_row) < (unsigned)H && (unsigned)(w0 - 1 + h_col) < (unsigned)W;
            dv_ok = h0 + dv_r < H;
        }
    };
    auto load_stage = [&]() {
        if constexpr (VEC && !UNPOOL) {
#pragma unroll
            for (int u = 0; u < DYV_IT; ++u) dyv[u] = clhip_buf_load4(rs_dy, dv_ok ? (dv_e0 + u * DKSV * plane_dy) * 4 : CLHIP_OOB, 0);
        } else {
#pragma unroll
            for (int u = 0; u < DY_IT; ++u) {
                const int e = dy_e0 + u * DKS * plane_dy;
                dyr[u] = clhip_buf_load(rs_dy, dy_ok ? e * 4 : CLHIP_OOB, 0);
                if constexpr (UNPOOL) dyi[u] = clhip_buf_load_u8(rs_di, dy_ok ? e : CLHIP_OOB, 0);
            }
        }
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < XV_IT; ++j) {
                const bool ok = xv_ok && ((j + 1) * XCPT <= G::CB || v_cl + j * XCPT < G::CB);
                xv[j] = clhip_buf_load4(rs_x, ok ? (v_e0 + j * XCPT * plane_hw) * 4 : CLHIP_OOB, 0);
            }
#pragma unroll
            for (int j = 0; j < XH_IT; ++j) {
                const bool ok = xh_ok && ((j + 1) * HCPT <= G::CB || h_cl + j * HCPT < G::CB);
                xh[j] = clhip_buf_load(rs_x, ok ? (h_e0 + j * HCPT * plane_hw) * 4 : CLHIP_OOB, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < X_IT; ++j) xr[j] = clhip_buf_load(rs_x, x_ok ? (x_e0 + j * CPT * plane_hw) * 4 : CLHIP_OOB, 0);
        }
    };
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto store_stage = [&](int bo) {
        if constexpr (VEC && !UNPOOL) {
#pragma unroll
            for (int u = 0; u < DYV_IT; ++u) {
                float* d = lds + bo + dv_dst0 + u * DKSV * G::LDP;
                *reinterpret_cast<f2*>(d) = f2{dyv[u].x, dyv[u].y};
                *reinterpret_cast<f2*>(d + 2) = f2{dyv[u].z, dyv[u].w};
            }
        } else {
#pragma unroll
        for (int u = 0; u < DY_IT; ++u) {
            const float v = dyr[u];
            if constexpr (UNPOOL) {
                const int cde = (int)dyi[u];
                float* d = lds + bo + dy_dst0 + u * DKS * G::LDP;                         // even offset: 8-byte aligned
                *reinterpret_cast<f2*>(d) = f2{cde == 0 ? v : 0.f, cde == 1 ? v : 0.f};
                *reinterpret_cast<f2*>(d + G::DW) = f2{cde == 2 ? v : 0.f, cde == 3 ? v : 0.f};
            } else {
                lds[bo + dy_dst0 + u * DKS * G::LDP] = v;
            }
        }
        }
        if constexpr (VEC) {
            if (v_thr) {
#pragma unroll
                for (int j = 0; j < XV_IT; ++j)
                    if ((j + 1) * XCPT <= G::CB || v_cl + j * XCPT < G::CB) {
                        float* d = lds + bo + v_dst0 + j * XCPT * G::PLANEP;
                        d[0] = xv[j].x; d[1] = xv[j].y; d[2] = xv[j].z; d[3] = xv[j].w;
                    }
            }
            if (h_thr) {
#pragma unroll
                for (int j = 0; j < XH_IT; ++j)
                    if ((j + 1) * HCPT <= G::CB || h_cl + j * HCPT < G::CB) lds[bo + h_dst0 + j * HCPT * G::PLANEP] = xh[j];
            }
        } else if (x_thr) {
#pragma unroll
            for (int j = 0; j < X_IT; ++j) lds[bo + x_dst0 + j * CPT * G::PLANEP] = xr[j];
        }
    };

    floatx4v acc[2][16];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int f = 0; f < 16; ++f) acc[k2][f] = floatx4v{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    const int a_lane = li * G::LDP;                                   // out-channel li (+ 16 rows: the second row tile)
    const int b_lane = G::DY_FLOATS + (wc * 16 + li) * G::PLANEP;     // in-channel 16 wc + li
    // A dY A^T without the negations of rows / columns 3 (folded into the output transform)
    auto dy_tf = [&](const float* p, float (&o)[16], float& bs) {
        const f2 y0 = *reinterpret_cast<const f2*>(p), y1 = *reinterpret_cast<const f2*>(p + G::DW);
        const float a = y0.x, b = y0.y, c = y1.x, d = y1.y;
        bs += (a + b) + (c + d);
        const float r1p = a + c, r1q = b + d, r2p = a - c, r2q = b - d;
        o[0] = a;   o[1] = a + b;     o[2] = a - b;     o[3] = b;
        o[4] = r1p; o[5] = r1p + r1q; o[6] = r1p - r1q; o[7] = r1q;
        o[8] = r2p; o[9] = r2p + r2q; o[10] = r2p - r2q; o[11] = r2q;
        o[12] = c;  o[13] = c + d;    o[14] = c - d;    o[15] = d;
    };

    if (st_begin < st_end) {
        begin_stage(st_begin);
        load_stage();
        store_stage(0);
        begin_stage(st_begin + 1);
        load_stage();
        __syncthreads();
    }
    for (int st = st_begin; st < st_end; ++st) {
        const int bo = ((st - st_begin) & 1) * G::BUF;
        store_stage(G::BUF - bo);            // stage st + 1 (loaded during the previous iteration) -> the other buffer
        begin_stage(st + 2);
        load_stage();
        const float* cur = lds + bo;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int t = 4 * (2 * wp + s2) + q, tr = t / TCS, tc = t % TCS;      // this lane's tile of the step
            const int doff = 2 * tr * G::DW + 2 * tc, xoff_s = 2 * tr * G::PW + 2 * tc;
            f2 xlo[4], xhi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                xlo[r] = *reinterpret_cast<const f2*>(cur + b_lane + xoff_s + r * G::PW);
                xhi[r] = *reinterpret_cast<const f2*>(cur + b_lane + xoff_s + r * G::PW + 2);
            }
            f2 tlo[4], thi[4];
            tlo[0] = xlo[0] - xlo[2]; thi[0] = xhi[0] - xhi[2];
            tlo[1] = xlo[1] + xlo[2]; thi[1] = xhi[1] + xhi[2];
            tlo[2] = xlo[2] - xlo[1]; thi[2] = xhi[2] - xhi[1];
            tlo[3] = xlo[1] - xlo[3]; thi[3] = xhi[1] - xhi[3];
            float vv[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f2 o = tlo[i] - thi[i];
                vv[4 * i + 0] = o.x;
                vv[4 * i + 3] = o.y;
                vv[4 * i + 1] = tlo[i].y + thi[i].x;
                vv[4 * i + 2] = thi[i].x - tlo[i].y;
            }
#if CLHIP_WGPS_FENCE
            // round 4: both transformed dy tiles first, then the 32 MFMAs of the step fenced off at raised priority (-3 % on the
            // un-pooling instances, neutral on the others: profiles/r04_w16g_priority_variants.txt)
            float av0[16], av1[16];
            dy_tf(cur + a_lane + doff, av0, bsum[0]);
            dy_tf(cur + a_lane + 16 * G::LDP + doff, av1, bsum[1]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(3);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[f], vv[f], acc[0][f], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[f], vv[f], acc[1][f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
#else
            float av[16];
            dy_tf(cur + a_lane + doff, av, bsum[0]);
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[0][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f], vv[f], acc[0][f], 0, 0, 0);
            dy_tf(cur + a_lane + 16 * G::LDP + doff, av, bsum[1]);
#pragma unroll
            for (int f = 0; f < 16; ++f) acc[1][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[f], vv[f], acc[1][f], 0, 0, 0);
#endif
        }
        __syncthreads();
    }

    // ---- the two tile-halves of an in-channel half exchange: wave (wc, wp) keeps row tile wp and receives the partner's part of it
    {
        float* mine = lds + wave * (65 * 64);
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(4 * f + r) * 64 + lane] = wp == 0 ? acc[1][f][r] : acc[0][f][r];
        mine[64 * 64 + lane] = wp == 0 ? bsum[1] : bsum[0];
    }
    __syncthreads();
    floatx4v fin[16];
    float bfin;
    {
        const float* theirs = lds + (wc + 2 * (1 - wp)) * (65 * 64);
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) fin[f][r] = (wp == 0 ? acc[0][f][r] : acc[1][f][r]) + theirs[(4 * f + r) * 64 + lane];
        bfin = (wp == 0 ? bsum[0] : bsum[1]) + theirs[64 * 64 + lane];
    }

    // ---- output transform dW = G^T M' G: register r of fin[f] of lane (li, q) = (k = k0 + 16 wp + 4 q + r, c = c0 + 16 wc + li)
    float* slab = part + (size_t)split * slab_stride;
    const __amdgpu_buffer_rsrc_t rs_slab = clhip_out_rsrc(slab);          // (output cache policy: common.hpp)
    const int cidx = c0 + wc * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) m[i][j] = ((i == 3) != (j == 3)) ? -fin[4 * i + j][r] : fin[4 * i + j][r];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s12 = 0.5f * (m[1][j] + m[2][j]);
            t[0][j] = m[0][j] + s12;
            t[1][j] = 0.5f * (m[1][j] - m[2][j]);
            t[2][j] = s12 + m[3][j];
        }
        const int k = k0 + wp * 16 + 4 * q + r;
        if (k < K && cidx < C) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float s12 = 0.5f * (t[a][1] + t[a][2]);
                clhip_buf_store(t[a][0] + s12, rs_slab, (int)(((3 * a + 0) * K + k) * C + cidx) * 4, 0);
                clhip_buf_store(0.5f * (t[a][1] - t[a][2]), rs_slab, (int)(((3 * a + 1) * K + k) * C + cidx) * 4, 0);
                clhip_buf_store(s12 + t[a][3], rs_slab, (int)(((3 * a + 2) * K + k) * C + cidx) * 4, 0);
            }
        }
    }
    if (ct == 0 && wc == 0) {            // bias sums: lanes of one li hold four tiles' partial sums of out-channel 16 wp + li
        bfin += __shfl_xor(bfin, 16, 64);
        bfin += __shfl_xor(bfin, 32, 64);
        const int k = k0 + wp * 16 + li;
        if (q == 0 && k < K) clhip_buf_store(bfin, rs_slab, (int)(9 * K * C + k) * 4, 0);
    }
}

template <int TCS, int TRS, bool UNPOOL, bool VEC>
__global__ __launch_bounds__(256, 2) void wino_wgrad_ps_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part, const uint8_t* __restrict__ unpool_idx,
    int N, int C, int K, int H, int W, int tiles_w, int tiles_h, int total_stages, int splits, int c_tiles, size_t slab_stride) {
    __shared__ __attribute__((aligned(16))) float lds[WGeoP<TCS, TRS>::LDS_FLOATS];
    wino_wgrad_ps_body<TCS, TRS, UNPOOL, VEC>(lds, (int)blockIdx.x, x, dy, part, unpool_idx, N, C, K, H, W, tiles_w, tiles_h,
                                              total_stages, splits, c_tiles, slab_stride);
}

static bool wgrad_ps_on() { return true; }       // (small layers: the pixel-split Winograd weight gradient, wino_wgrad_ps_kernel)

// ------------------------------------------------------------------------------------------------ backward of a small layer as ONE grid
// The deep layers of the VGG9s (16 x 16 and 8 x 8 maps at N = 200) give every launch of their backward pass 200 - 512 blocks: ONE
// round of blocks on 256 CUs x 2, all of them in the same phase at the same time — every block loads its first operands together,
// runs its MFMAs together, stores together — and the matrix pipe idles through the memory phases (0.17 - 0.35 of the pipe issued,
// DESIGN 6).  Backward-data and the weight gradient of a layer read the same dy and do not depend on each other.  Issued on two
// streams WITHOUT events between them the two launches of a layer take 20 - 35 % less than one after the other
// (profiles/r05_s_coresident_pair.txt: blocks of different phase share the CUs); with the events a plan needs the gain is gone
// (CLHIP_WGRAD_OVERLAP=2, round 4: slower).  Here the two launches are one grid: block b is a block of wino_conv16g_kernel
// (MODE 1) or a block of wino_wgrad_ps_kernel, the same device code as the two kernels of their own (their bodies are device
// functions over an LDS base and a block index), results bit-identical to the two launches.
// Order of the blocks: the weight-gradient list first, then the backward-data list.  Measured on the four deep layers of small_VGG9
// (N = 200, rocprofv3, profiles/r05_v_pair_orders.txt; the two launches of their own: 65.0 / 61.0 / 71.2 / 41.7 us): this order 58.6 /
// 53.4 / 55.2 / 30.4 us, backward-data first 60.7 / 56.2 / 53.7 / 30.7, the two kinds alternating in groups of 8 blocks (one per
// XCD) 61.8 / 58.2 / 55.5 / 30.7.
// DK: the backward-data half is a block of wino_conv16g_kernel<TC, TR, EROWS, 1, UNPOOL> (0) or, on 8 x 8 maps of layers with few
// in-channels, of wino_conv16_kernel<1, UNPOOL, 1> (1: one image x 64 channels per block, as launch_wino chooses on its own)
template <int DK, int TC, int TR, int EROWS, bool UNPOOL, int TCS, int TRS>
__global__ __launch_bounds__(256, 2) void wino_pair_kernel(
    // backward-data half (arguments of wino_conv16g_kernel<TC, TR, EROWS, 1, UNPOOL>)
    const float* __restrict__ d_in, const float* __restrict__ d_U, const float* __restrict__ d_mask, float* __restrict__ d_out,
    uint8_t* __restrict__ d_idx, int N, int d_Cin, int d_Cout, int H, int W, int d_tiles_w, int d_tiles_h, int d_npb,
    // weight-gradient half (arguments of wino_wgrad_ps_kernel<TCS, TRS, UNPOOL, true>)
    const float* __restrict__ w_x, float* __restrict__ w_part, int w_tiles_w, int w_tiles_h, int w_total, int w_splits, int w_ctiles,
    size_t w_slab, int nb_w) {
    constexpr int LD = DK ? wino16_lds_floats<UNPOOL, 1>() : wino16g_lds_floats<TC, TR, EROWS>(), LW = WGeoP<TCS, TRS>::LDS_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LD > LW ? LD : LW];
    const int b = (int)blockIdx.x;
    // weight-gradient blocks first (the longer ones), then the backward-data blocks
    const int kind = b < nb_w ? 1 : 0, idx = kind ? b : b - nb_w;
    if (kind == 0) {
        if constexpr (DK)
            wino_conv16_body<1, UNPOOL, 1>(lds, idx, d_in, d_U, nullptr, d_mask, d_out, d_idx, N, d_Cin, d_Cout, 0);
        else
            wino_conv16g_body<TC, TR, EROWS, 1, UNPOOL>(lds, idx, d_in, d_U, nullptr, d_mask, d_out, d_idx, N, d_Cin, d_Cout, H, W, 0,
                                                        d_tiles_w, d_tiles_h, d_npb);
    } else
        wino_wgrad_ps_body<TCS, TRS, UNPOOL, true>(lds, idx, w_x, d_in, w_part, d_idx, N, d_Cout, d_Cin, H, W, w_tiles_w, w_tiles_h,
                                                   w_total, w_splits, w_ctiles, w_slab);
}

template <int MODE, bool UNPOOL>
int launch_wino(const float* in, const float* U, const float* bias, const float* mask_src, float* out, uint8_t* pool_idx,
                int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    const int kts = (Cout + WKT - 1) / WKT;
    if (H == 8 && W == 8 && ((long long)N * 16 + 31) / 32 * ((Cout + 31) / 32) < WINO16_BELOW_UNITS) {
        // waves of 32 channels x one image; layers with few out-channels take 16-channel waves instead (twice as many)
        const bool narrow = (long long)N * ((Cout + 31) / 32) <= 512;
        const long long blocks = (long long)(narrow ? N : (N + 1) / 2) * kts;
        if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;
        if (narrow)
            hipLaunchKernelGGL((wino_conv16_kernel<MODE, UNPOOL, 1>), dim3((unsigned)blocks), dim3(256), 0, s, in, U, bias, mask_src, out,
                               pool_idx, N, Cin, Cout, relu);
        else
            hipLaunchKernelGGL((wino_conv16_kernel<MODE, UNPOOL, 2>), dim3((unsigned)blocks), dim3(256), 0, s, in, U, bias, mask_src, out,
                               pool_idx, N, Cin, Cout, relu);
        CLHIP_LAUNCH_CHECK();
        return 0;
    }
    if (wino16g_on()) {
        // 16-tile waves, two blocks per CU (wino_conv16g_kernel): every shape of this path except 8 x 8 maps with so few units
        // that the one-image waves of wino_conv16_kernel (above) spread better
#define WINO_16G(TC_, TR_, ER_, UNP_)                                                                                          \
        do {                                                                                                                   \
            const int tiles_w = ((W + 1) / 2 + TC_ - 1) / TC_, groups = ((H + 1) / 2 + ER_ - 1) / ER_;                          \
            constexpr int NE_ = 2 * TR_ / ER_;                                                                                  \
            const long long npb = (((long long)N * groups + NE_ - 1) / NE_) * tiles_w, blocks = npb * kts;                      \
            if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;                                                      \
            hipLaunchKernelGGL((wino_conv16g_kernel<TC_, TR_, ER_, MODE, UNP_>), dim3((unsigned)blocks), dim3(256), 0, s, in, U, \
                               bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, tiles_w, groups, (int)npb);             \
            CLHIP_LAUNCH_CHECK();                                                                                               \
            return 0;                                                                                                           \
        } while (0)
        if ((H | W) & 1) { if (W <= 16) WINO_16G(8, 2, 1, false); }
        else if (W >= 16) WINO_16G(8, 2, 4, UNPOOL);
        else if (H == 8 && W == 8) WINO_16G(4, 4, 4, UNPOOL);
#undef WINO_16G
    }
#define WINO_GEO(TCB_, TRB_, NIMG_)                                                                                          \
    do {                                                                                                                      \
        const int tiles_w = ((W + 1) / 2 + TCB_ - 1) / TCB_, tiles_h = ((H + 1) / 2 + TRB_ - 1) / TRB_, ngrp = (N + NIMG_ - 1) / NIMG_; \
        const long long npb = (long long)tiles_w * tiles_h * ngrp, blocks = npb * kts;                                        \
        if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;                                                        \
        hipLaunchKernelGGL((wino_conv_kernel<TCB_, TRB_, NIMG_, MODE, UNPOOL>), dim3((unsigned)blocks), dim3(256), 0, s, in, U, \
                           bias, mask_src, out, pool_idx, N, Cin, Cout, H, W, relu, tiles_w, tiles_h, (int)npb);              \
    } while (0)
    // what is left for the 32-tile kernel: even maps narrower than 16 other than 8 x 8 (the 16-tile kernel above took every other shape
    // of clhip_internal_wino_ok: even maps >= 16 wide, 8 x 8 maps, odd maps up to 16 wide)
    WINO_GEO(4, 4, 4);
#undef WINO_GEO
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// shapes this path takes: channel counts in whole chunks; even H, W (2x2 output tiles, 8-byte stores), or odd maps 9..16 wide
// through the row-packed geometry (no fused pooling there: callers pass pool_idx / unpool only for even maps)
bool clhip_internal_wino_ok(int Cin, int Cout, int H, int W) {
    if (Cin % WCK || Cin < 16 || Cout % 32 || H < 4 || W < 8) return false;
    if (((H | W) & 1) == 0) return true;
    return W > 8 && W <= 16;
}

size_t clhip_internal_wino_ws(int Cin, int Cout) {
    return (size_t)((Cout + WKT - 1) / WKT) * ((Cin + WCK - 1) / WCK) * WU_FLOATS * sizeof(float);
}

// forward (mode 0) / backward-data (mode 1) through the Winograd path.  ws: clhip_internal_wino_ws(Cin, Cout) bytes.
// For backward-data the caller passes Cin = the layer's OUT channels (channels of dy), Cout = its IN channels; `w` is the
// layer's weight tensor [K][C][3][3] in both modes.
// the convolution alone, on weights already transformed into U (clhip_internal_wino_weights or the single-layer kernel)
int clhip_internal_wino_conv_u(int mode, const float* in, const float* U, const float* bias, const float* mask_src, float* out,
                               uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    if (!in || !U || !out || !clhip_internal_wino_ok(Cin, Cout, H, W)) return CLHIP_EINVAL;
    if (((H | W) & 1) && (pool_idx || unpool)) return CLHIP_ENOTSUP;
    if (mode == 0) return launch_wino<0, false>(in, U, bias, nullptr, out, pool_idx, N, Cin, Cout, H, W, relu, s);
    if (unpool) return launch_wino<1, true>(in, U, nullptr, mask_src, out, pool_idx, N, Cin, Cout, H, W, 0, s);
    return launch_wino<1, false>(in, U, nullptr, mask_src, out, nullptr, N, Cin, Cout, H, W, 0, s);
}

// U for several (weights, mode) pairs in one launch (jobs: host array; Ko / Ci are the channel counts as the KERNEL sees them)
int clhip_internal_wino_weights(const clhip_wino_wt* jobs, int n, hipStream_t s) {
    if (n <= 0) return 0;
    if (!jobs) return CLHIP_EINVAL;
    for (int base = 0; base < n; base += WT_JOBS) {
        WtJobs J;
        J.n = n - base < WT_JOBS ? n - base : WT_JOBS;
        J.pad = 0;
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) {
            const clhip_wino_wt& q = jobs[base + i];
            if (!q.w || !q.U || q.Ko <= 0 || q.Ci <= 0) return CLHIP_EINVAL;
            J.j[i] = q;
            J.first[i] = blocks;
            const int total = ((q.Ko + WKT - 1) / WKT) * ((q.Ci + WCK - 1) / WCK) * WCK * WKT;
            blocks += (total + 255) / 256;
        }
        J.first[J.n] = blocks;
        hipLaunchKernelGGL(wino_weight_multi_kernel, dim3(blocks), dim3(256), 0, s, J);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

// Winograd U images (wj) AND bf16-split images (bj) of a pass in ONE launch when they fit one job table; otherwise (or when one kind
// is absent) the launches of their own.  Same device code per job as the two launches: the images hold the same bits.
int clhip_internal_weight_images(const clhip_wino_wt* wj, int nw, const clhip_wino_wt* bj, int nb, hipStream_t s) {
    if (nw <= 0 || nb <= 0 || nw + nb > WT_JOBS) {
        const int rc = clhip_internal_wino_weights(wj, nw, s);
        if (rc) return rc;
        return clhip_internal_bs_weights(bj, nb, s);
    }
    if (!wj || !bj) return CLHIP_EINVAL;
    WtJobs J;
    J.n = nw + nb;
    J.pad = 0;
    int blocks = 0;
    for (int i = 0; i < J.n; ++i) {
        const bool is_bs = i >= nw;
        const clhip_wino_wt& q = is_bs ? bj[i - nw] : wj[i];
        if (!q.w || !q.U || q.Ko <= 0 || q.Ci <= 0) return CLHIP_EINVAL;
        J.j[i] = q;
        J.first[i] = blocks;
        if (is_bs) {
            J.pad |= 1 << i;
            blocks += bs_weight_blocks(q);
        } else {
            const int total = ((q.Ko + WKT - 1) / WKT) * ((q.Ci + WCK - 1) / WCK) * WCK * WKT;
            blocks += (total + 255) / 256;
        }
    }
    J.first[J.n] = blocks;
    hipLaunchKernelGGL(wino_weight_multi_kernel, dim3(blocks), dim3(256), 0, s, J);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_internal_wino_conv(int mode, const float* in, const float* w, const float* bias, const float* mask_src, float* out,
                             uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, void* ws,
                             size_t ws_bytes, hipStream_t s) {
    if (!in || !w || !out || !ws || !clhip_internal_wino_ok(Cin, Cout, H, W) || ws_bytes < clhip_internal_wino_ws(Cin, Cout))
        return CLHIP_EINVAL;
    float* U = static_cast<float*>(ws);
    const int n_chunks = (Cin + WCK - 1) / WCK;
    const int total = ((Cout + WKT - 1) / WKT) * n_chunks * WCK * WKT;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, U, Cout, Cin, mode, n_chunks);
    CLHIP_LAUNCH_CHECK();
    return clhip_internal_wino_conv_u(mode, in, U, bias, mask_src, out, pool_idx, unpool, N, Cin, Cout, H, W, relu, s);
}

// ---- weight gradient through the Winograd path: slabs only, in the format of conv3x3_wgrad.hip (see wino_wgrad_kernel)
bool clhip_internal_wino_wgrad_ok(int C, int K, int H, int W) {
    return C % WKT == 0 && K % WKT == 0 && H >= 4 && W >= 8;      // odd H / W: half-outside tiles stage zeros (no fused un-pooling there)
}

// slabs the launch below wants (it takes fewer when the workspace is smaller)
size_t clhip_internal_wino_wgrad_ws(int N, int C, int K, int H, int W) {
    if (N <= 0 || !clhip_internal_wino_wgrad_ok(C, K, H, W)) return 0;
    const int kc_tiles = (K / WKT) * (C / WKT);
    return ((size_t)9 * K * C + K) * (size_t)((256 + kc_tiles - 1) / kc_tiles) * sizeof(float);
}

// Launch geometry of the weight gradient of one layer.  ps: the pixel-split kernel (wino_wgrad_ps_kernel) takes it, with `splits`
// pixel splits of its kc32 32 x 32 (k, c) tiles; else wino_wgrad_kernel with `splits` splits of its kc_tiles 64 x 64 tiles.
struct WgradGeo {
    bool wide, ps, vec;
    int tiles_w, tiles_h, kc_tiles, kc32;
    long long total, splits;
    size_t slab;
};

// CLHIP_ENOTSUP: not a shape of the Winograd weight gradient; CLHIP_ENOSPC: the workspace does not hold one slab
static int wino_wgrad_geo(const float* x, const float* dy, const uint8_t* unpool_idx, int N, int C, int K, int H, int W,
                          size_t ws_bytes, WgradGeo* g) {
    if (!clhip_internal_wino_wgrad_ok(C, K, H, W)) return CLHIP_ENOTSUP;
    if ((size_t)N * K * H * W >= ((size_t)1 << 29) || (size_t)N * C * H * W >= ((size_t)1 << 29)) return CLHIP_ENOTSUP;   // 32-bit byte offsets
    g->wide = W >= 16;
    const int TCS = g->wide ? 8 : 4, TRS = g->wide ? 2 : 4;
    if (((H | W) & 1) && unpool_idx) return CLHIP_ENOTSUP;
    g->tiles_w = ((W + 1) / 2 + TCS - 1) / TCS;
    g->tiles_h = ((H + 1) / 2 + TRS - 1) / TRS;
    g->total = (long long)g->tiles_w * g->tiles_h * N;
    if (g->total > 0x7fffffffLL) return CLHIP_ENOTSUP;
    g->kc_tiles = (K / WKT) * (C / WKT);
    g->slab = (size_t)9 * K * C + K;
    // one block per CU: the largest split count that keeps the grid within 256 blocks (18 (k, c) tiles x 15 splits = 270 blocks
    // ran as two rounds on AlexNet's 192 -> 384 layer: 405 us against 215 us with 14 splits)
    long long splits = g->kc_tiles >= 256 ? 1 : 256 / g->kc_tiles;
    if (splits > g->total) splits = g->total;
    const long long cap = (long long)(ws_bytes / (g->slab * sizeof(float)));
    if (cap < 1) return CLHIP_ENOSPC;
    if (splits > cap) splits = cap;
    g->splits = splits;
    // whole tiles in width and 16-byte-aligned tensors: stages staged in 16-byte pieces (VEC; other shapes: the scalar form)
    g->vec = W % (2 * TCS) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
    g->kc32 = (K / 32) * (C / 32);
    g->ps = wgrad_ps_on() && g->total < 16 * splits;      // (measured: layer 2 of the bench model, 12.5 stages per block, 99 -> 92 us; 50 stages: slower)
    if (g->ps) {
        // fewer than 16 stages per 64 x 64 tile block: 32 x 32 tiles, a quarter of the splits, two blocks per CU (wino_wgrad_ps_kernel)
        // blocks of the launch = (k, c) tiles x pixel splits: two per CU; one per CU on 8 x 8 maps with few (k, c) tiles, where a block
        // of the 512-block launch saw 3 - 6 stages (measured at N = 200, profiles/r05_o_wgps_blocks.txt: 64 -> 128 @8x8 30.5 -> 26.6 us,
        // 128 -> 128 @8x8 40.7 -> 39.3; every other layer of small_VGG9 is fastest at 512: 256 / 384 / 768 / 1024 blocks cost 1 - 28 %)
        // (inside the merged grids of the 16 x 16 layers, 400 backward-data blocks beside them: 256 / 384 / 400 / 512 / 624 / 800 weight-gradient
        // blocks give a bench step of 1.500 - 1.72 / 1.517 / 1.516 / 1.507 / 1.509 / 1.546 ms, profiles/r05_x_pair_wgrad_blocks.txt)
        const int target = ((long long)H * W <= 64 && g->kc32 <= 16) ? 256 : 512;
        long long sp = g->kc32 >= target ? 1 : target / g->kc32;
        if (sp > g->total) sp = g->total;
        if (sp > cap) sp = cap;
        g->splits = sp;
    }
    return 0;
}

int clhip_internal_wino_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N, int C,
                                      int K, int H, int W, void* ws, size_t ws_bytes, hipStream_t s, clhip_wgrad_job* job) {
    if (!x || !dy || !dw || !ws || !job || N <= 0) return CLHIP_EINVAL;
    WgradGeo g;
    const int rc = wino_wgrad_geo(x, dy, unpool_idx, N, C, K, H, W, ws_bytes, &g);
    if (rc) return rc;
    const bool wide = g.wide;
    const int tiles_w = g.tiles_w, tiles_h = g.tiles_h, kc_tiles = g.kc_tiles;
    const long long total = g.total, splits = g.splits;
    const size_t slab = g.slab;
    float* part = static_cast<float*>(ws);
    if (g.ps) {
        const long long sp = g.splits;
        const unsigned gridp = (unsigned)(g.kc32 * sp);
        const bool vec = g.vec;
#define WGP(TCS_, TRS_, UNP_)                                                                                                       \
        do {                                                                                                                       \
            if (vec) hipLaunchKernelGGL((wino_wgrad_ps_kernel<TCS_, TRS_, UNP_, true>), dim3(gridp), dim3(256), 0, s, x, dy, part,  \
                                        unpool_idx, N, C, K, H, W, tiles_w, tiles_h, (int)total, (int)sp, C / 32, slab);            \
            else hipLaunchKernelGGL((wino_wgrad_ps_kernel<TCS_, TRS_, UNP_, false>), dim3(gridp), dim3(256), 0, s, x, dy, part,     \
                                    unpool_idx, N, C, K, H, W, tiles_w, tiles_h, (int)total, (int)sp, C / 32, slab);                \
        } while (0)
        if (wide) { if (unpool_idx) WGP(8, 2, true); else WGP(8, 2, false); }
        else { if (unpool_idx) WGP(4, 4, true); else WGP(4, 4, false); }
#undef WGP
        CLHIP_LAUNCH_CHECK();
        *job = clhip_wgrad_job{part, dw, db, K, C, (int)sp};
        return 0;
    }
    const unsigned grid = (unsigned)(kc_tiles * splits);
    const bool vecg = g.vec;
#define WG(TCS_, TRS_, UNP_)                                                                                                         \
    do {                                                                                                                            \
        if (vecg) hipLaunchKernelGGL((wino_wgrad_kernel<TCS_, TRS_, 1, UNP_, true>), dim3(grid), dim3(256), 0, s, x, dy, part,       \
                                     unpool_idx, N, C, K, H, W, tiles_w, tiles_h, (int)total, (int)splits, C / WKT, slab);           \
        else hipLaunchKernelGGL((wino_wgrad_kernel<TCS_, TRS_, 1, UNP_, false>), dim3(grid), dim3(256), 0, s, x, dy, part,           \
                                unpool_idx, N, C, K, H, W, tiles_w, tiles_h, (int)total, (int)splits, C / WKT, slab);                \
    } while (0)
    if (wide) { if (unpool_idx) WG(8, 2, true); else WG(8, 2, false); }
    else { if (unpool_idx) WG(4, 4, true); else WG(4, 4, false); }
#undef WG
    CLHIP_LAUNCH_CHECK();
    *job = clhip_wgrad_job{part, dw, db, K, C, (int)splits};
    return 0;
}

// Backward of one 3x3 layer as ONE grid (wino_pair_kernel): dx = backward-data of dy on weights already transformed into U (mode 1
// of clhip_internal_wino_weights), and the weight-gradient slabs of (x, dy) into ws, exactly as clhip_internal_wino_conv_u(1, ...)
// followed by clhip_internal_wino_wgrad_partial would leave them (same device code, same block -> work mapping inside each half;
// one exception: on 8 x 8 maps with N * ceil(C / 32) > 512 and fewer than WINO16_BELOW_UNITS units launch_wino runs
// wino_conv16_kernel<1, UNPOOL, 2> where this grid runs wino_conv16g_body<4, 4, 4> — the same sums in another order; the layers
// of the VGG9 plans at the bench batch are bit-identical, tests/test_gpu_pair.py).
// C / K: the layer's in / out channels; unpool_idx: dy is the POOLED gradient + arg-max codes.  CLHIP_ENOTSUP: not a layer this
// grid takes (the caller issues the two launches): it takes even maps >= 16 wide and 8 x 8 maps whose weight gradient runs on the
// pixel-split kernel with 16-byte staging, at most CLHIP_PAIR_MAX_PIXELS pixels per map — the launches that fill one round of
// blocks or less; on larger maps the two launches fill the chip on their own (64 -> 64 @32x32: 190.8 us one after the other,
// 190.6 us co-resident, profiles/r05_s_coresident_pair.txt).
#ifndef CLHIP_PAIR_MAX_PIXELS
#define CLHIP_PAIR_MAX_PIXELS 256
#endif
int clhip_internal_wino_pair(const float* dy, const uint8_t* unpool_idx, const float* U, const float* mask_src, float* dx,
                             const float* x, float* dw, float* db, int N, int C, int K, int H, int W, void* ws, size_t ws_bytes,
                             hipStream_t s, clhip_wgrad_job* job) {
    if (!dy || !U || !dx || !x || !dw || !ws || !job || N <= 0) return CLHIP_EINVAL;
    if (((H | W) & 1) || (long long)H * W > CLHIP_PAIR_MAX_PIXELS || !clhip_internal_wino_ok(K, C, H, W)) return CLHIP_ENOTSUP;
    const bool wide = W >= 16, m88 = H == 8 && W == 8;
    if (!wide && !m88) return CLHIP_ENOTSUP;
    WgradGeo g;
    const int rc = wino_wgrad_geo(x, dy, unpool_idx, N, C, K, H, W, ws_bytes, &g);
    if (rc) return rc == CLHIP_ENOSPC ? CLHIP_ENOTSUP : rc;
    if (!g.ps || !g.vec || g.wide != wide) return CLHIP_ENOTSUP;
    // backward-data half: geometry of launch_wino's 16-tile launch (kernel Cin = K, Cout = C)
    const int TC = wide ? 8 : 4, TR = wide ? 2 : 4, ER = 4, NE = 2 * TR / ER;
    const int d_tiles_w = (W / 2 + TC - 1) / TC, d_groups = (H / 2 + ER - 1) / ER;
    // 8 x 8 maps of layers with few in-channels: one-image blocks of 16-channel waves (launch_wino's `narrow` launch of wino_conv16_kernel)
    const bool narrow = m88 && ((long long)N * 16 + 31) / 32 * ((C + 31) / 32) < WINO16_BELOW_UNITS && (long long)N * ((C + 31) / 32) <= 512;
    const long long npb = narrow ? N : (((long long)N * d_groups + NE - 1) / NE) * d_tiles_w, nb_d = npb * ((C + WKT - 1) / WKT);
    const long long nb_w = (long long)g.kc32 * g.splits;
    if (nb_d <= 0 || nb_w <= 0 || nb_d + nb_w > 0x7fffffffLL) return CLHIP_ENOTSUP;
    float* part = static_cast<float*>(ws);
    uint8_t* idx = const_cast<uint8_t*>(unpool_idx);
#define PAIR(DK_, TC_, TR_, UNP_)                                                                                                   \
    hipLaunchKernelGGL((wino_pair_kernel<DK_, TC_, TR_, 4, UNP_, TC_, TR_>), dim3((unsigned)(nb_d + nb_w)), dim3(256), 0, s, dy, U,  \
                       mask_src, dx, idx, N, K, C, H, W, d_tiles_w, d_groups, (int)npb, x, part, g.tiles_w, g.tiles_h, (int)g.total, \
                       (int)g.splits, C / 32, g.slab, (int)nb_w)
    if (wide) { if (unpool_idx) PAIR(0, 8, 2, true); else PAIR(0, 8, 2, false); }
    else if (narrow) { if (unpool_idx) PAIR(1, 4, 4, true); else PAIR(1, 4, 4, false); }
    else { if (unpool_idx) PAIR(0, 4, 4, true); else PAIR(0, 4, 4, false); }
#undef PAIR
    CLHIP_LAUNCH_CHECK();
    *job = clhip_wgrad_job{part, dw, db, K, C, (int)g.splits};
    return 0;
}

// plan-time form of the shape test above (16-byte-aligned tensors, enough workspace): does the backward of this layer run as one grid?
bool clhip_internal_wino_pair_shape(int N, int C, int K, int H, int W, int pooled) {
    if (N <= 0 || ((H | W) & 1) || (long long)H * W > CLHIP_PAIR_MAX_PIXELS || !clhip_internal_wino_ok(K, C, H, W)) return false;
    const bool wide = W >= 16, m88 = H == 8 && W == 8;
    if (!wide && !m88) return false;
    WgradGeo g;
    alignas(16) static const float a16[4] = {0.f, 0.f, 0.f, 0.f};
    static const uint8_t one = 0;
    if (wino_wgrad_geo(a16, a16, pooled ? &one : nullptr, N, C, K, H, W, (size_t)1 << 40, &g)) return false;
    return g.ps && g.vec && g.wide == wide;
}

extern "C" {

size_t clhip_conv3x3_wino_ws(int C, int K) {
    const size_t a = clhip_internal_wino_ws(C, K), b = clhip_internal_wino_ws(K, C);
    return a > b ? a : b;
}

int clhip_conv3x3_wino_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx_u8_or_null, int N, int C, int K,
                           int H, int W, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_wino_ok(C, K, H, W)) return CLHIP_ENOTSUP;
    return clhip_internal_wino_conv(0, x, w, b, nullptr, y, idx_u8_or_null, 0, N, C, K, H, W, relu, ws, ws_bytes, as_stream(stream));
}

int clhip_conv3x3_wino_bwd_data(const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src, float* dx,
                                int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (N <= 0 || !clhip_internal_wino_ok(K, C, H, W)) return CLHIP_ENOTSUP;
    return clhip_internal_wino_conv(1, dy, w, nullptr, relu_src, dx, const_cast<uint8_t*>(idx_u8_or_null), idx_u8_or_null != nullptr,
                                    N, K, C, H, W, 0, ws, ws_bytes, as_stream(stream));
}


size_t clhip_conv3x3_wino_bwd_weight_ws(int N, int C, int K, int H, int W) {
    const size_t a = clhip_internal_wino_wgrad_ws(N, C, K, H, W), b = clhip_conv3x3_bwd_weight_ws(N, C, K, H, W);
    return a > b ? a : b;
}

int clhip_conv3x3_wino_bwd_weight(const float* x, const float* dy, const uint8_t* idx_u8_or_null, float* dw, float* db, int N, int C,
                                  int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    clhip_wgrad_job job;
    int rc = clhip_internal_wino_wgrad_partial(x, dy, idx_u8_or_null, dw, db, N, C, K, H, W, ws, ws_bytes, as_stream(stream), &job);
    if (rc) return rc;
    return clhip_conv3x3_bwd_weight_reduce(ws, dw, db, K, C, job.splits, stream);
}

// workspace of clhip_conv3x3_wino_bwd: the transformed weights (16-byte multiple), then the weight-gradient slabs
static size_t wino_bwd_u_bytes(int C, int K) { return (clhip_internal_wino_ws(K, C) + 255) & ~(size_t)255; }

size_t clhip_conv3x3_wino_bwd_ws(int N, int C, int K, int H, int W) {
    const size_t slabs = clhip_internal_wino_wgrad_ws(N, C, K, H, W);
    return slabs ? wino_bwd_u_bytes(C, K) + slabs : 0;
}

int clhip_conv3x3_wino_bwd(const float* x, const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src,
                           float* dx, float* dw, float* db, int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dy || !w || !dx || !dw || !ws || N <= 0) return CLHIP_EINVAL;
    if (((H | W) & 1) || !clhip_internal_wino_ok(K, C, H, W) || !clhip_internal_wino_wgrad_ok(C, K, H, W)) return CLHIP_ENOTSUP;
    const size_t ub = wino_bwd_u_bytes(C, K);
    if (ws_bytes < clhip_conv3x3_wino_bwd_ws(N, C, K, H, W)) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    float* U = static_cast<float*>(ws);
    const int n_chunks = (K + WCK - 1) / WCK;
    const int total = ((C + WKT - 1) / WKT) * n_chunks * WCK * WKT;
    clhip_wgrad_job job;
    // (shape check first: nothing is launched for a layer the merged grid does not take)
    WgradGeo g;
    if (wino_wgrad_geo(x, dy, idx_u8_or_null, N, C, K, H, W, ws_bytes - ub, &g) || !g.ps || !g.vec) return CLHIP_ENOTSUP;
    if ((long long)H * W > CLHIP_PAIR_MAX_PIXELS || !(W >= 16 || (H == 8 && W == 8))) return CLHIP_ENOTSUP;
    hipLaunchKernelGGL(wino_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, U, C, K, 1, n_chunks);
    CLHIP_LAUNCH_CHECK();
    int rc = clhip_internal_wino_pair(dy, idx_u8_or_null, U, relu_src, dx, x, dw, db, N, C, K, H, W, static_cast<char*>(ws) + ub,
                                      ws_bytes - ub, s, &job);
    if (rc) return rc;
    return clhip_conv3x3_bwd_weight_reduce(static_cast<char*>(ws) + ub, dw, db, K, C, job.splits, stream);
}

}  // extern "C"
