// bswgrad5.hip — weight gradient of a KS x KS convolution (stride 1, padding KS / 2; KS = 5, and 3 for the maps bswgrad.hip does not take) on
// the bf16 matrix cores with fp32 operands split into three
// bf16 pieces each: autograd's convolution_backward w.r.t. weight and bias of AlexNet's second convolution, nn.Conv2d(64, 192, 5, padding=2)
// on 27 x 27 maps (models/net.py:96-125 via torchvision.models.alexnet).  On the gather-GEMM of conv2d.hip that launch is the longest of
// an AlexNet training step: 567 us at N = 128 (57.3 GFLOP = 101 TFLOP/s, 0.64 of the f32 matrix pipe; profiles/r06_l_alexnet_s2d.txt).
//
//   dW[k][c][r][s] = sum over n, y, x of dy[n][k][y][x] * X[n][c][y + r - 2][x + s - 2]
//
// The scheme of bswgrad.hip — M = 32 out-channels (A = dy), N = 32 in-channels (B = x shifted by the tap), the MFMA's 16-deep
// reduction = 16 consecutive pixels of one image row, operands straight from NCHW memory into registers, no LDS — with the 25 taps
// dealt out differently: 25 x 16 accumulators do not fit one wave, so a wave owns ONE column tap s and the five row taps r of its
// (k, c) tile (80 accumulators): per x row y_i it loads the 8 pixels shifted by s - 2 directly (the rows of a 27-wide map have no
// alignment to keep), splits them once, and multiplies them with the five dy rows y_i + 2 .. y_i - 2, which slide through a ring of six
// register sets (one new row per step, split inside the MFMA stream of the step before).  30 MFMAs and ~90 VALU instructions per row.
// The five waves of a (k, c) tile are neighbours in the grid (wave type = (tile, s), s fastest): they load the same dy rows and the same
// x lines.  Every wave takes a contiguous share of the (image, strip, row) list; one slab [25][K][C] (+ [K]) per share, reduced by
// conv3x3_wgrad.hip's fixed-order launch (bitwise deterministic).  Maps whose width is not a multiple of 16 (27!) mask the columns
// past the edge in their last strip.
//
// The same kernel with KS = 3 (a wave owns one column tap and the three row taps of TWO 32-out-channel tiles: 96 accumulators, the x row
// split once for both) takes the 3x3 layers whose maps bswgrad.hip cannot: AlexNet's 13 x 13 layers (one strip of 16 with three masked
// columns), where the Winograd weight gradient runs 243 us per layer at N = 128.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef __bf16 b5_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b5_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b5_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned b5_pk(float lo, float hi) {
    b5_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b5_bf16x2));
}
// (a, b) -> three packed bf16 pairs with a = a0 + a1 + a2, b likewise (bsconv.hip, bs_split2)
__device__ __forceinline__ void b5_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = b5_pk(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = b5_pk(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = b5_pk(ra, rb);
}

struct B5Op { clhip_u32x4 p[3]; };                // 8 pixels of one channel: three pieces
struct B5Raw { float4 lo, hi; };

__device__ __forceinline__ B5Op b5_split(const float (&v)[8]) {
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b5_split2(v[2 * j], v[2 * j + 1], q0[j], q1[j], q2[j]);
    B5Op o;
    o.p[0] = clhip_u32x4{q0[0], q0[1], q0[2], q0[3]};
    o.p[1] = clhip_u32x4{q1[0], q1[1], q1[2], q1[3]};
    o.p[2] = clhip_u32x4{q2[0], q2[1], q2[2], q2[3]};
    return o;
}

#ifndef B5_VPM
#define B5_VPM 3
#endif

// KS x KS taps, padding HP = KS / 2; KT tiles of 32 out-channels per wave (they share the split x row)
template <int KS, int KT>
__global__ __launch_bounds__(256, 2) void bs_wgradk_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                           int N, int C, int K, int H, int W, int types, int splits, size_t slab_stride) {
    constexpr int HP = KS / 2, RING = KS + 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    const int type = (int)(gw % types), split = (int)(gw / types);
    if (split >= splits) return;                    // (no barrier in this kernel)
    const int s = type % KS, tile = type / KS;
    const int c_tiles = C >> 5;
    const int ct = tile % c_tiles, kt = tile / c_tiles;
    const int k0 = kt * 32 * KT, c0 = ct * 32;
    const int strips = (W + 15) >> 4;
    const int plane = H * W;
    const int rows_total = N * strips * H;
    const int per = rows_total / splits, extra = rows_total % splits;
    const int r_begin = split * per + (split < extra ? split : extra), r_end = r_begin + per + (split < extra ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(x, (size_t)N * C * plane * 4);
    const __amdgpu_buffer_rsrc_t rs_d = clhip_rsrc(dy, (size_t)N * K * plane * 4);

    floatx16 acc[KT][KS];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][r][i] = 0.f;
    float bsum[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) bsum[t] = 0.f;

    for (int rr = r_begin; rr < r_end;) {
        const int ya = rr % H, colid = rr / H;
        const int sx = colid % strips, n = colid / strips;
        const int yb = min(H, ya + (r_end - rr));                     // rows [ya, yb) of column (n, sx)
        rr += yb - ya;
        const int col0 = 16 * sx + 8 * half;
        // dy: the lane's 8 pixels col0 .. col0 + 7 of out-channel k0 + 32 t + col; x: the 8 pixels col0 + s - HP .. of in-channel c0 + col —
        // a lane whose window starts left of the map loads from column 0 and moves its values up by `xsh` places
        const int nd = min(8, max(0, W - col0));                      // valid dy pixels of this lane (a prefix)
        const int w0 = col0 + s - HP;
        const int xsh = w0 < 0 ? -w0 : 0;
        const int nx_hi = min(8, W - w0);                            // x window elements e with e < nx_hi are inside the map (e >= xsh)
        int d_off[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) d_off[t] = nd > 0 ? ((n * K + k0 + 32 * t + col) * plane + col0) * 4 : CLHIP_OOB;
        const int x_off = nx_hi > 0 ? ((n * C + c0 + col) * plane + w0 + xsh) * 4 : CLHIP_OOB;
        // wave-uniform: some lane of this column has pixels outside the map (the last strips: past the right edge; the first strip: the
        // x window starts left of the map when s < HP)
        const bool edge = __builtin_amdgcn_ballot_w64(xsh > 0 || nx_hi < 8 || nd < 8) != 0;

        auto load_row = [&](const __amdgpu_buffer_rsrc_t& rs, int off, int y, bool ok) {
            B5Raw r;
            const int so = ok ? y * W * 4 : 0;
            r.lo = clhip_buf_load4(rs, ok ? off : CLHIP_OOB, so);
            r.hi = clhip_buf_load4(rs, ok ? off + 16 : CLHIP_OOB, so);
            return r;
        };
        auto load_x = [&](int y) { return load_row(rs_x, x_off, y, (unsigned)y < (unsigned)H); };
        auto split_d = [&](const B5Raw& r, int t) {
            float v[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
            if (edge) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = e < nd ? v[e] : 0.f;
            }
            bsum[t] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            return b5_split(v);
        };
        auto split_x = [&](const B5Raw& r) {
            float v[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
            if (edge) {
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // window element e = loaded element e - xsh (xsh in {0 .. HP}); outside the map: zero
                    const float a = v[e], b = e >= 1 ? v[e - 1] : 0.f, c = e >= 2 ? v[e - 2] : 0.f;
                    const float sel = xsh == 0 ? a : xsh == 1 ? b : c;
                    u[e] = (e >= xsh && e < nx_hi) ? sel : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = u[e];
            }
            return b5_split(v);
        };
        // acc += A x B: the six products, small ones first
        auto mm = [&](const B5Op& a, const B5Op& b, floatx16& d) {
#define B5_MM(I, J) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b5_bf16x8, a.p[I]), __builtin_bit_cast(b5_bf16x8, b.p[J]), d, 0, 0, 0)
            B5_MM(0, 2); B5_MM(2, 0); B5_MM(1, 1); B5_MM(0, 1); B5_MM(1, 0); B5_MM(0, 0);
#undef B5_MM
        };

        // x rows ys .. ye; at the step of x row y_i tap r meets dy row y_i + HP - r.  Ring slot of dy row y: (y - ys) mod RING (one more slot
        // than rows in use: the row of the next step is split while this step's MFMAs still read the oldest one).  Pipeline state at the
        // top of a step: the ring holds split dy(y_i - HP .. y_i + HP), b_cur = split x(y_i), rx / rd = the raw x(y_i + 1) / dy(y_i + HP + 1).
        const int ys = ya > HP ? ya - HP : 0, ye = yb - 1 + HP < H ? yb - 1 + HP : H - 1;
        B5Op ring[KT][RING], b_cur;
        B5Raw rx, rd[KT];
        auto load_d = [&](int y, int t) { return load_row(rs_d, d_off[t], y, y >= ya && y < yb); };
        {
            B5Raw d0[KT][KS];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int i = 0; i < KS; ++i) d0[t][i] = load_d(ys - HP + i, t);
            const B5Raw x0 = load_x(ys);
            rx = load_x(ys + 1);
#pragma unroll
            for (int t = 0; t < KT; ++t) rd[t] = load_d(ys + HP + 1, t);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
                for (int i = 0; i < KS; ++i) ring[t][(i - HP + RING) % RING] = split_d(d0[t][i], t);       // rel = i - HP
                ring[t][(HP + 1) % RING] = B5Op{};
            }
            b_cur = split_x(x0);
        }
        for (int yi = ys; yi <= ye; yi += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                if (yi + j > ye) break;
                const B5Raw rx_cur = rx;
                B5Raw rd_cur[KT];
#pragma unroll
                for (int t = 0; t < KT; ++t) rd_cur[t] = rd[t];
                rx = load_x(yi + j + 2);
#pragma unroll
                for (int t = 0; t < KT; ++t) rd[t] = load_d(yi + j + HP + 2, t);
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < KS; ++r) mm(ring[t][(j + HP - r + RING) % RING], b_cur, acc[t][r]);
                const B5Op b_next = split_x(rx_cur);
#pragma unroll
                for (int t = 0; t < KT; ++t) ring[t][(j + HP + 1) % RING] = split_d(rd_cur[t], t);
#pragma unroll
                for (int i = 0; i < 6 * KS * KT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, B5_VPM, 0);
                }
                b_cur = b_next;
            }
        }
    }

    // ---- slab of this share: acc[t][r][i] = dW[tap KS r + s][k0 + 32 t + row(i, lane)][c0 + col]
    float* slab = part + (size_t)split * slab_stride;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < KS; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) slab[((size_t)(KS * r + s) * K + k0 + 32 * t + mfma32_row(i, lane)) * C + c0 + col] = acc[t][r][i];
    if (ct == 0 && s == 0) {                       // bias sums: the two pixel halves of out-channel k0 + 32 t + col
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const float b = bsum[t] + __shfl_xor(bsum[t], 32, 64);
            if (half == 0) slab[(size_t)KS * KS * K * C + k0 + 32 * t + col] = b;
        }
    }
}

template <int KS> struct BwkCfg { static constexpr int KT = 1; };       // (KS = 3 with two out-channel tiles per wave spills 252 bytes at two waves per SIMD)

// shapes: whole 32-channel tiles on the input side, whole 32 KT-channel groups on the output side, tensors under 2 GB (32-bit byte offsets)
template <int KS>
bool bsk_ok(int N, int C, int K, int H, int W) {
    constexpr int KT = BwkCfg<KS>::KT;
    return N >= 1 && C >= 32 && C % 32 == 0 && K >= 32 * KT && K % (32 * KT) == 0 && W >= 8 && H >= 1 &&
           (size_t)N * C * H * W * 4 < 0x7fffffffull && (size_t)N * K * H * W * 4 < 0x7fffffffull;
}

template <int KS>
int bsk_splits(int N, int C, int K, int H, int W) {
    const int types = (K / (32 * BwkCfg<KS>::KT)) * (C / 32) * KS;
    const long long rows = (long long)N * ((W + 15) / 16) * H;
    long long s = 2048 / types;                    // two waves per SIMD
    if (s > rows) s = rows;
    return (int)(s < 1 ? 1 : s);
}

template <int KS>
size_t bsk_ws(int N, int C, int K, int H, int W) {
    if (!bsk_ok<KS>(N, C, K, H, W)) return 0;
    return (size_t)bsk_splits<KS>(N, C, K, H, W) * ((size_t)KS * KS * K * C + K) * 4;
}

template <int KS>
int bsk_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, hipStream_t s) {
    constexpr int KT = BwkCfg<KS>::KT;
    if (!bsk_ok<KS>(N, C, K, H, W)) return CLHIP_ENOTSUP;
    if (!x || !dy || !dw || !ws) return CLHIP_EINVAL;
    if (ws_bytes < bsk_ws<KS>(N, C, K, H, W)) return CLHIP_ENOSPC;
    const int splits = bsk_splits<KS>(N, C, K, H, W), types = (K / (32 * KT)) * (C / 32) * KS;
    const size_t slab = (size_t)KS * KS * K * C + K;
    const long long waves = (long long)types * splits;
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL((bs_wgradk_kernel<KS, KT>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, dy, part, N, C, K, H, W, types, splits, slab);
    CLHIP_LAUNCH_CHECK();
    clhip_wgrad_job job{part, dw, db, K, C, splits, KS * KS};
    return clhip_internal_wgrad_reduce_multi(&job, 1, s);
}

}  // namespace

bool clhip_internal_bs5_wgrad_ok(int N, int C, int K, int H, int W) { return bsk_ok<5>(N, C, K, H, W); }
size_t clhip_internal_bs5_wgrad_ws(int N, int C, int K, int H, int W) { return bsk_ws<5>(N, C, K, H, W); }
// dW [K][C][5][5], db [K] of the 5x5 / padding-2 convolution: slabs + the fixed-order reduction
int clhip_internal_bs5_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                             size_t ws_bytes, hipStream_t s) {
    return bsk_wgrad<5>(x, dy, dw, db, N, C, K, H, W, ws, ws_bytes, s);
}
// the 3x3 / padding-1 layer on the same kernel (any map width >= 8: the maps bswgrad.hip's 16-pixel-aligned kernel does not take)
size_t clhip_internal_bs3k_wgrad_ws(int N, int C, int K, int H, int W) { return bsk_ws<3>(N, C, K, H, W); }
int clhip_internal_bs3k_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                              size_t ws_bytes, hipStream_t s) {
    return bsk_wgrad<3>(x, dy, dw, db, N, C, K, H, W, ws, ws_bytes, s);
}

extern "C" {

size_t clhip_conv5x5_bs_bwd_weight_ws(int N, int C, int K, int H, int W) { return clhip_internal_bs5_wgrad_ws(N, C, K, H, W); }

int clhip_conv5x5_bs_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                                size_t ws_bytes, void* stream) {
    return clhip_internal_bs5_wgrad(x, dy, dw, db, N, C, K, H, W, ws, ws_bytes, as_stream(stream));
}

}  // extern "C"
