// bswgrad5.hip — weight gradient of a 5x5 convolution (stride 1, padding 2) on the bf16 matrix cores with fp32 operands split into three
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

__global__ __launch_bounds__(256, 2) void bs_wgrad5_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                           int N, int C, int K, int H, int W, int types, int splits, size_t slab_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    const int type = (int)(gw % types), split = (int)(gw / types);
    if (split >= splits) return;                    // (no barrier in this kernel)
    const int s = type % 5, tile = type / 5;
    const int c_tiles = C >> 5;
    const int ct = tile % c_tiles, kt = tile / c_tiles;
    const int k0 = kt * 32, c0 = ct * 32;
    const int strips = (W + 15) >> 4;
    const int plane = H * W;
    const int rows_total = N * strips * H;
    const int per = rows_total / splits, extra = rows_total % splits;
    const int r_begin = split * per + (split < extra ? split : extra), r_end = r_begin + per + (split < extra ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(x, (size_t)N * C * plane * 4);
    const __amdgpu_buffer_rsrc_t rs_d = clhip_rsrc(dy, (size_t)N * K * plane * 4);

    floatx16 acc[5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    float bsum = 0.f;

    for (int rr = r_begin; rr < r_end;) {
        const int ya = rr % H, colid = rr / H;
        const int sx = colid % strips, n = colid / strips;
        const int yb = min(H, ya + (r_end - rr));                     // rows [ya, yb) of column (n, sx)
        rr += yb - ya;
        const int col0 = 16 * sx + 8 * half;
        // dy: the lane's 8 pixels col0 .. col0 + 7 of out-channel k0 + col; x: the 8 pixels col0 + s - 2 .. of in-channel c0 + col — a lane
        // whose window starts left of the map loads from column 0 and moves its values up by `xsh` places
        const int nd = min(8, max(0, W - col0));                      // valid dy pixels of this lane (a prefix)
        const int w0 = col0 + s - 2;
        const int xsh = w0 < 0 ? -w0 : 0;
        const int nx_hi = min(8, W - w0);                            // x window elements e with e < nx_hi are inside the map (e >= xsh)
        const int d_off = nd > 0 ? ((n * K + k0 + col) * plane + col0) * 4 : CLHIP_OOB;
        const int x_off = nx_hi > 0 ? ((n * C + c0 + col) * plane + w0 + xsh) * 4 : CLHIP_OOB;
        // wave-uniform: some lane of this column has pixels outside the map (the last strips: past the right edge; the first strip: the
        // x window starts left of the map when s < 2)
        const bool edge = __builtin_amdgcn_ballot_w64(xsh > 0 || nx_hi < 8 || nd < 8) != 0;

        auto load_row = [&](const __amdgpu_buffer_rsrc_t& rs, int off, int y, bool ok) {
            B5Raw r;
            const int so = ok ? y * W * 4 : 0;
            r.lo = clhip_buf_load4(rs, ok ? off : CLHIP_OOB, so);
            r.hi = clhip_buf_load4(rs, ok ? off + 16 : CLHIP_OOB, so);
            return r;
        };
        auto load_d = [&](int y) { return load_row(rs_d, d_off, y, y >= ya && y < yb); };
        auto load_x = [&](int y) { return load_row(rs_x, x_off, y, (unsigned)y < (unsigned)H); };
        auto split_d = [&](const B5Raw& r) {
            float v[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
            if (edge) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = e < nd ? v[e] : 0.f;
            }
            bsum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            return b5_split(v);
        };
        auto split_x = [&](const B5Raw& r) {
            float v[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
            if (edge) {
                float u[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // window element e = loaded element e - xsh (xsh in {0, 1, 2}); outside the map: zero
                    const float a = v[e], b = e >= 1 ? v[e - 1] : 0.f, c = e >= 2 ? v[e - 2] : 0.f;
                    const float sel = xsh == 0 ? a : xsh == 1 ? b : c;
                    u[e] = (e >= xsh && e < nx_hi) ? sel : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = u[e];
            }
            return b5_split(v);
        };
        // acc[r] += A x B: the six products, small ones first
        auto mm = [&](const B5Op& a, const B5Op& b, int r) {
#define B5_MM(I, J) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b5_bf16x8, a.p[I]), __builtin_bit_cast(b5_bf16x8, b.p[J]), acc[r], 0, 0, 0)
            B5_MM(0, 2); B5_MM(2, 0); B5_MM(1, 1); B5_MM(0, 1); B5_MM(1, 0); B5_MM(0, 0);
#undef B5_MM
        };

        // x rows ys .. ye; at the step of x row y_i: tap r meets dy row y_i + 2 - r.  Pipeline state at the top of a step: ring slots hold
        // split dy(y_i - 2 .. y_i + 2), b_cur = split x(y_i), rx / rd = the raw x(y_i + 1) / dy(y_i + 3) in flight.
        const int ys = ya > 2 ? ya - 2 : 0, ye = yb + 1 < H ? yb + 1 : H - 1;
        B5Op a0, a1, a2, a3, a4, a5, b_cur;
        B5Raw rx, rd;
        {
            const B5Raw d0 = load_d(ys - 2), d1 = load_d(ys - 1), d2 = load_d(ys), d3 = load_d(ys + 1), d4 = load_d(ys + 2);
            const B5Raw x0 = load_x(ys);
            rx = load_x(ys + 1);
            rd = load_d(ys + 3);
            // slot(rel) = rel mod 6 with rel = y - ys: rows ys - 2 .. ys + 2 -> slots 4, 5, 0, 1, 2
            a4 = split_d(d0); a5 = split_d(d1); a0 = split_d(d2); a1 = split_d(d3); a2 = split_d(d4);
            a3 = B5Op{};
            b_cur = split_x(x0);
        }
        // one row: t0 .. t4 = dy(y_i + 2) .. dy(y_i - 2) (taps r = 0 .. 4), a_new <- dy(y_i + 3)
        auto step = [&](const B5Op& t0, const B5Op& t1, const B5Op& t2, const B5Op& t3, const B5Op& t4, B5Op& a_new, int yi) {
            const B5Raw rx_cur = rx, rd_cur = rd;
            rx = load_x(yi + 2);
            rd = load_d(yi + 4);
            mm(t0, b_cur, 0);
            mm(t1, b_cur, 1);
            mm(t2, b_cur, 2);
            mm(t3, b_cur, 3);
            mm(t4, b_cur, 4);
            const B5Op b_next = split_x(rx_cur);
            a_new = split_d(rd_cur);
#pragma unroll
            for (int i = 0; i < 30; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, B5_VPM, 0);
            }
            b_cur = b_next;
        };
        for (int yi = ys; yi <= ye; yi += 6) {
            step(a2, a1, a0, a5, a4, a3, yi);
            if (yi + 1 > ye) break;
            step(a3, a2, a1, a0, a5, a4, yi + 1);
            if (yi + 2 > ye) break;
            step(a4, a3, a2, a1, a0, a5, yi + 2);
            if (yi + 3 > ye) break;
            step(a5, a4, a3, a2, a1, a0, yi + 3);
            if (yi + 4 > ye) break;
            step(a0, a5, a4, a3, a2, a1, yi + 4);
            if (yi + 5 > ye) break;
            step(a1, a0, a5, a4, a3, a2, yi + 5);
        }
    }

    // ---- slab of this share: acc[r][i] = dW[tap 5 r + s][k0 + row(i, lane)][c0 + col]
    float* slab = part + (size_t)split * slab_stride;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[((size_t)(5 * r + s) * K + k0 + mfma32_row(i, lane)) * C + c0 + col] = acc[r][i];
    if (ct == 0 && s == 0) {                       // bias sums: the two pixel halves of out-channel k0 + col
        bsum += __shfl_xor(bsum, 32, 64);
        if (half == 0) slab[(size_t)25 * K * C + k0 + col] = bsum;
    }
}

}  // namespace

// shapes: whole 32-channel tiles on both sides, maps at least 16 wide, tensors under 2 GB (32-bit byte offsets)
bool clhip_internal_bs5_wgrad_ok(int N, int C, int K, int H, int W) {
    return N >= 1 && C >= 32 && C % 32 == 0 && K >= 32 && K % 32 == 0 && W >= 16 && H >= 1 &&
           (size_t)N * C * H * W * 4 < 0x7fffffffull && (size_t)N * K * H * W * 4 < 0x7fffffffull;
}

static int bs5_wgrad_splits(int N, int C, int K, int H, int W) {
    const int types = (K / 32) * (C / 32) * 5;
    const long long rows = (long long)N * ((W + 15) / 16) * H;
    long long s = 2048 / types;                    // two waves per SIMD
    if (s > rows) s = rows;
    return (int)(s < 1 ? 1 : s);
}

size_t clhip_internal_bs5_wgrad_ws(int N, int C, int K, int H, int W) {
    if (!clhip_internal_bs5_wgrad_ok(N, C, K, H, W)) return 0;
    return (size_t)bs5_wgrad_splits(N, C, K, H, W) * ((size_t)25 * K * C + K) * 4;
}

// dW [K][C][5][5], db [K] of the 5x5 / padding-2 convolution: slabs + the fixed-order reduction
int clhip_internal_bs5_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                             size_t ws_bytes, hipStream_t s) {
    if (!clhip_internal_bs5_wgrad_ok(N, C, K, H, W)) return CLHIP_ENOTSUP;
    if (!x || !dy || !dw || !ws) return CLHIP_EINVAL;
    if (ws_bytes < clhip_internal_bs5_wgrad_ws(N, C, K, H, W)) return CLHIP_ENOSPC;
    const int splits = bs5_wgrad_splits(N, C, K, H, W), types = (K / 32) * (C / 32) * 5;
    const size_t slab = (size_t)25 * K * C + K;
    const long long waves = (long long)types * splits;
    float* part = static_cast<float*>(ws);
    hipLaunchKernelGGL(bs_wgrad5_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, dy, part, N, C, K, H, W, types, splits, slab);
    CLHIP_LAUNCH_CHECK();
    clhip_wgrad_job job{part, dw, db, K, C, splits, 25};
    return clhip_internal_wgrad_reduce_multi(&job, 1, s);
}

extern "C" {

size_t clhip_conv5x5_bs_bwd_weight_ws(int N, int C, int K, int H, int W) { return clhip_internal_bs5_wgrad_ws(N, C, K, H, W); }

int clhip_conv5x5_bs_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                                size_t ws_bytes, void* stream) {
    return clhip_internal_bs5_wgrad(x, dy, dw, db, N, C, K, H, W, ws, ws_bytes, as_stream(stream));
}

}  // extern "C"
