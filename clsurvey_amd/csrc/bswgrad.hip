// bswgrad.hip — weight gradient of a 3x3 convolution (stride 1, padding 1) on the bf16 matrix cores with fp32 operands split into three
// bf16 pieces each: autograd's convolution_backward w.r.t. weight and bias of the conv layers of VGGSlim.forward
// (models/VGGSlim.py:27-40), the operator clhip_conv3x3_bwd_weight / clhip_conv3x3_wino_bwd_weight compute on the f32 matrix cores
// (37 % of a training pass there, at 0.28 - 0.47 of that pipe).  Arithmetic as in bsconv.hip: a = a0 + a1 + a2 exactly, the six
// products ai bj with i + j <= 2 are exact in fp32 and accumulate in fp32 inside v_mfma_f32_32x32x16_bf16.
//
//   dW[k][c][r][s] = sum over n, y, x of dy[n][k][y][x] * X[n][c][y + r - 1][x + s - 1]
//
// as a matrix product per tap: M = 32 out-channels k (A operand = dy), N = 32 in-channels c (B operand = x shifted by the tap), the
// MFMA's 16-deep reduction = 16 consecutive pixels of one image row.  Both operands are "8 consecutive pixels of one channel" per
// lane — NCHW as it lies in memory — so NOTHING goes through LDS and there is no barrier: a lane loads its 8 dy values and the 10 x
// values around them (16-byte aligned float4s + the two halo columns) straight into registers, splits them, and makes the three
// column taps of a piece from the 5 packed pairs P0..P4 of the 10 values: s = 0 -> P0..P3, s = 2 -> P1..P4, s = 1 -> v_alignbit of
// neighbouring pairs.  A wave owns a 32 x 32 (k, c) tile with all 9 taps in registers (144 accumulators) and walks down a 16-pixel
// wide strip of an image: the x row y_i meets the three dy rows y_i + 1, y_i, y_i - 1 (taps r = 0, 1, 2), which slide through
// registers — 54 MFMAs per row, 2.2 VALU instructions per MFMA; the raw values of a row are loaded two rows ahead and split one row
// ahead, inside the MFMA stream of the row before.  The four waves of a block are the 2 x 2 quadrants of a 64 x 64 (k, c) tile on
// the same pixels (each row is loaded by two of them: L1 hits).  A block works through a contiguous share of the (image, strip,
// row) list — equal shares to within one row — and writes ONE slab [9][K][C] (+ [K] bias sums) at the end, the format of
// conv3x3_wgrad.hip, reduced by the same fixed-order launches (bitwise deterministic).
//
// UNPOOL: dy is the POOLED gradient + 2x2 arg-max codes (common.hpp: 0..3 position, 4 dead) — the un-pooled row is rebuilt in
// registers from 4 pooled values + 4 codes per lane.
#include "common.hpp"
#include <cstdlib>

namespace {

typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bw_f32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ unsigned bw_pk(float lo, float hi) {
    bw_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bw_bf16x2));
}
// (a, b) -> three packed bf16 pairs with a = a0 + a1 + a2, b likewise (bsconv.hip, bs_split2)
__device__ __forceinline__ void bw_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = bw_pk(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = bw_pk(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = bw_pk(ra, rb);
}

struct BwA { clhip_u32x4 p[3]; };                 // one dy row of a lane: 8 pixels, three pieces
struct BwB { unsigned p[3][5]; };                 // one x row of a lane: 10 pixels as 5 pairs, three pieces

__device__ __forceinline__ bw_bf16x8 bw_op(unsigned a, unsigned b, unsigned c, unsigned d) {
    return __builtin_bit_cast(bw_bf16x8, clhip_u32x4{a, b, c, d});
}
// column tap s of piece i
template <int S>
__device__ __forceinline__ bw_bf16x8 bw_tap(const BwB& b, int i) {
    if constexpr (S == 0) return bw_op(b.p[i][0], b.p[i][1], b.p[i][2], b.p[i][3]);
    else if constexpr (S == 2) return bw_op(b.p[i][1], b.p[i][2], b.p[i][3], b.p[i][4]);
    else return bw_op(__builtin_amdgcn_alignbit(b.p[i][1], b.p[i][0], 16), __builtin_amdgcn_alignbit(b.p[i][2], b.p[i][1], 16),
                      __builtin_amdgcn_alignbit(b.p[i][3], b.p[i][2], 16), __builtin_amdgcn_alignbit(b.p[i][4], b.p[i][3], 16));
}

#if BW_ABL & 1
__device__ __forceinline__ floatx16 bw_fake_mfma(bw_bf16x8 a, bw_bf16x8 b, floatx16 c) {
    c[0] += __builtin_bit_cast(clhip_u32x4, a).x == 0x12345678u ? 1.f : 0.f;      // keeps the operands alive, issues no matrix instruction
    c[1] += __builtin_bit_cast(clhip_u32x4, b).y == 0x12345678u ? 1.f : 0.f;
    return c;
}
#define BW_MFMA(a, b, c) bw_fake_mfma(a, b, c)
#else
#define BW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

struct BwRawX { float4 lo, hi; float l, r; };
struct BwRawD { float4 lo, hi; unsigned codes; };

// Timing-only ablations (tools/experiments; results wrong by design; the product is built with 0): 1 no MFMAs, 2 no splits (the raw
// values are used as operand bits), 4 no loads (one load per segment)
#ifndef BW_ABL
#define BW_ABL 0
#endif
#ifndef BW_SCHED
#define BW_SCHED 1      // 1: the splits of the next rows are interleaved into the MFMA stream of an interior step (sched_group_barrier)
#endif

template <bool UNPOOL>
__global__ __launch_bounds__(256, 1) void bs_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const uint8_t* __restrict__ idx, float* __restrict__ part, int N, int C, int K,
                                                          int H, int W, int splits, long long rows_total, size_t slab_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 31, half = lane >> 5;
    const int split = blockIdx.x % splits, bt = blockIdx.x / splits;
    const int c_tiles = C >> 6;
    const int ct = bt % c_tiles, kt = bt / c_tiles;
    const int k0 = kt * 64 + (wave & 1) * 32, c0 = ct * 64 + (wave >> 1) * 32;
    const int strips = W >> 4;
    // the block's share of the (image, strip, row) list: a contiguous range, rows fastest — whole columns or pieces of columns
    const long long per = rows_total / splits, extra = rows_total % splits;
    const long long r_begin = split * per + (split < extra ? split : extra), r_end = r_begin + per + (split < extra ? 1 : 0);
    const int Hd = UNPOOL ? H >> 1 : H, Wd = UNPOOL ? W >> 1 : W;
    const int plane = H * W, plane_d = Hd * Wd;

    floatx16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float bsum = 0.f;

    for (long long rr = r_begin; rr < r_end;) {
        const int ya = (int)(rr % H);
        const long long colid = rr / H;
        const int sx = (int)(colid % strips), n = (int)(colid / strips);
        const int yb = (int)((long long)H < ya + (r_end - rr) ? H : ya + (r_end - rr));          // rows [ya, yb) of column (n, sx)
        rr += yb - ya;
        const int col0 = 16 * sx + 8 * half;
        const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(x + (size_t)n * C * plane, (size_t)C * plane * 4);
        const __amdgpu_buffer_rsrc_t rs_d = clhip_rsrc(dy + (size_t)n * K * plane_d, (size_t)K * plane_d * 4);
        const __amdgpu_buffer_rsrc_t rs_i = clhip_rsrc(UNPOOL ? idx + (size_t)n * K * plane_d : nullptr, UNPOOL ? (size_t)K * plane_d : 0);
        const int xo = ((c0 + col) * plane + col0) * 4;                   // byte offset of this lane's 8 pixels in row 0
        const int xo_l = col0 > 0 ? xo - 4 : CLHIP_OOB, xo_r = col0 + 8 < W ? xo + 32 : CLHIP_OOB;
        const int dox = UNPOOL ? ((k0 + col) * plane_d + (col0 >> 1)) * 4 : ((k0 + col) * plane_d + col0) * 4;
        const int dio = (k0 + col) * plane_d + (col0 >> 1);

        // rows outside the image (x) / outside [ya, yb) (dy: those rows belong to another block) read as zeros: out-of-range voffset
        auto load_x = [&](int y) {
            BwRawX r;
            const bool ok = (unsigned)y < (unsigned)H;
            const int so = (ok ? y : 0) * W * 4;
            r.lo = clhip_buf_load4(rs_x, ok ? xo : CLHIP_OOB, so);
            r.hi = clhip_buf_load4(rs_x, ok ? xo + 16 : CLHIP_OOB, so);
            r.l = clhip_buf_load(rs_x, ok ? xo_l : CLHIP_OOB, so);
            r.r = clhip_buf_load(rs_x, ok ? xo_r : CLHIP_OOB, so);
            return r;
        };
        auto load_d = [&](int y) {
            BwRawD r;
            const bool ok = y >= ya && y < yb;
            if constexpr (UNPOOL) {
                const int so = ok ? (y >> 1) * Wd : 0;
                r.lo = clhip_buf_load4(rs_d, ok ? dox : CLHIP_OOB, so * 4);
                r.hi = r.lo;
                r.codes = __builtin_amdgcn_raw_buffer_load_b32(rs_i, ok ? dio : CLHIP_OOB, so, 0);
            } else {
                const int so = (ok ? y : 0) * W * 4;
                r.lo = clhip_buf_load4(rs_d, ok ? dox : CLHIP_OOB, so);
                r.hi = clhip_buf_load4(rs_d, ok ? dox + 16 : CLHIP_OOB, so);
                r.codes = 0;
            }
            return r;
        };
        auto split_x = [&](const BwRawX& r) {
            BwB b;
#if BW_ABL & 2
            for (int i = 0; i < 3; ++i) {
                b.p[i][0] = __float_as_uint(r.l); b.p[i][1] = __float_as_uint(r.lo.y); b.p[i][2] = __float_as_uint(r.lo.w);
                b.p[i][3] = __float_as_uint(r.hi.y); b.p[i][4] = __float_as_uint(r.hi.w);
            }
            return b;
#endif
            bw_split2(r.l, r.lo.x, b.p[0][0], b.p[1][0], b.p[2][0]);
            bw_split2(r.lo.y, r.lo.z, b.p[0][1], b.p[1][1], b.p[2][1]);
            bw_split2(r.lo.w, r.hi.x, b.p[0][2], b.p[1][2], b.p[2][2]);
            bw_split2(r.hi.y, r.hi.z, b.p[0][3], b.p[1][3], b.p[2][3]);
            bw_split2(r.hi.w, r.r, b.p[0][4], b.p[1][4], b.p[2][4]);
            return b;
        };
        auto split_d = [&](const BwRawD& r, int y) {
            float v[8];
            if constexpr (UNPOOL) {
                const unsigned a2 = 2u * (unsigned)(y & 1);
                const float p[4] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned cde = (r.codes >> (8 * j)) & 0xffu;
                    v[2 * j] = cde == a2 ? p[j] : 0.f;
                    v[2 * j + 1] = cde == a2 + 1u ? p[j] : 0.f;
                }
            } else {
                v[0] = r.lo.x; v[1] = r.lo.y; v[2] = r.lo.z; v[3] = r.lo.w;
                v[4] = r.hi.x; v[5] = r.hi.y; v[6] = r.hi.z; v[7] = r.hi.w;
            }
            bsum += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            BwA a;
#if BW_ABL & 2
            for (int i = 0; i < 3; ++i) a.p[i] = clhip_u32x4{__float_as_uint(v[0]), __float_as_uint(v[2]), __float_as_uint(v[4]), __float_as_uint(v[6])};
            return a;
#endif
            unsigned q0[4], q1[4], q2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bw_split2(v[2 * j], v[2 * j + 1], q0[j], q1[j], q2[j]);
            a.p[0] = clhip_u32x4{q0[0], q0[1], q0[2], q0[3]};
            a.p[1] = clhip_u32x4{q1[0], q1[1], q1[2], q1[3]};
            a.p[2] = clhip_u32x4{q2[0], q2[1], q2[2], q2[3]};
            return a;
        };
        // acc[3 r + s] += A(row) x B(tap s): the six products, small ones first
        auto row_taps = [&](const BwA& a, const BwB& b, int r) {
#define BW_MM(S, I, J)                                                                                                                  \
    acc[3 * r + S] = BW_MFMA(__builtin_bit_cast(bw_bf16x8, a.p[I]), bw_tap<S>(b, J), acc[3 * r + S])
            BW_MM(0, 0, 2); BW_MM(1, 0, 2); BW_MM(2, 0, 2);
            BW_MM(0, 2, 0); BW_MM(1, 2, 0); BW_MM(2, 2, 0);
            BW_MM(0, 1, 1); BW_MM(1, 1, 1); BW_MM(2, 1, 1);
            BW_MM(0, 0, 1); BW_MM(1, 0, 1); BW_MM(2, 0, 1);
            BW_MM(0, 1, 0); BW_MM(1, 1, 0); BW_MM(2, 1, 0);
            BW_MM(0, 0, 0); BW_MM(1, 0, 0); BW_MM(2, 0, 0);
#undef BW_MM
        };

        // Pipeline state at the top of step yi (x row yi):  b_cur = split x(yi);  am1 / a0 / ap1 = split dy(yi - 1 / yi / yi + 1);
        // rx / rd = the raw x(yi + 1) / dy(yi + 2), in flight.  A step issues the loads of x(yi + 2) and dy(yi + 3), runs the row's MFMAs
        // and splits rx / rd for the next step.  The x rows -1 and H of a column (all zeros) are not run.
        const int ys = ya > 0 ? ya - 1 : 0, ye = yb < H ? yb : H - 1;          // x rows ys .. ye
        // fill the pipeline: the dy window and the x row of step ys, the raw values of step ys + 1 (rows outside the share read zeros)
        BwA am1, a0, ap1;
        BwB b_cur;
        BwRawX rx;
        BwRawD rd;
        {
            const BwRawD dm = load_d(ys - 1), d0 = load_d(ys), d1 = load_d(ys + 1);
            const BwRawX x0 = load_x(ys);
            rx = load_x(ys + 1);
            rd = load_d(ys + 2);
            am1 = split_d(dm, ys - 1);
            a0 = split_d(d0, ys);
            ap1 = split_d(d1, ys + 1);
            b_cur = split_x(x0);
        }
        // one row: loads two rows ahead, the 54 MFMAs of the row in ONE basic block (rows of another block's share / outside the image
        // are zero operands) with the splits of the next row's operands riding in their shadow
        auto step = [&](const BwA& m1, const BwA& z0, const BwA& p1, BwA& a_new, const BwB& bc, BwB& b_new, int yi) {
            const BwRawX rx_cur = rx;
            const BwRawD rd_cur = rd;
#if BW_ABL & 4
            if (yi == ys) { rx = load_x(yi + 2); rd = load_d(yi + 3); }
#else
            rx = load_x(yi + 2);
            rd = load_d(yi + 3);
#endif
            row_taps(p1, bc, 0);
            row_taps(z0, bc, 1);
            row_taps(m1, bc, 2);
            b_new = split_x(rx_cur);
            a_new = split_d(rd_cur, yi + 2);
#if BW_SCHED
#pragma unroll
            for (int i = 0; i < 54; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three VALU
            }
#endif
        };
        // four rows per trip: the dy window (am1, a0, ap1 + the row being split) and the two x buffers rotate by NAME, not by moves
        BwA a3;
        BwB b_alt;
        for (int yi = ys; yi <= ye; yi += 4) {
            step(am1, a0, ap1, a3, b_cur, b_alt, yi);
            if (yi + 1 > ye) break;
            step(a0, ap1, a3, am1, b_alt, b_cur, yi + 1);
            if (yi + 2 > ye) break;
            step(ap1, a3, am1, a0, b_cur, b_alt, yi + 2);
            if (yi + 3 > ye) break;
            step(a3, am1, a0, ap1, b_alt, b_cur, yi + 3);
        }
    }

    // ---- slab: acc[t][i] = dW[t][k0 + row(i, lane)][c0 + col]
    float* slab = part + (size_t)split * slab_stride;
    const __amdgpu_buffer_rsrc_t rs_slab = clhip_out_rsrc(slab);          // (output cache policy: common.hpp)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) clhip_buf_store(acc[t][i], rs_slab, (((t * K + k0 + mfma32_row(i, lane)) * C + c0 + col)) * 4, 0);
    if (ct == 0 && (wave >> 1) == 0) {          // bias sums: the two pixel halves of out-channel k0 + col
        bsum += __shfl_xor(bsum, 32, 64);
        if (half == 0) clhip_buf_store(bsum, rs_slab, (9 * K * C + k0 + col) * 4, 0);
    }
}

}  // namespace

// shapes: whole 64-channel tiles on both sides, rows of whole 16-pixel strips; pooled gradients only on even maps
bool clhip_internal_bs_wgrad_ok(int C, int K, int H, int W) {
    return C >= 64 && C % 64 == 0 && K >= 64 && K % 64 == 0 && W >= 16 && W % 16 == 0 && H >= 1 &&
           (size_t)C * H * W * 4 < 0x7fffffffull && (size_t)K * H * W * 4 < 0x7fffffffull;
}

static long long bs_wgrad_rows(int N, int H, int W) { return (long long)N * (W / 16) * H; }

// one block per CU (a wave holds 144 accumulators + the operand pipeline: one wave per SIMD); the (image, strip, row) list is cut into
// equal contiguous shares, so the blocks of a (k, c) tile finish together
static int bs_wgrad_splits(int N, int C, int K, int H, int W) {
    const int n_bt = (K / 64) * (C / 64);
    const long long rows = bs_wgrad_rows(N, H, W);
    long long s = 256 / n_bt;
    if (s < 1) s = 1;
    if (s > rows) s = rows;
    return (int)s;
}

// ... and where the plan executor prefers it to the Winograd f32 weight gradient (CLHIP_BS_WGRAD=0: nowhere, =2: wherever it runs).
// Measured at N = 200 (profiles/r06_p_bswgrad_builds.txt; us, Winograd / this, plain dy | pooled dy + codes):
//   64 -> 64 @32x32  93.0 / 90.5 | 93.6 / 84.6      64 -> 128 @32x32  154.3 / 145.0 | 163.8 / 138.2     64 -> 64 @16x16  35.0 / 36.8 | 38.3 / 36.0
//   64 -> 128 @16x16  51.0 / 51.2 | 55.4 / 49.2     128 -> 128 @16x16  84.7 / 82.2 | 90.8 / 77.4
//   128 -> 256 @16x16  149.0 / 143.6 | 163.8 / 132.1     256 -> 256 @16x16  277.6 / 264.4 | 305.5 / 240.9
// so: maps of 1024 pixels and more, and 16 x 16 maps from 128 x 128 channels on or behind a pool.  (Layers whose backward runs as one
// merged grid — wino_pair_kernel — keep it: the executor asks for this path only where the weight gradient is a launch of its own.)
bool clhip_internal_bs_wgrad_preferred(int C, int K, int H, int W, int pooled) {
    static const int mode = [] { const char* e = std::getenv("CLHIP_BS_WGRAD"); return e && e[0] >= '0' && e[0] <= '2' ? e[0] - '0' : 1; }();
    if (mode == 0 || !clhip_internal_bs_wgrad_ok(C, K, H, W)) return false;
    if (mode == 2) return true;
    const long long px = (long long)H * W;
    if (px >= 1024) return true;
    return (long long)C * K >= 128LL * 128 || (pooled && C * K >= 64 * 128);
}

size_t clhip_internal_bs_wgrad_ws(int N, int C, int K, int H, int W) {
    if (N <= 0 || !clhip_internal_bs_wgrad_ok(C, K, H, W)) return 0;
    return (size_t)bs_wgrad_splits(N, C, K, H, W) * ((size_t)9 * K * C + K) * 4;
}

// slabs only (job tells the caller's reduction how many); unpool_idx != NULL: dy is the pooled gradient [N][K][H/2][W/2]
int clhip_internal_bs_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N, int C, int K,
                                    int H, int W, void* ws, size_t ws_bytes, hipStream_t s, clhip_wgrad_job* job) {
    if (N <= 0 || !clhip_internal_bs_wgrad_ok(C, K, H, W) || (unpool_idx && ((H | W) & 1))) return CLHIP_ENOTSUP;
    if (!x || !dy || !dw || !ws || !job) return CLHIP_EINVAL;
    if (ws_bytes < clhip_internal_bs_wgrad_ws(N, C, K, H, W)) return CLHIP_ENOSPC;
    const int splits = bs_wgrad_splits(N, C, K, H, W);
    const long long units = bs_wgrad_rows(N, H, W);
    const size_t slab = (size_t)9 * K * C + K;
    const int blocks = (K / 64) * (C / 64) * splits;
    float* part = static_cast<float*>(ws);
    if (unpool_idx) hipLaunchKernelGGL(bs_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, s, x, dy, unpool_idx, part, N, C, K, H, W, splits, units, slab);
    else hipLaunchKernelGGL(bs_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, s, x, dy, unpool_idx, part, N, C, K, H, W, splits, units, slab);
    CLHIP_LAUNCH_CHECK();
    *job = clhip_wgrad_job{part, dw, db, K, C, splits};
    return 0;
}

extern "C" {

// (maps the 16-pixel-aligned kernel does not take — odd widths, 13 x 13 — go to the tap-split kernel of bswgrad5.hip; plain dy only)
size_t clhip_conv3x3_bs_bwd_weight_ws(int N, int C, int K, int H, int W) {
    const size_t a = clhip_internal_bs_wgrad_ws(N, C, K, H, W);
    return a ? a : clhip_internal_bs3k_wgrad_ws(N, C, K, H, W);
}

int clhip_conv3x3_bs_bwd_weight(const float* x, const float* dy, const uint8_t* idx_u8_or_null, float* dw, float* db, int N, int C, int K,
                                int H, int W, void* ws, size_t ws_bytes, void* stream) {
    if (!idx_u8_or_null && N > 0 && !clhip_internal_bs_wgrad_ok(C, K, H, W))
        return clhip_internal_bs3k_wgrad(x, dy, dw, db, N, C, K, H, W, ws, ws_bytes, as_stream(stream));
    clhip_wgrad_job job;
    const int rc = clhip_internal_bs_wgrad_partial(x, dy, idx_u8_or_null, dw, db, N, C, K, H, W, ws, ws_bytes, as_stream(stream), &job);
    if (rc) return rc;
    return clhip_conv3x3_bwd_weight_reduce(ws, dw, db, K, C, job.splits, stream);
}

}  // extern "C"
