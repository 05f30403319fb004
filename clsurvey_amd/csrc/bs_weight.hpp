// bs_weight.hpp — the split helpers and the weight-image body of the bf16-split convolutions (bsconv.hip), shared with wino.hip so that
// the Winograd U images and the bf16-split images of a pass are built by ONE launch (clhip_internal_weight_images): a kernel boundary
// between two launches of a few microseconds each costs as much as either.
#pragma once
#include "common.hpp"

namespace {

typedef __bf16 bs_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float bs_f32x2 __attribute__((ext_vector_type(2)));

constexpr int BS_BN = 64;            // output channels per block
constexpr int BS_CK = 16;            // input channels per k-step (K of v_mfma_f32_32x32x16_bf16)

// two floats -> two bf16 (round to nearest even), `lo` in bits 0..15: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned bs_pk(float lo, float hi) {
    bs_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bs_bf16x2));
}
// a = a0 + a1 + a2 exactly (up to 2^-26 |a|): pieces of (a, b) packed pairwise
__device__ __forceinline__ void bs_split2(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = bs_pk(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = bs_pk(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = bs_pk(ra, rb);
}
__device__ __forceinline__ void bs_split8(const float* v, clhip_u32x4& q0, clhip_u32x4& q1, clhip_u32x4& q2) {
    unsigned a[4], b[4], c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bs_split2(v[2 * i], v[2 * i + 1], a[i], b[i], c[i]);
    q0 = clhip_u32x4{a[0], a[1], a[2], a[3]};
    q1 = clhip_u32x4{b[0], b[1], b[2], b[3]};
    q2 = clhip_u32x4{c[0], c[1], c[2], c[3]};
}

// img[nt][chunk][tap][piece][lane] (16 bytes each): lane l of the B operand of n tile nt holds output channel ko = 32 nt + (l & 31)
// and input channels ci = 16 chunk + 8 (l >> 5) + e, e = 0..7, of tap (r, s) — MODE 0: w[ko][ci][r][s]; MODE 1 (backward-data: the
// kernel's input channels are the layer's output channels): w[ci][ko][ks - 1 - r][ks - 1 - s].  Ko / Ci: channel counts as the KERNEL
// sees them; ks x ks taps (3 x 3, or 5 x 5: AlexNet's second convolution, models/net.py:96-125).
// the work of block `lb` (256 threads) of the image of job q — the body of bs_weight_multi_kernel
__device__ __forceinline__ void bs_weight_block(const clhip_wino_wt& q, int lb, int tid) {
    const int n_chunks = (q.Ci + BS_CK - 1) / BS_CK, n_nt = (q.Ko + 31) / 32;
    const int ks = q.pad > 0 ? q.pad : 3, T = ks * ks;         // (clhip_wino_wt::pad carries the kernel size here: 0 = 3)
    const int t = lb * 256 + tid;
    const int lane = t & 63;
    int rest = t >> 6;
    const int tap = rest % T;
    rest /= T;
    const int chunk = rest % n_chunks, nt = rest / n_chunks;
    if (nt >= n_nt) return;
    const int ko = nt * 32 + (lane & 31), ci0 = chunk * BS_CK + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ci = ci0 + e;
        float x = 0.f;
        if (ko < q.Ko && ci < q.Ci)
            x = q.mode == 0 ? q.w[((size_t)ko * q.Ci + ci) * T + tap] : q.w[((size_t)ci * q.Ko + ko) * T + (T - 1 - tap)];
        v[e] = x;
    }
    clhip_u32x4 p0, p1, p2;
    bs_split8(v, p0, p1, p2);
    clhip_u32x4* img = reinterpret_cast<clhip_u32x4*>(q.U) + ((size_t)(nt * n_chunks + chunk) * 3 * T + tap * 3) * 64 + lane;
    img[0] = p0;
    img[64] = p1;
    img[128] = p2;
}
// blocks of the image of one job
static inline int bs_weight_blocks(const clhip_wino_wt& q) {
    const int ks = q.pad > 0 ? q.pad : 3;
    const int total = ((q.Ko + 31) / 32) * ((q.Ci + BS_CK - 1) / BS_CK) * ks * ks * 64;
    return (total + 255) / 256;
}

}  // namespace
