// The strided MFMA GEMM tile of gemm.hip as a device function, so that fc_chain.hip can run it next to the classifier's
// weight-gradient tiles in one launch (same code => same bits as the stand-alone gemm_mfma_kernel launch).
#pragma once
#include "common.hpp"

namespace {

constexpr int TM = 64, TN = 64, BK = 32, LD = 65;

// AK: A is contiguous along k (sak == 1); BKc: B is contiguous along k (sbk == 1)
// one 64x64 output tile (and K split) of the strided GEMM; `block` = tile * splits + split
// BKT: depth of a staged chunk (32).  Staging the whole k range of the batch-200 classifier shapes at once (BKT = 128, one
// barrier, 64 MFMAs) was measured SLOWER than four pipelined 32-deep chunks: 11.5 vs 8.4 us forward, 14.1 vs 10.8 us
// backward-data — the 64 scalar loads per thread in one burst take longer to land than the chunked loop hides.  The MFMA
// sequence (and so every bit of the result) does not depend on BKT.
template <bool AK, bool BKc, int BKT = BK>
__device__ __forceinline__ void gemm_tile(const clhip_gemm_args& g, int block) {
    const float* __restrict__ a = g.a; const float* __restrict__ b = g.b; float* __restrict__ out = g.out;
    const int M = g.M, N = g.N, K = g.K, n_tiles = g.n_tiles, splits = g.splits, k_per_split = g.k_per_split, relu = g.relu;
    const long sam = g.sam, sak = g.sak, sbk = g.sbk, sbn = g.sbn;
    const float* __restrict__ bias = g.bias; const float* __restrict__ mask_src = g.mask_src;
    __shared__ float as[BKT * LD];
    __shared__ float bs[BKT * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1, li = lane & 31, kk = lane >> 5;
    const int split = block % splits;
    const int tile = block / splits;
    const int tn = tile % n_tiles, tm = tile / n_tiles;
    const int m0 = tm * TM, n0 = tn * TN;
    const int k_begin = split * k_per_split;
    const int k_end = min(K, k_begin + k_per_split);

    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // register-staged software pipeline: chunk t+1 is in flight while chunk t feeds the MFMAs
    constexpr int A_IT = (TM * BKT) / 256, B_IT = (TN * BKT) / 256;
    float ar[A_IT], br[B_IT];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int e = tid + 256 * j;
            int ml, kl;
            if (AK) { ml = e / BKT; kl = e - ml * BKT; } else { kl = e / TM; ml = e - kl * TM; }
            int m = m0 + ml, k = k0 + kl;
            ar[j] = (m < M && k < k_end) ? a[(long)m * sam + (long)k * sak] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int e = tid + 256 * j;
            int nl, kl;
            if (BKc) { nl = e / BKT; kl = e - nl * BKT; } else { kl = e / TN; nl = e - kl * TN; }
            int n = n0 + nl, k = k0 + kl;
            br[j] = (n < N && k < k_end) ? b[(long)k * sbk + (long)n * sbn] : 0.f;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            int e = tid + 256 * j;
            int ml, kl;
            if (AK) { ml = e / BKT; kl = e - ml * BKT; } else { kl = e / TM; ml = e - kl * TM; }
            as[kl * LD + ml] = ar[j];
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            int e = tid + 256 * j;
            int nl, kl;
            if (BKc) { nl = e / BKT; kl = e - nl * BKT; } else { kl = e / TN; nl = e - kl * TN; }
            bs[kl * LD + nl] = br[j];
        }
    };
    if (k_begin < k_end) load_chunk(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += BKT) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (k0 + BKT < k_end) load_chunk(k0 + BKT);
        // the MFMA sequence is that of 32-deep chunks whatever BKT is: ceil(range / 32) x 16 steps, zero padded
        const int sub = (min(BKT, k_end - k0) + BK - 1) / BK;
        for (int sc = 0; sc < sub; ++sc) {
#pragma unroll
            for (int k2 = 0; k2 < BK; k2 += 2) {
                float av = as[(sc * BK + k2 + kk) * LD + wm * 32 + li];
                float bv = bs[(sc * BK + k2 + kk) * LD + wn * 32 + li];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
        }
    }

    const int n = n0 + wn * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = m0 + wm * 32 + mfma32_row(r, lane);
        if (m < M && n < N) {
            float v = acc[r];
            if (splits == 1) {
                if (bias) v += bias[n];
                if (relu) v = fmaxf(v, 0.f);
                if (mask_src) v = mask_src[(size_t)m * N + n] > 0.f ? v : 0.f;
                out[(size_t)m * N + n] = v;
            } else {
                out[((size_t)split * M + m) * N + n] = v;
            }
        }
    }
}


}  // namespace
