// Loss kernels: softmax cross-entropy (mean / sum) and the MAS sum-of-squares objective.
// Tiny tensors ([N<=~1k][C<=1k]); one block, fixed reduction order => deterministic.
#include "common.hpp"

namespace {

constexpr int LOSS_BLOCK = 1024;  // 16 waves

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One wave per row, rows strided over the 16 waves of the single block.  The tensor is tiny, so
// the kernel is pure latency: rows are fetched RB at a time per wave (all loads in flight
// together) before any arithmetic, instead of one dependent round trip per row.
template <int RB>
__global__ __launch_bounds__(LOSS_BLOCK) void softmax_ce_kernel(
    const float* __restrict__ logits_full, const int64_t* __restrict__ labels, int N, int C, int reduction,
    float* __restrict__ dlogits_full, float* __restrict__ loss_out, double* __restrict__ stats, int ld, int col_off) {
    // logits_full is [N][ld]; the loss is taken over columns [col_off, col_off + C) (labels are relative to
    // the slice); dlogits outside the slice are written as 0.
    const float* logits = logits_full + col_off;
    float* dlogits = dlogits_full + col_off;
    if (ld != C) {
        for (size_t i = threadIdx.x; i < (size_t)N * ld; i += LOSS_BLOCK) {
            int c = (int)(i % ld);
            if (c < col_off || c >= col_off + C) dlogits_full[i] = 0.f;
        }
    }
    __shared__ float s_loss[16];
    __shared__ int s_corr[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float scale = reduction == 0 ? 1.f / (float)N : 1.f;
    float wl = 0.f;   // this wave's loss sum (rows in increasing order)
    int wc = 0;
    const bool narrow = C <= 64;
    for (int row0 = wave; row0 < N; row0 += 16 * RB) {
        float zr[RB];
        int yr[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            int row = row0 + 16 * b;
            zr[b] = (narrow && row < N && lane < C) ? logits[(size_t)row * ld + lane] : -INFINITY;
            yr[b] = (row < N) ? (int)labels[row] : 0;
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            int row = row0 + 16 * b;
            if (row >= N) break;
            const float* z = logits + (size_t)row * ld;
            const int y = yr[b];
            float m = -INFINITY;
            int am = 0x7fffffff;
            if (narrow) {
                m = zr[b];
                am = lane < C ? lane : 0x7fffffff;
            } else {
                for (int c = lane; c < C; c += 64) {
                    float v = z[c];
                    if (v > m) { m = v; am = c; }
                }
            }
            float gm = wave_max(m);
            // first index attaining the max (torch.max tie rule: lowest index)
            int cand = (m == gm) ? am : 0x7fffffff;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
            float se = 0.f, zy;
            if (narrow) {
                float ex = lane < C ? expf(zr[b] - gm) : 0.f;
                se = wave_sum(ex);
                const float lse = logf(se);
                if (lane < C) dlogits[(size_t)row * ld + lane] = (expf(zr[b] - gm - lse) - (lane == y ? 1.f : 0.f)) * scale;
                zy = __shfl(zr[b], y, 64);
                wl += -(zy - gm - lse);
            } else {
                for (int c = lane; c < C; c += 64) se += expf(z[c] - gm);
                se = wave_sum(se);
                const float lse = logf(se);
                for (int c = lane; c < C; c += 64)
                    dlogits[(size_t)row * ld + c] = (expf(z[c] - gm - lse) - (c == y ? 1.f : 0.f)) * scale;
                zy = z[y];
                wl += -(zy - gm - lse);
            }
            wc += (cand == y) ? 1 : 0;
        }
    }
    if (lane == 0) { s_loss[wave] = wl; s_corr[wave] = wc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f; int c = 0;
        for (int i = 0; i < 16; ++i) { t += s_loss[i]; c += s_corr[i]; }
        t *= scale;
        loss_out[0] = t;
        if (stats) { stats[0] += (double)t; stats[1] += (double)c; }
    }
}

// Narrow heads (C <= 64, the 20-way task heads): one THREAD per row, no cross-lane traffic; rows are
// summed in index order by thread 0 afterwards (deterministic).  N <= 1024.
__global__ __launch_bounds__(LOSS_BLOCK) void softmax_ce_rows_kernel(
    const float* __restrict__ logits_full, const int64_t* __restrict__ labels, int N, int C, int reduction,
    float* __restrict__ dlogits_full, float* __restrict__ loss_out, double* __restrict__ stats, int ld, int col_off) {
    __shared__ float s_loss[LOSS_BLOCK];
    __shared__ unsigned char s_corr[LOSS_BLOCK];
    const int row = threadIdx.x;
    const float scale = reduction == 0 ? 1.f / (float)N : 1.f;
    float li = 0.f;
    int ok = 0;
    if (row < N) {
        const float* z = logits_full + (size_t)row * ld + col_off;
        float* dz = dlogits_full + (size_t)row * ld;
        const int y = (int)labels[row];
        float m = -INFINITY;
        int am = 0;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            float v = z[c];
            if (v > m) { m = v; am = c; }
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - m);
        const float lse = logf(se);
        for (int c = 0; c < ld; ++c) {
            const int cc = c - col_off;
            dz[c] = (cc >= 0 && cc < C) ? (expf(z[cc] - m - lse) - (cc == y ? 1.f : 0.f)) * scale : 0.f;
        }
        li = -(z[y] - m - lse);
        ok = (am == y);
    }
    s_loss[row] = li;
    s_corr[row] = (unsigned char)ok;
    __syncthreads();
    // fixed-order reduction: wave w sums rows w*64 .. w*64+63 with a butterfly, thread 0 adds the 16 wave sums
    __shared__ float w_loss[16];
    __shared__ int w_corr[16];
    float v = s_loss[row];
    int cc = s_corr[row];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); cc += __shfl_xor(cc, o, 64); }
    if ((threadIdx.x & 63) == 0) { w_loss[threadIdx.x >> 6] = v; w_corr[threadIdx.x >> 6] = cc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f; int c = 0;
        for (int w = 0; w < 16; ++w) { t += w_loss[w]; c += w_corr[w]; }
        t *= scale;
        loss_out[0] = t;
        if (stats) { stats[0] += (double)t; stats[1] += (double)c; }
    }
}

__global__ __launch_bounds__(LOSS_BLOCK) void mse_zero_sum_kernel(const float* __restrict__ z, size_t n,
                                                                  float* __restrict__ dz, float* __restrict__ loss_out) {
    __shared__ float s_part[16];
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < n; i += LOSS_BLOCK) {
        float v = z[i];
        acc += v * v;
        dz[i] = 2.f * v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += s_part[i];
        loss_out[0] = t;
    }
}

}  // namespace

extern "C" {

int clhip_softmax_ce_slice(const float* logits, const int64_t* labels_i64, int N, int ld, int col_off, int ncols,
                           int reduction, float* dlogits, float* loss_out, double* stats, void* stream) {
    if (!logits || !labels_i64 || !dlogits || !loss_out || N <= 0 || ncols <= 0 || col_off < 0 || col_off + ncols > ld)
        return CLHIP_EINVAL;
    if (reduction != 0 && reduction != 1) return CLHIP_EINVAL;
    if (ncols <= 64 && N <= LOSS_BLOCK && ld <= 4096)
        hipLaunchKernelGGL(softmax_ce_rows_kernel, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, labels_i64, N, ncols,
                           reduction, dlogits, loss_out, stats, ld, col_off);
    else
        hipLaunchKernelGGL(softmax_ce_kernel<16>, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, labels_i64, N, ncols,
                           reduction, dlogits, loss_out, stats, ld, col_off);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_softmax_ce(const float* logits, const int64_t* labels_i64, int N, int C, int reduction,
                     float* dlogits, float* loss_out, double* stats, void* stream) {
    return clhip_softmax_ce_slice(logits, labels_i64, N, C, 0, C, reduction, dlogits, loss_out, stats, stream);
}

int clhip_mse_zero_sum(const float* logits, size_t n, float* dlogits, float* loss_out, void* stream) {
    if (!logits || !dlogits || !loss_out || n == 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(mse_zero_sum_kernel, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, n, dlogits, loss_out);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
