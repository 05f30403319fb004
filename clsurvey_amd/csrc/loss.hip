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

// Same arithmetic as softmax_ce_rows_kernel with the whole [N][ld] logit block staged through LDS: coalesced loads and
// stores instead of three latency-serialised passes of row-strided global reads (11.7 us -> a few us at 200 x 20).
constexpr int CE_LDS_FLOATS = 12288;
__global__ __launch_bounds__(LOSS_BLOCK) void softmax_ce_rows_lds_kernel(
    const float* __restrict__ logits_full, const int64_t* __restrict__ labels, int N, int C, int reduction,
    float* __restrict__ dlogits_full, float* __restrict__ loss_out, double* __restrict__ stats, int ld, int col_off) {
    __shared__ float zs[CE_LDS_FLOATS];
    __shared__ float w_loss[16];
    __shared__ int w_corr[16];
    const int lds = ld | 1;                       // odd row stride: conflict-free per-row walks
    const int total = N * ld;
    for (int e = threadIdx.x; e < total; e += LOSS_BLOCK) {
        const int r = e / ld, c = e - r * ld;
        zs[r * lds + c] = logits_full[e];
    }
    __syncthreads();
    const int row = threadIdx.x;
    const float scale = reduction == 0 ? 1.f / (float)N : 1.f;
    float li = 0.f;
    int ok = 0;
    if (row < N) {
        float* z = zs + row * lds + col_off;
        const int y = (int)labels[row];
        float m = -INFINITY;
        int am = 0;
        for (int c = 0; c < C; ++c) {
            float v = z[c];
            if (v > m) { m = v; am = c; }
        }
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(z[c] - m);
        const float lse = logf(se);
        li = -(z[y] - m - lse);
        ok = (am == y);
        float* zr = zs + row * lds;
        for (int c = 0; c < ld; ++c) {
            const int cc = c - col_off;
            zr[c] = (cc >= 0 && cc < C) ? (expf(zr[c] - m - lse) - (cc == y ? 1.f : 0.f)) * scale : 0.f;
        }
    }
    float v = li;
    int cc = ok;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); cc += __shfl_xor(cc, o, 64); }
    if ((threadIdx.x & 63) == 0) { w_loss[threadIdx.x >> 6] = v; w_corr[threadIdx.x >> 6] = cc; }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += LOSS_BLOCK) {
        const int r = e / ld, c = e - r * ld;
        dlogits_full[e] = zs[r * lds + c];
    }
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


// LwF objective over stacked heads (methods/LwF/main_LWF.py:47-76,184-202): logits [N][ld] hold H heads side by side
// (head h = columns off[h] .. off[h] + size[h]); the LAST head is the new task (CrossEntropy, mean), every earlier head
// h is distilled towards teacher [N][ld_t] (same column layout, the old heads only):
//   L_h = 1/N sum_rows [ log sum_j exp((y_j - max y)/T) - sum_j p_j (y_j - max y)/T ],
//   p = softmax(t - max t)^(1/T) / sum(...)                                  (distillation_loss, :47-76)
//   total = task + lambda * sum_h L_h ;  d total / d y_j = lambda / (N T) (softmax((y - max)/T)_j - p_j) on old heads.
// One thread per row (N <= 1024); rows are summed in a fixed order.  loss_out[0] = task loss (what the reference logs),
// loss_out[1] = lambda * sum_h L_h; stats += (task loss, #correct on the new head).
constexpr int LWF_MAX_HEADS = 32;
struct LwfHeads { int n; int off[LWF_MAX_HEADS]; int size[LWF_MAX_HEADS]; };

__global__ __launch_bounds__(LOSS_BLOCK) void lwf_loss_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                              const float* __restrict__ teacher, LwfHeads hd, int N, int ld, int ld_t,
                                                              float T, float lam, int distill, float* __restrict__ dlogits,
                                                              float* __restrict__ loss_out, double* __restrict__ stats) {
    __shared__ float s_task[LOSS_BLOCK], s_dist[LOSS_BLOCK];
    __shared__ unsigned char s_corr[LOSS_BLOCK];
    const int row = threadIdx.x;
    float task = 0.f, dist = 0.f;
    int ok = 0;
    if (row < N) {
        const float* z = logits + (size_t)row * ld;
        float* dz = dlogits + (size_t)row * ld;
        const float invN = 1.f / (float)N;
        for (int h = 0; h < hd.n; ++h) {
            const int o = hd.off[h], C = hd.size[h];
            float m = -INFINITY; int am = 0;
            for (int c = 0; c < C; ++c) { const float v = z[o + c]; if (v > m) { m = v; am = c; } }
            if (h == hd.n - 1) {
                const int y = (int)labels[row];
                float se = 0.f;
                for (int c = 0; c < C; ++c) se += expf(z[o + c] - m);
                const float lse = logf(se);
                for (int c = 0; c < C; ++c) dz[o + c] = (expf(z[o + c] - m - lse) - (c == y ? 1.f : 0.f)) * invN;
                task = -(z[o + y] - m - lse);
                ok = (am == y);
            } else if (distill) {
                const float* t = teacher + (size_t)row * ld_t + o;
                float mt = -INFINITY;
                for (int c = 0; c < C; ++c) mt = fmaxf(mt, t[c]);
                float st = 0.f;
                for (int c = 0; c < C; ++c) st += expf(t[c] - mt);
                const float invT = 1.f / T;
                float sp = 0.f;                                   // sum_j softmax(t)_j^(1/T)
                for (int c = 0; c < C; ++c) sp += powf(expf(t[c] - mt) / st, invT);
                float sumex = 0.f;
                for (int c = 0; c < C; ++c) sumex += expf((z[o + c] - m) * invT);
                float cross = 0.f;
                const float g = lam * invN * invT;
                for (int c = 0; c < C; ++c) {
                    const float p = powf(expf(t[c] - mt) / st, invT) / sp;
                    const float ys = (z[o + c] - m) * invT;
                    cross += p * ys;
                    dz[o + c] = g * (expf(ys) / sumex - p);
                }
                dist += logf(sumex) - cross;
            } else {
                for (int c = 0; c < C; ++c) dz[o + c] = 0.f;
            }
        }
    }
    s_task[row] = task; s_dist[row] = dist; s_corr[row] = (unsigned char)ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f, d = 0.f; int c = 0;
        for (int i = 0; i < N; ++i) { t += s_task[i]; d += s_dist[i]; c += s_corr[i]; }       // fixed order
        t /= (float)N; d = lam * d / (float)N;
        loss_out[0] = t; loss_out[1] = d;
        if (stats) { stats[0] += (double)t; stats[1] += (double)c; }
    }
}

}  // namespace

extern "C" {

int clhip_softmax_ce_slice(const float* logits, const int64_t* labels_i64, int N, int ld, int col_off, int ncols,
                           int reduction, float* dlogits, float* loss_out, double* stats, void* stream) {
    if (!logits || !labels_i64 || !dlogits || !loss_out || N <= 0 || ncols <= 0 || col_off < 0 || col_off + ncols > ld)
        return CLHIP_EINVAL;
    if (reduction != 0 && reduction != 1) return CLHIP_EINVAL;
    if (ncols <= 64 && N <= LOSS_BLOCK && (size_t)N * (ld | 1) <= (size_t)CE_LDS_FLOATS)
        hipLaunchKernelGGL(softmax_ce_rows_lds_kernel, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, labels_i64, N, ncols,
                           reduction, dlogits, loss_out, stats, ld, col_off);
    else if (ncols <= 64 && N <= LOSS_BLOCK && ld <= 4096)
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

int clhip_lwf_loss(const float* logits, const int64_t* labels_i64, const float* teacher, const int* head_sizes, int n_heads,
                   int N, int ld, int ld_teacher, float T, float reg_lambda, int distill, float* dlogits, float* loss_out2,
                   double* stats, void* stream) {
    if (!logits || !labels_i64 || !head_sizes || !dlogits || !loss_out2 || n_heads < 1 || n_heads > LWF_MAX_HEADS) return CLHIP_EINVAL;
    if (N <= 0 || N > LOSS_BLOCK || (distill && n_heads > 1 && !teacher) || T <= 0.f) return CLHIP_EINVAL;
    LwfHeads hd; hd.n = n_heads;
    int off = 0;
    for (int h = 0; h < n_heads; ++h) { if (head_sizes[h] <= 0) return CLHIP_EINVAL; hd.off[h] = off; hd.size[h] = head_sizes[h]; off += head_sizes[h]; }
    if (off > ld || (distill && n_heads > 1 && off - head_sizes[n_heads - 1] > ld_teacher)) return CLHIP_EINVAL;
    hipLaunchKernelGGL(lwf_loss_kernel, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, labels_i64, teacher, hd, N, ld,
                       ld_teacher, T, reg_lambda, distill, dlogits, loss_out2, stats);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_mse_zero_sum(const float* logits, size_t n, float* dlogits, float* loss_out, void* stream) {
    if (!logits || !dlogits || !loss_out || n == 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(mse_zero_sum_kernel, dim3(1), dim3(LOSS_BLOCK), 0, as_stream(stream), logits, n, dlogits, loss_out);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
