// GEM memory-gradient kernels — restates methods/rehearsal/model/gem.py:20-80 (store_grad,
// overwrite_grad, project2cone2) and :275-277 (violation test).
//
// The reference keeps grads as a [P][n_tasks] matrix (column per task), copies it to the CPU and runs
// numpy float64 GEMMs over P for the QP inputs (gem.py:58-76).  Here a task's gradient is one
// CONTIGUOUS row of G[n_tasks][ld] (the ParamArena's gradient is already flat, so store_grad is a
// single axpy/copy), and ONE pass over the m selected rows yields the whole m x m Gram matrix in f64:
// 4*m bytes per parameter instead of the 4*(t+1) + 8*t*... of the numpy path.  Only the m*m doubles
// travel to the host for the tiny QP.
#include "common.hpp"

namespace {

constexpr int GB = 256;
constexpr int MAXM = 16;

struct RowSel { int idx[MAXM]; };
struct Coefs { float v[MAXM]; };

// VEC: 16-byte accesses (the arena gradient and the rows of G are 16-byte aligned, ld is a multiple of 4); the tail
// n % 4 is done by the first threads.  Per-element arithmetic is the same either way.
template <bool VEC>
__global__ __launch_bounds__(GB) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n, float alpha,
                                                  int assign) {
    const size_t stride = (size_t)gridDim.x * GB, t0 = (size_t)blockIdx.x * GB + threadIdx.x;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t i = t0; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        float4 b = assign ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<const float4*>(y)[i];
        b.x = assign ? alpha * a.x : b.x + alpha * a.x; b.y = assign ? alpha * a.y : b.y + alpha * a.y;
        b.z = assign ? alpha * a.z : b.z + alpha * a.z; b.w = assign ? alpha * a.w : b.w + alpha * a.w;
        reinterpret_cast<float4*>(y)[i] = b;
    }
    for (size_t i = n4 * 4 + t0; i < n; i += stride)
        y[i] = assign ? alpha * x[i] : y[i] + alpha * x[i];
}

// partial[b][pair] = sum over this block's column range of G[ri][c] * G[rj][c]   (i <= j)
template <int M, bool VEC>
__global__ __launch_bounds__(GB) void gram_partial_kernel(const float* __restrict__ G, size_t ld, RowSel sel, size_t n,
                                                          double* __restrict__ partial) {
    constexpr int NP = M * (M + 1) / 2;
    double acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = 0.0;
    size_t stride = (size_t)gridDim.x * GB;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t c4 = (size_t)blockIdx.x * GB + threadIdx.x; c4 < n4; c4 += stride) {
        float4 v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = reinterpret_cast<const float4*>(G + (size_t)sel.idx[i] * ld)[c4];
        int p = 0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) {
                acc[p] += (double)v[i].x * (double)v[j].x;
                acc[p] += (double)v[i].y * (double)v[j].y;
                acc[p] += (double)v[i].z * (double)v[j].z;
                acc[p] += (double)v[i].w * (double)v[j].w;
                ++p;
            }
    }
    for (size_t c = n4 * 4 + (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
        float v[M];
#pragma unroll
        for (int i = 0; i < M; ++i) v[i] = G[(size_t)sel.idx[i] * ld + c];
        int p = 0;
#pragma unroll
        for (int i = 0; i < M; ++i)
#pragma unroll
            for (int j = i; j < M; ++j) acc[p++] += (double)v[i] * (double)v[j];
    }
    __shared__ double red[GB];
    for (int p = 0; p < NP; ++p) {
        red[threadIdx.x] = acc[p];
        __syncthreads();
        for (int o = GB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(size_t)blockIdx.x * NP + p] = red[0];
        __syncthreads();
    }
}

// One block per pair (i <= j): 256 threads sum the per-block partials b = t, t + 256, ... and a fixed tree adds the 256
// sums (a single thread walking 2048 partials is 2048 dependent memory round trips: it took longer than the pass over G).
__global__ __launch_bounds__(256) void gram_final_kernel(const double* __restrict__ partial, int nblocks, int M,
                                                         double* __restrict__ out) {
    __shared__ double red[256];
    const int NP = M * (M + 1) / 2;
    const int p = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(size_t)b * NP + p];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int i = 0, rem = p;                       // unpack pair index -> (i, j)
        while (rem >= M - i) { rem -= M - i; ++i; }
        const int j = i + rem;
        out[i * M + j] = red[0];
        out[j * M + i] = red[0];
    }
}

template <bool VEC>
__global__ __launch_bounds__(GB) void project_kernel(const float* __restrict__ G, size_t ld, RowSel sel, Coefs cf, int m,
                                                     const float* __restrict__ g, float* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * GB;
    const size_t n4 = VEC ? n / 4 : 0;
    for (size_t c4 = (size_t)blockIdx.x * GB + threadIdx.x; c4 < n4; c4 += stride) {
        const float4 gv = reinterpret_cast<const float4*>(g)[c4];
        double s0 = (double)gv.x, s1 = (double)gv.y, s2 = (double)gv.z, s3 = (double)gv.w;
        for (int i = 0; i < m; ++i) {
            const float4 r = reinterpret_cast<const float4*>(G + (size_t)sel.idx[i] * ld)[c4];
            const double vi = (double)cf.v[i];
            s0 += vi * (double)r.x; s1 += vi * (double)r.y; s2 += vi * (double)r.z; s3 += vi * (double)r.w;
        }
        reinterpret_cast<float4*>(out)[c4] = make_float4((float)s0, (float)s1, (float)s2, (float)s3);
    }
    for (size_t c = n4 * 4 + (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
        // gem.py:79: x = v . memories + gradient, evaluated in f64 like numpy, rounded once to f32
        double s = (double)g[c];
        for (int i = 0; i < m; ++i) s += (double)cf.v[i] * (double)G[(size_t)sel.idx[i] * ld + c];
        out[c] = (float)s;
    }
}

template <int M>
int gram_launch(const float* G, size_t ld, const RowSel& sel, size_t n, double* partial, int blocks, bool vec, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((gram_partial_kernel<M, true>), dim3(blocks), dim3(GB), 0, s, G, ld, sel, n, partial);
    else hipLaunchKernelGGL((gram_partial_kernel<M, false>), dim3(blocks), dim3(GB), 0, s, G, ld, sel, n, partial);
    return 0;
}

constexpr int GRAM_BLOCKS = 2048;

}  // namespace

extern "C" {

int clhip_axpy(float* y, const float* x, size_t n, float alpha, int assign, void* stream) {
    if (!y || !x) return CLHIP_EINVAL;
    if (n == 0) return 0;
    if (aligned16(y) && aligned16(x))
        hipLaunchKernelGGL(axpy_kernel<true>, dim3(ew_grid(n / 4 + 1, GB)), dim3(GB), 0, as_stream(stream), y, x, n, alpha, assign);
    else
        hipLaunchKernelGGL(axpy_kernel<false>, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), y, x, n, alpha, assign);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

size_t clhip_gem_gram_ws(int m) {
    if (m < 1 || m > MAXM) return 0;
    return (size_t)GRAM_BLOCKS * (m * (m + 1) / 2) * sizeof(double);
}

// out_f64[m*m] (device) = Gram matrix of rows row_idx[0..m) of G (each row n floats, stride ld floats)
int clhip_gem_gram(const float* G, size_t ld, const int* row_idx_host, int m, size_t n, double* out_f64, void* ws,
                   size_t ws_bytes, void* stream) {
    if (!G || !row_idx_host || !out_f64 || !ws || m < 1 || m > MAXM || n == 0 || ws_bytes < clhip_gem_gram_ws(m))
        return CLHIP_EINVAL;
    RowSel sel{};
    for (int i = 0; i < m; ++i) sel.idx[i] = row_idx_host[i];
    hipStream_t s = as_stream(stream);
    const bool vec = aligned16(G) && (ld % 4 == 0);
    int blocks = ew_grid(vec ? n / 4 + 1 : n, GB);
    if (blocks > GRAM_BLOCKS) blocks = GRAM_BLOCKS;
    double* partial = static_cast<double*>(ws);
    switch (m) {
#define CASE(M_) case M_: gram_launch<M_>(G, ld, sel, n, partial, blocks, vec, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16)
#undef CASE
    }
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_final_kernel, dim3(m * (m + 1) / 2), dim3(256), 0, s, partial, blocks, m, out_f64);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// out[c] = g[c] + sum_i v[i] * G[row_idx[i]][c]      (gem.py:79 + overwrite_grad :38-55 when out = grad arena)
int clhip_gem_project(const float* G, size_t ld, const int* row_idx_host, const float* v_host, int m, const float* g,
                      float* out, size_t n, void* stream) {
    if (!G || !row_idx_host || !v_host || !g || !out || m < 1 || m > MAXM || n == 0) return CLHIP_EINVAL;
    RowSel sel{}; Coefs cf{};
    for (int i = 0; i < m; ++i) { sel.idx[i] = row_idx_host[i]; cf.v[i] = v_host[i]; }
    if (aligned16(G) && (ld % 4 == 0) && aligned16(g) && aligned16(out))
        hipLaunchKernelGGL(project_kernel<true>, dim3(ew_grid(n / 4 + 1, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, cf, m, g, out, n);
    else
        hipLaunchKernelGGL(project_kernel<false>, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, cf, m, g, out, n);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
