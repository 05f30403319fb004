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

__global__ __launch_bounds__(GB) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, size_t n, float alpha,
                                                  int assign) {
    size_t stride = (size_t)gridDim.x * GB;
    for (size_t i = (size_t)blockIdx.x * GB + threadIdx.x; i < n; i += stride)
        y[i] = assign ? alpha * x[i] : y[i] + alpha * x[i];
}

// partial[b][pair] = sum over this block's column range of G[ri][c] * G[rj][c]   (i <= j)
template <int M>
__global__ __launch_bounds__(GB) void gram_partial_kernel(const float* __restrict__ G, size_t ld, RowSel sel, size_t n,
                                                          double* __restrict__ partial) {
    constexpr int NP = M * (M + 1) / 2;
    double acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = 0.0;
    size_t stride = (size_t)gridDim.x * GB;
    for (size_t c = (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
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

__global__ void gram_final_kernel(const double* __restrict__ partial, int nblocks, int M, double* __restrict__ out) {
    const int NP = M * (M + 1) / 2;
    int p = threadIdx.x;
    if (p >= NP) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * NP + p];
    // unpack pair index -> (i, j)
    int i = 0, rem = p;
    while (rem >= M - i) { rem -= M - i; ++i; }
    int j = i + rem;
    out[i * M + j] = s;
    out[j * M + i] = s;
}

__global__ __launch_bounds__(GB) void project_kernel(const float* __restrict__ G, size_t ld, RowSel sel, Coefs cf, int m,
                                                     const float* __restrict__ g, float* __restrict__ out, size_t n) {
    size_t stride = (size_t)gridDim.x * GB;
    for (size_t c = (size_t)blockIdx.x * GB + threadIdx.x; c < n; c += stride) {
        // gem.py:79: x = v . memories + gradient, evaluated in f64 like numpy, rounded once to f32
        double s = (double)g[c];
        for (int i = 0; i < m; ++i) s += (double)cf.v[i] * (double)G[(size_t)sel.idx[i] * ld + c];
        out[c] = (float)s;
    }
}

template <int M>
int gram_launch(const float* G, size_t ld, const RowSel& sel, size_t n, double* partial, int blocks, hipStream_t s) {
    hipLaunchKernelGGL((gram_partial_kernel<M>), dim3(blocks), dim3(GB), 0, s, G, ld, sel, n, partial);
    return 0;
}

}  // namespace

extern "C" {

int clhip_axpy(float* y, const float* x, size_t n, float alpha, int assign, void* stream) {
    if (!y || !x) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), y, x, n, alpha, assign);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

size_t clhip_gem_gram_ws(int m) {
    if (m < 1 || m > MAXM) return 0;
    return (size_t)512 * (m * (m + 1) / 2) * sizeof(double);
}

// out_f64[m*m] (device) = Gram matrix of rows row_idx[0..m) of G (each row n floats, stride ld floats)
int clhip_gem_gram(const float* G, size_t ld, const int* row_idx_host, int m, size_t n, double* out_f64, void* ws,
                   size_t ws_bytes, void* stream) {
    if (!G || !row_idx_host || !out_f64 || !ws || m < 1 || m > MAXM || n == 0 || ws_bytes < clhip_gem_gram_ws(m))
        return CLHIP_EINVAL;
    RowSel sel{};
    for (int i = 0; i < m; ++i) sel.idx[i] = row_idx_host[i];
    hipStream_t s = as_stream(stream);
    int blocks = ew_grid(n, GB);
    if (blocks > 512) blocks = 512;
    double* partial = static_cast<double*>(ws);
    switch (m) {
#define CASE(M_) case M_: gram_launch<M_>(G, ld, sel, n, partial, blocks, s); break;
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        CASE(14) CASE(15) CASE(16)
#undef CASE
    }
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(gram_final_kernel, dim3(1), dim3(256), 0, s, partial, blocks, m, out_f64);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// out[c] = g[c] + sum_i v[i] * G[row_idx[i]][c]      (gem.py:79 + overwrite_grad :38-55 when out = grad arena)
int clhip_gem_project(const float* G, size_t ld, const int* row_idx_host, const float* v_host, int m, const float* g,
                      float* out, size_t n, void* stream) {
    if (!G || !row_idx_host || !v_host || !g || !out || m < 1 || m > MAXM || n == 0) return CLHIP_EINVAL;
    RowSel sel{}; Coefs cf{};
    for (int i = 0; i < m; ++i) { sel.idx[i] = row_idx_host[i]; cf.v[i] = v_host[i]; }
    hipLaunchKernelGGL(project_kernel, dim3(ew_grid(n, GB)), dim3(GB), 0, as_stream(stream), G, ld, sel, cf, m, g, out, n);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
