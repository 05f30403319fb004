// BatchNorm2d (+ReLU) for the '_BN' model variants (models/VGGSlim.py:27-40: conv -> BatchNorm2d -> ReLU).
//
// HBM-bound, NCHW in place: per-channel statistics are two-stage (BN_SPLIT image ranges per channel in f64, then a
// fixed-order finish) so they fill the chip and stay bitwise run-to-run deterministic; the normalisation and its
// backward are one pass each.  torch semantics (nn.BatchNorm2d defaults): training mode normalises with the batch
// mean and the BIASED variance and moves running_mean / running_var (UNBIASED variance) by `momentum`; eval mode uses
// the running statistics.  Backward in training mode:
//   dbeta = sum dyr, dgamma = sum dyr * xhat, dz = gamma * invstd * (dyr - dbeta / M - xhat * dgamma / M), M = N * HW,
// with dyr = dy * (y > 0) when the ReLU is fused; eval mode: dz = gamma * invstd * dyr.
#include "common.hpp"

namespace {

constexpr int BN_SPLIT = 16;

__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// part[(c * BN_SPLIT + sp) * 2 + {0,1}] = sum z, sum z^2 over the images of split sp
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ z, double* __restrict__ part, int N, int C, int HW) {
    __shared__ double red[256];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int n0 = (int)((long)N * sp / BN_SPLIT), n1 = (int)((long)N * (sp + 1) / BN_SPLIT);
    double s = 0.0, q = 0.0;
    for (int n = n0; n < n1; ++n) {
        const float* row = z + ((size_t)n * C + c) * HW;
        for (int e = threadIdx.x; e < HW; e += 256) {
            const double v = (double)row[e];
            s += v; q += v * v;
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { part[((size_t)c * BN_SPLIT + sp) * 2] = s; part[((size_t)c * BN_SPLIT + sp) * 2 + 1] = q; }
}

__global__ void bn_stats_finish_kernel(const double* __restrict__ part, float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                       float* __restrict__ running_mean, float* __restrict__ running_var, int C, double M,
                                       float momentum, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < BN_SPLIT; ++i) { s += part[((size_t)c * BN_SPLIT + i) * 2]; q += part[((size_t)c * BN_SPLIT + i) * 2 + 1]; }
    const double mean = s / M;
    double var = q / M - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}

__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                     float* __restrict__ save_mean, float* __restrict__ save_invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save_mean[c] = running_mean[c];
    save_invstd[c] = 1.f / sqrtf(running_var[c] + eps);
}

// y = (z - mean) * invstd * gamma + beta (ReLU optional); one (image, channel) plane per blockIdx.x
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_invstd, float* __restrict__ y, int C, int HW,
                                                       int relu) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float mu = save_mean[c], is = save_invstd[c], g = gamma[c], b = beta[c];
    const float* zi = z + plane * HW;
    float* yo = y + plane * HW;
    for (int e = threadIdx.x; e < HW; e += 256) {
        float v = (zi[e] - mu) * is * g + b;
        if (relu) v = fmaxf(v, 0.f);
        yo[e] = v;
    }
}

// part[(c * BN_SPLIT + sp) * 2 + {0,1}] = sum dyr, sum dyr * xhat
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                            const float* __restrict__ z, const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd, double* __restrict__ part, int N,
                                                            int C, int HW, int relu) {
    __shared__ double red[256];
    const int c = blockIdx.x, sp = blockIdx.y;
    const int n0 = (int)((long)N * sp / BN_SPLIT), n1 = (int)((long)N * (sp + 1) / BN_SPLIT);
    const float mu = save_mean[c], is = save_invstd[c];
    double s = 0.0, q = 0.0;
    for (int n = n0; n < n1; ++n) {
        const size_t o = ((size_t)n * C + c) * HW;
        for (int e = threadIdx.x; e < HW; e += 256) {
            float d = dy[o + e];
            if (relu && !(y[o + e] > 0.f)) d = 0.f;
            s += (double)d;
            q += (double)d * (double)((z[o + e] - mu) * is);
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) { part[((size_t)c * BN_SPLIT + sp) * 2] = s; part[((size_t)c * BN_SPLIT + sp) * 2 + 1] = q; }
}

__global__ void bn_bwd_finish_kernel(const double* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                     float* __restrict__ sums, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int i = 0; i < BN_SPLIT; ++i) { s += part[((size_t)c * BN_SPLIT + i) * 2]; q += part[((size_t)c * BN_SPLIT + i) * 2 + 1]; }
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    sums[2 * c] = (float)s;
    sums[2 * c + 1] = (float)q;
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           const float* __restrict__ z, const float* __restrict__ gamma,
                                                           const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                           const float* __restrict__ sums, float* __restrict__ dz, int C, int HW,
                                                           float inv_m, int training, int relu) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const float mu = save_mean[c], is = save_invstd[c], a = gamma[c] * is;
    const float sb = training ? sums[2 * c] * inv_m : 0.f, sg = training ? sums[2 * c + 1] * inv_m : 0.f;
    const size_t o = plane * HW;
    for (int e = threadIdx.x; e < HW; e += 256) {
        float d = dy[o + e];
        if (relu && !(y[o + e] > 0.f)) d = 0.f;
        const float xh = (z[o + e] - mu) * is;
        dz[o + e] = a * (d - sb - xh * sg);
    }
}

}  // namespace

extern "C" {

size_t clhip_bn_ws(int C) { return C > 0 ? ((size_t)C * BN_SPLIT * 2 * sizeof(double) + (size_t)C * 2 * sizeof(float) + 256) : 0; }

int clhip_bn_fwd(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                 float* save_mean, float* save_invstd, int N, int C, int HW, int training, float momentum, float eps, int relu,
                 void* ws, size_t ws_bytes, void* stream) {
    if (!z || !gamma || !beta || !y || !save_mean || !save_invstd || N <= 0 || C <= 0 || HW <= 0) return CLHIP_EINVAL;
    if (!training && (!running_mean || !running_var)) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    if (training) {
        if (!ws || ws_bytes < clhip_bn_ws(C)) return CLHIP_ENOSPC;
        double* part = static_cast<double*>(ws);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(C, BN_SPLIT), dim3(256), 0, s, z, part, N, C, HW);
        CLHIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, save_mean, save_invstd, running_mean,
                           running_var, C, (double)N * HW, momentum, eps);
    } else {
        hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, s, running_mean, running_var, save_mean,
                           save_invstd, C, eps);
    }
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((size_t)N * C)), dim3(256), 0, s, z, gamma, beta, save_mean, save_invstd, y, C, HW,
                       relu);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_bn_bwd(const float* dy, const float* y, const float* z, const float* gamma, const float* save_mean,
                 const float* save_invstd, float* dz, float* dgamma, float* dbeta, int N, int C, int HW, int training, int relu,
                 void* ws, size_t ws_bytes, void* stream) {
    if (!dy || !z || !gamma || !save_mean || !save_invstd || !dz || N <= 0 || C <= 0 || HW <= 0 || (relu && !y)) return CLHIP_EINVAL;
    if (!ws || ws_bytes < clhip_bn_ws(C)) return CLHIP_ENOSPC;
    hipStream_t s = as_stream(stream);
    double* part = static_cast<double*>(ws);
    float* sums = reinterpret_cast<float*>(part + (size_t)C * BN_SPLIT * 2);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C, BN_SPLIT), dim3(256), 0, s, dy, y, z, save_mean, save_invstd, part, N, C, HW, relu);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, s, part, dgamma, dbeta, sums, C);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((size_t)N * C)), dim3(256), 0, s, dy, y, z, gamma, save_mean, save_invstd, sums, dz,
                       C, HW, 1.f / ((float)N * (float)HW), training, relu);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
