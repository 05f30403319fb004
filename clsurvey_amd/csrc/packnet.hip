// PackNet mask kernels (uint8 task-index masks, bit-exact) — restates methods/packnet/prune.py and
// packnetSGD.py.  mask value = owning task index (1-based), 0 = free / pruned.  All kernels are
// HBM-bound single passes (5-13 B per weight); the k-th-magnitude cutoff is an exact 4-pass radix
// select on the fp32 bit pattern of |w| (prune.py:39 moves the tensor to the CPU for kthvalue).
#include "common.hpp"

namespace {

constexpr int PB = 256;

__global__ __launch_bounds__(PB) void finetune_mask_kernel(uint8_t* __restrict__ m, size_t n, uint8_t cur) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (m[i] == 0) m[i] = cur;                                        // prune.py:153
}

// ws layout (uint32): [0] prefix, [1] remaining k, [2] shift of the current digit, [3] count of
// candidates (first pass), [4..259] histogram
__global__ __launch_bounds__(PB) void kth_hist_kernel(const float* __restrict__ w, const uint8_t* __restrict__ m,
                                                      size_t n, uint8_t cur, uint32_t* __restrict__ ws) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = ws[0];
    const uint32_t shift = ws[2];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (m[i] != cur) continue;
        uint32_t key = __float_as_uint(fabsf(w[i]));
        // digits above `shift+8` must equal the prefix found so far
        bool match = (shift == 24) || ((key >> (shift + 8)) == prefix);
        if (match) atomicAdd(&h[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&ws[4 + threadIdx.x], h[threadIdx.x]);
}

__global__ void kth_select_kernel(uint32_t* __restrict__ ws, float* __restrict__ out, int* __restrict__ status) {
    if (threadIdx.x != 0) return;
    uint32_t k = ws[1];
    uint32_t cum = 0;
    int b = 0;
    for (; b < 256; ++b) {
        uint32_t c = ws[4 + b];
        if (cum + c >= k) break;
        cum += c;
    }
    if (b == 256) { if (status) status[0] = 1; b = 255; }     // k larger than the candidate count
    ws[0] = (ws[0] << 8) | (uint32_t)b;
    ws[1] = k - cum;
    for (int i = 0; i < 256; ++i) ws[4 + i] = 0;
    if (ws[2] == 0) out[0] = __uint_as_float(ws[0]);
    else ws[2] -= 8;
}

__global__ void kth_init_kernel(uint32_t* __restrict__ ws, uint32_t k, int* __restrict__ status) {
    int i = threadIdx.x;
    if (i < 260) ws[i] = 0;
    if (i + 256 < 260) ws[i + 256] = 0;
    __syncthreads();
    if (i == 0) { ws[1] = k; ws[2] = 24; if (status) status[0] = 0; }
}

__global__ __launch_bounds__(PB) void prune_kernel(float* __restrict__ w, uint8_t* __restrict__ m, size_t n, uint8_t cur,
                                                   const float* __restrict__ cutoff) {
    const float cut = cutoff[0];
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t mi = m[i];
        float wi = w[i];
        if (mi == cur && fabsf(wi) <= cut) { mi = 0; m[i] = 0; }          // prune.py:43-47
        if (mi == 0 && wi != 0.f) w[i] = 0.f;                             // prune.py:70-71
        else if (mi == 0) w[i] = 0.f;                                     // also turns -0.0 into +0.0 as masked assignment does
    }
}

__global__ __launch_bounds__(PB) void mask_grad_zero_kernel(float* __restrict__ g, const uint8_t* __restrict__ m, size_t n,
                                                            uint8_t cur) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        if (m[i] != cur) g[i] = 0.f;                                      // prune.py:83-88
}

__global__ __launch_bounds__(PB) void mask_weight_zero_kernel(float* __restrict__ w, const uint8_t* __restrict__ m, size_t n,
                                                              int mode, uint8_t idx) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint8_t mi = m[i];
        if (mi == 0 || (mode == 1 && mi > idx)) w[i] = 0.f;               // prune.py:105 / :116-117
    }
}

// do_batch tail (packnet/main.py:187-193) fused: make_grads_zero -> PacknetSGD.step -> make_pruned_zero
__global__ __launch_bounds__(PB) void packnet_step_kernel(float* __restrict__ theta, float* __restrict__ grad,
                                                          float* __restrict__ buf, const uint8_t* __restrict__ m, size_t n,
                                                          uint8_t cur, float lr, float momentum, float wd, int first) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float g = grad[i], th = theta[i];
        uint8_t mi = m ? m[i] : cur;
        if (m && mi != cur) { g = 0.f; grad[i] = 0.f; }
        float d = g;
        if (wd != 0.f) d = g + (wd * th) * (g != 0.f ? 1.f : 0.f);        // packnetSGD.py:40-43
        float b = first ? d : (buf[i] * momentum + d);
        buf[i] = b;
        th = th - lr * b;
        if (m && mi == 0) th = 0.f;
        theta[i] = th;
    }
}

}  // namespace

extern "C" {

int clhip_packnet_finetune_mask(uint8_t* mask_u8, size_t n, int cur, void* stream) {
    if (!mask_u8 || cur < 1 || cur > 254) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(finetune_mask_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, as_stream(stream), mask_u8, n, (uint8_t)cur);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

size_t clhip_packnet_kth_ws(void) { return 260 * sizeof(uint32_t) + 16; }

int clhip_packnet_kth_abs(const float* w, const uint8_t* mask_u8, size_t n, int cur, size_t k, float* out_cutoff,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!w || !mask_u8 || !out_cutoff || !ws || ws_bytes < clhip_packnet_kth_ws() || k < 1 || k > 0xffffffffull ||
        n == 0) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    uint32_t* u = static_cast<uint32_t*>(ws);
    int* status = reinterpret_cast<int*>(u + 260);
    hipLaunchKernelGGL(kth_init_kernel, dim3(1), dim3(256), 0, s, u, (uint32_t)k, status);
    CLHIP_LAUNCH_CHECK();
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(kth_hist_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, s, w, mask_u8, n, (uint8_t)cur, u);
        CLHIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(kth_select_kernel, dim3(1), dim3(64), 0, s, u, out_cutoff, status);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

int clhip_packnet_prune(float* w, uint8_t* mask_u8, size_t n, int cur, const float* cutoff_dev, void* stream) {
    if (!w || !mask_u8 || !cutoff_dev) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(prune_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, as_stream(stream), w, mask_u8, n, (uint8_t)cur, cutoff_dev);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_mask_grad_zero(float* grad, const uint8_t* mask_u8, size_t n, int cur, void* stream) {
    if (!grad || !mask_u8) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_grad_zero_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, as_stream(stream), grad, mask_u8, n, (uint8_t)cur);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_mask_weight_zero(float* w, const uint8_t* mask_u8, size_t n, int mode, int idx, void* stream) {
    if (!w || !mask_u8 || (mode != 0 && mode != 1)) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(mask_weight_zero_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, as_stream(stream), w, mask_u8, n, mode, (uint8_t)idx);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_packnet_sgd_step(float* theta, float* grad, float* buf, const uint8_t* mask_u8, size_t n, int cur, float lr,
                           float momentum, float wd, int first, void* stream) {
    if (!theta || !grad || !buf) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(packnet_step_kernel, dim3(ew_grid(n, PB)), dim3(PB), 0, as_stream(stream), theta, grad, buf, mask_u8, n,
                       (uint8_t)cur, lr, momentum, wd, first);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
