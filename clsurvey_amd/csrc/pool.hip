// 2x2 stride-2 max-pool forward (with 2-bit argmax) and backward (scatter), NCHW planes.
// HBM-bound: fwd reads 16 B + writes 5 B per window; bwd reads 5 B + writes 16 B.
#include "common.hpp"

namespace {

constexpr int PB = 256;

// One thread per output element; consecutive threads -> consecutive ow => coalesced
// float2 reads of both input rows.
__global__ __launch_bounds__(PB) void maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, size_t total, int H, int W) {
    const int OH = H >> 1, OW = W >> 1;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
        int ow = (int)(o % OW);
        size_t t = o / OW;
        int oh = (int)(t % OH);
        size_t nc = t / OH;
        const float* p = x + (nc * H + 2 * (size_t)oh) * W + 2 * ow;
        float2 r0 = *reinterpret_cast<const float2*>(p);
        float2 r1 = *reinterpret_cast<const float2*>(p + W);
        // ATen scan order (h, then w); strictly greater (or NaN) replaces => first max wins
        float m = r0.x; int a = 0;
        if (r0.y > m || r0.y != r0.y) { m = r0.y; a = 1; }
        if (r1.x > m || r1.x != r1.x) { m = r1.x; a = 2; }
        if (r1.y > m || r1.y != r1.y) { m = r1.y; a = 3; }
        y[o] = m;
        idx[o] = (uint8_t)a;
    }
}

__global__ __launch_bounds__(PB) void maxpool2_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          float* __restrict__ dx, size_t total, int H, int W) {
    const int OH = H >> 1, OW = W >> 1;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
        int ow = (int)(o % OW);
        size_t t = o / OW;
        int oh = (int)(t % OH);
        size_t nc = t / OH;
        float g = dy[o];
        int a = idx[o];
        float* p = dx + (nc * H + 2 * (size_t)oh) * W + 2 * ow;
        *reinterpret_cast<float2*>(p) = make_float2(a == 0 ? g : 0.f, a == 1 ? g : 0.f);
        *reinterpret_cast<float2*>(p + W) = make_float2(a == 2 ? g : 0.f, a == 3 ? g : 0.f);
    }
}

// General k x k / stride s max-pool without padding (AlexNet's overlapping 3x3 stride 2, torchvision alexnet
// features[2,5,12]): forward keeps the window position of the first maximum (ATen scan order), backward is a GATHER
// over the <= ceil(k/s)^2 windows that cover an input pixel (overlapping windows; no atomics => deterministic).
__global__ __launch_bounds__(PB) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         uint8_t* __restrict__ idx, size_t total, int H, int W, int OH, int OW,
                                                         int k, int s) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += stride) {
        const int ow = (int)(o % OW);
        const size_t t = o / OW;
        const int oh = (int)(t % OH);
        const size_t nc = t / OH;
        const float* p = x + (nc * H + (size_t)oh * s) * W + (size_t)ow * s;
        float m = p[0]; int a = 0;
        for (int r = 0; r < k; ++r)
            for (int c = 0; c < k; ++c) {
                const float v = p[(size_t)r * W + c];
                if (v > m || v != v) { m = v; a = r * k + c; }
            }
        y[o] = m;
        idx[o] = (uint8_t)a;
    }
}

__global__ __launch_bounds__(PB) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                         float* __restrict__ dx, size_t total_in, int H, int W, int OH, int OW,
                                                         int k, int s) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_in; e += stride) {
        const int w = (int)(e % W);
        const size_t t = e / W;
        const int h = (int)(t % H);
        const size_t nc = t / H;
        const int oh_lo = h - k + 1 > 0 ? (h - k + 1 + s - 1) / s : 0, oh_hi = min(OH - 1, h / s);
        const int ow_lo = w - k + 1 > 0 ? (w - k + 1 + s - 1) / s : 0, ow_hi = min(OW - 1, w / s);
        float g = 0.f;
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const size_t o = (nc * OH + oh) * OW + ow;
                if ((int)idx[o] == (h - oh * s) * k + (w - ow * s)) g += dy[o];
            }
        dx[e] = g;
    }
}

// The same gather for a compile-time window / stride, one block per (image, channel) plane, threads laid out as 32 columns
// x 8 rows: no integer divisions (the general kernel does three 64-bit ones per element), window ranges by shifts when
// the stride is 2.  Windows are visited in the same (oh, ow) order => the same sums.  AlexNet's 3x3 / 2 pools
// (models/net.py:96-125), the pooled plane (gradient + arg-max bytes) staged once in LDS: 104 -> 49 us per launch at N = 128.
// Forward for a compile-time window / stride: one block per plane, 32 columns x 8 rows of threads, no integer divisions;
// the same scan order (first maximum wins, NaN propagates) => the same values and codes as maxpool_fwd_kernel.
template <int K, int S>
__global__ __launch_bounds__(PB) void maxpool_fwd_ks_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            uint8_t* __restrict__ idx, int H, int W, int OH, int OW) {
    const size_t nc = blockIdx.x;
    const float* xp = x + nc * H * W;
    float* yp = y + nc * OH * OW;
    uint8_t* ip = idx + nc * OH * OW;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int oh = ty; oh < OH; oh += PB / 32)
        for (int ow = tx; ow < OW; ow += 32) {
            const float* p = xp + (oh * S) * W + ow * S;
            float m = p[0]; int a = 0;
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int c = 0; c < K; ++c) {
                    const float v = p[r * W + c];
                    if (v > m || v != v) { m = v; a = r * K + c; }
                }
            yp[oh * OW + ow] = m;
            ip[oh * OW + ow] = (uint8_t)a;
        }
}

constexpr int POOL_LDS_MAX = 1024;      // pooled plane (gradient + arg-max) staged in LDS: OH * OW <= 1024
template <int K, int S>
__global__ __launch_bounds__(PB) void maxpool_bwd_ks_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                            float* __restrict__ dx, int H, int W, int OH, int OW) {
    __shared__ float gs[POOL_LDS_MAX];
    __shared__ uint8_t cs[POOL_LDS_MAX];
    const size_t nc = blockIdx.x;
    const float* dyp = dy + nc * OH * OW;
    const uint8_t* ip = idx + nc * OH * OW;
    float* dxp = dx + nc * H * W;
    for (int o = threadIdx.x; o < OH * OW; o += PB) { gs[o] = dyp[o]; cs[o] = ip[o]; }      // coalesced, once per plane
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int h = ty; h < H; h += PB / 32) {
        const int oh_lo = h - K + 1 > 0 ? (h - K + 1 + S - 1) / S : 0, oh_hi = min(OH - 1, h / S);
        for (int w = tx; w < W; w += 32) {
            const int ow_lo = w - K + 1 > 0 ? (w - K + 1 + S - 1) / S : 0, ow_hi = min(OW - 1, w / S);
            float g = 0.f;
            for (int oh = oh_lo; oh <= oh_hi; ++oh)
                for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                    const int o = oh * OW + ow;
                    if ((int)cs[o] == (h - oh * S) * K + (w - ow * S)) g += gs[o];
                }
            dxp[h * W + w] = g;
        }
    }
}

}  // namespace

extern "C" {

int clhip_maxpool2_fwd(const float* x, float* y, uint8_t* idx_u8, int NC, int H, int W, void* stream) {
    if (!x || !y || !idx_u8 || NC <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return CLHIP_EINVAL;
    size_t total = (size_t)NC * (H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(ew_grid(total, PB)), dim3(PB), 0, as_stream(stream), x, y, idx_u8, total, H, W);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_maxpool2_bwd(const float* dy, const uint8_t* idx_u8, float* dx, int NC, int H, int W, void* stream) {
    if (!dy || !dx || !idx_u8 || NC <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return CLHIP_EINVAL;
    size_t total = (size_t)NC * (H / 2) * (W / 2);
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(ew_grid(total, PB)), dim3(PB), 0, as_stream(stream), dy, idx_u8, dx, total, H, W);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_maxpool_fwd(const float* x, float* y, uint8_t* idx_u8, int NC, int H, int W, int k, int stride, void* stream) {
    if (!x || !y || !idx_u8 || NC <= 0 || k < 1 || k > 15 || stride < 1 || H < k || W < k) return CLHIP_EINVAL;
    const int OH = (H - k) / stride + 1, OW = (W - k) / stride + 1;
    const size_t total = (size_t)NC * OH * OW;
    if (k == 3 && stride == 2) {
        hipLaunchKernelGGL((maxpool_fwd_ks_kernel<3, 2>), dim3((unsigned)NC), dim3(PB), 0, as_stream(stream), x, y, idx_u8, H, W, OH, OW);
        CLHIP_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total, PB)), dim3(PB), 0, as_stream(stream), x, y, idx_u8, total, H, W, OH, OW, k, stride);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_maxpool_bwd(const float* dy, const uint8_t* idx_u8, float* dx, int NC, int H, int W, int k, int stride, void* stream) {
    if (!dy || !dx || !idx_u8 || NC <= 0 || k < 1 || k > 15 || stride < 1 || H < k || W < k) return CLHIP_EINVAL;
    const int OH = (H - k) / stride + 1, OW = (W - k) / stride + 1;
    const size_t total_in = (size_t)NC * H * W;
    if (k == 3 && stride == 2 && OH * OW <= POOL_LDS_MAX) {
        hipLaunchKernelGGL((maxpool_bwd_ks_kernel<3, 2>), dim3((unsigned)NC), dim3(PB), 0, as_stream(stream), dy, idx_u8, dx, H, W, OH, OW);
        CLHIP_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total_in, PB)), dim3(PB), 0, as_stream(stream), dy, idx_u8, dx, total_in, H, W, OH, OW, k, stride);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
