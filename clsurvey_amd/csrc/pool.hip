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

}  // extern "C"
