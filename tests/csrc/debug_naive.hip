// Test-only device kernels: direct convolutions (one thread per output element, plain loops)
// and an MFMA fragment-layout probe.  They exist so a parity failure on the GPU box can be
// triaged between "harness / layout contract" and "MFMA kernel" in ONE gpurun call.
// Built into tests/libclhip_dbg.so (clsurvey_amd/build.py: build_test_lib); NOT part of libclhip.so.
#include "../../clsurvey_amd/csrc/common.hpp"

#pragma GCC visibility push(default)        // (built with -fvisibility=hidden like the product library: these four are exported)
extern "C" {
int clhip_dbg_conv3x3_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int K, int H, int W, int relu,
                          void* stream);
int clhip_dbg_conv3x3_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int K, int H,
                               int W, void* stream);
int clhip_dbg_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W,
                                 void* stream);
// MFMA fragment-layout probe: out[2][32][32]: D of the 32x32x2 f32 MFMA for rank-1 A, B patterns per k slice, stored through
// the documented fragment map
int clhip_dbg_mfma_probe(float* out_2048, void* stream);
}
#pragma GCC visibility pop

namespace {

__global__ void dbg_conv_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int K, int H,
                             int W, int relu) {
    size_t total = (size_t)N * K * H * W;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        int ww = o % W; size_t t = o / W; int hh = t % H; t /= H; int k = t % K; int n = t / K;
        float acc = b ? b[k] : 0.f;
        for (int c = 0; c < C; ++c)
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) {
                    int h = hh + r - 1, wi = ww + s - 1;
                    if (h >= 0 && h < H && wi >= 0 && wi < W)
                        acc += x[(((size_t)n * C + c) * H + h) * W + wi] * w[(((size_t)k * C + c) * 3 + r) * 3 + s];
                }
        y[o] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

__global__ void dbg_conv_bwd_data(const float* dy, const float* w, const float* m, float* dx, int N, int C, int K,
                                  int H, int W) {
    size_t total = (size_t)N * C * H * W;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        int ww = o % W; size_t t = o / W; int hh = t % H; t /= H; int c = t % C; int n = t / C;
        float acc = 0.f;
        for (int k = 0; k < K; ++k)
            for (int r = 0; r < 3; ++r)
                for (int s = 0; s < 3; ++s) {
                    int h = hh - r + 1, wi = ww - s + 1;
                    if (h >= 0 && h < H && wi >= 0 && wi < W)
                        acc += dy[(((size_t)n * K + k) * H + h) * W + wi] * w[(((size_t)k * C + c) * 3 + r) * 3 + s];
                }
        dx[o] = (m && !(m[o] > 0.f)) ? 0.f : acc;
    }
}

__global__ void dbg_conv_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K,
                                    int H, int W) {
    size_t total = (size_t)K * C * 9;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total + K; o += (size_t)gridDim.x * blockDim.x) {
        if (o >= total) {
            if (!db) continue;
            int k = o - total;
            float acc = 0.f;
            for (int n = 0; n < N; ++n)
                for (int i = 0; i < H * W; ++i) acc += dy[((size_t)n * K + k) * H * W + i];
            db[k] = acc;
            continue;
        }
        int s = o % 3; size_t t = o / 3; int r = t % 3; t /= 3; int c = t % C; int k = t / C;
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
            for (int hh = 0; hh < H; ++hh)
                for (int ww = 0; ww < W; ++ww) {
                    int h = hh + r - 1, wi = ww + s - 1;
                    if (h >= 0 && h < H && wi >= 0 && wi < W)
                        acc += dy[(((size_t)n * K + k) * H + hh) * W + ww] * x[(((size_t)n * C + c) * H + h) * W + wi];
                }
        dw[o] = acc;
    }
}

// out[0..1023]: D of a 32x32x2 MFMA with A[i][k] = (k==0 ? i+1 : 0), B[k][j] = (k==0 ? 100*(j+1) : 0)
// stored at out[row*32+col] using the documented map => expected out[i*32+j] = (i+1)*100*(j+1).
// out[1024..2047]: same with the k==1 slice (A[i][1] = i+1, B[1][j] = 100*(j+1)).
__global__ void dbg_mfma_probe(float* out) {
    int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
    for (int which = 0; which < 2; ++which) {
        floatx16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        float a = (kk == which) ? (float)(li + 1) : 0.f;
        float b = (kk == which) ? 100.f * (li + 1) : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 16; ++r) out[which * 1024 + mfma32_row(r, lane) * 32 + li] = acc[r];
    }
}

}  // namespace

extern "C" {

int clhip_dbg_conv3x3_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int K, int H, int W,
                          int relu, void* stream) {
    hipLaunchKernelGGL(dbg_conv_fwd, dim3(1024), dim3(256), 0, as_stream(stream), x, w, b, y, N, C, K, H, W, relu);
    CLHIP_LAUNCH_CHECK();
    return 0;
}
int clhip_dbg_conv3x3_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int K,
                               int H, int W, void* stream) {
    hipLaunchKernelGGL(dbg_conv_bwd_data, dim3(1024), dim3(256), 0, as_stream(stream), dy, w, relu_src, dx, N, C, K, H, W);
    CLHIP_LAUNCH_CHECK();
    return 0;
}
int clhip_dbg_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H,
                                 int W, void* stream) {
    hipLaunchKernelGGL(dbg_conv_bwd_weight, dim3(256), dim3(256), 0, as_stream(stream), x, dy, dw, db, N, C, K, H, W);
    CLHIP_LAUNCH_CHECK();
    return 0;
}
int clhip_dbg_mfma_probe(float* out_2048, void* stream) {
    hipLaunchKernelGGL(dbg_mfma_probe, dim3(1), dim3(64), 0, as_stream(stream), out_2048);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
