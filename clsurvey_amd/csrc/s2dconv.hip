// Strided first-layer convolution as a DENSE 3x3 convolution over phase planes (space-to-depth).
//
// AlexNet's first layer, nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2) (models/net.py:96-125 via torchvision.models.alexnet),
// is 18 % of a training step on the gather-GEMM of conv2d.hip (forward 257 us, weight gradient 567 us at N = 128: every MFMA operand
// is an 11 x 11 / stride-4 gather).  With s = stride, the padded input splits into s x s phases,
//     x'[n][ci * s^2 + p * s + q][i][j] = xpad[n][ci][s i + p][s j + q],            xpad = x with `pad` zeros around it,
// the R x R kernel (R <= 3 s) into 3 x 3 taps over those C s^2 planes,
//     w'[k][ci * s^2 + p * s + q][a][b] = w[k][ci][s a + p][s b + q]                (0 where s a + p >= R or s b + q >= R),
// and   y[n][k][oi][oj] = sum_{c', a, b} w'[k][c'][a][b] x'[n][c'][oi + a][oj + b]
// is a VALID 3x3 convolution — the interior of the pad-1 'same' convolution the 3x3 kernels of this library compute, read one
// row / column in: y[oi][oj] = same(x', w')[oi + 1][oj + 1].  So the layer runs on
//     forward          clhip_conv3x3_bs_fwd          (bsconv.hip: bf16 matrix cores, exact 3-piece fp32 splits)
//     weight gradient  clhip_conv3x3_wino_bwd_weight (wino.hip: F(2x2,3x3) on the f32 matrix cores)
// over FRAMES of FH x FW = (OH + 2 rounded up to even) x (OW + 2 rounded up to 16) pixels with the C s^2 = 48 planes padded to 64
// (zero planes, zero rows / columns beyond the data): 1.64 x the layer's multiplies, all of them dense.  dy is embedded in a zero
// frame at offset (1, 1), which makes the 3x3 'same' weight gradient the VALID one; dW' folds back to dW by the inverse index map.
// No backward-data: the layer reads the images.  Everything the reference computes for this layer (Conv2d forward, autograd's
// convolution_backward for weight and bias) — bit-for-bit it is a different summation order, held to torch CPU at 2e-5 / 5e-5 of scale
// like the 3x3 kernels (tests/test_gpu_parity.py::test_s2d_*).
#include "common.hpp"

namespace {

struct S2dGeo {
    int N, C, H, W, K, R, st, pd;
    int OH, OW, FH, FW, Cp, s2;
};

inline bool s2d_geo(int N, int C, int H, int W, int K, int R, int st, int pd, S2dGeo* g) {
    if (N <= 0 || C <= 0 || K <= 0 || st < 2 || R <= 2 * st || R > 3 * st || pd < 0 || pd >= R) return false;
    if (C * st * st > 64 || K % 64 != 0) return false;
    const int OH = (H + 2 * pd - R) / st + 1, OW = (W + 2 * pd - R) / st + 1;
    if (OH < 4 || OW < 4) return false;
    g->N = N; g->C = C; g->H = H; g->W = W; g->K = K; g->R = R; g->st = st; g->pd = pd;
    g->OH = OH; g->OW = OW; g->s2 = st * st; g->Cp = 64;
    g->FH = (OH + 2 + 1) & ~1;
    g->FW = (OW + 2 + 15) & ~15;
    return clhip_internal_bs_ok(g->Cp, K, g->FH, g->FW) && clhip_internal_wino_wgrad_ok(g->Cp, K, g->FH, g->FW);
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

struct S2dWs {
    size_t xf, f2, wp, dwp, inner, inner_bytes, total;
};

inline S2dWs s2d_layout(const S2dGeo& g) {
    S2dWs l;
    size_t off = 0;
    l.xf = off; off += up256((size_t)g.N * g.Cp * g.FH * g.FW * 4);
    l.f2 = off; off += up256((size_t)g.N * g.K * g.FH * g.FW * 4);
    l.wp = off; off += up256((size_t)g.K * g.Cp * 9 * 4);
    l.dwp = off; off += up256((size_t)g.K * g.Cp * 9 * 4);
    const size_t a = clhip_conv3x3_bs_ws(g.Cp, g.K), b = clhip_conv3x3_wino_bwd_weight_ws(g.N, g.Cp, g.K, g.FH, g.FW);
    l.inner = off; l.inner_bytes = up256(a > b ? a : b); off += l.inner_bytes;
    l.total = off;
    return l;
}

// ---- x -> x' frame.  One thread per (n, ci, p, i, 4 j): reads the s (= 4: one float4-sized run) consecutive padded-input columns
// s j .. s j + s - 1 of row s i + p for four j and writes them to the s phase planes q; lanes run along j, so both sides are
// contiguous.  Planes / rows / columns of the frame beyond the data are written as zeros by the same launch (the frame is scratch).
template <int S>
__global__ __launch_bounds__(256) void s2d_input_kernel(const float* __restrict__ x, float* __restrict__ xf, S2dGeo g) {
    const int jq = g.FW / 4;                                   // float4 columns of a frame row
    const size_t rows = (size_t)g.N * g.Cp / S * g.FH;         // (n, plane group of S planes = one (ci, p) or a zero group, i)
    const size_t total = rows * jq;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int j4 = (int)(t % jq);
        const size_t r = t / jq;
        const int i = (int)(r % g.FH);
        const size_t r2 = r / g.FH;
        const int grp = (int)(r2 % (g.Cp / S));                // = ci * S + p
        const int n = (int)(r2 / (g.Cp / S));
        const int ci = grp / S, p = grp - ci * S;
        float v[S][4];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) v[q][u] = 0.f;
        const int hh = S * i + p - g.pd;
        if (ci < g.C && i <= g.OH + 1 && (unsigned)hh < (unsigned)g.H) {
            const float* row = x + (((size_t)n * g.C + ci) * g.H + hh) * g.W;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = 4 * j4 + u;
                if (j <= g.OW + 1) {
#pragma unroll
                    for (int q = 0; q < S; ++q) {
                        const int ww = S * j + q - g.pd;
                        if ((unsigned)ww < (unsigned)g.W) v[q][u] = row[ww];
                    }
                }
            }
        }
        float* dst = xf + (((size_t)n * g.Cp + (size_t)grp * S) * g.FH + i) * g.FW + 4 * j4;
#pragma unroll
        for (int q = 0; q < S; ++q)
            *reinterpret_cast<float4*>(dst + (size_t)q * g.FH * g.FW) = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
    }
}

// ---- y[n][k][oi][oj] = frame[n][k][oi + 1][oj + 1]
__global__ __launch_bounds__(256) void s2d_crop_kernel(const float* __restrict__ f, float* __restrict__ y, S2dGeo g) {
    const size_t total = (size_t)g.N * g.K * g.OH * g.OW;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int oj = (int)(t % g.OW);
        const size_t r = t / g.OW;
        const int oi = (int)(r % g.OH);
        const size_t plane = r / g.OH;
        y[t] = f[(plane * g.FH + oi + 1) * g.FW + oj + 1];
    }
}

// ---- frame[n][k][oi + 1][oj + 1] = dy[n][k][oi][oj], zero elsewhere (whole frame written: float4 per thread)
__global__ __launch_bounds__(256) void s2d_embed_kernel(const float* __restrict__ dy, float* __restrict__ f, S2dGeo g) {
    const int jq = g.FW / 4;
    const size_t total = (size_t)g.N * g.K * g.FH * jq;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
        const int j4 = (int)(t % jq);
        const size_t r = t / jq;
        const int i = (int)(r % g.FH);
        const size_t plane = r / g.FH;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (i >= 1 && i <= g.OH) {
            const float* row = dy + (plane * g.OH + (i - 1)) * g.OW;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int oj = 4 * j4 + u - 1;
                if ((unsigned)oj < (unsigned)g.OW) v[u] = row[oj];
            }
        }
        *reinterpret_cast<float4*>(f + (plane * g.FH + i) * g.FW + 4 * j4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- w[k][ci][u][v] <-> w'[k][c'][a][b], c' = ci s^2 + p s + q, u = s a + p, v = s b + q
__global__ __launch_bounds__(256) void s2d_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, S2dGeo g) {
    const int total = g.K * g.Cp * 9;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int b = t % 3, a = (t / 3) % 3, cp = (t / 9) % g.Cp, k = t / (9 * g.Cp);
        const int ci = cp / g.s2, rem = cp - ci * g.s2, p = rem / g.st, q = rem - p * g.st;
        const int u = g.st * a + p, v = g.st * b + q;
        wp[t] = (ci < g.C && u < g.R && v < g.R) ? w[(((size_t)k * g.C + ci) * g.R + u) * g.R + v] : 0.f;
    }
}

__global__ __launch_bounds__(256) void s2d_wgrad_out_kernel(const float* __restrict__ dwp, float* __restrict__ dw, S2dGeo g) {
    const int total = g.K * g.C * g.R * g.R;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int v = t % g.R, u = (t / g.R) % g.R, ci = (t / (g.R * g.R)) % g.C, k = t / (g.R * g.R * g.C);
        const int a = u / g.st, p = u - a * g.st, b = v / g.st, q = v - b * g.st;
        dw[t] = dwp[(((size_t)k * g.Cp + ci * g.s2 + p * g.st + q) * 3 + a) * 3 + b];
    }
}

int launch_input(const float* x, float* xf, const S2dGeo& g, hipStream_t s) {
    const size_t total = (size_t)g.N * g.Cp / g.st * g.FH * (g.FW / 4);
    const int grid = ew_grid(total, 256);
    switch (g.st) {
        case 2: hipLaunchKernelGGL(s2d_input_kernel<2>, dim3(grid), dim3(256), 0, s, x, xf, g); break;
        case 4: hipLaunchKernelGGL(s2d_input_kernel<4>, dim3(grid), dim3(256), 0, s, x, xf, g); break;
        default: return CLHIP_ENOTSUP;
    }
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

size_t clhip_conv2d_s2d_ws(int N, int C, int H, int W, int K, int R, int stride, int pad) {
    S2dGeo g;
    if (!s2d_geo(N, C, H, W, K, R, stride, pad, &g) || (stride != 2 && stride != 4)) return 0;
    return s2d_layout(g).total;
}

int clhip_conv2d_s2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R, int stride,
                         int pad, int relu, void* ws, size_t ws_bytes, void* stream) {
    S2dGeo g;
    if (!s2d_geo(N, C, H, W, K, R, stride, pad, &g) || (stride != 2 && stride != 4)) return CLHIP_ENOTSUP;
    const S2dWs l = s2d_layout(g);
    if (!x || !w || !y || !ws || ws_bytes < l.total) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    char* base = static_cast<char*>(ws);
    float* xf = reinterpret_cast<float*>(base + l.xf);
    float* f2 = reinterpret_cast<float*>(base + l.f2);
    float* wp = reinterpret_cast<float*>(base + l.wp);
    int rc = launch_input(x, xf, g, s);
    if (rc) return rc;
    hipLaunchKernelGGL(s2d_weight_kernel, dim3((g.K * g.Cp * 9 + 255) / 256), dim3(256), 0, s, w, wp, g);
    CLHIP_LAUNCH_CHECK();
    rc = clhip_conv3x3_bs_fwd(xf, wp, b, f2, nullptr, N, g.Cp, K, g.FH, g.FW, relu, base + l.inner, l.inner_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(s2d_crop_kernel, dim3(ew_grid((size_t)N * K * g.OH * g.OW, 256)), dim3(256), 0, s, f2, y, g);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_conv2d_s2d_bwd_weight(const float* x_or_null, const float* dy, float* dw, float* db, int N, int C, int H, int W, int K, int R,
                                int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    S2dGeo g;
    if (!s2d_geo(N, C, H, W, K, R, stride, pad, &g) || (stride != 2 && stride != 4)) return CLHIP_ENOTSUP;
    const S2dWs l = s2d_layout(g);
    if (!dy || !dw || !db || !ws || ws_bytes < l.total) return CLHIP_EINVAL;
    hipStream_t s = as_stream(stream);
    char* base = static_cast<char*>(ws);
    float* xf = reinterpret_cast<float*>(base + l.xf);
    float* f2 = reinterpret_cast<float*>(base + l.f2);
    float* dwp = reinterpret_cast<float*>(base + l.dwp);
    int rc = 0;
    if (x_or_null) {                       // NULL: the x' frame of clhip_conv2d_s2d_fwd on the same ws and batch is still there
        rc = launch_input(x_or_null, xf, g, s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(s2d_embed_kernel, dim3(ew_grid((size_t)N * K * g.FH * (g.FW / 4), 256)), dim3(256), 0, s, dy, f2, g);
    CLHIP_LAUNCH_CHECK();
    rc = clhip_conv3x3_wino_bwd_weight(xf, f2, nullptr, dwp, db, N, g.Cp, K, g.FH, g.FW, base + l.inner, l.inner_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(s2d_wgrad_out_kernel, dim3((K * C * R * R + 255) / 256), dim3(256), 0, s, dwp, dw, g);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
