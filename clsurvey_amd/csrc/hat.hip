// HAT (hard attention to the task) parameter-side kernels — restates methods/HAT/networks/vgg_hat.py
// (gates, back-masks), approaches/hat.py (criterion regulariser) and HAT_utils.py:HAT_SGD.
//
// Design: the per-channel gate a_l = sigmoid(s * E_l[t]) multiplies layer l's OUTPUT in the reference
// (vgg_hat.py:104-116).  Multiplying an input channel by a constant is the same linear map as scaling
// that input-channel slice of the NEXT layer's weights, so the activation tensors stay un-gated and
// the conv / FC kernels are reused unchanged on W'[k][c][r] = W[k][c][r] * a_prev[c]:
//     forward / backward-data run on W';  dL/dW = (dL/dW') * a_prev[c];
//     dL/da_prev[c] = sum_{k,r} (dL/dW')[k][c][r] * W[k][c][r].
// Everything here is HBM-bound over the parameters (<= 9 M floats) or tiny per-channel vectors.
#include "common.hpp"

namespace {

constexpr int HB = 256;

__global__ __launch_bounds__(HB) void hat_gate_kernel(const float* __restrict__ e, int n, float s, float* __restrict__ a) {
    int i = blockIdx.x * HB + threadIdx.x;
    if (i < n) a[i] = 1.f / (1.f + expf(-s * e[i]));                      // vgg_hat.py:121-127
}

// out[k][c][r] = w[k][c][r] * (gate ? gate[c] : 1)
__global__ __launch_bounds__(HB) void hat_scale_weight_kernel(const float* __restrict__ w, const float* __restrict__ gate,
                                                              float* __restrict__ out, size_t total, size_t C, size_t R) {
    size_t stride = (size_t)gridDim.x * HB;
    for (size_t i = (size_t)blockIdx.x * HB + threadIdx.x; i < total; i += stride) {
        size_t c = (i / R) % C;
        out[i] = gate ? w[i] * gate[c] : w[i];
    }
}

// one block per input channel c: dgate[c] = sum_{k,r} gp*w (fixed order), dw = gp * gate[c]
__global__ __launch_bounds__(HB) void hat_weight_grad_kernel(const float* __restrict__ gp, const float* __restrict__ w,
                                                             const float* __restrict__ gate, float* __restrict__ dw,
                                                             float* __restrict__ dgate, int K, int C, int R) {
    __shared__ double part[HB];
    const int c = blockIdx.x;
    const float a = gate[c];
    double s = 0.0;
    const int n = K * R;
    for (int i = threadIdx.x; i < n; i += HB) {
        int k = i / R, r = i - k * R;
        size_t o = ((size_t)k * C + c) * R + r;
        float g = gp[o];
        s += (double)g * (double)w[o];
        dw[o] = g * a;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dgate[c] = (float)part[0];
}

// demb = (dgate + lamb_over_count * (1 - mask_pre)) * s * a * (1 - a)        hat.py:285-299 + sigmoid'
__global__ __launch_bounds__(HB) void hat_emb_grad_kernel(const float* __restrict__ dgate, const float* __restrict__ a,
                                                          const float* __restrict__ mp, int n, float s, float loc,
                                                          float* __restrict__ demb) {
    int i = blockIdx.x * HB + threadIdx.x;
    if (i < n) {
        float aux = mp ? 1.f - mp[i] : 1.f;
        float ai = a[i];
        demb[i] = (dgate[i] + loc * aux) * (s * ai * (1.f - ai));
    }
}

// sums[0] += sum a*(1-mp) ; sums[1] += sum (1-mp)   (single block => fixed order; launches are stream-ordered)
__global__ __launch_bounds__(HB) void hat_reg_sums_kernel(const float* __restrict__ a, const float* __restrict__ mp, int n,
                                                          double* __restrict__ sums) {
    __shared__ double p0[HB], p1[HB];
    double s0 = 0, s1 = 0;
    for (int i = threadIdx.x; i < n; i += HB) {
        float aux = mp ? 1.f - mp[i] : 1.f;
        s0 += (double)(a[i] * aux);
        s1 += (double)aux;
    }
    p0[threadIdx.x] = s0; p1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { p0[threadIdx.x] += p0[threadIdx.x + o]; p1[threadIdx.x] += p1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[0] += p0[0]; sums[1] += p1[0]; }
}

// out[k][c][r] = 1 - min(post[k], pre ? pre[c] : +inf)                         vgg_hat.py:258-295
__global__ __launch_bounds__(HB) void hat_backmask_kernel(const float* __restrict__ post, const float* __restrict__ pre,
                                                          float* __restrict__ out, size_t total, size_t C, size_t R) {
    size_t stride = (size_t)gridDim.x * HB;
    for (size_t i = (size_t)blockIdx.x * HB + threadIdx.x; i < total; i += stride) {
        size_t k = i / (C * R), c = (i / R) % C;
        float v = post[k];
        if (pre) v = fminf(v, pre[c]);
        out[i] = 1.f - v;
    }
}

// pass 1 of HAT_SGD.step (HAT_utils.py:211-229): weight decay (not on embeddings), back-mask, embedding
// gradient compensation; writes the modified grad and one partial sum of squares per block.
__global__ __launch_bounds__(HB) void hat_sgd_prep_kernel(const float* __restrict__ theta, float* __restrict__ grad,
                                                          const float* __restrict__ mask_back, size_t n, float wd,
                                                          int is_emb, int compensate, float s, float smax, float thres_cosh,
                                                          double* __restrict__ partial) {
    __shared__ double part[HB];
    double ss = 0.0;
    size_t stride = (size_t)gridDim.x * HB;
    for (size_t i = (size_t)blockIdx.x * HB + threadIdx.x; i < n; i += stride) {
        float g = grad[i], th = theta[i];
        if (wd != 0.f && !is_emb) g += wd * th;
        if (mask_back) g *= mask_back[i];
        if (is_emb && compensate) {
            float x = fminf(fmaxf(s * th, -thres_cosh), thres_cosh);
            float num = coshf(x) + 1.f, den = coshf(th) + 1.f;
            g *= smax / s * num / den;
        }
        grad[i] = g;
        ss += (double)g * (double)g;
    }
    part[threadIdx.x] = ss;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = part[0];
}

// pass 2: clip_grad_norm_(p, clipgrad) (coef = clip/(norm+1e-6) if < 1) then momentum SGD
__global__ __launch_bounds__(HB) void hat_sgd_apply_kernel(float* __restrict__ theta, float* __restrict__ grad,
                                                           float* __restrict__ buf, size_t n, float lr, float momentum,
                                                           int first, int do_clip, float clipgrad,
                                                           const double* __restrict__ partial, int nparts) {
    float coef = 1.f;
    if (do_clip) {
        double t = 0.0;
        for (int i = 0; i < nparts; ++i) t += partial[i];
        float norm = (float)sqrt(t);
        float c = clipgrad / (norm + 1e-6f);
        if (c < 1.f) coef = c;
    }
    size_t stride = (size_t)gridDim.x * HB;
    for (size_t i = (size_t)blockIdx.x * HB + threadIdx.x; i < n; i += stride) {
        float g = grad[i] * coef;
        grad[i] = g;
        float b = (momentum != 0.f) ? (first ? g : buf[i] * momentum + g) : g;
        if (momentum != 0.f) buf[i] = b;
        theta[i] = theta[i] - lr * b;
    }
}

__global__ __launch_bounds__(HB) void clamp_kernel(float* __restrict__ x, size_t n, float lo, float hi) {
    size_t stride = (size_t)gridDim.x * HB;
    for (size_t i = (size_t)blockIdx.x * HB + threadIdx.x; i < n; i += stride) x[i] = fminf(fmaxf(x[i], lo), hi);
}

}  // namespace

extern "C" {

int clhip_hat_gate(const float* emb_row, int n, float s, float* gate, void* stream) {
    if (!emb_row || !gate || n <= 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(hat_gate_kernel, dim3((n + HB - 1) / HB), dim3(HB), 0, as_stream(stream), emb_row, n, s, gate);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_scale_weight(const float* w, const float* gate_in, float* out, size_t K, size_t C, size_t R, void* stream) {
    if (!w || !out || K == 0 || C == 0 || R == 0) return CLHIP_EINVAL;
    size_t total = K * C * R;
    hipLaunchKernelGGL(hat_scale_weight_kernel, dim3(ew_grid(total, HB)), dim3(HB), 0, as_stream(stream), w, gate_in, out, total, C, R);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_weight_grad(const float* g_wprime, const float* w, const float* gate_in, float* dw, float* dgate_in,
                          int K, int C, int R, void* stream) {
    if (!g_wprime || !w || !gate_in || !dw || !dgate_in || K <= 0 || C <= 0 || R <= 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(hat_weight_grad_kernel, dim3(C), dim3(HB), 0, as_stream(stream), g_wprime, w, gate_in, dw, dgate_in, K, C, R);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_emb_grad(const float* dgate, const float* gate, const float* mask_pre, int n, float s, float lamb_over_count,
                       float* demb, void* stream) {
    if (!dgate || !gate || !demb || n <= 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(hat_emb_grad_kernel, dim3((n + HB - 1) / HB), dim3(HB), 0, as_stream(stream), dgate, gate, mask_pre, n, s,
                       lamb_over_count, demb);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_reg_sums(const float* gate, const float* mask_pre, int n, double* sums2, void* stream) {
    if (!gate || !sums2 || n <= 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(hat_reg_sums_kernel, dim3(1), dim3(HB), 0, as_stream(stream), gate, mask_pre, n, sums2);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_backmask(const float* a_post, const float* a_pre, float* out, size_t K, size_t C, size_t R, void* stream) {
    if (!a_post || !out || K == 0 || C == 0 || R == 0) return CLHIP_EINVAL;
    size_t total = K * C * R;
    hipLaunchKernelGGL(hat_backmask_kernel, dim3(ew_grid(total, HB)), dim3(HB), 0, as_stream(stream), a_post, a_pre, out, total, C, R);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

size_t clhip_hat_sgd_ws(void) { return 1024 * sizeof(double); }

int clhip_hat_sgd_step(float* theta, float* grad, float* buf, const float* mask_back, size_t n, float lr, float momentum,
                       float wd, int is_emb, int finetune, float s, float smax, float thres_cosh, float clipgrad, int first,
                       void* ws, size_t ws_bytes, void* stream) {
    if (!theta || !grad || !buf || !ws || ws_bytes < clhip_hat_sgd_ws() || n == 0) return CLHIP_EINVAL;
    hipStream_t st = as_stream(stream);
    int blocks = ew_grid(n, HB);
    if (blocks > 1024) blocks = 1024;
    double* partial = static_cast<double*>(ws);
    hipLaunchKernelGGL(hat_sgd_prep_kernel, dim3(blocks), dim3(HB), 0, st, theta, grad, mask_back, n, wd, is_emb, !finetune, s, smax,
                       thres_cosh, partial);
    CLHIP_LAUNCH_CHECK();
    hipLaunchKernelGGL(hat_sgd_apply_kernel, dim3(blocks), dim3(HB), 0, st, theta, grad, buf, n, lr, momentum, first, !finetune,
                       clipgrad, partial, blocks);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_clamp(float* x, size_t n, float lo, float hi, void* stream) {
    if (!x) return CLHIP_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(clamp_kernel, dim3(ew_grid(n, HB)), dim3(HB), 0, as_stream(stream), x, n, lo, hi);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
