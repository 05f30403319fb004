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


// ------------------------------------------------------------------------------------------------ whole-net launches
// One training batch of HAT used to cost ~100 parameter-sized launches next to the ~35 of the net itself (rocprofv3,
// profiles/r03a_hat_kernel_stats.csv: 26 x hat_sgd_apply at 21 us each — every thread summed up to 1024 f64 partials alone —
// 26 x prep, 8 x each of gate / scale / reg-sums / weight-grad / emb-grad / clamp / fill): 0.9 ms of a 6.4 ms step on
// wide_VGG9.  The *_multi entry points take job tables (passed by value as kernel arguments, <= HAT_JOBS per launch) and
// do the same arithmetic in 6 launches per batch.
constexpr int HAT_JOBS = 40;
constexpr int SGD_MAX_BLOCKS = 1024;        // blocks per parameter in the HAT_SGD launches (partials per parameter)

struct SgdJob { float* theta; float* grad; float* buf; const float* mask_back; size_t n; int is_emb; int first_block; int n_blocks; int pad; };
struct SgdJobs { int n; int pad; SgdJob j[HAT_JOBS]; };

__device__ __forceinline__ int job_of_block(const SgdJobs& J, int b) {
    int k = 0;
    for (int i = 1; i < J.n; ++i) k = (b >= J.j[i].first_block) ? i : k;
    return k;
}

__device__ __forceinline__ float hat_prep_one(float g, float th, float mb, bool has_mb, float wd, int is_emb, int compensate, float s,
                                              float smax, float thres_cosh) {
    if (wd != 0.f && !is_emb) g += wd * th;
    if (has_mb) g *= mb;
    if (is_emb && compensate) {
        float x = fminf(fmaxf(s * th, -thres_cosh), thres_cosh);
        float num = coshf(x) + 1.f, den = coshf(th) + 1.f;
        g *= smax / s * num / den;
    }
    return g;
}

__global__ __launch_bounds__(HB) void hat_sgd_prep_multi_kernel(SgdJobs J, float wd, int compensate, float s, float smax,
                                                                float thres_cosh, double* __restrict__ partial) {
    __shared__ double part[HB];
    const SgdJob& jb = J.j[job_of_block(J, blockIdx.x)];
    const float* __restrict__ theta = jb.theta;
    float* __restrict__ grad = jb.grad;
    const float* __restrict__ mask_back = jb.mask_back;
    const size_t n = jb.n, stride = (size_t)jb.n_blocks * HB;
    const int is_emb = jb.is_emb;
    const bool has_mb = mask_back != nullptr;
    const size_t first = (size_t)(blockIdx.x - jb.first_block) * HB + threadIdx.x;
    double ss = 0.0;
    // 16-byte body (a 51 M-element Linear weight of wide_VGG9 at 224x224 moved at 1.4 TB/s through the scalar loop)
    const bool vec = (((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)mask_back) & 15u) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = first; i < n4; i += stride) {
        float4 g = reinterpret_cast<float4*>(grad)[i];
        const float4 th = reinterpret_cast<const float4*>(theta)[i];
        const float4 mb = has_mb ? reinterpret_cast<const float4*>(mask_back)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        g.x = hat_prep_one(g.x, th.x, mb.x, has_mb, wd, is_emb, compensate, s, smax, thres_cosh);
        g.y = hat_prep_one(g.y, th.y, mb.y, has_mb, wd, is_emb, compensate, s, smax, thres_cosh);
        g.z = hat_prep_one(g.z, th.z, mb.z, has_mb, wd, is_emb, compensate, s, smax, thres_cosh);
        g.w = hat_prep_one(g.w, th.w, mb.w, has_mb, wd, is_emb, compensate, s, smax, thres_cosh);
        reinterpret_cast<float4*>(grad)[i] = g;
        ss += ((double)g.x * g.x + (double)g.y * g.y) + ((double)g.z * g.z + (double)g.w * g.w);
    }
    for (size_t i = 4 * n4 + first; i < n; i += stride) {
        const float g = hat_prep_one(grad[i], theta[i], has_mb ? mask_back[i] : 1.f, has_mb, wd, is_emb, compensate, s, smax, thres_cosh);
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

// clip_grad_norm_ per parameter: the block adds its parameter's <= HB partials as a fixed tree (one per thread), then
// applies momentum SGD and, on embeddings, the clamp of hat.py:238-240 (thres_emb > 0).
__global__ __launch_bounds__(HB) void hat_sgd_apply_multi_kernel(SgdJobs J, float lr, float momentum, int first, int do_clip,
                                                                 float clipgrad, float thres_emb,
                                                                 const double* __restrict__ partial) {
    __shared__ double part[HB];
    const SgdJob& jb = J.j[job_of_block(J, blockIdx.x)];
    float coef = 1.f;
    if (do_clip) {
        double t = 0.0;                                   // <= SGD_MAX_BLOCKS / HB partials per thread, then a fixed tree
        for (int i = threadIdx.x; i < jb.n_blocks; i += HB) t += partial[jb.first_block + i];
        part[threadIdx.x] = t;
        __syncthreads();
        for (int o = HB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
            __syncthreads();
        }
        const float norm = (float)sqrt(part[0]);
        const float c = clipgrad / (norm + 1e-6f);
        if (c < 1.f) coef = c;
    }
    float* __restrict__ theta = jb.theta;
    float* __restrict__ grad = jb.grad;
    float* __restrict__ buf = jb.buf;
    const size_t n = jb.n, stride = (size_t)jb.n_blocks * HB;
    const bool clamp = jb.is_emb && thres_emb > 0.f;
    const bool mom = momentum != 0.f;
    auto one = [&](float& g, float& b, float& th) {
        g *= coef;
        b = mom ? (first ? g : b * momentum + g) : g;
        th = th - lr * b;
        if (clamp) th = fminf(fmaxf(th, -thres_emb), thres_emb);
    };
    const size_t t0 = (size_t)(blockIdx.x - jb.first_block) * HB + threadIdx.x;
    const bool vec = (((uintptr_t)theta | (uintptr_t)grad | (uintptr_t)buf) & 15u) == 0;
    const size_t n4 = vec ? n / 4 : 0;
    for (size_t i = t0; i < n4; i += stride) {
        float4 g = reinterpret_cast<float4*>(grad)[i], th = reinterpret_cast<float4*>(theta)[i];
        float4 b = (mom && !first) ? reinterpret_cast<float4*>(buf)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        one(g.x, b.x, th.x); one(g.y, b.y, th.y); one(g.z, b.z, th.z); one(g.w, b.w, th.w);
        reinterpret_cast<float4*>(grad)[i] = g;
        if (mom) reinterpret_cast<float4*>(buf)[i] = b;
        reinterpret_cast<float4*>(theta)[i] = th;
    }
    for (size_t i = 4 * n4 + t0; i < n; i += stride) {
        float g = grad[i], th = theta[i], b = (mom && !first) ? buf[i] : 0.f;
        one(g, b, th);
        grad[i] = g;
        if (mom) buf[i] = b;
        theta[i] = th;
    }
}

struct GateJobs { int n; int pad; clhip_hat_gate_job j[HAT_JOBS]; };
// ONE block: every layer's gate a = sigmoid(s * E[t]) and the two sums of the regulariser (hat.py:285-299), layers and
// elements in a fixed order.
__global__ __launch_bounds__(HB) void hat_gates_multi_kernel(GateJobs J, float s, double* __restrict__ sums) {
    __shared__ double p0[HB], p1[HB];
    double s0 = 0, s1 = 0;
    for (int l = 0; l < J.n; ++l) {
        const clhip_hat_gate_job& jb = J.j[l];
        for (int i = threadIdx.x; i < jb.n; i += HB) {
            const float a = 1.f / (1.f + expf(-s * jb.emb_row[i]));
            jb.gate[i] = a;
            const float aux = jb.mask_pre ? 1.f - jb.mask_pre[i] : 1.f;
            s0 += (double)(a * aux);
            s1 += (double)aux;
        }
    }
    p0[threadIdx.x] = s0; p1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) { p0[threadIdx.x] += p0[threadIdx.x + o]; p1[threadIdx.x] += p1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && sums) { sums[0] = p0[0]; sums[1] = p1[0]; sums[2] = p0[0] / p1[0]; }
}

struct ScaleJob { const float* w; const float* gate; float* out; size_t total, C, R; int first_block; int n_blocks; };
struct ScaleJobs { int n; int pad; ScaleJob j[HAT_JOBS]; };
__global__ __launch_bounds__(HB) void hat_scale_multi_kernel(ScaleJobs J) {
    int k = 0;
    for (int i = 1; i < J.n; ++i) k = ((int)blockIdx.x >= J.j[i].first_block) ? i : k;
    const ScaleJob& jb = J.j[k];
    const float* __restrict__ w = jb.w;
    const float* __restrict__ gate = jb.gate;
    float* __restrict__ out = jb.out;
    const size_t stride = (size_t)jb.n_blocks * HB, first = (size_t)(blockIdx.x - jb.first_block) * HB + threadIdx.x;
    const unsigned R = (unsigned)jb.R, C = (unsigned)jb.C;
    const bool vec = ((((uintptr_t)w | (uintptr_t)out) & 15u) == 0) && (!gate || R % 4 == 0) && jb.total < 0xffffffffull;
    const size_t n4 = vec ? jb.total / 4 : 0;
    for (size_t i = first; i < n4; i += stride) {             // (R % 4 == 0: the four elements share their input channel)
        float4 v = reinterpret_cast<const float4*>(w)[i];
        if (gate) { const float a = gate[((unsigned)(4 * i) / R) % C]; v.x *= a; v.y *= a; v.z *= a; v.w *= a; }
        reinterpret_cast<float4*>(out)[i] = v;
    }
    for (size_t i = 4 * n4 + first; i < jb.total; i += stride)
        out[i] = gate ? w[i] * gate[jb.total < 0xffffffffull ? ((unsigned)i / R) % C : (i / jb.R) % jb.C] : w[i];
}

struct WgJob { float* g; const float* w; const float* gate; float* dgate; int K, C, R, first_block; };
struct WgJobs { int n; int pad; WgJob j[HAT_JOBS]; };
// one block per (layer, input channel c), as hat_weight_grad_kernel; in place on the gradient
__global__ __launch_bounds__(HB) void hat_weight_grad_multi_kernel(WgJobs J) {
    __shared__ double part[HB];
    int k = 0;
    for (int i = 1; i < J.n; ++i) k = ((int)blockIdx.x >= J.j[i].first_block) ? i : k;
    const WgJob& jb = J.j[k];
    const int c = blockIdx.x - jb.first_block, C = jb.C, R = jb.R;
    float* __restrict__ g = jb.g;
    const float* __restrict__ w = jb.w;
    const float a = jb.gate[c];
    double s = 0.0;
    const int n = jb.K * R;
    for (int i = threadIdx.x; i < n; i += HB) {
        const int kk = i / R, r = i - kk * R;
        const size_t o = ((size_t)kk * C + c) * R + r;
        const float gv = g[o];
        s += (double)gv * (double)w[o];
        g[o] = gv * a;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = HB / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) jb.dgate[c] = (float)part[0];
}

struct EmbJobs { int n; int pad; clhip_hat_emb_job j[HAT_JOBS]; };
// grid.x = layers: the dense [rows][n] embedding gradient of a layer, zero except row t (what autograd leaves in
// nn.Embedding.weight.grad); lamb / count with count read on the device when the caller has not got it
__global__ __launch_bounds__(HB) void hat_emb_grads_multi_kernel(EmbJobs J, float s, float lamb, float count,
                                                                 const double* __restrict__ sums) {
    const clhip_hat_emb_job& jb = J.j[blockIdx.x];
    const float loc = lamb / (count > 0.f ? count : (float)sums[1]);
    for (int i = threadIdx.x; i < jb.rows * jb.n; i += HB) {
        const int row = i / jb.n, c = i - row * jb.n;
        float v = 0.f;
        if (row == jb.t) {
            const float aux = jb.mask_pre ? 1.f - jb.mask_pre[c] : 1.f;
            const float ai = jb.gate[c];
            v = (jb.dgate[c] + loc * aux) * (s * ai * (1.f - ai));
        }
        jb.demb[i] = v;
    }
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

size_t clhip_hat_sgd_multi_ws(int n_params) { return (size_t)(n_params > 0 ? n_params : 1) * SGD_MAX_BLOCKS * sizeof(double); }

int clhip_hat_sgd_step_multi(const clhip_hat_param* params, int n_params, float lr, float momentum, float wd, int finetune,
                             float s, float smax, float thres_cosh, float clipgrad, float thres_emb, int first, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!params || n_params <= 0 || !ws || ws_bytes < clhip_hat_sgd_multi_ws(n_params)) return CLHIP_EINVAL;
    for (int i = 0; i < n_params; ++i)
        if (!params[i].theta || !params[i].grad || !params[i].buf || params[i].n == 0) return CLHIP_EINVAL;
    hipStream_t st = as_stream(stream);
    double* partial = static_cast<double*>(ws);
    for (int base = 0; base < n_params; base += HAT_JOBS) {
        SgdJobs J;
        J.n = n_params - base < HAT_JOBS ? n_params - base : HAT_JOBS;
        J.pad = 0;
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) {
            const clhip_hat_param& p = params[base + i];
            size_t nb = (p.n + (size_t)HB * 16 - 1) / ((size_t)HB * 16);       // ~16 elements per thread
            if (nb < 1) nb = 1;
            if (nb > SGD_MAX_BLOCKS) nb = SGD_MAX_BLOCKS;
            J.j[i] = SgdJob{p.theta, p.grad, p.buf, p.mask_back, p.n, p.is_emb, blocks, (int)nb, 0};
            blocks += (int)nb;
        }
        double* part = partial + (size_t)base * SGD_MAX_BLOCKS;
        hipLaunchKernelGGL(hat_sgd_prep_multi_kernel, dim3(blocks), dim3(HB), 0, st, J, wd, !finetune, s, smax, thres_cosh, part);
        CLHIP_LAUNCH_CHECK();
        hipLaunchKernelGGL(hat_sgd_apply_multi_kernel, dim3(blocks), dim3(HB), 0, st, J, lr, momentum, first, !finetune, clipgrad,
                           thres_emb, part);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

int clhip_hat_gates_multi(const clhip_hat_gate_job* jobs, int n_jobs, float s, double* sums2, void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > HAT_JOBS) return CLHIP_EINVAL;
    GateJobs J;
    J.n = n_jobs; J.pad = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].emb_row || !jobs[i].gate || jobs[i].n <= 0) return CLHIP_EINVAL;
        J.j[i] = jobs[i];
    }
    hipLaunchKernelGGL(hat_gates_multi_kernel, dim3(1), dim3(HB), 0, as_stream(stream), J, s, sums2);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_hat_scale_weights_multi(const clhip_hat_layer* layers, int n_layers, void* stream) {
    if (!layers || n_layers <= 0) return CLHIP_EINVAL;
    for (int base = 0; base < n_layers; base += HAT_JOBS) {
        ScaleJobs J;
        J.n = n_layers - base < HAT_JOBS ? n_layers - base : HAT_JOBS;
        J.pad = 0;
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) {
            const clhip_hat_layer& l = layers[base + i];
            if (!l.w || !l.out || l.K == 0 || l.C == 0 || l.R == 0) return CLHIP_EINVAL;
            const size_t total = l.K * l.C * l.R;
            size_t nb = (total + (size_t)HB * 16 - 1) / ((size_t)HB * 16);
            if (nb < 1) nb = 1;
            if (nb > 2048) nb = 2048;
            J.j[i] = ScaleJob{l.w, l.gate_in, l.out, total, l.C, l.R, blocks, (int)nb};
            blocks += (int)nb;
        }
        hipLaunchKernelGGL(hat_scale_multi_kernel, dim3(blocks), dim3(HB), 0, as_stream(stream), J);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

int clhip_hat_weight_grads_multi(const clhip_hat_wgrad_job* jobs, int n_jobs, void* stream) {
    if (!jobs || n_jobs <= 0) return CLHIP_EINVAL;
    for (int base = 0; base < n_jobs; base += HAT_JOBS) {
        WgJobs J;
        J.n = n_jobs - base < HAT_JOBS ? n_jobs - base : HAT_JOBS;
        J.pad = 0;
        int blocks = 0;
        for (int i = 0; i < J.n; ++i) {
            const clhip_hat_wgrad_job& w = jobs[base + i];
            if (!w.g || !w.w || !w.gate_in || !w.dgate_in || w.K <= 0 || w.C <= 0 || w.R <= 0) return CLHIP_EINVAL;
            J.j[i] = WgJob{w.g, w.w, w.gate_in, w.dgate_in, w.K, w.C, w.R, blocks};
            blocks += w.C;
        }
        hipLaunchKernelGGL(hat_weight_grad_multi_kernel, dim3(blocks), dim3(HB), 0, as_stream(stream), J);
        CLHIP_LAUNCH_CHECK();
    }
    return 0;
}

int clhip_hat_emb_grads_multi(const clhip_hat_emb_job* jobs, int n_jobs, float s, float lamb, float count, const double* sums2,
                              void* stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > HAT_JOBS || (count <= 0.f && !sums2)) return CLHIP_EINVAL;
    EmbJobs J;
    J.n = n_jobs; J.pad = 0;
    for (int i = 0; i < n_jobs; ++i) {
        if (!jobs[i].dgate || !jobs[i].gate || !jobs[i].demb || jobs[i].n <= 0 || jobs[i].rows <= 0) return CLHIP_EINVAL;
        J.j[i] = jobs[i];
    }
    hipLaunchKernelGGL(hat_emb_grads_multi_kernel, dim3(n_jobs), dim3(HB), 0, as_stream(stream), J, s, lamb, count, sums2);
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
