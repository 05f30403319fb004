// HBM-bound fused optimizer / importance-weight kernels (one pass over the parameter arena).
// Reference arithmetic restated per kernel; algorithmic bytes per parameter in DESIGN.md.
#include "common.hpp"

namespace {

constexpr int EW_BLOCK = 256;

// Generic driver: applies F to n elements, float4-vectorised when every pointer is 16B aligned.
template <class F>
__global__ __launch_bounds__(EW_BLOCK) void ew_kernel(size_t n, size_t n4, F f) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t i = tid; i < n4; i += stride) f.vec(i);
    for (size_t i = n4 * 4 + tid; i < n; i += stride) f.scalar(i);
}

template <class F>
int ew_launch(size_t n, bool vec_ok, F f, hipStream_t s) {
    if (n == 0) return 0;
    size_t n4 = vec_ok ? n / 4 : 0;
    size_t items = vec_ok ? (n4 > 0 ? n4 : 1) : n;
    hipLaunchKernelGGL(ew_kernel<F>, dim3(ew_grid(items, EW_BLOCK)), dim3(EW_BLOCK), 0, s, n, n4, f);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

struct RegSgd {
    float* theta; const float* grad; const float* omega; const float* init; float* buf;
    float two_lambda, lr, momentum, wd; int first;
    __device__ __forceinline__ float one(float th, float g, float om, float iv, float& b) const {
        float d = g + (th - iv) * (two_lambda * om);   // train_EWC.py:62-65
        d = d + wd * th;                               // :70-71
        b = first ? d : (b * momentum + d);            // :72-78
        return th - lr * b;                            // :83
    }
    __device__ __forceinline__ void scalar(size_t i) const {
        float b = first ? 0.f : buf[i];
        float om = omega ? omega[i] : 0.f, iv = omega ? init[i] : 0.f;
        float t = one(theta[i], grad[i], om, iv, b);
        buf[i] = b; theta[i] = t;
    }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 th = reinterpret_cast<const float4*>(theta)[i];
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        float4 om = omega ? reinterpret_cast<const float4*>(omega)[i] : make_float4(0, 0, 0, 0);
        float4 iv = omega ? reinterpret_cast<const float4*>(init)[i] : make_float4(0, 0, 0, 0);
        float4 b = first ? make_float4(0, 0, 0, 0) : reinterpret_cast<const float4*>(buf)[i];
        th.x = one(th.x, g.x, om.x, iv.x, b.x); th.y = one(th.y, g.y, om.y, iv.y, b.y);
        th.z = one(th.z, g.z, om.z, iv.z, b.z); th.w = one(th.w, g.w, om.w, iv.w, b.w);
        reinterpret_cast<float4*>(buf)[i] = b;
        reinterpret_cast<float4*>(theta)[i] = th;
    }
};

struct FisherAcc {
    float* omega; const float* grad; float data_len;
    __device__ __forceinline__ void scalar(size_t i) const { float g = grad[i]; omega[i] += g * g / data_len; }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 o = reinterpret_cast<float4*>(omega)[i];
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        o.x += g.x * g.x / data_len; o.y += g.y * g.y / data_len;
        o.z += g.z * g.z / data_len; o.w += g.w * g.w / data_len;
        reinterpret_cast<float4*>(omega)[i] = o;
    }
};

struct MasAcc {
    float* omega; const float* grad; float prev, curr;
    __device__ __forceinline__ float one(float o, float g) const { return (o * prev + fabsf(g)) / curr; }
    __device__ __forceinline__ void scalar(size_t i) const { omega[i] = one(omega[i], grad[i]); }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 o = reinterpret_cast<float4*>(omega)[i];
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        o.x = one(o.x, g.x); o.y = one(o.y, g.y); o.z = one(o.z, g.z); o.w = one(o.w, g.w);
        reinterpret_cast<float4*>(omega)[i] = o;
    }
};

struct SiStep {
    float* theta; const float* grad; const float* omega; const float* init; float* w; float* buf;
    float two_lambda, lr, momentum, wd; int first;
    __device__ __forceinline__ float one(float th, float g, float om, float iv, float& b, float& wv) const {
        float d = g + (th - iv) * (two_lambda * om);   // train_SI.py:69-73
        d = d + wd * th;                               // :82-83
        b = first ? d : (b * momentum + d);            // :85-96
        float tn = th - lr * b;                        // :98
        wv = wv + ((tn - th) * g) * -1.f;              // :99-120 (unregularised g, actual step)
        return tn;
    }
    __device__ __forceinline__ void scalar(size_t i) const {
        float b = first ? 0.f : buf[i]; float wv = w[i];
        float t = one(theta[i], grad[i], omega[i], init[i], b, wv);
        buf[i] = b; w[i] = wv; theta[i] = t;
    }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 th = reinterpret_cast<const float4*>(theta)[i];
        float4 g = reinterpret_cast<const float4*>(grad)[i];
        float4 om = reinterpret_cast<const float4*>(omega)[i];
        float4 iv = reinterpret_cast<const float4*>(init)[i];
        float4 wv = reinterpret_cast<const float4*>(w)[i];
        float4 b = first ? make_float4(0, 0, 0, 0) : reinterpret_cast<const float4*>(buf)[i];
        th.x = one(th.x, g.x, om.x, iv.x, b.x, wv.x); th.y = one(th.y, g.y, om.y, iv.y, b.y, wv.y);
        th.z = one(th.z, g.z, om.z, iv.z, b.z, wv.z); th.w = one(th.w, g.w, om.w, iv.w, b.w, wv.w);
        reinterpret_cast<float4*>(buf)[i] = b;
        reinterpret_cast<float4*>(w)[i] = wv;
        reinterpret_cast<float4*>(theta)[i] = th;
    }
};

struct SiCons {
    float* omega; float* w; const float* theta; float* init; float slack;
    __device__ __forceinline__ float one(float o, float wv, float th, float iv) const {
        float pd = th - iv;
        float t = wv / (pd * pd + slack);              // train_SI.py:331-334
        return o + fmaxf(t, 0.f);                      // :346-350
    }
    __device__ __forceinline__ void scalar(size_t i) const {
        float th = theta[i];
        omega[i] = one(omega[i], w[i], th, init[i]); w[i] = 0.f; init[i] = th;
    }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 th = reinterpret_cast<const float4*>(theta)[i];
        float4 o = reinterpret_cast<float4*>(omega)[i];
        float4 wv = reinterpret_cast<float4*>(w)[i];
        float4 iv = reinterpret_cast<float4*>(init)[i];
        o.x = one(o.x, wv.x, th.x, iv.x); o.y = one(o.y, wv.y, th.y, iv.y);
        o.z = one(o.z, wv.z, th.z, iv.z); o.w = one(o.w, wv.w, th.w, iv.w);
        reinterpret_cast<float4*>(omega)[i] = o;
        reinterpret_cast<float4*>(w)[i] = make_float4(0, 0, 0, 0);
        reinterpret_cast<float4*>(init)[i] = th;
    }
};

// IMM merge (methods/IMM/merge.py:185-242), one parameter tensor of up to 32 task models per launch:
//   mean-IMM: out = (theta_0 + ... + theta_{M-1}) / M                      (:222-232)
//   mode-IMM: out = sum_m (F_m / S) * theta_m,  S = sum of the precisions  (:226-228)
// Same operation order as the reference's per-tensor torch ops (running sum from zero, divide last), so the merged
// weights agree with it to the last bit.
constexpr int IMM_MAX = 32;
struct ImmPtrs { const float* theta[IMM_MAX]; const float* prec[IMM_MAX]; };
struct ImmMerge {
    ImmPtrs p; const float* sum_prec; float* out; int n_models; int mean_mode;
    __device__ __forceinline__ void scalar(size_t i) const {
        float acc = 0.f;
        if (mean_mode) {
            for (int m = 0; m < n_models; ++m) acc = __fadd_rn(acc, p.theta[m][i]);
            acc = __fdiv_rn(acc, (float)n_models);
        } else {
            const float sp = sum_prec[i];
            // separate, correctly rounded div / mul / add like the reference's three torch ops (no fma contraction)
            for (int m = 0; m < n_models; ++m) acc = __fadd_rn(acc, __fmul_rn(__fdiv_rn(p.prec[m][i], sp), p.theta[m][i]));
        }
        out[i] = acc;
    }
    __device__ __forceinline__ void vec(size_t i) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) scalar(4 * i + t);
    }
};

struct ReluBwd {
    const float* dy; const float* y; float* dx;
    __device__ __forceinline__ void scalar(size_t i) const { dx[i] = y[i] > 0.f ? dy[i] : 0.f; }
    __device__ __forceinline__ void vec(size_t i) const {
        float4 a = reinterpret_cast<const float4*>(dy)[i];
        float4 b = reinterpret_cast<const float4*>(y)[i];
        a.x = b.x > 0.f ? a.x : 0.f; a.y = b.y > 0.f ? a.y : 0.f;
        a.z = b.z > 0.f ? a.z : 0.f; a.w = b.w > 0.f ? a.w : 0.f;
        reinterpret_cast<float4*>(dx)[i] = a;
    }
};

// EBLL (methods/EBLL/AlexNet_EBLL.py:13-15: Linear + Sigmoid encoder; Finetune_SGD_EBLL.py:497: optim.Adadelta)
struct SigmoidFwd {
    const float* x; float* y;
    __device__ __forceinline__ void scalar(size_t i) const { y[i] = 1.f / (1.f + expf(-x[i])); }
    __device__ __forceinline__ void vec(size_t i) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) scalar(4 * i + t);
    }
};
struct SigmoidBwd {
    const float* dy; const float* y; float* dx;
    __device__ __forceinline__ void scalar(size_t i) const { const float v = y[i]; dx[i] = dy[i] * v * (1.f - v); }
    __device__ __forceinline__ void vec(size_t i) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) scalar(4 * i + t);
    }
};
// torch.optim.Adadelta (rho, eps, lr, weight_decay): square_avg = rho sq + (1 - rho) g^2; delta = sqrt(acc_delta + eps) /
// sqrt(square_avg + eps) * g; acc_delta = rho acc_delta + (1 - rho) delta^2; theta -= lr * delta
struct Adadelta {
    float* theta; const float* grad; float* sq; float* acc; float lr, rho, eps, wd;
    __device__ __forceinline__ void scalar(size_t i) const {
        float g = grad[i];
        const float th = theta[i];
        if (wd != 0.f) g = g + wd * th;
        const float s = sq[i] * rho + (1.f - rho) * g * g;
        const float a = acc[i];
        const float delta = sqrtf(a + eps) / sqrtf(s + eps) * g;
        sq[i] = s;
        acc[i] = a * rho + (1.f - rho) * delta * delta;
        theta[i] = th - lr * delta;
    }
    __device__ __forceinline__ void vec(size_t i) const {
#pragma unroll
        for (int t = 0; t < 4; ++t) scalar(4 * i + t);
    }
};

// nn.MSELoss() (mean over all elements) of a against b, with d loss / d a * scale; one block, fixed order
__global__ __launch_bounds__(1024) void mse_mean_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                        float scale, float* __restrict__ da, float* __restrict__ loss_out) {
    __shared__ double red[1024];
    double s = 0.0;
    const float g = 2.f * scale / (float)n;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const float d = a[i] - b[i];
        s += (double)d * (double)d;
        if (da) da[i] = g * d;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_out[0] = (float)(red[0] / (double)n);
}

}  // namespace

extern "C" {

int clhip_sigmoid_fwd(const float* x, float* y, size_t n, void* stream) {
    if (!x || !y) return CLHIP_EINVAL;
    return ew_launch(n, false, SigmoidFwd{x, y}, as_stream(stream));
}

int clhip_sigmoid_bwd(const float* dy, const float* y, float* dx, size_t n, void* stream) {
    if (!dy || !y || !dx) return CLHIP_EINVAL;
    return ew_launch(n, false, SigmoidBwd{dy, y, dx}, as_stream(stream));
}

int clhip_adadelta_step(float* theta, const float* grad, float* square_avg, float* acc_delta, size_t n, float lr, float rho,
                        float eps, float weight_decay, void* stream) {
    if (!theta || !grad || !square_avg || !acc_delta) return CLHIP_EINVAL;
    return ew_launch(n, false, Adadelta{theta, grad, square_avg, acc_delta, lr, rho, eps, weight_decay}, as_stream(stream));
}

int clhip_mse_mean(const float* a, const float* b, size_t n, float grad_scale, float* da, float* loss_out, void* stream) {
    if (!a || !b || !loss_out || n == 0) return CLHIP_EINVAL;
    hipLaunchKernelGGL(mse_mean_kernel, dim3(1), dim3(1024), 0, as_stream(stream), a, b, n, grad_scale, da, loss_out);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_reg_sgd_step(float* theta, const float* grad, const float* omega, const float* init_val,
                       float* buf, size_t n, float reg_lambda, float lr, float momentum, float wd,
                       int first, void* stream) {
    if (!theta || !grad || !buf || (omega && !init_val)) return CLHIP_EINVAL;
    RegSgd f{theta, grad, omega, init_val, buf, 2.f * reg_lambda, lr, momentum, wd, first};
    bool v = aligned16(theta) && aligned16(grad) && aligned16(buf) && (!omega || (aligned16(omega) && aligned16(init_val)));
    return ew_launch(n, v, f, as_stream(stream));
}

int clhip_fisher_accum(float* omega, const float* grad, size_t n, float data_len, void* stream) {
    if (!omega || !grad || data_len <= 0.f) return CLHIP_EINVAL;
    FisherAcc f{omega, grad, data_len};
    return ew_launch(n, aligned16(omega) && aligned16(grad), f, as_stream(stream));
}

int clhip_mas_accum(float* omega, const float* grad, size_t n, float prev_size, float curr_size, void* stream) {
    if (!omega || !grad || curr_size <= 0.f) return CLHIP_EINVAL;
    MasAcc f{omega, grad, prev_size, curr_size};
    return ew_launch(n, aligned16(omega) && aligned16(grad), f, as_stream(stream));
}

int clhip_si_step(float* theta, const float* grad, const float* omega, const float* init_val, float* w,
                  float* buf, size_t n, float reg_lambda, float lr, float momentum, float wd, int first,
                  void* stream) {
    if (!theta || !grad || !omega || !init_val || !w || !buf) return CLHIP_EINVAL;
    SiStep f{theta, grad, omega, init_val, w, buf, 2.f * reg_lambda, lr, momentum, wd, first};
    bool v = aligned16(theta) && aligned16(grad) && aligned16(omega) && aligned16(init_val) && aligned16(w) && aligned16(buf);
    return ew_launch(n, v, f, as_stream(stream));
}

int clhip_si_consolidate(float* omega, float* w, const float* theta, float* init_val, size_t n, float slack,
                         void* stream) {
    if (!omega || !w || !theta || !init_val) return CLHIP_EINVAL;
    SiCons f{omega, w, theta, init_val, slack};
    bool v = aligned16(omega) && aligned16(w) && aligned16(theta) && aligned16(init_val);
    return ew_launch(n, v, f, as_stream(stream));
}

int clhip_relu_bwd(const float* dy, const float* y, float* dx, size_t n, void* stream) {
    if (!dy || !y || !dx) return CLHIP_EINVAL;
    ReluBwd f{dy, y, dx};
    return ew_launch(n, aligned16(dy) && aligned16(y) && aligned16(dx), f, as_stream(stream));
}

int clhip_imm_merge(const float* const* thetas, const float* const* precisions, const float* sum_precision, int n_models,
                    size_t n, float* out, void* stream) {
    if (!thetas || !out || n_models < 1 || n_models > IMM_MAX) return CLHIP_EINVAL;
    if (precisions && !sum_precision) return CLHIP_EINVAL;
    ImmMerge f{};
    for (int m = 0; m < n_models; ++m) {
        if (!thetas[m] || (precisions && !precisions[m])) return CLHIP_EINVAL;
        f.p.theta[m] = thetas[m];
        f.p.prec[m] = precisions ? precisions[m] : nullptr;
    }
    f.sum_prec = sum_precision; f.out = out; f.n_models = n_models; f.mean_mode = precisions ? 0 : 1;
    return ew_launch(n, false, f, as_stream(stream));
}

int clhip_version(void) { return 100; }
const char* clhip_arch(void) { return "gfx950"; }

}  // extern "C"
