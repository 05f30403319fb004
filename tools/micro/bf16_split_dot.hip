// Micro-benchmark 4 (round 5): can fp32 dot products run on the bf16 matrix cores?
//
// An fp32 value splits EXACTLY into three bf16 pieces a = a0 + a1 + a2 (8 + 8 + 8 significand bits, round-to-nearest at every
// level).  a*b = sum_{i,j} ai*bj; every bf16 x bf16 product is exact in fp32, and the terms (i + j <= 2) — six of the nine —
// carry the product to 2^-26 relative.  v_mfma_f32_32x32x16_bf16 runs at 16x the rate of the f32-input MFMA, so six of them
// per 16-deep k-step are 2.67x the f32 matrix rate — IF the hardware's fp32 accumulation inside a bf16 MFMA is as good as an
// fmaf chain, and IF the VALU work of a conv loop (staging, splitting) issues under bf16 MFMAs where it does not under fp32 ones.
//
// Part A (numerics): D[32][32] = A[32][L] * B[L][32], L = 576 .. 4608 (9 x 64 .. 9 x 512 input channels of a 3x3 conv), against
//   fp64, for: the f32-input MFMA chain the product kernels use today (32x32x2), a host fmaf chain, bf16 x1 / x3 / x6, and x6 with
//   the five small terms in an accumulator of their own.  Error unit: |d - ref| / sum_k |a_ik * b_kj| (the scale an fp32
//   rounding analysis bounds), max and rms over the 1024 outputs, three input distributions.
// Part B (rate): bf16 32x32x16 MFMAs, 1 and 2 waves per SIMD; beside a VALU-only sibling wave; with n VALU fillers per MFMA in
//   the same wave; the same three for the f32-input 16x16x4 MFMA.  Times are for 256 blocks x iters; TF/s from the MFMA count.
//
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/micro/bf16_split_dot.hip -o /tmp/bf16_split_dot && /tmp/bf16_split_dot
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned short bf16_rn(float f) {          // round-to-nearest-even, finite inputs
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_f(unsigned short h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
__host__ __device__ inline void split3(float a, unsigned short& h0, unsigned short& h1, unsigned short& h2) {
    h0 = bf16_rn(a);
    float r = a - bf16_f(h0);          // exact (Sterbenz-like: a and its bf16 rounding share the exponent range)
    h1 = bf16_rn(r);
    r -= bf16_f(h1);
    h2 = bf16_rn(r);
}

// ---------------------------------------------------------------------------------------------------------------- part A
// One wave.  A row-major [32][L], B row-major [L][32].  out[v][32][32], v = variant.
__global__ __launch_bounds__(64) void numerics(const float* A, const float* B, int L, float* out) {
    const int l = threadIdx.x, i = l & 31, kh = l >> 5;
    // v0: f32-input MFMA chain (lane: A[i][k = kh], B[k = kh][j = i] per 2-deep step)
    floatx16 c0 = {0};
    for (int k = 0; k < L; k += 2) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * L + k + kh], B[(k + kh) * 32 + i], c0, 0, 0, 0);
    floatx16 c1 = {0}, c3 = {0}, c6 = {0}, c6h = {0}, c6l = {0};
    for (int k = 0; k < L; k += 16) {
        u16x8 a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            unsigned short h0, h1, h2;
            split3(A[i * L + k + 8 * kh + e], h0, h1, h2);
            a[0][e] = h0; a[1][e] = h1; a[2][e] = h2;
            split3(B[(k + 8 * kh + e) * 32 + i], h0, h1, h2);
            b[0][e] = h0; b[1][e] = h1; b[2][e] = h2;
        }
        auto mm = [&](int p, int q, floatx16 c) {
            return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[p]), __builtin_bit_cast(bf16x8, b[q]), c, 0, 0, 0);
        };
        c1 = mm(0, 0, c1);
        c3 = mm(0, 1, c3); c3 = mm(1, 0, c3); c3 = mm(0, 0, c3);
        c6 = mm(0, 2, c6); c6 = mm(2, 0, c6); c6 = mm(1, 1, c6); c6 = mm(0, 1, c6); c6 = mm(1, 0, c6); c6 = mm(0, 0, c6);
        c6l = mm(0, 2, c6l); c6l = mm(2, 0, c6l); c6l = mm(1, 1, c6l); c6l = mm(0, 1, c6l); c6l = mm(1, 0, c6l);
        c6h = mm(0, 0, c6h);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kh, o = row * 32 + i;
        out[0 * 1024 + o] = c0[r];
        out[1 * 1024 + o] = c1[r];
        out[2 * 1024 + o] = c3[r];
        out[3 * 1024 + o] = c6[r];
        out[4 * 1024 + o] = c6h[r] + c6l[r];
    }
}

static float gauss() {
    double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v));
}

static void part_a() {
    const char* vn[6] = {"f32 MFMA 32x32x2 chain", "bf16 x1", "bf16 x3 (a0b0+a0b1+a1b0)", "bf16 x6, one accumulator", "bf16 x6, small terms apart",
                         "host fmaf chain"};
    const char* dn[3] = {"normal x normal", "relu(normal) x 0.05*normal (activations x weights)", "1e-4*normal x relu(normal) (gradients x activations)"};
    printf("== part A: error / sum_k|a*b| against fp64, max and rms over 32x32 outputs (fp32 unit roundoff 2^-24 = 5.96e-08)\n");
    float *dA, *dB, *dO;
    const int Lmax = 4608;
    hipMalloc(&dA, 32 * Lmax * 4); hipMalloc(&dB, 32 * Lmax * 4); hipMalloc(&dO, 5 * 1024 * 4);
    for (int dist = 0; dist < 3; ++dist)
        for (int L : {576, 1152, 2304, 4608}) {
            std::vector<float> A(32 * L), B(32 * L), O(5 * 1024);
            srand(1234 + L + dist);
            for (auto& x : A) { float g = gauss(); x = dist == 0 ? g : dist == 1 ? (g > 0 ? g : 0.f) : 1e-4f * g; }
            for (auto& x : B) { float g = gauss(); x = dist == 0 ? g : dist == 1 ? 0.05f * g : (g > 0 ? g : 0.f); }
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            numerics<<<1, 64>>>(dA, dB, L, dO);
            hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
            printf("-- %s, L = %d\n", dn[dist], L);
            for (int v = 0; v < 6; ++v) {
                double mx = 0, sq = 0, mxr = 0;
                for (int i = 0; i < 32; ++i)
                    for (int j = 0; j < 32; ++j) {
                        double ref = 0, sc = 0;
                        float ch = 0.f;
                        for (int k = 0; k < L; ++k) {
                            ref += (double)A[i * L + k] * B[k * 32 + j];
                            sc += fabs((double)A[i * L + k] * B[k * 32 + j]);
                            ch = fmaf(A[i * L + k], B[k * 32 + j], ch);
                        }
                        const double got = v < 5 ? O[v * 1024 + i * 32 + j] : ch;
                        const double e = fabs(got - ref) / (sc > 0 ? sc : 1);
                        mx = fmax(mx, e); sq += e * e;
                        mxr = fmax(mxr, fabs(got - ref) / fmax(fabs(ref), 1e-30));
                    }
                printf("   %-30s max %.3e  rms %.3e   (max |err|/|ref| %.3e)\n", vn[v], mx, sqrt(sq / 1024), mxr);
            }
        }
    hipFree(dA); hipFree(dB); hipFree(dO);
}

// ---------------------------------------------------------------------------------------------------------------- part B
// MFMA body: 8 independent accumulators (bf16 32x32x16) / 8 (f32 16x16x4), FILL VALU fillers (independent v_fma) after each MFMA.
template <int FILL>
__device__ __forceinline__ void bf16_body(float* out, int iters, bf16x8 a, bf16x8 b, float fa, float fb) {
    floatx16 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; v[i] = threadIdx.x * 0.001f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + f) & 7]) : "v"(fa), "v"(fb));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 16; ++r) s += acc[i][r]; s += v[i]; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int FILL>
__device__ __forceinline__ void f32_body(float* out, int iters, float fa, float fb) {
    floatx4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 4; ++r) acc[i][r] = 0.f; v[i] = threadIdx.x * 0.001f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(fa), "v"(fb));
#pragma unroll
            for (int f = 0; f < FILL; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(i + f) & 7]) : "v"(fa), "v"(fb));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) { for (int r = 0; r < 4; ++r) s += acc[i][r]; s += v[i]; }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__device__ __forceinline__ void valu_body(float* out, int n, float fa, float fb) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 0.001f + r;
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(fa), "v"(fb));
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// waves [0, MW) run MFMAs (mode & 1); waves [MW, MW + VW) run VALU only (mode & 2): per MFMA iteration (8 MFMAs = 256 matrix
// cycles) a VALU wave issues `vper` x 16 v_fma
template <bool BF16, int FILL>
__global__ __launch_bounds__(512) void rate(float* out, int iters, int mode, int mw, int vper, bf16x8 a, bf16x8 b, float fa, float fb) {
    const int wave = threadIdx.x >> 6;
    if (wave < mw) {
        if (mode & 1) { if (BF16) bf16_body<FILL>(out, iters, a, b, fa, fb); else f32_body<FILL>(out, iters, fa, fb); }
    } else if (mode & 2) valu_body(out, iters * vper, fa, fb);
}

template <bool BF16, int FILL>
static float time_rate(float* out, int threads, int iters, int mode, int mw, int vper) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)1.0f; b[e] = (__bf16)0.5f; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<BF16, FILL><<<256, threads>>>(out, 50, mode, mw, vper, a, b, 1.0f, 0.5f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rate<BF16, FILL><<<256, threads>>>(out, iters, mode, mw, vper, a, b, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms);
    }
    return best;
}

template <bool BF16>
static void part_b_one(float* out, const char* name, double flop_per_mfma) {
    const int iters = 4000;
    auto tf = [&](float ms, int mw) { return 256.0 * mw * iters * 8 * flop_per_mfma / (ms * 1e-3) / 1e12; };
    float t;
    t = time_rate<BF16, 0>(out, 256, iters, 1, 4, 0);  printf("%s  1 wave/SIMD, MFMA only:                 %.3f ms  %.0f TF\n", name, t, tf(t, 4));
    const float t1 = t;
    t = time_rate<BF16, 0>(out, 512, iters, 1, 8, 0);  printf("%s  2 waves/SIMD, MFMA only:                %.3f ms  %.0f TF\n", name, t, tf(t, 8));
    for (int vper : {2, 4, 8}) {
        float tv = time_rate<BF16, 0>(out, 512, iters, 2, 4, vper);
        float tb = time_rate<BF16, 0>(out, 512, iters, 3, 4, vper);
        printf("%s  MFMA wave + VALU sibling (%3d v_fma per 8 MFMAs): valu alone %.3f, mfma alone %.3f, both %.3f ms (sum %.3f, max %.3f)\n",
               name, 16 * vper, tv, t1, tb, tv + t1, fmaxf(tv, t1));
    }
    t = time_rate<BF16, 1>(out, 256, iters, 1, 4, 0);  printf("%s  same wave, 1 v_fma per MFMA:            %.3f ms (x%.2f)\n", name, t, t / t1);
    t = time_rate<BF16, 2>(out, 256, iters, 1, 4, 0);  printf("%s  same wave, 2 v_fma per MFMA:            %.3f ms (x%.2f)\n", name, t, t / t1);
    t = time_rate<BF16, 4>(out, 256, iters, 1, 4, 0);  printf("%s  same wave, 4 v_fma per MFMA:            %.3f ms (x%.2f)\n", name, t, t / t1);
    t = time_rate<BF16, 6>(out, 256, iters, 1, 4, 0);  printf("%s  same wave, 6 v_fma per MFMA:            %.3f ms (x%.2f)\n", name, t, t / t1);
    t = time_rate<BF16, 8>(out, 256, iters, 1, 4, 0);  printf("%s  same wave, 8 v_fma per MFMA:            %.3f ms (x%.2f)\n", name, t, t / t1);
    t = time_rate<BF16, 4>(out, 512, iters, 1, 8, 0);  printf("%s  2 waves/SIMD, 4 v_fma per MFMA each:    %.3f ms  %.0f TF\n", name, t, tf(t, 8));
}

int main() {
    part_a();
    printf("== part B: issue rates (256 blocks, one per CU)\n");
    float* out; hipMalloc(&out, 256 * 512 * 4);
    part_b_one<true>(out, "bf16 32x32x16", 32.0 * 32 * 16 * 2);
    part_b_one<false>(out, "f32  16x16x4 ", 16.0 * 16 * 4 * 2);
    hipFree(out);
    return 0;
}
