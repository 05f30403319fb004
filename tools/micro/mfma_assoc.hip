// Micro-test: is the summation association of fp32 MFMA shapes interchangeable?
//   v_mfma_f32_32x32x2_f32 adds two products per instruction to its accumulator, v_mfma_f32_16x16x4_f32 four.  The conv
//   kernels' bit-compatibility rule forbids any change of association; if one 16x16x4 step gives the SAME
//   bits as two consecutive 32x32x2 steps over the same k order, 16x16 accumulator tiles (a 4x finer work unit for the
//   balance problem of DESIGN §9) would be admissible.  Every row of A and every column of B is the same vector here, so
//   each output element is sum_k a[k] b[k] in whatever order the hardware uses, independent of the register layout.
// Also fitted: CPU models of the per-instruction arithmetic (sequential fma chain; exact products + one rounding per
// instruction; exact products + one rounding per product pair).  Tuning aid, not part of libclhip.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int K = 576;      // 64 channels x 9 taps: one chain of the 64-channel layers

__global__ __launch_bounds__(64) void chains(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                             int* __restrict__ nonuniform) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const float* at = a + (size_t)t * K;
    const float* bt = b + (size_t)t * K;
    floatx16 c32;
    for (int r = 0; r < 16; ++r) c32[r] = 0.f;
    for (int k = 0; k < K; k += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[k + (lane >> 5)], bt[k + (lane >> 5)], c32, 0, 0, 0);
    floatx4 c16 = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; k += 4) c16 = __builtin_amdgcn_mfma_f32_16x16x4f32(at[k + (lane >> 4)], bt[k + (lane >> 4)], c16, 0, 0, 0);
    // two half chains in two accumulators, added at the end (what a split of the reduction over two waves would compute)
    floatx16 h0, h1;
    for (int r = 0; r < 16; ++r) h0[r] = h1[r] = 0.f;
    for (int k = 0; k < K / 2; k += 2) h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[k + (lane >> 5)], bt[k + (lane >> 5)], h0, 0, 0, 0);
    for (int k = K / 2; k < K; k += 2) h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(at[k + (lane >> 5)], bt[k + (lane >> 5)], h1, 0, 0, 0);
    bool uni = true;
    for (int r = 0; r < 16; ++r) uni = uni && (__float_as_uint(c32[r]) == __float_as_uint(c32[0]));
    for (int r = 0; r < 4; ++r) uni = uni && (__float_as_uint(c16[r]) == __float_as_uint(c16[0]));
    const unsigned first32 = __builtin_amdgcn_readfirstlane(__float_as_uint(c32[0]));
    const unsigned first16 = __builtin_amdgcn_readfirstlane(__float_as_uint(c16[0]));
    uni = uni && __float_as_uint(c32[0]) == first32 && __float_as_uint(c16[0]) == first16;
    if (!uni) atomicAdd(nonuniform, 1);
    if (lane == 0) {
        out[3 * t + 0] = c32[0];
        out[3 * t + 1] = c16[0];
        out[3 * t + 2] = h0[0] + h1[0];
    }
}

template <bool SMALL>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    float av = threadIdx.x * 1e-3f, bv = 1.f + threadIdx.x * 1e-4f, r = 0.f;
    if (SMALL) {
        floatx4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv + i, c[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
    } else {
        floatx16 c[2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 2; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv + i, c[i], 0, 0, 0);
        for (int i = 0; i < 2; ++i) r += c[i][0] + c[i][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <bool SMALL>
static void time_rate(const char* name) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rate<SMALL><<<256, 256>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate<SMALL><<<256, 256>>>(out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * iters * (SMALL ? 8 * 2.0 * 16 * 16 * 4 : 2 * 2.0 * 32 * 32 * 2);
    printf("%-40s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
    const int T = 8192;
    std::vector<float> a((size_t)T * K), b((size_t)T * K);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (size_t i = 0; i < a.size(); ++i) { a[i] = rnd(); b[i] = rnd() * 0.1f; }
    float *da, *db, *dout; int* dn;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, (size_t)T * 3 * 4); hipMalloc(&dn, 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipMemset(dn, 0, 4);
    chains<<<T, 64>>>(da, db, dout, dn);
    std::vector<float> out((size_t)T * 3);
    int nonuni = 0;
    if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
    hipMemcpy(&nonuni, dn, 4, hipMemcpyDeviceToHost);
    int same_16_32 = 0, same_split = 0, m_fma32 = 0, m_pair32 = 0, m_exact2_32 = 0, m_quad16 = 0, m_pairpair16 = 0, m_fma16 = 0;
    for (int t = 0; t < T; ++t) {
        const float* at = &a[(size_t)t * K];
        const float* bt = &b[(size_t)t * K];
        const uint32_t g32 = bits(out[3 * t]), g16 = bits(out[3 * t + 1]), gs = bits(out[3 * t + 2]);
        same_16_32 += g32 == g16;
        same_split += g32 == gs;
        float fma = 0.f, pair = 0.f, ex2 = 0.f, quad = 0.f, pp = 0.f;
        for (int k = 0; k < K; ++k) fma = fmaf(at[k], bt[k], fma);                        // sequential fma chain
        for (int k = 0; k < K; k += 2) {
            // products exact (double holds a 48-bit product), summed in double, ONE rounding per instruction
            ex2 = (float)((double)ex2 + (double)at[k] * bt[k] + (double)at[k + 1] * bt[k + 1]);
            // pair of products rounded to fp32 first, then added
            pair = pair + (float)((double)at[k] * bt[k] + (double)at[k + 1] * bt[k + 1]);
        }
        for (int k = 0; k < K; k += 4) {
            double d = (double)quad;
            for (int j = 0; j < 4; ++j) d += (double)at[k + j] * bt[k + j];
            quad = (float)d;
            pp = (float)((double)pp + (double)at[k] * bt[k] + (double)at[k + 1] * bt[k + 1]);
            pp = (float)((double)pp + (double)at[k + 2] * bt[k + 2] + (double)at[k + 3] * bt[k + 3]);
        }
        m_fma32 += bits(fma) == g32;  m_fma16 += bits(fma) == g16;
        m_pair32 += bits(pair) == g32;  m_exact2_32 += bits(ex2) == g32;
        m_quad16 += bits(quad) == g16;  m_pairpair16 += bits(pp) == g16;
    }
    printf("trials %d, K %d, outputs not uniform within a wave: %d\n", T, K, nonuni);
    printf("16x16x4 chain == 32x32x2 chain (bitwise):            %d / %d\n", same_16_32, T);
    printf("two half chains added == one 32x32x2 chain:          %d / %d\n", same_split, T);
    printf("32x32x2 == sequential fmaf chain:                    %d\n", m_fma32);
    printf("32x32x2 == acc + exact(p0 + p1), one rounding:       %d\n", m_exact2_32);
    printf("32x32x2 == acc + round(p0 + p1):                     %d\n", m_pair32);
    printf("16x16x4 == sequential fmaf chain:                    %d\n", m_fma16);
    printf("16x16x4 == acc + exact(p0..p3), one rounding:        %d\n", m_quad16);
    printf("16x16x4 == two exact-pair steps:                     %d\n", m_pairpair16);
    time_rate<false>("32x32x2, 2 accumulators, 1 wave / SIMD");
    time_rate<true>("16x16x4, 8 accumulators, 1 wave / SIMD");
    return 0;
}
