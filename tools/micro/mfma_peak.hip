// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this box in the issue patterns the
// conv kernels use?  (tuning aid, not part of libclhip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDSREAD>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = threadIdx.x * 1e-3f, bv[NACC];
    for (int a = 0; a < NACC; ++a) bv[a] = a + threadIdx.x * 1e-4f;
    const float* p = lds + (threadIdx.x & 63) * 65;
    for (int it = 0; it < iters; ++it) {
        if (LDSREAD) {
            av = p[it & 63];
#pragma unroll
            for (int a = 0; a < NACC; ++a) bv[a] = p[((it + a) & 63) + 100];
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[a], acc[a], 0, 0, 0);
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool LDSREAD>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDSREAD><<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDSREAD><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 4096.0;
    printf("%-28s blocks %4d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1, false>("1acc pure", 256);
    run<2, false>("2acc pure", 256);
    run<3, false>("3acc pure", 256);
    run<1, false>("1acc pure", 512);
    run<2, false>("2acc pure", 512);
    run<4, true>("4acc + 5 lds reads", 256);
    run<4, true>("4acc + 5 lds reads", 512);
    run<4, false>("4acc pure", 256);
    run<9, false>("9acc pure", 256);
    run<9, true>("9acc + 10 lds reads", 256);
    run<2, true>("2acc + 3 lds reads", 256);
    run<2, true>("2acc + 3 lds reads", 512);
    run<1, true>("1acc + 2 lds reads", 512);
    run<9, false>("9acc pure", 512);
    run<9, false>("9acc pure", 1024);
    return 0;
}
