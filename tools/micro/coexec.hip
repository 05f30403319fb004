// Micro-benchmark 3: do an MFMA-only wave and a VALU-only wave on the SAME SIMD overlap?
// 8 waves per block: waves 0-3 run MFMAs (accumulators in VGPRs or, with ACC_AGPR, in AccVGPRs via inline asm),
// waves 4-7 run dependent-free v_fma chains.  Reports each alone and both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool AGPR>
__device__ __forceinline__ void mfma_body(float* out, int iters, float a, float b) {
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

__device__ __forceinline__ void valu_body(float* out, int iters, float a, float b) {
    float v[16];
    for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 0.001f + r;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(a), "v"(b));
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += v[r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// mode: 1 = MFMA waves only, 2 = VALU waves only, 3 = both
template <bool AGPR>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, float a, float b) {
    const int wave = threadIdx.x >> 6;
    if (wave < 4) { if (mode & 1) mfma_body<AGPR>(out, iters, a, b); }
    else { if (mode & 2) valu_body(out, iters * 4, a, b); }      // 4 MFMA (256 cyc) vs 4*64 VALU (2 cyc each ~ 512 cyc)
}

template <bool AGPR>
void run(const char* name) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 1; mode <= 3; ++mode) {
        k<AGPR><<<256, 512>>>(out, 100, mode, 1.0f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<AGPR><<<256, 512>>>(out, 4000, mode, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-10s mode %d (%s): %.3f ms\n", name, mode, mode == 1 ? "mfma" : mode == 2 ? "valu" : "both", ms);
    }
    hipFree(out);
}

int main() {
    run<false>("acc=VGPR");
    run<true>("acc=AGPR");
    return 0;
}
