// Micro-benchmark 2: LDS-fed v_mfma_f32_32x32x2_f32 with explicit register double-buffering of the operands
// (the structure of conv3x3.hip's inner loop), to find the ceiling of that structure on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

// per "cp": 9 A reads + 9*NT B reads, 9*NT MFMAs; operands of cp+1 read before the MFMAs of cp
template <int NT, bool SB>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    floatx16 acc[NT];
    for (int a = 0; a < NT; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* wsb = lds + (lane >> 5) * 585 + (lane & 31);
    const float* xsb = lds + 6000 + (lane >> 5) * 204 + (lane & 31);
    float af[2][9], bf[2][NT][9];
    auto load = [&](int cp, int slot) {
#pragma unroll
        for (int rs = 0; rs < 9; ++rs) {
            af[slot][rs] = wsb[((2 * cp) * 9 + rs) * 65];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[slot][t][rs] = xsb[(2 * cp) * 204 + t * 34 + (rs / 3) * 34 + rs % 3];
        }
    };
    for (int it = 0; it < iters; ++it) {
        load(0, 0);
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) {
            if (cp + 1 < 4) load(cp + 1, (cp + 1) & 1);
            if (SB) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rs = 0; rs < 9; ++rs)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cp & 1][rs], bf[cp & 1][t][rs], acc[t], 0, 0, 0);
            if (SB) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int a = 0; a < NT; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NT, bool SB>
void run(const char* name, int blocks) {
    float* out; hipMalloc(&out, blocks * 256 * 4);
    int iters = 300;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NT, SB><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NT, SB><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 36.0 * NT * 4096.0;
    printf("%-34s blocks %4d: %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<2, true>("NT=2 prefetch+sched_barrier", 256);
    run<2, true>("NT=2 prefetch+sched_barrier", 512);
    run<2, false>("NT=2 prefetch, compiler order", 256);
    run<2, false>("NT=2 prefetch, compiler order", 512);
    run<1, true>("NT=1 prefetch+sched_barrier", 512);
    run<4, true>("NT=4 prefetch+sched_barrier", 256);
    run<4, true>("NT=4 prefetch+sched_barrier", 512);
    return 0;
}
