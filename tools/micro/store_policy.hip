// store_policy.hip — what a dependent kernel boundary costs behind a kernel that leaves X MB of freshly written lines in the XCD L2s,
// and whether the cache policy of the stores changes it (gfx950; buffer stores, aux bits: 1 = sc0, 2 = nt, 16 = sc1).
//   writer<AUX>: every thread writes float4s (grid-stride) over X MB, value derived from its index (+ a little ALU so that the kernel
//                is not purely store-bound: `work` fma per float4)
//   reader:      one load per block of what the writer wrote (a dependent kernel on the same stream)
// Timed with HIP events over `iters` writer + reader pairs; also the writer alone back to back.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/store_policy tools/micro/store_policy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void writer(float* out, size_t n4, int work, float seed) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0x7fffffff, 0x00020000);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float v = seed + (float)(i & 1023);
        for (int k = 0; k < work; ++k) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        const u32x4 q = {__float_as_uint(v), __float_as_uint(v + 1.f), __float_as_uint(v + 2.f), __float_as_uint(v + 3.f)};
        // (offsets stay below 2 GB: X <= 1024 MB)
        __builtin_amdgcn_raw_buffer_store_b128(q, r, (int)(i * 16), 0, AUX);
    }
}

__global__ __launch_bounds__(64) void reader(const float* in, size_t n, float* sink) {
    const size_t i = ((size_t)blockIdx.x * 7919 * 64 + threadIdx.x) % n;
    if (in[i] == -12345.f) sink[0] = 1.f;
}

template <int AUX>
static void run(float* buf, float* sink, size_t bytes, int work, int iters) {
    const size_t n4 = bytes / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float pair_ms = 0.f, alone_ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(writer<AUX>, dim3(2048), dim3(256), 0, 0, buf, n4, work, (float)it);
            hipLaunchKernelGGL(reader, dim3(256), dim3(64), 0, 0, buf, bytes / 4, sink);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0 || ms < pair_ms) pair_ms = ms;
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(writer<AUX>, dim3(2048), dim3(256), 0, 0, buf, n4, work, (float)it);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0 || ms < alone_ms) alone_ms = ms;
    }
    printf("X = %4zu MB  work %3d  aux %2d (%s%s%s)  writer+reader %7.2f us   writer back to back %7.2f us   = %6.2f TB/s\n", bytes >> 20, work, AUX,
           AUX & 1 ? "sc0 " : "", AUX & 16 ? "sc1 " : "", AUX & 2 ? "nt" : (AUX ? "" : "default"), pair_ms * 1e3 / iters, alone_ms * 1e3 / iters,
           bytes / (alone_ms * 1e-3 / iters) / 1e12);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 50;
    float *buf, *sink;
    hipMalloc(&buf, (size_t)1 << 30);
    hipMalloc(&sink, 256);
    hipMemset(buf, 0, (size_t)1 << 30);
    for (int work : {0, 64}) {
        for (size_t mb : {2, 8, 16, 32, 64, 128}) {
            const size_t bytes = mb << 20;
            run<0>(buf, sink, bytes, work, iters);
            run<2>(buf, sink, bytes, work, iters);
            run<16>(buf, sink, bytes, work, iters);
            run<17>(buf, sink, bytes, work, iters);
            run<18>(buf, sink, bytes, work, iters);
            run<19>(buf, sink, bytes, work, iters);
        }
    }
    // an empty pair: the boundary itself
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int it = 0; it < iters; ++it) {
            hipLaunchKernelGGL(reader, dim3(256), dim3(64), 0, 0, buf, (size_t)1 << 20, sink);
            hipLaunchKernelGGL(reader, dim3(256), dim3(64), 0, 0, buf, (size_t)1 << 20, sink);
        }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("two trivial kernels: %.2f us per pair\n", ms * 1e3 / iters);
    }
    return 0;
}
