// Shared helpers for libclhip (gfx950 only — no multi-arch or CUDA paths by design).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/clhip.h"

#define CLHIP_LAUNCH_CHECK()                          \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Grid for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8, capped.
static inline int ew_grid(size_t work_items, int block) {
    size_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

// 16 zero bytes in device memory. Out-of-range / halo elements are loaded FROM HERE (address select) instead
// of "load, then select 0": the select would consume the loaded register at once and make hipcc put an
// s_waitcnt vmcnt(0) right behind every global load, i.e. serialise each load's full latency in front of the
// MFMA loop (measured: conv kernels at ~55 % of what the same MFMA loop sustains in isolation).
static __device__ __attribute__((aligned(16))) float clhip_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // non-const: stays in the global address space (a const would be addrspace(4) and turn the selected load into flat_load)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32 fragment maps (cdna_hip_programming.md §3):
//   A operand: lane l supplies A[i = l & 31][k = l >> 5]
//   B operand: lane l supplies B[k = l >> 5][j = l & 31]
//   D: reg r of lane l holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
