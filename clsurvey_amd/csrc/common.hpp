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

// Raw buffer access (gfx9 buffer resource: base, stride 0, num_records in bytes, DATA_FORMAT_32): the hardware range
// check (voffset >= num_records; the scalar offset is NOT part of it) returns 0 for loads and drops stores, so a
// predicate costs one v_cndmask on the 32-bit offset instead of a select on the value (which waits for the load) or
// an exec-mask branch.  num_records is clipped so that CLHIP_OOB is always out of range.
constexpr int CLHIP_OOB = (int)0x80000000;

// 2x2 max-pool arg-max bytes of the fused conv + ReLU + pool kernels: 0..3 = window position (row-major, first maximum in
// ATen's scan order), CLHIP_POOL_DEAD = the window's maximum after ReLU is not positive.  max_pool2d backward routes the
// gradient to the arg-max position and ReLU backward then multiplies it by (output > 0) (VGGSlim.py:32,38): for a dead
// window that is 0 at every position, so the consumers' `code == position` tests implement BOTH backward operators and
// the backward-data kernel of the NEXT layer needs no (input > 0) mask (52 MB of reads per launch on layer 2 of the
// bench model, measured 21 us of 146).  clhip_maxpool2_fwd keeps producing plain arg-max bytes (no ReLU knowledge).
constexpr int CLHIP_POOL_DEAD = 4;
typedef unsigned int clhip_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t clhip_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes > 0x7fffffffull ? 0x7fffffff : (int)bytes, 0x00020000);
}
__device__ __forceinline__ float clhip_buf_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float4 clhip_buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const clhip_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ float2 clhip_buf_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ unsigned clhip_buf_load_u16(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0);
}
__device__ __forceinline__ unsigned clhip_buf_load_u8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0);
}
// Cache policy of the stores that carry a kernel's OUTPUT (activations, gradients, weight-gradient slabs) to the next kernel:
// sc1 = write-through at agent scope.  The XCD L2s are not coherent with each other, so a kernel ends by writing back whatever its
// stores left dirty in them; measured (tools/micro/store_policy.hip, profiles/r06c_store_policy.txt) a kernel that writes >= 16 MB
// ends 2 - 2.5 us earlier when its stores went through as they were issued (8 MB: ~1 us; `nt` alone: nothing).  Same bytes, same
// values.  CLHIP_ST_AUX=0 builds the default policy (aux bits of the buffer instructions: 1 = sc0, 2 = nt, 16 = sc1).
#ifndef CLHIP_ST_AUX
#define CLHIP_ST_AUX 16
#endif
__device__ __forceinline__ void clhip_buf_store(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, CLHIP_ST_AUX);
}
// (arg-max bytes keep the default policy: written through, their 1 .. 16-byte pieces reach memory one by one — the pooling forward of
// layer 2 wrote 19.0 MiB instead of 16.0, profiles/r06c_final_traffic.txt — and 3 MB of dirty lines cost nothing at the end of a kernel)
__device__ __forceinline__ void clhip_buf_store_u8(uint8_t v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b8(v, r, voff, soff, 0);
}
__device__ __forceinline__ void clhip_buf_store4(float4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    const clhip_u32x4 q = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, CLHIP_ST_AUX);
}
__device__ __forceinline__ void clhip_buf_store2(float2 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 q = {__float_as_uint(v.x), __float_as_uint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, CLHIP_ST_AUX);
}
// Output stores through a plain pointer with that policy: `base` must be wave-uniform (it becomes the buffer descriptor), `off` is
// the lane's ELEMENT offset from it (< 2^29 elements)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t clhip_out_rsrc(void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ size_t out_img_of(int C, int hw) { return (size_t)C * hw; }

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32 fragment maps (cdna_hip_programming.md §3):
//   A operand: lane l supplies A[i = l & 31][k = l >> 5]
//   B operand: lane l supplies B[k = l >> 5][j = l & 31]
//   D: reg r of lane l holds D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31]
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }


// fc_chain.hip: weight/bias gradients of the whole Linear stack in one launch. Offsets are float offsets:
// w_off / b_off into the parameter (and gradient) arena, act_off into the activation workspace, dz_off into the
// gradient scratch of the hidden layers.
#define CLHIP_FC_MAX 4
struct clhip_fc_chain {
    int n;
    long w_off[CLHIP_FC_MAX], b_off[CLHIP_FC_MAX];
    int din[CLHIP_FC_MAX], dout[CLHIP_FC_MAX], relu[CLHIP_FC_MAX];
    size_t act_off[CLHIP_FC_MAX], dz_off[CLHIP_FC_MAX];
};
int clhip_internal_fc_chain_ok(const clhip_fc_chain* d);
int clhip_internal_fc_chain_wgrad(const clhip_fc_chain* d, float* grads, const float* x, int N, const float* acts,
                                  const float* dlogits, const float* dz, hipStream_t s);
// gemm_body.hpp: arguments of one strided-GEMM launch (see gemm.hip)
struct clhip_gemm_args {
    const float* a; const float* b; float* out;
    int M, N, K; long sam, sak, sbk, sbn; int n_tiles, splits, k_per_split;
    const float* bias; const float* mask_src; int relu;
};
// backward-data GEMM of a Linear layer as arguments for a combined launch; returns its block count, 0 when the shape is split over K
int clhip_internal_fc_bwd_data_args(const float* dy, const float* w, const float* relu_src, float* dx, int M, int I, int O,
                                    clhip_gemm_args* args_out);
// fc_chain.hip: that GEMM and the weight / bias gradients of the whole classifier in ONE launch
int clhip_internal_fc_bwd_combo(const clhip_gemm_args* g, int gemm_blocks, const clhip_fc_chain* d, float* grads, const float* x,
                                int N, const float* acts, const float* dlogits, const float* dz, hipStream_t s);

// fused classifier tail (fc_chain.hip): layers 2..3 forward, cross-entropy, backward-data down to dz of h1
int clhip_internal_fc_tail_ok(const clhip_fc_chain* d);
int clhip_internal_fc_tail(const clhip_fc_chain* d, const float* params, float* acts, int N, const int64_t* labels,
                           int reduction, int col_off, int ncols, float* dlogits, float* fcdz, float* loss_out, double* stats,
                           void* row_scratch, unsigned* counter, int do_loss, int do_bwd, const float* h1_slabs, int live,
                           hipStream_t s);
int clhip_internal_fc_fwd_partial(const float* x, const float* w, int M, int I, int O, void* ws, size_t ws_bytes, int* live,
                                  hipStream_t s);

// conv3x3_wgrad.hip: deferred slab reduction (see clhip_internal_conv3x3_wgrad_partial)
#define CLHIP_WGRAD_JOBS_MAX 32
struct clhip_wgrad_job { const float* part; float* dw; float* db; int K, C, splits; int taps; };     // taps: 0 = 9 (3x3); 25 = 5x5 (bswgrad5.hip)
int clhip_internal_conv3x3_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N,
                                         int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream, clhip_wgrad_job* job);
int clhip_internal_wgrad_reduce_multi(const clhip_wgrad_job* jobs, int n, hipStream_t s);

// convkk.hip: K x K stride-1 convolution on small maps with the plane staged in LDS (AlexNet's 5x5 layer); mode 0 forward, 1 backward-data
bool clhip_internal_convkk_ok(int N, int Cin, int Cout, int H, int W, int R, int S, int stride, int pad);
int clhip_internal_convkk(int mode, const float* in, const float* w, const float* bias, const float* mask_src, float* out, int N,
                          int Cin, int Cout, int H, int W, int relu, hipStream_t s);

// wino.hip: 3x3 convolution by Winograd F(2x2, 3x3) (forward / backward-data); see the file's header
struct clhip_wino_wt { const float* w; float* U; int Ko, Ci, mode, pad; };
int clhip_internal_wino_weights(const clhip_wino_wt* jobs, int n, hipStream_t s);
int clhip_internal_weight_images(const clhip_wino_wt* wino_jobs, int n_wino, const clhip_wino_wt* bs_jobs, int n_bs, hipStream_t s);   // both kinds, one launch
int clhip_internal_wino_conv_u(int mode, const float* in, const float* U, const float* bias, const float* mask_src, float* out,
                               uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s);
bool clhip_internal_wino_ok(int Cin, int Cout, int H, int W);
size_t clhip_internal_wino_ws(int Cin, int Cout);
// bsconv.hip: the same operators on the bf16 matrix cores with fp32 operands split into three bf16 pieces (weight image in `wimg`)
bool clhip_internal_bs_ok(int Cin, int Cout, int H, int W);
bool clhip_internal_bs_preferred(int Cin, int Cout, int H, int W);
bool clhip_internal_bs5_preferred(int Cin, int Cout, int H, int W);
size_t clhip_internal_bs5_ws(int Cin, int Cout);
int clhip_internal_bs5_conv_u(int mode, const float* in, const void* wimg, const float* bias, const float* mask_src, float* out, int N,
                              int Cin, int Cout, int H, int W, int relu, hipStream_t s);
size_t clhip_internal_bs_ws(int Cin, int Cout);
int clhip_internal_bs_weights(const clhip_wino_wt* jobs, int n, hipStream_t s);
int clhip_internal_bs_conv_u(int mode, const float* in, const void* wimg, const float* bias, const float* mask_src, float* out,
                             uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s);
// bswgrad5.hip: 5x5 / padding-2 weight gradient on the bf16 matrix cores (AlexNet's second convolution): slabs + reduction
bool clhip_internal_bs5_wgrad_ok(int N, int C, int K, int H, int W);
size_t clhip_internal_bs5_wgrad_ws(int N, int C, int K, int H, int W);
int clhip_internal_bs5_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                             size_t ws_bytes, hipStream_t s);
size_t clhip_internal_bs3k_wgrad_ws(int N, int C, int K, int H, int W);         // the 3x3 layer on bswgrad5.hip's kernel (any map width >= 8)
int clhip_internal_bs3k_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                              size_t ws_bytes, hipStream_t s);
// bswgrad.hip: 3x3 weight gradient on the bf16 matrix cores (split fp32 operands, no LDS); slabs in conv3x3_wgrad.hip's format
bool clhip_internal_bs_wgrad_ok(int C, int K, int H, int W);
bool clhip_internal_bs_wgrad_preferred(int C, int K, int H, int W, int pooled);
size_t clhip_internal_bs_wgrad_ws(int N, int C, int K, int H, int W);
int clhip_internal_bs_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N, int C, int K,
                                    int H, int W, void* ws, size_t ws_bytes, hipStream_t s, clhip_wgrad_job* job);
bool clhip_internal_wino_wgrad_ok(int C, int K, int H, int W);
size_t clhip_internal_wino_wgrad_ws(int N, int C, int K, int H, int W);
int clhip_internal_wino_wgrad_partial(const float* x, const float* dy, const uint8_t* unpool_idx, float* dw, float* db, int N, int C,
                                      int K, int H, int W, void* ws, size_t ws_bytes, hipStream_t s, clhip_wgrad_job* job);
// backward-data + weight-gradient slabs of one 3x3 layer as ONE grid (wino.hip, wino_pair_kernel); CLHIP_ENOTSUP: two launches
int clhip_internal_wino_pair(const float* dy, const uint8_t* unpool_idx, const float* U, const float* mask_src, float* dx,
                             const float* x, float* dw, float* db, int N, int C, int K, int H, int W, void* ws, size_t ws_bytes,
                             hipStream_t s, clhip_wgrad_job* job);
bool clhip_internal_wino_pair_shape(int N, int C, int K, int H, int W, int pooled);
int clhip_internal_wino_conv(int mode, const float* in, const float* w, const float* bias, const float* mask_src, float* out,
                             uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, void* ws,
                             size_t ws_bytes, hipStream_t s);
