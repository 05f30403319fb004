// Weight / bias gradients of the whole classifier (nn.Linear stack of models/VGGSlim.py:68-74) in ONE launch.
//
// The per-layer path (gemm.hip) spends 3 x (GEMM + split-K reduction + bias column sum) = 9 launches per pass on
// ~27 MFLOP; at batch 200 each of them is launch-latency bound (62 us per pass in total).  Here every wave owns one
// 32x32 tile of some layer's dW = dz_l^T . h_{l-1} (reduction over the batch, two samples per MFMA), tiles of all
// layers share one grid, and the tiles of input-column 0 also produce db_l = colsum(dz_l): 13.6 us.  Operands are read
// with coalesced rows straight from global memory, 32 loads in flight per wave (named register batches: a
// runtime-indexed buffer would live in scratch).  Fixed summation order => bitwise run-to-run deterministic.
//
// (A fused forward / backward-data chain with activations kept in LDS was also tried: 134 us / 91 us against
// 25 us / 43 us for the GEMM launches, because its per-lane strided operand reads are latency-serialised; removed.)
#include "common.hpp"
#include "gemm_body.hpp"

namespace {

struct FcLayer { long w_off, b_off; int din, dout, relu; size_t act_off, dz_off; };
struct FcChain { FcLayer l[CLHIP_FC_MAX]; int n; };

// ---------------------------------------------------------------------------------------------- backward (weights)
// One wave per 32x32 tile of some layer's dW: dW[o][i] = sum_n dz[n][o] * h[n][i]; tiles with i-tile 0 also db[o].
struct FcTileMap { int first[CLHIP_FC_MAX + 1]; };     // prefix sums of tiles per layer

__device__ __forceinline__ void fc_chain_wgrad_block(const FcChain& c, const FcTileMap& tm, const float* __restrict__ x, int N,
                                                     const float* __restrict__ acts, const float* __restrict__ dlogits,
                                                     const float* __restrict__ dz, float* __restrict__ grads, int block) {
    const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
    const int t = block * 4 + (threadIdx.x >> 6);
    if (t >= tm.first[c.n]) return;
    int l = 0;
    while (t >= tm.first[l + 1]) ++l;
    const FcLayer L = c.l[l];
    const int nit = (L.din + 31) / 32;
    const int tt = t - tm.first[l], ot = tt / nit, it = tt - ot * nit;
    const float* dzl = l == c.n - 1 ? dlogits : dz + L.dz_off;
    const float* h = l ? acts + c.l[l - 1].act_off : x;
    const int o = ot * 32 + li, i = it * 32 + li;
    const bool ook = o < L.dout, iok = i < L.din;
    const float* ap = dzl + (ook ? o : 0);
    const float* bp = h + (iok ? i : 0);
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float asum = 0.f;
    const int npad = (N + 31) & ~31;
    float aA[16], bA[16], aB[16], bB[16];
    auto fetch = [&](int n0, float (&aq)[16], float (&bq)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int n = n0 + 2 * q + kk;
            const size_t nn = n < N ? n : 0;
            aq[q] = ap[nn * L.dout];
            bq[q] = bp[nn * L.din];
        }
    };
    auto compute = [&](int n0, const float (&aq)[16], const float (&bq)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const bool nok = n0 + 2 * q + kk < N;
            const float a = (nok && ook) ? aq[q] : 0.f;
            const float b = (nok && iok) ? bq[q] : 0.f;
            asum += a;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    };
    if (npad <= 256) {
        // the usual batch sizes: every operand load of the tile is in flight before the first MFMA (one memory latency
        // instead of one per 64 samples; 256 operand registers, the kernel runs one wave per SIMD anyway)
        float a8[8][16], b8[8][16];
#pragma unroll
        for (int bt = 0; bt < 8; ++bt)
            if (bt * 32 < npad) fetch(bt * 32, a8[bt], b8[bt]);
#pragma unroll
        for (int bt = 0; bt < 8; ++bt)
            if (bt * 32 < npad) compute(bt * 32, a8[bt], b8[bt]);
    } else {
        fetch(0, aA, bA);
        for (int n0 = 0; n0 < npad; n0 += 64) {
            const bool hasB = n0 + 32 < npad;
            if (hasB) fetch(n0 + 32, aB, bB);
            compute(n0, aA, bA);
            if (hasB) {
                if (n0 + 64 < npad) fetch(n0 + 64, aA, bA);
                compute(n0 + 32, aB, bB);
            }
        }
    }
    float* gw = grads + L.w_off;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ot * 32 + mfma32_row(r, lane);
        if (row < L.dout && iok) gw[(size_t)row * L.din + i] = acc[r];
    }
    if (it == 0) {
        const float tot = asum + __shfl_xor(asum, 32, 64);          // even samples + odd samples, fixed order
        if (kk == 0 && ook) grads[L.b_off + o] = tot;
    }
}

// ---------------------------------------------------------------------------------------------- classifier tail
// Everything between the first Linear layer's output h1 and the gradient w.r.t. h1, in ONE launch:
//     h2 = relu(h1 W2^T + b2), z = h2 W3^T + b3, softmax cross-entropy (loss, top-1 hits, dlogits),
//     dz2 = (dlogits W3) . [h2 > 0], dz1 = (dz2 W2) . [h1 > 0]
// (models/VGGSlim.py:68-74 with the 128-wide classifiers of small_VGG9_cl_128_128; train_EWC.py:179-186).  The samples
// of a batch are independent through all of it, so a workgroup owns 32 rows: W2, W3 and the row block of h1 are loaded
// once into LDS, the products run on the 32x32x2 MFMA with wave w owning output columns 32w..32w+31, activations and
// gradients go from one product to the next through LDS.  It replaces 2 forward GEMMs, the loss kernel and 2
// backward-data GEMMs (9.4 + 9.4 + 9.4 + 11.7 + 11.7 us of launch latency per pass at batch 200) by one ~10 us launch.
//
// Bit-compatibility with the per-layer path is part of the contract (the end-to-end fixtures are chaotic in the last
// bit): every product accumulates k in ascending pairs on the same MFMA instruction with the same zero padding to a
// multiple of 32 as gemm_mfma_kernel with one split, epilogues apply bias / ReLU / mask in the same order, the
// per-row loss code is the text of softmax_ce_rows_lds_kernel (loss.hip), and the loss / hit totals are formed by the
// last workgroup to finish in that kernel's order (64-row butterflies, then 16 sequential adds).
constexpr int FC_TAIL_ROWS_DEFAULT = 16;      // measured at N = 200 (rocprofv3, 40 launches): 32 rows 29.7 us, 16 rows 26.8, 8 rows 26.7
constexpr int TL = 129;            // LDS row stride in floats: odd, so walks along rows and along columns are conflict-free
constexpr int ZL = 33;
struct FcTail {
    long w2, b2, w3, b3, b1;       // float offsets into the parameter arena
    int d1, d2, d3, relu1, relu2, relu3;  // widths of h1, h2, logits
    size_t a1, a2, a3, dz1, dz2;   // float offsets: h1 / h2 / logits in the activation workspace, dz1 / dz2 in the fc gradient scratch
};

// acc = sum over k (ascending pairs) of A(k) * B(k), K a multiple of 32: operands of 16 MFMAs are fetched from LDS
// as a batch while the previous batch feeds the matrix pipe (named register batches: no runtime-indexed arrays)
template <class FA, class FB>
__device__ __forceinline__ floatx16 tail_mm(int K, FA a_at, FB b_at) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float aA[16], bA[16], aB[16], bB[16];
    auto fetch = [&](int k0, float (&aq)[16], float (&bq)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) { aq[q] = a_at(k0 + 2 * q); bq[q] = b_at(k0 + 2 * q); }
    };
    auto run = [&](const float (&aq)[16], const float (&bq)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q], bq[q], acc, 0, 0, 0);
    };
    fetch(0, aA, bA);
    for (int k0 = 0; k0 < K; k0 += 64) {
        const bool hasB = k0 + 32 < K;
        if (hasB) fetch(k0 + 32, aB, bB);
        run(aA, bA);
        if (hasB) {
            if (k0 + 64 < K) fetch(k0 + 64, aA, bA);
            run(aB, bB);
        }
    }
    return acc;
}

// RB = rows (samples) a workgroup owns: 32, 16 or 8.  The products always run on 32-row MFMA tiles (rows past RB are zero and
// are never stored), so a row's values do not depend on RB; fewer rows per workgroup = more workgroups on this latency-bound
// launch and all split-K slabs of a row block in flight at once (RB / 8 float4 per thread and slab).
template <int RB>
__global__ __launch_bounds__(256) void fc_tail_kernel(FcTail t, const float* __restrict__ params, const float* __restrict__ h1,
                                                      float* __restrict__ acts, int N, const int64_t* __restrict__ labels,
                                                      int reduction, int col_off, int C, float* __restrict__ dlogits,
                                                      float* __restrict__ fcdz, float* __restrict__ loss_out,
                                                      double* __restrict__ stats, float* row_loss, int* row_ok,
                                                      unsigned* counter, int do_loss, int do_bwd,
                                                      const float* __restrict__ slabs, int live) {
    __shared__ float W2s[128 * TL];
    __shared__ float W3s[32 * TL];
    __shared__ float h1s[32 * TL];
    __shared__ float h2s[32 * TL];
    __shared__ float dz2s[32 * TL];
    __shared__ float zs[32 * ZL];
    __shared__ float exs[32 * ZL];
    __shared__ float r_mx[32], r_lse[32];
    __shared__ int r_am[32], r_y[32];
    __shared__ float w_loss[16];
    __shared__ int w_corr[16];
    __shared__ unsigned s_ticket;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kk = lane >> 5;
    static_assert(RB == 32 || RB == 16 || RB == 8, "rows per workgroup");
    constexpr int JN = RB / 8;           // float4 of the h1 row block per thread (RB rows x 32 float4 / 256 threads at d1 = 128)
    constexpr int UN = 32 / JN;          // slabs per trip of the slab sum: 32 loads in flight per thread
    const int m0 = blockIdx.x * RB;
    const int K1 = (t.d1 + 31) & ~31, K2 = (t.d2 + 31) & ~31;       // padded widths of h1 / h2

    // ---- stage W2 [d2][d1], the h1 row block and W3 [d3][d2]: all loads in flight before the first LDS write;
    //      everything outside the real extents reads as zero (buffer range check), which is the GEMM kernel's padding
    const __amdgpu_buffer_rsrc_t r_w2 = clhip_rsrc(params + t.w2, (size_t)t.d2 * t.d1 * 4);
    const __amdgpu_buffer_rsrc_t r_w3 = clhip_rsrc(params + t.w3, (size_t)t.d3 * t.d2 * 4);
    const int rows_here = min(RB, N - m0);
    const __amdgpu_buffer_rsrc_t r_h1 = clhip_rsrc(h1 + (size_t)m0 * t.d1, (size_t)rows_here * t.d1 * 4);
    float4 q2[16], q1[4], q3[4];
    const int c1 = K1 >> 2, c2 = K2 >> 2;           // float4 columns
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int e = tid + 256 * j, o = e / c1, i = (e - o * c1) * 4;
        q2[j] = clhip_buf_load4(r_w2, (o < t.d2 && i < t.d1 && e < K2 * c1) ? (o * t.d1 + i) * 4 : CLHIP_OOB, 0);
    }
    if (!slabs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, r = e / c1, i = (e - r * c1) * 4;
            q1[j] = clhip_buf_load4(r_h1, (r < rows_here && i < t.d1 && e < 32 * c1) ? (r * t.d1 + i) * 4 : CLHIP_OOB, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, o = e / c2, i = (e - o * c2) * 4;
        q3[j] = clhip_buf_load4(r_w3, (o < t.d3 && i < t.d2 && e < 32 * c2) ? (o * t.d2 + i) * 4 : CLHIP_OOB, 0);
    }
    if (slabs) {
        // h1 = relu(sum of the first Linear layer's split-K slabs, ascending + b1): gemm_splitk_reduce_kernel's arithmetic on
        // this row block (slabs past `live` are all zero and adding them changes nothing); h1 also goes to the activation
        // workspace, where the weight-gradient launch and the per-layer path expect it
        const size_t slab = (size_t)N * t.d1;
        const __amdgpu_buffer_rsrc_t r_p = clhip_rsrc(slabs + (size_t)m0 * t.d1, ((size_t)(live - 1) * slab + (size_t)rows_here * t.d1) * 4);
        int off[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, r = e / c1, i = (e - r * c1) * 4;
            off[j] = (r < rows_here && i < t.d1 && e < 32 * c1) ? (r * t.d1 + i) * 4 : CLHIP_OOB;
            q1[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // (rows >= RB of the block: items j >= JN at d1 = 128, out of range by `off`; narrower layers keep the general loop)
        if (K1 == 128) {
            for (int s0 = 0; s0 < live; s0 += UN) {
                float4 pv[UN][JN];
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int j = 0; j < JN; ++j)
                        pv[u][j] = clhip_buf_load4(r_p, (s0 + u < live && off[j] != CLHIP_OOB) ? off[j] + (int)((size_t)(s0 + u) * slab * 4) : CLHIP_OOB, 0);
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int j = 0; j < JN; ++j) { q1[j].x += pv[u][j].x; q1[j].y += pv[u][j].y; q1[j].z += pv[u][j].z; q1[j].w += pv[u][j].w; }
            }
        } else {
            for (int s0 = 0; s0 < live; s0 += 8) {
                float4 pv[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pv[u][j] = clhip_buf_load4(r_p, (s0 + u < live && off[j] != CLHIP_OOB) ? off[j] + (int)((size_t)(s0 + u) * slab * 4) : CLHIP_OOB, 0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { q1[j].x += pv[u][j].x; q1[j].y += pv[u][j].y; q1[j].z += pv[u][j].z; q1[j].w += pv[u][j].w; }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j, r = e / c1, i = (e - r * c1) * 4;
            if (off[j] != CLHIP_OOB) {
                const float4 b = *reinterpret_cast<const float4*>(params + t.b1 + i);
                float4 v = q1[j];
                v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
                if (t.relu1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                q1[j] = v;
                *reinterpret_cast<float4*>(acts + t.a1 + (size_t)(m0 + r) * t.d1 + i) = v;
            }
        }
    }
    auto put4 = [](float* d, const float4& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, r = e / c1, i = (e - r * c1) * 4;
        if (e < 32 * c1) put4(h1s + r * TL + i, q1[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = tid + 256 * j, o = e / c2, i = (e - o * c2) * 4;
        if (e < 32 * c2) put4(W3s + o * TL + i, q3[j]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int e = tid + 256 * j, o = e / c1, i = (e - o * c1) * 4;
        if (e < K2 * c1) put4(W2s + o * TL + i, q2[j]);
    }
    __syncthreads();

#if defined(CLHIP_TAIL_STOP) && CLHIP_TAIL_STOP == 1
    return;
#endif
    const int n0 = wave * 32, n = n0 + li;
    floatx16 acc;
    // ---- h2 = relu(h1 . W2^T + b2): wave w owns columns 32w .. 32w+31 of h2
    if (n0 < K2) {
        acc = tail_mm(K1, [&](int k) { return h1s[li * TL + k + kk]; }, [&](int k) { return W2s[n * TL + k + kk]; });
        const float bias = n < t.d2 ? params[t.b2 + n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, lane), m = m0 + row;
            float v = acc[r];
            v += bias;
            if (t.relu2) v = fmaxf(v, 0.f);
            if (n >= t.d2) v = 0.f;
            h2s[row * TL + n] = v;
            if (row < rows_here && n < t.d2) acts[t.a2 + (size_t)m * t.d2 + n] = v;
        }
    }
#if defined(CLHIP_TAIL_STOP) && CLHIP_TAIL_STOP == 2
    return;
#endif
    __syncthreads();
    // ---- logits = h2 . W3^T + b3 (one column tile: d3 <= 32)
    if (wave == 0) {
        acc = tail_mm(K2, [&](int k) { return h2s[li * TL + k + kk]; }, [&](int k) { return W3s[li * TL + k + kk]; });
        const float bias = li < t.d3 ? params[t.b3 + li] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma32_row(r, lane), m = m0 + row;
            float v = acc[r];
            v += bias;
            if (t.relu3) v = fmaxf(v, 0.f);
            if (li >= t.d3) v = 0.f;
            zs[row * ZL + li] = v;
            if (row < rows_here && li < t.d3) acts[t.a3 + (size_t)m * t.d3 + li] = v;
        }
    }
#if defined(CLHIP_TAIL_STOP) && CLHIP_TAIL_STOP == 3
    return;
#endif
    if (!do_loss) return;
    __syncthreads();
    // ---- softmax cross-entropy per row: the per-row text of softmax_ce_rows_lds_kernel (loss.hip) — keep in sync
    const int ld = t.d3;
    const float scale = reduction == 0 ? 1.f / (float)N : 1.f;
    // (spread over the workgroup: the exponentials of a row are evaluated by different lanes, every sum still runs over
    //  the classes in ascending order in one lane, so each value is the one the row-per-thread text produces)
    if (tid < 32) {
        float mx = -INFINITY;
        int am = 0;
        if (tid < rows_here) {
            const float* z = zs + tid * ZL + col_off;
            for (int c = 0; c < C; ++c) {
                float v = z[c];
                if (v > mx) { mx = v; am = c; }
            }
        }
        r_mx[tid] = mx; r_am[tid] = am;
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
        const int row = e >> 5, c = e & 31;
        if (c < C) exs[row * ZL + c] = expf(zs[row * ZL + col_off + c] - r_mx[row]);
    }
    __syncthreads();
    if (tid < 32) {
        const int m = m0 + tid;
        if (tid < rows_here) {
            const float* z = zs + tid * ZL + col_off;
            const int y = (int)labels[m];
            const float mx = r_mx[tid];
            float se = 0.f;
            for (int c = 0; c < C; ++c) se += exs[tid * ZL + c];
            const float lse = logf(se);
            const float lrow = -(z[y] - mx - lse);
            r_lse[tid] = lse; r_y[tid] = y;
            row_loss[m] = lrow;
            row_ok[m] = (r_am[tid] == y);
        }
    }
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += 256) {
        const int row = e >> 5, c = e & 31, cc = c - col_off;
        float v = 0.f;
        if (row < rows_here && c < ld && cc >= 0 && cc < C)
            v = (expf(zs[row * ZL + c] - r_mx[row] - r_lse[row]) - (cc == r_y[row] ? 1.f : 0.f)) * scale;
        zs[row * ZL + c] = v;
    }
    __syncthreads();
    for (int e = tid; e < 32 * ld; e += 256) {
        const int row = e / ld, c = e - row * ld;
        if (row < rows_here) dlogits[(size_t)(m0 + row) * ld + c] = zs[row * ZL + c];
    }
#if defined(CLHIP_TAIL_STOP) && CLHIP_TAIL_STOP == 4
    return;
#endif
    if (do_bwd) {
        // ---- dz2 = (dlogits . W3) masked by h2 > 0: K = d3 padded to 32 (zs columns >= d3 and W3s rows >= d3 are zero)
        if (n0 < K2) {
            acc = tail_mm(32, [&](int k) { return zs[li * ZL + k + kk]; }, [&](int k) { return W3s[(k + kk) * TL + n]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, lane), m = m0 + row;
                float v = h2s[row * TL + n] > 0.f ? acc[r] : 0.f;
                if (n >= t.d2) v = 0.f;
                dz2s[row * TL + n] = v;
                if (row < rows_here && n < t.d2) fcdz[t.dz2 + (size_t)m * t.d2 + n] = v;
            }
        }
        __syncthreads();
        // ---- dz1 = (dz2 . W2) masked by h1 > 0
        if (n0 < K1) {
            acc = tail_mm(K2, [&](int k) { return dz2s[li * TL + k + kk]; }, [&](int k) { return W2s[(k + kk) * TL + n]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma32_row(r, lane), m = m0 + row;
                const float v = h1s[row * TL + n] > 0.f ? acc[r] : 0.f;
                if (row < rows_here && n < t.d1) fcdz[t.dz1 + (size_t)m * t.d1 + n] = v;
            }
        }
    }
#if defined(CLHIP_TAIL_STOP) && CLHIP_TAIL_STOP == 5
    return;
#endif
    // ---- totals: the last workgroup to arrive sums the per-row values in softmax_ce_rows_lds_kernel's order
    __threadfence();
    __syncthreads();
    if (tid == 0) s_ticket = atomicAdd(counter, 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();
    for (int g = wave; g < 16; g += 4) {
        const int row = g * 64 + lane;
        // agent-scope loads: the other workgroups' values come from L2, not from this CU's vector cache
        float v = row < N ? __hip_atomic_load(row_loss + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
        int cc = row < N ? __hip_atomic_load(row_ok + row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o, 64); cc += __shfl_xor(cc, o, 64); }
        if (lane == 0) { w_loss[g] = v; w_corr[g] = cc; }
    }
    __syncthreads();
    if (tid == 0) {
        float tt = 0.f; int c = 0;
        for (int w = 0; w < 16; ++w) { tt += w_loss[w]; c += w_corr[w]; }
        tt *= scale;
        loss_out[0] = tt;
        if (stats) { stats[0] += (double)tt; stats[1] += (double)c; }
        *counter = 0u;
    }
}

__global__ __launch_bounds__(256) void fc_chain_wgrad_kernel(FcChain c, FcTileMap tm, const float* __restrict__ x, int N,
                                                             const float* __restrict__ acts,
                                                             const float* __restrict__ dlogits, const float* __restrict__ dz,
                                                             float* __restrict__ grads) {
    fc_chain_wgrad_block(c, tm, x, N, acts, dlogits, dz, grads, blockIdx.x);
}

// Backward of the first Linear layer in one launch: blocks [0, gemm_blocks) are the 64x64 tiles of its backward-data GEMM
// (dx = dz1 . W1, masked by the layer input), the rest are the weight / bias gradient tiles of the whole classifier.  Both
// only read dz and the activations, both are latency-bound at batch 200 (11.7 + 13.7 us as two launches) and together
// they still fit the chip's 256 CUs.  Same device functions as the two stand-alone kernels => same bits.
__global__ __launch_bounds__(256) void fc_bwd_combo_kernel(clhip_gemm_args g, int gemm_blocks, FcChain c, FcTileMap tm,
                                                           const float* __restrict__ x, int N, const float* __restrict__ acts,
                                                           const float* __restrict__ dlogits, const float* __restrict__ dz,
                                                           float* __restrict__ grads) {
    if ((int)blockIdx.x < gemm_blocks) gemm_tile<true, false>(g, blockIdx.x);
    else fc_chain_wgrad_block(c, tm, x, N, acts, dlogits, dz, grads, (int)blockIdx.x - gemm_blocks);
}

FcChain to_chain(const clhip_fc_chain* d) {
    FcChain c;
    c.n = d->n;
    for (int l = 0; l < d->n; ++l)
        c.l[l] = FcLayer{d->w_off[l], d->b_off[l], d->din[l], d->dout[l], d->relu[l], d->act_off[l], d->dz_off[l]};
    return c;
}

}  // namespace

// Envelope of the fused weight-gradient launch (the caller falls back to the per-layer GEMMs otherwise).
int clhip_internal_fc_chain_ok(const clhip_fc_chain* d) {
    if (!d || d->n < 1 || d->n > CLHIP_FC_MAX) return 0;
    long tiles = 0;
    for (int l = 0; l < d->n; ++l) {
        if (d->din[l] <= 0 || d->dout[l] <= 0) return 0;
        if (l > 0 && d->din[l] != d->dout[l - 1]) return 0;
        tiles += (long)((d->dout[l] + 31) / 32) * ((d->din[l] + 31) / 32);
    }
    return tiles <= 4096;       // one wave per 32x32 tile re-reads its operand rows; 64x64 GEMM tiles suit wide layers better
}

int clhip_internal_fc_chain_wgrad(const clhip_fc_chain* d, float* grads, const float* x, int N, const float* acts,
                                  const float* dlogits, const float* dz, hipStream_t s) {
    const FcChain c = to_chain(d);
    FcTileMap tm;
    tm.first[0] = 0;
    for (int l = 0; l < c.n; ++l) tm.first[l + 1] = tm.first[l] + ((c.l[l].dout + 31) / 32) * ((c.l[l].din + 31) / 32);
    for (int l = c.n; l < CLHIP_FC_MAX; ++l) tm.first[l + 1] = tm.first[c.n];
    hipLaunchKernelGGL(fc_chain_wgrad_kernel, dim3((tm.first[c.n] + 3) / 4), dim3(256), 0, s, c, tm, x, N, acts, dlogits, dz, grads);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int clhip_internal_fc_bwd_combo(const clhip_gemm_args* g, int gemm_blocks, const clhip_fc_chain* d, float* grads, const float* x,
                                int N, const float* acts, const float* dlogits, const float* dz, hipStream_t s) {
    const FcChain c = to_chain(d);
    FcTileMap tm;
    tm.first[0] = 0;
    for (int l = 0; l < c.n; ++l) tm.first[l + 1] = tm.first[l] + ((c.l[l].dout + 31) / 32) * ((c.l[l].din + 31) / 32);
    for (int l = c.n; l < CLHIP_FC_MAX; ++l) tm.first[l + 1] = tm.first[c.n];
    hipLaunchKernelGGL(fc_bwd_combo_kernel, dim3(gemm_blocks + (tm.first[c.n] + 3) / 4), dim3(256), 0, s, *g, gemm_blocks, c, tm, x, N,
                       acts, dlogits, dz, grads);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

// Envelope of the fused tail: Linear-ReLU-Linear-(ReLU)-Linear with both hidden widths <= 128 (multiples of 4) and at most
// 32 logits; anything else runs the per-layer launches.
int clhip_internal_fc_tail_ok(const clhip_fc_chain* d) {
    if (!d || d->n != 3) return 0;
    if (d->din[1] != d->dout[0] || d->din[2] != d->dout[1]) return 0;
    if (d->din[1] > 128 || d->dout[1] > 128 || d->dout[2] > 32) return 0;
    if ((d->din[1] & 3) || (d->dout[1] & 3)) return 0;
    if ((d->w_off[1] & 3) || (d->w_off[2] & 3) || (d->act_off[0] & 3) || (d->b_off[0] & 3)) return 0;
    return 1;
}

int clhip_internal_fc_tail(const clhip_fc_chain* d, const float* params, float* acts, int N, const int64_t* labels,
                           int reduction, int col_off, int ncols, float* dlogits, float* fcdz, float* loss_out, double* stats,
                           void* row_scratch, unsigned* counter, int do_loss, int do_bwd, const float* h1_slabs, int live,
                           hipStream_t s) {
    FcTail t;
    t.w2 = d->w_off[1]; t.b2 = d->b_off[1]; t.w3 = d->w_off[2]; t.b3 = d->b_off[2]; t.b1 = d->b_off[0];
    t.d1 = d->din[1]; t.d2 = d->dout[1]; t.d3 = d->dout[2]; t.relu1 = d->relu[0]; t.relu2 = d->relu[1]; t.relu3 = d->relu[2];
    t.a1 = d->act_off[0]; t.a2 = d->act_off[1]; t.a3 = d->act_off[2]; t.dz1 = d->dz_off[0]; t.dz2 = d->dz_off[1];
    if (live <= 0) h1_slabs = nullptr;
    float* row_loss = static_cast<float*>(row_scratch);
    int* row_ok = reinterpret_cast<int*>(row_loss + N);
    // rows per workgroup (CLHIP_FC_TAIL_ROWS = 32 | 16 | 8; results do not depend on it, tests/test_gpu_fc_tail.py)
    static const int rows_env = [] { const char* e = getenv("CLHIP_FC_TAIL_ROWS"); return e && e[0] ? atoi(e) : 0; }();
    const int rb = rows_env == 32 || rows_env == 16 || rows_env == 8 ? rows_env : FC_TAIL_ROWS_DEFAULT;
#define FC_TAIL_GO(RB)                                                                                                              \
    hipLaunchKernelGGL(fc_tail_kernel<RB>, dim3((N + RB - 1) / RB), dim3(256), 0, s, t, params, acts + d->act_off[0], acts, N, labels, \
                       reduction, col_off, ncols, dlogits, fcdz, loss_out, stats, row_loss, row_ok, counter, do_loss, do_bwd,       \
                       h1_slabs, live)
    if (rb == 8) FC_TAIL_GO(8);
    else if (rb == 16) FC_TAIL_GO(16);
    else FC_TAIL_GO(32);
#undef FC_TAIL_GO
    CLHIP_LAUNCH_CHECK();
    return 0;
}
