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

namespace {

struct FcLayer { long w_off, b_off; int din, dout, relu; size_t act_off, dz_off; };
struct FcChain { FcLayer l[CLHIP_FC_MAX]; int n; };

// ---------------------------------------------------------------------------------------------- backward (weights)
// One wave per 32x32 tile of some layer's dW: dW[o][i] = sum_n dz[n][o] * h[n][i]; tiles with i-tile 0 also db[o].
struct FcTileMap { int first[CLHIP_FC_MAX + 1]; };     // prefix sums of tiles per layer

__global__ __launch_bounds__(256) void fc_chain_wgrad_kernel(FcChain c, FcTileMap tm, const float* __restrict__ x, int N,
                                                             const float* __restrict__ acts,
                                                             const float* __restrict__ dlogits, const float* __restrict__ dz,
                                                             float* __restrict__ grads) {
    const int lane = threadIdx.x & 63, li = lane & 31, kk = lane >> 5;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
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
