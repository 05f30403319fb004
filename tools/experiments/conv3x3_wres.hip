// EXPERIMENT, NOT BUILT INTO libclhip: measured slower than the chunked kernel once that kernel got buffer-based
// staging / epilogue (L2 forward 195 us vs 160 us, MI355X, round 1). Kept for the record of what was tried:
// LDS-resident weights, wave-private activation slabs, no per-chunk barrier. Ablation (tools/gpu_abl.sh): pure
// MFMA+LDS loop 113-121 TFLOP/s; its scalar global staging (+13 %) and store epilogue (+14 %) were never optimised.
//
// 3x3 pad-1 convolution, "weights resident" variant for layers whose whole 32-output-channel weight slice fits
// in LDS next to the activation slabs (C_in <= 64): forward and backward-data of the 64-channel VGG layers
// (models/VGGSlim.py:19-21 'small_VGG9' layers 2-5, first layers of base / wide).
//
// Why a second kernel (measured on MI355X, profiles/README.md): in the chunked kernel of conv3x3.hip every
// 128-pixel block re-stages its 64 x C x 9 weight slice (3x the bytes of the activations it stages) and its four
// waves meet at a barrier per 8-channel chunk; blocks live ~2 launch rounds, so prologues, epilogues and tails of
// all co-resident blocks coincide.  MFMA and VALU do not co-issue across waves of a SIMD on this part
// (tools/micro/coexec.hip: both = sum), so every staging / index instruction is paid in matrix-pipe time.
//
// Here: one persistent block per CU, 8 waves.  The block loads its 32 x (C*9) weight slice ONCE into LDS
// ([k][33], the MFMA A operand reads it conflict-free); after that every wave is an independent worker — it walks
// its own sequence of 64-pixel tiles, stages the halo of ITS tile per 8-channel chunk into a private double-buffered
// slab (register-staged prefetch one chunk ahead, across tile boundaries), and never meets a barrier again.
//   per wave and chunk: 17 global loads, 17 ds_write, 108 ds_read, 72 MFMA (32 out-channels x 64 pixels x 8 x 9).
// Tiles are dealt round-robin to SIMDs (both waves of a SIMD alternate), so the per-SIMD imbalance is < 1 tile.
#include "common.hpp"
#include <cstdlib>

namespace {

constexpr int WR_KT = 32;        // output channels per block
constexpr int WR_LDW = 33;       // weight row stride
constexpr int WR_CK = 8;         // channels per chunk
constexpr int WR_WAVES = 8;

template <int TW>
struct WGeo {
    static constexpr int TH = 64 / TW;               // rows of a wave tile
    static constexpr int TWP = TW + 2;
    static constexpr int PLANE = (TH + 2) * TWP;     // halo plane of one channel
    static constexpr int SLAB = WR_CK * PLANE;       // floats per wave and buffer
    static constexpr int S_IT = (SLAB + 63) / 64;    // staged elements per lane and chunk
};

// MODE 0: forward  (in = x [N][Cin][H][W], wt [Cout][Cin][3][3], out = y; bias / relu / optional fused 2x2 pool)
// MODE 1: backward-data (in = dy [N][Cin=K][H][W], wt [K][C][3][3], out = dx [N][Cout=C][H][W], taps flipped,
//         optional ReLU mask (mask_src > 0))
template <int TW, int MODE>
__global__ __launch_bounds__(512, 1) void conv3x3_wres_kernel(
    const float* __restrict__ in, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ mask_src, float* __restrict__ out, uint8_t* __restrict__ pool_idx,
    int N, int Cin, int Cout, int H, int relu, int tiles_per_img, int ntiles) {
    using G = WGeo<TW>;
    constexpr int W = TW;                             // the wave tile spans the image width
    extern __shared__ float smem[];
    const int kdim = Cin * 9;
    float* wsm = smem;                                // [kdim][33]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, kk = lane >> 5;
    float* slab = smem + kdim * WR_LDW + wave * (2 * G::SLAB);
    const int ko0 = blockIdx.y * WR_KT;
    const bool pool = MODE == 0 && pool_idx != nullptr;

    // ---- resident weights: A[k = c*9 + tap][ch]
    if (MODE == 0) {
        // rows ch: kdim contiguous floats
        const int nvec = kdim / 4;                    // Cin % 8 == 0 => kdim % 4 == 0
        for (int e = tid; e < WR_KT * nvec; e += 512) {
            const int ch = e / nvec, f = e - ch * nvec;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ko0 + ch < Cout) v = *reinterpret_cast<const float4*>(wt + (size_t)(ko0 + ch) * kdim + 4 * f);
            float* d = wsm + (4 * f) * WR_LDW + ch;
            d[0] = v.x; d[WR_LDW] = v.y; d[2 * WR_LDW] = v.z; d[3 * WR_LDW] = v.w;
        }
    } else {
        // wt[k_in][c][tap], this block's 32 c's: 288 contiguous floats per k_in; A[k_in*9 + (8 - tap)][c]
        const int nch = min(WR_KT, Cout - ko0);
        for (int e = tid; e < Cin * WR_KT * 9; e += 512) {
            const int kin = e / (WR_KT * 9), rem = e - kin * (WR_KT * 9);
            const int c = rem / 9, tap = rem - 9 * c;
            const float v = c < nch ? wt[((size_t)kin * Cout + ko0 + c) * 9 + tap] : 0.f;
            wsm[(kin * 9 + 8 - tap) * WR_LDW + c] = v;
        }
    }

    // ---- per-lane constants
    // subtile t (0/1), lane li -> pixel (row = 2*(li / TW) + t, col = li % TW): the 2x2 pool partner below is the
    // same lane of the other accumulator, the partner to the right is lane^1.
    const int prow = 2 * (li / TW), pcol = li % TW;
    const int pixoff = prow * G::TWP + pcol;                      // + t*TWP for subtile 1
    const int a_lane = kk * 9 * WR_LDW + li;
    const int b_lane = kk * G::PLANE + pixoff;

    // staging map: element e = lane + 64*i of the [8][TH+2][TW+2] slab; offsets relative to (chunk base, h0 - 1, -1)
    int s_off[G::S_IT];
    unsigned m_all = 0, m_top = 0, m_bot = 0;                      // bit i: element valid / in halo row 0 / in last halo row
    const int plane_hw = H * W;
#pragma unroll
    for (int i = 0; i < G::S_IT; ++i) {
        const int e = lane + 64 * i;
        const int c = e / G::PLANE, rem = e - c * G::PLANE;
        const int row = rem / G::TWP, col = rem - row * G::TWP;
        s_off[i] = c * plane_hw + (row - 1) * W + col - 1;
        if (e < G::SLAB && col >= 1 && col <= TW) {
            m_all |= 1u << i;
            if (row == 0) m_top |= 1u << i;
            if (row == G::TH + 1) m_bot |= 1u << i;
        }
    }
    float sv[G::S_IT];

    // ---- this wave's tile sequence: SIMD q = 4*block + (wave & 3) owns tiles q, q + nq, ...; its two waves alternate
    const int nq = gridDim.x * 4;
    const int q = blockIdx.x * 4 + (wave & 3);
    const int tstride = 2 * nq;
    int tile = q + nq * (wave >> 2);
    const int n_chunks = Cin / WR_CK;

    auto tile_origin = [&](int t, int& n, int& h0) {
        n = t / tiles_per_img;
        h0 = (t - n * tiles_per_img) * G::TH;
    };
    auto load_chunk = [&](int t, int chunk) {
        int n, h0;
        tile_origin(t, n, h0);
        unsigned m = m_all;
        if (h0 == 0) m &= ~m_top;
        if (h0 + G::TH >= H) m &= ~m_bot;
        const float* xb = in + ((size_t)n * Cin + chunk * WR_CK) * plane_hw + (size_t)h0 * W;
#pragma unroll
        for (int i = 0; i < G::S_IT; ++i) {
            const bool ok = (m >> i) & 1u;
            const float* p = xb + s_off[i];
            sv[i] = *(ok ? p : clhip_zero16);
        }
    };
    auto store_chunk = [&](int buf) {
        float* d = slab + buf * G::SLAB + lane;
#pragma unroll
        for (int i = 0; i < G::S_IT; ++i)
            if (lane + 64 * i < G::SLAB) d[64 * i] = sv[i];
    };

    __syncthreads();                                  // weights visible to every wave; the only barrier
    if (tile < ntiles) { load_chunk(tile, 0); store_chunk(0); }
    int buf = 0;
    const size_t out_img = (size_t)Cout * plane_hw;
    for (; tile < ntiles; tile += tstride) {
        floatx16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (int chunk = 0; chunk < n_chunks; ++chunk, buf ^= 1) {
            // prefetch the next work item of this wave's stream (next chunk, or chunk 0 of the next tile)
            const bool last = chunk + 1 == n_chunks;
            const int nt = last ? tile + tstride : tile, nc = last ? 0 : chunk + 1;
            const bool more = nt < ntiles;
#ifndef WRES_ABL_NOLOAD
            if (more) load_chunk(nt, nc);
#endif

            const float* ws = wsm + chunk * (WR_CK * 9 * WR_LDW) + a_lane;
            const float* xs = slab + buf * G::SLAB + b_lane;
            float af[2][9], bf[2][2][9];
            auto load_frag = [&](int cp, int slot) {
#pragma unroll
                for (int rs = 0; rs < 9; ++rs) {
                    const int r = rs / 3, s = rs - 3 * (rs / 3);
                    af[slot][rs] = ws[((2 * cp) * 9 + rs) * WR_LDW];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        bf[slot][t][rs] = xs[(2 * cp) * G::PLANE + (t + r) * G::TWP + s];
                }
            };
            load_frag(0, 0);
#pragma unroll
            for (int cp = 0; cp < WR_CK / 2; ++cp) {
                if (cp + 1 < WR_CK / 2) load_frag(cp + 1, (cp + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rs = 0; rs < 9; ++rs)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cp & 1][rs], bf[cp & 1][t][rs], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef WRES_ABL_NOSTORE
            if (more) store_chunk(buf ^ 1);          // private slab: LDS ops of one wave are ordered, no barrier
#endif
        }

#ifdef WRES_ABL_NOEPI
        { float sacc = 0.f; for (int r = 0; r < 16; ++r) sacc += acc[0][r] + acc[1][r]; if (sacc != 1.2345e30f) continue; }
#endif
        // ---- epilogue of this tile
        int n, h0;
        tile_origin(tile, n, h0);
        if (pool) {
            // fused ReLU + 2x2/2 max-pool (VGGSlim.py:32,38): candidates {(h,w),(h,w+1),(h+1,w),(h+1,w+1)} =
            // {acc[0] lane, lane^1, acc[1] lane, lane^1}; first maximum in that order wins (ATen)
            const int OH = H >> 1, OW = W >> 1;
            const bool writer = !(li & 1);
            const size_t obase = (size_t)n * Cout * OH * OW + (size_t)((h0 + prow) >> 1) * OW + (pcol >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ko = ko0 + mfma32_row(r, lane);
                const float bv = (bias && ko < Cout) ? bias[ko] : 0.f;
                const float tl = fmaxf(acc[0][r] + bv, 0.f), bl = fmaxf(acc[1][r] + bv, 0.f);
                const float tr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tl), 0xB1, 0xF, 0xF, true));
                const float br = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(bl), 0xB1, 0xF, 0xF, true));
                float m = tl; int am = 0;
                if (tr > m) { m = tr; am = 1; }
                if (bl > m) { m = bl; am = 2; }
                if (br > m) { m = br; am = 3; }
                if (writer && ko < Cout) {
                    const size_t o = obase + (size_t)ko * OH * OW;
                    out[o] = m;
                    pool_idx[o] = (uint8_t)am;
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const size_t pbase = (size_t)n * out_img + (size_t)(h0 + prow + t) * W + pcol;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ko = ko0 + mfma32_row(r, lane);
                    if (ko < Cout) {
                        const size_t o = pbase + (size_t)ko * plane_hw;
                        float v = acc[t][r];
                        if (MODE == 0) {
                            if (bias) v += bias[ko];
                            if (relu) v = fmaxf(v, 0.f);
                        } else {
                            if (mask_src) v = mask_src[o] > 0.f ? v : 0.f;
                        }
                        out[o] = v;
                    }
                }
            }
        }
    }
}

template <int TW, int MODE>
int launch_wres(const float* in, const float* wt, const float* bias, const float* mask_src, float* out, uint8_t* pool_idx,
                int N, int Cin, int Cout, int H, int relu, hipStream_t s) {
    using G = WGeo<TW>;
    const size_t lds = ((size_t)Cin * 9 * WR_LDW + (size_t)WR_WAVES * 2 * G::SLAB) * sizeof(float);
    static bool configured = false;                   // per template instance
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wres_kernel<TW, MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int tiles_per_img = H / G::TH;
    const long long ntiles = (long long)N * tiles_per_img;
    const int groups = (Cout + WR_KT - 1) / WR_KT;
    int gx = 256 / groups;                            // one persistent block per CU over all channel groups
    if (gx < 1) gx = 1;
    const long long need = (ntiles + 7) / 8;          // no more blocks than there are tiles for their waves
    if (gx > need) gx = (int)need;
    hipLaunchKernelGGL((conv3x3_wres_kernel<TW, MODE>), dim3(gx, groups), dim3(512), lds, s,
                       in, wt, bias, mask_src, out, pool_idx, N, Cin, Cout, H, relu, tiles_per_img, (int)ntiles);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// Returns CLHIP_ENOTSUP when the shape is outside this kernel's envelope (the caller then takes the chunked kernel).
// mode 0: forward (Cin = C, Cout = K); mode 1: backward-data (Cin = K, Cout = C, wt stays [K][C][3][3]).
int clhip_internal_conv3x3_wres(int mode, const float* in, const float* wt, const float* bias, const float* mask_src,
                                float* out, uint8_t* pool_idx, int N, int Cin, int Cout, int H, int W, int relu,
                                hipStream_t s) {
    static const bool disabled = getenv("CLHIP_NO_WRES") != nullptr;      // tuning aid (tools/conv_bench.py A/B)
    if (disabled) return CLHIP_ENOTSUP;
    if (Cin % WR_CK != 0 || Cin > 64 || Cin < 16 || (W != 32 && W != 16 && W != 8)) return CLHIP_ENOTSUP;
    if (H % (64 / W) != 0 || (long long)N * H * W < 64 * 64) return CLHIP_ENOTSUP;          // a persistent 8-wave block needs work
    if (!aligned16(wt)) return CLHIP_ENOTSUP;
    if (pool_idx && (H & 1)) return CLHIP_ENOTSUP;
    if ((long long)N * (Cin > Cout ? Cin : Cout) * H * W > 0x7fffffffLL) return CLHIP_ENOTSUP;  // 32-bit staging offsets
#define WRES(TW_) (mode == 0 ? launch_wres<TW_, 0>(in, wt, bias, mask_src, out, pool_idx, N, Cin, Cout, H, relu, s) \
                             : launch_wres<TW_, 1>(in, wt, bias, mask_src, out, pool_idx, N, Cin, Cout, H, relu, s))
    if (W == 32) return WRES(32);
    if (W == 16) return WRES(16);
    return WRES(8);
#undef WRES
}
