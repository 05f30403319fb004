// K x K convolution (odd K, stride 1, pad K / 2) on small maps — AlexNet's 5x5 layer (torchvision alexnet features[3] as
// pickled by models/net.py:96-125: 64 -> 192 channels on 27 x 27 maps) — forward and backward-data, direct convolution on
// v_mfma_f32_32x32x2_f32 with the input plane staged ONCE per channel chunk (with its halo) in LDS and every tap read from
// there.  The gather-GEMM of conv2d.hip pays 10-18 VALU instructions of address arithmetic per gathered element in the
// pipe the fp32 MFMAs issue from (measured on this layer: MFMA phase alone 486 us, gathers alone 405 us, together 616 us
// forward / 770 us backward-data at N = 128); here a tap is a constant offset from the lane's window origin.
//
//   forward        y[n][k][h][w]  = b[k] + sum_{c,r,s} w[k][c][r][s] x[n][c][h + r - P][w + s - P]
//   backward-data  dx[n][c][h][w] = sum_{k,r,s} w[k][c][K-1-r][K-1-s] dy[n][k][h + r - P][w + s - P]   (* (x > 0))
//                  = the same kernel with the channel roles swapped and the taps reversed while the weights are staged.
//
// Block = 64 out-channels x 384 consecutive pixels of one image plane (pixels are numbered linearly over the plane, so a
// 27 x 27 plane is 23 32-pixel subtiles with 7 idle slots instead of 4 x 4 tiles of 8 x 8 with 295); 4 waves, every wave
// holds BOTH 32-channel row tiles of 3 subtiles (6 accumulators of 16 registers): a 16-byte LDS read of the weight tile
// feeds 4 taps x 3 subtiles, an x read feeds 2 row tiles, 0.35 LDS instructions per MFMA.  In-channels go through LDS in
// chunks of 4 (2 MFMA k-pairs), double-buffered, global loads one chunk ahead in registers; 76 KB of LDS per block and
// < 256 registers per lane, so two blocks share a CU and cover each other's barriers and epilogues.
// Summation order of an output element: in-channel pairs ascending, taps ascending inside a pair — fixed, run to run.
#include "common.hpp"

namespace {

constexpr int QKT = 64;          // out channels per block
constexpr int QCK = 4;           // in channels per chunk
constexpr int QPW = 32;          // LDS halo row length: W + 2 * pad <= 32
constexpr int QROWS = 20;        // LDS halo rows per block
constexpr int QSUB = 3;          // 32-pixel subtiles per wave
constexpr int QPIX = 4 * QSUB * 32;      // pixels per block
constexpr int QPLANE = QROWS * QPW;

template <int KS>
struct QGeo {
    static constexpr int T = KS * KS, PAD = KS / 2;
    static constexpr int TP = (T + 3) / 4 * 4;                 // taps per (channel, out-channel) in LDS; TP / 4 odd => b128 reads conflict-free
    static_assert((TP / 4) % 2 == 1, "tap stride must be odd in 16-byte units");
    static constexpr int W_FLOATS = QCK * QKT * TP;
    static constexpr int X_FLOATS = QCK * QPLANE;
    static constexpr int BUF = W_FLOATS + X_FLOATS;
    static constexpr int X_IT = X_FLOATS / 256;
    static_assert(X_FLOATS % 256 == 0, "whole staging passes");
};

// MODE 0: forward (in = x, Cin = C, Cout = K, w[Cout][Cin][T]);  MODE 1: backward-data (in = dy, Cin = K, Cout = C, w[Cin][Cout][T])
template <int KS, int MODE>
__global__ __launch_bounds__(256, 2) void convkk_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ mask_src,
                                                        float* __restrict__ out, int N, int Cin, int Cout, int H, int W, int relu,
                                                        int parts, int kts) {
    using G = QGeo<KS>;
    constexpr int T = G::T, PAD = G::PAD, TP = G::TP;
    __shared__ __attribute__((aligned(16))) float lds[2 * G::BUF];
    __shared__ float bias_s[QKT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kk = lane >> 5;
    const int kt = blockIdx.x % kts, pb = blockIdx.x / kts;
    const int part = pb % parts, n = pb / parts;
    const int ko0 = kt * QKT;
    const int HW = H * W;
    const int p0 = part * QPIX;
    const int h_first = p0 / W;                       // first output row of the block; LDS row 0 = input row h_first - PAD
    const int n_chunks = Cin / QCK;
    if (MODE == 0 && tid < QKT) bias_s[tid] = (bias && ko0 + tid < Cout) ? bias[ko0 + tid] : 0.f;

    // ------------------------------------------------------------------ staging
    // weights: wave = channel of the chunk, lane = out-channel; T scalar loads per thread (the T taps of one (k, c) pair are
    // contiguous in w), written as 16-byte quads of taps — backward-data reverses the tap order here.
    const int wk = lane, wc = wave;
    const bool wk_ok = ko0 + wk < Cout;
    const __amdgpu_buffer_rsrc_t rs_w = clhip_rsrc(w, (size_t)Cin * Cout * T * sizeof(float));
    const int w_vo = wk_ok ? (MODE == 0 ? ((ko0 + wk) * Cin + wc) * T : (wc * Cout + ko0 + wk) * T) * 4 : CLHIP_OOB;
    const int w_step = (MODE == 0 ? QCK * T : QCK * Cout * T) * 4;         // bytes per chunk (scalar offset)
    const int w_dst = (wc * QKT + wk) * TP;
    float wr[T];
    // x: one scalar per halo-plane element; padding, plane tails and rows past the block read 0 through the offset predicate
    float xr[G::X_IT];
    int xoff[G::X_IT];
#pragma unroll
    for (int j = 0; j < G::X_IT; ++j) {
        const int e = tid + 256 * j;
        const int col = e & (QPW - 1), rc = e / QPW, cl = rc / QROWS, row = rc - cl * QROWS;
        const int h = h_first - PAD + row, ww = col - PAD;
        xoff[j] = (h >= 0 && h < H && ww >= 0 && ww < W) ? (cl * HW + h * W + ww) * 4 : CLHIP_OOB;
    }
    const float* in_img = in + (size_t)n * Cin * HW;
    auto load_chunk = [&](int chunk) {
        const int cw = chunk < n_chunks ? chunk : n_chunks - 1;            // the prefetches past the end re-read the last chunk
#pragma unroll
        for (int t = 0; t < T; ++t) wr[t] = clhip_buf_load(rs_w, w_vo, cw * w_step + t * 4);
        const __amdgpu_buffer_rsrc_t rs_x = clhip_rsrc(in_img + (size_t)cw * QCK * HW, (size_t)(Cin - cw * QCK) * HW * sizeof(float));
#pragma unroll
        for (int j = 0; j < G::X_IT; ++j) xr[j] = clhip_buf_load(rs_x, xoff[j], 0);
    };
    auto store_chunk = [&](int bo) {
        float* wd = lds + bo + w_dst;
#pragma unroll
        for (int q = 0; q < TP / 4; ++q) {
            floatx4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = 4 * q + i;                                   // tap as the MFMA loop numbers it
                v[i] = d < T ? wr[MODE == 0 ? d : T - 1 - d] : 0.f;
            }
            *reinterpret_cast<floatx4*>(wd + 4 * q) = v;
        }
        float* xs = lds + bo + G::W_FLOATS;
#pragma unroll
        for (int j = 0; j < G::X_IT; ++j) xs[tid + 256 * j] = xr[j];
    };

    // ------------------------------------------------------------------ lane operands
    // B: pixel p = p0 + (3 wave + sb) * 32 + li of the plane (clamped for addressing: slots past the plane compute garbage
    // that is never stored); window origin in the LDS halo = (h - h_first) * QPW + w, channel parity kk
    int xb[QSUB];
#pragma unroll
    for (int sb = 0; sb < QSUB; ++sb) {
        int p = p0 + (QSUB * wave + sb) * 32 + li;
        p = p < HW ? p : HW - 1;
        const int h = p / W, ww = p - h * W;
        xb[sb] = G::W_FLOATS + kk * QPLANE + (h - h_first) * QPW + ww;
    }
    const int a_off = (kk * QKT + li) * TP;                       // + ka * 32 * TP + cp * 2 * QKT * TP + 4 q

    floatx16 acc[2][QSUB];
#pragma unroll
    for (int ka = 0; ka < 2; ++ka)
#pragma unroll
        for (int sb = 0; sb < QSUB; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ka][sb][r] = 0.f;

    auto mac_pair = [&](const float* buf, int cp) {
        const float* ab = buf + a_off + cp * 2 * QKT * TP;
#pragma unroll
        for (int q = 0; q < TP / 4; ++q) {
            const floatx4 a0 = *reinterpret_cast<const floatx4*>(ab + 4 * q);
            const floatx4 a1 = *reinterpret_cast<const floatx4*>(ab + 32 * TP + 4 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 4 * q + i;
                if (t < T) {
                    const int r = t / KS, s = t - r * KS;
                    const int xo = cp * 2 * QPLANE + r * QPW + s;
                    float b[QSUB];
#pragma unroll
                    for (int sb = 0; sb < QSUB; ++sb) b[sb] = buf[xb[sb] + xo];
#pragma unroll
                    for (int sb = 0; sb < QSUB; ++sb) {
                        acc[0][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b[sb], acc[0][sb], 0, 0, 0);
                        acc[1][sb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b[sb], acc[1][sb], 0, 0, 0);
                    }
                }
            }
        }
    };

    load_chunk(0);
    store_chunk(0);
    load_chunk(1);
    __syncthreads();
    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        const int bo = (chunk & 1) * G::BUF;
        const float* cur = lds + bo;
        mac_pair(cur, 0);
        // chunk + 1 (in registers since the previous iteration) -> the other buffer: its last readers finished before the
        // barrier that ended the previous iteration; then chunk + 2 -> registers
        store_chunk(G::BUF - bo);
        load_chunk(chunk + 2);
        mac_pair(cur, 1);
        __syncthreads();
    }

    // ------------------------------------------------------------------ epilogue
    // register r of lane (li, kk) in acc[ka][sb]: out-channel ko0 + 32 ka + rch(r) + 4 kk, pixel of subtile sb
    auto rch = [](int r) { return (r & 3) + 8 * (r >> 2); };
    const size_t out_bytes = (size_t)N * Cout * HW * sizeof(float);
    const __amdgpu_buffer_rsrc_t rs_o = clhip_rsrc(out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_m = clhip_rsrc(MODE == 1 && mask_src ? mask_src : out, out_bytes);
#pragma unroll
    for (int ka = 0; ka < 2; ++ka) {
        if (ko0 + 32 * ka >= Cout) break;                          // Cout is a whole number of 32-channel row tiles
#pragma unroll
        for (int sb = 0; sb < QSUB; ++sb) {
            const int p = p0 + (QSUB * wave + sb) * 32 + li;
            const int vo = p < HW ? ((4 * kk) * HW + p) * 4 : CLHIP_OOB;
            const int so = ((n * Cout + ko0 + 32 * ka) * HW) * 4;
            float mk[16];
            if (MODE == 1 && mask_src) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mk[r] = clhip_buf_load(rs_m, vo, so + rch(r) * HW * 4);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[ka][sb][r];
                if (MODE == 0) {
                    v += bias_s[32 * ka + rch(r) + 4 * kk];
                    if (relu) v = fmaxf(v, 0.f);
                } else if (mask_src) {
                    v = mk[r] > 0.f ? v : 0.f;
                }
                clhip_buf_store(v, rs_o, vo, so + rch(r) * HW * 4);
            }
        }
    }
}

// rows of the LDS halo a block needs: output rows of its 384 pixels + 2 * pad
int convkk_rows(int H, int W, int pad) {
    const int HW = H * W, parts = (HW + QPIX - 1) / QPIX;
    int worst = 0;
    for (int part = 0; part < parts; ++part) {
        const int p0 = part * QPIX, p1 = (p0 + QPIX - 1 < HW - 1) ? p0 + QPIX - 1 : HW - 1;
        const int rows = p1 / W - p0 / W + 1 + 2 * pad;
        if (rows > worst) worst = rows;
    }
    return worst;
}

}  // namespace

// shapes the LDS-halo kernel takes; everything else stays on the gather-GEMM of conv2d.hip
bool clhip_internal_convkk_ok(int N, int Cin, int Cout, int H, int W, int R, int S, int stride, int pad) {
    if (R != 5 || S != 5 || stride != 1 || pad != 2) return false;
    if (N <= 0 || Cin % QCK || Cin < QCK || Cout % 32 || H < 1 || W < 1 || W + 2 * pad > QPW) return false;
    if (convkk_rows(H, W, pad) > QROWS) return false;
    const size_t hw = (size_t)H * W;
    return (size_t)N * Cin * hw * 4 < ((size_t)1 << 31) && (size_t)N * Cout * hw * 4 < ((size_t)1 << 31) &&
           (size_t)Cin * Cout * R * S * 4 < ((size_t)1 << 31);
}

// mode 0: forward (Cin = C, Cout = K);  mode 1: backward-data (in = dy, Cin = K, Cout = C, mask_src = the layer's input or NULL)
int clhip_internal_convkk(int mode, const float* in, const float* w, const float* bias, const float* mask_src, float* out, int N,
                          int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    const int parts = (H * W + QPIX - 1) / QPIX, kts = (Cout + QKT - 1) / QKT;
    const long long blocks = (long long)N * parts * kts;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return CLHIP_EINVAL;
    if (mode == 0)
        hipLaunchKernelGGL((convkk_kernel<5, 0>), dim3((unsigned)blocks), dim3(256), 0, s, in, w, bias, nullptr, out, N, Cin, Cout, H, W,
                           relu, parts, kts);
    else
        hipLaunchKernelGGL((convkk_kernel<5, 1>), dim3((unsigned)blocks), dim3(256), 0, s, in, w, nullptr, mask_src, out, N, Cin, Cout,
                           H, W, 0, parts, kts);
    CLHIP_LAUNCH_CHECK();
    return 0;
}
