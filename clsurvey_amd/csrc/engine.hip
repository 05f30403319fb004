// Static-plan executor for the reference's feature-extractor / classifier nets:
//   [conv (any square kernel / stride / pad) (+BatchNorm2d) (+ReLU) (+max-pool)]* -> [(Dropout) Linear (+ReLU)]*
// i.e. VGGSlim and its _BN / _DROP variants (models/VGGSlim.py:27-76) and torchvision's AlexNet (models/net.py:96-125).
// 3x3 stride-1 pad-1 layers take the conv3x3.hip fast path (fused ReLU + 2x2 pool forward, fused un-pool first-layer
// weight gradient, one deferred reduction launch for all weight-gradient slabs); everything else runs through the
// general kernels (conv2d.hip, pool.hip, bn.hip).  Side inputs per layer: dropout mask rows, an extra gradient for
// side branches (EBLL's code layers), BatchNorm running buffers.
//
// The reference trains through nn.Sequential + autograd, one Python dispatch per op per batch
// (EWC/train_EWC.py:181-189).  At N=200 a small_VGG9 step is < 1 ms of MFMA work, so the op
// sequence is planned once and every forward / loss / backward pass is ONE call that enqueues
// ~35 kernels on the caller's stream with all ReLU backward passes fused into the producing
// backward-data kernels.  The plan holds only shapes and offsets; parameters, gradients and
// the activation workspace are caller-owned device memory (torch allocations).
#include <vector>
#include <cstdlib>
#include <new>
#include "common.hpp"

namespace {

struct LayerPlan {
    int type;              // 0 conv3x3, 1 fc
    int cin, cout, relu, pool;
    long w_off, b_off;     // float offsets into the parameter / gradient arenas
    int h, w;              // conv: input spatial size
    int ks, st, pd;        // conv kernel / stride / pad (3,1,1 = the conv3x3.hip fast path)
    int oh, ow;            // conv output size
    int pk, ps;            // pool window / stride (2,2 = pool.hip's 2-bit fast path)
    int ph, pw;            // pooled size
    size_t in_elems, out_elems, pool_elems;   // per image
    size_t act_off;        // float offset of this layer's OUTPUT (post-ReLU, pre-pool) in ws
    size_t pool_off;       // float offset of pooled output
    size_t idx_off;        // byte offset of pool argmax
    int bn;                // BatchNorm2d between conv and ReLU
    long bn_w_off, bn_b_off;
    size_t z_off;          // bn: float offset of the convolution output (pre-BatchNorm) in ws
    size_t stat_off;       // bn: float offset of save_mean[cout], save_invstd[cout]
    float* rmean; float* rvar; float bn_momentum, bn_eps;
    size_t wino_uf, wino_ud;     // byte offsets of this layer's transformed weights (forward / backward-data) in the plan's Winograd region
    int wino_f, wino_d, wino_w;  // forward / backward-data / weight gradient through Winograd F(2x2,3x3) (wino.hip) instead of the direct kernels
    int bs5_f, bs5_d;            // 5 x 5 layers: forward / backward-data on the bf16-split kernel (image offsets in wino_uf / wino_ud)
    int bs5_w;                   // ... and their weight gradient (bswgrad5.hip; slabs in the plan's scratch)
    int bs_f, bs_d;              // ... forward / backward-data on the bf16 matrix cores with split fp32 operands (bsconv.hip): wino_f / wino_d
                                 // are set as well (the layer takes the prepared-weights path) and wino_uf / wino_ud hold its weight IMAGE
    int bs_w;                  // weight gradient on the bf16-split kernel (bswgrad.hip) instead of the Winograd / direct f32 kernels
    int s2d;                   // strided first layer through space-to-depth + the dense 3x3 kernels (s2dconv.hip); frames at s2d_off
    size_t s2d_off, s2d_bytes;
    int wg3;                   // weight gradient on the 3x3 kernel (else the general gather-GEMM)
    size_t wg_off, wg_bytes;   // this layer's own weight-gradient slabs (3x3 layers; reduced for all layers at once)
    const float* extra_grad;   // added to the gradient w.r.t. this layer's input (side branches: clhip_net_set_input_grad)
    const float* drop;     // dropout mask applied to this layer's INPUT (NULL = none); see clhip_net_set_dropout
    long drop_stride;      // floats between the mask rows of consecutive samples (0 = one row shared by the batch)
    int has_drop_buf;      // the masked input gets its own buffer (the un-masked activation stays readable: side branches)
    size_t drop_off;       // float offset of that buffer in ws
};

constexpr long long WINO_MIN_UNITS = 640;
constexpr long long WINO16_MIN_UNITS = 384;      // waves of the 8 x 8 variant (wino.hip, wino_conv16_kernel)
// CLHIP_WINO=0 keeps every layer on the direct kernels (A/B measurements, triage)
static bool wino_small_wgrad() { return true; }      // (layers with few stages per block: wino_wgrad_ps_kernel, wino.hip)

static bool wino_enabled() {
    const char* e = std::getenv("CLHIP_WINO");
    return !(e && e[0] == '0');
}

constexpr unsigned PROBE_RING = 64;
#ifndef CLHIP_OVERLAP_DEFAULT
#define CLHIP_OVERLAP_DEFAULT 0
#endif

struct NetPlan {
    std::vector<LayerPlan> layers;
    int max_batch, in_c, in_h, in_w, n_classes;
    size_t in_elems;
    size_t acts_floats;      // saved activations
    size_t idx_bytes;
    size_t grad_floats;      // one ping-pong gradient buffer
    size_t scratch_bytes;    // wgrad / fc split-K scratch
    size_t wino_bytes;       // transformed weights of the layer in flight (Winograd path)
    size_t off_wino;
    size_t off_s2d;          // frames of the space-to-depth first layer (its own region: the phase planes of the forward pass feed the
    int s2d_fresh_n;         // weight gradient of the same pass; s2d_fresh_n = the batch they were made for, 0 = stale)
    size_t total_bytes;
    // derived offsets (bytes) inside ws
    size_t off_acts, off_idx, off_g0, off_g1, off_scratch, off_dlogits, off_loss, off_fcdz, off_wg;
    int training;            // BatchNorm: batch statistics (1) or running statistics (0)
    int n_wg;                // 3x3 conv layers with deferred slab reduction (0: every layer reduces right away)
    // classifier = trailing Linear layers [fc_first, end): fused into three launches when it fits fc_chain.hip
    int fc_first;
    bool fc_fused;
    clhip_fc_chain chain;
    // layers fc_first+1 .. end forward + cross-entropy + backward-data down to dz(h1) in one launch (fc_tail_kernel); the
    // arrival counter of its last-workgroup reduction is the plan's own 4 bytes of device memory (zero between launches),
    // so a plan must not run on two streams at once (it never could: the activations live in one workspace)
    // measurement probe (clhip_net_probe): HIP events around the forward launch(es) of one layer, a ring of PROBE_RING
    // pairs so that recording never waits for the GPU
    int probe_layer;
    int probe_kind;          // 0 forward, 1 backward-data, 2 weight-gradient launch(es) of probe_layer
    unsigned probe_count;
    std::vector<hipEvent_t> probe_ev;
    bool fc_tail, no_combo;
    unsigned* tail_counter;
    size_t off_rows;
    // backward (optional, CLHIP_WGRAD_OVERLAP=1): the weight-gradient launches of the conv layers run on a side stream
    // next to the backward-data launch of the same layer (both only read dy)
    bool overlap;
    int overlap_mode;        // 0 off, 1 every conv layer (CLHIP_WGRAD_OVERLAP=1), 2 only layers whose launches under-fill the chip
    hipStream_t side;
    std::vector<hipEvent_t> ev_dy, ev_wg;      // per layer: dy ready (main -> side), weight gradient done (side -> main)
    ~NetPlan() {
        for (hipEvent_t e : ev_dy) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_wg) (void)hipEventDestroy(e);
        if (side) (void)hipStreamDestroy(side);
        if (tail_counter) (void)hipFree(tail_counter);
        for (hipEvent_t e : probe_ev) (void)hipEventDestroy(e);
    }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// h[n][f] *= mask[n * stride + f]: nn.Dropout in training mode with a caller-drawn mask (already divided by the
// retain probability), or GEM's one-row-per-observe masks (stride 0).  HBM-bound, in place.
__global__ void drop_scale_kernel(float* __restrict__ h, const float* __restrict__ mask, long stride, int feat, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / feat, f = i - n * feat;
        h[i] *= mask[n * stride + f];
    }
}

__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}

int add_inplace(float* a, const float* b, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, s, a, b, n);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

__global__ void drop_copy_kernel(const float* __restrict__ h, float* __restrict__ out, const float* __restrict__ mask, long stride,
                                 int feat, size_t total) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t n = i / feat, f = i - n * feat;
        out[i] = h[i] * mask[n * stride + f];
    }
}

int drop_copy(const float* h, float* out, const float* mask, long stride, size_t feat, int N, hipStream_t s) {
    const size_t total = feat * N;
    hipLaunchKernelGGL(drop_copy_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, h, out, mask, stride, (int)feat, total);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

int drop_scale(float* h, const float* mask, long stride, size_t feat, int N, hipStream_t s) {
    const size_t total = feat * N;
    hipLaunchKernelGGL(drop_scale_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, h, mask, stride, (int)feat, total);
    CLHIP_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

int clhip_net_create(const clhip_layer_desc* descs, int n_layers, int max_batch, int in_c, int in_h, int in_w,
                     void** out_handle) {
    if (!descs || n_layers <= 0 || max_batch <= 0 || !out_handle) return CLHIP_EINVAL;
    NetPlan* p = new (std::nothrow) NetPlan();
    if (!p) return CLHIP_ENOSPC;
    p->max_batch = max_batch; p->in_c = in_c; p->in_h = in_h; p->in_w = in_w;
    p->in_elems = (size_t)in_c * in_h * in_w;
    int c = in_c, h = in_h, w = in_w;
    size_t feat = p->in_elems;
    size_t acts = 0, idxb = 0, gmax = 0, scratch = 0, wg_total = 0, wino_ws = 0, s2d_total = 0;
    int n_wg = 0;
    bool seen_fc = false;
    for (int i = 0; i < n_layers; ++i) {
        LayerPlan L{};
        L.type = descs[i].type; L.cin = descs[i].cin; L.cout = descs[i].cout;
        L.relu = descs[i].relu; L.pool = descs[i].pool; L.w_off = descs[i].w_off; L.b_off = descs[i].b_off;
        if (L.type == 0) {
            if (seen_fc || L.cin != c) { delete p; return CLHIP_EINVAL; }
            L.h = h; L.w = w;
            L.ks = descs[i].ksize > 0 ? descs[i].ksize : 3;
            L.st = descs[i].ksize > 0 ? descs[i].stride : 1;
            L.pd = descs[i].ksize > 0 ? descs[i].pad : 1;
            if (L.st < 1 || L.pd < 0) { delete p; return CLHIP_EINVAL; }
            L.oh = (h + 2 * L.pd - L.ks) / L.st + 1; L.ow = (w + 2 * L.pd - L.ks) / L.st + 1;
            if (L.oh <= 0 || L.ow <= 0) { delete p; return CLHIP_EINVAL; }
            const bool vgg = L.ks == 3 && L.st == 1 && L.pd == 1;
            L.in_elems = (size_t)c * h * w;
            L.out_elems = (size_t)L.cout * L.oh * L.ow;
            L.bn = descs[i].bn ? 1 : 0; L.bn_w_off = descs[i].bn_w_off; L.bn_b_off = descs[i].bn_b_off;
            L.bn_momentum = 0.1f; L.bn_eps = 1e-5f;
            if (L.bn) {
                L.z_off = acts; acts += L.out_elems * max_batch;
                L.stat_off = acts; acts += align_up(2 * (size_t)L.cout, 4);
                const size_t bs = clhip_bn_ws(L.cout);
                if (bs > scratch) scratch = bs;
            }
            L.act_off = acts; acts += L.out_elems * max_batch;
            int oh = L.oh, ow = L.ow;
            if (L.pool) {
                L.pk = descs[i].pool_k > 0 ? descs[i].pool_k : 2;
                L.ps = descs[i].pool_k > 0 ? descs[i].pool_s : 2;
                if (L.ps < 1 || oh < L.pk || ow < L.pk) { delete p; return CLHIP_EINVAL; }
                if (L.pk == 2 && L.ps == 2 && ((oh & 1) || (ow & 1))) { delete p; return CLHIP_EINVAL; }
                L.ph = (oh - L.pk) / L.ps + 1; L.pw = (ow - L.pk) / L.ps + 1;
                L.pool_elems = (size_t)L.cout * L.ph * L.pw;
                L.pool_off = acts; acts += L.pool_elems * max_batch;
                L.idx_off = idxb; idxb += align_up(L.pool_elems * max_batch, 256);
                oh = L.ph; ow = L.pw;
            }
            // 3x3 layers whose rows are not 16-byte aligned (AlexNet's 13x13 maps) would take the scalar-staging variant of
            // the 3x3 weight-gradient kernel with 16x4-pixel tiles (169 of 256 slots): the dense gather-GEMM is 1.8x faster
            // there (measured 83 vs 46 TFLOP/s)
            L.wg3 = (vgg && (L.w % 4 == 0 || L.cin * 9 <= 32)) ? 1 : 0;
            size_t s = L.wg3 ? clhip_conv3x3_bwd_weight_ws(max_batch, L.cin, L.cout, L.h, L.w)
                             : clhip_conv2d_bwd_weight_ws(max_batch, L.cin, L.h, L.w, L.cout, L.ks, L.ks, L.st, L.pd);
            if (s > scratch) scratch = s;
            if (L.wg3) { L.wg_off = wg_total; L.wg_bytes = align_up(s, 256); wg_total += L.wg_bytes; ++n_wg; }
            // Winograd path: 2.25x fewer matrix instructions, but a wave's work unit is 32 channels x 32 TILES (128
            // pixels) with 256 accumulator registers, one wave per SIMD: it needs enough units to fill most of the 1024 SIMDs
            // (measured at N = 200: 64->64 @16x16, 800 units, 1.26x the direct kernel; 128->128 @8x8, 400 units, 0.8x)
            if (vgg && !L.bn) {
                const long long tiles = (long long)max_batch * ((L.h + 1) / 2) * ((L.w + 1) / 2);
                auto units = [&](int kout) { return ((tiles + 31) / 32) * ((kout + 31) / 32); };
                // (8 x 8 maps: the 16x16x4-MFMA variant has 16-tile units, one image x 32 channels per wave)
                auto enough = [&](int kout) {
                    if (units(kout) >= WINO_MIN_UNITS) return true;
                    return L.h == 8 && L.w == 8 && (long long)max_batch * ((kout + 31) / 32) >= WINO16_MIN_UNITS;
                };
                const bool wino = wino_enabled();
                L.wino_f = wino && clhip_internal_wino_ok(L.cin, L.cout, L.h, L.w) && enough(L.cout);
                L.wino_d = wino && i > 0 && clhip_internal_wino_ok(L.cout, L.cin, L.h, L.w) && enough(L.cin);
                // bf16 matrix cores with split fp32 operands (bsconv.hip) where that path is the faster one (large maps: see
                // clhip_internal_bs_preferred; CLHIP_BS=0 turns it off, =2 takes it wherever it can run); fused 2x2 pooling only exists
                // on even maps there
                const bool even = ((L.h | L.w) & 1) == 0;
                L.bs_f = clhip_internal_bs_preferred(L.cin, L.cout, L.h, L.w) && (even || !L.pool);
                L.bs_d = i > 0 && clhip_internal_bs_preferred(L.cout, L.cin, L.h, L.w) && (even || !L.pool);
                if (L.bs_f) L.wino_f = 1;
                if (L.bs_d) L.wino_d = 1;
                // weight gradient: the reduction (tiles) splits over ~256 blocks per 64x64 (k, c) tile; each block needs a
                // few 16-tile stages to amortise its 256-accumulator epilogue (measured: 12.5 stages per block 1.44x, 3.1 0.6x)
                if (wino && clhip_internal_wino_wgrad_ok(L.cin, L.cout, L.h, L.w)) {
                    const int kc = (L.cin / 64) * (L.cout / 64), splits = (256 + kc - 1) / kc;
                    const long long stages = (long long)max_batch * (((L.w + 1) / 2 + (L.w >= 16 ? 7 : 3)) / (L.w >= 16 ? 8 : 4)) *
                                             (((L.h + 1) / 2 + (L.w >= 16 ? 1 : 3)) / (L.w >= 16 ? 2 : 4));
                    // (fewer stages per block: wino_wgrad_ps_kernel — 32 x 32 tiles, a quarter of the splits — inside the same entry point;
                    // measured faster than the direct pixel-split kernel on those layers in round 3)
                    L.wino_w = stages >= 8LL * splits || wino_small_wgrad();
                    if (L.wino_w) {                   // its slabs live in this layer's weight-gradient region
                        const size_t ww = clhip_internal_wino_wgrad_ws(max_batch, L.cin, L.cout, L.h, L.w);
                        if (L.wg3) {
                            if (ww > L.wg_bytes) { wg_total += align_up(ww, 256) - L.wg_bytes; L.wg_bytes = align_up(ww, 256); }
                        } else {                      // (AlexNet's 13x13 layers: the direct path is the gather-GEMM, which has no slabs)
                            L.wg_off = wg_total; L.wg_bytes = align_up(ww, 256); wg_total += L.wg_bytes; ++n_wg;
                        }
                    }
                }
                // ... or on the bf16 matrix cores with split fp32 operands (bswgrad.hip) where that is the faster launch and the layer's
                // backward is not one merged grid: its slabs live in the same region
                {
                    const bool pooled22 = L.pool && L.pk == 2 && L.ps == 2;
                    const bool merged = i > 0 && L.wino_w && L.wino_d && !L.bs_d && L.wg_bytes &&
                                        clhip_internal_wino_pair_shape(max_batch, L.cin, L.cout, L.h, L.w, pooled22 ? 1 : 0);
                    if (!merged && clhip_internal_bs_wgrad_preferred(L.cin, L.cout, L.h, L.w, pooled22 ? 1 : 0)) {
                        const size_t bw = clhip_internal_bs_wgrad_ws(max_batch, L.cin, L.cout, L.h, L.w);
                        if (bw && L.wg3) {
                            L.bs_w = 1;
                            if (bw > L.wg_bytes) { wg_total += align_up(bw, 256) - L.wg_bytes; L.wg_bytes = align_up(bw, 256); }
                        }
                    }
                }
                // every Winograd layer keeps its own transformed weights: ONE transform launch per pass fills them all
                if (L.wino_f) { L.wino_uf = wino_ws; wino_ws += align_up(L.bs_f ? clhip_internal_bs_ws(L.cin, L.cout) : clhip_internal_wino_ws(L.cin, L.cout), 256); }
                if (L.wino_d) { L.wino_ud = wino_ws; wino_ws += align_up(L.bs_d ? clhip_internal_bs_ws(L.cout, L.cin) : clhip_internal_wino_ws(L.cout, L.cin), 256); }
            }
            // 5 x 5 / stride 1 / padding 2 layers (AlexNet's second convolution) on the bf16-split kernel: bs5_f / bs5_d, images in the
            // same region
            if (!vgg && !L.bn && L.ks == 5 && L.st == 1 && L.pd == 2) {
                L.bs5_f = clhip_internal_bs5_preferred(L.cin, L.cout, L.h, L.w);
                L.bs5_d = i > 0 && clhip_internal_bs5_preferred(L.cout, L.cin, L.h, L.w);
                // weight gradient: 372 us against the gather-GEMM's 589 at N = 128 (profiles/r06_x_bswgrad5.txt); CLHIP_BS_WGRAD=0: off
                {
                    const char* e = std::getenv("CLHIP_BS_WGRAD");
                    const size_t w5 = (e && e[0] == '0') ? 0 : clhip_internal_bs5_wgrad_ws(max_batch, L.cin, L.cout, L.h, L.w);
                    if (w5) { L.bs5_w = 1; if (w5 > scratch) scratch = w5; }
                }
                if (L.bs5_f) { L.wino_uf = wino_ws; wino_ws += align_up(clhip_internal_bs5_ws(L.cin, L.cout), 256); }
                if (L.bs5_d) { L.wino_ud = wino_ws; wino_ws += align_up(clhip_internal_bs5_ws(L.cout, L.cin), 256); }
            }
            // AlexNet's 11 x 11 / stride-4 first layer as a dense 3x3 convolution over the 48 phase planes of the padded input
            // (s2dconv.hip), CLHIP_S2D=1.  Off by default: measured at N = 128 (profiles/r06_l_alexnet_s2d.txt) the dense kernels take
            // 159 (forward) and 192 us (weight gradient) but building the frames and cropping the output costs 64 + 62 + 45 us more —
            // 285 / 237 us against the gather-GEMM's 257 / 217; the step 4.72 against 4.69 ms.  First layer only: no backward-data.
            if (!vgg && !L.bn && i == 0 && !descs[i].has_drop) {
                const char* e = std::getenv("CLHIP_S2D");
                const size_t sb = !(e && e[0] == '1') ? 0 : clhip_conv2d_s2d_ws(max_batch, L.cin, L.h, L.w, L.cout, L.ks, L.st, L.pd);
                if (sb) { L.s2d = 1; L.s2d_off = s2d_total; L.s2d_bytes = align_up(sb, 256); s2d_total += L.s2d_bytes; }
            }
            if (L.out_elems > gmax) gmax = L.out_elems;
            if (L.in_elems > gmax) gmax = L.in_elems;
            h = oh; w = ow;
            c = L.cout;
            feat = (size_t)c * h * w;
        } else if (L.type == 1) {
            seen_fc = true;
            if ((size_t)L.cin != feat) { delete p; return CLHIP_EINVAL; }
            L.in_elems = feat; L.out_elems = (size_t)L.cout;
            L.act_off = acts; acts += L.out_elems * max_batch;
            size_t s = clhip_fc_ws(max_batch, L.cin, L.cout);
            if (s > scratch) scratch = s;
            if (L.in_elems > gmax) gmax = L.in_elems;
            if (L.out_elems > gmax) gmax = L.out_elems;
            feat = L.out_elems;
        } else { delete p; return CLHIP_EINVAL; }
        L.has_drop_buf = (descs[i].has_drop && i > 0) ? 1 : 0;
        if (L.has_drop_buf) { L.drop_off = acts; acts += L.in_elems * max_batch; }
        p->layers.push_back(L);
    }
    p->n_classes = (int)feat;
    p->acts_floats = acts; p->idx_bytes = idxb; p->grad_floats = gmax * max_batch; p->scratch_bytes = scratch;
    size_t off = 0;
    p->off_acts = off; off += align_up(acts * 4, 256);
    p->off_idx = off; off += align_up(idxb, 256);
    p->off_g0 = off; off += align_up(p->grad_floats * 4, 256);
    p->off_g1 = off; off += align_up(p->grad_floats * 4, 256);
    p->off_scratch = off; off += align_up(scratch, 256);
    p->wino_bytes = wino_ws;
    p->off_wino = off; off += align_up(wino_ws, 256);
    p->off_dlogits = off; off += align_up((size_t)max_batch * p->n_classes * 4, 256);
    p->off_loss = off; off += 256;
    // fused classifier
    p->fc_first = n_layers;
    for (int i = 0; i < n_layers; ++i) if (p->layers[i].type == 1) { p->fc_first = i; break; }
    p->fc_fused = false;
    p->off_fcdz = off;
    const int nfc = n_layers - p->fc_first;
    if (nfc >= 1 && nfc <= CLHIP_FC_MAX) {
        clhip_fc_chain& ch = p->chain;
        ch.n = nfc;
        size_t dzf = 0;
        for (int l = 0; l < nfc; ++l) {
            const LayerPlan& L = p->layers[p->fc_first + l];
            ch.w_off[l] = L.w_off; ch.b_off[l] = L.b_off; ch.din[l] = L.cin; ch.dout[l] = L.cout; ch.relu[l] = L.relu;
            ch.act_off[l] = L.act_off;
            ch.dz_off[l] = dzf;
            if (l < nfc - 1) dzf += (size_t)L.cout * max_batch;
        }
        if (clhip_internal_fc_chain_ok(&ch)) {
            p->fc_fused = true;
            off += align_up(dzf * 4 + 256, 256);
            p->off_rows = off;                       // per-row loss / hit of the fused classifier tail
            off += align_up((size_t)max_batch * 8, 256);
        }
    }
    // own slabs per 3x3 layer so that ONE launch reduces them all at the end of backward (HBM is plentiful: 288 GB)
    p->off_wg = off;
    p->n_wg = (n_wg >= 2 && n_wg <= CLHIP_WGRAD_JOBS_MAX) ? n_wg : 0;
    if (p->n_wg) off += align_up(wg_total, 256);
    p->off_s2d = off; off += s2d_total;
    p->s2d_fresh_n = 0;
    p->total_bytes = off;
    p->training = 1;
    p->overlap = false;
    p->side = nullptr;
    // Measured on small_VGG9 (N = 200): 2.59 ms per bench step with the overlap against 2.45 ms without (the two
    // MFMA-bound kernels only take slots from each other); AlexNet N = 128: 9.65 vs 9.76 ms.  Hence off unless
    // CLHIP_WGRAD_OVERLAP=1; results are identical either way (same kernels, same order per stream).
    // Round 2 tried the side stream for the DEEP layers only (planes of 16x16 and smaller at batch 200: ~400 blocks, 1.5 per
    // CU, SIMD time left unused) with the slab reductions still deferred: 2.125 against 2.076 ms per step — the event
    // hand-offs and the two launches taking each other's slots cost more than the idle time they fill.  So: off by default;
    // CLHIP_WGRAD_OVERLAP=1 every conv layer, =2 the deep layers only.
    const char* ov = getenv("CLHIP_WGRAD_OVERLAP");
    bool any_bn = false;
    for (const LayerPlan& L : p->layers) any_bn = any_bn || L.bn;
    p->overlap_mode = (ov && ov[0] == '1') ? 1 : (ov && ov[0] == '2') ? 2 : CLHIP_OVERLAP_DEFAULT;
    if (p->overlap_mode && !any_bn) {         // BatchNorm backward shares `scratch` with the weight-gradient launches
        // needs a device: plans made on a host without one (shape tests) simply stay single-stream
        if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) == hipSuccess) {
            p->overlap = true;
            p->ev_dy.resize(n_layers); p->ev_wg.resize(n_layers);
            for (int i = 0; i < n_layers && p->overlap; ++i)
                if (hipEventCreateWithFlags(&p->ev_dy[i], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&p->ev_wg[i], hipEventDisableTiming) != hipSuccess) p->overlap = false;
        } else {
            (void)hipGetLastError();
            p->side = nullptr;
        }
    }
    p->probe_layer = -1;
    p->probe_kind = 0;
    p->probe_count = 0;
    p->fc_tail = false;
    p->no_combo = false;
    p->tail_counter = nullptr;
    const char* tl = getenv("CLHIP_FC_TAIL");      // CLHIP_FC_TAIL=0: per-layer launches (the bitwise reference of the fused tail)
    if (tl && tl[0] == '0') p->no_combo = true;
    if (p->fc_fused && !(tl && tl[0] == '0') && clhip_internal_fc_tail_ok(&p->chain) &&
        max_batch <= 1024) {
        // needs a device: plans made on a host without one (shape tests) keep the per-layer launches
        if (hipMalloc(reinterpret_cast<void**>(&p->tail_counter), 256) == hipSuccess &&
            hipMemset(p->tail_counter, 0, 256) == hipSuccess) {
            p->fc_tail = true;
        } else {
            (void)hipGetLastError();
            if (p->tail_counter) { (void)hipFree(p->tail_counter); p->tail_counter = nullptr; }
        }
    }
    *out_handle = p;
    return 0;
}

// Dropout in front of layer `layer` (> 0): mask is device memory, [N][in_elems] with row_stride floats between samples
// (row_stride 0: one row for the whole batch), values 0 or 1/p_retain.  NULL switches it off (eval mode).  The mask must
// stay valid and unchanged from a forward to its backward.
int clhip_net_set_dropout(void* handle, int layer, const float* mask, long row_stride) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer <= 0 || layer >= (int)p->layers.size() || row_stride < 0) return CLHIP_EINVAL;
    p->layers[layer].drop = mask;
    p->layers[layer].drop_stride = row_stride;
    return 0;
}

int clhip_net_set_bn(void* handle, int layer, float* running_mean, float* running_var, float momentum, float eps) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer < 0 || layer >= (int)p->layers.size() || !p->layers[layer].bn || !(eps > 0.f)) return CLHIP_EINVAL;
    LayerPlan& L = p->layers[layer];
    L.rmean = running_mean; L.rvar = running_var; L.bn_momentum = momentum; L.bn_eps = eps;
    return 0;
}

int clhip_net_layer_input(void* handle, int layer, size_t* ws_float_off, size_t* in_elems) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer <= 0 || layer >= (int)p->layers.size() || !ws_float_off || !in_elems) return CLHIP_EINVAL;
    const LayerPlan& P = p->layers[layer - 1];
    *ws_float_off = p->off_acts / sizeof(float) + ((P.type == 0 && P.pool) ? P.pool_off : P.act_off);
    *in_elems = p->layers[layer].in_elems;
    return 0;
}

int clhip_net_layer_pool_idx(void* handle, int layer, size_t* ws_byte_off, size_t* elems) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer < 0 || layer >= (int)p->layers.size() || !ws_byte_off || !elems) return CLHIP_EINVAL;
    const LayerPlan& L = p->layers[layer];
    if (L.type != 0 || !L.pool) return CLHIP_EINVAL;
    *ws_byte_off = p->off_idx + L.idx_off;
    *elems = L.pool_elems;
    return 0;
}

int clhip_net_layer_paths(void* handle, int layer) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer < 0 || layer >= (int)p->layers.size()) return CLHIP_EINVAL;
    const LayerPlan& L = p->layers[layer];
    // (bit 2: the Winograd weight gradient runs only inside the deferred-reduction scheme, net_backward_impl's `defer`; and even
    // then the kernel may hand single shapes back to the direct path — odd maps with a fused un-pool, > 2^31-byte offsets)
    const bool defer_capable = p->n_wg > 0 && !(p->overlap && p->overlap_mode == 1);
    // bits 3 / 4: the forward / backward-data launch is the bf16-split kernel (bsconv.hip), not Winograd (bits 0 / 1 then say
    // "prepared-weights path")
    // bit 5: the weight gradient is the bf16-split kernel (bswgrad.hip)
    return (L.wino_f ? 1 : 0) | (L.wino_d ? 2 : 0) | ((L.wino_w && defer_capable) ? 4 : 0) | ((L.bs_f || L.bs5_f) ? 8 : 0) |
           ((L.bs_d || L.bs5_d) ? 16 : 0) | ((L.bs_w && defer_capable) ? 32 : 0);
}

int clhip_net_set_input_grad(void* handle, int layer, const float* extra) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer <= 0 || layer >= (int)p->layers.size()) return CLHIP_EINVAL;
    p->layers[layer].extra_grad = extra;
    return 0;
}

// Measurement: time the forward launch(es) of plan layer `layer` with HIP events on the stream they are issued on
// (layer < 0: off).  clhip_net_probe_read waits for the recorded launches and returns their average duration over the
// last (at most 64) forward passes, then clears the count.
int clhip_net_probe(void* handle, int layer) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || layer >= (int)p->layers.size()) return CLHIP_EINVAL;
    if (layer >= 0 && p->probe_ev.empty()) {
        std::vector<hipEvent_t> made;
        made.reserve(2 * PROBE_RING);
        for (unsigned i = 0; i < 2 * PROBE_RING; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) {
                for (hipEvent_t m : made) (void)hipEventDestroy(m);
                return CLHIP_ENOTSUP;
            }
            made.push_back(e);
        }
        p->probe_ev.swap(made);
    }
    p->probe_layer = layer;
    p->probe_kind = 0;
    p->probe_count = 0;
    return 0;
}

int clhip_net_probe_kind(void* handle, int layer, int kind) {
    if (kind < 0 || kind > 2) return CLHIP_EINVAL;
    const int rc = clhip_net_probe(handle, layer);
    if (rc == 0) static_cast<NetPlan*>(handle)->probe_kind = kind;
    return rc;
}

int clhip_net_probe_read(void* handle, float* avg_us, int* count) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || !avg_us || !count) return CLHIP_EINVAL;
    const unsigned n = p->probe_count < PROBE_RING ? p->probe_count : PROBE_RING;
    double sum = 0.0;
    for (unsigned i = 0; i < n; ++i) {
        const unsigned slot = (p->probe_count - 1 - i) % PROBE_RING;
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(p->probe_ev[2 * slot + 1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, p->probe_ev[2 * slot], p->probe_ev[2 * slot + 1]);
        if (e != hipSuccess) return (int)e;
        sum += ms;
    }
    *avg_us = n ? (float)(sum / n * 1e3) : 0.f;
    *count = (int)n;
    p->probe_count = 0;
    return 0;
}

int clhip_net_set_training(void* handle, int training) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p) return CLHIP_EINVAL;
    p->training = training ? 1 : 0;
    return 0;
}

void clhip_net_destroy(void* handle) { delete static_cast<NetPlan*>(handle); }

size_t clhip_net_workspace_bytes(void* handle) { return handle ? static_cast<NetPlan*>(handle)->total_bytes : 0; }
int clhip_net_num_classes(void* handle) { return handle ? static_cast<NetPlan*>(handle)->n_classes : 0; }

// Forward pass; activations are kept in ws for a following backward. logits_out (optional) receives
// a copy of the [N][classes] logits.
// tail: 0 = every layer; 1 = stop behind the first Linear layer, the caller launches the fused tail itself (loss step);
// 2 = stop there and finish with the fused tail, forward only.
static bool tail_usable(const NetPlan* p, const float* params, const void* ws, int N) {
    if (!p->fc_tail) return false;
    for (size_t i = p->fc_first + 1; i < p->layers.size(); ++i)
        if (p->layers[i].drop || p->layers[i].extra_grad) return false;
    const float* acts = reinterpret_cast<const float*>(static_cast<const char*>(ws) + p->off_acts);
    return aligned16(params + p->chain.w_off[1]) && aligned16(params + p->chain.w_off[2]) && aligned16(acts + p->chain.act_off[0]) &&
           aligned16(params + p->chain.b_off[0]) &&
           (size_t)N * (p->n_classes | 1) <= 12288;        // the range in which the per-layer path uses softmax_ce_rows_lds_kernel
}

// the prepared-weights convolution of a layer: Winograd (wino.hip) or bf16-split (bsconv.hip), same arguments
static int plan_conv_u(int bs, int mode, const float* in, const float* U, const float* bias, const float* mask_src, float* out,
                       uint8_t* pool_idx, int unpool, int N, int Cin, int Cout, int H, int W, int relu, hipStream_t s) {
    return bs ? clhip_internal_bs_conv_u(mode, in, U, bias, mask_src, out, pool_idx, unpool, N, Cin, Cout, H, W, relu, s)
              : clhip_internal_wino_conv_u(mode, in, U, bias, mask_src, out, pool_idx, unpool, N, Cin, Cout, H, W, relu, s);
}

// transformed weights of every Winograd / bf16-split layer (forward set, backward-data set, or both) in one launch each
static int wino_prepare(NetPlan* p, const float* params, char* base, bool fwd, bool bwd, hipStream_t s) {
    std::vector<clhip_wino_wt> jobs, bsjobs;             // (sized by the plan: a net of any depth gets every transform)
    jobs.reserve(2 * p->layers.size());
    bsjobs.reserve(2 * p->layers.size());
    for (const LayerPlan& L : p->layers) {
        if (L.type != 0) continue;
        if (fwd && L.wino_f)
            (L.bs_f ? bsjobs : jobs).push_back(clhip_wino_wt{params + L.w_off, reinterpret_cast<float*>(base + p->off_wino + L.wino_uf), L.cout, L.cin, 0, 0});
        if (bwd && L.wino_d)
            (L.bs_d ? bsjobs : jobs).push_back(clhip_wino_wt{params + L.w_off, reinterpret_cast<float*>(base + p->off_wino + L.wino_ud), L.cin, L.cout, 1, 0});
        if (fwd && L.bs5_f)
            bsjobs.push_back(clhip_wino_wt{params + L.w_off, reinterpret_cast<float*>(base + p->off_wino + L.wino_uf), L.cout, L.cin, 0, 5});
        if (bwd && L.bs5_d)
            bsjobs.push_back(clhip_wino_wt{params + L.w_off, reinterpret_cast<float*>(base + p->off_wino + L.wino_ud), L.cin, L.cout, 1, 5});
    }
    // (ONE launch for the Winograd U images and the bf16-split images of the pass when they fit one job table: a boundary between two
    // launches of 5 - 7 us each costs as much as either)
    return clhip_internal_weight_images(jobs.data(), (int)jobs.size(), bsjobs.data(), (int)bsjobs.size(), s);
}

static int net_forward_impl(void* handle, const float* params, const float* x, int N, void* ws, float* logits_out,
                            void* stream, int tail, int* slabs_live = nullptr, bool prep_bwd = false) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || !params || !x || !ws || N <= 0 || N > p->max_batch) return CLHIP_EINVAL;
    char* base = static_cast<char*>(ws);
    float* acts = reinterpret_cast<float*>(base + p->off_acts);
    uint8_t* idx = reinterpret_cast<uint8_t*>(base + p->off_idx);
    void* scratch = base + p->off_scratch;
    if (p->wino_bytes) {
        int rcw = wino_prepare(p, params, base, true, prep_bwd, as_stream(stream));
        if (rcw) return rcw;
    }
    const float* cur = x;
    int rc;
    const size_t n_run = tail ? (size_t)p->fc_first + 1 : p->layers.size();
    int live = 0;            // split-K slabs of the first Linear layer left in `scratch` for the fused tail
    for (size_t li = 0; li < n_run; ++li) {
        const LayerPlan& L = p->layers[li];
        float* y = acts + L.act_off;
        const bool probed = (int)li == p->probe_layer && p->probe_kind == 0 && !p->probe_ev.empty();
        if (probed) (void)hipEventRecord(p->probe_ev[2 * (p->probe_count % PROBE_RING)], as_stream(stream));
        struct ProbeEnd {       // closes the pair when the layer's launches are issued (every exit of the loop body)
            NetPlan* p; bool on; hipStream_t s;
            ~ProbeEnd() { if (on) { (void)hipEventRecord(p->probe_ev[2 * (p->probe_count % PROBE_RING) + 1], s); ++p->probe_count; } }
        } probe_end{p, probed, as_stream(stream)};
        if (L.drop && L.has_drop_buf) {       // masked copy: backward reads it as this layer's input, cur stays intact
            float* dropped = acts + L.drop_off;
            rc = drop_copy(cur, dropped, L.drop, L.drop_stride, L.in_elems, N, as_stream(stream));
            if (rc) return rc;
            cur = dropped;
        } else if (L.drop) {    // li > 0: cur is the previous layer's saved output; masked in place so that backward sees h * m
            rc = drop_scale(const_cast<float*>(cur), L.drop, L.drop_stride, L.in_elems, N, as_stream(stream));
            if (rc) return rc;
        }
        const bool vgg = L.type == 0 && L.ks == 3 && L.st == 1 && L.pd == 1;
        const bool pool22 = L.pool && L.pk == 2 && L.ps == 2;
        if (L.type == 0 && (L.bn || !(vgg && (!L.pool || pool22)))) {
            // general geometry (AlexNet) and BatchNorm layers: separate conv (+bias, ReLU), BatchNorm and pool kernels
            float* zc = L.bn ? acts + L.z_off : y;
            const int crelu = L.bn ? 0 : L.relu;
            rc = (vgg && L.wino_f)
                     ? plan_conv_u(L.bs_f, 0, cur, reinterpret_cast<const float*>(base + p->off_wino + L.wino_uf), params + L.b_off,
                                                  nullptr, zc, nullptr, 0, N, L.cin, L.cout, L.h, L.w, crelu, as_stream(stream))
                 : vgg ? clhip_conv3x3_fwd(cur, params + L.w_off, params + L.b_off, zc, N, L.cin, L.cout, L.h, L.w, crelu, stream)
                 : L.bs5_f ? clhip_internal_bs5_conv_u(0, cur, base + p->off_wino + L.wino_uf, params + L.b_off, nullptr, zc, N, L.cin, L.cout,
                                                       L.h, L.w, crelu, as_stream(stream))
                 : L.s2d ? clhip_conv2d_s2d_fwd(cur, params + L.w_off, params + L.b_off, zc, N, L.cin, L.h, L.w, L.cout, L.ks, L.st, L.pd,
                                                crelu, base + p->off_s2d + L.s2d_off, L.s2d_bytes, stream)
                     : clhip_conv2d_fwd(cur, params + L.w_off, params + L.b_off, zc, N, L.cin, L.h, L.w, L.cout, L.ks, L.ks, L.st,
                                        L.pd, crelu, stream);
            if (rc) return rc;
            if (L.s2d) p->s2d_fresh_n = N;
            if (L.bn) {
                float* st = acts + L.stat_off;
                rc = clhip_bn_fwd(zc, params + L.bn_w_off, params + L.bn_b_off, L.rmean, L.rvar, y, st, st + L.cout, N, L.cout,
                                  L.oh * L.ow, p->training, L.bn_momentum, L.bn_eps, L.relu, scratch, p->scratch_bytes, stream);
                if (rc) return rc;
            }
            cur = y;
            if (L.pool) {
                float* pl = acts + L.pool_off;
                rc = pool22 ? clhip_maxpool2_fwd(y, pl, idx + L.idx_off, N * L.cout, L.oh, L.ow, stream)
                            : clhip_maxpool_fwd(y, pl, idx + L.idx_off, N * L.cout, L.oh, L.ow, L.pk, L.ps, stream);
                if (rc) return rc;
                cur = pl;
            }
        } else if (L.type == 0) {
            if (L.pool && L.relu) {
                // conv + bias + ReLU + max-pool in one kernel; the pre-pool tensor is never materialised
                float* pl = acts + L.pool_off;
                rc = L.wino_f ? plan_conv_u(L.bs_f, 0, cur, reinterpret_cast<const float*>(base + p->off_wino + L.wino_uf),
                                                           params + L.b_off, nullptr, pl, idx + L.idx_off, 0, N, L.cin, L.cout, L.h, L.w, 1,
                                                           as_stream(stream))
                              : clhip_conv3x3_relu_pool_fwd(cur, params + L.w_off, params + L.b_off, pl, idx + L.idx_off, N, L.cin,
                                                            L.cout, L.h, L.w, stream);
                if (rc) return rc;
                cur = pl;
            } else {
                rc = L.wino_f ? plan_conv_u(L.bs_f, 0, cur, reinterpret_cast<const float*>(base + p->off_wino + L.wino_uf),
                                                           params + L.b_off, nullptr, y, nullptr, 0, N, L.cin, L.cout, L.h, L.w, L.relu,
                                                           as_stream(stream))
                              : clhip_conv3x3_fwd(cur, params + L.w_off, params + L.b_off, y, N, L.cin, L.cout, L.h, L.w, L.relu, stream);
                if (rc) return rc;
                cur = y;
                if (L.pool) {
                    float* pl = acts + L.pool_off;
                    rc = clhip_maxpool2_fwd(y, pl, idx + L.idx_off, N * L.cout, L.h, L.w, stream);
                    if (rc) return rc;
                    cur = pl;
                }
            }
        } else {
            if (tail && (int)li == p->fc_first) {
                // the fused tail sums the split-K slabs of this layer itself (no reduction launch)
                rc = clhip_internal_fc_fwd_partial(cur, params + L.w_off, N, L.cin, L.cout, scratch, p->scratch_bytes, &live,
                                                   as_stream(stream));
                if (rc) return rc;
            }
            if (!live) {
                rc = clhip_fc_fwd(cur, params + L.w_off, params + L.b_off, y, N, L.cin, L.cout, L.relu, scratch,
                                  p->scratch_bytes, stream);
                if (rc) return rc;
            }
            cur = y;
        }
    }
    if (slabs_live) *slabs_live = live;
    if (tail == 1) return 0;
    if (tail == 2) {
        rc = clhip_internal_fc_tail(&p->chain, params, acts, N, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, nullptr,
                                    base + p->off_rows, p->tail_counter, 0, 0, static_cast<const float*>(scratch), live,
                                    as_stream(stream));
        if (rc) return rc;
        cur = acts + p->layers.back().act_off;
    }
    if (logits_out) {
        hipError_t e = hipMemcpyAsync(logits_out, cur, (size_t)N * p->n_classes * sizeof(float),
                                      hipMemcpyDeviceToDevice, as_stream(stream));
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

int clhip_net_forward(void* handle, const float* params, const float* x, int N, void* ws, float* logits_out,
                      void* stream) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || !params || !x || !ws || N <= 0 || N > p->max_batch) return CLHIP_EINVAL;
    return net_forward_impl(handle, params, x, N, ws, logits_out, stream, tail_usable(p, params, ws, N) ? 2 : 0);
}

// Backward pass from dlogits[N][classes] (device) through the activations saved by the last
// clhip_net_forward on the same ws.  Writes every parameter gradient into `grads` (same offsets
// as params; overwritten, not accumulated).
static int net_backward_impl(void* handle, const float* params, float* grads, const float* x, int N, void* ws,
                             const float* dlogits, void* stream, bool tail_done, bool wino_ready = false) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || !params || !grads || !x || !ws || !dlogits || N <= 0 || N > p->max_batch) return CLHIP_EINVAL;
    char* base = static_cast<char*>(ws);
    float* acts = reinterpret_cast<float*>(base + p->off_acts);
    uint8_t* idx = reinterpret_cast<uint8_t*>(base + p->off_idx);
    float* g[2] = {reinterpret_cast<float*>(base + p->off_g0), reinterpret_cast<float*>(base + p->off_g1)};
    void* scratch = base + p->off_scratch;
    const float* gin = dlogits;      // gradient w.r.t. the current layer's output (already ReLU-masked)
    int gin_buf = -1;                // which of g[0..1] holds gin (-1: dlogits / fcdz)
    int flip = 0;
    int rc;
    const int top = (int)p->layers.size() - 1;
    float* fcdz = reinterpret_cast<float*>(base + p->off_fcdz);
    hipStream_t main_s = as_stream(stream);
    if (p->wino_bytes && !wino_ready) {           // a backward on its own: the forward of the same call did not transform for it
        rc = wino_prepare(p, params, base, false, true, main_s);
        if (rc) return rc;
    }
    const bool ov = p->overlap && p->overlap_mode == 1;          // every layer on the side stream: immediate reductions
    const bool ov_small = p->overlap && p->overlap_mode == 2;    // only the under-filled deep layers; slabs stay deferred
    auto side_ok = [&](int layer) {
        if (ov) return true;
        const LayerPlan& Ls = p->layers[layer];
        return ov_small && Ls.type == 0 && Ls.wg3 && Ls.h * Ls.w <= 256;
    };
    hipEvent_t pending[2] = {nullptr, nullptr};      // side-stream reader of g[b] that must finish before g[b] is rewritten
    hipEvent_t last_side = nullptr;
    int taken = -1;
    const bool defer = p->n_wg > 0 && !ov;
    // layers whose backward-data and weight gradient can go out as one grid: plain 3x3 layers on the Winograd kernels with deferred
    // slabs, a backward-data to compute and nothing to apply to it afterwards (clhip_internal_wino_pair decides on the shape)
    auto pair_ok = [&](const LayerPlan& Lp, int layer) {
        return defer && layer > 0 && Lp.type == 0 && Lp.ks == 3 && Lp.st == 1 && Lp.pd == 1 && !Lp.bn && Lp.wino_w && Lp.wino_d &&
               !Lp.bs_d && !Lp.bs_w && Lp.wg_bytes && !Lp.drop && !Lp.extra_grad && !side_ok(layer);
    };
    clhip_wgrad_job jobs[CLHIP_WGRAD_JOBS_MAX];
    int n_jobs = 0;
    // next ping-pong buffer as an OUTPUT of a main-stream launch
    auto take = [&]() -> float* {
        const int bi = flip;
        flip ^= 1;
        if (pending[bi]) { (void)hipStreamWaitEvent(main_s, pending[bi], 0); pending[bi] = nullptr; }
        taken = bi;
        return g[bi];
    };
    // run `launch(stream)` (a weight-gradient launch reading dy = g[dy_buf]) on the side stream once main has produced dy
    auto on_side = [&](int layer, int dy_buf, auto&& launch) -> int {
        if (!side_ok(layer)) return launch(stream);
        hipError_t e = hipEventRecord(p->ev_dy[layer], main_s);
        if (e == hipSuccess) e = hipStreamWaitEvent(p->side, p->ev_dy[layer], 0);
        if (e != hipSuccess) return (int)e;
        const int r = launch((void*)p->side);
        if (r) return r;
        e = hipEventRecord(p->ev_wg[layer], p->side);
        if (e != hipSuccess) return (int)e;
        last_side = p->ev_wg[layer];
        if (dy_buf >= 0) pending[dy_buf] = p->ev_wg[layer];
        return 0;
    };
    for (int i = top; i >= 0; --i) {
        const LayerPlan& L = p->layers[i];
        // input of layer i = (pooled) output of layer i-1, or the image batch
        const float* xin = x;
        if (i > 0) {
            const LayerPlan& P = p->layers[i - 1];
            xin = acts + ((P.type == 0 && P.pool) ? P.pool_off : P.act_off);
            if (L.drop && L.has_drop_buf) xin = acts + L.drop_off;
        }
        // ReLU mask of this layer's backward-data, (input > 0).  Not needed when the input is the pooled output of a fused
        // conv + ReLU + pool launch: its arg-max bytes mark the windows without a positive maximum (CLHIP_POOL_DEAD) and
        // the un-pooling consumers of the gradient drop them (common.hpp)
        const float* xmask = xin;
        if (i > 0 && L.type == 0 && !L.drop) {
            const LayerPlan& P = p->layers[i - 1];
            if (P.type == 0 && !P.bn && P.ks == 3 && P.st == 1 && P.pd == 1 && P.pool && P.pk == 2 && P.ps == 2 && P.relu) xmask = nullptr;
        }
        if (L.type == 1 && tail_done && i > p->fc_first) {
            // the fused tail has already left dz of this layer's input in its fcdz slot
            gin = fcdz + p->chain.dz_off[i - p->fc_first - 1];
            gin_buf = -1;
            continue;
        }
        if (L.type == 1) {
            if (!p->fc_fused) {
                rc = clhip_fc_bwd_weight(xin, gin, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, scratch, p->scratch_bytes, stream);
                if (rc) return rc;
            }
            bool combo = false;
            clhip_fc_chain ch;
            if (p->fc_fused && i == p->fc_first) {
                ch = p->chain;       // inputs of the hidden layers: their masked copies where a dropout is active
                for (int l = 1; l < ch.n; ++l) {
                    const LayerPlan& Ll = p->layers[p->fc_first + l];
                    if (Ll.drop && Ll.has_drop_buf) ch.act_off[l - 1] = Ll.drop_off;
                }
            }
            if (i > 0) {
                // fused weight gradients (below) read the hidden layers' dz after the chain: keep them in their own slots
                float* gout;
                if (p->fc_fused && i > p->fc_first) { gout = fcdz + p->chain.dz_off[i - p->fc_first - 1]; gin_buf = -1; }
                else { gout = take(); gin_buf = taken; }
                // mask with (xin > 0): xin is a ReLU (or pooled ReLU) output
                clhip_gemm_args ga;
                const int gb = (p->fc_fused && i == p->fc_first && !p->no_combo)
                                   ? clhip_internal_fc_bwd_data_args(gin, params + L.w_off, xin, gout, N, L.cin, L.cout, &ga) : 0;
                if (gb > 0) {
                    // backward-data of the first Linear layer and every Linear layer's dW / db in ONE launch
                    rc = clhip_internal_fc_bwd_combo(&ga, gb, &ch, grads, xin, N, acts, dlogits, fcdz, main_s);
                    combo = true;
                } else {
                    rc = clhip_fc_bwd_data(gin, params + L.w_off, xin, gout, N, L.cin, L.cout, scratch, p->scratch_bytes, stream);
                }
                if (rc) return rc;
                gin = gout;
                if (L.drop) {
                    rc = drop_scale(gout, L.drop, L.drop_stride, L.in_elems, N, main_s);
                    if (rc) return rc;
                }
                if (L.extra_grad) {
                    rc = add_inplace(gout, L.extra_grad, L.in_elems * (size_t)N, main_s);
                    if (rc) return rc;
                }
            }
            if (p->fc_fused && i == p->fc_first && !combo) {
                // all dz_l are in place: dW_l, db_l of every Linear layer in one launch
                rc = clhip_internal_fc_chain_wgrad(&ch, grads, xin, N, acts, dlogits, fcdz, main_s);
                if (rc) return rc;
            }
            continue;
        }
        const bool vgg = L.ks == 3 && L.st == 1 && L.pd == 1;
        const bool pool22 = L.pool && L.pk == 2 && L.ps == 2;
        const float* gy = gin;
        int gy_buf = gin_buf;
        bool wdone = false;
        bool ddone = false;
        // measurement probe around this layer's backward-data / weight-gradient launch(es) (clhip_net_probe_kind)
        auto probe_on = [&](int kind) { return i == p->probe_layer && p->probe_kind == kind && !p->probe_ev.empty(); };
        // (events go on the stream the launch is issued on: a weight-gradient launch may run on the side stream, on_side)
        auto probe_stream = [&](int kind) { return (kind == 2 && side_ok(i)) ? p->side : main_s; };
        auto probe_begin = [&](int kind) {
            if (probe_on(kind)) (void)hipEventRecord(p->probe_ev[2 * (p->probe_count % PROBE_RING)], probe_stream(kind));
        };
        auto probe_end = [&](int kind) {
            if (probe_on(kind)) { (void)hipEventRecord(p->probe_ev[2 * (p->probe_count % PROBE_RING) + 1], probe_stream(kind)); ++p->probe_count; }
        };
        float* gout_d = nullptr;
        int gout_d_buf = -1;
        if (vgg && pool22 && !L.bn && L.wg3) {
            // fused max-pool backward: weight gradient and backward-data both rebuild the un-pooled gradient tile from
            // the POOLED gradient + the arg-max codes while staging it, when their kernels support the shape (first
            // layer: the small-C kernel; others: the 16-byte staging paths).  The 4x larger un-pooled tensor and the
            // clhip_maxpool2_bwd launch disappear.
            if (pair_ok(L, i)) {
                // backward-data and the weight-gradient slabs of this layer as ONE grid (wino.hip, wino_pair_kernel)
                gout_d = take(); gout_d_buf = taken;
                if (p->probe_kind) probe_begin(p->probe_kind);   // (kinds 1 and 2 both time the merged launch; a forward probe does not)
                rc = clhip_internal_wino_pair(gin, idx + L.idx_off, reinterpret_cast<const float*>(base + p->off_wino + L.wino_ud), xmask,
                                              gout_d, xin, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h, L.w,
                                              base + p->off_wg + L.wg_off, L.wg_bytes, main_s, &jobs[n_jobs]);
                if (rc == 0) { wdone = ddone = true; ++n_jobs; if (p->probe_kind) probe_end(p->probe_kind); }
                else if (rc != CLHIP_ENOTSUP) return rc;
            }
            if (!wdone) {
                probe_begin(2);
                rc = on_side(i, gin_buf, [&](void* st) {
                    if (defer && L.bs_w) {
                        const int r = clhip_internal_bs_wgrad_partial(xin, gin, idx + L.idx_off, grads + L.w_off, grads + L.b_off, N, L.cin,
                                                                      L.cout, L.h, L.w, base + p->off_wg + L.wg_off, L.wg_bytes, as_stream(st),
                                                                      &jobs[n_jobs]);
                        if (r != CLHIP_ENOTSUP && r != CLHIP_ENOSPC) return r;
                    }
                    if (defer && L.wino_w) {
                        const int r = clhip_internal_wino_wgrad_partial(xin, gin, idx + L.idx_off, grads + L.w_off, grads + L.b_off, N, L.cin,
                                                                        L.cout, L.h, L.w, base + p->off_wg + L.wg_off, L.wg_bytes,
                                                                        as_stream(st), &jobs[n_jobs]);
                        if (r != CLHIP_ENOTSUP && r != CLHIP_ENOSPC) return r;
                    }
                    if (defer)
                        return clhip_internal_conv3x3_wgrad_partial(xin, gin, idx + L.idx_off, grads + L.w_off, grads + L.b_off, N, L.cin,
                                                                    L.cout, L.h, L.w, base + p->off_wg + L.wg_off, L.wg_bytes, st,
                                                                    &jobs[n_jobs]);
                    return clhip_conv3x3_bwd_weight_unpool(xin, gin, idx + L.idx_off, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout,
                                                           L.h, L.w, scratch, p->scratch_bytes, st);
                });
                if (rc == 0) { wdone = true; if (defer) ++n_jobs; probe_end(2); }
                else if (rc != CLHIP_ENOTSUP) return rc;
            }
            if (wdone && !ddone && i > 0 && !L.drop && !L.extra_grad) {
                if (!gout_d) { gout_d = take(); gout_d_buf = taken; }
                probe_begin(1);
                rc = L.wino_d ? plan_conv_u(L.bs_d, 1, gin, reinterpret_cast<const float*>(base + p->off_wino + L.wino_ud), nullptr,
                                                           xmask, gout_d, idx + L.idx_off, 1, N, L.cout, L.cin, L.h, L.w, 0, as_stream(stream))
                              : clhip_conv3x3_bwd_data_unpool(gin, idx + L.idx_off, params + L.w_off, xmask, gout_d, N, L.cin, L.cout, L.h,
                                                              L.w, stream);
                if (rc == 0) { ddone = true; probe_end(1); }
                else if (rc != CLHIP_ENOTSUP) return rc;
            }
        }
        if (L.pool && !(wdone && (i == 0 || ddone))) {      // somebody still needs the un-pooled gradient
            float* gout = gout_d;
            if (gout) taken = gout_d_buf; else gout = take();
            rc = pool22 ? clhip_maxpool2_bwd(gin, idx + L.idx_off, gout, N * L.cout, L.oh, L.ow, stream)
                        : clhip_maxpool_bwd(gin, idx + L.idx_off, gout, N * L.cout, L.oh, L.ow, L.pk, L.ps, stream);
            if (rc) return rc;
            gy = gout; gy_buf = taken;
            gout_d = nullptr;
        }
        if (L.bn) {
            // dy (w.r.t. the ReLU output) -> dz (w.r.t. the convolution output), in place; dgamma, dbeta
            const float* st = acts + L.stat_off;
            float* dzb = const_cast<float*>(gy);
            if (gy_buf < 0) return CLHIP_EINVAL;      // a conv layer's incoming gradient always lives in g[0..1]
            rc = clhip_bn_bwd(gy, acts + L.act_off, acts + L.z_off, params + L.bn_w_off, st, st + L.cout, dzb, grads + L.bn_w_off,
                              grads + L.bn_b_off, N, L.cout, L.oh * L.ow, p->training, L.relu, scratch, p->scratch_bytes, stream);
            if (rc) return rc;
        }
        if (!wdone && !L.pool && !gout_d && pair_ok(L, i)) {
            gout_d = take(); gout_d_buf = taken;
            if (p->probe_kind) probe_begin(p->probe_kind);   // (kinds 1 and 2 both time the merged launch; a forward probe does not)
            rc = clhip_internal_wino_pair(gy, nullptr, reinterpret_cast<const float*>(base + p->off_wino + L.wino_ud), xmask, gout_d, xin,
                                          grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h, L.w, base + p->off_wg + L.wg_off,
                                          L.wg_bytes, main_s, &jobs[n_jobs]);
            if (rc == 0) { wdone = ddone = true; ++n_jobs; if (p->probe_kind) probe_end(p->probe_kind); }
            else if (rc != CLHIP_ENOTSUP) return rc;
        }
        if (!wdone) {
            bool job = false;                 // this layer left slabs for the deferred reduction
            probe_begin(2);
            rc = on_side(i, gy_buf, [&](void* st) {
                if (defer && L.bs_w && L.wg_bytes) {
                    const int r = clhip_internal_bs_wgrad_partial(xin, gy, nullptr, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h, L.w,
                                                                  base + p->off_wg + L.wg_off, L.wg_bytes, as_stream(st), &jobs[n_jobs]);
                    if (r == 0) job = true;
                    if (r != CLHIP_ENOTSUP && r != CLHIP_ENOSPC) return r;
                }
                if (defer && L.wino_w && L.wg_bytes) {
                    const int r = clhip_internal_wino_wgrad_partial(xin, gy, nullptr, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout,
                                                                    L.h, L.w, base + p->off_wg + L.wg_off, L.wg_bytes, as_stream(st),
                                                                    &jobs[n_jobs]);
                    if (r == 0) job = true;
                    if (r != CLHIP_ENOTSUP && r != CLHIP_ENOSPC) return r;
                }
                if (L.wg3 && defer) {
                    job = true;
                    return clhip_internal_conv3x3_wgrad_partial(xin, gy, nullptr, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h,
                                                                L.w, base + p->off_wg + L.wg_off, L.wg_bytes, st, &jobs[n_jobs]);
                }
                if (L.bs5_w) {
                    const int r = clhip_internal_bs5_wgrad(xin, gy, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h, L.w, scratch,
                                                           p->scratch_bytes, as_stream(st));
                    if (r != CLHIP_ENOTSUP && r != CLHIP_ENOSPC) return r;
                }
                if (L.s2d) {                  // (the phase planes of this pass's forward are still in the layer's frames)
                    const bool fresh = p->s2d_fresh_n == N;
                    p->s2d_fresh_n = 0;
                    return clhip_conv2d_s2d_bwd_weight(fresh ? nullptr : xin, gy, grads + L.w_off, grads + L.b_off, N, L.cin, L.h, L.w, L.cout,
                                                       L.ks, L.st, L.pd, base + p->off_s2d + L.s2d_off, L.s2d_bytes, st);
                }
                return L.wg3 ? clhip_conv3x3_bwd_weight(xin, gy, grads + L.w_off, grads + L.b_off, N, L.cin, L.cout, L.h, L.w, scratch,
                                                      p->scratch_bytes, st)
                           : clhip_conv2d_bwd_weight(xin, gy, grads + L.w_off, grads + L.b_off, N, L.cin, L.h, L.w, L.cout, L.ks, L.ks,
                                                     L.st, L.pd, scratch, p->scratch_bytes, st);
            });
            if (rc) return rc;
            probe_end(2);
            if (job) ++n_jobs;
        }
        if (i > 0 && ddone) {
            gin = gout_d; gin_buf = gout_d_buf;
        } else if (i > 0) {
            float* gout = gout_d;             // (taken for a merged launch that declined the shape)
            if (gout) taken = gout_d_buf; else gout = take();
            probe_begin(1);
            rc = (vgg && L.wino_d)
                     ? plan_conv_u(L.bs_d, 1, gy, reinterpret_cast<const float*>(base + p->off_wino + L.wino_ud), nullptr, xmask, gout,
                                                  nullptr, 0, N, L.cout, L.cin, L.h, L.w, 0, as_stream(stream))
                 : vgg ? clhip_conv3x3_bwd_data(gy, params + L.w_off, xmask, gout, N, L.cin, L.cout, L.h, L.w, stream)
                 : L.bs5_d ? clhip_internal_bs5_conv_u(1, gy, base + p->off_wino + L.wino_ud, nullptr, xmask, gout, N, L.cout, L.cin, L.h, L.w,
                                                       0, as_stream(stream))
                     : clhip_conv2d_bwd_data(gy, params + L.w_off, xmask, gout, N, L.cin, L.h, L.w, L.cout, L.ks, L.ks, L.st, L.pd, stream);
            if (rc) return rc;
            probe_end(1);
            gin = gout; gin_buf = taken;
            if (L.drop) {
                rc = drop_scale(gout, L.drop, L.drop_stride, L.in_elems, N, main_s);
                if (rc) return rc;
            }
            if (L.extra_grad) {
                rc = add_inplace(gout, L.extra_grad, L.in_elems * (size_t)N, main_s);
                if (rc) return rc;
            }
        }
    }
    if (last_side) {      // join: every slab / gradient of the side stream is complete on the caller's stream from here on
        hipError_t e = hipStreamWaitEvent(main_s, last_side, 0);
        if (e != hipSuccess) return (int)e;
    }
    if (n_jobs) {
        rc = clhip_internal_wgrad_reduce_multi(jobs, n_jobs, main_s);
        if (rc) return rc;
    }
    return 0;
}

int clhip_net_backward(void* handle, const float* params, float* grads, const float* x, int N, void* ws,
                       const float* dlogits, void* stream) {
    return net_backward_impl(handle, params, grads, x, N, ws, dlogits, stream, false);
}

// forward + loss (+ backward when grads != NULL) in one call.
//   loss_kind 0: CrossEntropy mean   1: CrossEntropy sum   2: sum of squared logits (MAS)
int clhip_net_loss_step_slice(void* handle, const float* params, float* grads, const float* x, const int64_t* labels,
                              int N, int loss_kind, int col_off, int ncols, void* ws, float* loss_out, double* stats,
                              float* logits_out, void* stream) {
    NetPlan* p = static_cast<NetPlan*>(handle);
    if (!p || !ws || (loss_kind != 2 && !labels)) return CLHIP_EINVAL;
    char* base = static_cast<char*>(ws);
    float* dlogits = reinterpret_cast<float*>(base + p->off_dlogits);
    float* loss_dev = loss_out ? loss_out : reinterpret_cast<float*>(base + p->off_loss);
    if (!params || !x || N <= 0 || N > p->max_batch) return CLHIP_EINVAL;
    const int nc = ncols > 0 ? ncols : p->n_classes - col_off;
    if (loss_kind != 2 && nc <= 64 && col_off >= 0 && col_off + nc <= p->n_classes && tail_usable(p, params, ws, N)) {
        // first Linear layer by the GEMM launches, then ONE launch for the rest of the classifier, the loss and the
        // backward-data chain down to dz(h1); backward resumes at the first Linear layer
        if (loss_kind != 0 && loss_kind != 1) return CLHIP_EINVAL;
        int live = 0;
        int rc = net_forward_impl(handle, params, x, N, ws, nullptr, stream, 1, &live, grads != nullptr);
        if (rc) return rc;
        float* acts = reinterpret_cast<float*>(base + p->off_acts);
        rc = clhip_internal_fc_tail(&p->chain, params, acts, N, labels, loss_kind, col_off, nc, dlogits,
                                    reinterpret_cast<float*>(base + p->off_fcdz), loss_dev, stats, base + p->off_rows,
                                    p->tail_counter, 1, grads ? 1 : 0, reinterpret_cast<const float*>(base + p->off_scratch), live,
                                    as_stream(stream));
        if (rc) return rc;
        if (logits_out) {
            hipError_t e = hipMemcpyAsync(logits_out, acts + p->layers.back().act_off, (size_t)N * p->n_classes * sizeof(float),
                                          hipMemcpyDeviceToDevice, as_stream(stream));
            if (e != hipSuccess) return (int)e;
        }
        if (grads) rc = net_backward_impl(handle, params, grads, x, N, ws, dlogits, stream, true, true);
        return rc;
    }
    int rc = net_forward_impl(handle, params, x, N, ws, logits_out, stream, 0, nullptr, grads != nullptr);
    if (rc) return rc;
    const LayerPlan& last = p->layers.back();
    const float* logits = reinterpret_cast<float*>(base + p->off_acts) + last.act_off;
    if (loss_kind == 2) rc = clhip_mse_zero_sum(logits, (size_t)N * p->n_classes, dlogits, loss_dev, stream);
    else rc = clhip_softmax_ce_slice(logits, labels, N, p->n_classes, col_off, ncols > 0 ? ncols : p->n_classes - col_off,
                                     loss_kind, dlogits, loss_dev, stats, stream);
    if (rc) return rc;
    if (grads) rc = net_backward_impl(handle, params, grads, x, N, ws, dlogits, stream, false, true);
    return rc;
}

int clhip_net_loss_step(void* handle, const float* params, float* grads, const float* x, const int64_t* labels,
                        int N, int loss_kind, void* ws, float* loss_out, double* stats, float* logits_out,
                        void* stream) {
    return clhip_net_loss_step_slice(handle, params, grads, x, labels, N, loss_kind, 0, 0, ws, loss_out, stats,
                                     logits_out, stream);
}

}  // extern "C"
