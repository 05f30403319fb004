/*
 * clhip.h — C ABI of libclhip.so: MI355X (gfx950 / CDNA4) kernels for the CLsurvey
 * per-task training + importance-weight hot path.
 *
 * The reference (Mattdl/CLsurvey) is pure Python on torch; every "kernel" it runs is an ATen
 * op reached from the files cited per entry point below (paths relative to /root/reference/src).
 * A maintainer binds this header with ctypes; no torch types cross it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (fp32 unless the name says _u8 / _i64 / _f64)
 *   - tensors are dense NCHW / row-major exactly as torch lays them out (weights [K][C][3][3])
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *   - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *     CLHIP_E* argument error. Kernels never allocate; scratch comes in through `ws`.
 */
#ifndef CLHIP_H
#define CLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the prototypes of this header are its whole dynamic symbol table (the
 * cross-translation-unit helpers of csrc/common.hpp stay inside the .so). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#define CLHIP_VISIBILITY_PUSHED 1
#endif

#define CLHIP_EINVAL (-1)   /* bad shape / null pointer */
#define CLHIP_ENOSPC (-2)   /* workspace too small */
#define CLHIP_ENOTSUP (-3)  /* shape not supported by this build */

/* Library version (major*10000 + minor*100 + patch) and the gfx arch it was built for. */
/* EBLL (methods/EBLL): the code layer of AutoEncoder (AlexNet_EBLL.py:9-26: Linear + Sigmoid), nn.MSELoss() on codes /
 * reconstructions (Finetune_SGD_EBLL.py:151, 297) with its gradient (times grad_scale) and optim.Adadelta (:497). */
int clhip_sigmoid_fwd(const float* x, float* y, size_t n, void* stream);
int clhip_sigmoid_bwd(const float* dy, const float* y, float* dx, size_t n, void* stream);
int clhip_mse_mean(const float* a, const float* b, size_t n, float grad_scale, float* da, float* loss_out, void* stream);
int clhip_adadelta_step(float* theta, const float* grad, float* square_avg, float* acc_delta, size_t n, float lr, float rho,
                        float eps, float weight_decay, void* stream);

int clhip_version(void);
const char* clhip_arch(void);

/* ------------------------------------------------------------------ convolution (MFMA fp32)
 * nn.Conv2d(C, K, 3, padding=1) (+ fused bias, ReLU) — models/VGGSlim.py:34-38.
 * y[N][K][H][W] = relu?(conv(x[N][C][H][W], w[K][C][3][3]) + b[K])                      */
int clhip_conv3x3_fwd(const float* x, const float* w, const float* b, float* y,
                      int N, int C, int K, int H, int W, int relu, void* stream);

/* Fused conv + bias + ReLU + 2x2/2 max-pool (VGGSlim.py:32-38): y_pool[N][K][H/2][W/2], idx_u8 = argmax
 * (0..3), or 4 for a window whose maximum after ReLU is not positive: max_pool2d backward followed by ReLU backward
 * (VGGSlim.py:32,38) passes nothing through such a window, so every consumer of the codes (`code == position`) applies
 * both, and backward-data of the NEXT layer may be called with relu_src = NULL.  The pre-pool activation is never
 * written (the first layer's would be 210 MB per batch).                                                    */
int clhip_conv3x3_relu_pool_fwd(const float* x, const float* w, const float* b, float* y_pool, uint8_t* idx_u8,
                                int N, int C, int K, int H, int W, void* stream);

/* autograd convolution_backward, data part. dx[N][C][H][W] from dy[N][K][H][W].
 * If relu_src != NULL: dx is multiplied by (relu_src > 0) — the fused ReLU backward
 * (threshold_backward) of the layer that PRODUCED this conv's input (VGGSlim.py:38).      */
int clhip_conv3x3_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx,
                           int N, int C, int K, int H, int W, void* stream);

/* convolution_backward, weight + bias part. dw[K][C][3][3], db[K] (db may be NULL).
 * Deterministic: split partial sums live in `ws` (>= clhip_conv3x3_bwd_weight_ws bytes)
 * and are reduced in a fixed order (utils.set_random contract, utilities/utils.py:52-58).  */
size_t clhip_conv3x3_bwd_weight_ws(int N, int C, int K, int H, int W);
int clhip_conv3x3_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                             int N, int C, int K, int H, int W, void* ws, size_t ws_bytes,
                             void* stream);

/* Backward-data of a conv whose ReLU output was 2x2-max-pooled, from the gradient w.r.t. the POOLED output + arg-max
 * codes (fused max_pool2d backward, as clhip_conv3x3_bwd_weight_unpool below).  16-byte-staging shapes only (K % 8 == 0,
 * C % 4 == 0, W % 4 == 0, whole tiles along w, aligned pointers): CLHIP_ENOTSUP otherwise.  relu_src as in
 * clhip_conv3x3_bwd_data.                                                                                      */
int clhip_conv3x3_bwd_data_unpool(const float* dy_pool, const uint8_t* idx_u8, const float* w, const float* relu_src,
                                  float* dx, int N, int C, int K, int H, int W, void* stream);

/* The same computation in two calls, for callers that defer the reduction (the plan executor reduces the slabs of
 * every layer of a backward pass in ONE launch): `_slabs` writes the per-split partial sums into ws and reports their
 * count, `_reduce` adds them in the fixed order of clhip_conv3x3_bwd_weight (bitwise the same dw / db).
 * idx_u8_or_null != NULL: dy is the POOLED gradient (see clhip_conv3x3_bwd_weight_unpool).                */
int clhip_conv3x3_bwd_weight_slabs(const float* x, const float* dy, const uint8_t* idx_u8_or_null, int N, int C, int K,
                                   int H, int W, void* ws, size_t ws_bytes, int* splits_out, void* stream);
int clhip_conv3x3_bwd_weight_reduce(const void* ws, float* dw, float* db, int K, int C, int splits, void* stream);

/* Same from the gradient w.r.t. the POOLED output + argmax (fused max_pool2d backward).  Implemented for the first-layer
 * kernel (C*9 <= 32) and for the 16-byte staging path of the general kernel (W % 4 == 0, whole tiles along w, aligned
 * pointers); returns CLHIP_ENOTSUP otherwise (callers then un-pool with clhip_maxpool2_bwd first).          */
int clhip_conv3x3_bwd_weight_unpool(const float* x, const float* dy_pool, const uint8_t* idx_u8, float* dw, float* db,
                                    int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ max-pool 2x2 stride 2
 * nn.MaxPool2d(2, 2) — models/VGGSlim.py:32. idx_u8 holds the argmax (0..3, row-major in the
 * window, first maximum wins as in ATen).  H, W are the INPUT sizes (even).  The backward kernel
 * also takes the fused kernels' code 4 ("no positive maximum": the window gets no gradient).        */
int clhip_maxpool2_fwd(const float* x, float* y, uint8_t* idx_u8, int NC, int H, int W, void* stream);
int clhip_maxpool2_bwd(const float* dy, const uint8_t* idx_u8, float* dx, int NC, int H, int W,
                       void* stream);

/* General k x k / stride max-pool, no padding (torchvision alexnet features[2,5,12]: MaxPool2d(3, 2), models/net.py:96-125).
 * idx = window position r*k + c of the first maximum (ATen scan order); backward gathers over the overlapping windows. */
int clhip_maxpool_fwd(const float* x, float* y, uint8_t* idx_u8, int NC, int H, int W, int k, int stride, void* stream);
int clhip_maxpool_bwd(const float* dy, const uint8_t* idx_u8, float* dx, int NC, int H, int W, int k, int stride, void* stream);

/* nn.BatchNorm2d (+ the ReLU behind it) of the '_BN' model variants (models/VGGSlim.py:27-40, models/net.py:152-156),
 * NCHW, HW = H*W.  training != 0: batch mean / biased variance, running_mean / running_var (may be NULL) moved by
 * `momentum` with the unbiased variance, as torch; training == 0: the running statistics.  save_mean / save_invstd [C]
 * are what the backward needs.  ws: clhip_bn_ws(C) bytes.  Backward: dy is the gradient w.r.t. y (post-ReLU when relu),
 * dz may alias dy; dgamma / dbeta (may be NULL) are overwritten. */
size_t clhip_bn_ws(int C);
int clhip_bn_fwd(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y,
                 float* save_mean, float* save_invstd, int N, int C, int HW, int training, float momentum, float eps, int relu,
                 void* ws, size_t ws_bytes, void* stream);
int clhip_bn_bwd(const float* dy, const float* y, const float* z, const float* gamma, const float* save_mean,
                 const float* save_invstd, float* dz, float* dgamma, float* dbeta, int N, int C, int HW, int training, int relu,
                 void* ws, size_t ws_bytes, void* stream);

/* General convolution (AlexNet: Conv2d(3,64,11,4,2), Conv2d(64,192,5,1,2), 3x3 pad 1 with 192/384/256 channels —
 * models/net.py:96-125 via torchvision.models.alexnet), NCHW / KCRS, zero padding `pad`, stride `stride`:
 * forward (+bias, optional ReLU), backward-data (optional ReLU mask relu_src > 0), backward-weight (+ db) with
 * caller-provided workspace of clhip_conv2d_bwd_weight_ws bytes. CLHIP_ENOTSUP if R*S > 256. */
size_t clhip_conv2d_bwd_weight_ws(int N, int C, int H, int W, int K, int R, int S, int stride, int pad);
int clhip_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R, int S,
                     int stride, int pad, int relu, void* stream);
int clhip_conv2d_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int H, int W, int K,
                          int R, int S, int stride, int pad, void* stream);
int clhip_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int H, int W, int K, int R,
                            int S, int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* Weight / bias gradient of the 3x3 layer on the bf16 matrix cores with exact 3-piece fp32 splits of BOTH operands (csrc/bswgrad.hip):
 * the operator of clhip_conv3x3_wino_bwd_weight (same arguments; idx_u8 != NULL: dy is the pooled gradient [N][K][H/2][W/2] + the
 * forward pass's arg-max codes), for C % 64 == 0, K % 64 == 0, W % 16 == 0; other maps from 8 pixels wide (13 x 13!) with C % 32 == 0,
 * K % 32 == 0 and a plain dy run on the tap-split kernel of csrc/bswgrad5.hip (CLHIP_ENOTSUP otherwise).  ws:
 * clhip_conv3x3_bs_bwd_weight_ws(N, C, K, H, W) bytes of slabs, reduced in a fixed order (bitwise deterministic). */
size_t clhip_conv3x3_bs_bwd_weight_ws(int N, int C, int K, int H, int W);
int clhip_conv3x3_bs_bwd_weight(const float* x, const float* dy, const uint8_t* idx_u8_or_null, float* dw, float* db, int N, int C, int K,
                                int H, int W, void* ws, size_t ws_bytes, void* stream);

/* Weight / bias gradient of the 5x5 / padding-2 convolution (AlexNet's Conv2d(64, 192, 5, padding=2), models/net.py:96-125) on the bf16
 * matrix cores with exact 3-piece fp32 splits of both operands (csrc/bswgrad5.hip): the operator of clhip_conv2d_bwd_weight(R = S = 5,
 * stride 1, pad 2), for C % 32 == 0, K % 32 == 0, W >= 16, tensors < 2 GB (CLHIP_ENOTSUP otherwise).  ws:
 * clhip_conv5x5_bs_bwd_weight_ws(N, C, K, H, W) bytes of slabs, reduced in a fixed order (bitwise deterministic). */
size_t clhip_conv5x5_bs_bwd_weight_ws(int N, int C, int K, int H, int W);
int clhip_conv5x5_bs_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int C, int K, int H, int W, void* ws,
                                size_t ws_bytes, void* stream);

/* The same strided convolution by space-to-depth (csrc/s2dconv.hip): stride s in {2, 4}, 2 s < R <= 3 s, C s^2 <= 64, K % 64 == 0 —
 * AlexNet's Conv2d(3, 64, 11, 4, 2), models/net.py:96-125 — becomes a dense 3x3 convolution over the C s^2 phase planes of the
 * padded input and runs on clhip_conv3x3_bs_fwd / clhip_conv3x3_wino_bwd_weight over frames kept in `ws`
 * (clhip_conv2d_s2d_ws bytes; 0 = shape not taken, the entry points then return CLHIP_ENOTSUP).  Same results as
 * clhip_conv2d_fwd / clhip_conv2d_bwd_weight up to fp32 summation order.  bwd_weight with x == NULL reuses the phase planes the
 * forward call left in the same ws (same batch); there is no backward-data (the layer reads the images). */
size_t clhip_conv2d_s2d_ws(int N, int C, int H, int W, int K, int R, int stride, int pad);
int clhip_conv2d_s2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int H, int W, int K, int R, int stride,
                         int pad, int relu, void* ws, size_t ws_bytes, void* stream);
int clhip_conv2d_s2d_bwd_weight(const float* x_or_null, const float* dy, float* dw, float* db, int N, int C, int H, int W, int K, int R,
                                int stride, int pad, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ fully connected (MFMA fp32)
 * nn.Linear — models/VGGSlim.py:68-74.  x[M][I], w[O][I], b[O], y[M][O].                   */
/* ws: optional split-K scratch (>= clhip_fc_ws(M,I,O) bytes); NULL => no K split (slower, same result
 * up to summation order).                                                                   */
size_t clhip_fc_ws(int M, int I, int O);
int clhip_fc_fwd(const float* x, const float* w, const float* b, float* y,
                 int M, int I, int O, int relu, void* ws, size_t ws_bytes, void* stream);
/* dx[M][I] = dy[M][O] . w[O][I], optionally masked by (relu_src[M][I] > 0). */
int clhip_fc_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx,
                      int M, int I, int O, void* ws, size_t ws_bytes, void* stream);
/* dw[O][I] = dy^T . x ; db[O] = column sums of dy (db may be NULL). */
int clhip_fc_bwd_weight(const float* x, const float* dy, float* dw, float* db,
                        int M, int I, int O, void* ws, size_t ws_bytes, void* stream);
/* y = relu(x) backward helper for non-fused callers: dx = dy * (y > 0). */
int clhip_relu_bwd(const float* dy, const float* y, float* dx, size_t n, void* stream);

/* ------------------------------------------------------------------ losses
 * CrossEntropyLoss (mean, reduction=0) — EWC/train_EWC.py:183; nll_loss(log_softmax, sum,
 * reduction=1) — EWC/main_EWC.py:148.  Writes loss_out[0], dlogits[N][C] (already scaled by
 * 1/N for mean) and, if stats != NULL, accumulates stats[0] += loss, stats[1] += #correct
 * (argmax == label; the running_loss / running_corrects of train_EWC.py:196-197) so the
 * host needs no per-batch .item() sync.  C <= 1024.                                         */
int clhip_softmax_ce(const float* logits, const int64_t* labels_i64, int N, int C, int reduction,
                     float* dlogits, float* loss_out, double* stats, void* stream);
/* Same over a column slice [col_off, col_off+ncols) of logits[N][ld] (labels relative to the slice; dlogits
 * outside the slice = 0): GEM's shared 200-way head, rehearsal/model/gem.py:259-263.          */
int clhip_softmax_ce_slice(const float* logits, const int64_t* labels_i64, int N, int ld, int col_off, int ncols,
                           int reduction, float* dlogits, float* loss_out, double* stats, void* stream);
/* MSELoss(size_average=False) against zeros — MAS/train_MAS.py:556-560: loss = sum(z^2),
 * dlogits = 2 z.                                                                            */
int clhip_mse_zero_sum(const float* logits, size_t n, float* dlogits, float* loss_out, void* stream);

/* LwF objective over stacked heads — LwF/main_LWF.py:47-76 (distillation_loss) and :184-202 (train_model_lwf):
 * logits [N][ld] = n_heads heads side by side (head_sizes, host array); last head = new task, CrossEntropy(mean);
 * every earlier head is distilled (temperature T) towards teacher [N][ld_teacher] (same column layout).
 * dlogits = d(task + reg_lambda * sum dist)/dlogits; loss_out2[0] = task loss, [1] = reg_lambda * sum of the
 * distillation terms; stats (optional, f64[2]) += (task loss, #correct on the new head). distill = 0: validation
 * (task loss / accuracy only, zero gradient on the old heads). N <= 1024.                                      */
int clhip_lwf_loss(const float* logits, const int64_t* labels_i64, const float* teacher, const int* head_sizes, int n_heads,
                   int N, int ld, int ld_teacher, float T, float reg_lambda, int distill, float* dlogits, float* loss_out2,
                   double* stats, void* stream);

/* ------------------------------------------------------------------ penalised optimizers
 * Weight_Regularized_SGD.step — EWC/train_EWC.py:23-86 == MAS/train_MAS.py:32-95.
 *   d = g + 2*lambda*omega*(theta - init);  d += wd*theta;
 *   buf = first ? d : momentum*buf + d;  theta -= lr*buf
 * omega == NULL  <=>  "p not in reg_params" (plain SGD, e.g. the fresh head).               */
int clhip_reg_sgd_step(float* theta, const float* grad, const float* omega, const float* init_val,
                       float* buf, size_t n, float reg_lambda, float lr, float momentum, float wd,
                       int first, void* stream);
/* diag_fisher accumulation — EWC/main_EWC.py:155: omega += grad^2 / data_len.               */
int clhip_fisher_accum(float* omega, const float* grad, size_t n, float data_len, void* stream);
/* Objective_After_SGD.step — MAS/train_MAS.py:167-173: omega = (omega*prev + |g|) / curr.   */
int clhip_mas_accum(float* omega, const float* grad, size_t n, float prev_size, float curr_size,
                    void* stream);
/* Elastic_SGD.step — SI/train_SI.py:28-126 (penalised momentum SGD + path integral w).      */
int clhip_si_step(float* theta, const float* grad, const float* omega, const float* init_val,
                  float* w, float* buf, size_t n, float reg_lambda, float lr, float momentum,
                  float wd, int first, void* stream);
/* update_reg_params — SI/train_SI.py:301-351: omega += max(w/((theta-init)^2+slack),0);
 * w = 0; init = theta.                                                                      */
int clhip_si_consolidate(float* omega, float* w, const float* theta, float* init_val, size_t n,
                         float slack, void* stream);

/* IMM_merge_models — IMM/merge.py:185-242, one parameter tensor of n_models (<= 32) task models:
 * precisions == NULL: mean-IMM  out = (sum_m theta_m) / n_models
 * else               mode-IMM  out = sum_m (precisions[m] / sum_precision) * theta_m
 * thetas / precisions are HOST arrays of device pointers.                                           */
int clhip_imm_merge(const float* const* thetas, const float* const* precisions, const float* sum_precision,
                    int n_models, size_t n, float* out, void* stream);

/* ------------------------------------------------------------------ PackNet masks (uint8, bit-exact)
 * methods/packnet/prune.py: mask value = owning task (1-based), 0 = free/pruned.
 *   finetune_mask  (:141-155)  mask[mask==0] = cur
 *   kth_abs        (:30-39)    cutoff = k-th smallest |w| over {mask==cur} (exact radix select;
 *                               k = round(prune_perc*numel) computed by the caller, k >= 1)
 *   prune          (:43-47,:64-71) mask[(|w|<=cutoff)&(mask==cur)] = 0 ; w[mask==0] = 0
 *   mask_grad_zero (:73-97)    grad[mask != cur] = 0
 *   mask_weight_zero mode 0 (:99-106 make_pruned_zero): w[mask==0]=0 ;
 *                    mode 1 (:108-118 apply_mask(idx)): w[mask==0 or mask>idx]=0
 *   packnet_sgd_step           fused do_batch tail (packnet/main.py:187-193): grads of foreign weights
 *                               -> 0, PacknetSGD.step (packnetSGD.py:35-56: weight decay masked by
 *                               grad != 0), pruned weights -> 0.  mask_u8 == NULL => plain PacknetSGD. */
int clhip_packnet_finetune_mask(uint8_t* mask_u8, size_t n, int cur, void* stream);
size_t clhip_packnet_kth_ws(void);
int clhip_packnet_kth_abs(const float* w, const uint8_t* mask_u8, size_t n, int cur, size_t k,
                          float* out_cutoff, void* ws, size_t ws_bytes, void* stream);
int clhip_packnet_prune(float* w, uint8_t* mask_u8, size_t n, int cur, const float* cutoff_dev, void* stream);
int clhip_mask_grad_zero(float* grad, const uint8_t* mask_u8, size_t n, int cur, void* stream);
int clhip_mask_weight_zero(float* w, const uint8_t* mask_u8, size_t n, int mode, int idx, void* stream);
int clhip_packnet_sgd_step(float* theta, float* grad, float* buf, const uint8_t* mask_u8, size_t n, int cur,
                           float lr, float momentum, float wd, int first, void* stream);

/* ------------------------------------------------------------------ 3x3 convolution by Winograd F(2x2, 3x3)
 * The same operators as clhip_conv3x3_fwd / _relu_pool_fwd / _bwd_data / _bwd_data_unpool (VGGSlim.py:27-40 and their
 * autograd backward) with 2.25x fewer matrix instructions per pixel: input / weight / output transforms fused into ONE
 * kernel (csrc/wino.hip), fp32, results equal to the direct kernels' up to rounding (~1e-6 of the output scale).
 * Shapes: C % 8 == 0, C >= 16, K % 32 == 0, H and W even (CLHIP_ENOTSUP otherwise: callers fall back to the direct
 * kernels).  idx_u8 != NULL on the forward: y is the 2x2-max-pooled activation [N][K][H/2][W/2] + arg-max codes;
 * idx_u8 != NULL on backward-data: dy is the gradient w.r.t. that pooled activation (fused max_pool2d backward).
 * ws: clhip_conv3x3_wino_ws(C, K) bytes (the transformed weights of this call).                              */
size_t clhip_conv3x3_wino_ws(int C, int K);
int clhip_conv3x3_wino_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx_u8_or_null, int N, int C, int K,
                           int H, int W, int relu, void* ws, size_t ws_bytes, void* stream);
int clhip_conv3x3_wino_bwd_data(const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src, float* dx,
                                int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* dW, db of the same convolution through the Winograd weight-gradient form G^T[(A dY A^T).*(B^T d B)]G (C % 64 == 0,
 * K % 64 == 0, even H, W; CLHIP_ENOTSUP otherwise).  ws: clhip_conv3x3_wino_bwd_weight_ws(N, C, K, H, W) bytes — partial slabs in
 * the format of clhip_conv3x3_bwd_weight_slabs, summed in a fixed order (deterministic).  idx_u8 != NULL: dy is the POOLED gradient. */
size_t clhip_conv3x3_wino_bwd_weight_ws(int N, int C, int K, int H, int W);
int clhip_conv3x3_wino_bwd_weight(const float* x, const float* dy, const uint8_t* idx_u8_or_null, float* dw, float* db, int N, int C,
                                  int K, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* The whole backward of one such layer — dx (clhip_conv3x3_wino_bwd_data) and dW, db (clhip_conv3x3_wino_bwd_weight) — as ONE
 * grid: the blocks of the two launches interleaved so that blocks in their memory phases share the CUs with blocks in their matrix
 * phases (the autograd backward of one conv of VGGSlim.py:27-40; results bit-identical to the two entry points above wherever they
 * run the same kernels — every layer the VGG9 plans merge, asserted per shape in tests/test_gpu_pair.py; on 8 x 8 maps with
 * N * ceil(C / 32) > 512 and few units the stand-alone backward-data launch is the two-wave-shape instance of wino_conv16_kernel and the
 * agreement is up to fp32 summation order).  Taken for the
 * layers whose launches fill one round of blocks or less: even maps of at most 256 pixels, 16 or more wide or 8 x 8, weight gradient
 * on the pixel-split kernel, 16-byte-aligned tensors; CLHIP_ENOTSUP otherwise (nothing launched: call the two entry points).
 * ws: clhip_conv3x3_wino_bwd_ws(N, C, K, H, W) bytes.                                                                             */
size_t clhip_conv3x3_wino_bwd_ws(int N, int C, int K, int H, int W);
int clhip_conv3x3_wino_bwd(const float* x, const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src,
                           float* dx, float* dw, float* db, int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ 3x3 convolution on the bf16 matrix cores, split fp32 operands
 * The same operators again (VGGSlim.py:27-40 and their autograd backward w.r.t. the input; arguments as the _wino_ entry points)
 * in DIRECT form on v_mfma_f32_32x32x16_bf16: every fp32 operand is split exactly into three bf16 pieces, the six products
 * a_i * b_j (i + j <= 2) of each 16-deep k-step are accumulated in fp32 inside the matrix core (csrc/bsconv.hip).  Measured error
 * against fp64 = that of an fp32 fmaf chain (profiles/r05_bf16_split_dot.txt); results equal the other paths' up to fp32 rounding.
 * Non-finite inputs: an operand that is +-Inf, or so large that its leading bf16 piece rounds to Inf (|v| > 3.39e38), splits into
 * Inf + NaN residuals, so such an output is NaN where the f32 kernels give +-Inf; either way the loss is non-finite and the trainers
 * stop the attempt (train_EWC.py:204-205).
 * Shapes: C % 32 == 0, K % 64 == 0 on the forward (K % 32, C % 64 on backward-data), H, W >= 4; fused pooling on even maps only;
 * CLHIP_ENOTSUP otherwise.  ws: clhip_conv3x3_bs_ws(C, K) bytes (the weight image of this call).                            */
size_t clhip_conv3x3_bs_ws(int C, int K);
int clhip_conv3x3_bs_fwd(const float* x, const float* w, const float* b, float* y, uint8_t* idx_u8_or_null, int N, int C, int K,
                         int H, int W, int relu, void* ws, size_t ws_bytes, void* stream);
int clhip_conv3x3_bs_bwd_data(const float* dy, const uint8_t* idx_u8_or_null, const float* w, const float* relu_src, float* dx,
                              int N, int C, int K, int H, int W, void* ws, size_t ws_bytes, void* stream);
/* The same kernel with 5 x 5 taps (stride 1, padding 2): torchvision AlexNet's features[3], nn.Conv2d(64, 192, 5, padding=2) + ReLU
 * (models/net.py:96-125) and its autograd w.r.t. the input (dx *= (relu_src > 0) when relu_src != NULL).  w: [K][C][5][5].
 * C % 32 == 0, K % 64 == 0 on the forward (roles swapped on backward-data), W > 8.  ws: clhip_conv5x5_bs_ws(C, K) bytes.      */
size_t clhip_conv5x5_bs_ws(int C, int K);
int clhip_conv5x5_bs_fwd(const float* x, const float* w, const float* b, float* y, int N, int C, int K, int H, int W, int relu,
                         void* ws, size_t ws_bytes, void* stream);
int clhip_conv5x5_bs_bwd_data(const float* dy, const float* w, const float* relu_src, float* dx, int N, int C, int K, int H, int W,
                              void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------ HAT gates / back-masks / HAT_SGD
 * methods/HAT/networks/vgg_hat.py, approaches/hat.py, HAT_utils.py.  Gates multiply layer outputs in the
 * reference (vgg_hat.py:104-116); here they are folded into the NEXT layer's weights
 * (W'[k][c][r] = W[k][c][r]*gate[c]) so the conv/FC kernels run unchanged on un-gated activations.
 *   hat_gate        (:121-127)  gate = sigmoid(s * emb_row)
 *   hat_scale_weight            out = w * gate_in[c]   (w viewed as [K][C][R]; gate_in NULL => copy)
 *   hat_weight_grad             dw = g_wprime * gate_in[c] ; dgate_in[c] = sum_{k,r} g_wprime * w
 *   hat_emb_grad                demb = (dgate + lamb_over_count*(1-mask_pre)) * s*a*(1-a)   (hat.py:285-299)
 *   hat_reg_sums                sums2[0] += sum gate*(1-mask_pre); sums2[1] += sum (1-mask_pre)
 *   hat_backmask    (:258-295)  out[k][c][r] = 1 - min(a_post[k], a_pre[c])   (a_pre NULL => 1 - a_post[k])
 *   hat_sgd_step    (HAT_utils.py:192-250) wd (not on embs), grad *= mask_back, embedding compensation
 *                   (smax/s)*(cosh(clamp(s*e,+-thres))+1)/(cosh(e)+1), clip_grad_norm(p, clipgrad), momentum SGD
 *   clamp           (hat.py:238-240) embeddings to +-6                                               */
int clhip_hat_gate(const float* emb_row, int n, float s, float* gate, void* stream);
int clhip_hat_scale_weight(const float* w, const float* gate_in, float* out, size_t K, size_t C, size_t R, void* stream);
int clhip_hat_weight_grad(const float* g_wprime, const float* w, const float* gate_in, float* dw, float* dgate_in,
                          int K, int C, int R, void* stream);
int clhip_hat_emb_grad(const float* dgate, const float* gate, const float* mask_pre, int n, float s,
                       float lamb_over_count, float* demb, void* stream);
int clhip_hat_reg_sums(const float* gate, const float* mask_pre, int n, double* sums2, void* stream);
int clhip_hat_backmask(const float* a_post, const float* a_pre, float* out, size_t K, size_t C, size_t R, void* stream);
size_t clhip_hat_sgd_ws(void);
int clhip_hat_sgd_step(float* theta, float* grad, float* buf, const float* mask_back, size_t n, float lr,
                       float momentum, float wd, int is_emb, int finetune, float s, float smax, float thres_cosh,
                       float clipgrad, int first, void* ws, size_t ws_bytes, void* stream);
int clhip_clamp(float* x, size_t n, float lo, float hi, void* stream);

/* The same arithmetic for a whole net per launch (job tables on the host, copied into the kernel arguments): one HAT
 * training batch is gates + regulariser sums (1 launch), W' for every layer (1), [the net's own plan], dW / dgate for every
 * gated layer (1), embedding gradients (1), HAT_SGD.step over every parameter incl. the embedding clamp (2).
 *   hat_gates_multi        gate_l = sigmoid(s * emb_row_l); sums2[0] = sum gate*(1-mask_pre), sums2[1] = sum (1-mask_pre)
 *                          and sums2[2] = their ratio (3 doubles, written, not accumulated; sums2 may be NULL)
 *   hat_scale_weights_multi  out = w * gate_in[c] per layer (gate_in NULL => copy: biases, the first layer)
 *   hat_weight_grads_multi   in place: dgate_in[c] = sum_{k,r} g*w ; g *= gate_in[c]
 *   hat_emb_grads_multi      demb[rows][n] = 0 except row t = (dgate + lamb/count*(1-mask_pre)) * s*a*(1-a); count <= 0 reads
 *                            sums2[1] on the device (no host synchronisation)
 *   hat_sgd_step_multi       HAT_utils.py:192-250 per parameter (clip_grad_norm_ is per parameter) + clamp of the embeddings to
 *                            +-thres_emb when thres_emb > 0 (hat.py:238-240); ws: clhip_hat_sgd_multi_ws(n_params) bytes   */
typedef struct { float* theta; float* grad; float* buf; const float* mask_back; size_t n; int is_emb; int reserved; } clhip_hat_param;
typedef struct { const float* emb_row; float* gate; const float* mask_pre; int n; int reserved; } clhip_hat_gate_job;
typedef struct { const float* w; const float* gate_in; float* out; size_t K, C, R; } clhip_hat_layer;
typedef struct { float* g; const float* w; const float* gate_in; float* dgate_in; int K, C, R, reserved; } clhip_hat_wgrad_job;
typedef struct { const float* dgate; const float* gate; const float* mask_pre; float* demb; int n, rows, t, reserved; } clhip_hat_emb_job;
size_t clhip_hat_sgd_multi_ws(int n_params);
int clhip_hat_sgd_step_multi(const clhip_hat_param* params, int n_params, float lr, float momentum, float wd, int finetune,
                             float s, float smax, float thres_cosh, float clipgrad, float thres_emb, int first, void* ws,
                             size_t ws_bytes, void* stream);
int clhip_hat_gates_multi(const clhip_hat_gate_job* jobs, int n_jobs, float s, double* sums2, void* stream);
int clhip_hat_scale_weights_multi(const clhip_hat_layer* layers, int n_layers, void* stream);
int clhip_hat_weight_grads_multi(const clhip_hat_wgrad_job* jobs, int n_jobs, void* stream);
int clhip_hat_emb_grads_multi(const clhip_hat_emb_job* jobs, int n_jobs, float s, float lamb, float count, const double* sums2,
                              void* stream);

/* ------------------------------------------------------------------ static-plan net executor
 * One call per pass for VGG-style nets instead of one Python dispatch per op
 * (replaces `outputs = model(inputs); loss.backward()` of EWC/train_EWC.py:181-187,
 * EWC/main_EWC.py:147-149, MAS/train_MAS.py:549-560, framework/inference.py:52-68).
 * params / grads are flat fp32 arenas; w_off / b_off are float offsets into them.          */
typedef struct {
    int type;            /* 0: conv3x3 pad 1 (+ReLU) (+2x2 max-pool)   1: Linear (+ReLU) */
    int cin, cout;       /* channels (conv) or in/out features (fc) */
    int relu, pool;      /* pool: 0 none, 1 = a max-pool follows (pool_k x pool_k, stride pool_s; 0/0 means 2x2 stride 2) */
    long w_off, b_off;
    int ksize, stride, pad;   /* conv geometry; ksize 0 means the VGG default 3x3, stride 1, pad 1 */
    int pool_k, pool_s;
    int bn;                   /* conv layers: 1 = BatchNorm2d between the convolution and the ReLU (clhip_net_set_bn) */
    long bn_w_off, bn_b_off;  /* float offsets of the BatchNorm weight / bias in the parameter arena */
    int has_drop;             /* a dropout may be set in front of this layer (clhip_net_set_dropout): its masked input gets
                               * its own buffer, so that the un-masked activation stays readable (clhip_net_layer_input);
                               * without it the mask is applied in place */
} clhip_layer_desc;

int clhip_net_create(const clhip_layer_desc* layers, int n_layers, int max_batch, int in_c, int in_h,
                     int in_w, void** out_handle);
/* nn.Dropout of the AlexNet classifier (torchvision alexnet: classifier[0], [3]) in training mode, and GEM's per-observe
 * masks (methods/rehearsal/GEM/gem.py:166-196): the mask (values 0 or 1/p_retain, drawn by the caller) multiplies the
 * INPUT of plan layer `layer` (> 0) in forward and the gradient w.r.t. it in backward.  mask [N][in_elems] with
 * row_stride floats between samples, row_stride 0 = one row for the whole batch; NULL = off (eval mode). */
int clhip_net_set_dropout(void* handle, int layer, const float* mask, long row_stride);
/* BatchNorm buffers of plan layer `layer` (device fp32 [cout]; updated in place by training-mode forwards) and the
 * module's momentum / eps; clhip_net_set_training switches every BatchNorm layer between batch and running statistics
 * (nn.Module.train / eval). */
int clhip_net_set_bn(void* handle, int layer, float* running_mean, float* running_var, float momentum, float eps);
int clhip_net_set_training(void* handle, int training);
/* Measurement only (bench.py's roofline object): HIP events around the forward launch(es) of plan layer `layer`, recorded
 * on the stream of each clhip_net_forward / clhip_net_loss_step call into a ring of 64 pairs (layer < 0: off).
 * clhip_net_probe_read waits for the recorded launches, returns their average duration in microseconds and how many
 * passes it covers, and restarts the count. */
int clhip_net_probe(void* handle, int layer);
/* the same around the layer's backward-data (kind 1) or weight-gradient (kind 2) launch(es); kind 0 = clhip_net_probe */
int clhip_net_probe_kind(void* handle, int layer, int kind);
int clhip_net_probe_read(void* handle, float* avg_us, int* count);
/* Where the arg-max bytes of a max-pooled conv layer live after a forward: byte offset into ws and bytes per
 * sample ([cout][oh][ow], window position r*k + c, first maximum wins).  With the saved activations
 * (clhip_net_layer_input) this is every non-linear decision the backward pass will use — what a parity harness
 * needs to judge gradients independently of ReLU / arg-max near-ties.  EINVAL for layers without a pool.   */
int clhip_net_layer_pool_idx(void* handle, int layer, size_t* ws_byte_off, size_t* elems_per_sample);
/* Which kernels the plan chose for a layer (measurement harnesses time the same ones): bit 0 forward, bit 1 backward-data, bit 2
 * weight gradient through a prepared-weights path instead of the direct f32 MFMA kernels — Winograd F(2x2,3x3) (csrc/wino.hip)
 * unless bit 3 (forward) / bit 4 (backward-data) says the launch is the bf16-split kernel (csrc/bsconv.hip); bit 5: the weight
 * gradient is the bf16-split kernel (csrc/bswgrad.hip); < 0 on error. */
int clhip_net_layer_paths(void* handle, int layer);

/* Side branches off a plan (EBLL's code layers on the flattened features, AlexNet_EBLL.py:110-117): the INPUT activation
 * of plan layer `layer` (> 0) lives at float offset *ws_float_off of the workspace after a forward (in_elems floats per
 * sample); `extra` ([N][in_elems], device, may be NULL = none) is added to the gradient w.r.t. that activation in the
 * following backward passes until changed. */
int clhip_net_layer_input(void* handle, int layer, size_t* ws_float_off, size_t* in_elems);
int clhip_net_set_input_grad(void* handle, int layer, const float* extra);
void clhip_net_destroy(void* handle);
size_t clhip_net_workspace_bytes(void* handle);
int clhip_net_num_classes(void* handle);
int clhip_net_forward(void* handle, const float* params, const float* x, int N, void* ws,
                      float* logits_out, void* stream);
int clhip_net_backward(void* handle, const float* params, float* grads, const float* x, int N, void* ws,
                       const float* dlogits, void* stream);
/* loss_kind 0: CE mean, 1: CE sum, 2: sum of squared logits.  grads == NULL => forward + loss only
 * (validation / test).  stats as in clhip_softmax_ce.
 * Small classifiers (Linear-ReLU-Linear-ReLU-Linear, hidden widths <= 128, <= 32 logits, no dropout inside) run
 * everything behind the first Linear layer's GEMM — split-K sum, Linear 2-3, the loss, backward-data down to the first
 * layer's output — as ONE launch, and the first layer's backward-data GEMM together with all Linear weight gradients as
 * another; results are bit-identical to the per-layer launches (CLHIP_FC_TAIL=0 at plan creation selects those).  The
 * fused launch keeps a 4-byte arrival counter in device memory owned by the plan: do not run ONE plan on two streams at
 * the same time (its activations live in one workspace, so that was never meaningful).                             */
int clhip_net_loss_step(void* handle, const float* params, float* grads, const float* x,
                        const int64_t* labels_i64, int N, int loss_kind, void* ws, float* loss_out,
                        double* stats, float* logits_out, void* stream);

/* cross-entropy over the class slice [col_off, col_off+ncols) of the shared head (ncols 0 => to the end) */
int clhip_net_loss_step_slice(void* handle, const float* params, float* grads, const float* x,
                              const int64_t* labels_i64, int N, int loss_kind, int col_off, int ncols, void* ws,
                              float* loss_out, double* stats, float* logits_out, void* stream);

/* ------------------------------------------------------------------ GEM memory gradients
 * rehearsal/model/gem.py:20-80,275-277.  G[n_tasks][ld]: one contiguous row per task.
 *   axpy        y = (assign ? 0 : y) + alpha*x : store_grad (:20-35) / accumulation over memory batches (:237-255)
 *   gem_gram    out_f64[m*m] = rows(row_idx) . rows(row_idx)^T in f64, one pass (the MM^T and M g of :70-73;
 *               its last row/column also carries the violation test g.G_tt of :275-277)
 *   gem_project out = g + sum_i v[i]*G[row_idx[i]]   (:79, written straight into the gradient arena = overwrite_grad) */
int clhip_axpy(float* y, const float* x, size_t n, float alpha, int assign, void* stream);
size_t clhip_gem_gram_ws(int m);
int clhip_gem_gram(const float* G, size_t ld, const int* row_idx_host, int m, size_t n, double* out_f64, void* ws,
                   size_t ws_bytes, void* stream);
int clhip_gem_project(const float* G, size_t ld, const int* row_idx_host, const float* v_host, int m, const float* g,
                      float* out, size_t n, void* stream);
/* project2cone2's QP itself (gem.py:58-80, quadprog.solve_qp = Goldfarb-Idnani dual active set) on the device, so that an
 * observe step never synchronises with the host:
 *   gem_qp          from clhip_gem_gram's f64 output over [memory rows..., current gradient LAST] (m rows, m - 1 <= 15
 *                   unknowns): P = 1/2 (MM^T + MM^T^T) + eps I, q = -M g, min 1/2 v^T P v - q^T v s.t. v >= margin;
 *                   v_out_f64[m - 1]; info_i32[0] = #{k : g . G_k < 0} (gem.py:275-277; 0 => v = 0), info_i32[1] = 0 ok,
 *                   1 iteration limit, 2 infeasible
 *   gem_project_dev clhip_gem_project with v and the violation count read from device memory: out = g when nothing is
 *                   violated (a no-op for out == g), else g + sum_i v[i] G[row_idx[i]]                                  */
int clhip_gem_qp(const double* gram_f64, int m, double margin, double eps, double* v_out_f64, int* info_i32, void* stream);
int clhip_gem_project_dev(const float* G, size_t ld, const int* row_idx_host, const double* v_dev_f64, const int* info_dev,
                          int m, const float* g, float* out, size_t n, void* stream);

#ifdef CLHIP_VISIBILITY_PUSHED
#pragma GCC visibility pop
#undef CLHIP_VISIBILITY_PUSHED
#endif

#ifdef __cplusplus
}
#endif
#endif /* CLHIP_H */
