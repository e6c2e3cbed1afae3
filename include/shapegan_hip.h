/* include/shapegan_hip.h — C ABI of libshapegan_hip.so (MI355X / gfx950 only).
 *
 * The reference (marian42/shapegan) has no FFI of its own: its hot path is the set of ATen kernels that
 * torch.nn launches underneath model/gan.py, model/autoencoder.py, model/progressive_gan.py and
 * model/sdf_net.py.  This header is the boundary a maintainer binds instead of those ATen calls; every entry
 * point names the reference site it replaces.  The Python shells in shapegan_amd/model/ (gan.py, ...) (same class names,
 * constructor arguments, state_dict keys and file names as the reference's model/ package) call these through
 * ctypes on tensor.data_ptr() — see INTEGRATION.md.
 *
 * Conventions
 *   - all tensors fp32, contiguous, device pointers; NCDHW for voxel tensors, row-major [N,features] for MLPs;
 *   - every call is asynchronous and ordered on `stream` (pass torch.cuda.current_stream().cuda_stream);
 *   - the caller owns every buffer, including workspaces (sizes from the *_workspace_bytes helpers); nothing is
 *     retained past the call; no global mutable state, so calls are re-entrant across threads/streams;
 *   - return 0 on success, <0 on error (SG_ERR_*); sg_last_error() gives a thread-local message;
 *   - `act` is one of SG_ACT_*, fused into the producing kernel's epilogue (slope = LeakyReLU negative slope).
 */
#ifndef SHAPEGAN_HIP_H
#define SHAPEGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_ABI_VERSION 8

typedef struct ihipStream_t* hipStream_t; /* the opaque handle hip_runtime_api.h declares (identical re-typedef) */

#define SG_ACT_NONE_C 0
#define SG_ACT_LEAKY_C 1
#define SG_ACT_RELU_C 2
#define SG_ACT_TANH_C 3
#define SG_ACT_SIGMOID_C 4

int sg_abi_version(void);
const char* sg_last_error(void);

/* ---- K1: nn.Conv3d(kernel 4, stride 2, padding 1) --------------------------------------------------------
 * reference: model/gan.py:49-53 (Discriminator), model/autoencoder.py:16-24 (encoder),
 *            model/progressive_gan.py:38 (optional_layers[i][0])  -> aten::convolution / convolution_backward.
 * x [batch,Cx,ID,IH,IW] -> y [batch,Cout,ID/2,IH/2,IW/2]; w [Cout,Cin_total,4,4,4].
 * Only input channels [0,Cin) are read (Cin < Cin_total reproduces from_SDF's zero padding,
 * model/progressive_gan.py:9-16, without materialising the zero channels). */
size_t sg_conv3d_k4s2p1_fwd_workspace_bytes(int batch, int Cin, int Cout, int OD, int OH, int OW);
int sg_conv3d_k4s2p1_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                         int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                         size_t workspace_bytes, hipStream_t stream);
/* Kept weight images (see sg_conv3d_k4s2p1_dgrad_keep below) for the forward form, and the images of up to 8 forthcoming
 * _keep calls packed in ONE launch: a critic update packs four images of two weights (model/gan.py:51-53 forward and input
 * gradient), one small launch each before.  kinds[i]: 0 forward, 1 input gradient; dims[8 i ..] = {batch, Cin, Cin_total, Cx, Cout,
 * ID, IH, IW} of call i; served[i] = 1: make call i with weights_unchanged = 1. */
int sg_conv3d_k4s2p1_fwd_keep(const float* x, const float* w, const float* bias, float* y, int batch, int Cin, int Cin_total,
                              int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                              size_t workspace_bytes, int weights_unchanged, hipStream_t stream);
/* 0: the call {kind, dims[8]} (as in sg_conv3d_k4s2p1_pack_images) with a workspace of workspace_bytes is not served by a kernel
 * with a kept weight image.  Otherwise a number that is equal for two calls exactly when they read the same image: callers key their
 * kept images on it instead of on the full shape (the batch size does not enter the LDS-halo kernels' images).  Host code only. */
long long sg_conv3d_k4s2p1_image_layout(int kind, const int* dims, size_t workspace_bytes);
int sg_conv3d_k4s2p1_pack_images(int n, const int* kinds, const float* const* weights, void* const* workspaces,
                                 const size_t* workspace_bytes, const int* dims, int* served, hipStream_t stream);
/* testing / tuning: force one forward implementation (0 = gather implicit GEMM, 1 = LDS-halo implicit GEMM) */
int sg_conv3d_k4s2p1_fwd_impl(const float* x, const float* w, const float* bias, float* y, int batch, int Cin,
                              int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                              void* workspace, size_t workspace_bytes, int impl, int debug, hipStream_t stream);
/* dx[batch,Cx(first Cin channels),ID,IH,IW] = conv^T(dy, w) (+bias[ci], act: used when this is a ConvTranspose fwd) */
size_t sg_conv3d_k4s2p1_dgrad_workspace_bytes(int Cout, int Cin);
/* the same with the batch and the output grid of the convolution (O = I/2): also covers the tap-plane scratch of the
 * one-channel layers (Cin == 1: ConvTranspose3d(C -> 1) forward / Conv3d(1 -> C) input gradient) */
size_t sg_conv3d_k4s2p1_dgrad_workspace_bytes_for(int batch, int Cin, int Cout, int OD, int OH, int OW);
int sg_conv3d_k4s2p1_dgrad(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                           int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                           void* workspace, size_t workspace_bytes, hipStream_t stream);
/* sg_conv3d_k4s2p1_dgrad with the packed weight image KEPT between calls — the forward of a ConvTranspose3d whose weights
 * did not change since its last call (model/gan.py:29-35: the WGAN generator is evaluated six times per 5+1 training unit of
 * train_wgan.py:60-84 and updated once).  `workspace` is a buffer the caller dedicates to this weight tensor (size as for
 * sg_conv3d_k4s2p1_dgrad); weights_unchanged != 0 promises that the previous call on it had the same weight values and the same
 * shapes and that nothing else wrote to it: the packing launch is then skipped.  Results are those of sg_conv3d_k4s2p1_dgrad. */
int sg_conv3d_k4s2p1_dgrad_keep(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                                int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                                void* workspace, size_t workspace_bytes, int weights_unchanged, hipStream_t stream);
int sg_conv3d_k4s2p1_dgrad_impl(const float* dy, const float* w, const float* bias, float* dx, int batch, int Cin,
                                int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope,
                                void* workspace, size_t workspace_bytes, int impl, hipStream_t stream); /* tests */
/* dw[Cout, Cin_total(first Cin channels written), 4,4,4] = sum_{n,o} dy * x-patches; split-K workspace optional */
size_t sg_conv3d_k4s2p1_wgrad_workspace_bytes(int batch, int Cin, int Cout, int OD, int OH, int OW);
int sg_conv3d_k4s2p1_wgrad_impl(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                                int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes, int impl,
                                hipStream_t stream); /* tests */
int sg_conv3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                           int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);
/* Weight AND bias gradient of y = act(conv(x) + b) from dy = dLoss/dy in one pass: dz = dy * act'(y) is formed inside the
 * weight-gradient kernel (act'(y) from the activated output, as sg_act_bwd), db[co] = sum of dz over batch and positions.  For a
 * layer whose input needs no gradient (the critic's first layer, model/gan.py:49 under train_wgan.py:69) the activation backward
 * then is no pass of its own.  Served shapes: sg_conv3d_k4s2p1_wgrad_act_eligible (one-channel layers, LeakyReLU / ReLU). */
/* The incoming gradient's fragment image of the LDS-halo weight-gradient kernel written by the gradient's PRODUCER:
 * sg_conv3d_k4s2p1_wgrad_dy_image says whether the weight-gradient call (dy [batch, Cout, O^3], the workspace it will get) is served
 * that way (1: the image is [mt_total][nslice][8][64] float4 at the start of that workspace; 0: use sg_conv3d_k4s2p1_wgrad);
 * sg_act_bwd_rowsum_pack8 = sg_act_bwd_rowsum on [N, C, 8, 8, 8] tensors (LeakyReLU / ReLU) + the image; sg_head_dot_bwd writes it
 * for 4^3 grids; sg_conv3d_k4s2p1_wgrad_prepacked = sg_conv3d_k4s2p1_wgrad without its packing launch. */
int sg_conv3d_k4s2p1_wgrad_dy_image(int batch, int Cin, int Cout, int OD, int OH, int OW, size_t workspace_bytes, int* mt_total,
                                    long* nslice);
int sg_act_bwd_rowsum_pack8(const float* y, const float* dy, float* dz, float* rowsum, void* dz_image, long N, int C, long nslice,
                            int act, float slope, hipStream_t stream);
int sg_conv3d_k4s2p1_wgrad_prepacked(const float* dy, const float* x, float* dw, int batch, int Cin, int Cin_total, int Cx,
                                     int Cout, int ID, int IH, int IW, void* workspace, size_t workspace_bytes, hipStream_t stream);
int sg_conv3d_k4s2p1_wgrad_act_eligible(int batch, int Cin, int Cout, int OD, int OH, int OW, int act);
int sg_conv3d_k4s2p1_wgrad_act(const float* dy, const float* y, const float* x, float* dw, float* db, int batch, int Cin,
                               int Cin_total, int Cx, int Cout, int ID, int IH, int IW, int act, float slope, void* workspace,
                               size_t workspace_bytes, hipStream_t stream);

/* ---- K2: nn.ConvTranspose3d(kernel 4, stride 2, padding 1) -------------------------------------------------
 * reference: model/gan.py:13,17,21 (Generator), model/autoencoder.py:55,59,63 (decoder).
 * x [batch,Cin_T,ID,IH,IW] -> y [batch,Cout_T,2ID,2IH,2IW]; w [Cin_T,Cout_T,4,4,4]. */
int sg_convT3d_k4s2p1_fwd(const float* x, const float* w, const float* bias, float* y, int batch, int Cin_T, int Cout_T,
                          int ID, int IH, int IW, int act, float slope, void* workspace, size_t workspace_bytes,
                          hipStream_t stream);
/* The same for C -> 1 channel (C <= 64, planes of at most 256 positions) with the INPUT taken through
 * act_in(x * in_scale[c] + in_shift[c]) inside the kernel's loads: a BatchNorm3d + LeakyReLU between the producing layer and this
 * one (model/gan.py:18-21, model/autoencoder.py:60-63) then never is a pass of its own — sg_bn_train_stats supplies
 * scale / shift.  in_act: none, LeakyReLU (0 <= in_slope <= 1) or ReLU. */
int sg_convT3d_k4s2p1_to1_pre_eligible(int batch, int C, int ID, int IH, int IW);
int sg_convT3d_k4s2p1_to1_pre(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                              const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW, int act,
                              float slope, hipStream_t stream);
/* the same with the kernel form chosen by the caller (tests / tuning; the library reads no environment variable): form 0 = the
 * dispatch rule, 1 / 2 = one output-row parity per workgroup with one / two plane walks, 3 / 4 = both row parities per workgroup
 * with one / two plane walks.  Every form computes the same sums in the same order (bit-identical results). */
int sg_convT3d_k4s2p1_to1_pre_impl(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                   const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                                   int act, float slope, int form, hipStream_t stream);
/* Several independent batches in one pass ("groups" of samples_per_group samples: e.g. the generator evaluations of four
 * consecutive critic updates, train_wgan.py:60-63, whose BatchNorm statistics are per evaluation): sample n takes row n /
 * samples_per_group of in_scale / in_shift ([groups][C]) and is written at y + (n / samples_per_group) * y_group_stride +
 * (n % samples_per_group) * (2I)^3 — each group into its own destination (the fake half of its critic batch). */
int sg_convT3d_k4s2p1_to1_pre_grouped(const float* x, const float* w, const float* bias, float* y, const float* in_scale,
                                      const float* in_shift, int in_act, float in_slope, int batch, int C, int ID, int IH, int IW,
                                      int act, float slope, int samples_per_group, long y_group_stride, hipStream_t stream);
int sg_convT3d_k4s2p1_dgrad(const float* dy, const float* w, float* dx, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, void* workspace, size_t workspace_bytes, hipStream_t stream);
int sg_convT3d_k4s2p1_wgrad(const float* dy, const float* x, float* dw, int batch, int Cin_T, int Cout_T, int ID, int IH,
                            int IW, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- K3/K6: GEMM + bias (+activation) -------------------------------------------------------------------------
 * reference: nn.Linear (model/autoencoder.py:34,41-42,45; model/progressive_gan.py:28,30) -> aten::addmm/mm, and the
 * kernel-4 stride-1 convolutions on 1^3 / 4^3 grids (model/gan.py:9,55; model/autoencoder.py:28,51).
 *   C(i,j) = act( sum_k A(i,k) B(k,j) + bias_i[i] + bias_j[j >> bias_j_shift] ), element strides for all operands;
 *   each of A, B needs a unit stride on one axis. */
size_t sg_gemm_workspace_bytes(int M, int N);
int sg_gemm(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, long sci, long scj,
            const float* bias_i, const float* bias_j, int bias_j_shift, int M, int N, int K, int act, float slope,
            void* workspace, size_t workspace_bytes, hipStream_t stream);
/* C[M,N] = A[M,K] B[N,K]^T, K contiguous in both operands and K >> M, N: the SDFNet weight gradients dW_l = dZ_l H_{l-1}^T over
 * the points of a batch (autograd of model/sdf_net.py:56-61).  Deterministic split-K; rows of C have stride ldc. */
size_t sg_gemm_nt_workspace_bytes(int M, int N, long K);
int sg_gemm_nt(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, long K, void* workspace,
               size_t workspace_bytes, hipStream_t stream);
/* up to 8 such products in one launch: member b multiplies A + a_off[b] with B + b_off[b] (element offsets, shared leading
 * dimensions and sizes) into C + c_off[b] with row stride ldc[b] — all weight gradients of an SDFNet backward at once */
size_t sg_gemm_nt_batched_workspace_bytes(int batch, int M, int N, long K);
int sg_gemm_nt_batched(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb, float* C,
                       const long* c_off, const long* ldc, int batch, int M, int N, long K, void* workspace,
                       size_t workspace_bytes, hipStream_t stream);
/* The same with B_b holding LayerNorm-normalised rows: C_b = A_b * relu(gamma_b (.) B_b + beta_b)^T with gamma_b = gamma + g_off[b],
 * beta_b = beta + g_off[b] ([N] each): the hidden weight gradients of the LayerNorm MLP (K7b) straight from the xhat images. */
int sg_gemm_nt_batched_lnrelu(const float* A, const long* a_off, long lda, const float* B, const long* b_off, long ldb,
                              const float* gamma, const float* beta, const long* g_off, float* C, const long* c_off, const long* ldc,
                              int batch, int M, int N, long K, void* workspace, size_t workspace_bytes, hipStream_t stream);
int sg_colsum(const float* x, float* out, int rows, int cols, long ld, hipStream_t stream); /* bias grads */
int sg_rowsum(const float* x, float* out, long rows, long len, long ld, hipStream_t stream);
/* rows [d*rows_per_dst, (d+1)*rows_per_dst) go to outs[d] (ndst <= 8 host-side pointers to device buffers): one launch for
 * bias gradients that live in separate slices of a flat gradient buffer */
int sg_rowsum_multi(const float* x, float* const* outs, const long* out_strides, int ndst, long rows_per_dst, long len, long ld,
                    hipStream_t stream); /* out_strides (optional, host array): element stride of each destination, default 1 */
/* out[r*nseg + s] = sum of x[r*ld + e] over e in [seg_off[s], seg_off[s+1])  (per-shape sums of SDFNet dZ columns) */
int sg_segsum(const float* x, float* out, long rows, long ld, const int64_t* seg_off, long nseg, hipStream_t stream);

/* ---- K4: nn.BatchNorm3d / nn.BatchNorm1d (+ fused following activation) -------------------------------------
 * reference: model/gan.py:10,14,18; model/autoencoder.py:17,21,25,29,38,46,56,60,64 -> aten::batch_norm(_backward).
 * x [N,C,S]; momentum/eps as torch (0.1 / 1e-5); running_var gets the unbiased estimate. */
size_t sg_bn_workspace_bytes(int C);
int sg_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                    float* running_mean, float* running_var, long long* num_batches_tracked, int N, int C, long S,
                    float eps, float momentum, int act, float slope, void* workspace, size_t workspace_bytes,
                    hipStream_t stream);
/* Batch statistics WITHOUT the normalised output: mean / invstd, the running-statistics update exactly as sg_bn_train_fwd does
 * it, and the affine map scale[c] = gamma[c] * invstd[c], shift[c] = beta[c] - mean[c] * scale[c] that a consumer applies on its
 * loads (sg_convT3d_k4s2p1_to1_pre). */
int sg_bn_train_stats(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                      float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift, int N,
                      int C, long S, float eps, float momentum, void* workspace, size_t workspace_bytes, hipStream_t stream);
/* The two above for a tensor [groups][N][C][S] whose groups are independent batches (statistics per group; save_mean / save_invstd
 * / scale / shift are [groups][C]; workspace: groups * sg_bn_workspace_bytes(C)).  The running statistics receive the groups'
 * updates one after the other, exactly as `groups` separate calls would apply them; num_batches_tracked += groups. */
int sg_bn_train_fwd_grouped(const float* x, const float* gamma, const float* beta, float* y, float* save_mean, float* save_invstd,
                            float* running_mean, float* running_var, long long* num_batches_tracked, int groups, int N, int C,
                            long S, float eps, float momentum, int act, float slope, void* workspace, size_t workspace_bytes,
                            hipStream_t stream);
int sg_bn_train_stats_grouped(const float* x, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                              float* running_mean, float* running_var, long long* num_batches_tracked, float* scale, float* shift,
                              int groups, int N, int C, long S, float eps, float momentum, void* workspace, size_t workspace_bytes,
                              hipStream_t stream);
int sg_bn_eval_fwd(const float* x, const float* gamma, const float* beta, float* y, const float* running_mean,
                   const float* running_var, float* save_mean, float* save_invstd, int N, int C, long S, float eps,
                   int act, float slope, hipStream_t stream);
int sg_bn_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
              const float* save_invstd, float* dx, float* dgamma, float* dbeta, int N, int C, long S, int train, int act,
              float slope, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- K5: activations (standalone; normally fused into K1-K4/K7 epilogues) ------------------------------------
 * sg_act_bwd takes the activation OUTPUT y (LeakyReLU/ReLU masks, 1-y^2, y(1-y)); it is also LeakyReLU's
 * double-backward (mask * gg) used by the WGAN-GP graph (train_hybrid_progressive_gan.py:102-111). */
int sg_act_fwd(const float* x, float* y, long n, int act, float slope, hipStream_t stream);
int sg_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float slope, hipStream_t stream);
/* dx = dy * act'(y) and rowsum[r] = sum of dx over row r of S voxels (r = sample * C + channel): the activation backward of a
 * conv layer with the row sums its bias gradient needs, in one pass */
int sg_act_bwd_rowsum(const float* y, const float* dy, float* dx, float* rowsum, long rows, long S, int act, float slope,
                      hipStream_t stream);

/* ---- K7: SDFNet fused MLP -------------------------------------------------------------------------------------
 * reference: SDFNet.forward, model/sdf_net.py:26-61 (+ autograd).  `params` = 16 device pointers in state_dict order
 * (layers1.{0,2,4,6}.{weight,bias}, layers2.{0,2,4,6}.{weight,bias}).  sg_sdfnet_pack builds the MFMA-fragment
 * image of the weights (call once per optimizer step); kin_used = 3+latent (per-point latents, reference
 * semantics) or 3 (per-shape latents folded into zb1/zb5 biases; replaces the [B*R^3, L] tiling of
 * train_hybrid_wgan.py:67-70 / train_hybrid_progressive_gan.py:90-93). */
size_t sg_sdfnet_packed_floats(int kin_used);
/* floats of the activation buffer `acts` of a training call (sg_sdfnet_fwd writes it, sg_sdfnet_bwd and the weight-gradient
 * GEMMs read it): the fp32 images H1..H7 [7][256][ldn], followed by their SIGN MASKS, unsigned short [7][16][ldn] — bit q of
 * mask[l][2 w + h][p] is (H_{l+1}[32 w + (q & 3) + 8 (q >> 2) + 4 h][p] > 0), one 16-bit word per point and group of 16 rows.
 * The backward takes ReLU' of H1..H6 from the masks (1/32 of the bytes) instead of re-reading the images. */
size_t sg_sdfnet_acts_floats(long ldn);
int sg_sdfnet_pack(const float* const* params, int latent, int kin_used, float* packed, hipStream_t stream);
int sg_sdfnet_fwd(const float* points, long points_period, const float* latent, const int64_t* latent_idx,
                  int latent_size, const float* packed, int kin_used, const float* zb1, const float* zb5,
                  long points_per_shape, const int* shape_index, float* out, float* acts, long ldn, long N,
                  hipStream_t stream);
/* Point tiles of the backward = rows of bias_partials: tile t covers the points [sg_sdfnet_bwd_tile_start(N, t),
 * sg_sdfnet_bwd_tile_start(N, t + 1)).  With tiles = ceil(N / 64), rem = tiles % 512, full = tiles - rem: 64 points per tile,
 * except for the last, partly filled round of workgroups (512 = two per CU):
 *   256 < rem <= 384: 64-point tiles up to full + 256 (one more per CU), the points after them in 32-point tiles;
 *   full > 0 and 0 < rem <= 256: the points from 64 * full on in 32-point tiles.
 * A pure function of N: the layout does not depend on the device. */
long sg_sdfnet_bwd_blocks(long N);
long sg_sdfnet_bwd_tile_start(long N, long t);
/* bias_partials (optional): TILE-MAJOR [blocks][SG_SDFNET_PARTIAL_ROW] (ABI 7; it was [14*256][blocks]: 3 584 scattered 4-byte
 * writes per tile).  Row t holds the sums over the points of tile t: floats [256 b, 256 b + 256), b = 0..6: the row sums of
 * dZ_{b+1} (sum over the tiles: bias gradients); float 14*256: the sum of dz8 (the layers2.6 bias gradient).  With `points` (the
 * xyz of the batch, as given to sg_sdfnet_fwd) also b = 7: sum_p dz8[p] H7[row][p] (the layers2.6 weight gradient) and b = 8+c /
 * 11+c: sum_p dZ1 / dZ5 [row][p] * xyz_c[p] (the three point columns of the layers1.0 / layers2.0 weight gradients) — sums the
 * kernel has the operands on chip for, instead of three more passes over the [256][N] images.  sg_sdfnet_bwd_finish reduces
 * them. */
#define SG_SDFNET_PARTIAL_ROW 3616 /* 14 * 256 + 32 */
#define SG_SDFGEN_PARTIAL_ROW 7200 /* SG_SDFNET_PARTIAL_ROW + 14 * 256: the LayerNorm form (sg_sdfgen_bwd) */
int sg_sdfnet_bwd(const float* dout, const float* out, const float* acts, float* dz, float* dz8, float* bias_partials,
                  const float* points, long points_period, float* dx, long dx_ld, const float* packed, int kin_used, long ldn,
                  long N, hipStream_t stream);
/* Per-shape mode: the latent fold zb1[s][o] = b1[o] + sum_k z[s][k] W1[o][3+k], zb5[s][o] = b5[o] + sum_k z[s][k] W5[o][259+k]
 * (W1 [256][3+L], W5 [256][259+L]: the reference multiplies these columns with the tiled latents of every point,
 * model/sdf_net.py:27,41,57-59), both [nshapes][256] outputs in one launch, accumulated in double and rounded once. */
int sg_sdfnet_shape_bias(const float* z, long nshapes, int latent, const float* W1, const float* b1, const float* W5, const float* b5,
                         float* zb1, float* zb5, hipStream_t stream);
/* sg_sdfnet_pack(params, latent, 3, packed) and sg_sdfnet_shape_bias(z, ..., zb1, zb5) of the same parameters in ONE launch (ABI 8):
 * every step of the shape-sorted auto-decoder needs both behind its optimizer step (train_sdf_autodecoder.py:80-91). */
int sg_sdfnet_pack_shape_bias(const float* const* params, int latent, float* packed, const float* z, long nshapes, float* zb1,
                              float* zb5, hipStream_t stream);
/* Per-shape mode: backward of the latent fold zb1[s][o] = b1[o] + sum_k z[s][k] W1[o][3+k], zb5[s][o] = b5[o] + sum_k z[s][k] W5[o][259+k]
 * (the latent columns of layers1.0 / layers2.0, model/sdf_net.py:27,41, enter sg_sdfnet_fwd as bias rows) from the per-shape sums
 * t1 / t5 [256][nshapes] of dZ1 / dZ5, in one launch: the latent columns of dW1 [256][3+L] / dW5 [256][259+L] written in place
 * (NULL, NULL to skip) and the latent gradient gz [nshapes][L] (NULL to skip); nshapes <= 6144. */
/* (reg_scale != 0: gz additionally receives reg_weight[s] * reg_scale * z[s][k] (reg_weight NULL: 1) — the gradient of the DeepSDF
 * latent regulariser SIGMA * mean(z_batch^2) through shape counts (train_sdf_autodecoder.py:88), so that the latent table gets ONE
 * gradient contribution, written where its flat-buffer slice lives, instead of two tensors that autograd adds and the optimizer copies) */
int sg_sdfnet_shape_bias_bwd(const float* t1, const float* t5, long nshapes, const float* z, int latent, const float* W1,
                             const float* W5, float* dW1, float* dW5, float* gz, const float* reg_weight, float reg_scale,
                             hipStream_t stream);
/* Everything that is derived from the bias_partials of ONE sg_sdfnet_bwd call, in one launch (replaces five: the segment sums,
 * two multi-row sums and the two-stage sum of dz8 of train_sdf_autodecoder.py:80-91's backward):
 *   bias_grads[0..6] (256 floats each; NULL: skip all column sums), b8_grad[1], and with `extended` (sg_sdfnet_bwd was given
 *   `points`) w8_grad[256] and the three point columns of dW1 / dW5: element (row, c) at w1_cols[row * w1_ld + c] / w5_cols[...];
 *   nseg > 0: t1[256][nseg], t5[256][nseg] = sums of dZ1 / dZ5 over every segment [seg_off[s], seg_off[s+1]) of points (per-shape
 *   sums: the latent-table gradient and the latent columns of the layers1.0 / layers2.0 weight gradients of the shape-sorted step),
 *   interior tiles from the partials, the cut tiles from `dz`.
 * Deterministic (two-level column sums, added in split order by whichever block finishes last).  `tickets`: 16 unsigned, zero
 * before the first call; every call leaves them zero (one buffer per stream of concurrent calls). */
size_t sg_sdfnet_bwd_finish_workspace_bytes(long N);
int sg_sdfnet_bwd_finish(const float* dz, const float* bias_partials, long ldn, long N, int extended, float* const* bias_grads,
                         float* w8_grad, float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld,
                         const int64_t* seg_off, long nseg, float* t1, float* t5, void* workspace, size_t workspace_bytes,
                         unsigned* tickets, hipStream_t stream);

/* ---- K7b: the LayerNorm form of the fused MLP — SDFGenerator (model/point_sdf_net.py:49-119) with hidden_channels 256 and
 * num_layers 8: x = relu(LayerNorm(lin_i(x) [+ z_lin(z)])) for i = 0..6 (:104-116), cat([x, pos]) in front of lins.4 (:100), a
 * plain Linear(256, 1) at the end (ABI 8).  The eight Linear layers have the shapes of an SDFNet without latent columns, so the
 * kernels are those of K7 with the LayerNorm statistics combined across the waves of a tile through LDS; the latent enters as
 * the per-shape rows zb1 = z_lin1(z) + lins.0.bias, zb5 = z_lin2(z) + lins.4.bias ([S,256], built by the caller: two small
 * Linear layers).  `params`: lins.{0..7}.{weight,bias} (16 pointers); `norm_params`: norms.{0..6}.{weight,bias} (14 pointers);
 * `packed`: sg_sdfnet_packed_floats(3) floats.
 * Training: `acts` (sg_sdfgen_acts_floats(ldn) floats) receives the images xhat_l = (x - mean) * rstd [7][256][ldn] (the
 * LayerNorm backward needs them where the ReLU is off too), the sign masks of the ReLU inputs (layout of sg_sdfnet_acts_floats)
 * and rstd [7][ldn].  sg_sdfgen_bwd (32-point tiles, sg_sdfgen_bwd_blocks(N) of them) writes dz8 = dout, the images dZ_l (gradient
 * of the LayerNorm INPUT) and per-tile partial sums [blocks][SG_SDFGEN_PARTIAL_ROW]: the SDFNet row (b = 0..13 and float 14*256,
 * see sg_sdfnet_bwd) followed at float SG_SDFNET_PARTIAL_ROW by 7 blocks sum_p dY_l xhat_l (LayerNorm weight gradients) and 7
 * blocks sum_p dY_l (LayerNorm bias gradients).  sg_sdfgen_bwd_finish reduces them (norm_grads[0..6]: weight, [7..13]: bias
 * gradients; tickets: 32 unsigned, zero); the six 256 x 256 weight gradients are sg_gemm_nt_batched_lnrelu products of the dZ and
 * xhat images. */
size_t sg_sdfgen_acts_floats(long ldn);
long sg_sdfgen_packed_norm_offset(int which); /* float offset of the packed LayerNorm weight (0) / bias (1) vectors [7][256] */
int sg_sdfgen_pack(const float* const* params, const float* const* norm_params, float* packed, hipStream_t stream);
int sg_sdfgen_fwd(const float* points, const float* packed, const float* zb1, const float* zb5, long points_per_shape,
                  const int* shape_index, float eps, float* out, float* acts, long ldn, long N, hipStream_t stream);
long sg_sdfgen_bwd_blocks(long N);
int sg_sdfgen_bwd(const float* dout, const float* acts, float* dz, float* dz8, float* partials, const float* points,
                  const float* packed, long ldn, long N, hipStream_t stream);
size_t sg_sdfgen_bwd_finish_workspace_bytes(long N);
int sg_sdfgen_bwd_finish(const float* dz, const float* partials, long ldn, long N, float* const* bias_grads, float* w8_grad,
                         float* b8_grad, float* w1_cols, long w1_ld, float* w5_cols, long w5_ld, float* const* norm_grads,
                         const int64_t* seg_off, long nseg, float* t1, float* t5, void* workspace, size_t workspace_bytes,
                         unsigned* tickets, hipStream_t stream);

/* ---- K8/K9/K10/K11: blends, reductions, latent-table rows, optimizers ------------------------------------------
 * reference: fade-in / GP lerp (model/progressive_gan.py:50, train_hybrid_progressive_gan.py:105), batch means
 * (train_wgan.py:68,82), latent_codes[model_indices] (train_sdf_autodecoder.py:78-82), optim.RMSprop / optim.Adam
 * with torch defaults (train_wgan.py:45-46, train_autoencoder.py:35, ...), clip_weights (model/gan.py:67-69;
 * fused into the RMSprop step when clip > 0). grad_scale multiplies the gradient on load (1/world for DP). */
int sg_axpby(const float* x, const float* y, float* out, long n, float a, float b, hipStream_t stream);
size_t sg_reduce_workspace_bytes(void);
int sg_reduce_sum(const float* x, float* out, long n, float scale, void* workspace, size_t workspace_bytes,
                  hipStream_t stream);
int sg_gather_rows(const float* table, const int64_t* idx, float* out, long n, int L, hipStream_t stream);
int sg_scatter_add_rows(const float* rows, long rows_ld, const int64_t* idx, float* table_grad, long n, int L,
                        hipStream_t stream);
/* Batch assembly of the auto-decoder step in shape-sorted order (train_sdf_autodecoder.py:78-85: model_indices =
 * indices // POINTCLOUD_SIZE, points[indices], sdf[indices]) as ONE stable counting sort on the shape id: out_points[n,3] /
 * out_sdf[n] / out_shape[n] hold the batch grouped by shape (entries of a shape keep their order in `indices`),
 * seg_off[nshapes+1] bounds every shape's run, counts[nshapes] = run lengths as floats (weights of the latent regulariser).
 * nshapes <= sg_sdf_batch_sort_max_shapes(); an index outside [0, nshapes*pointcloud_size) — an IndexError in the reference —
 * sets *bad_index_flag and is clamped.  The flag is an int the kernel can write — device memory, or pinned host memory (what the
 * Python shell passes: the host then polls it with a plain load, no copy / launch / synchronisation per step); sticky, caller-zeroed,
 * written only when an index is bad: it receives bad_index_value (!= 0; e.g. a call sequence number).  bad_index_device (optional):
 * a second sticky word, in DEVICE memory, set together with the first — the `skip_if_nonzero` word of sg_adam_step_guarded: the
 * reference raises before any update (train_sdf_autodecoder.py:79 -> :90-91), here the optimizer kernels behind the sort in the same
 * stream turn into no-ops from that batch on and the host raises when it next looks at the first word.  With the device word
 * given, both words keep the value of the FIRST call that met a bad index until the caller zeroes them. */
int sg_sdf_batch_sort_max_shapes(void);
size_t sg_sdf_batch_sort_workspace_bytes(long n, long nshapes);
int sg_sdf_batch_sort(const int64_t* indices, long n, long pointcloud_size, long nshapes, const float* points,
                      const float* sdf, float* out_points, float* out_sdf, int* out_shape, int64_t* seg_off, float* counts,
                      int* bad_index_flag, int* bad_index_device, int bad_index_value, void* workspace, size_t workspace_bytes,
                      hipStream_t stream);
int sg_rmsprop_step(float* p, const float* g, float* square_avg, long n, float lr, float alpha, float eps,
                    float grad_scale, float clip, hipStream_t stream);
int sg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                 float eps, long step, float grad_scale, hipStream_t stream);
/* Adam with the step counter (int64) in device memory: identical arithmetic, but the call has no host-side state that changes
 * between steps, so a captured hipGraph of a training step replays it.  corr_dev: SG_ADAM_DEV_WORDS 32-bit words, zero before the
 * first call — [0..1] the two bias corrections of the last applied step (for the host to read), [2] and [32 + 32 i], i < 16: the
 * arrival tickets of the one launch (the workgroup that is last to have read the counter advances it; two levels, 128 bytes apart:
 * atomics on one word are served one after the other), left at zero by every call. */
#define SG_ADAM_DEV_WORDS 544
int sg_adam_step_dev(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                     float eps, long long* step_dev, float* corr_dev, float grad_scale, hipStream_t stream);
/* The two Adam entries with a guard word (device memory, may be NULL = unguarded): when *skip_if_nonzero != 0 at the time the
 * kernels run, parameters, both moments and (…_dev) the device step counter are left exactly as they were. */
int sg_adam_step_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1, float beta2,
                         float eps, long step, float grad_scale, const int* skip_if_nonzero, hipStream_t stream);
int sg_adam_step_dev_guarded(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                             float beta2, float eps, long long* step_dev, float* corr_dev, float grad_scale,
                             const int* skip_if_nonzero, hipStream_t stream);
/* sg_adam_step_dev_guarded for up to four flat buffers in ONE launch (ABI 8): the optimizers of one training step
 * (train_sdf_autodecoder.py:44-45,90-91 steps the network's and the latent table's one after the other).  Every argument is an array
 * of nsets entries; one guard word (may be NULL) covers all. */
int sg_adam_step_dev_multi(int nsets, float* const* p, const float* const* g, float* const* exp_avg, float* const* exp_avg_sq,
                           const long* n, const float* lr, const float* beta1, const float* beta2, const float* eps,
                           long long* const* step_dev, float* const* corr_dev, const float* grad_scale, const int* skip_if_nonzero,
                           hipStream_t stream);
int sg_clamp(float* p, long n, float lo, float hi, hipStream_t stream);
/* clamp of `ntensors` tensors (host arrays of device pointers and element counts) in one launch per 16 tensors:
 * Discriminator.clip_weights (model/gan.py:67-69) over the module-level surface, eight parameter tensors per critic update */
int sg_clamp_multi(float* const* tensors, const long* counts, int ntensors, float lo, float hi, hipStream_t stream);

/* ---- K8/K9: loss compositions, gradient-penalty pieces, fade-in blend ---------------------------
 * Each loss is one streaming pass + a finishing wave (deterministic, double partial sums) writing a device scalar;
 * each backward is one elementwise pass that reads the upstream scalar gradient `gloss` from device memory.
 * sg_loss_weighted_l1: get_reconstruction_loss, train_autoencoder.py:57-62 (neg_weight 32: d *= 32 where target < 0),
 *   and with neg_weight 1 the DeepSDF data term mean|out - sdf|, train_sdf_autodecoder.py:88.
 * sg_loss_kld: kld_loss, train_autoencoder.py:54-55.
 * sg_loss_meansq: sum_r w_r |x_r|^2 / denom (w = 1 when row_weight is null): the latent regulariser
 *   mean(batch_latent_codes^2), train_sdf_autodecoder.py:88 (row_weight = how often a shape occurs in the batch).
 * sg_gradient_penalty: ((||g_b||_2 - 1)^2).mean() * weight over rows g_b of `grad` [B, M],
 *   train_hybrid_progressive_gan.py:110-111, train_point_gan.py:68-70; sg_lerp_rows: alpha*real + (1-alpha)*fake (:105).
 * sg_fade_blend: fade*x + half_scale*from_SDF(half) (model/progressive_gan.py:48-50): `half` [B,S] lands on channel 0 of
 *   x [B,C,S], the C-1 zero channels of from_SDF are never built (x may be NULL: the embedding alone); sg_channel0 is its
 *   adjoint w.r.t. `half`; sg_subsample2 / _adjoint: x_in[:, ::2, ::2, ::2] (bit-exact index work) and its adjoint. */
size_t sg_loss_workspace_bytes(void);
int sg_loss_weighted_l1_fwd(const float* out, const float* target, long n, float neg_weight, float* loss, void* workspace,
                            size_t workspace_bytes, hipStream_t stream);
int sg_loss_weighted_l1_bwd(const float* out, const float* target, const float* gloss, float* dout, long n, float neg_weight,
                            hipStream_t stream);
/* w_first * mean(x[0, n_first)) + w_rest * mean(x[n_first, n)) and its backward dx = gloss * (w / count) per part, one small
 * launch each: the WGAN losses mean(fake) - mean(real) over the critic's concatenated batch (train_wgan.py:68,
 * train_hybrid_wgan.py:89, train_hybrid_progressive_gan.py:147) and -mean(fake) (train_wgan.py:82; n_first = n, w_first = -1) */
int sg_loss_mean_split_fwd(const float* x, long n, long n_first, float w_first, float w_rest, float* loss, float* dx_unit,
                           hipStream_t stream);   /* dx_unit (optional, [n]): the backward for gloss == 1, from the same launch */
int sg_loss_mean_split_bwd(const float* gloss, float* dx, long n, long n_first, float w_first, float w_rest, hipStream_t stream);
/* Classic-GAN losses on the [B] vector of discriminator outputs, train_gan.py:30,78,84 (torch.nn.functional.binary_cross_entropy
 * against a constant target: mean of -(t max(log p, -100) + (1 - t) max(log(1 - p), -100)), backward g (p - t) / max((1 - p) p,
 * 1e-12) / n as torch computes it) and train_gan.py:65 (-torch.mean(torch.log(p))). */
int sg_loss_bce_fwd(const float* p, long n, float target, float* loss, hipStream_t stream);
int sg_loss_bce_bwd(const float* p, const float* gloss, float* dp, long n, float target, hipStream_t stream);
int sg_loss_neg_mean_log_fwd(const float* p, long n, float* loss, hipStream_t stream);
int sg_loss_neg_mean_log_bwd(const float* p, const float* gloss, float* dp, long n, hipStream_t stream);
/* VAE reparameterisation, model/autoencoder.py:77-82: z = mean + exp(0.5 log_variance) * eps (eps drawn by the caller);
 * backward: d mean = gz (no kernel), d log_variance = gz * eps * 0.5 * exp(0.5 log_variance). */
int sg_vae_reparam_fwd(const float* mean, const float* log_variance, const float* eps, float* z, long n, hipStream_t stream);
int sg_vae_reparam_bwd(const float* log_variance, const float* eps, const float* gz, float* dlog_variance, long n,
                       hipStream_t stream);
/* The critic's last layer together with the activation below it, model/gan.py:54-55 (LeakyReLU -> Conv3d(256 -> 1, kernel 4,
 * stride 1) on a 4^3 grid = one dot product over K = C * 64 values per sample):
 *   fwd  y[n] = bias[0] + sum_k act(z[n, k]) * w[k]          z [N, K] = the PRE-activation of the layer below, act in {none,
 *                                                            leaky, relu} is applied on load
 *   bwd  gz[n, k] = gy[n] * w[k] * act'(z[n, k]);  gw[k] = sum_n gy[n] * act(z[n, k]);  gb[0] = sum_n gy[n];
 *        gbz[c] = sum_{n, s} gz[n, c * S + s]  (the bias gradient of the layer below)     gw / gb / gbz optional, S must be 64
 * One streaming pass each way; every sum in a fixed order. */
int sg_head_dot_fwd(const float* z, const float* w, const float* bias, float* y, int N, long K, int act, float slope,
                    hipStream_t stream);
int sg_head_dot_bwd(const float* z, const float* w, const float* gy, float* gz, float* gw, float* gb, float* gbz, void* gz_image,
                    int N, int C, int S, int act, float slope, hipStream_t stream);
/* gz_image (optional; C a multiple of 128): gz once more, in the A-fragment order of the LDS-halo weight-gradient kernel for 4^3
 * grids, so that the convolution below needs no packing pass over gz — see sg_conv3d_k4s2p1_wgrad_dy_image. */
int sg_loss_kld_fwd(const float* mean, const float* log_variance, long n, float* loss, void* workspace, size_t workspace_bytes,
                    hipStream_t stream);
int sg_loss_kld_bwd(const float* mean, const float* log_variance, const float* gloss, float* dmean, float* dlog_variance,
                    long n, hipStream_t stream);
int sg_loss_meansq_fwd(const float* x, const float* row_weight, long rows, int L, double denom, float* loss, void* workspace,
                       size_t workspace_bytes, hipStream_t stream);
int sg_loss_meansq_bwd(const float* x, const float* row_weight, const float* gloss, float* dx, long rows, int L, double denom,
                       hipStream_t stream);
/* The whole DeepSDF loss of train_sdf_autodecoder.py:88, loss = mean|out - target| + sum_r w_r |z_r|^2 / denom, as ONE pass over
 * both operands + the finishing wave, and its backward (dout[n], dz[rows][L]) as one launch: the arithmetic and its order are
 * those of sg_loss_weighted_l1 with neg_weight 1, sg_loss_meansq and an fp32 add of the two rounded terms, bit for bit. */
int sg_loss_deepsdf_fwd(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                        double denom, float* loss, void* workspace, size_t workspace_bytes, hipStream_t stream);
int sg_loss_deepsdf_bwd(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                        double denom, const float* gloss, float* dout, float* dz, hipStream_t stream);
/* The same loss AND its gradient for an upstream gradient of exactly 1 (dout_unit[n], dz_unit[rows][L]; either may be NULL) in
 * ONE launch: the loss is the root of the trainer's backward (train_sdf_autodecoder.py:88-89).  Bit-identical to
 * sg_loss_deepsdf_fwd / sg_loss_deepsdf_bwd with gloss = 1; the workgroup that arrives last runs the finishing wave.
 * `ticket`: one unsigned, zero before the first call, left zero. */
int sg_loss_deepsdf_fused(const float* out, const float* target, long n, const float* z, const float* row_weight, long rows, int L,
                          double denom, float* loss, float* dout_unit, float* dz_unit, void* workspace, size_t workspace_bytes,
                          unsigned* ticket, hipStream_t stream);
/* voxel_difference, train_autoencoder.py:50-52: count[0] = #{e : (a[e] * b[e]) < 0} with the product rounded to fp32 as the
 * reference's `(input * target) < 0` does (a product that underflows to -0, or a NaN, does not count).  Integer arithmetic
 * throughout: bit-exact.  The caller divides by n (`torch.sum(wrong_signs).item() / wrong_signs.nelement()`). */
int sg_count_sign_mismatch(const float* a, const float* b, long n, long long* count, void* workspace, size_t workspace_bytes,
                           hipStream_t stream);
int sg_gradient_penalty_fwd(const float* grad, long B, long M, float weight, float* norms, float* loss, hipStream_t stream);
int sg_gradient_penalty_bwd(const float* grad, const float* norms, const float* gloss, float* dgrad, long B, long M,
                            float weight, hipStream_t stream);
int sg_lerp_rows(const float* a, const float* b, const float* alpha, float* out, long B, long M, hipStream_t stream);
int sg_fade_blend(const float* x, const float* half, float* out, long B, int C, long S, float fade, float half_scale,
                  hipStream_t stream);
int sg_channel0(const float* g, float* out, long B, int C, long S, float scale, hipStream_t stream);
int sg_subsample2(const float* x, float* out, long B, int R, hipStream_t stream);
int sg_subsample2_adjoint(const float* g, float* out, long B, int R, hipStream_t stream);
/* second-order term of sg_act_bwd for tanh / sigmoid: out = ggx * dy * d(act'(y))/dy  (-2y resp. 1-2y) */
int sg_act_bwd_dy(const float* y, const float* dy, const float* ggx, float* out, long n, int act, hipStream_t stream);

/* ---- input pipeline -------------------------------------------------------------------------
 * reference: VoxelDataset.__getitem__ (datasets.py:16-23): result.clamp_(-clamp, clamp); result /= clamp when
 * rescale_sdf.  out = clamp(x, -clamp, clamp) / divisor on the device (divisor <= 0: no division); x == out allowed.
 * Bit-exact with the reference's CPU arithmetic (NaN-propagating clamp, IEEE fp32 division). */
int sg_voxel_prepare(const float* x, float* out, long n, float clamp, float divisor, hipStream_t stream);

/* ---- PointNet-discriminator GAN family --------------------------------------------------------
 * reference: model/point_sdf_net.py.  SDFGenerator (:49-119): x = lin(x) [+ z_lin(z) per shape]; LayerNorm; ReLU —
 *   y[r] = act(gamma * normalise(x[r] + rowbias[r / rows_per_shape]) + beta), act in {none, relu}; rows may be strided
 *   (ld*), so the skip concat `cat([x, pos])` (:100) is a 259-float row whose first 256 columns LayerNorm writes.
 *   The backward returns dz (gradient of the pre-norm row, also of the per-shape bias rows) and dgamma / dbeta.
 * PointNet (:11-47): `x.max(dim=-2)[0]` over the points of a shape: x [B,P,C] -> out [B,C] + argmax (first
 *   occurrence); scatter = its adjoint (backward), gather = the adjoint of the scatter (double backward under the
 *   gradient penalty, train_point_gan.py:61-70).
 * sg_colsum_tall: out[b][c] = sum_r x[b*batch_stride + r*ld + c]: column sums of `batch` tall [rows, cols] matrices in
 *   two deterministic passes (bias gradients of per-point Linear layers: rows = points; per-shape sums for the
 *   z-injection gradient: batch = shapes). */
int sg_layernorm_fwd(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma,
                     const float* beta, float* y, long ldy, float* mean, float* rstd, long R, int C, float eps, int act,
                     hipStream_t stream);
size_t sg_layernorm_bwd_workspace_bytes(long R, int C);
int sg_layernorm_bwd(const float* x, long ldx, const float* rowbias, long rows_per_shape, const float* gamma, const float* y,
                     long ldy, const float* dy, long lddy, const float* mean, const float* rstd, float* dz, long lddz,
                     float* dgamma, float* dbeta, long R, int C, int act, void* workspace, size_t workspace_bytes,
                     hipStream_t stream);
size_t sg_colsum_tall_workspace_bytes(long batch, long rows, int cols);
int sg_colsum_tall(const float* x, float* out, long batch, long batch_stride, long rows, int cols, long ld, void* workspace,
                   size_t workspace_bytes, hipStream_t stream);
/* PointNet.nn1 (Linear 4 -> 64 -> 128 -> 256 -> 512, ReLU between; model/point_sdf_net.py:14-23) over whole clouds followed by the
 * max over each cloud's points (:40) as ONE fused launch that never writes the per-point layers (ABI 8): out [B][512] = the maxima,
 * idx [B][512] = the point of the cloud that holds each (int32; lowest point on a tie; NaN is ignored by the maximum).  x [B][P][4],
 * P a multiple of 32; `packed`: sg_pointnet_packed_floats() floats from sg_pointnet_pack(nn1.{0,2,4,6}.weight); biases:
 * nn1.{0,2,4,6}.bias.  The plain pass of the sparse-adjoint critic: the recorded evaluation runs on the selected points only. */
size_t sg_pointnet_packed_floats(void);
int sg_pointnet_pack(const float* const* weights, float* packed, hipStream_t stream);
size_t sg_pointnet_select_workspace_bytes(long B, long P);
int sg_pointnet_select(const float* x, const float* packed, const float* const* biases, long B, long P, float* out, int* idx,
                       void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t sg_segmax_workspace_bytes(long B, long P, int C);   /* scratch of sg_segmax_fwd (0 when it needs none) */
int sg_segmax_fwd(const float* x, float* out, int* idx, long B, long P, int C, void* workspace, size_t workspace_bytes,
                  hipStream_t stream);
int sg_segmax_scatter(const float* dy, const int* idx, float* dx, long B, long P, int C, hipStream_t stream);
int sg_segmax_gather(const float* x, const int* idx, float* out, long B, long P, int C, hipStream_t stream);
/* The diagonal last layer of the critic's selected-points pass (ABI 8): row r = b C + c of the gathered batch h [B*C][K] is the point
 * that holds the maximum of channel c, so nn1's last Linear (model/point_sdf_net.py:22) reduces to out[r] = bias[c] + h[r] . w[c]
 * (sg_rowdot); its adjoints out[r][k] = g[r] w[c][k] (sg_rowscale) and out[c][k] = sum_b g[b C + c] h[b C + c][k] (sg_rowouter)
 * make the three closed under differentiation (the gradient penalty of train_point_gan.py:61-70 differentiates twice). */
/* Deterministic adjoint of gathering C rows per group out of x [N][K] (the selected points of every cloud, rows of a group may repeat:
 * several channels can select the same point): dx[rows[b C + c]][k] += g[b C + c][k] without atomics — the first channel that names
 * a row receives the sum of its duplicates through a fixed tree of additions.  dx zeroed by the caller; rows of different groups are
 * distinct; C <= 1024, C K <= 8192. */
int sg_scatter_rows_grouped(const float* g, const int64_t* rows, float* dx, long B, int C, int K, hipStream_t stream);
int sg_rowdot(const float* h, const float* w, const float* bias, float* out, long B, int C, int K, hipStream_t stream);
int sg_rowscale(const float* g, const float* w, float* out, long B, int C, int K, hipStream_t stream);
int sg_rowouter(const float* g, const float* h, float* out, long B, int C, int K, hipStream_t stream);
/* torch_scatter.scatter_max over a ragged `batch` vector (model/point_sdf_net.py:42-43, train_point_gan_ref.py:109-110):
 * out[b][c] = max over rows i with batch[i] == b of x[i][c] (0 for a segment without members, as torch_scatter), arg = the
 * first row attaining it (-1 when empty); scatter = backward (zero-filled), gather = backward of the backward. */
size_t sg_scatter_max_workspace_bytes(long B, int C);
int sg_scatter_max_fwd(const float* x, const int64_t* batch, float* out, int* arg, long N, long B, int C, void* workspace,
                       size_t workspace_bytes, hipStream_t stream);
int sg_scatter_max_scatter(const float* dy, const int* arg, float* dx, long N, long B, int C, hipStream_t stream);
int sg_scatter_max_gather(const float* x, const int* arg, float* out, long N, long B, int C, hipStream_t stream);

/* ---- data-parallel gradient exchange: libshapegan_comm.so ----------------------------------------------
 * reference: nn.DataParallel's gradient reduce-add, train_hybrid_progressive_gan.py:62-68.  One process per GPU; one
 * ncclAllReduce(sum, fp32) of a slice of the flat gradient buffer per call, on the communicator's own stream, ordered after
 * the work already enqueued on `compute_stream` (sg_allreduce_launch) — backward kernels enqueued afterwards overlap with it;
 * sg_allreduce_wait makes `compute_stream` wait for every exchange launched so far.  No host synchronisation.  The library
 * owns the communicator, its stream and two events between init and destroy; buffers stay the caller's.  Rank 0 makes the
 * unique id (sg_allreduce_unique_id) and distributes it out of band (a file, MPI, torch.distributed's store). */
typedef struct sg_comm sg_comm;
const char* sg_comm_last_error(void);
/* libshapegan_comm.so does not link RCCL: sg_comm_bind opens the RCCL the host process names (NULL: "librccl.so.1" by the loader's
 * search order) — for a PyTorch process the one torch.distributed's nccl backend has mapped, so that both live in one RCCL
 * instance — and must be called before any sg_allreduce_* entry.  sg_comm_versions: the NCCL_VERSION_CODE of the header the
 * library was compiled against and the version the bound library reports; a different major version is refused by sg_comm_bind. */
int sg_comm_bind(const char* librccl_path);
int sg_comm_versions(int* header_version, int* runtime_version);
size_t sg_allreduce_unique_id_bytes(void);
int sg_allreduce_unique_id(void* id_out, size_t bytes);
int sg_allreduce_init(sg_comm** comm, int rank, int world, const void* unique_id, size_t id_bytes, int device);
/* read back from the communicator: ranks = ncclCommCount, rank = ncclCommUserRank, device = ncclCommCuDevice, rccl_version =
 * ncclGetVersion (any pointer may be NULL) */
int sg_allreduce_info(sg_comm* comm, int* ranks, int* rank, int* device, int* rccl_version);
int sg_allreduce_launch(sg_comm* comm, float* buf, long count, hipStream_t compute_stream);
int sg_allreduce_wait(sg_comm* comm, hipStream_t compute_stream);
int sg_allreduce_destroy(sg_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* SHAPEGAN_HIP_H */
