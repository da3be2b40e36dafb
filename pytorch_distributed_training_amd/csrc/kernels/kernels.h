// Host-side launchers of the gfx950 kernels.  Pure HIP (no torch headers): the bindings in
// csrc/binding validate shapes / dtypes / alignment and pass raw pointers + the current stream.
#pragma once
#include "../comm/ipc_allreduce.h"
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastdiv.h"

namespace pdt {

enum DType : int { kF32 = 0, kBF16 = 1, kU8 = 2 };

// ---- multi-tensor (device-resident table: rows of 8 x int64 per tensor, then one int64 per
//      4096-element chunk = (tensor_index << 32) | chunk_in_tensor) ---------------------------
constexpr int kMTChunk = 4096;
constexpr int kMTCols = 8;
void launch_multi_flatten(const int64_t* table, int ntensors, int nchunks, float scale, int src_dtype,
                          int dst_dtype, hipStream_t s);
// bucket flatten with the table in the kernel arguments (no device table, no H2D)
constexpr int kFlatMax = 64;
struct FlatArgs {
    const void* src[kFlatMax];
    void* dst[kFlatMax];
    int numel[kFlatMax];
    int chunk_start[kFlatMax + 1];  // prefix sums of 4096-element chunks; chunk_start[n] = total
    int n;
};
void launch_flatten_args(const FlatArgs& a, float scale, int src_dtype, int dst_dtype, hipStream_t s);
// A host table travels to the device in the ARGUMENTS of a tiny kernel (no host-to-device copy: legal while the
// stream is being captured into a hipGraph, and the graph keeps the words).
constexpr int kTableWords = 448;
struct TableWords {
    int64_t w[kTableWords];
};
void launch_write_words(const TableWords& t, int n, int64_t* dst, hipStream_t s);
void launch_multi_adam(const int64_t* table, int ntensors, int nchunks, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bias_corr1, float bias_corr2_sqrt,
                       const float* step_dev /* non-null: bias corrections from this device step count */,
                       int adamw, float grad_scale, int grad_dtype, hipStream_t s);
void launch_multi_sgd(const int64_t* table, int ntensors, int nchunks, float lr, float momentum, float dampening,
                      float weight_decay, int nesterov, int first, float grad_scale, int grad_dtype,
                      hipStream_t s);

// ---- layout / cast ---------------------------------------------------------------------------
// src: f32 NCHW (kF32) or u8 NHWC (kU8) -> dst bf16 NHWC, value * scale
void launch_to_nhwc_bf16(const void* src, int src_dtype, void* dst, int B, int C, int H, int W, float scale,
                         hipStream_t s);
// w [N][T][C] -> wt [C][T][N]  (bf16)
void launch_transpose_ntc(const void* w, void* wt, int N, int T, int C, hipStream_t s);
constexpr int kTransposeMax = 64;
struct TransposeArgs {
    const void* src[kTransposeMax];
    void* dst[kTransposeMax];
    int N[kTransposeMax], T[kTransposeMax], C[kTransposeMax];
    int block_start[kTransposeMax + 1];  // prefix sums of 32x32-tile blocks
    int n;
};
void launch_multi_transpose(const TransposeArgs& a, hipStream_t s);
// out[j] = sum_p slab[p*stride + j], j < n  (fp32, deterministic).  tmp: fp32 [kReduceChunks][n]
// workspace, required when nparts > 64 (two-level reduction).
constexpr int kReduceChunks = 32;
void launch_reduce_slab(const float* slab, int nparts, long stride, int n, float* out, float* tmp, hipStream_t s);
// first level only: out[max_chunks][n] gets <= max_chunks partial rows; returns the row count
// column sums of a row-major bf16 [M][N] matrix (N % 8 == 0) into fp32 partial rows slab[<= max_chunks][N];
// returns the number of rows written
int launch_colsum_bf16(const void* x, long M, int N, float* slab, int max_chunks, hipStream_t s);
int launch_reduce_rows_l1(const float* slab, int nparts, long stride, int n, float* out, int max_chunks,
                          hipStream_t s);

// ---- softmax cross-entropy (mean), fwd + bwd in one pass -------------------------------------
void launch_softmax_ce(const void* logits, int dtype, const int64_t* labels, int B, int N, float* row_loss,
                       void* dlogits, float* loss_out, hipStream_t s);

// ---- batch norm (NHWC bf16, per-channel vectors fp32) ----------------------------------------
int bn_parts(long M, int C);  // number of partial rows the stats / bwd-reduce kernels write
void launch_bn_stats(const void* x, long M, int C, float* slab /*[parts][2][C]*/, int parts, hipStream_t s);
// stats: partial rows [nparts][2][C] (nparts <= 512).  Sums them, writes mean / invstd, updates the
// running stats and num_batches_tracked (each optional) and optionally the reduced sums [2][C].
// mb (SyncBatchNorm over IPC mailboxes, world > 1): the kernel exchanges (count, mean, M2) with the peers itself and
// writes the global count to count_out[0]; null: single-rank statistics
void launch_bn_finalize(const float* stats, int nparts, int C, double count, const float* count_dev, float eps,
                        float momentum, float* running_mean, float* running_var, long* num_batches_tracked,
                        float* mean, float* invstd, float* sums_out, hipStream_t s, const IpcMailboxes* mb = nullptr,
                        float* count_out = nullptr);
// relu: also writes the 1-bit-per-element activity mask [M][C/8] that the backward kernels take as `z`
// per-channel BatchNorm coefficients of a tensor that enters another kernel un-normalised (all null: none)
struct BnAffine {
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;
};
// backward of the fused stem tail (even H, W; Ho = H/2, Wo = W/2): both BN-backward passes gather the pooled gradient
// themselves, the full-resolution gradient never exists.  slab [parts][2][C], sums_out [2][C], coef [3][C] workspaces.
void launch_bn_relu_maxpool_bwd(const void* dy, const uint8_t* argmax, const void* u, const uint8_t* mask, void* du, int B,
                                int H, int W, int C, int Ho, int Wo, const float* mean, const float* invstd,
                                const float* gamma, float count, const float* count_dev, float* slab, int parts,
                                float* sums_out, float* coef, hipStream_t s, const IpcMailboxes* mb = nullptr);
void launch_bn_apply(const void* x, const void* residual, void* z, uint8_t* mask /*[M][C/8] or null*/, long M, int C,
                     const float* mean, const float* invstd, const float* gamma, const float* beta, int relu,
                     const BnAffine& res_affine, hipStream_t s);
// stem tail fused: y = maxpool3x3/2/1(relu(bn(u))) without the full-resolution activation in memory; also writes
// the winning taps and the 1-bit ReLU mask [B*H*W][C/8] of the full-resolution pixels
void launch_bn_relu_maxpool_fwd(const void* u, void* y, uint8_t* argmax, uint8_t* mask, int B, int H, int W, int C,
                                int Ho, int Wo, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, hipStream_t s);
void launch_bn_bwd_reduce(const void* dz, const void* z, const void* x, long M, int C, const float* mean,
                          const float* invstd, int relu, float* slab /*[parts][2][C]*/, int parts,
                          hipStream_t s);
void launch_bn_bwd_elemt(const void* dz, const void* z, const void* x, void* dx, void* dres, long M, int C,
                         const float* mean, const float* invstd, const float* gamma, const float* psums /*[nps][2][C]*/,
                         int nps, float* sums_out /*[2][C] or null*/, float count, const float* count_dev, int relu,
                         float* coef /*[3][C] workspace or null*/, hipStream_t s,
                         const IpcMailboxes* mb = nullptr /* SyncBatchNorm: global sums exchanged in the kernel */);

// Two BatchNorms fed by ONE gradient (bottleneck bn3 + downsample BN): dz and the ReLU mask (may be null) are read
// once; per side the partial sums of its own reduction, outputs dx, the summed [2][C] and a [3][C] workspace.
struct BnBwdSide {
    const void* x;
    void* dx;
    const float *mean, *invstd, *gamma, *psums;
    int nps;
    float *sums_out, *coef;
};
void launch_bn_bwd_elemt2(const void* dz, const void* mask, long M, int C, float count, const float* count_dev,
                          const BnBwdSide& a, const BnBwdSide& b, hipStream_t s, const IpcMailboxes* mb = nullptr);

// ---- LayerNorm over the last dim of [M, D] bf16 (layernorm.hip) -----------------------------------
bool layernorm_dim_ok(int D);
int layernorm_bwd_blocks(long M);  // rows of the dgamma / dbeta partial slab
void launch_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          long M, int D, float eps, hipStream_t s);
// slab: fp32 [blocks][2][D] = per-block partial [dbeta, dgamma]
void launch_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                          void* dx, float* slab, int blocks, long M, int D, const void* addend /* or null */,
                          hipStream_t s);

// ---- fused self-attention (attention.hip): head_dim 32 / 64 / 128, any sequence length ------------------
// (head_dim 64 up to 256 tokens runs the whole-head kernels, everything else the block-streaming ones)
void set_attention_pipelined(bool on);  // whole-head launches of 5+ tiles: persistent pipelined kernels (default) or one head per workgroup
bool attention_shape_ok(int S, int head_dim);
bool attention_needs_dsum(int S, int head_dim);  // backward wants an fp32 [B,H,S] scratch (block-streaming kernels)
// qkv [B,S,3,H,hd] bf16 -> out [B,S,H,hd] bf16, lse [B,H,S] fp32; B * H <= 65535
void launch_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int head_dim, hipStream_t s);
void launch_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                          float* dsum /* [B,H,S] or null */, int B, int S, int H, int head_dim, hipStream_t s);

// ---- pools (NHWC bf16) -----------------------------------------------------------------------
void launch_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C, int Ho, int Wo,
                             hipStream_t s);
void launch_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C, int Ho,
                             int Wo, hipStream_t s);
void launch_global_avgpool_fwd(const void* x, void* y, int B, int HW, int C, hipStream_t s);
void launch_global_avgpool_bwd(const void* dy, void* dx, int B, int HW, int C, hipStream_t s);

// ---- implicit-GEMM convolution on MFMA --------------------------------------------------------
constexpr int kMaxTaps = 49;
struct ConvGeom {
    // input tensor (activations for fwd, dY for dgrad): [B, Hin, Win, C] bf16
    int B, Hin, Win, C;
    // output tensor: [B, Hout, Wout, N] bf16; this launch writes pixels (i*os + ooh, j*os + oow), i<nOH, j<nOW
    int Hout, Wout, N;
    int nOH, nOW, os, ooh, oow;
    int is;  // input step per output index: in_h = i*is + dh[t]
    int ntaps;        // taps of this launch
    int wtaps;        // T of the weight tensor [N][wtaps][C]
    // per tap: input offsets dh, dw (signed bytes) and the weight slot, packed in ONE dword so that the kernels
    // fetch a tap with a scalar load at a runtime index (byte tables would become vector loads)
    int32_t tap[kMaxTaps];
    void set_tap(int t, int dh, int dw, int wslot) {
        tap[t] = (dh & 0xff) | ((dw & 0xff) << 8) | ((wslot & 0xff) << 16);
    }
    // optional bf16 tensor with the output's shape that is added in the epilogue (fuses the residual
    // branch's gradient into the data-gradient kernel instead of a separate elementwise add)
    const void* addend;
    // optional 1-bit-per-element gate of the addend ([pixels, N/8] bytes, the activity mask of a BN+ReLU): the
    // epilogue adds addend where the bit is set.  Lets the residual branch hand over the UNMASKED gradient that
    // reached the block output plus that ReLU's mask instead of materialising the masked copy.
    const uint8_t* addend_mask;
    // addend_stride == 2: the addend is the COMPACT data gradient of a 1x1 stride-2 convolution of the same input
    // (the downsample branch): shape [B, ceil(Hout/2), ceil(Wout/2), N], added at the even (h, w) positions only.
    // Saves zero-filling a full-size tensor and reading it back (3/4 of it zeros).
    int addend_stride;
    // multiply-shift dividers for the pixel decompositions m -> (b, i, j) (filled by launch_conv_igemm): by nOW, nOH,
    // Wout, Hout.  64-bit '/' and '%' by runtime values cost ~40 vector instructions each; a short-K launch did
    // eight of them per thread in its prologue and up to 32 in its epilogue.
    FastDiv div_now, div_noh, div_wout, div_hout;
    // (set by launch_conv_igemm) the output tensor is large: store it with the non-temporal hint (see common.h)
    int stream_out;
    // elements per input pixel in memory; 0 = C (dense NHWC).  The packed stem runs with C = 32 (one filter
    // row: 8 kw slots x 4 channels, contiguous in memory) over pixels of 4 elements.
    int pix_elems;
    // BatchNorm-backward reduction fused into a data-gradient epilogue: when bnr_x != nullptr the output of this
    // launch is the gradient dz w.r.t. the OUTPUT of a BN+ReLU whose input was bnr_x (same shape as the output)
    // and whose 1-bit activity mask is bnr_mask; the epilogue then also writes, per m-tile, the partial rows
    // [sum(dz*m), sum(dz*m*xhat)] to the slab argument, so the standalone bn_bwd_reduce pass (a full re-read of
    // dz and x) is not needed.
    const void* bnr_x;
    const uint8_t* bnr_mask;
    const float* bnr_mean;
    const float* bnr_invstd;
};
// y = conv(x, w); when stats_slab != nullptr also writes per-(m-tile) partial [sum, sum_sq] rows
// stats_slab[m_tile][2][N] computed from the bf16-rounded outputs.
// cfg: 0 = tuned heuristic; 6 / 8 / 11 / 12 = 128-row tile builds (4-stage ring / 3-stage ring / 2-stage 128-register /
// 3-stage 128-register), 10 = 256x128 tile, 15 = 256x64 tile, 5 / 9 = register-staged builds (see conv_igemm.hip)
int conv_igemm_m_tiles(const ConvGeom& g, int* bm_out, int cfg = 0);
bool stream_nt(long bytes);  // bn.hip: non-temporal accesses from this tensor size on
void launch_conv_igemm(const void* x, const void* w, void* y, const ConvGeom& g, float* stats_slab,
                       hipStream_t s, int cfg = 0);

struct WgradGeom {
    int B, Hin, Win, C;    // x: [B,Hin,Win,C]
    int Hout, Wout, N;     // dy: [B,Hout,Wout,N]
    int KH, KW, stride, pad;
    int splits;            // M-splits (slab rows)
};
int conv_wgrad_splits(const WgradGeom& g);
// slab[split][N][KH*KW][C] fp32 partials (to be summed by launch_reduce_slab)
void launch_conv_wgrad(const void* dy, const void* x, float* slab, const WgradGeom& g, hipStream_t s);


// 64 -> 64 channel 3x3 / 1 / 1 weight gradient as a TN product out of an LDS window (wgrad_win64.hip);
// slab [wgrad_win64_grid(B, H)][64][3][3][64] fp32, one partial per workgroup
bool wgrad_win64_ok(int B, int H, int W, int C, int N, int KH, int KW, int stride, int pad);
int wgrad_win64_grid(int B, int H);
void launch_wgrad_win64(const void* dy, const void* x, float* slab, int B, int H, int W, hipStream_t s);
void set_wgrad_win64_mode(int mode);  // -1: PDT_WGRAD_WIN64 / work-size rule, 0 off, 2 whenever the geometry fits

// all-taps-per-workgroup weight gradient (conv_wgrad_win.hip): mode 0 = 3x3-style convs with C, N % 64 == 0;
// mode 1 = the packed 3-channel stem (source pixel = 2 image pixels x 4 channels, tap = filter row)
struct WinGeom {
    int B, Ho, Wo, N;     // dy [B,Ho,Wo,N]
    int Hs, Ws, xpix;     // x source [B,Hs,Ws,xpix]
    int C;                // output columns per tap (conv: Cin; stem: 32)
    int KH, KW;           // taps (stem: 7 x 1)
    int sh, sw, ph, pw;   // stride / padding in source pixels
    int mode;
    int R, splits;        // filled by wgrad_win_plan
};
bool wgrad_win_plan(WinGeom& g);  // false: shape not supported by the window kernel
// slab: fp32 [splits][N][KH*KW][C] (mode 0) or [splits][64][7][7][3] (mode 1)
void launch_wgrad_win(const void* dy, const void* x, float* slab, const WinGeom& g, hipStream_t s);

// ---- bf16 GEMM C = epilogue(A . B^T) on MFMA (gemm.hip): Linear fwd / dgrad, 1x1 convolutions -----------
struct GemmArgs {
    const void* A;       // [M][lda] bf16, K contiguous
    const void* B;       // [N][ldb] bf16, K contiguous
    void* C;             // [M][ldc] bf16
    long M;
    int N, K;            // N % 8 == 0, K % 8 == 0
    int lda, ldb, ldc;   // elements, multiples of 8
    const float* bias;   // [N] fp32 or null: added in fp32 before the rounding to bf16
    const void* residual;  // [M][ldr] bf16 or null: added in the store pass (after the activation)
    int ldr;
    const void* gelu_in;   // [M][ldu] bf16 or null: the stored value is multiplied by GELU'(gelu_in)
    int ldu;
    void* preact_out;      // [M][ldc] bf16 or null: the value before the activation is written here too
    int act;               // 0 none, 1 GELU (erf form)
    float* colsum_slab;    // [m_tiles][N] fp32 or null: per-m-tile column sums of the stored values
    // pointwise convolution in NHWC (ld == channels, no bias): 1 = forward with the BatchNorm statistics of the
    // rounded output, 2 = data gradient with a gated addend and / or the fused BatchNorm-backward sums
    int conv_mode;
    float* stats_slab;           // [m_tiles][2][N] fp32: mode 1 (sum, sum of squares), mode 2 with bnr_x (sum g, sum g*xhat)
    const void* addend;          // [M][N] bf16 or null
    const uint8_t* addend_mask;  // [M][N/8] one bit per element or null (ungated)
    const void* bnr_x;           // [M][N] bf16 or null: input of the BatchNorm whose backward sums are taken
    const uint8_t* bnr_mask;     // [M][N/8]: activity of the ReLU behind that BatchNorm
    const float* bnr_mean;       // [N]
    const float* bnr_invstd;     // [N]
    // mode 2, optional SECOND BatchNorm fed by the same gradient through the same ReLU gate (bottleneck bn3 + the BN of
    // the downsample branch): stats_slab2 [m_tiles][2][N] receives (sum g, sum g * xhat2)
    const void* bnr_x2;          // [M][N] bf16 or null
    const float* bnr_mean2;      // [N]
    const float* bnr_invstd2;    // [N]
    float* stats_slab2;          // [m_tiles][2][N]
    // spatial filter as implicit GEMM (stride 1, output grid == input grid H x W, rows of A are pixels): K runs over
    // ntaps x (K / 64) tiles; tap t reads pixel (y + tap_dh, x + tap_dw) of the same image (zeros outside) and the
    // weight columns [tap_w * K, (tap_w + 1) * K) of B, whose rows are ldb = wtaps * K long.  ntaps == 0: plain GEMM.
    int ntaps, H, W;
    int a_stream;  // (set by launch_gemm_nt) one tile column: A is read exactly once -> non-temporal DMA
    int32_t tap[9];  // (dh & 0xff) | (dw & 0xff) << 8 | wslot << 16
    FastDiv div_w, div_hw, div_kt;  // / W, / (H * W), / (K / 64)
    FastDiv div_band, div_gm, div_last;  // (launcher) tile index -> band / row in a full band / row in the last band
    int last_band, last_rows;            // (launcher) index of the last band of tile rows and its row count
    int bn;                         // tile width: 0 = the launcher picks (gemm_tile_width), 128 or 256
    int m_tiles, n_tiles, group_m;  // filled by the launcher
    int debug_nostore;              // timing diagnostics, bit mask: 1 output stores dropped (zero-size descriptor),
                                    // 2 epilogue skipped, 4 MFMAs skipped, 8 operand DMA skipped
};
// 3x3 / stride 1 / pad 1 with 64 -> 64 channels: weights in registers, input window staged once per tile (conv_win64.hip)
bool conv_win64_ok(int C, int N, int KH, int KW, int stride, int pad, int H, int W);
int conv_win64_grid(int B, int H);
void launch_conv_win64(const void* x, const void* w, void* y, float* slab, int B, int H, int W, int mirror,
                       const void* bnr_x, const uint8_t* bnr_mask, const float* bnr_mean, const float* bnr_invstd,
                       hipStream_t s);
// weight gradients of layers with Cout % 256 == 0 and Cin % 256 == 0 on 256 x 256 tiles (gemm_tn.hip); same slab
// layout and split-K contract as launch_conv_wgrad, which dispatches to it
bool gemm_tn_ok(const WgradGeom& g);
void set_gemm_tn_min_rows(long rows);  // reduction length from which it is used (default 8192; tests lower it)
int gemm_tn_splits(const WgradGeom& g);
void launch_gemm_tn(const void* dy, const void* x, float* slab, const WgradGeom& g, hipStream_t s);
int gemm_m_tiles(long M);
void set_gemm_tile_width(int bn);  // process-wide override of the choice (0: by shape; tests and A/B runs)
int gemm_tile_width(const GemmArgs& a);  // what launch_gemm_nt will use for this launch (a.bn == 0: by shape)
void launch_gemm_nt(const GemmArgs& a, hipStream_t s);

// ---- ViT glue (vit.hip) -----------------------------------------------------------------------------------
// x NHWC bf16 [B][G*P][G*P][3] -> patches [B*G*G][P*P*3] in (py, px, c) order; P % 8 == 0
void launch_patchify(const void* x, void* out, long B, int G, int P, hipStream_t s);
// out [B][T+1][D] = cat(cls, patch) + pos      (cls [D], pos [T+1][D] fp32; D % 8 == 0)
void launch_vit_embed_fwd(const void* patch, const float* cls, const float* pos, void* out, long B, int T, int D,
                          hipStream_t s);
// dx [B][T+1][D] -> dpatch [B][T][D], slab [slices][T+1][D] fp32 partial batch sums of dx
void launch_vit_embed_bwd(const void* dx, void* dpatch, float* slab, int B, int T, int D, int slices, hipStream_t s);

// ---- 3-channel stem (conv_stem.hip) -----------------------------------------------------------
// x NHWC bf16 [B,H,W,3] -> xp [B][H+6][W+8][4] (image at row +3 / col +4, zeros elsewhere, channel 3 = 0)
void launch_stem_pack_x(const void* x, void* xp, int B, int H, int W, hipStream_t s);
// w [64][7][7][3] -> wp [64][7][8][4] (kw slot 0 and channel 3 are zero)
void launch_stem_pack_w(const void* w, void* wp, int N, hipStream_t s);
// forward on the packed image as a window kernel (conv_stem_win.hip): weights in registers, 8 output rows per tile;
// slab [conv_stem_win_grid(B, Ho)][2][64] fp32 (per-channel sum / sum of squares of the rounded outputs) or null
bool conv_stem_win_ok(int B, int Ho, int Wo, int N);
int conv_stem_win_grid(int B, int Ho);
void launch_conv_stem_win(const void* xp, const void* wp, void* y, float* slab, int B, int Ho, int Wo, hipStream_t s);
// weight gradient on the packed image (same file): slab [conv_stem_wgrad_win_grid(B, Ho)][64][147] fp32 partials
bool conv_stem_wgrad_win_ok(int B, int Ho, int Wo, int N);
int conv_stem_wgrad_win_grid(int B, int Ho);
void launch_conv_stem_wgrad_win(const void* dy, const void* xp, float* slab, int B, int Ho, int Wo, hipStream_t s);
void set_stem_win_mode(int mode);  // -1: PDT_STEM_WIN / work-size rule, 0 off, 2 whenever the geometry fits

}  // namespace pdt
