/*
 * unetr_hip.h -- C ABI of libunetr_hip.so, the MI355X (gfx950) kernels behind UNETR's training hot path.
 *
 * The reference (ilkyyldz95/3DmedicalImageSegmentation) has no native code and no FFI: its hot path is
 * `UNETR(nn.Module)` (unetr.py:21-208) delegating to MONAI 0.6.0 blocks and torch ATen ops.  The boundary
 * a maintainer binds is therefore the nn.Module; this header is the C-ABI layer that
 * module calls through ctypes.  Each entry point cites the reference call it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless stated; nothing is allocated or freed here;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no host synchronisation;
 *   - return value 0 = ok, non-zero = UNETR_ERR_* (the Python shim raises RuntimeError);
 *   - `prec`: 0 = fp32 (v_mfma_f32_16x16x4_f32, bit-exact fp32 fma chains), 1 = bf16 operands with fp32
 *     accumulation (v_mfma_f32_16x16x32_bf16), 2 = "bf16x3": fp32 storage like prec 0, every operand element split into a
 *     bf16 (hi, lo) pair inside the kernels and contracted with two bf16 MFMAs per chunk pair (hi*hi + hi*lo + lo*hi + lo*lo,
 *     fp32 accumulation): ~16-bit products at four times the fp32 matrix rate -- the tolerance-grade mode (1e-3 on logits);
 *   - ACTIVATION STORAGE: on the conv side (3x3x3 / 1x1x1 / transposed convs, InstanceNorm, concat copies, out conv)
 *     every feature map and feature-map gradient is stored in the precision mode's activation type: fp32 with prec 0,
 *     **bf16 with prec 1** (half the HBM bytes of passes that are bandwidth-bound; statistics, weights, partial sums,
 *     logits and the input image stay fp32).  Entry points that take `prec` infer the type from it (`const void*`
 *     operands); entry points without one take `act16` (1 = bf16 feature maps).  Rows of a bf16 feature map are 16-byte
 *     aligned (pitch % 8 == 0).  The token side (ViT) keeps an fp32 residual stream plus bf16 operand copies;
 *   - feature maps are channels-last: [B, D, H, W, C] with a row pitch `ld` (floats between voxels), so a
 *     producer can write straight into one half of a concatenation buffer (torch.cat at MONAI
 *     UnetrUpBlock.forward, used by unetr.py:203-206, becomes free);
 *   - token matrices are [B*L, H] row-major, which IS the channels-last view of unetr.py:177-180 proj_feat.
 */
#ifndef UNETR_HIP_H
#define UNETR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature or a descriptor struct changes.  A binding (3dmedicalimagesegmentation_amd/_capi.py, or a
 * C caller) must compare unetr_abi_version() with the UNETR_ABI_VERSION it was written against before its first call: a stale
 * .so would otherwise shift arguments silently (a stream pointer in an int slot). */
#define UNETR_ABI_VERSION 17
int unetr_abi_version(void);

/* ---- generic MFMA GEMM: C[M,N] = epilogue(A[M,K] * B[K,N]) ------------------------------------------
 * replaces torch.nn.Linear forward/backward inside MONAI ViT (PatchEmbeddingBlock Linear, SABlock.qkv,
 * SABlock.out_proj, MLPBlock.linear1/2 -- built at unetr.py:78-89) and the 1x1x1 Conv3d of UnetResBlock.conv3
 * (unetr.py:90-174).  a_trans=0: A[m*lda+k]; a_trans=1: A[k*lda+m].  b_trans=0: B[n*ldb+k] (a torch
 * Linear weight [N,K]); b_trans=1: B[k*ldb+n]. */
typedef struct {
    int M, N, K, batch;
    int a_trans, b_trans;
    long lda, ldb, ldc;
    long strideA, strideB, strideC;   /* per-batch element strides */
    const float* bias;                /* [N] or NULL */
    const float* res;                 /* residual added after activation, row index = m % res_mod */
    long ldr, strideR;
    int res_mod;
    float* pre;                       /* optional copy of the pre-activation value (same ld as C) */
    const float* aux;                 /* act==2: multiply by gelu'(aux[m*ldaux+n]) */
    long ldaux;
    int act;                          /* 0 none, 1 exact-erf GELU, 2 GELU backward */
    int accumulate;                   /* C += result */
    float alpha;
    int prec;
    int b_x3words;                    /* prec == UNETR_PREC_BF16X3 only: B holds pre-split words [hi | lo << 16] (unetr_split_words of the
                                       * fp32 weight, same layout and pitch) instead of fp32 values -- the optimizer-maintained word shadow */
} unetr_gemm_desc;
int unetr_gemm(const unetr_gemm_desc* d, const float* A, const float* B, float* C,
               float* ws, size_t ws_bytes, void* stream);
/* bf16x3 mode: dst[i] = [hi | lo << 16] with hi = src[i] truncated to bf16, lo = bf16(src[i] - hi) (n % 4 == 0, 16-byte aligned):
 * the word shadow of a weight arena, re-derived after every optimizer step, that unetr_gemm reads with b_x3words = 1 */
int unetr_split_words(const float* src, void* dst, long n, void* stream);

/* ---- bf16-STORED operand GEMM (LDS-DMA staged): the same nn.Linear forward / data-gradient calls as unetr_gemm
 * (MONAI ViT Linear layers built at unetr.py:78-89) for the bf16 precision mode, where LayerNorm / attention / GELU
 * epilogues and the optimizer's weight shadow already hold the operands as bf16.
 *   C[M,N] = epilogue(A[M,K] * Bop):  A bf16 [M,K] (k contiguous, pitch lda);
 *   b_kn = 0: B bf16 [N,K] (pitch ldb; forward y = x W^T);  b_kn = 1: B bf16 [K,N] (pitch ldb; dgrad dx = dy W).
 * K must be a multiple of 64; pitches multiples of 8 elements.  Outputs: fp32 C (pitch ldc; may be NULL) and/or
 * bf16 Cb (pitch ldcb; may be NULL); `pre`/`accumulate` need C.  Epilogue fields as in unetr_gemm_desc. */
typedef struct {
    int M, N, K, b_kn;
    long lda, ldb, ldc, ldcb;
    const float* bias;
    const float* res;
    long ldr;
    int res_mod;
    float* pre;
    const float* aux;
    long ldaux;
    int act;
    int accumulate;
    float alpha;
    /* tc_cout > 0: the output is a 2x2x2 stride-2 ConvTranspose3d result scattered in place (UnetrPrUpBlock / UnetrUpBlock,
     * unetr.py:99-174): row m is voxel (b, z, y, x) of the INPUT grid [*, tc_d, tc_h, tc_w], column n = tap * tc_cout + co (the
     * tap-major weight pack, kind 4 of unetr_conv3_pack_grouped), and the value goes to Cb[outvox(m, tap) * ldcb + co] with
     * outvox = (b, 2z + tap/4, 2y + (tap/2)%2, 2x + tap%2) of the output grid.  bf16 output only (C and pre NULL). */
    int tc_d, tc_h, tc_w, tc_cout;
    /* x3 != 0: bf16x3 precision mode through these entry points (what the LayerNorm-riding forms unetr_gemm_bf16_ln_fwd / _ln_bwd need):
     * A is fp32 [M,K], B fp32 (x3 = 1) or pre-split words (x3 = 2: unetr_split_words), both in the b_kn layout rule above; C fp32 only;
     * K % 32 == 0, pitches % 4 == 0.  UNETR_ERR_UNSUPPORTED when the LDS-DMA kernel declines the shape (use unetr_gemm + the plain
     * LayerNorm entry points then). */
    int x3;
} unetr_gemm_bf16_desc;
int unetr_gemm_bf16(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C, void* Cb,
                    float* ws, size_t ws_bytes, void* stream);
/* fp32 -> bf16 round-to-nearest-even copy (weight shadows; torch .to(torch.bfloat16) semantics) */
int unetr_cast_bf16(const float* src, void* dst, long n, void* stream);
/* bf16x3 mode, weight gradients: fp32 [rows, cols] -> bf16 [3 rows, cols] stack of its (hi, lo) halves -- second == 0: [hi; hi; lo]
 * (the dY side), second == 1: [hi; lo; hi] (the X side) -- so that dY^T X over the stacked rows = hi*hi + hi*lo + lo*hi on the
 * bf16 grouped weight-gradient kernel (unetr_gemm_bf16_grouped_wgrad[_adamw]).  rows * cols % 8 == 0, 16-byte aligned. */
int unetr_split_stack_bf16(const float* src, void* dst, long rows, long cols, int second, void* stream);
/* every stack of a backward pass in one launch */
typedef struct { const float* src; void* dst; long rows, cols; int second; } unetr_split_problem;
int unetr_split_stack_bf16_grouped(const unetr_split_problem* probs, int n, void* stream);
/* out = a + b (fp32, n % 4 == 0, 16-byte aligned) and out_bf16 = bf16(out) (may be NULL): the sum autograd forms for a hidden
 * state with two consumers (unetr.py:197-201: hidden states 3, 6, 9 feed the next block AND encoder2-4) */
int unetr_add_cast_bf16(const float* a, const float* b, float* out, void* out_bf16, long n, void* stream);

/* Grouped launches for the work that is OFF the critical path of backward at batch 2: the weight gradients
 * dW_i[N_i,K_i] = dY_i[M_i,N_i]^T * X_i[M_i,K_i] of all transformer blocks (torch.nn.Linear backward, MONAI
 * ViT built at unetr.py:78-89) and the bias / position-embedding gradient column sums.  One launch covers up to
 * 48 problems (descriptors travel in the kernel-argument block), filling the chip without split-K slabs. */
typedef struct { const float* dy; const float* x; float* dw; int M, N, K; } unetr_grouped_problem;
int unetr_gemm_grouped_wgrad(const unetr_grouped_problem* probs, int n, int prec, void* stream);
/* the same grouped weight gradients on bf16-STORED dy / x (the pointers in unetr_grouped_problem then address bf16 data,
 * dense row-major [M,N] / [M,K]; M, N, K multiples of 8): LDS-DMA staging, transposing LDS reads for both operands */
int unetr_gemm_bf16_grouped_wgrad(const unetr_grouped_problem* probs, int n, void* stream);
/* x_bf16 != 0: x addresses bf16 data (the bf16 twin of a data gradient: fc1's bias gradient is summed from the same bf16 values its
 * weight gradient is formed from, and the fp32 copy of that gradient is never written) */
typedef struct { const void* x; float* out; long ld; int M, N, x_bf16; } unetr_colsum_problem;
int unetr_colsum_grouped(const unetr_colsum_problem* probs, int n, void* stream);

/* ---- 2x2x2 stride-2 transposed conv (nn.ConvTranspose3d, bias=False; unetr.py:99-174) ---------------
 * x: [B,D,H,W,Cin] pitch ldx; w: torch layout [Cin,Cout,2,2,2]; y: [B,2D,2H,2W,Cout] pitch ldy. */
int unetr_tconv_fwd(const float* x, long ldx, const float* w, float* y, long ldy,
                    int B, int D, int H, int W, int Cin, int Cout, int prec,
                    float* ws, size_t ws_bytes, void* stream);
int unetr_tconv_dgrad(const float* dy, long ldy, const float* w, float* dx, long ldx, int accumulate,
                      int B, int D, int H, int W, int Cin, int Cout, int prec,
                      float* ws, size_t ws_bytes, void* stream);
int unetr_tconv_wgrad(const float* x, long ldx, const float* dy, long ldy, float* dw,
                      int B, int D, int H, int W, int Cin, int Cout, int prec,
                      float* ws, size_t ws_bytes, void* stream);
/* Dedicated kernels for the same transposed conv at large-volume / small-channel layers (persistent workgroups over
 * 64-voxel tiles, LDS table of output-voxel bases, dy gathered voxel-major and read through transposing LDS loads).
 * Same arguments and results as unetr_tconv_fwd / unetr_tconv_wgrad; return "unsupported" (3) for shapes outside their
 * range (M = B*D*H*W < 2048, channel counts, alignment) -- the caller then uses the generic entry points above. */
int unetr_tconv2_fwd(const void* x, long ldx, const float* w, void* y, long ldy,
                     int B, int D, int H, int W, int Cin, int Cout, int prec,
                     float* ws, size_t ws_bytes, void* stream);
int unetr_tconv2_wgrad(const void* x, long ldx, const void* dy, long ldy, float* dw,
                       int B, int D, int H, int W, int Cin, int Cout, int prec,
                       float* ws, size_t ws_bytes, void* stream);
int unetr_tconv2_dgrad(const void* dy, long ldy, const float* w, void* dx, long ldx, int accumulate,
                       int B, int D, int H, int W, int Cin, int Cout, int prec,
                       float* ws, size_t ws_bytes, void* stream);
/* unetr_tconv2_wgrad without its reduce launch (see unetr_conv3_wgrad_parts): part [rows][Cin Cout 8] */
long unetr_tconv2_wgrad_rows(int B, int D, int H, int W, int Cin, int Cout);
int unetr_tconv2_wgrad_parts(const void* x, long ldx, const void* dy, long ldy, float* part, size_t part_bytes, long* rows_out,
                             int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream);
int unetr_tconv2_fwd_supported(long M, int Cin, int Cout, long ldx, long ldy);
int unetr_tconv2_wgrad_supported(long M, int Cin, int Cout, long ldx, long lddy);


/* ---- the same attention core on bf16-STORED q/k/v (bf16 mode, head dim 64): qkv bf16 [B*L, 3*heads*64] as written by
 * unetr_ln_gemm_bf16, out fp32 and/or bf16 [B*L, heads*64] (either may be NULL), lse [B, heads, L].  Backward: out_bf16 /
 * dout_bf16 bf16 [B*L, heads*64], writes dqkv_bf16 (required) and dqkv fp32 (optional), delta [B, heads, L] scratch.
 * Returns "unsupported" (3) for other head dims -- the caller then uses unetr_attention_fwd / _bwd on fp32 q/k/v. */
int unetr_attention_bf16_fwd(const void* qkv, float* out, void* out_bf16, float* lse, int B, int L, int heads, int dh,
                             float scale, void* stream);
int unetr_attention_bf16_bwd(const void* qkv, const void* out_bf16, const void* dout_bf16, const float* lse, float* dqkv,
                             void* dqkv_bf16, float* delta, int B, int L, int heads, int dh, float scale, void* stream);

/* ---- LayerNorm fused into the GEMM that consumes it (bf16 mode, small token counts): y = act(LayerNorm(x) W^T + bias).
 * The two calls per transformer block it replaces in MONAI's TransformerBlock.forward (built at unetr.py:78-89):
 * attn.qkv(norm1(x)) and mlp.linear1(norm2(x)) followed by GELU.  x fp32 [M,K] (pitch ldx), gamma / beta [K], W bf16 [N,K]
 * (pitch ldw, K contiguous).  Outputs (any may be NULL except that one of C / Cb is required): C fp32 / Cb bf16 [M,N]
 * after the activation (act: 0 none, 1 exact-erf GELU), pre fp32 [M,N] before it; xn bf16 [M,K] = the normalised rows,
 * mean / rstd [M] (the LayerNorm backward's and the weight gradient's inputs).  K % 64 == 0, K <= 1024, N % 4 == 0. */
typedef struct {
    const float* x; long ldx;
    const float* gamma; const float* beta; float eps;
    const void* W; long ldw;
    const float* bias; int act;
    float* pre; long ldpre;
    void* Cb; long ldcb;
    float* C; long ldc;
    void* xn; float* mean; float* rstd;
    int M, N, K;
} unetr_ln_gemm_desc;
int unetr_ln_gemm_bf16(const unetr_ln_gemm_desc* d, void* stream);

/* The small transposed convs (Cin a multiple of 64, bf16 mode) run as plain unetr_gemm_bf16 calls on torch's own weight
 * matrix [Cin, Cout*8]; these two move between the GEMM-side matrix t / g [M, Cout*8] (column = co*8 + tap) and the
 * voxel-major tensor y / dy [B, 2D, 2H, 2W, Cout] (pitch ldy): the pixel shuffle that is left of the "transposed conv". */
int unetr_pixel_shuffle2(const float* t, void* y, long ldy, int B, int D, int H, int W, int Cout, int act16, void* stream);
int unetr_pixel_unshuffle2_bf16(const void* dy, long lddy, void* g_bf16, int B, int D, int H, int W, int Cout, int act16, void* stream);

/* ---- column sums: out[n] (+)= sum_m x[m*ld+n]  (bias / position-embedding gradients) ---------------- */
int unetr_colsum(const float* x, long ld, int M, int N, float* out, int accumulate,
                 float* ws, size_t ws_bytes, void* stream);

/* ---- LayerNorm (nn.LayerNorm(H), eps 1e-5, affine; MONAI TransformerBlock.norm1/2, ViT.norm) -------- */
int unetr_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                        void* y_bf16 /* optional bf16 copy of y for unetr_gemm_bf16, or NULL; y itself may be NULL when only the bf16 copy is wanted */,
                        float* mean, float* rstd, int M, int H, float eps, void* stream);
/* dgamma == dbeta == NULL: only dx is produced and ws keeps the per-row-block partial sums, laid out
 * [ceil(M/4)][2][H] (dgamma row, dbeta row), for a later unetr_colsum_grouped over all LayerNorms of the step. */
int unetr_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                        const float* rstd, float* dx, void* dx_bf16 /* optional bf16 copy of dx, or NULL */,
                        const float* dres /* optional, added to dx */, float* dgamma, float* dbeta,
                        int M, int H, float* ws, size_t ws_bytes, void* stream);

/* nn.LayerNorm backward (norm1 / norm2 of MONAI TransformerBlock) applied to dy = A . B, the data gradient of the Linear layer
 * that consumed the LayerNorm output (SABlock.qkv / MLPBlock.linear1): when that GEMM is cut into K slabs the LayerNorm kernel
 * sums the slabs itself (same order as the separate reduce launch: bit-identical), saving one launch per LayerNorm.  `d` must
 * describe a plain product (alpha 1, no bias / activation / residual / accumulate, ldc == N); C [M, N] is scratch; the remaining
 * arguments are those of unetr_layernorm_bwd (ln_ws = its ws). */
int unetr_gemm_bf16_ln_bwd(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C,
                           const float* x, const float* gamma, const float* mean, const float* rstd,
                           float* dx, void* dx_bf16, const float* dres, float* dgamma, float* dbeta,
                           float* ln_ws, size_t ln_ws_bytes, float* ws, size_t ws_bytes, void* stream);

/* The forward counterpart: C = A . B + bias + res (MLPBlock.linear2 + the block's residual add) and nn.LayerNorm of C (norm1 of
 * the NEXT TransformerBlock) -- when the GEMM is cut into K slabs the LayerNorm kernel sums them, applies bias and residual,
 * writes C and normalises the row (bit-identical to the three separate launches).  `d`: bias / res allowed, no activation / pre /
 * accumulate, alpha 1, ldc == N.  y (fp32) and / or y_bf16 receive the normalised rows. */
int unetr_gemm_bf16_ln_fwd(const unetr_gemm_bf16_desc* d, const void* A, const void* B, float* C,
                           const float* gamma, const float* beta, float eps, float* y, void* y_bf16, float* mean, float* rstd,
                           float* ws, size_t ws_bytes, void* stream);

/* ---- multi-head self-attention core (MONAI SABlock.forward between qkv and out_proj) ----------------
 * qkv: [B*L, 3*Hd] with feature = which*Hd + head*dh + j;  out: [B*L, Hd] ("b h l d -> b l (h d)");
 * lse: [B, heads, L] log-sum-exp of the scaled scores (saved for backward). */
int unetr_attention_fwd(const float* qkv, float* out, void* out_bf16 /* optional bf16 copy, or NULL */, float* lse,
                        int B, int L, int heads, int dh, float scale, int prec, void* stream);
int unetr_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                        float* dqkv, void* dqkv_bf16 /* optional bf16 copy, or NULL */, float* delta,
                        int B, int L, int heads, int dh, float scale, int prec, void* stream);

/* ---- 3x3x3 (pad 1) and 1x1x1 conv, stride 1, no bias (nn.Conv3d in MONAI UnetResBlock.conv1/2/3) ------
 * General-shape path: implicit GEMM through the MFMA GEMM family (im2col operand loaders, no im2col
 * buffer).  wpack comes from unetr_conv_pack_weight: mode 0 = forward [Cout][KV][Cin]; mode 1 = data
 * gradient (taps flipped, in/out swapped) [Cin][KV][Cout], to be used with Cin/Cout swapped. */
int unetr_conv_pack_weight(const float* w, float* wpack, int Cin, int Cout, int KS, int mode, void* stream);
int unetr_conv_gemm_fwd(const float* x, long ldx, const float* wpack, float* y, long ldy, int accumulate,
                        int B, int D, int H, int W, int Cin, int Cout, int KS, int prec,
                        float* ws, size_t ws_bytes, void* stream);
int unetr_conv_gemm_wgrad(const float* x, long ldx, const float* dy, long ldy, float* dw,
                          int B, int D, int H, int W, int Cin, int Cout, int KS, int prec,
                          float* ws, size_t ws_bytes, void* stream);

/* Dedicated 3x3x3 path (csrc/conv3.hip): LDS-staged halo windows, one HBM/L2 read per tile for all 27 taps.
 * wpack (element type follows prec) comes from unetr_conv3_pack_weight, sized by unetr_conv3_packed_bytes;
 * mode 0 = forward, mode 1 = data gradient (then call unetr_conv3_fwd with Cin/Cout swapped).  Cout % 16 == 0. */
size_t unetr_conv3_packed_bytes(int Cin, int Cout, int mode, int prec);
int unetr_conv3_pack_weight(const float* w, void* wpack, int Cin, int Cout, int mode, int prec, void* stream);
int unetr_conv3_fwd(const void* x, long ldx, const void* wpack, void* y, long ldy, int accumulate,
                    int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream);
/* dy3/dw3 (both or neither): also produce the weight gradient dw3[Cout,Cin] of the 1x1x1 conv that shares the
 * input x (MONAI UnetResBlock.conv3 next to conv1) from its own output gradient dy3, inside the same pass. */
/* Fused forward of the first half of MONAI's UnetResBlock (unetr.py:90-98, :135-174): y = conv3x3x3(x) together with
 * its InstanceNorm statistics stats[B,Cout,2] = (mean, rstd), and -- when w3pack != NULL -- y3 = conv1x1x1(x) (the
 * block's conv3, which reads the same input) with stats3.  wpack from unetr_conv3_pack_weight(mode 0), w3pack from
 * unetr_conv3_pack_1x1.  Returns "unsupported" for shapes that do not run on the persistent kernel (caller then uses
 * unetr_conv3_fwd + unetr_gemm + unetr_instnorm_stats). */
int unetr_conv3_fwd_fused(const void* x, long ldx, const void* wpack, void* y, long ldy, float* stats,
                          const void* w3pack, void* y3, long ldy3, float* stats3, float eps,
                          int B, int D, int H, int W, int Cin, int Cout, int prec,
                          int x_f32 /* bf16 mode: x is the fp32 image (<= 16 channels), not a bf16 feature map */,
                          float* ws, size_t ws_bytes, void* stream);
/* The same launch without the statistics finalize: the InstanceNorm partial sums stay as rows part / part3 [B][rows][2][Cout]
 * (caller-allocated for UNETR_CONV3_MAX_ROWS rows; *rows_out = the rows this launch wrote, one per workgroup) and are reduced by
 * their consumer in its own prologue (unetr_instnorm_apply_fin) -- no finalize launch. */
#define UNETR_CONV3_MAX_ROWS 1024
int unetr_conv3_fwd_parts(const void* x, long ldx, const void* wpack, void* y, long ldy, float* part,
                          const void* w3pack, void* y3, long ldy3, float* part3, int* rows_out,
                          int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32, void* stream);
/* Data gradient dx[.,Cin] = conv3x3x3^T(dy[.,Cout]; w) of a conv whose input was lrelu(InstanceNorm(xn)) (UnetResBlock.conv2,
 * unetr.py:90-98 / :135-174 via MONAI), with the backward statistics of that norm formed in the epilogue: part [B][rows][2][Cin]
 * = per-workgroup sums of g and g*n, g = dx * lrelu'(n), n = (xn - mean) * rstd from stats [B][Cin][2].  Replaces the separate
 * reduction pass of unetr_instnorm_bwd over (dx, xn); consumed by unetr_instnorm_bwd_apply_fin(nsp = 2). */
int unetr_conv3_dgrad_stats(const void* dy, long lddy, const void* wpack_dgrad, void* dx, long lddx,
                            const void* xn, long ldxn, const float* stats, float* part, int* rows_out,
                            int B, int D, int H, int W, int Cin, int Cout, int prec, void* stream);
size_t unetr_conv3_packed_1x1_bytes(int Cin, int Cout, int prec);
int unetr_conv3_pack_1x1(const float* w3 /* [Cout,Cin] */, void* w3pack, int Cin, int Cout, int prec, void* stream);
/* Data gradient of the residual block's input in one launch: dx = conv3x3x3^T(dc1; w1) + conv1x1x1^T(dc3; w3)
 * (autograd of UnetResBlock.conv1 + conv3, both fed by the block input).  wpack_dgrad from unetr_conv3_pack_weight(mode 1)
 * of conv1's weight; w3 = conv3's weight [Cout, Cin] as stored; dc1 / dc3 [B,D,H,W,Cout] channels-last with pitches. */
int unetr_conv3_dgrad_fused(const void* dc1, long ld1, const void* wpack_dgrad, const void* dc3, long ld3, const float* w3,
                            const void* w3pack_t /* optional: kind-3 pack of w3 (unetr_conv3_pack_grouped); NULL = packed here from w3 */,
                            void* dx, long lddx, int B, int D, int H, int W, int Cin, int Cout, int prec,
                            float* ws, size_t ws_bytes, void* stream);
/* All weight re-packs of a step in one launch (descriptors in the kernel arguments, <= 64 per launch).  kind 0 / 1: what
 * unetr_conv3_pack_weight(mode 0 / 1) writes; kind 2: unetr_conv3_pack_1x1; kind 3: the transposed 1x1x1 layout that
 * unetr_conv3_dgrad_fused builds internally.  `out` buffers sized by unetr_conv3_packed_bytes / _packed_1x1_bytes. */
typedef struct { const float* w; void* out; int Cin, Cout, kind; } unetr_pack_problem;
int unetr_conv3_pack_grouped(const unetr_pack_problem* probs, int n, int prec, void* stream);
/* (mean, rstd) from InstanceNorm partial sums part[B][nchunk][2][C] (sum, sum of squares) */
int unetr_instnorm_stats_finalize(const float* part, int nchunk, int B, long V, int C, float eps, float* stats, void* stream);
int unetr_conv3_wgrad(const void* x, long ldx, const void* dy, long ldy, float* dw,
                      const void* dy3, long ldy3, float* dw3,
                      int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32 /* as in unetr_conv3_fwd_fused */,
                      float* ws, size_t ws_bytes, void* stream);
/* The same weight gradients WITHOUT their reduce launch: the per-workgroup partial sums stay in `part` -- [rows][27 Cin Cout] followed,
 * when dy3 is given, by [rows][Cin Cout]; *rows_out = rows written (= unetr_conv3_wgrad_rows for an unlimited buffer; fewer when
 * part_bytes forces it) -- and are summed later by unetr_reduce_rows_grouped together with every other weight-gradient reduction
 * of the backward pass. */
long unetr_conv3_wgrad_rows(int B, int D, int H, int W, int Cin, int Cout, int prec, int x_f32, int has3);
int unetr_conv3_wgrad_parts(const void* x, long ldx, const void* dy, long ldy, const void* dy3, long ldy3,
                            float* part, size_t part_bytes, long* rows_out, int B, int D, int H, int W, int Cin, int Cout, int prec,
                            int x_f32, void* stream);
/* dst[i] = sum over rows g of part[g][i], i < n, for every problem, in one launch (fixed summation order) */
typedef struct { const float* part; float* dst; long n; int rows; } unetr_reduce_problem;
int unetr_reduce_rows_grouped(const unetr_reduce_problem* probs, int n, void* stream);
/* probe of the ds_read_b64_tr_b16 lane map used by the bf16 weight-gradient kernel (test hook) */
int unetr_debug_tr16(const void* in_u16_64x64, void* out_u16_64x4, void* stream);

/* ---- InstanceNorm3d(affine=False, eps 1e-5) + LeakyReLU(0.01) + residual add -------------------------
 * stats: [B, C, 2] = (mean, rstd). */
int unetr_instnorm_stats(const void* x, long ld, int B, long V, int C, float eps, float* stats,
                         float* ws, size_t ws_bytes, int act16, void* stream);
/* y = lrelu?(norm(x;sa) [+ norm(x2;sb)]) */
int unetr_instnorm_apply(const void* x, long ldx, const float* sa, const void* x2, long ldx2, const float* sb,
                         void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream);
/* unetr_instnorm_apply with the statistics finalize folded into the kernel's prologue: part_a / part_b = the partial rows
 * [B][rows][2][C] of the conv launches that produced x / x2; stats_a / stats_b [B][C][2] are WRITTEN (for the backward kernels).
 * "unsupported" for C > 128 or unaligned rows: the caller then runs unetr_instnorm_stats_finalize + unetr_instnorm_apply. */
int unetr_instnorm_apply_fin(const void* x, long ldx, const float* part_a, int rows_a, const void* x2, long ldx2,
                             const float* part_b, int rows_b, float* stats_a, float* stats_b, float eps,
                             void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream);
/* the apply half of unetr_instnorm_bwd reading partial sums part [B][nrows][nsp][C] (nsp 2: rows of unetr_conv3_dgrad_stats,
 * single form only; nsp 3: the reduction pass's rows) and finalizing them in its prologue */
int unetr_instnorm_bwd_apply_fin(const void* dy, long lddy, const void* x, long ldx, const float* sa,
                                 const void* x2, long ldx2, const float* sb, const float* part, int nrows, int nsp,
                                 void* dx, long lddx, void* dx2, long lddx2, int B, long V, int C, int lrelu, int act16, void* stream);
/* The block end of the residual block that reads the IMAGE (encoder1, unetr.py:90-98; <= 4 input channels): its 1x1x1 branch
 * c3 = conv1x1x1(img; w3 [C][Cin]) is formed per voxel from the fp32 image [B][V][Cin] instead of being stored and re-read, forward
 * (y = lrelu(norm(x) + norm(c3)); part_b = the partial rows of c3's statistics from unetr_conv3_fwd_parts, which then takes
 * y3 = NULL) and backward (dx of the 3x3x3 branch; the branch's weight gradient as partial rows dw3_part [*rows_out][C][Cin],
 * caller-allocated for UNETR_IN_IMG_MAX_ROWS rows, summed by unetr_reduce_rows_grouped; ws >= B * 1024 * 3 * C floats). */
#define UNETR_IN_IMG_MAX_ROWS 512
int unetr_instnorm_apply_fin_img(const void* x, long ldx, const float* part_a, int rows_a, const float* img, int Cin, const float* w3,
                                 const float* part_b, int rows_b, float* stats_a, float* stats_b, float eps,
                                 void* y, long ldy, int B, long V, int C, int lrelu, int act16, void* stream);
int unetr_instnorm_bwd_img(const void* dy, long lddy, const void* x, long ldx, const float* sa, const float* img, int Cin,
                           const float* w3, const float* sb, void* dx, long lddx, float* dw3_part, int* rows_out,
                           int B, long V, int C, int lrelu, float* ws, size_t ws_bytes, int act16, void* stream);
/* backward of y = lrelu?(norm(x) [+ norm(x2)]): writes dx (and dx2). */
int unetr_instnorm_bwd(const void* dy, long lddy, const void* x, long ldx, const float* sa,
                       const void* x2, long ldx2, const float* sb, void* dx, long lddx, void* dx2, long lddx2,
                       int B, long V, int C, int lrelu, float* ws, size_t ws_bytes, int act16, void* stream);

/* ---- layout moves --------------------------------------------------------------------------------- */
/* NCDHW [B,C,V] <-> channels-last [B,V,C] (pitch ld) */
int unetr_nchw_to_nhwc(const float* x /* fp32 NCDHW */, void* y /* feature map */, long ldy, int B, int C, long V, int act16, void* stream);
int unetr_nhwc_to_nchw(const void* x /* feature map */, long ldx, float* y /* fp32 NCDHW */, int B, int C, long V, int accumulate, int act16, void* stream);
/* einops "b c (h p1) (w p2) (d p3) -> b (h w d) (p1 p2 p3 c)" (MONAI PatchEmbeddingBlock, perceptron) */
/* patches (fp32) and / or patches_bf16 (the bf16 GEMM operand) may be NULL, not both */
int unetr_patch_gather(const float* x, float* patches, void* patches_bf16, int B, int C, int D, int H, int W, int P, void* stream);
/* y[i] += inc[i], i < n: AdamW's per-parameter step counters (device-resident so that the step can be a hipGraph) */
int unetr_counter_add(float* y, const float* inc, int n, void* stream);
/* y[r, 0:cols] (+)= a[r, 0:cols] for row-pitched matrices (skip -> concat buffer, gradient sums) */
int unetr_copy_rows(void* y, long ldy, const void* a, long lda, long rows, int cols, int accumulate, int act16, void* stream);

/* ---- 1x1x1 out conv with bias, NCDHW logits (MONAI UnetOutBlock; unetr.py:175,207) ------------------ */
int unetr_outconv_fwd(const void* x, long ldx, const float* w, const float* bias, float* logits,
                      int B, long V, int Cin, int Cout, int act16, void* stream);
int unetr_outconv_bwd(const float* dlogits, const void* x, long ldx, const float* w, void* dx, long lddx,
                      float* dw, float* dbias, int B, long V, int Cin, int Cout,
                      float* ws, size_t ws_bytes, int act16, void* stream);

/* decoder2's block end folded into the out conv (reference: UnetrUpBlock's UnetResBlock followed by UnetOutBlock,
 * /root/reference/unetr.py:165-175,206-207): out = lrelu(IN(c2) + IN(c3)) is formed per voxel and never stored.
 * unetr_outconv_in_fwd: logits [B][Cout][V] from c2 / c3 [B][V][C] and their InstanceNorm partial rows; writes stats_a / stats_b.
 * unetr_outconv_in_bwd: dout = W^T dlogits (stored, pitch lddo), the InstanceNorm backward partial rows in_part [B][rows][3][C]
 * (rows = unetr_outconv_in_bwd_rows; feed unetr_instnorm_bwd_apply_fin with nsp = 3), dw [Cout][C], dbias [Cout].
 * UNETR_ERR_UNSUPPORTED (Cout > 4, C > 16, ...) = run the unfused sequence. */
int unetr_outconv_in_fwd(const void* c2, long ld2, const float* part_a, int rows_a, const void* c3, long ld3, const float* part_b,
                         int rows_b, float* stats_a, float* stats_b, float eps, const float* w, const float* bias, float* logits,
                         int B, long V, int C, int Cout, int act16, void* stream);
long unetr_outconv_in_bwd_rows(int B, long V, int C, int act16);
int unetr_outconv_in_bwd(const float* dlogits, const void* c2, long ld2, const float* sa, const void* c3, long ld3, const float* sb,
                         const float* w, void* dout, long lddo, float* in_part, float* dw, float* dbias,
                         int B, long V, int C, int Cout, float* ws, size_t ws_bytes, int act16, void* stream);

/* ---- DiceCELoss (unetr_segmentation_3d.py:404 and :477-482) -------------------------------------------
 * sigmoid_multilabel = 0: DiceCELoss(to_onehot_y=True, softmax=True); label [B,V] float-valued class ids.
 * sigmoid_multilabel = 1: DiceCELoss(to_onehot_y=False, sigmoid=True); label [B,C,V] float multi-label mask; the
 *   CE term is softmax cross entropy against argmax_c(label) (MONAI 0.6.0 DiceCELoss.ce with equal channel counts).
 * logits [B,C,V] NCDHW.  out[0]=loss, out[1]=dice term, out[2]=ce term.
 * coef: [B*C*2] per-(b,c) Dice gradient coefficients kept for backward. */
int unetr_dicece_fwd(const float* logits, const float* label, int B, int C, long V, int sigmoid_multilabel,
                     float smooth_nr, float smooth_dr, float* out, float* coef, float* ws, size_t ws_bytes, void* stream);
int unetr_dicece_bwd(const float* logits, const float* label, const float* coef, const float* dloss,
                     float* dlogits, int B, int C, long V, int sigmoid_multilabel, void* stream);

/* ---- validation side (unetr_segmentation_3d.py:103-132) -------------------------------------------------
 * unetr_sw_accumulate / unetr_sw_finalize: the blending step of monai.inferers.sliding_window_inference (MONAI 0.6.0,
 * called at :110): out[c, z0+z, y0+y, x0+x] += w * seg[c,z,y,x], count[...] += w for ONE window of ONE volume
 * (out / count point at that volume: [C,D,H,W] and [D,H,W]; importance = NULL means w = 1, mode="constant"), then
 * out[b,c,v] /= count[b,v].  One launch per window, in MONAI's window order.
 * unetr_dice_counts: the sums behind monai.metrics.DiceMetric (:485-486): counts[b][c] = (sum pred*y, sum pred, sum y).
 * from_logits = 1: pred = one_hot(argmax_c logits[B,C,V]) and y = class ids [B,V] (AsDiscrete(argmax, to_onehot) at
 * :405-406 fused in); from_logits = 0: pred and y are [B,C,V] (already discrete) tensors. */
int unetr_sw_accumulate(const float* seg, const float* importance, float* out, float* count, int C,
                        int rz, int ry, int rx, int D, int H, int W, int z0, int y0, int x0, void* stream);
int unetr_sw_finalize(float* out, const float* count, int B, int C, long V, void* stream);
int unetr_dice_counts(const float* pred, const float* y, int B, int C, long V, int from_logits, double* counts,
                      float* ws, size_t ws_bytes, void* stream);

/* ---- ranking pre-training losses (unetr_ranking_pretraining_3d.py:59-133 triplet construction, :202-217 BTLoss,
 * :219-236 ContrastiveLoss), fused: feat is the NCDHW feature map [4, C, S1, S2, S3] (2 volumes x 2 transforms: enc4 in
 * the "feat" stage, logits in the "recon" stage), slice_dim in {2,3,4} as in the reference, init_idx the random slice
 * offset the reference draws with np.random.choice (injected for determinism).  kind 0 = Bradley-Terry, 1 = contrastive.
 * W [C,16,16] keeps dL/d<v_i,v_j> for backward; dfeat must be zero-filled by the caller (only the 16 slices are written). */
size_t unetr_ranking_workspace_floats(int C, int S1, int S2, int S3, int slice_dim);
int unetr_ranking_loss_fwd(const float* feat, int C, int S1, int S2, int S3, int slice_dim, int init_idx,
                           float temperature, int kind, float* loss, float* W, float* ws, size_t ws_floats, void* stream);
int unetr_ranking_loss_bwd(const float* feat, int C, int S1, int S2, int S3, int slice_dim, int init_idx,
                           const float* W, const float* dloss, float* dfeat, void* stream);

/* ---- fused AdamW over one flat fp32 buffer (torch.optim.AdamW semantics; unetr_segmentation_3d.py:522) */
int unetr_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                float eps, float weight_decay, const float* step_dev,
                void* shadow_bf16 /* optional: bf16 copy of the updated p (GEMM weight shadow), or NULL */, void* stream);

/* data-parallel form of the same update (replaces the DistributedDataParallel gradient averaging that the reference's
 * single-GPU script does not need, BASELINE.json configs[2]): g is the all-reduced SUM in the communication buffer
 * (fp32, or bf16 when g_is_bf16), gscale = 1 / world size is applied on the fly. */
int unetr_adamw_reduced(float* p, const void* g, int g_is_bf16, float gscale, float* m, float* v, long n, float lr,
                        float beta1, float beta2, float eps, float weight_decay, const float* step_dev,
                        void* shadow_bf16, void* shadow_x3 /* optional: bf16x3 word shadow of the updated p (unetr_split_words), or NULL */,
                        void* stream);

/* ---- the single-GPU step's optimizer fused into the producer of the gradients (train_step.TrainStep(fuse_update=True)) ------
 * The four arenas (parameters, gradients, both moments; fp32, `total` elements, identical layout) and the optional bf16 shadow
 * arena of the parameters; steps = per-parameter step counts (float, already advanced for this step). */
typedef struct {
    float* param; const float* grad; float* m; float* v; void* shadow_bf16; const float* steps; long total;
    float lr, beta1, beta2, eps, weight_decay;
    void* shadow_x3;                  /* optional arena of bf16x3 words (unetr_split_words layout), written next to shadow_bf16; or NULL */
} unetr_adamw_arena;
/* grouped ViT weight gradients (unetr_gemm_bf16_grouped_wgrad) whose epilogue APPLIES AdamW instead of storing dW: every
 * probs[i].dw must address a slice of a->grad (it is not written); the parameter / moment / shadow slices at the same arena
 * offset are updated with g = dW, step count a->steps[step_index[i]].  Same bits as the gradient store followed by unetr_adamw
 * (torch.nn.Linear backward + torch.optim.AdamW.step, unetr_segmentation_3d.py:224-225), 8 bytes per weight less HBM traffic. */
int unetr_gemm_bf16_grouped_wgrad_adamw(const unetr_grouped_problem* probs, int n, const unetr_adamw_arena* a,
                                        const int* step_index, void* stream);
/* the data-parallel counterpart (bf16 gradient communication, BASELINE.json configs[2]): the same grouped weight gradients stored
 * as bf16 at the arena offset of probs[i].dw (a slice of grad_arena, NOT written) in out_bf16_arena -- the communication buffer --
 * i.e. what unetr_cast_bf16 makes of the fp32 gradient, without the fp32 store, the cast pass and its re-read; and the cast of
 * every other gradient range in one launch (table in DEVICE memory, 3 longs per range: element offsets lo, hi -- multiples of
 * 8 --, first block of the range in blocks of 8192 elements, running sum; n_blocks = total). */
int unetr_gemm_bf16_grouped_wgrad_bf16out(const unetr_grouped_problem* probs, int n, const float* grad_arena, void* out_bf16_arena,
                                          long total, void* stream);
int unetr_cast_bf16_ranges(const float* src_arena, void* dst_arena, const long* table_dev, int n_ranges, long n_blocks, void* stream);
/* AdamW over n_ranges arena ranges in ONE launch (the parameters the fused launch above did not cover: biases, LayerNorm,
 * conv weights ...).  table (DEVICE memory, 4 longs per range): element offset lo, element offset hi (multiples of 4 as the
 * arena packs them), index into a->steps, first block of the range (blocks of 4096 elements, running sum); n_blocks = total. */
int unetr_adamw_ranges(const unetr_adamw_arena* a, const long* table_dev, int n_ranges, long n_blocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
