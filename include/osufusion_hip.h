/* osufusion_hip.h -- C ABI of libosuf_hip.so: the MI355X (gfx950) kernels behind the OsuFusion denoiser hot path.
 *
 * The reference (fauzanardh/OsuFusion) has no FFI/plugin layer: its boundary is the Python nn.Module API of
 * osu_fusion.modules.{unet,residual,attention} / osu_fusion.models.diffusion.  Every
 * entry point below replaces the vendor kernel(s) that one reference call site reaches through torch; the
 * citation after "replaces:" is that call site (paths relative to /root/reference/osu_fusion).
 *
 * Conventions
 *   - plain device pointers + sizes + hipStream_t; no allocation, no host sync, graph-capturable.  Nothing is kept between
 *     calls except the one-time hipFuncSetAttribute registration of kernels that need > 64 KiB of LDS.  Kernel variants are
 *     chosen per call (`variant` arguments); the GEMM launchers additionally honour the documented A/B environment switches
 *     OSUF_GEMM_* / OSUF_TN_* (read on every call, never cached) -- measurement aids, not configuration;
 *   - return 0 on success, <0 for an argument error (-1 invalid, -2 unsupported), >0 = hipError_t of the launch;
 *   - activations are channels-last rows [B*L][C] ("rows"), C contiguous, row stride `ld*` in ELEMENTS;
 *   - dtype: 0 = fp32 storage (exact-f32 MFMA), 1 = bf16 storage (bf16 MFMA, fp32 accumulate);
 *   - all row strides / channel counts must be multiples of 8 elements, pointers 16-byte aligned.
 */
#ifndef OSUFUSION_HIP_H
#define OSUFUSION_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSUF_DT_F32 0
#define OSUF_DT_BF16 1
#define OSUF_DT_F32X3 2   /* osuf_gemm_nt / osuf_gemm_tn only: fp32 storage, every product as three bf16 MFMAs on split operands
                             (a = bf16(a) + bf16(a - bf16(a)): inputs kept to ~17 bits) -- 3/16 of the cost of the exact-f32 MFMA */
/* `variant` of the attention-backward entry points: AUTO picks by shape, PLAIN / PIPE force the plain or the software-pipelined kernel */
#define OSUF_ATTN_AUTO 0
#define OSUF_ATTN_PLAIN 1
#define OSUF_ATTN_PIPE 2
/* `dq_mode` of osuf_mqa_bwd_fused: fp32 atomics (default), or per-key-block slabs summed in a fixed order */
#define OSUF_DQ_ATOMIC 0          /* fp32 atomics; the sweep (256 or 512 keys per workgroup) is picked by shape */
#define OSUF_DQ_SLABS 1
#define OSUF_DQ_ATOMIC_256 2      /* force the 8-wave, 256-key sweep */
#define OSUF_DQ_ATOMIC_512 3      /* force the 4-wave, 512-key sweep (whole 512-key blocks: N % 512 == 0, else OSUF_EUNSUPPORTED) */
#define OSUF_DQ_ATOMIC_512A 5     /* the 512-key sweep with its hand-placed loop (tools/gen_attn_bwd512.py): dK / dV bit-identical to
                                     OSUF_DQ_ATOMIC_512; whole 512-key blocks and an even number of (head, query block) pairs per part */
#define OSUF_DQ_PREZEROED 0x100   /* flag, OR-ed into an atomic dq_mode: the dQ accumulator at the head of `workspace` was zero-filled by
                                     osuf_mqa_fwd_zdq of the same layer (the entry point then issues no memset) */
#define OSUF_DQ_TIMING_512 4      /* DEBUG: the 512-key sweep WITHOUT its atomics (prices the loop; dq comes back zero) -- refused with
                                     OSUF_EUNSUPPORTED unless the process environment holds OSUF_ALLOW_TIMING_BUILDS=1 */

int osuf_version(void);

/* ---- conv1d / linear as tap-GEMMs (gemm.hip) -------------------------------------------------------------
 * C[m][n] = act( sum_t sum_k A[rowmap(m,t)][k] * W[t][n][k] + bias[n] ) * silu'(U[m][n]) + R[m][n] * rscale[b][n]
 * rowmap modes: 0 plain (stride/pad), 1 reflect-right (Downsample), 2 nearest-x2 input (Upsample), 3 dgrad of 1.
 * C2 (optional) receives the pre-activation; stats (optional, double [B][2]) accumulates per-sample sum / sum^2
 * of the stored output for the following GroupNorm(1, C).
 * replaces: nn.Conv1d in Block.proj (modules/residual.py:70,76), res_conv (residual.py:115,137),
 *           Downsample/Upsample/Parallel convs (modules/unet.py:66-69,81-87,95-101,219-236), CrossEmbedLayer
 *           (unet.py:42-58), final_conv (unet.py:354,513); nn.Linear to_q/to_kv/to_out (unet.py:118-123,129-141),
 *           FeedForward (unet.py:149-156), time/cond/FiLM MLPs (unet.py:356-366, residual.py:104-111,126-129),
 *           GlobalContext 1x1 MLP (residual.py:22-27); and autograd's input-gradient of each of them. */
int osuf_gemm_nt(int dtype, const void* A, long lda, const void* W, long ldw, long tapstride,
                 void* C, long ldc, void* C2, long ldc2, const void* R, long ldr, const void* U, long ldu,
                 const float* bias, const float* rscale, double* stats,
                 int M, int N, int K, int taps, int Lin, int Lout, int stride, int pad, int mode, int act,
                 hipStream_t stream);

/* C = A W^T (one tap, no bias, no activation) and, from the same epilogue, the row constants of the flash backward:
 *   delta[b][h][l] = sum_{d < 64} bf16(C[b*L + l][h*64 + d]) * O[b*L + l][h*64 + d]            (N = heads * 64, M % L == 0)
 * replaces: autograd's input-gradient of Attention.to_out (modules/unet.py:123,141) + the sum(dO * O) pass of SDPA's backward
 * (attention.py:94-99) -- i.e. osuf_gemm_nt followed by osuf_attn_delta, without re-reading dO and O. */
int osuf_gemm_nt_rowdot(int dtype, const void* A, long lda, const void* W, long ldw, void* C, long ldc, const void* O, long ldo,
                        float* delta, int M, int N, int K, int L, int heads, hipStream_t stream);

/* dW (+)= sum_m dY[m][n1] * X[rowmap(m,t)][n2].  out_layout 0: dW[t][n1][n2] with row stride ldw and tap stride tapstride;
 * out_layout 1: dense dW[n1][n2][t], i.e. torch's (Cout, Cin, k) Conv1d weight layout, so the gradient can be accumulated
 * straight into the parameter's .grad.  accumulate 1: add into dW; 0: overwrite (no zero-init needed by the caller).
 * The reduction over m is split across workgroups.  With a caller-provided fp32 `workspace` of at least
 * osuf_gemm_tn_workspace_bytes(...) bytes the partial tiles are written with plain stores and summed by a second kernel
 * (deterministic); without it (NULL / too small / shape not eligible) partial sums are added with fp32 atomics.
 * replaces: autograd's weight-gradient of every Conv1d / Linear listed above. */
int osuf_gemm_tn(int dtype, const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, long tapstride,
                 int M, int N1, int N2, int taps, int Lin, int Lout, int stride, int pad, int mode,
                 int splits, int out_layout, int accumulate, float* workspace, long workspace_bytes, hipStream_t stream);
long osuf_gemm_tn_workspace_bytes(int dtype, int M, int N1, int N2, int taps);
/* osuf_gemm_tn + the bias gradient of the same layer from the same pass over dY: dbias[n1] += sum_m dY[m][n1] (fp32, N1 entries).  The
 * bf16 256x256 and merged-taps kernels sum the dY fragments they hold anyway; every other path (fp32 modes, small shapes) runs osuf_colsum
 * on dY itself (then N1 % 8 == 0 is required).
 * replaces: autograd's weight- and bias-gradient of one Conv1d / Linear (residual.py:70,115, unet.py:118-123,149-156). */
int osuf_gemm_tn_bias(int dtype, const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, long tapstride,
                      int M, int N1, int N2, int taps, int Lin, int Lout, int stride, int pad, int mode,
                      int splits, int out_layout, int accumulate, float* workspace, long workspace_bytes, float* dbias, hipStream_t stream);

/* out[n] += sum_m Y[m][n]     replaces: autograd's bias-gradient of Conv1d / Linear. */
int osuf_colsum(int dtype, const void* Y, long ldy, int M, int N, float* out, hipStream_t stream);

/* ---- GroupNorm(1,C) + FiLM + SiLU (norm.hip)    replaces: Block.forward_body (modules/residual.py:75-84) ---- */
int osuf_gn_finalize(const double* stats, float* mean_rstd, int B, long count, hipStream_t stream);
/* The same statistics without atomics, from the stored conv output (one extra read of y): per-chunk sums in `partial`
 * (osuf_gn_stats_workspace_bytes(M, C, L) bytes), added in a fixed order -- identical inputs give identical bits.  Used by the
 * sampling loop (models/diffusion.py:59-77, inference_gradio.py:128) so that two calls of sample() return the same beatmap;
 * the training step keeps the statistics fused into the GEMM epilogue (osuf_gemm_nt `stats`, fp32/fp64 atomics). */
int osuf_gn_stats(int dtype, const void* y, long ldy, double* partial, float* mean_rstd, int M, int C, int L, hipStream_t stream);
long osuf_gn_stats_workspace_bytes(int M, int C, int L);
int osuf_gn_apply_fwd(int dtype, const void* y, long ldy, void* h, long ldh, const float* mean_rstd, const float* gamma,
                      const float* beta, const float* scale_shift, int M, int C, int L, hipStream_t stream);
/* osuf_gn_finalize + osuf_gn_apply_fwd in one launch: stats[b] = raw (sum, sum of squares) over count = L*C elements; the kernel
 * finalises them itself and also writes (mean, rstd) to mr_out[B][2] (needed again by osuf_gn_bwd). */
int osuf_gn_apply_fwd_stats(int dtype, const void* y, long ldy, void* h, long ldh, const double* stats, long count, float* mr_out,
                            const float* gamma, const float* beta, const float* ss, int M, int C, int L, hipStream_t stream);
/* The bit-reproducible statistics without their second launch: osuf_gn_stats_parts is stage 1 of osuf_gn_stats alone (per-chunk sums into
 * partial[B][osuf_gn_stats_workspace_bytes / 16][2]); osuf_gn_apply_fwd_parts adds a sample's chunks in a fixed order inside the apply kernel
 * (every wave the same order: same bits in every workgroup) and also writes (mean, rstd) to mr_out[B][2].  The sampling loop's path since round 5. */
int osuf_gn_stats_parts(int dtype, const void* y, long ldy, double* partial, int M, int C, int L, hipStream_t stream);
int osuf_gn_apply_fwd_parts(int dtype, const void* y, long ldy, void* h, long ldh, const double* partial, float* mr_out,
                            const float* gamma, const float* beta, const float* ss, int M, int C, int L, hipStream_t stream);
/* T1234 [B][4][C] fp32 zeroed scratch; S [B][2] scratch; dss [B][2C] (may be NULL); dgamma/dbeta accumulated into; dbias (may be
 * NULL): gradient of the bias of the conv that produced y (= column sums of dy, residual.py:77 `self.proj`), accumulated into;
 * dyy (may be NULL, needs dbias): column sums of dy*y -- the DoRA magnitude gradient's numerator (lora_layers.py:86-90).
 * dgamma == dbeta == NULL: the identity "norm" of Block(norm=False) (residual.py:71): the caller passes mean 0 / rstd 1 / gamma 1 /
 * beta 0 and the kernels drop every statistics term (dy = (1 + scale) * dh * silu'(u)). */
int osuf_gn_bwd(int dtype, const void* dh, long lddh, const void* y, long ldy, void* dy, long lddy, const float* mean_rstd,
                const float* gamma, const float* beta, const float* scale_shift, float* T1234, float* S, float* dss,
                float* dgamma, float* dbeta, float* dbias, float* dyy, int M, int C, int L, hipStream_t stream);

/* ---- LayerNorm    replaces: Attention.norm (modules/unet.py:117,127) ------------------------------------- */
int osuf_ln_fwd(int dtype, const void* x, long ldx, void* out, long ldo, float* mean_rstd, const float* gamma, const float* beta,
                int M, int C, hipStream_t stream);
int osuf_ln_bwd(int dtype, const void* dy, long lddy, const void* x, long ldx, void* dx, long lddx, const float* mean_rstd,
                const float* gamma, float* dgamma, float* dbeta, int M, int C, hipStream_t stream);

/* ---- GlobalContext gate    replaces: GlobalContext.forward_body + `h * se(h)` + residual add
 *      (modules/residual.py:29-32,135-137) ------------------------------------------------------------------ */
int osuf_rowdot(int dtype, const void* h, long ldh, const float* w, long w_stride, const float* bias, float* out,
                int M, int C, int L, hipStream_t stream);
/* GlobalContext pooling (residual.py:29-31) in ONE pass over h: p[B*L] = softmax_n(h . wk + bk) and pooled[B][C] = sum_n p[n] h[b,n,:] (fp32), by a
 * running softmax per workgroup + a second stage that adds the workgroups' partials in order (`part`: osuf_gca_pool_workspace_bytes bytes of
 * scratch).  Replaces osuf_rowdot + osuf_softmax_rows + osuf_wcolsum (two reads of h) on that path; no atomics: bit-reproducible. */
long osuf_gca_pool_workspace_bytes(int M, int C, int L);
int osuf_gca_pool(int dtype, const void* h, long ldh, const float* wk, const float* bk, float* part, float* p, float* pooled,
                  int M, int C, int L, hipStream_t stream);
int osuf_softmax_rows(float* p, int B, int L, hipStream_t stream);
/* osuf_wcolsum: out[b][c] (+)= sum_l w[b][l] * a[b][l][c] (* bmul[b][l][c]).  partial == NULL: row chunks meet by fp32 atomics in a
 * zero-initialised `out`; partial = B * ceil(L / 64) * C floats: chunk sums are stored and added in chunk order (`out` overwritten,
 * bit-reproducible -- the sampling loop's path). */
int osuf_wcolsum(int dtype, const void* a, long lda, const void* bmul, long ldb, const float* w, float* out,
                 int B, int C, int L, float* partial, hipStream_t stream);
int osuf_gate_residual(int dtype, const void* h, long ldh, const float* gate, const void* res, long ldr, void* out, long ldo,
                       int M, int C, int L, hipStream_t stream);
int osuf_gca_bwd_apply(int dtype, const void* dout, long lddo, const void* h, long ldh, void* dh, long lddh, const float* p,
                       const float* gate, const float* dpooled, const float* sdot, const float* wk, float* dlogit,
                       int M, int C, int L, float* dwk, float* dbk, float* workspace, long workspace_bytes, hipStream_t stream);
long osuf_gca_bwd_apply_workspace_bytes(int M, int C);
/* (dwk [C] / dbk [1], optional, accumulated into: sum_m dlogit[m] * h[m][:] and sum_m dlogit[m], the gradients of to_k's weight and
 *  bias (residual.py:20,29), from the h rows osuf_gca_bwd_apply already holds.  With a workspace of
 *  osuf_gca_bwd_apply_workspace_bytes the per-workgroup partials go through a slab and a small second kernel; without it
 *  (NULL / too small) thousands of workgroups add to the same C addresses with atomics, which costs 25-40 us per launch.) */

/* ---- attention (attn.hip) -----------------------------------------------------------------------------------
 * replaces: RotaryPositionEmbedding.forward + apply_rotary_pos_emb (modules/attention.py:52-58, utils.py:25-32) and
 *           the q/k/v -> bf16 casts of Attend.forward (attention.py:87-92) */
int osuf_rope_cast(int dtype, const void* in, long ld_in, void* out_bf16, long ld_out, const float* cos_tab, const float* sin_tab,
                   int M, int N, int n_rot_heads, int n_heads_total, int head_dim, hipStream_t stream);
/* as osuf_rope_cast (head_dim 64), and the first n_q_heads heads -- the queries -- are multiplied by q_mul before their bf16 rounding: with
 * q_mul = scale * log2(e) the softmax scale of attention.py:94-99 rides the one rounding the reference's bf16 cast (attention.py:87-92) performs
 * anyway, and osuf_mqa_fwd_qs / osuf_mqa_bwd_fused_qs take Qs K^T straight as log2-domain scores. */
int osuf_rope_cast_qs(int dtype, const void* in, long ld_in, void* out_bf16, long ld_out, const float* cos_tab, const float* sin_tab,
                      int M, int N, int n_rot_heads, int n_heads_total, int head_dim, float q_mul, int n_q_heads, hipStream_t stream);
int osuf_rope_bwd(int dtype, const float* in, long ld_in, void* out, long ld_out, const float* cos_tab, const float* sin_tab,
                  int M, int N, int n_rot_heads, int n_heads_total, int head_dim, hipStream_t stream);
/* replaces: the GQA repeat (modules/unet.py:135) + F.scaled_dot_product_attention (attention.py:94-99) and its backward.
 * q/k/v/dout are bf16; o is written bf16-rounded in o_dtype; lse2 = log2-domain logsumexp [B][H][N]. */
int osuf_mqa_fwd(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                 float* lse2, int B, int H, int N, int head_dim, float scale, hipStream_t stream);
/* osuf_mqa_fwd for queries pre-scaled by scale * log2(e) (osuf_rope_cast_qs); same outputs, same lse2 convention; head_dim 64.
 * replaces: F.scaled_dot_product_attention at attention.py:94-99 (as osuf_mqa_fwd). */
int osuf_mqa_fwd_qs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                    float* lse2, int B, int H, int N, int head_dim, float scale, hipStream_t stream);
/* osuf_mqa_fwd (qs = 0) / osuf_mqa_fwd_qs (qs = 1) that also zero-fills zero_dq[B*N][H*64] (fp32): the dQ accumulator at the head of the
 * workspace of this layer's osuf_mqa_bwd_fused* call, which then takes dq_mode | OSUF_DQ_PREZEROED.  The forward loop is bound by the vector
 * pipe with HBM idle, so the fill costs nothing there; in front of the backward sweep it is 66 us per N = 4096 layer.  head_dim 64 only. */
int osuf_mqa_fwd_zdq(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                     float* lse2, int B, int H, int N, int head_dim, float scale, int qs, float* zero_dq, hipStream_t stream);
/* The forward on UN-rotated queries: q_raw = the q block of the q|kv projection as the GEMM left it (bf16).  The kernel rotates each wave's
 * 32 x 64 query tile (rope_cos / rope_sin: [N][32] fp32), multiplies it by q_mul = scale * log2 e and rounds it to bf16 once -- osuf_rope_cast_qs'
 * arithmetic -- and stores it to q_out ([B*N][ldqo], may be NULL: inference) for osuf_mqa_bwd_fused_qs.  k / v: already rotated / cast
 * (osuf_rope_cast on those two head blocks alone: 128 of the (H + 2) * 64 columns).  zero_dq: as osuf_mqa_fwd_zdq, may be NULL.  head_dim 64.
 * Replaces the q part of `apply_rotary_pos_emb` + the cast in front of SDPA (attention.py:52-58,87-92) -- a full read + write of the q|kv rows. */
int osuf_mqa_fwd_rope(const void* q_raw, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                      float* lse2, int B, int H, int N, int head_dim, float scale, const float* rope_cos, const float* rope_sin,
                      float q_mul, void* q_out, long ldqo, float* zero_dq, hipStream_t stream);
/* Attend(q, k, v, attn_mask) (attention.py:77-99): the reference casts the mask to bf16 and passes it to SDPA as an additive bias of
 * the scaled scores (so a bool mask adds 1.0 / 0.0 -- kept).  mask: bf16, element strides over (batch, head, query, key), 0 for a
 * broadcast dimension.  Inference only (no backward entry point); all head dims go through the generic kernel.
 * replaces: F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask) at attention.py:94-99. */
int osuf_mqa_fwd_masked(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo, int o_dtype,
                        float* lse2, const void* mask, long mask_b, long mask_h, long mask_q, long mask_k, int B, int H, int N,
                        int head_dim, float scale, hipStream_t stream);
int osuf_attn_delta(const void* dout, long lddo, const void* o, long ldo, int o_dtype, float* delta, int B, int H, int N,
                    int head_dim, hipStream_t stream);
/* dq / dk / dv are written in out_dtype (OSUF_DT_F32 or OSUF_DT_BF16).  rope_cos / rope_sin ([N][32] fp32, or both NULL): q and k
 * were rotated by apply_rotary_pos_emb (attention.py:52-58) before the attention; with the tables given, dq and dk come out as
 * gradients of the UN-rotated projections (the rotation's transpose rides the epilogue; dv is never rotated). */
int osuf_mqa_bwd_dq(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                    const float* lse2, const float* delta, void* dq, long lddq, int B, int H, int N, int head_dim, float scale,
                    int out_dtype, const float* rope_cos, const float* rope_sin, int variant, hipStream_t stream);
int osuf_mqa_bwd_dkv(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                     const float* lse2, const float* delta, void* dk, void* dv, long lddk, int B, int H, int N, int head_dim,
                     float scale, int out_dtype, const float* rope_cos, const float* rope_sin, float* workspace, long workspace_bytes,
                     int qsplit, int variant, hipStream_t stream);
/* Short sequences give the dK/dV kernel few 256-key workgroups (B=32, N=512: 64 on 256 CUs): with a 16-byte-aligned fp32 workspace
 * of this many bytes (0 = the shape does not need it) the query range is cut into 2 or 4 parts whose partial sums a finishing pass
 * adds in a fixed order (then scale, RoPE transpose, cast).  workspace NULL / too small: the unsplit kernel runs.
 * qsplit: 0 = the default split of the shape; 1..16 = force that many parts (the same value goes to osuf_mqa_bwd_dkv). */
long osuf_mqa_bwd_dkv_workspace_bytes(int B, int N, int qsplit);

/* The whole attention backward in ONE key-stationary sweep (S, dP and the exponentials are computed once per (query, key) pair
 * instead of once in each of the two kernels above): dK / dV stay in registers; every 256-key workgroup leaves its partial dQ in
 * `workspace` -- dq_mode OSUF_DQ_ATOMIC: fp32 atomics into one [B*N][H*64] buffer (sum order not fixed: the last fp32 bits vary
 * between runs, as with autograd's own SDPA backward); OSUF_DQ_SLABS: one slab per key block in the output's element type, plain
 * stores, added in key-block order by the finishing pass (bit-reproducible, slower) -- and a finishing pass scales / un-rotates /
 * casts it into dq.  workspace:
 * osuf_mqa_bwd_fused_workspace_bytes(B, H, N, out_dtype, qsplit, dq_mode) bytes, 16-byte aligned.  dq / dk / dv, rope tables,
 * qsplit: as above.
 * replaces: the same call sites as osuf_mqa_bwd_dq + osuf_mqa_bwd_dkv (backward of attention.py:94-99 under unet.py:125-141). */
int osuf_mqa_bwd_fused(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                       const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                       int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                       float* workspace, long workspace_bytes, int qsplit, int dq_mode, hipStream_t stream);
/* osuf_mqa_bwd_fused for queries pre-scaled by c = scale * log2(e) (q from osuf_rope_cast_qs, lse2 from osuf_mqa_fwd_qs): p = exp2(Qs K^T - lse2)
 * without a multiply (the generated 512-key loop drops 64 vector instructions per (head, 32-query block) pair); dq is still the gradient of the
 * UN-scaled rotated q (scale dS K), dk = (scale / c) dS^T Qs.  Same workspace, same arguments.
 * replaces: the backward of attention.py:94-99 under unet.py:125-141 (as osuf_mqa_bwd_fused). */
int osuf_mqa_bwd_fused_qs(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* dout, long lddo,
                          const float* lse2, const float* delta, void* dq, long lddq, void* dk, void* dv, long lddk, int B, int H,
                          int N, int head_dim, float scale, int out_dtype, const float* rope_cos, const float* rope_sin,
                          float* workspace, long workspace_bytes, int qsplit, int dq_mode, hipStream_t stream);
long osuf_mqa_bwd_fused_workspace_bytes(int B, int H, int N, int out_dtype, int qsplit, int dq_mode);

/* ---- layout / scheduler / optimizer (elementwise.hip) ----------------------------------------------------------
 * replaces: the (B,C,L) <-> (B,L,C) rearranges (modules/unet.py:180,183) at the model boundary, torch.cat (unet.py:500,
 *           507,510), DDIMScheduler.add_noise / .step (models/diffusion.py:96,75; diffusers 0.29.2), the CFG combine
 *           (unet.py:465), F.mse_loss + mask (diffusion.py:101-111), get_total_norm / clip_grad_norm_ / AdamW.step
 *           (trainer.py:32-39,302-307). */
int osuf_ncl_to_rows(int dtype, const float* in, void* out, long ld, int width, int B, int C, int L, int KT, hipStream_t stream);
int osuf_rows_to_ncl(int dtype, const void* in, long ld, float* out, int B, int C, int L, hipStream_t stream);
int osuf_copy2d(int src_dtype, const void* src, long lds, int dst_dtype, void* dst, long ldd, int M, int cols, hipStream_t stream);
int osuf_add2d(int dtype, const void* a, long lda, const void* b, long ldb, void* dst, long ldd, int M, int cols, hipStream_t stream);
int osuf_axpby_rows(const float* x, const float* y, const float* ca, const float* cb, float* out, int B, long per_sample, hipStream_t stream);
int osuf_ddim_step(const float* x, const float* cond, const float* nullp, float cond_scale, const float* coef, float* out,
                   int B, long per_sample, hipStream_t stream);
int osuf_mse(const float* pred, const float* target, const int* orig_len, float* grad, double* loss_sum, int B, int Dch, int L,
             hipStream_t stream);
int osuf_sqnorm(const float* g, long n, double* out, hipStream_t stream);
int osuf_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float wd,
               int step, const float* gscale, hipStream_t stream);
int osuf_clip_coef(const double* sumsq, float max_norm, float base, float* coef, float* total_norm, hipStream_t stream);
int osuf_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream);
/* GEMM operand layouts of one conv / linear weight, both from the fp32 master (O, I, k) in one pass -- what the reference gets
 * for free from cuDNN's filter transforms inside F.conv1d / its backward (modules/unet.py:61-101, residual.py:75-137):
 *   F[t][o][i] = w[o][i][t]                                (forward operand, [taps][N=O][K=I]; may be NULL)
 *   D[t'][i][o]: dkind 0 flipped taps (same conv), 1 Downsample dgrad (4 taps, tap 3 = reflected column), 2 Upsample dgrad
 *                (4 taps [w2, w1+w2, w0+w1, w0]); dkind 1/2 need k == 3.  (may be NULL)
 * out_dtype OSUF_DT_*; f_ld / d_ld = row strides and *_tapstride = tap strides of the destinations, in elements. */
int osuf_pack_weight(const float* w, int O, int I, int k, int out_dtype, void* F, long f_ld, long f_tapstride, void* D, long d_ld,
                     long d_tapstride, int dkind, hipStream_t stream);

/* The same pack for MANY weights in one launch (after an optimizer step every packed operand of the model is stale at once).
 * descs: DEVICE array of n descriptors, one per weight, fields as osuf_pack_weight's arguments; block0 = running sum of
 * ceil(O/32) * ceil(I/32) over the preceding descriptors (descs[0].block0 == 0), total_blocks = that sum over all n.
 * All destinations share out_dtype.  The table is only read: it can be built once and reused while the pointers stay put. */
#ifndef OSUF_PACK_DESC_DEFINED
#define OSUF_PACK_DESC_DEFINED
typedef struct osuf_pack_desc {
  const float* w;
  void* F;
  void* D;
  long f_ld, f_tapstride, d_ld, d_tapstride;
  int O, I, k, dkind;
  int block0, reserved;
} osuf_pack_desc;                                   /* 80 bytes */
#endif
int osuf_pack_weight_group(const osuf_pack_desc* descs, int n, int total_blocks, int out_dtype, hipStream_t stream);

/* LoRA / DoRA adapters folded into an effective weight (reference: osu_fusion/modules/lora_layers.py:16-26 get_weight_norm,
 * :72-92 DoraConv1dLayer.forward, :284-298 get_delta_weight; peft 0.12 DoraLinearLayer for attn.to_q / attn.to_kv as wired at
 * trainer_peft.py:236-244).  W (O, IK) fp32 frozen base, A (r, IK), B (O, r), mag (O) or NULL for plain LoRA, IK = in*k:
 *   V = W + scaling * B A;   g = mag / ||V||_row  (1 without mag);   Weff = g * V.   g (O floats) may be NULL. */
int osuf_dora_effective(const float* W, const float* A, const float* B, const float* mag, int O, int IK, int r, float scaling,
                        float* Weff, float* g, hipStream_t stream);
/* The same effective weight without the full-size fp32 round trip (what the training step uses):
 *   osuf_dora_gain            g[o] = mag[o] / ||W[o] + s (BA)[o]||_2 (mag NULL: g = 1; partial = workspace of ceil(I/32)*O floats), plus
 *                             the transposed rank-r operand (s g B)^T = [r][O] in f32 (sgbt32) and bf16 (sgbt16), either may be NULL
 *   osuf_pack_weight_adapted  osuf_pack_weight of g[o] * (w + s * B A), formed tile by tile from the frozen master (g NULL: g = 1) */
int osuf_dora_gain(const float* W, const float* A, const float* B, const float* mag, int O, int I, int k, int r, float scaling,
                   float* partial, float* g, float* sgbt32, void* sgbt16, hipStream_t stream);
int osuf_pack_weight_adapted(const float* w, const float* A, const float* B, const float* g, float scaling, int r, int O, int I, int k,
                             int out_dtype, void* F, long f_ld, long f_tapstride, void* D, long d_ld, long d_tapstride, int dkind,
                             hipStream_t stream);
/* Tail of the rank-r adapter gradients in one launch: dB (+)= sg[o] * tb[o][q]; dA[q][i][t] (+)= gt[k-1-t][i][q]; and, when dm is
 * given, dm (+)= (s0 - bias * s1) / m   (tb = dy^T u, gt = x^T du per tap, s0 = sum dy*y, s1 = sum dy; bias may be NULL). */
int osuf_adapter_finish(const float* tb, const float* sg, float* dB, const float* gt, float* dA, const float* s0, const float* s1,
                        const float* bias, const float* m, float* dm, int O, int I, int k, int r, int accumulate, hipStream_t stream);

/* ---- embedding-sized linears (skinny.hip)    replaces: time_mlp / cond_mlp (modules/unet.py:356-367), the FiLM projection
 *      (residual.py:104-111,126-133) and GlobalContext's squeeze-excite MLP (residual.py:20-26,33-37) -- nn.Linear / 1x1 Conv1d on
 *      (B, features) rows.  fp32 in / out, W (N, K) fp32 master read in place; mode OSUF_DT_BF16 rounds both operands to bf16
 *      (autocast semantics), OSUF_DT_F32 is exact f32.  in_act: 0 none, 1 SiLU applied to x; out_act: 0 none, 2 sigmoid.
 *        fwd: y = out_act(in_act(x) W^T + b)
 *        bwd: dz = dy * out_act'(y); dx = (dz W) * in_act'(x) [dx may be NULL]; dW (+)= dz^T in_act(x), db += colsum(dz)
 *             [dW / db may be NULL; db needs dW] ---- */
int osuf_skinny_fwd(int mode, const float* x, long ldx, const float* W, const float* bias, float* y, long ldy, int M, int N, int K,
                    int in_act, int out_act, hipStream_t stream);
int osuf_skinny_bwd(int mode, const float* dy, long lddy, const float* y, long ldy, const float* x, long ldx, const float* W,
                    float* dx, long lddx, float* dW, float* db, int M, int N, int K, int in_act, int out_act, int accumulate,
                    hipStream_t stream);

/* Group forms for linears that share ONE input x (M, K) -- the FiLM projections: every ResidualBlock applies
 * Sequential(SiLU, Linear(2048, 2C)) to the same embedding (residual.py:104-111, 35 of them per forward).  descs: DEVICE array.
 *   osuf_skinny_fwd_group: y_i = in_act(x) W_i^T + b_i for all i in one launch; block0 = running sum of ceil(N_i / 32).
 *   osuf_skinny_dx_group : dx = in_act'(x) * sum_i dy_i W_i (dx overwritten);      block0 = running sum of ceil(N_i / 512).
 * K and every N_i multiples of 8, rows 16-byte aligned (OSUF_EINVAL otherwise: use the per-linear entry points).  The weight
 * gradients stay per linear (osuf_skinny_bwd with dx == NULL), so each is complete as soon as its block's backward has run. */
#ifndef OSUF_LINEAR_DESC_DEFINED
#define OSUF_LINEAR_DESC_DEFINED
typedef struct osuf_linear_desc {
  const float* W;          /* (N, K) fp32 master weight */
  const float* bias;       /* (N) or NULL */
  float* y;                /* forward output rows, row stride ldy */
  const float* dy;         /* backward: gradient of y, row stride lddy */
  long ldy, lddy;
  int N, block0;
} osuf_linear_desc;                                 /* 56 bytes */
#endif
int osuf_skinny_fwd_group(int mode, const float* x, long ldx, const osuf_linear_desc* descs, int n, int total_blocks, int M, int K,
                          int in_act, hipStream_t stream);
int osuf_skinny_dx_group(int mode, const osuf_linear_desc* descs, int n, int total_slices, const float* x, long ldx, float* dx, long lddx,
                         int M, int K, int in_act, hipStream_t stream);

/* ---- audio front end (audio.hip)    replaces: scripts/dataset_creator.py:36-55 load_audio's feature step,
 *      np.log(np.abs(librosa.vqt(y, sr=22050, hop_length=176, fmin=C0, n_bins=96, bins_per_octave=12)) + 1e-10)
 *      (called from trainer.py:104, trainer_peft.py:107, inference_gradio.py:56).
 *      wave_pad: zero-padded mono waveform (n_pad floats, >= (frames-1)*hop + K); bank: [2*bins][K] fp32 wavelet bank (real rows,
 *      then imaginary rows; built by osufusion_amd/audio.py); scale: [bins] = sqrt(filter length); spec_ws: frames*2*bins floats;
 *      out[k][t] = log(scale[k] * |sum_n wave_pad[t*hop + n] * (bank[k][n] + i bank[bins+k][n])| + eps), row stride ldo.
 *      hop and K multiples of 4.  osuf_vqt_logmag is the second stage alone (spec: [frames][ld >= 2*bins]). ---- */
int osuf_log_vqt(const float* wave_pad, long n_pad, const float* bank, int K, int bins, int hop, const float* scale, float eps,
                 float* spec_ws, float* out, long ldo, long frames, hipStream_t stream);
int osuf_vqt_logmag(const float* spec, long ld, float* out, long ldo, const float* scale, int bins, long frames, float eps,
                    hipStream_t stream);

/* The two helpers of librosa.vqt's octave recursion (core/constantq.py: after each octave, while the hop is even, the signal is
 * resampled by 1/2 with res_type="soxr_hq", scale=True): osuf_fir_decimate2 -- out[m] = sum_j taps[j] * in[2m + j - (ntaps-1)/2],
 * zeros outside the input, ntaps odd (the taps carry the sqrt(2)); osuf_frame_rows -- out[t][i] = in[t*hop + i] for the octaves
 * whose hop (22, 11 samples) is not a multiple of 4 and cannot be read in place by osuf_log_vqt (call it with hop = K then). */
int osuf_fir_decimate2(const float* in, long n_in, const float* taps, int ntaps, float* out, long n_out, hipStream_t stream);
int osuf_frame_rows(const float* in, long n_in, int hop, int K, float* out, long frames, hipStream_t stream);

/* Measurement aid (no reference counterpart): sustained shader clock under an MFMA (mode 1) or VALU (mode 0) load.
 * out[2*block] = shader cycles, out[2*block+1] = 100 MHz wall ticks. */
int osuf_clock_probe(int blocks, int iters, int mode, long* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OSUFUSION_HIP_H */
