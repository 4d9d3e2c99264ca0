/* dcrafter_hip.h — C ABI of libdcrafter_hip.so: the MI355X (gfx950) device path of the DynamiCrafter denoising
 * loop (DDIM sampler -> 3D UNet forward -> AutoencoderKL encode/decode).
 *
 * The reference (87003697/DynamiCrafter) contains no native code; its "FFI" for this path is PyTorch's ATen
 * dispatch underneath the lvdm Python classes. Each entry point below therefore names the reference call site
 * (path:line under the reference root) whose ATen ops it replaces. The Python classes in
 * the dynamicrafter_amd/lvdm package keep the reference's constructor/forward signatures and call these through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every function returns 0 on success, a hipError_t (>0) from the launch, or a DC_ERR_* (<0) for bad arguments;
 *     nothing throws, nothing allocates, nothing synchronises: all work is enqueued on `stream` (a hipStream_t
 *     passed as void*), so every call is hipGraph-capturable.
 *   - device pointers only. bf16 = raw uint16 bits. Activations are channels-last rows: [rows, C] with
 *     row = ((b*T + t)*H + y)*W + x and `ld*` the row stride in elements.
 *   - weights are repacked once by the host: Linear [N][K] as in torch; conv3x3 [Cout][Cin/64][kh*kw][64];
 *     temporal conv [Cout][Cin/64][kt][64] (K = 64-channel slice, tap, channel); N zero-padded to a multiple of
 *     128 rows; biases/affine params fp32.
 */
#ifndef DCRAFTER_HIP_H
#define DCRAFTER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DC_ERR_SHAPE (-1)   /* unsupported / inconsistent shape or stride */
#define DC_ERR_ARG   (-2)   /* null pointer or bad enum */

#define DC_GEMM_OUT_F32 1   /* C is float32 (no residual) */
#define DC_GEMM_GEGLU   2   /* W = [value half ; gate half] (N = 2*Nout): C = (xW_v+b_v) * gelu_erf(xW_g+b_g) */
#define DC_GEMM_GELU    4   /* C = gelu_erf(xW + b)  (Resampler FeedForward, lvdm/modules/encoders/resampler.py:27-34) */

typedef struct DcGemmParams {
    const uint16_t* A;        /* activation rows (bf16) */
    const uint16_t* W;        /* weights [n_pad][K] (bf16), rows >= N are zero */
    void* C;                  /* output rows, bf16 or f32 (flags) */
    const float* bias;        /* [N] or NULL */
    const float* rowvec;      /* optional broadcast add: rowvec[(m / rows_per_vec) * rowvec_ld + n] (timestep-embedding add) */
    const uint16_t* residual; /* optional bf16 residual rows, added after everything else */
    int lda, ldc, ldr, rowvec_ld;
    int rows_per_vec;
    int M, N, K, n_pad;
    int mode;                 /* 0 plain, 1 conv2d 3x3, 2 temporal conv 3x1x1 */
    int Cin;                  /* channels per tap (mode 1/2): K = taps*Cin */
    int IH, IW, OH, OW;       /* mode 1: source frame size (before the fused upsample) and output frame size */
    int stride, pad, ups;     /* mode 1: stride 1|2, top/left pad (bottom/right implied by OH/OW), ups=1 -> nearest x2 of the source */
    int T, HW;                /* mode 2: frames per clip, rows per frame */
    int flags;
    float alpha;              /* output scale applied last (before the residual add) */
    void* workspace;          /* optional device scratch for split-K partial sums (dc_gemm_workspace_bytes()); NULL = never split */
    long long workspace_bytes;
} DcGemmParams;

/* Name of the kernel family the last dc_gemm_conv call of this host thread dispatched to. A thread-local DIAGNOSTIC label
 * for profilers (bench.py's per-kernel table): it carries no state any compute entry reads - the compute ABI stays
 * stateless. */
const char* dc_gemm_last_variant(void);

/* Which kernels take the launches with 320-wide tiles. bit 0: the one-wave-per-SIMD kernel on v_mfma_f32_16x16x32_bf16
 * (gemm_pipe16.h: activations straight into registers, no barrier in the K loop) for the 3x3 convs, bit 1: also for plain /
 * temporal launches with K >= 1920; bit 4 (16): the ping-pong kernel (gemm_pp.h: 4-wave workgroups, two per CU, whose epilogues
 * overlap each other's K loops) for GEGLU projections with K <= 640, bit 5 (32): for any K. Default 19 (bits 0, 1, 4): the matrix
 * pipe is power-managed on MI355X and holds a higher clock on the 16x16x32 shape. 0 = the 8-wave LDS-DMA kernels everywhere. Bit 3 is accepted and ignored (it once chose between
 * the two MFMA shapes of that kernel: plans 9 / 11 = 1 / 3); bit 2 is rejected (the 32x32x16 form and the LDS-window conv kernel
 * lost every measured shape and live in tools/experimental); bit 6 (64): the split-K plans "whole waves of tiles + the remainder
 * cut along K" as two launches instead of one (bit-identical results: tests, A/B); bit 7 (128): the narrow-output 3x3 conv kernel
 * (N <= 16: conv_out of the UNet and of the AE decoder) off, the tile kernels take those launches; bit 8 (256): likewise without the window kernel for N = 128 (the AE's
 * full-resolution ResnetBlock convs). All plans give the same results to bf16 rounding and each is
 * bit-reproducible. Process-wide (env DC_GEMM_PLAN sets the initial value, validated the same way). Returns the previous plan,
 * or DC_ERR_ARG. */
int dc_gemm_set_plan(int plan);

/* Recommended size of DcGemmParams.workspace (one buffer per stream; contents are scratch, no initialisation). */
int64_t dc_gemm_workspace_bytes(void);

/* Linear / 1x1 conv / conv2d 3x3 / Conv3d(3,1,1) on MFMA.
 * replaces: nn.Linear  lvdm/modules/attention.py:53-57,75-76,269,290,336,362,418,438
 *           nn.Conv2d  lvdm/modules/networks/openaimodel3d.py:68,96,154,179,187,386,545 (+F.interpolate :103 when ups=1)
 *           nn.Conv3d  lvdm/modules/networks/openaimodel3d.py:255-266
 *           nn.Conv2d  lvdm/modules/networks/ae_modules.py:31-50,96-106,117-126,161-186 ; lvdm/models/autoencoder.py:34-35
 *           GEGLU      lvdm/modules/attention.py:415-422 (DC_GEMM_GEGLU) */
int dc_gemm_conv(const DcGemmParams* p, void* stream);

/* GroupNorm (+ optional SiLU) over channels-last rows; statistics in fp32.
 * One "instance" = rows_per_inst consecutive rows sharing statistics: H*W for the 4-D norms, T*H*W for the 5-D
 * norms of TemporalConvBlock / TemporalTransformer.
 * replaces: GroupNormSpecific lvdm/basics.py:76-87; nn.GroupNorm openaimodel3d.py:256-265, attention.py:265,331;
 *           Normalize + nonlinearity lvdm/modules/networks/ae_modules.py:10-16
 * workspace: dc_groupnorm_workspace_bytes() bytes of device scratch. */
int dc_groupnorm(const uint16_t* x, int ldx, uint16_t* y, int ldy, const float* gamma, const float* beta,
                 int C, int groups, int n_inst, int rows_per_inst, float eps, int silu,
                 float* workspace, void* stream);
int64_t dc_groupnorm_workspace_bytes(int n_inst, int groups, int rows_per_inst);

/* LayerNorm over the last dim of [rows, C]. replaces nn.LayerNorm lvdm/modules/attention.py:225-227 */
int dc_layernorm(const uint16_t* x, int ldx, uint16_t* y, int ldy, const float* gamma, const float* beta,
                 int rows, int C, float eps, void* stream);

/* softmax(q k^T * scale) v, head_dim 64, flash-style (no score matrix in HBM).
 * q rows: [batch][Lq] at q + (b*q_bstride + i)*ldq + h*64 ; k/v rows: [batch][Lk] likewise with kv_bstride.
 * accumulate != 0: o += acc_scale * result (image cross-attention branch), else o = result.
 * All row strides % 8 == 0 and q/k/v/o 16-byte aligned (16-byte loads and stores).
 * replaces: CrossAttention.forward core lvdm/modules/attention.py:101-142 (einsum/softmax/einsum, and the
 *           xformers path :166-207) for spatial self-attention and text / image cross-attention. */
int dc_flash_attn_d64(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o,
                      int ldq, int ldk, int ldv, int ldo, int batch, int heads, int Lq, int Lk,
                      int64_t q_bstride, int64_t kv_bstride, float scale, int accumulate, float acc_scale,
                      void* stream);

/* Test / measurement hook for the long self-attention path of dc_flash_attn_d64 (Lq >= 512, Lk >= 256, Lk % 64 == 0, no
 * accumulate: the one-wave-per-SIMD pipelined kernel, which runs softmax without a running maximum and repeats a workgroup's
 * block with a running-max pass when a row sum leaves [2^-100, 2^100)). mode bit 0: run the running-max pass directly; bit 1:
 * two 32-row query blocks per wave for every shape (default: three when Lq % 384 == 0); thr (0..64, exp2 units): how far a
 * score must exceed the running max before that pass rescales its state. Process-wide (atomics: a launch reads each once);
 * default mode 0, thr 8. Returns 0 or DC_ERR_ARG (mode outside 0..3). */
int dc_flash_attn_set_mode(int mode, float thr);

/* Temporal self-attention over T <= 16 frames, head_dim 64: for every (clip b, position p, head h) the T rows
 * (b, t, p) attend to each other. qkv rows are [q | k | v] (3*heads*64 wide, row stride ld).
 * replaces: CrossAttention.forward lvdm/modules/attention.py:81-144 as used by TemporalTransformer :365-412. */
int dc_temporal_attn_d64(const uint16_t* qkv, int ld, uint16_t* o, int ldo, int B, int T, int HW, int heads,
                         float scale, void* stream);

/* out[m][n] = act_out( bias[n] + sum_k act_in(in[m][k]) * W[n][k] ), m < 8; in/out fp32, W bf16.
 * act: 0 none, 1 SiLU. accumulate: out += result.
 * replaces: time_embed / fps_embedding MLPs openaimodel3d.py:370-382,551,575 and ResBlock emb_layers :168-174,219 */
int dc_gemv_small(const float* in, int ld_in, const uint16_t* W, const float* bias, float* out, int ld_out,
                  int M, int N, int K, int act_in, int act_out, int accumulate, void* stream);

/* Sinusoidal embedding: out[b][0:half]=cos(t_b*f_i), out[b][half:]=sin(t_b*f_i), f_i=exp(-ln(max_period)*i/half).
 * t is read from device memory: t_table[t_index[0]*t_stride + b] if t_index != NULL else t_table[b].
 * replaces: timestep_embedding lvdm/models/utils_diffusion.py:8-28 */
int dc_timestep_embedding(const int64_t* t_table, const int32_t* t_index, int t_stride, float* out, int B, int dim,
                          float max_period, void* stream);

/* UNet input assembly: cat([x, c_concat], dim=1) of [B,Cx,T,H,W] + [B,Cc,T,H,W] fp32 tensors -> channels-last
 * bf16 rows [nrep*B*T*H*W, c_pad] (channels >= Cx+Cc zeroed); the batch is written nrep times (nrep=2 serves
 * the cond/uncond pair of classifier-free guidance from one latent).
 * replaces: torch.cat ddpm3d.py:1256 + rearrange openaimodel3d.py:566 */
int dc_pack_latent(const float* x, const float* cc, uint16_t* out, int B, int Cx, int Cc, int T, int HW,
                   int c_pad, int nrep, void* stream);

/* Generic layout helpers (channels-last bf16 <-> NCHW fp32), used at the AutoencoderKL boundary.
 * nchw_to_rows: x[N][C][HW] fp32 -> rows[N*HW][c_pad] bf16 (scaled by `scale`, pad channels zero)
 * rows_to_nchw: rows[N*HW][ld] (bf16 or f32) -> y[N][C][HW] fp32 */
int dc_nchw_to_rows(const float* x, uint16_t* out, int N, int C, int HW, int c_pad, float scale, void* stream);
int dc_rows_to_nchw(const void* rows, int ld, int rows_f32, float* y, int N, int C, int HW, float scale, void* stream);

/* im2col for a 3x3 / stride 1 / pad 1 conv with <= 8 input channels (the UNet's conv_in: 4 latent + 4 concat channels,
 * openaimodel3d.py:379-381 `conv_nd(dims, in_channels, model_channels, 3, padding=1)`): x rows [n_img*H*W][ldx] bf16 whose first
 * 8 channels are read -> out rows [n_img*H*W][ldo >= 128]: elements 8 t .. 8 t + 7 = the 8 channels of tap t = kh*3 + kw (zeros
 * outside the image), elements 72 .. 127 zero. The conv is then dc_gemm_conv (mode 0) on these rows with the weight reordered to
 * [Cout][128] (k = 8 (kh*3 + kw) + c): two K tiles instead of nine, each eight times denser. */
int dc_im2col3x3_c8(const uint16_t* x, int ldx, uint16_t* out, int ldo, int n_img, int H, int W, void* stream);

/* 2-D strided copy of bf16 rows (skip-connection concat: torch.cat openaimodel3d.py:596). cols % 8 == 0. */
int dc_copy2d(const uint16_t* src, int lds, uint16_t* dst, int ldd, int rows, int cols, void* stream);

/* Per-frame context assembly: context [B, 77 + T*L, D] fp32 -> [B*T, 77+L, D] bf16 where frame (b,t) gets the
 * 77 text tokens of b followed by image tokens t*L..t*L+L-1. replaces openaimodel3d.py:555-562 */
int dc_build_context(const float* ctx, uint16_t* out, int B, int T, int n_text, int L, int D, void* stream);

/* Row softmax fp32 -> bf16 (AutoencoderKL mid attention, ae_modules.py:68). */
int dc_softmax_rows(const float* x, int ldx, uint16_t* y, int ldy, int rows, int cols, void* stream);

/* dst[c][r] = src[r][c] on bf16 rows (V^T operand of the AutoencoderKL mid attention, ae_modules.py:71-74). */
int dc_transpose(const uint16_t* src, int lds, uint16_t* dst, int ldd, int rows, int cols, void* stream);

/* y = a + b elementwise on bf16 rows (AE residual adds where no GEMM epilogue is available). */
int dc_add_rows(const uint16_t* a, int lda, const uint16_t* b, int ldb, uint16_t* y, int ldy, int rows, int cols,
                void* stream);

/* DiagonalGaussianDistribution.sample x scale_factor: moments rows [N*HW][ld] (mean | logvar, zc channels each)
 * -> z[N][zc][HW] fp32 = scale * (mean + exp(0.5*clamp(logvar,-30,20)) * noise[N][zc][HW]); noise NULL -> mode().
 * replaces lvdm/distributions.py:25-40 + ddpm3d.py:611-618 */
int dc_vae_sample(const uint16_t* moments, int ld, const float* noise, float* z, int N, int zc, int HW,
                  float scale, void* stream);

typedef struct DcDdimParams {
    /* per-step scalar tables, indexed by step_index[0] (device) when step_index != NULL, else by `index` */
    const float* a_t;            /* ddim_alphas[index]            (fp32 as torch.full casts them, ddim.py:251) */
    const float* a_prev;         /* ddim_alphas_prev[index] */
    const float* sigma_t;        /* ddim_sigmas[index] */
    const float* sqrt_one_minus_at;
    const float* sqrt_acp_t;     /* model.sqrt_alphas_cumprod[t]           (v-param, ddpm3d.py:239-251) */
    const float* sqrt_1macp_t;   /* model.sqrt_one_minus_alphas_cumprod[t] */
    const float* scale_ratio;    /* ddim_scale_arr_prev[index]/ddim_scale_arr[index] or NULL (ddim.py:262-266) */
    const int32_t* step_index;   /* device counter or NULL */
    int index;
    int v_param;                 /* parameterization == "v" */
    float cfg_scale;             /* unconditional_guidance_scale; e_uncond may be NULL when 1.0 */
    float cfg_img;               /* 3-branch CFG (ddim_multiplecond.py:234); used when e_img != NULL */
    float guidance_rescale;      /* 0 -> off */
    float temperature;
    int e_nchw;                  /* 0: e_* are channels-last rows [B*T*HW, ld_e]; 1: e_* are [B, C, T*HW] like x */
    int64_t noise_step_stride;   /* with step_index: noise for this step starts at noise + step_index[0]*stride */
} DcDdimParams;

/* One DDIM update for B clips. e_* are the UNet outputs as channels-last fp32 rows [B*T*HW, ld_e] (first C
 * channels valid); x, noise, x_prev, pred_x0 are [B, C, T*HW] fp32 (the reference's NCTHW layout).
 * workspace: >= 16 * B * 256 floats.
 * replaces: p_sample_ddim lvdm/models/samplers/ddim.py:226-277 + rescale_noise_cfg utils_diffusion.py:147-157 */
int dc_ddim_step(const DcDdimParams* p, const float* e_cond, const float* e_uncond, const float* e_img, int ld_e,
                 const float* x, const float* noise, float* x_prev, float* pred_x0, int B, int C, int THW,
                 float* workspace, void* stream);

/* DynamiCrafter's dual cross-attention in one launch: o = softmax(s q k^T) v + scale2 * softmax(s q k2^T) v2 with two
 * independent softmaxes (text keys Lk, image keys Lk2) over the same queries; head_dim 64. q/o rows as in
 * dc_flash_attn_d64; k, v, k2, v2 rows share the stride ldkv and the batch stride kv_bstride (views into one fused
 * projection buffer). replaces CrossAttention.forward lvdm/modules/attention.py:128-142 (image_cross_attention) */
int dc_cross_attn_dual_d64(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* k2, const uint16_t* v2,
                           uint16_t* o, int ldq, int ldkv, int ldo, int batch, int heads, int Lq, int Lk, int Lk2,
                           int64_t q_bstride, int64_t kv_bstride, float scale, float scale2, void* stream);

/* Fused GEGLU FeedForward for dim = 320: out[M,320] = ((n W1v^T + b1v) * gelu(n W1g^T + b1g)) W2^T + b2 (+ residual),
 * n = x, or LayerNorm(x; ln_gamma, ln_beta, ln_eps) when ln_gamma != NULL (rounded to bf16 like dc_layernorm's output);
 * the [M,1280] intermediate never leaves the CU. x/out/residual: bf16 rows (ld % 8 == 0; residual may alias out and x).
 * w1: ff.net.0.proj.weight as dc_gemm_conv takes it, bf16 [>= 2560][320] (rows 0..1279 value, 1280..2559 gate), b1 fp32
 * [2560]; w2p: ff.net.2.weight bf16 [>= 320][1280] with the K order permuted inside every 32-channel chunk: position
 * 16 s + 8 h + e of a chunk holds channel 8 (2 s + e / 4) + 4 h + e % 4 (s, h in {0,1}, e in 0..7); b2 fp32 [320].
 * replaces FeedForward(GEGLU) lvdm/modules/attention.py:415-442 with norm3 and the residual add of
 * BasicTransformerBlock._forward :246 (x = ff(norm3(x)) + x) */
int dc_ff_geglu_fused320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                         const uint16_t* w1, const float* b1, const uint16_t* w2p, const float* b2, const uint16_t* residual,
                         int ldr, uint16_t* out, int ldo, int M, void* stream);

/* dc_ff_geglu_fused320 with the transformer's proj_out behind it: out = residual2 + h2 wp^T + bp, h2 = x + FeedForward(norm3(x))
 * (h2 itself is not stored). wp: proj_out.weight bf16 [>= 320][320] with the k order of w2p inside every 32-chunk; bp fp32
 * [320]; residual2 (the transformer's input) / out: bf16 rows; out may alias residual2 but not x.
 * replaces BasicTransformerBlock._forward :246 + SpatialTransformer.forward :307-310 / TemporalTransformer.forward :404-412
 * (proj_out and the `+ x_in`) of lvdm/modules/attention.py at the UNet's level 0 */
int dc_ff_geglu_proj_fused320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                              const uint16_t* w1, const float* b1, const uint16_t* w2p, const float* b2, const uint16_t* wp,
                              const float* bp, const uint16_t* residual2, int ldr2, uint16_t* out, int ldo, int M, void* stream);

/* Norm + Linear for dim K = 320 or 640: out[M,N] = n W^T (+ bias), N % 32 == 0; x/out: bf16 rows (ld % 8 == 0, 16-byte
 * aligned); w: bf16 [>= N][K] as dc_gemm_conv takes a Linear weight; bias fp32 [N] or NULL. The normalised rows live in
 * registers only (rounded to bf16 where dc_layernorm / dc_groupnorm round their outputs).
 * dc_ln_linear: n = x, or LayerNorm(x; ln_gamma, ln_beta, ln_eps) when ln_gamma != NULL.
 *   replaces norm1 -> attn1.to_q/k/v (one [3K, K] weight) and norm2 -> attn2.to_q of BasicTransformerBlock._forward
 *   lvdm/modules/attention.py:242-245 (CrossAttention.forward :101-105) at the UNet's levels 0 and 1
 * dc_gn_linear: n = GroupNorm(x) (no activation) with the statistics computed by dc_groupnorm_stats (fp32
 *   [n_inst][groups][2] = mean, rstd); rows_per_inst % 128 == 0 and M % rows_per_inst == 0 (a 128-row tile never straddles
 *   two instances).
 *   replaces SpatialTransformer.forward norm -> proj_in lvdm/modules/attention.py:296-301 and TemporalTransformer.forward
 *   norm -> proj_in :367-377 */
int dc_ln_linear(const uint16_t* x, int ldx, int K, const float* ln_gamma, const float* ln_beta, float ln_eps, const uint16_t* w,
                 const float* bias, uint16_t* out, int ldo, int M, int N, void* stream);
int dc_groupnorm_stats(const uint16_t* x, int ldx, int C, int groups, int n_inst, int rows_per_inst, float eps,
                       float* workspace, float* stats_out, void* stream);
int dc_gn_linear(const uint16_t* x, int ldx, int K, const float* gamma, const float* beta, const float* stats, int groups,
                 int rows_per_inst, const uint16_t* w, const float* bias, uint16_t* out, int ldo, int M, int N, void* stream);

/* out[M,N] = residual + x W^T + bias for K = 320 or 640 (N % 32 == 0; rows bf16, ld % 8 == 0, 16-byte aligned; residual may
 * alias out): the activation rows of a workgroup stay in registers (the kernel of dc_ln_linear without a norm), the residual
 * rows are fetched a chunk ahead in the stores' coalesced pattern and added in fp32 (one rounding).
 *   replaces attn1 / attn2 .to_out[0] + the residual add of BasicTransformerBlock._forward lvdm/modules/attention.py:242-245
 *   (CrossAttention.forward :143-144) and proj_out + x_in of SpatialTransformer / TemporalTransformer.forward :309-310, :404-412
 *   at the UNet's levels 0 and 1 */
int dc_linear_residual(const uint16_t* x, int ldx, int K, const uint16_t* w, const float* bias, const uint16_t* residual, int ldr,
                       uint16_t* out, int ldo, int M, int N, void* stream);

/* LayerNorm + to_q/k/v + attention over the T = 16 frames of every spatial position for dim 320 (5 heads x 64) in one launch:
 * out[M, 320] = softmax_T(q k^T scale) v, [q | k | v] = LayerNorm(x) wqkv^T, rows ordered (clip, frame, position), M = B*16*HW,
 * HW % 8 == 0. wqkv: bf16 [>= 960][320] (to_q, to_k, to_v rows). The qkv tensor never reaches HBM; bf16 roundings as in
 * dc_layernorm -> dc_gemm_conv -> dc_temporal_attn_d64. out must not alias x.
 * replaces TemporalTransformer's norm1 -> attn1 / norm2 -> attn2 up to (not including) to_out: lvdm/modules/attention.py:242-244
 * (BasicTransformerBlock._forward) with CrossAttention.forward :101-125, at the UNet's level 0 */
int dc_ln_qkv_temporal_attn320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                               const uint16_t* wqkv, uint16_t* out, int ldo, int B, int T, int HW, float scale, void* stream);
/* The same for dim 640 (10 heads x 64: level 1): wqkv bf16 [>= 1920][640]; out[M, 640]. */
int dc_ln_qkv_temporal_attn640(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                               const uint16_t* wqkv, uint16_t* out, int ldo, int B, int T, int HW, float scale, void* stream);

/* GroupNorm (statistics from dc_groupnorm_stats, instances = clips) + SiLU + temporal convolution (3,1,1), zero padding in
 * time, (+ residual) for C = 320 or 640 input channels in one launch; rows ordered (clip, frame, position), T = 16,
 * HW % 8 == 0, N % 32 == 0. w: the temporal-conv weight as dc_gemm_conv takes it (bf16 [>= N][3 C], k = (64-channel slice,
 * tap, channel)); bias fp32 [N]. The activated copy never reaches HBM; roundings as dc_groupnorm (silu) -> dc_gemm_conv.
 * replaces TemporalConvBlock conv1..conv4 lvdm/modules/networks/openaimodel3d.py:239-279 at the UNet's levels 0 and 1 */
int dc_gn_silu_tconv3(const uint16_t* x, int ldx, int C, const float* gamma, const float* beta, const float* stats, int groups,
                      const uint16_t* w, const float* bias, const uint16_t* residual, int ldr, uint16_t* out, int ldo, int B, int T,
                      int HW, int N, void* stream);

/* ---- conditioning encoders (once per clip) ---------------------------------------------------- */

/* Multi-head attention for any (even) head width d <= 256 and Lk <= 1024, optional causal mask (key j visible to query
 * i iff j <= i): o[b, i, h*d..] = softmax_j(scale * q[b,i,h] . k[b,j,h]) v[b,j,h]. q/o rows [B*Lq, >= heads*d], k/v rows
 * [B*Lk, >= heads*d] (bf16, row strides ld*). Needs dc_attn_small_lds_bytes(Lk, d) <= 160 KiB.
 * replaces open_clip's ResidualAttentionBlock attention (nn.MultiheadAttention, 16 heads x 80 in the ViT-H/14 vision
 * tower, 16 x 64 causal in the text tower) as driven by lvdm/modules/encoders/condition.py:216-234, 364-368 */
int dc_attn_small(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, int ldq, int ldk, int ldv, int ldo,
                  int B, int heads, int Lq, int Lk, int d, float scale, int causal, void* stream);
int64_t dc_attn_small_lds_bytes(int Lk, int d);

/* CLIP image preprocessing: img[N][3][H][W] fp32 in [-1,1] -> out[N][3][OH][OW] fp32 = normalize((resize(img)+1)/2).
 * resize = kornia.geometry.resize(bicubic, align_corners=True, antialias): when downscaling and antialias != 0, a
 * separable Gaussian blur (sigma = max((factor-1)/2, 0.001), ks = int(max(4 sigma, 3)) made odd, reflect borders) first;
 * tmp0/tmp1: N*3*H*W floats each (only used with the blur). mean3/std3: HOST pointers to 3 floats.
 * replaces FrozenOpenCLIPImageEmbedderV2.preprocess lvdm/modules/encoders/condition.py:322-330 */
int dc_clip_preprocess(const float* img, float* tmp0, float* tmp1, float* out, int N, int C, int H, int W, int OH, int OW,
                       int antialias, const float* mean3, const float* std3, void* stream);

/* Non-overlapping p x p patches of img[N][C][H][W] fp32 -> bf16 rows [N*(H/p)*(W/p)][kpad], column c*p*p + py*p + px
 * (the ViT patch embedding conv1 as a plain GEMM; zero columns up to kpad, kpad % 8 == 0).
 * replaces model.visual.conv1 lvdm/modules/encoders/condition.py:349-351 */
int dc_patchify(const float* img, uint16_t* rows, int N, int C, int H, int W, int p, int kpad, void* stream);

/* out[b*L + i][:] = table[tokens[b][i]][:] + pos[i][:] (bf16 rows of width D; token ids clamped to the vocabulary).
 * replaces token_embedding + positional_embedding lvdm/modules/encoders/condition.py:216-217 */
int dc_embed_tokens(const int64_t* tokens, const uint16_t* table, const uint16_t* pos, uint16_t* out, int B, int L, int D,
                    int vocab, void* stream);

/* Decoded clips video[N][C][T][H][W] fp32 in [-1,1] -> display frames out[T][H][N*W][C] uint8: clamp, (v+1)/2, *255,
 * truncation; the N clips of a batch side by side (torchvision make_grid(nrow=N, padding=0)).
 * replaces scripts/evaluation/inference.py:127-137 (save_results) / :151-160 (save_results_seperate, N = 1),
 * utils/save_video.py:35-42 */
int dc_frames_to_u8(const float* video, uint8_t* out, int N, int C, int T, int H, int W, void* stream);

/* Mask / x0 blend ahead of a DDIM step, in place on img [n] fp32: img = orig*mask + (1-mask)*img with
 * orig = x0 (clean != 0) or sqrt_acp_t[i]*x0 + sqrt_1macp_t[i]*qnoise (q_sample of x0 at the step's timestep);
 * i = step_index[0] (device counter; qnoise then starts at qnoise + i*noise_step_stride) or `index`.
 * replaces lvdm/models/samplers/ddim.py:174-180 + DDPM.q_sample lvdm/models/ddpm3d.py:305-308 */
int dc_mask_blend(float* img, const float* x0, const float* mask, const float* qnoise, const float* sqrt_acp_t,
                  const float* sqrt_1macp_t, const int32_t* step_index, int index, int64_t n,
                  int64_t noise_step_stride, int clean, void* stream);

/* step_index[0] += 1 (device-side loop counter for the graph-captured sampler). */
int dc_advance_counter(int32_t* counter, void* stream);

/* ---- stream / hipGraph plumbing (one capture per sampler configuration; replayed once per DDIM step) ---- */
int dc_stream_create(void** stream_out);
int dc_stream_destroy(void* stream);
int dc_stream_sync(void* stream);
int dc_graph_begin_capture(void* stream);
int dc_graph_end_capture(void* stream, void** graph_exec_out);
int dc_graph_launch(void* graph_exec, void* stream);
int dc_graph_destroy(void* graph_exec);
/* HIP-event timing on an arbitrary stream (bench.py roofline leg) */
int dc_event_create(void** ev_out);
int dc_event_record(void* ev, void* stream);
int dc_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out); /* synchronises on ev_stop */
int dc_event_destroy(void* ev);

/* The library's error word. Kernels that synchronise through LDS arrival counters (the one-wave-per-SIMD GEMM / conv kernel,
 * the ping-pong GEMM, the long self-attention) bound every wait so that a bookkeeping mistake cannot hang the GPU; a wave whose
 * wait ran out ORs a bit into one device word (1: gemm_pipe320x16, 2: flash attention K/V ring, 4: gemm_pp) and its results
 * are then NOT valid. This entry synchronises the device, copies the word to *out and, if `reset`, clears it. It is the one
 * entry that synchronises and is not capturable: call it at the host's own sync points (the sampler does after a run, bench.py
 * before it prints). Returns 0 or a hipError_t / DC_ERR_ARG. */
int dc_error_word_read(int* out, int reset);

const char* dc_version(void);

#ifdef __cplusplus
}
#endif
#endif
