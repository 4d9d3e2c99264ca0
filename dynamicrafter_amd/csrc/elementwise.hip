// Small / memory-bound kernels of the denoising path: embedding MLP GEMVs, sinusoidal embedding, layout packing
// at the latent / pixel boundaries, concat copies, row softmax, VAE posterior sampling, and the fused DDIM update.
#include "dc_common.h"
#include "dcrafter_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// out[m][n] = act_out(bias[n] + sum_k act_in(in[m][k]) W[n][k]);  one wave per output column n, M <= 8.
__global__ __launch_bounds__(256) void gemv_small_kernel(const float* __restrict__ in, int ld_in,
                                                         const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                         float* __restrict__ out, int ld_out, int M, int N, int K,
                                                         int act_in, int act_out, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) acc[m] = 0.f;
    const bf16_t* wrow = W + (size_t)n * K;
    for (int k0 = lane * 8; k0 < K; k0 += 64 * 8) {
        const uint4 raw = *reinterpret_cast<const uint4*>(wrow + k0);
        float w[8];
        unpack_bf8(raw, w);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (m < M) {
                const float* x = in + (size_t)m * ld_in + k0;
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xv = x[e];
                    if (act_in == 1) xv = silu_f(xv);
                    s += xv * w[e];
                }
                acc[m] += s;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (m < M) {
            float s = wave_sum(acc[m]);
            if (lane == 0) {
                if (bias) s += bias[n];
                if (act_out == 1) s = silu_f(s);
                float* dst = out + (size_t)m * ld_out + n;
                *dst = accumulate ? (*dst + s) : s;
            }
        }
    }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t_table, const int32_t* __restrict__ t_index,
                                          int t_stride, float* __restrict__ out, int B, int dim, float log_max_period) {
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    const int64_t off = t_index ? (int64_t)t_index[0] * t_stride : 0;
    const float t = (float)t_table[off + b];
    // freqs = exp(-ln(max_period) * i / half) in fp32, args = t * freqs (utils_diffusion.py:19-23)
    const float freq = expf(-log_max_period * (float)i / (float)half);
    const float arg = t * freq;
    out[(size_t)b * dim + i] = cosf(arg);
    out[(size_t)b * dim + half + i] = sinf(arg);
    if ((dim & 1) && i == 0) out[(size_t)b * dim + dim - 1] = 0.f;
}

// x [B][Cx][T*HW], cc [B][Cc][T*HW] fp32 -> rows [nrep][B][T*HW][c_pad] bf16
__global__ void pack_latent_kernel(const float* __restrict__ x, const float* __restrict__ cc, bf16_t* __restrict__ out,
                                   int B, int Cx, int Cc, int THW, int c_pad, int nrep) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (b, pos)
    const int64_t total = (int64_t)B * THW;
    if (idx >= total) return;
    const int b = (int)(idx / THW);
    const int pos = (int)(idx - (int64_t)b * THW);
    for (int c0 = 0; c0 < c_pad; c0 += 8) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            float v = 0.f;
            if (c < Cx) v = x[((size_t)b * Cx + c) * THW + pos];
            else if (c < Cx + Cc) v = cc[((size_t)b * Cc + (c - Cx)) * THW + pos];
            f[e] = v;
        }
        const uint4 pk = pack_bf8(f);
        for (int r = 0; r < nrep; ++r)
            *reinterpret_cast<uint4*>(out + ((size_t)r * total + idx) * c_pad + c0) = pk;
    }
}

__global__ void nchw_to_rows_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, int N, int C, int HW,
                                    int c_pad, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * HW;
    if (idx >= total) return;
    const int n = (int)(idx / HW);
    const int pos = (int)(idx - (int64_t)n * HW);
    for (int c0 = 0; c0 < c_pad; c0 += 8) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            f[e] = (c < C) ? x[((size_t)n * C + c) * HW + pos] * scale : 0.f;
        }
        *reinterpret_cast<uint4*>(out + (size_t)idx * c_pad + c0) = pack_bf8(f);
    }
}

__global__ void rows_to_nchw_kernel(const void* __restrict__ rows, int ld, int rows_f32, float* __restrict__ y, int N,
                                    int C, int HW, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * HW;
    if (idx >= total) return;
    const int n = (int)(idx / HW);
    const int pos = (int)(idx - (int64_t)n * HW);
    for (int c = 0; c < C; ++c) {
        float v;
        if (rows_f32) v = reinterpret_cast<const float*>(rows)[(size_t)idx * ld + c];
        else v = bf2f(reinterpret_cast<const bf16_t*>(rows)[(size_t)idx * ld + c]);
        y[((size_t)n * C + c) * HW + pos] = v * scale;
    }
}

__global__ void copy2d_kernel(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst, int ldd, int rows,
                              int vecs) {
    const int64_t total = (int64_t)rows * vecs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / vecs);
        const int v = (int)(idx - (int64_t)r * vecs);
        *reinterpret_cast<uint4*>(dst + (size_t)r * ldd + v * 8) =
            *reinterpret_cast<const uint4*>(src + (size_t)r * lds_ + v * 8);
    }
}

// conv_in (8 -> 320 channels, 3x3): as an implicit GEMM its K = 9 taps x a 64-channel slice of which 8 channels are real - eight
// times the MFMAs of the work. Gathered once into rows of 9 x 8 = 72 real K elements (+ 56 zeros: two K tiles of 64) it is a plain
// [M x 128] x [128 x 320] GEMM: 241 -> ~55 us at level 0 of the 1024 config. out[row][16 t .. 16 t + 15] (bytes) = the 8 channels
// of tap t = kh * 3 + kw of the row's pixel (zeros outside the image), t = 9 .. 15 zero. One 16-byte piece per thread.
__global__ void im2col3x3_c8_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ out, int ldo, int n_img, int H,
                                    int W) {
    const int64_t total = (int64_t)n_img * H * W * 16;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx & 15);
        const int64_t row = idx >> 4;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (t < 9) {
            const int px = (int)(row % W);
            const int64_t r1 = row / W;
            const int py = (int)(r1 % H);
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                v = *reinterpret_cast<const uint4*>(x + (size_t)(row + (int64_t)(t / 3 - 1) * W + (t % 3 - 1)) * ldx);
        }
        *reinterpret_cast<uint4*>(out + (size_t)row * ldo + t * 8) = v;
    }
}

__global__ void add_rows_kernel(const bf16_t* __restrict__ a, int lda, const bf16_t* __restrict__ b, int ldb,
                                bf16_t* __restrict__ y, int ldy, int rows, int vecs) {
    const int64_t total = (int64_t)rows * vecs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / vecs);
        const int v = (int)(idx - (int64_t)r * vecs);
        float fa[8], fb[8];
        unpack_bf8(*reinterpret_cast<const uint4*>(a + (size_t)r * lda + v * 8), fa);
        unpack_bf8(*reinterpret_cast<const uint4*>(b + (size_t)r * ldb + v * 8), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] += fb[e];
        *reinterpret_cast<uint4*>(y + (size_t)r * ldy + v * 8) = pack_bf8(fa);
    }
}

// ctx [B][n_text + T*L][D] fp32 -> out [B*T][n_text + L][D] bf16
__global__ void build_context_kernel(const float* __restrict__ ctx, bf16_t* __restrict__ out, int B, int T, int n_text,
                                     int L, int D) {
    const int vecs = D / 8;
    const int per_frame = (n_text + L) * vecs;
    const int64_t total = (int64_t)B * T * per_frame;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx / per_frame);
        const int rem = (int)(idx - (int64_t)f * per_frame);
        const int tok = rem / vecs, v = rem - tok * vecs;
        const int b = f / T, t = f - b * T;
        const int src_tok = tok < n_text ? tok : n_text + t * L + (tok - n_text);
        const float* s = ctx + ((size_t)b * (n_text + T * L) + src_tok) * D + v * 8;
        float fv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) fv[e] = s[e];
        *reinterpret_cast<uint4*>(out + ((size_t)f * (n_text + L) + tok) * D + v * 8) = pack_bf8(fv);
    }
}

// one workgroup per row; cols arbitrary
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y,
                                                           int ldy, int cols) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * ldx;
    bf16_t* yr = y + (size_t)row * ldy;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float mx = -3.0e38f;
    for (int c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < cols; c += 256) s += __expf(xr[c] - mx);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / s;
    for (int c = threadIdx.x; c < cols; c += 256) yr[c] = f2bf(__expf(xr[c] - mx) * inv);
}

__global__ void vae_sample_kernel(const bf16_t* __restrict__ mom, int ld, const float* __restrict__ noise,
                                  float* __restrict__ z, int N, int zc, int HW, float scale) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)N * HW;
    if (idx >= total) return;
    const int n = (int)(idx / HW);
    const int pos = (int)(idx - (int64_t)n * HW);
    for (int c = 0; c < zc; ++c) {
        const float mean = bf2f(mom[(size_t)idx * ld + c]);
        float lv = bf2f(mom[(size_t)idx * ld + zc + c]);
        lv = fminf(fmaxf(lv, -30.f), 20.f);
        float v = mean;
        if (noise) v += expf(0.5f * lv) * noise[((size_t)n * zc + c) * HW + pos];
        z[((size_t)n * zc + c) * HW + pos] = v * scale;
    }
}

// ---------------------------------------------------------------------------------------------------------
// DDIM update. Pass 1 (many workgroups): model_output = e_u + s (e_c - e_u) [+ 3-branch], per-block partial
// sums of {cfg, cfg^2, e_c, e_c^2} per sample. Pass 2: every workgroup re-reduces the (few) partials in a fixed
// order, then applies guidance rescale, v->eps/x0, dynamic rescale and the DDIM step elementwise.
constexpr int DDIM_BLOCKS = 256;   // partial blocks per sample

__device__ __forceinline__ size_t ddim_eoff(const DcDdimParams& p, int b, int c, int pos, int C, int THW, int ld_e) {
    // channels-last rows [b][pos][ld_e] (UNet row output) or the reference's [b][c][pos] layout
    return p.e_nchw ? ((size_t)b * C + c) * THW + pos : ((size_t)b * THW + pos) * ld_e + c;
}

__device__ __forceinline__ float ddim_cfg(const DcDdimParams& p, const float* ec, const float* eu, const float* ei,
                                          size_t off) {
    const float c = ec[off];
    if (!eu) return c;
    const float u = eu[off];
    if (ei) {   // e_u + cfg_img (e_ui - e_u) + s (e_c - e_ui)   (ddim_multiplecond.py:234)
        const float ui = ei[off];
        return u + p.cfg_img * (ui - u) + p.cfg_scale * (c - ui);
    }
    return u + p.cfg_scale * (c - u);
}

// (1 - a_prev - sigma^2).sqrt() with separately rounded multiply and subtractions, as torch evaluates it
// (ddim.py:271). hipcc contracts a*b-c into an FMA by default, which turns the exact 0 (or one-ulp) radicand of
// the first zero-terminal-SNR step into ~2e-8 and shifts x_prev by ~1e-4: contraction is switched off here.
__device__ __forceinline__ float ddim_dir_coef(float a_prev, float sigma) {
#pragma clang fp contract(off)
    const float s2 = sigma * sigma;
    const float one_minus = 1.f - a_prev;
    const float r = one_minus - s2;
    return sqrtf(fmaxf(r, 0.f));
}

__global__ __launch_bounds__(256) void ddim_partial_kernel(const DcDdimParams p, const float* __restrict__ ec,
                                                           const float* __restrict__ eu, const float* __restrict__ ei,
                                                           int ld_e, int C, int THW, float* __restrict__ ws) {
    __shared__ float red[4][4];
    const int b = blockIdx.y;
    const int64_t n = (int64_t)C * THW;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)DDIM_BLOCKS * 256) {
        const int pos = (int)(i / C), c = (int)(i - (int64_t)pos * C);
        const size_t off = ddim_eoff(p, b, c, pos, C, THW, ld_e);
        const float cfg = ddim_cfg(p, ec, eu, ei, off);
        const float e = ec[off];
        s0 += cfg; s1 += cfg * cfg; s2 += e; s3 += e * e;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s0; red[wave][1] = s1; red[wave][2] = s2; red[wave][3] = s3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        ws[((size_t)b * DDIM_BLOCKS + blockIdx.x) * 4 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(256) void ddim_apply_kernel(const DcDdimParams p, const float* __restrict__ ec,
                                                         const float* __restrict__ eu, const float* __restrict__ ei,
                                                         int ld_e, const float* x,      // x and x_prev may alias: the
                                                         const float* __restrict__ noise, float* x_prev,   // sampler updates in place
                                                         float* __restrict__ pred_x0, int C, int THW,
                                                         const float* __restrict__ ws) {
    __shared__ double tot[4];
    const int b = blockIdx.y;
    const int64_t n = (int64_t)C * THW;
    float factor = 1.0f;   // guidance rescale: cfg * (phi*std_text/std_cfg + 1 - phi)
    if (p.guidance_rescale > 0.f && eu) {
        if (threadIdx.x < 4) {
            double t = 0.0;
            for (int k = 0; k < DDIM_BLOCKS; ++k) t += (double)ws[((size_t)b * DDIM_BLOCKS + k) * 4 + threadIdx.x];
            tot[threadIdx.x] = t;
        }
        __syncthreads();
        const double cnt = (double)n;
        // unbiased std (torch.std default), utils_diffusion.py:152-153
        const double mean_cfg = tot[0] / cnt, mean_txt = tot[2] / cnt;
        const double var_cfg = fmax(tot[1] - cnt * mean_cfg * mean_cfg, 0.0) / (cnt - 1.0);
        const double var_txt = fmax(tot[3] - cnt * mean_txt * mean_txt, 0.0) / (cnt - 1.0);
        const float ratio = (float)sqrt(var_txt) / (float)sqrt(var_cfg);
        factor = p.guidance_rescale * ratio + (1.f - p.guidance_rescale);
    }
    const int idx = p.step_index ? p.step_index[0] : p.index;
    if (noise && p.step_index) noise += (size_t)idx * p.noise_step_stride;
    const float a_t = p.a_t[idx], a_prev = p.a_prev[idx], sigma = p.sigma_t[idx], s1m = p.sqrt_one_minus_at[idx];
    const float sq_acp = p.v_param ? p.sqrt_acp_t[idx] : 0.f;
    const float sq_1macp = p.v_param ? p.sqrt_1macp_t[idx] : 0.f;
    const float resc = p.scale_ratio ? p.scale_ratio[idx] : 1.0f;
    const float sqrt_at = sqrtf(a_t);
    const float sqrt_aprev = sqrtf(a_prev);
    // (1 - a_prev - sigma^2).sqrt() in fp32, same association as ddim.py:271. With zero-terminal-SNR +
    // uniform_trailing + eta=1 the radicand is +5.96e-8 at the first step; clamp at 0 so a one-ulp
    // difference can never produce NaN (the reference's get_fixed_ddim_sampler exists for that hazard).
    const float dir_coef = ddim_dir_coef(a_prev, sigma);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i / THW), pos = (int)(i - (int64_t)c * THW);     // NCTHW order for x / outputs
        const size_t eoff = ddim_eoff(p, b, c, pos, C, THW, ld_e);
        const size_t xoff = (size_t)b * n + i;
        const float mo = ddim_cfg(p, ec, eu, ei, eoff) * factor;
        const float xv = x[xoff];
        float e_t, px0;
        if (p.v_param) {
            e_t = sq_acp * mo + sq_1macp * xv;      // predict_eps_from_z_and_v  ddpm3d.py:247-251
            px0 = sq_acp * xv - sq_1macp * mo;      // predict_start_from_z_and_v ddpm3d.py:239-245
        } else {
            e_t = mo;
            px0 = (xv - s1m * e_t) / sqrt_at;       // ddim.py:258
        }
        px0 *= resc;                                 // ddim.py:262-266
        const float nz = noise ? sigma * noise[xoff] * p.temperature : 0.f;
        x_prev[xoff] = sqrt_aprev * px0 + dir_coef * e_t + nz;   // ddim.py:271-277
        pred_x0[xoff] = px0;
    }
}


// dst[c][r] = src[r][c] for bf16 rows; 64x64 tiles through LDS (pad column against bank conflicts)
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ src, int lds_, bf16_t* __restrict__ dst,
                                                        int ldd, int rows, int cols) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * lds_ + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[(size_t)c * ldd + r] = tile[tx][i];
    }
}

__global__ void advance_counter_kernel(int32_t* c) { if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += 1; }

inline int grid_for(int64_t total, int block, int cap) {
    int64_t g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}


// Inpainting-style blend ahead of a DDIM step (reference: lvdm/models/samplers/ddim.py:174-180): the latent is replaced by
// the (re-noised: DDPM.q_sample ddpm3d.py:305-308) original wherever mask == 1. fp32, in place, one element per thread.
__global__ void mask_blend_kernel(float* __restrict__ img, const float* __restrict__ x0, const float* __restrict__ mask,
                                  const float* __restrict__ qnoise, const float* __restrict__ sqrt_acp,
                                  const float* __restrict__ sqrt_1macp, const int32_t* __restrict__ step_index, int index,
                                  int64_t n, int64_t noise_step_stride, int clean) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int idx = step_index ? step_index[0] : index;
    float orig = x0[i];
    if (!clean) {
        const float q = qnoise[(step_index ? (int64_t)idx * noise_step_stride : 0) + i];
        orig = sqrt_acp[idx] * orig + sqrt_1macp[idx] * q;
    }
    const float m = mask[i];
    img[i] = orig * m + (1.0f - m) * img[i];
}


// Decoded clips -> display frames (reference: scripts/evaluation/inference.py:127-137, utils/save_video.py:35-42):
// clamp to [-1,1], (v+1)/2, *255, truncate to uint8; n clips side by side (make_grid(nrow=n, padding=0)); layout
// [t][h][n*w][c]. One thread per output pixel; reads are coalesced along w, writes along (w, c).
__global__ void frames_to_u8_kernel(const float* __restrict__ video, uint8_t* __restrict__ out, int N, int C, int T, int H,
                                    int W) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)T * H * N * W;
    if (idx >= total) return;
    const int gx = (int)(idx % ((int64_t)N * W));
    const int64_t r = idx / ((int64_t)N * W);
    const int y = (int)(r % H), t = (int)(r / H);
    const int n = gx / W, x = gx - n * W;
    for (int c = 0; c < C; ++c) {
        float v = video[((((size_t)n * C + c) * T + t) * H + y) * W + x];
        v = fminf(fmaxf(v, -1.0f), 1.0f);
        v = (v + 1.0f) / 2.0f;
        v = v * 255.0f;
        out[(size_t)idx * C + c] = (uint8_t)v;                   // truncation, as Tensor.to(torch.uint8)
    }
}

}  // namespace

extern "C" int dc_gemv_small(const float* in, int ld_in, const uint16_t* W, const float* bias, float* out, int ld_out,
                             int M, int N, int K, int act_in, int act_out, int accumulate, void* stream_) {
    if (!in || !W || !out) return DC_ERR_ARG;
    if (M < 1 || M > 8 || N < 1 || K % 8 != 0) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(gemv_small_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream_, in, ld_in, W, bias, out,
                       ld_out, M, N, K, act_in, act_out, accumulate);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_timestep_embedding(const int64_t* t_table, const int32_t* t_index, int t_stride, float* out, int B,
                                     int dim, float max_period, void* stream_) {
    if (!t_table || !out) return DC_ERR_ARG;
    if (B < 1 || dim < 2) return DC_ERR_SHAPE;
    const int total = B * (dim / 2);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, t_table,
                       t_index, t_stride, out, B, dim, logf(max_period));
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_pack_latent(const float* x, const float* cc, uint16_t* out, int B, int Cx, int Cc, int T, int HW,
                              int c_pad, int nrep, void* stream_) {
    if (!x || !out || (Cc > 0 && !cc)) return DC_ERR_ARG;
    if (c_pad % 8 != 0 || c_pad < Cx + Cc || nrep < 1) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)B * T * HW;
    hipLaunchKernelGGL(pack_latent_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x,
                       cc, out, B, Cx, Cc, T * HW, c_pad, nrep);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_nchw_to_rows(const float* x, uint16_t* out, int N, int C, int HW, int c_pad, float scale,
                               void* stream_) {
    if (!x || !out) return DC_ERR_ARG;
    if (c_pad % 8 != 0 || c_pad < C) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(nchw_to_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, x,
                       out, N, C, HW, c_pad, scale);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_rows_to_nchw(const void* rows, int ld, int rows_f32, float* y, int N, int C, int HW, float scale,
                               void* stream_) {
    if (!rows || !y) return DC_ERR_ARG;
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(rows_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       rows, ld, rows_f32, y, N, C, HW, scale);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_copy2d(const uint16_t* src, int lds_, uint16_t* dst, int ldd, int rows, int cols, void* stream_) {
    if (!src || !dst) return DC_ERR_ARG;
    if (cols % 8 || lds_ % 8 || ldd % 8 || rows < 1) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)rows * (cols / 8);
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream_, src, lds_, dst,
                       ldd, rows, cols / 8);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_im2col3x3_c8(const uint16_t* x, int ldx, uint16_t* out, int ldo, int n_img, int H, int W, void* stream_) {
    if (!x || !out) return DC_ERR_ARG;
    if (ldx % 8 || ldo % 8 || ldo < 128 || n_img < 1 || H < 1 || W < 1 || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)n_img * H * W * 16;
    hipLaunchKernelGGL(im2col3x3_c8_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, (hipStream_t)stream_, x, ldx, out, ldo,
                       n_img, H, W);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_add_rows(const uint16_t* a, int lda, const uint16_t* b, int ldb, uint16_t* y, int ldy, int rows,
                           int cols, void* stream_) {
    if (!a || !b || !y) return DC_ERR_ARG;
    if (cols % 8 || lda % 8 || ldb % 8 || ldy % 8 || rows < 1) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)rows * (cols / 8);
    hipLaunchKernelGGL(add_rows_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream_, a, lda, b,
                       ldb, y, ldy, rows, cols / 8);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_build_context(const float* ctx, uint16_t* out, int B, int T, int n_text, int L, int D, void* stream_) {
    if (!ctx || !out) return DC_ERR_ARG;
    if (D % 8 || B < 1 || T < 1 || n_text < 0 || L < 0) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)B * T * (n_text + L) * (D / 8);
    hipLaunchKernelGGL(build_context_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream_, ctx, out,
                       B, T, n_text, L, D);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_softmax_rows(const float* x, int ldx, uint16_t* y, int ldy, int rows, int cols, void* stream_) {
    if (!x || !y) return DC_ERR_ARG;
    if (rows < 1 || cols < 1) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream_, x, ldx, y, ldy, cols);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_vae_sample(const uint16_t* moments, int ld, const float* noise, float* z, int N, int zc, int HW,
                             float scale, void* stream_) {
    if (!moments || !z) return DC_ERR_ARG;
    const int64_t total = (int64_t)N * HW;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_,
                       moments, ld, noise, z, N, zc, HW, scale);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_ddim_step(const DcDdimParams* pp, const float* e_cond, const float* e_uncond, const float* e_img,
                            int ld_e, const float* x, const float* noise, float* x_prev, float* pred_x0, int B, int C,
                            int THW, float* workspace, void* stream_) {
    if (!pp || !e_cond || !x || !x_prev || !pred_x0 || !workspace) return DC_ERR_ARG;
    const DcDdimParams& p = *pp;
    if (!p.a_t || !p.a_prev || !p.sigma_t || !p.sqrt_one_minus_at) return DC_ERR_ARG;
    if (p.v_param && (!p.sqrt_acp_t || !p.sqrt_1macp_t)) return DC_ERR_ARG;
    if (B < 1 || C < 1 || THW < 1) return DC_ERR_SHAPE;
    hipStream_t stream = (hipStream_t)stream_;
    if (p.guidance_rescale > 0.f && e_uncond) {
        hipLaunchKernelGGL(ddim_partial_kernel, dim3(DDIM_BLOCKS, B), dim3(256), 0, stream, p, e_cond, e_uncond, e_img,
                           ld_e, C, THW, workspace);
        DC_CHECK_LAUNCH();
    }
    const int64_t n = (int64_t)C * THW;
    hipLaunchKernelGGL(ddim_apply_kernel, dim3(grid_for(n, 256, 1024), B), dim3(256), 0, stream, p, e_cond, e_uncond,
                       e_img, ld_e, x, noise, x_prev, pred_x0, C, THW, workspace);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_frames_to_u8(const float* video, uint8_t* out, int N, int C, int T, int H, int W, void* stream_) {
    if (!video || !out) return DC_ERR_ARG;
    if (N < 1 || C < 1 || T < 1 || H < 1 || W < 1) return DC_ERR_SHAPE;
    const int64_t total = (int64_t)T * H * N * W;
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, video, out,
                       N, C, T, H, W);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_mask_blend(float* img, const float* x0, const float* mask, const float* qnoise, const float* sqrt_acp_t,
                             const float* sqrt_1macp_t, const int32_t* step_index, int index, int64_t n,
                             int64_t noise_step_stride, int clean, void* stream_) {
    if (!img || !x0 || !mask) return DC_ERR_ARG;
    if (!clean && (!qnoise || !sqrt_acp_t || !sqrt_1macp_t)) return DC_ERR_ARG;
    if (n < 1) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(mask_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, img, x0, mask,
                       qnoise, sqrt_acp_t, sqrt_1macp_t, step_index, index, n, noise_step_stride, clean);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_advance_counter(int32_t* counter, void* stream_) {
    if (!counter) return DC_ERR_ARG;
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, counter);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_transpose(const uint16_t* src, int lds_, uint16_t* dst, int ldd, int rows, int cols, void* stream_) {
    if (!src || !dst) return DC_ERR_ARG;
    if (rows < 1 || cols < 1) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream_,
                       src, lds_, dst, ldd, rows, cols);
    DC_CHECK_LAUNCH();
    return 0;
}
