// Kernels of the conditioning encoders that run once per clip: the frozen OpenCLIP ViT-H/14 text
// and vision towers and the image preprocessing in front of the vision tower. The dense layers of both towers are
// dc_gemm_conv / dc_layernorm calls; what the denoising path has no kernel for lives here:
//   dc_attn_small     multi-head attention for ANY head width (vision tower: 16 heads x 80) with an optional causal
//                     mask (text tower), Lk <= 1024 - reference: open_clip ResidualAttentionBlock -> nn.MultiheadAttention
//                     as driven by lvdm/modules/encoders/condition.py:216-234, 364-368
//   dc_clip_preprocess  kornia.geometry.resize(bicubic, align_corners, antialias) + (x+1)/2 + CLIP mean/std
//                       (condition.py:322-330): separable Gaussian blur (reflect), bicubic (A = -0.75) sampling
//   dc_patchify       non-overlapping p x p patches -> GEMM rows (the ViT conv1 as an implicit GEMM, condition.py:349)
//   dc_embed_tokens   token embedding gather + positional embedding (condition.py:216-217)
// All of it is a few hundred microseconds per clip: plain wave-level code, no MFMA.
#include "dc_common.h"
#include "dcrafter_hip.h"

namespace {

constexpr int AS_ROWS = 32;        // query rows per workgroup (8 per wave)
constexpr int AS_MAXK = 16;        // keys per lane: Lk <= 64 * AS_MAXK

// One workgroup = (batch, head, 32 query rows); K and V of the head are staged once in LDS as bf16 with a row stride
// of d/2 + 1 dwords (odd: lanes reading the same column of 64 different rows hit 64 different banks). One wave per
// query row: scores with keys spread over lanes, softmax by wave reductions, P.V with channels spread over lanes.
__global__ __launch_bounds__(256) void attn_small_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                         const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int ldq,
                                                         int ldk, int ldv, int ldo, int heads, int Lq, int Lk, int d,
                                                         float scale, int causal, int q_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = blockIdx.x % q_tiles;
    const int bh = blockIdx.x / q_tiles;
    const int head = bh % heads, b = bh / heads;
    const int kstride = d / 2 + 1;                        // dwords per staged row
    uint32_t* sk = reinterpret_cast<uint32_t*>(smem);
    uint32_t* sv = sk + (size_t)Lk * kstride;
    float* sp = reinterpret_cast<float*>(sv + (size_t)Lk * kstride);      // [4 waves][Lk] probabilities
    float* sq = sp + 4 * Lk;                                              // [4 waves][d] query row

    const bf16_t* kb = k + (size_t)b * Lk * ldk + head * d;
    const bf16_t* vb = v + (size_t)b * Lk * ldv + head * d;
    const int hd = d / 2;
    for (int idx = tid; idx < Lk * hd; idx += 256) {
        const int j = idx / hd, c = idx - j * hd;
        sk[j * kstride + c] = *reinterpret_cast<const uint32_t*>(kb + (size_t)j * ldk + 2 * c);
        sv[j * kstride + c] = *reinterpret_cast<const uint32_t*>(vb + (size_t)j * ldv + 2 * c);
    }
    __syncthreads();

    float* myp = sp + wave * Lk;
    float* myq = sq + wave * d;
    for (int rr = 0; rr < AS_ROWS / 4; ++rr) {
        const int row = qt * AS_ROWS + rr * 4 + wave;
        if (row >= Lq) break;                              // wave-uniform
        const bf16_t* qrow = q + ((size_t)b * Lq + row) * ldq + head * d;
        for (int c = lane; c < d; c += 64) myq[c] = bf2f(qrow[c]) * scale;     // nn.MultiheadAttention scales q
        __builtin_amdgcn_wave_barrier();
        float s[AS_MAXK];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < AS_MAXK; ++i) {
            const int j = lane + 64 * i;
            s[i] = -INFINITY;
            if (j < Lk && !(causal && j > row)) {
                float acc = 0.f;
                const uint32_t* kr = sk + j * kstride;
                for (int c = 0; c < hd; ++c) {
                    const uint32_t w = kr[c];
                    acc = fmaf(myq[2 * c], __uint_as_float(w << 16), acc);
                    acc = fmaf(myq[2 * c + 1], __uint_as_float(w & 0xffff0000u), acc);
                }
                s[i] = acc;
            }
            mx = fmaxf(mx, s[i]);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < AS_MAXK; ++i) {
            const int j = lane + 64 * i;
            const float p = (s[i] == -INFINITY) ? 0.f : __expf(s[i] - mx);
            if (j < Lk) myp[j] = p;
            sum += p;
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        __builtin_amdgcn_wave_barrier();
        // o[c] = sum_j p_j V[j][c]: lane owns channel pairs lane, lane + 64 (d <= 256)
        const int jmax = causal ? min(Lk, row + 1) : Lk;
        for (int c = lane; c < hd; c += 64) {
            float a0 = 0.f, a1 = 0.f;
            for (int j = 0; j < jmax; ++j) {
                const uint32_t w = sv[j * kstride + c];
                const float p = myp[j];
                a0 = fmaf(p, __uint_as_float(w << 16), a0);
                a1 = fmaf(p, __uint_as_float(w & 0xffff0000u), a1);
            }
            *reinterpret_cast<uint32_t*>(o + ((size_t)b * Lq + row) * ldo + head * d + 2 * c) = pack_bf2(a0 * inv, a1 * inv);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- image preprocessing --------------------------------------------------------------------------------------------
// 1-D Gaussian taps as kornia builds them: x = i - ks/2, w = exp(-x^2 / (2 sigma^2)) normalised to sum 1.
__device__ __forceinline__ float gauss_tap(int i, int ks, float sigma, float norm) {
    const float x = (float)(i - ks / 2);
    return __expf(-x * x / (2.0f * sigma * sigma)) * norm;
}
__device__ __forceinline__ int reflect_idx(int i, int n) {       // torch 'reflect' padding (no edge repeat)
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// separable blur, one axis per launch: axis 0 = along W, 1 = along H. in/out [N*C][H][W] fp32.
__global__ void blur1d_kernel(const float* __restrict__ in, float* __restrict__ out, int planes, int H, int W, int ks,
                              float sigma, int axis) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)planes * H * W) return;
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int64_t base = idx - x - (int64_t)y * W;
    float norm = 0.f;
    for (int i = 0; i < ks; ++i) { const float t = (float)(i - ks / 2); norm += __expf(-t * t / (2.0f * sigma * sigma)); }
    norm = 1.0f / norm;
    float acc = 0.f;
    for (int i = 0; i < ks; ++i) {
        const int off = i - ks / 2;
        const float w = gauss_tap(i, ks, sigma, norm);
        if (axis == 0) acc += w * in[base + (int64_t)y * W + reflect_idx(x + off, W)];
        else acc += w * in[base + (int64_t)reflect_idx(y + off, H) * W + x];
    }
    out[idx] = acc;
}

// torch's bicubic convolution coefficients (A = -0.75), upsample_bicubic2d
__device__ __forceinline__ void cubic_coeffs(float t, float* c) {
    const float A = -0.75f;
    const float x0 = t + 1.0f, x1 = t, x2 = 1.0f - t, x3 = 2.0f - t;
    c[0] = ((A * x0 - 5.0f * A) * x0 + 8.0f * A) * x0 - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * x3 - 5.0f * A) * x3 + 8.0f * A) * x3 - 4.0f * A;
}

// bicubic resize with align_corners = True, then (v + 1) / 2 and per-channel (v - mean) / std. in [N][C][H][W] ->
// out [N][C][OH][OW] fp32.
__global__ void bicubic_norm_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int H, int W,
                                    int OH, int OW, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * C * OH * OW) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int c = (int)((idx / ((int64_t)OW * OH)) % C);
    const int64_t plane = idx / ((int64_t)OW * OH);
    const float sy = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.f;
    const float sx = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.f;
    const float fy = sy * oy, fx = sx * ox;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float cy[4], cx[4];
    cubic_coeffs(fy - iy, cy);
    cubic_coeffs(fx - ix, cx);
    const float* p = in + plane * (int64_t)H * W;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = min(max(iy - 1 + a, 0), H - 1);
        float r = 0.f;
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            const int xx = min(max(ix - 1 + bq, 0), W - 1);
            r += cx[bq] * p[(int64_t)yy * W + xx];
        }
        acc += cy[a] * r;
    }
    acc = (acc + 1.0f) / 2.0f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    out[idx] = (acc - mean) / sd;
}

// image [N][C][H][W] fp32 -> rows [N * (H/p) * (W/p)][kpad] bf16, column = c*p*p + py*p + px (Conv2d weight order)
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ rows, int N, int C, int H, int W,
                                int p, int kpad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int gh = H / p, gw = W / p;
    if (idx >= (int64_t)N * gh * gw * kpad) return;
    const int col = (int)(idx % kpad);
    const int64_t r = idx / kpad;
    float v = 0.f;
    if (col < C * p * p) {
        const int c = col / (p * p), py = (col / p) % p, px = col % p;
        const int gx = (int)(r % gw), gy = (int)((r / gw) % gh);
        const int64_t n = r / ((int64_t)gw * gh);
        v = img[((n * C + c) * H + gy * p + py) * (int64_t)W + gx * p + px];
    }
    rows[idx] = f2bf(v);
}

__global__ void embed_tokens_kernel(const int64_t* __restrict__ tokens, const bf16_t* __restrict__ table,
                                    const bf16_t* __restrict__ pos, bf16_t* __restrict__ out, int B, int L, int D,
                                    int vocab) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * L * D) return;
    const int c = (int)(idx % D);
    const int64_t r = idx / D;
    const int i = (int)(r % L);
    int64_t t = tokens[r];
    t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
    out[idx] = f2bf(bf2f(table[t * D + c]) + bf2f(pos[(int64_t)i * D + c]));
}

}  // namespace

extern "C" int64_t dc_attn_small_lds_bytes(int Lk, int d) {
    return (int64_t)2 * Lk * (d / 2 + 1) * 4 + (int64_t)4 * Lk * 4 + (int64_t)4 * d * 4;
}

extern "C" int dc_attn_small(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, int ldq, int ldk, int ldv,
                             int ldo, int B, int heads, int Lq, int Lk, int d, float scale, int causal, void* stream_) {
    if (!q || !k || !v || !o) return DC_ERR_ARG;
    if (B < 1 || heads < 1 || Lq < 1 || Lk < 1 || d < 2 || d % 2 || d > 256 || Lk > 64 * AS_MAXK) return DC_ERR_SHAPE;
    if ((ldq | ldk | ldv | ldo) % 2) return DC_ERR_SHAPE;
    const int64_t lds = dc_attn_small_lds_bytes(Lk, d);
    if (lds > 160 * 1024) return DC_ERR_SHAPE;
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&attn_small_kernel), 160 * 1024)) return e;
    const int q_tiles = (Lq + AS_ROWS - 1) / AS_ROWS;
    hipLaunchKernelGGL(attn_small_kernel, dim3(B * heads * q_tiles), dim3(256), (size_t)lds, (hipStream_t)stream_, q, k, v, o,
                       ldq, ldk, ldv, ldo, heads, Lq, Lk, d, scale, causal, q_tiles);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_clip_preprocess(const float* img, float* tmp0, float* tmp1, float* out, int N, int C, int H, int W, int OH,
                                  int OW, int antialias, const float* mean3, const float* std3, void* stream_) {
    if (!img || !out || !mean3 || !std3) return DC_ERR_ARG;
    if (N < 1 || C != 3 || H < 2 || W < 2 || OH < 1 || OW < 1) return DC_ERR_SHAPE;
    hipStream_t stream = (hipStream_t)stream_;
    const float* src = img;
    const float fy = (float)H / (float)OH, fx = (float)W / (float)OW;
    if (antialias && fmaxf(fy, fx) > 1.0f) {
        // kornia.geometry.transform.resize: sigma = max((factor - 1) / 2, 0.001), ks = int(max(4 sigma, 3)) made odd,
        // gaussian_blur2d(input, (ks_y, ks_x), (sigma_y, sigma_x)) with reflect borders, separable
        if (!tmp0 || !tmp1) return DC_ERR_ARG;
        const float sgy = fmaxf((fy - 1.0f) / 2.0f, 0.001f), sgx = fmaxf((fx - 1.0f) / 2.0f, 0.001f);
        int ky = (int)fmaxf(2.0f * 2.0f * sgy, 3.0f), kx = (int)fmaxf(2.0f * 2.0f * sgx, 3.0f);
        if (ky % 2 == 0) ++ky;
        if (kx % 2 == 0) ++kx;
        if (ky / 2 >= H || kx / 2 >= W) return DC_ERR_SHAPE;
        const int64_t tot = (int64_t)N * C * H * W;
        const unsigned grid = (unsigned)((tot + 255) / 256);
        hipLaunchKernelGGL(blur1d_kernel, dim3(grid), dim3(256), 0, stream, img, tmp0, N * C, H, W, kx, sgx, 0);
        DC_CHECK_LAUNCH();
        hipLaunchKernelGGL(blur1d_kernel, dim3(grid), dim3(256), 0, stream, (const float*)tmp0, tmp1, N * C, H, W, ky, sgy, 1);
        DC_CHECK_LAUNCH();
        src = tmp1;
    }
    const int64_t tot = (int64_t)N * C * OH * OW;
    hipLaunchKernelGGL(bicubic_norm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, src, out, N, C, H, W, OH,
                       OW, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_patchify(const float* img, uint16_t* rows, int N, int C, int H, int W, int p, int kpad, void* stream_) {
    if (!img || !rows) return DC_ERR_ARG;
    if (p < 1 || H % p || W % p || kpad < C * p * p || kpad % 8) return DC_ERR_SHAPE;
    const int64_t tot = (int64_t)N * (H / p) * (W / p) * kpad;
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, img, rows, N, C,
                       H, W, p, kpad);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_embed_tokens(const int64_t* tokens, const uint16_t* table, const uint16_t* pos, uint16_t* out, int B, int L,
                               int D, int vocab, void* stream_) {
    if (!tokens || !table || !pos || !out) return DC_ERR_ARG;
    if (B < 1 || L < 1 || D < 1 || vocab < 1) return DC_ERR_SHAPE;
    const int64_t tot = (int64_t)B * L * D;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, tokens, table,
                       pos, out, B, L, D, vocab);
    DC_CHECK_LAUNCH();
    return 0;
}
