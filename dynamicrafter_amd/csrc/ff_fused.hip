// Kernels that keep the activation rows of a workgroup in REGISTERS as MFMA operand fragments ("X-stationary") and stream
// only weights through LDS - the UNet's level-0 / level-1 row-wise operator chains, fused:
//   ff_geglu_fused320_kernel   x = x + ff2(geglu(ff1(norm3(x))))  [+ the transformer's proj_out and residual]      dim 320
//   norm_linear_kernel         Linear(LayerNorm(x)) / Linear(GroupNorm(x)) / Linear(x)                         K 320 / 640
//   ln_qkv_tattn320_kernel     LayerNorm + to_q/k/v + attention over the 16 frames of a position               dim 320
//   gn_silu_tconv_kernel       GroupNorm + SiLU + Conv3d (3,1,1) (+ identity) of a TemporalConvBlock            C 320 / 640
// Common technique: volatile LDS fragment loads + empty `asm volatile` anchors pin the LDS prefetch distance and the
// MFMA / VALU interleave (hipcc otherwise sinks every LDS read to just in front of its use); LDS-DMA pieces addressed by a
// scalar base + per-lane offsets computed once; counted `s_waitcnt vmcnt` (gfx950 counts stores too).
//
// ---- Fused GEGLU FeedForward for dim = 320 (the UNet's level-0 transformers):
//
//   out[M, 320] = ( (X W1v^T + b1v) * gelu(X W1g^T + b1g) ) W2^T + b2 + residual        reference: FeedForward /
//   GEGLU, lvdm/modules/attention.py:415-442, called from BasicTransformerBlock._forward :246
//
// without the [M, 1280] intermediate ever leaving the CU. As two GEMMs that intermediate is written and read back once
// per FeedForward: 2 x 755 MB per level-0 block at the 1024 config, 28 GB per denoising step, and the first GEMM's
// 256 x 256 tiles pay a GEGLU epilogue every five K tiles (K = 320).
//
// Structure: ONE wave per SIMD (4 waves per workgroup, 1 workgroup per CU, up to 512 registers per lane). A workgroup
// owns 128 rows; wave w owns rows [32 w, 32 w + 32) of it for the whole FeedForward:
//   * X fragments of its rows stay in registers (20 x bf16x8 = 80 VGPRs), loaded once per tile straight from HBM;
//   * the 1280 intermediate channels are walked in 40 chunks of 32. Per chunk the workgroup streams the 64 rows of W1
//     (32 value + 32 gate rows x 320, two-deep ring) and the 32 columns of W2 (320 x 32, three-deep ring) through LDS by
//     LDS-DMA (60 KB per chunk), one barrier per chunk;
//   * phase 1: Pv, Pg = X W1v^T, X W1g^T   (2 accumulators, 40 MFMAs 32x32x16), hand-interleaved with the GEGLU
//     arithmetic of the previous chunk;
//   * phase 2: out += P W2c^T              (10 accumulators, 20 MFMAs) - P never leaves registers: with the swapped
//     operand order a lane holds, for ITS row, 16 of the chunk's 32 channels, which is exactly a B-operand fragment
//     pair once the k order is agreed on; the host packs W2 with that order inside every 32-channel chunk
//     (position 16 s + 8 h + e  <->  channel 8 (2 s + e / 4) + 4 h + e % 4); the LDS-DMA pieces of the next chunks go
//     out one behind each of its MFMAs;
//   * epilogue: + b2, bf16, + residual, row-major stores through a wave-private LDS patch - or, PROJ, the result as B
//     fragments of the transformer's proj_out.
// Every weight fragment read from LDS feeds one MFMA (1 KB per MFMA = half the 256 B/clk LDS rate at the full MFMA rate)
// and a workgroup streams W1 + W2 (2.4 MB) per 128 rows: with one wave per SIMD issuing the 60 MFMAs, ~260 vector
// instructions and 15 LDS-DMA pieces of a chunk the kernel is instruction-issue-bound at about half the MFMA rate
// (DESIGN.md section 3.2 has the per-phase stamps).
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <stdint.h>
#include <type_traits>

namespace {

constexpr int FD = 320;            // model width (K of ff1, N of ff2)
constexpr int FM = 4 * FD;         // intermediate width 1280
constexpr int FCH = 32;            // intermediate channels per chunk
constexpr int FNCH = FM / FCH;     // 40 chunks
constexpr int FBM = 128;           // rows per workgroup tile
constexpr int W1_STAGE = 64 * FD * 2;          // 40 KB: 5 K tiles of [64 rows][128 B]
constexpr int W2_STAGE = FD * FCH * 2;         // 20 KB: [320 rows][64 B]
constexpr int W2_RING = 3;
constexpr int FF_LDS = 2 * W1_STAGE + W2_RING * W2_STAGE + 4 * 2048 + 2 * FM * 4;    // rings + epilogue patches + b1 (fp32)
static_assert(FF_LDS <= 160 * 1024, "LDS");

typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;
typedef const volatile __attribute__((address_space(3))) bf16x8_t lds_vfrag_t;   // pinned LDS fragment load (see PD)
__device__ __attribute__((aligned(16))) uint32_t g_zero_ff[8];
#ifdef DC_FF_STAMPS          // tool build only (tools/ff_stamps.py): shader-clock totals of the four parts of a chunk iteration
__device__ unsigned long long g_ff_stamps[8];
#define FF_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long t__ = __builtin_readcyclecounter(); \
                         st_acc[i] += t__ - st_last; st_last = t__; } while (0)
#else
#define FF_STAMP(i) do { } while (0)
#endif

#ifdef DC_FF_GELU_ERF
#define FF_GELU(x) gelu_erf_f(x)
#else
#define FF_GELU(x) gelu_phi_f(x)      // the last chunk's stand-alone GEGLU: the same function as the interleaved stream
#endif
__device__ __forceinline__ int off128(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int off64(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ void glds16f(const void* gsrc, unsigned lds_dst_uniform) {
    asm volatile(
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, off"
        :
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}


// LayerNorm of the rows held as X fragments (lanes (fr, 0) and (fr, 1) hold the two halves of row fr), two-pass in
// registers (BasicTransformerBlock norm1/2/3, lvdm/modules/attention.py:225-227, eps 1e-5); the result is rounded to bf16
// exactly where the stand-alone LayerNorm kernel rounds its output.
template <int KD>
__device__ __forceinline__ void ln_rows_inplace(bf16x8_t (&xf)[KD / 16], const float* ln_g, const float* ln_b, float eps, int fh) {
    float sum = 0.f;
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) {
        const u32x4_t w = __builtin_bit_cast(u32x4_t, xf[kk]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.0f / KD);
    // (opaque re-definitions between the passes: otherwise the 160 fp32 conversions of pass one are kept alive for passes
    // two and three - 160 more registers than the kernels have)
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) asm volatile("" : "+v"(xf[kk]));
    float q = 0.f;
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) {
        const u32x4_t w = __builtin_bit_cast(u32x4_t, xf[kk]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d0 = __uint_as_float(w[e] << 16) - mean, d1 = __uint_as_float(w[e] & 0xffff0000u) - mean;
            q += d0 * d0 + d1 * d1;
        }
    }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / KD) + eps);
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) asm volatile("" : "+v"(xf[kk]));
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) {
        // gamma / beta in groups of 5 k steps (80 floats in flight): hoisted all at once they are 320 registers
        if (kk % 5 == 0) asm volatile("" ::: "memory");
        const u32x4_t w = __builtin_bit_cast(u32x4_t, xf[kk]);
        const int c0 = kk * 16 + fh * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(ln_g + c0), g1 = *reinterpret_cast<const float4*>(ln_g + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ln_b + c0), b1 = *reinterpret_cast<const float4*>(ln_b + c0 + 4);
        u32x4_t o;
        o[0] = pack_bf2((__uint_as_float(w[0] << 16) - mean) * rstd * g0.x + b0.x, (__uint_as_float(w[0] & 0xffff0000u) - mean) * rstd * g0.y + b0.y);
        o[1] = pack_bf2((__uint_as_float(w[1] << 16) - mean) * rstd * g0.z + b0.z, (__uint_as_float(w[1] & 0xffff0000u) - mean) * rstd * g0.w + b0.w);
        o[2] = pack_bf2((__uint_as_float(w[2] << 16) - mean) * rstd * g1.x + b1.x, (__uint_as_float(w[2] & 0xffff0000u) - mean) * rstd * g1.y + b1.y);
        o[3] = pack_bf2((__uint_as_float(w[3] << 16) - mean) * rstd * g1.z + b1.z, (__uint_as_float(w[3] & 0xffff0000u) - mean) * rstd * g1.w + b1.w);
        xf[kk] = __builtin_bit_cast(bf16x8_t, o);
        asm volatile("" : "+v"(xf[kk]));          // ... and the arithmetic of a group is done before the next group loads
    }
}

// x * a[c] + b[c] on the rows held as X fragments (GroupNorm with the statistics already known: a = gamma rstd,
// b = beta - mean a, the arithmetic of gn_apply_kernel in norms.hip), rounded to bf16 like that kernel's output
template <int KD, bool SILU = false>
__device__ __forceinline__ void affine_rows_inplace(bf16x8_t (&xf)[KD / 16], const float* a, const float* b, int fh) {
#pragma unroll
    for (int kk = 0; kk < KD / 16; ++kk) {
        if (kk % 5 == 0) asm volatile("" ::: "memory");
        const u32x4_t w = __builtin_bit_cast(u32x4_t, xf[kk]);
        const int c0 = kk * 16 + fh * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(a + c0), g1 = *reinterpret_cast<const float4*>(a + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(b + c0), b1 = *reinterpret_cast<const float4*>(b + c0 + 4);
        float v[8] = {__uint_as_float(w[0] << 16) * g0.x + b0.x, __uint_as_float(w[0] & 0xffff0000u) * g0.y + b0.y,
                      __uint_as_float(w[1] << 16) * g0.z + b0.z, __uint_as_float(w[1] & 0xffff0000u) * g0.w + b0.w,
                      __uint_as_float(w[2] << 16) * g1.x + b1.x, __uint_as_float(w[2] & 0xffff0000u) * g1.y + b1.y,
                      __uint_as_float(w[3] << 16) * g1.z + b1.z, __uint_as_float(w[3] & 0xffff0000u) * g1.w + b1.w};
        if constexpr (SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        xf[kk] = __builtin_bit_cast(bf16x8_t, o);
        asm volatile("" : "+v"(xf[kk]));
    }
}

struct FfParams {
    const bf16_t* X; int ldx;
    const bf16_t* W1;            // [>= 2560][320]: rows 0..1279 value, 1280..2559 gate (torch ff.net.0.proj.weight order)
    const float* b1;             // [2560]
    const bf16_t* W2p;           // [>= 320][1280], k permuted inside every 32-chunk
    const float* b2;             // [320]
    const bf16_t* R; int ldr;    // residual or nullptr
    bf16_t* O; int ldo;
    int M;
    const float* ln_g;           // optional LayerNorm in front (norm3 of BasicTransformerBlock): X is normalised in registers
    const float* ln_b;
    float ln_eps;
    // optional Linear behind (the transformer's proj_out): y = R2 + (FeedForward result) Wp^T + bp; the FeedForward result is
    // then NOT stored. Wp: [>= 320][320], k permuted inside every 32-chunk like W2p.
    const bf16_t* Wp; const float* bp; const bf16_t* R2; int ldr2; bf16_t* O2; int ldo2;
};

template <bool LN, bool PROJ>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void ff_geglu_fused320_kernel(const FfParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    char* const w1s = smem;                              // 2 x W1_STAGE
    char* const w2s = smem + 2 * W1_STAGE;               // W2_RING x W2_STAGE
    char* const ebuf = smem + 2 * W1_STAGE + W2_RING * W2_STAGE + wave * 2048;

    const int tiles = (p.M + FBM - 1) / FBM;
    // ff1 bias (2560 floats) staged once: a global load per chunk sat ~500 cycles in front of every GEGLU
    float* const b1s = reinterpret_cast<float*>(smem + 2 * W1_STAGE + W2_RING * W2_STAGE + 4 * 2048);
    for (int i = tid; i < 2 * FM; i += 256) b1s[i] = p.b1[i];
    __syncthreads();

    // LDS-DMA of chunk j: W1 rows (10 pieces of 8 rows x 128 B per wave) into stage j & 1 of a two-deep ring, W2 columns
    // (5 pieces of 16 rows x 64 B per wave) into stage j % 3 of a three-deep ring. One piece = one
    // global_load_lds_dwordx4 = 1 KB. Addressing: a 64-bit SCALAR base per ring and chunk + a per-lane 32-bit offset
    // per piece, computed once per kernel (15 VGPRs). A piece costs the issuing wave ~60 cycles whatever surrounds it,
    // so the main loop places the pieces one by one behind the MFMAs of phase 2 (the matrix pipe works them off
    // meanwhile) instead of in a block at the top of the iteration (1150 cycles per chunk, a quarter of the kernel).
    unsigned vo1[10], vo2[5];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int u = wave * 10 + i;                     // (K tile t, 8-row group g): LDS offset u * 1024
        const int t = u >> 3, g = u & 7;                 // rows g*8 .. +7 of the 64-row chunk: g < 4 value, else gate
        const int row = (g < 4 ? g * 8 : FM + (g - 4) * 8) + (lane >> 3);
        vo1[i] = (unsigned)(row * (FD * 2) + t * 128 + (((lane & 7) ^ ((g * 4 + (lane >> 4)) & 7)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 5; ++i)                          // 16-row group wave*5 + i of the 320 output channels
        vo2[i] = (unsigned)(((wave * 5 + i) * 16 + (lane >> 2)) * (FM * 2) + (((lane & 3) ^ ((lane >> 4) & 3)) << 4));
#pragma unroll
    for (int i = 0; i < 10; ++i) asm volatile("" : "+v"(vo1[i]));    // opaque: or hipcc re-derives the wave-uniform parts
#pragma unroll                                                       // in the main loop (5 scalar instructions per piece)
    for (int i = 0; i < 5; ++i) asm volatile("" : "+v"(vo2[i]));
    auto dma_piece = [&](unsigned lds_dst, unsigned voff, uint64_t sbase) __attribute__((always_inline)) {
        // (m0 is NOT saved and restored around a piece: nothing else in these kernels uses it - checked in the ISA, as for gemm_pipe16.h - and
        //  two scalar moves per piece are 8-10 issue clocks of a one-wave-per-SIMD stream)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2"
            :
            : "s"(lds_dst), "v"(voff), "s"(sbase)
            : "memory");
    };
    auto w1_base = [&](int j) { return (uint64_t)(uintptr_t)p.W1 + (uint64_t)j * (FCH * FD * 2); };
    auto w2_base = [&](int j) { return (uint64_t)(uintptr_t)p.W2p + (uint64_t)j * (FCH * 2); };
    auto w1_dst = [&](int j, int i) { return lds_base + (j & 1) * W1_STAGE + (wave * 10 + i) * 1024; };
    auto w2_dst = [&](int stage, int i) { return lds_base + 2 * W1_STAGE + stage * W2_STAGE + (wave * 5 + i) * 1024; };
    auto issue_w1 = [&](int j) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 10; ++i) dma_piece(w1_dst(j, i), vo1[i], w1_base(j));
    };
    auto issue_w2 = [&](int j, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_piece(w2_dst(stage, i), vo2[i], w2_base(j));
    };

#ifdef DC_FF_STAMPS
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last = 0;
#endif
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int m0 = tile * FBM + wave * 32;
        // ---- X fragments of this wave's 32 rows (B operand: lane (row fr, half fh) holds k = 16 kk + 8 fh .. + 7)
        bf16x8_t xf[FD / 16];
        {
            int mr = m0 + fr;
            if (mr >= p.M) mr = p.M - 1;
            const bf16_t* xr = p.X + (size_t)mr * p.ldx + fh * 8;
#pragma unroll
            for (int kk = 0; kk < FD / 16; ++kk) xf[kk] = *reinterpret_cast<const bf16x8_t*>(xr + kk * 16);
            if constexpr (LN) ln_rows_inplace<FD>(xf, p.ln_g, p.ln_b, p.ln_eps, fh);
        }
        f32x16_t acc[FD / 32];
#pragma unroll
        for (int nb = 0; nb < FD / 32; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

        // LDS fragments are read PD k steps (phase 1: two fragments each) / PD2 fragments (phase 2) ahead of the MFMA that
        // consumes them - one wave per SIMD: nothing else hides the LDS latency. The loads are VOLATILE and every
        // consumer passes through an empty volatile asm: that pins the distance. With plain loads hipcc sank each
        // read to just in front of its MFMA to save registers (window of one, `s_waitcnt lgkmcnt(0)` per MFMA pair),
        // and both MFMA phases ran at 64 cycles per MFMA instead of 32.
        constexpr int PD = 4, PD2 = 6;
        // phase 1 of chunk c: value / gate pre-activations of its 32 channels for this wave's 32 rows
        auto phase1 = [&](int c, f32x16_t& av, f32x16_t& ag) __attribute__((always_inline)) {
            const char* s1 = w1s + (c & 1) * W1_STAGE;
#pragma unroll
            for (int r = 0; r < 16; ++r) { av[r] = 0.f; ag[r] = 0.f; }
            bf16x8_t wv[PD], wg[PD];
            auto rd1 = [&](int kk, int slot) __attribute__((always_inline)) {
                const char* kt = s1 + (kk >> 2) * 8192;
                wv[slot] = *(lds_vfrag_t*)((lds_char_t*)kt + off128(fr, (kk & 3) * 2 + fh));
                wg[slot] = *(lds_vfrag_t*)((lds_char_t*)kt + off128(32 + fr, (kk & 3) * 2 + fh));
            };
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) rd1(kk, kk);
#pragma unroll
            for (int kk = 0; kk < FD / 16; ++kk) {
                bf16x8_t cv = wv[kk % PD], cg = wg[kk % PD];
                if (kk + PD < FD / 16) rd1(kk + PD, kk % PD);
                asm volatile("" : "+v"(cv), "+v"(cg));
                av = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cv, xf[kk], av, 0, 0, 0);
                ag = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cg, xf[kk], ag, 0, 0, 0);
            }
        };
        // GEGLU of chunk c on its accumulators -> the two B-operand fragments of phase 2 (lane (row fr, fh) holds
        // channels 8 q + 4 fh + i of the chunk)
        auto geglu = [&](int c, const f32x16_t& av, const f32x16_t& ag, bf16x8_t (&pf)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4_t pw;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int q = 2 * s + h2;
                    const int n = c * FCH + 8 * q + 4 * fh;
                    const float4 bv = *reinterpret_cast<const float4*>(b1s + n);
                    const float4 bg = *reinterpret_cast<const float4*>(b1s + FM + n);
                    const float v0 = (av[4 * q + 0] + bv.x) * FF_GELU(ag[4 * q + 0] + bg.x);
                    const float v1 = (av[4 * q + 1] + bv.y) * FF_GELU(ag[4 * q + 1] + bg.y);
                    const float v2 = (av[4 * q + 2] + bv.z) * FF_GELU(ag[4 * q + 2] + bg.z);
                    const float v3 = (av[4 * q + 3] + bv.w) * FF_GELU(ag[4 * q + 3] + bg.w);
                    pw[2 * h2] = pack_bf2(v0, v1);
                    pw[2 * h2 + 1] = pack_bf2(v2, v3);
                }
                pf[s] = __builtin_bit_cast(bf16x8_t, pw);
            }
        };
        // phase 2 of chunk c (W2 ring stage st): out[row][ch] += P W2c^T (W2 packed in the matching k order). DW1 / DW2:
        // the wave's LDS-DMA pieces of W1(jd + 2) / W2(jd + 1) go out one behind each of the first MFMAs.
        const int a2_0 = off64(fr, fh), a2_1 = off64(fr, 2 + fh);
        auto phase2 = [&](int st, const bf16x8_t (&pf)[2], auto DW1, auto DW2, int jd, int st_next) __attribute__((always_inline)) {
            const char* s2 = w2s + st * W2_STAGE;
            bf16x8_t w2r[PD2];
            auto rd2 = [&](int idx, int slot) __attribute__((always_inline)) {      // idx = s * 10 + nb
                const int s = idx / (FD / 32), nb = idx % (FD / 32);
                // off64(nb * 32 + fr, s * 2 + fh) = nb * 2048 + (fr * 64 + swizzled chunk): nb only moves an immediate offset,
                // two lane-dependent bases cover all 20 fragments (spelled out: hipcc kept 20 address registers otherwise)
                w2r[slot] = *(lds_vfrag_t*)((lds_char_t*)s2 + (s ? a2_1 : a2_0) + nb * 2048);
            };
#pragma unroll
            for (int i = 0; i < PD2; ++i) rd2(i, i);
            const uint64_t b1n = w1_base(jd + 2), b2n = w2_base(jd + 1);
#pragma unroll
            for (int idx = 0; idx < 2 * (FD / 32); ++idx) {
                const int s = idx / (FD / 32), nb = idx % (FD / 32);
                bf16x8_t cw = w2r[idx % PD2];
                if (idx + PD2 < 2 * (FD / 32)) rd2(idx + PD2, idx % PD2);
                asm volatile("" : "+v"(cw));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cw, pf[s], acc[nb], 0, 0, 0);
                if (DW1.value && idx < 10) dma_piece(w1_dst(jd + 2, idx), vo1[idx], b1n);
                if (DW2.value && idx >= 10 && idx < 15) dma_piece(w2_dst(st_next, idx - 10), vo2[idx - 10], b2n);
            }
        };
        // phase 1 of chunk c + 1 and the GEGLU of chunk c in ONE hand-interleaved instruction stream. One wave per SIMD
        // means nothing else overlaps the matrix pipe with the vector ALU, and left to itself hipcc clusters the 40 MFMAs
        // and the GEGLU arithmetic into separate runs (2400 of 6900 cycles per chunk were GEGLU with an idle matrix pipe;
        // a sched_group_barrier pipeline over the block is silently dropped, sched_barrier does not order the MFMA
        // builtins during instruction selection). So the order is pinned with empty volatile asm statements that
        // "redefine" a value: they are ordered among themselves, the arithmetic in between hangs on them by data
        // dependence. Per k step: MFMA value | half of the GEGLU of one channel PAIR (packed fp32 math) | MFMA gate.
        // Even steps run the erf polynomial of pair kk / 2, odd steps its exp, the product and the bf16 pack.
        auto phase1_geglu = [&](int c, const f32x16_t& cv, const f32x16_t& cg, f32x16_t& nv, f32x16_t& ng, bf16x8_t (&pf)[2])
            __attribute__((always_inline)) {
            const char* s1 = w1s + ((c + 1) & 1) * W1_STAGE;
#pragma unroll
            for (int r = 0; r < 16; ++r) { nv[r] = 0.f; ng[r] = 0.f; }
            bf16x8_t wv[PD], wg[PD];
            auto rd1 = [&](int kk, int slot) __attribute__((always_inline)) {
                const char* kt = s1 + (kk >> 2) * 8192;
                wv[slot] = *(lds_vfrag_t*)((lds_char_t*)kt + off128(fr, (kk & 3) * 2 + fh));
                wg[slot] = *(lds_vfrag_t*)((lds_char_t*)kt + off128(32 + fr, (kk & 3) * 2 + fh));
            };
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) rd1(kk, kk);
            // ff1 bias of the 4 channels (quad q = kk / 4) in flight: the gate bias is last used at k step 4 q + 2, the value
            // bias at 4 q + 3; each is reloaded right there for quad q + 1, two k steps ahead of its first use (LDS latency)
            float4 bv = *reinterpret_cast<const float4*>(b1s + c * FCH + 4 * fh);
            float4 bg = *reinterpret_cast<const float4*>(b1s + FM + c * FCH + 4 * fh);
            float xg0, xg1, ww0, ww1;
#ifndef DC_FF_GELU_ERF
            float xc0, xc1, qq0, qq1;
#endif
            u32x4_t pw[2];
#pragma unroll
            for (int kk = 0; kk < FD / 16; ++kk) {
                bf16x8_t fv = wv[kk % PD], fg = wg[kk % PD];
                if (kk + PD < FD / 16) rd1(kk + PD, kk % PD);
                asm volatile("" : "+v"(fv), "+v"(fg));
                const int m = kk >> 1;                    // channel pair (values 2m, 2m + 1 of the accumulators)
                nv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fv, xf[kk], nv, 0, 0, 0);
                asm volatile("" : "+a"(nv));
                // gelu(x) = x/2 (1 + erf(x/sqrt2)) with erf by Abramowitz-Stegun 7.1.26 (as erf_as_f), regrouped to
                // max(x, 0) - w e:  t = 1 / (1 + 0.3275911 |x| / sqrt2),  w = |x|/2 (a1 t + .. + a5 t^5),  e = exp(-x^2/2)
                // - 21 vector instructions per channel pair instead of 27 (this stream is issue-bound)
                // (scalar f32 instructions on purpose: beside MFMAs a v_pk_fma_f32 costs ~22 cycles more than the two
                // v_fma_f32 it replaces - MI355X_MICROARCH 'price of one filler' - and this stream is issue-bound)
#ifdef DC_FF_GELU_ERF     // tool build (same-box A/B): the erf form of round 2 (Abramowitz-Stegun 7.1.26: v_rcp_f32 + v_exp_f32, 8 issue clocks each)
                if (kk < 16 && !(kk & 1)) {               // polynomial half
                    float g0 = cg[2 * m], g1 = cg[2 * m + 1];
                    asm volatile("" : "+v"(g0), "+v"(g1));
                    xg0 = g0 + ((m & 1) ? bg.z : bg.x);
                    xg1 = g1 + ((m & 1) ? bg.w : bg.y);
                    const float t0 = __builtin_amdgcn_rcpf(fmaf(fabsf(xg0), 0.3275911f * 0.70710678118654752f, 1.0f));
                    const float t1 = __builtin_amdgcn_rcpf(fmaf(fabsf(xg1), 0.3275911f * 0.70710678118654752f, 1.0f));
                    float q0 = fmaf(t0, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
                    float q1 = fmaf(t1, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
                    q0 = fmaf(q0, t0, 0.5f * 1.421413741f);  q1 = fmaf(q1, t1, 0.5f * 1.421413741f);
                    q0 = fmaf(q0, t0, 0.5f * -0.284496736f); q1 = fmaf(q1, t1, 0.5f * -0.284496736f);
                    q0 = fmaf(q0, t0, 0.5f * 0.254829592f);  q1 = fmaf(q1, t1, 0.5f * 0.254829592f);
                    ww0 = (q0 * t0) * fabsf(xg0);
                    ww1 = (q1 * t1) * fabsf(xg1);
                    asm volatile("" : "+v"(ww0), "+v"(ww1), "+v"(xg0), "+v"(xg1));
                    if ((kk & 3) == 2 && kk < 12)         // accumulator values 4 q .. 4 q + 3 are channels 8 q + 4 fh + i
                        bg = *reinterpret_cast<const float4*>(b1s + FM + c * FCH + 8 * ((kk >> 2) + 1) + 4 * fh);
                } else if (kk < 16) {                     // exp half, GEGLU product, bf16 pack
                    float v0 = cv[2 * m], v1 = cv[2 * m + 1];
                    asm volatile("" : "+v"(v0), "+v"(v1));
                    v0 += (m & 1) ? bv.z : bv.x;
                    v1 += (m & 1) ? bv.w : bv.y;
                    const float e0 = __builtin_amdgcn_exp2f((xg0 * (-0.5f * 1.4426950408889634f)) * xg0);
                    const float e1 = __builtin_amdgcn_exp2f((xg1 * (-0.5f * 1.4426950408889634f)) * xg1);
                    float r0, r1;                         // fmaxf() costs a canonicalising v_max_f32 x, x in front
                    asm("v_max_f32 %0, 0, %1" : "=v"(r0) : "v"(xg0));
                    asm("v_max_f32 %0, 0, %1" : "=v"(r1) : "v"(xg1));
                    const float o0 = v0 * fmaf(-ww0, e0, r0);
                    const float o1 = v1 * fmaf(-ww1, e1, r1);
                    unsigned w = pack_bf2(o0, o1);
                    asm volatile("" : "+v"(w));
                    pw[m >> 2][m & 3] = w;
                    if ((kk & 3) == 3 && kk < 12)
                        bv = *reinterpret_cast<const float4*>(b1s + c * FCH + 8 * ((kk >> 2) + 1) + 4 * fh);
                }
#else
                // gelu(x) = x Phi(x) with the clamped polynomial of the GEMM epilogues (phi_poly_f, dc_common.h: |error| <= 1.9e-4, below
                // the bf16 rounding of P): per element 13.5 full-rate instructions (54 issue clocks) instead of 15.5 + v_rcp_f32 +
                // v_exp_f32 (78) - this stream is issue-bound. Even steps: clamp, square, the first three Horner steps of the pair;
                // odd steps: the other three, Phi, the GEGLU product and the bf16 pack.
                if (kk < 16 && !(kk & 1)) {
                    float g0 = cg[2 * m], g1 = cg[2 * m + 1];
                    asm volatile("" : "+v"(g0), "+v"(g1));
                    xg0 = g0 + ((m & 1) ? bg.z : bg.x);
                    xg1 = g1 + ((m & 1) ? bg.w : bg.y);
                    xc0 = __builtin_amdgcn_fmed3f(xg0, -4.0f, 4.0f);
                    xc1 = __builtin_amdgcn_fmed3f(xg1, -4.0f, 4.0f);
                    ww0 = xc0 * xc0;
                    ww1 = xc1 * xc1;
                    float q0 = fmaf(2.258878772920525e-08f, ww0, -1.5888579127931735e-06f);
                    float q1 = fmaf(2.258878772920525e-08f, ww1, -1.5888579127931735e-06f);
                    q0 = fmaf(q0, ww0, 4.776452260557562e-05f);   q1 = fmaf(q1, ww1, 4.776452260557562e-05f);
                    q0 = fmaf(q0, ww0, -0.0008121939026750624f);  q1 = fmaf(q1, ww1, -0.0008121939026750624f);
                    qq0 = q0; qq1 = q1;
                    asm volatile("" : "+v"(qq0), "+v"(qq1), "+v"(ww0), "+v"(ww1), "+v"(xc0), "+v"(xc1), "+v"(xg0), "+v"(xg1));
                    if ((kk & 3) == 2 && kk < 12)         // accumulator values 4 q .. 4 q + 3 are channels 8 q + 4 fh + i
                        bg = *reinterpret_cast<const float4*>(b1s + FM + c * FCH + 8 * ((kk >> 2) + 1) + 4 * fh);
                } else if (kk < 16) {
                    float v0 = cv[2 * m], v1 = cv[2 * m + 1];
                    asm volatile("" : "+v"(v0), "+v"(v1));
                    v0 += (m & 1) ? bv.z : bv.x;
                    v1 += (m & 1) ? bv.w : bv.y;
                    float q0 = fmaf(qq0, ww0, 0.008763724006712437f), q1 = fmaf(qq1, ww1, 0.008763724006712437f);
                    q0 = fmaf(q0, ww0, -0.06455449014902115f);  q1 = fmaf(q1, ww1, -0.06455449014902115f);
                    q0 = fmaf(q0, ww0, 0.3978703022003174f);    q1 = fmaf(q1, ww1, 0.3978703022003174f);
                    const float o0 = (xg0 * fmaf(xc0, q0, 0.5f)) * v0;
                    const float o1 = (xg1 * fmaf(xc1, q1, 0.5f)) * v1;
                    unsigned w = pack_bf2(o0, o1);
                    asm volatile("" : "+v"(w));
                    pw[m >> 2][m & 3] = w;
                    if ((kk & 3) == 3 && kk < 12)
                        bv = *reinterpret_cast<const float4*>(b1s + c * FCH + 8 * ((kk >> 2) + 1) + 4 * fh);
                }
#endif
                ng = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg, xf[kk], ng, 0, 0, 0);
                asm volatile("" : "+a"(ng));
            }
            pf[0] = __builtin_bit_cast(bf16x8_t, pw[0]);
            pf[1] = __builtin_bit_cast(bf16x8_t, pw[1]);
        };
        // Software pipeline over the 40 chunks. Iteration j: [wait W1(j+1), W2(j); barrier] phase 2 of chunk j-1 with the
        // LDS-DMA pieces of W1(j+2) and W2(j+1) behind its MFMAs (W1 stage j&1 held W1(j), read in iteration j-1; W2 stage
        // (j+1)%3 held W2(j-2), read in iteration j-1: both free once every wave is past this barrier), then phase 1 of
        // chunk j+1 interleaved with the GEGLU of chunk j. The pieces have all of that to land in.
        typedef std::integral_constant<bool, true> yes_t;
        typedef std::integral_constant<bool, false> no_t;
        __builtin_amdgcn_s_barrier();                    // every wave is done with the previous tile's LDS stages
        issue_w1(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // W1(0) (and the X fragments)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_w1(1);
        issue_w2(0, 0);
        f32x16_t a0, a0g, a1, a1g;
        bf16x8_t pf[2];
        phase1(0, a0, a0g);
        auto top = [&]() __attribute__((always_inline)) {
            FF_STAMP(4);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            FF_STAMP(0);
        };
        top();                                           // iteration 0: no phase 2 yet
        issue_w1(2);
        issue_w2(1, 1);
        phase1_geglu(0, a0, a0g, a1, a1g, pf);
        FF_STAMP(2);
        int st = 0;                                      // ring stage of W2(j - 1)
        auto body = [&](int j, const f32x16_t& cv, const f32x16_t& cg, f32x16_t& nv, f32x16_t& ng, auto DW1, auto DW2)
            __attribute__((always_inline)) {
            top();
            const int st2 = st + 2 >= W2_RING ? st + 2 - W2_RING : st + 2;
            phase2(st, pf, DW1, DW2, j, st2);
            st = st + 1 >= W2_RING ? 0 : st + 1;
            FF_STAMP(3);
            phase1_geglu(j, cv, cg, nv, ng, pf);
            FF_STAMP(2);
        };
        for (int j = 1; j < FNCH - 3; j += 2) {          // j = 1 .. 36
            body(j, a1, a1g, a0, a0g, yes_t(), yes_t());
            body(j + 1, a0, a0g, a1, a1g, yes_t(), yes_t());
        }
        body(FNCH - 3, a1, a1g, a0, a0g, yes_t(), yes_t());          // 37: W1(39), W2(38)
        body(FNCH - 2, a0, a0g, a1, a1g, no_t(), yes_t());           // 38: W2(39) only
        top();                                                       // 39: nothing left to overlap the last GEGLU with
        phase2(st, pf, no_t(), no_t(), 0, 0);
        st = st + 1 >= W2_RING ? 0 : st + 1;
        geglu(FNCH - 1, a1, a1g, pf);
        phase2(st, pf, no_t(), no_t(), 0, 0);
        if constexpr (!PROJ) {
            // ---- epilogue: + b2, bf16, + residual; row-major through the wave-private patch (32 rows x 64 B per block)
            {
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int fr_e = lane_e & 31, fh_e = lane_e >> 5;
                const int rrow = lane_e >> 2, rc = lane_e & 3;
#pragma unroll
                for (int nb = 0; nb < FD / 32; ++nb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nb * 32 + 8 * q + 4 * fh_e;
                        const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n);
                        uint2 pk;
                        pk.x = pack_bf2(acc[nb][4 * q] + bv.x, acc[nb][4 * q + 1] + bv.y);
                        pk.y = pack_bf2(acc[nb][4 * q + 2] + bv.z, acc[nb][4 * q + 3] + bv.w);
                        *reinterpret_cast<uint2*>(ebuf + fr_e * 64 + (((2 * q + fh_e) ^ (((fr_e >> 1) & 3) << 1)) << 3)) = pk;
                    }
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int r = t * 16 + rrow;
                        u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4));
                        const int m = m0 + r;
                        if (m >= p.M) continue;
                        const int n = nb * 32 + rc * 8;
                        if (p.R) {
                            const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(p.R + (size_t)m * p.ldr + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                d[e] = pack_bf2(__uint_as_float(d[e] << 16) + __uint_as_float(rr[e] << 16),
                                                __uint_as_float(d[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u));
                        }
                        *reinterpret_cast<u32x4_t*>(p.O + (size_t)m * p.ldo + n) = d;
                    }
                }
            }
        } else {
            // ---- FeedForward result -> B fragments (bf16 where the stand-alone path rounds: acc + b2, then + residual), the
            // transformer's proj_out behind it: y = R2 + h2 Wp^T + bp. A lane holds channels 8 q + 4 fh + i of each 32-block
            // of ITS row, i.e. the two k-step fragments of that block in the k order Wp is packed in (as P feeds W2p).
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int fr_e = lane_e & 31, fh_e = lane_e >> 5;
            unsigned vo3[5];                             // pieces of a 32-row x 640-byte chunk of Wp (norm_linear's layout);
#pragma unroll                                           // derived here from the opaque lane id: five more registers held
            for (int i = 0; i < 5; ++i) {                // across the chunk loop spill
                const int u = wave * 5 + i;
                const int t = u >> 2, g = u & 3;
                vo3[i] = (unsigned)((g * 8 + (lane_e >> 3)) * (FD * 2) + t * 128 + (((lane_e & 7) ^ ((g * 4 + (lane_e >> 4)) & 7)) << 4));
            }
            {
                int mr = m0 + fr_e;
                if (mr >= p.M) mr = p.M - 1;
                const bf16_t* rrow = p.R + (size_t)mr * p.ldr + 4 * fh_e;
#pragma unroll
                for (int nb = 0; nb < FD / 32; ++nb) {
                    u32x4_t fw[2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nb * 32 + 8 * q;
                        const float4 bv = *reinterpret_cast<const float4*>(p.b2 + n + 4 * fh_e);
                        const uint2 rr = *reinterpret_cast<const uint2*>(rrow + n);
                        const unsigned d0 = pack_bf2(acc[nb][4 * q] + bv.x, acc[nb][4 * q + 1] + bv.y);
                        const unsigned d1 = pack_bf2(acc[nb][4 * q + 2] + bv.z, acc[nb][4 * q + 3] + bv.w);
                        fw[q >> 1][2 * (q & 1)] = pack_bf2(__uint_as_float(d0 << 16) + __uint_as_float(rr.x << 16),
                                                          __uint_as_float(d0 & 0xffff0000u) + __uint_as_float(rr.x & 0xffff0000u));
                        fw[q >> 1][2 * (q & 1) + 1] = pack_bf2(__uint_as_float(d1 << 16) + __uint_as_float(rr.y << 16),
                                                              __uint_as_float(d1 & 0xffff0000u) + __uint_as_float(rr.y & 0xffff0000u));
                    }
                    xf[2 * nb] = __builtin_bit_cast(bf16x8_t, fw[0]);
                    xf[2 * nb + 1] = __builtin_bit_cast(bf16x8_t, fw[1]);
                }
            }
            // Wp in chunks of 32 output channels (20 KB: 5 K tiles of [32 rows][128 B]) through two stages of the (idle) W2
            // ring, 5 LDS-DMA pieces per wave, one chunk ahead; counted waits: a chunk's pieces are followed by the previous
            // chunk's residual loads (2, already consumed) and stores (2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // every wave is done reading the W rings of the FeedForward
            asm volatile("" ::: "memory");
            auto wp_base = [&](int c) { return (uint64_t)(uintptr_t)p.Wp + (uint64_t)c * (32 * FD * 2); };
            auto wp_dst = [&](int c, int i) { return lds_base + 2 * W1_STAGE + (c & 1) * W2_STAGE + (wave * 5 + i) * 1024; };
#pragma unroll
            for (int i = 0; i < 5; ++i) dma_piece(wp_dst(0, i), vo3[i], wp_base(0));
            const bool full_tile = tile * FBM + FBM <= p.M;
            for (int c = 0; c < FD / 32; ++c) {
                if (c == 0 || !full_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char* s3 = w2s + (c & 1) * W2_STAGE;
                f32x16_t ya;
#pragma unroll
                for (int r = 0; r < 16; ++r) ya[r] = 0.f;
                constexpr int PD3 = 6;
                bf16x8_t wr[PD3];
                auto rd3 = [&](int kk, int sl) __attribute__((always_inline)) {
                    wr[sl] = *(lds_vfrag_t*)((lds_char_t*)s3 + (kk >> 2) * 4096 + off128(fr_e, (kk & 3) * 2 + fh_e));
                };
#pragma unroll
                for (int kk = 0; kk < PD3; ++kk) rd3(kk, kk);
                const bool more = c + 1 < FD / 32;
                const uint64_t nbase = wp_base(c + 1);
#pragma unroll
                for (int kk = 0; kk < FD / 16; ++kk) {
                    bf16x8_t f = wr[kk % PD3];
                    if (kk + PD3 < FD / 16) rd3(kk + PD3, kk % PD3);
                    asm volatile("" : "+v"(f));
                    ya = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, xf[kk], ya, 0, 0, 0);
                    if (kk < 5 && more) dma_piece(wp_dst(c + 1, kk), vo3[kk], nbase);
                }
                // chunk epilogue: + bp, bf16, + R2, row-major through the wave-private patch (32 rows x 64 B)
                const int n0 = c * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bp + n0 + 8 * q + 4 * fh_e);
                    uint2 pk;
                    pk.x = pack_bf2(ya[4 * q] + bv.x, ya[4 * q + 1] + bv.y);
                    pk.y = pack_bf2(ya[4 * q + 2] + bv.z, ya[4 * q + 3] + bv.w);
                    *reinterpret_cast<uint2*>(ebuf + fr_e * 64 + (((2 * q + fh_e) ^ (((fr_e >> 1) & 3) << 1)) << 3)) = pk;
                }
                const int rrow2 = lane_e >> 2, rc = lane_e & 3;
                u32x4_t rres[2];
                int mrow[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    mrow[t] = m0 + t * 16 + rrow2;
                    const int mc = mrow[t] < p.M ? mrow[t] : p.M - 1;
                    rres[t] = *reinterpret_cast<const u32x4_t*>(p.R2 + (size_t)mc * p.ldr2 + n0 + rc * 8);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = t * 16 + rrow2;
                    u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        d[e] = pack_bf2(__uint_as_float(d[e] << 16) + __uint_as_float(rres[t][e] << 16),
                                        __uint_as_float(d[e] & 0xffff0000u) + __uint_as_float(rres[t][e] & 0xffff0000u));
                    if (mrow[t] < p.M) *reinterpret_cast<u32x4_t*>(p.O2 + (size_t)mrow[t] * p.ldo2 + n0 + rc * 8) = d;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
#ifdef DC_FF_STAMPS
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 5; ++i) atomicAdd(&g_ff_stamps[i], st_acc[i]);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------
// Norm + Linear for dim K = 320 or 640 (level-0 / level-1 transformers): out[M, N] = norm(x) W^T (+ bias), N % 32 == 0,
// norm = LayerNorm over the row, or GroupNorm with known statistics (a per-instance affine map), or none.
//   reference: norm1 -> attn1.to_q/k/v, norm2 -> attn2.to_q (and the temporal blocks' two self-attentions),
//   BasicTransformerBlock._forward lvdm/modules/attention.py:242-245 with CrossAttention.forward :101-105; norm -> proj_in of
//   SpatialTransformer.forward :296-301 / TemporalTransformer.forward :367-377
// As norm kernel + GEMM the normalised copy is written and read back (2 x 189 MB per use at level 0 of the 1024 config) and
// the GEMM's 256 x 320 tiles read every activation row N / 320 times. Here a workgroup owns 128 rows: the (normalised) X
// fragments of a wave's 32 rows stay in registers as in the FeedForward kernel above, the weight streams through a
// two-deep LDS ring in stages of 32 output channels x 320 k (20 KB, 20 MFMAs per wave; K = 640: two stages per chunk),
// each chunk's 32 x 32 block is stored row-major through a wave-private LDS patch. ~50 KB of LDS and 153 (K = 320) / < 256
// (K = 640) registers: three / two workgroups per CU, i.e. waves on a SIMD that are never in step, which hides the X loads
// at the head of a tile and the stores of the epilogue. At K = 320 the kernel is bound by the HBM writes of `out`, not by
// the matrix pipe. A launch with few row tiles splits N over `nsplit` workgroups per tile (level 1: 576 tiles for 512
// slots would leave the second round almost empty).
constexpr int LCH = 32;                        // output channels per chunk
constexpr int LW_STAGE = LCH * FD * 2;         // 20 KB: 5 K tiles of [32 rows][128 B]
constexpr int LL_BIAS = 640;                   // bias entries kept in LDS (N <= 640: every bias-carrying use in the UNet)
// ring + epilogue patches + gamma / beta + bias (K = 320: 54272 B, three workgroups per CU still fit the 160 KB)
constexpr int ll_lds(int kh) { return 2 * LW_STAGE + 4 * 2048 + 2 * FD * kh * 4 + LL_BIAS * 4; }

struct LlParams {
    const bf16_t* X; int ldx;
    const bf16_t* W;             // [>= N][K]
    const float* bias;           // [N] or nullptr
    bf16_t* O; int ldo;
    int M, N;
    int nsplit, cpp;             // workgroups per row tile, chunks per workgroup
    const float* ln_g; const float* ln_b; float ln_eps;      // LayerNorm / GroupNorm gamma, beta
    const float2* gn_stats; int gn_groups, gn_rpi;           // NORM 2: (mean, rstd) [inst][group], rows per instance (% 128 == 0)
    const bf16_t* R; int ldr;                                // RES: out = R + n W^T + bias (R may be O: every element is read and
};                                                           // written by the same lane of the same workgroup)

// NORM 0: none, 1: LayerNorm over the row, 2: GroupNorm with known statistics; K = 320 KH; RES: residual operand
template <int NORM, int KH, bool RES = false>
__global__ __launch_bounds__(256, KH == 1 ? 3 : 2)
void norm_linear_kernel(const LlParams p) {
    constexpr int KD = FD * KH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    char* const ebuf = smem + 2 * LW_STAGE + wave * 2048;
    // the nsplit workgroups of a row tile read the same activation rows: they are 8 block ids apart, i.e. on one XCD at about the
    // same time, so that the second read is an L2 hit (consecutive ids go to different XCDs: counters showed X fetched nsplit times)
    const int bgrp = (int)blockIdx.x / (8 * p.nsplit), brem = (int)blockIdx.x - bgrp * (8 * p.nsplit);
    const int tile = bgrp * 8 + (brem & 7), part = brem >> 3;
    if (tile * FBM >= p.M) return;                                   // padding blocks of the last group of 8 tiles
    const int m0 = tile * FBM + wave * 32;
    const bool full_tile = tile * FBM + FBM <= p.M;                    // ragged last tile: uncounted vmcnt waits
    const int c_begin = part * p.cpp;
    int c_end = c_begin + p.cpp;
    if (c_end > p.N / LCH) c_end = p.N / LCH;
    if (c_begin >= c_end) return;

    // LDS-DMA of stage (chunk c, k half h): 32 weight rows x 640 B = 20 pieces of 8 rows x 128 B, 5 per wave
    unsigned vo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int u = wave * 5 + i;                      // (K tile t, 8-row group g): LDS offset u * 1024
        const int t = u >> 2, g = u & 3;
        vo[i] = (unsigned)((g * 8 + (lane >> 3)) * (KD * 2) + t * 128 + (((lane & 7) ^ ((g * 4 + (lane >> 4)) & 7)) << 4));
        asm volatile("" : "+v"(vo[i]));
    }
    auto dma_piece = [&](unsigned lds_dst, unsigned voff, uint64_t sbase) __attribute__((always_inline)) {
        // (m0 is NOT saved and restored around a piece: nothing else in these kernels uses it - checked in the ISA, as for gemm_pipe16.h - and
        //  two scalar moves per piece are 8-10 issue clocks of a one-wave-per-SIMD stream)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2"
            :
            : "s"(lds_dst), "v"(voff), "s"(sbase)
            : "memory");
    };
    auto w_base = [&](int c, int h) { return (uint64_t)(uintptr_t)p.W + (uint64_t)c * (LCH * KD * 2) + h * (FD * 2); };
    auto w_dst = [&](int slot, int i) { return lds_base + slot * LW_STAGE + (wave * 5 + i) * 1024; };

#pragma unroll
    for (int i = 0; i < 5; ++i) dma_piece(w_dst(0, i), vo[i], w_base(c_begin, 0));

    // ---- X fragments of this wave's 32 rows (B operand: lane (row fr, half fh) holds k = 16 kk + 8 fh .. + 7)
    bf16x8_t xf[KD / 16];
    float* const lns = reinterpret_cast<float*>(smem + 2 * LW_STAGE + 4 * 2048);
    float* const bls = lns + 2 * KD;
    const bool bias_lds = p.bias != nullptr && p.N <= LL_BIAS;
    {
        int mr = m0 + fr;
        if (mr >= p.M) mr = p.M - 1;
        const bf16_t* xr = p.X + (size_t)mr * p.ldx + fh * 8;
#pragma unroll
        for (int kk = 0; kk < KD / 16; ++kk) xf[kk] = *reinterpret_cast<const bf16x8_t*>(xr + kk * 16);
        // the bias in LDS: as global loads in the chunk epilogue they were two exposed round trips per 20 MFMAs
        if (bias_lds) for (int i = tid; i < p.N; i += 256) bls[i] = p.bias[i];
        if constexpr (NORM == 1) {
            for (int i = tid; i < KD; i += 256) { lns[i] = p.ln_g[i]; lns[KD + i] = p.ln_b[i]; }
        } else if constexpr (NORM == 2) {
            const int inst = (int)(((long long)tile * FBM) / p.gn_rpi);           // a tile never straddles two instances
            for (int i = tid; i < KD; i += 256) {
                const float2 st = p.gn_stats[(size_t)inst * p.gn_groups + i / (KD / p.gn_groups)];
                const float a = p.ln_g[i] * st.y;
                lns[i] = a; lns[KD + i] = p.ln_b[i] - st.x * a;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // X, gamma / beta and the first stage: the compiler's own
    __builtin_amdgcn_s_barrier();                                   // vmcnt accounting does not see the asm LDS-DMA, so the
    asm volatile("" ::: "memory");                                  // loads are waited for explicitly
    if constexpr (NORM == 1) ln_rows_inplace<KD>(xf, lns, lns + KD, p.ln_eps, fh);
    if constexpr (NORM == 2) affine_rows_inplace<KD>(xf, lns, lns + KD, fh);

    constexpr int PD = 6;
    int slot = 0;
    for (int c = c_begin; c < c_end; ++c) {
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        u32x4_t rres[2];                              // RES: this lane's two 16-byte pieces of the chunk's residual rows
#pragma unroll
        for (int h = 0; h < KH; ++h) {
            if (c > c_begin || h > 0) {
                // this stage was issued during the previous one, in front of that stage's two stores if it closed a chunk
                if (!full_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (h == 0) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (RES) {
                // the residual rows of this chunk, in the coalesced pattern of the stores (16 rows x 64 B per instruction),
                // requested a chunk's MFMAs ahead of the epilogue that adds them
                if (h == 0) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        int m = m0 + t * 16 + (lane >> 2);
                        if (m >= p.M) m = p.M - 1;
                        rres[t] = *reinterpret_cast<const u32x4_t*>(p.R + (size_t)m * p.ldr + c * LCH + (lane & 3) * 8);
                    }
                }
            }
            const char* s1 = smem + slot * LW_STAGE;
            bf16x8_t wr[PD];
            auto rd = [&](int kk, int sl) __attribute__((always_inline)) {
                wr[sl] = *(lds_vfrag_t*)((lds_char_t*)s1 + (kk >> 2) * 4096 + off128(fr, (kk & 3) * 2 + fh));
            };
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) rd(kk, kk);
            const bool more = h + 1 < KH || c + 1 < c_end;
            const uint64_t nb = h + 1 < KH ? w_base(c, h + 1) : w_base(c + 1, 0);
#pragma unroll
            for (int kk = 0; kk < FD / 16; ++kk) {
                bf16x8_t f = wr[kk % PD];
                if (kk + PD < FD / 16) rd(kk + PD, kk % PD);
                asm volatile("" : "+v"(f));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, xf[h * (FD / 16) + kk], acc, 0, 0, 0);
                if (kk < 5 && more) dma_piece(w_dst(slot ^ 1, kk), vo[kk], nb);
            }
            slot ^= 1;
        }
        // ---- chunk epilogue: + bias, bf16, row-major 16-byte stores through the wave-private patch (32 rows x 64 B)
        {
            const int n0 = c * LCH;
            const int rrow = lane >> 2, rc = lane & 3;
            if constexpr (RES) {
                // residual through the patch the other way round: row-major in, accumulator layout out - the sum is formed in
                // fp32 and rounded to bf16 once, as the tile GEMM's residual epilogue does (one wave's LDS operations execute in
                // order: no barrier)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = t * 16 + rrow;
                    *reinterpret_cast<u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4)) = rres[t];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 bv = {0.f, 0.f, 0.f, 0.f};
                if (bias_lds) bv = *reinterpret_cast<const float4*>(bls + n0 + 8 * q + 4 * fh);
                else if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n0 + 8 * q + 4 * fh);
                uint2* const slot = reinterpret_cast<uint2*>(ebuf + fr * 64 + (((2 * q + fh) ^ (((fr >> 1) & 3) << 1)) << 3));
                if constexpr (RES) {
                    const uint2 rv = *slot;
                    bv.x += __uint_as_float(rv.x << 16); bv.y += __uint_as_float(rv.x & 0xffff0000u);
                    bv.z += __uint_as_float(rv.y << 16); bv.w += __uint_as_float(rv.y & 0xffff0000u);
                }
                uint2 pk;
                pk.x = pack_bf2(acc[4 * q] + bv.x, acc[4 * q + 1] + bv.y);
                pk.y = pack_bf2(acc[4 * q + 2] + bv.z, acc[4 * q + 3] + bv.w);
                *slot = pk;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = t * 16 + rrow;
                const u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4));
                const int m = m0 + r;
                if (m < p.M) *reinterpret_cast<u32x4_t*>(p.O + (size_t)m * p.ldo + n0 + rc * 8) = d;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm + q/k/v projection + attention over the T = 16 frames of a spatial position, dim 320 (5 heads x 64), in ONE
// launch: att[M, 320] = softmax_T(q k^T scale) v with q, k, v = LayerNorm(x) Wq^T, Wk^T, Wv^T.
//   reference: TemporalTransformer.forward lvdm/modules/attention.py:365-412 -> BasicTransformerBlock._forward :242-244
//   (norm1 / norm2 -> attn1 / attn2, both self-attention over time) -> CrossAttention.forward :101-125
// As three kernels (LayerNorm, [M, 960] projection, temporal attention) the qkv tensor is written and read back: 2 x 566 MB
// per attention at the 1024 config, 11 times per UNet forward. Here a workgroup owns 8 positions x 16 frames = 128 rows
// (gathered: the frames of a position are HW rows apart), a wave 2 positions: its 32 (normalised) rows stay in registers
// as X fragments (norm_linear_kernel), the weight streams through LDS head by head (q, k, v: 6 chunks of 32 columns).
// The q and k blocks never leave registers: in accumulator layout a lane holds 16 of the 32 channels of ITS row, which
// is an MFMA operand fragment once both factors agree on the k order - S^T = K Q^T is 4 MFMAs on the packed accumulators.
// v goes through a wave-private LDS patch (transposed reads, as the flash kernel reads V^T); a 32 x 32 score block holds
// the two positions' 16 x 16 problems on its diagonal, the rest is masked. bf16 roundings are where the three-kernel
// path has them (q, k, v, P), the softmax is fp32.
constexpr int TA_VLD = 192;                    // bytes per v row in the patch (flash kernel's V_LD)
constexpr int TA_PATCH = 32 * TA_VLD;          // 6 KB per wave: v [32 rows][64 d], then the output rows of the head
constexpr int ta_ring(int kh) { return kh == 1 ? 2 : 4; }      // weight stages in LDS (dim 640 runs one workgroup per CU: nobody else
                                                                // covers the L2 latency of a stage, so it is requested three stages ahead)
constexpr int ta_lds(int kh) { return ta_ring(kh) * LW_STAGE + 4 * TA_PATCH + 2 * FD * kh * 4; }

struct TaParams {
    const bf16_t* X; int ldx;
    const bf16_t* W;             // [>= 3 C][C]: to_q, to_k, to_v rows (C = 320 KH)
    bf16_t* O; int ldo;
    const float* ln_g; const float* ln_b; float ln_eps;
    int HW;                      // positions per frame (% 8 == 0); T = 16
    float c;                     // scale * log2(e)
};

// KH = 1: dim 320 (level 0), 5 heads, two workgroups per CU. KH = 2: dim 640 (level 1), 10 heads: the X fragments of a row take
// 160 registers, a weight chunk (32 rows x 640 k) passes the ring as two stages (k halves, as norm_linear_kernel does), one
// workgroup per CU.
template <int KH>
__global__ __launch_bounds__(256, KH == 1 ? 2 : 1)
void ln_qkv_tattn_kernel(const TaParams p) {
    constexpr int KD = FD * KH;
    constexpr int NS = ta_ring(KH);                               // ring depth: a stage is requested NS - 1 stages ahead
    constexpr int SPH = 6 * KH;                                   // stages per head (SPH % NS == 0: slots are compile-time)
    static_assert(SPH % NS == 0, "ring phase");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    char* const vpatch = smem + NS * LW_STAGE + wave * TA_PATCH;
    float* const lns = reinterpret_cast<float*>(smem + NS * LW_STAGE + 4 * TA_PATCH);
    const int gpb = p.HW >> 3;                                    // 8-position groups per clip
    const int b = (int)blockIdx.x / gpb, p0 = ((int)blockIdx.x - b * gpb) * 8 + 2 * wave;
    auto grow = [&](int r) { return ((size_t)(b * 16 + (r & 15))) * p.HW + p0 + (r >> 4); };      // wave row r -> tensor row

    unsigned vo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int u = wave * 5 + i;
        const int t = u >> 2, g = u & 3;
        vo[i] = (unsigned)((g * 8 + (lane >> 3)) * (KD * 2) + t * 128 + (((lane & 7) ^ ((g * 4 + (lane >> 4)) & 7)) << 4));
        asm volatile("" : "+v"(vo[i]));
    }
    auto dma_piece = [&](unsigned lds_dst, unsigned voff, uint64_t sbase) __attribute__((always_inline)) {
        // (m0 is NOT saved and restored around a piece: nothing else in these kernels uses it - checked in the ISA, as for gemm_pipe16.h - and
        //  two scalar moves per piece are 8-10 issue clocks of a one-wave-per-SIMD stream)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2"
            :
            : "s"(lds_dst), "v"(voff), "s"(sbase)
            : "memory");
    };
    // chunk (head h, part): part 0,1 = q halves, 2,3 = k halves, 4,5 = v halves -> weight rows (part/2) * C + 64 h + 32 (part&1);
    // stage (chunk, kh): the k half kh of those rows
    auto w_base = [&](int h, int part, int kh) {
        return (uint64_t)(uintptr_t)p.W + (uint64_t)((part >> 1) * KD + 64 * h + 32 * (part & 1)) * (KD * 2) + (uint64_t)kh * (FD * 2);
    };
    auto w_dst = [&](int slot, int i) { return lds_base + slot * LW_STAGE + (wave * 5 + i) * 1024; };

    // gridDim.y workgroups share the heads of a row tile (DC_TA_HSPLIT=2 at dim 640: heads 0-4 / 5-9, x read twice - no faster)
    const int hpw = 5 * KH / (int)gridDim.y;
    const int h_begin = (int)blockIdx.y * hpw, h_end = h_begin + hpw;
    // stage index inside a head: lin = part * KH + kh (compile-time in the unrolled loops below)
#pragma unroll
    for (int d = 0; d < NS - 1; ++d)
#pragma unroll
        for (int i = 0; i < 5; ++i) dma_piece(w_dst(d, i), vo[i], w_base(h_begin + d / SPH, (d % SPH) / KH, d % KH));

    bf16x8_t xf[KD / 16];
    {
        const bf16_t* xr = p.X + grow(fr) * p.ldx + fh * 8;
#pragma unroll
        for (int kk = 0; kk < KD / 16; ++kk) xf[kk] = *reinterpret_cast<const bf16x8_t*>(xr + kk * 16);
        for (int i = tid; i < KD; i += 256) { lns[i] = p.ln_g[i]; lns[KD + i] = p.ln_b[i]; }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ln_rows_inplace<KD>(xf, lns, lns + KD, p.ln_eps, fh);

    constexpr int PD = 6;
    const int li = lane & 15;
    const int tr_off = (li >> 2) * TA_VLD + (((lane >> 4) & 1) * 16 + (li & 3) * 4) * 2;      // transposed v read (flash kernel)
    const int px = fr >> 4;                                       // which of the wave's two positions this lane's row is
    for (int h = h_begin; h < h_end; ++h) {
        bf16x8_t qf[4], kf[4];
#pragma unroll
        for (int part = 0; part < 6; ++part) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const int lin = part * KH + kh;                   // (compile-time after unrolling)
                const int slot = lin % NS;
                if (h > h_begin || lin > 0) {
                    // this stage's 5 pieces were issued NS - 1 stages ago; younger than them are the pieces of the stages between
                    // (5 each - fewer at the very end) and, on the first stage of a head, the previous head's 4 output stores
                    const int rem = (h + 1 == h_end) ? SPH - 1 - lin : NS;         // stages behind this one (>= NS - 2 is all that matters)
                    const int younger = rem < NS - 2 ? rem : NS - 2;
                    const int n = 5 * younger + ((lin == 0) ? 4 : 0);
                    if (n >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                const char* s1 = smem + slot * LW_STAGE;
                bf16x8_t wr[PD];
                auto rd = [&](int kk, int sl) __attribute__((always_inline)) {
                    wr[sl] = *(lds_vfrag_t*)((lds_char_t*)s1 + (kk >> 2) * 4096 + off128(fr, (kk & 3) * 2 + fh));
                };
#pragma unroll
                for (int kk = 0; kk < PD; ++kk) rd(kk, kk);
                // the stage NS - 1 ahead goes into the slot that stage lin - 1 has just left (every wave is past this stage's barrier)
                const int nl = lin + NS - 1;
                const int nh = h + nl / SPH;
                const bool more = nh < h_end;
                const uint64_t nb = w_base(nh, (nl % SPH) / KH, nl % KH);
#pragma unroll
                for (int kk = 0; kk < FD / 16; ++kk) {
                    bf16x8_t f = wr[kk % PD];
                    if (kk + PD < FD / 16) rd(kk + PD, kk % PD);
                    asm volatile("" : "+v"(f));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, xf[kh * (FD / 16) + kk], acc, 0, 0, 0);
                    if (kk < 5 && more) dma_piece(w_dst(nl % NS, kk), vo[kk], nb);
                }
            }
            if (part < 4) {
                // q / k block of this row: the two operand fragments of k steps 2 (part & 1), + 1
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    u32x4_t pw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(acc[8 * s2 + 2 * e], acc[8 * s2 + 2 * e + 1]);
                    if (part < 2) qf[2 * (part & 1) + s2] = __builtin_bit_cast(bf16x8_t, pw);
                    else kf[2 * (part & 1) + s2] = __builtin_bit_cast(bf16x8_t, pw);
                }
            } else {
                // v block -> the wave's patch, row-major [frame row][64 d]: channels 8 q + 4 fh + i of half (part & 1)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 pk;
                    pk.x = pack_bf2(acc[4 * q], acc[4 * q + 1]);
                    pk.y = pack_bf2(acc[4 * q + 2], acc[4 * q + 3]);
                    *reinterpret_cast<uint2*>(vpatch + fr * TA_VLD + (32 * (part & 1) + 8 * q + 4 * fh) * 2) = pk;
                }
            }
        }
        // ---- attention of head h over the 16 frames of each of the wave's two positions
        f32x16_t sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[j], qf[j], sc, 0, 0, 0);
        // lane = query row fr; sc[r] = score of key row n = (r & 3) + 8 (r >> 2) + 4 fh: the same position iff (r >> 3) == px
        float mx = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((r >> 3) != px) sc[r] = -1e30f;
            mx = fmaxf(mx, sc[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f((sc[r] - mx) * p.c);
            sum += sc[r];
        }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        bf16x8_t pf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t pw;
#pragma unroll
            for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(sc[8 * ks + 2 * e] * inv, sc[8 * ks + 2 * e + 1] * inv);
            pf[ks] = __builtin_bit_cast(bf16x8_t, pw);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's own v rows have landed in its patch
        __builtin_amdgcn_wave_barrier();
        f32x16_t oacc[2];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const char* vp = vpatch + (16 * ks + 4 * fh) * TA_VLD + db * 64 + tr_off;
                const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp));
                const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp + 8 * TA_VLD));
                bf16x8_t vf;
                vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[ks], oacc[db], 0, 0, 0);
            }
        }
        // ---- the head's 32 x 64 output block: bf16, row-major through the (now free) patch, 128-byte row segments
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 pk;
                pk.x = pack_bf2(oacc[db][4 * q], oacc[db][4 * q + 1]);
                pk.y = pack_bf2(oacc[db][4 * q + 2], oacc[db][4 * q + 3]);
                *reinterpret_cast<uint2*>(vpatch + off128(fr, 4 * db + q) + fh * 8) = pk;
            }
        {
            const int rrow = lane >> 3, rc = lane & 7;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = t * 8 + rrow;
                const u32x4_t d = *reinterpret_cast<const u32x4_t*>(vpatch + off128(r, rc));
                *reinterpret_cast<u32x4_t*>(p.O + grow(r) * p.ldo + 64 * h + rc * 8) = d;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm (known statistics) + SiLU + temporal convolution (3,1,1) (+ residual) for 320 channels in ONE launch:
//   out[b, f, p, :] = bias + sum_t W_t a[b, f + t - 1, p, :] (zero outside the clip),  a = silu(GroupNorm(x))
//   reference: TemporalConvBlock lvdm/modules/networks/openaimodel3d.py:239-279 (conv1..conv4 = GroupNorm(32) -> SiLU ->
//   Conv3d (3,1,1), zero padding in time); the block's `identity + x` (:279) rides on the last conv as `residual`.
// As GroupNorm (statistics, apply) + implicit GEMM the activated copy is written, then read three times (once per tap)
// through the L2. Here the rows are gathered as in ln_qkv_tattn320_kernel: a wave owns 2 positions x 16 frames, so the
// frames f - 1 and f + 1 of a row are its NEIGHBOUR LANES. The normalised, activated X fragments stay in registers; per
// 32-column chunk three products Y_t = X W_t^T (the same X, 20 MFMAs each) and
//   out[f] = Y_0[f - 1] + Y_1[f] + Y_2[f + 1]
// is two DPP row shifts (16-lane rows = the 16 frames of a position; the shifted-in lane is 0 = the zero padding).
struct TcParams {
    const bf16_t* X; int ldx;
    const bf16_t* W;             // PackedWeight.tconv3: [>= N][3 C], k = (64-channel slice, tap, channel in slice)
    const float* bias;           // [N]
    const bf16_t* R; int ldr;    // residual rows or nullptr
    bf16_t* O; int ldo;
    const float* gn_g; const float* gn_b; const float2* gn_stats; int gn_groups;      // stats [clip][group] = (mean, rstd)
    int HW, N;
    int nsplit, cpp;             // workgroups per row tile, chunks per workgroup (few row tiles: split N)
    int ntiles;                  // row tiles (the grid is padded to groups of 8 tiles x nsplit)
};

// C = 320 KH input channels; the taps run in the order 1, 0, 2 so that only `out` and the current product are live
template <int KH, bool RES>
__global__ __launch_bounds__(256, 2)
void gn_silu_tconv_kernel(const TcParams p) {
    constexpr int KD = FD * KH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    char* const ebuf = smem + 2 * LW_STAGE + wave * 2048;
    float* const lns = reinterpret_cast<float*>(smem + 2 * LW_STAGE + 4 * 2048);
    float* const bls = lns + 2 * KD;                     // the bias in LDS (as in norm_linear_kernel)
    const bool bias_lds = KH == 1 && p.N <= LL_BIAS;     // (C = 640 sits at 256 registers: it keeps its loads in the epilogue)
    const int bgrp = (int)blockIdx.x / (8 * p.nsplit), brem = (int)blockIdx.x - bgrp * (8 * p.nsplit);
    const int tile = bgrp * 8 + (brem & 7), part = brem >> 3;      // the parts of a tile on one XCD (see norm_linear_kernel)
    if (tile >= p.ntiles) return;
    const int gpb = p.HW >> 3;
    const int b = tile / gpb, p0 = (tile - b * gpb) * 8 + 2 * wave;
    auto grow = [&](int r) { return ((size_t)(b * 16 + (r & 15))) * p.HW + p0 + (r >> 4); };      // wave row r -> tensor row
    const int c_begin = part * p.cpp;
    int c_end = c_begin + p.cpp;
    if (c_end > p.N / LCH) c_end = p.N / LCH;
    if (c_begin >= c_end) return;

    // LDS-DMA of stage (chunk c, tap t, k half h): 32 weight rows x 320 channels of the tap = 5 K tiles of [32 rows][128 B];
    // in the packed weight the K tile of 64-channel slice s and tap t starts at column 192 s + 64 t
    unsigned vo[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int u = wave * 5 + i;
        const int t = u >> 2, g = u & 3;
        vo[i] = (unsigned)((g * 8 + (lane >> 3)) * (3 * KD * 2) + t * 384 + (((lane & 7) ^ ((g * 4 + (lane >> 4)) & 7)) << 4));
        asm volatile("" : "+v"(vo[i]));
    }
    auto dma_piece = [&](unsigned lds_dst, unsigned voff, uint64_t sbase) __attribute__((always_inline)) {
        // (m0 is NOT saved and restored around a piece: nothing else in these kernels uses it - checked in the ISA, as for gemm_pipe16.h - and
        //  two scalar moves per piece are 8-10 issue clocks of a one-wave-per-SIMD stream)
        asm volatile(
            "s_mov_b32 m0, %0\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2"
            :
            : "s"(lds_dst), "v"(voff), "s"(sbase)
            : "memory");
    };
    // stage index within a chunk: j = 0 .. 3 KH - 1, tap = order[j / KH], half = j % KH
    auto w_base = [&](int c, int j) {
        const int jt = j / KH, h = j - jt * KH;
        const int tap = jt == 0 ? 1 : (jt == 1 ? 0 : 2);
        return (uint64_t)(uintptr_t)p.W + (uint64_t)c * (LCH * 3 * KD * 2) + tap * 128 + h * (5 * 384);
    };
    auto w_dst = [&](int slot, int i) { return lds_base + slot * LW_STAGE + (wave * 5 + i) * 1024; };

#pragma unroll
    for (int i = 0; i < 5; ++i) dma_piece(w_dst(0, i), vo[i], w_base(c_begin, 0));

    bf16x8_t xf[KD / 16];
    {
        const bf16_t* xr = p.X + grow(fr) * p.ldx + fh * 8;
#pragma unroll
        for (int kk = 0; kk < KD / 16; ++kk) xf[kk] = *reinterpret_cast<const bf16x8_t*>(xr + kk * 16);
        for (int i = tid; i < KD; i += 256) {            // one clip per workgroup: a = gamma rstd, b = beta - mean a
            const float2 st = p.gn_stats[(size_t)b * p.gn_groups + i / (KD / p.gn_groups)];
            const float a = p.gn_g[i] * st.y;
            lns[i] = a; lns[KD + i] = p.gn_b[i] - st.x * a;
        }
        if (bias_lds) for (int i = tid; i < p.N; i += 256) bls[i] = p.bias[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    affine_rows_inplace<KD, true>(xf, lns, lns + KD, fh);

    constexpr int PD = KH == 1 ? 6 : 3;                  // K = 640: 160 registers of X fragments, a shorter LDS read window fits 256
    constexpr int NS = 3 * KH;                           // stages per chunk
    int slot = 0;
    for (int c = c_begin; c < c_end; ++c) {
        f32x16_t out, y;
        u32x4_t rres[2];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int jt = j / KH, h = j % KH;            // jt 0: tap 1 (centre) -> out; 1: tap 0 (frame - 1); 2: tap 2 (frame + 1)
            if (c > c_begin || j > 0) {
                // this stage was issued during the previous one, in front of the loads / stores that closed a chunk
                if (j == 0) { if (RES && KH != 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            if constexpr (RES && KH == 1) {
                // the residual rows of this chunk in the stores' coalesced pattern, requested a chunk's MFMAs ahead of the
                // epilogue (as loads inside the epilogue they were an exposed round trip per chunk)
                if (j == 0) {
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        rres[t] = *reinterpret_cast<const u32x4_t*>(p.R + grow(t * 16 + (lane >> 2)) * p.ldr + c * LCH + (lane & 3) * 8);
                }
            }
            const char* s1 = smem + slot * LW_STAGE;
            f32x16_t& acc = jt == 0 ? out : y;
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            bf16x8_t wr[PD];
            auto rd = [&](int kk, int sl) __attribute__((always_inline)) {
                wr[sl] = *(lds_vfrag_t*)((lds_char_t*)s1 + (kk >> 2) * 4096 + off128(fr, (kk & 3) * 2 + fh));
            };
#pragma unroll
            for (int kk = 0; kk < PD; ++kk) rd(kk, kk);
            const bool more = j + 1 < NS || c + 1 < c_end;
            const uint64_t nb = j + 1 < NS ? w_base(c, j + 1) : w_base(c + 1, 0);
#pragma unroll
            for (int kk = 0; kk < FD / 16; ++kk) {
                bf16x8_t f = wr[kk % PD];
                if (kk + PD < FD / 16) rd(kk + PD, kk % PD);
                asm volatile("" : "+v"(f));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, xf[h * (FD / 16) + kk], acc, 0, 0, 0);
                if (kk < 5 && more) dma_piece(w_dst(slot ^ 1, kk), vo[kk], nb);
            }
            slot ^= 1;
            if (jt > 0 && h == KH - 1) {
                // out[f] += Y_0[f - 1] (row_shr:1) resp. Y_2[f + 1] (row_shl:1): the 16 lanes of a DPP row are the 16 frames
                // of one position, the lane shifted in from outside the row is 0 = the zero padding in time
                // (copies first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this hipcc)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float yv = y[r];
                    out[r] += __int_as_float(jt == 1 ? __builtin_amdgcn_update_dpp(0, __float_as_int(yv), 0x111, 0xf, 0xf, false)
                                                     : __builtin_amdgcn_update_dpp(0, __float_as_int(yv), 0x101, 0xf, 0xf, false));
                }
            }
        }
        // ---- chunk epilogue: + bias, bf16, (+ residual), row-major through the wave-private patch; rows scattered back
        {
            const int n0 = c * LCH;
            const int rrow = lane >> 2, rc = lane & 3;
            if constexpr (RES) {
                if constexpr (KH != 1) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) rres[t] = *reinterpret_cast<const u32x4_t*>(p.R + grow(t * 16 + rrow) * p.ldr + n0 + rc * 8);
                }
                // residual through the patch the other way round (row-major in, accumulator layout out): the sum is formed in
                // fp32 and rounded once
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = t * 16 + rrow;
                    *reinterpret_cast<u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4)) = rres[t];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 bv = bias_lds ? *reinterpret_cast<const float4*>(bls + n0 + 8 * q + 4 * fh)
                                     : *reinterpret_cast<const float4*>(p.bias + n0 + 8 * q + 4 * fh);
                uint2* const slot2 = reinterpret_cast<uint2*>(ebuf + fr * 64 + (((2 * q + fh) ^ (((fr >> 1) & 3) << 1)) << 3));
                if constexpr (RES) {
                    const uint2 rv = *slot2;
                    bv.x += __uint_as_float(rv.x << 16); bv.y += __uint_as_float(rv.x & 0xffff0000u);
                    bv.z += __uint_as_float(rv.y << 16); bv.w += __uint_as_float(rv.y & 0xffff0000u);
                }
                uint2 pk;
                pk.x = pack_bf2(out[4 * q] + bv.x, out[4 * q + 1] + bv.y);
                pk.y = pack_bf2(out[4 * q + 2] + bv.z, out[4 * q + 3] + bv.w);
                *slot2 = pk;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int r = t * 16 + rrow;
                const u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4));
                *reinterpret_cast<u32x4_t*>(p.O + grow(r) * p.ldo + n0 + rc * 8) = d;
            }
        }
    }
}

}  // namespace

#ifdef DC_FF_STAMPS
extern "C" int dc_ff_debug_stamps(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ff_stamps), sizeof(unsigned long long) * 8);
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_ff_stamps), z, sizeof(z));
    }
    return (int)e;
}
#endif

template <bool LN, bool PROJ>
static int launch_ff_fused(const FfParams& p, hipStream_t stream) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&ff_geglu_fused320_kernel<LN, PROJ>), FF_LDS)) return e;
    const int tiles = (p.M + FBM - 1) / FBM;
    hipLaunchKernelGGL((ff_geglu_fused320_kernel<LN, PROJ>), dim3(tiles < 256 ? tiles : 256), dim3(256), FF_LDS, stream, p);
    DC_CHECK_LAUNCH();
    return 0;
}

static int ff_fused_common(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                           const uint16_t* w1, const float* b1, const uint16_t* w2p, const float* b2, const uint16_t* residual,
                           int ldr, uint16_t* out, int ldo, const uint16_t* wp, const float* bp, const uint16_t* residual2,
                           int ldr2, uint16_t* out2, int ldo2, int M, hipStream_t stream) {
    FfParams p;
    p.X = x; p.ldx = ldx; p.W1 = w1; p.b1 = b1; p.W2p = w2p; p.b2 = b2; p.R = residual; p.ldr = ldr; p.O = out; p.ldo = ldo;
    p.M = M; p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps;
    p.Wp = wp; p.bp = bp; p.R2 = residual2; p.ldr2 = ldr2; p.O2 = out2; p.ldo2 = ldo2;
    if (wp) return ln_gamma ? launch_ff_fused<true, true>(p, stream) : launch_ff_fused<false, true>(p, stream);
    return ln_gamma ? launch_ff_fused<true, false>(p, stream) : launch_ff_fused<false, false>(p, stream);
}

extern "C" int dc_ff_geglu_fused320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                    const uint16_t* w1, const float* b1, const uint16_t* w2p, const float* b2,
                                    const uint16_t* residual, int ldr, uint16_t* out, int ldo, int M, void* stream_) {
    if (!x || !w1 || !b1 || !w2p || !b2 || !out || ((ln_gamma == nullptr) != (ln_beta == nullptr))) return DC_ERR_ARG;
    if (M < 1 || ldx % 8 || ldo % 8 || (residual && ldr % 8)) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)(residual ? residual : out)) % 16) return DC_ERR_SHAPE;
    return ff_fused_common(x, ldx, ln_gamma, ln_beta, ln_eps, w1, b1, w2p, b2, residual, ldr, out, ldo, nullptr, nullptr,
                           nullptr, 0, nullptr, 0, M, (hipStream_t)stream_);
}

extern "C" int dc_ff_geglu_proj_fused320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                         const uint16_t* w1, const float* b1, const uint16_t* w2p, const float* b2,
                                         const uint16_t* wp, const float* bp, const uint16_t* residual2, int ldr2,
                                         uint16_t* out, int ldo, int M, void* stream_) {
    if (!x || !w1 || !b1 || !w2p || !b2 || !wp || !bp || !residual2 || !out || ((ln_gamma == nullptr) != (ln_beta == nullptr)))
        return DC_ERR_ARG;
    if (M < 1 || ldx % 8 || ldo % 8 || ldr2 % 8) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)wp | (uintptr_t)residual2) % 16) return DC_ERR_SHAPE;
    // the FeedForward's own residual is its input x
    return ff_fused_common(x, ldx, ln_gamma, ln_beta, ln_eps, w1, b1, w2p, b2, x, ldx, nullptr, 0, wp, bp, residual2, ldr2, out,
                           ldo, M, (hipStream_t)stream_);
}

template <int NORM, int KH, bool RES = false>
static int launch_norm_linear_t(const LlParams& p, dim3 grid, hipStream_t stream) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&norm_linear_kernel<NORM, KH, RES>), ll_lds(KH))) return e;
    hipLaunchKernelGGL((norm_linear_kernel<NORM, KH, RES>), grid, dim3(256), ll_lds(KH), stream, p);
    DC_CHECK_LAUNCH();
    return 0;
}

static int launch_norm_linear(int norm, int K, LlParams& p, hipStream_t stream) {
    // few row tiles: split N over several workgroups per tile until the launch has >= 3 rounds of work for its slots
    const int tiles = (p.M + FBM - 1) / FBM, nch = p.N / LCH;
    const int slots = 256 * (K == FD ? 3 : 2);
    int nsplit = 1;
    while (tiles * nsplit < 3 * slots && nsplit * 2 <= nch && nch / (nsplit * 2) >= 5) nsplit *= 2;
    p.nsplit = nsplit;
    p.cpp = (nch + nsplit - 1) / nsplit;
    const dim3 grid((tiles + 7) / 8 * 8 * nsplit);
    if (p.R) return K == FD ? launch_norm_linear_t<0, 1, true>(p, grid, stream) : launch_norm_linear_t<0, 2, true>(p, grid, stream);
    if (K == FD) {
        if (norm == 1) return launch_norm_linear_t<1, 1>(p, grid, stream);
        if (norm == 2) return launch_norm_linear_t<2, 1>(p, grid, stream);
        return launch_norm_linear_t<0, 1>(p, grid, stream);
    }
    if (norm == 1) return launch_norm_linear_t<1, 2>(p, grid, stream);
    if (norm == 2) return launch_norm_linear_t<2, 2>(p, grid, stream);
    return launch_norm_linear_t<0, 2>(p, grid, stream);
}

extern "C" int dc_ln_linear(const uint16_t* x, int ldx, int K, const float* ln_gamma, const float* ln_beta, float ln_eps,
                            const uint16_t* w, const float* bias, uint16_t* out, int ldo, int M, int N, void* stream_) {
    if (!x || !w || !out || ((ln_gamma == nullptr) != (ln_beta == nullptr))) return DC_ERR_ARG;
    if ((K != FD && K != 2 * FD) || M < 1 || N < LCH || N % LCH || ldx % 8 || ldo % 8) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w) % 16) return DC_ERR_SHAPE;
    LlParams p;
    p.X = x; p.ldx = ldx; p.W = w; p.bias = bias; p.O = out; p.ldo = ldo; p.M = M; p.N = N;
    p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps; p.gn_stats = nullptr; p.gn_groups = 1; p.gn_rpi = FBM;
    p.R = nullptr; p.ldr = 0;
    return launch_norm_linear(ln_gamma ? 1 : 0, K, p, (hipStream_t)stream_);
}

extern "C" int dc_linear_residual(const uint16_t* x, int ldx, int K, const uint16_t* w, const float* bias,
                                  const uint16_t* residual, int ldr, uint16_t* out, int ldo, int M, int N, void* stream_) {
    if (!x || !w || !out || !residual) return DC_ERR_ARG;
    if ((K != FD && K != 2 * FD) || M < 1 || N < LCH || N % LCH || ldx % 8 || ldo % 8 || ldr % 8) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)residual) % 16) return DC_ERR_SHAPE;
    LlParams p;
    p.X = x; p.ldx = ldx; p.W = w; p.bias = bias; p.O = out; p.ldo = ldo; p.M = M; p.N = N;
    p.ln_g = nullptr; p.ln_b = nullptr; p.ln_eps = 0.f; p.gn_stats = nullptr; p.gn_groups = 1; p.gn_rpi = FBM;
    p.R = residual; p.ldr = ldr;
    return launch_norm_linear(0, K, p, (hipStream_t)stream_);
}

extern "C" int dc_gn_linear(const uint16_t* x, int ldx, int K, const float* gamma, const float* beta, const float* stats,
                            int groups, int rows_per_inst, const uint16_t* w, const float* bias, uint16_t* out, int ldo,
                            int M, int N, void* stream_) {
    if (!x || !w || !out || !gamma || !beta || !stats) return DC_ERR_ARG;
    if ((K != FD && K != 2 * FD) || M < 1 || N < LCH || N % LCH || ldx % 8 || ldo % 8) return DC_ERR_SHAPE;
    if (groups < 1 || K % groups || rows_per_inst < FBM || rows_per_inst % FBM || M % rows_per_inst) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w) % 16) return DC_ERR_SHAPE;
    LlParams p;
    p.X = x; p.ldx = ldx; p.W = w; p.bias = bias; p.O = out; p.ldo = ldo; p.M = M; p.N = N;
    p.ln_g = gamma; p.ln_b = beta; p.ln_eps = 0.f;
    p.gn_stats = reinterpret_cast<const float2*>(stats); p.gn_groups = groups; p.gn_rpi = rows_per_inst;
    p.R = nullptr; p.ldr = 0;
    return launch_norm_linear(2, K, p, (hipStream_t)stream_);
}

template <int KH>
static int launch_ln_qkv_tattn(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                               const uint16_t* wqkv, uint16_t* out, int ldo, int B, int T, int HW, float scale, void* stream_) {
    if (!x || !ln_gamma || !ln_beta || !wqkv || !out) return DC_ERR_ARG;
    if (B < 1 || T != 16 || HW < 8 || HW % 8 || ldx % 8 || ldo % 8) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)wqkv) % 16) return DC_ERR_SHAPE;
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&ln_qkv_tattn_kernel<KH>), ta_lds(KH))) return e;
    TaParams p;
    p.X = x; p.ldx = ldx; p.W = wqkv; p.O = out; p.ldo = ldo; p.ln_g = ln_gamma; p.ln_b = ln_beta; p.ln_eps = ln_eps;
    p.HW = HW; p.c = scale * 1.4426950408889634f;
    static const int hsplit_env = [] { const char* e = getenv("DC_TA_HSPLIT"); return e ? atoi(e) : 0; }();
    const int hsplit = (KH == 2 && hsplit_env == 2) ? 2 : 1;      // heads of a row tile over 1 or 2 workgroups (measured: 256-261 vs 266-285 us)
    hipLaunchKernelGGL(ln_qkv_tattn_kernel<KH>, dim3((unsigned)(B * (HW / 8)), hsplit), dim3(256), ta_lds(KH), (hipStream_t)stream_, p);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_ln_qkv_temporal_attn320(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                          const uint16_t* wqkv, uint16_t* out, int ldo, int B, int T, int HW, float scale,
                                          void* stream_) {
    return launch_ln_qkv_tattn<1>(x, ldx, ln_gamma, ln_beta, ln_eps, wqkv, out, ldo, B, T, HW, scale, stream_);
}

extern "C" int dc_ln_qkv_temporal_attn640(const uint16_t* x, int ldx, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                          const uint16_t* wqkv, uint16_t* out, int ldo, int B, int T, int HW, float scale,
                                          void* stream_) {
    return launch_ln_qkv_tattn<2>(x, ldx, ln_gamma, ln_beta, ln_eps, wqkv, out, ldo, B, T, HW, scale, stream_);
}

template <int KH, bool RES>
static int launch_gn_silu_tconv(const TcParams& p, dim3 grid, hipStream_t stream) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gn_silu_tconv_kernel<KH, RES>), ll_lds(KH))) return e;
    hipLaunchKernelGGL((gn_silu_tconv_kernel<KH, RES>), grid, dim3(256), ll_lds(KH), stream, p);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_gn_silu_tconv3(const uint16_t* x, int ldx, int C, const float* gamma, const float* beta, const float* stats,
                                 int groups, const uint16_t* w, const float* bias, const uint16_t* residual, int ldr,
                                 uint16_t* out, int ldo, int B, int T, int HW, int N, void* stream_) {
    if (!x || !gamma || !beta || !stats || !w || !bias || !out) return DC_ERR_ARG;
    if ((C != FD && C != 2 * FD) || B < 1 || T != 16 || HW < 8 || HW % 8 || N < LCH || N % LCH || ldx % 8 || ldo % 8 ||
        (residual && ldr % 8)) return DC_ERR_SHAPE;
    if (groups < 1 || C % groups) return DC_ERR_SHAPE;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)w | (uintptr_t)(residual ? residual : out)) % 16) return DC_ERR_SHAPE;
    TcParams p;
    p.X = x; p.ldx = ldx; p.W = w; p.bias = bias; p.R = residual; p.ldr = ldr; p.O = out; p.ldo = ldo;
    p.gn_g = gamma; p.gn_b = beta; p.gn_stats = reinterpret_cast<const float2*>(stats); p.gn_groups = groups; p.HW = HW; p.N = N;
    // few row tiles: split N over several workgroups per tile until the launch has >= 3 rounds of work for its slots
    const int tiles = B * (HW / 8), nch = N / LCH, slots = 256 * 2;
    int nsplit = 1;
    while (tiles * nsplit < 3 * slots && nsplit * 2 <= nch && nch / (nsplit * 2) >= 5) nsplit *= 2;
    p.nsplit = nsplit; p.cpp = (nch + nsplit - 1) / nsplit; p.ntiles = tiles;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8 * nsplit));
    hipStream_t stream = (hipStream_t)stream_;
    if (C == FD) return residual ? launch_gn_silu_tconv<1, true>(p, grid, stream) : launch_gn_silu_tconv<1, false>(p, grid, stream);
    return residual ? launch_gn_silu_tconv<2, true>(p, grid, stream) : launch_gn_silu_tconv<2, false>(p, grid, stream);
}
