// Shared device helpers for the gfx950 (CDNA4 / MI355X) kernels of the DynamiCrafter denoising path.
// Wavefront = 64 lanes everywhere in this tree; nothing here is written for any other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef uint16_t bf16_t;  // raw bf16 bits; all activations/weights in HBM are bf16 unless a kernel says fp32

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 16x16x16 fragment (2 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // 16-byte staging register

#define DC_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;

// fp32 -> bf16, round-to-nearest-even, NaN stays NaN: a plain vector convert, which hipcc lowers to ONE
// v_cvt_pk_bf16_f32 per pair on gfx950 (the integer-arithmetic form costs ~8 VALU ops per element and is what made
// the attention kernel VALU-bound).
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}

__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

__device__ __forceinline__ void unpack_bf8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack_bf8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (1 ulp each; the outputs are rounded to bf16): an IEEE division here costs ~10
// more vector instructions per element and made the GroupNorm+SiLU pass instruction-bound instead of HBM-bound
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}

// erf-form GELU, torch.nn.functional.gelu default (reference: lvdm/modules/attention.py:422). erf by Abramowitz &
// Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the output): 1 v_rcp + 1 v_exp + 7 FMAs instead
// of libm erff's ~40-instruction branchy sequence, which showed up in the GEGLU epilogues (377 M evaluations per
// level-0 FeedForward).
__device__ __forceinline__ float erf_as_f(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752f)); }

// GELU for the GEGLU / GELU epilogues of the GEMMs: gelu(x) = x Phi(x) with Phi(x) - 1/2 = xc Q(xc^2), xc = x clamped to
// [-4, 4], Q a degree-6 minimax polynomial fitted under the constraint 4 Q(16) = 1/2 (so Phi is exactly 0 / 1 beyond the clamp,
// c0 nudged one ulp so that this also holds in fp32 FMA arithmetic). max |gelu_phi_f(x) - x Phi(x)| = 1.9e-4 over all x
// (tests/test_ops_gpu.py::test_gelu_phi_error pins it against torch's erf GELU), below the bf16 rounding of any output above
// 0.05 in magnitude. 11 full-rate vector instructions (v_med3, v_mul, 7 v_fma, 2 v_mul with the GEGLU product) instead of the
// 13 + v_rcp_f32 + v_exp_f32 of gelu_erf_f: the epilogues of the short-K GEGLU GEMMs are issue-bound.
__device__ __forceinline__ float phi_poly_f(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -4.0f, 4.0f);
    const float t = xc * xc;
    float q = fmaf(2.258878772920525e-08f, t, -1.5888579127931735e-06f);
    q = fmaf(q, t, 4.776452260557562e-05f);
    q = fmaf(q, t, -0.0008121939026750624f);
    q = fmaf(q, t, 0.008763724006712437f);
    q = fmaf(q, t, -0.06455449014902115f);
    q = fmaf(q, t, 0.3978703022003174f);
    return fmaf(xc, q, 0.5f);
}
__device__ __forceinline__ float gelu_phi_f(float x) { return x * phi_poly_f(x); }
#ifdef DC_GELU_ERF      // tool build (same-box A/B of the epilogue arithmetic): the erf form in every GEMM epilogue
#define DC_GELU(x) gelu_erf_f(x)
#else
#define DC_GELU(x) gelu_phi_f(x)
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a linear workgroup id: the 8 XCDs receive workgroups round-robin, so
// ids b and b+8 share an L2. Give every XCD a contiguous run of logical tiles.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// which kernel family dc_gemm_conv's dispatcher launched last on this host thread (profiling label; gemm_conv.hip)
void dc_note_variant(const char* name);

// gemm_conv_glds.hip: the current dc_gemm_set_plan value (kernel selection bits, include/dcrafter_hip.h)
int dc_gemm_plan_now();

// flash_pipe.hip: long self-attention (Lk % 64 == 0, Lk >= 256, no accumulate epilogue); called by dc_flash_attn_d64
int dc_flash_pipe_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, int ldq, int ldk, int ldv, int ldo,
                         int batch, int heads, int Lq, int Lk, int64_t q_bstride, int64_t kv_bstride, float c,
                         hipStream_t stream);

// The library's ERROR WORD (runtime.hip): one int per device. The kernels that synchronise through LDS arrival counters bound
// every wait (a bookkeeping mistake must not hang the GPU); a wave whose wait ran out ORs its bit into this word before it
// leaves, so the host can refuse results that were computed past a timed-out wait (dc_error_word_read). Returns the device
// address for the current device (nullptr on a HIP error); no allocation, no stream operation: safe under stream capture.
int* dc_error_word_device();
#define DC_ERRW_GEMM_PIPE 1      // gemm_pipe320x16_kernel: `landed` / `freed` counter wait timed out
#define DC_ERRW_FLASH_RING 2     // flash_attn_d64_pipe_kernel: K/V ring arrival counter wait timed out
#define DC_ERRW_GEMM_PP 4        // gemm_pp_kernel (ping-pong GEGLU / plain GEMM): counter wait timed out

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: kernels with more than 64 KB of dynamic LDS are
// configured once per device (bit mask; two host threads racing on the first call both set the attribute, which is harmless).
struct DcLdsOnce {
    std::atomic<unsigned long long> done{0};
    int ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return (int)e;
        const bool tracked = dev >= 0 && dev < 64;
        if (tracked && (done.load(std::memory_order_acquire) >> dev & 1ull)) return 0;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return (int)e;
        if (tracked) done.fetch_or(1ull << dev, std::memory_order_release);
        return 0;
    }
};

#define DC_CHECK_LAUNCH() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
