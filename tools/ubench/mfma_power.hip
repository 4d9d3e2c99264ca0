// What the matrix pipe sustains when nothing else runs: one wave per SIMD (or two), a loop of independent
// v_mfma_f32_32x32x16_bf16 on operands that stay in registers. The only variable is the VALUE of the operands: zeros,
// a constant, +-1 (sign bits only), or random bf16 bit patterns - the chip is power-managed, the clock it holds under an
// MFMA stream depends on how many bits toggle.       build: hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0 zeros | 1 all ones (1.0) | 2 +-1 random signs | 3 random normal-ish bf16 (random sign, exponent near 0, random mantissa)
// | 4 random 16-bit patterns restricted to finite values
__device__ unsigned make_pair(int mode, unsigned seed) {
    const unsigned h = hash32(seed);
    auto one = [&](unsigned r) -> unsigned {
        if (mode == 0) return 0u;
        if (mode == 1) return 0x3f80u;
        if (mode == 2) return 0x3f80u | ((r & 1u) << 15);
        if (mode == 3) return ((r & 1u) << 15) | ((0x7cu + ((r >> 1) & 7u)) << 7) | ((r >> 4) & 0x7fu);
        return ((r & 0x8000u)) | ((((r >> 7) & 0xffu) % 0xfeu) << 7) | (r & 0x7fu);
    };
    return one(h & 0xffffu) | (one(h >> 16) << 16);
}

// SCHED 0: both operand fragments change with every MFMA | 1: the A fragment stays for 8 consecutive MFMAs | 2: both stay
// SHAPE 0: v_mfma_f32_32x32x16_bf16 | 1: v_mfma_f32_16x16x32_bf16 (same flops per instruction pair: 2 per 32x32x16)
template <int NFRAG, int SCHED, int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(int iters, int mode, unsigned long long* stamps, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4_t a[NFRAG], b[NFRAG];
#pragma unroll
    for (int i = 0; i < NFRAG; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[i][e] = make_pair(mode, (blockIdx.x * 256 + threadIdx.x) * 64 + i * 8 + e);
            b[i][e] = make_pair(mode, (blockIdx.x * 256 + threadIdx.x) * 64 + i * 8 + e + 4 + 1000003);
        }
    f32x16_t acc[8];
    f32x4_t acc4[16];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ia = SCHED == 0 ? i % NFRAG : 0, ib = SCHED == 2 ? 0 : (i * 3 + 1) % NFRAG;
            if constexpr (SHAPE == 0) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ia]), __builtin_bit_cast(bf16x8_t, b[ib]), acc[i], 0, 0, 0);
            } else {
                // two 16x16x32 per 32x32x16 worth of flops, on accumulators of their own
                // (asm: with the builtin hipcc keeps these accumulators in the accumulator file and copies them per iteration)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[2 * i]) : "v"(a[ia]), "v"(b[ib]));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc4[2 * i + 1]) : "v"(a[ia]), "v"(b[(ib + 1) % NFRAG]));
            }
        }
        if (mode >= 3 && (it & 63) == 63) {        // keep the accumulators finite and their bits moving
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= 0.0009765625f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc4[i][r] *= 0.0009765625f;
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc4[i][r];
    if (s == 12345.678f) sink[0] = s;
    if (lane == 0 && wave == 0) { stamps[blockIdx.x * 2] = c1 - c0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    unsigned long long* st; float* sink;
    hipMalloc(&st, 4096 * 16); hipMalloc(&sink, 64);
    const char* names[] = {"zeros", "ones", "+-1 (sign bits)", "random sign/mantissa, exponents 2^-3..2^4", "random finite bf16 patterns"};
    auto run = [&](auto kern, const char* what, int grid, int mode) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kern<<<grid, 256>>>(iters / 10, mode, st, sink);      // warm-up: let the power management settle
        (void)hipEventRecord(e0);
        kern<<<grid, 256>>>(iters, mode, st, sink);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
        const double flop = (double)grid * 4 * iters * 8 * 32768.0;
        printf("%-34s %-44s %8.2f ms  %7.1f TFLOP/s  shader clock %.2f GHz\n", what, names[mode], ms, flop / ms / 1e9, (double)h[0] / (double)h[1] * 0.1);
    };
    for (int mode = 0; mode < 5; ++mode) run(mfma_loop<4, 0, 0>, "32x32x16, 1 wave/SIMD", 256, mode);
    for (int mode = 0; mode < 5; ++mode) run(mfma_loop<4, 0, 0>, "32x32x16, 2 waves/SIMD", 512, mode);
    run(mfma_loop<4, 1, 0>, "32x32x16, A kept for 8 MFMAs", 256, 3);
    run(mfma_loop<4, 2, 0>, "32x32x16, A and B kept", 256, 3);
    run(mfma_loop<4, 0, 1>, "16x16x32 (2 per 32x32x16)", 256, 0);
    run(mfma_loop<4, 0, 1>, "16x16x32 (2 per 32x32x16)", 256, 3);
    run(mfma_loop<4, 0, 1>, "16x16x32 (2 per 32x32x16)", 256, 4);
    run(mfma_loop<4, 1, 1>, "16x16x32, A kept for 8 pairs", 256, 3);
    run(mfma_loop<4, 0, 0>, "32x32x16 again", 256, 3);
    return 0;
}
