// Micro-benchmark: do MFMA (one wave) and VALU / transcendental work (the other wave of the same SIMD) co-execute on
// gfx950? 512 threads = 8 waves = 2 per SIMD. role[wave]: 0 idle, 1 MFMA loop, 2 VALU loop (exp2 + fma + add, the
// flash-attention softmax mix), 3 both interleaved in one wave (1 MFMA : 10 VALU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int RA, int RB>
__global__ __launch_bounds__(512) void k(int iters, float* sink, float c) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? RA : RB;
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
    float v[16]; float sum = 0.f;
    for (int r = 0; r < 16; ++r) v[r] = 0.01f * (threadIdx.x + r);
    for (int it = 0; it < iters; ++it) {
        if (role == 1 || role == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        if (role == 2 || role == 3) {
            // 16 x (fma, exp2, add) + 8 cvt-like packs = 56 vector instructions (14 per MFMA of the other role)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float p = __builtin_amdgcn_exp2f(fmaf(v[r], c, -1.0f)); v[r] = p; sum += p; }
        }
        if (role == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x402, 12, 0); }
        }
    }
    float s = sum;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 16; ++r) s += v[r];
    if (s == 123.456f) sink[0] = s;
}

template <int RA, int RB>
float run(const char* name, float* sink) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, 100, sink, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, iters, sink, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f ns/iter\n", name, ms, ms * 1e6 / iters);
    return ms;
}

int main() {
    float* sink; hipMalloc(&sink, 64);
    run<1, 0>("MFMA wave alone (4 MFMA / iter)", sink);
    run<2, 0>("VALU wave alone (48 VALU / iter)", sink);
    run<1, 1>("MFMA + MFMA", sink);
    run<2, 2>("VALU + VALU", sink);
    run<1, 2>("MFMA wave + VALU wave (same SIMD)", sink);
    run<3, 0>("one wave, interleaved MFMA and VALU", sink);
    run<3, 3>("two waves, each interleaved", sink);
    return 0;
}
