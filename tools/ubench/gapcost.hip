// Micro-benchmark: what does one MFMA gap cost a wave that is ALONE on its SIMD (the structure of flash_pipe.hip), by what
// the gap carries? 256 threads = one wave per SIMD, every instruction an `asm volatile` statement in source order. A
// "step" = 16 gaps: 8 score MFMAs (two chains of 4 into architectural VGPRs, first one with a separate C operand) and 8
// output MFMAs (four chains of 2 in the accumulator file); the vector fillers of a gap read the OTHER score buffer.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/gapcost tools/ubench/gapcost.hip && tools/ubench/gapcost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef const volatile __attribute__((address_space(3))) bf16x8_t lds_vfrag_t;
typedef __attribute__((address_space(3))) char lds_char_t;

enum { F_EXP = 1, F_ADD = 2, F_MAX = 4, F_LDS = 8, F_SACC = 16, F_CVT = 32, F_NOMFMA = 64, F_EXP1 = 128, F_ONLYS = 256,
       F_ONLYO = 512, F_NOP = 1024, F_FMA = 2048, F_LDSCONF = 4096, F_LDSNEAR = 8192 };

#define G_EXP(dst, src) asm volatile("v_exp_f32 %0, %1" : "=v"(dst) : "v"(src))
#define G_ADD(acc, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
#define G_FMA(acc, x) asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc) : "v"(x))
#define G_CVT(dst, lo, hi) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dst) : "v"(lo), "v"(hi))
#define G_MAX3(acc, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
#define G_MFMA_S0(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c))
#define G_MFMA_S(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
#define G_MFMA_A(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))

template <int F>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k(int iters, float* sink, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16_t S[2][2], O[2][2], SA[2], NM[2];
    bf16x8_t Q[2][4], Kf[4], Vf[2][2];
    u32x4_t P[2][2];
    float la = 0.f, lb = 0.f, mxa = -3e38f, mxb = -3e38f;
    for (int x = 0; x < 2; ++x)
        for (int r = 0; r < 16; ++r) {
            S[0][x][r] = -0.01f * (lane + r); S[1][x][r] = -0.02f * (lane + r); O[x][0][r] = 0.f; O[x][1][r] = 0.f;
            SA[x][r] = 0.f; NM[x][r] = -1.0f - x;
        }
    for (int x = 0; x < 2; ++x)
        for (int kk = 0; kk < 4; ++kk) {
            for (int e = 0; e < 8; ++e) Q[x][kk][e] = (short)(0x3c00 + lane + e + kk);
            asm volatile("" : "+a"(Q[x][kk]));
        }
    for (int kk = 0; kk < 4; ++kk) for (int e = 0; e < 8; ++e) Kf[kk][e] = (short)(0x3b00 + lane * 3 + e);
    for (int a = 0; a < 2; ++a) for (int b2 = 0; b2 < 2; ++b2) for (int e = 0; e < 8; ++e) Vf[a][b2][e] = (short)(0x3a00 + lane + e);
    for (int x = 0; x < 2; ++x) for (int ks = 0; ks < 2; ++ks) for (int e = 0; e < 4; ++e) P[x][ks][e] = 0x3c003c00u + lane;
    // F_LDS: conflict-free fragment reads (64 lanes x 16 consecutive bytes); F_LDSCONF: 128-byte lane stride (16-way conflicts)
    const lds_char_t* lbase = (const lds_char_t*)smem + ((F & F_LDSCONF) ? (lane & 31) * 128 + (lane >> 5) * 16 : lane * 16);
    unsigned long long t0 = 0;
    for (int it = -2; it < iters; ++it) {
        if (it == 0) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            f32x16_t (&Sw)[2] = S[par];
            f32x16_t (&Sr)[2] = S[par ^ 1];
            float pa[16], pb[16];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if constexpr ((F & F_LDS) && !(F & F_LDSNEAR)) {      // the request schedule of flash_pipe.hip: >= 4 gaps ahead of the use
                    if (g == 0 || g == 1 || g == 10 || g == 12) Kf[(g == 0) ? 2 : (g == 1) ? 3 : (g == 10) ? 0 : 1] = *(lds_vfrag_t*)(lbase + g * 4096 % 16384);
                    if (g == 2 || g == 3 || g == 5 || g == 7) Vf[(g >> 2) & 1][g & 1] = *(lds_vfrag_t*)(lbase + 16384 + g * 2048);
                }
                if constexpr ((F & F_LDS) && (F & F_LDSNEAR)) {       // every fragment requested 2 gaps ahead of its use
                    if (g == 14 || g == 0 || g == 2 || g == 4) Kf[((g + 2) & 15) >> 1] = *(lds_vfrag_t*)(lbase + g * 1024);
                    if (g == 6 || g == 8 || g == 10 || g == 12) Vf[((g - 6) >> 2) & 1][((g - 6) >> 1) & 1] = *(lds_vfrag_t*)(lbase + 16384 + g * 1024);
                }
                if constexpr (!(F & F_NOMFMA)) {
                    if (g < 8) {
                        if constexpr (!(F & F_ONLYO)) {
                            const int kk = g >> 1, x = g & 1;
                            if constexpr (F & F_SACC) { G_MFMA_A(SA[x], Kf[kk], Kf[kk]); }
                            else if (kk == 0) G_MFMA_S0(Sw[x], Kf[0], Q[x][0], NM[x]);
                            else G_MFMA_S(Sw[x], Kf[kk], Q[x][kk]);
                        } else {
                            const int i = g, ks = i >> 2, db = (i >> 1) & 1, x = i & 1;
                            G_MFMA_A(O[x][db], Vf[ks][db], P[x][ks]);
                        }
                    } else {
                        if constexpr (!(F & F_ONLYS)) {
                            const int i = g - 8, ks = i >> 2, db = (i >> 1) & 1, x = i & 1;
                            G_MFMA_A(O[x][db], Vf[ks][db], P[x][ks]);
                        } else {
                            const int kk = (g - 8) >> 1, x = g & 1;
                            G_MFMA_S(Sw[x], Kf[kk], Q[x][kk]);
                        }
                    }
                }
                const int ex = g >> 3, ei = g & 7;
                if constexpr (F & F_EXP) G_EXP(pa[g], Sr[ex][2 * ei]);
                if constexpr (F & F_MAX) { if (g >= 10) G_MAX3(mxa, Sr[0][g - 10], Sr[0][g - 9]); }
                if constexpr ((F & F_EXP) && !(F & F_EXP1)) G_EXP(pb[g], Sr[ex][2 * ei + 1]);
                if constexpr (F & F_EXP1) pb[g] = Sr[ex][2 * ei + 1];
                if constexpr (F & F_MAX) { if (g >= 10) G_MAX3(mxb, Sr[1][g - 10], Sr[1][g - 9]); }
                if constexpr (F & F_NOP) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0");
                if (g > 0) {
                    if constexpr (F & F_ADD) { G_ADD(la, pa[g - 1]); G_ADD(lb, pb[g - 1]); }
                    if constexpr (F & F_FMA) { G_FMA(la, Sr[ex][2 * ei]); G_FMA(lb, Sr[ex][2 * ei + 1]); G_FMA(mxa, Sr[ex ^ 1][2 * ei]); G_FMA(mxb, Sr[ex ^ 1][2 * ei + 1]); }
                    if constexpr (F & F_CVT) {
                        unsigned w;
                        G_CVT(w, pa[g - 1], pb[g - 1]);
                        P[(g - 1) >> 3][((g - 1) & 7) >> 2][(g - 1) & 3] = w;
                    }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    float s = la + lb + mxa + mxb;
    for (int x = 0; x < 2; ++x) for (int r = 0; r < 16; ++r) s += O[x][0][r] + O[x][1][r] + S[0][x][r] + S[1][x][r] + SA[x][r];
    if (s == 123.456f) sink[0] = s;
    if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int F>
void run(const char* name, float* sink, unsigned long long* clk) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F>), dim3(256), dim3(256), 0, 0, 200, sink, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F>), dim3(256), dim3(256), 0, 0, iters, sink, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    const double gaps = (double)iters * 32;
    printf("%-64s %7.1f clk/gap  %6.2f ns/gap  (%.2f GHz)\n", name, c / gaps, ms * 1e6 / gaps, c / (ms * 1e6));
}

int main() {
    float* sink; hipMalloc(&sink, 64);
    unsigned long long* clk; hipMalloc(&clk, 64);
    run<0>("MFMA only (8 S into VGPR + 8 O into AGPR per step)", sink, clk);
    run<F_ONLYS>("MFMA only, all 16 into VGPR chains", sink, clk);
    run<F_ONLYO>("MFMA only, all 16 into AGPR chains", sink, clk);
    run<F_SACC>("MFMA only, score chains in AGPR too", sink, clk);
    run<F_EXP | F_EXP1>("+ 1 exp", sink, clk);
    run<F_EXP>("+ 2 exp", sink, clk);
    run<F_EXP | F_CVT>("+ 2 exp + cvt", sink, clk);
    run<F_EXP | F_CVT | F_ADD>("+ 2 exp + cvt + 2 add", sink, clk);
    run<F_EXP | F_CVT | F_ADD | F_MAX>("+ 2 exp + cvt + 2 add + max3 (gaps 10-15: 2)", sink, clk);
    run<F_EXP | F_CVT | F_SACC>("+ 2 exp + cvt, score chains in AGPR", sink, clk);
    run<F_EXP | F_CVT | F_ADD | F_SACC>("+ 2 exp + cvt + 2 add, score chains in AGPR", sink, clk);
    run<F_EXP | F_CVT | F_LDS>("+ 2 exp + cvt + 12 LDS fragment reads / 16 gaps", sink, clk);
    run<F_LDS>("MFMA + LDS reads only", sink, clk);
    run<F_LDS | F_LDSCONF>("MFMA + LDS reads only, 16-way bank conflicts", sink, clk);
    run<F_LDS | F_LDSNEAR>("MFMA + LDS reads only, requested 2 gaps ahead", sink, clk);
    run<F_EXP | F_CVT | F_ADD | F_LDS>("+ 2 exp + cvt + 2 add + LDS reads", sink, clk);
    run<F_FMA>("+ 4 fma (reading score registers)", sink, clk);
    run<F_NOP>("+ 3 s_nop 0", sink, clk);
    run<F_NOMFMA | F_EXP | F_CVT>("no MFMA: 2 exp + cvt", sink, clk);
    run<F_NOMFMA | F_EXP | F_CVT | F_ADD>("no MFMA: 2 exp + cvt + 2 add", sink, clk);
    run<F_NOMFMA | F_FMA>("no MFMA: 4 fma", sink, clk);
    return 0;
}
