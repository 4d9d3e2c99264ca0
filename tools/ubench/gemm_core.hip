// Micro-benchmark: dissect the 256 x BN LDS-DMA GEMM main loop of gemm_conv_glds.hip on gfx950.
// One workgroup (512 threads, 8 waves as 4 x 2) per CU runs `iters` K tiles (BK = 64) with pieces switched off:
//   DMA  : issue the (4 + BN/64) LDS-DMA loads per wave per K tile (sources cache-resident: shared 1 MB panels)
//   BAR  : counted vmcnt + one raw s_barrier per K tile
//   LDSR : ds_read_b128 fragment reads each 16-wide K step (otherwise fragments are loaded once)
//   MFMA : the 2 x NB v_mfma_f32_32x32x16_bf16 per K step
// Reported: time per K tile per CU and the TFLOP/s the chip would reach at that rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BN, int STG, bool DMA, bool BAR, bool LDSR, bool MFMA, int VAR = 0>
__global__ __launch_bounds__(512) void core(const char* __restrict__ A, const char* __restrict__ W, int iters, int kwrap,
                                            float* sink) {
    constexpr int NB = BN / 64;
    constexpr int A_BYTES = 256 * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = 4, B_IT = BN / 64, LOADS = A_IT + B_IT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    const int srow = lane >> 3, pchunk = lane & 7;
    const int K = kwrap * 64;
    const char* a_ptr[A_IT];
    const char* b_ptr[B_IT];
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
        const int r = (q * 8 + wave) * 8 + srow;
        a_ptr[q] = A + ((size_t)r * K + (pchunk ^ ((r >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
        const int r = (q * 8 + wave) * 8 + srow;
        b_ptr[q] = W + ((size_t)r * K + (pchunk ^ ((r >> 1) & 7)) * 8) * 2;
    }
    int i_kt = 0, i_stage = 0;
    auto issue = [&]() __attribute__((always_inline)) {
        const int k0 = i_kt * 128;
        const unsigned sa = lds_base + i_stage * STAGE, sb = sa + A_BYTES;
#pragma unroll
        for (int q = 0; q < A_IT; ++q) glds16(a_ptr[q] + k0, sa + (q * 8 + wave) * 1024);
#pragma unroll
        for (int q = 0; q < B_IT; ++q) glds16(b_ptr[q] + k0, sb + (q * 8 + wave) * 1024);
        if (++i_stage >= STG) i_stage = 0;
        if (++i_kt >= kwrap) i_kt = 0;
    };
    f32x16_t acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // initialise LDS so fragment reads see defined data
    for (int i = tid; i < STG * STAGE / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    if (DMA) { issue(); if (STG >= 3) issue(); }
    bf16x8_t xf[2], wf[NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) xf[mb] = *reinterpret_cast<const bf16x8_t*>(smem + lds_off(wm * 64 + mb * 32 + fr, fh));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
        wf[nb] = *reinterpret_cast<const bf16x8_t*>(smem + A_BYTES + lds_off(wn * 32 * NB + nb * 32 + fr, fh));
    int stage = 0;
    constexpr bool SPREAD = (VAR & 1) != 0, NOPRIO = (VAR & 2) != 0, FRAGDB = (VAR & 4) != 0, NOBAR = (VAR & 8) != 0;
    constexpr bool VLOAD = (VAR & 16) != 0;
    bf16x8_t xg[2], wg[NB];      // second fragment set (FRAGDB)
    auto load_frags = [&](bf16x8_t* x, bf16x8_t* w_, const char* sa, const char* sb, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            x[mb] = *reinterpret_cast<const bf16x8_t*>(sa + lds_off(wm * 64 + mb * 32 + fr, kk * 2 + fh));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            w_[nb] = *reinterpret_cast<const bf16x8_t*>(sb + lds_off(wn * 32 * NB + nb * 32 + fr, kk * 2 + fh));
    };
    auto mfmas = [&](bf16x8_t* x, bf16x8_t* w_) __attribute__((always_inline)) {
        if (!NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_[nb], x[mb], acc[mb][nb], 0, 0, 0);
        if (!NOPRIO) __builtin_amdgcn_s_setprio(0);
    };
    for (int g = 0; g < iters; ++g) {
        if (DMA) { if (STG >= 3) wait_vm<LOADS>(); else wait_vm<0>(); }
        if (BAR && !NOBAR) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (DMA && !SPREAD && !VLOAD) issue();
        if (DMA && VLOAD) {
#pragma unroll
            for (int q = 0; q < A_IT; ++q) { auto v = *reinterpret_cast<const bf16x8_t*>(a_ptr[q] + i_kt * 128); asm volatile("" ::"v"(v)); }
#pragma unroll
            for (int q = 0; q < B_IT; ++q) { auto v = *reinterpret_cast<const bf16x8_t*>(b_ptr[q] + i_kt * 128); asm volatile("" ::"v"(v)); }
            if (++i_kt >= kwrap) i_kt = 0;
        }
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        if (LDSR && FRAGDB) load_frags(xf, wf, sa, sb, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (DMA && SPREAD) {
                const int k0 = i_kt * 128;
                const unsigned sa2 = lds_base + i_stage * STAGE, sb2 = sa2 + A_BYTES;
                glds16(a_ptr[kk] + k0, sa2 + (kk * 8 + wave) * 1024);
#pragma unroll
                for (int q = 0; q < B_IT; ++q) if ((q & 3) == kk) glds16(b_ptr[q] + k0, sb2 + (q * 8 + wave) * 1024);
                if (kk == 3) { if (++i_stage >= STG) i_stage = 0; if (++i_kt >= kwrap) i_kt = 0; }
            }
            if (LDSR && FRAGDB) {
                bf16x8_t* cx = (kk & 1) ? xg : xf; bf16x8_t* cw = (kk & 1) ? wg : wf;
                bf16x8_t* nx = (kk & 1) ? xf : xg; bf16x8_t* nw = (kk & 1) ? wf : wg;
                if (kk < 3) load_frags(nx, nw, sa, sb, kk + 1);
                if (MFMA) mfmas(cx, cw);
            } else {
                if (LDSR) load_frags(xf, wf, sa, sb, kk);
                if (MFMA) mfmas(xf, wf);
                else if (LDSR) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) asm volatile("" ::"v"(wf[nb]));
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) asm volatile("" ::"v"(xf[mb]));
                }
            }
        }
        if (++stage >= STG) stage = 0;
    }
    wait_vm<0>();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
}

template <int BN, int STG, bool DMA, bool BAR, bool LDSR, bool MFMA, int VAR = 0>
void run(const char* name, const char* A, const char* W, float* sink) {
    const int iters = 4000, grid = 256, kwrap = 16;
    const size_t lds = (size_t)STG * (256 * 128 + BN * 128);
    auto fn = &core<BN, STG, DMA, BAR, LDSR, MFMA, VAR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, 0, A, W, 200, kwrap, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, 0, A, W, iters, kwrap, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 256 * BN * 64 * (double)iters * grid;
    printf("BN=%3d STG=%d %-28s %8.3f ms  %6.0f ns/Ktile  %7.0f TF/s-equivalent\n", BN, STG, name, ms, ms * 1e6 / iters,
           flop / ms / 1e9);
    fflush(stdout);
}


// ---- one wave per SIMD: 256 threads (4 waves as 2 x 2), wave tile 128 x (BN/2): 4 x NB MFMA blocks, 512-register budget.
// Fragments double-buffered across the 16-wide K steps; LDS-DMA spread (each wave issues 2x the pieces).
template <int BN, int STG, bool DMA, bool FRAGDB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void core1(const char* __restrict__ A, const char* __restrict__ W, int iters, int kwrap, float* sink) {
    constexpr int NB = BN / 64;
    constexpr int A_BYTES = 256 * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = 8, B_IT = BN / 32, LOADS = A_IT + B_IT;          // per wave: 256 rows / (4 waves * 8 rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, fr = lane & 31, fh = lane >> 5;
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    const int srow = lane >> 3, pchunk = lane & 7;
    const int K = kwrap * 64;
    const char* a_ptr[A_IT];
    const char* b_ptr[B_IT];
#pragma unroll
    for (int q = 0; q < A_IT; ++q) {
        const int r = (q * 4 + wave) * 8 + srow;
        a_ptr[q] = A + ((size_t)r * K + (pchunk ^ ((r >> 1) & 7)) * 8) * 2;
    }
#pragma unroll
    for (int q = 0; q < B_IT; ++q) {
        const int r = (q * 4 + wave) * 8 + srow;
        b_ptr[q] = W + ((size_t)r * K + (pchunk ^ ((r >> 1) & 7)) * 8) * 2;
    }
    int i_kt = 0, i_stage = 0;
    auto issue_part = [&](int part) __attribute__((always_inline)) {        // part 0..3
        const int k0 = i_kt * 128;
        const unsigned sa = lds_base + i_stage * STAGE, sb = sa + A_BYTES;
#pragma unroll
        for (int q = 0; q < A_IT; ++q) if ((q & 3) == part) glds16(a_ptr[q] + k0, sa + (q * 4 + wave) * 1024);
#pragma unroll
        for (int q = 0; q < B_IT; ++q) if ((q & 3) == part) glds16(b_ptr[q] + k0, sb + (q * 4 + wave) * 1024);
        if (part == 3) { if (++i_stage >= STG) i_stage = 0; if (++i_kt >= kwrap) i_kt = 0; }
    };
    f32x16_t acc[4][NB];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int i = tid; i < STG * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u;
    __syncthreads();
    if (DMA) { for (int p = 0; p < 4; ++p) issue_part(p); if (STG >= 3) for (int p = 0; p < 4; ++p) issue_part(p); }
    bf16x8_t xf[2][4], wf[2][NB];
    auto load_frags = [&](int buf, const char* sa, const char* sb, int kk) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            xf[buf][mb] = *reinterpret_cast<const bf16x8_t*>(sa + lds_off(wm * 128 + mb * 32 + fr, kk * 2 + fh));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            wf[buf][nb] = *reinterpret_cast<const bf16x8_t*>(sb + lds_off(wn * 32 * NB + nb * 32 + fr, kk * 2 + fh));
    };
    int stage = 0;
    for (int g = 0; g < iters; ++g) {
        if (DMA) { if (STG >= 3) wait_vm<LOADS>(); else wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        if (FRAGDB) load_frags(0, sa, sb, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (DMA) issue_part(kk);
            const int cur = FRAGDB ? (kk & 1) : 0;
            if (FRAGDB) { if (kk < 3) load_frags(cur ^ 1, sa, sb, kk + 1); }
            else load_frags(0, sa, sb, kk);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[cur][nb], xf[cur][mb], acc[mb][nb], 0, 0, 0);
        }
        if (++stage >= STG) stage = 0;
    }
    wait_vm<0>();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 123.456f) sink[0] = s;
}

template <int BN, int STG, bool DMA, bool FRAGDB>
void run1(const char* name, const char* A, const char* W, float* sink) {
    const int iters = 4000, grid = 256, kwrap = 16;
    const size_t lds = (size_t)STG * (256 * 128 + BN * 128);
    auto fn = &core1<BN, STG, DMA, FRAGDB>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, A, W, 200, kwrap, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, 0, A, W, iters, kwrap, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 256 * BN * 64 * (double)iters * grid;
    printf("1 wave/SIMD BN=%3d STG=%d %-22s %8.3f ms  %6.0f ns/Ktile  %7.0f TF/s-equivalent\n", BN, STG, name, ms,
           ms * 1e6 / iters, flop / ms / 1e9);
    fflush(stdout);
}

template <int BN, int STG>
void suite(const char* A, const char* W, float* sink) {
    run<BN, STG, true, true, true, true>("full", A, W, sink);
    run<BN, STG, true, true, true, true, 2>("full noprio", A, W, sink);
    run<BN, STG, true, true, true, true, 3>("full noprio spread", A, W, sink);
    run<BN, STG, true, true, true, true, 6>("full noprio fragdb", A, W, sink);
    run<BN, STG, true, true, true, true, 7>("full noprio spread fragdb", A, W, sink);
    run<BN, STG, true, true, true, true, 5>("full spread fragdb (prio)", A, W, sink);
    run<BN, STG, false, true, true, true>("no DMA", A, W, sink);
    run<BN, STG, false, true, true, true, 2>("no DMA noprio", A, W, sink);
    run<BN, STG, false, true, true, true, 6>("no DMA noprio fragdb", A, W, sink);
    run<BN, STG, false, false, true, true, 6>("no DMA/bar noprio fragdb", A, W, sink);
    run<BN, STG, false, false, false, true>("MFMA only", A, W, sink);
    run<BN, STG, true, true, false, true, 2>("DMA+barrier+MFMA noprio", A, W, sink);
    run<BN, STG, true, true, false, true, 3>("DMA spread+bar+MFMA noprio", A, W, sink);
    run<BN, STG, true, true, false, true, 18>("VGPR loads+bar+MFMA noprio", A, W, sink);
}

int main() {
    char *A, *W; float* sink;
    hipMalloc(&A, 8 << 20); hipMemset(A, 0x3c, 8 << 20);
    hipMalloc(&W, 8 << 20); hipMemset(W, 0x3c, 8 << 20);
    hipMalloc(&sink, 64);
    run1<320, 2, true, true>("full fragdb", A, W, sink);
    run1<320, 2, true, false>("full", A, W, sink);
    run1<320, 2, false, true>("no DMA fragdb", A, W, sink);
    run1<256, 2, true, true>("full fragdb", A, W, sink);
    run1<256, 2, false, true>("no DMA fragdb", A, W, sink);
    run1<128, 3, true, true>("full fragdb", A, W, sink);
    run<320, 2, true, true, true, true, 3>("2 waves/SIMD noprio spread", A, W, sink);
    run<256, 2, true, true, true, true, 3>("2 waves/SIMD noprio spread", A, W, sink);
    run<128, 3, true, true, true, true, 3>("2 waves/SIMD noprio spread", A, W, sink);
    if (getenv("UBENCH_FULL")) suite<320, 2>(A, W, sink);
    if (getenv("UBENCH_FULL")) { suite<256, 2>(A, W, sink); suite<128, 3>(A, W, sink); }
    return 0;
}
