// Micro-benchmark: per-CU rate of staging L2-resident data into LDS on gfx950
//   mode 0: LDS-DMA (global_load_lds_dwordx4 via asm, as gemm_conv_glds.hip issues it)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only
// Each workgroup (512 threads) streams `iters` x 48 KB from a per-XCD-shared 2 MB region.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int MODE, int PER_WAVE>
__global__ __launch_bounds__(512) void k(const char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    u32x4 acc = {0, 0, 0, 0};
    size_t off = ((size_t)blockIdx.x * 49152) % region;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const size_t o = (off + (size_t)(j * 8 + wave) * 1024 + lane * 16) % region;
            if (MODE == 0) {
                glds16(src + o, lds_base + ((it & 1) * PER_WAVE * 8 + j * 8 + wave) * 1024);
            } else {
                u32x4 v = *reinterpret_cast<const u32x4*>(src + o);
                if (MODE == 1) *reinterpret_cast<u32x4*>(smem + ((it & 1) * PER_WAVE * 8 + j * 8 + wave) * 1024 + lane * 16) = v;
                else acc ^= v;
            }
        }
        if (MODE == 0) {
            if (it & 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
        }
        off = (off + 49152 * 257) % region;
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) acc = *reinterpret_cast<u32x4*>(smem + threadIdx.x * 16);
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

template <int MODE>
void run(const char* name, const char* src, size_t region, unsigned* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 49152);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, 6>), dim3(grid), dim3(512), 2 * 49152, 0, src, region, 100, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, 6>), dim3(grid), dim3(512), 2 * 49152, 0, src, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * 49152;
    printf("%-34s region %6.1f MB: %8.3f ms  %7.2f TB/s  %6.1f GB/s/CU  (%5.1f B/clk/CU @2.1GHz)\n", name, region / 1048576.0, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.1);
}

int main() {
    const size_t maxr = 1ull << 30;
    char* src; unsigned* sink;
    hipMalloc(&src, maxr); hipMemset(src, 1, maxr); hipMalloc(&sink, 64);
    for (size_t region : {size_t(2) << 20, size_t(16) << 20, size_t(128) << 20, size_t(1) << 30}) {
        run<0>("LDS-DMA glds16", src, region, sink);
        run<1>("global_load x4 + ds_write_b128", src, region, sink);
        run<2>("global_load x4 only", src, region, sink);
    }
    return 0;
}
