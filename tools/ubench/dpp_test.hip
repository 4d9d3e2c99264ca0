// row_shr:1 / row_shl:1 through __builtin_amdgcn_update_dpp on every element of a 16-float vector (hipcc check)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ void k(const float* in, float* out) {
    f32x16_t y;
    for (int r = 0; r < 16; ++r) y[r] = in[threadIdx.x * 16 + r];
    float o[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = y[r];
        o[r] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
    }
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = o[r];
}
int main() {
    float h[64 * 16], o[64 * 16];
    for (int i = 0; i < 64 * 16; ++i) h[i] = (float)i;
    float *di, *dout;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(h));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const float want = (l % 16 == 0) ? 0.f : h[(l - 1) * 16 + r];
        if (o[l * 16 + r] != want) { if (bad < 8) printf("lane %d r %d got %g want %g\n", l, r, o[l * 16 + r], want); ++bad; }
    }
    printf("bad %d\n", bad);
    return 0;
}
