// bf16 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M, N] = epilogue( gather(A)[M, K] * W[N, K]^T )
//
// One kernel template serves every dense contraction of the DynamiCrafter denoising path:
//   mode 0  plain GEMM          nn.Linear / 1x1 conv        (reference: lvdm/modules/attention.py:53-57,269,290,418,438)
//   mode 1  conv2d 3x3          stride 1|2, pad (sym / AE-asymmetric), optional fused nearest x2 upsample of the source
//                               (reference: openaimodel3d.py:68,96,103,151-180; ae_modules.py:96-106,117-126)
//   mode 2  temporal conv 3x1x1 zero-padded in time          (reference: openaimodel3d.py:255-266)
// Activations are channels-last rows [rows, C] (row = ((b*T + t)*H + y)*W + x), weights are [N][K] with
// K = taps*Cin ordered (tap, ci), i.e. a 3x3 weight is stored [Cout][kh][kw][Cin].
//
// Tiling: 128 x BN x 64 per workgroup, 4 waves as 2(M) x 2(N), v_mfma_f32_32x32x16_bf16 with the operands swapped
// (A-operand = weight fragment, B-operand = activation fragment) so that a lane ends up holding 4 consecutive
// output channels of one output row -> packed 8-byte LDS writes in the epilogue and coalesced row stores.
// Global -> register -> LDS staging with one barrier per K tile; the next tile's global loads are issued before
// the MFMA block of the current one (issue-early / write-late). LDS tiles are XOR-swizzled for conflict-free
// ds_read_b128 fragment reads.
#include "dc_common.h"
#include "dcrafter_hip.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 128-byte rows, 16-byte chunks; XOR the chunk with bits of the row so that 16 rows (distinct mod 16)
    // reading the same logical chunk hit 16 distinct 16-byte slots of the 256-byte bank row.
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int BN, bool GEGLU>
__global__ __launch_bounds__(NTHREADS) void gemm_conv_kernel(const DcGemmParams p) {
    constexpr int NB = BN / 64;              // 32-wide n-blocks per wave
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int B_ITERS = BN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int n_out = GEGLU ? (p.N >> 1) : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_m * tiles_n;
    const int swz = xcd_remap(blockIdx.x, nwg);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BNOUT;

    // ---- per-thread staging coordinates ----
    const int chunk = tid & 7;        // 16-byte chunk inside the 64-wide K slice
    const int srow = tid >> 3;        // 0..31
    // A rows handled by this thread: srow + 32*i
    int a_base[4];    // mode 0: row offset (elements) or -1; mode 1: n*IH*IW ; mode 2: row index
    int a_y[4], a_x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        if (m >= p.M) { a_base[i] = -1; a_y[i] = 0; a_x[i] = 0; continue; }
        if (p.mode == 0) {
            a_base[i] = m; a_y[i] = 0; a_x[i] = 0;
        } else if (p.mode == 1) {
            const int ohw = p.OH * p.OW;
            const int n = m / ohw;
            const int rem = m - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            a_base[i] = n * p.IH * p.IW;
            a_y[i] = oy * p.stride - p.pad;
            a_x[i] = ox * p.stride - p.pad;
        } else {
            a_base[i] = m;
            a_y[i] = (m / p.HW) % p.T;   // frame index t
            a_x[i] = 0;
        }
    }
    // B rows handled by this thread
    const bf16_t* b_ptr[B_ITERS];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
        const int r = srow + 32 * i;     // row inside the B tile
        int wrow;
        if (GEGLU) {
            wrow = (r < BN / 2) ? (n0 + r) : ((p.N >> 1) + n0 + (r - BN / 2));
        } else {
            wrow = n0 + r;
        }
        b_ptr[i] = p.W + (size_t)wrow * p.K + chunk * 8;   // W is zero-padded to a multiple of the tile in N
    }

    uint4 a_reg[4], b_reg[B_ITERS];
    const int nk = p.K / BK;

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        int tap = 0, ci0 = k0;
        if (p.mode != 0) { tap = k0 / p.Cin; ci0 = k0 - tap * p.Cin; }
        int dy = 0, dx = 0;
        if (p.mode == 1) { dy = tap / 3; dx = tap - dy * 3; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (a_base[i] >= 0) {
                if (p.mode == 0) {
                    v = *reinterpret_cast<const uint4*>(p.A + (size_t)a_base[i] * p.lda + k0 + chunk * 8);
                } else if (p.mode == 1) {
                    const int iy = a_y[i] + dy, ix = a_x[i] + dx;
                    const int eh = p.IH << p.ups, ew = p.IW << p.ups;
                    if (iy >= 0 && iy < eh && ix >= 0 && ix < ew) {
                        const int src = a_base[i] + (iy >> p.ups) * p.IW + (ix >> p.ups);
                        v = *reinterpret_cast<const uint4*>(p.A + (size_t)src * p.lda + ci0 + chunk * 8);
                    }
                } else {
                    const int tt = a_y[i] + tap - 1;
                    if (tt >= 0 && tt < p.T) {
                        const int src = a_base[i] + (tap - 1) * p.HW;
                        v = *reinterpret_cast<const uint4*>(p.A + (size_t)src * p.lda + ci0 + chunk * 8);
                    }
                }
            }
            a_reg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) b_reg[i] = *reinterpret_cast<const uint4*>(b_ptr[i] + k0);
    };

    auto store_tile = [&](int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sa + lds_off(srow + 32 * i, chunk)) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<uint4*>(sb + lds_off(srow + 32 * i, chunk)) = b_reg[i];
    };

    f32x16_t acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8_t xf[2], wf[NB];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                xf[mb] = *reinterpret_cast<const bf16x8_t*>(sa + lds_off(wm * 64 + mb * 32 + fr, kk * 2 + fh));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                int brow;
                if (GEGLU) brow = nb * (BN / 2) + wn * 32 + fr;      // nb 0 = value half, nb 1 = gate half
                else brow = wn * (32 * NB) + nb * 32 + fr;
                wf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + lds_off(brow, kk * 2 + fh));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // lane holds, for output row m = wm*64 + mb*32 + (lane&31), the channels
    //   nloc = nb*32 + 8*q + 4*(lane>>5) + {0..3}   (q = 0..3) in acc[mb][nb][4q..4q+3]
    const bool out_f32 = (p.flags & DC_GEMM_OUT_F32) != 0;
    constexpr int NOUTB = GEGLU ? 1 : NB;   // output n-blocks per wave
    constexpr int WN_OUT = 32 * NOUTB;      // output columns per wave
    constexpr int CS_LD = BNOUT * 2 + 8;    // bytes per row of the staging tile (padded: conflict-free b64 writes)

    if (out_f32) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int m = m0 + wm * 64 + mb * 32 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int nb = 0; nb < NOUTB; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + wn * WN_OUT + nb * 32 + 8 * q + 4 * fh;
                    if (n >= n_out) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[mb][nb][4 * q + e];
                        if (p.bias) x += p.bias[n + e];
                        if constexpr (GEGLU) {
                            float g = acc[mb][NB - 1][4 * q + e];
                            if (p.bias) g += p.bias[(p.N >> 1) + n + e];
                            x = x * gelu_erf_f(g);
                        }
                        v[e] = x * p.alpha;
                    }
                    float* dst = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
        }
        return;
    }

    char* cs = smem;   // reuse the pipeline buffers (all waves passed the final barrier of the K loop)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int mloc = wm * 64 + mb * 32 + fr;
        const int m = m0 + mloc;
        const float* rv = nullptr;
        if (p.rowvec && m < p.M) rv = p.rowvec + (size_t)(m / p.rows_per_vec) * p.rowvec_ld;
#pragma unroll
        for (int nb = 0; nb < NOUTB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nloc = wn * WN_OUT + nb * 32 + 8 * q + 4 * fh;
                const int n = n0 + nloc;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[mb][nb][4 * q + e];
                    if (n + e < n_out) {
                        if (p.bias) x += p.bias[n + e];
                        if constexpr (GEGLU) {
                            float g = acc[mb][NB - 1][4 * q + e];
                            if (p.bias) g += p.bias[(p.N >> 1) + n + e];
                            x = x * gelu_erf_f(g);
                        }
                        if (rv) x += rv[n + e];
                    }
                    v[e] = x * p.alpha;
                }
                uint2 pk;
                pk.x = pack_bf2(v[0], v[1]);
                pk.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(cs + mloc * CS_LD + nloc * 2) = pk;
            }
    }
    __syncthreads();
    constexpr int UPR = BNOUT / 4;            // 8-byte units per row
    constexpr int UNITS = BM * UPR;
    bf16_t* cptr = reinterpret_cast<bf16_t*>(p.C);
#pragma unroll
    for (int i = 0; i < UNITS / NTHREADS; ++i) {
        const int u = tid + NTHREADS * i;
        const int r = u / UPR;
        const int c = (u - r * UPR) * 4;
        const int m = m0 + r, n = n0 + c;
        if (m >= p.M || n >= n_out) continue;
        uint2 pk = *reinterpret_cast<const uint2*>(cs + r * CS_LD + c * 2);
        if (p.residual) {
            const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.ldr + n);
            float a0 = __uint_as_float(pk.x << 16) + __uint_as_float(rr.x << 16);
            float a1 = __uint_as_float(pk.x & 0xffff0000u) + __uint_as_float(rr.x & 0xffff0000u);
            float a2 = __uint_as_float(pk.y << 16) + __uint_as_float(rr.y << 16);
            float a3 = __uint_as_float(pk.y & 0xffff0000u) + __uint_as_float(rr.y & 0xffff0000u);
            pk.x = pack_bf2(a0, a1);
            pk.y = pack_bf2(a2, a3);
        }
        *reinterpret_cast<uint2*>(cptr + (size_t)m * p.ldc + n) = pk;
    }
}

template <int BN, bool GEGLU>
int launch(const DcGemmParams& p, hipStream_t stream) {
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    const int n_out = GEGLU ? p.N / 2 : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + BM - 1) / BM;
    const size_t lds = 2 * (BM * BK * 2 + BN * BK * 2);
    hipLaunchKernelGGL((gemm_conv_kernel<BN, GEGLU>), dim3(tiles_m * tiles_n), dim3(NTHREADS), lds, stream, p);
    DC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int dc_gemm_conv(const DcGemmParams* pp, void* stream_) {
    const DcGemmParams& p = *pp;
    hipStream_t stream = (hipStream_t)stream_;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return DC_ERR_SHAPE;
    if (p.K % BK != 0 || p.N % 4 != 0 || p.lda % 8 != 0 || p.ldc % 4 != 0) return DC_ERR_SHAPE;
    if (p.mode != 0 && (p.Cin % BK != 0)) return DC_ERR_SHAPE;
    if (p.mode == 1 && p.K != 9 * p.Cin) return DC_ERR_SHAPE;
    if (p.mode == 2 && p.K != 3 * p.Cin) return DC_ERR_SHAPE;
    if (p.residual && (p.ldr % 4 != 0)) return DC_ERR_SHAPE;
    const bool geglu = (p.flags & DC_GEMM_GEGLU) != 0;
    if (geglu) {
        if ((p.N / 2) % 64 != 0) return DC_ERR_SHAPE;
        if (p.n_pad < p.N) return DC_ERR_SHAPE;
        return launch<128, true>(p, stream);
    }
    // Tile choice: 128-wide N tiles unless that wastes >15% of the MFMA work on padding (N = 320 -> 64-wide).
    const int t128 = (p.N + 127) / 128 * 128;
    const bool use64 = (p.N <= 64) || ((float)t128 / (float)p.N > 1.15f);
    if (use64) {
        if (p.n_pad < (p.N + 63) / 64 * 64) return DC_ERR_SHAPE;
        return launch<64, false>(p, stream);
    }
    if (p.n_pad < t128) return DC_ERR_SHAPE;
    return launch<128, false>(p, stream);
}
