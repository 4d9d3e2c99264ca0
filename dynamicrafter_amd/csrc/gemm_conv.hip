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
// K = taps*Cin ordered (64-channel slice, tap, channel), i.e. a 3x3 weight is stored [Cout][Cin/64][kh*kw][64].
//
// Tiling: 128 x BN x 64 per workgroup, 4 waves as 2(M) x 2(N), v_mfma_f32_32x32x16_bf16 with the operands swapped
// (A-operand = weight fragment, B-operand = activation fragment) so that a lane ends up holding 4 consecutive
// output channels of one output row -> packed 8-byte LDS writes in the epilogue and coalesced row stores.
// Global -> register -> LDS staging with one barrier per K tile; the next tile's global loads are issued before
// the MFMA block of the current one (issue-early / write-late). LDS tiles are XOR-swizzled for conflict-free
// ds_read_b128 fragment reads.
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 128-byte rows, 16-byte chunks; XOR the chunk with bits of the row so that 16 rows (distinct mod 16)
    // reading the same logical chunk hit 16 distinct 16-byte slots of the 256-byte bank row.
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// 32 zero bytes in device memory: padded taps / tail rows load from here, so every global load of the main loop is
// unconditional (a branch around a load makes hipcc wait for it at the join: four serialised round trips per tile).
__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk[8];

template <int BN, bool GEGLU, int MODE>
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
    const bf16_t* const zero_ptr = reinterpret_cast<const bf16_t*>(g_zero_chunk);
    const bf16_t* a_ptr[4];   // MODE 0: row base + chunk (k0 added per tile); MODE 2: centre-tap row base + chunk
    int a_base[4];            // MODE 1: first row of the frame; -1 = row beyond M
    int a_y[4], a_x[4];       // MODE 1: top-left input coordinate of the window; MODE 2: frame index t
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + srow + 32 * i;
        const bool ok = m < p.M;
        a_base[i] = ok ? 0 : -1;
        a_y[i] = 0; a_x[i] = 0;
        a_ptr[i] = zero_ptr;
        if (MODE == 0) {
            if (ok) a_ptr[i] = p.A + (size_t)m * p.lda + chunk * 8;
        } else if (MODE == 1) {
            const int ohw = p.OH * p.OW;
            const int mm = ok ? m : 0;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            if (ok) a_base[i] = n * p.IH * p.IW;
            a_y[i] = oy * p.stride - p.pad;
            a_x[i] = ox * p.stride - p.pad;
        } else {
            if (ok) a_ptr[i] = p.A + (size_t)m * p.lda + chunk * 8;
            a_y[i] = ((ok ? m : 0) / p.HW) % p.T;   // frame index t
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

    u32x4_t a_reg[4], b_reg[B_ITERS];     // native vectors (HIP's uint4 wrapper struct defeats SROA -> scratch)
    const int nk = p.K / BK;

    auto load_tile = [&](int kt) __attribute__((always_inline)) {
        const int k0 = kt * BK;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16_t* src = (a_base[i] >= 0) ? a_ptr[i] + k0 : zero_ptr;
                a_reg[i] = *reinterpret_cast<const u32x4_t*>(src);
            }
        } else if (MODE == 1) {
            // K is ordered (64-channel slice, tap, channel): the 9 taps of one slice are consecutive K tiles, so
            // taps 2..9 re-read (shifted) rows that the first tap just pulled into L2
            const int cs = kt / 9;
            const int tap = kt - cs * 9;
            const int ci0 = cs * 64;
            const int dy = tap / 3, dx = tap - dy * 3;
            const int eh = p.IH << p.ups, ew = p.IW << p.ups;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int iy = a_y[i] + dy, ix = a_x[i] + dx;
                const bool ok = (a_base[i] >= 0) & (iy >= 0) & (iy < eh) & (ix >= 0) & (ix < ew);
                const int srcrow = a_base[i] + (iy >> p.ups) * p.IW + (ix >> p.ups);
                const bf16_t* src = ok ? p.A + (size_t)srcrow * p.lda + ci0 + chunk * 8 : zero_ptr;
                a_reg[i] = *reinterpret_cast<const u32x4_t*>(src);
            }
        } else {
            const int cs = kt / 3;
            const int tap = kt - cs * 3;
            const int ci0 = cs * 64;
            const long long shift = (long long)(tap - 1) * p.HW * p.lda + ci0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tt = a_y[i] + tap - 1;
                const bool ok = (a_base[i] >= 0) & (tt >= 0) & (tt < p.T);
                const bf16_t* src = ok ? a_ptr[i] + shift : zero_ptr;
                a_reg[i] = *reinterpret_cast<const u32x4_t*>(src);
            }
        }
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) b_reg[i] = *reinterpret_cast<const u32x4_t*>(b_ptr[i] + k0);
    };

    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4_t*>(sa + lds_off(srow + 32 * i, chunk)) = a_reg[i];
#pragma unroll
        for (int i = 0; i < B_ITERS; ++i) *reinterpret_cast<u32x4_t*>(sb + lds_off(srow + 32 * i, chunk)) = b_reg[i];
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
        // unconditional prefetch (the last iteration re-fetches its own tile into the idle buffer): keeping the
        // loads and the LDS writes out of branches lets the staging registers stay registers (a conditional
        // prefetch made hipcc spill them to scratch behind a vmcnt wait, serialising every tile on HBM latency)
        load_tile(kt + 1 < nk ? kt + 1 : kt);
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
        store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // Accumulators go through LDS as fp32, half a tile (64 rows) at a time, and come back row-major: every
    // thread then owns 4 consecutive channels of one row, so bias / embedding-add / GEGLU / residual are 16- or
    // 8-byte coalesced accesses and nothing is rounded before the final bf16 conversion.
    // Lane layout of acc[mb][nb]: row m = wm*64 + mb*32 + (lane&31); channels nb*32 + 8*q + 4*(lane>>5) + {0..3}
    // in registers 4q..4q+3.
    const bool out_f32 = (p.flags & DC_GEMM_OUT_F32) != 0;
    constexpr int CS_LD = BNOUT * 4 + 16;       // bytes per staged row (pad keeps b128 writes conflict-free)
    constexpr int HALF_ROWS = 64;
    constexpr int XG = GEGLU ? 2 : 1;            // value and gate planes
    constexpr int PLANE = HALF_ROWS * CS_LD;
    static_assert(XG * PLANE <= 2 * STAGE, "epilogue staging must fit the pipeline buffers");
    constexpr int UPR = BNOUT / 4;               // 4-channel units per row
    constexpr int UNITS = HALF_ROWS * UPR;
    char* cs = smem;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        if (mb) __syncthreads();                 // previous half fully read back
        {
            const int rloc = wm * 32 + fr;       // row inside the half: waves wm=0 -> 0..31, wm=1 -> 32..63
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int plane = GEGLU ? nb : 0;
                const int ncol0 = GEGLU ? wn * 32 : wn * (32 * NB) + nb * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nloc = ncol0 + 8 * q + 4 * fh;
                    float4 v = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2],
                                           acc[mb][nb][4 * q + 3]);
                    *reinterpret_cast<float4*>(cs + plane * PLANE + rloc * CS_LD + nloc * 4) = v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS / NTHREADS; ++i) {
            const int u = tid + NTHREADS * i;
            const int r = u / UPR;
            const int c = (u - r * UPR) * 4;
            // half-row r belongs to wave-row wm = r / 32: global row = m0 + wm*64 + mb*32 + (r % 32)
            const int m = m0 + (r >> 5) * 64 + mb * 32 + (r & 31);
            const int n = n0 + c;
            if (m >= p.M || n >= n_out) continue;
            float4 v = *reinterpret_cast<const float4*>(cs + r * CS_LD + c * 4);
            if (p.bias) {
                const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            }
            if constexpr (GEGLU) {
                float4 g = *reinterpret_cast<const float4*>(cs + PLANE + r * CS_LD + c * 4);
                if (p.bias) {
                    const float4 bg = *reinterpret_cast<const float4*>(p.bias + (p.N >> 1) + n);
                    g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
                }
                v.x *= DC_GELU(g.x); v.y *= DC_GELU(g.y); v.z *= DC_GELU(g.z); v.w *= DC_GELU(g.w);
            }
            if (p.flags & DC_GEMM_GELU) { v.x = DC_GELU(v.x); v.y = DC_GELU(v.y); v.z = DC_GELU(v.z); v.w = DC_GELU(v.w); }
            if (p.rowvec) {
                const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / p.rows_per_vec) * p.rowvec_ld + n);
                v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            if (p.alpha != 1.0f) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
            if (out_f32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
            } else {
                // bf16 rounding, then the residual add (as the reference adds two already-rounded tensors)
                uint2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                if (p.residual) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.ldr + n);
                    const float a0 = __uint_as_float(pk.x << 16) + __uint_as_float(rr.x << 16);
                    const float a1 = __uint_as_float(pk.x & 0xffff0000u) + __uint_as_float(rr.x & 0xffff0000u);
                    const float a2 = __uint_as_float(pk.y << 16) + __uint_as_float(rr.y << 16);
                    const float a3 = __uint_as_float(pk.y & 0xffff0000u) + __uint_as_float(rr.y & 0xffff0000u);
                    pk.x = pack_bf2(a0, a1);
                    pk.y = pack_bf2(a2, a3);
                }
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = pk;
            }
        }
    }
}

template <int BN, bool GEGLU, int MODE>
int launch_mode(const DcGemmParams& p, hipStream_t stream) {
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    const int n_out = GEGLU ? p.N / 2 : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + BM - 1) / BM;
    const size_t lds = 2 * (BM * BK * 2 + BN * BK * 2);
    dc_note_variant(GEGLU ? "gemm_conv_kernel<geglu>" : (BN == 64 ? "gemm_conv_kernel<64>" : "gemm_conv_kernel<128>"));
    hipLaunchKernelGGL((gemm_conv_kernel<BN, GEGLU, MODE>), dim3(tiles_m * tiles_n), dim3(NTHREADS), lds, stream, p);
    DC_CHECK_LAUNCH();
    return 0;
}

template <int BN, bool GEGLU>
int launch(const DcGemmParams& p, hipStream_t stream) {
    if (p.mode == 0) return launch_mode<BN, GEGLU, 0>(p, stream);
    if (GEGLU) return DC_ERR_ARG;
    if (p.mode == 1) return launch_mode<BN, false, 1>(p, stream);
    if (p.mode == 2) return launch_mode<BN, false, 2>(p, stream);
    return DC_ERR_ARG;
}


// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 conv with a NARROW output (N <= 16): the UNet's conv_out (320 -> 4, openaimodel3d.py:545) and the
// AutoencoderKL's decoder conv_out (128 -> 3, ae_modules.py:536). On the tile kernels N is padded to 64 columns (16x the MFMA
// work) and the nine taps re-stage every activation row through the L2: 174 us for 189 MB of input at the 1024 config.
// Here a workgroup owns NC_TH x NC_TW output pixels of one frame and walks the input in 64-channel slices: the slice of the
// (NC_TH + 2) x (NC_TW + 2) halo window is staged ONCE in LDS (coalesced 128-byte rows, zeros outside the image) and the nine
// taps read it at shifted pixel rows; v_mfma_f32_16x16x32_bf16 with the weights as the 16-row operand (rows >= N are zero
// lanes, never loaded), so a lane ends up with 4 output channels of one pixel. The weight fragments of a slice (9 taps x 2 k
// steps) live in registers, fetched straight from the packed weight (23 KB in all: cache-resident). HBM-bound by the input.
constexpr int NC_TH = 4, NC_TW = 64;
constexpr int NC_WIN_W = NC_TW + 2, NC_WIN = (NC_TH + 2) * NC_WIN_W;       // 396 window pixels
constexpr int NC_LDS = NC_WIN * 128;                                       // one 64-channel slice: 128 B per pixel

__device__ __forceinline__ int nc_off(int px, int chunk) { return px * 128 + ((chunk ^ (px & 7)) << 4); }

template <bool F32>
__global__ __launch_bounds__(256) void conv3x3_narrow_kernel(const DcGemmParams p, const int tiles_x, const int tiles_y) {
    __shared__ __attribute__((aligned(16))) char win[NC_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int per_frame = tiles_x * tiles_y;
    const int frame = (int)blockIdx.x / per_frame;
    const int trem = (int)blockIdx.x - frame * per_frame;
    const int ty0 = (trem / tiles_x) * NC_TH, tx0 = (trem % tiles_x) * NC_TW;
    const uint16_t* const fbase = p.A + (size_t)frame * p.IH * p.IW * p.lda;

    f32x4_t acc[NC_TW / 16];
#pragma unroll
    for (int b = 0; b < NC_TW / 16; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nslices = p.Cin / 64;
    for (int sl = 0; sl < nslices; ++sl) {
        // ---- weight fragments of the slice: lane (row lj, k quarter lq) holds k = 32 ks + 8 lq .. + 7 of (tap, ks)
        bf16x8_t wf[9][2];
        {
            const uint16_t* wrow = p.W + (size_t)lj * p.K + (size_t)sl * 576 + lq * 8;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4_t v = {0u, 0u, 0u, 0u};
                    if (lj < p.N) v = *reinterpret_cast<const u32x4_t*>(wrow + t * 64 + ks * 32);
                    wf[t][ks] = __builtin_bit_cast(bf16x8_t, v);
                }
        }
        if (sl > 0) __syncthreads();                    // every wave is done reading the previous slice's window
        // ---- the halo window of the slice: 8 threads per pixel (128 contiguous bytes), zeros outside the image
        // (all loads of a thread first, unconditional - pixels outside the image read the zero page -, then the LDS stores)
        constexpr int NC_IT = (NC_WIN * 8 + 255) / 256;
        u32x4_t stg[NC_IT];
#pragma unroll
        for (int it = 0; it < NC_IT; ++it) {
            int i = tid + it * 256;
            if (i >= NC_WIN * 8) i = NC_WIN * 8 - 1;
            const int px = i >> 3, ch = i & 7;
            const int wy = px / NC_WIN_W, wx = px - wy * NC_WIN_W;
            const int iy = ty0 + wy - 1, ix = tx0 + wx - 1;
            const bool in = iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const uint16_t* src = in ? fbase + ((size_t)iy * p.IW + ix) * p.lda + sl * 64 + ch * 8
                                     : reinterpret_cast<const uint16_t*>(g_zero_chunk);
            stg[it] = *reinterpret_cast<const u32x4_t*>(src);
        }
#pragma unroll
        for (int it = 0; it < NC_IT; ++it) {
            const int i = tid + it * 256;
            if (i < NC_WIN * 8) *reinterpret_cast<u32x4_t*>(win + nc_off(i >> 3, i & 7)) = stg[it];
        }
        __syncthreads();
        // ---- wave w = tile row w: 4 blocks of 16 pixels, 9 taps x 2 k steps each
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int b = 0; b < NC_TW / 16; ++b) {
                    const int px = (wave + dy) * NC_WIN_W + 16 * b + lj + dx;
                    const bf16x8_t xfrag = *reinterpret_cast<const bf16x8_t*>(win + nc_off(px, ks * 4 + lq));
                    acc[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], xfrag, acc[b], 0, 0, 0);
                }
        }
    }
    // ---- epilogue: lane (pixel lj of its block, channel quad lq) holds channels 4 lq .. 4 lq + 3
    const int oy = ty0 + wave;
    if (oy >= p.IH || 4 * lq >= p.N) return;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + 4 * lq);
#pragma unroll
    for (int b = 0; b < NC_TW / 16; ++b) {
        const int ox = tx0 + 16 * b + lj;
        if (ox >= p.IW) continue;
        const size_t row = ((size_t)frame * p.IH + oy) * p.IW + ox;
        const float v0 = (acc[b][0] + bv.x) * p.alpha, v1 = (acc[b][1] + bv.y) * p.alpha;
        const float v2 = (acc[b][2] + bv.z) * p.alpha, v3 = (acc[b][3] + bv.w) * p.alpha;
        if constexpr (F32) {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + row * p.ldc + 4 * lq) = make_float4(v0, v1, v2, v3);
        } else {
            uint2 pk;
            pk.x = pack_bf2(v0, v1); pk.y = pack_bf2(v2, v3);
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + row * p.ldc + 4 * lq) = pk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same window design for N = 128 output channels: the AutoencoderKL's full-resolution ResnetBlock convs (ae_modules.py:151-210;
// [2359296 x 128 x 1152] per 4 frames at 576 x 1024: nine launches per encode + decode call, 20 % of the AE). On the 256 x 128 tile
// kernel the nine taps re-stage every input row through L2 -> LDS (5.4 GB per launch: the CUs' operand path, not the matrix pipe,
// sets its 585 TFLOP/s). Here the halo window of a 4 x 64 pixel tile is staged once per 64-channel slice and the taps' weight
// tiles (128 rows x 64 channels = 16 KB) follow through a second LDS buffer, fetched a tap ahead into registers; a wave owns
// one tile row: 4 pixel blocks x 8 channel blocks of 16x16x32 accumulators (128 registers). Two workgroups per CU (66 KB each)
// cover each other's barriers. Epilogue: + bias, bf16 (+ residual), row-major 16-byte stores through the idle window buffer.
constexpr int WC_N = 128, WC_CB = WC_N / 16;
constexpr int WC_WST = WC_N * 128;                                          // one tap of one slice: [128 rows][128 B]
constexpr int NC_GROUPS = (NC_WIN + 7) / 8;                                 // LDS-DMA pieces of 8 pixels (1 KB) per window slice
constexpr int WC_WIN_BYTES = NC_GROUPS * 1024;                              // 50 KB + the tail of the last piece
constexpr int WC_LDS = WC_WIN_BYTES + WC_WST;

__global__ __launch_bounds__(256, 2) void conv3x3_window128_kernel(const DcGemmParams p, const int tiles_x, const int tiles_y) {
    extern __shared__ __attribute__((aligned(16))) char wsm[];
    char* const win = wsm;
    char* const wst = wsm + WC_WIN_BYTES;
    const unsigned lds_base = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)wsm);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lj = lane & 15, lq = lane >> 4;
    const int per_frame = tiles_x * tiles_y;
    const int frame = (int)blockIdx.x / per_frame;
    const int trem = (int)blockIdx.x - frame * per_frame;
    const int ty0 = (trem / tiles_x) * NC_TH, tx0 = (trem % tiles_x) * NC_TW;
    const uint16_t* const fbase = p.A + (size_t)frame * p.IH * p.IW * p.lda;

    f32x4_t acc[NC_TW / 16][WC_CB];
#pragma unroll
    for (int b = 0; b < NC_TW / 16; ++b)
#pragma unroll
        for (int c = 0; c < WC_CB; ++c) acc[b][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // a thread's share of a weight tap tile: 4 pieces of 16 bytes; piece i = tid + 256 i: row = i >> 3, chunk = i & 7
    u32x4_t wreg[4];
    auto w_fetch = [&](int sl, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = tid + 256 * i;
            wreg[i] = *reinterpret_cast<const u32x4_t*>(p.W + (size_t)(pc >> 3) * p.K + (size_t)sl * 576 + t * 64 + (pc & 7) * 8);
        }
    };
    auto w_store = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pc = tid + 256 * i;
            *reinterpret_cast<u32x4_t*>(wst + nc_off(pc >> 3, pc & 7)) = wreg[i];
        }
    };

    const int nslices = p.Cin / 64;
    w_fetch(0, 0);
#pragma unroll 1
    for (int sl = 0; sl < nslices; ++sl) {
        if (sl > 0) __syncthreads();                    // every wave is done with the previous slice's window and last tap
        // the halo window by LDS-DMA: a wave instruction moves 8 pixels x 128 B (1 KB, lane-linear in LDS; the chunk swizzle is applied
        // on the SOURCE side, pixels outside the image read the zero page) - no staging registers, all of a wave's 12-13 pieces in
        // flight at once (through registers they went in two batches: 256 registers at two waves per SIMD leave no room for 13)
        for (int g = wave; g < NC_GROUPS; g += 4) {
            const int px = g * 8 + (lane >> 3), ch = (lane & 7) ^ (px & 7);
            const int wy = px / NC_WIN_W, wx = px - wy * NC_WIN_W;
            const int iy = ty0 + wy - 1, ix = tx0 + wx - 1;
            const bool in = px < NC_WIN && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const uint16_t* src = in ? fbase + ((size_t)iy * p.IW + ix) * p.lda + sl * 64 + ch * 8
                                     : reinterpret_cast<const uint16_t*>(g_zero_chunk);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_base + g * 1024) : "memory");
        }
        w_store();                                      // tap 0 of this slice (fetched during the previous slice's last tap)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's window pieces have landed (hipcc does not count asm DMA)
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            // the next tap's weights (or the next slice's tap 0) on their way while this tap computes
            if (t + 1 < 9) w_fetch(sl, t + 1);
            else if (sl + 1 < nslices) w_fetch(sl + 1, 0);
            const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t xb[NC_TW / 16];
#pragma unroll
                for (int b = 0; b < NC_TW / 16; ++b) {
                    const int px = (wave + dy) * NC_WIN_W + 16 * b + lj + dx;
                    xb[b] = *reinterpret_cast<const bf16x8_t*>(win + nc_off(px, ks * 4 + lq));
                }
#pragma unroll
                for (int c = 0; c < WC_CB; ++c) {
                    const bf16x8_t wa = *reinterpret_cast<const bf16x8_t*>(wst + nc_off(c * 16 + lj, ks * 4 + lq));
#pragma unroll
                    for (int b = 0; b < NC_TW / 16; ++b)
                        acc[b][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[b], acc[b][c], 0, 0, 0);
                }
            }
            if (t + 1 < 9) {
                __syncthreads();                        // every wave has read this tap's weights
                w_store();
                __syncthreads();
            }
        }
    }
    // ---- epilogue: + bias, bf16, through the (now idle) window buffer into row-major order, 64 channels at a time: the wave's
    // 64 pixels x 128 bytes in its private 8 KB region, read back as 8 pixels x 128 contiguous bytes per instruction (16 bytes
    // per lane); the residual rows are fetched in the same pattern and added after the first rounding (the tile kernels' order)
    __syncthreads();                                    // every wave is done with the window and the last tap
    const int oy = ty0 + wave;
    if (oy >= p.IH) return;
    char* const patch = wsm + wave * 8192;
    uint16_t* const cbase = reinterpret_cast<uint16_t*>(p.C);
    const int rpx = lane >> 3, rch = lane & 7;
    const size_t row0 = ((size_t)frame * p.IH + oy) * p.IW + tx0;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        u32x4_t rres[8];
        if (p.residual) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                int px = it * 8 + rpx;
                if (tx0 + px >= p.IW) px = 0;
                rres[it] = *reinterpret_cast<const u32x4_t*>(p.residual + (row0 + px) * p.ldr + hf * 64 + rch * 8);
            }
        }
#pragma unroll
        for (int b = 0; b < NC_TW / 16; ++b)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int c = hf * 4 + c4;
                float4 v = make_float4(acc[b][c][0], acc[b][c][1], acc[b][c][2], acc[b][c][3]);
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + 16 * c + 4 * lq);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                uint2 pk;
                pk.x = pack_bf2(v.x, v.y); pk.y = pack_bf2(v.z, v.w);
                const int px = 16 * b + lj;                 // 8-byte slot (c4, lq) of the pixel's 128-byte row, chunk-swizzled
                *reinterpret_cast<uint2*>(patch + nc_off(px, 2 * c4 + (lq >> 1)) + (lq & 1) * 8) = pk;
            }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int px = it * 8 + rpx;
            u32x4_t d = *reinterpret_cast<const u32x4_t*>(patch + nc_off(px, rch));
            if (p.residual) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    d[e] = pack_bf2(__uint_as_float(d[e] << 16) + __uint_as_float(rres[it][e] << 16),
                                    __uint_as_float(d[e] & 0xffff0000u) + __uint_as_float(rres[it][e] & 0xffff0000u));
            }
            if (tx0 + px < p.IW) *reinterpret_cast<u32x4_t*>(cbase + (row0 + px) * p.ldc + hf * 64 + rch * 8) = d;
        }
    }
}

inline bool window128_conv_ok(const DcGemmParams& p) {
    // N = 256: two launches over the channel halves (the input is read twice; still ahead of the 256-wide tile kernel)
    if ((dc_gemm_plan_now() & 256) || p.mode != 1 || (p.N != WC_N && p.N != 2 * WC_N) || p.stride != 1 || p.pad != 1 || p.ups) return false;
    if (p.OH != p.IH || p.OW != p.IW || p.Cin % 64 != 0 || p.K != 9 * p.Cin || p.n_pad < p.N) return false;
    if (p.rowvec || p.flags || p.alpha != 1.0f) return false;
    if (p.M % (p.IH * p.IW) != 0 || ((uintptr_t)p.A % 16) != 0 || ((uintptr_t)p.W % 16) != 0 || ((uintptr_t)p.C % 8) != 0) return false;
    if (p.ldc % 8 != 0 || ((uintptr_t)p.C % 16) != 0 || (p.residual && (p.ldr % 8 != 0 || ((uintptr_t)p.residual % 16) != 0))) return false;
    // whole-chip launches only: below ~2 tiles per workgroup slot the tile kernels' finer split wins
    const long long tiles = (long long)(p.M / (p.IH * p.IW)) * ((p.IW + NC_TW - 1) / NC_TW) * ((p.IH + NC_TH - 1) / NC_TH);
    return tiles >= 1024;
}

int launch_window128_conv(const DcGemmParams& p, hipStream_t stream) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&conv3x3_window128_kernel), WC_LDS)) return e;
    const int tiles_x = (p.IW + NC_TW - 1) / NC_TW, tiles_y = (p.IH + NC_TH - 1) / NC_TH;
    const long long grid = (long long)(p.M / (p.IH * p.IW)) * tiles_x * tiles_y;
    if (grid <= 0 || grid > 0x7fffffffll) return DC_ERR_SHAPE;
    dc_note_variant(p.N == WC_N ? "conv3x3_window128_kernel" : "conv3x3_window128_kernel x2");
    for (int h = 0; h < p.N / WC_N; ++h) {
        DcGemmParams q = p;                             // channel half h: weight rows, bias, output and residual columns [128 h, 128 h + 128)
        q.W = p.W + (size_t)h * WC_N * p.K;
        if (p.bias) q.bias = p.bias + h * WC_N;
        q.C = reinterpret_cast<uint16_t*>(p.C) + h * WC_N;
        if (p.residual) q.residual = p.residual + h * WC_N;
        q.N = WC_N;
        hipLaunchKernelGGL(conv3x3_window128_kernel, dim3((unsigned)grid), dim3(256), WC_LDS, stream, q, tiles_x, tiles_y);
        DC_CHECK_LAUNCH();
    }
    return 0;
}

// the launches conv3x3_narrow_kernel takes: N a multiple of 4 (the bias / store quads), no epilogue extras
inline bool narrow_conv_ok(const DcGemmParams& p) {
    if ((dc_gemm_plan_now() & 128) || p.mode != 1 || p.N > 16 || p.N % 4 != 0 || p.stride != 1 || p.pad != 1 || p.ups) return false;
    if (p.OH != p.IH || p.OW != p.IW || p.Cin % 64 != 0 || p.K != 9 * p.Cin) return false;
    if (p.residual || p.rowvec || (p.flags & ~DC_GEMM_OUT_F32)) return false;
    if (p.M % (p.IH * p.IW) != 0 || ((uintptr_t)p.A % 16) != 0 || ((uintptr_t)p.W % 16) != 0) return false;
    if ((p.flags & DC_GEMM_OUT_F32) ? ((uintptr_t)p.C % 16 != 0) : (((uintptr_t)p.C % 8) != 0)) return false;
    return true;
}

int launch_narrow_conv(const DcGemmParams& p, hipStream_t stream) {
    const int tiles_x = (p.IW + NC_TW - 1) / NC_TW, tiles_y = (p.IH + NC_TH - 1) / NC_TH;
    const long long grid = (long long)(p.M / (p.IH * p.IW)) * tiles_x * tiles_y;
    if (grid <= 0 || grid > 0x7fffffffll) return DC_ERR_SHAPE;
    dc_note_variant("conv3x3_narrow_kernel");
    if (p.flags & DC_GEMM_OUT_F32)
        hipLaunchKernelGGL(conv3x3_narrow_kernel<true>, dim3((unsigned)grid), dim3(256), 0, stream, p, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(conv3x3_narrow_kernel<false>, dim3((unsigned)grid), dim3(256), 0, stream, p, tiles_x, tiles_y);
    DC_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int dc_gemm_conv_glds_try(const DcGemmParams& p, hipStream_t stream);   // gemm_conv_glds.hip

static bool glds_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("DC_GEMM_GLDS");
        v = (e && e[0] == '0') ? 0 : 1;
    }
    return v == 1;
}

static thread_local const char* g_last_variant = "none";
void dc_note_variant(const char* name) { g_last_variant = name; }
extern "C" const char* dc_gemm_last_variant(void) { return g_last_variant; }

extern "C" int64_t dc_gemm_workspace_bytes(void) {
    // split-K partials of the largest plan dispatched: 8 splits x 32 remainder tiles (or 3 x 72 tiles) of 256 x 320 fp32
    return (int64_t)8 * 40 * 256 * 320 * 4;
}

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
    if (geglu && ((p.N / 2) % 64 != 0 || p.n_pad < p.N)) return DC_ERR_SHAPE;
    if (p.n_pad < (p.N + 127) / 128 * 128) return DC_ERR_SHAPE;
    if (narrow_conv_ok(p)) return launch_narrow_conv(p, stream);      // conv_out of the UNet / the AE decoder: N <= 16
    if (window128_conv_ok(p)) return launch_window128_conv(p, stream);    // the AE's full-resolution ResnetBlock convs: N = 128
    if (glds_enabled()) {                      // big launches: 256-row LDS-DMA pipeline
        const int r = dc_gemm_conv_glds_try(p, stream);
        if (r != -100) return r;
    }
    if (geglu) return launch<128, true>(p, stream);
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
