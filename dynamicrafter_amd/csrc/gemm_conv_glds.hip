// Large-tile variant of dc_gemm_conv for the big-M launches: 256 x BN x 64 tiles, 8 waves (4 x 2), operands staged
// HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no ds_write), a 3-stage LDS ring with two
// K tiles in flight behind a counted s_waitcnt vmcnt(N) and ONE raw s_barrier per K tile.
//
// Why a second kernel: PMC on the 128-row register-staged kernel (profiles/) shows the MFMA pipe busy 33 % of the
// time with waves parked on the per-tile vmcnt(0)+barrier (32 %) and issue stalls (31 %): its prefetch distance is
// one tile and every staged byte costs a ds_write_b128 (13 cycles per wave instruction).
//
// LDS-DMA is issued from inline asm on purpose: hipcc tracks builtin LDS-DMA as pending LDS writes and puts
// s_waitcnt vmcnt(0) in front of the next ds_read, draining the ring every K step. The asm form is invisible to
// that pass, so the waits are counted by hand here (N = glds per wave per tile x tiles left in flight); the only
// compiler-counted vector-memory operations of this kernel are in the epilogue, after a full drain.
// The LDS image is the same XOR-swizzled [row][8 x 16 B] layout as the small kernel; since LDS-DMA writes
// lane-linear (base + lane*16), the swizzle is applied to the per-lane SOURCE address.
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

constexpr int GBM = 256;
constexpr int GBK = 64;
constexpr int GNT = 512;
// Tile shapes (BN x stages): 128 x 3, 256 x 2 and 320 x 2. The wider tiles exist because the per-CU vector-memory
// path moves 64 B/clk: a 256x128x64 tile stages 48 KB per 1024 MFMA-cycles per SIMD (47 B/clk, 3/4 of that path),
// 256x256 31 B/clk, 256x320 28 B/clk. 320 = the UNet's channel quantum (N = 320, 640, 1280, 2560 tile exactly).

typedef __attribute__((address_space(3))) char lds_char_t;

__device__ __attribute__((aligned(16))) uint32_t g_zero_chunk2[8];

__device__ __forceinline__ int lds_off2(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// one wave instruction: 64 lanes x 16 B -> LDS [lds_dst, lds_dst + 1024), lane-linear
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Split-K scheduling of a launch (host side: dc_gemm_conv_glds_try). A launch covers the logical tiles
// [tile_begin, tile_begin + tile_count); with splits > 1, blockIdx.y selects a K range and the workgroup writes raw
// fp32 accumulators to `partial` ([split][tile - tile_begin][256][BN]); splitk_reduce_kernel sums the splits in a
// fixed order and applies the epilogue. Used where whole tiles cannot fill the chip: UNet level 3 (72 tiles of
// 256 x 320) and the 32-tile remainder wave of level 2.
struct GemmSplit {
    float* partial;
    int splits;
    int tile_begin;
    int tile_count;
    int* err;          // the library's error word (dc_common.h); set by the launchers of kernels with bounded counter waits
    // gemm_pipe320x16_kernel only - the "whole waves + split remainder" plan as ONE launch (1-D grid): the first `whole` workgroups
    // compute the tiles [0, whole) with the normal epilogue, workgroup whole + s * tile_count + i computes K range s of tile
    // tile_begin + i into `partial`. The hardware hands out workgroups in order, so a CU that finishes its whole tile picks up a
    // K range at once instead of idling to the end of the launch (one launch boundary less per conv). 0: classic launch.
    int whole;
};

template <int BN, bool GEGLU, int MODE, int GSTAGES>
__global__ __launch_bounds__(GNT) void gemm_conv_glds_kernel(const DcGemmParams p, const GemmSplit sp) {
    constexpr int NB = BN / 64;             // 32-wide n-blocks per wave (waves are 4 (M) x 2 (N))
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    constexpr int A_BYTES = GBM * GBK * 2;
    constexpr int B_BYTES = BN * GBK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = 4;                 // glds per wave per tile for A: 256 rows * 8 chunks / 512 lanes
    constexpr int B_IT = BN / 64;           // for B
    constexpr int LOADS = A_IT + B_IT;
    static_assert(GSTAGES == 2 || GSTAGES == 3, "ring depth");
    static_assert(!GEGLU || (NB % 2 == 0), "GEGLU needs value and gate blocks per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int n_out = GEGLU ? (p.N >> 1) : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + GBM - 1) / GBM;
    const int swz = sp.tile_begin + xcd_remap(blockIdx.x, sp.tile_count);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int m0 = tile_m * GBM;
    const int n0 = tile_n * BNOUT;

    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    const bf16_t* const zero_ptr = reinterpret_cast<const bf16_t*>(g_zero_chunk2);

    // ---- staging coordinates: wave instruction j covers LDS slots (j*8 + wave)*64 + lane; slot = row*8 + phys chunk
    const int srow = lane >> 3;                       // row inside the 8-row group of this instruction
    const int pchunk = lane & 7;
    const bf16_t* a_ptr[A_IT];
    int a_base[A_IT], a_y[A_IT], a_x[A_IT];
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int r = (j * 8 + wave) * 8 + srow;      // 0..255
        const int chunk = pchunk ^ ((r >> 1) & 7);    // logical 16-byte chunk this lane must fetch
        const int m = m0 + r;
        const bool ok = m < p.M;
        a_base[j] = ok ? 0 : -1;
        a_y[j] = 0; a_x[j] = 0;
        a_ptr[j] = zero_ptr;
        if (MODE == 0) {
            if (ok) a_ptr[j] = p.A + (size_t)m * p.lda + chunk * 8;
        } else if (MODE == 1) {
            // conv3x3 without upsampling: pointer to tap (0,0) of this output row (may lie outside the image: only
            // dereferenced when its mask bit is set) + a 9-bit validity mask; the per-tap offset is wave-uniform
            const int ohw = p.OH * p.OW;
            const int mm = ok ? m : 0;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            int mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mask |= 1 << t;
            }
            a_base[j] = mask;
            a_ptr[j] = p.A + ((long long)n * p.IH * p.IW + (long long)iy0 * p.IW + ix0) * p.lda + chunk * 8;
        } else if (MODE == 3) {
            const int ohw = p.OH * p.OW;
            const int mm = ok ? m : 0;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            if (ok) a_base[j] = n * p.IH * p.IW;
            a_y[j] = oy * p.stride - p.pad;
            a_x[j] = ox * p.stride - p.pad;
            a_ptr[j] = p.A + chunk * 8;               // + row*lda + ci0 per tile
        } else {
            if (ok) a_ptr[j] = p.A + (size_t)m * p.lda + chunk * 8;
            a_y[j] = ((ok ? m : 0) / p.HW) % p.T;
        }
    }
    const bf16_t* b_ptr[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int r = (j * 8 + wave) * 8 + srow;      // row inside the B tile
        const int chunk = pchunk ^ ((r >> 1) & 7);
        int wrow;
        if (GEGLU) wrow = (r < BN / 2) ? (n0 + r) : ((p.N >> 1) + n0 + (r - BN / 2));
        else wrow = n0 + r;
        b_ptr[j] = p.W + (size_t)wrow * p.K + chunk * 8;
    }

    // this workgroup's K tiles: all of them, or the blockIdx.y-th of sp.splits near-equal ranges
    const int nk_all = p.K / GBK;
    const int kt_lo = (int)(((long long)blockIdx.y * nk_all) / sp.splits);
    const int nk = (int)(((long long)(blockIdx.y + 1) * nk_all) / sp.splits);      // exclusive end

    // One quarter of a K tile's LDS-DMA: A instruction `part` and the B instructions j with j % 4 == part. The main
    // loop issues one quarter ahead of each 16-wide K step instead of the whole tile in one burst: a burst of
    // (4 + BN/64) x 8 KB per workgroup stalls the MFMA stream behind it (tools/ubench/gemm_core.hip: 256 x 320 tile,
    // cache-resident operands, 1.39 -> 1.69 PFLOP/s-equivalent with spread issue and without s_setprio).
    auto issue_part = [&](int kt, int stage, int part) __attribute__((always_inline)) {
        const int k0 = kt * GBK;
        const unsigned sa = lds_base + stage * STAGE;
        const unsigned sb = sa + A_BYTES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            if (j != part) continue;
            const bf16_t* src;
            if (MODE == 0) {
                src = (a_base[j] >= 0) ? a_ptr[j] + k0 : zero_ptr;
            } else if (MODE == 1) {
                // K is ordered (64-channel slice, tap, channel): the 9 taps of one slice are consecutive K tiles, so
                // taps 2..9 re-read (shifted) rows that the first tap just pulled into L2
                const int cs = kt / 9;
                const int tap = kt - cs * 9;
                const int dy = tap / 3, dx = tap - dy * 3;
                const long long toff = (long long)(dy * p.IW + dx) * p.lda + cs * 64;      // wave-uniform
                src = ((a_base[j] >> tap) & 1) ? a_ptr[j] + toff : zero_ptr;
            } else if (MODE == 3) {
                const int cs = kt / 9;
                const int tap = kt - cs * 9;
                const int ci0 = cs * 64;
                const int dy = tap / 3, dx = tap - dy * 3;
                const int eh = p.IH << p.ups, ew = p.IW << p.ups;
                const int iy = a_y[j] + dy, ix = a_x[j] + dx;
                const bool ok = (a_base[j] >= 0) & (iy >= 0) & (iy < eh) & (ix >= 0) & (ix < ew);
                const int srcrow = a_base[j] + (iy >> p.ups) * p.IW + (ix >> p.ups);
                src = ok ? a_ptr[j] + (size_t)srcrow * p.lda + ci0 : zero_ptr;
            } else {
                const int cs = kt / 3;
                const int tap = kt - cs * 3;
                const int ci0 = cs * 64;
                const long long shift = (long long)(tap - 1) * p.HW * p.lda + ci0;
                const int tt = a_y[j] + tap - 1;
                const bool ok = (a_base[j] >= 0) & (tt >= 0) & (tt < p.T);
                src = ok ? a_ptr[j] + shift : zero_ptr;
            }
            glds16(src, sa + (j * 8 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            if ((j & 3) == part) glds16(b_ptr[j] + k0, sb + (j * 8 + wave) * 1024);
    };
    auto issue_tile = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_part(kt, stage, part);
    };

    f32x16_t acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue_tile(kt_lo, 0);
    if (GSTAGES == 3 && kt_lo + 1 < nk) issue_tile(kt_lo + 1, 1);

    const int fr = lane & 31, fh = lane >> 5;
    int stage = 0;
#ifdef DC_GEMM_STAMPS
    // tool build only (tools/gemm_stamps.py): shader-clock time per wave spent in the vmcnt wait, at the barrier and in
    // the K-step body (DMA issue + fragment reads + MFMA), summed over the K loop
    unsigned long long st_vm = 0, st_bar = 0, st_body = 0;
    unsigned long long st_t = __builtin_readcyclecounter();
#endif
    for (int kt = kt_lo; kt < nk; ++kt) {
        // tile kt has landed when at most the younger in-flight tile's LOADS remain outstanding
        if (GSTAGES == 3 && kt + 1 < nk) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
#ifdef DC_GEMM_STAMPS
        { const unsigned long long t = __builtin_readcyclecounter(); st_vm += t - st_t; st_t = t; }
#endif
        __builtin_amdgcn_s_barrier();        // everyone's share of tile kt is in LDS; everyone is done with tile kt-1
        asm volatile("" ::: "memory");
#ifdef DC_GEMM_STAMPS
        { const unsigned long long t = __builtin_readcyclecounter(); st_bar += t - st_t; st_t = t; }
#endif
        const bool more = kt + GSTAGES - 1 < nk;
        int s2 = stage + GSTAGES - 1; if (s2 >= GSTAGES) s2 -= GSTAGES;
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < GBK / 16; ++kk) {
            // a quarter of the next K tile's LDS-DMA ahead of every K step (DC_GLDS_ISSUE=1: one burst, =2: two halves ahead
            // of steps 0 and 1 - tool builds; same-box A/B of the whole step: 155.0 (halves) vs 154.6 ms (quarters))
#if defined(DC_GLDS_ISSUE) && DC_GLDS_ISSUE == 1
            if (more && kk == 0) issue_tile(kt + GSTAGES - 1, s2);
#elif defined(DC_GLDS_ISSUE) && DC_GLDS_ISSUE == 2
            if (more && kk < 2) { issue_part(kt + GSTAGES - 1, s2, 2 * kk); issue_part(kt + GSTAGES - 1, s2, 2 * kk + 1); }
#else
            if (more) issue_part(kt + GSTAGES - 1, s2, kk);
#endif
            bf16x8_t xf[2], wf[NB];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                xf[mb] = *reinterpret_cast<const bf16x8_t*>(sa + lds_off2(wm * 64 + mb * 32 + fr, kk * 2 + fh));
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                int brow;
                if (GEGLU) brow = (nb < NB / 2 ? 0 : BN / 2) + wn * (BN / 4) + (nb % (NB / 2 > 0 ? NB / 2 : 1)) * 32;
                else brow = wn * (32 * NB) + nb * 32;
                wf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + lds_off2(brow + fr, kk * 2 + fh));
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
        }
        ++stage; if (stage >= GSTAGES) stage = 0;
#ifdef DC_GEMM_STAMPS
        { const unsigned long long t = __builtin_readcyclecounter(); st_body += t - st_t; st_t = t; }
#endif
    }
    wait_vmcnt<0>();
    __syncthreads();                         // all fragment reads done before the ring is reused for the epilogue
#ifdef DC_GEMM_STAMPS
    if (lane == 0 && p.workspace && !sp.partial && (size_t)(blockIdx.x * 8 + wave + 1) * 32 <= (size_t)p.workspace_bytes) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(p.workspace) + (size_t)(blockIdx.x * 8 + wave) * 4;
        out[0] = st_vm; out[1] = st_bar; out[2] = st_body; out[3] = (unsigned long long)(nk - kt_lo);
    }
#endif

    // ---------------- epilogue: fp32 through LDS, four passes (mb x wave column), coalesced row-major read-back.
    // Pass (mb, ws): the four waves with wn == ws stage acc[mb][*] for their 128 rows; then every thread reads back
    // 4 consecutive channels of one row and applies bias / GEGLU / embedding add / alpha / residual.
    const bool out_f32 = (p.flags & DC_GEMM_OUT_F32) != 0;
    constexpr int PCOLS = BNOUT / 2;             // output columns staged per pass
    constexpr int CS_LD = PCOLS * 4 + 16;
    constexpr int PROWS = 128;
    constexpr int XG = GEGLU ? 2 : 1;
    constexpr int PLANE = PROWS * CS_LD;
    static_assert(XG * PLANE <= GSTAGES * STAGE, "epilogue staging must fit the ring");
    constexpr int UPR = PCOLS / 4;
    constexpr int UNITS = PROWS * UPR;
    static_assert(UNITS % GNT == 0, "read-back units per thread");
    constexpr int NBX = GEGLU ? NB / 2 : NB;     // value blocks per wave
    char* cs = smem;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int mb = pass >> 1, ws = pass & 1;
        if (pass) __syncthreads();
        if (wn == ws) {
            const int rloc = wm * 32 + fr;            // 0..127
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int plane = (GEGLU && nb >= NBX) ? 1 : 0;
                const int ncol0 = (GEGLU ? (nb % NBX) : nb) * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nloc = ncol0 + 8 * q + 4 * fh;
                    float4 v;
                    if (mb == 0) v = make_float4(acc[0][nb][4 * q], acc[0][nb][4 * q + 1], acc[0][nb][4 * q + 2], acc[0][nb][4 * q + 3]);
                    else v = make_float4(acc[1][nb][4 * q], acc[1][nb][4 * q + 1], acc[1][nb][4 * q + 2], acc[1][nb][4 * q + 3]);
                    *reinterpret_cast<float4*>(cs + plane * PLANE + rloc * CS_LD + nloc * 4) = v;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < UNITS / GNT; ++i) {
            const int u = tid + GNT * i;
            const int r = u / UPR;
            const int c = (u - r * UPR) * 4;
            const int m = m0 + (r >> 5) * 64 + mb * 32 + (r & 31);
            const int n = n0 + ws * PCOLS + c;
            if constexpr (!GEGLU) {
                if (sp.partial) {          // split-K: raw accumulators, tile-dense layout; the reduce kernel does the rest
                    const float4 raw = *reinterpret_cast<const float4*>(cs + r * CS_LD + c * 4);
                    const size_t slot = (size_t)blockIdx.y * sp.tile_count + (size_t)(swz - sp.tile_begin);
                    const int rt = (r >> 5) * 64 + mb * 32 + (r & 31);
                    *reinterpret_cast<float4*>(sp.partial + (slot * GBM + rt) * BN + ws * PCOLS + c) = raw;
                    continue;
                }
            }
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

// Sum of the split-K partials of the tiles [tile_begin, tile_begin + tile_count) in split order, then the same
// epilogue as above (bias, GELU, per-row-group vector, alpha, bf16 rounding, residual). One thread per 4 channels.
template <int BN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const DcGemmParams p, const GemmSplit sp) {
    constexpr int UPR = BN / 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int t_loc = blockIdx.y;
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= GBM * UPR) return;
    const int r = u / UPR, c = (u - r * UPR) * 4;
    const int tile = sp.tile_begin + t_loc;
    const int m = (tile / tiles_n) * GBM + r, n = (tile % tiles_n) * BN + c;
    if (m >= p.M || n >= p.N) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < sp.splits; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(sp.partial + (((size_t)s * sp.tile_count + t_loc) * GBM + r) * BN + c);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (p.bias) {
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
    }
    if (p.flags & DC_GEMM_GELU) { v.x = DC_GELU(v.x); v.y = DC_GELU(v.y); v.z = DC_GELU(v.z); v.w = DC_GELU(v.w); }
    if (p.rowvec) {
        const float4 rv = *reinterpret_cast<const float4*>(p.rowvec + (size_t)(m / p.rows_per_vec) * p.rowvec_ld + n);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    if (p.alpha != 1.0f) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
    if (p.flags & DC_GEMM_OUT_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
        return;
    }
    uint2 pk;
    pk.x = pack_bf2(v.x, v.y);
    pk.y = pack_bf2(v.z, v.w);
    if (p.residual) {
        const uint2 rr = *reinterpret_cast<const uint2*>(p.residual + (size_t)m * p.ldr + n);
        pk.x = pack_bf2(__uint_as_float(pk.x << 16) + __uint_as_float(rr.x << 16),
                        __uint_as_float(pk.x & 0xffff0000u) + __uint_as_float(rr.x & 0xffff0000u));
        pk.y = pack_bf2(__uint_as_float(pk.y << 16) + __uint_as_float(rr.y << 16),
                        __uint_as_float(pk.y & 0xffff0000u) + __uint_as_float(rr.y & 0xffff0000u));
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = pk;
}

// ---------------------------------------------------------------------------------------------------------
// Persistent variant for plain GEMMs (mode 0): one workgroup per CU walks a sequence of output tiles and treats
// their K tiles as ONE stream through the LDS ring, so the LDS-DMA of the next output tile is already in flight
// while the current one finishes and runs its epilogue. The short-K linears of the UNet (K = 320 / 640: 5-10 K
// tiles) are latency-bound otherwise: every output tile paid a cold prologue, K sequential waits and an epilogue
// with nothing in flight (measured 250 us for [294912 x 320 x 320], HBM floor ~110 us).
// The epilogue goes straight from the accumulators to global memory (8-byte pieces; no LDS: the ring is busy).
// vmcnt bookkeeping: DMA is counted by hand (asm); the epilogue's loads/stores are compiler-counted. Retirement
// is in issue order, and every compiler-visible operation is issued AFTER the DMA it could be confused with is
// already older than the hand-counted window, so each counted wait can only over-wait, never under-wait.
// MODE as in gemm_conv_glds_kernel: 0 plain rows, 1 conv3x3 (no upsampling), 2 temporal 3-tap conv. The short-K convs
// (level-0/1 ResBlock and TemporalConvBlock convs: 15-90 K tiles) gain from the same cross-tile pipelining.
// Epilogue of one 256 x BN output tile of the persistent kernels, straight from the accumulators (8 waves as 4 x 2, each
// 64 rows x BNOUT/2 columns): bias / GEGLU / GELU / per-row-group vector / alpha in the accumulator layout, then - for bf16
// outputs - through a 2 KB wave-private LDS patch `ebuf` one 32 x 32 block at a time: the accumulator layout (lane = row,
// 4 channels) would store 16-byte pieces of 32 different rows per instruction, and those scattered stores cost as
// much as the whole K loop of a short-K GEMM ([294912 x 320 x 320]: 143 us with them, 66 us without). Read back
// row-major, a lane owns 8 consecutive channels and one instruction moves 16 rows x 64 contiguous bytes. The residual
// (EPI 1) is fetched in the same row-major pattern. LDS operations of one wave execute in order: the patch needs no
// barrier. (Tried and dropped: packing the whole tile first, taking the next K tile's wait and barrier before the
// stores and loading all residual blocks ahead of the first store - slower.)
template <int BN, bool GEGLU, int EPI>
__device__ __forceinline__ void persist_epilogue(f32x16_t (&acc)[2][BN / 64], const DcGemmParams& p, int m0, int n0, int n_out,
                                                 int wm, int wn, int lane, char* ebuf) {
    constexpr int NB = BN / 64;
    constexpr int NBX = GEGLU ? NB / 2 : NB;
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    constexpr int WCOLS = BNOUT / 2;
    // lane coordinates re-derived behind an opaque move: otherwise every address below is loop-invariant, gets
    // hoisted out of the K loop and is kept alive (spilled) across it
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int fr_e = lane_e & 31, fh_e = lane_e >> 5;

    // bias / GEGLU / GELU / per-row-group vector / alpha, in the accumulator layout
    auto finish = [&](int mb, int nb, int q) __attribute__((always_inline)) -> float4 {
        const int m = m0 + wm * 64 + mb * 32 + fr_e;
        const float* rv = p.rowvec ? p.rowvec + (size_t)((m < p.M ? m : 0) / p.rows_per_vec) * p.rowvec_ld : nullptr;
        const int n = n0 + wn * WCOLS + nb * 32 + 8 * q + 4 * fh_e;
        const bool nok = n < n_out;
        float4 v = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
        if (p.bias && nok) {
            const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        if constexpr (GEGLU) {
            float4 gt = make_float4(acc[mb][nb + NBX][4 * q], acc[mb][nb + NBX][4 * q + 1],
                                    acc[mb][nb + NBX][4 * q + 2], acc[mb][nb + NBX][4 * q + 3]);
            if (p.bias && nok) {
                const float4 bg = *reinterpret_cast<const float4*>(p.bias + (p.N >> 1) + n);
                gt.x += bg.x; gt.y += bg.y; gt.z += bg.z; gt.w += bg.w;
            }
            v.x *= DC_GELU(gt.x); v.y *= DC_GELU(gt.y); v.z *= DC_GELU(gt.z); v.w *= DC_GELU(gt.w);
        }
        if (p.flags & DC_GEMM_GELU) { v.x = DC_GELU(v.x); v.y = DC_GELU(v.y); v.z = DC_GELU(v.z); v.w = DC_GELU(v.w); }
        if (rv && nok) {
            const float4 r4 = *reinterpret_cast<const float4*>(rv + n);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        }
        if (p.alpha != 1.0f) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
        return v;
    };
    if constexpr (EPI == 2) {
        // fp32 outputs (VAE attention scores): direct accumulator-layout stores
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < NBX; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = m0 + wm * 64 + mb * 32 + fr_e;
                    const int n = n0 + wn * WCOLS + nb * 32 + 8 * q + 4 * fh_e;
                    const float4 v = finish(mb, nb, q);
                    if (m < p.M && n < n_out) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
                }
    } else {
                const int rrow = lane_e >> 2, rc = lane_e & 3;     // read-back coordinates: row inside a 16-row pass, chunk
        // 32-bit byte offsets from the (uniform) base pointers: one VGPR per (half, pass) row, the n-block is an
        // instruction immediate (dispatch guarantees M * ld * 2 < 4 GiB, N % 8 == 0)
        const int ncol = n0 + wn * WCOLS + rc * 8;
        const char* const rbase = reinterpret_cast<const char*>(p.residual);
        char* const cbase = reinterpret_cast<char*>(p.C);
        auto row_off = [&](int mb, int t, int ld) __attribute__((always_inline)) -> unsigned {
            int mr = m0 + wm * 64 + mb * 32 + t * 16 + rrow;
            if (mr >= p.M) mr = p.M - 1;                    // clamped rows are loaded, never stored
            return ((unsigned)mr * (unsigned)ld + (unsigned)(ncol + 8 <= n_out ? ncol : 0)) * 2u;
        };
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            unsigned co[2], ro[2];
            bool rok[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                co[t] = row_off(mb, t, p.ldc);
                ro[t] = row_off(mb, t, p.ldr);
                rok[t] = m0 + wm * 64 + mb * 32 + t * 16 + rrow < p.M;
            }
#pragma unroll
            for (int nb = 0; nb < NBX; ++nb) {
                u32x4_t rr[2];
                if constexpr (EPI == 1) {
                    const int nbo = (ncol + nb * 32 + 8 <= n_out) ? nb * 64 : 0;      // stay inside the row
#pragma unroll
                    for (int t = 0; t < 2; ++t) rr[t] = *reinterpret_cast<const u32x4_t*>(rbase + ro[t] + nbo);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // row fr, 8-byte slot 2q+fh, XOR-swizzled by an even number per row pair
                    const float4 v = finish(mb, nb, q);
                    uint2 pk;
                    pk.x = pack_bf2(v.x, v.y);
                    pk.y = pack_bf2(v.z, v.w);
                    *reinterpret_cast<uint2*>(ebuf + fr_e * 64 + (((2 * q + fh_e) ^ (((fr_e >> 1) & 3) << 1)) << 3)) = pk;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int r = t * 16 + rrow;
                    u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + r * 64 + ((rc ^ ((r >> 1) & 3)) << 4));
                    if (!rok[t] || ncol + nb * 32 + 8 > n_out) continue;
                    if constexpr (EPI == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            d[e] = pack_bf2(__uint_as_float(d[e] << 16) + __uint_as_float(rr[t][e] << 16),
                                            __uint_as_float(d[e] & 0xffff0000u) + __uint_as_float(rr[t][e] & 0xffff0000u));
                    }
                    *reinterpret_cast<u32x4_t*>(cbase + co[t] + nb * 64) = d;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// EPI: 0 bf16 | 1 bf16 + residual loaded in the epilogue | 2 fp32 | 3 bf16 + residual streamed through the ring.
// EPI 3: the residual tile [256 x BN] follows the K tiles of its output tile through the LDS ring as BN/64 more
// A-operand tiles and is added on the MFMA pipe (acc += R * I with an identity weight fragment built in registers):
// no vector-memory load is left in the epilogue. With EPI 1 every 32 x 32 block waits a full HBM round trip for
// its residual rows with nothing else in flight (compiler-counted vmcnt(0)): [294912 x 320 x 320] 198 us vs 129 us
// without the residual, for 31 us worth of extra bytes. The sum is rounded to bf16 once (EPI 1 rounds the GEMM
// result, then the sum).
// Logical tile id -> (row panel, column tile) of the persistent kernels. The 32 workgroups of an XCD work on 32 consecutive
// logical ids at a time and share a 4 MB L2: column-fastest order makes them 1.6 row panels x all 20 / 40 column tiles of a
// GEGLU projection - every workgroup streams its own weight tile, 6.5 MB per round through an L2 that keeps none of it
// (counters: 5.2x the algorithmic bytes). group = 8: blocks of 4 panels x 8 column tiles, so a weight tile is shared by 4
// and an activation panel by 8 workgroups that run in step (4 + 8 operand streams per round instead of 1.6 + 20).
__device__ __forceinline__ void persist_tile(int logical, int tiles_m, int tiles_n, int group, int& tm, int& tn) {
    if (group <= 1 || tiles_n <= group) { tn = logical % tiles_n; tm = logical / tiles_n; return; }
    const int PG = 32 / group;                          // panels per block
    const int g = logical / (PG * tiles_n);             // panel group
    const int r = logical - g * (PG * tiles_n);
    int pg = tiles_m - g * PG;                          // panels in this group (the last one may have fewer)
    if (pg > PG) pg = PG;
    const int full = (tiles_n / group) * (pg * group);  // tiles in full-width column chunks
    if (r < full) {
        const int c = r / (pg * group), w = r - c * (pg * group);
        tn = c * group + w / pg; tm = g * PG + w % pg;
    } else {
        const int rr = r - full;
        tn = (tiles_n / group) * group + rr / pg; tm = g * PG + rr % pg;
    }
}

#include "gemm_pipe.h"
#include "gemm_pipe16.h"
#include "gemm_pp.h"

template <int BN, bool GEGLU, int GSTAGES, int EPI, int MODE>
__global__ __launch_bounds__(GNT) void gemm_persist_kernel(const DcGemmParams p, const int tile_group) {
    static_assert(MODE == 0 || !GEGLU, "GEGLU is a plain-GEMM epilogue");
    static_assert(EPI != 3 || !GEGLU, "the ring residual is a plain bf16 epilogue");
    constexpr bool RES_RING = (EPI == 3);
    constexpr int NR = RES_RING ? (GEGLU ? 0 : BN / 64) : 0;      // residual tiles per output tile: tile j = 32 columns of
                                                                  // each wave column, [n0 + 32j, +32) | [n0 + BN/2 + 32j, +32)
    constexpr int NB = BN / 64;
    constexpr int NBX = GEGLU ? NB / 2 : NB;
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    constexpr int WCOLS = BNOUT / 2;            // output columns per wave
    constexpr int A_BYTES = GBM * GBK * 2;
    constexpr int B_BYTES = BN * GBK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int A_IT = 4;
    constexpr int B_IT = BN / 64;
    constexpr int LOADS = A_IT + B_IT;
    static_assert(GSTAGES >= 2 && GSTAGES <= 4, "ring depth");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;

    const int n_out = GEGLU ? (p.N >> 1) : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + GBM - 1) / GBM;
    const int ntiles = tiles_m * tiles_n;
    const int G = gridDim.x;
    const int nj = (ntiles - (int)blockIdx.x + G - 1) / G;      // output tiles of this workgroup
    const int nk = p.K / GBK;
    const int nkr = nk + NR;                                    // ring tiles per output tile
    const int total = nj * nkr;

    const unsigned lds_base = (unsigned)(unsigned long)((lds_char_t*)smem);
    const bf16_t* const zero_ptr = reinterpret_cast<const bf16_t*>(g_zero_chunk2);
    const int srow = lane >> 3;
    const int pchunk = lane & 7;

    // ---- issue side: descriptors of the tile whose K tiles are being requested
    const bf16_t* a_ptr[A_IT];
    int a_aux[A_IT];            // MODE 1: 9-bit tap validity mask; MODE 2: frame index of the row (or -1000: row >= M)
    const bf16_t* b_ptr[B_IT];
    int i_tile = 0, i_kt = 0, i_stage = 0;
    int i_m0 = 0, i_n0 = 0;     // origin of the issue tile (ring residual)
    auto set_issue_tile = [&](int j) __attribute__((always_inline)) {
        const int logical = xcd_remap((int)blockIdx.x + j * G, ntiles);
        int tn, tmi;
        persist_tile(logical, tiles_m, tiles_n, tile_group, tmi, tn);
        const int m0 = tmi * GBM, n0 = tn * BNOUT;
        i_m0 = m0; i_n0 = n0;
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
            const int r = (q * 8 + wave) * 8 + srow;
            const int chunk = pchunk ^ ((r >> 1) & 7);
            const int m = m0 + r;
            const bool ok = m < p.M;
            a_aux[q] = 0;
            if (MODE == 0) {
                a_ptr[q] = ok ? p.A + (size_t)m * p.lda + chunk * 8 : nullptr;
            } else if (MODE == 1) {
                const int ohw = p.OH * p.OW;
                const int mm = ok ? m : 0;
                const int n = mm / ohw;
                const int rem = mm - n * ohw;
                const int oy = rem / p.OW;
                const int ox = rem - oy * p.OW;
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                    if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mask |= 1 << t;
                }
                a_aux[q] = mask;
                a_ptr[q] = p.A + ((long long)n * p.IH * p.IW + (long long)iy0 * p.IW + ix0) * p.lda + chunk * 8;
            } else {
                a_ptr[q] = p.A + (size_t)(ok ? m : 0) * p.lda + chunk * 8;
                a_aux[q] = ok ? (m / p.HW) % p.T : -1000;
            }
        }
#pragma unroll
        for (int q = 0; q < B_IT; ++q) {
            const int r = (q * 8 + wave) * 8 + srow;
            const int chunk = pchunk ^ ((r >> 1) & 7);
            int wrow;
            if (GEGLU) wrow = (r < BN / 2) ? (n0 + r) : ((p.N >> 1) + n0 + (r - BN / 2));
            else wrow = n0 + r;
            b_ptr[q] = p.W + (size_t)wrow * p.K + chunk * 8;
        }
    };
    // quarter `part` of the next K tile's LDS-DMA (see issue_part above); part 3 advances the issue cursor
    auto issue_next_part = [&](int part) __attribute__((always_inline)) {
        const int k0 = i_kt * GBK;
        const unsigned sa = lds_base + i_stage * STAGE;
        const unsigned sb = sa + A_BYTES;
        const bool res_tile = RES_RING && i_kt >= nk;             // wave-uniform
#pragma unroll
        for (int q = 0; q < A_IT; ++q) {
            if (q != part) continue;
            const bf16_t* src;
            if (res_tile) {
                // residual rows of the issue tile staged like an activation tile: 16-byte chunks 0-3 = the 32 columns of
                // block (i_kt - nk) of wave column 0, chunks 4-7 = the same block of wave column 1
                const int r = (q * 8 + wave) * 8 + srow;
                const int chunk = pchunk ^ ((r >> 1) & 7);
                const int m = i_m0 + r;
                const int col = i_n0 + (i_kt - nk) * 32 + (chunk >> 2) * (BN / 2) + (chunk & 3) * 8;
                src = (m < p.M) ? p.residual + (size_t)m * p.ldr + col : zero_ptr;
            } else if (MODE == 0) {
                src = a_ptr[q] ? a_ptr[q] + k0 : zero_ptr;
            } else if (MODE == 1) {
                const int cs = i_kt / 9;
                const int tap = i_kt - cs * 9;
                const int dy = tap / 3, dx = tap - dy * 3;
                const long long toff = (long long)(dy * p.IW + dx) * p.lda + cs * 64;      // wave-uniform
                src = ((a_aux[q] >> tap) & 1) ? a_ptr[q] + toff : zero_ptr;
            } else {
                const int cs = i_kt / 3;
                const int tap = i_kt - cs * 3;
                const long long shift = (long long)(tap - 1) * p.HW * p.lda + cs * 64;
                const int tt = a_aux[q] + tap - 1;
                src = (tt >= 0 && tt < p.T) ? a_ptr[q] + shift : zero_ptr;
            }
            glds16(src, sa + (q * 8 + wave) * 1024);
        }
        if (!res_tile) {
#pragma unroll
            for (int q = 0; q < B_IT; ++q)
                if ((q & 3) == part) glds16(b_ptr[q] + k0, sb + (q * 8 + wave) * 1024);
        }
        if (part == 3) {
            if (++i_stage >= GSTAGES) i_stage = 0;
            if (++i_kt >= nkr) {
                i_kt = 0;
                ++i_tile;
                if (i_tile < nj) set_issue_tile(i_tile);
            }
        }
    };
    auto issue_next = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int part = 0; part < 4; ++part) issue_next_part(part);
    };

    f32x16_t acc[2][NB];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    set_issue_tile(0);
    issue_next();
    if (GSTAGES >= 3 && total > 1) issue_next();
    if (GSTAGES == 4 && total > 2) issue_next();

    int c_tile = 0, c_kt = 0, stage = 0;
    for (int g = 0; g < total; ++g) {
        // tile g has landed once at most the (GSTAGES-2) younger in-flight tiles' DMAs remain outstanding
        const int younger = total - 1 - g;
        if (GSTAGES == 4 && younger >= 2) {
            static_assert(GSTAGES != 4 || !RES_RING, "4-deep ring: residual tiles not counted");
            wait_vmcnt<2 * LOADS>();
        } else if (GSTAGES >= 3 && younger >= 1) {
            // the one younger tile in flight is a residual tile (A_IT loads, no weight rows) or a K tile (LOADS)
            if (RES_RING && c_kt + 1 >= nk) wait_vmcnt<A_IT>(); else wait_vmcnt<LOADS>();
        } else {
            wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = g + GSTAGES - 1 < total;
        const char* sa = smem + stage * STAGE;
        const char* sb = sa + A_BYTES;
        {
#pragma unroll
            for (int kk = 0; kk < GBK / 16; ++kk) {
                if (more) issue_next_part(kk);
                bf16x8_t xf[2], wf[NB];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    xf[mb] = *reinterpret_cast<const bf16x8_t*>(sa + lds_off2(wm * 64 + mb * 32 + fr, kk * 2 + fh));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    int brow;
                    if (GEGLU) brow = (nb < NBX ? 0 : BN / 2) + wn * (BN / 4) + (nb % NBX) * 32;
                    else brow = wn * (32 * NB) + nb * 32;
                    wf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + lds_off2(brow + fr, kk * 2 + fh));
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb)
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], xf[mb], acc[mb][nb], 0, 0, 0);
            }
        }
        if (++stage >= GSTAGES) stage = 0;
        if (++c_kt >= nk) {
            if constexpr (RES_RING) {
                // ---- the residual tiles of this output tile: ring tiles g+1 .. g+NB, block j of both wave columns each
                // (static accumulator indices: a data-dependent choice of blocks makes hipcc shuffle and spill them)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    ++g;
                    const int younger_r = total - 1 - g;
                    if (GSTAGES >= 3 && younger_r >= 1) {
                        if (j + 1 < NB) wait_vmcnt<A_IT>(); else wait_vmcnt<LOADS>();
                    } else {
                        wait_vmcnt<0>();
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (g + GSTAGES - 1 < total) {
#pragma unroll
                        for (int kk = 0; kk < GBK / 16; ++kk) issue_next_part(kk);
                    }
                    const char* sr = smem + stage * STAGE;
                    int lane_r = lane;
                    asm volatile("" : "+v"(lane_r));             // keep these addresses out of the K loop's live set
                    const int fr_r = lane_r & 31, fh_r = lane_r >> 5;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        // identity weight fragment: W'[n = fr][k = 16 i + 8 fh + e] = (n == k)
                        const int e1 = fr_r - 16 * i - 8 * fh_r;
                        bf16x8_t idf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) idf[e] = (e == e1) ? (short)0x3F80 : (short)0;
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const bf16x8_t xr = *reinterpret_cast<const bf16x8_t*>(
                                sr + lds_off2(wm * 64 + mb * 32 + fr_r, (wn * 2 + i) * 2 + fh_r));
                            acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(idf, xr, acc[mb][j], 0, 0, 0);
                        }
                    }
                    if (++stage >= GSTAGES) stage = 0;
                }
            }
            // ---- epilogue of output tile c_tile.
            // bf16 outputs go through a 2 KB wave-private LDS patch behind the ring, one 32 x 32 block at a time: the
            // accumulator layout (lane = row, 4 channels) would store 16-byte pieces of 32 different rows per
            // instruction, and those scattered stores cost as much as the whole K loop of a short-K GEMM
            // ([294912 x 320 x 320]: 143 us with them, 66 us without). Read back row-major, a lane owns 8 consecutive
            // channels and one instruction moves 16 rows x 64 contiguous bytes.
            // The residual is fetched in the same row-major pattern. LDS operations of one wave execute in order: the
            // patch needs no barrier. (Tried and dropped: packing the whole tile first, taking the next K tile's wait
            // and barrier before the stores and loading all residual blocks ahead of the first store - slower.)
            c_kt = 0;
            const int logical = xcd_remap((int)blockIdx.x + c_tile * G, ntiles);
            ++c_tile;
            int tn, tmi;
            persist_tile(logical, tiles_m, tiles_n, tile_group, tmi, tn);
            const int m0 = tmi * GBM, n0 = tn * BNOUT;
            persist_epilogue<BN, GEGLU, EPI>(acc, p, m0, n0, n_out, wm, wn, lane, smem + GSTAGES * STAGE + wave * 2048);
            zero_acc();
        }
    }
}

template <int BN, bool GEGLU, int EPI, int MODE, int ST_FORCE>
int launch_persist_epi_st(const DcGemmParams& p, hipStream_t stream, int grid);

template <int BN, bool GEGLU, int EPI, int MODE>
int launch_persist_epi(const DcGemmParams& p, hipStream_t stream, int grid) {
    // tool switch (tools/gemm_probe.py): DC_GEMM_STAGES=2 runs the <= 128-wide tiles on a 2-deep ring to expose what the
    // prefetch depth is worth on HBM-sourced operands
    static const int st = [] { const char* e = getenv("DC_GEMM_STAGES"); return e ? atoi(e) : 0; }();
    if constexpr (BN <= 128 && MODE == 0 && !GEGLU) { if (st == 2) return launch_persist_epi_st<BN, GEGLU, EPI, MODE, 2>(p, stream, grid); }
    return launch_persist_epi_st<BN, GEGLU, EPI, MODE, 0>(p, stream, grid);
}

template <int BN, bool GEGLU, int EPI, int MODE, int ST_FORCE>
int launch_persist_epi_st(const DcGemmParams& p, hipStream_t stream, int grid) {
    // ring depth by LDS budget (160 KB minus 8 x 2 KB of epilogue patches): 64-wide 3 x 40 KB, 128-wide 3 x 48 KB,
    // 256/320-wide 2 x 64/72 KB
    constexpr int ST = ST_FORCE ? ST_FORCE : ((BN <= 128) ? 3 : 2);
    constexpr size_t lds = (size_t)ST * (GBM * GBK * 2 + BN * GBK * 2) + 8 * 2048;
    static_assert(lds <= 163840, "LDS budget");
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_persist_kernel<BN, GEGLU, ST, EPI, MODE>), (int)lds)) return e;
    dc_note_variant(GEGLU ? (BN == 256 ? "gemm_persist_kernel<256,geglu>" : "gemm_persist_kernel<128,geglu>")
                    : MODE == 1 ? "gemm_persist_kernel<320,conv>" : MODE == 2 ? "gemm_persist_kernel<320,tconv>"
                    : BN == 320 ? (EPI == 1 || EPI == 3 ? "gemm_persist_kernel<320,residual>" : "gemm_persist_kernel<320>")
                    : BN == 128 ? "gemm_persist_kernel<128>" : "gemm_persist_kernel<64>");
    static const int group = [] { const char* e = getenv("DC_GEMM_GROUP"); return e ? atoi(e) : 8; }();
    hipLaunchKernelGGL((gemm_persist_kernel<BN, GEGLU, ST, EPI, MODE>), dim3(grid), dim3(GNT), lds, stream, p,
                       (group == 2 || group == 4 || group == 8 || group == 16) ? group : 1);
    DC_CHECK_LAUNCH();
    return 0;
}

// The residual can ride the LDS ring (EPI 3) when it is added to the raw GEMM result (no activation / scale between)
// and its 64-column tiles are whole and 16-byte aligned. DC_GEMM_RING_RESIDUAL=0: epilogue loads (EPI 1), for A/B runs.
inline bool ring_residual_ok(const DcGemmParams& p) {
    static const int on = [] { const char* e = getenv("DC_GEMM_RING_RESIDUAL"); return e ? atoi(e) : 1; }();
    return on && p.residual && !(p.flags & (DC_GEMM_GELU | DC_GEMM_GEGLU | DC_GEMM_OUT_F32)) && p.alpha == 1.0f &&
           (p.N % 64 == 0) && (p.ldr % 8 == 0) && ((uintptr_t)p.residual % 16 == 0);
}

// epilogue flavour: 0 bf16, 1 bf16 + residual, 2 fp32, 3 bf16 + residual through the ring
template <int BN, bool GEGLU>
int launch_persist(const DcGemmParams& p, hipStream_t stream, int grid) {
    if (p.flags & DC_GEMM_OUT_F32) {
        if constexpr (GEGLU) return DC_ERR_ARG;
        else return launch_persist_epi<BN, GEGLU, 2, 0>(p, stream, grid);
    }
    if (p.residual) {
        if constexpr (!GEGLU) { if (ring_residual_ok(p) && p.N % BN == 0) return launch_persist_epi<BN, GEGLU, 3, 0>(p, stream, grid); }
        return launch_persist_epi<BN, GEGLU, 1, 0>(p, stream, grid);
    }
    return launch_persist_epi<BN, GEGLU, 0, 0>(p, stream, grid);
}

// 320-wide persistent conv3x3 (no upsampling) / temporal conv, bf16 outputs
int launch_persist_conv320(const DcGemmParams& p, hipStream_t stream) {
    const bool ring = ring_residual_ok(p);
    if (p.mode == 1) return p.residual ? (ring ? launch_persist_epi<320, false, 3, 1>(p, stream, 256)
                                               : launch_persist_epi<320, false, 1, 1>(p, stream, 256))
                                       : launch_persist_epi<320, false, 0, 1>(p, stream, 256);
    return p.residual ? (ring ? launch_persist_epi<320, false, 3, 2>(p, stream, 256)
                              : launch_persist_epi<320, false, 1, 2>(p, stream, 256))
                      : launch_persist_epi<320, false, 0, 2>(p, stream, 256);
}

template <int BN, bool GEGLU, int MODE, int GSTAGES>
int launch_glds(const DcGemmParams& p, hipStream_t stream) {
    constexpr int BNOUT = GEGLU ? BN / 2 : BN;
    const int n_out = GEGLU ? p.N / 2 : p.N;
    const int tiles_n = (n_out + BNOUT - 1) / BNOUT;
    const int tiles_m = (p.M + GBM - 1) / GBM;
    constexpr size_t lds = (size_t)GSTAGES * (GBM * GBK * 2 + BN * GBK * 2);
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_conv_glds_kernel<BN, GEGLU, MODE, GSTAGES>), (int)lds)) return e;
    GemmSplit sp;
    sp.partial = nullptr; sp.splits = 1; sp.tile_begin = 0; sp.tile_count = tiles_m * tiles_n; sp.err = nullptr; sp.whole = 0;
    dc_note_variant(GEGLU ? "gemm_conv_glds_kernel<geglu>"
                    : BN == 320 ? (MODE == 0 ? "gemm_conv_glds_kernel<320>" : MODE == 2 ? "gemm_conv_glds_kernel<320,tconv>" : "gemm_conv_glds_kernel<320,conv>")
                    : BN == 256 ? (MODE == 0 ? "gemm_conv_glds_kernel<256>" : MODE == 2 ? "gemm_conv_glds_kernel<256,tconv>" : "gemm_conv_glds_kernel<256,conv>")
                    : (MODE == 0 ? "gemm_conv_glds_kernel<128>" : MODE == 2 ? "gemm_conv_glds_kernel<128,tconv>" : "gemm_conv_glds_kernel<128,conv>"));
    hipLaunchKernelGGL((gemm_conv_glds_kernel<BN, GEGLU, MODE, GSTAGES>), dim3(tiles_m * tiles_n), dim3(GNT), lds, stream, p, sp);
    DC_CHECK_LAUNCH();
    return 0;
}

// The one-wave-per-SIMD 256 x 320 kernel (gemm_pipe16.h) takes a launch when dc_gemm_set_plan selects it and its addressing
// applies: bf16 output through the row-major epilogue, whole 64-channel slices, activation offsets below 2^31.
// default 3 (bit 0: every 3x3 conv, bit 1: also the long-K (>= 1920) plain and temporal launches): 10-18 % faster than the
// 8-wave kernel there (the chip holds a higher clock on v_mfma_f32_16x16x32_bf16); short-K launches stay on the
// persistent kernels. Bit 3 of older plan values (9 / 11) is accepted and dropped.
// bit 4 (16): the ping-pong kernel (gemm_pp.h: 4-wave workgroups, two per CU) for the GEGLU projections with K <= 640 and
// >= 1024 tiles; bit 5 (32): without the K limit (tests, A/B).
// bit 6 (64): the "whole waves + split remainder" plans as two launches (round 3's form) instead of one (GemmSplit::whole): tests, A/B.
constexpr int PLAN_DEFAULT = 3 | 16;
// bit 7 (128): conv3x3_narrow_kernel (gemm_conv.hip: conv_out of the UNet / AE decoder) off - the tile kernels take those launches.
// bit 8 (256): conv3x3_window128_kernel (gemm_conv.hip: the AE's full-resolution N = 128 convs) off.
constexpr int PLAN_MASK = 3 | 16 | 32 | 64 | 128 | 256;
inline bool plan_valid(int plan) { return plan >= 0 && plan <= 511 && !(plan & 4); }
std::atomic<int> g_gemm_plan{[] {
    const char* e = getenv("DC_GEMM_PLAN");
    const int v = e ? atoi(e) : PLAN_DEFAULT;
    return plan_valid(v) ? (v & PLAN_MASK) : PLAN_DEFAULT;
}()};
inline int pipe_min_k() { static const int v = [] { const char* e = getenv("DC_PIPE_MIN_K"); return e ? atoi(e) : 1920; }(); return v; }

template <int MODE, int EPI>
int launch_pipe_shape(const DcGemmParams& p, hipStream_t stream, GemmSplit sp, int gx, int gy) {
    sp.err = dc_error_word_device();
    if (!sp.err) return DC_ERR_ARG;
    return launch_pipe320x16<MODE, EPI>(p, stream, sp, gx, gy);
}

inline bool pipe_ok(const DcGemmParams& p, int plan) {
    if (!(plan & 3) || (!(plan & 2) && p.mode != 1)) return false;
    if (p.mode != 1 && p.K < pipe_min_k()) return false;
    if (p.flags & (DC_GEMM_GEGLU | DC_GEMM_OUT_F32)) return false;
    if (p.ups && !(p.mode == 1 && p.ups == 1 && p.stride == 1 && p.pad == 1 && p.OH == 2 * p.IH && p.OW == 2 * p.IW)) return false;
    if (p.N % 320 != 0 || p.n_pad < p.N || p.K % 64 != 0 || p.lda % 8 != 0 || ((uintptr_t)p.A % 16) != 0) return false;
    if (p.ldc % 8 != 0 || ((uintptr_t)p.C % 16) != 0 || (long long)p.M * p.ldc >= (1ll << 31)) return false;
    if (p.residual && (p.ldr % 8 != 0 || ((uintptr_t)p.residual % 16) != 0 || (long long)p.M * p.ldr >= (1ll << 31))) return false;
    long long rows = p.M, extra = 0;
    if (p.mode == 1) { if (p.Cin % 64 != 0 || p.K != 9 * p.Cin) return false; rows = (long long)(p.M / (p.OH * p.OW) + 1) * p.IH * p.IW; extra = 2ll * p.IW + 2; }
    else if (p.mode == 2) { if (p.Cin % 64 != 0 || p.K != 3 * p.Cin) return false; extra = 2ll * p.HW; }
    else if (p.mode != 0) return false;
    if ((rows + extra) * p.lda * 2 >= (1ll << 31)) return false;
    if (320ll * p.K * 2 >= (1ll << 31)) return false;
    return true;
}

int launch_pipe_whole(const DcGemmParams& p, hipStream_t stream) {
    const int ntiles = ((p.M + GBM - 1) / GBM) * (p.N / 320);
    GemmSplit sp;
    sp.partial = nullptr; sp.splits = 1; sp.tile_begin = 0; sp.tile_count = ntiles; sp.err = nullptr; sp.whole = 0;
    dc_note_variant(p.mode == 0 ? "gemm_pipe320x16_kernel" : p.mode == 2 ? "gemm_pipe320x16_kernel<tconv>"
                    : p.ups ? "gemm_pipe320x16_kernel<conv,ups>" : "gemm_pipe320x16_kernel<conv>");
    if (p.mode == 0) return p.residual ? launch_pipe_shape<0, 1>(p, stream, sp, ntiles, 1) : launch_pipe_shape<0, 0>(p, stream, sp, ntiles, 1);
    if (p.mode == 1 && p.ups) return p.residual ? launch_pipe_shape<3, 1>(p, stream, sp, ntiles, 1) : launch_pipe_shape<3, 0>(p, stream, sp, ntiles, 1);
    if (p.mode == 1) return p.residual ? launch_pipe_shape<1, 1>(p, stream, sp, ntiles, 1) : launch_pipe_shape<1, 0>(p, stream, sp, ntiles, 1);
    return p.residual ? launch_pipe_shape<2, 1>(p, stream, sp, ntiles, 1) : launch_pipe_shape<2, 0>(p, stream, sp, ntiles, 1);
}

// 320-wide tiles with split-K: `full` leading tiles as whole tiles (0 = none), the remaining tiles cut into `splits`
// K ranges + reduce. Partials: splits * (ntiles - full) * 256 * 320 floats of workspace.
template <int MODE, int BN = 320>
int launch_glds320_split(const DcGemmParams& p, hipStream_t stream, int ntiles, int full, int splits, bool use_pipe, bool merged) {
    constexpr int GSTAGES = 2;
    constexpr size_t lds = (size_t)GSTAGES * (GBM * GBK * 2 + BN * GBK * 2);
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_conv_glds_kernel<BN, false, MODE, GSTAGES>), (int)lds)) return e;
    if constexpr (BN == 320) {
        if (use_pipe) {
            dc_note_variant(MODE == 0 ? "gemm_pipe320x16_kernel+splitk" : MODE == 2 ? "gemm_pipe320x16_kernel<tconv>+splitk"
                            : MODE == 3 ? "gemm_pipe320x16_kernel<conv,ups>+splitk" : "gemm_pipe320x16_kernel<conv>+splitk");
            GemmSplit sp;
            sp.err = nullptr; sp.whole = 0;
            if (full > 0 && merged && full % 8 == 0) {
                // whole tiles and the K ranges of the remainder tiles in ONE launch (GemmSplit::whole)
                sp.partial = reinterpret_cast<float*>(p.workspace); sp.splits = splits; sp.tile_begin = full; sp.tile_count = ntiles - full;
                sp.whole = full;
                const int gx = full + sp.tile_count * splits;
                if (const int e = p.residual ? launch_pipe_shape<MODE, 1>(p, stream, sp, gx, 1) : launch_pipe_shape<MODE, 0>(p, stream, sp, gx, 1)) return e;
                sp.whole = 0;
            } else {
                if (full > 0) {
                    sp.partial = nullptr; sp.splits = 1; sp.tile_begin = 0; sp.tile_count = full;
                    if (const int e = p.residual ? launch_pipe_shape<MODE, 1>(p, stream, sp, full, 1) : launch_pipe_shape<MODE, 0>(p, stream, sp, full, 1)) return e;
                }
                sp.partial = reinterpret_cast<float*>(p.workspace); sp.splits = splits; sp.tile_begin = full; sp.tile_count = ntiles - full;
                if (const int e = launch_pipe_shape<MODE, 0>(p, stream, sp, sp.tile_count, splits)) return e;
            }
            hipLaunchKernelGGL((splitk_reduce_kernel<BN>), dim3((GBM * (BN / 4) + 255) / 256, sp.tile_count), dim3(256), 0, stream, p, sp);
            DC_CHECK_LAUNCH();
            return 0;
        }
    }
    if constexpr (BN == 320)
        dc_note_variant(MODE == 0 ? "gemm_conv_glds_kernel<320>+splitk" : MODE == 2 ? "gemm_conv_glds_kernel<320,tconv>+splitk" : "gemm_conv_glds_kernel<320,conv>+splitk");
    else
        dc_note_variant(MODE == 0 ? "gemm_conv_glds_kernel<256>+splitk" : MODE == 2 ? "gemm_conv_glds_kernel<256,tconv>+splitk" : "gemm_conv_glds_kernel<256,conv>+splitk");
    GemmSplit sp;
    sp.err = nullptr; sp.whole = 0;
    if (full > 0) {
        sp.partial = nullptr; sp.splits = 1; sp.tile_begin = 0; sp.tile_count = full;
        hipLaunchKernelGGL((gemm_conv_glds_kernel<BN, false, MODE, GSTAGES>), dim3(full), dim3(GNT), lds, stream, p, sp);
        DC_CHECK_LAUNCH();
    }
    sp.partial = reinterpret_cast<float*>(p.workspace); sp.splits = splits; sp.tile_begin = full; sp.tile_count = ntiles - full;
    hipLaunchKernelGGL((gemm_conv_glds_kernel<BN, false, MODE, GSTAGES>), dim3(sp.tile_count, splits), dim3(GNT), lds, stream, p, sp);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL((splitk_reduce_kernel<BN>), dim3((GBM * (BN / 4) + 255) / 256, sp.tile_count), dim3(256), 0, stream, p, sp);
    DC_CHECK_LAUNCH();
    return 0;
}

template <int BN, int GSTAGES>
int launch_glds_mode(const DcGemmParams& p, hipStream_t stream) {
    if (p.mode == 0) return launch_glds<BN, false, 0, GSTAGES>(p, stream);
    if (p.mode == 1) return p.ups ? launch_glds<BN, false, 3, GSTAGES>(p, stream) : launch_glds<BN, false, 1, GSTAGES>(p, stream);
    return launch_glds<BN, false, 2, GSTAGES>(p, stream);
}

inline int persist_max_k() {
    static const int v = [] { const char* e = getenv("DC_GEMM_PERSIST_MAXK"); return e ? atoi(e) : 2560; }();
    return v;
}

// fraction of the chip's workgroup slots a grid of `wgs` one-per-CU workgroups keeps busy
inline float wave_eff(int wgs) { return (float)wgs / (float)(((wgs + 255) / 256) * 256); }

}  // namespace

// Returns -100 when no large-tile variant applies (caller falls back to the 128-row kernel).
int dc_gemm_conv_glds_try(const DcGemmParams& p, hipStream_t stream) {
    const bool geglu = (p.flags & DC_GEMM_GEGLU) != 0;
    const int n_out = geglu ? p.N / 2 : p.N;
    const int tiles_m = (p.M + GBM - 1) / GBM;
    static const int force = [] { const char* e = getenv("DC_GEMM_TILE"); return e ? atoi(e) : 0; }();
    static const int persist = [] { const char* e = getenv("DC_GEMM_PERSIST"); return e ? atoi(e) : 1; }();
    if (geglu && p.mode != 0) return DC_ERR_ARG;
    // the persistent kernel's row-major epilogue moves 16 bytes per lane
    const bool out_f32 = (p.flags & DC_GEMM_OUT_F32) != 0;
    const bool epi16 = out_f32 || ((n_out % 8 == 0) && (p.ldc % 8 == 0) && ((long long)p.M * p.ldc < (1ll << 31)) &&
                                   ((long long)p.M * p.ldr < (1ll << 31)) && ((uintptr_t)p.C % 16 == 0) &&
                                   (!p.residual || ((p.ldr % 8 == 0) && ((uintptr_t)p.residual % 16 == 0))));
    // the plan is read ONCE per dispatch: the label (dc_note_variant) and the kernel always agree, whatever a concurrent
    // dc_gemm_set_plan does
    const int plan = g_gemm_plan.load(std::memory_order_relaxed);
    const bool use_pipe = pipe_ok(p, plan);
    const bool prefer_pipe = (force == 0) && use_pipe;
    if (persist && force == 0 && p.mode == 0 && epi16 && p.K <= persist_max_k() && !prefer_pipe) {
        // one workgroup per CU; needs at least 2 output tiles per workgroup to have anything to overlap
        static const int wide = [] { const char* e = getenv("DC_GEMM_PERSIST_WIDE"); return e ? atoi(e) : 1; }();
        // ping-pong kernel: measured ahead of the 8-wave kernel at K = 640 (547-560 vs 577-584 us on [73728 x 640 -> 2 x 2560]), level
        // with it at K = 1280, behind on the 18-row-panel level-3 launch (profiles/r04_pp_ab.txt): K <= 640 only
        // (plan bit 5 lifts the K limit: tests and same-box A/B)
        if (geglu && (plan & 16) && ((plan & 32) || p.K <= 640) && pp_ok(p) && tiles_m * (n_out / 64) >= 1024) return launch_pp(p, stream);
        if (geglu) {
            // each workgroup re-streams its A panel once per N tile: the wider tile halves that traffic
            if (wide && n_out % 128 == 0 && tiles_m * (n_out / 128) >= 512) return launch_persist<256, true>(p, stream, 256);
            const int nt = tiles_m * (n_out / 64);
            if (nt >= 512) return launch_persist<128, true>(p, stream, 256);
        } else {
            // N = 320*k: one 320-wide tile per A panel -> the activation rows cross the CU's load path once
            // (concurrent N tiles all stream the panel at HBM rate; only in-CU reuse is free)
            if (wide && p.N % 320 == 0 && p.n_pad >= p.N && tiles_m * (p.N / 320) >= 512)
                return launch_persist<320, false>(p, stream, 256);
            const int t128n = (p.N + 127) / 128;
            const float waste = (float)(t128n * 128) / (float)p.N;
            if (waste <= 1.15f && tiles_m * t128n >= 512) return launch_persist<128, false>(p, stream, 256);
            const int t64n = (p.N + 63) / 64;
            if (waste > 1.15f && tiles_m * t64n >= 512) return launch_persist<64, false>(p, stream, 256);
        }
    }
    static const int persist_conv_maxk = [] { const char* e = getenv("DC_GEMM_PERSIST_CONV_MAXK"); return e ? atoi(e) : 2880; }();
    if (persist && force == 0 && !prefer_pipe && !geglu && !out_f32 && epi16 && (p.mode == 2 || (p.mode == 1 && !p.ups)) &&
        p.N % 320 == 0 && p.n_pad >= p.N && tiles_m * (p.N / 320) >= (p.mode == 1 ? 1024 : 512) && p.K <= persist_conv_maxk)
        return launch_persist_conv320(p, stream);      // (3x3 convs with 512..1023 tiles measured faster on the split-K plan:
                                                       //  [147456x320x2880] 684 vs 572 TF/s, [73728x640x2880] 890 vs 734)
    if (geglu) {
        const int w256 = (n_out % 128 == 0 && p.n_pad >= p.N) ? tiles_m * (n_out / 128) : 0;
        const int w128 = tiles_m * (n_out / 64);
        if ((force == 256 || (force == 0 && w256 >= 512 && wave_eff(w256) > 0.8f)) && w256 > 0)
            return launch_glds<256, true, 0, 2>(p, stream);
        if (w128 < 384) return -100;
        return launch_glds<128, true, 0, 3>(p, stream);
    }
    // candidates: 320-wide (exact for the UNet widths), 128-wide; relative MFMA-side efficiency estimates measured on
    // conv shapes: 320-wide ~1.25x the 128-wide tile when both fill the chip
    const int t128 = (p.N + 127) / 128;
    const bool n320 = (p.N % 320 == 0) && (p.n_pad >= p.N);
    const int w320 = n320 ? tiles_m * (p.N / 320) : 0;
    const int w128 = tiles_m * t128;
    const float waste128 = (float)(t128 * 128) / (float)p.N;
    // (3x3 convs on the one-wave kernel: 1.4x, and down to 128 tiles - the level-1 -> 2 Downsample conv [18432 x 640 x 5760], 144
    //  tiles, measured 168 us there against 224 us on the 128-row kernel, tools/gemm_bench.py --only down3x3)
    const bool pipe_conv = use_pipe && p.mode == 1 && force == 0;
    float s320 = n320 ? (pipe_conv ? 1.4f : 1.25f) * wave_eff(w320) : 0.f;
    float s128 = wave_eff(w128) / waste128;
    if (w320 < (pipe_conv ? 128 : 200)) s320 = 0.f;
    if (w128 < 384) s128 = 0.f;
    // split-K plans for the 320-wide tile (needs the caller's workspace)
    static const int splitk = [] { const char* e = getenv("DC_GEMM_SPLITK"); return e ? atoi(e) : 1; }();
    if (splitk && force == 0 && n320 && p.workspace && p.n_pad >= p.N) {
        const int nk = p.K / GBK;
        int full = 0, splits = 0;
        if (w320 < 200 && w320 >= 8) {
            // few tiles (UNet level 3): cut every tile so that tiles x splits just fills the chip
            splits = 256 / w320;
            if (splits > 12) splits = 12;
            while (splits > 2 && nk / splits < 12) --splits;
        } else if (w320 > 256 && w320 <= 1280 && (w320 % 256) != 0 && (w320 % 256) <= 128) {
            // whole waves of 256 tiles, then the partial last wave cut along K so that it fills the chip
            // (level 2: 288 = 256 + 32 x 8; level 1, N = 640: 576 = 512 + 64 x 4)
            full = (w320 / 256) * 256;
            splits = 256 / (w320 - full);
            if (splits > 8) splits = 8;
            while (splits > 2 && nk / splits < 8) --splits;
        }
        const size_t need = (size_t)splits * (size_t)(w320 - full) * GBM * 320 * sizeof(float);
        if (splits >= 2 && nk / splits >= (full ? 8 : 12) && need <= (size_t)p.workspace_bytes) {
            const bool merged = !(plan & 64);
            if (p.mode == 0) return launch_glds320_split<0>(p, stream, w320, full, splits, use_pipe, merged);
            if (p.mode == 1) return p.ups ? launch_glds320_split<3>(p, stream, w320, full, splits, use_pipe, merged)
                                          : launch_glds320_split<1>(p, stream, w320, full, splits, use_pipe, merged);
            return launch_glds320_split<2>(p, stream, w320, full, splits, use_pipe, merged);
        }
    }
    // the same plan for the 256-wide tile (the AutoencoderKL's widths): its 72 x 128 level at 4 frames per call is 144 row tiles x 2 = 288
    // tiles of 256 x 256 = 256 whole + 32 x 8 K ranges instead of 576 tiles of 256 x 128 in 2.25 rounds (DC_GEMM_SPLITK256=0: off)
    static const int splitk256 = [] { const char* e = getenv("DC_GEMM_SPLITK256"); return e ? atoi(e) : 1; }();
    if (splitk && splitk256 && force == 0 && !n320 && !out_f32 && p.workspace && p.N % 256 == 0 && p.n_pad >= p.N && (p.mode != 1 || !p.ups)) {
        const int w256s = tiles_m * (p.N / 256), nk = p.K / GBK;
        if (w256s > 256 && w256s <= 1280 && (w256s % 256) != 0 && (w256s % 256) <= 128) {
            const int full = (w256s / 256) * 256;
            int splits = 256 / (w256s - full);
            if (splits > 8) splits = 8;
            while (splits > 2 && nk / splits < 8) --splits;
            const size_t need = (size_t)splits * (size_t)(w256s - full) * GBM * 256 * sizeof(float);
            if (splits >= 2 && nk / splits >= 8 && need <= (size_t)p.workspace_bytes) {
                if (p.mode == 0) return launch_glds320_split<0, 256>(p, stream, w256s, full, splits, false, false);
                if (p.mode == 1) return launch_glds320_split<1, 256>(p, stream, w256s, full, splits, false, false);
                return launch_glds320_split<2, 256>(p, stream, w256s, full, splits, false, false);
            }
        }
    }
    if (force == 320 && n320) return use_pipe ? launch_pipe_whole(p, stream) : launch_glds_mode<320, 2>(p, stream);
    if (force == 128 && w128 > 0 && waste128 <= 1.15f) return launch_glds_mode<128, 3>(p, stream);
    if (force == 1) return -100;
    // 256-wide plain tile for the AutoencoderKL widths (N = 256 / 512: not multiples of 320)
    const int w256 = (!n320 && p.N % 256 == 0 && p.n_pad >= p.N) ? tiles_m * (p.N / 256) : 0;
    if ((force == 256 || force == 0) && w256 >= 200 && 1.2f * wave_eff(w256) >= s128) return launch_glds_mode<256, 2>(p, stream);
    if (s320 > 0.f && s320 >= s128) return use_pipe ? launch_pipe_whole(p, stream) : launch_glds_mode<320, 2>(p, stream);
    if (s128 > 0.f && waste128 <= 1.15f) return launch_glds_mode<128, 3>(p, stream);
    return -100;
}

int dc_gemm_plan_now() { return g_gemm_plan.load(std::memory_order_relaxed); }

extern "C" int dc_gemm_set_plan(int plan) {
    if (!plan_valid(plan)) return DC_ERR_ARG;
    return g_gemm_plan.exchange(plan & PLAN_MASK, std::memory_order_relaxed);
}
