// conv3x3 (stride 1, pad 1) as 256 x 320 tiles with the ACTIVATION WINDOW of a tile resident in LDS: the nine taps of a
// channel slice read the same input rows (shifted by (dy-1) W + (dx-1)), so the 258 + 2 W rows a tile touches are staged
// ONCE per 32-channel half slice and the taps read them at shifted LDS rows. (Included by gemm_conv_glds.hip after
// gemm_pipe.h; same one-wave-per-SIMD structure, counters instead of barriers.)
//
// Why: gemm_pipe320_kernel's tile loop runs at 79 % of the MFMA rate IN CLOCKS, but the chip holds 1.59 GHz under it
// (tools/pipe_stamps.py: 2.22 GHz for the MFMA stream alone, 2.05 GHz with the weight DMA + fragment reads, 1.81 GHz when
// the activation bytes come by LDS-DMA, 1.59 GHz as loads into registers): the convs are POWER-bound and what costs is
// bringing 32 KB of activations per K tile from L2 into every CU, nine times over. Here that traffic is 258 + 2 W rows x
// 64 B per NINE K tiles (4.5 - 8 x less), the rest comes out of LDS.
//
// Layout of a workgroup (4 waves, one per SIMD): 2 x 2 waves, a wave owns 128 rows x 160 columns (20 accumulator blocks as
// in gemm_pipe.h; 4 activation + 5 weight fragments per 20 MFMAs instead of 2 + 10: a quarter less LDS traffic).
// K order: (64-channel slice, 32-channel half, tap) - a "window" = one half slice = 9 K tiles of 32.
//   LDS: weight ring 3 x [320 rows][64 B] (20 KB stages), two activation windows of 528 rows x 64 B, 8 KB of zeros (padded
//   taps and tail rows read there), epilogue patches, four counters.
// Per K tile (40 MFMAs) a wave issues 10 weight-fragment reads, 8 activation-fragment reads, 5 LDS-DMA pieces of the weight
// tile two ahead and - in tiles 1..5 of a window - the 9 pieces of its share of the NEXT window.
// Synchronisation (all LDS counters, posted well before they are needed):
//   landed / freed    the weight ring, as in gemm_pipe.h (per K tile);
//   a_landed          a wave's 9 pieces of the next window are in LDS (posted in tile 7 behind a counted vmcnt);
//                     checked in tile 8, before the first fragment of the next window is requested;
//   a_freed           a wave has requested its last fragment of a window (end of tile 8); checked in tile 1 of the next
//                     window, before the buffer is overwritten with the window after it.
#pragma once

constexpr int CP_BST = 320 * 64;                       // one weight tile
constexpr int CP_WROWS = 528;                          // window rows (33 pieces of 16): 258 + 2 W <= 528 -> W <= 135
constexpr int CP_WBYTES = CP_WROWS * 64;
constexpr int CP_OFF_A = 3 * CP_BST;                   // 61440
constexpr int CP_OFF_Z = CP_OFF_A + 2 * CP_WBYTES;     // 129024
constexpr int CP_OFF_P = CP_OFF_Z + 8192;              // 137216
constexpr int CP_OFF_C = CP_OFF_P + 8192;              // 145408
constexpr int CP_LDS = CP_OFF_C + 64;

__device__ __forceinline__ void cp_settle(f32x16_t (&acc)[2][2][5]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+a"(acc[0][0][0]), "+a"(acc[0][0][1]), "+a"(acc[0][0][2]), "+a"(acc[0][0][3]), "+a"(acc[0][1][0]), "+a"(acc[0][1][1]),
                   "+a"(acc[0][1][2]), "+a"(acc[0][1][3]), "+a"(acc[1][0][0]), "+a"(acc[1][0][1]), "+a"(acc[1][0][2]), "+a"(acc[1][0][3]),
                   "+a"(acc[1][1][0]), "+a"(acc[1][1][1]), "+a"(acc[1][1][2]), "+a"(acc[1][1][3]), "+v"(acc[0][0][4]), "+v"(acc[0][1][4]),
                   "+v"(acc[1][0][4]), "+v"(acc[1][1][4]));
}

// A wait on an LDS counter. Every wave posts every counter the same number of times, so a wait always ends; the bound (about
// 10 ms, once per wave) only keeps a future bookkeeping mistake from hanging the GPU - the results are then wrong, loudly.
#define CP_SPIN(cond, reread)                                              \
    do {                                                                   \
        int spins__ = 0;                                                   \
        while (!gave_up && (cond)) { reread; if (++spins__ > 200000) gave_up = 1; } \
    } while (0)

// vector-memory operations a wave issues in tile i of a window: 5 weight pieces, then this many window pieces
__host__ __device__ constexpr int cp_na(int i) { return (i >= 1 && i <= 4) ? 2 : (i == 5 ? 1 : 0); }

template <int EPI>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv3_pipe320_kernel(const DcGemmParams p, const GemmSplit sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;

    const int tiles_n = p.N / 320;
    const int swz = sp.tile_begin + xcd_remap(blockIdx.x, sp.tile_count);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int m0 = tile_m * GBM;
    const int n0 = tile_n * 320;
    const int W = p.IW;

    // this workgroup's windows (half slices): all of them, or the blockIdx.y-th of sp.splits near-equal ranges
    const int nwin_all = p.Cin / 32;
    const int win_lo = (int)(((long long)blockIdx.y * nwin_all) / sp.splits);
    const int win_hi = (int)(((long long)(blockIdx.y + 1) * nwin_all) / sp.splits);
    const int nwin = win_hi - win_lo;

    gp_lds_int_t* const cnt = (gp_lds_int_t*)(smem + CP_OFF_C);     // [0] landed [1] freed [2] a_landed [3] a_freed
    if (tid < 4) cnt[tid] = 0;
    for (int i = tid; i < 8192 / 16; i += 256) *reinterpret_cast<u32x4_t*>(smem + CP_OFF_Z + i * 16) = u32x4_t{0u, 0u, 0u, 0u};

    const unsigned lds_base = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
    const unsigned lda2 = (unsigned)p.lda * 2u;
    auto dma = [&](unsigned lds_dst, unsigned voff, unsigned long long sbase) __attribute__((always_inline)) {
#ifdef CP_DBG_NODMA
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    };

    // ---- weight tile (window w, tap): columns (9 cs + tap) 64 + 32 h of the packed [N][9 Cin] matrix; piece q = 4 j + wave is
    // 16 rows x 64 B, lane-linear in LDS, the 16-byte chunks of a row XOR-swizzled by (row >> 2) & 3 on the source side (64-byte rows:
    // the 16 lanes of a ds_read_b128 group hold rows {0-3, 12-15, 20-27} + const and must fall on 16 distinct 16-byte slots of 256 B)
    unsigned voffB[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int row = (j * 4 + wave) * 16 + (lane >> 2);
        voffB[j] = (unsigned)row * (unsigned)p.K * 2u + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        asm volatile("" : "+v"(voffB[j]));
    }
    auto w_src = [&](int w, int tap) __attribute__((always_inline)) -> unsigned long long {
        const int col = ((w >> 1) * 9 + tap) * 64 + (w & 1) * 32;
        return (unsigned long long)(uintptr_t)p.W + ((unsigned long long)n0 * p.K + (unsigned long long)col) * 2ull;
    };
    // ---- activation window w: input rows m0 - W - 1 + j (clamped into the tensor: rows outside are only ever read by taps that
    // are masked), bytes [128 cs + 64 h, +64); piece (4 a + wave, at most 32) = 16 rows
    unsigned voffA[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        int pc = 4 * a + wave; if (pc > 32) pc = 32;
        int r = m0 - W - 1 + pc * 16 + (lane >> 2);
        r = r < 0 ? 0 : (r > p.M - 1 ? p.M - 1 : r);
        voffA[a] = (unsigned)r * lda2 + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
        asm volatile("" : "+v"(voffA[a]));
    }
    auto a_src = [&](int w) __attribute__((always_inline)) -> unsigned long long {
        return (unsigned long long)(uintptr_t)p.A + (unsigned long long)((w >> 1) * 128 + (w & 1) * 64);
    };
    auto a_dst = [&](int par, int a) __attribute__((always_inline)) -> unsigned {
        int pc = 4 * a + wave; if (pc > 32) pc = 32;
        return lds_base + CP_OFF_A + par * CP_WBYTES + pc * 1024;
    };

    // ---- fragment addresses. Weights: fragment (s, nb) = rows 160 wc + 32 nb + fr, chunk 2 s + fh
    unsigned bofs[3][2];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bofs[st][s] = lds_base + st * CP_BST + (wc * 160 + fr) * 64 + (((2 * s + fh) ^ ((fr >> 2) & 3)) << 4);
            asm volatile("" : "+v"(bofs[st][s]));
        }
    // Activations: output row 128 wr + 32 mb + fr of the tile reads window row j = (that) + dy W + dx; validity of the nine taps
    int mask[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int m = m0 + wr * 128 + mb * 32 + fr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int ohw = p.OH * p.OW;
        const int rem = mm - (mm / ohw) * ohw;
        const int oy = rem / p.OW;
        const int ox = rem - oy * p.OW;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1 << t;
        }
        mask[mb] = mk;
    }
    const int jbase = wr * 128 + fr;                       // + 32 mb: an immediate (2048 B) of the fragment read
    const unsigned zsel = lds_base + CP_OFF_Z;
    // addresses of the four activation fragments of k step s for tap TAP out of the window buffer `par`: sel[mb] + 2048 mb
    auto calc_sel = [&](int tap, int s, int par, unsigned (&sel)[4]) __attribute__((always_inline)) {
        const int dy = tap / 3, dx = tap - dy * 3;
        const int jj = jbase + dy * W + dx;
        const unsigned a = lds_base + CP_OFF_A + par * CP_WBYTES + (unsigned)jj * 64u + (unsigned)((((2 * s + fh) << 4)) ^ ((jj << 2) & 0x30));
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) sel[mb] = ((mask[mb] >> tap) & 1) ? a : zsel;
    };

    bf16x8_t Bf[3];
    bf16x8_t Af[2][4];
    auto rd_b = [&](int slot, unsigned addr, int nb) __attribute__((always_inline)) {
#ifdef CP_DBG_NOLDS
        return;
#endif
        Bf[slot] = *(gp_lds_frag_t*)((const __attribute__((address_space(3))) char*)(uintptr_t)addr + nb * 2048);
    };
    auto rd_a = [&](int s, int mb, unsigned addr) __attribute__((always_inline)) {
#ifdef CP_DBG_NOLDS
        return;
#endif
        Af[s][mb] = *(gp_lds_frag_t*)((const __attribute__((address_space(3))) char*)(uintptr_t)addr + mb * 2048);
    };

    f32x16_t acc[2][2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][k][j][r] = 0.f;

    // ---- prologue: the first window, weight tiles 0 and 1; plain barriers here
    {
        const unsigned long long as = a_src(win_lo);
#pragma unroll
        for (int a = 0; a < 9; ++a) dma(a_dst(0, a), voffA[a], as);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned long long ws = w_src(win_lo, t);
#pragma unroll
            for (int j = 0; j < 5; ++j) dma(lds_base + t * CP_BST + (j * 4 + wave) * 1024, voffB[j], ws);
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (tid == 0) { cnt[0] = 4; cnt[2] = 4; }              // as if every wave had posted weight tile 0 and window 0 (tile 1 is posted in tile 0)
    __syncthreads();
    unsigned selS0[4], selS1[4];
    calc_sel(0, 0, 0, selS0);
    calc_sel(0, 1, 0, selS1);
    rd_b(0, bofs[0][0], 0);
    rd_b(1, bofs[0][0], 1);
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) rd_a(0, mb, selS0[mb]);
    cp_settle(acc);

    int seen_f = 0, seen_l = 0, seen_af = 0, seen_al = 0, gave_up = 0;
    // One K tile = tap TAP of window index wi (local), buffer parity par. t = 9 wi + TAP.
    auto tile = [&](auto TAP_, int wi, int par) __attribute__((always_inline)) {
        constexpr int TAP = decltype(TAP_)::value;
        constexpr int ST = TAP % 3, ST1 = (TAP + 1) % 3, ST2 = (TAP + 2) % 3;
        constexpr int TAPN = (TAP + 1) % 9;
        const int t = wi * 9 + TAP;
        const int parn = (TAP == 8) ? (par ^ 1) : par;      // buffer of the next tile's window
        // weight tile two ahead: (window, tap) of K tile t + 2, clamped to the last tile of the range
        int w2 = win_lo + wi, tap2 = TAP + 2;
        if (tap2 >= 9) { tap2 -= 9; w2 += 1; }
        if (w2 >= win_hi) { w2 = win_hi - 1; tap2 = 8; }
        const int wnext = (win_lo + wi + 1 < win_hi) ? win_lo + wi + 1 : win_hi - 1;      // window staged during this one
        unsigned long long ws = 0, as = 0;
        f32x16_t (&acc1)[2][2][5] = acc;
        bf16x8_t (&Bf1)[3] = Bf;
        bf16x8_t (&Af1)[2][4] = Af;
        unsigned (&s0)[4] = selS0;
        unsigned (&s1)[4] = selS1;
        gp_for(std::make_integer_sequence<int, 40>{}, [&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            f32x16_t (&acc_)[2][2][5] = acc1;
            bf16x8_t (&Bf_)[3] = Bf1;
            bf16x8_t (&Af_)[2][4] = Af1;
            if constexpr ((g & 3) == 0) {
                // weight fragment two ahead of the one the next four MFMAs use; the last two reads are the next tile's first
                constexpr int F = g / 4 + 2;
                if constexpr (F < 10) rd_b((TAP * 10 + F) % 3, bofs[ST][F / 5], F % 5);
                else rd_b((TAP * 10 + F) % 3, bofs[ST1][0], F - 10);
            } else if constexpr ((g & 3) == 2) {
                // activation fragments: k step 1 of this tile (gaps 2..14), k step 0 of the next tile (gaps 22..34)
                if constexpr (g <= 14) rd_a(1, (g - 2) >> 2, s1[(g - 2) >> 2]);
                else if constexpr (g >= 22 && g <= 34) rd_a(0, (g - 22) >> 2, s0[(g - 22) >> 2]);
            } else if constexpr ((g & 3) == 1) {
                if constexpr (g == 1) {
                    calc_sel(TAPN, 0, parn, s0);            // (selS0 was last used in gap 34 of the previous tile)
                    ws = w_src(w2, tap2);
                    as = a_src(wnext);
                }
                if constexpr (g >= 5 && g <= 21) {
                    constexpr int j = (g - 5) >> 2;
                    dma(lds_base + ST2 * CP_BST + (j * 4 + wave) * 1024, voffB[j], ws);
                }
                if constexpr (g == 17) calc_sel(TAPN, 1, parn, s1);      // (selS1 was last used in gap 14)
                if constexpr ((g == 25 && cp_na(TAP) >= 1) || (g == 29 && cp_na(TAP) >= 2)) {
                    constexpr int a = (TAP - 1) * 2 + (g == 29 ? 1 : 0);
                    dma(a_dst(par ^ 1, a), voffA[a], as);
                }
            } else {
                if constexpr (g == 3) {
                    // stage ST2 held tile t - 1: every wave is past its last fragment of it
                    seen_f = *(volatile gp_lds_int_t*)(cnt + 1);
#ifndef CP_DBG_NOSYNC
                    CP_SPIN(__builtin_amdgcn_readfirstlane(seen_f) < 4 * t, seen_f = *(volatile gp_lds_int_t*)(cnt + 1));
#endif
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 7 && TAP == 1) seen_af = *(volatile gp_lds_int_t*)(cnt + 3);
                if constexpr (g == 19 && TAP == 1) {
                    // the other window buffer held window wi - 1: every wave has requested its last fragment of it
#ifndef CP_DBG_NOSYNC
                    CP_SPIN(__builtin_amdgcn_readfirstlane(seen_af) < 4 * wi, seen_af = *(volatile gp_lds_int_t*)(cnt + 3));
#endif
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 11 && TAP == 8) seen_al = *(volatile gp_lds_int_t*)(cnt + 2);
                if constexpr (g == 19 && TAP == 8) {
                    // all four shares of the next window have landed (its first fragment is requested in gap 22)
#ifndef CP_DBG_NOSYNC
                    CP_SPIN(wi + 1 < nwin && __builtin_amdgcn_readfirstlane(seen_al) < 4 * (wi + 2), seen_al = *(volatile gp_lds_int_t*)(cnt + 2));
#endif
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 23) {
                    // this wave's pieces of tile t + 1 (issued a tile ago) are in LDS; in tile 7 so are its pieces of the next window
                    wait_vmcnt<cp_na((TAP + 8) % 9) + 5>();
                    if (lane == 0) {
                        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if constexpr (TAP == 7) __hip_atomic_fetch_add(cnt + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if constexpr (g == 27) seen_l = *(volatile gp_lds_int_t*)cnt;
                if constexpr (g == 31) {
                    // all four shares of tile t + 1 have landed (it is first read in the next gap)
#ifndef CP_DBG_NOSYNC
                    CP_SPIN(t + 1 < 9 * nwin && __builtin_amdgcn_readfirstlane(seen_l) < 4 * (t + 2), seen_l = *(volatile gp_lds_int_t*)cnt);
#endif
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 35) {
                    if (lane == 0) __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if constexpr (g == 39 && TAP == 8) {
                    if (lane == 0) __hip_atomic_fetch_add(cnt + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            constexpr int s = g / 20, nb = (g % 20) >> 2, mb = g & 3;
            if constexpr (nb < 4) GP_MFMA_A(acc_[mb >> 1][mb & 1][nb], Bf_[(TAP * 10 + s * 5 + nb) % 3], Af_[s][mb]);
            else GP_MFMA_V(acc_[mb >> 1][mb & 1][nb], Bf_[(TAP * 10 + s * 5 + nb) % 3], Af_[s][mb]);
        });
    };
#ifdef GP_STAMPS
    const unsigned long long st_loop0 = __builtin_readcyclecounter();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int wi = 0; wi < nwin; ++wi) {
        const int par = wi & 1;
        tile(gp_ic<0>{}, wi, par);
        tile(gp_ic<1>{}, wi, par);
        tile(gp_ic<2>{}, wi, par);
        tile(gp_ic<3>{}, wi, par);
        tile(gp_ic<4>{}, wi, par);
        tile(gp_ic<5>{}, wi, par);
        tile(gp_ic<6>{}, wi, par);
        tile(gp_ic<7>{}, wi, par);
        tile(gp_ic<8>{}, wi, par);
    }
    wait_vmcnt<0>();
    cp_settle(acc);
#ifdef GP_STAMPS
    if (lane == 0 && p.workspace && !sp.partial && (size_t)(blockIdx.x * 4 + wave + 1) * 64 <= (size_t)p.workspace_bytes) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(p.workspace) + (size_t)(blockIdx.x * 4 + wave) * 8;
        out[0] = 0; out[1] = 0; out[2] = 0; out[3] = (unsigned long long)gave_up;
        out[4] = __builtin_readcyclecounter() - st_loop0; out[5] = (unsigned long long)(9 * nwin / 2);      // in 80-MFMA units
        out[6] = __builtin_amdgcn_s_memrealtime() - st_real0;
    }
#endif

    if (sp.partial) {
        // split-K: raw fp32 accumulators, [split][tile][256][320]; splitk_reduce_kernel applies the epilogue
        const size_t slot = (size_t)blockIdx.y * sp.tile_count + (size_t)(swz - sp.tile_begin);
        float* const dst = sp.partial + (slot * GBM + wr * 128 + fr) * 320 + wc * 160 + 4 * fh;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 5; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16_t& b = acc[mb >> 1][mb & 1][nb];
                    *reinterpret_cast<float4*>(dst + (size_t)mb * 32 * 320 + nb * 32 + 8 * q) = make_float4(b[4 * q], b[4 * q + 1], b[4 * q + 2], b[4 * q + 3]);
                }
        return;
    }
    persist_epilogue<320, false, EPI>(acc[0], p, m0, n0, p.N, wr * 2, wc, lane, smem + CP_OFF_P + wave * 2048);
    persist_epilogue<320, false, EPI>(acc[1], p, m0, n0, p.N, wr * 2 + 1, wc, lane, smem + CP_OFF_P + wave * 2048);
}

template <int EPI>
int launch_conv_pipe(const DcGemmParams& p, hipStream_t stream, const GemmSplit& sp, int grid_x, int grid_y) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&conv3_pipe320_kernel<EPI>), CP_LDS)) return e;
    hipLaunchKernelGGL((conv3_pipe320_kernel<EPI>), dim3(grid_x, grid_y), dim3(256), CP_LDS, stream, p, sp);
    DC_CHECK_LAUNCH();
    return 0;
}
