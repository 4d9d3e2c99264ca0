// 256 x 320 x 64 GEMM / implicit-GEMM conv tiles with ONE wave per SIMD and a hand-placed instruction stream (included by
// gemm_conv_glds.hip, inside its namespace: shares GemmSplit, splitk_reduce_kernel and persist_epilogue).
//
// Why a third GEMM kernel: the 8-wave LDS-DMA kernel keeps the matrix pipe 42 % busy on the long-K convs: its
// two waves per SIMD meet at one s_barrier per K tile, the prefetch distance is one K tile, and hipcc orders the K step.
// What flash_pipe.hip showed for attention holds for a GEMM with less effort (no vector work beside the MFMAs):
//   * a workgroup is 4 waves, a wave owns 64 rows x all 320 columns (20 accumulator blocks = 320 registers, 256 of them in
//     the accumulator file) and is alone on its SIMD;
//   * the ACTIVATION operand never touches LDS: a lane's fragment of K step s is 16 contiguous bytes of its row, so the
//     wave loads its own 64 rows straight into registers (buffer loads: per-lane row offset in a VGPR, tap / channel-slice
//     offset in an SGPR, padded taps and tail rows through the out-of-range rule = zeros), THREE K tiles deep;
//   * only the WEIGHT tile (320 x 64, 40 KB) goes through LDS: LDS-DMA into a ring of 3 stages, every wave reads all of it
//     (40 ds_read_b128 per 80 MFMAs);
//   * no s_barrier in the K loop: two LDS counters. `landed`: a wave adds 1 when ITS share of a weight tile has landed
//     (counted s_waitcnt vmcnt in the middle of the previous tile); a tile is first read when 4 x (tile + 1) arrivals are
//     seen. `freed`: a wave adds 1 behind its last fragment read of a tile; the stage is overwritten when all four have.
//     Both are posted most of a tile before they are needed, so waves drift instead of meeting;
//   * every instruction of the tile loop is an asm volatile statement or a volatile LDS load: the order is the source
//     order. Even gaps between MFMAs carry one fragment read (4 fragments = 8 MFMAs ahead, ring of 5 registers sets), odd
//     gaps one vector-memory instruction or one counter operation.
// vmcnt is counted by hand: per tile a wave issues 10 LDS-DMA pieces (weights of tile t+2), then 8 buffer loads
// (activations of tile t+2), always in this order, so "tile t's activations are here" and "my pieces of tile t+1 are in
// LDS" are both vmcnt(18).
#pragma once

constexpr int GP_STAGE = 320 * 64 * 2;            // one weight tile: [320 rows][128 B], 16-byte chunks XOR-swizzled by row pair
constexpr int GP_NST = 3;
constexpr int GP_RING = GP_NST * GP_STAGE;        // 120 KB
constexpr int GP_CNT = GP_RING + 4 * 2048;        // behind the four epilogue patches
#ifdef GP_DBG_A_DMA
constexpr int GP_LDS = GP_CNT + 64 + 24 * 1024;
#else
constexpr int GP_LDS = GP_CNT + 64;
#endif

typedef __attribute__((ext_vector_type(4))) int gp_i32x4_t;
typedef const volatile __attribute__((address_space(3))) bf16x8_t gp_lds_frag_t;
typedef __attribute__((address_space(3))) int gp_lds_int_t;

#ifdef GP_STAMPS    // tool build (tools/pipe_stamps.py): shader clocks a wave spends at its four waiting points, summed over the K loop
#define GP_ST_BEGIN() const unsigned long long st_t0__ = __builtin_readcyclecounter()
#define GP_ST_END(i) st_acc[i] += __builtin_readcyclecounter() - st_t0__
#else
#define GP_ST_BEGIN() do { } while (0)
#define GP_ST_END(i) do { } while (0)
#endif
// A wait on an LDS counter. Every wave posts every counter the same number of times, so a wait always ends; the bound (about
// 10 ms, once per wave) only keeps a future bookkeeping mistake from hanging the GPU - the results are then wrong, loudly.
#define GP_SPIN(cond, reread)                                              \
    do {                                                                   \
        int spins__ = 0;                                                   \
        while (!gave_up && (cond)) { reread; if (++spins__ > 200000) gave_up = 1; } \
    } while (0)

template <int V> using gp_ic = std::integral_constant<int, V>;
template <int... G, class F>
__device__ __forceinline__ void gp_for(std::integer_sequence<int, G...>, F&& f) { (f(gp_ic<G>{}), ...); }

#define GP_MFMA_A(d, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(w), "v"(x))
#define GP_MFMA_V(d, w, x) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(w), "v"(x))
#define GP_LOAD_A_(dst, vo, rs, so, IMM) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:" IMM : "=&v"(dst) : "v"(vo), "s"(rs), "s"(so))
// K step s of the tile = bytes [32 s, 32 s + 32) of the row's 128-byte slice (the immediate must be a literal)
#ifdef GP_DBG_NOA
#define GP_LOAD_A(dst, vo, rs, so, s) do { } while (0)
#else
#define GP_LOAD_A(dst, vo, rs, so, s)                          \
    do {                                                       \
        if ((s) == 0) GP_LOAD_A_(dst, vo, rs, so, "0");        \
        else if ((s) == 1) GP_LOAD_A_(dst, vo, rs, so, "32");  \
        else if ((s) == 2) GP_LOAD_A_(dst, vo, rs, so, "64");  \
        else GP_LOAD_A_(dst, vo, rs, so, "96");                \
    } while (0)
#endif

// padding statement between compiler-generated accesses of the accumulators and the asm MFMAs (hipcc pads neither
// direction for an asm statement)
__device__ __forceinline__ void gp_settle(f32x16_t (&acc)[2][10]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]),
                   "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]),
                   "+a"(acc[1][4]), "+a"(acc[1][5]), "+a"(acc[1][6]), "+a"(acc[1][7]), "+v"(acc[0][8]), "+v"(acc[0][9]),
                   "+v"(acc[1][8]), "+v"(acc[1][9]));
}

// MODE 0 plain rows | 1 conv3x3 (any stride / pad, no upsampling) | 2 temporal 3-tap conv.  EPI 0 bf16 | 1 bf16 + residual.
template <int MODE, int EPI>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_pipe320_kernel(const DcGemmParams p, const GemmSplit sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    const int tiles_n = p.N / 320;
    const int swz = sp.tile_begin + xcd_remap(blockIdx.x, sp.tile_count);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int m0 = tile_m * GBM;
    const int n0 = tile_n * 320;

    const int nk_all = p.K / GBK;
    const int kt_lo = (int)(((long long)blockIdx.y * nk_all) / sp.splits);
    const int kt_hi = (int)(((long long)(blockIdx.y + 1) * nk_all) / sp.splits);
    const int nk = kt_hi - kt_lo;

    gp_lds_int_t* const cnt_landed = (gp_lds_int_t*)(smem + GP_CNT);
    gp_lds_int_t* const cnt_freed = cnt_landed + 1;
    if (tid < 2) cnt_landed[tid] = 0;
    __syncthreads();
    int gave_up = 0;                                        // see GP_SPIN

    // ---- activation rows: lane (fr, fh) holds bytes [32 s + 16 fh, +16) of K tile slices of rows 32 mb + fr of its wave
    const unsigned lda2 = (unsigned)p.lda * 2u;
    unsigned rowoff[2];
    int mask[2];
    long long bias = 0;                                     // the descriptor's base lies `bias` bytes in front of p.A
    if (MODE == 1) bias = (long long)(p.pad * p.IW + p.pad) * lda2;
    if (MODE == 2) bias = (long long)p.HW * lda2;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int m = m0 + wave * 64 + mb * 32 + fr;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if (MODE == 0) {
            rowoff[mb] = (unsigned)mm * lda2 + fh * 16;
            mask[mb] = ok ? 1 : 0;
        } else if (MODE == 1) {
            // offset of tap (0, 0) of this output row (+ bias: never negative) and a 9-bit validity mask
            const int ohw = p.OH * p.OW;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = iy0 + t / 3, ix = ix0 + t % 3;
                if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1 << t;
            }
            mask[mb] = mk;
            rowoff[mb] = (unsigned)(((n * p.IH + iy0 + p.pad) * p.IW + ix0 + p.pad)) * lda2 + fh * 16;
        } else {
            const int frame = (mm / p.HW) % p.T;
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tt = frame + t - 1;
                if (ok && tt >= 0 && tt < p.T) mk |= 1 << t;
            }
            mask[mb] = mk;
            rowoff[mb] = (unsigned)mm * lda2 + fh * 16;
        }
    }
    gp_i32x4_t ars;
    {
        const unsigned long long ab = (unsigned long long)(uintptr_t)p.A - (unsigned long long)bias;
        ars[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)ab);
        ars[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((ab >> 32) & 0xffffu));
        ars[2] = 0x7fffffff;
        ars[3] = 0x00020000;
    }
    // per-tile part of the activation address: tap / channel-slice offset (scalar) and the lanes' row offsets with padded
    // taps and tail rows sent out of range (= zeros)
    auto a_tile = [&](int kt, unsigned (&vo)[2], int& soff) __attribute__((always_inline)) {
        int tap = 0;
        if (MODE == 0) {
            soff = kt * 128;
        } else if (MODE == 1) {
            const int cs = kt / 9;
            tap = kt - cs * 9;
            const int dy = tap / 3, dx = tap - dy * 3;
            soff = (dy * p.IW + dx) * (int)lda2 + cs * 128;
        } else {
            const int cs = kt / 3;
            tap = kt - cs * 3;
            soff = tap * p.HW * (int)lda2 + cs * 128;
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) vo[mb] = ((mask[mb] >> tap) & 1) ? rowoff[mb] : 0x80000000u;
#if defined(GP_DBG_A_ROWS) || defined(GP_DBG_A_DMA)
        vo[0] = (unsigned)bias + (unsigned)((m0 & 1023) + wave * 64 + (lane >> 3)) * lda2 + (lane & 7) * 16;     // (stays inside the first rows)
#elif defined(GP_DBG_A_SAMELINE)
        vo[0] = (unsigned)bias + (unsigned)((m0 & 1023) + wave * 64) * lda2 + (lane & 7) * 16;
#endif
    };

    // ---- weight tile by LDS-DMA: piece q = 4 j + wave (8 rows x 128 B = 1 KB, lane-linear in LDS), the swizzle on the source
    const unsigned lds_base = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
    unsigned voffB[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int row = (j * 4 + wave) * 8 + (lane >> 3);
        voffB[j] = (unsigned)row * (unsigned)p.K * 2u + (unsigned)(((lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7)) << 4);
        asm volatile("" : "+v"(voffB[j]));
    }
    auto w_src = [&](int kt) __attribute__((always_inline)) -> unsigned long long {
        return (unsigned long long)(uintptr_t)p.W + ((unsigned long long)n0 * p.K + (unsigned long long)kt * 64) * 2ull;
    };
    auto dma = [&](unsigned lds_dst, unsigned voff, unsigned long long sbase) __attribute__((always_inline)) {
#ifdef GP_DBG_NODMA          // tool builds (tools/pipe_variants.sh): parts of the tile loop switched off, results wrong
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    };

    // ---- fragment reads: weight fragment (s, nb) = rows 32 nb + fr, 16-byte chunk 2 s + fh
    unsigned bofs[3][4];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bofs[st][s] = lds_base + st * GP_STAGE + fr * 128 + (((2 * s + fh) ^ ((fr >> 1) & 7)) << 4);
            asm volatile("" : "+v"(bofs[st][s]));
        }
    bf16x8_t Bf[5];
    auto rd_b = [&](int slot, unsigned addr, int nb) __attribute__((always_inline)) {
#ifdef GP_DBG_NOLDS
        return;
#endif
        Bf[slot] = *(gp_lds_frag_t*)((const __attribute__((address_space(3))) char*)(uintptr_t)addr + nb * 4096);
    };

    u32x4_t A[3][2][4];
    f32x16_t acc[2][10];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: operands of tiles 0 and 1 in the loop's order (10 pieces, 8 loads per tile)
    auto clampk = [&](int t) { return kt_lo + (t < nk ? t : nk - 1); };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned long long ws = w_src(clampk(t));
#pragma unroll
        for (int j = 0; j < 10; ++j) dma(lds_base + t * GP_STAGE + (j * 4 + wave) * 1024, voffB[j], ws);
        unsigned vo[2];
        int soff;
        a_tile(clampk(t), vo, soff);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (t == 0) GP_LOAD_A(A[0][i & 1][i >> 1], vo[i & 1], ars, soff, i >> 1);
            else GP_LOAD_A(A[1][i & 1][i >> 1], vo[i & 1], ars, soff, i >> 1);
        }
    }
    wait_vmcnt<26>();                                       // this wave's pieces of tile 0 are in LDS
    if (lane == 0) __hip_atomic_fetch_add(cnt_landed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
        int seen = *(volatile gp_lds_int_t*)cnt_landed;
        GP_SPIN(__builtin_amdgcn_readfirstlane(seen) < 4, seen = *(volatile gp_lds_int_t*)cnt_landed);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int f = 0; f < 4; ++f) rd_b(f, bofs[0][0], f);
    gp_settle(acc);

    int seen_f = 0, seen_l = 0;
#ifdef GP_STAMPS
    unsigned long long st_acc[4] = {0, 0, 0, 0};
    const unsigned long long st_loop0 = __builtin_readcyclecounter();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto tile = [&](auto ST_, int t) __attribute__((always_inline)) {
        constexpr int ST = decltype(ST_)::value, ST1 = (ST + 1) % 3, ST2 = (ST + 2) % 3;
        const int kt2 = clampk(t + 2);
        unsigned vo[2] = {0u, 0u};
        int soff = 0;
        unsigned long long ws = 0;
        f32x16_t (&acc1)[2][10] = acc;
        bf16x8_t (&Bf1)[5] = Bf;
        u32x4_t (&A1)[3][2][4] = A;
        // one MFMA and what is issued in front of it; a fold over the gap index, not a loop: hipcc unrolls an 80-trip loop of
        // this size only partially, and the register arrays then live in scratch
        gp_for(std::make_integer_sequence<int, 80>{}, [&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            // (asm operands do not capture in a generic lambda: name the state through references first)
            f32x16_t (&acc_)[2][10] = acc1;
            bf16x8_t (&Bf_)[5] = Bf1;
            u32x4_t (&A_)[3][2][4] = A1;
            if constexpr (g == 0) {
                GP_ST_BEGIN();
                wait_vmcnt<18>();                           // the activations of tile t (requested two tiles ago)
                GP_ST_END(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(A_[ST][i & 1][i >> 1]));
            }
            if constexpr (!(g & 1)) {
                // fragment 4 ahead of the one the next two MFMAs use; the last four reads are the next tile's first fragments
                constexpr int F = g / 2 + 4;
                if constexpr (F < 40) rd_b(F % 5, bofs[ST][F / 10], F % 10);
                else rd_b(F % 5, bofs[ST1][0], F - 40);
            } else {
                if constexpr (g == 1) seen_f = *(volatile gp_lds_int_t*)cnt_freed;
                if constexpr (g == 3) {
                    // stage ST2 held tile t-1: every wave is past its last fragment of it
                    GP_ST_BEGIN();
#ifndef GP_DBG_NOSYNC
                    GP_SPIN(__builtin_amdgcn_readfirstlane(seen_f) < 4 * t, seen_f = *(volatile gp_lds_int_t*)cnt_freed);
#endif
                    GP_ST_END(2);
                    asm volatile("" ::: "memory");
                    ws = w_src(kt2);
                }
                if constexpr (g >= 5 && g <= 41 && ((g - 5) & 3) == 0) {
                    constexpr int j = (g - 5) >> 2;
                    dma(lds_base + ST2 * GP_STAGE + (j * 4 + wave) * 1024, voffB[j], ws);
                }
                if constexpr (g == 43) {
                    GP_ST_BEGIN();
                    wait_vmcnt<18>();                       // this wave's pieces of tile t+1 (issued a tile ago) are in LDS
                    GP_ST_END(1);
                    if (lane == 0) __hip_atomic_fetch_add(cnt_landed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a_tile(kt2, vo, soff);
                }
                if constexpr (g >= 45 && g <= 73 && ((g - 45) & 3) == 0) {
                    constexpr int i = (g - 45) >> 2;
#if defined(GP_DBG_A_DMA)            // tool build: the activation bytes by LDS-DMA into a spare LDS area instead of registers
                    dma(lds_base + GP_CNT + 64 + ((wave * 8 + i) % 24) * 1024, vo[0],
                        (unsigned long long)(uintptr_t)p.A + (unsigned long long)(unsigned)(soff + i * 8 * (int)lda2));
#elif defined(GP_DBG_A_ROWS)         // tool build: whole 128-byte rows per lane group (8 lines per instruction; wrong fragments)
                    GP_LOAD_A_(A_[ST2][i & 1][i >> 1], vo[0], ars, soff + i * 8 * (int)lda2, "0");
#elif defined(GP_DBG_A_SAMELINE)     // tool build: every lane of the wave reads the same row
                    GP_LOAD_A_(A_[ST2][i & 1][i >> 1], vo[0], ars, soff, "0");
#else
                    GP_LOAD_A(A_[ST2][i & 1][i >> 1], vo[i & 1], ars, soff, i >> 1);
#endif
                }
                if constexpr (g == 59) seen_l = *(volatile gp_lds_int_t*)cnt_landed;
                if constexpr (g == 71) {
                    // all four shares of tile t+1 have landed (it is first read in the next gap)
                    GP_ST_BEGIN();
#ifndef GP_DBG_NOSYNC
                    GP_SPIN(t + 1 < nk && __builtin_amdgcn_readfirstlane(seen_l) < 4 * (t + 2), seen_l = *(volatile gp_lds_int_t*)cnt_landed);
#endif
                    GP_ST_END(3);
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 75) {
                    if (lane == 0) __hip_atomic_fetch_add(cnt_freed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            constexpr int s = g / 20, i2 = g % 20, nb = i2 >> 1, mb = i2 & 1, f = s * 10 + nb;
            if constexpr (nb < 8) GP_MFMA_A(acc_[mb][nb], Bf_[f % 5], A_[ST][mb][s]);
            else GP_MFMA_V(acc_[mb][nb], Bf_[f % 5], A_[ST][mb][s]);
        });
    };
    for (int t = 0; t < nk; t += 3) {
        tile(gp_ic<0>{}, t);
        if (t + 1 >= nk) break;
        tile(gp_ic<1>{}, t + 1);
        if (t + 2 >= nk) break;
        tile(gp_ic<2>{}, t + 2);
    }
    wait_vmcnt<0>();
    gp_settle(acc);
#ifdef GP_STAMPS
    if (lane == 0 && p.workspace && !sp.partial && (size_t)(blockIdx.x * 4 + wave + 1) * 64 <= (size_t)p.workspace_bytes) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(p.workspace) + (size_t)(blockIdx.x * 4 + wave) * 8;
        out[0] = st_acc[0]; out[1] = st_acc[1]; out[2] = st_acc[2]; out[3] = st_acc[3];
        out[4] = __builtin_readcyclecounter() - st_loop0; out[5] = (unsigned long long)nk;
        out[6] = __builtin_amdgcn_s_memrealtime() - st_real0;
    }
#endif

    if (sp.partial) {
        // split-K: raw fp32 accumulators, [split][tile][256][320]; splitk_reduce_kernel applies the epilogue
        const size_t slot = (size_t)blockIdx.y * sp.tile_count + (size_t)(swz - sp.tile_begin);
        float* const dst = sp.partial + (slot * GBM + wave * 64 + fr) * 320 + 4 * fh;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 10; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = make_float4(acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]);
                    *reinterpret_cast<float4*>(dst + (size_t)mb * 32 * 320 + nb * 32 + 8 * q) = v;
                }
        return;
    }
    persist_epilogue<640, false, EPI>(acc, p, m0, n0, p.N, wave, 0, lane, smem + GP_RING + wave * 2048);
}

template <int MODE, int EPI>
int launch_pipe320(const DcGemmParams& p, hipStream_t stream, const GemmSplit& sp, int grid_x, int grid_y) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_pipe320_kernel<MODE, EPI>), GP_LDS)) return e;
    hipLaunchKernelGGL((gemm_pipe320_kernel<MODE, EPI>), dim3(grid_x, grid_y), dim3(256), GP_LDS, stream, p, sp);
    DC_CHECK_LAUNCH();
    return 0;
}
