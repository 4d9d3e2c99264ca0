// "Ping-pong" GEMM for the short-K launches whose epilogue is a large share of a tile: the GEGLU projections of levels 1-3
// ([73728 x 640 -> 2 x 2560], [18432 x 1280 -> 2 x 5120], [4608 x 1280 -> 2 x 5120])
// (included by gemm_conv_glds.hip inside its namespace, behind gemm_pipe.h / gemm_pipe16.h whose ring helpers it uses).
//
// Why another kernel. The persistent 8-wave kernel computes these at 0.27-0.35 of the matrix peak: PMC says MFMA busy 37-45 %,
// a third of the wave time parked - all eight waves of a CU enter the erf-GELU / convert / store epilogue together, and no MFMA
// issues meanwhile. The one-wave-per-SIMD kernel (gemm_pipe16.h) does not help: behind its K loop the epilogue is
// dead time for the matrix pipe as well. What helps is somebody else's K loop running on the SIMD during my epilogue:
//   * a workgroup is gemm_pipe16's 4 waves, but a wave owns 64 rows x 128 weight rows = 32 accumulator blocks of 16 x 16 =
//     128 accumulator registers, so the kernel fits 256 registers and TWO workgroups share a CU (72 KB of LDS each): on every
//     SIMD one wave of each. The two workgroups are independent launches of work - different tiles, no shared state, no
//     barrier, no counter between them - so they drift apart on their own: while one converts and stores, the other has
//     the matrix pipe to itself, and while both are in their K loops they share it (MI355X_MICROARCH 'Two waves per SIMD').
//   * inside a workgroup: v_mfma_f32_16x16x32_bf16 (the shape the chip clocks highest on), LDS counters `landed` / `freed`
//     instead of s_barrier, every instruction of the tile loop an asm volatile statement or a volatile LDS load in source order -
//     as in gemm_pipe16.h. NOT as there: the activations go through LDS too. The first version loaded them straight into
//     registers in fragment layout (a lane = one row x 16 bytes) and was bound by exactly that: a buffer load whose 64 lanes
//     touch 16 rows makes every quad of lanes 4 different cache lines, 64 address cycles per instruction instead of 16, and at
//     64 x 128 accumulators per wave there are 8 such loads per 64 MFMAs (gemm_pipe16: per 160) - tool builds with the operands
//     aliased to one L2-hot panel ran no faster (572 -> 540 us), with the activation loads made free 572 -> 391 us
//     (profiles/r04_pp_variants.txt). LDS-DMA pieces are lane-linear: a piece = 16 rows x 64 contiguous bytes, quads whole.
//   * K tiles of 32 (64-byte LDS rows, chunk slot XOR (row >> 1) & 3: conflict-free for this fragment layout), rings of 3 stages:
//     activations 16 KB per stage - each wave stages and reads only ITS 64 rows (4 KB), so that half of the ring needs no
//     counter, only the wave's own vmcnt - weights 8 KB per stage, shared: 72 KB. The epilogue patch of a wave is its own
//     activation region.
//   * GEGLU: the weight tile is 64 value rows (n0 ..) + 64 gate rows (N/2 + n0 ..); a lane holds value and gate of the same
//     (row, channel) in accumulator blocks cb and cb + 4, the product x * gelu(gate) is formed in registers and 16 rows x 64
//     channels leave through the wave's 2 KB LDS patch as FULL 128-byte lines (8 rows per store instruction).
// Per K tile of 32 a wave issues 32 MFMAs (8 column blocks x 4 row blocks), 8 weight- and 4 activation-fragment reads and 6
// LDS-DMA pieces (4 of its own activation rows, 2 of the weight tile) for tile t + 2.
// vmcnt by hand: per tile the wave issues exactly those 6 pieces, so in the middle of tile t "everything but the 6 youngest" =
// the pieces of tile t + 1. Behind the end of the K range the same pieces are issued in a form that moves one cache line.
#pragma once

constexpr int PP_ROWS = 128;                       // weight rows per tile (GEGLU: 64 value + 64 gate -> 64 output columns)
constexpr int PP_K = 32;                           // K tile
constexpr int PP_NST = 3;
constexpr int PP_ASTAGE = 256 * PP_K * 2;          // 16 KB: [4 waves][64 rows][64 B]
constexpr int PP_WSTAGE = PP_ROWS * PP_K * 2;      // 8 KB: [128 rows][64 B]
constexpr int PP_WBASE = PP_NST * PP_ASTAGE;       // 48 KB
constexpr int PP_CNT = PP_WBASE + PP_NST * PP_WSTAGE;      // 72 KB
constexpr int PP_LDS = PP_CNT + 64;                // two workgroups per CU

__device__ __forceinline__ void pp_settle(gp_f32x4_t (&acc)[4][8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
        asm volatile("s_nop 7\n\ts_nop 7"
                     : "+a"(acc[2 * h][0]), "+a"(acc[2 * h][1]), "+a"(acc[2 * h][2]), "+a"(acc[2 * h][3]), "+a"(acc[2 * h][4]),
                       "+a"(acc[2 * h][5]), "+a"(acc[2 * h][6]), "+a"(acc[2 * h][7]), "+a"(acc[2 * h + 1][0]), "+a"(acc[2 * h + 1][1]),
                       "+a"(acc[2 * h + 1][2]), "+a"(acc[2 * h + 1][3]), "+a"(acc[2 * h + 1][4]), "+a"(acc[2 * h + 1][5]),
                       "+a"(acc[2 * h + 1][6]), "+a"(acc[2 * h + 1][7]));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
}

// out[M, N/2] = (x Wv^T + bv) * gelu(x Wg^T + bg), Wv = W rows [0, N/2), Wg = rows [N/2, N).
// Requirements (pp_ok): DC_GEMM_GEGLU, mode 0, K % 32 == 0, output columns % 64 == 0, 16-byte
// aligned rows, byte offsets below 2^31, no rowvec / alpha / fp32 output.
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_pp_kernel(const DcGemmParams p, const GemmSplit sp, const int tile_group) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BNOUT = 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    const int n_out = p.N >> 1;
    const int tiles_n = n_out / BNOUT;
    const int tiles_m = (p.M + GBM - 1) / GBM;
    int tile_m, tile_n;
    persist_tile(xcd_remap(blockIdx.x, sp.tile_count), tiles_m, tiles_n, tile_group, tile_m, tile_n);
    const int m0 = tile_m * GBM;
    const int n0 = tile_n * BNOUT;
    const int nk = p.K / PP_K;

    gp_lds_int_t* const cnt_landed = (gp_lds_int_t*)(smem + PP_CNT);
    gp_lds_int_t* const cnt_freed = cnt_landed + 1;
    if (tid < 2) cnt_landed[tid] = 0;
    __syncthreads();
    int gave_up = 0;                                        // see GP_SPIN (gemm_pipe.h)

    // ---- LDS-DMA pieces (1 KB, lane-linear in LDS = 16 rows x 64 B): lane l carries row l >> 2 of the piece, chunk slot l & 3,
    // which holds source chunk (l & 3) ^ ((row >> 1) & 3) = (l & 3) ^ ((l >> 3) & 3) (a piece starts at a multiple of 16 rows)
    const unsigned lds_base = (unsigned)(uintptr_t)((const __attribute__((address_space(3))) char*)smem);
    const unsigned src_chunk = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) << 4);
    unsigned voffA[4], voffW[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                           // activations: piece j = rows 16 j .. + 15 of this wave's 64
        int m = m0 + wave * 64 + j * 16 + (lane >> 2);
        if (m >= p.M) m = p.M - 1;                          // clamped rows are computed, never stored
#ifdef PP_DBG_ALIAS_A       // tool build (tools/pp_variants.sh; results wrong): every workgroup reads row panel 0
        m = wave * 64 + j * 16 + (lane >> 2);
#endif
        voffA[j] = (unsigned)m * (unsigned)p.lda * 2u + src_chunk;
        asm volatile("" : "+v"(voffA[j]));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                           // weights: piece 2 wave + j = tile rows 16 (2 wave + j) .. + 15
        const int row = (2 * wave + j) * 16 + (lane >> 2);
        int wrow = row < 64 ? n0 + row : (p.N >> 1) + n0 + (row - 64);
#ifdef PP_DBG_ALIAS_W       // tool build: every workgroup reads weight tile 0
        wrow = row;
#endif
        voffW[j] = (unsigned)wrow * (unsigned)p.K * 2u + src_chunk;
        asm volatile("" : "+v"(voffW[j]));
    }
    const unsigned long long a_base = (unsigned long long)(uintptr_t)p.A, w_base = (unsigned long long)(uintptr_t)p.W;
    auto dma = [&](unsigned lds_dst, unsigned voff, unsigned long long sbase) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    };
    const unsigned a_dst = lds_base + wave * 4096;          // + stage * PP_ASTAGE + j * 1024
    const unsigned w_dst = lds_base + PP_WBASE + wave * 2048;      // + stage * PP_WSTAGE + j * 1024

    // ---- fragment reads: row 16 b + lr of a 64-byte-row image, chunk lq
    unsigned aofs = lds_base + wave * 4096 + lr * 64 + ((lq ^ ((lr >> 1) & 3)) << 4);
    unsigned wofs = lds_base + PP_WBASE + lr * 64 + ((lq ^ ((lr >> 1) & 3)) << 4);
    asm volatile("" : "+v"(aofs), "+v"(wofs));
    bf16x8_t Wf[4];        // ring of 4, read 3 fragments (12 MFMAs) ahead
    bf16x8_t Af[2][4];     // the 4 row blocks of a tile; the other set receives the next tile's
    auto rd = [&](bf16x8_t& dst, unsigned addr, int imm) __attribute__((always_inline)) {
        dst = *(gp_lds_frag_t*)((const __attribute__((address_space(3))) char*)(uintptr_t)addr + imm);
    };

    gp_f32x4_t acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: tiles 0 and 1 (6 + 6 pieces; nk >= 2 by pp_ok)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned long long as = a_base + (unsigned long long)t * 64ull, ws = w_base + (unsigned long long)t * 64ull;
#pragma unroll
        for (int j = 0; j < 4; ++j) dma(a_dst + t * PP_ASTAGE + j * 1024, voffA[j], as);
#pragma unroll
        for (int j = 0; j < 2; ++j) dma(w_dst + t * PP_WSTAGE + j * 1024, voffW[j], ws);
    }
    wait_vmcnt<6>();                                        // this wave's pieces of tile 0
    if (lane == 0) __hip_atomic_fetch_add(cnt_landed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    {
        int seen = *(volatile gp_lds_int_t*)cnt_landed;
        GP_SPIN(__builtin_amdgcn_readfirstlane(seen) < 4, seen = *(volatile gp_lds_int_t*)cnt_landed);
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) rd(Af[0][rb], aofs, rb * 1024);
#pragma unroll
    for (int f = 0; f < 3; ++f) rd(Wf[f], wofs, f * 1024);
    pp_settle(acc);

    int seen_f = 0, seen_l = 0;
    // Tile t: stage ST = t % 3, activation fragment set AB = t & 1. The body issues the same 6 pieces in every tile, so that the
    // hand-counted vmcnt window never changes and no branch sits between MFMAs; pieces for tiles behind the end of the K range
    // get lane offset 0 (64 lanes read one 16-byte chunk: one cache line) into a stage nobody will read.
    auto tile = [&](auto ST_, auto AB_, int t) __attribute__((always_inline)) {
        constexpr int ST = decltype(ST_)::value, ST1 = (ST + 1) % 3, ST2 = (ST + 2) % 3, AB = decltype(AB_)::value;
        const bool more1 = t + 1 < nk, more2 = t + 2 < nk;          // wave-uniform
        const unsigned long long koff = (unsigned long long)(more2 ? t + 2 : 0) * 64ull;
        unsigned va[4], vw[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = more2 ? voffA[i] : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) vw[i] = more2 ? voffW[i] : 0u;
#ifdef PP_DBG_DUMMY_A       // tool builds (tools/pp_variants.sh; results wrong): the activation / weight pieces move one cache line each
#pragma unroll
        for (int i = 0; i < 4; ++i) va[i] = 0u;
#endif
#ifdef PP_DBG_DUMMY_W
#pragma unroll
        for (int i = 0; i < 2; ++i) vw[i] = 0u;
#endif
        gp_f32x4_t (&acc1)[4][8] = acc;
        bf16x8_t (&Wf1)[4] = Wf;
        bf16x8_t (&Af1)[2][4] = Af;
        unsigned (&va1)[4] = va;
        unsigned (&vw1)[2] = vw;
        gp_for(std::make_integer_sequence<int, 32>{}, [&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            gp_f32x4_t (&acc_)[4][8] = acc1;
            bf16x8_t (&Wf_)[4] = Wf1;
            bf16x8_t (&Af_)[2][4] = Af1;
            unsigned (&va_)[4] = va1;
            unsigned (&vw_)[2] = vw1;
            if constexpr ((g & 3) == 0) {
                // weight fragment 3 ahead of the one the next four MFMAs use (its slot held fragment F - 4, whose MFMAs are
                // issued); the last three reads are the next tile's first fragments (garbage behind the last tile, never used)
                constexpr int F = g / 4 + 3;
                if constexpr (F < 8) rd(Wf_[F % 4], wofs, ST * PP_WSTAGE + F * 1024);
                else rd(Wf_[F % 4], wofs, ST1 * PP_WSTAGE + (F - 8) * 1024);
            }
            if constexpr (g == 1) seen_f = *(volatile gp_lds_int_t*)cnt_freed;
            // this wave's activation rows of tile t + 2: its own region of stage ST2, which it finished reading in tile t - 2
            if constexpr (g == 2 || g == 3 || g == 5 || g == 6) {
                constexpr int j = g < 4 ? g - 2 : g - 3;
                dma(a_dst + ST2 * PP_ASTAGE + j * 1024, va_[j], a_base + koff);
            }
            if constexpr (g == 7) {
                // weight stage ST2 held tile t - 1: every wave is past its last fragment of it
                GP_SPIN(__builtin_amdgcn_readfirstlane(seen_f) < 4 * t, seen_f = *(volatile gp_lds_int_t*)cnt_freed);
                asm volatile("" ::: "memory");
            }
            if constexpr (g == 9 || g == 10) dma(w_dst + ST2 * PP_WSTAGE + (g - 9) * 1024, vw_[g - 9], w_base + koff);
            if constexpr (g == 11) {
                wait_vmcnt<6>();                            // everything but this tile's 6 pieces: the pieces of tile t + 1 are in LDS
                if (lane == 0) __hip_atomic_fetch_add(cnt_landed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if constexpr (g == 13 || g == 14 || g == 17 || g == 18) {
                constexpr int rb = g < 16 ? g - 13 : g - 15;
                rd(Af_[AB ^ 1][rb], aofs, ST1 * PP_ASTAGE + rb * 1024);
            }
            if constexpr (g == 15) seen_l = *(volatile gp_lds_int_t*)cnt_landed;
            if constexpr (g == 19) {
                // all four shares of weight tile t + 1 have landed (it is first read at gap 20)
                GP_SPIN(more1 && __builtin_amdgcn_readfirstlane(seen_l) < 4 * (t + 2), seen_l = *(volatile gp_lds_int_t*)cnt_landed);
                asm volatile("" ::: "memory");
            }
            if constexpr (g == 21) {
                // behind the last fragment read of weight stage ST (gap 16): one wave's LDS operations execute in order
                if (lane == 0) __hip_atomic_fetch_add(cnt_freed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            constexpr int cb = g / 4, rb = g & 3;
            GP16_MFMA_A(acc_[rb][cb], Wf_[cb % 4], Af_[AB][rb]);
        });
    };
    for (int t = 0; t < nk; t += 6) {
        tile(gp_ic<0>{}, gp_ic<0>{}, t);
        if (t + 1 >= nk) break;
        tile(gp_ic<1>{}, gp_ic<1>{}, t + 1);
        if (t + 2 >= nk) break;
        tile(gp_ic<2>{}, gp_ic<0>{}, t + 2);
        if (t + 3 >= nk) break;
        tile(gp_ic<0>{}, gp_ic<1>{}, t + 3);
        if (t + 4 >= nk) break;
        tile(gp_ic<1>{}, gp_ic<0>{}, t + 4);
        if (t + 5 >= nk) break;
        tile(gp_ic<2>{}, gp_ic<1>{}, t + 5);
    }
    wait_vmcnt<0>();
    pp_settle(acc);
    if (gave_up && lane == 0) atomicOr(sp.err, DC_ERRW_GEMM_PP);       // a counter wait timed out: this tile is not valid

    // ---- epilogue. A lane holds, of output row 16 rb + lr, the weight rows 16 cb + 4 lq .. + 3 of every column block.
    // 16 rows x 64 channels (128 B per row) at a time through the wave's 2 KB patch: 8-byte slots XOR-swizzled by the row pair
    // (writes 2-way, the 16-byte read-back conflict-free); read back row-major, one store instruction = 8 rows x 128 B.
    char* const ebuf = smem + wave * 4096;                  // the wave's own activation region of stage 0: nothing is in flight, nobody else reads it
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                        // (or the addresses below are hoisted above the K loop)
    const int er = lane_e & 15, eq = lane_e >> 4;           // accumulator coordinates
    const int rrow = lane_e >> 3, rc = lane_e & 7;          // read-back coordinates: row of 8, 16-byte piece of the 128 bytes
    char* const cbase = reinterpret_cast<char*>(p.C);
    float4 bv[4], bg[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int n = n0 + c * 16 + 4 * eq;
        bv[c] = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bg[c] = p.bias ? *reinterpret_cast<const float4*>(p.bias + (p.N >> 1) + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        unsigned co[2];
        bool rok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int mr = m0 + wave * 64 + rb * 16 + h * 8 + rrow;
            rok[h] = mr < p.M;
            if (!rok[h]) mr = p.M - 1;                      // clamped rows are computed, never stored
            co[h] = ((unsigned)mr * (unsigned)p.ldc + (unsigned)(n0 + rc * 8)) * 2u;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = make_float4(acc[rb][c][0] + bv[c].x, acc[rb][c][1] + bv[c].y, acc[rb][c][2] + bv[c].z, acc[rb][c][3] + bv[c].w);
            v.x *= DC_GELU(acc[rb][c + 4][0] + bg[c].x); v.y *= DC_GELU(acc[rb][c + 4][1] + bg[c].y);
            v.z *= DC_GELU(acc[rb][c + 4][2] + bg[c].z); v.w *= DC_GELU(acc[rb][c + 4][3] + bg[c].w);
            uint2 pk;
            pk.x = pack_bf2(v.x, v.y);
            pk.y = pack_bf2(v.z, v.w);
            *reinterpret_cast<uint2*>(ebuf + er * 128 + (((4 * c + eq) ^ (((er >> 1) & 7) << 1)) << 3)) = pk;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = h * 8 + rrow;
            const u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + row * 128 + ((rc ^ ((row >> 1) & 7)) << 4));
#ifndef PP_DBG_NO_STORE
            if (rok[h]) *reinterpret_cast<u32x4_t*>(cbase + co[h]) = d;
#else
            if (rok[h] && d[0] == 0x12345678u) *reinterpret_cast<u32x4_t*>(cbase + co[h]) = d;
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

inline bool pp_ok(const DcGemmParams& p) {
    if (!(p.flags & DC_GEMM_GEGLU) || (p.flags & (DC_GEMM_OUT_F32 | DC_GEMM_GELU)) || p.mode != 0 || p.rowvec || p.residual || p.alpha != 1.0f) return false;
    const int n_out = p.N / 2;
    if (p.K % PP_K != 0 || p.K < 2 * PP_K || n_out % 64 != 0 || p.n_pad < p.N) return false;
    if (p.lda % 8 != 0 || ((uintptr_t)p.A % 16) != 0 || (long long)p.M * p.lda * 2 >= (1ll << 31)) return false;
    if (p.ldc % 8 != 0 || ((uintptr_t)p.C % 16) != 0 || (long long)p.M * p.ldc * 2 >= (1ll << 32)) return false;
    if ((long long)p.N * p.K * 2 >= (1ll << 32)) return false;
    if (p.bias && ((uintptr_t)p.bias % 16) != 0) return false;
    return true;
}

int launch_pp(const DcGemmParams& p, hipStream_t stream) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_pp_kernel), PP_LDS)) return e;
    GemmSplit sp;
    sp.partial = nullptr; sp.splits = 1; sp.tile_begin = 0; sp.whole = 0;
    sp.tile_count = ((p.M + GBM - 1) / GBM) * ((p.N / 2) / 64);
    sp.err = dc_error_word_device();
    if (!sp.err) return DC_ERR_ARG;
    dc_note_variant("gemm_pp_kernel<geglu>");
    hipLaunchKernelGGL(gemm_pp_kernel, dim3(sp.tile_count), dim3(256), PP_LDS, stream, p, sp, 8);
    DC_CHECK_LAUNCH();
    return 0;
}
