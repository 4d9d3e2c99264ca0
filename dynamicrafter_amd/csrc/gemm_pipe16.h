// The one-wave-per-SIMD 256 x 320 x 64 GEMM / implicit-GEMM conv kernel (design: gemm_pipe.h) on v_mfma_f32_16x16x32_bf16 (+ MODE 3: 3x3 conv behind a
// fused nearest x2 upsampling). It began on v_mfma_f32_32x32x16_bf16 (tools/experimental/gemm_pipe32.h): the same tiles, ring, counters and byte streams, the other
// bf16 MFMA shape. Why: under an MFMA stream the chip is power-managed and the clock it holds depends on the
// shape - on random operands the 16x16x32 loop delivers more FLOP/s than the 32x32x16 loop at equal cycles per FLOP
// (MI355X_MICROARCH.md 'DVFS give-back' item 7; tools/ubench/mfma_power). Same output tile per wave (64 rows x 320 columns):
//   per K tile of 64: 160 MFMAs (4 row blocks x 20 column blocks x 2 k steps of 32) instead of 80, still 40 weight-fragment
//   reads (a fragment = 16 weight rows x 32 k = the same 1 KB, used by 4 MFMAs) and 8 activation loads (16 rows x 64
//   contiguous bytes per row: better coalesced than the 32 rows x 32 bytes of the other shape);
//   accumulators: 80 blocks of 4 registers, 64 of them in the accumulator file;
//   the LDS image of the weight tile, its swizzle and the LDS-DMA stream are unchanged (a ds_read_b128 group - lanes
//   {0-3, 12-15, 20-27} - now holds rows {0-3, 12-15} at chunk 4 s and rows {4-11} at chunk 4 s + 1: 16 distinct slots).
// MFMA gaps are 16 clocks, 8 of them free for other issue: weight-fragment reads sit in gaps = 0 mod 4, vector-memory and
// counter operations in gaps = 2 mod 4.
#pragma once

typedef __attribute__((ext_vector_type(4))) float gp_f32x4_t;

#define GP16_MFMA_A(d, w, x) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(w), "v"(x))
#define GP16_MFMA_V(d, w, x) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(w), "v"(x))

// padding statements between compiler-generated accesses of the accumulators and the asm MFMAs, 20 blocks per statement
__device__ __forceinline__ void gp16_settle(gp_f32x4_t (&acc)[4][20]) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        asm volatile("s_nop 7\n\ts_nop 7"
                     : "+a"(acc[rb][0]), "+a"(acc[rb][1]), "+a"(acc[rb][2]), "+a"(acc[rb][3]), "+a"(acc[rb][4]), "+a"(acc[rb][5]),
                       "+a"(acc[rb][6]), "+a"(acc[rb][7]), "+a"(acc[rb][8]), "+a"(acc[rb][9]), "+a"(acc[rb][10]), "+a"(acc[rb][11]),
                       "+a"(acc[rb][12]), "+a"(acc[rb][13]), "+a"(acc[rb][14]), "+a"(acc[rb][15]), "+v"(acc[rb][16]), "+v"(acc[rb][17]),
                       "+v"(acc[rb][18]), "+v"(acc[rb][19]));
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
}

template <int MODE, int EPI>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_pipe320x16_kernel(const DcGemmParams p, const GemmSplit sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    const int tiles_n = p.N / 320;
    // which tile, which K range, which epilogue: the launch's (classic), or per workgroup under the merged plan (GemmSplit::whole)
    int bx = blockIdx.x, by = blockIdx.y, tile_begin = sp.tile_begin, tile_count = sp.tile_count, nsplit = sp.splits;
    float* partial = sp.partial;
    if (sp.whole > 0) {
        if (bx < sp.whole) { tile_begin = 0; tile_count = sp.whole; nsplit = 1; partial = nullptr; by = 0; }
        else { bx -= sp.whole; by = bx / tile_count; bx -= by * tile_count; }
    }
    const int swz = tile_begin + xcd_remap(bx, tile_count);
    const int tile_n = swz % tiles_n;
    const int tile_m = swz / tiles_n;
    const int m0 = tile_m * GBM;
    const int n0 = tile_n * 320;

    const int nk_all = p.K / GBK;
    const int kt_lo = (int)(((long long)by * nk_all) / nsplit);
    const int kt_hi = (int)(((long long)(by + 1) * nk_all) / nsplit);
    const int nk = kt_hi - kt_lo;

    gp_lds_int_t* const cnt_landed = (gp_lds_int_t*)(smem + GP_CNT);
    gp_lds_int_t* const cnt_freed = cnt_landed + 1;
    if (tid < 2) cnt_landed[tid] = 0;
    __syncthreads();
    int gave_up = 0;                                        // see GP_SPIN

    // ---- activation rows: lane (lr, lq) holds bytes [64 s + 16 lq, +16) of the K tile's slice of rows 16 rb + lr of its wave
    const unsigned lda2 = (unsigned)p.lda * 2u;
    unsigned rowoff[4];
    int mask[4];
    long long bias = 0;                                     // the descriptor's base lies `bias` bytes in front of p.A
    if (MODE == 1) bias = (long long)(p.pad * p.IW + p.pad) * lda2;
    if (MODE == 2) bias = (long long)p.HW * lda2;
    if (MODE == 3) bias = (long long)(p.IW + 1) * lda2;
#ifdef GP_DBG_QUAD_ROWS      // tool build (results wrong): the four lanes of a quad read 64 contiguous bytes of ONE row (addresses and
                             // tap masks of that row) - what the activation loads would cost the texture-address path if the fragment
                             // layout allowed coalescing
    const int lr_a = lr & ~3, lq_a = lane & 3;
#else
    const int lr_a = lr, lq_a = lq;
#endif
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int m = m0 + wave * 64 + rb * 16 + lr_a;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        if (MODE == 0) {
            rowoff[rb] = (unsigned)mm * lda2 + lq_a * 16;
            mask[rb] = ok ? 1 : 0;
        } else if (MODE == 1) {
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
            mask[rb] = mk;
            rowoff[rb] = (unsigned)(((n * p.IH + iy0 + p.pad) * p.IW + ix0 + p.pad)) * lda2 + lq_a * 16;
        } else if (MODE == 3) {
            // nearest x2 upsampling fused into the conv (stride 1, pad 1): tap (dy, dx) of output pixel (oy, ox) reads input pixel
            // ((oy + dy - 1) >> 1, (ox + dx - 1) >> 1) = centre (oy >> 1, ox >> 1) + (ry, rx), ry = {py - 1, 0, py}[dy] with py = oy & 1
            // (rx likewise): the offset is the centre's (+ bias) and two per-lane selects by the parities (mask bits 9, 10)
            const int ohw = p.OH * p.OW;
            const int n = mm / ohw;
            const int rem = mm - n * ohw;
            const int oy = rem / p.OW;
            const int ox = rem - oy * p.OW;
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int uy = oy + t / 3 - 1, ux = ox + t % 3 - 1;
                if (ok && uy >= 0 && uy < p.OH && ux >= 0 && ux < p.OW) mk |= 1 << t;
            }
            mask[rb] = mk | ((oy & 1) << 9) | ((ox & 1) << 10);
            rowoff[rb] = (unsigned)(((n * p.IH + (oy >> 1) + 1) * p.IW + (ox >> 1) + 1)) * lda2 + lq_a * 16;
        } else {
            const int frame = (mm / p.HW) % p.T;
            int mk = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tt = frame + t - 1;
                if (ok && tt >= 0 && tt < p.T) mk |= 1 << t;
            }
            mask[rb] = mk;
            rowoff[rb] = (unsigned)mm * lda2 + lq_a * 16;
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
    auto a_tile = [&](int kt, unsigned (&vo)[4], int& soff) __attribute__((always_inline)) {
        int tap = 0;
        if (MODE == 0) {
            soff = kt * 128;
        } else if (MODE == 1) {
            const int cs = kt / 9;
            tap = kt - cs * 9;
            const int dy = tap / 3, dx = tap - dy * 3;
            soff = (dy * p.IW + dx) * (int)lda2 + cs * 128;
        } else if (MODE == 3) {
            const int cs = kt / 9;
            tap = kt - cs * 9;
            const int dy = tap / 3, dx = tap - dy * 3;
            soff = cs * 128;
            const int y0 = dy == 0 ? -p.IW * (int)lda2 : 0, y1 = dy == 2 ? p.IW * (int)lda2 : 0;      // by the row parity
            const int x0 = dx == 0 ? -(int)lda2 : 0, x1 = dx == 2 ? (int)lda2 : 0;                    // by the column parity
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const unsigned o = rowoff[rb] + (unsigned)(((mask[rb] >> 9) & 1) ? y1 : y0) + (unsigned)(((mask[rb] >> 10) & 1) ? x1 : x0);
                vo[rb] = ((mask[rb] >> tap) & 1) ? o : 0x80000000u;
            }
            return;
        } else {
            const int cs = kt / 3;
            tap = kt - cs * 3;
            soff = tap * p.HW * (int)lda2 + cs * 128;
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) vo[rb] = ((mask[rb] >> tap) & 1) ? rowoff[rb] : 0x80000000u;
    };

    // ---- weight tile by LDS-DMA: exactly as in gemm_pipe.h
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
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_dst), "v"(voff), "s"(sbase) : "memory");
    };

    // ---- fragment reads: weight fragment (s, cb) = rows 16 cb + lr, 16-byte chunk 4 s + lq
    unsigned bofs[3][2];
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bofs[st][s] = lds_base + st * GP_STAGE + lr * 128 + (((4 * s + lq) ^ ((lr >> 1) & 7)) << 4);
            asm volatile("" : "+v"(bofs[st][s]));
        }
    bf16x8_t Bf[5];
    auto rd_b = [&](int slot, unsigned addr, int cb) __attribute__((always_inline)) {
        Bf[slot] = *(gp_lds_frag_t*)((const __attribute__((address_space(3))) char*)(uintptr_t)addr + cb * 2048);
    };

    u32x4_t A[3][4][2];
    gp_f32x4_t acc[4][20];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 20; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // ---- prologue: operands of tiles 0 and 1 in the loop's order (10 pieces, 8 loads per tile)
    auto clampk = [&](int t) { return kt_lo + (t < nk ? t : nk - 1); };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned long long ws = w_src(clampk(t));
#pragma unroll
        for (int j = 0; j < 10; ++j) dma(lds_base + t * GP_STAGE + (j * 4 + wave) * 1024, voffB[j], ws);
        unsigned vo[4];
        int soff;
        a_tile(clampk(t), vo, soff);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if ((i >> 2) == 0) GP_LOAD_A_(A[t][i & 3][0], vo[i & 3], ars, soff, "0");
            else GP_LOAD_A_(A[t][i & 3][1], vo[i & 3], ars, soff, "64");
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
    gp16_settle(acc);

    int seen_f = 0, seen_l = 0;
    auto tile = [&](auto ST_, int t) __attribute__((always_inline)) {
        constexpr int ST = decltype(ST_)::value, ST1 = (ST + 1) % 3, ST2 = (ST + 2) % 3;
        const int kt2 = clampk(t + 2);
        unsigned vo[4] = {0u, 0u, 0u, 0u};
        int soff = 0;
        unsigned long long ws = 0;
        gp_f32x4_t (&acc1)[4][20] = acc;
        bf16x8_t (&Bf1)[5] = Bf;
        u32x4_t (&A1)[3][4][2] = A;
        gp_for(std::make_integer_sequence<int, 160>{}, [&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value;
            gp_f32x4_t (&acc_)[4][20] = acc1;
            bf16x8_t (&Bf_)[5] = Bf1;
            u32x4_t (&A_)[3][4][2] = A1;
            if constexpr (g == 0) {
                wait_vmcnt<18>();                           // the activations of tile t (requested two tiles ago)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(A_[ST][i & 3][i >> 2]));
            }
            if constexpr ((g & 3) == 0) {
                // fragment 4 ahead of the one the next four MFMAs use; the last four reads are the next tile's first fragments
                constexpr int F = g / 4 + 4;
                if constexpr (F < 40) rd_b(F % 5, bofs[ST][F / 20], F % 20);
                else rd_b(F % 5, bofs[ST1][0], F - 40);
            } else if constexpr ((g & 3) == 2) {
                if constexpr (g == 2) seen_f = *(volatile gp_lds_int_t*)cnt_freed;
                if constexpr (g == 6) {
                    // stage ST2 held tile t-1: every wave is past its last fragment of it
                    GP_SPIN(__builtin_amdgcn_readfirstlane(seen_f) < 4 * t, seen_f = *(volatile gp_lds_int_t*)cnt_freed);
                    asm volatile("" ::: "memory");
                    ws = w_src(kt2);
                }
                if constexpr (g >= 10 && g <= 82 && ((g - 10) & 7) == 0) {
                    constexpr int j = (g - 10) >> 3;
                    dma(lds_base + ST2 * GP_STAGE + (j * 4 + wave) * 1024, voffB[j], ws);
                }
                if constexpr (g == 86) {
                    wait_vmcnt<18>();                       // this wave's pieces of tile t+1 (issued a tile ago) are in LDS
#ifdef GP_DBG_SKIP_POST     // tool build (tests/test_error_word_gpu.py): wave 3 forgets one arrival - every wait on `landed` then times out
                    if (lane == 0 && !(wave == 3 && t == 1))
#else
                    if (lane == 0)
#endif
                        __hip_atomic_fetch_add(cnt_landed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    a_tile(kt2, vo, soff);
                }
                if constexpr (g >= 90 && g <= 146 && ((g - 90) & 7) == 0) {
                    constexpr int i = (g - 90) >> 3;
                    if constexpr ((i >> 2) == 0) GP_LOAD_A_(A_[ST2][i & 3][0], vo[i & 3], ars, soff, "0");
                    else GP_LOAD_A_(A_[ST2][i & 3][1], vo[i & 3], ars, soff, "64");
                }
                if constexpr (g == 118) seen_l = *(volatile gp_lds_int_t*)cnt_landed;
                if constexpr (g == 142) {
                    // all four shares of tile t+1 have landed (it is first read two gaps on)
                    GP_SPIN(t + 1 < nk && __builtin_amdgcn_readfirstlane(seen_l) < 4 * (t + 2), seen_l = *(volatile gp_lds_int_t*)cnt_landed);
                    asm volatile("" ::: "memory");
                }
                if constexpr (g == 150) {
                    if (lane == 0) __hip_atomic_fetch_add(cnt_freed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            constexpr int f = g / 4, rb = g & 3, s = f / 20, cb = f % 20;
            if constexpr (cb < 16) GP16_MFMA_A(acc_[rb][cb], Bf_[f % 5], A_[ST][rb][s]);
            else GP16_MFMA_V(acc_[rb][cb], Bf_[f % 5], A_[ST][rb][s]);
        });
    };
#ifdef GP_STAMPS        // tool build (tools/pipe_stamps.py): clocks of the tile loop and the shader clock the chip holds under it
    const unsigned long long st_loop0 = __builtin_readcyclecounter();
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int t = 0; t < nk; t += 3) {
        tile(gp_ic<0>{}, t);
        if (t + 1 >= nk) break;
        tile(gp_ic<1>{}, t + 1);
        if (t + 2 >= nk) break;
        tile(gp_ic<2>{}, t + 2);
    }
    wait_vmcnt<0>();
    gp16_settle(acc);
    if (gave_up && lane == 0) atomicOr(sp.err, DC_ERRW_GEMM_PIPE);      // a counter wait timed out: this tile is not valid
#ifdef GP_STAMPS
    if (lane == 0 && p.workspace && !sp.partial && (size_t)(blockIdx.x * 4 + wave + 1) * 64 <= (size_t)p.workspace_bytes) {
        unsigned long long* out = reinterpret_cast<unsigned long long*>(p.workspace) + (size_t)(blockIdx.x * 4 + wave) * 8;
        out[0] = 0; out[1] = 0; out[2] = 0; out[3] = 0;
        out[4] = __builtin_readcyclecounter() - st_loop0; out[5] = (unsigned long long)nk;
        out[6] = __builtin_amdgcn_s_memrealtime() - st_real0;
    }
#endif

    // ---- epilogue. A lane holds, of output row 16 rb + lr, the channels 16 cb + 4 lq .. + 3 of every column block.
    if (partial) {
        // split-K: raw fp32 accumulators, [split][tile][256][320]; splitk_reduce_kernel applies the epilogue
        const size_t slot = (size_t)by * tile_count + (size_t)(swz - tile_begin);
        float* const dst = partial + (slot * GBM + wave * 64 + lr) * 320 + 4 * lq;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int cb = 0; cb < 20; ++cb)
                *reinterpret_cast<float4*>(dst + (size_t)rb * 16 * 320 + cb * 16) = make_float4(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
        return;
    }
    // bias / GELU / per-row-group vector / alpha in the accumulator layout, then two column blocks at a time (16 rows x 32
    // channels) through the wave's 2 KB LDS patch into row-major order: a lane ends up with 8 consecutive channels of a row, one
    // store instruction moves 16 rows x 64 contiguous bytes; the residual is fetched in the same pattern
    char* const ebuf = smem + GP_RING + wave * 2048;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                        // (or the addresses below are hoisted above the K loop and spilled)
    const int er = lane_e & 15, eq = lane_e >> 4;           // accumulator coordinates
    const int rrow = lane_e >> 2, rc = lane_e & 3;          // read-back coordinates: row of the 16, 16-byte piece of the 64 bytes
    const char* const rbase = reinterpret_cast<const char*>(p.residual);
    char* const cbase = reinterpret_cast<char*>(p.C);
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int m_acc = m0 + wave * 64 + rb * 16 + er;
        const float* rv = p.rowvec ? p.rowvec + (size_t)((m_acc < p.M ? m_acc : 0) / p.rows_per_vec) * p.rowvec_ld : nullptr;
        int mr = m0 + wave * 64 + rb * 16 + rrow;
        const bool rok = mr < p.M;
        if (!rok) mr = p.M - 1;                             // clamped rows are loaded, never stored
        const unsigned co = ((unsigned)mr * (unsigned)p.ldc + (unsigned)(n0 + rc * 8)) * 2u;
        const unsigned ro = ((unsigned)mr * (unsigned)p.ldr + (unsigned)(n0 + rc * 8)) * 2u;
#pragma unroll
        for (int cp = 0; cp < 10; ++cp) {
            u32x4_t rr;
            if constexpr (EPI == 1) rr = *reinterpret_cast<const u32x4_t*>(rbase + ro + cp * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cb = 2 * cp + h;
                const int n = n0 + cb * 16 + 4 * eq;
                float4 v = make_float4(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
                if (p.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(p.bias + n);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (p.flags & DC_GEMM_GELU) { v.x = DC_GELU(v.x); v.y = DC_GELU(v.y); v.z = DC_GELU(v.z); v.w = DC_GELU(v.w); }
                if (rv) {
                    const float4 r4 = *reinterpret_cast<const float4*>(rv + n);
                    v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                }
                if (p.alpha != 1.0f) { v.x *= p.alpha; v.y *= p.alpha; v.z *= p.alpha; v.w *= p.alpha; }
                uint2 pk;
                pk.x = pack_bf2(v.x, v.y);
                pk.y = pack_bf2(v.z, v.w);
                // patch: row er (64 bytes), 8-byte slot 4 h + eq, XOR-swizzled by the row pair so that the 16-byte read-back below
                // and these 8-byte stores both spread over the banks
                *reinterpret_cast<uint2*>(ebuf + er * 64 + ((((4 * h + eq) ^ (((er >> 1) & 3) << 1))) << 3)) = pk;
            }
            u32x4_t d = *reinterpret_cast<const u32x4_t*>(ebuf + rrow * 64 + ((rc ^ ((rrow >> 1) & 3)) << 4));
            if constexpr (EPI == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    d[e] = pack_bf2(__uint_as_float(d[e] << 16) + __uint_as_float(rr[e] << 16),
                                    __uint_as_float(d[e] & 0xffff0000u) + __uint_as_float(rr[e] & 0xffff0000u));
            }
            if (rok) *reinterpret_cast<u32x4_t*>(cbase + co + cp * 64) = d;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int MODE, int EPI>
int launch_pipe320x16(const DcGemmParams& p, hipStream_t stream, const GemmSplit& sp, int grid_x, int grid_y) {
    static DcLdsOnce lds_once;
    if (const int e = lds_once.ensure(reinterpret_cast<const void*>(&gemm_pipe320x16_kernel<MODE, EPI>), GP_LDS)) return e;
    hipLaunchKernelGGL((gemm_pipe320x16_kernel<MODE, EPI>), dim3(grid_x, grid_y), dim3(256), GP_LDS, stream, p, sp);
    DC_CHECK_LAUNCH();
    return 0;
}
