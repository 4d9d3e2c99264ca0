// Spatial self-attention, head_dim 64, for gfx950: ONE wave per SIMD, software-pipelined by hand.
//
//   o = softmax(q k^T * scale) v        reference: CrossAttention.forward, lvdm/modules/attention.py:101-125
//
// The two-waves-per-SIMD kernel of attention.hip spends ~1750 cycles per (64 queries x 64 keys) of a wave where the matrix
// pipe needs 1024: its two waves reach the exp / pack / max stream together. Here a workgroup is 4 waves = 256 query rows,
// a wave owns 64 rows (two 32-row blocks) and the whole register file, and the vector work of one 32-key HALF TILE is
// placed by hand in the gaps between the MFMAs of its neighbours:
//
//   step h (16 MFMAs):   gaps 0..7   S(h+2) = K(h+2) (cQ)^T - m     [QK^T of the half tile two ahead]
//                        gaps 8..15  O^T  += V(h)^T P(h)^T          [PV of the current half tile]
//                        gaps 16..19 l += 1^T P(h)^T                [row sums on the matrix pipe]
//   beside them:         gaps 0..15  P(h+1) = exp2(S(h+1)), bf16 pack                   (2 v_exp_f32 + 1 pack per gap)
//                        gaps 16..19 the LDS fragment requests of step h+1              (3 per gap)
//                        gaps 8..11  one 16-byte global load (even steps) / LDS store (odd steps) of the K/V ring each
//
// Issue slots are what bounds this kernel (and the two-waves one: both measured 1750 cycles of issue per 64-key tile), so
// the vector stream is kept to what softmax cannot do without - one v_exp_f32 per score, one pack per pair:
//   * NO RUNNING MAX in the main pass. m is the row maximum of the FIRST 32 keys and stays: softmax is invariant under the
//     shift, exp2(s - m) keeps its relative precision anywhere in the fp32 / bf16 exponent range, and fp32 sums of 9216
//     terms are safe while every term is below 2^100. A row whose later scores exceed its first-half-tile maximum by more
//     than ~100 (69 nats) shows up as a row sum that is not < 2^100 (or NaN): then the whole workgroup repeats its block
//     with the tracking pass (TRACK: v_max3 chains in gaps 10..15, the rescale decision one half tile late, threshold
//     `thr`) - the classical online softmax, kept as the fallback that makes the kernel total.
//   * row sums as 4 more MFMAs per step (a ones fragment as A operand) instead of 32 v_add_f32: the matrix pipe has the
//     room (40 of ~44 gaps' worth per tile), the issue port does not.
//
// Every instruction of that stream is an `asm volatile` statement: hipcc allocates the registers and counts the LDS
// loads, the order is the source order (volatile statements are not reordered among themselves). Distances that the
// hardware does not interlock and hipcc does not pad inside asm are kept by construction: an MFMA result is first read by
// a vector instruction >= 2 gaps later; a v_exp_f32 result is consumed >= 2 instructions later; P is packed a step before
// the MFMA that reads it; the rare path ends in s_nop padding.
// S lives in architectural VGPRs (the exp / max stream reads it), O and Q in the accumulator file, -m as two 16-register
// tuples that are the C operand of the first MFMA of a score chain (no per-tile accumulator initialisation).
// K / V tiles of 64 keys: global -> registers -> LDS ring of 4 stages, one barrier per tile; K fragments by ds_read_b128
// (XOR-swizzled rows), V^T fragments by ds_read_b64_tr_b16. Measured (tools/flash_stamp_variants.sh, shader clocks per 64-key
// tile): with the staging as one block at the end of a tile and the fragment requests inside the exp gaps 2040, of which
// 435 the staging block + its drain and barrier and 245 the requests; MFMAs + vector stream alone 1350 (40 MFMAs = 1280).
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <type_traits>
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) char lds_char_t;
typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;
typedef const volatile __attribute__((address_space(3))) bf16x8_t lds_vfrag_t;

constexpr int FP_VLD = 192;                 // bytes per V row in LDS (4 consecutive rows on 4 distinct 64-byte bank quarters)
constexpr int FP_KBYTES = 64 * 128;
constexpr int FP_VBYTES = 64 * FP_VLD;
constexpr int FP_STAGE = FP_KBYTES + FP_VBYTES;     // 20 KB
constexpr int FP_NSTAGE = 4;
constexpr int FP_LDS = FP_NSTAGE * FP_STAGE;        // 80 KB
constexpr int FP_ROWS = 256;                        // query rows per workgroup

template <int V> using ic = std::integral_constant<int, V>;

// tool builds only (tools/flash_variants.sh): switch parts of the main loop off to see what bounds it; results are wrong
#ifdef FP_DBG_NOEX
constexpr bool FP_LOOP_EX = false;
#else
constexpr bool FP_LOOP_EX = true;
#endif
#ifdef FP_STAMPS             // tool build: shader clocks and 100 MHz ticks around the tile loop (tools/flash_stamps.py)
__device__ unsigned long long g_fp_stamps[4];
#endif

#define FP_EXP(dst, src) asm volatile("v_exp_f32 %0, %1" : "=v"(dst) : "v"(src))
#define FP_ADD(acc, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
#define FP_CVT(dst, lo, hi) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(dst) : "v"(lo), "v"(hi))
#define FP_MAX3(acc, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b))
// S chain (D, C in VGPRs; B = Q fragment in the accumulator file)
#define FP_MFMA_S0(d, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c))
#define FP_MFMA_SZ(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b))
#define FP_MFMA_S(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b))
// O and row-sum chains (D = C in the accumulator file)
#define FP_MFMA_O(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))

__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void flash_attn_d64_pipe_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                bf16_t* __restrict__ o, int ldq, int ldk, int ldv, int ldo, int heads, int Lq, int Lk,
                                int64_t q_bstride, int64_t kv_bstride, float c /* scale*log2(e) */, int q_tiles, float thr,
                                int force_track) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;

    const int nwg = gridDim.x;
    const int id = xcd_remap(blockIdx.x, nwg);
    const int qt = id % q_tiles;
    const int bh = id / q_tiles;
    const int head = bh % heads;
    const int b = bh / heads;

    const bf16_t* qb = q + (size_t)b * q_bstride * ldq + head * 64;
    const bf16_t* kb = k + (size_t)b * kv_bstride * ldk + head * 64;
    const bf16_t* vb = v + (size_t)b * kv_bstride * ldv + head * 64;
    bf16_t* ob = o + (size_t)b * q_bstride * ldo + head * 64;

    // ---- Q fragments (B operand of S^T = K (cQ)^T): lane (r, h) holds cQ[row][16kk + 8h .. +7] of its two query blocks
    int qrow[2];
    bf16x8_t Q[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        qrow[x] = qt * FP_ROWS + (wave * 2 + x) * 32 + fr;
        const int qc = qrow[x] < Lq ? qrow[x] : Lq - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // Q is pre-multiplied by scale*log2(e) (one more bf16 rounding of Q, the size of the one it already has): the
            // scores leave the MFMA in exp2 units and, with the accumulator started at -m, as s - m
            const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qb + (size_t)qc * ldq + kk * 16 + fh * 8);
            u32x4_t sc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[e] = pack_bf2(__uint_as_float(raw[e] << 16) * c, __uint_as_float(raw[e] & 0xffff0000u) * c);
            Q[x][kk] = __builtin_bit_cast(bf16x8_t, sc);
            asm volatile("" : "+a"(Q[x][kk]));        // lives in the accumulator file from here on (B operand only)
        }
    }

    // ---- staging: 64 rows x 8 chunks of 16 B per tensor and tile, 2 rows per thread; uniform tile base + per-lane offsets
    const int chunk = tid & 7, srow = tid >> 3;
    const int nt = Lk >> 6;
    u32x4_t kreg[2], vreg[2];
    unsigned kgo[2], vgo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        kgo[i] = (unsigned)((srow + 32 * i) * ldk + chunk * 8) * 2u;
        vgo[i] = (unsigned)((srow + 32 * i) * ldv + chunk * 8) * 2u;
    }
    // piece j of a tile: 0 = K rows 0..31, 1 = V rows 0..31, 2 = K rows 32..63, 3 = V rows 32..63 (one 16-byte load / store each).
    // Buffer loads: descriptor of the (batch, head) slice in SGPRs, the lane's offset in a VGPR that never changes, the tile's
    // offset in an SGPR - one instruction per piece, no per-lane address arithmetic
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
    auto load_piece = [&](int t, int j) __attribute__((always_inline)) {
        t = t < nt ? t : nt - 1;                      // past the end: the last tile again (finite data, never used)
#ifndef FP_DBG_NOLOAD
        if (j & 1) vreg[j >> 1] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vgo[j >> 1], t * 128 * ldv, 0);
        else kreg[j >> 1] = __builtin_amdgcn_raw_buffer_load_b128(krs, kgo[j >> 1], t * 128 * ldk, 0);
#endif
    };
    auto store_piece = [&](int stage, int j) __attribute__((always_inline)) {
        char* sk = smem + stage * FP_STAGE;
        const int r = srow + 32 * (j >> 1);
#ifndef FP_DBG_NOSTORE
        if (j & 1) *reinterpret_cast<u32x4_t*>(sk + FP_KBYTES + r * FP_VLD + chunk * 16) = vreg[j >> 1];
        else *reinterpret_cast<u32x4_t*>(sk + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = kreg[j >> 1];
#endif
    };

    // ---- per-lane LDS offsets of the fragment reads
    // K fragment kk of a 32-key half tile (rows 32 PAR + fr): 16-byte chunk (2 kk + fh) ^ ((fr >> 1) & 7) of the row
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = fr * 128 + (((kk * 2 + fh) ^ ((fr >> 1) & 7)) << 4);
    // V^T fragment: lane i of a 16-lane group supplies row (i >> 2), columns 4 (i & 3).. of the 16-column block
    // ((lane >> 4) & 1); keys 4 fh + (i >> 2) of a 16-key k step (+ 8 for the high half of the fragment)
    const int li = lane & 15;
    const int voff = FP_KBYTES + (4 * fh + (li >> 2)) * FP_VLD + (((lane >> 4) & 1) * 16 + (li & 3) * 4) * 2;
    const unsigned lds0 = (unsigned)(uintptr_t)((const lds_char_t*)smem);

    // ---- state
    f32x16_t S[2][2];          // [half-tile parity][query block]: scores minus m, exp2 units
    u32x4_t P[2][2][2];        // [half-tile parity][query block][k step]: exp2(S) packed to bf16 = B operand of the PV MFMAs
    f32x16_t O[2][2];          // [query block][32-wide slice of d]
    f32x16_t L[2];             // row sums (every register of a lane holds the sum of its query row)
    f32x16_t NM[2];            // -m[x] in all 16 registers: C operand of the first MFMA of a score chain
    float m_run[2], mx[2][2];  // mx: two independent v_max3 chains per query block (tracking pass)
    float mm = 0.f;
    bf16x8_t Kf[4];
    bf16x8_t Vf[2][2];
    u32x4_t ONES = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+v"(ONES));

    // fragment reads take complete per-lane LDS addresses (kept in registers, advanced once per tile; the parity of the half
    // tile and the fragment inside the step are immediates)
    auto rd_k = [&](int slot, unsigned addr, int par) __attribute__((always_inline)) {
        Kf[slot] = *(lds_vfrag_t*)((const lds_char_t*)(uintptr_t)addr + par * 4096);
    };
    bf16x4_t vlo;
    auto rd_v_lo = [&](int ks, int db, unsigned addr, int par) __attribute__((always_inline)) {
        const lds_char_t* vp = (const lds_char_t*)(uintptr_t)addr + (par * 32 + 16 * ks) * FP_VLD + db * 64;
        vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp));
    };
    auto rd_v_hi = [&](int ks, int db, unsigned addr, int par) __attribute__((always_inline)) {
        const lds_char_t* vp = (const lds_char_t*)(uintptr_t)addr + (par * 32 + 16 * ks) * FP_VLD + db * 64;
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp + 8 * FP_VLD));
        bf16x8_t f;
        f[0] = vlo[0]; f[1] = vlo[1]; f[2] = vlo[2]; f[3] = vlo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        Vf[ks][db] = f;
    };

    // first = the row maximum of the first half tile becomes m. Otherwise (tracking pass only): some row's running max
    // grew by more than thr - everything still at the old max (O, the row sums, the packed P of the half tile whose PV is
    // pending) is multiplied by 2^-step exactly once, the scores of the half tile that raised the max are shifted by it
    auto rescale = [&](f32x16_t (&Sp)[2], u32x4_t (&Pp)[2][2], bool first) __attribute__((always_inline)) {
        // asm MFMA results (O, L, S) -> vector reads below: the padding statement takes the accumulators as in/out operands, so
        // hipcc cannot read them above it (it had hoisted the v_accvgpr_read of L out of the branch, right behind the row-sum
        // MFMAs - garbage on every rescale)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]),
                     "+a"(L[0]), "+a"(L[1]), "+v"(Sp[0]), "+v"(Sp[1]));
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float r = fmaxf(mx[x][0], mx[x][1]);
            r = fmaxf(r, __shfl_xor(r, 32, 64));
            const float step = first ? r : fmaxf(r, 0.f);
            m_run[x] += step;
            if (!first) {
                const float alpha = __builtin_amdgcn_exp2f(-step);
#pragma unroll
                for (int e = 0; e < 16; ++e) { O[x][0][e] *= alpha; O[x][1][e] *= alpha; L[x][e] *= alpha; }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w = Pp[x][ks][e];
                        Pp[x][ks][e] = pack_bf2(__uint_as_float(w << 16) * alpha, __uint_as_float(w & 0xffff0000u) * alpha);
                    }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { Sp[x][e] -= step; NM[x][e] = -m_run[x]; }
        }
        // vector writes above -> asm MFMA operands below
        asm volatile("s_nop 7\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]), "+a"(L[0]), "+a"(L[1]),
                     "+v"(Sp[0]), "+v"(Sp[1]), "+v"(NM[0]), "+v"(NM[1]), "+v"(Pp[0][0]), "+v"(Pp[0][1]), "+v"(Pp[1][0]),
                     "+v"(Pp[1][1]));
    };

    // One half-tile step = 16 gaps, + 4 for the row-sum MFMAs when it has PV. Every LDS fragment of a step is requested in the
    // last four gaps of the step BEFORE it (in steps with PV those are the row-sum gaps, which carry no vector work): kn =
    // addresses of the next step's four K fragments, vn = its V^T fragment base, NK / NV = does the next step need them.
    // The ring, per tile iteration T (steps 2T, 2T+1; tile X is last requested in step 2X, first in step 2X-3):
    //   MEM 1 (step 2T):   gaps 4..7 the four 16-byte LDS stores of tile T+2 (loaded an iteration ago: ~1500 cycles of cover
    //                      for the loads; a half-iteration was not enough, the stores waited ~400 cycles per tile) into the
    //                      stage of tile T-2; gaps 8..11 the four global loads of tile T+3 into the same staging registers;
    //   MEM 2 (step 2T+1): in front of gap 8 `s_waitcnt lgkmcnt(0)` + the workgroup barrier (the step's own requests went out
    //                      >= 8 gaps earlier, so the wait is free): behind it every wave's stores of tile T+2 are complete and
    //                      its requests of tile T-1 have returned. Tile T+2 is first requested in gaps 16..19 of this step.
    // In-gap order: the MFMA, this gap's two v_exp_f32, then the pack of the PREVIOUS gap's exponentials, v_max3 in between -
    // no statement directly follows one that produced an operand of it (hipcc pads such pairs of asm statements).
    auto step = [&](auto PAR_, auto DEC_, auto QK_, auto ZERO_, auto MX_, auto EX_, auto PV_, auto NK_, auto NV_, auto MEM_,
                    const unsigned (&kn)[4], int knpar, unsigned vn, int vnpar, int tile, int stage, unsigned vc = 0) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, MEM = decltype(MEM_)::value;
        constexpr bool DEC = decltype(DEC_)::value, QK = decltype(QK_)::value, ZERO = decltype(ZERO_)::value;
        constexpr bool MX = decltype(MX_)::value, EX = decltype(EX_)::value, PV = decltype(PV_)::value;
        constexpr bool NK = decltype(NK_)::value, NV = decltype(NV_)::value;
        constexpr int NG = PV ? 20 : 16;
        f32x16_t (&Sw)[2] = S[PAR];
        f32x16_t (&Sr)[2] = S[PAR ^ 1];
        u32x4_t (&Pr)[2][2] = P[PAR];
        u32x4_t (&Pw)[2][2] = P[PAR ^ 1];
        if constexpr (DEC) {
            if (__builtin_amdgcn_ballot_w64(mm > thr) != 0) rescale(Sr, Pr, false);
        }
        if constexpr (MX) { mx[0][0] = -3.0e38f; mx[0][1] = -3.0e38f; mx[1][0] = -3.0e38f; mx[1][1] = -3.0e38f; }
        float pa[16], pb[16];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // ---- memory instructions of this gap
#ifndef FP_DBG_NOBAR
            if constexpr (MEM == 2) { if (g == 8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
#ifndef FP_DBG_NOSTAGE
            if constexpr (MEM == 1) {
                if (g >= 4 && g < 8) store_piece(stage, g - 4);
                if (g >= 8 && g < 12) load_piece(tile, g - 8);
            }
#endif
#ifndef FP_DBG_NOLDS
            {
                const int j = g - (NG - 4);          // 0..3 in the last four gaps: three fragment reads each
                if constexpr (NK) {
                    if (j == 0) { rd_k(0, kn[0], knpar); rd_k(1, kn[1], knpar); rd_k(2, kn[2], knpar); }
                    if (j == 1) rd_k(3, kn[3], knpar);
                }
#ifndef FP_V_IN_STEP
                if constexpr (NV) {
                    if (j == 1) { rd_v_lo(0, 0, vn, vnpar); rd_v_hi(0, 0, vn, vnpar); }
                    if (j == 2) { rd_v_lo(0, 1, vn, vnpar); rd_v_hi(0, 1, vn, vnpar); rd_v_lo(1, 0, vn, vnpar); }
                    if (j == 3) { rd_v_hi(1, 0, vn, vnpar); rd_v_lo(1, 1, vn, vnpar); rd_v_hi(1, 1, vn, vnpar); }
                }
#else
                if constexpr (PV) {           // tool build: the step's own V^T fragments in its gaps 0..3
                    if (g == 0) { rd_v_lo(0, 0, vc, PAR); rd_v_hi(0, 0, vc, PAR); }
                    if (g == 1) { rd_v_lo(0, 1, vc, PAR); rd_v_hi(0, 1, vc, PAR); }
                    if (g == 2) { rd_v_lo(1, 0, vc, PAR); rd_v_hi(1, 0, vc, PAR); }
                    if (g == 3) { rd_v_lo(1, 1, vc, PAR); rd_v_hi(1, 1, vc, PAR); }
                }
#endif
            }
#endif
            // ---- the matrix instruction of this gap
            if (g < 8) {
                if constexpr (QK) {
                    const int kk = g >> 1, x = g & 1;
                    if (kk == 0) {
                        if constexpr (ZERO) FP_MFMA_SZ(Sw[x], Kf[0], Q[x][0]);
                        else FP_MFMA_S0(Sw[x], Kf[0], Q[x][0], NM[x]);
                    } else {
                        FP_MFMA_S(Sw[x], Kf[kk], Q[x][kk]);
                    }
                }
            } else if (g < 16) {
                if constexpr (PV) {
                    const int i = g - 8, ks = i >> 2, db = (i >> 1) & 1, x = i & 1;
                    FP_MFMA_O(O[x][db], Vf[ks][db], Pr[x][ks]);
                }
            } else {
                const int i = g - 16, ks = i >> 1, x = i & 1;
                FP_MFMA_O(L[x], ONES, Pr[x][ks]);
            }
            if (g >= 16) continue;
            // ---- vector work of this gap
            // tracking pass: 16 v_max3 (8 per query block) in gaps 10..15: the last MFMA of block 0's chain is gap 6, of block
            // 1's gap 7, and an MFMA result must not be read by a vector instruction within ~20 issue slots (no interlock, no
            // padding inside asm)
            constexpr int first_of[7] = {0, 3, 6, 9, 12, 14, 16};
            int mj = 16, mj_end = 16;
            if constexpr (MX) {
                if constexpr (!PV) { if (g == 10) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); }
                if (g >= 10) { mj = first_of[g - 10]; mj_end = first_of[g - 9]; }
            }
            auto max3_one = [&]() __attribute__((always_inline)) {
                if (mj < mj_end) {
                    const int x = mj >> 3, e = (mj & 7) * 2;
                    FP_MAX3(mx[x][mj & 1], Sw[x][e], Sw[x][e + 1]);
                    ++mj;
                }
            };
            const int ex = g >> 3, ei = g & 7;
            if constexpr (EX) FP_EXP(pa[g], Sr[ex][2 * ei]);
            max3_one();
            if constexpr (EX) FP_EXP(pb[g], Sr[ex][2 * ei + 1]);
            max3_one();
            if constexpr (EX) {
                if (g > 0) {
                    const int px = (g - 1) >> 3, pi = (g - 1) & 7;
                    max3_one();
                    unsigned w;
                    FP_CVT(w, pa[g - 1], pb[g - 1]);
                    Pw[px][pi >> 2][pi & 3] = w;
                }
            }
            max3_one();
            if (g == 15) {
                float t3;
                if constexpr (MX) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t3) : "v"(mx[0][0]), "v"(mx[0][1]), "v"(mx[1][0]));
                if constexpr (EX) asm volatile("s_nop 1");
                if constexpr (MX) asm volatile("v_max_f32 %0, %1, %2" : "=v"(mm) : "v"(t3), "v"(mx[1][1]));
                if constexpr (EX) {
                    unsigned w;
                    FP_CVT(w, pa[15], pb[15]);
                    Pw[1][1][3] = w;
                }
            }
        }
    };

    // ---- one pass over all keys. TRACK = false: m = row max of the first half tile, no v_max3 / decisions afterwards
    auto run = [&](auto TRACK_) __attribute__((always_inline)) {
        using TRACK = std::bool_constant<decltype(TRACK_)::value>;
        using EXL = std::bool_constant<FP_LOOP_EX>;
        using T_ = std::true_type;
        using F_ = std::false_type;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            mx[x][0] = 0.f; mx[x][1] = 0.f; m_run[x] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[x][0][r] = 0.f; O[x][1][r] = 0.f; L[x][r] = 0.f; }
        }
        mm = 0.f;
        // prologue: tiles 0 and 1 into the ring, tile 2 into the staging registers (stored by the loop's first step)
        {
            u32x4_t kk0[2], vv0[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) load_piece(0, j);
#pragma unroll
            for (int i = 0; i < 2; ++i) { kk0[i] = kreg[i]; vv0[i] = vreg[i]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) load_piece(1, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) store_piece(1, j);
#pragma unroll
            for (int i = 0; i < 2; ++i) { kreg[i] = kk0[i]; vreg[i] = vv0[i]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) store_piece(0, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) load_piece(2, j);
        }
        __syncthreads();
        unsigned ka[4], kb2[4], va, vb2;         // K / V fragment addresses of two ring stages
        auto set_addr = [&](unsigned (&kx)[4], unsigned& vx, int kst, int vst) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) { kx[kk] = lds0 + kst * FP_STAGE + koff[kk]; asm volatile("" : "+v"(kx[kk])); }
            vx = lds0 + vst * FP_STAGE + voff;
            asm volatile("" : "+v"(vx));         // opaque: or hipcc re-derives the addresses at every read
        };
        set_addr(ka, va, 0, 0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) rd_k(kk, ka[kk], 0);
        // h = -2: S(0) = K(0) cQ^T and its row max; requests K(1)
        step(ic<0>{}, F_{}, T_{}, T_{}, T_{}, F_{}, F_{}, T_{}, F_{}, ic<0>{}, ka, 1, va, 0, 0, 0);
        rescale(S[0], P[0], true);
        // h = -1: S(1) = K(1) cQ^T - m, P(0); requests K(2) (tile 1) and V(0)
        set_addr(kb2, vb2, 1, 0);
        step(ic<1>{}, F_{}, T_{}, F_{}, TRACK{}, T_{}, F_{}, T_{}, T_{}, ic<0>{}, kb2, 0, va, 0, 0, 0);

#ifdef FP_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
        // tile iteration T: steps 2T, 2T+1 = QK of tile T+1, PV of tile T; stores of tile T+2, loads of tile T+3
        for (int T = 0; T < nt - 1; ++T) {
            set_addr(ka, va, (T + 1) & 3, T & 3);            // this iteration's second halves: K(T+1) rows 32.., V(T) rows 32..
            set_addr(kb2, vb2, (T + 2) & 3, (T + 1) & 3);    // the next iteration's first halves: K(T+2), V(T+1)
            step(ic<0>{}, TRACK{}, T_{}, F_{}, TRACK{}, EXL{}, T_{}, T_{}, T_{}, ic<1>{}, ka, 1, va, 1, T + 3, (T + 2) & 3, va);
            step(ic<1>{}, TRACK{}, T_{}, F_{}, TRACK{}, EXL{}, T_{}, T_{}, T_{}, ic<2>{}, kb2, 0, vb2, 0, 0, 0, va);
        }
#ifdef FP_STAMPS
        if (tid == 0) {
            const unsigned long long st1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
            atomicAdd(&g_fp_stamps[0], st1 - st0); atomicAdd(&g_fp_stamps[1], sr1 - sr0);
            atomicAdd(&g_fp_stamps[2], (unsigned long long)(nt - 1)); atomicAdd(&g_fp_stamps[3], 1ull);
        }
#endif
        // the last tile's PV (its V fragments for the first half were requested by the loop's last step)
        set_addr(ka, va, 0, (nt - 1) & 3);
        step(ic<0>{}, TRACK{}, F_{}, F_{}, F_{}, T_{}, T_{}, F_{}, T_{}, ic<0>{}, ka, 0, va, 1, 0, 0, va);
        step(ic<1>{}, F_{}, F_{}, F_{}, F_{}, F_{}, T_{}, F_{}, F_{}, ic<0>{}, ka, 0, va, 0, 0, 0, va);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]),
                     "+a"(L[0]), "+a"(L[1]));
    };

    float l_tot[2];
    auto row_sums = [&]() __attribute__((always_inline)) {
        bool bad = false;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            l_tot[x] = L[x][0];
            bad |= !(l_tot[x] < 1.2676506e30f);          // 2^100; also true for NaN
        }
        return bad;
    };
    if (!force_track) run(std::false_type{});
#if defined(FP_DBG_NOEX) || defined(FP_DBG_NOLDS)
    const bool bad = (row_sums(), false);
#else
    const bool bad = force_track ? true : row_sums();
#endif
    if (__syncthreads_or(bad ? 1 : 0)) {                 // whole workgroup: the K / V ring and its barriers are shared
        run(std::true_type{});
        row_sums();
    }

    // ---- epilogue: normalise, bf16, store
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const float inv = 1.0f / l_tot[x];
        if (qrow[x] < Lq) {
            bf16_t* orow = ob + (size_t)qrow[x] * ldo;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int d = db * 32 + 8 * qd + 4 * fh;
                    uint2 pk;
                    pk.x = pack_bf2(O[x][db][4 * qd + 0] * inv, O[x][db][4 * qd + 1] * inv);
                    pk.y = pack_bf2(O[x][db][4 * qd + 2] * inv, O[x][db][4 * qd + 3] * inv);
                    *reinterpret_cast<uint2*>(orow + d) = pk;
                }
        }
    }
}

}  // namespace

#ifdef FP_STAMPS
extern "C" int dc_fp_debug_stamps(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fp_stamps), sizeof(unsigned long long) * 4);
    if (e == hipSuccess && reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_fp_stamps), z, sizeof(z));
    }
    return (int)e;
}
#endif

// Mode of the long self-attention kernel (process-wide; tests and same-box A/B): bit 0 = run the tracking (online-softmax)
// pass directly instead of as the fallback; thr = threshold of the tracking pass, in exp2 units, by which a score must
// exceed the running max before the state is rescaled (P <= 2^thr otherwise). Initial values from DC_FLASH_TRACK /
// DC_FLASH_THR.
static int g_fp_mode = [] { const char* t = getenv("DC_FLASH_TRACK"); return (t && t[0] == '1') ? 1 : 0; }();
static float g_fp_thr = [] { const char* e = getenv("DC_FLASH_THR"); return e ? (float)atof(e) : 8.0f; }();
extern "C" int dc_flash_attn_set_mode(int mode, float thr) {
    if (mode < 0 || mode > 1 || !(thr >= 0.f) || thr > 64.f) return DC_ERR_ARG;
    g_fp_mode = mode; g_fp_thr = thr;
    return 0;
}

// Launcher used by dc_flash_attn_d64 (attention.hip) for the shapes this kernel covers: Lk a multiple of 64 and >= 256,
// no accumulate epilogue. Returns 0 or a hipError_t.
int dc_flash_pipe_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, int ldq, int ldk, int ldv, int ldo,
                         int batch, int heads, int Lq, int Lk, int64_t q_bstride, int64_t kv_bstride, float c,
                         hipStream_t stream) {
    const float thr = g_fp_thr;
    const int force_track = g_fp_mode & 1;
    static bool configured[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DC_ERR_ARG;
    if (dev < 0 || dev >= 16 || !configured[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)flash_attn_d64_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0 && dev < 16) configured[dev] = true;
    }
    const int q_tiles = (Lq + FP_ROWS - 1) / FP_ROWS;
    const long long nwg = (long long)q_tiles * heads * batch;
    if (nwg > 0x7fffffffLL) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(flash_attn_d64_pipe_kernel, dim3((unsigned)nwg), dim3(256), FP_LDS, stream, q, k, v, o, ldq, ldk, ldv, ldo,
                       heads, Lq, Lk, q_bstride, kv_bstride, c, q_tiles, thr, force_track);
    DC_CHECK_LAUNCH();
    return 0;
}
