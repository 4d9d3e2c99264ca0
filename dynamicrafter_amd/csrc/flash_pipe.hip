// Spatial self-attention, head_dim 64, for gfx950: ONE wave per SIMD, software-pipelined by hand.
//
//   o = softmax(q k^T * scale) v        reference: CrossAttention.forward, lvdm/modules/attention.py:101-125
//
// The two-waves-per-SIMD kernel of attention.hip spends ~1750 cycles per (64 queries x 64 keys) of a wave where the matrix
// pipe needs 1024: its two waves reach the exp / pack / max stream together. Here a workgroup is 4 waves, a wave owns QB
// 32-row query blocks (QB = 3: 384 rows per workgroup) and the whole register file, and the vector work of one 32-key HALF
// TILE is placed by hand in the gaps between the MFMAs of its neighbours:
//
//   step h (10 QB MFMAs): gaps 0..4QB-1    S(h+2) = K(h+2) (cQ)^T        [QK^T of the half tile two ahead]
//                         gaps 4QB..8QB-1  O^T  += V(h)^T P(h)^T         [PV of the current half tile]
//                         gaps 8QB..10QB-1 l += 1^T P(h)^T               [row sums on the matrix pipe, a ones fragment as A]
//   beside them:          gaps 0..8QB-1    P(h+1) = exp2(S(h+1)), bf16 pack            (2 v_exp_f32 + 1 pack per gap)
//                         gaps 8QB..       every memory instruction: the 12 LDS fragment requests of step h+1 and, in even
//                                          steps, the 4 LDS stores + 4 global loads of the K/V ring
//
// What bounds it is the issue port of the one wave (tools/ubench/gapcost.hip: an MFMA gap hides 2 v_exp_f32 + 1 pack, 33.5
// clocks; two more v_add_f32 make it 41; tools/flash_variants.sh + flash_stamps.py on this kernel: a memory instruction inside
// an exp gap costs ~40 clocks, in a bare MFMA gap ~10), so the vector stream is kept to what softmax cannot do without - one
// v_exp_f32 per score, one pack per pair - and everything else sits in the row-sum gaps:
//   * NO RUNNING MAX in the main pass: P = exp2(s) with the shift m = 0. Softmax is invariant under the shift, exp2 keeps its
//     relative precision anywhere in the fp32 / bf16 exponent range, and fp32 sums of 9216 terms are safe while every term
//     is below 2^100. A row whose scores leave +-100 (69 nats) shows up as a row sum outside [2^-100, 2^100) (or NaN): then
//     the whole workgroup repeats its block with the tracking pass (TRACK: v_max3 chains beside the PV MFMAs, the rescale
//     decision one half tile late, threshold `thr`) - the classical online softmax, kept as the fallback that makes the
//     kernel total.
//   * row sums as 2QB more MFMAs per step instead of 16QB v_add_f32: the matrix pipe has the room, the issue port does not;
//     their gaps carry no vector work and are where the memory instructions go.
//   * QB = 3: the 12 fragment requests and 8 ring instructions of a tile are shared by 1.5x the MFMAs.
//
// Every instruction of the stream is an `asm volatile` statement: hipcc allocates the registers and counts the LDS / buffer
// loads, the order is the source order (volatile statements are not reordered among themselves). Distances that the
// hardware does not interlock and hipcc does not pad inside asm are kept by construction: an MFMA result is first read by a
// vector instruction >= 2 gaps later; a v_exp_f32 result is consumed >= 2 instructions later; P is packed a step before the
// MFMA that reads it; compiler-generated code that touches MFMA results sits behind a padding statement that takes them as
// in/out operands (hipcc had hoisted v_accvgpr_read of the row sums above a plain s_nop statement).
// S lives in architectural VGPRs (the exp / max stream reads it), O, l and Q in the accumulator file.
// K / V tiles of 64 keys: global -> registers -> LDS ring of 4 stages, one barrier per tile; K fragments by ds_read_b128
// (XOR-swizzled rows), V^T fragments by ds_read_b64_tr_b16.
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
constexpr int FP_RING = FP_NSTAGE * FP_STAGE;       // 80 KB
constexpr int FP_LDS = FP_RING + 64;                // + the arrival counter of the ring

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
// S chain (D, C in VGPRs; A = K fragment and B = Q fragment in the accumulator file: the LDS reads land there directly, the
// architectural VGPRs are kept for what the vector ALU touches - S, P, addresses)
#define FP_MFMA_SZ(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "a"(a), "a"(b))
#define FP_MFMA_S(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b))
// O and row-sum chains (D = C in the accumulator file, B = P in VGPRs). The V^T fragments stay in VGPRs: they are assembled
// from two transposed reads, and as "a" operands hipcc put the v_accvgpr_write copies right in front of the MFMA that reads
// them - a hazard it does not pad for an asm statement (wrong results, no fault)
#define FP_MFMA_O(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b))
#define FP_MFMA_L(d, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(d) : "a"(a), "v"(b))


// padding statement that takes the accumulators of all query blocks as in/out operands (see the header)
template <int QB>
__device__ __forceinline__ void fp_settle(f32x16_t (&O)[QB][2], f32x16_t (&L)[QB]) {
    if constexpr (QB == 2)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]), "+a"(L[0]),
                     "+a"(L[1]));
    else
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[1][0]), "+a"(O[1][1]), "+a"(O[2][0]),
                     "+a"(O[2][1]), "+a"(L[0]), "+a"(L[1]), "+a"(L[2]));
}

template <int QB>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_waves_per_eu(1, 1)))
void flash_attn_d64_pipe_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                bf16_t* __restrict__ o, int ldq, int ldk, int ldv, int ldo, int heads, int Lq, int Lk,
                                int64_t q_bstride, int64_t kv_bstride, float c /* scale*log2(e) */, int q_tiles, float thr,
                                int force_track, int* __restrict__ err /* the library's error word (runtime.hip) */) {
    static_assert(QB == 2 || QB == 3, "query blocks per wave");
    constexpr int ROWS = 128 * QB;        // query rows per workgroup
    constexpr int NQK = 4 * QB;           // score MFMAs of a step (gaps 0 .. NQK-1)
    constexpr int NEX = 8 * QB;           // gaps that carry an exp pair; gaps NQK .. NEX-1 hold the PV MFMAs
    constexpr int NL = 2 * QB;            // row-sum MFMAs (gaps NEX .. NEX+NL-1)
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

    // ---- Q fragments (B operand of S^T = K (cQ)^T): lane (r, h) holds cQ[row][16kk + 8h .. +7] of its query blocks
    int qrow[QB];
    bf16x8_t Q[QB][4];
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        qrow[x] = qt * ROWS + (wave * QB + x) * 32 + fr;
        const int qc = qrow[x] < Lq ? qrow[x] : Lq - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // Q is pre-multiplied by scale*log2(e) (one more bf16 rounding of Q, the size of the one it already has): the
            // scores leave the MFMA in exp2 units
            const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qb + (size_t)qc * ldq + kk * 16 + fh * 8);
            u32x4_t sc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[e] = pack_bf2(__uint_as_float(raw[e] << 16) * c, __uint_as_float(raw[e] & 0xffff0000u) * c);
            Q[x][kk] = __builtin_bit_cast(bf16x8_t, sc);
            asm volatile("" : "+a"(Q[x][kk]));        // lives in the accumulator file from here on (B operand only)
        }
    }

    // ---- staging: 64 rows x 8 chunks of 16 B per tensor and tile, 2 rows per thread.
    // piece j of a tile: 0 = K rows 0..31, 1 = V rows 0..31, 2 = K rows 32..63, 3 = V rows 32..63 (one 16-byte load / store each).
    // Buffer loads: descriptor of the (batch, head) slice in SGPRs, the lane's offset in a VGPR that never changes, the tile's
    // offset in an SGPR - one instruction per piece, no per-lane address arithmetic
    const int chunk = tid & 7, srow = tid >> 3;
    const int nt = Lk >> 6;
    u32x4_t kreg[2], vreg[2];
    const unsigned kgo = (unsigned)(srow * ldk + chunk * 8) * 2u, vgo = (unsigned)(srow * ldv + chunk * 8) * 2u;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
    auto load_piece = [&](int t, int j) __attribute__((always_inline)) {
        t = t < nt ? t : nt - 1;                      // past the end: the last tile again (finite data, never used)
        const int row0 = t * 64 + (j >> 1) * 32;      // rows 32.. of the tile through the scalar offset
#ifndef FP_DBG_NOLOAD
        if (j & 1) vreg[j >> 1] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vgo, row0 * 2 * ldv, 0);
        else kreg[j >> 1] = __builtin_amdgcn_raw_buffer_load_b128(krs, kgo, row0 * 2 * ldk, 0);
#endif
    };
    // per-lane LDS store offsets inside a stage (K rows XOR-swizzled by 16-byte chunk - the swizzle of row r + 32 is that of row
    // r -, V rows padded to FP_VLD): rows 32.. are immediates
    const unsigned kso = (unsigned)(srow * 128 + ((chunk ^ ((srow >> 1) & 7)) << 4));
    const unsigned vso = (unsigned)(FP_KBYTES + srow * FP_VLD + chunk * 16);
    auto store_piece = [&](int stage, int j) __attribute__((always_inline)) {
        char* sk = smem + stage * FP_STAGE;
#ifndef FP_DBG_NOSTORE
        if (j & 1) *reinterpret_cast<u32x4_t*>(sk + vso + (j >> 1) * 32 * FP_VLD) = vreg[j >> 1];
        else *reinterpret_cast<u32x4_t*>(sk + kso + (j >> 1) * 4096) = kreg[j >> 1];
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
    f32x16_t S[2][QB];         // [half-tile parity][query block]: scores (minus m in the tracking pass), exp2 units
    u32x4_t P[2][QB][2];       // [half-tile parity][query block][k step]: exp2(S) packed to bf16 = B operand of the PV MFMAs
    f32x16_t O[QB][2];         // [query block][32-wide slice of d]
    f32x16_t L[QB];            // row sums (every register of a lane holds the sum of its query row)
    float m_run[QB], mx[QB][2];   // tracking pass: running max; two independent v_max3 chains per query block
    float mm = 0.f;
    bf16x8_t Kf[4];
    bf16x8_t Vf[2][2];
    u32x4_t ONES = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    asm volatile("" : "+a"(ONES));

    // fragment reads take complete per-lane LDS addresses (kept in registers, advanced once per tile; the parity of the half
    // tile and the fragment inside the step are immediates)
    auto rd_k = [&](int slot, unsigned addr, int par) __attribute__((always_inline)) {
        Kf[slot] = *(lds_vfrag_t*)((const lds_char_t*)(uintptr_t)addr + par * 4096);
    };
    bf16x4_t vlo;
    auto rd_v_half = [&](int idx, unsigned addr, int par) __attribute__((always_inline)) {      // idx = 4 ks + 2 db + (0 lo, 1 hi)
        const int ks = idx >> 2, db = (idx >> 1) & 1;
        const lds_char_t* vp = (const lds_char_t*)(uintptr_t)addr + (par * 32 + 16 * ks) * FP_VLD + db * 64;
        if (!(idx & 1)) {
            vlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp));
        } else {
            const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp + 8 * FP_VLD));
            bf16x8_t f;
            f[0] = vlo[0]; f[1] = vlo[1]; f[2] = vlo[2]; f[3] = vlo[3];
            f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
            Vf[ks][db] = f;
        }
    };

    // Tracking pass only. first = the row maximum of the first half tile becomes m. Otherwise some row's running max grew by
    // more than thr: everything still at the old max (O, the row sums, the packed P of the half tile whose PV is pending) is
    // multiplied by 2^-step exactly once, the scores of the half tile that raised the max are shifted by it
    auto rescale = [&](f32x16_t (&Sp)[QB], u32x4_t (&Pp)[QB][2], bool first) __attribute__((always_inline)) {
        fp_settle<QB>(O, L);                                      // asm MFMA results -> vector reads below
#pragma unroll
        for (int x = 0; x < QB; ++x) asm volatile("" : "+v"(Sp[x]));
#pragma unroll
        for (int x = 0; x < QB; ++x) {
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
            for (int e = 0; e < 16; ++e) Sp[x][e] -= step;
        }
        fp_settle<QB>(O, L);                                      // vector writes above -> asm MFMA operands below
#pragma unroll
        for (int x = 0; x < QB; ++x) asm volatile("" : "+v"(Sp[x]), "+v"(Pp[x][0]), "+v"(Pp[x][1]));
    };

    // One half-tile step. TR: tracking pass (score chains start at -m, v_max3 chains beside the PV MFMAs, decision DEC at the
    // top). The memory instructions sit in the last NL gaps (the row-sum gaps when the step has PV), spread evenly in this
    // order: the next step's four K fragments (kn: their addresses, NK: needed) and eight V^T fragment halves (vn, NV), then,
    // RING 1 (even steps of the loop), the four LDS stores of tile T+2 (loaded an iteration ago) into the stage of tile T-2 and
    // the four global loads of tile T+3 into the same staging registers, then the wave's arrival. RING 2 (odd steps): in front
    // of those gaps the check that all four waves have arrived (`tile` = 4 (T + 1)): then every wave's stores of tile T+2 are
    // complete - it is first requested right there - and its requests of tile T-1 have returned.
    // In-gap order: the MFMA, this gap's two v_exp_f32, then the pack of the PREVIOUS gap's exponentials, v_max3 in between -
    // no statement directly follows one that produced an operand of it (hipcc pads such pairs of asm statements).
    __attribute__((address_space(3))) int* const ring_cnt = (__attribute__((address_space(3))) int*)(smem + FP_RING);
    int seen = 0, gave_up = 0;
    auto step = [&](auto PAR_, auto TR_, auto DEC_, auto QK_, auto MX_, auto EX_, auto PV_, auto NK_, auto NV_, auto RING_,
                    const unsigned (&kn)[4], int knpar, unsigned vn, int vnpar, int tile, int stage) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PAR_)::value, RING = decltype(RING_)::value;
        constexpr bool TR = decltype(TR_)::value, DEC = decltype(DEC_)::value, QK = decltype(QK_)::value;
        constexpr bool MX = decltype(MX_)::value, EX = decltype(EX_)::value, PV = decltype(PV_)::value;
        constexpr bool NK = decltype(NK_)::value, NV = decltype(NV_)::value;
        constexpr int NG = PV ? NEX + NL : NEX;
        constexpr int NSLOT = (NK ? 4 : 0) + (NV ? 8 : 0) + (RING == 1 ? 8 : 0);
        f32x16_t (&Sw)[QB] = S[PAR];
        f32x16_t (&Sr)[QB] = S[PAR ^ 1];
        u32x4_t (&Pr)[QB][2] = P[PAR];
        u32x4_t (&Pw)[QB][2] = P[PAR ^ 1];
        if constexpr (DEC) {
            if (__builtin_amdgcn_ballot_w64(mm > thr) != 0) rescale(Sr, Pr, false);
        }
        if constexpr (MX) {
#pragma unroll
            for (int x = 0; x < QB; ++x) { mx[x][0] = -3.0e38f; mx[x][1] = -3.0e38f; }
        }
        float pa[NEX], pb[NEX];
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            // ---- memory instructions of this gap
            const int j = g - (NG - NL);              // 0 .. NL-1 in the last NL gaps
#ifndef FP_DBG_NOBAR
#ifdef FP_HW_BARRIER
            if constexpr (RING == 2) { if (j == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
            // the ring's synchronisation without s_barrier (which cost ~280 clocks per tile: every wave waits for the slowest at
            // every tile): a wave ARRIVES by adding 1 to an LDS counter behind its stores of tile T+2 (one wave's LDS operations
            // execute in order: whoever sees the add sees the stores, and the wave's own requests of older tiles have returned),
            // and a step later, before it requests tile T+2, checks that all four have arrived at this iteration - normally
            // long true; the value is requested four gaps ahead of the check. Every wave arrives once per step, so the wait always
            // ends; the bound (about 10 ms, once per wave) keeps a future bookkeeping mistake from hanging the GPU: the wave then
            // carries on with whatever the ring holds and the kernel reports DC_ERRW_FLASH_RING in the error word
            if constexpr (RING == 2) {
                if (g == NEX - 4) seen = *(volatile __attribute__((address_space(3))) int*)ring_cnt;
                if (j == 0) {
                    int spins = 0;
                    while (!gave_up && __builtin_amdgcn_readfirstlane(seen) < tile) {
                        seen = *(volatile __attribute__((address_space(3))) int*)ring_cnt;
                        if (++spins > 200000) gave_up = 1;
                    }
                    asm volatile("" ::: "memory");
                }
            }
#endif
#endif
            if constexpr (NSLOT > 0) {
#pragma unroll
                for (int sl = 0; sl < NSLOT; ++sl) {
                    if (j < 0 || sl * NL / NSLOT != j) continue;
                    int idx = sl;
#ifndef FP_DBG_NOLDS
                    if (NK && idx < 4) rd_k(idx, kn[idx], knpar);
#endif
                    if (NK) idx -= 4;
#ifndef FP_DBG_NOLDS
                    if (NV && idx >= 0 && idx < 8) rd_v_half(idx, vn, vnpar);
#endif
                    if (NV) idx -= 8;
#ifndef FP_DBG_NOSTAGE
                    if (RING == 1 && idx >= 0 && idx < 4) store_piece(stage, idx);
                    if (RING == 1 && idx >= 4 && idx < 8) load_piece(tile, idx - 4);
#endif
                }
            }
#if !defined(FP_DBG_NOBAR) && !defined(FP_HW_BARRIER)
            // arrival: behind the last of this step's LDS stores (the slots above)
            if constexpr (RING == 1) { if (g == NG - 1 && lane == 0) __hip_atomic_fetch_add(ring_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
            // ---- the matrix instruction of this gap
            if (g < NQK) {
                if constexpr (QK) {
                    const int kk = g / QB, x = g % QB;
                    if (kk == 0) {
                        if constexpr (TR) {               // the chain starts at -m: 16 moves (the fallback pass may be slow)
#pragma unroll
                            for (int e = 0; e < 16; ++e) Sw[x][e] = -m_run[x];
                            asm volatile("s_nop 1" : "+v"(Sw[x]));
                            FP_MFMA_S(Sw[x], Kf[0], Q[x][0]);
                        } else {
                            FP_MFMA_SZ(Sw[x], Kf[0], Q[x][0]);
                        }
                    } else {
                        FP_MFMA_S(Sw[x], Kf[kk], Q[x][kk]);
                    }
                }
            } else if (g < NEX) {
                if constexpr (PV) {
                    const int i = g - NQK, ks = i / (2 * QB), db = (i / QB) & 1, x = i % QB;
                    FP_MFMA_O(O[x][db], Vf[ks][db], Pr[x][ks]);
                }
            } else {
                const int i = g - NEX, ks = i / QB, x = i % QB;
                FP_MFMA_L(L[x], ONES, Pr[x][ks]);
            }
            if (g >= NEX) continue;
            // ---- vector work of this gap
            // tracking pass: 8 QB v_max3 in gaps NQK+2 .. NEX-1: block x's score chain ends in gap NQK-QB+x, and an MFMA result
            // must not be read by a vector instruction within ~20 issue slots (no interlock, no padding inside asm)
            int mj = 0, mj_end = 0;
            if constexpr (MX) {
                constexpr int G0 = NQK + 2, NGM = NEX - G0, NOP = 8 * QB;
                if constexpr (!PV) { if (g == G0) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7"); }
                if (g >= G0) { mj = (g - G0) * NOP / NGM; mj_end = (g - G0 + 1) * NOP / NGM; }
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
            if (g == NEX - 1) {
                if constexpr (MX) {                  // mm = max over all chains (statements of one chain never adjacent)
                    float t0, t1;
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(mx[0][0]), "v"(mx[0][1]), "v"(mx[1][0]));
                    if constexpr (QB == 3) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(mx[1][1]), "v"(mx[2][0]), "v"(mx[2][1]));
                    else asm volatile("v_mov_b32 %0, %1" : "=v"(t1) : "v"(mx[1][1]));
                    asm volatile("s_nop 1");
                    asm volatile("v_max_f32 %0, %1, %2" : "=v"(mm) : "v"(t0), "v"(t1));
                }
                if constexpr (EX) {
                    asm volatile("s_nop 1");
                    unsigned w;
                    FP_CVT(w, pa[NEX - 1], pb[NEX - 1]);
                    Pw[QB - 1][1][3] = w;
                }
            }
        }
    };

    // ---- one pass over all keys. TRACK = false: m = 0, no v_max3 / decisions
    auto run = [&](auto TRACK_) __attribute__((always_inline)) {
        using TRACK = std::bool_constant<decltype(TRACK_)::value>;
        using EXL = std::bool_constant<FP_LOOP_EX>;
        using T_ = std::true_type;
        using F_ = std::false_type;
#pragma unroll
        for (int x = 0; x < QB; ++x) {
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
        if (tid == 0) *ring_cnt = 0;
        __syncthreads();
#ifdef FP_STAGGER
        // waves that run the same program in step hit the LDS and the issue ports together; the arrival counter tolerates
        // a skew of up to one step, so the waves can be started a fraction of a step apart (wave w sleeps w * FP_STAGGER * 64 clocks)
        for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(FP_STAGGER);
#endif
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
        // h = -2: S(0) = K(0) cQ^T (tracking pass: and its row max -> m); requests K(1)
        step(ic<0>{}, TRACK{}, F_{}, T_{}, TRACK{}, F_{}, F_{}, T_{}, F_{}, ic<0>{}, ka, 1, va, 0, 0, 0);
        if constexpr (TRACK::value) rescale(S[0], P[0], true);
        // h = -1: S(1), P(0); requests K(2) (tile 1) and V(0)
        set_addr(kb2, vb2, 1, 0);
        step(ic<1>{}, TRACK{}, F_{}, T_{}, TRACK{}, T_{}, F_{}, T_{}, T_{}, ic<0>{}, kb2, 0, va, 0, 0, 0);

#ifdef FP_STAMPS
        const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
        // tile iteration T: steps 2T, 2T+1 = QK of tile T+1, PV of tile T; stores of tile T+2, loads of tile T+3
        for (int T = 0; T < nt - 1; ++T) {
            set_addr(ka, va, (T + 1) & 3, T & 3);            // this iteration's second halves: K(T+1) rows 32.., V(T) rows 32..
            set_addr(kb2, vb2, (T + 2) & 3, (T + 1) & 3);    // the next iteration's first halves: K(T+2), V(T+1)
            step(ic<0>{}, TRACK{}, TRACK{}, T_{}, TRACK{}, EXL{}, T_{}, T_{}, T_{}, ic<1>{}, ka, 1, va, 1, T + 3, (T + 2) & 3);
            step(ic<1>{}, TRACK{}, TRACK{}, T_{}, TRACK{}, EXL{}, T_{}, T_{}, T_{}, ic<2>{}, kb2, 0, vb2, 0, 4 * (T + 1), 0);
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
        step(ic<0>{}, TRACK{}, TRACK{}, F_{}, F_{}, T_{}, T_{}, F_{}, T_{}, ic<0>{}, ka, 0, va, 1, 0, 0);
        step(ic<1>{}, TRACK{}, F_{}, F_{}, F_{}, F_{}, T_{}, F_{}, F_{}, ic<0>{}, ka, 0, va, 0, 0, 0);
        fp_settle<QB>(O, L);
    };

    float l_tot[QB];
    auto row_sums = [&]() __attribute__((always_inline)) {
        bool bad = false;
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            l_tot[x] = L[x][0];
            bad |= !(l_tot[x] < 1.2676506e30f) || !(l_tot[x] > 7.8886091e-31f);     // outside [2^-100, 2^100), or NaN
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

    if (gave_up && lane == 0) atomicOr(err, DC_ERRW_FLASH_RING);

    // ---- epilogue: normalise, bf16, store
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        const float inv = 1.0f / l_tot[x];
        // a row's 8 channels 8 qd .. 8 qd + 7 are split over its two lanes (fh = 0 / 1: 4 each). One v_permlane32_swap per
        // dword between the groups qd and qd + 1 gives the lower lane the 16 contiguous bytes of group qd and the upper lane
        // those of group qd + 1: 4 16-byte stores per row block instead of 8 8-byte ones (the store tail is issue-bound)
        bf16_t* orow = ob + (size_t)(qrow[x] < Lq ? qrow[x] : Lq - 1) * ldo;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; qd += 2) {
                unsigned a0 = pack_bf2(O[x][db][4 * qd + 0] * inv, O[x][db][4 * qd + 1] * inv);
                unsigned a1 = pack_bf2(O[x][db][4 * qd + 2] * inv, O[x][db][4 * qd + 3] * inv);
                unsigned b0 = pack_bf2(O[x][db][4 * qd + 4] * inv, O[x][db][4 * qd + 5] * inv);
                unsigned b1 = pack_bf2(O[x][db][4 * qd + 6] * inv, O[x][db][4 * qd + 7] * inv);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4_t pk = {r0[0], r1[0], r0[1], r1[1]};
                if (qrow[x] < Lq) *reinterpret_cast<u32x4_t*>(orow + db * 32 + 8 * (qd + fh)) = pk;
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
// pass directly instead of as the fallback; bit 1 = two query blocks per wave (256 rows per workgroup) for every shape (default:
// three when Lq is a multiple of 384); thr = threshold of the tracking pass, in exp2 units, by which a score must exceed the
// running max before the state is rescaled (P <= 2^thr otherwise). Initial values from DC_FLASH_TRACK / DC_FLASH_QB2 /
// DC_FLASH_THR. (The 16x16x32 main pass that bit 2 once selected lives in tools/experimental/flash_pipe16.hip.)
static std::atomic<int> g_fp_mode{[] {
    const char* t = getenv("DC_FLASH_TRACK"); const char* b = getenv("DC_FLASH_QB2");
    return ((t && t[0] == '1') ? 1 : 0) | ((b && b[0] == '1') ? 2 : 0);
}()};
static std::atomic<float> g_fp_thr{[] { const char* e = getenv("DC_FLASH_THR"); const float v = e ? (float)atof(e) : 8.0f; return (v >= 0.f && v <= 64.f) ? v : 8.0f; }()};
extern "C" int dc_flash_attn_set_mode(int mode, float thr) {
    if (mode < 0 || mode > 3 || !(thr >= 0.f) || thr > 64.f) return DC_ERR_ARG;
    g_fp_mode.store(mode, std::memory_order_relaxed); g_fp_thr.store(thr, std::memory_order_relaxed);
    return 0;
}

// Launcher used by dc_flash_attn_d64 (attention.hip) for the shapes this kernel covers: Lk a multiple of 64 and >= 256,
// no accumulate epilogue. Returns 0 or a hipError_t.
int dc_flash_pipe_launch(const bf16_t* q, const bf16_t* k, const bf16_t* v, bf16_t* o, int ldq, int ldk, int ldv, int ldo,
                         int batch, int heads, int Lq, int Lk, int64_t q_bstride, int64_t kv_bstride, float c,
                         hipStream_t stream) {
    const float thr = g_fp_thr.load(std::memory_order_relaxed);
    const int mode = g_fp_mode.load(std::memory_order_relaxed);
    const int force_track = mode & 1;
    int* const err = dc_error_word_device();
    if (!err) return DC_ERR_ARG;
    static DcLdsOnce once2, once3;
    if (const int e = once2.ensure((const void*)flash_attn_d64_pipe_kernel<2>, FP_LDS)) return e;
    if (const int e = once3.ensure((const void*)flash_attn_d64_pipe_kernel<3>, FP_LDS)) return e;
    const bool qb3 = !(mode & 2) && Lq % 384 == 0;
    const int rows = qb3 ? 384 : 256;
    const int q_tiles = (Lq + rows - 1) / rows;
    const long long nwg = (long long)q_tiles * heads * batch;
    if (nwg > 0x7fffffffLL) return DC_ERR_SHAPE;
    // the byte offsets of the K / V buffer loads are 32-bit
    if ((long long)Lk * ldk * 2 >= 0x7fffffffLL || (long long)Lk * ldv * 2 >= 0x7fffffffLL) return DC_ERR_SHAPE;
    if (qb3)
        hipLaunchKernelGGL(flash_attn_d64_pipe_kernel<3>, dim3((unsigned)nwg), dim3(256), FP_LDS, stream, q, k, v, o, ldq, ldk, ldv,
                           ldo, heads, Lq, Lk, q_bstride, kv_bstride, c, q_tiles, thr, force_track, err);
    else
        hipLaunchKernelGGL(flash_attn_d64_pipe_kernel<2>, dim3((unsigned)nwg), dim3(256), FP_LDS, stream, q, k, v, o, ldq, ldk, ldv,
                           ldo, heads, Lq, Lk, q_bstride, kv_bstride, c, q_tiles, thr, force_track, err);
    DC_CHECK_LAUNCH();
    return 0;
}
