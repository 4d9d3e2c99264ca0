// Attention kernels for gfx950, head_dim 64.
//
// dc_flash_attn_d64: softmax(q k^T * scale) v without materialising scores (spatial self-attention N = L up to
//   9216 tokens per frame; text (L=77) and image (L=16) cross-attention with the o += s*result epilogue).
//   reference: CrossAttention.forward lvdm/modules/attention.py:101-142.
//   Structure: workgroup = 4 waves x 32 query rows; KV tiles of 64 keys staged global -> registers -> LDS
//   (double-buffered, next tile's loads issued before the MFMA block). Scores are computed transposed,
//   S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so a lane owns one query row: the softmax max/sum are per-lane
//   scalars and the S^T accumulator is, after exp2 and bf16 packing, directly the B operand of O^T += V^T P^T
//   (k order permuted as the accumulator layout dictates; V^T fragments come from ds_read_b64_tr_b16).
//
// dc_temporal_attn_d64: attention across the T<=16 frames of one spatial position (one wave per
//   (clip, position, head)); reference: TemporalTransformer.forward attention.py:365-412 -> CrossAttention :81-144.
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((address_space(3))) bf16x4_t lds_bf16x4_t;

__device__ __forceinline__ int k_lds_off(int row, int chunk) {
    return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}
constexpr int V_LD = 192;  // bytes per V row in LDS: 4 consecutive rows land on 4 distinct 64-byte bank quarters

constexpr int FA_BQ = 128;   // query rows per workgroup
constexpr int FA_BKV = 64;   // keys per tile
constexpr int FA_KBYTES = FA_BKV * 128;
constexpr int FA_VBYTES = FA_BKV * V_LD;
constexpr int FA_STAGE = FA_KBYTES + FA_VBYTES;

// QB = 32-row query blocks per wave. QB = 2 reads every K / V fragment from LDS once for two query blocks: at QB = 1
// the kernel moves 16 KB of LDS reads per wave per 16 MFMAs, i.e. ~256 B/clk/CU at two workgroups per CU — the LDS
// bandwidth itself — so doubling the MFMAs per fragment is what lifts the bound.
// DUAL: two key/value sets with SEPARATE softmaxes over one query block - the text (k, v, Lk) and image (k2, v2, Lk2)
// halves of DynamiCrafter's cross-attention, out = attn_text + acc_scale * attn_image (attention.py:128-142) - in one
// pass over q and one store of o (as two launches the image pass re-read q and read-modify-wrote o).
template <int QB, bool DUAL>
__global__ __launch_bounds__(256, 2) void flash_attn_d64_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, bf16_t* __restrict__ o,
    int ldq, int ldk, int ldv, int ldo, int heads, int Lq, int Lk_first, int64_t q_bstride, int64_t kv_bstride,
    float c /* scale*log2(e) */, int accumulate, float acc_scale, int q_tiles, const bf16_t* __restrict__ k2,
    const bf16_t* __restrict__ v2, int Lk2) {
    static_assert(!DUAL || QB == 1, "the dual form keeps a second output accumulator: one query block per wave");
    int Lk = Lk_first;
    __shared__ __attribute__((aligned(16))) char smem[2 * FA_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // ---- Q fragments (B operand): lane (r, h) holds Q[row][16kk + 8h .. +7] for each of its QB query blocks
    int qrow[QB];
    bf16x8_t qf[QB][4];
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        qrow[x] = qt * (FA_BQ * QB) + (wave * QB + x) * 32 + fr;
        const int qc = qrow[x] < Lq ? qrow[x] : Lq - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            // Q is pre-multiplied by scale*log2(e) (one extra bf16 rounding of Q, the size of the one it already has):
            // the scores then leave the MFMA in exp2 units and, with the accumulator started at -m, as s - m directly
            const u32x4_t raw = *reinterpret_cast<const u32x4_t*>(qb + (size_t)qc * ldq + kk * 16 + fh * 8);
            u32x4_t sc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[e] = pack_bf2(__uint_as_float(raw[e] << 16) * c, __uint_as_float(raw[e] & 0xffff0000u) * c);
            qf[x][kk] = __builtin_bit_cast(bf16x8_t, sc);
        }
    }

    // ---- staging coordinates: 64 rows x 8 chunks per tensor, 2 rows per thread
    const int chunk = tid & 7, srow = tid >> 3;
    u32x4_t kreg[2], vreg[2];
    // Unconditional loads: rows past Lk are clamped to the last key (finite data). Their scores are masked to
    // -1e30 below, so P = 0 there and a finite V contributes exactly 0 — no zero fill, no branch around a load.
    auto load_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int j = t * FA_BKV + srow + 32 * i;
            j = j < Lk ? j : Lk - 1;
            kreg[i] = *reinterpret_cast<const u32x4_t*>(kb + (size_t)j * ldk + chunk * 8);
            vreg[i] = *reinterpret_cast<const u32x4_t*>(vb + (size_t)j * ldv + chunk * 8);
        }
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        char* sk = smem + buf * FA_STAGE;
        char* sv = sk + FA_KBYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = srow + 32 * i;
            *reinterpret_cast<u32x4_t*>(sk + k_lds_off(r, chunk)) = kreg[i];
            *reinterpret_cast<u32x4_t*>(sv + r * V_LD + chunk * 16) = vreg[i];
        }
    };

    f32x16_t oacc[QB][2];
    f32x16_t ofin[DUAL ? QB : 1][2];  // DUAL: normalised text result (+ scaled image result)
    float m_run[QB], l_run[QB];       // m_run: running row max in exp2 units (0 until the first tile sets it)
    // per-lane byte offset of the transposed V read inside a 4-row block: lane i of a 16-lane group supplies
    // row (i>>2), columns 4*(i&3).. of the 16-column block ((lane>>4)&1)
    const int li = lane & 15;
    const int tr_off = (li >> 2) * V_LD + (((lane >> 4) & 1) * 16 + (li & 3) * 4) * 2;
#pragma unroll 1
    for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
    if (DUAL && pass == 1) {          // second key/value set; the LDS ring is free (the tile loop ends on a barrier)
        kb = k2 + (size_t)b * kv_bstride * ldk + head * 64;
        vb = v2 + (size_t)b * kv_bstride * ldv + head * 64;
        Lk = Lk2;
    }
#pragma unroll
    for (int x = 0; x < QB; ++x) {
        m_run[x] = 0.f; l_run[x] = 0.f;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[x][d][r] = 0.f;
    }

    const int nt = (Lk + FA_BKV - 1) / FA_BKV;
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        load_tile(t + 1 < nt ? t + 1 : t);               // unconditional prefetch (keeps staging in registers)
        const char* sk = smem + buf * FA_STAGE;
        const char* sv = sk + FA_KBYTES;

        // S^T[x][jb] = K[jb] (c Q[x])^T - m[x]: the accumulators start at minus the running max of the earlier tiles, so
        // the common case needs no per-score subtraction. Every K fragment is read once for all QB query blocks.
        f32x16_t s[QB][2];
#pragma unroll
        for (int x = 0; x < QB; ++x)
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[x][jb][r] = -m_run[x];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + k_lds_off(jb * 32 + fr, kk * 2 + fh));
#pragma unroll
                for (int x = 0; x < QB; ++x)
                    s[x][jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[x][kk], s[x][jb], 0, 0, 0);
            }
        // mask keys beyond Lk (last tile only)
        if ((t + 1) * FA_BKV > Lk) {
#pragma unroll
            for (int x = 0; x < QB; ++x)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = t * FA_BKV + jb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        if (j >= Lk) s[x][jb][r] = -1e30f;
                    }
        }
        // online softmax: this lane owns one query row per block; its partner (lane ^ 32) holds the other keys
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            float mx = fmaxf(fmaxf(s[x][0][0], s[x][0][1]), s[x][0][2]);
#pragma unroll
            for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s[x][0][r]), s[x][0][r + 1]);
            mx = fmaxf(mx, s[x][0][15]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[x][1][r]), s[x][1][r + 1]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // mx = (tile max) - m_run. The running max moves only when some row's grew (or on the first tile, where
            // it is simply set): then the scores are shifted by the step and the running state rescaled. Otherwise
            // the step is 0 and alpha 1 exactly, so skipping the block is exact.
            if (t == 0 || __any(mx > 0.f)) {
                const float step = (t == 0) ? mx : fmaxf(mx, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-step);
                m_run[x] += step;
                l_run[x] *= alpha;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[x][d][r] *= alpha;
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[x][jb][r] -= step;
            }
            float psum = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(s[x][jb][r]);
                    s[x][jb][r] = p;
                    psum += p;
                }
            l_run[x] += psum;
        }

        // O^T[x][db] += V^T[db][keys] P[x]^T[keys] — every V^T fragment is read once for all QB query blocks
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int jbase = jb * 32 + 16 * ks + 4 * fh;
                bf16x8_t pf[QB];          // P fragments are packed right before use: S registers die here
#pragma unroll
                for (int x = 0; x < QB; ++x) {
                    u32x4_t pw;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(s[x][jb][8 * ks + 2 * e], s[x][jb][8 * ks + 2 * e + 1]);
                    pf[x] = __builtin_bit_cast(bf16x8_t, pw);
                }
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* vp = sv + jbase * V_LD + db * 64 + tr_off;
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp + 8 * V_LD));
                    bf16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
                    for (int x = 0; x < QB; ++x)
                        oacc[x][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[x], oacc[x][db], 0, 0, 0);
                }
            }
        store_tile(buf ^ 1);
        __syncthreads();
    }
    if constexpr (DUAL) {
        // fold this pass into ofin: text result as is, image result times acc_scale
#pragma unroll
        for (int x = 0; x < QB; ++x) {
            const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
            const float wgt = (pass == 0 ? 1.0f : acc_scale) / l_tot;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) ofin[x][d][r] = (pass == 0 ? 0.f : ofin[x][d][r]) + wgt * oacc[x][d][r];
        }
    }
    }   // pass

#pragma unroll
    for (int x = 0; x < QB; ++x) {
        const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
        const float inv = DUAL ? 1.0f : 1.0f / l_tot;
        if constexpr (DUAL) {
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[x][d][r] = ofin[x][d][r];
        }
        if (!DUAL && accumulate) {
            if (qrow[x] < Lq) {
                bf16_t* orow = ob + (size_t)qrow[x] * ldo;
#pragma unroll
                for (int db = 0; db < 2; ++db)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int d = db * 32 + 8 * qd + 4 * fh;
                        uint2* dst = reinterpret_cast<uint2*>(orow + d);
                        const uint2 old = *dst;
                        uint2 pk;
                        pk.x = pack_bf2(__uint_as_float(old.x << 16) + acc_scale * (oacc[x][db][4 * qd + 0] * inv),
                                        __uint_as_float(old.x & 0xffff0000u) + acc_scale * (oacc[x][db][4 * qd + 1] * inv));
                        pk.y = pack_bf2(__uint_as_float(old.y << 16) + acc_scale * (oacc[x][db][4 * qd + 2] * inv),
                                        __uint_as_float(old.y & 0xffff0000u) + acc_scale * (oacc[x][db][4 * qd + 3] * inv));
                        *dst = pk;
                    }
            }
        } else {
            // a row's channels 8 qd .. 8 qd + 7 are split over its two lanes: one v_permlane32_swap per dword between the
            // groups qd and qd + 1 turns eight 8-byte stores per row block into four 16-byte ones (issue-bound store tail; with
            // 93 keys the epilogue is a large part of a cross-attention workgroup)
            bf16_t* orow = ob + (size_t)(qrow[x] < Lq ? qrow[x] : Lq - 1) * ldo;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int qd = 0; qd < 4; qd += 2) {
                    unsigned a0 = pack_bf2(oacc[x][db][4 * qd + 0] * inv, oacc[x][db][4 * qd + 1] * inv);
                    unsigned a1 = pack_bf2(oacc[x][db][4 * qd + 2] * inv, oacc[x][db][4 * qd + 3] * inv);
                    unsigned b0 = pack_bf2(oacc[x][db][4 * qd + 4] * inv, oacc[x][db][4 * qd + 5] * inv);
                    unsigned b1 = pack_bf2(oacc[x][db][4 * qd + 6] * inv, oacc[x][db][4 * qd + 7] * inv);
                    const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    u32x4_t pk = {r0[0], r1[0], r0[1], r1[1]};
                    if (qrow[x] < Lq) *reinterpret_cast<u32x4_t*>(orow + db * 32 + 8 * (qd + fh)) = pk;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Text + image cross-attention with the keys RESIDENT: both key / value sets of a (batch item, head) - 77 text and 16 image
// tokens in DynamiCrafter - are staged into LDS once and the workgroup then walks up to XA_QT_MAX query tiles of 128 rows under them.
// Why: flash_attn_d64_kernel<1, true> stages the sets tile by tile per 128 query rows - three global -> register -> LDS ->
// barrier round trips (text 2 tiles, image 1) whose latency nothing hides (the arithmetic between them is ~30 MFMAs), 24 KB of
// K / V through the L2 -> CU path for 32 KB of q and o - and ran the level-0 launches at 1.9 TB/s of q + o. Here a query tile
// costs its own bytes only, the next tile's q is requested before the current one is computed, and there is no barrier in the
// loop. All keys of a set are visible at once, so the softmax is a plain two-pass one (max, then exp and sum): no running state.
// 64 < Lk <= 96, Lk2 <= 32 (dc_cross_attn_dual_d64 keeps the general kernel for anything else). Same fragment scheme as above:
// S^T = K (cQ)^T so a lane owns a query row; the packed S^T accumulator is the B operand of O^T += V^T P^T; V^T via tr reads.
constexpr int XA_QT_MAX = 8;                                // query tiles of 128 rows per workgroup: chosen per launch (below)
constexpr int XA_K1 = 0, XA_V1 = 96 * 128, XA_K2 = XA_V1 + 96 * V_LD, XA_V2 = XA_K2 + 32 * 128, XA_LDS = XA_V2 + 32 * V_LD;

__global__ __launch_bounds__(256, 2) void cross_attn_resident_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, const bf16_t* __restrict__ k2,
    const bf16_t* __restrict__ v2, bf16_t* __restrict__ o, int ldq, int ldkv, int ldo, int heads, int Lq, int Lk, int Lk2,
    int64_t q_bstride, int64_t kv_bstride, float c /* scale*log2(e) */, float acc_scale, int q_groups, int qt) {
    __shared__ __attribute__((aligned(16))) char smem[XA_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int qg = id % q_groups;
    const int bh = id / q_groups;
    const int head = bh % heads;
    const int b = bh / heads;
    const bf16_t* qb = q + (size_t)b * q_bstride * ldq + head * 64;
    bf16_t* ob = o + (size_t)b * q_bstride * ldo + head * 64;

    // ---- both key / value sets -> LDS, once (rows past a set's length repeat its last row: finite data, masked below)
    {
        const int chunk = tid & 7, srow = tid >> 3;
        const bf16_t* kb1 = k + (size_t)b * kv_bstride * ldkv + head * 64;
        const bf16_t* vb1 = v + (size_t)b * kv_bstride * ldkv + head * 64;
        const bf16_t* kb2 = k2 + (size_t)b * kv_bstride * ldkv + head * 64;
        const bf16_t* vb2 = v2 + (size_t)b * kv_bstride * ldkv + head * 64;
        u32x4_t kr[4], vr[4];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            int j = srow + 32 * i;
            j = j < Lk ? j : Lk - 1;
            kr[i] = *reinterpret_cast<const u32x4_t*>(kb1 + (size_t)j * ldkv + chunk * 8);
            vr[i] = *reinterpret_cast<const u32x4_t*>(vb1 + (size_t)j * ldkv + chunk * 8);
        }
        {
            const int j = srow < Lk2 ? srow : Lk2 - 1;
            kr[3] = *reinterpret_cast<const u32x4_t*>(kb2 + (size_t)j * ldkv + chunk * 8);
            vr[3] = *reinterpret_cast<const u32x4_t*>(vb2 + (size_t)j * ldkv + chunk * 8);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = srow + 32 * i;
            *reinterpret_cast<u32x4_t*>(smem + XA_K1 + k_lds_off(r, chunk)) = kr[i];
            *reinterpret_cast<u32x4_t*>(smem + XA_V1 + r * V_LD + chunk * 16) = vr[i];
        }
        *reinterpret_cast<u32x4_t*>(smem + XA_K2 + k_lds_off(srow, chunk)) = kr[3];
        *reinterpret_cast<u32x4_t*>(smem + XA_V2 + srow * V_LD + chunk * 16) = vr[3];
    }
    const int li = lane & 15;
    const int tr_off = (li >> 2) * V_LD + (((lane >> 4) & 1) * 16 + (li & 3) * 4) * 2;

    const int row0 = qg * (qt * 128) + wave * 32 + fr;              // this lane's query row in tile 0
    auto load_q = [&](int tile, u32x4_t (&raw)[4]) __attribute__((always_inline)) {
        int qr = row0 + tile * 128;
        qr = qr < Lq ? qr : Lq - 1;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) raw[kk] = *reinterpret_cast<const u32x4_t*>(qb + (size_t)qr * ldq + kk * 16 + fh * 8);
    };
    u32x4_t qraw[4];
    load_q(0, qraw);
    __syncthreads();                                                // the only barrier: K / V are in place

    // scores of one set: NB key blocks of 32; softmax in place (s -> p), returns the row sum
    auto scores = [&](const char* sk, int nkeys, auto NB_, f32x16_t (&s)[3], const bf16x8_t (&qf)[4]) __attribute__((always_inline)) -> float {
        constexpr int NB = decltype(NB_)::value;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[jb][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(sk + k_lds_off(jb * 32 + fr, kk * 2 + fh));
                s[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s[jb], 0, 0, 0);
            }
        }
        // keys past the set's length (only the last block can hold any)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = (NB - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
            if (j >= nkeys) s[NB - 1][r] = -1e30f;
        }
        float mx = s[0][0];
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[jb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                     // the partner lane holds the row's other keys
        float sum = 0.f;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(s[jb][r] - mx);
                s[jb][r] = pv;
                sum += pv;
            }
        return sum + __shfl_xor(sum, 32, 64);
    };
    // O^T += V^T P^T over the NB key blocks of a set
    auto pv_mma = [&](const char* sv, auto NB_, const f32x16_t (&s)[3], f32x16_t (&oacc)[2]) __attribute__((always_inline)) {
        constexpr int NB = decltype(NB_)::value;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int jbase = jb * 32 + 16 * ks + 4 * fh;
                u32x4_t pw;
#pragma unroll
                for (int e = 0; e < 4; ++e) pw[e] = pack_bf2(s[jb][8 * ks + 2 * e], s[jb][8 * ks + 2 * e + 1]);
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pw);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* vp = sv + jbase * V_LD + db * 64 + tr_off;
                    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp));
                    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)(vp + 8 * V_LD));
                    bf16x8_t vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
    };

    const int n_tiles = qt;
#pragma unroll 1
    for (int tile = 0; tile < n_tiles; ++tile) {
        const int qrow = row0 + tile * 128;
        if (qg * (qt * 128) + tile * 128 >= Lq) break;              // workgroup-uniform: no rows left in this group
        // Q fragments (B operand), pre-multiplied by scale*log2(e): the scores leave the MFMA in exp2 units
        bf16x8_t qf[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4_t sc;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                sc[e] = pack_bf2(__uint_as_float(qraw[kk][e] << 16) * c, __uint_as_float(qraw[kk][e] & 0xffff0000u) * c);
            qf[kk] = __builtin_bit_cast(bf16x8_t, sc);
        }
        if (tile + 1 < n_tiles) load_q(tile + 1, qraw);             // next tile's rows (clamped): in flight under this tile
        f32x16_t s[3], o1[2], o2[2];
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) { o1[d][r] = 0.f; o2[d][r] = 0.f; }
        const float l1 = scores(smem + XA_K1, Lk, std::integral_constant<int, 3>{}, s, qf);
        pv_mma(smem + XA_V1, std::integral_constant<int, 3>{}, s, o1);
        const float l2 = scores(smem + XA_K2, Lk2, std::integral_constant<int, 1>{}, s, qf);
        pv_mma(smem + XA_V2, std::integral_constant<int, 1>{}, s, o2);
        const float w1 = 1.0f / l1, w2 = acc_scale / l2;
        // store: a row's channels 8 qd .. 8 qd + 7 are split over its two lanes; one v_permlane32_swap per dword between the
        // groups qd and qd + 1 gives each lane 16 contiguous bytes
        bf16_t* orow = ob + (size_t)(qrow < Lq ? qrow : Lq - 1) * ldo;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int qd = 0; qd < 4; qd += 2) {
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = w1 * o1[db][4 * qd + e] + w2 * o2[db][4 * qd + e];
                const unsigned a0 = pack_bf2(f[0], f[1]), a1 = pack_bf2(f[2], f[3]);
                const unsigned b0 = pack_bf2(f[4], f[5]), b1 = pack_bf2(f[6], f[7]);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                u32x4_t pk = {r0[0], r1[0], r0[1], r1[1]};
                if (qrow < Lq) *reinterpret_cast<u32x4_t*>(orow + db * 32 + 8 * (qd + fh)) = pk;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Temporal attention: one wave per (b, p, head). T <= 16 (rows beyond T are masked).
//   S^T[t'][t] = K Q^T            : v_mfma_f32_16x16x32_bf16 x2 (d = 64), fragments straight from global
//   softmax over t' (4 regs x 4 lane groups)
//   O^T[d][t]  = V^T P^T          : v_mfma_f32_16x16x16_bf16 x4, V^T via LDS + ds_read_b64_tr_b16,
//                                    P^T is the accumulator itself (lane: col t, rows t' = 4g + r)
__global__ __launch_bounds__(256) void temporal_attn_d64_kernel(const bf16_t* __restrict__ qkv, int ld,
                                                                bf16_t* __restrict__ o, int ldo, int B, int T, int HW,
                                                                int heads, float c, int total) {
    __shared__ __attribute__((aligned(16))) char smem[4 * 16 * 128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int item = blockIdx.x * 4 + wave;
    const bool active = item < total;
    if (!active) item = total - 1;   // keep the wave in step (EXEC must be full for the transposed LDS read)
    const int head = item % heads;
    const int bp = item / heads;
    const int p = bp % HW;
    const int b = bp / HW;
    const int C = heads * 64;

    const int t16 = lane & 15, g = lane >> 4;
    const int t_c = t16 < T ? t16 : T - 1;
    const size_t row = ((size_t)b * T + t_c) * HW + p;
    const bf16_t* rq = qkv + row * ld + head * 64;
    const bf16_t* rk = rq + C;
    const bf16_t* rv = rq + 2 * C;

    // fragments: A = K[t'=t16][d = 32s + 8g ..], B = Q[t=t16][same d]
    bf16x8_t kf[2], qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kf[s] = *reinterpret_cast<const bf16x8_t*>(rk + 32 * s + 8 * g);
        qf[s] = *reinterpret_cast<const bf16x8_t*>(rq + 32 * s + 8 * g);
    }
    // V tile -> LDS (row-major [t'][64], 128-byte rows): lane writes 32 bytes of row (lane>>2), cols 16*(lane&3)..
    char* sv = smem + wave * (16 * 128);
    {
        const int vr = lane >> 2, vc = (lane & 3) * 16;
        const int vr_c = vr < T ? vr : T - 1;
        const bf16_t* src = qkv + (((size_t)b * T + vr_c) * HW + p) * ld + head * 64 + 2 * C + vc;
        uint4 a = *reinterpret_cast<const uint4*>(src);
        uint4 bq = *reinterpret_cast<const uint4*>(src + 8);
        if (vr >= T) { a = make_uint4(0, 0, 0, 0); bq = a; }
        *reinterpret_cast<uint4*>(sv + vr * 128 + vc * 2) = a;
        *reinterpret_cast<uint4*>(sv + vr * 128 + vc * 2 + 16) = bq;
    }
    (void)rv;
    f32x4_t s4 = {0.f, 0.f, 0.f, 0.f};
    s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[0], qf[0], s4, 0, 0, 0);
    s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[1], qf[1], s4, 0, 0, 0);
    // lane: column t = t16, rows t' = 4g + r
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (4 * g + r >= T) s4[r] = -1e30f;
        mx = fmaxf(mx, s4[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s4[r] = __builtin_amdgcn_exp2f((s4[r] - mx) * c);
        sum += s4[r];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    bf16x4_t pf;
    {
        const uint32_t p01 = pack_bf2(s4[0] * inv, s4[1] * inv), p23 = pack_bf2(s4[2] * inv, s4[3] * inv);
        pf[0] = (short)(p01 & 0xffff); pf[1] = (short)(p01 >> 16);
        pf[2] = (short)(p23 & 0xffff); pf[3] = (short)(p23 >> 16);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    // O^T[n] (rows d = 16n + 4g + r, col t): A = V^T block n: lane i of group g gets V[4g + 0..3][16n + i]
    bf16_t* orow = o + row * ldo + head * 64;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const char* vp = sv + (4 * g + (t16 >> 2)) * 128 + (16 * n + (t16 & 3) * 4) * 2;
        const bf16x4_t vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_bf16x4_t*)vp);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, pf, acc, 0, 0, 0);
        if (active && t16 < T) {
            uint2 pk;
            pk.x = pack_bf2(acc[0], acc[1]);
            pk.y = pack_bf2(acc[2], acc[3]);
            *reinterpret_cast<uint2*>(orow + 16 * n + 4 * g) = pk;
        }
    }
}

}  // namespace

extern "C" int dc_flash_attn_d64(const uint16_t* q, const uint16_t* k, const uint16_t* v, uint16_t* o, int ldq, int ldk,
                                 int ldv, int ldo, int batch, int heads, int Lq, int Lk, int64_t q_bstride,
                                 int64_t kv_bstride, float scale, int accumulate, float acc_scale, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !k || !v || !o) return DC_ERR_ARG;
    if (batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0) return DC_ERR_SHAPE;
    // 16-byte loads of q / k / v (buffer descriptors in flash_pipe.hip) and 16-byte output stores, also in the accumulate epilogue
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 8 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)) return DC_ERR_SHAPE;
    const float c = scale * 1.4426950408889634f;
    // long self-attention: the one-wave-per-SIMD software-pipelined kernel (flash_pipe.hip); DC_FLASH_PIPE=0 keeps the
    // two-waves-per-SIMD kernel below (same-box A/B)
    static const bool use_pipe = [] { const char* e = getenv("DC_FLASH_PIPE"); return !(e && e[0] == '0'); }();
    if (use_pipe && !accumulate && Lq >= 512 && Lk >= 256 && Lk % 64 == 0)
        return dc_flash_pipe_launch(q, k, v, o, ldq, ldk, ldv, ldo, batch, heads, Lq, Lk, q_bstride, kv_bstride, c, stream);
    // two query blocks per wave (256 rows per workgroup) once there are enough rows and keys to pay for it
    const bool wide = Lq >= 512 && Lk >= 256;
    const int rows_wg = wide ? 2 * FA_BQ : FA_BQ;
    const int q_tiles = (Lq + rows_wg - 1) / rows_wg;
    const long long nwg = (long long)q_tiles * heads * batch;
    if (nwg > 0x7fffffffLL) return DC_ERR_SHAPE;
    if (wide)
        hipLaunchKernelGGL((flash_attn_d64_kernel<2, false>), dim3((unsigned)nwg), dim3(256), 0, stream, q, k, v, o, ldq, ldk, ldv,
                           ldo, heads, Lq, Lk, q_bstride, kv_bstride, c, accumulate, acc_scale, q_tiles,
                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
    else
        hipLaunchKernelGGL((flash_attn_d64_kernel<1, false>), dim3((unsigned)nwg), dim3(256), 0, stream, q, k, v, o, ldq, ldk, ldv,
                           ldo, heads, Lq, Lk, q_bstride, kv_bstride, c, accumulate, acc_scale, q_tiles,
                           (const bf16_t*)nullptr, (const bf16_t*)nullptr, 0);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_cross_attn_dual_d64(const uint16_t* q, const uint16_t* k, const uint16_t* v, const uint16_t* k2,
                                      const uint16_t* v2, uint16_t* o, int ldq, int ldkv, int ldo, int batch, int heads,
                                      int Lq, int Lk, int Lk2, int64_t q_bstride, int64_t kv_bstride, float scale,
                                      float scale2, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!q || !k || !v || !k2 || !v2 || !o) return DC_ERR_ARG;
    if (batch <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0 || Lk2 <= 0) return DC_ERR_SHAPE;
    if (ldq % 8 || ldkv % 8 || ldo % 8 || (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)k2 | (uintptr_t)v2 | (uintptr_t)o) & 15)) return DC_ERR_SHAPE;   // 16-byte loads and stores
    const float c = scale * 1.4426950408889634f;
    // the sets of DynamiCrafter (77 text + 16 image tokens) stay resident in LDS under XA_QT query tiles per workgroup;
    // DC_XATTN_RESIDENT=0 keeps the tile-by-tile kernel (same-box A/B)
    static const bool resident = [] { const char* e = getenv("DC_XATTN_RESIDENT"); return !(e && e[0] == '0'); }();
    if (resident && Lk > 64 && Lk <= 96 && Lk2 <= 32 && Lq >= 512) {
        // tiles per workgroup: the value whose grid wastes the least of its last round of 3 workgroups x 256 CUs (ties: the
        // larger, which amortises the staging over more rows)
        const int tiles = (Lq + 127) / 128;
        int qt = 1;
        long long best = -1;
        for (int t = 1; t <= XA_QT_MAX; ++t) {
            const long long wgs = (long long)((tiles + t - 1) / t) * heads * batch;
            const long long cost = ((wgs + 767) / 768) * t;
            if (best < 0 || cost <= best) { best = cost; qt = t; }
        }
        const int q_groups = (tiles + qt - 1) / qt;
        const long long nwg_r = (long long)q_groups * heads * batch;
        if (nwg_r > 0x7fffffffLL) return DC_ERR_SHAPE;
        hipLaunchKernelGGL(cross_attn_resident_kernel, dim3((unsigned)nwg_r), dim3(256), 0, stream, q, k, v, k2, v2, o, ldq, ldkv, ldo,
                           heads, Lq, Lk, Lk2, q_bstride, kv_bstride, c, scale2, q_groups, qt);
        DC_CHECK_LAUNCH();
        return 0;
    }
    const int q_tiles = (Lq + FA_BQ - 1) / FA_BQ;
    const long long nwg = (long long)q_tiles * heads * batch;
    if (nwg > 0x7fffffffLL) return DC_ERR_SHAPE;
    hipLaunchKernelGGL((flash_attn_d64_kernel<1, true>), dim3((unsigned)nwg), dim3(256), 0, stream, q, k, v, o, ldq, ldkv, ldkv,
                       ldo, heads, Lq, Lk, q_bstride, kv_bstride, c, 0, scale2, q_tiles, k2, v2, Lk2);
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_temporal_attn_d64(const uint16_t* qkv, int ld, uint16_t* o, int ldo, int B, int T, int HW, int heads,
                                    float scale, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!qkv || !o) return DC_ERR_ARG;
    if (T <= 0 || T > 16 || B <= 0 || HW <= 0 || heads <= 0 || ld % 8 || ldo % 4) return DC_ERR_SHAPE;
    const long long total = (long long)B * HW * heads;
    if (total > 0x7fffffffLL) return DC_ERR_SHAPE;
    const int grid = (int)((total + 3) / 4);
    const float c = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(temporal_attn_d64_kernel, dim3(grid), dim3(256), 0, stream, qkv, ld, o, ldo, B, T, HW, heads, c,
                       (int)total);
    DC_CHECK_LAUNCH();
    return 0;
}
