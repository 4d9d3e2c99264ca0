// GroupNorm(+SiLU) and LayerNorm over channels-last bf16 rows; fp32 statistics. HBM-bound kernels:
// 16-byte loads/stores per lane, each row read twice (stats pass + apply pass) and written once.
//
// GroupNorm is three launches: partial moments per (instance, row-chunk, group) -> finalize (mean, rstd) per
// (instance, group) -> normalise*affine(+SiLU). All reductions have a fixed order: results are bitwise
// reproducible run to run. The moments are (mean, M2 = sum of squared deviations), accumulated per thread around its
// first sample and merged with Chan's parallel update - not E[x^2] - mean^2, whose cancellation error ~1e-7 mean^2/var
// reaches percent level on channels with a large DC offset (up to 2.4 M elements per group at the AE's 576x1024 levels).
//   reference: GroupNormSpecific lvdm/basics.py:76-87 (eps 1e-5, fp32), nn.GroupNorm(32, C) in
//   TemporalConvBlock openaimodel3d.py:256-265 (5-D: statistics span T*H*W), transformer norms
//   attention.py:265,331 (eps 1e-6), AE Normalize ae_modules.py:15-16 (eps 1e-6) + swish :10-12.
#include "dc_common.h"
#include "dcrafter_hip.h"
#include <stdlib.h>

namespace {

struct GnGeom { int chunks; int rows_per_chunk; };

__host__ __device__ inline GnGeom gn_geom(int n_inst, int rows_per_inst) {
    int chunks = (2048 + n_inst - 1) / n_inst;
    int rpc = (rows_per_inst + chunks - 1) / chunks;
    if (rpc < 64) rpc = 64;
    GnGeom g;
    g.rows_per_chunk = rpc;
    g.chunks = (rows_per_inst + rpc - 1) / rpc;
    return g;
}

// grid (chunks, n_inst), block = vecs * rpp threads (vecs = C/8)
__global__ void gn_partial_kernel(const bf16_t* __restrict__ x, int ldx, int C, int groups, int rows_per_inst,
                                  int rows_per_chunk, int chunks, int vecs, int rpp, float2* __restrict__ partial) {
    extern __shared__ float sm[];   // [2][rpp][C]
    const int inst = blockIdx.y, chunk = blockIdx.x;
    const int v = threadIdx.x % vecs;
    const int ro = threadIdx.x / vecs;
    const int r_begin = chunk * rows_per_chunk;
    int r_end = r_begin + rows_per_chunk;
    if (r_end > rows_per_inst) r_end = rows_per_inst;
    float s[8], ss[8], K[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; K[e] = 0.f; }
    const bf16_t* base = x + (size_t)inst * rows_per_inst * ldx + v * 8;
    int r = r_begin + ro;
    const int n_t = r < r_end ? (r_end - r + rpp - 1) / rpp : 0;      // samples per channel of this thread
    bool have_k = false;        // shift = the thread's first sample of each channel, taken from the first batch of loads
                                // (a separate load ahead of the loop costs a serial HBM round trip per chunk)
    // eight independent 16-byte loads in flight per lane (HBM latency, not issue rate, bounds this pass)
    for (; r + 7 * rpp < r_end; r += 8 * rpp) {
        uint4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const uint4*>(base + (size_t)(r + u * rpp) * ldx);
        if (!have_k) { unpack_bf8(raw[0], K); have_k = true; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float f[8];
            unpack_bf8(raw[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = f[e] - K[e]; s[e] += d; ss[e] += d * d; }
        }
    }
    for (; r < r_end; r += rpp) {
        const uint4 raw = *reinterpret_cast<const uint4*>(base + (size_t)r * ldx);
        float f[8];
        unpack_bf8(raw, f);
        if (!have_k) { unpack_bf8(raw, K); have_k = true; }
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[e] - K[e]; s[e] += d; ss[e] += d * d; }
    }
    // per (row slot, channel): mean and M2 of the thread's n_t samples
    float* sm_s = sm;                 // means
    float* sm_q = sm + rpp * C;       // M2
    const float inv_n = n_t > 0 ? 1.0f / (float)n_t : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sm_s[ro * C + v * 8 + e] = K[e] + s[e] * inv_n;
        sm_q[ro * C + v * 8 + e] = ss[e] - s[e] * s[e] * inv_n;
    }
    __syncthreads();
    if ((int)threadIdx.x < groups) {
        const int cpg = C / groups;
        const int g = threadIdx.x;
        // Chan merge in a fixed order: row slots outer, channels inner
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int rr = 0; rr < rpp; ++rr) {
            const int first = r_begin + rr;
            const float nb = first < r_end ? (float)((r_end - first + rpp - 1) / rpp) : 0.f;
            if (nb == 0.f) continue;
            for (int c = 0; c < cpg; ++c) {
                const float mb = sm_s[rr * C + g * cpg + c], qb = sm_q[rr * C + g * cpg + c];
                const float nn = n + nb;
                const float w = nb * __builtin_amdgcn_rcpf(nn);   // 1-ulp reciprocal: a weight, not a sum (an IEEE divide
                const float delta = mb - mean;                    // here is ~10 instructions on a 60-step serial chain)
                mean += delta * w;
                m2 += qb + delta * delta * (n * w);
                n = nn;
            }
        }
        partial[((size_t)inst * chunks + chunk) * groups + g] = make_float2(mean, m2);
    }
}

// grid (groups, n_inst), one wave each: reduce the chunk partials of one (instance, group) in a fixed order (lane-strided
// sums, then a fixed shuffle tree) -> (mean, rstd). One workgroup per instance took 12 us on the 5-D norms (2 instances
// x 1024 chunks); spread over groups it is launch-latency-bound.
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float2* __restrict__ partial, int chunks, int groups,
                                                         int rows_per_inst, int rows_per_chunk, int cpg, float eps,
                                                         float2* __restrict__ stats) {
    const int g = blockIdx.x, inst = blockIdx.y;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int c = threadIdx.x; c < chunks; c += 64) {
        const float2 p = partial[((size_t)inst * chunks + c) * groups + g];
        int rows = rows_per_inst - c * rows_per_chunk;
        if (rows > rows_per_chunk) rows = rows_per_chunk;
        const float nb = (float)rows * (float)cpg;
        const float nn = n + nb;
        const float w = nb * __builtin_amdgcn_rcpf(nn);
        const float delta = p.x - mean;
        mean += delta * w;
        m2 += p.y + delta * delta * (n * w);
        n = nn;
    }
    // fixed shuffle tree of Chan merges (lanes without chunks carry n = 0)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
        const float nn = n + nb;
        if (nn > 0.f) {                                  // only lane 0's merge chain reaches the result; its order is fixed
            const float w = nb * __builtin_amdgcn_rcpf(nn);
            const float delta = mb - mean;
            mean += delta * w;
            m2 += qb + delta * delta * (n * w);
            n = nn;
        }
    }
    if (threadIdx.x == 0) {
        float var = m2 / n;
        if (var < 0.f) var = 0.f;
        stats[(size_t)inst * groups + g] = make_float2(mean, rsqrtf(var + eps));
    }
}

// FIN: the (mean, rstd) of the instance's groups are reduced from the chunk partials HERE instead of by gn_finalize_kernel - one
// launch (5.6 us + a kernel boundary) less per norm. Round 4 first tried this with gn_finalize_kernel's own layout (a wave per
// group, lanes = chunks): 8-11 latency chains back to back per wave, +7 us per workgroup, slower than the launch. This layout puts
// EIGHT lanes on a group (thread t: group t / 8, chunks t % 8, + 8, ...): 32 groups in one pass of 256 threads, each lane a
// serial chain of chunks / 8 Chan merges and three xor-shuffle merges - one L2 round trip and ~11 dependent merges per
// workgroup. Fixed order: deterministic, but not the bits of gn_finalize_kernel (the two never feed the same consumer).
template <bool FIN>
__global__ void gn_apply_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int C, int groups,
                                int rows_per_inst, int rows_per_chunk, int vecs, int rpp, int silu,
                                const float2* __restrict__ stats, int chunks, float eps) {
    const int inst = blockIdx.y, chunk = blockIdx.x;
    __shared__ float2 st_s[64];
    if constexpr (FIN) {
        const int t = threadIdx.x, sub = t & 7;
        const int gpp = (int)blockDim.x >> 3;                    // groups per pass (blockDim.x % 8 == 0: vecs * rpp, vecs = C / 8, C % 64 == 0 checked by the host)
        const float cpg = (float)(C / groups);
        for (int g0 = 0; g0 < groups; g0 += gpp) {
            const int g = g0 + (t >> 3);
            float n = 0.f, mean = 0.f, m2 = 0.f;
            if (g < groups) {
                for (int c = sub; c < chunks; c += 8) {
                    const float2 p = stats[((size_t)inst * chunks + c) * groups + g];
                    int rows = rows_per_inst - c * rows_per_chunk;
                    if (rows > rows_per_chunk) rows = rows_per_chunk;
                    const float nb = (float)rows * cpg;
                    const float nn = n + nb;
                    const float w = nb * __builtin_amdgcn_rcpf(nn);
                    const float delta = p.x - mean;
                    mean += delta * w;
                    m2 += p.y + delta * delta * (n * w);
                    n = nn;
                }
            }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
                const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mean, o, 64), qb = __shfl_xor(m2, o, 64);
                const float nn = n + nb;
                if (nn > 0.f) {
                    // symmetric form: both partners of a pair compute the same merged (n, mean, m2) bit for bit
                    const float lo_n = (sub & o) ? nb : n, lo_m = (sub & o) ? mb : mean, lo_q = (sub & o) ? qb : m2;
                    const float hi_n = (sub & o) ? n : nb, hi_m = (sub & o) ? mean : mb, hi_q = (sub & o) ? m2 : qb;
                    const float w = hi_n * __builtin_amdgcn_rcpf(nn);
                    const float delta = hi_m - lo_m;
                    mean = lo_m + delta * w;
                    m2 = lo_q + hi_q + delta * delta * (lo_n * w);
                    n = nn;
                }
            }
            if (g < groups && sub == 0) {
                float var = m2 / n;
                if (var < 0.f) var = 0.f;
                st_s[g] = make_float2(mean, rsqrtf(var + eps));
            }
        }
        __syncthreads();
    }
    const int v = threadIdx.x % vecs;
    const int ro = threadIdx.x / vecs;
    const int r_begin = chunk * rows_per_chunk;
    int r_end = r_begin + rows_per_chunk;
    if (r_end > rows_per_inst) r_end = rows_per_inst;
    const int cpg = C / groups;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = v * 8 + e;
        const float2 st = FIN ? st_s[c / cpg] : stats[(size_t)inst * groups + c / cpg];
        const float g = gamma[c] * st.y;
        sc[e] = g;
        sh[e] = beta[c] - st.x * g;
    }
    const bf16_t* xb = x + (size_t)inst * rows_per_inst * ldx + v * 8;
    bf16_t* yb = y + (size_t)inst * rows_per_inst * ldy + v * 8;
    int r = r_begin + ro;
    for (; r + 7 * rpp < r_end; r += 8 * rpp) {
        uint4 raw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) raw[u] = *reinterpret_cast<const uint4*>(xb + (size_t)(r + u * rpp) * ldx);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float f[8];
            unpack_bf8(raw[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float t = f[e] * sc[e] + sh[e];
                if (silu) t = silu_f(t);
                f[e] = t;
            }
            *reinterpret_cast<uint4*>(yb + (size_t)(r + u * rpp) * ldy) = pack_bf8(f);
        }
    }
    for (; r < r_end; r += rpp) {
        const uint4 raw = *reinterpret_cast<const uint4*>(xb + (size_t)r * ldx);
        float f[8];
        unpack_bf8(raw, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = f[e] * sc[e] + sh[e];
            if (silu) t = silu_f(t);
            f[e] = t;
        }
        *reinterpret_cast<uint4*>(yb + (size_t)r * ldy) = pack_bf8(f);
    }
}

// one wave per row, MAXV 16-byte vectors per lane
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, int ldx, bf16_t* __restrict__ y,
                                                        int ldy, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int vecs = C >> 3;
    const float invC = 1.0f / (float)C;
    for (int row0 = (blockIdx.x * 4 + wave) * 2; row0 < rows; row0 += gridDim.x * 8) {
        // two rows per wave in flight (independent load chains)
        float f[2][MAXV][8];
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row0 + u < rows ? row0 + u : rows - 1;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int v = lane + 64 * j;
                if (v < vecs) {
                    const uint4 raw = *reinterpret_cast<const uint4*>(x + (size_t)row * ldx + v * 8);
                    unpack_bf8(raw, f[u][j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[u] += f[u][j][e];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = row0 + u;
            const float mean = wave_sum(s[u]) * invC;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < MAXV; ++j) {
                const int v = lane + 64 * j;
                if (v < vecs) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = f[u][j][e] - mean; q += d * d; }
                }
            }
            const float rstd = rsqrtf(wave_sum(q) * invC + eps);
            if (row < rows) {
#pragma unroll
                for (int j = 0; j < MAXV; ++j) {
                    const int v = lane + 64 * j;
                    if (v < vecs) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int c = v * 8 + e;
                            o[e] = (f[u][j][e] - mean) * rstd * gamma[c] + beta[c];
                        }
                        *reinterpret_cast<uint4*>(y + (size_t)row * ldy + v * 8) = pack_bf8(o);
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int64_t dc_groupnorm_workspace_bytes(int n_inst, int groups, int rows_per_inst) {
    const GnGeom g = gn_geom(n_inst, rows_per_inst);
    return (int64_t)sizeof(float2) * ((int64_t)n_inst * g.chunks * groups + (int64_t)n_inst * groups) + 256;
}

extern "C" int dc_groupnorm(const uint16_t* x, int ldx, uint16_t* y, int ldy, const float* gamma, const float* beta,
                            int C, int groups, int n_inst, int rows_per_inst, float eps, int silu,
                            float* workspace, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || !gamma || !beta || !workspace) return DC_ERR_ARG;
    if (C % 8 != 0 || groups <= 0 || groups > 64 || C % groups != 0 || ldx % 8 != 0 || ldy % 8 != 0) return DC_ERR_SHAPE;
    if (n_inst <= 0 || rows_per_inst <= 0) return DC_ERR_SHAPE;
    const int vecs = C / 8;
    if (vecs > 1024) return DC_ERR_SHAPE;
    int rpp = 256 / vecs;
    if (rpp < 1) rpp = 1;
    const int threads = vecs * rpp;
    const GnGeom g = gn_geom(n_inst, rows_per_inst);
    float2* partial = reinterpret_cast<float2*>(workspace);
    float2* stats = partial + (size_t)n_inst * g.chunks * groups;
    const size_t lds = (size_t)2 * rpp * C * sizeof(float);
    if (lds > 64 * 1024) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(g.chunks, n_inst), dim3(threads), lds, stream, x, ldx, C, groups,
                       rows_per_inst, g.rows_per_chunk, g.chunks, vecs, rpp, partial);
    DC_CHECK_LAUNCH();
    // few chunks per instance (the 4-D norms of the UNet: 32 frames x 64 chunks): the apply pass reduces the partials itself
    // (gn_apply_kernel<true>); DC_GN_FUSE_FINALIZE=0 keeps the separate launch (same-box A/B)
    static const bool fuse_fin = [] { const char* e = getenv("DC_GN_FUSE_FINALIZE"); return !(e && e[0] == '0'); }();
    if (fuse_fin && g.chunks <= 64 && threads % 8 == 0) {
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(g.chunks, n_inst), dim3(threads), 0, stream, x, ldx, y, ldy, gamma, beta,
                           C, groups, rows_per_inst, g.rows_per_chunk, vecs, rpp, silu, (const float2*)partial, g.chunks, eps);
        DC_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n_inst), dim3(64), 0, stream, partial, g.chunks, groups, rows_per_inst,
                       g.rows_per_chunk, C / groups, eps, stats);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(g.chunks, n_inst), dim3(threads), 0, stream, x, ldx, y, ldy, gamma, beta,
                       C, groups, rows_per_inst, g.rows_per_chunk, vecs, rpp, silu, (const float2*)stats, g.chunks, eps);
    DC_CHECK_LAUNCH();
    return 0;
}

// statistics only: (mean, rstd) per (instance, group) -> stats_out [n_inst][groups][2] fp32, for consumers that apply the
// normalisation themselves (dc_gn_linear320 folds it into the Linear that follows)
extern "C" int dc_groupnorm_stats(const uint16_t* x, int ldx, int C, int groups, int n_inst, int rows_per_inst, float eps,
                                  float* workspace, float* stats_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !workspace || !stats_out) return DC_ERR_ARG;
    if (C % 8 != 0 || groups <= 0 || groups > 64 || C % groups != 0 || ldx % 8 != 0) return DC_ERR_SHAPE;
    if (n_inst <= 0 || rows_per_inst <= 0) return DC_ERR_SHAPE;
    const int vecs = C / 8;
    if (vecs > 1024) return DC_ERR_SHAPE;
    int rpp = 256 / vecs;
    if (rpp < 1) rpp = 1;
    const int threads = vecs * rpp;
    const GnGeom g = gn_geom(n_inst, rows_per_inst);
    float2* partial = reinterpret_cast<float2*>(workspace);
    const size_t lds = (size_t)2 * rpp * C * sizeof(float);
    if (lds > 64 * 1024) return DC_ERR_SHAPE;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(g.chunks, n_inst), dim3(threads), lds, stream, x, ldx, C, groups,
                       rows_per_inst, g.rows_per_chunk, g.chunks, vecs, rpp, partial);
    DC_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n_inst), dim3(64), 0, stream, partial, g.chunks, groups, rows_per_inst,
                       g.rows_per_chunk, C / groups, eps, reinterpret_cast<float2*>(stats_out));
    DC_CHECK_LAUNCH();
    return 0;
}

extern "C" int dc_layernorm(const uint16_t* x, int ldx, uint16_t* y, int ldy, const float* gamma, const float* beta,
                            int rows, int C, float eps, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !y || !gamma || !beta) return DC_ERR_ARG;
    if (C % 8 != 0 || C > 2048 || ldx % 8 != 0 || ldy % 8 != 0 || rows <= 0) return DC_ERR_SHAPE;
    int grid = (rows + 7) / 8;
    if (grid > 8192) grid = 8192;
    const int vecs = C / 8;
    if (vecs <= 64)
        hipLaunchKernelGGL(layernorm_kernel<1>, dim3(grid), dim3(256), 0, stream, x, ldx, y, ldy, gamma, beta, rows, C, eps);
    else if (vecs <= 128)
        hipLaunchKernelGGL(layernorm_kernel<2>, dim3(grid), dim3(256), 0, stream, x, ldx, y, ldy, gamma, beta, rows, C, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3(grid), dim3(256), 0, stream, x, ldx, y, ldy, gamma, beta, rows, C, eps);
    DC_CHECK_LAUNCH();
    return 0;
}
