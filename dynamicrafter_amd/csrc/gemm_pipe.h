// Shared preamble of the one-wave-per-SIMD 256 x 320 x 64 GEMM / implicit-GEMM conv kernel (gemm_pipe16.h; included by
// gemm_conv_glds.hip inside its namespace: shares GemmSplit, splitk_reduce_kernel and persist_epilogue): ring geometry, the
// bounded counter wait, the unrolling helper and the activation load macro. (The kernel's first form, on
// v_mfma_f32_32x32x16_bf16, lost to the 16x16x32 form on every shape - DESIGN 3.4 - and lives in tools/experimental/gemm_pipe32.h.)
// The design, as it stands in gemm_pipe16.h:
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
constexpr int GP_LDS = GP_CNT + 64;

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
// 10 ms, once per wave) only keeps a future bookkeeping mistake from hanging the GPU. A wave that gave up carries on with
// whatever the ring holds - its tile is garbage - and ORs DC_ERRW_GEMM_PIPE into the library's error word (GemmSplit::err)
// before it leaves: dc_error_word_read / ops.check_error_word make that loud on the host.
#define GP_SPIN(cond, reread)                                              \
    do {                                                                   \
        int spins__ = 0;                                                   \
        while (!gave_up && (cond)) { reread; if (++spins__ > 200000) gave_up = 1; } \
    } while (0)

template <int V> using gp_ic = std::integral_constant<int, V>;
template <int... G, class F>
__device__ __forceinline__ void gp_for(std::integer_sequence<int, G...>, F&& f) { (f(gp_ic<G>{}), ...); }

#define GP_LOAD_A_(dst, vo, rs, so, IMM) \
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:" IMM : "=&v"(dst) : "v"(vo), "s"(rs), "s"(so))
