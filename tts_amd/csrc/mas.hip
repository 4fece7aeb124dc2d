// Monotonic alignment search (MAS) for gfx950.
//
// Replaces TTS/tts/utils/monotonic_align/core.pyx:11-47 (maximum_path_each / maximum_path_c) and
// the host glue of TTS/tts/utils/helpers.py:178-194 (value*mask, lengths from the mask, D2H/H2D).
//
// Design (one workgroup per batch item; integer/fp32 add-compare work, dependency-latency bound — deliberately NOT
// reshaped into a GEMM):
//   * forward DP (T_x <= 512: mas_forward_mw_kernel) — lane l of DP wave r owns row x = 64 r + l; the R = ceil(T_x/64) row
//     groups sweep the columns as a skewed pipeline of R waves.  The previous column lives in one register per lane, the
//     x-1 neighbour comes from a DPP wave shift inside a group and from an 8-byte {value, column} LDS ring slot between
//     groups.  The other waves of the block stream [T_x x 32]-column tiles HBM -> LDS with coalesced 128-byte row segments,
//     transposing into a [y][x] LDS image (row pitch 64R+1 => conflict-free both ways), double-buffered against the DP
//     waves, and stream finished tiles back.  Each cell is ONE fp32 add on the same operands as the reference, so values
//     and therefore the path are bit-exact for any traversal order.  (T_x > 512: mas_forward_kernel, one DP wave holding
//     all row groups in registers.)
//   * the backtrack needs only `value[x,y-1] < value[x-1,y-1]`; the DP waves get that predicate for free (they already
//     hold both operands) and store it as ballot bit-planes dirs[b][y][r] (8 bytes per 64 rows) instead of re-reading
//     4-byte values.
//   * backtrack — per 64-column chunk every lane extracts a 64-row window of its column's bit-plane around the current
//     index, then wave 0 walks the chunk with scalar readlane ops while the other waves write the previous chunk's
//     256-byte path row segments (zeros included).
// Round 6 (profiles/r06_mas_column_step_ab.txt; [32,257,770]: 207 -> 106 us, forward 142 -> 76, backtrack 59 -> 26): a lone
// wave issues roughly one instruction per 8-10 cycles, so both serial chains were cut to the instructions of the recurrence —
// mas_forward_mw2_kernel<R, NEED_COPY> (~23 instructions per column and DP wave, no taken branch on the straight path) and the
// WALK2 backtrack (6 per column).  The older kernels stay selectable (TTSAMD_MAS_MW=1, TTSAMD_MAS_BT=1) and tested.
#include "common.h"

#include <cstdlib>
#include <type_traits>
#include <utility>

namespace ttsamd {

constexpr int kMasThreads = 512;          // 8 waves: 1 DP + 7 tile movers
constexpr int kMasMovers = kMasThreads / kWave - 1;

template <int YT> struct MasTile {
    static constexpr int kShift = (YT == 32) ? 5 : (YT == 16) ? 4 : 3;
    static constexpr int kRowsPerInstr = kWave / YT;
};

// ---- tile movers -----------------------------------------------------------------------------
template <int YT, int kBatch = 16>
__device__ __forceinline__ void mas_stage_tile(float *lds, const float *in,
                                               const float *__restrict__ mask, long base, int Tx,
                                               int Ty, int XP, int tile, int mover, int lane, int nmov = kMasMovers)
{
    const int yl = lane & (YT - 1);
    const int xs = lane >> MasTile<YT>::kShift;
    const int y = tile * YT + yl;
    const int ngroups = (Tx + MasTile<YT>::kRowsPerInstr - 1) / MasTile<YT>::kRowsPerInstr;
    // batches of 16 row groups: all loads of a batch are issued before the first LDS write (clamped addresses + select,
    // no branch between them), so a tile costs ~one HBM round trip per batch instead of one per row group
    // a NULL mask reads `in` a second time (L1 hits) and multiplies by 1: one straight-line body, no branch per element
    // (hipcc otherwise branches around every mask load and waits for each)
    const float *mk = mask ? mask : in;
    const float one = mask ? 0.f : 1.f;
    for (int g0 = mover; g0 < ngroups; g0 += nmov * kBatch) {
        float v[kBatch], m[kBatch];
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int x = (g0 + i * nmov) * MasTile<YT>::kRowsPerInstr + xs;
            const bool ok = (x < Tx) && (y < Ty);
            const long off = ok ? base + (long)x * Ty + y : base;
            v[i] = in[off];
            m[i] = mk[off];
        }
#pragma unroll
        for (int i = 0; i < kBatch; ++i) {
            const int x = (g0 + i * nmov) * MasTile<YT>::kRowsPerInstr + xs;
            if ((x < Tx) && (y < Ty)) lds[yl * XP + x] = mask ? v[i] * m[i] : v[i];
        }
    }
}

template <int YT>
__device__ __forceinline__ void mas_writeback_tile(const float *lds, float *out,
                                                   long base, int Tx, int Ty, int XP, int tile,
                                                   int mover, int lane, int nmov = kMasMovers)
{
    const int yl = lane & (YT - 1);
    const int xs = lane >> MasTile<YT>::kShift;
    const int y = tile * YT + yl;
    const int ngroups = (Tx + MasTile<YT>::kRowsPerInstr - 1) / MasTile<YT>::kRowsPerInstr;
#pragma unroll 8
    for (int g = mover; g < ngroups; g += nmov) {
        const int x = g * MasTile<YT>::kRowsPerInstr + xs;
        if (x < Tx && y < Ty) out[base + (long)x * Ty + y] = lds[yl * XP + x];
    }
}

// ---- forward DP ------------------------------------------------------------------------------
template <int RMAX, int YT, bool EXACT>
__global__ __launch_bounds__(kMasThreads) void mas_forward_kernel(
    const float *in_values /* may alias dp_values (in-place mirror) */,
    const float *__restrict__ mask, float *dp_values, unsigned long long *__restrict__ dirs,
    const int *__restrict__ t_xs, const int *__restrict__ t_ys, int Tx, int Ty, int R_rt, float neg)
{
    // EXACT: the number of 64-row groups is the template constant (T_x <= 512 gets its own instantiation), so every
    // per-group guard folds at compile time; with a runtime R hipcc turned the column step into ~600 instructions of
    // branches and register shuffling (2100 cycles per column).
    const int R = EXACT ? RMAX : R_rt;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int XP = 64 * R + 1;
    float *buf0 = smem;
    float *buf1 = smem + YT * XP;
    const int t_x = min(t_xs[b], Tx);
    const int t_y = min(t_ys[b], Ty);
    const long base = (long)b * Tx * Ty;
    const int nt = (Ty + YT - 1) / YT;
    const bool need_copy = (dp_values != nullptr);
    // With no dp_values output only the columns the DP touches have to be staged.
    const int nt_work = need_copy ? nt : ((t_x > 0 && t_y > 0) ? (min(t_y, Ty) + YT - 1) / YT : 0);

    float prev[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) prev[r] = 0.f;

    if (wave > 0 && nt_work > 0)
        mas_stage_tile<YT>(buf0, in_values, mask, base, Tx, Ty, XP, 0, wave - 1, lane);
    __syncthreads();

    for (int t = 0; t < nt_work; ++t) {
        float *cur = (t & 1) ? buf1 : buf0;
        float *oth = (t & 1) ? buf0 : buf1;
        if (wave == 0) {
            const int y_end = min(t_y, min(Ty, (t + 1) * YT));
            if (t_x > 0) {
                // column values are prefetched one column ahead of the DP step that consumes them (LDS latency off the
                // serial chain); the x-1 neighbour comes from a DPP wave shift (`v_mov_b32_dpp wave_shr:1`, one VALU op)
                // instead of `__shfl_up` (ds_bpermute: an LDS round trip per row group and column), the carry between
                // 64-row groups from a scalar readlane.
                float cnext[RMAX];
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const int x = r * 64 + lane;
                    cnext[r] = (r < R && x < Tx && t * YT < y_end) ? cur[x] : 0.f;
                }
                for (int y = t * YT; y < y_end; ++y) {
                    const int yl = y - t * YT;
                    const int x_lo = max(0, t_x + y - t_y);
                    const int x_hi = min(t_x, y + 1);
                    float cval[RMAX];
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) {
                        cval[r] = cnext[r];
                        const int x = r * 64 + lane;
                        if (r < R && y + 1 < y_end) cnext[r] = (x < Tx) ? cur[(yl + 1) * XP + x] : 0.f;
                    }
                    float up[RMAX];
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) {
                        if (r < R) {
                            float u = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                                                    0, __builtin_bit_cast(int, prev[r]), 0x138, 0xf, 0xf, false));
                            if (r > 0) {
                                const float carry = __builtin_bit_cast(
                                    float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, prev[r - 1]), 63));
                                if (lane == 0) u = carry;
                            }
                            up[r] = u;
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RMAX; ++r) {
                        if (r < R) {
                            const int x = r * 64 + lane;
                            // direction bit-plane of column y-1: value[x,y-1] < value[x-1,y-1]
                            if (y > 0) {
                                const unsigned long long bits = __ballot(x >= 1 && prev[r] < up[r]);
                                if (lane == 0) dirs[((long)b * Ty + (y - 1)) * R + r] = bits;
                            }
                            const float c = cval[r];
                            const float v_cur = (x == y) ? neg : prev[r];
                            const float v_prev = (x == 0) ? (y == 0 ? 0.f : neg) : up[r];
                            const float nv = fmaxf(v_cur, v_prev) + c;
                            const bool inb = (x >= x_lo) && (x < x_hi);
                            const float keep = inb ? nv : c;
                            prev[r] = keep;
                            if (need_copy && inb) cur[yl * XP + x] = keep;
                        }
                    }
                }
            }
        } else {
            if (need_copy && t > 0)
                mas_writeback_tile<YT>(oth, dp_values, base, Tx, Ty, XP, t - 1, wave - 1, lane);
            if (t + 1 < nt_work)
                mas_stage_tile<YT>(oth, in_values, mask, base, Tx, Ty, XP, t + 1, wave - 1, lane);
        }
        __syncthreads();
    }
    if (need_copy && nt_work > 0 && wave > 0) {
        const float *last = ((nt_work - 1) & 1) ? buf1 : buf0;
        mas_writeback_tile<YT>(last, dp_values, base, Tx, Ty, XP, nt_work - 1, wave - 1, lane);
    }
}

// ---- forward DP, one wave per 64-row group (T_x <= 512) ------------------------------------------
// The column sweep is serial in y, but row group r of column y only needs row 64r-1 of column y-1 from the group
// below it: the R row groups run as a skewed pipeline of R waves (wave r works on column y while wave r-1 is already
// past it), each holding ONE previous-column register.  The boundary value travels through a 64-slot LDS ring per
// wave: the producer's lane 63 publishes {value, column tag} as ONE 8-byte ds_write, the consumer reads the slot with
// one 8-byte ds_read — requested right after it finished its own previous column, so the LDS round trip is off the
// serial chain whenever the producer is ahead — and re-reads only while the tag is not the column it needs.  Consumers
// never block producers, and every wave meets the tile movers at the tile barriers (32 columns), which also bounds
// the skew below the ring size (a slot is reused 64 columns later).  Same single fp32 add per cell as the reference
// => bit-exact, like the one-wave kernel.
// 12 waves per item: R DP waves + (12 - R) tile movers.  Measured on [32,257,770] (R = 5), forward kernel alone: one DP wave
// 520 us -> this kernel 250 us (round 2; with the DP switched off the movers need 70 us, with the movers off the DP needs the
// same 250 us: the DP waves are instruction-issue bound, ~75 instructions and a dozen branches per column and wave) -> 150 us
// with the straight-line column step of round 3 (~40 instructions; whole maximum_path 0.282 -> 0.207 ms, [256,257,770] 0.313 ->
// 0.245 ms = 1.1e11 cells/s; raising the DP waves' issue priority with s_setprio changed nothing).
constexpr int kMasMwWaves = 12;
constexpr int kMasRing = 64;

__global__ __launch_bounds__(64 * kMasMwWaves) void mas_forward_mw_kernel(
    const float *in_values, const float *__restrict__ mask, float *dp_values, unsigned long long *__restrict__ dirs,
    const int *__restrict__ t_xs, const int *__restrict__ t_ys, int Tx, int Ty, int R, float neg)
{
    constexpr int YT = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int XP = 64 * R + 1;
    float *buf0 = smem;
    float *buf1 = smem + YT * XP;
    // [R][kMasRing] {value, column}: accessed with relaxed workgroup-scope atomics = plain ds_read_b64 / ds_write_b64 that
    // the compiler may neither cache nor reorder against each other (a `volatile` pointer here is demoted to FLAT accesses
    // with system-scope cache bits and a full wait after each one)
    unsigned long long *ring = reinterpret_cast<unsigned long long *>(smem + 2 * YT * XP);
#define MAS_RING_LD(i) __hip_atomic_load(ring + (i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define MAS_RING_ST(i, v) __hip_atomic_store(ring + (i), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
    const int t_x = min(t_xs[b], Tx);
    const int t_y = min(t_ys[b], Ty);
    const long base = (long)b * Tx * Ty;
    const int nt = (Ty + YT - 1) / YT;
    const bool need_copy = (dp_values != nullptr);
    const int nt_work = need_copy ? nt : ((t_x > 0 && t_y > 0) ? (min(t_y, Ty) + YT - 1) / YT : 0);
    const bool dp_wave = wave < R;
    const int mover = wave - R;
    const int nmov = kMasMwWaves - R;

    for (int i = threadIdx.x; i < R * kMasRing; i += blockDim.x) MAS_RING_ST(i, ~0ull);     // tag -1: nothing published
    for (int i = threadIdx.x; i < 2 * YT * XP; i += blockDim.x) {   // padding rows x >= Tx (no mover writes them) read as 0
        const int xx = i % XP;
        if (xx >= Tx) smem[i] = 0.f;
    }
    if (!dp_wave && nt_work > 0)
        mas_stage_tile<YT, 16>(buf0, in_values, mask, base, Tx, Ty, XP, 0, mover, lane, nmov);
    __syncthreads();

    const int r = wave;
    const int x = r * 64 + lane;
    float prev = 0.f;      // this row's value in the previous column
    // Round 3: the column step as straight-line code.  The round-2 loop spent ~75 instructions and a dozen branches per
    // column (y == 0 / r > 0 / lane == 0 / in-band cases re-derived every column); here column 0 is peeled (it is just
    // prev = value), the first row group is its own instantiation (no ring, row 0's "no upper neighbour"), band membership is
    // two compares on d = y - x against the constant t_y - t_x (x_lo <= x < x_hi  <=>  0 <= y - x <= t_y - t_x for x < t_x),
    // the next column's value and ring slot are requested unconditionally (one row past the tile stays inside the LDS
    // allocation), in-place values are written for every lane (outside the band keep == value), and max() is the bare
    // instruction (fmaxf first canonicalises both operands).  Same fp32 add per cell => still bit-exact.
    const int band = t_y - t_x;
    const int xe = (x < t_x) ? x : 0x3fffffff;                 // rows beyond t_x never enter the band
    const bool publish = (r + 1 < R);
    auto vmax = [](float a_, float b_) {
        float o;
        asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a_), "v"(b_));
        return o;
    };
    auto dp_tile = [&](auto first_tag, float *cur, int t) {
        constexpr bool kFirst = decltype(first_tag)::value;
        int y = t * YT;
        const int y_end = min(t_y, min(Ty, (t + 1) * YT));
        if (y >= y_end) return;
        float *cp = cur + x;                                    // this lane's row in the tile, one column = XP floats
        float cnext = cp[0];
        if (t == 0) {                                           // column 0: value[x,0] = value (x = 0: max(neg, 0) + value)
            prev = cnext;
            cnext = cp[XP];
            if (publish && lane == 63) MAS_RING_ST(r * kMasRing, (unsigned long long)__builtin_bit_cast(unsigned, prev));
            cp += XP;
            y = 1;
            if (y >= y_end) return;
        }
        unsigned long long wq = ~0ull;                          // ring slot of column y - 1 (requested a column ahead)
        if (!kFirst) wq = MAS_RING_LD((r - 1) * kMasRing + ((y - 1) & (kMasRing - 1)));
        unsigned long long *dp = dirs + ((long)b * Ty + (y - 1)) * R + r;     // bit-plane word of column y - 1
        int d = y - xe;
        for (; y < y_end; ++y, ++d, cp += XP, dp += R) {
            const float c = cnext;
            cnext = cp[XP];
            // x-1 neighbour of the previous column: DPP wave shift inside the group, the ring across groups
            float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, prev), 0x138, 0xf, 0xf, false));
            if constexpr (!kFirst) {
                while ((int)(wq >> 32) != y - 1) wq = MAS_RING_LD((r - 1) * kMasRing + ((y - 1) & (kMasRing - 1)));
                const float carry = __builtin_bit_cast(float, (unsigned)(wq & 0xffffffffull));
                wq = MAS_RING_LD((r - 1) * kMasRing + (y & (kMasRing - 1)));      // the slot the NEXT column needs
                up = (lane == 0) ? carry : up;
            }
            // direction bit-plane of column y-1: value[x,y-1] < value[x-1,y-1] (row 0 has no upper neighbour)
            const unsigned long long bits = __ballot((kFirst ? (lane != 0) : true) && prev < up);
            if (lane == 0) *dp = bits;
            const float v_cur = (d == 0) ? neg : prev;                           // x == y
            const float v_prev = (kFirst && lane == 0) ? neg : up;               // x == 0
            const float nv = vmax(v_cur, v_prev) + c;
            const bool inb = (d >= 0) && (d <= band);
            prev = inb ? nv : c;
            if (need_copy) cp[0] = prev;
            if (publish && lane == 63)
                MAS_RING_ST(r * kMasRing + (y & (kMasRing - 1)), ((unsigned long long)(unsigned)y << 32) | __builtin_bit_cast(unsigned, prev));
        }
    };
    for (int t = 0; t < nt_work; ++t) {
        float *cur = (t & 1) ? buf1 : buf0;
        float *oth = (t & 1) ? buf0 : buf1;
        if (dp_wave) {
            if (t_x > 0) {
                if (r == 0) dp_tile(std::true_type{}, cur, t);
                else dp_tile(std::false_type{}, cur, t);
            }
        } else {
            if (need_copy && t > 0)
                mas_writeback_tile<YT>(oth, dp_values, base, Tx, Ty, XP, t - 1, mover, lane, nmov);
            if (t + 1 < nt_work)
                mas_stage_tile<YT, 16>(oth, in_values, mask, base, Tx, Ty, XP, t + 1, mover, lane, nmov);
        }
        __syncthreads();
    }
    if (need_copy && nt_work > 0 && !dp_wave) {
        const float *last = ((nt_work - 1) & 1) ? buf1 : buf0;
        mas_writeback_tile<YT>(last, dp_values, base, Tx, Ty, XP, nt_work - 1, mover, lane, nmov);
    }
}

#undef MAS_RING_LD
#undef MAS_RING_ST

// Round 6: the same skewed pipeline with a branch-free column step.  What the ISA of the kernel above spends per column and DP
// wave besides the five instructions of the recurrence: the bit-plane word stored by lane 0 (save exec / branch / two moves /
// store / restore), the ring slot published by lane 63 (the same dance + address arithmetic), `lane == 0 ? carry : up`,
// ballot lowered through v_cndmask + v_cmp_ne, a wave-uniform branch around the in-place write, register rotation at the
// loop's back edge: ~50 instructions, five branches.  Here:
//   * the carry from the row group below is the `old` operand of the DPP wave shift (lane 0 has no source lane and keeps it);
//     the first row group passes max_neg_val there (row 0's missing neighbour);
//   * a column's direction word goes into lane (y mod 32) of a register pair with two v_writelane and is stored once per
//     32-column tile by one coalesced store;
//   * every lane publishes {value, column} each column: lane 63 into the ring, the others into a per-wave dump strip — one
//     unconditional ds_write_b64, no exec games;
//   * band membership is ONE unsigned compare of d = y - x against t_y - t_x; NEED_COPY is a template parameter.
// Same fp32 add per cell on the same operands => bit-exact (tests/test_mas_gpu.py runs every case through both kernels).
constexpr int kMasDump = 128;

template <int N, class F, int... I>
__device__ __forceinline__ void mas_static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void mas_static_for(F &&f)
{
    mas_static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// R (row groups = DP waves) is a template parameter: the tile's row pitch 64 R + 1 is a constant, and a FULL 32-column tile is
// walked by an unrolled body in which every LDS address of the column step (next value, in-place write, ring slot to read, ring /
// dump slot to publish) is base + immediate and the direction word's lane is an instruction constant — per column and DP wave
// ~22 instructions (13 VALU, 3-4 LDS, one readfirstlane + compare + branch for the ring tag) instead of ~43; the first tile
// (column 0) and a last partial tile keep the generic loop.
template <int R, bool NEED_COPY>
__global__ __launch_bounds__(64 * kMasMwWaves) void mas_forward_mw2_kernel(
    const float *in_values, const float *__restrict__ mask, float *dp_values, unsigned long long *__restrict__ dirs,
    const int *__restrict__ t_xs, const int *__restrict__ t_ys, int Tx, int Ty, float neg)
{
    constexpr int YT = 32;
    constexpr int XP = 64 * R + 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    float *buf0 = smem;
    float *buf1 = smem + YT * XP;
    unsigned long long *ring = reinterpret_cast<unsigned long long *>(smem + 2 * YT * XP);       // [R][kMasRing] {value, column}
    unsigned long long *dump = ring + R * kMasRing;                                              // [R][kMasDump]
    auto ring_ld = [](const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto ring_st = [](unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    const int t_x = min(t_xs[b], Tx);
    const int t_y = min(t_ys[b], Ty);
    const long base = (long)b * Tx * Ty;
    const int nt = (Ty + YT - 1) / YT;
    const int nt_work = NEED_COPY ? nt : ((t_x > 0 && t_y > 0) ? (min(t_y, Ty) + YT - 1) / YT : 0);
    const bool dp_wave = wave < R;
    const int mover = wave - R;
    constexpr int nmov = kMasMwWaves - R;

    for (int i = threadIdx.x; i < R * kMasRing; i += blockDim.x) ring_st(ring + i, ~0ull);      // tag -1: nothing published
    for (int i = threadIdx.x; i < 2 * YT * XP; i += blockDim.x) {   // padding rows x >= Tx (no mover writes them) read as 0
        const int xx = i % XP;
        if (xx >= Tx) smem[i] = 0.f;
    }
    if (!dp_wave && nt_work > 0)
        mas_stage_tile<YT, 16>(buf0, in_values, mask, base, Tx, Ty, XP, 0, mover, lane, nmov);
    __syncthreads();

    const int r = wave;
    const int x = r * 64 + lane;
    float prev = 0.f;      // this row's value in the previous column
    const int band = t_y - t_x;
    const unsigned band_u = band >= 0 ? (unsigned)band : 0u;
    const unsigned xe = (x < t_x && band >= 0) ? (unsigned)x : 0x3fffffffu;                      // rows outside never enter the band
    // where this lane publishes: lane 63 of a wave with a consumer into the ring, every other lane into the wave's dump strip
    unsigned long long *pub = (r + 1 < R && lane == 63) ? ring + r * kMasRing : dump + (dp_wave ? r : 0) * kMasDump + lane;
    const unsigned long long *sub = ring + (r > 0 ? r - 1 : 0) * kMasRing;                       // the row group below publishes here
    auto vmax = [](float a_, float b_) {
        float o;
        asm("v_max_f32 %0, %1, %2" : "=v"(o) : "v"(a_), "v"(b_));
        return o;
    };
    auto dp_tile = [&](auto first_tag, float *cur, int t) {
        constexpr bool kFirst = decltype(first_tag)::value;
        const int y0 = t * YT;
        const int y_end = min(t_y, min(Ty, (t + 1) * YT));
        if (y0 >= y_end) return;
        float *cp = cur + x;                                    // this lane's row in the tile, one column = XP floats
        int acc_lo = 0, acc_hi = 0;                             // lane j: direction word of column y0 - 1 + j
        if (t > 0 && y_end - y0 == YT) {
            // ---- a full tile: 32 unrolled column steps, immediates everywhere ------------------------------------------------
            const int half = (t & 1) * YT;                      // y & 63 = half + k
            const unsigned long long *sub_t = sub + half;
            unsigned long long *pub_t = pub + half;
            float cnext = cp[0];
            unsigned long long wq = ~0ull;
            if constexpr (!kFirst) wq = ring_ld(sub + ((y0 - 1) & (kMasRing - 1)));
            const unsigned d0 = (unsigned)y0 - xe;
            mas_static_for<YT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int y = y0 + k;
                const float c = cnext;
                cnext = cp[(k + 1) * XP];
                int old = __builtin_bit_cast(int, neg);
                if constexpr (!kFirst) {
                    __builtin_amdgcn_sched_barrier(0);          // the look at the tag stays HERE, a whole step after its request
                    // the first look is peeled out of the retry loop: on the straight path the wait covers only the (older) ring read,
                    // not the publish and the value read issued after it — as the loop's header it waited for every LDS access
                    if (__builtin_expect(__builtin_amdgcn_readfirstlane((int)(wq >> 32)) != y - 1, 0)) {
                        do wq = ring_ld(k == 0 ? sub + ((y0 - 1) & (kMasRing - 1)) : sub_t + (k - 1));
                        while (__builtin_amdgcn_readfirstlane((int)(wq >> 32)) != y - 1);
                    }
                    old = (int)(unsigned)(wq & 0xffffffffull);
                    wq = ring_ld(sub_t + k);                                          // the slot the NEXT column needs
                }
                const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(old, __builtin_bit_cast(int, prev), 0x138, 0xf, 0xf, false));
                unsigned long long bits = __builtin_amdgcn_ballot_w64(prev < up);
                if constexpr (kFirst) bits &= ~1ull;
                asm volatile("v_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
                             : "+v"(acc_lo), "+v"(acc_hi)
                             : "s"((int)(unsigned)(bits & 0xffffffffull)), "s"((int)(unsigned)(bits >> 32)), "n"(k));
                const unsigned d = d0 + (unsigned)k;
                const float v_cur = (d == 0u) ? neg : prev;                          // x == y
                const float nv = vmax(v_cur, up) + c;
                prev = (d <= band_u) ? nv : c;
                if constexpr (NEED_COPY) cp[k * XP] = prev;
                ring_st(pub_t + k, ((unsigned long long)(unsigned)y << 32) | __builtin_bit_cast(unsigned, prev));
            });
            if (lane < YT) dirs[((long)b * Ty + (y0 - 1 + lane)) * R + r] = ((unsigned long long)(unsigned)acc_hi << 32) | (unsigned)acc_lo;
            return;
        }
        // ---- the first tile (column 0 is just prev = value) and a last, partly filled one: the generic loop --------------------
        int y = y0;
        float cnext = cp[0];
        if (t == 0) {                                           // column 0: value[x,0] = value (x = 0: max(neg, 0) + value)
            prev = cnext;
            cnext = cp[XP];
            ring_st(pub, (unsigned long long)__builtin_bit_cast(unsigned, prev));
            cp += XP;
            y = 1;
        }
        if (y < y_end) {
            unsigned long long wq = ~0ull;                      // ring slot of column y - 1 (requested a column ahead)
            if (!kFirst) wq = ring_ld(sub + ((y - 1) & (kMasRing - 1)));
            for (; y < y_end; ++y, cp += XP) {
                const float c = cnext;
                cnext = cp[XP];
                // x-1 neighbour of the previous column: DPP wave shift inside the group; lane 0 keeps `old` = the carry
                int old = __builtin_bit_cast(int, neg);
                if constexpr (!kFirst) {
                    while (__builtin_amdgcn_readfirstlane((int)(wq >> 32)) != y - 1) wq = ring_ld(sub + ((y - 1) & (kMasRing - 1)));
                    old = (int)(unsigned)(wq & 0xffffffffull);
                    wq = ring_ld(sub + (y & (kMasRing - 1)));                         // the slot the NEXT column needs
                }
                const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(old, __builtin_bit_cast(int, prev), 0x138, 0xf, 0xf, false));
                // direction word of column y-1: value[x,y-1] < value[x-1,y-1] (row 0 has no upper neighbour)
                unsigned long long bits = __builtin_amdgcn_ballot_w64(prev < up);
                if constexpr (kFirst) bits &= ~1ull;
                // (no writelane builtin in this clang; two scalar sources of one VOP3 = the lane select goes through m0)
                asm volatile("s_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %4, m0"
                             : "+v"(acc_lo), "+v"(acc_hi)
                             : "s"((int)(unsigned)(bits & 0xffffffffull)), "s"(y - y0), "s"((int)(unsigned)(bits >> 32))
                             : "m0");
                const unsigned d = (unsigned)y - xe;
                const float v_cur = (d == 0u) ? neg : prev;                          // x == y
                const float nv = vmax(v_cur, up) + c;
                prev = (d <= band_u) ? nv : c;
                if constexpr (NEED_COPY) cp[0] = prev;
                ring_st(pub + (y & (kMasRing - 1)), ((unsigned long long)(unsigned)y << 32) | __builtin_bit_cast(unsigned, prev));
            }
        }
        const int j0 = (t == 0) ? 1 : 0;
        if (lane >= j0 && lane < y_end - y0)
            dirs[((long)b * Ty + (y0 - 1 + lane)) * R + r] = ((unsigned long long)(unsigned)acc_hi << 32) | (unsigned)acc_lo;
    };
    for (int t = 0; t < nt_work; ++t) {
        float *cur = (t & 1) ? buf1 : buf0;
        float *oth = (t & 1) ? buf0 : buf1;
        if (dp_wave) {
            if (t_x > 0) {
                if (r == 0) dp_tile(std::true_type{}, cur, t);
                else dp_tile(std::false_type{}, cur, t);
            }
        } else {
            if (NEED_COPY && t > 0)
                mas_writeback_tile<YT>(oth, dp_values, base, Tx, Ty, XP, t - 1, mover, lane, nmov);
            if (t + 1 < nt_work)
                mas_stage_tile<YT, 16>(oth, in_values, mask, base, Tx, Ty, XP, t + 1, mover, lane, nmov);
        }
        __syncthreads();
    }
    if (NEED_COPY && nt_work > 0 && !dp_wave) {
        const float *last = ((nt_work - 1) & 1) ? buf1 : buf0;
        mas_writeback_tile<YT>(last, dp_values, base, Tx, Ty, XP, nt_work - 1, mover, lane, nmov);
    }
}

// ---- forward DP, any T_x (T_x > 2048) ----------------------------------------------------------------
// The kernels above keep a [T_x x 32]-column tile (and the previous column) on the CU; core.pyx:11-47 has no bound on t_x, so
// beyond 32 row groups this kernel takes over.  One workgroup of 16 waves per item; the previous column lives in a
// double-buffered column array — in LDS up to 16 384 rows, in the caller's workspace (global memory, L1/L2 resident) beyond —
// and the block steps column by column with one barrier per column.  A thread's cell of column y+1 is requested before the
// barrier of column y (its row-major neighbour lines are re-used for 32 columns from L1/L2).  Same single fp32 add per
// cell => bit-exact; the direction bit-planes have the layout the backtrack kernel reads.
constexpr int kMasBigThreads = 1024;
constexpr int kMasBigLdsRows = 16384;

template <bool GSTATE>   // column state in the workspace (global memory) instead of LDS
__global__ __launch_bounds__(kMasBigThreads) void mas_forward_big_kernel(
    const float *in_values, const float *__restrict__ mask, float *dp_values, unsigned long long *__restrict__ dirs,
    float *gcol /* [B][2][Tx] or nullptr (LDS) */, const int *__restrict__ t_xs, const int *__restrict__ t_ys, int Tx, int Ty,
    int R, float neg)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int t_x = min(t_xs[b], Tx);
    const int t_y = min(t_ys[b], Ty);
    const long base = (long)b * Tx * Ty;
    // (two instantiations rather than one generic pointer: flat accesses into the LDS aperture faulted on this stack)
    auto col_ptr = [&]() {
        if constexpr (GSTATE) return gcol + (long)b * 2 * Tx;
        else return smem;
    };
    auto *const col = col_ptr();
    const bool need_copy = (dp_values != nullptr);
    const int nrow = R * 64;                                    // rows swept (whole 64-row groups: ballots are per group)
    // When the caller wants the full value*mask matrix back (dp_values_out != values_in), cells outside the band are copied.
    const bool copy_all = need_copy && (dp_values != in_values || mask != nullptr);
    if (t_x <= 0 || t_y <= 0) {
        if (copy_all)
            for (long i = tid; i < (long)Tx * Ty; i += kMasBigThreads) dp_values[base + i] = mask ? in_values[base + i] * mask[base + i] : in_values[base + i];
        return;
    }
    for (int x = tid; x < Tx; x += kMasBigThreads) col[x] = 0.f;
    __syncthreads();
    const int y_stop = copy_all ? Ty : t_y;
    for (int y = 0; y < y_stop; ++y) {
        const auto *cp = col + (y & 1) * Tx;                    // column y-1
        auto *cn = col + ((y + 1) & 1) * Tx;                    // column y
        const int x_lo = max(0, t_x + y - t_y);
        const int x_hi = min(t_x, y + 1);
        const bool dp_col = y < t_y;
        for (int x = tid; x < nrow; x += kMasBigThreads) {
            const bool xin = x < Tx;
            float c = 0.f;
            if (xin) {
                const long o = base + (long)x * Ty + y;
                c = mask ? in_values[o] * mask[o] : in_values[o];
            }
            const float prev = xin ? cp[x] : 0.f;
            const float up = (x >= 1 && x - 1 < Tx) ? cp[x - 1] : 0.f;
            if (dp_col && y > 0) {                               // direction bit-plane of column y-1
                const unsigned long long bits = __ballot(xin && x >= 1 && prev < up);
                if ((tid & 63) == 0) dirs[((long)b * Ty + (y - 1)) * R + (x >> 6)] = bits;
            }
            const float v_cur = (x == y) ? neg : prev;
            const float v_prev = (x == 0) ? (y == 0 ? 0.f : neg) : up;
            const float nv = fmaxf(v_cur, v_prev) + c;
            const bool inb = dp_col && (x >= x_lo) && (x < x_hi);
            const float keep = inb ? nv : c;
            if (xin) {
                cn[x] = keep;
                if (need_copy && (inb || copy_all)) dp_values[base + (long)x * Ty + y] = keep;
            }
        }
        __syncthreads();
    }
}

// ---- backtrack -------------------------------------------------------------------------------
// Backtrack: wave 0 walks the columns from the last to the first in 64-column chunks (the index chain is serial:
// core.pyx:34-37); per chunk every lane holds a 64-row window of its column's direction bit-plane around the index the
// chunk starts from, and the walk itself is scalar readlane work.  The bit-planes of the NEXT chunk are requested before
// the current chunk is walked (its start index can only be 0..64 rows below the current one: three candidate row groups
// are fetched and two selected afterwards), and waves 1..7 write chunk c's path columns while wave 0 already walks chunk
// c-1 (double-buffered index row, one barrier per chunk) — neither an HBM round trip nor the path stores sit on the
// serial chain.
constexpr int kMasBtThreads = 512;

// WALK2 (round 6, default): the chunk's serial walk with nothing but the recurrence on the chain.  The two conditions of
// core.pyx:35 that do not come from the DP values — `index == y` (on the diagonal the path must step) and `index != 0` — and the
// chunk's column range are folded into the per-column window words by the lanes BEFORE the walk (a few vector instructions per
// chunk); the walk then carries one scalar, the bit position k = index - index0 + 63, through 64 unrolled steps of
// {two v_readlane, one v_writelane recording k, shift, and, subtract} with instruction-constant lanes, and the indices are
// rebuilt from the recorded k afterwards.  The round-2 walk spent ~25 instructions per column on the same chain.
template <typename PathT, bool WALK2>
__global__ __launch_bounds__(kMasBtThreads) void mas_backtrack_kernel(
    PathT *__restrict__ paths, const unsigned long long *__restrict__ dirs,
    const int *__restrict__ t_xs, const int *__restrict__ t_ys, int Tx, int Ty, int R,
    int prezeroed)
{
    __shared__ int s_idx[2][64];
    const int b = blockIdx.x;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int t_x = min(t_xs[b], Tx);
    const int t_y = min(t_ys[b], Ty);
    const bool valid = (t_x > 0) && (t_y > 0);
    const long base = (long)b * Tx * Ty;
    int idx = t_x - 1;  // wave-uniform (core.pyx:18)
    const int nchunks = (Ty + 63) / 64;
    constexpr int kWriters = kMasBtThreads / 64 - 1;

    // candidate bit-planes of chunk c for a start index whose row group is rg, rg-1 (window: + one group below each)
    unsigned long long cand[3] = {0ull, 0ull, 0ull};
    int cand_rg = 0;
    auto fetch = [&](int c, int rg) {
        const int y = c * 64 + lane;
        cand_rg = rg;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int r = rg - i;
            cand[i] = (valid && wave == 0 && c >= 0 && y >= 1 && y < t_y && r >= 0)
                          ? dirs[((long)b * Ty + (y - 1)) * R + r] : 0ull;
        }
    };
    if (wave == 0) fetch(nchunks - 1, idx >> 6);

    for (int c = nchunks - 1; c >= 0; --c) {
        const int y_lo = c * 64;
        if (wave == 0) {
            int myidx = -1;
            const int idx0 = idx;
            const int r0 = idx0 >> 6;
            const int sel = cand_rg - r0;            // 0 or 1: which candidate pair belongs to this start index
            const unsigned long long hi = sel == 0 ? cand[0] : cand[1];
            const unsigned long long lo = sel == 0 ? cand[1] : cand[2];
            fetch(c - 1, r0);                        // next chunk: in flight during this chunk's walk
            if (valid && y_lo < t_y) {
                const int s = (idx0 & 63) + 1;  // window bit k <-> row idx0-63+k
                const unsigned long long window = (s == 64) ? hi : ((lo >> s) | (hi << (64 - s)));
                const int wlo = (int)(unsigned)(window & 0xffffffffull);
                const int whi = (int)(unsigned)(window >> 32);
                const int jmax = min(63, t_y - 1 - y_lo);
                if constexpr (WALK2) {
                    const int sidx0 = __builtin_amdgcn_readfirstlane(idx);
                    const int yy = y_lo + lane;
                    unsigned long long w = window;
                    const int kd = yy - sidx0 + 63;                                 // the diagonal row (index == y) in window bits
                    if ((unsigned)kd < 64u) w |= 1ull << kd;
                    if (sidx0 <= 63) w &= ~(1ull << (63 - sidx0));                  // row 0 never steps
                    if (yy <= 0 || lane > jmax) w = 0ull;                           // column 0 / columns outside the item
                    const int vlo = (int)(unsigned)(w & 0xffffffffull), vhi = (int)(unsigned)(w >> 32);
                    int sk = 63, vk = 63;
                    mas_static_for<64>([&](auto jc) {
                        constexpr int j = 63 - decltype(jc)::value;
                        const unsigned ulo = (unsigned)__builtin_amdgcn_readlane(vlo, j);
                        const unsigned uhi = (unsigned)__builtin_amdgcn_readlane(vhi, j);
                        asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vk) : "s"(sk), "n"(j));
                        const unsigned long long ww = ((unsigned long long)uhi << 32) | ulo;
                        sk = __builtin_amdgcn_readfirstlane(sk - (int)((ww >> (sk & 63)) & 1ull));
                    });
                    myidx = (lane <= jmax) ? sidx0 - 63 + vk : -1;
                    idx = sidx0 - 63 + sk;
                } else {
                // the index chain is serial (core.pyx:34-37): kept in a SCALAR register (readfirstlane pins it: left to itself
                // hipcc ran the chain on the vector ALU — a 64-bit vector shift and three selects per column, ~140 cycles;
                // the scalar chain is a handful of one-cycle ops), the per-lane bookkeeping (myidx) stays off the chain
                int sidx = __builtin_amdgcn_readfirstlane(idx);
                const int sidx0 = sidx;
                for (int j = jmax; j >= 0; --j) {
                    const int yy = y_lo + j;
                    if (lane == j) myidx = sidx;
                    const unsigned ulo = (unsigned)__builtin_amdgcn_readlane(wlo, j);
                    const unsigned uhi = (unsigned)__builtin_amdgcn_readlane(whi, j);
                    const unsigned long long w = ((unsigned long long)uhi << 32) | ulo;
                    const int k = sidx - sidx0 + 63;
                    const bool dec = (sidx != 0) && (yy > 0) && ((sidx == yy) || ((w >> k) & 1ull));
                    sidx = __builtin_amdgcn_readfirstlane(sidx - (dec ? 1 : 0));
                }
                idx = sidx;
                }
            }
            s_idx[c & 1][lane] = myidx;
            if (prezeroed && myidx >= 0) paths[base + (long)myidx * Ty + y_lo + lane] = (PathT)1;
        }
        __syncthreads();        // chunk c's index row is published; the writers of chunk c+1 are done with the other row
        if (wave > 0 && !prezeroed) {
            const int y = y_lo + lane;
            const int m = s_idx[c & 1][lane];
            if (y < Ty)
                for (int x = wave - 1; x < Tx; x += kWriters) paths[base + (long)x * Ty + y] = (x == m) ? (PathT)1 : (PathT)0;
        }
    }
}

// t_xs[b] = sum_x mask[b,x,0]; t_ys[b] = sum_y mask[b,0,y]   (helpers.py:191-192)
__global__ void mask_lengths_kernel(int *__restrict__ t_xs, int *__restrict__ t_ys,
                                    const float *__restrict__ mask, int Tx, int Ty)
{
    __shared__ float red[2][4];
    const int b = blockIdx.x;
    const float *m = mask + (long)b * Tx * Ty;
    float sx = 0.f, sy = 0.f;
    for (int x = threadIdx.x; x < Tx; x += blockDim.x) sx += m[(long)x * Ty];
    for (int y = threadIdx.x; y < Ty; y += blockDim.x) sy += m[y];
    for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_down(sx, o);
        sy += __shfl_down(sy, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = sx; red[1][wave] = sy; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { a += red[0][w]; c += red[1][w]; }
        t_xs[b] = (int)a;
        t_ys[b] = (int)c;
    }
}

static int launch_forward_mw(const float *in, const float *mask, float *dp, unsigned long long *dirs, const int *t_xs,
                             const int *t_ys, int B, int Tx, int Ty, int R, float neg, hipStream_t st)
{
    // A/B switch: TTSAMD_MAS_MW=1 selects the round-3 column step (branches per column), default the branch-free one
    static const bool v1 = getenv("TTSAMD_MAS_MW") && atoi(getenv("TTSAMD_MAS_MW")) == 1;
    if (!v1) {
        const size_t lds2 = ((size_t)2 * 32 * (64 * R + 1) + (size_t)2 * R * (kMasRing + kMasDump)) * sizeof(float);
        static std::atomic<unsigned long long> done[2][9];
#define TTSAMD_MAS_MW2(n)                                                                                                              \
    case n:                                                                                                                            \
        if (dp) {                                                                                                                      \
            TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(mas_forward_mw2_kernel<n, true>), 160 * 1024, done[1][n]));   \
            hipLaunchKernelGGL((mas_forward_mw2_kernel<n, true>), dim3(B), dim3(64 * kMasMwWaves), lds2, st, in, mask, dp, dirs, t_xs, \
                               t_ys, Tx, Ty, neg);                                                                                     \
        } else {                                                                                                                       \
            TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(mas_forward_mw2_kernel<n, false>), 160 * 1024, done[0][n]));  \
            hipLaunchKernelGGL((mas_forward_mw2_kernel<n, false>), dim3(B), dim3(64 * kMasMwWaves), lds2, st, in, mask, dp, dirs,      \
                               t_xs, t_ys, Tx, Ty, neg);                                                                               \
        }                                                                                                                              \
        break;
        switch (R) {
            TTSAMD_MAS_MW2(1) TTSAMD_MAS_MW2(2) TTSAMD_MAS_MW2(3) TTSAMD_MAS_MW2(4)
            TTSAMD_MAS_MW2(5) TTSAMD_MAS_MW2(6) TTSAMD_MAS_MW2(7) TTSAMD_MAS_MW2(8)
            default: set_error("maximum_path: %d row groups have no skewed-pipeline instantiation", R); return TTSAMD_ERR_INVALID;
        }
#undef TTSAMD_MAS_MW2
        TTSAMD_LAUNCH_CHECK();
        return TTSAMD_OK;
    }
    const size_t lds = ((size_t)2 * 32 * (64 * R + 1) + (size_t)2 * R * kMasRing) * sizeof(float);
    static std::atomic<unsigned long long> lds_attr_done{0};
    TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(mas_forward_mw_kernel), 160 * 1024, lds_attr_done));
    hipLaunchKernelGGL(mas_forward_mw_kernel, dim3(B), dim3(64 * kMasMwWaves), lds, st, in, mask, dp, dirs, t_xs, t_ys,
                       Tx, Ty, R, neg);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

template <int RMAX, int YT, bool EXACT = false>
static int launch_forward(const float *in, const float *mask, float *dp, unsigned long long *dirs,
                          const int *t_xs, const int *t_ys, int B, int Tx, int Ty, int R, float neg,
                          hipStream_t st)
{
    const size_t lds = (size_t)2 * YT * (64 * R + 1) * sizeof(float);
    auto kern = mas_forward_kernel<RMAX, YT, EXACT>;
    TTSAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(B), dim3(kMasThreads), lds, st, in, mask, dp, dirs, t_xs, t_ys, Tx,
                       Ty, R, neg);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

}  // namespace ttsamd

using namespace ttsamd;

// test hooks: TTSAMD_MAS_FORCE_BIG=1 routes every shape through the any-T_x kernel, =2 also puts its column state in global
// memory (what T_x > 16 384 does) — tests/test_mas_gpu.py runs the bit-exactness cases through both
static int mas_force_big()
{
    static const int v = getenv("TTSAMD_MAS_FORCE_BIG") ? atoi(getenv("TTSAMD_MAS_FORCE_BIG")) : 0;
    return v;
}

static size_t mas_dirs_bytes(int b, int t_x, int t_y)
{
    const size_t R = (size_t)(t_x + 63) / 64;
    return (size_t)b * (size_t)t_y * R * sizeof(unsigned long long);
}

extern "C" size_t ttsamd_maximum_path_workspace_bytes(int b, int t_x, int t_y)
{
    if (b <= 0 || t_x <= 0 || t_y <= 0) return 0;
    // direction bit-planes (+ beyond 16 384 rows the any-T_x kernel's double-buffered column state)
    const bool gstate = t_x > kMasBigLdsRows || mas_force_big() == 2;
    return ((mas_dirs_bytes(b, t_x, t_y) + 15) & ~(size_t)15) + (gstate ? (size_t)b * 2 * (size_t)t_x * sizeof(float) : 0);
}

extern "C" int ttsamd_maximum_path(void *paths, const float *values_in, const float *mask,
                                   float *dp_values_out, const int32_t *t_xs, const int32_t *t_ys,
                                   int b, int t_x, int t_y, float max_neg_val, void *workspace,
                                   size_t workspace_bytes, int flags, void *stream)
{
    TTSAMD_CHECK_ARG(b >= 0 && t_x >= 0 && t_y >= 0, "maximum_path: negative shape");
    if (b == 0 || t_x == 0 || t_y == 0) return TTSAMD_OK;
    TTSAMD_CHECK_ARG(paths && values_in && t_xs && t_ys, "maximum_path: NULL pointer");
    const int R = (t_x + 63) / 64;
    const size_t need = ttsamd_maximum_path_workspace_bytes(b, t_x, t_y);
    TTSAMD_CHECK_ARG(workspace && workspace_bytes >= need, "maximum_path: workspace too small (%zu < %zu)",
                     workspace_bytes, need);
    hipStream_t st = as_stream(stream);
    auto *dirs = reinterpret_cast<unsigned long long *>(workspace);
    int rc;
    static const bool single_wave = getenv("TTSAMD_MAS_SINGLE_WAVE") != nullptr;   // A/B switch: the one-DP-wave kernel
    const int force_big = mas_force_big();
    if (R > 32 || force_big) {
        const bool gstate = t_x > kMasBigLdsRows || force_big == 2;
        float *gcol = nullptr;
        if (gstate) {
            const size_t off = (mas_dirs_bytes(b, t_x, t_y) + 15) & ~(size_t)15;
            gcol = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + off);
        }
        const size_t lds = gstate ? 0 : (size_t)2 * t_x * sizeof(float);
        static std::atomic<unsigned long long> lds_attr_done{0};
        if (gstate) {
            hipLaunchKernelGGL(mas_forward_big_kernel<true>, dim3(b), dim3(kMasBigThreads), 0, st, values_in, mask, dp_values_out, dirs,
                               gcol, t_xs, t_ys, t_x, t_y, R, max_neg_val);
        } else {
            TTSAMD_HIP(ensure_dynamic_lds(reinterpret_cast<const void *>(mas_forward_big_kernel<false>),
                                          2 * kMasBigLdsRows * (int)sizeof(float), lds_attr_done));
            hipLaunchKernelGGL(mas_forward_big_kernel<false>, dim3(b), dim3(kMasBigThreads), lds, st, values_in, mask, dp_values_out,
                               dirs, gcol, t_xs, t_ys, t_x, t_y, R, max_neg_val);
        }
        rc = (hipGetLastError() == hipSuccess) ? TTSAMD_OK : TTSAMD_ERR_HIP;
        if (rc != TTSAMD_OK) set_error("maximum_path: launch of the any-T_x kernel failed");
    } else if (R <= 8 && !single_wave) {
        rc = launch_forward_mw(values_in, mask, dp_values_out, dirs, t_xs, t_ys, b, t_x, t_y, R, max_neg_val, st);
    } else
#define TTSAMD_MAS_EXACT(n) \
    case n: rc = launch_forward<n, 32, true>(values_in, mask, dp_values_out, dirs, t_xs, t_ys, b, t_x, t_y, R, max_neg_val, st); break;
    if (R <= 8) {
        switch (R) {
            TTSAMD_MAS_EXACT(1) TTSAMD_MAS_EXACT(2) TTSAMD_MAS_EXACT(3) TTSAMD_MAS_EXACT(4)
            TTSAMD_MAS_EXACT(5) TTSAMD_MAS_EXACT(6) TTSAMD_MAS_EXACT(7) TTSAMD_MAS_EXACT(8)
            default: rc = TTSAMD_ERR_INVALID;
        }
    }
#undef TTSAMD_MAS_EXACT
    else if (R <= 16) rc = launch_forward<16, 16>(values_in, mask, dp_values_out, dirs, t_xs, t_ys, b, t_x, t_y, R, max_neg_val, st);
    else              rc = launch_forward<32, 8>(values_in, mask, dp_values_out, dirs, t_xs, t_ys, b, t_x, t_y, R, max_neg_val, st);
    if (rc != TTSAMD_OK) return rc;
    const int prezeroed = (flags & TTSAMD_MAS_PATHS_PREZEROED) ? 1 : 0;
    static const bool walk1 = getenv("TTSAMD_MAS_BT") && atoi(getenv("TTSAMD_MAS_BT")) == 1;     // A/B switch: the round-2 walk
#define TTSAMD_MAS_BT(T, W) \
    hipLaunchKernelGGL((mas_backtrack_kernel<T, W>), dim3(b), dim3(kMasBtThreads), 0, st, (T *)paths, dirs, t_xs, t_ys, t_x, t_y, R, prezeroed)
    if (flags & TTSAMD_MAS_PATHS_F32) {
        if (walk1) TTSAMD_MAS_BT(float, false); else TTSAMD_MAS_BT(float, true);
    } else {
        if (walk1) TTSAMD_MAS_BT(int, false); else TTSAMD_MAS_BT(int, true);
    }
#undef TTSAMD_MAS_BT
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}

extern "C" int ttsamd_maximum_path_c(int32_t *paths, float *values, const int32_t *t_xs,
                                     const int32_t *t_ys, int b, int t_x, int t_y, float max_neg_val,
                                     void *stream)
{
    const size_t need = ttsamd_maximum_path_workspace_bytes(b, t_x, t_y);
    if (need == 0) return TTSAMD_OK;
    hipStream_t st = as_stream(stream);
    void *ws = nullptr;
    TTSAMD_HIP(hipMallocAsync(&ws, need, st));
    const int rc = ttsamd_maximum_path(paths, values, nullptr, values, t_xs, t_ys, b, t_x, t_y, max_neg_val,
                                       ws, need, TTSAMD_MAS_PATHS_PREZEROED, stream);
    hipError_t e = hipFreeAsync(ws, st);
    if (rc != TTSAMD_OK) return rc;
    TTSAMD_HIP(e);
    return TTSAMD_OK;
}

extern "C" int ttsamd_mask_lengths(int32_t *t_xs, int32_t *t_ys, const float *mask, int b, int t_x,
                                   int t_y, void *stream)
{
    TTSAMD_CHECK_ARG(b >= 0 && t_x > 0 && t_y > 0 && t_xs && t_ys && mask, "mask_lengths: bad args");
    if (b == 0) return TTSAMD_OK;
    hipLaunchKernelGGL(mask_lengths_kernel, dim3(b), dim3(256), 0, as_stream(stream), t_xs, t_ys, mask, t_x, t_y);
    TTSAMD_LAUNCH_CHECK();
    return TTSAMD_OK;
}
