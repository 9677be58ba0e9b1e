// ranenv_numeric.hpp -- device helpers shared by the step kernels (ranenv_step.hip) and the sidecar kernels (ranenv_aux.hip): numpy's
// arithmetic on the device (isclose, correctly rounded division, pairwise sums over LDS rows), DPP row sums / scans, the SE row
// reduction (software-pipelined stream + numpy's pairwise order), the SE gather, Philox-4x32-10 and the Poisson table walk.
#pragma once
#include <type_traits>

#include "ranenv_internal.h"

#define DEVFN __device__ __forceinline__

#ifndef RANENV_FAST_DIV
#define RANENV_FAST_DIV 1   /* 0: every f64 division through the plain operator (A/B and the parity check of ddiv itself) */
#endif

namespace {
using namespace ranenv_dev;

// ---------------------------------------------------------------------------------------------
// numpy arithmetic on the device
// ---------------------------------------------------------------------------------------------
DEVFN bool d_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }

// a / b, correctly rounded, for operands whose quotient needs no scaling: the compiler's f64 division without its three guard
// instructions (v_div_scale x 2 -- they return their operands unchanged unless an exponent sits near the ends of the range --, and
// v_div_fixup, which passes the quotient through unless an operand is 0 / inf / nan / denormal): reciprocal estimate, two Newton
// steps, quotient, one residual correction -- the same instructions in the same order, so the same bits.  8 instead of 11 vector
// instructions, and the step kernel makes ~30 divisions per wave and TTI.  Only where the divisor is a positive normal number
// whenever the result is USED (packet sizes, counts, sums guarded by the caller; magnitudes 1e-9...1e12); the intent-drift formulas
// and the means that may be 0 / 0 in the reference too keep the plain operator.
DEVFN double ddiv(double a, double b)
{
#if RANENV_FAST_DIV
    double r = __builtin_amdgcn_rcp(b);
    double e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    e = fma(-b, r, 1.0);
    r = fma(r, e, r);
    const double q = a * r;
    const double res = fma(-b, q, a);
    return fma(res, r, q);
#else
    return a / b;
#endif
}

// numpy pairwise_sum of n <= 16 doubles: missing elements count as +0.0, which turns numpy's three
// shapes for n <= 16 (n < 8 plain loop; 8 <= n < 16 tree of the first 8 + sequential tail; n == 16
// tree of 8 pair sums) into plain expressions, since x + 0.0 == x exactly.
// The row of 16 doubles sits in LDS and its entries at positions >= n are +0.0 (every writer
// in this file zero-pads its rows), so no per-element select is needed; all 16 reads issue back to back.
// Of numpy's three shapes only those some lane of the wave needs are evaluated (wave-uniform tests):
// an instruction costs the same with one active lane as with 64.
// NP (template) = how many leading entries of a row can be non-zero at all in this build of the kernel (the largest slice /
// the number of slices, rounded up to 8, 10 or 16): entries from NP on are never read, their additions (+0.0) never issued.
template <int NP>
DEVFN double np_sum_lds(const double *row, int n)
{
    static_assert(NP >= 8 && NP <= 16, "row builds: 8, 10, 16");
    constexpr int NY = NP - 8;             // entries of the second half that can be non-zero
    // The two halves of the row are read one after the other, so that 8 (not 16) doubles are alive at a time in the
    // common shapes; only numpy's n == 16 shape pairs element j with element 8 + j and re-reads the first half.
    double res = 0.0;
    double x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = row[j];
    double t8 = 0.0;
    if (__builtin_amdgcn_ballot_w64(n < 8) != 0) {
        const double seq = ((((((x[0] + x[1]) + x[2]) + x[3]) + x[4]) + x[5]) + x[6]) + x[7];
        res = n < 8 ? seq : res;
    }
    const bool mid = n >= 8 && n < 16;
    if (__builtin_amdgcn_ballot_w64(mid) != 0)
        t8 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    if (NY > 0 && __builtin_amdgcn_ballot_w64(n > 8) != 0) {            // (n == 8: the tree alone, nothing to add)
        double y[NY > 0 ? NY : 1];
#pragma unroll
        for (int j = 0; j < NY; j++) y[j] = row[8 + j];
        if (__builtin_amdgcn_ballot_w64(mid) != 0) {
#pragma unroll
            for (int j = 0; j < (NY < 7 ? NY : 7); j++) t8 += y[j];
        }
        if constexpr (NP == 16) {
            if (__builtin_amdgcn_ballot_w64(n >= 16) != 0) {
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = row[j];
                const double t16 = (((x[0] + y[0]) + (x[1] + y[1])) + ((x[2] + y[2]) + (x[3] + y[3]))) +
                                   (((x[4] + y[4]) + (x[5] + y[5])) + ((x[6] + y[6]) + (x[7] + y[7])));
                res = n >= 16 ? t16 : res;
            }
        }
    }
    return mid ? t8 : res;
}
DEVFN double np_sum16_lds(const double *row, int n) { return np_sum_lds<16>(row, n); }

DEVFN bool d_apply_op(int op, double a, double b)
{
    switch (op) {
    case RANENV_OP_GE: return a >= b;
    case RANENV_OP_LE: return a <= b;
    case RANENV_OP_EQ: return a == b;
    case RANENV_OP_GT: return a > b;
    default: return a < b;
    }
}

// Sums and inclusive scans over the 16 lanes of a DPP row (= one slice's lanes): data-parallel-primitive
// moves inside the VALU instead of ds_bpermute round trips through the LDS crossbar.
template <int CTRL> DEVFN int dpp_row(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
DEVFN int row16_sum(int x)          // every lane gets the sum of its row: rotate right by 1, 2, 4, 8
{
    x += dpp_row<0x121>(x); x += dpp_row<0x122>(x); x += dpp_row<0x124>(x); x += dpp_row<0x128>(x);
    return x;
}
DEVFN double row16_sum_f64(double x)  // the same for a double (two 32-bit moves per step); a fixed tree, not numpy's order
{
    auto rot = [](double v, auto ctrl) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = dpp_row<decltype(ctrl)::value>((int)b), hi = dpp_row<decltype(ctrl)::value>((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    };
    x += rot(x, std::integral_constant<int, 0x121>{}); x += rot(x, std::integral_constant<int, 0x122>{});
    x += rot(x, std::integral_constant<int, 0x124>{}); x += rot(x, std::integral_constant<int, 0x128>{});
    return x;
}
// max of two doubles that are not NaN: ONE v_max_f64 (fmax() canonicalises both operands first -- two more instructions -- for signalling NaNs)
DEVFN double dmax(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
DEVFN double row16_max_f64(double x)  // every lane gets the maximum of its row (no NaNs here: comparisons and v_max agree)
{
    auto rot = [](double v, auto ctrl) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = dpp_row<decltype(ctrl)::value>((int)b), hi = dpp_row<decltype(ctrl)::value>((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    };
    x = dmax(x, rot(x, std::integral_constant<int, 0x121>{})); x = dmax(x, rot(x, std::integral_constant<int, 0x122>{}));
    x = dmax(x, rot(x, std::integral_constant<int, 0x124>{})); x = dmax(x, rot(x, std::integral_constant<int, 0x128>{}));
    return x;
}
DEVFN double wave_sum_f64(double x)   // sum over the 64 lanes of the wave (all active): rows by DPP, then the four row sums
{
    x = row16_sum_f64(x);
    auto lane = [](double v, int l) {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), l);
        return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    };
    return (lane(x, 0) + lane(x, 16)) + (lane(x, 32) + lane(x, 48));
}
DEVFN double half_sum_f64(double x)   // sum over the 32 lanes of this lane's half of the wave (packed waves): two DPP rows, then across them
{
    x = row16_sum_f64(x);
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __shfl_xor((int)b, 16), hi = __shfl_xor((int)(b >> 32), 16);
    const double y = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    return (threadIdx.x & 16) ? y + x : x + y;       // (lower row first in both lanes: one order of the two addends)
}
// global_atomic_add_f64 without a return value: nothing waits for it (the library is built with the atomic optimizer
// off: every add here already comes from one lane)
DEVFN void acc_add(double *p, double v)
{
    typedef __attribute__((address_space(1))) double *gptr;
    (void)__builtin_amdgcn_global_atomic_fadd_f64((gptr)p, v);
}
DEVFN int row16_scan(int x)         // inclusive prefix sum: shift right by 1, 2, 4, 8 (zeros shifted in)
{
    x += dpp_row<0x111>(x); x += dpp_row<0x112>(x); x += dpp_row<0x114>(x); x += dpp_row<0x118>(x);
    return x;
}

// ---------------------------------------------------------------------------------------------
// SE row reduction: software-pipelined stream + numpy's pairwise order
// ---------------------------------------------------------------------------------------------
// Lane u reduces row u of an RB-major tile (element r at byte offset r*U*4 + u*4).  Loads go through
// a wave-uniform buffer descriptor: the row offset is a scalar, the lane offset one VGPR, so a load
// costs no vector address arithmetic.  SE_NQ groups of 8 loads rotate through fixed registers (no
// moves), i.e. up to 8*SE_NQ loads per lane are in flight while a group is being summed (measured:
// 16, 24 and 48 give the same kernel time within 1 us).
//
// Summation order = numpy's pairwise_sum (see np_sum_lds): the row is cut into leaves of <= 128
// RBs by halving at multiples of 8; inside a leaf, accumulator j takes the elements j mod 8, the
// eight accumulators are combined as a fixed tree and the (< 8) tail is added sequentially.  All
// leaves except the last are multiples of 8 long, so leaves and 8-groups stay aligned.
struct RowPlan {          // wave-uniform
    int n_leaves, len0, len1, len2, len3;
    bool lsplit, rsplit;
};

DEVFN RowPlan make_row_plan(int n)
{
    RowPlan pl;
    pl.n_leaves = 1; pl.len0 = n; pl.len1 = 0; pl.len2 = 0; pl.len3 = 0; pl.lsplit = false; pl.rsplit = false;
    if (n > 128) {
        int n2 = n / 2; n2 -= n2 % 8;
        const int nr = n - n2;
        int l0 = n2, l1 = 0, r0 = nr, r1 = 0;
        if (n2 > 128) { int h = n2 / 2; h -= h % 8; l0 = h; l1 = n2 - h; pl.lsplit = true; }
        if (nr > 128) { int h = nr / 2; h -= h % 8; r0 = h; r1 = nr - h; pl.rsplit = true; }
        pl.len0 = l0;
        if (pl.lsplit) { pl.len1 = l1; pl.len2 = r0; pl.len3 = r1; }
        else { pl.len1 = r0; pl.len2 = r1; }
        pl.n_leaves = 2 + (pl.lsplit ? 1 : 0) + (pl.rsplit ? 1 : 0);
    }
    return pl;
}

// 8-row groups of the SE tile in flight per lane: the lean streaming kernel (step / reset at row widths 8 and 10: 96 VGPRs without spills;
// the dense-mask and 16-wide builds keep 1), the build for batches that do not fill the CUs anyway (4 waves per SIMD), and the whole-row
// builds of a batch at <= 2 waves per SIMD (R = 135: 16 groups of 8 + the tail group -- no load is requested inside the stream phase)
constexpr int SE_DEPTH_LEAN = 2, SE_DEPTH_SMALL = 4, SE_DEPTH_TINY = 17, SE_DEPTH_PACKED = 2;
#ifndef RANENV_CACHE_HINTS
#define RANENV_CACHE_HINTS 1       /* 0: plain loads and stores everywhere (A/B of the hints only).  1: the tile loads of the big-batch streaming builds and
                                      the gather builds' loads from the UE-major copy carry the non-temporal bit (gfx94x cache policy bits: 1 = sc0, 2 = nt,
                                      16 = sc1), and the gather builds store what is not read again soon non-temporally (see nt_store) */
#endif
constexpr int SE_AUX_NT = RANENV_CACHE_HINTS ? 2 : 0;

// Two tile layouts (ranenv_bind_se_pool / ranenv_bind_se_pool_quad), a wave-uniform flag of the launch:
//   RB-major       [R][U]        one dword per RB and lane: 8 load instructions per group of 8 RBs
//   RB-quad-major  [R/4][U][4]   four consecutive RBs of a UE side by side: one dwordx4 per four RBs, 2 instructions per group.  A wave-load
//                                then covers 1 KB of contiguous memory instead of 256 B and a tile takes a quarter of the memory
//                                instructions: the same registers in flight stream 6.6 instead of 5.6 TB/s at the headline's
//                                occupancy (tools/tile_probe.hip, profiles/r05_ab_log.txt), and a whole row of 135 RBs is 34
//                                instructions per lane -- below the 63 a wave can have in flight (vmcnt is a 6-bit counter).
typedef float se_v4f __attribute__((ext_vector_type(4)));
template <int SE_NQ>              // 8-row groups in flight per lane
struct SeStream {
    float q[SE_NQ][8];
    __amdgpu_buffer_rsrc_t rsrc;   // wave-uniform descriptor of the tile (SGPRs)
    int voff, row_bytes;           // lane's byte offset inside a (quad-)row; bytes per (quad-)row
    bool quad;
    int last_row;                  // byte offset of the tile's last (quad-)row
    // cache policy of the tile loads: non-temporal in the builds for big batches (queue of <= 2 groups), where the tiles would push the
    // per-UE state out of the caches (see nt_store); plain in the deep-queue builds of small batches, which are latency-bound and lose
    // 9 % with the hint (configs[1]: 18.9 -> 20.6 us per TTI)
    static constexpr int AUX = SE_NQ <= 2 ? SE_AUX_NT : 0;
    DEVFN void load(float (&dst)[8], int r0)            // r0: a multiple of 8
    {
        // The scalar offset of a buffer load takes no part in the descriptor's range check: rows past the tile (the
        // padding of the last, partial group, never summed) are clamped to the last row instead (scalar min).
        if (quad) {
            const int s0 = (r0 >> 2) * row_bytes, s1 = s0 + row_bytes;
            const se_v4f a = __builtin_bit_cast(se_v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s0 < last_row ? s0 : last_row, AUX));
            const se_v4f b = __builtin_bit_cast(se_v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, s1 < last_row ? s1 : last_row, AUX));
            dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
            return;
        }
        int soff = r0 * row_bytes;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int so = soff < last_row ? soff : last_row;
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, so, AUX));
            soff += row_bytes;
        }
    }
    DEVFN void init(const float *tile, int U, int u, int R, bool quad_ = false)
    {
        quad = quad_;
        const int rows = quad ? (R + 3) >> 2 : R, rbytes = quad ? U * 16 : U * 4;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, rows * rbytes, 0x00020000);
        voff = quad ? u * 16 : u * 4; row_bytes = rbytes; last_row = (rows - 1) * rbytes;
#pragma unroll
        for (int d = 0; d < SE_NQ; d++) if (d * 8 < R) load(q[d], d * 8);
    }
    // the queue as row_sums sees it: slot d's eight values (`after` = groups requested behind it: unused here, the
    // compiler counts its own loads; a queue kept in LDS by buffer_load ... lds with hand-written waits was built on this
    // interface, measured and dropped, profiles/r03_ab_log.txt), and the request that refills the slot
    static constexpr int NSLOT = SE_NQ;
    DEVFN void take(int d, float (&x)[8], int) {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = q[d][j];
    }
    DEVFN void refill(int d, int r0) { load(q[d], r0); }
};


// The same queue for a packed wave (two envs per wave: the tile differs between the halves, so no wave-uniform descriptor): each
// lane walks its own column of its own tile through an ordinary global pointer.
template <int SE_NQ>
struct SeStreamLane {
    float q[SE_NQ][8];
    const float *col;              // tile + u (RB-major) / tile + 4 u (RB-quad-major)
    int U, R;
    bool quad;
    static constexpr int NSLOT = SE_NQ;
    DEVFN void load(float (&dst)[8], int r0)
    {
        if (quad) {
            const int nq = (R + 3) >> 2, q0 = r0 >> 2, q1 = q0 + 1 < nq ? q0 + 1 : nq - 1;
#if RANENV_CACHE_HINTS
            const se_v4f a = __builtin_nontemporal_load((const se_v4f *)(col + (size_t)q0 * U * 4)), b = __builtin_nontemporal_load((const se_v4f *)(col + (size_t)q1 * U * 4));
#else
            const se_v4f a = *(const se_v4f *)(col + (size_t)q0 * U * 4), b = *(const se_v4f *)(col + (size_t)q1 * U * 4);
#endif
            dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { const int r = r0 + j < R ? r0 + j : R - 1; dst[j] = col[(size_t)r * U]; }
    }
    DEVFN void init(const float *tile, int U_, int u, int R_, bool quad_ = false)
    {
        quad = quad_;
        col = tile + (quad ? 4 * u : u); U = U_; R = R_;
#pragma unroll
        for (int d = 0; d < SE_NQ; d++) if (d * 8 < R) load(q[d], d * 8);
    }
    DEVFN void take(int d, float (&x)[8], int) {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = q[d][j];
    }
    DEVFN void refill(int d, int r0) { load(q[d], r0); }
};

// Sums of one row: `full` over all R RBs, `part` over the RBs selected by in(r).
// Accumulators start at 0.0 instead of being initialised with the leaf's first group: 0.0 + x == x
// exactly, so the result is numpy's bit for bit while the loop body stays branch-free; the only
// control flow per 8-group is one wave-uniform "leaf finished?" test.  Only the row's last leaf can
// have a tail (R mod 8 elements); it is added sequentially after the loop, as numpy does.
// `after_issue` runs once, before the last turn of the queue (no load is requested in that turn): what the caller
// loads there completes behind the tile (loads retire in order) while the last groups are being summed, and needs
// no register during the rest of the stream.
// PE (RANENV_F_SCALE_PER_ELEMENT): `part` = sum of (sched * se) * scale with every product rounded on its own before it is added, as
// np.sum(sched * se * (BW / R)) would; without it the caller scales the sum (-ffp-contract=off: the product below is not fused).
template <bool PE = false, typename Src, typename InFn, typename Hook>
DEVFN void row_sums(Src &st, int R, InFn in, double &full, double &part, Hook after_issue, const double scale = 1.0)
{
    constexpr int SE_NQ = Src::NSLOT;
    const RowPlan pl = make_row_plan(R);
    const int tail = R & 7, G = R >> 3;
    const int GT = G + (tail > 0 ? 1 : 0);                   // groups requested in all (the partial one included)
    double f[8], g[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { f[j] = 0.0; g[j] = 0.0; }
    double fr = 0.0, gr = 0.0, lf = 0.0, lg = 0.0, rf = 0.0, rg = 0.0;
    int leaf = 0, left_in_leaf = pl.len0 >> 3;            // wave-uniform cursor
    // fold a finished leaf into its half of the top-level split (first + second, in that order)
    auto fold = [&](int k) {
        const bool left = pl.lsplit ? (k < 2) : (k < 1);
        const bool first = pl.lsplit ? (k == 0 || k == 2) : (k <= 1);
        if (left) { if (first) { lf = fr; lg = gr; } else { lf = lf + fr; lg = lg + gr; } }
        else      { if (first) { rf = fr; rg = gr; } else { rf = rf + fr; rg = rg + gr; } }
    };
    auto consume = [&](const float (&x)[8], int r0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            // part += sched * se with sched in {0, 1}: one fused multiply-add is exact here (the product is
            // either x or 0), and cheaper than selecting a 64-bit addend
            const double d = (double)x[j];
            f[j] += d;
#if RANENV_DIAG != 11      /* ablation 11: the full sum alone (what a stream costs without the masked half) */
            g[j] = fma(PE ? d * scale : d, in(r0 + j) ? 1.0 : 0.0, g[j]);
#endif
        }
        if (--left_in_leaf == 0) {
            fr = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
            gr = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
#pragma unroll
            for (int j = 0; j < 8; j++) { f[j] = 0.0; g[j] = 0.0; }
            if (!(leaf == pl.n_leaves - 1 && tail > 0)) fold(leaf);
            leaf += 1;
            // arithmetic select (scalar ALU); an if-chain here gets turned into a stack table
            left_in_leaf = ((leaf == 1) * pl.len1 + (leaf == 2) * pl.len2 + (leaf == 3) * pl.len3) >> 3;
        }
    };
    auto add_tail = [&](const float (&x)[8], int r0) {
        if (G == 0) { fr = 0.0; gr = 0.0; }                  // n < 8: numpy's plain loop from 0.0
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (j < tail) {
                const double d = (double)x[j];
                fr += d;
                gr = fma(PE ? d * scale : d, in(r0 + j) ? 1.0 : 0.0, gr);
            }
        }
        fold(pl.n_leaves - 1);
    };
    auto pass = [&](int gi) {          // one turn of the queue: slot d holds group gi + d
#pragma unroll
        for (int d = 0; d < SE_NQ; d++) {
            if (gi + d < G) {
                float x[8];
                const int behind = GT - 1 - (gi + d);
                st.take(d, x, behind < SE_NQ - 1 ? behind : SE_NQ - 1);
                consume(x, (gi + d) * 8);
                if ((gi + d + SE_NQ) * 8 < R) st.refill(d, (gi + d + SE_NQ) * 8);
            }
        }
    };
    const int last = G > 0 ? ((G - 1) / SE_NQ) * SE_NQ : 0;      // first group of the last turn
#pragma unroll 1
    for (int gi = 0; gi < last; gi += SE_NQ) pass(gi);
    after_issue();
    if (G > 0) pass(last);
    if (tail > 0) {
        const int m = G % SE_NQ;
#pragma unroll
        for (int d = 0; d < SE_NQ; d++) if (m == d) { float x[8]; st.take(d, x, 0); add_tail(x, G * 8); }
    }
    if (pl.n_leaves == 1) { full = lf; part = lg; return; }
    full = lf + rf; part = lg + rg;
}

// ---------------------------------------------------------------------------------------------
// SE gather (ranenv_set_se_mode GATHER): the masked sum alone, from a UE-major copy of the tile.
// Every consumer of a tile except UEs.step reads only np.mean over all RBs per UE (agents/ib_sched.py:110-116,146-157,
// agents/common.py:567-573,648-654): a function of the tile alone, exogenous like the tile (results/gen_results.py:1587-1635
// checks that), so it is computed once per pooled tile (se_mean_pool) instead of once per env and TTI.  What is left per TTI
// is sum_r sched[u,r] * SE[u,r]: each RB belongs to one UE, so an env touches R elements, not U x R.
// Lane u owns [s, s + c) of row u (element r at byte offset row + r * 4, rows padded to a multiple of 8 floats) and walks
// the aligned 8-groups its range touches.  The result is, bit for bit, what row_sums gives for `part`: numpy's order puts
// element r into accumulator (r - leaf start) mod 8 of its leaf, an element outside the range adds +0.0 there (x + 0.0 == x
// exactly), so groups without an element of the range can be skipped; the accumulator tree, the sequential tail and the
// folding of the leaves are the same expressions as in row_sums.
// Loads: two 16-byte buffer loads per group with the whole offset in the VGPR (range-checked: a lane that has no group
// left gets an offset past the descriptor and reads 0 without touching memory), two groups in flight per lane.
// ---------------------------------------------------------------------------------------------
template <int PACK = 1, int DEPTH = 2, bool PE = false>     // DEPTH: 8-RB groups in flight per lane (1: the packed one-TTI build, which has no register to spare)
DEVFN double gather_part(const float *tile, int tile_bytes, int row_bytes_off, int R, unsigned s, unsigned c, const double scale = 1.0)
{
    constexpr int OOB = 0x7ffffff0;
    const RowPlan pl = make_row_plan(R);
    const int tail = R & 7;
    typedef float v4f __attribute__((ext_vector_type(4)));
    // (a packed wave's halves read different tiles: per-lane pointers and a predicate instead of a wave-uniform descriptor)
    __amdgpu_buffer_rsrc_t rsrc;
    if constexpr (PACK == 1) rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, tile_bytes, 0x00020000);
    auto ld8 = [&](float (&q)[8], int off) {
        if constexpr (PACK == 1) {
            const v4f a = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, SE_AUX_NT));
            const v4f b = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off < OOB ? off + 16 : OOB, 0, SE_AUX_NT));
            q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
        } else {
            v4f a = {0.0f, 0.0f, 0.0f, 0.0f}, b = a;
            if (off < tile_bytes) {
                const v4f *src = (const v4f *)((const char *)tile + off);
                a = src[0]; b = src[1];
            }
            q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w; q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
        }
    };
    auto in = [=](int r) { return ((unsigned)r - s) < c; };
    double lg = 0.0, rg = 0.0;
    int base = 0;
#pragma unroll 1
    for (int k = 0; k < pl.n_leaves; k++) {                       // wave-uniform
        const int len = (k == 0) * pl.len0 + (k == 1) * pl.len1 + (k == 2) * pl.len2 + (k == 3) * pl.len3;
        const int end = base + (len & ~7);                        // first RB behind the leaf's full groups
        const bool last = k == pl.n_leaves - 1;
        const int lo = (int)s > base ? (int)s : base, hi = (int)(s + c) < end ? (int)(s + c) : end;
        int g0 = 0, ng = 0;                                       // this lane's groups inside the leaf
        if (lo < hi) { g0 = (lo - base) >> 3; ng = ((hi - 1 - base) >> 3) - g0 + 1; }
        const int first = row_bytes_off + (base + g0 * 8) * 4;
        float q0[8], q1[8];
        ld8(q0, 0 < ng ? first : OOB);
        if constexpr (DEPTH == 2) ld8(q1, 1 < ng ? first + 32 : OOB);
        double g[8];
#pragma unroll
        for (int j = 0; j < 8; j++) g[j] = 0.0;
        auto consume = [&](const float (&x)[8], int r0) {
#pragma unroll
            for (int j = 0; j < 8; j++) g[j] = fma(PE ? (double)x[j] * scale : (double)x[j], in(r0 + j) ? 1.0 : 0.0, g[j]);
        };
        if constexpr (DEPTH == 2) {
#pragma unroll 1
            for (int i = 0; __builtin_amdgcn_ballot_w64(i < ng) != 0; i += 2) {
                // a lane past its last group consumes zeros at RBs outside its range: +0.0
                consume(q0, base + (g0 + i) * 8);
                ld8(q0, i + 2 < ng ? first + (i + 2) * 32 : OOB);
                consume(q1, base + (g0 + i + 1) * 8);
                ld8(q1, i + 3 < ng ? first + (i + 3) * 32 : OOB);
            }
        } else {
#pragma unroll 1
            for (int i = 0; __builtin_amdgcn_ballot_w64(i < ng) != 0; i += 1) {
                consume(q0, base + (g0 + i) * 8);
                ld8(q0, i + 1 < ng ? first + (i + 1) * 32 : OOB);
            }
        }
        double gr = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
        if (last && tail > 0) {
            // the row's last R mod 8 RBs, added one after the other behind the tree; few ranges reach them, and a wave
            // none of whose lanes does skips the load (the adds would all be + 0.0)
            const bool want_tail = (int)(s + c) > end && c > 0;
            if (__builtin_amdgcn_ballot_w64(want_tail) != 0) {
                ld8(q0, want_tail ? row_bytes_off + end * 4 : OOB);
#pragma unroll
                for (int j = 0; j < 7; j++)
                    if (j < tail) gr = fma(PE ? (double)q0[j] * scale : (double)q0[j], in(end + j) ? 1.0 : 0.0, gr);
            }
        }
        const bool left = pl.lsplit ? (k < 2) : (k < 1);
        const bool first_of_half = pl.lsplit ? (k == 0 || k == 2) : (k <= 1);
        if (left) lg = first_of_half ? gr : lg + gr;
        else      rg = first_of_half ? gr : rg + gr;
        base += len;
    }
    return pl.n_leaves == 1 ? lg : lg + rg;
}

// ---------------------------------------------------------------------------------------------
// The same masked sum with the loads SPREAD over the wave's lanes (round 6).  Only ~16 of an env's UEs get RBs at a TTI, and the one
// with the most walks 5-6 groups one memory round trip after the other while the other lanes wait: gather_part's phase is ~4.5 dependent
// round trips per TTI (tools/rb_range_stats.py).  Here every (owner lane, 8-RB group) pair that gather_part would load is a work item;
// the items of the wave (~33 at the headline size: every RB belongs to one UE) are numbered by a wave scan, item i is loaded and masked by
// LANE i -- all of them in ONE round trip --, and the owners then add their items' eight values in group order, fetched from the helper
// lanes by ds_bpermute (the LDS crossbar, no memory).  The additions, their order, the tree per leaf, the sequential tail and the folding of
// the leaves are gather_part's: bit for bit the same sum.  `desc`: 64 16-bit item descriptors of this wave in LDS (owner lane | group << 6 |
// tail << 12).  Returns false -- nothing done -- when the wave has more than 63 items (lane 63 must stay a lane of zeros): the caller falls
// back to gather_part.
// ---------------------------------------------------------------------------------------------
DEVFN int wave_excl_scan_i32(int x, int &total)      // exclusive prefix sum over the 64 lanes (all active); total = the wave's sum
{
    const int incl = row16_scan(x);                  // within the 16-lane rows
    const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47),
              t3 = __builtin_amdgcn_readlane(incl, 63);
    const int row = (int)(threadIdx.x & 63u) >> 4;
    const int before = (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
    total = t0 + t1 + t2 + t3;
    return incl - x + before;
}
DEVFN double bperm_f64(double v, int src_lane)       // lane src_lane's v (every lane executes this)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)b), hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
template <bool PE>
DEVFN bool gather_part_spread(const float *tile, int tile_bytes, int row_bytes_off, int R, unsigned s, unsigned c, const double scale,
                              unsigned short *desc, double &part)
{
    constexpr int OOB = 0x7ffffff0;
    typedef float v4f __attribute__((ext_vector_type(4)));
    const RowPlan pl = make_row_plan(R);
    const int tail = R & 7, lane = (int)(threadIdx.x & 63u);
    auto leaf_len = [&](int k) { return (k == 0) * pl.len0 + (k == 1) * pl.len1 + (k == 2) * pl.len2 + (k == 3) * pl.len3; };
    // this lane's groups inside leaf [base, base + len): first group (absolute index) and how many -- as gather_part counts them
    auto my_groups = [&](int base, int len, int &g0, int &ng) {
        const int end = base + (len & ~7);
        const int lo = (int)s > base ? (int)s : base, hi = (int)(s + c) < end ? (int)(s + c) : end;
        g0 = 0; ng = 0;
        if (lo < hi) { g0 = lo >> 3; ng = ((hi - 1) >> 3) - g0 + 1; }
    };
    const int row_end = R & ~7;                                     // first RB of the row's tail group
    const bool want_tail = tail > 0 && (int)(s + c) > row_end && c > 0;
    // 1. count and number the items
    int n_items = want_tail ? 1 : 0;
    {
        int base = 0;
#pragma unroll 1
        for (int k = 0; k < pl.n_leaves; k++) { int g0, ng; my_groups(base, leaf_len(k), g0, ng); n_items += ng; base += leaf_len(k); }
    }
    int total = 0;
    const int off = wave_excl_scan_i32(n_items, total);
    if (total > 63) return false;                                   // (wave-uniform)
    // 2. the owners describe their items, in the order they will add them: leaf by leaf, groups ascending, the tail last
    {
        int base = 0, at = off;
#pragma unroll 1
        for (int k = 0; k < pl.n_leaves; k++) {
            int g0, ng; my_groups(base, leaf_len(k), g0, ng);
#pragma unroll 1
            for (int m = 0; __builtin_amdgcn_ballot_w64(m < ng) != 0; m++)
                if (m < ng) desc[at + m] = (unsigned short)(lane | ((g0 + m) << 6));
            at += ng; base += leaf_len(k);
        }
        if (want_tail) desc[at] = (unsigned short)(lane | ((row_end >> 3) << 6) | (1 << 12));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (this wave's own area: an LDS wait, no barrier)
    // 3. lane i loads and masks item i
    double m[8];
    {
        const bool have = lane < total;
        const int d = have ? (int)desc[lane] : 0;
        const int owner = d & 63, gi = (d >> 6) & 63;
        const unsigned os = (unsigned)__builtin_amdgcn_ds_bpermute(owner << 2, (int)s), oc = (unsigned)__builtin_amdgcn_ds_bpermute(owner << 2, (int)c);
        const int orow = __builtin_amdgcn_ds_bpermute(owner << 2, row_bytes_off);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, tile_bytes, 0x00020000);
        const int o = have ? orow + gi * 32 : OOB;
        const v4f a = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, SE_AUX_NT));
        const v4f b = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o < OOB ? o + 16 : OOB, 0, SE_AUX_NT));
        const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const unsigned r0 = (unsigned)(gi * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const bool in = have && ((r0 + (unsigned)j - os) < oc);
            // (the product with 0 / 1 and the sum with +0.0 are exact: the owner's `g + m` below is gather_part's fma(x, mask, g) bit for bit)
            m[j] = fma(PE ? (double)x[j] * scale : (double)x[j], in ? 1.0 : 0.0, 0.0);
        }
    }
    // 4. the owners add their items in order (a lane past its last item reads lane 63: zeros, since total <= 63)
    double lg = 0.0, rg = 0.0;
    int base = 0, at = off;
#pragma unroll 1
    for (int k = 0; k < pl.n_leaves; k++) {
        int g0, ng; my_groups(base, leaf_len(k), g0, ng);
        double g[8];
#pragma unroll
        for (int j = 0; j < 8; j++) g[j] = 0.0;
#pragma unroll 1
        for (int it = 0; __builtin_amdgcn_ballot_w64(it < ng) != 0; it++) {
            const int src = it < ng ? at + it : 63;
#pragma unroll
            for (int j = 0; j < 8; j++) g[j] += bperm_f64(m[j], src);
        }
        double gr = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
        at += ng;
        if (k == pl.n_leaves - 1 && tail > 0) {
            if (__builtin_amdgcn_ballot_w64(want_tail) != 0) {
                const int src = want_tail ? at : 63;
#pragma unroll
                for (int j = 0; j < 7; j++)
                    if (j < tail) gr += bperm_f64(m[j], src);
            }
        }
        const bool left = pl.lsplit ? (k < 2) : (k < 1);
        const bool first_of_half = pl.lsplit ? (k == 0 || k == 2) : (k <= 1);
        if (left) lg = first_of_half ? gr : lg + gr;
        else      rg = first_of_half ? gr : rg + gr;
        base += leaf_len(k);
    }
    part = pl.n_leaves == 1 ? lg : lg + rg;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Counter-based random numbers: Philox-4x32-10 (Salmon et al., SC'11).  One call = 128 random bits that depend
// only on (key, counter): the exogenous inputs of an env never depend on what the agent did
// (results/gen_results.py:1587-1635 checks exactly that across agents).
// ---------------------------------------------------------------------------------------------
DEVFN void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Poisson draw by table inversion: u = 64 random bits, result = smallest k with u < cdf[k] (255 at most); the guide table
// (indexed by the top 6 bits of u) gives a k at or below the answer.  The walk from there looks at eight entries per turn, requested
// together: one memory round trip per turn instead of one per entry (a wave walks as long as its slowest lane, and every
// round trip is 1-2 us of the UE step under load).
DEVFN int poisson_draw(const unsigned long long *cdf, const uint8_t *guide, unsigned long long u)
{
    int k = guide[u >> 58];
    constexpr int WIN = 8;
    for (;;) {
        unsigned long long c[WIN];
#pragma unroll
        for (int j = 0; j < WIN; j++) c[j] = cdf[k + j < 255 ? k + j : 255];
        int adv = 0;
        bool on = true;
#pragma unroll
        for (int j = 0; j < WIN; j++) { on = on && k + j < 255 && c[j] <= u; adv += on ? 1 : 0; }
        k += adv;
        if (adv < WIN) break;
    }
    return k;
}

}  // namespace
