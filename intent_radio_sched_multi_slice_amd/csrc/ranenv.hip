// ranenv.hip -- MI355X (gfx950) implementation of the C ABI in include/ranenv.h.
//
// One workgroup steps one environment for one TTI.  Thread u of the workgroup owns UE u for
// the whole step: it streams UE u's spectral-efficiency row, updates UE u's packet queue and
// computes UE u's intent drift, all in registers.  Only slice-level work (inter-slice RBG
// split, intra-slice RR/PF/MT, per-slice means, reward) goes through LDS.
//
// HBM layout (B envs, S slices, U UEs, R RBs, L = max_age_cap+1, D = hist_depth):
//   SE pool        float32 [tile][R][U]   RB-major: at RB r the U lanes of a workgroup read U
//                                          consecutive floats -> coalesced 4-byte loads, and a
//                                          lane walks its own row r = 0..R-1 in numpy's
//                                          pairwise-summation order with 8 accumulators.
//   traffic pool   int32   [row][U]
//   per-UE state   [B][U]  queue_pkts i32, queue_age_sum i64, front i32, front_rem i32, fifo i32,
//                          win_sent i64, win_dropped i64, se_mean f64
//   age ring       int2    [B][L][U]      circular list of (arrival TTI, packets) per UE; the queue
//                                          is FIFO, so (head entry, its remainder, the list)
//                                          describe exactly the age histogram Buffer keeps
//                                          (oracle/ranenv_oracle.c) while a step touches only the
//                                          inserted / expired / drained entries.
//   10-TTI window  int32   [B][D][U] x2   pkt_effective_thr and dropped_pkts of the last D pushes
//   scenario pool  small SoA tables, shared by all envs replaying a scenario (L2 resident)
//
// Reference behaviour restated here (file:line under lasseufpa/intent_radio_sched_multi_slice):
//   agents/ib_sched.py:223-349 action_format, :63-204 obs_space_format, :206-221 calculate_reward
//   agents/common.py:442-505 scores_to_rbs/round_int_equal_sum, :508-701 RR/PF/MT,
//   :9-340 get_metric_value/intent_drift_calc, :343-378 calculate_slice_ue_obs, :381-439 reward
//   agents/marr.py:40-47, agents/mapf.py:41-111 baseline policies
//   sixg_radio_mgmt UEs/Buffer (un-vendored): normative restatement in oracle/ranenv_oracle.c
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ranenv.h"

#define DEVFN __device__ __forceinline__

#ifndef RANENV_DIAG
#define RANENV_DIAG 0   /* diagnostic builds only: 1-4 skip a phase, 9 stamps s_memtime per phase */
#endif

namespace {

enum { MODE_STEP = 0, MODE_DENSE = 1, MODE_RESET = 2 };

// ---------------------------------------------------------------------------------------------
// kernel parameters
// ---------------------------------------------------------------------------------------------
struct Tables {  // scenario pool on the device, rows of [n_scenarios]
    int32_t *slice_i32;  // [NS][S][8] active, has_req, nues, buffer_size, buffer_latency, message_size, nparams, sorted
    double  *slice_f64;  // [NS][S][2] priority, traffic
    int32_t *param_i32;  // [NS][S][3][2] metric, op
    double  *param_f64;  // [NS][S][3]
    int32_t *slice_ues;  // [NS][S][Us]
    int32_t *ue_slice, *ue_pos, *ue_pkt_size, *ue_max_pkts, *ue_max_age;  // [NS][U]
};

struct State {
    int32_t *queue_pkts; int64_t *queue_age_sum; int32_t *front; int32_t *front_rem; int32_t *fifo;
    int64_t *win_sent; int64_t *win_dropped; double *se_mean;
    int2 *age_ring; int32_t *ring_sent; int32_t *ring_drop;
    int32_t *hist_len; int32_t *n_push; int32_t *step_no; int32_t *se_pos; int32_t *trf_pos;
    int32_t *pkt_incoming, *pkt_throughputs, *pkt_effective_thr, *dropped_pkts, *rb_start, *rb_count;
    int8_t *mask_inter, *mask_intra; double *policy_scores;
};

struct KP {
    int B, S, U, R, G, Us, D, L, max_steps, flags, policy, fixed_intra;
    double bw_hz, bw_per_rb, over, norm_traffic, norm_ues, norm_se;
    Tables tab;
    State st;
    const ranenv_episode *episodes;
    const float *se_pool; long long se_stride;
    const int32_t *trf_pool;
    // per-call inputs (may be null)
    const uint8_t *env_mask;
    const double *scores; const uint8_t *intra; const double *traffic_bits; const float *se_tiles;
    const uint8_t *dense;
    // outputs (may be null)
    float *obs_inter; float *obs_intra; double *reward; uint8_t *done;
};

// LDS carve-up, shared by host (size) and device (offsets). All sizes in bytes, doubles first.
// Rows that the unrolled 16-wide readers may over-read are padded (PAD entries).
struct LdsLayout {
    int d_occ, d_sem, d_hmean, d_semn, d_occn, d_part, d_drift, c_a, c_b, c_c, c_d, d_scores, d_tmp,
        d_slvals, d_slflags, d_par, d_slf, i_slice, i_pos, i_pkt, i_maxp, i_maxage, i_start, i_count, i_sl,
        i_slues, i_par, i_rbs, i_off, i_cnt, i_sel, i_nz, i_choice, i_misc, f_obs_inter, f_obs_intra, total;
};
constexpr int PAD = 16;

__host__ __device__ inline LdsLayout make_layout(int S, int U, int Us)
{
    // Regions whose lifetimes do not overlap share storage:
    //   d_sem (previous mean SE, read by P3)     <-> d_semn (new mean SE, written by P4)
    //   d_hmean (window mean, read by P1/P3)     <-> d_part (allocated-RB SE sum, written by P4)
    //   c_a|c_b|c_c (P1/P3 per-slice rows)       <-> d_drift (zeroed after P4, written by P5)
    //   i_pkt|i_maxp|i_maxage|i_start (<= P5)    <-> f_obs_inter|f_obs_intra (P6)
    LdsLayout l;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o += (bytes + 7) & ~7; return r; };
    l.d_occ = take(8 * U);  l.d_sem = take(8 * U);  l.d_hmean = take(8 * U); l.d_occn = take(8 * U);
    l.d_semn = l.d_sem; l.d_part = l.d_hmean;
    l.c_a = take(8 * (S * Us + PAD)); l.c_b = take(8 * (S * Us + PAD)); l.c_c = take(8 * (S * Us + PAD));
    l.d_drift = l.c_a;                               // 8*(3*S*Us + 3*PAD) bytes, exactly c_a..c_c
    l.c_d = take(8 * (S * Us + PAD));
    l.d_scores = take(8 * (S + PAD)); l.d_tmp = take(8 * (5 * S + PAD));
    l.d_slvals = take(8 * 3 * S); l.d_slflags = take(8 * 3 * S);
    l.d_par = take(8 * 3 * S); l.d_slf = take(8 * 2 * S);
    l.i_slice = take(4 * U); l.i_pos = take(4 * U); l.i_count = take(4 * U);
    const int ue_tail = 4 * 4 * U, obs = 4 * 10 * S + 4 * S * (2 * Us + 9);
    const int shared = take(ue_tail > obs ? ue_tail : obs);
    l.i_pkt = shared; l.i_maxp = shared + 4 * U; l.i_maxage = shared + 8 * U; l.i_start = shared + 12 * U;
    l.f_obs_inter = shared; l.f_obs_intra = shared + 4 * 10 * S;
    l.i_sl = take(4 * 8 * S); l.i_slues = take(4 * (S * Us + PAD)); l.i_par = take(4 * 6 * S);
    l.i_rbs = take(4 * (S + PAD)); l.i_off = take(4 * (S + PAD)); l.i_cnt = take(4 * (S * Us + PAD));
    l.i_sel = take(4 * (S * Us + PAD)); l.i_nz = take(4 * S); l.i_choice = take(4 * S);
    l.i_misc = take(4 * 8);
    l.total = o;
    return l;
}

// ---------------------------------------------------------------------------------------------
// numpy arithmetic on the device
// ---------------------------------------------------------------------------------------------
DEVFN bool d_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }

// Visit i = 0..n-1 in chunks of 16 with the body fully unrolled: the LDS reads of a chunk are
// independent and issue back to back (one latency per chunk instead of one per element).  The body
// gets (i, valid); it may read element i unconditionally (arrays are padded by PAD) and must ignore
// the value when !valid.
template <typename F>
DEVFN void for16(int n, F body)
{
    for (int i0 = 0; i0 < n; i0 += 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) body(i0 + j, i0 + j < n);
    }
}

// numpy pairwise_sum (one leaf, n <= 128) over strided doubles in LDS.  n <= 16 is the hot case
// (slices, UEs of a slice): all 16 reads are issued up front, missing elements read as +0.0, which
// makes the three numpy shapes (n < 8 plain loop; 8 <= n < 16 tree of 8 + sequential tail; n == 16
// tree of 8 pair sums) plain expressions (x + 0.0 == x exactly).
DEVFN double np_sum_lds(const double *a, int n, int stride)
{
    if (n <= 16) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; j++) { const double v = a[j * stride]; x[j] = j < n ? v : 0.0; }
        const double seq = ((((((x[0] + x[1]) + x[2]) + x[3]) + x[4]) + x[5]) + x[6]) + x[7];
        double t8 = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
#pragma unroll
        for (int j = 8; j < 15; j++) t8 += x[j];
        const double t16 = (((x[0] + x[8]) + (x[1] + x[9])) + ((x[2] + x[10]) + (x[3] + x[11]))) +
                           (((x[4] + x[12]) + (x[5] + x[13])) + ((x[6] + x[14]) + (x[7] + x[15])));
        return n < 8 ? seq : (n < 16 ? t8 : t16);
    }
    double r0 = a[0], r1 = a[stride], r2 = a[2 * stride], r3 = a[3 * stride];
    double r4 = a[4 * stride], r5 = a[5 * stride], r6 = a[6 * stride], r7 = a[7 * stride];
    int i;
    const int m = n - (n % 8);
    for (i = 8; i < m; i += 8) {
        const double *q = a + i * stride;
        r0 += q[0]; r1 += q[stride]; r2 += q[2 * stride]; r3 += q[3 * stride];
        r4 += q[4 * stride]; r5 += q[5 * stride]; r6 += q[6 * stride]; r7 += q[7 * stride];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i * stride];
    return res;
}

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

// ---------------------------------------------------------------------------------------------
// SE row reduction: software-pipelined stream + numpy's pairwise order
// ---------------------------------------------------------------------------------------------
// Lane u reduces row u of an RB-major tile (element r at byte offset r*U*4 + u*4).  Loads go through
// a wave-uniform buffer descriptor: the row offset is a scalar, the lane offset one VGPR, so a load
// costs no vector address arithmetic.  Three groups of 8 loads rotate through named registers
// (qa/qb/qc, no moves), i.e. 16-24 loads per lane stay in flight while a group is being summed.
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

struct SeStream {
    float qa[8], qb[8], qc[8];
    __amdgpu_buffer_rsrc_t rsrc;   // wave-uniform descriptor of the tile (SGPRs)
    int voff, row_bytes, r_last;

    DEVFN void load(float (&dst)[8], int r0)
    {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int r = r0 + j;
            const int rr = r < r_last ? r : r_last;                       // scalar clamp: always in bounds
            dst[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, rr * row_bytes, 0));
        }
    }
    DEVFN void init(const float *tile, int U, int u, int R)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(tile), 0, U * R * 4, 0x00020000);
        voff = u * 4; row_bytes = U * 4; r_last = R - 1;
        load(qa, 0);
        if (R > 8) load(qb, 8);
        if (R > 16) load(qc, 16);
    }
};

// Sums of one row: `full` over all R RBs, `part` over the RBs selected by in(r).
// Accumulators start at 0.0 instead of being initialised with the leaf's first group: 0.0 + x == x
// exactly, so the result is numpy's bit for bit while the loop body stays branch-free; the only
// control flow per 8-group is one wave-uniform "leaf finished?" test.  Only the row's last leaf can
// have a tail (R mod 8 elements); it is added sequentially after the loop, as numpy does.
template <typename InFn>
DEVFN void row_sums(SeStream &st, int R, InFn in, double &full, double &part)
{
    const RowPlan pl = make_row_plan(R);
    const int tail = R & 7, G = R >> 3;
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
            const float xs = in(r0 + j) ? x[j] : 0.0f;
            f[j] += (double)x[j];
            g[j] += (double)xs;
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
                const float xs = in(r0 + j) ? x[j] : 0.0f;
                fr += (double)x[j];
                gr += (double)xs;
            }
        }
        fold(pl.n_leaves - 1);
    };
#pragma unroll 1
    for (int gi = 0; gi < G; gi += 3) {
        const int r0 = gi * 8;
        consume(st.qa, r0);
        if (r0 + 24 < R) st.load(st.qa, r0 + 24);
        if (gi + 1 < G) {
            consume(st.qb, r0 + 8);
            if (r0 + 32 < R) st.load(st.qb, r0 + 32);
        }
        if (gi + 2 < G) {
            consume(st.qc, r0 + 16);
            if (r0 + 40 < R) st.load(st.qc, r0 + 40);
        }
    }
    if (tail > 0) {
        const int m3 = G % 3;
        if (m3 == 0) add_tail(st.qa, G * 8);
        else if (m3 == 1) add_tail(st.qb, G * 8);
        else add_tail(st.qc, G * 8);
    }
    if (pl.n_leaves == 1) { full = lf; part = lg; return; }
    full = lf + rf; part = lg + rg;
}

// ---------------------------------------------------------------------------------------------
// the step kernel
// ---------------------------------------------------------------------------------------------
#define RANENV_STAMP(k)                                                                        \
    do {                                                                                       \
        if (RANENV_DIAG == 9 && MODE == MODE_STEP) {                                           \
            unsigned long long ts_;                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_)::"memory");       \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            if (threadIdx.x == 0) stamps[(k)] = ts_;                                           \
        }                                                                                      \
    } while (0)

// One wavefront (64 lanes) steps one environment: every exchange between lanes goes through LDS
// inside the wave, so the phase boundaries below cost an LDS wait, not a workgroup barrier, and a CU
// keeps twice as many environments in flight as with a two-wave workgroup.  UE u is handled by lane
// u mod 64 in pass u / 64 of every per-UE phase; slice s by lane s of the per-slice phases.
constexpr int WAVE = 64;

// PASSES > 0: number of 64-UE passes known at compile time (loops fully unrolled, so the passes'
// dependency chains interleave); PASSES == 0: generic run-time loop.
#define FOR_PASS _Pragma("unroll") for (int pass = 0; pass < (PASSES > 0 ? PASSES : passes); pass++)

template <int MODE, int PASSES>
__global__ void __launch_bounds__(WAVE) ranenv_kernel(const KP p)
{
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    if (p.env_mask != nullptr && p.env_mask[b] == 0) return;  // uniform per workgroup
    unsigned long long stamps[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    RANENV_STAMP(0);

    const int S = p.S, U = p.U, R = p.R, Us = p.Us, D = p.D;
    const int W = 2 * Us + 9;
    const int passes = PASSES > 0 ? PASSES : (U + WAVE - 1) / WAVE;
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsLayout lo = make_layout(S, U, Us);
    double *d_occ = (double *)(smem + lo.d_occ), *d_sem = (double *)(smem + lo.d_sem);
    double *d_hmean = (double *)(smem + lo.d_hmean), *d_semn = (double *)(smem + lo.d_semn);
    double *d_occn = (double *)(smem + lo.d_occn), *d_part = (double *)(smem + lo.d_part);
    double *d_drift = (double *)(smem + lo.d_drift);
    double *c_a = (double *)(smem + lo.c_a), *c_b = (double *)(smem + lo.c_b);
    double *c_c = (double *)(smem + lo.c_c), *c_d = (double *)(smem + lo.c_d);
    double *d_scores = (double *)(smem + lo.d_scores), *d_tmp = (double *)(smem + lo.d_tmp);
    double *d_slvals = (double *)(smem + lo.d_slvals), *d_slflags = (double *)(smem + lo.d_slflags);
    double *d_par = (double *)(smem + lo.d_par), *d_slf = (double *)(smem + lo.d_slf);
    int *i_slice = (int *)(smem + lo.i_slice), *i_pos = (int *)(smem + lo.i_pos), *i_pkt = (int *)(smem + lo.i_pkt);
    int *i_maxp = (int *)(smem + lo.i_maxp), *i_maxage = (int *)(smem + lo.i_maxage);
    int *i_start = (int *)(smem + lo.i_start), *i_count = (int *)(smem + lo.i_count);
    int *i_sl = (int *)(smem + lo.i_sl), *i_slues = (int *)(smem + lo.i_slues), *i_par = (int *)(smem + lo.i_par);
    int *i_rbs = (int *)(smem + lo.i_rbs), *i_off = (int *)(smem + lo.i_off), *i_cnt = (int *)(smem + lo.i_cnt);
    int *i_sel = (int *)(smem + lo.i_sel), *i_nz = (int *)(smem + lo.i_nz), *i_choice = (int *)(smem + lo.i_choice);
    int *i_misc = (int *)(smem + lo.i_misc);
    float *f_obs_inter = (float *)(smem + lo.f_obs_inter), *f_obs_intra = (float *)(smem + lo.f_obs_intra);

    // ---- P0: per-env scalars, per-UE state, scenario tables -------------------------------------
    // Everything read here is the same for the whole wave: pin it to scalar registers so the
    // SE loads below use the saddr + 32-bit lane-offset form instead of per-lane 64-bit addresses.
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [](long long v) {
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi32 << 32) | lo32);
    };
    ranenv_episode ep = p.episodes[b];
    ep.scenario = uni(ep.scenario); ep.se_len = uni(ep.se_len); ep.se_offset = uni(ep.se_offset);
    ep.trf_len = uni(ep.trf_len); ep.trf_offset = uni(ep.trf_offset);
    ep.se_base = uni64(ep.se_base); ep.trf_base = uni64(ep.trf_base);
    const int sc = ep.scenario;
    const int step = (MODE == MODE_RESET) ? 0 : uni(p.st.step_no[b]);
    int hlen = uni(p.st.hist_len[b]);
    const int npush = uni(p.st.n_push[b]);                    // kept in [0, D)
    // position inside the env's SE / traffic trace: offset at reset, +1 (wrapping) per TTI
    const int se_pos = (MODE == MODE_RESET) ? ep.se_offset : uni(p.st.se_pos[b]);
    const int trf_pos = (MODE == MODE_RESET) ? ep.trf_offset : uni(p.st.trf_pos[b]);
    const bool clear_hist = MODE == MODE_RESET && (p.flags & RANENV_F_CLEAR_HISTORY_ON_RESET);
    if (clear_hist) hlen = 0;
    const int t = step;
    const int hlen_new = hlen < D ? hlen + 1 : D;

    const float *tile;
    if (p.se_tiles != nullptr) tile = p.se_tiles + (size_t)b * U * R;
    else tile = p.se_pool + (size_t)(ep.se_base + (long long)se_pos) * (size_t)p.se_stride;

    // SE stream of pass 0: its first loads are in flight while the allocation phases run
    SeStream se_a, se_b;
    se_a.init(tile, U, lane < U ? lane : U - 1, R);

    FOR_PASS {
        const int u = pass * WAVE + lane;
        if (u < U) {
            const size_t su = (size_t)b * U + u, tu = (size_t)sc * U + u;
            const int slc = p.tab.ue_slice[tu], pos = p.tab.ue_pos[tu];
            const int pkt = p.tab.ue_pkt_size[tu], maxp = p.tab.ue_max_pkts[tu], maxage = p.tab.ue_max_age[tu];
            int total = 0; long long wsent = 0; double sem_prev = 0.0;
            if (MODE != MODE_RESET) { total = p.st.queue_pkts[su]; sem_prev = p.st.se_mean[su]; }
            if (!clear_hist) wsent = p.st.win_sent[su];
            const double occ = (double)total / (double)maxp;
            const double hm = hlen > 0 ? (double)wsent / (double)hlen : 0.0;
            i_slice[u] = slc; i_pos[u] = pos; i_pkt[u] = pkt; i_maxp[u] = maxp; i_maxage[u] = maxage;
            d_occ[u] = occ; d_sem[u] = sem_prev; d_hmean[u] = hm;
            i_start[u] = 0; i_count[u] = 0;
            if (slc >= 0) { c_a[slc * Us + pos] = occ; c_b[slc * Us + pos] = hm; }   // per-slice rows for MAPF
        }
    }
    if (lane < S) {
        const int s = lane;
        const int32_t *si = p.tab.slice_i32 + ((size_t)sc * S + s) * 8;
#pragma unroll
        for (int k = 0; k < 8; k++) i_sl[s * 8 + k] = si[k];
        d_slf[s * 2 + 0] = p.tab.slice_f64[((size_t)sc * S + s) * 2 + 0];
        d_slf[s * 2 + 1] = p.tab.slice_f64[((size_t)sc * S + s) * 2 + 1];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            i_par[s * 6 + 2 * k + 0] = p.tab.param_i32[(((size_t)sc * S + s) * 3 + k) * 2 + 0];
            i_par[s * 6 + 2 * k + 1] = p.tab.param_i32[(((size_t)sc * S + s) * 3 + k) * 2 + 1];
            d_par[s * 3 + k] = p.tab.param_f64[((size_t)sc * S + s) * 3 + k];
        }
        int choice = p.fixed_intra;
        if (choice == RANENV_INTRA_PER_SLICE) choice = (MODE == MODE_STEP && p.intra) ? (int)p.intra[(size_t)b * S + s] : RANENV_INTRA_RR;
        i_choice[s] = choice;
    }
    for (int i = lane; i < S * Us; i += WAVE) i_slues[i] = p.tab.slice_ues[(size_t)sc * S * Us + i];
    __syncthreads();
    RANENV_STAMP(1);

    // slice table accessors
    auto sl_active = [&](int s) { return i_sl[s * 8 + 0]; };
    auto sl_hasreq = [&](int s) { return i_sl[s * 8 + 1]; };
    auto sl_nues = [&](int s) { return i_sl[s * 8 + 2]; };
    auto sl_bsize = [&](int s) { return i_sl[s * 8 + 3]; };
    auto sl_blat = [&](int s) { return i_sl[s * 8 + 4]; };
    auto sl_msg = [&](int s) { return i_sl[s * 8 + 5]; };
    auto sl_npar = [&](int s) { return i_sl[s * 8 + 6]; };
    auto sl_sorted = [&](int pos) { return i_sl[pos * 8 + 7]; };

    if (MODE == MODE_STEP && RANENV_DIAG != 1) {
        // ---- P1: inter-slice scores: external, MARR (marr.py:40-47) or MAPF (mapf.py:41-111) ----
        const bool ext = p.scores != nullptr;
        if (!ext && p.policy == RANENV_POLICY_MAPF) {
            if (lane < S) {
                const int s = lane;
                double occ_mb = 0.0, thr_mb = 0.0;
                if (sl_active(s)) {
                    const int n = sl_nues(s);
                    const double pkt = (double)sl_msg(s), bmax = (double)sl_bsize(s);
                    occ_mb = ((np_sum_lds(c_a + s * Us, n, 1) / (double)n * bmax) * pkt) / 1e6;
                    thr_mb = ((np_sum_lds(c_b + s * Us, n, 1) / (double)n) * pkt) / 1e6;
                }
                d_tmp[s] = occ_mb; d_tmp[S + s] = thr_mb;
            }
            __syncthreads();
            if (lane < S) {
                const int s = lane;
                double mx = d_tmp[0];
                for16(S, [&](int j, bool ok) { const double v = d_tmp[j]; mx = (ok && v > mx) ? v : mx; });
                double w = d_isclose(d_tmp[S + s], 0.0) ? 2.0 * mx : d_tmp[s] / d_tmp[S + s];
                if (!sl_active(s)) w = 0.0;
                d_tmp[2 * S + s] = w;
            }
            __syncthreads();
            if (lane < S) {
                const double ws = np_sum_lds(d_tmp + 2 * S, S, 1);
                d_scores[lane] = (ws > 0.0 ? d_tmp[2 * S + lane] / ws : 2.0) - 1.0;
            }
        } else if (lane < S) {
            d_scores[lane] = ext ? p.scores[(size_t)b * S + lane] : (sl_nues(lane) > 0 ? 1.0 : -1.0);
        }
        __syncthreads();
        RANENV_STAMP(2);

        // ---- P2: inter-slice RBG split, one lane per slice (ib_sched.py:240-269, ------------------
        //      common.py:442-461 scores_to_rbs, :481-505 round_int_equal_sum)
        double *t_ap1 = d_tmp, *t_assoc = d_tmp + S, *t_v = d_tmp + 2 * S, *t_nz = d_tmp + 3 * S;
        const int T = R / p.G;                                              // floor(R / G), uniform
        double my_a = -1.0, my_v = 0.0;
        if (lane < S) {
            const int s = lane;
            p.st.policy_scores[(size_t)b * S + s] = d_scores[s];
            my_a = sl_active(s) ? d_scores[sl_sorted(s)] : -1.0;             // ib_sched.py:247-255
            t_ap1[s] = my_a + 1.0;
            t_assoc[s] = (double)sl_active(s);
        }
        __syncthreads();
        int any_active = 0;
        for16(S, [&](int j, bool ok) { const double v = t_assoc[j]; any_active += (ok && v != 0.0) ? 1 : 0; });
        if (lane < S && any_active) {
            const double ssum = np_sum_lds(t_ap1, S, 1);
            if (ssum != 0.0) my_v = (double)T * (my_a + 1.0) / ssum;
            else my_v = ((double)T / np_sum_lds(t_assoc, S, 1)) * t_assoc[lane];
            t_v[lane] = my_v;
        }
        __syncthreads();
        // compaction of the non-zero values (:484-485): entry of slice s goes to its rank among them
        int m_nz = 0, my_slot = 0;
        for16(S, [&](int j, bool ok) {
            const double v = t_v[j];
            const int nzf = (ok && any_active && v != 0.0) ? 1 : 0;
            m_nz += nzf; my_slot += (j < lane) ? nzf : 0;
        });
        if (lane < S && any_active && my_v != 0.0) t_nz[my_slot] = my_v;
        __syncthreads();
        int my_prop = 0;
        if (lane < S && any_active) {
            const double tot = np_sum_lds(t_nz, m_nz, 1);
            my_prop = my_v != 0.0 ? (int)((double)T * my_v / tot) : 0;      // :488-490 floor of a value >= 0
            i_rbs[lane] = my_prop;
        }
        __syncthreads();
        int mine = 0;
        if (lane < S && any_active) {
            int acc = 0, rank = 0;
            for16(S, [&](int j, bool ok) {
                const int pr = i_rbs[j]; const double xj = t_v[j];
                acc += ok ? pr : 0;
                rank += (ok && xj != 0.0 && (xj > my_v || (xj == my_v && j > lane))) ? 1 : 0;
            });
            const int adj = T - acc;                                         // :493-499
            int extra = 0;
            if (my_v != 0.0 && adj > 0 && m_nz > 0)   // hand-out i goes to sorted[i % m]; adj < m in exact arithmetic
                extra = adj < m_nz ? (rank < adj ? 1 : 0) : (adj / m_nz + (rank < adj % m_nz ? 1 : 0));
            mine = (my_prop + extra) * p.G;                                  // ib_sched.py:268
        }
        __syncthreads();
        if (lane < S) i_rbs[lane] = mine;
        __syncthreads();
        if (lane < S) {
            int off = 0;
            for16(S, [&](int j, bool ok) { const int v = i_rbs[j]; off += (ok && j < lane) ? v : 0; });
            i_off[lane] = off;
        }
        __syncthreads();
        RANENV_STAMP(3);

        // ---- P3: intra-slice scheduling, one lane per UE (ib_sched.py:272-344) -------------------
        //      RR common.py:508-555, PF :558-636, MT :639-701, distribute_rbs_ues :464-478
        // stage A: RR selection flag and throughput_available
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s), n_rbs = i_rbs[s];
                const double occ = d_occ[u];
                i_sel[row + pos] = d_isclose(occ, 0.0) ? 0 : 1;              // RR: UEs with packets (:519-524)
                if (i_choice[s] != RANENV_INTRA_RR) {
                    const double slice_bw = (double)n_rbs * p.bw_hz / (double)R;   // :573-578
                    const double cap = d_sem[u] * slice_bw / (double)n;
                    const double backlog = occ * (double)i_maxp[u] * (double)i_pkt[u];
                    c_a[row + pos] = cap < backlog ? cap : backlog;
                }
            }
        }
        __syncthreads();
        // stage B: PF weights / MT weights
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active && i_choice[s] != RANENV_INTRA_RR) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s);
                const double my_avail = c_a[row + pos];
                double my_num = my_avail;                                      // MT: weights = avail
                if (i_choice[s] == RANENV_INTRA_PF) {                          // :584-602
                    double max_avail = c_a[row];
                    for16(n, [&](int k, bool ok) { const double av = c_a[row + k]; max_avail = (ok && av > max_avail) ? av : max_avail; });
                    double snt = d_hmean[u] * (double)i_pkt[u];
                    if (d_isclose(my_avail, 0.0)) snt = 1.0;
                    my_num = d_isclose(snt, 0.0) ? 2.0 * max_avail : my_avail / snt;
                }
                c_b[row + pos] = my_num;
            }
        }
        __syncthreads();
        // stage C: proportional values, or fall back to round robin when the weights sum to 0
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active && i_choice[s] != RANENV_INTRA_RR) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s), n_rbs = i_rbs[s];
                const double wsum = np_sum_lds(c_b + row, n, 1);               // :603-608 (same for the whole slice)
                c_c[row + pos] = wsum != 0.0 ? (double)n_rbs * c_b[row + pos] / wsum : 0.0;
                if (pos == 0) i_nz[s] = wsum != 0.0 ? 1 : 0;                   // slice takes the round_int path
            }
        }
        __syncthreads();
        // stage D: compaction of each slice's non-zero values (:484-485), then floor shares
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active && i_choice[s] != RANENV_INTRA_RR && i_nz[s] != 0) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s);
                const double my_val = c_c[row + pos];
                int slot = 0;
                for16(n, [&](int k, bool ok) { const double v = c_c[row + k]; slot += (ok && k < pos && v != 0.0) ? 1 : 0; });
                if (my_val != 0.0) c_d[row + slot] = my_val;
            }
        }
        __syncthreads();
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active && i_choice[s] != RANENV_INTRA_RR && i_nz[s] != 0) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s), n_rbs = i_rbs[s];
                const double my_val = c_c[row + pos];
                int m = 0;
                for16(n, [&](int k, bool ok) { const double v = c_c[row + k]; m += (ok && v != 0.0) ? 1 : 0; });
                const double tot = np_sum_lds(c_d + row, m, 1);
                i_cnt[row + pos] = my_val != 0.0 ? (int)((double)n_rbs * my_val / tot) : 0;   // floor of a value >= 0
            }
        }
        __syncthreads();
        // stage E: hand out the remainder (round_int_equal_sum) or split round-robin
        FOR_PASS {
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            int my_cnt = 0;
            if (s >= 0 && any_active) {
                const int row = s * Us, pos = i_pos[u], n = sl_nues(s), n_rbs = i_rbs[s];
                const int choice = i_choice[s];
                if (choice != RANENV_INTRA_RR && i_nz[s] != 0) {
                    const double my_val = c_c[row + pos];
                    int acc = 0, m = 0, rank = 0;
                    for16(n, [&](int k, bool ok) {
                        const int c = i_cnt[row + k]; const double xk = c_c[row + k];
                        acc += ok ? c : 0;
                        m += (ok && xk != 0.0) ? 1 : 0;
                        rank += (ok && xk != 0.0 && (xk > my_val || (xk == my_val && k > pos))) ? 1 : 0;
                    });
                    my_cnt = i_cnt[row + pos];
                    const int adj = n_rbs - acc;
                    if (my_val != 0.0 && adj > 0 && m > 0)
                        my_cnt += adj < m ? (rank < adj ? 1 : 0) : (adj / m + (rank < adj % m ? 1 : 0));
                } else {
                    // round_robin: account_buffer only when it is the slice's own choice (:508-555, :609-617)
                    const bool account = choice == RANENV_INTRA_RR;
                    int k_sel = 0, idx = 0;
                    for16(n, [&](int k, bool ok) { const int f = (ok && account) ? i_sel[row + k] : 0; k_sel += f; idx += (k < pos) ? f : 0; });
                    const bool all = (k_sel == 0);
                    if (all) { k_sel = n; idx = pos; }
                    const bool sel = all || i_sel[row + pos] != 0;
                    const unsigned each = (unsigned)n_rbs / (unsigned)k_sel, rem = (unsigned)n_rbs - each * (unsigned)k_sel;
                    my_cnt = sel ? (int)(each + ((unsigned)idx < rem ? 1u : 0u)) : 0;
                }
            }
            if (u < U) i_count[u] = my_cnt;
        }
        __syncthreads();
        FOR_PASS {        // counts in slice order for the prefix below
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active) i_cnt[s * Us + i_pos[u]] = i_count[u];
        }
        __syncthreads();
        FOR_PASS {        // :464-478 contiguous ranges
            const int u = pass * WAVE + lane;
            const int s = u < U ? i_slice[u] : -1;
            if (s >= 0 && any_active) {
                const int row = s * Us, pos = i_pos[u];
                int start = i_off[s];
                for16(pos, [&](int k, bool ok) { const int c = i_cnt[row + k]; start += ok ? c : 0; });
                i_start[u] = start;
            }
        }
        __syncthreads();
    }
    RANENV_STAMP(4);

    // ---- P4: every UE's SE row: mean over all RBs and sum over its allocated RBs ------------------
    // two streams alternate so that the next pass's first loads fly under the current pass's sums
    auto do_rows = [&](SeStream &st, int u) {
        double se_full = 0.0, se_part = 0.0;
        const int uu = u < U ? u : U - 1;                  // idle lanes shadow the last UE (loads stay in bounds)
        if (RANENV_DIAG == 2) {
        } else if (MODE == MODE_STEP) {
            const unsigned ust = (unsigned)i_start[uu], ucn = (unsigned)i_count[uu];
            row_sums(st, R, [=](int r) { return ((unsigned)r - ust) < ucn; }, se_full, se_part);
        } else if (MODE == MODE_DENSE) {
            const uint8_t *mrow = p.dense + ((size_t)b * U + uu) * R;
            row_sums(st, R, [=](int r) { return mrow[r] != 0; }, se_full, se_part);
            int cnt = 0, first = 0; bool seen = false;
            for (int r = 0; r < R; r++) {
                if (mrow[r] != 0) { cnt++; if (!seen) { first = r; seen = true; } }
            }
            if (u < U) { i_count[u] = cnt; i_start[u] = first; }
        } else {
            row_sums(st, R, [](int) { return false; }, se_full, se_part);
        }
        if (u < U) { d_semn[u] = se_full / (double)R; d_part[u] = se_part; }
    };
#pragma unroll
    for (int pass = 0; pass < (PASSES > 0 ? PASSES : passes); pass += 2) {
        const int u0 = pass * WAVE + lane, u1 = u0 + WAVE;
        if (pass + 1 < passes) se_b.init(tile, U, u1 < U ? u1 : U - 1, R);
        do_rows(se_a, u0);
        if (pass + 1 < passes) {
            const int u2 = u1 + WAVE;
            if (pass + 2 < passes) se_a.init(tile, U, u2 < U ? u2 : U - 1, R);
            do_rows(se_b, u1);
        }
    }
    __syncthreads();
    for (int i = lane; i < S * Us * 3; i += WAVE) d_drift[i] = 0.0;   // shares storage with the P3 rows
    __syncthreads();
    RANENV_STAMP(5);

    // ---- P5: UEs.step for every UE (oracle/ranenv_oracle.c buffer_receive/buffer_send) ------------
    FOR_PASS {
        const int u = pass * WAVE + lane;
        if (u < U && RANENV_DIAG != 3) {
            const size_t su = (size_t)b * U + u;
            const int pkt_size = i_pkt[u], max_pkts = i_maxp[u], max_age = i_maxage[u];
            int total = 0, front = 0, front_rem = 0, fifo = 0;
            long long sum_age = 0, win_sent = 0, win_drop = 0;
            if (MODE != MODE_RESET) {
                total = p.st.queue_pkts[su]; sum_age = p.st.queue_age_sum[su];
                front = p.st.front[su]; front_rem = p.st.front_rem[su]; fifo = p.st.fifo[su];
            }
            if (!clear_hist) { win_sent = p.st.win_sent[su]; win_drop = p.st.win_dropped[su]; }
            // push slot of the 10-TTI window (IBSched.last_unformatted_obs.appendleft, ib_sched.py:64)
            int32_t *rs = p.st.ring_sent + ((size_t)b * D + npush) * U + u;
            int32_t *rd = p.st.ring_drop + ((size_t)b * D + npush) * U + u;
            int old_s = 0, old_d = 0;
            if (hlen == D) { old_s = *rs; old_d = *rd; }
            long long dropped = 0, sent = 0, pkt_in = 0, pkt_thr = 0;
            if (MODE != MODE_RESET) {
                const double traffic = p.traffic_bits
                    ? p.traffic_bits[su]
                    : (double)p.trf_pool[((size_t)ep.trf_base + (size_t)trf_pos) * U + u];
                const double psz = (double)pkt_size;
                // floor of non-negative values; v_cvt_i32_f64 truncates and saturates (host validates < 2^31)
                pkt_thr = (int)((d_part[u] * p.bw_per_rb) / psz);
                pkt_in = (int)(traffic / psz);
                const int L = p.L;
                // The queue is FIFO, so the age histogram Buffer keeps is exactly a list of
                // (arrival TTI, packets) entries in arrival order.  ring[e] holds entry e of a circular
                // list (head index + entry count per UE); only TTIs that admitted packets make an entry,
                // so expiring / draining costs one load per consumed entry and never a scan.
                int2 *ring = p.st.age_ring + (size_t)b * L * U + u;
                int head = fifo & 0xffff, nent = (int)((unsigned)fifo >> 16);
                auto pop_head = [&]() { nent--; head = head + 1 == L ? 0 : head + 1; };
                auto load_head = [&]() { const int2 e = ring[(size_t)head * U]; front = e.x; front_rem = e.y; };
                // receive_packets: the bin older than max_age expires ...
                if (nent > 0 && front == t - max_age - 1) {
                    dropped += front_rem; total -= front_rem; sum_age -= (long long)max_age * front_rem;
                    front_rem = 0;
                    pop_head();
                    if (nent > 0) load_head();
                }
                sum_age += total;                                   // ... everything left ages one TTI ...
                const long long space = (long long)max_pkts - total; // ... arrivals admitted up to capacity
                const long long adm = pkt_in < space ? pkt_in : space;
                dropped += pkt_in - adm;
                if (adm > 0) {
                    int tail = head + nent; tail = tail >= L ? tail - L : tail;
                    ring[(size_t)tail * U] = make_int2(t, (int)adm);
                    if (nent == 0) { front = t; front_rem = (int)adm; }
                    nent++;
                    total += (int)adm;
                }
                // send_packets: drain oldest first
                long long cap = pkt_thr;
                while (cap > 0 && nent > 0) {
                    const long long take = cap < front_rem ? cap : front_rem;
                    front_rem -= (int)take; total -= (int)take; cap -= take; sent += take;
                    sum_age -= (long long)(t - front) * take;
                    if (front_rem == 0) {
                        pop_head();
                        if (nent > 0) {
                            if (nent == 1 && adm > 0) { front = t; front_rem = (int)adm; }   // this TTI's entry
                            else load_head();
                        }
                    }
                }
                fifo = head | (nent << 16);
            }
            win_sent += sent - old_s; win_drop += dropped - old_d;
            *rs = (int32_t)sent; *rd = (int32_t)dropped;
            // state + raw outputs
            const double se_mean_new = d_semn[u];
            p.st.queue_pkts[su] = total; p.st.queue_age_sum[su] = sum_age;
            p.st.front[su] = front; p.st.front_rem[su] = front_rem; p.st.fifo[su] = fifo;
            p.st.win_sent[su] = win_sent; p.st.win_dropped[su] = win_drop;
            p.st.se_mean[su] = se_mean_new;
            p.st.pkt_effective_thr[su] = (int32_t)sent; p.st.dropped_pkts[su] = (int32_t)dropped;
            if (MODE == MODE_RESET) { i_count[u] = 0; i_start[u] = 0; }
            p.st.rb_start[su] = i_start[u]; p.st.rb_count[su] = i_count[u];
            if (!(p.flags & RANENV_F_NO_RAW_OUTPUT)) {
                p.st.pkt_incoming[su] = (int32_t)pkt_in; p.st.pkt_throughputs[su] = (int32_t)pkt_thr;
            }
            const double occ_new = (double)total / (double)max_pkts;
            const double lat_new = total > 0 ? (double)sum_age / (double)total : 0.0;
            d_occn[u] = occ_new;

            // intent drift of this UE (agents/common.py:68-340)
            const int s = i_slice[u];
            if (s >= 0 && sl_hasreq(s)) {
                const double o = p.over;
                const int npar = sl_npar(s), ue_pos = i_pos[u];
                for (int q = 0; q < npar; q++) {
                    const int metric = i_par[s * 6 + 2 * q], op = i_par[s * 6 + 2 * q + 1];
                    const double value = d_par[s * 3 + q];
                    double res;
                    if (metric == RANENV_METRIC_THROUGHPUT) {
                        double x = ((double)sent * (double)sl_msg(s)) / 1e6;           // common.py:25-31
                        bool zero = d_isclose(occ_new, 0.0);                            // :100-119
                        if (hlen_new > 1) zero = zero || d_isclose(d_occ[u], 0.0);
                        if (zero) x = value * (1.1 + o);
                        if (d_apply_op(op, x, value)) res = (x > value * (1.0 + o)) ? 1.0 : (x - value) / (value * o);
                        else res = -((value - x) / value);
                    } else if (metric == RANENV_METRIC_RELIABILITY) {
                        const double dw = (double)win_drop, sw = (double)win_sent;      // :32-53
                        const double buffer_pkts = occ_new * (double)sl_bsize(s) + dw + sw;
                        const double x = buffer_pkts != 0.0 ? dw / buffer_pkts : 0.0;
                        const double band = (100.0 - value) / 100.0;
                        if (d_apply_op(op, 100.0 * (1.0 - x), value)) res = (x < band * (1.0 - o)) ? 1.0 : (band - x) / (band * o);
                        else res = -((x - band) / (value / 100.0));
                    } else {
                        const double x = lat_new;                                       // :58-61
                        if (d_apply_op(op, x, value)) res = (x < value * (1.0 - o)) ? 1.0 : (value - x) / (value * o);
                        else res = -((x - value) / ((double)sl_blat(s) - value));
                    }
                    d_drift[((size_t)s * Us + ue_pos) * 3 + metric] = res;
                }
            }
        }
    }
    __syncthreads();
    RANENV_STAMP(6);

    // ---- P6: per-slice observation rows, sorted order (ib_sched.py:91-200) ------------------------
    if (lane < S && RANENV_DIAG != 4) {
        const int pos = lane;
        const int s = sl_sorted(pos);
        const int n = sl_nues(s);
        double sv[3] = {-2.0, -2.0, -2.0};
        if (n > 0 && sl_hasreq(s)) {                                  // common.py:343-378
            const int npar = sl_npar(s);
            for (int q = 0; q < npar; q++) {
                const int m = i_par[s * 6 + 2 * q];
                const double mean = np_sum_lds(d_drift + (size_t)s * Us * 3 + m, n, 3) / (double)n;
                sv[0] = m == 0 ? mean : sv[0]; sv[1] = m == 1 ? mean : sv[1]; sv[2] = m == 2 ? mean : sv[2];
            }
        }
        const double traffic_req = sl_active(s) == 1 ? d_slf[s * 2 + 1] : 0.0;
        const double priority = n != 0 ? d_slf[s * 2 + 0] : 0.0;
        double am[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const bool undeclared = d_isclose(sv[m], -2.0);
            am[m] = undeclared ? 0.0 : 1.0;
            sv[m] = undeclared ? 0.0 : sv[m];
        }
        double *se_u = c_d + (size_t)s * Us;
        double rbs_alloc = 0.0;
        float *oa = f_obs_intra + (size_t)s * W;
        for16(Us, [&](int k, bool ok) {
            const int ue = i_slues[s * Us + k];
            const bool have = ok && k < n;
            const int uei = have ? ue : 0;
            const double sem = d_semn[uei], occn = d_occn[uei];
            const int cnt = i_count[uei];
            if (have) { se_u[k] = sem; rbs_alloc += (double)cnt; }
            if (ok) {
                oa[9 + k] = have ? (float)occn : 0.0f;
                oa[9 + Us + k] = have ? (float)(sem / p.norm_se) : 0.0f;
            }
        });
        const double se_slice = n > 0 ? np_sum_lds(se_u, n, 1) / (double)n : 0.0;
        float *oi = f_obs_inter + pos * 10;
        oi[0] = (float)sv[0]; oi[1] = (float)sv[1]; oi[2] = (float)sv[2];
        oi[3] = (float)am[0]; oi[4] = (float)am[1]; oi[5] = (float)am[2];
        oi[6] = (float)priority; oi[7] = (float)(traffic_req / p.norm_traffic);
        oi[8] = (float)((double)n / p.norm_ues); oi[9] = (float)(se_slice / p.norm_se);
        oa[0] = oi[0]; oa[1] = oi[1]; oa[2] = oi[2]; oa[3] = oi[3]; oa[4] = oi[4]; oa[5] = oi[5];
        oa[6] = (float)(rbs_alloc / (double)R); oa[7] = oi[7]; oa[8] = oi[8];
        // float64 values for the reward (calculate_reward reads the same numbers), indexed by slice
        double mn = 0.0; int cntm = 0;                              // common.py:400-407
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const double v = sv[m];
            if (!d_isclose(v, -2.0)) { mn = (cntm == 0 || v < mn) ? v : mn; cntm++; }
        }
        d_slvals[s] = sl_active(s) ? (cntm > 0 ? mn : 1.0) : 0.0;   // active_observations[s]
        d_slflags[s] = sl_active(s) ? d_slf[s * 2 + 0] : 0.0;        // slice_priorities[s]
        // player_{s+1} reward (common.py:428-437)
        double r = 0.0; int cnt = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            if (am[m] > 0.0) { r = (cnt == 0 || sv[m] < r) ? sv[m] : r; cnt++; }
        }
        if (p.reward) p.reward[(size_t)b * (S + 1) + s + 1] = cnt > 0 ? r : 0.0;
        if (MODE == MODE_RESET) {
            p.st.mask_inter[(size_t)b * S + s] = (int8_t)sl_active(s);
            for (int k = 0; k < Us; k++) p.st.mask_intra[((size_t)b * S + s) * Us + k] = k < n ? 1 : 0;
        }
    }
    __syncthreads();
    RANENV_STAMP(7);

    // ---- P7: player_0 reward (common.py:389-427 after unsort_slices, ib_sched.py:372-392) ----------
    if (RANENV_DIAG != 4) {
        // every lane evaluates the (cheap, uniform) selection logic; lane 0 stores
        int n_neg = 0, n_prio_neg = 0;
        for16(S, [&](int s, bool ok) {
            const double ao = d_slvals[s], pr = d_slflags[s];
            n_neg += (ok && ao < 0.0) ? 1 : 0;
            n_prio_neg += (ok && pr * ao < 0.0) ? 1 : 0;
        });
        // compact the selected entries in slice order (np.mean over the boolean-indexed array)
        const int mode_sel = n_neg == 0 ? 0 : (n_prio_neg != 0 ? 1 : 2);
        bool mine = false; int slot = 0, m = 0;
        for16(S, [&](int s, bool ok) {
            const double ao = d_slvals[s], pr = d_slflags[s];
            const bool sel = ok && (mode_sel == 0 ? true : (mode_sel == 1 ? (ao * pr < 0.0) : (ao < 0.0)));
            m += sel ? 1 : 0; slot += (sel && s < lane) ? 1 : 0;
            mine = (s == lane) ? sel : mine;
        });
        if (lane < S && mine) d_tmp[slot] = d_slvals[lane];
        __syncthreads();
        if (lane == 0) {
            double rew = np_sum_lds(d_tmp, m, 1) / (double)m;
            if (mode_sel == 1) rew -= 1.0;
            if (p.reward) p.reward[(size_t)b * (S + 1)] = rew;
            const int step_new = (MODE == MODE_RESET) ? 0 : step + 1;
            p.st.step_no[b] = step_new;
            p.st.hist_len[b] = hlen_new;
            p.st.n_push[b] = npush + 1 == D ? 0 : npush + 1;
            if (MODE == MODE_RESET) { p.st.se_pos[b] = se_pos; p.st.trf_pos[b] = trf_pos; }
            else {
                p.st.se_pos[b] = se_pos + 1 >= ep.se_len ? 0 : se_pos + 1;
                p.st.trf_pos[b] = trf_pos + 1 >= ep.trf_len ? 0 : trf_pos + 1;
            }
            if (p.done) p.done[b] = (MODE != MODE_RESET && step_new >= p.max_steps) ? 1 : 0;
        }
    }
    RANENV_STAMP(8);
    if (RANENV_DIAG == 9 && MODE == MODE_STEP && lane == 0 && S >= 9) {
        for (int k = 0; k < 9; k++) p.st.policy_scores[(size_t)b * S + k] = (double)(stamps[k] - stamps[0]);
    }
    if (p.obs_inter) for (int i = lane; i < S * 10; i += WAVE) p.obs_inter[(size_t)b * S * 10 + i] = f_obs_inter[i];
    if (p.obs_intra) for (int i = lane; i < S * W; i += WAVE) p.obs_intra[(size_t)b * S * W + i] = f_obs_intra[i];
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
thread_local std::string g_last_error;

}  // namespace

struct ranenv {
    ranenv_config cfg;
    KP kp;
    std::vector<void *> allocs;
    ranenv_episode *d_episodes = nullptr;
    bool have_scenarios = false, have_episodes = false;
    int64_t se_tiles_n = 0, trf_rows_n = 0;   // extents of the bound pools (0 = none)
    int nt = 0, lds_bytes = 0;
    std::string err;
};

namespace {

int fail(ranenv_handle h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    g_last_error = buf;
    return code;
}

#define HIP_TRY(h, call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(h, RANENV_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int dev_alloc(ranenv_handle h, T **out, size_t count)
{
    void *ptr = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) return fail(h, RANENV_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    e = hipMemset(ptr, 0, bytes);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "hipMemset: %s", hipGetErrorString(e));
    h->allocs.push_back(ptr);
    *out = (T *)ptr;
    return RANENV_OK;
}

template <int MODE>
hipError_t launch(ranenv_handle h, const KP &kp, hipStream_t stream)
{
    const dim3 grid(kp.B), block(WAVE);
    const size_t lds = (size_t)h->lds_bytes;
    const int passes = (kp.U + WAVE - 1) / WAVE;
    if (passes == 1) hipLaunchKernelGGL((ranenv_kernel<MODE, 1>), grid, block, lds, stream, kp);
    else if (passes == 2) hipLaunchKernelGGL((ranenv_kernel<MODE, 2>), grid, block, lds, stream, kp);
    else hipLaunchKernelGGL((ranenv_kernel<MODE, 0>), grid, block, lds, stream, kp);
    return hipGetLastError();
}

template <int MODE, int PASSES>
hipError_t set_lds_attr1(int bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&ranenv_kernel<MODE, PASSES>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int MODE>
hipError_t set_lds_attr(int bytes)
{
    hipError_t e = set_lds_attr1<MODE, 1>(bytes);
    if (e == hipSuccess) e = set_lds_attr1<MODE, 2>(bytes);
    if (e == hipSuccess) e = set_lds_attr1<MODE, 0>(bytes);
    return e;
}

}  // namespace

extern "C" {

const char *ranenv_last_error(ranenv_handle h) { return h ? h->err.c_str() : g_last_error.c_str(); }
int ranenv_abi_version(void) { return RANENV_ABI_VERSION; }

int ranenv_create(const ranenv_config *cfg, ranenv_handle *out)
{
    if (!cfg || !out) return fail(nullptr, RANENV_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != RANENV_ABI_VERSION) return fail(nullptr, RANENV_E_INVALID, "abi_version %d != %d", cfg->abi_version, RANENV_ABI_VERSION);
    const int S = cfg->n_slices, U = cfg->n_ues, R = cfg->n_rbs, Us = cfg->max_ues_slice;
    if (cfg->batch < 1 || S < 1 || S > 64 || U < 1 || U > 1024 || R < 1 || R > 512 || Us < 1 || Us > 128 ||
        cfg->rbs_per_rbg < 1 || cfg->rbs_per_rbg > R || cfg->hist_depth < 1 || cfg->hist_depth > 64 ||
        cfg->max_age_cap < 1 || cfg->max_age_cap > 65000 || cfg->max_steps < 1 || cfg->n_scenarios < 1)
        return fail(nullptr, RANENV_E_INVALID,
                    "unsupported sizes: need 1<=S<=64, 1<=U<=1024, 1<=R<=512, 1<=Us<=128, 1<=G<=R, 1<=hist_depth<=64");
    if (!(cfg->bandwidth_hz > 0.0)) return fail(nullptr, RANENV_E_INVALID, "bandwidth_hz must be positive");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, RANENV_E_HIP, "no HIP device: %s", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, RANENV_E_INVALID, "device %d out of range (%d)", cfg->device, ndev);
    ranenv_handle h = new (std::nothrow) ranenv();
    if (!h) return fail(nullptr, RANENV_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    HIP_TRY(h, hipSetDevice(cfg->device));
    const size_t B = (size_t)cfg->batch, NS = (size_t)cfg->n_scenarios, D = (size_t)cfg->hist_depth;
    const size_t L = (size_t)cfg->max_age_cap + 1;
    KP &kp = h->kp;
    memset(&kp, 0, sizeof(kp));
    kp.B = cfg->batch; kp.S = S; kp.U = U; kp.R = R; kp.G = cfg->rbs_per_rbg; kp.Us = Us; kp.D = cfg->hist_depth;
    kp.L = (int)L; kp.max_steps = cfg->max_steps; kp.flags = cfg->flags;
    kp.policy = RANENV_POLICY_MARR; kp.fixed_intra = RANENV_INTRA_RR;
    kp.bw_hz = cfg->bandwidth_hz; kp.bw_per_rb = cfg->bandwidth_hz / (double)R; kp.over = cfg->overfulfill;
    kp.norm_traffic = cfg->norm_traffic; kp.norm_ues = cfg->norm_ues; kp.norm_se = cfg->norm_se;
    int rc = RANENV_OK;
#define ALLOC(field, count) if (rc == RANENV_OK) rc = dev_alloc(h, &field, (count))
    ALLOC(kp.tab.slice_i32, NS * S * 8); ALLOC(kp.tab.slice_f64, NS * S * 2);
    ALLOC(kp.tab.param_i32, NS * S * 6); ALLOC(kp.tab.param_f64, NS * S * 3);
    ALLOC(kp.tab.slice_ues, NS * S * Us);
    ALLOC(kp.tab.ue_slice, NS * U); ALLOC(kp.tab.ue_pos, NS * U); ALLOC(kp.tab.ue_pkt_size, NS * U);
    ALLOC(kp.tab.ue_max_pkts, NS * U); ALLOC(kp.tab.ue_max_age, NS * U);
    ALLOC(kp.st.queue_pkts, B * U); ALLOC(kp.st.queue_age_sum, B * U); ALLOC(kp.st.front, B * U);
    ALLOC(kp.st.front_rem, B * U); ALLOC(kp.st.fifo, B * U); ALLOC(kp.st.win_sent, B * U); ALLOC(kp.st.win_dropped, B * U);
    ALLOC(kp.st.se_mean, B * U);
    ALLOC(kp.st.age_ring, B * L * U); ALLOC(kp.st.ring_sent, B * D * U); ALLOC(kp.st.ring_drop, B * D * U);
    ALLOC(kp.st.hist_len, B); ALLOC(kp.st.n_push, B); ALLOC(kp.st.step_no, B);
    ALLOC(kp.st.se_pos, B); ALLOC(kp.st.trf_pos, B);
    ALLOC(kp.st.pkt_incoming, B * U); ALLOC(kp.st.pkt_throughputs, B * U); ALLOC(kp.st.pkt_effective_thr, B * U);
    ALLOC(kp.st.dropped_pkts, B * U); ALLOC(kp.st.rb_start, B * U); ALLOC(kp.st.rb_count, B * U);
    ALLOC(kp.st.mask_inter, B * S); ALLOC(kp.st.mask_intra, B * S * Us); ALLOC(kp.st.policy_scores, B * S);
    ALLOC(h->d_episodes, B);
#undef ALLOC
    if (rc != RANENV_OK) { std::string m = h->err; ranenv_destroy(h); g_last_error = m; return rc; }
    kp.episodes = h->d_episodes;
    h->nt = WAVE;
    h->lds_bytes = make_layout(S, U, Us).total;
    if (h->lds_bytes > 160 * 1024) { ranenv_destroy(h); return fail(nullptr, RANENV_E_INVALID, "LDS need %d B exceeds 160 KiB", h->lds_bytes); }
    e = set_lds_attr<MODE_STEP>(h->lds_bytes);
    if (e == hipSuccess) e = set_lds_attr<MODE_DENSE>(h->lds_bytes);
    if (e == hipSuccess) e = set_lds_attr<MODE_RESET>(h->lds_bytes);
    if (e != hipSuccess) {
        ranenv_destroy(h);
        return fail(nullptr, RANENV_E_HIP, "no usable gfx950 kernel image (hipFuncSetAttribute: %s)", hipGetErrorString(e));
    }
    *out = h;
    return RANENV_OK;
}

int ranenv_destroy(ranenv_handle h)
{
    if (!h) return RANENV_OK;
    for (void *p : h->allocs) (void)hipFree(p);
    delete h;
    return RANENV_OK;
}

int ranenv_load_scenarios(ranenv_handle h, int32_t first, int32_t count, const ranenv_scenario_tables *t, void *stream_)
{
    if (!h || !t) return fail(h, RANENV_E_INVALID, "null argument");
    const int S = h->cfg.n_slices, U = h->cfg.n_ues, Us = h->cfg.max_ues_slice;
    if (first < 0 || count < 1 || first + count > h->cfg.n_scenarios) return fail(h, RANENV_E_INVALID, "scenario rows [%d,%d) outside pool of %d", first, first + count, h->cfg.n_scenarios);
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t n = (size_t)count;
    // validate + pack on the host
    std::vector<int32_t> si(n * S * 8), pi(n * S * 6);
    std::vector<double> sf(n * S * 2), pf(n * S * 3);
    for (size_t i = 0; i < n * S; i++) {
        const int nues = t->slice_nues[i], npar = t->slice_nparams[i], srt = t->sorted_slices[i];
        if (nues < 0 || nues > Us) return fail(h, RANENV_E_INVALID, "slice_nues %d outside [0,%d]", nues, Us);
        if (npar < 0 || npar > 3) return fail(h, RANENV_E_INVALID, "slice_nparams %d outside [0,3]", npar);
        if (srt < 0 || srt >= S) return fail(h, RANENV_E_INVALID, "sorted_slices entry %d outside [0,%d)", srt, S);
        if (t->slice_has_req[i] && nues > 0 && (t->slice_message_size[i] <= 0 || t->slice_buffer_size[i] <= 0))
            return fail(h, RANENV_E_INVALID, "message_size and buffer_size must be positive");
        int32_t *d = &si[i * 8];
        d[0] = t->slice_active[i]; d[1] = t->slice_has_req[i]; d[2] = nues; d[3] = t->slice_buffer_size[i];
        d[4] = t->slice_buffer_latency[i]; d[5] = t->slice_message_size[i]; d[6] = npar; d[7] = srt;
        sf[i * 2] = t->slice_priority[i]; sf[i * 2 + 1] = t->slice_traffic[i];
        for (int k = 0; k < 3; k++) {
            const int m = t->param_metric[i * 3 + k], op = t->param_op[i * 3 + k];
            if (k < npar && (m < 0 || m > 2 || op < 0 || op > 4)) return fail(h, RANENV_E_INVALID, "bad intent parameter (metric %d, op %d)", m, op);
            pi[(i * 3 + k) * 2] = m; pi[(i * 3 + k) * 2 + 1] = op; pf[i * 3 + k] = t->param_value[i * 3 + k];
        }
        for (int k = 0; k < nues; k++) {
            const int ue = t->slice_ues[i * Us + k];
            if (ue < 0 || ue >= U) return fail(h, RANENV_E_INVALID, "slice_ues entry %d outside [0,%d)", ue, U);
        }
    }
    for (size_t i = 0; i < n; i++) {   // sorted_slices must be a permutation
        std::vector<char> seen(S, 0);
        for (int s = 0; s < S; s++) seen[t->sorted_slices[i * S + s]] = 1;
        for (int s = 0; s < S; s++) if (!seen[s]) return fail(h, RANENV_E_INVALID, "sorted_slices row %zu is not a permutation", i);
    }
    for (size_t i = 0; i < n * U; i++) {
        if (t->ue_pkt_size[i] <= 0 || t->ue_max_pkts[i] <= 0) return fail(h, RANENV_E_INVALID, "ue_pkt_size / ue_max_pkts must be positive");
        if (t->ue_max_age[i] < 0 || t->ue_max_age[i] > h->cfg.max_age_cap) return fail(h, RANENV_E_INVALID, "ue_max_age %d outside [0, max_age_cap=%d]", t->ue_max_age[i], h->cfg.max_age_cap);
        if (t->ue_slice[i] < -1 || t->ue_slice[i] >= S) return fail(h, RANENV_E_INVALID, "ue_slice %d outside [-1,%d)", t->ue_slice[i], S);
        if (t->ue_pos[i] < 0 || t->ue_pos[i] >= Us) return fail(h, RANENV_E_INVALID, "ue_pos %d outside [0,%d)", t->ue_pos[i], Us);
    }
    const size_t f = (size_t)first;
    const Tables &d = h->kp.tab;
#define PUT(dst, src, elems, type) HIP_TRY(h, hipMemcpyAsync((dst), (src), (elems) * sizeof(type), hipMemcpyHostToDevice, stream))
    PUT(d.slice_i32 + f * S * 8, si.data(), n * S * 8, int32_t);
    PUT(d.slice_f64 + f * S * 2, sf.data(), n * S * 2, double);
    PUT(d.param_i32 + f * S * 6, pi.data(), n * S * 6, int32_t);
    PUT(d.param_f64 + f * S * 3, pf.data(), n * S * 3, double);
    PUT(d.slice_ues + f * S * Us, t->slice_ues, n * S * Us, int32_t);
    PUT(d.ue_slice + f * U, t->ue_slice, n * U, int32_t);
    PUT(d.ue_pos + f * U, t->ue_pos, n * U, int32_t);
    PUT(d.ue_pkt_size + f * U, t->ue_pkt_size, n * U, int32_t);
    PUT(d.ue_max_pkts + f * U, t->ue_max_pkts, n * U, int32_t);
    PUT(d.ue_max_age + f * U, t->ue_max_age, n * U, int32_t);
#undef PUT
    HIP_TRY(h, hipStreamSynchronize(stream));  // staging vectors die at return
    h->have_scenarios = true;
    return RANENV_OK;
}

int ranenv_bind_se_pool(ranenv_handle h, const float *dev_pool, int64_t n_tiles, int64_t tile_stride)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (dev_pool == nullptr) {
        h->kp.se_pool = nullptr; h->kp.se_stride = 0; h->se_tiles_n = 0;
        return RANENV_OK;
    }
    if (n_tiles < 1 || tile_stride < (int64_t)h->cfg.n_ues * h->cfg.n_rbs)
        return fail(h, RANENV_E_INVALID, "SE pool needs n_tiles >= 1 and tile_stride >= U*R");
    h->kp.se_pool = dev_pool; h->kp.se_stride = tile_stride; h->se_tiles_n = n_tiles;
    h->have_episodes = false;  // descriptors are re-validated against the new pool
    return RANENV_OK;
}

int ranenv_bind_traffic_pool(ranenv_handle h, const int32_t *dev_pool, int64_t n_rows)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (dev_pool != nullptr && n_rows < 1) return fail(h, RANENV_E_INVALID, "traffic pool needs n_rows >= 1");
    h->kp.trf_pool = dev_pool; h->trf_rows_n = dev_pool ? n_rows : 0;
    h->have_episodes = false;
    return RANENV_OK;
}

int ranenv_set_episodes(ranenv_handle h, const ranenv_episode *eps, void *stream_)
{
    if (!h || !eps) return fail(h, RANENV_E_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int b = 0; b < h->cfg.batch; b++) {
        const ranenv_episode &e = eps[b];
        if (e.scenario < 0 || e.scenario >= h->cfg.n_scenarios) return fail(h, RANENV_E_INVALID, "env %d: scenario %d outside pool of %d", b, e.scenario, h->cfg.n_scenarios);
        if (e.se_len < 1 || e.se_offset < 0 || e.se_offset >= e.se_len || e.se_base < 0 || e.trf_len < 1 ||
            e.trf_offset < 0 || e.trf_offset >= e.trf_len || e.trf_base < 0)
            return fail(h, RANENV_E_INVALID, "env %d: need len >= 1, 0 <= offset < len, base >= 0", b);
        if (h->kp.se_pool && e.se_base + e.se_len > h->se_tiles_n)
            return fail(h, RANENV_E_INVALID, "env %d: SE trace [%lld,+%d) exceeds the bound pool of %lld tiles", b, (long long)e.se_base, e.se_len, (long long)h->se_tiles_n);
        if (h->kp.trf_pool && e.trf_base + e.trf_len > h->trf_rows_n)
            return fail(h, RANENV_E_INVALID, "env %d: traffic trace [%lld,+%d) exceeds the bound pool of %lld rows", b, (long long)e.trf_base, e.trf_len, (long long)h->trf_rows_n);
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipMemcpyAsync(h->d_episodes, eps, sizeof(ranenv_episode) * (size_t)h->cfg.batch, hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->have_episodes = true;
    return RANENV_OK;
}

int ranenv_set_policy(ranenv_handle h, int32_t policy, int32_t fixed_intra)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (policy < RANENV_POLICY_EXTERNAL || policy > RANENV_POLICY_MAPF) return fail(h, RANENV_E_INVALID, "unknown policy %d", policy);
    if (!(fixed_intra == RANENV_INTRA_RR || fixed_intra == RANENV_INTRA_PF || fixed_intra == RANENV_INTRA_MT || fixed_intra == RANENV_INTRA_PER_SLICE))
        return fail(h, RANENV_E_INVALID, "unknown intra-slice scheduler %d", fixed_intra);
    h->kp.policy = policy; h->kp.fixed_intra = fixed_intra;
    return RANENV_OK;
}

static int check_ready(ranenv_handle h, const float *se_tiles, const double *traffic_bits, bool need_traffic)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (!h->have_scenarios) return fail(h, RANENV_E_STATE, "no scenarios loaded (ranenv_load_scenarios)");
    if (!h->have_episodes) return fail(h, RANENV_E_STATE, "no episode descriptors (ranenv_set_episodes)");
    if (!se_tiles && !h->kp.se_pool) return fail(h, RANENV_E_STATE, "no SE tiles given and no SE pool bound");
    if (need_traffic && !traffic_bits && !h->kp.trf_pool) return fail(h, RANENV_E_STATE, "no traffic given and no traffic pool bound");
    return RANENV_OK;
}

int ranenv_reset(ranenv_handle h, const uint8_t *env_mask, const float *se_tiles, float *obs_inter, float *obs_intra,
                 double *reward, void *stream)
{
    int rc = check_ready(h, se_tiles, nullptr, false);
    if (rc != RANENV_OK) return rc;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KP kp = h->kp;
    kp.env_mask = env_mask; kp.se_tiles = se_tiles; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = nullptr;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = nullptr;
    hipError_t e = launch<MODE_RESET>(h, kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "reset launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_step(ranenv_handle h, const double *scores, const uint8_t *intra, const double *traffic_bits,
                const float *se_tiles, float *obs_inter, float *obs_intra, double *reward, uint8_t *done, void *stream)
{
    int rc = check_ready(h, se_tiles, traffic_bits, true);
    if (rc != RANENV_OK) return rc;
    if (!scores && h->kp.policy == RANENV_POLICY_EXTERNAL) return fail(h, RANENV_E_STATE, "policy is EXTERNAL but no inter-slice scores were given");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KP kp = h->kp;
    kp.env_mask = nullptr; kp.se_tiles = se_tiles; kp.scores = scores; kp.intra = intra; kp.traffic_bits = traffic_bits;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = done;
    hipError_t e = launch<MODE_STEP>(h, kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "step launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_step_dense(ranenv_handle h, const uint8_t *dense, const double *traffic_bits, const float *se_tiles,
                      float *obs_inter, float *obs_intra, double *reward, uint8_t *done, void *stream)
{
    int rc = check_ready(h, se_tiles, traffic_bits, true);
    if (rc != RANENV_OK) return rc;
    if (!dense) return fail(h, RANENV_E_INVALID, "null sched_decision");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KP kp = h->kp;
    kp.env_mask = nullptr; kp.se_tiles = se_tiles; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = traffic_bits;
    kp.dense = dense; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = done;
    hipError_t e = launch<MODE_DENSE>(h, kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "dense step launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_get_views(ranenv_handle h, ranenv_views *out)
{
    if (!h || !out) return fail(h, RANENV_E_INVALID, "null argument");
    const State &s = h->kp.st;
    out->pkt_incoming = s.pkt_incoming; out->pkt_throughputs = s.pkt_throughputs;
    out->pkt_effective_thr = s.pkt_effective_thr; out->dropped_pkts = s.dropped_pkts;
    out->queue_pkts = s.queue_pkts; out->queue_age_sum = (int64_t *)s.queue_age_sum;
    out->rb_start = s.rb_start; out->rb_count = s.rb_count; out->se_mean = s.se_mean;
    out->win_sent = (int64_t *)s.win_sent; out->win_dropped = (int64_t *)s.win_dropped;
    out->step_number = s.step_no; out->hist_len = s.hist_len;
    out->mask_inter = s.mask_inter; out->mask_intra = s.mask_intra; out->policy_scores = s.policy_scores;
    return RANENV_OK;
}

int ranenv_launch_info(ranenv_handle h, int32_t *grid, int32_t *block, int32_t *lds_bytes)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (grid) *grid = h->cfg.batch;
    if (block) *block = h->nt;
    if (lds_bytes) *lds_bytes = h->lds_bytes;
    return RANENV_OK;
}

}  // extern "C"
