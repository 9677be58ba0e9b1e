// ranenv.hip -- MI355X (gfx950) implementation of the C ABI in include/ranenv.h.
//
// One TTI = one kernel, one workgroup per environment, thread u owns UE u: the workgroup first turns the
// scores into RB ranges (inter-slice RBG split by 16 lanes, intra-slice RR/PF/MT by the UEs through LDS
// rows), then thread u streams UE u's spectral-efficiency row, updates UE u's packet queue and computes
// UE u's intent drift, all in registers; per-slice means, observation rows and rewards then go through LDS.
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
#include <hip/hip_ext.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "ranenv.h"

#define DEVFN __device__ __forceinline__

#ifndef RANENV_FAST_DIV
#define RANENV_FAST_DIV 1   /* 0: every f64 division through the plain operator (A/B and the parity check of ddiv itself) */
#endif
#ifndef RANENV_DIAG
#define RANENV_DIAG 0   /* diagnostic builds only: 1-5 skip phases of the step kernel, 9 stamps s_memtime at its
                           phase boundaries (tools/stamps.py) */
#endif

namespace {

enum { MODE_STEP = 0, MODE_DENSE = 1, MODE_RESET = 2,
       MODE_PE = 4 };   // ORed into a step / dense build's MODE: RANENV_F_SCALE_PER_ELEMENT (the masked SE sum scales every element by BW / R before adding)

// ---------------------------------------------------------------------------------------------
// kernel parameters
// ---------------------------------------------------------------------------------------------
// The handle's arrays are few allocations ("slabs") with many equally shaped fields each (field k of a slab at
// k * stride): one base pointer per slab in the kernel arguments instead of one per field.  50 pointers cost 100
// SGPRs, more than a wave has; the step kernel kept spilling them to VGPR lanes and reading them back.
struct Tables {  // scenario pool on the device, rows of [n_scenarios]
    int32_t *slice_i32;  // [NS][S][8] active, has_req, nues, buffer_size, buffer_latency, message_size, nparams, sorted
    double  *slice_f64;  // [NS][S][2] priority, traffic
    int32_t *param_i32;  // [NS][S][3][2] metric, op -- in the slice's own order (the head kernel), then the same BY METRIC (the step kernel):
                         // [NS][S][3][2] declared?, op of metric m (a later parameter for the same metric has overwritten an earlier one)
    double  *param_f64;  // [NS][S][3] value in the slice's own order, then [NS][S][3] value of metric m (1.0 where undeclared)
    int32_t *slice_ues;  // [NS][S][Us]
    int32_t *slot;       // [3][NS][S*16] slot_ue (UE id, -1 = empty slot), slot_mp, slot_pk (its max_pkts / pkt_size)
    int32_t *slice_usecase;                 // [NS][S] SchedColORAN: bit 0 eMBB, bit 1 URLLC
    int32_t *ue;         // [2][6][NS][U] ue_slice, ue_pos, ue_pkt_size, ue_max_pkts, ue_max_age, lane_ue -- all in LANE order: lane l
                         // of the step kernel owns UE lane_ue[l].  Set 0 (compact steps): a scenario's UEs in slices first
                         // (ascending UE id), the idle ones behind them, so that the waves beyond the last UE in a slice have
                         // nothing to step.  Set 1 (full-width launches): lane l = UE l, the coalesced order for the SE stream
};

struct State {
    int32_t *u4;         // [13][B][U] 4-byte per-UE fields (the ST_* accessors below name them)
    int64_t *u8;         // [4][B][U]  8-byte per-UE fields: queue_age_sum, win_sent, win_dropped (int64), se_mean (double)
    int32_t *b4;         // [10][B]    per-env counters
    int2 *age_ring; int32_t *ring_sent; int32_t *ring_drop;
    int8_t *mask_inter, *mask_intra; double *policy_scores;
    double *next_scores; // with next_rb_start / next_rb_count: the allocation made at the end of a step for the next one
                         // (device policy), valid while alloc_gen[e] == KP::alloc_gen
};
enum { N_U4 = 13, N_U8 = 4, N_B4 = 10, N_TUE = 12 };
#define ST_queue_pkts(p) ((p).st.u4 + (size_t)(0) * (size_t)(p).BU)
#define ST_front(p) ((p).st.u4 + (size_t)(1) * (size_t)(p).BU)
#define ST_front_rem(p) ((p).st.u4 + (size_t)(2) * (size_t)(p).BU)
#define ST_fifo(p) ((p).st.u4 + (size_t)(3) * (size_t)(p).BU)
#define ST_pkt_incoming(p) ((p).st.u4 + (size_t)(4) * (size_t)(p).BU)
#define ST_pkt_throughputs(p) ((p).st.u4 + (size_t)(5) * (size_t)(p).BU)
#define ST_pkt_effective_thr(p) ((p).st.u4 + (size_t)(6) * (size_t)(p).BU)
#define ST_dropped_pkts(p) ((p).st.u4 + (size_t)(7) * (size_t)(p).BU)
#define ST_rb_start(p) ((p).st.u4 + (size_t)(8) * (size_t)(p).BU)
#define ST_rb_count(p) ((p).st.u4 + (size_t)(9) * (size_t)(p).BU)
#define ST_next_rb_start(p) ((p).st.u4 + (size_t)(10) * (size_t)(p).BU)
#define ST_next_rb_count(p) ((p).st.u4 + (size_t)(11) * (size_t)(p).BU)
#define ST_last_push(p) ((p).st.u4 + (size_t)(12) * (size_t)(p).BU)
#define ST_queue_age_sum(p) ((int64_t *)((p).st.u8 + (size_t)(0) * (size_t)(p).BU))
#define ST_win_sent(p) ((int64_t *)((p).st.u8 + (size_t)(1) * (size_t)(p).BU))
#define ST_win_dropped(p) ((int64_t *)((p).st.u8 + (size_t)(2) * (size_t)(p).BU))
#define ST_se_mean(p) ((double *)((p).st.u8 + (size_t)(3) * (size_t)(p).BU))
#define ST_hist_len(p) ((p).st.b4 + (size_t)(0) * (size_t)(p).B)
#define ST_n_push(p) ((p).st.b4 + (size_t)(1) * (size_t)(p).B)
#define ST_step_no(p) ((p).st.b4 + (size_t)(2) * (size_t)(p).B)
#define ST_se_pos(p) ((p).st.b4 + (size_t)(3) * (size_t)(p).B)
#define ST_trf_pos(p) ((p).st.b4 + (size_t)(4) * (size_t)(p).B)
#define ST_alloc_gen(p) ((p).st.b4 + (size_t)(5) * (size_t)(p).B)
#define ST_episode_no(p) ((p).st.b4 + (size_t)(6) * (size_t)(p).B)
#define ST_reset_count(p) ((p).st.b4 + (size_t)(7) * (size_t)(p).B)
#define ST_push_total(p) ((p).st.b4 + (size_t)(8) * (size_t)(p).B)
#define ST_clear_mark(p) ((p).st.b4 + (size_t)(9) * (size_t)(p).B)
#define ST_age_ring(p) ((p).st.age_ring)
#define ST_ring_sent(p) ((p).st.ring_sent)
#define ST_ring_drop(p) ((p).st.ring_drop)
#define ST_mask_inter(p) ((p).st.mask_inter)
#define ST_mask_intra(p) ((p).st.mask_intra)
#define ST_policy_scores(p) ((p).st.policy_scores)
#define ST_next_scores(p) ((p).st.next_scores)
#define TB_ue_slice(p) ((p).tab.ue + (size_t)(0) * (size_t)(p).NSU)
#define TB_ue_pos(p) ((p).tab.ue + (size_t)(1) * (size_t)(p).NSU)
#define TB_ue_pkt_size(p) ((p).tab.ue + (size_t)(2) * (size_t)(p).NSU)
#define TB_ue_max_pkts(p) ((p).tab.ue + (size_t)(3) * (size_t)(p).NSU)
#define TB_ue_max_age(p) ((p).tab.ue + (size_t)(4) * (size_t)(p).NSU)
#define TB_lane_ue(p) ((p).tab.ue + (size_t)(5) * (size_t)(p).NSU)
#define TB_slot_ue(p) ((p).tab.slot + (size_t)(0) * (size_t)(p).NSL)
#define TB_slot_mp(p) ((p).tab.slot + (size_t)(1) * (size_t)(p).NSL)
#define TB_slot_pk(p) ((p).tab.slot + (size_t)(2) * (size_t)(p).NSL)
#define TB_slice_i32(p) ((p).tab.slice_i32)
#define TB_slice_f64(p) ((p).tab.slice_f64)
#define TB_param_i32(p) ((p).tab.param_i32)
#define TB_param_f64(p) ((p).tab.param_f64)
#define TB_slice_ues(p) ((p).tab.slice_ues)
#define TB_slice_usecase(p) ((p).tab.slice_usecase)

// Work queue of the persistent rollout, one set per workgroup class (hot words on lines of their own).
struct PersistCtl {
    unsigned fresh[8][32];            // [x][0]: cursor into shard x of the class's env list (entries x, x + 8, x + 16, ...)
    int spare; int pad0[31];
    unsigned exited; unsigned pad5[31];   // workgroups of this launch that have left: the last one resets the cursors for the next launch
    int abort; int pad1[31];          // a wait gave up: every workgroup leaves
    struct { unsigned head; unsigned pad2[31]; unsigned tail; unsigned pad3[31]; int avail; int pad4[31]; } q[8];   // ready queue of XCD x: head / tail tickets (monotonic), entries committed and not yet claimed
    unsigned long long stat[8][16];   // per XCD (a line each): [0] chunks kept, [1] pushes, [2] pops, [3] fresh takes, [4] polls that found nothing
};

struct KP {
    int B, S, U, R, G, Us, D, L, max_steps, flags, policy, fixed_intra;
    long long BU, NSU, NSL;   // slab strides: B*U, n_scenarios*U, n_scenarios*S*16
    int T;    // R / G: allocation units of the inter-slice split (an integer division costs a wave ~60 instructions: made once, on the host)
    int e0;   // first env of this launch
    int n_tti;       // TTIs this launch steps every env through (>= 1; more than one only inside ranenv_rollout)
    int alloc_gen;   // host generation of (policy, scenarios, episodes): a stored next-TTI allocation of another generation is stale
    int compact;     // step only the UEs that are in a slice (lanes are ordered slice members first): waves without one leave
                     // at once.  Set by the host when it is exact: UEs outside every slice get no traffic (see idle_traffic_ok)
    int late;        // 0 (default): every step allocates at its head; 1: a hashed half of the envs, 2: all envs allocate for the
                     // next TTI at the end of the step (device policy only), so that heads and tails of workgroups differ in
                     // what they load the CU with (RANENV_LATE; the default while a launch was one TTI)
    double bw_hz, bw_per_rb, over, norm_traffic, norm_ues, norm_se;
    Tables tab;
    State st;
    const ranenv_episode *episodes;
    const float *se_pool; long long se_stride;   // RB-major or RB-quad-major pool (streaming kernels); the UE-major copy for the gather kernels
    int se_quad;                                 // the bound pool is RB-quad-major [R/4][U][4] (ranenv_bind_se_pool_quad); explicit per-step tiles stay RB-major
    const double *se_mean_pool;                  // gather kernels: [tile][U] mean SE over the RBs of every pooled tile (sidecar)
    int se_rp;                                   // gather kernels: floats per UE row of the UE-major copy (R rounded up to 8)
    const int32_t *trf_pool;
    // counter-based traffic (ranenv_set_traffic_generator): Poisson draws keyed (seed; env id, episode, step, UE)
    int trf_gen; int env_id_base; unsigned long long trf_seed;
    const unsigned long long *pois_cdf;   // [NS][S][256] floor(P(X <= k) * 2^64), saturated
    const uint8_t *pois_guide;            // [NS][S][64]  smallest k with cdf[k] > j * 2^58
    const int32_t *max_steps_env;         // [B] per-env episode length or null (= max_steps)
    double *acc;                          // [B][8] running sums of the current episode (ranenv_enable_metrics) or null
    // per-call inputs (may be null)
    const uint8_t *env_mask;
    const double *scores; const uint8_t *intra; const double *traffic_bits; const float *se_tiles;
    const uint8_t *dense;
    // outputs (may be null)
    float *obs_inter; float *obs_intra; double *reward; uint8_t *done;
    // alternative heads (SchedTWC / SchedColORAN), bound by ranenv_bind_head_outputs
    float *head_obs; double *head_reward;
    // persistent rollout (ranenv_persist_kernel): this launch's workgroup class
    const int32_t *p_list;            // the class's envs
    int p_count, p_chunk;             // how many; TTIs of an env between two visits of the work queue
    const int32_t *m_list;            // mixed step launches (ranenv_core_kernel_mixed): the narrow class's envs (p_list: the wide class's)
    const int32_t *m_counts;          // ... and how many there are of each, on the device ([0] narrow, [1] wide): no host read-back
    struct PersistCtl *p_ctl;         // the class's counters and per-XCD queue heads
    unsigned long long *p_slots;      // [8][p_cap] queue entries {index + 1, item}
    int p_cap;                        // entries per queue (a power of two >= the batch)
    int *p_err;                       // sticky error word of the handle (a wait gave up), in host memory mapped for the device
};

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
DEVFN double row16_max_f64(double x)  // every lane gets the maximum of its row (no NaNs here: comparisons and v_max agree)
{
    auto rot = [](double v, auto ctrl) {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = dpp_row<decltype(ctrl)::value>((int)b), hi = dpp_row<decltype(ctrl)::value>((int)(b >> 32));
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    };
    x = fmax(x, rot(x, std::integral_constant<int, 0x121>{})); x = fmax(x, rot(x, std::integral_constant<int, 0x122>{}));
    x = fmax(x, rot(x, std::integral_constant<int, 0x124>{})); x = fmax(x, rot(x, std::integral_constant<int, 0x128>{}));
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

constexpr int WAVE = 64;
constexpr int GRP = 16;   // lanes per (env) group in alloc1/obs and per (env, slice) group in alloc2

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

#ifndef RANENV_SE_DEPTH
#define RANENV_SE_DEPTH 2          /* 8-row groups of the SE tile in flight per lane in the lean streaming kernel (step / reset at row
                                      widths 8 and 10: 96 VGPRs without spills; the dense-mask and 16-wide builds keep 1) */
#endif
#ifndef RANENV_DEFER_STATE
#define RANENV_DEFER_STATE 2   /* the part of the UE state the allocation does not need is requested 0: at kernel entry, 1: before
                                  the queue's last turn, 2: after the stream (default: ~20 registers fewer while the tile
                                  streams; with 8 loads in flight per lane the kernel fits 96 VGPRs = 5 waves per SIMD
                                  without spills; measured A/B in profiles/r02_ab_log.txt) */
#endif
#ifndef RANENV_GATHER_STATE_FIRST
#define RANENV_GATHER_STATE_FIRST 0
#endif
#ifndef RANENV_OBS_STAGE
#define RANENV_OBS_STAGE 1
#endif
#ifndef RANENV_COLD_ARGS
#define RANENV_COLD_ARGS 1
#endif
#ifndef RANENV_METRICS
#define RANENV_METRICS 1           /* 0 compiles the episode-metric sums out (A/B of their cost only) */
#endif
#ifndef RANENV_WARM_ENTRY
#define RANENV_WARM_ENTRY 1        /* 0: every TTI of a multi-TTI launch enters like the first (loads everything back) */
#endif
#ifndef RANENV_LATE_BUILT
#define RANENV_LATE_BUILT 1        /* 0 compiles the allocation-ahead path out */
#endif
#ifndef RANENV_LATE_DEFAULT
#define RANENV_LATE_DEFAULT 0      /* (1 until launches ran several TTIs: their workgroups drift apart by themselves, and allocating
                                      ahead only costs its round trip through HBM -- rollout 62.1 -> 61.2, gather 37.8 -> 36.6 us per TTI) */
#endif
#ifndef RANENV_GATHER_CARRY
#define RANENV_GATHER_CARRY 0      /* 1: the persistent SE gather build too carries the UE state between the TTIs of a chunk and requests the next TTI's
                                      inputs ahead (CARRY).  Measured and left off: +15 registers = 21 spills at 5 waves per SIMD (33.3 against 31.2 us
                                      per TTI) or 4 waves per SIMD without spills (31.7-32.3): the UE step gets slower, not faster -- that kernel is
                                      bound by VALU issue and LDS / barrier latency at full residency, not by these round trips (profiles/r05_ab_log.txt) */
#endif
#ifndef RANENV_SE_AUX
#define RANENV_SE_AUX 2            /* cache policy bits of the tile loads (gfx94x: 1 = sc0, 2 = nt, 16 = sc1); 0 = the round-4 loads.  See nt_store */
#endif
#ifndef RANENV_SE_NT_LANE
#define RANENV_SE_NT_LANE 1        /* the packed builds' tile loads (per-lane pointers) non-temporal as well */
#endif
#ifndef RANENV_SE_DEPTH_SMALL
#define RANENV_SE_DEPTH_SMALL 4   /* the same for batches that do not fill the CUs anyway (step kernel built for 4 waves per SIMD) */
#endif

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
    static constexpr int AUX = SE_NQ <= 2 ? RANENV_SE_AUX : 0;
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
#if RANENV_SE_NT_LANE
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
#ifndef RANENV_GATHER_AUX
#define RANENV_GATHER_AUX 2        /* cache policy bits of the gather's loads from the UE-major copy (2 = nt: each is read once per TTI; 0 = plain) */
#endif
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
            const v4f a = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, RANENV_GATHER_AUX));
            const v4f b = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off < OOB ? off + 16 : OOB, 0, RANENV_GATHER_AUX));
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
// (indexed by the top 6 bits of u) gives a k at or below the answer.  The walk from there looks at four entries per turn, requested
// together: one memory round trip per turn instead of one per entry (a wave walks as long as its slowest lane, and every
// round trip is 1-2 us of the UE step under load).
#ifndef RANENV_NARROW_PRIO
#define RANENV_NARROW_PRIO 1        /* s_setprio of the one-wave class's waves inside the persistent launches (0: none): that class finishes a rollout
                                       ~7 % behind the two-wave class; issuing first evens them out (gather: K = 20 -2.4 %, K = 200 -0.5 %; streaming: nothing) */
#endif
#ifndef RANENV_POISSON_WINDOW
#define RANENV_POISSON_WINDOW 8     /* 1: the plain walk, one entry per turn */
#endif
DEVFN int poisson_draw(const unsigned long long *cdf, const uint8_t *guide, unsigned long long u)
{
    int k = guide[u >> 58];
#if RANENV_POISSON_WINDOW <= 1
    while (k < 255 && cdf[k] <= u) k++;
#else
    constexpr int WIN = RANENV_POISSON_WINDOW;
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
#endif
    return k;
}

#if RANENV_DIAG == 9   /* diagnostic build: s_memtime (100 MHz) of thread 0 at up to S phase boundaries of the step
                          kernel, dumped into policy_scores[e][k] instead of the scores (tools/stamps.py) */
#define RANENV_STAMP(k) do { if (!PERSIST && threadIdx.x == 0 && (k) < p.S) \
    ST_policy_scores(p)[(size_t)(p.e0 + blockIdx.x) * p.S + (k)] = (double)__builtin_amdgcn_s_memrealtime(); } while (0)
#elif RANENV_DIAG == 12 /* diagnostic build for the PERSISTENT launches (tools/persist_phases.py): thread 0 adds the time since its previous
                          stamp to policy_scores[e][k] (fire-and-forget atomics): [1..8] the phases of a TTI, [0] the gap between two TTIs of
                          a chunk, [9] TTIs counted; the scores themselves are not written.  Needs S >= 10. */
#define RANENV_STAMP(k) do { if constexpr (PERSIST) if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); \
    double *d_ = &ST_policy_scores(p)[(size_t)e * p.S]; \
    if ((k) == 0) { if (warm) acc_add(d_, (double)(now_ - cy.last_stamp)); acc_add(d_ + 9, 1.0); } else acc_add(d_ + (k), (double)(now_ - stamp_prev)); \
    stamp_prev = now_; if ((k) == 8) cy.last_stamp = now_; } } while (0)
#else
#define RANENV_STAMP(k) do { } while (0)
#endif

// =============================================================================================
// The step kernel: one workgroup = one env, thread u owns UE u, roles in sequence
//   (0) alloc    IBSched.action_format agents/ib_sched.py:223-349 for this TTI (MODE_STEP only):
//                policy MARR agents/marr.py:40-47 / MAPF agents/mapf.py:41-111 and the inter-slice split
//                (scores_to_rbs / round_int_equal_sum agents/common.py:442-505) by lanes 0..15 of wave 0
//                (lane = slice); intra-slice round_robin :508-555 / proportional_fairness :558-636 /
//                max_throughput :639-701 / distribute_rbs_ues :464-478 by thread = UE, the UEs of a slice
//                meeting in that slice's LDS rows (indexed by position in the slice).  Everything it
//                needs from HBM is state the UE role loads anyway; the first SE loads are already in
//                flight while it runs.
//   (1) stream   thread = UE: SE row sums in numpy's pairwise order (SeStream / row_sums)
//   (2) UE step  thread = UE: capacity -> UEs.step -> 10-TTI window -> intent drift
//                (oracle/ranenv_oracle.c; agents/common.py:68-340)
//   (3) obs      thread = slice (sorted position), threads 0..15: calculate_slice_ue_obs
//                agents/common.py:343-378, IBSched.obs_space_format agents/ib_sched.py:91-200,
//                calculate_reward :206-221 + common.py:381-439, per-env counters
// The scenario's slice tables are staged in LDS once per workgroup: every role reads them from there.
// =============================================================================================
constexpr int CORE_NT = GRP * GRP;   // 256 = largest U

template <int NP>           // row width of the build: S <= NP slices, <= NP UEs per slice
struct SharedCore {
    // per slice 4 rows of NP doubles by UE position: allocation scratch, then drift x3 + mean SE for (3).  The lanes of
    // a wave belong to different slices and read the same position of their own slice's row: the 2-double pad keeps
    // the slices off one bank (16-byte alignment of the rows kept for 128-bit LDS reads).  Sized by NP, not by 16: with one
    // wave per env (compact steps) it is LDS that caps the workgroups of a CU -- 5.6 KB instead of 12.2 KB at NP = 10
    double rows[NP][4 * NP + 2];
    double xr[4][GRP];            // cross-slice rows
    double pf[NP][3];             // param value                     } slice tables of this env's scenario
    double sf[NP][2];             // priority, traffic               }
    int si[NP][8];                // active, has_req, nues, buffer_size, buffer_latency, message_size, nparams, sorted
    int pi[NP][6];                // (metric, op) x 3
    int cnt[NP][NP + 4];          // RBs of each slot (padded like rows)
    unsigned msk[NP][2];              // per slice, one bit per UE position, set with LDS atomic ORs by the slice's UEs and read as ONE word:
                                      // [0] the UE's buffer is not empty, [1] its PF / MT value is non-zero.  All zero between two allocations
    int rbs[GRP], off[GRP];       // RBs of each slice and its first RB
#if RANENV_OBS_STAGE
    // this TTI's observation rows, staged here and written out by wave 0 as whole lines: written one float per lane and
    // instruction they were ~170 partial-line store requests per env (profiles/r02_pmc_memsys.txt)
    float ob_inter[NP * 10];
    float ob_intra[NP * (2 * NP + 9)];
#endif
};

// Workgroup barrier for exchanges through LDS only: waits for this wave's LDS operations, not for its global loads and
// stores (the UE role's ~17 state stores need not be acknowledged before the obs role starts, and the SE loads requested ahead
// of the allocation need not land before its first exchange).  Nothing in the step kernel passes data between threads through
// global memory.  Where global memory IS handed over -- to the next TTI of the same workgroup without a warm entry, or to another
// workgroup (persistent rollout) -- full_sync() below is used: __syncthreads() alone is NOT enough, the compiler's
// workgroup-scope fence waits for lgkmcnt only (no vmcnt(0) outside tgsplit mode: ADVICE r4, seen in the shipped ISA).
#ifndef RANENV_LDS_BARRIER
#define RANENV_LDS_BARRIER 1
#endif
// `narrow`: this wave is a workgroup of its own inside a two-wave block (ranenv_core_kernel_mixed: two one-wave envs per block):
// its exchanges through LDS are between its own lanes, so it waits for its LDS operations and must NOT take part in a block
// barrier -- the block's other wave steps another env, or has left.
DEVFN void wg_sync(const bool narrow = false)
{
    if (narrow) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
#if RANENV_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}
DEVFN void full_sync(const bool narrow = false)      // every memory operation of this wave is complete (stores acknowledged by L2), then the barrier
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (narrow) return;                              // (a narrow wave is a workgroup of its own)
    __syncthreads();
}

template <int NP> DEVFN double *srow(SharedCore<NP> &sh, int s, int k) { return &sh.rows[s][k * NP]; }

// Element of a per-env row of a global array: the array's base and the env's row (`row_bytes`, wave-uniform: scalar
// arithmetic, a pair of SGPRs) + the lane's own byte offset in ONE VGPR -- the "saddr" form of global_load / global_store --
// instead of a 64-bit address per array and lane (a VGPR pair and two vector adds each; the step kernel touches ~30 arrays).
// The row address is made opaque where it is used: otherwise the optimiser forms array + row + lane once, as a 64-bit vector
// value, and carries it from the load at the top of the step to the store at its end.
// (PACK = 2: two envs per wave, lanes 0-31 / 32-63 -- the row differs between the halves.  The array's base stays the scalar part
// and row + lane offset go into the ONE 32-bit register of the saddr form: the host packs waves only while every array a packed
// step addresses this way stays below 4 GB -- pack_fits_32 -- so the sum cannot wrap.  A 64-bit address per array and lane cost
// the packed builds 13 spilled registers)
template <int PACK = 1, typename T> DEVFN T &row_at(T *array, size_t row_bytes, unsigned lane_bytes)
{
    if constexpr (PACK == 1) {
        char *row = (char *)array + row_bytes;
        asm volatile("" : "+s"(row));
        return *(T *)(row + lane_bytes);
    } else {
        const unsigned off = (unsigned)row_bytes + lane_bytes;
        return *(T *)((char *)array + off);
    }
}

// Cache hints (round 5; same-box A/B in profiles/r05_ab_log.txt).  The SE tile is read once per TTI and never again: its loads carry the
// non-temporal bit, so that 292 MB of tiles per TTI do not push the per-UE state -- re-read at the very next TTI -- out of the caches
// (streaming rollout -2...-3 %).  In the SE gather builds, which stream no tile, the same goes for what the kernel WRITES and will not read
// again soon (observation rows, raw outputs, age-list and window-ring entries; gather -3.7 %; no gain for the streaming builds, so they
// keep plain stores) and for the sidecar reads.
#ifndef RANENV_NT_STORES
#define RANENV_NT_STORES 1         /* 0: plain stores in the gather builds too */
#endif
template <bool NT, typename T> DEVFN void nt_store(T &dst, const T v)
{
    if constexpr (NT && RANENV_NT_STORES != 0) {
        if constexpr (sizeof(T) == 8 && !std::is_floating_point<T>::value && !std::is_integral<T>::value)      // (int2: as one 8-byte word)
            __builtin_nontemporal_store(__builtin_bit_cast(long long, v), (long long *)&dst);
        else
            __builtin_nontemporal_store(v, &dst);
    } else {
        dst = v;
    }
}

// Role (0).  Called by every thread of the workgroup (it contains barriers); `have` = this thread's UE is
// in a slice (slc, position pos).  q / mp / pk: queue length, buffer size, packet size; wsent: packets sent
// in the window, hlen its length; sem: mean SE of the previous tile.  Rows of sh.rows are zero beyond a
// slice's UE count on entry and on exit (np_sum16_lds relies on it); the entries below it are scratch.
template <int NP, int PACK = 1, typename P>
DEVFN void alloc_front(const P &p, SharedCore<NP> &sh, int tid, int e, int hlen, bool have, int slc, int pos,
                       int q, int mp, int pk, long long wsent, double sem, int &rb_start, int &rb_count, double *scores_out,
                       const bool narrow = false)
{
    auto &xs = sh.xr;
    auto wave_sync = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    const int S = p.S;
    const bool mapf = p.scores == nullptr && p.policy == RANENV_POLICY_MAPF;
    const int sl = have ? slc : 0;                 // idle threads read row 0 and write nothing
    double *r0 = srow(sh, sl, 0), *r1 = srow(sh, sl, 1), *r2 = srow(sh, sl, 2), *r3 = srow(sh, sl, 3);
    int choice = p.fixed_intra;                    // requested now, used after the inter-slice part
    if (choice == RANENV_INTRA_PER_SLICE) choice = (p.intra && have) ? (int)p.intra[(size_t)e * S + sl] : RANENV_INTRA_RR;
    const double occ = ddiv((double)q, (double)mp);
    const double hm = hlen > 0 ? ddiv((double)wsent, (double)hlen) : 0.0;
    const bool has_pkts = have && !d_isclose(occ, 0.0);
    if (have) { r0[pos] = occ; r1[pos] = hm; }
    if (has_pkts) atomicOr(&sh.msk[sl][0], 1u << pos);
    wg_sync(narrow);

    // ---- inter-slice: lane t < 16 of wave 0 is slice t ----------------------------------------------
    if (tid < WAVE) {            // the other waves go straight to the barrier below
        const int s1 = tid;
        const bool ok1 = tid < GRP && s1 < S;
        int active = 0, nues1 = 0, sorted = 0;
        if (ok1) { active = sh.si[s1][0]; nues1 = sh.si[s1][2]; sorted = sh.si[s1][7]; }
        double score = -1.0;
        if (mapf) {
            // backlog and sent Mbit of every slice (mapf.py:63-90): the two rows side by side, lanes 0..15 the occupancy row and
            // lanes 16..31 the window row of slice (lane & 15), instead of one after the other on 16 lanes
            {
                const int half = tid >> 4, sl2 = tid & (GRP - 1);
                double v2 = 0.0;
                if (tid < 2 * GRP && sl2 < S && sh.si[sl2][0] != 0) {
                    const int n2 = sh.si[sl2][2];
                    v2 = ddiv(np_sum_lds<NP>(srow(sh, sl2, half), n2), (double)n2);
                    if (half == 0) v2 = v2 * (double)sh.si[sl2][3];                     // x buffer size
                    v2 = ddiv(v2 * (double)sh.si[sl2][5], 1e6);                         // x message size, to Mbit
                }
                if (tid < 2 * GRP) xs[half][sl2] = v2;
            }
            wave_sync();
            double occ_mb = 0.0, thr_mb = 0.0;
            if (tid < GRP) { occ_mb = xs[0][s1]; thr_mb = xs[1][s1]; }
            double w = 0.0;
            if (tid < GRP) {
                // the largest backlog over the slices (np.max over all S entries): every slice lane holds its own, a DPP row maximum
                // instead of ten LDS reads and compares per lane
                const double mx = row16_max_f64(s1 < S ? occ_mb : -__builtin_inf());
                w = d_isclose(thr_mb, 0.0) ? 2.0 * mx : ddiv(occ_mb, thr_mb);                       // :91-100
                if (!active) w = 0.0;
                xs[2][s1] = ok1 ? w : 0.0;
            }
            wave_sync();
            if (tid < GRP) {
                const double ws = np_sum_lds<NP>(xs[2], S);
                score = (ws > 0.0 ? ddiv(w, ws) : 2.0) - 1.0;                                       // :105-109
            }
            wave_sync();
        } else if (ok1) {
            score = p.scores ? row_at<PACK>(p.scores, (size_t)e * S * 8, (unsigned)s1 * 8u) : (nues1 > 0 ? 1.0 : -1.0);   // marr.py:40-47
        }
#if RANENV_DIAG != 9 && RANENV_DIAG != 12
        if (ok1) row_at<PACK>(scores_out, (size_t)e * S * 8, (unsigned)s1 * 8u) = score;
#endif
        if (tid < GRP) xs[3][s1] = score;
        wave_sync();
        const int T = p.T;
        double my_a = -1.0;
        if (tid < GRP) {
            my_a = (ok1 && active) ? xs[3][sorted] : -1.0;                                           // ib_sched.py:247-255
            xs[0][s1] = ok1 ? my_a + 1.0 : 0.0;
        }
        wave_sync();
        double my_v = 0.0; bool nzf = false; int m_nz = 0, slot = 0;
        if (tid < GRP) {
            // np.sum(association) adds small integers: exact in any order, so an integer row sum does it
            const double ssum = np_sum_lds<NP>(xs[0], S), asum = (double)row16_sum(ok1 ? active : 0);
            if (ok1 && asum != 0.0) my_v = ssum != 0.0 ? ddiv((double)T * (my_a + 1.0), ssum) : ddiv((double)T, asum) * (double)active;
            nzf = my_v != 0.0;
            // compaction of the non-zero values in slice order (common.py:484-485): they move to the front,
            // the zeros fill the slots behind them: every slot is written exactly once
            // (a packed wave: this env's 16 slice lanes start at lane 32 of the wave for the second env)
            const unsigned gm = (unsigned)((__ballot(nzf) >> (PACK == 2 ? (threadIdx.x & 32u) : 0u)) & 0xffffull), below = (1u << s1) - 1u;
            m_nz = __popc(gm); slot = nzf ? __popc(gm & below) : m_nz + __popc(~gm & below);
            xs[2][slot] = my_v; xs[3][s1] = my_v;
        }
        wave_sync();
        if (tid < GRP) {
            const double tot = np_sum_lds<NP>(xs[2], m_nz);
            const int my_prop = nzf ? (int)ddiv((double)T * my_v, tot) : 0;               // :488-490 (value >= 0)
            const int acc = row16_sum(my_prop);
            const int adj = T - acc;                                                     // :493-499
            int extra = 0;
            if (nzf && adj > 0) {
                int rank = 0;
#pragma unroll
                for (int j = 0; j < NP; j++) { const double xj = xs[3][j]; rank += (xj != 0.0 && (xj > my_v || (xj == my_v && j > s1))) ? 1 : 0; }
                // the floors leave fewer than m_nz units over unless rounding interferes: the integer division (~50 vector
                // instructions with its remainder) only where some lane needs it
                extra = rank < adj ? 1 : 0;
                if (__builtin_amdgcn_ballot_w64(adj >= m_nz) != 0 && adj >= m_nz) extra = adj / m_nz + (rank < adj % m_nz ? 1 : 0);
            }
            const int mine = (my_prop + extra) * p.G;                                    // ib_sched.py:268
            const int incl = row16_scan(mine);
            sh.rbs[s1] = mine; sh.off[s1] = incl - mine;
        }
    }
    wg_sync(narrow);

    // ---- intra-slice: thread = UE; a slice's UEs exchange through its rows ---------------------------
    const int n = have ? sh.si[sl][2] : 0;
    const int n_rbs = have ? sh.rbs[sl] : 0, off = have ? sh.off[sl] : 0;
    // PF / MT weights.  With round-robin fixed for every slice (MARR's and the heads' setting, a kernel argument, so
    // uniform over the workgroup) none of this is needed -- not even its barriers.
    const bool all_rr = p.fixed_intra == RANENV_INTRA_RR;
    bool use_round = false, nzv = false;
    double my_val = 0.0;
    int prop = 0, m_v = 0;
    const unsigned below = (1u << pos) - 1u;
    if (!all_rr) {
        double avail = 0.0;                      // evaluated by every slice (a per-slice choice may need it)
        if (have) {
            const double slice_bw = ddiv((double)n_rbs * p.bw_hz, (double)p.R);         // common.py:573-578
            const double cap = ddiv(sem * slice_bw, (double)n);
            const double backlog = occ * (double)mp * (double)pk;
            avail = cap < backlog ? cap : backlog;
            r0[pos] = avail;                     // (the occupancy row was consumed by the inter-slice part)
        }
        wg_sync(narrow);
        double num = avail;                                                            // MT: weights = avail
        if (choice == RANENV_INTRA_PF) {                                               // :584-602
            double snt = hm * (double)pk;
            if (d_isclose(avail, 0.0)) snt = 1.0;
            const bool starved = d_isclose(snt, 0.0);
            double max_avail = 0.0;
            if (__builtin_amdgcn_ballot_w64(starved) != 0) {       // the slice maximum is only read by UEs that sent nothing
                max_avail = r0[0];
#pragma unroll
                for (int k = 1; k < NP; k++) { const double av = r0[k]; max_avail = (k < n && av > max_avail) ? av : max_avail; }
            }
            num = starved ? 2.0 * max_avail : ddiv(avail, snt);
        }
        if (have) r1[pos] = num;
        wg_sync(narrow);
        const double wsum = np_sum_lds<NP>(r1, n);
        use_round = n > 0 && wsum != 0.0 && choice != RANENV_INTRA_RR;                 // :603-608
        my_val = (use_round && have) ? ddiv((double)n_rbs * num, wsum) : 0.0;
        nzv = my_val != 0.0;
        if (have) r2[pos] = my_val;
        if (have && nzv) atomicOr(&sh.msk[sl][1], 1u << pos);
        wg_sync(narrow);
        const unsigned gmv = sh.msk[sl][1];      // which positions of the slice hold a non-zero value
        m_v = __popc(gmv);
        const int slot_v = nzv ? __popc(gmv & below) : m_v + __popc(~gmv & below & 0xffffu);
        if (have) r3[slot_v] = my_val;                                                 // compaction (:484-485); zeros go behind
        wg_sync(narrow);
        if (use_round) {
            const double tot = np_sum_lds<NP>(r3, m_v);
            prop = nzv ? (int)ddiv((double)n_rbs * my_val, tot) : 0;                   // floor of a value >= 0
        }
    }
    if (!all_rr) {
        if (have) sh.cnt[sl][pos] = prop;
        wg_sync(narrow);
    }
    int count = 0;
    if (use_round) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) acc += sh.cnt[sl][k];
        const int adj = n_rbs - acc;
        count = prop;
        if (nzv && adj > 0) {
            int rank = 0;
#pragma unroll
            for (int k = 0; k < NP; k++) { const double xk = r2[k]; rank += (xk != 0.0 && (xk > my_val || (xk == my_val && k > pos))) ? 1 : 0; }
            int more = rank < adj ? 1 : 0;
            if (__builtin_amdgcn_ballot_w64(adj >= m_v) != 0 && adj >= m_v) more = adj / m_v + (rank < adj % m_v ? 1 : 0);      // (as above)
            count += more;
        }
    } else {
        // round_robin; the buffer filter applies only when RR is the slice's own choice (:508-555, :609-617)
        unsigned gmr = sh.msk[sl][0];
        if (choice != RANENV_INTRA_RR) gmr = 0u;
        int k_sel = __popc(gmr), idx = __popc(gmr & below);
        const bool all = (k_sel == 0);
        if (all) { k_sel = n; idx = pos; }
        if (have && (all || has_pkts) && k_sel > 0) {
            const unsigned each = (unsigned)n_rbs / (unsigned)k_sel, rem = (unsigned)n_rbs - each * (unsigned)k_sel;
            count = (int)(each + ((unsigned)idx < rem ? 1u : 0u));
        }
    }
    if (!all_rr) wg_sync(narrow);                                                  // every prop was read
    if (have) sh.cnt[sl][pos] = count;
    wg_sync(narrow);
    if (have && pos == 0) { sh.msk[sl][0] = 0u; sh.msk[sl][1] = 0u; }              // (both masks were read in front of that barrier; the next ORs are barriers away)
    int before = 0;                                                                // :464-478 contiguous ranges
#pragma unroll
    for (int k = 0; k < NP; k++) before += k < pos ? sh.cnt[sl][k] : 0;
    rb_start = have ? off + before : 0;
    rb_count = have ? count : 0;
}

// The kernel body, instantiated per build (see the kernels behind it): NQ = groups of 8 SE loads in flight per lane;
// GATHER = the SE gather mode (the tile's per-UE mean from the sidecar, the masked sum by gather_part from the UE-major
// copy; p.se_pool / p.se_stride then describe that copy) instead of streaming the whole RB-major tile.
// What a launch that steps several TTIs (step_loop) hands from one TTI to the next in registers instead of storing it and
// loading it back: the env's counters (uniform), the lane's table row and the three values of its UE's state that the
// allocation reads.  Every dependent load the entry does not make is ~1.5 us of a workgroup's life under load -- hit or miss:
// it queues behind the other workgroups' SE loads.
struct StepCarry {
    ranenv_episode ep;
    int t, hlen, npush, se_pos, trf_pos, ptot, cmark, episode_no;
    int u, slc, ue_pos, pkt_size, max_pkts, max_age, total;
    long long win_sent;
    double sem_prev;
    // CARRY builds (see step_body): the rest of the UE's state, and what the next TTI would otherwise load at its entry or in the
    // middle of its UE step -- requested a TTI ahead: the window slots it gives up, its traffic word, the two age-list entries behind the head
    long long sum_age, win_drop;
    int front, front_rem, fifo;
    int pf_old_s, pf_old_d, pf_traffic;
    int2 pf1, pf2;
#if RANENV_DIAG == 12
    unsigned long long last_stamp;
#endif
};

template <int MODE_X, int NQ, bool GATHER, int NP, bool PERSIST = false, int PACK = 1, bool MIX = false, typename P>
DEVFN bool step_body(const P &p, StepCarry &cy, const bool warm, const int e_in,   // warm: `cy` holds what the previous TTI of this launch left
                     std::conditional_t<PACK == 2, SeStreamLane<GATHER ? 1 : NQ>, SeStream<GATHER ? 1 : NQ>> *se_carry = nullptr,
                     const bool se_ready = false, const bool se_next = false, const bool narrow_in = false)
{                                     // -> true: this wave has left for good (nothing to do at later TTIs of the launch either)
    constexpr int MODE = MODE_X & 3;
    constexpr bool PE = (MODE_X & MODE_PE) != 0;
    static_assert(!PE || (MODE != MODE_RESET && !PERSIST && PACK == 1 && !MIX), "per-element scaling: the lean step / dense builds");
    static_assert(!(GATHER && MODE == MODE_DENSE), "a dense sched_decision reads whole rows: streaming only");
    // PACK = 2 (envs of at most 32 UEs, one-wave workgroups): the wave steps TWO envs, lanes 0-31 the first, lanes 32-63 the second --
    // `tid` is the lane within the env's half, everything per env (index, counters, episode, tile, LDS image, row addresses) is a
    // per-lane value that happens to be equal across a half, and the 16-lane slice groups are DPP rows 0 / 2 of the wave.
    static_assert(PACK == 1 || (PACK == 2 && MODE == MODE_STEP && !PERSIST), "packed waves: step launches only");
    constexpr int LW = WAVE / PACK;                  // lanes per env
    constexpr int GDEPTH = (GATHER && NQ == 0) ? 1 : 2;      // gather builds: NQ = 0 asks for one 8-RB group in flight instead of two
    // CARRY (persistent builds with registers to spare): a TTI that follows another TTI of the same env in the same chunk loads
    // NOTHING of the UE's state -- the previous TTI hands all of it over in registers (StepCarry) and has requested, behind its own
    // stores, what only the next TTI's position determines: the two window slots that push will give up, its traffic word, and
    // the two age-list entries behind the head (the UE step pops ~1 entry per TTI; each pop used to be a dependent load in the
    // middle of the step).  A workgroup's TTI is a chain of dependent round trips (1-2 us each under load); this removes the state
    // round trip and the age-list ones.  For the whole-row streaming build it also frees the way for the tile: memory operations
    // retire in issue order, so the burst of the next TTI's tile may only follow the TTI's last dependent load -- with the UE step
    // loading nothing the burst moves from behind the UE step to right behind the stream phase.
    constexpr bool CARRY = PERSIST && MODE == MODE_STEP && PACK == 1 && ((!GATHER && NQ >= 8) || (GATHER && RANENV_GATHER_CARRY != 0));
    // MIX (ranenv_core_kernel_mixed): a two-wave block steps either one env of more than 64 slice members with both waves, or --
    // `narrow` -- two envs of at most 64, one per wave, each wave a workgroup of its own: its own LDS image, lanes counted from its
    // own first lane, no block barrier (wg_sync(narrow))
    static_assert(!MIX || (PACK == 1 && MODE == MODE_STEP), "mixed blocks: step launches, one env per wave or per block");
    const bool narrow = MIX && narrow_in;
    __shared__ SharedCore<NP> shs[MIX ? 2 : PACK];
    int e_ = PACK == 2 ? e_in + (int)(threadIdx.x >> 5) : __builtin_amdgcn_readfirstlane(e_in);   // e_in: p.e0 + blockIdx.x (x PACK), or a persistent workgroup's env
    int tid_ = PACK == 2 ? (int)(threadIdx.x & 31u) : (narrow ? (int)(threadIdx.x & 63u) : (int)threadIdx.x);
    SharedCore<NP> &sh = shs[PACK == 2 ? (threadIdx.x >> 5) : (narrow ? (threadIdx.x >> 6) : 0)];
    auto &xr = sh.xr;
    // (opaque to the optimiser: inside step_loop nothing derived from them is carried from one TTI to the next in registers)
    if constexpr (PACK == 1) asm volatile("" : "+s"(e_));
    asm volatile("" : "+v"(tid_));
    const int e = e_, tid = tid_;
    // (inside a persistent launch no thread ever leaves the body early -- there is no env mask inside a rollout and every wave
    // is needed again for the next env --, and the exits are compiled out: a divergent way out of the persistent loops would
    // make the loop-carried wave-uniform values divergent in the compiler's eyes)
    if (!PERSIST && PACK == 1 && p.env_mask != nullptr && p.env_mask[e] == 0) return true;  // uniform per workgroup
    const int S = p.S, U = p.U, R = p.R, D = p.D, Us = p.Us;
    const int W = 2 * Us + 9;
    auto uni = [](int v) { if constexpr (PACK == 2) return v; else return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [](long long v) {
        if constexpr (PACK == 2) return v;
        const unsigned lo32 = __builtin_amdgcn_readfirstlane((unsigned)v);
        const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
        return (long long)(((unsigned long long)hi32 << 32) | lo32);
    };
#if RANENV_DIAG == 12
    unsigned long long stamp_prev = 0;
#endif
    RANENV_STAMP(0);
#if RANENV_COLD_ARGS
    // The kernel's argument block, read in place: a field that only a late role needs is fetched there (one scalar load)
    // instead of sitting in -- or being spilled from -- SGPRs since kernel entry.  (The laundering keeps the compiler from
    // merging these loads with the by-value copy it loads up front.)
    typedef const __attribute__((address_space(4))) KP *kp_const_t;
    kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kc));
#define COLD(f) (kc->f)
#else
#define COLD(f) (p.f)
#endif
    ranenv_episode ep;
    int t, hlen, npush, se_pos, trf_pos;
    if (!warm) {
        ep = p.episodes[e];
        ep.scenario = uni(ep.scenario); ep.se_offset = uni(ep.se_offset); ep.trf_offset = uni(ep.trf_offset);
        ep.se_len = uni(ep.se_len); ep.trf_len = uni(ep.trf_len);
        ep.se_base = uni64(ep.se_base); ep.trf_base = uni64(ep.trf_base);
        t = (MODE == MODE_RESET) ? 0 : uni(ST_step_no(p)[e]);
        hlen = uni(ST_hist_len(p)[e]);
        npush = uni(ST_n_push(p)[e]);                    // kept in [0, D)
        // a position persisted under an older, longer trace must not index past the current one
        se_pos = (MODE == MODE_RESET) ? ep.se_offset : uni(ST_se_pos(p)[e]);
        trf_pos = (MODE == MODE_RESET) ? ep.trf_offset : uni(ST_trf_pos(p)[e]);
    } else {
        // (wave-uniform by construction; the readfirstlane costs nothing where the compiler already holds the value in an SGPR and
        // keeps the scalar uses below legal where its divergence analysis gave up on a value carried around the persistent loops)
        ep.scenario = uni(cy.ep.scenario); ep.se_offset = uni(cy.ep.se_offset); ep.trf_offset = uni(cy.ep.trf_offset);
        ep.se_len = uni(cy.ep.se_len); ep.trf_len = uni(cy.ep.trf_len); ep.reserved = 0;
        ep.se_base = uni64(cy.ep.se_base); ep.trf_base = uni64(cy.ep.trf_base);
        t = uni(cy.t); hlen = uni(cy.hlen); npush = uni(cy.npush); se_pos = uni(cy.se_pos); trf_pos = uni(cy.trf_pos);
    }
    const int sc = ep.scenario;
    se_pos = se_pos < ep.se_len ? se_pos : 0;
    trf_pos = trf_pos < ep.trf_len ? trf_pos : 0;
    const int hlen_old = hlen;                                // window length the allocation sees
    const bool clear_hist = MODE == MODE_RESET && (p.flags & RANENV_F_CLEAR_HISTORY_ON_RESET);
    if (clear_hist) hlen = 0;
    const int hlen_new = hlen < D ? hlen + 1 : D;
    const float *tile;
    const long long tile_no = ep.se_base + (long long)se_pos;
    if (!GATHER && p.se_tiles != nullptr) tile = p.se_tiles + (size_t)e * U * R;
    else tile = p.se_pool + (size_t)tile_no * (size_t)p.se_stride;

    // ---- loads, in the order they are needed: memory operations retire in issue order (vmcnt), so what the
    // allocation waits for (tables, UE state) is issued before the SE tile and does not queue behind it
    // Lane l owns UE lane_ue[l] of its scenario: the UEs that are in a slice first, the idle ones behind (the tables are
    // stored in that order).  A step in compact mode touches slice members only: a UE outside every slice receives no
    // traffic (the host made sure, idle_traffic_ok), is allocated nothing and is read by no observation, so its state
    // stays what the last reset left; the pushes its 10-TTI window misses meanwhile are made up for when it is stepped
    // again (catch-up below).  Waves that hold no slice member leave before the first barrier: at the headline size 76 %
    // of the scenarios have at most 64 UEs in slices, and their envs run one wave instead of two.
    const bool compact = MODE == MODE_STEP && p.compact != 0;
    const int lane = tid < U ? tid : U - 1;
    // (addressing: see row_at -- a uniform row per array, the lane's byte offset in one register)
    const size_t tb_row = ((size_t)sc * U + (compact ? (size_t)0 : (size_t)6 * (size_t)p.NSU)) * 4;
    const unsigned lane4 = (unsigned)lane * 4u;
#define TBL(f) row_at<PACK>(TB_##f(p), tb_row, lane4)
    int u, slc, ue_pos, pkt_size, max_pkts, max_age;
    if (!warm) {
        u = compact ? TBL(lane_ue) : lane;      // (set 1 is the identity: no load, and the state loads need not wait for it)
        slc = TBL(ue_slice); ue_pos = TBL(ue_pos);
        pkt_size = TBL(ue_pkt_size); max_pkts = TBL(ue_max_pkts); max_age = TBL(ue_max_age);
    } else {
        u = cy.u; slc = cy.slc; ue_pos = cy.ue_pos; pkt_size = cy.pkt_size; max_pkts = cy.max_pkts; max_age = cy.max_age;
        // (opaque, like the thread id above: what is derived from them -- LDS and row addresses -- is formed anew every TTI)
        asm volatile("" : "+v"(u), "+v"(slc), "+v"(ue_pos));
    }
#undef TBL
    const bool act = tid < U && !(compact && slc < 0);
    // (wave 0 stays: it runs the slice roles; inside a persistent launch every wave stays -- the launch's blocks have as many
    // waves as the env's class needs)
    if (!PERSIST && PACK == 1 && !warm && compact && __builtin_amdgcn_ballot_w64(act) == 0 && tid >= WAVE) return true;
    const size_t er4 = (size_t)e * U * 4, er8 = (size_t)e * U * 8;      // this env's row of a per-UE array of 4- / 8-byte elements
    const unsigned u4 = (unsigned)u * 4u, u8 = (unsigned)u * 8u;
#define UE4(f) row_at<PACK>(ST_##f(p), er4, u4)
#define UE8(f) row_at<PACK>(ST_##f(p), er8, u8)
    // window pushes of this env so far (wraps; only differences are used); index of the first push behind the last clearing
    const int ptot = uni(warm ? cy.ptot : ST_push_total(p)[e]);
    const int cmark = uni(warm ? cy.cmark : ST_clear_mark(p)[e]);
    int lastp = ptot;                               // index behind this UE's last push
    int total = 0, front = 0, front_rem = 0, fifo = 0, rb_start = 0, rb_count = 0;
    long long sum_age = 0, win_sent = 0, win_drop = 0;
    double sem_prev = 0.0;
    if (warm) { total = cy.total; win_sent = cy.win_sent; sem_prev = cy.sem_prev; }
    else {
        if (MODE != MODE_RESET) total = UE4(queue_pkts);
        if (!clear_hist) win_sent = UE8(win_sent);
    }
    // (this push's slots of the two rings: the addresses are formed where they are used, not carried through the step)
    auto ring_s = [&]() { return &row_at<PACK>(ST_ring_sent(p), ((size_t)e * D + npush) * U * 4, u4); };
    auto ring_d = [&]() { return &row_at<PACK>(ST_ring_drop(p), ((size_t)e * D + npush) * U * 4, u4); };
    int old_s = 0, old_d = 0;
    double traffic = 0.0;
    const bool gen_traffic = MODE != MODE_RESET && p.traffic_bits == nullptr && p.trf_gen != 0;
    const bool carried = CARRY && warm;
    auto rest_of_state = [&]() {
        if (carried) {           // (this lane pushed at the previous TTI: lastp == ptot already)
            sum_age = cy.sum_age; front = cy.front; front_rem = cy.front_rem; fifo = cy.fifo; win_drop = cy.win_drop;
            old_s = cy.pf_old_s; old_d = cy.pf_old_d;
            if (!gen_traffic) traffic = (double)cy.pf_traffic;
            return;
        }
        if (MODE != MODE_RESET) {
            sum_age = UE8(queue_age_sum);
            front = UE4(front); front_rem = UE4(front_rem); fifo = UE4(fifo);
        }
        if (!clear_hist) { win_drop = UE8(win_dropped); lastp = UE4(last_push); }
        if (hlen == D) { old_s = *ring_s(); old_d = *ring_d(); }
        if (MODE != MODE_RESET && !gen_traffic)
            traffic = p.traffic_bits ? row_at<PACK>(p.traffic_bits, er8, u8)
                                     : (double)row_at<PACK>(p.trf_pool, ((size_t)ep.trf_base + (size_t)trf_pos) * U * 4, u4);
    };
    // When the rest of the UE's state is requested: behind the stream by default (see RANENV_DEFER_STATE: registers), but at
    // entry in the build that has registers to spare (the whole-row queue of a batch at <= 2 waves per SIMD): there its round
    // trip -- 1.5-2 us of every chain when exposed -- runs under the allocation and the stream.
    constexpr int DEFER = (!GATHER && NQ >= 8) ? 0 : RANENV_DEFER_STATE;
    if constexpr (DEFER == 0) rest_of_state();
    if (MODE == MODE_STEP && !warm) sem_prev = UE8(se_mean);
    double sem_tile = 0.0;                          // gather: this tile's mean SE of UE u, from the sidecar
    if (GATHER) sem_tile = row_at<PACK>(p.se_mean_pool, (size_t)tile_no * U * 8, u8);
    // the scenario's slice tables, parked in LDS below by wave 0 (the other waves may have left): up to two words per lane
    int st_si0 = 0, st_si1 = 0, st_pi0 = 0, st_pi1 = 0; double st_pf = 0.0, st_sf = 0.0;
    if (tid < WAVE && !warm) {
        const unsigned t4 = (unsigned)tid * 4u, t8 = (unsigned)tid * 8u;
        if (tid < S * 8) st_si0 = row_at<PACK>(TB_slice_i32(p), (size_t)sc * S * 32, t4);
        if (tid + LW < S * 8) st_si1 = row_at<PACK>(TB_slice_i32(p), (size_t)sc * S * 32, t4 + LW * 4u);
        // (the intent parameters BY METRIC: the second block of the two tables, NS * S rows behind the first; NS * S = NSL / 16)
        const size_t by_metric = (size_t)(p.NSL / GRP) * 24;
        if (tid < S * 6) st_pi0 = row_at<PACK>(TB_param_i32(p), by_metric + (size_t)sc * S * 24, t4);
        if (tid + LW < S * 6) st_pi1 = row_at<PACK>(TB_param_i32(p), by_metric + (size_t)sc * S * 24, t4 + LW * 4u);
        if (tid < S * 3) st_pf = row_at<PACK>(TB_param_f64(p), by_metric + (size_t)sc * S * 24, t8);
        if (tid < S * 2) st_sf = row_at<PACK>(TB_slice_f64(p), (size_t)sc * S * 16, t8);
    }
    // device policy: this TTI's allocation may have been made at the end of the previous step
    bool pre = false;
    if (RANENV_LATE_BUILT && MODE == MODE_STEP && p.scores == nullptr && p.late != 0 && !warm) pre = uni(ST_alloc_gen(p)[e]) == p.alloc_gen;
#if RANENV_DIAG == 6 || RANENV_DIAG == 7 || RANENV_DIAG == 8    /* ablations: 8 = no allocation and no obs tail; 10 = allocation + tail only; 6 = entry + stream only (ranges from the stored allocation, sums written out); 7 = no allocation */
    if (MODE == MODE_STEP) pre = true;
#endif
    if (pre) {
        rb_start = UE4(next_rb_start); rb_count = UE4(next_rb_count);
#if RANENV_DIAG != 9 && RANENV_DIAG != 12
        if (tid < S) row_at<PACK>(ST_policy_scores(p), (size_t)e * S * 8, (unsigned)tid * 8u) = row_at<PACK>(ST_next_scores(p), (size_t)e * S * 8, (unsigned)tid * 8u);
#endif
    }
    const int episode_no = warm ? uni(cy.episode_no) : (gen_traffic ? uni(ST_episode_no(p)[e]) : 0);
    asm volatile("" ::: "memory");                 // keep the SE loads behind the loads above
    // SE_AHEAD (the whole-row build of a batch at <= 2 waves per SIMD): the queue lives in the caller's loop, and a TTI that is
    // followed by another one of the same env requests that TTI's tile before its own observation tail (below): the loads are
    // in flight through the tail, the next entry and the next allocation -- ~8 us of the chain -- and the stream phase finds them
    // landed.  The registers are there (2 waves per SIMD: 256 VGPRs), nothing else of the chain depends on the tile.
    constexpr bool SE_AHEAD = PERSIST && !GATHER && MODE == MODE_STEP && NQ >= 8;
    typedef std::conditional_t<PACK == 2, SeStreamLane<GATHER ? 1 : NQ>, SeStream<GATHER ? 1 : NQ>> SeQ;
    SeQ se_local;
    SeQ &se1 = SE_AHEAD ? *se_carry : se_local;
    const bool se_quad = !GATHER && p.se_quad != 0 && p.se_tiles == nullptr;   // (explicit per-step tiles are RB-major)
    if (!GATHER && !(SE_AHEAD && se_ready)) se1.init(tile, U, u, R, se_quad);          // lane = UE
    asm volatile("" ::: "memory");
    // wave 0 zeroes what can be read of the per-slice rows (NP positions of S slices: nothing reads further) and parks the tables.
    // A warm TTI finds both as it needs them: the tables are the scenario's, and every role writes a slice's rows at its
    // members' positions only, so what lies beyond them is still the zeros of the launch's first TTI.
    if (tid < WAVE && !warm) {
        for (int i = tid; i < S * 4 * NP; i += LW) {
            const int sl0 = i / (4 * NP), rem = i - sl0 * (4 * NP), k0 = rem / NP, j0 = rem - k0 * NP;
            sh.rows[sl0][k0 * NP + j0] = 0.0;
        }
        for (int i = tid; i < S * NP; i += LW) { const int sl0 = i / NP, j0 = i - sl0 * NP; sh.cnt[sl0][j0] = 0; }
        if (tid < S * 2) (&sh.msk[0][0])[tid] = 0u;
        if (tid < S * 8) (&sh.si[0][0])[tid] = st_si0;
        if (tid + LW < S * 8) (&sh.si[0][0])[tid + LW] = st_si1;
        if (tid < S * 6) (&sh.pi[0][0])[tid] = st_pi0;
        if (tid + LW < S * 6) (&sh.pi[0][0])[tid + LW] = st_pi1;
        if (tid < S * 3) (&sh.pf[0][0])[tid] = st_pf;
        if (tid < S * 2) (&sh.sf[0][0])[tid] = st_sf;
    }
    if (!warm) wg_sync(narrow);            // (a warm TTI starts behind step_loop's barrier)
    RANENV_STAMP(1);

    // ---- (0) this TTI's allocation --------------------------------------------------------------------
    if (MODE == MODE_STEP && !pre)
        alloc_front<NP, PACK>(p, sh, tid, e, hlen_old, act && slc >= 0, slc, ue_pos, total, max_pkts, pkt_size, win_sent, sem_prev,
                    rb_start, rb_count, ST_policy_scores(p), narrow);
    RANENV_STAMP(2);

    // MultSliceTraffic.step (traffics/mult_slice.py:24-32) drawn instead of replayed: Poisson(slice Mbps) * 1e6 bits for the UEs of a
    // slice that has a request, 0 elsewhere.  A function of (env, episode, TTI, UE) alone.
    auto draw_traffic = [&]() -> double {
        if (!(slc >= 0 && sh.si[slc][1] != 0 && sh.sf[slc][1] > 0.0)) return 0.0;
        unsigned rnd[4];
        philox4x32_10((unsigned)(COLD(env_id_base) + e), (unsigned)episode_no, (unsigned)t, (unsigned)u,
                      (unsigned)COLD(trf_seed), (unsigned)(COLD(trf_seed) >> 32), rnd);
        const size_t row = (size_t)sc * S + slc;
        const int k = poisson_draw(COLD(pois_cdf) + row * 256, COLD(pois_guide) + row * 64, ((unsigned long long)rnd[1] << 32) | rnd[0]);
        return (double)k * 1e6;
    };
    // (CARRY builds: drawn here, ahead of the stream, so that the UE step behind it makes no dependent load -- the table look-ups
    // retire behind the tile, which the stream phase waits for anyway)
    if (CARRY && gen_traffic && act) traffic = draw_traffic();

    // ---- (1) SE row sums -------------------------------------------------------------------------
    double my_full = 0.0, my_part = 0.0;
    auto hook = [&]() {
        if constexpr (DEFER == 1) rest_of_state();
    };
    if constexpr (GATHER) {
#if RANENV_GATHER_STATE_FIRST
        rest_of_state();          // requested ahead of the gather: both latencies run together
#endif
        if (MODE == MODE_STEP) my_part = gather_part<PACK, GDEPTH, PE>(tile, U * p.se_rp * 4, u * p.se_rp * 4, R, (unsigned)rb_start, (unsigned)rb_count, PE ? COLD(bw_per_rb) : 1.0);
    } else if constexpr (MODE == MODE_STEP) {
        const unsigned us1 = (unsigned)rb_start, uc1 = (unsigned)rb_count;
#if RANENV_DIAG == 1 || RANENV_DIAG == 10
        { float x0[8]; se1.take(0, x0, 0); my_full = (double)x0[0] + (double)us1; my_part = (double)uc1; }
#elif RANENV_DIAG == 2
        row_sums(se1, R, [=](int r) { return false; }, my_full, my_part, hook); my_part = (double)(us1 + uc1);
#else
        row_sums<PE>(se1, R, [=](int r) { return ((unsigned)r - us1) < uc1; }, my_full, my_part, hook, PE ? COLD(bw_per_rb) : 1.0);
#endif
    } else if constexpr (MODE == MODE_DENSE) {
        const uint8_t *mrow = p.dense + ((size_t)e * U + u) * R;
        row_sums<PE>(se1, R, [=](int r) { return mrow[r] != 0; }, my_full, my_part, hook, PE ? COLD(bw_per_rb) : 1.0);
    } else {
        row_sums(se1, R, [](int) { return false; }, my_full, my_part, hook);
    }
    if constexpr (DEFER == 2) {
        if (!(GATHER && RANENV_GATHER_STATE_FIRST))
            rest_of_state();        // after the stream: its latency is exposed, its registers were free for the queue
    }
    if constexpr (SE_AHEAD && CARRY) {
        // The next TTI's tile of this env (the position it will derive itself: cy.se_pos below), requested as soon as the queue's
        // registers are free: in a carried TTI nothing the UE step needs is loaded behind it (in the first TTI of a chunk the UE step's
        // age-list loads queue behind the burst: once per chunk).
        if (se_next) {
            const int pos_next = se_pos + 1 >= ep.se_len ? 0 : se_pos + 1;
            asm volatile("" ::: "memory");
            se1.init(p.se_pool + (size_t)(ep.se_base + (long long)pos_next) * (size_t)p.se_stride, U, u, R, se_quad);
            asm volatile("" ::: "memory");
        }
    }
    RANENV_STAMP(3);
    wg_sync(narrow);        // every thread is done with the allocation's use of the per-slice rows
    RANENV_STAMP(4);

    // ---- (2) UEs.step for UE tid -------------------------------------------------------------------
    // np.isclose(previous buffer occupancy, 0) (common.py:108-118): occupancy = total / max_pkts.  Exact
    // shortcuts: an empty queue is 0; a queue above 2e-8 * max_pkts is not close to 0; in between, divide.
    double sem_new = 0.0;
    int sent_u = 0, drop_u = 0;              // this UE's packets sent / dropped (episode metrics)
    bool prev_empty = total == 0;
    if (total != 0 && !((double)total > 2e-8 * (double)max_pkts)) prev_empty = d_isclose((double)total / (double)max_pkts, 0.0);
#if RANENV_DIAG == 6
    if (act) { UE8(se_mean) = my_full; UE8(queue_age_sum) = (long long)my_part; }
#endif
#if RANENV_DIAG == 6
    if (false) {
#elif RANENV_DIAG == 3 || RANENV_DIAG == 5 || RANENV_DIAG == 10
    if (act && my_full < -1.0) {
#else
    if (act) {
#endif
        if (MODE == MODE_DENSE) {
            const uint8_t *mrow = p.dense + ((size_t)e * U + u) * R;
            bool seen = false;
            for (int r = 0; r < R; r++) {
                if (mrow[r] != 0) { rb_count++; if (!seen) { rb_start = r; seen = true; } }
            }
        }
        const double se_mean_new = GATHER ? sem_tile : ddiv(my_full, (double)R), se_part = my_part;
        int dropped = 0, sent = 0, pkt_in = 0, pkt_thr = 0;     // all < 2^31 (host validates the packet counts)
        int adm_now = 0;                                        // packets admitted at this TTI (its age-list entry, if any)
        if (MODE != MODE_RESET) {
            const double psz = (double)pkt_size;
            // floor of non-negative values; v_cvt_i32_f64 truncates and saturates (host validates < 2^31)
            if (gen_traffic && !CARRY) traffic = draw_traffic();
            pkt_thr = (int)ddiv(PE ? se_part : se_part * COLD(bw_per_rb), psz);
            pkt_in = (int)ddiv(traffic, psz);
            const int L = p.L;
            // The queue is FIFO, so the age histogram Buffer keeps is exactly a list of (arrival TTI,
            // packets) entries in arrival order.  ring[k] holds entry k of a circular list (head index +
            // entry count per UE); only TTIs that admitted packets make an entry, so expiring / draining
            // costs one load per consumed entry and never a scan.
            int2 *ring_env = ST_age_ring(p) + (size_t)e * L * U;        // (uniform; entry k of this UE at [k * U + u])
            int head = fifo & 0xffff, nent = (int)((unsigned)fifo >> 16);
            auto pop_head = [&]() { nent--; head = head + 1 == L ? 0 : head + 1; };
            // the k-th pop of a TTI that leaves an older entry at the head needs entry k behind the head the TTI started with: the
            // first two were requested at the end of the previous TTI (carried), only a UE that pops more loads here
            int loads = 0;
            const int pf1x = cy.pf1.x, pf1y = cy.pf1.y, pf2x = cy.pf2.x, pf2y = cy.pf2.y;      // (by value: StepCarry must stay in registers)
            auto load_head = [&]() {
                loads++;
                if (carried && loads <= 2) { front = loads == 1 ? pf1x : pf2x; front_rem = loads == 1 ? pf1y : pf2y; }
                else { const int2 en = row_at<PACK>(ring_env, 0, (unsigned)(head * U + u) * 8u); front = en.x; front_rem = en.y; }
            };
            if (nent > 0 && front == t - max_age - 1) {         // receive: the bin older than max_age expires
                dropped += front_rem; total -= front_rem; sum_age -= (long long)max_age * front_rem;
                front_rem = 0;
                pop_head();
                if (nent > 0) load_head();
            }
            sum_age += total;                                     // everything left ages one TTI
            const int space = max_pkts - total;                   // arrivals admitted up to capacity
            const int adm = pkt_in < space ? pkt_in : space;
            dropped += pkt_in - adm;
            if (adm > 0) {
                int tail = head + nent; tail = tail >= L ? tail - L : tail;
                nt_store<GATHER>(row_at<PACK>(ring_env, 0, (unsigned)(tail * U + u) * 8u), make_int2(t, adm));
                if (nent == 0) { front = t; front_rem = adm; }
                nent++;
                total += adm;
            }
            int cap = pkt_thr;                                    // send: drain oldest first
            while (cap > 0 && nent > 0) {
                const int take = cap < front_rem ? cap : front_rem;
                front_rem -= take; total -= take; cap -= take; sent += take;
                sum_age -= (long long)(t - front) * take;
                if (front_rem == 0) {
                    pop_head();
                    if (nent > 0) {
                        if (nent == 1 && adm > 0) { front = t; front_rem = adm; }   // this TTI's entry
                        else load_head();
                    }
                }
            }
            fifo = head | (nent << 16);
            adm_now = adm;
        }
        // A UE that was not stepped for a while (outside every slice, compact mode) missed the pushes [lastp, ptot): each
        // would have pushed zeros.  Made up for here, oldest first; only the last D matter.  A push at index q finds the
        // window min(D, q - cmark) long, and only a full window gives up what its slot holds.  (Empty in the steady state.)
        if (!clear_hist && lastp != ptot) {
            int q = ptot - lastp > D ? ptot - D : lastp;
            if (ptot - q == D) { old_s = 0; old_d = 0; }         // the slot of this push is among them: it will hold a zero
            for (; q != ptot; q++) {
                int slot = npush - (ptot - q); slot += slot < 0 ? D : 0;
                const unsigned so = (unsigned)(slot * U + u) * 4u;
                int32_t *qs = &row_at<PACK>(ST_ring_sent(p), (size_t)e * D * U * 4, so), *qd = &row_at<PACK>(ST_ring_drop(p), (size_t)e * D * U * 4, so);
                if (q - cmark >= D) { win_sent -= *qs; win_drop -= *qd; }
                *qs = 0; *qd = 0;
            }
        }
        // A clearing reset also empties the UE's ring: the catch-up above subtracts what a slot holds on the assumption that
        // it belongs to the current window era, and a UE that pushed fewer than D times since the clear and then sat out
        // more than D pushes would otherwise give up values of the era before (a reset steps every UE: full width).
        if (MODE == MODE_RESET && clear_hist) {
            for (int k = 0; k < D; k++) {
                row_at<PACK>(ST_ring_sent(p), ((size_t)e * D + k) * U * 4, u4) = 0;
                row_at<PACK>(ST_ring_drop(p), ((size_t)e * D + k) * U * 4, u4) = 0;
            }
        }
        // push into the 10-TTI window (IBSched.last_unformatted_obs.appendleft, ib_sched.py:64)
        win_sent += sent - old_s; win_drop += dropped - old_d;
        nt_store<GATHER>(*ring_s(), (int32_t)sent); nt_store<GATHER>(*ring_d(), (int32_t)dropped);
        UE4(last_push) = ptot + 1;
        UE4(queue_pkts) = total; UE8(queue_age_sum) = sum_age;
        UE4(front) = front; UE4(front_rem) = front_rem; UE4(fifo) = fifo;
        UE8(win_sent) = win_sent; UE8(win_dropped) = win_drop;
        UE4(pkt_effective_thr) = (int32_t)sent; UE4(dropped_pkts) = (int32_t)dropped;
        if (!(COLD(flags) & RANENV_F_NO_RAW_OUTPUT)) {
            nt_store<GATHER>(UE4(pkt_incoming), (int32_t)pkt_in); nt_store<GATHER>(UE4(pkt_throughputs), (int32_t)pkt_thr);
        }
        sent_u = sent; drop_u = dropped;

        UE8(se_mean) = se_mean_new; sem_new = se_mean_new;
        UE4(rb_start) = rb_start; UE4(rb_count) = rb_count;
        if constexpr (CARRY) {
            // Hand-over to the next TTI of this chunk (if there is one): the state in registers, and -- requested here, behind this
            // TTI's stores (catch-up and push included: a load behind a store of the same lane sees it) -- what its position determines.
            cy.sum_age = sum_age; cy.win_drop = win_drop; cy.front = front; cy.front_rem = front_rem; cy.fifo = fifo;
            int ps = 0, pd = 0, pt = 0;
            int2 e1 = make_int2(0, 0), e2 = make_int2(0, 0);
            if (se_next) {
                const int np1 = npush + 1 == D ? 0 : npush + 1;
                if (hlen_new == D) {                            // the next push finds the window full: it gives up what slot np1 holds
                    if (D == 1) { ps = sent; pd = dropped; }    // (a one-deep window: the slot this TTI has just written)
                    else {
                        ps = row_at<PACK>(ST_ring_sent(p), ((size_t)e * D + np1) * U * 4, u4);
                        pd = row_at<PACK>(ST_ring_drop(p), ((size_t)e * D + np1) * U * 4, u4);
                    }
                }
                if (!gen_traffic) {
                    const int tp1 = trf_pos + 1 >= ep.trf_len ? 0 : trf_pos + 1;
                    pt = row_at<PACK>(p.trf_pool, ((size_t)ep.trf_base + (size_t)tp1) * U * 4, u4);
                }
                const int L = p.L, head = fifo & 0xffff, nent = (int)((unsigned)fifo >> 16);
                const int h1 = head + 1 >= L ? head + 1 - L : head + 1, h2 = head + 2 >= L ? head + 2 - L : head + 2;
                int2 *ring_env = ST_age_ring(p) + (size_t)e * L * U;
                // (this TTI's own entry is the list's last and untouched unless it is the head: taken from registers, not loaded back)
                const bool own1 = adm_now > 0 && nent == 2, own2 = adm_now > 0 && nent == 3;
                if (own1) e1 = make_int2(t, adm_now); else if (nent > 1) e1 = row_at<PACK>(ring_env, 0, (unsigned)(h1 * U + u) * 8u);
                if (own2) e2 = make_int2(t, adm_now); else if (nent > 2) e2 = row_at<PACK>(ring_env, 0, (unsigned)(h2 * U + u) * 8u);
            }
            cy.pf_old_s = ps; cy.pf_old_d = pd; cy.pf_traffic = pt; cy.pf1 = e1; cy.pf2 = e2;
        }
        const double occ_new = ddiv((double)total, (double)max_pkts);
        const double lat_new = total > 0 ? ddiv((double)sum_age, (double)total) : 0.0;
        // ---- intent drift of this UE (agents/common.py:68-340) ----------------------------------------
        // The slice lists up to three parameters in its own order; the lanes of a wave belong to different
        // slices, so "for each parameter: switch on its metric" would run all three formulas three times.
        // Instead: find this slice's (value, operator) for each metric, then run each formula once.
        double dres[3] = {0.0, 0.0, 0.0};
        // slice row for the drift, from the tables parked in LDS (read here, not at the top of the role: 17 registers
        // that would otherwise be alive through the buffer update)
        asm volatile("" ::: "memory");
        int has_req = 0, bsize = 1, blat = 1, msg = 1;
        bool dec[3] = {false, false, false};
        double val[3] = {1.0, 1.0, 1.0};
        int opm[3] = {0, 0, 0};
        if (slc >= 0) {
            const int *si = sh.si[slc];
            has_req = si[1]; bsize = si[3]; blat = si[4]; msg = si[5];
            // (value, operator) of each metric: resolved on the host when the scenario was loaded (a later parameter for the same metric
            // overwrites, :132-335), parked by metric
#pragma unroll
            for (int m = 0; m < 3; m++) { dec[m] = sh.pi[slc][2 * m] != 0; opm[m] = sh.pi[slc][2 * m + 1]; val[m] = sh.pf[slc][m]; }
        }
        if (slc >= 0 && has_req) {
            const double o = COLD(over);
            // Each formula is "intent met ? a / b : -(c / d)" (plus a cap at 1 when over-fulfilled): one division
            // on the selected operands gives the bits of whichever arm is taken.
            if (dec[RANENV_METRIC_THROUGHPUT]) {
                const double value = val[RANENV_METRIC_THROUGHPUT];
                double x = ddiv((double)sent * (double)msg, 1e6);                       // common.py:25-31
                bool zero = d_isclose(occ_new, 0.0);                                    // :100-119
                if (hlen_new > 1) zero = zero || prev_empty;
                if (zero) x = value * (1.1 + o);
                const bool met = d_apply_op(opm[RANENV_METRIC_THROUGHPUT], x, value);
                const double q = (met ? x - value : value - x) / (met ? value * o : value);
                dres[RANENV_METRIC_THROUGHPUT] = met ? ((x > value * (1.0 + o)) ? 1.0 : q) : -q;
            }
            if (dec[RANENV_METRIC_RELIABILITY]) {
                const double value = val[RANENV_METRIC_RELIABILITY];
                const double dw = (double)win_drop, sw = (double)win_sent;              // :32-53
                const double buffer_pkts = occ_new * (double)bsize + dw + sw;
                const double x = buffer_pkts != 0.0 ? dw / buffer_pkts : 0.0;
                const double band = (100.0 - value) / 100.0;
                const bool met = d_apply_op(opm[RANENV_METRIC_RELIABILITY], 100.0 * (1.0 - x), value);
                const double q = (met ? band - x : x - band) / (met ? band * o : value / 100.0);
                dres[RANENV_METRIC_RELIABILITY] = met ? ((x < band * (1.0 - o)) ? 1.0 : q) : -q;
            }
            if (dec[RANENV_METRIC_LATENCY]) {
                const double value = val[RANENV_METRIC_LATENCY];
                const double x = lat_new;                                               // :58-61
                const bool met = d_apply_op(opm[RANENV_METRIC_LATENCY], x, value);
                const double q = (met ? value - x : x - value) / (met ? value * o : (double)blat - value);
                dres[RANENV_METRIC_LATENCY] = met ? ((x < value * (1.0 - o)) ? 1.0 : q) : -q;
            }
        }

        if (slc >= 0) {
            // rows for this TTI's observation
            srow(sh, slc, 0)[ue_pos] = dres[0]; srow(sh, slc, 1)[ue_pos] = dres[1]; srow(sh, slc, 2)[ue_pos] = dres[2];
            srow(sh, slc, 3)[ue_pos] = se_mean_new;
            sh.cnt[slc][ue_pos] = rb_count;
            if (COLD(obs_intra) && ue_pos < Us) {                                          // per-UE entries (:186-200)
#if RANENV_OBS_STAGE
                float *oa = sh.ob_intra + slc * W;
#else
                float *oa = COLD(obs_intra) + ((size_t)e * S + slc) * W;
#endif
                oa[9 + ue_pos] = (float)occ_new;
                oa[9 + Us + ue_pos] = (float)ddiv(se_mean_new, COLD(norm_se));
            }
        }
    }
    if (MODE != MODE_RESET && (RANENV_METRICS && COLD(acc) != nullptr)) {
        // episode metrics: packet totals of the env, one add per wave (integers in doubles: exact in any order)
        const double ws = PACK == 2 ? half_sum_f64((double)sent_u) : wave_sum_f64((double)sent_u);
        const double wd = PACK == 2 ? half_sum_f64((double)drop_u) : wave_sum_f64((double)drop_u);
        if ((tid & (LW - 1)) == 0) { acc_add(COLD(acc) + (size_t)e * 8 + 6, ws); acc_add(COLD(acc) + (size_t)e * 8 + 7, wd); }
    }
    RANENV_STAMP(5);
    if constexpr (SE_AHEAD && !CARRY) {
        if (se_next) {         // the next TTI's tile of this env (the position it will derive itself: cy.se_pos below)
            const int pos_next = se_pos + 1 >= ep.se_len ? 0 : se_pos + 1;
            asm volatile("" ::: "memory");           // (behind this TTI's state stores in program order: they need no register)
            se1.init(p.se_pool + (size_t)(ep.se_base + (long long)pos_next) * (size_t)p.se_stride, U, u, R, se_quad);
            asm volatile("" ::: "memory");
        }
    }
    wg_sync(narrow);
    RANENV_STAMP(6);
    do {
#if RANENV_DIAG == 4 || RANENV_DIAG == 5 || RANENV_DIAG == 6 || RANENV_DIAG == 8   /* ablation: no observation tail, but the per-env counters move on (tiles keep changing) */
#if RANENV_DIAG == 6
    if (true) {
#else
    if (my_full >= -1.0) {
#endif
        if (tid == 0) {
            ST_step_no(p)[e] = (MODE == MODE_RESET) ? 0 : t + 1; ST_hist_len(p)[e] = hlen_new; ST_n_push(p)[e] = npush + 1 == D ? 0 : npush + 1;
            ST_push_total(p)[e] = ptot + 1;
            ST_se_pos(p)[e] = (MODE == MODE_RESET) ? ep.se_offset : (se_pos + 1 >= ep.se_len ? 0 : se_pos + 1);
            ST_trf_pos(p)[e] = (MODE == MODE_RESET) ? ep.trf_offset : (trf_pos + 1 >= ep.trf_len ? 0 : trf_pos + 1);
        }
        break;
    }
#endif
    if (tid >= WAVE) break;                  // (3) is done by wave 0
    // The four means of every slice over its UEs -- the three drift rows and the SE row (common.py:343-378, ib_sched.py:146-157)
    // -- one per lane (lane = 16 * row + sorted position) instead of four one after the other on 16 lanes: they are
    // independent chains of LDS reads and additions, and this wave has nothing else to issue.
    for (int pass4 = 0; pass4 < PACK; pass4++) {      // (a packed env has two 16-lane rows: the four means in two passes)
        const int row4 = (tid >> 4) + 2 * pass4, sp4 = tid & (GRP - 1);
        double mean4 = 0.0;
        int s4 = 0, n4 = 0;
        if (sp4 < S) { s4 = sh.si[sp4][7]; n4 = sh.si[s4][2]; }
        const double sum4 = np_sum_lds<NP>(srow(sh, s4, row4), n4);
        if (n4 > 0) mean4 = ddiv(sum4, (double)n4);
        xr[row4][sp4] = mean4;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tid >= GRP) break;

    // ---- thread t < 16: slice at sorted position t (ib_sched.py:91) --------------------------------
    const int spos = tid;
    const double mean_row[4] = {xr[0][tid], xr[1][tid], xr[2][tid], xr[3][tid]};
    xr[0][tid] = 0.0; xr[1][tid] = 0.0;      // the reward's rows (filled by slice index below) start from zero
    const bool ok = spos < S;
    double sv[3] = {-2.0, -2.0, -2.0};
    int s = 0, active = 0;
    double priority_tab = 0.0;
    if (ok) {
        s = sh.si[spos][7];
        const int *si = sh.si[s];
        active = si[0];
        const int has_req = si[1], n = si[2];
        int rbs_s = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) rbs_s += sh.cnt[s][k];
        priority_tab = sh.sf[s][0];
        const double traffic_tab = sh.sf[s][1];
        if (n > 0 && has_req) {                                                    // common.py:343-378
#pragma unroll
            for (int m = 0; m < 3; m++) sv[m] = sh.pi[s][2 * m] != 0 ? mean_row[m] : sv[m];      // (declared metrics: the table is by metric)
        }
        const double traffic_req = active == 1 ? traffic_tab : 0.0;                // ib_sched.py:125-134
        const double priority = n != 0 ? priority_tab : 0.0;                       // :135-141
        double am[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {                                              // :142-145
            const bool undeclared = d_isclose(sv[m], -2.0);
            am[m] = undeclared ? 0.0 : 1.0;
            sv[m] = undeclared ? 0.0 : sv[m];
        }
        const double se_slice = n > 0 ? mean_row[3] : 0.0;                                      // :146-157
        const float o0 = (float)sv[0], o1 = (float)sv[1], o2 = (float)sv[2];
        const float a0 = (float)am[0], a1 = (float)am[1], a2 = (float)am[2];
        const float tr = (float)ddiv(traffic_req, COLD(norm_traffic)), nu = (float)ddiv((double)n, COLD(norm_ues));
        if (COLD(obs_inter)) {                                                         // :160-173
#if RANENV_OBS_STAGE
            float *oi = sh.ob_inter + spos * 10;
#else
            float *oi = COLD(obs_inter) + ((size_t)e * S + spos) * 10;
#endif
            oi[0] = o0; oi[1] = o1; oi[2] = o2; oi[3] = a0; oi[4] = a1; oi[5] = a2;
            oi[6] = (float)priority; oi[7] = tr; oi[8] = nu; oi[9] = (float)ddiv(se_slice, COLD(norm_se));
        }
        if (COLD(obs_intra)) {
#if RANENV_OBS_STAGE
            float *oa = sh.ob_intra + s * W;
#else
            float *oa = COLD(obs_intra) + ((size_t)e * S + s) * W;
#endif
            oa[0] = o0; oa[1] = o1; oa[2] = o2; oa[3] = a0; oa[4] = a1; oa[5] = a2;
            oa[6] = (float)ddiv((double)rbs_s, (double)R); oa[7] = tr; oa[8] = nu;
            for (int k = n; k < Us; k++) { oa[9 + k] = 0.0f; oa[9 + Us + k] = 0.0f; }
        }
        // player_{s+1} reward (common.py:428-437)
        double r = 0.0; int cnt = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            if (am[m] > 0.0) { r = (cnt == 0 || sv[m] < r) ? sv[m] : r; cnt++; }
        }
        if (COLD(reward)) row_at<PACK>(COLD(reward), (size_t)e * (S + 1) * 8, (unsigned)(s + 1) * 8u) = cnt > 0 ? r : 0.0;
        if (MODE == MODE_RESET) {
            ST_mask_inter(p)[(size_t)e * S + s] = (int8_t)active;
            for (int k = 0; k < Us; k++) ST_mask_intra(p)[((size_t)e * S + s) * Us + k] = k < n ? 1 : 0;
        }
        // active_observations / slice_priorities indexed by slice (common.py:389-408)
        double mn = 0.0; int cntm = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const double v = sv[m];
            if (!d_isclose(v, -2.0)) { mn = (cntm == 0 || v < mn) ? v : mn; cntm++; }
        }
        xr[0][s] = active ? (cntm > 0 ? mn : 1.0) : 0.0;
        xr[1][s] = active ? priority_tab : 0.0;
    }
    // only one wave is left: LDS traffic between its lanes needs an LDS wait, not a barrier
    auto wave_sync = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    wave_sync();
    // ---- player_0 reward (common.py:409-427) ------------------------------------------------------
    // (how many slices are in violation, all / priority ones: every slice lane holds its own entry -- two ballots instead of a loop over
    // the LDS rows in every lane)
    const double my_ao = xr[0][tid], my_pr = xr[1][tid];
    const unsigned row_sh = PACK == 2 ? (threadIdx.x & 32u) : 0u;
    const int n_neg = __popc((unsigned)((__ballot(tid < S && my_ao < 0.0) >> row_sh) & 0xffffull));
    const int n_prio_neg = __popc((unsigned)((__ballot(tid < S && my_pr * my_ao < 0.0) >> row_sh) & 0xffffull));
    const int mode_sel = n_neg == 0 ? 0 : (n_prio_neg != 0 ? 1 : 2);
    // episode metrics: distance to fulfilment = sum of the negative slice drifts (entries beyond S are 0), all slices and
    // priority slices (priority is 0 or 1), as a fixed tree over the 16 lanes
    double dist = 0.0, prio_dist = 0.0;
    if (MODE != MODE_RESET && (RANENV_METRICS && COLD(acc) != nullptr)) {
        dist = row16_sum_f64(fmin(my_ao, 0.0)); prio_dist = row16_sum_f64(fmin(my_ao * my_pr, 0.0));
    }
    const bool my_sel = tid < S && (mode_sel == 0 ? true : (mode_sel == 1 ? (my_ao * my_pr < 0.0) : (my_ao < 0.0)));
    const unsigned gm = (unsigned)((__ballot(my_sel) >> (PACK == 2 ? (threadIdx.x & 32u) : 0u)) & 0xffffull);
    const int m_sel = __popc(gm), cslot = __popc(gm & ((1u << tid) - 1u));
    xr[2][tid] = 0.0;
    wave_sync();
    if (my_sel) xr[2][cslot] = my_ao;             // selected entries in slice order (np.mean of a[mask])
    wave_sync();
    if (tid == 0) {
        double rew = np_sum_lds<NP>(xr[2], m_sel) / (double)m_sel;
        if (mode_sel == 1) rew -= 1.0;
        if (COLD(reward)) COLD(reward)[(size_t)e * (S + 1)] = rew;
        // Episode metrics (ranenv_enable_metrics): running sums of what the paper's evaluation reads per TTI
        // (results/gen_results.py:874-1022: slices in violation, distance to fulfilment, all slices / priority slices
        // only), of the inter-slice reward and of the packet totals.  One writer per env and TTI; fire-and-forget adds.
        if ((RANENV_METRICS && COLD(acc) != nullptr)) {
            double *a = COLD(acc) + (size_t)e * 8;
            if (MODE == MODE_RESET) {
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = 0.0;
            } else {
                acc_add(a + 0, 1.0); acc_add(a + 1, rew); acc_add(a + 2, (double)n_neg); acc_add(a + 3, (double)n_prio_neg);
                acc_add(a + 4, dist); acc_add(a + 5, prio_dist);       // ([6], [7]: by the UE role, one add per wave)
            }
        }
        // per-env counters: everything was read as a scalar at kernel entry (hlen already reflects a cleared window)
        const int step_new = (MODE == MODE_RESET) ? 0 : t + 1;
        ST_step_no(p)[e] = step_new;
        ST_hist_len(p)[e] = hlen_new;
        ST_n_push(p)[e] = npush + 1 == D ? 0 : npush + 1;
        ST_push_total(p)[e] = ptot + 1;
        // the first push of the current window era; kept within 2 D of the counter, which is as good as exact (a window is
        // full after D pushes) and survives the counter's wrap-around
        if (clear_hist) ST_clear_mark(p)[e] = ptot;
        else if (ptot + 1 - cmark > 2 * D) ST_clear_mark(p)[e] = ptot + 1 - 2 * D;
        if (MODE == MODE_RESET) { ST_se_pos(p)[e] = ep.se_offset; ST_trf_pos(p)[e] = ep.trf_offset; }
        else {
            ST_se_pos(p)[e] = se_pos + 1 >= ep.se_len ? 0 : se_pos + 1;
            ST_trf_pos(p)[e] = trf_pos + 1 >= ep.trf_len ? 0 : trf_pos + 1;
        }
        const int max_steps_e = COLD(max_steps_env) ? COLD(max_steps_env)[e] : COLD(max_steps);
        if (COLD(done)) COLD(done)[e] = (MODE != MODE_RESET && step_new >= max_steps_e) ? 1 : 0;
    }
    } while (0);
#if RANENV_OBS_STAGE
    // wave 0 writes the staged observation rows out, a float per lane and whole lines per instruction (the per-UE entries were
    // staged before the barrier in front of (3), the per-slice ones by this wave's first 16 lanes just now)
    if (tid < WAVE && (COLD(obs_inter) || COLD(obs_intra))) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (COLD(obs_inter)) {
            float *dst = COLD(obs_inter) + (size_t)e * S * 10;
            for (int i = tid; i < S * 10; i += LW) nt_store<GATHER>(row_at<PACK>(dst, 0, (unsigned)i * 4u), sh.ob_inter[i]);
        }
        if (COLD(obs_intra)) {
            float *dst = COLD(obs_intra) + (size_t)e * S * W;
            for (int i = tid; i < S * W; i += LW) nt_store<GATHER>(row_at<PACK>(dst, 0, (unsigned)i * 4u), sh.ob_intra[i]);
        }
    }
#endif
    RANENV_STAMP(7);

    // ---- (0') the next TTI's allocation, from the state this step leaves behind ----------------------
    bool late = false;
    if (RANENV_LATE_BUILT && MODE == MODE_STEP) late = p.scores == nullptr && (p.late == 2 || (p.late == 1 && (((unsigned)e * 0x9E3779B1u) >> 16 & 1u)));
#if RANENV_DIAG == 6 || RANENV_DIAG == 7 || RANENV_DIAG == 8
    if (MODE == MODE_STEP) late = false;
#endif
    if (late) {
        wg_sync(narrow);                     // (3) is done with the per-slice rows
        int ns = 0, nc = 0;
        alloc_front<NP, PACK>(p, sh, tid, e, hlen_new, act && slc >= 0, slc, ue_pos, total, max_pkts, pkt_size, win_sent, sem_new,
                    ns, nc, ST_next_scores(p), narrow);
        if (act) { UE4(next_rb_start) = ns; UE4(next_rb_count) = nc; }
    }
    if (tid == 0) ST_alloc_gen(p)[e] = late ? p.alloc_gen : 0;
    RANENV_STAMP(8);
    if (MODE == MODE_STEP) {         // for the next TTI of this launch, if there is one (what tid 0 has just stored, and this lane's own)
        cy.ep = ep; cy.t = t + 1; cy.hlen = hlen_new; cy.npush = npush + 1 == D ? 0 : npush + 1;
        cy.se_pos = se_pos + 1 >= ep.se_len ? 0 : se_pos + 1; cy.trf_pos = trf_pos + 1 >= ep.trf_len ? 0 : trf_pos + 1;
        cy.ptot = ptot + 1; cy.cmark = (ptot + 1 - cmark > 2 * D) ? ptot + 1 - 2 * D : cmark; cy.episode_no = episode_no;
        cy.u = u; cy.slc = slc; cy.ue_pos = ue_pos; cy.pkt_size = pkt_size; cy.max_pkts = max_pkts; cy.max_age = max_age;
        cy.total = total; cy.win_sent = win_sent; cy.sem_prev = act ? sem_new : sem_prev;
    }
    return false;
#undef UE4
#undef UE8
}

// Several TTIs of one env in one launch (ranenv_rollout with a device policy, no episode end in between): the workgroup
// steps its env again as soon as it is done, from the state it has just written (its own CU's L1 / L2 hold it), instead
// of ending and being launched again.  Between TTIs without a warm entry: every store of the workgroup is out and visible to
// its other waves (full_sync: explicit vmcnt(0) + barrier; the waves of a workgroup share their CU's L1).
template <int MODE_X, int NQ, bool GATHER, int NP, bool MANY, int PACK = 1, bool MIX = false>      // MANY: the build for launches of more than one TTI
DEVFN void step_loop(const KP &p)
{
    constexpr int MODE = MODE_X & 3;
    if constexpr (MODE == MODE_STEP) {
        // MIX: which env(s) this block steps comes from the class lists (ranenv_persist_classify_kernel): the first p_count blocks take one
        // env of the wide class each, the others two envs of the narrow class, one per wave
        bool narrow = false;
        int e_mix = 0;
        if constexpr (MIX) {
            // (the grid is the launch's env count, an upper bound: blocks beyond wide + ceil(narrow / 2) leave at once)
            const int b = (int)blockIdx.x, n_wide = p.m_counts[1], n_narrow = p.m_counts[0];
            if (b < n_wide) e_mix = p.p_list[b];
            else {
                narrow = true;
                const int idx = 2 * (b - n_wide) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                if (idx >= n_narrow) return;              // (also: an odd number of narrow envs, the last block's second wave has none)
                e_mix = p.m_list[idx];
            }
        }
        // Every TTI reads the kernel's arguments in place, through a pointer the optimiser cannot see through: nothing
        // derived from them is hoisted out of the loop and carried (= spilled) across a whole TTI.
        typedef const __attribute__((address_space(4))) KP *kp_const_t;
        const int n = MANY ? (p.n_tti < 1 ? 1 : p.n_tti) : 1;      // (a launch steps at least once whatever the host left in the field)
        StepCarry cy = {};
        bool warm = false;
        for (int k = 0; k < n; k++) {
            kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kc));
            if (step_body<MODE_X, NQ, GATHER, NP, false, PACK, MIX>(*kc, cy, warm, MIX ? e_mix : kc->e0 + (int)blockIdx.x * PACK, nullptr, false, false, narrow)) return;
            if (k + 1 < n) {
                // The next TTI takes over in registers what it would otherwise load back (StepCarry) -- unless it has to look
                // for an allocation made ahead (RANENV_LATE) -- and then only LDS has to be handed over between the waves.
                warm = MANY && RANENV_WARM_ENTRY && kc->late == 0;
                if (warm) wg_sync(narrow); else full_sync(narrow);
            }
        }
    } else {
        StepCarry cy = {};
        step_body<MODE_X, NQ, GATHER, NP>(p, cy, false, p.e0 + (int)blockIdx.x);
    }
}
#undef COLD

// =============================================================================================
// Persistent rollout (option "persist"): ONE launch per workgroup class takes every env of the batch through all the TTIs of
// a ranenv_rollout call (up to the next episode end).  Why: a launch of one workgroup per env wants more slots than the chip
// has (4096 envs, ~3700 slots at the headline size), the workgroups that waited run last and alone, and every launch
// boundary pays that drain again (profiles/r03_ab_log.txt: a batch that is resident at once steps 10 % faster).  Here the
// grid is what fits, and a workgroup that finishes a chunk of TTIs of its env looks whether anybody is waiting:
//   * envs nobody has started yet (`fresh`: cursors over the class's env list, one shard per XCD label, taken first), or
//   * envs that another workgroup of THIS XCD has put down between two chunks (the XCD's ready queue);
//   if so it puts its env down (pushes it on its XCD's ready queue) and takes the waiting one, else it carries on with its
//   own env -- warm, registers and all -- so that a batch that is resident at once never touches the queues.
// Classes: a compact step needs one wave per 64 slice members of the env's scenario (lanes are ordered members first), and a
// wave that idles through a persistent launch would hold a wave slot for nothing; so the envs are sorted by the waves they
// need (ranenv_persist_classify_kernel) and each class gets a launch of its own with blocks of that many waves.
// Hand-over between workgroups: per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another
// CU's stores (MI355X_MICROARCH.md, Workgroup dispatch).  An env is therefore bound to the XCD that first touched it
// in this launch (fresh envs were last written by an earlier kernel: visible everywhere): its later chunks go through
// that XCD's own queue, producer and consumer share the L2, the producer's stores are acknowledged by that L2 before the push
// (s_waitcnt vmcnt(0) in every wave, workgroup barrier), and the consumer invalidates its CU's L1 (agent-scope acquire)
// behind the pop, before any wave of it loads.  The XCD is read from HW_REG_XCC_ID, not inferred from blockIdx.
// Nobody waits for work: a workgroup that finds no fresh env and its XCD's queue empty leaves (persist_pull says why that is
// safe).  The one spin -- on a queue entry whose pusher holds the ticket but has not written it yet -- is bounded (sticky error
// word, every workgroup leaves).
// =============================================================================================
struct PersistLocal { int item, next, keep, fresh_mask, xcc, pad; };
enum { PERSIST_EXIT = -1, PERSIST_NONE = -2 };
enum { PERSIST_ENV_BITS = 20 };           // item = env | TTIs done << 20

// (statistics of the queues, one fire-and-forget add per event from lane 0: ranenv_get_option "persist_stat_*")
#define PSTAT(k) ((void)__hip_atomic_fetch_add(&p.p_ctl->stat[pl.xcc][k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
DEVFN unsigned pq_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN int pq_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lane 0 only.  -> item, or PERSIST_NONE
template <typename P> DEVFN int persist_try_fresh(const P &p, PersistLocal &pl)
{
    PersistCtl *c = p.p_ctl;
    for (int s8 = 0; s8 < 8 && pl.fresh_mask != 0; s8++) {
        const int x = (pl.xcc + s8) & 7;
        if (!(pl.fresh_mask >> x & 1)) continue;
        const unsigned j = __hip_atomic_fetch_add(&c->fresh[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long i = (long long)x + 8ll * (long long)j;
        if (i < (long long)p.p_count) { PSTAT(3); return p.p_list[i]; }  // (TTIs done: 0)
        pl.fresh_mask &= ~(1 << x);
    }
    return PERSIST_NONE;
}

// lane 0 only.  -> item (L1 of this CU invalidated behind the pop), PERSIST_NONE if nothing is committed to the queue.
// Three words per queue, no compare-and-swap: `avail` counts committed entries nobody has claimed (a semaphore: whoever takes it
// from > 0 owns exactly one entry, whoever finds it <= 0 gives it back and goes), `head` hands the claimed entries out in
// order, `tail` hands out the slots to write.  A claimed slot may still be in the hands of its pusher (ticket taken, store on
// its way): the claimer spins on that slot's tag, a bounded wait.  (The first version popped by compare-and-swap on `head`:
// three dependent loads and the swap per attempt, and with a few hundred workgroups of an XCD at the queue 35 of 36 attempts
// lost -- profiles/r04_ab_log.txt.)
template <typename P> DEVFN int persist_try_pop(const P &p, const PersistLocal &pl)
{
    PersistCtl *c = p.p_ctl;
    if (pq_ldi(&c->q[pl.xcc].avail) <= 0) return PERSIST_NONE;
    const int a = __hip_atomic_fetch_add(&c->q[pl.xcc].avail, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a <= 0) { __hip_atomic_fetch_add(&c->q[pl.xcc].avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return PERSIST_NONE; }
    const unsigned h = __hip_atomic_fetch_add(&c->q[pl.xcc].head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long *slot = p.p_slots + (size_t)pl.xcc * (size_t)p.p_cap + (h & (unsigned)(p.p_cap - 1));
    unsigned long long ent = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (unsigned spin = 0; (unsigned)(ent >> 32) != h + 1u; spin++) {
        PSTAT(4);
        if (spin > (1u << 22) || pq_ldi(&c->abort) != 0) {               // seconds on one slot: never in a correct run
            __hip_atomic_store(&c->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.p_err) __hip_atomic_store(p.p_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (host memory: ranenv::h_perr)
            return PERSIST_EXIT;
        }
        __builtin_amdgcn_s_sleep(2);
        ent = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    PSTAT(2);
#ifndef RANENV_PERSIST_NO_ACQUIRE      /* timing experiments only: results may be stale without it */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // buffer_inv sc1: this CU's L1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    return (int)(unsigned)ent;
}

template <typename P> DEVFN void persist_push(const P &p, const PersistLocal &pl, int item)
{
    PersistCtl *c = p.p_ctl;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (every wave waited for its stores in front of the chunk-end barrier: full_sync)
    const unsigned idx = __hip_atomic_fetch_add(&c->q[pl.xcc].tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long *slot = p.p_slots + (size_t)pl.xcc * (size_t)p.p_cap + (idx & (unsigned)(p.p_cap - 1));
    __hip_atomic_store(slot, ((unsigned long long)(idx + 1u) << 32) | (unsigned)item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->q[pl.xcc].avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PSTAT(1);
}

// lane 0 only: the next env of this workgroup -> item or PERSIST_EXIT.
// A workgroup that finds no fresh env and nothing committed to its XCD's queue LEAVES; it does not wait.  Nothing it would
// have to serve can be lost: an env is put down only by a workgroup that stays alive and comes back to the queue (at its next
// chunk end, or when its own env is through: it leaves only past an empty queue), so an env on a queue always has a live
// workgroup of its XCD.  (The first version polled here until the class had finished: a few hundred sleeping workgroups
// polling three words of HBM-side state every microsecond cost the running ones a factor of four, profiles/r04_ab_log.txt.)
// The freed slots go to the grid's workgroups that did not fit at first.
template <typename P> DEVFN int persist_pull(const P &p, PersistLocal &pl)
{
    if (pl.next != PERSIST_NONE) { const int it = pl.next; pl.next = PERSIST_NONE; return it; }
    int it = persist_try_fresh(p, pl);
    if (it != PERSIST_NONE) return it;
    it = persist_try_pop(p, pl);
    return it != PERSIST_NONE ? it : PERSIST_EXIT;
}

// lane 0 only, behind the barrier that follows a chunk: -> 1 the workgroup keeps its env, 0 it has let go of it
template <typename P> DEVFN int persist_finish(const P &p, PersistLocal &pl, int e, int done, int n_tti)
{
    PersistCtl *c = p.p_ctl;
    if (done >= n_tti) return 0;
    if (pq_ldi(&c->abort) != 0) {        // a wait gave up somewhere in this class: the env is dropped here, and the host is told (again)
        if (p.p_err) __hip_atomic_store(p.p_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return 0;
    }
    const int fresh = persist_try_fresh(p, pl);
    if (fresh == PERSIST_NONE) {
        if (pq_ldi(&c->q[pl.xcc].avail) <= 0) { PSTAT(0); return 1; }    // nobody is waiting
        const unsigned h = pq_ld(&c->q[pl.xcc].head);
        // Somebody is -- but a swap only helps when the env at the head of the queue is BEHIND this one: with every
        // finisher swapping, every chunk of every env would go through the queue; this way a round of chunks costs one swap per
        // waiting env (a racy look at the head entry: a heuristic, whichever way it goes the state stays consistent).
        const unsigned long long ent = __hip_atomic_load(p.p_slots + (size_t)pl.xcc * (size_t)p.p_cap + (h & (unsigned)(p.p_cap - 1)),
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(ent >> 32) == h + 1u && (int)((unsigned)ent >> PERSIST_ENV_BITS) >= done) { PSTAT(0); return 1; }
    }
    pl.next = fresh;
    persist_push(p, pl, e | (done << PERSIST_ENV_BITS));
    return 0;
}

template <int NQ, bool GATHER, int NP>
DEVFN void persist_loop()
{
    typedef const __attribute__((address_space(4))) KP *kp_const_t;
    __shared__ PersistLocal pl;
    const int tid0 = threadIdx.x;
    if (tid0 == 0) {
        pl.next = PERSIST_NONE; pl.fresh_mask = 0xff; pl.keep = 0;
        pl.xcc = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u);      // HW_REG_XCC_ID, bits 3:0
    }
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    for (;;) {
        {
            kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kc));
            if (tid0 == 0) pl.item = persist_pull(*kc, pl);
        }
        __syncthreads();
        const int item = uni(pl.item);
        if (item < 0) {
            // the last workgroup out resets what the next launch of this class starts from (nobody is left to read the cursors;
            // the queue tickets are monotonic and stay)
            if (tid0 == 0) {
                kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kc));
                PersistCtl *c = kc->p_ctl;
                const unsigned n = __hip_atomic_fetch_add(&c->exited, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (n + 1u == gridDim.x) {
                    for (int x = 0; x < 8; x++) __hip_atomic_store(&c->fresh[x][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&c->exited, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            return;
        }
        const int e = item & ((1 << PERSIST_ENV_BITS) - 1);
        int done = (int)((unsigned)item >> PERSIST_ENV_BITS);
        StepCarry cy = {};
        SeStream<GATHER ? 1 : NQ> seq;               // (the whole-row build requests a TTI's tile one TTI ahead: step_body, SE_AHEAD)
        bool warm = false, se_ready = false;
        for (;;) {                                   // chunks of this env for as long as nobody is waiting
            kp_const_t kc0 = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(kc0));
            const int n_tti = kc0->n_tti, left = n_tti - done;
            // (an env's first chunk is 1..chunk TTIs long by a hash of its index: the workgroups of a launch start together, and
            // chunks of one length would bring all of them to the queues at the same moments)
            int want = kc0->p_chunk;
            if (done == 0 && want > 1 && want < n_tti) want = 1 + (int)((((unsigned)e * 0x9E3779B1u) >> 16) % (unsigned)want);
            const int n = left < want ? left : want;
            for (int k = 0; k < n; k++) {
                kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kc));
                // another TTI of this env follows in this launch and enters warm if this workgroup makes it: the next one of the chunk, or
                // -- the workgroup keeps its env at most chunk ends -- the next chunk's first (what is requested ahead for it is wasted
                // when the env changes hands)
                const bool ahead = RANENV_WARM_ENTRY != 0 && done + k + 1 < n_tti;
                (void)step_body<MODE_STEP, NQ, GATHER, NP, true>(*kc, cy, warm, e, &seq, se_ready, ahead);
                se_ready = ahead;
                if (k + 1 < n) { warm = RANENV_WARM_ENTRY != 0; if (warm) wg_sync(); else full_sync(); }
            }
            done += n;
            // Hand-over point: EVERY wave waits for its own stores to be acknowledged by the XCD's L2 (explicit vmcnt(0): the
            // barrier's fence does not), then all meet; only then may lane 0 publish the env to another workgroup.
            full_sync();
            if (tid0 == 0) {
                kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(kc));
                pl.keep = persist_finish(*kc, pl, e, done, n_tti);
            }
            __syncthreads();
            if (uni(pl.keep) == 0) break;
            warm = RANENV_WARM_ENTRY != 0;           // `cy` is what the chunk's last TTI left
        }
    }
}

#ifndef RANENV_PERSIST_WAVES_PER_EU
#define RANENV_PERSIST_WAVES_PER_EU 5
#endif
#ifndef RANENV_PERSIST_GATHER_WAVES
#define RANENV_PERSIST_GATHER_WAVES 5
#endif
#define RANENV_PERSIST_WPE ((NP == 16) ? 4 : (GATHER ? RANENV_PERSIST_GATHER_WAVES : RANENV_PERSIST_WAVES_PER_EU))
template <bool GATHER, int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(RANENV_PERSIST_WPE, RANENV_PERSIST_WPE))) ranenv_persist_kernel(const KP p)
{
    (void)p;                                         // (read in place, like step_loop)
#if RANENV_DIAG == 0 || RANENV_DIAG == 12            /* (the other diagnostic / ablation builds run the launch-per-chunk rollout only) */
#if RANENV_NARROW_PRIO
    // the one-wave class finishes a rollout after the two-wave class (its envs have half the loads in flight): its waves issue first
    if (blockDim.x == WAVE) __builtin_amdgcn_s_setprio(RANENV_NARROW_PRIO);
#endif
    persist_loop<GATHER ? 1 : RANENV_SE_DEPTH, GATHER, NP>();
#endif
}

// The same for a batch that leaves the chip at <= 2 waves per SIMD (BASELINE configs[1], B 1024): 256 VGPRs are there for the
// taking, so a lane keeps its whole SE row in flight (16 groups of 8 loads): the stream phase of a workgroup's chain is one
// memory latency instead of four (profiles/r04_ab_log.txt: 23.4 against 26.6 us per TTI).  Streaming only.
#ifndef RANENV_SE_DEPTH_TINY
#define RANENV_SE_DEPTH_TINY 17       /* R = 135: 16 groups of 8 + the tail group -- the whole row, no load is requested inside the stream phase */
#endif
template <int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) ranenv_persist_kernel_tiny(const KP p)
{
    (void)p;
#if RANENV_DIAG == 0 || RANENV_DIAG == 12
    persist_loop<RANENV_SE_DEPTH_TINY, false, NP>();
#endif
}

// ... and as an ordinary one-TTI launch for the same batches (env.step() of a small batch: what an SB3 / RLlib trainer with a few hundred envs
// calls): the whole row requested at entry together with all of the UE's state -- the step is one chain of latencies, and this removes
// three of the stream's four.
template <int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) ranenv_core_kernel_tiny1(const KP p)
{
    step_loop<MODE_STEP, RANENV_SE_DEPTH_TINY, false, NP, false>(p);
}

// Sort the envs by the waves a compact step of theirs needs: class c = ceil(slice members of the env's scenario / 64) - 1.
// ONE workgroup (the counts are built in LDS: no memset in front, one launch in all).  `flag`: a device word that
// ranenv_advance_kernel sets when an env has restarted -- without `force` the kernel does nothing unless the word is set, and it
// clears it: an auto-reset loop in which no episode ended pays one empty launch, not a re-sort (and no host read-back of `done`).
__global__ void __launch_bounds__(1024) ranenv_persist_classify_kernel(const ranenv_episode *eps, const int32_t *members, int B, int n_class,
                                                                       int one_class, int32_t *list, int32_t *count, int *flag, int force)
{
    __shared__ int cnt[CORE_NT / WAVE];
    if (!force && *flag == 0) return;                 // (uniform: every thread reads the word before thread 0 clears it, behind the barriers)
    if (threadIdx.x < CORE_NT / WAVE) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int e = (int)threadIdx.x; e < B; e += (int)blockDim.x) {
        const int m = members[eps[e].scenario];
        int c = m <= 0 ? 0 : (m + WAVE - 1) / WAVE - 1;
        c = c < n_class ? c : n_class - 1;
        if (one_class) c = n_class - 1;           // a batch far below what the chip holds: idle waves cost nothing, a second launch does
        const int pos = atomicAdd(&cnt[c], 1);
        list[(size_t)c * B + pos] = e;
    }
    __syncthreads();
    if ((int)threadIdx.x < n_class) count[threadIdx.x] = cnt[threadIdx.x];
    if (threadIdx.x == 0) *flag = 0;
}
// Two builds of the step kernel.  A batch that fills the machine (more workgroups than 8 per CU) runs the lean one:
// 96 VGPRs = 5 waves per SIMD = 10 workgroups per CU, 16 SE loads in flight per lane (8 until the build stopped hoisting
// at machine level, which freed the registers for the second group) -- occupancy hides more latency than a still deeper
// queue (measured, profiles/r02_ab_log.txt, r03_ab_log.txt).  A small batch is resident at once whatever the register
// count, so it takes the build with 128 VGPRs and 32 loads in flight.
#ifndef RANENV_WAVES_PER_EU
#define RANENV_WAVES_PER_EU 5      /* experiment knob: waves per SIMD of the lean build (0 = compiler's choice) */
#endif
#if RANENV_WAVES_PER_EU > 0
#define RANENV_CORE_ATTR __attribute__((amdgpu_waves_per_eu(RANENV_WAVES_PER_EU, RANENV_WAVES_PER_EU)))
#else
#define RANENV_CORE_ATTR
#endif
// (MANY: a launch of several TTIs, ranenv_rollout only, runs a build of its own -- the one-TTI build stays free of the
// warm entry's second path through the role, which costs it 1-2 %)
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) RANENV_CORE_ATTR ranenv_core_kernel(const KP p)
{
    step_loop<MODE, ((MODE & 3) == MODE_DENSE || NP == 16) ? 1 : RANENV_SE_DEPTH, false, NP, MANY>(p);
}
#ifndef RANENV_SMALL_WAVES_PER_EU
#define RANENV_SMALL_WAVES_PER_EU 4
#endif
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(RANENV_SMALL_WAVES_PER_EU, RANENV_SMALL_WAVES_PER_EU))) ranenv_core_kernel_small(const KP p)
{
    step_loop<MODE, RANENV_SE_DEPTH_SMALL, false, NP, MANY>(p);
}
// The SE gather build (ranenv_set_se_mode): no tile stream, so no queue registers; one build for every batch size.
#ifndef RANENV_GATHER_WAVES_PER_EU
#define RANENV_GATHER_WAVES_PER_EU 5
#endif
// (the 16-wide row build keeps 16-entry rows of doubles alive in the allocation and does not fit 96 registers with the per-TTI loop
// around it -- 2...10 spilled VGPRs -- so it is built for 4 waves per SIMD: S or Us above 10 is not a BASELINE size, and a spill in
// every TTI costs more than the fifth wave gains, profiles/r03_ab_log.txt.  No kernel of the library has scratch:
// tests/test_kernel_resources.py reads the shipped code object's metadata)
#define RANENV_WPE_NP(w) ((NP == 16 && MANY) ? 4 : (w))
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(RANENV_WPE_NP(RANENV_GATHER_WAVES_PER_EU), RANENV_WPE_NP(RANENV_GATHER_WAVES_PER_EU))))
ranenv_core_kernel_gather(const KP p) { step_loop<MODE, 1, true, NP, MANY>(p); }
// Every build above exists for three row widths NP (see np_sum_lds): 8, 10 (BASELINE's 10 slices / 10 UEs per slice), 16.

// Mixed blocks (round 4): a step launch of one two-wave workgroup per env holds ~2 560 envs of 100 UEs at a time (and its compact form
// ~3 700: the idle second wave of an env of <= 64 slice members still needs a slot to start), so a TTI of 4096 envs is two rounds.
// Here the launch is one block per env of the WIDE class (> 64 members: both waves) and one block per TWO envs of the NARROW class
// (one wave each, no block barrier between them): 1 023 + 1 537 blocks = every wave slot of the chip, the whole batch resident in one
// round.  Compact lane order (the classes are defined by it); the env lists are the persistent rollout's.
template <int NP, bool MANY, bool GATHER>
__global__ void __launch_bounds__(2 * WAVE) RANENV_CORE_ATTR ranenv_core_kernel_mixed(const KP p)
{
    step_loop<MODE_STEP, GATHER ? 1 : RANENV_SE_DEPTH, GATHER, NP, MANY, 1, true>(p);
}

// Packed waves (round 4): envs of at most 32 UEs and 8 slices -- the reference's own size, S 5 / U 25 -- leave 39 of a wave's 64 lanes
// idle, and the chip holds as many waves as it holds; one wave steps TWO envs (lanes 0-31 / 32-63, step_body's PACK = 2): half as many
// waves per env-step.  What is wave-uniform in the other builds is per-lane here (more registers: 4 waves per SIMD), so it is a build
// of its own, for step launches of an even number of envs; reset and dense launches keep one env per wave (same state layout).
#ifndef RANENV_PACK_NQ
#define RANENV_PACK_NQ 2           /* 8-RB groups in flight per lane in the packed streaming builds */
#endif
#ifndef RANENV_PACK_WPE
#define RANENV_PACK_WPE 4
#endif
template <int NP, bool MANY, bool GATHER>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(RANENV_PACK_WPE, RANENV_PACK_WPE))) ranenv_core_kernel_packed(const KP p)
{
    step_loop<MODE_STEP, GATHER ? (MANY ? 1 : 0) : RANENV_PACK_NQ, GATHER, NP, MANY, 2>(p);        // (one-TTI gather build: gather depth 1, it has no register to spare)
}

// ---------------------------------------------------------------------------------------------
// Sidecars of the SE pool for the gather mode, built once per bound pool (ranenv_set_se_mode):
//   mean[tile][u]    = np.mean(SE[u, :]) in float64, numpy's pairwise order (row_sums' `full`, divided by R): bit for bit what
//                      the streaming kernel derives from the tile every TTI
//   um[tile][u][Rp]  = the tile UE-major, rows padded with zeros to Rp = R rounded up to 8 floats
// One workgroup per tile, thread = UE for the means; the copy is a plain index transform (reads served by L2).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(CORE_NT) ranenv_se_sidecar_kernel(const float *pool, long long stride, long long tile0, int U, int R, int Rp,
                                                                    int quad, double *mean, float *um)
{
    const long long t = tile0 + blockIdx.x;
    const float *tile = pool + (size_t)t * (size_t)stride;
    const int tid = threadIdx.x;
    const int u = tid < U ? tid : U - 1;
    SeStream<4> se;
    se.init(tile, U, u, R, quad != 0);
    double full = 0.0, part = 0.0;
    row_sums(se, R, [](int) { return false; }, full, part, []() {});
    if (tid < U) mean[(size_t)t * U + tid] = full / (double)R;
    float *out = um + (size_t)t * (size_t)U * Rp;
    for (int i = tid; i < U * Rp; i += (int)blockDim.x) {
        const int uu = i / Rp, r = i - uu * Rp;
        out[i] = r < R ? (quad ? tile[((size_t)(r >> 2) * U + uu) * 4 + (r & 3)] : tile[(size_t)r * U + uu]) : 0.0f;
    }
}

// RB-major [n][R][U] -> RB-quad-major [n][ceil(R/4)][U][4] (ranenv_se_retile_quad): one float4 of the output per thread, zeros behind RB R-1
__global__ void __launch_bounds__(256) ranenv_se_retile_quad_kernel(const float *src, float *dst, long long n_quads, int U, int R)
{
    const int Rq = (R + 3) >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / ((long long)Rq * U);
        const int rem = (int)(i - t * (long long)Rq * U), qr = rem / U, u = rem - qr * U;
        const float *tile = src + (size_t)t * (size_t)U * R;
        se_v4f v;
        v.x = tile[(size_t)(4 * qr) * U + u];
        v.y = 4 * qr + 1 < R ? tile[(size_t)(4 * qr + 1) * U + u] : 0.0f;
        v.z = 4 * qr + 2 < R ? tile[(size_t)(4 * qr + 2) * U + u] : 0.0f;
        v.w = 4 * qr + 3 < R ? tile[(size_t)(4 * qr + 3) * U + u] : 0.0f;
        ((se_v4f *)dst)[i] = v;
    }
}

// Gather-only ingest (ranenv_bind_se_gather_from_power): the same two sidecars straight from QuaDRiGa received power
// (channels/quadriga.py:56-69), without an RB-major float32 pool ever existing.  The float32 SE of an element is what
// ranenv_se_from_power would have stored; the mean runs through row_sums over those float32 values, so both sidecars are bit for
// bit what ranenv_set_se_mode builds from the pool ranenv_se_from_power writes.
struct PowerStream {           // row_sums' source interface over a tile of float64 power, converted on the way in
    static constexpr int NSLOT = 2;
    const double *tile; int U, u, R; double tx, noise;
    float q[NSLOT][8];
    DEVFN float se_of(int r) const
    {
        const int rr = r < R ? r : R - 1;                              // (padding rows of the last group: never summed)
        return (float)log2(1.0 + (tx * tile[(size_t)rr * U + u]) / (0.0 + noise));
    }
    DEVFN void refill(int d, int r0) {
#pragma unroll
        for (int j = 0; j < 8; j++) q[d][j] = se_of(r0 + j);
    }
    DEVFN void init() { for (int d = 0; d < NSLOT; d++) if (d * 8 < R) refill(d, d * 8); }
    DEVFN void take(int d, float (&x)[8], int) {
#pragma unroll
        for (int j = 0; j < 8; j++) x[j] = q[d][j];
    }
};

__global__ void __launch_bounds__(CORE_NT) ranenv_se_sidecar_from_power_kernel(const double *power, long long tile0, int U, int R, int Rp,
                                                                               double tx, double noise, double *mean, float *um)
{
    const long long t = tile0 + blockIdx.x;
    const double *tile = power + (size_t)t * (size_t)U * (size_t)R;
    const int tid = threadIdx.x;
    PowerStream ps;
    ps.tile = tile; ps.U = U; ps.u = tid < U ? tid : U - 1; ps.R = R; ps.tx = tx; ps.noise = noise;
    ps.init();
    double full = 0.0, part = 0.0;
    row_sums(ps, R, [](int) { return false; }, full, part, []() {});
    if (tid < U) mean[(size_t)t * U + tid] = full / (double)R;
    float *out = um + (size_t)t * (size_t)U * Rp;
    for (int i = tid; i < U * Rp; i += (int)blockDim.x) {
        const int uu = i / Rp, r = i - uu * Rp;
        out[i] = r < R ? (float)log2(1.0 + (tx * tile[(size_t)r * U + uu]) / (0.0 + noise)) : 0.0f;
    }
}

// =============================================================================================
// Alternative heads: the observation of SchedTWC / SchedColORAN (agents/sched_twc.py:165-346:
// 3 requirements + 7 slice means per slice, slices in index order, metric-major) and their rewards
// (sched_twc.py:348-413, sched_colran.py:348-419), from the state the core kernel just wrote.
// One workgroup = one env, thread = slot (slice, UE position), launched after the core kernel when
// head outputs are bound.
//
// These agents push every raw observation twice into their 10-deep deque (sched_twc.py:174-177), so
// their window is the last D/2 TTIs counted twice, and "the previous entry" is the current TTI again:
// entry i of their deque is TTI i/2 of the window ring.
// =============================================================================================
struct SharedHead {
    double rows[GRP][10][GRP];    // per slice: mean SE, served Mbps, effective Mbps, occupancy, latency, loss,
                                  //            raw capacity, drift x3 -- by UE position, zero padded
    double sv[GRP][3];            // slice drift means (-2: not declared)
    double thr_raw[GRP], occ_m[GRP];
    int nues[GRP];
    double terms[3 * GRP], nw[3 * GRP];   // the reward's terms and weights (one lane fills them: LDS, not 784 B of scratch per lane)
};

// numpy pairwise_sum of n < 128 doubles by one lane
DEVFN double np_sum_seq(const double *a, int n)
{
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

__global__ void __launch_bounds__(CORE_NT) ranenv_head_kernel(const KP p)
{
    __shared__ SharedHead sh;
    auto wave_sync = []() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    const int e = p.e0 + blockIdx.x, tid = threadIdx.x;
    if (p.env_mask != nullptr && p.env_mask[e] == 0) return;
    const int S = p.S, U = p.U, D = p.D;
    const int sc = __builtin_amdgcn_readfirstlane(p.episodes[e].scenario);
    const int hlen = __builtin_amdgcn_readfirstlane(ST_hist_len(p)[e]);      // counters after this TTI's push
    const int npush = __builtin_amdgcn_readfirstlane(ST_n_push(p)[e]);
    const int s = tid / GRP, pos = tid % GRP;
    const bool in_grid = s < S;
    const int NS16 = S * GRP;
    int ue = -1, mp = 1;
    if (tid < NS16) { const size_t ts = (size_t)sc * NS16 + tid; ue = TB_slot_ue(p)[ts]; mp = TB_slot_mp(p)[ts]; }
    const bool have = ue >= 0;
    const int gsh = (tid & 63) & ~(GRP - 1);
    const int n = __popc((unsigned)((__ballot(have) >> gsh) & 0xffffull));
    int active = 0, has_req = 0, bsize = 1, blat = 1, msg = 1, npar = 0;
    int pm[3] = {0, 0, 0}, po[3] = {0, 0, 0};
    double pv[3] = {0.0, 0.0, 0.0}, traffic_tab = 0.0;
    if (in_grid) {
        const size_t row = (size_t)sc * S + s;
        const int32_t *si = TB_slice_i32(p) + row * 8;
        active = si[0]; has_req = si[1]; bsize = si[3]; blat = si[4]; msg = si[5]; npar = si[6];
        traffic_tab = TB_slice_f64(p)[row * 2 + 1];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            pm[k] = TB_param_i32(p)[(row * 3 + k) * 2 + 0];
            po[k] = TB_param_i32(p)[(row * 3 + k) * 2 + 1];
            pv[k] = TB_param_f64(p)[row * 3 + k];
        }
    }
    // ---- the UE of this slot ----------------------------------------------------------------------
    double vals[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (have) {
        const size_t su = (size_t)e * U + ue;
        const int total = ST_queue_pkts(p)[su];
        const long long sum_age = ST_queue_age_sum(p)[su];
        const double sem = ST_se_mean(p)[su];
        const double sent = (double)ST_pkt_effective_thr(p)[su], thr = (double)ST_pkt_throughputs(p)[su];
        // the heads' window: deque entry i is TTI i/2; view_len = min(2*hlen, D)
        const int vlen = 2 * hlen < D ? 2 * hlen : D;
        double sw = 0.0, dw = 0.0;
        for (int j = 0; 2 * j < vlen; j++) {
            int idx = npush - 1 - j; idx += idx < 0 ? D : 0;
            const double mult = 2 * j + 1 < vlen ? 2.0 : 1.0;
            sw += mult * (double)ST_ring_sent(p)[((size_t)e * D + idx) * U + ue];
            dw += mult * (double)ST_ring_drop(p)[((size_t)e * D + idx) * U + ue];
        }
        const double occ = (double)total / (double)mp;
        const double lat = total > 0 ? (double)sum_age / (double)total : 0.0;
        const double bp = occ * (double)bsize + dw + sw;                     // common.py:32-53
        const double loss = bp != 0.0 ? dw / bp : 0.0;
        vals[0] = sem;
        vals[1] = thr * (double)msg / 1e6;                                   // sched_twc.py:255-266
        vals[2] = sent * (double)msg / 1e6;                                  // :269-280
        vals[3] = occ; vals[4] = lat; vals[5] = loss; vals[6] = thr;
        if (has_req) {                                                       // common.py:68-340, heads' deque
            const double o = p.over;
#pragma unroll
            for (int qi = 0; qi < 3; qi++) {
                if (qi < npar) {
                    const int metric = pm[qi], op = po[qi];
                    const double value = pv[qi];
                    double res;
                    if (metric == RANENV_METRIC_THROUGHPUT) {
                        double x = (sent * (double)msg) / 1e6;
                        if (d_isclose(occ, 0.0)) x = value * (1.1 + o);      // entry 1 of their deque = this TTI
                        if (d_apply_op(op, x, value)) res = (x > value * (1.0 + o)) ? 1.0 : (x - value) / (value * o);
                        else res = -((value - x) / value);
                    } else if (metric == RANENV_METRIC_RELIABILITY) {
                        const double x = loss;
                        const double band = (100.0 - value) / 100.0;
                        if (d_apply_op(op, 100.0 * (1.0 - x), value)) res = (x < band * (1.0 - o)) ? 1.0 : (band - x) / (band * o);
                        else res = -((x - band) / (value / 100.0));
                    } else {
                        const double x = lat;
                        if (d_apply_op(op, x, value)) res = (x < value * (1.0 - o)) ? 1.0 : (value - x) / (value * o);
                        else res = -((x - value) / ((double)blat - value));
                    }
                    vals[7] = metric == 0 ? res : vals[7]; vals[8] = metric == 1 ? res : vals[8]; vals[9] = metric == 2 ? res : vals[9];
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; k++) sh.rows[s][k][pos] = vals[k];
    wave_sync();
    // ---- the slice (lane 0 of its 16 writes) -------------------------------------------------------
    if (in_grid && pos == 0) {
        float *o = p.head_obs ? p.head_obs + (size_t)e * 10 * S : nullptr;
        double m[7] = {0, 0, 0, 0, 0, 0, 0};
        if (n > 0) {
#pragma unroll
            for (int k = 0; k < 7; k++) m[k] = np_sum16_lds(sh.rows[s][k], n) / (double)n;
        }
        double sv[3] = {-2.0, -2.0, -2.0};
        double req[3] = {0.0, 0.0, 0.0};
        if (n > 0 && has_req) {
#pragma unroll
            for (int qi = 0; qi < 3; qi++) {
                if (qi < npar) {
                    const int mt = pm[qi];
                    const double mean = np_sum16_lds(sh.rows[s][7 + mt], n) / (double)n;
                    sv[0] = mt == 0 ? mean : sv[0]; sv[1] = mt == 1 ? mean : sv[1]; sv[2] = mt == 2 ? mean : sv[2];
                    // requirements = [reliability, latency, throughput]            sched_twc.py:216-226
                    req[0] = mt == RANENV_METRIC_RELIABILITY ? pv[qi] : req[0];
                    req[1] = mt == RANENV_METRIC_LATENCY ? pv[qi] : req[1];
                    req[2] = mt == RANENV_METRIC_THROUGHPUT ? pv[qi] : req[2];
                }
            }
        }
        if (o) {
            o[3 * s + 0] = (float)req[0]; o[3 * s + 1] = (float)req[1]; o[3 * s + 2] = (float)req[2];
#pragma unroll
            for (int k = 0; k < 6; k++) o[(3 + k) * S + s] = (float)m[k];
            o[9 * S + s] = (float)(d_isclose((double)active, 1.0) ? traffic_tab : 0.0);  // :325-337
        }
        sh.sv[s][0] = sv[0]; sh.sv[s][1] = sv[1]; sh.sv[s][2] = sv[2];
        sh.thr_raw[s] = m[6]; sh.occ_m[s] = m[3];
        sh.nues[s] = n;
    }
    __syncthreads();
    // ---- the rewards (one lane; a few dozen values) ------------------------------------------------
    if (tid == 0 && p.head_reward) {
        double *terms = sh.terms, *nw = sh.nw;
        int q = 0;
        double r_col = 0.0;
        for (int sl = 0; sl < S; sl++) {
            const int nu = sh.nues[sl];
            if (nu == 0) continue;                                               // sched_twc.py:364-365
            const size_t row = (size_t)sc * S + sl;
            const double w = TB_slice_f64(p)[row * 2 + 0] != 0.0 ? 2.0 : 1.0;    // :382-391
            for (int k = 0; k < 3; k++) {
                const double v = sh.sv[sl][k];
                if (d_isclose(v, -2.0) || !(v < 0.0)) continue;                  // :376-378, :395-399
                terms[q] = v; nw[q] = w; q++;
            }
            const int32_t *si = TB_slice_i32(p) + row * 8;
            if (si[0] != 0) {                                                    // sched_colran.py:372-419
                const int uc = TB_slice_usecase(p)[row];
                const double pkt = (double)si[5];
                if (uc & 1) r_col += ((sh.thr_raw[sl] * pkt) / 1e6) / 200.0;
                if (uc & 2) r_col -= ((sh.occ_m[sl] * (double)si[3]) * pkt / 1e6) / 2000.0;
            }
        }
        const double wsum = np_sum_seq(nw, q);
        double r_twc = 0.0;
        if (!d_isclose(wsum, 0.0)) {
            for (int i = 0; i < q; i++) terms[i] = terms[i] * nw[i] / wsum;
            r_twc = np_sum_seq(terms, q);
        }
        p.head_reward[(size_t)e * 2 + 0] = r_twc;
        p.head_reward[(size_t)e * 2 + 1] = r_col;
    }
}

// ---------------------------------------------------------------------------------------------
// Channel ingest: received power -> spectral efficiency (channels/quadriga.py:56-69), elementwise.
// 8 B read + 4 B written per element; two elements per thread and grid-stride, 16-byte loads.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ranenv_se_from_power_kernel(const double *power, float *se, long long n,
                                                                   double tx_per_rb, double noise)
{
    const long long stride = (long long)gridDim.x * blockDim.x * 2;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 2; i < n; i += stride) {
        if (i + 1 < n && ((size_t)(power + i) & 15) == 0 && ((size_t)(se + i) & 7) == 0) {
            const double2 g = *reinterpret_cast<const double2 *>(power + i);
            float2 o;
            o.x = (float)log2(1.0 + (tx_per_rb * g.x) / (0.0 + noise));
            o.y = (float)log2(1.0 + (tx_per_rb * g.y) / (0.0 + noise));
            *reinterpret_cast<float2 *>(se + i) = o;
        } else {
            se[i] = (float)log2(1.0 + (tx_per_rb * power[i]) / (0.0 + noise));
            if (i + 1 < n) se[i + 1] = (float)log2(1.0 + (tx_per_rb * power[i + 1]) / (0.0 + noise));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Auto-reset, part 1 (part 2 is the step kernel in RESET mode under the mask written here): one workgroup
// per env.  For an env whose episode just ended (done != 0): keep its terminal observation, pick the next
// episode number -- sequential from `initial`, or random in [initial, max) (simu.py:361,377,546; the draw is
// counter-based: seed, env id, resets so far) -- and install that episode's descriptor from the table.
// ---------------------------------------------------------------------------------------------
struct AdvanceArgs {
    const uint8_t *done; uint8_t *mask;
    ranenv_episode *episodes; const ranenv_episode *table; int table_first, table_n;
    int32_t *episode_no, *reset_count;
    int initial, max_ep, random, env_id_base; unsigned long long seed;
    const float *obs_inter, *obs_intra, *head_obs; float *term_inter, *term_intra, *term_head;
    int n_inter, n_intra, n_head;
    int e0;                                       // first env of this launch (batch partitions)
    int *cls_flag;                                // set when an env restarts: the class lists of the mixed / persistent launches are stale
    const double *acc; double *ep_acc; int32_t *ep_n; int ep_slots;   // episode metrics: running sums -> per-episode log
};

__global__ void __launch_bounds__(64) ranenv_advance_kernel(const AdvanceArgs a)
{
    const int e = a.e0 + blockIdx.x, tid = threadIdx.x;
    const bool d = a.done[e] != 0;
    if (tid == 0) a.mask[e] = d ? 1 : 0;
    if (!d) return;
    if (a.acc) {          // the finished episode's sums go to the env's log (the reset that follows zeroes the running sums)
        const int n = a.ep_n[e];                  // read by every thread of this one wave before thread 8 stores
        if (tid < 8 && n < a.ep_slots) a.ep_acc[((size_t)e * a.ep_slots + n) * 8 + tid] = a.acc[(size_t)e * 8 + tid];
        if (tid == 8) a.ep_n[e] = n + 1;
    }
    if (a.term_inter) for (int i = tid; i < a.n_inter; i += 64) a.term_inter[(size_t)e * a.n_inter + i] = a.obs_inter[(size_t)e * a.n_inter + i];
    if (a.term_intra) for (int i = tid; i < a.n_intra; i += 64) a.term_intra[(size_t)e * a.n_intra + i] = a.obs_intra[(size_t)e * a.n_intra + i];
    if (a.term_head && a.head_obs) for (int i = tid; i < a.n_head; i += 64) a.term_head[(size_t)e * a.n_head + i] = a.head_obs[(size_t)e * a.n_head + i];
    if (tid == 0) {
        const int cur = a.episode_no[e], cnt = a.reset_count[e] + 1;
        int next;
        if (a.random) {
            unsigned rnd[4];
            philox4x32_10((unsigned)(a.env_id_base + e), (unsigned)cnt, 0x45504953u /* "EPIS" */, 0u,
                          (unsigned)a.seed, (unsigned)(a.seed >> 32), rnd);
            next = a.initial + (int)(rnd[0] % (unsigned)(a.max_ep - a.initial));
        } else {
            next = cur + 1 < a.max_ep ? cur + 1 : a.initial;
        }
        a.episode_no[e] = next; a.reset_count[e] = cnt;
        a.episodes[e] = a.table[next - a.table_first];
        if (a.cls_flag) *a.cls_flag = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// Do the traffic traces carry bits for UEs outside every slice?  (MultSliceTraffic.step never does: it draws for the UEs
// of slices with a request only, traffics/mult_slice.py:24-32.)  One workgroup per episode descriptor scans the rows of
// its traffic trace at the idle UEs of its scenario.  Only when none does may a step leave idle UEs alone (KP::compact).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ranenv_idle_traffic_kernel(const ranenv_episode *eps, const int32_t *pool, int U,
                                                                  const int32_t *lane_slice, const int32_t *lane_ue, int *violations)
{
    const ranenv_episode ep = eps[blockIdx.x];
    int bad = 0;
    for (int l = threadIdx.x; l < U; l += (int)blockDim.x) {
        const size_t tu = (size_t)ep.scenario * U + l;
        if (lane_slice[tu] >= 0) continue;
        const int ue = lane_ue[tu];
        for (int row = 0; row < ep.trf_len; row++) bad |= pool[((size_t)ep.trf_base + (size_t)row) * U + ue] != 0;
    }
    if (bad) atomicOr(violations, 1);
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
    int alloc_gen = 1;                          // bumped by everything a stored next-TTI allocation depends on
    ranenv_episode *d_ep_table = nullptr; int ep_table_first = 0, ep_table_n = 0;     // auto-reset: episode number -> descriptor
    int ar_initial = 0, ar_max = 0, ar_random = 0; unsigned long long ar_seed = 0; bool ar_on = false;
    uint8_t *d_ar_mask = nullptr;
    double *d_acc = nullptr, *d_ep_acc = nullptr; int32_t *d_ep_n = nullptr; int ep_slots = 0;   // ranenv_enable_metrics
    std::vector<int32_t> host_max_steps;        // copy of ranenv_set_max_steps' array (the multi-episode rollout follows the step counters)
    // Host shadow of the per-env step counters (what they will be once everything enqueued so far has run): `done` is a function of
    // the counter alone (step >= the env's episode length), so ranenv_autoreset knows WITHOUT reading anything back whether an episode
    // ended at the TTI just enqueued -- and enqueues nothing when none did (an RL loop calls it behind every step: three small launches
    // = 7 us per TTI saved, profiles/r05_ab_log.txt).  Valid from a reset of the whole batch until something the host cannot follow
    // (a masked reset by the caller).
    std::vector<int32_t> sh_steps; bool sh_valid = false;
    const uint8_t *last_done = nullptr;         // the `done` buffer the steps write (the shortcut applies to that buffer only)
    unsigned long long *d_pois_cdf = nullptr; uint8_t *d_pois_guide = nullptr; int32_t *d_max_steps = nullptr;
    std::vector<double> slice_traffic;          // [NS][S] host copy (traffic generator tables)
    std::vector<int32_t> slice_has_req;
    int64_t se_tiles_n = 0, trf_rows_n = 0;   // extents of the bound pools (0 = none)
    // SE gather mode (ranenv_set_se_mode): sidecars of the bound pool, owned by the handle
    int se_mode = RANENV_SE_STREAM;
    double *d_se_mean = nullptr; float *d_se_um = nullptr; int se_rp = 0;
    // compact steps (KP::compact): allowed while UEs outside every slice provably receive no traffic
    int persist = -1;              // ranenv_rollout as one persistent work-queue launch per workgroup class (option "persist"):
                                   // 0 never, 1 whenever possible, -1 (default) where it was measured to win: SE gather mode with a
                                   // batch that fills the CUs, and either mode with a batch of <= 2 waves per SIMD
    int persist_chunk = 10;        // TTIs of an env between two visits of the work queue
    int n_cus = 256;               // compute units of the device (ranenv_create)
    bool pack = true;              // two envs per wave where the sizes allow (option "pack")
    int mix = 1;                   // whole-batch step launches of two-wave workgroups as mixed blocks (option "mix"): 0 never, 1 where the
                                   // batch does not fit the chip anyway (auto), 2 also for batches that do (tests)
    int persist_grid = 0;          // experiment: cap on the workgroups of a persistent launch, in wave slots (0 = what the chip holds)
    std::vector<int32_t> members_host; int32_t *d_members = nullptr;      // [NS] UEs in slices per scenario
    int32_t *d_plist = nullptr, *d_pcount = nullptr; PersistCtl *d_pctl = nullptr; unsigned long long *d_pslots = nullptr;
    int p_nclass = 0, p_cap = 0;
    std::vector<int32_t> pcount_host; bool pclass_dirty = true, pcount_host_stale = true;
    bool pclass_maybe = false;     // an auto-reset ran since the lists were built: they are stale IF an env restarted (the device knows:
    int *d_cls_flag = nullptr;     // ... this word, set by ranenv_advance_kernel, tested and cleared by the classify kernel)
    int *h_perr = nullptr;         // sticky error word of the persistent launches, in host memory the device can write (a wait gave up)
    int *d_perr_dev = nullptr;     // ... its address as the device sees it
    int perr_seen = 0;             // ... what of it has been reported
    int persist_inject = 0;        // test hook (option "persist_inject_abort"): the next persistent launch finds its abort word set
    int last_rollout_persistent = 0, last_rollout_launches = 0;   // what the last ranenv_rollout call ran (read-only options)
    int p_wave_slots[2] = {0, 0};  // wave slots per CU of the persistent kernel (streaming, gather build), from the occupancy query
    long long prof_env_ttis = 0;   // env-TTIs covered by the launches timed since ranenv_profile_begin
    int fuse = 0;                  // TTIs per launch inside ranenv_rollout: 0 = chosen per rollout, n = at most n (1 = off)
    std::vector<int> fuse_first;   // override of the length of partition k's first launch of a rollout (RANENV_FUSE_FIRST=a,b,c)
    long long prof_ttis = 0;       // TTIs covered by the launches timed since ranenv_profile_begin
    bool compact_enabled = true, idle_check_dirty = true, pool_idle_zero = false, table_idle_zero = false;
    bool idle_state_clean = true;               // no step so far can have given an idle UE packets (else: full width until a full reset)
    int *d_violations = nullptr;
    int nt = 0;                                 // threads of the core kernel (one per UE, whole waves)
    int np = 16;                                // row width of the step kernel's build: max(S, Us) rounded up to 8, 10 or 16
    int nslot = 0;                              // threads of the head kernel (one per slot, whole waves)
    int tiny_step = 1;                          // option "tiny_step": one-TTI step launches of a batch at <= 2 waves per SIMD run the whole-row build
    bool small_batch = false;                   // at most 8 workgroups per CU: the 128-VGPR build with the deeper SE queue
    // ranenv_profile_begin / _end: the dispatch's own start / stop timestamps of every step-kernel launch
    // (hipExtLaunchKernel's events: valid with further launches queued behind, unlike events recorded between launches)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;            // pairs (start, stop), one per launch
    size_t prof_used = 0;
    // batch partitions (ranenv_set_partitions): envs [part_lo[k], part_lo[k+1]) are stepped by their own launch on
    // their own stream, so that one partition's ramp and tail run under the other partitions' steady state
    int n_parts = 1;
    std::vector<hipStream_t> part_stream;
    std::vector<hipEvent_t> part_done, part_in;
    std::vector<int> part_lo;
    hipEvent_t ev_in = nullptr;
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

size_t NS_all(ranenv_handle h) { return (size_t)h->cfg.n_scenarios * (size_t)h->cfg.n_slices; }

// Inversion tables of the traffic generator, one row per (scenario, slice): cdf[k] = floor(P(X <= k) * 2^64) for
// X ~ Poisson(slice Mbps), k = 0..255 (lower half summed upwards, upper half as 1 - survival, the survival
// function summed from the tail so that the far tail keeps its relative precision), and the 64-entry guide.
int build_poisson_tables(ranenv_handle h, hipStream_t stream)
{
    const size_t rows = NS_all(h);
    if (h->slice_traffic.size() != rows) return RANENV_OK;          // no scenarios yet: built when they are loaded
    std::vector<unsigned long long> cdf(rows * 256, ~0ull);
    std::vector<uint8_t> guide(rows * 64, 0);
    const long double two64 = 18446744073709551616.0L;
    for (size_t r = 0; r < rows; r++) {
        const double lam = h->slice_traffic[r];
        if (!h->slice_has_req[r] || lam == 0.0) continue;          // never sampled
        if (!(lam > 0.0) || lam > 128.0)
            return fail(h, RANENV_E_INVALID, "traffic generator: slice traffic %g Mbps outside (0, 128] (256-entry inversion table)", lam);
        long double pmf[256], ll = logl((long double)lam);
        for (int k = 0; k < 256; k++) pmf[k] = expl((long double)k * ll - (long double)lam - lgammal((long double)k + 1.0L));
        const int mode = (int)lam;
        unsigned long long *c = &cdf[r * 256];
        long double cum = 0.0L;
        for (int k = 0; k <= mode; k++) { cum += pmf[k]; const long double v = floorl(cum * two64); c[k] = v >= two64 ? ~0ull : (unsigned long long)v; }
        long double sf = 0.0L;                                      // P(X > k), from the tail
        for (int k = 255; k > mode; k--) {
            const long double v = ceill(sf * two64);
            c[k] = v <= 0.0L ? ~0ull : (v >= two64 ? 0ull : (unsigned long long)(two64 - v));
            sf += pmf[k];
        }
        c[255] = ~0ull;
        for (int k = 1; k < 256; k++) if (c[k] < c[k - 1]) c[k] = c[k - 1];      // monotone across the seam at the mode
        uint8_t *g = &guide[r * 64];
        int k = 0;
        for (int j = 0; j < 64; j++) {
            const unsigned long long lo = (unsigned long long)j << 58;
            while (k < 255 && c[k] <= lo) k++;
            g[j] = (uint8_t)k;
        }
    }
    if (!h->d_pois_cdf) {
        if (dev_alloc(h, &h->d_pois_cdf, rows * 256) != RANENV_OK || dev_alloc(h, &h->d_pois_guide, rows * 64) != RANENV_OK) return RANENV_E_NOMEM;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_pois_cdf, cdf.data(), cdf.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_pois_guide, guide.data(), guide.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->kp.pois_cdf = h->d_pois_cdf; h->kp.pois_guide = h->d_pois_guide;
    return RANENV_OK;
}

bool persist_tiny(ranenv_handle h);
// RANENV_F_SCALE_PER_ELEMENT: every step / dense launch runs the lean build compiled for that convention -- no mixed blocks, packed
// waves, small-batch / whole-row builds or persistent launches (those exist for the default convention only)
bool scale_per_element(ranenv_handle h) { return (h->cfg.flags & RANENV_F_SCALE_PER_ELEMENT) != 0; }

// The build of the step kernel for this handle: SE gather or streaming (lean / small-batch / whole-row), row width NP.
template <int MODE, int NP, bool MANY>
void launch_kernels_of(ranenv_handle h, const KP &kp, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, bool gather)
{
    // RANENV_F_SCALE_PER_ELEMENT: the lean builds with the other rounding of the masked SE sum (MODE_PE); a reset sums no masked row
    if constexpr (MODE != MODE_RESET) {
        if (scale_per_element(h)) {
            if (gather) {
                if constexpr (MODE == MODE_STEP) {
                    if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel_gather<MODE | MODE_PE, NP, MANY>), grid, block, 0, stream, ev0, ev1, 0, kp);
                    else hipLaunchKernelGGL((ranenv_core_kernel_gather<MODE | MODE_PE, NP, MANY>), grid, block, 0, stream, kp);
                }
            } else if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel<MODE | MODE_PE, NP, MANY>), grid, block, 0, stream, ev0, ev1, 0, kp);
            else hipLaunchKernelGGL((ranenv_core_kernel<MODE | MODE_PE, NP, MANY>), grid, block, 0, stream, kp);
            return;
        }
    }
    if (gather) {
        if constexpr (MODE != MODE_DENSE) {
            if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel_gather<MODE, NP, MANY>), grid, block, 0, stream, ev0, ev1, 0, kp);
            else hipLaunchKernelGGL((ranenv_core_kernel_gather<MODE, NP, MANY>), grid, block, 0, stream, kp);
        }
    } else if (MODE == MODE_STEP && !MANY && h->tiny_step && persist_tiny(h)) {      // a batch at <= 2 waves per SIMD: the whole-row build
        if constexpr (MODE == MODE_STEP && !MANY) {
            if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel_tiny1<NP>), grid, block, 0, stream, ev0, ev1, 0, kp);
            else hipLaunchKernelGGL((ranenv_core_kernel_tiny1<NP>), grid, block, 0, stream, kp);
        }
    } else if (ev0) {       // (the extended launch costs the host several times an ordinary one: only while profiling)
        if (h->small_batch) hipExtLaunchKernelGGL((ranenv_core_kernel_small<MODE, NP, MANY>), grid, block, 0, stream, ev0, ev1, 0, kp);
        else hipExtLaunchKernelGGL((ranenv_core_kernel<MODE, NP, MANY>), grid, block, 0, stream, ev0, ev1, 0, kp);
    } else {
        if (h->small_batch) hipLaunchKernelGGL((ranenv_core_kernel_small<MODE, NP, MANY>), grid, block, 0, stream, kp);
        else hipLaunchKernelGGL((ranenv_core_kernel<MODE, NP, MANY>), grid, block, 0, stream, kp);
    }
}
template <int MODE, int NP>
void launch_kernels(ranenv_handle h, const KP &kp, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, bool gather)
{
    if constexpr (MODE == MODE_STEP) {
        if (kp.n_tti > 1) { launch_kernels_of<MODE, NP, true>(h, kp, grid, block, stream, ev0, ev1, gather); return; }
    }
    launch_kernels_of<MODE, NP, false>(h, kp, grid, block, stream, ev0, ev1, gather);
}

int persist_prepare(ranenv_handle h, hipStream_t stream, bool need_host_counts);
bool persist_tiny(ranenv_handle h);
bool stream_capturing(hipStream_t stream);

// Packed waves address a per-env row as (uniform array base) + (32-bit row + lane offset), see row_at<2>: every array they
// address that way must stay below 4 GB.  True for every size a packed step makes sense at (the reference's: megabytes); a handle
// with pools beyond that steps one env per wave.
bool pack_fits_32(ranenv_handle h)
{
    const unsigned long long lim = 1ull << 32, B = (unsigned long long)h->cfg.batch, U = (unsigned long long)h->cfg.n_ues,
                             S = (unsigned long long)h->cfg.n_slices, NS = (unsigned long long)h->cfg.n_scenarios,
                             D = (unsigned long long)h->cfg.hist_depth, W = 2ull * h->cfg.max_ues_slice + 9ull;
    return (unsigned long long)N_TUE * NS * U * 4 < lim && B * U * 8 < lim && B * D * U * 4 < lim && NS * S * 32 < lim &&
           B * S * W * 4 < lim && (unsigned long long)h->trf_rows_n * U * 4 < lim && (unsigned long long)h->se_tiles_n * U * 8 < lim;
}

// One launch of the step kernel for envs [e0, e0 + n) on `stream` (+ the head kernel when bound).
template <int MODE>
hipError_t launch_range(ranenv_handle h, KP kp, int e0, int n, hipStream_t stream)
{
    kp.e0 = e0;
    const dim3 grid((unsigned)n), block((unsigned)h->nt);
    // SE gather mode: tiles replayed from the pool are read through the sidecars; explicit per-step tiles and dense
    // sched_decisions (whole rows are needed) keep the streaming kernel
    bool gather = false;
    if constexpr (MODE != MODE_DENSE) gather = h->se_mode == RANENV_SE_GATHER && kp.se_tiles == nullptr;
    // Compact steps pay off for the gather kernels throughout (-3...-7 %).  The streaming kernels want lane = UE: their row
    // loads are coalesced in that order (a wave reads 256 contiguous bytes per RB; slice members first scatters its lanes
    // over the whole 400-byte row), so they step compactly only where it was measured to win: under ranenv_rollout's
    // overlapping partitions (-4 %; +15 % for two alternating ranges, +1.5 % for one launch per TTI).
    // Mixed blocks (ranenv_core_kernel_mixed): the whole batch in one launch of one block per wide env + one per two narrow envs -- all of it
    // resident in one round.  For launches of the whole batch of two-wave workgroups, where a compact step is exact.
    bool mixed = false;
    if constexpr (MODE == MODE_STEP) {
        // (whole-batch launches only: for the ranges of a partitioned batch per-range lists were built and measured -- two alternating
        // ranges 47.8 against 48.0 us per TTI in gather mode, and the streaming kernel loses the lane = UE order it wants there: dropped)
        mixed = !scale_per_element(h) && h->mix != 0 && kp.compact != 0 && h->nt == 2 * WAVE && e0 == 0 && n == h->cfg.batch && kp.env_mask == nullptr &&
                (h->mix == 2 || !persist_tiny(h)) && RANENV_DIAG == 0;
        if (mixed && persist_prepare(h, stream, false) != RANENV_OK) return hipErrorUnknown;
    }
    if (!mixed && !gather && kp.compact != 2) kp.compact = 0;
    if (kp.compact) kp.compact = 1;
    if (gather) {
        kp.se_pool = h->d_se_um; kp.se_stride = (long long)h->cfg.n_ues * h->se_rp;
        kp.se_mean_pool = h->d_se_mean; kp.se_rp = h->se_rp;
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (h->prof_on) {                              // two more events from the pool
        while (h->prof_ev.size() < h->prof_used + 2) {
            hipEvent_t e = nullptr;
            const hipError_t ce = hipEventCreate(&e);
            if (ce != hipSuccess) return ce;
            h->prof_ev.push_back(e);
        }
        ev0 = h->prof_ev[h->prof_used]; ev1 = h->prof_ev[h->prof_used + 1];
        h->prof_used += 2;
        h->prof_ttis += MODE == MODE_STEP ? kp.n_tti : 1;
        h->prof_env_ttis += (long long)n * (MODE == MODE_STEP ? kp.n_tti : 1);
    }
    if constexpr (MODE == MODE_STEP) {
        if (mixed) {
            const int B = h->cfg.batch;
            KP kq = kp;
            kq.late = 0;
            kq.p_list = h->d_plist + (size_t)B; kq.m_list = h->d_plist; kq.m_counts = h->d_pcount;
            const dim3 mgrid((unsigned)n), mblock((unsigned)(2 * WAVE));       // (an upper bound of wide + ceil(narrow / 2))
            const bool many = kq.n_tti > 1;
#define MIXED_LAUNCH(NP_, MANY_, GATHER_) do { \
                if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel_mixed<NP_, MANY_, GATHER_>), mgrid, mblock, 0, stream, ev0, ev1, 0, kq); \
                else hipLaunchKernelGGL((ranenv_core_kernel_mixed<NP_, MANY_, GATHER_>), mgrid, mblock, 0, stream, kq); } while (0)
#define MIXED_NP(NP_) do { \
                if (gather) { if (many) MIXED_LAUNCH(NP_, true, true); else MIXED_LAUNCH(NP_, false, true); } \
                else { if (many) MIXED_LAUNCH(NP_, true, false); else MIXED_LAUNCH(NP_, false, false); } } while (0)
            switch (h->np) { case 8: MIXED_NP(8); break; case 10: MIXED_NP(10); break; default: MIXED_NP(16); break; }
#undef MIXED_NP
#undef MIXED_LAUNCH
            if (kp.head_obs || kp.head_reward) hipLaunchKernelGGL(ranenv_head_kernel, grid, dim3((unsigned)h->nslot), 0, stream, kp);
            return hipGetLastError();
        }
        // packed waves: two envs per wave for envs of <= 32 UEs / <= 8 slices (see ranenv_core_kernel_packed)
        if (!scale_per_element(h) && h->pack && h->np == 8 && h->cfg.n_ues <= 32 && h->nt == WAVE && (n & 1) == 0 && kp.env_mask == nullptr && pack_fits_32(h)) {
            KP kq = kp;
            kq.late = 0;
            const dim3 pgrid((unsigned)(n / 2)), pblock((unsigned)WAVE);
            const bool many = kq.n_tti > 1;
#define PACKED_LAUNCH(MANY_, GATHER_) do { \
                if (ev0) hipExtLaunchKernelGGL((ranenv_core_kernel_packed<8, MANY_, GATHER_>), pgrid, pblock, 0, stream, ev0, ev1, 0, kq); \
                else hipLaunchKernelGGL((ranenv_core_kernel_packed<8, MANY_, GATHER_>), pgrid, pblock, 0, stream, kq); } while (0)
            if (gather) { if (many) PACKED_LAUNCH(true, true); else PACKED_LAUNCH(false, true); }
            else { if (many) PACKED_LAUNCH(true, false); else PACKED_LAUNCH(false, false); }
#undef PACKED_LAUNCH
            if (kp.head_obs || kp.head_reward) hipLaunchKernelGGL(ranenv_head_kernel, grid, dim3((unsigned)h->nslot), 0, stream, kp);
            return hipGetLastError();
        }
    }
    switch (h->np) {
    case 8: launch_kernels<MODE, 8>(h, kp, grid, block, stream, ev0, ev1, gather); break;
    case 10: launch_kernels<MODE, 10>(h, kp, grid, block, stream, ev0, ev1, gather); break;
    default: launch_kernels<MODE, 16>(h, kp, grid, block, stream, ev0, ev1, gather); break;
    }
    if (kp.head_obs || kp.head_reward) hipLaunchKernelGGL(ranenv_head_kernel, grid, dim3((unsigned)h->nslot), 0, stream, kp);
    return hipGetLastError();
}

// May the step that `kp` describes leave the UEs outside every slice alone?  Yes when they get no traffic: the device
// generator never draws for them; a traffic pool is examined once per change of pools / scenarios / episodes (a kernel over
// the episode descriptors and, with auto-reset, over the episode table, then one read-back); explicit per-step traffic is
// not examined at all (full width).
int compact_for(ranenv_handle h, const KP &kp, hipStream_t stream, int *out)
{
    *out = 0;
    // a step that may hand idle UEs packets (explicit traffic, an unexamined or offending pool) leaves them with queues that
    // only full-width steps keep ageing: compact steps stay off until a reset of the whole batch
    if (kp.traffic_bits != nullptr || kp.dense != nullptr) { h->idle_state_clean = false; return RANENV_OK; }
    if (kp.trf_gen) { *out = (h->compact_enabled && h->idle_state_clean) ? 1 : 0; return RANENV_OK; }
    if (!kp.trf_pool) return RANENV_OK;
    if (!h->compact_enabled) return RANENV_OK;
    if (h->idle_check_dirty && stream_capturing(stream)) return RANENV_OK;     // (the examination reads back: a captured step that comes before it runs at full width)
    if (h->idle_check_dirty) {
        if (!h->d_violations && dev_alloc(h, &h->d_violations, 2) != RANENV_OK) return RANENV_E_NOMEM;
        HIP_TRY(h, hipMemsetAsync(h->d_violations, 0, 2 * sizeof(int), stream));
        const int U = h->cfg.n_ues;
        hipLaunchKernelGGL(ranenv_idle_traffic_kernel, dim3((unsigned)h->cfg.batch), dim3(256), 0, stream, h->d_episodes, kp.trf_pool, U,
                           TB_ue_slice(h->kp), TB_lane_ue(h->kp), h->d_violations);
        if (h->d_ep_table)
            hipLaunchKernelGGL(ranenv_idle_traffic_kernel, dim3((unsigned)h->ep_table_n), dim3(256), 0, stream, h->d_ep_table, kp.trf_pool, U,
                               TB_ue_slice(h->kp), TB_lane_ue(h->kp), h->d_violations + 1);
        int v[2] = {1, 1};
        HIP_TRY(h, hipMemcpyAsync(v, h->d_violations, sizeof(v), hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
        h->pool_idle_zero = v[0] == 0; h->table_idle_zero = h->d_ep_table ? v[1] == 0 : true;
        h->idle_check_dirty = false;
    }
    const bool zero = h->pool_idle_zero && (!h->ar_on || h->table_idle_zero);
    if (!zero) h->idle_state_clean = false;
    *out = (zero && h->compact_enabled && h->idle_state_clean) ? 1 : 0;
    return RANENV_OK;
}

// One TTI of the whole batch.  Without partitions: one launch on the caller's stream.  With partitions: one launch
// per partition on the partition's own stream; `join_in` orders them behind what the caller's stream holds so far
// (inputs), `join_out` orders the caller's stream behind them (outputs).  ranenv_rollout enqueues n TTIs with a join
// only before the first and after the last: partition k's TTI t+1 then follows its own TTI t directly, whatever the
// other partitions are doing -- envs are independent, nothing else orders them.
template <typename Body>      // Body(e0, n, stream) -> hipError_t: what one partition enqueues for one TTI
hipError_t for_partitions(ranenv_handle h, hipStream_t stream, bool join_in, bool join_out, Body body)
{
    hipError_t le = hipSuccess;
    if (h->n_parts <= 1) {
        le = body(0, h->cfg.batch, stream);
        if (le != hipSuccess) return le;
        // RANENV_F_SYNC_CHECK: surface asynchronous kernel faults at the call that caused them
        if (h->cfg.flags & RANENV_F_SYNC_CHECK) return hipStreamSynchronize(stream);
        return hipSuccess;
    }
    // (nothing pending on the caller's stream = nothing for the partitions to wait for: no event round trip between the queues)
    // (hipStreamQuery is illegal on a capturing stream: a caller that graph-captures its step keeps the event)
    if (join_in) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs == hipStreamCaptureStatusNone && hipStreamQuery(stream) == hipSuccess) join_in = false;
    }
    if (join_in) {
        le = hipEventRecord(h->ev_in, stream);
        if (le != hipSuccess) return le;
    }
    // partition 0 runs on the caller's stream itself (a process has few hardware queues -- 4 by default -- and streams
    // beyond them share one, i.e. run one after the other) and is enqueued first: it waits for no event, so the GPU has
    // work ~10 us after the call instead of after the other partitions' event waits; partitions 1.. on the handle's streams
    le = body(h->part_lo[0], h->part_lo[1] - h->part_lo[0], stream);
    if (le != hipSuccess) return le;
    for (int k = 1; k < h->n_parts; k++) {
        hipStream_t ps = h->part_stream[k];
        if (join_in) { le = hipStreamWaitEvent(ps, h->ev_in, 0); if (le != hipSuccess) return le; }
        le = body(h->part_lo[k], h->part_lo[k + 1] - h->part_lo[k], ps);
        if (le != hipSuccess) return le;
        if (join_out) { le = hipEventRecord(h->part_done[k], ps); if (le != hipSuccess) return le; }
    }
    if (join_out)
        for (int k = 1; k < h->n_parts; k++) { le = hipStreamWaitEvent(stream, h->part_done[k], 0); if (le != hipSuccess) return le; }
    if (h->cfg.flags & RANENV_F_SYNC_CHECK) {
        le = hipStreamSynchronize(stream);
        for (int k = 1; k < h->n_parts && le == hipSuccess; k++) le = hipStreamSynchronize(h->part_stream[k]);
        return le;
    }
    return hipSuccess;
}

// What every launch of a call shares: the host's allocation generation, and whether the next TTI's allocation may be made
// ahead (role 0').  It may only when nothing a later call passes can change it: with the intra-slice scheduler taken from
// the step's own intra_choice argument (RANENV_INTRA_PER_SLICE) an allocation made at the end of TTI t would use TTI t's
// choices for TTI t+1, so there every step allocates at its head and no stored allocation is consumed.
void finalize_kp(ranenv_handle h, KP &kp)
{
    kp.alloc_gen = h->alloc_gen;
    kp.n_tti = 1;
    if (kp.fixed_intra == RANENV_INTRA_PER_SLICE) kp.late = 0;
}

template <int MODE>
hipError_t launch(ranenv_handle h, KP kp, hipStream_t stream, bool join_in = true, bool join_out = true)
{
    finalize_kp(h, kp);
    return for_partitions(h, stream, join_in, join_out,
                          [&](int e0, int n, hipStream_t s) { return launch_range<MODE>(h, kp, e0, n, s); });
}

// Auto-reset: the arguments of the advance kernel for this handle's tables and the caller's buffers
AdvanceArgs advance_args(ranenv_handle h, const uint8_t *dev_done, float *obs_inter, float *obs_intra,
                         float *term_obs_inter, float *term_obs_intra, float *term_obs_head)
{
    const int S = h->cfg.n_slices, Us = h->cfg.max_ues_slice;
    AdvanceArgs a;
    a.done = dev_done; a.mask = h->d_ar_mask; a.episodes = h->d_episodes; a.table = h->d_ep_table;
    a.table_first = h->ep_table_first; a.table_n = h->ep_table_n;
    a.episode_no = ST_episode_no(h->kp); a.reset_count = ST_reset_count(h->kp);
    a.initial = h->ar_initial; a.max_ep = h->ar_max; a.random = h->ar_random; a.env_id_base = h->kp.env_id_base; a.seed = h->ar_seed;
    a.obs_inter = obs_inter; a.obs_intra = obs_intra; a.head_obs = h->kp.head_obs;
    a.term_inter = obs_inter ? term_obs_inter : nullptr; a.term_intra = obs_intra ? term_obs_intra : nullptr; a.term_head = term_obs_head;
    a.n_inter = S * 10; a.n_intra = S * (2 * Us + 9); a.n_head = S * 10;
    a.e0 = 0;
    a.cls_flag = h->d_cls_flag;
    a.acc = h->kp.acc; a.ep_acc = h->d_ep_acc; a.ep_n = h->d_ep_n; a.ep_slots = h->ep_slots;
    return a;
}

int max_steps_of_env(ranenv_handle h, int b) { return h->host_max_steps.empty() ? h->cfg.max_steps : h->host_max_steps[(size_t)b]; }

void shadow_steps_add(ranenv_handle h, int lo, int hi, int n, const uint8_t *done, hipStream_t stream)      // n TTIs enqueued for envs [lo, hi)
{
    if (done) h->last_done = done;
    if (!h->sh_valid) return;
    if (stream_capturing(stream)) { h->sh_valid = false; return; }       // (a graph may be replayed any number of times)
    int32_t *s = h->sh_steps.data();
    for (int b = lo; b < hi; b++) s[b] += n;
}
// envs of [lo, hi) whose episode ended at the TTI enqueued last: -1 = unknown (ask the device), else how many
int shadow_due(ranenv_handle h, int lo, int hi, const uint8_t *dev_done, hipStream_t stream)
{
    if (!h->sh_valid || dev_done == nullptr || dev_done != h->last_done || stream_capturing(stream)) return -1;
    int due = 0;
    for (int b = lo; b < hi; b++) due += h->sh_steps[(size_t)b] >= max_steps_of_env(h, b) ? 1 : 0;
    return due;
}
void shadow_reset_due(ranenv_handle h, int lo, int hi)      // the auto-reset that was just enqueued restarts exactly those envs
{
    if (!h->sh_valid) return;
    for (int b = lo; b < hi; b++) if (h->sh_steps[(size_t)b] >= max_steps_of_env(h, b)) h->sh_steps[(size_t)b] = 0;
}

// ---- persistent rollout (option "persist"), host side ------------------------------------------------------------
hipError_t ensure_streams(ranenv_handle h, size_t n)      // handle-owned streams / events [1, n) exist (index 0 = the caller's stream)
{
    while (h->part_stream.size() < n) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        h->part_stream.push_back(st);
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        h->part_done.push_back(ev);
        ev = nullptr;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        h->part_in.push_back(ev);
    }
    if (!h->ev_in) return hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming);
    return hipSuccess;
}

// A batch whose widest blocks all together stay within 2 waves per SIMD (8 per CU): one class, and -- streaming -- the build
// with the whole SE row in flight.
bool persist_tiny(ranenv_handle h)
{
    return (long long)h->cfg.batch * (h->nt / WAVE) <= 8ll * h->n_cus;
}

template <bool GATHER>
const void *persist_kernel_of(int np)
{
    switch (np) {
    case 8: return reinterpret_cast<const void *>(&ranenv_persist_kernel<GATHER, 8>);
    case 10: return reinterpret_cast<const void *>(&ranenv_persist_kernel<GATHER, 10>);
    default: return reinterpret_cast<const void *>(&ranenv_persist_kernel<GATHER, 16>);
    }
}

template <bool GATHER, int NP>
void launch_persist_of(const KP &kp, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1)
{
    if (ev0) hipExtLaunchKernelGGL((ranenv_persist_kernel<GATHER, NP>), grid, block, 0, stream, ev0, ev1, 0, kp);
    else hipLaunchKernelGGL((ranenv_persist_kernel<GATHER, NP>), grid, block, 0, stream, kp);
}
template <bool GATHER>
void launch_persist(int np, const KP &kp, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1)
{
    switch (np) {
    case 8: launch_persist_of<GATHER, 8>(kp, grid, block, stream, ev0, ev1); break;
    case 10: launch_persist_of<GATHER, 10>(kp, grid, block, stream, ev0, ev1); break;
    default: launch_persist_of<GATHER, 16>(kp, grid, block, stream, ev0, ev1); break;
    }
}

bool stream_capturing(hipStream_t stream)      // (an error of the query itself counts as "capturing": the careful path)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return true; }
    return cs != hipStreamCaptureStatusNone;
}

// The buffers of the work queues (once per handle) and, whenever scenarios / episodes changed, the envs sorted by class.
// Three states of the lists: clean; `pclass_dirty` (the host changed scenarios / episodes, or followed an episode end itself: re-sort);
// `pclass_maybe` (an auto-reset ran: re-sort only if the device's flag says an env restarted -- the host does not read `done`).
// On a CAPTURING stream the sort is enqueued unconditionally and the host's flags stay as they are: what a replay of the graph
// finds in the episode descriptors is not what the host knows now (set_episodes / reset / an eager auto-reset between replays).
int persist_prepare(ranenv_handle h, hipStream_t stream, bool need_host_counts)
{
    const int B = h->cfg.batch, NC = h->nt / WAVE;
    if (!h->d_plist) {
        int cap = 64;
        while (cap < B) cap <<= 1;
        h->p_nclass = NC; h->p_cap = cap;
        if (dev_alloc(h, &h->d_plist, (size_t)NC * B) != RANENV_OK || dev_alloc(h, &h->d_pcount, (size_t)NC) != RANENV_OK ||
            dev_alloc(h, &h->d_pctl, (size_t)NC) != RANENV_OK || dev_alloc(h, &h->d_pslots, (size_t)NC * 8 * (size_t)cap) != RANENV_OK)
            return RANENV_E_NOMEM;
        // the sticky error word lives in host memory the device can write: the host looks at it without a device sync
        HIP_TRY(h, hipHostMalloc((void **)&h->h_perr, sizeof(int), hipHostMallocMapped));
        *h->h_perr = 0;
        HIP_TRY(h, hipHostGetDevicePointer((void **)&h->d_perr_dev, h->h_perr, 0));       // (the same address with unified addressing; asked for, not assumed)
        h->pcount_host.assign((size_t)NC, 0);
        h->pclass_dirty = true;
    }
    const bool capturing = stream_capturing(stream);
    if (capturing && need_host_counts) return fail(h, RANENV_E_STATE, "a persistent rollout reads its class counts back: not inside a stream capture");
    const int one_class = (persist_tiny(h) && h->mix != 2) ? 1 : 0;
    if (h->pclass_dirty || capturing) {
        hipLaunchKernelGGL(ranenv_persist_classify_kernel, dim3(1), dim3(1024), 0, stream, h->d_episodes, h->d_members, B, NC, one_class,
                           h->d_plist, h->d_pcount, h->d_cls_flag, 1);
        if (!capturing) { h->pclass_dirty = false; h->pclass_maybe = false; h->pcount_host_stale = true; }
    } else if (h->pclass_maybe) {
        hipLaunchKernelGGL(ranenv_persist_classify_kernel, dim3(1), dim3(1024), 0, stream, h->d_episodes, h->d_members, B, NC, one_class,
                           h->d_plist, h->d_pcount, h->d_cls_flag, 0);
        h->pclass_maybe = false; h->pcount_host_stale = true;
    }
    if (need_host_counts && h->pcount_host_stale) {       // (the persistent launches size their grids by them; mixed launches read them on the device)
        HIP_TRY(h, hipMemcpyAsync(h->pcount_host.data(), h->d_pcount, sizeof(int32_t) * (size_t)NC, hipMemcpyDeviceToHost, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
        h->pcount_host_stale = false;
    }
    return RANENV_OK;
}

// A wait inside a persistent launch gave up (PersistCtl::abort: every workgroup of that class then drops its env after the
// current chunk, so the envs have advanced different numbers of TTIs).  Seen through the host-visible error word at the next call:
// the queues and cursors are cleared, the persistent rollout is switched off for this handle (the launch-per-chunk rollout
// takes over) and the call fails -- the batch has to be reset.
int persist_check_errors(ranenv_handle h)
{
    if (!h->h_perr || *(volatile int *)h->h_perr == 0) return RANENV_OK;
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemset(h->d_pctl, 0, sizeof(PersistCtl) * (size_t)h->p_nclass));
    HIP_TRY(h, hipMemset(h->d_pslots, 0, sizeof(unsigned long long) * (size_t)h->p_nclass * 8 * (size_t)h->p_cap));
    *(volatile int *)h->h_perr = 0;
    h->perr_seen++;
    h->persist = 0;
    return fail(h, RANENV_E_STATE, "a persistent rollout launch gave up waiting on its work queue (sticky error word): the envs of the batch "
                "have advanced different numbers of TTIs -- reset the batch; the persistent rollout is now off for this handle "
                "(option persist = 0), its queues were cleared");
}

// One persistent launch per non-empty class for `n_tti` TTIs of every env: the class with the widest blocks on the caller's
// stream (enqueued first: a block of several waves needs that many free slots on one CU), the others on handle-owned streams
// between an event pair.
int persist_launch(ranenv_handle h, KP kp, int n_tti, hipStream_t stream)
{
    const int B = h->cfg.batch, NC = h->p_nclass;
    const bool gather = h->se_mode == RANENV_SE_GATHER;
    kp.n_tti = n_tti; kp.late = 0; kp.compact = 1; kp.e0 = 0;
    kp.p_chunk = h->persist_chunk; kp.p_cap = h->p_cap; kp.p_err = h->d_perr_dev;
    if (gather) {
        kp.se_pool = h->d_se_um; kp.se_stride = (long long)h->cfg.n_ues * h->se_rp;
        kp.se_mean_pool = h->d_se_mean; kp.se_rp = h->se_rp;
    }
    const bool tiny = persist_tiny(h);            // (then every env is in the widest class and the grid is the batch)
    int &slots_cu = h->p_wave_slots[gather ? 1 : 0];
    if (slots_cu == 0) {
        int nb = 0;
        const void *fn = gather ? persist_kernel_of<true>(h->np) : persist_kernel_of<false>(h->np);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, WAVE, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 16; }
        slots_cu = nb;
    }
    long long W = (long long)slots_cu * h->n_cus, demand = 0;
    if (h->persist_grid > 0 && h->persist_grid < W) W = h->persist_grid;
    int n_used = 0;
    for (int c = 0; c < NC; c++) { demand += (long long)h->pcount_host[(size_t)c] * (c + 1); n_used += h->pcount_host[(size_t)c] > 0 ? 1 : 0; }
    if (demand == 0) return RANENV_OK;
    hipError_t e = ensure_streams(h, (size_t)(n_used > 1 ? n_used : 1));
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "persistent rollout, streams: %s", hipGetErrorString(e));
    // (the other classes' streams pick up behind what the caller's stream holds -- unless it holds nothing: then there is nothing to
    // wait for, and no signal has to cross between two hardware queues before the largest class may start; not while capturing)
    bool join_in = n_used > 1;
    if (join_in) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs == hipStreamCaptureStatusNone && hipStreamQuery(stream) == hipSuccess) join_in = false;
    }
    if (join_in) HIP_TRY(h, hipEventRecord(h->ev_in, stream));
    int k = 0;                                     // stream index: 0 = the caller's
    for (int c = NC - 1; c >= 0; c--) {
        const int n = h->pcount_host[(size_t)c];
        if (n == 0) continue;
        long long g = demand <= W ? n : (long long)n * W / demand;
        if (g < 1) g = 1;
        if (g > n) g = n;
        hipStream_t s = k == 0 ? stream : h->part_stream[(size_t)k];
        if (k > 0 && join_in) HIP_TRY(h, hipStreamWaitEvent(s, h->ev_in, 0));
        KP kc = kp;
        // (every env of the class has a workgroup of its own and all of them are resident: nobody can ever be waiting, so the
        // launch is one chunk -- no looks at the queues, no staggered first chunk)
        if (tiny && g == n) kc.p_chunk = n_tti > 1 ? n_tti : 1;
        kc.p_list = h->d_plist + (size_t)c * B; kc.p_count = n; kc.p_ctl = h->d_pctl + c;
        kc.p_slots = h->d_pslots + (size_t)c * 8 * (size_t)h->p_cap;
        if (h->persist_inject) {                   // test hook: this launch finds a wait already given up
            const int one = 1;
            HIP_TRY(h, hipMemcpyAsync(&kc.p_ctl->abort, &one, sizeof(int), hipMemcpyHostToDevice, s));      // (the DEVICE then raises the host-visible word)
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->prof_on) {
            while (h->prof_ev.size() < h->prof_used + 2) {
                hipEvent_t pe = nullptr;
                HIP_TRY(h, hipEventCreate(&pe));
                h->prof_ev.push_back(pe);
            }
            ev0 = h->prof_ev[h->prof_used]; ev1 = h->prof_ev[h->prof_used + 1];
            h->prof_used += 2; h->prof_ttis += n_tti; h->prof_env_ttis += (long long)n * n_tti;
        }
        const dim3 grid((unsigned)g), block((unsigned)((c + 1) * WAVE));
        if (gather) launch_persist<true>(h->np, kc, grid, block, s, ev0, ev1);
        else if (tiny) {
            switch (h->np) {
            case 8: if (ev0) hipExtLaunchKernelGGL((ranenv_persist_kernel_tiny<8>), grid, block, 0, s, ev0, ev1, 0, kc); else hipLaunchKernelGGL((ranenv_persist_kernel_tiny<8>), grid, block, 0, s, kc); break;
            case 10: if (ev0) hipExtLaunchKernelGGL((ranenv_persist_kernel_tiny<10>), grid, block, 0, s, ev0, ev1, 0, kc); else hipLaunchKernelGGL((ranenv_persist_kernel_tiny<10>), grid, block, 0, s, kc); break;
            default: if (ev0) hipExtLaunchKernelGGL((ranenv_persist_kernel_tiny<16>), grid, block, 0, s, ev0, ev1, 0, kc); else hipLaunchKernelGGL((ranenv_persist_kernel_tiny<16>), grid, block, 0, s, kc); break;
            }
        }
        else launch_persist<false>(h->np, kc, grid, block, s, ev0, ev1);
        if (k > 0) HIP_TRY(h, hipEventRecord(h->part_done[(size_t)k], s));
        k++;
    }
    for (int j = 1; j < k; j++) HIP_TRY(h, hipStreamWaitEvent(stream, h->part_done[(size_t)j], 0));
    h->persist_inject = 0;
    h->last_rollout_launches += k;
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "persistent rollout launch: %s", hipGetErrorString(e));
    if (h->cfg.flags & RANENV_F_SYNC_CHECK) HIP_TRY(h, hipStreamSynchronize(stream));
    return RANENV_OK;
}

// Tuning / debug options (include/ranenv.h, "Options"): ONE setter behind ranenv_set_option, and ONE place where the
// process environment is read (ranenv_create -> apply_env_options: RANENV_<KEY IN CAPITALS>=value presets the same
// options for handles created afterwards; the test suite and the A/B tools run whole passes under them).
// None of them changes a result: they select a launch schedule or a build of the step kernel.
int set_option(ranenv_handle h, const std::string &k, long long v)
{
    if (k == "compact") { h->compact_enabled = v != 0; return RANENV_OK; }
    if (k == "fuse") { h->fuse = v < 0 ? 0 : (v > 64 ? 64 : (int)v); return RANENV_OK; }
    if (k == "late") { h->kp.late = v < 0 ? 0 : (v > 2 ? 2 : (int)v); h->alloc_gen++; return RANENV_OK; }
    if (k == "row_width") {
        const int m = h->cfg.n_slices > h->cfg.max_ues_slice ? h->cfg.n_slices : h->cfg.max_ues_slice;
        if (!((v == 8 || v == 10 || v == 16) && v >= m))
            return fail(h, RANENV_E_INVALID, "row_width must be 8, 10 or 16 and >= max(S, Us) = %d", m);
        h->np = (int)v;
        return RANENV_OK;
    }
    if (k == "small_batch") { h->small_batch = v != 0; return RANENV_OK; }
    if (k == "tiny_step") { h->tiny_step = v != 0 ? 1 : 0; return RANENV_OK; }
    if (k == "persist") { h->persist = v < 0 ? -1 : (v != 0 ? 1 : 0); return RANENV_OK; }
    if (k == "persist_chunk") { h->persist_chunk = v < 1 ? 1 : (v > 1000 ? 1000 : (int)v); return RANENV_OK; }
    if (k == "persist_grid") { h->persist_grid = v < 0 ? 0 : (int)v; return RANENV_OK; }
    if (k == "pack") { h->pack = v != 0; return RANENV_OK; }
    if (k == "mix") { h->mix = v < 0 ? 0 : (v > 2 ? 2 : (int)v); h->pclass_dirty = true; return RANENV_OK; }
    if (k == "persist_inject_abort") { h->persist_inject = v != 0 ? 1 : 0; return RANENV_OK; }     // test hook, see persist_check_errors
    if (k.rfind("fuse_first", 0) == 0 && k.size() == 11 && k[10] >= '0' && k[10] <= '9') {
        const size_t i = (size_t)(k[10] - '0');
        if (h->fuse_first.size() <= i) h->fuse_first.resize(i + 1, 0);
        h->fuse_first[i] = v < 0 ? 0 : (int)v;
        return RANENV_OK;
    }
    return fail(h, RANENV_E_INVALID, "unknown option '%s'", k.c_str());
}

void apply_env_options(ranenv_handle h)
{
    static const char *const keys[] = {"compact", "fuse", "late", "row_width", "small_batch", "tiny_step", "persist", "persist_chunk", "persist_grid", "pack", "mix"};
    for (const char *key : keys) {
        std::string name = "RANENV_";
        for (const char *c = key; *c; c++) name += (char)toupper((unsigned char)*c);
        if (const char *v = getenv(name.c_str())) (void)set_option(h, key, atoll(v));      // (an unusable value is ignored)
    }
    if (const char *ff = getenv("RANENV_FUSE_FIRST")) {      // a list: a,b,c = partitions 0, 1, 2
        int i = 0;
        for (const char *c = ff; *c && i < 10; i++) {
            (void)set_option(h, std::string("fuse_first") + (char)('0' + i), atoll(c));
            while (*c && *c != ',') c++;
            if (*c) c++;
        }
    }
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
    if (cfg->batch < 1 || S < 1 || S > GRP || U < 1 || U > CORE_NT || R < 1 || R > 512 || Us < 1 || Us > GRP ||
        cfg->rbs_per_rbg < 1 || cfg->rbs_per_rbg > R || cfg->hist_depth < 1 || cfg->hist_depth > 64 ||
        cfg->max_age_cap < 1 || cfg->max_age_cap > 65000 || cfg->max_steps < 1 || cfg->n_scenarios < 1)
        return fail(nullptr, RANENV_E_INVALID,
                    "unsupported sizes: need 1<=S<=16, 1<=U<=256, 1<=R<=512, 1<=Us<=16, 1<=G<=R, 1<=hist_depth<=64");
    if (!(cfg->bandwidth_hz > 0.0)) return fail(nullptr, RANENV_E_INVALID, "bandwidth_hz must be positive");
    if (cfg->flags & ~(RANENV_F_CLEAR_HISTORY_ON_RESET | RANENV_F_NO_RAW_OUTPUT | RANENV_F_SYNC_CHECK | RANENV_F_SCALE_PER_ELEMENT))
        return fail(nullptr, RANENV_E_INVALID, "unknown bits in flags (0x%x)", (unsigned)cfg->flags);
    if (!(cfg->norm_traffic > 0.0) || !(cfg->norm_ues > 0.0) || !(cfg->norm_se > 0.0))
        return fail(nullptr, RANENV_E_INVALID, "norm_traffic, norm_ues, norm_se (the observation's normalisers, agents/ib_sched.py:166-168) must be positive");
    if (R > 128) {   // the row reduction follows numpy's pairwise split two levels deep: every leaf must be <= 128 RBs
        int n2 = R / 2; n2 -= n2 % 8;
        const int halves[2] = {n2, R - n2};
        for (int k = 0; k < 2; k++) {
            int a = halves[k], b = 0;
            if (a > 128) { int hh = a / 2; hh -= hh % 8; b = a - hh; a = hh; }
            if (a > 128 || b > 128)
                return fail(nullptr, RANENV_E_INVALID, "n_rbs %d needs a third level of numpy's pairwise split (a leaf of %d RBs): "
                            "supported are R <= 488 and the R in [489,512] whose quarters stay <= 128", R, a > b ? a : b);
        }
    }
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
    kp.L = (int)L; kp.max_steps = cfg->max_steps; kp.flags = cfg->flags; kp.T = R / cfg->rbs_per_rbg;
    kp.policy = RANENV_POLICY_MARR; kp.fixed_intra = RANENV_INTRA_RR;
    kp.late = RANENV_LATE_DEFAULT;
    kp.bw_hz = cfg->bandwidth_hz; kp.bw_per_rb = cfg->bandwidth_hz / (double)R; kp.over = cfg->overfulfill;
    kp.norm_traffic = cfg->norm_traffic; kp.norm_ues = cfg->norm_ues; kp.norm_se = cfg->norm_se;
    int rc = RANENV_OK;
#define ALLOC(field, count) if (rc == RANENV_OK) rc = dev_alloc(h, &field, (count))
    const size_t NSL = (size_t)S * GRP;
    kp.BU = (long long)(B * U); kp.NSU = (long long)(NS * U); kp.NSL = (long long)(NS * NSL);
    ALLOC(kp.tab.slice_i32, NS * S * 8); ALLOC(kp.tab.slice_f64, NS * S * 2);
    ALLOC(kp.tab.param_i32, 2 * NS * S * 6); ALLOC(kp.tab.param_f64, 2 * NS * S * 3);
    ALLOC(kp.tab.slice_ues, NS * S * Us); ALLOC(kp.tab.slice_usecase, NS * S);
    ALLOC(kp.tab.ue, (size_t)N_TUE * NS * U); ALLOC(kp.tab.slot, 3 * NS * NSL);
    ALLOC(kp.st.u4, (size_t)N_U4 * B * U); ALLOC(kp.st.u8, (size_t)N_U8 * B * U); ALLOC(kp.st.b4, (size_t)N_B4 * B);
    ALLOC(kp.st.age_ring, B * L * U); ALLOC(kp.st.ring_sent, B * D * U); ALLOC(kp.st.ring_drop, B * D * U);
    ALLOC(kp.st.mask_inter, B * S); ALLOC(kp.st.mask_intra, B * S * Us); ALLOC(kp.st.policy_scores, B * S);
    ALLOC(kp.st.next_scores, B * S);
    ALLOC(h->d_episodes, B); ALLOC(h->d_ar_mask, B); ALLOC(h->d_members, NS); ALLOC(h->d_cls_flag, 1);
#undef ALLOC
    if (rc != RANENV_OK) { std::string m = h->err; ranenv_destroy(h); g_last_error = m; return rc; }
    kp.episodes = h->d_episodes;
    h->nt = (U + WAVE - 1) / WAVE * WAVE;               // step kernel: one lane per UE ...
    if (h->nt < (S * 8 + WAVE - 1) / WAVE * WAVE) h->nt = (S * 8 + WAVE - 1) / WAVE * WAVE;   // ... and per slice-table word
    h->nslot = (S * GRP + WAVE - 1) / WAVE * WAVE;      // head kernel: one lane per slot
    {
        const int m = S > Us ? S : Us;
        h->np = m <= 8 ? 8 : (m <= 10 ? 10 : 16);
    }
    {   // fail at create, not at the first step, when the code object has no gfx950 image
        hipFuncAttributes fa;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) {
            h->n_cus = prop.multiProcessorCount;
            h->small_batch = (long long)cfg->batch <= 8ll * prop.multiProcessorCount;
        }
        e = hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&ranenv_core_kernel<MODE_STEP, 16, false>));
        if (e != hipSuccess) {
            ranenv_destroy(h);
            return fail(nullptr, RANENV_E_HIP, "no usable gfx950 kernel image (hipFuncGetAttributes: %s)", hipGetErrorString(e));
        }
    }
    apply_env_options(h);
    *out = h;
    return RANENV_OK;
}

int ranenv_set_option(ranenv_handle h, const char *key, int64_t value)
{
    if (!h || !key) return fail(h, RANENV_E_INVALID, "null argument");
    return set_option(h, key, (long long)value);
}

int ranenv_get_option(ranenv_handle h, const char *key, int64_t *value)
{
    if (!h || !key || !value) return fail(h, RANENV_E_INVALID, "null argument");
    const std::string k(key);
    if (k == "compact") *value = h->compact_enabled ? 1 : 0;
    else if (k == "fuse") *value = h->fuse;
    else if (k == "late") *value = h->kp.late;
    else if (k == "row_width") *value = h->np;
    else if (k == "small_batch") *value = h->small_batch ? 1 : 0;
    else if (k == "tiny_step") *value = h->tiny_step;
    else if (k == "persist") *value = h->persist;
    else if (k == "persist_chunk") *value = h->persist_chunk;
    else if (k == "persist_grid") *value = h->persist_grid;
    else if (k == "pack") *value = h->pack ? 1 : 0;
    else if (k == "mix") *value = h->mix;
    else if (k.rfind("persist_stat_", 0) == 0) {      // keep / push / pop / fresh / idle_polls, summed over classes and XCDs
        static const char *const names[] = {"keep", "push", "pop", "fresh", "idle_polls"};
        int which = -1;
        for (int i = 0; i < 5; i++) if (k == std::string("persist_stat_") + names[i]) which = i;
        if (which < 0) return fail(h, RANENV_E_INVALID, "unknown option '%s'", key);
        long long tot = 0;
        if (h->d_pctl) {
            HIP_TRY(h, hipSetDevice(h->cfg.device)); HIP_TRY(h, hipDeviceSynchronize());
            std::vector<PersistCtl> ctl((size_t)h->p_nclass);
            HIP_TRY(h, hipMemcpy(ctl.data(), h->d_pctl, sizeof(PersistCtl) * ctl.size(), hipMemcpyDeviceToHost));
            for (auto &c : ctl) for (int x = 0; x < 8; x++) tot += (long long)c.stat[x][which];
        }
        *value = tot;
    }
    else if (k == "persist_errors") {          // persistent launches that gave up a wait: reported so far + pending (0 in every correct run)
        int v = 0;
        if (h->h_perr) { HIP_TRY(h, hipSetDevice(h->cfg.device)); HIP_TRY(h, hipDeviceSynchronize()); v = *(volatile int *)h->h_perr != 0 ? 1 : 0; }
        *value = h->perr_seen + v;
    }
    else if (k == "last_rollout_persistent") *value = h->last_rollout_persistent;      // what the last ranenv_rollout call ran:
    else if (k == "last_rollout_launches") *value = h->last_rollout_launches;          // 1 = persistent work-queue launches; step-kernel launches enqueued
    else if (k.rfind("fuse_first", 0) == 0 && k.size() == 11 && k[10] >= '0' && k[10] <= '9')
        *value = (size_t)(k[10] - '0') < h->fuse_first.size() ? h->fuse_first[(size_t)(k[10] - '0')] : 0;
    else return fail(h, RANENV_E_INVALID, "unknown option '%s'", key);
    return RANENV_OK;
}

int ranenv_destroy(ranenv_handle h)
{
    if (!h) return RANENV_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto &e : h->prof_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->part_done) if (e) (void)hipEventDestroy(e);
    for (auto &e : h->part_in) if (e) (void)hipEventDestroy(e);
    for (auto &st : h->part_stream) if (st) (void)hipStreamDestroy(st);
    if (h->ev_in) (void)hipEventDestroy(h->ev_in);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->h_perr) (void)hipHostFree(h->h_perr);
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
    std::vector<int32_t> si(n * S * 8), pi(n * S * 6), bmi(n * S * 6);
    std::vector<double> sf(n * S * 2), pf(n * S * 3), bmf(n * S * 3);
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
        // the same by metric, as intent_drift_calc walks the parameters (agents/common.py:132-335: a later one for the same metric wins)
        for (int m = 0; m < 3; m++) { bmi[(i * 3 + m) * 2] = 0; bmi[(i * 3 + m) * 2 + 1] = 0; bmf[i * 3 + m] = 1.0; }
        for (int k = 0; k < npar; k++) {
            const int m = t->param_metric[i * 3 + k];
            bmi[(i * 3 + m) * 2] = 1; bmi[(i * 3 + m) * 2 + 1] = t->param_op[i * 3 + k]; bmf[i * 3 + m] = t->param_value[i * 3 + k];
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
    // slot tables: slot = slice*16 + position -> UE id and that UE's buffer parameters
    const size_t NSL = (size_t)S * GRP;
    std::vector<int32_t> sue(n * NSL, -1), smp(n * NSL, 1), spk(n * NSL, 1);
    for (size_t i = 0; i < n; i++)
        for (int sl = 0; sl < S; sl++)
            for (int k = 0; k < t->slice_nues[i * S + sl]; k++) {
                const int ue = t->slice_ues[(i * S + sl) * Us + k];
                const size_t o = i * NSL + (size_t)sl * GRP + k;
                if (t->ue_slice[i * U + ue] != sl || t->ue_pos[i * U + ue] != k)
                    return fail(h, RANENV_E_INVALID, "scenario %zu: ue_slice/ue_pos disagree with slice_ues", i);
                sue[o] = ue; smp[o] = t->ue_max_pkts[i * U + ue]; spk[o] = t->ue_pkt_size[i * U + ue];
            }
    // per-UE tables in lane order: a scenario's UEs in slices first (ascending UE id), the idle ones behind
    std::vector<int32_t> lt[N_TUE];
    for (auto &v : lt) v.resize(n * (size_t)U);
    for (size_t i = 0; i < n; i++) {
        int l = 0;
        for (int pass = 0; pass < 2; pass++)
            for (int ue = 0; ue < U; ue++) {
                const size_t o = i * U + ue;
                if ((t->ue_slice[o] >= 0) != (pass == 0)) continue;
                const size_t d = i * U + (size_t)l++;
                lt[0][d] = t->ue_slice[o]; lt[1][d] = t->ue_pos[o]; lt[2][d] = t->ue_pkt_size[o];
                lt[3][d] = t->ue_max_pkts[o]; lt[4][d] = t->ue_max_age[o]; lt[5][d] = ue;
            }
    }
    h->idle_check_dirty = true;                 // which UEs are idle changed: traffic traces are re-examined before compact steps
    if (h->members_host.size() != (size_t)h->cfg.n_scenarios) h->members_host.assign((size_t)h->cfg.n_scenarios, 0);
    for (size_t i = 0; i < n; i++) {
        int m = 0;
        for (int ue = 0; ue < U; ue++) m += t->ue_slice[i * U + ue] >= 0 ? 1 : 0;
        h->members_host[(size_t)first + i] = m;
    }
    h->pclass_dirty = true;
    const size_t f = (size_t)first;
    const Tables &d = h->kp.tab;
    const KP &k = h->kp;
#define PUT(dst, src, elems, type) HIP_TRY(h, hipMemcpyAsync((dst), (src), (elems) * sizeof(type), hipMemcpyHostToDevice, stream))
    PUT(d.slice_i32 + f * S * 8, si.data(), n * S * 8, int32_t);
    PUT(d.slice_f64 + f * S * 2, sf.data(), n * S * 2, double);
    PUT(d.param_i32 + f * S * 6, pi.data(), n * S * 6, int32_t);
    PUT(d.param_f64 + f * S * 3, pf.data(), n * S * 3, double);
    PUT(d.param_i32 + ((size_t)h->cfg.n_scenarios + f) * S * 6, bmi.data(), n * S * 6, int32_t);
    PUT(d.param_f64 + ((size_t)h->cfg.n_scenarios + f) * S * 3, bmf.data(), n * S * 3, double);
    PUT(d.slice_ues + f * S * Us, t->slice_ues, n * S * Us, int32_t);
    PUT(TB_ue_slice(k) + f * U, lt[0].data(), n * U, int32_t);
    PUT(TB_ue_pos(k) + f * U, lt[1].data(), n * U, int32_t);
    PUT(TB_ue_pkt_size(k) + f * U, lt[2].data(), n * U, int32_t);
    PUT(TB_ue_max_pkts(k) + f * U, lt[3].data(), n * U, int32_t);
    PUT(TB_ue_max_age(k) + f * U, lt[4].data(), n * U, int32_t);
    PUT(TB_lane_ue(k) + f * U, lt[5].data(), n * U, int32_t);
    {   // set 1: lane = UE
        const size_t set1 = (size_t)6 * (size_t)k.NSU;
        std::vector<int32_t> ident(n * (size_t)U);
        for (size_t i = 0; i < ident.size(); i++) ident[i] = (int32_t)(i % (size_t)U);
        PUT(TB_ue_slice(k) + set1 + f * U, t->ue_slice, n * U, int32_t);
        PUT(TB_ue_pos(k) + set1 + f * U, t->ue_pos, n * U, int32_t);
        PUT(TB_ue_pkt_size(k) + set1 + f * U, t->ue_pkt_size, n * U, int32_t);
        PUT(TB_ue_max_pkts(k) + set1 + f * U, t->ue_max_pkts, n * U, int32_t);
        PUT(TB_ue_max_age(k) + set1 + f * U, t->ue_max_age, n * U, int32_t);
        PUT(TB_lane_ue(k) + set1 + f * U, ident.data(), n * U, int32_t);
        HIP_TRY(h, hipStreamSynchronize(stream));          // `ident` dies here
    }
    PUT(h->d_members + f, h->members_host.data() + f, n, int32_t);
    PUT(TB_slot_ue(k) + f * NSL, sue.data(), n * NSL, int32_t);
    PUT(TB_slot_mp(k) + f * NSL, smp.data(), n * NSL, int32_t);
    PUT(TB_slot_pk(k) + f * NSL, spk.data(), n * NSL, int32_t);
#undef PUT
    HIP_TRY(h, hipStreamSynchronize(stream));  // staging vectors die at return
    h->have_scenarios = true; h->alloc_gen++;
    if (h->slice_traffic.size() != NS_all(h)) { h->slice_traffic.assign(NS_all(h), 0.0); h->slice_has_req.assign(NS_all(h), 0); }
    for (size_t i = 0; i < n * S; i++) { h->slice_traffic[f * S + i] = t->slice_traffic[i]; h->slice_has_req[f * S + i] = t->slice_has_req[i]; }
    if (h->kp.trf_gen) { const int rc = build_poisson_tables(h, stream); if (rc != RANENV_OK) return rc; }
    return RANENV_OK;
}

int ranenv_bind_se_pool(ranenv_handle h, const float *dev_pool, int64_t n_tiles, int64_t tile_stride)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    h->se_mode = RANENV_SE_STREAM;             // the sidecars describe the pool they were built from
    if (dev_pool == nullptr) {
        h->kp.se_pool = nullptr; h->kp.se_stride = 0; h->se_tiles_n = 0; h->kp.se_quad = 0;
        return RANENV_OK;
    }
    if (n_tiles < 1 || tile_stride < (int64_t)h->cfg.n_ues * h->cfg.n_rbs)
        return fail(h, RANENV_E_INVALID, "SE pool needs n_tiles >= 1 and tile_stride >= U*R");
    h->kp.se_pool = dev_pool; h->kp.se_stride = tile_stride; h->se_tiles_n = n_tiles; h->kp.se_quad = 0;
    h->have_episodes = false;  // descriptors are re-validated against the new pool
    return RANENV_OK;
}

int ranenv_bind_se_pool_quad(ranenv_handle h, const float *dev_pool, int64_t n_tiles, int64_t tile_stride)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    const int64_t need = (int64_t)((h->cfg.n_rbs + 3) / 4) * h->cfg.n_ues * 4;
    if (dev_pool == nullptr) return ranenv_bind_se_pool(h, nullptr, 0, 0);
    if (n_tiles < 1 || tile_stride < need || (tile_stride & 3) != 0 || ((uintptr_t)dev_pool & 15) != 0)
        return fail(h, RANENV_E_INVALID, "RB-quad-major SE pool needs n_tiles >= 1, tile_stride >= ceil(R/4)*U*4 = %lld floats and a multiple of 4, "
                    "and a 16-byte aligned pool", (long long)need);
    h->se_mode = RANENV_SE_STREAM;
    h->kp.se_pool = dev_pool; h->kp.se_stride = tile_stride; h->se_tiles_n = n_tiles; h->kp.se_quad = 1;
    h->have_episodes = false;
    return RANENV_OK;
}

int ranenv_se_retile_quad(const float *dev_rb_major, float *dev_quad, int64_t n_tiles, int32_t n_ues, int32_t n_rbs, void *stream)
{
    if (!dev_rb_major || !dev_quad) return fail(nullptr, RANENV_E_INVALID, "null argument");
    if (n_tiles < 0 || n_ues < 1 || n_rbs < 1) return fail(nullptr, RANENV_E_INVALID, "bad sizes");
    if (((uintptr_t)dev_quad & 15) != 0) return fail(nullptr, RANENV_E_INVALID, "the RB-quad-major pool must be 16-byte aligned");
    const long long n_quads = (long long)n_tiles * ((n_rbs + 3) / 4) * n_ues;
    if (n_quads == 0) return RANENV_OK;
    long long blocks = (n_quads + 255) / 256;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(ranenv_se_retile_quad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dev_rb_major, dev_quad, n_quads,
                       (int)n_ues, (int)n_rbs);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, RANENV_E_HIP, "se_retile_quad launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_bind_traffic_pool(ranenv_handle h, const int32_t *dev_pool, int64_t n_rows)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (dev_pool != nullptr && n_rows < 1) return fail(h, RANENV_E_INVALID, "traffic pool needs n_rows >= 1");
    h->kp.trf_pool = dev_pool; h->trf_rows_n = dev_pool ? n_rows : 0;
    h->have_episodes = false; h->idle_check_dirty = true;
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
        if (h->se_tiles_n > 0 && e.se_base + e.se_len > h->se_tiles_n)
            return fail(h, RANENV_E_INVALID, "env %d: SE trace [%lld,+%d) exceeds the bound pool of %lld tiles", b, (long long)e.se_base, e.se_len, (long long)h->se_tiles_n);
        if (h->kp.trf_pool && e.trf_base + e.trf_len > h->trf_rows_n)
            return fail(h, RANENV_E_INVALID, "env %d: traffic trace [%lld,+%d) exceeds the bound pool of %lld rows", b, (long long)e.trf_base, e.trf_len, (long long)h->trf_rows_n);
    }
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipMemcpyAsync(h->d_episodes, eps, sizeof(ranenv_episode) * (size_t)h->cfg.batch, hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->have_episodes = true; h->alloc_gen++; h->idle_check_dirty = true; h->pclass_dirty = true;
    return RANENV_OK;
}

int ranenv_set_policy(ranenv_handle h, int32_t policy, int32_t fixed_intra)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (policy < RANENV_POLICY_EXTERNAL || policy > RANENV_POLICY_MAPF) return fail(h, RANENV_E_INVALID, "unknown policy %d", policy);
    if (!(fixed_intra == RANENV_INTRA_RR || fixed_intra == RANENV_INTRA_PF || fixed_intra == RANENV_INTRA_MT || fixed_intra == RANENV_INTRA_PER_SLICE))
        return fail(h, RANENV_E_INVALID, "unknown intra-slice scheduler %d", fixed_intra);
    h->kp.policy = policy; h->kp.fixed_intra = fixed_intra; h->alloc_gen++;
    return RANENV_OK;
}

static int check_ready(ranenv_handle h, const float *se_tiles, const double *traffic_bits, bool need_traffic)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (!h->have_scenarios) return fail(h, RANENV_E_STATE, "no scenarios loaded (ranenv_load_scenarios)");
    if (!h->have_episodes) return fail(h, RANENV_E_STATE, "no episode descriptors (ranenv_set_episodes)");
    // (a handle whose sidecars came straight from power -- ranenv_bind_se_gather_from_power -- replays tiles without an RB-major pool)
    if (!se_tiles && !h->kp.se_pool && !(h->se_mode == RANENV_SE_GATHER && h->d_se_mean))
        return fail(h, RANENV_E_STATE, "no SE tiles given and no SE pool bound");
    if (need_traffic && !traffic_bits && !h->kp.trf_pool && !h->kp.trf_gen)
        return fail(h, RANENV_E_STATE, "no traffic given, no traffic pool bound and no traffic generator set");
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
    if (env_mask == nullptr) h->idle_state_clean = true;        // every queue of the batch is empty again
    if (env_mask == nullptr) { h->sh_steps.assign((size_t)h->cfg.batch, 0); h->sh_valid = true; }
    else h->sh_valid = false;                                   // (which envs restart is on the device)
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
    rc = compact_for(h, kp, (hipStream_t)stream, &kp.compact);
    if (rc != RANENV_OK) return rc;
    hipError_t e = launch<MODE_STEP>(h, kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "step launch: %s", hipGetErrorString(e));
    shadow_steps_add(h, 0, h->cfg.batch, 1, done, (hipStream_t)stream);
    return RANENV_OK;
}

int ranenv_step_dense(ranenv_handle h, const uint8_t *dense, const double *traffic_bits, const float *se_tiles,
                      float *obs_inter, float *obs_intra, double *reward, uint8_t *done, void *stream)
{
    int rc = check_ready(h, se_tiles, traffic_bits, true);
    if (rc != RANENV_OK) return rc;
    if (!dense) return fail(h, RANENV_E_INVALID, "null sched_decision");
    if (!se_tiles && !h->kp.se_pool) return fail(h, RANENV_E_STATE, "a dense step reads whole SE rows: it needs explicit tiles or an RB-major pool (this handle has gather sidecars only)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KP kp = h->kp;
    kp.env_mask = nullptr; kp.se_tiles = se_tiles; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = traffic_bits;
    kp.dense = dense; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = done;
    h->idle_state_clean = false;                 // (a dense decision is the facade's path: explicit traffic, any UE)
    hipError_t e = launch<MODE_DENSE>(h, kp, (hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "dense step launch: %s", hipGetErrorString(e));
    shadow_steps_add(h, 0, h->cfg.batch, 1, done, (hipStream_t)stream);
    return RANENV_OK;
}

int ranenv_step_range(ranenv_handle h, int32_t env_first, int32_t env_count, const double *scores, const uint8_t *intra,
                      const double *traffic_bits, const float *se_tiles, float *obs_inter, float *obs_intra, double *reward,
                      uint8_t *done, void *stream)
{
    int rc = check_ready(h, se_tiles, traffic_bits, true);
    if (rc != RANENV_OK) return rc;
    if (env_first < 0 || env_count < 1 || (long long)env_first + env_count > h->cfg.batch)
        return fail(h, RANENV_E_INVALID, "envs [%d,%d) outside the batch of %d", env_first, env_first + env_count, h->cfg.batch);
    if (!scores && h->kp.policy == RANENV_POLICY_EXTERNAL) return fail(h, RANENV_E_STATE, "policy is EXTERNAL but no inter-slice scores were given");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    KP kp = h->kp;
    kp.env_mask = nullptr; kp.se_tiles = se_tiles; kp.scores = scores; kp.intra = intra; kp.traffic_bits = traffic_bits;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = done;
    finalize_kp(h, kp);
    rc = compact_for(h, kp, (hipStream_t)stream, &kp.compact);
    if (rc != RANENV_OK) return rc;
    hipError_t e = launch_range<MODE_STEP>(h, kp, env_first, env_count, (hipStream_t)stream);
    if (e == hipSuccess && (h->cfg.flags & RANENV_F_SYNC_CHECK)) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "step launch (envs [%d,%d)): %s", env_first, env_first + env_count, hipGetErrorString(e));
    shadow_steps_add(h, env_first, env_first + env_count, 1, done, (hipStream_t)stream);
    return RANENV_OK;
}

int ranenv_step_part(ranenv_handle h, int32_t part, const double *scores, const uint8_t *intra, const double *traffic_bits,
                     const float *se_tiles, float *obs_inter, float *obs_intra, double *reward, uint8_t *done, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (part < 0 || part >= h->n_parts || h->part_lo.empty()) return fail(h, RANENV_E_INVALID, "partition %d outside [0,%d) (ranenv_set_partitions)", part, h->n_parts);
    hipStream_t stream = (hipStream_t)stream_, ps = h->part_stream[(size_t)part];
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    // the partition's stream picks up behind what the caller's stream holds now (the producer of the scores) -- unless the
    // caller works on the partition's stream itself (ranenv_get_part_stream): then stream order is all that is needed, and
    // no signal crosses between hardware queues (a cross-queue dependency costs ~15 us each way on this GPU)
    if (stream != ps) {
        HIP_TRY(h, hipEventRecord(h->part_in[(size_t)part], stream));
        HIP_TRY(h, hipStreamWaitEvent(ps, h->part_in[(size_t)part], 0));
    }
    const int rc = ranenv_step_range(h, h->part_lo[(size_t)part], h->part_lo[(size_t)part + 1] - h->part_lo[(size_t)part], scores, intra,
                                     traffic_bits, se_tiles, obs_inter, obs_intra, reward, done, ps);
    if (rc != RANENV_OK) return rc;
    // ... and leaves an event for ranenv_wait_part
    HIP_TRY(h, hipEventRecord(h->part_done[(size_t)part], ps));
    return RANENV_OK;
}

int ranenv_wait_part(ranenv_handle h, int32_t part, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (part < 0 || part >= h->n_parts || h->part_lo.empty()) return fail(h, RANENV_E_INVALID, "partition %d outside [0,%d) (ranenv_set_partitions)", part, h->n_parts);
    if ((hipStream_t)stream_ != h->part_stream[(size_t)part])
        HIP_TRY(h, hipStreamWaitEvent((hipStream_t)stream_, h->part_done[(size_t)part], 0));
    return RANENV_OK;
}

int ranenv_get_part_stream(ranenv_handle h, int32_t part, void **stream)
{
    if (!h || !stream) return fail(h, RANENV_E_INVALID, "null argument");
    if (part < 0 || part >= h->n_parts || h->part_lo.empty()) return fail(h, RANENV_E_INVALID, "partition %d outside [0,%d) (ranenv_set_partitions)", part, h->n_parts);
    *stream = (void *)h->part_stream[(size_t)part];
    return RANENV_OK;
}

int ranenv_get_partition(ranenv_handle h, int32_t part, int32_t *env_first, int32_t *env_count)
{
    if (!h || !env_first || !env_count) return fail(h, RANENV_E_INVALID, "null argument");
    if (h->part_lo.empty()) { if (part != 0) return fail(h, RANENV_E_INVALID, "partition %d outside [0,1)", part); *env_first = 0; *env_count = h->cfg.batch; return RANENV_OK; }
    if (part < 0 || part >= h->n_parts) return fail(h, RANENV_E_INVALID, "partition %d outside [0,%d)", part, h->n_parts);
    *env_first = h->part_lo[(size_t)part]; *env_count = h->part_lo[(size_t)part + 1] - h->part_lo[(size_t)part];
    return RANENV_OK;
}

int ranenv_set_se_mode(ranenv_handle h, int32_t mode, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (mode != RANENV_SE_STREAM && mode != RANENV_SE_GATHER) return fail(h, RANENV_E_INVALID, "unknown SE mode %d", mode);
    if (mode == RANENV_SE_STREAM) {
        if (!h->kp.se_pool && h->se_mode == RANENV_SE_GATHER)
            return fail(h, RANENV_E_STATE, "this handle's sidecars came straight from power: there is no RB-major pool to stream (ranenv_bind_se_pool)");
        h->se_mode = RANENV_SE_STREAM; return RANENV_OK;
    }
    if (!h->kp.se_pool) {
        if (h->d_se_mean && h->se_tiles_n > 0) { h->se_mode = RANENV_SE_GATHER; return RANENV_OK; }      // sidecars straight from power
        return fail(h, RANENV_E_STATE, "the SE gather mode needs a bound SE pool (ranenv_bind_se_pool) or sidecars from power (ranenv_bind_se_gather_from_power)");
    }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    const int U = h->cfg.n_ues, R = h->cfg.n_rbs, Rp = (R + 7) & ~7;
    const size_t nt = (size_t)h->se_tiles_n;
    // (re)build the sidecars for the pool as it is now: older ones are released first
    auto drop = [&](void *ptr) {
        if (!ptr) return;
        for (size_t i = 0; i < h->allocs.size(); i++) if (h->allocs[i] == ptr) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
        (void)hipFree(ptr);
    };
    HIP_TRY(h, hipDeviceSynchronize());
    drop(h->d_se_mean); drop(h->d_se_um); h->d_se_mean = nullptr; h->d_se_um = nullptr;
    void *pm = nullptr, *pu = nullptr;
    hipError_t e = hipMalloc(&pm, nt * (size_t)U * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&pu, nt * (size_t)U * (size_t)Rp * sizeof(float));
    if (e != hipSuccess) {
        if (pm) (void)hipFree(pm);
        return fail(h, RANENV_E_NOMEM, "SE gather sidecars (%zu tiles: %.2f GB): %s", nt,
                    (double)(nt * (size_t)U * (8 + 4 * (size_t)Rp)) / 1e9, hipGetErrorString(e));
    }
    h->allocs.push_back(pm); h->allocs.push_back(pu);
    h->d_se_mean = (double *)pm; h->d_se_um = (float *)pu; h->se_rp = Rp;
    for (size_t t0 = 0; t0 < nt; t0 += 1u << 20) {              // grid.x stays far below its limit
        const size_t n = nt - t0 < (1u << 20) ? nt - t0 : (1u << 20);
        hipLaunchKernelGGL(ranenv_se_sidecar_kernel, dim3((unsigned)n), dim3((unsigned)h->nt), 0, stream, h->kp.se_pool,
                           (long long)h->kp.se_stride, (long long)t0, U, R, Rp, h->kp.se_quad, h->d_se_mean, h->d_se_um);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "SE sidecar launch: %s", hipGetErrorString(e));
    // The sidecars are read by launches on other streams (the partitions' own): a one-off multi-GB build that started with a
    // device synchronisation also ends with one, instead of an event every partition stream would have to wait for.
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->se_mode = RANENV_SE_GATHER;
    return RANENV_OK;
}

int ranenv_bind_se_gather_from_power(ranenv_handle h, const double *dev_power, int64_t n_tiles, double tx_power_per_rb,
                                     double noise_power, void *stream_)
{
    if (!h || !dev_power) return fail(h, RANENV_E_INVALID, "null argument");
    if (n_tiles < 1) return fail(h, RANENV_E_INVALID, "n_tiles must be >= 1");
    if (!(noise_power > 0.0)) return fail(h, RANENV_E_INVALID, "noise_power must be positive");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    const int U = h->cfg.n_ues, R = h->cfg.n_rbs, Rp = (R + 7) & ~7;
    const size_t nt = (size_t)n_tiles;
    auto drop = [&](void *ptr) {
        if (!ptr) return;
        for (size_t i = 0; i < h->allocs.size(); i++) if (h->allocs[i] == ptr) { h->allocs.erase(h->allocs.begin() + (long)i); break; }
        (void)hipFree(ptr);
    };
    HIP_TRY(h, hipDeviceSynchronize());
    drop(h->d_se_mean); drop(h->d_se_um); h->d_se_mean = nullptr; h->d_se_um = nullptr;
    void *pm = nullptr, *pu = nullptr;
    hipError_t e = hipMalloc(&pm, nt * (size_t)U * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&pu, nt * (size_t)U * (size_t)Rp * sizeof(float));
    if (e != hipSuccess) {
        if (pm) (void)hipFree(pm);
        return fail(h, RANENV_E_NOMEM, "SE gather sidecars (%zu tiles: %.2f GB): %s", nt, (double)(nt * (size_t)U * (8 + 4 * (size_t)Rp)) / 1e9, hipGetErrorString(e));
    }
    h->allocs.push_back(pm); h->allocs.push_back(pu);
    h->d_se_mean = (double *)pm; h->d_se_um = (float *)pu; h->se_rp = Rp;
    for (size_t t0 = 0; t0 < nt; t0 += 1u << 20) {
        const size_t n = nt - t0 < (1u << 20) ? nt - t0 : (1u << 20);
        hipLaunchKernelGGL(ranenv_se_sidecar_from_power_kernel, dim3((unsigned)n), dim3((unsigned)h->nt), 0, stream, dev_power, (long long)t0,
                           U, R, Rp, tx_power_per_rb, noise_power, h->d_se_mean, h->d_se_um);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "SE sidecar-from-power launch: %s", hipGetErrorString(e));
    HIP_TRY(h, hipStreamSynchronize(stream));        // (read by launches on other streams; the power array may be freed by the caller now)
    h->kp.se_pool = nullptr; h->kp.se_stride = 0;    // no RB-major pool: pooled tiles exist as sidecars only
    h->se_tiles_n = n_tiles; h->se_mode = RANENV_SE_GATHER;
    h->have_episodes = false;                        // descriptors are re-validated against the new tile count
    return RANENV_OK;
}

int ranenv_get_se_sidecars(ranenv_handle h, double **dev_row_mean, float **dev_ue_major, int32_t *row_floats)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (!h->d_se_mean) return fail(h, RANENV_E_STATE, "no SE sidecars (ranenv_set_se_mode GATHER builds them)");
    if (dev_row_mean) *dev_row_mean = h->d_se_mean;
    if (dev_ue_major) *dev_ue_major = h->d_se_um;
    if (row_floats) *row_floats = h->se_rp;
    return RANENV_OK;
}

int ranenv_profile_begin(ranenv_handle h)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->prof_used = 0; h->prof_ttis = 0; h->prof_env_ttis = 0; h->prof_on = true;
    return RANENV_OK;
}

int ranenv_profile_ttis(ranenv_handle h, int64_t *n_ttis)
{
    if (!h || !n_ttis) return fail(h, RANENV_E_INVALID, "null argument");
    *n_ttis = (int64_t)h->prof_ttis;
    return RANENV_OK;
}

int ranenv_profile_work(ranenv_handle h, int64_t *n_ttis, int64_t *n_env_ttis)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (n_ttis) *n_ttis = (int64_t)h->prof_ttis;
    if (n_env_ttis) *n_env_ttis = (int64_t)h->prof_env_ttis;
    return RANENV_OK;
}

int ranenv_profile_end(ranenv_handle h, double *avg_ms, int32_t *n_launches)
{
    if (!h || !avg_ms || !n_launches) return fail(h, RANENV_E_INVALID, "null argument");
    if (!h->prof_on) return fail(h, RANENV_E_STATE, "ranenv_profile_begin was not called");
    h->prof_on = false;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    double acc = 0.0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
        acc += (double)ms;
    }
    *n_launches = (int32_t)(h->prof_used / 2);
    *avg_ms = h->prof_used ? acc / (double)(h->prof_used / 2) : 0.0;
    return RANENV_OK;
}

int ranenv_set_partitions(ranenv_handle h, int32_t n_parts)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (n_parts < 1 || n_parts > 16 || n_parts > h->cfg.batch) return fail(h, RANENV_E_INVALID, "n_parts must be in [1, min(16, batch)]");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    while ((int)h->part_stream.size() < n_parts) {
        hipStream_t st = nullptr; hipEvent_t ev = nullptr;
        HIP_TRY(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        h->part_stream.push_back(st);
        HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->part_done.push_back(ev);
        ev = nullptr;
        HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->part_in.push_back(ev);
    }
    if (!h->ev_in) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_in, hipEventDisableTiming));
    h->part_lo.assign((size_t)n_parts + 1, 0);
    // (an even batch is cut into even ranges where that is possible: packed waves step two envs each, ranenv_core_kernel_packed)
    const int B = h->cfg.batch, unit = (B % 2 == 0 && B / 2 >= n_parts) ? 2 : 1;
    const int base = (B / unit) / n_parts, rem = (B / unit) % n_parts;
    for (int k = 0; k < n_parts; k++) h->part_lo[k + 1] = h->part_lo[k] + unit * (base + (k < rem ? 1 : 0));
    h->n_parts = n_parts;
    return RANENV_OK;
}

int ranenv_rollout(ranenv_handle h, int32_t n_steps, float *obs_inter, float *obs_intra, double *reward, uint8_t *done, void *stream_)
{
    int rc = check_ready(h, nullptr, nullptr, true);
    if (rc != RANENV_OK) return rc;
    if (n_steps < 1) return fail(h, RANENV_E_INVALID, "n_steps must be >= 1");
    if (h->kp.policy == RANENV_POLICY_EXTERNAL) return fail(h, RANENV_E_STATE, "a rollout needs a device policy (ranenv_set_policy MARR / MAPF)");
    const bool have_se = h->kp.se_pool != nullptr || (h->se_mode == RANENV_SE_GATHER && h->d_se_mean != nullptr);
    if (!have_se || (!h->kp.trf_pool && !h->kp.trf_gen)) return fail(h, RANENV_E_STATE, "a rollout replays the bound SE pool and traffic pool / generator");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    rc = persist_check_errors(h);                  // (of the persistent launches of earlier calls that have completed)
    if (rc != RANENV_OK) return rc;
    KP kp = h->kp;
    kp.env_mask = nullptr; kp.se_tiles = nullptr; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = nullptr;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = reward; kp.done = done;
    hipStream_t stream = (hipStream_t)stream_;
    h->last_rollout_persistent = 0; h->last_rollout_launches = 0;
    finalize_kp(h, kp);
    rc = compact_for(h, kp, stream, &kp.compact);
    if (rc != RANENV_OK) return rc;
    if (kp.compact) kp.compact = 2;                 // (2: the streaming kernels may step compactly too, see launch_range)
    // With auto-reset on, an env whose episode ends inside the rollout moves on to its next episode without the host:
    // the advance kernel + the step kernel in RESET mode follow that TTI's step on the partition's stream.  They are only
    // enqueued for TTIs at which some env of the partition finishes: the step counters are read once here and followed
    // on the host (nothing but this rollout changes them until it returns).
    std::vector<int32_t> steps;
    AdvanceArgs adv{};
    KP kpr = kp;
    if (h->ar_on) {
        if (!done) return fail(h, RANENV_E_INVALID, "a rollout with auto-reset needs the done buffer");
        steps.resize((size_t)h->cfg.batch);
        HIP_TRY(h, hipStreamSynchronize(stream));
        HIP_TRY(h, hipMemcpy(steps.data(), ST_step_no(h->kp), sizeof(int32_t) * steps.size(), hipMemcpyDeviceToHost));
        adv = advance_args(h, done, obs_inter, obs_intra, nullptr, nullptr, nullptr);
        kpr.env_mask = h->d_ar_mask; kpr.reward = nullptr; kpr.done = nullptr; kpr.compact = 0;
        kpr.head_reward = nullptr;               // the terminal transition's head rewards stay, like reward / done
    }
    const bool follow = h->ar_on;
    // Option "persist": one persistent work-queue launch per workgroup class for all the TTIs up to the next episode end
    // (ranenv_persist_kernel), on the caller's stream (+ one handle-owned stream per further class), whatever the partitions.
    // Needs compact steps (the classes are those of the compact lane order) and no head kernel behind every TTI.
    // (auto, streaming: rollouts of 16...64 TTIs of a batch the chip holds at once -- see below)
    const bool stream_short = h->se_mode != RANENV_SE_GATHER && !h->small_batch && (long long)h->cfg.batch <= 20ll * h->n_cus && n_steps >= 16 && n_steps <= 64;
    const bool persist_wanted = (RANENV_DIAG == 0 || RANENV_DIAG == 12) && (h->persist == 1 || (h->persist < 0 && ((h->se_mode == RANENV_SE_GATHER && !h->small_batch && (long long)h->cfg.batch <= 44ll * h->n_cus) || persist_tiny(h) || stream_short)));
    // (auto: where it was measured to win or tie -- profiles/r04_ab_log.txt.  Gather mode: B 1024 -4...-6 %, 2048 -1 %, 4096 -6 %, 8192 -2 % per
    // TTI; a batch of several times what the chip holds -- 16 384 one-wave envs at the reference's own size -- swaps at every chunk and
    // loses 7 %.  Streaming: -10 % at <= 2 waves per SIMD with the whole-row build; at B 4096 a tie: six same-box pairs against the
    // launches of <= 10 TTIs over three partitions, between -5 and +6 % for rollouts of 200 TTIs (mean +0.2 %) and between -1 and +4 % for
    // rollouts of 20 (mean +0.6 %) -- the streaming kernel is bound by HBM either way -- so there it stayed off unless asked for.
    // Round 5, RB-quad-major pool + non-temporal tile loads: same-box pairs on five boxes (profiles/r05_ab_log.txt) give -0.5...-3 % for
    // rollouts of 20 (mean -1.6 %), -5 % for 16, about -1 % for 40...100, a tie at 200 and +9 % for rollouts of 10 (the staggered first chunk is
    // most of such a call); B 8192 loses 7 % (more workgroups than slots: every chunk swaps).  Hence: on for 16...64 TTIs at <= 20 envs per CU.)
    // (auto: not when episodes end at many different TTIs inside this call -- per-env episode lengths, envs reset at different times:
    // every episode end ends the persistent launches, re-sorts the envs and reads the class counts back; the launch-per-chunk
    // rollout follows the ends per partition without a host sync)
    bool persist_ok = persist_wanted && !scale_per_element(h) && kp.compact != 0 && !(kp.head_obs || kp.head_reward) && (h->cfg.batch >> PERSIST_ENV_BITS) == 0 &&
                      !stream_capturing(stream);      // (it reads the class counts back)
    if (persist_ok && h->persist < 0 && follow) {
        std::vector<int> ends;
        for (int b = 0; b < h->cfg.batch && ends.size() <= 2; b++) {
            const int d = max_steps_of_env(h, b) - steps[(size_t)b];
            if (d < n_steps && std::find(ends.begin(), ends.end(), d) == ends.end()) ends.push_back(d);
        }
        if (ends.size() > 2) persist_ok = false;
    }
    if (persist_ok) {
        h->last_rollout_persistent = 1;
        for (int done_ttis = 0; done_ttis < n_steps;) {
            int n_tti = n_steps - done_ttis;
            if (follow) {
                for (int b = 0; b < h->cfg.batch; b++) {
                    const int d = max_steps_of_env(h, b) - steps[(size_t)b];
                    if (d < n_tti) n_tti = d;
                }
                if (n_tti < 1) n_tti = 1;
            }
            if (n_tti >= (1 << (31 - PERSIST_ENV_BITS))) n_tti = (1 << (31 - PERSIST_ENV_BITS)) - 1;
            rc = persist_prepare(h, stream, true);
            if (rc != RANENV_OK) return rc;
            rc = persist_launch(h, kp, n_tti, stream);
            if (rc != RANENV_OK) return rc;
            done_ttis += n_tti;
            if (!follow) continue;
            bool any = false;
            for (int b = 0; b < h->cfg.batch; b++) {
                steps[(size_t)b] += n_tti;
                if (steps[(size_t)b] >= max_steps_of_env(h, b)) { any = true; steps[(size_t)b] = 0; }
            }
            if (!any) continue;
            h->pclass_dirty = true;               // the restarted envs' scenarios
            hipLaunchKernelGGL(ranenv_advance_kernel, dim3((unsigned)h->cfg.batch), dim3(64), 0, stream, adv);
            const hipError_t re = launch_range<MODE_RESET>(h, kpr, 0, h->cfg.batch, stream);
            if (re != hipSuccess) return fail(h, RANENV_E_HIP, "persistent rollout, reset launch: %s", hipGetErrorString(re));
        }
        if (follow) { h->sh_steps = steps; h->sh_valid = true; h->last_done = done; }      // (read from the device above, followed exactly since)
        else shadow_steps_add(h, 0, h->cfg.batch, n_steps, done, stream);
        return RANENV_OK;
    }
    // A launch takes its envs through several TTIs where nothing has to happen in between (see step_loop): no head kernel
    // behind every step, and -- with auto-reset -- no episode end before the launch's last TTI.  How many: a quarter of
    // the rollout, at most 10 (measured, profiles/r03_ab_log.txt: longer launches gain nothing more and lengthen the
    // drain at the rollout's end, where the workgroups that waited for a free slot run last and alone).
    int fuse = h->fuse > 0 ? h->fuse : (n_steps / 4 < 1 ? 1 : (n_steps / 4 > 10 ? 10 : n_steps / 4));
    if (kp.head_obs || kp.head_reward) fuse = 1;
    auto max_steps_of = [&](int b) { return h->host_max_steps.empty() ? h->cfg.max_steps : h->host_max_steps[(size_t)b]; };
    // Every partition walks through the n_steps TTIs in launches of its own: `pdone[k]` TTIs are enqueued for partition k.
    const int np = h->n_parts > 1 ? h->n_parts : 1;
    std::vector<int> pdone((size_t)np, 0), pn((size_t)np, 0);
    auto part_of = [&](int e0) { for (int k = 0; k < np; k++) if (np > 1 && h->part_lo[k] == e0) return k; return 0; };
    for (int round = 0;; round++) {
        bool any_left = false, last = true;
        for (int k = 0; k < np; k++) {
            const int left = n_steps - pdone[(size_t)k];
            int n_tti = left < fuse ? left : fuse;
            if (round == 0 && fuse > 1 && np > 1) {
                // The partitions' first launches differ in length, the one enqueued last (the highest partition)
                // starting with a single TTI: it is the one whose workgroups find the slots taken (4096 envs want
                // 3738), and after one short launch its late starters are through instead of holding its chain up for a
                // whole long one; from then on the partitions' launch boundaries no longer coincide (K = 20: -2 % streaming,
                // -4 % gather; profiles/r03_ab_log.txt).  RANENV_FUSE_FIRST=a,b,c overrides (0 = the common length).
                int first = k == np - 1 ? 1 : ((k & 1) ? (3 * fuse + 4) / 5 : fuse);
                if (!h->fuse_first.empty()) first = (size_t)k < h->fuse_first.size() ? h->fuse_first[(size_t)k] : 0;
                if (first > 0 && first < n_tti) n_tti = first;
            }
            if (follow && n_tti > 1) {
                int to_end = n_tti;              // TTIs until the first episode of the partition ends (that TTI included)
                const int lo = np > 1 ? h->part_lo[k] : 0, hi = np > 1 ? h->part_lo[k + 1] : h->cfg.batch;
                for (int b = lo; b < hi; b++) {
                    const int d = max_steps_of(b) - steps[(size_t)b];
                    if (d < to_end) to_end = d;
                }
                n_tti = to_end < 1 ? 1 : to_end;
            }
            pn[(size_t)k] = n_tti > 0 ? n_tti : 0;
            if (pn[(size_t)k] > 0) any_left = true;
            if (pdone[(size_t)k] + pn[(size_t)k] < n_steps) last = false;
        }
        if (!any_left) break;
        const hipError_t e = for_partitions(h, stream, round == 0, last, [&](int e0, int n, hipStream_t s) -> hipError_t {
            const int n_tti = pn[(size_t)part_of(e0)];
            if (n_tti == 0) return hipSuccess;                                    // this partition is through
            KP kpk = kp;
            kpk.n_tti = n_tti;
            h->last_rollout_launches++;
            hipError_t le = launch_range<MODE_STEP>(h, kpk, e0, n, s);
            if (le != hipSuccess || !follow) return le;
            bool any = false;
            for (int b = e0; b < e0 + n; b++) {
                steps[(size_t)b] += n_tti;
                if (steps[(size_t)b] >= max_steps_of(b)) { any = true; steps[(size_t)b] = 0; }
            }
            if (!any) return hipSuccess;
            AdvanceArgs a = adv; a.e0 = e0;
            h->pclass_dirty = true;
            hipLaunchKernelGGL(ranenv_advance_kernel, dim3((unsigned)n), dim3(64), 0, s, a);
            return launch_range<MODE_RESET>(h, kpr, e0, n, s);
        });
        if (e != hipSuccess) return fail(h, RANENV_E_HIP, "rollout, round %d of launches: %s", round, hipGetErrorString(e));
        for (int k = 0; k < np; k++) pdone[(size_t)k] += pn[(size_t)k];
    }
    if (follow) { h->sh_steps = steps; h->sh_valid = true; h->last_done = done; }
    else shadow_steps_add(h, 0, h->cfg.batch, n_steps, done, stream);
    return RANENV_OK;
}

int ranenv_enable_metrics(ranenv_handle h, int32_t episode_slots, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (episode_slots < 0) { h->kp.acc = nullptr; return RANENV_OK; }      // off (what was accumulated stays readable)
    if (h->d_acc && episode_slots != h->ep_slots)
        return fail(h, RANENV_E_STATE, "episode metrics were enabled with %d slots per env", h->ep_slots);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    const size_t B = (size_t)h->cfg.batch;
    if (!h->d_acc) {
        if (dev_alloc(h, &h->d_acc, B * 8) != RANENV_OK || dev_alloc(h, &h->d_ep_n, B) != RANENV_OK) return RANENV_E_NOMEM;
        if (episode_slots > 0 && dev_alloc(h, &h->d_ep_acc, B * (size_t)episode_slots * 8) != RANENV_OK) return RANENV_E_NOMEM;
        h->ep_slots = episode_slots;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_acc, 0, sizeof(double) * B * 8, stream));
    HIP_TRY(h, hipMemsetAsync(h->d_ep_n, 0, sizeof(int32_t) * B, stream));
    if (h->d_ep_acc) HIP_TRY(h, hipMemsetAsync(h->d_ep_acc, 0, sizeof(double) * B * (size_t)h->ep_slots * 8, stream));
    h->kp.acc = h->d_acc;
    return RANENV_OK;
}

int ranenv_get_metrics(ranenv_handle h, double **dev_running, double **dev_episode_log, int32_t **dev_episodes_done, int32_t *episode_slots)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (!h->d_acc) return fail(h, RANENV_E_STATE, "episode metrics are not enabled (ranenv_enable_metrics)");
    if (dev_running) *dev_running = h->d_acc;
    if (dev_episode_log) *dev_episode_log = h->d_ep_acc;
    if (dev_episodes_done) *dev_episodes_done = h->d_ep_n;
    if (episode_slots) *episode_slots = h->ep_slots;
    return RANENV_OK;
}

int ranenv_set_traffic_generator(ranenv_handle h, int32_t enable, uint64_t seed, int32_t env_id_base, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (!enable) { h->kp.trf_gen = 0; return RANENV_OK; }
    if (env_id_base < 0) return fail(h, RANENV_E_INVALID, "env_id_base must be >= 0");
    h->kp.trf_seed = seed; h->kp.env_id_base = env_id_base;
    h->kp.trf_gen = 1;
    const int rc = build_poisson_tables(h, (hipStream_t)stream_);
    if (rc != RANENV_OK) h->kp.trf_gen = 0;
    return rc;
}

int ranenv_get_poisson_tables(ranenv_handle h, uint64_t *host_cdf, uint8_t *host_guide)
{
    if (!h || !host_cdf || !host_guide) return fail(h, RANENV_E_INVALID, "null argument");
    if (!h->kp.trf_gen || !h->d_pois_cdf) return fail(h, RANENV_E_STATE, "the traffic generator is not enabled");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(host_cdf, h->d_pois_cdf, NS_all(h) * 256 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(host_guide, h->d_pois_guide, NS_all(h) * 64, hipMemcpyDeviceToHost));
    return RANENV_OK;
}

int ranenv_set_max_steps(ranenv_handle h, const int32_t *host_max_steps, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    h->sh_valid = false;           // (`done` of a step already enqueued was decided under the old lengths: the shadow restarts at the next full reset)
    if (!host_max_steps) { h->kp.max_steps_env = nullptr; h->host_max_steps.clear(); return RANENV_OK; }
    for (int b = 0; b < h->cfg.batch; b++) if (host_max_steps[b] < 1) return fail(h, RANENV_E_INVALID, "env %d: max_steps must be >= 1", b);
    if (!h->d_max_steps && dev_alloc(h, &h->d_max_steps, (size_t)h->cfg.batch) != RANENV_OK) return RANENV_E_NOMEM;
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipMemcpyAsync(h->d_max_steps, host_max_steps, sizeof(int32_t) * (size_t)h->cfg.batch, hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->kp.max_steps_env = h->d_max_steps;
    h->host_max_steps.assign(host_max_steps, host_max_steps + h->cfg.batch);
    return RANENV_OK;
}

static int check_episode(ranenv_handle h, const ranenv_episode &e, const char *what, long long idx)
{
    if (e.scenario < 0 || e.scenario >= h->cfg.n_scenarios) return fail(h, RANENV_E_INVALID, "%s %lld: scenario %d outside pool of %d", what, idx, e.scenario, h->cfg.n_scenarios);
    if (e.se_len < 1 || e.se_offset < 0 || e.se_offset >= e.se_len || e.se_base < 0 || e.trf_len < 1 ||
        e.trf_offset < 0 || e.trf_offset >= e.trf_len || e.trf_base < 0)
        return fail(h, RANENV_E_INVALID, "%s %lld: need len >= 1, 0 <= offset < len, base >= 0", what, idx);
    if (h->se_tiles_n > 0 && e.se_base + e.se_len > h->se_tiles_n)
        return fail(h, RANENV_E_INVALID, "%s %lld: SE trace [%lld,+%d) exceeds the bound pool of %lld tiles", what, idx, (long long)e.se_base, e.se_len, (long long)h->se_tiles_n);
    if (h->kp.trf_pool && e.trf_base + e.trf_len > h->trf_rows_n)
        return fail(h, RANENV_E_INVALID, "%s %lld: traffic trace [%lld,+%d) exceeds the bound pool of %lld rows", what, idx, (long long)e.trf_base, e.trf_len, (long long)h->trf_rows_n);
    return RANENV_OK;
}

int ranenv_set_episode_table(ranenv_handle h, const ranenv_episode *host_table, int32_t first_episode, int32_t n_episodes, void *stream_)
{
    if (!h || !host_table) return fail(h, RANENV_E_INVALID, "null argument");
    if (n_episodes < 1 || first_episode < 0) return fail(h, RANENV_E_INVALID, "need n_episodes >= 1 and first_episode >= 0");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    for (int i = 0; i < n_episodes; i++) { const int rc = check_episode(h, host_table[i], "episode table entry", i); if (rc != RANENV_OK) return rc; }
    ranenv_episode *d = nullptr;
    if (dev_alloc(h, &d, (size_t)n_episodes) != RANENV_OK) return RANENV_E_NOMEM;      // (an older table stays allocated until destroy)
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipMemcpyAsync(d, host_table, sizeof(ranenv_episode) * (size_t)n_episodes, hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    h->d_ep_table = d; h->ep_table_first = first_episode; h->ep_table_n = n_episodes; h->idle_check_dirty = true;
    h->ar_on = false;                      // the rule is re-validated against the new table
    return RANENV_OK;
}

int ranenv_set_autoreset(ranenv_handle h, int32_t enable, int32_t initial_episode, int32_t max_episode, int32_t random_episodes,
                         uint64_t seed, const int32_t *host_episode_no, void *stream_)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (!enable) { h->ar_on = false; return RANENV_OK; }
    if (!h->d_ep_table) return fail(h, RANENV_E_STATE, "no episode table (ranenv_set_episode_table)");
    if (initial_episode < h->ep_table_first || max_episode <= initial_episode || max_episode > h->ep_table_first + h->ep_table_n)
        return fail(h, RANENV_E_INVALID, "episodes [%d,%d) must be a non-empty range inside the table [%d,%d)", initial_episode, max_episode,
                    h->ep_table_first, h->ep_table_first + h->ep_table_n);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    if (host_episode_no) {
        for (int b = 0; b < h->cfg.batch; b++)
            if (host_episode_no[b] < h->ep_table_first || host_episode_no[b] >= h->ep_table_first + h->ep_table_n)
                return fail(h, RANENV_E_INVALID, "env %d: episode number %d outside the table", b, host_episode_no[b]);
        HIP_TRY(h, hipMemcpyAsync(ST_episode_no(h->kp), host_episode_no, sizeof(int32_t) * (size_t)h->cfg.batch, hipMemcpyHostToDevice, stream));
        HIP_TRY(h, hipMemsetAsync(ST_reset_count(h->kp), 0, sizeof(int32_t) * (size_t)h->cfg.batch, stream));
        HIP_TRY(h, hipStreamSynchronize(stream));
    }
    h->ar_initial = initial_episode; h->ar_max = max_episode; h->ar_random = random_episodes ? 1 : 0; h->ar_seed = seed;
    h->ar_on = true;
    return RANENV_OK;
}

int ranenv_autoreset(ranenv_handle h, const uint8_t *dev_done, float *obs_inter, float *obs_intra,
                     float *term_obs_inter, float *term_obs_intra, float *term_obs_head, void *stream_)
{
    if (!h || !dev_done) return fail(h, RANENV_E_INVALID, "null argument");
    if (!h->ar_on) return fail(h, RANENV_E_STATE, "auto-reset is not configured (ranenv_set_autoreset)");
    int rc = check_ready(h, nullptr, nullptr, false);
    if (rc != RANENV_OK) return rc;
    if (!h->kp.se_pool && !(h->se_mode == RANENV_SE_GATHER && h->d_se_mean))
        return fail(h, RANENV_E_STATE, "auto-reset needs a bound SE pool (the reset observes the new episode's first tile)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    // no episode ended at the TTI enqueued last (the host follows the step counters, see ranenv::sh_steps): nothing to enqueue
    {
        const int due = shadow_due(h, 0, h->cfg.batch, dev_done, stream);
        if (due == 0) return RANENV_OK;
        if (due < 0) h->sh_valid = false;        // (the device decides by flags the host cannot follow: the shadow ends here)
        else shadow_reset_due(h, 0, h->cfg.batch);
    }
    const AdvanceArgs a = advance_args(h, dev_done, obs_inter, obs_intra, term_obs_inter, term_obs_intra, term_obs_head);
    h->pclass_maybe = true;                      // (scenarios of the restarted envs, if any: the advance kernel sets the device's flag)
    hipLaunchKernelGGL(ranenv_advance_kernel, dim3((unsigned)h->cfg.batch), dim3(64), 0, stream, a);
    KP kp = h->kp;
    kp.env_mask = h->d_ar_mask; kp.se_tiles = nullptr; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = nullptr;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = nullptr; kp.done = nullptr;   // the step's rewards stay
    kp.head_reward = nullptr;                    // ... those of the alternative heads too (head_obs gets the new episode's first observation)
    const hipError_t e = launch<MODE_RESET>(h, kp, stream);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "auto-reset launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_autoreset_part(ranenv_handle h, int32_t part, const uint8_t *dev_done, float *obs_inter, float *obs_intra,
                          float *term_obs_inter, float *term_obs_intra, float *term_obs_head, void *stream_)
{
    if (!h || !dev_done) return fail(h, RANENV_E_INVALID, "null argument");
    if (!h->ar_on) return fail(h, RANENV_E_STATE, "auto-reset is not configured (ranenv_set_autoreset)");
    if (part < 0 || part >= h->n_parts || h->part_lo.empty()) return fail(h, RANENV_E_INVALID, "partition %d outside [0,%d) (ranenv_set_partitions)", part, h->n_parts);
    int rc = check_ready(h, nullptr, nullptr, false);
    if (rc != RANENV_OK) return rc;
    if (!h->kp.se_pool && !(h->se_mode == RANENV_SE_GATHER && h->d_se_mean))
        return fail(h, RANENV_E_STATE, "auto-reset needs a bound SE pool (the reset observes the new episode's first tile)");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_, ps = h->part_stream[(size_t)part];
    const int e0 = h->part_lo[(size_t)part], n = h->part_lo[(size_t)part + 1] - e0;
    {
        const int due = stream_capturing(stream) ? -1 : shadow_due(h, e0, e0 + n, dev_done, ps);
        if (due == 0) {                            // (no episode of this range ended: see ranenv_autoreset)
            HIP_TRY(h, hipEventRecord(h->part_done[(size_t)part], ps));                    // ranenv_wait_part still finds its event
            return RANENV_OK;
        }
        if (due < 0) h->sh_valid = false;
        else shadow_reset_due(h, e0, e0 + n);
    }
    if (stream != ps) {
        HIP_TRY(h, hipEventRecord(h->part_in[(size_t)part], stream));
        HIP_TRY(h, hipStreamWaitEvent(ps, h->part_in[(size_t)part], 0));
    }
    AdvanceArgs a = advance_args(h, dev_done, obs_inter, obs_intra, term_obs_inter, term_obs_intra, term_obs_head);
    a.e0 = e0;
    h->pclass_maybe = true;
    hipLaunchKernelGGL(ranenv_advance_kernel, dim3((unsigned)n), dim3(64), 0, ps, a);
    KP kp = h->kp;
    kp.env_mask = h->d_ar_mask; kp.se_tiles = nullptr; kp.scores = nullptr; kp.intra = nullptr; kp.traffic_bits = nullptr;
    kp.dense = nullptr; kp.obs_inter = obs_inter; kp.obs_intra = obs_intra; kp.reward = nullptr; kp.done = nullptr;
    kp.head_reward = nullptr; kp.compact = 0;
    finalize_kp(h, kp);
    const hipError_t e = launch_range<MODE_RESET>(h, kp, e0, n, ps);
    if (e != hipSuccess) return fail(h, RANENV_E_HIP, "auto-reset launch (partition %d): %s", part, hipGetErrorString(e));
    HIP_TRY(h, hipEventRecord(h->part_done[(size_t)part], ps));
    return RANENV_OK;
}

int ranenv_get_views(ranenv_handle h, ranenv_views *out)
{
    if (!h || !out) return fail(h, RANENV_E_INVALID, "null argument");
    const KP &k = h->kp;
    out->pkt_incoming = ST_pkt_incoming(k); out->pkt_throughputs = ST_pkt_throughputs(k);
    out->pkt_effective_thr = ST_pkt_effective_thr(k); out->dropped_pkts = ST_dropped_pkts(k);
    out->queue_pkts = ST_queue_pkts(k); out->queue_age_sum = ST_queue_age_sum(k);
    out->rb_start = ST_rb_start(k); out->rb_count = ST_rb_count(k); out->se_mean = ST_se_mean(k);
    out->win_sent = ST_win_sent(k); out->win_dropped = ST_win_dropped(k);
    out->step_number = ST_step_no(k); out->hist_len = ST_hist_len(k);
    out->mask_inter = ST_mask_inter(k); out->mask_intra = ST_mask_intra(k); out->policy_scores = ST_policy_scores(k);
    out->episode_number = ST_episode_no(k);
    out->episodes = reinterpret_cast<int32_t *>(h->d_episodes);
    return RANENV_OK;
}

int ranenv_bind_head_outputs(ranenv_handle h, float *dev_obs_head, double *dev_reward_head)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if ((dev_obs_head || dev_reward_head) && (h->cfg.flags & RANENV_F_NO_RAW_OUTPUT))
        return fail(h, RANENV_E_STATE, "the heads read pkt_throughputs: not available with RANENV_F_NO_RAW_OUTPUT");
    h->kp.head_obs = dev_obs_head; h->kp.head_reward = dev_reward_head;
    return RANENV_OK;
}

int ranenv_set_slice_usecase(ranenv_handle h, int32_t first, int32_t count, const int32_t *usecase, void *stream_)
{
    if (!h || !usecase) return fail(h, RANENV_E_INVALID, "null argument");
    if (first < 0 || count < 1 || first + count > h->cfg.n_scenarios) return fail(h, RANENV_E_INVALID, "scenario rows [%d,%d) outside pool of %d", first, first + count, h->cfg.n_scenarios);
    const size_t n = (size_t)count * h->cfg.n_slices;
    for (size_t i = 0; i < n; i++) if (usecase[i] < 0 || usecase[i] > 3) return fail(h, RANENV_E_INVALID, "use-case bits must be in [0,3]");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(h, hipMemcpyAsync(TB_slice_usecase(h->kp) + (size_t)first * h->cfg.n_slices, usecase, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(h, hipStreamSynchronize(stream));
    return RANENV_OK;
}

int ranenv_se_from_power(const double *dev_power, float *dev_se, int64_t n_elems, double tx_power_per_rb,
                         double noise_power, void *stream)
{
    if (!dev_power || !dev_se) return fail(nullptr, RANENV_E_INVALID, "null argument");
    if (n_elems < 0) return fail(nullptr, RANENV_E_INVALID, "negative element count");
    if (!(noise_power > 0.0)) return fail(nullptr, RANENV_E_INVALID, "noise_power must be positive");
    if (n_elems == 0) return RANENV_OK;
    long long blocks = (n_elems + 511) / 512;
    if (blocks > 256 * 64) blocks = 256 * 64;          // grid-stride beyond 64 workgroups per CU
    hipLaunchKernelGGL(ranenv_se_from_power_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       dev_power, dev_se, (long long)n_elems, tx_power_per_rb, noise_power);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, RANENV_E_HIP, "se_from_power launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_launch_info(ranenv_handle h, int32_t *grid, int32_t *block, int32_t *lds_bytes)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (grid) *grid = h->cfg.batch;
    if (block) *block = h->nt;
    if (lds_bytes) *lds_bytes = (int32_t)(h->np == 8 ? sizeof(SharedCore<8>) : h->np == 10 ? sizeof(SharedCore<10>) : sizeof(SharedCore<16>));
    return RANENV_OK;
}

}  // extern "C"
