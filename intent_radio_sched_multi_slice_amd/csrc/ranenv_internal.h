// ranenv_internal.h -- what the translation units of libranenv_hip.so share: the kernel-argument block (KP) and the structures it
// points to, the slab accessors, and the LAUNCH TABLE -- the functions through which the host side of the ABI (ranenv_host.cpp)
// reaches the kernels, which live in ranenv_step.hip (one object per row width NP, -DRANENV_NP=8 / 10 / 16: they compile in parallel)
// and ranenv_aux.hip (the small kernels: class sort, sidecars, re-tiling, ingest, heads, episode advance, traffic examination).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstddef>
#include <cstdint>

#include "ranenv.h"

#ifndef RANENV_DIAG
#define RANENV_DIAG 0   /* diagnostic builds only (tools/build_variants.sh; see the table in DESIGN.md section 8): 3 / 4 / 7 / 11 skip the UE step / the
                           observation tail / the allocation / the masked half of the stream (tools/valu_phases.sh counts instructions by difference),
                           9 / 12 stamp s_memtime at the phase boundaries of the one-TTI / the persistent launches (tools/stamps.py, persist_phases.py) */
#endif

namespace ranenv_dev {

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
    int32_t *u4;         // [11][B][U] 4-byte per-UE fields (the ST_* accessors below name them)
    int64_t *u8;         // [4][B][U]  8-byte per-UE fields: queue_age_sum, win_sent, win_dropped (int64), se_mean (double)
    int32_t *b4;         // [9][B]     per-env counters
    int2 *age_ring; int32_t *ring_sent; int32_t *ring_drop;
    int8_t *mask_inter, *mask_intra; double *policy_scores;
};
enum { N_U4 = 11, N_U8 = 4, N_B4 = 9, N_TUE = 12 };
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
#define ST_last_push(p) ((p).st.u4 + (size_t)(10) * (size_t)(p).BU)
#define ST_queue_age_sum(p) ((int64_t *)((p).st.u8 + (size_t)(0) * (size_t)(p).BU))
#define ST_win_sent(p) ((int64_t *)((p).st.u8 + (size_t)(1) * (size_t)(p).BU))
#define ST_win_dropped(p) ((int64_t *)((p).st.u8 + (size_t)(2) * (size_t)(p).BU))
#define ST_se_mean(p) ((double *)((p).st.u8 + (size_t)(3) * (size_t)(p).BU))
#define ST_hist_len(p) ((p).st.b4 + (size_t)(0) * (size_t)(p).B)
#define ST_n_push(p) ((p).st.b4 + (size_t)(1) * (size_t)(p).B)
#define ST_step_no(p) ((p).st.b4 + (size_t)(2) * (size_t)(p).B)
#define ST_se_pos(p) ((p).st.b4 + (size_t)(3) * (size_t)(p).B)
#define ST_trf_pos(p) ((p).st.b4 + (size_t)(4) * (size_t)(p).B)
#define ST_episode_no(p) ((p).st.b4 + (size_t)(5) * (size_t)(p).B)
#define ST_reset_count(p) ((p).st.b4 + (size_t)(6) * (size_t)(p).B)
#define ST_push_total(p) ((p).st.b4 + (size_t)(7) * (size_t)(p).B)
#define ST_clear_mark(p) ((p).st.b4 + (size_t)(8) * (size_t)(p).B)
#define ST_age_ring(p) ((p).st.age_ring)
#define ST_ring_sent(p) ((p).st.ring_sent)
#define ST_ring_drop(p) ((p).st.ring_drop)
#define ST_mask_inter(p) ((p).st.mask_inter)
#define ST_mask_intra(p) ((p).st.mask_intra)
#define ST_policy_scores(p) ((p).st.policy_scores)
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
    int compact;     // step only the UEs that are in a slice (lanes are ordered slice members first): waves without one leave
                     // at once.  Set by the host when it is exact: UEs outside every slice get no traffic (see idle_traffic_ok)
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

constexpr int WAVE = 64;
constexpr int GRP = 16;   // lanes per (env) group in alloc1/obs and per (env, slice) group in alloc2

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

enum { PERSIST_ENV_BITS = 20 };           // persistent rollout: queue item = env | TTIs done << 20
constexpr int CORE_NT = GRP * GRP;   // 256 = largest U = threads of the widest step-kernel block

// ---------------------------------------------------------------------------------------------
// Launch table
// ---------------------------------------------------------------------------------------------
// Builds of the step kernel (all from step_body, ranenv_step_body.hpp)
enum StepBuild {
    SB_LEAN = 0,        // ranenv_core_kernel<MODE, NP, MANY>: batches that fill the chip (5 waves per SIMD)
    SB_SMALL,           // ranenv_core_kernel_small: <= 8 workgroups per CU (4 waves per SIMD, deeper SE queue)
    SB_GATHER,          // ranenv_core_kernel_gather: SE gather mode
    SB_TINY1,           // ranenv_core_kernel_tiny1: one-TTI step launches of a batch at <= 2 waves per SIMD (whole row in flight)
    SB_MIXED,           // ranenv_core_kernel_mixed<NP, MANY, GATHER>: one block per wide env + one per two narrow envs
    SB_PACKED,          // ranenv_core_kernel_packed<8, MANY, GATHER>: two envs per wave (NP = 8 only)
    SB_PERSIST,         // ranenv_persist_kernel<GATHER, NP>: work-queue rollout
    SB_PERSIST_TINY,    // ranenv_persist_kernel_tiny<NP>: the same at <= 2 waves per SIMD, streaming only
};
struct StepLaunch {
    int np;             // row width of the build: 8, 10, 16
    int build;          // StepBuild
    int mode;           // MODE_STEP / MODE_DENSE / MODE_RESET, | MODE_PE for the per-element builds (lean / gather only)
    bool many;          // a launch of several TTIs (MODE_STEP only)
    bool gather;        // SB_MIXED / SB_PACKED / SB_PERSIST: the SE gather build
};
// ranenv_step.hip, one object per row width.  hipErrorInvalidDeviceFunction: that combination is not built.
// ev0 / ev1 non-null: an extended launch that records the dispatch's own start / stop timestamps.
hipError_t launch_step_np8(const StepLaunch &, dim3 grid, dim3 block, hipStream_t, hipEvent_t ev0, hipEvent_t ev1, const KP &);
hipError_t launch_step_np10(const StepLaunch &, dim3 grid, dim3 block, hipStream_t, hipEvent_t ev0, hipEvent_t ev1, const KP &);
hipError_t launch_step_np16(const StepLaunch &, dim3 grid, dim3 block, hipStream_t, hipEvent_t ev0, hipEvent_t ev1, const KP &);
const void *step_kernel_ptr_np8(const StepLaunch &);      // (for occupancy / attribute queries)
const void *step_kernel_ptr_np10(const StepLaunch &);
const void *step_kernel_ptr_np16(const StepLaunch &);
size_t shared_core_bytes_np8();
size_t shared_core_bytes_np10();
size_t shared_core_bytes_np16();
inline hipError_t launch_step(const StepLaunch &l, dim3 grid, dim3 block, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1, const KP &kp)
{
    switch (l.np) {
    case 8: return launch_step_np8(l, grid, block, s, ev0, ev1, kp);
    case 10: return launch_step_np10(l, grid, block, s, ev0, ev1, kp);
    default: return launch_step_np16(l, grid, block, s, ev0, ev1, kp);
    }
}
inline const void *step_kernel_ptr(const StepLaunch &l)
{
    return l.np == 8 ? step_kernel_ptr_np8(l) : (l.np == 10 ? step_kernel_ptr_np10(l) : step_kernel_ptr_np16(l));
}
inline size_t shared_core_bytes(int np) { return np == 8 ? shared_core_bytes_np8() : (np == 10 ? shared_core_bytes_np10() : shared_core_bytes_np16()); }

// ranenv_aux.hip: the small kernels (enqueue only; errors through hipGetLastError)
void launch_classify(hipStream_t, const ranenv_episode *eps, const int32_t *members, int B, int n_class, int one_class, int32_t *list,
                     int32_t *count, int *flag, int force);
void launch_se_sidecar(hipStream_t, unsigned n_tiles, unsigned block, const float *pool, long long stride, long long tile0, int U, int R, int Rp,
                       int quad, double *mean, float *um);
void launch_se_sidecar_from_power(hipStream_t, unsigned n_tiles, unsigned block, const double *power, long long tile0, int U, int R, int Rp,
                                  double tx, double noise, double *mean, float *um);
void launch_se_retile_quad(hipStream_t, unsigned blocks, const float *src, float *dst, long long n_quads, int U, int R);
void launch_se_from_power(hipStream_t, unsigned blocks, const double *power, float *se, long long n, double tx_per_rb, double noise);
void launch_ddiv_selftest(hipStream_t, const double *a, const double *b, double *fast, double *ieee, long long n);
void launch_head(hipStream_t, dim3 grid, dim3 block, const KP &);
void launch_advance(hipStream_t, unsigned n_envs, const AdvanceArgs &);
void launch_idle_traffic(hipStream_t, unsigned n_eps, const ranenv_episode *eps, const int32_t *pool, int U, const int32_t *lane_slice,
                         const int32_t *lane_ue, int *violations);

}  // namespace ranenv_dev
