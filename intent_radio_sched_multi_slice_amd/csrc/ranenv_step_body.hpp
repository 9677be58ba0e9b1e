// ranenv_step_body.hpp -- the step kernel of the MI355X (gfx950) implementation of the C ABI in include/ranenv.h.
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
#pragma once
#include "ranenv_numeric.hpp"

namespace {

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
    int acc[NP];                      // per slice: sum of the intra-slice floors (LDS atomic adds by the slice's UEs; zero between two allocations)
    unsigned msk[NP][2];              // per slice, one bit per UE position, set with LDS atomic ORs by the slice's UEs and read as ONE word:
                                      // [0] the UE's buffer is not empty, [1] its PF / MT value is non-zero.  All zero between two allocations
    int rbs[GRP], off[GRP];       // RBs of each slice and its first RB
    // this TTI's observation rows, staged here and written out by wave 0 as whole lines: written one float per lane and
    // instruction they were ~170 partial-line store requests per env (profiles/r02_pmc_memsys.txt)
    float ob_inter[NP * 10];
    float ob_intra[NP * (2 * NP + 9)];
};

// Workgroup barrier for exchanges through LDS only: waits for this wave's LDS operations, not for its global loads and
// stores (the UE role's ~17 state stores need not be acknowledged before the obs role starts, and the SE loads requested ahead
// of the allocation need not land before its first exchange).  Nothing in the step kernel passes data between threads through
// global memory.  Where global memory IS handed over -- to the next TTI of the same workgroup without a warm entry, or to another
// workgroup (persistent rollout) -- full_sync() below is used: __syncthreads() alone is NOT enough, the compiler's
// workgroup-scope fence waits for lgkmcnt only (no vmcnt(0) outside tgsplit mode: ADVICE r4, seen in the shipped ISA).
// `narrow`: this wave is a workgroup of its own inside a two-wave block (ranenv_core_kernel_mixed: two one-wave envs per block):
// its exchanges through LDS are between its own lanes, so it waits for its LDS operations and must NOT take part in a block
// barrier -- the block's other wave steps another env, or has left.
DEVFN void wg_sync(const bool narrow = false)
{
    if (narrow) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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
        unsigned off = (unsigned)row_bytes + lane_bytes;
        asm volatile("" : "+v"(off));              // (the 64-bit address is formed at every use, not carried from the load to the store)
        return *(T *)((char *)array + off);
    }
}

// Cache hints (round 5; same-box A/B in profiles/r05_ab_log.txt).  The SE tile is read once per TTI and never again: its loads carry the
// non-temporal bit, so that 292 MB of tiles per TTI do not push the per-UE state -- re-read at the very next TTI -- out of the caches
// (streaming rollout -2...-3 %).  In the SE gather builds, which stream no tile, the same goes for what the kernel WRITES and will not read
// again soon (observation rows, raw outputs, age-list and window-ring entries; gather -3.7 %; no gain for the streaming builds, so they
// keep plain stores) and for the sidecar reads.
template <bool NT, typename T> DEVFN void nt_store(T &dst, const T v)
{
    if constexpr (NT && RANENV_CACHE_HINTS != 0) {
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
                if (__builtin_amdgcn_ballot_w64(my_v < 0.0) == 0) {       // (scores inside [-1, 1]: no negative value; see the intra-slice loop)
#pragma unroll
                    for (int j = 0; j < NP; j++) { const double xj = xs[3][j]; rank += (j > s1 ? xj >= my_v : xj > my_v) ? 1 : 0; }
                } else {
#pragma unroll
                    for (int j = 0; j < NP; j++) { const double xj = xs[3][j]; rank += (xj != 0.0 && (xj > my_v || (xj == my_v && j > s1))) ? 1 : 0; }
                }
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
                // (np.max over the slice's n entries: the row is zero beyond them and no entry is negative, so the maximum over all NP is the same)
                max_avail = r0[0];
#pragma unroll
                for (int k = 1; k < NP; k++) max_avail = dmax(max_avail, r0[k]);
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
        // the slice's sum of the floors: one LDS atomic add per UE and one read (was: a row of ten floors, read back by every UE)
        if (have && prop != 0) atomicAdd(&sh.acc[sl], prop);
        wg_sync(narrow);
    }
    int count = 0;
    if (use_round) {
        const int adj = n_rbs - sh.acc[sl];
        count = prop;
        if (nzv && adj > 0) {
            // how many of the slice's non-zero values come before this one in np.argsort(values)[::-1] (stable: among equals the higher
            // position first).  Values are positive unless the caller's SE tiles are negative: then a value that is greater or equal is
            // non-zero by itself and the test for zero drops out (one wave-uniform look decides which loop runs)
            int rank = 0;
            if (__builtin_amdgcn_ballot_w64(my_val < 0.0) == 0) {
#pragma unroll
                for (int k = 0; k < NP; k++) { const double xk = r2[k]; rank += (k > pos ? xk >= my_val : xk > my_val) ? 1 : 0; }
            } else {
#pragma unroll
                for (int k = 0; k < NP; k++) { const double xk = r2[k]; rank += (xk != 0.0 && (xk > my_val || (xk == my_val && k > pos))) ? 1 : 0; }
            }
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
    if (have) sh.cnt[sl][pos] = count;
    wg_sync(narrow);
    if (have && pos == 0) { sh.msk[sl][0] = 0u; sh.msk[sl][1] = 0u; sh.acc[sl] = 0; }   // (masks and sum were read in front of that barrier; the next ORs / adds are barriers away)
    int before = 0;                                                                // :464-478 contiguous ranges
#pragma unroll
    for (int k = 0; k < NP; k++) before += (int)((below >> k) & 1u) * sh.cnt[sl][k];   // (bit k of `below`: k < pos)
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
    constexpr bool CARRY = PERSIST && MODE == MODE_STEP && PACK == 1 && !GATHER && NQ >= 8;
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
    // The kernel's argument block, read in place: a field that only a late role needs is fetched there (one scalar load)
    // instead of sitting in -- or being spilled from -- SGPRs since kernel entry.  (The laundering keeps the compiler from
    // merging these loads with the by-value copy it loads up front.)
    typedef const __attribute__((address_space(4))) KP *kp_const_t;
    kp_const_t kc = (kp_const_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kc));
#define COLD(f) (kc->f)
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
    // (a packed wave holds the episode descriptor per lane: the traffic row's byte offset -- below 4 GB there, pack_fits_32 -- as ONE
    // register from here to the load behind the stream, instead of the 64-bit base and the position)
    unsigned trf_row32 = 0;
    if constexpr (PACK == 2) {
        trf_row32 = (unsigned)(((size_t)ep.trf_base + (size_t)trf_pos) * (size_t)p.U * 4);
        asm volatile("" : "+v"(trf_row32));          // (formed here, not re-derived at the use from operands kept alive for it)
    }
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
                                     : (double)row_at<PACK>(p.trf_pool, PACK == 2 ? (size_t)trf_row32 : ((size_t)ep.trf_base + (size_t)trf_pos) * U * 4, u4);
    };
    // When the rest of the UE's state is requested: behind the stream by default (~20 registers fewer while the tile streams: with 8
    // loads in flight per lane the kernel fits 96 VGPRs = 5 waves per SIMD without spills, profiles/r02_ab_log.txt), but at
    // entry in the build that has registers to spare (the whole-row queue of a batch at <= 2 waves per SIMD): there its round
    // trip -- 1.5-2 us of every chain when exposed -- runs under the allocation and the stream.
    constexpr bool STATE_AT_ENTRY = !GATHER && NQ >= 8;
    if constexpr (STATE_AT_ENTRY) rest_of_state();
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
#if RANENV_DIAG == 7      /* ablation: no allocation -- the UE's range of the previous TTI stands in */
    if (MODE == MODE_STEP) { rb_start = UE4(rb_start); rb_count = UE4(rb_count); }
#endif
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
        if (tid < S) sh.acc[tid] = 0;
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
    // (behind a condition the optimiser cannot see through -- always true: without it the scheduler moves address arithmetic of the later
    // roles up across the allocation, their registers are alive through it, and the builds at the 96-register limit spill)
    int do_alloc = 1;
    asm volatile("" : "+s"(do_alloc));
    if (MODE == MODE_STEP && RANENV_DIAG != 7 && do_alloc != 0)
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
    auto hook = []() {};
    if constexpr (GATHER) {
        if (MODE == MODE_STEP) {
            bool spread = false;
            if constexpr (PACK == 1) {
                // (this wave's 64 item descriptors live in the cross-slice rows: free between the allocation and the observation tail, and every
                // reader of those rows writes what it reads first)
                unsigned short *desc = reinterpret_cast<unsigned short *>(&sh.xr[0][0]) + (narrow ? 0 : (tid >> 6)) * 64;
                spread = gather_part_spread<PE>(tile, U * p.se_rp * 4, u * p.se_rp * 4, R, (unsigned)rb_start, (unsigned)rb_count, PE ? COLD(bw_per_rb) : 1.0, desc, my_part);
            }
            if (!spread) my_part = gather_part<PACK, GDEPTH, PE>(tile, U * p.se_rp * 4, u * p.se_rp * 4, R, (unsigned)rb_start, (unsigned)rb_count, PE ? COLD(bw_per_rb) : 1.0);
        }
    } else if constexpr (MODE == MODE_STEP) {
        const unsigned us1 = (unsigned)rb_start, uc1 = (unsigned)rb_count;
        row_sums<PE>(se1, R, [=](int r) { return ((unsigned)r - us1) < uc1; }, my_full, my_part, hook, PE ? COLD(bw_per_rb) : 1.0);
    } else if constexpr (MODE == MODE_DENSE) {
        const uint8_t *mrow = p.dense + ((size_t)e * U + u) * R;
        row_sums<PE>(se1, R, [=](int r) { return mrow[r] != 0; }, my_full, my_part, hook, PE ? COLD(bw_per_rb) : 1.0);
    } else {
        row_sums(se1, R, [](int) { return false; }, my_full, my_part, hook);
    }
    if constexpr (!STATE_AT_ENTRY) rest_of_state();        // after the stream: its latency is exposed, its registers were free for the queue
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
#if RANENV_DIAG == 3      /* ablation: no UE step (the condition is never true, and the compiler cannot know) */
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
            // (explicit traffic is the caller's double -- inf, huge and denormal values included --: the plain division, which saturates
            // as before; pooled / generated traffic is an int32 count of bits)
            pkt_in = p.traffic_bits ? (int)(traffic / psz) : (int)ddiv(traffic, psz);
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
                float *oa = sh.ob_intra + slc * W;
                oa[9 + ue_pos] = (float)occ_new;
                oa[9 + Us + ue_pos] = (float)ddiv(se_mean_new, COLD(norm_se));
            }
        }
    }
    if (MODE != MODE_RESET && COLD(acc) != nullptr) {
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
#if RANENV_DIAG == 4      /* ablation: no observation tail, but the per-env counters move on (tiles keep changing) */
    if (my_full >= -1.0) {
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
            float *oi = sh.ob_inter + spos * 10;
            oi[0] = o0; oi[1] = o1; oi[2] = o2; oi[3] = a0; oi[4] = a1; oi[5] = a2;
            oi[6] = (float)priority; oi[7] = tr; oi[8] = nu; oi[9] = (float)ddiv(se_slice, COLD(norm_se));
        }
        if (COLD(obs_intra)) {
            float *oa = sh.ob_intra + s * W;
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
    if (MODE != MODE_RESET && COLD(acc) != nullptr) {
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
        if (COLD(acc) != nullptr) {
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
    RANENV_STAMP(7);

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
            // XCD k = b & 7 steps a contiguous part of each class's list (see step_loop's b_perm below): among the wide blocks the k-th eighth;
            // among the narrow blocks -- they start at workgroup id n_wide, i.e. at XCD n_wide & 7 -- the blocks of XCD k, counted in XCD order
            const int k8 = b & 7;
            if (b < n_wide) {
                const int q_ = n_wide >> 3, r_ = n_wide & 7;
                e_mix = p.p_list[k8 * q_ + (k8 < r_ ? k8 : r_) + (b >> 3)];
            } else {
                narrow = true;
                const int nbn = (n_narrow + 1) >> 1, o = n_wide & 7, c = b - n_wide + o;      // c & 7 == k8
                if (b - n_wide >= nbn) return;
                // narrow blocks on XCD x: those c in [o, o + nbn) with c & 7 == x
                auto cnt = [&](int x) { const int hi = o + nbn - 1; return hi < x ? 0 : ((hi - x) >> 3) + 1 - (x < o ? 1 : 0); };
                int start = 0;
                for (int x = 0; x < 8; x++) start += x < k8 ? cnt(x) : 0;
                const int pos = start + (c >> 3) - (k8 < o ? 1 : 0);
                const int idx = 2 * pos + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                if (idx >= n_narrow) return;
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
            // consecutive workgroup ids go round the 8 XCDs (MI355X_MICROARCH.md, Workgroup dispatch): workgroup b = 8 i + k steps the i-th env
            // of the k-th CONTIGUOUS eighth of the launch's envs, so that an XCD touches one eighth of every per-env array (see persist_try_fresh)
            const int nb_ = (int)gridDim.x, xk_ = (int)blockIdx.x & 7, q_ = nb_ >> 3, r_ = nb_ & 7;
            const int b_perm = xk_ * q_ + (xk_ < r_ ? xk_ : r_) + ((int)blockIdx.x >> 3);
            if (step_body<MODE_X, NQ, GATHER, NP, false, PACK, MIX>(*kc, cy, warm, MIX ? e_mix : kc->e0 + b_perm * PACK, nullptr, false, false, narrow)) return;
            if (k + 1 < n) {
                // The next TTI takes over in registers what it would otherwise load back (StepCarry): only LDS has to be handed over
                // between the waves.
                warm = MANY;
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

// (statistics of the queues, one fire-and-forget add per event from lane 0: ranenv_get_option "persist_stat_*")
#define PSTAT(k) ((void)__hip_atomic_fetch_add(&p.p_ctl->stat[xcc][k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
#define XCC_OF(pl) const int xcc = __builtin_amdgcn_readfirstlane((pl).xcc)      /* (one lane runs these functions: a scalar, not a register pair per use) */
DEVFN unsigned pq_ld(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVFN int pq_ldi(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// lane 0 only.  -> item, or PERSIST_NONE
template <typename P> DEVFN int persist_try_fresh(const P &p, PersistLocal &pl)
{
    PersistCtl *c = p.p_ctl;
    XCC_OF(pl);
    for (int s8 = 0; s8 < 8 && pl.fresh_mask != 0; s8++) {
        const int x = (xcc + s8) & 7;
        if (!(pl.fresh_mask >> x & 1)) continue;
        const unsigned j = __hip_atomic_fetch_add(&c->fresh[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (shard x = a CONTIGUOUS eighth of the class's list, which is in env order: an XCD's workgroups then touch one eighth of every
        // per-env array, and its L2 keeps more of it -- HBM traffic per TTI -2.8 % streaming, -8 % gather, -6.5 % at the native size; against
        // the interleaved shards x + 8 j: -1 % per TTI in every schedule, profiles/r06_ab_log.txt)
        const long long per = ((long long)p.p_count + 7) >> 3, i = (long long)x * per + (long long)j;
        if ((long long)j < per && i < (long long)p.p_count) { PSTAT(3); return p.p_list[i]; }  // (TTIs done: 0)
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
    XCC_OF(pl);
    if (pq_ldi(&c->q[xcc].avail) <= 0) return PERSIST_NONE;
    const int a = __hip_atomic_fetch_add(&c->q[xcc].avail, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a <= 0) { __hip_atomic_fetch_add(&c->q[xcc].avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return PERSIST_NONE; }
    const unsigned h = __hip_atomic_fetch_add(&c->q[xcc].head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long *slot = p.p_slots + (size_t)xcc * (size_t)p.p_cap + (h & (unsigned)(p.p_cap - 1));
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
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // buffer_inv sc1: this CU's L1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return (int)(unsigned)ent;
}

template <typename P> DEVFN void persist_push(const P &p, const PersistLocal &pl, int item)
{
    PersistCtl *c = p.p_ctl;
    XCC_OF(pl);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (every wave waited for its stores in front of the chunk-end barrier: full_sync)
    const unsigned idx = __hip_atomic_fetch_add(&c->q[xcc].tail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long *slot = p.p_slots + (size_t)xcc * (size_t)p.p_cap + (idx & (unsigned)(p.p_cap - 1));
    __hip_atomic_store(slot, ((unsigned long long)(idx + 1u) << 32) | (unsigned)item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->q[xcc].avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    XCC_OF(pl);
    if (done >= n_tti) return 0;
    if (pq_ldi(&c->abort) != 0) {        // a wait gave up somewhere in this class: the env is dropped here, and the host is told (again)
        if (p.p_err) __hip_atomic_store(p.p_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return 0;
    }
    const int fresh = persist_try_fresh(p, pl);
    if (fresh == PERSIST_NONE) {
        if (pq_ldi(&c->q[xcc].avail) <= 0) { PSTAT(0); return 1; }    // nobody is waiting
        const unsigned h = pq_ld(&c->q[xcc].head);
        // Somebody is -- but a swap only helps when the env at the head of the queue is BEHIND this one: with every
        // finisher swapping, every chunk of every env would go through the queue; this way a round of chunks costs one swap per
        // waiting env (a racy look at the head entry: a heuristic, whichever way it goes the state stays consistent).
        const unsigned long long ent = __hip_atomic_load(p.p_slots + (size_t)xcc * (size_t)p.p_cap + (h & (unsigned)(p.p_cap - 1)),
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
                const bool ahead = done + k + 1 < n_tti;
                (void)step_body<MODE_STEP, NQ, GATHER, NP, true>(*kc, cy, warm, e, &seq, se_ready, ahead);
                se_ready = ahead;
                if (k + 1 < n) { warm = true; wg_sync(); }
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
            warm = true;                             // `cy` is what the chunk's last TTI left
        }
    }
}

#ifndef RANENV_CLASS_PRIO
#define RANENV_CLASS_PRIO 1         /* s_setprio 1 for the waves of the workgroup class that finishes a persistent rollout last (0: none).  Gather mode: the one-wave
                                       class (K = 20 -2.4 %, round 5); streaming: the two-wave class -- since round 6's trims it ends a 20-TTI rollout ~45 us
                                       behind the one-wave class (same-box sextuples, K = 20: -1.8 % against priority on the one-wave class, -0.5 % against none;
                                       the same setting costs gather +1.5 %: profiles/r06_ab_log.txt) */
#endif
// waves per SIMD the builds are compiled for: 5 (96 VGPRs) everywhere except the 16-wide row builds of multi-TTI launches (4: they keep
// 16-entry rows of doubles alive in the allocation and would spill at 96), the small-batch and packed builds (4) and the whole-row builds (2)
#define RANENV_PERSIST_WPE ((NP == 16) ? 4 : 5)
template <bool GATHER, int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(RANENV_PERSIST_WPE, RANENV_PERSIST_WPE))) ranenv_persist_kernel(const KP p)
{
    (void)p;                                         // (read in place, like step_loop)
#if RANENV_DIAG == 0 || RANENV_DIAG == 12            /* (the other diagnostic / ablation builds run the launch-per-chunk rollout only) */
#if RANENV_CLASS_PRIO
    if ((blockDim.x == WAVE) == GATHER) __builtin_amdgcn_s_setprio(RANENV_CLASS_PRIO);
#endif
    persist_loop<GATHER ? 1 : SE_DEPTH_LEAN, GATHER, NP>();
#endif
}

// The same for a batch that leaves the chip at <= 2 waves per SIMD (BASELINE configs[1], B 1024): 256 VGPRs are there for the
// taking, so a lane keeps its whole SE row in flight (16 groups of 8 loads): the stream phase of a workgroup's chain is one
// memory latency instead of four (profiles/r04_ab_log.txt: 23.4 against 26.6 us per TTI).  Streaming only.
template <int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) ranenv_persist_kernel_tiny(const KP p)
{
    (void)p;
#if RANENV_DIAG == 0 || RANENV_DIAG == 12
    persist_loop<SE_DEPTH_TINY, false, NP>();
#endif
}

// ... and as an ordinary one-TTI launch for the same batches (env.step() of a small batch: what an SB3 / RLlib trainer with a few hundred envs
// calls): the whole row requested at entry together with all of the UE's state -- the step is one chain of latencies, and this removes
// three of the stream's four.
template <int NP>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) ranenv_core_kernel_tiny1(const KP p)
{
    step_loop<MODE_STEP, SE_DEPTH_TINY, false, NP, false>(p);
}
// Two builds of the step kernel.  A batch that fills the machine (more workgroups than 8 per CU) runs the lean one:
// 96 VGPRs = 5 waves per SIMD = 10 workgroups per CU, 16 SE loads in flight per lane (8 until the build stopped hoisting
// at machine level, which freed the registers for the second group) -- occupancy hides more latency than a still deeper
// queue (measured, profiles/r02_ab_log.txt, r03_ab_log.txt).  A small batch is resident at once whatever the register
// count, so it takes the build with 128 VGPRs and 32 loads in flight.
#define RANENV_CORE_ATTR __attribute__((amdgpu_waves_per_eu(5, 5)))
// (MANY: a launch of several TTIs, ranenv_rollout only, runs a build of its own -- the one-TTI build stays free of the
// warm entry's second path through the role, which costs it 1-2 %)
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) RANENV_CORE_ATTR ranenv_core_kernel(const KP p)
{
    step_loop<MODE, ((MODE & 3) == MODE_DENSE || NP == 16) ? 1 : SE_DEPTH_LEAN, false, NP, MANY>(p);
}
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(4, 4))) ranenv_core_kernel_small(const KP p)
{
    step_loop<MODE, SE_DEPTH_SMALL, false, NP, MANY>(p);
}
// The SE gather build (ranenv_set_se_mode): no tile stream, so no queue registers; one build for every batch size.
// (the 16-wide row build keeps 16-entry rows of doubles alive in the allocation and does not fit 96 registers with the per-TTI loop
// around it -- 2...10 spilled VGPRs -- so it is built for 4 waves per SIMD: S or Us above 10 is not a BASELINE size, and a spill in
// every TTI costs more than the fifth wave gains, profiles/r03_ab_log.txt.  No kernel of the library has scratch:
// tests/test_kernel_resources.py reads the shipped code object's metadata)
#define RANENV_WPE_NP(w) ((NP == 16 && MANY) ? 4 : (w))
template <int MODE, int NP, bool MANY>
__global__ void __launch_bounds__(CORE_NT) __attribute__((amdgpu_waves_per_eu(RANENV_WPE_NP(5), RANENV_WPE_NP(5))))
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
    step_loop<MODE_STEP, GATHER ? 1 : SE_DEPTH_LEAN, GATHER, NP, MANY, 1, true>(p);
}

// Packed waves (round 4): envs of at most 32 UEs and 8 slices -- the reference's own size, S 5 / U 25 -- leave 39 of a wave's 64 lanes
// idle, and the chip holds as many waves as it holds; one wave steps TWO envs (lanes 0-31 / 32-63, step_body's PACK = 2): half as many
// waves per env-step.  What is wave-uniform in the other builds is per-lane here (more registers: 4 waves per SIMD), so it is a build
// of its own, for step launches of an even number of envs; reset and dense launches keep one env per wave (same state layout).
template <int NP, bool MANY, bool GATHER>
__global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) ranenv_core_kernel_packed(const KP p)
{
    step_loop<MODE_STEP, GATHER ? (MANY ? 1 : 0) : SE_DEPTH_PACKED, GATHER, NP, MANY, 2>(p);        // (one-TTI gather build: gather depth 1, it has no register to spare)
}

}  // namespace
