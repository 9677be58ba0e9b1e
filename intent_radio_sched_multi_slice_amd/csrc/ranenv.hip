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
//   per-UE state   [B][U]  queue_pkts i32, queue_age_sum i64, front i32, front_rem i32,
//                          win_sent i64, win_dropped i64, se_mean f64
//   age ring       int32   [B][L][U]      packets admitted at TTI (t mod L); the queue is FIFO,
//                                          so (front, front_rem, ring) describe exactly the age
//                                          histogram Buffer keeps (oracle/ranenv_oracle.c) while
//                                          a step touches only the inserted / expired / drained
//                                          entries.
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
    int32_t *queue_pkts; int64_t *queue_age_sum; int32_t *front; int32_t *front_rem;
    int64_t *win_sent; int64_t *win_dropped; double *se_mean;
    int32_t *age_ring; int32_t *ring_sent; int32_t *ring_drop;
    int32_t *hist_len; int32_t *n_push; int32_t *step_no;
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
struct LdsLayout {
    int d_occ, d_sem, d_hmean, d_semn, d_occn, d_drift, d_slice, d_scores, d_tmp, d_slvals, d_slflags,
        d_par, d_slf, i_maxpkts, i_pktsize, i_start, i_count, i_sl, i_slues, i_par, i_rbs, i_off, i_cnt,
        i_nzi, i_misc, f_obs_inter, f_obs_intra, total;
};

__host__ __device__ inline LdsLayout make_layout(int S, int U, int Us)
{
    LdsLayout l;
    int o = 0;
    auto take = [&](int bytes) { int r = o; o += (bytes + 7) & ~7; return r; };
    l.d_occ = take(8 * U);  l.d_sem = take(8 * U);  l.d_hmean = take(8 * U);
    l.d_semn = take(8 * U); l.d_occn = take(8 * U);
    l.d_drift = take(8 * S * Us * 3);
    l.d_slice = take(8 * S * 4 * Us);
    l.d_scores = take(8 * S); l.d_tmp = take(8 * 5 * S);
    l.d_slvals = take(8 * 3 * S); l.d_slflags = take(8 * 3 * S);
    l.d_par = take(8 * 3 * S); l.d_slf = take(8 * 2 * S);
    l.i_maxpkts = take(4 * U); l.i_pktsize = take(4 * U); l.i_start = take(4 * U); l.i_count = take(4 * U);
    l.i_sl = take(4 * 8 * S); l.i_slues = take(4 * S * Us); l.i_par = take(4 * 6 * S);
    l.i_rbs = take(4 * S); l.i_off = take(4 * S); l.i_cnt = take(4 * S * Us); l.i_nzi = take(4 * S * Us + 4 * S);
    l.i_misc = take(4 * 8);
    l.f_obs_inter = take(4 * 10 * S); l.f_obs_intra = take(4 * S * (2 * Us + 9));
    l.total = o;
    return l;
}

// ---------------------------------------------------------------------------------------------
// numpy arithmetic on the device
// ---------------------------------------------------------------------------------------------
DEVFN bool d_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }

// numpy pairwise_sum for n <= 128 (one leaf), strided doubles in LDS.
DEVFN double np_sum_lds(const double *a, int n, int stride)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i * stride];
        return res;
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

// agents/common.py:481-505 round_int_equal_sum, one lane, n <= 128.
// Remainder hand-out: descending value, ties by descending index (= stable argsort reversed).
DEVFN void d_round_int_equal_sum(const double *v, int n, long long target, int *out, double *nzv, int *nzi)
{
    int m = 0;
    for (int i = 0; i < n; i++) {
        out[i] = 0;
        const double x = v[i];
        if (x != 0.0) { nzi[m] = i; nzv[m] = x; m++; }
    }
    const double total = np_sum_lds(nzv, m, 1);
    long long acc = 0;
    for (int i = 0; i < m; i++) {
        const long long pr = (long long)floor((double)target * nzv[i] / total);
        out[nzi[i]] = (int)pr;
        acc += pr;
    }
    const long long adj = target - acc;
    if (m > 0 && adj > 0) {
        const long long q = adj / m, r = adj % m;
        for (int i = 0; i < m; i++) {
            const double xi = nzv[i];
            int rank = 0;
            for (int j = 0; j < m; j++) {
                const double xj = nzv[j];
                rank += (xj > xi || (xj == xi && j > i)) ? 1 : 0;
            }
            out[nzi[i]] += (int)(q + (rank < r ? 1 : 0));
        }
    }
}

// agents/common.py:508-555 round_robin for one slice (one lane).
DEVFN void d_round_robin(const double *d_occ, const int *slues, int n, long long n_rbs, bool account_buffer,
                         int *counts)
{
    int k_sel = 0;
    if (account_buffer)
        for (int k = 0; k < n; k++) k_sel += d_isclose(d_occ[slues[k]], 0.0) ? 0 : 1;
    const bool all = (k_sel == 0);
    if (all) k_sel = n;
    const long long each = (long long)floor((double)n_rbs / (double)k_sel);
    const long long rem = n_rbs % k_sel;
    int i = 0;
    for (int k = 0; k < n; k++) {
        const bool sel = all || !d_isclose(d_occ[slues[k]], 0.0);
        counts[k] = sel ? (int)(each + (i < rem ? 1 : 0)) : 0;
        i += sel ? 1 : 0;
    }
}

// One leaf (n <= 128) of numpy's pairwise sum over the RB-major SE column of this lane.
// full = sum of the row, part = sum over RBs selected by in(r).  col points at tile[r0*U + u].
template <typename InFn>
DEVFN void row_leaf(const float *col, int U, int r0, int n, InFn in, double &full, double &part)
{
    if (n < 8) {
        double f = 0.0, g = 0.0;
        for (int i = 0; i < n; i++) {
            const double x = (double)col[(size_t)i * U];
            f += x;
            g += in(r0 + i) ? x : 0.0;
        }
        full = f; part = g;
        return;
    }
    double f[8], g[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double x = (double)col[(size_t)j * U];
        f[j] = x;
        g[j] = in(r0 + j) ? x : 0.0;
    }
    const int m = n - (n % 8);
    int i;
    for (i = 8; i < m; i += 8) {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) xs[j] = col[(size_t)(i + j) * U];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const double x = (double)xs[j];
            f[j] += x;
            g[j] += in(r0 + i + j) ? x : 0.0;
        }
    }
    double fr = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
    double gr = ((g[0] + g[1]) + (g[2] + g[3])) + ((g[4] + g[5]) + (g[6] + g[7]));
    for (; i < n; i++) {
        const double x = (double)col[(size_t)i * U];
        fr += x;
        gr += in(r0 + i) ? x : 0.0;
    }
    full = fr; part = gr;
}

// numpy pairwise sum of a whole row of n RBs: split at n/2 rounded down to a multiple of 8 while
// n > 128 (two levels are enough for n <= 512, checked at create).
template <typename InFn>
DEVFN void row_sums(const float *col, int U, int n, InFn in, double &full, double &part)
{
    if (n <= 128) { row_leaf(col, U, 0, n, in, full, part); return; }
    int n2 = n / 2; n2 -= n2 % 8;
    double fl, gl, fr, gr;
    auto half = [&](int off, int len, double &f, double &g) {
        if (len <= 128) { row_leaf(col + (size_t)off * U, U, off, len, in, f, g); return; }
        int h = len / 2; h -= h % 8;
        double f0, g0, f1, g1;
        row_leaf(col + (size_t)off * U, U, off, h, in, f0, g0);
        row_leaf(col + (size_t)(off + h) * U, U, off + h, len - h, in, f1, g1);
        f = f0 + f1; g = g0 + g1;
    };
    half(0, n2, fl, gl);
    half(n2, n - n2, fr, gr);
    full = fl + fr; part = gl + gr;
}

// ---------------------------------------------------------------------------------------------
// the step kernel
// ---------------------------------------------------------------------------------------------
template <int MODE, int NT>
__global__ void __launch_bounds__(NT) ranenv_kernel(const KP p)
{
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    if (p.env_mask != nullptr && p.env_mask[b] == 0) return;  // uniform per workgroup

    const int S = p.S, U = p.U, R = p.R, Us = p.Us, D = p.D;
    const int W = 2 * Us + 9;
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsLayout lo = make_layout(S, U, Us);
    double *d_occ = (double *)(smem + lo.d_occ), *d_sem = (double *)(smem + lo.d_sem);
    double *d_hmean = (double *)(smem + lo.d_hmean), *d_semn = (double *)(smem + lo.d_semn);
    double *d_occn = (double *)(smem + lo.d_occn), *d_drift = (double *)(smem + lo.d_drift);
    double *d_slice = (double *)(smem + lo.d_slice), *d_scores = (double *)(smem + lo.d_scores);
    double *d_tmp = (double *)(smem + lo.d_tmp), *d_slvals = (double *)(smem + lo.d_slvals);
    double *d_slflags = (double *)(smem + lo.d_slflags), *d_par = (double *)(smem + lo.d_par);
    double *d_slf = (double *)(smem + lo.d_slf);
    int *i_maxpkts = (int *)(smem + lo.i_maxpkts), *i_pktsize = (int *)(smem + lo.i_pktsize);
    int *i_start = (int *)(smem + lo.i_start), *i_count = (int *)(smem + lo.i_count);
    int *i_sl = (int *)(smem + lo.i_sl), *i_slues = (int *)(smem + lo.i_slues), *i_par = (int *)(smem + lo.i_par);
    int *i_rbs = (int *)(smem + lo.i_rbs), *i_off = (int *)(smem + lo.i_off), *i_cnt = (int *)(smem + lo.i_cnt);
    int *i_nzi = (int *)(smem + lo.i_nzi), *i_misc = (int *)(smem + lo.i_misc);
    float *f_obs_inter = (float *)(smem + lo.f_obs_inter), *f_obs_intra = (float *)(smem + lo.f_obs_intra);

    // ---- P0: per-env scalars, per-UE state, scenario tables -------------------------------------
    const ranenv_episode ep = p.episodes[b];
    const int sc = ep.scenario;
    const int step = (MODE == MODE_RESET) ? 0 : p.st.step_no[b];
    int hlen = p.st.hist_len[b];
    const int npush = p.st.n_push[b];
    if (MODE == MODE_RESET && (p.flags & RANENV_F_CLEAR_HISTORY_ON_RESET)) hlen = 0;
    const int t = step;

    const float *tile;
    if (p.se_tiles != nullptr) {
        tile = p.se_tiles + (size_t)b * U * R;
    } else {
        const long long ti = ep.se_base + (long long)((ep.se_offset + t) % ep.se_len);
        tile = p.se_pool + (size_t)ti * (size_t)p.se_stride;
    }

    const bool is_ue = tid < U;
    const int u = tid;
    const size_t su = (size_t)b * U + u;       // index into [B][U] state
    const size_t tu = (size_t)sc * U + u;      // index into [NS][U] tables
    int total = 0, front = 0, front_rem = 0, ue_slice = -1, ue_pos = 0, pkt_size = 1, max_pkts = 1, max_age = 0;
    long long sum_age = 0, win_sent = 0, win_drop = 0;
    if (is_ue) {
        ue_slice = p.tab.ue_slice[tu]; ue_pos = p.tab.ue_pos[tu];
        pkt_size = p.tab.ue_pkt_size[tu]; max_pkts = p.tab.ue_max_pkts[tu]; max_age = p.tab.ue_max_age[tu];
        double sem_prev = 0.0;
        if (MODE != MODE_RESET) {
            total = p.st.queue_pkts[su]; sum_age = p.st.queue_age_sum[su];
            front = p.st.front[su]; front_rem = p.st.front_rem[su];
            sem_prev = p.st.se_mean[su];
        }
        if (!(MODE == MODE_RESET && (p.flags & RANENV_F_CLEAR_HISTORY_ON_RESET))) {
            win_sent = p.st.win_sent[su]; win_drop = p.st.win_dropped[su];
        }
        d_occ[u] = (double)total / (double)max_pkts;
        d_sem[u] = sem_prev;
        d_hmean[u] = hlen > 0 ? (double)win_sent / (double)hlen : 0.0;
        i_maxpkts[u] = max_pkts; i_pktsize[u] = pkt_size;
        i_start[u] = 0; i_count[u] = 0;
    }
    if (tid < S) {
        const int s = tid;
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
    }
    for (int i = tid; i < S * Us; i += NT) i_slues[i] = p.tab.slice_ues[(size_t)sc * S * Us + i];
    for (int i = tid; i < S * Us * 3; i += NT) d_drift[i] = 0.0;
    __syncthreads();

    // slice table accessors
    auto sl_active = [&](int s) { return i_sl[s * 8 + 0]; };
    auto sl_hasreq = [&](int s) { return i_sl[s * 8 + 1]; };
    auto sl_nues = [&](int s) { return i_sl[s * 8 + 2]; };
    auto sl_bsize = [&](int s) { return i_sl[s * 8 + 3]; };
    auto sl_blat = [&](int s) { return i_sl[s * 8 + 4]; };
    auto sl_msg = [&](int s) { return i_sl[s * 8 + 5]; };
    auto sl_npar = [&](int s) { return i_sl[s * 8 + 6]; };
    auto sl_sorted = [&](int pos) { return i_sl[pos * 8 + 7]; };

    if (MODE == MODE_STEP) {
        // ---- P1: inter-slice scores: external, MARR (marr.py:40-47) or MAPF (mapf.py:41-111) ----
        const bool ext = p.scores != nullptr;
        if (!ext && p.policy == RANENV_POLICY_MAPF) {
            if (tid < S) {
                const int s = tid;
                double occ_mb = 0.0, thr_mb = 0.0;
                if (sl_active(s)) {
                    const int n = sl_nues(s);
                    double *tmp = d_slice + (size_t)s * 4 * Us;
                    const double pkt = (double)sl_msg(s), bmax = (double)sl_bsize(s);
                    for (int k = 0; k < n; k++) tmp[k] = d_occ[i_slues[s * Us + k]];
                    occ_mb = ((np_sum_lds(tmp, n, 1) / (double)n * bmax) * pkt) / 1e6;
                    for (int k = 0; k < n; k++) tmp[k] = d_hmean[i_slues[s * Us + k]];
                    thr_mb = ((np_sum_lds(tmp, n, 1) / (double)n) * pkt) / 1e6;
                }
                d_tmp[s] = occ_mb; d_tmp[S + s] = thr_mb;
            }
            __syncthreads();
            if (tid < S) {
                const int s = tid;
                double mx = d_tmp[0];
                for (int j = 1; j < S; j++) mx = d_tmp[j] > mx ? d_tmp[j] : mx;
                double w = d_isclose(d_tmp[S + s], 0.0) ? 2.0 * mx : d_tmp[s] / d_tmp[S + s];
                if (!sl_active(s)) w = 0.0;
                d_tmp[2 * S + s] = w;
            }
            __syncthreads();
            if (tid < S) {
                const double ws = np_sum_lds(d_tmp + 2 * S, S, 1);
                d_scores[tid] = (ws > 0.0 ? d_tmp[2 * S + tid] / ws : 2.0) - 1.0;
            }
        } else if (tid < S) {
            d_scores[tid] = ext ? p.scores[(size_t)b * S + tid] : (sl_nues(tid) > 0 ? 1.0 : -1.0);
        }
        __syncthreads();
        if (tid < S) p.st.policy_scores[(size_t)b * S + tid] = d_scores[tid];

        // ---- P2: inter-slice RBG split, one lane (ib_sched.py:240-269, common.py:442-461) -------
        if (tid == 0) {
            int any_active = 0;
            for (int i = 0; i < S; i++) any_active += sl_active(i);
            i_misc[0] = any_active;
            if (any_active) {
                double *a = d_tmp, *ap1 = d_tmp + S, *v = d_tmp + 2 * S, *nzv = d_tmp + 3 * S;
                for (int i = 0; i < S; i++) {
                    a[i] = sl_active(i) ? d_scores[sl_sorted(i)] : -1.0;   // ib_sched.py:247-255
                    ap1[i] = a[i] + 1.0;
                }
                const long long T = (long long)floor((double)R / (double)p.G);
                const double ssum = np_sum_lds(ap1, S, 1);
                if (ssum != 0.0) {
                    for (int i = 0; i < S; i++) v[i] = (double)T * (a[i] + 1.0) / ssum;
                } else {
                    for (int i = 0; i < S; i++) ap1[i] = (double)sl_active(i);
                    const double per = (double)T / np_sum_lds(ap1, S, 1);
                    for (int i = 0; i < S; i++) v[i] = per * ap1[i];
                }
                d_round_int_equal_sum(v, S, T, i_rbs, nzv, i_nzi + S * Us);
                int off = 0;
                for (int i = 0; i < S; i++) { i_rbs[i] *= p.G; i_off[i] = off; off += i_rbs[i]; }
            }
        }
        __syncthreads();

        // ---- P3: intra-slice scheduling, one lane per slice (ib_sched.py:272-344) ---------------
        if (tid < S && i_misc[0] != 0 && sl_nues(tid) > 0) {
            const int s = tid, n = sl_nues(s);
            const long long n_rbs = i_rbs[s];
            const int *slues = i_slues + s * Us;
            int *counts = i_cnt + s * Us;
            int choice = p.fixed_intra;
            if (choice == RANENV_INTRA_PER_SLICE) choice = p.intra ? (int)p.intra[(size_t)b * S + s] : RANENV_INTRA_RR;
            if (choice == RANENV_INTRA_RR) {
                d_round_robin(d_occ, slues, n, n_rbs, true, counts);
            } else {
                double *avail = d_slice + (size_t)s * 4 * Us, *wts = avail + Us, *vals = avail + 2 * Us, *nzv = avail + 3 * Us;
                const double slice_bw = (double)n_rbs * p.bw_hz / (double)R;     // common.py:573-578
                double max_avail = 0.0;
                for (int k = 0; k < n; k++) {
                    const int ue = slues[k];
                    const double cap = d_sem[ue] * slice_bw / (double)n;
                    const double backlog = d_occ[ue] * (double)i_maxpkts[ue] * (double)i_pktsize[ue];
                    const double av = cap < backlog ? cap : backlog;
                    avail[k] = av;
                    max_avail = (k == 0 || av > max_avail) ? av : max_avail;
                }
                const double *num = avail;
                if (choice == RANENV_INTRA_PF) {                                   // common.py:584-602
                    for (int k = 0; k < n; k++) {
                        const int ue = slues[k];
                        double snt = d_hmean[ue] * (double)i_pktsize[ue];
                        if (d_isclose(avail[k], 0.0)) snt = 1.0;
                        wts[k] = d_isclose(snt, 0.0) ? 2.0 * max_avail : avail[k] / snt;
                    }
                    num = wts;
                }
                const double wsum = np_sum_lds(num, n, 1);
                if (wsum != 0.0) {
                    for (int k = 0; k < n; k++) vals[k] = (double)n_rbs * num[k] / wsum;
                    d_round_int_equal_sum(vals, n, n_rbs, counts, nzv, i_nzi + s * Us);
                } else {
                    d_round_robin(d_occ, slues, n, n_rbs, false, counts);          // common.py:609-617
                }
            }
            int pos = i_off[s];                                                    // common.py:464-478
            for (int k = 0; k < n; k++) {
                const int ue = slues[k];
                i_start[ue] = pos; i_count[ue] = counts[k];
                pos += counts[k];
            }
        }
        __syncthreads();
    }

    // ---- P4: this UE's SE row: mean over all RBs and sum over its allocated RBs -------------------
    int rb_start = 0, rb_count = 0;
    double se_full = 0.0, se_part = 0.0;
    if (is_ue) {
        const float *col = tile + u;
        if (MODE == MODE_STEP) {
            rb_start = i_start[u]; rb_count = i_count[u];
            const unsigned ust = (unsigned)rb_start, ucn = (unsigned)rb_count;
            row_sums(col, U, R, [=](int r) { return ((unsigned)r - ust) < ucn; }, se_full, se_part);
        } else if (MODE == MODE_DENSE) {
            const uint8_t *mrow = p.dense + ((size_t)b * U + u) * R;
            row_sums(col, U, R, [=](int r) { return mrow[r] != 0; }, se_full, se_part);
            bool seen = false;
            for (int r = 0; r < R; r++) {
                if (mrow[r] != 0) { rb_count++; if (!seen) { rb_start = r; seen = true; } }
            }
            i_count[u] = rb_count;
        } else {
            row_sums(col, U, R, [](int) { return false; }, se_full, se_part);
        }
    }
    const double se_mean_new = se_full / (double)R;

    // ---- P5: UEs.step for this UE (oracle/ranenv_oracle.c buffer_receive/buffer_send) -------------
    long long dropped = 0, sent = 0, pkt_in = 0, pkt_thr = 0;
    const int hlen_new = hlen < D ? hlen + 1 : D;
    if (is_ue) {
        if (MODE != MODE_RESET) {
            const double traffic = p.traffic_bits
                ? p.traffic_bits[su]
                : (double)p.trf_pool[((size_t)ep.trf_base + (size_t)((ep.trf_offset + t) % ep.trf_len)) * U + u];
            const double psz = (double)pkt_size;
            pkt_thr = (long long)floor((se_part * p.bw_per_rb) / psz);
            pkt_in = (long long)floor(traffic / psz);
            const int L = p.L;
            const int tslot = t % L;
            int32_t *ring = p.st.age_ring + (size_t)b * L * U + u;
            auto slot_of = [&](int f) { int sl = tslot - (t - f); return sl < 0 ? sl + L : sl; };
            // receive_packets: the bin older than max_age expires ...
            if (total > 0 && front == t - max_age - 1) {
                dropped += front_rem; total -= front_rem; sum_age -= (long long)max_age * front_rem;
                front_rem = 0;
                if (total > 0) {
                    do { front++; front_rem = ring[(size_t)slot_of(front) * U]; } while (front_rem == 0 && front < t - 1);
                }
            }
            sum_age += total;                                   // ... everything left ages one TTI ...
            const long long space = (long long)max_pkts - total; // ... arrivals admitted up to capacity
            const long long adm = pkt_in < space ? pkt_in : space;
            dropped += pkt_in - adm;
            ring[(size_t)tslot * U] = (int32_t)adm;
            if (total == 0) { front = t; front_rem = (int)adm; }
            total += (int)adm;
            // send_packets: drain oldest first
            long long cap = pkt_thr;
            while (cap > 0 && total > 0) {
                const long long take = cap < front_rem ? cap : front_rem;
                front_rem -= (int)take; total -= (int)take; cap -= take; sent += take;
                sum_age -= (long long)(t - front) * take;
                if (front_rem == 0 && total > 0) {
                    do {
                        front++;
                        front_rem = (front == t) ? (int)adm : ring[(size_t)slot_of(front) * U];
                    } while (front_rem == 0 && front < t);
                }
            }
        }
        // push into the 10-TTI window (IBSched.last_unformatted_obs.appendleft, ib_sched.py:64)
        const int wslot = npush % D;
        int32_t *rs = p.st.ring_sent + ((size_t)b * D + wslot) * U + u;
        int32_t *rd = p.st.ring_drop + ((size_t)b * D + wslot) * U + u;
        if (hlen == D) { win_sent -= *rs; win_drop -= *rd; }
        win_sent += sent; win_drop += dropped;
        *rs = (int32_t)sent; *rd = (int32_t)dropped;
        // state + raw outputs
        p.st.queue_pkts[su] = total; p.st.queue_age_sum[su] = sum_age;
        p.st.front[su] = front; p.st.front_rem[su] = front_rem;
        p.st.win_sent[su] = win_sent; p.st.win_dropped[su] = win_drop;
        p.st.se_mean[su] = se_mean_new;
        p.st.pkt_effective_thr[su] = (int32_t)sent; p.st.dropped_pkts[su] = (int32_t)dropped;
        p.st.rb_start[su] = rb_start; p.st.rb_count[su] = rb_count;
        if (!(p.flags & RANENV_F_NO_RAW_OUTPUT)) {
            p.st.pkt_incoming[su] = (int32_t)pkt_in; p.st.pkt_throughputs[su] = (int32_t)pkt_thr;
        }
        const double occ_new = (double)total / (double)max_pkts;
        const double lat_new = total > 0 ? (double)sum_age / (double)total : 0.0;
        d_occn[u] = occ_new; d_semn[u] = se_mean_new;
        if (MODE == MODE_RESET) i_count[u] = 0;

        // intent drift of this UE (agents/common.py:68-340)
        const int s = ue_slice;
        if (s >= 0 && sl_hasreq(s)) {
            const double o = p.over;
            const int npar = sl_npar(s);
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
    __syncthreads();

    // ---- P6: per-slice observation rows, sorted order (ib_sched.py:91-200) ------------------------
    if (tid < S) {
        const int pos = tid;
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
        double *se_u = d_slice + (size_t)s * 4 * Us;
        double rbs_alloc = 0.0;
        for (int k = 0; k < n; k++) {
            const int ue = i_slues[s * Us + k];
            se_u[k] = d_semn[ue];
            rbs_alloc += (double)i_count[ue];
        }
        const double se_slice = n > 0 ? np_sum_lds(se_u, n, 1) / (double)n : 0.0;
        float *oi = f_obs_inter + pos * 10;
        oi[0] = (float)sv[0]; oi[1] = (float)sv[1]; oi[2] = (float)sv[2];
        oi[3] = (float)am[0]; oi[4] = (float)am[1]; oi[5] = (float)am[2];
        oi[6] = (float)priority; oi[7] = (float)(traffic_req / p.norm_traffic);
        oi[8] = (float)((double)n / p.norm_ues); oi[9] = (float)(se_slice / p.norm_se);
        float *oa = f_obs_intra + (size_t)s * W;
        oa[0] = oi[0]; oa[1] = oi[1]; oa[2] = oi[2]; oa[3] = oi[3]; oa[4] = oi[4]; oa[5] = oi[5];
        oa[6] = (float)(rbs_alloc / (double)R); oa[7] = oi[7]; oa[8] = oi[8];
        for (int k = 0; k < Us; k++) {
            oa[9 + k] = k < n ? (float)d_occn[i_slues[s * Us + k]] : 0.0f;
            oa[9 + Us + k] = k < n ? (float)(se_u[k] / p.norm_se) : 0.0f;
        }
        // keep the float64 values for the reward (calculate_reward reads the same numbers)
        d_slvals[pos * 3 + 0] = sv[0]; d_slvals[pos * 3 + 1] = sv[1]; d_slvals[pos * 3 + 2] = sv[2];
        d_slflags[pos * 3 + 0] = am[0]; d_slflags[pos * 3 + 1] = am[1]; d_slflags[pos * 3 + 2] = am[2];
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

    // ---- P7: player_0 reward (common.py:389-427 after unsort_slices, ib_sched.py:372-392) ----------
    if (tid == 0) {
        double *active_obs = d_tmp, *prio = d_tmp + S, *sel = d_tmp + 2 * S;
        for (int s = 0; s < S; s++) { active_obs[s] = 0.0; prio[s] = 0.0; }
        for (int pos = 0; pos < S; pos++) {
            const int s = sl_sorted(pos);
            if (!sl_active(s)) continue;
            prio[s] = d_slf[s * 2 + 0];
            double mn = 0.0; int cnt = 0;
            for (int m = 0; m < 3; m++) {
                const double v = d_slvals[pos * 3 + m];
                if (d_isclose(v, -2.0)) continue;
                mn = (cnt == 0 || v < mn) ? v : mn;
                cnt++;
            }
            active_obs[s] = cnt > 0 ? mn : 1.0;
        }
        int n_neg = 0, n_prio_neg = 0;
        for (int s = 0; s < S; s++) {
            n_neg += active_obs[s] < 0.0 ? 1 : 0;
            n_prio_neg += (prio[s] * active_obs[s] < 0.0) ? 1 : 0;
        }
        double rew;
        if (n_neg == 0) {
            rew = np_sum_lds(active_obs, S, 1) / (double)S;
        } else if (n_prio_neg != 0) {
            int m = 0;
            for (int s = 0; s < S; s++) if (active_obs[s] * prio[s] < 0.0) sel[m++] = active_obs[s];
            rew = np_sum_lds(sel, m, 1) / (double)m - 1.0;
        } else {
            int m = 0;
            for (int s = 0; s < S; s++) if (active_obs[s] < 0.0) sel[m++] = active_obs[s];
            rew = np_sum_lds(sel, m, 1) / (double)m;
        }
        if (p.reward) p.reward[(size_t)b * (S + 1)] = rew;
        const int step_new = (MODE == MODE_RESET) ? 0 : step + 1;
        p.st.step_no[b] = step_new;
        p.st.hist_len[b] = hlen_new;
        p.st.n_push[b] = (npush + 1) % (D * 1024);
        if (p.done) p.done[b] = (MODE != MODE_RESET && step_new >= p.max_steps) ? 1 : 0;
    }
    if (p.obs_inter) for (int i = tid; i < S * 10; i += NT) p.obs_inter[(size_t)b * S * 10 + i] = f_obs_inter[i];
    if (p.obs_intra) for (int i = tid; i < S * W; i += NT) p.obs_intra[(size_t)b * S * W + i] = f_obs_intra[i];
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
    dim3 grid(kp.B), block(h->nt);
    const size_t lds = (size_t)h->lds_bytes;
    switch (h->nt) {
    case 64:   hipLaunchKernelGGL((ranenv_kernel<MODE, 64>), grid, block, lds, stream, kp); break;
    case 128:  hipLaunchKernelGGL((ranenv_kernel<MODE, 128>), grid, block, lds, stream, kp); break;
    case 256:  hipLaunchKernelGGL((ranenv_kernel<MODE, 256>), grid, block, lds, stream, kp); break;
    case 512:  hipLaunchKernelGGL((ranenv_kernel<MODE, 512>), grid, block, lds, stream, kp); break;
    default:   hipLaunchKernelGGL((ranenv_kernel<MODE, 1024>), grid, block, lds, stream, kp); break;
    }
    return hipGetLastError();
}

template <int MODE, int NT>
hipError_t set_lds_attr(int bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&ranenv_kernel<MODE, NT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int NT>
hipError_t set_lds_attr_all(int bytes)
{
    hipError_t e = set_lds_attr<MODE_STEP, NT>(bytes);
    if (e == hipSuccess) e = set_lds_attr<MODE_DENSE, NT>(bytes);
    if (e == hipSuccess) e = set_lds_attr<MODE_RESET, NT>(bytes);
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
    if (cfg->batch < 1 || S < 1 || S > 128 || U < 1 || U > 1024 || R < 1 || R > 512 || Us < 1 || Us > 128 ||
        cfg->rbs_per_rbg < 1 || cfg->rbs_per_rbg > R || cfg->hist_depth < 1 || cfg->hist_depth > 64 ||
        cfg->max_age_cap < 1 || cfg->max_steps < 1 || cfg->n_scenarios < 1)
        return fail(nullptr, RANENV_E_INVALID,
                    "unsupported sizes: need 1<=S<=128, 1<=U<=1024, 1<=R<=512, 1<=Us<=128, 1<=G<=R, 1<=hist_depth<=64");
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
    ALLOC(kp.st.front_rem, B * U); ALLOC(kp.st.win_sent, B * U); ALLOC(kp.st.win_dropped, B * U);
    ALLOC(kp.st.se_mean, B * U);
    ALLOC(kp.st.age_ring, B * L * U); ALLOC(kp.st.ring_sent, B * D * U); ALLOC(kp.st.ring_drop, B * D * U);
    ALLOC(kp.st.hist_len, B); ALLOC(kp.st.n_push, B); ALLOC(kp.st.step_no, B);
    ALLOC(kp.st.pkt_incoming, B * U); ALLOC(kp.st.pkt_throughputs, B * U); ALLOC(kp.st.pkt_effective_thr, B * U);
    ALLOC(kp.st.dropped_pkts, B * U); ALLOC(kp.st.rb_start, B * U); ALLOC(kp.st.rb_count, B * U);
    ALLOC(kp.st.mask_inter, B * S); ALLOC(kp.st.mask_intra, B * S * Us); ALLOC(kp.st.policy_scores, B * S);
    ALLOC(h->d_episodes, B);
#undef ALLOC
    if (rc != RANENV_OK) { std::string m = h->err; ranenv_destroy(h); g_last_error = m; return rc; }
    kp.episodes = h->d_episodes;
    h->nt = U <= 64 ? 64 : U <= 128 ? 128 : U <= 256 ? 256 : U <= 512 ? 512 : 1024;
    h->lds_bytes = make_layout(S, U, Us).total;
    if (h->lds_bytes > 160 * 1024) { ranenv_destroy(h); return fail(nullptr, RANENV_E_INVALID, "LDS need %d B exceeds 160 KiB", h->lds_bytes); }
    switch (h->nt) {
    case 64: e = set_lds_attr_all<64>(h->lds_bytes); break;
    case 128: e = set_lds_attr_all<128>(h->lds_bytes); break;
    case 256: e = set_lds_attr_all<256>(h->lds_bytes); break;
    case 512: e = set_lds_attr_all<512>(h->lds_bytes); break;
    default: e = set_lds_attr_all<1024>(h->lds_bytes); break;
    }
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
        if (e.se_len < 1 || e.se_offset < 0 || e.se_base < 0 || e.trf_len < 1 || e.trf_offset < 0 || e.trf_base < 0)
            return fail(h, RANENV_E_INVALID, "env %d: episode lengths must be >= 1 and offsets/bases >= 0", b);
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
