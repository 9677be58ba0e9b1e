// ranenv_aux.hip -- the small kernels of libranenv_hip.so and their launch functions (ranenv_internal.h, "launch table").
#include "ranenv_numeric.hpp"

namespace {

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

// ddiv() against the compiler's IEEE division on caller-supplied operands (ranenv_selftest_ddiv: the parity check of the guard-free
// sequence inside the shipped build -- no second build with RANENV_FAST_DIV=0 needed)
__global__ void __launch_bounds__(256) ranenv_ddiv_selftest_kernel(const double *a, const double *b, double *fast, double *ieee, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        fast[i] = ddiv(a[i], b[i]);
        ieee[i] = a[i] / b[i];
    }
}

}  // namespace

namespace ranenv_dev {

void launch_ddiv_selftest(hipStream_t s, const double *a, const double *b, double *fast, double *ieee, long long n)
{
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(ranenv_ddiv_selftest_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), 0, s, a, b, fast, ieee, n);
}

void launch_classify(hipStream_t s, const ranenv_episode *eps, const int32_t *members, int B, int n_class, int one_class, int32_t *list,
                     int32_t *count, int *flag, int force)
{
    hipLaunchKernelGGL(ranenv_persist_classify_kernel, dim3(1), dim3(1024), 0, s, eps, members, B, n_class, one_class, list, count, flag, force);
}
void launch_se_sidecar(hipStream_t s, unsigned n_tiles, unsigned block, const float *pool, long long stride, long long tile0, int U, int R, int Rp,
                       int quad, double *mean, float *um)
{
    hipLaunchKernelGGL(ranenv_se_sidecar_kernel, dim3(n_tiles), dim3(block), 0, s, pool, stride, tile0, U, R, Rp, quad, mean, um);
}
void launch_se_sidecar_from_power(hipStream_t s, unsigned n_tiles, unsigned block, const double *power, long long tile0, int U, int R, int Rp,
                                  double tx, double noise, double *mean, float *um)
{
    hipLaunchKernelGGL(ranenv_se_sidecar_from_power_kernel, dim3(n_tiles), dim3(block), 0, s, power, tile0, U, R, Rp, tx, noise, mean, um);
}
void launch_se_retile_quad(hipStream_t s, unsigned blocks, const float *src, float *dst, long long n_quads, int U, int R)
{
    hipLaunchKernelGGL(ranenv_se_retile_quad_kernel, dim3(blocks), dim3(256), 0, s, src, dst, n_quads, U, R);
}
void launch_se_from_power(hipStream_t s, unsigned blocks, const double *power, float *se, long long n, double tx_per_rb, double noise)
{
    hipLaunchKernelGGL(ranenv_se_from_power_kernel, dim3(blocks), dim3(256), 0, s, power, se, n, tx_per_rb, noise);
}
void launch_head(hipStream_t s, dim3 grid, dim3 block, const KP &kp) { hipLaunchKernelGGL(ranenv_head_kernel, grid, block, 0, s, kp); }
void launch_advance(hipStream_t s, unsigned n_envs, const AdvanceArgs &a) { hipLaunchKernelGGL(ranenv_advance_kernel, dim3(n_envs), dim3(64), 0, s, a); }
void launch_idle_traffic(hipStream_t s, unsigned n_eps, const ranenv_episode *eps, const int32_t *pool, int U, const int32_t *lane_slice,
                         const int32_t *lane_ue, int *violations)
{
    hipLaunchKernelGGL(ranenv_idle_traffic_kernel, dim3(n_eps), dim3(256), 0, s, eps, pool, U, lane_slice, lane_ue, violations);
}

}  // namespace ranenv_dev
