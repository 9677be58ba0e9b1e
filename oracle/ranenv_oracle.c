/*
 * ranenv_oracle.c -- CPU restatement of the per-TTI RAN-slicing env step.
 * TEST INFRASTRUCTURE ONLY (see ranenv_oracle.h for scope and parity status).
 *
 * Every function names the upstream file:line it follows.  Paths are relative to
 * lasseufpa/intent_radio_sched_multi_slice.  The env core (sixg_radio_mgmt) is an
 * un-vendored submodule: PARITY UNPINNED for UEs/Buffer/step ordering.
 */
#include "ranenv_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------ */
/* numpy arithmetic                                                                */
/* ------------------------------------------------------------------------------ */

/* numpy/_core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum: what np.sum / np.mean
 * run for a float64 reduction along one axis.  n < 8: plain loop; n <= 128: eight
 * strided accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) then the tail;
 * larger: split at n/2 rounded down to a multiple of 8. */
double orc_np_sum(const double *a, int64_t n, int64_t stride)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; i++) res += a[i * stride];
        return res;
    }
    if (n <= 128) {
        double r[8];
        int64_t i;
        for (int j = 0; j < 8; j++) r[j] = a[j * stride];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[(i + j) * stride];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i * stride];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return orc_np_sum(a, n2, stride) + orc_np_sum(a + n2 * stride, n - n2, stride);
}

static double np_mean(const double *a, int64_t n, int64_t stride)
{
    return orc_np_sum(a, n, stride) / (double)n;
}

/* np.isclose(a, b) with default rtol=1e-5, atol=1e-8 (finite inputs). */
static int np_isclose(double a, double b) { return fabs(a - b) <= (1e-8 + 1e-5 * fabs(b)); }

/* np.argsort(v, kind="stable").  The reference calls np.argsort with the default
 * kind (agents/common.py:496, agents/ib_sched.py:370), whose order among EQUAL keys
 * depends on the numpy build / CPU.  Canonical rule used by the whole
 * build: stable ascending (what numpy 1.26 gives for n <= 16 without AVX-512). */
void orc_stable_argsort(const double *v, int n, int32_t *idx)
{
    for (int i = 0; i < n; i++) idx[i] = i;
    for (int i = 1; i < n; i++) { /* insertion sort: stable */
        int32_t k = idx[i];
        int j = i - 1;
        while (j >= 0 && v[idx[j]] > v[k]) { idx[j + 1] = idx[j]; j--; }
        idx[j + 1] = k;
    }
}

static int apply_op(int op, double a, double b)
{
    switch (op) { /* associations/mult_slice.py:48-55 */
    case ORC_OP_GE: return a >= b;
    case ORC_OP_LE: return a <= b;
    case ORC_OP_EQ: return a == b;
    case ORC_OP_GT: return a > b;
    case ORC_OP_LT: return a < b;
    default: return 0;
    }
}

/* ------------------------------------------------------------------------------ */
/* agents/common.py stateless pieces                                               */
/* ------------------------------------------------------------------------------ */

/* agents/common.py:481-505 round_int_equal_sum */
void orc_round_int_equal_sum(const double *v, int n, int64_t target, int64_t *out)
{
    double *nzv = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int32_t *nzi = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t *ord = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int64_t *prop = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; i++) {
        out[i] = 0;
        if (v[i] != 0.0) { nzi[m] = i; nzv[m] = v[i]; m++; }          /* :484-485 */
    }
    double total = orc_np_sum(nzv, m, 1);
    int64_t acc = 0;
    for (int i = 0; i < m; i++) {                                      /* :488-490 */
        prop[i] = (int64_t)floor((double)target * nzv[i] / total);
        acc += prop[i];
    }
    int64_t adjustment = target - acc;                                 /* :493 */
    if (m > 0 && adjustment > 0) {
        orc_stable_argsort(nzv, m, ord);                               /* :496, [::-1] below */
        for (int64_t i = 0; i < adjustment; i++) {                     /* :497-499 */
            int32_t index = ord[m - 1 - (int)(i % m)];
            prop[index] += 1;
        }
    }
    for (int i = 0; i < m; i++) out[nzi[i]] = prop[i];                 /* :502-503 */
    free(nzv); free(nzi); free(ord); free(prop);
}

/* agents/common.py:442-461 scores_to_rbs */
void orc_scores_to_rbs(const double *action, int n, int64_t total_rbs,
                       const double *association, int64_t *out)
{
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) tmp[i] = action[i] + 1.0;
    double s = orc_np_sum(tmp, n, 1);
    if (s != 0.0) {
        for (int i = 0; i < n; i++) tmp[i] = (double)total_rbs * (action[i] + 1.0) / s;
    } else {
        double sa = orc_np_sum(association, n, 1);
        double per = (double)total_rbs / sa;
        for (int i = 0; i < n; i++) tmp[i] = per * association[i];
    }
    orc_round_int_equal_sum(tmp, n, total_rbs, out);
    free(tmp);
}

/* agents/ib_sched.py:351-370 IBSched.sort_slices */
void orc_sort_slices(const int32_t *slice_nues, const double *slice_traffic,
                     const int32_t *slice_has_req, int n, int32_t *sorted_out)
{
    double *key = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        key[i] = (double)slice_nues[i] * (slice_has_req[i] ? slice_traffic[i] : 0.0);
    orc_stable_argsort(key, n, sorted_out);
    free(key);
}

/* ------------------------------------------------------------------------------ */
/* env object                                                                      */
/* ------------------------------------------------------------------------------ */

typedef struct {           /* one raw observation kept in IBSched.last_unformatted_obs */
    double *sent;          /* pkt_effective_thr   [U] */
    double *dropped;       /* dropped_pkts        [U] */
    double *occ;           /* buffer_occupancies  [U] */
    double *lat;           /* buffer_latencies    [U] */
    double *se_mean;       /* np.mean(spectral_efficiencies[0,u,:]) [U] */
    double *rowsum;        /* np.sum(sched_decision, axis=2)[0]     [U] */
} raw_rec;

struct orc_env {
    orc_cfg cfg;
    const orc_scenario *sc;
    int step_number;
    /* sixg_radio_mgmt.UEs: one Buffer per UE = age histogram [0..max_age] */
    int64_t *buf;          /* U * (max_age_cap+1) */
    /* last raw metrics */
    double *pkt_incoming, *pkt_throughputs;
    /* IBSched deque (maxlen = hist_depth); recs[(head + i) % depth] is deque[i] */
    raw_rec *recs;
    int head, hist_len;
    /* formatted observation + reward */
    double *drift;         /* S*Us*3 */
    double *obs_inter;     /* S*10 */
    int8_t *mask_inter;    /* S */
    double *obs_intra;     /* S*(2*Us+9) */
    int8_t *mask_intra;    /* S*Us */
    double *reward;        /* S+1 */
    double *scratch;       /* max(U, R, S) doubles x 4 */
    int scale_per_element; /* orc_env_set_scale_per_element (include/ranenv.h RANENV_F_SCALE_PER_ELEMENT) */
};

static raw_rec *deque_at(const orc_env *e, int i)
{
    return &e->recs[(e->head + i) % e->cfg.hist_depth];
}

static raw_rec *deque_appendleft(orc_env *e)
{
    int d = e->cfg.hist_depth;
    e->head = (e->head + d - 1) % d;
    if (e->hist_len < d) e->hist_len++;
    return &e->recs[e->head];
}

orc_env *orc_env_create(const orc_cfg *cfg)
{
    orc_env *e = (orc_env *)calloc(1, sizeof(orc_env));
    e->cfg = *cfg;
    int S = cfg->n_slices, U = cfg->n_ues, R = cfg->n_rbs, Us = cfg->max_ues_slice;
    int L = cfg->max_age_cap + 1;
    e->buf = (int64_t *)calloc((size_t)U * L, sizeof(int64_t));
    e->pkt_incoming = (double *)calloc(U, sizeof(double));
    e->pkt_throughputs = (double *)calloc(U, sizeof(double));
    e->recs = (raw_rec *)calloc(cfg->hist_depth, sizeof(raw_rec));
    for (int i = 0; i < cfg->hist_depth; i++) {
        raw_rec *r = &e->recs[i];
        r->sent = (double *)calloc(U, sizeof(double));
        r->dropped = (double *)calloc(U, sizeof(double));
        r->occ = (double *)calloc(U, sizeof(double));
        r->lat = (double *)calloc(U, sizeof(double));
        r->se_mean = (double *)calloc(U, sizeof(double));
        r->rowsum = (double *)calloc(U, sizeof(double));
    }
    e->drift = (double *)calloc((size_t)S * Us * 3, sizeof(double));
    e->obs_inter = (double *)calloc((size_t)S * 10, sizeof(double));
    e->mask_inter = (int8_t *)calloc(S, 1);
    e->obs_intra = (double *)calloc((size_t)S * (2 * Us + 9), sizeof(double));
    e->mask_intra = (int8_t *)calloc((size_t)S * Us, 1);
    e->reward = (double *)calloc(S + 1, sizeof(double));
    int m = U > R ? U : R;
    if (S > m) m = S;
    e->scratch = (double *)calloc((size_t)m * 4 + 16, sizeof(double));
    return e;
}

void orc_env_destroy(orc_env *e)
{
    if (!e) return;
    for (int i = 0; i < e->cfg.hist_depth; i++) {
        raw_rec *r = &e->recs[i];
        free(r->sent); free(r->dropped); free(r->occ); free(r->lat); free(r->se_mean); free(r->rowsum);
    }
    free(e->recs); free(e->buf); free(e->pkt_incoming); free(e->pkt_throughputs);
    free(e->drift); free(e->obs_inter); free(e->mask_inter); free(e->obs_intra);
    free(e->mask_intra); free(e->reward); free(e->scratch);
    free(e);
}

void orc_env_set_scale_per_element(orc_env *e, int on) { e->scale_per_element = on != 0; }

void orc_env_clear(orc_env *e)
{
    int U = e->cfg.n_ues, L = e->cfg.max_age_cap + 1;
    memset(e->buf, 0, sizeof(int64_t) * (size_t)U * L);
    e->head = 0; e->hist_len = 0; e->step_number = 0;
    for (int i = 0; i < e->cfg.hist_depth; i++) {
        raw_rec *r = &e->recs[i];
        memset(r->sent, 0, sizeof(double) * U); memset(r->dropped, 0, sizeof(double) * U);
        memset(r->occ, 0, sizeof(double) * U);  memset(r->lat, 0, sizeof(double) * U);
        memset(r->se_mean, 0, sizeof(double) * U); memset(r->rowsum, 0, sizeof(double) * U);
    }
}

void orc_env_set_scenario(orc_env *e, const orc_scenario *sc) { e->sc = sc; }

/* ------------------------------------------------------------------------------ */
/* env core: sixg_radio_mgmt Buffer / UEs  (PARITY UNPINNED)       */
/* ------------------------------------------------------------------------------ */

/* Buffer.receive_packets: age every queued packet by one TTI, drop the bin that
 * exceeds max_packets_age (gen_assoc_mult_slice.py:218-224 pins
 * max_packets_age == slice buffer_latency), then admit the arrivals up to
 * max_packets_buffer; the excess is dropped. Returns dropped packets. */
static int64_t buffer_receive(int64_t *hist, int max_age, int64_t max_pkts, int64_t n_in)
{
    int64_t dropped = hist[max_age];
    for (int a = max_age; a > 0; a--) hist[a] = hist[a - 1];
    hist[0] = 0;
    int64_t total = 0;
    for (int a = 0; a <= max_age; a++) total += hist[a];
    if (total + n_in <= max_pkts) {
        hist[0] = n_in;
    } else {
        dropped += n_in - (max_pkts - total);
        hist[0] = max_pkts - total;
    }
    return dropped;
}

/* Buffer.send_packets: drain oldest-first up to `capacity`; returns packets sent
 * (the "effective" throughput, agents/sched_twc.py:253-277 distinguishes it from
 * the capacity pkt_throughputs). */
static int64_t buffer_send(int64_t *hist, int max_age, int64_t capacity)
{
    int64_t sent = 0;
    for (int a = max_age; a >= 0 && capacity > 0; a--) {
        int64_t take = hist[a] < capacity ? hist[a] : capacity;
        hist[a] -= take; capacity -= take; sent += take;
    }
    return sent;
}

static void fill_se_mean(const orc_env *e, const float *se_tile, double *se_mean)
{
    int U = e->cfg.n_ues, R = e->cfg.n_rbs;
    double *row = e->scratch;
    for (int u = 0; u < U; u++) {
        for (int r = 0; r < R; r++) row[r] = (double)se_tile[(size_t)u * R + r];
        se_mean[u] = np_mean(row, R, 1);   /* np.mean(se[0, u, :]) ib_sched.py:110-116 */
    }
}

/* ------------------------------------------------------------------------------ */
/* agents/common.py intent drift                                                   */
/* ------------------------------------------------------------------------------ */

/* The deque as an agent sees it.  IBSched pushes every raw observation once (ib_sched.py:64).
 * SchedTWC / SchedColORAN push it twice (sched_twc.py:174-177: fake_agent.obs_space_format appends,
 * then the head appends again), so their 10-deep deque holds the last 5 TTIs, each twice:
 * entry i is TTI i/2.  dup = 1 or 2. */
static int view_len(const orc_env *e, int dup)
{
    int n = dup * e->hist_len;
    return n < e->cfg.hist_depth ? n : e->cfg.hist_depth;
}
static const raw_rec *view_at(const orc_env *e, int dup, int i) { return deque_at(e, i / dup); }

/* agents/common.py:9-65 get_metric_value for one UE of slice s. */
static double get_metric_value(const orc_env *e, int dup, int metric, int s, int ue)
{
    const orc_scenario *sc = e->sc;
    const raw_rec *r0 = deque_at(e, 0);
    if (metric == ORC_METRIC_THROUGHPUT)                               /* :25-31 */
        return (r0->sent[ue] * (double)sc->slice_message_size[s]) / 1e6;
    if (metric == ORC_METRIC_RELIABILITY) {                            /* :32-53, pkt-loss form */
        double sent_w = 0.0, drop_w = 0.0;                             /* calc_metric_interval */
        for (int i = 0; i < view_len(e, dup); i++) {
            sent_w += view_at(e, dup, i)->sent[ue];
            drop_w += view_at(e, dup, i)->dropped[ue];
        }
        double buffer_pkts = r0->occ[ue] * (double)sc->slice_buffer_size[s] + drop_w + sent_w;
        return buffer_pkts != 0.0 ? drop_w / buffer_pkts : 0.0;
    }
    return r0->lat[ue];                                                /* :58-61 */
}

/* agents/common.py:68-340 intent_drift_calc (reliability_pkt_loss=True) into drift[S*Us*3]. */
static void intent_drift_into(const orc_env *e, int dup, double *drift)
{
    const orc_scenario *sc = e->sc;
    int S = e->cfg.n_slices, Us = e->cfg.max_ues_slice;
    double o = e->cfg.overfulfill;
    memset(drift, 0, sizeof(double) * (size_t)S * Us * 3);
    const raw_rec *r0 = deque_at(e, 0);
    for (int s = 0; s < S; s++) {
        if (!sc->slice_has_req[s]) continue;                           /* :86-87 */
        int n = sc->slice_nues[s];
        for (int p = 0; p < sc->slice_nparams[s]; p++) {
            int metric = sc->param_metric[s * 3 + p];
            int op = sc->param_op[s * 3 + p];
            double value = sc->param_value[s * 3 + p];
            for (int k = 0; k < n; k++) {
                int ue = sc->slice_ues[s * Us + k];
                double x = get_metric_value(e, dup, metric, s, ue);
                double *dst = &drift[((size_t)s * Us + k) * 3 + metric];
                if (metric == ORC_METRIC_THROUGHPUT) {
                    /* :100-119 empty buffer now or in the previous deque entry => over-fulfilled */
                    int zero = np_isclose(r0->occ[ue], 0.0);
                    if (view_len(e, dup) > 1) zero = zero || np_isclose(view_at(e, dup, 1)->occ[ue], 0.0);
                    if (zero) x = value * (1.1 + o);
                    if (apply_op(op, x, value)) {                      /* :132-134 */
                        if (x > value * (1.0 + o)) *dst += 1.0;        /* :141-168 */
                        else *dst += (x - value) / (value * o);
                    } else {
                        *dst -= (value - x) / value;                   /* :171-181 */
                    }
                } else if (metric == ORC_METRIC_RELIABILITY) {
                    double band = (100.0 - value) / 100.0;
                    if (apply_op(op, 100.0 * (1.0 - x), value)) {      /* :124-126 */
                        if (x < band * (1.0 - o)) *dst += 1.0;         /* :188-220 */
                        else *dst += (band - x) / (band * o);
                    } else {
                        *dst -= (x - band) / (value / 100.0);          /* :223-233 */
                    }
                } else {
                    double max_latency = (double)sc->slice_buffer_latency[s]; /* :284-287 */
                    if (apply_op(op, x, value)) {
                        if (x < value * (1.0 - o)) *dst += 1.0;        /* :289-319 */
                        else *dst += (value - x) / (value * o);
                    } else {
                        *dst -= (x - value) / (max_latency - value);   /* :322-335 */
                    }
                }
            }
        }
    }
}

static void intent_drift_calc(orc_env *e) { intent_drift_into(e, 1, e->drift); }

/* ------------------------------------------------------------------------------ */
/* agents/ib_sched.py obs_space_format + calculate_reward                          */
/* ------------------------------------------------------------------------------ */

static void obs_space_format(orc_env *e)
{
    const orc_scenario *sc = e->sc;
    const orc_cfg *c = &e->cfg;
    int S = c->n_slices, Us = c->max_ues_slice;
    int W = 2 * Us + 9;
    const raw_rec *r0 = deque_at(e, 0);
    intent_drift_calc(e);                                              /* ib_sched.py:68-72 */
    for (int s = 0; s < S; s++) e->mask_inter[s] = (int8_t)sc->slice_active[s]; /* :76-81 */
    for (int pos = 0; pos < S; pos++) {                                /* :91 sorted order */
        int s = sc->sorted_slices[pos];
        int n = sc->slice_nues[s];
        /* agents/common.py:343-378 calculate_slice_ue_obs */
        double slice_values[3] = {-2.0, -2.0, -2.0};
        if (n > 0 && sc->slice_has_req[s]) {
            for (int p = 0; p < sc->slice_nparams[s]; p++) {
                int m = sc->param_metric[s * 3 + p];
                slice_values[m] = np_mean(&e->drift[((size_t)s * Us) * 3 + m], n, 3);
            }
        }
        double traffic_req = sc->slice_active[s] == 1 ? sc->slice_traffic[s] : 0.0; /* :125-134 */
        double priority = n != 0 ? sc->slice_priority[s] : 0.0;                     /* :135-141 */
        double active_metrics[3];
        for (int m = 0; m < 3; m++) {                                  /* :142-145 */
            int undeclared = np_isclose(slice_values[m], -2.0);
            active_metrics[m] = undeclared ? 0.0 : 1.0;
            if (undeclared) slice_values[m] = 0.0;
        }
        double *se_u = e->scratch;                                     /* per-UE mean SE */
        for (int k = 0; k < n; k++) se_u[k] = r0->se_mean[sc->slice_ues[s * Us + k]];
        double se_slice = n > 0 ? np_mean(se_u, n, 1) : 0.0;           /* :146-157 */
        double *oi = &e->obs_inter[pos * 10];                          /* :160-173 */
        oi[0] = slice_values[0]; oi[1] = slice_values[1]; oi[2] = slice_values[2];
        oi[3] = active_metrics[0]; oi[4] = active_metrics[1]; oi[5] = active_metrics[2];
        oi[6] = priority;
        oi[7] = traffic_req / c->norm_traffic;
        oi[8] = (double)n / c->norm_ues;
        oi[9] = se_slice / c->norm_se;
        double rbs_alloc = 0.0;                                        /* :176-181 */
        for (int k = 0; k < n; k++) rbs_alloc += r0->rowsum[sc->slice_ues[s * Us + k]];
        double *oa = &e->obs_intra[(size_t)s * W];                     /* :186-200 */
        oa[0] = slice_values[0]; oa[1] = slice_values[1]; oa[2] = slice_values[2];
        oa[3] = active_metrics[0]; oa[4] = active_metrics[1]; oa[5] = active_metrics[2];
        oa[6] = rbs_alloc / (double)c->n_rbs;
        oa[7] = traffic_req / c->norm_traffic;
        oa[8] = (double)n / c->norm_ues;
        for (int k = 0; k < Us; k++) {
            oa[9 + k] = k < n ? r0->occ[sc->slice_ues[s * Us + k]] : 0.0;
            oa[9 + Us + k] = k < n ? se_u[k] / c->norm_se : 0.0;
            e->mask_intra[s * Us + k] = k < n ? 1 : 0;
        }
    }
}

/* ib_sched.py:206-221 calculate_reward + :372-392 unsort_slices +
 * agents/common.py:381-439 calculate_reward_no_mask (priority_flag=True). */
static void calculate_reward(orc_env *e)
{
    const orc_scenario *sc = e->sc;
    int S = e->cfg.n_slices, Us = e->cfg.max_ues_slice, W = 2 * Us + 9;
    double *active_obs = e->scratch;            /* [S] */
    double *prio = e->scratch + S;              /* [S] */
    double *sel = e->scratch + 2 * S;           /* [S] */
    for (int s = 0; s < S; s++) { active_obs[s] = 0.0; prio[s] = 0.0; }
    for (int pos = 0; pos < S; pos++) {
        int s = sc->sorted_slices[pos];         /* unsorted[sorted[idx]] = obs[idx] */
        if (!sc->slice_active[s]) continue;     /* elements_idx, common.py:391-393 */
        prio[s] = sc->slice_priority[s];
        double mn = 0.0; int cnt = 0;
        for (int m = 0; m < 3; m++) {           /* common.py:400-407 */
            double v = e->obs_inter[pos * 10 + m];
            if (np_isclose(v, -2.0)) continue;
            mn = cnt == 0 ? v : (v < mn ? v : mn);
            cnt++;
        }
        active_obs[s] = cnt > 0 ? mn : 1.0;
    }
    int n_neg = 0, n_prio_neg = 0;
    for (int s = 0; s < S; s++) {
        if (active_obs[s] < 0.0) n_neg++;
        if (prio[s] * active_obs[s] < 0.0) n_prio_neg++;
    }
    if (n_neg == 0) {                                                  /* :409-410 */
        e->reward[0] = np_mean(active_obs, S, 1);
    } else if (n_prio_neg != 0) {                                      /* :411-422 */
        int m = 0;
        for (int s = 0; s < S; s++) if (active_obs[s] * prio[s] < 0.0) sel[m++] = active_obs[s];
        e->reward[0] = np_mean(sel, m, 1) - 1.0;
    } else {                                                           /* :423-427 */
        int m = 0;
        for (int s = 0; s < S; s++) if (active_obs[s] < 0.0) sel[m++] = active_obs[s];
        e->reward[0] = np_mean(sel, m, 1);
    }
    for (int s = 0; s < S; s++) {                                      /* :428-437 */
        const double *oa = &e->obs_intra[(size_t)s * W];
        double r = 0.0; int cnt = 0;
        for (int m = 0; m < 3; m++) {
            if (oa[3 + m] > 0.0) {
                r = cnt == 0 ? oa[m] : (oa[m] < r ? oa[m] : r);
                cnt++;
            }
        }
        e->reward[s + 1] = cnt > 0 ? r : 0.0;
    }
}

/* ------------------------------------------------------------------------------ */
/* agents/common.py intra-slice schedulers                                         */
/* ------------------------------------------------------------------------------ */

/* agents/common.py:508-555 round_robin.  counts[k] for k-th UE of the slice. */
static void round_robin(const orc_env *e, int s, int64_t n_rbs, int account_buffer, int64_t *counts)
{
    const orc_scenario *sc = e->sc;
    int Us = e->cfg.max_ues_slice, n = sc->slice_nues[s];
    const raw_rec *r0 = deque_at(e, 0);
    int k_sel = 0;
    int sel[1024];
    if (account_buffer) {                                              /* :519-524 */
        for (int k = 0; k < n; k++)
            if (!np_isclose(r0->occ[sc->slice_ues[s * Us + k]], 0.0)) sel[k_sel++] = k;
    }
    if (k_sel == 0) { for (int k = 0; k < n; k++) sel[k] = k; k_sel = n; }
    for (int k = 0; k < n; k++) counts[k] = 0;
    int64_t each = (int64_t)floor((double)n_rbs / (double)k_sel);     /* :525-527 */
    int64_t rem = n_rbs % k_sel;                                       /* :528-529 */
    for (int i = 0; i < k_sel; i++) counts[sel[i]] = each + (i < rem ? 1 : 0);
}

/* throughput_available of agents/common.py:567-583 (PF) and :648-664 (MT). */
static void throughput_available(const orc_env *e, int s, int64_t n_rbs, double *avail)
{
    const orc_scenario *sc = e->sc;
    int Us = e->cfg.max_ues_slice, n = sc->slice_nues[s];
    const raw_rec *r0 = deque_at(e, 0);
    double slice_bw = (double)n_rbs * e->cfg.bandwidth_hz / (double)e->cfg.n_rbs;
    for (int k = 0; k < n; k++) {
        int ue = sc->slice_ues[s * Us + k];
        double cap = r0->se_mean[ue] * slice_bw / (double)n;
        double backlog = r0->occ[ue] * (double)sc->ue_max_pkts[ue] * (double)sc->ue_pkt_size[ue];
        avail[k] = cap < backlog ? cap : backlog;   /* np.minimum */
    }
}

/* agents/common.py:558-636 proportional_fairness */
static void proportional_fairness(const orc_env *e, int s, int64_t n_rbs, int64_t *counts)
{
    const orc_scenario *sc = e->sc;
    int Us = e->cfg.max_ues_slice, n = sc->slice_nues[s];
    double *avail = e->scratch, *weights = e->scratch + Us, *vals = e->scratch + 2 * Us;
    throughput_available(e, s, n_rbs, avail);
    double max_avail = avail[0];
    for (int k = 1; k < n; k++) if (avail[k] > max_avail) max_avail = avail[k];
    for (int k = 0; k < n; k++) {
        int ue = sc->slice_ues[s * Us + k];
        double acc = 0.0;                                              /* :584-590 mean over deque */
        for (int i = 0; i < e->hist_len; i++) acc += deque_at(e, i)->sent[ue];
        double snt = (acc / (double)e->hist_len) * (double)sc->ue_pkt_size[ue]; /* :591 */
        if (np_isclose(avail[k], 0.0)) snt = 1.0;                      /* :592-594 */
        weights[k] = np_isclose(snt, 0.0) ? 2.0 * max_avail : avail[k] / snt; /* :595-602 */
    }
    double wsum = orc_np_sum(weights, n, 1);
    if (wsum != 0.0) {                                                 /* :603-608 */
        for (int k = 0; k < n; k++) vals[k] = (double)n_rbs * weights[k] / wsum;
        orc_round_int_equal_sum(vals, n, n_rbs, counts);
    } else {
        round_robin(e, s, n_rbs, 0, counts);                           /* :609-617 */
    }
}

/* agents/common.py:639-701 max_throughput */
static void max_throughput(const orc_env *e, int s, int64_t n_rbs, int64_t *counts)
{
    int Us = e->cfg.max_ues_slice, n = e->sc->slice_nues[s];
    double *avail = e->scratch, *vals = e->scratch + 2 * Us;
    throughput_available(e, s, n_rbs, avail);
    double asum = orc_np_sum(avail, n, 1);
    if (asum != 0.0) {
        for (int k = 0; k < n; k++) vals[k] = (double)n_rbs * avail[k] / asum;
        orc_round_int_equal_sum(vals, n, n_rbs, counts);
    } else {
        round_robin(e, s, n_rbs, 0, counts);
    }
}

/* agents/ib_sched.py:223-349 IBSched.action_format */
void orc_action_format(orc_env *e, const double *inter_scores, const int32_t *intra_choice,
                       int32_t *rb_start, int32_t *rb_count, uint8_t *dense)
{
    const orc_scenario *sc = e->sc;
    int S = e->cfg.n_slices, U = e->cfg.n_ues, R = e->cfg.n_rbs, Us = e->cfg.max_ues_slice;
    for (int u = 0; u < U; u++) { rb_start[u] = 0; rb_count[u] = 0; }
    if (dense) memset(dense, 0, (size_t)U * R);
    int any_active = 0;
    for (int s = 0; s < S; s++) any_active += sc->slice_active[s];
    if (any_active == 0) return;                                       /* :240-245 */
    double action[256], assoc[256];
    int64_t rbs_per_slice[256], counts[1024];
    for (int i = 0; i < S; i++) {
        action[i] = inter_scores[sc->sorted_slices[i]];                /* :247 (not the inverse) */
        assoc[i] = (double)sc->slice_active[i];
    }
    for (int i = 0; i < S; i++) if (sc->slice_active[i] == 0) action[i] = -1.0; /* :248-255 */
    int64_t n_rbgs = (int64_t)floor((double)R / (double)e->cfg.rbs_per_rbg);    /* :261-263 */
    orc_scores_to_rbs(action, S, n_rbgs, assoc, rbs_per_slice);
    for (int i = 0; i < S; i++) rbs_per_slice[i] *= e->cfg.rbs_per_rbg;        /* :268 */
    int64_t rb_idx = 0;
    for (int s = 0; s < S; s++) {                                      /* :272-344 */
        int n = sc->slice_nues[s];
        if (n > 0) {
            switch (intra_choice[s]) {
            case ORC_INTRA_RR: round_robin(e, s, rbs_per_slice[s], 1, counts); break;
            case ORC_INTRA_PF: proportional_fairness(e, s, rbs_per_slice[s], counts); break;
            default:           max_throughput(e, s, rbs_per_slice[s], counts); break;
            }
            /* agents/common.py:464-478 distribute_rbs_ues: contiguous ranges, UE order */
            int64_t pos = rb_idx;
            for (int k = 0; k < n; k++) {
                int ue = sc->slice_ues[s * Us + k];
                rb_start[ue] = (int32_t)pos; rb_count[ue] = (int32_t)counts[k];
                if (dense) for (int64_t r = pos; r < pos + counts[k]; r++) dense[(size_t)ue * R + r] = 1;
                pos += counts[k];
            }
        }
        rb_idx += rbs_per_slice[s];   /* rb_idx = sum(rbs_per_slice[:slice_idx]) common.py:471 */
    }
}

/* ------------------------------------------------------------------------------ */
/* CommunicationEnv.step / reset (PARITY UNPINNED; order per SURVEY.md 3.1)        */
/* ------------------------------------------------------------------------------ */

void orc_env_core_step(orc_env *e, const uint8_t *dense, const float *se_tile, const double *traffic_bits)
{
    const orc_scenario *sc = e->sc;
    int U = e->cfg.n_ues, R = e->cfg.n_rbs, L = e->cfg.max_age_cap + 1;
    double bw_per_rb = e->cfg.bandwidth_hz / (double)R;
    raw_rec *rec = deque_appendleft(e);        /* filled below, then observed */
    double *row = e->scratch;
    for (int u = 0; u < U; u++) {
        /* UEs.get_pkt_throughputs: floor(sum_r sched*SE * BW/R / pkt_size) -- the sum is scaled (the build's normative choice);
         * scale_per_element: floor(sum_r (sched*SE * BW/R) / pkt_size) -- every product is rounded before it is added, as
         * np.sum(sched * se * (BW / R)) would */
        double cnt = 0.0;
        for (int r = 0; r < R; r++) {
            int on = dense[(size_t)u * R + r] != 0;
            double v = (double)se_tile[(size_t)u * R + r];
            if (e->scale_per_element) v = v * bw_per_rb;
            row[r] = on ? v : 0.0;
            cnt += on;
        }
        double bits = orc_np_sum(row, R, 1);
        if (!e->scale_per_element) bits = bits * bw_per_rb;
        double pkt_size = (double)sc->ue_pkt_size[u];
        int64_t pkt_thr = (int64_t)floor(bits / pkt_size);
        int64_t pkt_in = (int64_t)floor(traffic_bits[u] / pkt_size);   /* UEs.get_pkt_incoming */
        int64_t *hist = &e->buf[(size_t)u * L];
        int max_age = sc->ue_max_age[u];
        int64_t max_pkts = sc->ue_max_pkts[u];
        int64_t dropped = buffer_receive(hist, max_age, max_pkts, pkt_in);
        int64_t sent = buffer_send(hist, max_age, pkt_thr);
        int64_t total = 0, age_sum = 0;
        for (int a = 0; a <= max_age; a++) { total += hist[a]; age_sum += (int64_t)a * hist[a]; }
        e->pkt_incoming[u] = (double)pkt_in;
        e->pkt_throughputs[u] = (double)pkt_thr;
        rec->sent[u] = (double)sent;
        rec->dropped[u] = (double)dropped;
        rec->occ[u] = (double)total / (double)max_pkts;               /* Buffer.get_buffer_occupancy */
        rec->lat[u] = total > 0 ? (double)age_sum / (double)total : 0.0; /* Buffer.get_avg_delay */
        rec->rowsum[u] = cnt;
    }
    fill_se_mean(e, se_tile, rec->se_mean);
    e->step_number += 1;
    obs_space_format(e);
    calculate_reward(e);
}

void orc_env_step(orc_env *e, const double *inter_scores, const int32_t *intra_choice,
                  const float *se_tile, const double *traffic_bits)
{
    int U = e->cfg.n_ues, R = e->cfg.n_rbs;
    int32_t *rb_start = (int32_t *)malloc(sizeof(int32_t) * U);
    int32_t *rb_count = (int32_t *)malloc(sizeof(int32_t) * U);
    uint8_t *dense = (uint8_t *)malloc((size_t)U * R);
    orc_action_format(e, inter_scores, intra_choice, rb_start, rb_count, dense);
    orc_env_core_step(e, dense, se_tile, traffic_bits);
    free(rb_start); free(rb_count); free(dense);
}

void orc_env_reset(orc_env *e, const float *se_tile)
{
    int U = e->cfg.n_ues, L = e->cfg.max_age_cap + 1;
    memset(e->buf, 0, sizeof(int64_t) * (size_t)U * L);   /* fresh UEs / Buffer objects */
    e->step_number = 0;
    raw_rec *rec = deque_appendleft(e);                    /* deque survives (ib_sched.py:51) */
    for (int u = 0; u < U; u++) {
        e->pkt_incoming[u] = 0.0; e->pkt_throughputs[u] = 0.0;
        rec->sent[u] = 0.0; rec->dropped[u] = 0.0; rec->occ[u] = 0.0; rec->lat[u] = 0.0;
        rec->rowsum[u] = 0.0;
    }
    fill_se_mean(e, se_tile, rec->se_mean);
    obs_space_format(e);
    calculate_reward(e);
}

void orc_agent_observe(orc_env *e, const double *sent, const double *dropped, const double *occ,
                       const double *lat, const float *se_tile, const double *sched_rowsum)
{
    int U = e->cfg.n_ues;
    raw_rec *rec = deque_appendleft(e);
    memcpy(rec->sent, sent, sizeof(double) * U);
    memcpy(rec->dropped, dropped, sizeof(double) * U);
    memcpy(rec->occ, occ, sizeof(double) * U);
    memcpy(rec->lat, lat, sizeof(double) * U);
    memcpy(rec->rowsum, sched_rowsum, sizeof(double) * U);
    fill_se_mean(e, se_tile, rec->se_mean);
    obs_space_format(e);
    calculate_reward(e);
}

/* ------------------------------------------------------------------------------ */
/* baseline policies                                                               */
/* ------------------------------------------------------------------------------ */

/* agents/marr.py:40-47 MARR.step */
void orc_policy_marr(const orc_env *e, double *inter_scores)
{
    for (int s = 0; s < e->cfg.n_slices; s++) inter_scores[s] = e->sc->slice_nues[s] > 0 ? 1.0 : -1.0;
}

/* agents/mapf.py:41-111 MAPF.step */
void orc_policy_mapf(const orc_env *e, double *inter_scores)
{
    const orc_scenario *sc = e->sc;
    int S = e->cfg.n_slices, Us = e->cfg.max_ues_slice;
    const raw_rec *r0 = deque_at(e, 0);
    double *occ_mb = e->scratch, *thr_mb = e->scratch + S, *w = e->scratch + 2 * S, *tmp = e->scratch + 3 * S;
    for (int s = 0; s < S; s++) { occ_mb[s] = 0.0; thr_mb[s] = 0.0; }
    for (int s = 0; s < S; s++) {
        if (!sc->slice_active[s]) continue;                            /* :50-53 */
        int n = sc->slice_nues[s];
        double pkt = (double)sc->slice_message_size[s], bmax = (double)sc->slice_buffer_size[s];
        for (int k = 0; k < n; k++) tmp[k] = r0->occ[sc->slice_ues[s * Us + k]];
        occ_mb[s] = ((np_mean(tmp, n, 1) * bmax) * pkt) / 1e6;         /* :63-74 */
        for (int k = 0; k < n; k++) {                                  /* :75-90 */
            int ue = sc->slice_ues[s * Us + k];
            double acc = 0.0;
            for (int i = 0; i < e->hist_len; i++) acc += deque_at(e, i)->sent[ue];
            tmp[k] = acc / (double)e->hist_len;
        }
        thr_mb[s] = (np_mean(tmp, n, 1) * pkt) / 1e6;
    }
    double mx = occ_mb[0];
    for (int s = 1; s < S; s++) if (occ_mb[s] > mx) mx = occ_mb[s];
    for (int s = 0; s < S; s++) {                                      /* :91-104 */
        w[s] = np_isclose(thr_mb[s], 0.0) ? 2.0 * mx : occ_mb[s] / thr_mb[s];
        if (!sc->slice_active[s]) w[s] = 0.0;
    }
    double ws = orc_np_sum(w, S, 1);                                   /* :105-109 */
    for (int s = 0; s < S; s++) inter_scores[s] = (ws > 0.0 ? w[s] / ws : 2.0) - 1.0;
}

/* ------------------------------------------------------------------------------ */
/* alternative heads: SchedTWC / SchedColORAN (agents/sched_twc.py, sched_colran.py) */
/* ------------------------------------------------------------------------------ */

void orc_env_set_pkt_throughputs(orc_env *e, const double *pkt_throughputs)
{
    memcpy(e->pkt_throughputs, pkt_throughputs, sizeof(double) * e->cfg.n_ues);
}

/* Observation of SchedTWC.obs_space_format (sched_twc.py:165-346; SchedColORAN's is the same): slices
 * in index order (enable_sort_slices=False, :80), metric-major:
 *   [requirements (reliability, latency, throughput) x S | mean SE | served Mbps | effective Mbps |
 *    buffer occupancy | buffer latency | packet loss rate | requested Mbps] = 10*S values.
 * Rewards: SchedTWC.calculate_reward (:348-413) and SchedColORAN.calculate_reward
 * (sched_colran.py:348-419).  usecase[s]: bit 0 = eMBB, bit 1 = URLLC (the slice-name table at
 * sched_colran.py:356-367).  The head's deque holds every TTI twice (see view_len). */
void orc_env_get_heads(const orc_env *e, const int32_t *usecase, double *obs, double *reward_twc,
                       double *reward_colran)
{
    const orc_scenario *sc = e->sc;
    int S = e->cfg.n_slices, Us = e->cfg.max_ues_slice;
    const raw_rec *r0 = deque_at(e, 0);
    double *tmp = e->scratch;
    double *drift = (double *)malloc(sizeof(double) * (size_t)S * Us * 3);
    intent_drift_into(e, 2, drift);
    double *req = obs, *se = obs + 3 * S, *thr = se + S, *eff = thr + S, *occ = eff + S, *lat = occ + S,
           *loss = lat + S, *rqt = loss + S;
    for (int s = 0; s < S; s++) {
        int n = sc->slice_nues[s];
        req[3 * s] = req[3 * s + 1] = req[3 * s + 2] = 0.0;
        if (n != 0 && sc->slice_has_req[s]) {                           /* :216-226 */
            for (int p = 0; p < sc->slice_nparams[s]; p++) {
                int m = sc->param_metric[s * 3 + p];
                double v = sc->param_value[s * 3 + p];
                if (m == ORC_METRIC_RELIABILITY) req[3 * s] = v;
                else if (m == ORC_METRIC_LATENCY) req[3 * s + 1] = v;
                else req[3 * s + 2] = v;
            }
        }
        double pkt_size = n != 0 ? (double)sc->slice_message_size[s] : 0.0;   /* :231-237 */
        if (n == 0) {                       /* np.mean(np.array([0])) everywhere */
            se[s] = thr[s] = eff[s] = occ[s] = lat[s] = loss[s] = 0.0;
        } else {
            for (int k = 0; k < n; k++) tmp[k] = r0->se_mean[sc->slice_ues[s * Us + k]];
            se[s] = np_mean(tmp, n, 1);                                          /* :240-252 */
            for (int k = 0; k < n; k++) tmp[k] = e->pkt_throughputs[sc->slice_ues[s * Us + k]] * pkt_size / 1e6;
            thr[s] = np_mean(tmp, n, 1);                                         /* :255-266 */
            for (int k = 0; k < n; k++) tmp[k] = r0->sent[sc->slice_ues[s * Us + k]] * pkt_size / 1e6;
            eff[s] = np_mean(tmp, n, 1);                                         /* :269-280 */
            for (int k = 0; k < n; k++) tmp[k] = r0->occ[sc->slice_ues[s * Us + k]];
            occ[s] = np_mean(tmp, n, 1);                                         /* :283-293 */
            for (int k = 0; k < n; k++) tmp[k] = r0->lat[sc->slice_ues[s * Us + k]];
            lat[s] = np_mean(tmp, n, 1);                                         /* :296-306 */
            for (int k = 0; k < n; k++) tmp[k] = get_metric_value(e, 2, ORC_METRIC_RELIABILITY, s, sc->slice_ues[s * Us + k]);
            loss[s] = np_mean(tmp, n, 1);                                        /* :309-322 */
        }
        rqt[s] = np_isclose((double)sc->slice_active[s], 1.0) ? sc->slice_traffic[s] : 0.0;   /* :325-337 */
    }
    /* SchedTWC.calculate_reward: negative slice drifts, weighted 2 for priority slices */
    {
        double vals[48], wts[48], terms[48];
        int m = 0;
        for (int s = 0; s < S; s++) {
            int n = sc->slice_nues[s];
            if (n == 0) continue;                                                /* :364-365 */
            double sv[3] = {-2.0, -2.0, -2.0};                                   /* calculate_slice_ue_obs */
            if (sc->slice_has_req[s])
                for (int p = 0; p < sc->slice_nparams[s]; p++) {
                    int mt = sc->param_metric[s * 3 + p];
                    sv[mt] = np_mean(&drift[((size_t)s * Us) * 3 + mt], n, 3);
                }
            double w = sc->slice_priority[s] != 0.0 ? 2.0 : 1.0;                 /* :382-391 */
            for (int k = 0; k < 3; k++) {
                if (np_isclose(sv[k], -2.0)) continue;                           /* :376-378 */
                vals[m] = sv[k] > 0.0 ? 0.0 : sv[k];                             /* :395-397 */
                wts[m] = w; m++;
            }
        }
        int q = 0; double nw[48];
        for (int i = 0; i < m; i++) if (vals[i] < 0.0) { terms[q] = vals[i]; nw[q] = wts[i]; q++; }
        double wsum = orc_np_sum(nw, q, 1);
        if (np_isclose(wsum, 0.0)) *reward_twc = 0.0;                            /* :401-410 */
        else {
            for (int i = 0; i < q; i++) terms[i] = terms[i] * nw[i] / wsum;
            *reward_twc = orc_np_sum(terms, q, 1);
        }
    }
    /* SchedColORAN.calculate_reward: throughput of eMBB slices up, buffered Mbit of URLLC slices down */
    {
        double r = 0.0;
        for (int s = 0; s < S; s++) {
            if (sc->slice_active[s] == 0) continue;                              /* active_slice_idx */
            int n = sc->slice_nues[s];
            if (n == 0) continue;
            double pkt_size = (double)sc->slice_message_size[s];
            if (usecase[s] & 1) {
                for (int k = 0; k < n; k++) tmp[k] = e->pkt_throughputs[sc->slice_ues[s * Us + k]];
                double st = (np_mean(tmp, n, 1) * pkt_size) / 1e6;
                r += st / 200.0;
            }
            if (usecase[s] & 2) {
                for (int k = 0; k < n; k++) tmp[k] = r0->occ[sc->slice_ues[s * Us + k]];
                double sb = (np_mean(tmp, n, 1) * (double)sc->slice_buffer_size[s]) * pkt_size / 1e6;
                r -= sb / 2000.0;
            }
        }
        *reward_colran = r;
    }
    free(drift);
}

/* ------------------------------------------------------------------------------ */
/* read-back                                                                       */
/* ------------------------------------------------------------------------------ */

void orc_env_get_raw(const orc_env *e, double *pkt_incoming, double *pkt_throughputs,
                     double *sent, double *dropped, double *occ, double *lat)
{
    int U = e->cfg.n_ues;
    const raw_rec *r0 = deque_at(e, 0);
    if (pkt_incoming) memcpy(pkt_incoming, e->pkt_incoming, sizeof(double) * U);
    if (pkt_throughputs) memcpy(pkt_throughputs, e->pkt_throughputs, sizeof(double) * U);
    if (sent) memcpy(sent, r0->sent, sizeof(double) * U);
    if (dropped) memcpy(dropped, r0->dropped, sizeof(double) * U);
    if (occ) memcpy(occ, r0->occ, sizeof(double) * U);
    if (lat) memcpy(lat, r0->lat, sizeof(double) * U);
}

void orc_env_get_obs(const orc_env *e, double *obs_inter, int8_t *mask_inter,
                     double *obs_intra, int8_t *mask_intra, double *reward)
{
    int S = e->cfg.n_slices, Us = e->cfg.max_ues_slice;
    if (obs_inter) memcpy(obs_inter, e->obs_inter, sizeof(double) * S * 10);
    if (mask_inter) memcpy(mask_inter, e->mask_inter, S);
    if (obs_intra) memcpy(obs_intra, e->obs_intra, sizeof(double) * S * (2 * Us + 9));
    if (mask_intra) memcpy(mask_intra, e->mask_intra, (size_t)S * Us);
    if (reward) memcpy(reward, e->reward, sizeof(double) * (S + 1));
}

void orc_env_get_drift(const orc_env *e, double *drift)
{
    memcpy(drift, e->drift, sizeof(double) * (size_t)e->cfg.n_slices * e->cfg.max_ues_slice * 3);
}

int orc_env_step_number(const orc_env *e) { return e->step_number; }
int orc_env_hist_len(const orc_env *e) { return e->hist_len; }

void orc_env_get_buffer(const orc_env *e, int u, int64_t *hist_out)
{
    int L = e->cfg.max_age_cap + 1;
    memcpy(hist_out, &e->buf[(size_t)u * L], sizeof(int64_t) * L);
}

/* ------------------------------------------------------------------------------ */
/* batch driver (CPU baseline timing and bulk parity checks)                       */
/* ------------------------------------------------------------------------------ */

/* Step n independent envs once.  policy: 0 = scores given, 1 = MARR, 2 = MAPF.
 * se_pool + tile_index[i]*U*R is env i's SE tile; traffic is [n*U] bits.
 * OpenMP over envs when built with -fopenmp (n_threads <= 0: runtime default). */
void orc_batch_step(orc_env **envs, int n, int policy, const double *scores, const int32_t *intra,
                    const float *se_pool, const int64_t *tile_index, const double *traffic,
                    int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int i = 0; i < n; i++) {
        orc_env *e = envs[i];
        int S = e->cfg.n_slices, U = e->cfg.n_ues, R = e->cfg.n_rbs;
        double sc_local[256];
        const double *sc = scores ? scores + (size_t)i * S : sc_local;
        if (policy == 1) { orc_policy_marr(e, sc_local); sc = sc_local; }
        else if (policy == 2) { orc_policy_mapf(e, sc_local); sc = sc_local; }
        orc_env_step(e, sc, intra + (size_t)i * S, se_pool + (size_t)tile_index[i] * U * R,
                     traffic + (size_t)i * U);
    }
}

void orc_batch_reset(orc_env **envs, int n, const float *se_pool, const int64_t *tile_index,
                     int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int i = 0; i < n; i++) {
        orc_env *e = envs[i];
        orc_env_reset(e, se_pool + (size_t)tile_index[i] * e->cfg.n_ues * e->cfg.n_rbs);
    }
}
