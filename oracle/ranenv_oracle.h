/*
 * ranenv_oracle.h -- CPU restatement of the per-TTI RAN-slicing env step.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / reported CPU baseline.  The product path is the HIP
 * library declared in include/ranenv.h.
 *
 * What is restated (citations are paths under the upstream repository
 * lasseufpa/intent_radio_sched_multi_slice):
 *   agents/common.py      round_int_equal_sum, scores_to_rbs, distribute_rbs_ues,
 *                         round_robin, proportional_fairness, max_throughput,
 *                         get_metric_value, intent_drift_calc,
 *                         calculate_slice_ue_obs, calculate_reward_no_mask
 *   agents/ib_sched.py    IBSched.obs_space_format / calculate_reward /
 *                         action_format / sort_slices / unsort_slices
 *   agents/marr.py, agents/mapf.py   MARR.step, MAPF.step
 *   sixg_radio_mgmt       UEs.step, Buffer.receive_packets / send_packets,
 *                         CommunicationEnv.step / reset  -- this module is an
 *                         un-vendored git submodule (.gitmodules:1-3, pinned
 *                         version unknown), so these follow the constraints its
 *                         call sites impose.
 *
 * PARITY STATUS
 *   agent side (common.py / ib_sched.py / marr.py / mapf.py): PINNED by golden
 *     vectors generated from the reference's own functions
 *     (tests/golden/gen_golden.py writes the .npz fixtures in tests/golden).
 *   env core (UEs / Buffer / step ordering): PARITY UNPINNED -- no reference
 *     source, test or fixture exists for it; this file is the normative spec.
 *
 * Numerics: all arithmetic is IEEE double in the same operation order numpy uses,
 * including numpy's pairwise summation (orc_np_sum), because integer RB
 * allocations depend on floor() of such sums.
 */
#ifndef RANENV_ORACLE_H
#define RANENV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_METRIC_THROUGHPUT = 0, ORC_METRIC_RELIABILITY = 1, ORC_METRIC_LATENCY = 2 };
/* expectation_params operators, associations/mult_slice.py:48-55 */
enum { ORC_OP_GE = 0, ORC_OP_LE = 1, ORC_OP_EQ = 2, ORC_OP_GT = 3, ORC_OP_LT = 4 };
enum { ORC_INTRA_RR = 0, ORC_INTRA_PF = 1, ORC_INTRA_MT = 2 };

typedef struct {
    int32_t n_slices;      /* S  max_number_slices                        */
    int32_t n_ues;         /* U  max_number_ues                           */
    int32_t n_rbs;         /* R  num_available_rbs[0]                     */
    int32_t rbs_per_rbg;   /* G  IBSched.rbs_per_rbg  (ib_sched.py:56)    */
    int32_t max_ues_slice; /* Us IBSched.max_number_ues_slice (:50)       */
    int32_t hist_depth;    /* max_obs_memory = 10 (ib_sched.py:49)        */
    int32_t max_age_cap;   /* largest buffer_latency any UE may get       */
    int32_t max_steps;     /* max_number_steps (env_config/mult_slice.yml:10) */
    double  bandwidth_hz;  /* comm_env.bandwidths[0]                      */
    double  overfulfill;   /* intent_overfulfillment_rate = 0.2 (:53)     */
    double  norm_traffic;  /* 120.0  ib_sched.py:166                      */
    double  norm_ues;      /* 5.0    ib_sched.py:167                      */
    double  norm_se;       /* 40.0   ib_sched.py:168                      */
} orc_cfg;

/* One scenario = association + slice_req of one episode, flattened. */
typedef struct {
    const int32_t *slice_active;         /* [S]    basestation_slice_assoc[0,s]         */
    const int32_t *slice_has_req;        /* [S]    slice_req[f"slice_{s}"] != {}        */
    const int32_t *slice_nues;           /* [S]    sum(slice_ue_assoc[s])               */
    const int32_t *slice_ues;            /* [S*Us] nonzero(slice_ue_assoc[s]) ascending, -1 pad */
    const double  *slice_priority;       /* [S]    slice_req[s]["priority"]             */
    const double  *slice_traffic;        /* [S]    ["ues"]["traffic"] (Mbps)            */
    const int32_t *slice_buffer_size;    /* [S]    ["ues"]["buffer_size"] (pkts)        */
    const int32_t *slice_buffer_latency; /* [S]    ["ues"]["buffer_latency"] (TTIs)     */
    const int32_t *slice_message_size;   /* [S]    ["ues"]["message_size"] (bits)       */
    const int32_t *slice_nparams;        /* [S]    len(["parameters"])  (0..3)          */
    const int32_t *param_metric;         /* [S*3]  ORC_METRIC_*                         */
    const int32_t *param_op;             /* [S*3]  ORC_OP_*                             */
    const double  *param_value;          /* [S*3]                                       */
    const int32_t *sorted_slices;        /* [S]    IBSched.sorted_slices                */
    const int32_t *ue_pkt_size;          /* [U]    ues.pkt_sizes                        */
    const int32_t *ue_max_pkts;          /* [U]    ues.max_buffer_pkts                  */
    const int32_t *ue_max_age;           /* [U]    ues.buffers[u].max_packets_age       */
} orc_scenario;

typedef struct orc_env orc_env;

/* ---- numpy arithmetic helpers ------------------------------------------------ */
double orc_np_sum(const double *a, int64_t n, int64_t stride);
void   orc_stable_argsort(const double *v, int n, int32_t *idx_out);

/* ---- agents/common.py, stateless pieces -------------------------------------- */
void orc_round_int_equal_sum(const double *v, int n, int64_t target, int64_t *out);
void orc_scores_to_rbs(const double *action, int n, int64_t total_rbs,
                       const double *association, int64_t *out);
/* IBSched.sort_slices (ib_sched.py:351-370), stable tie rule */
void orc_sort_slices(const int32_t *slice_nues, const double *slice_traffic,
                     const int32_t *slice_has_req, int n, int32_t *sorted_out);

/* ---- stateful env ------------------------------------------------------------ */
orc_env *orc_env_create(const orc_cfg *cfg);
void     orc_env_destroy(orc_env *e);
/* Forget everything, including the IBSched deque (a fresh process). */
void     orc_env_clear(orc_env *e);
/* RANENV_F_SCALE_PER_ELEMENT's convention for UEs.get_pkt_throughputs (default off: the sum is scaled by BW / R; on: every
 * element is scaled and rounded before it is added).  Survives orc_env_clear. */
void     orc_env_set_scale_per_element(orc_env *e, int on);
/* Install the scenario of the next episode (pointers must outlive its use). */
void     orc_env_set_scenario(orc_env *e, const orc_scenario *sc);

/* CommunicationEnv.reset: new UEs/buffers, step 0, returns the formatted initial obs
 * (zero metrics, SE tile of step 0).  The IBSched deque is NOT cleared
 * (ib_sched.py:51 has no reset hook). */
void orc_env_reset(orc_env *e, const float *se_tile /* U*R */);

/* IBSched.action_format (ib_sched.py:223-349).
 * inter_scores[S] is action["player_0"]; intra_choice[S] is action["player_{s+1}"]
 * (ORC_INTRA_*); writes per-UE contiguous RB range and (optionally) the dense mask. */
void orc_action_format(orc_env *e, const double *inter_scores, const int32_t *intra_choice,
                       int32_t *rb_start /* U */, int32_t *rb_count /* U */,
                       uint8_t *dense_mask_or_null /* U*R */);

/* UEs.step + raw observation + obs_space_format + calculate_reward with an
 * explicit dense scheduling mask (the compatibility-facade path). */
void orc_env_core_step(orc_env *e, const uint8_t *dense_mask /* U*R */,
                       const float *se_tile /* U*R */, const double *traffic_bits /* U */);

/* Whole CommunicationEnv.step: action_format -> UEs.step -> obs -> reward. */
void orc_env_step(orc_env *e, const double *inter_scores, const int32_t *intra_choice,
                  const float *se_tile, const double *traffic_bits);

/* Push an externally produced raw observation through obs_space_format +
 * calculate_reward only (golden tests of the agent side).  sched_rowsum[U] is
 * sum(sched_decision, axis=2)[0]. */
void orc_agent_observe(orc_env *e, const double *pkt_effective_thr, const double *dropped_pkts,
                       const double *buffer_occupancies, const double *buffer_latencies,
                       const float *se_tile, const double *sched_rowsum);

/* Alternative heads (SchedTWC / SchedColORAN): 10*S observation values and the two rewards computed
 * from the same history, with the heads' double-push deque (agents/sched_twc.py:165-413,
 * agents/sched_colran.py:348-419).  usecase[S]: bit 0 eMBB, bit 1 URLLC. */
void orc_env_set_pkt_throughputs(orc_env *e, const double *pkt_throughputs /* U */);
void orc_env_get_heads(const orc_env *e, const int32_t *usecase, double *obs /* 10*S */,
                       double *reward_twc, double *reward_colran);

/* Baseline policies: MARR.step (marr.py:40-47), MAPF.step (mapf.py:41-111). */
void orc_policy_marr(const orc_env *e, double *inter_scores /* S */);
void orc_policy_mapf(const orc_env *e, double *inter_scores /* S */);

/* ---- read-back ---------------------------------------------------------------- */
/* raw metrics of the last step, all [U] doubles like the reference's float arrays */
void orc_env_get_raw(const orc_env *e, double *pkt_incoming, double *pkt_throughputs,
                     double *pkt_effective_thr, double *dropped_pkts,
                     double *buffer_occupancies, double *buffer_latencies);
/* formatted observation of the last obs_space_format:
 *   obs_inter [S*10] in sorted-slice order (player_0 "observations")
 *   mask_inter[S]    player_0 "action_mask"
 *   obs_intra [S*(2*Us+9)] row s = player_{s+1} "observations"
 *   mask_intra[S*Us]
 *   reward    [S+1]  reward[0]=player_0, reward[s+1]=player_{s+1}            */
void orc_env_get_obs(const orc_env *e, double *obs_inter, int8_t *mask_inter,
                     double *obs_intra, int8_t *mask_intra, double *reward);
void orc_env_get_drift(const orc_env *e, double *drift /* S*Us*3 */);
int  orc_env_step_number(const orc_env *e);
int  orc_env_hist_len(const orc_env *e);
/* age histogram of UE u (Buffer.buffer), length max_age_cap+1 */
void orc_env_get_buffer(const orc_env *e, int u, int64_t *hist_out);

/* ---- batch driver (OpenMP over envs) ----------------------------------------- */
void orc_batch_step(orc_env **envs, int n, int policy, const double *scores, const int32_t *intra,
                    const float *se_pool, const int64_t *tile_index, const double *traffic,
                    int n_threads);
void orc_batch_reset(orc_env **envs, int n, const float *se_pool, const int64_t *tile_index,
                     int n_threads);

#ifdef __cplusplus
}
#endif
#endif
