/*
 * ranenv.h -- C ABI of the MI355X-native batched RAN-slicing environment step.
 *
 * One handle = B independent environments resident in the HBM of one GPU.  The library
 * (libranenv_hip.so, built from intent_radio_sched_multi_slice_amd/csrc/ranenv_*.{hip,hpp,cpp} for
 * gfx950) replaces, for the per-TTI hot path only, what the reference does in Python:
 *
 *   reference interface (lasseufpa/intent_radio_sched_multi_slice)      entry point here
 *   -----------------------------------------------------------------   -------------------
 *   MARLCommEnv(...) construction                   simu.py:348-362     ranenv_create
 *   association.step / update_ues  associations/mult_slice.py:350-488   ranenv_load_scenarios,
 *                                                                       ranenv_set_episodes
 *   channel.step  (SE tile per TTI)     channels/quadriga.py:38-76      ranenv_bind_se_pool, ranenv_bind_se_pool_quad
 *                                                                       (+ ranenv_se_from_power, ranenv_se_retile_quad)
 *   traffic.step  (offered bits)        traffics/mult_slice.py:15-34    ranenv_bind_traffic_pool
 *   env.reset(seed, options)                        simu.py:547-554     ranenv_reset
 *   env.step(action):                               simu.py:559         ranenv_step
 *     IBSched.action_format            agents/ib_sched.py:223-349
 *     UEs.step / Buffer (sixg_radio_mgmt, un-vendored submodule)
 *     IBSched.obs_space_format         agents/ib_sched.py:63-204
 *     IBSched.calculate_reward         agents/ib_sched.py:206-221
 *   env.step with a caller-made sched_decision (any agent's
 *     action_format callback, simu.py:405-411)                          ranenv_step_dense
 *   MARR.step / MAPF.step   agents/marr.py:40-47, agents/mapf.py:41-111 ranenv_set_policy
 *   raw observation dict fields         agents/ib_sched.py:78-181       ranenv_get_views
 *
 * Conventions
 *   - every function returns 0 on success or a negative RANENV_E_* code; the message is
 *     available from ranenv_last_error(handle) (handle may be NULL for create failures).
 *     No exception crosses this boundary.
 *   - "dev" pointers are device pointers owned by the caller (e.g. torch tensors); "host"
 *     pointers are ordinary host memory, copied before the call returns.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls on one
 *     handle must be serialised by the caller; work is enqueued, not synchronised.
 *   - there is no CPU fallback: if no gfx950 device/kernel image is usable the calls fail.
 *   - sizes supported by this build: S <= 16, max_ues_slice <= 16, U <= 256, R <= 512,
 *     hist_depth <= 64; packet counts must stay below 2^31 (checked when scenarios are loaded).
 *   - SE tiles are RB-major: element (rb r, ue u) of a tile at offset r*U + u (the order of the reference's .mat); a POOL may also be
 *     bound RB-quad-major, element at ((r/4)*U + u)*4 + r%4 (ranenv_bind_se_pool_quad: 16-byte loads, same results).
 */
#ifndef RANENV_H
#define RANENV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RANENV_ABI_VERSION 9

enum {
    RANENV_OK = 0,
    RANENV_E_INVALID = -1,   /* bad argument / unsupported size               */
    RANENV_E_HIP = -2,       /* a HIP runtime call failed                     */
    RANENV_E_STATE = -3,     /* call order (e.g. step before scenarios/pools) */
    RANENV_E_NOMEM = -4
};

enum { RANENV_POLICY_EXTERNAL = 0, RANENV_POLICY_MARR = 1, RANENV_POLICY_MAPF = 2 };
enum { RANENV_INTRA_RR = 0, RANENV_INTRA_PF = 1, RANENV_INTRA_MT = 2, RANENV_INTRA_PER_SLICE = 255 };
enum { RANENV_SE_STREAM = 0, RANENV_SE_GATHER = 1 };
enum { RANENV_METRIC_THROUGHPUT = 0, RANENV_METRIC_RELIABILITY = 1, RANENV_METRIC_LATENCY = 2 };
enum { RANENV_OP_GE = 0, RANENV_OP_LE = 1, RANENV_OP_EQ = 2, RANENV_OP_GT = 3, RANENV_OP_LT = 4 };

/* flags */
#define RANENV_F_CLEAR_HISTORY_ON_RESET 0x1 /* default off: the reference never clears the
                                               10-TTI window (agents/ib_sched.py:51)      */
#define RANENV_F_NO_RAW_OUTPUT          0x2 /* skip pkt_incoming / pkt_throughputs stores  */
#define RANENV_F_SYNC_CHECK             0x4 /* debug: reset / step / step_dense wait for their
                                               kernels and return RANENV_E_HIP on an asynchronous
                                               fault instead of leaving it to a later call     */
#define RANENV_F_SCALE_PER_ELEMENT      0x8 /* default off.  How UEs.get_pkt_throughputs rounds (the env core sixg_radio_mgmt is absent
                                               from the reference snapshot: parity unpinned).  Off:
                                               floor(np.sum(sched * se) * (BW / R) / pkt_size) -- the sum is scaled.  On:
                                               floor(np.sum(sched * se * (BW / R)) / pkt_size) -- every element is scaled (and rounded)
                                               before it is added, in numpy's pairwise order either way.  The two differ by one packet
                                               on rare inputs.  With the flag every step / dense launch runs the lean build compiled
                                               for that convention: no mixed blocks, packed waves, small-batch / whole-row builds or
                                               persistent launches (the options stay readable and have no effect).               */

typedef struct ranenv *ranenv_handle;

typedef struct {
    int32_t abi_version;   /* RANENV_ABI_VERSION                                        */
    int32_t device;        /* HIP device ordinal                                        */
    int32_t batch;         /* B   environments on this device                           */
    int32_t n_slices;      /* S   max_number_slices          (env_config/mult_slice.yml:13) */
    int32_t n_ues;         /* U   max_number_ues             (:14)                      */
    int32_t n_rbs;         /* R   num_available_rbs[0]       (:5)                       */
    int32_t rbs_per_rbg;   /* G   IBSched.rbs_per_rbg        (agents/ib_sched.py:56)    */
    int32_t max_ues_slice; /* Us  IBSched.max_number_ues_slice (:50)                    */
    int32_t hist_depth;    /* max_obs_memory = 10            (:49)                      */
    int32_t max_age_cap;   /* largest buffer_latency a scenario may carry (<= 400 in
                              associations/mult_slice.py:225)                           */
    int32_t max_steps;     /* max_number_steps               (env_config/mult_slice.yml:10) */
    int32_t n_scenarios;   /* rows of the scenario pool                                 */
    int32_t flags;         /* RANENV_F_*                                                */
    int32_t reserved;
    double  bandwidth_hz;  /* bandwidths[0]                  (env_config/mult_slice.yml:2)  */
    double  overfulfill;   /* intent_overfulfillment_rate = 0.2 (agents/ib_sched.py:53) */
    double  norm_traffic;  /* 120.0 (agents/ib_sched.py:166)                            */
    double  norm_ues;      /* 5.0   (:167)                                              */
    double  norm_se;       /* 40.0  (:168)                                              */
} ranenv_config;

/* Scenario pool rows, host pointers, row-major [count][...]; same content as the
 * reference's (basestation_slice_assoc, slice_ue_assoc, slice_req) triple and the per-UE
 * buffer parameters pushed by update_ues (associations/mult_slice.py:468-488). */
typedef struct {
    const int32_t *slice_active;         /* [n][S]                                  */
    const int32_t *slice_has_req;        /* [n][S]                                  */
    const int32_t *slice_nues;           /* [n][S]                                  */
    const int32_t *slice_ues;            /* [n][S][Us] ascending UE ids, -1 padded  */
    const double  *slice_priority;       /* [n][S]                                  */
    const double  *slice_traffic;        /* [n][S] Mbps                             */
    const int32_t *slice_buffer_size;    /* [n][S] packets                          */
    const int32_t *slice_buffer_latency; /* [n][S] TTIs                             */
    const int32_t *slice_message_size;   /* [n][S] bits                             */
    const int32_t *slice_nparams;        /* [n][S] 0..3                             */
    const int32_t *param_metric;         /* [n][S][3] RANENV_METRIC_*               */
    const int32_t *param_op;             /* [n][S][3] RANENV_OP_*                   */
    const double  *param_value;          /* [n][S][3]                               */
    const int32_t *sorted_slices;        /* [n][S] IBSched.sorted_slices            */
    const int32_t *ue_slice;             /* [n][U] -1 = idle                        */
    const int32_t *ue_pos;               /* [n][U] position in its slice            */
    const int32_t *ue_pkt_size;          /* [n][U]                                  */
    const int32_t *ue_max_pkts;          /* [n][U]                                  */
    const int32_t *ue_max_age;           /* [n][U]                                  */
} ranenv_scenario_tables;

/* Which scenario / channel trace / traffic trace an env replays during its episode
 * (QuadrigaChannel.choose_episode channels/quadriga.py:78-87,
 *  MultSliceAssociation.choose_episode associations/mult_slice.py:444-452).
 * Tile used at TTI t: se_base + (se_offset + t) % se_len; same for traffic rows. */
typedef struct {
    int32_t scenario;
    int32_t se_len;
    int64_t se_base;
    int32_t se_offset;
    int32_t trf_len;
    int64_t trf_base;
    int32_t trf_offset;
    int32_t reserved;
} ranenv_episode;

/* Device-side views for raw-observation fields and state (all owned by the handle). */
typedef struct {
    int32_t *pkt_incoming;      /* [B][U] floor(traffic / pkt_size)                        */
    int32_t *pkt_throughputs;   /* [B][U] capacity in packets                              */
    int32_t *pkt_effective_thr; /* [B][U] packets sent this TTI                            */
    int32_t *dropped_pkts;      /* [B][U]                                                  */
    int32_t *queue_pkts;        /* [B][U] buffer_occupancies = queue_pkts / max_buffer_pkts */
    int64_t *queue_age_sum;     /* [B][U] buffer_latencies  = queue_age_sum / queue_pkts   */
    int32_t *rb_start;          /* [B][U] sched_decision[u] is ones on [rb_start, +rb_count) */
    int32_t *rb_count;          /* [B][U]                                                  */
    double  *se_mean;           /* [B][U] mean SE over RBs of the last tile (of UEs in a slice; for a UE outside every
                                   slice it is that of the last reset / full-width step: see "compact steps" below) */
    int64_t *win_sent;          /* [B][U] sum of pkt_effective_thr over the <=10-TTI window */
    int64_t *win_dropped;       /* [B][U]                                                  */
    int32_t *step_number;       /* [B]                                                     */
    int32_t *hist_len;          /* [B]    len(IBSched.last_unformatted_obs)                */
    int8_t  *mask_inter;        /* [B][S]     player_0 action_mask                         */
    int8_t  *mask_intra;        /* [B][S][Us] player_{s+1} action_mask                     */
    double  *policy_scores;     /* [B][S] inter-slice scores used by the last step         */
    int32_t *episode_number;    /* [B]    episode an env is playing (auto-reset; traffic generator key) */
    int32_t *episodes;          /* [B][10] the ranenv_episode descriptors as they are on the device, viewed as
                                   int32 words (word 0 = scenario): auto-reset rewrites them without the host   */
} ranenv_views;

const char *ranenv_last_error(ranenv_handle h);
int ranenv_abi_version(void);

int ranenv_create(const ranenv_config *cfg, ranenv_handle *out);
int ranenv_destroy(ranenv_handle h);

/* Copy `count` scenario rows (host) into the pool at rows [first, first+count). */
int ranenv_load_scenarios(ranenv_handle h, int32_t first, int32_t count,
                          const ranenv_scenario_tables *host_tables, void *stream);

/* SE pool: float32 tiles of R*U values, RB-major (value of RB r, UE u at r*U + u), tile i at
 * dev + i*tile_stride floats; tile_stride >= U*R. */
int ranenv_bind_se_pool(ranenv_handle h, const float *dev_pool, int64_t n_tiles, int64_t tile_stride);
/* The same pool in RB-QUAD-MAJOR order: four consecutive RBs of a UE side by side, float32 [tile][ceil(R/4)][U][4] (value of RB r, UE u at
 * ((r / 4) * U + u) * 4 + r % 4; zeros behind RB R-1), tile i at dev + i*tile_stride floats, tile_stride >= ceil(R/4)*U*4 and a multiple
 * of 4, dev 16-byte aligned.  What replays pooled tiles (reset / step / step_range / step_part / rollout / step_dense without explicit
 * tiles / the sidecar build of ranenv_set_se_mode) then loads 16 bytes per lane and instruction instead of 4: a quarter of the memory
 * instructions, 1 KB instead of 256 B of contiguous memory per wave-load, 6.6 instead of 5.6 TB/s of stream at the headline's occupancy
 * (DESIGN.md 4j).  Results are identical bit for bit (same values, same summation order).  Explicit per-step dev_se_tiles stay RB-major.
 * ranenv_se_retile_quad converts an RB-major pool [n_tiles][R][U] (device pointers, no handle). */
int ranenv_bind_se_pool_quad(ranenv_handle h, const float *dev_pool, int64_t n_tiles, int64_t tile_stride);
int ranenv_se_retile_quad(const float *dev_rb_major, float *dev_quad, int64_t n_tiles, int32_t n_ues, int32_t n_rbs, void *stream);
/* Traffic pool: int32 offered bits, row i = [U] at dev + i*U. */
int ranenv_bind_traffic_pool(ranenv_handle h, const int32_t *dev_pool, int64_t n_rows);

/* Per-env episode descriptors, host pointer [B]. */
int ranenv_set_episodes(ranenv_handle h, const ranenv_episode *host_episodes, void *stream);

/* Inter-slice policy computed on the device when the step gets no scores, and the
 * intra-slice scheduler (RANENV_INTRA_PER_SLICE = take it from the step's intra_choice). */
int ranenv_set_policy(ranenv_handle h, int32_t policy, int32_t fixed_intra);

/* CommunicationEnv.reset for the envs with env_mask[b] != 0 (NULL = all): fresh buffers,
 * step 0, observation of the zero raw state with the episode's first SE tile.
 * Outputs may be NULL. dev_se_tiles: [B][U*R] explicit tiles or NULL to use the pool. */
int ranenv_reset(ranenv_handle h, const uint8_t *dev_env_mask, const float *dev_se_tiles,
                 float *dev_obs_inter /* [B][S*10] */, float *dev_obs_intra /* [B][S][2*Us+9] */,
                 double *dev_reward /* [B][S+1] */, void *stream);

/* One TTI for every env.
 *   dev_inter_scores [B][S] double  action["player_0"]; NULL = device policy
 *   dev_intra_choice [B][S] uint8   action["player_{s+1}"]; NULL = fixed_intra
 *   dev_traffic_bits [B][U] double  traffic.step() output; NULL = traffic pool
 *   dev_se_tiles     [B][R][U] float channel.step() output, RB-major; NULL = SE pool
 *   outputs: obs (float32), reward (double, [0] = player_0), done (uint8: step == max_steps) */
int ranenv_step(ranenv_handle h, const double *dev_inter_scores, const uint8_t *dev_intra_choice,
                const double *dev_traffic_bits, const float *dev_se_tiles,
                float *dev_obs_inter, float *dev_obs_intra, double *dev_reward, uint8_t *dev_done,
                void *stream);

/* Same TTI but with a caller-made dense sched_decision [B][U][R] uint8 (any 0/1 pattern),
 * as an agent's own action_format callback produces it. */
int ranenv_step_dense(ranenv_handle h, const uint8_t *dev_sched_decision,
                      const double *dev_traffic_bits, const float *dev_se_tiles,
                      float *dev_obs_inter, float *dev_obs_intra, double *dev_reward, uint8_t *dev_done,
                      void *stream);

/* Compact steps.  A UE outside every slice is allocated nothing and read by no observation; when it also receives no traffic
 * (MultSliceTraffic.step draws for the UEs of slices with a request only, traffics/mult_slice.py:24-32), stepping it changes
 * nothing but its 10-TTI window, into which it pushes zeros.  ranenv_step / _step_range / _step_part / _rollout therefore
 * step only the UEs that are in a slice (the kernel's lanes are ordered slice members first; waves without one leave at
 * once) whenever that is exact: the traffic comes from the device generator, or from a traffic pool that the library has
 * examined (once per change of pool, scenarios, episodes or episode table: a kernel over the traces, one read-back at the next
 * step) and found empty for every idle UE; and no earlier step can have given an idle UE packets (explicit dev_traffic_bits
 * and dense steps switch compact steps off until the next ranenv_reset of the whole batch).  A UE that comes back into a
 * slice (a reset into another scenario) first makes up for the zero pushes its window missed.  Results are identical to
 * full-width steps; the one visible difference is views.se_mean of idle UEs.  Call ranenv_set_episodes again after changing
 * a bound traffic pool's contents. */

/* The same TTI for the envs [env_first, env_first + env_count) only, enqueued on `stream` and nothing else: no batch
 * partitions, no joins.  All array arguments are the whole-batch arrays of ranenv_step (indexed by env).  This is the
 * building block for a learner in the loop (the reference trains PPO through env.step, simu.py:555-566, with 10
 * concurrent env runners, agents/ray_agent.py:296-300): step two halves of the batch alternately on two streams, and
 * while the policy consumes one half's observations the other half's TTI occupies the GPU.  Ordering between a range's
 * launches and the producer of its scores / consumer of its outputs is the caller's (events), as with any stream. */
int ranenv_step_range(ranenv_handle h, int32_t env_first, int32_t env_count,
                      const double *dev_inter_scores, const uint8_t *dev_intra_choice,
                      const double *dev_traffic_bits, const float *dev_se_tiles,
                      float *dev_obs_inter, float *dev_obs_intra, double *dev_reward, uint8_t *dev_done,
                      void *stream);

/* The same for partition `part` of ranenv_set_partitions, with the stream plumbing done here (one call per launch, no
 * event objects on the caller's side): ranenv_step_part orders the partition's own stream behind what `stream` holds
 * now, enqueues the partition's TTI there and returns; ranenv_wait_part orders `stream` behind that TTI (no host
 * sync).  Pattern for two partitions A, B and a policy running on `stream`:
 *     step_part(A); step_part(B);  loop { wait_part(A); scores_A = policy(obs_A); step_part(A);   -- B's TTI is running
 *                                         wait_part(B); scores_B = policy(obs_B); step_part(B); } -- A's TTI is running
 * The caller must leave partition k's rows of the input arrays alone between step_part(k) and wait_part(k).
 * ranenv_get_partition returns a partition's env range.
 * Every dependency between two streams is a signal between two hardware queues (measured here: ~37 us from the end of a
 * partition's TTI, through a small policy kernel on another stream, to the start of its next TTI).  A learner that runs
 * partition k's policy on partition k's own stream (ranenv_get_part_stream: a hipStream_t owned by the handle, valid until
 * the next ranenv_set_partitions / ranenv_destroy) and passes that stream to step_part / wait_part needs none: both calls
 * then rely on stream order alone, and each partition is an independent in-order chain TTI -> policy -> TTI on its own
 * queue -- the reference's concurrent env runners (agents/ray_agent.py:296-300), one per stream. */
int ranenv_step_part(ranenv_handle h, int32_t part,
                     const double *dev_inter_scores, const uint8_t *dev_intra_choice,
                     const double *dev_traffic_bits, const float *dev_se_tiles,
                     float *dev_obs_inter, float *dev_obs_intra, double *dev_reward, uint8_t *dev_done,
                     void *stream);
int ranenv_wait_part(ranenv_handle h, int32_t part, void *stream);
int ranenv_get_partition(ranenv_handle h, int32_t part, int32_t *env_first, int32_t *env_count);
int ranenv_get_part_stream(ranenv_handle h, int32_t part, void **stream);

/* How steps and resets read SE tiles replayed from the bound pool.
 *   RANENV_SE_STREAM (default)  every TTI streams the env's whole U x R tile: per UE the sum over all RBs (its mean is
 *                               what the observation, PF / MT and MAPF consume: agents/ib_sched.py:110-116,146-157,
 *                               agents/common.py:567-573,648-654, agents/mapf.py:75-90) and the sum over its allocated RBs.
 *   RANENV_SE_GATHER            the per-UE mean over all RBs is a function of the tile alone -- exogenous, identical
 *                               whatever the agent does (results/gen_results.py:1587-1635) -- so it is computed once per
 *                               pooled tile ([tile][U] float64, numpy's pairwise order: bit-identical to the streamed
 *                               value); a TTI then reads that row and, from a UE-major copy of the pool, only the RBs
 *                               each UE was allocated (R elements per env instead of U x R).  Results are bit-identical.
 *                               The call builds both sidecars for the pool bound now (handle-owned: n_tiles * U * (8 +
 *                               4 * roundup(R, 8)) bytes); call it again after changing the pool's contents;
 *                               ranenv_bind_se_pool falls back to STREAM.  Steps with explicit dev_se_tiles and
 *                               ranenv_step_dense keep streaming. */
int ranenv_set_se_mode(ranenv_handle h, int32_t mode, void *stream);
/* Gather-only ingest (channels/quadriga.py:56-76 for a handle that will only ever run the gather mode): builds both sidecars
 * straight from QuaDRiGa received power -- float64 [n_tiles][R][U], RB-major like the .mat -- and switches the handle to
 * RANENV_SE_GATHER; no RB-major float32 pool exists or is needed (footprint per tile U * (8 + 4 * roundup(R, 8)) bytes instead of
 * that plus U * R * 4).  The sidecars are bit for bit what ranenv_se_from_power + ranenv_bind_se_pool + ranenv_set_se_mode(GATHER)
 * build.  Afterwards: reset / step / step_range / step_part / rollout / auto-reset replay pooled tiles through the sidecars;
 * steps with explicit dev_se_tiles stream those; ranenv_step_dense needs explicit tiles; ranenv_set_se_mode(STREAM) fails until a
 * pool is bound (ranenv_bind_se_pool, which drops the gather mode as always).  The power array may be freed when the call returns. */
int ranenv_bind_se_gather_from_power(ranenv_handle h, const double *dev_power, int64_t n_tiles, double tx_power_per_rb,
                                     double noise_power, void *stream);
/* Diagnostic: the gather mode's sidecars: row_mean [n_tiles][U] float64, ue_major [n_tiles][U][row_floats] float32. */
int ranenv_get_se_sidecars(ranenv_handle h, double **dev_row_mean, float **dev_ue_major, int32_t *row_floats);

/* Per-launch timing: between ranenv_profile_begin and ranenv_profile_end every launch of the step kernel carries its
 * dispatch's own start / stop timestamps (hipExtLaunchKernel events, valid with further launches queued behind it).
 * ranenv_profile_end waits for the device and returns the average duration in ms over n_launches launches (with
 * partitions: one launch per partition and TTI). */
int ranenv_profile_begin(ranenv_handle h);
int ranenv_profile_end(ranenv_handle h, double *avg_ms, int32_t *n_launches);
/* TTIs covered by the launches timed since ranenv_profile_begin: inside ranenv_rollout one launch may take its envs
 * through several TTIs (see there), so n_ttis / n_launches TTIs went into the average launch. */
int ranenv_profile_ttis(ranenv_handle h, int64_t *n_ttis);
/* The same, and the env-TTIs (envs of a launch x its TTIs, summed over the timed launches): with the persistent rollout the
 * launches of one call differ in how many envs they step. */
int ranenv_profile_work(ranenv_handle h, int64_t *n_ttis, int64_t *n_env_ttis);

/* Batch partitions: envs are independent, so the batch can be stepped as n_parts contiguous ranges, each by its own
 * launch on its own (handle-owned) HIP stream.  A launch has a ramp and a tail during which CUs idle; with partitions
 * one range's ramp and tail run under the other ranges' steady state.  reset / step / step_dense stay ordered with the
 * caller's stream (they wait for what it holds and it waits for them), so nothing changes for a caller that consumes
 * every TTI's outputs.  n_parts = 1 (default) launches on the caller's stream itself. */
int ranenv_set_partitions(ranenv_handle h, int32_t n_parts);

/* n_steps TTIs under the device policy (the reference's MARR / MAPF evaluation loop, simu.py:555-566, where no learner
 * sits between two TTIs), enqueued in one call: per TTI the same launches as ranenv_step, but the caller's stream is
 * joined only before the first and after the last TTI, so that with partitions range k's TTI t+1 follows its own
 * TTI t directly.  The outputs hold the last TTI's values.  Needs a device policy and bound pools / generator.
 * Where nothing has to happen between two TTIs (no head kernel bound; with auto-reset: no episode of the batch ends),
 * one launch takes its envs through several of them -- up to a quarter of n_steps, at most 10: the workgroup steps
 * its env again from the state it has just written instead of ending and being launched again.  Results are the
 * same bit for bit. */
int ranenv_rollout(ranenv_handle h, int32_t n_steps, float *dev_obs_inter, float *dev_obs_intra,
                   double *dev_reward, uint8_t *dev_done, void *stream);
/* With auto-reset enabled (ranenv_set_autoreset below) the rollout runs through episode ends: an env whose episode
 * finishes at a TTI gets its next episode installed and CommunicationEnv.reset applied right behind that TTI's step,
 * on its partition's stream, before its next TTI -- the reference's evaluation loop over many episodes
 * (simu.py:547-566) without the host between two TTIs.  dev_done is required then. */

/* Episode metrics on the device (what the paper's evaluation derives per TTI from the history files,
 * results/gen_results.py:874-1022, kept as running sums so that a rollout needs no per-TTI read-back).  Per env 8
 * float64 sums over the TTIs of the current episode (a reset zeroes them):
 *   [0] TTIs   [1] inter-slice reward (player_0, calculate_reward_no_mask)
 *   [2] active slices in violation (minimum declared intent drift < 0)      [3] the same, priority slices only
 *   [4] distance to fulfilment: sum over slices of their negative minimum drift   [5] priority slices only
 *   [6] packets sent (pkt_effective_thr over the UEs)    [7] packets dropped
 * When an episode ends under auto-reset, its sums are appended to the env's log of episode_slots rows (later episodes
 * are only counted).  ranenv_enable_metrics(h, slots >= 0) allocates, zeroes and switches on; slots < 0 switches off.
 * ranenv_get_metrics returns device pointers: running [B][8], episode_log [B][slots][8] (NULL if slots == 0),
 * episodes_done [B] int32. */
int ranenv_enable_metrics(ranenv_handle h, int32_t episode_slots, void *stream);
int ranenv_get_metrics(ranenv_handle h, double **dev_running, double **dev_episode_log, int32_t **dev_episodes_done,
                       int32_t *episode_slots);

int ranenv_get_views(ranenv_handle h, ranenv_views *out);

/* Offered traffic drawn on the device instead of replayed from the traffic pool: for every UE of a slice with
 * a request, Poisson(slice Mbps) * 1e6 bits per TTI -- the law of MultSliceTraffic.step, traffics/mult_slice.py:24-32
 * -- from a counter-based generator (Philox-4x32-10) keyed by `seed` with counter (env_id_base + env, episode
 * number, step, UE): a function of those alone, never of the actions (the reference checks that exogenous inputs
 * are identical across agents, results/gen_results.py:1587-1635).  Statistically, not bit-wise, equal to numpy's
 * Generator.poisson stream; an explicit dev_traffic_bits argument of a step still takes precedence.
 * Slice traffic must lie in (0, 128] Mbps (256-entry inversion tables, rebuilt whenever scenarios are loaded). */
int ranenv_set_traffic_generator(ranenv_handle h, int32_t enable, uint64_t seed, int32_t env_id_base, void *stream);
/* Diagnostic: copy the generator's inversion tables to the host: cdf [n_scenarios][S][256] uint64
 * (floor(P(X <= k) * 2^64)), guide [n_scenarios][S][64] uint8.  A draw is the smallest k with u < cdf[k], where
 * u = (out[1] << 32 | out[0]) of Philox-4x32-10(counter = (env id, episode, step, UE), key = (seed lo, seed hi)). */
int ranenv_get_poisson_tables(ranenv_handle h, uint64_t *host_cdf, uint8_t *host_guide);

/* Per-env episode length (host array [B]; NULL = cfg.max_steps for all): done = step_number >= max_steps[b]. */
int ranenv_set_max_steps(ranenv_handle h, const int32_t *host_max_steps, void *stream);

/* Episode advance on the device (env.reset() after `terminated`, simu.py:547-566, without the host in the loop).
 *   ranenv_set_episode_table  descriptor of every episode number in [first_episode, first_episode + n): what
 *                             choose_episode of the association / channel plugins resolves per episode
 *                             (associations/mult_slice.py:444-452, mult_slice_seq.py:38-46, channels/quadriga.py:78-87,
 *                             quadriga_seq.py:28-39)
 *   ranenv_set_autoreset      the rule: next episode = current + 1 (wrapping from max_episode back to
 *                             initial_episode), or uniform in [initial_episode, max_episode) when random_episodes
 *                             (enable_random_episodes, simu.py:361,377); host_episode_no [B] = the episode every env is
 *                             playing now (NULL keeps the device's numbers)
 *   ranenv_autoreset          enqueue after a step: every env with dev_done != 0 gets its terminal observation
 *                             copied to the term_* buffers (each may be NULL), the next episode's descriptor
 *                             installed and CommunicationEnv.reset applied (obs_* receive the new episode's first
 *                             observation; the step's rewards and done flags are left as they are).  No host sync.
 *                             Option "autoreset_shortcut" = 1 (default 0; opt-in): the caller promises that dev_done holds
 *                             exactly what the last step wrote.  `done` is then a function of the env's step counter alone
 *                             (step >= its episode length) and the host follows the counters from a reset of the whole batch
 *                             on: when dev_done is the buffer the last step wrote and no episode ended at that TTI, the call
 *                             enqueues NOTHING (an RL loop calls it behind every step).  What the host cannot follow --
 *                             another buffer, a caller's masked reset before, a stream capture, ranenv_get_views (its views
 *                             are writable, step_number included), a persistent launch that gave up -- ends the shadow until
 *                             the next reset of the whole batch and is left to the device.  With the option at 0 the device
 *                             always reads dev_done: a caller may OR its own truncation flags into the buffer. */
int ranenv_set_episode_table(ranenv_handle h, const ranenv_episode *host_table, int32_t first_episode,
                             int32_t n_episodes, void *stream);
int ranenv_set_autoreset(ranenv_handle h, int32_t enable, int32_t initial_episode, int32_t max_episode,
                         int32_t random_episodes, uint64_t seed, const int32_t *host_episode_no, void *stream);
int ranenv_autoreset(ranenv_handle h, const uint8_t *dev_done, float *dev_obs_inter, float *dev_obs_intra,
                     float *dev_term_obs_inter, float *dev_term_obs_intra, float *dev_term_obs_head, void *stream);
/* The same for partition `part` only, enqueued on the partition's stream behind its last ranenv_step_part (stream plumbing as
 * in ranenv_step_part; ranenv_wait_part then also covers the reset): a learner that steps ranges of the batch as in-order
 * chains keeps the episode advance on the device and inside each chain.  All arrays are the whole-batch arrays. */
int ranenv_autoreset_part(ranenv_handle h, int32_t part, const uint8_t *dev_done, float *dev_obs_inter, float *dev_obs_intra,
                          float *dev_term_obs_inter, float *dev_term_obs_intra, float *dev_term_obs_head, void *stream);

/* Options: tuning and debug knobs of the launch schedule.  NONE of them changes a result -- every setting is covered by the
 * bit-for-bit tests -- they select a build of the step kernel or the way launches are issued.  They are set per handle by
 * ranenv_set_option(h, key, value); ranenv_create presets them from the process environment, ONE variable per key,
 * RANENV_<KEY IN CAPITALS> (read in one place, apply_env_options in csrc/ranenv_host.cpp; the GPU test suite and the A/B tools under
 * tools/ run whole passes that way).  Nothing else in the library reads the environment.
 *
 *   key            env variable         default   meaning
 *   "compact"      RANENV_COMPACT       1         0: never step compactly (see "Compact steps" above), always full width
 *   "fuse"         RANENV_FUSE          0         TTIs one launch of ranenv_rollout takes its envs through: 0 = chosen per
 *                                                 rollout (a quarter of n_steps, at most 10), n = at most n, 1 = one TTI per launch
 *   "fuse_first0"  RANENV_FUSE_FIRST    0         length of partition 0's FIRST launch of a rollout (0 = the staggered default:
 *   ... "fuse_first9"  (= a,b,c list)             the partition enqueued last starts with one TTI); keys 0..9 = partitions 0..9
 *   "row_width"    RANENV_ROW_WIDTH     auto      8, 10 or 16 >= max(S, Us): LDS row width the step kernel is built for
 *   "small_batch"  RANENV_SMALL_BATCH   auto      1: the streaming build with 128 VGPRs and 32 SE loads in flight per lane (chosen
 *                                                 automatically when the batch leaves the CUs at <= 8 workgroups), 0: the lean one
 *   "tiny_step"    RANENV_TINY_STEP     1         one-TTI step launches of a batch that stays within 2 waves per SIMD (<= 1024 envs of 100 UEs) run a build with
 *                                                 the whole SE row and all of the UE's state requested at entry (256 VGPRs): a small batch's step is one chain
 *                                                 of latencies; 0: the "small_batch" build
 *   "mix"          RANENV_MIX           1         a step launch of the whole batch of two-wave workgroups (64 < U <= 128) runs as MIXED BLOCKS where a
 *                                                 compact step is exact: one block per env of more than 64 slice members, one block per TWO envs
 *                                                 of at most 64 (one wave each) -- the whole batch resident in one round; 0: never, 2: also for
 *                                                 batches that fit the chip anyway (tests)
 *   "pack"         RANENV_PACK          1         envs of at most 32 UEs and 8 slices / 8 UEs per slice (the reference's own size) are stepped
 *                                                 TWO per wave, lanes 0-31 / 32-63, wherever a step launch covers an even number of them
 *                                                 (ranenv_set_partitions cuts an even batch into even ranges); 0: one env per wave
 *   "persist"      RANENV_PERSIST       -1        -1: where it was measured to win (SE gather mode with a batch above 8 workgroups per CU
 *                                                 and up to about twice what the chip holds; either mode with a batch that stays within 2 waves
 *                                                 per SIMD; the streaming kernel at up to 20 envs per CU for rollouts of 4...64 TTIs, where it is 2-9 % ahead -- longer ones are a tie, larger batches lose), 0: never, 1: wherever possible -- ranenv_rollout runs as ONE persistent launch per workgroup class for all the
 *                                                 TTIs up to the next episode end: the envs are sorted by the waves a compact step
 *                                                 of theirs needs (64 slice members per wave), each class gets a grid of what the
 *                                                 chip holds, and a workgroup that finishes a chunk of TTIs hands its env over
 *                                                 (per-XCD ready queues) only when another env is waiting for a slot.  Needs compact
 *                                                 steps and no bound head outputs (else the rollout runs as described above)
 *   "persist_chunk" RANENV_PERSIST_CHUNK 10       TTIs of an env between two looks at the queues (a launch of n TTIs uses min(this, n - 1): an env's first chunk
 *                                                 is 1...chunk TTIs long by a hash of its index, and a launch without a chunk end inside it runs 8-9 % slower)
 *   "persist_grid" RANENV_PERSIST_GRID  0         cap on the wave slots the persistent grids are sized for (0 = the occupancy
 *                                                 query x CUs); small values force hand-overs (tests)
 *   "persist_errors" (read only)                  persistent launches in which a wait gave up after ~1 s (0 in every correct run).  Such a launch
 *                                                 drops its envs after their current chunk; the NEXT ranenv_rollout call sees the sticky error word
 *                                                 (host-visible memory, no device sync), clears the queues, sets "persist" to 0 for the handle and
 *                                                 fails with RANENV_E_STATE: the envs have advanced different numbers of TTIs, reset the batch
 *   "persist_inject_abort" (no env variable)      test hook: 1 = the next persistent launch finds a wait already given up
 *   "autoreset_shortcut" RANENV_AUTORESET_SHORTCUT 0  1: ranenv_autoreset / _part trust the host's shadow of the step counters (see ranenv_autoreset) and
 *                                                 enqueue nothing at a TTI at which no episode ended; 0: the device reads dev_done every time.
 *                                                 (This one selects whose flags count -- with 1 the caller must not modify dev_done.)
 *   "last_rollout_persistent" / "last_rollout_launches" (read only)   what the last ranenv_rollout call ran: 1 = persistent work-queue
 *                                                 launches (else launches of <= 10 TTIs per partition); how many step-kernel launches it enqueued
 *   "persist_stat_keep" / "_push" / "_pop" / "_fresh" / "_idle_polls" (read only)   queue statistics of the persistent launches so far, summed over
 *                                                 classes and XCDs: chunk ends at which the workgroup kept its env, envs put down, envs taken
 *                                                 from a ready queue, envs taken fresh, spins on a slot whose pusher had not written yet
 *
 * Python host layer only (batched_env.py, not this library): RANENV_SE_MODE=gather makes BatchedRanEnv.bind_se_pool switch
 * to the SE gather mode, RANENV_LIB=<path> loads another build of this library.
 * ranenv_get_option reads a key back.  Unknown keys and unusable values return RANENV_E_INVALID. */
int ranenv_set_option(ranenv_handle h, const char *key, int64_t value);
int ranenv_get_option(ranenv_handle h, const char *key, int64_t *value);

/* Diagnostic, pure host arithmetic (no GPU, no handle): would a handle of this configuration with pools of these extents step two envs
 * per wave (option "pack")?  Packed waves address a per-env row as array base + a 32-bit offset, so every array they address that way
 * -- per-UE tables and state slabs, window rings, slice tables, the intent-parameter tables (two blocks), score / observation / reward
 * rows, the traffic pool, the sidecar of per-tile means -- must stay below 4 GB.  1 = yes, 0 = no (one env per wave), < 0 = error. */
int ranenv_packed_step_fits(const ranenv_config *cfg, int64_t traffic_rows, int64_t se_tiles);

/* Self-test of the kernels' division.  The step kernels divide doubles with the compiler's correctly rounded sequence WITHOUT its three guard
 * instructions (v_div_scale x 2, v_div_fixup) wherever the divisor is a positive normal number of moderate magnitude whenever the result is
 * used -- packet sizes, counts, sums, the validated normalisers; explicit traffic (the caller's doubles) goes through the plain operator.
 * This entry point runs both forms inside the SHIPPED build on the caller's operands (device pointers, n elements): fast[i] = the guard-free
 * sequence, ieee[i] = a[i] / b[i].  They are bit-identical for a = 0 or a in [2^-900, 2^900], b in [1e-30, 1e30] and a quotient in
 * [2^-900, 2^900] (tests/test_gpu_flags_and_errors.py: the kernels' operands -- counts, sizes, sums of 1e-9 ... 1e12 -- and the edges of that
 * domain); outside it -- b zero, denormal, infinite or NaN, or operands so small that the correction step's residual is a denormal -- only
 * ieee is defined.  No handle (errors: ranenv_last_error(NULL)). */
int ranenv_selftest_ddiv(const double *dev_a, const double *dev_b, double *dev_fast, double *dev_ieee, int64_t n, void *stream);

/* Last launch geometry (for roofline accounting): grid blocks, block threads, LDS bytes. */
int ranenv_launch_info(ranenv_handle h, int32_t *grid, int32_t *block, int32_t *lds_bytes);

/* Alternative heads.  SchedTWC and SchedColORAN (agents/sched_twc.py, agents/sched_colran.py)
 * are single-agent wrappers around IBSched: same action_format with fixed_intra = "rr" (sched_twc.py:415-422,
 * i.e. ranenv_set_policy(EXTERNAL, RR) + ranenv_step with inter-slice scores), but their own observation
 * (sched_twc.py:165-346: per slice 3 requirement values and 7 slice means, slices in index order, laid out
 * metric-major: [reliability, latency, throughput req of slice 0, ... of slice S-1 | mean SE x S | served Mbps |
 * effective Mbps | buffer occupancy | buffer latency | packet-loss rate | requested Mbps]) and rewards
 * (sched_twc.py:348-413; sched_colran.py:348-419).  Once outputs are bound, reset / step / step_dense also
 * launch the head kernel:
 *   dev_obs_head    float32 [B][10*S]
 *   dev_reward_head float64 [B][2]     [0] SchedTWC.calculate_reward, [1] SchedColORAN.calculate_reward
 * NULL, NULL unbinds.  The heads read pkt_throughputs, so they refuse RANENV_F_NO_RAW_OUTPUT. */
int ranenv_bind_head_outputs(ranenv_handle h, float *dev_obs_head, double *dev_reward_head);
/* SchedColORAN's slice-name table (sched_colran.py:356-367) as data: host array [count][S], bit 0 = eMBB,
 * bit 1 = URLLC, for scenario-pool rows [first, first+count).  Default 0 (no reward term). */
int ranenv_set_slice_usecase(ranenv_handle h, int32_t first, int32_t count, const int32_t *usecase, void *stream);

/* Channel ingest: QuaDRiGa received power -> spectral efficiency, element by element,
 *     se = float32( log2(1 + tx_power_per_rb * g / (0 + noise_power)) )            (float64 arithmetic)
 * replaces QuadrigaChannel.step's per-TTI transform (channels/quadriga.py:56-69; tx_power_per_rb =
 * transmission_power / num_available_rbs = 100 / R, noise_power = 10e-14, inter-cell interference 0).
 * The per-step slice of target_cell_power is RB-major like the SE pool (:70-72 transposes it for the
 * agents), so a power array [n][R][U] maps onto an SE pool [n][R][U] without a transpose.  Device
 * pointers, n_elems elements; no handle (errors: ranenv_last_error(NULL)). */
int ranenv_se_from_power(const double *dev_power, float *dev_se, int64_t n_elems,
                         double tx_power_per_rb, double noise_power, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RANENV_H */
