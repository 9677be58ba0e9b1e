// ranenv_host.cpp -- host side of the C ABI in include/ranenv.h: handles, validation, pools, launch schedules (partitions, rollouts,
// persistent work-queue launches, ranges), options, episode advance.  Plain C++ on the HIP runtime API; every kernel is reached through
// the launch table of ranenv_internal.h.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ranenv_internal.h"

using namespace ranenv_dev;

namespace {
thread_local std::string g_last_error;

}  // namespace

struct ranenv {
    ranenv_config cfg;
    KP kp;
    std::vector<void *> allocs;
    ranenv_episode *d_episodes = nullptr;
    bool have_scenarios = false, have_episodes = false;
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
    int autoreset_shortcut = 0;                 // option "autoreset_shortcut" (default 0: ranenv_autoreset reads dev_done, every env with a non-zero flag restarts)
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

// The build of the step kernel for this handle and launch: SE gather or streaming (lean / small-batch / whole-row), one or several TTIs.
template <int MODE>
void launch_kernels(ranenv_handle h, const KP &kp, dim3 grid, dim3 block, hipStream_t stream, hipEvent_t ev0, hipEvent_t ev1, bool gather)
{
    StepLaunch l;
    l.np = h->np; l.mode = MODE; l.many = MODE == MODE_STEP && kp.n_tti > 1; l.gather = gather; l.build = SB_LEAN;
    // RANENV_F_SCALE_PER_ELEMENT: the lean builds with the other rounding of the masked SE sum (MODE_PE); a reset sums no masked row
    if (MODE != MODE_RESET && scale_per_element(h)) {
        if (gather && MODE != MODE_STEP) return;
        l.mode = MODE | MODE_PE; l.build = gather ? SB_GATHER : SB_LEAN;
    } else if (gather) {
        if (MODE == MODE_DENSE) return;
        l.build = SB_GATHER;
    } else if (MODE == MODE_STEP && !l.many && h->tiny_step && persist_tiny(h)) {      // a batch at <= 2 waves per SIMD: the whole-row build
        l.build = SB_TINY1;
    } else if (h->small_batch) {
        l.build = SB_SMALL;
    }
    (void)launch_step(l, grid, block, stream, ev0, ev1, kp);
}

int persist_prepare(ranenv_handle h, hipStream_t stream, bool need_host_counts);
bool persist_tiny(ranenv_handle h);
bool stream_capturing(hipStream_t stream);

// Packed waves address a per-env row as (uniform array base) + (32-bit row + lane offset), see row_at<2>: every array they
// address that way must stay below 4 GB.  True for every size a packed step makes sense at (the reference's: megabytes); a handle
// with pools beyond that steps one env per wave.
bool pack_fits_32_of(const ranenv_config &cfg, long long trf_rows_n, long long se_tiles_n)
{
    const unsigned long long lim = 1ull << 32, B = (unsigned long long)cfg.batch, U = (unsigned long long)cfg.n_ues,
                             S = (unsigned long long)cfg.n_slices, NS = (unsigned long long)cfg.n_scenarios,
                             D = (unsigned long long)cfg.hist_depth, W = 2ull * cfg.max_ues_slice + 9ull;
    // (one term per array a packed step addresses as base + 32-bit offset, each the array's whole allocation in bytes: the per-UE tables
    // (two lane orders), the per-UE state slabs, the window rings, the slice tables, the intent parameters -- two blocks, the second BY
    // METRIC --, the score rows, the observation rows, the reward rows, the traffic pool / explicit traffic, the gather sidecar of means)
    return (unsigned long long)N_TUE * NS * U * 4 < lim && (unsigned long long)N_U4 * B * U * 4 < lim && (unsigned long long)N_U8 * B * U * 8 < lim &&
           B * D * U * 4 < lim && NS * S * 32 < lim && 2 * NS * S * 24 < lim && NS * S * 16 < lim && B * S * 8 < lim && B * (S + 1) * 8 < lim &&
           B * S * W * 4 < lim && (unsigned long long)trf_rows_n * U * 4 < lim && (unsigned long long)se_tiles_n * U * 8 < lim;
}
bool pack_fits_32(ranenv_handle h) { return pack_fits_32_of(h->cfg, h->trf_rows_n, h->se_tiles_n); }

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
            kq.p_list = h->d_plist + (size_t)B; kq.m_list = h->d_plist; kq.m_counts = h->d_pcount;
            const dim3 mgrid((unsigned)n), mblock((unsigned)(2 * WAVE));       // (an upper bound of wide + ceil(narrow / 2))
            const StepLaunch l{h->np, SB_MIXED, MODE_STEP, kq.n_tti > 1, gather};
            (void)launch_step(l, mgrid, mblock, stream, ev0, ev1, kq);
            if (kp.head_obs || kp.head_reward) launch_head(stream, grid, dim3((unsigned)h->nslot), kp);
            return hipGetLastError();
        }
        // packed waves: two envs per wave for envs of <= 32 UEs / <= 8 slices (see ranenv_core_kernel_packed)
        if (!scale_per_element(h) && h->pack && h->np == 8 && h->cfg.n_ues <= 32 && h->nt == WAVE && (n & 1) == 0 && kp.env_mask == nullptr && pack_fits_32(h)) {
            KP kq = kp;
            const dim3 pgrid((unsigned)(n / 2)), pblock((unsigned)WAVE);
            const StepLaunch l{8, SB_PACKED, MODE_STEP, kq.n_tti > 1, gather};
            (void)launch_step(l, pgrid, pblock, stream, ev0, ev1, kq);
            if (kp.head_obs || kp.head_reward) launch_head(stream, grid, dim3((unsigned)h->nslot), kp);
            return hipGetLastError();
        }
    }
    launch_kernels<MODE>(h, kp, grid, block, stream, ev0, ev1, gather);
    if (kp.head_obs || kp.head_reward) launch_head(stream, grid, dim3((unsigned)h->nslot), kp);
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
        launch_idle_traffic(stream, (unsigned)h->cfg.batch, h->d_episodes, kp.trf_pool, U, TB_ue_slice(h->kp), TB_lane_ue(h->kp), h->d_violations);
        if (h->d_ep_table)
            launch_idle_traffic(stream, (unsigned)h->ep_table_n, h->d_ep_table, kp.trf_pool, U, TB_ue_slice(h->kp), TB_lane_ue(h->kp), h->d_violations + 1);
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

void finalize_kp(ranenv_handle, KP &kp) { kp.n_tti = 1; }      // (what every launch of a call shares: one TTI unless ranenv_rollout says more)

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
    if (!h->autoreset_shortcut || !h->sh_valid || dev_done == nullptr || dev_done != h->last_done || stream_capturing(stream)) return -1;
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
        launch_classify(stream, h->d_episodes, h->d_members, B, NC, one_class, h->d_plist, h->d_pcount, h->d_cls_flag, 1);
        if (!capturing) { h->pclass_dirty = false; h->pclass_maybe = false; h->pcount_host_stale = true; }
    } else if (h->pclass_maybe) {
        launch_classify(stream, h->d_episodes, h->d_members, B, NC, one_class, h->d_plist, h->d_pcount, h->d_cls_flag, 0);
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
    h->sh_valid = false;           // the envs have advanced different numbers of TTIs: the host no longer knows the step counters
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
    kp.n_tti = n_tti; kp.compact = 1; kp.e0 = 0;
    // (a chunk is always shorter than the launch: an env's FIRST chunk is then 1...chunk TTIs long by a hash of its index, and the envs
    // reach their chunk ends -- a wait for their stores, a look at the queues -- at different TTIs instead of never: there they find the env cursors exhausted
    // and remember it; without a chunk end all workgroups finish together and walk the 8 exhausted cursors with 41 000 device-scope fetch-adds.  Measured, round 6: rollouts of
    // 6 / 8 / 10 TTIs with the default chunk of 10 cost 8-9 % MORE than the launch-per-chunk rollout, with a chunk below the rollout's length
    // 6-7 % LESS, profiles/r06_ab_log.txt)
    kp.p_chunk = h->persist_chunk < n_tti ? h->persist_chunk : (n_tti > 1 ? n_tti - 1 : 1);
    kp.p_cap = h->p_cap; kp.p_err = h->d_perr_dev;
    if (gather) {
        kp.se_pool = h->d_se_um; kp.se_stride = (long long)h->cfg.n_ues * h->se_rp;
        kp.se_mean_pool = h->d_se_mean; kp.se_rp = h->se_rp;
    }
    const bool tiny = persist_tiny(h);            // (then every env is in the widest class and the grid is the batch)
    int &slots_cu = h->p_wave_slots[gather ? 1 : 0];
    if (slots_cu == 0) {
        int nb = 0;
        const void *fn = step_kernel_ptr(StepLaunch{h->np, SB_PERSIST, MODE_STEP, true, gather});
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
        // (the envs are NOT bound to workgroups statically, although at B 4096 the grids equal the class sizes -- 5119 of 5120 wave slots: two-wave blocks
        // do not pack perfectly among one-wave blocks, a few dozen workgroups start late, and an env bound to one of those would wait for a whole
        // rollout of somebody else; through the cursors the resident workgroups pick those envs up at their chunk ends.  Measured, round 6:
        // block i <- list[i] costs +21 % per TTI at K = 20 and +27 % at K = 200 in gather mode, profiles/r06_ab_log.txt.  Nor is a launch ever ONE chunk,
        // not even for a batch whose every env has a resident workgroup of its own (configs[1]: round 5 ran those as one chunk): see kp.p_chunk above --
        // with chunks configs[1] steps 1.2-1.5 % faster at K = 10 / 20)
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
        (void)launch_step(StepLaunch{h->np, (!gather && tiny) ? SB_PERSIST_TINY : SB_PERSIST, MODE_STEP, true, gather}, grid, block, s, ev0, ev1, kc);
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
    if (k == "autoreset_shortcut") { h->autoreset_shortcut = v != 0 ? 1 : 0; return RANENV_OK; }
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
    static const char *const keys[] = {"compact", "fuse", "row_width", "small_batch", "tiny_step", "persist", "persist_chunk", "persist_grid", "pack", "mix", "autoreset_shortcut"};
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
    // (the kernels divide by these with the guard-free sequence of ddiv: positive NORMAL numbers of moderate magnitude only)
    auto sane = [](double v) { return v >= 1e-30 && v <= 1e30; };
    if (!sane(cfg->bandwidth_hz)) return fail(nullptr, RANENV_E_INVALID, "bandwidth_hz must be a finite positive number in [1e-30, 1e30]");
    if (cfg->flags & ~(RANENV_F_CLEAR_HISTORY_ON_RESET | RANENV_F_NO_RAW_OUTPUT | RANENV_F_SYNC_CHECK | RANENV_F_SCALE_PER_ELEMENT))
        return fail(nullptr, RANENV_E_INVALID, "unknown bits in flags (0x%x)", (unsigned)cfg->flags);
    if (!sane(cfg->norm_traffic) || !sane(cfg->norm_ues) || !sane(cfg->norm_se))
        return fail(nullptr, RANENV_E_INVALID, "norm_traffic, norm_ues, norm_se (the observation's normalisers, agents/ib_sched.py:166-168) must be finite positive numbers in [1e-30, 1e30]");
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
        e = hipFuncGetAttributes(&fa, step_kernel_ptr(StepLaunch{16, SB_LEAN, MODE_STEP, false, false}));
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
    else if (k == "row_width") *value = h->np;
    else if (k == "small_batch") *value = h->small_batch ? 1 : 0;
    else if (k == "tiny_step") *value = h->tiny_step;
    else if (k == "persist") *value = h->persist;
    else if (k == "persist_chunk") *value = h->persist_chunk;
    else if (k == "persist_grid") *value = h->persist_grid;
    else if (k == "pack") *value = h->pack ? 1 : 0;
    else if (k == "mix") *value = h->mix;
    else if (k == "autoreset_shortcut") *value = h->autoreset_shortcut;
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
    h->have_scenarios = true;
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
    launch_se_retile_quad((hipStream_t)stream, (unsigned)blocks, dev_rb_major, dev_quad, n_quads, (int)n_ues, (int)n_rbs);
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
    h->have_episodes = true; h->idle_check_dirty = true; h->pclass_dirty = true;
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
        launch_se_sidecar(stream, (unsigned)n, (unsigned)h->nt, h->kp.se_pool, (long long)h->kp.se_stride, (long long)t0, U, R, Rp, h->kp.se_quad,
                          h->d_se_mean, h->d_se_um);
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
        launch_se_sidecar_from_power(stream, (unsigned)n, (unsigned)h->nt, dev_power, (long long)t0, U, R, Rp, tx_power_per_rb, noise_power,
                                     h->d_se_mean, h->d_se_um);
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
    // (auto, streaming: rollouts of 4...64 TTIs of a batch the chip holds at once -- see below)
    const bool stream_short = h->se_mode != RANENV_SE_GATHER && !h->small_batch && (long long)h->cfg.batch <= 20ll * h->n_cus && n_steps >= 4 && n_steps <= 64;
    const bool persist_wanted = (RANENV_DIAG == 0 || RANENV_DIAG == 12) && (h->persist == 1 || (h->persist < 0 && ((h->se_mode == RANENV_SE_GATHER && !h->small_batch && (long long)h->cfg.batch <= 44ll * h->n_cus) || persist_tiny(h) || stream_short)));
    // (auto: where it was measured to win or tie -- profiles/r04_ab_log.txt.  Gather mode: B 1024 -4...-6 %, 2048 -1 %, 4096 -6 %, 8192 -2 % per
    // TTI; a batch of several times what the chip holds -- 16 384 one-wave envs at the reference's own size -- swaps at every chunk and
    // loses 7 %.  Streaming: -10 % at <= 2 waves per SIMD with the whole-row build; at B 4096 a tie: six same-box pairs against the
    // launches of <= 10 TTIs over three partitions, between -5 and +6 % for rollouts of 200 TTIs (mean +0.2 %) and between -1 and +4 % for
    // rollouts of 20 (mean +0.6 %) -- the streaming kernel is bound by HBM either way -- so there it stayed off unless asked for.
    // Round 5, RB-quad-major pool + non-temporal tile loads: same-box pairs on five boxes (profiles/r05_ab_log.txt) give -0.5...-3 % for
    // rollouts of 20 (mean -1.6 %), -5 % for 16, about -1 % for 40...100, a tie at 200 and +9 % for rollouts of 10 (the staggered first chunk is
    // most of such a call); B 8192 loses 7 % (more workgroups than slots: every chunk swaps).  Hence: on for 16...64 TTIs at <= 20 envs per CU.
    // Round 6: what made the short rollouts lose was the chunk, not the schedule -- with a chunk shorter than the launch (persist_launch) rollouts of 6 / 8 / 10 / 12
    // TTIs are 9 / 9 / 10 / 6 % ahead of the launch-per-chunk rollout (gather mode, always persistent at this size: 12-14 % ahead of itself); 5 and 4 TTIs: 7 and 4 % ahead, 3 a tie, 2 behind by 16 %: on from 4 TTIs.)
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
            launch_advance(stream, (unsigned)h->cfg.batch, adv);
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
            launch_advance(s, (unsigned)n, a);
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
    launch_advance(stream, (unsigned)h->cfg.batch, a);
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
    launch_advance(ps, (unsigned)n, a);
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
    h->sh_valid = false;           // (the views are writable, step_number included: the host's shadow of the counters ends here; the next full reset restarts it)
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
    launch_se_from_power((hipStream_t)stream, (unsigned)blocks, dev_power, dev_se, (long long)n_elems, tx_power_per_rb, noise_power);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, RANENV_E_HIP, "se_from_power launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_packed_step_fits(const ranenv_config *cfg, int64_t traffic_rows, int64_t se_tiles)
{
    if (!cfg || traffic_rows < 0 || se_tiles < 0) return fail(nullptr, RANENV_E_INVALID, "null / negative argument");
    return pack_fits_32_of(*cfg, (long long)traffic_rows, (long long)se_tiles) ? 1 : 0;
}

int ranenv_selftest_ddiv(const double *dev_a, const double *dev_b, double *dev_fast, double *dev_ieee, int64_t n, void *stream)
{
    if (!dev_a || !dev_b || !dev_fast || !dev_ieee) return fail(nullptr, RANENV_E_INVALID, "null argument");
    if (n < 0) return fail(nullptr, RANENV_E_INVALID, "negative element count");
    if (n == 0) return RANENV_OK;
    launch_ddiv_selftest((hipStream_t)stream, dev_a, dev_b, dev_fast, dev_ieee, (long long)n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, RANENV_E_HIP, "ddiv self-test launch: %s", hipGetErrorString(e));
    return RANENV_OK;
}

int ranenv_launch_info(ranenv_handle h, int32_t *grid, int32_t *block, int32_t *lds_bytes)
{
    if (!h) return fail(h, RANENV_E_INVALID, "null handle");
    if (grid) *grid = h->cfg.batch;
    if (block) *block = h->nt;
    if (lds_bytes) *lds_bytes = (int32_t)shared_core_bytes(h->np);
    return RANENV_OK;
}

}  // extern "C"

