// f110_step.hip -- part of the C ABI (include/f110_hip.h) over the gfx950 kernels; see f110_internal.h for the units.
#define F110_UNIT_STEP
#define F110_UNIT_NOISE  // (the per-env noise rows are produced by a launch of the step)
#include "f110_internal.h"

// ---------------------------------------------------------------- launches
// Every kernel of the step path goes through emit(): launched at once on a stream (eager, or inside somebody's stream
// capture), or recorded as a node description for a HIP graph the library builds itself (f110_graph_create).
struct KernelLaunch {
    const void *func;
    dim3 grid, block;
    unsigned shmem;
    std::vector<char> args; // the kernel's single by-value argument block
};

struct Sink {
    hipStream_t st = nullptr;
    std::vector<KernelLaunch> *record = nullptr;
};

template <typename Args>
static int emit(const Sink &k, const void *func, dim3 grid, dim3 block, unsigned shmem, const Args &a, hipEvent_t ev0 = nullptr,
                hipEvent_t ev1 = nullptr)
{
    static_assert(__is_trivially_copyable(Args), "kernel argument blocks are copied byte for byte");
    if (k.record) {
        KernelLaunch l;
        l.func = func; l.grid = grid; l.block = block; l.shmem = shmem;
        l.args.assign((const char *)&a, (const char *)&a + sizeof(Args));
        k.record->push_back(std::move(l));
        return F110_OK;
    }
    void *params[1] = {(void *)&a};
    // plain launches unless the measurement aid attached events (a captured hipGraph then holds ordinary kernel nodes)
    if (ev0 || ev1) HIP_TRY(hipExtLaunchKernel(func, grid, block, params, shmem, k.st, ev0, ev1, 0));
    else HIP_TRY(hipLaunchKernel(func, grid, block, params, shmem, k.st));
    return F110_OK;
}

static Sink make_sink(f110_handle *h, hipStream_t st, std::vector<KernelLaunch> *record = nullptr);

static ScanDev scan_dev(const f110_handle *h)
{
    ScanDev s;
    s.nb = h->cfg.num_beams; s.theta_dis = h->cfg.theta_dis; s.fov = h->cfg.fov; s.eps = h->cfg.eps;
    s.max_range = h->cfg.max_range; s.inc = h->theta_inc; s.inc_fx = (unsigned long long)std::llround(h->theta_inc * 1099511627776.0); s.cs_len = h->cs_len; s.cs = h->d_cs;
    return s;
}

// ev0 / ev1 (measurement aid, may be null): start / stop events attached to the dispatch itself, which costs
// less than bracketing the launch with two hipEventRecord calls (those add two barrier packets to the queue)
// which scan instantiation a launch may use: origin unrotated / resolution a power of two for EVERY map its cars touch
struct MapKind { bool ident, pow2; };

template <int SM>
static int launch_scan_t(MapKind kind, const ScanArgs &a, const Sink &k, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr)
{
    int waves = 0;
    for (int i = 0; i < a.n_stages; i++) waves += a.stage_cars[i] << a.stage_log2w[i];
    const dim3 grid(a.wg_single ? waves : (waves + SCAN_WAVES - 1) / SCAN_WAVES), block(a.wg_single ? WAVE : SCAN_THREADS);
    // sweeps only: F110_SCAN_PAD_LDS=<bytes> of unused dynamic LDS per workgroup caps the workgroups per CU (160 KiB / (8.6 KiB + pad)),
    // i.e. emulates a lower occupancy without touching the kernel
    static const unsigned pad_lds = getenv("F110_SCAN_PAD_LDS") ? (unsigned)atoi(getenv("F110_SCAN_PAD_LDS")) : 0u;
    const void *f = kind.ident && kind.pow2 ? (const void *)&scan_kernel<true, true, SM>
                  : kind.ident              ? (const void *)&scan_kernel<true, false, SM>
                  : kind.pow2               ? (const void *)&scan_kernel<false, true, SM>
                                        : (const void *)&scan_kernel<false, false, SM>;
    return emit(k, f, grid, block, pad_lds, a, ev0, ev1);
}

// Waves per car.  Measured on MI355X (profiles/r01g, r01i): a wave's lifetime is bounded
// below by its longest ray (~50 us), so splitting a car's beams over several waves only
// pays while the chip is nearly empty: scan time at 256 / 1024 cars 76 -> 49 us and
// 87 -> 65 us with 8 waves per car, but 121 -> 143 us at 4096 cars (prologues and the
// shorter queues' tails eat the extra parallelism).  F110_WPC overrides the choice.
static int waves_per_car(int n_cars, int num_beams)
{
    static const char *env = getenv("F110_WPC");
    int wpc = env ? atoi(env) : (n_cars <= 1024 ? 8 : (n_cars <= 2048 ? 4 : 1));
    if (wpc != 2 && wpc != 4 && wpc != 8) wpc = 1;
    const int nch = (num_beams + 63) / 64;
    while (wpc > 1 && wpc > nch) wpc /= 2;
    return wpc;
}

#if defined(F110_TIMELINE)
// diagnostics build only (tools/timeline.py): per-wave time stamps of the last scan / car-group launch
static unsigned long long *g_timeline = nullptr;
static const size_t TIMELINE_WAVES = (size_t)1 << 20;
static unsigned long long *timeline_buffer()
{
    if (!g_timeline && hipMalloc((void **)&g_timeline, TIMELINE_WAVES * 4 * sizeof(unsigned long long)) != hipSuccess) g_timeline = nullptr;
    if (g_timeline) (void)hipMemset(g_timeline, 0, TIMELINE_WAVES * 4 * sizeof(unsigned long long));
    return g_timeline;
}
extern "C" int f110_debug_timeline(unsigned long long *out_host, int64_t n_waves)
{
    if (!g_timeline || !out_host || n_waves < 0 || (size_t)n_waves > TIMELINE_WAVES) return F110_E_INVALID;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_host, g_timeline, (size_t)n_waves * 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return F110_OK;
}
#endif

struct StageSpec { int cars, lg; }; // cars < 0: "*", the remaining cars

// "cars:log2waves,..." with at most one "*": strict syntax (f110_set_scan_stages refuses what this refuses)
static bool parse_stage_spec(const char *p, std::vector<StageSpec> &spec, const char **why)
{
    spec.clear();
    int stars = 0;
    if (!p || !*p) { *why = "empty"; return false; }
    for (;;) {
        int cars = -1, lg = 0;
        if (*p == '*') { p++; stars++; }
        else if (*p >= '0' && *p <= '9') {
            long v = strtol(p, (char **)&p, 10);
            if (v > 0x3fffffff) { *why = "car count too large"; return false; }
            cars = (int)v;
        } else { *why = "expected a car count or *"; return false; }
        if (*p == ':') {
            p++;
            if (!(*p >= '0' && *p <= '9')) { *why = "expected log2(waves per car) after ':'"; return false; }
            long v = strtol(p, (char **)&p, 10);
            if (v > SCAN_MAX_LOG2W) { *why = "log2(waves per car) above 3"; return false; }
            lg = (int)v;
        }
        spec.push_back({cars, lg});
        if (*p == ',') { p++; continue; }
        if (*p) { *why = "unexpected character"; return false; }
        break;
    }
    if (stars > 1) { *why = "more than one *"; return false; }
    if (spec.size() > 6) { *why = "more than 6 stages"; return false; }
    return true;
}

// Every pointer a scan launch dereferences without a test of its own, checked on the host: a null here is an error
// code, on the device it is "Memory access fault ... on address (nil)" in every wave (round 2, gpurun_out/r02d).
static int check_scan_args(const ScanArgs &a, const char *who)
{
    if (a.n_cars < 1 || a.agents < 1 || a.scan.nb < 2 || a.scan.nb > MAX_CHUNKS * 64) return fail(F110_E_INVALID, "%s: %d cars, %d agents, %d beams", who, a.n_cars, a.agents, a.scan.nb);
    if (!a.maps || !a.scan.cs || a.scan.cs_len < a.scan.theta_dis || !a.chunk_beam0 || !a.pose_src) return fail(F110_E_INVALID, "%s: a table of the scan is missing (maps / {cos,sin} LUT / chunk order / poses)", who);
    if (!a.out_f32 && !a.out_f64) return fail(F110_E_INVALID, "%s: no output buffer", who);
    if (a.state && (!a.noise_step || !a.noise_base || a.noise_cap < 1 || !a.beam_cosines || !a.in_collision || !a.pending_reset))
        return fail(F110_E_INVALID, "%s: a buffer of the step's scan is missing (noise / beam cosines / in_collision / pending_reset)", who);
    if (!a.state && a.reset_only) return fail(F110_E_INVALID, "%s: reset_only without the step's buffers", who);
    return F110_OK;
}

static int launch_scan(f110_handle *h, const ScanArgs &a_in, const Sink &st, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                       const MapKind *kind_or_null = nullptr)
{
    const MapKind kind = kind_or_null ? *kind_or_null : MapKind{h->ident, h->pow2};
    int rc_args = check_scan_args(a_in, "scan launch");
    if (rc_args) return rc_args;
    ScanArgs a = a_in;
    a.wpc = waves_per_car(a.n_cars, a.scan.nb);
    // Drain of a launch: workgroups are dispatched in index order and nothing follows the last ones,
    // so the chip empties over one wave lifetime (about half of it lost: ~5 % at 65 536 cars -- the gap
    // that two half-size launches from two processes close by overlapping).  The last cars therefore
    // run as 4 short waves each: the wave -> car mapping is a list of stages (cars, log2 waves per car).
    // Measured (profiles/r01j): 65 536 cars 0.702 -> 0.672 ms for any tail of 1 000 .. 2 048 cars (it has to
    // cover the last of the slowest cars), 32 768: 0.380 -> 0.368, 16 384: 0.225 -> 0.218, 8 192: neutral,
    // 4 096: 0.126 -> 0.105 with half of the cars split; graded tails (halves, quarters, eighths) and graded
    // heads changed nothing.  F110_STAGES="cars:log2waves,..." with one "*" for the remaining cars overrides
    // the choice below (e.g. "*:0,2048:2" is the default for big launches).
    // (Measured and dropped in round 2, profiles/r02_multicar_waves_sweep.txt: stages that give one wave K = 2, 4, 8
    // consecutive cars to march back to back, so that a wave drains once per K cars -- 0.705 ms at best against
    // 0.664 ms: the leaner refill of one car per wave and the finer-grained launch win.)
    static const char *stages_env0 = getenv("F110_STAGES");
    const char *stages_env = h->stages.empty() ? stages_env0 : h->stages.c_str();
    const int nch = (a.scan.nb + 63) / 64;
    int lg_all = a.wpc >= 8 ? 3 : a.wpc >= 4 ? 2 : a.wpc >= 2 ? 1 : 0;
    typedef StageSpec St;
    std::vector<St> stv;
    if (h->stages.empty() && (lg_all > 0 || nch < 8)) stv.push_back({a.n_cars, lg_all});
    else {
        std::vector<St> spec;
        const char *why = nullptr;
        if (!stages_env || !parse_stage_spec(stages_env, spec, &why)) { // (a malformed F110_STAGES: the built-in choice)
            // (envs of several agents: 4 096 -- 16 384 x 2: scan 0.396 -> 0.388 ms, 32 768 x 2: 0.697 -> 0.672; 8 192 x 4: flat;
            // one agent: 4 096 is 1 % worse than 2 048 at 65 536 cars and 2.5 % worse at 32 768; profiles/r04_scan_stores.txt N)
            const int tail = std::min(a.agents >= 2 ? 4096 : 2048, a.n_cars / 2);
            spec = {{-1, 0}, {tail, 2}};
        }
        int fixed = 0;
        for (auto &x : spec) if (x.cars >= 0) { x.cars -= x.cars % SCAN_WAVES; fixed += x.cars; }
        // a list written for the step's car count may not fit a function-level scan of fewer poses: whole cars then
        if (fixed > a.n_cars) { spec = {{-1, 0}}; fixed = 0; }
        bool star = false;
        for (auto &x : spec) if (x.cars < 0 && !star) { x.cars = a.n_cars - fixed; star = true; }
        if (!star) spec.push_back({a.n_cars - fixed, 0});
        for (auto &x : spec) if (x.cars > 0) stv.push_back(x);
        // a workgroup never mixes two stages: every stage's wave count is a multiple of SCAN_WAVES
        for (size_t i = 0; i + 1 < stv.size(); i++) {
            const int w = stv[i].cars << stv[i].lg;
            if (w % SCAN_WAVES) { stv.assign(1, {a.n_cars, 0}); break; }
        }
    }
    if (stv.size() > (size_t)SCAN_MAX_STAGES) stv.assign(1, {a.n_cars, 0});
    a.n_stages = (int)stv.size();
    // what the kernel assumes about the stage list, checked here where a mistake costs an error code instead of a
    // wave -> car mapping that runs off the argument block
    if (a.n_stages < 1 || a.n_stages > SCAN_MAX_STAGES) return fail(F110_E_INVALID, "scan launch: %d stages (1..%d)", a.n_stages, SCAN_MAX_STAGES);
    {
        long long cars = 0;
        for (const St &x : stv) {
            if (x.cars < 0 || x.lg < 0 || x.lg > SCAN_MAX_LOG2W) return fail(F110_E_INVALID, "scan launch: stage (%d cars, 2^%d waves per car) out of range", x.cars, x.lg);
            cars += x.cars;
        }
        if (cars != a.n_cars) return fail(F110_E_INVALID, "scan launch: the stages cover %lld cars, the launch has %d", cars, a.n_cars);
    }
    for (int i = 0; i < 8; i++) { a.stage_cars[i] = i < a.n_stages ? stv[i].cars : 0; a.stage_log2w[i] = i < a.n_stages ? stv[i].lg : 0; }
#if defined(F110_TIMELINE)
    a.timeline = timeline_buffer();
#endif
    // the step's scan with streaming stores, except in very large launches (profiles/r04_scan_stores.txt L);
    // F110_SCAN_STORES=plain|stream overrides (A/B runs)
    static const char *stores_env = getenv("F110_SCAN_STORES");
    const bool plain = stores_env ? strcmp(stores_env, "plain") == 0 : a.n_cars > 327680;
    return !a.state ? launch_scan_t<0>(kind, a, st, ev0, ev1) : plain ? launch_scan_t<2>(kind, a, st, ev0, ev1) : launch_scan_t<1>(kind, a, st, ev0, ev1);
}

static void fill_scan_args(const f110_handle *h, ScanArgs &s, int reset_only)
{
    const f110_config &c = h->cfg;
    const f110_buffers &b = h->bufs;
    s.maps = h->d_maps; s.env_map = h->multi ? h->d_env_map : nullptr; s.scan = scan_dev(h); s.n_cars = c.num_envs * c.num_agents; s.agents = c.num_agents;
    s.pose_src = b.state; s.pose_stride = 7; s.yaw_off = 4;
    s.state = b.state; s.noise_step = b.noise_step; s.chunk_beam0 = h->d_chunk0;
    s.side = h->d_side; s.side_max = h->side_max;
    s.noise_base = h->d_noise; s.noise_cap = (int)h->noise_cap; s.noise_mask = (int)(h->noise_cap - 1); s.noise_slots = h->noise_slots;
    s.env_noise = h->multi_noise ? h->d_env_noise : nullptr; s.dev_err = h->d_err;
    s.order = (h->scan_order && !h->multi) ? h->scan_order : nullptr; // (a workgroup stages one LUT: car order when maps differ)
    s.wg_single = h->multi && h->wg_single; s.n_maps = F110_MAX_MAPS;
    if (h->per_env_noise) { s.noise_base = h->d_env_rows; s.noise_cap = 1; s.noise_mask = 0; s.noise_slots = c.num_envs; s.env_noise = h->d_env_ident; }
    s.beam_cosines = h->d_beam_cosines; s.ttc_thresh = c.ttc_thresh;
    s.in_collision = b.in_collision; s.pending_reset = b.pending_reset; s.reset_only = reset_only;
    s.out_f32 = b.scans; s.out_f64 = b.scans_f64; s.lookups = b.lookups;
}

// Instrumentation follows the measurement aid's sampling: while f110_profile_begin is active, the per-car lookup counters
// are only fed by the steps that also carry the event pair (an atomic per wave costs 2.5 % of a 65 536-env step,
// profiles/r03_event_cost.txt), so bytes and time of the roofline come from the same launches.
static void sample_lookups(const f110_handle *h, bool sampled_step, ScanArgs &s)
{
    if (h->prof_on && !sampled_step) s.lookups = nullptr;
}

// The step of every env: dynamics_kernel -> scan_kernel -> env_kernel, or for A > 1 -> post_scan_kernel (env bookkeeping and
// the opponents' set-up side by side) -> opp_apply_kernel.  (Two other
// forms -- a scan that also closes the step of a one-agent env, and a workgroup per car with a shared beam queue -- were
// built, held to ==, measured slower at every size and removed: tools/variants/car_group_and_closing_scan.patch,
// profiles/r03_step_forms.txt.)
static int run_step(f110_handle *h, const double *actions, int reset_only, const Sink &st)
{
    const f110_config &c = h->cfg;
    const f110_buffers &b = h->bufs;
    const int N = c.num_envs * c.num_agents;
    const bool prof = h->prof_on && !st.record && (h->prof_seq++ % h->prof_every) == h->prof_every / 2 && (size_t)(2 * h->prof_n + 1) < h->prof_ev.size();
    hipEvent_t ev0 = prof ? h->prof_ev[2 * h->prof_n] : nullptr, ev1 = prof ? h->prof_ev[2 * h->prof_n + 1] : nullptr;
    int rc;

    if (h->per_env_noise) {
        // the row every env's scan is about to add (row `pending ? 0 : noise_step`), from the env's own generator
        NoiseGenArgs g;
        memset(&g, 0, sizeof(g));
        g.gen = h->d_env_gen; g.seeds = h->d_env_seed; g.base = h->d_env_rows; g.mask = 0; g.cap = 1; g.nb = c.num_beams;
        g.pcg_tab = h->d_pcg_tab; g.env_row = b.noise_step; g.env_row_stride = c.num_agents; g.n_env = c.num_envs;
        g.reset_only = reset_only; g.env_pending = b.pending_reset;
        if ((rc = emit(st, (const void *)&noise_rows_kernel, dim3((c.num_envs + 3) / 4), dim3(256), 0, g))) return rc;
    }
    {
        DynArgs d;
        d.n_cars = N; d.agents = c.num_agents; d.state = b.state; d.steer_buf = b.steer_buf; d.steer_cnt = b.steer_cnt;
        d.noise_step = b.noise_step; d.actions = actions; d.spawn = b.spawn; d.pending_reset = b.pending_reset;
        d.was_pending = h->d_was_pending; d.reset_only = reset_only; d.pose_snap = b.pose_snap; d.in_collision = b.in_collision; d.params = h->d_params; d.env_params = h->multi_params ? h->d_env_params : nullptr; d.param_slots = h->param_slots; d.dev_err = h->d_err; d.noise = h->d_noise_desc;
        d.time_step = c.timestep; d.integrator = c.integrator;
        if ((rc = emit(st, (const void *)&dynamics_kernel, dim3((N + 255) / 256), dim3(256), 0, d))) return rc;
    }

    // the scan (the launch the measurement aid brackets)
    {
        ScanArgs s;
        memset(&s, 0, sizeof(s));
        fill_scan_args(h, s, reset_only);
        sample_lookups(h, prof, s);
        if (h->multi && !(h->ident && h->pow2)) {
            // env blocks on maps of different kinds: one launch per run of envs of one kind, so that a single map with an
            // odd resolution or a rotated origin does not put every car on the general instantiation
            int e0 = 0;
            rc = F110_OK;
            while (e0 < c.num_envs && !rc) {
                const f110_handle::MapSlot &s0 = h->slots[h->h_env_map[e0]];
                int e1 = e0 + 1;
                while (e1 < c.num_envs && h->slots[h->h_env_map[e1]].ident == s0.ident && h->slots[h->h_env_map[e1]].pow2 == s0.pow2) e1++;
                ScanArgs sub = s;
                sub.car_base = e0 * c.num_agents;
                sub.n_cars = (e1 - e0) * c.num_agents;
                const MapKind kind{s0.ident, s0.pow2};
                rc = launch_scan(h, sub, st, e0 == 0 ? ev0 : nullptr, e0 == 0 ? ev1 : nullptr, &kind);
                e0 = e1;
            }
        } else rc = launch_scan(h, s, st, ev0, ev1);
    }
    if (rc) return rc;
    if (prof) h->prof_n++;

    EnvArgs e;
    e.n_envs = c.num_envs; e.agents = c.num_agents; e.ego_idx = c.ego_idx; e.autoreset = c.autoreset;
    e.reset_only = reset_only; e.state = b.state; e.noise_step = b.noise_step; e.pose_snap = b.pose_snap; e.spawn = b.spawn;
    e.in_collision = b.in_collision; e.collisions = b.collisions; e.collision_idx = b.collision_idx;
    e.start_rot = b.start_rot; e.near_start = b.near_start; e.toggles = b.toggles; e.lap_counts = b.lap_counts;
    e.lap_times = b.lap_times; e.current_time = b.current_time; e.pending_reset = b.pending_reset; e.done = b.done; e.checkpoint_done = b.checkpoint_done;
    e.time_step = c.timestep; e.params = h->d_params; e.env_params = h->multi_params ? h->d_env_params : nullptr; e.param_slots = h->param_slots; e.dev_err = h->d_err;
    const int env_blocks = (c.num_envs + 127) / 128;
    if (c.num_agents == 1) return emit(st, (const void *)&env_kernel<true>, dim3(env_blocks), dim3(128), 0, e);

    // A > 1: env bookkeeping and the opponents' set-up side by side in one launch, then the ray cast
    PostScanArgs ps;
    memset(&ps, 0, sizeof(ps));
    ps.e = e; ps.env_blocks = env_blocks;
    OppArgs &o = ps.o;
    o.n_cars = N; o.agents = c.num_agents; o.nb = c.num_beams; o.state = b.state; o.pose_snap = b.pose_snap;
    o.in_collision = b.in_collision; o.scan_angles = h->d_scan_angles; o.beam_cs = h->d_beam_cs; o.params = h->d_params; o.env_params = h->multi_params ? h->d_env_params : nullptr;
    o.pending_reset = h->d_was_pending; o.reset_only = reset_only; o.scans32 = b.scans; o.scans64 = b.scans_f64;
    o.pairs = h->d_opp_pairs; o.param_slots = h->param_slots; o.dev_err = h->d_err;
    const int npairs = N * (c.num_agents - 1);
    if ((rc = emit(st, (const void *)&post_scan_kernel, dim3(env_blocks + (4 * npairs + 127) / 128), dim3(128), 0, ps))) return rc; // four lanes per pair
    return emit(st, (const void *)&opp_apply_kernel, dim3((int)(((long long)OPP_GROUP * N + 255) / 256)), dim3(256), 0, ps.o); // OPP_GROUP lanes per car
}

static Sink make_sink(f110_handle *h, hipStream_t st, std::vector<KernelLaunch> *record)
{
    Sink k;
    (void)h;
    k.st = st; k.record = record;
    return k;
}

static int check_ready(f110_handle *h, const char *who, bool launches_on_callers_stream = true)
{
    if (!h) return fail(F110_E_INVALID, "%s: null handle", who);
    if (launches_on_callers_stream)
        if (int rc = check_device(h, who)) return rc;
    if (!h->has_map) return fail(F110_E_NOMAP, "Map is not set for scan simulator.");
    if (!h->bound) return fail(F110_E_UNBOUND, "%s: f110_bind has not been called", who);
    return F110_OK;
}

__global__ void arm_reset_kernel(const double *poses, const uint8_t *mask, int n_envs, int agents, double *spawn,
                                 uint8_t *pending)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n_envs) return;
    if (mask && !mask[env]) return;
    for (int i = 0; i < agents * 3; i++) spawn[(size_t)env * agents * 3 + i] = poses[(size_t)env * agents * 3 + i];
    pending[env] = 1;
}

extern "C" int f110_reset(f110_handle *h, const double *poses, const uint8_t *mask, void *stream)
{
    int rc = check_ready(h, "f110_reset");
    if (rc) return rc;
    if (!poses) return fail(F110_E_INVALID, "Number of poses for reset does not match number of agents.");
    hipStream_t st = (hipStream_t)stream;
    const f110_config &c = h->cfg;
    hipLaunchKernelGGL(arm_reset_kernel, dim3((c.num_envs + 255) / 256), dim3(256), 0, st, poses, mask, c.num_envs,
                       c.num_agents, h->bufs.spawn, h->bufs.pending_reset);
    HIP_TRY(hipGetLastError());
    // the zero-action step of F110Env.reset; actions are not read for pending envs
    return run_step(h, nullptr, 1, make_sink(h, st));
}

extern "C" int f110_step(f110_handle *h, const double *actions, void *stream)
{
    int rc = check_ready(h, "f110_step");
    if (rc) return rc;
    if (!actions) return fail(F110_E_INVALID, "f110_step: null actions");
    return run_step(h, actions, 0, make_sink(h, (hipStream_t)stream));
}

// ---------------------------------------------------------------- one env's observation in one buffer
extern "C" int64_t f110_pack_env_size(f110_handle *h)
{
    if (!h) return 0;
    return (int64_t)h->cfg.num_agents * (11 + h->cfg.num_beams) + 2;
}

extern "C" int f110_pack_env(f110_handle *h, int32_t env, double *out_dev, void *stream)
{
    int rc = check_ready(h, "f110_pack_env");
    if (rc) return rc;
    if (!out_dev || env < 0 || env >= h->cfg.num_envs) return fail(env < 0 || env >= h->cfg.num_envs ? F110_E_INDEX : F110_E_INVALID, "f110_pack_env: env %d of %d, out %p", env, h->cfg.num_envs, (void *)out_dev);
    const f110_buffers &b = h->bufs;
    PackArgs a;
    a.env = env; a.agents = h->cfg.num_agents; a.nb = h->cfg.num_beams; a.state = b.state; a.collisions = b.collisions; a.lap_times = b.lap_times;
    a.lap_counts = b.lap_counts; a.toggles = b.toggles; a.current_time = b.current_time; a.done = b.done; a.scans64 = b.scans_f64; a.scans32 = b.scans;
    a.out = out_dev;
    const int n = (int)f110_pack_env_size(h);
    hipLaunchKernelGGL(pack_env_kernel, dim3(std::min((n + 255) / 256, 256)), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

// ---------------------------------------------------------------- the step as a HIP graph built by the library
struct f110_graph {
    f110_handle *h = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    int64_t epoch = 0;
    int nodes = 0;
    std::vector<KernelLaunch> launches; // node argument blocks must outlive hipGraphAddKernelNode only, kept for clarity
};

extern "C" void f110_graph_destroy(f110_graph *g)
{
    if (!g) return;
    DeviceScope on_dev(g->h ? g->h->cfg.device : 0);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->cap) (void)hipStreamDestroy(g->cap);
    delete g;
}

extern "C" int f110_graph_create(f110_handle *h, const double *actions, int32_t how, f110_graph **out)
{
    int rc = check_ready(h, "f110_graph_create", false); // builds on the handle's device itself (ON_DEVICE below)
    if (rc) return rc;
    if (!actions || !out) return fail(F110_E_INVALID, "f110_graph_create: null argument");
    if (how != F110_GRAPH_NODES && how != F110_GRAPH_CAPTURE) return fail(F110_E_INVALID, "f110_graph_create: how = %d (0 kernel nodes, 1 stream capture)", how);
    ON_DEVICE(h->cfg.device);
    f110_graph *g = new (std::nothrow) f110_graph;
    if (!g) return fail(F110_E_INVALID, "f110_graph_create: out of host memory");
    g->h = h; g->epoch = h->epoch;
    const bool prof = h->prof_on;
    h->prof_on = false; // events cannot ride on graph nodes
    hipError_t e = hipSuccess;
    if (how == F110_GRAPH_NODES) {
        rc = run_step(h, actions, 0, make_sink(h, nullptr, &g->launches));
        if (!rc) {
            e = hipGraphCreate(&g->graph, 0);
            hipGraphNode_t prev = nullptr;
            for (size_t i = 0; e == hipSuccess && i < g->launches.size(); i++) {
                KernelLaunch &l = g->launches[i];
                void *params[1] = {(void *)l.args.data()};
                hipKernelNodeParams np;
                memset(&np, 0, sizeof(np));
                np.func = const_cast<void *>(l.func); np.gridDim = l.grid; np.blockDim = l.block; np.sharedMemBytes = l.shmem;
                np.kernelParams = params; np.extra = nullptr;
                hipGraphNode_t node = nullptr;
                e = hipGraphAddKernelNode(&node, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &np); // a chain: each kernel reads what the one before wrote
                prev = node;
            }
            g->nodes = (int)g->launches.size();
        }
    } else {
        e = hipStreamCreateWithFlags(&g->cap, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamBeginCapture(g->cap, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            rc = run_step(h, actions, 0, make_sink(h, g->cap));
            e = hipStreamEndCapture(g->cap, &g->graph);
            size_t n = 0;
            if (e == hipSuccess && hipGraphGetNodes(g->graph, nullptr, &n) == hipSuccess) g->nodes = (int)n;
        }
    }
    h->prof_on = prof;
    if (!rc && e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (rc || e != hipSuccess) {
        if (!rc) rc = fail(F110_E_HIP, "f110_graph_create: %s", hipGetErrorString(e));
        f110_graph_destroy(g);
        return rc;
    }
    *out = g;
    return F110_OK;
}

extern "C" int f110_graph_launch(f110_graph *g, void *stream)
{
    if (!g || !g->exec) return fail(F110_E_INVALID, "f110_graph_launch: null graph");
    if (g->epoch != g->h->epoch)
        return fail(F110_E_INVALID, "f110_graph_launch: the graph is stale (a table, map, binding or launch setting of the handle "
                                    "changed since f110_graph_create: f110_launch_epoch moved from %lld to %lld); create it again",
                    (long long)g->epoch, (long long)g->h->epoch);
    if (int rc = check_device(g->h, "f110_graph_launch")) return rc;
    HIP_TRY(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return F110_OK;
}

extern "C" int f110_graph_info(f110_graph *g, int32_t *nodes, const char *dot_path)
{
    if (!g) return fail(F110_E_INVALID, "f110_graph_info: null graph");
    if (nodes) *nodes = g->nodes;
    if (dot_path && *dot_path) HIP_TRY(hipGraphDebugDotPrint(g->graph, dot_path, 0));
    return F110_OK;
}

extern "C" int f110_set_scan_stages(f110_handle *h, const char *spec)
{
    if (!h) return fail(F110_E_INVALID, "f110_set_scan_stages: null handle");
    if (spec && *spec) {
        std::vector<StageSpec> parsed;
        const char *why = nullptr;
        if (!parse_stage_spec(spec, parsed, &why)) return fail(F110_E_INVALID, "f110_set_scan_stages: \"%s\": %s", spec, why);
        long long fixed = 0;
        for (const StageSpec &x : parsed) if (x.cars > 0) fixed += x.cars;
        if (fixed > (long long)h->cfg.num_envs * h->cfg.num_agents)
            return fail(F110_E_INVALID, "f110_set_scan_stages: \"%s\" names %lld cars, the handle has %d", spec, fixed, h->cfg.num_envs * h->cfg.num_agents);
    }
    h->stages = spec ? spec : "";
    h->epoch++;
    return F110_OK;
}

extern "C" int f110_set_scan_order(f110_handle *h, const int32_t *order_dev)
{
    if (!h) return fail(F110_E_INVALID, "f110_set_scan_order: null handle");
    if (h->scan_order == order_dev) return F110_OK;
    h->scan_order = order_dev;
    h->epoch++; // the scan takes the pointer by value: a captured step is stale (the array's CONTENTS may change under it)
    return F110_OK;
}

extern "C" int f110_launch_epoch(f110_handle *h, int64_t *epoch)
{
    if (!h || !epoch) return fail(F110_E_INVALID, "f110_launch_epoch: null argument");
    *epoch = h->epoch;
    return F110_OK;
}

// ---------------------------------------------------------------- measurement aid
static void prof_clear(f110_handle *h)
{
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    h->prof_ev.clear();
    h->prof_n = 0;
    h->prof_on = false;
}

extern "C" int f110_profile_every(f110_handle *h, int32_t every)
{
    if (!h || every < 1) return fail(F110_E_INVALID, "f110_profile_every: bad arguments");
    h->prof_every = every;
    return F110_OK;
}

extern "C" int f110_profile_begin(f110_handle *h, int32_t max_launches)
{
    if (!h || max_launches < 1 || max_launches > (1 << 20)) return fail(F110_E_INVALID, "f110_profile_begin: bad arguments");
    ON_DEVICE(h->cfg.device);
    prof_clear(h);
    h->prof_seq = 0;
    h->prof_ev.resize((size_t)2 * max_launches);
    for (auto &e : h->prof_ev) HIP_TRY(hipEventCreate(&e));
    h->prof_on = true;
    return F110_OK;
}

extern "C" int f110_profile_end(f110_handle *h, double *ms_total, int32_t *launches)
{
    if (!h || !ms_total || !launches) return fail(F110_E_INVALID, "f110_profile_end: null argument");
    if (!h->prof_on) return fail(F110_E_INVALID, "f110_profile_end: f110_profile_begin has not been called");
    double tot = 0;
    if (h->prof_n > 0) HIP_TRY(hipEventSynchronize(h->prof_ev[2 * h->prof_n - 1]));
    for (int i = 0; i < h->prof_n; i++) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
        tot += ms;
    }
    *ms_total = tot;
    *launches = h->prof_n;
    prof_clear(h);
    return F110_OK;
}

// ---------------------------------------------------------------- function-level entry points
extern "C" int f110_scan(f110_handle *h, const double *poses, int32_t n, double *out64, float *out32,
                         uint32_t *lookups, void *stream)
{
    if (!h || n < 0) return fail(F110_E_INVALID, "f110_scan: bad arguments");
    if (!h->has_map) return fail(F110_E_NOMAP, "Map is not set for scan simulator.");
    if (n == 0) return F110_OK;
    if (!poses || (!out64 && !out32)) return fail(F110_E_INVALID, "f110_scan: null pose or output pointer");
    if (int rc = check_device(h, "f110_scan")) return rc;
    ScanArgs s;
    memset(&s, 0, sizeof(s));
    s.maps = h->d_maps; s.n_maps = F110_MAX_MAPS; s.scan = scan_dev(h); s.n_cars = n; s.agents = 1;
    s.pose_src = poses; s.pose_stride = 3; s.yaw_off = 2;
    s.out_f32 = out32; s.out_f64 = out64; s.lookups = lookups; s.chunk_beam0 = h->d_chunk0;
    Sink k;
    k.st = (hipStream_t)stream;
    return launch_scan(h, s, k);
}

extern "C" int f110_update_pose(f110_handle *h, double *state, double *steer_buf, int32_t *steer_cnt,
                                const double *actions, int32_t n, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !state || !steer_buf || !steer_cnt || !actions || n < 0)
        return fail(F110_E_INVALID, "f110_update_pose: bad arguments");
    if (int rc = check_device(h, "f110_update_pose")) return rc;
    DynArgs d;
    memset(&d, 0, sizeof(d));
    d.n_cars = n; d.agents = 1; d.state = state; d.steer_buf = steer_buf; d.steer_cnt = steer_cnt; d.actions = actions;
    d.params = h->d_params; d.param_slots = h->param_slots; d.dev_err = h->d_err; d.time_step = h->cfg.timestep; d.integrator = h->cfg.integrator;
    hipLaunchKernelGGL(dynamics_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_vehicle_dynamics(f110_handle *h, const double *x, const double *u, int32_t n, int32_t kinematic,
                                     double *f, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !x || !u || !f || n < 0) return fail(F110_E_INVALID, "f110_vehicle_dynamics: bad arguments");
    if (int rc = check_device(h, "f110_vehicle_dynamics")) return rc;
    hipLaunchKernelGGL(rhs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, u, n, kinematic,
                       h->d_params, f);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_get_vertices(f110_handle *h, const double *poses, int32_t n, double *verts, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !poses || !verts || n < 0) return fail(F110_E_INVALID, "f110_get_vertices: bad arguments");
    if (int rc = check_device(h, "f110_get_vertices")) return rc;
    hipLaunchKernelGGL(vertices_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, poses, n,
                       h->h_params[0].v[P_LENGTH], h->h_params[0].v[P_WIDTH], verts);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_gjk_pairs(f110_handle *h, const double *va, const double *vb, int32_t n, uint8_t *hit, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !va || !vb || !hit || n < 0) return fail(F110_E_INVALID, "f110_gjk_pairs: bad arguments");
    if (int rc = check_device(h, "f110_gjk_pairs")) return rc;
    hipLaunchKernelGGL(gjk_pairs_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, va, vb, n, hit);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_collision_multiple(f110_handle *h, const double *verts, int32_t n, int32_t A, uint8_t *col,
                                       int32_t *cidx, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !verts || !col || !cidx || n < 0 || A < 1) return fail(F110_E_INVALID, "f110_collision_multiple: bad arguments");
    if (int rc = check_device(h, "f110_collision_multiple")) return rc;
    hipLaunchKernelGGL(collision_multiple_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, verts, n, A,
                       col, cidx);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_check_ttc(f110_handle *h, const double *scans, const double *vel, int32_t n, uint8_t *hit,
                              void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !scans || !vel || !hit || n < 0) return fail(F110_E_INVALID, "f110_check_ttc: bad arguments");
    if (int rc = check_device(h, "f110_check_ttc")) return rc;
    hipLaunchKernelGGL(ttc_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, scans, vel, n, h->cfg.num_beams,
                       h->d_beam_cosines, h->d_side, h->cfg.ttc_thresh, hit);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_ray_cast(f110_handle *h, const double *ego, const double *verts, int32_t n, double *scans,
                             int32_t *span, void *stream)
{
    if (h && n == 0) return F110_OK;
    if (!h || !ego || !verts || !scans || n < 0) return fail(F110_E_INVALID, "f110_ray_cast: bad arguments");
    if (int rc = check_device(h, "f110_ray_cast")) return rc;
    hipLaunchKernelGGL(ray_cast_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, ego, verts, n,
                       h->cfg.num_beams, h->d_scan_angles, h->d_beam_cs, scans, span);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_check_done(f110_handle *h, const double *poses, const double *start_poses, const double *start_rot,
                               const double *current_time, const uint8_t *collisions, int32_t n, int32_t num_agents,
                               int32_t ego_idx, uint8_t *near_start, int32_t *toggles, int32_t *lap_counts,
                               double *lap_times, uint8_t *done, uint8_t *checkpoint_done, void *stream)
{
    (void)h; // stateless (the strip width and the 0.1 threshold are constants of f110_env.py:216-231)
    if (n < 0 || num_agents < 1 || num_agents > F110_MAX_AGENTS) return fail(F110_E_INVALID, "f110_check_done: bad arguments");
    if (ego_idx < 0 || ego_idx >= num_agents) return fail(F110_E_INDEX, "f110_check_done: ego_idx %d out of range", ego_idx);
    if (n == 0) return F110_OK;
    if (!poses || !start_poses || !start_rot || !current_time || !collisions || !near_start || !toggles || !lap_counts ||
        !lap_times || !done)
        return fail(F110_E_INVALID, "f110_check_done: null pointer (only checkpoint_done is optional)");
    CheckDoneArgs a;
    a.n_envs = n; a.agents = num_agents; a.ego_idx = ego_idx; a.poses = poses; a.start = start_poses; a.start_rot = start_rot;
    a.current_time = current_time; a.collisions = collisions; a.near_start = near_start; a.toggles = toggles;
    a.lap_counts = lap_counts; a.lap_times = lap_times; a.done = done; a.checkpoint_done = checkpoint_done;
    hipLaunchKernelGGL(check_done_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}
