// f110_abi.hip -- C ABI (include/f110_hip.h) over the gfx950 kernels.
// Host side: handle, map pipeline (exact EDT -> u16 cell codes + fp64 LUT),
// table uploads, kernel launches on the caller's stream.  No allocation and no
// synchronisation inside f110_step / f110_reset.
#include "../../include/f110_hip.h"
#include "f110_kernels.h"
#include "f110_planner.h"
#include "f110_bitmap.h"
#include "f110_mapgen.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

using namespace f110;

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(F110_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Makes `dev` the calling thread's current device for the scope of one library call and restores the caller's own
// afterwards: a process that drives several GPUs (or several handles on different GPUs) keeps ITS current device across
// every call.  f110_step / f110_reset and the function-level entry points do not switch -- they launch on the caller's
// stream, which belongs to the caller's current device -- they check (check_device) and refuse a mismatch.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ON_DEVICE(dev)            \
    DeviceScope dev_scope_(dev);  \
    HIP_TRY(dev_scope_.err)

struct f110_handle {
    f110_config cfg;
    // Vehicle parameters, [slots][1 + A]: per params slot (the `params` one reference env was constructed with) entry 0 =
    // Simulator.params (GJK vertices, base_classes.py:542), entry 1 + i = RaceCar.params of agent i
    std::vector<Params> h_params;
    int param_slots = 1;
    Params *d_params = nullptr;
    int d_params_slots = 0;           // slots the device allocation holds
    int32_t *d_env_params = nullptr;  // dev [B] params slot of every env; passed to the kernels only when `multi_params`
    bool multi_params = false;
    OppPair *d_opp_pairs = nullptr;   // [N, A-1] opponent ray-cast scratch (never allocated in f110_step)
    uint8_t *d_was_pending = nullptr; // [B] pending_reset as the step's first kernel found it
    bool has_map = false, bound = false;
    // Bumped whenever a later f110_step would enqueue different kernels or by-value arguments than an earlier one
    // (a table re-allocated, another scan instantiation selected, buffers re-bound): f110_launch_epoch.
    int64_t epoch = 0;
    std::string stages;               // f110_set_scan_stages override ("" = F110_STAGES or the built-in choice)
    f110_buffers bufs;
    // device tables owned by the handle
    double *d_scan_angles = nullptr, *d_beam_cosines = nullptr, *d_side = nullptr;
    uint16_t *d_chunk0 = nullptr;
    double2 *d_cs = nullptr;          // interleaved {cos, sin} LUT (repeated, see upload_cs)
    int cs_len = 0;
    double2 *d_beam_cs = nullptr;     // {cos, sin}(scan_angles) for the opponent ray cast
    std::vector<double> h_sines, h_cosines;
    // Lidar noise (f110_noise.h): [noise_slots][cap][nb] noise rows, rows lo .. hi-1 present; the kernels
    // find it through d_noise_desc, whose address never changes
    double *d_noise = nullptr;        // [noise_slots][noise_cap][num_beams] noise rows (a ring per slot)
    double side_max = 0.0;            // largest finite side distance (the scan's pre-test for iTTC candidates)
    long long noise_cap = 0, noise_lo = 0, noise_hi = 0; // cap: rows per slot (a power of two); noise off: cap 1, hi = "infinity"
    int noise_slots = 1;
    bool noise_on = false;
    NoiseDesc *d_noise_desc = nullptr;
    struct NoiseSlot {
        int kind = 0;                 // 0 unset (zeros), 1 host-fed table, 2 generator
        std::vector<double> rows;     // host-fed: [T, nb]
        long long T = 0;              // rows the slot can serve: host-fed = table length, generator = rows produced
        NoiseGen seed;                // generator: the stream at row 0
    };
    NoiseSlot nslots[F110_MAX_NOISE_SLOTS];
    NoiseGen *d_noise_gen = nullptr;  // [F110_MAX_NOISE_SLOTS] device generator states
    int32_t *d_env_noise = nullptr;   // dev [B] noise slot of every env; passed only when `multi_noise`
    bool multi_noise = false;
    hipStream_t noise_stream = nullptr; // the generator runs here, beside the caller's stream (f110_noise_prefetch)
    hipEvent_t noise_ev = nullptr;
    long long noise_pending_hi = 0;   // rows a prefetch in flight on noise_stream will have produced (0: none in flight)
    // Ordering of the side stream behind the caller's: recorded on the caller's stream whenever the floor is raised (the steps
    // enqueued so far may still read the rows below it, whose ring places the next prefetch recycles) and whenever a generator
    // kernel is enqueued there (it reads and writes the same generator states); the next prefetch waits for it.
    hipEvent_t order_ev = nullptr;
    bool order_ev_set = false;
    // prepared raceline of f110_pure_pursuit (f110_pure_pursuit_prepare): grid of candidate lists (f110_planner.h PlanGrid)
    const double *plan_wp = nullptr; int plan_M = 0; bool plan_ok = false;
    PlanGrid plan_grid;
    uint8_t *d_plan_count = nullptr; uint16_t *d_plan_cand = nullptr;
    u128 *d_pcg_tab = nullptr;        // [2][65] powers and partial sums of the LCG multiplier (f110_noise.h NoiseGenArgs::pcg_tab)
    // per-env noise (f110_set_noise_per_env): every env its own generator and ONE row, produced in front of every step's scan
    bool per_env_noise = false;
    NoiseGen *d_env_gen = nullptr, *d_env_seed = nullptr;   // [num_envs]
    double *d_env_rows = nullptr;                           // [num_envs][num_beams]
    int32_t *d_env_ident = nullptr;                         // [num_envs] env -> slot = env
    NoiseMark *d_marks = nullptr;     // [noise_slots][marks_cap] generator state at every 64th row (f110_noise.h NoiseMark)
    long long marks_cap = 0;
    int marks_slots = 0;
    struct Retired { void *ptr; hipEvent_t ev; };
    std::vector<Retired> retired;     // old noise tables, freed once the work that may read them has drained
    uint32_t *d_err = nullptr;        // device error word (f110_device_errors)
    std::vector<double> h_side;       // side distances (host copy of d_side)
    // Maps.  Slot 0 is "the" map of the reference's API; further slots let blocks of envs of one shard run on
    // different maps (one handle standing in for many F110Env instances with their own map each).
    struct MapSlot {
        uint16_t *d_cells = nullptr, *d_cells_far = nullptr;
        double *d_lut = nullptr, *d_lut_lds = nullptr, *d_dt = nullptr;
        MapDev dev;                   // host copy of d_maps[slot]
        bool used = false, ident = false, pow2 = false;
    };
    MapSlot slots[F110_MAX_MAPS];
    MapDev *d_maps = nullptr;         // dev [F110_MAX_MAPS] descriptors read by scan_kernel
    int32_t *d_env_map = nullptr;     // dev [B] slot of every env; only passed to the kernel when `multi`
    std::vector<int32_t> h_env_map;   // host copy (all 0 until f110_assign_maps)
    bool multi = false;
    bool ident = false, pow2 = false; // AND over the used slots: selects the scan_kernel instantiation
    double theta_inc = 0;
    // measurement aid (f110_profile_begin/end)
    std::vector<hipEvent_t> prof_ev; // pairs: [2*i] before, [2*i+1] after the scan launch
    int prof_n = 0, prof_every = 1, prof_seq = 0; // events ride on every prof_every-th step's scan launch (the middle one of
                                                  // each run of prof_every steps: on a clock ramp the samples' mean is then the steps' mean)
    bool prof_on = false;
};

extern "C" const char *f110_last_error(void) { return g_err; }

// Launches go to the caller's stream, which belongs to the calling thread's CURRENT device: it must be the handle's.
// (Checked, not switched: hipGetDevice is a thread-local read; switching would cost two runtime calls per step and
// still leave the caller's stream on the wrong device.)
static int check_current_device(int dev, const char *who)
{
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != dev)
        return fail(F110_E_INVALID, "%s: the handle lives on device %d but the calling thread's current device is %d; make the "
                    "handle's device current (hipSetDevice / torch.cuda.device) and pass a stream of that device", who, dev, cur);
    return F110_OK;
}



static int check_device(const f110_handle *h, const char *who) { return check_current_device(h->cfg.device, who); }

// ---------------------------------------------------------------- exact squared EDT (host)
// Meijster, Roerdink, Hesselink (2000): two passes, integer arithmetic only, so
// resolution*sqrt(d2) reproduces scipy.ndimage.distance_transform_edt bit for bit
// (exact Euclidean distances; reference call site laser_models.py:52).
extern "C" int f110_edt_squared(const uint8_t *mask, int32_t H, int32_t W, uint32_t *d2)
{
    if (!mask || !d2 || H <= 0 || W <= 0 || H > 32768 || W > 32768) return fail(F110_E_INVALID, "f110_edt_squared: bad arguments (size 1..32768)");
    const int64_t INF = (int64_t)H + W + 1;
    std::vector<int64_t> g((size_t)H * W);
    bool any_zero = false;
    for (int x = 0; x < W; x++) {
        // distance along the column to the nearest zero cell
        g[x] = mask[x] ? INF : 0;
        for (int y = 1; y < H; y++) {
            size_t i = (size_t)y * W + x;
            g[i] = mask[i] ? (g[i - W] >= INF ? INF : g[i - W] + 1) : 0;
        }
        for (int y = H - 2; y >= 0; y--) {
            size_t i = (size_t)y * W + x;
            if (g[i + W] < g[i]) g[i] = g[i + W] + 1 < g[i] ? g[i + W] + 1 : g[i];
        }
    }
    for (size_t i = 0; i < (size_t)H * W; i++)
        if (!mask[i]) { any_zero = true; break; }
    if (!any_zero) return fail(F110_E_INVALID, "f110_edt_squared: map has no occupied cell");
    // columns without any occupied cell carry g = INF (> any real distance), which the
    // lower-envelope scan handles without special cases since INF^2 exceeds every candidate
    std::vector<int> s(W), t(W);
    for (int y = 0; y < H; y++) {
        const int64_t *gr = &g[(size_t)y * W];
        auto f = [&](int64_t x, int64_t i) { return (x - i) * (x - i) + gr[i] * gr[i]; };
        auto sep = [&](int64_t i, int64_t u) {
            int64_t num = u * u - i * i + gr[u] * gr[u] - gr[i] * gr[i];
            int64_t den = 2 * (u - i);
            int64_t q = num / den;
            if ((num % den != 0) && ((num < 0) != (den < 0))) q--; // floor division
            return q;
        };
        int q = 0;
        s[0] = 0;
        t[0] = 0;
        for (int u = 1; u < W; u++) {
            while (q >= 0 && f(t[q], s[q]) > f(t[q], u)) q--;
            if (q < 0) {
                q = 0;
                s[0] = u;
            } else {
                int64_t w = 1 + sep(s[q], u);
                if (w < W) {
                    q++;
                    s[q] = u;
                    t[q] = (int)w;
                }
            }
        }
        for (int u = W - 1; u >= 0; u--) {
            d2[(size_t)y * W + u] = (uint32_t)f(u, s[q]);
            if (u == t[q]) q--;
        }
    }
    return F110_OK;
}

// ---------------------------------------------------------------- handle
static void default_tables(const f110_config &c, std::vector<double> &sines, std::vector<double> &cosines,
                           std::vector<double> &ang, std::vector<double> &bcos, std::vector<double> &side)
{
    // laser_models.py:379-381: np.linspace(0, 2*pi, theta_dis) (endpoint included)
    sines.resize(c.theta_dis);
    cosines.resize(c.theta_dis);
    const double step = (2 * F110_PI - 0.0) / (c.theta_dis - 1);
    for (int i = 0; i < c.theta_dis; i++) {
        double th = i == c.theta_dis - 1 ? 2 * F110_PI : 0.0 + i * step;
        sines[i] = std::sin(th);
        cosines[i] = std::cos(th);
    }
    // base_classes.py:123-156
    ang.resize(c.num_beams);
    bcos.resize(c.num_beams);
    side.resize(c.num_beams);
    const double incr = c.fov / (c.num_beams - 1);
    const double dist_sides = c.params[P_WIDTH] / 2.;
    const double dist_fr = (c.params[P_LF] + c.params[P_LR]) / 2.;
    for (int i = 0; i < c.num_beams; i++) {
        double angle = -c.fov / 2. + i * incr;
        double to_side, to_fr;
        ang[i] = angle;
        bcos[i] = std::cos(angle);
        if (angle > 0) {
            if (angle < F110_PI / 2) { to_side = dist_sides / std::sin(angle); to_fr = dist_fr / std::cos(angle); }
            else { to_side = dist_sides / std::cos(angle - F110_PI / 2.); to_fr = dist_fr / std::sin(angle - F110_PI / 2.); }
        } else {
            if (angle > -F110_PI / 2) { to_side = dist_sides / std::sin(-angle); to_fr = dist_fr / std::cos(-angle); }
            else { to_side = dist_sides / std::cos(-angle - F110_PI / 2); to_fr = dist_fr / std::sin(-angle - F110_PI / 2); }
        }
        side[i] = to_side < to_fr ? to_side : to_fr;
    }
}

static int upload(double **dst, const double *src, size_t n)
{
    if (!*dst) HIP_TRY(hipMalloc((void **)dst, n * sizeof(double)));
    HIP_TRY(hipMemcpy(*dst, src, n * sizeof(double), hipMemcpyHostToDevice));
    return F110_OK;
}

// Order in which a car's beams are handed to idle lanes: chunks of 64 angularly
// adjacent beams (adjacent rays sample neighbouring cells, which keeps a wave's gathers
// on few cache lines), the chunks sorted so that rays along the car's longitudinal axis
// -- they run down the track and need the most march steps -- start first and the short
// side rays fill the tail (key: |sin| of the chunk's centre angle).  A trailing partial
// chunk goes last so that slot k maps to beam chunk0[k >> 6] + (k & 63).
static int set_beam_order(f110_handle *h)
{
    const int nb = h->cfg.num_beams;
    const double incr = h->cfg.fov / (nb - 1);
    const int nchunks = (nb + 63) / 64, nfull = nb / 64;
    std::vector<std::pair<double, int>> key;
    for (int c = 0; c < nfull; c++) {
        const double centre = -h->cfg.fov / 2. + (64 * c + 31.5) * incr;
        key.push_back({std::fabs(std::sin(centre)), 64 * c});
    }
    std::sort(key.begin(), key.end());
    std::vector<uint16_t> chunk0;
    for (auto &k : key) chunk0.push_back((uint16_t)k.second);
    if (nchunks > nfull) chunk0.push_back((uint16_t)(64 * nfull));
    if (!h->d_chunk0) HIP_TRY(hipMalloc((void **)&h->d_chunk0, MAX_CHUNKS * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(h->d_chunk0, chunk0.data(), chunk0.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return F110_OK;
}

// {cos, sin} of the beam angles (libm), used by the opponent ray cast's angle addition.
static int upload_beam_cs(f110_handle *h, const double *scan_angles)
{
    const int n = h->cfg.num_beams;
    std::vector<double2> cs(n);
    for (int i = 0; i < n; i++) { cs[i].x = std::cos(scan_angles[i]); cs[i].y = std::sin(scan_angles[i]); }
    if (!h->d_beam_cs) HIP_TRY(hipMalloc((void **)&h->d_beam_cs, n * sizeof(double2)));
    HIP_TRY(hipMemcpy(h->d_beam_cs, cs.data(), n * sizeof(double2), hipMemcpyHostToDevice));
    return F110_OK;
}

static int upload_params(f110_handle *h);
static int noise_init(f110_handle *h);

// scratch of the opponent ray cast: allocated here, never in f110_step
static int alloc_opp_pairs(f110_handle *h)
{
    if (h->cfg.num_agents < 2) return F110_OK;
    const size_t n = (size_t)h->cfg.num_envs * h->cfg.num_agents * (h->cfg.num_agents - 1);
    HIP_TRY(hipMalloc((void **)&h->d_opp_pairs, n * sizeof(OppPair)));
    HIP_TRY(hipMemset(h->d_opp_pairs, 0, n * sizeof(OppPair)));
    HIP_TRY(hipMalloc((void **)&h->d_was_pending, (size_t)h->cfg.num_envs));
    HIP_TRY(hipMemset(h->d_was_pending, 0, (size_t)h->cfg.num_envs));
    return F110_OK;
}

// The scan reads a beam's side distance only where the iTTC test could fire: scan value below (largest side distance +
// the candidate margin).  Non-finite entries can never make a candidate (the reference's comparison is false for them).
static void set_side_max(f110_handle *h)
{
    double m = 0.0;
    for (double v : h->h_side) if (std::isfinite(v) && v > m) m = v;
    h->side_max = m;
}

// (Re)builds the interleaved {cos, sin} device table from the host copies.
static int upload_cs(f110_handle *h)
{
    // repeated so that an un-wrapped index theta_index + b*increment stays inside:
    // start < theta_dis, span <= fov/(2 pi) * theta_dis * nb/(nb-1)
    const int td = h->cfg.theta_dis;
    const int reps = 2 + (int)std::ceil(std::fabs(h->cfg.fov) / (2 * F110_PI) * h->cfg.num_beams / (h->cfg.num_beams - 1.0));
    const int n = td * reps;
    std::vector<double2> cs(n);
    for (int i = 0; i < n; i++) { cs[i].x = h->h_cosines[i % td]; cs[i].y = h->h_sines[i % td]; }
    if (h->d_cs && h->cs_len != n) { (void)hipFree(h->d_cs); h->d_cs = nullptr; }
    if (!h->d_cs) HIP_TRY(hipMalloc((void **)&h->d_cs, n * sizeof(double2)));
    HIP_TRY(hipMemcpy(h->d_cs, cs.data(), n * sizeof(double2), hipMemcpyHostToDevice));
    h->cs_len = n;
    return F110_OK;
}

extern "C" int f110_create(const f110_config *cfg, f110_handle **out)
{
    if (!cfg || !out) return fail(F110_E_INVALID, "f110_create: null argument");
    if (cfg->num_envs < 1 || cfg->num_agents < 1 || cfg->num_agents > F110_MAX_AGENTS)
        return fail(F110_E_INVALID, "f110_create: num_envs=%d num_agents=%d out of range (agents 1..%d)",
                    cfg->num_envs, cfg->num_agents, F110_MAX_AGENTS);
    // Index arithmetic (audited in round 5): every offset that multiplies a car index by a row length (scans, state, pairs,
    // noise rows) is formed in 64 bits; what stays in 32 bits is the car count itself, wave / thread indices derived from it
    // (up to 8 waves per car, 4 lanes per (car, opponent) pair, 64 lanes per car) and offsets inside one car's row
    // (beams * 8 < 2^15).  Hence: cars <= 2^26 and car-opponent pairs <= 2^28.
    if ((long long)cfg->num_envs * cfg->num_agents > F110_MAX_CARS ||
        (long long)cfg->num_envs * cfg->num_agents * (cfg->num_agents - 1) > 4ll * F110_MAX_CARS)
        return fail(F110_E_INVALID, "f110_create: %d envs x %d agents: a handle steps at most %d cars (and %lld car-opponent pairs); shard the batch",
                    cfg->num_envs, cfg->num_agents, F110_MAX_CARS, 4ll * F110_MAX_CARS);
    if (cfg->num_beams < 2 || cfg->num_beams > 4096 || cfg->theta_dis < 2)
        return fail(F110_E_INVALID, "f110_create: num_beams=%d (2..4096) theta_dis=%d", cfg->num_beams, cfg->theta_dis);
    if (cfg->integrator != F110_RK4 && cfg->integrator != F110_EULER)
        return fail(F110_E_INVALID, "f110_create: invalid integrator %d (RK4=1, Euler=2)", cfg->integrator);
    if (cfg->ego_idx < 0 || cfg->ego_idx >= cfg->num_agents)
        return fail(F110_E_INDEX, "f110_create: ego_idx %d out of range", cfg->ego_idx);
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(F110_E_HIP, "f110_create: device %d not available (%d HIP devices)", cfg->device, ndev);
    ON_DEVICE(cfg->device);
    f110_handle *h = new (std::nothrow) f110_handle();
    if (!h) return fail(F110_E_INVALID, "f110_create: out of host memory");
    h->cfg = *cfg;
    {
        Params p0;
        memcpy(p0.v, cfg->params, sizeof(double) * P_COUNT);
        h->h_params.assign((size_t)cfg->num_agents + 1, p0); // slot 0: Simulator.params + every agent's RaceCar.params
    }
    memset(&h->bufs, 0, sizeof(h->bufs));
    for (auto &sl : h->slots) memset(&sl.dev, 0, sizeof(sl.dev));
    // laser_models.py:367-368
    const double angle_increment = cfg->fov / (cfg->num_beams - 1);
    h->theta_inc = cfg->theta_dis * angle_increment / (2. * F110_PI);
    std::vector<double> s, c, ang, bcos, side;
    default_tables(*cfg, s, c, ang, bcos, side);
    int rc;
    h->h_sines = s;
    h->h_cosines = c;
    h->h_side = side;
    set_side_max(h);
    if ((rc = upload_cs(h)) ||
        (rc = upload(&h->d_scan_angles, ang.data(), ang.size())) || (rc = upload_beam_cs(h, ang.data())) ||
        (rc = upload(&h->d_beam_cosines, bcos.data(), bcos.size())) ||
        (rc = upload(&h->d_side, side.data(), side.size())) || (rc = noise_init(h)) || (rc = set_beam_order(h)) || (rc = upload_params(h)) || (rc = alloc_opp_pairs(h))) {
        f110_destroy(h);
        return rc;
    }
    *out = h;
    return F110_OK;
}

extern "C" void f110_destroy(f110_handle *h)
{
    if (!h) return;
    DeviceScope on_dev(h->cfg.device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {h->d_cs, h->d_beam_cs, h->d_noise, h->d_noise_desc, h->d_noise_gen, h->d_env_noise, h->d_scan_angles, h->d_beam_cosines,
                    h->d_side, h->d_chunk0, h->d_params, h->d_env_params, h->d_opp_pairs, h->d_was_pending, h->d_maps, h->d_env_map, h->d_err};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &r : h->retired) { (void)hipFree(r.ptr); (void)hipEventDestroy(r.ev); }
    if (h->noise_ev) (void)hipEventDestroy(h->noise_ev);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    if (h->d_marks) (void)hipFree(h->d_marks);
    for (void *q : {(void *)h->d_plan_count, (void *)h->d_plan_cand, (void *)h->d_pcg_tab, (void *)h->d_env_gen, (void *)h->d_env_seed, (void *)h->d_env_rows, (void *)h->d_env_ident})
        if (q) (void)hipFree(q);
    if (h->noise_stream) (void)hipStreamDestroy(h->noise_stream);
    for (auto &sl : h->slots)
        for (void *p : {(void *)sl.d_cells, (void *)sl.d_cells_far, (void *)sl.d_lut, (void *)sl.d_lut_lds, (void *)sl.d_dt})
            if (p) (void)hipFree(p);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    delete h;
}

static int upload_params(f110_handle *h)
{
    const int A1 = h->cfg.num_agents + 1;
    // enqueued steps may still read the table
    HIP_TRY(hipDeviceSynchronize());
    if (h->d_params_slots < h->param_slots) {
        if (h->d_params) { (void)hipFree(h->d_params); h->d_params = nullptr; }
        HIP_TRY(hipMalloc((void **)&h->d_params, sizeof(Params) * (size_t)h->param_slots * A1));
        h->d_params_slots = h->param_slots;
        h->epoch++; // the kernels take the pointer by value
    }
    HIP_TRY(hipMemcpy(h->d_params, h->h_params.data(), sizeof(Params) * (size_t)h->param_slots * A1, hipMemcpyHostToDevice));
    return F110_OK;
}

extern "C" int f110_update_params(f110_handle *h, const double *p, int32_t agent_idx)
{
    if (!h || !p) return fail(F110_E_INVALID, "f110_update_params: null argument");
    if (agent_idx >= h->cfg.num_agents) return fail(F110_E_INDEX, "Index given is out of bounds for list of agents.");
    ON_DEVICE(h->cfg.device);
    const int A1 = h->cfg.num_agents + 1;
    for (int sl = 0; sl < h->param_slots; sl++)
        for (int i = 0; i < h->cfg.num_agents; i++)
            if (agent_idx < 0 || agent_idx == i) memcpy(h->h_params[(size_t)sl * A1 + 1 + i].v, p, sizeof(double) * P_COUNT);
    return upload_params(h);
}

// ---- per-env constructor arguments: params slots
static int check_params18(const double *p, const char *who)
{
    for (int i = 0; i < P_COUNT; i++)
        if (!std::isfinite(p[i])) return fail(F110_E_INVALID, "%s: parameter %d is not finite", who, i);
    return F110_OK;
}

extern "C" int f110_set_params_slots(f110_handle *h, const double *params, int32_t n_slots)
{
    if (!h || !params) return fail(F110_E_INVALID, "f110_set_params_slots: null argument");
    if (n_slots < 1 || n_slots > h->cfg.num_envs) return fail(F110_E_INDEX, "f110_set_params_slots: %d slots (1..num_envs = %d)", n_slots, h->cfg.num_envs);
    for (int sl = 0; sl < n_slots; sl++)
        if (int rc = check_params18(params + (size_t)sl * P_COUNT, "f110_set_params_slots")) return rc;
    ON_DEVICE(h->cfg.device);
    const int A1 = h->cfg.num_agents + 1;
    h->h_params.resize((size_t)n_slots * A1);
    for (int sl = 0; sl < n_slots; sl++)
        for (int i = 0; i < A1; i++) memcpy(h->h_params[(size_t)sl * A1 + i].v, params + (size_t)sl * P_COUNT, sizeof(double) * P_COUNT);
    const bool shrunk = n_slots < h->param_slots;
    h->param_slots = n_slots;
    if (shrunk && h->multi_params) { h->multi_params = false; h->epoch++; } // the assignment may name slots that are gone: all envs back on slot 0
    return upload_params(h);
}

extern "C" int f110_set_params_slot(f110_handle *h, int32_t slot, const double *p, int32_t agent_idx)
{
    if (!h || !p) return fail(F110_E_INVALID, "f110_set_params_slot: null argument");
    if (slot < 0 || slot >= h->cfg.num_envs) return fail(F110_E_INDEX, "f110_set_params_slot: slot %d outside 0..%d", slot, h->cfg.num_envs - 1);
    if (agent_idx >= h->cfg.num_agents) return fail(F110_E_INDEX, "Index given is out of bounds for list of agents.");
    if (int rc = check_params18(p, "f110_set_params_slot")) return rc;
    ON_DEVICE(h->cfg.device);
    const int A1 = h->cfg.num_agents + 1;
    if (slot >= h->param_slots) { // new slots start as copies of slot 0
        h->h_params.resize((size_t)(slot + 1) * A1);
        for (int sl = h->param_slots; sl <= slot; sl++)
            for (int i = 0; i < A1; i++) h->h_params[(size_t)sl * A1 + i] = h->h_params[i];
        h->param_slots = slot + 1;
    }
    for (int i = 0; i < A1; i++) {
        const bool sim = i == 0;
        if (agent_idx < 0 || (!sim && agent_idx == i - 1)) memcpy(h->h_params[(size_t)slot * A1 + i].v, p, sizeof(double) * P_COUNT);
    }
    return upload_params(h);
}

extern "C" int f110_assign_params(f110_handle *h, const int32_t *slot_of_env)
{
    if (!h) return fail(F110_E_INVALID, "f110_assign_params: null handle");
    const int B = h->cfg.num_envs;
    std::vector<int32_t> m(B, 0);
    bool multi = false;
    if (slot_of_env)
        for (int e = 0; e < B; e++) {
            if (slot_of_env[e] < 0 || slot_of_env[e] >= h->param_slots)
                return fail(F110_E_INDEX, "f110_assign_params: env %d uses params slot %d, the handle has %d", e, slot_of_env[e], h->param_slots);
            m[e] = slot_of_env[e];
            multi = multi || m[e] != 0;
        }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    if (!h->d_env_params) HIP_TRY(hipMalloc((void **)&h->d_env_params, sizeof(int32_t) * B));
    HIP_TRY(hipMemcpy(h->d_env_params, m.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    h->multi_params = multi;
    h->epoch++;
    return F110_OK;
}

extern "C" int f110_set_tables(f110_handle *h, const double *sines, const double *cosines, const double *ang,
                               const double *bcos, const double *side)
{
    if (!h) return fail(F110_E_INVALID, "f110_set_tables: null handle");
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // enqueued steps may still read the tables being replaced
    int rc = F110_OK;
    if (sines) h->h_sines.assign(sines, sines + h->cfg.theta_dis);
    if (cosines) h->h_cosines.assign(cosines, cosines + h->cfg.theta_dis);
    if ((sines || cosines) && (rc = upload_cs(h))) return rc;
    if (ang && ((rc = upload(&h->d_scan_angles, ang, h->cfg.num_beams)) || (rc = upload_beam_cs(h, ang)))) return rc;
    if (bcos && (rc = upload(&h->d_beam_cosines, bcos, h->cfg.num_beams))) return rc;
    h->epoch++;
    if (side) {
        if ((rc = upload(&h->d_side, side, h->cfg.num_beams))) return rc;
        h->h_side.assign(side, side + h->cfg.num_beams);
        set_side_max(h);
    }
    return rc;
}

// Publishes the device tables of a freshly built map in slot `slot` (both pipelines end here).
static int finish_map(f110_handle *h, int slot, int H, int W, int Hp, size_t n_tiled, double res, double ox, double oy, double oc,
                      double os, double oob, unsigned lut_len)
{
    f110_handle::MapSlot &sl = h->slots[slot];
    MapDev &m = sl.dev;
    m.cells = sl.d_cells; m.cells_far = sl.d_cells_far; m.lut = sl.d_lut; m.lut_lds = sl.d_lut_lds; m.dt = sl.d_dt;
    m.H = H; m.W = W; m.strip_bytes = (unsigned)Hp * 16u; m.cells_bytes = (unsigned)(n_tiled * sizeof(uint16_t)); m.res = res; m.rinv = 1.0 / res;
    m.ox = ox; m.oy = oy; m.oc = oc; m.os = os;
    m.wres = W * res; // width * resolution (laser_models.py:79)
    m.hres = H * res;
    m.oob = oob;      // dt[-1, -1]
    m.lut_len = lut_len;
    int e = 0;
    sl.pow2 = std::frexp(res, &e) == 0.5;
    sl.ident = (oc == 1.0 && os == 0.0);
    sl.used = true;
    if (!h->d_maps) {
        HIP_TRY(hipMalloc((void **)&h->d_maps, sizeof(MapDev) * F110_MAX_MAPS));
        HIP_TRY(hipMemset(h->d_maps, 0, sizeof(MapDev) * F110_MAX_MAPS));
    }
    HIP_TRY(hipMemcpy(h->d_maps + slot, &m, sizeof(MapDev), hipMemcpyHostToDevice));
    h->ident = h->pow2 = true;
    for (const auto &u : h->slots)
        if (u.used) { h->ident = h->ident && u.ident; h->pow2 = h->pow2 && u.pow2; }
    h->has_map = h->slots[0].used;
    h->epoch++;
    return F110_OK;
}

// Builds the device map from a host fp64 distance table (and, when known, its
// exact squared form).  Cells whose value is not resolution*sqrt(integer) keep
// the escape code and are served from the fp64 table.
static int install_map(f110_handle *h, int slot, const double *dt, const uint32_t *d2_or_null, int H, int W, double res,
                       double ox, double oy, double oc, double os)
{
    f110_handle::MapSlot &sl = h->slots[slot];
    const size_t n = (size_t)H * W;
    // padded table (one border cell on every side), 8-column strips: see MapDev
    const int Hp = map_rows_padded(H);
    const size_t n_tiled = map_cells(H, W);
    std::vector<uint16_t> cells(n_tiled, 0), cells_far(n_tiled, 0); // (border and padding: code 0 = LDS slot 0 = dt[-1, -1])
    // exact squared distance of every cell (ESC64: not of the form resolution*sqrt(integer))
    const uint64_t ESC64 = ~0ull;
    std::vector<uint64_t> d2v(n);
    for (size_t i = 0; i < n; i++) {
        uint64_t d2;
        if (d2_or_null) d2 = d2_or_null[i];
        else {
            double q = dt[i] / res;
            double r = std::nearbyint(q * q);
            d2 = (r >= 0 && r < 4.0e18) ? (uint64_t)r : ESC64;
            if (d2 != ESC64 && res * std::sqrt((double)d2) != dt[i]) d2 = ESC64;
        }
        d2v[i] = d2;
    }
    // Codes are RANKS among the distinct d2 values of this map (ascending), not d2 itself:
    // squared distances are sums of two squares, so the 1023 LDS slots reach d2 ~ 3 900
    // (62 cells) instead of 1 022 (32 cells) -- the middle of a 5 m wide road still hits LDS.
    std::vector<uint64_t> distinct(d2v);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    if (!distinct.empty() && distinct.back() == ESC64) distinct.pop_back();
    if (distinct.empty() || distinct.front() != 0) distinct.insert(distinct.begin(), 0); // rank 0 <-> 0.0 (parked rays)
    const size_t n_lut = std::max<size_t>(LUT_LDS, std::min<size_t>(distinct.size(), CODE_ESC)); // ranks 0..65534 are encodable
    std::vector<double> lut(n_lut, 0.0);
    for (size_t k = 0; k < n_lut && k < distinct.size(); k++) lut[k] = res * std::sqrt((double)distinct[k]);
    for (size_t i = 0; i < n; i++) {
        const size_t t = cell_elem((int)(i / W), (int)(i % W), Hp);
        size_t rank = CODE_ESC;
        if (d2v[i] != ESC64) rank = std::min<size_t>(std::lower_bound(distinct.begin(), distinct.end(), d2v[i]) - distinct.begin(), CODE_ESC);
        cells[t] = (uint16_t)cell_code((unsigned)rank);
        cells_far[t] = (uint16_t)rank;
    }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // the previous map may still be in use by enqueued steps
    if (sl.d_cells) { (void)hipFree(sl.d_cells); sl.d_cells = nullptr; }
    if (sl.d_cells_far) { (void)hipFree(sl.d_cells_far); sl.d_cells_far = nullptr; }
    if (sl.d_dt) { (void)hipFree(sl.d_dt); sl.d_dt = nullptr; }
    if (sl.d_lut) { (void)hipFree(sl.d_lut); sl.d_lut = nullptr; } // its length depends on the map
    sl.used = false;
    h->has_map = h->slots[0].used;
    HIP_TRY(hipMalloc((void **)&sl.d_cells, n_tiled * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&sl.d_dt, n * sizeof(double)));
    HIP_TRY(hipMemcpy(sl.d_cells, cells.data(), n_tiled * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&sl.d_cells_far, n_tiled * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(sl.d_cells_far, cells_far.data(), n_tiled * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sl.d_dt, dt, n * sizeof(double), hipMemcpyHostToDevice));
    int rc = upload(&sl.d_lut, lut.data(), lut.size());
    if (rc) return rc;
    std::vector<double> lut_lds(LUT_LDS);
    lut_lds[SLOT_OOB] = dt[n - 1];    // dt[-1, -1]: what code 0 (the border: a look-up outside the map) reads
    std::copy(lut.begin(), lut.begin() + LDS_RANKS, lut_lds.begin() + 1);
    lut_lds[SLOT_FAR] = -0.0;         // the far marker (OFF_FAR cells take the second table): see MapDev
    if ((rc = upload(&sl.d_lut_lds, lut_lds.data(), lut_lds.size()))) return rc;
    return finish_map(h, slot, H, W, Hp, n_tiled, res, ox, oy, oc, os, dt[n - 1], (unsigned)lut.size());
}

// ---------------------------------------------------------------- map pipeline on the device
namespace {
struct DevTemp { // frees its device scratch on every exit path
    std::vector<void *> ptrs;
    ~DevTemp() { for (void *p : ptrs) if (p) (void)hipFree(p); }
    template <typename T> hipError_t alloc(T **out, size_t n)
    {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, n * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return e;
    }
};
} // namespace

static int check_edt_size(int H, int W, const char *who)
{
    if (H < 1 || W < 1 || H > 32768 || W > 32768) return fail(F110_E_INVALID, "%s: map size %dx%d outside 1..32768", who, H, W);
    return F110_OK;
}

// exact squared EDT of a device mask into a device table; *max_d2_dev (optional) receives the largest value
static int edt_squared_device(const uint8_t *mask_dev, int H, int W, uint32_t *d2_dev, unsigned *g_scratch, unsigned *max_d2_dev,
                              hipStream_t st)
{
    hipLaunchKernelGGL(edt_columns_kernel, dim3((W + 255) / 256), dim3(256), 0, st, mask_dev, H, W, g_scratch);
    HIP_TRY(hipGetLastError());
    const size_t lds = (size_t)W * sizeof(unsigned);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)edt_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(edt_rows_kernel, dim3(H), dim3(256), lds, st, g_scratch, H, W, d2_dev, max_d2_dev);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_edt_squared_dev(const uint8_t *mask_dev, int32_t H, int32_t W, uint32_t *d2_dev, void *stream)
{
    if (!mask_dev || !d2_dev) return fail(F110_E_INVALID, "f110_edt_squared_dev: null pointer");
    int rc = check_edt_size(H, W, "f110_edt_squared_dev");
    if (rc) return rc;
    DevTemp tmp;
    unsigned *g = nullptr;
    HIP_TRY(tmp.alloc(&g, (size_t)H * W));
    if ((rc = edt_squared_device(mask_dev, H, W, d2_dev, g, nullptr, (hipStream_t)stream))) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); // the scratch is freed on return
    return F110_OK;
}

// Occupancy mask (device) -> all map tables, without leaving the GPU.  mask: nonzero = free.
static int install_map_occupancy_dev(f110_handle *h, int slot, const uint8_t *mask_dev, int H, int W, double res, double ox,
                                     double oy, double oc, double os)
{
    f110_handle::MapSlot &sl = h->slots[slot];
    const size_t n = (size_t)H * W;
    const int Hp = map_rows_padded(H);
    const size_t n_tiled = map_cells(H, W);
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // the previous map may still be in use by enqueued steps
    hipStream_t st = nullptr;
    DevTemp tmp;
    unsigned *g = nullptr, *d2 = nullptr, *maxv = nullptr, *bits = nullptr, *prefix = nullptr, *sums = nullptr;
    HIP_TRY(tmp.alloc(&g, n));
    HIP_TRY(tmp.alloc(&d2, n));
    HIP_TRY(tmp.alloc(&maxv, 2));
    HIP_TRY(hipMemsetAsync(maxv, 0, 2 * sizeof(unsigned), st));
    int rc = edt_squared_device(mask_dev, H, W, d2, g, maxv, st);
    if (rc) return rc;
    unsigned max_d2 = 0;
    HIP_TRY(hipMemcpy(&max_d2, maxv, sizeof(unsigned), hipMemcpyDeviceToHost));
    // ranks of the distinct d2 values: presence bitmap + exclusive prefix of its popcounts
    const int n_words = (int)(((size_t)max_d2 + 32) / 32);
    const int n_blocks = (n_words + SCAN_BLOCK_WORDS - 1) / SCAN_BLOCK_WORDS;
    HIP_TRY(tmp.alloc(&bits, (size_t)n_words));
    HIP_TRY(tmp.alloc(&prefix, (size_t)n_words));
    HIP_TRY(tmp.alloc(&sums, (size_t)n_blocks));
    HIP_TRY(hipMemsetAsync(bits, 0, (size_t)n_words * sizeof(unsigned), st));
    hipLaunchKernelGGL(d2_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d2, n, bits);
    hipLaunchKernelGGL(rank_block_sums_kernel, dim3(n_blocks), dim3(256), 0, st, bits, n_words, sums);
    hipLaunchKernelGGL(rank_scan_sums_kernel, dim3(1), dim3(64), 0, st, sums, n_blocks, maxv + 1);
    hipLaunchKernelGGL(rank_word_prefix_kernel, dim3(n_blocks), dim3(256), 0, st, bits, n_words, sums, prefix);
    HIP_TRY(hipGetLastError());
    // the handle's tables
    if (sl.d_cells) { (void)hipFree(sl.d_cells); sl.d_cells = nullptr; }
    if (sl.d_cells_far) { (void)hipFree(sl.d_cells_far); sl.d_cells_far = nullptr; }
    if (sl.d_dt) { (void)hipFree(sl.d_dt); sl.d_dt = nullptr; }
    if (sl.d_lut) { (void)hipFree(sl.d_lut); sl.d_lut = nullptr; }
    sl.used = false;
    h->has_map = h->slots[0].used;
    HIP_TRY(hipMalloc((void **)&sl.d_cells, n_tiled * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&sl.d_cells_far, n_tiled * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&sl.d_dt, n * sizeof(double)));
    const unsigned n_lut = CODE_ESC; // ranks 0..65534 are encodable; unused slots stay 0.0
    HIP_TRY(hipMalloc((void **)&sl.d_lut, (size_t)n_lut * sizeof(double)));
    HIP_TRY(hipMemsetAsync(sl.d_lut, 0, (size_t)n_lut * sizeof(double), st));
    hipLaunchKernelGGL(map_fill_border_kernel, dim3((unsigned)((n_tiled + 255) / 256)), dim3(256), 0, st, sl.d_cells, sl.d_cells_far, n_tiled);
    hipLaunchKernelGGL(map_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d2, H, W, Hp, bits, prefix, res, sl.d_cells,
                       sl.d_cells_far, sl.d_dt);
    hipLaunchKernelGGL(map_lut_kernel, dim3((n_words + 255) / 256), dim3(256), 0, st, bits, n_words, prefix, res, sl.d_lut, n_lut);
    HIP_TRY(hipGetLastError());
    // LDS image of the LUT: its first slots, with the two special ones (see MapDev)
    std::vector<double> lut_lds(LUT_LDS);
    double oob = 0;
    HIP_TRY(hipMemcpy(lut_lds.data() + 1, sl.d_lut, LDS_RANKS * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&oob, sl.d_dt + (n - 1), sizeof(double), hipMemcpyDeviceToHost));
    lut_lds[SLOT_OOB] = oob;
    lut_lds[SLOT_FAR] = -0.0;
    if ((rc = upload(&sl.d_lut_lds, lut_lds.data(), lut_lds.size()))) return rc;
    return finish_map(h, slot, H, W, Hp, n_tiled, res, ox, oy, oc, os, oob, n_lut);
}


extern "C" int f110_track_mask(const double *pts_dev, int32_t n_pts, int32_t closed, int32_t H, int32_t W, double x0, double y0,
                               double pixel, double offset, double half_stroke, uint8_t *mask_dev, void *stream)
{
    if (!pts_dev || !mask_dev) return fail(F110_E_INVALID, "f110_track_mask: null pointer");
    if (n_pts < 2 || H < 1 || W < 1 || H > 32768 || W > 32768 || !(pixel > 0) || !(half_stroke >= 0))
        return fail(F110_E_INVALID, "f110_track_mask: bad arguments (n_pts=%d, %dx%d, pixel=%g)", n_pts, H, W, pixel);
    hipLaunchKernelGGL(track_mask_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, pts_dev, n_pts,
                       closed ? 1 : 0, H, W, x0, y0, pixel, offset, half_stroke, mask_dev);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

static int check_map_args(f110_handle *h, const void *p, int H, int W, double res, const char *who)
{
    if (!h || !p) return fail(F110_E_INVALID, "%s: null argument", who);
    if (H < 1 || W < 1 || (int64_t)(H + 10) * (W + 10) > (int64_t)1 << 30 || H + 10 >= (1 << 20)) return fail(F110_E_INVALID, "%s: bad map size %dx%d", who, H, W);
    if (!(res > 0) || !std::isfinite(res)) return fail(F110_E_INVALID, "%s: bad resolution %g", who, res);
    return F110_OK;
}

static int check_slot(f110_handle *h, int slot, const char *who)
{
    if (!h) return fail(F110_E_INVALID, "%s: null handle", who);
    if (slot < 0 || slot >= F110_MAX_MAPS) return fail(F110_E_INDEX, "%s: map slot %d outside 0..%d", who, slot, F110_MAX_MAPS - 1);
    return F110_OK;
}

extern "C" int f110_set_map_slot_occupancy(f110_handle *h, int32_t slot, const uint8_t *mask, int32_t H, int32_t W, double res,
                                           double ox, double oy, double oc, double os)
{
    int rc = check_slot(h, slot, "f110_set_map_slot_occupancy");
    if (rc || (rc = check_map_args(h, mask, H, W, res, "f110_set_map_occupancy")) || (rc = check_edt_size(H, W, "f110_set_map_occupancy")))
        return rc;
    const size_t n = (size_t)H * W;
    if (!memchr(mask, 0, n)) return fail(F110_E_INVALID, "f110_set_map_occupancy: map has no occupied cell");
    ON_DEVICE(h->cfg.device);
    DevTemp tmp;
    uint8_t *mask_dev = nullptr;
    HIP_TRY(tmp.alloc(&mask_dev, n));
    HIP_TRY(hipMemcpy(mask_dev, mask, n, hipMemcpyHostToDevice));
    return install_map_occupancy_dev(h, slot, mask_dev, H, W, res, ox, oy, oc, os);
}

extern "C" int f110_set_map_slot_occupancy_dev(f110_handle *h, int32_t slot, const uint8_t *mask_dev, int32_t H, int32_t W,
                                               double res, double ox, double oy, double oc, double os)
{
    int rc = check_slot(h, slot, "f110_set_map_slot_occupancy_dev");
    if (rc || (rc = check_map_args(h, mask_dev, H, W, res, "f110_set_map_occupancy_dev")) ||
        (rc = check_edt_size(H, W, "f110_set_map_occupancy_dev")))
        return rc;
    return install_map_occupancy_dev(h, slot, mask_dev, H, W, res, ox, oy, oc, os);
}

extern "C" int f110_set_map_occupancy(f110_handle *h, const uint8_t *mask, int32_t H, int32_t W, double res,
                                      double ox, double oy, double oc, double os)
{
    return f110_set_map_slot_occupancy(h, 0, mask, H, W, res, ox, oy, oc, os);
}

extern "C" int f110_set_map_occupancy_dev(f110_handle *h, const uint8_t *mask_dev, int32_t H, int32_t W, double res,
                                          double ox, double oy, double oc, double os)
{
    return f110_set_map_slot_occupancy_dev(h, 0, mask_dev, H, W, res, ox, oy, oc, os);
}

extern "C" int f110_set_map_dt(f110_handle *h, const double *dt, int32_t H, int32_t W, double res, double ox,
                               double oy, double oc, double os)
{
    int rc = check_map_args(h, dt, H, W, res, "f110_set_map_dt");
    if (rc) return rc;
    return install_map(h, 0, dt, nullptr, H, W, res, ox, oy, oc, os);
}

extern "C" int f110_get_map_slot_dt(f110_handle *h, int32_t slot, double *out)
{
    int rc = check_slot(h, slot, "f110_get_map_slot_dt");
    if (rc) return rc;
    if (!out) return fail(F110_E_INVALID, "f110_get_map_slot_dt: null argument");
    const f110_handle::MapSlot &sl = h->slots[slot];
    if (!sl.used) return fail(F110_E_NOMAP, "Map is not set for scan simulator.");
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipMemcpy(out, sl.d_dt, (size_t)sl.dev.H * sl.dev.W * sizeof(double), hipMemcpyDeviceToHost));
    return F110_OK;
}

extern "C" int f110_get_map_dt(f110_handle *h, double *out) { return f110_get_map_slot_dt(h, 0, out); }

// env -> map slot.  The cars of one scan workgroup (SCAN_WAVES consecutive cars) share the LDS copy of their
// map's LUT, so they must be on the same map: with blocks of envs per map that holds whenever a block's car
// count is a multiple of SCAN_WAVES.
extern "C" int f110_assign_maps(f110_handle *h, const int32_t *map_of_env)
{
    if (!h) return fail(F110_E_INVALID, "f110_assign_maps: null handle");
    const int B = h->cfg.num_envs, A = h->cfg.num_agents, N = B * A;
    std::vector<int32_t> m(B, 0);
    bool multi = false;
    if (map_of_env)
        for (int e = 0; e < B; e++) {
            const int k = map_of_env[e];
            if (k < 0 || k >= F110_MAX_MAPS || !h->slots[k].used)
                return fail(F110_E_INDEX, "f110_assign_maps: env %d uses map slot %d, which holds no map", e, k);
            m[e] = k;
            multi = multi || k != 0;
        }
    if (multi) {
        if (N % SCAN_WAVES) return fail(F110_E_INVALID, "f110_assign_maps: %d cars is not a multiple of %d", N, SCAN_WAVES);
        for (int c = 0; c < N; c += SCAN_WAVES)
            for (int j = 1; j < SCAN_WAVES; j++)
                if (m[(c + j) / A] != m[c / A])
                    return fail(F110_E_INVALID, "f110_assign_maps: cars %d and %d share a scan workgroup but not a map "
                                "(give every map a block of envs whose car count is a multiple of %d)", c, c + j, SCAN_WAVES);
    }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // enqueued steps may still read the table
    if (!h->d_env_map) HIP_TRY(hipMalloc((void **)&h->d_env_map, sizeof(int32_t) * B));
    HIP_TRY(hipMemcpy(h->d_env_map, m.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    h->h_env_map = m;
    h->multi = multi;
    h->epoch++;
    return F110_OK;
}

// ---------------------------------------------------------------- lidar noise (f110_noise.h)
__global__ void noise_publish_kernel(NoiseDesc *dst, NoiseDesc d) { *dst = d; }

static long long pow2_at_least(long long n)
{
    long long c = 1;
    while (c < n) c <<= 1;
    return c;
}

static void noise_reap(f110_handle *h, bool all)
{
    for (size_t i = 0; i < h->retired.size();) {
        if (all || hipEventQuery(h->retired[i].ev) == hipSuccess) {
            (void)hipFree(h->retired[i].ptr);
            (void)hipEventDestroy(h->retired[i].ev);
            h->retired.erase(h->retired.begin() + i);
        } else i++;
    }
}

// rows every active slot can serve
static void noise_recompute_hi(f110_handle *h)
{
    long long hi = -1;
    for (int sl = 0; sl < h->noise_slots; sl++) {
        const auto &ns = h->nslots[sl];
        if (ns.kind == 0) continue;
        hi = hi < 0 ? ns.T : std::min(hi, ns.T);
    }
    h->noise_on = hi >= 0;
    h->noise_hi = hi < 0 ? 0 : hi;
}

// the descriptor the kernels read, written in stream order
static int noise_publish(f110_handle *h, hipStream_t st)
{
    NoiseDesc d;
    if (h->per_env_noise) { // one row per env, produced on demand: every row counter is "in the table"
        d.base = h->d_env_rows; d.cap = 1; d.mask = 0; d.lo = 0; d.hi = 0x7fffffff; d.slots = h->cfg.num_envs; d.pad = 0;
        hipLaunchKernelGGL(noise_publish_kernel, dim3(1), dim3(1), 0, st, h->d_noise_desc, d);
        HIP_TRY(hipGetLastError());
        return F110_OK;
    }
    d.base = h->d_noise; d.cap = (int)h->noise_cap; d.mask = (int)(h->noise_cap - 1);
    d.lo = h->noise_on ? (int)std::min(h->noise_lo, (long long)0x7fffffff) : 0;
    d.slots = h->noise_slots; d.pad = 0;
    d.hi = h->noise_on ? (int)std::min(h->noise_hi, (long long)0x7fffffff) : 0x7fffffff; // noise off: every row is the row of zeros
    hipLaunchKernelGGL(noise_publish_kernel, dim3(1), dim3(1), 0, st, h->d_noise_desc, d);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

// Cold paths publish on the null stream and wait for it: the callers' streams do not synchronise with the null stream, and a
// later publish in stream order must not be overtaken by this one.
static int noise_publish_cold(f110_handle *h)
{
    int rc = noise_publish(h, nullptr);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(nullptr));
    return F110_OK;
}

// (Re)allocates the table for `slots` slots of `cap` rows, every pair {0, side}; rows lo .. hi-1 of the old table move
// over.  Cold path: synchronises the device, so nothing reads the old table any more and the new one is complete on return.
static int noise_resize(f110_handle *h, int slots, long long cap)
{
    const int nb = h->cfg.num_beams;
    if ((long long)slots * cap >= 0x7fffffffll) return fail(F110_E_INVALID, "noise table: %d slots x %lld rows exceed 2^31 rows", slots, cap);
    HIP_TRY(hipDeviceSynchronize());
    noise_reap(h, true);
    double *nt = nullptr;
    const size_t total = (size_t)slots * (size_t)cap;
    HIP_TRY(hipMalloc((void **)&nt, total * nb * sizeof(double)));
    {
        const long long items = (long long)total * nb;
        hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, nullptr, (const double *)nullptr,
                           (long long)total, nb, nt, 0, (long long)total, (long long)0x7fffffffffffffffll);
    }
    if (h->d_noise && h->noise_on && h->noise_hi > h->noise_lo) {
        const int ms = std::min(slots, h->noise_slots);
        const long long items = (h->noise_hi - h->noise_lo) * nb * ms;
        hipLaunchKernelGGL(noise_move_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, nullptr, h->d_noise, h->noise_cap,
                           h->noise_cap - 1, nt, cap, cap - 1, ms, h->noise_lo, h->noise_hi, nb);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (h->d_noise) (void)hipFree(h->d_noise);
    h->d_noise = nt;
    h->noise_cap = cap;
    h->noise_slots = slots;
    h->epoch++; // the scan takes the table's base and size by value (ScanArgs::noise_base): a re-allocation is a new launch
    return noise_publish_cold(h);
}

static int noise_init(f110_handle *h)
{
    HIP_TRY(hipMalloc((void **)&h->d_noise_desc, sizeof(NoiseDesc)));
    HIP_TRY(hipMalloc((void **)&h->d_noise_gen, sizeof(NoiseGen) * F110_MAX_NOISE_SLOTS));
    HIP_TRY(hipMemset(h->d_noise_gen, 0, sizeof(NoiseGen) * F110_MAX_NOISE_SLOTS));
    HIP_TRY(hipMalloc((void **)&h->d_err, sizeof(uint32_t)));
    HIP_TRY(hipMemset(h->d_err, 0, sizeof(uint32_t)));
    HIP_TRY(hipStreamCreateWithFlags(&h->noise_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&h->noise_ev, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming));
    {   // M^j and 1 + M + ... + M^(j-1), j = 0 .. 64 (mod 2^128)
        typedef unsigned __int128 u128h;
        const u128h M = ((u128h)0x2360ED051FC65DA4ull << 64) | (u128h)0x4385DF649FCCF645ull;
        u128h tab[130];
        u128h pw = 1, sm = 0;
        for (int j = 0; j <= 64; j++) { tab[j] = pw; tab[65 + j] = sm; sm = sm * M + 1; pw *= M; }
        HIP_TRY(hipMalloc((void **)&h->d_pcg_tab, sizeof(tab)));
        HIP_TRY(hipMemcpy(h->d_pcg_tab, tab, sizeof(tab), hipMemcpyHostToDevice));
    }
    return noise_resize(h, 1, 1); // noise off: one row of zeros
}

// a prefetch in flight on the generator's stream becomes part of the table for work enqueued on `st` from now on
static int noise_absorb_pending(f110_handle *h, hipStream_t st)
{
    if (!h->noise_pending_hi) return F110_OK;
    HIP_TRY(hipStreamWaitEvent(st, h->noise_ev, 0));
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 2) h->nslots[sl].T = std::max(h->nslots[sl].T, h->noise_pending_hi);
    h->noise_pending_hi = 0;
    noise_recompute_hi(h);
    return noise_publish(h, st);
}

static bool noise_has_generators(const f110_handle *h)
{
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 2) return true;
    return false;
}

// room for the marks of rows 0 .. rows-1 of every slot (cold path when it grows: synchronises)
static int noise_marks_reserve(f110_handle *h, long long rows)
{
    const long long need = rows / NOISE_MARK_ROWS + 2;
    if (h->d_marks && h->marks_slots == h->noise_slots && need <= h->marks_cap) return F110_OK;
    long long cap = std::max<long long>(h->marks_cap, 1 << 12);
    while (cap < need) cap <<= 1;
    HIP_TRY(hipDeviceSynchronize());
    NoiseMark *nm = nullptr;
    HIP_TRY(hipMalloc((void **)&nm, sizeof(NoiseMark) * (size_t)cap * (size_t)h->noise_slots));
    HIP_TRY(hipMemset(nm, 0, sizeof(NoiseMark) * (size_t)cap * (size_t)h->noise_slots));
    if (h->d_marks && h->marks_cap > 0)
        for (int sl = 0; sl < std::min(h->marks_slots, h->noise_slots); sl++)
            HIP_TRY(hipMemcpy(nm + (size_t)sl * cap, h->d_marks + (size_t)sl * h->marks_cap, sizeof(NoiseMark) * (size_t)h->marks_cap, hipMemcpyDeviceToDevice));
    if (h->d_marks) (void)hipFree(h->d_marks);
    h->d_marks = nm; h->marks_cap = cap; h->marks_slots = h->noise_slots;
    return F110_OK;
}

// Brings every generator slot to r1 rows (a multiple of 64), 64 rows per launch: every launch leaves the mark of the row it
// starts at (f110_noise.h NoiseMark), so that dropped rows can be produced again without rewinding the stream.
static int noise_launch_generator(f110_handle *h, long long r1, hipStream_t st)
{
    long long have = r1;
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 2) have = std::min(have, std::max(h->nslots[sl].T, h->noise_pending_hi));
    int rc = noise_marks_reserve(h, r1);
    if (rc) return rc;
    NoiseGenArgs g;
    memset(&g, 0, sizeof(g));
    g.gen = h->d_noise_gen; g.base = h->d_noise; g.mask = h->noise_cap - 1; g.cap = h->noise_cap; g.lo = h->noise_lo;
    g.nb = h->cfg.num_beams; g.marks = h->d_marks; g.marks_cap = h->marks_cap; g.redo = 0; g.chunk0 = 0; g.pcg_tab = h->d_pcg_tab;
    for (long long r = (have / NOISE_MARK_ROWS + 1) * NOISE_MARK_ROWS; ; r += NOISE_MARK_ROWS) {
        g.r1 = std::min(r, r1);
        hipLaunchKernelGGL(noise_rows_kernel, dim3(h->noise_slots), dim3(64), 0, st, g);
        if (r >= r1) break;
    }
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

// Rows [lo, hi) of every generator slot are produced AGAIN from the marks (they were dropped from the ring when the floor rose):
// one wavefront per slot and 64 rows, all at once -- the generators stay where they are.
static int noise_redo_rows(f110_handle *h, long long lo, long long hi, hipStream_t st)
{
    if (hi <= lo) return F110_OK;
    NoiseGenArgs g;
    memset(&g, 0, sizeof(g));
    g.gen = h->d_noise_gen; g.base = h->d_noise; g.mask = h->noise_cap - 1; g.cap = h->noise_cap; g.lo = lo; g.r1 = hi;
    g.nb = h->cfg.num_beams; g.marks = h->d_marks; g.marks_cap = h->marks_cap; g.redo = 1; g.chunk0 = lo / NOISE_MARK_ROWS; g.pcg_tab = h->d_pcg_tab;
    const long long chunks = (hi + NOISE_MARK_ROWS - 1) / NOISE_MARK_ROWS - g.chunk0;
    for (long long c0 = 0; c0 < chunks; c0 += 32768) { // (grid.y <= 65535)
        NoiseGenArgs gg = g;
        gg.chunk0 = g.chunk0 + c0;
        hipLaunchKernelGGL(noise_rows_kernel, dim3(h->noise_slots, (unsigned)std::min<long long>(32768, chunks - c0)), dim3(64), 0, st, gg);
    }
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

// every generator slot restarts at row 0 (its seed state); rows below the floor will be skipped, not stored
static int noise_restart_generators(f110_handle *h)
{
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    h->noise_pending_hi = 0;
    HIP_TRY(hipDeviceSynchronize());
    for (int sl = 0; sl < h->noise_slots; sl++) {
        auto &ns = h->nslots[sl];
        if (ns.kind != 2) continue;
        ns.T = 0;
        HIP_TRY(hipMemcpy(h->d_noise_gen + sl, &ns.seed, sizeof(NoiseGen), hipMemcpyHostToDevice));
    }
    noise_recompute_hi(h);
    return F110_OK;
}

static int check_noise_slot(f110_handle *h, int slot, const char *who)
{
    if (!h) return fail(F110_E_INVALID, "%s: null handle", who);
    if (slot < 0 || slot >= F110_MAX_NOISE_SLOTS) return fail(F110_E_INDEX, "%s: noise slot %d outside 0..%d", who, slot, F110_MAX_NOISE_SLOTS - 1);
    return F110_OK;
}

static void leave_per_env_noise(f110_handle *h)
{
    if (!h->per_env_noise) return;
    (void)hipDeviceSynchronize();
    h->per_env_noise = false;
    h->epoch++;
}

extern "C" int f110_set_noise_slot(f110_handle *h, int32_t slot, const double *tbl, int64_t T)
{
    int rc = check_noise_slot(h, slot, "f110_set_noise_slot");
    if (rc) return rc;
    leave_per_env_noise(h);
    if (T < 1 || !tbl) return fail(F110_E_INVALID, "f110_set_noise_slot: bad table (T >= 1 rows; f110_set_noise_table(h, NULL, 0) switches noise off)");
    ON_DEVICE(h->cfg.device);
    const int nb = h->cfg.num_beams;
    if (h->noise_lo > 0) { h->noise_lo = 0; if ((rc = noise_restart_generators(h))) return rc; } // host-fed rows start at 0
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    HIP_TRY(hipDeviceSynchronize());
    auto &ns = h->nslots[slot];
    ns.kind = 1;
    HIP_TRY(hipMemset(h->d_noise_gen + slot, 0, sizeof(NoiseGen))); // (the slot may have held a generator)
    ns.rows.assign(tbl, tbl + (size_t)T * nb);
    ns.T = T;
    const int slots = std::max(h->noise_slots, slot + 1);
    const long long cap = std::max(h->noise_cap, pow2_at_least(T));
    if (slots != h->noise_slots || cap != h->noise_cap || !h->noise_on) {
        // (first table after "noise off": the one-row table makes way)
        const bool was_on = h->noise_on;
        if (!was_on) { h->noise_lo = 0; h->noise_hi = 0; }
        if ((rc = noise_resize(h, slots, std::max(cap, (long long)2)))) return rc;
    }
    {   // stage the rows on the device and place them in the slot's ring
        DevTemp tmp;
        double *stage = nullptr;
        HIP_TRY(tmp.alloc(&stage, (size_t)T * nb));
        HIP_TRY(hipMemcpy(stage, tbl, (size_t)T * nb * sizeof(double), hipMemcpyHostToDevice));
        const long long items = (long long)T * nb;
        hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, nullptr, (const double *)stage,
                           (long long)T, nb, h->d_noise, slot, h->noise_cap, h->noise_cap - 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    noise_recompute_hi(h);
    return noise_publish_cold(h);
}

extern "C" int f110_set_noise_table(f110_handle *h, const double *tbl, int64_t T)
{
    if (!h) return fail(F110_E_INVALID, "f110_set_noise_table: null handle");
    if (T < 0 || (T > 0 && !tbl)) return fail(F110_E_INVALID, "f110_set_noise_table: bad table");
    if (T > 0) return f110_set_noise_slot(h, 0, tbl, T);
    // noise off: every slot forgets its table / generator
    ON_DEVICE(h->cfg.device);
    leave_per_env_noise(h);
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    h->noise_pending_hi = 0;
    for (auto &ns : h->nslots) { ns.kind = 0; ns.rows.clear(); ns.rows.shrink_to_fit(); ns.T = 0; }
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(h->d_noise_gen, 0, sizeof(NoiseGen) * F110_MAX_NOISE_SLOTS));
    h->noise_on = false; h->noise_lo = 0; h->noise_hi = 0;
    if (h->multi_noise) { h->multi_noise = false; h->epoch++; }
    return noise_resize(h, 1, 1);
}

extern "C" int f110_set_noise_generator(f110_handle *h, int32_t slot, const uint64_t *pcg64, double std_dev)
{
    int rc = check_noise_slot(h, slot, "f110_set_noise_generator");
    if (rc) return rc;
    if (!pcg64 || !(std_dev >= 0) || !std::isfinite(std_dev)) return fail(F110_E_INVALID, "f110_set_noise_generator: bad arguments");
    if (!(pcg64[2] & 1ull)) return fail(F110_E_INVALID, "f110_set_noise_generator: the PCG64 increment must be odd");
    ON_DEVICE(h->cfg.device);
    leave_per_env_noise(h);
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    HIP_TRY(hipDeviceSynchronize());
    auto &ns = h->nslots[slot];
    ns.kind = 2;
    ns.rows.clear();
    ns.T = 0;
    {   // the state whose output is the first raw value: one LCG step from NumPy's stored state (pcg64.h: step, then output)
        typedef unsigned __int128 u128h;
        const u128h M = ((u128h)0x2360ED051FC65DA4ull << 64) | (u128h)0x4385DF649FCCF645ull;
        const u128h st = ((u128h)pcg64[1] << 64) | pcg64[0], inc = ((u128h)pcg64[3] << 64) | pcg64[2];
        const u128h t = st * M + inc;
        memset(&ns.seed, 0, sizeof(ns.seed));
        ns.seed.t_lo = (unsigned long long)t; ns.seed.t_hi = (unsigned long long)(t >> 64);
        ns.seed.inc_lo = pcg64[2]; ns.seed.inc_hi = pcg64[3];
        ns.seed.std = std_dev; ns.seed.rows = 0; ns.seed.on = 1;
    }
    const int slots = std::max(h->noise_slots, slot + 1);
    if (slots != h->noise_slots || !h->noise_on || h->noise_cap < 2) {
        if (!h->noise_on) { h->noise_lo = 0; h->noise_hi = 0; }
        if ((rc = noise_resize(h, slots, std::max(h->noise_cap, (long long)F110_NOISE_INITIAL_ROWS)))) return rc;
    }
    // a new stream in one slot: every generator slot goes back to row 0, so that all of them stand at the same row again
    h->noise_lo = 0;
    h->noise_on = true;
    if ((rc = noise_restart_generators(h))) return rc;
    return noise_publish_cold(h);
}

// Every env its own stream (reference: every F110Env is constructed with its own `seed`, f110_env.py:102-105; its cars re-create
// default_rng(seed) at every reset, base_classes.py:117,202).  No table of rows per seed and no limit on the number of seeds:
// an env's generator state lives on the device and the row its scan adds is produced in front of the scan, every step
// (noise_rows_kernel in per-env mode, one wavefront per env).  pcg64 = host [num_envs][4] {state_lo, state_hi, inc_lo, inc_hi}.
extern "C" int f110_set_noise_per_env(f110_handle *h, const uint64_t *pcg64, double std_dev)
{
    if (!h || !pcg64 || !(std_dev >= 0) || !std::isfinite(std_dev)) return fail(F110_E_INVALID, "f110_set_noise_per_env: bad arguments");
    const int B = h->cfg.num_envs, nb = h->cfg.num_beams;
    std::vector<NoiseGen> seeds((size_t)B);
    typedef unsigned __int128 u128h;
    const u128h M = ((u128h)0x2360ED051FC65DA4ull << 64) | (u128h)0x4385DF649FCCF645ull;
    for (int e = 0; e < B; e++) {
        const uint64_t *w = pcg64 + (size_t)e * 4;
        if (!(w[2] & 1ull)) return fail(F110_E_INVALID, "f110_set_noise_per_env: env %d: the PCG64 increment must be odd", e);
        const u128h st = ((u128h)w[1] << 64) | w[0], inc = ((u128h)w[3] << 64) | w[2];
        const u128h t = st * M + inc; // the state whose output is the first raw value (pcg64.h: step, then output)
        NoiseGen &g = seeds[(size_t)e];
        memset(&g, 0, sizeof(g));
        g.t_lo = (unsigned long long)t; g.t_hi = (unsigned long long)(t >> 64); g.inc_lo = w[2]; g.inc_hi = w[3];
        g.std = std_dev; g.rows = 0; g.on = 1;
    }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    HIP_TRY(hipDeviceSynchronize());
    if (!h->d_env_gen) {
        HIP_TRY(hipMalloc((void **)&h->d_env_gen, sizeof(NoiseGen) * (size_t)B));
        HIP_TRY(hipMalloc((void **)&h->d_env_seed, sizeof(NoiseGen) * (size_t)B));
        HIP_TRY(hipMalloc((void **)&h->d_env_rows, sizeof(double) * (size_t)B * nb));
        HIP_TRY(hipMalloc((void **)&h->d_env_ident, sizeof(int32_t) * (size_t)B));
        std::vector<int32_t> id((size_t)B);
        for (int e = 0; e < B; e++) id[(size_t)e] = e;
        HIP_TRY(hipMemcpy(h->d_env_ident, id.data(), sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(h->d_env_seed, seeds.data(), sizeof(NoiseGen) * (size_t)B, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_env_gen, seeds.data(), sizeof(NoiseGen) * (size_t)B, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->d_env_rows, 0, sizeof(double) * (size_t)B * nb));
    h->per_env_noise = true;
    h->noise_on = true;
    h->epoch++;
    return noise_publish_cold(h);
}

extern "C" int f110_noise_prefetch(f110_handle *h, int64_t rows)
{
    if (!h) return fail(F110_E_INVALID, "f110_noise_prefetch: null handle");
    if (h->per_env_noise) return F110_OK;
    if (!h->noise_on || !noise_has_generators(h) || h->noise_pending_hi) return F110_OK;
    long long have = 0x7fffffffffffffffll;
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 2) have = std::min(have, h->nslots[sl].T);
    if (rows <= have) return F110_OK;
    const long long r1 = (rows + 63) & ~63ll;
    if (r1 - h->noise_lo > h->noise_cap) return F110_OK; // needs a larger table: f110_noise_ensure grows it when the rows are due
    ON_DEVICE(h->cfg.device);
    // The generator appends rows have .. r1-1 into ring places whose previous tenants lie below the floor.  Steps that were
    // enqueued BEFORE the floor was raised may still read those tenants, and a generator kernel enqueued on the caller's
    // stream (f110_noise_ensure) works on the same generator states: both recorded `order_ev` there, and this launch waits for it.
    if (h->order_ev_set) { HIP_TRY(hipStreamWaitEvent(h->noise_stream, h->order_ev, 0)); h->order_ev_set = false; }
    if (int rc = noise_launch_generator(h, r1, h->noise_stream)) return rc;
    HIP_TRY(hipEventRecord(h->noise_ev, h->noise_stream));
    h->noise_pending_hi = r1;
    return F110_OK;
}

extern "C" int f110_noise_ensure(f110_handle *h, int64_t rows, void *stream)
{
    if (!h) return fail(F110_E_INVALID, "f110_noise_ensure: null handle");
    if (h->per_env_noise || !h->noise_on || rows <= h->noise_hi) return F110_OK; // (per-env rows are produced by the step itself)
    if (int rc = check_device(h, "f110_noise_ensure")) return rc;
    hipStream_t st = (hipStream_t)stream;
    noise_reap(h, false);
    int rc = noise_absorb_pending(h, st);
    if (rc) return rc;
    if (rows <= h->noise_hi) return F110_OK;
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 1 && h->nslots[sl].T < rows)
            return fail(F110_E_INVALID, "f110_noise_ensure: noise slot %d is a host table of %lld rows, %lld are needed (upload a longer "
                        "table with f110_set_noise_slot, or use f110_set_noise_generator)", sl, h->nslots[sl].T, (long long)rows);
    const long long r1 = (rows + 63) & ~63ll;
    if (r1 - h->noise_lo > h->noise_cap) // the ring is too small for rows lo .. r1-1: a larger one (cold path, synchronises)
        if ((rc = noise_resize(h, h->noise_slots, pow2_at_least(std::max(2 * h->noise_cap, r1 - h->noise_lo))))) return rc;
    if ((rc = noise_launch_generator(h, r1, st))) return rc;
    for (int sl = 0; sl < h->noise_slots; sl++)
        if (h->nslots[sl].kind == 2) h->nslots[sl].T = r1;
    noise_recompute_hi(h);
    if ((rc = noise_publish(h, st))) return rc;
    HIP_TRY(hipEventRecord(h->order_ev, st)); // the next prefetch (side stream) runs behind this generator launch
    h->order_ev_set = true;
    return F110_OK;
}

extern "C" int f110_noise_set_floor(f110_handle *h, int64_t lo, void *stream)
{
    if (!h || lo < 0) return fail(F110_E_INVALID, "f110_noise_set_floor: bad arguments");
    if (h->per_env_noise || !h->noise_on || lo == h->noise_lo) return F110_OK;
    if (int rc = check_device(h, "f110_noise_set_floor")) return rc;
    if (lo > h->noise_lo) {
        for (int sl = 0; sl < h->noise_slots; sl++)
            if (h->nslots[sl].kind == 1) return fail(F110_E_INVALID, "f110_noise_set_floor: noise slot %d is a host table (rows are only dropped from generated noise)", sl);
        if (lo > h->noise_hi) return fail(F110_E_INVALID, "f110_noise_set_floor: floor %lld above the %lld rows produced", (long long)lo, h->noise_hi);
        h->noise_lo = lo;
        if (int rc = noise_publish(h, (hipStream_t)stream)) return rc;
        // the steps enqueued so far may read rows below the new floor: the prefetch that recycles their places waits for them
        HIP_TRY(hipEventRecord(h->order_ev, (hipStream_t)stream));
        h->order_ev_set = true;
        return F110_OK;
    }
    // The floor comes down (a car was reset while others run on): rows lo .. old floor - 1 are produced again, from the marks
    // the generators left every 64 rows -- one wavefront per slot and 64 rows, in the caller's stream; the generators themselves
    // stay where they are.  (Until round 5 every generator was rewound to its seed and re-ran the whole stream, one wavefront
    // per seed at ~15 us per row.)  The ring has to span floor .. rows produced: it grows if it must (cold path).
    hipStream_t st = (hipStream_t)stream;
    int rc = noise_absorb_pending(h, st);
    if (rc) return rc;
    const long long old_lo = h->noise_lo;
    if (h->noise_hi - lo > h->noise_cap)
        if ((rc = noise_resize(h, h->noise_slots, pow2_at_least(h->noise_hi - lo)))) return rc;
    h->noise_lo = lo;
    if ((rc = noise_redo_rows(h, lo, std::min(old_lo, h->noise_hi), st))) return rc;
    if ((rc = noise_publish(h, st))) return rc;
    HIP_TRY(hipEventRecord(h->order_ev, st));
    h->order_ev_set = true;
    return F110_OK;
}

extern "C" int f110_noise_info(f110_handle *h, int64_t *lo, int64_t *hi, int64_t *cap, int32_t *slots, int64_t *bytes)
{
    if (!h) return fail(F110_E_INVALID, "f110_noise_info: null handle");
    if (lo) *lo = h->noise_lo;
    if (hi) *hi = h->noise_on ? h->noise_hi : 0; // (rows a prefetch is still producing are not counted: f110_noise_ensure makes them readable)
    if (cap) *cap = h->noise_cap;
    if (slots) *slots = h->noise_slots;
    if (bytes) {
        long long b = (long long)h->noise_slots * h->noise_cap * h->cfg.num_beams * (long long)sizeof(double);
        noise_reap(h, false);
        *bytes = b * (1 + (long long)h->retired.size());
    }
    return F110_OK;
}

extern "C" int f110_noise_read(f110_handle *h, int32_t slot, int64_t row0, int64_t n_rows, double *out)
{
    int rc = check_noise_slot(h, slot, "f110_noise_read");
    if (rc) return rc;
    if (!out || n_rows < 0) return fail(F110_E_INVALID, "f110_noise_read: bad arguments");
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipStreamSynchronize(h->noise_stream));
    HIP_TRY(hipDeviceSynchronize());
    const long long hi = h->noise_on ? std::max(h->noise_hi, h->noise_pending_hi) : 0;
    if (slot >= h->noise_slots || row0 < h->noise_lo || row0 + n_rows > hi)
        return fail(F110_E_INDEX, "f110_noise_read: rows %lld..%lld of slot %d; the table holds rows %lld..%lld of %d slots", (long long)row0,
                    (long long)(row0 + n_rows - 1), slot, h->noise_lo, hi - 1, h->noise_slots);
    const int nb = h->cfg.num_beams;
    for (long long r = row0; r < row0 + n_rows; r++)
        HIP_TRY(hipMemcpy(out + (size_t)(r - row0) * nb, h->d_noise + ((size_t)slot * h->noise_cap + (size_t)(r & (h->noise_cap - 1))) * nb,
                          (size_t)nb * sizeof(double), hipMemcpyDeviceToHost));
    return F110_OK;
}

extern "C" int f110_assign_noise(f110_handle *h, const int32_t *slot_of_env)
{
    if (!h) return fail(F110_E_INVALID, "f110_assign_noise: null handle");
    const int B = h->cfg.num_envs;
    std::vector<int32_t> m(B, 0);
    bool multi = false;
    if (slot_of_env)
        for (int e = 0; e < B; e++) {
            const int k = slot_of_env[e];
            if (k < 0 || k >= h->noise_slots || (h->noise_on && h->nslots[k].kind == 0))
                return fail(F110_E_INDEX, "f110_assign_noise: env %d uses noise slot %d, which holds neither a table nor a generator", e, k);
            m[e] = k;
            multi = multi || k != 0;
        }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    if (!h->d_env_noise) HIP_TRY(hipMalloc((void **)&h->d_env_noise, sizeof(int32_t) * B));
    HIP_TRY(hipMemcpy(h->d_env_noise, m.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    h->multi_noise = multi;
    h->epoch++;
    return F110_OK;
}

#if defined(F110_BOUNDS)
// bounds-checked build only: one checked access that is out of range on purpose, so that a test can see the report arrive
__global__ void bounds_selftest_kernel(uint32_t *err, int idx, int len) { F110_BCHK(idx < len, BT_SELFTEST, err); }
extern "C" int f110_bounds_selftest(f110_handle *h)
{
    if (!h) return fail(F110_E_INVALID, "f110_bounds_selftest: null handle");
    ON_DEVICE(h->cfg.device);
    hipLaunchKernelGGL(bounds_selftest_kernel, dim3(1), dim3(1), 0, nullptr, h->d_err, 7, 7);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}
#endif

extern "C" int f110_device_errors(f110_handle *h, uint32_t *flags)
{
    if (!h || !flags) return fail(F110_E_INVALID, "f110_device_errors: null argument");
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(flags, h->d_err, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (*flags) HIP_TRY(hipMemset(h->d_err, 0, sizeof(uint32_t)));
    return F110_OK;
}

extern "C" int f110_bind(f110_handle *h, const f110_buffers *b)
{
    if (!h || !b) return fail(F110_E_INVALID, "f110_bind: null argument");
    const void *req[] = {b->state, b->steer_buf, b->steer_cnt, b->noise_step, b->spawn, b->start_rot,
                         b->near_start, b->toggles, b->current_time, b->pending_reset, b->scans,
                         b->pose_snap, b->collisions, b->collision_idx, b->in_collision, b->lap_counts,
                         b->lap_times, b->done};
    for (const void *p : req)
        if (!p) return fail(F110_E_INVALID, "f110_bind: a required buffer is NULL (only scans_f64, checkpoint_done and lookups are optional)");
    h->bufs = *b;
    h->bound = true;
    h->epoch++;
    return F110_OK;
}

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
    const dim3 grid((waves + SCAN_WAVES - 1) / SCAN_WAVES), block(SCAN_THREADS);
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

// ---------------------------------------------------------------- planner
// per device: the LDS a workgroup may use (queried once), the dynamic-LDS attribute already granted to
// pure_pursuit_kernel, and the two-entry raceline header {0, M} of the global-memory fallback of f110_pure_pursuit
struct DevLds { int max_bytes = -1; size_t pp_attr = 0; int32_t *hdr = nullptr; int hdr_m = -1; };
static DevLds &device_lds(int dev)
{
    static DevLds tab[64];
    DevLds &d = tab[dev & 63];
    if (d.max_bytes < 0) {
        int v = 0;
        d.max_bytes = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess ? v : 0;
        // the attribute may report the 64 KiB every kernel gets without asking; gfx950 grants 160 KiB per workgroup
        // through hipFuncAttributeMaxDynamicSharedMemorySize
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) == 0)
            d.max_bytes = std::max(d.max_bytes, 160 * 1024);
    }
    return d;
}

// {0, M} on the device for the single-raceline fallback.  Allocated once per device (not inside a stream capture: a
// caller that captures a policy with a long raceline makes one eager call first); the 8-byte upload is synchronous.
static int32_t *single_track_offsets(int dev, int M)
{
    DevLds &d = device_lds(dev);
    if (!d.hdr && hipMalloc((void **)&d.hdr, 2 * sizeof(int32_t)) != hipSuccess) { d.hdr = nullptr; return nullptr; }
    if (d.hdr_m != M) {
        const int32_t off[2] = {0, M};
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(d.hdr, off, sizeof(off), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        d.hdr_m = M;
    }
    return d.hdr;
}

// Builds the grid of candidate lists for one raceline (dev [M,3]) in the handle: a cold path (the raceline is copied to the host,
// ~0.1 s for the 783-point example raceline).  The caller promises to call it again when the raceline's values change; the pointer
// and M are what f110_pure_pursuit matches.  cell: edge of a grid cell in metres (0: 0.25); margin: how far around the raceline's
// bounding box the grid reaches (0: 3 m) -- poses beyond it are planned by the exhaustive search.
extern "C" int f110_pure_pursuit_prepare(f110_handle *h, const double *waypoints, int32_t M, double cell, double margin, void *stream)
{
    if (!h || !waypoints) return fail(F110_E_INVALID, "f110_pure_pursuit_prepare: null argument");
    if (int rc = check_device(h, "f110_pure_pursuit_prepare")) return rc;
    h->plan_ok = false;
    if (M < 2 || M > 65535) return fail(F110_E_INVALID, "f110_pure_pursuit_prepare: M=%d waypoints (2..65535)", M);
    if (!(cell >= 0) || !(margin >= 0) || !std::isfinite(cell) || !std::isfinite(margin)) return fail(F110_E_INVALID, "f110_pure_pursuit_prepare: bad cell / margin");
    if (cell == 0) cell = 0.25;
    if (margin == 0) margin = 3.0;
    std::vector<double> wp((size_t)M * 3);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    HIP_TRY(hipMemcpy(wp.data(), waypoints, wp.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int nseg = M - 1;
    double xl = 1e300, xh = -1e300, yl = 1e300, yh = -1e300;
    bool finite = true, degenerate = false;
    for (int i = 0; i < M; i++) {
        const double x = wp[3 * (size_t)i], y = wp[3 * (size_t)i + 1];
        finite = finite && std::isfinite(x) && std::isfinite(y);
        xl = std::min(xl, x); xh = std::max(xh, x); yl = std::min(yl, y); yh = std::max(yh, y);
    }
    if (!finite) return fail(F110_E_INVALID, "f110_pure_pursuit_prepare: the raceline has non-finite points");
    for (int i = 0; i < nseg; i++) {
        const double dx = wp[3 * (size_t)i + 3] - wp[3 * (size_t)i], dy = wp[3 * (size_t)i + 4] - wp[3 * (size_t)i + 1];
        if (dx * dx + dy * dy == 0.0) degenerate = true;
    }
    PlanGrid g;
    memset(&g, 0, sizeof(g));
    g.x0 = xl - margin; g.y0 = yl - margin; g.inv_cell = 1.0 / cell;
    const double gw = std::ceil((xh + margin - g.x0) / cell), gh = std::ceil((yh + margin - g.y0) / cell);
    if (!(gw >= 1 && gh >= 1) || gw * gh > 16.0e6) return fail(F110_E_INVALID, "f110_pure_pursuit_prepare: grid of %.0f x %.0f cells (choose a larger cell)", gw, gh);
    g.gw = (int)gw; g.gh = (int)gh; g.degenerate = degenerate ? 1 : 0;
    const size_t cells = (size_t)g.gw * g.gh;
    std::vector<uint8_t> count(cells, 0);
    std::vector<uint16_t> cand(cells * PG_CAP, 0);
    if (!degenerate) {
        // segments bucketed by a coarse grid first, so that a cell only looks at the segments that can matter
        const double hd = 0.5 * cell * std::sqrt(2.0);
        auto seg_dist = [&](int i, double px, double py) {
            const double x0 = wp[3 * (size_t)i], y0 = wp[3 * (size_t)i + 1];
            const double dx = wp[3 * (size_t)i + 3] - x0, dy = wp[3 * (size_t)i + 4] - y0;
            const double l2 = dx * dx + dy * dy;
            double t = ((px - x0) * dx + (py - y0) * dy) / l2;
            t = t < 0.0 ? 0.0 : t; t = t > 1.0 ? 1.0 : t;
            const double qx = px - (x0 + t * dx), qy = py - (y0 + t * dy);
            return std::sqrt(qx * qx + qy * qy);
        };
        std::vector<double> dist((size_t)nseg);
        for (int iy = 0; iy < g.gh; iy++)
            for (int ix = 0; ix < g.gw; ix++) {
                const double cx = g.x0 + (ix + 0.5) * cell, cy = g.y0 + (iy + 0.5) * cell;
                double D = 1e300;
                for (int i = 0; i < nseg; i++) { dist[(size_t)i] = seg_dist(i, cx, cy); D = std::min(D, dist[(size_t)i]); }
                const double lim = D + 2.0 * hd + 1e-6;
                unsigned n = 0;
                const size_t c = (size_t)iy * g.gw + ix;
                for (int i = 0; i < nseg && n <= (unsigned)PG_CAP; i++)
                    if (dist[(size_t)i] <= lim) { if (n < (unsigned)PG_CAP) cand[c * PG_CAP + n] = (uint16_t)i; n++; }
                count[c] = n > (unsigned)PG_CAP ? (uint8_t)PG_ALL : (uint8_t)n;
            }
    }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // an enqueued plan may still read the previous grid
    if (h->d_plan_count) { (void)hipFree(h->d_plan_count); h->d_plan_count = nullptr; }
    if (h->d_plan_cand) { (void)hipFree(h->d_plan_cand); h->d_plan_cand = nullptr; }
    HIP_TRY(hipMalloc((void **)&h->d_plan_count, cells));
    HIP_TRY(hipMalloc((void **)&h->d_plan_cand, cells * PG_CAP * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(h->d_plan_count, count.data(), cells, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->d_plan_cand, cand.data(), cells * PG_CAP * sizeof(uint16_t), hipMemcpyHostToDevice));
    g.count = h->d_plan_count; g.cand = h->d_plan_cand;
    h->plan_grid = g; h->plan_wp = waypoints; h->plan_M = M; h->plan_ok = true;
    return F110_OK;
}

extern "C" int f110_pure_pursuit(f110_handle *h, const double *waypoints, int32_t M, double lookahead, double vgain,
                                 double wheelbase, double max_reacquire, const double *state, int32_t n,
                                 double *actions, void *stream)
{
    // stateless: the handle is optional (NULL: the launch goes to the calling thread's current device)
    if (n < 0) return fail(F110_E_INVALID, "f110_pure_pursuit: bad arguments");
    if (n == 0) return F110_OK;
    if (h) if (int rc = check_device(h, "f110_pure_pursuit")) return rc;
    if (!waypoints || !state || !actions) return fail(F110_E_INVALID, "f110_pure_pursuit: null pointer");
    if (M < 2) return fail(F110_E_INVALID, "f110_pure_pursuit: M=%d waypoints (a raceline has at least 2)", M);
    if (h && h->plan_ok && h->plan_wp == waypoints && h->plan_M == M) {
        // a prepared raceline: one lane per car over the grid's candidate lists
        PlanArgs a;
        a.waypoints = waypoints; a.M = M; a.lookahead = lookahead; a.vgain = vgain; a.wheelbase = wheelbase;
        a.max_reacquire = max_reacquire; a.state = state; a.n = n; a.actions = actions;
        hipLaunchKernelGGL(pure_pursuit_grid_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, h->plan_grid);
        HIP_TRY(hipGetLastError());
        return F110_OK;
    }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    const size_t smem = pure_pursuit_lds_bytes(M);
    const DevLds &dl = device_lds(dev);
    if (dl.max_bytes <= 0) return fail(F110_E_HIP, "f110_pure_pursuit: cannot query the LDS size of device %d", dev);
    if (smem + 1024 > (size_t)dl.max_bytes) {
        // The raceline does not fit the LDS of this device (gfx950: 160 KiB, about 6 400 points): the global-memory
        // form, without a workspace for block boxes -- every block is evaluated.  f110_pure_pursuit_tracks with a
        // workspace is the fast way to plan on long or many racelines.
        PlanTracksArgs t;
        memset(&t, 0, sizeof(t));
        int32_t *off = single_track_offsets(dev, M);
        if (!off) return fail(F110_E_HIP, "f110_pure_pursuit: no device memory for the raceline header");
        t.t.waypoints = waypoints; t.t.offsets = off; t.t.K = 1; t.t.boxes = nullptr; t.track_of_car = nullptr;
        t.lookahead = lookahead; t.vgain = vgain; t.wheelbase = wheelbase; t.max_reacquire = max_reacquire;
        t.state = state; t.n = n; t.actions = actions;
        hipLaunchKernelGGL(pure_pursuit_tracks_kernel, dim3((n + PPG_WAVES - 1) / PPG_WAVES), dim3(PPG_WAVES * 64), 0, (hipStream_t)stream, t);
        HIP_TRY(hipGetLastError());
        return F110_OK;
    }
    PlanArgs a;
    a.waypoints = waypoints; a.M = M; a.lookahead = lookahead; a.vgain = vgain; a.wheelbase = wheelbase;
    a.max_reacquire = max_reacquire; a.state = state; a.n = n; a.actions = actions;
    if (smem > 64 * 1024 && smem > dl.pp_attr) { // raised once per device and size, not on every call (nor inside a captured policy)
        HIP_TRY(hipFuncSetAttribute((const void *)pure_pursuit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        device_lds(dev).pp_attr = smem;
    }
    hipLaunchKernelGGL(pure_pursuit_kernel, dim3((n + PP_WAVES - 1) / PP_WAVES), dim3(PP_WAVES * 64), smem, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int64_t f110_pure_pursuit_workspace(int32_t total_points, int32_t K)
{
    if (total_points < 0 || K < 0) return 0;
    return (((int64_t)total_points >> 6) + K) * 5;
}

extern "C" int f110_pure_pursuit_tracks(f110_handle *h, const double *waypoints, const int32_t *offsets_dev,
                                        const int32_t *offsets_host, int32_t K, const int32_t *track_of_car, double lookahead,
                                        double vgain, double wheelbase, double max_reacquire, const double *state, int32_t n,
                                        double *actions, double *workspace, int32_t boxes_valid, void *stream)
{
    if (n < 0 || K < 1) return fail(F110_E_INVALID, "f110_pure_pursuit_tracks: bad arguments (n=%d, K=%d)", n, K);
    if (h) if (int rc = check_device(h, "f110_pure_pursuit_tracks")) return rc;
    if (!waypoints || !offsets_dev || !offsets_host || !workspace) return fail(F110_E_INVALID, "f110_pure_pursuit_tracks: null pointer");
    int max_m = 0;
    if (offsets_host[0] != 0) return fail(F110_E_INVALID, "f110_pure_pursuit_tracks: offsets[0] must be 0");
    for (int k = 0; k < K; k++) {
        const int64_t m = (int64_t)offsets_host[k + 1] - offsets_host[k];
        if (m < 2 || m > 0x3fffffff) return fail(F110_E_INVALID, "f110_pure_pursuit_tracks: raceline %d has %lld points (at least 2)", k, (long long)m);
        max_m = std::max(max_m, (int)m);
    }
    TrackSet t;
    t.waypoints = waypoints; t.offsets = offsets_dev; t.K = K; t.boxes = workspace;
    if (!boxes_valid) {
        const int max_blocks = (max_m - 1 + 63) / 64;
        hipLaunchKernelGGL(track_boxes_kernel, dim3((max_blocks + 3) / 4, K), dim3(256), 0, (hipStream_t)stream, t);
        HIP_TRY(hipGetLastError());
    }
    if (n == 0) return F110_OK;
    if (!state || !actions) return fail(F110_E_INVALID, "f110_pure_pursuit_tracks: null pointer");
    PlanTracksArgs a;
    a.t = t; a.track_of_car = track_of_car; a.lookahead = lookahead; a.vgain = vgain; a.wheelbase = wheelbase;
    a.max_reacquire = max_reacquire; a.state = state; a.n = n; a.actions = actions;
    hipLaunchKernelGGL(pure_pursuit_tracks_kernel, dim3((n + PPG_WAVES - 1) / PPG_WAVES), dim3(PPG_WAVES * 64), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
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
    s.maps = h->d_maps; s.scan = scan_dev(h); s.n_cars = n; s.agents = 1;
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

// ---------------------------------------------------------------- scan -> bitmap
struct f110_bitmap {
    f110_bitmap_config cfg;
    int32_t *d_idx = nullptr;
    double *d_cos = nullptr, *d_sin = nullptr;
    int S = 0;
    size_t lds = 0;
};

extern "C" void f110_bitmap_destroy(f110_bitmap *b)
{
    if (!b) return;
    DeviceScope on_dev(b->cfg.device);
    if (b->d_idx) (void)hipFree(b->d_idx);
    if (b->d_cos) (void)hipFree(b->d_cos);
    if (b->d_sin) (void)hipFree(b->d_sin);
    delete b;
}

extern "C" int f110_bitmap_create(const f110_bitmap_config *cfg, const int32_t *indices, const double *cosines,
                                  const double *sines, f110_bitmap **out)
{
    if (!cfg || !indices || !cosines || !sines || !out) return fail(F110_E_INVALID, "f110_bitmap_create: null argument");
    const int T = cfg->target_beam_count;
    // the reference's assertions (lidar.py:50-56)
    if (!(T > 0 && T < cfg->num_beams)) return fail(F110_E_INVALID, "target_beam_count must satisfy 0 < %d < len(scan) = %d", T, cfg->num_beams);
    if (T > 2048) return fail(F110_E_INVALID, "target_beam_count %d > 2048", T);
    if (cfg->rows <= 0 || cfg->cols <= 0) return fail(F110_E_INVALID, "output_image_dims must be at least 1x1");
    if (cfg->rows > 4096 || cfg->cols > 4096) return fail(F110_E_INVALID, "output_image_dims above 4096 are not supported");
    if (cfg->channels != 1 && cfg->channels != 3 && cfg->channels != 4) return fail(F110_E_INVALID, "channels must 1, 3, or 4");
    if (cfg->draw_mode < F110_BITMAP_FILL || cfg->draw_mode > F110_BITMAP_RAYS) return fail(F110_E_INVALID, "draw_mode must be FILL, POLYGON or RAYS");
    for (int k = 0; k < T; k++)
        if (indices[k] < 0 || indices[k] >= cfg->num_beams) return fail(F110_E_INDEX, "beam index %d out of range", indices[k]);
    int S = (cfg->cols + 31) / 32;
    S |= 1; // odd row pitch: the per-row parity pass is LDS-bank-conflict free
    const size_t lds = bitmap_lds_bytes(T, cfg->rows, S);
    if (lds > 150 * 1024) return fail(F110_E_INVALID, "image %dx%d with %d beams needs %zu bytes of LDS (limit 150 KiB)", cfg->rows, cfg->cols, T, lds);
    f110_bitmap *b = new (std::nothrow) f110_bitmap;
    if (!b) return fail(F110_E_INVALID, "out of memory");
    b->cfg = *cfg; b->S = S; b->lds = lds;
    DeviceScope on_dev(cfg->device);
    if (on_dev.err != hipSuccess) { delete b; return fail(F110_E_HIP, "hipSetDevice(%d) failed", cfg->device); }
    hipError_t e = hipMalloc((void **)&b->d_idx, T * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_cos, T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_sin, T * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(b->d_idx, indices, T * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_cos, cosines, T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_sin, sines, T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && lds > 64 * 1024)
        e = hipFuncSetAttribute((const void *)bitmap_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { f110_bitmap_destroy(b); return fail(F110_E_HIP, "f110_bitmap_create: %s", hipGetErrorString(e)); }
    *out = b;
    return F110_OK;
}

extern "C" int f110_bitmap_render(f110_bitmap *b, const void *scans, int32_t scans_f64, int64_t n, int64_t stride,
                                  uint8_t *out, void *stream)
{
    if (!b || n < 0) return fail(F110_E_INVALID, "f110_bitmap_render: bad arguments");
    if (n == 0) return F110_OK;
    if (!scans || !out) return fail(F110_E_INVALID, "f110_bitmap_render: null pointer");
    if (stride < b->cfg.num_beams || n > 0x7fffffff) return fail(F110_E_INVALID, "f110_bitmap_render: stride %lld < num_beams or n too large", (long long)stride);
    if ((uintptr_t)out % 16) return fail(F110_E_INVALID, "f110_bitmap_render: out must be 16-byte aligned");
    if (int rc = check_current_device(b->cfg.device, "f110_bitmap_render")) return rc;
    BitmapArgs a;
    a.scans = scans; a.is_f64 = scans_f64 != 0; a.stride = stride; a.n = (int)n;
    a.idx = b->d_idx; a.cosv = b->d_cos; a.sinv = b->d_sin; a.T = b->cfg.target_beam_count;
    a.rows = b->cfg.rows; a.cols = b->cfg.cols; a.channels = b->cfg.channels; a.mode = b->cfg.draw_mode;
    a.bg = b->cfg.bg_value; a.draw = b->cfg.draw_value; a.draw_center = b->cfg.draw_center;
    a.scale = b->cfg.scaling_factor; a.out = out; a.S = b->S;
    hipLaunchKernelGGL(bitmap_kernel, dim3((unsigned)n), dim3(BM_THREADS), b->lds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_bitmap_points(f110_bitmap *b, const void *scans, int32_t scans_f64, int64_t n, int64_t stride,
                                  int32_t *points, void *stream)
{
    if (!b || n < 0) return fail(F110_E_INVALID, "f110_bitmap_points: bad arguments");
    if (n == 0) return F110_OK;
    if (!scans || !points) return fail(F110_E_INVALID, "f110_bitmap_points: null pointer");
    if (stride < b->cfg.num_beams || n > 0x7fffffff) return fail(F110_E_INVALID, "f110_bitmap_points: stride %lld < num_beams or n too large", (long long)stride);
    if (int rc = check_current_device(b->cfg.device, "f110_bitmap_points")) return rc;
    BitmapArgs a;
    memset(&a, 0, sizeof(a));
    a.scans = scans; a.is_f64 = scans_f64 != 0; a.stride = stride; a.n = (int)n;
    a.idx = b->d_idx; a.cosv = b->d_cos; a.sinv = b->d_sin; a.T = b->cfg.target_beam_count;
    a.rows = b->cfg.rows; a.cols = b->cfg.cols; a.scale = b->cfg.scaling_factor;
    const long long items = (long long)n * a.T;
    hipLaunchKernelGGL(bitmap_points_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, points);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}

extern "C" int f110_scan_occupancy(const void *scans, int32_t scans_f64, int64_t n, int64_t stride, int32_t num_beams,
                                   const double *cosines, const double *sines, double max_range, double lo, double hi,
                                   int32_t grid, uint8_t *out, void *stream)
{
    if (n < 0 || num_beams <= 0 || grid <= 0 || grid > 1024) return fail(F110_E_INVALID, "f110_scan_occupancy: bad arguments");
    if (n == 0) return F110_OK;
    if (!scans || !cosines || !sines || !out) return fail(F110_E_INVALID, "f110_scan_occupancy: null pointer");
    if (stride < num_beams || n > 0x7fffffff) return fail(F110_E_INVALID, "f110_scan_occupancy: stride < num_beams or n too large");
    if ((uintptr_t)out % 16) return fail(F110_E_INVALID, "f110_scan_occupancy: out must be 16-byte aligned");
    OccArgs a;
    a.scans = scans; a.is_f64 = scans_f64 != 0; a.stride = stride; a.n = (int)n; a.num_beams = num_beams;
    a.cosv = cosines; a.sinv = sines; a.max_range = max_range; a.lo = lo; a.hi = hi; a.grid = grid; a.out = out;
    const size_t lds = (size_t)((grid * grid + 31) / 32) * 4;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)occupancy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(occupancy_kernel, dim3((unsigned)n), dim3(BM_THREADS), lds, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return F110_OK;
}
