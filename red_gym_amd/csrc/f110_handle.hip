// f110_handle.hip -- part of the C ABI (include/f110_hip.h) over the gfx950 kernels; see f110_internal.h for the units.
#include "f110_internal.h"

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}



extern "C" const char *f110_last_error(void) { return g_err; }

// Launches go to the caller's stream, which belongs to the calling thread's CURRENT device: it must be the handle's.
// (Checked, not switched: hipGetDevice is a thread-local read; switching would cost two runtime calls per step and
// still leave the caller's stream on the wrong device.)
int check_current_device(int dev, const char *who)
{
    int cur = -1;
    HIP_TRY(hipGetDevice(&cur));
    if (cur != dev)
        return fail(F110_E_INVALID, "%s: the handle lives on device %d but the calling thread's current device is %d; make the "
                    "handle's device current (hipSetDevice / torch.cuda.device) and pass a stream of that device", who, dev, cur);
    return F110_OK;
}



int check_device(const f110_handle *h, const char *who) { return check_current_device(h->cfg.device, who); }

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

int upload(double **dst, const double *src, size_t n)
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
