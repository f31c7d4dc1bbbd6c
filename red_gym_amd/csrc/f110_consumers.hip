// f110_consumers.hip -- part of the C ABI (include/f110_hip.h) over the gfx950 kernels; see f110_internal.h for the units.
#define F110_UNIT_CONSUMERS
#include "f110_internal.h"

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

// ---------------------------------------------------------------- scan -> bitmap
static const void *bitmap_fn(size_t lds, int mode, int channels)
{
    if (bm_fetch_ahead(mode, channels)) return (const void *)&bitmap_kernel<6, true>; // (its LDS leaves room for three workgroups per CU at most)
    return bm_waves_per_eu(lds) == 8 ? (const void *)&bitmap_kernel<8, false> : (const void *)&bitmap_kernel<6, false>;
}

struct f110_bitmap {
    f110_bitmap_config cfg;
    int32_t *d_idx = nullptr;
    double *d_cos = nullptr, *d_sin = nullptr;
    int S = 0;
    size_t lds[2] = {0, 0};  // dynamic LDS of a launch on fp32 / fp64 scans (the ranges' staging buffers differ)
    int resident[2] = {0, 0}; // workgroups of bitmap_kernel the device runs at once (the launch's grid: a workgroup loops over images)
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
    if (cfg->num_beams > 65536) return fail(F110_E_INVALID, "scans of more than 65536 beams are not supported (%d)", cfg->num_beams);
    if (cfg->rows <= 0 || cfg->cols <= 0) return fail(F110_E_INVALID, "output_image_dims must be at least 1x1");
    if (cfg->rows > 4096 || cfg->cols > 4096) return fail(F110_E_INVALID, "output_image_dims above 4096 are not supported");
    if (cfg->channels != 1 && cfg->channels != 3 && cfg->channels != 4) return fail(F110_E_INVALID, "channels must 1, 3, or 4");
    if (cfg->draw_mode < F110_BITMAP_FILL || cfg->draw_mode > F110_BITMAP_RAYS) return fail(F110_E_INVALID, "draw_mode must be FILL, POLYGON or RAYS");
    for (int k = 0; k < T; k++)
        if (indices[k] < 0 || indices[k] >= cfg->num_beams) return fail(F110_E_INDEX, "beam index %d out of range", indices[k]);
    int S = (cfg->cols + 31) / 32;
    S |= 1; // odd row pitch: the per-row parity pass is LDS-bank-conflict free
    const size_t lds = bitmap_lds_bytes(T, cfg->rows, S, cfg->draw_mode, cfg->channels, 1); // (fp64 scans: the larger of the two layouts)
    if (lds > 150 * 1024) return fail(F110_E_INVALID, "image %dx%d with %d beams needs %zu bytes of LDS (limit 150 KiB)", cfg->rows, cfg->cols, T, lds);
    f110_bitmap *b = new (std::nothrow) f110_bitmap;
    if (!b) return fail(F110_E_INVALID, "out of memory");
    b->cfg = *cfg; b->S = S; b->lds[1] = lds; b->lds[0] = bitmap_lds_bytes(T, cfg->rows, S, cfg->draw_mode, cfg->channels, 0);
    DeviceScope on_dev(cfg->device);
    if (on_dev.err != hipSuccess) { delete b; return fail(F110_E_HIP, "hipSetDevice(%d) failed", cfg->device); }
    hipError_t e = hipMalloc((void **)&b->d_idx, T * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_cos, T * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&b->d_sin, T * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(b->d_idx, indices, T * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_cos, cosines, T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(b->d_sin, sines, T * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess && lds > 64 * 1024)
        e = hipFuncSetAttribute(bitmap_fn(lds, cfg->draw_mode, cfg->channels), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) {
        int dev = 0;
        hipDeviceProp_t prop;
        e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
        for (int f = 0; f < 2 && e == hipSuccess; f++) {
            int per_cu = 0;
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bitmap_fn(b->lds[f], cfg->draw_mode, cfg->channels), BM_THREADS, b->lds[f]);
            b->resident[f] = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
        }
    }
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
    a.scale = b->cfg.scaling_factor; a.out = out; a.S = b->S; a.qcap = bm_queue_cap(a.T, a.mode); a.tl = nullptr;
#if defined(F110_BM_TIMELINE)
    // diagnostics build only: every launch is followed by a synchronisation and a table of the stage times on stderr
    static unsigned long long *tl = nullptr; static size_t tl_n = 0;
    if (tl_n < (size_t)n) { if (tl) (void)hipFree(tl); HIP_TRY(hipMalloc((void **)&tl, (size_t)n * BM_TL * 8)); tl_n = (size_t)n; }
    a.tl = tl;
#endif
    const char *grid_env = getenv("F110_BM_GRID"); // sweeps and tests only: workgroups of the launch (read per call)
    // fetch-ahead shape: as many workgroups as the device runs at once, each looping over images; else one per image
    const int64_t grid = !bm_fetch_ahead(a.mode, a.channels) ? n : std::min<int64_t>(n, grid_env && atoi(grid_env) > 0 ? atoi(grid_env) : b->resident[a.is_f64]);
    void *params[1] = {(void *)&a};
    HIP_TRY(hipLaunchKernel(bitmap_fn(b->lds[a.is_f64], a.mode, a.channels), dim3((unsigned)grid), dim3(BM_THREADS), params, b->lds[a.is_f64], (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
#if defined(F110_BM_TIMELINE)
    {
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        std::vector<unsigned long long> h((size_t)n * BM_TL);
        HIP_TRY(hipMemcpy(h.data(), tl, h.size() * 8, hipMemcpyDeviceToHost));
        static const char *names[] = {"zero", "direct pass", "records", "prefix", "walk", "parity", "next points", "store issue"};
        double sum[8] = {0}; unsigned long long t0 = ~0ull, t1 = 0;
        for (int64_t i = 0; i < n; i++) {
            const unsigned long long *s = &h[(size_t)i * BM_TL];
            unsigned long long prev = s[0];
            static const int order[8] = {1, 2, 3, 4, 5, 6, 8, 7};
            for (int j = 0; j < 8; j++) { const int k = order[j]; if (s[k]) { sum[j] += (double)(s[k] - prev); prev = s[k]; } }
            t0 = std::min(t0, s[0]); t1 = std::max(t1, s[7]);
        }
        fprintf(stderr, "bitmap timeline (%lld images, 100 MHz ticks -> us): launch %.1f us;", (long long)n, (t1 - t0) / 100.0);
        for (int j = 0; j < 8; j++) fprintf(stderr, " %s %.2f", names[j], sum[j] / n / 100.0);
        // between a workgroup's images (last stamp of one -> first stamp of the next) and a workgroup's life (first stamp of
        // its first image -> last stamp of its last one); workgroup g draws bm_image_of(g, round, grid, n)
        double gap = 0, life = 0, life_min = 1e30, life_max = 0, first = 0, first_max = 0; long long gaps = 0;
        const bool ahead = bm_fetch_ahead(a.mode, a.channels);
        for (int64_t g = 0; g < grid; g++) {
            int64_t prev = g, last = g;
            for (int it = 1; ahead; it++) {
                const int64_t i = bm_image_of((int)g, it, (int)grid, (int)n);
                if (i >= n) break;
                gap += (double)(h[(size_t)i * BM_TL] - h[(size_t)prev * BM_TL + 7]); gaps++;
                prev = last = i;
            }
            const double l = (double)(h[(size_t)last * BM_TL + 7] - h[(size_t)g * BM_TL]), f = (double)(h[(size_t)g * BM_TL] - t0);
            life += l; life_min = std::min(life_min, l); life_max = std::max(life_max, l); first += f; first_max = std::max(first_max, f);
        }
        if (gaps) fprintf(stderr, " | between images %.2f (grid %lld)", gap / gaps / 100.0, (long long)grid);
        if (const char *dump = getenv("F110_BM_TL_DUMP")) { // raw stamps for offline analysis
            if (FILE *f = fopen(dump, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        }
        const double gn = (double)grid;
        fprintf(stderr, " | workgroup life mean %.1f min %.1f max %.1f, first stamp after launch start mean %.1f max %.1f", life / gn / 100.0, life_min / 100.0, life_max / 100.0, first / gn / 100.0, first_max / 100.0);
        fprintf(stderr, "\n");
    }
#endif
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
