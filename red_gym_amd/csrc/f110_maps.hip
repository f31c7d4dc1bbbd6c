// f110_maps.hip -- part of the C ABI (include/f110_hip.h) over the gfx950 kernels; see f110_internal.h for the units.
#define F110_UNIT_MAPS
#include "f110_internal.h"

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

// env -> map slot.  The SCAN_WAVES consecutive cars of a scan workgroup share the LDS copy of their map's LUT; where
// that holds for every workgroup (maps in blocks of envs whose car count is a multiple of SCAN_WAVES) the scan keeps that
// shape.  Otherwise -- a map per env, odd blocks -- it runs one wave per workgroup, each staging its own car's LUT:
// same results, fewer waves per CU (the LDS copies then cap a CU at 19 waves instead of 32).
extern "C" int f110_assign_maps(f110_handle *h, const int32_t *map_of_env)
{
    if (!h) return fail(F110_E_INVALID, "f110_assign_maps: null handle");
    const int B = h->cfg.num_envs, A = h->cfg.num_agents, N = B * A;
    std::vector<int32_t> m(B, 0);
    bool multi = false, single = false;
    if (map_of_env)
        for (int e = 0; e < B; e++) {
            const int k = map_of_env[e];
            if (k < 0 || k >= F110_MAX_MAPS || !h->slots[k].used)
                return fail(F110_E_INDEX, "f110_assign_maps: env %d uses map slot %d, which holds no map", e, k);
            m[e] = k;
            multi = multi || k != 0;
        }
    if (multi)
        for (int c = 0; c < N && !single; c += SCAN_WAVES)
            for (int j = 1; j < SCAN_WAVES && c + j < N; j++)
                if (m[(c + j) / A] != m[c / A]) { single = true; break; }
    ON_DEVICE(h->cfg.device);
    HIP_TRY(hipDeviceSynchronize()); // enqueued steps may still read the table
    if (!h->d_env_map) HIP_TRY(hipMalloc((void **)&h->d_env_map, sizeof(int32_t) * B));
    HIP_TRY(hipMemcpy(h->d_env_map, m.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice));
    h->h_env_map = m;
    h->multi = multi;
    h->wg_single = single;
    h->epoch++;
    return F110_OK;
}
