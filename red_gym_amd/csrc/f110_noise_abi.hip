// f110_noise_abi.hip -- part of the C ABI (include/f110_hip.h) over the gfx950 kernels; see f110_internal.h for the units.
#define F110_UNIT_NOISE
#include "f110_internal.h"

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

int noise_init(f110_handle *h)
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
