// f110_kernels.h -- the kernels of one batched env step on gfx950, in launch order:
//   dynamics_kernel   (lane per car)        RaceCar.update_pose minus the scan (+ reset)
//   scan_kernel       (wave per car)        ScanSimulator2D.scan + noise + iTTC
//   opp_setup_kernel  (lane per car pair)   \ RaceCar.ray_cast_agents, only when A > 1
//   opp_apply_kernel  (wave per car)        /
//   env_kernel        (lane per env)        GJK, collision flags, iTTC state update, lap timing, done, autoreset
// plus small function-level kernels used by the parity entry points.
#pragma once
#include "f110_device.h"
#include "f110_noise.h"

#pragma clang fp contract(off)

namespace f110 {

// clamp x to [lo, hi] in one instruction (the compiler only forms v_med3_i32 when it can
// prove lo <= hi, which it cannot for a runtime map size)
__device__ inline int med3_i32(int x, int lo, int hi)
{
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
    return r;
}

// wave-wide vote straight on the condition mask (HIP's __ballot round-trips through a VGPR)
__device__ inline unsigned long long vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr int WAVE = 64;
// bits of the device error word (f110_device_errors; include/f110_hip.h F110_DEVERR_*)
constexpr uint32_t DEVERR_NOISE_WINDOW = 1u, DEVERR_BOUNDS = 2u;
constexpr int TL_MAX_WAVES = 8; // (diagnostics builds: per-wave stamps of a workgroup)
#ifndef F110_SCAN_WAVES
#define F110_SCAN_WAVES 2
#endif
#ifndef F110_REFILL_MIN_IDLE
#define F110_REFILL_MIN_IDLE 40
#endif
constexpr int SCAN_WAVES = F110_SCAN_WAVES;   // cars per workgroup (one wavefront each)
constexpr int SCAN_THREADS = SCAN_WAVES * WAVE;
constexpr int REFILL_MIN_IDLE = F110_REFILL_MIN_IDLE; // refill the wave's beam slots once this many lanes idle

// Cell table.  Each map cell stores, as a u16, the BYTE OFFSET of its distance inside the LDS
// copy of the LUT: 8*k, k = RANK of the cell's exact squared distance d2 (in cells, to the
// nearest obstacle) among the distinct d2 values of the map, for k <= 1021 (squared
// distances are sums of two squares, so that reaches d2 ~ 3 900 = 62 cells); OFF_FAR for
// larger ranks and for cells of a user table that are not resolution*sqrt(int) -- those
// re-read a second table (u16 rank, 65535 = "use the fp64 table") in a rarely taken branch.
// Rank 0 is always d2 = 0, so LDS offset 0 is the distance 0.0.
// The table has a one-cell BORDER on every side holding OFF_BORDER, whose LDS slot holds
// dt[-1,-1]: the reference's out-of-bounds read (laser_models.py:80-81,:103) becomes an
// ordinary lookup of a clamped index -- no bounds compare, no select, no index clamp or
// scaling in the march loop (the loaded value addresses the ds_read directly).
// Layout: 8-column strips, map cell (r, c), r in -1..H, c in -1..W, at [(c >> 3) + 1][r + 1][c & 7] (arithmetic
// shift: the left border column is the last column of strip 0), so one 128-B cache line holds an 8x8-cell block.  The 64 rays of a wave sample neighbouring
// points, so a gather touches fewer lines than with a row-major table (which measured
// ~40 L1 accesses per 64-lane gather and made the kernel L1-tag-rate bound), and the byte
// offset is two shift-adds and one multiply-add: (c >> 3) * (strip_bytes - 16) + (c << 1) + (r << 4) + strip_bytes + 16.
constexpr int LUT_LDS = 1024;                           // LDS LUT slots
constexpr unsigned SLOT_FAR = LUT_LDS - 2, SLOT_BORDER = LUT_LDS - 1;
constexpr unsigned OFF_FAR = 8 * SLOT_FAR, OFF_BORDER = 8 * SLOT_BORDER;
constexpr unsigned CODE_ESC = 65535;                    // second table: read the fp64 table instead

struct MapDev {
    const uint16_t *cells;  // padded strips [(W >> 3) + 2][Hp][8] of LDS byte offsets
    const uint16_t *cells_far; // same layout: rank (<= 65534) of the cells marked OFF_FAR, 65535 = fp64 table
    unsigned cells_bytes;
    unsigned strip_bytes;   // Hp * 16, Hp = H + 2 rounded up to a multiple of 8
    const double *lut;      // [<=65534] resolution*sqrt(d2_k), indexed by rank k
    const double *lut_lds;  // [LUT_LDS] image staged in LDS: lut[0..SLOT_FAR-1], unused, dt[-1,-1]
    const double *dt;       // [H*W] exact fp64 distance table (escape path, rarely touched)
    int H, W;
    double res, rinv, ox, oy, oc, os, wres, hres, oob; // oob = dt[H-1][W-1]
};

// device-side view of MapDev with the cell table behind a buffer resource descriptor
struct MapView {
    __amdgpu_buffer_rsrc_t cells_rsrc;
    unsigned strip_bytes, row_bias, strip_m16; // row_bias = strip_bytes + 16, strip_m16 = strip_bytes - 16
    const MapDev *desc;  // rare paths (far cells, escape cells) re-read their table pointers from the descriptor:
                         // three 64-bit pointers less to keep in scalar registers across the march loop
    int H, W;
    double res, rinv, ox, oy, oc, os, wres, hres;
    double nox, noy; // -ox * rinv, -oy * rinv (exact when rinv is a power of two)
    __device__ void init(const MapDev &m)
    {
        strip_bytes = m.strip_bytes; row_bias = m.strip_bytes + 16u; strip_m16 = m.strip_bytes - 16u; desc = &m; H = m.H; W = m.W; res = m.res; rinv = m.rinv;
        ox = m.ox; oy = m.oy; oc = m.oc; os = m.os; wres = m.wres; hres = m.hres;
        nox = -m.ox * m.rinv; noy = -m.oy * m.rinv;
    }
};

struct ScanDev {
    int nb, theta_dis;
    double fov, eps, max_range, inc; // inc = theta_index_increment (laser_models.py:368)
    unsigned long long inc_fx;       // inc in 24.40 fixed point
    int cs_len;                      // entries of cs (theta_dis * repetitions)
    const double2 *cs;               // [cs_len] {cos, sin} of the LUT angles (laser_models.py:379-381), repeated
};

// laser_models.py:56-104: (x, y) -> distance-table value, branch-free.  IDENT: origin
// yaw == 0 (c=1, s=0: the rotation is the identity in exact arithmetic).  POW2:
// resolution is a power of two, so q = x_rot * (1/res) IS the reference's quotient and
// "x_rot < 0 or x_rot >= width*res" (:79) is exactly "floor(q) outside [0, W)", which the
// clamp to [-1, W] maps onto the table's border.
template <bool IDENT, bool POW2>
__device__ inline double dist_lookup(const MapView &m, const double *lds_lut, double x, double y, bool live)
{
    double xr = 0, yr = 0, qx, qy;
    if (IDENT && POW2) {
        // q = (x - ox) * 2^k is ONE fma: scaling by a power of two commutes with rounding, so
        // fma(x, 2^k, -ox*2^k) == fl(x - ox) * 2^k bit for bit (the reference's two roundings
        // collapse because the second is exact).  Explicit fma: contraction stays off.
        qx = __builtin_fma(x, m.rinv, m.nox);
        qy = __builtin_fma(y, m.rinv, m.noy);
    } else {
        const double xt = x - m.ox, yt = y - m.oy;
        if (IDENT) { xr = xt; yr = yt; }
        else { xr = xt * m.oc + yt * m.os; yr = -xt * m.os + yt * m.oc; }
        qx = xr * m.rinv;
        qy = yr * m.rinv;
    }
    const double fx = floor(qx), fy = floor(qy);
    int ci = (int)fx, ri = (int)fy; // saturating conversion; the clamp below finishes the job
    if (!POW2) {
        // int(x_rot/resolution) and the bounds test need the IEEE quotient: x_rot*(1/res) is
        // within ~2e-12 of it, so only quotients within 1e-9 of an integer (where truncation
        // or a bound could flip) replay the reference's own expressions.
        const double rx = qx - fx, ry = qy - fy;
        const bool near_int = (rx < 1e-9) || (rx > 1. - 1e-9) || (ry < 1e-9) || (ry > 1. - 1e-9);
        if (__builtin_expect(vote(near_int) != 0ull, 0)) {
            if (near_int) {
                const bool out = (xr < 0) || (xr >= m.wres) || (yr < 0) || (yr >= m.hres);
                ci = out ? -1 : min((int)(xr / m.res), m.W - 1);
                ri = out ? -1 : min((int)(yr / m.res), m.H - 1);
            }
        }
    }
    const int cc = med3_i32(ci, -1, m.W);      // column -1..W (both ends are border cells)
    const int rr = med3_i32(ri, -1, m.H);      // row -1..H
    // Byte offset of cell (rr, cc): strip (cc >> 3) + 1 (arithmetic shift: column -1 is the last column of
    // strip 0), 16 bytes per row inside a strip.  The +1 strip and the +1 border row ride in the constant
    // of the shift-add (`row_bias` = strip_bytes + 16, a multiple of 16), so no add is spent on the padding
    // and the offset never goes negative; asm so that the constant is not re-associated into a trailing add.
    // (c >> 3) * S + (c & 7) * 2 == (c >> 3) * (S - 16) + c * 2: no masking of the column bits needed
    unsigned row16;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(row16) : "v"(rr), "s"(m.row_bias));
    unsigned rc;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(rc) : "v"(cc), "v"(row16));
    unsigned off = (unsigned)(__mul24(cc >> 3, (int)m.strip_m16) + (int)rc);
    // a finished ray presents an out-of-range offset: the hardware range check answers 0
    // (= LDS offset 0 = distance 0.0, which parks the ray: total += 0, x += 0*c) without
    // occupying the L1 tag pipeline
    off = live ? off : 0xffffffffu;
    // buffer load: 32-bit per-lane offset against a scalar descriptor
    const unsigned code = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(m.cells_rsrc, (int)off, 0, 0);
    // common case: the loaded value IS the LDS byte offset of the distance: one ds_read_b64
    double d = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lds_lut) + code);
    // pin the LDS read: otherwise the compiler folds it and the rare global reads below
    // into one flat_load through a selected generic pointer
    asm volatile("" : "+v"(d));
    const bool far = code == OFF_FAR;
    if (__builtin_expect(vote(far) != 0ull, 0)) {
        if (far) {
            const MapDev *dp = m.desc;
            asm volatile("" : "+s"(dp)); // opaque: the loads below stay here instead of being hoisted to the kernel entry
            const unsigned rank = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(dp->cells_far) + (size_t)off);
            d = (rank != CODE_ESC) ? dp->lut[rank] : dp->dt[(size_t)(unsigned)rr * (unsigned)m.W + (unsigned)cc];
        }
    }
    return d;
}

// np.fmod(t, td) (laser_models.py:170) without the generic library loop: for |t/td| < 2^31
// the result t - trunc(t/td)*td is exact (fmod results are representable and q*td is an
// exact product); the rounded quotient can only be one too large in magnitude, which
// shows as a remainder of the wrong sign and is redone with the corrected quotient.
__device__ inline double fmod_small(double t, double td)
{
    const double qf = trunc(t / td);
    if (!(fabs(qf) < 2147483648.0)) return fmod(t, td);
    double r = t - qf * td;
    if ((t >= 0 && r < 0) || (t < 0 && r > 0)) r = t - (qf - (t >= 0 ? 1.0 : -1.0)) * td;
    return r;
}

// laser_models.py:167-184: LUT index of beam b.  The reference advances theta_index by
// num_beams sequential fp64 adds (wrapping at theta_dis); in 24.40 fixed point
// T0 + b*INC differs from that recurrence by < 1e-9, so the integer part agrees unless
// the fraction is within 1e-7 of 0 or 1 -- then this lane replays the recurrence exactly.
// The index is NOT wrapped: the {cos, sin} table is stored `cs_reps` times back to back
// (wrapping subtracts theta_dis exactly, so entry idx and idx - theta_dis are the same).
__device__ inline int beam_theta_index(unsigned long long T0, double t0w, int b, const ScanDev &s)
{
    const unsigned inc_lo = (unsigned)s.inc_fx, inc_hi = (unsigned)(s.inc_fx >> 32);
    unsigned long long t = (unsigned long long)(unsigned)b * inc_lo + T0;    // v_mad_u64_u32
    unsigned hi = (unsigned)(t >> 32) + (unsigned)b * inc_hi;                // inc_hi, b < 2^24
    const unsigned lo = (unsigned)t;
    int idx = (int)(hi >> 8);
    const unsigned frac = __builtin_amdgcn_alignbit(hi, lo, 8);             // top 32 fraction bits
    if (__builtin_expect(frac + 430u < 860u || idx >= s.cs_len, 0)) {        // within 1e-7 of an integer
        const double td = (double)s.theta_dis;
        double tt = t0w;
        for (int j = 0; j < b; j++) {
            tt += s.inc;
            while (tt >= td) tt -= td;
        }
        idx = (int)tt;
    }
    return idx;
}

struct ScanArgs {
    const MapDev *maps;         // dev [K] map descriptors
    const int32_t *env_map;     // dev [B] map of every env, or NULL (all envs on maps[0]); the cars of one
                                // workgroup share a map (f110_assign_maps checks it): its LUT is staged per group
    ScanDev scan;
    int n_cars;             // cars of THIS launch: car_base .. car_base + n_cars - 1
    int car_base;           // first car (a shard whose env blocks sit on maps of different kinds -- resolution a power
                            // of two or not, origin rotated or not -- is scanned block by block, each with its own instantiation)
    int agents;             // A (cars of one env are consecutive)
    int wpc;                // wavefronts per car (power of two): small batches split a car's beams over
                            // several waves so that the chip is still filled; chunk position p goes to wave p % wpc
    // Wave -> (car, part) mapping: consecutive STAGES of cars, stage s giving each of its stage_cars[s] cars
    // 2^stage_log2w[s] waves (launch_scan explains the choice).  Read through `rare`, not held in registers.
    int n_stages;
    int stage_cars[8];          // SCAN_MAX_STAGES
    int stage_log2w[8];         // each 0..SCAN_MAX_LOG2W
    // pose source: pose = (src[car*stride], src[car*stride+1], src[car*stride+yaw_off])
    const double *pose_src;
    int pose_stride, yaw_off;
    // full-step extras (all NULL for the function-level scan)
    const double *state;         // [N,7]: velocity for the iTTC test
    const int32_t *noise_step;   // [N]
    const NoiseDesc *noise;      // device descriptor of the noise table ({noise of a row, side distance} pairs: one 16-B gather
                                 // per beam taken); read once per car, its address is fixed for the handle's life
    const int32_t *env_noise;    // [B] noise slot (= seed) of every env, or NULL (all envs on slot 0)
    uint32_t *dev_err;           // device error word (f110_device_errors): F110_DEVERR_* bits, or NULL
    const double *beam_cosines;  // [nb]
    double ttc_thresh;
    uint8_t *in_collision;       // [N]
    const uint8_t *pending_reset;// [B]
    int reset_only;              // 1: only envs with pending_reset are processed
    const uint16_t *chunk_beam0; // [ceil(nb/64)] first beam of the k-th 64-beam chunk to be marched (long rays first)
    // outputs
    float *out_f32;              // [N,nb] or NULL
    double *out_f64;             // [N,nb] or NULL
    uint32_t *lookups;           // [N] or NULL (accumulated)
    unsigned long long *timeline; // diagnostics (builds with -DF110_TIMELINE only, tools/timeline.py): per wave
                                  // {start, rays started, end} in 100 MHz ticks and (car << 8 | part); else NULL
};

// One wavefront per car.  Lanes own rays; a finished ray idles (its lookups return 0.0) until at least REFILL_MIN_IDLE lanes are idle, then every idle
// lane (a) finishes its previous beam -- noise, iTTC candidate test, fp32/fp64 store --
// and (b) takes the next beam of the car.  No LDS staging of the scan: the only LDS
// use is the 8 KiB distance LUT shared by the workgroup, so occupancy is register-bound.
// STEP: full env step (noise + iTTC + state update); false: ScanSimulator2D.scan(pose, None).
constexpr int MAX_CHUNKS = 64; // beams are handed out in chunks of 64 (num_beams <= 4096)

// scan_kernel re-reads its argument block through the kernarg segment pointer, which is only the same block
// if ScanArgs is the kernel's ONLY argument, passed by value at offset 0, and trivially copyable (the launch
// memcpy's it).  The stage list is a fixed array inside it: launch_scan checks the count and the exponents.
constexpr int SCAN_MAX_STAGES = 8, SCAN_MAX_LOG2W = 3;
static_assert(__is_trivially_copyable(ScanArgs), "ScanArgs is copied into the kernarg segment byte for byte");
static_assert(offsetof(ScanArgs, maps) == 0, "kernarg re-read assumes the argument block starts with ScanArgs");
static_assert(sizeof(((ScanArgs *)0)->stage_cars) == SCAN_MAX_STAGES * sizeof(int) &&
              sizeof(((ScanArgs *)0)->stage_log2w) == SCAN_MAX_STAGES * sizeof(int), "stage list capacity");
static_assert(sizeof(ScanArgs) <= 4096, "kernarg segment size");

// SM 0: ScanSimulator2D.scan(pose, None); 1: the scan of a step (noise, iTTC flag; env_kernel follows).
template <bool IDENT, bool POW2, int SM>
#ifndef F110_SCAN_MIN_WAVES
#define F110_SCAN_MIN_WAVES 8
#endif
__global__ __launch_bounds__(SCAN_THREADS, F110_SCAN_MIN_WAVES) void scan_kernel(ScanArgs a)
{
    constexpr bool STEP = SM >= 1;
    __shared__ __attribute__((aligned(16))) double s_lut[LUT_LDS];
    __shared__ int s_chunk0[MAX_CHUNKS];
    // the same argument block addressed through the kernarg segment (ScanArgs is the only kernel argument): rarely
    // needed fields are re-read through it where they are used instead of being held in SGPRs for the whole kernel
#if defined(__HIP_DEVICE_COMPILE__)
    const ScanArgs *rare = (const ScanArgs *)__builtin_amdgcn_kernarg_segment_ptr();
#else
    const ScanArgs *rare = &a; // host pass of the single-source compile: never executed
#endif
#if defined(F110_TIMELINE)
    __shared__ volatile unsigned long long s_tl[TL_MAX_WAVES][2]; // stamps wait in LDS, not in registers, for the end of the wave
#endif
    const int nb = a.scan.nb;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int wid = blockIdx.x * SCAN_WAVES + wave; // wave-uniform (scalar)
    // wave -> (car, part of its beam queue).  Kept to one extra argument and shifts: this kernel sits at
    // the 80-SGPR budget of 8 waves/SIMD, and a scalar spilled inside the refill loop costs ~3 % of the launch.
    int wpc, car, part;
    {
        int t = wid, c = 0, st = 0;
        const int ns = rare->n_stages;
        for (; st < ns; st++) {
            const int cars_s = rare->stage_cars[st], w = cars_s << rare->stage_log2w[st];
            if (t < w) break;
            t -= w; c += cars_s;
        }
        const int lg = st < ns ? rare->stage_log2w[st] : 0;
        wpc = 1 << lg; car = st < ns ? c + (t >> lg) : a.n_cars; part = t & (wpc - 1);
    }
    // the car's map (wave-uniform: scalar loads); waves past the last car still help to stage the LUT
    const int car_c = rare->car_base + min(car, a.n_cars - 1);
    const MapDev &md = a.maps[a.env_map ? a.env_map[car_c / a.agents] : 0];
    {   // LDS image of the LUT prepared by the host (slot SLOT_BORDER = dt[-1,-1]): 16-B copies
        const double2 *src = reinterpret_cast<const double2 *>(md.lut_lds);
        double2 *dst = reinterpret_cast<double2 *>(s_lut);
        for (int i = threadIdx.x; i < LUT_LDS / 2; i += SCAN_THREADS) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < ((nb + 63) >> 6); i += SCAN_THREADS) s_chunk0[i] = a.chunk_beam0[i];
#if defined(F110_TIMELINE)
    { unsigned long long t = wall_clock64(); asm volatile("" : "+v"(t)); if (lane == 0) s_tl[wave][0] = t; }
#endif
    __syncthreads();
    MapView mv;
    mv.cells_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(md.cells), 0, (int)md.cells_bytes, 0x00020000);
    mv.init(md);
    if (car >= a.n_cars) return;
    car += rare->car_base; // (from here on the car's index in the shard)
    // this wave's slice of the car's beam queue: chunk positions part, part+wpc, ...
    const int nch = (nb + 63) >> 6;
    const int my_chunks = nch > part ? (nch - part + wpc - 1) / wpc : 0;
    const int owns_last = my_chunks > 0 && ((nch - 1) % wpc) == part;
    const int nbl = my_chunks * 64 - (owns_last ? nch * 64 - nb : 0); // beams of this wave
    if (a.reset_only && !a.pending_reset[car / a.agents]) return;

    const double px = a.pose_src[(size_t)car * a.pose_stride];
    const double py = a.pose_src[(size_t)car * a.pose_stride + 1];
    const double yaw = a.pose_src[(size_t)car * a.pose_stride + a.yaw_off];
    const double eps = a.scan.eps, max_range = a.scan.max_range;

    // per-car constants of the finishing stage
    const double vel = STEP ? a.state[(size_t)car * 7 + 3] : 0.0;
    const bool do_ttc = STEP && vel != 0.0;               // laser_models.py:206
    // iTTC hit needs 0 <= (v - side)/(vel*cos) < thresh, hence |v - side| < thresh*|vel|:
    // only such candidate beams pay the exact fp64 division
    const double cand = a.ttc_thresh * fabs(vel) * 1.000000001;
    const double2 *__restrict__ ns = nullptr;
    if (STEP) {
        // the car's noise row: row `scans since its reset` of its env's slot (a ring of nd.cap rows per slot)
        const NoiseDesc nd = *rare->noise;
        const long long row = (long long)a.noise_step[car];
        const int32_t *en = rare->env_noise;
        const long long slot = en ? (long long)en[car / a.agents] : 0ll;
        if (__builtin_expect(row < nd.lo || row >= nd.hi, 0)) {
            // the host keeps the table ahead of every car (Engine._ensure_noise); a row outside it is reported, never silent
            if (lane == 0 && rare->dev_err) atomicOr(rare->dev_err, DEVERR_NOISE_WINDOW);
        }
        ns = nd.base + (size_t)(slot * nd.cap + (row & nd.mask)) * (size_t)nb;
    }
    float *o32 = a.out_f32 ? a.out_f32 + (size_t)car * nb : nullptr;
    double *o64 = a.out_f64 ? a.out_f64 + (size_t)car * nb : nullptr;
    bool hit = false;

    // finishing stage of one beam: clamp (laser_models.py:143-144), noise (:450-452),
    // stores, iTTC (:189-217).  nzv / sdv: noise and side distance of the beam, loaded by
    // the caller ahead of time.
    auto emit = [&](int i, double tot, double nzv, double sdv) {
        double v = __builtin_fmin(tot, max_range); // :143-144 (a NaN total, i.e. a NaN pose, also clamps)
        if (STEP) v += nzv;
        if (o32) *reinterpret_cast<float *>(reinterpret_cast<char *>(o32) + (size_t)((unsigned)i * 4u)) = (float)v;
        if (o64) *reinterpret_cast<double *>(reinterpret_cast<char *>(o64) + (size_t)((unsigned)i * 8u)) = v;
        if (do_ttc) {
            const double sd = v - sdv;
            if (__builtin_expect(fabs(sd) < cand, 0)) {
                const ScanArgs *ra = rare;
                asm volatile("" : "+s"(ra)); // re-read the rarely needed arguments here instead of holding them in SGPRs
                const double proj_vel = vel * ra->beam_cosines[i];
                const double ttc = sd / proj_vel;
                if ((ttc < ra->ttc_thresh) && (ttc >= 0.0)) hit = true;
            }
        }
    };

    // ---- ray march (laser_models.py:107-186) -------------------------------------
    // The first table read of every beam is at the car itself (:129): done once.
    const double d0 = dist_lookup<IDENT, POW2>(mv, s_lut, px, py, true);
#if defined(F110_TIMELINE)
    { unsigned long long t = wall_clock64(); asm volatile("" : "+v"(t)); if (lane == 0) s_tl[wave][1] = t; }
#endif
    unsigned nlook = (unsigned)nbl; // the reference reads the table once per beam before marching
    if (!(d0 > eps && d0 <= max_range)) {
        for (int k = lane; k < nbl; k += WAVE) {
            const int i = s_chunk0[(k >> 6) * wpc + part] + (k & 63);
            const double2 v = STEP ? ns[i] : make_double2(0.0, 0.0);
            emit(i, d0, v.x, v.y);
        }
    } else {
        const double td = (double)a.scan.theta_dis;
        double t0w = td * (yaw - a.scan.fov / 2.) / (2. * F110_PI);
        t0w = fmod_small(t0w, td);
        while (t0w < 0) t0w += td;
        // 24.40 fixed point of t0w in [0, theta_dis); a NaN / out-of-range yaw falls to the slow path
        const unsigned long long T0 = (t0w >= 0 && t0w < td) ? (unsigned long long)(t0w * 1099511627776.0) : ~0ull;

        int next = 0;           // wave-uniform: next unassigned slot of the beam order
        bool active = false;
        int beam = -1;          // beam whose result `total` holds (-1: none)
        double x = px, y = py, c = 0, s = 0, total = 0;
        double nz = 0, sd = 0;  // noise and side distance of the lane's beam (fetched when the beam is taken)
        for (;;) {
            // ---- refill phase: idle lanes finish their beam and take the next one ----
            const unsigned long long idle = vote(!active);
            const int nidle = __popcll(idle);
            if (!active) {
                // all independent loads first (one memory round trip).  The noise / side-distance entry is fetched
                // for the beam being TAKEN and carried in registers until the beam is finished: idle lanes take
                // consecutive beams, so this gather touches a few cache lines, where a gather by the FINISHED beams
                // (scattered over the scan) touched a line per lane -- the L1's tag pipeline is what bounds this kernel
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                const int k = next + rank;
                const bool take = k < nbl;
                const int kk = take ? k : 0;
                const int b = s_chunk0[(kk >> 6) * wpc + part] + (kk & 63);
                const double2 nsv = STEP ? *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(ns) + (size_t)((unsigned)b * 16u))
                                         : make_double2(0.0, 0.0);
                const double nzv = nz, sdv = sd;
                const int ti = beam_theta_index(T0, t0w, b, a.scan);
                const double2 cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.scan.cs) + (size_t)((unsigned)ti * 16u)); // second round trip, overlapped with emit()
                if (beam >= 0) emit(beam, total, nzv, sdv);
                beam = -1;
                if (take) {
                    c = cs.x;
                    s = cs.y;
                    x = px + d0 * c;
                    y = py + d0 * s;
                    total = d0;
                    beam = b;
                    nz = nsv.x;
                    sd = nsv.y;
                    active = true;
                }
            }
            next += nidle;
            int nact = __popcll(vote(active));
            if (nact == 0) break;
            // ---- march phase: every lane steps (idle lanes are parked by d = 0) until
            // enough lanes are idle again or, once no beams are left, the wave has drained ----
            const int go = next < nbl ? WAVE - REFILL_MIN_IDLE : 0; // keep marching while nact > go
#if defined(F110_DRAIN_PRIO)
            if (go == 0) __builtin_amdgcn_s_setprio(F110_DRAIN_PRIO); // experiment: a draining wave ends on its longest ray's dependent chain
#endif
            do {
                nlook += (unsigned)nact;
                const double d = dist_lookup<IDENT, POW2>(mv, s_lut, x, y, active);
                total += d;
                x += d * c;
                y += d * s;
                const bool c1 = d > eps, c2 = total <= max_range;
                active = c1 && c2;
                nact = __popcll(vote(c1) & vote(c2)); // two direct compare masks: no bool round trip
            } while (nact > go);
        }
    }
    const ScanArgs *ra = rare;
    asm volatile("" : "+s"(ra));
    if (ra->lookups && lane == 0) atomicAdd(&ra->lookups[car], nlook);
#if defined(F110_TIMELINE)
    unsigned long long tl_end = wall_clock64();
    asm volatile("" : "+v"(tl_end));
    if (ra->timeline && lane == 0) {
        unsigned long long *tl = ra->timeline + (size_t)wid * 4;
        tl[0] = s_tl[wave][0]; tl[1] = s_tl[wave][1]; tl[2] = tl_end; tl[3] = ((unsigned long long)car << 8) | (unsigned)part | ((unsigned long long)wpc << 40);
    }
#endif

    // ---- iTTC result: the flag only; env_kernel zeroes the state (base_classes.py:241-250)
    // once every wave of the car is done.  Plain store: all writers store the same 1.
    if (STEP) {
        if (vote(hit) != 0ull && lane == 0) ra->in_collision[car] = 1;
    }
}

// ------------------------------------------------------------------ opponents (A > 1)
// RaceCar.ray_cast_agents (base_classes.py:204-225): the car's CURRENT pose (yaw already
// zeroed by an iTTC hit, :245) against the other cars' post-integration snapshot poses.
// Two kernels: a set-up with all the fp64 trigonometry (one lane per (car, opponent) pair:
// opponent corners, blocked beam span, beam-direction rotation) and a lean apply kernel
// (one lane per (car, beam)) that tests the four edges for the beams inside a span.  A
// single wave-per-car kernel with the trig inlined needed 162 VGPRs and was latency-bound
// at 2 waves/SIMD (0.165 ms at 32 768 cars); kept out of scan_kernel in any case so the
// march loop stays at 45 VGPRs.
struct OppPair {
    double px, py, cA, sA; // ego position, cos/sin(yaw + pi/2)
    double v[8];           // opponent corners rl, rr, fr, fl
    double qx, qy, reach;  // opponent centre relative to the ego, padded half diagonal: a ray whose line passes
                           // the centre at more than `reach` cannot cross an edge (conservative pre-test, see opp_apply)
    int lo, hi;            // get_blocked_view_indices span (lo > hi: nothing to do)
    unsigned long long chunks; // bit c: the 64-beam chunk c of [lo, hi] holds beams whose LINE passes the opponent's bounding
                               // circle (a chunk-granular form of opp_apply's per-beam pre-test: the other chunks are skipped)
};

struct OppArgs {
    int n_cars, agents, nb;
    const double *state;      // [N,7]
    const double *pose_snap;  // [N,3]
    const uint8_t *in_collision; // [N]
    const double *scan_angles;
    const double2 *beam_cs;   // [nb] {cos, sin}(scan_angles)
    const Params *params;     // [slots, 1 + agents] (see DynArgs): a car sizes its opponents with its OWN params (base_classes.py:221)
    const int32_t *env_params;// [B] or NULL
    const uint8_t *pending_reset;
    int reset_only;
    OppPair *pairs;           // [N, agents-1] scratch owned by the handle
    float *scans32;           // [N,nb] or NULL
    double *scans64;          // [N,nb] or NULL
};

__global__ __launch_bounds__(128) void opp_setup_kernel(OppArgs a)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = a.agents - 1;
    if (p >= a.n_cars * per) return;
    const int car = p / per, jj = p % per;
    const int env = car / a.agents, a0 = env * a.agents, self = car - a0;
    OppPair &o = a.pairs[p];
    if (a.reset_only && !a.pending_reset[env]) { o.lo = 1; o.hi = 0; o.chunks = 0ull; return; }
    const int j = jj < self ? jj : jj + 1; // opponents in agent order, skipping the car itself (:574)
    const double *st = a.state + (size_t)car * 7;
    // an iTTC hit zeroes the yaw before the ray cast (base_classes.py:245); env_kernel applies it
    const double px = st[0], py = st[1], pyaw = a.in_collision[car] ? 0.0 : st[4];
    const double *op = a.pose_snap + (size_t)(a0 + j) * 3;
    const Params &P = a.params[(size_t)(a.env_params ? a.env_params[env] : 0) * (a.agents + 1) + 1 + self];
    double verts[4][2];
    get_vertices(op[0], op[1], op[2], P.v[P_LENGTH], P.v[P_WIDTH], verts);
    // laser_models.py:283-315
    const double ex = cos(pyaw), ey = sin(pyaw);
    const double ego_ang = atan2(ey, ex);
    int lo = 0, hi = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const double vx = verts[i][0] - px, vy = verts[i][1] - py;
        const double norm = sqrt(vx * vx + vy * vy);
        const double ux = vx / norm, uy = vy / norm;
        double angle = ego_ang - atan2(uy, ux);
        if (angle > F110_PI) angle = angle - 2 * F110_PI;
        else if (angle < -F110_PI) angle = angle + 2 * F110_PI;
        const int ind = argmin_abs_diff_sorted(a.scan_angles, a.nb, -angle);
        if (i == 0) { lo = hi = ind; }
        else { lo = ind < lo ? ind : lo; hi = ind > hi ? ind : hi; }
    }
    if (lo > a.nb - 1 || hi > a.nb - 1) { lo = 1; hi = 0; } // only reachable with NaN inputs
    const double A = pyaw + F110_PI / 2.;
    o.px = px; o.py = py; o.cA = cos(A); o.sA = sin(A);
#pragma unroll
    for (int k = 0; k < 4; k++) { o.v[2 * k] = verts[k][0]; o.v[2 * k + 1] = verts[k][1]; }
    {
        const double cx = 0.5 * (verts[0][0] + verts[2][0]), cy = 0.5 * (verts[0][1] + verts[2][1]);
        const double dx = verts[0][0] - verts[2][0], dy = verts[0][1] - verts[2][1];
        o.qx = cx - px; o.qy = cy - py;
        o.reach = 0.5 * sqrt(dx * dx + dy * dy) * 1.000001 + 1e-9;
        if (!(o.reach == o.reach)) o.reach = __builtin_inf(); // NaN poses: no pre-test
    }
    o.lo = lo; o.hi = hi;
    // Which chunks can matter.  A beam can only be cut short if its line passes within `reach` of the opponent's
    // centre (opp_apply's pre-test), i.e. if its direction, modulo pi, is within asin(reach / |q|) of the direction to
    // the centre.  With the opponent straight behind the car the reference's span is the whole scan (the corner
    // angles straddle +-pi, laser_models.py:293-315) although only a few dozen beams point at it or away from it:
    // opp_apply then walks 2..4 chunks instead of 17.  Conservative (margins of a beam increment and 1e-6 rad), so the
    // beams tested inside the visited chunks -- and the results -- are those of the per-beam pre-test alone.
    unsigned long long mask = 0ull;
    if (lo <= hi) {
        const double qn = sqrt(o.qx * o.qx + o.qy * o.qy);
        const double sa0 = a.scan_angles[0];
        const double incr = (a.scan_angles[a.nb - 1] - sa0) / (double)(a.nb - 1);
        const bool all = !(qn > o.reach * 1.000001) || !(o.reach < __builtin_inf()) || !(incr > 0.0);
        if (all) {
            for (int c = lo >> 6; c <= (hi >> 6); c++) mask |= 1ull << c;
        } else {
            const double span = asin(o.reach / qn) + 2.0 * incr + 1e-6;
            const double phi = remainder(atan2(o.qy, o.qx) - pyaw, 2.0 * F110_PI); // direction to the centre in the scan's frame
            const double inv = 1.0 / incr;
            // the direction, its opposite (lines, not rays) and their images one turn away: beam-index intervals
            for (int k = -2; k <= 2; k++) {
                const double centre = phi + (double)k * F110_PI;
                const double a0 = (centre - span - sa0) * inv, a1 = (centre + span - sa0) * inv;
                if (!(a1 >= (double)lo) || !(a0 <= (double)hi)) continue;
                const int i0 = a0 > (double)lo ? (int)floor(a0) : lo, i1 = a1 < (double)hi ? (int)ceil(a1) : hi;
                if (i0 > i1) continue;
                const int c0 = i0 >> 6, c1 = i1 >> 6;
                mask |= (~0ull >> (63 - c1)) & ~((1ull << c0) - 1ull);
            }
        }
    }
    o.chunks = mask;
}

// One wave per car, walking the 64-beam chunks that some opponent's span touches (most cars see their
// opponent in a few dozen beams; a wave per (car, chunk) spent its time being launched: 17 waves per car,
// 16 of them leaving at once).
__global__ __launch_bounds__(256) void opp_apply_kernel(OppArgs a)
{
    const int car = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    if (car >= a.n_cars) return;
    const int lane = threadIdx.x & 63, per = a.agents - 1;
    const OppPair *pairs = a.pairs + (size_t)car * per; // wave-uniform: scalar loads
    unsigned long long todo = 0ull; // chunks some opponent's span touches AND whose beams can reach it (opp_setup)
    for (int jj = 0; jj < per; jj++)
        if (pairs[jj].lo <= pairs[jj].hi) todo |= pairs[jj].chunks;
    while (todo) {
        const int chunk = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const int base = chunk << 6;
        const int i = base + lane;
        double best = __builtin_inf();
        for (int jj = 0; jj < per; jj++) {
            const OppPair &o = pairs[jj];
            if (!((o.chunks >> chunk) & 1ull)) continue; // uniform: nothing of this opponent in this chunk
            if (i < o.lo || i > o.hi || i >= a.nb) continue;
            const double2 cs = a.beam_cs[i];
            const double v3x = o.cA * cs.x - o.sA * cs.y, v3y = o.sA * cs.x + o.cA * cs.y;
            // v3 is the ray's unit normal (laser_models.py:262): |q . v3| is the distance of the opponent's centre
            // from the ray's line.  Beyond the padded half diagonal no edge can be crossed (every get_range would
            // return inf), which is the case for ~95 % of the beams when the span is the whole scan.
            if (!(fabs(o.qx * v3x + o.qy * v3y) <= o.reach)) continue;
            // The ray's direction is (v3y, -v3x).  If the whole bounding circle lies BEHIND the car along it, every edge
            // point has a negative ray parameter: get_range rejects it (d1 >= 0, :271) -- unless an edge is exactly
            // parallel to the ray (denom == 0: the collinear branch answers whatever the direction, :275-280).
            if (o.qx * v3y - o.qy * v3x < -o.reach) {
                bool parallel = false;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int en = (e + 1) & 3;
                    const double denom = (o.v[2 * en] - o.v[2 * e]) * v3x + (o.v[2 * en + 1] - o.v[2 * e + 1]) * v3y;
                    parallel = parallel || !(fabs(denom) > 0.0);
                }
                if (!parallel) continue;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int en = (e + 1) & 3;
                const double r = get_range(o.px, o.py, v3x, v3y, o.v[2 * e], o.v[2 * e + 1], o.v[2 * en], o.v[2 * en + 1]);
                if (r < best) best = r;
            }
        }
        if (best < __builtin_inf()) {
            if (a.scans64) { double *s = a.scans64 + (size_t)car * a.nb + i; if (best < *s) *s = best; }
            if (a.scans32) { float *s = a.scans32 + (size_t)car * a.nb + i; const float b32 = (float)best; if (b32 < *s) *s = b32; }
        }
    }
}

// ------------------------------------------------------------------ dynamics (lane per car)
struct DynArgs {
    int n_cars, agents;
    double *state;        // [N,7]
    double *steer_buf;    // [N,2]
    int32_t *steer_cnt;   // [N]
    int32_t *noise_step;  // [N] or NULL
    const double *actions;// [N,2] (steer, speed)
    const double *spawn;  // [N,3] or NULL
    const uint8_t *pending_reset; // [B] or NULL
    int reset_only;
    double *pose_snap;    // [N,3] or NULL
    uint8_t *in_collision;// [N] or NULL: cleared here, set by scan_kernel
    // Vehicle parameters: [slots, 1 + agents] -- per params slot (= the `params` one reference env was constructed with,
    // f110_env.py:125-128) entry 0 is Simulator.params (GJK vertices, base_classes.py:542), entry 1 + i RaceCar.params of
    // agent i (:84,169, changed by update_params :507-527)
    const Params *params;
    const int32_t *env_params;  // [B] params slot of every env, or NULL (all envs on slot 0)
    double time_step;
    int integrator;
};

// The single-track model switches to its kinematic form below 0.5 m/s (dynamic_models.py:152): a wavefront that holds
// one slow car among 63 fast ones executes BOTH forms at every RK4 stage (the slow form is a third of the instructions
// of a step, and with autoreset a few per cent of the cars are always just leaving their spawn pose -- enough to put a
// slow car into most wavefronts).  The block therefore deals its cars out so that the slow ones (and the idle lanes)
// share the LAST wavefronts: lane l works on car s_perm[l], the others' waves skip the kinematic code altogether.
// Which lane integrates a car does not change a bit of its result.
__global__ __launch_bounds__(256) void dynamics_kernel(DynArgs a)
{
    __shared__ int s_perm[256];
    __shared__ int s_cnt[2][4]; // per wave: fast cars, slow cars
    int car;
    {
        const int c0 = blockIdx.x * blockDim.x + threadIdx.x;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        bool work = c0 < a.n_cars, slow = true;
        if (work) {
            const bool pend0 = a.pending_reset && a.pending_reset[c0 / a.agents];
            if (a.reset_only && !pend0) work = false;
            else slow = pend0 || !(fabs(a.state[(size_t)c0 * 7 + 3]) >= 0.5); // a reset car starts at rest
        }
        const unsigned long long mf = __builtin_amdgcn_ballot_w64(work && !slow), ms = __builtin_amdgcn_ballot_w64(work && slow);
        if (lane == 0) { s_cnt[0][wave] = __popcll(mf); s_cnt[1][wave] = __popcll(ms); }
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s_perm[i] = -1;
        __syncthreads();
        int fast_before = 0, slow_before = 0, fast_total = 0, slow_total = 0;
        for (int w = 0; w < 4; w++) {
            if (w < wave) { fast_before += s_cnt[0][w]; slow_before += s_cnt[1][w]; }
            fast_total += s_cnt[0][w]; slow_total += s_cnt[1][w];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        // fast cars fill the block's lanes from the front, slow cars from the back (idle lanes in between)
        if (work && !slow) s_perm[fast_before + __popcll(mf & below)] = c0;
        if (work && slow) s_perm[255 - (slow_before + __popcll(ms & below))] = c0;
        __syncthreads();
        car = s_perm[threadIdx.x];
        (void)fast_total; (void)slow_total;
    }
    if (car < 0) return;
    const int env = car / a.agents;
    const bool pend = a.pending_reset && a.pending_reset[env];
    double st[7], sb[2];
    int sc;
    double steer, speed;
    if (pend) {
        // RaceCar.reset (base_classes.py:181-202) followed by the zero-action step of
        // F110Env.reset (f110_env.py:335-336)
#pragma unroll
        for (int i = 0; i < 7; i++) st[i] = 0.;
        st[0] = a.spawn[(size_t)car * 3];
        st[1] = a.spawn[(size_t)car * 3 + 1];
        st[4] = a.spawn[(size_t)car * 3 + 2];
        sb[0] = sb[1] = 0.;
        sc = 0;
        steer = 0.;
        speed = 0.;
        if (a.noise_step) a.noise_step[car] = 0;
    } else {
#pragma unroll
        for (int i = 0; i < 7; i++) st[i] = a.state[(size_t)car * 7 + i];
        sb[0] = a.steer_buf[(size_t)car * 2];
        sb[1] = a.steer_buf[(size_t)car * 2 + 1];
        sc = a.steer_cnt[car];
        steer = a.actions[(size_t)car * 2];
        speed = a.actions[(size_t)car * 2 + 1];
    }
    const Params P = a.params[(size_t)(a.env_params ? a.env_params[env] : 0) * (a.agents + 1) + 1 + car % a.agents];
    update_pose(st, sb, sc, steer, speed, P, a.time_step, a.integrator);
#pragma unroll
    for (int i = 0; i < 7; i++) a.state[(size_t)car * 7 + i] = st[i];
    a.steer_buf[(size_t)car * 2] = sb[0];
    a.steer_buf[(size_t)car * 2 + 1] = sb[1];
    a.steer_cnt[car] = sc;
    if (a.in_collision) a.in_collision[car] = 0;
    if (a.pose_snap) {
        a.pose_snap[(size_t)car * 3] = st[0];
        a.pose_snap[(size_t)car * 3 + 1] = st[1];
        a.pose_snap[(size_t)car * 3 + 2] = st[4];
    }
}

// ------------------------------------------------------------------ env bookkeeping (lane per env)
struct EnvArgs {
    int n_envs, agents, ego_idx, autoreset, reset_only;
    double *state;            // [N,7]: state[3:] zeroed here on an iTTC hit
    int32_t *noise_step;      // [N]: one noise row consumed per scan
    const double *pose_snap;  // [N,3]
    const double *spawn;      // [N,3]
    const uint8_t *in_collision; // [N]
    uint8_t *collisions;      // [N]
    int32_t *collision_idx;   // [N]
    double *start_rot;        // [B,4]
    uint8_t *near_start;      // [N]
    int32_t *toggles;         // [N]
    int32_t *lap_counts;      // [N]
    double *lap_times;        // [N]
    double *current_time;     // [B]
    uint8_t *pending_reset;   // [B]
    uint8_t *done;            // [B]
    uint8_t *checkpoint_done; // [N] or NULL
    const Params *params;     // [slots, 1 + agents] (see DynArgs): entry 0 of the env's slot sizes the GJK quads
    const int32_t *env_params;// [B] or NULL
    double time_step;
};

// collision_models.py:185-212 on A <= 8 quads held in registers/scratch
__device__ inline void collision_multiple_dev(const double *poses /*[A,3]*/, int A, double L, double W,
                                              uint8_t *col, int32_t *cidx)
{
    for (int i = 0; i < A; i++) { col[i] = 0; cidx[i] = -1; }
    for (int i = 0; i < A - 1; i++) {
        double vi[4][2];
        get_vertices(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2], L, W, vi);
        for (int j = i + 1; j < A; j++) {
            double vj[4][2];
            get_vertices(poses[3 * j], poses[3 * j + 1], poses[3 * j + 2], L, W, vj);
            if (gjk_collision(vi, vj)) {
                col[i] = 1; col[j] = 1;
                cidx[i] = j; cidx[j] = i;
            }
        }
    }
}

// F110Env._check_done (f110_env.py:202-244) for the A cars of one env: every car's offset from its OWN start
// position, rotated by the EGO's start rotation (:219-221, :329), folded onto the 2 m wide start strip (:223-229),
// `closes = dist2 <= 0.1` (:231), toggle on every change of near_start (:232-239), lap_counts = toggles // 2 (:238),
// lap_times follows current_time while toggles < 4 (:239-240).  Returns all(toggles >= 4).
// xy: car i's position at xy[i*stride], xy[i*stride+1]; start: [A,3] (x, y, theta).
__device__ inline bool check_done_dev(const double *xy, int stride, const double *start, int A, double r00, double r01,
                                      double r10, double r11, double current_time, uint8_t *near_start, int32_t *toggles,
                                      int32_t *lap_counts, double *lap_times, uint8_t *checkpoint_done)
{
    const double left_t = 2, right_t = 2;
    bool all_done = true;
    for (int i = 0; i < A; i++) {
        const double px = xy[(size_t)i * stride] - start[(size_t)i * 3];
        const double py = xy[(size_t)i * stride + 1] - start[(size_t)i * 3 + 1];
        const double dx = r00 * px + r01 * py;
        double temp_y = r10 * px + r11 * py;
        if (temp_y > left_t) temp_y -= left_t;
        else if (temp_y < -right_t) temp_y = -right_t - temp_y;
        else temp_y = 0;
        const double dist2 = dx * dx + temp_y * temp_y;
        const bool closes = dist2 <= 0.1;
        bool ns = near_start[i] != 0;
        int tg = toggles[i];
        if (closes && !ns) { ns = true; tg += 1; }
        else if (!closes && ns) { ns = false; tg += 1; }
        near_start[i] = ns ? 1 : 0;
        toggles[i] = tg;
        lap_counts[i] = tg / 2;
        if (tg < 4) lap_times[i] = current_time;
        if (checkpoint_done) checkpoint_done[i] = tg >= 4 ? 1 : 0;
        if (!(tg >= 4)) all_done = false;
    }
    return all_done;
}

// function-level _check_done: lane per env (f110_check_done)
struct CheckDoneArgs {
    int n_envs, agents, ego_idx;
    const double *poses;        // [n,A,3]
    const double *start;        // [n,A,3]
    const double *start_rot;    // [n,4] row-major 2x2
    const double *current_time; // [n]
    const uint8_t *collisions;  // [n,A]
    uint8_t *near_start;        // [n,A] in/out
    int32_t *toggles;           // [n,A] in/out
    int32_t *lap_counts;        // [n,A]
    double *lap_times;          // [n,A] in/out
    uint8_t *done;              // [n]
    uint8_t *checkpoint_done;   // [n,A] or NULL
};

__global__ __launch_bounds__(128) void check_done_kernel(CheckDoneArgs a)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n_envs) return;
    const int A = a.agents, c0 = env * A;
    const double *R = a.start_rot + (size_t)env * 4;
    const bool all_done = check_done_dev(a.poses + (size_t)c0 * 3, 3, a.start + (size_t)c0 * 3, A, R[0], R[1], R[2], R[3],
                                         a.current_time[env], a.near_start + c0, a.toggles + c0, a.lap_counts + c0,
                                         a.lap_times + c0, a.checkpoint_done ? a.checkpoint_done + c0 : nullptr);
    a.done[env] = ((a.collisions[c0 + a.ego_idx] != 0) || all_done) ? 1 : 0; // :242
}

__global__ __launch_bounds__(128) void env_kernel(EnvArgs a)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n_envs) return;
    const bool pend = a.pending_reset[env] != 0;
    if (a.reset_only && !pend) return;
    const int A = a.agents, c0 = env * A;
    // Simulator.check_collision (base_classes.py:529-543) on the post-integration poses
    const Params &SP = a.params[(size_t)(a.env_params ? a.env_params[env] : 0) * (A + 1)]; // Simulator.params (:542)
    collision_multiple_dev(a.pose_snap + (size_t)c0 * 3, A, SP.v[P_LENGTH], SP.v[P_WIDTH],
                           a.collisions + c0, a.collision_idx + c0);
    for (int i = 0; i < A; i++) {
        if (a.in_collision[c0 + i]) {
            a.collisions[c0 + i] = 1; // :581-582
            double *st = a.state + (size_t)(c0 + i) * 7; // check_ttc, base_classes.py:244-247
            st[3] = 0.; st[4] = 0.; st[5] = 0.; st[6] = 0.;
        }
        a.noise_step[c0 + i] += 1;
    }
    double ct = a.current_time[env];
    double r00, r01, r10, r11;
    if (pend) {
        // F110Env.reset (f110_env.py:318-329)
        ct = 0.0;
        const double th = -a.spawn[(size_t)(c0 + a.ego_idx) * 3 + 2];
        r00 = cos(th); r01 = -sin(th); r10 = sin(th); r11 = cos(th);
        a.start_rot[(size_t)env * 4] = r00; a.start_rot[(size_t)env * 4 + 1] = r01;
        a.start_rot[(size_t)env * 4 + 2] = r10; a.start_rot[(size_t)env * 4 + 3] = r11;
        for (int i = 0; i < A; i++) { a.near_start[c0 + i] = 1; a.toggles[c0 + i] = 0; }
        a.pending_reset[env] = 0;
    } else {
        r00 = a.start_rot[(size_t)env * 4]; r01 = a.start_rot[(size_t)env * 4 + 1];
        r10 = a.start_rot[(size_t)env * 4 + 2]; r11 = a.start_rot[(size_t)env * 4 + 3];
    }
    ct = ct + a.time_step; // f110_env.py:293
    a.current_time[env] = ct;
    const bool all_done = check_done_dev(a.state + (size_t)c0 * 7, 7, a.spawn + (size_t)c0 * 3, A, r00, r01, r10, r11, ct,
                                         a.near_start + c0, a.toggles + c0, a.lap_counts + c0, a.lap_times + c0,
                                         a.checkpoint_done ? a.checkpoint_done + c0 : nullptr);
    const bool dn = (a.collisions[c0 + a.ego_idx] != 0) || all_done;
    a.done[env] = dn ? 1 : 0;
    if (a.autoreset && dn) a.pending_reset[env] = 1;
}

// ------------------------------------------------------------------ function-level kernels
// dynamic_models.py:91-121 / :124-176 right-hand sides (the reference's KAT surface)
__global__ void rhs_kernel(const double *x, const double *u, int n, int kinematic, const Params *params, double *f)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Params P = params[1]; // slot 0, agent 0
    double xs[7], fs[7];
    for (int k = 0; k < 7; k++) xs[k] = x[(size_t)i * 7 + k];
    if (kinematic) {
        const double *p = P.v;
        const double lwb = p[P_LF] + p[P_LR];
        const double u0 = steering_constraint(xs[2], u[2 * i], p[P_SMIN], p[P_SMAX], p[P_SVMIN], p[P_SVMAX]);
        const double u1 = accl_constraints(xs[3], u[2 * i + 1], p[P_VSWITCH], p[P_AMAX], p[P_VMIN], p[P_VMAX]);
        fs[0] = xs[3] * cos(xs[4]); fs[1] = xs[3] * sin(xs[4]); fs[2] = u0; fs[3] = u1; fs[4] = xs[3] / lwb * tan(xs[2]);
        fs[5] = 0; fs[6] = 0;
    } else {
        vehicle_dynamics_st(xs, u[2 * i], u[2 * i + 1], P, fs);
    }
    for (int k = 0; k < 7; k++) f[(size_t)i * 7 + k] = fs[k];
}

__global__ void vertices_kernel(const double *poses, int n, double L, double W, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[4][2];
    get_vertices(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2], L, W, v);
    for (int k = 0; k < 4; k++) { out[(size_t)i * 8 + 2 * k] = v[k][0]; out[(size_t)i * 8 + 2 * k + 1] = v[k][1]; }
}

__global__ void gjk_pairs_kernel(const double *va, const double *vb, int n, uint8_t *hit)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[4][2], b[4][2];
    for (int k = 0; k < 4; k++) {
        a[k][0] = va[(size_t)i * 8 + 2 * k]; a[k][1] = va[(size_t)i * 8 + 2 * k + 1];
        b[k][0] = vb[(size_t)i * 8 + 2 * k]; b[k][1] = vb[(size_t)i * 8 + 2 * k + 1];
    }
    hit[i] = gjk_collision(a, b) ? 1 : 0;
}

__global__ void collision_multiple_kernel(const double *verts, int n, int A, uint8_t *col, int32_t *cidx)
{
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const double *v = verts + (size_t)g * A * 8;
    uint8_t *c = col + (size_t)g * A;
    int32_t *x = cidx + (size_t)g * A;
    for (int i = 0; i < A; i++) { c[i] = 0; x[i] = -1; }
    for (int i = 0; i < A - 1; i++) {
        double vi[4][2];
        for (int k = 0; k < 4; k++) { vi[k][0] = v[i * 8 + 2 * k]; vi[k][1] = v[i * 8 + 2 * k + 1]; }
        for (int j = i + 1; j < A; j++) {
            double vj[4][2];
            for (int k = 0; k < 4; k++) { vj[k][0] = v[j * 8 + 2 * k]; vj[k][1] = v[j * 8 + 2 * k + 1]; }
            if (gjk_collision(vi, vj)) { c[i] = 1; c[j] = 1; x[i] = j; x[j] = i; }
        }
    }
}

// check_ttc_jit (laser_models.py:189-217): wave per scan
__global__ void ttc_kernel(const double *scans, const double *vel, int n, int nb, const double *beam_cosines,
                           const double *side_distances, double thresh, uint8_t *hit)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const double v = vel[row];
    bool h = false;
    if (v != 0.0) {
        for (int i = lane; i < nb; i += WAVE) {
            const double proj_vel = v * beam_cosines[i];
            const double ttc = (scans[(size_t)row * nb + i] - side_distances[i]) / proj_vel;
            if ((ttc < thresh) && (ttc >= 0.0)) h = true;
        }
    }
    const bool any = __ballot(h) != 0ull;
    if (lane == 0) hit[row] = any ? 1 : 0;
}

// ray_cast (laser_models.py:319-346): wave per (ego, opponent quad)
__global__ void ray_cast_kernel(const double *ego, const double *verts, int n, int nb, const double *scan_angles,
                                const double2 *beam_cs,
                                double *scans, int32_t *span)
{
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    double v[4][2];
    for (int k = 0; k < 4; k++) { v[k][0] = verts[(size_t)row * 8 + 2 * k]; v[k][1] = verts[(size_t)row * 8 + 2 * k + 1]; }
    ray_cast_wave(ego[3 * row], ego[3 * row + 1], ego[3 * row + 2], v, scan_angles, beam_cs, nb, lane,
                  scans + (size_t)row * nb, nullptr, span ? span + 2 * row : nullptr);
}

} // namespace f110
