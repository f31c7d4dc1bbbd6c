// f110_kernels.h -- the kernels of one batched env step on gfx950, in launch order:
//   dynamics_kernel   (lane per car)        RaceCar.update_pose minus the scan (+ reset)
//   scan_kernel       (wave per car)        ScanSimulator2D.scan + noise + iTTC
//   opp_setup_kernel  (4 lanes per car pair) \ RaceCar.ray_cast_agents, only when A > 1
//   opp_apply_kernel  (wave per car)         /
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

// Bounds-checked debug build (-DF110_BOUNDS; `tools/build_variant.sh bounds -DF110_BOUNDS`, selected with F110_LIB; SURVEY 5
// "race detection / sanitizers": GPU AddressSanitizer is not available on this pool, the CPU oracle runs under ASan / UBSan).
// Every index a kernel of the step path forms from DATA -- a cell code, a rank, a slot number, a beam number, a noise row --
// is checked against its table before use; a violation ORs DEVERR_BOUNDS and the table's bit (8 + BT_*) into the handle's
// device error word (f110_device_errors) and the access is redirected to a valid element, so the run goes on and the
// report names the table.  The whole -m gpu suite is run against this build once per round (profiles/r04_bounds_build.txt).
enum { BT_LUT_CODE, BT_CELLS_FAR, BT_LUT_RANK, BT_DT, BT_NOISE_BEAM, BT_CS_TABLE, BT_CHUNK_ORDER, BT_MAP_SLOT, BT_NOISE_SLOT,
       BT_PARAMS_SLOT, BT_SCAN_STORE, BT_OPP_BEAM, BT_STAGE_LIST, BT_SELFTEST };
#if defined(F110_BOUNDS)
#define F110_BCHK(ok, table, errp) \
    do { if (!(ok)) { uint32_t *e_ = (errp); if (e_) atomicOr(e_, DEVERR_BOUNDS | (1u << (8 + (table)))); } } while (0)
#define F110_BOUNDS_ONLY(...) __VA_ARGS__
#else
#define F110_BCHK(ok, table, errp) do { } while (0)
#define F110_BOUNDS_ONLY(...)
#endif
constexpr int TL_MAX_WAVES = 8; // (diagnostics builds: per-wave stamps of a workgroup)
#ifndef F110_SCAN_WAVES
#define F110_SCAN_WAVES 2
#endif
#ifndef F110_REFILL_MIN_IDLE
#define F110_REFILL_MIN_IDLE 44
#endif
constexpr int SCAN_WAVES = F110_SCAN_WAVES;   // cars per workgroup (one wavefront each)
constexpr int SCAN_THREADS = SCAN_WAVES * WAVE;
constexpr int REFILL_MIN_IDLE = F110_REFILL_MIN_IDLE; // refill the wave's beam slots once this many lanes idle

// Cell table.  Each map cell stores, as a u16, the BYTE OFFSET of its distance inside the LDS copy of the LUT:
// 8 * (k + 1), k = RANK of the cell's exact squared distance d2 (in cells, to the nearest obstacle) among the distinct d2
// values of the map, for k < LDS_RANKS = 1022 (squared distances are sums of two squares, so that reaches d2 ~ 3 900 =
// 62 cells); OFF_FAR for larger ranks and for cells of a user table that are not resolution*sqrt(int) -- those re-read a
// second table (u16 rank, 65535 = "use the fp64 table") in a rarely taken branch.
// The table has a one-cell BORDER on every side holding code 0, and LDS slot 0 holds dt[-1,-1]: the reference's
// out-of-bounds read (laser_models.py:80-81,:103) becomes an ordinary lookup of a clamped index -- no bounds compare, no
// select in the march loop (the loaded value addresses the ds_read directly).
// The far marker's LDS slot holds -0.0: as a distance it is an exact no-op (total += -0.0, x += -0.0 * c) that ends the ray's
// march (-0.0 > eps is false); the wave looks at the sign of its parked lanes' last distance once per refill, not once per look-up.
// Layout: 8-column strips, map cell (r, c), r in -1..H, c in -1..W, at [(c >> 3) + 1][r + 1][c & 7] (arithmetic
// shift: the left border column is the last column of strip 0), so one 128-B cache line holds an 8x8-cell block.  The 64 rays of a wave sample neighbouring
// points, so a gather touches fewer lines than with a row-major table (which measured
// ~40 L1 accesses per 64-lane gather and made the kernel L1-tag-rate bound), and the byte
// offset is two shift-adds and one multiply-add: (c >> 3) * (strip_bytes - 16) + (c << 1) + (r << 4) + strip_bytes + 16.
// (Round 5 tried to have the address unit form an equivalent layout -- a swizzled structured descriptor, index = row, offset =
// 2 * column, the descriptor's range check as the row clamp: two VALU instructions instead of six -- and it is exact and 26 %
// slower: an `idxen` load merges at most two lanes per access, profiles/r05_swizzle_probe.txt.)
#ifndef F110_LUT_LDS
#define F110_LUT_LDS 1024
#endif
constexpr int LUT_LDS = F110_LUT_LDS;                   // LDS LUT slots
constexpr unsigned SLOT_OOB = 0, SLOT_FAR = LUT_LDS - 1; // slot 0: dt[-1,-1]; slots 1 .. LUT_LDS-2: ranks 0 .. LDS_RANKS-1; last: far marker
constexpr unsigned LDS_RANKS = LUT_LDS - 2;
constexpr unsigned OFF_FAR = 8 * SLOT_FAR;
constexpr unsigned CODE_ESC = 65535;                    // second table: read the fp64 table instead
__host__ __device__ inline unsigned cell_code(unsigned rank) { return rank < LDS_RANKS ? 8u * (rank + 1u) : OFF_FAR; }
// geometry of the padded strip table
__host__ __device__ inline int map_rows_padded(int H) { return ((H + 2 + 7) >> 3) << 3; }
__host__ __device__ inline size_t map_cells(int H, int W) { return (size_t)((W >> 3) + 2) * map_rows_padded(H) * 8; } // strip 0 only holds the left border column
// element index of map cell (r, c), -1 <= r <= H, -1 <= c <= W
__host__ __device__ inline size_t cell_elem(int r, int c, int Hp) { return ((size_t)((c >> 3) + 1) * Hp + (size_t)(r + 1)) * 8 + (size_t)(c & 7); }

struct MapDev {
    const uint16_t *cells;  // padded strips [(W >> 3) + 2][Hp][8] of LDS byte offsets
    const uint16_t *cells_far; // same layout: rank (<= 65534) of the cells marked OFF_FAR, 65535 = fp64 table
    unsigned cells_bytes;
    unsigned strip_bytes;   // Hp * 16, Hp = H + 2 rounded up to a multiple of 8
    const double *lut;      // [lut_len <= 65535] resolution*sqrt(d2_k), indexed by rank k
    const double *lut_lds;  // [LUT_LDS] image staged in LDS: dt[-1,-1], lut[0..LDS_RANKS-1], -0.0
    const double *dt;       // [H*W] exact fp64 distance table (escape path, rarely touched)
    int H, W;
    double res, rinv, ox, oy, oc, os, wres, hres, oob; // oob = dt[H-1][W-1]
    unsigned lut_len;       // entries of lut
};

// device-side view of MapDev with the cell table behind a buffer resource descriptor
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct MapView {
    __amdgpu_buffer_rsrc_t cells_rsrc;
    u32x4 cells_words;   // the same descriptor as four words (an asm statement's operand)
    unsigned row_bias, strip_m16; // row_bias = strip_bytes + 16, strip_m16 = strip_bytes - 16
    const MapDev *desc;  // rare paths (far cells, escape cells) re-read their table pointers from the descriptor:
                         // pointers that need not be kept in scalar registers across the march loop
    int H, W;
    double res, rinv, ox, oy, oc, os, wres, hres;
    double nox, noy; // -ox * rinv, -oy * rinv (exact when rinv is a power of two)
    F110_BOUNDS_ONLY(uint32_t *err = nullptr; unsigned cells_bytes = 0;)
    __device__ void init(const MapDev &m)
    {
        F110_BOUNDS_ONLY(cells_bytes = m.cells_bytes;)
        cells_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t *>(m.cells), 0, (int)m.cells_bytes, 0x00020000);
        cells_words.x = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)m.cells);
        cells_words.y = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)m.cells >> 32) & 0xffffu);
        cells_words.z = __builtin_amdgcn_readfirstlane(m.cells_bytes); cells_words.w = __builtin_amdgcn_readfirstlane(0x00020000u);
        row_bias = m.strip_bytes + 16u; strip_m16 = m.strip_bytes - 16u; desc = &m; H = m.H; W = m.W; res = m.res; rinv = m.rinv;
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
// "x_rot < 0 or x_rot >= width*res" (:79) is exactly "floor(q) outside [0, W)".
// laser_models.py:71-84 (xy_2_rc): the cell (column ci, row ri) of a point, un-clamped (a saturating conversion: any value
// outside [0, W) x [0, H) means "out of bounds", which cell_offset maps onto the table's border = the reference's dt[-1, -1] read).
template <bool IDENT, bool POW2>
__device__ inline void cell_index(const MapView &m, double x, double y, int &ci, int &ri)
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
    ci = (int)fx; ri = (int)fy; // saturating conversion; the clamp in cell_offset finishes the job
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
}

// byte offset of the cell under (column ci, row ri) in the strip table: both clamped onto the border
__device__ inline unsigned cell_offset(const MapView &m, int ci, int ri)
{
    const int cc = med3_i32(ci, -1, m.W);      // column -1..W (both ends are border cells)
    const int rr = med3_i32(ri, -1, m.H);      // row -1..H
    // strip (cc >> 3) + 1 (arithmetic shift: column -1 is the last column of strip 0), 16 bytes per row inside a strip.  The
    // +1 strip and the +1 border row ride in the constant of the shift-add (`row_bias` = strip_bytes + 16, a multiple of 16),
    // so no add is spent on the padding and the offset never goes negative; asm so that the constant is not re-associated
    // into a trailing add.  (c >> 3) * S + (c & 7) * 2 == (c >> 3) * (S - 16) + c * 2: no masking of the column bits needed
    unsigned row16;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(row16) : "v"(rr), "s"(m.row_bias));
    unsigned rc;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(rc) : "v"(cc), "v"(row16));
    return (unsigned)(__mul24(cc >> 3, (int)m.strip_m16) + (int)rc);
}

// One table look-up (laser_models.py:56-104 distance_transform): the value of the cell under (x, y), or -0.0 when the cell
// carries the far marker (dist_lookup_far finishes those).  The caller runs it under the EXEC mask of the rays that are
// still marching: a finished ray issues nothing.
template <bool IDENT, bool POW2>
__device__ inline double dist_lookup(const MapView &m, const double *lds_lut, double x, double y)
{
    int ci, ri;
    cell_index<IDENT, POW2>(m, x, y, ci, ri);
    const unsigned off = cell_offset(m, ci, ri);
    // buffer load: 32-bit per-lane offset against a scalar descriptor
    unsigned code = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(m.cells_rsrc, (int)off, 0, 0);
#if defined(F110_BOUNDS)
    F110_BCHK(off + 2u <= m.cells_bytes, BT_LUT_CODE, m.err);
    F110_BCHK(code <= OFF_FAR && (code & 7u) == 0u, BT_LUT_CODE, m.err);
    code = code <= OFF_FAR ? (code & ~7u) : 0u;
#endif
    // the loaded value IS the LDS byte offset of the distance: one ds_read_b64
    double d = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(lds_lut) + code);
    asm volatile("" : "+v"(d)); // pin the LDS read (keeps it a ds_read, not a flat load through a selected pointer)
    return d;
}

// the rare path behind the far marker: the cell's full rank from the second table, then the global LUT or the fp64 table
template <bool IDENT, bool POW2>
__device__ inline double dist_lookup_far(const MapView &m, double x, double y)
{
    int ci, ri;
    cell_index<IDENT, POW2>(m, x, y, ci, ri);
    unsigned off = cell_offset(m, ci, ri);
    const MapDev *dp = m.desc;
    asm volatile("" : "+s"(dp)); // opaque: the loads below stay here instead of being hoisted to the kernel entry
#if defined(F110_BOUNDS)
    F110_BCHK(off + 2u <= dp->cells_bytes, BT_CELLS_FAR, m.err);
    if (off + 2u > dp->cells_bytes) off = 0u;
#endif
    unsigned rank = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(dp->cells_far) + (size_t)off);
    const bool inside = (unsigned)ri < (unsigned)m.H && (unsigned)ci < (unsigned)m.W; // (a border cell never carries the far marker)
#if defined(F110_BOUNDS)
    F110_BCHK(rank == CODE_ESC || rank < dp->lut_len, BT_LUT_RANK, m.err);
    if (rank != CODE_ESC && rank >= dp->lut_len) rank = 0u;
    F110_BCHK(rank != CODE_ESC || inside, BT_DT, m.err);
#endif
    if (rank == CODE_ESC && !inside) return dp->oob;
    return (rank != CODE_ESC) ? dp->lut[rank] : dp->dt[(size_t)(unsigned)ri * (unsigned)m.W + (unsigned)ci];
}
__device__ inline bool is_far_marker(double d) { return __double2hiint(d) == (int)0x80000000; } // -0.0 (no table value is negative)

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
// `guard2`: twice the guard band in units of 2^-32, less one (860 = 2 x 1e-7), or 0xffffffff when T0 is not a valid fixed-point
// start (NaN / infinite yaw): then every lane replays.  For a valid T0 the un-wrapped index stays inside the repeated table
// (upload_cs sizes it), so no per-lane compare against its length -- and no re-read of that length in every refill -- is needed.
__device__ inline int beam_theta_index(unsigned long long T0, double t0w, int b, const ScanDev &s, unsigned guard2)
{
    const unsigned inc_lo = (unsigned)s.inc_fx, inc_hi = (unsigned)(s.inc_fx >> 32);
    unsigned long long t = (unsigned long long)(unsigned)b * inc_lo + T0;    // v_mad_u64_u32
    unsigned hi = (unsigned)(t >> 32) + (unsigned)b * inc_hi;                // inc_hi, b < 2^24
    const unsigned lo = (unsigned)t;
    int idx = (int)(hi >> 8);
    const unsigned frac = __builtin_amdgcn_alignbit(hi, lo, 8);             // top 32 fraction bits
    if (__builtin_expect(frac + 430u <= guard2, 0)) {                         // within 1e-7 of an integer
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

// The march phase of a wave for a map whose origin is not rotated and whose resolution is a power of two (cell_index's
// one-fma form), written out: every ray that is still marching takes table look-ups (laser_models.py:129-142) until at most
// `go` rays are left.  The loop runs under the EXEC mask of the marching rays and narrows it with v_cmpx as rays finish, so a
// finished ray issues no look-up, keeps its total and costs no select; what remains per iteration is
//   2 fma + 2 floor + 2 cvt (the cell), 2 med3 + 2 shift-add + shift + mad (its byte offset), the look-up (buffer_load_ushort ->
//   ds_read_b64), total += d, x += d * c, y += d * s (5, contraction off), 2 v_cmpx          = 19 VALU, 5 SALU
// against 21 VALU + 11 SALU for the compiler's branch-free form of round 4 (a select that parked finished lanes on an
// out-of-range offset, a compare for the far marker, the active mask kept in SGPRs by s_and / s_andn2 / s_or).
//   am: in, the rays marching; out, the rays still marching.  nlook += look-ups made.  d: every lane's last table value.
// The loads and their waits are inside the statement (the compiler does not count an asm load).  The LDS LUT must sit at LDS
// address 0 (scan_kernel checks it).
__device__ inline void march_ident_pow2(const MapView &m, double &x, double &y, double &total, double &d, double c, double s,
                                        double eps, double max_range, unsigned long long &am, int go, unsigned &nlook, int &nact)
{
    unsigned long long sx;
    double q0, q1;
    int t0, t1, t2;
    asm volatile(
        "s_mov_b64 %[sx], exec\n\t"
        "s_mov_b64 exec, %[am]\n"
        "1:\n\t"
        "s_add_u32 %[nl], %[nl], %[na]\n\t"
        "v_fma_f64 %[q0], %[rinv], %[x], %[nox]\n\t"
        "v_fma_f64 %[q1], %[rinv], %[y], %[noy]\n\t"
        "v_floor_f64 %[q0], %[q0]\n\t"
        "v_floor_f64 %[q1], %[q1]\n\t"
        "v_cvt_i32_f64 %[t0], %[q0]\n\t"
        "v_cvt_i32_f64 %[t1], %[q1]\n\t"
        "v_med3_i32 %[t0], %[t0], -1, %[W]\n\t"
        "v_med3_i32 %[t1], %[t1], -1, %[H]\n\t"
        "v_lshl_add_u32 %[t1], %[t1], 4, %[rb]\n\t"
        "v_ashrrev_i32 %[t2], 3, %[t0]\n\t"
        "v_lshl_add_u32 %[t1], %[t0], 1, %[t1]\n\t"
        "v_mad_i32_i24 %[t2], %[t2], %[sm], %[t1]\n\t"
        "buffer_load_ushort %[t2], %[t2], %[rsrc], 0 offen\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "ds_read_b64 %[d], %[t2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_f64 %[tot], %[tot], %[d]\n\t"
        "v_mul_f64 %[q0], %[c], %[d]\n\t"
        "v_mul_f64 %[q1], %[s], %[d]\n\t"
        "v_add_f64 %[x], %[x], %[q0]\n\t"
        "v_add_f64 %[y], %[y], %[q1]\n\t"
        "v_cmpx_lt_f64 vcc, %[eps], %[d]\n\t"
        "v_cmpx_ge_f64 vcc, %[mr], %[tot]\n\t"
        "s_bcnt1_i32_b64 %[na], exec\n\t"
        "s_cmp_gt_i32 %[na], %[go]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 %[am], exec\n\t"
        "s_mov_b64 exec, %[sx]"
        : [x] "+v"(x), [y] "+v"(y), [tot] "+v"(total), [d] "+v"(d), [am] "+s"(am), [nl] "+s"(nlook), [na] "+s"(nact),
          [sx] "=&s"(sx), [q0] "=&v"(q0), [q1] "=&v"(q1), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
        : [c] "v"(c), [s] "v"(s), [nox] "v"(m.nox), [noy] "v"(m.noy), [rinv] "s"(m.rinv), [W] "s"(m.W), [H] "s"(m.H),
          [rb] "s"(m.row_bias), [sm] "s"(m.strip_m16), [rsrc] "s"(m.cells_words), [eps] "s"(eps), [mr] "s"(max_range), [go] "s"(go)
        : "vcc", "scc", "memory");
}

// The same loop for a car that is not absurdly far from its map (scan_kernel decides per car: |cell coordinates of the car| +
// max_range / resolution + 2 below `march_fast_limit`; every look-up of a ray is made within max_range of the car, because the
// march only continues while total <= max_range).  Two things become possible, both exact:
//  * floor + int conversion by the MAGIC NUMBER: q + 1.5 * 2^52 rounded TOWARD MINUS INFINITY is floor(q) + 1.5 * 2^52 exactly (the
//    sum's ulp is 1), and the low dword of that double is floor(q) as a two's complement int for |q| < 2^31: one v_add_f64 under
//    round mode -inf (s_setreg on MODE's f64 rounding field around the pair; the fma that forms q and the sums of the march stay
//    round-to-nearest) instead of v_floor_f64 + v_cvt_i32_f64;
//  * no clamp of the COLUMN: a column outside [-8, W + 8) forms an offset outside the table (negative ones wrap to large
//    unsigned values), which the descriptor's range check answers with 0 = the border's code; the columns in between lie in the
//    border strips.  That needs (c >> 3) * strip_bytes below 2^31, which the limit guarantees.  (The row still needs its clamp:
//    a row beyond the strip would land in the neighbouring strip.)
// 16 VALU + 6 SALU per iteration.  The magic sums and the products live in v[60:63]: an asm operand cannot name the low dword
// of a register pair, so the statement uses those four registers by name and declares them clobbered.
__device__ inline void march_ident_pow2_fast(const MapView &m, double &x, double &y, double &total, double &d, double c, double s,
                                             double eps, double max_range, unsigned long long &am, int go, unsigned &nlook, int &nact)
{
    unsigned long long sx;
    asm volatile(
        "s_mov_b64 %[sx], exec\n\t"
        "s_mov_b64 exec, %[am]\n"
        "1:\n\t"
        "s_add_u32 %[nl], %[nl], %[na]\n\t"
        "v_fma_f64 v[60:61], %[rinv], %[x], %[nox]\n\t"
        "v_fma_f64 v[62:63], %[rinv], %[y], %[noy]\n\t"
#if defined(F110_X_NOMAGIC) // timing experiment
        "v_floor_f64 v[60:61], v[60:61]\n\t"
        "v_floor_f64 v[62:63], v[62:63]\n\t"
        "v_cvt_i32_f64 v60, v[60:61]\n\t"
        "v_cvt_i32_f64 v62, v[62:63]\n\t"
#else
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"      // f64 rounding: toward -inf
        "v_add_f64 v[60:61], v[60:61], %[magic]\n\t"
        "v_add_f64 v[62:63], v[62:63], %[magic]\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\t"      // back to nearest-even
#endif
#if defined(F110_X_PAD)
        "v_mov_b32 v61, v61\n\tv_mov_b32 v61, v61\n\tv_mov_b32 v61, v61\n\t"
#endif
        "v_med3_i32 v62, v62, -1, %[H]\n\t"
        "v_ashrrev_i32 v61, 3, v60\n\t"
        "v_lshl_add_u32 v62, v62, 4, %[rb]\n\t"
        "v_lshl_add_u32 v62, v60, 1, v62\n\t"
        "v_mad_i32_i24 v61, v61, %[sm], v62\n\t"
        "buffer_load_ushort v61, v61, %[rsrc], 0 offen\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "ds_read_b64 %[d], v61\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_f64 %[tot], %[tot], %[d]\n\t"
        "v_mul_f64 v[60:61], %[c], %[d]\n\t"
        "v_mul_f64 v[62:63], %[s], %[d]\n\t"
        "v_add_f64 %[x], %[x], v[60:61]\n\t"
        "v_add_f64 %[y], %[y], v[62:63]\n\t"
        "v_cmpx_lt_f64 vcc, %[eps], %[d]\n\t"
        "v_cmpx_ge_f64 vcc, %[mr], %[tot]\n\t"
        "s_bcnt1_i32_b64 %[na], exec\n\t"
        "s_cmp_gt_i32 %[na], %[go]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b64 %[am], exec\n\t"
        "s_mov_b64 exec, %[sx]"
        : [x] "+v"(x), [y] "+v"(y), [tot] "+v"(total), [d] "+v"(d), [am] "+s"(am), [nl] "+s"(nlook), [na] "+s"(nact), [sx] "=&s"(sx)
        : [c] "v"(c), [s] "v"(s), [nox] "v"(m.nox), [noy] "v"(m.noy), [rinv] "s"(m.rinv), [H] "s"(m.H),
          [rb] "s"(m.row_bias), [sm] "s"(m.strip_m16), [rsrc] "s"(m.cells_words), [eps] "s"(eps), [mr] "s"(max_range), [go] "s"(go),
          [magic] "s"(6755399441055744.0)
        : "vcc", "scc", "memory", "v60", "v61", "v62", "v63");
}
// The march for a map whose resolution is NOT a power of two (most F1TENTH maps: 0.05 m) and whose origin is not rotated, for
// cars near their map (the same per-car test as above).  q = (x - ox) * (1 / res) is within ~2e-16 relative of the reference's
// quotient (x - ox) / res, so its floor is the reference's cell unless q lies within 1e-9 of an integer: the loop computes the
// fractions of both coordinates and LEAVES (flag = 1, nothing of the iteration done) when any marching ray is that close; the
// caller then runs that one iteration through dist_lookup, which replays the reference's own division (cell_index), and comes
// back.  Otherwise as march_ident_pow2_fast: floor by the magic number under round-toward -inf (and back to a double to form the
// fraction), EXEC-masked, v_cmpx.  27 VALU + 7 SALU per iteration; the compiler's loop over dist_lookup is ~33 + ~20.
// a wave-uniform double as a scalar-register operand (the compiler may hold it in vector registers where it feeds vector code)
__device__ inline double uniform_f64(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ inline int march_ident_np_fast(const MapView &m, double &x, double &y, double &total, double &d, double c, double s,
                                          double eps, double max_range, unsigned long long &am, int go, unsigned &nlook, int &nact)
{
    unsigned long long sx;
    double qx, qy, fx, fy;
    int flag;
    // (wave-uniform by construction; said again for the register allocator: the values travel round a loop through compiler code)
    am = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(am >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)am);
    nlook = (unsigned)__builtin_amdgcn_readfirstlane((int)nlook);
    nact = __builtin_amdgcn_readfirstlane(nact);
    go = __builtin_amdgcn_readfirstlane(go);
    asm volatile(
        "s_mov_b64 %[sx], exec\n\t"
        "s_mov_b64 exec, %[am]\n"
        "1:\n\t"
        "v_add_f64 %[qx], %[x], -%[ox]\n\t"
        "v_add_f64 %[qy], %[y], -%[oy]\n\t"
        "v_mul_f64 %[qx], %[rinv], %[qx]\n\t"
        "v_mul_f64 %[qy], %[rinv], %[qy]\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\t"      // f64 rounding: toward -inf
        "v_add_f64 v[60:61], %[qx], %[magic]\n\t"
        "v_add_f64 v[62:63], %[qy], %[magic]\n\t"
        "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\t"      // back to nearest-even
        "v_add_f64 %[fx], v[60:61], -%[magic]\n\t"                // floor(q) as a double (exact)
        "v_add_f64 %[fy], v[62:63], -%[magic]\n\t"
        "v_add_f64 %[fx], %[qx], -%[fx]\n\t"                      // fraction (exact)
        "v_add_f64 %[fy], %[qy], -%[fy]\n\t"
        "v_add_f64 %[fx], %[fx], -0.5\n\t"
        "v_add_f64 %[fy], %[fy], -0.5\n\t"
        "v_max_f64 %[fx], |%[fx]|, |%[fy]|\n\t"
        "v_cmp_lt_f64 vcc, %[thr], %[fx]\n\t"                     // a fraction within 1e-9 of 0 or 1
        "s_cbranch_vccnz 3f\n\t"
        "s_add_u32 %[nl], %[nl], %[na]\n\t"
        "v_med3_i32 v62, v62, -1, %[H]\n\t"
        "v_ashrrev_i32 v61, 3, v60\n\t"
        "v_lshl_add_u32 v62, v62, 4, %[rb]\n\t"
        "v_lshl_add_u32 v62, v60, 1, v62\n\t"
        "v_mad_i32_i24 v61, v61, %[sm], v62\n\t"
        "buffer_load_ushort v61, v61, %[rsrc], 0 offen\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "ds_read_b64 %[d], v61\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_add_f64 %[tot], %[tot], %[d]\n\t"
        "v_mul_f64 v[60:61], %[c], %[d]\n\t"
        "v_mul_f64 v[62:63], %[s], %[d]\n\t"
        "v_add_f64 %[x], %[x], v[60:61]\n\t"
        "v_add_f64 %[y], %[y], v[62:63]\n\t"
        "v_cmpx_lt_f64 vcc, %[eps], %[d]\n\t"
        "v_cmpx_ge_f64 vcc, %[mr], %[tot]\n\t"
        "s_bcnt1_i32_b64 %[na], exec\n\t"
        "s_cmp_gt_i32 %[na], %[go]\n\t"
        "s_cbranch_scc1 1b\n\t"
        "s_mov_b32 %[flag], 0\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "s_mov_b32 %[flag], 1\n"
        "4:\n\t"
        "s_mov_b64 %[am], exec\n\t"
        "s_mov_b64 exec, %[sx]"
        : [x] "+v"(x), [y] "+v"(y), [tot] "+v"(total), [d] "+v"(d), [am] "+s"(am), [nl] "+s"(nlook), [na] "+s"(nact), [sx] "=&s"(sx),
          [qx] "=&v"(qx), [qy] "=&v"(qy), [fx] "=&v"(fx), [fy] "=&v"(fy), [flag] "=&s"(flag)
        : [c] "v"(c), [s] "v"(s), [ox] "s"(uniform_f64(m.ox)), [oy] "s"(uniform_f64(m.oy)), [rinv] "s"(uniform_f64(m.rinv)), [H] "s"(m.H),
          [rb] "s"(m.row_bias), [sm] "s"(m.strip_m16), [rsrc] "s"(m.cells_words), [eps] "s"(eps), [mr] "s"(max_range), [go] "s"(go),
          [magic] "s"(6755399441055744.0), [thr] "s"(0.5 - 1e-9)
        : "vcc", "scc", "memory", "v60", "v61", "v62", "v63");
    return flag;
}
// the largest |cell coordinate| a look-up of the fast march may have: (c >> 3) * strip_bytes stays below 2^31 with room to
// spare, and far inside the magic number's 2^31 (a NaN or infinite pose fails the test and takes the clamped loop)
__device__ inline double march_fast_limit(const MapView &m) { return (double)((0x7fffffffu / (m.row_bias + 16u)) * 8u) - 64.0; }

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
    // The noise table (one 8-B gather per beam taken): [noise_slots][noise_cap][nb], a ring of noise_cap = noise_mask + 1 rows
    // per slot.  Base and size travel BY VALUE -- through the device-resident
    // descriptor every wave started with a chain of two dependent scalar loads, 0.9 % of the launch (profiles/r04_noise.txt)
    // -- so they only change when the table is re-allocated (f110_launch_epoch moves then; a ring that follows the cars
    // never is).  The window of rows that are present moves all the time: it stays behind the descriptor and is checked
    // by dynamics_kernel, off this kernel's path.
    const double *noise_base;
    int noise_mask, noise_cap, noise_slots;
    const double *side;          // [nb] side distances (base_classes.py:123-156), read only for iTTC candidates
    double side_max;             // their largest finite value: scan values at or above side_max + margin cannot be candidates
    const int32_t *env_noise;    // [B] noise slot (= seed) of every env, or NULL (all envs on slot 0)
    uint32_t *dev_err;           // device error word (f110_device_errors): F110_DEVERR_* bits, or NULL
    const double *beam_cosines;  // [nb]
    double ttc_thresh;
    uint8_t *in_collision;       // [N]
    const uint8_t *pending_reset;// [B]
    int reset_only;              // 1: only envs with pending_reset are processed
    const uint16_t *chunk_beam0; // [ceil(nb/64)] first beam of the k-th 64-beam chunk to be marched (long rays first)
    // Launch order (or NULL = car order): the wave that would march car i marches car order[i].  A permutation of the shard's
    // cars that only changes WHICH wave marches WHICH car -- results are indexed by the car -- so that cars standing on the
    // same noise row can be launched next to each other (f110_set_scan_order; Engine keeps it sorted by the envs' row counters:
    // in a batch whose envs were reset at different times every env reads its own row, 566 MB per step at 65 536 envs, and the
    // rows of neighbouring waves then come from the L1 / L2 instead of HBM).  Single-map handles only (a workgroup stages ONE LUT).
    const int32_t *order;
    // 1: workgroups of ONE wave (block = 64 threads) -- every car stages its own map's LUT, so neighbouring cars may stand on
    // different maps (f110_assign_maps with a map per env); 0: SCAN_WAVES cars per workgroup share one LUT copy.
    int wg_single;
    int n_maps;             // slots of `maps` (bounds build)
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

// SM 0: ScanSimulator2D.scan(pose, None); 1: the scan of a step (noise, iTTC flag; env_kernel follows); 2: the same with
// ordinary instead of streaming stores for the fp32 scan (launches of more than ~300 000 cars, see emit).
template <bool IDENT, bool POW2, int SM>
#ifndef F110_SCAN_MIN_WAVES
#define F110_SCAN_MIN_WAVES 8
#endif
__global__ __launch_bounds__(SCAN_THREADS, F110_SCAN_MIN_WAVES) void scan_kernel(ScanArgs a)
{
    constexpr bool STEP = SM >= 1;
    // ONE LDS object, the LUT first: march_ident_pow2 addresses the LUT by the cell codes alone, i.e. the LUT sits at LDS
    // address 0 (the kernel has no other LDS variable; every parity test would fail otherwise)
    __shared__ struct __attribute__((aligned(16))) { double lut[LUT_LDS]; int chunk0[MAX_CHUNKS]; } s_mem;
    double *const s_lut = s_mem.lut;
    int *const s_chunk0 = s_mem.chunk0;
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
    const int wg_single = rare->wg_single;
    const int wid = wg_single ? (int)blockIdx.x : (int)blockIdx.x * SCAN_WAVES + wave; // wave-uniform (scalar)
    // wave -> (car, part of its beam queue).  Kept to one extra argument and shifts: this kernel sits at
    // the 80-SGPR budget of 8 waves/SIMD, and a scalar spilled inside the refill loop costs ~3 % of the launch.
    int wpc, car, part, lg; // (wpc = 2^lg: the divisions by it below are shifts -- a scalar integer division is ~25 dependent instructions)
    {
        // (the stage list read from the kernel arguments with constant indices instead -- no dependent loads -- measured no better)
        int t = wid, c = 0, st = 0;
        const int ns = rare->n_stages;
        for (; st < ns; st++) {
            const int cars_s = rare->stage_cars[st], w = cars_s << rare->stage_log2w[st];
            if (t < w) break;
            t -= w; c += cars_s;
        }
        lg = st < ns ? rare->stage_log2w[st] : 0;
        wpc = 1 << lg; car = st < ns ? c + (t >> lg) : a.n_cars; part = t & (wpc - 1);
    }
    // the car's map (wave-uniform: scalar loads); waves past the last car still help to stage the LUT
    int car_c = rare->car_base + min(car, a.n_cars - 1);
    if (rare->order) car_c = rare->order[car_c]; // (launch position -> car; wave-uniform: a scalar load)
    const int env_c = a.agents == 1 ? car_c : car_c / a.agents; // (one agent: no division at run time)
    F110_BCHK(rare->n_stages >= 1 && rare->n_stages <= SCAN_MAX_STAGES, BT_STAGE_LIST, rare->dev_err);
    int map_slot = a.env_map ? a.env_map[env_c] : 0;
#if defined(F110_BOUNDS)
    F110_BCHK((unsigned)map_slot < (unsigned)rare->n_maps, BT_MAP_SLOT, rare->dev_err);
    if ((unsigned)map_slot >= (unsigned)rare->n_maps) map_slot = 0;
#endif
    const MapDev &md = a.maps[map_slot];
    {   // LDS image of the LUT prepared by the host (slot 0 = dt[-1,-1], last slot = the far marker): 16-B copies
        const double2 *src = reinterpret_cast<const double2 *>(md.lut_lds);
        double2 *dst = reinterpret_cast<double2 *>(s_lut);
        for (int i = threadIdx.x; i < LUT_LDS / 2; i += SCAN_THREADS) dst[i] = src[i];
        if (SCAN_WAVES > 1 && wg_single) // (a lone wave copies the other waves' shares too)
            for (int k = 1; k < SCAN_WAVES; k++)
                for (int i = threadIdx.x + k * WAVE; i < LUT_LDS / 2; i += SCAN_THREADS) dst[i] = src[i];
    }
    static_assert(MAX_CHUNKS <= WAVE, "one pass of one wave stages the chunk table");
    if ((int)threadIdx.x < ((nb + 63) >> 6)) s_chunk0[threadIdx.x] = a.chunk_beam0[threadIdx.x];
#if defined(F110_TIMELINE)
    { unsigned long long t = wall_clock64(); asm volatile("" : "+v"(t)); if (lane == 0) s_tl[wave][0] = t; }
#endif
    __syncthreads();
    MapView mv;
    mv.init(md);
    F110_BOUNDS_ONLY(mv.err = rare->dev_err;)
    if (car >= a.n_cars) return;
    car = car_c; // (from here on the car's index in the shard)
    // this wave's slice of the car's beam queue: chunk positions part, part+wpc, ...
    const int nch = (nb + 63) >> 6;
    const int my_chunks = nch > part ? (nch - part + wpc - 1) >> lg : 0;
    const int owns_last = my_chunks > 0 && ((nch - 1) & (wpc - 1)) == part;
    const int nbl = my_chunks * 64 - (owns_last ? nch * 64 - nb : 0); // beams of this wave
    if (a.reset_only && !a.pending_reset[env_c]) return; // (car == car_c here: the waves past the last car have left)

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
    // ... and only scan values below (largest side distance + cand) can be candidates at all: |v - side_i| < cand needs
    // v < side_i + cand <= side_max + cand (the margin covers the roundings of the sum and of v - side_i), so the beam's side
    // distance is read in that rare case only and the noise rows hold nothing but noise
    // (the margin is added, not multiplied in: a table of negative side distances must not pull the bound the wrong way)
    const double side_pre = do_ttc ? (rare->side_max + cand) + 1e-9 * (fabs(rare->side_max) + cand) : -__builtin_inf(); // (no iTTC test: no value is below it)
    const double *__restrict__ ns = nullptr;
    if (STEP) {
        // the car's noise row: row `scans since its reset` of its env's slot (a ring of noise_cap rows per slot)
        const int row = a.noise_step[car];
        int slot = a.env_noise ? a.env_noise[env_c] : 0;
#if defined(F110_BOUNDS)
        F110_BCHK(slot >= 0 && slot < rare->noise_slots, BT_NOISE_SLOT, rare->dev_err);
        if (!(slot >= 0 && slot < rare->noise_slots)) slot = 0;
#endif
        ns = a.noise_base + (size_t)(unsigned)(slot * a.noise_cap + (row & a.noise_mask)) * (size_t)(unsigned)nb; // (slots * cap rows < 2^31: noise_resize)
    }
    float *o32 = a.out_f32 ? a.out_f32 + (size_t)car * nb : nullptr;
    double *o64 = a.out_f64 ? a.out_f64 + (size_t)car * nb : nullptr;
    bool hit = false;

    // finishing stage of one beam: clamp (laser_models.py:143-144), noise (:450-452),
    // stores, iTTC (:189-217).  nzv: noise of the beam, loaded by the caller ahead of time.
    auto emit = [&](int i, double tot, double nzv) {
#if defined(F110_BOUNDS)
        F110_BCHK((unsigned)i < (unsigned)nb, BT_SCAN_STORE, rare->dev_err);
        if ((unsigned)i >= (unsigned)nb) return;
#endif
        double v; // :143-144 min(total, max_range) (a NaN total, i.e. a NaN pose, also clamps: v_min_f64 returns the other operand)
        asm("v_min_f64 %0, %1, %2" : "=v"(v) : "v"(tot), "s"(max_range)); // (fmin() would first quieten both operands: two more instructions)
        if (STEP) v += nzv;
        // Streaming (non-temporal) stores: the scan is written once and read by later kernels only; as ordinary stores the
        // 27 scattered store instructions of a car took their turn in the L1 beside the table look-ups, which are what bounds
        // this kernel -- 0.683 -> 0.654 ms per 65 536 cars (profiles/r04_scan_stores.txt)
        // (Very large launches are the exception -- 524 288 cars: 4.78 ms with ordinary stores, 4.95 ms with streaming ones, whose
        // partial lines reach the HBM un-merged; 65 536: 0.683 / 0.654, 262 144: 2.476 / 2.459 -- so the host picks the
        // instantiation by the launch's size.  A run-time flag tested here costs 3.4 % of the launch.)
        if (o32) {
            float *q = reinterpret_cast<float *>(reinterpret_cast<char *>(o32) + (size_t)((unsigned)i * 4u));
            if (SM == 2) *q = (float)v;
            else __builtin_nontemporal_store((float)v, q);
        }
        if (o64) {
            double *q64 = reinterpret_cast<double *>(reinterpret_cast<char *>(o64) + (size_t)((unsigned)i * 8u));
            if (SM == 2) *q64 = v;
            else __builtin_nontemporal_store(v, q64);
        }
        if (__builtin_expect(v < side_pre, 0)) {
            const ScanArgs *ra = rare;
            asm volatile("" : "+s"(ra)); // re-read the rarely needed arguments here instead of holding them in SGPRs
            const double sd = v - ra->side[i];
            if (fabs(sd) < cand) {
                const double proj_vel = vel * ra->beam_cosines[i];
                const double ttc = sd / proj_vel;
                if ((ttc < ra->ttc_thresh) && (ttc >= 0.0)) hit = true;
            }
        }
    };

    // ---- ray march (laser_models.py:107-186) -------------------------------------
    // The first table read of every beam is at the car itself (:129): done once.
    double d0 = dist_lookup<IDENT, POW2>(mv, s_lut, px, py);
    if (__builtin_expect(is_far_marker(d0), 0)) d0 = dist_lookup_far<IDENT, POW2>(mv, px, py); // (wave-uniform: every lane reads the car's own cell)
#if defined(F110_TIMELINE)
    { unsigned long long t = wall_clock64(); asm volatile("" : "+v"(t)); if (lane == 0) s_tl[wave][1] = t; }
#endif
#if defined(F110_TIMELINE)
    unsigned tl_wit = 0, tl_wit_dry = 0; // march iterations of the wave so far / when its queue ran dry
    unsigned tl_refills = 0, tl_phases = 0; // refill phases in which beams were taken / passes of the outer loop
#endif
    unsigned nlook = (unsigned)nbl; // the reference reads the table once per beam before marching
    if (!(d0 > eps && d0 <= max_range)) {
        for (int k = lane; k < nbl; k += WAVE) {
            F110_BCHK((k >> 6) * wpc + part < MAX_CHUNKS, BT_CHUNK_ORDER, rare->dev_err);
            int i = s_chunk0[((k >> 6) * wpc + part) & (MAX_CHUNKS - 1)] + (k & 63);
#if defined(F110_BOUNDS)
            F110_BCHK((unsigned)i < (unsigned)nb, BT_NOISE_BEAM, rare->dev_err);
            if ((unsigned)i >= (unsigned)nb) i = 0;
#endif
            emit(i, d0, STEP ? ns[i] : 0.0);
        }
    } else {
        const double td = (double)a.scan.theta_dis;
        double t0w = td * (yaw - a.scan.fov / 2.) / (2. * F110_PI);
        t0w = fmod_small(t0w, td);
        while (t0w < 0) t0w += td;
        // 24.40 fixed point of t0w in [0, theta_dis); a NaN / out-of-range yaw falls to the slow path
        const bool t0_ok = t0w >= 0 && t0w < td;
        const unsigned long long T0 = t0_ok ? (unsigned long long)(t0w * 1099511627776.0) : 0ull;
        const unsigned guard2 = t0_ok ? 859u : 0xffffffffu;

#if defined(F110_TIMELINE)
        bool tl_dry = false;
#endif
        // (wave-uniform) may this car's rays take the fast march?  Every look-up lies within max_range of the car.
        bool fast = false;
        if (IDENT) {
            const double reach = max_range * mv.rinv + 2.0, lim = march_fast_limit(mv);
            const double q0x = POW2 ? __builtin_fma(px, mv.rinv, mv.nox) : (px - mv.ox) * mv.rinv;
            const double q0y = POW2 ? __builtin_fma(py, mv.rinv, mv.noy) : (py - mv.oy) * mv.rinv;
            fast = fabs(q0x) + reach < lim && fabs(q0y) + reach < lim;
        }
        int next = 0;           // wave-uniform: next unassigned slot of the beam order
        bool active = false;
        int beam = -1;          // beam whose result `total` holds (-1: none)
        double x = px, y = py, c = 0, s = 0, total = 0;
        double d = 0;           // the lane's last table value (kept across the phases: a parked -0.0 is a cell of the second table)
        double nz = 0;          // noise of the lane's beam (fetched when the beam is taken)
        for (;;) {
            // ---- cells of the second table (rare): a lane that read the far marker stopped with an exact no-op; finish its
            // look-up here, once per phase instead of one compare per look-up
            {
                const bool farp = !active && is_far_marker(d);
                if (__builtin_expect(vote(farp) != 0ull, 0)) {
                    if (farp) {
                        d = dist_lookup_far<IDENT, POW2>(mv, x, y);
                        total += d;
                        x += d * c;
                        y += d * s;
                        active = (d > eps) && (total <= max_range);
                    }
                }
            }
            // ---- refill phase: idle lanes finish their beam and take the next one ----
            const unsigned long long idle = vote(!active);
            const int nidle = __popcll(idle);
#if defined(F110_TIMELINE)
            tl_phases++;
            if (next < nbl) tl_refills++;
#endif
            if (!active) {
                // all independent loads first (one memory round trip).  The noise entry is fetched
                // for the beam being TAKEN and carried in registers until the beam is finished: idle lanes take
                // consecutive beams, so this gather touches a few cache lines, where a gather by the FINISHED beams
                // (scattered over the scan) touched a line per lane -- the L1's tag pipeline is what bounds this kernel
                const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32),
                                    __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
                const int k = next + rank;
                const bool take = k < nbl;
                const int kk = take ? k : 0;
                F110_BCHK((kk >> 6) * wpc + part < MAX_CHUNKS, BT_CHUNK_ORDER, rare->dev_err);
                int b = s_chunk0[((kk >> 6) * wpc + part) & (MAX_CHUNKS - 1)] + (kk & 63);
#if defined(F110_BOUNDS)
                F110_BCHK((unsigned)b < (unsigned)nb, BT_NOISE_BEAM, rare->dev_err);
                if ((unsigned)b >= (unsigned)nb) b = 0;
#endif
#if defined(F110_X_NONOISE) // timing experiment: the upper bound of what the noise gather costs (results invalid)
                const double nsv = 0.0;
#else
                const double nsv = STEP ? *reinterpret_cast<const double *>(reinterpret_cast<const char *>(ns) + (size_t)((unsigned)b * 8u)) : 0.0;
#endif
                const double nzv = nz;
                int ti = beam_theta_index(T0, t0w, b, a.scan, guard2);
#if defined(F110_BOUNDS)
                F110_BCHK((unsigned)ti < (unsigned)a.scan.cs_len, BT_CS_TABLE, rare->dev_err);
                if ((unsigned)ti >= (unsigned)a.scan.cs_len) ti = 0;
#endif
                const double2 cs = *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(a.scan.cs) + (size_t)((unsigned)ti * 16u)); // second round trip, overlapped with emit()
                if (beam >= 0) emit(beam, total, nzv);
                beam = -1;
                c = cs.x; // (every idle lane: one that takes no beam never marches again, and the far look-ups are finished above)
                s = cs.y;
                if (take) {
                    x = px + d0 * c;
                    y = py + d0 * s;
                    total = d0;
                    beam = b;
                    nz = nsv;
                    active = true;
                }
            }
            next += nidle;
            int nact = __popcll(vote(active));
            if (nact == 0) break;
            // ---- march phase: the rays that are still marching step, under their own EXEC mask (a finished ray issues no
            // look-up and keeps its total), until enough lanes are idle again or, once no beams are left, the wave has drained ----
            const int go = next < nbl ? WAVE - REFILL_MIN_IDLE : 0; // keep marching while nact > go
#if defined(F110_DRAIN_PRIO)
            if (go == 0) __builtin_amdgcn_s_setprio(F110_DRAIN_PRIO); // experiment: a draining wave ends on its longest ray's dependent chain
#endif
#if defined(F110_TIMELINE)
            if (go == 0 && !tl_dry) { unsigned long long t = wall_clock64(); asm volatile("" : "+v"(t)); if (lane == 0) s_tl[wave][1] = t; tl_dry = true; tl_wit_dry = tl_wit; }
#endif
#if !defined(F110_TIMELINE) && !defined(F110_BOUNDS)
            if (IDENT && POW2) {
                unsigned long long am = vote(active);
                if (fast) march_ident_pow2_fast(mv, x, y, total, d, c, s, eps, max_range, am, go, nlook, nact);
                else march_ident_pow2(mv, x, y, total, d, c, s, eps, max_range, am, go, nlook, nact);
                active = ((am >> lane) & 1ull) != 0ull;
                continue;
            }
            if (IDENT && !POW2 && fast) {
                unsigned long long am = vote(active);
                while (march_ident_np_fast(mv, x, y, total, d, c, s, eps, max_range, am, go, nlook, nact)) {
                    // a marching ray's quotient lies within 1e-9 of an integer: this one iteration through dist_lookup, which
                    // replays the reference's own division for such lanes (cell_index)
                    nlook += (unsigned)nact;
                    bool act = __builtin_amdgcn_inverse_ballot_w64(am);
                    if (act) {
                        d = dist_lookup<IDENT, POW2>(mv, s_lut, x, y);
                        total += d;
                        x += d * c;
                        y += d * s;
                        act = (d > eps) && (total <= max_range);
                    }
                    am = vote(act);
                    nact = __popcll(am);
                    if (nact <= go) break;
                }
                active = ((am >> lane) & 1ull) != 0ull;
                continue;
            }
#endif
            do {
#if defined(F110_TIMELINE)
                tl_wit++;
#endif
                nlook += (unsigned)nact;
                if (active) {
                    d = dist_lookup<IDENT, POW2>(mv, s_lut, x, y);
                    total += d;
                    x += d * c;
                    y += d * s;
                    active = (d > eps) && (total <= max_range);
                }
                nact = __popcll(vote(active));
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
        // {start, queue ran dry (or first rays), end}; car << 8 | part | wpc << 40 | iterations after the queue ran dry << 44
        // (time stamps are 100 MHz ticks: their top 16 bits are free; they carry the wave's march iterations and its refills)
        tl[0] = s_tl[wave][0] | ((unsigned long long)min(tl_wit, 0xffffu) << 48); tl[1] = s_tl[wave][1] | ((unsigned long long)min(tl_phases, 0xffffu) << 48);
        tl[2] = tl_end | ((unsigned long long)min(tl_refills, 0xffffu) << 48);
        tl[3] = ((unsigned long long)car << 8) | (unsigned)part | ((unsigned long long)wpc << 40) | ((unsigned long long)min(tl_wit - tl_wit_dry, 0xfffffu) << 44);
    }
#endif

    // ---- iTTC result: the flag only; env_kernel zeroes the state (base_classes.py:241-250)
    // once every wave of the car is done.  Plain store: all writers store the same 1.
    if (STEP) {
        if (vote(hit) != 0ull && lane == 0) ra->in_collision[car] = 1;
    }
}

// ------------------------------------------------------------------ opponents (A > 1)
// RaceCar.ray_cast_agents (base_classes.py:204-225) -> ray_cast (laser_models.py:319-346): the car's CURRENT pose (yaw
// already zeroed by an iTTC hit, :245) against the other cars' post-integration snapshot poses.
// Two kernels behind the scan:
//   opp_setup_kernel, FOUR LANES PER (car, opponent) PAIR (lane c = corner c = edge c of the opponent's quad): every lane
//     takes one corner through the arctan2 / arg-min of get_blocked_view_indices (:283-315) and one edge through what
//     get_range (:250-280) computes from the pose and two corners alone; span, angular hull and the beam intervals worth
//     visiting come from quad-wide min / max;
//   opp_apply_kernel, a group of OPP_GROUP lanes per car (a whole wave as built): the pair is staged in LDS by vector loads
//     (no chain of dependent scalar loads at the head of the wave) and the lanes share the beams of the intervals.  Per
//     beam the front-facing edges are tested with get_range's own conditions (:271-274), decided without dividing; the
//     nearest hit is min over the passing edges of fl(cross / denom), and because rounding is monotonic that is fl() of
//     the EXACT smallest quotient: the edges are compared as exact fractions (frac_less), one division per beam.
// The kernel's time is instruction issue (PMC: one wave-iteration costs the chip what its VALU can issue), so what counts
// is instructions per beam and idle lanes: groups of 8 / 16 / 32 / 64 lanes take 52 / 33 / 27 / 23 us at 32 768 cars (a
// wave waits for its slowest group), and four lanes per car for set-up AND ray cast in one kernel 67 us.  History: one wave-per-car kernel
// with the trig inlined 0.165 ms at 32 768 cars (162 VGPRs) -> set-up / apply split 108 us -> wave per car 86 ->
// conservative pre-test 50 -> chunk masks 44 + 17 us (round 3) -> round 4: intervals from the corners' angular hull,
// per-edge constants, one division per beam, four lanes per pair in the set-up: 39 + 11 us -> front-facing edges only,
// the pair in LDS: 23 + 13 us (profiles/r04_opponents.txt).  Round 4 also measured two other placements, both exact, both slower: the ray
// cast INSIDE scan_kernel at the end of the wave that marched the car (+37 us on the scan: a wave that lingers for a chain
// of dependent loads keeps its slot from a marching one), and the set-up on a side stream beside the scan (the fork /
// join event waits cost the stream more than the set-up takes).
struct OppArgs {
    int n_cars, agents, nb;
    const double *state;      // [N,7]
    const double *pose_snap;  // [N,3]
    const uint8_t *in_collision; // [N]
    const double *scan_angles;
    const double2 *beam_cs;   // [nb] {cos, sin}(scan_angles)
    const Params *params;     // [slots, 1 + agents] (see DynArgs): a car sizes its opponents with its OWN params (base_classes.py:221)
    const int32_t *env_params;// [B] or NULL
    const uint8_t *pending_reset; // [B] as the step found it (dynamics_kernel's snapshot)
    int reset_only;
    struct OppPair *pairs;    // [N, agents-1] scratch owned by the handle
    float *scans32;           // [N,nb] or NULL
    double *scans64;          // [N,nb] or NULL
    int param_slots;          // (bounds-checked build only)
    uint32_t *dev_err;
};

// the params slot of an env (DynArgs::params): checked in the bounds build
__device__ inline int params_slot_of(const int32_t *env_params, int env, int param_slots, uint32_t *dev_err)
{
    int sl = env_params ? env_params[env] : 0;
#if defined(F110_BOUNDS)
    F110_BCHK((unsigned)sl < (unsigned)param_slots, BT_PARAMS_SLOT, dev_err);
    if ((unsigned)sl >= (unsigned)param_slots) sl = 0;
#endif
    (void)param_slots; (void)dev_err;
    return sl;
}

// exact "n1 / d1 < n2 / d2" for non-negative numerators and positive denominators: the rounded products decide unless
// they are equal, then the exact residuals of the two products do (fma).  No overflow / underflow for physical ranges.
__device__ inline bool frac_less(double n1, double d1, double n2, double d2)
{
    const double p1 = n1 * d2, p2 = n2 * d1;
    if (p1 != p2) return p1 < p2;
    return __builtin_fma(n1, d2, -p1) < __builtin_fma(n2, d1, -p2);
}

// One beam of the ray cast against one opponent: get_range over the four edges (see the section comment), then the
// in-place minimum (laser_models.py:343-344) on the fp64 and / or fp32 scan.
struct OppPairRegs {
    double px, py, cA, sA, qx, qy, reach;         // ego position; cos / sin(yaw + pi/2); opponent centre relative to the ego and
                                                  // its padded half diagonal (conservative pre-test, opp_test_beam)
    // Per edge: o - va (:258), vb - va (:259), cross(v2, v1) (:267), and its end points va, vb (collinear branch only).
    // The edges are stored FRONT-FACING FIRST (n_front of them: the car lies on their outer side).  A ray from outside a
    // convex quad enters through a front-facing edge, and the point where it leaves through a back-facing one is never
    // nearer, so min over the passing edges (:336-341) is decided by the front-facing ones; the others only keep their
    // test for the collinear branch (denom == 0, :275-280), which answers whatever the facing.  With the car INSIDE the
    // quad (overlapping cars) no edge faces it: n_front = 4, every edge takes the full test.
    double v1x[4], v1y[4], v2x[4], v2y[4], cr[4];
    double ax[4], ay[4], bx[4], by[4];
    int n_front, pad;
};
constexpr int OPP_MAX_IV = 3;
// what opp_setup_kernel leaves for opp_apply_kernel
struct OppPair {
    OppPairRegs r;
    // Disjoint, ascending beam intervals [iv[2k], iv[2k+1]] inside the reference's span [lo, hi] that hold every beam
    // get_range can answer for: a finite range needs the ray to point AT the quad, i.e. its angle inside the angular hull
    // of the four corners as seen from the car (or, for the collinear branch :275-280, exactly away from it: hull + pi).
    // Normally the hull IS [lo, hi].  With the opponent behind the car the corner angles straddle +-pi and the
    // reference's span becomes nearly the whole scan (:293-315) although only the few beams at its two ends -- and, for
    // the collinear branch, the beams pointing straight ahead -- lie in the hull: three short intervals instead of
    // ~1 000 beams.  Conservative (two beams of margin), so the beams that can be modified are all inside and the
    // results are the per-beam tests'.
    int n_iv, total, iv[2 * OPP_MAX_IV]; // total = beams in the intervals (0: nothing to do)
};

__device__ inline void opp_test_beam(const OppPairRegs *o, const double2 *__restrict__ beam_cs, int i, float *s32, double *s64)
{
    // (callers pass 0 <= i < num_beams: checked there, and reported by the bounds-checked build)
    // (*o lives in LDS: its fields are read where they are used, edge by edge, so that few of them are live at a time and
    // the kernel keeps 8 waves per SIMD -- its time is memory latency, which only more waves hide)
    const double2 cs = beam_cs[i];
    const double cA = o->cA, sA = o->sA;
    const double v3x = cA * cs.x - sA * cs.y, v3y = sA * cs.x + cA * cs.y;
    // v3 is the ray's unit normal (laser_models.py:262): |q . v3| is the distance of the opponent's centre
    // from the ray's line.  Beyond the padded half diagonal no edge can be crossed (every get_range would return inf).
    const double qx = o->qx, qy = o->qy, reach = o->reach;
    if (!(fabs(qx * v3x + qy * v3y) <= reach)) return;
    // The ray's direction is (v3y, -v3x).  If the whole bounding circle lies BEHIND the car along it, every edge
    // point has a negative ray parameter: get_range rejects it (d1 >= 0, :271) -- unless an edge is exactly
    // parallel to the ray (denom == 0: the collinear branch answers whatever the direction, :275-280).
    const bool behind = qx * v3y - qy * v3x < -reach;
    bool has = false;
    double bn = 0.0, bd = 1.0;          // the nearest hit so far as a fraction cross / denom (signed, as get_range divides them)
    double direct = __builtin_inf();    // distances the collinear branch produced
    const int n_front = o->n_front;
#pragma unroll 1
    for (int e = 0; e < 4; e++) {
        const double denom = o->v2x[e] * v3x + o->v2y[e] * v3y;          // :266
        if (fabs(denom) > 0.0) {
            if (behind || e >= n_front) continue;
            // d1 = cross / denom >= 0, 0 <= d2 = dot / denom <= 1 (:271-274) decided without dividing:
            // the sign of an IEEE quotient is the sign product, and fl(q) <= 1 <=> q <= 1.
            const double cr = o->cr[e];
            const double dt = o->v1x[e] * v3x + o->v1y[e] * v3y;       // :268
            const bool dpos = denom > 0.0;
            const bool d1_ok = (cr == 0.0) || ((cr > 0.0) == dpos);
            const bool d2_ge0 = (dt == 0.0) || ((dt > 0.0) == dpos);
            const bool d2_le1 = dpos ? (dt <= denom) : (dt >= denom);
            if (d1_ok && d2_ge0 && d2_le1) {
                if (!has || frac_less(fabs(cr), fabs(denom), fabs(bn), fabs(bd))) { bn = cr; bd = denom; }
                has = true;
            }
        } else {
            // are_collinear(o, va, vb) :233-247, then the nearer corner (:278-280)
            const double px = o->px, py = o->py;
            const double bax = o->ax[e] - px, bay = o->ay[e] - py;
            const double cax = px - o->bx[e], cay = py - o->by[e];
            if (fabs(bax * cay - bay * cax) < 1e-8) {
                const double ebx = o->bx[e] - px, eby = o->by[e] - py;
                const double da = sqrt(bax * bax + bay * bay), db = sqrt(ebx * ebx + eby * eby);
                const double d = da < db ? da : db;
                if (d < direct) direct = d;
            }
        }
    }
    double best = direct;
    if (has) { const double q = bn / bd; if (q < best) best = q; }   // :273 distance = d1
    if (best < __builtin_inf()) {
        if (s64) { double *s = s64 + i; if (best < *s) *s = best; }
        if (s32) { float *s = s32 + i; const float b32 = (float)best; if (b32 < *s) *s = b32; }
    }
}

// beam number tt of the intervals taken as one index space (-1: past the end)
__device__ inline int opp_iv_beam(const int *iv, int n_iv, int tt)
{
    int i = -1, rem = tt;
#pragma unroll
    for (int k = 0; k < OPP_MAX_IV; k++)
        if (k < n_iv) {
            const int len = iv[2 * k + 1] - iv[2 * k] + 1;
            if (i < 0 && rem < len) i = iv[2 * k] + rem;
            rem -= len;
        }
    return i;
}

// Four lanes per (car, opponent) pair; t = index of the lane among all pairs' lanes.
__device__ inline void opp_setup_body(const OppArgs &a, int t)
{
    const int p = t >> 2, c = t & 3;
    const int per = a.agents - 1;
    if (p >= a.n_cars * per) return; // (whole quads leave together)
    const int car = p / per, jj = p % per;
    const int env = car / a.agents, a0 = env * a.agents, self = car - a0;
    OppPair &out = a.pairs[p];
    if (a.reset_only && !a.pending_reset[env]) { if (c == 0) { out.n_iv = 0; out.total = 0; } return; }
    const int j = jj < self ? jj : jj + 1; // opponents in agent order, skipping the car itself (:574)
    const double *st = a.state + (size_t)car * 7;
    // an iTTC hit zeroes the yaw before the ray cast (base_classes.py:245); env_kernel writes the zero into the state later
    const double px = st[0], py = st[1], pyaw = a.in_collision[car] ? 0.0 : st[4];
    const double *op = a.pose_snap + (size_t)(a0 + j) * 3;
    const Params &P = a.params[(size_t)params_slot_of(a.env_params, env, a.param_slots, a.dev_err) * (a.agents + 1) + 1 + self];
    double verts[4][2];
    get_vertices(op[0], op[1], op[2], P.v[P_LENGTH], P.v[P_WIDTH], verts);
    const int cn = (c + 1) & 3;
    double cx, cy, nx, ny; // this lane's corner and the next one (the edge c -> c + 1)
    vsel(verts, c, cx, cy);
    vsel(verts, cn, nx, ny);
    // laser_models.py:283-315
    double ex, ey;
    sincos(pyaw, &ey, &ex);
    const double ego_ang = atan2(ey, ex);
    const double vx = cx - px, vy = cy - py;
    const double norm = sqrt(vx * vx + vy * vy);
    const double ux = vx / norm, uy = vy / norm;
    double angle = ego_ang - atan2(uy, ux);
    if (angle > F110_PI) angle = angle - 2 * F110_PI;
    else if (angle < -F110_PI) angle = angle + 2 * F110_PI;
    const double bang = -angle; // the corner's direction in the scan's frame, [-pi, pi]
    const int ind = argmin_abs_diff_sorted(a.scan_angles, a.nb, bang);
    int lo = ind, hi = ind;
    double bmin = bang, bmax = bang;
    double neg_max = bang < 0.0 ? bang : -__builtin_inf(), pos_min = bang >= 0.0 ? bang : __builtin_inf();
    bool nan_any = !(bang == bang);
#pragma unroll
    for (int off = 1; off <= 2; off <<= 1) {
        const int o_lo = __shfl_xor(lo, off), o_hi = __shfl_xor(hi, off);
        lo = o_lo < lo ? o_lo : lo;
        hi = o_hi > hi ? o_hi : hi;
        const double t0 = __shfl_xor(bmin, off), t1 = __shfl_xor(bmax, off), t2 = __shfl_xor(neg_max, off), t3 = __shfl_xor(pos_min, off);
        bmin = t0 < bmin ? t0 : bmin; bmax = t1 > bmax ? t1 : bmax;
        neg_max = t2 > neg_max ? t2 : neg_max; pos_min = t3 < pos_min ? t3 : pos_min;
        nan_any = nan_any || (__shfl_xor((int)nan_any, off) != 0);
    }
    const bool valid = !(lo > a.nb - 1 || hi > a.nb - 1 || nan_any); // (invalid only with NaN inputs; quad-uniform)
    // this lane's edge: what get_range computes from the pose and the two corners alone; front-facing edges first
    int n_front;
    {
        const double v1x = px - cx, v1y = py - cy;     // laser_models.py:258
        const double v2x = nx - cx, v2y = ny - cy;     // :259
        const double cr = v2x * v1y - v2y * v1x;       // cross(v2, v1) :267, :220-230
        // The car is on the outer side of edge c iff cross(v2, v1) has the sign opposite to the quad's orientation (twice
        // its signed area, from the diagonals).  cross == 0 (the car on the edge's line) and NaNs count as front-facing.
        const double orient = (verts[2][0] - verts[0][0]) * (verts[3][1] - verts[1][1]) - (verts[2][1] - verts[0][1]) * (verts[3][0] - verts[1][0]);
        const bool back = (orient > 0.0 && cr > 0.0) || (orient < 0.0 && cr < 0.0);
        const int q0 = (threadIdx.x & 63) & ~3;
        int before_f = 0, before_b = 0, nf = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int be = __shfl((int)back, q0 + e);
            nf += be ? 0 : 1;
            if (e < c) { before_f += be ? 0 : 1; before_b += be ? 1 : 0; }
        }
        n_front = nf == 0 ? 4 : nf; // (no edge faces a car inside the quad: all take the full test, in their own order)
        const int pos = nf == 0 ? c : (back ? nf + before_b : before_f);
        out.r.v1x[pos] = v1x; out.r.v1y[pos] = v1y; out.r.v2x[pos] = v2x; out.r.v2y[pos] = v2y;
        out.r.cr[pos] = cr;
        out.r.ax[pos] = cx; out.r.ay[pos] = cy; out.r.bx[pos] = nx; out.r.by[pos] = ny;
    }
    if (c != 0) return;
    out.r.n_front = n_front; out.r.pad = 0;
    // cos / sin(yaw + pi/2) = (-sin, cos)(yaw): within an ulp of the reference's cos(fl(fl(yaw + angle) + pi/2)) route, like
    // the angle-addition form it feeds (f110_device.h, ray_cast_wave)
    out.r.px = px; out.r.py = py; out.r.cA = -ey; out.r.sA = ex;
    const double mx = 0.5 * (verts[0][0] + verts[2][0]), my = 0.5 * (verts[0][1] + verts[2][1]);
    const double ddx = verts[0][0] - verts[2][0], ddy = verts[0][1] - verts[2][1];
    const double qx = mx - px, qy = my - py;
    double reach = 0.5 * sqrt(ddx * ddx + ddy * ddy) * 1.000001 + 1e-9;
    if (!(reach == reach)) reach = __builtin_inf(); // NaN poses: no pre-test
    out.r.qx = qx; out.r.qy = qy; out.r.reach = reach;
    int n_iv = 0, iv[2 * OPP_MAX_IV] = {0, -1, 0, -1, 0, -1};
    if (valid) {
        const double qn = sqrt(qx * qx + qy * qy);
        const double sa0 = a.scan_angles[0];
        const double incr = (a.scan_angles[a.nb - 1] - sa0) / (double)(a.nb - 1);
        // the car inside the opponent's bounding circle (the hull may be anything), NaNs, a degenerate beam table: every beam
        const bool all = !(qn > reach * 1.000001) || !(reach < __builtin_inf()) || !(incr > 0.0);
        if (all || !(bmax - bmin > F110_PI)) {
            // the hull is the arc from the lowest to the highest corner angle: [lo, hi] itself (hull + pi lies outside it)
            iv[0] = lo; iv[1] = hi; n_iv = 1;
        } else {
            // the corners straddle +-pi: the hull is [-pi, neg_max] + [pos_min, pi], hull + pi the arc around 0 between them
            const double inv = 1.0 / incr;
            const double cand[3][2] = {{(double)lo, (neg_max - sa0) * inv + 2.0},
                                       {(pos_min - F110_PI - sa0) * inv - 2.0, (neg_max + F110_PI - sa0) * inv + 2.0},
                                       {(pos_min - sa0) * inv - 2.0, (double)hi}};
            int end = lo - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                double b0 = cand[k][0], b1 = cand[k][1];
                if (!(b0 == b0) || !(b1 == b1)) { b0 = (double)lo; b1 = (double)hi; }
                int i0 = b0 > (double)(end + 1) ? (int)floor(b0) : end + 1;
                const int i1 = b1 < (double)hi ? (int)ceil(b1) : hi;
                if (i0 < lo) i0 = lo;
                if (i0 > i1) continue;
                iv[2 * n_iv] = i0; iv[2 * n_iv + 1] = i1; n_iv++;
                end = i1;
            }
        }
    }
    int total = 0;
#pragma unroll
    for (int k = 0; k < OPP_MAX_IV; k++) {
        out.iv[2 * k] = iv[2 * k]; out.iv[2 * k + 1] = iv[2 * k + 1];
        if (k < n_iv) total += iv[2 * k + 1] - iv[2 * k] + 1;
    }
    out.n_iv = n_iv; out.total = total;
}

#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(128) void opp_setup_kernel(OppArgs a) { opp_setup_body(a, blockIdx.x * blockDim.x + threadIdx.x); }
#endif

#ifndef F110_OPP_GROUP
#define F110_OPP_GROUP 64
#endif
constexpr int OPP_GROUP = F110_OPP_GROUP; // lanes per car in opp_apply_kernel (measured at 32 768 cars: 8 lanes 52 us, 16: 33, 32: 27, 64: 23)
constexpr int OPP_GROUP_MAX = 256; // a pair with more beams than this is walked by the whole wave, not by its group

#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(256, 8) void opp_apply_kernel(OppArgs a)
{
    __shared__ OppPair s_pair[256 / OPP_GROUP]; // one per group
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = t & (OPP_GROUP - 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = threadIdx.x / OPP_GROUP;
    // (no lane leaves early: long lists are served by all 64 lanes of the wave)
    const bool in_range = (t / OPP_GROUP) < a.n_cars;
    const int car = in_range ? t / OPP_GROUP : a.n_cars - 1;
    const int per = a.agents - 1;
    float *s32 = a.scans32 ? a.scans32 + (size_t)car * a.nb : nullptr;
    double *s64 = a.scans64 ? a.scans64 + (size_t)car * a.nb : nullptr;
    constexpr int WORDS = (int)(sizeof(OppPair) / 4);
    for (int jj = 0; jj < per; jj++) { // the opponents of a car one after the other: each is an in-place minimum on the same scan
        // the group's pair into LDS (its lanes read consecutive words: one or two lines per group)
        {
            const unsigned *src = reinterpret_cast<const unsigned *>(a.pairs + (size_t)car * per + jj);
            unsigned *dst = reinterpret_cast<unsigned *>(&s_pair[grp]);
            for (int w = sub; w < WORDS; w += OPP_GROUP) dst[w] = src[w];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const OppPair *pr = &s_pair[grp];
        const int total = in_range ? pr->total : 0;
        const bool longlist = total > OPP_GROUP_MAX; // (group-uniform)
        if (total > 0 && !longlist) {
            const int n_iv = pr->n_iv;
            for (int tt = sub; tt < total; tt += OPP_GROUP) {
                const int i = opp_iv_beam(pr->iv, n_iv, tt);
                F110_BCHK(i >= 0 && i < a.nb, BT_OPP_BEAM, a.dev_err);
                if (i >= 0 && i < a.nb) opp_test_beam(&pr->r, a.beam_cs, i, s32, s64);
            }
        }
        unsigned long long todo = __builtin_amdgcn_ballot_w64(longlist && sub == 0);
        while (todo) {
            const int src_lane = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int wcar = __shfl(car, src_lane);
            const OppPair *wp = &s_pair[(wave * WAVE + src_lane) / OPP_GROUP]; // that group's copy, in this wave's part of the array
            const int wn = wp->n_iv, wtotal = wp->total;
            float *w32 = a.scans32 ? a.scans32 + (size_t)wcar * a.nb : nullptr;
            double *w64 = a.scans64 ? a.scans64 + (size_t)wcar * a.nb : nullptr;
            for (int tt = lane; tt < wtotal; tt += WAVE) {
                const int i = opp_iv_beam(wp->iv, wn, tt);
                if (i >= 0 && i < a.nb) opp_test_beam(&wp->r, a.beam_cs, i, w32, w64);
            }
        }
        __builtin_amdgcn_wave_barrier(); // the LDS copies are overwritten by the next opponent's
    }
}
#endif

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
    uint8_t *was_pending; // [B] or NULL: pending_reset as this step found it (env_kernel clears / re-arms the flag itself)
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
    int param_slots;            // slots `params` holds (read by the bounds-checked build only)
    uint32_t *dev_err;          // device error word
    const NoiseDesc *noise;     // the rows the noise table holds (or NULL): the scan behind this kernel reads row noise_step[car] unchecked
};

// The single-track model switches to its kinematic form below 0.5 m/s (dynamic_models.py:152): a wavefront that holds
// one slow car among 63 fast ones executes BOTH forms at every RK4 stage (the slow form is a third of the instructions
// of a step, and with autoreset a few per cent of the cars are always just leaving their spawn pose -- enough to put a
// slow car into most wavefronts).  The block therefore deals its cars out so that the slow ones (and the idle lanes)
// share the LAST wavefronts: lane l works on car s_perm[l], the others' waves skip the kinematic code altogether.
// Which lane integrates a car does not change a bit of its result.
#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(256) void dynamics_kernel(DynArgs a)
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
    if (a.was_pending) a.was_pending[env] = pend ? 1 : 0; // (every car of the env stores the same byte)
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
    // (the noise window's bounds and the car's row travel with the loads above: read after the stores below they were one more
    // memory round trip at the end of a kernel that is nothing but latency)
    int nrow = 0, nlo = 0, nhi = 0x7fffffff;
    if (a.noise && a.noise_step) { nrow = pend ? 0 : a.noise_step[car]; nlo = a.noise->lo; nhi = a.noise->hi; }
    const Params P = a.params[(size_t)params_slot_of(a.env_params, env, a.param_slots, a.dev_err) * (a.agents + 1) + 1 + car % a.agents];
    update_pose(st, sb, sc, steer, speed, P, a.time_step, a.integrator);
#pragma unroll
    for (int i = 0; i < 7; i++) a.state[(size_t)car * 7 + i] = st[i];
    a.steer_buf[(size_t)car * 2] = sb[0];
    a.steer_buf[(size_t)car * 2 + 1] = sb[1];
    a.steer_cnt[car] = sc;
    if (a.in_collision) a.in_collision[car] = 0;
    // the host keeps the noise table ahead of every car (Engine._ensure_noise); a row outside it is reported, never silent
    if (__builtin_expect(nrow < nlo || nrow >= nhi, 0))
        if (a.dev_err) atomicOr(a.dev_err, DEVERR_NOISE_WINDOW);
    if (a.pose_snap) {
        a.pose_snap[(size_t)car * 3] = st[0];
        a.pose_snap[(size_t)car * 3 + 1] = st[1];
        a.pose_snap[(size_t)car * 3 + 2] = st[4];
    }
}
#endif

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
    int param_slots;          // (bounds-checked build only)
    uint32_t *dev_err;
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

#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(128) void check_done_kernel(CheckDoneArgs a)
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
#endif

// ONE: the env has one agent (no pair to test: the GJK code is not even compiled in, the kernel is a third of the size)
template <bool ONE>
__device__ inline void env_body(const EnvArgs &a, int env)
{
    if (env >= a.n_envs) return;
    const bool pend = a.pending_reset[env] != 0;
    if (a.reset_only && !pend) return;
    const int A = ONE ? 1 : a.agents, c0 = env * A;
    if (ONE) {
        a.collisions[c0] = 0; a.collision_idx[c0] = -1; // collision_multiple (collision_models.py:185-212) on one quad
    } else {
        // Simulator.check_collision (base_classes.py:529-543) on the post-integration poses
        const Params &SP = a.params[(size_t)params_slot_of(a.env_params, env, a.param_slots, a.dev_err) * (A + 1)]; // Simulator.params (:542)
        collision_multiple_dev(a.pose_snap + (size_t)c0 * 3, A, SP.v[P_LENGTH], SP.v[P_WIDTH],
                               a.collisions + c0, a.collision_idx + c0);
    }
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
        double sth, cth;
        sincos(th, &sth, &cth);
        r00 = cth; r01 = -sth; r10 = sth; r11 = cth;
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

template <bool ONE>
__global__ __launch_bounds__(128) void env_kernel(EnvArgs a) { env_body<ONE>(a, blockIdx.x * blockDim.x + threadIdx.x); }

// A > 1: the env bookkeeping and the opponents' set-up in ONE launch.  Both are small kernels whose time is latency (256 and
// 2 048 waves), and neither reads what the other writes -- except that env_body zeroes the yaw of a car whose iTTC fired, for
// which the set-up uses 0 anyway, and clears pending_reset, of which the set-up reads dynamics_kernel's snapshot
// (OppArgs::pending_reset = was_pending) -- so the first env_blocks workgroups do one and the rest the other, side by side.
struct PostScanArgs {
    EnvArgs e;
    OppArgs o;
    int env_blocks;
};

#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(128) void post_scan_kernel(PostScanArgs a)
{
    if ((int)blockIdx.x < a.env_blocks) env_body<false>(a.e, blockIdx.x * blockDim.x + threadIdx.x);
    else opp_setup_body(a.o, (blockIdx.x - a.env_blocks) * blockDim.x + threadIdx.x);
}
#endif

// ------------------------------------------------------------------ one env's observation in one buffer
// The single-env facade (red_gym_amd.F110Env = the reference's Gym API on a batch of one) returns NumPy / Python objects
// every step: instead of a device -> host copy per field, one kernel gathers env `env` into one fp64 row
//   [A*7 state | A collisions | A lap_times | A lap_counts | A toggles | current_time | done | A*nb scans]
// (every small field is exactly representable in fp64) and ONE copy takes it to the host.
struct PackArgs {
    int env, agents, nb;
    const double *state; const uint8_t *collisions; const double *lap_times; const int32_t *lap_counts; const int32_t *toggles;
    const double *current_time; const uint8_t *done; const double *scans64; const float *scans32;
    double *out;
};

#if defined(F110_UNIT_STEP)
static __global__ __launch_bounds__(256) void pack_env_kernel(PackArgs a)
{
    const int A = a.agents, c0 = a.env * A;
    const int n_small = 11 * A + 2, n = n_small + A * a.nb;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double v;
        if (i < 7 * A) v = a.state[(size_t)c0 * 7 + i];
        else if (i < 8 * A) v = (double)a.collisions[c0 + i - 7 * A];
        else if (i < 9 * A) v = a.lap_times[c0 + i - 8 * A];
        else if (i < 10 * A) v = (double)a.lap_counts[c0 + i - 9 * A];
        else if (i < 11 * A) v = (double)a.toggles[c0 + i - 10 * A];
        else if (i == 11 * A) v = a.current_time[a.env];
        else if (i == 11 * A + 1) v = (double)a.done[a.env];
        else {
            const size_t k = (size_t)c0 * a.nb + (size_t)(i - n_small);
            v = a.scans64 ? a.scans64[k] : (double)a.scans32[k];
        }
        a.out[i] = v;
    }
}
#endif

// ------------------------------------------------------------------ function-level kernels
// dynamic_models.py:91-121 / :124-176 right-hand sides (the reference's KAT surface)
#if defined(F110_UNIT_STEP)
static __global__ void rhs_kernel(const double *x, const double *u, int n, int kinematic, const Params *params, double *f)
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
#endif

#if defined(F110_UNIT_STEP)
static __global__ void vertices_kernel(const double *poses, int n, double L, double W, double *out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[4][2];
    get_vertices(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2], L, W, v);
    for (int k = 0; k < 4; k++) { out[(size_t)i * 8 + 2 * k] = v[k][0]; out[(size_t)i * 8 + 2 * k + 1] = v[k][1]; }
}
#endif

#if defined(F110_UNIT_STEP)
static __global__ void gjk_pairs_kernel(const double *va, const double *vb, int n, uint8_t *hit)
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
#endif

#if defined(F110_UNIT_STEP)
static __global__ void collision_multiple_kernel(const double *verts, int n, int A, uint8_t *col, int32_t *cidx)
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
#endif

// check_ttc_jit (laser_models.py:189-217): wave per scan
#if defined(F110_UNIT_STEP)
static __global__ void ttc_kernel(const double *scans, const double *vel, int n, int nb, const double *beam_cosines,
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
#endif

// ray_cast (laser_models.py:319-346): wave per (ego, opponent quad)
#if defined(F110_UNIT_STEP)
static __global__ void ray_cast_kernel(const double *ego, const double *verts, int n, int nb, const double *scan_angles,
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
#endif

} // namespace f110
