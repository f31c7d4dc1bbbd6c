// f110_noise.h -- the lidar noise of the step path, generated on the GPU.
//
// Reference: every scan of every car adds `rng.normal(0., 0.01, size=num_beams)` (laser_models.py:450-452) drawn from a
// per-car `np.random.default_rng(seed)` that is re-created at every reset (base_classes.py:117,202).  All cars of an env
// share the seed, so the row a car adds depends only on (seed, scans since its reset): the rows of one seed are kept
// ONCE per seed ("noise slot") in a device table indexed by each car's own counter, and produced here, on the device,
// by a bit-level restatement of what NumPy executes for that call:
//   * PCG64 (numpy/random/src/pcg64/pcg64.h, pcg_setseq_128_xsl_rr_64): 128-bit LCG step, XSL-RR output of the new state;
//   * random_standard_normal (numpy/random/src/distributions/distributions.c): 256-layer ziggurat -- 99.3 % of the draws
//     are one table compare; the wedge test uses exp(), the tail (|x| > 3.654, 0.026 % of the draws) log1p();
//   * random_normal: loc + scale * x.
// The stream is sequential (a draw consumes one raw value, or more after a rejection), so ONE WAVEFRONT per seed walks it,
// 64 raw values at a time (noise_rows_kernel below; lane j holds the generator state of raw position p + j by LCG
// jump-ahead).  Exactness: every accepted value outside the tail is the product of an integer and a table entry (no library
// call); the wedge test compares against exp() (a decision, and the only one a library rounding could flip: none has been
// seen in 1e8 draws); the tail's log1p is glibc's own sequence of roundings (log1p_glibc below), so the rows are NumPy's bits.
#pragma once
#include "f110_device.h"
#include "f110_ziggurat.h"

#pragma clang fp contract(off)

namespace f110 {

typedef unsigned __int128 u128;

// What the scan kernel needs to find a car's noise row.  Lives in device memory at an address that never changes for
// the handle's life (the kernel's by-value arguments -- and therefore captured hipGraphs -- survive every growth).
// Row r of slot s: base[(s * cap + (r & mask)) * num_beams + beam] = the beam's noise; rows lo <= r < hi are present (a ring once
// lo > 0); noise off: cap = 1, one row of zeros.  (Rounds 3-4 kept {noise, side distance} pairs here so that one gather fed the
// iTTC test as well; once the cars of a batch stand on different rows the rows stream from L2 / HBM, and the 8 redundant bytes per
// beam cost 4.6 % of a 65 536-car launch -- the side distance is now read only for iTTC candidates, profiles/r04_scan_stores.txt M.)
struct NoiseDesc {
    const double *base;
    int mask, cap, lo, hi; // (a car's row counter is an int32: f110_buffers.noise_step)
    int slots, pad;        // slots the table holds (read by the bounds-checked build only)
};

// generator of one slot: t = the LCG state whose output is the NEXT raw value of the stream
struct NoiseGen {
    unsigned long long t_lo, t_hi, inc_lo, inc_hi;
    double std;
    long long rows;   // rows produced so far (the stream stands at the start of row `rows`)
    int on, pad;
};

__device__ inline u128 pcg_mult() { return ((u128)0x2360ED051FC65DA4ull << 64) | (u128)0x4385DF649FCCF645ull; } // PCG_DEFAULT_MULTIPLIER_128

__device__ inline unsigned long long pcg_out(u128 s) // pcg_output_xsl_rr_128_64
{
    const unsigned long long hi = (unsigned long long)(s >> 64), lo = (unsigned long long)s;
    const unsigned long long x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((0u - rot) & 63u));
}

__device__ inline double pcg_double(unsigned long long raw) { return (double)(raw >> 11) * (1.0 / 9007199254740992.0); }

__device__ inline u128 shfl128(u128 v, int src)
{
    unsigned w0 = (unsigned)v, w1 = (unsigned)(v >> 32), w2 = (unsigned)(v >> 64), w3 = (unsigned)(v >> 96);
    w0 = (unsigned)__shfl((int)w0, src); w1 = (unsigned)__shfl((int)w1, src);
    w2 = (unsigned)__shfl((int)w2, src); w3 = (unsigned)__shfl((int)w3, src);
    return ((u128)w3 << 96) | ((u128)w2 << 64) | ((u128)w1 << 32) | (u128)w0;
}

__device__ inline unsigned long long shfl64(unsigned long long v, int src)
{
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}

// the stream of a slot at the start of row 64 * k (its LCG state): written when the row is first reached, so that rows which
// were dropped from the ring can be produced again from the nearest mark -- in parallel, one wavefront per 64 rows -- instead of
// from the seed
struct NoiseMark { unsigned long long t_lo, t_hi; };
constexpr int NOISE_MARK_ROWS = 64;

// log1p as glibc computes it (sysdeps/ieee754/dbl-64/s_log1p.c, 2.35: the fdlibm algorithm -- argument reduction to
// 1 + f in [sqrt(2)/2, sqrt(2)), the degree-7 polynomial in z = (f / (2 + f))^2 regrouped as R1 + z2*R2 + z4*R3 + z6*R4 --
// built for x86-64 without contraction), restated operation by operation: NumPy's ziggurat calls the C library's log1p for
// its tail draws (0.026 % of the draws), and fdlibm's result is within an ulp of the true value but not the correctly rounded
// one, so "NumPy's bits" means this sequence of roundings.  tools/log1p_model.py is the same text in Python, compared with
// the local libm bit for bit over 5e5 arguments (and the libm's constants read from its binary); the device math library's
// log1p differed in 4 of 2 156 tail draws (round 4).  Domain here: x = -u, u in [0, 1).
__device__ inline double log1p_glibc(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
                 Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
                 Lp7 = 1.479819860511658591e-01;
    const int hx = __double2hiint(x);
    const int ax = hx & 0x7fffffff;
    int k = 1, hu = 1;
    double f = 0.0, c = 0.0;
    if (hx < 0x3FDA827A) {                                  // x < 0.41422
        if (ax >= 0x3ff00000) return x == -1.0 ? -__builtin_inf() : __builtin_nan(""); // x <= -1 (not reached: u < 1)
        if (ax < 0x3e200000) {                              // |x| < 2^-29
            if (ax < 0x3c900000) return x;                  // |x| < 2^-54
            return x - x * x * 0.5;
        }
        if (hx > 0 || hx <= (int)0xbfd2bec3) { k = 0; f = x; hu = 1; } // -0.2929 < x < 0.41422
    } else if (hx >= 0x7ff00000) return x + x;
    if (k != 0) {
        double u;
        if (hx < 0x43400000) {
            u = 1.0 + x;
            hu = __double2hiint(u);
            k = (hu >> 20) - 1023;
            c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);    // correction term
            c = c / u;
        } else {
            u = x;
            hu = __double2hiint(u);
            k = (hu >> 20) - 1023;
            c = 0.0;
        }
        hu &= 0x000fffff;
        if (hu < 0x6a09e) u = __hiloint2double(hu | 0x3ff00000, __double2loint(u));           // normalize u
        else { k += 1; u = __hiloint2double(hu | 0x3fe00000, __double2loint(u)); hu = (0x00100000 - hu) >> 2; } // normalize u / 2
        f = u - 1.0;
    }
    const double hfsq = (0.5 * f) * f;
    if (hu == 0) {                                          // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            c = c + (double)k * ln2_lo;
            return (double)k * ln2_hi + c;
        }
        const double R = hfsq * (1.0 - 0.66666666666666666 * f);
        if (k == 0) return f - R;
        return (double)k * ln2_hi - ((R - ((double)k * ln2_lo + c)) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double R1 = z * Lp1, z2 = z * z;
    const double R2 = Lp2 + z * Lp3, z4 = z2 * z2;
    const double R3 = Lp4 + z * Lp5, z6 = z4 * z2;
    const double R4 = Lp6 + z * Lp7;
    const double R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return (double)k * ln2_hi - ((hfsq - (s * (hfsq + R) + ((double)k * ln2_lo + c))) - f);
}

struct NoiseGenArgs {
    NoiseGen *gen;        // [slots]
    double *base;         // the table being filled
    long long mask, cap;
    long long lo;         // rows below lo are generated (the stream must advance) but not stored
    long long r1;         // every active slot is brought to r1 rows (a multiple of NOISE_MARK_ROWS when marks are kept)
    int nb;
    NoiseMark *marks;     // [slots][marks_cap] or NULL
    long long marks_cap;
    // re-production of dropped rows (blockIdx.y = chunk): rows [max(lo, 64 * (chunk0 + y)), min(r1, 64 * (chunk0 + y + 1))) from
    // marks[slot][chunk0 + y]; the generators' own states are neither read nor written
    int redo;
    long long chunk0;
    // powers and partial sums of the LCG multiplier: pcg_tab[j] = M^j, pcg_tab[65 + j] = 1 + M + ... + M^(j-1), j = 0 .. 64
    // (f110_noise_abi.hip computes them once): lane j's start state and the 64-step jump are two multiply-adds instead of loops
    const u128 *pcg_tab;
    // PER-ENV mode (f110_set_noise_per_env: every env its own seed, no limit on their number): slot = env, the table holds ONE
    // row per env (cap = 1), and every step produces the row the env's scan is about to add -- row `pend ? 0 : env_row[slot *
    // env_row_stride]` of the env's stream -- from the state the previous step left (or from the seed after a reset; or, after
    // a checkpoint was loaded, by running the stream forward from the seed without storing).  blockDim = 64 * waves, one
    // wavefront per env.
    const int32_t *env_row;
    int env_row_stride, n_env, reset_only;
    const uint8_t *env_pending;
    const NoiseGen *seeds;
};

// One wavefront per noise slot (grid = slots).
//
// The stream is walked in WINDOWS of 64 raw values, one per lane, and a window is always consumed whole: every lane treats
// its raw value as a candidate (99.3 % are accepted by one table compare); a rejected candidate resolves itself
// SPECULATIVELY, stepping a private copy of its own generator state through the raw values it would consume if it really
// were a candidate (wedge: one; tail: two per trial) -- no lane needs another lane's value.  Which lanes ARE candidates is
// then settled in stream order with a few scalar operations: a candidate's extra raws are not candidates (they are skipped,
// into the next window if need be: `skip`), everything else is.  The accepted candidates are numbered by a prefix count and
// stored as consecutive beams.  The window then advances by exactly 64 positions (one 128-bit multiply-add per lane), so
// there is no re-basing shuffle and the only state carried from window to window is (skip, beams produced).
#if defined(F110_UNIT_NOISE)
static __global__ __launch_bounds__(256) void noise_rows_kernel(NoiseGenArgs a)
{
    __shared__ unsigned long long s_ki[256];
    __shared__ double s_wi[256], s_fi[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) { s_ki[i] = ZIG_KI[i]; s_wi[i] = ZIG_WI[i]; s_fi[i] = ZIG_FI[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int slot = a.env_row ? blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) : blockIdx.x;
    if (a.env_row && slot >= a.n_env) return;
    NoiseGen g = a.gen[slot];
    if (a.env_row) {
        const bool pend = a.env_pending && a.env_pending[slot];
        if (a.reset_only && !pend) return;
        const long long r = pend ? 0 : (long long)a.env_row[(size_t)slot * (size_t)a.env_row_stride];
        if (r <= 0 || g.rows > r) { const NoiseGen sd = a.seeds[slot]; g.t_lo = sd.t_lo; g.t_hi = sd.t_hi; g.rows = 0; }
        a.lo = r < 0 ? 0 : r; a.r1 = a.lo + 1;
    }
    if (a.redo) {
        const long long ch = a.chunk0 + blockIdx.y;
        if (!g.on || !a.marks || ch >= a.marks_cap) return;
        const NoiseMark mk = a.marks[(size_t)slot * (size_t)a.marks_cap + (size_t)ch];
        g.t_lo = mk.t_lo; g.t_hi = mk.t_hi; g.rows = ch * NOISE_MARK_ROWS;
        a.r1 = a.r1 < g.rows + NOISE_MARK_ROWS ? a.r1 : g.rows + NOISE_MARK_ROWS;
    }
    if (!g.on || g.rows >= a.r1) return; // (wave-uniform)
    // the mark of the row this launch starts at (a launch ends where the next one starts: every multiple of 64 rows gets one)
    if (!a.redo && a.marks && lane == 0 && g.rows % NOISE_MARK_ROWS == 0 && g.rows / NOISE_MARK_ROWS < a.marks_cap) {
        NoiseMark mk; mk.t_lo = g.t_lo; mk.t_hi = g.t_hi;
        a.marks[(size_t)slot * (size_t)a.marks_cap + (size_t)(g.rows / NOISE_MARK_ROWS)] = mk;
    }
    const u128 M = pcg_mult(), inc = ((u128)g.inc_hi << 64) | (u128)g.inc_lo;
    // 64 steps at once: s -> A * s + C, A = M^64, C = (1 + M + ... + M^63) * inc
    const u128 A = a.pcg_tab[64], C = a.pcg_tab[65 + 64] * inc;
    u128 T = ((u128)g.t_hi << 64) | (u128)g.t_lo;
    T = a.pcg_tab[lane] * T + a.pcg_tab[65 + lane] * inc; // lane j: the state whose output is raw value p + j (j LCG steps ahead)
    const double std = g.std;
    const int nb = a.nb;
    const unsigned long long below = (1ull << lane) - 1ull;
    long long row = g.rows;
    int o = 0;     // beams of `row` produced so far
    int skip = 0;  // leading raw values of this window that belong to a candidate of an earlier window
    for (;;) {
        // distributions.c random_standard_normal: r = next_uint64; idx = r & 0xff; r >>= 8; sign = r & 1;
        // rabs = (r >> 1) & 0x000fffffffffffff; x = rabs * wi[idx]; if (sign) x = -x; if (rabs < ki[idx]) return x;
        unsigned long long r = pcg_out(T);
        const int idx = (int)(r & 0xffull);
        r >>= 8;
        const bool neg = (r & 1ull) != 0;
        const unsigned long long rabs = (r >> 1) & 0x000fffffffffffffull;
        double val = (double)rabs * s_wi[idx];
        if (neg) val = -val;
        const bool fast = rabs < s_ki[idx];
        int extras = 0;   // raw values this lane consumes beyond its own IF it is a candidate
        bool emits = true; // ... and whether it then yields a value (a rejected wedge draw does not: the draw starts over)
        const unsigned long long slow = __builtin_amdgcn_ballot_w64(!fast);
        if (slow != 0ull) {
            if (!fast) {
                u128 Q = T;
                if (idx == 0) {
                    // tail: xx = -inv_r * log1p(-U), yy = -log1p(-U) until yy + yy > xx * xx
                    for (;;) {
                        Q = Q * M + inc; const double u1 = pcg_double(pcg_out(Q));
                        Q = Q * M + inc; const double u2 = pcg_double(pcg_out(Q));
                        extras += 2;
                        const double xx = -ZIG_NOR_INV_R * log1p_glibc(-u1);
                        const double yy = -log1p_glibc(-u2);
                        if (yy + yy > xx * xx) { val = ((rabs >> 8) & 1ull) ? -(ZIG_NOR_R + xx) : ZIG_NOR_R + xx; break; }
                    }
                } else {
                    // wedge: ((fi[idx-1] - fi[idx]) * U + fi[idx]) < exp(-0.5 * x * x) ? return x : draw again
                    Q = Q * M + inc; const double u = pcg_double(pcg_out(Q));
                    extras = 1;
                    emits = ((s_fi[idx - 1] - s_fi[idx]) * u + s_fi[idx]) < exp(-0.5 * val * val);
                }
            }
        }
        // ---- which lanes are candidates: in stream order, a candidate's extras are not
        unsigned long long skipped = skip >= 64 ? ~0ull : ((1ull << skip) - 1ull);
        int carry = skip > 64 ? skip - 64 : 0;
        unsigned long long m = slow & ~skipped;
        while (m) {
            const int l = (int)__builtin_ctzll(m);
            m &= m - 1ull;
            const int e = __builtin_amdgcn_readlane(extras, l);
            int end = l + e;
            if (end > 63) { carry = carry > end - 63 ? carry : end - 63; end = 63; }
            if (end > l) {
                const unsigned long long hi = end == 63 ? ~0ull : ((1ull << (end + 1)) - 1ull);
                const unsigned long long range = hi & ~((2ull << l) - 1ull); // positions l+1 .. end
                skipped |= range;
                m &= ~range;
            }
        }
        const unsigned long long emitm = ~skipped & __builtin_amdgcn_ballot_w64(emits);
        const int n_emit = __popcll(emitm);
        const bool mine = (emitm >> lane) & 1ull;
        const int rank = __popcll(emitm & below);
        const long long left = (a.r1 - row) * (long long)nb - (long long)o; // beams this launch still has to produce
        const bool last = (long long)n_emit >= left;
        // ---- store: beam number o + rank of row `row`, running on into the next rows
        if (mine && (!last || (long long)rank < left)) {
            int gb = o + rank;
            long long rw = row;
            while (gb >= nb) { gb -= nb; rw++; }
            if (rw >= a.lo)
                a.base[((size_t)slot * (size_t)a.cap + (size_t)(rw & a.mask)) * (size_t)nb + gb] = 0.0 + std * val; // random_normal: loc + scale * x
        }
        if (last) {
            // the launch ends inside this window: the stream stands behind the candidate that produced the last beam
            unsigned long long mm = emitm;
            for (long long i = 1; i < left; i++) mm &= mm - 1ull;
            const int L = (int)__builtin_ctzll(mm);
            const int q = L + 1 + __builtin_amdgcn_readlane(extras, L); // (extras is 0 for a fast candidate)
            u128 Tq = shfl128(T, q < 63 ? q : 63);
            for (int i = 63; i < q; i++) Tq = Tq * M + inc;
            if (lane == 0 && !a.redo) {
                a.gen[slot].t_lo = (unsigned long long)Tq;
                a.gen[slot].t_hi = (unsigned long long)(Tq >> 64);
                a.gen[slot].rows = a.r1;
            }
            return;
        }
        o += n_emit;
        while (o >= nb) { o -= nb; row++; }
        skip = carry;
        T = A * T + C;
    }
}
#endif

// host-fed slot: plain fp64 rows [T, nb] (device staging copy; NULL: zeros) -> rows 0 .. T-1 of the slot's ring
#if defined(F110_UNIT_NOISE)
static __global__ void noise_fill_kernel(const double *rows, long long T, int nb, double *base, int slot, long long cap, long long mask)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * nb) return;
    const long long r = i / nb;
    const int b = (int)(i - r * nb);
    base[((size_t)slot * (size_t)cap + (size_t)(r & mask)) * (size_t)nb + b] = rows ? rows[i] : 0.0;
}
#endif

// growth: rows lo .. hi-1 of every slot move to their places in a larger ring
#if defined(F110_UNIT_NOISE)
static __global__ void noise_move_kernel(const double *src, long long scap, long long smask, double *dst, long long dcap,
                                  long long dmask, int slots, long long lo, long long hi, int nb)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (hi - lo) * nb;
    if (i >= per * slots) return;
    const int s = (int)(i / per);
    const long long k = i - (long long)s * per;
    const long long r = lo + k / nb;
    const int b = (int)(k % nb);
    dst[((size_t)s * (size_t)dcap + (size_t)(r & dmask)) * (size_t)nb + b] = src[((size_t)s * (size_t)scap + (size_t)(r & smask)) * (size_t)nb + b];
}
#endif

} // namespace f110
