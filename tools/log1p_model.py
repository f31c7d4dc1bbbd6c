"""glibc 2.35's log1p for fp64 (sysdeps/ieee754/dbl-64/s_log1p.c: the fdlibm algorithm with the polynomial regrouped as
R1 + z2*R2 + z4*R3 + z6*R4; x86-64 build: SSE2, no contraction -- checked against the disassembly of the local libm and its
constants), restated operation by operation in NumPy fp64 and compared bit for bit with math.log1p.  It is the CPU model of
`log1p_glibc` in red_gym_amd/csrc/f110_noise.h (the ziggurat's tail draws: -log1p(-u), u in [0, 1)).
    python tools/log1p_model.py [n]"""
import math
import struct
import sys

import numpy as np

LN2_HI = float.fromhex('0x1.62e42fee00000p-1')
LN2_LO = float.fromhex('0x1.a39ef35793c76p-33')
LP = [None] + [struct.unpack('<d', bytes.fromhex(h)[::-1])[0] for h in (
    '3fe5555555555593', '3fd999999997fa04', '3fd2492494229359', '3fcc71c51d8e78af', '3fc7466496cb03de', '3fc39a09d078c69f',
    '3fc2f112df3e5244')]


def hi_word(x):
    return struct.unpack('<q', struct.pack('<d', x))[0] >> 32   # signed high word


def set_hi(x, hi):
    lo = struct.unpack('<Q', struct.pack('<d', x))[0] & 0xffffffff
    return struct.unpack('<d', struct.pack('<Q', ((hi & 0xffffffff) << 32) | lo))[0]


def log1p_glibc(x):
    """x in (-1, +inf); every operation is one IEEE fp64 operation (Python floats), in glibc's order"""
    hx = hi_word(x)
    ax = hx & 0x7fffffff
    k, f, hu, c = 1, 0.0, 1, 0.0
    if hx < 0x3FDA827A:
        if ax >= 0x3ff00000:
            return -math.inf if x == -1.0 else math.nan
        if ax < 0x3e200000:
            if ax < 0x3c900000:
                return x
            return x - x * x * 0.5
        if hx > 0 or hx <= 0xbfd2bec3 - (1 << 32):
            k, f, hu = 0, x, 1
    elif hx >= 0x7ff00000:
        return x + x
    if k != 0:
        if hx < 0x43400000:
            u = 1.0 + x
            hu = hi_word(u)
            k = (hu >> 20) - 1023
            c = (1.0 - (u - x)) if k > 0 else (x - (u - 1.0))
            c = c / u
        else:
            u = x
            hu = hi_word(u)
            k = (hu >> 20) - 1023
            c = 0.0
        hu &= 0x000fffff
        if hu < 0x6a09e:
            u = set_hi(u, hu | 0x3ff00000)
        else:
            k += 1
            u = set_hi(u, hu | 0x3fe00000)
            hu = (0x00100000 - hu) >> 2
        f = u - 1.0
    hfsq = (0.5 * f) * f
    if hu == 0:
        if f == 0.0:
            if k == 0:
                return 0.0
            c = c + k * LN2_LO
            return k * LN2_HI + c
        R = hfsq * (1.0 - 0.66666666666666666 * f)
        if k == 0:
            return f - R
        return k * LN2_HI - ((R - (k * LN2_LO + c)) - f)
    s = f / (2.0 + f)
    z = s * s
    R1 = z * LP[1]
    z2 = z * z
    R2 = LP[2] + z * LP[3]
    z4 = z2 * z2
    R3 = LP[4] + z * LP[5]
    z6 = z4 * z2
    R4 = LP[6] + z * LP[7]
    R = ((R1 + z2 * R2) + z4 * R3) + z6 * R4
    if k == 0:
        return f - (hfsq - s * (hfsq + R))
    return k * LN2_HI - ((hfsq - (s * (hfsq + R) + (k * LN2_LO + c))) - f)


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    rng = np.random.default_rng(1)
    us = np.concatenate([rng.random(n), (rng.integers(0, 1 << 53, n // 4) * 2.0 ** -53), 1.0 - 2.0 ** -rng.integers(1, 54, 2000),
                         2.0 ** -rng.uniform(1, 60, 20000), [0.0, 2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 0.2929, 0.29289321881345254]])
    bad = 0
    for u in us:
        a, b = log1p_glibc(-float(u)), math.log1p(-float(u))
        if a != b and not (math.isnan(a) and math.isnan(b)):
            bad += 1
            if bad < 10:
                print('MISMATCH u=%r: model %r libm %r' % (u, a, b))
    print('%d arguments -u, u in [0, 1): %d differ from math.log1p' % (len(us), bad))
    sys.exit(1 if bad else 0)
