/* TEST INFRASTRUCTURE: drives oracle/f110_oracle.c under -fsanitize=address,undefined on a
 * small synthetic map (a ring corridor): scans, env reset/step with two agents, GJK,
 * opponent ray cast, lap logic.  Built and run by tests/test_oracle_sanitize.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "f110_oracle.c"

int main(void)
{
    const int H = 96, W = 128, nb = 270, td = 2000;
    const double res = 0.05;
    double *dt = malloc(sizeof(double) * H * W);
    /* brute-force EDT of a rectangular ring: free corridor between two rectangles */
    unsigned char *free_ = malloc(H * W);
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            int outer = r > 8 && r < H - 8 && c > 8 && c < W - 8;
            int inner = r > 30 && r < H - 30 && c > 30 && c < W - 30;
            free_[r * W + c] = outer && !inner;
        }
    for (int r = 0; r < H; r++)
        for (int c = 0; c < W; c++) {
            double best = 1e30;
            if (!free_[r * W + c]) best = 0;
            else
                for (int rr = 0; rr < H; rr++)
                    for (int cc = 0; cc < W; cc++)
                        if (!free_[rr * W + cc]) {
                            double d = (double)(rr - r) * (rr - r) + (double)(cc - c) * (cc - c);
                            if (d < best) best = d;
                        }
            dt[r * W + c] = res * sqrt(best);
        }
    double *sines = malloc(sizeof(double) * td), *cosines = malloc(sizeof(double) * td);
    for (int i = 0; i < td; i++) { double th = 2 * ORC_PI * i / (td - 1); sines[i] = sin(th); cosines[i] = cos(th); }
    orc_map m = {H, W, res, -1.0, -1.0, 1.0, 0.0, dt};
    double fov = 4.7;
    orc_scan_cfg s = {nb, td, fov, 1e-4, 30.0, td * (fov / (nb - 1)) / (2 * ORC_PI), sines, cosines};
    double params[P_COUNT] = {1.0489, 4.718, 5.4562, 0.15875, 0.17145, 0.074, 3.74, 0.04712, -0.4189, 0.4189,
                              -3.2, 3.2, 7.319, 9.51, -5.0, 20.0, 0.31, 0.58};
    double *ang = malloc(sizeof(double) * nb), *bc = malloc(sizeof(double) * nb), *sd = malloc(sizeof(double) * nb);
    orc_beam_tables(nb, fov, params[P_WIDTH], params[P_LF], params[P_LR], ang, bc, sd);
    double *noise = calloc((size_t)8 * nb, sizeof(double));
    orc_env *e = malloc(orc_env_sizeof());
    orc_env_init(e, 2, 0, ORC_RK4, 0.01, params, &s, &m, ang, bc, sd, noise, 8);
    double poses[6] = {0.0, 0.0, 0.1, 0.3, 0.05, 0.2}; /* in the corridor's corner region, overlapping cars */
    double *scans = malloc(sizeof(double) * 2 * nb);
    int done = orc_env_reset(e, poses, scans);
    double chk = 0;
    for (int k = 0; k < 60; k++) {
        double act[4] = {0.2 * sin(k * 0.3), 2.0 + k * 0.05, -0.1, 1.0};
        done |= orc_env_step(e, act, scans);
        for (int i = 0; i < 2 * nb; i++) chk += scans[i];
    }
    /* poses outside the map and inside walls */
    double far_poses[9] = {-5, -5, 0.3, 100, 100, 1.0, 0.5, 0.5, 2.0};
    double *sc3 = malloc(sizeof(double) * 3 * nb);
    int64_t lk[3];
    orc_scan_batch(far_poses, 3, &s, &m, sc3, lk);
    for (int i = 0; i < 3 * nb; i++) chk += sc3[i];
    printf("ok done=%d checksum=%.6f lookups=%lld\n", done, chk, (long long)(lk[0] + lk[1] + lk[2]));
    free(dt); free(free_); free(sines); free(cosines); free(ang); free(bc); free(sd); free(noise); free(e); free(scans); free(sc3);
    return 0;
}
