/* Plain-C consumer of include/mi355scf.h (no Python, no torch): H2/STO-3G at R = 1.4 a0.
 * Builds the libcint-layout arrays by hand, gets S/T/V, the resident ERIs and J/K for D = 2 c c^T of the
 * sigma_g orbital, and prints the RHF energy (Szabo-Ostlund: -1.1167 Ha).  Compiled and run by
 * tests/test_gpu_c_abi.py:  gcc c_abi_smoke.c -I../include -L<csrc> -lmi355scf -L/opt/rocm/lib -lamdhip64 -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mi355scf.h"
#include <hip/hip_runtime_api.h>

static double gint(int n, double a) { return tgamma((n + 1) * 0.5) / (2.0 * pow(a, (n + 1) * 0.5)); }

int main(void)
{
    const double ex[3] = {3.42525091, 0.62391373, 0.16885540}, cf[3] = {0.15432897, 0.53532814, 0.44463454};
    double env[64] = {0};
    int32_t atm[2][6] = {{0}}, bas[2][8] = {{0}};
    int p = 20;
    for (int a = 0; a < 2; a++) { atm[a][0] = 1; atm[a][1] = p; env[p + 2] = a ? 1.4 : 0.0; p += 4; }
    int pe = p; for (int i = 0; i < 3; i++) env[p++] = ex[i];
    double c[3], s = 0;
    for (int i = 0; i < 3; i++) c[i] = cf[i] / sqrt(gint(2, 2 * ex[i]));
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) s += c[i] * c[j] * gint(2, ex[i] + ex[j]);
    int pc = p; for (int i = 0; i < 3; i++) env[p++] = c[i] / sqrt(s);
    for (int a = 0; a < 2; a++) { bas[a][0] = a; bas[a][1] = 0; bas[a][2] = 3; bas[a][3] = 1; bas[a][5] = pe; bas[a][6] = pc; }

    mi_ctx *ctx = NULL;
    if (mi_ctx_create(&atm[0][0], 2, &bas[0][0], 2, env, p, 0, &ctx)) { fprintf(stderr, "%s\n", mi_last_error()); return 1; }
    double *d, h[4 * 6];
    if (hipMalloc((void **)&d, sizeof(double) * 4 * 6) != hipSuccess) return 2;
    double *dS = d, *dT = d + 4, *dV = d + 8, *dD = d + 12, *dJ = d + 16, *dK = d + 20;
    if (mi_int1e(ctx, dS, dT, dV, NULL, NULL, NULL) || mi_eri_prepare(ctx, 1e-13, 0, 1, NULL)) { fprintf(stderr, "%s\n", mi_last_error()); return 3; }
    hipMemcpy(h, d, sizeof(double) * 12, hipMemcpyDeviceToHost);
    double S12 = h[1], n2 = 1.0 / (2.0 + 2.0 * S12);            /* sigma_g = (a+b)/sqrt(2+2S) */
    double D[4] = {2 * n2, 2 * n2, 2 * n2, 2 * n2};
    hipMemcpy(dD, D, sizeof D, hipMemcpyHostToDevice);
    if (mi_build_jk(ctx, dD, 1, dJ, dK, NULL)) { fprintf(stderr, "%s\n", mi_last_error()); return 4; }
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(double) * 24, hipMemcpyDeviceToHost);
    double e = 0;
    for (int i = 0; i < 4; i++) e += D[i] * (h[4 + i] + h[8 + i] + 0.5 * (h[16 + i] - 0.5 * h[20 + i]));
    e += 1.0 / 1.4;
    mi_eri_stats st;
    mi_eri_get_stats(ctx, &st);
    printf("S12 = %.6f  E(RHF) = %.8f  tiles = %ld\n", S12, e, (long)st.n_tiles);
    mi_ctx_destroy(ctx);
    mi_release_cache();
    hipFree(d);
    return (fabs(e + 1.1167143) < 1e-6 && fabs(S12 - 0.6593182) < 1e-6) ? 0 : 5;
}
