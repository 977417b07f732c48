/*
 * oracle.c -- CPU restatement of the SCF hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product path (computational-chemistry-ai_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference (/root/reference) contains no arithmetic for this path -- it calls
 * pyscf==2.8.0 / gpu4pyscf (templates/calculate_energy.py:145-156,199-206;
 * .devcontainer/Dockerfile:119,156), neither of which is vendored, installed or fetchable here, and
 * it holds no tests or golden values.  This file therefore restates the
 * published algorithms those packages implement, and is pinned only by textbook known answers
 * (tests/test_oracle_known_answers.py): Szabo-Ostlund H2/HeH+ STO-3G, closed-form s-type integrals,
 * Boys-function values from mpmath, and remembered PySCF energies labelled "unverified-memory".
 *
 * Algorithm choice: McMurchie-Davidson Hermite expansion with Boys functions (J. Comput. Phys. 26,
 * 218 (1978)) -- deliberately NOT the Rys quadrature the HIP path uses, so the two cross-check.
 *
 * What each function stands in for (un-vendored upstream, names from SURVEY.md section 8a):
 *   orc_int1e        -> pyscf mol.intor('int1e_ovlp'|'int1e_kin'|'int1e_nuc')    (row a2)
 *   orc_schwarz      -> libcvhf CVHFnr_int2e_q_cond                              (row a3)
 *   orc_eri_shell    -> libcint int2e_sph                                        (row a4)
 *   orc_jk_direct    -> libcvhf CVHFnr_direct_drv + nrs8 J/K digestion           (rows a5, a6)
 *   orc_eri_full     -> mol.intor('int2e')                                       (test helper)
 *   orc_incore_*     -> mol.intor('int2e', aosym='s8') cached by the SCF object, and
 *   orc_jk_incore    -> libcvhf CVHFnrs8_incore_drv (PySCF's in-core J/K)        (rows a5, a6)
 *
 * Basis arrays follow the libcint atm/bas/env convention (include/mi355scf.h); nctr must be 1.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ATM_SLOTS 6
#define BAS_SLOTS 8
#define CHARGE_OF 0
#define PTR_COORD 1
#define ATOM_OF 0
#define ANG_OF 1
#define NPRIM_OF 2
#define NCTR_OF 3
#define PTR_EXP 5
#define PTR_COEFF 6

#define LMAX 4            /* g (auxiliary density-fitting shells; orbital tables stop at f) */
#define LMAX1 (LMAX + 1)
#define NCART_MAX 15
#define NSPH_MAX 9
#define TMAX (4 * LMAX + 1) /* Hermite order range for an ERI: 0..16 */

static const double PI = 3.14159265358979323846;

/* ---------- cartesian component tables and real solid harmonics -------------------------------- */

static int ncart_of(int l) { return (l + 1) * (l + 2) / 2; }

static void cart_powers(int l, int pw[][3])
{
    int n = 0;
    for (int lx = l; lx >= 0; lx--)
        for (int ly = l - lx; ly >= 0; ly--) {
            pw[n][0] = lx; pw[n][1] = ly; pw[n][2] = l - lx - ly; n++;
        }
}

/* c2s[l][cart][sph]: real solid harmonics, unit-normalised on the sphere, order m=-l..l except
 * l=1 which is (x,y,z) -- the PySCF AO convention [MEM].  Cart order xx,xy,xz,yy,yz,zz / xxx,... */
static void c2s_matrix(int l, double c[NCART_MAX][NSPH_MAX])
{
    memset(c, 0, sizeof(double) * NCART_MAX * NSPH_MAX);
    if (l == 0) { c[0][0] = 0.28209479177387814; return; }
    if (l == 1) { c[0][0] = c[1][1] = c[2][2] = 0.4886025119029199; return; }
    if (l == 2) {
        /* xx0 xy1 xz2 yy3 yz4 zz5 */
        c[1][0] = 1.0925484305920792;                                     /* xy */
        c[4][1] = 1.0925484305920792;                                     /* yz */
        c[0][2] = -0.31539156525252005; c[3][2] = -0.31539156525252005; c[5][2] = 0.6307831305050401;
        c[2][3] = 1.0925484305920792;                                     /* xz */
        c[0][4] = 0.5462742152960396;  c[3][4] = -0.5462742152960396;     /* xx-yy */
        return;
    }
    if (l == 3) {
        /* xxx0 xxy1 xxz2 xyy3 xyz4 xzz5 yyy6 yyz7 yzz8 zzz9 */
        c[1][0] = 3 * 0.5900435899266435; c[6][0] = -0.5900435899266435;             /* 3x2y - y3 */
        c[4][1] = 2.890611442640554;                                                 /* xyz */
        c[8][2] = 4 * 0.4570457994644658; c[1][2] = -0.4570457994644658; c[6][2] = -0.4570457994644658;
        c[9][3] = 2 * 0.3731763325901154; c[2][3] = -3 * 0.3731763325901154; c[7][3] = -3 * 0.3731763325901154;
        c[5][4] = 4 * 0.4570457994644658; c[0][4] = -0.4570457994644658; c[3][4] = -0.4570457994644658;
        c[2][5] = 1.445305721320277;  c[7][5] = -1.445305721320277;                  /* z(x2-y2) */
        c[0][6] = 0.5900435899266435; c[3][6] = -3 * 0.5900435899266435;             /* x3 - 3xy2 */
        return;
    }
    if (l == 4) {
        /* xxxx0 xxxy1 xxxz2 xxyy3 xxyz4 xxzz5 xyyy6 xyyz7 xyzz8 xzzz9 yyyy10 yyyz11 yyzz12 yzzz13 zzzz14;
         * real solid harmonics of the standard tables (r^2 = x^2 + y^2 + z^2 expanded):
         *  m=-4: A xy(x2-y2)   m=-3: B yz(3x2-y2)   m=-2: C xy(7z2-r2)   m=-1: D yz(7z2-3r2)   m=0: E (35z4-30z2r2+3r4)
         *  m=+1: D xz(7z2-3r2) m=+2: F (x2-y2)(7z2-r2)   m=+3: B xz(x2-3y2)   m=+4: G (x4-6x2y2+y4) */
        const double A = 0.75 * sqrt(35.0 / PI), B = 0.75 * sqrt(35.0 / (2.0 * PI)), C = 0.75 * sqrt(5.0 / PI),
                     D = 0.75 * sqrt(5.0 / (2.0 * PI)), E = 0.1875 * sqrt(1.0 / PI), F = 0.375 * sqrt(5.0 / PI),
                     G = 0.1875 * sqrt(35.0 / PI);
        c[1][0] = A; c[6][0] = -A;
        c[4][1] = 3 * B; c[11][1] = -B;
        c[8][2] = 6 * C; c[1][2] = -C; c[6][2] = -C;
        c[13][3] = 4 * D; c[4][3] = -3 * D; c[11][3] = -3 * D;
        c[0][4] = 3 * E; c[10][4] = 3 * E; c[14][4] = 8 * E; c[3][4] = 6 * E; c[5][4] = -24 * E; c[12][4] = -24 * E;
        c[9][5] = 4 * D; c[2][5] = -3 * D; c[7][5] = -3 * D;
        c[0][6] = -F; c[10][6] = F; c[5][6] = 6 * F; c[12][6] = -6 * F;
        c[2][7] = B; c[7][7] = -3 * B;
        c[0][8] = G; c[3][8] = -6 * G; c[10][8] = G;
        return;
    }
}

/* ---------- Boys function ------------------------------------------------------------------------ */

void orc_boys(int mmax, double T, double *F)
{
    if (T < 1e-14) {
        for (int m = 0; m <= mmax; m++) F[m] = 1.0 / (2 * m + 1);
        return;
    }
    if (T > 36.0 + mmax) {
        /* asymptotic F0 = sqrt(pi/T)/2, upward recursion is stable here */
        double e = exp(-T);
        F[0] = 0.5 * sqrt(PI / T) * erf(sqrt(T));
        for (int m = 0; m < mmax; m++) F[m + 1] = ((2 * m + 1) * F[m] - e) / (2 * T);
        return;
    }
    /* series for the top order, then downward recursion */
    double e = exp(-T);
    double term = 1.0 / (2 * mmax + 1), sum = term;
    for (int k = 1; k < 400; k++) {
        term *= 2 * T / (2 * mmax + 2 * k + 1);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    F[mmax] = e * sum;
    for (int m = mmax; m > 0; m--) F[m - 1] = (2 * T * F[m] + e) / (2 * m - 1);
}

/* ---------- Hermite expansion coefficients ----------------------------------------------------- */

/* E[i][j][t], 0<=i<=imax, 0<=j<=jmax, 0<=t<=i+j; includes exp(-mu X^2) in E[0][0][0]. */
#define EI (LMAX + 3)      /* room for l+2 (kinetic) */
#define ET (2 * LMAX + 5)
typedef double etab_t[EI][EI][ET];

static void hermite_E(int imax, int jmax, double a, double b, double XAB, etab_t E)
{
    double p = a + b, mu = a * b / p;
    double XPA = -b / p * XAB, XPB = a / p * XAB, h = 0.5 / p;
    memset(E, 0, sizeof(etab_t));
    E[0][0][0] = exp(-mu * XAB * XAB);
    for (int i = 0; i < imax; i++)
        for (int t = 0; t <= i + 1; t++) {
            double v = XPA * E[i][0][t] + (t + 1) * E[i][0][t + 1];
            if (t > 0) v += h * E[i][0][t - 1];
            E[i + 1][0][t] = v;
        }
    for (int j = 0; j < jmax; j++)
        for (int i = 0; i <= imax; i++)
            for (int t = 0; t <= i + j + 1; t++) {
                double v = XPB * E[i][j][t] + (t + 1) * E[i][j][t + 1];
                if (t > 0) v += h * E[i][j][t - 1];
                E[i][j + 1][t] = v;
            }
}

/* ---------- Hermite Coulomb integrals R_tuv ---------------------------------------------------- */

/* R[t][u][v] for t+u+v <= L, given alpha and PQ vector. */
typedef double rtab_t[TMAX][TMAX][TMAX];

static void hermite_R(int L, double alpha, const double PQ[3], rtab_t R)
{
    static __thread double Rn[TMAX + 1][TMAX][TMAX][TMAX];
    double F[TMAX + 1];
    double T = alpha * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
    orc_boys(L, T, F);
    double f = 1.0;
    for (int n = 0; n <= L; n++) { Rn[n][0][0][0] = f * F[n]; f *= -2.0 * alpha; }
    /* build up: R^n_{tuv} from R^{n+1} */
    for (int n = L - 1; n >= 0; n--) {
        int M = L - n; /* max t+u+v at this level */
        for (int t = 0; t <= M; t++)
            for (int u = 0; u + t <= M; u++)
                for (int v = 0; v + u + t <= M; v++) {
                    if (t + u + v == 0) continue;
                    double val;
                    if (t > 0) {
                        val = PQ[0] * Rn[n + 1][t - 1][u][v];
                        if (t > 1) val += (t - 1) * Rn[n + 1][t - 2][u][v];
                    } else if (u > 0) {
                        val = PQ[1] * Rn[n + 1][t][u - 1][v];
                        if (u > 1) val += (u - 1) * Rn[n + 1][t][u - 2][v];
                    } else {
                        val = PQ[2] * Rn[n + 1][t][u][v - 1];
                        if (v > 1) val += (v - 1) * Rn[n + 1][t][u][v - 2];
                    }
                    Rn[n][t][u][v] = val;
                }
    }
    for (int t = 0; t <= L; t++)
        for (int u = 0; u + t <= L; u++)
            for (int v = 0; v + u + t <= L; v++) R[t][u][v] = Rn[0][t][u][v];
}

/* ---------- shell helpers --------------------------------------------------------------------- */

typedef struct {
    int l, nprim, atom;
    const double *exps, *coef, *r;
} shell_t;

static shell_t get_shell(const int *atm, const int *bas, const double *env, int ish)
{
    shell_t s;
    const int *b = bas + ish * BAS_SLOTS;
    s.l = b[ANG_OF]; s.nprim = b[NPRIM_OF]; s.atom = b[ATOM_OF];
    s.exps = env + b[PTR_EXP]; s.coef = env + b[PTR_COEFF];
    s.r = env + atm[s.atom * ATM_SLOTS + PTR_COORD];
    return s;
}

int orc_nao(const int *bas, int nbas)
{
    int n = 0;
    for (int i = 0; i < nbas; i++) n += 2 * bas[i * BAS_SLOTS + ANG_OF] + 1;
    return n;
}

static void ao_offsets(const int *bas, int nbas, int *loc)
{
    loc[0] = 0;
    for (int i = 0; i < nbas; i++) loc[i + 1] = loc[i] + 2 * bas[i * BAS_SLOTS + ANG_OF] + 1;
}

int orc_check(const int *bas, int nbas)
{
    for (int i = 0; i < nbas; i++) {
        if (bas[i * BAS_SLOTS + NCTR_OF] != 1) return -1;
        if (bas[i * BAS_SLOTS + ANG_OF] > LMAX) return -2;
    }
    return 0;
}

/* transform a cartesian pair block [nca][ncb] -> spherical [nsa][nsb] */
static void c2s_pair(int la, int lb, const double *cart, double *sph)
{
    double ca[NCART_MAX][NSPH_MAX], cb[NCART_MAX][NSPH_MAX];
    c2s_matrix(la, ca); c2s_matrix(lb, cb);
    int nca = ncart_of(la), ncb = ncart_of(lb), nsa = 2 * la + 1, nsb = 2 * lb + 1;
    double tmp[NSPH_MAX][NCART_MAX];
    for (int i = 0; i < nsa; i++)
        for (int b = 0; b < ncb; b++) {
            double s = 0;
            for (int a = 0; a < nca; a++) s += ca[a][i] * cart[a * ncb + b];
            tmp[i][b] = s;
        }
    for (int i = 0; i < nsa; i++)
        for (int j = 0; j < nsb; j++) {
            double s = 0;
            for (int b = 0; b < ncb; b++) s += tmp[i][b] * cb[b][j];
            sph[i * nsb + j] = s;
        }
}

/* ---------- one-electron integrals ------------------------------------------------------------- */

/* S, T, V: [nao][nao] row-major; dip: 3 x [nao][nao] about `origin` (may be NULL -> skipped). */
void orc_int1e(const int *atm, int natm, const int *bas, int nbas, const double *env,
               double *S, double *T, double *V, double *dip, const double *origin)
{
    int nao = orc_nao(bas, nbas);
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
#pragma omp parallel for schedule(dynamic)
    for (int ish = 0; ish < nbas; ish++) {
        etab_t Ex, Ey, Ez;
        rtab_t R;
        for (int jsh = 0; jsh <= ish; jsh++) {
            shell_t A = get_shell(atm, bas, env, ish), B = get_shell(atm, bas, env, jsh);
            int la = A.l, lb = B.l, nca = ncart_of(la), ncb = ncart_of(lb);
            int pa[NCART_MAX][3], pb[NCART_MAX][3];
            cart_powers(la, pa); cart_powers(lb, pb);
            double cs[NCART_MAX * NCART_MAX] = {0}, ct[NCART_MAX * NCART_MAX] = {0},
                   cv[NCART_MAX * NCART_MAX] = {0}, cd[3][NCART_MAX * NCART_MAX];
            memset(cd, 0, sizeof(cd));
            double AB[3] = {A.r[0] - B.r[0], A.r[1] - B.r[1], A.r[2] - B.r[2]};
            for (int ip = 0; ip < A.nprim; ip++)
                for (int jp = 0; jp < B.nprim; jp++) {
                    double a = A.exps[ip], b = B.exps[jp], p = a + b;
                    double cc = A.coef[ip] * B.coef[jp];
                    double P[3];
                    for (int d = 0; d < 3; d++) P[d] = (a * A.r[d] + b * B.r[d]) / p;
                    hermite_E(la + 1, lb + 2, a, b, AB[0], Ex);
                    hermite_E(la + 1, lb + 2, a, b, AB[1], Ey);
                    hermite_E(la + 1, lb + 2, a, b, AB[2], Ez);
                    double pref = pow(PI / p, 1.5) * cc;
                    etab_t *E[3] = {&Ex, &Ey, &Ez};
                    for (int ia = 0; ia < nca; ia++)
                        for (int ib = 0; ib < ncb; ib++) {
                            double s1[3], t1[3], x1[3];
                            for (int d = 0; d < 3; d++) {
                                int i = pa[ia][d], j = pb[ib][d];
                                double sij = (*E[d])[i][j][0];
                                double tij = -2.0 * b * (2 * j + 1) * sij + 4.0 * b * b * (*E[d])[i][j + 2][0];
                                if (j >= 2) tij += j * (j - 1) * (*E[d])[i][j - 2][0];
                                s1[d] = sij; t1[d] = -0.5 * tij;
                                /* <i| (x - O) |j> = S(i+1,j) + (A - O) S(i,j) */
                                double o = origin ? origin[d] : 0.0;
                                x1[d] = (*E[d])[i + 1][j][0] + (A.r[d] - o) * sij;
                            }
                            int k = ia * ncb + ib;
                            cs[k] += pref * s1[0] * s1[1] * s1[2];
                            ct[k] += pref * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
                            cd[0][k] += pref * x1[0] * s1[1] * s1[2];
                            cd[1][k] += pref * s1[0] * x1[1] * s1[2];
                            cd[2][k] += pref * s1[0] * s1[1] * x1[2];
                        }
                    /* nuclear attraction */
                    for (int ic = 0; ic < natm; ic++) {
                        double Z = atm[ic * ATM_SLOTS + CHARGE_OF];
                        if (Z == 0) continue;
                        const double *C = env + atm[ic * ATM_SLOTS + PTR_COORD];
                        double PC[3] = {P[0] - C[0], P[1] - C[1], P[2] - C[2]};
                        hermite_R(la + lb, p, PC, R);
                        double pv = -Z * 2.0 * PI / p * cc;
                        for (int ia = 0; ia < nca; ia++)
                            for (int ib = 0; ib < ncb; ib++) {
                                double s = 0;
                                for (int t = 0; t <= pa[ia][0] + pb[ib][0]; t++)
                                    for (int u = 0; u <= pa[ia][1] + pb[ib][1]; u++)
                                        for (int v = 0; v <= pa[ia][2] + pb[ib][2]; v++)
                                            s += Ex[pa[ia][0]][pb[ib][0]][t] * Ey[pa[ia][1]][pb[ib][1]][u] *
                                                 Ez[pa[ia][2]][pb[ib][2]][v] * R[t][u][v];
                                cv[ia * ncb + ib] += pv * s;
                            }
                    }
                }
            double sph[NSPH_MAX * NSPH_MAX];
            int nsa = 2 * la + 1, nsb = 2 * lb + 1;
            double *outs[6] = {S, T, V, dip, dip ? dip + (size_t)nao * nao : NULL, dip ? dip + 2 * (size_t)nao * nao : NULL};
            double *ins[6] = {cs, ct, cv, cd[0], cd[1], cd[2]};
            for (int m = 0; m < 6; m++) {
                if (!outs[m]) continue;
                c2s_pair(la, lb, ins[m], sph);
                for (int i = 0; i < nsa; i++)
                    for (int j = 0; j < nsb; j++) {
                        outs[m][(size_t)(loc[ish] + i) * nao + loc[jsh] + j] = sph[i * nsb + j];
                        outs[m][(size_t)(loc[jsh] + j) * nao + loc[ish] + i] = sph[i * nsb + j];
                    }
            }
        }
    }
    free(loc);
}

/* ---------- shell-pair data for ERIs ----------------------------------------------------------- */

typedef struct {
    int la, lb, npp;          /* primitive pairs */
    double *p, *P, *K;        /* [npp], [npp][3], [npp] (coef product, exp factors are inside E) */
    double *E;                /* [npp][3][la+1][lb+1][la+lb+1] */
    int esz;                  /* stride of one direction table */
} pair_t;

static void build_pair(const shell_t *A, const shell_t *B, pair_t *pr)
{
    int la = A->l, lb = B->l;
    pr->la = la; pr->lb = lb; pr->npp = A->nprim * B->nprim;
    pr->esz = (la + 1) * (lb + 1) * (la + lb + 1);
    pr->p = (double *)malloc(sizeof(double) * pr->npp * 5);
    pr->P = pr->p + pr->npp; pr->K = pr->p + 4 * pr->npp;
    pr->E = (double *)malloc(sizeof(double) * pr->npp * 3 * pr->esz);
    etab_t E;
    int n = 0;
    for (int ip = 0; ip < A->nprim; ip++)
        for (int jp = 0; jp < B->nprim; jp++, n++) {
            double a = A->exps[ip], b = B->exps[jp], p = a + b;
            pr->p[n] = p;
            pr->K[n] = A->coef[ip] * B->coef[jp];
            for (int d = 0; d < 3; d++) {
                pr->P[3 * n + d] = (a * A->r[d] + b * B->r[d]) / p;
                hermite_E(la, lb, a, b, A->r[d] - B->r[d], E);
                double *dst = pr->E + ((size_t)n * 3 + d) * pr->esz;
                for (int i = 0; i <= la; i++)
                    for (int j = 0; j <= lb; j++)
                        for (int t = 0; t <= la + lb; t++)
                            dst[(i * (lb + 1) + j) * (la + lb + 1) + t] = (t <= i + j) ? E[i][j][t] : 0.0;
            }
        }
}

static void free_pair(pair_t *pr) { free(pr->p); free(pr->E); }

/* cartesian ERI block [nca*ncb][ncc*ncd] for a shell quartet from two pair_t */
static void eri_cart(const pair_t *ab, const pair_t *cd, double *out)
{
    int la = ab->la, lb = ab->lb, lc = cd->la, ld = cd->lb;
    int nca = ncart_of(la), ncb = ncart_of(lb), ncc = ncart_of(lc), ncd = ncart_of(ld);
    int nab = nca * ncb, ncdn = ncc * ncd;
    int Lab = la + lb, Lcd = lc + ld, L = Lab + Lcd;
    int pa[NCART_MAX][3], pb[NCART_MAX][3], pc[NCART_MAX][3], pd[NCART_MAX][3];
    cart_powers(la, pa); cart_powers(lb, pb); cart_powers(lc, pc); cart_powers(ld, pd);
    memset(out, 0, sizeof(double) * nab * ncdn);
    int H = Lab + 1;
    static __thread double G[(2 * LMAX + 1) * (2 * LMAX + 1) * (2 * LMAX + 1)];
    rtab_t R;
    int tab = la + lb + 1, tcd = lc + ld + 1;
    for (int n1 = 0; n1 < ab->npp; n1++) {
        double p = ab->p[n1];
        const double *P = ab->P + 3 * n1;
        const double *Eab = ab->E + (size_t)n1 * 3 * ab->esz;
        for (int n2 = 0; n2 < cd->npp; n2++) {
            double q = cd->p[n2];
            const double *Q = cd->P + 3 * n2;
            const double *Ecd = cd->E + (size_t)n2 * 3 * cd->esz;
            double alpha = p * q / (p + q);
            double PQ[3] = {P[0] - Q[0], P[1] - Q[1], P[2] - Q[2]};
            double pref = 2.0 * pow(PI, 2.5) / (p * q * sqrt(p + q)) * ab->K[n1] * cd->K[n2];
            hermite_R(L, alpha, PQ, R);
            for (int ic = 0; ic < ncc; ic++)
                for (int id = 0; id < ncd; id++) {
                    const double *ex = Ecd + 0 * cd->esz + (pc[ic][0] * (ld + 1) + pd[id][0]) * tcd;
                    const double *ey = Ecd + 1 * cd->esz + (pc[ic][1] * (ld + 1) + pd[id][1]) * tcd;
                    const double *ez = Ecd + 2 * cd->esz + (pc[ic][2] * (ld + 1) + pd[id][2]) * tcd;
                    int Tx = pc[ic][0] + pd[id][0], Ty = pc[ic][1] + pd[id][1], Tz = pc[ic][2] + pd[id][2];
                    /* G[t][u][v] = sum_ket (-1)^(tau+nu+phi) Ecd R[t+tau][u+nu][v+phi] */
                    for (int t = 0; t <= Lab; t++)
                        for (int u = 0; u + t <= Lab; u++)
                            for (int v = 0; v + u + t <= Lab; v++) {
                                double s = 0;
                                for (int a = 0; a <= Tx; a++)
                                    for (int b = 0; b <= Ty; b++)
                                        for (int c = 0; c <= Tz; c++) {
                                            double e = ex[a] * ey[b] * ez[c];
                                            if ((a + b + c) & 1) e = -e;
                                            s += e * R[t + a][u + b][v + c];
                                        }
                                G[(t * H + u) * H + v] = s;
                            }
                    for (int ia = 0; ia < nca; ia++)
                        for (int ib = 0; ib < ncb; ib++) {
                            const double *fx = Eab + 0 * ab->esz + (pa[ia][0] * (lb + 1) + pb[ib][0]) * tab;
                            const double *fy = Eab + 1 * ab->esz + (pa[ia][1] * (lb + 1) + pb[ib][1]) * tab;
                            const double *fz = Eab + 2 * ab->esz + (pa[ia][2] * (lb + 1) + pb[ib][2]) * tab;
                            int Sx = pa[ia][0] + pb[ib][0], Sy = pa[ia][1] + pb[ib][1], Sz = pa[ia][2] + pb[ib][2];
                            double s = 0;
                            for (int t = 0; t <= Sx; t++)
                                for (int u = 0; u <= Sy; u++)
                                    for (int v = 0; v <= Sz; v++)
                                        s += fx[t] * fy[u] * fz[v] * G[(t * H + u) * H + v];
                            out[(ia * ncb + ib) * ncdn + ic * ncd + id] += pref * s;
                        }
                }
        }
    }
}

/* cart -> spherical for a quartet block [nca][ncb][ncc][ncd] -> [nsa][nsb][nsc][nsd] */
static void c2s_quartet(int la, int lb, int lc, int ld, const double *cart, double *sph, double *work)
{
    int nca = ncart_of(la), ncb = ncart_of(lb), ncc = ncart_of(lc), ncd = ncart_of(ld);
    int nsa = 2 * la + 1, nsb = 2 * lb + 1, nsc = 2 * lc + 1, nsd = 2 * ld + 1;
    /* bra */
    double *t1 = work; /* [nsa*nsb][ncc*ncd] */
    int ncdn = ncc * ncd;
    double col[NCART_MAX * NCART_MAX], sp[NSPH_MAX * NSPH_MAX];
    for (int k = 0; k < ncdn; k++) {
        for (int ab = 0; ab < nca * ncb; ab++) col[ab] = cart[(size_t)ab * ncdn + k];
        c2s_pair(la, lb, col, sp);
        for (int ab = 0; ab < nsa * nsb; ab++) t1[(size_t)ab * ncdn + k] = sp[ab];
    }
    for (int ab = 0; ab < nsa * nsb; ab++)
        c2s_pair(lc, ld, t1 + (size_t)ab * ncdn, sph + (size_t)ab * nsc * nsd);
}

/* ---------- public: ERIs ---------------------------------------------------------------------- */

/* (ij|kl) spherical block for shells i,j,k,l -> out[di][dj][dk][dl] */
void orc_eri_shell(const int *atm, int natm, const int *bas, int nbas, const double *env,
                   int i, int j, int k, int l, double *out)
{
    shell_t A = get_shell(atm, bas, env, i), B = get_shell(atm, bas, env, j);
    shell_t C = get_shell(atm, bas, env, k), D = get_shell(atm, bas, env, l);
    pair_t ab, cd;
    build_pair(&A, &B, &ab); build_pair(&C, &D, &cd);
    size_t nc = (size_t)ncart_of(A.l) * ncart_of(B.l) * ncart_of(C.l) * ncart_of(D.l);
    double *cart = (double *)malloc(sizeof(double) * nc * 2);
    eri_cart(&ab, &cd, cart);
    c2s_quartet(A.l, B.l, C.l, D.l, cart, out, cart + nc);
    free(cart); free_pair(&ab); free_pair(&cd);
}

static pair_t *all_pairs(const int *atm, const int *bas, int nbas, const double *env)
{
    pair_t *pp = (pair_t *)malloc(sizeof(pair_t) * (size_t)nbas * (nbas + 1) / 2);
    for (int i = 0; i < nbas; i++)
        for (int j = 0; j <= i; j++) {
            shell_t A = get_shell(atm, bas, env, i), B = get_shell(atm, bas, env, j);
            build_pair(&A, &B, pp + (size_t)i * (i + 1) / 2 + j);
        }
    return pp;
}

static void free_pairs(pair_t *pp, int nbas)
{
    for (size_t n = 0; n < (size_t)nbas * (nbas + 1) / 2; n++) free_pair(pp + n);
    free(pp);
}

#define QBUF (NCART_MAX * NCART_MAX * NCART_MAX * NCART_MAX)

/* q[i][j] = sqrt(max |(ab|ab)|) over the components of shell pair (i,j): Schwarz bounds */
void orc_schwarz(const int *atm, int natm, const int *bas, int nbas, const double *env, double *q)
{
    pair_t *pp = all_pairs(atm, bas, nbas, env);
#pragma omp parallel
    {
        double *cart = (double *)malloc(sizeof(double) * QBUF * 3);
#pragma omp for schedule(dynamic)
        for (int i = 0; i < nbas; i++)
            for (int j = 0; j <= i; j++) {
                pair_t *ab = pp + (size_t)i * (i + 1) / 2 + j;
                eri_cart(ab, ab, cart);
                c2s_quartet(ab->la, ab->lb, ab->la, ab->lb, cart, cart + QBUF, cart + 2 * QBUF);
                int n = (2 * ab->la + 1) * (2 * ab->lb + 1);
                double m = 0;
                for (int k = 0; k < n; k++) { double v = fabs(cart[QBUF + (size_t)k * n + k]); if (v > m) m = v; }
                q[i * nbas + j] = q[j * nbas + i] = sqrt(m);
            }
        free(cart);
    }
    free_pairs(pp, nbas);
}

/* full (ij|kl) tensor [nao]^4, for small molecules only */
void orc_eri_full(const int *atm, int natm, const int *bas, int nbas, const double *env, double *eri)
{
    int nao = orc_nao(bas, nbas);
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
    pair_t *pp = all_pairs(atm, bas, nbas, env);
    size_t n1 = nao, n2 = n1 * nao, n3 = n2 * nao;
#pragma omp parallel
    {
        double *cart = (double *)malloc(sizeof(double) * QBUF * 3);
#pragma omp for schedule(dynamic)
        for (int ij = 0; ij < nbas * (nbas + 1) / 2; ij++) {
            int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
            while ((i + 1) * (i + 2) / 2 <= ij) i++;
            while (i * (i + 1) / 2 > ij) i--;
            int j = ij - i * (i + 1) / 2;
            for (int kl = 0; kl <= ij; kl++) {
                int k = (int)((sqrt(8.0 * kl + 1) - 1) / 2);
                while ((k + 1) * (k + 2) / 2 <= kl) k++;
                while (k * (k + 1) / 2 > kl) k--;
                int l = kl - k * (k + 1) / 2;
                pair_t *ab = pp + ij, *cd = pp + kl;
                eri_cart(ab, cd, cart);
                double *sph = cart + QBUF;
                c2s_quartet(ab->la, ab->lb, cd->la, cd->lb, cart, sph, cart + 2 * QBUF);
                int di = 2 * ab->la + 1, dj = 2 * ab->lb + 1, dk = 2 * cd->la + 1, dl = 2 * cd->lb + 1;
                for (int a = 0; a < di; a++)
                    for (int b = 0; b < dj; b++)
                        for (int c = 0; c < dk; c++)
                            for (int d = 0; d < dl; d++) {
                                double v = sph[((a * dj + b) * dk + c) * dl + d];
                                size_t I = loc[i] + a, J = loc[j] + b, K = loc[k] + c, Lx = loc[l] + d;
                                eri[I * n3 + J * n2 + K * n1 + Lx] = v; eri[J * n3 + I * n2 + K * n1 + Lx] = v;
                                eri[I * n3 + J * n2 + Lx * n1 + K] = v; eri[J * n3 + I * n2 + Lx * n1 + K] = v;
                                eri[K * n3 + Lx * n2 + I * n1 + J] = v; eri[Lx * n3 + K * n2 + I * n1 + J] = v;
                                eri[K * n3 + Lx * n2 + J * n1 + I] = v; eri[Lx * n3 + K * n2 + J * n1 + I] = v;
                            }
            }
        }
        free(cart);
    }
    free_pairs(pp, nbas);
    free(loc);
}

/* ---------- public: direct J/K with 8-fold symmetry and Schwarz screening ---------------------- */

/* D, J, K: [nao][nao].  J_ij = sum_kl (ij|kl) D_kl ; K_ik = sum_jl (ij|kl) D_jl.
 * Returns the number of shell quartets evaluated. */
long orc_jk_direct(const int *atm, int natm, const int *bas, int nbas, const double *env,
                   const double *D, double *J, double *K, double tol)
{
    int nao = orc_nao(bas, nbas);
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
    double *q = (double *)malloc(sizeof(double) * nbas * nbas);
    orc_schwarz(atm, natm, bas, nbas, env, q);
    pair_t *pp = all_pairs(atm, bas, nbas, env);
    size_t nn = (size_t)nao * nao;
    memset(J, 0, sizeof(double) * nn);
    memset(K, 0, sizeof(double) * nn);
    long nquart = 0;
    int npair = nbas * (nbas + 1) / 2;
#pragma omp parallel reduction(+ : nquart)
    {
        double *cart = (double *)malloc(sizeof(double) * QBUF * 3);
        double *Ja = (double *)calloc(nn, sizeof(double)), *Ka = (double *)calloc(nn, sizeof(double));
#pragma omp for schedule(dynamic, 4)
        for (int ij = npair - 1; ij >= 0; ij--) {
            int i = (int)((sqrt(8.0 * ij + 1) - 1) / 2);
            while ((i + 1) * (i + 2) / 2 <= ij) i++;
            while (i * (i + 1) / 2 > ij) i--;
            int j = ij - i * (i + 1) / 2;
            double qij = q[i * nbas + j];
            for (int kl = 0; kl <= ij; kl++) {
                int k = (int)((sqrt(8.0 * kl + 1) - 1) / 2);
                while ((k + 1) * (k + 2) / 2 <= kl) k++;
                while (k * (k + 1) / 2 > kl) k--;
                int l = kl - k * (k + 1) / 2;
                if (qij * q[k * nbas + l] < tol) continue;
                nquart++;
                pair_t *ab = pp + ij, *cd = pp + kl;
                eri_cart(ab, cd, cart);
                double *sph = cart + QBUF;
                c2s_quartet(ab->la, ab->lb, cd->la, cd->lb, cart, sph, cart + 2 * QBUF);
                double w = 1.0;
                if (i == j) w *= 0.5;
                if (k == l) w *= 0.5;
                if (ij == kl) w *= 0.5;
                int di = 2 * ab->la + 1, dj = 2 * ab->lb + 1, dk = 2 * cd->la + 1, dl = 2 * cd->lb + 1;
                for (int a = 0; a < di; a++)
                    for (int b = 0; b < dj; b++)
                        for (int c = 0; c < dk; c++)
                            for (int d = 0; d < dl; d++) {
                                double v = w * sph[((a * dj + b) * dk + c) * dl + d];
                                size_t I = loc[i] + a, Jx = loc[j] + b, Kx = loc[k] + c, Lx = loc[l] + d;
                                Ja[I * nao + Jx] += v * D[Kx * nao + Lx];
                                Ja[Kx * nao + Lx] += v * D[I * nao + Jx];
                                Ka[I * nao + Kx] += v * D[Jx * nao + Lx];
                                Ka[I * nao + Lx] += v * D[Jx * nao + Kx];
                                Ka[Jx * nao + Kx] += v * D[I * nao + Lx];
                                Ka[Jx * nao + Lx] += v * D[I * nao + Kx];
                            }
            }
        }
#pragma omp critical
        {
            for (size_t n = 0; n < nn; n++) { J[n] += Ja[n]; K[n] += Ka[n]; }
        }
        free(Ja); free(Ka); free(cart);
    }
    /* J = 2 (Jacc + Jacc^T), K = Kacc + Kacc^T */
    for (int a = 0; a < nao; a++)
        for (int b = 0; b <= a; b++) {
            double js = 2.0 * (J[a * nao + b] + J[b * nao + a]);
            double ks = K[a * nao + b] + K[b * nao + a];
            J[a * nao + b] = J[b * nao + a] = js;
            K[a * nao + b] = K[b * nao + a] = ks;
        }
    free_pairs(pp, nbas);
    free(q); free(loc);
    return nquart;
}

/* ---------- public: in-core J/K from packed ERIs ------------------------------------------------- */

/* What PySCF's CPU path does when the 8-fold-unique ERI array fits `max_memory` (N = 264 is 4.9 GB): the SCF object
 * caches `mol.intor('int2e', aosym='s8')` once and every cycle runs `_vhf.incore(eri, dm)` -> CVHFnrs8_incore_drv [MEM]
 * (call site in the reference: templates/calculate_energy.py:199-206, the CPU rung `scf.RHF(mol).kernel()`).
 *
 * Layout: row ij = i(i+1)/2 + j (AO indices, i >= j) holds (ij|kl) for kl = 0..ij at buf[row_off[ij] + kl].  With
 * row_off[ij] = ij(ij+1)/2 this is exactly the s8 array.  For bench.py's BOUNDED CPU sample only the rows of every
 * `stride`-th shell pair (pair index % stride == phase) are kept (row_off = -1 elsewhere).
 *
 * orc_incore_layout -> doubles needed, fills row_off[nao(nao+1)/2].
 * orc_incore_fill   -> evaluates the rows' quartets (Schwarz q_ij q_kl >= tol), returns the number of shell quartets.
 * orc_jk_incore     -> J, K (this row subset's contribution; the full matrices for stride = 1). */
long orc_incore_layout(const int *bas, int nbas, int stride, int phase, long *row_off)
{
    int nao = orc_nao(bas, nbas);
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
    long npair = (long)nao * (nao + 1) / 2, off = 0;
    for (long n = 0; n < npair; n++) row_off[n] = -1;
    for (int si = 0; si < nbas; si++)
        for (int sj = 0; sj <= si; sj++) {
            long P = (long)si * (si + 1) / 2 + sj;
            if (stride > 1 && P % stride != phase) continue;
            for (int I = loc[si]; I < loc[si + 1]; I++)
                for (int J = loc[sj]; J < loc[sj + 1] && J <= I; J++) {
                    long ij = (long)I * (I + 1) / 2 + J;
                    row_off[ij] = off;
                    off += ij + 1;
                }
        }
    free(loc);
    return off;
}

long orc_incore_fill(const int *atm, int natm, const int *bas, int nbas, const double *env, double tol, int stride,
                     int phase, const long *row_off, double *buf)
{
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
    double *q = (double *)malloc(sizeof(double) * nbas * nbas);
    orc_schwarz(atm, natm, bas, nbas, env, q);
    pair_t *pp = all_pairs(atm, bas, nbas, env);
    long nquart = 0;
    int npair = nbas * (nbas + 1) / 2;
#pragma omp parallel reduction(+ : nquart)
    {
        double *cart = (double *)malloc(sizeof(double) * QBUF * 3);
#pragma omp for schedule(dynamic, 1)
        for (int P = npair - 1; P >= 0; P--) {
            if (stride > 1 && P % stride != phase) continue;
            int si = (int)((sqrt(8.0 * P + 1) - 1) / 2);
            while ((si + 1) * (si + 2) / 2 <= P) si++;
            while (si * (si + 1) / 2 > P) si--;
            int sj = P - si * (si + 1) / 2;
            /* every ket shell pair (sk >= sl) with sk <= si can hold AO pairs kl <= ij */
            for (int sk = 0; sk <= si; sk++)
                for (int sl = 0; sl <= sk; sl++) {
                    if (q[si * nbas + sj] * q[sk * nbas + sl] < tol) continue;
                    pair_t *ab = pp + P, *cd = pp + (size_t)sk * (sk + 1) / 2 + sl;
                    int di = 2 * ab->la + 1, dj = 2 * ab->lb + 1, dk = 2 * cd->la + 1, dl = 2 * cd->lb + 1;
                    /* any element with kl <= ij?  smallest kl of the block vs largest ij of the block */
                    long ijmax = (long)(loc[si] + di - 1) * (loc[si] + di) / 2 + (loc[sj] + dj - 1);
                    long klmin = (long)loc[sk] * (loc[sk] + 1) / 2 + loc[sl];
                    if (klmin > ijmax) continue;
                    nquart++;
                    eri_cart(ab, cd, cart);
                    double *sph = cart + QBUF;
                    c2s_quartet(ab->la, ab->lb, cd->la, cd->lb, cart, sph, cart + 2 * QBUF);
                    for (int a = 0; a < di; a++)
                        for (int b = 0; b < dj; b++) {
                            long I = loc[si] + a, J = loc[sj] + b;
                            if (J > I) continue;
                            long ij = I * (I + 1) / 2 + J;
                            double *row = buf + row_off[ij];
                            for (int c = 0; c < dk; c++)
                                for (int d = 0; d < dl; d++) {
                                    long K = loc[sk] + c, L = loc[sl] + d;
                                    if (L > K) continue;
                                    long kl = K * (K + 1) / 2 + L;
                                    if (kl <= ij) row[kl] = sph[((a * dj + b) * dk + c) * dl + d];
                                }
                        }
                }
        }
        free(cart);
    }
    free_pairs(pp, nbas);
    free(q); free(loc);
    return nquart;
}

void orc_jk_incore(const int *bas, int nbas, const long *row_off, const double *buf, const double *D, double *J, double *K)
{
    int nao = orc_nao(bas, nbas);
    size_t nn = (size_t)nao * nao;
    memset(J, 0, sizeof(double) * nn);
    memset(K, 0, sizeof(double) * nn);
    /* the rows present (all of them, or bench.py's sample), longest first, dealt dynamically to the threads */
    long npair_ao = (long)nao * (nao + 1) / 2, nrows = 0;
    int *rowI = (int *)malloc(sizeof(int) * npair_ao), *rowJ = (int *)malloc(sizeof(int) * npair_ao);
    for (int I = nao - 1; I >= 0; I--)
        for (int Jx = I; Jx >= 0; Jx--)
            if (row_off[(long)I * (I + 1) / 2 + Jx] >= 0) { rowI[nrows] = I; rowJ[nrows] = Jx; nrows++; }
    /* thread-private accumulators in one block, summed afterwards by ALL threads (each a slice of the matrix): a critical
     * section here serialised 128 x 2 N^2 additions -- a third of the digestion time of benzene/cc-pVTZ on 128 threads.
     * The inner loop is branch-free: every element is digested with the row weight, the two special elements of a row
     * segment -- (K,K), weight 1/2, and the last element (ij|ij), another 1/2 -- are corrected afterwards. */
    int nth = 1;
#ifdef _OPENMP
    nth = omp_get_max_threads();
#endif
    double *acc = (double *)malloc((size_t)nth * 2 * nn * sizeof(double));
#pragma omp parallel num_threads(nth)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *Ja = acc + (size_t)tid * 2 * nn, *Ka = Ja + nn;
        memset(Ja, 0, 2 * nn * sizeof(double));     /* first touch by the owner */
#pragma omp for schedule(dynamic, 16)
        for (long r = 0; r < nrows; r++) {
            const int I = rowI[r], Jx = rowJ[r];
            const long ij = (long)I * (I + 1) / 2 + Jx;
            const double *row = buf + row_off[ij];
            const double W = (I == Jx) ? 0.5 : 1.0;
            const double *DI = D + (size_t)I * nao, *DJ = D + (size_t)Jx * nao;
            double *KI = Ka + (size_t)I * nao, *KJ = Ka + (size_t)Jx * nao;
            const double dijW = DI[Jx] * W;
            double jij = 0.0;
            long kl = 0;
            for (int Kx = 0; Kx <= I; Kx++) {
                const int lmax = (Kx == I) ? Jx : Kx;
                const double *seg = row + kl, *DK = D + (size_t)Kx * nao;
                double *JK = Ja + (size_t)Kx * nao;
                const double dikW = DI[Kx] * W, djkW = DJ[Kx] * W;
                double kik = 0.0, kjk = 0.0, js = 0.0;
                for (int Lx = 0; Lx <= lmax; Lx++) {
                    const double v = seg[Lx];
                    js += v * DK[Lx];
                    JK[Lx] += v * dijW;
                    kik += v * DJ[Lx];
                    KI[Lx] += v * djkW;
                    kjk += v * DI[Lx];
                    KJ[Lx] += v * dikW;
                }
                if (lmax == Kx) {            /* (K,K) carries weight 1/2: take half of it back */
                    const double c = -0.5 * seg[Kx];
                    js += c * DK[Kx];
                    JK[Kx] += c * dijW;
                    kik += c * DJ[Kx];
                    KI[Kx] += c * djkW;
                    kjk += c * DI[Kx];
                    KJ[Kx] += c * dikW;
                }
                if (Kx == I) {               /* last element of the row, (ij|ij): another factor 1/2 */
                    const double c = -0.5 * ((I == Jx) ? 0.5 : 1.0) * seg[Jx];
                    js += c * DK[Jx];
                    JK[Jx] += c * dijW;
                    kik += c * DJ[Jx];
                    KI[Jx] += c * djkW;
                    kjk += c * DI[Jx];
                    KJ[Jx] += c * dikW;
                }
                KI[Kx] += kik * W;
                KJ[Kx] += kjk * W;
                jij += js;
                kl += lmax + 1;
            }
            Ja[(size_t)I * nao + Jx] += jij * W;
        }
        /* every thread sums one contiguous slice of [J|K] over the thread copies (contiguous reads per copy) */
        {
            const long tot = 2 * (long)nn, per = (tot + nth - 1) / nth;
            const long lo = (long)tid * per, hi = (lo + per < tot) ? lo + per : tot;
            for (long n = lo; n < hi; n++) { if (n < (long)nn) J[n] = 0.0; else K[n - nn] = 0.0; }
#pragma omp barrier
            for (int t = 0; t < nth; t++) {
                const double *src = acc + (size_t)t * 2 * nn;
                for (long n = lo; n < hi; n++) { if (n < (long)nn) J[n] += src[n]; else K[n - nn] += src[n]; }
            }
        }
    }
    free(acc);
    free(rowI); free(rowJ);
    for (int a = 0; a < nao; a++)
        for (int b = 0; b <= a; b++) {
            double js = 2.0 * (J[a * nao + b] + J[b * nao + a]);
            double ks = K[a * nao + b] + K[b * nao + a];
            J[a * nao + b] = J[b * nao + a] = js;
            K[a * nao + b] = K[b * nao + a] = ks;
        }
}

/* One shell block of J and of K by brute force over ALL ket shells (no symmetry, no screening): the full-size
 * spot check for molecules whose complete J/K the oracle cannot afford (C60/6-31G*, ibuprofen/def2-TZVP).
 *   Jblk[a][b] = sum_{cd} (ab|cd) D_cd   for a in shell sa, b in shell sb
 *   Kblk[a][c] = sum_{bd} (ab|cd) D_bd   for a in shell sa, c in shell sb     (both blocks use the SAME shell pair sa, sb) */
void orc_jk_shellblock(const int *atm, int natm, const int *bas, int nbas, const double *env, int sa, int sb,
                       const double *D, double *Jblk, double *Kblk)
{
    int nao = orc_nao(bas, nbas);
    int *loc = (int *)malloc(sizeof(int) * (nbas + 1));
    ao_offsets(bas, nbas, loc);
    shell_t A = get_shell(atm, bas, env, sa), B = get_shell(atm, bas, env, sb);
    int da = 2 * A.l + 1, db = 2 * B.l + 1;
    for (int n = 0; n < da * db; n++) { Jblk[n] = 0.0; Kblk[n] = 0.0; }
    pair_t ab;
    build_pair(&A, &B, &ab);
#pragma omp parallel
    {
        double *cart = (double *)malloc(sizeof(double) * QBUF * 3);
        double Jl[NSPH_MAX * NSPH_MAX] = {0}, Kl[NSPH_MAX * NSPH_MAX] = {0};
#pragma omp for schedule(dynamic, 4) collapse(2)
        for (int sc = 0; sc < nbas; sc++)
            for (int sd = 0; sd < nbas; sd++) {
                shell_t C = get_shell(atm, bas, env, sc), Dd = get_shell(atm, bas, env, sd);
                int dc = 2 * C.l + 1, dd = 2 * Dd.l + 1;
                double *sph = cart + QBUF;
                pair_t cd;
                /* J: (sa sb | sc sd) */
                build_pair(&C, &Dd, &cd);
                eri_cart(&ab, &cd, cart);
                c2s_quartet(A.l, B.l, C.l, Dd.l, cart, sph, cart + 2 * QBUF);
                for (int a = 0; a < da; a++)
                    for (int b = 0; b < db; b++)
                        for (int c = 0; c < dc; c++)
                            for (int d = 0; d < dd; d++)
                                Jl[a * db + b] += sph[((a * db + b) * dc + c) * dd + d] * D[(size_t)(loc[sc] + c) * nao + loc[sd] + d];
                free_pair(&cd);
                /* K: (sa sc | sb sd), contracted over the SECOND and FOURTH index */
                pair_t ac, bd;
                build_pair(&A, &C, &ac);
                build_pair(&B, &Dd, &bd);
                eri_cart(&ac, &bd, cart);
                c2s_quartet(A.l, C.l, B.l, Dd.l, cart, sph, cart + 2 * QBUF);
                for (int a = 0; a < da; a++)
                    for (int c = 0; c < dc; c++)
                        for (int b = 0; b < db; b++)
                            for (int d = 0; d < dd; d++)
                                Kl[a * db + b] += sph[((a * dc + c) * db + b) * dd + d] * D[(size_t)(loc[sc] + c) * nao + loc[sd] + d];
                free_pair(&ac); free_pair(&bd);
            }
#pragma omp critical
        for (int n = 0; n < da * db; n++) { Jblk[n] += Jl[n]; Kblk[n] += Kl[n]; }
        free(cart);
    }
    free_pair(&ab);
    free(loc);
}

/* number of OpenMP threads the following calls use (bench.py: the CPU share the container really has, cgroup cpu.max) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* c2s table for tests and the numpy AO evaluator: out[NCART_MAX = 15][NSPH_MAX = 9] */
void orc_c2s(int l, double *out)
{
    double c[NCART_MAX][NSPH_MAX];
    c2s_matrix(l, c);
    memcpy(out, c, sizeof(c));
}
