// mi355scf.hip -- MI355X (gfx950 / CDNA4) SCF Fock-build engine: C ABI + HIP kernels.
//
// See include/mi355scf.h for the boundary and DESIGN.md for the data layout.  Kernels:
//   int1e_kernel            S, T, V, dipole (Obara-Saika overlap recurrences + Rys nuclear attraction)
//   eri_rys_kernel          contracted [e0|f0] integrals by Rys quadrature, one wave per shell quartet,
//                           2-D recurrence tables staged in LDS
//   eri_transform_scatter   HRR + cart->sph as two small dense products, scatter into 8^4 AO tiles
//   schwarz_diag_kernel     q_ab = sqrt(max (ab|ab))                      (row a3)
//   jk_tiles_kernel         one-pass J+K digestion of the HBM-resident tiles (rows a5, a6)
//   pad/finalize, DIIS helpers                                             (row a10)
// Written for gfx950 only: 64-lane waves are assumed throughout.

#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <mutex>
#include <map>
#include <unordered_map>
#include <vector>

#include "../../include/mi355scf.h"
#include "rys_tables.h"

#define LMAX 4      /* g shells: auxiliary (density-fitting) contexts only; orbital shells stop at f (LMAX_1E) */
#define LMAX_1E 3   /* one-electron / AO-value kernels: their per-thread blocks are sized for l <= f */
#define NPC ((LMAX + 1) * (LMAX + 2) / 2) /* pair classes (la>=lb) */
#define BLK 8
#define ATM_SLOTS 6
#define BAS_SLOTS 8

static thread_local std::string g_err;
static int fail(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *mi_last_error(void) { return g_err.c_str(); }

// -------------------------------------------------------------------------------------------------
// Device-memory pool for the MANY small and medium buffers of a context (pair records, matrices, directories, per-call
// scratch): a geometry step creates and destroys ~150 of them, and every hipMalloc / hipFree maps or unmaps pages and
// synchronises the device (38 ms for the mi_ctx_destroy of ibuprofen/def2-TZVP alone).  dev_free parks a block in a per-device
// free list by size class (device-synchronised first, like hipFree, so a parked block is never still being read), dev_malloc
// takes one from there.  Blocks above 1 GiB, the tile store (TileArena) and the hand-over buffers (Scratch) bypass the pool.
// At most POOL_CAP bytes are parked per device; mi_release_cache and an out-of-memory hipMalloc empty the pool.
// -------------------------------------------------------------------------------------------------
static const size_t POOL_CAP = (size_t)6 << 30, POOL_MAX_BLOCK = (size_t)1 << 30;
struct DevPool {
    std::mutex mu;
    std::unordered_map<void *, size_t> live;                  // pooled blocks handed out: pointer -> size class
    std::map<size_t, std::vector<void *>> parked;             // size class -> free blocks
    size_t parked_bytes = 0;
};
static DevPool g_pool[16];
static size_t pool_class(size_t n)
{
    if (n <= 256) return 256;
    size_t c = 256;
    while (c < n) c <<= 1;
    const size_t three_q = c / 4 * 3;                          // classes 2^k and 3 * 2^(k-2): at most 33 % of slack
    return (n <= three_q && three_q >= 256) ? three_q : c;
}
static void pool_flush(int dev)
{
    DevPool &P = g_pool[dev & 15];
    for (auto &kv : P.parked)
        for (void *q : kv.second) hipFree(q);
    P.parked.clear();
    P.parked_bytes = 0;
}
static hipError_t dev_malloc_impl(void **out, size_t n)
{
    int dev = 0;
    hipGetDevice(&dev);
    if (n > POOL_MAX_BLOCK) return hipMalloc(out, n);
    DevPool &P = g_pool[dev & 15];
    const size_t cls = pool_class(std::max<size_t>(n, 1));
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.parked.find(cls);
    if (it != P.parked.end() && !it->second.empty()) {
        *out = it->second.back();
        it->second.pop_back();
        P.parked_bytes -= cls;
        P.live[*out] = cls;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, cls);
    if (e != hipSuccess) {                                    // out of memory: give the parked blocks back and try again
        (void)hipGetLastError();
        pool_flush(dev);
        e = hipMalloc(out, cls);
    }
    if (e == hipSuccess) P.live[*out] = cls;
    return e;
}
template <class T> static hipError_t dev_malloc(T **out, size_t n) { return dev_malloc_impl((void **)out, n); }
static hipError_t dev_free(void *q)
{
    if (!q) return hipSuccess;
    int dev = 0;
    hipGetDevice(&dev);
    DevPool &P = g_pool[dev & 15];
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(q);
        if (it != P.live.end()) { cls = it->second; P.live.erase(it); }
    }
    if (!cls) return hipFree(q);                               // not one of ours (large block, or allocated on another device)
    hipDeviceSynchronize();                                   // what hipFree would have done: nobody reads the block any more
    std::lock_guard<std::mutex> g(P.mu);
    if (P.parked_bytes + cls > POOL_CAP) return hipFree(q);
    P.parked[cls].push_back(q);
    P.parked_bytes += cls;
    return hipSuccess;
}
static size_t pool_parked_bytes(int dev) { DevPool &P = g_pool[dev & 15]; std::lock_guard<std::mutex> g(P.mu); return P.parked_bytes; }

// Host-side OpenMP is used only for the per-pair transformation matrices.  Idle workers must not spin
// (libomp's default 200 ms block time starves the Python/torch threads between calls).
static int g_rows_g32 = 1;   // (option grad_rows_g32) row gradient kernel: lane groups of 32 (two quartets per wave) up to 64 rows
static int host_threads()
{
    static int n = [] {
        setenv("KMP_BLOCKTIME", "0", 0);
        const char *e = getenv("MI355_HOST_THREADS");
        int v = e ? atoi(e) : 16;
        return v < 1 ? 1 : v;
    }();
    return n;
}
extern "C" int mi_abi_version(void) { return 2; }

// =================================================================================================
// Real solid harmonics (generic l): coefficients of the unit-normalised real harmonics r^l Y_lm in
// monomials x^a y^b z^c; order m=-l..l, except l=1 -> (x,y,z) (PySCF AO convention [MEM]).
// Formula: Helgaker, Jorgensen, Olsen, "Molecular Electronic-Structure Theory", eq. 6.4.47-6.4.50.
// =================================================================================================
static inline int ncart(int l) { return (l + 1) * (l + 2) / 2; }
static inline int cart_index(int l, int lx, int ly) /* order: lx desc, then ly desc */
{
    int n = 0;
    for (int x = l; x > lx; x--) n += l - x + 1;
    return n + (l - lx - ly);
}
static double fact(int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }
static double binom(int n, int k) { if (k < 0 || k > n) return 0; return fact(n) / (fact(k) * fact(n - k)); }

static void c2s_generic(int l, std::vector<double> &out) /* [ncart][2l+1] */
{
    int nc = ncart(l), ns = 2 * l + 1;
    out.assign((size_t)nc * ns, 0.0);
    for (int m = -l; m <= l; m++) {
        int am = std::abs(m);
        double N = 1.0 / (std::pow(2.0, am) * fact(l)) * std::sqrt(2.0 * fact(l + am) * fact(l - am) / (m == 0 ? 2.0 : 1.0));
        N *= std::sqrt((2 * l + 1) / (4.0 * M_PI));
        int col = (l == 1) ? (m == 1 ? 0 : (m == -1 ? 1 : 2)) : (m + l);
        for (int t = 0; t <= (l - am) / 2; t++)
            for (int u = 0; u <= t; u++) {
                /* v = vm, vm+1, ... with 2v <= am ; vm = 0 (m>=0) or 1/2 (m<0): use v2 = 2v */
                for (int v2 = (m < 0 ? 1 : 0); v2 <= am; v2 += 2) {
                    double C = std::pow(-1.0, t + (v2 - (m < 0 ? 1 : 0)) / 2) * std::pow(0.25, t) * binom(l, t) *
                               binom(l - t, am + t) * binom(t, u) * binom(am, v2);
                    int px = 2 * t + am - 2 * u - v2, py = 2 * u + v2, pz = l - 2 * t - am;
                    if (px < 0 || py < 0 || pz < 0) continue;
                    out[(size_t)cart_index(l, px, py) * ns + col] += N * C;
                }
            }
    }
}

extern "C" int mi_c2s_table(int l, double *out)
{
    if (l < 0 || l > 6) return fail("mi_c2s_table: l=%d out of range", l);
    std::vector<double> c;
    c2s_generic(l, c);
    memcpy(out, c.data(), c.size() * sizeof(double));
    return 0;
}

// =================================================================================================
// Rys roots and weights from the generated Chebyshev tables (host + device versions)
// =================================================================================================
struct RysDev {
    const double *cheb;   // RYS_CHEB
    const double *herm_r; // [(NMAX+1)*NMAX]
    const double *herm_w;
    int off[RYS_NMAX + 2];
    int nint[RYS_NMAX + 1];
};

template <class T>
__device__ inline double rys_eval_impl(const T &R, int n, int f, double x)
{
    int ni = R.nint[n];
    if (x < ni * RYS_H) {
        int iv = (int)(x * (1.0 / RYS_H));
        if (iv >= ni) iv = ni - 1;
        double s = (x - (iv * RYS_H + 0.5 * RYS_H)) * (2.0 / RYS_H);
        const double *c = R.cheb + R.off[n] + (size_t)(iv * 2 * n + f) * (RYS_DEG + 1);
        double b1 = 0.0, b2 = 0.0, s2 = 2.0 * s;
#pragma unroll
        for (int k = RYS_DEG; k >= 1; k--) {
            double t = s2 * b1 - b2 + c[k];
            b2 = b1;
            b1 = t;
        }
        return s * b1 - b2 + c[0];
    }
    if (f < n) return R.herm_r[n * RYS_NMAX + f] / x;
    return R.herm_w[n * RYS_NMAX + (f - n)] * rsqrt(x);
}
__device__ inline double rys_eval(const RysDev &R, int n, int f, double x) { return rys_eval_impl(R, n, f, x); }

extern "C" int mi_rys_roots_host(int n, double x, double *roots, double *weights)
{
    if (n < 1 || n > RYS_NMAX) return fail("nroots out of range");
    struct H { const double *cheb, *herm_r, *herm_w; int off[RYS_NMAX + 2]; int nint[RYS_NMAX + 1]; } R;
    R.cheb = RYS_CHEB_H; R.herm_r = &RYS_HERM_R_H[0][0]; R.herm_w = &RYS_HERM_W_H[0][0];
    for (int i = 0; i <= RYS_NMAX; i++) { R.off[i] = i == 0 ? 0 : RYS_OFFSET_H[i]; R.nint[i] = RYS_NINT_H[i]; }
    for (int f = 0; f < n; f++) {
        // host path mirrors rys_eval_impl but with std:: functions
        int ni = R.nint[n];
        for (int pass = 0; pass < 2; pass++) {
            int ff = f + pass * n;
            double val;
            if (x < ni * RYS_H) {
                int iv = (int)(x / RYS_H);
                if (iv >= ni) iv = ni - 1;
                double s = (x - (iv * RYS_H + 0.5 * RYS_H)) * (2.0 / RYS_H);
                const double *c = R.cheb + R.off[n] + (size_t)(iv * 2 * n + ff) * (RYS_DEG + 1);
                double b1 = 0, b2 = 0;
                for (int k = RYS_DEG; k >= 1; k--) { double t = 2 * s * b1 - b2 + c[k]; b2 = b1; b1 = t; }
                val = s * b1 - b2 + c[0];
            } else if (pass == 0) val = R.herm_r[n * RYS_NMAX + f] / x;
            else val = R.herm_w[n * RYS_NMAX + f] / std::sqrt(x);
            (pass == 0 ? roots : weights)[f] = val;
        }
    }
    return 0;
}

// =================================================================================================
// Context
// =================================================================================================
struct ShellH {
    int atom, l, nprim, ao;   // ao: first AO in the TILE order of the ERI store (see set_tile_order); ao_nat: in the caller's order
    int ao_nat;
    const double *exps, *coef;
    double r[3];
};

struct PairRec { // one shell pair (l_i >= l_j)
    int sh_i, sh_j, ao_i, ao_j;
    int prim_off, nprim; // primitive pairs: 8 doubles each {p,Px,Py,Pz,PAx,PAy,PAz,K}
    int m_off;           // offset (doubles) of the [ns_i*ns_j][ne] HRR*c2s matrix
    int pad;
};

struct PairClass {
    // gradient variants of each (sorted) pair: [orientation 0: first = sh_i | 1: first = sh_j][0: l+1 | 1: l-1]
    std::vector<PairRec> g_recs[2][2];
    PairRec *d_g_recs[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    int la, lb;
    int ne;                        // number of [e0| cartesian components, e = la..la+lb
    int nsab;                      // (2la+1)(2lb+1)
    std::vector<PairRec> recs;     // sorted by q desc after Schwarz
    std::vector<double> q;         // Schwarz bound per rec
    std::vector<double> cq;        // non-increasing along the sorted list: max q over the rec's cluster (== q without clustering)
    PairRec *d_recs = nullptr;
    double *d_q = nullptr;
    // primitive-pair counts of the sorted list: mean, and mean over windows of 4 consecutive pairs of the window maximum
    // (four consecutive tasks share a wave in the 16-lane-group kernels and run as long as the longest of them)
    double mean_np = 1.0, max4_np = 1.0;
    int max_np = 1;                // largest primitive-pair count of a pair of the class
};

struct TileInfo { int I, J, K, L; };
struct RunRec { int J, K, L, first, count; };

// ---- tile geometry.  A tile (I J|K L) is a list of rows (j, m) -- j = 0..7, m = l-pair 0..3, j-major -- of 16-byte chunks
// {T[i,j,k,2m], T[i,j,k,2m+1]}, the chunks of a row ordered (i, k).  Off the block diagonals a row holds all bi*bk lanes.
// Where I == J only i >= j is stored (weight 1 off the diagonal, 1/2 on it: the kernels' contractions are symmetric in
// (i,j) for a symmetric density, so the dropped half would only repeat the kept one), where K == L only k >= 2m (elements
// with k < l inside a kept chunk are zero).  Rows shrink accordingly and stay contiguous in lane order, so every load of the
// digestion kernel is still a coalesced run of 16-byte chunks; the two-fold redundancy of diagonal tiles is gone from the stream.
__host__ __device__ inline int tile_i0(bool dij, int j, int bi) { return dij ? (j < bi ? j : bi) : 0; }
__host__ __device__ inline int tile_k0(bool dkl, int m, int bk) { return dkl ? (2 * m < bk ? 2 * m : bk) : 0; }
__host__ __device__ inline int tile_pi(bool dij, int j, int bi)   // rows of i-lanes before row j
{
    if (!dij) return j * bi;
    const int jc = j < bi ? j : bi;
    return jc * bi - jc * (jc - 1) / 2;
}
__host__ __device__ inline int tile_qk(bool dkl, int m, int bk)   // k-lanes per i in the l-pairs before m
{
    if (!dkl) return m * bk;
    int q = 0;
    for (int mm = 0; mm < m; mm++) q += bk - tile_k0(true, mm, bk);
    return q;
}
__host__ __device__ inline int tile_chunks(bool dij, bool dkl, int bi, int bk) { return tile_pi(dij, BLK, bi) * tile_qk(dkl, 4, bk); }
// chunk index of (ii, jj, kk, m) inside its tile, -1 when that lane is not stored
__host__ __device__ inline int tile_chunk(bool dij, bool dkl, int bi, int bk, int ii, int jj, int kk, int m)
{
    const int i0 = tile_i0(dij, jj, bi), k0 = tile_k0(dkl, m, bk);
    if (ii < i0 || kk < k0) return -1;
    return tile_pi(dij, jj, bi) * tile_qk(dkl, 4, bk) + (bi - i0) * tile_qk(dkl, m, bk) + (ii - i0) * (bk - k0) + (kk - k0);
}
// offset in doubles of element (ii,jj,kk,ll) and the weight it is stored with (0: not stored)
// tri = false: the round-1 layout (full rows everywhere, 1/2 per block coincidence on both mirror elements)
__host__ __device__ inline int64_t tile_elem(bool tri, bool dij, bool dkl, bool dpp, int bi, int bk, int ii, int jj, int kk, int ll, double *w)
{
    double wt = dpp ? 0.5 : 1.0;
    if (!tri) {
        if (dij) wt *= 0.5;
        if (dkl) wt *= 0.5;
        *w = wt;
        return (int64_t)tile_chunk(false, false, bi, bk, ii, jj, kk, ll >> 1) * 2 + (ll & 1);
    }
    if (dij) { if (ii < jj) { *w = 0.0; return -1; } if (ii == jj) wt *= 0.5; }
    if (dkl) { if (kk < ll) { *w = 0.0; return -1; } if (kk == ll) wt *= 0.5; }
    *w = wt;
    return (int64_t)tile_chunk(dij, dkl, bi, bk, ii, jj, kk, ll >> 1) * 2 + (ll & 1);
}

struct mi_ctx {
    int device = 0;
    int natm = 0, nbas = 0, nao = 0, nblk = 0, npad = 0;
    std::vector<int32_t> atm, bas;
    std::vector<double> env;
    std::vector<ShellH> shells;
    // device-side basis for 1e kernel
    double *d_env = nullptr;
    int32_t *d_bas = nullptr, *d_atm = nullptr;
    int *d_shell_ao = nullptr;
    double *d_shell_xyz = nullptr;       // [nbas][3]
    double *d_c2s = nullptr;             // concatenated c2s tables l=0..LMAX
    int c2s_off[LMAX + 2];
    RysDev rys;                          // device pointers inside
    double *d_rys_cheb = nullptr, *d_herm_r = nullptr, *d_herm_w = nullptr;
    // pairs
    PairClass pc[NPC];
    double *d_prim = nullptr;            // all primitive-pair records
    double *d_M = nullptr;               // all transformation matrices
    std::vector<double> h_prim, h_M;     // host copies (gradient variants are appended lazily)
    double tol = 1e-13;
    int rank = 0, nranks = 1;
    bool grad_ready = false;
    // host half of prepare_grad_records (variant records, derivative matrices: 0.08-0.1 s for ibuprofen/def2-TZVP) on a helper
    // thread started at the end of mi_eri_prepare (`grad_prefetch`): it overlaps the SCF loop of a geometry step
    std::thread grad_worker;
    bool grad_host_ready = false;
    int grad_host_rc = 0;
    std::string grad_host_err;
    int opt_grad_prefetch = 0;
    // component index tables per class quadruple (built lazily)
    // tiles
    int64_t n_tiles = 0;
    std::vector<TileInfo> tiles;
    std::vector<int64_t> tile_off;
    std::vector<RunRec> runs;
    int32_t *d_tile_table = nullptr;     // [nbp(nbp+1)/2] -> local tile index or -1
    uint8_t *d_tile_present = nullptr;   // sharded contexts only: 1 where the slot is resident on another rank
    int64_t *d_tile_off = nullptr;
    int *d_tile_I = nullptr;
    RunRec *d_runs = nullptr;
    RunRec *d_segs = nullptr;            // runs cut at wave boundaries
    int *d_wave_seg = nullptr;           // [nwaves+1] segment range of each wave
    int n_jk_waves = 0;
    int n_jk_cached = 0;                 // leading J/K work items kept in the Infinity Cache (default-policy loads)
    double *d_tiles = nullptr;
    int64_t tile_doubles = 0, tile_alloc = 0;
    // J/K work buffers
    double *d_Dpad = nullptr, *d_Jacc = nullptr, *d_Kacc = nullptr;
    int ldp = 0;
    double *d_red = nullptr;
    mi_eri_stats stats{};
    int64_t mem_need_bytes = 0, mem_free_bytes = 0; // of the last mi_eri_prepare (also when it returned MI_ERR_NOMEM)
    bool eri_ready = false;
    // tunables (mi_set_option)
    int opt_runmax = 0;      // tiles per J/K work item (0 = auto: ntiles/2048 clamped to [8,64])
    int opt_jk_waves = 0;    // 0: one wave per work item, longest first; >0: that many waves, equal-cost shares
    int opt_jk_nt = 1;       // nontemporal loads for the tile stream
    double opt_grad_dtol = 1e-13; // gradient: skip quartets with q_ab q_cd max|G| below this (0: Schwarz only)
    int opt_xf_mfma_min = 300; // transform kernel: MFMA tiles only when the spherical block has at least this many elements
    double opt_tpq_maxprim = 32.0; // thread-per-quartet kernels only when the mean primitive quartets per shell quartet stay below this
    int opt_eri_tpq = 1;     // thread-per-quartet fused ERI kernels for the low angular classes (0: wave-per-quartet pair everywhere)
    int opt_jk_pair = -1;    // n_dm = 2: one pass with two waves per work item (-1: for stores > 16 GB, 0: one pass per density, 1: always)
    int opt_tri_tiles = 1;   // block-diagonal tiles store triangular rows (0: the full-row layout of round 1); next mi_eri_prepare
    int tri = 1;             // layout of the current store
    int opt_jk_cache_mb = 160; // MiB of tiles read with the default cache policy when the tensor exceeds the Infinity Cache (0: none)
    int opt_grad_rows = 1;   // gradient: row kernel (eri_grad_rows_kernel) for the classes it covers
    int opt_grad_rows_min = 20;  // ... when the two derivative blocks have at least this many rows
    int opt_grad_live = 1;   // gradient: wave-per-quartet launches walk the compacted list of density-screened quartets
    int opt_jk_kjlt = 0;     // J+K: K_JL reduced per tile instead of run-wide accumulators (two waves per SIMD), experiment
    int opt_jk_dpp = 1;      // per-tile reduce-scatters of the J/K kernel through DPP moves (0: ds_bpermute, the round-1/2 path)
    int opt_jk_pipe = -1;    // software-pipelined half-tile kernel for the K-carrying builds (-1: when the tensor is cache-resident)
    int opt_ao_order = 1;    // tile AO order: 1 = angular-momentum major (all s, all p, ... ; tiles become class-homogeneous), 0 = caller's
    int opt_ket_cluster = 1; // kets ordered by (block pair of the store, shell) inside Schwarz-ordered clusters instead of by q alone
    int opt_xf_mlds = 0;     // transform kernel: stage the per-pair matrices in LDS when a quartet's blocks then fit this many KB (0: never; measured 1.3-3x SLOWER at 16-96)
    int opt_rys_qpw_maxcomp = 256;   // Rys kernel: four quartets per wave for classes up to this many [e0|f0] components ...
    double opt_rys_qpw_maxprim = 4.0; // ... whose quartets have at most about this many primitive quartets (0 components: off)
    int opt_xf_qpw_max = 40;         // transform kernel: four quartets per wave up to this many spherical elements per quartet
    // MiB of the [e0|f0] hand-over buffer between the Rys and the transform launches, and of each of the two buffers of the
    // gradient.  Measured on ibuprofen/def2-TZVP (tools/eri_bench.py, ERI_OPTS): 12 / 48 / 256 / 1024 / 4096 MiB -> evaluation
    // 0.53 / 0.38 / 0.353 / 0.353 / 0.339 s, gradient (2 x half of that) 3.9 / 2.1 / 1.23 / 1.14 / 1.11 s: few large launches beat
    // many small ones (drain + ramp of latency-bound waves per launch); keeping a batch inside the Infinity Cache buys nothing
    int opt_work_mb = 2048;
    int opt_grad_work_mb = 1024;
    int opt_vmat_fold_mt = 3;   // xc_vmat_fold: 64-row tiles per workgroup (1..5)
    int opt_rys_fine = 1;    // Rys kernel: also 2 and 8 components per lane (fewer registers, more waves for the classes in between)
    int opt_eri_fused = 0;   // mid / high classes: fused Rys + transform + scatter kernel (1) instead of the two-launch pair with its hand-over buffer -- measured SLOWER (0.273 vs 0.245 s): default off
    int opt_task_table = 1;  // wave-per-quartet kernels read (bra, ket) of a task from a table written once per class pair
    int opt_prim_lds = 0;    // Rys kernel: primitive-pair records of the quartet staged in LDS
    int opt_xcd_map = 1;     // ERI kernels: consecutive task chunks stay on one XCD (its L2 merges the pieces of a line)
    int ao_order = 0;        // order of the current shells[].ao / d_perm
    int *d_perm = nullptr, *d_iperm = nullptr;   // caller AO -> tile AO and back
    std::vector<int> perm, iperm;
    int opt_vmat_xcd = 1;    // xc_vmat: XCD-aware workgroup order (tiles of one split share an XCD's L2)
    int opt_vmat_wgs = 0;    // xc_vmat: workgroups aimed at by the split over the grid points (0: 1024 = two per CU; -1: round-1 formula)
    int opt_sp2_persist = 0; // planned purification as ONE resident launch with grid barriers (1: release/acquire fences, 2: write-through
                             // stores + L2-bypassing loads) -- measured SLOWER than one launch per pass (0), see sp2_plan_persist_kernel
    unsigned *d_sp2_bar = nullptr; // [0] arrival counter (monotonic), [1] abort tag
    double *d_xt_scratch = nullptr; // xc_tail_kernel: per-workgroup partials + ticket
    unsigned sp2_bar_base = 0, sp2_tag = 0;
    int n_cu = 0;
};

static inline int pc_index(int la, int lb) { return la * (la + 1) / 2 + lb; }
static inline int ne_of(int la, int lb) { int n = 0; for (int e = la; e <= la + lb; e++) n += ncart(e); return n; }

static double gaussian_int(int n, double a) { return std::tgamma((n + 1) * 0.5) / (2.0 * std::pow(a, (n + 1) * 0.5)); }

// AO order of the ERI store ("tile order").  order 0: the caller's (shell by shell as in `bas`).  order 1: angular-momentum
// major -- all s shells, then all p, d, f (shells of one l in the caller's sequence) -- so that an 8-AO block holds shells of
// ONE l (but for three boundary blocks): every tile then belongs to one angular class, and the pieces of a 64-byte line of the
// store (4 consecutive k x one l pair) come from quartets of the SAME class, i.e. of one launch, which lets neighbouring
// lanes / workgroups complete the line together.  The permutation is internal: densities are gathered into
// tile order when they are padded, J / K / F are scattered back by the finalize kernels.
static int set_tile_order(mi_ctx *c, int order)
{
    if (c->ao_order == order) return 0;
    const int nbas = c->nbas;
    std::vector<int> seq(nbas);
    std::iota(seq.begin(), seq.end(), 0);
    if (order == 1) std::stable_sort(seq.begin(), seq.end(), [&](int a, int b) { return c->shells[a].l < c->shells[b].l; });
    c->perm.assign(std::max(c->nao, 1), 0);
    c->iperm.assign(std::max(c->nao, 1), 0);
    int pos = 0;
    for (int s_ : seq) {
        ShellH &S = c->shells[s_];
        S.ao = pos;
        for (int m = 0; m < 2 * S.l + 1; m++) { c->perm[S.ao_nat + m] = pos + m; c->iperm[pos + m] = S.ao_nat + m; }
        pos += 2 * S.l + 1;
    }
    HIPCHK(hipMemcpy(c->d_perm, c->perm.data(), sizeof(int) * c->perm.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_iperm, c->iperm.data(), sizeof(int) * c->iperm.size(), hipMemcpyHostToDevice));
    c->ao_order = order;
    return 0;
}

struct HostStash {
    std::mutex mu;
    std::vector<double> h_M, h_prim;
    std::vector<TileInfo> tiles;
    std::vector<int64_t> tile_off;
};
static HostStash g_stash[16];

extern "C" int mi_ctx_create(const int32_t *atm, int natm, const int32_t *bas, int nbas, const double *env,
                             int nenv, int device_id, mi_ctx **out)
{
    if (!atm || !bas || !env || !out || natm <= 0 || nbas <= 0) return fail("mi_ctx_create: bad arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail("mi_ctx_create: no HIP device available (this engine has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail("mi_ctx_create: device %d out of range (have %d)", device_id, ndev);
    HIPCHK(hipSetDevice(device_id));
    mi_ctx *c = new mi_ctx();
    c->device = device_id;
    c->natm = natm; c->nbas = nbas;
    c->atm.assign(atm, atm + (size_t)natm * ATM_SLOTS);
    c->bas.assign(bas, bas + (size_t)nbas * BAS_SLOTS);
    c->env.assign(env, env + nenv);
    int ao = 0;
    std::vector<int> shell_ao(nbas + 1);
    for (int i = 0; i < nbas; i++) {
        const int32_t *b = bas + (size_t)i * BAS_SLOTS;
        if (b[3] != 1) { delete c; return fail("shell %d: nctr=%d unsupported (split general contractions)", i, b[3]); }
        if (b[1] < 0 || b[1] > LMAX) { delete c; return fail("shell %d: l=%d unsupported (max %d)", i, b[1], LMAX); }
        ShellH s;
        s.atom = b[0]; s.l = b[1]; s.nprim = b[2]; s.ao = ao; s.ao_nat = ao;
        s.exps = c->env.data() + b[5]; s.coef = c->env.data() + b[6];
        const double *r = c->env.data() + atm[(size_t)s.atom * ATM_SLOTS + 1];
        s.r[0] = r[0]; s.r[1] = r[1]; s.r[2] = r[2];
        c->shells.push_back(s);
        shell_ao[i] = ao;
        ao += 2 * s.l + 1;
    }
    shell_ao[nbas] = ao;
    c->nao = ao;
    c->nblk = (ao + BLK - 1) / BLK;
    c->npad = c->nblk * BLK;
    c->ldp = c->npad + BLK;
    // device copies
    HIPCHK(dev_malloc(&c->d_env, sizeof(double) * nenv));
    HIPCHK(hipMemcpy(c->d_env, env, sizeof(double) * nenv, hipMemcpyHostToDevice));
    HIPCHK(dev_malloc(&c->d_bas, sizeof(int32_t) * nbas * BAS_SLOTS));
    HIPCHK(hipMemcpy(c->d_bas, bas, sizeof(int32_t) * nbas * BAS_SLOTS, hipMemcpyHostToDevice));
    HIPCHK(dev_malloc(&c->d_atm, sizeof(int32_t) * natm * ATM_SLOTS));
    HIPCHK(hipMemcpy(c->d_atm, atm, sizeof(int32_t) * natm * ATM_SLOTS, hipMemcpyHostToDevice));
    HIPCHK(dev_malloc(&c->d_shell_ao, sizeof(int) * (nbas + 1)));
    HIPCHK(hipMemcpy(c->d_shell_ao, shell_ao.data(), sizeof(int) * (nbas + 1), hipMemcpyHostToDevice));
    {
        std::vector<double> xyz((size_t)nbas * 3);
        for (int i = 0; i < nbas; i++)
            for (int d = 0; d < 3; d++) xyz[3 * i + d] = c->shells[i].r[d];
        HIPCHK(dev_malloc(&c->d_shell_xyz, sizeof(double) * xyz.size()));
        HIPCHK(hipMemcpy(c->d_shell_xyz, xyz.data(), sizeof(double) * xyz.size(), hipMemcpyHostToDevice));
    }
    // c2s tables
    std::vector<double> all;
    for (int l = 0; l <= LMAX; l++) {
        std::vector<double> t;
        c2s_generic(l, t);
        c->c2s_off[l] = (int)all.size();
        all.insert(all.end(), t.begin(), t.end());
    }
    c->c2s_off[LMAX + 1] = (int)all.size();
    HIPCHK(dev_malloc(&c->d_c2s, sizeof(double) * all.size()));
    HIPCHK(hipMemcpy(c->d_c2s, all.data(), sizeof(double) * all.size(), hipMemcpyHostToDevice));
    // Rys tables
    HIPCHK(dev_malloc(&c->d_rys_cheb, sizeof(double) * RYS_CHEB_SIZE));
    HIPCHK(hipMemcpy(c->d_rys_cheb, RYS_CHEB_H, sizeof(double) * RYS_CHEB_SIZE, hipMemcpyHostToDevice));
    HIPCHK(dev_malloc(&c->d_herm_r, sizeof(RYS_HERM_R_H)));
    HIPCHK(hipMemcpy(c->d_herm_r, RYS_HERM_R_H, sizeof(RYS_HERM_R_H), hipMemcpyHostToDevice));
    HIPCHK(dev_malloc(&c->d_herm_w, sizeof(RYS_HERM_W_H)));
    HIPCHK(hipMemcpy(c->d_herm_w, RYS_HERM_W_H, sizeof(RYS_HERM_W_H), hipMemcpyHostToDevice));
    c->rys.cheb = c->d_rys_cheb; c->rys.herm_r = c->d_herm_r; c->rys.herm_w = c->d_herm_w;
    for (int i = 0; i <= RYS_NMAX + 1; i++) c->rys.off[i] = RYS_OFFSET_H[i];
    for (int i = 0; i <= RYS_NMAX; i++) c->rys.nint[i] = RYS_NINT_H[i];
    // shift: RYS_OFFSET_H[n] is the offset of block n (n>=1) -- see generator: offs[0]=0 is block 1.
    for (int n = 1; n <= RYS_NMAX; n++) c->rys.off[n] = RYS_OFFSET_H[n];
    // J/K buffers
    size_t pp = (size_t)c->ldp * c->ldp;
    // two of each: the spin pair of UHF / UKS is digested by one launch (jk_tiles_pair_kernel)
    HIPCHK(dev_malloc(&c->d_Dpad, sizeof(double) * 2 * pp));
    HIPCHK(dev_malloc(&c->d_Jacc, sizeof(double) * 2 * pp));
    HIPCHK(dev_malloc(&c->d_Kacc, sizeof(double) * 2 * pp));
    HIPCHK(dev_malloc(&c->d_red, sizeof(double) * 4096));
    HIPCHK(dev_malloc(&c->d_perm, sizeof(int) * std::max(ao, 1)));
    HIPCHK(dev_malloc(&c->d_iperm, sizeof(int) * std::max(ao, 1)));
    c->ao_order = -1;
    if (set_tile_order(c, 0)) { return -1; }
    {   // large host vectors left behind by the previous context of this device (mi_ctx_destroy): capacity without page faults
        HostStash &H = g_stash[c->device & 15];
        std::lock_guard<std::mutex> g(H.mu);
        H.h_M.clear(); H.h_prim.clear(); H.tiles.clear(); H.tile_off.clear();
        c->h_M.swap(H.h_M); c->h_prim.swap(H.h_prim); c->tiles.swap(H.tiles); c->tile_off.swap(H.tile_off);
    }
    *out = c;
    return 0;
}

// The resident tile store can be ~100 GB; hipMalloc of that size costs 0.1-2 s.  A freed store is parked
// (one slot per device) and reused by the next context -- e.g. every step of a geometry optimisation.
struct TileArena { double *ptr = nullptr; int64_t doubles = 0; };
static TileArena g_arena[16];
static std::vector<int64_t> g_arena_alloc_bytes;   // size of every fresh tile-store allocation of this process

static int arena_take(int dev, int64_t need, double **out)
{
    TileArena &a = g_arena[dev & 15];
    if (a.ptr && a.doubles >= need && a.doubles <= need + need / 4 + (1 << 20)) {
        *out = a.ptr; a.ptr = nullptr; a.doubles = 0;
        return 0;
    }
    if (a.ptr) { hipFree(a.ptr); a.ptr = nullptr; a.doubles = 0; }
    if (hipMalloc((void **)out, sizeof(double) * need) != hipSuccess) {   // the pool's parked blocks may be what is missing
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> g(g_pool[dev & 15].mu); pool_flush(dev); }
        HIPCHK(hipMalloc((void **)out, sizeof(double) * need));
    }
    g_arena_alloc_bytes.push_back((int64_t)sizeof(double) * need);
    return 0;
}
// Fresh device allocations of tile stores of at least `min_bytes` made by this process (a parked store that is reused does
// not count): a geometry optimisation should show ONE large one for all its steps (the atomic-guess engines make tiny ones).
extern "C" int64_t mi_tile_store_allocations(int64_t min_bytes)
{
    int64_t n = 0;
    for (int64_t b : g_arena_alloc_bytes) n += b >= min_bytes;
    return n;
}

static void arena_give(int dev, double *p, int64_t doubles)
{
    TileArena &a = g_arena[dev & 15];
    if (a.ptr) hipFree(a.ptr);
    a.ptr = p; a.doubles = doubles;
}

// Scratch buffers that every mi_eri_prepare / mi_grad_eri of a geometry optimisation would otherwise allocate and free again
// (2 GB hand-over buffer, 2 x 1 GB for the gradient, task lists of a few hundred MB): kept per device between calls.  One slot
// per purpose; a slot that is in use (a second context working on the same device from another thread) is not shared -- the
// caller then gets a private allocation that is freed on return.  hipMalloc / hipFree of these sizes cost 30-150 ms per
// geometry step on the slower boxes of the pool.
enum { SCR_EVAL_WORK = 0, SCR_EVAL_TASKS, SCR_GRAD_WP, SCR_GRAD_WM, SCR_GRAD_TASKS, SCR_GRAD_LIVE, SCR_NSLOT };
struct ScratchSlot { void *p = nullptr; size_t bytes = 0; bool busy = false; };
static ScratchSlot g_scratch[16][SCR_NSLOT];
static std::mutex g_scratch_mu;
struct Scratch {   // RAII handle of one slot (or of a private allocation)
    int dev = 0, slot = -1;
    void *p = nullptr;
    size_t bytes = 0;
    bool cached = false;
    ~Scratch() { release(); }
    // at least `need` bytes, contents undefined; grows by 25 % when it has to (a grown cached slot stays grown)
    int ensure(int dev_, int slot_, size_t need)
    {
        if (p && bytes >= need) return 0;
        if (slot < 0) {
            dev = dev_; slot = slot_;
            std::lock_guard<std::mutex> g(g_scratch_mu);
            ScratchSlot &S = g_scratch[dev & 15][slot];
            if (!S.busy) { S.busy = true; cached = true; p = S.p; bytes = S.bytes; }
        }
        if (p && bytes >= need) return 0;
        if (p) { hipFree(p); p = nullptr; bytes = 0; }
        if (cached) { std::lock_guard<std::mutex> g(g_scratch_mu); g_scratch[dev & 15][slot].p = nullptr; g_scratch[dev & 15][slot].bytes = 0; }
        const size_t want = need + need / 4;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return fail("out of device memory for a %zu-byte scratch buffer", want); }
        bytes = want;
        if (cached) { std::lock_guard<std::mutex> g(g_scratch_mu); g_scratch[dev & 15][slot].p = p; g_scratch[dev & 15][slot].bytes = bytes; }
        return 0;
    }
    void release()
    {
        if (slot < 0) return;
        if (cached) { std::lock_guard<std::mutex> g(g_scratch_mu); g_scratch[dev & 15][slot].busy = false; }
        else if (p) hipFree(p);
        p = nullptr; bytes = 0; slot = -1; cached = false;
    }
};

extern "C" void mi_release_cache(void)
{
    {
        std::lock_guard<std::mutex> g(g_scratch_mu);
        for (auto &dv : g_scratch)
            for (auto &S : dv)
                if (!S.busy && S.p) { hipFree(S.p); S.p = nullptr; S.bytes = 0; }
    }
    for (auto &a : g_arena) { if (a.ptr) hipFree(a.ptr); a.ptr = nullptr; a.doubles = 0; }
    for (int d = 0; d < 16; d++) { std::lock_guard<std::mutex> g(g_pool[d].mu); if (g_pool[d].parked_bytes) { hipSetDevice(d); pool_flush(d); } }
}

static void free_eri(mi_ctx *c)
{
    if (c->grad_worker.joinable()) c->grad_worker.join();
    c->grad_host_ready = false;
    if (c->d_tiles) { arena_give(c->device, c->d_tiles, c->tile_alloc); c->d_tiles = nullptr; }
    for (int i = 0; i < NPC; i++) {
        if (c->pc[i].d_recs) dev_free(c->pc[i].d_recs);
        if (c->pc[i].d_q) dev_free(c->pc[i].d_q);
        c->pc[i].d_recs = nullptr; c->pc[i].d_q = nullptr;
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++) {
                if (c->pc[i].d_g_recs[o][sg]) dev_free(c->pc[i].d_g_recs[o][sg]);
                c->pc[i].d_g_recs[o][sg] = nullptr;
                c->pc[i].g_recs[o][sg].clear();
            }
        c->pc[i].recs.clear(); c->pc[i].q.clear();
    }
    void *ptrs[] = {c->d_prim, c->d_M, c->d_tile_table, c->d_tile_off, c->d_tile_I, c->d_runs, c->d_tiles, c->d_segs, c->d_wave_seg,
                    c->d_tile_present};
    for (void *p : ptrs) if (p) dev_free(p);
    c->d_tile_present = nullptr;
    c->d_prim = c->d_M = nullptr; c->d_tile_table = nullptr; c->d_tile_off = nullptr; c->d_tile_I = nullptr;
    c->d_runs = nullptr; c->d_tiles = nullptr; c->d_segs = nullptr; c->d_wave_seg = nullptr;
    c->eri_ready = false;
}

// Drop the resident tile store and the pair data of the last mi_eri_prepare (the store is parked for reuse, see TileArena).
// Used when the ranks of a sharded run agree on the direct mode although THIS rank's shard would have fitted.
extern "C" int mi_eri_release(mi_ctx *c)
{
    if (!c) return fail("mi_eri_release: null context");
    hipSetDevice(c->device);
    free_eri(c);
    return 0;
}

extern "C" void mi_ctx_destroy(mi_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    free_eri(c);
    void *ptrs[] = {c->d_env, c->d_bas, c->d_atm, c->d_shell_ao, c->d_c2s, c->d_rys_cheb, c->d_herm_r, c->d_herm_w,
                    c->d_Dpad, c->d_Jacc, c->d_Kacc, c->d_red, c->d_shell_xyz, c->d_sp2_bar, c->d_xt_scratch, c->d_perm, c->d_iperm};
    for (void *p : ptrs) if (p) dev_free(p);
    // the large host vectors of the context (pair records and gradient matrices, tile directory: several hundred MB for
    // ibuprofen/def2-TZVP) are handed to the next context on this device instead of being unmapped here (~40 ms) and faulted
    // in again there: a geometry step replaces its context
    {
        HostStash &H = g_stash[c->device & 15];
        std::lock_guard<std::mutex> g(H.mu);
        if (c->h_M.capacity() > H.h_M.capacity()) H.h_M.swap(c->h_M);
        if (c->h_prim.capacity() > H.h_prim.capacity()) H.h_prim.swap(c->h_prim);
        if (c->tiles.capacity() > H.tiles.capacity()) H.tiles.swap(c->tiles);
        if (c->tile_off.capacity() > H.tile_off.capacity()) H.tile_off.swap(c->tile_off);
    }
    delete c;
}

extern "C" int mi_ctx_nao(const mi_ctx *c) { return c ? c->nao : -1; }

extern "C" int mi_set_option(mi_ctx *c, const char *key, double value)
{
    if (!c || !key) return fail("mi_set_option: null argument");
    std::string k(key);
    if (k == "runmax") c->opt_runmax = (int)value;           // takes effect at the next mi_eri_prepare
    else if (k == "jk_waves") c->opt_jk_waves = (int)value; // takes effect at the next mi_eri_prepare
    else if (k == "jk_nt") c->opt_jk_nt = (int)value;
    else if (k == "jk_pipe") c->opt_jk_pipe = (int)value;
    else if (k == "jk_dpp") c->opt_jk_dpp = (int)value;
    else if (k == "jk_kjlt") c->opt_jk_kjlt = (int)value;
    else if (k == "grad_live") c->opt_grad_live = (int)value;
    else if (k == "grad_rows") c->opt_grad_rows = (int)value;
    else if (k == "grad_rows_min") c->opt_grad_rows_min = (int)value;
    else if (k == "grad_rows_g32") g_rows_g32 = (int)value;
    else if (k == "jk_cache_mb") c->opt_jk_cache_mb = (int)value;
    else if (k == "jk_pair") c->opt_jk_pair = (int)value;
    else if (k == "sp2_persist") c->opt_sp2_persist = (int)value;
    else if (k == "vmat_wgs") c->opt_vmat_wgs = (int)value;
    else if (k == "vmat_xcd") c->opt_vmat_xcd = (int)value;
    else if (k == "tri_tiles") c->opt_tri_tiles = (int)value;       // takes effect at the next mi_eri_prepare
    else if (k == "ao_order") c->opt_ao_order = (int)value;         // takes effect at the next mi_eri_prepare
    else if (k == "ket_cluster") c->opt_ket_cluster = (int)value;   // takes effect at the next mi_eri_prepare
    else if (k == "xcd_map") c->opt_xcd_map = (int)value;
    else if (k == "prim_lds") c->opt_prim_lds = (int)value;
    else if (k == "task_table") c->opt_task_table = (int)value;
    else if (k == "eri_fused") c->opt_eri_fused = (int)value;
    else if (k == "rys_fine") c->opt_rys_fine = (int)value;
    else if (k == "vmat_fold_mt") c->opt_vmat_fold_mt = (int)value;
    else if (k == "work_mb") c->opt_work_mb = (int)value;
    else if (k == "grad_work_mb") c->opt_grad_work_mb = (int)value;
    else if (k == "rys_qpw_maxcomp") c->opt_rys_qpw_maxcomp = (int)value;
    else if (k == "rys_qpw_maxprim") c->opt_rys_qpw_maxprim = value;
    else if (k == "xf_qpw_max") c->opt_xf_qpw_max = (int)value;
    else if (k == "xf_mlds") c->opt_xf_mlds = (int)value;
    else if (k == "eri_tpq") c->opt_eri_tpq = (int)value;
    else if (k == "tpq_maxprim") c->opt_tpq_maxprim = value;
    else if (k == "xf_mfma_min") c->opt_xf_mfma_min = (int)value;   // takes effect at the next mi_eri_prepare
    else if (k == "grad_dtol") c->opt_grad_dtol = value;
    else if (k == "grad_prefetch") c->opt_grad_prefetch = (int)value;   // takes effect at the next mi_eri_prepare
    else return fail("mi_set_option: unknown key '%s'", key);
    return 0;
}

// =================================================================================================
// One-electron integrals: one thread per shell pair (i >= j)
// =================================================================================================
#define NC1 10 /* ncart(LMAX_1E) */

__device__ inline void cart_pow(int l, int idx, int &x, int &y, int &z)
{
    int n = 0;
    for (int lx = l; lx >= 0; lx--) {
        int cnt = l - lx + 1;
        if (idx < n + cnt) { x = lx; y = l - lx - (idx - n); z = l - x - y; return; }
        n += cnt;
    }
    x = y = z = 0;
}

struct Int1eArgs {
    const int32_t *atm, *bas;
    const double *env;
    const int *shell_ao;
    const double *c2s;
    int c2s_off[LMAX + 2];
    RysDev rys;
    int natm, nbas, nao;
    double *S, *T, *V, *dip;
    double org[3];
};

// grid.y = natm + 1: slice y < natm adds the nuclear-attraction contribution of nucleus y (FP64 atomics into a zeroed
// V: natm-fold more parallelism for the only part that scales with the number of atoms), slice y == natm writes S, T
// and the dipole blocks.
__global__ __launch_bounds__(64) void int1e_kernel(Int1eArgs A)
{
    int pid = blockIdx.x * blockDim.x + threadIdx.x;
    int npair = A.nbas * (A.nbas + 1) / 2;
    if (pid >= npair) return;
    const int vatom = (int)blockIdx.y < A.natm ? (int)blockIdx.y : -1; // -1: the S/T/dipole slice
    if (vatom >= 0 && (A.V == nullptr || A.atm[vatom * ATM_SLOTS + 0] == 0)) return;
    int ish = (int)((sqrt(8.0 * pid + 1.0) - 1.0) * 0.5);
    while ((ish + 1) * (ish + 2) / 2 <= pid) ish++;
    while (ish * (ish + 1) / 2 > pid) ish--;
    int jsh = pid - ish * (ish + 1) / 2;
    const int32_t *bi = A.bas + ish * BAS_SLOTS, *bj = A.bas + jsh * BAS_SLOTS;
    int la = bi[1], lb = bj[1];
    const double *ra = A.env + A.atm[bi[0] * ATM_SLOTS + 1], *rb = A.env + A.atm[bj[0] * ATM_SLOTS + 1];
    int nca = (la + 1) * (la + 2) / 2, ncb = (lb + 1) * (lb + 2) / 2;
    // 6 cartesian blocks: S, T, V, x, y, z
    double blk[6][NC1 * NC1];
    for (int m = 0; m < 6; m++)
        for (int k = 0; k < nca * ncb; k++) blk[m][k] = 0.0;
    double AB[3] = {ra[0] - rb[0], ra[1] - rb[1], ra[2] - rb[2]};
    for (int ip = 0; ip < bi[2]; ip++)
        for (int jp = 0; jp < bj[2]; jp++) {
            double a = A.env[bi[5] + ip], b = A.env[bj[5] + jp];
            double cc = A.env[bi[6] + ip] * A.env[bj[6] + jp];
            double p = a + b, mu = a * b / p, h = 0.5 / p;
            double ex = exp(-mu * (AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2]));
            double P[3], PA[3], PB[3];
            for (int d = 0; d < 3; d++) { P[d] = (a * ra[d] + b * rb[d]) / p; PA[d] = P[d] - ra[d]; PB[d] = P[d] - rb[d]; }
            // 1-D overlaps s[d][i][j], i <= la+1, j <= lb+2
            double s[3][LMAX_1E + 2][LMAX_1E + 3];
            for (int d = 0; d < 3; d++) {
                s[d][0][0] = 1.0;
                for (int i = 0; i <= la; i++)
                    s[d][i + 1][0] = PA[d] * s[d][i][0] + (i > 0 ? i * h * s[d][i - 1][0] : 0.0);
                for (int j = 0; j <= lb + 1; j++)
                    for (int i = 0; i <= la + 1; i++)
                        s[d][i][j + 1] = PB[d] * s[d][i][j] + (i > 0 ? i * h * s[d][i - 1][j] : 0.0) + (j > 0 ? j * h * s[d][i][j - 1] : 0.0);
            }
            double pref = cc * ex * pow(M_PI / p, 1.5);
            for (int ia = 0; ia < (vatom < 0 ? nca : 0); ia++) {
                int pa[3];
                cart_pow(la, ia, pa[0], pa[1], pa[2]);
                for (int ib = 0; ib < ncb; ib++) {
                    int pb[3];
                    cart_pow(lb, ib, pb[0], pb[1], pb[2]);
                    double s1[3], t1[3], x1[3];
                    for (int d = 0; d < 3; d++) {
                        int i = pa[d], j = pb[d];
                        double sij = s[d][i][j];
                        double tij = -2.0 * b * (2 * j + 1) * sij + 4.0 * b * b * s[d][i][j + 2];
                        if (j >= 2) tij += j * (j - 1) * s[d][i][j - 2];
                        s1[d] = sij; t1[d] = -0.5 * tij;
                        x1[d] = s[d][i + 1][j] + (ra[d] - A.org[d]) * sij;
                    }
                    int k = ia * ncb + ib;
                    blk[0][k] += pref * s1[0] * s1[1] * s1[2];
                    blk[1][k] += pref * (t1[0] * s1[1] * s1[2] + s1[0] * t1[1] * s1[2] + s1[0] * s1[1] * t1[2]);
                    blk[3][k] += pref * x1[0] * s1[1] * s1[2];
                    blk[4][k] += pref * s1[0] * x1[1] * s1[2];
                    blk[5][k] += pref * s1[0] * s1[1] * x1[2];
                }
            }
            // nuclear attraction by Rys quadrature, nroots = (la+lb)/2 + 1
            int nr = (la + lb) / 2 + 1;
            double pv = cc * ex * 2.0 * M_PI / p;
            for (int ic = (vatom < 0 ? A.natm : vatom); ic < (vatom < 0 ? A.natm : vatom + 1); ic++) {
                double Z = A.atm[ic * ATM_SLOTS + 0];
                if (Z == 0.0) continue;
                const double *C = A.env + A.atm[ic * ATM_SLOTS + 1];
                double PC[3] = {P[0] - C[0], P[1] - C[1], P[2] - C[2]};
                double x = p * (PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]);
                for (int r = 0; r < nr; r++) {
                    double u = rys_eval(A.rys, nr, r, x), w = rys_eval(A.rys, nr, nr + r, x);
                    double g[3][2 * LMAX_1E + 1][LMAX_1E + 1]; // g[d][i][j]
                    double b10 = (1.0 - u) * h;
                    for (int d = 0; d < 3; d++) {
                        double c00 = PA[d] - u * PC[d];
                        g[d][0][0] = 1.0;
                        for (int i = 0; i < la + lb; i++)
                            g[d][i + 1][0] = c00 * g[d][i][0] + (i > 0 ? i * b10 * g[d][i - 1][0] : 0.0);
                        for (int j = 0; j < lb; j++)
                            for (int i = 0; i <= la + lb - j - 1; i++) g[d][i][j + 1] = g[d][i + 1][j] + AB[d] * g[d][i][j];
                    }
                    double f = -Z * pv * w;
                    for (int ia = 0; ia < nca; ia++) {
                        int pa[3];
                        cart_pow(la, ia, pa[0], pa[1], pa[2]);
                        for (int ib = 0; ib < ncb; ib++) {
                            int pb[3];
                            cart_pow(lb, ib, pb[0], pb[1], pb[2]);
                            blk[2][ia * ncb + ib] += f * g[0][pa[0]][pb[0]] * g[1][pa[1]][pb[1]] * g[2][pa[2]][pb[2]];
                        }
                    }
                }
            }
        }
    // cart -> sph and store both triangles
    const double *ca = A.c2s + A.c2s_off[la], *cb = A.c2s + A.c2s_off[lb];
    int nsa = 2 * la + 1, nsb = 2 * lb + 1;
    int ao_i = A.shell_ao[ish], ao_j = A.shell_ao[jsh];
    double *outs[6] = {A.S, A.T, A.V, A.dip, A.dip ? A.dip + (size_t)A.nao * A.nao : nullptr,
                       A.dip ? A.dip + 2 * (size_t)A.nao * A.nao : nullptr};
    for (int m = 0; m < 6; m++) {
        if (!outs[m] || (vatom >= 0) != (m == 2)) continue;
        for (int i = 0; i < nsa; i++)
            for (int j = 0; j < nsb; j++) {
                double v = 0.0;
                for (int a = 0; a < nca; a++) {
                    double t = 0.0;
                    for (int b = 0; b < ncb; b++) t += blk[m][a * ncb + b] * cb[b * nsb + j];
                    v += ca[a * nsa + i] * t;
                }
                if (m == 2) {   // one nucleus' share: accumulate (the diagonal shell pair visits (i,j) and (j,i) itself)
                    atomicAdd(&outs[m][(size_t)(ao_i + i) * A.nao + ao_j + j], v);
                    if (ish != jsh) atomicAdd(&outs[m][(size_t)(ao_j + j) * A.nao + ao_i + i], v);
                } else {
                    outs[m][(size_t)(ao_i + i) * A.nao + ao_j + j] = v;
                    outs[m][(size_t)(ao_j + j) * A.nao + ao_i + i] = v;
                }
            }
    }
}

// orbital paths (one-electron integrals, AO values on the grid, resident ERI tiles and their gradients) stop at f shells
static int check_orbital_lmax(const mi_ctx *c, const char *who)
{
    for (const ShellH &sh : c->shells)
        if (sh.l > LMAX_1E) return fail("%s: shell with l = %d (orbital shells stop at l = %d; g shells are for auxiliary contexts)", who, sh.l, LMAX_1E);
    return 0;
}

extern "C" int mi_int1e(mi_ctx *c, double *d_S, double *d_T, double *d_V, double *d_dip, const double *origin, void *stream)
{
    if (c && check_orbital_lmax(c, "mi_int1e")) return -1;
    if (!c) return fail("mi_int1e: null context");
    HIPCHK(hipSetDevice(c->device));
    Int1eArgs A;
    A.atm = c->d_atm; A.bas = c->d_bas; A.env = c->d_env; A.shell_ao = c->d_shell_ao; A.c2s = c->d_c2s;
    for (int i = 0; i <= LMAX + 1; i++) A.c2s_off[i] = c->c2s_off[i];
    A.rys = c->rys; A.natm = c->natm; A.nbas = c->nbas; A.nao = c->nao;
    A.S = d_S; A.T = d_T; A.V = d_V; A.dip = d_dip;
    for (int d = 0; d < 3; d++) A.org[d] = origin ? origin[d] : 0.0;
    int npair = c->nbas * (c->nbas + 1) / 2;
    if (d_V) HIPCHK(hipMemsetAsync(d_V, 0, sizeof(double) * (size_t)c->nao * c->nao, (hipStream_t)stream));
    hipLaunchKernelGGL(int1e_kernel, dim3((npair + 63) / 64, c->natm + 1), dim3(64), 0, (hipStream_t)stream, A);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// ERI generation, kernel 1: contracted [e0|f0] integrals by Rys quadrature.
//
// One 64-lane wave per shell quartet.  Per batch of PB primitive quartets:
//   phase R  lanes <-> (primitive quartet, root-or-weight function): Chebyshev evaluation -> LDS
//   phase A  lanes <-> (primitive quartet, root, direction): 2-D vertical recurrence table
//            T[n][m], n <= la+lb, m <= lc+ld  -> LDS
//   phase C  lanes <-> output components (e,f): sum over slots of Tx*Ty*Tz, accumulators in VGPRs
// =================================================================================================
struct TaskIdx { int ib, ik; };   // (bra, ket) of one task (get_task)
struct EriArgs {
    const PairRec *bra, *ket;
    const double *prim;
    const int64_t *prefix;   // [nbra+1] cumulative task counts
    int nbra;
    int64_t t0, ntask;       // this launch covers tasks [t0, t0+ntask)
    int la, lb, lc, ld;
    int nmax, mmax, nroots, tsz; // tsz = (nmax+1)*(mmax+1)
    int ncomp, PB;
    const uint32_t *comp;    // [ncomp] packed ix | iy<<10 | iz<<20
    double *work;            // [ntask][ncomp]
    RysDev rys;
    int diag;                // 1: Schwarz mode, ket == bra and task b -> (b,b)
    int swap;                // 1: the task's (bra index, ket index) address (ket[], bra[]) instead
    const int32_t *own_table; // non-null on a sharded context: skip quartets none of whose tiles live on this rank
    int ni, nj, nk, nl;       // spherical shell sizes (for the ownership test)
    // gradient only: density-weighted Schwarz screening  q_ab q_cd max|G| < dtol  (dmax == nullptr: off)
    const double *q_bra, *q_ket, *dmax; // Schwarz factors of the two pair lists; max |D| per shell pair [nbas][nbas]
    int nbas_d;
    double dtol, hyb;
    // host side only (kernel choice): primitive-pair statistics of the shared (bra) and the varying (ket) pair list
    double h_shared_np, h_vary_mean, h_vary_max4;
    const TaskIdx *tasks;    // non-null: (bra, ket) per task, precomputed (get_task)
    int prim_lds;            // > 0: stage the primitive-pair records of the quartet in LDS (room for this many records per quartet)
    double qtol;             // > 0: skip tasks with q_bra q_ket < qtol (kets of a surviving cluster that fail the Schwarz test themselves)
    unsigned xcd;            // > 0: XCD-aware block map with chunks of this many blocks (xcd_block), grid rounded up to 8 * xcd
};

// Block index -> position in the task order such that `C` consecutive positions run on ONE XCD (blocks b and b + 8 share an
// XCD under the observed round-robin placement; speed only, never correctness): the pieces of a tile line written by
// neighbouring tasks then meet in one L2.  Chunks alternate over the XCDs, so the load stays balanced along the task list.
// The grid must be a multiple of 8 C (eri_grid); positions beyond the task count idle.
__device__ inline unsigned xcd_block(unsigned b, unsigned C)
{
    if (C == 0u) return b;
    const unsigned x = b & 7u, r = b >> 3;
    return ((r / C) * 8u + x) * C + (r % C);
}
static inline unsigned eri_grid(int64_t nblocks, unsigned C)
{
    if (C == 0u) return (unsigned)nblocks;
    const int64_t m = 8 * (int64_t)C;
    return (unsigned)((nblocks + m - 1) / m * m);
}

// Upper bound of |G| = |D_ab D_cd - hyb/4 (D_ac D_bd + D_ad D_bc)| over the AO quadruples of a shell quartet.
__device__ inline double quartet_density_bound(const double *dmax, int nb, int a, int b, int c, int d, double hyb)
{
    double coul = dmax[a * nb + b] * dmax[c * nb + d];
    double exch = dmax[a * nb + c] * dmax[b * nb + d] + dmax[a * nb + d] * dmax[b * nb + c];
    return coul + 0.25 * hyb * exch;
}

// Task t of a class pair -> (bra pair index, ket index).  `prefix` holds the nbra+1 cumulative task counts followed by
// a coarse index (one entry per 1024 tasks: the bra index of task g*1024, see append_coarse_index), so the search
// touches two neighbouring coarse entries and then only the few prefix entries between them.
__device__ inline void find_task(const int64_t *prefix, int nbra, int64_t t, int &ib, int &ik)
{
    const int64_t *coarse = prefix + nbra + 1;
    const int64_t g = t >> 10;
    int lo = (int)coarse[g], hi = min((int)coarse[g + 1] + 1, nbra); // prefix[lo] <= t < prefix[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (prefix[mid] <= t) lo = mid; else hi = mid;
    }
    ib = lo;
    ik = (int)(t - prefix[lo]);
}

// (bra, ket) of every task of a class pair, written once (fill_tasks_kernel) when several launches walk the same task list --
// the Rys and transform launches of the evaluation, the 3 permutations x 3 launches of the gradient: one coalesced 8-byte load
// per quartet instead of the two or three DEPENDENT global loads of find_task (~2 us of a ~10 us quartet in those latency-
// bound kernels).
__device__ inline void get_task(const TaskIdx *tasks, const int64_t *prefix, int nbra, int64_t t, int &ib, int &ik)
{
    if (tasks) { const TaskIdx q = tasks[t]; ib = q.ib; ik = q.ik; }
    else find_task(prefix, nbra, t, ib, ik);
}
__global__ __launch_bounds__(256) void fill_tasks_kernel(const int64_t *prefix, int nbra, int64_t ntask, TaskIdx *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntask) return;
    int ib, ik;
    find_task(prefix, nbra, t, ib, ik);
    out[t] = TaskIdx{ib, ik};
}

// Gradient, density-weighted screening: the (bra, ket) list of the LIVE quartets of a class pair, in task order (round 3).  The
// bound is symmetric in the roles of the four shells, so one list serves the three permutations x three launches of the class:
// before, every launch started a wave per Schwarz-surviving quartet and the dead ones (q_ab q_cd max|G| < grad_dtol) left after
// decoding their task and reading two pair records, two Schwarz factors and six density maxima.
struct LiveArgs {
    const PairRec *bra, *ket;
    const int64_t *prefix;
    int nbra;
    int64_t ntask;
    const double *q_bra, *q_ket, *dmax;
    int nbas_d;
    double hyb, dtol;
};
__device__ inline bool task_is_live(const LiveArgs &A, int64_t t, int &ib, int &ik)
{
    find_task(A.prefix, A.nbra, t, ib, ik);
    const int a = A.bra[ib].sh_i, b = A.bra[ib].sh_j, c = A.ket[ik].sh_i, d = A.ket[ik].sh_j;
    return !(A.q_bra[ib] * A.q_ket[ik] * quartet_density_bound(A.dmax, A.nbas_d, a, b, c, d, A.hyb) < A.dtol);
}
__global__ __launch_bounds__(256) void live_count_kernel(LiveArgs A, int *block_counts)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int ib, ik;
    const bool live = t < A.ntask && task_is_live(A, t, ib, ik);
    const int n = __syncthreads_count(live);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = n;
}
// exclusive prefix sum of `n` block counts by ONE workgroup (n <= a few 1e5); offsets[n] = total
__global__ __launch_bounds__(1024) void live_scan_kernel(const int *counts, int n, int64_t *offsets)
{
    __shared__ int64_t part[1024];
    __shared__ int64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int idx = base + threadIdx.x;
        const int64_t v = idx < n ? counts[idx] : 0;
        part[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {      // Hillis-Steele inclusive scan
            const int64_t add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (idx < n) offsets[idx] = carry + part[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}
__global__ __launch_bounds__(256) void live_fill_kernel(LiveArgs A, const int64_t *offsets, TaskIdx *out)
{
    __shared__ int wave_base[4];
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int ib = 0, ik = 0;
    const bool live = t < A.ntask && task_is_live(A, t, ib, ik);
    const unsigned long long m = __ballot(live);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_base[wave] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; w++) base += wave_base[w];
    if (live) out[offsets[blockIdx.x] + base + __popcll(m & ((1ull << lane) - 1ull))] = TaskIdx{ib, ik};
}

// Does any AO quadruple of the shell quartet land in a tile that is resident on this rank?  (<= 16 block
// combinations; every symmetry image of a quadruple maps to the same canonical tile.)
__device__ inline bool quartet_has_resident_tile(const int32_t *table, int ao_i, int ni, int ao_j, int nj, int ao_k, int nk,
                                                 int ao_l, int nl, int lane, int gsz = 64, int grp = 0)
{
    // `lane` is the lane inside a group of `gsz` (>= 16) lanes working on this quartet; groups vote separately
    bool hit = false;
    if (lane < 16) {
        int I = (lane & 1) ? (ao_i + ni - 1) >> 3 : ao_i >> 3;
        int J = (lane & 2) ? (ao_j + nj - 1) >> 3 : ao_j >> 3;
        int K = (lane & 4) ? (ao_k + nk - 1) >> 3 : ao_k >> 3;
        int L = (lane & 8) ? (ao_l + nl - 1) >> 3 : ao_l >> 3;
        int hi = max(I, J), lo = min(I, J), bij = hi * (hi + 1) / 2 + lo;
        hi = max(K, L); lo = min(K, L);
        int bkl = hi * (hi + 1) / 2 + lo;
        int bmax = max(bij, bkl), bmin = min(bij, bkl);
        hit = table[(size_t)bmax * (bmax + 1) / 2 + bmin] >= 0;
    }
    const unsigned long long votes = __ballot(hit);
    const unsigned long long gmask = (gsz >= 64 ? ~0ull : ((1ull << gsz) - 1ull) << (gsz * grp));
    return (votes & gmask) != 0ull;
}

// GSZ = lanes per shell quartet: 64, or 16 (four quartets per wave) for the low angular classes, whose handful of
// components and roots leave a 64-lane wave idle while it waits on its chain of dependent loads.
template <int MAXC, int GSZ>
__global__ __launch_bounds__(64) void eri_rys_kernel(EriArgs A)
{
    extern __shared__ double lds_all[];
    constexpr int QPW = 64 / GSZ;
    const int grp = QPW == 1 ? 0 : threadIdx.x / GSZ, lane = QPW == 1 ? threadIdx.x : threadIdx.x % GSZ;
    const int tl = (int)xcd_block(blockIdx.x, A.xcd) * QPW + grp; // task inside this launch
    if (QPW == 1 && tl >= A.ntask) return;   // (grid rounded up for the XCD map)
    bool live = QPW == 1 || tl < A.ntask;
    const int64_t task = A.t0 + (live ? tl : 0);
    int ib, ik;
    if (A.diag) { ib = (int)task; ik = ib; }
    else get_task(A.tasks, A.prefix, A.nbra, task, ib, ik);
    if (A.swap) { int t_ = ib; ib = ik; ik = t_; }
    const PairRec ab = A.bra[ib], cd = A.ket[ik];
    if (A.qtol > 0.0 && A.q_bra[ib] * A.q_ket[ik] < A.qtol) { if (QPW == 1) return; live = false; }
    if (QPW == 1) { // one quartet per wave: negligible / non-resident quartets leave at once
        if (A.own_table && !quartet_has_resident_tile(A.own_table, ab.ao_i, A.ni, ab.ao_j, A.nj, cd.ao_i, A.nk, cd.ao_j, A.nl, lane)) return;
        if (A.dmax && A.q_bra[ib] * A.q_ket[ik] * quartet_density_bound(A.dmax, A.nbas_d, ab.sh_i, ab.sh_j, cd.sh_i, cd.sh_j, A.hyb) < A.dtol) return;
    } else {
        if (A.own_table)
            live = quartet_has_resident_tile(A.own_table, ab.ao_i, A.ni, ab.ao_j, A.nj, cd.ao_i, A.nk, cd.ao_j, A.nl, lane, GSZ, grp) && live;
        if (A.dmax && A.q_bra[ib] * A.q_ket[ik] * quartet_density_bound(A.dmax, A.nbas_d, ab.sh_i, ab.sh_j, cd.sh_i, cd.sh_j, A.hyb) < A.dtol) live = false;
    }
    const int n = A.nroots, tsz = A.tsz, M1 = A.mmax + 1;
    const int ncd = cd.nprim, nPQ = live ? ab.nprim * ncd : 0;
    const int PB = A.PB;
    const size_t lds_per = (size_t)PB * n * 3 * tsz + (size_t)PB * 2 * n + (size_t)A.prim_lds * 8;
    double *lds = QPW == 1 ? lds_all : lds_all + (size_t)grp * lds_per;
    double *T0 = lds;                       // [PB*n][3][tsz]
    double *rw = lds + (size_t)PB * n * 3 * tsz; // [PB][2n]
    // primitive-pair records {p, P, P-A, K} of the bra and the ket pair: staged once per quartet in LDS (coalesced 64-byte
    // records) so that the R and A phases of every batch read them on-chip instead of paying a global round trip each
    const double *prim_b = A.prim + (size_t)ab.prim_off * 8, *prim_k = A.prim + (size_t)cd.prim_off * 8;
    if (A.prim_lds > 0) {
        double *pl = rw + (size_t)PB * 2 * n;
        const int nb8 = ab.nprim * 8, nk8 = ncd * 8;
        if (live && ab.nprim + ncd <= A.prim_lds) {
            for (int x = lane; x < nb8; x += GSZ) pl[x] = prim_b[x];
            for (int x = lane; x < nk8; x += GSZ) pl[nb8 + x] = prim_k[x];
            prim_b = pl; prim_k = pl + nb8;
        }
        __syncthreads();
    }
    double *wout = A.work + (size_t)tl * A.ncomp;
    int nPQ_all = nPQ; // uniform trip count over the quartets sharing this wave (the barriers below sit in the loop)
    if (QPW > 1)
        for (int o = GSZ; o < 64; o <<= 1) nPQ_all = max(nPQ_all, __shfl_xor(nPQ_all, o));

    for (int c0 = 0; c0 < A.ncomp; c0 += GSZ * MAXC) {
        double acc[MAXC];
        uint32_t idx[MAXC];
#pragma unroll
        for (int ci = 0; ci < MAXC; ci++) {
            acc[ci] = 0.0;
            int c = c0 + ci * GSZ + lane;
            idx[ci] = (c < A.ncomp) ? A.comp[c] : 0xFFFFFFFFu;
        }
        for (int pq0 = 0; pq0 < nPQ_all; pq0 += PB) {
            const int npq = QPW == 1 ? min(PB, nPQ - pq0) : max(0, min(PB, nPQ - pq0));
            // ---- phase R
            if (lane < npq * 2 * n) {
                int pql = lane / (2 * n), f = lane - pql * 2 * n;
                int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
                const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
                double p = b[0], q = k[0];
                double dx = b[1] - k[1], dy = b[2] - k[2], dz = b[3] - k[3];
                double x = p * q / (p + q) * (dx * dx + dy * dy + dz * dz);
                rw[pql * 2 * n + f] = rys_eval(A.rys, n, f, x);
            }
            __syncthreads();
            // ---- phase A
            if (lane < npq * n * 3) {
                int pql = lane / (3 * n), rem = lane - pql * 3 * n, r = rem / 3, d = rem - r * 3;
                int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
                const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
                double p = b[0], q = k[0], pq1 = 1.0 / (p + q);
                double u = rw[pql * 2 * n + r];
                double PQd = b[1 + d] - k[1 + d];
                double b00 = 0.5 * u * pq1;
                double b10 = 0.5 / p * (1.0 - u * q * pq1);
                double b01 = 0.5 / q * (1.0 - u * p * pq1);
                double c00 = b[4 + d] - u * q * pq1 * PQd;
                double c01 = k[4 + d] + u * p * pq1 * PQd;
                double *T = T0 + ((size_t)(pql * n + r) * 3 + d) * tsz;
                double t00 = 1.0;
                if (d == 2) {
                    double w = rw[pql * 2 * n + n + r];
                    t00 = w * b[7] * k[7] * 34.986836655249725 /* 2 pi^2.5 */ * pq1 * sqrt(p + q) / (p * q) ;
                    // 2 pi^(5/2) / (p q sqrt(p+q)) = 2 pi^2.5 * sqrt(p+q)/(p q (p+q))
                }
                T[0] = t00;
                double tm = 0.0, tc = t00;
                for (int i = 0; i < A.nmax; i++) {
                    double tn = c00 * tc + i * b10 * tm;
                    T[(i + 1) * M1] = tn;
                    tm = tc; tc = tn;
                }
                for (int m = 0; m < A.mmax; m++)
                    for (int i = 0; i <= A.nmax; i++) {
                        double v = c01 * T[i * M1 + m];
                        if (m > 0) v += m * b01 * T[i * M1 + m - 1];
                        if (i > 0) v += i * b00 * T[(i - 1) * M1 + m];
                        T[i * M1 + m + 1] = v;
                    }
            }
            __syncthreads();
            // ---- phase C
            const int nslot = npq * n;
            for (int s = 0; s < nslot; s++) {
                const double *Tx = T0 + (size_t)s * 3 * tsz, *Ty = Tx + tsz, *Tz = Ty + tsz;
#pragma unroll
                for (int ci = 0; ci < MAXC; ci++) {
                    uint32_t w = idx[ci];
                    if (w != 0xFFFFFFFFu) acc[ci] += Tx[w & 1023u] * Ty[(w >> 10) & 1023u] * Tz[(w >> 20) & 1023u];
                }
            }
            __syncthreads();
        }
        if (QPW == 1 || live) {
#pragma unroll
            for (int ci = 0; ci < MAXC; ci++) {
                int c = c0 + ci * GSZ + lane;
                if (c < A.ncomp) wout[c] = acc[ci];
            }
        }
    }
}

// The Rys phases of eri_rys_kernel for ONE quartet on ONE wave with the contracted [e0|f0] block written to LDS instead of the
// hand-over buffer: the first half of the fused evaluation kernel (eri_fused_kernel).  `scratch` holds the recurrence tables and
// the roots / weights of a batch ([PB n][3][tsz] + [PB][2 n] doubles), `out` the block ([ncomp] doubles, both in LDS).
template <int MAXC>
__device__ __forceinline__ void rys_core(const EriArgs &A, const PairRec &ab, const PairRec &cd, const int lane, double *scratch, double *out)
{
    const int n = A.nroots, tsz = A.tsz, M1 = A.mmax + 1;
    const int ncd = cd.nprim, nPQ = ab.nprim * ncd;
    const int PB = A.PB;
    double *T0 = scratch;                              // [PB*n][3][tsz]
    double *rw = scratch + (size_t)PB * n * 3 * tsz;   // [PB][2n]
    const double *prim_b = A.prim + (size_t)ab.prim_off * 8, *prim_k = A.prim + (size_t)cd.prim_off * 8;
    for (int c0 = 0; c0 < A.ncomp; c0 += 64 * MAXC) {
        double acc[MAXC];
        uint32_t idx[MAXC];
#pragma unroll
        for (int ci = 0; ci < MAXC; ci++) {
            acc[ci] = 0.0;
            const int c = c0 + ci * 64 + lane;
            idx[ci] = (c < A.ncomp) ? A.comp[c] : 0xFFFFFFFFu;
        }
        for (int pq0 = 0; pq0 < nPQ; pq0 += PB) {
            const int npq = min(PB, nPQ - pq0);
            if (lane < npq * 2 * n) {                  // roots and weights
                const int pql = lane / (2 * n), f = lane - pql * 2 * n;
                const int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
                const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
                const double p = b[0], q = k[0];
                const double dx = b[1] - k[1], dy = b[2] - k[2], dz = b[3] - k[3];
                rw[pql * 2 * n + f] = rys_eval(A.rys, n, f, p * q / (p + q) * (dx * dx + dy * dy + dz * dz));
            }
            __syncthreads();
            if (lane < npq * n * 3) {                  // 2-D recurrence tables
                const int pql = lane / (3 * n), rem = lane - pql * 3 * n, r = rem / 3, d = rem - r * 3;
                const int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
                const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
                const double p = b[0], q = k[0], pq1 = 1.0 / (p + q);
                const double u = rw[pql * 2 * n + r];
                const double PQd = b[1 + d] - k[1 + d];
                const double b00 = 0.5 * u * pq1, b10 = 0.5 / p * (1.0 - u * q * pq1), b01 = 0.5 / q * (1.0 - u * p * pq1);
                const double c00 = b[4 + d] - u * q * pq1 * PQd, c01 = k[4 + d] + u * p * pq1 * PQd;
                double *T = T0 + ((size_t)(pql * n + r) * 3 + d) * tsz;
                double t00 = 1.0;
                if (d == 2) t00 = rw[pql * 2 * n + n + r] * b[7] * k[7] * 34.986836655249725 /* 2 pi^2.5 */ * pq1 * sqrt(p + q) / (p * q);
                T[0] = t00;
                double tm = 0.0, tc = t00;
                for (int i = 0; i < A.nmax; i++) {
                    const double tn = c00 * tc + i * b10 * tm;
                    T[(i + 1) * M1] = tn;
                    tm = tc; tc = tn;
                }
                for (int m = 0; m < A.mmax; m++)
                    for (int i = 0; i <= A.nmax; i++) {
                        double v = c01 * T[i * M1 + m];
                        if (m > 0) v += m * b01 * T[i * M1 + m - 1];
                        if (i > 0) v += i * b00 * T[(i - 1) * M1 + m];
                        T[i * M1 + m + 1] = v;
                    }
            }
            __syncthreads();
            const int nslot = npq * n;                 // component products
            for (int s_ = 0; s_ < nslot; s_++) {
                const double *Tx = T0 + (size_t)s_ * 3 * tsz, *Ty = Tx + tsz, *Tz = Ty + tsz;
#pragma unroll
                for (int ci = 0; ci < MAXC; ci++) {
                    const uint32_t w = idx[ci];
                    if (w != 0xFFFFFFFFu) acc[ci] += Tx[w & 1023u] * Ty[(w >> 10) & 1023u] * Tz[(w >> 20) & 1023u];
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int ci = 0; ci < MAXC; ci++) {
            const int c = c0 + ci * 64 + lane;
            if (c < A.ncomp) out[c] = acc[ci];
        }
    }
}

typedef double d4_t __attribute__((ext_vector_type(4)));

// One 16x16 tile of C = A B on a single wave with v_mfma_f64_16x16x4_f64 (used by the per-quartet transformation
// kernels, whose matrices are a few tens of rows/columns: one operand load per 16 FMAs instead of two per FMA).
//   A(m,k) = A[m*sam + k*sak] for m < M, B(k,n) = B[k*sbk + n*sbn] for n < N, k < K; out-of-range elements read as 0.
// Result layout: element r of the return value is C[m0 + (lane>>4) + 4r][n0 + (lane&15)].
// Operand loads go out in chunks of WAVE_MFMA_CHUNK k steps before the chunk's MFMAs: the A / B operands of the per-quartet
// transforms come from global memory (per-pair matrices) or LDS, and a load -> wait -> MFMA chain per k step made the kernels
// latency bound (one ~1 us round trip per MFMA: 200 us for an (ff|fd) quartet).
#define WAVE_MFMA_CHUNK 4
__device__ inline d4_t wave_mfma_tile(const double *A, int sam, int sak, int M, const double *B, int sbk, int sbn, int N, int K, int m0,
                                      int n0, int lane)
{
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    const int m = m0 + (lane & 15), n = n0 + (lane & 15), kq = lane >> 4;
    const bool mv = m < M, nv = n < N;
    const double *pa = A + (size_t)(mv ? m : 0) * sam, *pb = B + (size_t)(nv ? n : 0) * sbn;
    for (int k0 = 0; k0 < K; k0 += 4 * WAVE_MFMA_CHUNK) {
        double av[WAVE_MFMA_CHUNK], bv[WAVE_MFMA_CHUNK];
#pragma unroll
        for (int s_ = 0; s_ < WAVE_MFMA_CHUNK; s_++) {     // the chunk's operand loads go out together: one round trip per 16 k
            const int k = k0 + 4 * s_ + kq;
            const bool kv = k < K;
            av[s_] = (mv && kv) ? pa[(size_t)k * sak] : 0.0;
            bv[s_] = (nv && kv) ? pb[(size_t)k * sbk] : 0.0;
        }
#pragma unroll
        for (int s_ = 0; s_ < WAVE_MFMA_CHUNK; s_++)
            if (k0 + 4 * s_ < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s_], bv[s_], acc, 0, 0, 0);
    }
    return acc;
}
// MFMA pays when the zero padding of the 16x16x4 tiles wastes less than ~3/4 of the instruction
__device__ __host__ inline bool mfma_worthwhile(int M, int N, int K)
{
    const int pm = (M + 15) / 16 * 16, pn = (N + 15) / 16 * 16, pk = (K + 3) / 4 * 4;
    return 4 * M * N * K >= pm * pn * pk;
}

// =================================================================================================
// ERI generation, kernel 2: out[ab][cd] = M_ab . E0 . M_cd^T (HRR + cart->sph folded into M), then
// scatter every symmetry image that lands in a canonical resident tile.  One wave per quartet.
// Tile element (i,j,k,l) lives at  off + (((j*4 + l/2) * (bi*bk) + i*bk + k) * 2 + (l&1)).
// =================================================================================================
struct XfArgs {
    const PairRec *bra, *ket;
    const double *Mbuf;
    const int64_t *prefix;
    int nbra;
    int64_t t0, ntask;       // this launch covers tasks [t0, t0+ntask)
    int ne, nf, nsab, nscd, nsb, nsd;
    const double *work;
    int ncomp;
    const int32_t *tile_table;
    const int64_t *tile_off;
    double *tiles;
    int nao;
    int check_owner, ni, nj, nk, nl;
    // density fitting (mi_df_build): dense output instead of the tile scatter.  dense_mode 1: (ij|P) -> dense_out[i][j][P]
    // and [j][i][P] (ld = dense_n auxiliary functions); 2: (P|Q) -> dense_out[P][Q] and [Q][P].  The fitted function of a
    // "pair" (P, unit s) is its first shell.
    double *dense_out;
    int dense_mode, dense_n;
    int tri;                 // triangular rows in block-diagonal tiles (tile geometry)
    const double *q_bra, *q_ket; // with qtol > 0: the same per-task Schwarz rejection as the Rys kernel made
    double qtol;
    unsigned xcd;            // XCD-aware block map (xcd_block)
    const TaskIdx *tasks;    // non-null: (bra, ket) per task, precomputed (get_task)
    int m_lds;               // 1: the two per-pair transformation matrices are staged in LDS with the E0 block (one memory round trip
                             // for all operands; a global load per k step made the products a chain of dependent round trips)
};

// Read-only, wave-uniform operands (work-item records, tile directory, the J-L density rows) go through the
// constant address space so that they are fetched with scalar loads into SGPRs instead of per-lane vector loads.
#define MI_CONST_AS __attribute__((address_space(4)))
template <class T> __device__ inline const MI_CONST_AS T *as_const(const T *p)
{
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const MI_CONST_AS T *)p;
#pragma clang diagnostic pop
}

__device__ inline void put_tile(const XfArgs &A, int i, int j, int k, int l, double v)
{
    int I = i >> 3, J = j >> 3, K = k >> 3, L = l >> 3;
    if (I < J || K < L) return;
    int bij = I * (I + 1) / 2 + J, bkl = K * (K + 1) / 2 + L;
    if (bij < bkl) return;
    // directory look-ups through the constant address space: known not to alias the tile stores, so the compiler may
    // overlap the two dependent loads of one element with those of the next instead of serialising load-load-store chains
    int32_t t = as_const(A.tile_table)[(size_t)bij * (bij + 1) / 2 + bkl];
    if (t < 0) return;
    int bi = min(BLK, A.nao - I * BLK), bk = min(BLK, A.nao - K * BLK);
    double w;
    const int64_t e = tile_elem(A.tri != 0, I == J, K == L, bij == bkl, bi, bk, i & 7, j & 7, k & 7, l & 7, &w);
    if (e >= 0) A.tiles[as_const(A.tile_off)[t] + e] = w * v;
}

// MFMA = true: instantiation with the matrix-core paths for the large angular classes; MFMA = false: lean kernel for
// the small classes that make up most quartets.  GSZ = lanes per quartet (64, or 16 = four quartets per wave, as
// in eri_rys_kernel).  (Four waves sharing a quartet's LDS blocks were measured slower: 0.50 vs 0.46 s for
// ibuprofen/def2-TZVP.)
// element (i,j,k,l) of one symmetry image whose tile base (-1: no resident canonical tile) is already known
__device__ inline void put_tile_at(const XfArgs &A, int64_t base, int i, int j, int k, int l, double v)
{
    if (base < 0) return;
    const int I = i >> 3, J = j >> 3, K = k >> 3, L = l >> 3;
    const int bij = I * (I + 1) / 2 + J, bkl = K * (K + 1) / 2 + L;
    const int bi = min(BLK, A.nao - I * BLK), bk = min(BLK, A.nao - K * BLK);
    double w;
    const int64_t e = tile_elem(A.tri != 0, I == J, K == L, bij == bkl, bi, bk, i & 7, j & 7, k & 7, l & 7, &w);
    if (e >= 0) A.tiles[base + e] = w * v;
}

// FMAXC = 0: the [e0|f0] block comes from the hand-over buffer (A.work) of a preceding eri_rys_kernel launch.
// FMAXC > 0 (GSZ = 64 only): FUSED evaluation -- the Rys phases (rys_core<FMAXC>, arguments *R) run first in this wave and leave
// the block in LDS: no hand-over buffer (51.8 of the 139.8 GB written by an evaluation of ibuprofen/def2-TZVP), one launch
// and one task look-up per quartet instead of two.
template <bool MFMA, int GSZ, int FMAXC>
__device__ __forceinline__ void xf_body(const XfArgs &A, const EriArgs *R)
{
    extern __shared__ double lds_all[];
    constexpr int QPW = 64 / GSZ;
    __shared__ int64_t tbase_all[QPW][8][16];   // tile base per (symmetry image, 2x2x2x2 sub-block of the quartet's AO ranges)
    const int grp = threadIdx.x / GSZ, lane = threadIdx.x % GSZ;
    const int64_t tl = (int64_t)xcd_block(blockIdx.x, A.xcd) * QPW + grp;
    bool live = tl < A.ntask;
    int ib, ik;
    get_task(A.tasks, A.prefix, A.nbra, A.t0 + (live ? tl : 0), ib, ik);
    const PairRec ab = A.bra[ib], cd = A.ket[ik];
    if (A.qtol > 0.0 && A.q_bra[ib] * A.q_ket[ik] < A.qtol) live = false;
    if (A.check_owner)
        live = quartet_has_resident_tile(A.tile_table, ab.ao_i, A.ni, ab.ao_j, A.nj, cd.ao_i, A.nk, cd.ao_j, A.nl, lane, GSZ, grp) && live;
    if (QPW == 1 && !live) return;
    const double *E0g = FMAXC > 0 ? nullptr : A.work + (size_t)tl * A.ncomp;
    const size_t lds_m = A.m_lds ? (size_t)2 * A.nsab * A.ne + (size_t)2 * A.nscd * A.nf : 0;
    double *E0 = lds_all + (size_t)grp * ((size_t)A.ne * A.nf + (size_t)A.nsab * A.nf + lds_m); // [ne][nf]
    double *X = E0 + A.ne * A.nf;                                                       // [nsab][nf]
    // (M_ab / M_cd^T staged in LDS as well was measured SLOWER -- ibuprofen/def2-TZVP 0.325 -> 0.342 s: this kernel is bound by
    // its scattered tile stores, see DESIGN.md 3.2, and the larger LDS footprint only costs occupancy)
    const double *Mab = A.Mbuf + ab.m_off, *Mcd = A.Mbuf + cd.m_off;
    if (A.m_lds) {   // M and M^T of both pairs, as stored (2 nsab ne and 2 nscd nf doubles)
        double *ML = X + (size_t)A.nsab * A.nf;
        const int na = 2 * A.nsab * A.ne, nc = 2 * A.nscd * A.nf;
        if (live) {
            for (int c = lane; c < na; c += GSZ) ML[c] = Mab[c];
            for (int c = lane; c < nc; c += GSZ) ML[na + c] = Mcd[c];
        }
        Mab = ML; Mcd = ML + na;
    }
    const double *MabT = Mab + A.nsab * A.ne, *McdT = Mcd + A.nscd * A.nf;   // [e][r], [f][c]
    if (FMAXC > 0) rys_core<(FMAXC > 0 ? FMAXC : 1)>(*R, ab, cd, lane, X /* tables over the not yet used X block */, E0);
    else if (live)
        for (int c = lane; c < A.ne * A.nf; c += GSZ) E0[c] = E0g[c];
    // symmetry images: role of (a0,a1,a2,a3) = (ab.i, ab.j, cd.i, cd.j) in (i,j,k,l); bit 0 swaps the bra pair, bit 1 the ket
    // pair, bit 2 exchanges bra and ket.  Which images can land in a canonical tile (I >= J, K >= L) at all is decided once per
    // quartet from the block ranges of the four shells, and so is the tile of every sub-block: the directory look-ups (two
    // dependent global loads) happen 128 times per quartet in parallel instead of once per element and image in sequence.
    const int nsh[4] = {A.nsab / A.nsb, A.nsb, A.nscd / A.nsd, A.nsd};
    const int aos[4] = {ab.ao_i, ab.ao_j, cd.ao_i, cd.ao_j};
    int lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { lo[q] = aos[q] >> 3; hi[q] = (aos[q] + nsh[q] - 1) >> 3; }
    auto ok = [&](int a, int b, int c, int d) { return hi[a] >= lo[b] && hi[c] >= lo[d]; };
    const unsigned mask = (ok(0, 1, 2, 3) ? 1u : 0u) | (ok(1, 0, 2, 3) ? 2u : 0u) | (ok(0, 1, 3, 2) ? 4u : 0u) | (ok(1, 0, 3, 2) ? 8u : 0u) |
                          (ok(2, 3, 0, 1) ? 16u : 0u) | (ok(3, 2, 0, 1) ? 32u : 0u) | (ok(2, 3, 1, 0) ? 64u : 0u) | (ok(3, 2, 1, 0) ? 128u : 0u);
    int64_t (*tbase)[16] = tbase_all[grp];
    if (live && !A.dense_mode) {
        for (int idx = lane; idx < 128; idx += GSZ) {
            const int img = idx >> 4;
            int64_t base = -1;
            if (mask & (1u << img)) {
                // blocks of the four shells in their own order, then permuted like the indices of this image
                int blk[4];
#pragma unroll
                for (int q = 0; q < 4; q++) blk[q] = lo[q] + ((idx >> q) & 1);
                const bool valid = blk[0] <= hi[0] && blk[1] <= hi[1] && blk[2] <= hi[2] && blk[3] <= hi[3];
                // emit() numbers the exchanged images (l,k,i,j) = 5 and (k,l,j,i) = 6: pair swaps are applied AFTER the exchange there
                const int f = img == 5 ? 6 : (img == 6 ? 5 : img);
                int r0 = (f & 1) ? 1 : 0, r1 = (f & 1) ? 0 : 1, r2 = (f & 2) ? 3 : 2, r3 = (f & 2) ? 2 : 3;
                if (f & 4) { int t_ = r0; r0 = r2; r2 = t_; t_ = r1; r1 = r3; r3 = t_; }
                const int I = blk[r0], J = blk[r1], K = blk[r2], L = blk[r3];
                if (valid && I >= J && K >= L) {
                    const int bij = I * (I + 1) / 2 + J, bkl = K * (K + 1) / 2 + L;
                    if (bij >= bkl) {
                        const int32_t t = A.tile_table[(size_t)bij * (bij + 1) / 2 + bkl];
                        if (t >= 0) base = A.tile_off[t];
                    }
                }
            }
            tbase[img][idx & 15] = base;
        }
    }
    __syncthreads();
    // Result block O[nsab][nscd] goes to LDS first (over E0, which is dead by then), then leaves for the tiles in ADDRESS order.
    double *O = E0;
    auto emit_dense = [&](int r, int c, double s) {   // density fitting: dense (ij|P) / (P|Q) output
        int sa = r / A.nsb, sb = r - sa * A.nsb, sc = c / A.nsd;
        int i = ab.ao_i + sa, j = ab.ao_j + sb, k = cd.ao_i + sc;
        if (A.dense_mode == 1) {
            A.dense_out[((size_t)i * A.nao + j) * A.dense_n + k] = s;
            A.dense_out[((size_t)j * A.nao + i) * A.dense_n + k] = s;
        } else {
            A.dense_out[(size_t)i * A.dense_n + k] = s;
            A.dense_out[(size_t)k * A.dense_n + i] = s;
        }
    };
    // X = Mab E0, out = X Mcd^T: FP64 MFMA tiles for the large angular classes, per-lane dot products otherwise
    if (MFMA && GSZ == 64 && mfma_worthwhile(A.nsab, A.nf, A.ne)) {
        for (int m0 = 0; m0 < A.nsab; m0 += 16)
            for (int n0 = 0; n0 < A.nf; n0 += 16) {
                d4_t x = wave_mfma_tile(MabT /* M^T [e][r] */, 1, A.nsab, A.nsab, E0, A.nf, 1, A.nf, A.ne, m0, n0, lane);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int r = m0 + (lane >> 4) + 4 * q, f = n0 + (lane & 15);
                    if (r < A.nsab && f < A.nf) X[r * A.nf + f] = x[q];
                }
            }
    } else if (live) {
        for (int o = lane; o < A.nsab * A.nf; o += GSZ) {
            int r = o / A.nf, f = o - r * A.nf;
            double s = 0.0;
            for (int e = 0; e < A.ne; e++) s += Mab[r * A.ne + e] * E0[e * A.nf + f];
            X[o] = s;
        }
    }
    __syncthreads();
    if (MFMA && GSZ == 64 && mfma_worthwhile(A.nsab, A.nscd, A.nf)) {
        for (int m0 = 0; m0 < A.nsab; m0 += 16)
            for (int n0 = 0; n0 < A.nscd; n0 += 16) {
                d4_t o4 = wave_mfma_tile(X, A.nf, 1, A.nsab, McdT /* M^T [f][c] */, A.nscd, 1, A.nscd, A.nf, m0, n0, lane);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    int r = m0 + (lane >> 4) + 4 * q, c = n0 + (lane & 15);
                    if (r < A.nsab && c < A.nscd) { if (A.dense_mode) emit_dense(r, c, o4[q]); else O[r * A.nscd + c] = o4[q]; }
                }
            }
    } else if (live) {
        for (int o = lane; o < A.nsab * A.nscd; o += GSZ) {
            int r = o / A.nscd, c = o - r * A.nscd;
            double s = 0.0;
            for (int f = 0; f < A.nf; f++) s += X[r * A.nf + f] * McdT[f * A.nscd + c];
            if (A.dense_mode) emit_dense(r, c, s); else O[o] = s;
        }
    }
    if (A.dense_mode) return;
    __syncthreads();
    if (!live) return;
    // Scatter, one symmetry image at a time, in the address order of the tile layout: a tile row is (j, l pair) and inside it
    // the 16-byte chunks run over (i, k), so the lanes enumerate (j | l pair | i | k | l parity) of the image's roles -- one
    // store instruction then covers runs of 2 n_k consecutive doubles per i (48 / 80 / 112 bytes for p / d / f kets) instead of
    // 64 isolated doubles in 64 different rows, and the neighbouring kets (neighbouring tasks, same XCD: xcd_block) complete
    // the lines.  Role tables: image bit 0 swaps the bra pair, bit 1 the ket pair, bit 2 exchanges bra and ket (numbering of
    // `tbase`: 5 = (l,k,i,j), 6 = (k,l,j,i)).
    {
        const int n0_ = nsh[0], n1_ = nsh[1], n2_ = nsh[2], n3_ = nsh[3];
#pragma unroll 1
        for (int img = 0; img < 8; img++) {
            if (!(mask & (1u << img))) continue;
            // shell (0..3 = ab.i, ab.j, cd.i, cd.j) playing the i, j, k, l role of this image
            const int ri = (0x32321010 >> (4 * img)) & 3, rj = (0x23230101 >> (4 * img)) & 3;
            const int rk = (0x11003322 >> (4 * img)) & 3, rl = (0x00112233 >> (4 * img)) & 3;
            auto pick = [](int a, int b, int c_, int d, int q) { return q == 0 ? a : (q == 1 ? b : (q == 2 ? c_ : d)); };
            const int Ni = pick(n0_, n1_, n2_, n3_, ri), Nj = pick(n0_, n1_, n2_, n3_, rj), Nk = pick(n0_, n1_, n2_, n3_, rk),
                      Nl = pick(n0_, n1_, n2_, n3_, rl);
            const int Ai = pick(aos[0], aos[1], aos[2], aos[3], ri), Aj = pick(aos[0], aos[1], aos[2], aos[3], rj),
                      Ak = pick(aos[0], aos[1], aos[2], aos[3], rk), Al = pick(aos[0], aos[1], aos[2], aos[3], rl);
            const int p0 = Al >> 1, npair = ((Al + Nl - 1) >> 1) - p0 + 1;
            const int row = Ni * Nk * 2, tot = Nj * npair * row;
            const float inv_row = 1.0f / (float)row, inv_nk = 1.0f / (float)Nk, inv_np = 1.0f / (float)npair;
            // position of each role's index inside O[(sa nsb + sb) nscd + sc nsd + sd] and inside the sub-block number
            const int st_a = n1_ * A.nscd, st_b = A.nscd, st_c = n3_, st_d = 1;
            const int Si = pick(st_a, st_b, st_c, st_d, ri), Sj = pick(st_a, st_b, st_c, st_d, rj), Sk = pick(st_a, st_b, st_c, st_d, rk),
                      Sl = pick(st_a, st_b, st_c, st_d, rl);
            const int Li = pick(lo[0], lo[1], lo[2], lo[3], ri), Lj = pick(lo[0], lo[1], lo[2], lo[3], rj), Lk = pick(lo[0], lo[1], lo[2], lo[3], rk),
                      Ll = pick(lo[0], lo[1], lo[2], lo[3], rl);
            for (int e = lane; e < tot; e += GSZ) {
                int t = (int)(((float)e + 0.5f) * inv_row);          // (jb, mp) row; exact for these small integers
                const int w = e - t * row;
                const int par = w & 1, h = w >> 1;
                const int ia = (int)(((float)h + 0.5f) * inv_nk), kc = h - ia * Nk;
                const int jb = (int)(((float)t + 0.5f) * inv_np), mp = t - jb * npair;
                const int labs = ((p0 + mp) << 1) + par;
                if (labs < Al || labs >= Al + Nl) continue;
                const double v = O[ia * Si + jb * Sj + kc * Sk + (labs - Al) * Sl];
                const int gi = Ai + ia, gj = Aj + jb, gk = Ak + kc, gl = labs;
                const int sub = (((gi >> 3) - Li) << ri) | (((gj >> 3) - Lj) << rj) | (((gk >> 3) - Lk) << rk) | (((gl >> 3) - Ll) << rl);
                put_tile_at(A, tbase[img][sub], gi, gj, gk, gl, v);
            }
        }
    }
}

template <bool MFMA, int GSZ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void eri_transform_scatter(XfArgs A)
{
    xf_body<MFMA, GSZ, 0>(A, nullptr);
}

// Fused evaluation of the mid / high classes: Rys quadrature, HRR + cart->sph products and the scatter of one quartet in ONE wave
// and ONE launch (round 3).  FMAXC = [e0|f0] components per lane kept in registers during the primitive loop.
struct FusedArgs { XfArgs X; EriArgs R; };
template <bool MFMA, int FMAXC>
__global__ __launch_bounds__(64) void eri_fused_kernel(FusedArgs F)
{
    xf_body<MFMA, 64, FMAXC>(F.X, &F.R);
}

// =================================================================================================
// ERI generation for the LOW angular classes: one THREAD per contracted shell quartet, everything in registers, one launch.
//
// The wave-per-quartet pair (eri_rys_kernel + eri_transform_scatter) is latency bound on the classes that make up three
// quarters of all quartets ((ss|ss) ... (dp|ps), (fs|ds)): a handful of components keeps 64 lanes idle and the [e0|f0] block
// makes a round trip through memory between the two kernels.  Here the angular momenta are template parameters, so that the
// primitive loop (Rys roots -> 2-D recurrence -> component products), the horizontal recurrence, the cartesian->spherical
// transforms and the scatter into the resident tiles unroll completely into register code; 64 quartets per wave are in
// flight instead of 1-4.  The Chebyshev coefficients of the Rys roots/weights for this root count (3.5-13 KB) are staged in
// LDS once per workgroup (the per-lane table look-ups then never leave the CU).
// Eligible: NE * NF <= 64 accumulators and <= 3 roots (17 classes up to (dp|ps), (ds|ds), (fs|ds), (fd|ss)).
// =================================================================================================
__host__ __device__ constexpr int c_ncart(int l) { return (l + 1) * (l + 2) / 2; }
__host__ __device__ constexpr int c_cidx(int l, int lx, int ly) { return (l - lx) * (l - lx + 1) / 2 + (l - lx - ly); }
__host__ __device__ constexpr int c_eoff(int la, int deg) { int n = 0; for (int e = la; e < deg; e++) n += c_ncart(e); return n; }
__host__ __device__ constexpr int c_ne(int la, int lb) { return c_eoff(la, la + lb + 1); }
__host__ __device__ constexpr int c_binom(int n, int k) { int r = 1; for (int i = 1; i <= k; i++) r = r * (n - k + i) / i; return r; }

struct TpqArgs {
    const PairRec *bra, *ket;
    const double *prim;
    const int64_t *prefix;
    int nbra;
    int64_t t0, ntask;
    const double *c2s;
    int c2s_off[LMAX + 2];
    RysDev rys;
    XfArgs X;       // tile directory (tile_table, tile_off, tiles, nao) for put_tile
    const double *shell_xyz; // [nbas][3] shell centres (Bohr)
    int check_owner;
    const double *q_bra, *q_ket; // per-task Schwarz rejection (qtol > 0), see EriArgs
    double qtol;
    unsigned xcd;            // XCD-aware block map (xcd_block)
};

// HRR + cart->sph of one shell pair applied to one index of a register array:
//   in [NE] (E-index, degrees L1 .. L1+L2) -> out [(2L1+1)(2L2+1)];  element (idx, o) lives at idx * SI + o * SO.
template <int L1, int L2, int NO, int SI_IN, int SO_IN, int SI_OUT, int SO_OUT>
__device__ __forceinline__ void tpq_pair_transform(const double *in, double *out, const double AB[3], const MI_CONST_AS double *c1,
                                                   const MI_CONST_AS double *c2)
{
    constexpr int NCA = c_ncart(L1), NCB = c_ncart(L2), NSA = 2 * L1 + 1, NSB = 2 * L2 + 1;
    double pw[3][L2 + 1];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        pw[d][0] = 1.0;
#pragma unroll
        for (int q = 1; q <= L2; q++) pw[d][q] = pw[d][q - 1] * AB[d];
    }
#pragma unroll
    for (int o = 0; o < NO; o++) {
        // horizontal recurrence (closed form): (a b| = sum_{i <= b} C(b,i) AB^(b-i) (a+i 0|
        double g[NCA * NCB];
#pragma unroll
        for (int ax = L1; ax >= 0; ax--)
#pragma unroll
            for (int ay = L1 - ax; ay >= 0; ay--) {
                const int az = L1 - ax - ay, ia = c_cidx(L1, ax, ay);
#pragma unroll
                for (int bx = L2; bx >= 0; bx--)
#pragma unroll
                    for (int by = L2 - bx; by >= 0; by--) {
                        const int bz = L2 - bx - by, ib = c_cidx(L2, bx, by);
                        double v = 0.0;
#pragma unroll
                        for (int ix = 0; ix <= bx; ix++)
#pragma unroll
                            for (int iy = 0; iy <= by; iy++)
#pragma unroll
                                for (int iz = 0; iz <= bz; iz++) {
                                    const int deg = L1 + ix + iy + iz;
                                    const int e = c_eoff(L1, deg) + c_cidx(deg, ax + ix, ay + iy);
                                    const double cf = (double)(c_binom(bx, ix) * c_binom(by, iy) * c_binom(bz, iz));
                                    v = fma(cf * pw[0][bx - ix] * pw[1][by - iy] * pw[2][bz - iz], in[e * SI_IN + o * SO_IN], v);
                                }
                        g[ia * NCB + ib] = v;
                    }
            }
        // cartesian -> spherical on both shells (coefficient tables in constant memory: scalar loads)
        double h[NSA * NCB];
#pragma unroll
        for (int s1 = 0; s1 < NSA; s1++)
#pragma unroll
            for (int b = 0; b < NCB; b++) {
                double v = 0.0;
#pragma unroll
                for (int a = 0; a < NCA; a++) v = fma(c1[a * NSA + s1], g[a * NCB + b], v);
                h[s1 * NCB + b] = v;
            }
#pragma unroll
        for (int s1 = 0; s1 < NSA; s1++)
#pragma unroll
            for (int s2 = 0; s2 < NSB; s2++) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < NCB; b++) v = fma(c2[b * NSB + s2], h[s1 * NCB + b], v);
                out[(s1 * NSB + s2) * SI_OUT + o * SO_OUT] = v;
            }
    }
}

#define TPQ_BLOCK 128
template <int LA, int LB, int LC, int LD>
__global__ __launch_bounds__(TPQ_BLOCK) void eri_tpq_kernel(TpqArgs A)
{
    constexpr int NR = (LA + LB + LC + LD) / 2 + 1;
    constexpr int NMAX = LA + LB, MMAX = LC + LD;
    constexpr int NE = c_ne(LA, LB), NF = c_ne(LC, LD);
    constexpr int NSAB = (2 * LA + 1) * (2 * LB + 1), NSCD = (2 * LC + 1) * (2 * LD + 1);
    extern __shared__ double cheb[];   // Chebyshev coefficients of the 2 NR root/weight functions, all intervals
    const int nint = A.rys.nint[NR];
    {
        const int ntab = nint * 2 * NR * (RYS_DEG + 1);
        const double *src = A.rys.cheb + A.rys.off[NR];
        for (int q = threadIdx.x; q < ntab; q += TPQ_BLOCK) cheb[q] = src[q];
    }
    __syncthreads();
    const int64_t tl = (int64_t)xcd_block(blockIdx.x, A.xcd) * TPQ_BLOCK + threadIdx.x;
    if (tl >= A.ntask) return;
    int ib, ik;
    find_task(A.prefix, A.nbra, A.t0 + tl, ib, ik);
    if (A.qtol > 0.0 && A.q_bra[ib] * A.q_ket[ik] < A.qtol) return;
    const PairRec ab = A.bra[ib], cd = A.ket[ik];
    constexpr int ni = 2 * LA + 1, nj = 2 * LB + 1, nk = 2 * LC + 1, nl = 2 * LD + 1;
    // block ranges of the four shells: which index images can land in a canonical tile, and is anything resident here
    const int lo[4] = {ab.ao_i >> 3, ab.ao_j >> 3, cd.ao_i >> 3, cd.ao_j >> 3};
    const int hi[4] = {(ab.ao_i + ni - 1) >> 3, (ab.ao_j + nj - 1) >> 3, (cd.ao_i + nk - 1) >> 3, (cd.ao_j + nl - 1) >> 3};
    if (A.check_owner) {
        bool hit = false;
        for (int q = 0; q < 16; q++) {
            int I = (q & 1) ? hi[0] : lo[0], J = (q & 2) ? hi[1] : lo[1], K = (q & 4) ? hi[2] : lo[2], L = (q & 8) ? hi[3] : lo[3];
            int h1 = max(I, J), l1 = min(I, J), bij = h1 * (h1 + 1) / 2 + l1;
            int h2 = max(K, L), l2 = min(K, L), bkl = h2 * (h2 + 1) / 2 + l2;
            int bmax = max(bij, bkl), bmin = min(bij, bkl);
            hit = hit || A.X.tile_table[(size_t)bmax * (bmax + 1) / 2 + bmin] >= 0;
        }
        if (!hit) return;
    }

    double acc[NE * NF];
#pragma unroll
    for (int q = 0; q < NE * NF; q++) acc[q] = 0.0;
    for (int ip = 0; ip < ab.nprim; ip++) {
        const double *b = A.prim + (size_t)(ab.prim_off + ip) * 8;
        const double p = b[0], Px = b[1], Py = b[2], Pz = b[3], Kab = b[7];
        const double PA[3] = {b[4], b[5], b[6]};
        for (int jp = 0; jp < cd.nprim; jp++) {
            const double *kk = A.prim + (size_t)(cd.prim_off + jp) * 8;
            const double q = kk[0];
            const double PQ[3] = {Px - kk[1], Py - kk[2], Pz - kk[3]};
            const double QC[3] = {kk[4], kk[5], kk[6]};
            const double pq1 = 1.0 / (p + q);
            const double x = p * q * pq1 * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
            const double pref = Kab * kk[7] * 34.986836655249725 /* 2 pi^2.5 */ * pq1 * sqrt(p + q) / (p * q);
            // Rys roots u_r and weights w_r: Chebyshev interpolation from the LDS copy, asymptotic form beyond the table
            double u[NR], w[NR];
            if (x < nint * RYS_H) {
                int iv = (int)(x * (1.0 / RYS_H));
                if (iv >= nint) iv = nint - 1;
                const double sx = (x - (iv * RYS_H + 0.5 * RYS_H)) * (2.0 / RYS_H), s2 = 2.0 * sx;
                const double *cb = cheb + (size_t)iv * 2 * NR * (RYS_DEG + 1);
#pragma unroll
                for (int f = 0; f < 2 * NR; f++) {
                    const double *cc = cb + f * (RYS_DEG + 1);
                    double b1 = 0.0, b2 = 0.0;
#pragma unroll
                    for (int kq = RYS_DEG; kq >= 1; kq--) {
                        double tq = s2 * b1 - b2 + cc[kq];
                        b2 = b1;
                        b1 = tq;
                    }
                    const double val = sx * b1 - b2 + cc[0];
                    if (f < NR) u[f] = val; else w[f - NR] = val;
                }
            } else {
                const double rx = 1.0 / x, rsx = rsqrt(x);
#pragma unroll
                for (int f = 0; f < NR; f++) { u[f] = A.rys.herm_r[NR * RYS_NMAX + f] * rx; w[f] = A.rys.herm_w[NR * RYS_NMAX + f] * rsx; }
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                const double ur = u[r];
                const double b00 = 0.5 * ur * pq1, b10 = 0.5 / p * (1.0 - ur * q * pq1), b01 = 0.5 / q * (1.0 - ur * p * pq1);
                double T[3][NMAX + 1][MMAX + 1];
#pragma unroll
                for (int d = 0; d < 3; d++) {
                    const double c00 = PA[d] - ur * q * pq1 * PQ[d], c01 = QC[d] + ur * p * pq1 * PQ[d];
                    T[d][0][0] = d == 2 ? w[r] * pref : 1.0;
#pragma unroll
                    for (int n = 0; n < NMAX; n++) T[d][n + 1][0] = c00 * T[d][n][0] + (n > 0 ? n * b10 * T[d][n - 1][0] : 0.0);
#pragma unroll
                    for (int m = 0; m < MMAX; m++)
#pragma unroll
                        for (int n = 0; n <= NMAX; n++) {
                            double v = c01 * T[d][n][m];
                            if (m > 0) v = fma(m * b01, T[d][n][m - 1], v);
                            if (n > 0) v = fma(n * b00, T[d][n - 1][m], v);
                            T[d][n][m + 1] = v;
                        }
                }
                // component products: e = (ex,ey,ez) of degree LA..LA+LB, f of degree LC..LC+LD (orders as build_comp_table)
#pragma unroll
                for (int de = LA; de <= LA + LB; de++)
#pragma unroll
                    for (int ex = de; ex >= 0; ex--)
#pragma unroll
                        for (int ey = de - ex; ey >= 0; ey--) {
                            const int ez = de - ex - ey, ie = c_eoff(LA, de) + c_cidx(de, ex, ey);
#pragma unroll
                            for (int df = LC; df <= LC + LD; df++)
#pragma unroll
                                for (int fx = df; fx >= 0; fx--)
#pragma unroll
                                    for (int fy = df - fx; fy >= 0; fy--) {
                                        const int fz = df - fx - fy, jf = c_eoff(LC, df) + c_cidx(df, fx, fy);
                                        acc[ie * NF + jf] = fma(T[0][ex][fx] * T[1][ey][fy], T[2][ez][fz], acc[ie * NF + jf]);
                                    }
                        }
            }
        }
    }
    // HRR + cart->sph: bra index first ([NE][NF] -> [NSAB][NF]), then the ket index ([NSAB][NF] -> [NSAB][NSCD])
    const MI_CONST_AS double *c2s = as_const(A.c2s);
    // AB = A - B of each pair (first minus second shell of the record) from the shell centres
    double ABv[3], CDv[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        ABv[d] = A.shell_xyz[3 * ab.sh_i + d] - A.shell_xyz[3 * ab.sh_j + d];
        CDv[d] = A.shell_xyz[3 * cd.sh_i + d] - A.shell_xyz[3 * cd.sh_j + d];
    }
    double X1[NSAB * NF];
    tpq_pair_transform<LA, LB, NF, NF, 1, NF, 1>(acc, X1, ABv, c2s + A.c2s_off[LA], c2s + A.c2s_off[LB]);
    double out[NSAB * NSCD];
    tpq_pair_transform<LC, LD, NSAB, 1, NF, 1, NSCD>(X1, out, CDv, c2s + A.c2s_off[LC], c2s + A.c2s_off[LD]);
    // scatter every symmetry image that lands in a canonical resident tile.  The directory look-up (two dependent global
    // loads) is done once per change of tile, not per element: the elements of an image touch at most 16 tiles, usually one.
    auto ok = [&](int a, int b, int c, int d) { return hi[a] >= lo[b] && hi[c] >= lo[d]; };
    const unsigned mask = (ok(0, 1, 2, 3) ? 1u : 0u) | (ok(1, 0, 2, 3) ? 2u : 0u) | (ok(0, 1, 3, 2) ? 4u : 0u) | (ok(1, 0, 3, 2) ? 8u : 0u) |
                          (ok(2, 3, 0, 1) ? 16u : 0u) | (ok(3, 2, 0, 1) ? 32u : 0u) | (ok(2, 3, 1, 0) ? 64u : 0u) | (ok(3, 2, 1, 0) ? 128u : 0u);
    const MI_CONST_AS int32_t *ttab = as_const(A.X.tile_table);
    const MI_CONST_AS int64_t *toff = as_const(A.X.tile_off);
    const int nao = A.X.nao;
#pragma unroll
    for (int img = 0; img < 8; img++) {
        // images 5 and 6 of this construction are (k,l,j,i) and (l,k,i,j): bits 6 and 5 of the block-range mask
        const int mbit = img == 5 ? 6 : (img == 6 ? 5 : img);
        if (!(mask & (1u << mbit))) continue;
        size_t slot_c = ~(size_t)0;
        int64_t base_c = -1;
#pragma unroll
        for (int sa = 0; sa < ni; sa++)
#pragma unroll
            for (int sb = 0; sb < nj; sb++)
#pragma unroll
                for (int sc = 0; sc < nk; sc++)
#pragma unroll
                    for (int sd = 0; sd < nl; sd++) {
                        const int a0 = ab.ao_i + sa, a1 = ab.ao_j + sb, a2 = cd.ao_i + sc, a3 = cd.ao_j + sd;
                        // image `img`: bit 0 swaps the bra pair, bit 1 the ket pair, bit 2 exchanges bra and ket
                        int i = (img & 1) ? a1 : a0, j = (img & 1) ? a0 : a1, k = (img & 2) ? a3 : a2, l = (img & 2) ? a2 : a3;
                        if (img & 4) { int t_ = i; i = k; k = t_; t_ = j; j = l; l = t_; }
                        const int I = i >> 3, J = j >> 3, K = k >> 3, L = l >> 3;
                        if (I < J || K < L) continue;
                        const int bij = I * (I + 1) / 2 + J, bkl = K * (K + 1) / 2 + L;
                        if (bij < bkl) continue;
                        const size_t slot = (size_t)bij * (bij + 1) / 2 + bkl;
                        if (slot != slot_c) {
                            slot_c = slot;
                            const int32_t t = ttab[slot];
                            base_c = t >= 0 ? toff[t] : -1;
                        }
                        if (base_c < 0) continue;
                        const int bi = min(BLK, nao - I * BLK), bk = min(BLK, nao - K * BLK);
                        double w;
                        const int64_t e = tile_elem(A.X.tri != 0, I == J, K == L, bij == bkl, bi, bk, i & 7, j & 7, k & 7, l & 7, &w);
                        if (e >= 0) A.X.tiles[base_c + e] = w * out[(sa * nj + sb) * NSCD + sc * nl + sd];
                    }
    }
}

// Schwarz: q[b] = sqrt(max_ab |(ab|ab)|) from the diagonal-task E0 blocks.
struct SchwarzArgs {
    const PairRec *bra;
    const double *Mbuf;
    const double *work;
    int ne, nsab, ncomp;
    double *q;
};
__global__ __launch_bounds__(64) void schwarz_diag_kernel(SchwarzArgs A)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const PairRec ab = A.bra[blockIdx.x];
    const double *E0g = A.work + (size_t)blockIdx.x * A.ncomp;
    for (int c = lane; c < A.ne * A.ne; c += 64) lds[c] = E0g[c];
    __syncthreads();
    const double *M = A.Mbuf + ab.m_off;
    double mx = 0.0;
    for (int r = lane; r < A.nsab; r += 64) {
        double s = 0.0;
        for (int e = 0; e < A.ne; e++) {
            double t = 0.0;
            for (int f = 0; f < A.ne; f++) t += lds[e * A.ne + f] * M[r * A.ne + f];
            s += M[r * A.ne + e] * t;
        }
        mx = fmax(mx, fabs(s));
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if (lane == 0) A.q[blockIdx.x] = sqrt(mx);
}

// =================================================================================================
// Host side of ERI preparation
// =================================================================================================
static void build_comp_table(int la, int lb, int lc, int ld, std::vector<uint32_t> &comp)
{
    int M1 = lc + ld + 1;
    comp.clear();
    for (int e = la; e <= la + lb; e++)
        for (int ex = e; ex >= 0; ex--)
            for (int ey = e - ex; ey >= 0; ey--) {
                int ez = e - ex - ey;
                for (int f = lc; f <= lc + ld; f++)
                    for (int fx = f; fx >= 0; fx--)
                        for (int fy = f - fx; fy >= 0; fy--) {
                            int fz = f - fx - fy;
                            uint32_t ix = ex * M1 + fx, iy = ey * M1 + fy, iz = ez * M1 + fz;
                            comp.push_back(ix | (iy << 10) | (iz << 20));
                        }
            }
}

// M[(sa,sb)][e] = sum_{a,b cart} c2s_a[a][sa] c2s_b[b][sb] prod_d C(b_d,i_d) AB_d^(b_d-i_d), e = a + i
static void build_M(int la, int lb, const double AB[3], const std::vector<double> &ca, const std::vector<double> &cb,
                    double *M /* [nsa*nsb][ne] */)
{
    int nsa = 2 * la + 1, nsb = 2 * lb + 1, ne = ne_of(la, lb);
    std::fill(M, M + (size_t)nsa * nsb * ne, 0.0);
    // e-offsets by degree
    int eoff[2 * LMAX + 2];
    eoff[la] = 0;
    for (int e = la; e < la + lb + 1; e++) eoff[e + 1] = eoff[e] + ncart(e);
    double pw[3][LMAX + 1];
    for (int d = 0; d < 3; d++) { pw[d][0] = 1.0; for (int k = 1; k <= LMAX; k++) pw[d][k] = pw[d][k - 1] * AB[d]; }
    int ia = 0;
    for (int ax = la; ax >= 0; ax--)
        for (int ay = la - ax; ay >= 0; ay--, ia++) {
            int az = la - ax - ay;
            int ibx = 0;
            for (int bx = lb; bx >= 0; bx--)
                for (int by = lb - bx; by >= 0; by--, ibx++) {
                    int bz = lb - bx - by;
                    for (int ix = 0; ix <= bx; ix++)
                        for (int iy = 0; iy <= by; iy++)
                            for (int iz = 0; iz <= bz; iz++) {
                                double coef = binom(bx, ix) * binom(by, iy) * binom(bz, iz) * pw[0][bx - ix] * pw[1][by - iy] * pw[2][bz - iz];
                                int deg = la + ix + iy + iz;
                                int e = eoff[deg] + cart_index(deg, ax + ix, ay + iy);
                                for (int sa = 0; sa < nsa; sa++) {
                                    double c1 = ca[(size_t)ia * nsa + sa];
                                    if (c1 == 0.0) continue;
                                    for (int sb = 0; sb < nsb; sb++) {
                                        double c2 = cb[(size_t)ibx * nsb + sb];
                                        if (c2 != 0.0) M[(size_t)(sa * nsb + sb) * ne + e] += c1 * c2 * coef;
                                    }
                                }
                            }
                }
        }
}

// coarse[g] = largest bra index ib with prefix[ib] <= g*1024, for g = 0 .. ntask/1024 + 1 (appended to `prefix`)
static void append_coarse_index(std::vector<int64_t> &prefix)
{
    const int nbra = (int)prefix.size() - 1;
    const int64_t ntask = prefix.back();
    const int64_t ng = (ntask >> 10) + 2;
    prefix.reserve(prefix.size() + (size_t)ng);
    int ib = 0;
    for (int64_t g = 0; g < ng; g++) {
        const int64_t t = g << 10;
        while (ib < nbra && prefix[ib + 1] <= t) ib++;
        prefix.push_back(ib);
    }
}

// Task list of a (bra class, ket class) pair: bra b visits the leading kets of the (bound-ordered, see mi_eri_prepare step 3)
// ket list whose bound cq can pass the Schwarz test with the bra's own q; inside one class only kets up to the bra itself
// (canonical quartets).  `prefix` = cumulative counts + the coarse search index; returns the number of tasks.  Kets inside a
// surviving cluster whose own q fails the test are rejected per task by the kernels (qtol).
static int64_t class_prefix(const PairClass &B, const PairClass &Kc, bool same, double tol, std::vector<int64_t> &prefix)
{
    prefix.assign(B.recs.size() + 1, 0);
    for (size_t b = 0; b < B.recs.size(); b++) {
        const double thr = tol / B.q[b];
        size_t lo = 0, hi = Kc.cq.size();
        while (lo < hi) { size_t mid = (lo + hi) / 2; if (Kc.cq[mid] >= thr) lo = mid + 1; else hi = mid; }
        int64_t cnt = (int64_t)lo;
        if (same) cnt = std::min<int64_t>(cnt, (int64_t)b + 1);
        prefix[b + 1] = prefix[b] + cnt;
    }
    const int64_t ntask = prefix.back();
    append_coarse_index(prefix);
    return ntask;
}

template <class T>
static int upload(T **dst, const std::vector<T> &v)
{
    if (*dst) { dev_free(*dst); *dst = nullptr; }
    size_t n = std::max<size_t>(v.size(), 1);
    HIPCHK(dev_malloc_impl((void **)dst, sizeof(T) * n));
    if (!v.empty()) HIPCHK(hipMemcpy(*dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return 0;
}

// Classes that run four quartets per wave (measured per class on ibuprofen/def2-TZVP: with more roots or components
// the 16-lane groups need more primitive batches / passes than they win by overlapping four latency chains).
static bool eri_small_class(const EriArgs &E) { return E.nroots <= 2 && E.ncomp <= 18; }

static int launch_eri(mi_ctx *c, EriArgs &E, int nblocks, hipStream_t st)
{
    // E.ntask tasks; the low angular classes run four quartets per wave (16 lanes each) unless their contraction depth
    // makes the four quartets of a wave too unequal: cost model in primitive batches per four quartets
    if (E.ntask != nblocks) return fail("launch_eri: ntask/grid mismatch");
    if (nblocks < 2048) E.xcd = 0;   // too small for the chunked map to matter
    int perlane = (E.ncomp + 63) / 64;
    if (eri_small_class(E) && !E.diag && E.h_shared_np > 0.0) {
        const int pb16 = std::max(1, 16 / (3 * E.nroots)), pb64 = std::max(1, 64 / (3 * E.nroots));
        const double cost16 = std::ceil(E.h_shared_np * E.h_vary_max4 / pb16) + 1.5;
        const double cost64 = 4.0 * (std::ceil(E.h_shared_np * E.h_vary_mean / pb64) + 1.5);
        if (cost16 < cost64) {
            EriArgs G = E;
            G.PB = pb16;
            size_t shm = 4 * sizeof(double) * ((size_t)G.PB * G.nroots * 3 * G.tsz + (size_t)G.PB * 2 * G.nroots + (size_t)G.prim_lds * 8);
            hipLaunchKernelGGL((eri_rys_kernel<2, 16>), dim3(eri_grid((nblocks + 3) / 4, G.xcd)), dim3(64), shm, st, G);
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
    // Round 3: four quartets per wave also for the MID classes when the contraction is shallow (uncontracted d / f shells: one
    // primitive quartet keeps 2 n of 64 lanes busy in the root phase and 3 n in the recurrence phase of a one-quartet wave).
    if (c->opt_rys_qpw_maxcomp > 0 && !E.diag && E.nroots <= 5 && E.ncomp > 18 && E.ncomp <= c->opt_rys_qpw_maxcomp && E.h_shared_np > 0.0 &&
        E.h_shared_np * E.h_vary_max4 <= c->opt_rys_qpw_maxprim) {
        EriArgs G = E;
        G.PB = std::max(1, 16 / (3 * E.nroots));
        const size_t shm = 4 * sizeof(double) * ((size_t)G.PB * G.nroots * 3 * G.tsz + (size_t)G.PB * 2 * G.nroots + (size_t)G.prim_lds * 8);
        const dim3 grid(eri_grid((nblocks + 3) / 4, G.xcd));
        const int per16 = (E.ncomp + 15) / 16;
        if (shm <= 64 * 1024) {
            if (per16 <= 4) hipLaunchKernelGGL((eri_rys_kernel<4, 16>), grid, dim3(64), shm, st, G);
            else if (per16 <= 8) hipLaunchKernelGGL((eri_rys_kernel<8, 16>), grid, dim3(64), shm, st, G);
            else hipLaunchKernelGGL((eri_rys_kernel<16, 16>), grid, dim3(64), shm, st, G);
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
    size_t shm = sizeof(double) * ((size_t)E.PB * E.nroots * 3 * E.tsz + (size_t)E.PB * 2 * E.nroots + (size_t)E.prim_lds * 8);
    if (perlane <= 1) hipLaunchKernelGGL((eri_rys_kernel<1, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    else if (perlane <= 2 && c->opt_rys_fine) hipLaunchKernelGGL((eri_rys_kernel<2, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    else if (perlane <= 4) hipLaunchKernelGGL((eri_rys_kernel<4, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    else if (perlane <= 8 && c->opt_rys_fine) hipLaunchKernelGGL((eri_rys_kernel<8, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    else if (perlane <= 16) hipLaunchKernelGGL((eri_rys_kernel<16, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    else hipLaunchKernelGGL((eri_rys_kernel<32, 64>), dim3(eri_grid(nblocks, E.xcd)), dim3(64), shm, st, E);
    HIPCHK(hipGetLastError());
    return 0;
}

static void setup_eri_dims(EriArgs &E, int la, int lb, int lc, int ld)
{
    E.la = la; E.lb = lb; E.lc = lc; E.ld = ld;
    E.nmax = la + lb; E.mmax = lc + ld;
    E.nroots = (la + lb + lc + ld) / 2 + 1;
    E.tsz = ((E.nmax + 1) * (E.mmax + 1)) | 1; // table stride in LDS, odd so that the per-(slot,direction) tables start in different banks
    E.ncomp = ne_of(la, lb) * ne_of(lc, ld);
    int pb = 64 / (3 * E.nroots);
    E.PB = std::max(1, pb);
}

// Launch the thread-per-quartet kernel of an eligible class: returns 1 if launched, 0 if the class is not covered, -1 on error.
template <int LA, int LB, int LC, int LD>
static int launch_eri_tpq_t(const TpqArgs &Q, hipStream_t st)
{
    constexpr int NR = (LA + LB + LC + LD) / 2 + 1;
    const size_t shm = sizeof(double) * (size_t)RYS_NINT_H[NR] * 2 * NR * (RYS_DEG + 1);
    const int64_t MAXB = (int64_t)1 << 24;
    for (int64_t t0 = 0; t0 < Q.ntask; t0 += MAXB * TPQ_BLOCK) {   // grid.x stays well below 2^31 (also after the XCD-map rounding)
        TpqArgs P = Q;
        P.t0 = Q.t0 + t0;
        P.ntask = std::min<int64_t>(Q.ntask - t0, MAXB * TPQ_BLOCK);
        const int64_t nblk = (P.ntask + TPQ_BLOCK - 1) / TPQ_BLOCK;
        if (nblk < 1024) P.xcd = 0;
        hipLaunchKernelGGL((eri_tpq_kernel<LA, LB, LC, LD>), dim3(eri_grid(nblk, P.xcd)), dim3(TPQ_BLOCK), shm, st, P);
    }
    if (hipGetLastError() != hipSuccess) return fail("eri_tpq_kernel launch failed");
    return 1;
}

static bool tpq_has_class(int la, int lb, int lc, int ld);
static int launch_eri_tpq(int la, int lb, int lc, int ld, const TpqArgs &Q, hipStream_t st)
{
    const int key = ((la * 4 + lb) * 4 + lc) * 4 + ld;
#define TPQ_CASE(a, b, c_, d) case (((a) * 4 + (b)) * 4 + (c_)) * 4 + (d): return launch_eri_tpq_t<a, b, c_, d>(Q, st)
    switch (key) {
        TPQ_CASE(0, 0, 0, 0); TPQ_CASE(1, 0, 0, 0); TPQ_CASE(1, 0, 1, 0); TPQ_CASE(1, 1, 0, 0); TPQ_CASE(1, 1, 1, 0);
        TPQ_CASE(2, 0, 0, 0); TPQ_CASE(2, 0, 1, 0); TPQ_CASE(2, 0, 1, 1); TPQ_CASE(2, 0, 2, 0); TPQ_CASE(2, 1, 0, 0);
        TPQ_CASE(2, 1, 1, 0); TPQ_CASE(2, 2, 0, 0); TPQ_CASE(3, 0, 0, 0); TPQ_CASE(3, 0, 1, 0); TPQ_CASE(3, 0, 2, 0);
        TPQ_CASE(3, 1, 0, 0); TPQ_CASE(3, 2, 0, 0);
        TPQ_CASE(2, 1, 2, 0); TPQ_CASE(2, 2, 1, 0); TPQ_CASE(3, 3, 0, 0);   // 96 / 93 / 74 accumulators: one wave per SIMD, still ahead
        TPQ_CASE(3, 1, 1, 0); TPQ_CASE(3, 0, 1, 1); TPQ_CASE(3, 0, 3, 0);   // round 3 candidates: 75 / 90 / 100 accumulators
    default: return 0;
    }
#undef TPQ_CASE
}
static bool tpq_has_class(int la, int lb, int lc, int ld)
{
    static const int keys[][4] = {{0,0,0,0},{1,0,0,0},{1,0,1,0},{1,1,0,0},{1,1,1,0},{2,0,0,0},{2,0,1,0},{2,0,1,1},{2,0,2,0},{2,1,0,0},
                                  {2,1,1,0},{2,2,0,0},{3,0,0,0},{3,0,1,0},{3,0,2,0},{3,1,0,0},{3,2,0,0},{2,1,2,0},{2,2,1,0},{3,3,0,0},
                                  {3,1,1,0},{3,0,1,1},{3,0,3,0}};
    for (const auto &k : keys) if (k[0] == la && k[1] == lb && k[2] == lc && k[3] == ld) return true;
    return false;
}

// Sharding plan of the resident tile store: enumerate the (J,K,L) runs that survive the block-pair
// Schwarz test, then deal them to ranks longest-processing-time first (largest run to the least loaded rank) by the bytes a
// J/K build streams for them.  Pure host code and a pure function of its arguments, so every rank derives the same plan.
struct RunPlan { int J, K, L, count, owner; int64_t bytes; };

// doubles a tile occupies in the store: its chunks, rounded up to 256 bytes so that rows of following tiles stay sector-aligned
static inline int64_t tile_doubles_padded(bool dij, bool dkl, int bi, int bk) { return ((int64_t)tile_chunks(dij, dkl, bi, bk) * 2 + 31) / 32 * 32; }

static void plan_runs(int nao, const std::vector<double> &Qblk, double qmax, double tol, int nranks, bool tri, std::vector<RunPlan> &plan)
{
    const int nblk = (nao + BLK - 1) / BLK, nbp = nblk * (nblk + 1) / 2;
    auto bsize = [&](int B) { return std::min(BLK, nao - B * BLK); };
    plan.clear();
    // (13.6 M candidate tiles for ibuprofen/def2-TZVP: the ket block pairs are scanned by the host threads, their run lists
    // joined in kl order -- the plan does not depend on the number of threads)
    std::vector<std::vector<RunPlan>> per(nbp);
#pragma omp parallel for schedule(dynamic, 8) num_threads(host_threads())
    for (int kl = 0; kl < nbp; kl++) {
        if (!(Qblk[kl] * qmax >= tol)) continue;
        int K = (int)((std::sqrt(8.0 * kl + 1.0) - 1.0) / 2.0);
        while ((K + 1) * (K + 2) / 2 <= kl) K++;
        while (K * (K + 1) / 2 > kl) K--;
        const int L = kl - K * (K + 1) / 2;
        const int bk = bsize(K);
        for (int J = 0; J < nblk; J++) {
            RunPlan r{J, K, L, 0, 0, 0};
            for (int I = J; I < nblk; I++) {
                int ij = I * (I + 1) / 2 + J;
                if (ij < kl || Qblk[ij] * Qblk[kl] < tol) continue;
                r.count++;
                r.bytes += tile_doubles_padded(tri && I == J, tri && K == L, bsize(I), bk) * 8;
            }
            if (r.count) per[kl].push_back(r);
        }
    }
    size_t nrun = 0;
    for (const auto &v : per) nrun += v.size();
    plan.reserve(nrun);
    for (const auto &v : per) plan.insert(plan.end(), v.begin(), v.end());
    if (nranks <= 1) return;
    std::vector<int> ord(plan.size());
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return plan[a].bytes > plan[b].bytes; });
    // least-loaded rank: binary heap of (load, rank); ties go to the lower rank so the plan is reproducible
    std::vector<std::pair<int64_t, int>> heap(nranks);
    for (int r = 0; r < nranks; r++) heap[r] = {0, r};
    auto cmp = [](const std::pair<int64_t, int> &a, const std::pair<int64_t, int> &b) { return a > b; }; // min-heap
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (int o : ord) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        plan[o].owner = heap.back().second;
        heap.back().first += plan[o].bytes;
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
}

// Host-only view of the plan (no GPU, no context): bytes streamed per rank for a given block-pair Schwarz table
// qblk[nbp] (nbp = nblk(nblk+1)/2, nblk = ceil(nao/8), pair index I(I+1)/2+J).  Used by the CPU tests of the balance.
extern "C" int mi_plan_shards(int nao, const double *qblk, double tol, int nranks, int64_t *bytes_per_rank, int64_t *runs_per_rank)
{
    if (nao <= 0 || !qblk || nranks < 1 || !bytes_per_rank) return fail("mi_plan_shards: bad argument");
    const int nblk = (nao + BLK - 1) / BLK, nbp = nblk * (nblk + 1) / 2;
    std::vector<double> Q(qblk, qblk + nbp);
    double qmax = 0.0;
    for (double v : Q) qmax = std::max(qmax, v);
    std::vector<RunPlan> plan;
    plan_runs(nao, Q, qmax, tol, nranks, true, plan);
    for (int r = 0; r < nranks; r++) { bytes_per_rank[r] = 0; if (runs_per_rank) runs_per_rank[r] = 0; }
    for (const RunPlan &p : plan) { bytes_per_rank[p.owner] += p.bytes; if (runs_per_rank) runs_per_rank[p.owner]++; }
    return 0;
}

static int grad_records_host(mi_ctx *c);

extern "C" int mi_eri_prepare(mi_ctx *c, double tol, int rank, int nranks, void *stream)
{
    if (c && check_orbital_lmax(c, "mi_eri_prepare")) return -1;
    if (!c) return fail("mi_eri_prepare: null context");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("mi_eri_prepare: bad rank/nranks");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    auto t_start = std::chrono::steady_clock::now();
    auto t_phase = t_start;
    auto lap = [&](const char *what) {
        if (!getenv("MI355_DEBUG")) return;
        hipStreamSynchronize(st);
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[mi355] eri_prepare %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    free_eri(c);
    if (set_tile_order(c, c->opt_ao_order ? 1 : 0)) return -1;
    const int nbas = c->nbas;
    std::vector<std::vector<double>> c2s(LMAX + 1);
    for (int l = 0; l <= LMAX; l++) c2s_generic(l, c2s[l]);

    // ---- 1. shell pairs, primitive-pair records, transformation matrices
    std::vector<double> &prim = c->h_prim, &Mbuf = c->h_M;
    prim.clear(); Mbuf.clear();
    c->tol = tol; c->grad_ready = false; c->rank = rank; c->nranks = nranks;
    for (int la = 0; la <= LMAX; la++)
        for (int lb = 0; lb <= la; lb++) {
            PairClass &P = c->pc[pc_index(la, lb)];
            P.la = la; P.lb = lb; P.ne = ne_of(la, lb); P.nsab = (2 * la + 1) * (2 * lb + 1);
        }
    for (int A = 0; A < nbas; A++)
        for (int B = 0; B <= A; B++) {
            int si = A, sj = B;
            if (c->shells[si].l < c->shells[sj].l) std::swap(si, sj);
            const ShellH &I = c->shells[si], &J = c->shells[sj];
            PairClass &P = c->pc[pc_index(I.l, J.l)];
            double AB[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
            double r2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
            PairRec R;
            R.sh_i = si; R.sh_j = sj; R.ao_i = I.ao; R.ao_j = J.ao; R.pad = 0;
            R.prim_off = (int)(prim.size() / 8);
            int np = 0;
            for (int ip = 0; ip < I.nprim; ip++)
                for (int jp = 0; jp < J.nprim; jp++) {
                    double a = I.exps[ip], b = J.exps[jp], p = a + b, mu = a * b / p;
                    if (mu * r2 > 80.0) continue; // exp(-80) = 1.8e-35: below double resolution of any sum
                    double K = I.coef[ip] * J.coef[jp] * std::exp(-mu * r2);
                    double Pc[3];
                    for (int d = 0; d < 3; d++) Pc[d] = (a * I.r[d] + b * J.r[d]) / p;
                    double rec[8] = {p, Pc[0], Pc[1], Pc[2], Pc[0] - I.r[0], Pc[1] - I.r[1], Pc[2] - I.r[2], K};
                    prim.insert(prim.end(), rec, rec + 8);
                    np++;
                }
            if (np == 0) continue;
            R.nprim = np;
            R.m_off = (int)Mbuf.size();
            Mbuf.resize(Mbuf.size() + (size_t)2 * P.nsab * P.ne); // M [nsab][ne] followed by its transpose [ne][nsab]
            P.recs.push_back(R);
        }
    {   // HRR*c2s matrices of all pairs (OpenMP: this is the costly host part of the setup)
        std::vector<const PairRec *> all;
        for (int ci = 0; ci < NPC; ci++)
            for (const PairRec &R : c->pc[ci].recs) all.push_back(&R);
#pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads())
        for (size_t q = 0; q < all.size(); q++) {
            const ShellH &I = c->shells[all[q]->sh_i], &J = c->shells[all[q]->sh_j];
            double AB[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
            double *M = Mbuf.data() + all[q]->m_off;
            build_M(I.l, J.l, AB, c2s[I.l], c2s[J.l], M);
            // transposed copy: the MFMA transform reads both operands with the lane index on the contiguous dimension
            const int nsab = (2 * I.l + 1) * (2 * J.l + 1), ne = ne_of(I.l, J.l);
            double *Mt = M + (size_t)nsab * ne;
            for (int r = 0; r < nsab; r++)
                for (int e = 0; e < ne; e++) Mt[(size_t)e * nsab + r] = M[(size_t)r * ne + e];
        }
    }
    if (Mbuf.size() > (size_t)INT32_MAX) return fail("transformation-matrix buffer exceeds 2^31 doubles");
    if (upload(&c->d_prim, prim)) return -1;
    if (upload(&c->d_M, Mbuf)) return -1;

    lap("pairs + M matrices");
    // workspace for cartesian intermediates
    // [e0|f0] blocks travel from the Rys launch to the transform launch through this buffer: small enough that a batch written by
    // one launch is still in the 256 MiB Infinity Cache when the next reads it (`work_mb`)
    // sized to what the molecule can need (atoms of the initial guess: a few MiB), at most `work_mb`
    size_t WORK_DOUBLES = (size_t)8 << 17;
    {
        const size_t cap = (size_t)std::max(8, c->opt_work_mb) << 17;
        for (int bc = 0; bc < NPC; bc++)
            for (int kc = 0; kc <= bc; kc++) {
                const PairClass &B = c->pc[bc], &Kc = c->pc[kc];
                const double need = (double)B.recs.size() * (double)Kc.recs.size() * (double)B.ne * (double)Kc.ne;
                WORK_DOUBLES = (size_t)std::min<double>((double)cap, std::max<double>((double)WORK_DOUBLES, need));
            }
    }
    Scratch scr_work, scr_tasks;   // (returned to the per-device cache when this call ends, whichever way)
    if (scr_work.ensure(c->device, SCR_EVAL_WORK, sizeof(double) * WORK_DOUBLES)) return -1;
    double *d_work = (double *)scr_work.p;
    uint32_t *d_comp = nullptr;
    HIPCHK(dev_malloc(&d_comp, sizeof(uint32_t) * 8192));

    // ---- 2. Schwarz bounds per pair (GPU)
    double qmax = 0.0;
    for (int ci = 0; ci < NPC; ci++) {
        PairClass &P = c->pc[ci];
        if (P.recs.empty()) continue;
        if (upload(&P.d_recs, P.recs)) return -1;
        P.q.assign(P.recs.size(), 0.0);
        HIPCHK(dev_malloc(&P.d_q, sizeof(double) * P.recs.size()));
        EriArgs E{};
        setup_eri_dims(E, P.la, P.lb, P.la, P.lb);
        std::vector<uint32_t> comp;
        build_comp_table(P.la, P.lb, P.la, P.lb, comp);
        HIPCHK(hipMemcpyAsync(d_comp, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        E.bra = E.ket = P.d_recs; E.prim = c->d_prim; E.prefix = nullptr; E.nbra = (int)P.recs.size();
        E.comp = d_comp; E.work = d_work; E.rys = c->rys; E.diag = 1;
        size_t per = WORK_DOUBLES / E.ncomp;
        for (size_t b0 = 0; b0 < P.recs.size(); b0 += per) {
            int nb = (int)std::min(per, P.recs.size() - b0);
            E.t0 = (int64_t)b0; E.ntask = nb;
            if (launch_eri(c, E, nb, st)) return -1;
            SchwarzArgs S{P.d_recs + b0, c->d_M, d_work, P.ne, P.nsab, E.ncomp, P.d_q + b0};
            hipLaunchKernelGGL(schwarz_diag_kernel, dim3(nb), dim3(64), sizeof(double) * P.ne * P.ne, st, S);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(P.q.data(), P.d_q, sizeof(double) * P.q.size(), hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    for (int ci = 0; ci < NPC; ci++)
        for (double v : c->pc[ci].q) qmax = std::max(qmax, v);

    lap("schwarz");
    // The store of the previous geometry is parked and will most likely be taken again below: its zero-fill (100 GB, 15 ms for
    // ibuprofen) is queued NOW, so that it runs while the host sorts the pairs and plans tiles, runs and segments (~25 ms)
    // instead of after them.  (If the parked store turns out not to fit it is freed by arena_take and the fresh one is filled.)
    double *early_zero_ptr = nullptr;
    int64_t early_zero_doubles = 0;
    {
        TileArena &pk = g_arena[c->device & 15];
        if (pk.ptr && pk.doubles > 0 && !getenv("MI355_DEBUG")) {   // (the debug laps synchronise: keep their attribution)
            early_zero_ptr = pk.ptr; early_zero_doubles = pk.doubles;
            HIPCHK(hipMemsetAsync(pk.ptr, 0, sizeof(double) * (size_t)pk.doubles, st));
        }
    }
    // ---- 3. sort pairs by q (descending), drop negligible ones; block-pair Schwarz bounds
    const int nblk = c->nblk;
    const int nbp = nblk * (nblk + 1) / 2;
    std::vector<double> Qblk(nbp, 0.0);
    for (int ci = 0; ci < NPC; ci++) {
        PairClass &P = c->pc[ci];
        std::vector<int> ord(P.recs.size());
        std::iota(ord.begin(), ord.end(), 0);
        // Order of the pair list.  Tasks are (bra, ket) with the ket index running fastest, and a bra only visits the leading
        // kets whose bound can survive the Schwarz test, so the list must be non-increasing in a bound.  With `ket_cluster` that
        // bound is the maximum q of the pair's CLUSTER -- the pairs whose first AOs fall into the same (block, block) pair of
        // the store -- and inside a cluster the pairs follow their AO position: kets that complete one another's 64-byte lines
        // of a tile (neighbouring k shells / l shells) are then neighbouring tasks, i.e. neighbouring lanes of the thread-per-
        // quartet kernels and neighbouring workgroups of the others (which the XCD-aware block map keeps on one L2).
        std::vector<double> cqv(P.recs.size());
        std::vector<int64_t> cid(P.recs.size(), 0);
        if (c->opt_ket_cluster) {
            std::vector<std::pair<int64_t, double>> cm;   // cluster id -> max q over the surviving pairs
            cm.reserve(P.recs.size());
            for (size_t o = 0; o < P.recs.size(); o++) {
                cid[o] = (int64_t)(P.recs[o].ao_i / BLK) * c->nblk + P.recs[o].ao_j / BLK;
                if (P.q[o] * qmax >= tol) cm.push_back({cid[o], P.q[o]});
            }
            std::sort(cm.begin(), cm.end());
            std::vector<std::pair<int64_t, double>> cmax;
            for (const auto &e : cm) { if (cmax.empty() || cmax.back().first != e.first) cmax.push_back(e); else cmax.back().second = std::max(cmax.back().second, e.second); }
            for (size_t o = 0; o < P.recs.size(); o++) {
                auto it = std::lower_bound(cmax.begin(), cmax.end(), std::make_pair(cid[o], -1.0));
                cqv[o] = (it != cmax.end() && it->first == cid[o]) ? it->second : P.q[o];
            }
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) {
                if (cqv[a] != cqv[b]) return cqv[a] > cqv[b];
                if (cid[a] != cid[b]) return cid[a] < cid[b];
                if (P.recs[a].ao_i != P.recs[b].ao_i) return P.recs[a].ao_i < P.recs[b].ao_i;
                return P.recs[a].ao_j < P.recs[b].ao_j;
            });
        } else {
            for (size_t o = 0; o < P.recs.size(); o++) cqv[o] = P.q[o];
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return P.q[a] > P.q[b]; });
        }
        std::vector<PairRec> r2;
        std::vector<double> q2, cq2;
        for (int o : ord) {
            if (P.q[o] * qmax < tol) { if (c->opt_ket_cluster) continue; else break; }
            r2.push_back(P.recs[o]);
            q2.push_back(P.q[o]);
            cq2.push_back(cqv[o]);
            const PairRec &R = P.recs[o];
            int ni = 2 * P.la + 1, nj = 2 * P.lb + 1;
            for (int I = R.ao_i / BLK; I <= (R.ao_i + ni - 1) / BLK; I++)
                for (int J = R.ao_j / BLK; J <= (R.ao_j + nj - 1) / BLK; J++) {
                    int hi = std::max(I, J), lo = std::min(I, J);
                    double &Q = Qblk[hi * (hi + 1) / 2 + lo];
                    Q = std::max(Q, P.q[o]);
                }
        }
        P.cq.swap(cq2);
        P.recs.swap(r2);
        P.q.swap(q2);
        if (upload(&P.d_recs, P.recs)) return -1;
        if (upload(&P.d_q, P.q)) return -1; // Schwarz factors in the sorted order (density-weighted screening of the gradient)
        {
            double sm = 0.0, s4 = 0.0;
            const size_t np_ = P.recs.size();
            P.max_np = 1;
            for (size_t r = 0; r < np_; r++) { sm += P.recs[r].nprim; P.max_np = std::max(P.max_np, P.recs[r].nprim); }
            for (size_t r = 0; r < np_; r += 4) {
                int mx = 0;
                for (size_t u = r; u < std::min(np_, r + 4); u++) mx = std::max(mx, P.recs[u].nprim);
                s4 += mx;
            }
            P.mean_np = np_ ? sm / np_ : 1.0;
            P.max4_np = np_ ? s4 / ((np_ + 3) / 4) : 1.0;
        }
    }

    lap("sort pairs");
    // ---- 4. tiles and runs.  Run = tiles sharing (J,K,L), ordered by I.  Runs are the sharding unit: plan_runs deals them
    // to ranks longest-processing-time first by streamed bytes (every rank computes the same plan).
    std::vector<int> bpI(nbp), bpJ(nbp);
    for (int I = 0, n = 0; I < nblk; I++)
        for (int J = 0; J <= I; J++, n++) { bpI[n] = I; bpJ[n] = J; }
    auto bsize = [&](int B) { return std::min(BLK, c->nao - B * BLK); };
    c->tiles.clear(); c->tile_off.clear(); c->runs.clear();
    std::vector<int32_t> table((size_t)nbp * (nbp + 1) / 2, -1);
    int64_t off = 0, nuniq = 0;
    {
        std::vector<RunPlan> plan;
        c->tri = c->opt_tri_tiles != 0;
        plan_runs(c->nao, Qblk, qmax, tol, nranks, c->tri != 0, plan);
        std::vector<uint8_t> present(nranks > 1 ? table.size() : 0, 0); // sharded store: slots that live on SOME rank
        // first tile and first double of every run of this rank from the plan's counts / bytes, then the runs are filled by the
        // host threads (disjoint slices of the tile list, disjoint directory slots)
        const size_t nplan = plan.size();
        std::vector<int64_t> first_tile(nplan + 1, 0), first_off(nplan + 1, 0);
        std::vector<int> run_slot(nplan, -1);
        int nown = 0;
        for (size_t q = 0; q < nplan; q++) {
            const bool mine = plan[q].owner == rank;
            first_tile[q + 1] = first_tile[q] + (mine ? plan[q].count : 0);
            first_off[q + 1] = first_off[q] + (mine ? plan[q].bytes / 8 : 0);
            if (mine) run_slot[q] = nown++;
        }
        if (first_tile[nplan] >= INT32_MAX) return fail("too many tiles");
        c->tiles.resize((size_t)first_tile[nplan]);
        c->tile_off.resize((size_t)first_tile[nplan]);
        c->runs.resize((size_t)nown);
        int bad = 0;
#pragma omp parallel for schedule(dynamic, 256) num_threads(host_threads()) reduction(+ : nuniq) reduction(| : bad)
        for (size_t q = 0; q < nplan; q++) {
            const RunPlan &rp = plan[q];
            const int J = rp.J, K = rp.K, L = rp.L, kl = K * (K + 1) / 2 + L;
            if (rp.owner != rank) {
                for (int I = J; I < nblk; I++) {
                    int ij = I * (I + 1) / 2 + J;
                    if (ij >= kl && Qblk[ij] * Qblk[kl] >= tol) present[(size_t)ij * (ij + 1) / 2 + kl] = 1;
                }
                continue;
            }
            int tid = (int)first_tile[q];
            int64_t o = first_off[q];
            RunRec cur{J, K, L, tid, 0};
            for (int I = J; I < nblk; I++) {
                int ij = I * (I + 1) / 2 + J;
                if (ij < kl) continue;
                if (Qblk[ij] * Qblk[kl] < tol) continue;
                table[(size_t)ij * (ij + 1) / 2 + kl] = (int32_t)tid;
                c->tiles[tid] = {I, J, K, L};
                c->tile_off[tid] = o;
                int bi = bsize(I), bj = bsize(J), bk = bsize(K), bl = bsize(L);
                o += tile_doubles_padded(c->tri && I == J, c->tri && K == L, bi, bk); // j is always padded to 8 rows (J==last implies I==last: rare)
                int64_t nij = (I > J) ? (int64_t)bi * bj : (int64_t)bi * (bi + 1) / 2;
                int64_t nkl = (K > L) ? (int64_t)bk * bl : (int64_t)bk * (bk + 1) / 2;
                nuniq += (ij > kl) ? nij * nkl : nij * (nij + 1) / 2;
                cur.count++; tid++;
            }
            if (cur.count != rp.count || o != first_off[q + 1]) bad |= 1;
            c->runs[run_slot[q]] = cur;
        }
        if (bad) return fail("internal: run plan / tile enumeration mismatch");
        off = first_off[nplan];
        if (c->d_tile_present) { dev_free(c->d_tile_present); c->d_tile_present = nullptr; }
        if (nranks > 1 && upload(&c->d_tile_present, present)) return -1;
    }
    c->n_tiles = (int64_t)c->tiles.size();
    c->tile_doubles = off;
    if (c->n_tiles >= INT32_MAX) return fail("too many tiles");
    lap("  plan + tile enumeration");
    if (upload(&c->d_tile_table, table)) return -1;
    if (upload(&c->d_tile_off, c->tile_off)) return -1;
    {
        std::vector<int> tI(c->tiles.size());
        for (size_t i = 0; i < tI.size(); i++) tI[i] = c->tiles[i].I;
        if (upload(&c->d_tile_I, tI)) return -1;
        if (upload(&c->d_runs, c->runs)) return -1;
        lap("  directory uploads");
        // J/K work items ("segments"): runs cut into chunks; either one wave per item (longest first, the
        // hardware dispatcher balances) or a fixed number of waves with equal-cost contiguous shares.
        auto tile_cost = [&](int tid) {
            int64_t nd = (tid + 1 < (int)c->tile_off.size() ? c->tile_off[tid + 1] : off) - c->tile_off[tid];
            return (double)nd * 8.0 + 8192.0; // streamed bytes + fixed per-tile overhead
        };
        const int chunk = c->opt_runmax > 0 ? c->opt_runmax : (int)std::min<int64_t>(64, std::max<int64_t>(8, c->n_tiles / 2048));
        std::vector<RunRec> segs;
        std::vector<int> wave_seg;
        int nw, w = 0;
        if (c->opt_jk_waves <= 0) {
            for (const RunRec &r : c->runs)
                for (int t0 = 0; t0 < r.count; t0 += chunk) segs.push_back(RunRec{r.J, r.K, r.L, r.first + t0, std::min(chunk, r.count - t0)});
            std::vector<double> cost(segs.size(), 0.0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
            for (size_t q = 0; q < segs.size(); q++)
                for (int t = 0; t < segs[q].count; t++) cost[q] += tile_cost(segs[q].first + t);
            std::vector<int> ord(segs.size());
            std::iota(ord.begin(), ord.end(), 0);
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cost[a] > cost[b]; });
            std::vector<RunRec> sorted(segs.size());
            for (size_t q = 0; q < segs.size(); q++) sorted[q] = segs[ord[q]];
            segs.swap(sorted);
            c->n_jk_cached = 0;
            if (off * 8 > ((int64_t)256 << 20) && c->opt_jk_cache_mb > 0) {
                double acc_b = 0.0;
                const double lim = (double)c->opt_jk_cache_mb * 1048576.0;
                std::vector<double> sc(segs.size(), 0.0);
#pragma omp parallel for schedule(static) num_threads(host_threads())
                for (size_t q = 0; q < segs.size(); q++)
                    for (int t = 0; t < segs[q].count; t++) sc[q] += tile_cost(segs[q].first + t) - 8192.0;
                while (c->n_jk_cached < (int)segs.size() && acc_b + sc[c->n_jk_cached] <= lim) acc_b += sc[c->n_jk_cached++];
            }
            nw = (int)segs.size();
            wave_seg.resize(nw + 1);
            std::iota(wave_seg.begin(), wave_seg.end(), 0);
            w = nw - 1;
        } else {
        nw = (int)std::min<int64_t>(c->opt_jk_waves, std::max<int64_t>(c->n_tiles, 1));
        wave_seg.assign(nw + 1, 0);
        double total_cost = 0.0;
        for (int tid = 0; tid < (int)c->n_tiles; tid++) total_cost += tile_cost(tid);
        double per = total_cost / nw, acc_cost = 0.0;
        for (const RunRec &r : c->runs) {
            RunRec cur{r.J, r.K, r.L, r.first, 0};
            for (int t = 0; t < r.count; t++) {
                int tid = r.first + t;
                double tc = tile_cost(tid);
                int wt = std::min(nw - 1, (int)((acc_cost + 0.5 * tc) / per));
                acc_cost += tc;
                if (wt != w) {
                    if (cur.count) segs.push_back(cur);
                    for (int x = w + 1; x <= wt; x++) wave_seg[x] = (int)segs.size();
                    w = wt;
                    cur = RunRec{r.J, r.K, r.L, tid, 0};
                }
                cur.count++;
            }
            if (cur.count) segs.push_back(cur);
        }
        }
        for (int x = w + 1; x <= nw; x++) wave_seg[x] = (int)segs.size();
        c->n_jk_waves = nw;
        if (getenv("MI355_DEBUG")) {
            int mx = 0, mn = 1 << 30, maxseg = 0;
            for (int x = 0; x < nw; x++) {
                int nt = 0;
                for (int sgi = wave_seg[x]; sgi < wave_seg[x + 1]; sgi++) nt += segs[sgi].count;
                mx = std::max(mx, nt); mn = std::min(mn, nt); maxseg = std::max(maxseg, wave_seg[x + 1] - wave_seg[x]);
            }
            fprintf(stderr, "[mi355] tiles=%ld runs=%zu segs=%zu waves=%d tiles/wave min=%d max=%d maxsegs/wave=%d\n",
                    (long)c->n_tiles, c->runs.size(), segs.size(), nw, mn, mx, maxseg);
        }
        if (upload(&c->d_segs, segs)) return -1;
        if (upload(&c->d_wave_seg, wave_seg)) return -1;
    }
    lap("  J/K segments");
    size_t freeb = 0, totb = 0;
    HIPCHK(hipMemGetInfo(&freeb, &totb));
    freeb += (size_t)g_arena[c->device & 15].doubles * 8; // a parked store is reusable (or freed) by arena_take
    freeb += pool_parked_bytes(c->device);                // ... and so are the pool's parked blocks
    c->mem_need_bytes = (int64_t)off * 8;
    c->mem_free_bytes = (int64_t)freeb;
    if ((size_t)off * 8 + ((size_t)1 << 30) > freeb) {
        dev_free(d_comp);
        fail("resident ERI store needs %.1f GB but only %.1f GB of HBM is free; shard over more GPUs",
             off * 8e-9, freeb * 1e-9);
        return MI_ERR_NOMEM; // sizes: mi_eri_get_memory
    }
    c->tile_alloc = std::max<int64_t>(off, 1);
    {
        TileArena &pk = g_arena[c->device & 15];
        if (pk.ptr && pk.doubles >= c->tile_alloc && pk.doubles <= c->tile_alloc + c->tile_alloc / 4 + (1 << 20)) c->tile_alloc = pk.doubles;
        else {
            // a fresh allocation: 2 % of head room, so that the next geometry of an optimisation (a few more tiles survive the
            // Schwarz test) still fits the parked store -- re-allocating 103 GB took 6 s on a box of the pool (measured in a
            // geometry step of ibuprofen/def2-TZVP: 7.6 s instead of 1.9 s)
            const int64_t padded = c->tile_alloc + c->tile_alloc / 50;
            if ((size_t)padded * 8 + ((size_t)1 << 30) <= freeb) c->tile_alloc = padded;
        }
    }
    if (arena_take(c->device, c->tile_alloc, &c->d_tiles)) return -1;
    if (!(c->d_tiles == early_zero_ptr && off <= early_zero_doubles))
        HIPCHK(hipMemsetAsync(c->d_tiles, 0, sizeof(double) * std::max<int64_t>(off, 1), st));

    lap("tiles/runs/segments + alloc");
    // ---- 5. evaluate every Schwarz-surviving canonical shell quartet, class by class
    int64_t nquart = 0;
    int64_t *d_prefix = nullptr;
    size_t prefix_cap = 0;
    TaskIdx *d_tasks = nullptr;
    size_t tasks_cap = 0;
    for (int bc = 0; bc < NPC; bc++)
        for (int kc = 0; kc <= bc; kc++) {
            PairClass &B = c->pc[bc], &Kc = c->pc[kc];
            if (B.recs.empty() || Kc.recs.empty()) continue;
            std::vector<int64_t> prefix;
            const int64_t ntask = class_prefix(B, Kc, bc == kc, tol, prefix);
            if (ntask == 0) continue;
            nquart += ntask;
            if (prefix.size() > prefix_cap) {
                if (d_prefix) dev_free(d_prefix);
                prefix_cap = prefix.size() * 2;
                HIPCHK(dev_malloc(&d_prefix, sizeof(int64_t) * prefix_cap));
            }
            HIPCHK(hipMemcpyAsync(d_prefix, prefix.data(), sizeof(int64_t) * prefix.size(), hipMemcpyHostToDevice, st));
            EriArgs E{};
            setup_eri_dims(E, B.la, B.lb, Kc.la, Kc.lb);
            const bool tpq_class = c->opt_eri_tpq && B.mean_np * Kc.mean_np <= c->opt_tpq_maxprim && ntask >= 32768 &&
                                   tpq_has_class(B.la, B.lb, Kc.la, Kc.lb);
            if (c->opt_task_table && !tpq_class && ntask >= 65536) {   // the wave-per-quartet pair walks the task list twice
                if ((size_t)ntask > tasks_cap) {
                    if (scr_tasks.ensure(c->device, SCR_EVAL_TASKS, sizeof(TaskIdx) * (size_t)ntask)) return -1;
                    d_tasks = (TaskIdx *)scr_tasks.p;
                    tasks_cap = scr_tasks.bytes / sizeof(TaskIdx);
                }
                hipLaunchKernelGGL(fill_tasks_kernel, dim3((unsigned)((ntask + 255) / 256)), dim3(256), 0, st, d_prefix, (int)B.recs.size(), ntask, d_tasks);
                E.tasks = d_tasks;
            }
            std::vector<uint32_t> comp;
            build_comp_table(B.la, B.lb, Kc.la, Kc.lb, comp);
            HIPCHK(hipMemcpyAsync(d_comp, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st)); // host vectors go out of scope below
            E.bra = B.d_recs; E.ket = Kc.d_recs; E.prim = c->d_prim; E.prefix = d_prefix; E.nbra = (int)B.recs.size();
            E.comp = d_comp; E.work = d_work; E.rys = c->rys; E.diag = 0;
            E.ni = 2 * B.la + 1; E.nj = 2 * B.lb + 1; E.nk = 2 * Kc.la + 1; E.nl = 2 * Kc.lb + 1;
            E.own_table = nranks > 1 ? c->d_tile_table : nullptr;
            E.h_shared_np = B.mean_np; E.h_vary_mean = Kc.mean_np; E.h_vary_max4 = Kc.max4_np;
            E.q_bra = B.d_q; E.q_ket = Kc.d_q; E.qtol = c->opt_ket_cluster ? tol : 0.0;
            E.prim_lds = (c->opt_prim_lds && B.max_np + Kc.max_np <= 160) ? B.max_np + Kc.max_np : 0;
            const unsigned xcd_wave = c->opt_xcd_map ? 32u : 0u, xcd_tpq = c->opt_xcd_map ? 8u : 0u;
            E.xcd = xcd_wave;
            XfArgs X{};
            X.bra = B.d_recs; X.ket = Kc.d_recs; X.Mbuf = c->d_M; X.prefix = d_prefix; X.nbra = E.nbra;
            X.ne = B.ne; X.nf = Kc.ne; X.nsab = B.nsab; X.nscd = Kc.nsab; X.nsb = 2 * B.lb + 1; X.nsd = 2 * Kc.lb + 1;
            X.work = d_work; X.ncomp = E.ncomp; X.tile_table = c->d_tile_table; X.tile_off = c->d_tile_off;
            X.tiles = c->d_tiles; X.nao = c->nao; X.tri = c->tri;
            X.check_owner = nranks > 1; X.ni = E.ni; X.nj = E.nj; X.nk = E.nk; X.nl = E.nl;
            X.q_bra = E.q_bra; X.q_ket = E.q_ket; X.qtol = E.qtol; X.xcd = xcd_wave; X.tasks = E.tasks;
            int64_t per = std::min<int64_t>((int64_t)(WORK_DOUBLES / E.ncomp), (int64_t)1 << 24);
            size_t shm2 = sizeof(double) * ((size_t)X.ne * X.nf + (size_t)X.nsab * X.nf);
            {
                const size_t with_m = shm2 + sizeof(double) * ((size_t)2 * X.nsab * X.ne + (size_t)2 * X.nscd * X.nf);
                X.m_lds = (c->opt_xf_mlds && with_m <= (size_t)c->opt_xf_mlds * 1024) ? 1 : 0;
                if (X.m_lds) shm2 = with_m;
            }
            // matrix-core path only for the large classes: below, the lean per-lane kernel (56 VGPRs, 8 waves per SIMD) hides the
            // per-quartet latency chain better than MFMA tiles at 3-4 waves per SIMD (measured per class on ibuprofen/def2-TZVP)
            const bool xf_mfma = (mfma_worthwhile(X.nsab, X.nf, X.ne) || mfma_worthwhile(X.nsab, X.nscd, X.nf)) &&
                                 X.nsab * X.nscd >= c->opt_xf_mfma_min;
            const bool xf_small = !xf_mfma && X.nsab * X.nscd <= c->opt_xf_qpw_max && shm2 * 4 <= 64 * 1024; // four quartets per wave
            if (shm2 > 64 * 1024)
                HIPCHK(hipFuncSetAttribute(xf_mfma ? (const void *)eri_transform_scatter<true, 64> : (const void *)eri_transform_scatter<false, 64>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2));
            const bool dbg = getenv("MI355_DEBUG") != nullptr && getenv("MI355_DEBUG")[0] == '2';
            double t_rys = 0.0, t_xf = 0.0;
            // low angular classes: one thread per quartet, fused Rys + HRR + cart->sph + scatter (eri_tpq_kernel)
            // ... when the contraction is shallow enough: a thread walks ALL primitive quartets of its shell quartet, so deeply
            // contracted shells (cc-pVXZ s shells: hundreds of primitive quartets) serialise and the wave-per-quartet path, which
            // spreads them over lanes, wins (measured: benzene/cc-pVTZ 54 vs 63 ms, ibuprofen/def2-TZVP 0.40 vs 0.29 s)
            if (c->opt_eri_tpq && B.mean_np * Kc.mean_np <= c->opt_tpq_maxprim && ntask >= 32768) {
                TpqArgs Q{};
                Q.bra = B.d_recs; Q.ket = Kc.d_recs; Q.prim = c->d_prim; Q.prefix = d_prefix; Q.nbra = E.nbra; Q.t0 = 0; Q.ntask = ntask;
                Q.c2s = c->d_c2s;
                for (int q = 0; q <= LMAX + 1; q++) Q.c2s_off[q] = c->c2s_off[q];
                Q.rys = c->rys; Q.X = X; Q.shell_xyz = c->d_shell_xyz; Q.check_owner = nranks > 1;
                Q.q_bra = E.q_bra; Q.q_ket = E.q_ket; Q.qtol = E.qtol; Q.xcd = xcd_tpq;
                auto ta = std::chrono::steady_clock::now();
                if (dbg) hipStreamSynchronize(st);
                const int used = launch_eri_tpq(B.la, B.lb, Kc.la, Kc.lb, Q, st);
                if (used < 0) return -1;
                if (used) {
                    if (dbg) {
                        hipStreamSynchronize(st);
                        t_rys = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
                        fprintf(stderr, "[mi355] eri class (%d%d|%d%d): %ld quartets, thread-per-quartet fused kernel %.4f s (%.2f ns/q)\n", B.la, B.lb,
                                Kc.la, Kc.lb, (long)ntask, t_rys, t_rys / ntask * 1e9);
                    }
                    continue;
                }
            }
            if (c->opt_eri_fused && !E.diag) {
                // one fused launch per (at most 2^30) tasks: no hand-over buffer, so no batching by its size
                FusedArgs F;
                F.R = E; F.X = X;
                F.R.PB = std::max(1, std::min(E.PB, B.max_np * Kc.max_np));   // uncontracted d / f shells: one primitive quartet per batch
                F.R.prim_lds = 0; F.R.own_table = nullptr; F.R.qtol = 0.0; F.R.dmax = nullptr;   // (the transform half screens: check_owner, qtol)
                F.X.m_lds = 0;
                const size_t scratch = (size_t)F.R.PB * E.nroots * 3 * E.tsz + (size_t)F.R.PB * 2 * E.nroots;
                const size_t shmf = sizeof(double) * ((size_t)X.ne * X.nf + std::max((size_t)X.nsab * X.nf, scratch));
                if (shmf <= 160 * 1024) {
                    const int perlane = (E.ncomp + 63) / 64;
                    auto ta = std::chrono::steady_clock::now();
                    if (dbg) hipStreamSynchronize(st);
                    for (int64_t t0 = 0; t0 < ntask; t0 += (int64_t)1 << 30) {
                        const int64_t nb = std::min<int64_t>((int64_t)1 << 30, ntask - t0);
                        F.X.t0 = t0; F.X.ntask = nb; F.X.xcd = nb >= 2048 ? xcd_wave : 0u;
                        const dim3 grid(eri_grid(nb, F.X.xcd));
#define FUSED_LAUNCH(MF, MC)                                                                                                               \
    do {                                                                                                                                   \
        if (shmf > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)eri_fused_kernel<MF, MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmf)); \
        hipLaunchKernelGGL((eri_fused_kernel<MF, MC>), grid, dim3(64), shmf, st, F);                                                        \
    } while (0)
                        if (xf_mfma) {
                            if (perlane <= 4) FUSED_LAUNCH(true, 4); else if (perlane <= 16) FUSED_LAUNCH(true, 16); else FUSED_LAUNCH(true, 32);
                        } else {
                            if (perlane <= 1) FUSED_LAUNCH(false, 1); else if (perlane <= 4) FUSED_LAUNCH(false, 4);
                            else if (perlane <= 16) FUSED_LAUNCH(false, 16); else FUSED_LAUNCH(false, 32);
                        }
#undef FUSED_LAUNCH
                        HIPCHK(hipGetLastError());
                    }
                    if (dbg) {
                        hipStreamSynchronize(st);
                        t_rys = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
                        fprintf(stderr, "[mi355] eri class (%d%d|%d%d): %ld quartets, fused rys+transform+scatter %.4f s (%.2f ns/q)\n", B.la, B.lb,
                                Kc.la, Kc.lb, (long)ntask, t_rys, t_rys / ntask * 1e9);
                    }
                    continue;
                }
            }
            for (int64_t t0 = 0; t0 < ntask; t0 += per) {
                int nb = (int)std::min<int64_t>(per, ntask - t0);
                E.t0 = t0; E.ntask = nb; X.t0 = t0;
                auto ta = std::chrono::steady_clock::now();
                if (dbg) hipStreamSynchronize(st);
                if (launch_eri(c, E, nb, st)) return -1;
                if (dbg) { hipStreamSynchronize(st); t_rys += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count(); ta = std::chrono::steady_clock::now(); }
                X.ntask = nb;
                X.xcd = nb >= 2048 ? xcd_wave : 0u;
                E.xcd = xcd_wave;   // (launch_eri switches the map off for small launches)
                if (xf_mfma) hipLaunchKernelGGL((eri_transform_scatter<true, 64>), dim3(eri_grid(nb, X.xcd)), dim3(64), shm2, st, X);
                else if (xf_small) hipLaunchKernelGGL((eri_transform_scatter<false, 16>), dim3(eri_grid((nb + 3) / 4, X.xcd)), dim3(64), shm2 * 4, st, X);
                else hipLaunchKernelGGL((eri_transform_scatter<false, 64>), dim3(eri_grid(nb, X.xcd)), dim3(64), shm2, st, X);
                HIPCHK(hipGetLastError());
                if (dbg) { hipStreamSynchronize(st); t_xf += std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count(); }
            }
            if (dbg)
                fprintf(stderr, "[mi355] eri class (%d%d|%d%d): %ld quartets, rys %.4f s (%.2f ns/q), transform+scatter %.4f s (%.2f ns/q)\n", B.la, B.lb,
                        Kc.la, Kc.lb, (long)ntask, t_rys, t_rys / ntask * 1e9, t_xf, t_xf / ntask * 1e9);
        }
    HIPCHK(hipStreamSynchronize(st));
    lap("quartet evaluation");
    if (d_prefix) dev_free(d_prefix);
    dev_free(d_comp);
    scr_work.release(); scr_tasks.release();
    lap("free scratch");
    c->stats.n_tiles = c->n_tiles;
    c->stats.n_runs = (int64_t)c->runs.size();
    c->stats.stored_bytes = off * 8;
    c->stats.n_unique_eri = nuniq;
    c->stats.n_quartets = nquart;
    c->stats.seconds_eri = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    c->eri_ready = true;
    if (c->opt_grad_prefetch) {
        c->grad_host_rc = 0;
        c->grad_worker = std::thread([c]() {
            c->grad_host_rc = grad_records_host(c);
            if (c->grad_host_rc) c->grad_host_err = mi_last_error();   // (thread-local error string of the helper)
        });
    }
    return 0;
}

// Schwarz factors q[ish][jsh] = sqrt(max |(ab|ab)|) of the shell pairs kept by the last mi_eri_prepare (0 for pairs dropped
// as negligible: q * q_max < tol), host array [nbas][nbas] (tests: against the oracle's restatement of CVHFnr_int2e_q_cond).
extern "C" int mi_schwarz_get(const mi_ctx *c, double *q)
{
    if (!c || !q) return fail("mi_schwarz_get: null argument");
    if (!c->eri_ready) return fail("mi_schwarz_get: call mi_eri_prepare first");
    std::fill(q, q + (size_t)c->nbas * c->nbas, 0.0);
    for (int ci = 0; ci < NPC; ci++) {
        const PairClass &P = c->pc[ci];
        for (size_t r = 0; r < P.recs.size(); r++) {
            q[(size_t)P.recs[r].sh_i * c->nbas + P.recs[r].sh_j] = P.q[r];
            q[(size_t)P.recs[r].sh_j * c->nbas + P.recs[r].sh_i] = P.q[r];
        }
    }
    return 0;
}

extern "C" int mi_eri_get_memory(const mi_ctx *c, int64_t *need_bytes, int64_t *free_bytes)
{
    if (!c || !need_bytes || !free_bytes) return fail("mi_eri_get_memory: null argument");
    *need_bytes = c->mem_need_bytes;
    *free_bytes = c->mem_free_bytes;
    return 0;
}

// =================================================================================================
// Density fitting: three-index (ij|P) and two-index (P|Q) Coulomb integrals over an auxiliary
// basis with the SAME Rys kernels -- an auxiliary function P is handled as the "shell pair" (P, unit s function), i.e. a
// four-centre quartet (ij|P 1).  Stands in for libcint int3c2e_sph / int2c2e_sph behind `mf.density_fit()` [MEM]; the
// contractions with the density (J, K) are dense FP64 GEMMs done by the caller.
// `aux` is an ordinary context built from the auxiliary basis whose LAST shell is the unit function: an s primitive with
// exponent 0 and coefficient sqrt(4 pi) (so that coefficient * Y_00 = 1).
// d_int3c: [nao][nao][naux] (both (i,j) and (j,i) written), d_int2c: [naux][naux]; either may be NULL.
// =================================================================================================
struct DfPairs {
    std::vector<PairRec> recs[NPC];       // orbital shell pairs by class
    std::vector<PairRec> aux[LMAX + 1];   // (P, unit) "pairs" by l_P
    std::vector<double> prim, Mbuf;
};

extern "C" int mi_df_build(mi_ctx *c, mi_ctx *aux, double *d_int3c, double *d_int2c, void *stream)
{
    if (!c || !aux) return fail("mi_df_build: null context");
    if (aux->nbas < 2) return fail("mi_df_build: the auxiliary context needs at least one function plus the unit shell");
    const ShellH &U = aux->shells.back();
    if (U.l != 0 || U.nprim != 1 || U.exps[0] != 0.0) return fail("mi_df_build: the last auxiliary shell must be the unit s function (exponent 0)");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int naux = aux->nao - 1, unit_ao = aux->nao - 1, nP = aux->nbas - 1;
    std::vector<std::vector<double>> c2s(LMAX + 1);
    for (int l = 0; l <= LMAX; l++) c2s_generic(l, c2s[l]);
    DfPairs D;
    // orbital pairs (as in mi_eri_prepare step 1, no Schwarz sorting: every pair is fitted)
    for (int A = 0; A < c->nbas; A++)
        for (int B = 0; B <= A; B++) {
            int si = A, sj = B;
            if (c->shells[si].l < c->shells[sj].l) std::swap(si, sj);
            const ShellH &I = c->shells[si], &J = c->shells[sj];
            double AB[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
            double r2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
            PairRec R{si, sj, I.ao_nat, J.ao_nat, (int)(D.prim.size() / 8), 0, 0, 0};   // dense output in the caller's AO order
            for (int ip = 0; ip < I.nprim; ip++)
                for (int jp = 0; jp < J.nprim; jp++) {
                    double a = I.exps[ip], b = J.exps[jp], p = a + b, mu = a * b / p;
                    if (mu * r2 > 80.0) continue;
                    double K = I.coef[ip] * J.coef[jp] * std::exp(-mu * r2), Pc[3];
                    for (int d = 0; d < 3; d++) Pc[d] = (a * I.r[d] + b * J.r[d]) / p;
                    double rec[8] = {p, Pc[0], Pc[1], Pc[2], Pc[0] - I.r[0], Pc[1] - I.r[1], Pc[2] - I.r[2], K};
                    D.prim.insert(D.prim.end(), rec, rec + 8);
                    R.nprim++;
                }
            if (R.nprim == 0) continue;
            const int nsab = (2 * I.l + 1) * (2 * J.l + 1), ne = ne_of(I.l, J.l);
            R.m_off = (int)D.Mbuf.size();
            D.Mbuf.resize(D.Mbuf.size() + (size_t)2 * nsab * ne);
            build_M(I.l, J.l, AB, c2s[I.l], c2s[J.l], D.Mbuf.data() + R.m_off);
            double *M = D.Mbuf.data() + R.m_off, *Mt = M + (size_t)nsab * ne;
            for (int r = 0; r < nsab; r++)
                for (int e = 0; e < ne; e++) Mt[(size_t)e * nsab + r] = M[(size_t)r * ne + e];
            D.recs[pc_index(I.l, J.l)].push_back(R);
        }
    // auxiliary "pairs" (P, unit): p = alpha, centre P = A, P - A = 0, K = c_P * c_unit
    for (int Pn = 0; Pn < nP; Pn++) {
        const ShellH &S = aux->shells[Pn];
        PairRec R{Pn, aux->nbas - 1, S.ao_nat, unit_ao, (int)(D.prim.size() / 8), S.nprim, 0, 0};
        for (int ip = 0; ip < S.nprim; ip++) {
            double rec[8] = {S.exps[ip], S.r[0], S.r[1], S.r[2], 0.0, 0.0, 0.0, S.coef[ip] * U.coef[0]};
            D.prim.insert(D.prim.end(), rec, rec + 8);
        }
        const int ns = 2 * S.l + 1, ne = ncart(S.l);
        double AB[3] = {0.0, 0.0, 0.0};
        R.m_off = (int)D.Mbuf.size();
        D.Mbuf.resize(D.Mbuf.size() + (size_t)2 * ns * ne);
        build_M(S.l, 0, AB, c2s[S.l], c2s[0], D.Mbuf.data() + R.m_off);
        double *M = D.Mbuf.data() + R.m_off, *Mt = M + (size_t)ns * ne;
        for (int r = 0; r < ns; r++)
            for (int e = 0; e < ne; e++) Mt[(size_t)e * ns + r] = M[(size_t)r * ne + e];
        // c2s of the unit function: build_M multiplied by c2s[0] = 1/sqrt(4 pi); its coefficient sqrt(4 pi) restores 1
        D.aux[S.l].push_back(R);
    }
    if (D.Mbuf.size() > (size_t)INT32_MAX) return fail("mi_df_build: transformation-matrix buffer exceeds 2^31 doubles");
    double *d_prim = nullptr, *d_M = nullptr, *d_work = nullptr;
    uint32_t *d_comp = nullptr;
    int64_t *d_prefix = nullptr;
    PairRec *d_rec_o[NPC] = {nullptr}, *d_rec_a[LMAX + 1] = {nullptr};
    if (upload(&d_prim, D.prim) || upload(&d_M, D.Mbuf)) return -1;
    for (int q = 0; q < NPC; q++) if (!D.recs[q].empty() && upload(&d_rec_o[q], D.recs[q])) return -1;
    for (int l = 0; l <= LMAX; l++) if (!D.aux[l].empty() && upload(&d_rec_a[l], D.aux[l])) return -1;
    const size_t WORK_DOUBLES = (size_t)32 << 20;
    HIPCHK(dev_malloc(&d_work, sizeof(double) * WORK_DOUBLES));
    HIPCHK(dev_malloc(&d_comp, sizeof(uint32_t) * 8192));
    size_t prefix_cap = 0;
    // one pass per (bra class, auxiliary l): bra = orbital pairs (3-index) or auxiliary pairs (2-index)
    auto run = [&](const PairRec *d_bra, int nbra, int la, int lb, int lk, int mode, double *out, int ld_nao) -> int {
        const int nket = (int)D.aux[lk].size();
        if (nbra == 0 || nket == 0) return 0;
        std::vector<int64_t> prefix(nbra + 1);
        for (int b = 0; b <= nbra; b++) prefix[b] = (int64_t)b * nket;
        const int64_t ntask = prefix.back();
        append_coarse_index(prefix);
        if (prefix.size() > prefix_cap) {
            if (d_prefix) dev_free(d_prefix);
            prefix_cap = prefix.size() * 2;
            HIPCHK(dev_malloc(&d_prefix, sizeof(int64_t) * prefix_cap));
        }
        HIPCHK(hipMemcpyAsync(d_prefix, prefix.data(), sizeof(int64_t) * prefix.size(), hipMemcpyHostToDevice, st));
        EriArgs E{};
        setup_eri_dims(E, la, lb, lk, 0);
        std::vector<uint32_t> comp;
        build_comp_table(la, lb, lk, 0, comp);
        HIPCHK(hipMemcpyAsync(d_comp, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        E.bra = d_bra; E.ket = d_rec_a[lk]; E.prim = d_prim; E.prefix = d_prefix; E.nbra = nbra;
        E.comp = d_comp; E.work = d_work; E.rys = c->rys; E.diag = 0;
        E.ni = 2 * la + 1; E.nj = 2 * lb + 1; E.nk = 2 * lk + 1; E.nl = 1;
        XfArgs X{};
        X.bra = d_bra; X.ket = d_rec_a[lk]; X.Mbuf = d_M; X.prefix = d_prefix; X.nbra = nbra;
        X.ne = ne_of(la, lb); X.nf = ncart(lk); X.nsab = (2 * la + 1) * (2 * lb + 1); X.nscd = 2 * lk + 1; X.nsb = 2 * lb + 1; X.nsd = 1;
        X.work = d_work; X.ncomp = E.ncomp; X.nao = ld_nao; X.check_owner = 0;
        X.ni = E.ni; X.nj = E.nj; X.nk = E.nk; X.nl = 1;
        X.dense_out = out; X.dense_mode = mode; X.dense_n = naux;
        const size_t shm2 = sizeof(double) * ((size_t)X.ne * X.nf + (size_t)X.nsab * X.nf);
        const int64_t per = std::min<int64_t>((int64_t)(WORK_DOUBLES / E.ncomp), (int64_t)1 << 22);
        for (int64_t t0 = 0; t0 < ntask; t0 += per) {
            const int nb = (int)std::min<int64_t>(per, ntask - t0);
            E.t0 = t0; E.ntask = nb; X.t0 = t0; X.ntask = nb;
            if (launch_eri(c, E, nb, st)) return -1;
            hipLaunchKernelGGL((eri_transform_scatter<false, 64>), dim3(nb), dim3(64), shm2, st, X);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    };
    int rc = 0;
    if (d_int3c)
        for (int la = 0; la <= LMAX && !rc; la++)
            for (int lb = 0; lb <= la && !rc; lb++)
                for (int lk = 0; lk <= LMAX && !rc; lk++) {
                    const int q = pc_index(la, lb);
                    rc = run(d_rec_o[q], (int)D.recs[q].size(), la, lb, lk, 1, d_int3c, c->nao);
                }
    if (d_int2c)
        for (int lp = 0; lp <= LMAX && !rc; lp++)
            for (int lk = 0; lk <= LMAX && !rc; lk++) rc = run(d_rec_a[lp], (int)D.aux[lp].size(), lp, 0, lk, 2, d_int2c, naux);
    for (int q = 0; q < NPC; q++) if (d_rec_o[q]) dev_free(d_rec_o[q]);
    for (int l = 0; l <= LMAX; l++) if (d_rec_a[l]) dev_free(d_rec_a[l]);
    dev_free(d_prim); dev_free(d_M); dev_free(d_work); dev_free(d_comp);
    if (d_prefix) dev_free(d_prefix);
    return rc;
}

extern "C" int mi_eri_get_stats(const mi_ctx *c, mi_eri_stats *out)
{
    if (!c || !out) return fail("mi_eri_get_stats: null argument");
    *out = c->stats;
    return 0;
}

// =================================================================================================
// J/K digestion of the resident tiles.
//
// One wave per run (tiles sharing J,K,L; I varies).  Lane (i,k) = (lane>>3, lane&7) owns the 8x8
// (j,l) sub-block T[i,:,k,:] of each tile and performs all six contractions per loaded value:
//   per tile  : K_IK (in-lane), J_IJ and K_IL (reduce over the 8 k-lanes)
//   per run   : J_KL and K_JK (reduce over the 8 i-lanes), K_JL (reduce over all 64 lanes)
// Partial blocks are added with FP64 global atomics into padded accumulators; the finalize kernel
// forms J = 2 (Jacc + Jacc^T), K = Kacc + Kacc^T (tile values are pre-weighted, see put_tile).
// =================================================================================================
typedef double d2_t __attribute__((ext_vector_type(2)));

struct JkArgs {
    const double *tiles;
    const int64_t *tile_off;
    const int *tile_I;
    const RunRec *runs;      // segments (runs cut at wave boundaries)
    const int *wave_seg;     // [nwaves+1]
    int nruns;               // number of waves
    const double *D; // padded [ldp][ldp]
    double *Jacc, *Kacc;
    int ld, nao;
    int n_cached;            // leading work items read with the default cache policy (kept in the Infinity Cache)
    int tri;                 // triangular rows in block-diagonal tiles (tile geometry)
    size_t pair_stride;      // jk_tiles_pair_kernel: doubles between the two padded densities / accumulator sets
    int pair_sync;           // barrier per tile between the two waves
    int dpp;                 // per-tile reduce-scatters on the DPP path (reduce8_low) instead of ds_bpermute
};

__device__ inline double red_select_xor(double a, double b, bool hi, int mask)
{
    // keep = hi ? b : a ; send = hi ? a : b ; returns keep + partner's send
    double keep = hi ? b : a, send = hi ? a : b;
    return keep + __shfl_xor(send, mask);
}

// reduce-scatter 8 values over the 3 lane bits {m2, m1, m0}; afterwards the lane holds element
// ((lane&m2)?4:0)+((lane&m1)?2:0)+((lane&m0)?1:0)
__device__ inline double reduce8(const double v[8], int lane, int m2, int m1, int m0)
{
    double a[4], b[2];
    bool h2 = lane & m2, h1 = lane & m1, h0 = lane & m0;
#pragma unroll
    for (int t = 0; t < 4; t++) a[t] = red_select_xor(v[t], v[t + 4], h2, m2);
#pragma unroll
    for (int t = 0; t < 2; t++) b[t] = red_select_xor(a[t], a[t + 2], h1, m1);
    return red_select_xor(b[0], b[1], h0, m0);
}

// The same reduce-scatter over the three LOW lane bits (the 8 k-lanes of one i) on the DPP data path instead of the LDS crossbar
// (ds_bpermute: two per double and a ~100-cycle round trip per dependent step, fully exposed at one wave per SIMD).  Step 1 pairs
// lane b with 7 - b (row_half_mirror) instead of b ^ 4: both have opposite bit 2, which is all the first step needs; steps 2 and 3
// (quad_perm) then combine disjoint groups exactly as the xor pattern does, so the lane -> element map is unchanged.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    union { double d; int w[2]; } u, r;
    u.d = v;
    r.w[0] = __builtin_amdgcn_update_dpp(0, u.w[0], CTRL, 0xF, 0xF, false);
    r.w[1] = __builtin_amdgcn_update_dpp(0, u.w[1], CTRL, 0xF, 0xF, false);
    return r.d;
}
__device__ __forceinline__ double reduce8_low(const double v[8], int lane)
{
    constexpr int HALF_MIRROR = 0x141, QP_XOR2 = 0x4E, QP_XOR1 = 0xB1;
    double a[4], b[2];
    const bool h2 = lane & 4, h1 = lane & 2, h0 = lane & 1;
#pragma unroll
    for (int t = 0; t < 4; t++) a[t] = (h2 ? v[t + 4] : v[t]) + dpp_move<HALF_MIRROR>(h2 ? v[t] : v[t + 4]);
#pragma unroll
    for (int t = 0; t < 2; t++) b[t] = (h1 ? a[t + 2] : a[t]) + dpp_move<QP_XOR2>(h1 ? a[t] : a[t + 2]);
    return (h0 ? b[1] : b[0]) + dpp_move<QP_XOR1>(h0 ? b[0] : b[1]);
}

// One 16-byte chunk {T[i,j,k,2lp], T[i,j,k,2lp+1]} of the lane's sub-block.  DIJ / DKL: the tile lies on the I == J / K == L
// block diagonal and stores triangular rows (tile geometry above); for <false,false> this is the plain
// T[(j*4+lp)*bi*bk + i*bk + k] of a full tile.
template <bool DIJ, bool DKL, bool NT>
__device__ __forceinline__ d2_t jk_load_chunk(const d2_t *__restrict__ tile, int j, int lp, int bi, int bk, int i, int k, bool inb)
{
    d2_t x = {0.0, 0.0};
    if (!DIJ && !DKL) {
        if (inb) { const d2_t *q = tile + (i * bk + k) + (size_t)(j * 4 + lp) * (bi * bk); x = NT ? __builtin_nontemporal_load(q) : *q; }
    } else {
        const int i0 = tile_i0(DIJ, j, bi), k0 = tile_k0(DKL, lp, bk);
        const int row = tile_pi(DIJ, j, bi) * tile_qk(DKL, 4, bk) + (bi - i0) * tile_qk(DKL, lp, bk);   // wave-uniform
        if (inb && i >= i0 && k >= k0) { const d2_t *q = tile + row + (i - i0) * (bk - k0) + (k - k0); x = NT ? __builtin_nontemporal_load(q) : *q; }
    }
    return x;
}

// Digestion of one tile by one wave: 32 chunk loads per lane, six contractions per value, per-tile outputs.
// Register budget (J+K, gfx950 ISA): 256 VGPRs + 158 AGPRs, one wave per SIMD; 128 of them are the run-wide K_JL accumulators.
// The compiled tile body is ~1 700 instructions for 384 FMAs: each 16-byte load costs 9 (exec mask for partial blocks, the
// row stride read back from a spilled SGPR, a 64-bit multiply-add), the rest is AGPR traffic, selects and ds_bpermute of the
// reduce-scatters.  A specialisation for full 8x8x8x8 tiles (immediate offsets, no masks) let the scheduler hoist all 32 loads,
// which pushed the kernel to 512 registers and 2.23 ms (from 0.78) -- rejected; scheduling barriers around the tile body did not
// help (498 registers, 2.19 ms).  For the Coulomb-only build the same specialisation is harmless but its gain (0.742 vs 0.754-0.772
// ms on one box, 0.770 vs 0.768-0.772 on another; +8 % on the cache-resident cc-pVDZ tensor) is within box-to-box noise: not kept.
// Any rewrite has to bound the loads in flight by construction.
// Row split (round 3, measured and removed; commit "Experiment: J+K digestion with the j rows of a tile split between two waves"):
// a workgroup of two waves, wave w owning T[i, 4w..4w+3, k, :], 32 K_JL accumulators and partial K_IK / K_IL sums merged through
// LDS (one barrier per tile), fits 256 registers (two waves per SIMD, 132 bytes of scratch) and gives the same J, K, but every
// per-tile fixed cost (density rows, reduce-scatters, directory reads) is paid by both waves: 1.14 ms with the LDS merge, 1.21 ms
// with both waves issuing their own atomics, against 0.86 ms for the one-wave kernel on the same box (benzene/cc-pVTZ).
// KJLT (round 3): K_JL is reduced over the 64 lanes PER TILE (8 DPP reduce-scatters over the k-lanes while the rows stream in,
// one reduce-scatter over the i-lanes, one more 512-byte atomic) instead of living in 64 run-wide accumulators per lane: 128
// registers less, which is what keeps the K-carrying kernel at one wave per SIMD.
template <bool WITH_J, bool WITH_K, bool NT, bool DIJ, bool DKL, bool KJLT = false>
__device__ __forceinline__ void jk_digest_tile(const JkArgs &A, const int lane, const int i, const int k, const int I0, const int J0,
                                               const int K0, const int L0, const int ld, const int bk, const int64_t toff,
                                               const double (&dKL)[8], const double (&dJK)[8], double (&kjl)[8][8], double (&jkl)[8],
                                               double (&kjk)[8])
{
    const double *__restrict__ D = A.D;
    const MI_CONST_AS double *Du = as_const(A.D);
    const int bi = min(BLK, A.nao - I0);
    const bool active = (i < bi) && (k < bk);
    const d2_t *__restrict__ T = reinterpret_cast<const d2_t *>(A.tiles + toff);
    double dIJ[8], dIL[8];
#pragma unroll
    for (int j = 0; j < 8; j++) dIJ[j] = D[(size_t)(I0 + i) * ld + J0 + j];
#pragma unroll
    for (int l = 0; l < 8; l++) dIL[l] = D[(size_t)(I0 + i) * ld + L0 + l];
    const double dIK = D[(size_t)(I0 + i) * ld + K0 + k];
    double kik = 0.0, jij[8], kil[8], kjl_row[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { jij[j] = 0.0; kil[j] = 0.0; kjl_row[j] = 0.0; }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        double v[8];
#pragma unroll
        for (int lp = 0; lp < 4; lp++) {
            const d2_t x = jk_load_chunk<DIJ, DKL, NT>(T, j, lp, bi, bk, i, k, active);
            v[2 * lp] = x.x; v[2 * lp + 1] = x.y;
        }
        const MI_CONST_AS double *dJL = Du + (size_t)(J0 + j) * ld + L0; // wave-uniform row (SGPRs)
        double pjl[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            const double x = v[l];
            if (WITH_K) {
                kik = fma(x, dJL[l], kik);
                kil[l] = fma(x, dJK[j], kil[l]);
                if (KJLT) pjl[l] = x * dIK; else kjl[j][l] = fma(x, dIK, kjl[j][l]);
                kjk[j] = fma(x, dIL[l], kjk[j]);
            }
            if (WITH_J) {
                jij[j] = fma(x, dKL[l], jij[j]);
                jkl[l] = fma(x, dIJ[j], jkl[l]);
            }
        }
        if (WITH_K && KJLT) kjl_row[j] = reduce8_low(pjl, lane);   // sum over the 8 k-lanes of this i; lane holds l = k
    }
    if (WITH_K && KJLT) {
        const double r = reduce8(kjl_row, lane, 32, 16, 8);          // sum over the 8 i-lanes; lane holds j = i
        atomicAdd(&A.Kacc[(size_t)(J0 + i) * ld + L0 + k], r);
    }
    // per-tile outputs: three wave-wide FP64 atomics (1.5 KB added per 32 KB tile; executed at the memory side, not in L2).
    // Their cost, measured by dropping them (wrong J/K, timing only; benzene/cc-pVTZ, same box): J+K launch 0.752 ms without,
    // 0.789-0.84 ms with -- 5-7 % of the launch, about what a scratch buffer of plain stores plus a reduction pass would cost
    // (the same 0.24 GB written, then read), so the atomics stay.
    if (WITH_K) {
        atomicAdd(&A.Kacc[(size_t)(I0 + i) * ld + K0 + k], kik);
        double r = A.dpp ? reduce8_low(kil, lane) : reduce8(kil, lane, 4, 2, 1); // lane holds l = k
        atomicAdd(&A.Kacc[(size_t)(I0 + i) * ld + L0 + k], r);
    }
    if (WITH_J) {
        double r = A.dpp ? reduce8_low(jij, lane) : reduce8(jij, lane, 4, 2, 1); // lane holds j = k
        atomicAdd(&A.Jacc[(size_t)(I0 + i) * ld + J0 + k], r);
    }
}

template <bool WITH_J, bool WITH_K, bool NT, bool PAIR = false, bool KJLT = false>
__device__ __forceinline__ void jk_segment(const JkArgs &A, const int seg)
{
    const int lane = threadIdx.x & 63;
    const int i = lane >> 3, k = lane & 7;
    const MI_CONST_AS int *tile_I = as_const(A.tile_I);
    const MI_CONST_AS int64_t *tile_off = as_const(A.tile_off);
  {
    const MI_CONST_AS RunRec *rr = as_const(A.runs) + seg;
    const RunRec R{rr->J, rr->K, rr->L, rr->first, rr->count};
    const int J0 = R.J * BLK, K0 = R.K * BLK, L0 = R.L * BLK;
    const int ld = A.ld;
    const int bk = min(BLK, A.nao - K0); // tiles are padded to 8 j-rows
    const double *__restrict__ D = A.D;

    // run-invariant density rows
    double dKL[8], dJK[8];
#pragma unroll
    for (int l = 0; l < 8; l++) dKL[l] = D[(size_t)(K0 + k) * ld + L0 + l];
#pragma unroll
    for (int j = 0; j < 8; j++) dJK[j] = D[(size_t)(J0 + j) * ld + K0 + k];
    double kjl[8][8], jkl[8], kjk[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        jkl[j] = 0.0; kjk[j] = 0.0;
#pragma unroll
        for (int l = 0; l < 8; l++) kjl[j][l] = 0.0;
    }

    // the directory entry of tile t+1 is fetched while tile t is being digested
    int I_next = tile_I[R.first];
    int64_t off_next = tile_off[R.first];
    int t = 0;
    // only the first tile of a run can lie on the I == J diagonal (tiles of a run are ordered by I >= J); K == L holds for
    // the whole run or not at all
    if (A.tri && R.K == R.L) {
        if (I_next == R.J) {
            const int I0 = I_next * BLK;
            const int64_t toff = off_next;
            if (R.count > 1) { I_next = tile_I[R.first + 1]; off_next = tile_off[R.first + 1]; }
            if (PAIR && A.pair_sync) __syncthreads();   // the two waves of a pair stay within one tile of each other: the second reader hits L2
            jk_digest_tile<WITH_J, WITH_K, NT, true, true, KJLT>(A, lane, i, k, I0, J0, K0, L0, ld, bk, toff, dKL, dJK, kjl, jkl, kjk);
            t = 1;
        }
        for (; t < R.count; t++) {
            const int I0 = I_next * BLK;
            const int64_t toff = off_next;
            if (t + 1 < R.count) { I_next = tile_I[R.first + t + 1]; off_next = tile_off[R.first + t + 1]; }
            if (PAIR && A.pair_sync) __syncthreads();   // the two waves of a pair stay within one tile of each other: the second reader hits L2
            jk_digest_tile<WITH_J, WITH_K, NT, false, true, KJLT>(A, lane, i, k, I0, J0, K0, L0, ld, bk, toff, dKL, dJK, kjl, jkl, kjk);
        }
    } else {
        if (A.tri && I_next == R.J) {
            const int I0 = I_next * BLK;
            const int64_t toff = off_next;
            if (R.count > 1) { I_next = tile_I[R.first + 1]; off_next = tile_off[R.first + 1]; }
            if (PAIR && A.pair_sync) __syncthreads();   // the two waves of a pair stay within one tile of each other: the second reader hits L2
            jk_digest_tile<WITH_J, WITH_K, NT, true, false, KJLT>(A, lane, i, k, I0, J0, K0, L0, ld, bk, toff, dKL, dJK, kjl, jkl, kjk);
            t = 1;
        }
        for (; t < R.count; t++) {
            const int I0 = I_next * BLK;
            const int64_t toff = off_next;
            if (t + 1 < R.count) { I_next = tile_I[R.first + t + 1]; off_next = tile_off[R.first + t + 1]; }
            if (PAIR && A.pair_sync) __syncthreads();   // the two waves of a pair stay within one tile of each other: the second reader hits L2
            jk_digest_tile<WITH_J, WITH_K, NT, false, false, KJLT>(A, lane, i, k, I0, J0, K0, L0, ld, bk, toff, dKL, dJK, kjl, jkl, kjk);
        }
    }
    // per-run outputs
    if (WITH_J) {
        double r = reduce8(jkl, lane, 32, 16, 8); // lane holds l = i
        atomicAdd(&A.Jacc[(size_t)(K0 + k) * ld + L0 + i], r);
    }
    if (WITH_K) {
        double r = reduce8(kjk, lane, 32, 16, 8); // lane holds j = i
        atomicAdd(&A.Kacc[(size_t)(J0 + i) * ld + K0 + k], r);
        // K_JL: 64 values over 64 lanes; first over i-lanes for each l-row... do it as 8 x reduce8 then reduce8
        if (KJLT) return;   // (already added per tile)
        double s[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            double col[8];
#pragma unroll
            for (int j = 0; j < 8; j++) col[j] = kjl[j][l];
            s[l] = reduce8(col, lane, 32, 16, 8); // lane holds j = i, summed over i-lanes, for this l
        }
        double r2 = reduce8(s, lane, 4, 2, 1); // lane holds l = k, summed over k-lanes
        atomicAdd(&A.Kacc[(size_t)(J0 + i) * ld + L0 + k], r2);
    }
  }
}

// The first `n_cached` work items (the longest segments) are read with the default cache policy, everything else with
// nontemporal loads: on a tensor larger than the 256 MiB Infinity Cache the nontemporal stream does not evict them, so that
// share of the tiles is served on-die in every SCF cycle after the first instead of from HBM.
template <bool WITH_J, bool WITH_K, bool NT>
__global__ __launch_bounds__(64) void jk_tiles_kernel(JkArgs A)
{
    const MI_CONST_AS int *wave_seg = as_const(A.wave_seg);
    const int seg_end = wave_seg[blockIdx.x + 1];
    for (int seg = wave_seg[blockIdx.x]; seg < seg_end; seg++) {
        if (NT && seg < A.n_cached) jk_segment<WITH_J, WITH_K, false>(A, seg);
        else jk_segment<WITH_J, WITH_K, NT>(A, seg);
    }
}


// Two densities in ONE pass over the tiles (the spin pair of UHF / UKS).  A single wave cannot hold two sets of
// run-wide accumulators (2 x 64 K_JL + ... : the compiler spills, section 3.1 of DESIGN.md), so a workgroup of TWO waves
// digests one work item: wave 0 with density / accumulators 0, wave 1 with set 1, each with the register budget of the
// single-density kernel, on two SIMDs of one CU.  Both read the same tiles; a barrier per tile keeps them within one tile of
// each other, so the second reader of a 16-byte chunk finds it in L2 (default cache policy: no nontemporal hint here) and
// HBM is read once for both densities.
// J+K with K_JL reduced per tile (KJLT): 128 registers fewer, so two waves fit a SIMD (jk_kjlt option)
template <bool NT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void jk_tiles_kjlt_kernel(JkArgs A)
{
    const MI_CONST_AS int *wave_seg = as_const(A.wave_seg);
    const int seg_end = wave_seg[blockIdx.x + 1];
    for (int seg = wave_seg[blockIdx.x]; seg < seg_end; seg++) {
        if (NT && seg < A.n_cached) jk_segment<true, true, false, false, true>(A, seg);
        else jk_segment<true, true, NT, false, true>(A, seg);
    }
}
template <bool WITH_J, bool WITH_K, bool NT>
__global__ __launch_bounds__(128) void jk_tiles_pair_kernel(JkArgs A)
{
    const int wave = threadIdx.x >> 6;
    JkArgs B = A;
    B.D = A.D + wave * A.pair_stride;
    B.Jacc = A.Jacc + wave * A.pair_stride;
    B.Kacc = A.Kacc + wave * A.pair_stride;
    const MI_CONST_AS int *wave_seg = as_const(A.wave_seg);
    const int seg_end = wave_seg[blockIdx.x + 1];
    for (int seg = wave_seg[blockIdx.x]; seg < seg_end; seg++) {
        if (NT && seg < A.n_cached) jk_segment<WITH_J, WITH_K, false, true>(B, seg);
        else jk_segment<WITH_J, WITH_K, NT, true>(B, seg);
    }
}

// Software-pipelined variant for the K-carrying builds (one wave per SIMD, so a wave has to overlap its own
// loads with its own arithmetic): a tile is digested as two halves of 4 j-rows (16 double2 chunks per lane
// each); while half A of tile t is being contracted, half B of tile t is in flight, and while half B is being
// contracted, half A of tile t+1 is in flight.  Same register budget as holding one whole tile.
template <bool DIJ, bool DKL, bool NT>
__device__ __forceinline__ void jk_load_half_t(d2_t (&buf)[16], const d2_t *__restrict__ tile, int bi, int bk, int i, int k, int half)
{
    const bool inb = (i < bi) && (k < bk);
#pragma unroll
    for (int c = 0; c < 16; c++) buf[c] = jk_load_chunk<DIJ, DKL, NT>(tile, half * 4 + (c >> 2), c & 3, bi, bk, i, k, inb);
}
// dij / dkl: wave-uniform diagonal flags of the tile (triangular rows, tile geometry above)
template <bool NT>
__device__ __forceinline__ void jk_load_half(d2_t (&buf)[16], const d2_t *__restrict__ tile, int bi, int bk, int i, int k, bool dij, bool dkl,
                                             int half)
{
    if (dkl) {
        if (dij) jk_load_half_t<true, true, NT>(buf, tile, bi, bk, i, k, half);
        else jk_load_half_t<false, true, NT>(buf, tile, bi, bk, i, k, half);
    } else {
        if (dij) jk_load_half_t<true, false, NT>(buf, tile, bi, bk, i, k, half);
        else jk_load_half_t<false, false, NT>(buf, tile, bi, bk, i, k, half);
    }
}

template <bool WITH_J, bool NT, bool TRI>
__global__ __launch_bounds__(64) void jk_tiles_pipe_kernel(JkArgs A)
{
    const int lane = threadIdx.x;
    const int i = lane >> 3, k = lane & 7;
    const MI_CONST_AS int *wave_seg = as_const(A.wave_seg);
    const MI_CONST_AS int *tile_I = as_const(A.tile_I);
    const MI_CONST_AS int64_t *tile_off = as_const(A.tile_off);
    const MI_CONST_AS double *Du = as_const(A.D);
    const int seg_end = wave_seg[blockIdx.x + 1];
  for (int seg = wave_seg[blockIdx.x]; seg < seg_end; seg++) {
    const MI_CONST_AS RunRec *rr = as_const(A.runs) + seg;
    const RunRec R{rr->J, rr->K, rr->L, rr->first, rr->count};
    const int J0 = R.J * BLK, K0 = R.K * BLK, L0 = R.L * BLK;
    const int ld = A.ld;
    const int bk = min(BLK, A.nao - K0);
    const double *__restrict__ D = A.D;

    double dKL[8], dJK[8];
#pragma unroll
    for (int l = 0; l < 8; l++) dKL[l] = D[(size_t)(K0 + k) * ld + L0 + l];
#pragma unroll
    for (int j = 0; j < 8; j++) dJK[j] = D[(size_t)(J0 + j) * ld + K0 + k];
    double kjl[8][8], jkl[8], kjk[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        jkl[j] = 0.0; kjk[j] = 0.0;
#pragma unroll
        for (int l = 0; l < 8; l++) kjl[j][l] = 0.0;
    }

    d2_t bufA[16], bufB[16];
    const bool dkl = TRI && R.K == R.L;
    int I_cur = tile_I[R.first];
    int64_t off_cur = tile_off[R.first];
    {
        const int bi0 = min(BLK, A.nao - I_cur * BLK);
        jk_load_half<NT>(bufA, reinterpret_cast<const d2_t *>(A.tiles + off_cur), bi0, bk, i, k, TRI && I_cur == R.J, dkl, 0);
    }
    for (int t = 0; t < R.count; t++) {
        const int tid = R.first + t;
        const int I0 = I_cur * BLK;
        const int bi = min(BLK, A.nao - I0);
        const bool more = t + 1 < R.count;
        int I_nx = I_cur;
        int64_t off_nx = off_cur;
        if (more) { I_nx = tile_I[tid + 1]; off_nx = tile_off[tid + 1]; }
        double dIJ[8], dIL[8];
#pragma unroll
        for (int j = 0; j < 8; j++) dIJ[j] = D[(size_t)(I0 + i) * ld + J0 + j];
#pragma unroll
        for (int l = 0; l < 8; l++) dIL[l] = D[(size_t)(I0 + i) * ld + L0 + l];
        const double dIK = D[(size_t)(I0 + i) * ld + K0 + k];
        jk_load_half<NT>(bufB, reinterpret_cast<const d2_t *>(A.tiles + off_cur), bi, bk, i, k, TRI && I_cur == R.J, dkl, 1);

        double kik = 0.0, jij[8], kil[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { jij[j] = 0.0; kil[j] = 0.0; }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            d2_t (&buf)[16] = half ? bufB : bufA;
#pragma unroll
            for (int jj = 0; jj < 4; jj++) {
                const int j = half * 4 + jj;
                const MI_CONST_AS double *dJL = Du + (size_t)(J0 + j) * ld + L0;
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    const double x = (l & 1) ? buf[jj * 4 + (l >> 1)].y : buf[jj * 4 + (l >> 1)].x;
                    kik = fma(x, dJL[l], kik);
                    kil[l] = fma(x, dJK[j], kil[l]);
                    kjl[j][l] = fma(x, dIK, kjl[j][l]);
                    kjk[j] = fma(x, dIL[l], kjk[j]);
                    if (WITH_J) {
                        jij[j] = fma(x, dKL[l], jij[j]);
                        jkl[l] = fma(x, dIJ[j], jkl[l]);
                    }
                }
            }
            if (half == 0 && more) {
                const int bin = min(BLK, A.nao - I_nx * BLK);
                jk_load_half<NT>(bufA, reinterpret_cast<const d2_t *>(A.tiles + off_nx), bin, bk, i, k, false, dkl, 0);   // only a run's first tile can have I == J
            }
        }
        atomicAdd(&A.Kacc[(size_t)(I0 + i) * ld + K0 + k], kik);
        {
            double r = reduce8(kil, lane, 4, 2, 1);
            atomicAdd(&A.Kacc[(size_t)(I0 + i) * ld + L0 + k], r);
        }
        if (WITH_J) {
            double r = reduce8(jij, lane, 4, 2, 1);
            atomicAdd(&A.Jacc[(size_t)(I0 + i) * ld + J0 + k], r);
        }
        I_cur = I_nx; off_cur = off_nx;
    }
    if (WITH_J) {
        double r = reduce8(jkl, lane, 32, 16, 8);
        atomicAdd(&A.Jacc[(size_t)(K0 + k) * ld + L0 + i], r);
    }
    {
        double r = reduce8(kjk, lane, 32, 16, 8);
        atomicAdd(&A.Kacc[(size_t)(J0 + i) * ld + K0 + k], r);
        double s8[8];
#pragma unroll
        for (int l = 0; l < 8; l++) {
            double col[8];
#pragma unroll
            for (int j = 0; j < 8; j++) col[j] = kjl[j][l];
            s8[l] = reduce8(col, lane, 32, 16, 8);
        }
        double r2 = reduce8(s8, lane, 4, 2, 1);
        atomicAdd(&A.Kacc[(size_t)(J0 + i) * ld + L0 + k], r2);
    }
  }
}

// Dense (nao^4) copy of the resident tiles for post-SCF methods on small molecules (MP2 behind `pyscf.mp`): every stored
// element is un-weighted (tiles hold 1/2 per block coincidence) and written to its eight symmetry images.
__global__ __launch_bounds__(256) void eri_unpack_kernel(const double *tiles, const int64_t *tile_off, const TileInfo *info, int nao,
                                                         int tri, double *out, const int *__restrict__ iperm)
{
    const TileInfo T = info[blockIdx.x];
    const int bi = min(BLK, nao - T.I * BLK), bk = min(BLK, nao - T.K * BLK);
    const int bij = T.I * (T.I + 1) / 2 + T.J, bkl = T.K * (T.K + 1) / 2 + T.L;
    const double *src = tiles + tile_off[blockIdx.x];
    const int ntot = BLK * BLK * bi * bk;
    const size_t n1 = nao, n2 = n1 * nao, n3 = n2 * nao;
    for (int idx = threadIdx.x; idx < ntot; idx += blockDim.x) {
        const int ll = idx & 7, jj = (idx >> 3) & 7, pos = idx >> 6;
        const int ii = pos / bk, kk = pos - ii * bk;
        size_t i = T.I * BLK + ii, j = T.J * BLK + jj, k = T.K * BLK + kk, l = T.L * BLK + ll;
        if (j >= n1 || l >= n1) continue;
        double w;
        const int64_t e = tile_elem(tri != 0, T.I == T.J, T.K == T.L, bij == bkl, bi, bk, ii, jj, kk, ll, &w);
        if (e < 0) continue;                      // the (j,i) / (l,k) partner writes this image
        const double v = src[e] / w;
        i = iperm[i]; j = iperm[j]; k = iperm[k]; l = iperm[l];   // tile order -> the caller's AO order
        out[i * n3 + j * n2 + k * n1 + l] = v; out[j * n3 + i * n2 + k * n1 + l] = v;
        out[i * n3 + j * n2 + l * n1 + k] = v; out[j * n3 + i * n2 + l * n1 + k] = v;
        out[k * n3 + l * n2 + i * n1 + j] = v; out[l * n3 + k * n2 + i * n1 + j] = v;
        out[k * n3 + l * n2 + j * n1 + i] = v; out[l * n3 + k * n2 + j * n1 + i] = v;
    }
}

extern "C" int mi_eri_unpack(mi_ctx *c, double *d_out, void *stream)
{
    if (!c || !d_out) return fail("mi_eri_unpack: null argument");
    if (!c->eri_ready) return fail("mi_eri_unpack: call mi_eri_prepare first");
    if (c->nranks != 1) return fail("mi_eri_unpack: needs the whole (unsharded) tile store");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t n = c->nao;
    HIPCHK(hipMemsetAsync(d_out, 0, sizeof(double) * n * n * n * n, st));   // screened-out tiles stay zero
    if (c->n_tiles == 0) return 0;
    TileInfo *d_info = nullptr;
    if (upload(&d_info, c->tiles)) return -1;
    hipLaunchKernelGGL(eri_unpack_kernel, dim3((unsigned)c->n_tiles), dim3(256), 0, st, c->d_tiles, c->d_tile_off, d_info, c->nao, c->tri, d_out, c->d_iperm);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    dev_free(d_info);
    return 0;
}

// One shell quartet (ish jsh|ksh lsh) read back from the resident tiles: element -> canonical tile position (the inverse of
// put_tile), un-weighted.  NaN where the tile is not resident on this rank (sharded store), 0 where it was screened out.
__global__ void eri_read_quartet_kernel(const double *tiles, const int64_t *tile_off, const int32_t *table, int nao, int ai, int ni,
                                        int aj, int nj, int ak, int nk, int al, int nl, const uint8_t *present, int tri, double *out)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ni * nj * nk * nl) return;
    int d = idx % nl, c = (idx / nl) % nk, b = (idx / (nl * nk)) % nj, a = idx / (nl * nk * nj);
    int i = ai + a, j = aj + b, k = ak + c, l = al + d;
    if ((i >> 3) < (j >> 3)) { int t = i; i = j; j = t; }
    if ((k >> 3) < (l >> 3)) { int t = k; k = l; l = t; }
    int I = i >> 3, J = j >> 3, K = k >> 3, L = l >> 3;
    int bij = I * (I + 1) / 2 + J, bkl = K * (K + 1) / 2 + L;
    if (bij < bkl) { int t = i; i = k; k = t; t = j; j = l; l = t; t = bij; bij = bkl; bkl = t; I = i >> 3; J = j >> 3; K = k >> 3; L = l >> 3; }
    const size_t slot = (size_t)bij * (bij + 1) / 2 + bkl;
    const int32_t t = table[slot];
    if (t < 0) { out[idx] = (present && present[slot]) ? __builtin_nan("") : 0.0; return; }
    const int bi = min(BLK, nao - I * BLK), bk = min(BLK, nao - K * BLK);
    int ii = i & 7, jj = j & 7, kk = k & 7, ll = l & 7;
    if (tri && I == J && ii < jj) { int t_ = ii; ii = jj; jj = t_; }
    if (tri && K == L && kk < ll) { int t_ = kk; kk = ll; ll = t_; }
    double w;
    const int64_t e = tile_elem(tri != 0, I == J, K == L, bij == bkl, bi, bk, ii, jj, kk, ll, &w);
    out[idx] = tiles[tile_off[t] + e] / w;
}

extern "C" int mi_eri_read_quartet(mi_ctx *c, int ish, int jsh, int ksh, int lsh, double *out)
{
    if (!c || !out) return fail("mi_eri_read_quartet: null argument");
    if (!c->eri_ready) return fail("mi_eri_read_quartet: call mi_eri_prepare first");
    const int sh[4] = {ish, jsh, ksh, lsh};
    for (int s_ : sh) if (s_ < 0 || s_ >= c->nbas) return fail("mi_eri_read_quartet: shell index out of range");
    HIPCHK(hipSetDevice(c->device));
    int ao[4], n[4];
    for (int q = 0; q < 4; q++) { ao[q] = c->shells[sh[q]].ao; n[q] = 2 * c->shells[sh[q]].l + 1; }
    const int tot = n[0] * n[1] * n[2] * n[3];
    double *d_out = nullptr;
    HIPCHK(dev_malloc(&d_out, sizeof(double) * tot));
    hipLaunchKernelGGL(eri_read_quartet_kernel, dim3((tot + 255) / 256), dim3(256), 0, nullptr, c->d_tiles, c->d_tile_off, c->d_tile_table,
                       c->nao, ao[0], n[0], ao[1], n[1], ao[2], n[2], ao[3], n[3], c->d_tile_present, c->tri, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, d_out, sizeof(double) * tot, hipMemcpyDeviceToHost));
    dev_free(d_out);
    return 0;
}

// `iperm`: tile AO -> caller AO (the padded copy is in the tile order of the ERI store)
__global__ void pad_density_kernel(const double *D, double *Dp, int nao, int ld, const int *__restrict__ iperm)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ld * ld) return;
    int r = idx / ld, c = idx - r * ld;
    Dp[idx] = (r < nao && c < nao) ? D[(size_t)iperm[r] * nao + iperm[c]] : 0.0;
}

// D -> zero-padded [ld][ld] copy, and the J/K accumulators cleared in the same launch (one kernel instead of three per build)
__global__ void pad_density_clear_kernel(const double *D, double *Dp, double *Jacc, double *Kacc, int nao, int ld, const int *__restrict__ iperm)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ld * ld) return;
    int r = idx / ld, c = idx - r * ld;
    Dp[idx] = (r < nao && c < nao) ? D[(size_t)iperm[r] * nao + iperm[c]] : 0.0;
    if (Jacc) Jacc[idx] = 0.0;
    if (Kacc) Kacc[idx] = 0.0;
}

// `perm`: caller AO -> tile AO (the accumulators are in tile order, J and K leave in the caller's)
__global__ void finalize_jk_kernel(const double *Jacc, const double *Kacc, double *J, double *K, int nao, int ld, const int *__restrict__ perm)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nao * nao) return;
    int r = idx / nao, c = idx - r * nao;
    r = perm[r]; c = perm[c];
    if (J) J[idx] = 2.0 * (Jacc[(size_t)r * ld + c] + Jacc[(size_t)c * ld + r]);
    if (K) K[idx] = Kacc[(size_t)r * ld + c] + Kacc[(size_t)c * ld + r];
}

static int launch_jk(mi_ctx *c, bool wj, bool wk, hipStream_t st)
{
    JkArgs A{c->d_tiles, c->d_tile_off, c->d_tile_I, c->d_segs, c->d_wave_seg, c->n_jk_waves, c->d_Dpad, c->d_Jacc, c->d_Kacc, c->ldp, c->nao,
             c->n_jk_cached, c->tri, (size_t)c->ldp * c->ldp, 0, c->opt_jk_dpp};
    if (c->n_tiles == 0) return 0;
    dim3 g(A.nruns), b(64);
    // nontemporal loads only when the tensor cannot stay in the 256 MiB Infinity Cache between SCF cycles
    const bool nt = c->opt_jk_nt != 0 && (c->opt_jk_nt > 1 || c->tile_doubles * 8 > ((int64_t)256 << 20));
    // the half-tile pipeline only exists for full-row tiles: with triangular rows (the default) the plain kernel is the faster
    // one on cache-resident tensors too (benzene/cc-pVDZ J+K 39.8 us vs 41.5 us for full rows + pipeline, 48 us for both)
    const bool pipe = !c->tri && (c->opt_jk_pipe > 0 || (c->opt_jk_pipe < 0 && c->tile_doubles * 8 <= ((int64_t)256 << 20)));
    if (pipe && wk) {
        if (wj) { if (nt) hipLaunchKernelGGL((jk_tiles_pipe_kernel<true, true, false>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_pipe_kernel<true, false, false>), g, b, 0, st, A); }
        else hipLaunchKernelGGL((jk_tiles_pipe_kernel<false, true, false>), g, b, 0, st, A);
    }
    else if (wj && wk && c->opt_jk_kjlt) { if (nt) hipLaunchKernelGGL((jk_tiles_kjlt_kernel<true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_kjlt_kernel<false>), g, b, 0, st, A); }
    else if (wj && wk) { if (nt) hipLaunchKernelGGL((jk_tiles_kernel<true, true, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_kernel<true, true, false>), g, b, 0, st, A); }
    else if (wj) { if (nt) hipLaunchKernelGGL((jk_tiles_kernel<true, false, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_kernel<true, false, false>), g, b, 0, st, A); }
    else { if (nt) hipLaunchKernelGGL((jk_tiles_kernel<false, true, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_kernel<false, true, false>), g, b, 0, st, A); }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_build_jk(mi_ctx *c, const double *d_D, int n_dm, double *d_J, double *d_K, void *stream)
{
    if (!c || !d_D) return fail("mi_build_jk: null argument");
    if (!c->eri_ready) return fail("mi_build_jk: call mi_eri_prepare first");
    if (!d_J && !d_K) return 0;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    size_t nn = (size_t)c->nao * c->nao, pp = (size_t)c->ldp * c->ldp;
    // one pass for the pair when it pays: measured -9 % on the 103 GB ibuprofen tensor (UHF cycle 38 -> 34.5 ms), -3 % ... +60 %
    // (erratic) on the 4.9 GB benzene/cc-pVTZ tensor -- so by default only for stores beyond 16 GB (jk_pair = 1: always, 0: never)
    const bool use_pair = c->opt_jk_pair > 0 || (c->opt_jk_pair < 0 && c->tile_doubles * 8 > ((int64_t)16 << 30));
    if (n_dm == 2 && use_pair && c->n_tiles > 0) {
        for (int m = 0; m < 2; m++)
            hipLaunchKernelGGL(pad_density_clear_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_D + m * nn, c->d_Dpad + m * pp,
                               d_J ? c->d_Jacc + m * pp : nullptr, d_K ? c->d_Kacc + m * pp : nullptr, c->nao, c->ldp, c->d_iperm);
        JkArgs A{c->d_tiles, c->d_tile_off, c->d_tile_I, c->d_segs, c->d_wave_seg, c->n_jk_waves, c->d_Dpad, c->d_Jacc, c->d_Kacc, c->ldp, c->nao,
                 c->n_jk_cached, c->tri, pp, 1, c->opt_jk_dpp};
        dim3 g(A.nruns), b(128);
        // same cache policy as the single-density kernel: nontemporal stream + default-policy prefix for tensors beyond the
        // Infinity Cache (an all-default-policy stream of 4.9 GB made the launch time erratic: 1.65 ... 2.75 ms)
        const bool nt = c->opt_jk_nt != 0 && (c->opt_jk_nt > 1 || c->tile_doubles * 8 > ((int64_t)256 << 20));
        if (d_J && d_K) { if (nt) hipLaunchKernelGGL((jk_tiles_pair_kernel<true, true, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_pair_kernel<true, true, false>), g, b, 0, st, A); }
        else if (d_J) { if (nt) hipLaunchKernelGGL((jk_tiles_pair_kernel<true, false, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_pair_kernel<true, false, false>), g, b, 0, st, A); }
        else { if (nt) hipLaunchKernelGGL((jk_tiles_pair_kernel<false, true, true>), g, b, 0, st, A); else hipLaunchKernelGGL((jk_tiles_pair_kernel<false, true, false>), g, b, 0, st, A); }
        HIPCHK(hipGetLastError());
        for (int m = 0; m < 2; m++)
            hipLaunchKernelGGL(finalize_jk_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, c->d_Jacc + m * pp, c->d_Kacc + m * pp,
                               d_J ? d_J + m * nn : nullptr, d_K ? d_K + m * nn : nullptr, c->nao, c->ldp, c->d_perm);
        HIPCHK(hipGetLastError());
        return 0;
    }
    for (int m = 0; m < n_dm; m++) {
        hipLaunchKernelGGL(pad_density_clear_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_D + m * nn, c->d_Dpad,
                           d_J ? c->d_Jacc : nullptr, d_K ? c->d_Kacc : nullptr, c->nao, c->ldp, c->d_iperm);
        if (launch_jk(c, d_J != nullptr, d_K != nullptr, st)) return -1;
        hipLaunchKernelGGL(finalize_jk_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, c->d_Jacc, c->d_Kacc,
                           d_J ? d_J + m * nn : nullptr, d_K ? d_K + m * nn : nullptr, c->nao, c->ldp, c->d_perm);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

extern "C" int mi_time_jk_variant(mi_ctx *c, const double *d_D, int with_j, int with_k, int reps, double *ms, void *stream)
{
    if (!c || !d_D || !ms || reps < 1 || (!with_j && !with_k)) return fail("mi_time_jk_variant: bad argument");
    if (!c->eri_ready) return fail("mi_time_jk_kernel: call mi_eri_prepare first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    size_t pp = (size_t)c->ldp * c->ldp;
    hipLaunchKernelGGL(pad_density_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_D, c->d_Dpad, c->nao, c->ldp, c->d_iperm);
    HIPCHK(hipMemsetAsync(c->d_Jacc, 0, sizeof(double) * pp, st));
    HIPCHK(hipMemsetAsync(c->d_Kacc, 0, sizeof(double) * pp, st));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    if (launch_jk(c, with_j != 0, with_k != 0, st)) return -1; // warm
    HIPCHK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; r++)
        if (launch_jk(c, with_j != 0, with_k != 0, st)) return -1;
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    *ms = (double)t / reps;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return 0;
}

extern "C" int mi_time_jk_kernel(mi_ctx *c, const double *d_D, int reps, double *ms, void *stream)
{
    return mi_time_jk_variant(c, d_D, 1, 1, reps, ms, stream);
}

// =================================================================================================
// DIIS helpers
// =================================================================================================
__global__ void diis_errvec_kernel(const double *sdf, double *err, int n)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int r = idx / n, c = idx - r * n;
    err[idx] = sdf[(size_t)c * n + r] - sdf[idx];
}

extern "C" int mi_diis_errvec(mi_ctx *c, const double *d_SDF, double *d_err, void *stream)
{
    if (!c) return fail("null context");
    int n = c->nao;
    hipLaunchKernelGGL(diis_errvec_kernel, dim3((n * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_SDF, d_err, n);
    HIPCHK(hipGetLastError());
    return 0;
}

struct CoefPack { double c[16]; };
__global__ void diis_combine_kernel(const double *hist, CoefPack cf, int n, size_t nn, double *out)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nn) return;
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(cf.c[i], hist[(size_t)i * nn + idx], s);
    out[idx] = s;
}

extern "C" int mi_diis_combine(mi_ctx *c, const double *d_hist, const double *coef, int n, double *d_out, void *stream)
{
    if (!c || n < 1 || n > 16) return fail("mi_diis_combine: bad argument");
    CoefPack cf{};
    for (int i = 0; i < n; i++) cf.c[i] = coef[i];
    size_t nn = (size_t)c->nao * c->nao;
    hipLaunchKernelGGL(diis_combine_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_hist, cf, n, nn, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

// Gram row of the DIIS error vectors: out[i * DIIS_NS + s] = partial dot product <hist_e[i], e> over slice s of the
// vector (grid = n x DIIS_NS blocks; the caller adds the DIIS_NS partials in index order: deterministic, and 16x the
// parallelism of one block per history vector).
#define DIIS_NS 16
__global__ __launch_bounds__(256) void diis_dots_kernel(const double *hist, const double *e, size_t nn, double *out)
{
    __shared__ double sh[4];
    const double *h = hist + (size_t)blockIdx.x * nn;
    const size_t per = (nn + DIIS_NS - 1) / DIIS_NS, lo = per * blockIdx.y, hi = lo + per < nn ? lo + per : nn;
    double p[4] = {0, 0, 0, 0};
    size_t i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = fma(h[i + u * 256], e[i + u * 256], p[u]);
    }
    for (; i < hi; i += 256) p[0] = fma(h[i], e[i], p[0]);
    double s = (p[0] + p[1]) + (p[2] + p[3]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x * DIIS_NS + blockIdx.y] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int mi_diis_dots_dev_n(mi_ctx *c, const double *d_hist_e, const double *d_e, int n, int64_t len, double *d_out, void *stream)
{
    if (!c || n < 1 || n > 64 || len < 1 || !d_out) return fail("mi_diis_dots_dev_n: bad argument");
    hipLaunchKernelGGL(diis_dots_kernel, dim3(n, DIIS_NS), dim3(256), 0, (hipStream_t)stream, d_hist_e, d_e, (size_t)len, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_diis_dots_dev(mi_ctx *c, const double *d_hist_e, const double *d_e, int n, double *d_out, void *stream)
{
    if (!c) return fail("mi_diis_dots_dev: bad argument");
    return mi_diis_dots_dev_n(c, d_hist_e, d_e, n, (int64_t)c->nao * c->nao, d_out, stream);
}

// CDIIS without the host: one small workgroup adds the Gram-row partials of the newest error vector in index order, updates the
// device-resident B matrix (row and column `slot` of [space][space]) and solves Pulay's (m+1) x (m+1) system
//   [0 1^T; 1 B] [lambda; c] = [1; 0]
// by Gaussian elimination with partial pivoting (what LAPACK's dgesv behind numpy.linalg.solve does in PySCF's
// scf.diis -> lib.diis [MEM]); c[0..m) goes to d_coef for diis_combine_dev_kernel.  A singular or non-finite system (error
// vectors exactly zero) falls back to c = e_slot, i.e. no extrapolation.  Same instruction sequence on every rank of a
// sharded run, so the coefficients are bit-identical there.
#define DIIS_MAXM 16
// value of `x` on lane `l` (wave-uniform l) through v_readlane: no LDS crossbar trip, unlike __shfl with a runtime lane
__device__ __forceinline__ double lane_bcast(double x, int l)
{
    const int ls = __builtin_amdgcn_readfirstlane(l);
    const long long b = __double_as_longlong(x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), ls);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), ls);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// One wave, the augmented system [n][n + 1] in LDS, every loop rolled: 5 KB of code.  Lane r owns row r, the pivot is found with a
// wave reduction, the pivot row is read as LDS broadcasts, back substitution is column oriented.  Three implementations measured
// the same 14 us per launch (4.8 us of it the floor of any kernel here): rows in LDS with lane 0 scanning for the pivot; a fully
// unrolled register / v_readlane version (41 KB of code); this one.  What is left is a chain of ~60 dependent LDS / cross-lane
// round trips on one wave plus two dependent global round trips (partials in, coefficients out).
__global__ __launch_bounds__(64) void diis_solve_kernel(const double *__restrict__ part, int m, int slot, int space, double *B, double *__restrict__ coef)
{
    __shared__ double A[DIIS_MAXM + 1][DIIS_MAXM + 3];   // column n holds the right-hand side
    __shared__ double X[DIIS_MAXM + 1];
    const int t = threadIdx.x;
    const int n = m + 1;
    // older Gram entries of this lane's row and the partial sums of the newest row: all requested before anything waits
    double bold[DIIS_MAXM];
#pragma unroll
    for (int j = 0; j < DIIS_MAXM; j++) bold[j] = (t >= 1 && t <= m && j < m) ? B[(t - 1) * space + j] : 0.0;
    double dot = 0.0;                         // lane i < m: <e_i, e_slot>
    if (t < m) {
        double pq[DIIS_NS];
#pragma unroll
        for (int q = 0; q < DIIS_NS; q++) pq[q] = part[t * DIIS_NS + q];
#pragma unroll
        for (int q = 0; q < DIIS_NS; q++) dot += pq[q];
        B[slot * space + t] = dot;
        B[t * space + slot] = dot;
        X[t] = dot;                           // staged for the other lanes (row / column `slot` of the system)
    }
    __syncthreads();
    if (t < n) {
#pragma unroll
        for (int j = 0; j <= DIIS_MAXM; j++) {
            double v = 0.0;
            if (j < n) {
                if (t == 0) v = (j == 0) ? 0.0 : 1.0;
                else if (j == 0) v = 1.0;
                else if (t - 1 == slot) v = X[j - 1];
                else if (j - 1 == slot) v = X[t - 1];
                else v = bold[j - 1];
            }
            A[t][j] = v;
        }
        A[t][n] = (t == 0) ? 1.0 : 0.0;
    }
    __syncthreads();
    bool ok = true;
    for (int k = 0; k < n && ok; k++) {
        // pivot: largest |A[r][k]|, r = k .. n-1, the first one on ties (LAPACK's idamax)
        double val = (t >= k && t < n) ? fabs(A[t][k]) : -1.0;
        int p = t;
        for (int o = 32; o > 0; o >>= 1) {
            const double v2 = __shfl_xor(val, o);
            const int p2 = __shfl_xor(p, o);
            if (v2 > val || (v2 == val && p2 < p)) { val = v2; p = p2; }
        }
        if (!(val > 0.0) || !isfinite(val)) { ok = false; break; }        // wave-uniform
        if (p != k && t <= n) { const double x = A[k][t]; A[k][t] = A[p][t]; A[p][t] = x; }   // lane t swaps column t
        __syncthreads();
        if (t > k && t < n) {
            const double f = A[t][k] / A[k][k];
            for (int j = k; j <= n; j++) A[t][j] = fma(-f, A[k][j], A[t][j]);
        }
        __syncthreads();
    }
    if (ok) {
        for (int i = n - 1; i >= 0; i--) {
            if (t == i) X[i] = A[i][n] / A[i][i];
            __syncthreads();
            const double xi = X[i];
            if (!isfinite(xi)) ok = false;                                   // wave-uniform (every lane reads the same value)
            if (t < i) A[t][n] = fma(-A[t][i], xi, A[t][n]);
            __syncthreads();
        }
    }
    if (t >= 1 && t <= m) coef[t - 1] = ok ? X[t] : ((t - 1) == slot ? 1.0 : 0.0);
}

extern "C" int mi_diis_solve(mi_ctx *c, const double *d_part, int m, int slot, int space, double *d_B, double *d_coef, void *stream)
{
    if (!c || !d_part || !d_B || !d_coef || m < 1 || m > DIIS_MAXM || space < m || slot < 0 || slot >= m) return fail("mi_diis_solve: bad argument");
    hipLaunchKernelGGL(diis_solve_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_part, m, slot, space, d_B, d_coef);
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void diis_combine_dev_kernel(const double *hist, const double *coef, int n, size_t nn, double *out)
{
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nn) return;
    const MI_CONST_AS double *cf = as_const(coef);
    double s = 0.0;
    for (int i = 0; i < n; i++) s = fma(cf[i], hist[(size_t)i * nn + idx], s);
    out[idx] = s;
}

extern "C" int mi_diis_combine_dev_n(mi_ctx *c, const double *d_hist, const double *d_coef, int n, int64_t len, double *d_out, void *stream)
{
    if (!c || !d_hist || !d_coef || !d_out || n < 1 || n > DIIS_MAXM || len < 1) return fail("mi_diis_combine_dev_n: bad argument");
    hipLaunchKernelGGL(diis_combine_dev_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_hist, d_coef, n, (size_t)len, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_diis_combine_dev(mi_ctx *c, const double *d_hist, const double *d_coef, int n, double *d_out, void *stream)
{
    if (!c) return fail("mi_diis_combine_dev: bad argument");
    return mi_diis_combine_dev_n(c, d_hist, d_coef, n, (int64_t)c->nao * c->nao, d_out, stream);
}

extern "C" int mi_diis_dots(mi_ctx *c, const double *d_hist_e, const double *d_e, int n, double *out, void *stream)
{
    if (!c || n < 1 || n > 64) return fail("mi_diis_dots: bad argument");
    size_t nn = (size_t)c->nao * c->nao;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(diis_dots_kernel, dim3(n, DIIS_NS), dim3(256), 0, st, d_hist_e, d_e, nn, c->d_red);
    HIPCHK(hipGetLastError());
    double part[64 * DIIS_NS];
    HIPCHK(hipMemcpyAsync(part, c->d_red, sizeof(double) * n * DIIS_NS, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) {
        double s = 0.0;
        for (int q = 0; q < DIIS_NS; q++) s += part[i * DIIS_NS + q];
        out[i] = s;
    }
    return 0;
}

// =================================================================================================
// DFT: Becke weights, AO values on the grid, density, XC functionals, weighted AOs (rows a7-a9)
// =================================================================================================
#define MAXATM_LDS 1024

struct BeckeArgs {
    const double *coords; // [ng][3]
    const int32_t *atom_of;
    const double *vol;
    const double *atom_xyz; // [natm][3]
    const double *adj;      // [natm][natm] Treutler adjustment a_ij
    int natm;
    int64_t ng;
    double *w;
};

// One thread per grid point.  P_i = prod_{j != i} s(mu_ij), 3x iterated Becke polynomial, with
// mu' = mu + a_ij (1 - mu^2); w = vol * P_owner / sum_i P_i.
__global__ __launch_bounds__(256) void becke_weights_kernel(BeckeArgs A)
{
    extern __shared__ double sh[]; // atom coordinates
    for (int i = threadIdx.x; i < A.natm * 3; i += blockDim.x) sh[i] = A.atom_xyz[i];
    __syncthreads();
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.ng) return;
    const double x = A.coords[3 * g], y = A.coords[3 * g + 1], z = A.coords[3 * g + 2];
    const int own = A.atom_of[g];
    double psum = 0.0, pown = 0.0;
    for (int i = 0; i < A.natm; i++) {
        double dxi = x - sh[3 * i], dyi = y - sh[3 * i + 1], dzi = z - sh[3 * i + 2];
        double ri = sqrt(dxi * dxi + dyi * dyi + dzi * dzi);
        double p = 1.0;
        for (int j = 0; j < A.natm; j++) {
            if (j == i) continue;
            double dxj = x - sh[3 * j], dyj = y - sh[3 * j + 1], dzj = z - sh[3 * j + 2];
            double rj = sqrt(dxj * dxj + dyj * dyj + dzj * dzj);
            double ax = sh[3 * i] - sh[3 * j], ay = sh[3 * i + 1] - sh[3 * j + 1], az = sh[3 * i + 2] - sh[3 * j + 2];
            double mu = (ri - rj) * rsqrt(ax * ax + ay * ay + az * az);
            mu = mu + A.adj[i * A.natm + j] * (1.0 - mu * mu);
            mu = (3.0 - mu * mu) * mu * 0.5;
            mu = (3.0 - mu * mu) * mu * 0.5;
            mu = (3.0 - mu * mu) * mu * 0.5;
            p *= 0.5 * (1.0 - mu);
        }
        psum += p;
        if (i == own) pown = p;
    }
    A.w[g] = A.vol[g] * pown / psum;
}

extern "C" int mi_grid_becke(mi_ctx *c, const double *d_coords, const int32_t *d_atom_of, const double *d_vol, int64_t ng,
                             const double *d_adjust, double *d_weights, void *stream)
{
    if (!c || !d_coords || !d_atom_of || !d_vol || !d_adjust || !d_weights) return fail("mi_grid_becke: null argument");
    if (c->natm > MAXATM_LDS) return fail("mi_grid_becke: too many atoms");
    HIPCHK(hipSetDevice(c->device));
    std::vector<double> xyz(c->natm * 3);
    for (int i = 0; i < c->natm; i++)
        for (int d = 0; d < 3; d++) xyz[3 * i + d] = c->env[c->atm[i * ATM_SLOTS + 1] + d];
    double *d_xyz = nullptr;
    HIPCHK(dev_malloc(&d_xyz, sizeof(double) * xyz.size()));
    HIPCHK(hipMemcpy(d_xyz, xyz.data(), sizeof(double) * xyz.size(), hipMemcpyHostToDevice));
    BeckeArgs A{d_coords, d_atom_of, d_vol, d_xyz, d_adjust, c->natm, ng, d_weights};
    hipLaunchKernelGGL(becke_weights_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), sizeof(double) * 3 * c->natm,
                       (hipStream_t)stream, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    dev_free(d_xyz);
    return 0;
}

struct AoArgs {
    const int32_t *atm, *bas;
    const double *env;
    const int *shell_ao;
    const double *c2s;
    int c2s_off[LMAX + 2];
    int nbas, nao, deriv;
    const double *coords; // [ng][3]
    int64_t ng;
    double *ao;           // [1 | 4 | 10][nao][ng]: value; +gradient; +second derivatives xx,xy,xz,yy,yz,zz
};

// One thread per grid point, loop over shells (wave-uniform shell data -> scalar loads); stores are
// coalesced along the grid index.
// One shell at one grid point with compile-time angular momentum and derivative order: every loop bound is a constant, so
// the spherical accumulators (up to 10 x 7 doubles for an f shell with second derivatives) live in registers instead of
// the 528 B of scratch per thread the runtime-l version needed.
template <int L, int DERIV>
__device__ __forceinline__ void eval_ao_shell(const AoArgs &A, const int32_t *b, int ish, int64_t g, double x, double y, double z)
{
    constexpr int ns = 2 * L + 1;
    const int np = b[2];
    const double r2 = x * x + y * y + z * z;
    double rad = 0.0, drad = 0.0, d2rad = 0.0;
    for (int p = 0; p < np; p++) {
        double a = A.env[b[5] + p];
        double e = A.env[b[6] + p] * exp(-a * r2);
        rad += e;
        if (DERIV >= 1) drad -= 2.0 * a * e;
        if (DERIV >= 2) d2rad += 4.0 * a * a * e;
    }
    const double *c2s = A.c2s + A.c2s_off[L];
    double s[ns], s1[3][ns], s2[6][ns];
#pragma unroll
    for (int m = 0; m < ns; m++) {
        s[m] = 0.0;
#pragma unroll
        for (int q = 0; q < 3; q++) s1[q][m] = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) s2[q][m] = 0.0;
    }
    double px[L + 1], py[L + 1], pz[L + 1];
    px[0] = py[0] = pz[0] = 1.0;
#pragma unroll
    for (int k = 1; k <= L; k++) { px[k] = px[k - 1] * x; py[k] = py[k - 1] * y; pz[k] = pz[k - 1] * z; }
    int k = 0;
#pragma unroll
    for (int lx = L; lx >= 0; lx--)
#pragma unroll
        for (int ly = L - lx; ly >= 0; ly--, k++) {
            const int lz = L - lx - ly;
            const double v = px[lx] * py[ly] * pz[lz];
            double dx = 0.0, dy = 0.0, dz = 0.0, h2[6] = {0, 0, 0, 0, 0, 0};
            if (DERIV >= 1) {
                dx = lx ? lx * px[lx > 0 ? lx - 1 : 0] * py[ly] * pz[lz] : 0.0;
                dy = ly ? ly * px[lx] * py[ly > 0 ? ly - 1 : 0] * pz[lz] : 0.0;
                dz = lz ? lz * px[lx] * py[ly] * pz[lz > 0 ? lz - 1 : 0] : 0.0;
            }
            if (DERIV >= 2) {
                h2[0] = lx > 1 ? lx * (lx - 1) * px[lx > 1 ? lx - 2 : 0] * py[ly] * pz[lz] : 0.0;
                h2[1] = (lx && ly) ? lx * ly * px[lx > 0 ? lx - 1 : 0] * py[ly > 0 ? ly - 1 : 0] * pz[lz] : 0.0;
                h2[2] = (lx && lz) ? lx * lz * px[lx > 0 ? lx - 1 : 0] * py[ly] * pz[lz > 0 ? lz - 1 : 0] : 0.0;
                h2[3] = ly > 1 ? ly * (ly - 1) * px[lx] * py[ly > 1 ? ly - 2 : 0] * pz[lz] : 0.0;
                h2[4] = (ly && lz) ? ly * lz * px[lx] * py[ly > 0 ? ly - 1 : 0] * pz[lz > 0 ? lz - 1 : 0] : 0.0;
                h2[5] = lz > 1 ? lz * (lz - 1) * px[lx] * py[ly] * pz[lz > 1 ? lz - 2 : 0] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < ns; m++) {
                const double cc = c2s[k * ns + m];
                s[m] += cc * v;
                if (DERIV >= 1) { s1[0][m] += cc * dx; s1[1][m] += cc * dy; s1[2][m] += cc * dz; }
                if (DERIV >= 2) {
#pragma unroll
                    for (int q = 0; q < 6; q++) s2[q][m] += cc * h2[q];
                }
            }
        }
    const int ao0 = A.shell_ao[ish];
    const size_t comp = (size_t)A.nao * A.ng;
    const double xyz[3] = {x, y, z};
#pragma unroll
    for (int m = 0; m < ns; m++) {
        const size_t o = (size_t)(ao0 + m) * A.ng + g;
        A.ao[o] = rad * s[m];
        if (DERIV >= 1) {
#pragma unroll
            for (int q = 0; q < 3; q++) A.ao[(1 + q) * comp + o] = drad * xyz[q] * s[m] + rad * s1[q][m];
        }
        if (DERIV >= 2) {
            int q = 0;
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = i; j < 3; j++, q++)
                    A.ao[(4 + q) * comp + o] = d2rad * xyz[i] * xyz[j] * s[m] + (i == j ? drad * s[m] : 0.0) +
                                               drad * (xyz[i] * s1[j][m] + xyz[j] * s1[i][m]) + rad * s2[q][m];
        }
    }
}

template <int DERIV>
__global__ __launch_bounds__(256) void eval_ao_kernel(AoArgs A)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= A.ng) return;
    const double gx = A.coords[3 * g], gy = A.coords[3 * g + 1], gz = A.coords[3 * g + 2];
    for (int ish = 0; ish < A.nbas; ish++) {
        const int32_t *b = A.bas + ish * BAS_SLOTS;
        const double *R = A.env + A.atm[b[0] * ATM_SLOTS + 1];
        const double x = gx - R[0], y = gy - R[1], z = gz - R[2];
        switch (b[1]) {   // wave-uniform
        case 0: eval_ao_shell<0, DERIV>(A, b, ish, g, x, y, z); break;
        case 1: eval_ao_shell<1, DERIV>(A, b, ish, g, x, y, z); break;
        case 2: eval_ao_shell<2, DERIV>(A, b, ish, g, x, y, z); break;
        default: eval_ao_shell<3, DERIV>(A, b, ish, g, x, y, z); break;
        }
    }
}

extern "C" int mi_eval_ao(mi_ctx *c, const double *d_coords, int64_t ng, int deriv, double *d_ao, void *stream)
{
    if (c && check_orbital_lmax(c, "mi_eval_ao")) return -1;
    if (!c || !d_coords || !d_ao) return fail("mi_eval_ao: null argument");
    HIPCHK(hipSetDevice(c->device));
    AoArgs A;
    A.atm = c->d_atm; A.bas = c->d_bas; A.env = c->d_env; A.shell_ao = c->d_shell_ao; A.c2s = c->d_c2s;
    for (int i = 0; i <= LMAX + 1; i++) A.c2s_off[i] = c->c2s_off[i];
    A.nbas = c->nbas; A.nao = c->nao; A.deriv = deriv; A.coords = d_coords; A.ng = ng; A.ao = d_ao;
    const dim3 grid((unsigned)((ng + 255) / 256)), block(256);
    if (deriv <= 0) hipLaunchKernelGGL(eval_ao_kernel<0>, grid, block, 0, (hipStream_t)stream, A);
    else if (deriv == 1) hipLaunchKernelGGL(eval_ao_kernel<1>, grid, block, 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(eval_ao_kernel<2>, grid, block, 0, (hipStream_t)stream, A);
    HIPCHK(hipGetLastError());
    return 0;
}

// rho[0][g] = sum_mu ao0*C ; rho[1..3][g] = 2 sum_mu ao_k*C   with C = D @ ao0
__global__ __launch_bounds__(256) void xc_rho_kernel(const double *ao, const double *C, int nao, int64_t ng, int deriv, double *rho)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const size_t comp = (size_t)nao * ng;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int m = 0; m < nao; m++) {
        size_t o = (size_t)m * ng + g;
        double cv = C[o];
        r0 = fma(ao[o], cv, r0);
        if (deriv) {
            r1 = fma(ao[comp + o], cv, r1);
            r2 = fma(ao[2 * comp + o], cv, r2);
            r3 = fma(ao[3 * comp + o], cv, r3);
        }
    }
    rho[g] = r0;
    if (deriv) { rho[ng + g] = 2 * r1; rho[2 * ng + g] = 2 * r2; rho[3 * ng + g] = 2 * r3; }
}

extern "C" int mi_xc_rho(mi_ctx *c, const double *d_ao, const double *d_C, int64_t ng, int deriv, double *d_rho, void *stream)
{
    if (!c || !d_ao || !d_C || !d_rho) return fail("mi_xc_rho: null argument");
    hipLaunchKernelGGL(xc_rho_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_ao, d_C, c->nao, ng, deriv, d_rho);
    HIPCHK(hipGetLastError());
    return 0;
}

// The same densities from OCCUPIED-ORBITAL values on the grid (numint.eval_rho2 [MEM]): psi[(1|4)][nocc][ng] = Z^T ao with
// D = Z Z^T (rank nocc << nao), rho = sum_i psi_i^2, grad rho = 2 sum_i psi_i grad psi_i, tau = 1/2 sum_i |grad psi_i|^2.
// nao/nocc times fewer bytes and flops than the D.ao route when the density is a projector (every SCF cycle).
__global__ __launch_bounds__(256) void xc_rho_mo_kernel(const double *psi, int nocc, int64_t ng, int deriv, double *rho, double *tau)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const size_t comp = (size_t)nocc * ng;
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, tk = 0;
    for (int m = 0; m < nocc; m++) {
        size_t o = (size_t)m * ng + g;
        const double p0 = psi[o];
        r0 = fma(p0, p0, r0);
        if (deriv) {
            const double px = psi[comp + o], py = psi[2 * comp + o], pz = psi[3 * comp + o];
            r1 = fma(px, p0, r1); r2 = fma(py, p0, r2); r3 = fma(pz, p0, r3);
            tk = fma(px, px, fma(py, py, fma(pz, pz, tk)));
        }
    }
    rho[g] = r0;
    if (deriv) { rho[ng + g] = 2 * r1; rho[2 * ng + g] = 2 * r2; rho[3 * ng + g] = 2 * r3; }
    if (tau) tau[g] = 0.5 * tk;
}

// Warm start of the low-rank (Nystrom) factor of the projector, one launch instead of ~14 elementwise / reduction launches:
//   good  = every element of Zt [nocc][n] is finite and the Cholesky factorisation reported info == 0
//   scale = 1 / sqrt(max(sum_c Zt[0][c]^2, 1e-300))
//   G[r][c] = good ? 0.05 G0[r][c] + scale Zt[c][r] : G0[r][c]           (G, G0: [n][nocc])
// Every workgroup recomputes the two reductions in the same fixed order (the factor is a few hundred KB in L2) and writes its
// share of G: deterministic, no second launch.  (dft.RKS._lowrank_factor; PySCF has no counterpart -- it takes the occupied
// orbitals from the diagonalisation [MEM: numint.eval_rho2].)
__global__ __launch_bounds__(256) void nystrom_warm_kernel(const double *__restrict__ Zt, const int *__restrict__ info,
                                                           const double *__restrict__ G0, double *__restrict__ G, int nocc, int n)
{
    __shared__ double sh_s[4];
    __shared__ int sh_f[4];
    const int t = threadIdx.x;
    const size_t tot = (size_t)nocc * n;
    int finite = 1;
    double s = 0.0;
    for (size_t idx = t; idx < tot; idx += 256) {
        const double v = Zt[idx];
        finite &= (int)isfinite(v);
        if (idx < (size_t)n) s = fma(v, v, s);     // row 0
    }
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); finite &= __shfl_xor(finite, o); }
    if ((t & 63) == 0) { sh_s[t >> 6] = s; sh_f[t >> 6] = finite; }
    __syncthreads();
    const double ssum = (sh_s[0] + sh_s[1]) + (sh_s[2] + sh_s[3]);
    const bool good = (sh_f[0] & sh_f[1] & sh_f[2] & sh_f[3]) && info[0] == 0;
    const double scale = 1.0 / sqrt(fmax(ssum, 1e-300));
    for (size_t idx = (size_t)blockIdx.x * 256 + t; idx < tot; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / nocc), c = (int)(idx - (size_t)r * nocc);
        G[idx] = good ? fma(scale, Zt[(size_t)c * n + r], 0.05 * G0[idx]) : G0[idx];
    }
}

// Quadrature sums of one grid block in ONE launch: tail[q] += sum_g w[g] v_q[g] for q < nv (N_elec, E_xc; the two spin counts
// and E_xc for UKS) -- replaces nv rocBLAS dots (two launches each) and nv adds.  Deterministic: every workgroup reduces a
// contiguous chunk in a fixed tree and stores its partials; the workgroup that arrives LAST (ticket from an agent-scope atomic,
// fences around it) adds all partials in index order.  scratch: [XT_MAXWG][3] doubles + the ticket counter, owned by the context.
#define XT_MAXWG 256
struct XcTailArgs { const double *w; const double *v[3]; int nv; int64_t ng; double *tail; double *scratch; unsigned *ticket; };
__global__ __launch_bounds__(256) void xc_tail_kernel(XcTailArgs A)
{
    __shared__ double sh[3][4];
    __shared__ int last;
    const int t = threadIdx.x, nwg = gridDim.x;
    const int64_t per = (A.ng + nwg - 1) / nwg, lo = per * blockIdx.x, hi = min(A.ng, lo + per);
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t g = lo + t; g < hi; g += 256) {
        const double wg = A.w[g];
#pragma unroll
        for (int q = 0; q < 3; q++) if (q < A.nv) s[q] = fma(wg, A.v[q][g], s[q]);
    }
#pragma unroll
    for (int q = 0; q < 3; q++) {
        for (int o = 32; o > 0; o >>= 1) s[q] += __shfl_xor(s[q], o);
        if ((t & 63) == 0) sh[q][t >> 6] = s[q];
    }
    __syncthreads();
    if (t == 0) {
        for (int q = 0; q < A.nv; q++) A.scratch[blockIdx.x * 3 + q] = (sh[q][0] + sh[q][1]) + (sh[q][2] + sh[q][3]);
        __threadfence();                                             // partials visible before the ticket
        const unsigned k = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = (k == (unsigned)nwg - 1u);
        if (last) {
            __threadfence();
            for (int q = 0; q < A.nv; q++) {
                double tot = 0.0;
                for (int b = 0; b < nwg; b++) tot += __hip_atomic_load(A.scratch + b * 3 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                A.tail[q] += tot;
            }
            __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream order)
        }
    }
}

extern "C" int mi_xc_tail(mi_ctx *c, const double *d_w, const double *d_v0, const double *d_v1, const double *d_v2, int64_t ng,
                          double *d_tail, void *stream)
{
    if (!c || !d_w || !d_v0 || !d_tail || ng < 1) return fail("mi_xc_tail: bad argument");
    if (!c->d_xt_scratch) {
        HIPCHK(hipSetDevice(c->device));
        HIPCHK(dev_malloc(&c->d_xt_scratch, sizeof(double) * (XT_MAXWG * 3 + 1)));
        HIPCHK(hipMemsetAsync(c->d_xt_scratch, 0, sizeof(double) * (XT_MAXWG * 3 + 1), (hipStream_t)stream));
    }
    XcTailArgs A{};
    A.w = d_w; A.v[0] = d_v0; A.v[1] = d_v1; A.v[2] = d_v2; A.nv = d_v2 ? 3 : (d_v1 ? 2 : 1); A.ng = ng; A.tail = d_tail;
    A.scratch = c->d_xt_scratch; A.ticket = reinterpret_cast<unsigned *>(c->d_xt_scratch + XT_MAXWG * 3);
    const int nwg = (int)std::max<int64_t>(1, std::min<int64_t>(XT_MAXWG, (ng + 2047) / 2048));
    hipLaunchKernelGGL(xc_tail_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, A);
    HIPCHK(hipGetLastError());
    return 0;
}

// Cholesky factor of M = G^T X G [nocc][nocc] (lower, row-major) and the triangular solve Zt = R^-1 W^T in ONE launch (replaces
// torch.linalg.cholesky_ex + solve_triangular: potf2, reset_info, iota, triu and trsm launches, 48 us per cycle at nocc = 21).
// W: [n][nocc] row-major; Zt: [nocc][n].  Every workgroup of 64 threads factors M itself in LDS (nocc <= 64: at most 90 k flops)
// and then solves for its 64 columns, row by row: z_i = (w_i - sum_{j<i} R_ij z_j) / R_ii with the earlier z_j kept in LDS.  info[0] = 0, or k + 1 for the first non-positive / non-finite pivot (LAPACK's convention);
// in that case Zt is filled with NaN so that the caller's electron count fails its check.
#define NYS_MAXOCC 64
__global__ __launch_bounds__(64) void nystrom_factor_kernel(const double *__restrict__ M, const double *__restrict__ W, int n, int nocc,
                                                            double *Zt, int *__restrict__ info)
{
    __shared__ double L[NYS_MAXOCC][NYS_MAXOCC + 1];
    __shared__ double zb[NYS_MAXOCC][64];
    __shared__ int fail_at;
    const int t = threadIdx.x;
    for (int idx = t; idx < nocc * nocc; idx += 64) { const int r = idx / nocc, c = idx - r * nocc; L[r][c] = M[idx]; }
    if (t == 0) fail_at = 0;
    __syncthreads();
    for (int k = 0; k < nocc; k++) {
        if (t == 0) {
            const double d = L[k][k];
            if (!(d > 0.0) || !isfinite(d)) { if (fail_at == 0) fail_at = k + 1; L[k][k] = 1.0; }
            else L[k][k] = sqrt(d);
        }
        __syncthreads();
        if (t > k && t < nocc) L[t][k] /= L[k][k];
        __syncthreads();
        const int m = nocc - k - 1;                       // trailing block (i, j), k < j <= i < nocc
        for (int idx = t; idx < m * m; idx += 64) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) L[i][j] = fma(-L[i][k], L[j][k], L[i][j]);
        }
        __syncthreads();
    }
    const bool failed = fail_at != 0;
    if (blockIdx.x == 0 && t == 0) info[0] = fail_at;
    const int c = blockIdx.x * 64 + t;
    if (c >= n) return;
    const double *w = W + (size_t)c * nocc;
    for (int i = 0; i < nocc; i++) {
        double z = w[i];
        for (int j = 0; j < i; j++) z = fma(-L[i][j], zb[j][t], z);    // this thread's earlier z_j: LDS column t, conflict-free
        z /= L[i][i];
        zb[i][t] = z;
        Zt[(size_t)i * n + c] = failed ? __builtin_nan("") : z;
    }
}

extern "C" int mi_nystrom_factor(mi_ctx *c, const double *d_M, const double *d_W, int n, int nocc, double *d_Zt, int *d_info, void *stream)
{
    if (!c || !d_M || !d_W || !d_Zt || !d_info || n < 1 || nocc < 1) return fail("mi_nystrom_factor: bad argument");
    if (nocc > NYS_MAXOCC) return fail("mi_nystrom_factor: at most %d columns", NYS_MAXOCC);
    hipLaunchKernelGGL(nystrom_factor_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_M, d_W, n, nocc, d_Zt, d_info);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_nystrom_warm(mi_ctx *c, const double *d_Zt, const int *d_info, const double *d_G0, double *d_G, int nocc, int n,
                               void *stream)
{
    if (!c || !d_Zt || !d_info || !d_G0 || !d_G || nocc < 1 || n < 1) return fail("mi_nystrom_warm: bad argument");
    const int nb = (int)std::min<size_t>(64, ((size_t)nocc * n + 255) / 256);
    hipLaunchKernelGGL(nystrom_warm_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d_Zt, d_info, d_G0, d_G, nocc, n);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_xc_rho_mo(mi_ctx *c, const double *d_psi, int nocc, int64_t ng, int deriv, double *d_rho, double *d_tau, void *stream)
{
    if (!c || !d_psi || !d_rho || nocc < 1 || (d_tau && !deriv)) return fail("mi_xc_rho_mo: bad argument");
    hipLaunchKernelGGL(xc_rho_mo_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_psi, nocc, ng, deriv, d_rho, d_tau);
    HIPCHK(hipGetLastError());
    return 0;
}

// The two steps above in ONE pass over the AO values, without psi ever leaving the registers: thread = grid point, the
// occupied orbitals in chunks of IC (4 IC accumulators: psi and its gradient), Z row by row through scalar loads
// (Zp[nao][ldz], orbital index fastest, zero-padded to a multiple of IC).  The AO values (the only large operand: 4 nao ng
// doubles) are streamed once per chunk, coalesced along the grid; rocBLAS needs 0.53 ms for the [21 x 264] x [264 x 123158]
// x 4 products alone (128-wide tiles on a 21-row output) where this kernel is bound by reading 1 GB of AO values.
template <int IC, bool DERIV>
__global__ __launch_bounds__(256) void xc_rho_lowrank_kernel(const double *__restrict__ ao, const double *__restrict__ Zp, int nao, int ldz,
                                                             int64_t ng, double *__restrict__ rho, double *__restrict__ tau)
{
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t g = gi < ng ? gi : ng - 1;
    const size_t comp = (size_t)nao * ng;
    const MI_CONST_AS double *Zc = as_const(Zp);
    double r0 = 0, r1 = 0, r2 = 0, r3 = 0, tk = 0;
    for (int c0 = 0; c0 < ldz; c0 += IC) {
        double p0[IC], px[DERIV ? IC : 1], py[DERIV ? IC : 1], pz[DERIV ? IC : 1];
#pragma unroll
        for (int i = 0; i < IC; i++) { p0[i] = 0.0; if (DERIV) { px[i] = 0.0; py[i] = 0.0; pz[i] = 0.0; } }
#pragma unroll 2
        for (int m = 0; m < nao; m++) {
            const size_t o = (size_t)m * ng + g;
            const double a0 = ao[o];
            double ax = 0, ay = 0, az = 0;
            if (DERIV) { ax = ao[comp + o]; ay = ao[2 * comp + o]; az = ao[3 * comp + o]; }
            const MI_CONST_AS double *zr = Zc + (size_t)m * ldz + c0;
#pragma unroll
            for (int i = 0; i < IC; i++) {
                const double z = zr[i];
                p0[i] = fma(z, a0, p0[i]);
                if (DERIV) { px[i] = fma(z, ax, px[i]); py[i] = fma(z, ay, py[i]); pz[i] = fma(z, az, pz[i]); }
            }
        }
#pragma unroll
        for (int i = 0; i < IC; i++) {
            r0 = fma(p0[i], p0[i], r0);
            if (DERIV) {
                r1 = fma(px[i], p0[i], r1); r2 = fma(py[i], p0[i], r2); r3 = fma(pz[i], p0[i], r3);
                tk = fma(px[i], px[i], fma(py[i], py[i], fma(pz[i], pz[i], tk)));
            }
        }
    }
    if (gi >= ng) return;
    rho[g] = r0;
    if (DERIV) { rho[ng + g] = 2 * r1; rho[2 * ng + g] = 2 * r2; rho[3 * ng + g] = 2 * r3; if (tau) tau[g] = 0.5 * tk; }
}

// n_occ > 24 (ibuprofen: 56 -> three chunks) makes the kernel above walk the AO block once per chunk.  Tried and rejected (round 2,
// late): one wave per chunk over the same 64 grid points, partial densities added through LDS -- the AO block leaves HBM once, but
// the ibuprofen RKS cycle went from 27.3-29.5 to 32.9-34.0 ms (64-point workgroups, three waves contending for the same lines).
extern "C" int mi_xc_rho_lowrank(mi_ctx *c, const double *d_ao, const double *d_Zp, int ldz, int64_t ng, int deriv, double *d_rho,
                                 double *d_tau, void *stream)
{
    if (!c || !d_ao || !d_Zp || !d_rho || ldz < 1 || ng < 1 || (d_tau && !deriv)) return fail("mi_xc_rho_lowrank: bad argument");
    const dim3 g((unsigned)((ng + 255) / 256)), b(256);
    hipStream_t st = (hipStream_t)stream;
    if (deriv) {
        if (ldz % 24) return fail("mi_xc_rho_lowrank: ldz must be a multiple of 24 (GGA / meta-GGA chunk)");
        hipLaunchKernelGGL((xc_rho_lowrank_kernel<24, true>), g, b, 0, st, d_ao, d_Zp, c->nao, ldz, ng, d_rho, d_tau);
    } else {
        if (ldz % 32) return fail("mi_xc_rho_lowrank: ldz must be a multiple of 32 (LDA chunk)");
        hipLaunchKernelGGL((xc_rho_lowrank_kernel<32, false>), g, b, 0, st, d_ao, d_Zp, c->nao, ldz, ng, d_rho, d_tau);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- forward-mode dual numbers: value + N partial derivatives (N = 2: d/drho, d/dsigma of the closed-shell
// functionals; N = 5: d/d(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb) of the spin-polarised ones)
template <int N>
struct DN {
    double v, d[N];
    __device__ DN() {}
    __device__ DN(double a) : v(a) {
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = 0.0;
    }
    __device__ static DN var(double a, int which) { DN x(a); x.d[which] = 1.0; return x; }
};
template <int N> __device__ inline DN<N> operator+(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ inline DN<N> operator-(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ inline DN<N> operator-(DN<N> a) { DN<N> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ inline DN<N> operator*(DN<N> a, DN<N> b) { DN<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ inline DN<N> operator/(DN<N> a, DN<N> b) { DN<N> r; double iv = 1.0 / b.v, q = a.v * iv; r.v = q;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = (a.d[i] - q * b.d[i]) * iv; return r; }
template <int N> __device__ inline DN<N> chain(DN<N> a, double f, double df) { DN<N> r; r.v = f;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = df * a.d[i]; return r; }
template <int N> __device__ inline DN<N> dexp(DN<N> a) { double e = exp(a.v); return chain(a, e, e); }
template <int N> __device__ inline DN<N> dlog(DN<N> a) { return chain(a, log(a.v), 1.0 / a.v); }
template <int N> __device__ inline DN<N> dsqrt(DN<N> a) { double s = sqrt(a.v); return chain(a, s, 0.5 / s); }
template <int N> __device__ inline DN<N> dpow(DN<N> a, double p) { double f = pow(a.v, p); return chain(a, f, p * f / a.v); }
template <int N> __device__ inline DN<N> datan(DN<N> a) { return chain(a, atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
template <int N> __device__ inline DN<N> dasinh(DN<N> a) { return chain(a, asinh(a.v), rsqrt(1.0 + a.v * a.v)); }
typedef DN<2> D2;
typedef DN<5> D5;

enum { XC_SLATER = 1, XC_B88 = 2, XC_VWN_RPA = 3, XC_VWN5 = 4, XC_LYP = 5, XC_PBE_X = 6, XC_PBE_C = 7,
       XC_TPSS_X = 8, XC_TPSS_C = 9, XC_M062X_X = 10, XC_M062X_C = 11 };   // 8..11: meta-GGA (need tau)

// ---- closed-shell energy densities per volume e(rho, sigma); T is a dual-number type
template <class T> __device__ inline T f_slater(T rho) { return T(-0.7385587663820224) * dpow(rho, 4.0 / 3.0); } // -(3/4)(3/pi)^(1/3)

template <class T> __device__ inline T f_b88(T rho, T sig)
{
    const double beta = 0.0042;
    T rs = rho * T(0.5);                 // one spin channel
    T r43 = dpow(rs, 4.0 / 3.0);
    T x = dsqrt(sig * T(0.25) + T(1e-300)) / r43;
    T corr = T(-beta) * r43 * x * x / (T(1.0) + T(6.0 * beta) * x * dasinh(x));
    return f_slater(rho) + T(2.0) * corr;
}

// VWN fit: correlation energy per electron as a function of x = sqrt(rs)
template <class T> __device__ inline T vwn_eps(T x, double A, double x0, double b, double c)
{
    T X = x * x + T(b) * x + T(c);
    double X0 = x0 * x0 + b * x0 + c, Q = sqrt(4 * c - b * b);
    T at = datan(T(Q) / (T(2.0) * x + T(b)));
    T xm = x - T(x0);
    return T(A) * (dlog(x * x / X) + T(2 * b / Q) * at - T(b * x0 / X0) * (dlog(xm * xm / X) + T(2 * (b + 2 * x0) / Q) * at));
}

template <class T> __device__ inline T f_vwn(T rho, double A, double x0, double b, double c)
{
    T rs = dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0);
    return rho * vwn_eps(dsqrt(rs), A, x0, b, c);
}

template <class T> __device__ inline T f_lyp(T rho, T sig)
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double CF = 2.871234000188191; // (3/10)(3 pi^2)^(2/3)
    T t = dpow(rho, -1.0 / 3.0);
    T Dn = T(1.0) + T(d) * t;
    T om = dexp(T(-c) * t) / Dn * dpow(rho, -11.0 / 3.0);
    T dl = T(c) * t + T(d) * t / Dn;
    T br = T(CF) * dpow(rho, 14.0 / 3.0) - rho * rho * sig * (T(1.0 / 24.0) + T(7.0 / 72.0) * dl);
    return T(-a) * rho / Dn - T(a * b) * om * br;
}

template <class T> __device__ inline T f_pbe_x(T rho, T sig)
{
    const double kappa = 0.804, mu = 0.06672455060314922 * M_PI * M_PI / 3.0;
    T kf = dpow(T(3.0 * M_PI * M_PI) * rho, 1.0 / 3.0);
    T s2 = sig / (T(4.0) * kf * kf * rho * rho);
    T F = T(1.0 + kappa) - T(kappa) / (T(1.0) + T(mu / kappa) * s2);
    return f_slater(rho) * F;
}

// PW92 form G(rs; A, a1, b1..b4) = -2A(1 + a1 rs) ln(1 + 1/(2A(b1 x + b2 x^2 + b3 x^3 + b4 x^4))), x = sqrt(rs)
template <class T> __device__ inline T pw92_g(T rs, double A, double a1, double b1, double b2, double b3, double b4)
{
    T x = dsqrt(rs);
    return T(-2 * A) * (T(1.0) + T(a1) * rs) * dlog(T(1.0) + T(1.0) / (T(2 * A) * (T(b1) * x + T(b2) * rs + T(b3) * rs * x + T(b4) * rs * rs)));
}

template <class T> __device__ inline T f_pbe_c(T rho, T sig)
{
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;
    T rs = dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0);
    T ec = pw92_g(rs, 0.031090690869654895, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    T kf = dpow(T(3.0 * M_PI * M_PI) * rho, 1.0 / 3.0);
    T ks2 = T(4.0 / M_PI) * kf;
    T t2 = sig / (T(4.0) * ks2 * rho * rho);
    T Aa = T(beta / gamma) / (dexp(-ec / T(gamma)) - T(1.0));
    T At2 = Aa * t2;
    T H = T(gamma) * dlog(T(1.0) + T(beta / gamma) * t2 * (T(1.0) + At2) / (T(1.0) + At2 + At2 * At2));
    return rho * (ec + H);
}

// ---- spin-polarised forms e(rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb)
// exchange: spin scaling  E_x[ra, rb] = (E_x[2 ra] + E_x[2 rb]) / 2  with sigma -> 4 sigma_ss
template <class T, class F> __device__ inline T spin_scaled_exchange(T ra, T rb, T saa, T sbb, F fx)
{
    T e(0.0);
    if (ra.v > 1e-12) e = e + T(0.5) * fx(T(2.0) * ra, T(4.0) * saa);
    if (rb.v > 1e-12) e = e + T(0.5) * fx(T(2.0) * rb, T(4.0) * sbb);
    return e;
}
// zeta = (ra - rb)/rho clipped away from +-1 (the derivatives of (1 +- zeta)^(4/3) etc. stay finite)
template <class T> __device__ inline T spin_zeta(T ra, T rb)
{
    T z = (ra - rb) / (ra + rb);
    const double lim = 1.0 - 1e-10;
    if (z.v > lim) z = T(lim);
    if (z.v < -lim) z = T(-lim);
    return z;
}
template <class T> __device__ inline T spin_fzeta(T z) // ((1+z)^(4/3) + (1-z)^(4/3) - 2) / (2^(4/3) - 2)
{
    return (dpow(T(1.0) + z, 4.0 / 3.0) + dpow(T(1.0) - z, 4.0 / 3.0) - T(2.0)) * T(1.0 / (2.5198420997897464 - 2.0));
}
// VWN-RPA (libxc LDA_C_VWN_RPA, the correlation of B3LYP): eps = eps_P + (eps_F - eps_P) f(zeta), RPA fits
template <class T> __device__ inline T f_vwn_rpa_spin(T ra, T rb)
{
    T rho = ra + rb;
    T x = dsqrt(dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0));
    T eP = vwn_eps(x, 0.0310907, -0.409286, 13.0720, 42.7198);
    T eF = vwn_eps(x, 0.01554535, -0.743294, 20.1231, 101.578);
    return rho * (eP + (eF - eP) * spin_fzeta(spin_zeta(ra, rb)));
}
// VWN5: eps = eps_P + alpha_c f(z)/f''(0) (1 - z^4) + (eps_F - eps_P) f(z) z^4
template <class T> __device__ inline T f_vwn5_spin(T ra, T rb)
{
    T rho = ra + rb;
    T x = dsqrt(dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0));
    T eP = vwn_eps(x, 0.0310907, -0.10498, 3.72744, 12.9352);
    T eF = vwn_eps(x, 0.01554535, -0.32500, 7.06042, 18.0578);
    T ac = vwn_eps(x, -1.0 / (6.0 * M_PI * M_PI), -0.0047584, 1.13107, 13.0045);
    T z = spin_zeta(ra, rb), fz = spin_fzeta(z), z4 = z * z * z * z;
    return rho * (eP + ac * fz * T(1.0 / 1.7099209341613657) * (T(1.0) - z4) + (eF - eP) * fz * z4);
}
// LYP, open-shell form of Miehlich, Savin, Stoll, Preuss, CPL 157, 200 (1989)
template <class T> __device__ inline T f_lyp_spin(T ra, T rb, T saa, T sab, T sbb)
{
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double CF = 2.871234000188191;
    T rho = ra + rb;
    T sig = saa + T(2.0) * sab + sbb;
    T t = dpow(rho, -1.0 / 3.0);
    T Dn = T(1.0) + T(d) * t;
    T om = dexp(T(-c) * t) / Dn * dpow(rho, -11.0 / 3.0);
    T dl = T(c) * t + T(d) * t / Dn;
    T rab = ra * rb;
    T pa(0.0), pb(0.0);
    if (ra.v > 1e-14) pa = dpow(ra, 8.0 / 3.0);
    if (rb.v > 1e-14) pb = dpow(rb, 8.0 / 3.0);
    T t1 = T(12.699208415745595 * CF) * (pa + pb) /* 2^(11/3) C_F */ + (T(47.0 / 18.0) - T(7.0 / 18.0) * dl) * sig -
           (T(2.5) - dl * T(1.0 / 18.0)) * (saa + sbb) - (dl - T(11.0)) * T(1.0 / 9.0) * (ra / rho * saa + rb / rho * sbb);
    T br = rab * t1 - T(2.0 / 3.0) * rho * rho * sig + (T(2.0 / 3.0) * rho * rho - ra * ra) * sbb + (T(2.0 / 3.0) * rho * rho - rb * rb) * saa;
    return T(-4.0 * a) * rab / (Dn * rho) - T(a * b) * om * br;
}
// PBE correlation with the PW92 spin interpolation and phi(zeta)
template <class T> __device__ inline T f_pbe_c_spin(T ra, T rb, T saa, T sab, T sbb)
{
    const double beta = 0.06672455060314922, gamma = 0.031090690869654895;
    T rho = ra + rb;
    T sig = saa + T(2.0) * sab + sbb;
    T rs = dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0);
    T e0 = pw92_g(rs, 0.031090690869654895, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    T e1 = pw92_g(rs, 0.015545345434827448, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    T mac = pw92_g(rs, 0.016886863940389627, 0.11125, 10.357, 3.6231, 0.88026, 0.49671); // = -alpha_c
    T z = spin_zeta(ra, rb), fz = spin_fzeta(z), z4 = z * z * z * z;
    T ec = e0 - mac * fz * T(1.0 / 1.7099209341613657) * (T(1.0) - z4) + (e1 - e0) * fz * z4;
    T phi = T(0.5) * (dpow(T(1.0) + z, 2.0 / 3.0) + dpow(T(1.0) - z, 2.0 / 3.0));
    T phi3 = phi * phi * phi;
    T kf = dpow(T(3.0 * M_PI * M_PI) * rho, 1.0 / 3.0);
    T ks2 = T(4.0 / M_PI) * kf;
    T t2 = sig / (T(4.0) * phi * phi * ks2 * rho * rho);
    T Aa = T(beta / gamma) / (dexp(-ec / (T(gamma) * phi3)) - T(1.0));
    T At2 = Aa * t2;
    T H = T(gamma) * phi3 * dlog(T(1.0) + T(beta / gamma) * t2 * (T(1.0) + At2) / (T(1.0) + At2 + At2 * At2));
    return rho * (ec + H);
}

// =================================================================================================
// meta-GGA functionals (`--method M06-2X` at templates/calculate_energy.py:263, the default of
// templates/calculate_bde.py:105,502).  Spin-resolved energy densities per volume e(rho_a, rho_b, sigma_aa, sigma_ab,
// sigma_bb, tau_a, tau_b), tau_s = 1/2 sum_i |grad phi_i,s|^2, coded once on the dual-number type; the closed-shell kernel
// calls them with rho_s = rho/2, sigma_ss' = sigma/4, tau_s = tau/2.
//   TPSS      Tao, Perdew, Staroverov, Scuseria, PRL 91, 146401 (2003): exchange eq. 10, correlation = revPKZB (eqs. 11-14)
//   M06-2X    Zhao, Truhlar, Theor. Chem. Acc. 120, 215 (2008): PBE-exchange x kinetic-energy-density series (the VS98-type
//             exchange term is zero for M06-2X), M05-type + VS98-type correlation; 54 % exact exchange.
// PARAMETER TABLES ENTERED FROM MEMORY (no libxc here): "unverified-memory".  What IS checked (tests): the uniform-gas limit
// (a_0 + X = 1, c_0 + d_0 = 1 for both correlation channels), vanishing correlation and the exact TPSS exchange energy for
// one-electron densities, derivatives against the complex-step oracle.
// =================================================================================================
template <class T> __device__ inline T dmaxv(T a, T b) { return a.v >= b.v ? a : b; }

// PW92 correlation energy per particle of the spin-polarised uniform gas
template <class T> __device__ inline T pw92_eps_spin(T ra, T rb)
{
    T rho = ra + rb;
    T rs = dpow(T(0.75 / M_PI) / rho, 1.0 / 3.0);
    T e0 = pw92_g(rs, 0.031090690869654895, 0.21370, 7.5957, 3.5876, 1.6382, 0.49294);
    T e1 = pw92_g(rs, 0.015545345434827448, 0.20548, 14.1189, 6.1977, 3.3662, 0.62517);
    T mac = pw92_g(rs, 0.016886863940389627, 0.11125, 10.357, 3.6231, 0.88026, 0.49671);
    T z = spin_zeta(ra, rb), fz = spin_fzeta(z), z4 = z * z * z * z;
    return e0 - mac * fz * T(1.0 / 1.7099209341613657) * (T(1.0) - z4) + (e1 - e0) * fz * z4;
}

// ---- TPSS exchange of a spin-UNpolarised density (rho, sigma = |grad rho|^2, tau); spin scaling handles the rest
template <class T> __device__ inline T f_tpss_x_unpol(T rho, T sig, T tau)
{
    const double kappa = 0.804, bb = 0.40, cc = 1.59096, ee = 1.537, mu = 0.21951, se = 1.2397580409105368 /* sqrt(e) */;
    T p = sig / (T(4.0 * 9.570780000627305 /* (3 pi^2)^(2/3) */) * dpow(rho, 8.0 / 3.0));
    T tw = sig / (T(8.0) * rho);
    T z = tw / tau;
    if (z.v > 1.0) z = T(1.0);                                   // tau_W <= tau for N-representable input
    T tunif = T(0.3 * 9.570780000627305) * dpow(rho, 5.0 / 3.0);
    T alpha = (tau - tw) / tunif;
    if (alpha.v < 0.0) alpha = T(0.0);
    T qb = T(0.45) * (alpha - T(1.0)) / dsqrt(T(1.0) + T(bb) * alpha * (alpha - T(1.0))) + T(2.0 / 3.0) * p;
    T z2 = z * z;
    T t1 = (T(10.0 / 81.0) + T(cc) * z2 / ((T(1.0) + z2) * (T(1.0) + z2))) * p;
    T t2 = T(146.0 / 2025.0) * qb * qb;
    T t3 = T(-73.0 / 405.0) * qb * dsqrt(T(0.5) * (T(0.6) * z) * (T(0.6) * z) + T(0.5) * p * p + T(1e-300));
    T t4 = T((10.0 / 81.0) * (10.0 / 81.0) / kappa) * p * p;
    T t5 = T(2.0 * se * (10.0 / 81.0)) * (T(0.6) * z) * (T(0.6) * z);
    T t6 = T(ee * mu) * p * p * p;
    T den = T(1.0) + T(se) * p;
    T x = (t1 + t2 + t3 + t4 + t5 + t6) / (den * den);
    T Fx = T(1.0 + kappa) - T(kappa) / (T(1.0) + x / T(kappa));
    return f_slater(rho) * Fx;
}
template <class T> __device__ inline T f_tpss_x_spin(T ra, T rb, T saa, T sbb, T ta, T tb)
{
    T e(0.0);
    if (ra.v > 1e-12 && ta.v > 1e-14) e = e + T(0.5) * f_tpss_x_unpol(T(2.0) * ra, T(4.0) * saa, T(2.0) * ta);
    if (rb.v > 1e-12 && tb.v > 1e-14) e = e + T(0.5) * f_tpss_x_unpol(T(2.0) * rb, T(4.0) * sbb, T(2.0) * tb);
    return e;
}
// ---- TPSS correlation (revPKZB)
template <class T> __device__ inline T f_tpss_c_spin(T ra, T rb, T saa, T sab, T sbb, T ta, T tb)
{
    const double d = 2.8;
    T rho = ra + rb, sig = saa + T(2.0) * sab + sbb, tau = ta + tb;
    if (tau.v < 1e-14) return T(0.0);
    T tw = sig / (T(8.0) * rho);
    T z = tw / tau;
    if (z.v > 1.0) z = T(1.0);
    T zeta = spin_zeta(ra, rb);
    // xi = |grad zeta| / (2 (3 pi^2 rho)^(1/3)),  |grad zeta|^2 = 4 (rb^2 saa - 2 ra rb sab + ra^2 sbb) / rho^4
    T gz2 = T(4.0) * (rb * rb * saa - T(2.0) * ra * rb * sab + ra * ra * sbb) / (rho * rho * rho * rho);
    if (gz2.v < 0.0) gz2 = T(0.0);
    T xi2 = gz2 / (T(4.0) * dpow(T(3.0 * M_PI * M_PI) * rho, 2.0 / 3.0));
    T z2_ = zeta * zeta;
    T C0 = T(0.53) + T(0.87) * z2_ + T(0.50) * z2_ * z2_ + T(2.26) * z2_ * z2_ * z2_;
    T den = T(1.0) + xi2 * T(0.5) * (dpow(T(1.0) + zeta, -4.0 / 3.0) + dpow(T(1.0) - zeta, -4.0 / 3.0));
    T den2 = den * den;
    T C = C0 / (den2 * den2);
    T epbe = f_pbe_c_spin(ra, rb, saa, sab, sbb) / rho;
    T ea = epbe, eb = epbe;
    if (ra.v > 1e-12) ea = dmaxv(f_pbe_c_spin(ra, T(0.0), saa, T(0.0), T(0.0)) / ra, epbe);
    if (rb.v > 1e-12) eb = dmaxv(f_pbe_c_spin(rb, T(0.0), sbb, T(0.0), T(0.0)) / rb, epbe);
    T zz = z * z;
    T erev = epbe * (T(1.0) + C * zz) - (T(1.0) + C) * zz * (ra / rho * ea + rb / rho * eb);
    return rho * erev * (T(1.0) + T(d) * erev * zz * z);
}

// ---- M06-2X.  Parameter tables: unverified-memory (see the header of this block).
__device__ __constant__ double M062X_A[12] = {4.600000e-01, -2.206052e-01, -9.431788e-02, 2.164494e+00, -2.556466e+00, -1.422133e+01,
                                              1.555044e+01, 3.598078e+01, -2.722754e+01, -3.924093e+01, 1.522808e+01, 1.522227e+01};
__device__ __constant__ double M062X_CSS[5] = {3.097855e-01, -5.528642e+00, 1.347420e+01, -3.213623e+01, 2.846742e+01};
__device__ __constant__ double M062X_CAB[5] = {8.833596e-01, 3.357972e+01, -7.043548e+01, 4.978271e+01, -1.852891e+01};
__device__ __constant__ double M062X_DSS[6] = {6.902145e-01, 9.847204e-02, 2.214797e-01, -1.968264e-03, -6.775479e-03, 0.0};
__device__ __constant__ double M062X_DAB[6] = {1.166404e-01, -9.120847e-02, -6.726189e-02, 6.720580e-05, 8.448011e-04, 0.0};
#define M06_CF 9.115599744691194 /* (3/5)(6 pi^2)^(2/3) */

template <class T> __device__ inline T f_m06_x_channel(T r, T s, T tau, const double *a) // one spin channel: r = rho_s
{
    T tl = T(0.3 * 15.192666241151989 /* (6 pi^2)^(2/3) */) * dpow(r, 5.0 / 3.0);
    T t = tl / tau;
    T w = (t - T(1.0)) / (t + T(1.0));
    T fw(a[11]);
#pragma unroll
    for (int i = 10; i >= 0; i--) fw = fw * w + T(a[i]);
    return T(0.5) * f_pbe_x(T(2.0) * r, T(4.0) * s) * fw;
}
template <class T> __device__ inline T f_m062x_x_spin(T ra, T rb, T saa, T sbb, T ta, T tb)
{
    T e(0.0);
    if (ra.v > 1e-12 && ta.v > 1e-14) e = e + f_m06_x_channel(ra, saa, ta, M062X_A);
    if (rb.v > 1e-12 && tb.v > 1e-14) e = e + f_m06_x_channel(rb, sbb, tb, M062X_A);
    return e;
}
template <class T> __device__ inline T m06_h(T x2, T z, const double *dc, double alpha)
{
    T g = T(1.0) + T(alpha) * (x2 + z);
    return T(dc[0]) / g + (T(dc[1]) * x2 + T(dc[2]) * z) / (g * g) + (T(dc[3]) * x2 * x2 + T(dc[4]) * x2 * z + T(dc[5]) * z * z) / (g * g * g);
}
template <class T> __device__ inline T m06_g(T x2, const double *cc, double gamma)
{
    T u = T(gamma) * x2 / (T(1.0) + T(gamma) * x2);
    T g(cc[4]);
#pragma unroll
    for (int i = 3; i >= 0; i--) g = g * u + T(cc[i]);
    return g;
}
template <class T> __device__ inline T f_m062x_c_spin(T ra, T rb, T saa, T sbb, T ta, T tb)
{
    const bool ha = ra.v > 1e-12 && ta.v > 1e-14, hb = rb.v > 1e-12 && tb.v > 1e-14;
    T e(0.0), ess_a(0.0), ess_b(0.0), x2a(0.0), x2b(0.0), za(0.0), zb(0.0);
    if (ha) {
        x2a = saa / dpow(ra, 8.0 / 3.0);
        za = T(2.0) * ta / dpow(ra, 5.0 / 3.0) - T(M06_CF);
        ess_a = ra * pw92_eps_spin(ra, T(0.0));
        T Dsic = T(1.0) - x2a / (T(4.0) * (za + T(M06_CF)));
        if (Dsic.v < 0.0) Dsic = T(0.0);
        e = e + ess_a * (m06_g(x2a, M062X_CSS, 0.06) + m06_h(x2a, za, M062X_DSS, 0.00515088)) * Dsic;
    }
    if (hb) {
        x2b = sbb / dpow(rb, 8.0 / 3.0);
        zb = T(2.0) * tb / dpow(rb, 5.0 / 3.0) - T(M06_CF);
        ess_b = rb * pw92_eps_spin(rb, T(0.0));
        T Dsic = T(1.0) - x2b / (T(4.0) * (zb + T(M06_CF)));
        if (Dsic.v < 0.0) Dsic = T(0.0);
        e = e + ess_b * (m06_g(x2b, M062X_CSS, 0.06) + m06_h(x2b, zb, M062X_DSS, 0.00515088)) * Dsic;
    }
    if (ha && hb) {
        T eab = (ra + rb) * pw92_eps_spin(ra, rb) - ess_a - ess_b;
        e = e + eab * (m06_g(x2a + x2b, M062X_CAB, 0.0031) + m06_h(x2a + x2b, za + zb, M062X_DAB, 0.00304966));
    }
    return e;
}

struct XcSpec { int n; int kind[8]; double coef[8]; };

// exc[g] = e(rho,sigma) per volume; wv[0] = 0.5 w de/drho ; wv[1..3] = 2 w de/dsigma * grad rho
__global__ __launch_bounds__(256) void xc_eval_kernel(XcSpec X, const double *rho, const double *w, int64_t ng, int gga,
                                                      double *exc, double *wv, double *vrho_out, double *vsig_out)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    double r = rho[g];
    double gx = gga ? rho[ng + g] : 0.0, gy = gga ? rho[2 * ng + g] : 0.0, gz = gga ? rho[3 * ng + g] : 0.0;
    double e = 0.0, vr = 0.0, vs = 0.0;
    if (r > 1e-10) {
        D2 R = D2::var(r, 0), S = D2::var(gx * gx + gy * gy + gz * gz, 1);
        D2 acc(0.0);
        for (int t = 0; t < X.n; t++) {
            D2 f;
            switch (X.kind[t]) {
            case XC_SLATER: f = f_slater(R); break;
            case XC_B88: f = f_b88(R, S); break;
            case XC_VWN_RPA: f = f_vwn(R, 0.0310907, -0.409286, 13.0720, 42.7198); break;
            case XC_VWN5: f = f_vwn(R, 0.0310907, -0.10498, 3.72744, 12.9352); break;
            case XC_LYP: f = f_lyp(R, S); break;
            case XC_PBE_X: f = f_pbe_x(R, S); break;
            case XC_PBE_C: f = f_pbe_c(R, S); break;
            default: f = D2(0.0);
            }
            acc = acc + D2(X.coef[t]) * f;
        }
        e = acc.v; vr = acc.d[0]; vs = acc.d[1];
    }
    if (exc) exc[g] = e;
    if (vrho_out) vrho_out[g] = vr;
    if (vsig_out) vsig_out[g] = vs;
    if (wv) {
        double ww = w[g];
        wv[g] = 0.5 * ww * vr;
        if (gga) {
            double f = 2.0 * ww * vs;
            wv[ng + g] = f * gx; wv[2 * ng + g] = f * gy; wv[3 * ng + g] = f * gz;
        }
    }
}

// Spin-polarised evaluation (UKS): rho_s[0] = density, rho_s[1..3] = its gradient.  exc[g] = e per volume;
// wv_s[0] = 0.5 w de/drho_s ; wv_s[1..3] = w (2 de/dsigma_ss grad rho_s + de/dsigma_ab grad rho_s')  (same consumer as the
// closed-shell wv: V_s = ao^T aow_s + transpose)
__global__ __launch_bounds__(256) void xc_eval_spin_kernel(XcSpec X, const double *rhoa, const double *rhob, const double *w, int64_t ng,
                                                           int gga, double *exc, double *wva, double *wvb)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const double ra = fmax(rhoa[g], 0.0), rb = fmax(rhob[g], 0.0);
    double ga[3] = {0, 0, 0}, gb[3] = {0, 0, 0};
    if (gga)
        for (int k = 0; k < 3; k++) { ga[k] = rhoa[(k + 1) * ng + g]; gb[k] = rhob[(k + 1) * ng + g]; }
    double e = 0.0, v[5] = {0, 0, 0, 0, 0};
    if (ra + rb > 1e-10) {
        D5 Ra = D5::var(ra, 0), Rb = D5::var(rb, 1);
        D5 Saa = D5::var(ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2], 2);
        D5 Sab = D5::var(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2], 3);
        D5 Sbb = D5::var(gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2], 4);
        D5 acc(0.0);
        for (int t = 0; t < X.n; t++) {
            D5 f(0.0);
            switch (X.kind[t]) {
            case XC_SLATER: f = spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](D5 r, D5) { return f_slater(r); }); break;
            case XC_B88: f = spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](D5 r, D5 s_) { return f_b88(r, s_); }); break;
            case XC_PBE_X: f = spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](D5 r, D5 s_) { return f_pbe_x(r, s_); }); break;
            case XC_VWN_RPA: f = f_vwn_rpa_spin(Ra, Rb); break;
            case XC_VWN5: f = f_vwn5_spin(Ra, Rb); break;
            case XC_LYP: f = f_lyp_spin(Ra, Rb, Saa, Sab, Sbb); break;
            case XC_PBE_C: f = f_pbe_c_spin(Ra, Rb, Saa, Sab, Sbb); break;
            default: break;
            }
            acc = acc + D5(X.coef[t]) * f;
        }
        e = acc.v;
        for (int k = 0; k < 5; k++) v[k] = acc.d[k];
    }
    if (exc) exc[g] = e;
    const double ww = w[g];
    wva[g] = 0.5 * ww * v[0];
    wvb[g] = 0.5 * ww * v[1];
    if (gga)
        for (int k = 0; k < 3; k++) {
            wva[(k + 1) * ng + g] = ww * (2.0 * v[2] * ga[k] + v[3] * gb[k]);
            wvb[(k + 1) * ng + g] = ww * (2.0 * v[4] * gb[k] + v[3] * ga[k]);
        }
}

extern "C" int mi_xc_eval_spin(const int32_t *kinds, const double *coefs, int nterms, const double *d_rhoa, const double *d_rhob,
                               const double *d_w, int64_t ng, int gga, double *d_exc, double *d_wva, double *d_wvb, void *stream)
{
    if (nterms < 0 || nterms > 8) return fail("mi_xc_eval_spin: at most 8 functional terms");
    if (!d_rhoa || !d_rhob || !d_w || !d_wva || !d_wvb) return fail("mi_xc_eval_spin: null argument");
    XcSpec X{};
    X.n = nterms;
    for (int i = 0; i < nterms; i++) { X.kind[i] = kinds[i]; X.coef[i] = coefs[i]; }
    if (ng <= 0) return 0;
    hipLaunchKernelGGL(xc_eval_spin_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, d_rhoa, d_rhob, d_w,
                       ng, gga, d_exc, d_wva, d_wvb);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_xc_eval(const int32_t *kinds, const double *coefs, int nterms, const double *d_rho, const double *d_w,
                          int64_t ng, int gga, double *d_exc, double *d_wv, double *d_vrho, double *d_vsigma, void *stream)
{
    if (nterms < 0 || nterms > 8) return fail("mi_xc_eval: at most 8 functional terms");
    if (!d_rho || (d_wv && !d_w)) return fail("mi_xc_eval: null argument");
    XcSpec X{};
    X.n = nterms;
    for (int i = 0; i < nterms; i++) {
        if (kinds[i] < XC_SLATER || kinds[i] > XC_PBE_C) return fail("mi_xc_eval: unknown functional id %d", kinds[i]);
        X.kind[i] = kinds[i]; X.coef[i] = coefs[i];
    }
    hipLaunchKernelGGL(xc_eval_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, d_rho, d_w, ng, gga,
                       d_exc, d_wv, d_vrho, d_vsigma);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- meta-GGA evaluation.  Closed shell: rho[0..3] = density and gradient, tau = 1/2 sum_i |grad phi_i|^2 (all electrons);
// wv[0] = 0.5 w de/drho, wv[1..3] = 2 w de/dsigma grad rho, wv[4] = 0.25 w de/dtau  (V = vmat + vmat^T with
// vmat = ao0^T (wv0 ao0 + wv_k ao_k) + sum_k ao_k^T (wv4 ao_k)).
typedef DN<3> D3;
typedef DN<7> D7;
template <class T>
__device__ inline T xc_term_spin(int kind, T Ra, T Rb, T Saa, T Sab, T Sbb, T Ta, T Tb)
{
    switch (kind) {
    case XC_SLATER: return spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](T r, T) { return f_slater(r); });
    case XC_B88: return spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](T r, T s_) { return f_b88(r, s_); });
    case XC_PBE_X: return spin_scaled_exchange(Ra, Rb, Saa, Sbb, [](T r, T s_) { return f_pbe_x(r, s_); });
    case XC_VWN_RPA: return f_vwn_rpa_spin(Ra, Rb);
    case XC_VWN5: return f_vwn5_spin(Ra, Rb);
    case XC_LYP: return f_lyp_spin(Ra, Rb, Saa, Sab, Sbb);
    case XC_PBE_C: return f_pbe_c_spin(Ra, Rb, Saa, Sab, Sbb);
    case XC_TPSS_X: return f_tpss_x_spin(Ra, Rb, Saa, Sbb, Ta, Tb);
    case XC_TPSS_C: return f_tpss_c_spin(Ra, Rb, Saa, Sab, Sbb, Ta, Tb);
    case XC_M062X_X: return f_m062x_x_spin(Ra, Rb, Saa, Sbb, Ta, Tb);
    case XC_M062X_C: return f_m062x_c_spin(Ra, Rb, Saa, Sbb, Ta, Tb);
    default: return T(0.0);
    }
}

__global__ __launch_bounds__(256) void xc_eval_mgga_kernel(XcSpec X, const double *rho, const double *tau, const double *w, int64_t ng,
                                                           double *exc, double *wv)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const double r = rho[g], gx = rho[ng + g], gy = rho[2 * ng + g], gz = rho[3 * ng + g], t = tau[g];
    double e = 0.0, vr = 0.0, vs = 0.0, vt = 0.0;
    if (r > 1e-10) {
        D3 R = D3::var(r, 0), S = D3::var(gx * gx + gy * gy + gz * gz, 1), Tt = D3::var(fmax(t, 0.0), 2);
        D3 Rh = R * D3(0.5), S4 = S * D3(0.25), Th = Tt * D3(0.5);
        D3 acc(0.0);
        for (int q = 0; q < X.n; q++) acc = acc + D3(X.coef[q]) * xc_term_spin(X.kind[q], Rh, Rh, S4, S4, S4, Th, Th);
        e = acc.v; vr = acc.d[0]; vs = acc.d[1]; vt = acc.d[2];
    }
    if (exc) exc[g] = e;
    if (wv) {
        const double ww = w[g], f = 2.0 * ww * vs;
        wv[g] = 0.5 * ww * vr;
        wv[ng + g] = f * gx; wv[2 * ng + g] = f * gy; wv[3 * ng + g] = f * gz;
        wv[4 * ng + g] = 0.25 * ww * vt;
    }
}

__global__ __launch_bounds__(256) void xc_eval_mgga_spin_kernel(XcSpec X, const double *rhoa, const double *rhob, const double *taua,
                                                                const double *taub, const double *w, int64_t ng, double *exc, double *wva,
                                                                double *wvb)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const double ra = fmax(rhoa[g], 0.0), rb = fmax(rhob[g], 0.0);
    double ga[3], gb[3];
    for (int k = 0; k < 3; k++) { ga[k] = rhoa[(k + 1) * ng + g]; gb[k] = rhob[(k + 1) * ng + g]; }
    double e = 0.0, v[7] = {0, 0, 0, 0, 0, 0, 0};
    if (ra + rb > 1e-10) {
        D7 Ra = D7::var(ra, 0), Rb = D7::var(rb, 1);
        D7 Saa = D7::var(ga[0] * ga[0] + ga[1] * ga[1] + ga[2] * ga[2], 2);
        D7 Sab = D7::var(ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2], 3);
        D7 Sbb = D7::var(gb[0] * gb[0] + gb[1] * gb[1] + gb[2] * gb[2], 4);
        D7 Ta = D7::var(fmax(taua[g], 0.0), 5), Tb = D7::var(fmax(taub[g], 0.0), 6);
        D7 acc(0.0);
        for (int q = 0; q < X.n; q++) acc = acc + D7(X.coef[q]) * xc_term_spin(X.kind[q], Ra, Rb, Saa, Sab, Sbb, Ta, Tb);
        e = acc.v;
        for (int k = 0; k < 7; k++) v[k] = acc.d[k];
    }
    if (exc) exc[g] = e;
    const double ww = w[g];
    wva[g] = 0.5 * ww * v[0];
    wvb[g] = 0.5 * ww * v[1];
    for (int k = 0; k < 3; k++) {
        wva[(k + 1) * ng + g] = ww * (2.0 * v[2] * ga[k] + v[3] * gb[k]);
        wvb[(k + 1) * ng + g] = ww * (2.0 * v[4] * gb[k] + v[3] * ga[k]);
    }
    wva[4 * ng + g] = 0.25 * ww * v[5];
    wvb[4 * ng + g] = 0.25 * ww * v[6];
}

static int fill_xc_spec(XcSpec &X, const int32_t *kinds, const double *coefs, int nterms)
{
    if (nterms < 0 || nterms > 8) return fail("xc: at most 8 functional terms");
    X = XcSpec{};
    X.n = nterms;
    for (int i = 0; i < nterms; i++) {
        if (kinds[i] < XC_SLATER || kinds[i] > XC_M062X_C) return fail("xc: unknown functional id %d", kinds[i]);
        X.kind[i] = kinds[i]; X.coef[i] = coefs[i];
    }
    return 0;
}

extern "C" int mi_xc_eval_mgga(const int32_t *kinds, const double *coefs, int nterms, const double *d_rho, const double *d_tau,
                               const double *d_w, int64_t ng, double *d_exc, double *d_wv, void *stream)
{
    if (!d_rho || !d_tau || (d_wv && !d_w)) return fail("mi_xc_eval_mgga: null argument");
    XcSpec X;
    if (fill_xc_spec(X, kinds, coefs, nterms)) return -1;
    if (ng <= 0) return 0;
    hipLaunchKernelGGL(xc_eval_mgga_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, d_rho, d_tau, d_w, ng,
                       d_exc, d_wv);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int mi_xc_eval_mgga_spin(const int32_t *kinds, const double *coefs, int nterms, const double *d_rhoa, const double *d_rhob,
                                    const double *d_taua, const double *d_taub, const double *d_w, int64_t ng, double *d_exc, double *d_wva,
                                    double *d_wvb, void *stream)
{
    if (!d_rhoa || !d_rhob || !d_taua || !d_taub || !d_w || !d_wva || !d_wvb) return fail("mi_xc_eval_mgga_spin: null argument");
    XcSpec X;
    if (fill_xc_spec(X, kinds, coefs, nterms)) return -1;
    if (ng <= 0) return 0;
    hipLaunchKernelGGL(xc_eval_mgga_spin_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, d_rhoa, d_rhob,
                       d_taua, d_taub, d_w, ng, d_exc, d_wva, d_wvb);
    HIPCHK(hipGetLastError());
    return 0;
}

// aow[mu][g] = ao0*wv0 + sum_k ao_k*wv_k
__global__ __launch_bounds__(256) void xc_aow_kernel(const double *ao, const double *wv, int nao, int64_t ng, int gga, double *aow)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    const size_t comp = (size_t)nao * ng;
    double w0 = wv[g], w1 = gga ? wv[ng + g] : 0, w2 = gga ? wv[2 * ng + g] : 0, w3 = gga ? wv[3 * ng + g] : 0;
    for (int m = blockIdx.y; m < nao; m += gridDim.y) {
        size_t o = (size_t)m * ng + g;
        double v = ao[o] * w0;
        if (gga) v += ao[comp + o] * w1 + ao[2 * comp + o] * w2 + ao[3 * comp + o] * w3;
        aow[o] = v;
    }
}

extern "C" int mi_xc_aow(mi_ctx *c, const double *d_ao, const double *d_wv, int64_t ng, int gga, double *d_aow, void *stream)
{
    if (!c || !d_ao || !d_wv || !d_aow) return fail("mi_xc_aow: null argument");
    dim3 grid((unsigned)((ng + 255) / 256), (unsigned)std::min(c->nao, 64));
    hipLaunchKernelGGL(xc_aow_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_ao, d_wv, c->nao, ng, gga, d_aow);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// SP2 density-matrix purification helpers (row a11 without a diagonalisation): the X*X products are
// rocBLAS DGEMMs; these kernels fuse everything else so one purification step is 2 launches.
// =================================================================================================
// Gershgorin discs, one wave per row (coalesced): rows[r] = F_rr - R_r, rows[n + r] = F_rr + R_r
__global__ __launch_bounds__(256) void sp2_bounds_kernel(const double *F, int n, double *rows)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    double s = 0.0;
    for (int c = lane; c < n; c += 64) s += fabs(F[(size_t)r * n + c]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        double d = F[(size_t)r * n + r];
        s -= fabs(d);
        rows[r] = d - s;
        rows[n + r] = d + s;
    }
}

// X0 = (emax I - F) / (emax - emin); every block reduces the 2n disc bounds itself (cheap, no extra launch)
__global__ __launch_bounds__(256) void sp2_init_kernel(const double *F, const double *rows, int n, double *X)
{
    __shared__ double smin[256], smax[256];
    double lo = 1e300, hi = -1e300;
    for (int r = threadIdx.x; r < n; r += 256) { lo = fmin(lo, rows[r]); hi = fmax(hi, rows[n + r]); }
    smin[threadIdx.x] = lo; smax[threadIdx.x] = hi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
            smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
        }
        __syncthreads();
    }
    const double emin = smin[0], emax = smax[0];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    int r = (int)(idx / n), c = (int)(idx - (size_t)r * n);
    X[idx] = ((r == c ? emax : 0.0) - F[idx]) / (emax - emin);
}

// out <- X2 if |tr X2 - N| < |2 tr X - tr X2 - N| else 2X - X2 ; traces recomputed per block (2n loads)
__global__ __launch_bounds__(256) void sp2_update_kernel(double *X, const double *X2, int n, double target, double *traces)
{
    __shared__ double s1[256], s2[256];
    double a = 0.0, b = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) { a += X[(size_t)r * n + r]; b += X2[(size_t)r * n + r]; }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { s1[threadIdx.x] += s1[threadIdx.x + o]; s2[threadIdx.x] += s2[threadIdx.x + o]; }
        __syncthreads();
    }
    const double tx = s1[0], tx2 = s2[0];
    const bool sq = fabs(tx2 - target) < fabs(2.0 * tx - tx2 - target);
    if (blockIdx.x == 0 && threadIdx.x == 0) { traces[0] = tx; traces[1] = tx2; }
    // the update goes to a separate buffer (after the two trace slots), so no block races with another
    // block's diagonal reads of X
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)n * n) {
        double x = X[idx], x2 = X2[idx];
        traces[2 + idx] = sq ? x2 : 2.0 * x - x2; // out buffer follows the 2 trace slots
    }
}

extern "C" int mi_sp2_init(mi_ctx *c, const double *d_F, double *d_X, double *d_work, void *stream)
{
    if (!c || !d_F || !d_X || !d_work) return fail("mi_sp2_init: null argument");
    int n = c->nao;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sp2_bounds_kernel, dim3((n + 3) / 4), dim3(256), 0, st, d_F, n, d_work);
    hipLaunchKernelGGL(sp2_init_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, st, d_F, d_work, n, d_X);
    HIPCHK(hipGetLastError());
    return 0;
}

// d_out_with_traces: [2 + n*n] doubles: traces (tr X, tr X2 of the INPUT) then the updated matrix
extern "C" int mi_sp2_update(mi_ctx *c, double *d_X, const double *d_X2, double n_occ, double *d_out_with_traces, void *stream)
{
    if (!c || !d_X || !d_X2 || !d_out_with_traces) return fail("mi_sp2_update: null argument");
    int n = c->nao;
    hipLaunchKernelGGL(sp2_update_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_X, d_X2, n, n_occ,
                       d_out_with_traces);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// Analytic nuclear gradient, one-electron part (row a15):
//   g[X] += 2 sum_{mu on X, nu} [ D_mu,nu <d mu|T+V|nu> - W_mu,nu <d mu|nu> ]  (basis-function derivative)
//   g[C] -= 2 sum_{mu,nu} D_mu,nu <d mu|v_C|nu>                               (Hellmann-Feynman, via
//                                                                               translational invariance)
// One thread per ORDERED shell pair (ish, jsh); the derivative acts on the first shell.  d/dA_x of a
// primitive cartesian Gaussian = 2a (a+1_x) - a_x (a-1_x), applied inside the primitive loop.
// =================================================================================================
struct Grad1eArgs {
    const int32_t *atm, *bas;
    const double *env;
    const int *shell_ao;
    const double *c2s;
    int c2s_off[LMAX + 2];
    RysDev rys;
    int natm, nbas, nao;
    const double *D, *W; // [nao][nao]
    double *grad;        // [natm][3]
};

__global__ __launch_bounds__(64) void int1e_grad_kernel(Grad1eArgs A)
{
    int pid = blockIdx.x * blockDim.x + threadIdx.x;
    if (pid >= A.nbas * A.nbas) return;
    // grid.y = natm + 1 slices, as in int1e_kernel: y < natm: nuclear attraction of nucleus y; y == natm: overlap + kinetic
    const int vatom = (int)blockIdx.y < A.natm ? (int)blockIdx.y : -1;
    if (vatom >= 0 && A.atm[vatom * ATM_SLOTS + 0] == 0) return;
    int ish = pid / A.nbas, jsh = pid - ish * A.nbas;
    const int32_t *bi = A.bas + ish * BAS_SLOTS, *bj = A.bas + jsh * BAS_SLOTS;
    int la = bi[1], lb = bj[1];
    const double *ra = A.env + A.atm[bi[0] * ATM_SLOTS + 1], *rb = A.env + A.atm[bj[0] * ATM_SLOTS + 1];
    int nca = (la + 1) * (la + 2) / 2, ncb = (lb + 1) * (lb + 2) / 2;
    int nsa = 2 * la + 1, nsb = 2 * lb + 1;
    int ao_i = A.shell_ao[ish], ao_j = A.shell_ao[jsh];
    const double *ca = A.c2s + A.c2s_off[la], *cb = A.c2s + A.c2s_off[lb];
    // back-transform the density blocks to cartesian components
    double Dc[NC1 * NC1], Wc[NC1 * NC1];
    for (int a = 0; a < nca; a++)
        for (int b = 0; b < ncb; b++) {
            double sd = 0.0, sw = 0.0;
            for (int i = 0; i < nsa; i++) {
                double t = 0.0, u = 0.0;
                for (int j = 0; j < nsb; j++) {
                    t += A.D[(size_t)(ao_i + i) * A.nao + ao_j + j] * cb[b * nsb + j];
                    u += A.W[(size_t)(ao_i + i) * A.nao + ao_j + j] * cb[b * nsb + j];
                }
                sd += ca[a * nsa + i] * t; sw += ca[a * nsa + i] * u;
            }
            Dc[a * ncb + b] = sd; Wc[a * ncb + b] = sw;
        }
    double gA[3] = {0.0, 0.0, 0.0};
    double AB[3] = {ra[0] - rb[0], ra[1] - rb[1], ra[2] - rb[2]};
    for (int ip = 0; ip < bi[2]; ip++)
        for (int jp = 0; jp < bj[2]; jp++) {
            double a = A.env[bi[5] + ip], b = A.env[bj[5] + jp];
            double cc = A.env[bi[6] + ip] * A.env[bj[6] + jp];
            double p = a + b, mu = a * b / p, h = 0.5 / p;
            double ex = exp(-mu * (AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2]));
            double P[3], PA[3], PB[3];
            for (int d = 0; d < 3; d++) { P[d] = (a * ra[d] + b * rb[d]) / p; PA[d] = P[d] - ra[d]; PB[d] = P[d] - rb[d]; }
            double s[3][LMAX_1E + 3][LMAX_1E + 3]; // s[d][i][j], i <= la+1, j <= lb+2
            for (int d = 0; d < 3; d++) {
                s[d][0][0] = 1.0;
                for (int i = 0; i <= la; i++) s[d][i + 1][0] = PA[d] * s[d][i][0] + (i > 0 ? i * h * s[d][i - 1][0] : 0.0);
                for (int j = 0; j <= lb + 1; j++)
                    for (int i = 0; i <= la + 1; i++)
                        s[d][i][j + 1] = PB[d] * s[d][i][j] + (i > 0 ? i * h * s[d][i - 1][j] : 0.0) + (j > 0 ? j * h * s[d][i][j - 1] : 0.0);
            }
            double pref = cc * ex * pow(M_PI / p, 1.5);
            // overlap and kinetic: 1-D factors S(i,j), T(i,j) for i in {a-1, a, a+1}  (only in the y == natm slice)
            for (int ia = 0; ia < (vatom < 0 ? nca : 0); ia++) {
                int pa[3];
                cart_pow(la, ia, pa[0], pa[1], pa[2]);
                for (int ib = 0; ib < ncb; ib++) {
                    int pb[3];
                    cart_pow(lb, ib, pb[0], pb[1], pb[2]);
                    double s0[3], t0[3], sd[3], td[3]; // plain and differentiated 1-D factors
                    for (int d = 0; d < 3; d++) {
                        int i = pa[d], j = pb[d];
                        auto S1 = [&](int ii) { return s[d][ii][j]; };
                        auto T1 = [&](int ii) {
                            double t = -2.0 * b * (2 * j + 1) * s[d][ii][j] + 4.0 * b * b * s[d][ii][j + 2];
                            if (j >= 2) t += j * (j - 1) * s[d][ii][j - 2];
                            return -0.5 * t;
                        };
                        s0[d] = S1(i); t0[d] = T1(i);
                        sd[d] = 2.0 * a * S1(i + 1) - (i > 0 ? i * S1(i - 1) : 0.0);
                        td[d] = 2.0 * a * T1(i + 1) - (i > 0 ? i * T1(i - 1) : 0.0);
                    }
                    double dc = Dc[ia * ncb + ib], wc = Wc[ia * ncb + ib];
                    for (int x = 0; x < 3; x++) {
                        int y = (x + 1) % 3, z = (x + 2) % 3;
                        double dS = sd[x] * s0[y] * s0[z];
                        double dT = td[x] * s0[y] * s0[z] + sd[x] * t0[y] * s0[z] + sd[x] * s0[y] * t0[z];
                        gA[x] += 2.0 * pref * (dc * dT - wc * dS);
                    }
                }
            }
            // nuclear attraction with the differentiated bra: Rys, nroots = (la+lb+1)/2 + 1
            int nr = (la + lb + 1) / 2 + 1;
            double pv = cc * ex * 2.0 * M_PI / p;
            for (int ic = (vatom < 0 ? A.natm : vatom); ic < (vatom < 0 ? A.natm : vatom + 1); ic++) {
                double Z = A.atm[ic * ATM_SLOTS + 0];
                if (Z == 0.0) continue;
                const double *C = A.env + A.atm[ic * ATM_SLOTS + 1];
                double PC[3] = {P[0] - C[0], P[1] - C[1], P[2] - C[2]};
                double xarg = p * (PC[0] * PC[0] + PC[1] * PC[1] + PC[2] * PC[2]);
                double gc[3] = {0.0, 0.0, 0.0};
                for (int r = 0; r < nr; r++) {
                    double u = rys_eval(A.rys, nr, r, xarg), w = rys_eval(A.rys, nr, nr + r, xarg);
                    double g[3][2 * LMAX_1E + 2][LMAX_1E + 1];
                    double b10 = (1.0 - u) * h;
                    for (int d = 0; d < 3; d++) {
                        double c00 = PA[d] - u * PC[d];
                        g[d][0][0] = 1.0;
                        for (int i = 0; i < la + lb + 1; i++) g[d][i + 1][0] = c00 * g[d][i][0] + (i > 0 ? i * b10 * g[d][i - 1][0] : 0.0);
                        for (int j = 0; j < lb; j++)
                            for (int i = 0; i <= la + lb - j; i++) g[d][i][j + 1] = g[d][i + 1][j] + AB[d] * g[d][i][j];
                    }
                    double f = -Z * pv * w;
                    for (int ia = 0; ia < nca; ia++) {
                        int pa[3];
                        cart_pow(la, ia, pa[0], pa[1], pa[2]);
                        for (int ib = 0; ib < ncb; ib++) {
                            int pb[3];
                            cart_pow(lb, ib, pb[0], pb[1], pb[2]);
                            double v0[3], vd[3];
                            for (int d = 0; d < 3; d++) {
                                int i = pa[d], j = pb[d];
                                v0[d] = g[d][i][j];
                                vd[d] = 2.0 * a * g[d][i + 1][j] - (i > 0 ? i * g[d][i - 1][j] : 0.0);
                            }
                            double dc = f * Dc[ia * ncb + ib];
                            gc[0] += dc * vd[0] * v0[1] * v0[2];
                            gc[1] += dc * v0[0] * vd[1] * v0[2];
                            gc[2] += dc * v0[0] * v0[1] * vd[2];
                        }
                    }
                }
                for (int x = 0; x < 3; x++) {
                    gA[x] += 2.0 * gc[x];
                    atomicAdd(&A.grad[ic * 3 + x], -2.0 * gc[x]);
                }
            }
        }
    for (int x = 0; x < 3; x++) atomicAdd(&A.grad[bi[0] * 3 + x], gA[x]);
}

extern "C" int mi_grad_1e(mi_ctx *c, const double *d_D, const double *d_W, double *d_grad, void *stream)
{
    if (c && check_orbital_lmax(c, "mi_grad_1e")) return -1;
    if (!c || !d_D || !d_W || !d_grad) return fail("mi_grad_1e: null argument");
    HIPCHK(hipSetDevice(c->device));
    Grad1eArgs A;
    A.atm = c->d_atm; A.bas = c->d_bas; A.env = c->d_env; A.shell_ao = c->d_shell_ao; A.c2s = c->d_c2s;
    for (int i = 0; i <= LMAX + 1; i++) A.c2s_off[i] = c->c2s_off[i];
    A.rys = c->rys; A.natm = c->natm; A.nbas = c->nbas; A.nao = c->nao;
    A.D = d_D; A.W = d_W; A.grad = d_grad;
    int n = c->nbas * c->nbas;
    hipLaunchKernelGGL(int1e_grad_kernel, dim3((n + 63) / 64, c->natm + 1), dim3(64), 0, (hipStream_t)stream, A);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// Analytic nuclear gradient, two-electron part (row a15).
//
//   dE2/dR_X = 2 sum_{mu on X; nu,lam,sig} (d mu nu|lam sig) G,  G = D_mn D_ls - (hyb/4)(D_ml D_ns + D_ms D_nl)
//
// For every Schwarz-surviving canonical shell quartet the four role permutations (ij|kl),(ji|kl),(kl|ij),
// (lk|ij) are evaluated with the derivative on the FIRST shell.  d/dA of a contracted shell of angular
// momentum l is a combination of an (l+1) shell with coefficients 2*alpha*c and an (l-1) shell with
// coefficients c, so the same Rys kernel produces the two [e0|f0] blocks; `eri_grad_contract` applies
// HRR + derivative + cart->sph folded into per-pair matrices, contracts with G and adds 4*w_q*sum to
// grad[atom of the first shell].
// =================================================================================================
static void build_M_deriv(int l1, int l2, int sign, const double AB[3], const std::vector<double> &c1,
                          const std::vector<double> &c2, double *M /* [3][ns1*ns2][ne'] */)
{
    int lp = l1 + sign;
    int ns1 = 2 * l1 + 1, ns2 = 2 * l2 + 1, ne = ne_of(lp, l2);
    std::fill(M, M + (size_t)3 * ns1 * ns2 * ne, 0.0);
    int eoff[2 * LMAX + 4];
    eoff[lp] = 0;
    for (int e = lp; e < lp + l2 + 1; e++) eoff[e + 1] = eoff[e] + ncart(e);
    double pw[3][LMAX + 1];
    for (int d = 0; d < 3; d++) { pw[d][0] = 1.0; for (int k = 1; k <= LMAX; k++) pw[d][k] = pw[d][k - 1] * AB[d]; }
    int ia = 0;
    for (int ax = l1; ax >= 0; ax--)
        for (int ay = l1 - ax; ay >= 0; ay--, ia++) {
            int a[3] = {ax, ay, l1 - ax - ay};
            for (int x = 0; x < 3; x++) {
                double dcoef;
                int ap[3] = {a[0], a[1], a[2]};
                if (sign > 0) { ap[x] += 1; dcoef = 1.0; }
                else { if (a[x] == 0) continue; ap[x] -= 1; dcoef = -(double)a[x]; }
                int ib = 0;
                for (int bx = l2; bx >= 0; bx--)
                    for (int by = l2 - bx; by >= 0; by--, ib++) {
                        int bz = l2 - bx - by;
                        for (int ix = 0; ix <= bx; ix++)
                            for (int iy = 0; iy <= by; iy++)
                                for (int iz = 0; iz <= bz; iz++) {
                                    double coef = dcoef * binom(bx, ix) * binom(by, iy) * binom(bz, iz) * pw[0][bx - ix] * pw[1][by - iy] * pw[2][bz - iz];
                                    int deg = lp + ix + iy + iz;
                                    int e = eoff[deg] + cart_index(deg, ap[0] + ix, ap[1] + iy);
                                    for (int sa = 0; sa < ns1; sa++) {
                                        double v1 = c1[(size_t)ia * ns1 + sa];
                                        if (v1 == 0.0) continue;
                                        for (int sb = 0; sb < ns2; sb++) {
                                            double v2 = c2[(size_t)ib * ns2 + sb];
                                            if (v2 != 0.0) M[((size_t)x * ns1 * ns2 + sa * ns2 + sb) * ne + e] += v1 * v2 * coef;
                                        }
                                    }
                                }
                    }
            }
        }
}

// host half: sizes, primitive records and derivative matrices of every variant pair (no HIP call: may run on the helper thread)
static int grad_records_host(mi_ctx *c)
{
    std::vector<std::vector<double>> c2s(LMAX + 1);
    for (int l = 0; l <= LMAX; l++) c2s_generic(l, c2s[l]);
    std::vector<double> &prim = c->h_prim, &Mbuf = c->h_M;
    // pass 1 (serial, cheap): sizes and offsets of every variant record
    struct Job { int ci, r, o, sg; };
    std::vector<Job> jobs;
    size_t prim_end = prim.size() / 8, m_end = Mbuf.size();
    for (int ci = 0; ci < NPC; ci++) {
        PairClass &P = c->pc[ci];
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++) P.g_recs[o][sg].assign(P.recs.size(), PairRec{-1, -1, 0, 0, 0, 0, 0, 0});
        for (size_t r = 0; r < P.recs.size(); r++)
            for (int o = 0; o < 2; o++) {
                int s1 = o == 0 ? P.recs[r].sh_i : P.recs[r].sh_j, s2 = o == 0 ? P.recs[r].sh_j : P.recs[r].sh_i;
                const ShellH &I = c->shells[s1], &J = c->shells[s2];
                double r2 = 0.0;
                for (int d = 0; d < 3; d++) r2 += (I.r[d] - J.r[d]) * (I.r[d] - J.r[d]);
                int np = 0;
                for (int ip = 0; ip < I.nprim; ip++)
                    for (int jp = 0; jp < J.nprim; jp++)
                        if (I.exps[ip] * J.exps[jp] / (I.exps[ip] + J.exps[jp]) * r2 <= 80.0) np++;
                for (int sg = 0; sg < 2; sg++) {
                    int sign = sg == 0 ? +1 : -1;
                    if (I.l + sign < 0) continue;
                    PairRec R;
                    R.sh_i = s1; R.sh_j = s2; R.ao_i = I.ao; R.ao_j = J.ao; R.pad = 0;
                    R.prim_off = (int)prim_end; R.nprim = np;
                    prim_end += np;
                    size_t msz = (size_t)3 * (2 * I.l + 1) * (2 * J.l + 1) * ne_of(I.l + sign, J.l);
                    if (m_end + msz > (size_t)INT32_MAX) return fail("gradient transformation matrices exceed 2^31 doubles");
                    R.m_off = (int)m_end;
                    m_end += msz;
                    P.g_recs[o][sg][r] = R;
                    jobs.push_back({ci, (int)r, o, sg});
                }
            }
    }
    prim.resize(prim_end * 8);
    Mbuf.resize(m_end);
    // pass 2 (OpenMP): fill primitive records and HRR*derivative*c2s matrices
#pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads())
    for (size_t q = 0; q < jobs.size(); q++) {
        const Job jb = jobs[q];
        PairClass &P = c->pc[jb.ci];
        const PairRec &R = P.g_recs[jb.o][jb.sg][jb.r];
        const ShellH &I = c->shells[R.sh_i], &J = c->shells[R.sh_j];
        const int sign = jb.sg == 0 ? +1 : -1;
        double AB[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
        double r2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
        double *dst = prim.data() + (size_t)R.prim_off * 8;
        for (int ip = 0; ip < I.nprim; ip++)
            for (int jp = 0; jp < J.nprim; jp++) {
                double a = I.exps[ip], b = J.exps[jp], p = a + b, mu = a * b / p;
                if (mu * r2 > 80.0) continue;
                double K = I.coef[ip] * J.coef[jp] * std::exp(-mu * r2) * (sign > 0 ? 2.0 * a : 1.0);
                double Pc[3];
                for (int d = 0; d < 3; d++) Pc[d] = (a * I.r[d] + b * J.r[d]) / p;
                double rec[8] = {p, Pc[0], Pc[1], Pc[2], Pc[0] - I.r[0], Pc[1] - I.r[1], Pc[2] - I.r[2], K};
                memcpy(dst, rec, sizeof rec);
                dst += 8;
            }
        build_M_deriv(I.l, J.l, sign, AB, c2s[I.l], c2s[J.l], Mbuf.data() + R.m_off);
    }
    c->grad_host_ready = true;
    return 0;
}

static int prepare_grad_records(mi_ctx *c)
{
    if (c->grad_ready) return 0;
    if (c->grad_worker.joinable()) {
        c->grad_worker.join();
        if (c->grad_host_rc) return fail("%s", c->grad_host_err.c_str());
    }
    if (!c->grad_host_ready && grad_records_host(c)) return -1;
    std::vector<double> &prim = c->h_prim, &Mbuf = c->h_M;
    for (int ci = 0; ci < NPC; ci++)
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++)
                if (upload(&c->pc[ci].d_g_recs[o][sg], c->pc[ci].g_recs[o][sg])) return -1;
    if (upload(&c->d_prim, prim)) return -1;
    if (upload(&c->d_M, Mbuf)) return -1;
    c->grad_ready = true;
    return 0;
}

struct GradXfArgs {
    const PairRec *dplus, *dminus; // derivative variants of the differentiated pair (dminus may be null)
    const PairRec *ket;            // base records of the other pair
    const double *Mbuf;
    const int64_t *prefix;
    int nbra;
    int64_t t0;
    int swap, same_class;
    int ne_p, ne_m, nf, ns1, ns2, nscd, nsd;
    const double *work_p, *work_m;
    int ncomp_p, ncomp_m;
    const double *D; // padded total density
    const double *Dm; // padded spin density Da - Db (UHF/UKS) or nullptr
    int ld;
    double hyb;
    const int *shell_atom;
    double *grad;
    int inv_from_second; // translational invariance: 1: the skipped shell is dp.sh_j, 0: it is cd.sh_i
    int natm3;           // grad points to GRAD_COPIES private copies of [natm*3] (atomic contention relief)
    int64_t nbatch;      // tasks in this launch
    const TaskIdx *tasks; // non-null: (bra, ket) per task, precomputed (get_task)
    const double *q_bra, *q_ket, *dmax; // same screening as EriArgs (the Rys kernel left these quartets' blocks unwritten)
    int nbas_d;
    double dtol;
    // density fitting (mi_df_grad, template flag DF): the two-particle density is not a product of D's but a dense tensor,
    // G[(a b), (P)] = Z[a * zs_i + b * zs_j + P]  (three-index: Z3[a][b][P]; two-index: Z2[P][Q], zs_j = 0), the ket is an
    // auxiliary "pair" (P, unit function) whose atoms come from `ket_atom`, and the weight is w0 instead of 4.
    const double *Z;
    int64_t zs_i, zs_j;
    const int *ket_atom;
    double w0;
};
#define GRAD_COPIES 4096

// GSZ = lanes per quartet: 16 (four quartets per wave, small angular classes) or 64 (256 = four waves sharing one
// quartet's LDS blocks is supported by the code but was measured slower and is not launched).
template <int GSZ, bool MFMA, bool DF = false>
__global__ __launch_bounds__(GSZ > 64 ? GSZ : 64) void eri_grad_contract(GradXfArgs A)
{
    // sum_{x-independent part first}:  g[x] = sum_{r,e} M^x[r][e] * Z[r][e],  Z[r][e] = sum_f E0[e][f] Y[r][f],
    // Y[r][f] = sum_c G[r][c] Mcd[c][f]  -- contracting the two-particle density FIRST makes the work
    // independent of the derivative direction (3x fewer flops than forming the derivative integrals).
    extern __shared__ double lds_all[];
    constexpr int QPW = GSZ >= 64 ? 1 : 64 / GSZ;
    constexpr int NW = GSZ >= 64 ? GSZ / 64 : 1;
    const int wl = threadIdx.x & 63, wave = threadIdx.x >> 6; // lane within the wave / wave within the workgroup (MFMA tiles)
    const int grp = threadIdx.x / GSZ, lane = threadIdx.x % GSZ;
    const int64_t tl = (int64_t)blockIdx.x * QPW + grp; // task index inside this batch
    const bool in_batch = tl < A.nbatch;
    const bool has_m = A.dminus != nullptr && A.ne_m > 0;
    const int nsab = A.ns1 * A.ns2, nf = A.nf;
    const size_t region = (size_t)A.ne_p * nf + (has_m ? (size_t)A.ne_m * nf : 0) + (size_t)nsab * A.nscd + (size_t)nsab * nf;
    double *lds = lds_all + (size_t)grp * region;
    double *E0p = lds;                                        // [ne_p][nf]
    double *E0m = E0p + (size_t)A.ne_p * nf;                  // [ne_m][nf]
    double *G = E0m + (has_m ? (size_t)A.ne_m * nf : 0);      // [nsab][nscd]
    double *Y = G + (size_t)nsab * A.nscd;                    // [nsab][nf]
    int ib = 0, ik = 0;
    if (in_batch) get_task(A.tasks, A.prefix, A.nbra, A.t0 + tl, ib, ik);
    const bool same_pair = A.same_class && ib == ik;
    if (A.swap) { int t_ = ib; ib = ik; ik = t_; }
    const PairRec dp = A.dplus[ib], cd = A.ket[ik];
    const bool live = in_batch && !(A.dmax && A.q_bra[ib] * A.q_ket[ik] *
                                    quartet_density_bound(A.dmax, A.nbas_d, dp.sh_i, dp.sh_j, cd.sh_i, cd.sh_j, A.hyb) < A.dtol);
    int m_off_m = 0;
    const double *D = A.D;
    const int ld = A.ld;
    // the first E0PRE x GSZ elements of the plus block travel through registers: their loads are issued here and land in LDS only
    // after the Y product, so the HBM round trip of the hand-over block overlaps the G gather and the first matrix product
    constexpr int E0PRE = (GSZ >= 64 && MFMA) ? 8 : 0;
    double pre[E0PRE > 0 ? E0PRE : 1];
    const int n_e0p = A.ne_p * nf;
    if (live) {
        const double *gp = A.work_p + (size_t)tl * A.ncomp_p;
        if (E0PRE > 0) {
#pragma unroll
            for (int q = 0; q < E0PRE; q++) { const int c = lane + q * GSZ; pre[q] = c < n_e0p ? gp[c] : 0.0; }
        }
        for (int c = lane + E0PRE * GSZ; c < n_e0p; c += GSZ) E0p[c] = gp[c];
        if (has_m) {
            m_off_m = A.dminus[ib].m_off;
            const double *gm = A.work_m + (size_t)tl * A.ncomp_m;
            for (int c = lane; c < A.ne_m * nf; c += GSZ) E0m[c] = gm[c];
        }
        for (int o = lane; o < nsab * A.nscd; o += GSZ) {
            int r = o / A.nscd, c = o - r * A.nscd;
            int sa = r / A.ns2, sb = r - sa * A.ns2, sc = c / A.nsd, sd = c - sc * A.nsd;
            int i = dp.ao_i + sa, j = dp.ao_j + sb, k = cd.ao_i + sc, l = cd.ao_j + sd;
            if (DF) { G[o] = A.Z[(int64_t)i * A.zs_i + (int64_t)j * A.zs_j + k]; continue; }
            double ex = D[(size_t)i * ld + k] * D[(size_t)j * ld + l] + D[(size_t)i * ld + l] * D[(size_t)j * ld + k];
            if (A.Dm) // sum_s Ds x Ds = (D x D + M x M) / 2
                ex += A.Dm[(size_t)i * ld + k] * A.Dm[(size_t)j * ld + l] + A.Dm[(size_t)i * ld + l] * A.Dm[(size_t)j * ld + k];
            G[o] = D[(size_t)i * ld + j] * D[(size_t)k * ld + l] - 0.25 * A.hyb * ex;
        }
    }
    __syncthreads();
    // the three small matrix products run on the FP64 MFMA pipe when the whole wave works on one quartet and the
    // 16x16x4 padding does not eat the gain (wave-uniform decisions); otherwise plain per-lane dot products
    if (live) {
        const double *Mcd = A.Mbuf + cd.m_off;
        if (MFMA && GSZ >= 64 && mfma_worthwhile(nsab, nf, A.nscd)) {
            int tile = 0;
            for (int m0 = 0; m0 < nsab; m0 += 16)
                for (int n0 = 0; n0 < nf; n0 += 16) {
                    if ((tile++) % NW != wave) continue;
                    d4_t y = wave_mfma_tile(G, A.nscd, 1, nsab, Mcd, nf, 1, nf, A.nscd, m0, n0, wl);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        int r = m0 + (wl >> 4) + 4 * q, f = n0 + (wl & 15);
                        if (r < nsab && f < nf) Y[r * nf + f] = y[q];
                    }
                }
        } else {
            for (int o = lane; o < nsab * nf; o += GSZ) {
                int r = o / nf, f = o - r * nf;
                double s = 0.0;
                for (int c = 0; c < A.nscd; c++) s += G[r * A.nscd + c] * Mcd[c * nf + f];
                Y[o] = s;
            }
        }
    }
    if (E0PRE > 0 && live) {
#pragma unroll
        for (int q = 0; q < E0PRE; q++) { const int c = lane + q * GSZ; if (c < n_e0p) E0p[c] = pre[q]; }
    }
    __syncthreads();
    double acc[3] = {0.0, 0.0, 0.0};
    if (live) {
        for (int var = 0; var < (has_m ? 2 : 1); var++) {
            const double *Mx = A.Mbuf + (var == 0 ? dp.m_off : m_off_m);
            const double *E0 = var == 0 ? E0p : E0m;
            const int ne = var == 0 ? A.ne_p : A.ne_m;
            const size_t xs = (size_t)nsab * ne;
            if (MFMA && GSZ >= 64 && mfma_worthwhile(nsab, ne, nf)) {
                int tile = 0;
                for (int m0 = 0; m0 < nsab; m0 += 16)
                    for (int n0 = 0; n0 < ne; n0 += 16) {
                        if ((tile++) % NW != wave) continue;
                        d4_t z = wave_mfma_tile(Y, nf, 1, nsab, E0, 1, nf, ne, nf, m0, n0, wl); // Z[r][e] = sum_f Y[r][f] E0[e][f]
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            int r = m0 + (wl >> 4) + 4 * q, e = n0 + (wl & 15);
                            if (r < nsab && e < ne) {
                                size_t o = (size_t)r * ne + e;
                                acc[0] += Mx[o] * z[q]; acc[1] += Mx[xs + o] * z[q]; acc[2] += Mx[2 * xs + o] * z[q];
                            }
                        }
                    }
            } else {
                for (int o = lane; o < nsab * ne; o += GSZ) {
                    int r = o / ne, e = o - r * ne;
                    double z = 0.0;
                    for (int f = 0; f < nf; f++) z += E0[e * nf + f] * Y[r * nf + f];
                    acc[0] += Mx[o] * z; acc[1] += Mx[xs + o] * z; acc[2] += Mx[2 * xs + o] * z;
                }
            }
        }
    }
    double w = DF ? A.w0 : 4.0;
    if (dp.sh_i == dp.sh_j) w *= 0.5;
    if (!DF && cd.sh_i == cd.sh_j) w *= 0.5;
    if (same_pair) w *= 0.5;
    for (int x = 0; x < 3; x++) {
        double v = acc[x];
        for (int o = (GSZ < 64 ? GSZ : 64) / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (live && (lane & ((GSZ < 64 ? GSZ : 64) - 1)) == 0) { // one partial sum per wave (per 16-lane group for GSZ = 16)
            double *gc = A.grad + (size_t)((blockIdx.x * QPW + grp) & (GRAD_COPIES - 1)) * A.natm3;
            atomicAdd(&gc[A.shell_atom[dp.sh_i] * 3 + x], w * v);
            // the skipped permutation (derivative on the first shell of the bra pair P) by invariance; density fitting:
            // the auxiliary centre takes minus the force on the differentiated orbital (or auxiliary) shell
            const int other = DF ? A.ket_atom[cd.sh_i] : A.shell_atom[A.inv_from_second ? dp.sh_j : cd.sh_i];
            atomicAdd(&gc[other * 3 + x], -w * v);
        }
    }
}

// =================================================================================================
// Derivative ERIs contracted with the two-particle density, MID / HIGH classes, "row" kernel (round 3).
//
// One wave per (live quartet, permutation), NOTHING handed over: lane e owns ROW e of the contracted [e0|f0] blocks of both
// derivative variants (rows 0 .. ne_p-1: the (l1+1, l2) block, rows ne_p .. ne_p+ne_m-1: the (l1-1, l2) block; ROWS rows per
// lane) and keeps its NF = ne(LC, LD) elements in REGISTERS; the other pair (LC, LD) is a template parameter, so the NF
// products of a row per primitive quartet and root are straight-line code over 3 (LC+LD+1) table values the lane reads from
// LDS once.  Roots, weights and the 2-D recurrence tables are computed by the wave exactly as in eri_rys_kernel (runtime l1,
// l2); the (l1-1) rows share them -- same Gaussian product, the factor K-/K+ = 1 / (2 alpha) is applied per primitive pair.
// The gradient is LINEAR in the block, so the row is contracted where it lives:
//     Yk[c]  = sum_f E0[e][f] Mcd[c][f]                       (ket HRR + cart->sph, matrix in LDS)
//     t_r    = sum_c Yk[c] G[r][c]                            (two-particle density block in LDS)
//     g[x]  += Mx[x][r][e] t_r                                (row e of the derivative x HRR x c2s matrices, adjoint use)
// and the wave sums g over its lanes.  Against the Rys -> hand-over -> eri_grad_contract pipeline this drops both hand-over
// buffers (written and re-read once per variant), two of the three launches and their task decodes, all barriers outside the
// Rys phases, and the Z product; every table value is read from LDS once per row instead of three per component.
// =================================================================================================
template <int LC, int LD>
struct FTab {
    static constexpr int NF = c_ne(LC, LD);
    int fx[NF], fy[NF], fz[NF];
    constexpr FTab() : fx{}, fy{}, fz{}
    {
        int n = 0;
        for (int f = LC; f <= LC + LD; f++)
            for (int a = f; a >= 0; a--)
                for (int b = f - a; b >= 0; b--) { fx[n] = a; fy[n] = b; fz[n] = f - a - b; n++; }
    }
};

struct GradRowsArgs {
    const PairRec *dplus, *dminus, *ket;
    const double *prim, *Mbuf;
    const TaskIdx *tasks;
    const int64_t *prefix;
    int nbra;
    int64_t t0, ntask;
    int swap, same_class;
    int nmax, nroots, tsz, PB;          // Rys set-up of the (l1+1, l2 | lc, ld) variant
    int ne_p, ne_m, nsab, ns2;
    const uint32_t *comp_p, *comp_m;    // component tables of the two variants (row bases: entry e * NF)
    RysDev rys;
    const double *D, *Dm;
    int ld;
    double hyb;
    const int *shell_atom;
    double *grad;
    int natm3, inv_from_second;
    const double *q_bra, *q_ket, *dmax;
    int nbas_d;
    double dtol;
};

// GSZ lanes per quartet (64 / GSZ quartets per wave): the Rys phases keep 2 n .. 3 n lanes per primitive quartet busy and the
// rows of the low bra classes fill a fraction of a wave, so the small classes run two or four latency chains side by side.
template <int LC, int LD, int ROWS, int GSZ>
__global__ __launch_bounds__(64) void eri_grad_rows_kernel(GradRowsArgs A)
{
    constexpr int NF = c_ne(LC, LD), MMAX = LC + LD, M1 = MMAX + 1;
    constexpr int NSC = 2 * LC + 1, NSD = 2 * LD + 1, NSCD = NSC * NSD;
    constexpr int QPW = 64 / GSZ;
    constexpr FTab<LC, LD> FT{};
    extern __shared__ double lds_all[];
    const int grp = QPW == 1 ? 0 : threadIdx.x / GSZ, lane = QPW == 1 ? threadIdx.x : threadIdx.x % GSZ;
    const int64_t tl = (int64_t)blockIdx.x * QPW + grp;
    bool live = tl < A.ntask;
    int ib, ik;
    get_task(A.tasks, A.prefix, A.nbra, A.t0 + (live ? tl : 0), ib, ik);
    const bool same_pair = A.same_class && ib == ik;
    if (A.swap) { int t_ = ib; ib = ik; ik = t_; }
    const bool has_m = A.ne_m > 0;
    const PairRec dp = A.dplus[ib], cd = A.ket[ik];
    if (live && A.dmax && A.q_bra[ib] * A.q_ket[ik] * quartet_density_bound(A.dmax, A.nbas_d, dp.sh_i, dp.sh_j, cd.sh_i, cd.sh_j, A.hyb) < A.dtol)
        live = false;
    if (QPW == 1 && !live) return;
    int m_prim = dp.prim_off, m_off_m = 0;
    if (has_m) { const PairRec dm = A.dminus[ib]; m_prim = dm.prim_off; m_off_m = dm.m_off; }
    const int n = A.nroots, tsz = A.tsz, PB = A.PB, nsab = A.nsab;
    const size_t lds_per = (size_t)PB * n * 3 * tsz + (size_t)PB * 2 * n + PB + NSCD * NF + (size_t)nsab * NSCD;
    double *T0 = lds_all + (size_t)grp * lds_per;          // [PB n][3][tsz]
    double *rw = T0 + (size_t)PB * n * 3 * tsz;            // [PB][2 n]
    double *rho = rw + (size_t)PB * 2 * n;                 // [PB]   K- / K+ of the bra primitive pair
    double *McdL = rho + PB;                               // [NSCD][NF]
    double *G = McdL + NSCD * NF;                          // [nsab][NSCD]
    // ket matrix and two-particle density block (used after the Rys loop: its barriers order these writes)
    if (live) {
        const double *Mcd = A.Mbuf + cd.m_off;
        for (int q = lane; q < NSCD * NF; q += GSZ) McdL[q] = Mcd[q];
        const double *D = A.D;
        const int ld = A.ld, ns2 = A.ns2;
        for (int o = lane; o < nsab * NSCD; o += GSZ) {      // G[r][c], r = (sa, sb) of the differentiated pair, c = (sc, sd) of the other
            const int r = o / NSCD, c = o - r * NSCD;
            const int sa = r / ns2, sb = r - sa * ns2, sc = c / NSD, sd = c - sc * NSD;
            const int i = dp.ao_i + sa, j = dp.ao_j + sb, k = cd.ao_i + sc, l = cd.ao_j + sd;
            double ex = D[(size_t)i * ld + k] * D[(size_t)j * ld + l] + D[(size_t)i * ld + l] * D[(size_t)j * ld + k];
            if (A.Dm) // sum_s Ds x Ds = (D x D + M x M) / 2
                ex += A.Dm[(size_t)i * ld + k] * A.Dm[(size_t)j * ld + l] + A.Dm[(size_t)i * ld + l] * A.Dm[(size_t)j * ld + k];
            G[o] = D[(size_t)i * ld + j] * D[(size_t)k * ld + l] - 0.25 * A.hyb * ex;
        }
    }
    const int ne = A.ne_p + A.ne_m;
    int bx[ROWS], by[ROWS], bz[ROWS];
    bool valid[ROWS], minus[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        const int e = lane + GSZ * rr;
        valid[rr] = live && e < ne;
        minus[rr] = e >= A.ne_p;
        uint32_t w = 0;
        if (valid[rr]) w = minus[rr] ? A.comp_m[(size_t)(e - A.ne_p) * NF] : A.comp_p[(size_t)e * NF];
        bx[rr] = (int)(w & 1023u) - LC; by[rr] = (int)((w >> 10) & 1023u); bz[rr] = (int)((w >> 20) & 1023u);
    }
    double acc[ROWS][NF];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++)
#pragma unroll
        for (int f = 0; f < NF; f++) acc[rr][f] = 0.0;
    const int ncd = cd.nprim, nPQ = live ? dp.nprim * ncd : 0;
    int nPQ_all = nPQ;   // uniform trip count over the quartets sharing this wave (the barriers sit in the loop)
    if (QPW > 1)
        for (int o = GSZ; o < 64; o <<= 1) nPQ_all = max(nPQ_all, __shfl_xor(nPQ_all, o));
    const double *prim_b = A.prim + (size_t)dp.prim_off * 8, *prim_k = A.prim + (size_t)cd.prim_off * 8;
    const double *prim_m = A.prim + (size_t)m_prim * 8;
    for (int pq0 = 0; pq0 < nPQ_all; pq0 += PB) {
        const int npq = max(0, min(PB, nPQ - pq0));
        for (int idx = lane; idx < npq * 2 * n; idx += GSZ) {   // roots and weights
            const int pql = idx / (2 * n), f = idx - pql * 2 * n;
            const int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
            const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
            const double p = b[0], q = k[0];
            const double dx = b[1] - k[1], dy = b[2] - k[2], dz = b[3] - k[3];
            rw[pql * 2 * n + f] = rys_eval(A.rys, n, f, p * q / (p + q) * (dx * dx + dy * dy + dz * dz));
        }
        if (has_m)
            for (int idx = lane; idx < npq; idx += GSZ) {       // K- / K+ of the primitive pairs of this batch
                const int ip = (pq0 + idx) / ncd;
                rho[idx] = prim_m[(size_t)ip * 8 + 7] / prim_b[(size_t)ip * 8 + 7];
            }
        __syncthreads();
        for (int idx = lane; idx < npq * n * 3; idx += GSZ) {    // 2-D recurrence tables
            const int pql = idx / (3 * n), rem = idx - pql * 3 * n, r = rem / 3, d = rem - r * 3;
            const int pq = pq0 + pql, ip = pq / ncd, jp = pq - ip * ncd;
            const double *b = prim_b + (size_t)ip * 8, *k = prim_k + (size_t)jp * 8;
            const double p = b[0], q = k[0], pq1 = 1.0 / (p + q);
            const double u = rw[pql * 2 * n + r];
            const double PQd = b[1 + d] - k[1 + d];
            const double b00 = 0.5 * u * pq1, b10 = 0.5 / p * (1.0 - u * q * pq1), b01 = 0.5 / q * (1.0 - u * p * pq1);
            const double c00 = b[4 + d] - u * q * pq1 * PQd, c01 = k[4 + d] + u * p * pq1 * PQd;
            double *T = T0 + ((size_t)(pql * n + r) * 3 + d) * tsz;
            double t00 = 1.0;
            if (d == 2) t00 = rw[pql * 2 * n + n + r] * b[7] * k[7] * 34.986836655249725 /* 2 pi^2.5 */ * pq1 * sqrt(p + q) / (p * q);
            T[0] = t00;
            double tm = 0.0, tc = t00;
            for (int i = 0; i < A.nmax; i++) {
                const double tn = c00 * tc + i * b10 * tm;
                T[(i + 1) * M1] = tn;
                tm = tc; tc = tn;
            }
            for (int m = 0; m < MMAX; m++)
                for (int i = 0; i <= A.nmax; i++) {
                    double v = c01 * T[i * M1 + m];
                    if (m > 0) v += m * b01 * T[i * M1 + m - 1];
                    if (i > 0) v += i * b00 * T[(i - 1) * M1 + m];
                    T[i * M1 + m + 1] = v;
                }
        }
        __syncthreads();
        const int nslot = npq * n;
        for (int s_ = 0; s_ < nslot; s_++) {
            const double *Tx = T0 + (size_t)s_ * 3 * tsz, *Ty = Tx + tsz, *Tz = Ty + tsz;
            const double rs = has_m ? rho[s_ / n] : 0.0;
#pragma unroll
            for (int rr = 0; rr < ROWS; rr++) {
                if (!valid[rr]) continue;
                double x[M1], y[M1], z[M1];
                const double sc = minus[rr] ? rs : 1.0;
#pragma unroll
                for (int m = 0; m < M1; m++) { x[m] = Tx[bx[rr] + m]; y[m] = Ty[by[rr] + m]; z[m] = Tz[bz[rr] + m] * sc; }
#pragma unroll
                for (int f = 0; f < NF; f++) acc[rr][f] = fma(x[FT.fx[f]] * y[FT.fy[f]], z[FT.fz[f]], acc[rr][f]);
            }
        }
        __syncthreads();
    }
    double g[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        if (!valid[rr]) continue;
        double Yk[NSCD];
#pragma unroll
        for (int c = 0; c < NSCD; c++) {
            double s = 0.0;
#pragma unroll
            for (int f = 0; f < NF; f++) s = fma(acc[rr][f], McdL[c * NF + f], s);
            Yk[c] = s;
        }
        const int e = lane + GSZ * rr;
        const int nev = minus[rr] ? A.ne_m : A.ne_p, el = minus[rr] ? e - A.ne_p : e;
        const double *Mx = A.Mbuf + (minus[rr] ? m_off_m : dp.m_off) + el;
        const size_t xs = (size_t)nsab * nev;
        // the derivative-matrix elements of four r go out together (they depend on nothing computed here): one round trip per
        // four rows of G instead of one per row
        for (int r0 = 0; r0 < nsab; r0 += 4) {
            double mx[4][3];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const size_t o = (size_t)min(r0 + u, nsab - 1) * nev;
                mx[u][0] = Mx[o]; mx[u][1] = Mx[xs + o]; mx[u][2] = Mx[2 * xs + o];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (r0 + u >= nsab) break;
                double t = 0.0;
#pragma unroll
                for (int c = 0; c < NSCD; c++) t = fma(Yk[c], G[(r0 + u) * NSCD + c], t);
                g[0] = fma(mx[u][0], t, g[0]); g[1] = fma(mx[u][1], t, g[1]); g[2] = fma(mx[u][2], t, g[2]);
            }
        }
    }
    double w = 4.0;
    if (dp.sh_i == dp.sh_j) w *= 0.5;
    if (cd.sh_i == cd.sh_j) w *= 0.5;
    if (same_pair) w *= 0.5;
#pragma unroll
    for (int x = 0; x < 3; x++) {
        double v = g[x];
        for (int o = GSZ / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (live && lane == 0) {
            double *gc = A.grad + (size_t)((blockIdx.x * QPW + grp) & (GRAD_COPIES - 1)) * A.natm3;
            atomicAdd(&gc[A.shell_atom[dp.sh_i] * 3 + x], w * v);
            atomicAdd(&gc[A.shell_atom[A.inv_from_second ? dp.sh_j : cd.sh_i] * 3 + x], -w * v);
        }
    }
}

// rows per lane and lanes per quartet by the number of rows ne of the two derivative blocks:
//   ne <= 64: (2, 32), two quartets per wave   ne <= 128: (2, 64)   ne <= 192: (3, 64)
// Measured on one box (ibuprofen/def2-TZVP, grad_dtol 1e-10, whole two-electron gradient): pipeline only 0.535 s; row kernel for
// ne > 64 only 0.504; for ne >= 33 0.495 (one quartet per wave there: 0.502); for ne >= 20 0.492; four quartets per wave
// (16 lanes) for ne <= 32: 0.518 -- the small blocks stay on the hand-over pipeline / thread-per-quartet kernels.
struct RowsGeom { int rows, gsz; };
static RowsGeom rows_geometry(int ne)
{
    if (g_rows_g32 && ne <= 64) return {2, 32};
    if (ne <= 64) return {1, 64};
    if (ne <= 128) return {2, 64};
    if (ne <= 192) return {3, 64};
    return {0, 0};
}
template <int LC, int LD, int ROWS, int GSZ>
static int launch_grad_rows_g(const GradRowsArgs &R, size_t shm_per_quartet, hipStream_t st)
{
    constexpr int QPW = 64 / GSZ;
    const size_t shm = shm_per_quartet * QPW;
    if (shm > 160 * 1024) return 0;
    if (shm > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void *)eri_grad_rows_kernel<LC, LD, ROWS, GSZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((eri_grad_rows_kernel<LC, LD, ROWS, GSZ>), dim3((unsigned)((R.ntask + QPW - 1) / QPW)), dim3(64), shm, st, R);
    HIPCHK(hipGetLastError());
    return 1;
}
template <int LC, int LD>
static int launch_grad_rows_t(RowsGeom gm, const GradRowsArgs &R, size_t shm_q, hipStream_t st)
{
    if (gm.rows == 2 && gm.gsz == 32) return launch_grad_rows_g<LC, LD, 2, 32>(R, shm_q, st);
    if (gm.rows == 1 && gm.gsz == 64) return launch_grad_rows_g<LC, LD, 1, 64>(R, shm_q, st);
    if (gm.rows == 2 && gm.gsz == 64) return launch_grad_rows_g<LC, LD, 2, 64>(R, shm_q, st);
    if (gm.rows == 3 && gm.gsz == 64) return launch_grad_rows_g<LC, LD, 3, 64>(R, shm_q, st);
    return 0;
}
// 1: launched, 0: no row kernel for this (other pair, number of rows), -1: error.  dry: only answers the question.
static int launch_grad_rows(int lc, int ld, int ne, const GradRowsArgs &R, size_t shm_q, hipStream_t st, bool dry = false)
{
    const RowsGeom gm = rows_geometry(ne);
    if (gm.rows == 0) return 0;
#define ROWS_CASE(a, b) if (lc == a && ld == b) return dry ? 1 : launch_grad_rows_t<a, b>(gm, R, shm_q, st)
    ROWS_CASE(0, 0); ROWS_CASE(1, 0); ROWS_CASE(1, 1); ROWS_CASE(2, 0); ROWS_CASE(2, 1); ROWS_CASE(2, 2); ROWS_CASE(3, 0); ROWS_CASE(3, 1);
#undef ROWS_CASE
    return 0;
}

// =================================================================================================
// Derivative ERIs contracted with the two-particle density, LOW angular classes: one THREAD per (canonical quartet, role
// permutation), everything in registers (the gradient counterpart of eri_tpq_kernel; round 1 spent 1.6 of the 1.9 s of the
// ibuprofen/def2-TZVP gradient in the wave-per-quartet path of these classes).
//   d/dA_x (a b| = 2 alpha (a+1_x b| - a_x (a-1_x b|: the primitive loop accumulates the [e0|f0] blocks of the (L1+1) shell
//   (coefficients 2 alpha c, folded into the "plus" pair record) and of the (L1-1) shell side by side (same roots, same 2-D
//   recurrence tables); the ket is transformed first, then for each ket component the bra goes HRR -> derivative -> cart->sph
//   and is contracted at once with G = D_ab D_cd - hyb/4 (D_ac D_bd + D_ad D_bc) (+ spin term): no derivative block is stored.
// L1 = differentiated shell, L2 = its partner (either order), (LC >= LD) the other pair.
// =================================================================================================
struct TpqGradArgs {
    const PairRec *dplus, *dminus, *ket;
    const double *prim;
    const int64_t *prefix;
    int nbra;
    int64_t t0, ntask;
    int swap, same_class;
    const double *c2s;
    int c2s_off[LMAX + 2];
    RysDev rys;
    const double *shell_xyz;
    const double *D, *Dm;     // padded total / spin density
    int ld;
    double hyb;
    const int *shell_atom;
    double *grad;             // GRAD_COPIES private copies of [natm3]
    int natm3, inv_from_second;
    const double *q_bra, *q_ket, *dmax;
    int nbas_d;
    double dtol;
    const TaskIdx *tasks;     // non-null: the compacted list of live quartets (already screened: dmax is null then)
};

template <int L1, int L2, int LC, int LD>
__global__ __launch_bounds__(TPQ_BLOCK) void eri_tpq_grad_kernel(TpqGradArgs A)
{
    constexpr bool HASM = L1 > 0;
    constexpr int LP = L1 + 1, LM = HASM ? L1 - 1 : 0;
    constexpr int NR = (LP + L2 + LC + LD) / 2 + 1;
    constexpr int NMAX = LP + L2, MMAX = LC + LD;
    constexpr int NEP = c_ne(LP, L2), NEM = HASM ? c_ne(LM, L2) : 0, NF = c_ne(LC, LD);
    constexpr int NS1 = 2 * L1 + 1, NS2 = 2 * L2 + 1, NSC = 2 * LC + 1, NSD = 2 * LD + 1, NSCD = NSC * NSD;
    constexpr int NK1 = c_ncart(L1), NK2 = c_ncart(L2), NCP = c_ncart(LP), NCM = HASM ? c_ncart(LM) : 1;
    extern __shared__ double cheb[];
    const int nint = A.rys.nint[NR];
    {
        const int ntab = nint * 2 * NR * (RYS_DEG + 1);
        const double *src = A.rys.cheb + A.rys.off[NR];
        for (int q = threadIdx.x; q < ntab; q += TPQ_BLOCK) cheb[q] = src[q];
    }
    __syncthreads();
    const int64_t tl = (int64_t)blockIdx.x * TPQ_BLOCK + threadIdx.x;
    bool live = tl < A.ntask;
    int ib = 0, ik = 0;
    if (live) get_task(A.tasks, A.prefix, A.nbra, A.t0 + tl, ib, ik);
    const bool same_pair = A.same_class && ib == ik;
    if (A.swap) { int t_ = ib; ib = ik; ik = t_; }
    const PairRec dp = A.dplus[ib], cd = A.ket[ik];
    if (live && A.dmax && A.q_bra[ib] * A.q_ket[ik] * quartet_density_bound(A.dmax, A.nbas_d, dp.sh_i, dp.sh_j, cd.sh_i, cd.sh_j, A.hyb) < A.dtol)
        live = false;
    double force[3] = {0.0, 0.0, 0.0};
    if (live) {
        int mprim_off = 0;
        if (HASM) mprim_off = A.dminus[ib].prim_off;
        double accp[NEP * NF], accm[(HASM ? NEM : 1) * NF];
#pragma unroll
        for (int q = 0; q < NEP * NF; q++) accp[q] = 0.0;
#pragma unroll
        for (int q = 0; q < (HASM ? NEM : 1) * NF; q++) accm[q] = 0.0;
        for (int ip = 0; ip < dp.nprim; ip++) {
            const double *b = A.prim + (size_t)(dp.prim_off + ip) * 8;
            const double p = b[0], Px = b[1], Py = b[2], Pz = b[3], Kp = b[7];
            const double Km = HASM ? A.prim[(size_t)(mprim_off + ip) * 8 + 7] : 0.0;
            const double PA[3] = {b[4], b[5], b[6]};
            for (int jp = 0; jp < cd.nprim; jp++) {
                const double *kk = A.prim + (size_t)(cd.prim_off + jp) * 8;
                const double q = kk[0];
                const double PQ[3] = {Px - kk[1], Py - kk[2], Pz - kk[3]};
                const double QC[3] = {kk[4], kk[5], kk[6]};
                const double pq1 = 1.0 / (p + q);
                const double x = p * q * pq1 * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]);
                const double pref = kk[7] * 34.986836655249725 /* 2 pi^2.5 */ * pq1 * sqrt(p + q) / (p * q);
                double u[NR], w[NR];
                if (x < nint * RYS_H) {
                    int iv = (int)(x * (1.0 / RYS_H));
                    if (iv >= nint) iv = nint - 1;
                    const double sx = (x - (iv * RYS_H + 0.5 * RYS_H)) * (2.0 / RYS_H), s2 = 2.0 * sx;
                    const double *cb = cheb + (size_t)iv * 2 * NR * (RYS_DEG + 1);
#pragma unroll
                    for (int f = 0; f < 2 * NR; f++) {
                        const double *cc = cb + f * (RYS_DEG + 1);
                        double b1 = 0.0, b2 = 0.0;
#pragma unroll
                        for (int kq = RYS_DEG; kq >= 1; kq--) {
                            double tq = s2 * b1 - b2 + cc[kq];
                            b2 = b1;
                            b1 = tq;
                        }
                        const double val = sx * b1 - b2 + cc[0];
                        if (f < NR) u[f] = val; else w[f - NR] = val;
                    }
                } else {
                    const double rx = 1.0 / x, rsx = rsqrt(x);
#pragma unroll
                    for (int f = 0; f < NR; f++) { u[f] = A.rys.herm_r[NR * RYS_NMAX + f] * rx; w[f] = A.rys.herm_w[NR * RYS_NMAX + f] * rsx; }
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    const double ur = u[r];
                    const double b00 = 0.5 * ur * pq1, b10 = 0.5 / p * (1.0 - ur * q * pq1), b01 = 0.5 / q * (1.0 - ur * p * pq1);
                    double T[3][NMAX + 1][MMAX + 1];
#pragma unroll
                    for (int d = 0; d < 3; d++) {
                        const double c00 = PA[d] - ur * q * pq1 * PQ[d], c01 = QC[d] + ur * p * pq1 * PQ[d];
                        T[d][0][0] = d == 2 ? w[r] * pref : 1.0;
#pragma unroll
                        for (int n = 0; n < NMAX; n++) T[d][n + 1][0] = c00 * T[d][n][0] + (n > 0 ? n * b10 * T[d][n - 1][0] : 0.0);
#pragma unroll
                        for (int m = 0; m < MMAX; m++)
#pragma unroll
                            for (int n = 0; n <= NMAX; n++) {
                                double v = c01 * T[d][n][m];
                                if (m > 0) v = fma(m * b01, T[d][n][m - 1], v);
                                if (n > 0) v = fma(n * b00, T[d][n - 1][m], v);
                                T[d][n][m + 1] = v;
                            }
                    }
#pragma unroll
                    for (int df = LC; df <= LC + LD; df++)
#pragma unroll
                        for (int fx = df; fx >= 0; fx--)
#pragma unroll
                            for (int fy = df - fx; fy >= 0; fy--) {
                                const int fz = df - fx - fy, jf = c_eoff(LC, df) + c_cidx(df, fx, fy);
                                // plus block: degrees LP .. LP + L2
#pragma unroll
                                for (int de = LP; de <= LP + L2; de++)
#pragma unroll
                                    for (int ex = de; ex >= 0; ex--)
#pragma unroll
                                        for (int ey = de - ex; ey >= 0; ey--) {
                                            const int ez = de - ex - ey, ie = c_eoff(LP, de) + c_cidx(de, ex, ey);
                                            accp[ie * NF + jf] = fma(Kp * T[0][ex][fx] * T[1][ey][fy], T[2][ez][fz], accp[ie * NF + jf]);
                                        }
                                if (HASM) {
#pragma unroll
                                    for (int de = LM; de <= LM + L2; de++)
#pragma unroll
                                        for (int ex = de; ex >= 0; ex--)
#pragma unroll
                                            for (int ey = de - ex; ey >= 0; ey--) {
                                                const int ez = de - ex - ey, ie = c_eoff(LM, de) + c_cidx(de, ex, ey);
                                                accm[ie * NF + jf] = fma(Km * T[0][ex][fx] * T[1][ey][fy], T[2][ez][fz], accm[ie * NF + jf]);
                                            }
                                }
                            }
                }
            }
        }
        // ---- ket: HRR + cart->sph on the f index of both blocks
        const MI_CONST_AS double *c2s = as_const(A.c2s);
        double ABv[3], CDv[3];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            ABv[d] = A.shell_xyz[3 * dp.sh_i + d] - A.shell_xyz[3 * dp.sh_j + d];
            CDv[d] = A.shell_xyz[3 * cd.sh_i + d] - A.shell_xyz[3 * cd.sh_j + d];
        }
        double Ep[NEP * NSCD], Em[(HASM ? NEM : 1) * NSCD];
        tpq_pair_transform<LC, LD, NEP, 1, NF, 1, NSCD>(accp, Ep, CDv, c2s + A.c2s_off[LC], c2s + A.c2s_off[LD]);
        if (HASM) tpq_pair_transform<LC, LD, NEM, 1, NF, 1, NSCD>(accm, Em, CDv, c2s + A.c2s_off[LC], c2s + A.c2s_off[LD]);
        // ---- bra per ket component: HRR (closed form) -> derivative -> cart->sph -> contraction with G
        double pw[3][L2 + 1];
#pragma unroll
        for (int d = 0; d < 3; d++) {
            pw[d][0] = 1.0;
#pragma unroll
            for (int q = 1; q <= L2; q++) pw[d][q] = pw[d][q - 1] * ABv[d];
        }
        const MI_CONST_AS double *c1 = c2s + A.c2s_off[L1], *c2 = c2s + A.c2s_off[L2];
        const double *D = A.D;
        const int ld = A.ld;
#pragma unroll
        for (int col = 0; col < NSCD; col++) {
            const int sc = col / NSD, sd = col - sc * NSD;
            // (a' b| for the plus (a' of degree LP) and minus (degree LM) shells
            double gp[NCP * NK2], gm[NCM * NK2];
#pragma unroll
            for (int pass = 0; pass < (HASM ? 2 : 1); pass++) {
                const int LX = pass == 0 ? LP : LM;
#pragma unroll
                for (int ax = LX; ax >= 0; ax--)
#pragma unroll
                    for (int ay = LX - ax; ay >= 0; ay--) {
                        const int ia = c_cidx(LX, ax, ay);
#pragma unroll
                        for (int bx = L2; bx >= 0; bx--)
#pragma unroll
                            for (int by = L2 - bx; by >= 0; by--) {
                                const int bz = L2 - bx - by, ibb = c_cidx(L2, bx, by);
                                double v = 0.0;
#pragma unroll
                                for (int ix = 0; ix <= bx; ix++)
#pragma unroll
                                    for (int iy = 0; iy <= by; iy++)
#pragma unroll
                                        for (int iz = 0; iz <= bz; iz++) {
                                            const int deg = LX + ix + iy + iz;
                                            const int e = c_eoff(LX, deg) + c_cidx(deg, ax + ix, ay + iy);
                                            const double cf = (double)(c_binom(bx, ix) * c_binom(by, iy) * c_binom(bz, iz));
                                            const double src = pass == 0 ? Ep[e * NSCD + col] : Em[(HASM ? e : 0) * NSCD + col];
                                            v = fma(cf * pw[0][bx - ix] * pw[1][by - iy] * pw[2][bz - iz], src, v);
                                        }
                                if (pass == 0) gp[ia * NK2 + ibb] = v; else gm[ia * NK2 + ibb] = v;
                            }
                    }
            }
            // density factor of this ket component for every bra component
            const int k = cd.ao_i + sc, l = cd.ao_j + sd;
            double Gd[NS1 * NS2];
#pragma unroll
            for (int sa = 0; sa < NS1; sa++)
#pragma unroll
                for (int sb = 0; sb < NS2; sb++) {
                    const int i = dp.ao_i + sa, j = dp.ao_j + sb;
                    double ex = D[(size_t)i * ld + k] * D[(size_t)j * ld + l] + D[(size_t)i * ld + l] * D[(size_t)j * ld + k];
                    if (A.Dm) ex += A.Dm[(size_t)i * ld + k] * A.Dm[(size_t)j * ld + l] + A.Dm[(size_t)i * ld + l] * A.Dm[(size_t)j * ld + k];
                    Gd[sa * NS2 + sb] = D[(size_t)i * ld + j] * D[(size_t)k * ld + l] - 0.25 * A.hyb * ex;
                }
            // Gc[a][b] = sum_{sa,sb} c1[a][sa] c2[b][sb] Gd[sa][sb]   (density back-transformed to cartesians: 3x fewer products)
            double h2[NS1 * NK2];
#pragma unroll
            for (int sa = 0; sa < NS1; sa++)
#pragma unroll
                for (int bq = 0; bq < NK2; bq++) {
                    double v = 0.0;
#pragma unroll
                    for (int sb = 0; sb < NS2; sb++) v = fma(c2[bq * NS2 + sb], Gd[sa * NS2 + sb], v);
                    h2[sa * NK2 + bq] = v;
                }
            double Gc[NK1 * NK2];
#pragma unroll
            for (int a = 0; a < NK1; a++)
#pragma unroll
                for (int bq = 0; bq < NK2; bq++) {
                    double v = 0.0;
#pragma unroll
                    for (int sa = 0; sa < NS1; sa++) v = fma(c1[a * NS1 + sa], h2[sa * NK2 + bq], v);
                    Gc[a * NK2 + bq] = v;
                }
            // derivative: d_x (a b| = (a+1_x b|_plus - a_x (a-1_x b|_minus
#pragma unroll
            for (int ax = L1; ax >= 0; ax--)
#pragma unroll
                for (int ay = L1 - ax; ay >= 0; ay--) {
                    const int az = L1 - ax - ay, ia = c_cidx(L1, ax, ay);
                    const int apow[3] = {ax, ay, az};
#pragma unroll
                    for (int xdir = 0; xdir < 3; xdir++) {
                        const int px = ax + (xdir == 0), py = ay + (xdir == 1);
                        const int ipl = c_cidx(LP, px, py);
#pragma unroll
                        for (int bq = 0; bq < NK2; bq++) {
                            double dv = gp[ipl * NK2 + bq];
                            if (HASM && apow[xdir] > 0) {
                                const int mx = ax - (xdir == 0), my = ay - (xdir == 1);
                                dv -= apow[xdir] * gm[c_cidx(LM, mx, my) * NK2 + bq];
                            }
                            force[xdir] = fma(dv, Gc[ia * NK2 + bq], force[xdir]);
                        }
                    }
                }
        }
        double wq = 4.0;
        if (dp.sh_i == dp.sh_j) wq *= 0.5;
        if (cd.sh_i == cd.sh_j) wq *= 0.5;
        if (same_pair) wq *= 0.5;
#pragma unroll
        for (int xdir = 0; xdir < 3; xdir++) force[xdir] *= wq;
    }
    // ---- forces: +f on the atom of the differentiated shell, -f on the atom of the skipped permutation.  Lanes of a wave
    // mostly share both atoms (consecutive tasks share the bra pair): reduce over the wave first when they all do.
    const int a1 = A.shell_atom[dp.sh_i], a2 = A.shell_atom[A.inv_from_second ? dp.sh_j : cd.sh_i];
    const int lane = threadIdx.x & 63;
    const bool uniform = __all(a1 == __shfl(a1, 0) && a2 == __shfl(a2, 0));
    double *gc = A.grad + (size_t)(blockIdx.x & (GRAD_COPIES - 1)) * A.natm3;
    if (uniform) {
#pragma unroll
        for (int xdir = 0; xdir < 3; xdir++) {
            double v = force[xdir];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if (lane == 0 && v != 0.0) { atomicAdd(&gc[a1 * 3 + xdir], v); atomicAdd(&gc[a2 * 3 + xdir], -v); }
        }
    } else if (live) {
#pragma unroll
        for (int xdir = 0; xdir < 3; xdir++) { atomicAdd(&gc[a1 * 3 + xdir], force[xdir]); atomicAdd(&gc[a2 * 3 + xdir], -force[xdir]); }
    }
}

template <int L1, int L2, int LC, int LD>
static int launch_eri_tpq_grad_t(const TpqGradArgs &Q, hipStream_t st)
{
    constexpr int NR = (L1 + 1 + L2 + LC + LD) / 2 + 1;
    const size_t shm = sizeof(double) * (size_t)RYS_NINT_H[NR] * 2 * NR * (RYS_DEG + 1);
    if (Q.ntask <= 0) return 1;   // nothing in this rank's share of the class (the class is still "handled")
    hipLaunchKernelGGL((eri_tpq_grad_kernel<L1, L2, LC, LD>), dim3((unsigned)((Q.ntask + TPQ_BLOCK - 1) / TPQ_BLOCK)), dim3(TPQ_BLOCK), shm, st, Q);
    if (hipGetLastError() != hipSuccess) return fail("eri_tpq_grad_kernel launch failed");
    return 1;
}

// returns 1 if the class (l1 = differentiated shell, l2 = partner | lc >= ld) has a thread-per-quartet kernel, 0 otherwise
// (dry = true: only answers the question)
static int launch_eri_tpq_grad(int l1, int l2, int lc, int ld, const TpqGradArgs &Q, hipStream_t st, bool dry = false)
{
    if (l1 > 3 || l2 > 3 || lc > 3 || ld > 3) return 0;
    const int key = ((l1 * 4 + l2) * 4 + lc) * 4 + ld;
#define TPQG_CASE(a, b, c_, d) case (((a) * 4 + (b)) * 4 + (c_)) * 4 + (d): return dry ? 1 : launch_eri_tpq_grad_t<a, b, c_, d>(Q, st)
    switch (key) {
        // other pair (ss)
        TPQG_CASE(0, 0, 0, 0); TPQG_CASE(1, 0, 0, 0); TPQG_CASE(0, 1, 0, 0); TPQG_CASE(1, 1, 0, 0); TPQG_CASE(2, 0, 0, 0);
        TPQG_CASE(2, 1, 0, 0); TPQG_CASE(1, 2, 0, 0); TPQG_CASE(3, 0, 0, 0); TPQG_CASE(2, 2, 0, 0);
        TPQG_CASE(3, 1, 0, 0);   // (sd|ss), (sf|ss), (pf|ss): the compiler leaves their index arrays in scratch -> wave-per-quartet path
        // other pair (ps)
        TPQG_CASE(0, 0, 1, 0); TPQG_CASE(1, 0, 1, 0); TPQG_CASE(0, 1, 1, 0); TPQG_CASE(1, 1, 1, 0); TPQG_CASE(2, 0, 1, 0); TPQG_CASE(0, 2, 1, 0);
        TPQG_CASE(3, 0, 1, 0);
        // other pair (pp), (ds), (fs), (dp): (NEP + NEM) * NF accumulators <= ~64
        TPQG_CASE(0, 0, 1, 1); TPQG_CASE(1, 0, 1, 1);
        TPQG_CASE(0, 0, 2, 0); TPQG_CASE(1, 0, 2, 0); TPQG_CASE(0, 1, 2, 0);
        TPQG_CASE(0, 0, 3, 0);
        TPQG_CASE(0, 0, 2, 1);
        // round 3: candidates with up to ~150 accumulators (one wave per SIMD); kept only where the compiler needs no scratch
        TPQG_CASE(0, 1, 1, 1); TPQG_CASE(1, 0, 3, 0); TPQG_CASE(0, 1, 3, 0); TPQG_CASE(1, 1, 2, 0);
        TPQG_CASE(1, 0, 2, 1); TPQG_CASE(0, 1, 2, 1); TPQG_CASE(2, 0, 2, 0); TPQG_CASE(1, 2, 1, 0); TPQG_CASE(0, 2, 2, 0);
        TPQG_CASE(2, 0, 1, 1); TPQG_CASE(0, 0, 3, 1); TPQG_CASE(0, 0, 2, 2);
        TPQG_CASE(2, 1, 1, 0);
    default: return 0;
    }
#undef TPQG_CASE
}

static bool tpq_grad_has_class(int l1, int l2, int lc, int ld)
{
    TpqGradArgs none{};
    return launch_eri_tpq_grad(l1, l2, lc, ld, none, nullptr, true) == 1;
}

// max |D| over the AO block of every shell pair (density-weighted screening of the derivative quartets)
__global__ void shell_dmax_kernel(const double *D, const double *Dm, int ld, const int *sh_ao, const int *sh_n, int nbas, double *out)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nbas * nbas) return;
    int a = idx / nbas, b = idx - a * nbas;
    double m = 0.0;
    for (int i = 0; i < sh_n[a]; i++)
        for (int j = 0; j < sh_n[b]; j++) {
            size_t o = (size_t)(sh_ao[a] + i) * ld + sh_ao[b] + j;
            // open shell: |D| + |M| bounds both spin densities (the exchange bound then covers D x D + M x M)
            m = fmax(m, fabs(D[o]) + (Dm ? fabs(Dm[o]) : 0.0));
        }
    out[idx] = m;
}

__global__ void grad_reduce_copies_kernel(const double *copies, int natm3, double *grad)
{
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= natm3) return;
    double s = 0.0;
    for (int c = 0; c < GRAD_COPIES; c++) s += copies[(size_t)c * natm3 + idx];
    grad[idx] += s;
}

extern "C" int mi_grad_eri_sharded(mi_ctx *c, const double *d_D, const double *d_Dspin, double hyb, double *d_grad, int rank, int nranks,
                                   void *stream);

extern "C" int mi_grad_eri(mi_ctx *c, const double *d_D, double hyb, double *d_grad, void *stream)
{
    return c ? mi_grad_eri_sharded(c, d_D, nullptr, hyb, d_grad, c->rank, c->nranks, stream) : fail("mi_grad_eri: null argument");
}

extern "C" int mi_grad_eri_spin(mi_ctx *c, const double *d_D, const double *d_Dspin, double hyb, double *d_grad, void *stream)
{
    return c ? mi_grad_eri_sharded(c, d_D, d_Dspin, hyb, d_grad, c->rank, c->nranks, stream) : fail("mi_grad_eri: null argument");
}

// (rank, nranks): which share of the derivative-quartet batches this call evaluates.  It is an ARGUMENT, not the split of
// the last mi_eri_prepare: in direct mode the tile store is prepared group by group (rank*ng + v of nranks*ng) while the
// gradient is still shared between the nranks processes only.
extern "C" int mi_grad_eri_sharded(mi_ctx *c, const double *d_D, const double *d_Dspin, double hyb, double *d_grad, int rank, int nranks,
                                   void *stream)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("mi_grad_eri: bad rank/nranks");
    if (!c || !d_D || !d_grad) return fail("mi_grad_eri: null argument");
    if (!c->eri_ready) return fail("mi_grad_eri: call mi_eri_prepare first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    auto tg0 = std::chrono::steady_clock::now();
    if (prepare_grad_records(c)) return -1;
    auto tg1 = std::chrono::steady_clock::now();
    size_t pp = (size_t)c->ldp * c->ldp;
    hipLaunchKernelGGL(pad_density_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_D, c->d_Dpad, c->nao, c->ldp, c->d_iperm);
    double *d_Mpad = nullptr; // spin density Da - Db, padded like D (open shell only)
    if (d_Dspin) {
        HIPCHK(dev_malloc(&d_Mpad, sizeof(double) * pp));
        hipLaunchKernelGGL(pad_density_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_Dspin, d_Mpad, c->nao, c->ldp, c->d_iperm);
    }
    std::vector<int> shell_atom(c->nbas);
    for (int i = 0; i < c->nbas; i++) shell_atom[i] = c->shells[i].atom;
    int *d_shell_atom = nullptr;
    if (upload(&d_shell_atom, shell_atom)) return -1;
    // density-weighted screening: q_ab q_cd max|G| < grad_dtol skips the quartet (0 disables)
    double *d_dmax = nullptr;
    int *d_sh_ao = nullptr, *d_sh_n = nullptr;
    if (c->opt_grad_dtol > 0.0) {
        std::vector<int> sh_ao(c->nbas), sh_n(c->nbas);
        for (int i = 0; i < c->nbas; i++) { sh_ao[i] = c->shells[i].ao; sh_n[i] = 2 * c->shells[i].l + 1; }
        if (upload(&d_sh_ao, sh_ao) || upload(&d_sh_n, sh_n)) return -1;
        HIPCHK(dev_malloc(&d_dmax, sizeof(double) * (size_t)c->nbas * c->nbas));
        hipLaunchKernelGGL(shell_dmax_kernel, dim3((c->nbas * c->nbas + 255) / 256), dim3(256), 0, st, c->d_Dpad, d_Mpad, c->ldp, d_sh_ao, d_sh_n,
                           c->nbas, d_dmax);
        HIPCHK(hipGetLastError());
    }
    const int natm3 = c->natm * 3;
    double *d_gcopies = nullptr;
    HIPCHK(dev_malloc(&d_gcopies, sizeof(double) * (size_t)GRAD_COPIES * natm3));
    HIPCHK(hipMemsetAsync(d_gcopies, 0, sizeof(double) * (size_t)GRAD_COPIES * natm3, st));
    size_t WORK_DOUBLES = (size_t)8 << 17; // per buffer (plus / minus): what the molecule can need, at most `grad_work_mb`
    {
        const size_t cap = (size_t)std::max(8, c->opt_grad_work_mb) << 17;
        for (int bc = 0; bc < NPC; bc++)
            for (int kc = 0; kc <= bc; kc++) {
                const PairClass &B = c->pc[bc], &Kc = c->pc[kc];
                const int ne_up = std::max(ne_of(B.la + 1, B.lb), ne_of(Kc.la + 1, Kc.lb));
                const double need = (double)B.recs.size() * (double)Kc.recs.size() * (double)ne_up * (double)std::max(B.ne, Kc.ne);
                WORK_DOUBLES = (size_t)std::min<double>((double)cap, std::max<double>((double)WORK_DOUBLES, need));
            }
    }
    Scratch scr_wp, scr_wm, scr_gtasks, scr_live;
    if (scr_wp.ensure(c->device, SCR_GRAD_WP, sizeof(double) * WORK_DOUBLES) || scr_wm.ensure(c->device, SCR_GRAD_WM, sizeof(double) * WORK_DOUBLES)) return -1;
    double *d_wp = (double *)scr_wp.p, *d_wm = (double *)scr_wm.p;
    uint32_t *d_comp_p = nullptr, *d_comp_m = nullptr;
    HIPCHK(dev_malloc(&d_comp_p, sizeof(uint32_t) * 16384));
    HIPCHK(dev_malloc(&d_comp_m, sizeof(uint32_t) * 16384));
    int64_t *d_prefix = nullptr;
    size_t prefix_cap = 0;
    TaskIdx *d_tasks = nullptr, *d_live_tasks = nullptr;
    size_t tasks_cap = 0, live_tasks_cap = 0, live_cap = 0;
    int *d_live_counts = nullptr;
    int64_t *d_live_off = nullptr;
    const double tol = c->tol;
    int64_t batch_counter = 0;
    for (int bc = 0; bc < NPC; bc++)
        for (int kc = 0; kc <= bc; kc++) {
            PairClass &B = c->pc[bc], &Kc = c->pc[kc];
            if (B.recs.empty() || Kc.recs.empty()) continue;
            std::vector<int64_t> prefix;
            const int64_t ntask = class_prefix(B, Kc, bc == kc, tol, prefix);
            if (ntask == 0) continue;
            if (prefix.size() > prefix_cap) {
                if (d_prefix) dev_free(d_prefix);
                prefix_cap = prefix.size() * 2;
                HIPCHK(dev_malloc(&d_prefix, sizeof(int64_t) * prefix_cap));
            }
            HIPCHK(hipMemcpyAsync(d_prefix, prefix.data(), sizeof(int64_t) * prefix.size(), hipMemcpyHostToDevice, st));
            const TaskIdx *tasks_dev = nullptr;
            const bool live_ok = d_dmax && c->opt_grad_live && ntask >= 4096 && ntask <= ((int64_t)1 << 28);   // see ensure_live below
            if (c->opt_task_table && ntask >= 65536 && ntask <= ((int64_t)1 << 28) && !live_ok) {   // up to 3 permutations x 3 launches walk this task list
                if ((size_t)ntask > tasks_cap) {
                    if (scr_gtasks.ensure(c->device, SCR_GRAD_TASKS, sizeof(TaskIdx) * (size_t)ntask)) return -1;
                    d_tasks = (TaskIdx *)scr_gtasks.p;
                    tasks_cap = scr_gtasks.bytes / sizeof(TaskIdx);
                }
                hipLaunchKernelGGL(fill_tasks_kernel, dim3((unsigned)((ntask + 255) / 256)), dim3(256), 0, st, d_prefix, (int)B.recs.size(), ntask, d_tasks);
                tasks_dev = d_tasks;
            }
            HIPCHK(hipStreamSynchronize(st));
            // wave-per-quartet launches of this class walk the list of LIVE quartets (density-weighted screening done once, see
            // live_fill_kernel); built on first need -- classes wholly on the thread-per-quartet path never ask for it
            int64_t nlive = -1;
            auto ensure_live = [&]() -> int {
                // (lists beyond 2^28 quartets -- 2 GB -- are not built: those launches screen per wave as before)
                if (nlive >= 0 || !live_ok) return 0;
                const int nblk = (int)((ntask + 255) / 256);
                if ((size_t)nblk + 1 > live_cap) {
                    if (d_live_counts) dev_free(d_live_counts);
                    if (d_live_off) dev_free(d_live_off);
                    live_cap = (size_t)nblk + 1 + (size_t)nblk / 4;
                    HIPCHK(dev_malloc(&d_live_counts, sizeof(int) * live_cap));
                    HIPCHK(dev_malloc(&d_live_off, sizeof(int64_t) * live_cap));
                }
                if ((size_t)ntask > live_tasks_cap) {
                    if (scr_live.ensure(c->device, SCR_GRAD_LIVE, sizeof(TaskIdx) * (size_t)ntask)) return -1;
                    d_live_tasks = (TaskIdx *)scr_live.p;
                    live_tasks_cap = scr_live.bytes / sizeof(TaskIdx);
                }
                LiveArgs L{B.d_recs, Kc.d_recs, d_prefix, (int)B.recs.size(), ntask, B.d_q, Kc.d_q, d_dmax, c->nbas, hyb, c->opt_grad_dtol};
                hipLaunchKernelGGL(live_count_kernel, dim3(nblk), dim3(256), 0, st, L, d_live_counts);
                hipLaunchKernelGGL(live_scan_kernel, dim3(1), dim3(1024), 0, st, d_live_counts, nblk, d_live_off);
                hipLaunchKernelGGL(live_fill_kernel, dim3(nblk), dim3(256), 0, st, L, d_live_off, d_live_tasks);
                HIPCHK(hipGetLastError());
                int64_t n = 0;
                HIPCHK(hipMemcpyAsync(&n, d_live_off + nblk, sizeof(int64_t), hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                nlive = n;
                if (getenv("MI355_DEBUG"))
                    fprintf(stderr, "[mi355] grad class pair (%d%d|%d%d): %ld of %ld quartets live\n", B.la, B.lb, Kc.la, Kc.lb, (long)nlive, (long)ntask);
                return 0;
            };
            // perm 0 (derivative on P.sh_i, the costliest: highest l) is skipped: sum of the four forces = 0
            for (int perm = 1; perm < 4; perm++) {
                const bool swap = perm >= 2;
                const int orient = perm & 1;
                PairClass &Dc = swap ? Kc : B;   // class of the differentiated pair
                PairClass &Oc = swap ? B : Kc;   // class of the other pair (plain ket)
                const int l1 = orient == 0 ? Dc.la : Dc.lb, l2 = orient == 0 ? Dc.lb : Dc.la;
                const int lc = Oc.la, ldd = Oc.lb;
                if (c->opt_eri_tpq && B.mean_np * Kc.mean_np <= c->opt_tpq_maxprim && ntask >= 32768) {   // low classes, shallow contraction: thread per (quartet, permutation)
                    TpqGradArgs Q{};
                    Q.dplus = Dc.d_g_recs[orient][0]; Q.dminus = l1 >= 1 ? Dc.d_g_recs[orient][1] : nullptr; Q.ket = Oc.d_recs;
                    Q.prim = c->d_prim; Q.prefix = d_prefix; Q.nbra = (int)B.recs.size();
                    // this rank's contiguous share of the task list (the compacted live list when the class has a kernel)
                    const bool has_tpq = tpq_grad_has_class(l1, l2, lc, ldd);
                    if (has_tpq && ensure_live()) return -1;
                    const bool live_q = has_tpq && nlive >= 0;
                    const int64_t nt_q = live_q ? nlive : ntask;
                    const int64_t lo_t = nt_q * rank / nranks, hi_t = nt_q * (rank + 1) / nranks;
                    Q.t0 = lo_t; Q.ntask = hi_t - lo_t;
                    Q.tasks = live_q ? d_live_tasks : nullptr;
                    Q.swap = swap ? 1 : 0; Q.same_class = (bc == kc);
                    Q.c2s = c->d_c2s;
                    for (int q = 0; q <= LMAX + 1; q++) Q.c2s_off[q] = c->c2s_off[q];
                    Q.rys = c->rys; Q.shell_xyz = c->d_shell_xyz; Q.D = c->d_Dpad; Q.Dm = d_Mpad; Q.ld = c->ldp; Q.hyb = hyb;
                    Q.shell_atom = d_shell_atom; Q.grad = d_gcopies; Q.natm3 = natm3; Q.inv_from_second = swap ? 0 : 1;
                    Q.q_bra = Dc.d_q; Q.q_ket = Oc.d_q; Q.dmax = live_q ? nullptr : d_dmax; Q.nbas_d = c->nbas; Q.dtol = c->opt_grad_dtol;
                    const bool dbg1 = getenv("MI355_DEBUG") != nullptr;
                    auto tq0 = std::chrono::steady_clock::now();
                    if (dbg1) hipStreamSynchronize(st);
                    const int used = launch_eri_tpq_grad(l1, l2, lc, ldd, Q, st);
                    if (used < 0) return -1;
                    if (used) {
                        if (dbg1) {
                            hipStreamSynchronize(st);
                            fprintf(stderr, "[mi355] grad class (%d%d|%d%d) perm %d: %ld quartets, %.3f s (thread per quartet)\n", l1, l2, lc, ldd, perm,
                                    (long)ntask, std::chrono::duration<double>(std::chrono::steady_clock::now() - tq0).count());
                        }
                        continue;
                    }
                }
                if (ensure_live()) return -1;
                const bool use_live = nlive >= 0;
                if (use_live && nlive == 0) continue;
                const int64_t ntask_w = use_live ? nlive : ntask;            // tasks the wave-per-quartet launches walk
                const TaskIdx *tasks_w = use_live ? d_live_tasks : tasks_dev;
                const double *dmax_w = use_live ? nullptr : d_dmax;          // the live list is already screened
                EriArgs Ep{}, Em{};
                setup_eri_dims(Ep, l1 + 1, l2, lc, ldd);
                const bool has_m = l1 >= 1;
                if (has_m) setup_eri_dims(Em, l1 - 1, l2, lc, ldd);
                std::vector<uint32_t> comp;
                build_comp_table(l1 + 1, l2, lc, ldd, comp);
                if (comp.size() > 16384) return fail("component table too large");
                HIPCHK(hipMemcpyAsync(d_comp_p, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
                HIPCHK(hipStreamSynchronize(st));
                if (has_m) {
                    build_comp_table(l1 - 1, l2, lc, ldd, comp);
                    HIPCHK(hipMemcpyAsync(d_comp_m, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
                    HIPCHK(hipStreamSynchronize(st));
                }
                Ep.bra = Dc.d_g_recs[orient][0]; Ep.ket = Oc.d_recs; Ep.prim = c->d_prim; Ep.prefix = d_prefix; Ep.nbra = (int)B.recs.size();
                Ep.comp = d_comp_p; Ep.work = d_wp; Ep.rys = c->rys; Ep.diag = 0; Ep.swap = swap ? 1 : 0;
                if (has_m) {
                    Em.bra = Dc.d_g_recs[orient][1]; Em.ket = Oc.d_recs; Em.prim = c->d_prim; Em.prefix = d_prefix; Em.nbra = Ep.nbra;
                    Em.comp = d_comp_m; Em.work = d_wm; Em.rys = c->rys; Em.diag = 0; Em.swap = Ep.swap;
                }
                if (c->opt_grad_rows) {   // row kernel: no hand-over, one launch (see eri_grad_rows_kernel)
                    const int ne_p = ne_of(l1 + 1, l2), ne_m = has_m ? ne_of(l1 - 1, l2) : 0, ne_all = ne_p + ne_m;
                    GradRowsArgs R{};
                    if (ne_all >= c->opt_grad_rows_min && launch_grad_rows(lc, ldd, ne_all, R, 0, st, true) == 1) {
                        R.dplus = Dc.d_g_recs[orient][0]; R.dminus = has_m ? Dc.d_g_recs[orient][1] : nullptr; R.ket = Oc.d_recs;
                        R.prim = c->d_prim; R.Mbuf = c->d_M; R.tasks = tasks_w; R.prefix = d_prefix; R.nbra = (int)B.recs.size();
                        const int64_t lo_t = ntask_w * rank / nranks, hi_t = ntask_w * (rank + 1) / nranks;
                        R.t0 = lo_t; R.ntask = hi_t - lo_t;
                        R.swap = swap ? 1 : 0; R.same_class = (bc == kc);
                        const RowsGeom gm = rows_geometry(ne_all);
                        R.nmax = Ep.nmax; R.nroots = Ep.nroots; R.tsz = Ep.tsz; R.PB = std::max(1, gm.gsz / (3 * Ep.nroots));
                        R.ne_p = ne_p; R.ne_m = ne_m; R.nsab = (2 * l1 + 1) * (2 * l2 + 1); R.ns2 = 2 * l2 + 1;
                        R.comp_p = d_comp_p; R.comp_m = d_comp_m; R.rys = c->rys;
                        R.D = c->d_Dpad; R.Dm = d_Mpad; R.ld = c->ldp; R.hyb = hyb; R.shell_atom = d_shell_atom; R.grad = d_gcopies; R.natm3 = natm3;
                        R.inv_from_second = swap ? 0 : 1;
                        R.q_bra = Dc.d_q; R.q_ket = Oc.d_q; R.dmax = dmax_w; R.nbas_d = c->nbas; R.dtol = c->opt_grad_dtol;
                        const int nscd_ = Oc.nsab, nf_ = Oc.ne;
                        const size_t shm_q = sizeof(double) * ((size_t)R.PB * R.nroots * 3 * R.tsz + (size_t)R.PB * 2 * R.nroots + (size_t)R.PB +
                                                               (size_t)nscd_ * nf_ + (size_t)R.nsab * nscd_);
                        if (R.ntask < ((int64_t)1 << 31)) {
                            const bool dbgr = getenv("MI355_DEBUG") != nullptr;
                            auto tr0 = std::chrono::steady_clock::now();
                            if (dbgr) hipStreamSynchronize(st);
                            const int used = R.ntask > 0 ? launch_grad_rows(lc, ldd, ne_all, R, shm_q, st) : 1;
                            if (used < 0) return -1;
                            if (used) {
                                if (dbgr) {
                                    hipStreamSynchronize(st);
                                    fprintf(stderr, "[mi355] grad class (%d%d|%d%d) perm %d: %ld quartets, %.4f s (row kernel, %d rows x %d lanes)\n", l1, l2, lc,
                                            ldd, perm, (long)ntask_w, std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count(), gm.rows, gm.gsz);
                                }
                                continue;
                            }
                        }
                    }
                }
                Ep.tasks = Em.tasks = tasks_w;
                Ep.prim_lds = Em.prim_lds = (c->opt_prim_lds && B.max_np + Kc.max_np <= 160) ? B.max_np + Kc.max_np : 0;
                Ep.h_shared_np = B.mean_np; Ep.h_vary_mean = Kc.mean_np; Ep.h_vary_max4 = Kc.max4_np;
                Em.h_shared_np = B.mean_np; Em.h_vary_mean = Kc.mean_np; Em.h_vary_max4 = Kc.max4_np;
                Ep.q_bra = Dc.d_q; Ep.q_ket = Oc.d_q; Ep.dmax = dmax_w; Ep.nbas_d = c->nbas; Ep.dtol = c->opt_grad_dtol; Ep.hyb = hyb;
                if (has_m) { Em.q_bra = Ep.q_bra; Em.q_ket = Ep.q_ket; Em.dmax = dmax_w; Em.nbas_d = c->nbas; Em.dtol = Ep.dtol; Em.hyb = hyb; }
                GradXfArgs X{};
                X.q_bra = Ep.q_bra; X.q_ket = Ep.q_ket; X.dmax = dmax_w; X.nbas_d = c->nbas; X.dtol = Ep.dtol;
                X.dplus = Dc.d_g_recs[orient][0]; X.dminus = has_m ? Dc.d_g_recs[orient][1] : nullptr; X.ket = Oc.d_recs;
                X.Mbuf = c->d_M; X.prefix = d_prefix; X.nbra = Ep.nbra; X.swap = Ep.swap; X.same_class = (bc == kc);
                X.ne_p = ne_of(l1 + 1, l2); X.ne_m = has_m ? ne_of(l1 - 1, l2) : 0; X.nf = Oc.ne;
                X.ns1 = 2 * l1 + 1; X.ns2 = 2 * l2 + 1; X.nscd = Oc.nsab; X.nsd = 2 * Oc.lb + 1;
                X.work_p = d_wp; X.work_m = d_wm; X.ncomp_p = Ep.ncomp; X.ncomp_m = has_m ? Em.ncomp : 0;
                X.D = c->d_Dpad; X.Dm = d_Mpad; X.ld = c->ldp; X.hyb = hyb; X.shell_atom = d_shell_atom; X.grad = d_gcopies; X.natm3 = natm3;
                X.inv_from_second = swap ? 0 : 1;
                X.tasks = tasks_w;
                size_t shm = sizeof(double) * ((size_t)X.ne_p * X.nf + (size_t)X.ne_m * X.nf + (size_t)X.ns1 * X.ns2 * (X.nf + X.nscd));
                if (shm > 160 * 1024) return fail("gradient contraction needs %zu bytes of LDS", shm);
                const bool dbg = getenv("MI355_DEBUG") != nullptr;
                auto tc0 = std::chrono::steady_clock::now();
                if (dbg) hipStreamSynchronize(st);
                int64_t per = std::min<int64_t>((int64_t)(WORK_DOUBLES / Ep.ncomp), (int64_t)1 << 23);
                if (nranks > 1) per = std::min<int64_t>(per, std::max<int64_t>(1024, ntask_w / (8 * nranks)));
                for (int64_t t0 = 0; t0 < ntask_w; t0 += per) {
                    if ((int)((batch_counter++) % nranks) != rank) continue; // batches dealt round-robin to ranks
                    int nb = (int)std::min<int64_t>(per, ntask_w - t0);
                    Ep.t0 = t0; Ep.ntask = nb;
                    if (launch_eri(c, Ep, nb, st)) return -1;
                    if (has_m) { Em.t0 = t0; Em.ntask = nb; if (launch_eri(c, Em, nb, st)) return -1; }
                    X.t0 = t0; X.nbatch = nb;
                    const int big = std::max({X.ns1 * X.ns2 * X.nscd, X.ns1 * X.ns2 * X.nf, X.ns1 * X.ns2 * X.ne_p});
                    if (big <= 48 && shm * 4 <= 64 * 1024) { // small classes: four quartets per wave (16 lanes each)
                        hipLaunchKernelGGL((eri_grad_contract<16, false>), dim3((nb + 3) / 4), dim3(64), shm * 4, st, X);
                    } else if (mfma_worthwhile(X.ns1 * X.ns2, X.nf, X.nscd) || mfma_worthwhile(X.ns1 * X.ns2, X.ne_p, X.nf) ||
                               (has_m && mfma_worthwhile(X.ns1 * X.ns2, X.ne_m, X.nf))) {
                        // (two or four waves sharing one quartet's LDS blocks -- eri_grad_contract<128|256, true> -- were measured at
                        // -2 % / +8 %, the six density sub-blocks of G staged in LDS first at +2 %: the kernel is parked 70 % of its
                        // wave cycles, but neither on a shortage of lanes nor on the G gather)
                        if (shm > 64 * 1024)
                            HIPCHK(hipFuncSetAttribute((const void *)eri_grad_contract<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                        hipLaunchKernelGGL((eri_grad_contract<64, true>), dim3(nb), dim3(64), shm, st, X);
                    } else {
                        if (shm > 64 * 1024)
                            HIPCHK(hipFuncSetAttribute((const void *)eri_grad_contract<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                        hipLaunchKernelGGL((eri_grad_contract<64, false>), dim3(nb), dim3(64), shm, st, X);
                    }
                    HIPCHK(hipGetLastError());
                }
                if (dbg) {
                    hipStreamSynchronize(st);
                    fprintf(stderr, "[mi355] grad class (%d%d|%d%d) perm %d: %ld quartets, %.3f s\n", l1, l2, lc, ldd, perm, (long)ntask,
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count());
                }
            }
        }
    hipLaunchKernelGGL(grad_reduce_copies_kernel, dim3((natm3 + 63) / 64), dim3(64), 0, st, d_gcopies, natm3, d_grad);
    HIPCHK(hipStreamSynchronize(st));
    dev_free(d_gcopies);
    if (getenv("MI355_DEBUG"))
        fprintf(stderr, "[mi355] grad_eri: variant records %.3f s, derivative quartets %.3f s\n",
                std::chrono::duration<double>(tg1 - tg0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tg1).count());
    if (d_prefix) dev_free(d_prefix);
    if (d_live_counts) dev_free(d_live_counts);
    if (d_live_off) dev_free(d_live_off);
    dev_free(d_comp_p); dev_free(d_comp_m); dev_free(d_shell_atom);   // (the Scratch handles go back to the per-device cache)
    if (d_dmax) dev_free(d_dmax);
    if (d_Mpad) dev_free(d_Mpad);
    if (d_sh_ao) dev_free(d_sh_ao);
    if (d_sh_n) dev_free(d_sh_n);
    return 0;
}

// =================================================================================================
// Nuclear gradient of the density-fitted two-electron energy (`mf.density_fit().nuc_grad_method()`
// is PySCF idiom [MEM], the reference never calls it):
//
//   grad_X += sum_{ab,P} Z3[a][b][P] d/dX (ab|P)  +  sum_{PQ} Z2[P][Q] d/dX (P|Q)
//
// Z3 (symmetric in ab) and Z2 (symmetric) are the three- and two-index densities the host forms from the fitted tensor
// (python/mi355scf/df.py: Z3 = c_P D_ab - hyb/2 Gamma^P_ab, Z2 = -c c^T / 2 + hyb/4 C.Gamma).  The derivative integrals go
// through the SAME Rys kernel and contraction kernel as the four-centre gradient: the differentiated pair is an orbital pair
// (either shell: two launches) or an auxiliary "pair" (P, unit function), its (l+1) / (l-1) variants carry 2 alpha c / c, the
// ket is an auxiliary pair, and the force on the auxiliary centre follows from translational invariance (minus the force on
// the differentiated shell).  Batches are dealt round-robin to `nranks` callers; the caller sums the partial gradients.
// =================================================================================================
extern "C" int mi_df_grad(mi_ctx *c, mi_ctx *aux, const double *d_Z3, const double *d_Z2, double *d_grad, int rank, int nranks, void *stream)
{
    if (!c || !aux || !d_grad) return fail("mi_df_grad: null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail("mi_df_grad: bad rank/nranks");
    if (aux->nbas < 2) return fail("mi_df_grad: the auxiliary context needs at least one function plus the unit shell");
    if (aux->natm != c->natm) return fail("mi_df_grad: orbital and auxiliary contexts must share the atom list");
    const ShellH &U = aux->shells.back();
    if (U.l != 0 || U.nprim != 1 || U.exps[0] != 0.0) return fail("mi_df_grad: the last auxiliary shell must be the unit s function (exponent 0)");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const int naux = aux->nao - 1, unit_ao = aux->nao - 1, nP = aux->nbas - 1, nao = c->nao;
    std::vector<std::vector<double>> c2s(LMAX + 1);
    for (int l = 0; l <= LMAX; l++) c2s_generic(l, c2s[l]);
    std::vector<double> prim, Mbuf;
    // variant records: orbital pairs [class][orientation][l+1 | l-1] (parallel arrays), auxiliary pairs [l][l+1 | l-1]; plain auxiliary kets [l]
    std::vector<PairRec> orb[NPC][2][2], axv[LMAX + 1][2], axk[LMAX + 1];
    // pass 1 (serial): offsets of every variant record; pass 2 (OpenMP): primitive records and HRR x derivative x c2s matrices
    struct Job { const ShellH *I, *J; PairRec *R; int sign; };
    std::vector<Job> jobs;
    size_t prim_end = 0, m_end = 0;
    auto add_variant = [&](const ShellH &I, const ShellH &J, int s1, int s2, int ao1, int ao2, int sign, PairRec &R) {
        double r2 = 0.0;
        for (int d = 0; d < 3; d++) r2 += (I.r[d] - J.r[d]) * (I.r[d] - J.r[d]);
        int np = 0;
        for (int ip = 0; ip < I.nprim; ip++)
            for (int jp = 0; jp < J.nprim; jp++)
                if (I.exps[ip] * J.exps[jp] / (I.exps[ip] + J.exps[jp]) * r2 <= 80.0) np++;
        R = PairRec{s1, s2, ao1, ao2, (int)prim_end, np, (int)m_end, 0};
        prim_end += np;
        m_end += (size_t)3 * (2 * I.l + 1) * (2 * J.l + 1) * ne_of(I.l + sign, J.l);
    };
    auto fill_variant = [&](const Job &jb) {
        const ShellH &I = *jb.I, &J = *jb.J;
        double AB[3] = {I.r[0] - J.r[0], I.r[1] - J.r[1], I.r[2] - J.r[2]};
        double r2 = AB[0] * AB[0] + AB[1] * AB[1] + AB[2] * AB[2];
        double *dst = prim.data() + (size_t)jb.R->prim_off * 8;
        for (int ip = 0; ip < I.nprim; ip++)
            for (int jp = 0; jp < J.nprim; jp++) {
                double a = I.exps[ip], b = J.exps[jp], p = a + b, mu = a * b / p;
                if (mu * r2 > 80.0) continue;
                double K = I.coef[ip] * J.coef[jp] * std::exp(-mu * r2) * (jb.sign > 0 ? 2.0 * a : 1.0), Pc[3];
                for (int d = 0; d < 3; d++) Pc[d] = (a * I.r[d] + b * J.r[d]) / p;
                double rec[8] = {p, Pc[0], Pc[1], Pc[2], Pc[0] - I.r[0], Pc[1] - I.r[1], Pc[2] - I.r[2], K};
                memcpy(dst, rec, sizeof rec);
                dst += 8;
            }
        build_M_deriv(I.l, J.l, jb.sign, AB, c2s[I.l], c2s[J.l], Mbuf.data() + jb.R->m_off);
    };
    if (d_Z3)
        for (int A = 0; A < c->nbas; A++)
            for (int B = 0; B <= A; B++) {
                int si = A, sj = B;
                if (c->shells[si].l < c->shells[sj].l) std::swap(si, sj);
                const ShellH &I = c->shells[si], &J = c->shells[sj];
                double r2 = 0.0;
                for (int d = 0; d < 3; d++) r2 += (I.r[d] - J.r[d]) * (I.r[d] - J.r[d]);
                bool any = false;
                for (int ip = 0; ip < I.nprim && !any; ip++)
                    for (int jp = 0; jp < J.nprim; jp++)
                        if (I.exps[ip] * J.exps[jp] / (I.exps[ip] + J.exps[jp]) * r2 <= 80.0) { any = true; break; }
                if (!any) continue;
                const int q = pc_index(I.l, J.l);
                for (int o = 0; o < 2; o++) {
                    const ShellH &F = o == 0 ? I : J, &S = o == 0 ? J : I;
                    const int s1 = o == 0 ? si : sj, s2 = o == 0 ? sj : si;
                    for (int sg = 0; sg < 2; sg++) {
                        PairRec R{-1, -1, 0, 0, 0, 0, 0, 0};
                        if (F.l + (sg == 0 ? 1 : -1) >= 0) add_variant(F, S, s1, s2, F.ao_nat, S.ao_nat, sg == 0 ? +1 : -1, R);
                        orb[q][o][sg].push_back(R);
                    }
                }
            }
    for (int Pn = 0; Pn < nP; Pn++) {
        const ShellH &S = aux->shells[Pn];
        for (int sg = 0; sg < 2; sg++) {
            PairRec R{-1, -1, 0, 0, 0, 0, 0, 0};
            if (S.l + (sg == 0 ? 1 : -1) >= 0) add_variant(S, U, Pn, aux->nbas - 1, S.ao_nat, unit_ao, sg == 0 ? +1 : -1, R);
            axv[S.l][sg].push_back(R);
        }
        PairRec R{Pn, aux->nbas - 1, S.ao_nat, unit_ao, (int)prim_end, S.nprim, (int)m_end, 0};     // plain ket record (filled below)
        prim_end += S.nprim;
        m_end += (size_t)(2 * S.l + 1) * ncart(S.l);
        axk[S.l].push_back(R);
    }
    if (m_end > (size_t)INT32_MAX || prim_end > (size_t)INT32_MAX) return fail("mi_df_grad: record buffers exceed 2^31 entries");
    prim.resize(prim_end * 8);
    Mbuf.resize(m_end);
    for (int q = 0; q < NPC; q++)
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++)
                for (PairRec &R : orb[q][o][sg])
                    if (R.sh_i >= 0) jobs.push_back({&c->shells[R.sh_i], &c->shells[R.sh_j], &R, sg == 0 ? +1 : -1});
    for (int l = 0; l <= LMAX; l++)
        for (int sg = 0; sg < 2; sg++)
            for (PairRec &R : axv[l][sg])
                if (R.sh_i >= 0) jobs.push_back({&aux->shells[R.sh_i], &U, &R, sg == 0 ? +1 : -1});
#pragma omp parallel for schedule(dynamic, 64) num_threads(host_threads())
    for (size_t q = 0; q < jobs.size(); q++) fill_variant(jobs[q]);
    for (int l = 0; l <= LMAX; l++)
        for (PairRec &R : axk[l]) {
            const ShellH &S = aux->shells[R.sh_i];
            for (int ip = 0; ip < S.nprim; ip++) {
                double rec[8] = {S.exps[ip], S.r[0], S.r[1], S.r[2], 0.0, 0.0, 0.0, S.coef[ip] * U.coef[0]};
                memcpy(prim.data() + ((size_t)R.prim_off + ip) * 8, rec, sizeof rec);
            }
            double AB[3] = {0.0, 0.0, 0.0};
            build_M(S.l, 0, AB, c2s[S.l], c2s[0], Mbuf.data() + R.m_off);
        }
    std::vector<int> atom_o(c->nbas), atom_a(aux->nbas);
    for (int i = 0; i < c->nbas; i++) atom_o[i] = c->shells[i].atom;
    for (int i = 0; i < aux->nbas; i++) atom_a[i] = aux->shells[i].atom;
    double *d_prim = nullptr, *d_M = nullptr, *d_wp = nullptr, *d_wm = nullptr, *d_gcopies = nullptr;
    uint32_t *d_comp_p = nullptr, *d_comp_m = nullptr;
    int64_t *d_prefix = nullptr;
    int *d_atom_o = nullptr, *d_atom_a = nullptr;
    PairRec *d_orb[NPC][2][2] = {}, *d_axv[LMAX + 1][2] = {}, *d_axk[LMAX + 1] = {};
    if (upload(&d_prim, prim) || upload(&d_M, Mbuf) || upload(&d_atom_o, atom_o) || upload(&d_atom_a, atom_a)) return -1;
    for (int q = 0; q < NPC; q++)
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++)
                if (!orb[q][o][sg].empty() && upload(&d_orb[q][o][sg], orb[q][o][sg])) return -1;
    for (int l = 0; l <= LMAX; l++) {
        for (int sg = 0; sg < 2; sg++)
            if (!axv[l][sg].empty() && upload(&d_axv[l][sg], axv[l][sg])) return -1;
        if (!axk[l].empty() && upload(&d_axk[l], axk[l])) return -1;
    }
    const size_t WORK_DOUBLES = (size_t)32 << 20;
    HIPCHK(dev_malloc(&d_wp, sizeof(double) * WORK_DOUBLES));
    HIPCHK(dev_malloc(&d_wm, sizeof(double) * WORK_DOUBLES));
    HIPCHK(dev_malloc(&d_comp_p, sizeof(uint32_t) * 16384));
    HIPCHK(dev_malloc(&d_comp_m, sizeof(uint32_t) * 16384));
    const int natm3 = c->natm * 3;
    HIPCHK(dev_malloc(&d_gcopies, sizeof(double) * (size_t)GRAD_COPIES * natm3));
    HIPCHK(hipMemsetAsync(d_gcopies, 0, sizeof(double) * (size_t)GRAD_COPIES * natm3, st));
    size_t prefix_cap = 0;
    int64_t batch_counter = 0;
    const bool dbg = getenv("MI355_DEBUG") != nullptr;
    // one pass: differentiated pairs `dp`/`dm` (l1 first, l2 second) against the auxiliary kets of angular momentum lk
    auto run = [&](const PairRec *dp, const PairRec *dm, int nbra, int l1, int l2, int lk, const double *Z, int64_t zs_i, int64_t zs_j,
                   const int *bra_atom, double w0) -> int {
        const int nket = (int)axk[lk].size();
        if (nbra == 0 || nket == 0) return 0;
        const auto tr0 = std::chrono::steady_clock::now();
        std::vector<int64_t> prefix(nbra + 1);
        for (int b = 0; b <= nbra; b++) prefix[b] = (int64_t)b * nket;
        const int64_t ntask = prefix.back();
        append_coarse_index(prefix);
        if (prefix.size() > prefix_cap) {
            if (d_prefix) dev_free(d_prefix);
            prefix_cap = prefix.size() * 2;
            HIPCHK(dev_malloc(&d_prefix, sizeof(int64_t) * prefix_cap));
        }
        HIPCHK(hipMemcpyAsync(d_prefix, prefix.data(), sizeof(int64_t) * prefix.size(), hipMemcpyHostToDevice, st));
        const bool has_m = l1 >= 1;
        EriArgs Ep{}, Em{};
        setup_eri_dims(Ep, l1 + 1, l2, lk, 0);
        if (has_m) setup_eri_dims(Em, l1 - 1, l2, lk, 0);
        std::vector<uint32_t> comp;
        build_comp_table(l1 + 1, l2, lk, 0, comp);
        if (comp.size() > 16384) return fail("component table too large");
        HIPCHK(hipMemcpyAsync(d_comp_p, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        if (has_m) {
            build_comp_table(l1 - 1, l2, lk, 0, comp);
            HIPCHK(hipMemcpyAsync(d_comp_m, comp.data(), sizeof(uint32_t) * comp.size(), hipMemcpyHostToDevice, st));
            HIPCHK(hipStreamSynchronize(st));
        }
        Ep.bra = dp; Ep.ket = d_axk[lk]; Ep.prim = d_prim; Ep.prefix = d_prefix; Ep.nbra = nbra; Ep.comp = d_comp_p; Ep.work = d_wp;
        Ep.rys = c->rys; Ep.diag = 0;
        if (has_m) { Em.bra = dm; Em.ket = d_axk[lk]; Em.prim = d_prim; Em.prefix = d_prefix; Em.nbra = nbra; Em.comp = d_comp_m; Em.work = d_wm;
                     Em.rys = c->rys; Em.diag = 0; }
        GradXfArgs X{};
        X.dplus = dp; X.dminus = has_m ? dm : nullptr; X.ket = d_axk[lk]; X.Mbuf = d_M; X.prefix = d_prefix; X.nbra = nbra;
        X.ne_p = ne_of(l1 + 1, l2); X.ne_m = has_m ? ne_of(l1 - 1, l2) : 0; X.nf = ncart(lk);
        X.ns1 = 2 * l1 + 1; X.ns2 = 2 * l2 + 1; X.nscd = 2 * lk + 1; X.nsd = 1;
        X.work_p = d_wp; X.work_m = d_wm; X.ncomp_p = Ep.ncomp; X.ncomp_m = has_m ? Em.ncomp : 0;
        X.shell_atom = bra_atom; X.ket_atom = d_atom_a; X.grad = d_gcopies; X.natm3 = natm3;
        X.Z = Z; X.zs_i = zs_i; X.zs_j = zs_j; X.w0 = w0;
        const size_t shm = sizeof(double) * ((size_t)X.ne_p * X.nf + (size_t)X.ne_m * X.nf + (size_t)X.ns1 * X.ns2 * (X.nf + X.nscd));
        if (shm > 160 * 1024) return fail("gradient contraction needs %zu bytes of LDS", shm);
        int64_t per = std::min<int64_t>((int64_t)(WORK_DOUBLES / Ep.ncomp), (int64_t)1 << 22);
        if (nranks > 1) per = std::min<int64_t>(per, std::max<int64_t>(1024, ntask / (8 * nranks)));
        const bool mf = mfma_worthwhile(X.ns1 * X.ns2, X.nf, X.nscd) || mfma_worthwhile(X.ns1 * X.ns2, X.ne_p, X.nf) ||
                        (has_m && mfma_worthwhile(X.ns1 * X.ns2, X.ne_m, X.nf));
        for (int64_t t0 = 0; t0 < ntask; t0 += per) {
            if ((int)((batch_counter++) % nranks) != rank) continue;
            const int nb = (int)std::min<int64_t>(per, ntask - t0);
            Ep.t0 = t0; Ep.ntask = nb;
            if (launch_eri(c, Ep, nb, st)) return -1;
            if (has_m) { Em.t0 = t0; Em.ntask = nb; if (launch_eri(c, Em, nb, st)) return -1; }
            X.t0 = t0; X.nbatch = nb;
            if (mf) {
                if (shm > 64 * 1024)
                    HIPCHK(hipFuncSetAttribute((const void *)eri_grad_contract<64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                hipLaunchKernelGGL((eri_grad_contract<64, true, true>), dim3(nb), dim3(64), shm, st, X);
            } else {
                if (shm > 64 * 1024)
                    HIPCHK(hipFuncSetAttribute((const void *)eri_grad_contract<64, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                hipLaunchKernelGGL((eri_grad_contract<64, false, true>), dim3(nb), dim3(64), shm, st, X);
            }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipStreamSynchronize(st));
        if (dbg)
            fprintf(stderr, "[mi355] df_grad (%d %d|%d): %ld tasks, %.3f s\n", l1, l2, lk, (long)ntask,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count());
        return 0;
    };
    int rc = 0;
    if (d_Z3)   // sum_ab over canonical shell pairs: weight 2 (1 on the diagonal, halved in the kernel), both orientations
        for (int la = 0; la <= LMAX && !rc; la++)
            for (int lb = 0; lb <= la && !rc; lb++)
                for (int o = 0; o < 2 && !rc; o++)
                    for (int lk = 0; lk <= LMAX && !rc; lk++) {
                        const int q = pc_index(la, lb);
                        rc = run(d_orb[q][o][0], d_orb[q][o][1], (int)orb[q][o][0].size(), o == 0 ? la : lb, o == 0 ? lb : la, lk, d_Z3,
                                 (int64_t)nao * naux, (int64_t)naux, d_atom_o, 2.0);
                    }
    if (d_Z2)   // every ordered pair (P, Q): +v on the centre of P, -v on the centre of Q
        for (int lp = 0; lp <= LMAX && !rc; lp++)
            for (int lk = 0; lk <= LMAX && !rc; lk++)
                rc = run(d_axv[lp][0], d_axv[lp][1], (int)axv[lp][0].size(), lp, 0, lk, d_Z2, (int64_t)naux, 0, d_atom_a, 1.0);
    if (!rc) {
        hipLaunchKernelGGL(grad_reduce_copies_kernel, dim3((natm3 + 63) / 64), dim3(64), 0, st, d_gcopies, natm3, d_grad);
        HIPCHK(hipStreamSynchronize(st));
    }
    for (int q = 0; q < NPC; q++)
        for (int o = 0; o < 2; o++)
            for (int sg = 0; sg < 2; sg++) if (d_orb[q][o][sg]) dev_free(d_orb[q][o][sg]);
    for (int l = 0; l <= LMAX; l++) {
        for (int sg = 0; sg < 2; sg++) if (d_axv[l][sg]) dev_free(d_axv[l][sg]);
        if (d_axk[l]) dev_free(d_axk[l]);
    }
    dev_free(d_prim); dev_free(d_M); dev_free(d_wp); dev_free(d_wm); dev_free(d_comp_p); dev_free(d_comp_m); dev_free(d_gcopies);
    dev_free(d_atom_o); dev_free(d_atom_a);
    if (d_prefix) dev_free(d_prefix);
    return rc;
}

// =================================================================================================
// Fused SP2 step for small matrices (N <= 512): ONE launch per purification step instead of a rocBLAS
// DGEMM (launch-latency bound at 10 us for N=114) plus an update kernel.
//   Xc  = first ? Xp : ( |tr X2p - N| < |2 tr Xp - tr X2p - N| ? X2p : 2 Xp - X2p )   (formed on the fly)
//   X2c = Xc * Xc^T (X is symmetric), 16x16 output tile per workgroup on v_mfma_f64_16x16x4_f64,
//   K split over the 4 waves of the workgroup, row panels staged in LDS; every diagonal workgroup b writes its share of
//   tr Xc, tr X2c to trc[2b], trc[2b+1] and the next launch adds the SP2_TRS slots in index order for its branch decision
//   (no atomics: bit-identical on every rank of a sharded run).
// =================================================================================================
#define SP2_TRS 32 /* trace slots per step = max diagonal workgroups (N <= 512) */
// LDS row padding of the panels (doubles).  The MFMA operand read is "16 rows x 4 consecutive k" per wave; a row stride of
// 2 (mod 32) doubles = 4 banks (mod 64) puts the 32 lanes of a half wave on 64 distinct banks, where +4 makes rows r and r + 8
// share banks.  Measured (round 2): no difference in kernel time for either padding, here (7.3-7.4 us per pass) or in xc_vmat
// (518 vs 543 us on boxes whose J/K pass differed by more) -- the operand reads are not what bounds these kernels.
#define SP2_PAD 2
// MAXM = 16-column groups a thread loads per panel row in one batch: all 4 x MAXM loads of a thread are in flight together
// (one memory round trip for kpad <= 16 MAXM; larger matrices take two batches).  Thread t owns row t/16 and columns
// t%16 + 16 m of both panels: no integer division, 128-byte segments per row and instruction.
// PLAN = true: the step is a quadratic fixed by the host, Xc = ca X2p + cb Xp + cc I (no dependence on the previous
// launch's traces; `first`: Xc = cb Xp + cc I, the affine map of the Fock matrix); PLAN = false: trace-correcting SP2.
struct Sp2Coef { double a, b, c; double out_scale; };   // out_scale: factor on the stored X (0 = 1); the traces stay those of X
template <int MAXM, bool PLAN>
__global__ __launch_bounds__(256) void sp2_fused_kernel(const double *__restrict__ Xp, const double *__restrict__ X2p,
                                                        const double *__restrict__ trp, int first, int n, int kpad, double target,
                                                        double *__restrict__ Xc, double *__restrict__ X2c, double *__restrict__ trc,
                                                        Sp2Coef cf)
{
    extern __shared__ double lds[];
    double *Pa = lds;                       // [16][kpad+SP2_PAD]  rows i0..i0+15 of Xc
    double *Pb = lds + 16 * (kpad + SP2_PAD);     // [16][kpad+SP2_PAD]  rows j0..j0+15 of Xc
    double *red = Pb + 16 * (kpad + SP2_PAD);     // [4][256]  (its first 64 doubles also stage the partial traces)
    const int ldp = kpad + SP2_PAD;
    // X and X^2 are symmetric: only the tiles (I >= J) are computed (grid = nb(nb+1)/2 workgroups -- 153 for N = 264, one round
    // on 256 CUs instead of 289 in two) and every off-diagonal tile is stored with its mirror image.  The mirrored values are
    // bit-identical to what the (J, I) workgroup used to compute (same products, same k order).
    int bI = (int)((sqrt(8.0 * blockIdx.x + 1.0) - 1.0) * 0.5);
    while ((bI + 1) * (bI + 2) / 2 <= (int)blockIdx.x) bI++;
    while (bI * (bI + 1) / 2 > (int)blockIdx.x) bI--;
    const int bJ = (int)blockIdx.x - bI * (bI + 1) / 2;
    const int i0 = bI * 16, j0 = bJ * 16;
    const int t = threadIdx.x, r = t >> 4, c = t & 15;
    const int nbd = (n + 15) / 16;
    // partial traces of the previous launch: ONE coalesced load (thread b <- slot b), staged in LDS and added in index order
    // by everybody -- issued together with the panel loads so that the latencies overlap
    double trv = 0.0;
    if (!PLAN && !first && t < 2 * nbd) trv = trp[t];
    const bool ra = i0 + r < n, rb = j0 + r < n;
    const double *xa_row = Xp + (size_t)(i0 + r) * n, *xb_row = Xp + (size_t)(j0 + r) * n;
    const double *ya_row = X2p + (size_t)(i0 + r) * n, *yb_row = X2p + (size_t)(j0 + r) * n;
    const int mtot = kpad >> 4;
    bool sq = false;
    for (int m0 = 0; m0 < mtot; m0 += MAXM) {
        double xa[MAXM], ya[MAXM], xb[MAXM], yb[MAXM];
#pragma unroll
        for (int u = 0; u < MAXM; u++) {
            const int k = c + 16 * (m0 + u);
            const bool in = (m0 + u < mtot) && k < n;
            xa[u] = (in && ra) ? xa_row[k] : 0.0;
            xb[u] = (in && rb) ? xb_row[k] : 0.0;
            ya[u] = (in && ra && !first) ? ya_row[k] : 0.0;
            yb[u] = (in && rb && !first) ? yb_row[k] : 0.0;
        }
        if (!PLAN && m0 == 0 && !first) {
            red[t] = trv;
            __syncthreads();
            double tx = 0.0, tx2 = 0.0;
            for (int b = 0; b < nbd; b++) { tx += red[2 * b]; tx2 += red[2 * b + 1]; }
            sq = fabs(tx2 - target) < fabs(2.0 * tx - tx2 - target);
            __syncthreads();                 // `red` is reused by the K-split reduction below
        }
#pragma unroll
        for (int u = 0; u < MAXM; u++) {
            const int k = c + 16 * (m0 + u);
            if (m0 + u < mtot) {
                if (PLAN) {   // zero padding (k >= n, rows >= n) must stay zero: the identity term only on real diagonal elements
                    const double da = (ra && k < n && k == i0 + r) ? cf.c : 0.0, db = (rb && k < n && k == j0 + r) ? cf.c : 0.0;
                    Pa[r * ldp + k] = fma(cf.a, ya[u], fma(cf.b, xa[u], da));
                    Pb[r * ldp + k] = fma(cf.a, yb[u], fma(cf.b, xb[u], db));
                } else {
                    Pa[r * ldp + k] = first ? xa[u] : (sq ? ya[u] : 2.0 * xa[u] - ya[u]);
                    Pb[r * ldp + k] = first ? xb[u] : (sq ? yb[u] : 2.0 * xb[u] - yb[u]);
                }
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = kpad / 4; // k-range per wave (multiple of 4)
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double *pa = Pa + (lane & 15) * ldp + wave * kq + (lane >> 4);
    const double *pb = Pb + (lane & 15) * ldp + wave * kq + (lane >> 4);
    for (int k = 0; k < kq; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], pb[k], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) red[wave * 256 + q * 64 + lane] = acc[q];
    __syncthreads();
    if (wave == 0) {
        double tr2 = 0.0, tr1 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double v = (red[q * 64 + lane] + red[256 + q * 64 + lane]) + (red[512 + q * 64 + lane] + red[768 + q * 64 + lane]);
            int col = lane & 15, row = (lane >> 4) + 4 * q; // f64 MFMA C/D layout
            int gi = i0 + row, gj = j0 + col;
            if (gi < n && gj < n) {
                X2c[(size_t)gi * n + gj] = v;
                double xc = Pa[row * ldp + gj];
                const double osc = (PLAN && cf.out_scale != 0.0) ? cf.out_scale : 1.0;
                Xc[(size_t)gi * n + gj] = osc * xc;
                if (bI != bJ) {
                    X2c[(size_t)gj * n + gi] = v;
                    Xc[(size_t)gj * n + gi] = osc * Pb[col * ldp + gi];
                }
                if (gi == gj) { tr2 += v; tr1 += xc; }
            }
        }
        if (bI == bJ) {
            for (int o = 32; o > 0; o >>= 1) { tr1 += __shfl_xor(tr1, o); tr2 += __shfl_xor(tr2, o); }
            if (lane == 0) { trc[2 * bI] = tr1; trc[2 * bI + 1] = tr2; }
        }
    }
}

// Planned purification, one matrix per pass: the polynomial of the NEXT pass is applied in the epilogue of this one,
//   pass k:  reads X_k (pass 0: b_in F + c_in I formed while loading), computes the tile of X_k^2 and the traces of X_k, X_k^2,
//            writes X_{k+1} = a X_k^2 + b X_k + c I   (last pass: a = c = 0, b = out_scale: the result itself),
// so a pass loads two row panels and stores one tile where the two-matrix version loaded four and stored two.  Same triangular
// grid, same mirror stores, same MFMA/K-split order as sp2_fused_kernel: the traces are bit-identical to that kernel's.
__device__ inline void sp2_tile_of_block(int b, int &bI, int &bJ)
{
    bI = (int)((sqrt(8.0 * b + 1.0) - 1.0) * 0.5);
    while ((bI + 1) * (bI + 2) / 2 <= b) bI++;
    while (bI * (bI + 1) / 2 > b) bI--;
    bJ = b - bI * (bI + 1) / 2;
}
// one pass for the tile (bI, bJ); no __restrict__ on the matrices: the persistent kernel below ping-pongs between two buffers
#define SP2_LD(p) (COH ? __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *(p))
#define SP2_ST(p, v) do { if (COH) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(p) = (v); } while (0)
template <int MAXM, bool COH>
__device__ __forceinline__ void sp2_plan_pass(const double *Xin, int n, int kpad, double b_in, double c_in, double *Xout, double *trc,
                                              Sp2Coef nx, int bI, int bJ, double *lds)
{
    double *Pa = lds;                       // [16][kpad+SP2_PAD]  rows i0..i0+15 of X_k
    double *Pb = lds + 16 * (kpad + SP2_PAD);     // [16][kpad+SP2_PAD]  rows j0..j0+15 of X_k
    double *red = Pb + 16 * (kpad + SP2_PAD);     // [4][256]
    const int ldp = kpad + SP2_PAD;
    const int i0 = bI * 16, j0 = bJ * 16;
    const int t = threadIdx.x, r = t >> 4, c = t & 15;
    const bool ra = i0 + r < n, rb = j0 + r < n;
    const double *xa_row = Xin + (size_t)(i0 + r) * n, *xb_row = Xin + (size_t)(j0 + r) * n;
    const int mtot = kpad >> 4;
    for (int m0 = 0; m0 < mtot; m0 += MAXM) {
        double xa[MAXM], xb[MAXM];
#pragma unroll
        for (int u = 0; u < MAXM; u++) {
            const int k = c + 16 * (m0 + u);
            const bool in = (m0 + u < mtot) && k < n;
            xa[u] = (in && ra) ? SP2_LD(xa_row + k) : 0.0;
            xb[u] = (in && (bI != bJ) && rb) ? SP2_LD(xb_row + k) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < MAXM; u++) {
            const int k = c + 16 * (m0 + u);
            if (m0 + u < mtot) {   // zero padding (k >= n, rows >= n) must stay zero: the identity term only on real diagonal elements
                const double da = (ra && k < n && k == i0 + r) ? c_in : 0.0, db = (rb && k < n && k == j0 + r) ? c_in : 0.0;
                const double va = fma(b_in, xa[u], da);
                Pa[r * ldp + k] = va;
                Pb[r * ldp + k] = (bI != bJ) ? fma(b_in, xb[u], db) : va;
            }
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = kpad / 4; // k-range per wave (multiple of 4)
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    const double *pa = Pa + (lane & 15) * ldp + wave * kq + (lane >> 4);
    const double *pb = Pb + (lane & 15) * ldp + wave * kq + (lane >> 4);
    for (int k = 0; k < kq; k += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k], pb[k], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; q++) red[wave * 256 + q * 64 + lane] = acc[q];
    __syncthreads();
    if (wave == 0) {
        double tr2 = 0.0, tr1 = 0.0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            double v = (red[q * 64 + lane] + red[256 + q * 64 + lane]) + (red[512 + q * 64 + lane] + red[768 + q * 64 + lane]);
            int col = lane & 15, row = (lane >> 4) + 4 * q; // f64 MFMA C/D layout
            int gi = i0 + row, gj = j0 + col;
            if (gi < n && gj < n) {
                const double xc = Pa[row * ldp + gj];
                const double dg = (gi == gj) ? nx.c : 0.0;
                SP2_ST(Xout + (size_t)gi * n + gj, fma(nx.a, v, fma(nx.b, xc, dg)));
                if (bI != bJ) SP2_ST(Xout + (size_t)gj * n + gi, fma(nx.a, v, nx.b * Pb[col * ldp + gi]));
                if (gi == gj) { tr2 += v; tr1 += xc; }
            }
        }
        if (bI == bJ) {
            for (int o = 32; o > 0; o >>= 1) { tr1 += __shfl_xor(tr1, o); tr2 += __shfl_xor(tr2, o); }
            if (lane == 0) { trc[2 * bI] = tr1; trc[2 * bI + 1] = tr2; }
        }
    }
}
template <int MAXM>
__global__ __launch_bounds__(256) void sp2_plan_kernel(const double *__restrict__ Xin, int n, int kpad, double b_in, double c_in,
                                                       double *__restrict__ Xout, double *__restrict__ trc, Sp2Coef nx)
{
    extern __shared__ double lds[];
    int bI, bJ;
    sp2_tile_of_block((int)blockIdx.x, bI, bJ);
    sp2_plan_pass<MAXM, false>(Xin, n, kpad, b_in, c_in, Xout, trc, nx, bI, bJ, lds);
}

// The whole planned sequence in ONE launch (option sp2_persist, OFF by default): the grid of nb(nb+1)/2 <= #CU workgroups stays
// resident and the passes are separated by a grid barrier on a counter in device memory instead of a kernel boundary.  Same
// per-pass code (sp2_plan_pass), hence bit-identical matrices and traces (tests/test_gpu_sp2_persist.py).
//   COH = false: agent-scope release / acquire around the counter (buffer_wbl2 sc1 before the arrival, buffer_inv sc1 after the
//                wait: the L2 of each XCD is written back / invalidated, as at a kernel boundary);
//   COH = true : the matrix itself moves with agent-scope accesses (global_store ... sc1 writes through, global_load ... sc1
//                misses the XCD's L2), the counter with relaxed atomics: no cache maintenance at all.
// MEASURED (benzene, 19 passes, profiles/r02_sp2_resident_experiment.log): cc-pVTZ one launch per pass 140-149 us, COH=false
// 262 us, COH=true 192 us; cc-pVDZ 111 / 107 / 103 us.  A pass through the barrier costs four dependent fabric round trips
// (load miss, store acknowledgement, arrival atomic, poll) of ~2 us each, a kernel boundary 4.6 us in total: the 21 launches per
// cycle that VERDICT r01 item 4 wanted below 15 are the cheaper form, and the default stays one launch per pass.
//   bar[0]: arrival counter, never reset -- the host hands in the value it must reach (`base` + passes x workgroups, modulo
//   2^32); bar[1]: abort flag (= the launch's tag).  A workgroup that waits longer than SP2_BAR_TICKS (0.25 s of the 100 MHz clock: cannot happen
//   while the grid is co-resident, which the host checks) raises the flag, everybody leaves, and the traces of the last pass
//   are NaN -- the caller's validation then takes the diagonalisation path.
#define SP2_PLAN_MAXPASS 48
#define SP2_BAR_TICKS 25000000LL
struct Sp2PlanArg { int nit; double coef[3 * (SP2_PLAN_MAXPASS + 1)]; };
template <bool COH>
__device__ inline bool sp2_grid_barrier(unsigned *bar, unsigned target, unsigned tag)
{
    __shared__ int ok_sh;
    if (COH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through stores are acknowledged
    __syncthreads();                          // every store of this workgroup is issued and acknowledged
    if (threadIdx.x == 0) {
        int ok = 1;
        if (COH) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long t0 = wall_clock64();
        while ((int)((COH ? __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : __hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (__hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) { ok = 0; break; }
            if (wall_clock64() - t0 > SP2_BAR_TICKS) { __hip_atomic_store(bar + 1, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; break; }
        }
        ok_sh = ok;
    }
    __syncthreads();
    return ok_sh != 0;
}
template <int MAXM, bool COH>
__global__ __launch_bounds__(256) void sp2_plan_persist_kernel(const double *F, int n, int kpad, double *A, double *B, double osc,
                                                               double *tr, unsigned *bar, unsigned base, unsigned tag, Sp2PlanArg P)
{
    extern __shared__ double lds[];
    int bI, bJ;
    sp2_tile_of_block((int)blockIdx.x, bI, bJ);
    constexpr int TS = 2 * SP2_TRS;
    const double *in = F;
    double *out = A, *other = B;
    double b_in = P.coef[1], c_in = P.coef[2];
    bool ok = true;
    for (int k = 0; k <= P.nit; k++) {
        const Sp2Coef nx = k < P.nit ? Sp2Coef{P.coef[3 * (k + 1)], P.coef[3 * (k + 1) + 1], P.coef[3 * (k + 1) + 2], 0.0}
                                     : Sp2Coef{0.0, osc, 0.0, 0.0};
        sp2_plan_pass<MAXM, COH>(in, n, kpad, b_in, c_in, out, tr + TS * k, nx, bI, bJ, lds);
        if (k == P.nit) break;
        ok = sp2_grid_barrier<COH>(bar, base + (unsigned)(k + 1) * gridDim.x, tag);
        if (!ok) break;
        in = out;
        double *sw = out; out = other; other = sw;
        b_in = 1.0; c_in = 0.0;
    }
    if (!ok && bI == bJ && threadIdx.x == 0) {
        tr[TS * P.nit + 2 * bI] = __builtin_nan("");
        tr[TS * P.nit + 2 * bI + 1] = __builtin_nan("");
    }
}
typedef void (*sp2_plan_fn)(const double *, int, int, double, double, double *, double *, Sp2Coef);
typedef void (*sp2_persist_fn)(const double *, int, int, double *, double *, double, double *, unsigned *, unsigned, unsigned, Sp2PlanArg);
template <bool COH> static sp2_persist_fn sp2_persist_for_t(int kpad)
{
    const int m = kpad >> 4;
    if (m <= 8) return sp2_plan_persist_kernel<8, COH>;
    if (m <= 12) return sp2_plan_persist_kernel<12, COH>;
    if (m <= 16) return sp2_plan_persist_kernel<16, COH>;
    if (m <= 20) return sp2_plan_persist_kernel<20, COH>;
    return sp2_plan_persist_kernel<16, COH>; // two batches
}
static sp2_persist_fn sp2_persist_for(int kpad, bool coh) { return coh ? sp2_persist_for_t<true>(kpad) : sp2_persist_for_t<false>(kpad); }

static sp2_plan_fn sp2_plan_for(int kpad)
{
    const int m = kpad >> 4;
    if (m <= 8) return sp2_plan_kernel<8>;
    if (m <= 12) return sp2_plan_kernel<12>;
    if (m <= 16) return sp2_plan_kernel<16>;
    if (m <= 20) return sp2_plan_kernel<20>;
    return sp2_plan_kernel<16>; // two batches
}

typedef void (*sp2_fused_fn)(const double *, const double *, const double *, int, int, int, double, double *, double *, double *, Sp2Coef);
template <bool PLAN> static sp2_fused_fn sp2_fused_for_t(int kpad)
{
    const int m = kpad >> 4;
    if (m <= 8) return sp2_fused_kernel<8, PLAN>;
    if (m <= 12) return sp2_fused_kernel<12, PLAN>;
    if (m <= 16) return sp2_fused_kernel<16, PLAN>;
    if (m <= 20) return sp2_fused_kernel<20, PLAN>;
    return sp2_fused_kernel<16, PLAN>; // two batches
}
static sp2_fused_fn sp2_fused_for(int kpad) { return sp2_fused_for_t<false>(kpad); }

// d_X (in/out), d_X2 (in if have_x2, out), d_work: 2*n*n doubles, d_tr: (nit+2)*2*SP2_TRS doubles (device).
// On return d_X = X_nit, d_X2 = X_nit^2 and d_tr_out points at the 2*ceil(n/16) partial traces {tr X, tr X^2} interleaved
// (device pointer into d_tr; the caller adds them in index order).
extern "C" int mi_sp2_iterate(mi_ctx *c, double *d_X, double *d_X2, int nit, double n_occ, int have_x2, double *d_work,
                              double *d_tr, double **d_tr_out, void *stream)
{
    if (!c || !d_X || !d_X2 || !d_work || !d_tr || !d_tr_out || nit < 0) return fail("mi_sp2_iterate: bad argument");
    const int n = c->nao;
    if (n > 512) return fail("mi_sp2_iterate: fused path is for N <= 512");
    hipStream_t st = (hipStream_t)stream;
    const int kpad = ((n + 15) / 16) * 16;
    const size_t shm = sizeof(double) * (2 * 16 * (kpad + SP2_PAD) + 4 * 256);
    const int nb = (n + 15) / 16;
    dim3 grid(nb * (nb + 1) / 2), block(256);
    const size_t nn = (size_t)n * n;
    (void)have_x2; // X^2 and the traces of the incoming X are always (re)derived by the first pass
    constexpr int TS = 2 * SP2_TRS;
    const sp2_fused_fn sp2_fused_kernel = sp2_fused_for(kpad);
    double *cur_x = d_X, *cur_x2 = d_X2, *nxt_x = d_work, *nxt_x2 = d_work + nn;
    int slot = 0;
    hipLaunchKernelGGL(sp2_fused_kernel, grid, block, shm, st, cur_x, cur_x2, d_tr, 1, n, kpad, n_occ, nxt_x, nxt_x2, d_tr + TS * slot, Sp2Coef{});
    std::swap(cur_x, nxt_x); std::swap(cur_x2, nxt_x2);
    for (int it = 0; it < nit; it++) {
        hipLaunchKernelGGL(sp2_fused_kernel, grid, block, shm, st, cur_x, cur_x2, d_tr + TS * slot, 0, n, kpad, n_occ, nxt_x, nxt_x2,
                           d_tr + TS * (slot + 1), Sp2Coef{});
        slot++;
        std::swap(cur_x, nxt_x); std::swap(cur_x2, nxt_x2);
    }
    HIPCHK(hipGetLastError());
    if (cur_x != d_X) {
        HIPCHK(hipMemcpyAsync(d_X, cur_x, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(d_X2, cur_x2, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    }
    *d_tr_out = d_tr + TS * slot;
    return 0;
}

// Same passes on two caller-owned [X | X2] buffers (2 n^2 doubles each, X0 in d_A[0 : n^2]) WITHOUT the final copy:
// *d_res tells which of the two buffers holds {X_nit, X_nit^2}.
extern "C" int mi_sp2_iterate_pingpong(mi_ctx *c, double *d_A, double *d_B, int nit, double n_occ, double *d_tr, double **d_tr_out,
                                       double **d_res, void *stream)
{
    if (!c || !d_A || !d_B || !d_tr || !d_tr_out || !d_res || nit < 0) return fail("mi_sp2_iterate_pingpong: bad argument");
    const int n = c->nao;
    if (n > 512) return fail("mi_sp2_iterate_pingpong: fused path is for N <= 512");
    hipStream_t st = (hipStream_t)stream;
    const int kpad = ((n + 15) / 16) * 16;
    const size_t shm = sizeof(double) * (2 * 16 * (kpad + SP2_PAD) + 4 * 256);
    const int nb = (n + 15) / 16;
    dim3 grid(nb * (nb + 1) / 2), block(256);
    const size_t nn = (size_t)n * n;
    constexpr int TS = 2 * SP2_TRS;
    const sp2_fused_fn sp2_fused_kernel = sp2_fused_for(kpad);
    double *cur = d_A, *nxt = d_B;
    hipLaunchKernelGGL(sp2_fused_kernel, grid, block, shm, st, cur, cur + nn, d_tr, 1, n, kpad, n_occ, nxt, nxt + nn, d_tr, Sp2Coef{});
    std::swap(cur, nxt);
    for (int it = 0; it < nit; it++) {
        hipLaunchKernelGGL(sp2_fused_kernel, grid, block, shm, st, cur, cur + nn, d_tr + TS * it, 0, n, kpad, n_occ, nxt, nxt + nn,
                           d_tr + TS * (it + 1), Sp2Coef{});
        std::swap(cur, nxt);
    }
    HIPCHK(hipGetLastError());
    *d_tr_out = d_tr + TS * nit;
    *d_res = cur;
    return 0;
}

// Planned purification (round 2): the host knows inner bounds of the HOMO / LUMO and outer bounds of the spectrum (from the
// last diagonalisation) and fixes the whole sequence of quadratics in advance -- each step folds one band of the spectrum
// about a point inside it, (x - c)^2 or -(x - c)^2, and rescales to [0, 1]: about half the steps of trace-correcting SP2.
// Pass 0 maps the Fock matrix, X_0 = coef[1] F + coef[2] I; pass k = 1..nit applies X_k = a X_{k-1}^2 + b X_{k-1} + c I
// (coef[3k..3k+2]); every pass leaves the partial traces of X_k and X_k^2 (validation by the caller: tr(X - X^2), tr X).
// d_F is only read; d_A, d_B: two buffers of (at least) n^2 doubles; *d_res = the one holding out_scale * X_nit.
extern "C" int mi_sp2_iterate_planned(mi_ctx *c, const double *d_F, double *d_A, double *d_B, int nit, const double *coef, double out_scale,
                                      double *d_tr, double **d_tr_out, double **d_res, void *stream)
{
    if (!c || !d_F || !d_A || !d_B || !coef || !d_tr || !d_tr_out || !d_res || nit < 0) return fail("mi_sp2_iterate_planned: bad argument");
    const int n = c->nao;
    if (n > 512) return fail("mi_sp2_iterate_planned: fused path is for N <= 512");
    hipStream_t st = (hipStream_t)stream;
    const int kpad = ((n + 15) / 16) * 16;
    const size_t shm = sizeof(double) * (2 * 16 * (kpad + SP2_PAD) + 4 * 256);
    const int nb = (n + 15) / 16;
    dim3 grid(nb * (nb + 1) / 2), block(256);
    const size_t nn = (size_t)n * n;
    constexpr int TS = 2 * SP2_TRS;
    const sp2_plan_fn kern = sp2_plan_for(kpad);
    const double osc = out_scale != 0.0 ? out_scale : 1.0;
    auto next_coef = [&](int k) {   // polynomial applied in the epilogue of pass k: X_{k+1}, or the (scaled) result after the last pass
        return k < nit ? Sp2Coef{coef[3 * (k + 1)], coef[3 * (k + 1) + 1], coef[3 * (k + 1) + 2], 0.0} : Sp2Coef{0.0, osc, 0.0, 0.0};
    };
    double *cur = d_A, *nxt = d_B;
    if (c->opt_sp2_persist && nit >= 1 && nit <= SP2_PLAN_MAXPASS) {
        if (!c->n_cu) {
            HIPCHK(hipSetDevice(c->device));
            HIPCHK(hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, c->device));
        }
        // co-residency: one workgroup per CU (79 KB of LDS each), so the grid must not exceed the CU count
        if ((int)grid.x <= c->n_cu) {
            if (!c->d_sp2_bar) {
                HIPCHK(dev_malloc(&c->d_sp2_bar, 2 * sizeof(unsigned)));
                HIPCHK(hipMemsetAsync(c->d_sp2_bar, 0, 2 * sizeof(unsigned), st));
                c->sp2_bar_base = 0; c->sp2_tag = 0;
            }
            Sp2PlanArg P;
            P.nit = nit;
            for (int i = 0; i < 3 * (nit + 1); i++) P.coef[i] = coef[i];
            c->sp2_tag++;
            if (c->sp2_tag == 0) c->sp2_tag = 1;
            hipLaunchKernelGGL(sp2_persist_for(kpad, c->opt_sp2_persist == 2), grid, block, shm, st, d_F, n, kpad, d_A, d_B, osc, d_tr, c->d_sp2_bar, c->sp2_bar_base,
                               c->sp2_tag, P);
            HIPCHK(hipGetLastError());
            c->sp2_bar_base += (unsigned)nit * grid.x;
            *d_tr_out = d_tr + TS * nit;
            *d_res = (nit % 2 == 0) ? d_A : d_B;   // pass k writes A for even k, B for odd k
            return 0;
        }
    }
    hipLaunchKernelGGL(kern, grid, block, shm, st, d_F, n, kpad, coef[1], coef[2], cur, d_tr, next_coef(0));
    for (int it = 1; it <= nit; it++) {
        hipLaunchKernelGGL(kern, grid, block, shm, st, cur, n, kpad, 1.0, 0.0, nxt, d_tr + TS * it, next_coef(it));
        std::swap(cur, nxt);
    }
    HIPCHK(hipGetLastError());
    *d_tr_out = d_tr + TS * nit;
    *d_res = cur;
    return 0;
}

// =================================================================================================
// Fused elementwise pieces of the SCF cycle (fewer launches per cycle)
// =================================================================================================
// F = h + J - (kscale) K (+ Vxc), and part[block] = this block's share of sum D*(h + 0.5*(J - kscale K))  [one-electron +
// Coulomb/exchange energy].  Partial sums in a FIXED order instead of atomics: the replicated algebra of a sharded run must
// give bit-identical results on every rank (the caller adds the ceil(nn/256) partials in index order).
__global__ __launch_bounds__(256) void fock_energy_kernel(const double *h, const double *J, const double *K, const double *Vxc,
                                                          const double *D, double kscale, size_t nn, double *F, double *part)
{
    __shared__ double sh[4];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (idx < nn) {
        double v2 = J[idx] - (K ? kscale * K[idx] : 0.0);
        double f = h[idx] + v2 + (Vxc ? Vxc[idx] : 0.0);
        F[idx] = f;
        e = D[idx] * (h[idx] + 0.5 * v2);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int mi_reduce_blocks(const mi_ctx *c) { return c ? (int)(((size_t)c->nao * c->nao + 255) / 256) : -1; }

extern "C" int mi_fock_energy(mi_ctx *c, const double *d_h, const double *d_J, const double *d_K, const double *d_Vxc,
                              const double *d_D, double kscale, double *d_F, double *d_part, void *stream)
{
    if (!c || !d_h || !d_J || !d_D || !d_F || !d_part) return fail("mi_fock_energy: null argument");
    size_t nn = (size_t)c->nao * c->nao;
    hipLaunchKernelGGL(fock_energy_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_h, d_J, d_K, d_Vxc,
                       d_D, kscale, nn, d_F, d_part);
    HIPCHK(hipGetLastError());
    return 0;
}

// J/K digestion and Fock / energy assembly without the intermediate J and K matrices (single rank, resident tiles): the
// epilogue reads the padded accumulators directly,
//   J = 2 (Jacc + Jacc^T), K = Kacc + Kacc^T, F = h + J - kscale K (+ V + V^T for an UNsymmetrised XC matrix V),
//   part[block] = this block's share of sum D (h + 0.5 (J - kscale K))      (same fixed-order partials as fock_energy_kernel)
// -- one launch instead of finalize_jk_kernel + (V + V^T) + fock_energy_kernel, and 2 N^2 doubles less written and re-read.
__global__ __launch_bounds__(256) void finalize_fock_kernel(const double *Jacc, const double *Kacc, int ld, const double *h,
                                                            const double *Vun, const double *D, double kscale, int nao, double *F,
                                                            double *part, const int *__restrict__ perm)
{
    __shared__ double sh[4];
    const size_t nn = (size_t)nao * nao, idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (idx < nn) {
        const int r = (int)(idx / nao), c = (int)(idx - (size_t)r * nao);
        const int pr = perm[r], pc = perm[c];     // the accumulators are in the tile order of the ERI store
        const double j = 2.0 * (Jacc[(size_t)pr * ld + pc] + Jacc[(size_t)pc * ld + pr]);
        const double k = Kacc ? Kacc[(size_t)pr * ld + pc] + Kacc[(size_t)pc * ld + pr] : 0.0;
        const double v2 = j - (Kacc ? kscale * k : 0.0);
        const double hh = h[idx];
        F[idx] = hh + v2 + (Vun ? Vun[idx] + Vun[(size_t)c * nao + r] : 0.0);
        e = D[idx] * (hh + 0.5 * v2);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int mi_build_fock(mi_ctx *c, const double *d_D, const double *d_h, const double *d_Vun, int with_k, double kscale, double *d_F,
                             double *d_part, void *stream)
{
    if (!c || !d_D || !d_h || !d_F || !d_part) return fail("mi_build_fock: null argument");
    if (!c->eri_ready) return fail("mi_build_fock: call mi_eri_prepare first");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nn = (size_t)c->nao * c->nao, pp = (size_t)c->ldp * c->ldp;
    hipLaunchKernelGGL(pad_density_clear_kernel, dim3((unsigned)((pp + 255) / 256)), dim3(256), 0, st, d_D, c->d_Dpad, c->d_Jacc,
                       with_k ? c->d_Kacc : nullptr, c->nao, c->ldp, c->d_iperm);
    if (launch_jk(c, true, with_k != 0, st)) return -1;
    hipLaunchKernelGGL(finalize_fock_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, c->d_Jacc, with_k ? c->d_Kacc : nullptr,
                       c->ldp, d_h, d_Vun, d_D, kscale, c->nao, d_F, d_part, c->d_perm);
    HIPCHK(hipGetLastError());
    return 0;
}

// E = M - M^T and part[block] = this block's share of sum E^2 (fixed-order partials, see fock_energy_kernel)
__global__ __launch_bounds__(256) void commutator_norm_kernel(const double *M, int n, double *E, double *part)
{
    __shared__ double sh[4];
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0.0;
    if (idx < (size_t)n * n) {
        int r = (int)(idx / n), cidx = (int)(idx - (size_t)r * n);
        double v = M[idx] - M[(size_t)cidx * n + r];
        E[idx] = v;
        s = v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

extern "C" int mi_commutator_norm(mi_ctx *c, const double *d_M, double *d_E, double *d_part, void *stream)
{
    if (!c || !d_M || !d_E || !d_part) return fail("mi_commutator_norm: null argument");
    int n = c->nao;
    hipLaunchKernelGGL(commutator_norm_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_M, n, d_E,
                       d_part);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// Vxc accumulation  C[m][n] += sum_g A[m][g] * W[n][g]   (A = ao0, W = weighted AOs; K = grid dimension).
// rocBLAS runs this shape (tiny M,N ~ N_ao, K ~ 3e4) at < 1 TFLOP/s (no split-K: a handful of workgroups),
// so it is done here: 64x64 output tile per workgroup, grid.z splits the grid points, row panels staged in
// LDS (the contraction index is the contiguous one: coalesced loads), v_mfma_f64_16x16x4_f64, 2x2 MFMA tiles
// per wave, FP64 atomics to combine the splits.
// =================================================================================================
#define VM_T 64
#define VM_KS 64
#define VM_PAD 2 /* see SP2_PAD: conflict-free operand reads for 16 rows x 4 k */
__global__ __launch_bounds__(256) void xc_vmat_kernel(const double *__restrict__ A, const double *__restrict__ W, int nao, int64_t ng,
                                                      int64_t kchunk, double *C, int xcd_order)
{
    __shared__ double Pa[VM_T][VM_KS + VM_PAD], Pb[VM_T][VM_KS + VM_PAD];
    // 1-D grid of nt^2 x nsplit workgroups.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with an L2
    // of its own; with the natural order the nt tiles that share a row panel of one split land on different XCDs and every XCD
    // fetches that panel itself (nt = 5: 2.6 GB through the fabric per pass at N = 264).  XCD-aware order (nsplit a multiple of 8):
    // workgroup L belongs to XCD L % 8 and works on split (L % 8) + 8 (L / (8 nt^2)), tile (L / 8) % nt^2 -- all tiles of a split
    // run next to each other on one XCD and share its L2.
    const int nt = (nao + VM_T - 1) / VM_T, ntt = nt * nt;
    const int L = (int)blockIdx.x;
    int tile, split;
    if (xcd_order) { split = (L & 7) + 8 * (L / (8 * ntt)); tile = (L >> 3) % ntt; }
    else { tile = L % ntt; split = L / ntt; }
    const int m0 = (tile / nt) * VM_T, n0 = (tile % nt) * VM_T;
    const int64_t kbeg = (int64_t)split * kchunk, kend = min(ng, kbeg + kchunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qm = (wave >> 1) * 32, qn = (wave & 1) * 32;
    // which of this wave's 2 x 2 blocks of 16 rows / columns lie inside the matrix (wave-uniform)
    const bool va0 = m0 + qm < nao, va1 = m0 + qm + 16 < nao, vb0 = n0 + qn < nao, vb1 = n0 + qn + 16 < nao;
    d4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
    // register-staged software pipeline: the global loads of sub-chunk c+1 are in flight while sub-chunk c is
    // multiplied out of LDS
    constexpr int PER = VM_T * VM_KS / 256; // 16 elements of each panel per thread
    double ra[PER], rb[PER];
    auto gload = [&](int64_t k0) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            int idx = u * 256 + threadIdx.x;
            int r = idx / VM_KS, k = idx - r * VM_KS;
            int64_t g = k0 + k;
            ra[u] = (m0 + r < nao && g < kend) ? A[(size_t)(m0 + r) * ng + g] : 0.0;
            rb[u] = (n0 + r < nao && g < kend) ? W[(size_t)(n0 + r) * ng + g] : 0.0;
        }
    };
    if (kbeg < kend) gload(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += VM_KS) {
#pragma unroll
        for (int u = 0; u < PER; u++) {
            int idx = u * 256 + threadIdx.x;
            int r = idx / VM_KS, k = idx - r * VM_KS;
            Pa[r][k] = ra[u];
            Pb[r][k] = rb[u];
        }
        __syncthreads();
        if (k0 + VM_KS < kend) gload(k0 + VM_KS);
        if (va1 && vb1) {
#pragma unroll 4
            for (int kk = 0; kk < VM_KS; kk += 4) {
                const int kc = kk + (lane >> 4);
                double a0 = Pa[qm + (lane & 15)][kc], a1 = Pa[qm + 16 + (lane & 15)][kc];
                double b0 = Pb[qn + (lane & 15)][kc], b1 = Pb[qn + 16 + (lane & 15)][kc];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        } else if (va0 && vb0) {
            // edge tile (nao is not a multiple of 64): only the 16x16 blocks that reach into the matrix are multiplied, so the
            // mostly empty last tile row / column costs a half or a quarter of a full tile instead of the same
#pragma unroll 4
            for (int kk = 0; kk < VM_KS; kk += 4) {
                const int kc = kk + (lane >> 4);
                const double a0 = Pa[qm + (lane & 15)][kc], b0 = Pb[qn + (lane & 15)][kc];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                if (vb1) acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, Pb[qn + 16 + (lane & 15)][kc], acc[0][1], 0, 0, 0);
                if (va1) acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pa[qm + 16 + (lane & 15)][kc], b0, acc[1][0], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = m0 + qm + a * 16 + (lane >> 4) + 4 * r, col = n0 + qn + b * 16 + (lane & 15); // f64 MFMA C/D layout
                if (row < nao && col < nao) atomicAdd(&C[(size_t)row * nao + col], acc[a][b][r]);
            }
}

extern "C" int mi_xc_vmat(mi_ctx *c, const double *d_ao0, const double *d_aow, int64_t ng, double *d_vmat, void *stream)
{
    if (!c || !d_ao0 || !d_aow || !d_vmat) return fail("mi_xc_vmat: null argument");
    const int nt = (c->nao + VM_T - 1) / VM_T;
    // Split over the grid points: nt^2 x nsplit workgroups, two of which fit a CU (68 KB of LDS each), i.e. 512 run at a time.
    // Round 1 rounded nsplit UP to reach 1024 -- 1025 workgroups for benzene/cc-pVTZ, 1053 for ibuprofen/def2-TZVP: a third,
    // almost empty round.  Rounding DOWN (<= 1024: two full rounds): 0.563 -> 0.525 ms (123 k points, N = 264) and 0.900 ->
    // 0.778 ms (55 k points, N = 573; 46.7 TFLOP/s = 59 % of the FP64-MFMA peak); 768 / 1536 are worse, 2048 equal, 4096 better
    // for N = 264 (0.485) and worse for N = 573 (0.836: 130 MB of split atomics) -- tools/vmat_ab.py.  Other tile shapes
    // (64x64 / 128x64 / 64x128 / 128x128 outputs, 16-64 points per LDS stage) were within +-10 % of this kernel, none better on both.
    const int64_t wgs = c->opt_vmat_wgs > 0 ? c->opt_vmat_wgs : 1024;
    int64_t nsplit = c->opt_vmat_wgs >= 0 ? std::max<int64_t>(1, std::min<int64_t>((ng + 511) / 512, wgs / (nt * nt)))
                                          : std::max<int64_t>(1, std::min<int64_t>((ng + 511) / 512, (1024 + nt * nt - 1) / (nt * nt)));
    int64_t kchunk = ((ng + nsplit - 1) / nsplit + VM_KS - 1) / VM_KS * VM_KS;
    nsplit = (ng + kchunk - 1) / kchunk;
    int xcd_order = 0;
    const int64_t ns8 = nsplit / 8 * 8;       // the XCD-aware order needs a multiple of 8 splits: taken when that costs < 10 % of them
    if (c->opt_vmat_xcd && ns8 >= 8 && ns8 * 10 >= nsplit * 9) {
        nsplit = ns8;
        kchunk = ((ng + nsplit - 1) / nsplit + VM_KS - 1) / VM_KS * VM_KS;
        xcd_order = 1;
    }
    hipLaunchKernelGGL(xc_vmat_kernel, dim3((unsigned)(nt * nt * nsplit)), dim3(256), 0, (hipStream_t)stream, d_ao0, d_aow, c->nao, ng, kchunk,
                       d_vmat, xcd_order);
    HIPCHK(hipGetLastError());
    return 0;
}

// =================================================================================================
// Vxc accumulation with the weighted AOs formed ON THE FLY (round 3):  C[m][n] += sum_g ao_0[m][g] * W[n][g],
//   W[n][g] = sum_c wv_c[g] ao_c[n][g]   (c = 0 for LDA, 0..3 for GGA)  -- what xc_aow_kernel used to write out and xc_vmat_kernel
// to read back (a 1.3 GB pass of 0.24 ms per build on benzene/cc-pVTZ that existed only to feed this product).
// Tiling: a workgroup owns MT x 64 rows (up to ALL rows of the matrix) x 64 columns and a slice of the grid points, so the
// four AO components of the B operand are read ONCE per row block instead of once per 64-row tile, and a wave holds MT x 4
// MFMA tiles (16 MT rows x 64 columns): MT + 4 LDS operand reads feed 4 MT v_mfma_f64_16x16x4_f64 per k step (0.45 reads per
// MFMA at MT = 5 against 1.0 in xc_vmat_kernel, whose counters show 52 % issue stalls at 42 % of the MFMA peak).
// =================================================================================================
#define VF_KS 16
#define VF_PAD 2
#define VF_THREADS 256
// 256 threads = 4 waves, wave w owns the 16 MT rows of slab w and all 64 columns (MT x 4 MFMA tiles), 16-point chunks.
// MEASURED SLOWER than the xc_aow + xc_vmat pair it was meant to replace (benzene/cc-pVTZ, 123 k points; the pair: 0.24 + 0.51 ms):
//   this kernel, MT = 1 / 2 / 3 (two waves per SIMD)                         1.78 / 1.26 / 1.15 ms
//   MT = 5 (all 264 rows in one workgroup, 336 VGPRs, one wave per SIMD)        1.45 ms with 32-point chunks; forced to 256 VGPRs: 2.2 ms
//   MT = 5, 512 threads, double-buffered LDS stage, one workgroup per CU      0.88 ms (32.5 % of the MFMA peak by the MOPS counter)
// The four-component B operand costs 4 loads + 3 FMAs per element in the loader of a kernel whose limit is already the
// load -> LDS -> MFMA hand-over (xc_vmat_kernel: 52 % issue stalls), and the row blocks that would amortise it do not fit the
// registers of two waves per SIMD.  Kept as an option (`RKS.xc_vmat_fold`, default off), not used.
template <int MT>
__global__ __launch_bounds__(VF_THREADS) void xc_vmat_fold_kernel(const double *__restrict__ ao, const double *__restrict__ wv, int ncomp, int nao,
                                                                  int64_t ng, int64_t kchunk, int nct, int nrb, double *C, int xcd_order)
{
    extern __shared__ double lds_all[];
    constexpr int MR = MT * 64, LDS_LD = VF_KS + VF_PAD;
    double *Pa = lds_all, *Pb = lds_all + (size_t)MR * LDS_LD;
    const int ntt = nct * nrb;
    const int L = (int)blockIdx.x;
    int tile, split;
    if (xcd_order) { split = (L & 7) + 8 * (L / (8 * ntt)); tile = (L >> 3) % ntt; }
    else { tile = L % ntt; split = L / ntt; }
    const int rb = tile / nct, ct = tile - rb * nct;
    const int m0 = rb * MR, n0 = ct * 64;
    const int64_t kbeg = (int64_t)split * kchunk, kend = min(ng, kbeg + kchunk);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wr0 = wave * (MT * 16);
    const size_t comp = (size_t)nao * ng;
    d4_t acc[MT][4];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = d4_t{0.0, 0.0, 0.0, 0.0};
    constexpr int PA = MR * VF_KS / VF_THREADS, PB = 64 * VF_KS / VF_THREADS;   // 4 MT and 4 elements per thread
    const int kq = threadIdx.x & (VF_KS - 1), rq = threadIdx.x >> 4;           // element u of a thread: row rq + 16 u, point kq
    double ra[PA], rb_[PB];
    auto gload = [&](int64_t k0) {
        const int64_t g = k0 + kq;
        const bool gv = g < kend;
        double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
        if (gv) {
            w0 = wv[g];
            if (ncomp > 1) { w1 = wv[ng + g]; w2 = wv[2 * ng + g]; w3 = wv[3 * ng + g]; }
        }
        const double *pa = ao + (size_t)(m0 + rq) * ng + g;
#pragma unroll
        for (int u = 0; u < PA; u++) ra[u] = (gv && m0 + rq + 16 * u < nao) ? pa[(size_t)(16 * u) * ng] : 0.0;
        const double *pb = ao + (size_t)(n0 + rq) * ng + g;
#pragma unroll
        for (int u = 0; u < PB; u++) {
            double v = 0.0;
            if (gv && n0 + rq + 16 * u < nao) {
                const double *q = pb + (size_t)(16 * u) * ng;
                v = q[0] * w0;
                if (ncomp > 1) v = fma(q[comp], w1, fma(q[2 * comp], w2, fma(q[3 * comp], w3, v)));
            }
            rb_[u] = v;
        }
    };
    if (kbeg < kend) gload(kbeg);
    const bool rows_in = m0 + wr0 < nao;   // wave-uniform
    for (int64_t k0 = kbeg; k0 < kend; k0 += VF_KS) {
#pragma unroll
        for (int u = 0; u < PA; u++) Pa[(rq + 16 * u) * LDS_LD + kq] = ra[u];
#pragma unroll
        for (int u = 0; u < PB; u++) Pb[(rq + 16 * u) * LDS_LD + kq] = rb_[u];
        __syncthreads();
        if (k0 + VF_KS < kend) gload(k0 + VF_KS);       // in flight while this chunk is multiplied
        if (rows_in) {
#pragma unroll
            for (int kk = 0; kk < VF_KS; kk += 4) {
                const int kc = kk + (lane >> 4);
                double b[4];
#pragma unroll
                for (int j = 0; j < 4; j++) b[j] = Pb[(j * 16 + (lane & 15)) * LDS_LD + kc];
#pragma unroll
                for (int a = 0; a < MT; a++) {
                    const double av = Pa[(wr0 + a * 16 + (lane & 15)) * LDS_LD + kc];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[a][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[j], acc[a][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + wr0 + a * 16 + (lane >> 4) + 4 * r, col = n0 + j * 16 + (lane & 15);   // f64 MFMA C/D layout
                if (row < nao && col < nao) atomicAdd(&C[(size_t)row * nao + col], acc[a][j][r]);
            }
}

template <int MT>
static int launch_vmat_fold(mi_ctx *c, const double *d_ao, const double *d_wv, int ncomp, int64_t ng, int nrb, double *d_vmat, hipStream_t st)
{
    const int nct = (c->nao + 63) / 64, ntt = nct * nrb;
    const size_t shm = sizeof(double) * (size_t)(MT * 64 + 64) * (VF_KS + VF_PAD);
    // one workgroup per CU fits (LDS) at MT = 5: aim at one full round of workgroups, at least 512 points per split
    const int64_t wgs = c->opt_vmat_wgs > 0 ? c->opt_vmat_wgs : 512;   // two workgroups per CU
    int64_t nsplit = std::max<int64_t>(1, std::min<int64_t>((ng + 511) / 512, wgs / ntt));
    int64_t kchunk = ((ng + nsplit - 1) / nsplit + VF_KS - 1) / VF_KS * VF_KS;
    nsplit = (ng + kchunk - 1) / kchunk;
    int xcd_order = 0;
    const int64_t ns8 = nsplit / 8 * 8;
    if (c->opt_vmat_xcd && ns8 >= 8 && ns8 * 10 >= nsplit * 9) {
        nsplit = ns8;
        kchunk = ((ng + nsplit - 1) / nsplit + VF_KS - 1) / VF_KS * VF_KS;
        xcd_order = 1;
    }
    if (shm > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void *)xc_vmat_fold_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((xc_vmat_fold_kernel<MT>), dim3((unsigned)(ntt * nsplit)), dim3(VF_THREADS), shm, st, d_ao, d_wv, ncomp, c->nao, ng, kchunk, nct, nrb,
                       d_vmat, xcd_order);
    HIPCHK(hipGetLastError());
    return 0;
}

// d_ao: [ncomp >= 1 | 4][nao][ng] AO values (component 0 = values, 1..3 = gradient), d_wv: [1 | 4][ng] weights x potential
// (as mi_xc_aow takes them), gga = 0: LDA (component 0 only).  d_vmat += ao_0 . (sum_c wv_c ao_c)^T, unsymmetrised.
extern "C" int mi_xc_vmat_fold(mi_ctx *c, const double *d_ao, const double *d_wv, int64_t ng, int gga, double *d_vmat, void *stream)
{
    if (!c || !d_ao || !d_wv || !d_vmat) return fail("mi_xc_vmat_fold: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int nt = (c->nao + 63) / 64;                 // 64-row tiles of the matrix
    const int cap = std::max(1, std::min(5, c->opt_vmat_fold_mt));
    const int nrb = (nt + cap - 1) / cap;              // row blocks of at most `cap` 64-row tiles (registers: two waves per SIMD up to 3)
    const int mt = (nt + nrb - 1) / nrb;               // tiles per row block
    const int ncomp = gga ? 4 : 1;
    switch (mt) {
    case 1: return launch_vmat_fold<1>(c, d_ao, d_wv, ncomp, ng, nrb, d_vmat, st);
    case 2: return launch_vmat_fold<2>(c, d_ao, d_wv, ncomp, ng, nrb, d_vmat, st);
    case 3: return launch_vmat_fold<3>(c, d_ao, d_wv, ncomp, ng, nrb, d_vmat, st);
    case 4: return launch_vmat_fold<4>(c, d_ao, d_wv, ncomp, ng, nrb, d_vmat, st);
    default: return launch_vmat_fold<5>(c, d_ao, d_wv, ncomp, ng, nrb, d_vmat, st);
    }
}
