/*
 * mi355scf.h -- C ABI of the MI355X (gfx950) SCF Fock-build engine.
 *
 * Drop-in boundary.  The reference has no native plugin ABI: its scripts call
 * the Python surface of pyscf / gpu4pyscf, which in turn bind C libraries (libcint / libcvhf /
 * gpu4pyscf's libgvhf_rys, libgdft) through ctypes.  Each entry point below names the reference call
 * site that triggers it and the upstream C routine it stands in for [MEM = un-vendored upstream,
 * named from memory, not present under /root/reference].
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns 0 on success and a
 * negative code on failure (text via mi_last_error()); no exceptions cross the ABI.  `atm`, `bas`,
 * `env` follow the libcint layout (atm[natm][6], bas[nbas][8], env[]) that a PySCF-shaped Mole already
 * holds; only nctr == 1 shells with l <= 3 are accepted.  Pointers named d_* are DEVICE pointers
 * (e.g. torch.Tensor.data_ptr()) on the context's device; everything else is host memory.  `stream` is
 * a hipStream_t passed as void* (NULL = default stream).  All matrices are row-major [nao][nao] FP64.
 * A context is bound to one GPU and is not thread-safe; use one context per device/process.
 */
#ifndef MI355SCF_H
#define MI355SCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mi_ctx mi_ctx;

/* Last error text of the calling thread. */
const char *mi_last_error(void);

/* ABI version of this header/library pair. */
int mi_abi_version(void);

/* Create / destroy an engine context for one molecule+basis on GPU `device_id`.
 * Replaces: gto.Mole.build() handing _atm/_bas/_env to libcint (templates/calculate_energy.py:89-101;
 * templates/optimize_geometry.py:48-54). */
int mi_ctx_create(const int32_t *atm, int natm, const int32_t *bas, int nbas, const double *env,
                  int nenv, int device_id, mi_ctx **out);
void mi_ctx_destroy(mi_ctx *ctx);
int mi_ctx_nao(const mi_ctx *ctx);

/* A destroyed context parks its (possibly ~100 GB) tile store for reuse by the next context on the same
 * device (geometry optimisation); hand-over buffers, task lists and the pool of small device buffers are kept per device
 * as well.  This frees all of them.  (The library expects ONE host thread per device at a time: a second context working on
 * the same device from another thread gets private scratch buffers, but parked blocks are shared.) */
void mi_release_cache(void);

/* Tunables (project-defined, no reference counterpart).  Unknown keys are an error.
 * Effective at the next mi_eri_prepare:
 *   "runmax"      tiles per J/K work item (0 = auto)
 *   "jk_waves"    0 = one wave per work item, longest first; N > 0 = N waves with equal-cost contiguous shares
 *   "tri_tiles"   1 = block-diagonal tiles store triangular rows (default), 0 = full rows
 *   "xf_mfma_min" transform kernel: FP64-MFMA tiles for spherical blocks of at least this many elements
 *   "eri_tpq", "tpq_maxprim"  thread-per-quartet kernels for the low angular classes / their contraction-depth limit
 * Immediate:
 *   "jk_nt"       nontemporal loads for the tile stream (1 = when the tensor exceeds the Infinity Cache, 2 = always, 0 = never)
 *   "jk_cache_mb" MiB of tiles read with the default cache policy on such tensors (default 160)
 *   "jk_pipe"     half-tile software pipeline for full-row tiles (-1 auto)
 *   "jk_pair"     n_dm = 2: one pass with two waves per work item (-1 = for stores > 16 GB, 0 = one pass per density, 1 = always)
 *   "grad_dtol"   derivative quartets with q_ab q_cd max|G| below this are skipped (default 1e-13, 0 = Schwarz only)
 *   "grad_live"   1 (default): that test runs once per class pair and the derivative launches walk the compacted list of live
 *                 quartets; 0: every launch screens per wave
 *   "grad_rows", "grad_rows_min", "grad_rows_g32"  row gradient kernel for the mid / high classes (1 = on, default), for
 *                 derivative blocks of at least this many rows (20), two quartets per wave up to 64 rows (1)
 *   "vmat_xcd"    xc_vmat: XCD-aware workgroup order (1, default: the tiles of one split share an XCD's L2; 0: natural order)
 *   "vmat_wgs"    xc_vmat: workgroups aimed at by the split over the grid points (0 = 1024, two full rounds; -1 = round-1 formula)
 *   "sp2_persist" planned purification as ONE resident launch with grid barriers (0 = one launch per pass, default and faster;
 *                 1 = release/acquire fences, 2 = write-through stores + L2-bypassing loads) */
int mi_set_option(mi_ctx *ctx, const char *key, double value);

/* One-electron integrals into device buffers (any of them may be NULL): overlap S, kinetic T, nuclear
 * attraction V, dipole d_dip[3][nao][nao] about `origin` (host double[3], NULL = zero).
 * Replaces: libcint int1e_ovlp_sph / int1e_kin_sph / int1e_nuc_sph / int1e_r_sph [MEM], reached from
 * mf.kernel() -> get_ovlp/get_hcore (templates/calculate_energy.py:155,205) and mf.dip_moment
 * (templates/calculate_energy.py:253). */
int mi_int1e(mi_ctx *ctx, double *d_S, double *d_T, double *d_V, double *d_dip, const double *origin,
             void *stream);

/* Evaluate the Schwarz-screened symmetry-unique four-centre ERIs once with the Rys-quadrature kernels
 * and keep them resident in HBM as 8x8x8x8 AO tiles.  `tol` is the Schwarz threshold
 * (PySCF direct_scf_tol = 1e-13 [MEM]).  (rank, nranks) select this process's shard of the tile runs
 * (SURVEY.md section 8e); pass (0, 1) for a single GPU.
 * Replaces: libcint int2e_sph + libcvhf CVHFnr_int2e_q_cond [MEM] / gpu4pyscf libgvhf_rys [MEM],
 * reached from mf.kernel() -> get_jk (templates/calculate_energy.py:155; optimize_geometry.py:90). */
int mi_eri_prepare(mi_ctx *ctx, double tol, int rank, int nranks, void *stream);
/* mi_eri_prepare returns MI_ERR_NOMEM (instead of -1) when this rank's share of the tile store does not fit in free HBM;
 * the caller then shards over more GPUs or falls back to the direct mode.  mi_eri_get_memory reports the bytes the last
 * mi_eri_prepare needed and the free HBM it saw (valid after success and after MI_ERR_NOMEM). */
#define MI_ERR_NOMEM (-2)
int mi_eri_get_memory(const mi_ctx *ctx, int64_t *need_bytes, int64_t *free_bytes);
/* Drop the resident store of the last mi_eri_prepare (parked for reuse by the next one).  A sharded run whose ranks agree on
 * the direct mode (because SOME rank's shard did not fit) calls this on the ranks whose shard did fit. */
int mi_eri_release(mi_ctx *ctx);
/* Fresh device allocations of tile stores of at least `min_bytes` made by this process so far (re-use of the parked store of
 * a destroyed / released context does not count): every step of optimize(mf) (templates/optimize_geometry.py:99) should reuse
 * ONE allocation. */
int64_t mi_tile_store_allocations(int64_t min_bytes);

/* Sharding plan without a GPU or a context: the (J,K,L) tile runs that survive the block-pair Schwarz table
 * qblk[nblk(nblk+1)/2] (nblk = ceil(nao/8); entry I(I+1)/2+J = max Schwarz factor of the shell pairs touching AO blocks I,J)
 * are dealt to `nranks` ranks longest-processing-time first by streamed bytes -- the same plan mi_eri_prepare follows
 * ("deal cost-balanced batches").  Outputs: bytes and runs per rank. */
int mi_plan_shards(int nao, const double *qblk, double tol, int nranks, int64_t *bytes_per_rank, int64_t *runs_per_rank);

/* Density fitting (`mf.density_fit()`): three-index (ij|P) and two-index (P|Q) Coulomb
 * integrals over the auxiliary basis held by the context `aux` -- an ordinary context (mi_ctx_create on the auxiliary
 * atm/bas/env) whose LAST shell is the unit function (s primitive, exponent 0, coefficient sqrt(4 pi)).  Evaluated by the same
 * Rys kernels as the four-centre integrals ((ij|P 1) quartets).  d_int3c[nao][nao][naux], d_int2c[naux][naux], naux =
 * mi_ctx_nao(aux) - 1; either may be NULL.  Replaces libcint int3c2e_sph / int2c2e_sph behind pyscf.df [MEM]. */
int mi_df_build(mi_ctx *ctx, mi_ctx *aux, double *d_int3c, double *d_int2c, void *stream);

/* Nuclear gradient of the fitted two-electron energy: d_grad[natm][3] += sum_{ab,P} Z3[a][b][P] d(ab|P)/dX
 * + sum_{PQ} Z2[P][Q] d(P|Q)/dX, with the three-index density d_Z3[nao][nao][naux] (symmetric in ab) and the two-index density
 * d_Z2[naux][naux] (symmetric) formed by the caller from the fitted tensor (either may be NULL).  `aux` as in mi_df_build (same
 * atom list as ctx).  The derivative-integral batches are dealt round-robin to `nranks` callers (rank 0 of 1: everything);
 * the caller adds the partial gradients.  Replaces libcint int3c2e_ip1 / int3c2e_ip2 / int2c2e_ip1 behind
 * pyscf.df.grad.rhf.get_jk [MEM] (`mf.density_fit().nuc_grad_method()`; the reference never calls it). */
int mi_df_grad(mi_ctx *ctx, mi_ctx *aux, const double *d_Z3, const double *d_Z2, double *d_grad, int rank, int nranks, void *stream);

/* Schwarz factors of the last mi_eri_prepare: q[nbas][nbas] (host), q_ab = sqrt(max |(ab|ab)|), 0 for dropped pairs.
 * Replaces: libcvhf CVHFnr_int2e_q_cond [MEM]. */
int mi_schwarz_get(const mi_ctx *ctx, double *q);

/* Test/debug: one shell quartet (ish jsh|ksh lsh) read back from the resident tiles into host memory
 * out[2li+1][2lj+1][2lk+1][2ll+1]; 0 where Schwarz screening dropped the tile, NaN where the tile lives on another rank.
 * Lets the parity tests compare individual integrals with libcint-style int2e_sph shell blocks [MEM] (oracle: orc_eri_shell). */
int mi_eri_read_quartet(mi_ctx *ctx, int ish, int jsh, int ksh, int lsh, double *out);

/* Statistics of the resident ERI store. */
typedef struct {
    int64_t n_tiles;          /* tiles resident on this rank                                    */
    int64_t n_runs;           /* (J,K,L) runs                                                    */
    int64_t stored_bytes;     /* bytes of tile payload streamed per J/K build on this rank       */
    int64_t n_unique_eri;     /* symmetry-unique (i>=j,k>=l,ij>=kl) ERIs inside the resident tiles */
    int64_t n_quartets;       /* shell quartets evaluated                                       */
    double  seconds_eri;      /* wall time of the one-off ERI evaluation                        */
} mi_eri_stats;
int mi_eri_get_stats(const mi_ctx *ctx, mi_eri_stats *out);

/* Coulomb and exchange matrices from the resident ERIs for n_dm density matrices:
 *   J_ij = sum_kl (ij|kl) D_kl,   K_ik = sum_jl (ij|kl) D_jl.
 * d_D, d_J, d_K: [n_dm][nao][nao]; d_J or d_K may be NULL to skip that matrix.  On a sharded context
 * the result is this rank's partial sum; the caller all-reduces (RCCL) across ranks.
 * Replaces: libcvhf CVHFnr_direct_drv with CVHFnrs8_ji_s2kl / CVHFnrs8_li_s2kj [MEM] / gpu4pyscf
 * RYS_build_jk [MEM], reached from get_jk / get_veff inside mf.kernel(). */
int mi_build_jk(mi_ctx *ctx, const double *d_D, int n_dm, double *d_J, double *d_K, void *stream);

/* Dense [nao^4] copy of the resident ERIs (chemists' notation (ij|kl), all eight symmetry images) for post-SCF methods on
 * small molecules: `mp.MP2(mf).kernel()` in templates/calculate_interaction.py:116-120.  Unsharded contexts only. */
int mi_eri_unpack(mi_ctx *ctx, double *d_out, void *stream);

/* Time `reps` back-to-back launches of the J/K digestion kernel alone with HIP events on `stream`
 * and return the average milliseconds per launch (bench.py's roofline leg). */
int mi_time_jk_kernel(mi_ctx *ctx, const double *d_D, int reps, double *ms_per_launch, void *stream);
/* Same measurement for the J-only (with_k = 0: the pure-functional RKS build) or K-only kernel variant. */
int mi_time_jk_variant(mi_ctx *ctx, const double *d_D, int with_j, int with_k, int reps, double *ms_per_launch,
                       void *stream);

/* DIIS (Pulay) helpers on device: given F, D, S form e = S D F - F D S ... */
/* err = A - A^T written in place of nothing: d_err[nao*nao] = d_SDF - d_SDF^T (fused epilogue). */
int mi_diis_errvec(mi_ctx *ctx, const double *d_SDF, double *d_err, void *stream);
/* d_out = sum_i coef[i] * d_hist[i]  (hist: n matrices of nao*nao, contiguous). coef on host. */
int mi_diis_combine(mi_ctx *ctx, const double *d_hist, const double *coef, int n, double *d_out,
                    void *stream);
/* Gram matrix row: out[i] = <d_hist_e[i], d_e> for i < n (host output, synchronises `stream`). */
int mi_diis_dots(mi_ctx *ctx, const double *d_hist_e, const double *d_e, int n, double *out,
                 void *stream);
/* Same, result left on the device (d_out[n]), no synchronisation: lets the SCF step fetch the Gram row,
 * the energy and the gradient norm with ONE device-to-host copy. */
int mi_diis_dots_dev(mi_ctx *ctx, const double *d_hist_e, const double *d_e, int n, double *d_out,
                     void *stream);
/* Round 2: the Pulay solve itself on the device, so that a cycle's extrapolation needs no host round trip (scf.diis.CDIIS ->
 * lib.diis.DIIS.extrapolate: numpy.linalg.solve of the (m+1)x(m+1) B-matrix system).  d_part = the [m][16] partials
 * mi_diis_dots_dev just wrote for the error vector in history slot `slot`; d_B [space][space] is the device-resident Gram
 * matrix (row/column `slot` are replaced), d_coef[0..m) receives the coefficients (no extrapolation, c = e_slot, when the
 * system is singular).  mi_diis_combine_dev = mi_diis_combine with the coefficients read from device memory. */
int mi_diis_solve(mi_ctx *ctx, const double *d_part, int m, int slot, int space, double *d_B, double *d_coef, void *stream);
int mi_diis_combine_dev(mi_ctx *ctx, const double *d_hist, const double *d_coef, int n, double *d_out, void *stream);
/* The same two with an explicit vector length `len` (doubles per history entry) instead of nao^2: the stacked (F_alpha, F_beta)
 * pair of UHF / UKS (scf.diis.CDIIS on the spin-stacked Fock matrix, templates/calculate_bde.py:126-128) has len = 2 nao^2. */
int mi_diis_dots_dev_n(mi_ctx *ctx, const double *d_hist_e, const double *d_e, int n, int64_t len, double *d_out, void *stream);
int mi_diis_combine_dev_n(mi_ctx *ctx, const double *d_hist, const double *d_coef, int n, int64_t len, double *d_out, void *stream);


/* ---- density from the Fock matrix without diagonalisation (row a11) ----------------------------- */
/* SP2 purification (Niklasson 2002) in an orthonormal basis; the X*X products are the caller's DGEMMs.
 * mi_sp2_init: X0 = (emax*I - F)/(emax - emin) with Gershgorin bounds; d_work: >= 2*n doubles.
 * mi_sp2_update: given X and X2 = X*X writes {tr X, tr X2, X_next[n*n]} to d_out_with_traces, where
 * X_next = X2 if |tr X2 - n_occ| < |2 tr X - tr X2 - n_occ| else 2X - X2.
 * Stand in for the LAPACK eig inside PySCF's SCF.eig (templates/calculate_energy.py:155 -> kernel()). */
int mi_sp2_init(mi_ctx *ctx, const double *d_F, double *d_X, double *d_work, void *stream);
int mi_sp2_update(mi_ctx *ctx, double *d_X, const double *d_X2, double n_occ, double *d_out_with_traces,
                  void *stream);

/* Fused SP2 for N <= 512: one launch per step (X*X on v_mfma_f64_16x16x4_f64 + branch + update + traces).
 * d_X holds X0 (or a previous iterate), d_X2 receives X^2; d_work: 2*n*n doubles; d_tr: 64*(nit+2) doubles.
 * Runs one squaring pass plus `nit` update+square passes; *d_tr_out points at ceil(n/16) interleaved PARTIAL traces
 * {tr X, tr X^2} (device); their sums in index order are the traces -- fixed-order partial sums instead of atomics, so that
 * the replicated algebra of a sharded run is bit-identical on every rank (no control-scalar broadcast needed). */
int mi_sp2_iterate(mi_ctx *ctx, double *d_X, double *d_X2, int nit, double n_occ, int have_x2,
                   double *d_work, double *d_tr, double **d_tr_out, void *stream);
/* Same passes on two caller-owned [X | X2] buffers without the final copy; *d_res = the buffer holding the result. */
int mi_sp2_iterate_pingpong(mi_ctx *ctx, double *d_A, double *d_B, int nit, double n_occ, double *d_tr, double **d_tr_out,
                            double **d_res, void *stream);

/* Planned purification: the whole sequence of quadratics is fixed by the caller from bounds of the spectrum (outer) and of
 * the HOMO / LUMO (inner) -- see mi355scf/sp2plan.py.  coef[3*(nit+1)]: pass 0 forms X_0 = coef[1] F + coef[2] I from the
 * (orthonormal-basis) Fock matrix d_F, pass k applies X_k = coef[3k] X^2 + coef[3k+1] X + coef[3k+2] I.  Buffers and trace
 * of every pass as in mi_sp2_iterate_pingpong (the caller validates tr(X - X^2) and tr X of the last pass).  One matrix per
 * pass: d_A, d_B need n^2 doubles each, *d_res = the one that holds the result out_scale * X_nit (2 = closed-shell density
 * matrix in the orthonormal basis; 0 or 1 = the projector itself; X_nit^2 is not stored); the traces are those of the
 * unscaled X. */
int mi_sp2_iterate_planned(mi_ctx *ctx, const double *d_F, double *d_A, double *d_B, int nit, const double *coef, double out_scale,
                           double *d_tr, double **d_tr_out, double **d_res, void *stream);

/* Fused elementwise pieces of one SCF cycle (rows a11/a12: get_fock + energy_elec, orbital-gradient norm):
 * mi_fock_energy: F = h + J - kscale*K (+Vxc); d_part[b] = block b's share of sum D*(h + (J - kscale*K)/2).  d_K, d_Vxc may be NULL.
 * mi_commutator_norm: E = M - M^T; d_part[b] = block b's share of |E|_F^2.
 * d_part has mi_reduce_blocks(ctx) = ceil(nao^2/256) entries; the caller adds them in index order (deterministic). */
int mi_reduce_blocks(const mi_ctx *ctx);
int mi_fock_energy(mi_ctx *ctx, const double *d_h, const double *d_J, const double *d_K, const double *d_Vxc,
                   const double *d_D, double kscale, double *d_F, double *d_part, void *stream);
int mi_commutator_norm(mi_ctx *ctx, const double *d_M, double *d_E, double *d_part, void *stream);

/* ---- DFT -------------------------------------------------------------- */

/* Becke fuzzy-cell weights for `ng` atom-centred grid points: d_coords[ng][3] (Bohr), d_atom_of[ng]
 * (owning atom), d_vol[ng] (radial x angular volume element), d_adjust[natm][natm] (Treutler radius
 * adjustment a_ij, device) -> d_weights[ng].  Replaces libdft VXCgen_grid [MEM], reached from
 * dft.RKS(mol) -> Grids.build (templates/calculate_energy.py:148,163; optimize_geometry.py:72,86). */
int mi_grid_becke(mi_ctx *ctx, const double *d_coords, const int32_t *d_atom_of, const double *d_vol,
                  int64_t ng, const double *d_adjust, double *d_weights, void *stream);

/* AO values (deriv=0), +gradient (deriv=1), +second derivatives xx,xy,xz,yy,yz,zz (deriv=2, for the XC
 * nuclear gradient) on grid points: d_ao[(1|4|10)][nao][ng].
 * Replaces libdft GTOval_sph_deriv1 / gpu4pyscf GDFTeval_gto [MEM] (numint.eval_ao). */
int mi_eval_ao(mi_ctx *ctx, const double *d_coords, int64_t ng, int deriv, double *d_ao, void *stream);

/* rho and grad rho from d_C = D @ ao0 ([nao][ng]): d_rho[(1|4)][ng].  (numint.eval_rho [MEM]) */
int mi_xc_rho(mi_ctx *ctx, const double *d_ao, const double *d_C, int64_t ng, int deriv, double *d_rho,
              void *stream);
/* Round 2: the same from occupied-orbital values d_psi[(1|4)][nocc][ng] = Z^T ao of a factorised density D = Z Z^T
 * (numint.eval_rho2 [MEM], what PySCF uses when mo_coeff / mo_occ are known): rho = sum psi^2, grad rho = 2 sum psi grad psi,
 * and (d_tau non-null, deriv = 1) the kinetic-energy density tau = 1/2 sum |grad psi|^2 of the meta-GGAs. */
int mi_xc_rho_mo(mi_ctx *ctx, const double *d_psi, int nocc, int64_t ng, int deriv, double *d_rho, double *d_tau, void *stream);
/* Fock matrix and energy partial sums straight from the J/K accumulators (single rank, resident tiles; the matrices J and K are
 * never formed): d_F = d_h + J - kscale K (+ d_Vun + d_Vun^T when d_Vun, an UNsymmetrised XC matrix, is given); d_part as in
 * mi_fock_energy.  with_k = 0: Coulomb only.  Same kernels as mi_build_jk for the digestion.
 * Replaces (with mi_build_jk / mi_fock_energy): get_veff + get_fock of the SCF classes, reached from mf.kernel(),
 * templates/calculate_energy.py:205. */
int mi_build_fock(mi_ctx *ctx, const double *d_D, const double *d_h, const double *d_Vun, int with_k, double kscale, double *d_F,
                  double *d_part, void *stream);
/* Warm start of the low-rank factor of the projector (project-defined; no reference counterpart): d_G[n][nocc] =
 * good ? 0.05 d_G0 + scale d_Zt^T : d_G0, good = (all of d_Zt[nocc][n] finite) && *d_info == 0, scale = 1 / |row 0 of d_Zt|. */
int mi_nystrom_warm(mi_ctx *ctx, const double *d_Zt, const int *d_info, const double *d_G0, double *d_G, int nocc, int n,
                    void *stream);
/* d_Zt[nocc][n] = R^-1 d_W^T with d_M[nocc][nocc] = R R^T (lower Cholesky factor), d_W[n][nocc]; *d_info = 0 or the 1-based index
 * of the first bad pivot (then d_Zt is NaN).  nocc <= 64.  Stands in for torch.linalg.cholesky_ex + solve_triangular. */
int mi_nystrom_factor(mi_ctx *ctx, const double *d_M, const double *d_W, int n, int nocc, double *d_Zt, int *d_info, void *stream);
/* d_tail[q] += sum_g d_w[g] d_vq[g] for the non-NULL d_v0..d_v2 (quadrature sums N_elec, E_xc of one grid block), one launch,
 * fixed summation order.  Stands in for the numpy dots at the end of numint.nr_rks / nr_uks [MEM]. */
int mi_xc_tail(mi_ctx *ctx, const double *d_w, const double *d_v0, const double *d_v1, const double *d_v2, int64_t ng, double *d_tail,
               void *stream);
/* Both steps fused (psi never stored): d_Zp[nao][ldz] = Z with the orbital index fastest, zero-padded to ldz = a multiple of
 * 24 (deriv = 1) or 32 (deriv = 0) columns; d_ao as in mi_xc_rho. */
int mi_xc_rho_lowrank(mi_ctx *ctx, const double *d_ao, const double *d_Zp, int ldz, int64_t ng, int deriv, double *d_rho,
                      double *d_tau, void *stream);

/* Closed-shell XC energy density and potential on the grid.  kinds[]: 1 Slater, 2 B88, 3 VWN-RPA,
 * 4 VWN5, 5 LYP, 6 PBE-x, 7 PBE-c with weights coefs[].  Outputs (any may be NULL): d_exc[ng] energy
 * per volume; d_wv[(1|4)][ng] = {w*vrho/2, 2*w*vsigma*grad rho}; d_vrho, d_vsigma raw derivatives.
 * Replaces libxc (HYB_GGA_XC_B3LYP etc.) reached through mf.xc (templates/calculate_energy.py:149). */
int mi_xc_eval(const int32_t *kinds, const double *coefs, int nterms, const double *d_rho,
               const double *d_w, int64_t ng, int gga, double *d_exc, double *d_wv, double *d_vrho,
               double *d_vsigma, void *stream);
/* Spin-polarised form for UKS (reference call sites templates/calculate_bde.py:128,140,197,215): rho_s = [rho, grad rho]
 * of spin s; wv_s feed mi_xc_aow / mi_xc_vmat exactly like the closed-shell wv and give V_xc of spin s. */
int mi_xc_eval_spin(const int32_t *kinds, const double *coefs, int nterms, const double *d_rhoa, const double *d_rhob,
                    const double *d_w, int64_t ng, int gga, double *d_exc, double *d_wva, double *d_wvb, void *stream);

/* d_aow[nao][ng] = sum_c d_ao[c] * d_wv[c]; Vxc = ao0 @ aow^T + transpose is then one DGEMM. */
int mi_xc_aow(mi_ctx *ctx, const double *d_ao, const double *d_wv, int64_t ng, int gga, double *d_aow,
              void *stream);

/* ---- analytic nuclear gradient ---------------------------------------------- */
/* d_grad[natm][3] += 2 sum D <d mu|T+V|nu> - 2 sum W <d mu|nu> + Hellmann-Feynman terms.  d_W is the
 * energy-weighted density (D F D / 2).  Replaces libcint int1e_ipovlp/ipkin/ipnuc/iprinv [MEM], reached
 * through optimize(mf) -> mf.nuc_grad_method() (templates/optimize_geometry.py:99). */
int mi_grad_1e(mi_ctx *ctx, const double *d_D, const double *d_W, double *d_grad, void *stream);

/* d_grad[natm][3] += dE2/dR at fixed density: 2 sum (d mu nu|lam sig) [D D - hyb/4 (D D + D D)] over the
 * Schwarz-surviving quartets, derivative ERIs by the same Rys kernel on (l+1)/(l-1) auxiliary shells.
 * Replaces libcint int2e_ip1 + libcvhf nrs2/nrs4 J/K gradient contractions / gpu4pyscf rys gradient
 * kernels [MEM] (mf.nuc_grad_method().get_jk).  On a sharded context the quartet batches are dealt
 * round-robin to ranks: the result is a partial sum to be all-reduced by the caller. */
int mi_grad_eri(mi_ctx *ctx, const double *d_D, double hyb, double *d_grad, void *stream);
/* Open-shell form (UHF/UKS, templates/calculate_bde.py:224 optimises radicals): d_D = Da + Db, d_Dspin = Da - Db
 * (NULL: closed shell); the exchange part contracts sum_s Ds x Ds = (D x D + M x M) / 2. */
int mi_grad_eri_spin(mi_ctx *ctx, const double *d_D, const double *d_Dspin, double hyb, double *d_grad, void *stream);
/* Same with the share of the derivative-quartet batches given explicitly (the two entry points above use the
 * (rank, nranks) of the last mi_eri_prepare, which in direct mode is a tile GROUP index, not the process's rank). */
int mi_grad_eri_sharded(mi_ctx *ctx, const double *d_D, const double *d_Dspin, double hyb, double *d_grad, int rank,
                        int nranks, void *stream);

/* meta-GGA evaluation (functional ids 8 TPSS exchange, 9 TPSS correlation, 10 M06-2X exchange, 11 M06-2X correlation; ids
 * 1-7 may be mixed in): d_tau[ng] = 1/2 sum_i |grad phi_i|^2; d_wv[5][ng] as mi_xc_eval plus d_wv[4] = w/4 de/dtau (the caller
 * adds sum_k ao_k^T (wv4 ao_k) to the unsymmetrised V_xc).  `--method M06-2X` at templates/calculate_energy.py:263;
 * default functional of templates/calculate_bde.py:105.  Stands in for libxc MGGA_X/C_TPSS, HYB_MGGA_X_M06_2X, MGGA_C_M06_2X
 * [MEM]; the M06-2X parameter tables are entered from memory (unverified). */
int mi_xc_eval_mgga(const int32_t *kinds, const double *coefs, int nterms, const double *d_rho, const double *d_tau,
                    const double *d_w, int64_t ng, double *d_exc, double *d_wv, void *stream);
/* Spin-polarised form (UKS): per spin rho_s[4][ng], tau_s[ng] -> wv_s[5][ng]. */
int mi_xc_eval_mgga_spin(const int32_t *kinds, const double *coefs, int nterms, const double *d_rhoa, const double *d_rhob,
                         const double *d_taua, const double *d_taub, const double *d_w, int64_t ng, double *d_exc, double *d_wva,
                         double *d_wvb, void *stream);

/* d_vmat[nao][nao] += ao0 . aow^T over the grid block (split-K FP64 MFMA kernel; rocBLAS has no split-K for
 * this tiny-M,N / huge-K shape and runs it at < 1 TFLOP/s).  The caller symmetrises (Vxc = vmat + vmat^T). */
int mi_xc_vmat(mi_ctx *ctx, const double *d_ao0, const double *d_aow, int64_t ng, double *d_vmat, void *stream);
/* Round 3: the same accumulation with the weighted AOs formed on the fly from the AO values and the weights x potential that
 * mi_xc_aow would take (d_ao [1|4][nao][ng], d_wv [1|4][ng], gga = 0: LDA): d_vmat += ao_0 . (sum_c wv_c ao_c)^T.  Replaces the
 * mi_xc_aow + mi_xc_vmat pair of numint.nr_rks's `_scale_ao` + `ao.T @ aow` [MEM] for LDA / GGA functionals. */
int mi_xc_vmat_fold(mi_ctx *ctx, const double *d_ao, const double *d_wv, int64_t ng, int gga, double *d_vmat, void *stream);

/* Real-solid-harmonic coefficient table used by the kernels: out[ncart(l)][2l+1] (host). */
int mi_c2s_table(int l, double *out);

/* Rys roots/weights as evaluated by the device tables, for tests: roots[n], weights[n] (host). */
int mi_rys_roots_host(int nroots, double x, double *roots, double *weights);

#ifdef __cplusplus
}
#endif
#endif /* MI355SCF_H */
