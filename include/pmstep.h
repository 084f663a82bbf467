/*
 * pmstep.h -- C ABI of libpmstep.so, the MI355X (gfx950) particle-mesh stepper.
 *
 * Drop-in boundary for PySCo's per-step hot path.  PySCo has no FFI of its own: its "native"
 * layer is Numba-jitted Python, so each entry point below names the Python function
 * (file:line under pysco/) whose body it replaces.  INTEGRATION.md shows the ctypes stubs a
 * PySCo maintainer would add to call these from mesh.py / fourier.py / utils.py / solver.py.
 *
 * Conventions
 *  - every pointer named d_* is a DEVICE pointer (hipMalloc'd by the caller, e.g. a torch tensor's
 *    data_ptr()); h_* are host pointers.  No ownership is transferred; nothing is retained
 *    after the call returns except inside pm_ctx.
 *  - all grids are C-contiguous fp32 [n,n,n] (k fastest), vector grids [n,n,n,3] (AoS, as
 *    mesh.derivative returns), particle arrays [np,3] fp32, half-spectra [n,n,n/2+1] complex64.
 *  - n is a power of two >= 4 (2**ncoarse, solver.py:75); cell indices fit int16 in the reference
 *    (mesh.py:2506) so n <= 16384 here.
 *  - launches are asynchronous on the context's stream unless the function returns a host value.
 *  - return value: 0 on success, negative on error (pm_last_error() gives the text):
 *      -1 invalid argument, -2 HIP runtime error, -3 rocFFT error, -4 unsupported.
 */
#ifndef PMSTEP_H
#define PMSTEP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pm_ctx pm_ctx;

/* mass-assignment schemes: values are the reference's MAS_index (solver.py:82,85) */
enum { PM_NGP = 1, PM_CIC = 2, PM_TSC = 3 };
/* k-space Green's functions (fourier.py:460, 502, 555) */
enum { PM_GREEN_PLAIN = 0, PM_GREEN_COMPENSATED = 1, PM_GREEN_7PT = 2 };

int pm_version(void);
const char *pm_last_error(void);

/* Context: grid size, stream, rocFFT plans + work buffer, small device scratch.  `stream` is a
 * hipStream_t (NULL = default stream).  One context per (process, grid size). */
int pm_ctx_create(pm_ctx **out, int n, void *stream);
int pm_ctx_destroy(pm_ctx *ctx);
int pm_ctx_set_stream(pm_ctx *ctx, void *stream);
int pm_ctx_sync(pm_ctx *ctx);
/* Device-side timing on the context's stream (hipEvent); used by bench.py for the roofline. */
int pm_timer_start(pm_ctx *ctx);
int pm_timer_stop_ms(pm_ctx *ctx, float *h_ms); /* synchronises */

/* ---- mesh.py ------------------------------------------------------------------------------ */
/* mesh.NGP :2240 / mesh.CIC :2284 / mesh.TSC :2468 (and TSC_seq :2363): d_rho is overwritten.
 * `sorted_hint` picks the kernel, never the result: 0 = one global atomic per contribution (any
 * order), 1 = one workgroup per LDS tile over the tile's range of a Morton-sorted stream
 * (utils.reorder_particles order; drifted / unsorted particles fall back per particle),
 * 2 = one workgroup per 2048-particle chunk with a re-anchored LDS tile. */
int pm_deposit(pm_ctx *ctx, int scheme, const float *d_pos, int64_t np, float *d_rho, int sorted_hint);
/* mesh.invNGP_vec :2639 / invCIC_vec :2752 / invTSC_vec :2960  (ncomp = 3, grid [n,n,n,3])
 * mesh.invNGP :2600 / invCIC :2679 / invTSC :2827             (ncomp = 1, grid [n,n,n])      */
int pm_interp(pm_ctx *ctx, int scheme, int ncomp, const float *d_grid, const float *d_pos, int64_t np,
              float *d_out);
/* mesh.derivative :2072 -> derivative2/3/5/7 :639-850.  d_force [n,n,n,3] = +grad(phi). */
int pm_gradient(pm_ctx *ctx, int order, const float *d_phi, float *d_force);
/* mesh.derivative_fR :2112 -> derivativeK_fR_n1/n2 :860-2069: grad(phi) + f*grad(u**(fr_n+1)). */
int pm_gradient_fr(pm_ctx *ctx, int order, int fr_n, float f, const float *d_phi, const float *d_u,
                   float *d_force);
/* pm_gradient_fr from 128^3 walks x with the stencil's planes in registers, like pm_gradient (same bits); 0: the row kernel */
int pm_set_fr_march_gradient(pm_ctx *ctx, int on);

/* ---- utils.py ----------------------------------------------------------------------------- */
int pm_axpy(pm_ctx *ctx, float *d_y, const float *d_x, float a, int64_t count);              /* :264 */
int pm_scale(pm_ctx *ctx, float *d_y, float a, int64_t count);                                /* :410 */
/* The memory traffic of the fused kick + drift of np particles with every value left as it is (two [np, 3] arrays
 * rewritten in place, a third read): the probe main.tune_placement times over combinations of allocations.  No
 * reference counterpart. */
int pm_probe_update(pm_ctx *ctx, float *d_a, float *d_b, const float *d_c, int64_t np);
int pm_linear_operator(pm_ctx *ctx, float *d_out, const float *d_x, float f1, float f2,
                       int64_t count);                                                        /* :644, :684 */
int pm_linear_operator_vectors(pm_ctx *ctx, float *d_x, float f1, const float *d_y, float f2,
                               int64_t count);                                                /* :721 */
int pm_max_abs(pm_ctx *ctx, const float *d_x, int64_t count, float *h_out);                   /* :220 (syncs) */
int pm_periodic_wrap(pm_ctx *ctx, float *d_pos, int64_t count);                               /* :1120 */
/* morton.positions_to_keys morton.py:113 */
int pm_morton_keys(pm_ctx *ctx, const float *d_pos, int64_t np, int64_t *d_keys);
/* utils.reorder_particles :1019: stable radix sort by Morton key, then gather of up to three
 * [np,3] arrays (d_vel/d_acc and their outputs may be NULL).  d_perm (int64[np], may be NULL)
 * receives the permutation (out[i] = in[perm[i]]). */
int pm_reorder_particles(pm_ctx *ctx, int64_t np, const float *d_pos, const float *d_vel, const float *d_acc,
                         float *d_pos_out, float *d_vel_out, float *d_acc_out, int64_t *d_perm);

/* ---- fourier.py --------------------------------------------------------------------------- */
/* fft_3D_real :104 (unnormalised r2c) and ifft_3D_real :251 (c2r carrying 1/n^3).
 * d_xk is float2[n*n*(n/2+1)].  pm_fft_c2r may overwrite d_xk. */
int pm_fft_r2c(pm_ctx *ctx, const float *d_x, void *d_xk);
int pm_fft_c2r(pm_ctx *ctx, void *d_xk, float *d_x);
/* inverse_laplacian :460 / _compensated(p) :502 / _7pt :555, in place on d_xk.
 * h_k2 and h_s are the per-axis fp32 tables (length n) the host builds with the reference's own
 * expressions: k2[i] = k_i**2 and s[i] = sinc(k_i/n) (compensated) or sin(pi k_i/n)**2 (7pt). */
int pm_greens(pm_ctx *ctx, int kind, int p, void *d_xk, const float *h_k2, const float *h_s, float c0);
/* fourier_grid_to_Pk :22-100: bins |k| into int(|k|+0.5), deconvolves sinc^p, also zeroes
 * d_xk[0,0,0].  Outputs are host arrays of length n: per-bin sums of |k|, |delta_k|^2, counts
 * (the caller slices [1:kmax] and divides, as :94-99). */
/* fourier.gradient_inverse_laplacian (fourier.py:606, p = 0) / gradient_inverse_laplacian_compensated (:660,
 * p = MAS order): the force of the `full_fft` solver in k-space, out_m = -i k_m / (2 pi k^2) W^(-2p) x with the
 * k = 0 mode zeroed.  d_out3 receives three half-spectra [3][n, n, n/2+1] complex64 (planar: each goes through
 * pm_fft_c2r on its own; the reference's [n, n, n/2+1, 3] layout is a host-side view).  h_kfreq: signed
 * frequencies i or i - n as fp32; h_sinc: sinc(k h) per axis as for pm_pk. */
int pm_grad_inverse_laplacian(pm_ctx *ctx, int p, const void *d_xk, void *d_out3, const float *h_kfreq,
                              const float *h_sinc);
/* ---- initial conditions (initial_conditions.py:108-158, fourier.py:730-905) -------- */
/* fourier.gradient (fourier.py:730): out_m = i 2 pi k_m x, three planar half-spectra [3][n, n, n/2+1] */
int pm_kspace_gradient(pm_ctx *ctx, const void *d_xk, void *d_out3, const float *h_kfreq);
/* fourier.hessian (:784; a2 = b2 = -1): -k_a1 k_b1 4 pi^2 x;  fourier.sum_of_hessian (:842): -(k_a1 k_b1 + k_a2 k_b2) 4 pi^2 x;
 * fourier.diff_of_hessian (:907; second pair given as 4 + axis): -(k_a1 k_b1 - k_a2 k_b2) 4 pi^2 x */
int pm_kspace_hessian(pm_ctx *ctx, const void *d_xk, void *d_out, const float *h_kfreq, int a1, int b1, int a2, int b2);
/* dealiased_ICS (Orszag's 3/2 rule, initial_conditions.py:1008, 1034-1038 and the same lines of every n-LPT source
 * term): the products of Hessians are formed on a grid of m = 3 n / 2 cells per side.  pm_dealias_resize is
 * initial_conditions.pad (:1859-1894; up = 1: half-spectrum [n, n, n/2+1] -> [m, m, m/2+1]) and trim (:1897-1928;
 * up = 0: the way back): zeros, then the four low / high corner blocks in x and y times kz < n/2 (the Nyquist planes
 * of the n grid are dropped both ways); d_in != d_out.  The _padded entry points are pm_kspace_hessian (h_kfreq: m
 * signed frequencies), pm_fft_r2c and pm_fft_c2r (1/m^3) on the m grid. */
int pm_dealias_resize(pm_ctx *ctx, const void *d_in, void *d_out, int up);
int pm_kspace_hessian_padded(pm_ctx *ctx, const void *d_xk, void *d_out, const float *h_kfreq, int a1, int b1, int a2, int b2);
int pm_fft_r2c_padded(pm_ctx *ctx, const float *d_x, void *d_xk);
int pm_fft_c2r_padded(pm_ctx *ctx, void *d_xk, float *d_x);
/* y = f a b (accumulate 0: utils.prod_vector_vector_inplace, utils.py:500) or y += f a b (1: add_vector_vector_inplace, :330) */
int pm_addcmul(pm_ctx *ctx, float *d_y, float f, const float *d_a, const float *d_b, int64_t count, int accumulate);
/* y += f a b c (utils.add_vector_vector_vector_inplace, utils.py:368) */
int pm_addcmul3(pm_ctx *ctx, float *d_y, float f, const float *d_a, const float *d_b, const float *d_c, int64_t count);
/* initialise_1LPT_center / _edge (mode 0 / 1: x = q - D psi, v = -(D fH) psi, initial_conditions.py:1688-1806) and
 * add_nLPT (mode 2: x += D psi, v += (D fH) psi, :1809-1856) on a displacement field psi [n, n, n, 3]; ctx of size n */
int pm_lpt_displace(pm_ctx *ctx, const float *d_psi, float dplus, float dfh, int mode, float *d_pos, float *d_vel);
/* three planar arrays of m floats -> [m, 3] (the force grid's layout; the tail of fourier.ifft_3D_real_grad, :343) */
int pm_interleave3(pm_ctx *ctx, const float *d_planar, int64_t m, float *d_out);
int pm_pk(pm_ctx *ctx, int p, void *d_xk, const float *h_sinc, double *h_ksum, double *h_pksum,
          double *h_nmodes);

/* ---- fused step pieces (integration.py:192-264, solver.py:453-522) ---------------------------- */
/* solver.fft :453 without the P(k) branch: r2c -> Green's function -> c2r.  d_rhs is preserved
 * unless d_phi == d_rhs. */
int pm_poisson_fft(pm_ctx *ctx, int kind, int p, const float *d_rhs, float *d_phi, const float *h_k2,
                   const float *h_s, float c0);
/* Same result as pm_poisson_fft without rocFFT: five LDS-resident 1-D FFT passes with the Green's
 * function applied between the forward and inverse x transforms (32 <= n <= 1024). */
int pm_poisson_fft_fused(pm_ctx *ctx, int kind, int p, const float *d_rhs, float *d_phi, const float *h_k2,
                         const float *h_s, float c0);
/* pm_poisson_fft_fused on a density deposited by pm_kick_drift_wrap_deposit_khalo (d_rhs = that d_rho) */
int pm_poisson_fft_fused_khalo(pm_ctx *ctx, int kind, int p, const float *d_rhs, const float *d_khalo, float *d_phi,
                               const float *h_k2, const float *h_s, float c0);
/* The three stages of pm_poisson_fft_fused, exposed for the slab-decomposed solve (nx = local x-planes):
 *   pm_fftp_forward : d_rhs [nx,n,n] real            -> d_C [n (ky), n/2+1 (kz), nx]   (z r2c + y forward)
 *   pm_fftp_xgreen  : in place on d_C laid out [n/nx segments][nky][n/2+1][nx] -- what an all-to-all over
 *                     x-slabs delivers -- x forward, G(kx, ky0+ky, kz)/n^3, x inverse
 *   pm_fftp_inverse : d_C [n, n/2+1, nx]            -> d_phi [nx,n,n] real              (y inverse + z c2r) */
int pm_fftp_forward(pm_ctx *ctx, const float *d_rhs, int nx, void *d_C);
/* The same stage by stage, so that a slab driver can pipeline the passes against its all-to-alls: the y passes
 * and the x pass take a kz range [kz0, kz0 + nkz) and work on a spectrum chunk [n (ky), nkz, nx] (after the
 * all-to-all: [nseg][nky][nkz][nx]); the z passes go through the context's internal B buffer (all kz). */
int pm_fftp_z_forward(pm_ctx *ctx, const float *d_rhs, int nx);
int pm_fftp_z_forward_khalo(pm_ctx *ctx, const float *d_rhs, int nx, const float *d_khalo); /* see pm_kick_drift_wrap_deposit_slab_khalo */
int pm_fftp_y_forward(pm_ctx *ctx, int nx, int kz0, int nkz, void *d_C);
int pm_fftp_xgreen_kz(pm_ctx *ctx, int kind, int p, void *d_C, int nx, int nky, int ky0, int kz0, int nkz,
                      const float *h_k2, const float *h_s, float c0);
int pm_fftp_y_inverse(pm_ctx *ctx, const void *d_C, int nx, int kz0, int nkz);
int pm_fftp_z_inverse(pm_ctx *ctx, int nx, float *d_phi);
int pm_fftp_xgreen(pm_ctx *ctx, int kind, int p, void *d_C, int nx, int nky, int ky0, const float *h_k2,
                   const float *h_s, float c0);
int pm_fftp_inverse(pm_ctx *ctx, const void *d_C, int nx, float *d_phi);
/* leapfrog first half, integration.py:250-258: v += -half_dt*a ; x += dt*v ; periodic_wrap(x) */
int pm_kick_drift_wrap(pm_ctx *ctx, float *d_pos, float *d_vel, const float *d_acc, float half_dt, float dt,
                       int64_t np);

/* solver.py:205-209 + integration.py:262 + the next step's integration.py:294, 325 in one pass:
 *   acc = inv<scheme>_vec(force, pos);  if (d_vel) vel += -half_dt*acc;
 * and max|acc|, max|vel| (after the kick) are left in the context for pm_last_maxabs().  LDS-tiled
 * for Morton-sorted streams; any particle order gives the same result.  Needs n >= 16. */
int pm_interp_kick(pm_ctx *ctx, int scheme, const float *d_force, const float *d_pos, int64_t np, float *d_acc,
                   float *d_vel, float half_dt);
/* Optional preparation for the tile-range kernels (pm_deposit with sorted_hint 1, pm_interp_kick,
 * pm_gradient_interp_kick), no reference counterpart.  A stream that was Morton-sorted some steps ago
 * (main.py:146-150 re-orders every n_reorder steps) has stragglers: particles filed under a tile they
 * have left.  Served one by one from global memory they dominate the kernels already at 1 % (scattered
 * atomics / loads).  pm_tile_lists computes the tile ranges once and, in one pass over the positions,
 * the list of stragglers grouped by the tile they are in now, so that every later call on the SAME d_pos /
 * np serves all particles from LDS.  The lists are dropped by pm_tile_lists(ctx, NULL, 0), by any ctx call
 * that writes positions, and must be dropped by the caller before d_pos is modified otherwise.  Results
 * are identical with and without (gather: same bits; deposit: same sums up to atomic ordering). */
int pm_tile_lists(pm_ctx *ctx, const float *d_pos, int64_t np);
/* Outcome of the most recent straggler scan on this context (pm_tile_lists, pm_kick_drift_wrap_deposit[_slab]):
 * number of particles outside the tile they are filed under, the np it was counted over, and whether the lists
 * hold them all (they have room for every particle; a sub-list can overflow).  *stragglers = -1 before any scan.  Lets the caller re-sort
 * (pm_reorder_particles) when the sort has aged instead of on a fixed cadence only. */
int pm_tile_lists_info(pm_ctx *ctx, int64_t *stragglers, int64_t *np, int *complete);
/* pm_kick_drift_wrap (integration.py:250-258) + pm_deposit (mesh.py:2240/2284/2468) in one kernel, for a
 * stream that is (approximately) Morton-sorted: the update streams through HBM while the LDS does the
 * deposit, and the positions are read once.  d_pos / d_vel are updated in place exactly as by
 * pm_kick_drift_wrap; d_rho receives the density of the NEW positions (same sums as pm_deposit up to
 * atomic ordering).  The tile ranges (of the positions before the drift) and the list of particles that
 * ended up outside the tile they are filed under stay prepared: a following pm_tile_lists(ctx, d_pos, np)
 * keeps them, so the gather of the same step reuses them.  Single-GPU layout, n >= 16, np < 2^32. */
int pm_kick_drift_wrap_deposit(pm_ctx *ctx, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                               float half_dt, float dt, int64_t np, float *d_rho);
/* The same with the halo cells of the tiles kept out of d_rho.  The flush of a tile (8 x 16 x 16 cells + one halo
 * cell per side) adds to the density with global float atomics, which the memory side executes as one 64-byte
 * read-modify-write per touched sector, whatever it carries: at 1024^3 the flush is 5 of the call's 17 ms, half
 * of it the single k-halo cells at the two ends of every tile row.  Here the first workgroup of every tile uses
 * no atomics: it STORES the tile's own cells into d_rho (which need not be cleared) and its halo cells into
 * d_khalo (pm_khalo_floats(n) floats, every one written), laid out as
 *   K  [tile column (ti, tj)][li 0..9][low, high][tk][lj 0..17] the k-halo cells,
 *   Hi [tile column (ti, tj)][low, high][lj 0..17][k 0..n)     the i-halo planes (their cells inside in k),
 *   Hj [tile column (ti, tj)][low, high][li 1..8][k 0..n)      the j-halo rows (inside in i and k);
 * only surplus workgroups of over-full tiles and listed stragglers still add to d_rho atomically.  d_rho then
 * lacks exactly the contributions held in d_khalo.  Consumers: pm_poisson_fft_fused_khalo (its z pass adds the
 * buffer to the density lines it loads) or pm_khalo_apply (adds it into d_rho). */
int pm_kick_drift_wrap_deposit_khalo(pm_ctx *ctx, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                     float half_dt, float dt, int64_t np, float *d_rho, float *d_khalo);
int pm_khalo_apply(pm_ctx *ctx, float *d_rho, const float *d_khalo);
int64_t pm_khalo_floats(int n);
/* The same on a slab (pm_set_slab, pm_set_tile_window in force): d_rho is the rank's [nx_slab, n, n] density
 * array, which must be wide enough (ghost planes) for the footprint of every particle AFTER the drift -- the
 * caller bounds the drift (it knows dt, max|v|, max|a|) and chooses the ghost width; the particles whose new x
 * left [x_lo, x_hi) are deposited here, by their old owner, and listed as by pm_kick_drift_wrap_slab.  The tile
 * lists are dropped: the migration that follows renumbers the particles. */
int pm_kick_drift_wrap_deposit_slab(pm_ctx *ctx, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                    float half_dt, float dt, int64_t np, float *d_rho, float x_lo, float x_hi,
                                    int64_t *d_leave_idx, int cap, int *h_count);
/* ... with the halo buffer of pm_kick_drift_wrap_deposit_khalo.  d_rho = the tile window's planes + the same number
 * (>= 1) of ghost planes either side; d_khalo = pm_khalo_floats_slab(n, nx_window) floats, the window's rows of tiles
 * only.  The owners store the window's planes (not cleared first); the ghost planes are cleared here and receive, with
 * atomics, what leaves the window through its x faces: the outward halo cells of its first and last row of tiles and
 * the clouds of particles in tiles outside it (leavers).  After the caller has folded the ghost planes into the
 * neighbours' edge planes, pm_fftp_z_forward_khalo adds the buffer to the density lines it loads (no atomics: every
 * line of planes 1 .. nx - 2 takes its neighbours' entries as on one GPU, the first and last plane only those of
 * their own column of tiles). */
int pm_kick_drift_wrap_deposit_slab_khalo(pm_ctx *ctx, int scheme, float *d_pos, float *d_vel, const float *d_acc,
                                          float half_dt, float dt, int64_t np, float *d_rho, float x_lo, float x_hi,
                                          int64_t *d_leave_idx, int cap, int *h_count, float *d_khalo);
int64_t pm_khalo_floats_slab(int n, int nx_window);
/* mesh.derivative (mesh.py:2072, orders 2/3/5/7) + pm_interp_kick in one kernel: every workgroup stages the
 * potential of its tile in LDS, differentiates it there with the reference's stencil and gathers from the
 * result, so the [n,n,n,3] force grid is never written or read.  Same bits as pm_gradient followed by
 * pm_interp_kick.  d_phi holds nx_phi planes starting at global plane x0_phi (single GPU: n, 0; slabs: the
 * owned planes + reach+1 ghosts each side).  d_vel may be NULL (no kick).  Needs n >= 16 and a 16-byte
 * aligned d_phi (any hipMalloc / torch allocation is). */
int pm_gradient_interp_kick(pm_ctx *ctx, int order, int scheme, const float *d_phi, int nx_phi, int x0_phi,
                            const float *d_pos, int64_t np, float *d_acc, float *d_vel, float half_dt);
/* pm_gradient_interp_kick with the second kick, doubling as an incremental re-sort of the particle stream
 * (the job of utils.reorder_particles, utils.py:1019, which main.py:146-150 calls every n_reorder steps; no
 * reference counterpart for doing it inside the step).  The gather reads every particle's position and velocity
 * and writes its velocity and acceleration anyway; here it writes position, kicked velocity and acceleration OUT
 * OF PLACE, tile after tile (8 x 16 x 16 cells, the unit of the tile-range kernels), every particle into the
 * range of the tile it is expected to be in after the next drift (x + dt_next * (v - (dt + dt_next) / 2 * a_old),
 * clamped to one cell, the Courant bound of integration.py:324-326; decided and recorded per particle by the
 * preceding pm_kick_drift_wrap_deposit[_khalo], which has a_old; dt_next from pm_set_predicted_dt, 0 = the tile it
 * is in now).  The
 * *_out stream enters the next step with (nearly) no stragglers, for 12 more bytes per particle than the
 * in-place gather.  Values are those pm_gradient_interp_kick computes (same
 * bits); only the particle order changes -- inside a tile it is the order of arrival, which no kernel result
 * depends on.  d_id / d_id_out (both or neither): uint32 tags carried along, so a caller can follow the
 * permutation.  Needs the ranges, lists and stay counts pm_kick_drift_wrap_deposit[_khalo] left prepared for
 * exactly d_pos / np (fails otherwise; check pm_tile_lists_info's `complete` first).  The context remembers the
 * tile offsets of d_pos_out: the next pm_kick_drift_wrap_deposit on it skips the range search.
 * order = 0: d_phi is the force grid [n, n, n, 3] (pm_interp_kick's input: the f(R) path, whose force is not the
 * gradient of one potential).  Single-GPU layout, n >= 16, max|acc| / max|vel| left for pm_last_maxabs as usual. */
int pm_gradient_interp_kick_rebucket(pm_ctx *ctx, int order, int scheme, const float *d_phi, const float *d_pos,
                                     const float *d_vel, int64_t np, float half_dt, float *d_pos_out, float *d_vel_out,
                                     float *d_acc_out, const unsigned *d_id, unsigned *d_id_out);
/* The same on a slab (multi-GPU x-decomposition; no reference counterpart): the particles change arrays between the
 * fused kick (pm_kick_drift_wrap_deposit_slab, which knows where they will go) and the gather (migration of the
 * leavers to their new owners), so the kick only RECORDS the destination codes -- one byte per particle, relative to
 * the tile the particle is in, into the caller's array set by pm_set_destination_codes (NULL: off) -- the caller
 * migrates that array with the particles, and pm_rebucket_prepare counts the destinations of the stream as it is
 * after the migration (needs complete lists from pm_tile_lists for exactly d_pos / np; destinations outside the
 * tile window become "stays": those particles are about to leave the rank).  pm_gradient_interp_kick_rebucket_slab
 * is then pm_gradient_interp_kick_rebucket with pm_gradient_interp_kick's potential-slab geometry. */
int pm_set_destination_codes(pm_ctx *ctx, unsigned char *d_codes);
int pm_rebucket_prepare(pm_ctx *ctx, const float *d_pos, int64_t np, unsigned char *d_codes);
/* pm_tile_lists + pm_rebucket_prepare in one pass over the positions (what the slab step calls after its migration) */
int pm_tile_lists_codes(pm_ctx *ctx, const float *d_pos, int64_t np, unsigned char *d_codes);
/* The caller changed the particle count of the stream the last re-bucketing gather wrote (refilled leavers' slots,
 * appended or dropped the surplus): the tile offsets on file are clamped to the new count, so that the next
 * pm_tile_lists[_codes] / pm_kick_drift_wrap_deposit on d_pos keeps using them (the surplus past the old count is
 * filed under the last tile of the tile window in force at this call).  No-op for any other array. */
int pm_sorted_stream_resized(pm_ctx *ctx, const float *d_pos, int64_t np_new);
int pm_gradient_interp_kick_rebucket_slab(pm_ctx *ctx, int order, int scheme, const float *d_phi, int nx_phi, int x0_phi,
                                          const float *d_pos, const float *d_vel, int64_t np, float half_dt,
                                          float *d_pos_out, float *d_vel_out, float *d_acc_out, const unsigned *d_id,
                                          unsigned *d_id_out);
/* Guess of the next step's dt for the filing above; must be set before the pm_kick_drift_wrap_deposit[_khalo]
 * whose counts the re-bucketing gather uses (the gather takes the value those counts were made with). */
int pm_set_predicted_dt(pm_ctx *ctx, float dt_next);
/* ... and of the growth of the accelerations over the coming step: the filing uses scale * a_old for the
 * acceleration it does not know yet (1, the default: a_old itself).  pm_last_acc_meanabs gives the caller what to
 * extrapolate: the mean |a_x| over a sample of particles (every 64th tile of a large launch, every tile of a small one; 0 below 1024 particles) of the last fused gather, fetched by the same read-back as pm_last_maxabs (call that first; host value, no synchronisation). */
int pm_set_predicted_acc_scale(pm_ctx *ctx, float scale);
int pm_last_acc_meanabs(pm_ctx *ctx, double *h_meanabs);
/* utils.max_abs of the acc / vel arrays written by the last pm_interp_kick (syncs).  *h_max_vel is 0
 * if that call had d_vel == NULL. */
int pm_last_maxabs(pm_ctx *ctx, float *h_max_acc, float *h_max_vel);
/* kernel choice for pm_interp_kick, same meaning as pm_deposit's sorted_hint 1 / 2 (default 1) */
int pm_set_gather_mode(pm_ctx *ctx, int mode);

/* ---- multigrid pieces (laplacian.py, cubic.py, mond.py, mesh.py transfer operators) --------------
 * The V-cycle recursion stays on the host (multigrid.py:474-579); every function takes the grid
 * size `n` of the level it works on.  Norm functions return a host value and synchronise. */
/* mesh.restriction :14 (scale = 0.125f) / mesh.minus_restriction :63 (scale = -0.125f): [nf]^3 -> [nf/2]^3 */
int pm_restriction(pm_ctx *ctx, const float *d_x, int nf, float scale, float *d_out);
/* mesh.prolongation :206 (add = 0) / mesh.add_prolongation :334 (add = 1): [nc]^3 -> (+=) [2nc]^3 */
int pm_prolongation(pm_ctx *ctx, float *d_y, const float *d_xc, int nc, int add);
/* laplacian.operator :12 (mode 0, d_b unused) / laplacian.residual :63 (mode 1: b - Ax) */
int pm_lap_operator(pm_ctx *ctx, const float *d_x, const float *d_b, int n, int mode, float *d_out);
/* laplacian.restrict_residual :125: fused residual + restriction, [n]^3 -> [n/2]^3 */
int pm_lap_restrict_residual(pm_ctx *ctx, const float *d_x, const float *d_b, int n, float *d_out);
/* laplacian.residual_error :327: sqrt(sum (b - Ax)^2) */
int pm_lap_residual_error(pm_ctx *ctx, const float *d_x, const float *d_b, int n, float *h_out);
/* sqrt(sum (f*a - b)^2): tail of laplacian.truncation_error :529 (f = 1) and cubic.truncation_error :1057 (f = 4) */
int pm_diff_norm(pm_ctx *ctx, const float *d_a, const float *d_b, float f, int64_t count, float *h_out);
/* multigrid.V_cycle_FAS (multigrid.py:521-579) for the f(R) scalaron on x in place, the whole recursion issued from C
 * (fr_n = 1: cubic.py operators, 2: quartic.py): smoothing, restricted residual, restricted solution / density term,
 * coarse right-hand side 4 res + L(x_c), coarse cycle, x += prolongation(x_corr - x_c), smoothing.  Same kernels,
 * same bits as the per-operator entry points called in that order.  n >= 8. */
int pm_fr_vcycle(pm_ctx *ctx, int fr_n, float *d_x, const float *d_b, float q, int n, int npre, int npost);
/* cubic.smoothing / smoothing_with_rhs (cubic.py:1065, 1100; quartic.py:1066, 1101): n_smoothing red-black sweeps of
 * the scalaron with omega = 1.25, result in x; d_rhs may be NULL.  For n >= 256 each sweep is one pass over the grid,
 * as pm_lap_smoothing (the same marching kernel with the cubic / quartic cell update), out of place into a scratch grid
 * of the context; same bits as pm_cubic / pm_quartic_gauss_seidel called n_smoothing times.  pm_fr_vcycle's sweeps are
 * these.  pm_set_fused_sweep(ctx, 0) makes every sweep the two half-sweep launches again. */
int pm_fr_smoothing(pm_ctx *ctx, int fr_n, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                    int n_smoothing);
/* laplacian.smoothing (laplacian.py:1026): n_smoothing red-black sweeps with omega = 1.25 on x in place.  For
 * n >= 256 each sweep is ONE pass over the grid (red and black updated by a workgroup marching through LDS planes,
 * out of place into a scratch grid of the context, copied back after an odd count): x and b read once, x written once
 * per sweep instead of twice.  Same bits as pm_lap_gauss_seidel called n_smoothing times.  pm_set_fused_sweep(ctx, 0)
 * makes every sweep the two half-sweep launches again (comparison / fallback). */
int pm_lap_smoothing(pm_ctx *ctx, float *d_x, const float *d_b, int n, int n_smoothing);
int pm_set_fused_sweep(pm_ctx *ctx, int on);
/* multigrid.V_cycle (multigrid.py:474-517) on x in place, the whole recursion down to the 4^3 grid issued from C:
 * Npre smoothing sweeps, restricted residual, first guess -h^2/6 res (laplacian.py:765), coarse cycle (below 8^3 cells
 * only Npre sweeps), added prolongation, Npost sweeps; omega = 1.25.  Same kernels, same bits as the per-operator
 * entry points called in that order.  n >= 8. */
int pm_lap_vcycle(pm_ctx *ctx, float *d_x, const float *d_b, int n, int npre, int npost);
/* A cycle is 40-90 small launches (the levels below 64^3 are launch-bound): pm_lap_vcycle captures the sequence as a
 * hipGraph the first time it sees a set of arguments and replays it when the same arrays come back (8 graphs kept per
 * context; switches itself off if the caller's arrays keep changing or any graph call fails).  mode 0: plain launches.
 * pm_vcycle_graph_stats: cycles replayed / captured so far.  No reference counterpart. */
/* on (default): pm_lap_vcycle runs every level from 16^3 down (22 launches) in one workgroup with the grids in LDS,
 * same expressions in the same order (same bits); off: level by level. */
int pm_set_fused_coarse(pm_ctx *ctx, int on);
/* on (default): the f(R) smoothers (pm_cubic / pm_quartic_gauss_seidel, pm_fr_vcycle) run their row kernels for
 * n >= 16 -- eight consecutive k-cells per thread, 16-byte accesses, same expression per cell; off: a cell per thread. */
int pm_set_fr_rows_sweep(pm_ctx *ctx, int on);
/* on (default): pm_mond_rhs evaluates the simple interpolating function (mond.py:16-34, kind 0) as
 * 1/2 + sqrt(1/4 + g0 / |grad|) with the hardware's 1-ulp reciprocal square root and square root (1.5e-7 relative, the
 * size of the exact chain's own roundings); off: IEEE sqrt, divide, sqrt as the reference writes it (comparison). */
int pm_set_mond_fast_nu(pm_ctx *ctx, int on);
/* on (default): pm_fr_vcycle takes the restricted operator / residual, solution and density term of a level of 512^3 and
 * more in ONE marching pass (the operator's values never reach memory; same expression, same order of the eight-cell
 * sums: the same bits); off: operator kernel + three pm_restriction launches (comparison). */
int pm_set_fr_fused_restrict(pm_ctx *ctx, int on);
int pm_set_vcycle_graphs(pm_ctx *ctx, int mode);
int pm_vcycle_graph_stats(pm_ctx *ctx, unsigned *replayed, unsigned *captured);
/* laplacian.gauss_seidel :844: one red + one black sweep with over-relaxation f_relax, in place */
int pm_lap_gauss_seidel(pm_ctx *ctx, float *d_x, const float *d_b, int n, float f_relax);
/* cubic.operator :23 (d_rhs NULL) / cubic.residual_with_rhs :90 */
int pm_cubic_operator(pm_ctx *ctx, const float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                      float *d_out);
/* cubic.residual_error :844 */
int pm_cubic_residual_error(pm_ctx *ctx, const float *d_x, const float *d_b, float q, int n, float *h_out);
/* cubic.initialise_potential :217 */
int pm_cubic_initialise_potential(pm_ctx *ctx, const float *d_b, float q, int n, float *d_out);
/* cubic.gauss_seidel :269 (d_rhs NULL) / cubic.gauss_seidel_with_rhs :446 */
int pm_cubic_gauss_seidel(pm_ctx *ctx, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                          float f_relax);
/* quartic.py (f(R) n = 2) twins of the four cubic entry points: operator :20 / residual_with_rhs :86,
 * residual_error :844, initialise_potential :214, gauss_seidel :270 / gauss_seidel_with_rhs :448 */
int pm_quartic_operator(pm_ctx *ctx, const float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                        float *d_out);
int pm_quartic_residual_error(pm_ctx *ctx, const float *d_x, const float *d_b, float q, int n, float *h_out);
int pm_quartic_initialise_potential(pm_ctx *ctx, const float *d_b, float q, int n, float *d_out);
int pm_quartic_gauss_seidel(pm_ctx *ctx, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                            float f_relax);
/* mond.rhs_simple :171 / rhs_n :322 / rhs_beta :476 / rhs_gamma :630 / rhs_delta :784
 * (kind 0..4; alpha = n, beta, gamma or delta): d_out = div[nu(|grad pot|/g0) grad pot] */
int pm_mond_rhs(pm_ctx *ctx, const float *d_pot, float *d_out, int n, float g0, int kind, float alpha);

/* ---- slab decomposition (multi-GPU; no reference counterpart: PySCo is single-process) ------------
 * A rank owns nx_own = n/G consecutive x-planes and the particles inside them.  Its local grids are
 * [nx, n, n] with ghost planes; pm_set_slab tells the particle<->mesh kernels (pm_deposit, pm_interp,
 * pm_interp_kick) that local plane 0 is global plane x0 (periodic) and that there are nx planes.
 * Default (single GPU): nx = n, x0 = 0. */
int pm_set_slab(pm_ctx *ctx, int nx, int x0);
/* Planes [x_lo, x_lo + nx) that ALL particles handed to the tile-range kernels live in (a rank's owned planes
 * after migration): only the tiles of that window get a workgroup instead of the whole mesh's.  x_lo and nx
 * multiples of 8, nx/8 a power of two; anything else (and the default 0, n) enumerates every tile. */
int pm_set_tile_window(pm_ctx *ctx, int x_lo, int nx);
/* gradient on a slab: d_phi holds nx_phi planes from global plane x0_phi, d_force receives nx_force
 * planes from x0_force; needs phi on [x0_force - R, x0_force + nx_force + R), R = 1, 1, 2, 3 for order 2, 3, 5, 7 */
int pm_gradient_slab(pm_ctx *ctx, int order, const float *d_phi, int nx_phi, int x0_phi, float *d_force,
                     int nx_force, int x0_force);
/* pm_kick_drift_wrap that also lists the particles whose new x left the slab [x_lo, x_hi): their
 * indices (unordered) go to d_leave_idx[0 .. min(*h_count, cap)), *h_count receives how many there
 * were (synchronises; if it exceeds cap the caller rescans). */
int pm_kick_drift_wrap_slab(pm_ctx *ctx, float *d_pos, float *d_vel, const float *d_acc, float half_dt, float dt,
                            int64_t np, float x_lo, float x_hi, int64_t *d_leave_idx, int cap, int *h_count);
/* Multigrid level operations on a slab (linear V-cycle and the QUMOND source with halo exchange): every
 * level array is [nx + 2, n, n] -- the nx owned planes of that level plus one ghost plane on each side,
 * which the caller fills from its ring neighbours before each call that reads x-neighbours.  Same
 * kernels and the same arithmetic per cell as the single-GPU entry points (laplacian.py:12-226, 327-381,
 * 844-1022; mesh.py:14-455; mond.py:171-932).
 *   pm_lap_gauss_seidel_slab: ONE colour half-sweep (1 = red, i+j+k odd; 0 = black), x_lo = global index of
 *                             the first owned plane; the ghosts must be refreshed between the two halves
 *   pm_lap_restrict_residual_slab / pm_restriction_slab: fine [nx+2] -> coarse [nx/2+2] (nx even)
 *   pm_prolongation_slab: coarse [nxc+2] (ghosts filled) -> fine [2 nxc + 2]
 *   pm_lap_residual_sumsq_slab / pm_diff_sumsq: local fp64 sum of squares (ranks add them, then sqrt) */
int pm_lap_gauss_seidel_slab(pm_ctx *ctx, float *d_x, const float *d_b, int n, int nx, int x_lo, int colour,
                             float f_relax);
int pm_lap_restrict_residual_slab(pm_ctx *ctx, const float *d_x, const float *d_b, int n, int nx, float *d_out);
int pm_restriction_slab(pm_ctx *ctx, const float *d_x, int nf, int nxf, float scale, float *d_out);
int pm_prolongation_slab(pm_ctx *ctx, float *d_y, const float *d_xc, int nc, int nxc, int add);
int pm_lap_operator_slab(pm_ctx *ctx, const float *d_x, const float *d_b, int n, int nx, int mode, float *d_out);
int pm_lap_residual_sumsq_slab(pm_ctx *ctx, const float *d_x, const float *d_b, int n, int nx, double *h_sum);
int pm_diff_sumsq(pm_ctx *ctx, const float *d_a, const float *d_b, float f, int64_t count, double *h_sum);
int pm_mond_rhs_slab(pm_ctx *ctx, const float *d_pot, float *d_out, int n, int nx, float g0, int kind, float alpha);
/* f(R) scalaron operator on a slab (cubic.py / quartic.py; fr_n = 1 / 2), same conventions: pm_fr_operator_slab
 * is operator (d_rhs NULL) or residual_with_rhs, pm_fr_gauss_seidel_slab ONE colour half-sweep (with or without
 * rhs), pm_fr_residual_sumsq_slab the local sum of squares of the operator; pm_gradient_fr_slab the f(R) force
 * (mesh.derivative_fR) from potential and scalaron slabs with ghosts. */
int pm_fr_operator_slab(pm_ctx *ctx, int fr_n, const float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                        int nx, float *d_out);
int pm_fr_residual_sumsq_slab(pm_ctx *ctx, int fr_n, const float *d_x, const float *d_b, float q, int n, int nx,
                              double *h_sum);
int pm_fr_gauss_seidel_slab(pm_ctx *ctx, int fr_n, float *d_x, const float *d_b, float q, const float *d_rhs, int n,
                            int nx, int x_lo, int colour, float f_relax);
int pm_gradient_fr_slab(pm_ctx *ctx, int order, int fr_n, float f, const float *d_phi, const float *d_u, int nx_phi,
                        int x0_phi, float *d_force, int nx_force, int x0_force);
/* cubic / quartic initialise_potential on `count` cells of a level with n cells per side (slab level arrays) */
int pm_cubic_initialise_potential_flat(pm_ctx *ctx, const float *d_b, float q, int n, int64_t count, float *d_out);
int pm_quartic_initialise_potential_flat(pm_ctx *ctx, const float *d_b, float q, int n, int64_t count, float *d_out);

/* slab FFT pieces: batched 2-D r2c / c2r over (y,z) on nplanes x-planes ([np,n,n] <-> [np,n,n/2+1]; the
 * c2r carries 1/n^3), and the in-place strided 1-D c2c along x of an [n, ncolumns] complex array
 * (the layout after the all-to-all transpose, ncolumns = (n/G)*(n/2+1)). */
int pm_fft2d_r2c(pm_ctx *ctx, const float *d_x, void *d_xk, int nplanes);
int pm_fft2d_c2r(pm_ctx *ctx, void *d_xk, float *d_x, int nplanes);
int pm_fft1d_x(pm_ctx *ctx, void *d_data, int64_t ncolumns, int inverse);
/* pm_greens on the transposed layout x[i, j0:j0+nj, k] */
int pm_greens_slab(pm_ctx *ctx, int kind, int p, void *d_xk, const float *h_k2, const float *h_s, float c0, int j0,
                   int nj);

/* multigrid.linear :57-83 / multigrid.FAS :129-138, the stopping loop included: V-cycles until the residual norm is
 * under `tolerance` or no longer halves; *h_cycles V-cycles were taken, *h_residual is the last norm.
 * max_cycles <= 0: unbounded, as the reference. */
int pm_lap_solve(pm_ctx *ctx, float *d_x, const float *d_b, int n, int npre, int npost, float tolerance, int max_cycles,
                 int *h_cycles, float *h_residual);
int pm_fr_solve(pm_ctx *ctx, int fr_n, float *d_x, const float *d_b, float q, int n, int npre, int npost, float tolerance,
                int max_cycles, int *h_cycles, float *h_residual);

/* "The kick re-buckets" (round 3; the single-GPU loop's incremental re-sort, utils.reorder_particles :1019 in effect):
 * pm_gradient_interp_kick_codes = pm_gradient_interp_kick with the kick, in place (d_vel kicked, d_acc written, nothing
 * permuted), which also DECIDES where every particle is filed after the next drift (pm_set_predicted_dt's guess) -- a flag
 * byte for the particles that stay in their tile, an incoming list per tile for the others; needs complete tile lists
 * for d_pos / np.  pm_kick_drift_wrap_deposit_rebucket = pm_kick_drift_wrap_deposit[_khalo] (d_khalo NULL: plain flush)
 * obeying those decisions: position and velocity of the new step are written OUT OF PLACE into d_pos_out / d_vel_out, tile
 * after tile (d_pos / d_vel / d_acc are only read; optional uint32 tags follow), and the new stream's tile offsets, ranges
 * and straggler lists are left prepared.  pm_rebucket_kick_ready: *h_ready = 1 if the decisions on file are for exactly
 * d_pos / np and no list overflowed (otherwise take the in-place step and re-bucket with
 * pm_gradient_interp_kick_rebucket, which has no lists).  A tile's incoming list holds 256 entries + a quarter of the
 * tile's own population + an eighth (x faces) / a sixteenth (y, z faces) of its face neighbours': what evenly filled
 * neighbours can send in one drift of at most a cell; 0.94 entries per particle in all. */
int pm_gradient_interp_kick_codes(pm_ctx *ctx, int order, int scheme, const float *d_phi, const float *d_pos, float *d_vel,
                                  int64_t np, float half_dt, float *d_acc);
int pm_rebucket_kick_ready(pm_ctx *ctx, const float *d_pos, int64_t np, int *h_ready);
int pm_kick_drift_wrap_deposit_rebucket(pm_ctx *ctx, int scheme, const float *d_pos, const float *d_vel, const float *d_acc,
                                        float half_dt, float dt, int64_t np, float *d_rho, float *d_khalo, float *d_pos_out,
                                        float *d_vel_out, const unsigned *d_id, unsigned *d_id_out);

/* ---- the fp64 device mode (pm_create(N, Np, ngpu, real_bytes)) ---------------------------
 * pm_ctx_create_real(..., real_bytes): 4 = pm_ctx_create (the reference's float32, every entry point above);
 * 8 = a context of the fp64 mode of the Newtonian FFT step, served by the pm_real64_* entry points below, which take
 * arrays of double in the layouts above.  The reference has no fp64 mode (mesh.py:2467, fourier.py:21): these are the
 * same formulas in double, there to state the <= 1e-10 tolerance between the build's fp64 CPU and fp64 device paths
 * (BASELINE.md section 6).  Each names the reference function whose formula it evaluates. */
int pm_ctx_create_real(pm_ctx **out, int n, void *stream, int real_bytes);
int pm_ctx_real_bytes(pm_ctx *ctx);
/* integration.py:250-258: v += -half_dt a; x += dt v; utils.periodic_wrap(x) */
int pm_real64_kick_drift_wrap(pm_ctx *ctx, double *d_pos, double *d_vel, const double *d_acc, double half_dt, double dt,
                              int64_t np);
/* utils.linear_operator_inplace :644-684 (x = a x + b) and prod_vector_scalar_inplace (b = 0) */
int pm_real64_scale_add(pm_ctx *ctx, double *d_x, double a, double b, int64_t n);
/* utils.max_abs :410-430 */
int pm_real64_max_abs(pm_ctx *ctx, const double *d_x, int64_t n, double *h_out);
/* mesh.NGP :2240 / CIC :2284 / TSC :2468; d_rho [n,n,n] is overwritten */
int pm_real64_deposit(pm_ctx *ctx, int scheme, const double *d_pos, int64_t np, double *d_rho);
/* the two above in one pass over the particles (integration.py:250-258 followed by the deposit of solver.py:75-90): same
 * expressions per particle, position and velocity updated in place */
int pm_real64_kick_drift_wrap_deposit(pm_ctx *ctx, int scheme, double *d_pos, double *d_vel, const double *d_acc, double half_dt,
                                      double dt, int64_t np, double *d_rho);
/* solver.fft :453-522: r2c, Green's function (pm_greens' kinds and tables, in double), c2r with 1/n^3.
 * d_rhs may be overwritten. */
int pm_real64_poisson_fft(pm_ctx *ctx, int kind, int p, const double *d_rhs, double *d_phi, const double *h_k2,
                          const double *h_s, double c0);
/* mesh.derivative :2072 (order 2, 3, 5, 7): d_force [n,n,n,3] */
int pm_real64_gradient(pm_ctx *ctx, int order, const double *d_phi, double *d_force);
/* mesh.invNGP_vec :2639 / invCIC_vec :2752 / invTSC_vec :2960 into d_acc; with d_vel (may be NULL) also the second
 * kick v += -half_dt a (integration.py:262); h_max_acc / h_max_vel (may be NULL): max|a|, max|v| (integration.py:267-326) */
int pm_real64_interp_kick(pm_ctx *ctx, int scheme, const double *d_force, const double *d_pos, int64_t np, double *d_acc,
                          double *d_vel, double half_dt, double *h_max_acc, double *h_max_vel);
/* fourier.fourier_grid_to_Pk :22-100 of fft_3D_real(d_rho): per-bin sums of |k|, |delta_k|^2 W^(-2p), mode counts (n each) */
int pm_real64_pk(pm_ctx *ctx, int p, double *d_rho, const double *h_sinc, double *h_ksum, double *h_pksum, double *h_nmodes);
/* utils.reorder_particles (utils.py:1019-1116) for float64 arrays: stable sort by Morton key (morton.py:42-137), the
 * optional velocity / acceleration carried along (in/out pairs), d_perm (optional, int64[np]): new[i] = old[perm[i]].
 * key_bits in 1..21: only the top key_bits bits per axis decide the order; -1: only the 8 x 16 x 16-cell tile of the
 * context's mesh does (all the kernels care about; two radix passes at 512^3); 0: all 21 bits, the reference's key. */
int pm_real64_reorder_particles(pm_ctx *ctx, int64_t np, const double *d_pos, const double *d_vel, const double *d_acc,
                                double *d_pos_out, double *d_vel_out, double *d_acc_out, int64_t *d_perm, int key_bits);
/* What the last pm_real64_interp_kick (with a read-back of the maxima) saw of the stream's order: *h_aged particles of
 * *h_np were served one by one through L2 because fewer than 32 particles of their 2048-particle chunk shared their
 * tile, and the *h_chunks workgroups made *h_passes tile passes (a sorted stream: the few particles of near-empty tiles,
 * and at most tiles + chunks passes; a shuffled one: nearly every particle).  A performance measure only -- every kernel
 * serves every particle whatever the order -- on which the host loop decides to re-sort
 * (pysco_amd.main.reorder_policy).  *h_np = -1: no gather since the context was created / re-sorted. */
int pm_real64_stream_age(pm_ctx *ctx, int64_t *h_aged, int64_t *h_np, int64_t *h_passes, int64_t *h_chunks);
/* sizes the context's sort workspace for np particles now (a device allocation of a few GB: ~0.15 s that the first
 * pm_real64_reorder_particles would otherwise spend inside the step loop) */
int pm_real64_reserve_sort(pm_ctx *ctx, int64_t np);

#ifdef __cplusplus
}
#endif
#endif /* PMSTEP_H */
