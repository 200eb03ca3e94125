/* mlmcpi_hip.h -- C ABI of the MI355X (gfx950) sweep engine.
 *
 * Drop-in boundary for the inner MCMC sweep of eikehmueller/mlmcpathintegral: every entry point
 * replaces the body of one virtual method of the reference's Action / Sampler / QoI interfaces
 * (cited per function, paths relative to the reference's src/), batched over B independent chains
 * that live in HBM.  Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, a negative mlmcpi_status otherwise;
 *     mlmcpi_last_error() gives the message of the calling thread's last failure;
 *   - pointers named d_* are DEVICE pointers (from mlmcpi_malloc or any HIP allocator, e.g. a
 *     torch tensor's data_ptr()); everything else is host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous with respect to the host unless stated otherwise;
 *   - state layout is the reference's SampleState layout, chain-major:
 *       1-D paths      x[b*M + j]                         (common/samplestate.hh:19-53)
 *       GFF            phi[b*Mt*Mx + Mt*j + i]            (lattice/lattice2d.hh:230-245)
 *       Schwinger      theta[b*2*Mt*Mx + 2*Mt*j + 2*i + mu]  (lattice/lattice2d.hh:348-354)
 *   - randomness is counter based: Philox4x32-10 keyed by `seed`, counter =
 *     (site, chain0 + b, step, purpose<<24 | sub).  Results do not depend on grid shape, tile
 *     size, batch composition or number of GPUs -- to the last bit.
 *   - BIT REPRODUCIBILITY ACROSS LAUNCH PLANS holds for a FIXED (n_overrelax, fuse, MLMCPI_OR_KERNEL) setting only.
 *     The overrelaxation sweeps of one launch of the Schwinger action (mlmcpi_lattice_sweep_draw*) and of
 *     the rotor (mlmcpi_path_sweep_draw*) are evaluated as ONE closed form (K sweeps = one signed sum of
 *     2 K plaquettes / path differences per link / site): the same map as K single sweeps, other rounding.
 *     K sweeps in one launch and the same sweeps in two launches -- another `fuse`, or
 *     MLMCPI_OR_KERNEL=block (which makes BOTH actions sweep by sweep) -- differ in the last bits
 *     (<= 4e-14 Schwinger at K = 10, <= 2e-15 rotor).  A heat-bath accept/reject decision that sits on
 *     such a difference flips, after which two chains diverge: compare, checkpoint and resume runs under
 *     one launch plan, and record it (K per launch: bench.py's config.overrelaxation_launches).  The
 *     sweep-by-sweep kernels (GFF always) agree bit for bit whatever the plan.  Spelled out in DESIGN.md 3 / 9.
 */
#ifndef MLMCPI_HIP_H
#define MLMCPI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLMCPI_ABI_VERSION 1

enum mlmcpi_status {
  MLMCPI_OK = 0,
  MLMCPI_ERR_INVALID = -1,     /* bad argument (sizes, parity, NULL pointers) */
  MLMCPI_ERR_HIP = -2,         /* HIP runtime error, see mlmcpi_last_error() */
  MLMCPI_ERR_UNSUPPORTED = -3, /* operation not defined for this action (action/action.hh:73-96) */
  MLMCPI_ERR_NO_DEVICE = -4
};

/* action kinds */
enum mlmcpi_action_kind {
  MLMCPI_HARMONIC = 0,  /* action/qm/harmonicoscillatoraction.{hh,cc} */
  MLMCPI_QUARTIC = 1,   /* action/qm/quarticoscillatoraction.{hh,cc} */
  MLMCPI_ROTOR = 2,     /* action/qm/rotoraction.{hh,cc} */
  MLMCPI_GFF = 3,       /* action/qft/gffaction.{hh,cc} (n_gibbs_smooth = 0) */
  MLMCPI_SCHWINGER = 4  /* action/qft/quenchedschwingeraction.{hh,cc} */
};

/* 1-D path action: lattice/lattice1d.hh:60-101 (M_lat, T_final, a = T_final/M_lat) + the action's
 * own parameters (m0, mu2; lambda, x0 for the quartic oscillator). */
typedef struct mlmcpi_path_action {
  int32_t kind;
  uint32_t M;
  double T_final, m0, mu2, lambda, x0;
} mlmcpi_path_action;

/* 2-D lattice action on an unrotated Mt x Mx periodic lattice (lattice/lattice2d.hh:98-437).
 * GFF uses `mass` (mu2 = (mass/Mt)^2, action/qft/gffaction.hh:174-181); Schwinger uses `beta`. */
typedef struct mlmcpi_lattice_action {
  int32_t kind;
  uint32_t Mt, Mx;
  double beta, mass;
} mlmcpi_lattice_action;

/* ---- runtime plumbing ---------------------------------------------------------------------- */
int mlmcpi_abi_version(void);
const char *mlmcpi_last_error(void);
int mlmcpi_device_count(int *count);
int mlmcpi_set_device(int device);
int mlmcpi_device_name(char *buf, size_t len);
int mlmcpi_malloc(void **d_ptr, size_t bytes);
int mlmcpi_free(void *d_ptr);
int mlmcpi_memset(void *d_ptr, int value, size_t bytes, void *stream);
int mlmcpi_copy_h2d(void *d_dst, const void *src, size_t bytes, void *stream);
int mlmcpi_copy_d2h(void *dst, const void *d_src, size_t bytes, void *stream);
int mlmcpi_copy_d2d(void *d_dst, const void *d_src, size_t bytes, void *stream);
int mlmcpi_stream_synchronize(void *stream);
/* Tuning knobs -- they change no result beyond the last bits.  Read from the environment once, at the first use in the
 * process; this call changes one afterwards (value "" or NULL resets it): MLMCPI_SWEEP_TILE=TWxTHxNT (tile and workgroup
 * size of the generic sweep kernels; also forces them), MLMCPI_OR_KERNEL=perm|block|lds|patch (Schwinger overrelaxation in
 * closed form -- the default where 64 x 64 or 64 x 32 tiles divide the lattice -- or sweep by sweep on 4 x 4 register
 * blocks, LDS resident, 2 x 2 register patches; the sweep-by-sweep kernels agree with each other bit for bit and with the
 * closed form to 4e-14), MLMCPI_OR_THREADS=256|512|1024 (workgroup size of the LDS-resident kernel), MLMCPI_OR_HEAT=
 * fused|split|wide|narrow (the heat-bath sweep behind the last overrelaxation launch: in it or in a launch of its own;
 * workgroup size of the fused launch). */
int mlmcpi_set_option(const char *name, const char *value);

/* ---- index maps (host, integer, bit-exact) --------------------------------------------------
 * lattice/lattice2d.hh:230-268 (vertex), :348-375 (link); rotated != 0 selects the 45-degree
 * sublattice numbering. */
uint32_t mlmcpi_vertex_cart2lin(uint32_t Mt, uint32_t Mx, int rotated, int i, int j);
void mlmcpi_vertex_lin2cart(uint32_t Mt, uint32_t Mx, int rotated, uint32_t ell, int *i, int *j);
uint32_t mlmcpi_link_cart2lin(uint32_t Mt, uint32_t Mx, int i, int j, int mu);
void mlmcpi_link_lin2cart(uint32_t Mt, uint32_t Mx, uint32_t ell, int *i, int *j, int *mu);
/* neighbour tables as the Lattice constructors build them: lattice/lattice1d.cc:11-18 (M x 2),
 * lattice/lattice2d.cc:137-155 (nvertices x 8; +i,-i,+j,-j then the four diagonals) */
int mlmcpi_neighbours_1d(uint32_t M, uint32_t *out);
int mlmcpi_neighbours_2d(uint32_t Mt, uint32_t Mx, int rotated, uint32_t *out);

/* ---- 1-D paths ------------------------------------------------------------------------------ */
/* Action::evaluate (action/action.hh:60-61): d_S[b] = S[x_b]. */
int mlmcpi_path_evaluate(const mlmcpi_path_action *act, const double *d_x, uint32_t B, double *d_S, void *stream);
/* Action::force (action/action.hh:112-113): d_f[b*M+j] = dS/dx_j. */
int mlmcpi_path_force(const mlmcpi_path_action *act, const double *d_x, double *d_f, uint32_t B, void *stream);
/* Action::initialise_state (action/action.hh:122-123): rotor U(-pi,pi) per site
 * (rotoraction.cc:82-89), zeros for HO / quartic. */
int mlmcpi_path_initialise(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint64_t seed, uint32_t chain0,
                           void *stream);
/* QoIXsquared::evaluate (qoi/qm/qoixsquared.cc:7-20) and QoISusceptibility::evaluate
 * (qoi/qm/qoisusceptibility.cc:8-23); d_out[b]. */
int mlmcpi_qoi_xsquared(const double *d_x, uint32_t M, uint32_t B, double *d_out, void *stream);
int mlmcpi_qoi_susceptibility(const double *d_x, uint32_t M, double T_final, uint32_t B, double *d_out,
                              void *stream);

/* HMCSampler::draw (sampler/hmcsampler.cc:8-19) = n_rep x single_step (:22-69), fused: momenta,
 * nt+1 force evaluations, both action evaluations, kinetic energies and the global Metropolis
 * test run on the device; momenta never touch HBM.  Repetition r of this call uses Philox step
 * traj0 + r; as in the reference, repetitions after the first acceptance are skipped.
 *   d_x        [B*M]  current states, updated in place where accepted
 *   d_work     workspace of mlmcpi_path_hmc_workspace_bytes() bytes
 *   d_accept   [B] int32, 1 where the draw was accepted (MCMCStep::accepted, mcmcstep.hh:47)
 *   d_energies optional [B*4]: S(x_cur), T(p_0), S(x_trial), T(p_end) of the last repetition run */
int mlmcpi_path_hmc_workspace_bytes(const mlmcpi_path_action *act, uint32_t B, uint32_t nt, size_t *bytes);
int mlmcpi_path_hmc_draw(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint32_t nt, double dt,
                         uint32_t n_rep, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                         int32_t *d_accept, double *d_energies, void *stream);

/* n_draws consecutive HMCSampler::draw calls, each followed by a QoI evaluation -- the body of the loop
 * of MonteCarloSingleLevel::evaluate (montecarlo/montecarlosinglelevel.cc:59-77) -- without returning
 * to the host: for paths that fit one workgroup (M <= 8192, multiple of 64) everything, including the
 * Metropolis tests and the QoIs, runs in ONE launch with the state in registers throughout.  Results equal
 * n_draws x (mlmcpi_path_hmc_draw with traj0 + d*n_rep, then the QoI kernel) up to fp contraction.
 *   qoi_kind  0 none, 1 QoIXsquared, 2 QoISusceptibility;  d_qoi [B*n_draws] (chain-major) for the
 *   one-launch path, [n_draws*B] (draw-major) for segmented paths -- see mlmcpi_path_hmc_run_layout();
 *   d_accept_count [B] int32 (optional): accepted draws per chain. */
int mlmcpi_path_hmc_run(const mlmcpi_path_action *act, double *d_x, uint32_t B, uint32_t nt, double dt, uint32_t n_rep,
                        uint32_t n_draws, int qoi_kind, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                        double *d_qoi, int32_t *d_accept_count, void *stream);
/* 1 if d_qoi of mlmcpi_path_hmc_run is chain-major [B][n_draws] for this action / nt, 0 if draw-major */
int mlmcpi_path_hmc_run_layout(const mlmcpi_path_action *act, uint32_t B, uint32_t nt, int32_t *chain_major);

/* OverrelaxedHeatBathSampler::draw (sampler/overrelaxedheatbathsampler.cc:8-31) for the rotor:
 * n_overrelax sweeps of RotorAction::overrelaxation_update (rotoraction.cc:40-56) then n_heatbath
 * sweeps of heatbath_update (:20-37), even sites then odd sites within each sweep.  Sweep s of
 * this call uses Philox step sweep0 + s.  d_x is updated in place; d_scratch is B*M doubles.
 * Up to 16 overrelaxation sweeps of a launch are applied in closed form (more: launches of equal depth);
 * MLMCPI_OR_KERNEL=block sweeps one by one, 8 per launch.  The two agree to <= 2e-15 without a heat bath
 * behind (the same map, other rounding), NOT bit for bit: see "bit reproducibility across launch plans"
 * at the top of this header. */
int mlmcpi_path_sweep_draw(const mlmcpi_path_action *act, double *d_x, double *d_scratch, uint32_t B,
                           uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                           uint32_t sweep0, void *stream);

/* The same with the input left untouched (see mlmcpi_lattice_sweep_draw_from): reads d_src, alternates between d_w0 and
 * d_w1 (which may equal d_src), *result_in = 0 / 1 names the buffer holding the result; no final copy. */
int mlmcpi_path_sweep_draw_from(const mlmcpi_path_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                                uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                                int32_t *result_in, void *stream);
/* mlmcpi_path_sweep_draw_from with the topological susceptibility of the new sample (QoISusceptibility,
 * qoi/qm/qoisusceptibility.cc:8-23: chi = Q^2 / T, Q = sum_j mod_2pi(x_j - x_{j-1}) / 2 pi) summed inside the draw's last
 * launch, while the segments are in LDS: d_qoi[b]; and, with d_acc != NULL, stats->record_sample of it into d_acc[B][5] in
 * the same call (mlmcpi_stats_accumulate's recurrence): one pass of the loop at montecarlo/montecarlosinglelevel.cc:59-77.
 * Same value as mlmcpi_qoi_susceptibility on the result up to the order of the summation. */
int mlmcpi_path_sweep_draw_qoi(const mlmcpi_path_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                               uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                               double *d_qoi, double *d_acc, int32_t *result_in, void *stream);

/* Action::heatbath_update / overrelaxation_update(state, l) (action/action.hh:73-96; rotoraction.cc:20-56): the update of
 * site l for every chain of the batch -- of the n sites d_sites[0 .. n) (device memory) in list order when d_sites is not
 * NULL, of the single site `site` otherwise.  One thread per chain walks the list, so the updates are sequential within
 * a chain exactly as in the reference's loops (overrelaxedheatbathsampler.cc:8-31: lexicographic or shuffled index
 * sets); chains run in parallel.  heat != 0: heat bath, else overrelaxation.  Random numbers: Philox (site, chain,
 * step), the sweeps' contract, so the sites of one colour visited with a sweep's step reproduce that phase of the sweep.
 * This is the reference's CPU inner loop kept for callers that walk index sets themselves; the fast path is a sweep. */
int mlmcpi_path_site_updates(const mlmcpi_path_action *act, double *d_x, uint32_t B, const uint32_t *d_sites, uint32_t n,
                             uint32_t site, int32_t heat, uint64_t seed, uint32_t chain0, uint32_t step, void *stream);

/* TwoLevelMetropolisStep::draw (montecarlo/twolevelmetropolisstep.cc:35-89) with QMAction::copy_from_{coarse,fine}
 * (action/qm/qmaction.cc:7-24) and the action's conditioned fine action: Gaussian for the harmonic / quartic
 * oscillator (action/qm/gaussianconditionedfineaction.cc:7-43), ExpSin2 for the rotor
 * (action/qm/rotorconditionedfineaction.cc:7-43; theta'[2j+1] = mod_2pi(Wmin + ExpSin2(2 W''))):
 *   theta'[2j] = x_coarse[j];  theta'[2j+1] ~ N(Wmin(theta'[2j], theta'[2j+2]), 1/W'')
 *   dS = [S_f(theta') - S_f(theta)] + [S_c(theta_C) - S_c(x_coarse)] + [S_cfa(theta) - S_cfa(theta')]
 *   accept with min(1, exp(-dS)); accepted chains get theta <- theta'.
 * `fine` lives on M sites, `coarse` on M/2 (its parameters are the caller's: same as fine for the quartic
 * oscillator, renormalised or not for the HO).  d_theta [B*M] is the step's current fine state
 * (MCMCStep::set_state), d_x_coarse [B*M/2] the coarse-level proposal.  d_terms (optional, [B*3]) receives
 * the three action differences.  Philox step = `step`. */
int mlmcpi_path_twolevel_workspace_bytes(const mlmcpi_path_action *fine, uint32_t B, size_t *bytes);
int mlmcpi_path_twolevel_draw(const mlmcpi_path_action *fine, const mlmcpi_path_action *coarse, const double *d_x_coarse,
                              double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step, void *d_work,
                              int32_t *d_accept, double *d_terms, void *stream);

/* The same with a per-chain mask (d_mask may be NULL = all ones; d_mask may be the d_accept of the level below): chains
 * with d_mask[b] == 0 are left alone -- d_accept[b] = 0, d_theta[b] untouched, no random numbers of the step consumed for
 * them.  This is the `if (not accept) break` of HierarchicalSampler::draw (sampler/hierarchicalsampler.cc:62-76) for a
 * batch of chains: a chain whose move was rejected on a coarser level does not move on the finer ones. */
int mlmcpi_path_twolevel_draw_masked(const mlmcpi_path_action *fine, const mlmcpi_path_action *coarse, const double *d_x_coarse,
                                     double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step, void *d_work,
                                     const int32_t *d_mask, int32_t *d_accept, double *d_terms, void *stream);

/* QMAction::copy_from_fine / copy_from_coarse (action/qm/qmaction.cc:7-24): coarse[j] <-> fine[2j]; the odd
 * fine sites are left untouched.  d_fine [B*2*M_coarse], d_coarse [B*M_coarse]. */
int mlmcpi_path_copy_from_fine(const double *d_fine, double *d_coarse, uint32_t M_coarse, uint32_t B, void *stream);
int mlmcpi_path_copy_from_coarse(const double *d_coarse, double *d_fine, uint32_t M_coarse, uint32_t B, void *stream);
/* Exact sampler of the harmonic oscillator (the action is its own Sampler in the reference).
 * mlmcpi_ho_cholesky_factor: HarmonicOscillatorAction::build_covariance (action/qm/harmonicoscillatoraction.cc:38-56)
 *   on the host: lower Cholesky factor L of the covariance = inverse of the circulant precision matrix, O(M^3);
 *   h_LT [M][M] (host memory) receives L^T row-major, h_LT[k*M + j] = L[j][k].  M_lat <= 4096.
 * mlmcpi_path_exact_draw: HarmonicOscillatorAction::draw (:59-66): d_x[b] = L y[b], y ~ N(0,1)^M from Philox
 *   (entries 2m, 2m+1 = the Box-Muller pair of site m, purpose P_EXACT, step `step`), for B chains -- a dense
 *   [B x M] . [M x M] fp64 product on the matrix cores.  d_LT = h_LT copied to the device. */
int mlmcpi_ho_cholesky_factor(const mlmcpi_path_action *act, double *h_LT);
int mlmcpi_path_exact_draw(const mlmcpi_path_action *act, const double *d_LT, double *d_x, uint32_t B, uint64_t seed,
                           uint32_t chain0, uint32_t step, void *stream);

/* ---- 2-D lattices --------------------------------------------------------------------------- */
int mlmcpi_lattice_state_size(const mlmcpi_lattice_action *act, uint32_t *n); /* Action::sample_size */
int mlmcpi_lattice_evaluate(const mlmcpi_lattice_action *act, const double *d_phi, uint32_t B, double *d_S,
                            void *stream);
int mlmcpi_lattice_force(const mlmcpi_lattice_action *act, const double *d_phi, double *d_f, uint32_t B,
                         void *stream);
/* Schwinger: U(-pi,pi) per link (quenchedschwingeraction.cc:198-204).  GFF: an exact draw from the action's
 * distribution, as in the reference (gffaction.cc:121-123: initialise_state = draw) -- by spectral synthesis
 * (see mlmcpi_lattice_exact_draw; its own Philox sub-stream) instead of the reference's sparse Cholesky solve. */
int mlmcpi_lattice_initialise(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed,
                              uint32_t chain0, void *stream);
/* OverrelaxedHeatBathSampler::draw on a 2-D action: n_overrelax overrelaxation sweeps then
 * n_heatbath heat-bath sweeps (gffaction.cc:33-42,68-77; quenchedschwingeraction.cc:25-65 with
 * distribution/expcosdistribution.hh:51-65), multicolour order (GFF: (i+j) even, odd; Schwinger:
 * mu=0 & j even, mu=0 & j odd, mu=1 & i even, mu=1 & i odd).  Mt and Mx must be even.  Sweep s
 * uses Philox step sweep0 + s.  d_phi is updated in place; d_scratch has the same size.
 * `fuse` = max number of consecutive overrelaxation sweeps fused into one launch (0 = library default: Schwinger closed
 * form 10; register-block kernels 6, in launches of equal depth, on lattices that 64 x 64 tiles divide; 4 otherwise.  The
 * heat-bath sweep behind the last overrelaxation launch rides in it where the fused kernels apply, else it gets a launch
 * of its own); results do not depend on it beyond the last bits of the Schwinger closed form (see the contract above). */
int mlmcpi_lattice_sweep_draw(const mlmcpi_lattice_action *act, double *d_phi, double *d_scratch, uint32_t B,
                              uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                              uint32_t sweep0, uint32_t fuse, void *stream);
/* Action::heatbath_update / overrelaxation_update(state, l) on a 2-D action (gffaction.cc:33-42,68-77;
 * quenchedschwingeraction.cc:46-65): as mlmcpi_path_site_updates; l is a vertex (GFF) or link (Schwinger) index. */
int mlmcpi_lattice_site_updates(const mlmcpi_lattice_action *act, double *d_state, uint32_t B, const uint32_t *d_sites,
                                uint32_t n, uint32_t site, int32_t heat, uint64_t seed, uint32_t chain0, uint32_t step,
                                void *stream);
/* Same, without the final device-to-device copy: the sweeps ping-pong between d_a (input) and d_b;
 * *result_in_b tells the caller which buffer holds the result (swap your pointers when it is 1). */
int mlmcpi_lattice_sweep_draw_pingpong(const mlmcpi_lattice_action *act, double *d_a, double *d_b, uint32_t B,
                                       uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                       uint32_t sweep0, uint32_t fuse, int32_t *result_in_b, void *stream);
/* Same, with the input left untouched: the first launch reads d_src (never written), the launches then alternate between
 * the work buffers d_w0 and d_w1; *result_in = 0 / 1 names the work buffer that holds the result.  d_w1 may equal
 * d_src (then this is the ping-pong form).  This is what Sampler::draw(out) needs to hand `out` the new sample without a
 * copy while the caller still holds the previous one (sampler/overrelaxedheatbathsampler.cc:30 copies the state out). */
int mlmcpi_lattice_sweep_draw_from(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1,
                                   uint32_t B, uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                   uint32_t sweep0, uint32_t fuse, int32_t *result_in, void *stream);
/* mlmcpi_lattice_sweep_draw_from with the QoI of the new sample summed inside the draw's last launch, while the tile is
 * still in LDS: Sampler::draw + QoI::evaluate of the loop at montecarlo/montecarlosinglelevel.cc:59-77 in one pass over
 * the state instead of two.  qoi_kind 1 = QoIAvgPlaquette (qoi/qft/qoiavgplaquette.cc:8-27), 2 = QoI2DSusceptibility
 * (qoi/qft/qoi2dsusceptibility.cc:8-27), both for the quenched Schwinger action; 3 = QoI2DPhiSquared
 * (qoi/qft/qoi2dphisquared.cc:8-15) for the GFF action; d_qoi[b].  n_heatbath >= 1 (the draw has to end with a heat-bath
 * sweep) and a QoI of the action at hand: MLMCPI_ERR_UNSUPPORTED otherwise, and the caller evaluates the QoI
 * separately.  Same values as mlmcpi_qoi_* on the result up to the order of the summation. */
int mlmcpi_lattice_sweep_draw_qoi(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1, uint32_t B,
                                  uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0, uint32_t sweep0,
                                  uint32_t fuse, int32_t qoi_kind, double *d_qoi, int32_t *result_in, void *stream);
/* One pass of the sampling loop at montecarlo/montecarlosinglelevel.cc:59-77 in one call: sampler->draw, qoi->evaluate and
 * stats->record_sample -- mlmcpi_lattice_sweep_draw_qoi followed by mlmcpi_stats_accumulate(d_acc, d_qoi, B), with the
 * moments updated by the launch that finishes the QoI (one launch less per sample; same values).  d_acc[B][5] as for
 * mlmcpi_stats_accumulate. */
int mlmcpi_lattice_sweep_draw_qoi_record(const mlmcpi_lattice_action *act, const double *d_src, double *d_w0, double *d_w1,
                                         uint32_t B, uint32_t n_overrelax, uint32_t n_heatbath, uint64_t seed, uint32_t chain0,
                                         uint32_t sweep0, uint32_t fuse, int32_t qoi_kind, double *d_qoi, double *d_acc,
                                         int32_t *result_in, void *stream);
/* Action::copy_from_fine / copy_from_coarse between a lattice and its next-coarser level, coarsening
 * factors rt, rx in {1, 2} in the temporal / spatial direction (CoarsenBoth = 2,2; CoarsenTemporal = 2,1;
 * CoarsenSpatial = 1,2; lattice/lattice2d.cc:24-47).  `fine` describes the FINE lattice.
 *   Schwinger (quenchedschwingeraction.cc:92-195): coarse links = mod_2pi(sum of the fine links they span);
 *     copy_from_coarse halves a coarse link over its two fine links and leaves the fine-only links untouched.
 *   GFF (gffaction.cc:97-118): vertices (rt i, rx j) <-> (i, j). */
int mlmcpi_lattice_copy_from_fine(const mlmcpi_lattice_action *fine, uint32_t rt, uint32_t rx, const double *d_fine,
                                  double *d_coarse, uint32_t B, void *stream);
int mlmcpi_lattice_copy_from_coarse(const mlmcpi_lattice_action *fine, uint32_t rt, uint32_t rx, const double *d_coarse,
                                    double *d_fine, uint32_t B, void *stream);
/* TwoLevelMetropolisStep::draw (montecarlo/twolevelmetropolisstep.cc:35-89) on the quenched Schwinger lattice:
 * copy_from_coarse, the conditioned fine action's fill_fine_points and evaluate, copy_from_fine, the three action
 * differences, the Metropolis test and the copy of accepted states.  Conditioned fine action by coarsening
 * (quenchedschwingerconditionedfineaction.hh:218-238):
 *   one direction halved (CoarsenTemporal / CoarsenSpatial / levels of CoarsenAlternate):
 *     QuenchedSchwingerSemiConditionedFineAction (.cc:130-204, 332-379): uniform shifts + ExpCos draws;
 *   both directions halved (CoarsenBoth): QuenchedSchwingerConditionedFineAction (.cc:7-78, 207-289):
 *     BesselProductDistribution for beta <= 8, ApproximateBesselProductDistribution beyond.
 * `coarse` carries the coarse lattice extents and the coarse beta (QuenchedSchwingerAction::coarse_action).
 *   d_phi_coarse [B][2 Mt_c Mx_c]  coarse-level proposal;  d_theta [B][2 Mt Mx]  current fine state (updated when
 *   accepted);  d_accept [B];  d_terms [B][3] = (dS_fine, dS_coarse, dS_trial) or NULL. */
int mlmcpi_lattice_twolevel_workspace_bytes(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse,
                                            uint32_t B, size_t *bytes);
int mlmcpi_lattice_twolevel_draw(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse,
                                 const double *d_phi_coarse, double *d_theta, uint32_t B, uint64_t seed,
                                 uint32_t chain0, uint32_t step, void *d_work, int32_t *d_accept, double *d_terms,
                                 void *stream);
/* The same step with the conditioned fine action chosen by the caller: cfa_kind 0 = what the reference's factory picks
 * for the lattice (quenchedschwingerconditionedfineaction.hh:218-238), 1 = QuenchedSchwingerGaussianConditionedFineAction
 * (quenchedschwingerconditionedfineaction.cc:81-134, 293-327; lattices coarsened in both directions): uniform splits of
 * the coarse links, the four interior links of every 2 x 2 block from GaussianFillinDistribution
 * (distribution/gaussianfillindistribution.{hh,cc}; Philox purpose 13 of the coarse cell). */
int mlmcpi_lattice_twolevel_draw_cfa(const mlmcpi_lattice_action *fine, const mlmcpi_lattice_action *coarse, int32_t cfa_kind,
                                     const double *d_phi_coarse, double *d_theta, uint32_t B, uint64_t seed, uint32_t chain0,
                                     uint32_t step, void *d_work, int32_t *d_accept, double *d_terms, void *stream);
/* Exact sampler of the Gaussian free field: GFFAction::draw / initialise_state (action/qft/gffaction.cc:121-123,
 * 200-213; the reference goes through a sparse Cholesky factor built by Eigen, which it cannot build beyond ~64^2).
 * On the periodic lattice the precision matrix is diagonal in Fourier space, so the draw is a spectral synthesis:
 *   phi(x) = Re sum_k (n0_k + i n1_k) e^{+i k x} / sqrt(N lambda(k)),  lambda(k) = 4 + mu2 - 2 cos(2 pi k_t/Mt) - 2 cos(2 pi k_x/Mx),
 * normals of mode l = k_x Mt + k_t from Philox (site l, purpose P_EXACT, step `step`); batched 2-D inverse FFT (hipFFT).
 * d_work: mlmcpi_lattice_exact_workspace_bytes (one complex field per chain). */
int mlmcpi_lattice_exact_workspace_bytes(const mlmcpi_lattice_action *act, uint32_t B, size_t *bytes);
int mlmcpi_lattice_exact_draw(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0,
                              uint32_t step, void *d_work, void *stream);
/* ---- Gaussian free field: levels of a coarsening hierarchy and the two-level step between them ------
 * A level = GFFAction(lattice, fine_lattice, mass, n_gibbs_smooth, omega) (action/qft/gffaction.hh:120-146) on the lattice
 * (Mt, Mx, coarsening type, coarsening level); CoarsenRotate levels of odd depth are rotated lattices
 * (lattice/lattice2d.hh:98-437).  The reference's coarse_action() is (coarse lattice, mass, n_gibbs_smooth = 2, omega = 1)
 * (gffaction.hh:201-208).  Levels with n_gibbs_smooth > 0 evaluate 1/2 phi^T Qhat phi with the dense smoothed precision
 * matrix of gffaction.cc:126-173 (built on the host, levels of up to 4096 vertices); index maps come from tables like the
 * reference's.  The finest level of a run keeps using mlmcpi_lattice_* (stencil kernels, FFT sampler). */
typedef struct mlmcpi_gff_level mlmcpi_gff_level;
int mlmcpi_gff_level_create(uint32_t Mt, uint32_t Mx, int32_t coarsening_type, int32_t level, double mass, int32_t n_gibbs_smooth,
                            double omega, mlmcpi_gff_level **out);
int mlmcpi_gff_level_destroy(mlmcpi_gff_level *level);
/* sample_size; number of vertices / extents of the next-coarser lattice (0 if there is none); mu2 (gffaction.hh:174-181) */
int mlmcpi_gff_level_info(const mlmcpi_gff_level *level, uint32_t *n_vertices, uint32_t *n_coarse, uint32_t *Mt_coarse,
                          uint32_t *Mx_coarse, double *mu2);
/* lattice2d.cc:82-134 (host): pairs[2 n_coarse] = (fine index, coarse index) ascending in the fine index = fine2coarse_map;
 * fineonly[n_vertices - n_coarse] ascending = fineonly_vertices */
int mlmcpi_gff_level_tables(const mlmcpi_gff_level *level, uint32_t *pairs, uint32_t *fineonly);
/* host copies of the dense matrices, row major [N][N]: which = 0 Qhat (gffaction.cc:165-167), 1 inverse of the Cholesky
 * factor of the plain precision matrix (the exact sampler, gffaction.cc:169-173) */
int mlmcpi_gff_level_matrix(mlmcpi_gff_level *level, int32_t which, double *h_out);
/* GFFAction::evaluate (gffaction.cc:8-30) */
int mlmcpi_gff_level_evaluate(mlmcpi_gff_level *level, const double *d_phi, uint32_t B, double *d_S, void *stream);
/* GFFAction::draw (gffaction.cc:200-213): exact draw + n_gibbs_smooth lexicographic sweeps of
 * global_heatbath_update_eff (:45-66); Philox (pair l >> 1, branch l & 1) with purpose 12 (white noise) / 11 (sweep k = sub) */
int mlmcpi_gff_level_draw(mlmcpi_gff_level *level, double *d_phi, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step,
                          void *stream);
/* GFFAction::copy_from_fine / copy_from_coarse (gffaction.cc:97-118); `fine` is the finer of the two levels */
int mlmcpi_gff_copy_from_fine(mlmcpi_gff_level *fine, const double *d_fine, double *d_coarse, uint32_t B, void *stream);
int mlmcpi_gff_copy_from_coarse(mlmcpi_gff_level *fine, const double *d_coarse, double *d_fine, uint32_t B, void *stream);
/* GFFConditionedFineAction (gffconditionedfineaction.cc:7-49): fill_fine_points (fine-only vertex l ~ N(sigma^2 Delta,
 * sigma^2), normal = Philox(site l, purpose 7); d_S[b] = the conditioned action of the filled state) and evaluate */
int mlmcpi_gff_cfa_fill(mlmcpi_gff_level *fine, double *d_state, uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step,
                        double *d_S, void *stream);
int mlmcpi_gff_cfa_evaluate(mlmcpi_gff_level *fine, const double *d_state, uint32_t B, double *d_S, void *stream);
/* TwoLevelMetropolisStep::draw (twolevelmetropolisstep.cc:35-89) for a GFF level and its coarsening: d_theta is the current
 * fine state (updated in place on acceptance), d_terms[b] = (dS_fine, dS_coarse, dS_trial) (may be NULL) */
int mlmcpi_gff_twolevel_workspace_bytes(const mlmcpi_gff_level *fine, uint32_t B, size_t *bytes);
int mlmcpi_gff_twolevel_draw(mlmcpi_gff_level *fine, mlmcpi_gff_level *coarse, const double *d_phi_coarse, double *d_theta,
                             uint32_t B, uint64_t seed, uint32_t chain0, uint32_t step, void *d_work, int32_t *d_accept,
                             double *d_terms, void *stream);

/* QoI2DPhiSquared (qoi/qft/qoi2dphisquared.cc:8-15), QoIAvgPlaquette (qoi/qft/qoiavgplaquette.cc:8-27),
 * QoI2DSusceptibility (qoi/qft/qoi2dsusceptibility.cc:8-27); d_out[b]. */
int mlmcpi_qoi_phi_squared(const double *d_phi, uint32_t n_vertices, uint32_t B, double *d_out, void *stream);
int mlmcpi_qoi_avg_plaquette(const double *d_theta, uint32_t Mt, uint32_t Mx, uint32_t B, double *d_out,
                             void *stream);
int mlmcpi_qoi_2d_susceptibility(const double *d_theta, uint32_t Mt, uint32_t Mx, uint32_t B, double *d_out,
                                 void *stream);

/* Generic HMCSampler::draw for a 2-D action (streaming leapfrog: one fused force + momentum +
 * position kernel per step).  Same contract as mlmcpi_path_hmc_draw. */
int mlmcpi_lattice_hmc_workspace_bytes(const mlmcpi_lattice_action *act, uint32_t B, size_t *bytes);
int mlmcpi_lattice_hmc_draw(const mlmcpi_lattice_action *act, double *d_phi, uint32_t B, uint32_t nt, double dt,
                            uint32_t n_rep, uint64_t seed, uint32_t chain0, uint32_t traj0, void *d_work,
                            int32_t *d_accept, double *d_energies, void *stream);

/* ---- host-only analytic helpers of the quenched Schwinger model (no GPU needed) --------------------------
 * V chi_t(beta, P) = (P / beta) Phi_chi(beta, P) (common/auxilliary.cc:30-33, 44-79, 98-193: the value
 * QoI2DSusceptibility::evaluate's expectation is compared with, qoi/qft/qoi2dsusceptibility.cc:30-34), and the coarse
 * coupling matched to it: the root x of chi_t(x beta, P / rho) = chi_t(beta, P) in [0.01, 2] by bisection, times beta
 * (action/qft/quenchedschwingerrenormalisation.cc:7-64; rho = 4 when both directions are coarsened, else 2; the
 * reference's fall-back x = 1 / rho when the interval holds no root).  Own quadrature in place of GSL's. */
int mlmcpi_schwinger_chit_analytical(double beta, uint32_t n_plaq, double *chit);
int mlmcpi_schwinger_beta_coarse_nonperturbative(double beta, uint32_t n_plaq, int32_t rho_refine, double *beta_coarse);

/* ---- device-side statistics accumulation (common/statistics.cc:4-27, batched) ---------------
 * d_acc holds, per chain, the packed sums [n, sum q, sum q^2, sum q^3, sum q^4] that the
 * cross-rank reduction (one RCCL all-reduce of B*5 doubles) combines; see DESIGN.md. */
int mlmcpi_stats_accumulate(double *d_acc, const double *d_q, uint32_t B, void *stream);
/* Statistics::record_sample WITH the autocorrelation window (common/statistics.cc:4-27: running average, running averages
 * S_k of Q_j Q_{j-k} for k < window over a deque of the last `window` values), batched: d_state[B][2 * window + 3] =
 * per chain [n, average, S_0 .. S_{window-1}, head, ring[window]], zero-initialised by the caller.  What
 * MonteCarloMultiLevel::draw_coarse_sample (montecarlo/montecarlomultilevel.cc:170-190) re-reads on every coarse draw:
 * tau_int = max(1, 1 + 2 sum_{k>=1} (1 - k/n)(S_k - avg^2)/(S_0 - avg^2)) (statistics.cc:38-61), computed by the caller
 * from the state (per chain, or with the autocovariances averaged over the chains of a batch). */
int mlmcpi_stats_window_record(double *d_state, const double *d_q, uint32_t B, uint32_t window, void *stream);

/* ---- test hooks: raw RNG streams, single draws (used by parity tests only) -------------------- */
int mlmcpi_test_philox(const uint32_t *ctr4, const uint32_t *key2, uint32_t *out4);
/* d_out[4*k..] = (u0,u1,n0,n1) for site k of n sites */
int mlmcpi_test_random(uint64_t seed, uint32_t chain, uint32_t step, uint32_t purpose, uint32_t sub, uint32_t n,
                       double *d_out, void *stream);
/* d_out[k] = ExpCos draw for site k with staples (d_xp[k], d_xm[k]); ExpSin2 draw with d_sigma[k] */
int mlmcpi_test_expcos(uint64_t seed, uint32_t chain, uint32_t step, double beta, const double *d_xp,
                       const double *d_xm, uint32_t n, double *d_out, void *stream);
int mlmcpi_test_expsin2(uint64_t seed, uint32_t chain, uint32_t step, const double *d_sigma, uint32_t n,
                        double *d_out, void *stream);
/* d_out[k] = a heat-bath draw for site k of the stream between x_p = d_xp[k] and x_m = d_xm[k], conditional
 * exp(scale / 2 [cos(x - x_p) + cos(x - x_m)]), from the tabulated step-envelope sampler the sweeps use for actions with
 * scale = 2 beta (Schwinger) or 2 m0 / a (rotor) <= 16 (the range the sweeps draw from this sampler: round 5; 4 before) */
int mlmcpi_test_vs_draw(uint64_t seed, uint32_t chain, uint32_t step, double scale, const double *d_xp, const double *d_xm,
                        uint32_t n, double *d_out, void *stream);
/* that sampler's table for an action of the given scale (host only, no GPU needed): sel[8][64] = bin of a selector
 * value per concentration class, lw[8][8] = log2 of the acceptance factor of a bin */
int mlmcpi_vs_table(double scale, uint8_t *sel, float *lw);

#ifdef __cplusplus
}
#endif
#endif /* MLMCPI_HIP_H */
