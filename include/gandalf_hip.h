/* gandalf_hip.h -- C ABI of libgandalf_hip.so: the MI355X (gfx950) replacement for GANDALF's
 * grad-h SPH + KD-tree gravity hot path.
 *
 * The reference (SJaffa/gandalf v0.4.0) has no FFI for this path: its seam is the set of C++
 * virtual interfaces picked by string parameters in GradhSphSimulation::ProcessSphParameters
 * (reference src/GradhSph/GradhSphSimulation.cpp:129-246).  Each entry point below names the
 * reference member function(s) it replaces.  A reference-side shell class that binds them is
 * shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every pointer is a HOST pointer unless the
 *     name ends in _dev.
 *   - all entry points are synchronous at return and must be called from one host thread per
 *     context (the reference's callers are serial: SphSimulation::MainLoop).
 *   - return value: 0 ok, >0 recoverable (GH_ERR_CAPACITY ...), <0 fatal; gh_last_error() gives
 *     the message (the host shell maps fatal codes to ExceptionHandler::raise, Exception.cpp:50).
 *   - particle arrays cross the boundary in the caller's order ("iorig order"); the library keeps
 *     its own tree-ordered SoA copy on the device.
 *   - precision: double only (reference PRECISION=DOUBLE, Precision.h:44-50).
 */
#ifndef GANDALF_HIP_H
#define GANDALF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gh_ctx gh_ctx;

enum { GH_OK = 0, GH_ERR_CAPACITY = 1, GH_ERR_NOTCONVERGED = 2,
       GH_ERR_INVALID = -1, GH_ERR_HIP = -2, GH_ERR_UNSUPPORTED = -3 };

/* boundary types per face (reference DomainBox.h boundaryEnum: open / periodic / mirror / wall) */
enum { GH_BOUNDARY_OPEN = 0, GH_BOUNDARY_PERIODIC = 1, GH_BOUNDARY_MIRROR = 2 };
/* kernels (reference SmoothingKernel.h: M4Kernel :101-240, QuinticKernel :251-408, TabulatedKernel :547-756).
 * GH_KERNEL_M4_TAB = the reference's default pair kernel = m4, tabulated_kernel = 1: 1000-entry piecewise-constant
 * tables of the M4 functions (TabulatedKernel.cpp:57-100); GH_KERNEL_QUINTIC_TAB likewise for kernel = quintic. */
enum { GH_KERNEL_M4 = 0, GH_KERNEL_QUINTIC = 1, GH_KERNEL_M4_TAB = 2, GH_KERNEL_QUINTIC_TAB = 3 };
/* gas_eos (reference EOS.h; energy_eqn = AdiabaticEOS.cpp, isothermal = IsothermalEOS.cpp,
 * barotropic = BarotropicEOS.cpp) */
enum { GH_EOS_ENERGY_EQN = 0, GH_EOS_ISOTHERMAL = 1, GH_EOS_BAROTROPIC = 2 };
/* avisc / acond (reference Sph.h aviscenum, acondenum) */
enum { GH_AVISC_NONE = 0, GH_AVISC_MON97 = 1, GH_AVISC_MON97MM97 = 2 /* avisc = mon97 + time_dependent_avisc = mm97 */,
       GH_AVISC_MON97CD2010 = 3 /* + time_dependent_avisc = cd2010 (Sph.h:364-456); hydro only */ };
enum { GH_ACOND_NONE = 0, GH_ACOND_WADSLEY2008 = 1, GH_ACOND_PRICE2008 = 2 };
/* multipole / gravity_mac (reference Tree.h MAC_Type; NeighbourSearch.h:350-475) */
enum { GH_MULTIPOLE_MONOPOLE = 0, GH_MULTIPOLE_QUADRUPOLE = 1, GH_MULTIPOLE_FAST_MONOPOLE = 2 /* NeighbourSearch.h:481-794 */,
       GH_MULTIPOLE_FAST_QUADRUPOLE = 3 /* NeighbourSearch.h:601-720, 796-827 */ };
enum { GH_MAC_GEOMETRIC = 0, GH_MAC_GADGET2 = 1, GH_MAC_EIGENMAC = 2 };   /* Tree.h:413-432 open_cell_for_gravity */

/* Parameter block: the hot-path subset of the reference's parameter file
 * (Parameters.cpp:160-520; key names in comments). */
typedef struct gh_config {
  int32_t ndim;            /* ndim: 1, 2 or 3 */
  int32_t kernel;          /* GH_KERNEL_*: kernel = m4 | quintic, each with tabulated_kernel = 0 | 1 */
  int32_t gas_eos;         /* gas_eos */
  int32_t avisc;           /* avisc */
  int32_t acond;           /* acond */
  int32_t self_gravity;    /* self_gravity */
  int32_t hydro_forces;    /* hydro_forces */
  int32_t multipole;       /* multipole */
  int32_t gravity_mac;     /* gravity_mac */
  int32_t Nleafmax;        /* Nleafmax */
  int32_t energy_integration; /* 1 if gas_eos = energy_eqn and energy_integration != none */
  int32_t device;          /* HIP device ordinal */
  int32_t boundary_lhs[3]; /* boundary_lhs[k] */
  int32_t boundary_rhs[3]; /* boundary_rhs[k] */
  int32_t Nlevels;         /* Nlevels: 1 = global timestep, > 1 = hierarchical block timesteps (Simulation.cpp:1764-2200) */
  int32_t level_diff_max;  /* level_diff_max: largest level difference tolerated between SPH neighbours (SphLeapfrogKDK.cpp:284-330) */
  int32_t ntreebuildstep;  /* ntreebuildstep: the tree is rebuilt every ntreebuildstep steps (and on the first step after the setup) and
                            * re-stocked in between (HydroTree::BuildTree, HydroTree.cpp:325-343); <= 1: rebuilt every step */
  int32_t ntreestockstep;  /* ntreestockstep: between rebuilds the tree is re-stocked every ntreestockstep steps and its cells drift with
                            * their mean velocity on the others (Tree::ExtrapolateCellProperties, Tree.cpp:172-198; faithful while
                            * no particle has left its cell's drifted box) */
  int32_t sph_single_timestep; /* sph_single_timestep: with Nlevels > 1 all gas particles share the highest occupied level (Simulation.cpp:1890-1900, 2090-2096) */
  int32_t reserved_;
  double  boxmin[3];       /* boxmin[k] */
  double  boxmax[3];       /* boxmax[k] */
  double  h_fac;           /* h_fac */
  double  h_converge;      /* h_converge */
  double  alpha_visc;      /* alpha_visc */
  double  beta_visc;       /* beta_visc */
  double  gamma_eos;       /* gamma_eos */
  double  temp0;           /* temp0   (isothermal / barotropic) */
  double  mu_bar;          /* mu_bar */
  double  rho_bary;        /* rho_bary (barotropic) */
  double  thetamaxsqd;     /* thetamaxsqd */
  double  courant_mult;    /* courant_mult */
  double  accel_mult;      /* accel_mult */
  double  energy_mult;     /* energy_mult */
  double  macerror;        /* macerror (gravity_mac = gadget2 / eigenmac) */
  double  alpha_visc_min;  /* alpha_visc_min (time_dependent_avisc = mm97) */
  /* sink particles (SphSimulation.cpp:116-136; all zero = no sinks).  Sink runs go through gh_hybrid_setup / gh_hybrid_step
   * with a gh_nbody context that holds the sinks' stars (it may start empty); global timestep, tree rebuilt every step. */
  int32_t sink_particles;  /* sink_particles */
  int32_t create_sinks;    /* create_sinks */
  int32_t smooth_accretion;/* smooth_accretion */
  int32_t sink_radius_mode;/* sink_radius_mode: 0 fixed, 1 hmult, 2 anything else (radius = kernrange*h) */
  int32_t Nsinkfixed;      /* Nsinkfixed (-1: no limit) */
  int32_t reserved2_;
  double  rho_sink;        /* rho_sink, in code units (SphSimulation.cpp:128-129) */
  double  sink_radius;     /* sink_radius (code units for mode fixed, multiples of h for hmult) */
  double  alpha_ss;        /* alpha_ss */
  double  smooth_accrete_frac; /* smooth_accrete_frac */
  double  smooth_accrete_dt;   /* smooth_accrete_dt */
} gh_config;

/* field ids for gh_download / gh_upload_field (values are per particle; vectors are [N][ndim]) */
enum {
  GH_F_R = 0, GH_F_V, GH_F_A, GH_F_ATREE, GH_F_R0, GH_F_V0, GH_F_A0,          /* vectors */
  GH_F_M, GH_F_H, GH_F_U, GH_F_U0, GH_F_DUDT, GH_F_DUDT0, GH_F_RHO, GH_F_INVOMEGA, GH_F_ZETA,
  GH_F_HFACTOR, GH_F_HRANGESQD, GH_F_SOUND, GH_F_PRESSURE, GH_F_DIV_V, GH_F_GPOT, GH_F_GPOT_HYDRO,
  GH_F_ALPHA, GH_F_DALPHADT, GH_F_DT, GH_F_DT_NEXT, GH_F_TLAST,                 /* scalars */
  /* block timesteps (Particle.h:137-142), integers carried as doubles: level, levelneib, nstep, nlast and the flag
   * word (bit 0 active, bit 1 end_timestep) */
  GH_F_LEVEL, GH_F_LEVELNEIB, GH_F_NSTEP, GH_F_NLAST, GH_F_FLAGS,
  /* sink runs: Particle::sinkid (-1 = not inside a sink); GH_F_FLAGS then also carries bit 2 = dead (accreted; gone after the
   * next tree build, Hydrodynamics.h:158-202) and bit 3 = potmin (GradhSph.cpp:270-280; maintained only where rho >= rho_sink,
   * which is where Sinks::SearchForNewSinkParticles reads it) */
  GH_F_SINKID,
  GH_F_COUNT
};

/* per-call statistics the roofline accounting is computed from */
typedef struct gh_stats {
  int64_t n_particles;     /* particles processed */
  int64_t n_iterations;    /* density: sum over particles of h-iterations executed */
  int64_t n_candidates;    /* density: sum over particles and iterations of candidates inside the
                              reference's (kernrange*hmax)^2 cull (GradhSphTree.cpp:205-219);
                              forces: hydro pairs evaluated */
  int64_t n_direct;        /* gravity: direct (Newtonian) particle interactions */
  int64_t n_cells;         /* gravity: cell (multipole) interactions */
  int64_t n_retries;       /* density: groups redone with a larger search radius */
  double  kernel_ms;       /* device time of the call's kernels (hipEvent) */
  /* gravity evaluation: what the kernel loads, counted once per LEAF (its 4-6 particles share every loaded entry) */
  int64_t n_leaf_cells;    /* accepted-cell list entries read (4-byte id + 32-byte centre-of-mass record each) */
  int64_t n_leaf_direct;   /* particles of direct-sum leaves read (32-byte (r, m) record each) */
  int64_t n_leaf_cand;     /* hydro-candidate particles read and classified (40 bytes each) */
} gh_stats;

/* ---- lifetime ----------------------------------------------------------------------------- */
/* replaces: new GradhSph<ndim,Kernel> + new GradhSphTree + new SphLeapfrogKDK
 * (GradhSphSimulation.cpp:131-246) */
int gh_create(const gh_config *cfg, gh_ctx **out);
void gh_destroy(gh_ctx *ctx);
const char *gh_last_error(const gh_ctx *ctx);

/* ---- particle data ------------------------------------------------------------------------ */
/* replaces: GradhSph::AllocateMemory + the IC copy loop (GradhSph.cpp:83-111, e.g. UniformIc.cpp:118-129).
 * r, v: [N][ndim]; m, h, u: [N].  v, u may be NULL (zeros). Sets r0=r, v0=v, a=0, alpha=alpha_visc,
 * iorig=i, all particles active (SphSimulation.cpp:245-256). */
int gh_upload_particles(gh_ctx *ctx, int64_t N, const double *r, const double *v, const double *m,
                        const double *h, const double *u);
/* overwrite one field (caller order) -- used to restart from / compare with reference snapshots */
int gh_upload_field(gh_ctx *ctx, int field, const double *src);
/* replaces: reading sph->GetSphParticleArray() (Sph.h:139).  dst is [N] or [N][ndim], caller order */
int gh_download(gh_ctx *ctx, int field, double *dst);
int64_t gh_num_particles(const gh_ctx *ctx);

/* ---- tree --------------------------------------------------------------------------------- */
/* replaces: HydroTree::BuildTree(rebuild=true,...) -> KDTree::BuildTree + StockTree
 * (HydroTree.cpp:310-372, KDTree.cpp:220-313, 442-595, 808-1083).  Reorders the device copy of the
 * particles into tree order. */
int gh_build_tree(gh_ctx *ctx);
/* replaces: HydroTree::BuildTree with its own arguments (HydroTree.cpp:310-372) for callers that keep the reference's MainLoop
 * (include/reference_shell/HipSphTree.h): rebuild when n % ntreebuildstep == 0 or rebuild_tree (KDTree::BuildTree), re-stock the
 * existing cells when n % ntreestockstep == 0 (KDTree::StockTree, :760-1083), otherwise let the cells drift with their stocked
 * mean velocities over `timestep` (Tree::ExtrapolateCellProperties, Tree.cpp:172-198).  Positions / h uploaded since the last
 * call are what is stocked.  (gh_step applies the same schedule itself from gh_config.) */
int gh_build_tree_scheduled(gh_ctx *ctx, int rebuild_tree, int n, int ntreebuildstep, int ntreestockstep, double timestep);
/* tree export for parity tests, in the reference's pre-order cell numbering (KDTree.cpp:362-433).
 * Any pointer may be NULL.  Ncell = 2*gtot-1.  cell_first/cell_N index the `order` array:
 * order[cell_first[c] .. +cell_N[c]) are the caller-order ids of the particles of cell c. */
/* Hybrid gas + star runs: the stars as the gas sees them (positions [nstars][ndim], masses, smoothing lengths;
 * nbody_softening as in the parameter file).  While nstars > 0
 *   - gh_update_density adds the star term of zeta (GradhSph::ComputeH, GradhSph.cpp:288-307, conservative_sph_star_gravity = 1),
 *   - gh_update_all_forces adds the stars' kernel-softened gravity to the gas (GradhSph::ComputeStarGravForces,
 *     GradhSph.cpp:699-743),
 *   - gh_star_gas_forces returns the gas' gravity on every star through the gas tree (HydroTree::UpdateAllStarGasForces,
 *     HydroTree.cpp:552-657; Tree::ComputeStarGravityInteractionList, Tree.cpp:748-885;
 *     NbodyLeapfrogKDK::CalculateDirectHydroForces, NbodyLeapfrogKDK.cpp:151-239): a [nstars][ndim], gpot [nstars],
 *     to be added to the star-star sums of gh_nbody_forces. */
int gh_set_stars(gh_ctx *ctx, int64_t nstars, const double *r, const double *m, const double *h, int nbody_softening);
int gh_star_gas_forces(gh_ctx *ctx, double *a, double *gpot);

/* Hierarchical block timesteps (Nlevels > 1).  Replaces Simulation::ComputeBlockTimesteps (Simulation.cpp:1764-2200),
 * SphLeapfrogKDK::CheckTimesteps (SphLeapfrogKDK.cpp:284-330) and the active-particle bookkeeping of
 * SphSimulation::MainLoop (SphSimulation.cpp:574-880); gh_setup / gh_step run them when cfg.Nlevels > 1.  The integer
 * clock of the reference's Simulation object (n, nresync, level_max, level_step) and dt_max can be set and read for
 * restarts: clock4 = {n, nresync, level_max, level_step}. */
int gh_set_block_clock(gh_ctx *ctx, int n, int nresync, int level_max, int level_step, double dt_max);
int gh_get_block_clock(gh_ctx *ctx, int32_t *clock4, double *dt_max);
/* number of particle force evaluations (active particles summed over the steps, the N_active of SURVEY.md 8d's metric)
 * since the last reset; block-timestep runs only */
int gh_get_active_count(gh_ctx *ctx, int64_t *nactive, int reset);

int gh_tree_size(gh_ctx *ctx, int32_t *Ncell, int32_t *ltot, int32_t *gtot);
int gh_export_tree(gh_ctx *ctx, int32_t *cell_level, int32_t *cell_first, int32_t *cell_N,
                   double *bbmin, double *bbmax, double *hboxmin, double *hboxmax, double *rcell,
                   double *com, double *mass, double *rmax, double *hmax, double *cdistsqd,
                   int32_t *order);

/* ---- the hot path ------------------------------------------------------------------------- */
/* replaces: GradhSphTree::UpdateAllSphProperties -> GradhSph::ComputeH + ComputeThermalProperties
 * + KDTree::UpdateHmaxValues (GradhSphTree.cpp:83-271, GradhSph.cpp:142-347, KDTree.cpp:1128).
 * Periodic images are made on the fly instead of materialising ghost particles
 * (replaces HydroTree::SearchBoundaryGhostParticles/BuildGhostTree, HydroTree.cpp:382-543). */
int gh_update_density(gh_ctx *ctx, gh_stats *stats);
/* replaces: Sph::ZeroAccelerations (Sph.cpp:126-140) */
int gh_zero_accelerations(gh_ctx *ctx);
/* replaces: GradhSphTree::UpdateAllSphHydroForces -> Tree::ComputeNeighbourAndGhostList,
 * NeighbourManager::EndSearch/GetParticleNeib, GradhSph::ComputeSphHydroForces
 * (GradhSphTree.cpp:280-435, Tree.cpp:562-617, NeighbourManager.h:368-543, GradhSph.cpp:361-460) */
int gh_update_hydro_forces(gh_ctx *ctx, gh_stats *stats);
/* replaces: GradhSphTree::UpdateAllSphForces -> Tree::ComputeGravityInteractionAndGhostList,
 * GradhSph::ComputeSphHydroGravForces/ComputeDirectGravForces, ComputeCellMonopoleForces
 * (GradhSphTree.cpp:444-657, Tree.cpp:628-735, GradhSph.cpp:474-690, NeighbourSearch.h:350-377) */
int gh_update_all_forces(gh_ctx *ctx, gh_stats *stats);

/* ---- leapfrog KDK glue --------------------------------------------------------------------- */
/* replaces: SphLeapfrogKDK::AdvanceParticles (SphLeapfrogKDK.cpp:76-127) + CheckBoundaries */
int gh_kdk_advance(gh_ctx *ctx, int n, double t, double timestep);
/* replaces: Simulation::ComputeGlobalTimestep + SphIntegration::Timestep
 * (Simulation.cpp:1669-1754, SphIntegration.cpp:81-134); sets dt_next of every particle */
int gh_compute_global_timestep(gh_ctx *ctx, double *dt_min);
/* replaces: SphLeapfrogKDK::EndTimestep (SphLeapfrogKDK.cpp:219-272) */
int gh_kdk_end(gh_ctx *ctx, int n, double t, double timestep);

/* ---- whole steps --------------------------------------------------------------------------- */
/* replaces: the force part of SphSimulation::PostInitialConditionsSetup (SphSimulation.cpp:204-565):
 * tree + density twice (three times if the caller's h is only InitialSmoothingLengthGuess, Sph.cpp:76-119,
 * i.e. initial_h_provided = 0), forces, first timestep, KDK end. */
int gh_setup(gh_ctx *ctx, int initial_h_provided, double *timestep);
/* set the simulation clock (reference SimulationBase::t, ::timestep) - used when a run is continued
 * from a state uploaded field by field */
int gh_set_time(gh_ctx *ctx, double t, double timestep);
/* replaces: SphSimulation::MainLoop (SphSimulation.cpp:574-880) for Nlevels=1, no stars/sinks.
 * Runs nsteps steps without returning to the host in between. */
int gh_step(gh_ctx *ctx, int nsteps, double *t, double *timestep);

/* ---- queries for parity tests -------------------------------------------------------------- */
/* replaces: NeighbourSearch::GetGatherNeighbourList (HydroTree.cpp / Tree.cpp:208-280): ids (caller
 * order) of particles with |r_j - r_i|^2 < (kernrange*h_i)^2 for every i, CSR form.
 * offsets: [N+1]; ids: [cap]; returns GH_ERR_CAPACITY (and the needed size in offsets[N]) if
 * cap is too small. */
int gh_gather_neighbours(gh_ctx *ctx, int64_t cap, int64_t *offsets, int32_t *ids);
/* replaces: NeighbourSearch::GetGatherNeighbourList(rp, rsearch, ...) -> Tree::ComputeGatherNeighbourList (NeighbourSearch.h:82,
 * Tree.cpp:208-280; the sink search uses it): caller-order ids of the particles with |r - rp|^2 < rsearch^2 around an arbitrary
 * point rp[ndim].  Periodic / mirror domains: particles also count through their images (the reference finds those in its ghost
 * tree, HydroTree.cpp:451-471; the id returned is the real particle's).  Sink runs: accreted (dead) particles are left out
 * (Tree.cpp:252).  Returns the count (>= 0); -1 = the reference's overflow answer, given whenever the complete list leaves
 * less than Nleafmax free slots of `cap` (the reference refuses a leaf cell unless Nneib + Nleafmax < Nneibmax, Tree.cpp:247);
 * GH_ERR_HIP / GH_ERR_UNSUPPORTED (bad arguments, no tree, more than one rank). */
int gh_gather_neighbours_at(gh_ctx *ctx, const double *rp, double rsearch, int32_t *list, int32_t cap);
/* device time (ms) spent in each phase since the last call to gh_reset_timers, reference block
 * names (CodeTiming: BUILD_TREE, SPH_PROPERTIES, SPH_HYDRO_FORCES / SPH_ALL_FORCES, KDK) */
enum { GH_T_BUILD_TREE = 0, GH_T_SPH_PROPERTIES, GH_T_SPH_FORCES, GH_T_KDK,
       GH_T_GRAV_WALK /* interaction-list walk of the self-gravity pass; GH_T_SPH_FORCES is then the evaluation */,
       GH_T_COUNT };
int gh_get_timers(gh_ctx *ctx, double *ms /* [GH_T_COUNT] */, gh_stats *density, gh_stats *forces);
int gh_reset_timers(gh_ctx *ctx);

/* ---- multi-GPU (one process per GPU) -----------------------------------
 * Replaces the reference's MPI layer for this path: MpiKDTreeDecomposition::CreateInitialDecomposition
 * (src/Mpi/MpiKDTreeDecomposition.cpp:56-135), MpiControl::UpdateAllBoundingBoxes / SendReceiveGhosts /
 * ExportParticlesBeforeForceLoop / GetExportedParticlesAccelerations (src/Mpi/MpiControl.cpp:329-337, 745-1150),
 * HydroTree::BuildPrunedTree / the pruned-tree exchange (src/Tree/HydroTree.cpp:1044-1230) and the MPI_Allreduce of
 * the timestep (src/Common/Simulation.cpp:1738).
 *
 * nranks = 2^L processes, one GPU each.  Rank r owns level-L cell r of the GLOBAL KD-tree: the L shared top levels
 * are split at exact global medians, so the union of the ranks' subtrees is the tree one GPU builds and every rank
 * computes for its particles what one GPU computes for them.  A rank holds its own particles plus the halo it imports
 * per phase (cells and particles of other ranks that its tree walks can reach).
 *
 * The library does not talk to a network itself: the host supplies two collectives over DEVICE buffers, to be
 * enqueued on (or ordered against) the given HIP stream - RCCL through torch.distributed in gandalf_amd/multigpu.py,
 * ncclAllGather / grouped ncclSend+ncclRecv (or MPI_Allgather / MPI_Alltoallv) in a C++ host.  Return 0 on success. */
typedef struct gh_comm_ops {
  void *user;
  /* every rank contributes `bytes` bytes from send_dev; recv_dev receives nranks*bytes, ordered by rank */
  int (*allgather)(void *user, const void *send_dev, void *recv_dev, int64_t bytes, void *hip_stream);
  /* send_bytes[r] bytes to rank r from consecutive blocks of send_dev (rank order), recv_bytes[r] from rank r into
   * consecutive blocks of recv_dev; the byte counts are HOST arrays [nranks] */
  int (*alltoallv)(void *user, const void *send_dev, const int64_t *send_bytes, void *recv_dev, const int64_t *recv_bytes,
                   void *hip_stream);
} gh_comm_ops;
/* call after gh_create and before gh_upload_particles.  nranks = 1 (ops may be NULL) is the single-GPU default.
 * Multi-rank runs: open or periodic boundaries, global or block timesteps, sinks and stars (gh_hybrid_setup / gh_hybrid_step are
 * collective calls then; the star context is the same on every rank), geometric MAC, constant-alpha viscosity, tree rebuilt
 * every step; anything else is refused here (GH_ERR_UNSUPPORTED).  gh_upload_particles then takes the WHOLE initial condition on
 * every rank and keeps this rank's share; gh_download / gh_upload_field touch only this rank's own particles.
 * gh_build_tree, gh_update_density, gh_update_*_forces, gh_setup and gh_step are collective calls from then on. */
int gh_comm_init(gh_ctx *ctx, int rank, int nranks, const gh_comm_ops *ops);
/* ---- native RCCL transport (gandalf_amd/csrc/rccl_comm.hip): a ready-made gh_comm_ops ---------------------------
 * replaces the MPI calls of the reference's hot path one for one: MPI_Allgather (src/Mpi/MpiControl.cpp:329-337) ->
 * ncclAllGather; MPI_Alltoallv (:1073-1150) and the pruned-tree MPI_Isend / MPI_Irecv (src/Tree/HydroTree.cpp:1044-1230)
 * -> one group of ncclSend / ncclRecv pairs (a direct xGMI link per pair).  RCCL is dlopen'ed at the first call.
 * One process per GPU: rank 0 calls gh_rccl_unique_id and hands the 128 bytes to the other ranks by whatever launched
 * them (MPI_Bcast gandalf.cpp-style, a torch.distributed store, a file); every rank then calls gh_rccl_create - a
 * collective, like MPI_Init + MPI_Comm_rank (src/Common/gandalf.cpp:51-103).  One process driving several GPUs with a
 * thread per GPU: gh_rccl_create_all (ncclCommInitAll).  gh_rccl_ops(c) is what gh_comm_init takes. */
typedef struct gh_rccl gh_rccl;
int gh_rccl_unique_id(void *id128);
int gh_rccl_create(gh_rccl **out, int rank, int nranks, const void *id128, int device);
int gh_rccl_create_all(gh_rccl **out /* [ndev] */, int ndev, const int *devices);
const gh_comm_ops *gh_rccl_ops(gh_rccl *c);
const char *gh_rccl_last_error(gh_rccl *c);
const char *gh_rccl_load_error(void);          /* "" when RCCL could be loaded */
/* collectives issued through this communicator since the last reset (bench.py reports them per step) */
int gh_rccl_counters(gh_rccl *c, int64_t *n_allgather, int64_t *n_alltoallv, int64_t *bytes, int reset);
void gh_rccl_destroy(gh_rccl *c);
/* halo import for the tree walks of one phase; gh_update_density / gh_update_*_forces / gh_step call it themselves */
enum { GH_HALO_DENSITY = 0 /* cell boxes + (r, m) */, GH_HALO_HYDRO = 1, GH_HALO_GRAVITY = 2 /* all cell records + force records */ };
int gh_exchange_halo(gh_ctx *ctx, int phase);
/* all-gather of the top levels of every rank's subtree (boxes, h-boxes, centres of mass, masses, quadrupoles) and
 * re-stocking of the shared levels above them; gh_build_tree / gh_update_density call it themselves */
int gh_allgather_multipoles(gh_ctx *ctx);
/* this rank's particle range in the global tree order, and the number of particles it currently holds (own + imported) */
int gh_comm_info(gh_ctx *ctx, int64_t *own_first, int64_t *own_count, int64_t *held);
/* the HIP stream (hipStream_t) every entry point enqueues on */
void *gh_stream(gh_ctx *ctx);
/* replaces: KDTree::UpdateHmaxValues (KDTree.cpp:1128-1208); gh_update_density calls it itself */
int gh_update_hmax(gh_ctx *ctx);
/* device pointer of a field's tree-ordered storage: component k of a vector field, k=0 for scalars */
void *gh_field_dev(gh_ctx *ctx, int field, int k);

/* ---- N-body (stars): direct sum + leapfrog KDK ------------------------------------------------
 * replaces: Nbody<ndim>::CalculateDirectGravForces (src/Nbody/Nbody.cpp:233-287),
 * NbodyLeapfrogKDK::CalculateDirectSmoothedGravForces (src/Nbody/NbodyLeapfrogKDK.cpp:78-142),
 * AdvanceParticles / CorrectionTerms / EndTimestep / Timestep (:253-400) and the star part of
 * Simulation::ComputeGlobalTimestep (src/Common/Simulation.cpp:1720-1745), in the order of
 * NbodySimulation::MainLoop (src/Nbody/NbodySimulation.cpp:311-404).  Global timestep (Nlevels = 1),
 * open boundaries, no sub-systems / perturbers.  Stars stay in caller order (no tree). */
typedef struct gh_nbody gh_nbody;
enum { GH_NB_R = 0, GH_NB_V, GH_NB_A, GH_NB_ADOT, GH_NB_GPOT, GH_NB_FIELDS,
       /* start-of-step copies and the time of the last step end (upload_field / download only) */
       GH_NB_R0 = GH_NB_FIELDS, GH_NB_V0, GH_NB_A0, GH_NB_TLAST, GH_NB_ALLFIELDS };
/* softening: 0 = Newtonian point masses, 1 = M4-kernel softened with mean h (nbody_softening) */
int gh_nbody_create(int ndim, int softening, double nbody_mult, int device, gh_nbody **out);
void gh_nbody_destroy(gh_nbody *nb);
const char *gh_nbody_last_error(const gh_nbody *nb);
/* r, v: [N][ndim] row-major; m, h: [N] */
int gh_nbody_upload(gh_nbody *nb, int64_t N, const double *r, const double *v, const double *m, const double *h);
/* out: [N][ndim] for vector fields, [N] for GH_NB_GPOT */
int gh_nbody_download(gh_nbody *nb, int field, double *out);
/* zero a, adot, gpot of all stars and sum all pairs (every star active) */
int gh_nbody_forces(gh_nbody *nb);
/* PostInitialConditionsSetup for stars: forces, first global timestep, EndTimestep.  Returns dt. */
int gh_nbody_setup(gh_nbody *nb, double *timestep);
/* nsteps x MainLoop.  Returns t and the next timestep. */
int gh_nbody_step(gh_nbody *nb, int nsteps, double *t, double *timestep);
/* state of a restart / of a hybrid run: overwrite one field ([N][ndim], or [N] for GH_NB_GPOT / GH_NB_TLAST) */
int gh_nbody_upload_field(gh_nbody *nb, int field, const double *src);
/* Hybrid gas + stars: nsteps of SphSimulation::MainLoop with a global timestep (SphSimulation.cpp:574-880, Npec = 1) -
 * both species advance, the gas passes see the stars (gh_set_stars is refreshed from the star context every step), the stars
 * get the gas' tree forces (gh_star_gas_forces) on top of their direct sum, the timestep is the minimum over both
 * (Simulation::ComputeGlobalTimestep, Simulation.cpp:1669-1754).  gas must hold the post-setup state (gh_set_time). */
int gh_hybrid_step(gh_ctx *gas, gh_nbody *stars, int nsteps, double *t, double *timestep);
/* sink runs (cfg.sink_particles = 1): gh_hybrid_setup / gh_hybrid_step do the whole of MainLoop including
 * Sinks::SearchForNewSinkParticles / CreateNewSinkParticle / AccreteMassToSinks (Sinks.cpp:118-777) and the removal of accreted
 * particles before every tree build (Hydrodynamics::DoDeleteDeadParticles, Hydrodynamics.h:158-202): gh_num_particles and
 * gh_nbody_num_stars change between calls, and gh_download returns the arrays in the reference's compacted particle order.
 * gh_get_sinks: the SinkParticle records (Sinks.h:48-100).  rec [n][GH_SINK_NREC] = radius, mmax, menc, dmdt, ketot, gpetot,
 * rotketot, utot, taccrete, trad, trot, tvisc, angmom[3], 1/h of the parent gas particle, Hydrodynamics::mmean;
 * irec [n][2] = star number, Ngas.  Either array may be NULL. */
#define GH_SINK_NREC 17
int gh_get_sinks(gh_ctx *ctx, int *nsinks, double *rec, int *irec);
int64_t gh_nbody_num_stars(const gh_nbody *nb);
/* which: 0 m, 1 h, 2 dt_internal (NbodyParticle.h) */
int gh_nbody_download_scalar(gh_nbody *nb, int which, double *out);
/* ... and its PostInitialConditionsSetup (SphSimulation.cpp:204-565): gas after gh_upload_particles, stars after
 * gh_nbody_upload; leaves both contexts in the post-setup state and returns the first timestep */
int gh_hybrid_setup(gh_ctx *gas, gh_nbody *stars, int initial_h_provided, double *timestep);


#ifdef __cplusplus
}
#endif
#endif /* GANDALF_HIP_H */
