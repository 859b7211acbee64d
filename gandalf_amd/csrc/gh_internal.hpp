// gh_internal.hpp -- context, device data layout and small helpers of libgandalf_hip.
//
// Data layout in HBM:
//   * particles: structure of arrays, one double array per scalar / vector component, stored in
//     TREE ORDER (leaf cells contiguous), double-buffered so that a tree rebuild is one gather pass;
//   * derived per-pass packs (posm = x,y,z,m as 32-byte records) for the LDS tiles;
//   * KD-tree cells in heap order (children of n are 2n+1, 2n+2), four records per cell split by consumer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include "../../include/gandalf_hip.h"

#define GH_WAVE 64

// component arrays: 7 vectors x 3 + 21 scalars
enum FieldD {
  D_RX = 0, D_RY, D_RZ, D_VX, D_VY, D_VZ, D_AX, D_AY, D_AZ, D_ATX, D_ATY, D_ATZ,
  D_R0X, D_R0Y, D_R0Z, D_V0X, D_V0Y, D_V0Z, D_A0X, D_A0Y, D_A0Z,
  D_M, D_H, D_U, D_U0, D_DUDT, D_DUDT0, D_RHO, D_INVOMEGA, D_ZETA, D_HFACTOR, D_HRANGESQD,
  D_SOUND, D_PRESSURE, D_DIV_V, D_GPOT, D_GPOT_HYDRO, D_ALPHA, D_DALPHADT, D_DT, D_DT_NEXT,
  D_TLAST,
  D_SINKID,        // sink runs: Particle::sinkid as a double (-1 = none); sink runs permute every field (flags: dead, potmin)
  // block timesteps (Nlevels > 1): integers carried as doubles so that they travel with the particle through the
  // tree-order permutation; D_FLAGS bit 0 = active, bit 1 = end_timestep.  Untouched (and not permuted) for Nlevels = 1.
  D_LEVEL, D_LEVELNEIB, D_NSTEP, D_NLAST, D_FLAGS, D_COUNT
};
#define D_COUNT_BASE D_SINKID     /* fields of a global-timestep run without sinks */
#define GH_FLAG_ACTIVE 1
#define GH_FLAG_END 2
#define GH_FLAG_DEAD 4
#define GH_FLAG_POTMIN 8

// KD-tree cell records (heap order), split by consumer so that every walk touches one 64-byte line per
// node.  Field meanings follow TreeCellBase (reference TreeCell.h:16-49).
struct alignas(64) CellBox {       // density / gather walks
  double bbmin[3], bbmax[3];       // tight bounding box of particle positions
  int first, N;                    // particle range in tree order
  double pad;
};
struct alignas(64) CellH {         // hydro walk (with CellBox); the only record UpdateHmaxValues rewrites
  double hbmin[3], hbmax[3];       // bounding box of r -/+ kernrange*h
  double hmax;                     // max h in cell
  double pad;
};
struct alignas(64) CellGeo {       // gravity walk
  double rcell[3];                 // centre of bb
  double rmax;                     // half diagonal of bb
  double cdistsqd;                 // max(rmax^2, hmax^2)/thetamaxsqd at stock time
  double hmax;                     // copy of CellH::hmax
  int first, N;
  double mac;                      // eigenvalue MAC length^2 (gravity_mac = eigenmac; KDTree.cpp:1054-1076), else 0
};
struct alignas(32) CellCom {       // accepted cells only
  double com[3];                   // centre of mass
  double m;                        // mass
};

struct alignas(64) CellQuad {      // traceless quadrupole about com (multipole = quadrupole only)
  double q[5];                     // xx, xy, yy, xz, yz  (zz = -xx-yy), KDTree.cpp:929-944
  double pad[3];
};

struct DevicePtrs {                // everything a kernel needs, passed by value
  double *f[D_COUNT];              // current particle arrays (tree order)
  int *iorig;                      // caller-order id of each particle
  double4 *posm;                   // (x,y,z,m) pack
  double4 *hrec;                   // 4 x double4 per particle: the force tiles' neighbour record
  CellBox *cbox;
  CellH *ch;
  CellGeo *cgeo;
  CellCom *ccom;
  CellQuad *cquad;                 // nullptr unless multipole = quadrupole
  double *leaf_amin;               // [gtot] per-leaf factor of the relative MACs at stock time: min |atree| (gadget2) or
                                   // max gpot^(-2/3) (eigenmac); nullptr for the geometric MAC
  double macerror; int mac_stock;  // what the stocking kernels need of the MAC
  const int *cfirst, *cN;          // static per-cell particle ranges (heap order)
  int N, ndim, ltot, gtot, lgroup, ngroups, leafocc;
  int levels;                      // Nlevels > 1: targets are the ACTIVE particles only (flags bit 0)
  int sinks;                       // cfg.sink_particles: flags bit 2 = dead, D_SINKID valid
  double *pm_invhsqd, *pm_cullsqd; // create_sinks: 1/h^2 of the last h iteration and the (kernrange*hmax)^2 of the successful ComputeH call
                                   // per particle - what the potential-minimum test of GradhSph.cpp:270-280 runs on (else nullptr)
  const int *leafact;              // extrapolated tree + block timesteps: active-particle count of every leaf AS OF THE LAST STOCKING
                                   // (the reference does not refresh cell.Nactive on extrapolation steps, SphSimulation.cpp:663) or nullptr
};

// host mirror of the stars of a sink run ([n][3] vectors, [n] scalars): what Sinks.cpp reads and writes of StarParticle
struct gh_host_stars {
  size_t n = 0;
  std::vector<double> r, v, a, adot, r0, v0, a0, m, h, gpot, tlast, dti;
  std::vector<int> level, nstep, nlast, endflag;      // block-timestep ladder (global timestep: 0, 1, 0)
  void resize(size_t k) {
    n = k;
    level.resize(k, 0); nstep.resize(k, 1); nlast.resize(k, 0); endflag.resize(k, 0);
    std::vector<double> *v3[] = {&r, &v, &a, &adot, &r0, &v0, &a0};
    for (auto *q : v3) q->resize(3*k, 0.0);
    std::vector<double> *v1[] = {&m, &h, &gpot, &tlast, &dti};
    for (auto *q : v1) q->resize(k, 0.0);
  }
  void append() { resize(n + 1); }
};
// SinkParticle (Sinks.h:48-100); the star is gh_nbody star number istar; invh = 1/h of the gas particle the sink was made from
#ifndef GH_SINK_NREC
#define GH_SINK_NREC 17
#endif
struct gh_sink_rec {
  int istar = -1, Ngas = 0;
  double radius = 0, mmax = 0, menc = 0, dmdt = 0, ketot = 0, gpetot = 0, rotketot = 0, utot = 0, taccrete = 0, trad = 0, trot = 0, tvisc = 0;
  double angmom[3] = {0, 0, 0};
  double invh = 0;
};

struct gh_ctx {
  gh_config cfg;
  int ndim = 3;
  int64_t N = 0, Ncap = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // particle storage, double buffered
  double *fbuf[2][D_COUNT] = {};
  int *iorig[2] = {};
  int cur = 0;
  double4 *posm = nullptr;
  double4 *hrec = nullptr;
  double **d_ptrtab = nullptr;     // device table of 2*D_COUNT pointers for the permute kernel

  // tree
  int ltot = 0, gtot = 0, Ncell = 0, lgroup = 0, ngroups = 0, leafocc = 0;
  int64_t tree_layout_N = -1;
  std::vector<int> h_cfirst, h_cN, h_cleft;
  std::unordered_map<void*, size_t> tree_bytes;   // capacity of every buffer gh_alloc_tree owns, keyed by the address of its pointer
  int *cfirst = nullptr, *cN = nullptr, *cleft = nullptr;
  CellBox *cbox = nullptr;
  CellH *ch = nullptr;
  CellGeo *cgeo = nullptr;
  CellCom *ccom = nullptr;
  CellQuad *cquad = nullptr;
  double *pm_invhsqd = nullptr, *pm_cullsqd = nullptr;   // create_sinks only (DevicePtrs)
  std::vector<gh_sink_rec> sinks;  // sink runs (sinks.hip)
  double mmean = 0.0;              // Hydrodynamics::mmean
  void *sink_scratch = nullptr;
  double *cvel = nullptr;          // [Ncell][3] mass-weighted mean velocity at stock time (ntreestockstep > 1 only)
  int *qs_ids = nullptr; double *qs_keys = nullptr;   // exact (quick-select order) build, tree.hip
  double *qw_k[2] = {nullptr, nullptr}; int *qw_i[2] = {nullptr, nullptr}, *qw_rk = nullptr, *qw_gp = nullptr, *qw_blk = nullptr; void *qw_st = nullptr; size_t qw_words = 0;   // ... its device-wide passes (top levels)
  bool exact_armed = false;        // a build split equal coordinates: every later build runs the gated exact kernels
  bool sink_exact_sticky = false;  // ... decided for good
  bool sink_exact = true;          // sink runs: builds keep the reference's quick-select order (needed once a particle is dense enough to be a sink candidate, or a sink exists; see gh_tree_build_impl)
  int *leafact = nullptr;          // [gtot] active particles per leaf at the last stocking (ntreestockstep > 1 and Nlevels > 1 only)
  double *leaf_amin = nullptr;
  bool mac_bootstrap = false;      // gh_setup's first force pass of a relative MAC runs geometric (SphSimulation.cpp:381-388)
  double *ktab = nullptr;          // tabulated kernel tables [GH_TAB_COUNT][GH_TAB_RES] (device), or nullptr
  double *dbbmin = nullptr, *dbbmax = nullptr;   // divide-time boxes [Ncell][3]
  int *kdiv = nullptr;
  int *P[2][3] = {};               // presorted permutations, double buffered
  int *cellnode[2] = {};
  unsigned char *side = nullptr;
  unsigned long long *W[3] = {};   // ballot words
  unsigned int *Wpre[3] = {};      // exclusive prefix of popcounts
  double *sortkeys = nullptr, *sortkeys_out = nullptr;
  int *sortvals = nullptr;
  void *sorttemp = nullptr; size_t sorttemp_bytes = 0;   // 3 slabs (one per axis: the argsorts run on 3 streams)
  hipStream_t aux[2] = {}; hipEvent_t ev_fork = nullptr, ev_join[2] = {};
  int iota_N = -1;
  int lsub = 0;                    // first level built by the LDS-resident subtree kernel
  double *redbuf = nullptr;        // reduction scratch
  bool tree_valid = false;
  bool tree_stale = false;         // cells extrapolated since the last stocking (Tree::ExtrapolateCellProperties): per-leaf searches
  bool rebuild_tree = true;        // SimulationBase::rebuild_tree: raised by upload / setup, lowered at the end of a step

  // gravity interaction lists in HBM (gravity.hip)
  int *gl_cells = nullptr, *gl_dirl = nullptr, *gl_hydl = nullptr, *gl_len = nullptr, *gl_gcells = nullptr, *gl_glen = nullptr;
  size_t glist_leaves = 0;
  int glist_caps = 0;
  int glist_mul[3] = {1, 1, 1}, glist_cap[3] = {0, 0, 0}, glist_max[3] = {0, 0, 0}, glist_checked = -1;   // headroom tracking (gh_grav_list_headroom)

  // candidate range lists of the split density path (density.hip)
  void *dl_rl = nullptr; int *dl_rlen = nullptr; int dl_groups = 0;

  // statistics / timers
  unsigned long long *d_stats = nullptr;   // device counters
  int *d_flags = nullptr;                  // device error flags
  struct EvPair { hipEvent_t a, b; };
  std::vector<EvPair> ev_used[GH_T_COUNT];    // recorded, not yet read back
  std::vector<EvPair> ev_free;
  double timers[GH_T_COUNT] = {};
  double dom_ms[GH_T_COUNT] = {};             // time of the dominant kernel of each phase
  long dom_calls[GH_T_COUNT] = {};
  gh_stats st_density = {}, st_forces = {};

  // time integration state (reference SimulationBase: n, Nsteps, t, timestep)
  int n = 0, Nsteps = 0;
  double t = 0.0, timestep = 0.0;
  // block-timestep clock (Simulation.h: nresync, level_max, level_step, dt_max); the device copy is authoritative
  int nresync = 0, level_max = 0, level_step = 0;
  double dt_max = 0.0;
  int *d_blk = nullptr;            // device {n, nresync, level_max, level_step, level_max_new, activecount, nfactor_mul, nfactor_div}

  // stars of a hybrid gas + N-body run as the gas sees them (stars.hip)
  double4 *star_posm = nullptr; double *star_h = nullptr, *star_out = nullptr;
  int nstars = 0, star_softening = 1; int64_t star_cap = 0;

  // multi-GPU (comm.hip): rank r of nranks = 2^L owns level-L cell r of the global KD-tree - the static particle range
  // [own_first, own_first + own_count) of the global tree-order index space; everything when nranks == 1
  int rank = 0, nranks = 1, L = 0;
  int64_t own_first = 0, own_count = 0;
  int64_t own_held = -1;           // sink runs on several ranks: particles in the own range after dead ones left, until the next migration evens the ranks out (-1: own_count)
  int iota_p0 = -1;
  bool in_step = false;            // inside gh_step's global-timestep loop: the tree build may skip arrays the step rewrites anyway
  bool tree_valid_once = false;    // gh_build_tree_scheduled: a tree has been built for the current particle set
  struct gh_dd *dd = nullptr;
  // gh_step: the check "did any rank's density walk leave its imported halo" is deferred to the count exchange of the
  // force-phase halo (one collective and one host synchronisation less per step)
  bool dd_defer_miss = false, dd_miss_pending = false;
};

#define GH_MAX_RANKS 16

// kernel launch dispatch on (ndim, smoothing kernel): L(ND, KT) is the launch macro of the call site
#ifdef GH_PROBE_3D_M4
// development builds (scripts/probe/mkvariant.sh): only the 3-D M4 instantiations, for quick A/B timing of one kernel
#define GH_DISPATCH(ctx, L) { L(3, 0) }
#else
#define GH_DISPATCH(ctx, L)                                                                                  \
  if ((ctx)->cfg.kernel == GH_KERNEL_QUINTIC) {                                                               \
    if ((ctx)->ndim == 1) { L(1, 1) } else if ((ctx)->ndim == 2) { L(2, 1) } else { L(3, 1) }                 \
  }                                                                                                           \
  else if ((ctx)->cfg.kernel == GH_KERNEL_M4_TAB) {                                                           \
    if ((ctx)->ndim == 1) { L(1, 2) } else if ((ctx)->ndim == 2) { L(2, 2) } else { L(3, 2) }                 \
  }                                                                                                           \
  else if ((ctx)->cfg.kernel == GH_KERNEL_QUINTIC_TAB) {                                                      \
    if ((ctx)->ndim == 1) { L(1, 3) } else if ((ctx)->ndim == 2) { L(2, 3) } else { L(3, 3) }                 \
  }                                                                                                           \
  else {                                                                                                      \
    if ((ctx)->ndim == 1) { L(1, 0) } else if ((ctx)->ndim == 2) { L(2, 0) } else { L(3, 0) }                 \
  }
#endif

#define GH_CHECK(ctx, call)                                                                   \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                        \
      return GH_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

static inline int gh_fail(gh_ctx *ctx, int code, const std::string &msg)
{
  ctx->err = msg;
  return code;
}

static inline int cdiv(int64_t a, int64_t b) { return (int) ((a + b - 1)/b); }

// error-flag bits written by kernels
enum { FLAG_FRONTIER_OVERFLOW = 1, FLAG_LEAFLIST_OVERFLOW = 2, FLAG_H_NOT_CONVERGED = 4,
       FLAG_ILIST_OVERFLOW = 8,
       FLAG_LET_MISS = 16,     // multi-GPU: a walk reached a remote cell that the halo exchange did not import
       FLAG_DD_SPLIT = 32 };   // multi-GPU: a top-level median split could not be resolved (too many equal coordinates)

// stats slots
enum { ST_ITER = 0, ST_CAND, ST_RETRY, ST_PAIRS, ST_DIRECT, ST_CELLS, ST_COUNT };
#define ST_TOTAL (ST_COUNT + 8)   /* + diagnostic cycle stamps (GH_STAMPS builds) */

DevicePtrs gh_dev(gh_ctx *ctx);
DevicePtrs gh_dev_own(gh_ctx *ctx);   // this rank's own particles only (elementwise kernels)
int gh_alloc_particles(gh_ctx *ctx, int64_t N);
int gh_alloc_tree(gh_ctx *ctx);
int gh_tree_build_impl(gh_ctx *ctx);
int gh_tree_build_checked(gh_ctx *ctx);   // ... and, at a host synchronisation point, redone in exact mode if it met its first tie
int gh_tree_extrapolate_impl(gh_ctx *ctx);  // Tree::ExtrapolateCellProperties: cells drift with their stocked mean velocity
int gh_leaf_active_counters(gh_ctx *ctx);   // KDTree::UpdateActiveParticleCounters (KDTree.cpp:1217-1254)
int gh_tree_restock_impl(gh_ctx *ctx);   // KDTree::StockTree: same cells and particle order, properties from the current r, h
int gh_update_hmax_impl(gh_ctx *ctx);
// the *_impl functions only enqueue work on ctx->stream (no host synchronisation)
int gh_density_impl(gh_ctx *ctx, bool count, bool redo_only = false);   // redo_only: just the groups a deferred miss check found
int gh_hydro_forces_impl(gh_ctx *ctx, bool count);
int gh_all_forces_impl(gh_ctx *ctx, bool count);
int gh_grav_list_headroom(gh_ctx *ctx);   // longest interaction lists -> capacities of the next pass (called where the host synchronises anyway)
int gh_grav_lists_impl(gh_ctx *ctx, bool count);   // two-kernel gravity with interaction lists in HBM
int gh_zero_acc_impl(gh_ctx *ctx);
int gh_kdk_advance_impl(gh_ctx *ctx, int n, double t, double timestep);
int gh_kdk_end_impl(gh_ctx *ctx, int n, double t, double timestep);
int gh_timestep_impl(gh_ctx *ctx);          // leaves min dt in ctx->redbuf[0] and writes dt_next
int gh_pack_posm(gh_ctx *ctx);
// hybrid runs (stars.hip): star term of zeta after a density pass, gas <- stars after a gravity pass
int gh_zeta_stars_impl(gh_ctx *ctx);
int gh_gas_star_forces_impl(gh_ctx *ctx);
int gh_cullen_dehnen_impl(gh_ctx *ctx);               // cd2010.hip: alpha, dalphadt after a density pass
// block timesteps (integrate.hip)
int gh_thermal_all_impl(gh_ctx *ctx);
int gh_block_timesteps_impl(gh_ctx *ctx);
int gh_check_timesteps_impl(gh_ctx *ctx);
int gh_block_gas_passes(gh_ctx *ctx);       // api.hip: tree + density / force passes of a block-timestep step
int gh_block_begin_step(gh_ctx *ctx);       // api.hip: n++, t += timestep, drift
int gh_block_pull(gh_ctx *ctx);             // api.hip: device block clock -> ctx->n, nresync, level_max, level_step, dt_max
int gh_block_timesteps_hybrid(gh_ctx *ctx, gh_host_stars &S, double nbody_mult);   // ComputeBlockTimesteps with the stars of a sink run
// phase timing with HIP events on ctx->stream; read back by gh_sync_collect
int gh_phase_begin(gh_ctx *ctx, int phase);
int gh_phase_end(gh_ctx *ctx, int phase);
// synchronise the stream, read event times / error flags / counters
int gh_sync_collect(gh_ctx *ctx, const char *where);
struct Domain; struct EosParams;
void gh_fill_domain(const gh_ctx *ctx, Domain &dom);
void gh_fill_eos(const gh_ctx *ctx, EosParams &e);
void gh_shard_groups(const gh_ctx *ctx, int rank, int &g0, int &g1);
// multi-GPU (comm.hip); all no-ops on one rank
int gh_dd_exchange(gh_ctx *ctx, int phase);    // halo / locally-essential-tree import for the walks of `phase`
int gh_dd_exchange_margin(gh_ctx *ctx, int phase, double widen);   // ... with the density search radius widened
#define GH_DD_REDO_DENSITY 1000                 /* gh_dd_exchange: a deferred density miss was found - not an error, see gh_force_halo */
int gh_force_halo(gh_ctx *ctx, int phase);     // forces.hip: force records of the own particles, then the halo of `phase` (redoes a deferred density miss)
int gh_dd_any(gh_ctx *ctx, const unsigned int *count_dev, int *any);   // any rank's counter non-zero? (collective, synchronises)
int gh_dd_min_dt(gh_ctx *ctx);                 // time[1] = min over ranks
int gh_dd_reduce_int(gh_ctx *ctx, int *word_dev, int op);   // one device word: 0 = max, 1 = sum over ranks (block clock)
int gh_dd_gatherv(gh_ctx *ctx, const void *src, size_t bytes, std::vector<char> &out, std::vector<size_t> &sizes);   // ragged all-gather to the hosts (sink runs)
int gh_dd_return_levelneib(gh_ctx *ctx);       // block timesteps: levelneib raised on imported copies -> max at their owners
void gh_dd_free(gh_ctx *ctx);
// sinks.hip
int gh_sinks_potmin(gh_ctx *ctx);           // potential-minimum flag of the particles with rho >= rho_sink (after a density pass)
int gh_sinks_delete_dead(gh_ctx *ctx);      // Hydrodynamics::DoDeleteDeadParticles before a tree build
int gh_sinks_step(gh_ctx *ctx, gh_host_stars &S, double t, double timestep);   // search + create + accrete (SphSimulation.cpp:820-838)
void gh_sinks_free(gh_ctx *ctx);
