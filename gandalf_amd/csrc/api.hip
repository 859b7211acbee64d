// api.hip -- the extern "C" surface of libgandalf_hip.so (include/gandalf_hip.h) and the step driver.
#include "gh_internal.hpp"
#include "walk.hpp"
#include "sph_kernels.hpp"
#include <algorithm>
#include <cmath>

int gh_advance_time_impl(gh_ctx *ctx);

struct CtxExtra { double *d_time; };
static double *g_dummy = nullptr;

// ctx->redbuf layout: [0 .. 256*6) reduction partials, then 2 doubles of {t, timestep}
double *gh_time_dev(gh_ctx *ctx) { return ctx->redbuf + 256*6; }

static int field_comp(int field, int *first, int *ncomp, int ndim)
{
  switch (field) {
    case GH_F_R: *first = D_RX; *ncomp = ndim; return 0;
    case GH_F_V: *first = D_VX; *ncomp = ndim; return 0;
    case GH_F_A: *first = D_AX; *ncomp = ndim; return 0;
    case GH_F_ATREE: *first = D_ATX; *ncomp = ndim; return 0;
    case GH_F_R0: *first = D_R0X; *ncomp = ndim; return 0;
    case GH_F_V0: *first = D_V0X; *ncomp = ndim; return 0;
    case GH_F_A0: *first = D_A0X; *ncomp = ndim; return 0;
    default: break;
  }
  if (field >= GH_F_M && field <= GH_F_TLAST) { *first = D_M + (field - GH_F_M); *ncomp = 1; return 0; }
  if (field >= GH_F_LEVEL && field <= GH_F_FLAGS) { *first = D_LEVEL + (field - GH_F_LEVEL); *ncomp = 1; return 0; }
  if (field == GH_F_SINKID) { *first = D_SINKID; *ncomp = 1; return 0; }
  return -1;
}

// ------------------------------------------------------------------------------------------------
// phase timing
// ------------------------------------------------------------------------------------------------
int gh_phase_begin(gh_ctx *ctx, int phase)
{
  gh_ctx::EvPair p;
  if (!ctx->ev_free.empty()) { p = ctx->ev_free.back(); ctx->ev_free.pop_back(); }
  else { GH_CHECK(ctx, hipEventCreate(&p.a)); GH_CHECK(ctx, hipEventCreate(&p.b)); }
  GH_CHECK(ctx, hipEventRecord(p.a, ctx->stream));
  ctx->ev_used[phase].push_back(p);
  return GH_OK;
}

int gh_phase_end(gh_ctx *ctx, int phase)
{
  GH_CHECK(ctx, hipEventRecord(ctx->ev_used[phase].back().b, ctx->stream));
  return GH_OK;
}

int gh_sync_collect(gh_ctx *ctx, const char *where)
{
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int ph = 0; ph < GH_T_COUNT; ph++) {
    for (auto &p : ctx->ev_used[ph]) {
      float ms = 0.f;
      GH_CHECK(ctx, hipEventElapsedTime(&ms, p.a, p.b));
      ctx->timers[ph] += ms;
      ctx->dom_ms[ph] += ms;
      ctx->dom_calls[ph]++;
      ctx->ev_free.push_back(p);
    }
    ctx->ev_used[ph].clear();
  }
  int flags = 0;
  GH_CHECK(ctx, hipMemcpy(&flags, ctx->d_flags, sizeof(int), hipMemcpyDeviceToHost));
  if (!ctx->exact_armed && ctx->nranks == 1) {
    // a tree build split equal coordinates at a median: from now on every build also enqueues the (gated) exact kernels
    int tie[2] = {0, 0};
    GH_CHECK(ctx, hipMemcpy(tie, ctx->d_blk + 13, sizeof(tie), hipMemcpyDeviceToHost));
    if (tie[0] | tie[1]) ctx->exact_armed = true;
  }
  if (ctx->nranks > 1 && ctx->tree_valid) {
    // equal coordinates at a median split (lattice initial conditions): on one rank the tree is then rebuilt with the
    // reference's own quick-select order; across ranks the shared levels order ties by particle id, which is not the
    // reference's tree.  Every rank has the flag of every other (it travels with the subtree tops): all stop here.
    int tie[2] = {0, 0};
    GH_CHECK(ctx, hipMemcpy(tie, ctx->d_blk + 13, sizeof(tie), hipMemcpyDeviceToHost));
    if (tie[0] | tie[1])
      return gh_fail(ctx, GH_ERR_UNSUPPORTED, std::string(where) + ": multi-GPU: a median split separates particles with equal coordinates (lattice "
                     "initial conditions?) - the reference's quick-select tie order is reproduced on one rank only");
  }
  if (ctx->cfg.sink_particles && ctx->nranks == 1 && !ctx->sink_exact_sticky) {
    // sink runs: does any tree build have to keep the reference's order inside the leaves yet?  (see gh_tree_build_impl;
    // -1 = no density pass has counted yet; once needed, always needed)
    int dense = -1;
    GH_CHECK(ctx, hipMemcpy(&dense, ctx->d_blk + 19, sizeof(int), hipMemcpyDeviceToHost));
    const bool need = ctx->cfg.create_sinks != 1 || dense != 0 || !ctx->sinks.empty() || ctx->nstars > 0 || getenv("GH_SINK_EXACT_ALWAYS");
    if (need && dense >= 0) ctx->sink_exact_sticky = true;
    ctx->sink_exact = need;
  }
  if (ctx->cfg.self_gravity && !flags) { const int rc = gh_grav_list_headroom(ctx); if (rc) return rc; }
  if (flags) {
    GH_CHECK(ctx, hipMemset(ctx->d_flags, 0, sizeof(int)));
    std::string m = std::string(where) + ":";
    if (flags & FLAG_FRONTIER_OVERFLOW) m += " tree-walk stack overflow (GH_SCAP)";
    if (flags & FLAG_LEAFLIST_OVERFLOW) m += " candidate list overflow";
    if (flags & FLAG_ILIST_OVERFLOW) m += " interaction list overflow";
    if (flags & FLAG_H_NOT_CONVERGED) m += " h-rho iteration did not converge (GradhSph.cpp:249)";
    if (flags & FLAG_LET_MISS) m += " multi-GPU: a tree walk left the imported halo (smoothing lengths grew past the exchange margin)";
    if (flags & FLAG_DD_SPLIT) m += " multi-GPU: top-level median split unresolved (more equal coordinates than the candidate buffer holds)";
    ctx->err = m;
    return (flags & FLAG_H_NOT_CONVERGED) && !(flags & ~FLAG_H_NOT_CONVERGED) ? GH_ERR_NOTCONVERGED : GH_ERR_CAPACITY;
  }
  return GH_OK;
}

static int read_stats(gh_ctx *ctx, gh_stats *st, int phase)
{
  unsigned long long hs[ST_TOTAL];
  GH_CHECK(ctx, hipMemcpy(hs, ctx->d_stats, sizeof(hs), hipMemcpyDeviceToHost));
  GH_CHECK(ctx, hipMemset(ctx->d_stats, 0, sizeof(hs)));
#ifdef GH_STAMPS
  fprintf(stderr, "[stamps] phase %d:", phase);
  for (int k = 0; k < 8; k++) fprintf(stderr, " %.4g", (double) hs[ST_COUNT + k]);
  fprintf(stderr, "\n");
#endif
  gh_stats loc;
  memset(&loc, 0, sizeof(loc));
  loc.n_particles = ctx->N;
  loc.n_iterations = phase == GH_T_SPH_PROPERTIES ? (int64_t) hs[ST_ITER] : 0;
  loc.n_candidates = phase == GH_T_SPH_PROPERTIES ? (int64_t) hs[ST_CAND] : (int64_t) hs[ST_PAIRS];
  loc.n_retries = phase == GH_T_SPH_PROPERTIES ? (int64_t) hs[ST_RETRY] : 0;
  if (phase != GH_T_SPH_PROPERTIES) { loc.n_leaf_cells = (int64_t) hs[ST_ITER]; loc.n_leaf_direct = (int64_t) hs[ST_RETRY]; loc.n_leaf_cand = (int64_t) hs[ST_CAND]; }
  loc.n_direct = phase == GH_T_SPH_PROPERTIES ? (int64_t) hs[ST_PAIRS] : (int64_t) hs[ST_DIRECT];   // density: candidate slots tested
  loc.n_cells = (int64_t) hs[ST_CELLS];
  loc.kernel_ms = ctx->dom_calls[phase] ? ctx->dom_ms[phase]/ctx->dom_calls[phase] : 0.0;
  if (st) *st = loc;
  if (phase == GH_T_SPH_PROPERTIES) ctx->st_density = loc; else ctx->st_forces = loc;
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------

// ---- tabulated kernel: host-side construction of the tables (TabulatedKernel.cpp:57-100, SmoothingKernel.h:581-597)
#include "host_kernels.hpp"
namespace {
template <class HK> static void fill_tables(const HK &k, double R, std::vector<double> &t)
{
  const int res = GH_TAB_RES;
  const double R2 = R*R, step = R/res, stepsq = R2/res;
  for (int i = 0; i < res; i++) {
    t[GH_TAB_W1*res + i] = k.w1(step*i);
    t[GH_TAB_WGRAV*res + i] = k.wgrav(step*i);
    t[GH_TAB_WPOT*res + i] = k.wpot(step*i);
    t[GH_TAB_W0S2*res + i] = k.w0(std::sqrt(stepsq*i));
    t[GH_TAB_WOMEGAS2*res + i] = k.womega(std::sqrt(stepsq*i));
    t[GH_TAB_WZETAS2*res + i] = k.wzeta(std::sqrt(stepsq*i));
  }
}
}

static int gh_build_kernel_tables(gh_ctx *ctx)
{
  std::vector<double> t((size_t) GH_TAB_COUNT*GH_TAB_RES);
  if (ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) fill_tables(HostQuintic(ctx->ndim), 3.0, t);
  else fill_tables(HostM4(ctx->ndim), 2.0, t);
  GH_CHECK(ctx, hipMalloc((void**) &ctx->ktab, sizeof(double)*t.size()));
  GH_CHECK(ctx, hipMemcpy(ctx->ktab, t.data(), sizeof(double)*t.size(), hipMemcpyHostToDevice));
  return GH_OK;
}

extern "C" int gh_create(const gh_config *cfg, gh_ctx **out)
{
  if (!cfg || !out) return GH_ERR_INVALID;
  *out = nullptr;
  if (cfg->ndim < 1 || cfg->ndim > 3) return GH_ERR_INVALID;
  gh_ctx *ctx = new gh_ctx();
  ctx->cfg = *cfg;
  ctx->ndim = cfg->ndim;
  *out = ctx;
  if (cfg->kernel < GH_KERNEL_M4 || cfg->kernel > GH_KERNEL_QUINTIC_TAB)
    return gh_fail(ctx, GH_ERR_UNSUPPORTED, "kernels built: m4, quintic, each with tabulated_kernel = 0 or 1");
  if (cfg->Nleafmax < 1 || cfg->Nleafmax > 32) return gh_fail(ctx, GH_ERR_INVALID, "Nleafmax out of range");
  if (cfg->sink_particles && (cfg->ntreebuildstep > 1 || !cfg->self_gravity || cfg->ndim != 3 || (cfg->Nlevels > 1 && cfg->sph_single_timestep)))
    return gh_fail(ctx, GH_ERR_UNSUPPORTED, "sink runs: 3-D, self_gravity = 1, tree rebuilt every step (ntreebuildstep = 1), no sph_single_timestep");
  if (cfg->sink_particles && !(cfg->rho_sink > 0.0)) return gh_fail(ctx, GH_ERR_INVALID, "sink_particles = 1 needs rho_sink > 0");
  // extrapolated trees (ntreestockstep > 1) are searched the reference's way - per leaf cell against the drifted boxes,
  // losing the neighbours the reference loses (walk_dfs_stream_masked) - for open boundaries and the density / force
  // walks only; what is not restated is refused here rather than computed differently from the reference
  if (cfg->ntreestockstep > 1) {
    bool images = false;
    for (int k = 0; k < cfg->ndim; k++) images = images || cfg->boundary_lhs[k] != GH_BOUNDARY_OPEN || cfg->boundary_rhs[k] != GH_BOUNDARY_OPEN;
    if (images) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "ntreestockstep > 1 (extrapolated trees): open boundaries only - the reference's search of a drifted tree through periodic / mirror images is not restated");
    if (cfg->avisc == GH_AVISC_MON97CD2010) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "ntreestockstep > 1 (extrapolated trees): not with time_dependent_avisc = cd2010 - its gather pass is not restated for a drifted tree");
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return gh_fail(ctx, GH_ERR_HIP, "no HIP device: libgandalf_hip has no CPU path");
  GH_CHECK(ctx, hipSetDevice(cfg->device));
  GH_CHECK(ctx, hipStreamCreate(&ctx->stream));
  for (int i = 0; i < 2; i++) {
    GH_CHECK(ctx, hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking));
    GH_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming));
  }
  GH_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->redbuf, sizeof(double)*(256*6 + 8)));
  GH_CHECK(ctx, hipMemset(ctx->redbuf, 0, sizeof(double)*(256*6 + 8)));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->d_stats, sizeof(unsigned long long)*ST_TOTAL));
  GH_CHECK(ctx, hipMemset(ctx->d_stats, 0, sizeof(unsigned long long)*ST_TOTAL));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->d_flags, sizeof(int)));
  GH_CHECK(ctx, hipMemset(ctx->d_flags, 0, sizeof(int)));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->d_ptrtab, sizeof(double*)*4*D_COUNT));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->d_blk, sizeof(int)*24));
  GH_CHECK(ctx, hipMemset(ctx->d_blk, 0, sizeof(int)*24));
  if (cfg->kernel == GH_KERNEL_M4_TAB || cfg->kernel == GH_KERNEL_QUINTIC_TAB) { const int rc = gh_build_kernel_tables(ctx); if (rc) return rc; }
  return GH_OK;
}

static void free_particles(gh_ctx *ctx)
{
  for (int b = 0; b < 2; b++) {
    for (int f = 0; f < D_COUNT; f++) { if (ctx->fbuf[b][f]) (void) hipFree(ctx->fbuf[b][f]); ctx->fbuf[b][f] = nullptr; }
    if (ctx->iorig[b]) (void) hipFree(ctx->iorig[b]); ctx->iorig[b] = nullptr;
    for (int k = 0; k < 3; k++) { if (ctx->P[b][k]) (void) hipFree(ctx->P[b][k]); ctx->P[b][k] = nullptr; }
    if (ctx->cellnode[b]) (void) hipFree(ctx->cellnode[b]); ctx->cellnode[b] = nullptr;
  }
  for (int k = 0; k < 3; k++) {
    if (ctx->W[k]) (void) hipFree(ctx->W[k]); ctx->W[k] = nullptr;
    if (ctx->Wpre[k]) (void) hipFree(ctx->Wpre[k]); ctx->Wpre[k] = nullptr;
  }
  void *ptrs[] = {ctx->posm, ctx->hrec, ctx->side, ctx->sortkeys_out, ctx->sortvals, ctx->qs_ids, ctx->qs_keys, ctx->pm_invhsqd, ctx->pm_cullsqd,
                  ctx->qw_k[0], ctx->qw_k[1], ctx->qw_i[0], ctx->qw_i[1], ctx->qw_rk, ctx->qw_gp, ctx->qw_blk, ctx->qw_st};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  ctx->qs_ids = nullptr; ctx->qs_keys = nullptr; ctx->pm_invhsqd = nullptr; ctx->pm_cullsqd = nullptr;
  ctx->qw_k[0] = ctx->qw_k[1] = nullptr; ctx->qw_i[0] = ctx->qw_i[1] = nullptr; ctx->qw_rk = nullptr; ctx->qw_gp = nullptr; ctx->qw_blk = nullptr; ctx->qw_st = nullptr; ctx->qw_words = 0;
  ctx->posm = nullptr; ctx->hrec = nullptr; ctx->side = nullptr; ctx->sortkeys_out = nullptr; ctx->sortvals = nullptr;
  ctx->iota_N = -1;
  ctx->Ncap = 0; ctx->N = 0; ctx->tree_layout_N = -1; ctx->tree_valid = false;
}

extern "C" void gh_destroy(gh_ctx *ctx)
{
  if (!ctx) return;
  if (ctx->stream) (void) hipStreamSynchronize(ctx->stream);
  gh_dd_free(ctx);
  gh_sinks_free(ctx);
  free_particles(ctx);
  void *ptrs[] = {ctx->dl_rl, ctx->dl_rlen, ctx->d_blk, ctx->gl_cells, ctx->gl_dirl, ctx->gl_hydl, ctx->gl_len, ctx->gl_gcells, ctx->gl_glen, ctx->cfirst, ctx->cN, ctx->cleft, ctx->cbox, ctx->ch, ctx->cgeo, ctx->ccom, ctx->cquad, ctx->cvel, ctx->leafact, ctx->star_posm, ctx->star_h, ctx->star_out, ctx->ktab, ctx->leaf_amin, ctx->dbbmin, ctx->dbbmax, ctx->kdiv, ctx->sorttemp,
                  ctx->redbuf, ctx->d_stats, ctx->d_flags, ctx->d_ptrtab};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  for (int ph = 0; ph < GH_T_COUNT; ph++) for (auto &p : ctx->ev_used[ph]) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
  for (auto &p : ctx->ev_free) { (void) hipEventDestroy(p.a); (void) hipEventDestroy(p.b); }
  for (int i = 0; i < 2; i++) { if (ctx->aux[i]) (void) hipStreamDestroy(ctx->aux[i]); if (ctx->ev_join[i]) (void) hipEventDestroy(ctx->ev_join[i]); }
  if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
  if (ctx->stream) (void) hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char *gh_last_error(const gh_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" int64_t gh_num_particles(const gh_ctx *ctx) { return ctx ? ctx->N : 0; }

int gh_alloc_particles(gh_ctx *ctx, int64_t N)
{
  if (N <= ctx->Ncap) { ctx->N = N; return GH_OK; }
  free_particles(ctx);
  const size_t n = (size_t) N;
  for (int b = 0; b < 2; b++) {
    for (int f = 0; f < D_COUNT; f++) {
      GH_CHECK(ctx, hipMalloc((void**) &ctx->fbuf[b][f], sizeof(double)*n));
      GH_CHECK(ctx, hipMemset(ctx->fbuf[b][f], 0, sizeof(double)*n));
    }
    GH_CHECK(ctx, hipMalloc((void**) &ctx->iorig[b], sizeof(int)*n));
    for (int k = 0; k < 3; k++) GH_CHECK(ctx, hipMalloc((void**) &ctx->P[b][k], sizeof(int)*n));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->cellnode[b], sizeof(int)*n));
  }
  const size_t nwords = (n + 63)/64 + 1;
  for (int k = 0; k < 3; k++) {
    GH_CHECK(ctx, hipMalloc((void**) &ctx->W[k], sizeof(unsigned long long)*nwords));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->Wpre[k], sizeof(unsigned int)*nwords));
  }
  GH_CHECK(ctx, hipMalloc((void**) &ctx->posm, sizeof(double4)*n));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->hrec, sizeof(double4)*4*n));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->side, n));
  if (ctx->cfg.sink_particles && ctx->cfg.create_sinks == 1) {
    GH_CHECK(ctx, hipMalloc((void**) &ctx->pm_invhsqd, sizeof(double)*n));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->pm_cullsqd, sizeof(double)*n));
  }
  GH_CHECK(ctx, hipMalloc((void**) &ctx->sortkeys_out, sizeof(double)*3*n));
  GH_CHECK(ctx, hipMalloc((void**) &ctx->sortvals, sizeof(int)*n));
  {
    // permute tables: [0] = gather buffer 0 -> 1, [1] = gather buffer 1 -> 0 (src pointers then dst pointers)
    double *tab[4*D_COUNT];
    for (int c = 0; c < 2; c++)
      for (int f = 0; f < D_COUNT; f++) { tab[c*2*D_COUNT + f] = ctx->fbuf[c][f]; tab[c*2*D_COUNT + D_COUNT + f] = ctx->fbuf[c ^ 1][f]; }
    GH_CHECK(ctx, hipMemcpy(ctx->d_ptrtab, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  ctx->Ncap = N; ctx->N = N;
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// particle transfer (caller order <-> tree order)
// ------------------------------------------------------------------------------------------------
extern "C" int gh_upload_particles(gh_ctx *ctx, int64_t N, const double *r, const double *v, const double *m,
                                   const double *h, const double *u)
{
  if (!ctx || N <= 0 || !r || !m || !h) return ctx ? gh_fail(ctx, GH_ERR_INVALID, "gh_upload_particles: bad arguments") : GH_ERR_INVALID;
  if (N >= (1ll << 31)) return gh_fail(ctx, GH_ERR_INVALID, "N too large");
  int rc = gh_alloc_particles(ctx, N);
  if (rc) return rc;
  if ((rc = gh_alloc_tree(ctx))) return rc;              // fixes this rank's own particle range
  ctx->cur = 0;
  const int nd = ctx->ndim;
  const size_t n = (size_t) N;
  // multi-GPU: the caller passes the whole initial condition on every rank; this rank keeps positions
  // [own_first, own_first + own_count) of it (any split will do - the first tree build migrates every particle to the
  // rank that owns its cell).  Nothing else is uploaded.
  const size_t o0 = (size_t) ctx->own_first, on = (size_t) ctx->own_count;
  std::vector<double> tmp(on);
  auto put = [&](int comp, const double *src, int stride, int off, double fill) -> int {
    if (src) for (size_t i = 0; i < on; i++) tmp[i] = src[(o0 + i)*stride + off];
    else std::fill(tmp.begin(), tmp.end(), fill);
    GH_CHECK(ctx, hipMemcpy(ctx->fbuf[0][comp] + o0, tmp.data(), sizeof(double)*on, hipMemcpyHostToDevice));
    return GH_OK;
  };
  for (int f = 0; f < D_COUNT; f++) GH_CHECK(ctx, hipMemset(ctx->fbuf[0][f], 0, sizeof(double)*n));
  for (int k = 0; k < nd; k++) {
    if ((rc = put(D_RX + k, r, nd, k, 0.0))) return rc;
    if ((rc = put(D_R0X + k, r, nd, k, 0.0))) return rc;
    if ((rc = put(D_VX + k, v, nd, k, 0.0))) return rc;
    if ((rc = put(D_V0X + k, v, nd, k, 0.0))) return rc;
  }
  if ((rc = put(D_M, m, 1, 0, 0.0))) return rc;
  if ((rc = put(D_H, h, 1, 0, 0.0))) return rc;
  if ((rc = put(D_U, u, 1, 0, 0.0))) return rc;
  if ((rc = put(D_U0, u, 1, 0, 0.0))) return rc;
  // SphSimulation.cpp:252-257: alpha = alpha_visc, or alpha_visc_min with time-dependent viscosity
  if ((rc = put(D_ALPHA, nullptr, 1, 0, (ctx->cfg.avisc == GH_AVISC_MON97MM97 || ctx->cfg.avisc == GH_AVISC_MON97CD2010) ? ctx->cfg.alpha_visc_min : ctx->cfg.alpha_visc))) return rc;
  if ((rc = put(D_FLAGS, nullptr, 1, 0, 1.0))) return rc;              // every particle active until the first EndTimestep
  if ((rc = put(D_SINKID, nullptr, 1, 0, -1.0))) return rc;            // Particle constructor, Particle.h:183
  ctx->sinks.clear(); ctx->mmean = 0.0;
  std::vector<int> ids(n);
  for (size_t i = 0; i < n; i++) ids[i] = (int) i;
  GH_CHECK(ctx, hipMemcpy(ctx->iorig[0], ids.data(), sizeof(int)*n, hipMemcpyHostToDevice));
  ctx->tree_valid = false; ctx->tree_valid_once = false;
  ctx->n = 0; ctx->Nsteps = 0; ctx->t = 0.0; ctx->timestep = 0.0;
  ctx->nresync = 0; ctx->level_max = 0; ctx->level_step = 0; ctx->dt_max = 0.0;
  ctx->rebuild_tree = true;
  ctx->exact_armed = false;
  ctx->sink_exact = true; ctx->sink_exact_sticky = false;
  ctx->glist_checked = -1;
  double tt[3] = {0.0, 0.0, 0.0};
  GH_CHECK(ctx, hipMemcpy(gh_time_dev(ctx), tt, sizeof(tt), hipMemcpyHostToDevice));
  GH_CHECK(ctx, hipMemset(ctx->d_blk, 0, sizeof(int)*24));
  { const int none = -1; GH_CHECK(ctx, hipMemcpy(ctx->d_blk + 19, &none, sizeof(int), hipMemcpyHostToDevice)); }     // "no density pass has counted yet"
  return GH_OK;
}

static int fetch_iorig(gh_ctx *ctx, std::vector<int> &ids)
{
  ids.resize((size_t) ctx->N);
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  GH_CHECK(ctx, hipMemcpy(ids.data(), ctx->iorig[ctx->cur], sizeof(int)*ids.size(), hipMemcpyDeviceToHost));
  return GH_OK;
}

extern "C" int gh_download(gh_ctx *ctx, int field, double *dst)
{
  if (!ctx || !dst) return GH_ERR_INVALID;
  int first, nc;
  if (field_comp(field, &first, &nc, ctx->ndim)) return gh_fail(ctx, GH_ERR_INVALID, "gh_download: bad field");
  std::vector<int> ids;
  int rc = fetch_iorig(ctx, ids);
  if (rc) return rc;
  // multi-GPU: only this rank's own particles are written (the caller merges the ranks' arrays)
  const size_t o0 = (size_t) ctx->own_first, n = (size_t) ctx->own_count;
  std::vector<double> tmp(n);
  for (int k = 0; k < nc; k++) {
    GH_CHECK(ctx, hipMemcpy(tmp.data(), ctx->fbuf[ctx->cur][first + k] + o0, sizeof(double)*n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) dst[(size_t) ids[o0 + i]*nc + k] = tmp[i];
  }
  return GH_OK;
}

extern "C" int gh_upload_field(gh_ctx *ctx, int field, const double *src)
{
  if (!ctx || !src) return GH_ERR_INVALID;
  int first, nc;
  if (field_comp(field, &first, &nc, ctx->ndim)) return gh_fail(ctx, GH_ERR_INVALID, "gh_upload_field: bad field");
  std::vector<int> ids;
  int rc = fetch_iorig(ctx, ids);
  if (rc) return rc;
  const size_t o0 = (size_t) ctx->own_first, n = (size_t) ctx->own_count;
  std::vector<double> tmp(n);
  for (int k = 0; k < nc; k++) {
    for (size_t i = 0; i < n; i++) tmp[i] = src[(size_t) ids[o0 + i]*nc + k];
    GH_CHECK(ctx, hipMemcpy(ctx->fbuf[ctx->cur][first + k] + o0, tmp.data(), sizeof(double)*n, hipMemcpyHostToDevice));
  }
  if (field == GH_F_R || field == GH_F_M) gh_pack_posm(ctx);
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// tree
// ------------------------------------------------------------------------------------------------
extern "C" int gh_build_tree(gh_ctx *ctx)
{
  if (!ctx || ctx->N <= 0) return GH_ERR_INVALID;
  int rc = gh_tree_build_checked(ctx);
  if (rc) return rc;
  return gh_sync_collect(ctx, "gh_build_tree");
}

int gh_tree_restock_impl(gh_ctx *ctx);
// HydroTree::BuildTree with its own arguments (HydroTree.cpp:310-372): rebuild when n % ntreebuildstep == 0 or rebuild_tree,
// re-stock when n % ntreestockstep == 0, otherwise let the cells drift by their mean velocity times `timestep`
extern "C" int gh_build_tree_scheduled(gh_ctx *ctx, int rebuild_tree, int n, int ntreebuildstep, int ntreestockstep, double timestep)
{
  if (!ctx || ctx->N <= 0 || ntreebuildstep < 1 || ntreestockstep < 1) return GH_ERR_INVALID;
  if (ctx->nranks > 1 && (ntreebuildstep > 1)) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: the tree is rebuilt every step (ntreebuildstep = 1)");
  int rc;
  if (n%ntreebuildstep == 0 || rebuild_tree || !ctx->tree_valid_once || ctx->tree_layout_N != ctx->N) rc = gh_tree_build_checked(ctx);
  else if (n%ntreestockstep == 0) rc = gh_tree_restock_impl(ctx);
  else {
    ctx->timestep = timestep;
    double tt[2] = {ctx->t, timestep};
    GH_CHECK(ctx, hipMemcpy(gh_time_dev(ctx), tt, sizeof(tt), hipMemcpyHostToDevice));
    rc = gh_tree_extrapolate_impl(ctx);
  }
  if (rc) return rc;
  ctx->tree_valid_once = true;
  return gh_sync_collect(ctx, "gh_build_tree_scheduled");
}

extern "C" int gh_tree_size(gh_ctx *ctx, int32_t *Ncell, int32_t *ltot, int32_t *gtot)
{
  if (!ctx || !ctx->tree_valid) return GH_ERR_INVALID;
  if (Ncell) *Ncell = ctx->Ncell;
  if (ltot) *ltot = ctx->ltot;
  if (gtot) *gtot = ctx->gtot;
  return GH_OK;
}

extern "C" int gh_export_tree(gh_ctx *ctx, int32_t *cell_level, int32_t *cell_first, int32_t *cell_N,
                              double *bbmin, double *bbmax, double *hboxmin, double *hboxmax, double *rcell,
                              double *com, double *mass, double *rmax, double *hmax, double *cdistsqd,
                              int32_t *order)
{
  if (!ctx || !ctx->tree_valid) return ctx ? gh_fail(ctx, GH_ERR_INVALID, "gh_export_tree: no tree") : GH_ERR_INVALID;
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const int Nc = ctx->Ncell, nd = ctx->ndim;
  std::vector<CellBox> hb(Nc);
  std::vector<CellH> hh(Nc);
  std::vector<CellGeo> hg(Nc);
  std::vector<CellCom> hc(Nc);
  GH_CHECK(ctx, hipMemcpy(hb.data(), ctx->cbox, sizeof(CellBox)*Nc, hipMemcpyDeviceToHost));
  GH_CHECK(ctx, hipMemcpy(hh.data(), ctx->ch, sizeof(CellH)*Nc, hipMemcpyDeviceToHost));
  GH_CHECK(ctx, hipMemcpy(hg.data(), ctx->cgeo, sizeof(CellGeo)*Nc, hipMemcpyDeviceToHost));
  GH_CHECK(ctx, hipMemcpy(hc.data(), ctx->ccom, sizeof(CellCom)*Nc, hipMemcpyDeviceToHost));
  // heap index -> reference pre-order id: child1 = c+1, child2 = c + 2^(ltot-level)  (KDTree.cpp:412-417)
  std::vector<int> pre(Nc), lev(Nc);
  pre[0] = 0; lev[0] = 0;
  for (int n = 0; n < ctx->gtot - 1; n++) {
    pre[2*n + 1] = pre[n] + 1;
    pre[2*n + 2] = pre[n] + (1 << (ctx->ltot - lev[n]));
    lev[2*n + 1] = lev[2*n + 2] = lev[n] + 1;
  }
  for (int n = 0; n < Nc; n++) {
    const int c = pre[n];
    if (cell_level) cell_level[c] = lev[n];
    if (cell_first) cell_first[c] = ctx->h_cfirst[n];
    if (cell_N) cell_N[c] = ctx->h_cN[n];
    for (int k = 0; k < nd; k++) {
      if (bbmin) bbmin[c*nd + k] = hb[n].bbmin[k];
      if (bbmax) bbmax[c*nd + k] = hb[n].bbmax[k];
      if (hboxmin) hboxmin[c*nd + k] = hh[n].hbmin[k];
      if (hboxmax) hboxmax[c*nd + k] = hh[n].hbmax[k];
      if (rcell) rcell[c*nd + k] = hg[n].rcell[k];
      if (com) com[c*nd + k] = hc[n].com[k];
    }
    if (mass) mass[c] = hc[n].m;
    if (rmax) rmax[c] = hg[n].rmax;
    if (hmax) hmax[c] = hh[n].hmax;
    if (cdistsqd) cdistsqd[c] = hg[n].cdistsqd;
  }
  if (order) GH_CHECK(ctx, hipMemcpy(order, ctx->iorig[ctx->cur], sizeof(int)*(size_t) ctx->N, hipMemcpyDeviceToHost));
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// hot path entry points
// ------------------------------------------------------------------------------------------------
static int density_and_hmax(gh_ctx *ctx, bool count)
{
  int rc = gh_dd_exchange(ctx, GH_HALO_DENSITY);         // multi-GPU: import the gather halo (cells + positions)
  if (rc) return rc;
  if ((rc = gh_density_impl(ctx, count))) return rc;
  gh_zeta_stars_impl(ctx);                              // hybrid runs: star term of zeta (GradhSph.cpp:288-307)
  if ((rc = gh_sinks_potmin(ctx))) return rc;           // sink runs: potential-minimum flag at the end of ComputeH (GradhSph.cpp:270-280)
  if (ctx->cfg.avisc == GH_AVISC_MON97CD2010) {         // Cullen & Dehnen switch at the end of ComputeH (GradhSph.cpp:319-321)
    if ((rc = gh_cullen_dehnen_impl(ctx))) return rc;
  }
  return gh_update_hmax_impl(ctx);                      // tree->UpdateAllHmaxValues, GradhSphTree.cpp:268
}

extern "C" int gh_update_density(gh_ctx *ctx, gh_stats *stats)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->dom_ms[GH_T_SPH_PROPERTIES] = 0; ctx->dom_calls[GH_T_SPH_PROPERTIES] = 0;
  int rc = density_and_hmax(ctx, stats != nullptr);
  if (rc) return rc;
  rc = gh_sync_collect(ctx, "gh_update_density");
  if (rc) return rc;
  if (stats) return read_stats(ctx, stats, GH_T_SPH_PROPERTIES);
  return GH_OK;
}

extern "C" int gh_zero_accelerations(gh_ctx *ctx)
{
  if (!ctx) return GH_ERR_INVALID;
  gh_zero_acc_impl(ctx);
  return gh_sync_collect(ctx, "gh_zero_accelerations");
}

extern "C" int gh_update_hydro_forces(gh_ctx *ctx, gh_stats *stats)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->dom_ms[GH_T_SPH_FORCES] = 0; ctx->dom_calls[GH_T_SPH_FORCES] = 0;
  int rc = gh_hydro_forces_impl(ctx, stats != nullptr);
  if (rc) return rc;
  if ((rc = gh_dd_return_levelneib(ctx))) return rc;
  rc = gh_sync_collect(ctx, "gh_update_hydro_forces");
  if (rc) return rc;
  if (stats) return read_stats(ctx, stats, GH_T_SPH_FORCES);
  return GH_OK;
}

extern "C" int gh_update_all_forces(gh_ctx *ctx, gh_stats *stats)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->dom_ms[GH_T_SPH_FORCES] = 0; ctx->dom_calls[GH_T_SPH_FORCES] = 0;
  int rc = gh_all_forces_impl(ctx, stats != nullptr);
  if (rc) return rc;
  gh_gas_star_forces_impl(ctx);                         // hybrid runs: gas <- stars (GradhSphTree.cpp:600-607)
  if ((rc = gh_dd_return_levelneib(ctx))) return rc;
  rc = gh_sync_collect(ctx, "gh_update_all_forces");
  if (rc) return rc;
  if (stats) return read_stats(ctx, stats, GH_T_SPH_FORCES);
  return GH_OK;
}

static int forces_impl(gh_ctx *ctx)
{
  int rc;
  if (ctx->cfg.self_gravity) { if ((rc = gh_all_forces_impl(ctx, false))) return rc; if ((rc = gh_gas_star_forces_impl(ctx))) return rc; }
  else if (ctx->cfg.hydro_forces) { if ((rc = gh_hydro_forces_impl(ctx, false))) return rc; }
  else return gh_fail(ctx, GH_ERR_INVALID, "Error: No forces included in simulation");   // SphSimulation.cpp:474
  return gh_dd_return_levelneib(ctx);                   // multi-GPU + block timesteps: levelneib raised on halo copies goes home
}

// ------------------------------------------------------------------------------------------------
// KDK glue
// ------------------------------------------------------------------------------------------------
static int push_time(gh_ctx *ctx)
{
  double tt[2] = {ctx->t, ctx->timestep};
  GH_CHECK(ctx, hipMemcpyAsync(gh_time_dev(ctx), tt, sizeof(tt), hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

__global__ void k_block_tick(int *blk) { blk[0] = blk[0] + 1; }

// block-timestep clock: host mirror <-> device {n, nresync, level_max, level_step}, dt_max = time[2]
static int push_block(gh_ctx *ctx)
{
  int blk[8] = {ctx->n, ctx->nresync, ctx->level_max, ctx->level_step, 0, 0, 1, 1};
  GH_CHECK(ctx, hipMemcpy(ctx->d_blk, blk, sizeof(blk), hipMemcpyHostToDevice));
  GH_CHECK(ctx, hipMemcpy(gh_time_dev(ctx) + 2, &ctx->dt_max, sizeof(double), hipMemcpyHostToDevice));
  return GH_OK;
}
static int pull_block(gh_ctx *ctx)
{
  int blk[8];
  GH_CHECK(ctx, hipMemcpy(blk, ctx->d_blk, sizeof(blk), hipMemcpyDeviceToHost));
  ctx->n = blk[0]; ctx->nresync = blk[1]; ctx->level_max = blk[2]; ctx->level_step = blk[3];
  GH_CHECK(ctx, hipMemcpy(&ctx->dt_max, gh_time_dev(ctx) + 2, sizeof(double), hipMemcpyDeviceToHost));
  return GH_OK;
}

extern "C" int gh_set_block_clock(gh_ctx *ctx, int n, int nresync, int level_max, int level_step, double dt_max)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->n = n; ctx->nresync = nresync; ctx->level_max = level_max; ctx->level_step = level_step; ctx->dt_max = dt_max;
  return push_block(ctx);
}

extern "C" int gh_get_active_count(gh_ctx *ctx, int64_t *nactive, int reset)
{
  if (!ctx || !nactive) return GH_ERR_INVALID;
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  unsigned long long v = 0;
  GH_CHECK(ctx, hipMemcpy(&v, ctx->d_blk + 8, sizeof(v), hipMemcpyDeviceToHost));
  *nactive = (int64_t) v;
  if (reset) GH_CHECK(ctx, hipMemset(ctx->d_blk + 8, 0, sizeof(v)));
  return GH_OK;
}

extern "C" int gh_get_block_clock(gh_ctx *ctx, int32_t *clock4, double *dt_max)
{
  if (!ctx) return GH_ERR_INVALID;
  if (clock4) { clock4[0] = ctx->n; clock4[1] = ctx->nresync; clock4[2] = ctx->level_max; clock4[3] = ctx->level_step; }
  if (dt_max) *dt_max = ctx->dt_max;
  return GH_OK;
}

static int pull_time(gh_ctx *ctx)
{
  double tt[2];
  GH_CHECK(ctx, hipMemcpy(tt, gh_time_dev(ctx), sizeof(tt), hipMemcpyDeviceToHost));
  ctx->t = tt[0]; ctx->timestep = tt[1];
  return GH_OK;
}

extern "C" int gh_kdk_advance(gh_ctx *ctx, int n, double t, double timestep)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->n = n; ctx->t = t; ctx->timestep = timestep;
  int rc = push_time(ctx);
  if (rc) return rc;
  gh_phase_begin(ctx, GH_T_KDK);
  gh_kdk_advance_impl(ctx, n, t, timestep);
  gh_phase_end(ctx, GH_T_KDK);
  ctx->tree_valid = false;
  return gh_sync_collect(ctx, "gh_kdk_advance");
}

extern "C" int gh_compute_global_timestep(gh_ctx *ctx, double *dt_min)
{
  if (!ctx) return GH_ERR_INVALID;
  gh_phase_begin(ctx, GH_T_KDK);
  gh_timestep_impl(ctx);
  gh_phase_end(ctx, GH_T_KDK);
  int rc = gh_sync_collect(ctx, "gh_compute_global_timestep");
  if (rc) return rc;
  rc = pull_time(ctx);
  if (rc) return rc;
  ctx->n = 0;
  if (dt_min) *dt_min = ctx->timestep;
  return GH_OK;
}

extern "C" int gh_kdk_end(gh_ctx *ctx, int n, double t, double timestep)
{
  if (!ctx) return GH_ERR_INVALID;
  (void) n; (void) timestep;
  ctx->t = t;
  double tt = t;
  GH_CHECK(ctx, hipMemcpy(gh_time_dev(ctx), &tt, sizeof(double), hipMemcpyHostToDevice));
  gh_phase_begin(ctx, GH_T_KDK);
  gh_kdk_end_impl(ctx, n, t, timestep);
  gh_phase_end(ctx, GH_T_KDK);
  return gh_sync_collect(ctx, "gh_kdk_end");
}

// ------------------------------------------------------------------------------------------------
// whole steps
// ------------------------------------------------------------------------------------------------
static int build_tree_timed(gh_ctx *ctx)
{
  gh_phase_begin(ctx, GH_T_BUILD_TREE);
  int rc = gh_tree_build_impl(ctx);
  gh_phase_end(ctx, GH_T_BUILD_TREE);
  return rc;
}

// build at a point where the host may synchronise (gh_build_tree, the setup passes): a build that meets the context's
// first tie between equal coordinates at a median is redone at once in exact mode, so that lattice initial conditions
// get the reference's tree from the first pass on.  (A first tie inside a multi-step gh_step call is resolved by the
// fast rule for that one build; the exact kernels are armed from the next synchronisation on.)
int gh_tree_build_checked(gh_ctx *ctx)
{
  int rc = build_tree_timed(ctx);
  if (rc || ctx->exact_armed || ctx->nranks > 1) return rc;
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  int tie[2] = {0, 0};
  GH_CHECK(ctx, hipMemcpy(tie, ctx->d_blk + 13, sizeof(tie), hipMemcpyDeviceToHost));
  if (!(tie[0] | tie[1])) return GH_OK;
  ctx->exact_armed = true;
  return build_tree_timed(ctx);
}

// HydroTree::BuildTree as MainLoop calls it (HydroTree.cpp:325-343): rebuild every ntreebuildstep steps and on the first
// step after the setup (rebuild_tree), re-stock the existing tree otherwise
static int step_tree_timed(gh_ctx *ctx)
{
  const int ntb = ctx->cfg.ntreebuildstep;
  // sink runs: accreted particles leave the arrays before the tree is rebuilt (HydroTree.cpp:334-335)
  if (ctx->cfg.sink_particles) { const int rc = gh_sinks_delete_dead(ctx); if (rc) return rc; }
  if (ntb <= 1 || ctx->nranks > 1 || ctx->rebuild_tree || ctx->Nsteps%ntb == 0 || ctx->tree_layout_N != ctx->N) return build_tree_timed(ctx);
  gh_phase_begin(ctx, GH_T_BUILD_TREE);
  // re-stock every ntreestockstep steps, otherwise let the cells drift with their mean velocity (Tree.cpp:172-198)
  int rc = (ctx->cfg.ntreestockstep <= 1 || ctx->Nsteps%ctx->cfg.ntreestockstep == 0) ? gh_tree_restock_impl(ctx) : gh_tree_extrapolate_impl(ctx);
  gh_phase_end(ctx, GH_T_BUILD_TREE);
  return rc;
}

// the gas passes of one MainLoop call, enqueued only (used by gh_hybrid_step, nbody.hip)
int gh_hybrid_gas_passes(gh_ctx *ctx)
{
  int rc;
  if ((rc = step_tree_timed(ctx))) return rc;
  if ((rc = density_and_hmax(ctx, false))) return rc;
  gh_zero_acc_impl(ctx);
  if ((rc = forces_impl(ctx))) return rc;
  ctx->rebuild_tree = false;
  return GH_OK;
}

int gh_setup_passes(gh_ctx *ctx, int initial_h_provided);

extern "C" int gh_setup(gh_ctx *ctx, int initial_h_provided, double *timestep)
{
  if (!ctx || ctx->N <= 0) return GH_ERR_INVALID;
  int rc;
  if ((rc = gh_setup_passes(ctx, initial_h_provided))) return rc;
  // r0,v0,a0 = r,v,a (:483-489) happens in kdk_end below because dt = 0 leaves v unchanged
  // the simulation time is 0 after gh_upload_particles, or what gh_set_time put there (runs started from a snapshot)
  ctx->timestep = 0.0; ctx->n = 0;
  if ((rc = push_time(ctx))) return rc;
  if (ctx->cfg.Nlevels > 1) {
    ctx->nresync = 0;
    if ((rc = push_block(ctx))) return rc;
    gh_block_timesteps_impl(ctx);                        // ComputeBlockTimesteps (:539): n == nresync == 0, resynchronise
  }
  else gh_timestep_impl(ctx);                            // ComputeGlobalTimestep (:538)
  gh_kdk_end_impl(ctx, 0, 0.0, 0.0);                     // hydroint->EndTimestep (:551)
  if ((rc = gh_sync_collect(ctx, "gh_setup"))) return rc;
  if ((rc = pull_time(ctx))) return rc;
  if (ctx->cfg.Nlevels > 1 && (rc = pull_block(ctx))) return rc;
  if (timestep) *timestep = ctx->timestep;
  return GH_OK;
}

// the density and force passes of PostInitialConditionsSetup (enqueued; also the first half of gh_hybrid_setup, nbody.hip)
int gh_setup_passes(gh_ctx *ctx, int initial_h_provided)
{
  int rc;
  // SphSimulation::PostInitialConditionsSetup (SphSimulation.cpp:266-345): density with the guessed h
  // first if no h was provided, then tree + density, then (iteration loop :381-473) tree + density + forces
  const int npass = initial_h_provided ? 2 : 3;
  for (int p = 0; p < npass; p++) {
    if ((rc = gh_tree_build_checked(ctx))) return rc;
    if ((rc = density_and_hmax(ctx, false))) return rc;
    if ((rc = gh_sync_collect(ctx, "gh_setup/density"))) return rc;
  }
  // relative MACs need accelerations: the reference runs this force pass with the geometric MAC, rebuilds the tree
  // (which stocks amin from it) and repeats the pass with the requested MAC (SphSimulation.cpp:381-473)
  const bool relmac = ctx->cfg.self_gravity && ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC;
  ctx->mac_bootstrap = relmac;
  gh_zero_acc_impl(ctx);
  if ((rc = forces_impl(ctx))) return rc;
  ctx->mac_bootstrap = false;
  if (relmac) {
    if ((rc = gh_tree_build_checked(ctx))) return rc;
    gh_zero_acc_impl(ctx);
    if ((rc = forces_impl(ctx))) return rc;
  }
  return GH_OK;
}

extern "C" int gh_set_time(gh_ctx *ctx, double t, double timestep)
{
  if (!ctx) return GH_ERR_INVALID;
  ctx->t = t; ctx->timestep = timestep;
  return push_time(ctx);
}

// one MainLoop call with hierarchical block timesteps (SphSimulation.cpp:574-880, Nlevels > 1): only the particles at
// the end of their own step are active (density, forces, kick); neighbours on longer steps are drifted (r, v, u) and
// their pressure / sound refreshed; CheckTimesteps may wake neighbours of fast particles, which repeats the passes
// the tree and the density / force passes of a block-timestep step, repeated while CheckTimesteps wakes particles
// (SphSimulation.cpp:634-755); synchronises
int gh_block_gas_passes(gh_ctx *ctx)
{
  int rc;
  if ((rc = step_tree_timed(ctx))) return rc;
  for (int pass = 0; ; pass++) {
    if (pass > 0) gh_leaf_active_counters(ctx);          // "if (activecount > 0) UpdateActiveParticleCounters" (:663)
    if ((rc = density_and_hmax(ctx, false))) return rc;
    gh_zero_acc_impl(ctx);
    gh_thermal_all_impl(ctx);                            // :665-679, nradstep = 1
    if ((rc = forces_impl(ctx))) return rc;
    gh_check_timesteps_impl(ctx);                        // all active flags off, then CheckTimesteps (:739-750)
    if ((rc = gh_sync_collect(ctx, "gh_step/block"))) return rc;
    int blk[8];
    GH_CHECK(ctx, hipMemcpy(blk, ctx->d_blk, sizeof(blk), hipMemcpyDeviceToHost));
    if (blk[5] == 0) break;
    GH_CHECK(ctx, hipMemsetAsync(ctx->d_blk + 5, 0, sizeof(int), ctx->stream));
  }
  return GH_OK;
}
// n = n + 1, t = t + timestep, drift (the start of MainLoop, :585-600)
int gh_block_begin_step(gh_ctx *ctx)
{
  hipLaunchKernelGGL(k_block_tick, dim3(1), dim3(1), 0, ctx->stream, ctx->d_blk);   // n = n + 1 (:585)
  ctx->n++; ctx->Nsteps++;
  gh_advance_time_impl(ctx);
  gh_phase_begin(ctx, GH_T_KDK);
  gh_kdk_advance_impl(ctx, ctx->n, 0.0, 0.0);          // all particles drift; active = end of own step
  gh_phase_end(ctx, GH_T_KDK);
  return GH_OK;
}
int gh_block_pull(gh_ctx *ctx) { return pull_block(ctx); }

// one MainLoop call with hierarchical block timesteps (SphSimulation.cpp:574-880, Nlevels > 1): only the particles at
// the end of their own step are active (density, forces, kick); neighbours on longer steps are drifted (r, v, u) and
// their pressure / sound refreshed; CheckTimesteps may wake neighbours of fast particles, which repeats the passes
static int block_step(gh_ctx *ctx)
{
  int rc;
  gh_block_begin_step(ctx);
  if ((rc = gh_block_gas_passes(ctx))) return rc;
  gh_phase_begin(ctx, GH_T_KDK);
  gh_block_timesteps_impl(ctx);                          // ComputeBlockTimesteps (:842)
  gh_kdk_end_impl(ctx, 0, 0.0, 0.0);                     // EndTimestep of the particles that finished their step
  gh_phase_end(ctx, GH_T_KDK);
  if ((rc = gh_sync_collect(ctx, "gh_step/block"))) return rc;
  ctx->rebuild_tree = false;
  return pull_block(ctx);
}

extern "C" int gh_step(gh_ctx *ctx, int nsteps, double *t, double *timestep)
{
  if (!ctx || ctx->N <= 0) return GH_ERR_INVALID;
  int rc;
  if (ctx->cfg.Nlevels > 1) {
    for (int s = 0; s < nsteps; s++) if ((rc = block_step(ctx))) return rc;
    if ((rc = pull_time(ctx))) return rc;
    if (t) *t = ctx->t;
    if (timestep) *timestep = ctx->timestep;
    return GH_OK;
  }
  for (int s = 0; s < nsteps; s++) {
    // SphSimulation::MainLoop (SphSimulation.cpp:585-876), Nlevels = 1, no stars
    ctx->n++; ctx->Nsteps++;
    gh_advance_time_impl(ctx);                           // t = t + timestep
    gh_phase_begin(ctx, GH_T_KDK);
    gh_kdk_advance_impl(ctx, ctx->n, 0.0, 0.0);          // AdvanceParticles + CheckBoundaries
    gh_phase_end(ctx, GH_T_KDK);
    ctx->in_step = true;
    rc = step_tree_timed(ctx);                           // BuildTree: rebuild or re-stock
    ctx->in_step = false;
    if (rc) return rc;
    ctx->dd_defer_miss = ctx->nranks > 1;                // multi-GPU: the halo-miss check of the density pass rides in the force-phase exchange
    rc = density_and_hmax(ctx, false);                   // UpdateAllSphProperties
    ctx->dd_defer_miss = false;
    if (rc) return rc;
    gh_zero_acc_impl(ctx);                               // ZeroAccelerations
    if ((rc = forces_impl(ctx))) return rc;              // UpdateAllSph(Hydro)Forces
    gh_phase_begin(ctx, GH_T_KDK);
    gh_timestep_impl(ctx);                               // ComputeGlobalTimestep
    gh_kdk_end_impl(ctx, 0, 0.0, 0.0);                   // EndTimestep
    gh_phase_end(ctx, GH_T_KDK);
    ctx->n = 0;
    ctx->rebuild_tree = false;
  }
  if ((rc = gh_sync_collect(ctx, "gh_step"))) return rc;
  if ((rc = pull_time(ctx))) return rc;
  if (t) *t = ctx->t;
  if (timestep) *timestep = ctx->timestep;
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// timers / shards
// ------------------------------------------------------------------------------------------------
extern "C" int gh_get_timers(gh_ctx *ctx, double *ms, gh_stats *density, gh_stats *forces)
{
  if (!ctx) return GH_ERR_INVALID;
  if (ms) for (int k = 0; k < GH_T_COUNT; k++) ms[k] = ctx->timers[k];
  if (density) { *density = ctx->st_density; density->kernel_ms = ctx->dom_calls[GH_T_SPH_PROPERTIES] ? ctx->dom_ms[GH_T_SPH_PROPERTIES]/ctx->dom_calls[GH_T_SPH_PROPERTIES] : 0.0; }
  if (forces) { *forces = ctx->st_forces; forces->kernel_ms = ctx->dom_calls[GH_T_SPH_FORCES] ? ctx->dom_ms[GH_T_SPH_FORCES]/ctx->dom_calls[GH_T_SPH_FORCES] : 0.0; }
  return GH_OK;
}

extern "C" int gh_reset_timers(gh_ctx *ctx)
{
  if (!ctx) return GH_ERR_INVALID;
  for (int k = 0; k < GH_T_COUNT; k++) { ctx->timers[k] = 0.0; ctx->dom_ms[k] = 0.0; ctx->dom_calls[k] = 0; }
  return GH_OK;
}

extern "C" void *gh_stream(gh_ctx *ctx) { return ctx ? (void*) ctx->stream : nullptr; }

extern "C" int gh_update_hmax(gh_ctx *ctx)
{
  if (!ctx || !ctx->tree_valid) return GH_ERR_INVALID;
  gh_update_hmax_impl(ctx);
  return gh_sync_collect(ctx, "gh_update_hmax");
}

extern "C" void *gh_field_dev(gh_ctx *ctx, int field, int k)
{
  if (!ctx) return nullptr;
  int first, nc;
  if (field_comp(field, &first, &nc, ctx->ndim) || k < 0 || k >= nc) return nullptr;
  return ctx->fbuf[ctx->cur][first + k];
}

// ------------------------------------------------------------------------------------------------
// gather neighbour query (parity tests)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gather_count_fill(DevicePtrs d, Domain dom, double kernrange, int pass,
                                                          const long long *offsets, long long *counts, int *ids, int *flags)
{
  __shared__ WalkLDS<int> L;
  __shared__ double s_x[64], s_y[64], s_z[64];
  __shared__ int s_id[64];
  const int lane = threadIdx.x;
  const int q = blockIdx.x;
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (gN == 0) return;
  const bool act = lane < gN;
  const int i = gfirst + (act ? lane : 0);
  double ri[3] = {0, 0, 0};
  for (int k = 0; k < d.ndim; k++) ri[k] = d.f[D_RX + k][i];
  const double hi_ = d.f[D_H][i];
  const double rs = kernrange*hi_;
  const double rs2 = rs*rs;
  const CellBox gb = d.cbox[gnode];
  const double hs = wave_max(act ? hi_ : 0.0);
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    lo[k] = k < d.ndim ? gb.bbmin[k] - kernrange*hs : -1e300;
    hi[k] = k < d.ndim ? gb.bbmax[k] + kernrange*hs : 1e300;
  }
  const unsigned int codes = image_codes(dom, d.ndim, lo, hi);
  const int myorig = d.iorig[i];
  long long cnt = 0;
  const long long base = pass ? offsets[myorig] : 0;
  auto cls = [&](int n, int code, bool &open, bool &emit, int &first, int &c) {
    const CellBox b = d.cbox[n];
    const int cn = b.N;
    if (cn == 0) return;
    double sg[3], sh[3];
    code_xform(dom, code, sg, sh);
    for (int k = 0; k < d.ndim; k++) {
      double bmin, bmax;
      image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin, bmax);
      if (lo[k] > bmax) return;
      if (bmin > hi[k]) return;
    }
    if (n >= d.gtot - 1) { emit = true; first = b.first; c = cn; }
    else open = true;
  };
  auto tile = [&](bool valid, int j, int code) {
    double x = 1e30, y = 1e30, z = 1e30; int id = -1;
    if (valid) {
      double sg[3], sh[3]; code_xform(dom, code, sg, sh);
      x = sg[0]*d.f[D_RX][j] + sh[0];
      y = d.ndim > 1 ? sg[1]*d.f[D_RY][j] + sh[1] : 0.0;
      z = d.ndim > 2 ? sg[2]*d.f[D_RZ][j] + sh[2] : 0.0;
      id = d.iorig[j];
    }
    s_x[lane] = x; s_y[lane] = y; s_z[lane] = z; s_id[lane] = id;
    __syncthreads();
    if (act) {
      for (int c = 0; c < 64; c++) {
        if (s_id[c] < 0) continue;
        const double dx = s_x[c] - ri[0]; double r2 = dx*dx;
        if (d.ndim > 1) { const double dy = s_y[c] - ri[1]; r2 += dy*dy; }
        if (d.ndim > 2) { const double dz = s_z[c] - ri[2]; r2 += dz*dz; }
        if (r2 < rs2) {                                   // Tree.cpp:255 (strict <)
          if (pass) ids[base + cnt] = s_id[c];
          cnt++;
        }
      }
    }
    __syncthreads();
  };
  walk_dfs_stream(d, L, codes, cls, tile, flags);
  if (act && !pass) counts[myorig] = cnt;
}

extern "C" int gh_gather_neighbours(gh_ctx *ctx, int64_t cap, int64_t *offsets, int32_t *ids)
{
  if (!ctx || !offsets) return GH_ERR_INVALID;
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_INVALID, "gh_gather_neighbours: no tree");
  if (ctx->nranks > 1) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "gh_gather_neighbours: single-rank query");
  const size_t n = (size_t) ctx->N;
  long long *d_counts = nullptr, *d_off = nullptr; int *d_ids = nullptr;
  struct Scratch {          // freed on every return path
    long long *&a, *&b; int *&c;
    ~Scratch() { if (a) (void) hipFree(a); if (b) (void) hipFree(b); if (c) (void) hipFree(c); }
  } scratch{d_counts, d_off, d_ids};
  GH_CHECK(ctx, hipMalloc((void**) &d_counts, sizeof(long long)*n));
  GH_CHECK(ctx, hipMalloc((void**) &d_off, sizeof(long long)*(n + 1)));
  Domain dom; gh_fill_domain(ctx, dom);
  DevicePtrs d = gh_dev(ctx);
  const double kr = (ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0;
  hipLaunchKernelGGL(k_gather_count_fill, dim3(ctx->ngroups), dim3(64), 0, ctx->stream, d, dom, kr, 0, d_off, d_counts, d_ids, ctx->d_flags);
  int rc = gh_sync_collect(ctx, "gh_gather_neighbours");
  if (rc) return rc;
  std::vector<long long> cnt(n);
  GH_CHECK(ctx, hipMemcpy(cnt.data(), d_counts, sizeof(long long)*n, hipMemcpyDeviceToHost));
  offsets[0] = 0;
  for (size_t i = 0; i < n; i++) offsets[i + 1] = offsets[i] + cnt[i];
  if (offsets[n] > cap || !ids) return gh_fail(ctx, GH_ERR_CAPACITY, "gh_gather_neighbours: ids buffer too small");
  GH_CHECK(ctx, hipMemcpy(d_off, offsets, sizeof(long long)*(n + 1), hipMemcpyHostToDevice));
  GH_CHECK(ctx, hipMalloc((void**) &d_ids, sizeof(int)*(size_t) std::max<int64_t>(offsets[n], 1)));
  hipLaunchKernelGGL(k_gather_count_fill, dim3(ctx->ngroups), dim3(64), 0, ctx->stream, d, dom, kr, 1, d_off, d_counts, d_ids, ctx->d_flags);
  rc = gh_sync_collect(ctx, "gh_gather_neighbours");
  if (!rc) GH_CHECK(ctx, hipMemcpy(ids, d_ids, sizeof(int)*(size_t) offsets[n], hipMemcpyDeviceToHost));
  return rc;
}

// ------------------------------------------------------------------------------------------------
// point gather query (Tree::ComputeGatherNeighbourList(part, rp, rsearch, ...), Tree.cpp:208-280): ids of the particles
// with |r - rp|^2 < rsearch^2.  Cells are opened on |rcell - rp|^2 < (rsearch + rmax)^2 like the reference; one wave.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gather_at(DevicePtrs d, Domain dom, double px, double py, double pz, double rsearch, int cap, int *out, int *count, int *flags)
{
  __shared__ WalkLDS<int> L;
  const int lane = threadIdx.x;
  const unsigned long long lt = lanemask_lt();
  const double rp[3] = {px, py, pz};
  const double rs2 = rsearch*rsearch;
  // periodic / mirror domains: the reference also searches its ghost tree (HydroTree::GetGatherNeighbourList,
  // HydroTree.cpp:451-471) and callers map a ghost to its real parent - here the images are made on the fly and the
  // parent's id is what comes back
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) { lo[k] = rp[k] - rsearch; hi[k] = rp[k] + rsearch; }
  const unsigned int codes = image_codes(dom, d.ndim, lo, hi);
  int n_out = 0;
  auto cls = [&](int n, int code, bool &open, bool &emit, int &first, int &c) {
    const CellGeo g = d.cgeo[n];
    if (n >= d.gtot - 1 && g.N == 0) return;
    double sg[3], sh[3];
    code_xform(dom, code, sg, sh);
    double dd = 0.0;
    for (int k = 0; k < d.ndim; k++) { const double dx = sg[k]*g.rcell[k] + sh[k] - rp[k]; dd += dx*dx; }
    if (!(dd < (rsearch + g.rmax)*(rsearch + g.rmax))) return;
    if (n >= d.gtot - 1) { emit = true; first = g.first; c = g.N; }
    else open = true;
  };
  auto tile = [&](bool valid, int j, int code) {
    bool in = false;
    if (valid) {
      double sg[3], sh[3];
      code_xform(dom, code, sg, sh);
      double dd = 0.0;
      for (int k = 0; k < d.ndim; k++) { const double dx = sg[k]*d.f[D_RX + k][j] + sh[k] - rp[k]; dd += dx*dx; }
      in = dd < rs2;
      if (in && d.sinks && ((int) d.f[D_FLAGS][j] & GH_FLAG_DEAD)) in = false;       // "!partdata[i].flags.is_dead()", Tree.cpp:252
    }
    const unsigned long long m = __ballot(in);
    if (in) { const int pos = n_out + __popcll(m & lt); if (pos < cap) out[pos] = d.iorig[j]; }
    n_out += __popcll(m);
  };
  walk_dfs_stream(d, L, codes, cls, tile, flags);
  if (lane == 0) *count = n_out;
}

extern "C" int gh_gather_neighbours_at(gh_ctx *ctx, const double *rp, double rsearch, int32_t *list, int32_t cap)
{
  // -1 is the reference's overflow answer here, so argument errors use GH_ERR_UNSUPPORTED
  if (!ctx || !rp || !list || cap <= 0) return GH_ERR_UNSUPPORTED;
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "gh_gather_neighbours_at: no tree");
  if (ctx->nranks > 1) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "gh_gather_neighbours_at: single-rank query");
  int *d_out = nullptr, *d_cnt = nullptr;
  struct Scratch { int *&a, *&b; ~Scratch() { if (a) (void) hipFree(a); if (b) (void) hipFree(b); } } scratch{d_out, d_cnt};
  GH_CHECK(ctx, hipMalloc((void**) &d_out, sizeof(int)*(size_t) cap));
  GH_CHECK(ctx, hipMalloc((void**) &d_cnt, sizeof(int)));
  Domain dom;
  gh_fill_domain(ctx, dom);
  hipLaunchKernelGGL(k_gather_at, dim3(1), dim3(64), 0, ctx->stream, gh_dev(ctx), dom, rp[0], ctx->ndim > 1 ? rp[1] : 0.0, ctx->ndim > 2 ? rp[2] : 0.0,
                     rsearch, (int) cap, d_out, d_cnt, ctx->d_flags);
  int rc = gh_sync_collect(ctx, "gh_gather_neighbours_at");
  if (rc) return rc;
  int cnt = 0;
  GH_CHECK(ctx, hipMemcpy(&cnt, d_cnt, sizeof(int), hipMemcpyDeviceToHost));
  // the reference's "buffer too small" answer: it refuses a leaf cell unless Nneib + Nleafmax < Nneibmax (Tree.cpp:247,
  // 263-265) - which leaf trips that depends on its walk order, so the answer here is -1 whenever the complete list
  // leaves less headroom than one leaf cell (callers double the buffer and ask again, GradhSphTree.cpp:172-185)
  if (cnt + ctx->cfg.Nleafmax >= cap) return -1;
  GH_CHECK(ctx, hipMemcpy(list, d_out, sizeof(int)*(size_t) cnt, hipMemcpyDeviceToHost));
  return cnt;
}
