// stars.hip -- coupling of star (N-body) particles to the gas of a hybrid run.
//
// Replaces, for gas + stars runs (reference tests/gravhydro_tests/hybridplummer.dat lineage):
//   * the star term of zeta in GradhSph::ComputeH          (GradhSph.cpp:288-307, conservative_sph_star_gravity = 1)
//   * GradhSph::ComputeStarGravForces  (gas <- stars)      (GradhSph.cpp:699-743, called from GradhSphTree.cpp:600-607)
//   * HydroTree::UpdateAllStarGasForces (stars <- gas)     (HydroTree.cpp:552-657) with
//     Tree::ComputeStarGravityInteractionList (Tree.cpp:748-885), NbodyLeapfrogKDK::CalculateDirectHydroForces
//     (NbodyLeapfrogKDK.cpp:151-239), ComputeCellMonopoleForces / ComputeCellQuadrupoleForces (NeighbourSearch.h:350-475)
//
// The number of stars is small (tens to thousands): the gas-side terms are streaming kernels over the active gas
// particles with the star table in LDS; the star-side walk is one wavefront per star with a level-synchronous frontier in
// LDS, each lane classifying one tree node per round and evaluating what it accepts itself.
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

#define GH_STAR_TILE 256
#define GH_STAR_FRONT 4096

struct StarTab { const double4 *posm; const double *h; int n; int softening; };

__device__ __forceinline__ double star_wave_sum(double v)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// zeta += dh/drho * invomega * sum_s m_s invhsqd wzeta(s^2), on top of the gas-only zeta the density kernel stored
template <int ND, int KT>
__global__ void k_zeta_stars(DevicePtrs d, StarTab S, const double *ktab)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ double4 s_pm[GH_STAR_TILE];
  __shared__ double s_h[GH_STAR_TILE];
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  const bool act = i < d.N && (!d.levels || ((int) d.f[D_FLAGS][i] & 1));
  double r[3] = {0.0, 0.0, 0.0}, h = 1.0, sum = 0.0;
  if (act) { for (int k = 0; k < ND; k++) r[k] = d.f[D_RX + k][i]; h = d.f[D_H][i]; }
  for (int s0 = 0; s0 < S.n; s0 += GH_STAR_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < GH_STAR_TILE && s0 + t < S.n; t += blockDim.x) { s_pm[t] = S.posm[s0 + t]; s_h[t] = S.h[s0 + t]; }
    __syncthreads();
    const int nt = min(GH_STAR_TILE, S.n - s0);
    if (act) for (int t = 0; t < nt; t++) {
      const double4 p = s_pm[t];
      double ih = S.softening == 1 ? 2.0/(h + s_h[t]) : 2.0/h;
      const double ih2 = ih*ih;
      double dr2 = (p.x - r[0])*(p.x - r[0]);
      if (ND > 1) dr2 += (p.y - r[1])*(p.y - r[1]);
      if (ND > 2) dr2 += (p.z - r[2])*(p.z - r[2]);
      sum += p.w*ih2*K::t_wzetas2(dr2*ih2, ktab);
    }
  }
  if (act && !(d.sinks && d.f[D_SINKID][i] != -1.0)) {                         // (zeta = 0 inside a sink, GradhSph.cpp:309-312)
    const double deriv = -(1.0/(double) ND)*h/d.f[D_RHO][i];                  // h_rho_deriv, Sph.h:264
    d.f[D_ZETA][i] += deriv*sum*d.f[D_INVOMEGA][i];
  }
}

// gas <- stars: kernel-softened with the mean smoothing length, added to atree, a and gpot (not gpot_hydro)
template <int ND, int KT>
__global__ void k_gas_star_forces(DevicePtrs d, StarTab S, const double *ktab)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ double4 s_pm[GH_STAR_TILE];
  __shared__ double s_h[GH_STAR_TILE];
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  const bool act = i < d.N && (!d.levels || ((int) d.f[D_FLAGS][i] & 1));
  double r[3] = {0.0, 0.0, 0.0}, h = 1.0, a[3] = {0.0, 0.0, 0.0}, gp = 0.0;
  if (act) { for (int k = 0; k < ND; k++) r[k] = d.f[D_RX + k][i]; h = d.f[D_H][i]; }
  for (int s0 = 0; s0 < S.n; s0 += GH_STAR_TILE) {
    __syncthreads();
    for (int t = threadIdx.x; t < GH_STAR_TILE && s0 + t < S.n; t += blockDim.x) { s_pm[t] = S.posm[s0 + t]; s_h[t] = S.h[s0 + t]; }
    __syncthreads();
    const int nt = min(GH_STAR_TILE, S.n - s0);
    if (act) for (int t = 0; t < nt; t++) {
      const double4 p = s_pm[t];
      double dr[3] = {p.x - r[0], ND > 1 ? p.y - r[1] : 0.0, ND > 2 ? p.z - r[2] : 0.0};
      const double drsqd = dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2] + GH_SMALL;
      const double invdrmag = fast_rsqrt(drsqd);
      const double drmag = drsqd*invdrmag;
      const double ihm = 2.0/(h + s_h[t]);
      const double s = drmag*ihm, invs = 1.0/s;
      const double paux = p.w*ihm*ihm*K::t_wgrav(s, invs, ktab)*invdrmag;
      for (int k = 0; k < ND; k++) a[k] += paux*dr[k];
      gp += p.w*ihm*K::t_wpot(s, invs, ktab);
    }
  }
  if (act) {
    for (int k = 0; k < ND; k++) { d.f[D_ATX + k][i] += a[k]; d.f[D_AX + k][i] += a[k]; }
    d.f[D_GPOT][i] += gp;
  }
}

// stars <- gas: one wavefront per star
template <int ND, int KT, bool QUAD>
//
// Several ranks: every rank walks the shared top levels alike and its OWN subtree below them; a node's contribution is
// summed by the rank that owns it (level >= L: the rank whose cell it lies in; above: the rank of its leftmost descendant),
// the other ranks' subtrees are skipped, and the ranks' partial sums are added up on the host (gh_star_gas_forces) - the
// reference's MPI_Allreduce of the star forces (MpiControl / NbodySimulation: every rank holds all stars).  L = 0: one rank.
__global__ __launch_bounds__(64) void k_star_gas_forces(DevicePtrs d, StarTab S, const double *ktab, double *out_a, double *out_gpot, int *flags, int L, int self)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ int s_front[2][GH_STAR_FRONT];
  const int lane = threadIdx.x, si = blockIdx.x;
  const unsigned long long lt = lanemask_lt();
  const double4 sp = S.posm[si];
  const double sh = S.h[si];
  const double rs[3] = {sp.x, sp.y, sp.z};
  const double hrangemax = K::kernrange*sh;
  double a[3] = {0.0, 0.0, 0.0}, gp = 0.0;
  const int leaf0 = d.gtot - 1;
  int nf = 1, cur = 0;
  if (lane == 0) s_front[0][0] = 0;
  __syncthreads();
  while (nf > 0) {
    int nnext = 0;
    for (int base = 0; base < nf; base += 64) {
      const bool have = base + lane < nf;
      const int n = have ? s_front[cur][base + lane] : 0;
      bool open = false;
      const int lev = 31 - __clz(n + 1), jlev = n - ((1 << lev) - 1);
      const bool mine = (lev >= L ? jlev >> (lev - L) : jlev << (L - lev)) == self;
      if (have && (mine || lev < L)) {
        const CellGeo g = d.cgeo[n];
        double drsqd = 0.0;
        for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - rs[k]; drsqd += dx*dx; }
        const bool isleaf = n >= leaf0;
        const double dn = 0.5*hrangemax + g.rmax + 0.5*K::kernrange*g.hmax;
        if (drsqd < dn*dn) {                                   // overlap: smoothed neighbours (Tree.cpp:811-832)
          if (!isleaf) open = true;
          else for (int t = 0; t < g.N; t++) {
            const double4 p = d.posm[g.first + t];
            const double hj = d.f[D_H][g.first + t];
            double dr[3] = {p.x - rs[0], ND > 1 ? p.y - rs[1] : 0.0, ND > 2 ? p.z - rs[2] : 0.0};
            const double r2 = dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2];
            const double invdrmag = r2 > 0.0 ? fast_rsqrt(r2) : 0.0;
            const double drmag = r2*invdrmag;
            const double ihm = 2.0/(sh + hj);
            const double s = drmag*ihm, invs = s > 0.0 ? 1.0/s : 0.0;
            const double paux = p.w*ihm*ihm*K::t_wgrav(s, invs, ktab)*invdrmag;
            for (int k = 0; k < ND; k++) a[k] += paux*dr[k];
            gp += p.w*ihm*K::t_wpot(s, invs, ktab);
          }
        }
        else if (g.N > 0 && drsqd > g.cdistsqd) {              // far: cell, or its only particle (:834-851)
          if (!mine) { /* a shared top cell another rank sums */ }
          else if (isleaf && g.N == 1) {
            const double4 p = d.posm[g.first];
            double dr[3] = {p.x - rs[0], ND > 1 ? p.y - rs[1] : 0.0, ND > 2 ? p.z - rs[2] : 0.0};
            const double invdrmag = fast_rsqrt(dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2]);
            const double paux = p.w*invdrmag*invdrmag*invdrmag;
            for (int k = 0; k < ND; k++) a[k] += paux*dr[k];
            gp += p.w*invdrmag;
          }
          else {
            const CellCom c = d.ccom[n];
            if (!QUAD) {                                       // NeighbourSearch.h:350-377
              double dr[3] = {c.com[0] - rs[0], ND > 1 ? c.com[1] - rs[1] : 0.0, ND > 2 ? c.com[2] - rs[2] : 0.0};
              const double invdrmag = fast_rsqrt(dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2] + GH_SMALL);
              const double m3 = c.m*invdrmag*invdrmag*invdrmag;
              gp += c.m*invdrmag;
              for (int k = 0; k < ND; k++) a[k] += m3*dr[k];
            }
            else {                                             // NeighbourSearch.h:384-475
              const CellQuad cq = d.cquad[n];
              double dr[3] = {rs[0] - c.com[0], ND > 1 ? rs[1] - c.com[1] : 0.0, ND > 2 ? rs[2] - c.com[2] : 0.0};
              const double invdrmag = fast_rsqrt(dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2] + GH_SMALL);
              const double invdrsqd = invdrmag*invdrmag, invdr5 = invdrsqd*invdrsqd*invdrmag;
              const double Q5 = -(cq.q[0] + cq.q[2]);
              const double qscalar = cq.q[0]*dr[0]*dr[0] + cq.q[2]*dr[1]*dr[1] + Q5*dr[2]*dr[2] +
                                     2.0*(cq.q[1]*dr[0]*dr[1] + cq.q[3]*dr[0]*dr[2] + cq.q[4]*dr[1]*dr[2]);
              const double qfactor = 2.5*qscalar*invdr5*invdrsqd;
              const double m3 = c.m*invdrsqd*invdrmag;
              a[0] += (cq.q[0]*dr[0] + cq.q[1]*dr[1] + cq.q[3]*dr[2])*invdr5 - qfactor*dr[0] - m3*dr[0];
              if (ND > 1) a[1] += (cq.q[1]*dr[0] + cq.q[2]*dr[1] + cq.q[4]*dr[2])*invdr5 - qfactor*dr[1] - m3*dr[1];
              if (ND > 2) a[2] += (cq.q[3]*dr[0] + cq.q[4]*dr[1] + Q5*dr[2])*invdr5 - qfactor*dr[2] - m3*dr[2];
              gp += c.m*invdrmag + 0.5*qscalar*invdr5;
            }
          }
        }
        else if (g.N > 0) {                                    // too close for the cell: open, or direct particles (:853-873)
          if (!isleaf) open = true;
          else for (int t = 0; t < g.N; t++) {
            const double4 p = d.posm[g.first + t];
            double dr[3] = {p.x - rs[0], ND > 1 ? p.y - rs[1] : 0.0, ND > 2 ? p.z - rs[2] : 0.0};
            const double r2 = dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2];
            const double invdrmag = r2 > 0.0 ? fast_rsqrt(r2) : 0.0;
            const double paux = p.w*invdrmag*invdrmag*invdrmag;
            for (int k = 0; k < ND; k++) a[k] += paux*dr[k];
            gp += p.w*invdrmag;
          }
        }
      }
      const unsigned long long om = __ballot(open);
      if (open) {
        const int pos = nnext + 2*__popcll(om & lt);
        if (pos + 1 < GH_STAR_FRONT) { s_front[cur ^ 1][pos] = 2*n + 1; s_front[cur ^ 1][pos + 1] = 2*n + 2; }
      }
      nnext += 2*__popcll(om);
    }
    if (nnext > GH_STAR_FRONT) { if (lane == 0) atomicOr(flags, FLAG_FRONTIER_OVERFLOW); nnext = GH_STAR_FRONT; }
    __syncthreads();
    nf = nnext; cur ^= 1;
  }
  for (int k = 0; k < ND; k++) { const double v = star_wave_sum(a[k]); if (lane == 0) out_a[(size_t) si*ND + k] = v; }
  const double g = star_wave_sum(gp);
  if (lane == 0) out_gpot[si] = g;
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static StarTab star_tab(gh_ctx *ctx)
{
  StarTab S;
  S.posm = ctx->star_posm; S.h = ctx->star_h; S.n = ctx->nstars; S.softening = ctx->star_softening;
  return S;
}

extern "C" int gh_set_stars(gh_ctx *ctx, int64_t nstars, const double *r, const double *m, const double *h, int nbody_softening)
{
  if (!ctx || nstars < 0 || (nstars > 0 && (!r || !m || !h))) return GH_ERR_INVALID;
  if (nstars > ctx->star_cap) {
    if (ctx->star_posm) (void) hipFree(ctx->star_posm);
    if (ctx->star_h) (void) hipFree(ctx->star_h);
    if (ctx->star_out) (void) hipFree(ctx->star_out);
    ctx->star_posm = nullptr; ctx->star_h = nullptr; ctx->star_out = nullptr; ctx->star_cap = 0; ctx->nstars = 0;
    GH_CHECK(ctx, hipMalloc((void**) &ctx->star_posm, sizeof(double4)*(size_t) nstars));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->star_h, sizeof(double)*(size_t) nstars));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->star_out, sizeof(double)*4*(size_t) nstars));
    ctx->star_cap = nstars;
  }
  ctx->nstars = (int) nstars; ctx->star_softening = nbody_softening;
  if (nstars == 0) return GH_OK;
  std::vector<double4> pm((size_t) nstars);
  const int nd = ctx->ndim;
  for (int64_t i = 0; i < nstars; i++) {
    pm[i].x = r[i*nd]; pm[i].y = nd > 1 ? r[i*nd + 1] : 0.0; pm[i].z = nd > 2 ? r[i*nd + 2] : 0.0; pm[i].w = m[i];
  }
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  GH_CHECK(ctx, hipMemcpy(ctx->star_posm, pm.data(), sizeof(double4)*(size_t) nstars, hipMemcpyHostToDevice));
  GH_CHECK(ctx, hipMemcpy(ctx->star_h, h, sizeof(double)*(size_t) nstars, hipMemcpyHostToDevice));
  return GH_OK;
}

int gh_zeta_stars_impl(gh_ctx *ctx)
{
  if (ctx->nstars <= 0) return GH_OK;
  DevicePtrs d = gh_dev_own(ctx);
  const StarTab S = star_tab(ctx);
#define LAUNCH(ND_, KT_) hipLaunchKernelGGL((k_zeta_stars<ND_, KT_>), dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, d, S, ctx->ktab);
  GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  return GH_OK;
}

int gh_gas_star_forces_impl(gh_ctx *ctx)
{
  if (ctx->nstars <= 0) return GH_OK;
  DevicePtrs d = gh_dev_own(ctx);
  const StarTab S = star_tab(ctx);
#define LAUNCH(ND_, KT_) hipLaunchKernelGGL((k_gas_star_forces<ND_, KT_>), dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, d, S, ctx->ktab);
  GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  return GH_OK;
}

extern "C" int gh_star_gas_forces(gh_ctx *ctx, double *a, double *gpot)
{
  if (!ctx || !a || !gpot) return GH_ERR_INVALID;
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_INVALID, "gh_star_gas_forces: no tree");
  if (ctx->nstars <= 0) return GH_OK;
  DevicePtrs d = gh_dev(ctx);
  const StarTab S = star_tab(ctx);
  const bool quad = ctx->cfg.multipole == GH_MULTIPOLE_QUADRUPOLE || ctx->cfg.multipole == GH_MULTIPOLE_FAST_QUADRUPOLE;
  double *oa = ctx->star_out, *og = ctx->star_out + (size_t) 3*ctx->nstars;
#define LAUNCH(ND_, KT_)                                                                                                         \
  if (quad) hipLaunchKernelGGL((k_star_gas_forces<ND_, KT_, true>), dim3(ctx->nstars), dim3(64), 0, ctx->stream, d, S, ctx->ktab, oa, og, ctx->d_flags, ctx->L, ctx->rank); \
  else hipLaunchKernelGGL((k_star_gas_forces<ND_, KT_, false>), dim3(ctx->nstars), dim3(64), 0, ctx->stream, d, S, ctx->ktab, oa, og, ctx->d_flags, ctx->L, ctx->rank);
  GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  int rc = gh_sync_collect(ctx, "gh_star_gas_forces");
  if (rc) return rc;
  if (ctx->nranks > 1) {
    // the ranks' partial sums, added in rank order on every rank alike
    const size_t ns = (size_t) ctx->nstars, nd = (size_t) ctx->ndim;
    std::vector<char> all; std::vector<size_t> sizes;
    if ((rc = gh_dd_gatherv(ctx, ctx->star_out, sizeof(double)*4*ns, all, sizes))) return rc;
    for (size_t i = 0; i < ns*nd; i++) a[i] = 0.0;
    for (size_t i = 0; i < ns; i++) gpot[i] = 0.0;
    for (int r = 0; r < ctx->nranks; r++) {
      const double *blk = (const double*) all.data() + (size_t) r*4*ns;
      for (size_t i = 0; i < ns*nd; i++) a[i] += blk[i];
      for (size_t i = 0; i < ns; i++) gpot[i] += blk[3*ns + i];
    }
    return GH_OK;
  }
  GH_CHECK(ctx, hipMemcpy(a, oa, sizeof(double)*(size_t) ctx->ndim*ctx->nstars, hipMemcpyDeviceToHost));
  GH_CHECK(ctx, hipMemcpy(gpot, og, sizeof(double)*(size_t) ctx->nstars, hipMemcpyDeviceToHost));
  return GH_OK;
}
