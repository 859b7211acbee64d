// density.hip -- grad-h SPH density / smoothing-length pass on the GPU.
//
// Replaces GradhSphTree::UpdateAllSphProperties + GradhSph::ComputeH + ComputeThermalProperties
// (reference src/GradhSph/GradhSphTree.cpp:83-271, src/GradhSph/GradhSph.cpp:142-347).
//
// Mapping: one wavefront per group (<= 64 particles of one small KD subtree), one lane per target
// particle.  For every h-iteration the wave walks the tree for the whole group and streams the candidate
// particles through a 64-entry SoA tile in LDS:
//   phase 1: every lane tests all 64 tile entries (LDS broadcast reads) and builds a 64-bit mask of
//            the entries inside its kernel support;
//   phase 2: every lane walks its own mask and evaluates the three kernel sums only for those.
// The h fixed-point iteration of ComputeH runs per lane with the reference's state machine (30 fixed-point
// steps, then bisection, GradhSph.cpp:184-257).  Every iteration re-walks the tree (streaming depth-first
// walk, walk.hpp) with a search volume that covers kernrange*h of every lane still iterating, so there
// is no candidate list to overflow: where the reference returns 0 ("h grew beyond the candidate list",
// :255) and redoes the whole cell with hmax*1.05, the wave simply continues - the iterates are the same
// numbers because every iterate is summed over a complete list.
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"
#include <cstdlib>

struct DensityParams {
  Domain dom;
  EosParams eos;
  double h_fac, h_converge;
  double rho_sink;    // sink runs: h >= h_fac (m / rho_sink)^(1/ndim) (GradhSph.cpp:163-169); 0 = no sinks
  const double *ktab; // kernel tables (tabulated_kernel = 1) or nullptr
  const int *only_if; // fused kernel as the fallback of the split path: per-group flags, process a group only if
  int only_val;       // only_if[group] == only_val (nullptr: all groups).  Flag values: 1 = list overflow / leaf retry,
  int *fbout;         // 2 = multi-GPU: the walk reached a cell the halo exchange did not import - nothing is stored for
  unsigned int *miss_count;   // the group, it is redone after a wider import (gh_density_impl); fbout = the flag array
#ifdef GH_DEBUG_BLOCKTIME
  double *dbg;        // [ngroups][8] per-group timing / work record (profiling builds only)
#endif
  int group0;        // first group of this rank's shard
};

#define DB 4      /* candidate tiles per phase-2 batch */
#define GH_DENS_NQ 4        /* quarter-groups per particle group (<= 16 targets each): the unit of the evaluation kernel */
#define GH_DENS_QSHIFT 27   /* range entries: count in the low 27 bits of .y, quarter mask above */
// quarter-group (0 .. nq-1) of particle i of group node gnode: nq = min(GH_DENS_NQ, leaves per group) subtrees
__device__ __forceinline__ int dens_quarter(const DevicePtrs &d, int gnode, int i)
{
  const int nleaf = 1 << (d.ltot - d.lgroup);
  int n = gnode, qq = 0;
  for (int nq = min(GH_DENS_NQ, nleaf); nq > 1; nq >>= 1) {
    const int c2 = 2*n + 2;
    const bool right = i >= d.cfirst[c2];
    n = right ? c2 : 2*n + 1;
    qq = 2*qq + (right ? 1 : 0);
  }
  return qq;
}
typedef float float2_t __attribute__((ext_vector_type(2)));

// x^(1/ND), the reference's pow(x, invndim) (Sph.h:259): ND = 1, 2 exactly; ND = 3 through cbrt, which agrees with pow(x, 1.0/3.0)
// to an ulp or two (h is compared at 1e-12) and needs a third of pow's registers - the kernel spilled around this call
template <int ND> __device__ __forceinline__ double root_nd(double x)
{
  return ND == 1 ? x : (ND == 2 ? sqrt(x) : cbrt(x));
}

// normalise and store the converged sums of one particle (GradhSph.cpp:262-317)
template <int ND, class K>
__device__ __forceinline__ void density_store(const DevicePtrs &d, const DensityParams &P, int i, double mi, double ui,
                                              double rho, double omg, double zet, double hlo, double invhsqd_last, double hmaxl)
{
  const double invndim = 1.0/(double) ND;
  const double h = fmax(P.h_fac*root_nd<ND>(mi/rho), hlo);
  const double invh1 = 1.0/h;
  const double hfac1 = powN<ND>(invh1)*invh1;
  const double deriv = -invndim*h/rho;                                     // h_rho_deriv, Sph.h:264
  double invomega = 1.0 - deriv*omg;
  invomega = 1.0/invomega;
  double zeta = deriv*zet*invomega;
  if (d.sinks && d.f[D_SINKID][i] != -1.0) { invomega = 1.0; zeta = 0.0; }   // inside a sink, GradhSph.cpp:309-312
  if (d.pm_invhsqd) { d.pm_invhsqd[i] = invhsqd_last; d.pm_cullsqd[i] = K::kernrangesqd*hmaxl*hmaxl; }
  double sound, press;
  eos_eval(P.eos, rho, ui, sound, press);
  d.f[D_H][i] = h;
  d.f[D_RHO][i] = rho;
  d.f[D_INVOMEGA][i] = invomega;
  d.f[D_ZETA][i] = zeta;
  d.f[D_HFACTOR][i] = hfac1;
  d.f[D_HRANGESQD][i] = K::kernrangesqd*h*h;
  d.f[D_DIV_V][i] = 0.0;
  d.f[D_U][i] = ui;
  d.f[D_SOUND][i] = sound;
  d.f[D_PRESSURE][i] = press;
}


// STALE: the tree has been extrapolated (ntreestockstep > 1, Tree.cpp:172-198) since its last stocking - the search is
// then the reference's per-leaf-cell one against the drifted boxes (walk_dfs_stream_masked), including its list filter
// |r_j - rcell|^2 < (rmax + kernrange*hmax)^2 with the drifted centre (Tree.cpp:369-376), so that the neighbours the
// reference loses are lost here too.
template <int ND, bool COUNT, int KT, bool STALE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 8))) void k_density(DevicePtrs d, DensityParams P, unsigned long long *stats, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ WalkLDS<int> L;
  __shared__ unsigned short s_smask[STALE ? GH_SCAP : 1], s_tagm[STALE ? DB*64 : 1];   // leaf masks of stack entries / tile slots
  __shared__ double s_lrc[STALE ? 16 : 1][3], s_lrm[STALE ? 16 : 1];                   // drifted leaf centres, leaf rmax
  __shared__ double s_x[DB*64], s_y[DB*64], s_z[DB*64], s_m[DB*64];   // a batch of DB candidate tiles
  __shared__ unsigned long long s_mask[DB][64];                       // per tile, per lane: entries in support
  __shared__ __attribute__((aligned(8))) float s_fx[64], s_fy[64], s_fz[64];  // current tile, fp32, relative to the group centre
  __shared__ double s_lb[16][6], s_lhs[16], s_hl[64];                 // leaf boxes, per-leaf search h, lanes' h

  const int lane = threadIdx.x;
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  if (P.only_if) {                                       // fallback launch: quarter-groups the split path flagged, nothing else
    bool any = false;
    for (int k = 0; k < GH_DENS_NQ; k++) any = any || P.only_if[GH_DENS_NQ*q + k] == P.only_val;
    if (!any) return;
  }
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (gN == 0) return;
#ifdef GH_DEBUG_BLOCKTIME
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_tiles = 0, dbg_passes = 0;
#endif
  // block timesteps: only the active particles are targets (GradhSphTree.cpp:128-131)
  bool act = lane < gN && (!d.levels || ((int) d.f[D_FLAGS][gfirst + lane] & 1));
  // (the evaluation kernel has stored the other quarter-groups' results already: their particles are not targets here -
  // the reference's unit of a retry is the leaf cell, and a quarter-group is a set of whole leaf cells)
  if (P.only_if && act) act = P.only_if[GH_DENS_NQ*q + dens_quarter(d, gnode, gfirst + lane)] == P.only_val;
  if (STALE && d.leafact) {
    // ... of the cells the reference still has on its active list (cell.Nactive of the last stocking, see k_leaf_nactive)
    const int ip = gfirst + (lane < gN ? lane : 0);
    int ln = gnode;
    while (ln < d.gtot - 1) { const int c2 = 2*ln + 2; ln = (ip >= d.cfirst[c2]) ? c2 : 2*ln + 1; }
    act = act && d.leafact[ln - (d.gtot - 1)] > 0;
  }
  const int i = gfirst + (act ? lane : 0);

  const double invndim = 1.0/(double) ND;
  double ri[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < ND; k++) ri[k] = d.f[D_RX + k][i];
  const double mi = d.f[D_M][i];
  double ui = d.f[D_U][i];
  const CellBox gb = d.cbox[gnode];
  double gc[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < ND; k++) gc[k] = 0.5*(gb.bbmin[k] + gb.bbmax[k]);
  float tf[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < ND; k++) tf[k] = (float) (ri[k] - gc[k]);

  // The reference runs ComputeH per LEAF cell with hmax = 1.05^k * cell.hmax: a particle whose iterate exceeds
  // hmax makes ComputeH return 0 and the whole cell is redone from the stored h with the next k
  // (GradhSphTree.cpp:141-226, GradhSph.cpp:255).  hmax is also the upper bound of the bisection fallback, so
  // the try index matters for particles that need more than 30 iterations: keep it per leaf.
  int leafn = gnode;
  while (leafn < d.gtot - 1) { const int c2 = 2*leafn + 2; leafn = (i >= d.cfirst[c2]) ? c2 : 2*leafn + 1; }
  unsigned long long leafmates = 0ull;                 // lanes of the same leaf cell
  for (int l = 0; l < (1 << (d.ltot - d.lgroup)); l++) {
    const int ln = (d.gtot - 1) + (gnode - ((1 << d.lgroup) - 1))*(1 << (d.ltot - d.lgroup)) + l;
    const unsigned long long m = __ballot(act && leafn == ln);
    if (leafn == ln) leafmates = m;
  }
  double hmaxl = 1.05*d.ch[leafn].hmax;
  const int nleaf = 1 << (d.ltot - d.lgroup);
  const int leafnode0 = (d.gtot - 1) + (gnode - ((1 << d.lgroup) - 1))*nleaf;
  if (lane < nleaf) {
    const CellBox lbx = d.cbox[leafnode0 + lane];
    for (int k = 0; k < 3; k++) { s_lb[lane][k] = lbx.bbmin[k]; s_lb[lane][3 + k] = lbx.bbmax[k]; }
    if (STALE) {
      const CellGeo lg = d.cgeo[leafnode0 + lane];
      for (int k = 0; k < 3; k++) s_lrc[lane][k] = lg.rcell[k];
      s_lrm[lane] = lg.rmax;
    }
  }
  const int li = leafn - leafnode0;                      // the lane's leaf within the group

  // per-lane iteration state (GradhSph.cpp:148-158)
  const double h0 = d.f[D_H][i];
  // sink runs: lower bound of h; a cell whose hmax is below it leaves the particle as it is ("return -1", GradhSph.cpp:163-169)
  const double hfloor = P.rho_sink > 0.0 ? P.h_fac*pow(mi/P.rho_sink, invndim) : 0.0;
  double h = h0, hlo = hfloor, hup = hmaxl;
  int iter = 0;
  bool done = !act || hmaxl < hfloor;
  double rho = 0.0, omg = 0.0, zet = 0.0;
  double invh = 0.0, hfactor = 0.0, invhsqd = 0.0;
  unsigned long long n_iter = 0, n_cand = 0, n_retry = 0, n_tested = 0;
  // reference cull radius of the first try, for the candidate statistic: kernrange*1.05*hmax(leaf)
  const double cullsqd = K::kernrangesqd*hmaxl*hmaxl;

  bool miss = false;
  // ---- h iteration (GradhSph.cpp:184-257); every pass re-walks the tree with the radius it needs
  for (;;) {
    const bool running = !done;
    if (!__any(running)) break;
    // the search volume covers kernrange*h of every lane still iterating (Tree.cpp:319-328 uses
    // bb +/- kernrange*hmax; where the reference's hmax is too small it retries with hmax*1.05)
    const double hs = wave_max(running ? (STALE ? hmaxl : h) : 0.0);
    double lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
      lo[k] = k < ND ? gb.bbmin[k] - K::kernrange*hs : -1e300;
      hi[k] = k < ND ? gb.bbmax[k] + K::kernrange*hs : 1e300;
    }
    if (running) {
      iter++; n_iter++;
      invh = 1.0/h;
      hfactor = powN<ND>(invh);
      invhsqd = invh*invh;
      rho = 0.0; omg = 0.0; zet = 0.0;
    }
    // Elongated groups (thin KD cells of a sparse halo that reach into a dense region) would stream every
    // particle near their bounding box with the largest h of the group: there, cells are also culled against
    // each leaf box with that leaf's own search radius (wave-uniform switch, below).
    s_hl[lane] = running ? (STALE ? hmaxl : h) : 0.0;      // STALE: the reference's search radius is that of the try, kernrange*hmax
    __syncthreads();
    if (lane < nleaf) {
      const int ln = leafnode0 + lane;
      double hm = 0.0;
      for (int j = d.cfirst[ln] - gfirst; j < d.cfirst[ln] - gfirst + d.cN[ln]; j++) hm = fmax(hm, s_hl[j]);
      s_lhs[lane] = hm;
    }
    __syncthreads();
    bool leafcull = false;
    {
      // switch it on when the search radii differ a lot inside the group or the group box is long compared with
      // the smallest search radius - then the group-level box test lets through far more than the leaves need
      const double hl = lane < nleaf ? s_lhs[lane] : 0.0;
      const double hmx = wave_max(hl), hmn = wave_min(hl > 0.0 ? hl : 1e300);
      double extmax = 0.0;
      for (int k = 0; k < ND; k++) extmax = fmax(extmax, gb.bbmax[k] - gb.bbmin[k]);
      leafcull = hmx > 1.3*hmn || extmax > 4.0*K::kernrange*hmn;
    }
    const unsigned int codes = image_codes(P.dom, ND, lo, hi);
    const double rs2cut = (K::kernrange*hs)*(K::kernrange*hs)*(1.0 + 1e-12);
    auto cls = [&](int n, int code, bool &open, bool &emit, int &first, int &cnt) {
      const CellBox b = d.cbox[n];
      const int cn = b.N;
      if (cn < 0) miss = true;                             // multi-GPU: a remote cell the halo exchange did not import
      if (cn <= 0) return;
      double sg[3], sh[3];
      code_xform(P.dom, code, sg, sh);
      bool inside = true;
      double gap2 = 0.0;                                  // squared distance between the two boxes
      for (int k = 0; k < ND; k++) {
        double bmin, bmax;
        image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin, bmax);
        if (lo[k] > bmax) return;                         // BoxOverlap, InlineFuncs.h:362-390 (inclusive)
        if (bmin > hi[k]) return;
        if (bmin < lo[k] || bmax > hi[k]) inside = false;
        const double g1 = bmin - gb.bbmax[k], g2 = gb.bbmin[k] - bmax;
        const double gk = fmax(fmax(g1, g2), 0.0);
        gap2 += gk*gk;
      }
      // no particle of the cell can be within kernrange*hs of any particle of the group: the reference
      // would list it (box test) and sum zeros for it; skipping it changes nothing
      if (gap2 > rs2cut) return;
      if (leafcull) {
        bool any = false;
        for (int l = 0; l < nleaf && !any; l++) {
          const double hl = s_lhs[l];
          if (hl > 0.0) {
            double g2 = 0.0;
            for (int k = 0; k < ND; k++) {
              double bmin, bmax;
              image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin, bmax);
              const double gk = fmax(fmax(bmin - s_lb[l][3 + k], s_lb[l][k] - bmax), 0.0);
              g2 += gk*gk;
            }
            const double rl = K::kernrange*hl;
            any = g2 <= rl*rl*(1.0 + 1e-12);
          }
        }
        if (!any) return;
        inside = false;        // whole-subtree emission is a group-box shortcut: descend to the leaves instead
      }
      if (inside || n >= d.gtot - 1) { emit = true; first = b.first; cnt = cn; }
      else open = true;
    };
    // fp32 cull threshold: a superset of {invhsqd*r2 < kernrangesqd}.  Coordinates relative to the group
    // centre are exact to 6e-8*Rmax in fp32, so distances are off by < 1e-6*Rmax + 1e-6*d; phase 2
    // re-evaluates every survivor in fp64, and W(s >= 2) = 0 exactly, so extra survivors add zeros.
    double Rmax = 0.0;
    for (int k = 0; k < ND; k++) Rmax = fmax(Rmax, 0.5*(gb.bbmax[k] - gb.bbmin[k]) + K::kernrange*hs);
    const float thr = running ? (float) ((K::kernrange*h*(1.0 + 1e-6) + 2e-6*Rmax)*(K::kernrange*h*(1.0 + 1e-6) + 2e-6*Rmax)*(1.0 + 1e-6)) : -1.0f;
    int nb = 0;                                          // tiles in the current batch
    auto process_batch = [&]() {
      // phase 2: every lane walks its own masks over the whole batch (lane balance: a lane's in-support
      // entries per batch vary far less than per tile)
      int b = 0;
      unsigned long long mask = nb > 0 ? s_mask[0][lane] : 0ull;
      for (;;) {
        while (mask == 0ull && b + 1 < nb) mask = s_mask[++b][lane];
        if (!__any(mask != 0ull)) break;
        if (mask != 0ull) {
          const int c = b*64 + __ffsll((long long) mask) - 1;
          mask &= mask - 1ull;
          double dr[3] = {0.0, 0.0, 0.0};
          dr[0] = s_x[c] - ri[0];
          if (ND > 1) dr[1] = s_y[c] - ri[1];
          if (ND > 2) dr[2] = s_z[c] - ri[2];
          double mj = s_m[c];
          double r2 = dr[0]*dr[0];
          if (ND > 1) r2 += dr[1]*dr[1];
          if (ND > 2) r2 += dr[2]*dr[2];
          if (STALE) {
            // the candidate is on this particle's list only if the walk of ITS leaf reached the candidate's leaf and the
            // reference's list filter keeps it: |r_j - rcell|^2 < (rmax + kernrange*hmax)^2   (Tree.cpp:369-376)
            double dc = s_x[c] - s_lrc[li][0], d2c = dc*dc;
            if (ND > 1) { dc = s_y[c] - s_lrc[li][1]; d2c += dc*dc; }
            if (ND > 2) { dc = s_z[c] - s_lrc[li][2]; d2c += dc*dc; }
            const double hrm = s_lrm[li] + K::kernrange*hmaxl;
            if (!((s_tagm[c] >> li) & 1) || !(d2c < hrm*hrm)) mj = 0.0;
          }
          {
#pragma clang fp contract(fast)
            const double s2 = invhsqd*r2;                              // w0_s2(ssqd) etc., GradhSph.cpp:200-203
            double kw0, kwom, kwz;
            K::t_dens3(s2, P.ktab, kw0, kwom, kwz);
            rho += mj*kw0;
            omg += mj*invh*kwom;
            zet += mj*kwz;
          }
        }
      }
      __syncthreads();
      nb = 0;
    };
    auto tile = [&](bool valid, int j, int code) {
      const int o = nb*64;
      {
        double x = 1e30, y = 1e30, z = 1e30, m = 0.0;
        if (valid) {
          const double4 v = d.posm[j];
          double sg[3], sh[3];
          code_xform(P.dom, STALE ? (code & 31) : code, sg, sh);
          x = sg[0]*v.x + sh[0]; y = sg[1]*v.y + sh[1]; z = sg[2]*v.z + sh[2]; m = v.w;
        }
        s_x[o + lane] = x; s_y[o + lane] = y; s_z[o + lane] = z; s_m[o + lane] = m;
        if (STALE) s_tagm[o + lane] = (unsigned short) (valid ? (code >> 5) : 0);
        s_fx[lane] = (float) (x - gc[0]); s_fy[lane] = ND > 1 ? (float) (y - gc[1]) : 0.f; s_fz[lane] = ND > 2 ? (float) (z - gc[2]) : 0.f;
      }
      __syncthreads();
      // phase 1: support mask, packed fp32 (two candidates per instruction)
      unsigned int mlo = 0, mhi = 0;
      {
        const float2_t tx = {tf[0], tf[0]}, ty = {tf[1], tf[1]}, tz = {tf[2], tf[2]};
        const float2_t *fx = (const float2_t*) s_fx, *fy = (const float2_t*) s_fy, *fz = (const float2_t*) s_fz;
#pragma unroll
        for (int c2 = 0; c2 < 32; c2++) {
          const float2_t dx = fx[c2] - tx;
          float2_t r2 = dx*dx;
          if (ND > 1) { const float2_t dy = fy[c2] - ty; r2 = dy*dy + r2; }
          if (ND > 2) { const float2_t dz = fz[c2] - tz; r2 = dz*dz + r2; }
          const int c = 2*c2;
          if (c < 32) { mlo |= (r2.x < thr) ? (1u << c) : 0u; mlo |= (r2.y < thr) ? (2u << c) : 0u; }
          else { mhi |= (r2.x < thr) ? (1u << (c - 32)) : 0u; mhi |= (r2.y < thr) ? (2u << (c - 32)) : 0u; }
        }
      }
      s_mask[nb][lane] = (unsigned long long) mlo | ((unsigned long long) mhi << 32);
      if (COUNT) {
        if (running) {
          for (int c = 0; c < 64; c++) {
            double r2 = 0.0;
            { const double dx = s_x[o + c] - ri[0]; r2 = dx*dx; }
            if (ND > 1) { const double dy = s_y[o + c] - ri[1]; r2 += dy*dy; }
            if (ND > 2) { const double dz = s_z[o + c] - ri[2]; r2 += dz*dz; }
            if (r2 + GH_SMALL <= cullsqd) n_cand++;
          }
          n_tested += 64;
        }
      }
#ifdef GH_DEBUG_BLOCKTIME
      dbg_tiles++;
#endif
      nb++;
      if (nb == DB) process_batch();
      else __syncthreads();
    };
#ifdef GH_DEBUG_BLOCKTIME
    dbg_passes++;
#endif
    if (STALE) {
      // per (node, leaf): BoxOverlap(leaf.bb -/+ kernrange*hmax, node.bb), inclusive (Tree.cpp:319-328, InlineFuncs.h:362-390)
      auto clsm = [&](int n, int code, unsigned int inmask, int &first, int &cnt) -> unsigned int {
        const CellBox b = d.cbox[n];
        first = b.first; cnt = b.N;
        if (b.N <= 0) return 0u;
        double sg[3], sh[3];
        code_xform(P.dom, code, sg, sh);
        double bmin[3] = {0.0, 0.0, 0.0}, bmax[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < ND; k++) image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin[k], bmax[k]);
        unsigned int om = 0;
        for (int l = 0; l < nleaf; l++) {
          const double hl = s_lhs[l];
          if (!((inmask >> l) & 1) || !(hl > 0.0)) continue;
          bool ov = true;
          for (int k = 0; k < ND; k++) {
            if (s_lb[l][k] - K::kernrange*hl > bmax[k]) ov = false;
            if (bmin[k] > s_lb[l][3 + k] + K::kernrange*hl) ov = false;
          }
          if (ov) om |= 1u << l;
        }
        return om;
      };
      unsigned int m0 = 0;
      for (int l = 0; l < nleaf; l++) if (s_lhs[l] > 0.0) m0 |= 1u << l;
      walk_dfs_stream_masked(d, L, s_smask, codes, m0, clsm, tile, flags);
    }
    else walk_dfs_stream(d, L, codes, cls, tile, flags);
    process_batch();
    if (__any(miss)) {                                     // incomplete sums: store nothing, the host widens the import
      if (lane == 0) atomicAdd(P.miss_count, 1u);
      // (a fallback launch keeps the flags of quarter-groups it was not asked to do)
      if (lane < GH_DENS_NQ && (!P.only_if || P.only_if[GH_DENS_NQ*q + lane] == P.only_val)) P.fbout[GH_DENS_NQ*q + lane] = 2;
      return;
    }

    bool failed = false;
    if (running) {
      rho *= hfactor; omg *= hfactor; zet *= invhsqd;
      const double hnew = P.h_fac*root_nd<ND>(mi/rho);                     // h_rho_func, Sph.h:259
      if (rho > 0.0 && h > hlo && fabs(h - hnew)*invh < P.h_converge) done = true;
      else {
        if (iter < 30) h = hnew;
        else if (iter == 30) h = 0.5*(hlo + hup);
        else if (iter < 150) {
          if (rho < GH_SMALL || h > hnew) hup = h; else hlo = h;
          h = 0.5*(hlo + hup);
        }
        else { atomicOr(flags, FLAG_H_NOT_CONVERGED); done = true; }
        if (!done) {
          if (!isfinite(h)) { atomicOr(flags, FLAG_H_NOT_CONVERGED); done = true; }
          else if (h > hmaxl) failed = true;                                 // "return 0", :255
          else if (!(h > hlo && h < hup)) done = true;                       // loop exit, :257
        }
      }
    }
    // a failed particle restarts its whole leaf cell with hmax*1.05, from the stored h (GradhSphTree.cpp:172-226)
    const unsigned long long fm = __ballot(failed);
    if (fm != 0ull && act && (leafmates & fm) != 0ull) {
      hmaxl = 1.05*hmaxl;
      h = h0; hlo = hfloor; hup = hmaxl; iter = 0; done = hmaxl < hfloor;
      if (failed) n_retry++;
    }
  }

#ifdef GH_DEBUG_BLOCKTIME
  if (lane == 0 && P.dbg) {
    double *o = P.dbg + (size_t) q*8;
    o[0] = (double) (wall_clock64() - dbg_t0); o[1] = (double) dbg_tiles; o[2] = (double) dbg_passes; o[3] = (double) gN;
    for (int k = 0; k < 3; k++) o[4 + k] = gb.bbmax[k] - gb.bbmin[k];
    o[7] = d.ch[gnode].hmax;
  }
#endif
  if (P.only_if && lane < GH_DENS_NQ && P.only_if[GH_DENS_NQ*q + lane] == P.only_val) P.fbout[GH_DENS_NQ*q + lane] = 0;
  // ---- normalise and store (GradhSph.cpp:262-317)
  if (act && !(hmaxl < hfloor)) density_store<ND, K>(d, P, i, mi, ui, rho, omg, zet, hlo, invhsqd, hmaxl);
  if (COUNT) {
    const unsigned long long a = wave_sum_u64(act ? n_iter : 0), b = wave_sum_u64(act ? n_cand : 0), c = wave_sum_u64(n_retry);
    const unsigned long long t = wave_sum_u64(act ? n_tested : 0);
    if (lane == 0) { atomicAdd(&stats[ST_ITER], a); atomicAdd(&stats[ST_CAND], b); atomicAdd(&stats[ST_RETRY], c); atomicAdd(&stats[ST_PAIRS], t); }
  }
}


// ================================================================================================
// Split path: walk once, evaluate from a candidate list in HBM.
//
// The fused kernel above is bound by the latency of its serial tree walk (one dependent cell load per pop
// step, at the 2 waves / SIMD its 256 registers allow).  The split path separates the two jobs:
//
//   k_dens_walk : one wavefront per group, nothing but the walk - a stack in LDS, ~50 registers, 8 waves / SIMD,
//                 so that the walks of 32 groups per CU overlap their latencies.  The search volume is the
//                 reference's own: every leaf cell of the group searches with kernrange * 1.05 * hmax(leaf)
//                 (GradhSphTree.cpp:141-170; hmax is the bound ComputeH enforces on every iterate,
//                 GradhSph.cpp:255).  Candidate particle RANGES (a leaf, or a whole subtree inside the
//                 search box - contiguous in tree order) go to a per-group list in HBM.
//   k_dens_eval : one wavefront per group, lane = target particle, no walk: streams the listed ranges through
//                 the LDS tiles (packed-fp32 cull, fp64 sums over the survivors) once per h iteration - in
//                 steady state exactly once.
//
// A group whose list overflows, or whose h iteration makes the reference redo a leaf cell with hmax * 1.05
// (a larger search volume than the list covers), is flagged and done by the fused kernel afterwards
// (device-side decision, no host round trip); the evaluation kernel leaves such a group untouched.
// ================================================================================================
struct DensLists {
  int2 *rl;          // [ngroups][rcap]: (first | image code << GH_NODE_BITS, count | quarter mask << GH_DENS_QSHIFT)
  int *rlen;         // [ngroups]
  int *fb;           // [ngroups][GH_DENS_NQ]: 1 = the fused kernel redoes this quarter-group, 2 = ... after a wider halo import
  int rcap;
};

#ifndef GH_DWALK_WPE
#define GH_DWALK_WPE 6      /* 80 VGPRs: at 8 waves (64 VGPRs) the quarter-mask code spilled inside the walk loop (0.33 -> 0.31 ms) */
#endif
template <int ND, int KT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GH_DWALK_WPE, 8))) void k_dens_walk(DevicePtrs d, DensityParams P, DensLists G, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ int s_stack[GH_SCAP];
  __shared__ double s_lb[16][6], s_lhs[16];
  __shared__ double s_qb[GH_DENS_NQ][6], s_qh[GH_DENS_NQ];      // quarter-groups: box of their target leaves, search h

  const int lane = threadIdx.x;
  const unsigned long long lt = lanemask_lt();
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (lane == 0) G.rlen[q] = 0;
  if (lane < GH_DENS_NQ) G.fb[GH_DENS_NQ*q + lane] = 0;
  if (gN == 0) return;
  const int nleaf = 1 << (d.ltot - d.lgroup);
  const int nq = min(GH_DENS_NQ, nleaf), lpq = nleaf/nq;      // quarter-groups of this group, leaves per quarter
  const int leafnode0 = (d.gtot - 1) + q*nleaf;
  // per leaf: bounding box and search h = 1.05 * hmax(leaf); 0 for leaves without a target particle
  double hl = 0.0;
  if (lane < nleaf) {
    const int ln = leafnode0 + lane;
    const CellBox lbx = d.cbox[ln];
    for (int k = 0; k < 3; k++) { s_lb[lane][k] = lbx.bbmin[k]; s_lb[lane][3 + k] = lbx.bbmax[k]; }
    bool any = lbx.N > 0;
    if (any && d.levels) {                              // block timesteps: targets are the active particles only
      any = false;
      for (int t = 0; t < lbx.N; t++) any = any || ((int) d.f[D_FLAGS][lbx.first + t] & 1);
    }
    hl = any ? 1.05*d.ch[ln].hmax : 0.0;
    s_lhs[lane] = hl;
  }
  const double hs = wave_max(hl), hmn = wave_min(hl > 0.0 ? hl : 1e300);
  if (!(hs > 0.0)) return;
  __syncthreads();
  if (lane < nq) {
    double qh = 0.0, b[6] = {1e300, 1e300, 1e300, -1e300, -1e300, -1e300};
    for (int l = lane*lpq; l < (lane + 1)*lpq; l++) {
      if (s_lhs[l] > 0.0) {
        qh = fmax(qh, s_lhs[l]);
        for (int k = 0; k < 3; k++) { b[k] = fmin(b[k], s_lb[l][k]); b[3 + k] = fmax(b[3 + k], s_lb[l][3 + k]); }
      }
    }
    for (int k = 0; k < 6; k++) s_qb[lane][k] = b[k];
    s_qh[lane] = qh;
  }
  const CellBox gb = d.cbox[gnode];
  double lo[3], hi[3];
  double extmax = 0.0;
  for (int k = 0; k < 3; k++) {
    lo[k] = k < ND ? gb.bbmin[k] - K::kernrange*hs : -1e300;
    hi[k] = k < ND ? gb.bbmax[k] + K::kernrange*hs : 1e300;
    if (k < ND) extmax = fmax(extmax, gb.bbmax[k] - gb.bbmin[k]);
  }
  // leaf-level culling where the group-level box test would let through far more than the leaves need (see k_density)
  const bool leafcull = hs > 1.3*hmn || extmax > 4.0*K::kernrange*hmn;
  const unsigned int codes = image_codes(P.dom, ND, lo, hi);
  const double rs2cut = (K::kernrange*hs)*(K::kernrange*hs)*(1.0 + 1e-12);
  int top = 0;
  if (codes == 1u && d.lgroup >= 6) {
    // no periodic / mirror images: start from the 64 cells of level 6 instead of the root - the first pop classifies
    // them side by side instead of descending six levels one dependent cell load at a time
    s_stack[lane] = 63 + lane;
    top = 64;
  }
  else {
    for (int c = 0; c < 27; c++) {
      if (codes & (1u << c)) { if (lane == 0) s_stack[top] = 0 | (c << GH_NODE_BITS); top++; }
    }
  }
  __syncthreads();
  int nout = 0;
  bool overflow = false, miss = false;
  int2 *rl = G.rl + (size_t) q*G.rcap;
  while (top > 0) {
    const int p = pop_width(top);
    const int newtop = top - p;
    bool open = false, emit = false;
    int n = 0, code = 0, first = 0, cnt = 0;
    unsigned int qm = 0;
    if (lane < p) {
      const int e = s_stack[top - 1 - lane];
      n = e & GH_NODE_MASK; code = e >> GH_NODE_BITS;
      const CellBox b = d.cbox[n];
      if (b.N < 0) miss = true;                            // multi-GPU: a remote cell the halo exchange did not import
      if (b.N > 0) {
        double sg[3], sh[3];
        code_xform(P.dom, code, sg, sh);
        bool keep = true, inside = true;
        double gap2 = 0.0;                                 // squared distance between the two boxes
        double bmin[3], bmax[3];
        for (int k = 0; k < ND; k++) {
          image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin[k], bmax[k]);
          if (lo[k] > bmax[k] || bmin[k] > hi[k]) keep = false;           // BoxOverlap, InlineFuncs.h:362-390 (inclusive)
          if (bmin[k] < lo[k] || bmax[k] > hi[k]) inside = false;
          const double gk = fmax(fmax(bmin[k] - gb.bbmax[k], gb.bbmin[k] - bmax[k]), 0.0);
          gap2 += gk*gk;
        }
        // farther than kernrange*hs from every particle of the group: the reference lists it and sums zeros for it
        if (gap2 > rs2cut) keep = false;
        if (keep && leafcull) {
          bool any = false;
          for (int l = 0; l < nleaf && !any; l++) {
            const double hl_ = s_lhs[l];
            if (hl_ > 0.0) {
              double g2 = 0.0;
              for (int k = 0; k < ND; k++) {
                const double gk = fmax(fmax(bmin[k] - s_lb[l][3 + k], s_lb[l][k] - bmax[k]), 0.0);
                g2 += gk*gk;
              }
              const double rl_ = K::kernrange*hl_;
              any = g2 <= rl_*rl_*(1.0 + 1e-12);
            }
          }
          keep = any;
          inside = false;      // whole-subtree emission is a group-box shortcut: descend to the leaves instead
        }
        if (keep) {
          if (inside || n >= d.gtot - 1) {
            // which quarter-groups can have a particle of theirs within kernrange * h of this range?  (the evaluation
            // kernel runs per quarter and expands only the ranges that carry its bit)
            for (int qq = 0; qq < nq; qq++) {
              const double hq = s_qh[qq];
              if (hq > 0.0) {
                double g2 = 0.0;
                for (int k = 0; k < ND; k++) {
                  const double gk = fmax(fmax(bmin[k] - s_qb[qq][3 + k], s_qb[qq][k] - bmax[k]), 0.0);
                  g2 += gk*gk;
                }
                const double rq = K::kernrange*hq;
                if (g2 <= rq*rq*(1.0 + 1e-12)) qm |= 1u << qq;
              }
            }
            if (qm) { emit = true; first = b.first; cnt = b.N; }
          }
          else open = true;
        }
      }
    }
    const unsigned long long om = __ballot(open), em = __ballot(emit);
    __syncthreads();
    if (open) {
      const int pos = newtop + 2*__popcll(om & lt);
      if (pos + 1 < GH_SCAP) {
        s_stack[pos] = (2*n + 1) | (code << GH_NODE_BITS);
        s_stack[pos + 1] = (2*n + 2) | (code << GH_NODE_BITS);
      }
    }
    top = newtop + 2*__popcll(om);
    if (top > GH_SCAP) { overflow = true; top = GH_SCAP; }
    if (emit) {
      const int pos = nout + __popcll(em & lt);
      if (pos < G.rcap) rl[pos] = make_int2(first | (code << GH_NODE_BITS), cnt | (int) (qm << GH_DENS_QSHIFT));
    }
    nout += __popcll(em);
    __syncthreads();
  }
  if (nout > G.rcap) overflow = true;
  const bool anymiss = __any(miss);
  if (lane == 0) {
    G.rlen[q] = (overflow || anymiss) ? 0 : nout;
    if (anymiss) atomicAdd(P.miss_count, 1u);
  }
  if (lane < GH_DENS_NQ) G.fb[GH_DENS_NQ*q + lane] = anymiss ? 2 : (overflow ? 1 : 0);
}

#define GH_DENS_RCAP 512   /* candidate ranges per group held by the split path */
#ifndef GH_DENS_WPE
#define GH_DENS_WPE 4
#endif
#define GH_DENS_ICAP 1024   /* candidate slots of one quarter-group held in LDS as particle indices */
typedef float float4_t __attribute__((ext_vector_type(4)));

// Evaluation: one wavefront per QUARTER-group (<= 64/S target particles: 16 for the usual 16-leaf group, S = 4), S
// sub-lanes per target.  A 64-particle group culls ~1 000 candidate slots per particle to keep ~70 (its search box is 9
// interparticle spacings wide, a kernel 5); a quarter-group's box is 7.5 wide - ~420 slots, the ranges the walk tagged
// with the quarter's bit - and the S sub-lanes of a target share them: lane (t, s) tests the candidate pairs s, s + S, ...
// of a tile in packed fp32 and evaluates the survivors in fp64; the three sums are added over the sub-lanes (two
// xor-shuffles) at the end of an h iteration, after which all sub-lanes take the same decision.  Per 64 targets that is
// 2.4 x fewer cull tests, and the fp64 work of a target (whose neighbours come in runs of consecutive slots: leaves) is
// dealt round-robin to its sub-lanes.  S = 2 / 1: groups of 2 / 1 leaves (Nleafmax 16 / 32).
template <int ND, bool COUNT, int KT, int S>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GH_DENS_WPE, 8))) void k_dens_eval(DevicePtrs d, DensityParams P, DensLists G,
                                                                                       unsigned long long *stats, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  constexpr int TC = 64/S;                                 // targets per wave
  constexpr int PPT = 32/S;                                // candidate pairs per tile and lane
  constexpr int NB = S;                                    // tiles per batch: NB * 2 * PPT = 64 mask bits per lane (three words
                                                           // per batch - one pass of the fp64 loop over a whole iteration's tiles - measured slower: 1.29 vs 1.10 ms)
  __shared__ int s_idx[GH_DENS_ICAP];                      // particle index | image code << GH_NODE_BITS, -1 = padding
  // a tile in fp32, relative to the quarter's centre: per PAIR of candidates {x0, x1, y0, y1, z0, z1, -, -} (32-byte
  // records: one 16-byte + one 8-byte LDS read feed two packed-fp32 distance evaluations); two buffers - tile t + 1 is
  // staged while tile t is culled.  The fp64 records of the survivors are read again from the (x, y, z, m) pack - L2
  // hits, the tile's lanes have just loaded them - rather than staged in LDS: 5 KB of LDS per wave instead of 15.
  __shared__ __attribute__((aligned(16))) float s_f[2][32][8];

  const int lane = threadIdx.x;
  const int tl = lane & (TC - 1), sl = lane/TC;            // target slot, sub-lane
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int qq = blockIdx.y;
  const int gnode = (1 << d.lgroup) - 1 + q;
  if (d.cN[gnode] == 0 || G.fb[GH_DENS_NQ*q + qq]) return;
  const int nleaf = 1 << (d.ltot - d.lgroup);
  const int nq = min(GH_DENS_NQ, nleaf);
  if (qq >= nq) return;
  int lq = 0;
  while ((1 << lq) < nq) lq++;
  const int qnode = (1 << (d.lgroup + lq)) - 1 + q*nq + qq;
  const int qfirst = d.cfirst[qnode], qN = d.cN[qnode];
  if (qN == 0) return;
  if (qN > TC) { if (lane == 0) G.fb[GH_DENS_NQ*q + qq] = 1; return; }      // (cannot happen with balanced median splits)
  const int nr = G.rlen[q];
  const bool act = tl < qN && (!d.levels || ((int) d.f[D_FLAGS][qfirst + tl] & 1));
  if (!__any(act)) return;

  // ---- the candidate ranges that carry this quarter's bit are expanded into a flat index list in LDS, GH_DENS_ICAP slots
  //      at a time: the tile loop then knows every address in advance (the next tile's records are in flight while the
  //      current one is culled).  Most quarters fit in one fill, which is then kept for all h iterations.
  const int2 *rl = G.rl + (size_t) q*G.rcap;
  int c0 = 0, ntot = 0;
  bool whole = false;
  // returns false if a single chunk of 64 ranges does not fit (whole subtrees inside the search box: a halo group whose
  // kernels cover the core) - the fused kernel streams those
  auto fill = [&]() -> bool {
    ntot = 0;
    const int cstart = c0;
    while (c0 < nr) {
      const int e = c0 + lane;
      int2 ent = make_int2(0, 0);
      if (e < nr) ent = rl[e];
      const int first = ent.x & GH_NODE_MASK, code = ent.x >> GH_NODE_BITS;
      const int cnt = ((ent.y >> (GH_DENS_QSHIFT + qq)) & 1) ? (ent.y & ((1 << GH_DENS_QSHIFT) - 1)) : 0;
      int inc = cnt;
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
      const int excl = inc - cnt, tot = __shfl(inc, 63, 64);
      if (ntot + tot > GH_DENS_ICAP) {
        if (ntot == 0) return false;
        break;
      }
      const bool big = cnt > 8;
      if (!big) for (int k = 0; k < cnt; k++) s_idx[ntot + excl + k] = (first + k) | (code << GH_NODE_BITS);
      unsigned long long bm = __ballot(big);               // whole subtrees inside the search box: the wave writes them together
      while (bm) {
        const int src = __ffsll((long long) bm) - 1;
        bm &= bm - 1ull;
        const int f = __shfl(first, src, 64), c = __shfl(cnt, src, 64), o = __shfl(excl, src, 64), cd = __shfl(code, src, 64);
        for (int k = lane; k < c; k += 64) s_idx[ntot + o + k] = (f + k) | (cd << GH_NODE_BITS);
      }
      ntot += tot;
      c0 += 64;
    }
    const int padded = (ntot + 63) & ~63;
    if (ntot + lane < padded) s_idx[ntot + lane] = -1;
    whole = cstart == 0 && c0 >= nr;
    __syncthreads();
    return true;
  };

  const int i = qfirst + (act ? tl : 0);
  const double invndim = 1.0/(double) ND;
  double ri[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < ND; k++) ri[k] = d.f[D_RX + k][i];
  const double mi = d.f[D_M][i];
  double ui = d.f[D_U][i];
  const CellBox gb = d.cbox[qnode];
  double gc[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < ND; k++) gc[k] = 0.5*(gb.bbmin[k] + gb.bbmax[k]);
  float tf[3] = {0.f, 0.f, 0.f};
  for (int k = 0; k < ND; k++) tf[k] = (float) (ri[k] - gc[k]);
  // the lane's leaf cell: hmax = 1.05 * cell.hmax bounds its iterates (first try of GradhSphTree.cpp:141-226)
  int leafn = qnode;
  while (leafn < d.gtot - 1) { const int c2 = 2*leafn + 2; leafn = (i >= d.cfirst[c2]) ? c2 : 2*leafn + 1; }
  const double hmaxl = 1.05*d.ch[leafn].hmax;
  const double h0 = d.f[D_H][i];
  const double hfloor = P.rho_sink > 0.0 ? P.h_fac*pow(mi/P.rho_sink, 1.0/(double) ND) : 0.0;   // sink runs, see k_density
  double h = h0, hlo = hfloor, hup = hmaxl;
  int iter = 0;
  bool done = !act || hmaxl < hfloor;
  double rho = 0.0, omg = 0.0, zet = 0.0;
  double invh = 0.0, hfactor = 0.0, invhsqd = 0.0;
  unsigned long long n_iter = 0, n_cand = 0, n_tested = 0;
  const double cullsqd = K::kernrangesqd*hmaxl*hmaxl;
  double Rmax = 0.0;                                       // largest |coordinate - quarter centre| a candidate in reach can have
  {
    double hs = wave_max(act ? hmaxl : 0.0);
    for (int k = 0; k < ND; k++) Rmax = fmax(Rmax, 0.5*(gb.bbmax[k] - gb.bbmin[k]) + K::kernrange*hs);
  }
  const bool images = P.dom.periodic[0] | P.dom.periodic[1] | P.dom.periodic[2] | P.dom.mirror[0][0] | P.dom.mirror[0][1] |
                      P.dom.mirror[1][0] | P.dom.mirror[1][1] | P.dom.mirror[2][0] | P.dom.mirror[2][1];

  for (;;) {
    const bool running = !done;
    if (!__any(running)) break;
    if (running) {
      iter++; n_iter++;
      invh = 1.0/h;
      hfactor = powN<ND>(invh);
      invhsqd = invh*invh;
    }
    double prho = 0.0, pomg = 0.0, pzet = 0.0;            // this lane's share of the three sums
    // fp32 cull threshold: a superset of {invhsqd*r2 < kernrangesqd} (see k_density)
    const float thr = running ? (float) ((K::kernrange*h*(1.0 + 1e-6) + 2e-6*Rmax)*(K::kernrange*h*(1.0 + 1e-6) + 2e-6*Rmax)*(1.0 + 1e-6)) : -1.0f;
    int nb = 0, tbase = 0;                                 // tiles in the current batch, first tile of the batch
    unsigned long long mk = 0ull;                          // bit (tile of the batch)*2*PPT + 2*(own pair) + (member of the pair)
    auto process_batch = [&]() {
      while (__any(mk != 0ull)) {
        if (mk != 0ull) {
          const int p = __ffsll((long long) mk) - 1;
          mk &= mk - 1ull;
          const int b = p/(2*PPT), u = p - b*(2*PPT);
          const int id = s_idx[(tbase + b)*64 + (u >> 1)*2*S + (u & 1)*S + sl];        // own pair u/2, member u%2 (see stage)
          double4 v = d.posm[id & GH_NODE_MASK];
          if (images) {
            double sg[3], sh[3];
            code_xform(P.dom, id >> GH_NODE_BITS, sg, sh);
            v.x = sg[0]*v.x + sh[0]; v.y = sg[1]*v.y + sh[1]; v.z = sg[2]*v.z + sh[2];
          }
          double dr[3] = {0.0, 0.0, 0.0};
          dr[0] = v.x - ri[0];
          if (ND > 1) dr[1] = v.y - ri[1];
          if (ND > 2) dr[2] = v.z - ri[2];
          const double mj = v.w;
          double r2 = dr[0]*dr[0];
          if (ND > 1) r2 += dr[1]*dr[1];
          if (ND > 2) r2 += dr[2]*dr[2];
          {
#pragma clang fp contract(fast)
            const double s2 = invhsqd*r2;                              // w0_s2(ssqd) etc., GradhSph.cpp:200-203
            double kw0, kwom, kwz;
            K::t_dens3(s2, P.ktab, kw0, kwom, kwz);
            prho += mj*kw0;
            pomg += mj*invh*kwom;
            pzet += mj*kwz;
          }
        }
      }
      nb = 0;
    };
    // software pipeline over the tiles: record of tile t+1 loaded while tile t is staged and culled
    if (!whole) c0 = 0;
    for (bool first_fill = true; first_fill || c0 < nr; first_fill = false) {
      if (!whole || ntot == 0) {
        __syncthreads();
        if (!fill()) {
          if (lane == 0) G.fb[GH_DENS_NQ*q + qq] = 1;
          return;
        }
      }
      const int ntiles = (ntot + 63) >> 6;
      // stage(t): tile t's candidates, one per lane, as fp32 offsets from the quarter's centre
      auto stage = [&](int t) {
        const int id = s_idx[t*64 + lane];
        double x = 1e30, y = 1e30, z = 1e30;
        if (id >= 0) {
          const double4 v = d.posm[id & GH_NODE_MASK];
          x = v.x; y = v.y; z = v.z;
          if (images) {
            double sg[3], sh[3];
            code_xform(P.dom, id >> GH_NODE_BITS, sg, sh);
            x = sg[0]*v.x + sh[0]; y = sg[1]*v.y + sh[1]; z = sg[2]*v.z + sh[2];
          }
        }
        // candidate c of the tile goes to pair record (c / 2S)*S + c % S, member (c / S) % 2: consecutive candidates - the
        // particles of one leaf, which are neighbours together or not at all - are dealt to different sub-lanes, so the fp64
        // work of a target is shared evenly (with pairs of adjacent candidates per sub-lane the busiest lane of a wave had
        // twice the mean)
        float (*f)[8] = s_f[t & 1];
        const int pr = (lane/(2*S))*S + lane%S, pb = (lane/S) & 1;
        f[pr][pb] = (float) (x - gc[0]);
        f[pr][2 + pb] = ND > 1 ? (float) (y - gc[1]) : 0.f;
        f[pr][4 + pb] = ND > 2 ? (float) (z - gc[2]) : 0.f;
      };
      stage(0);
      __syncthreads();
      tbase = 0;
      for (int t = 0; t < ntiles; t++) {
        if (t + 1 < ntiles) stage(t + 1);                  // into the other buffer
        // support mask in packed fp32: dd = |r_c - r_i|^2 - thr from three packed FMAs per candidate pair; its sign bit
        // (set = inside the conservative threshold) is shifted into the mask by one v_alignbit_b32 per candidate: own pair k
        // (candidates 2S*k + sl, + S) lands in bits 31 - 2k, 30 - 2k of the word, which is bit-reversed at the end.
        unsigned int mw = 0;
        {
          const float (*f)[8] = s_f[t & 1];
          const float2_t tx = {tf[0], tf[0]}, ty = {tf[1], tf[1]}, tz = {tf[2], tf[2]}, nthr = {-thr, -thr};
#pragma unroll
          for (int k = 0; k < PPT; k++) {
            const int c2 = sl + k*S;
            const float4_t xy = *((const float4_t*) &f[c2][0]);
            const float2_t zz = *((const float2_t*) &f[c2][4]);
            const float2_t dx = (float2_t) {xy.x, xy.y} - tx;
            float2_t dd = __builtin_elementwise_fma(dx, dx, nthr);
            if (ND > 1) { const float2_t dy = (float2_t) {xy.z, xy.w} - ty; dd = __builtin_elementwise_fma(dy, dy, dd); }
            if (ND > 2) { const float2_t dz = zz - tz; dd = __builtin_elementwise_fma(dz, dz, dd); }
            if constexpr (S == 1) { if (k == 16) { mk |= (unsigned long long) __brev(mw); mw = 0; } }       // S = 1: 64 bits in two words
            mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(dd.x), 31);
            mw = __builtin_amdgcn_alignbit(mw, __float_as_uint(dd.y), 31);
          }
        }
        // after PPT pairs the word holds 2*PPT bits, first pair on top: reversed, pair k sits at bits 2k, 2k + 1
        if constexpr (S == 1) mk |= (unsigned long long) __brev(mw) << 32;
        else mk |= (unsigned long long) (__brev(mw) >> (32 - 2*PPT)) << (nb*2*PPT);
        if (COUNT) {
          if (running) {
            for (int k = 0; k < 2*PPT; k++) {
              const int id = s_idx[t*64 + (k >> 1)*2*S + (k & 1)*S + sl];
              if (id >= 0) {
                double4 v = d.posm[id & GH_NODE_MASK];
                if (images) {
                  double sg[3], sh[3];
                  code_xform(P.dom, id >> GH_NODE_BITS, sg, sh);
                  v.x = sg[0]*v.x + sh[0]; v.y = sg[1]*v.y + sh[1]; v.z = sg[2]*v.z + sh[2];
                }
                double r2 = 0.0;
                { const double dx = v.x - ri[0]; r2 = dx*dx; }
                if (ND > 1) { const double dy = v.y - ri[1]; r2 += dy*dy; }
                if (ND > 2) { const double dz = v.z - ri[2]; r2 += dz*dz; }
                if (r2 + GH_SMALL <= cullsqd) n_cand++;
              }
            }
            n_tested += 2*PPT;
          }
        }
        nb++;
        if (nb == NB) { process_batch(); tbase = t + 1; }
        __syncthreads();                                   // tile t + 1 is staged, tile t's buffer is free again
      }
      if (nb > 0) process_batch();                         // before the next fill replaces the index list
      if (whole) break;
    }
    process_batch();
    // the S sub-lanes of a target add their partial sums (lanes tl + TC*s): all of them then hold the same three numbers
    for (int off = TC; off < 64; off <<= 1) { prho += __shfl_xor(prho, off, 64); pomg += __shfl_xor(pomg, off, 64); pzet += __shfl_xor(pzet, off, 64); }

    bool failed = false;
    if (running) {
      rho = prho*hfactor; omg = pomg*hfactor; zet = pzet*invhsqd;
      const double hnew = P.h_fac*root_nd<ND>(mi/rho);                     // h_rho_func, Sph.h:259
      if (rho > 0.0 && h > hlo && fabs(h - hnew)*invh < P.h_converge) done = true;
      else {
        if (iter < 30) h = hnew;
        else if (iter == 30) h = 0.5*(hlo + hup);
        else if (iter < 150) {
          if (rho < GH_SMALL || h > hnew) hup = h; else hlo = h;
          h = 0.5*(hlo + hup);
        }
        else { atomicOr(flags, FLAG_H_NOT_CONVERGED); done = true; }
        if (!done) {
          if (!isfinite(h)) { atomicOr(flags, FLAG_H_NOT_CONVERGED); done = true; }
          else if (h > hmaxl) failed = true;                                 // "return 0", :255
          else if (!(h > hlo && h < hup)) done = true;                       // loop exit, :257
        }
      }
    }
    // the reference now redoes the leaf cell with hmax*1.05 - a search volume the list does not cover: hand the
    // quarter-group (whole leaf cells; nothing of it has been stored yet) to the fused kernel
    if (__any(failed)) {
      if (lane == 0) G.fb[GH_DENS_NQ*q + qq] = 1;
      return;
    }
  }

  if (act && sl == 0 && !(hmaxl < hfloor)) density_store<ND, K>(d, P, i, mi, ui, rho, omg, zet, hlo, invhsqd, hmaxl);
  if (COUNT) {
    const unsigned long long a = wave_sum_u64(act && sl == 0 ? n_iter : 0), b = wave_sum_u64(act ? n_cand : 0);
    const unsigned long long t = wave_sum_u64(act ? n_tested : 0);
    if (lane == 0) { atomicAdd(&stats[ST_ITER], a); atomicAdd(&stats[ST_CAND], b); atomicAdd(&stats[ST_PAIRS], t); }
  }
}

static void fill_domain(const gh_ctx *ctx, Domain &dom)
{
  for (int k = 0; k < 3; k++) {
    const bool per = k < ctx->ndim && ctx->cfg.boundary_lhs[k] == GH_BOUNDARY_PERIODIC;
    dom.periodic[k] = per ? 1 : 0;
    dom.mirror[k][0] = k < ctx->ndim && ctx->cfg.boundary_lhs[k] == GH_BOUNDARY_MIRROR;
    dom.mirror[k][1] = k < ctx->ndim && ctx->cfg.boundary_rhs[k] == GH_BOUNDARY_MIRROR;
    dom.bmin[k] = ctx->cfg.boxmin[k]; dom.bmax[k] = ctx->cfg.boxmax[k];
    dom.size[k] = per ? ctx->cfg.boxmax[k] - ctx->cfg.boxmin[k] : 0.0;
    dom.half[k] = 0.5*dom.size[k];
  }
}

void gh_fill_domain(const gh_ctx *ctx, Domain &dom) { fill_domain(ctx, dom); }

void gh_fill_eos(const gh_ctx *ctx, EosParams &e)
{
  e.kind = ctx->cfg.gas_eos; e.gamma = ctx->cfg.gamma_eos; e.gammam1 = ctx->cfg.gamma_eos - 1.0;
  e.temp0 = ctx->cfg.temp0; e.mu_bar = ctx->cfg.mu_bar; e.rho_bary = ctx->cfg.rho_bary;
}

void gh_shard_groups(const gh_ctx *ctx, int rank, int &g0, int &g1)
{
  const int64_t ng = ctx->ngroups;
  g0 = (int) (ng*rank/ctx->nranks);
  g1 = (int) (ng*(rank + 1)/ctx->nranks);
}

int gh_density_impl(gh_ctx *ctx, bool count, bool redo_only)
{
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_INVALID, "gh_update_density: no tree (call gh_build_tree)");
  DevicePtrs d = gh_dev(ctx);
  DensityParams P;
  fill_domain(ctx, P.dom);
  gh_fill_eos(ctx, P.eos);
  P.h_fac = ctx->cfg.h_fac; P.h_converge = ctx->cfg.h_converge; P.ktab = ctx->ktab;
  P.rho_sink = ctx->cfg.sink_particles ? ctx->cfg.rho_sink : 0.0;
#ifdef GH_DEBUG_BLOCKTIME
  static double *dbgbuf = nullptr;
  if (!dbgbuf) (void) hipMalloc((void**) &dbgbuf, sizeof(double)*8*(size_t) ctx->ngroups);
  (void) hipMemset(dbgbuf, 0, sizeof(double)*8*(size_t) ctx->ngroups);
  P.dbg = dbgbuf;
#endif
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  P.group0 = g0;
  P.only_if = nullptr; P.only_val = 1;
  const int nblocks = g1 - g0;
  hipStream_t s = ctx->stream;
  // split path (walk -> candidate range lists in HBM -> evaluation); GH_DENSITY_FUSED=1 keeps the fused kernel alone
  // an extrapolated tree (ntreestockstep > 1) is searched the reference's way, leaf cell by leaf cell (k_density<.., STALE>)
  const bool stale = ctx->tree_stale;
  const bool fused_only = stale || getenv("GH_DENSITY_FUSED") != nullptr;
  DensLists G;
  G.rcap = GH_DENS_RCAP;
  if (ctx->dl_groups != ctx->ngroups) {
    for (void *p : {(void*) ctx->dl_rl, (void*) ctx->dl_rlen}) if (p) (void) hipFree(p);
    ctx->dl_rl = nullptr; ctx->dl_rlen = nullptr; ctx->dl_groups = 0;
    GH_CHECK(ctx, hipMalloc((void**) &ctx->dl_rl, sizeof(int2)*(size_t) ctx->ngroups*G.rcap));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->dl_rlen, sizeof(int)*(1 + GH_DENS_NQ)*(size_t) ctx->ngroups));
    GH_CHECK(ctx, hipMemset(ctx->dl_rlen, 0, sizeof(int)*(1 + GH_DENS_NQ)*(size_t) ctx->ngroups));
    ctx->dl_groups = ctx->ngroups;
  }
  G.rl = (int2*) ctx->dl_rl; G.rlen = ctx->dl_rlen; G.fb = ctx->dl_rlen + ctx->ngroups;
  P.fbout = G.fb;
  P.miss_count = (unsigned int*) (ctx->d_blk + 12);
  const bool dd = ctx->nranks > 1;
  if (dd && !redo_only) GH_CHECK(ctx, hipMemsetAsync(P.miss_count, 0, sizeof(unsigned int), s));
  if (dd && fused_only && !redo_only) GH_CHECK(ctx, hipMemsetAsync(G.fb, 0, sizeof(int)*GH_DENS_NQ*(size_t) ctx->ngroups, s));
  gh_phase_begin(ctx, GH_T_SPH_PROPERTIES);
  // evaluation: one wave per quarter-group (min(4, leaves per group) subtrees of <= 64/S particles, S sub-lanes per target)
  const int nqg = std::min(GH_DENS_NQ, 1 << (ctx->ltot - ctx->lgroup));
#define LAUNCH_EVAL(ND_, KT_, S_) \
    if (count) hipLaunchKernelGGL((k_dens_eval<ND_, true, KT_, S_>), dim3(nblocks, nqg), dim3(64), 0, s, d, P, G, ctx->d_stats, ctx->d_flags); \
    else hipLaunchKernelGGL((k_dens_eval<ND_, false, KT_, S_>), dim3(nblocks, nqg), dim3(64), 0, s, d, P, G, ctx->d_stats, ctx->d_flags);
  if (nblocks > 0 && !fused_only && !redo_only) {
#define LAUNCH(ND_, KT_)                                                                                      \
    hipLaunchKernelGGL((k_dens_walk<ND_, KT_>), dim3(nblocks), dim3(64), 0, s, d, P, G, ctx->d_flags);        \
    if (nqg == 4) { LAUNCH_EVAL(ND_, KT_, 4) } else if (nqg == 2) { LAUNCH_EVAL(ND_, KT_, 2) } else { LAUNCH_EVAL(ND_, KT_, 1) }
    GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
    P.only_if = G.fb;        // the fused kernel redoes what the split path flagged (early exit per group otherwise)
  }
#define LAUNCH(ND_, KT_)                                                                                      \
    if (stale) hipLaunchKernelGGL((k_density<ND_, false, KT_, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
    else if (count) hipLaunchKernelGGL((k_density<ND_, true, KT_>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
    else hipLaunchKernelGGL((k_density<ND_, false, KT_>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags);
  if (nblocks > 0 && !redo_only) { GH_DISPATCH(ctx, LAUNCH) }
  if (dd && ctx->dd_defer_miss && !redo_only) ctx->dd_miss_pending = true;      // checked with the force-phase halo counts
  else if (dd) {
    // multi-GPU: groups whose walk left the imported halo stored nothing; widen the import and redo just those (the
    // h iteration may grow a smoothing length past any margin fixed in advance - e.g. the first pass from a guessed h).
    // The decision is collective: every rank takes part in every widened exchange.
    P.only_if = G.fb; P.only_val = 2;
    double widen = 1.0;
    for (int attempt = 0; ; attempt++) {
      int any = 1, rc = 0;                                // redo_only: the deferred check already found a miss
      if (!(redo_only && attempt == 0)) rc = gh_dd_any(ctx, P.miss_count, &any);        // sum over ranks of the miss counters (synchronises)
      if (rc) return rc;
      if (!any) break;
      if (attempt >= 24) return gh_fail(ctx, GH_ERR_CAPACITY, "gh_update_density: halo import did not converge");
      widen *= 2.0;
      if ((rc = gh_dd_exchange_margin(ctx, GH_HALO_DENSITY, widen))) return rc;
      GH_CHECK(ctx, hipMemsetAsync(P.miss_count, 0, sizeof(unsigned int), s));
      if (nblocks > 0) { GH_DISPATCH(ctx, LAUNCH) }
    }
  }
#undef LAUNCH
  gh_phase_end(ctx, GH_T_SPH_PROPERTIES);
#ifdef GH_DEBUG_BLOCKTIME
  if (const char *f = getenv("GH_DEBUG_BLOCKTIME_FILE")) {
    std::vector<double> h((size_t) 8*ctx->ngroups);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipMemcpy(h.data(), dbgbuf, sizeof(double)*h.size(), hipMemcpyDeviceToHost);
    FILE *fp = fopen(f, "wb");
    if (fp) { fwrite(h.data(), sizeof(double), h.size(), fp); fclose(fp); }
  }
#endif
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

