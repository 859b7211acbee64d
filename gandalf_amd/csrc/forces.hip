// forces.hip -- grad-h SPH hydro forces, and hydro + self-gravity forces, on the GPU.
//
// Replaces GradhSphTree::UpdateAllSphHydroForces / UpdateAllSphForces and everything they call per cell:
// Tree::ComputeNeighbourAndGhostList, Tree::ComputeGravityInteractionAndGhostList (+ open_cell_for_gravity),
// NeighbourManager::_EndSearch / TrimNeighbourLists, GradhSph::ComputeSphHydroForces /
// ComputeSphHydroGravForces / ComputeDirectGravForces and ComputeCellMonopoleForces
// (reference src/GradhSph/GradhSphTree.cpp:280-657, src/Tree/Tree.cpp:562-735, src/Headers/Tree.h:413-432,
//  src/Headers/NeighbourManager.h:368-543, src/GradhSph/GradhSph.cpp:361-690,
//  src/Headers/NeighbourSearch.h:350-377).
//
// Mapping: one wavefront per group of <= 64 particles (= up to 16 leaf cells of the KD-tree), one lane
// per particle.  The reference builds one interaction list per LEAF CELL; to evaluate exactly the same
// interactions the wave walks the tree once for all of its leaves and keeps, per frontier node, a
// 16-bit mask of the leaves that still descend through it.  Every classification of the reference's
// walk is taken per (node, leaf) with the leaf's own rcell/rmax/hmax; accepted cells and near-field
// particles go to LDS lists tagged with the mask of leaves they belong to, and are flushed through
// broadcast LDS tiles: lanes whose leaf bit is clear skip the entry.  The walks are the streaming
// depth-first walks of walk.hpp (bounded LDS whatever the size of the interaction region).
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

#include "force_common.hpp"
#include <cstdlib>

// ================================================================================================
// hydro only                                                     (GradhSphTree.cpp:280-435)
// ================================================================================================
// STALE: extrapolated tree (see k_density<.., STALE>): the candidate walk is the reference's per-leaf-cell one
// (Tree.cpp:562-617 against the drifted boxes) followed by NeighbourManager::_EndSearch's list filter with the drifted
// cell centre (NeighbourManager.h:440-455), so that the neighbours the reference loses are lost here too.
template <int ND, bool COUNT, int KT, bool LV, bool STALE = false>
__global__ __launch_bounds__(64) void k_hydro_forces(DevicePtrs d, ForceParams P, unsigned long long *stats, int *flags)
{
  __shared__ WalkLDS<int> L;
  __shared__ double s_t[T_NFA][64];
  __shared__ int s_tj[LV ? 64 : 1], s_tlv[LV ? 64 : 1];   // block timesteps: particle index and level of every tile slot
  __shared__ unsigned short s_smask[STALE ? GH_SCAP : 1], s_tagm[STALE ? 64 : 1];
  __shared__ double s_lbb[STALE ? 16 : 1][6], s_lhb[STALE ? 16 : 1][6], s_lrc[STALE ? 16 : 1][3], s_lrm[STALE ? 16 : 1], s_lhm[STALE ? 16 : 1];

  const int lane = threadIdx.x;
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (gN == 0) return;
  // block timesteps (Nlevels > 1: the LV instantiation): targets are the active particles only and the pair loop also
  // maintains levelneib
  constexpr bool lv = LV;
  bool act = lane < gN && (!lv || ((int) d.f[D_FLAGS][gfirst + lane] & 1));
  if (STALE && d.leafact) {                         // cells on the reference's (stale) active list only, see k_leaf_nactive
    const int ip = gfirst + (lane < gN ? lane : 0);
    int ln = gnode;
    while (ln < d.gtot - 1) { const int c2 = 2*ln + 2; ln = (ip >= d.cfirst[c2]) ? c2 : 2*ln + 1; }
    act = act && d.leafact[ln - (d.gtot - 1)] > 0;
  }
  if (lv && !__any(act)) return;
  const int i = gfirst + (act ? lane : 0);
  const int mylevel = lv ? (int) d.f[D_LEVEL][i] : 0;
  int lnmax = 0;
  TargetI ti;
  load_target(d, i, ND, ti);
  // STALE: the leaf cells of the group, and which of them hold an active particle (Tree::ComputeActiveCellList)
  const int nleaf = 1 << (d.ltot - d.lgroup);
  const int leafnode0 = (d.gtot - 1) + (gnode - ((1 << d.lgroup) - 1))*nleaf;
  int li = 0;
  unsigned int m0 = 0;
  if (STALE) {
    int leafn = gnode;
    while (leafn < d.gtot - 1) { const int c2 = 2*leafn + 2; leafn = (i >= d.cfirst[c2]) ? c2 : 2*leafn + 1; }
    li = leafn - leafnode0;
    if (lane < nleaf) {
      const CellBox lb = d.cbox[leafnode0 + lane]; const CellH lh = d.ch[leafnode0 + lane]; const CellGeo lg = d.cgeo[leafnode0 + lane];
      for (int k = 0; k < 3; k++) {
        s_lbb[lane][k] = lb.bbmin[k]; s_lbb[lane][3 + k] = lb.bbmax[k]; s_lhb[lane][k] = lh.hbmin[k]; s_lhb[lane][3 + k] = lh.hbmax[k];
        s_lrc[lane][k] = lg.rcell[k];
      }
      s_lrm[lane] = lg.rmax; s_lhm[lane] = lh.hmax;
    }
    for (int l = 0; l < nleaf; l++) if (__any(act && li == l)) m0 |= 1u << l;
    __syncthreads();
  }
  const bool mm97 = P.avisc == GH_AVISC_MON97MM97;
  const bool tdav = mm97 || P.avisc == GH_AVISC_MON97CD2010;      // per-particle alpha (the pair term uses the mean)
  ti.alpha = tdav ? d.f[D_ALPHA][i] : 0.0;
  Accum A;
  for (int k = 0; k < 3; k++) { A.a[k] = 0.0; A.at[k] = 0.0; }
  A.dudt = 0.0; A.div_v = 0.0; A.gpot = 0.0;
  unsigned long long n_pairs = 0;

  // candidate cells: overlap(cell.bb, other.hbox) || overlap(cell.hbox, other.bb)   (Tree.cpp:579-580)
  const CellBox gb = d.cbox[gnode];
  const CellH gh = d.ch[gnode];
  const double hr_root = KSel<ND, KT>::type::kernrange*d.ch[0].hmax;
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) {
    lo[k] = k < ND ? fmin(gh.hbmin[k], gb.bbmin[k] - hr_root) : -1e300;
    hi[k] = k < ND ? fmax(gh.hbmax[k], gb.bbmax[k] + hr_root) : 1e300;
  }
  const unsigned int codes = image_codes(P.dom, ND, lo, hi);
  auto cls = [&](int n, int code, bool &open, bool &emit, int &first, int &cnt) {
    const CellBox b = d.cbox[n];
    const int cn = b.N;
    if (cn < 0) atomicOr(flags, FLAG_LET_MISS);            // multi-GPU: a remote cell the halo exchange did not import
    if (cn <= 0) return;
    double sg[3], sh[3];
    code_xform(P.dom, code, sg, sh);
    const CellH bh = d.ch[n];
    bool o1 = true, o2 = true, inside = true;
    for (int k = 0; k < ND; k++) {
      double bmin, bmax, hmin, hmax_;
      image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin, bmax);
      image_interval(sg[k], sh[k], bh.hbmin[k], bh.hbmax[k], hmin, hmax_);
      if (gb.bbmin[k] > hmax_ || hmin > gb.bbmax[k]) o1 = false;
      if (gh.hbmin[k] > bmax || bmin > gh.hbmax[k]) o2 = false;
      if (bmin < gh.hbmin[k] || bmax > gh.hbmax[k]) inside = false;
    }
    if (!(o1 || o2)) return;
    if (inside || n >= d.gtot - 1) { emit = true; first = b.first; cnt = cn; }
    else open = true;
  };
  auto tile = [&](bool valid, int j, int tag) {
    const int code = STALE ? (tag & 31) : tag;
    {
      double sg[3], sh[3];
      code_xform(P.dom, code, sg, sh);
      stage_neib(d, ND, s_t, lane, j, sg, sh, valid);
      if (STALE) s_tagm[lane] = (unsigned short) (valid ? (tag >> 5) : 0);
      if (tdav) s_t[T_ALPHA][lane] = valid ? d.f[D_ALPHA][j] : 0.0;
      if (lv) { s_tj[lane] = valid ? j : 0; s_tlv[lane] = valid ? (int) d.f[D_LEVEL][j] : 0; }
    }
    __syncthreads();
    unsigned long long mask = 0;
    if (act) {
#pragma unroll 16
      for (int c = 0; c < 64; c++) {
        double r2;
        { const double dx = s_t[T_X][c] - ti.r[0]; r2 = dx*dx; }
        if (ND > 1) { const double dy = s_t[T_Y][c] - ti.r[1]; r2 += dy*dy; }
        if (ND > 2) { const double dz = s_t[T_Z][c] - ti.r[2]; r2 += dz*dz; }
        // neighbour unless (r2 >= hrangesqd_i && r2 >= hrangesqd_j)      (NeighbourManager.h:521)
        if (!(r2 >= ti.hr2 && r2 >= s_t[T_HR2][c])) {
          bool keep = true;
          if (STALE) {
            // on this particle's list only if the walk of ITS leaf reached the candidate's leaf and _EndSearch keeps it:
            // |r_j - rcell|^2 < (rmax + kernrange*hmax_cell)^2 or < (rmax + kernrange*h_j)^2
            double dc = s_t[T_X][c] - s_lrc[li][0], d2c = dc*dc;
            if (ND > 1) { dc = s_t[T_Y][c] - s_lrc[li][1]; d2c += dc*dc; }
            if (ND > 2) { dc = s_t[T_Z][c] - s_lrc[li][2]; d2c += dc*dc; }
            const double h1 = s_lrm[li] + KSel<ND, KT>::type::kernrange*s_lhm[li], h2 = s_lrm[li] + KSel<ND, KT>::type::kernrange/s_t[T_INVH][c];
            keep = ((s_tagm[c] >> li) & 1) && (d2c < h1*h1 || d2c < h2*h2);
          }
          if (keep) mask |= 1ull << c;
        }
      }
    }
    while (__any(mask != 0ull)) {
      if (mask != 0ull) {
        const int c = __ffsll((long long) mask) - 1;
        mask &= mask - 1ull;
        double dr[3] = {0.0, 0.0, 0.0};
        dr[0] = s_t[T_X][c] - ti.r[0];
        if (ND > 1) dr[1] = s_t[T_Y][c] - ti.r[1];
        if (ND > 2) dr[2] = s_t[T_Z][c] - ti.r[2];
        double r2 = dr[0]*dr[0];
        if (ND > 1) r2 += dr[1]*dr[1];
        if (ND > 2) r2 += dr[2]*dr[2];
        Neib nbr;
        neib_from_tile(nbr, s_t, c);
        nbr.alpha = tdav ? s_t[T_ALPHA][c] : 0.0;
        sph_pair<ND, false, KT>(P, ti, A, nbr, dr, r2);
        if (COUNT) n_pairs++;
        if (lv) { lnmax = max(lnmax, s_tlv[c]); raise_levelneib(d, s_tj[c], mylevel); }
      }
    }
    __syncthreads();
  };
  if (STALE) {
    // per (node, leaf): overlap(leaf.bb, node.hbox) || overlap(leaf.hbox, node.bb)          (Tree.cpp:579-580)
    auto clsm = [&](int n, int code, unsigned int inmask, int &first, int &cnt) -> unsigned int {
      const CellBox b = d.cbox[n];
      first = b.first; cnt = b.N;
      if (b.N <= 0) return 0u;
      const CellH bh = d.ch[n];
      double sg[3], sh[3];
      code_xform(P.dom, code, sg, sh);
      double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0}, hmin[3] = {0, 0, 0}, hmx[3] = {0, 0, 0};
      for (int k = 0; k < ND; k++) {
        image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin[k], bmax[k]);
        image_interval(sg[k], sh[k], bh.hbmin[k], bh.hbmax[k], hmin[k], hmx[k]);
      }
      unsigned int om = 0;
      for (int l = 0; l < nleaf; l++) {
        if (!((inmask >> l) & 1)) continue;
        bool o1 = true, o2 = true;
        for (int k = 0; k < ND; k++) {
          if (s_lbb[l][k] > hmx[k] || hmin[k] > s_lbb[l][3 + k]) o1 = false;
          if (s_lhb[l][k] > bmax[k] || bmin[k] > s_lhb[l][3 + k]) o2 = false;
        }
        if (o1 || o2) om |= 1u << l;
      }
      return om;
    };
    walk_dfs_stream_masked(d, L, s_smask, codes, m0, clsm, tile, flags);
  }
  else walk_dfs_stream(d, L, codes, cls, tile, flags);

  if (act) {
    // GradhSph.cpp:451-453 then GradhSphTree.cpp:396-404 (accumulate on the zeroed main array)
    A.div_v *= ti.invrho;
    A.dudt -= ti.press*A.div_v*ti.invrho*d.f[D_INVOMEGA][i];
    for (int k = 0; k < ND; k++) d.f[D_AX + k][i] += A.a[k];
    d.f[D_DUDT][i] += A.dudt;
    d.f[D_DIV_V][i] += A.div_v;
    // GradhSph.cpp:454-457; the driver ACCUMULATES it on the main array (GradhSphTree.cpp:403) and nothing zeroes
    // it between steps (Sph::ZeroAccelerations does not) - restated as it is
    if (mm97) d.f[D_DALPHADT][i] += 0.1*ti.sound*(P.alpha_visc_min - ti.alpha)*ti.invh + fmax(-A.div_v, 0.0)*(P.alpha_visc - ti.alpha);
    if (lv) raise_levelneib(d, i, lnmax);                 // GradhSph.cpp:445, GradhSphTree.cpp:405
  }
  if (COUNT) {
    const unsigned long long a = wave_sum_u64(act ? n_pairs : 0);
    if (lane == 0) atomicAdd(&stats[ST_PAIRS], a);
  }
}

// ================================================================================================
// hydro + self-gravity                                           (GradhSphTree.cpp:444-657)
// ================================================================================================
#define GH_MAXLEAF 16
#define GH_CCAP 132          /* far-field entry list capacity (flushed when more than 64 are pending; +4 padding) */
#define GH_NDCAP 72          /* per-leaf list of direct-only near leaves (rare: only opened, non-overlapping leaves) */
#define GH_NHCAP 96          /* per-leaf list of near leaves with hydro candidates */

#ifdef GH_STAMPS
#define STAMP(var) const long long var = clock64()
#define STAMP_ADD(slot, t0) st_acc[slot] += clock64() - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(slot, t0)
#endif

// near-list entry: first particle (26 bits, like GH_NODE_BITS) | count << 26 (6 bits: leaves of up to 32 particles, Nleafmax <= 32)
__device__ __forceinline__ int near_entry(int first, int n) { return first | (int) ((unsigned int) n << 26); }

template <int ND, bool COUNT, int KT, bool TDAV = false>
__global__ __launch_bounds__(64) void k_grav_forces(DevicePtrs d, ForceParams P, unsigned long long *stats, int *flags, const int *only_if)
{
  typedef typename KSel<ND, KT>::type K;
  if (only_if && !*only_if) return;                   // fallback launch that is not needed
#ifdef GH_STAMPS
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long st_t0 = clock64();
#endif
  __shared__ int s_stack[GH_SCAP];
  __shared__ unsigned short s_smask[GH_SCAP];      // leaf mask of every stack entry
  __shared__ double s_cx[GH_CCAP], s_cy[GH_CCAP], s_cz[GH_CCAP], s_cm[GH_CCAP];
  __shared__ unsigned short s_cmask[GH_CCAP];
  // near field: the reference's interaction list is per LEAF CELL, and the near-field leaves of the 16
  // leaves of a group overlap only partly (a lane needs ~1/5 of their union).  So each leaf keeps its
  // own lists of near leaves and every lane streams exactly its leaf's particles.
  __shared__ int s_ndir[GH_MAXLEAF][GH_NDCAP], s_nhyd[GH_MAXLEAF][GH_NHCAP];
  __shared__ int s_nlen[2][GH_MAXLEAF];
  // per lane: which streamed candidates are SPH neighbours.  2 048 (entry, slot) bits: the near lists are flushed before
  // (pending entries + 64) * leaf width can exceed them (hthr below; leaves of up to 32 particles)
  __shared__ unsigned int s_sphbits[64][64];
  __shared__ double s_lrc[GH_MAXLEAF][3], s_lrmax[GH_MAXLEAF], s_lhr[GH_MAXLEAF];

  const int lane = threadIdx.x;
  const unsigned long long lt = lanemask_lt();
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (gN == 0) return;
  const bool lv = d.levels != 0;                  // block timesteps: see k_hydro_forces
  const bool act = lane < gN && (!lv || ((int) d.f[D_FLAGS][gfirst + lane] & 1));
  const int i = gfirst + (act ? lane : 0);
  const int mylevel = lv ? (int) d.f[D_LEVEL][i] : 0;
  int lnmax = 0;
  const int nl = 1 << (d.ltot - d.lgroup);            // leaves in this group (<= 16)
  const int leafnode0 = (d.gtot - 1) + q*nl;

  // leaf geometry of the group (the reference walks per leaf cell with these, Tree.cpp:639-643)
  unsigned int allmask = 0;
  const CellGeo gg = d.cgeo[gnode];
  double Rg = 0.0, Lm = 0.0, Lr = 0.0;               // group-level bounds for the quick classification
  {
    double rg = 0.0, lm = 0.0, lr = 0.0;
    if (lane < nl) {
      const CellGeo g = d.cgeo[leafnode0 + lane];
      for (int k = 0; k < 3; k++) s_lrc[lane][k] = g.rcell[k];
      s_lrmax[lane] = g.rmax;
      s_lhr[lane] = K::kernrange*g.hmax;
      if (g.N > 0) {
        double dd = 0.0;
        for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - gg.rcell[k]; dd += dx*dx; }
        rg = sqrt(dd); lm = g.rmax + K::kernrange*g.hmax; lr = g.rmax;
      }
    }
    Rg = wave_max(rg)*(1.0 + 1e-12); Lm = wave_max(lm); Lr = wave_max(lr);
  }
  for (int l = 0; l < nl; l++) if (d.cN[leafnode0 + l] > 0) allmask |= 1u << l;
  int myleaf = 0;
  for (int l = 1; l < nl; l++) if (i >= d.cfirst[leafnode0 + l]) myleaf = l;
  __syncthreads();

  TargetI ti;
  load_target(d, i, ND, ti);
  if (TDAV) ti.alpha = d.f[D_ALPHA][i];
  Accum A;
  for (int k = 0; k < 3; k++) { A.a[k] = 0.0; A.at[k] = 0.0; }
  A.dudt = 0.0; A.div_v = 0.0;
  A.gpot = (d.f[D_M][i]/d.f[D_H][i])*K::t_wpot0(P.ktab);      // self term, GradhSphTree.cpp:512
  unsigned long long n_pairs = 0, n_direct = 0, n_cells = 0;
  const int occ = d.leafocc;
  const int hthr = min(GH_NHCAP - 64, 2048/occ - 64);   // near lists are flushed beyond this many pending hydro entries

  int maxlen_d = 0, maxlen_h = 0;                     // wave-uniform maxima of the near-list lengths
  int len_d[GH_MAXLEAF], len_h[GH_MAXLEAF];
#pragma unroll
  for (int l = 0; l < GH_MAXLEAF; l++) { len_d[l] = 0; len_h[l] = 0; }
  int ncell = 0;
  // ---- cells: monopole                                           (NeighbourSearch.h:350-377)
  auto flush_cells = [&]() {
    STAMP(tc0);
    // pad to a multiple of 4 with empty entries so that the loop can be unrolled without a remainder
    if (lane < 4 && ncell + lane < GH_CCAP) { s_cmask[ncell + lane] = 0; s_cm[ncell + lane] = 0.0; s_cx[ncell + lane] = 1e30; s_cy[ncell + lane] = 1e30; s_cz[ncell + lane] = 1e30; }
    __syncthreads();
    const int nc4 = (ncell + 3) & ~3;
    for (int c0 = 0; c0 < nc4; c0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u;
        const bool take = (s_cmask[c] >> myleaf) & 1;
        point_mass<ND>(ti, A, s_cx[c], s_cy[c], s_cz[c], take ? s_cm[c] : 0.0);
        if (COUNT) n_cells += take ? 1 : 0;
      }
    }
    __syncthreads();
    ncell = 0;
    STAMP_ADD(1, tc0);
  };
  // ---- near field: every lane streams the particles of its own leaf's near leaves
  auto flush_near = [&]() {
    STAMP(tn0);
    if (lane == 0) {
#pragma unroll
      for (int l = 0; l < GH_MAXLEAF; l++) { s_nlen[0][l] = len_d[l]; s_nlen[1][l] = len_h[l]; }
    }
    __syncthreads();
    const int Ld = s_nlen[0][myleaf], Lh = s_nlen[1][myleaf];
    const int maxd = maxlen_d, maxh = maxlen_h;
    // direct-only leaves: Newtonian particle terms                  (GradhSph.cpp:671-686).
    // Software pipeline: the records of entry e+1 are in flight while entry e is evaluated.
    {
      // (unit = six consecutive slots of an entry's leaf: leaves wider than 6 particles take nck units per entry)
      const int nck = (occ + 5)/6;
      auto dload = [&](int u, double4 (&v)[6], int &cnt) {
        const int e = u/nck, k0 = (u - e*nck)*6;
        const int ent = e < Ld ? s_ndir[myleaf][e] : 0;
        const int first = (ent & 0x3ffffff) + k0;
        cnt = act ? max((int) ((unsigned int) ent >> 26) - k0, 0) : 0;
#pragma unroll
        for (int k = 0; k < 6; k++) { v[k].x = 1e30; v[k].y = 1e30; v[k].z = 1e30; v[k].w = 0.0; if (k < cnt) v[k] = d.posm[first + k]; }
      };
      auto dcomp = [&](const double4 (&v)[6], int cnt) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
          if (k < occ) point_mass<ND>(ti, A, v[k].x, v[k].y, v[k].z, v[k].w);
        }
        if (COUNT) n_direct += min(cnt, 6);
      };
      double4 va[6], vb[6];
      int ca = 0, cb = 0;
      const int maxu = maxd*nck;
      if (maxu > 0) dload(0, va, ca);
      for (int e = 0; e < maxu; e += 2) {
        dload(e + 1, vb, cb);
        dcomp(va, ca);
        dload(e + 2, va, ca);
        if (e + 1 < maxu) dcomp(vb, cb);
      }
    }
    STAMP_ADD(2, tn0);
    STAMP(th0);
    // leaves with hydro candidates: per pair SPH neighbour or direct  (NeighbourManager.h:521-533).
    // Pass 1 streams every candidate of the lane's leaf (position, mass, hrangesqd only), evaluates the
    // direct ones at once and records the SPH neighbours in a bit set; pass 2 walks the lane's own bits
    // and evaluates the (long) SPH pair term only for those - no lane waits on another lane's branch.
    {
      const int nt = maxh*occ;                       // flattened (entry, slot) index, same for all lanes
      auto hload = [&](int t, double4 &q0, double &hr2, bool &valid) {
        const int e = t/occ, k = t - e*occ;
        const int ent = e < Lh ? s_nhyd[myleaf][e] : 0;
        const int first = ent & 0x3ffffff, cnt = act ? (int) ((unsigned int) ent >> 26) : 0;
        valid = k < cnt;
        q0.x = 1e30; q0.y = 1e30; q0.z = 1e30; q0.w = 0.0; hr2 = 0.0;
        if (valid) { const double4 *r = d.hrec + 4*(size_t) (first + k); q0 = r[0]; hr2 = r[1].w; }
      };
      unsigned int cur = 0;
      auto hcomp = [&](int t, const double4 &q0, double hr2, bool valid) {
        double dr[3] = {0.0, 0.0, 0.0};
        dr[0] = q0.x - ti.r[0];
        if (ND > 1) dr[1] = q0.y - ti.r[1];
        if (ND > 2) dr[2] = q0.z - ti.r[2];
        double r2 = dr[0]*dr[0];
        if (ND > 1) r2 += dr[1]*dr[1];
        if (ND > 2) r2 += dr[2]*dr[2];
        const bool sph = valid && !(r2 >= ti.hr2 && r2 >= hr2);
        cur |= (sph ? 1u : 0u) << (t & 31);
        {
#pragma clang fp contract(fast)
          const double mj = sph ? 0.0 : q0.w;                           // 0 for empty slots and SPH neighbours
          const double invdrmag = fast_rsqrt(r2 + GH_SMALL);
          const double minvdr3 = mj*(invdrmag*invdrmag*invdrmag);
          for (int kk = 0; kk < ND; kk++) A.at[kk] += dr[kk]*minvdr3;
          A.gpot += mj*invdrmag;
        }
        if (COUNT) n_direct += (valid && !sph) ? 1 : 0;
        if ((t & 31) == 31) { s_sphbits[t >> 5][lane] = cur; cur = 0; }
      };
      double4 qa, qb; double ha = 0.0, hb = 0.0; bool vala = false, valb = false;
      if (nt > 0) hload(0, qa, ha, vala);
      for (int t = 0; t < nt; t += 2) {
        hload(t + 1, qb, hb, valb);
        hcomp(t, qa, ha, vala);
        hload(t + 2, qa, ha, vala);
        if (t + 1 < nt) hcomp(t + 1, qb, hb, valb);
      }
      const int nw = (nt + 31) >> 5;
      if (nt & 31) s_sphbits[nt >> 5][lane] = cur;
      // pass 2 (each lane reads back only what it wrote: no barrier needed)
      int w = 0;
      unsigned int bits = nw > 0 ? s_sphbits[0][lane] : 0u;
      for (;;) {
        while (bits == 0u && w + 1 < nw) bits = s_sphbits[++w][lane];
        if (!__any(bits != 0u)) break;
        if (bits != 0u) {
          const int t = w*32 + __ffs((int) bits) - 1;
          bits &= bits - 1u;
          const int e = t/occ, k = t - e*occ;
          const int first = s_nhyd[myleaf][e] & 0x3ffffff;
          const double4 *r = d.hrec + 4*(size_t) (first + k);
          const double4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3];
          Neib nb;
          nb.x = q0.x; nb.y = q0.y; nb.z = q0.z; nb.m = q0.w; nb.vx = q1.x; nb.vy = q1.y; nb.vz = q1.z; nb.hr2 = q1.w;
          nb.invh = q2.x; nb.hfac = q2.y; nb.pfac = q2.z; nb.invrho = q2.w; nb.sound = q3.x; nb.zeta = q3.y; nb.u = q3.z; nb.press = q3.w;
          double dr[3] = {0.0, 0.0, 0.0};
          dr[0] = nb.x - ti.r[0];
          if (ND > 1) dr[1] = nb.y - ti.r[1];
          if (ND > 2) dr[2] = nb.z - ti.r[2];
          double r2 = dr[0]*dr[0];
          if (ND > 1) r2 += dr[1]*dr[1];
          if (ND > 2) r2 += dr[2]*dr[2];
          if (TDAV) nb.alpha = d.f[D_ALPHA][first + k];
          sph_pair<ND, true, KT, TDAV>(P, ti, A, nb, dr, r2);
          if (COUNT) n_pairs++;
          if (lv && act) { lnmax = max(lnmax, (int) d.f[D_LEVEL][first + k]); raise_levelneib(d, first + k, mylevel); }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < GH_MAXLEAF; l++) { len_d[l] = 0; len_h[l] = 0; }
    maxlen_d = 0; maxlen_h = 0;
    STAMP_ADD(3, th0);
  };

  // ---- depth-first walk with per-leaf masks                       (Tree.cpp:648-731)
  if (lane == 0) { s_stack[0] = 0; s_smask[0] = (unsigned short) allmask; }
  __syncthreads();
  int top = 1;
  const int leaf0 = d.gtot - 1;
  while (top > 0) {
    if (ncell > 64) flush_cells();
    if (maxlen_d > GH_NDCAP - 64 || maxlen_h > hthr) flush_near();
    STAMP(tw0);
    const int p = pop_width(top);
    const int newtop = top - p;
    unsigned int openm = 0, cellm = 0, hydm = 0, dirm = 0;
    int n = 0; bool isleaf = false;
    CellGeo g;
    g.first = 0; g.N = 0;
    if (lane < p) {
      n = s_stack[top - 1 - lane];
      const unsigned int fm = s_smask[top - 1 - lane];
      g = d.cgeo[n];
      if (g.N < 0) { atomicOr(flags, FLAG_LET_MISS); g.N = 0; }     // multi-GPU: a remote cell the halo exchange did not import
      isleaf = n >= leaf0;
      const double khr = K::kernrange*g.hmax;
      // quick classification of the node against the whole group.  Every leaf centre lies within Rg of
      // the group centre, so |dr_leaf| >= D - Rg for all leaves; if that lower bound already clears
      // both the overlap distances and the opening distance, each per-leaf test below would say "cell".
      double D2 = 0.0;
      for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - gg.rcell[k]; D2 += dx*dx; }
      const double Dm = (sqrt(D2) - Rg)*(1.0 - 1e-12);
      const double Tn = g.rmax + fmax(Lm, Lr + khr);
      if (g.N > 0 && Dm > Tn && Dm*Dm > g.cdistsqd) {
        if (isleaf && g.N == 1) dirm = fm; else cellm = fm;
      }
      else {
        for (int l = 0; l < nl; l++) {
          if (!((fm >> l) & 1)) continue;
          double drsqd = 0.0;
          for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - s_lrc[l][k]; drsqd += dx*dx; }
          const double d1 = g.rmax + s_lrmax[l] + s_lhr[l];
          const double d2 = s_lrmax[l] + g.rmax + khr;
          if (drsqd <= d1*d1 || drsqd <= d2*d2) {                  // overlap -> hydro candidates / open
            if (!isleaf) openm |= 1u << l;
            else if (g.N > 0) hydm |= 1u << l;
          }
          else if (g.N == 0) { }
          else if (!(drsqd < g.cdistsqd)) {                         // !open_cell_for_gravity (geometric MAC)
            if (isleaf && g.N == 1) dirm |= 1u << l;
            else cellm |= 1u << l;
          }
          else {
            if (!isleaf) openm |= 1u << l;
            else dirm |= 1u << l;
          }
        }
      }
    }
    const unsigned long long om = __ballot(openm != 0), cm = __ballot(cellm != 0);
    const bool anynear = __any((hydm | dirm) != 0);
    __syncthreads();
    if (openm) {
      const int pos = newtop + 2*__popcll(om & lt);
      if (pos + 1 < GH_SCAP) {
        s_stack[pos] = 2*n + 1; s_smask[pos] = (unsigned short) openm;
        s_stack[pos + 1] = 2*n + 2; s_smask[pos + 1] = (unsigned short) openm;
      }
    }
    top = newtop + 2*__popcll(om);
    if (top > GH_SCAP) { if (lane == 0) atomicOr(flags, FLAG_FRONTIER_OVERFLOW); top = GH_SCAP; }
    if (cellm) {
      const int pos = ncell + __popcll(cm & lt);
      const CellCom cm_ = d.ccom[n];
      s_cx[pos] = cm_.com[0]; s_cy[pos] = cm_.com[1]; s_cz[pos] = cm_.com[2]; s_cm[pos] = cm_.m;
      s_cmask[pos] = (unsigned short) cellm;
    }
    ncell += __popcll(cm);
    if (anynear) {
      // append this step's near leaves to the lists of the target leaves they belong to; the list
      // lengths are wave-uniform and live in scalar registers (len_d / len_h), the loop is fully unrolled
      const int ent = near_entry(g.first, g.N);
#pragma unroll
      for (int l = 0; l < GH_MAXLEAF; l++) {
        if (l < nl) {
          const bool bh = (hydm >> l) & 1, bd = (dirm >> l) & 1;
          const unsigned long long mh_ = __ballot(bh), md_ = __ballot(bd);
          if (mh_) {
            if (bh) { const int pos = len_h[l] + __popcll(mh_ & lt); if (pos < GH_NHCAP) s_nhyd[l][pos] = ent; }
            len_h[l] += __popcll(mh_);
            maxlen_h = max(maxlen_h, len_h[l]);
          }
          if (md_) {
            if (bd) { const int pos = len_d[l] + __popcll(md_ & lt); if (pos < GH_NDCAP) s_ndir[l][pos] = ent; }
            len_d[l] += __popcll(md_);
            maxlen_d = max(maxlen_d, len_d[l]);
          }
        }
      }
    }
    __syncthreads();
    STAMP_ADD(0, tw0);
  }
  flush_cells();
  flush_near();

  if (act) {
    // GradhSph.cpp:577-578 then GradhSphTree.cpp:596-619
    A.div_v *= ti.invrho;
    A.dudt -= ti.press*A.div_v*ti.invrho*d.f[D_INVOMEGA][i];
    for (int k = 0; k < ND; k++) {
      double a = d.f[D_AX + k][i];
      a += A.a[k];
      a += A.at[k];
      d.f[D_AX + k][i] = a;
      d.f[D_ATX + k][i] += A.at[k];
    }
    d.f[D_GPOT][i] += A.gpot;
    d.f[D_GPOT_HYDRO][i] += A.gpot;
    d.f[D_DUDT][i] += A.dudt;
    d.f[D_DIV_V][i] += A.div_v;
    if (lv) raise_levelneib(d, i, lnmax);
  }
  if (COUNT) {
    const unsigned long long a = wave_sum_u64(act ? n_pairs : 0), b = wave_sum_u64(act ? n_direct : 0), c = wave_sum_u64(act ? n_cells : 0);
    if (lane == 0) { atomicAdd(&stats[ST_PAIRS], a); atomicAdd(&stats[ST_DIRECT], b); atomicAdd(&stats[ST_CELLS], c); }
  }
#ifdef GH_STAMPS
  st_acc[6] = clock64() - st_t0;
  if (lane == 0) for (int k = 0; k < 7; k++) atomicAdd(&stats[ST_COUNT + k], (unsigned long long) st_acc[k]);
#endif
}

static void fill_force_params(gh_ctx *ctx, ForceParams &P)
{
  gh_fill_domain(ctx, P.dom);
  gh_fill_eos(ctx, P.eos);
  P.alpha_visc = ctx->cfg.alpha_visc; P.beta_visc = ctx->cfg.beta_visc; P.alpha_visc_min = ctx->cfg.alpha_visc_min;
  P.avisc = ctx->cfg.avisc; P.acond = ctx->cfg.acond; P.ktab = ctx->ktab;
  if (ctx->cfg.self_gravity && ctx->cfg.avisc == GH_AVISC_MON97MM97) {      // see sph_pair: alpha never leaves alpha_visc_min
    P.avisc = GH_AVISC_MON97; P.alpha_visc = ctx->cfg.alpha_visc_min;
  }
  P.macerror = ctx->cfg.macerror; P.mac = ctx->mac_bootstrap ? GH_MAC_GEOMETRIC : ctx->cfg.gravity_mac;
  P.fastquad = ctx->cfg.multipole == GH_MULTIPOLE_FAST_QUADRUPOLE ? 1 : 0;
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  P.group0 = g0;
  P.stale = ctx->tree_stale ? 1 : 0;
}

// force records of the own particles (k_pack_hydro), then the import of the halo the walks of `phase` need.  In gh_step
// the density pass leaves its "did a walk leave the imported halo" check to this exchange's count block; if some rank
// did miss, all ranks widen the density import, redo those groups and the cells' hmax, and come back here.
int gh_force_halo(gh_ctx *ctx, int phase)
{
  for (;;) {
    hipLaunchKernelGGL(k_pack_hydro, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx));
    int rc = gh_dd_exchange(ctx, phase);
    if (rc != GH_DD_REDO_DENSITY) return rc;
    if ((rc = gh_density_impl(ctx, false, true))) return rc;
    if ((rc = gh_update_hmax_impl(ctx))) return rc;
  }
}

int gh_hydro_forces_impl(gh_ctx *ctx, bool count)
{
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_INVALID, "gh_update_hydro_forces: no tree");
  DevicePtrs d = gh_dev(ctx);
  ForceParams P;
  fill_force_params(ctx, P);
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  const int nblocks = g1 - g0;
  hipStream_t s = ctx->stream;
  { const int rc = gh_force_halo(ctx, GH_HALO_HYDRO); if (rc) return rc; }
  gh_phase_begin(ctx, GH_T_SPH_FORCES);
  if (nblocks > 0) {
#define LAUNCH(ND_, KT_)                                                                                            \
    if (ctx->tree_stale) { \
      if (ctx->cfg.Nlevels > 1) hipLaunchKernelGGL((k_hydro_forces<ND_, false, KT_, true, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
      else hipLaunchKernelGGL((k_hydro_forces<ND_, false, KT_, false, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
    } \
    else if (ctx->cfg.Nlevels > 1) { \
      if (count) hipLaunchKernelGGL((k_hydro_forces<ND_, true, KT_, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
      else hipLaunchKernelGGL((k_hydro_forces<ND_, false, KT_, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
    } \
    else if (count) hipLaunchKernelGGL((k_hydro_forces<ND_, true, KT_, false>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags); \
    else hipLaunchKernelGGL((k_hydro_forces<ND_, false, KT_, false>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags);
    GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  }
  gh_phase_end(ctx, GH_T_SPH_FORCES);
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

// launch of the fused gravity kernel; with only_if != NULL the kernel returns at once unless *only_if is set
int gh_grav_fused_launch(gh_ctx *ctx, bool count, const int *only_if)
{
  DevicePtrs d = gh_dev(ctx);
  ForceParams P;
  fill_force_params(ctx, P);
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  const int nblocks = g1 - g0;
  hipStream_t s = ctx->stream;
  if (nblocks > 0) {
#define LAUNCH(ND_, KT_)                                                                                           \
    if (P.avisc == GH_AVISC_MON97CD2010 && count) hipLaunchKernelGGL((k_grav_forces<ND_, true, KT_, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags, only_if); \
    else if (P.avisc == GH_AVISC_MON97CD2010) hipLaunchKernelGGL((k_grav_forces<ND_, false, KT_, true>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags, only_if); \
    else if (count) hipLaunchKernelGGL((k_grav_forces<ND_, true, KT_>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags, only_if); \
    else hipLaunchKernelGGL((k_grav_forces<ND_, false, KT_>), dim3(nblocks), dim3(64), 0, s, d, P, ctx->d_stats, ctx->d_flags, only_if);
    GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  }
  return GH_OK;
}

int gh_all_forces_impl(gh_ctx *ctx, bool count)
{
  if (!ctx->tree_valid) return gh_fail(ctx, GH_ERR_INVALID, "gh_update_all_forces: no tree");
  // cd2010 with self-gravity: the pair terms need every particle's own alpha (ComputeH updates it): done by the fused kernel,
  // which evaluates monopoles with the geometric MAC only
  const bool cd_grav = ctx->cfg.avisc == GH_AVISC_MON97CD2010;
  if (cd_grav && (ctx->cfg.multipole != GH_MULTIPOLE_MONOPOLE || ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC || ctx->nranks > 1))
    return gh_fail(ctx, GH_ERR_UNSUPPORTED, "time_dependent_avisc = cd2010 with self-gravity: multipole = monopole, gravity_mac = geometric, one rank");
  for (int k = 0; k < ctx->ndim; k++)
    if (ctx->cfg.boundary_lhs[k] != GH_BOUNDARY_OPEN || ctx->cfg.boundary_rhs[k] != GH_BOUNDARY_OPEN)
      return gh_fail(ctx, GH_ERR_UNSUPPORTED, "self-gravity needs open boundaries (periodic gravity = Ewald, out of scope)");
  if ((ctx->cfg.multipole < GH_MULTIPOLE_MONOPOLE || ctx->cfg.multipole > GH_MULTIPOLE_FAST_QUADRUPOLE) ||
      (ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC && ctx->cfg.gravity_mac != GH_MAC_GADGET2 && ctx->cfg.gravity_mac != GH_MAC_EIGENMAC))
    return gh_fail(ctx, GH_ERR_UNSUPPORTED, "built: multipole=monopole|quadrupole|fast_monopole|fast_quadrupole, gravity_mac=geometric|gadget2|eigenmac");
  const bool quad = ctx->cfg.multipole != GH_MULTIPOLE_MONOPOLE || ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC;   // list kernels only
  if ((1 << (ctx->ltot - ctx->lgroup)) > GH_MAXLEAF) return gh_fail(ctx, GH_ERR_INVALID, "group has too many leaves");
  {
    // default: walk + evaluation kernels with the interaction lists in HBM (gravity.hip);
    // GH_GRAV_FUSED=1 selects the single fused kernel (leaves wider than 6 particles: the evaluation kernel takes them in chunks)
    const char *fused = getenv("GH_GRAV_FUSED");
    if (!(fused && fused[0] == '1') && !cd_grav) return gh_grav_lists_impl(ctx, count);
  }
  if (quad) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multipole=quadrupole / gravity_mac=gadget2 need the list kernels (Nleafmax <= 6)");
  { const int rc = gh_force_halo(ctx, GH_HALO_GRAVITY); if (rc) return rc; }
  gh_phase_begin(ctx, GH_T_SPH_FORCES);
  int rc = gh_grav_fused_launch(ctx, count, nullptr);
  gh_phase_end(ctx, GH_T_SPH_FORCES);
  if (rc) return rc;
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}
