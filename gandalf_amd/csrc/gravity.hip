// gravity.hip -- hydro + self-gravity forces as two kernels with the interaction lists in HBM.
//
// Replaces the same reference code as k_grav_forces in forces.hip (GradhSphTree::UpdateAllSphForces and
// everything it calls per leaf cell, reference src/GradhSph/GradhSphTree.cpp:444-657, src/Tree/Tree.cpp:628-735,
// src/Headers/NeighbourManager.h:368-543, src/GradhSph/GradhSph.cpp:474-690, NeighbourSearch.h:350-377).
//
// The reference builds one interaction list per LEAF CELL and then loops particle x list entry.  A
// leaf's list (~1300 cells + ~100 near leaves at theta = 0.5) is far longer than a leaf is wide (4-6
// particles), so the parallel axis of the evaluation is the LIST, not the particles:
//
//   k_grav_walk : one wavefront per group of <= 16 leaves walks the tree once with a 16-bit leaf mask
//                 per stack entry (every (node, leaf) decision is the reference's, taken with the leaf's
//                 own rcell/rmax/hmax) and appends node / leaf ids to the per-leaf lists in HBM
//                 (three lists per leaf: accepted cells, direct-only leaves, leaves with hydro candidates).
//                 With 288 GB of HBM the lists of a 1M-particle run (1.8 GB) are no concern.
//   k_grav_eval : one wavefront per leaf; lane = list entry.  Every lane evaluates its entry against the
//                 leaf's 4-6 particles (target data broadcast from LDS) and the wave reduces the partial
//                 sums at the end: all 64 lanes are busy whatever the masks were.  Hydro candidates are
//                 classified per (particle, candidate); the real SPH neighbours of each particle are
//                 compacted (ballot) into an LDS list and evaluated 64 pairs at a time.
#include "force_common.hpp"
#include <cstdlib>

int gh_grav_fused_launch(gh_ctx *ctx, bool count, const int *only_if);

#define GH_MAXLEAF 16
#define GH_MAXOCC 6          /* max particles per leaf handled by the evaluation kernel (Nleafmax default 6) */
#define GH_SPHCAP 448        /* SPH neighbours of one particle held in LDS (SURVEY: max 393 at 4k Plummer) */

struct GravLists {
  int *cells, *dirl;         // [gtot][cap_*]: node ids / (first | N << 26) leaf entries
  int *gcells;               // [ngroups][cap_g]: cells accepted by EVERY leaf of a group (stored once)
  int *glen;                 // [ngroups]
  int2 *hydl;                // [gtot][cap_h]: (first, count) particle ranges - a leaf, or a whole subtree
  int *len;                  // [gtot][3]
  int *fallback;             // set when a list overflowed: the fused kernel redoes the call
  int cap_c, cap_d, cap_h, cap_g;
#ifdef GH_DEBUG_BLOCKTIME
  double *dbgw, *dbge;       // per-group walk record [ngroups][8], per-leaf eval time [gtot]
#endif
};

// ================================================================================================
// walk                                                           (Tree.cpp:628-735, Tree.h:413-432)
// ================================================================================================
template <int ND, int KT, int MAC>
__global__ __launch_bounds__(64) void k_grav_walk(DevicePtrs d, ForceParams P, GravLists G, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ int s_stack[GH_SCAP];
  __shared__ unsigned short s_smask[GH_SCAP];
  __shared__ int s_qnode[128];                     // nodes waiting for the per-leaf tests, with their leaf masks
  __shared__ unsigned short s_qmask[128];
  __shared__ double s_lrc[GH_MAXLEAF][3], s_lrmax[GH_MAXLEAF], s_lhr[GH_MAXLEAF], s_lamin[GH_MAXLEAF];

  const int lane = threadIdx.x;
  const unsigned long long lt = lanemask_lt();
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int gnode = (1 << d.lgroup) - 1 + q;
  if (d.cN[gnode] == 0) return;
#ifdef GH_DEBUG_BLOCKTIME
  const unsigned long long dbg_t0 = wall_clock64();
  unsigned long long dbg_steps = 0, dbg_perleaf = 0;
#endif
  const int nl = 1 << (d.ltot - d.lgroup);
  const int leaf0 = d.gtot - 1;
  const int leafnode0 = leaf0 + q*nl;

  unsigned int allmask = 0;
  const CellGeo gg = d.cgeo[gnode];
  constexpr bool gadget = MAC == GH_MAC_GADGET2;        // relative MACs of open_cell_for_gravity, Tree.h:413-432
  constexpr bool eigen = MAC == GH_MAC_EIGENMAC;
  double Rg = 0.0, Lm = 0.0, Lr = 0.0, Amin = 0.0, Amax = 0.0;
  {
    double rg = 0.0, lm = 0.0, lr = 0.0, amn = 9.9e20, amx = 0.0;
    if (lane < nl) {
      const CellGeo g = d.cgeo[leafnode0 + lane];
      for (int k = 0; k < 3; k++) s_lrc[lane][k] = g.rcell[k];
      s_lrmax[lane] = g.rmax;
      s_lhr[lane] = K::kernrange*g.hmax;
      const double am = (gadget || eigen) ? d.leaf_amin[q*nl + lane] : 0.0;
      s_lamin[lane] = am;
      if (g.N > 0) {
        double dd = 0.0;
        for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - gg.rcell[k]; dd += dx*dx; }
        rg = sqrt(dd); lm = g.rmax + K::kernrange*g.hmax; lr = g.rmax;
        amn = am; amx = am;
      }
    }
    Rg = wave_max(rg)*(1.0 + 1e-12); Lm = wave_max(lm); Lr = wave_max(lr);
    Amin = wave_min(amn); Amax = wave_max(amx);
  }
  for (int l = 0; l < nl; l++) if (d.cN[leafnode0 + l] > 0) allmask |= 1u << l;
  if (d.levels) {
    // block timesteps: only leaves with an active particle get lists (Tree::ComputeActiveCellList, Tree.cpp:91-115);
    // the evaluation kernel never looks at the others
    unsigned int am = 0;
    for (int l = 0; l < nl; l++) {
      const int f0 = d.cfirst[leafnode0 + l], cn = d.cN[leafnode0 + l];
      bool a = false;
      for (int t = 0; t < cn; t++) a = a || ((int) d.f[D_FLAGS][f0 + t] & 1);
      if (d.leafact) a = a && d.leafact[q*nl + l] > 0;      // extrapolated tree: the reference's stale active-cell list
      if (a) am |= 1u << l;
    }
    allmask &= am;
    if (!allmask) return;
  }
  // list lengths: wave-uniform, in scalar registers (loops over leaves are fully unrolled)
  int len_c[GH_MAXLEAF], len_d[GH_MAXLEAF], len_h[GH_MAXLEAF];
#pragma unroll
  for (int l = 0; l < GH_MAXLEAF; l++) { len_c[l] = 0; len_d[l] = 0; len_h[l] = 0; }
  if (lane == 0) { s_stack[0] = 0; s_smask[0] = (unsigned short) allmask; }
  __syncthreads();
  int top = 1, len_g = 0, qn = 0;
  bool overflow = false;
  // Two phases per round.  A: pop up to 64 nodes and classify each against the WHOLE group - every leaf accepts it
  // (-> the group's shared list), every leaf opens it (-> children pushed), or undecided (-> queued).  B: once 64
  // undecided nodes are queued (or the stack is empty) one full wave takes them through the per-leaf tests.  Most
  // nodes are decided in A (1 test instead of 16), and the per-leaf loop - the cost of this kernel - only ever runs
  // with all lanes busy.
  while (top > 0 || qn > 0) {
    if (top > 0 && qn <= 64) {
      const int p = pop_width(top);
      const int newtop = top - p;
#ifdef GH_DEBUG_BLOCKTIME
      dbg_steps++;
#endif
      int n = 0; unsigned int fm = 0;
      bool acc = false, opn = false, slow = false;
      if (lane < p) {
        n = s_stack[top - 1 - lane];
        fm = s_smask[top - 1 - lane];
        const CellGeo g = d.cgeo[n];
        if (g.N < 0) atomicOr(flags, FLAG_LET_MISS);            // multi-GPU: a remote cell the halo exchange did not import
        const bool isleaf = n >= leaf0;
        const double khr = K::kernrange*g.hmax;
        // lower / upper bound D -/+ Rg on every leaf's distance to the node
        double D2 = 0.0;
        for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - gg.rcell[k]; D2 += dx*dx; }
        const double Ds = sqrt(D2);
        const double Dm = (Ds - Rg)*(1.0 - 1e-12), Dp = (Ds + Rg)*(1.0 + 1e-12);
        const double Tn = g.rmax + fmax(Lm, Lr + khr);
        // gadget2: open if drsqd^2*amin*macerror < rmax^2*m; rm2 is that right-hand side (0 disables the test)
        const double rm2 = gadget ? g.rmax*g.rmax*d.ccom[n].m : 0.0;
        // eigenmac: open if drsqd < cell.mac*macfactor(leaf); the group bound uses the largest factor
        if (g.N <= 0) { }                                       // empty cell (or a remote one that was not imported): nothing below it
        else if (Dm > Tn && Dm*Dm > g.cdistsqd && !(gadget && (Dm*Dm)*(Dm*Dm)*Amin*P.macerror*(1.0 - 1e-12) < rm2) &&
                 !(eigen && !((Dm*Dm)*(1.0 - 1e-12) >= g.mac*Amax))) {
          // every leaf clears the overlap and opening distances: "cell" for all.  Only the complete group goes to the
          // shared list from here; partial masks and single-particle leaves take the per-leaf path
          if (fm == allmask && !(isleaf && g.N == 1)) acc = true; else slow = true;
        }
        // every leaf opens it: the node is inside every leaf's opening distance (geometric MAC: drsqd < cdistsqd, whether
        // through the overlap branch or the MAC branch) and too large to lie inside any leaf's overlap range
        else if (!isleaf && !gadget && !eigen && g.rmax > Lm && Dp*Dp < g.cdistsqd) opn = true;
        else slow = true;
      }
      const unsigned long long gm = __ballot(acc), om = __ballot(opn), qm = __ballot(slow);
      if (gm) {
        const int pos = len_g + __popcll(gm & lt);
        if (acc) { if (pos < G.cap_g) G.gcells[(size_t) q*G.cap_g + pos] = n; else overflow = true; }
        len_g += __popcll(gm);
      }
      __syncthreads();
      if (opn) {
        const int pos = newtop + 2*__popcll(om & lt);
        if (pos + 1 < GH_SCAP) {
          s_stack[pos] = 2*n + 1; s_smask[pos] = (unsigned short) fm;
          s_stack[pos + 1] = 2*n + 2; s_smask[pos + 1] = (unsigned short) fm;
        }
      }
      top = newtop + 2*__popcll(om);
      if (top > GH_SCAP) { overflow = true; top = GH_SCAP; }
      if (slow) { const int pos = qn + __popcll(qm & lt); s_qnode[pos] = n; s_qmask[pos] = (unsigned short) fm; }
      qn += __popcll(qm);
      __syncthreads();
    }
    if (qn > 0 && (qn >= 64 || top == 0)) {
      // (a batch pushes two entries per lane at most: sized to the room left on the stack, never less than 4 lanes)
      const int room = (GH_SCAP - top) >> 1;
      const int b = min(min(qn, 64), max(room, 4));
      unsigned int openm = 0, cellm = 0, hydm = 0, dirm = 0;
      int n = 0; bool isleaf = false;
      CellGeo g;
      g.first = 0; g.N = 0;
      if (lane < b) {
        n = s_qnode[qn - 1 - lane];
        const unsigned int fm = s_qmask[qn - 1 - lane];
        g = d.cgeo[n];
        if (g.N < 0) g.N = 0;
        isleaf = n >= leaf0;
        const double khr = K::kernrange*g.hmax;
        const double rm2 = gadget ? g.rmax*g.rmax*d.ccom[n].m : 0.0;
        for (int l = 0; l < nl; l++) {
          if (!((fm >> l) & 1)) continue;
          double drsqd = 0.0;
          for (int k = 0; k < ND; k++) { const double dx = g.rcell[k] - s_lrc[l][k]; drsqd += dx*dx; }
          const double d1 = g.rmax + s_lrmax[l] + s_lhr[l];
          const double d2 = s_lrmax[l] + g.rmax + khr;
          if (drsqd <= d1*d1 || drsqd <= d2*d2) {                  // overlap -> hydro candidates / open
            if (isleaf) { if (g.N > 0) hydm |= 1u << l; }
            else {
              // the node's bounding sphere lies inside the leaf's overlap range: every descendant leaf
              // (its centre is within g.rmax of this centre) passes the first overlap test, so the walk
              // would list ALL its particles as hydro candidates - emit the whole range, do not open
              const double rr = s_lrmax[l] + s_lhr[l];
              const double dd_ = sqrt(drsqd) + g.rmax;
              // (not with an extrapolated tree: the drifted centres of a node and its descendants no longer nest)
              if (g.N > 0 && !P.stale && dd_ <= rr*(1.0 - 1e-12)) hydm |= 1u << l;
              else openm |= 1u << l;
            }
          }
          else if (g.N == 0) { }
          else if (!(drsqd < g.cdistsqd) && !(gadget && drsqd*drsqd*s_lamin[l]*P.macerror < rm2) &&
                   !(eigen && drsqd < g.mac*s_lamin[l])) {                                          // !open_cell_for_gravity
            if (isleaf && g.N == 1) dirm |= 1u << l;
            else cellm |= 1u << l;
          }
          else {
            if (!isleaf) openm |= 1u << l;
            else dirm |= 1u << l;
          }
        }
      }
      qn -= b;
      // cells that every (non-empty) leaf of the group accepts go once into the group's shared list
      const bool gfull = cellm != 0 && cellm == allmask;
      const unsigned long long gm = __ballot(gfull);
      if (gm) {
        const int pos = len_g + __popcll(gm & lt);
        if (gfull) { if (pos < G.cap_g) G.gcells[(size_t) q*G.cap_g + pos] = n; else overflow = true; }
        len_g += __popcll(gm);
      }
      if (gfull) cellm = 0;
      const unsigned long long om = __ballot(openm != 0);
      const bool anycell = __any(cellm != 0), anynear = __any((hydm | dirm) != 0);
      __syncthreads();
      if (openm) {
        const int pos = top + 2*__popcll(om & lt);
        if (pos + 1 < GH_SCAP) {
          s_stack[pos] = 2*n + 1; s_smask[pos] = (unsigned short) openm;
          s_stack[pos + 1] = 2*n + 2; s_smask[pos + 1] = (unsigned short) openm;
        }
      }
      top = top + 2*__popcll(om);
      if (top > GH_SCAP) { overflow = true; top = GH_SCAP; }
      if (anycell | anynear) {
        const int nent = g.first | (int) ((unsigned int) g.N << 26);
#pragma unroll
        for (int l = 0; l < GH_MAXLEAF; l++) {
          if (l < nl) {
            const size_t leaf = (size_t) (q*nl + l);
            if (anycell) {
              const bool bc = (cellm >> l) & 1;
              const unsigned long long m_ = __ballot(bc);
              if (m_) {
                const int pos = len_c[l] + __popcll(m_ & lt);
                if (bc) { if (pos < G.cap_c) G.cells[leaf*G.cap_c + pos] = n; else overflow = true; }
                len_c[l] += __popcll(m_);
              }
            }
            if (anynear) {
              const bool bh = (hydm >> l) & 1, bd = (dirm >> l) & 1;
              const unsigned long long mh_ = __ballot(bh), md_ = __ballot(bd);
              if (mh_) {
                const int pos = len_h[l] + __popcll(mh_ & lt);
                if (bh) { if (pos < G.cap_h) G.hydl[leaf*G.cap_h + pos] = make_int2(g.first, g.N); else overflow = true; }
                len_h[l] += __popcll(mh_);
              }
              if (md_) {
                const int pos = len_d[l] + __popcll(md_ & lt);
                if (bd) { if (pos < G.cap_d) G.dirl[leaf*G.cap_d + pos] = nent; else overflow = true; }
                len_d[l] += __popcll(md_);
              }
            }
          }
        }
      }
      __syncthreads();
    }
  }
#ifdef GH_DEBUG_BLOCKTIME
  if (lane == 0 && G.dbgw) {
    double *o = G.dbgw + (size_t) q*8;
    o[0] = (double) (wall_clock64() - dbg_t0); o[1] = (double) dbg_steps; o[2] = (double) len_g; o[3] = (double) len_c[0];
    o[4] = (double) len_h[0]; o[5] = (double) len_d[0]; o[6] = Rg; o[7] = gg.hmax;
  }
#endif
  if (__any(overflow) && lane == 0) atomicOr(G.fallback, 1);
  if (lane == 0) G.glen[q] = min(len_g, G.cap_g);
  if (lane == 0) {
#pragma unroll
    for (int l = 0; l < GH_MAXLEAF; l++) {
      if (l < nl) {
        const size_t leaf = (size_t) (q*nl + l);
        G.len[leaf*3 + 0] = min(len_c[l], G.cap_c);
        G.len[leaf*3 + 1] = min(len_d[l], G.cap_d);
        G.len[leaf*3 + 2] = min(len_h[l], G.cap_h);
      }
    }
  }
}

// ================================================================================================
// evaluation                                                      (GradhSphTree.cpp:505-619)
// ================================================================================================
__device__ __forceinline__ double wave_sum_d(double v)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct PMAcc { double at[3], gpot; };              // point-mass partial sums of one target particle

// ---- wave reduction of FOUR values at once with the gfx950 lane-swap instructions ----------------
// v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32 of the second, so
// "swap, then add the two results" halves the lanes of two values in 3 instructions with no select and no
// LDS traffic; v_permlane16_swap does the same on 16-lane rows.  Four values end up one per row, then four
// DPP row rotations finish the sum inside each row: 21 VALU instructions instead of 4 x 18 shuffles+adds.
__device__ __forceinline__ double swap32_add(double a, double b)
{
  const auto r0 = __builtin_amdgcn_permlane32_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
  return __hiloint2double((int) r1[0], (int) r0[0]) + __hiloint2double((int) r1[1], (int) r0[1]);
}
__device__ __forceinline__ double swap16_add(double a, double b)
{
  const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned) __double2loint(a), (unsigned) __double2loint(b), false, false);
  const auto r1 = __builtin_amdgcn_permlane16_swap((unsigned) __double2hiint(a), (unsigned) __double2hiint(b), false, false);
  return __hiloint2double((int) r1[0], (int) r0[0]) + __hiloint2double((int) r1[1], (int) r0[1]);
}
template <int N> __device__ __forceinline__ double row_ror_add(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, false);
  return v + __hiloint2double(hi, lo);
}
// returns, in every lane of row q (= lane >> 4), the wave sum of (v0, v2, v1, v3)[q]
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3)
{
  double e = swap16_add(swap32_add(v0, v1), swap32_add(v2, v3));
  e = row_ror_add<8>(e); e = row_ror_add<4>(e); e = row_ror_add<2>(e); e = row_ror_add<1>(e);
  return e;
}

template <int ND>
__device__ __forceinline__ void point_mass_pm(const TargetI &ti, PMAcc &A, double x, double y, double z, double m)
{
#pragma clang fp contract(fast)
  double dr[3] = {0.0, 0.0, 0.0};
  dr[0] = x - ti.r[0];
  if (ND > 1) dr[1] = y - ti.r[1];
  if (ND > 2) dr[2] = z - ti.r[2];
  double drsqd = dr[0]*dr[0];
  if (ND > 1) drsqd += dr[1]*dr[1];
  if (ND > 2) drsqd += dr[2]*dr[2];
  drsqd += GH_SMALL;
  const double invdrmag = fast_rsqrt1(drsqd);
  const double minvdr3 = m*(invdrmag*invdrmag*invdrmag);
  A.gpot += m*invdrmag;
  for (int k = 0; k < ND; k++) A.at[k] += dr[k]*minvdr3;
}


// Monopole + quadrupole term of one accepted cell on one target (ComputeQuadropole, NeighbourSearch.h:384-460);
// dr = r_p - r_cell as the reference writes it.  One reciprocal square root replaces its division + sqrt.
template <int ND>
__device__ __forceinline__ void point_mass_quad(const TargetI &ti, PMAcc &A, double x, double y, double z, double m, const double *q)
{
#pragma clang fp contract(fast)
  double dr[3] = {0.0, 0.0, 0.0};
  dr[0] = ti.r[0] - x;
  if (ND > 1) dr[1] = ti.r[1] - y;
  if (ND > 2) dr[2] = ti.r[2] - z;
  double drsqd = dr[0]*dr[0];
  if (ND > 1) drsqd += dr[1]*dr[1];
  if (ND > 2) drsqd += dr[2]*dr[2];
  drsqd += GH_SMALL;
  const double invdrmag = fast_rsqrt(drsqd);
  const double invdrsqd = invdrmag*invdrmag;
  const double invdr3 = invdrsqd*invdrmag;
  const double invdr5 = invdrsqd*invdr3;
  double qscalar, qd[3] = {0.0, 0.0, 0.0};
  if (ND == 3) {
    const double qzz = -(q[0] + q[2]);
    qscalar = q[0]*dr[0]*dr[0] + q[2]*dr[1]*dr[1] + qzz*dr[2]*dr[2] + 2.0*(q[1]*dr[0]*dr[1] + q[3]*dr[0]*dr[2] + q[4]*dr[1]*dr[2]);
    qd[0] = q[0]*dr[0] + q[1]*dr[1] + q[3]*dr[2];
    qd[1] = q[1]*dr[0] + q[2]*dr[1] + q[4]*dr[2];
    qd[2] = q[3]*dr[0] + q[4]*dr[1] + qzz*dr[2];
  }
  else if (ND == 2) {
    qscalar = q[0]*dr[0]*dr[0] + q[2]*dr[1]*dr[1] + 2.0*q[1]*dr[0]*dr[1];
    qd[0] = q[0]*dr[0] + q[1]*dr[1];
    qd[1] = q[1]*dr[0] + q[2]*dr[1];
  }
  else {
    qscalar = q[0]*dr[0]*dr[0];
    qd[0] = q[0]*dr[0];
  }
  const double qfactor = 2.5*qscalar*invdr5*invdrsqd;
  const double mfac = m*invdr3 + qfactor;
  for (int k = 0; k < ND; k++) A.at[k] += qd[k]*invdr5 - mfac*dr[k];
  A.gpot += m*invdrmag + 0.5*qscalar*invdr5;
}

template <int ND, bool COUNT, int MAXOCC, int KT, int MP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_grav_eval(DevicePtrs d, ForceParams P, GravLists G, int leaf_begin,
                                                   unsigned long long *stats, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  // M4: keeps 16 workgroups per CU inside 160 KB of LDS; quintic: (3/2)^3 more neighbours per particle
  constexpr int SPHCAP = (KT == 1 || KT == 3) ? 1024 : (MAXOCC <= 4 ? 416 : GH_SPHCAP);
  __shared__ TargetI s_tg[MAXOCC];
  __shared__ int s_sph[MAXOCC][SPHCAP];
  __shared__ RangeRing s_ring;
  __shared__ int s_pre[64];
  __shared__ double s_out[MAXOCC][10];             // a[3], at[3], dudt, div_v, gpot, spare
  __shared__ double s_fm[12];                      // fast monopole: pot, a[3], q[6] at the leaf's COM

  const int lane = threadIdx.x;
  const unsigned long long lt = lanemask_lt();
  if (*G.fallback) {                                    // a list overflowed: the fused kernel does this call
    if ((MP != 0 || P.mac != GH_MAC_GEOMETRIC) && lane == 0) atomicOr(flags, FLAG_ILIST_OVERFLOW);   // ... which has neither quadrupoles nor relative MACs: report it
    return;
  }
  const int gl = leaf_begin + block_to_group(blockIdx.x, gridDim.x);     // leaf index (tree order)
#ifdef GH_DEBUG_BLOCKTIME
  const unsigned long long dbg_t0 = wall_clock64();
#endif
  const int node = (d.gtot - 1) + gl;
  // leaves wider than MAXOCC particles (Nleafmax > 6; the reference's bossbodenheimer.dat ships 8) are evaluated in
  // chunks of MAXOCC targets, one wave per chunk (blockIdx.y), each against the leaf's whole interaction list
  const int first = d.cfirst[node] + (int) blockIdx.y*MAXOCC, Nt = min(MAXOCC, d.cN[node] - (int) blockIdx.y*MAXOCC);
  if (Nt <= 0) return;
  const int occ = d.leafocc;
  // block timesteps (Nlevels > 1, wave-uniform flag): a leaf without active particles has no work
  // (Tree::ComputeActiveCellList, Tree.cpp:91-115); otherwise all of its particles are evaluated, the active ones stored
  const bool lv = d.levels != 0;
  unsigned int actmask = ~0u;
  if (lv) {
    actmask = 0;
    for (int t = 0; t < Nt; t++) if ((int) d.f[D_FLAGS][first + t] & 1) actmask |= 1u << t;
    if (d.leafact && d.leafact[gl] <= 0) actmask = 0;
    if (!actmask) return;
  }

  // slots >= Nt of a partly filled leaf hold a copy of particle 0: the point-mass loops then run unguarded
  // over all MAXOCC slots (independent chains the scheduler can interleave); their sums are never stored
  if (lane < MAXOCC) {
    TargetI t;
    load_target(d, first + (lane < Nt ? lane : 0), ND, t);
    s_tg[lane] = t;
  }
  if (lane < Nt) {
    for (int k = 0; k < 10; k++) s_out[lane][k] = 0.0;
    s_out[lane][8] = (d.f[D_M][first + lane]/d.f[D_H][first + lane])*K::t_wpot0(P.ktab);    // self term, GradhSphTree.cpp:512
  }
  __syncthreads();
  const int lenc = G.len[(size_t) gl*3 + 0], lend = G.len[(size_t) gl*3 + 1], lenh = G.len[(size_t) gl*3 + 2];
  const int *cells = G.cells + (size_t) gl*G.cap_c, *dirl = G.dirl + (size_t) gl*G.cap_d;
  const int nlg = 1 << (d.ltot - d.lgroup);
  const int grp = gl/nlg;
  const int leng = G.glen[grp];
  const int *gcells = G.gcells + (size_t) grp*G.cap_g;
  const int2 *hydl = G.hydl + (size_t) gl*G.cap_h;

  // per-lane partial sums of the point-mass terms, one set per target particle
  PMAcc acc[MAXOCC];
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) { for (int k = 0; k < 3; k++) acc[i].at[k] = 0.0; acc[i].gpot = 0.0; }
  unsigned long long n_cells = 0, n_direct = 0, n_pairs = 0;
  unsigned long long n_lcell = 0, n_ldir = 0, n_lcand = 0;     // what the wave LOADS: list entries / particles, once per leaf

  // ---- accepted cells: monopole terms                            (NeighbourSearch.h:350-377)
  // two-level gather (node id from the list, then the 32-byte COM record from the L2-resident table):
  // ids run two chunks ahead, records one chunk ahead of the arithmetic
  {
    // entries [0, leng) come from the group's shared list, [leng, leng + lenc) from the leaf's own
    const int ltot_ = leng + lenc;
    // all loads are unconditional (clamped index, mass zeroed afterwards): loads inside divergent branches
    // make the compiler wait for vmcnt(0), which would also wait for the prefetches
    auto idraw = [&](int c0) -> int {               // raw id of entry c0 + lane (clamped index)
      const int e = c0 + lane;
      const int ec = e < ltot_ ? e : ltot_ - 1;
      const int *p = ec < leng ? gcells + ec : cells + (ec - leng);
      return *p;
    };
    auto idfix = [&](int raw, int c0) -> int { return c0 + lane < ltot_ ? raw : -1; };
    auto recload = [&](int id, double4 &v) {         // raw load; the caller zeroes the mass of id < 0 entries
      const double4 *c = (const double4*) &d.ccom[id < 0 ? 0 : id];
      v = *c;
    };
    auto ccomp = [&](const double4 &v) {
#pragma unroll
      for (int i = 0; i < MAXOCC; i++) point_mass_pm<ND>(s_tg[i], acc[i], v.x, v.y, v.z, v.w);
    };
    if (MP == 2) {
      // fast monopole (FastMultipoleForces::AddMonopoleContribution / ApplyForcesTaylor, NeighbourSearch.h:561-583,
      // 737-745): every entry is evaluated ONCE, at the leaf's centre of mass - potential, field and field gradient -
      // and the particles get the first-order Taylor expansion.  lane = entry, ten sums per lane, one wave reduction.
      const CellCom lc = d.ccom[node];
      double f_pot = 0.0, f_a[3] = {0.0, 0.0, 0.0}, f_q[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      // same software pipeline as the monopole loop: ids two chunks ahead, records one chunk ahead
      const bool fq = P.fastquad != 0;
      auto fm_term = [&](int id, const double4 &v) {
        #pragma clang fp contract(fast)
        if (fq) {
          // fast_quadrupole: FastMultipoleForces::AddQuadrupoleContribution (NeighbourSearch.h:601-720) on top of the
          // monopole part below; note its dr = rc - cell.r and the softened r^2 (+1e-20), unlike the monopole part
          const CellQuad cq = d.cquad[id < 0 ? 0 : id];
          double e[3] = {0.0, 0.0, 0.0};
          for (int k = 0; k < ND; k++) e[k] = lc.com[k] - (k == 0 ? v.x : (k == 1 ? v.y : v.z));
          double e2 = e[0]*e[0];
          if (ND > 1) e2 += e[1]*e[1];
          if (ND > 2) e2 += e[2]*e[2];
          e2 += GH_SMALL;
          const double im = id < 0 ? 0.0 : fast_rsqrt(e2);
          const double i2 = im*im, i5 = i2*i2*im;
          const double Q0 = cq.q[0], Q1 = cq.q[1], Q2 = cq.q[2], Q3 = cq.q[3], Q4 = cq.q[4], Q5 = -(cq.q[0] + cq.q[2]);
          double qs, qx[3] = {0.0, 0.0, 0.0};
          if (ND == 3) {
            qs = Q0*e[0]*e[0] + Q2*e[1]*e[1] + Q5*e[2]*e[2] + 2.0*(Q1*e[0]*e[1] + Q3*e[0]*e[2] + Q4*e[1]*e[2]);
            qx[0] = (Q0*e[0] + Q1*e[1] + Q3*e[2])*i5; qx[1] = (Q1*e[0] + Q2*e[1] + Q4*e[2])*i5; qx[2] = (Q3*e[0] + Q4*e[1] + Q5*e[2])*i5;
          }
          else if (ND == 2) {
            qs = Q0*e[0]*e[0] + Q2*e[1]*e[1] + 2.0*Q1*e[0]*e[1];
            qx[0] = (Q0*e[0] + Q1*e[1])*i5; qx[1] = (Q1*e[0] + Q2*e[1])*i5;
          }
          else { qs = Q0*e[0]*e[0]; qx[0] = Q0*e[0]*i5; }
          const double qf = 2.5*qs*i5*i2;
          f_pot += 0.5*qs*i5;
          for (int k = 0; k < ND; k++) f_a[k] += qx[k] - qf*e[k];
          for (int k = 0; k < ND; k++) qx[k] *= 5.0*i2;
          f_q[0] += qf*(7.0*e[0]*e[0]*i2 - 1) - (qx[0]*e[0] + qx[0]*e[0] - Q0*i5);
          if (ND > 1) {
            f_q[1] += qf*(7.0*e[0]*e[1]*i2) - (qx[0]*e[1] + qx[1]*e[0] - Q1*i5);
            f_q[2] += qf*(7.0*e[1]*e[1]*i2 - 1) - (qx[1]*e[1] + qx[1]*e[1] - Q2*i5);
          }
          if (ND > 2) {
            f_q[3] += qf*(7.0*e[0]*e[2]*i2) - (qx[0]*e[2] + qx[2]*e[0] - Q3*i5);
            f_q[4] += qf*(7.0*e[1]*e[2]*i2) - (qx[1]*e[2] + qx[2]*e[1] - Q4*i5);
            f_q[5] += qf*(7.0*e[2]*e[2]*i2 - 1) - (qx[2]*e[2] + qx[2]*e[2] - Q5*i5);
          }
        }
        double dr[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < ND; k++) dr[k] = (k == 0 ? v.x : (k == 1 ? v.y : v.z)) - lc.com[k];
        double drsqd = dr[0]*dr[0];
        if (ND > 1) drsqd += dr[1]*dr[1];
        if (ND > 2) drsqd += dr[2]*dr[2];
        const double invdrmag = id < 0 ? 0.0 : fast_rsqrt(drsqd);
        const double invdrsqd = invdrmag*invdrmag;
        double mc = id < 0 ? 0.0 : v.w;
        f_pot += mc*invdrmag;
        mc *= invdrsqd*invdrmag;
        for (int k = 0; k < ND; k++) f_a[k] += mc*dr[k];
        f_q[0] += mc*(3.0*dr[0]*dr[0]*invdrsqd - 1);
        if (ND > 1) { f_q[1] += mc*(3.0*dr[0]*dr[1]*invdrsqd); f_q[2] += mc*(3.0*dr[1]*dr[1]*invdrsqd - 1); }
        if (ND > 2) { f_q[3] += mc*(3.0*dr[2]*dr[0]*invdrsqd); f_q[4] += mc*(3.0*dr[2]*dr[1]*invdrsqd); f_q[5] += mc*(3.0*dr[2]*dr[2]*invdrsqd - 1); }
      };
      if (ltot_ > 0) {
        int id1 = idfix(idraw(0), 0), id2 = idfix(idraw(64), 64);
        double4 vcur, vnext;
        vcur = *((const double4*) &d.ccom[id1 < 0 ? 0 : id1]);
        for (int c0 = 0; c0 < ltot_; c0 += 64) {
          int id3 = idraw(c0 + 128);
          vnext = *((const double4*) &d.ccom[id2 < 0 ? 0 : id2]);
          __builtin_amdgcn_sched_barrier(0);
          fm_term(id1, vcur);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(vnext.x), "+v"(vnext.y), "+v"(vnext.z), "+v"(vnext.w), "+v"(id3));
          vcur = vnext; id1 = id2; id2 = idfix(id3, c0 + 128);
        }
      }
      const double r0 = wave_sum4(f_pot, f_a[0], f_a[1], f_a[2]);          // rows: pot, a1, a0, a2
      const double r1 = wave_sum4(f_q[0], f_q[1], f_q[2], f_q[3]);          // rows: q0, q2, q1, q3
      const double r2 = wave_sum4(f_q[4], f_q[5], 0.0, 0.0);                // rows: q4, -, q5, -
      if ((lane & 15) == 0) {
        const int qd = lane >> 4;
        s_fm[qd == 0 ? 0 : (qd == 1 ? 2 : (qd == 2 ? 1 : 3))] = r0;       // s_fm[0..3] = pot, a0, a1, a2
        s_fm[4 + (qd == 0 ? 0 : (qd == 1 ? 2 : (qd == 2 ? 1 : 3)))] = r1;  // s_fm[4..7] = q0..q3
        if (qd == 0) s_fm[8] = r2;
        if (qd == 2) s_fm[9] = r2;
      }
      __syncthreads();
      if (lane < Nt) {
        const TargetI &tt = s_tg[lane];
        double dr[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < ND; k++) dr[k] = tt.r[k] - lc.com[k];
        const double *q = s_fm + 4;
        double a3[3] = {0.0, 0.0, 0.0};
        if (ND == 3) {
          a3[0] = s_fm[1] + q[0]*dr[0] + q[1]*dr[1] + q[3]*dr[2];
          a3[1] = s_fm[2] + q[1]*dr[0] + q[2]*dr[1] + q[4]*dr[2];
          a3[2] = s_fm[3] + q[3]*dr[0] + q[4]*dr[1] + q[5]*dr[2];
        }
        else if (ND == 2) { a3[0] = s_fm[1] + q[0]*dr[0] + q[1]*dr[1]; a3[1] = s_fm[2] + q[1]*dr[0] + q[2]*dr[1]; }
        else a3[0] = s_fm[1] + q[0]*dr[0];
        double gp = s_fm[0];
        for (int k = 0; k < ND; k++) gp += s_fm[1 + k]*dr[k];                // dphi == ac
        for (int k = 0; k < ND; k++) s_out[lane][3 + k] += a3[k];
        s_out[lane][8] += gp;
      }
      __syncthreads();
    }
    else if (MP == 1) {
      // quadrupole moments: a second 40-byte gather per entry.  Same software pipeline as the monopole loop below:
      // ids two chunks ahead, both records one chunk ahead of the arithmetic, pinned with the asm statement.
      if (ltot_ > 0) {
        auto qload = [&](int id, double4 &v, double *q) {
          const int ic = id < 0 ? 0 : id;
          v = *((const double4*) &d.ccom[ic]);
          const double4 qa = *((const double4*) &d.cquad[ic]);
          const double qb = d.cquad[ic].q[4];
          q[0] = qa.x; q[1] = qa.y; q[2] = qa.z; q[3] = qa.w; q[4] = qb;
        };
        int id1 = idfix(idraw(0), 0), id2 = idfix(idraw(64), 64);
        double4 vcur, vnext;
        double qcur[5], qnext[5];
        qload(id1, vcur, qcur);
        if (id1 < 0) { vcur.w = 0.0; for (int k = 0; k < 5; k++) qcur[k] = 0.0; }
        for (int c0 = 0; c0 < ltot_; c0 += 64) {
          int id3 = idraw(c0 + 128);
          qload(id2, vnext, qnext);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MAXOCC; i++) point_mass_quad<ND>(s_tg[i], acc[i], vcur.x, vcur.y, vcur.z, vcur.w, qcur);
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" : "+v"(vnext.x), "+v"(vnext.y), "+v"(vnext.z), "+v"(vnext.w), "+v"(id3),
                            "+v"(qnext[0]), "+v"(qnext[1]), "+v"(qnext[2]), "+v"(qnext[3]), "+v"(qnext[4]));
          if (id2 < 0) { vnext.w = 0.0; for (int k = 0; k < 5; k++) qnext[k] = 0.0; }
          vcur = vnext;
          for (int k = 0; k < 5; k++) qcur[k] = qnext[k];
          id2 = idfix(id3, c0 + 128);
        }
      }
    }
    else if (ltot_ > 0) {
      int id1 = idfix(idraw(0), 0), id2 = idfix(idraw(64), 64);
      double4 vcur, vnext;
      recload(id1, vcur);
      vcur.w = id1 < 0 ? 0.0 : vcur.w;
      for (int c0 = 0; c0 < ltot_; c0 += 64) {
        int id3 = idraw(c0 + 128);
        recload(id2, vnext);
        __builtin_amdgcn_sched_barrier(0);        // keep the loads above the arithmetic they overlap with
        ccomp(vcur);
        __builtin_amdgcn_sched_barrier(0);
        // pin the prefetched record to this iteration: without it the optimiser moves the x/y/z load to the
        // top of the next iteration (fewer live registers) and the L2 latency is back on the critical path
        asm volatile("" : "+v"(vnext.x), "+v"(vnext.y), "+v"(vnext.z), "+v"(vnext.w), "+v"(id3));
        vnext.w = id2 < 0 ? 0.0 : vnext.w;
        vcur = vnext; id2 = idfix(id3, c0 + 128);
      }
    }
    if (COUNT) { n_cells += (unsigned long long) ltot_*Nt; n_lcell += (unsigned long long) ltot_; }      // counted once per wave below
  }
  // ---- direct-only leaves: Newtonian particle terms              (GradhSph.cpp:671-686)
  for (int c0 = 0; c0 < lend; c0 += 64) {
    const int e = c0 + lane;
    const int ent = dirl[e < lend ? e : lend - 1];
    const int pf = ent & 0x3ffffff, pn = e < lend ? (int) ((unsigned int) ent >> 26) : 0;
    // particle k of the entry's leaf; clamped unconditional loads, one ahead of the arithmetic
    double4 v = d.posm[pf];
    int kmax = pn;
    for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, __shfl_xor(kmax, off, 64));
    for (int k = 0; k < kmax; k++) {
      double4 vn = d.posm[pf + (k + 1 < pn ? k + 1 : 0)];
      __builtin_amdgcn_sched_barrier(0);
      const double mk = k < pn ? v.w : 0.0;
#pragma unroll
      for (int i = 0; i < MAXOCC; i++) point_mass_pm<ND>(s_tg[i], acc[i], v.x, v.y, v.z, mk);
      if (COUNT) { n_direct += (k < pn) ? Nt : 0; n_ldir += (k < pn) ? 1 : 0; }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(vn.x), "+v"(vn.y), "+v"(vn.z), "+v"(vn.w));
      v = vn;
    }
  }
  // ---- leaves with hydro candidates: classify every (particle, candidate); direct ones at once, SPH
  //      neighbours compacted per particle into LDS              (NeighbourManager.h:521-533)
  int nsph[MAXOCC];
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) nsph[i] = 0;
  // extrapolated tree: NeighbourManager::_EndSearch's list filter with the leaf's drifted centre (NeighbourManager.h:440-455);
  // a hydro candidate that fails it is demoted to the direct list - here: never classified as an SPH neighbour
  const CellGeo lgeo = d.cgeo[node];
  auto hyd_process = [&](bool valid, int j, const double4 &q0, double hr2) {
    bool keepleaf = true;
    if (P.stale) {
      double dc = q0.x - lgeo.rcell[0], d2c = dc*dc;
      if (ND > 1) { dc = q0.y - lgeo.rcell[1]; d2c += dc*dc; }
      if (ND > 2) { dc = q0.z - lgeo.rcell[2]; d2c += dc*dc; }
      const double h1 = lgeo.rmax + K::kernrange*lgeo.hmax, h2 = lgeo.rmax + sqrt(hr2);
      keepleaf = d2c < h1*h1 || d2c < h2*h2;
    }
#pragma unroll
    for (int i = 0; i < MAXOCC; i++) {
      if (i < Nt) {
        const TargetI &ti = s_tg[i];
        double dr[3] = {0.0, 0.0, 0.0};
        dr[0] = q0.x - ti.r[0];
        if (ND > 1) dr[1] = q0.y - ti.r[1];
        if (ND > 2) dr[2] = q0.z - ti.r[2];
        double r2 = dr[0]*dr[0];
        if (ND > 1) r2 += dr[1]*dr[1];
        if (ND > 2) r2 += dr[2]*dr[2];
        const bool sph = valid && keepleaf && !(r2 >= ti.hr2 && r2 >= hr2);
        const unsigned long long sm = __ballot(sph);
        if (sm) {
          const int pos = nsph[i] + __popcll(sm & lt);
          if (sph && pos < SPHCAP) s_sph[i][pos] = j;
          nsph[i] += __popcll(sm);
        }
        {
#pragma clang fp contract(fast)
          const double mj = sph ? 0.0 : q0.w;
          const double invdrmag = fast_rsqrt1(r2 + GH_SMALL);
          const double minvdr3 = mj*(invdrmag*invdrmag*invdrmag);
          for (int kk = 0; kk < ND; kk++) acc[i].at[kk] += dr[kk]*minvdr3;
          acc[i].gpot += mj*invdrmag;
        }
        if (COUNT) n_direct += (valid && !sph) ? 1 : 0;
      }
    }
  };
  auto hyd_tile = [&](bool valid, int j, int) {
    // position and mass from the 32-byte pack (consecutive slots of a range share cache lines); the support radius from
    // its SoA array - 8 coalesced bytes per candidate instead of one word out of every 128-byte neighbour record (the halo
    // import fills the array for imported particles too, k_let_unpack)
    const int jc = valid ? j : 0;
    double4 q0 = d.posm[jc];
    double hr2 = d.f[D_HRANGESQD][jc];                 // (from the record: 3.44 ms per launch at 1M; from the array: 3.25)
    if (!valid) { q0.x = 1e30; q0.y = 1e30; q0.z = 1e30; q0.w = 0.0; hr2 = 0.0; }
    if (COUNT) n_lcand += valid ? 1 : 0;
    hyd_process(valid, j, q0, hr2);
  };
  // Expansion of the (first, count) ranges into 64-candidate tiles.  Usual case (<= GH_RBCAP ranges, < 32*(GH_RBCAP-1)
  // candidates): one pass over the list leaves, in the ring's storage, the compacted range table (first - slot offset)
  // and a bit array with the start slot of every range; a tile then finds its ranges with one 64-bit mask read and a
  // popcount per lane instead of a prefix scan and a binary search per tile.
  bool hyd_fast = lenh <= GH_RBCAP;
  int hyd_slots = 0;
  if (hyd_fast) {
    int *s_radj = s_ring.first;                       // [k]: first particle of range k minus its first slot
    unsigned int *s_bits = (unsigned int*) s_ring.cnt; // bit s: slot s starts a range
    for (int w = lane; w < GH_RBCAP; w += 64) s_bits[w] = 0u;
    __syncthreads();
    int kc = 0;
    for (int c0 = 0; c0 < lenh; c0 += 64) {
      const int e = c0 + lane;
      int2 ent = make_int2(0, 0);
      if (e < lenh) ent = hydl[e];
      const int c = ent.y > 0 ? ent.y : 0;
      int inc = c;
      for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
      const int tot = __shfl(inc, 63, 64);
      if (hyd_slots + tot > 32*(GH_RBCAP - 2)) { hyd_fast = false; break; }
      const int excl = hyd_slots + inc - c;
      const unsigned long long vm = __ballot(c > 0);
      if (c > 0) {
        s_radj[kc + __popcll(vm & lt)] = ent.x - excl;
        atomicOr(&s_bits[excl >> 5], 1u << (excl & 31));
      }
      kc += __popcll(vm);
      hyd_slots += tot;
    }
    __syncthreads();
  }
  if (hyd_fast) {
    const int *s_radj = s_ring.first;
    const unsigned int *s_bits = (const unsigned int*) s_ring.cnt;
    const unsigned long long le = lt | (1ull << lane);
    int cbase = -1;                                   // ranges started before this tile, minus one
    for (int t0 = 0; t0 < hyd_slots; t0 += 64) {
      const unsigned int lo = s_bits[t0 >> 5], hi = s_bits[(t0 >> 5) + 1];
      const unsigned long long m = (unsigned long long) lo | ((unsigned long long) hi << 32);
      const bool valid = t0 + lane < hyd_slots;
      const int idx = cbase + __popcll(m & le);
      const int j = valid ? s_radj[idx] + t0 + lane : 0;
      hyd_tile(valid, j, 0);
      cbase += __popcll(m);
    }
    __syncthreads();
  }
  else {
    __syncthreads();
    RangeState R; R.nrb = 0; R.nslots = 0;
    for (int c0 = 0; c0 < lenh; c0 += 64) {
      const int e = c0 + lane;
      int2 ent = make_int2(0, 0);
      if (e < lenh) ent = hydl[e];
      const unsigned long long vm = __ballot(ent.y > 0);
      if (ent.y > 0) { const int pos = R.nrb + __popcll(vm & lt); s_ring.first[pos] = ent.x; s_ring.cnt[pos] = ent.y; s_ring.tag[pos] = 0; }
      R.nrb += __popcll(vm);
      R.nslots += wave_sum_i(ent.y);
      range_drain_raw(s_ring.first, s_ring.cnt, s_ring.tag, s_pre, R, false, hyd_tile);
    }
    range_drain_raw(s_ring.first, s_ring.cnt, s_ring.tag, s_pre, R, true, hyd_tile);
  }
  // targets with more than SPHCAP SPH neighbours (a halo particle whose kernel covers the core has O(N)): their
  // list is dropped and they are redone by streaming, below
  unsigned int ovfmask = 0;
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) if (nsph[i] > SPHCAP) ovfmask |= 1u << i;      // nsph counts dropped entries too
  __syncthreads();
  // the point-mass sums are complete: reduce them now (rows: at0, at2, at1, gpot) so that their 8 registers per
  // target are free during the pair loops, which are the register-pressure peak of the kernel
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) {
    if (i < Nt) {
      const double e = wave_sum4(acc[i].at[0], acc[i].at[1], acc[i].at[2], acc[i].gpot);
      if ((lane & 15) == 0) { const int q = lane >> 4; s_out[i][q == 0 ? 3 : (q == 1 ? 5 : (q == 2 ? 4 : 8))] += e; }
    }
  }
  double gps[MAXOCC];                                   // SPH part of the potential, reduced together at the end
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) gps[i] = 0.0;
  // ---- SPH pairs, 64 at a time per target particle               (GradhSph.cpp:474-585)
#pragma unroll
  for (int i = 0; i < MAXOCC; i++) {
    if (i < Nt) {
      const TargetI ti = s_tg[i];
      const int ns = ((ovfmask >> i) & 1u) ? 0 : min(nsph[i], SPHCAP);
      Accum A;
      for (int k = 0; k < 3; k++) { A.a[k] = 0.0; A.at[k] = 0.0; }
      A.dudt = 0.0; A.div_v = 0.0; A.gpot = 0.0;
      const int mylevel = lv ? (int) d.f[D_LEVEL][first + i] : 0;
      int lnm = 0;
      for (int c0 = 0; c0 < ns; c0 += 64) {
        if (c0 + lane < ns) {
          const int j = s_sph[i][c0 + lane];
          const double4 *r = d.hrec + 4*(size_t) j;
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
          sph_pair<ND, true, KT>(P, ti, A, nb, dr, r2);
          if (lv && ((actmask >> i) & 1u)) { lnm = max(lnm, (int) d.f[D_LEVEL][j]); raise_levelneib(d, j, mylevel); }
        }
      }
      if (lv && ((actmask >> i) & 1u) && ns > 0) {         // GradhSph.cpp:569, GradhSphTree.cpp:619
        for (int off = 32; off > 0; off >>= 1) lnm = max(lnm, __shfl_xor(lnm, off));
        if (lane == 0) raise_levelneib(d, first + i, lnm);
      }
      if (COUNT) n_pairs += (unsigned long long) ns;
      // reduce this particle's sums over the wave, four values per pass (rows hold values 0,2,1,3);
      // the potential is reduced for all particles together after the loop
      gps[i] = A.gpot;
      const double e0 = wave_sum4(A.a[0], A.a[1], A.a[2], A.dudt);
      const double e1 = wave_sum4(A.at[0], A.at[1], A.at[2], A.div_v);
      if ((lane & 15) == 0) {
        const int q = lane >> 4;                       // row -> value: a0, a2, a1, dudt | at0, at2, at1, div_v
        const int k0 = q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : 6));
        const int k1 = q == 0 ? 3 : (q == 1 ? 5 : (q == 2 ? 4 : 7));
        s_out[i][k0] += e0;
        s_out[i][k1] += e1;
      }
    }
  }
  {
    const double g0 = wave_sum4(gps[0], gps[MAXOCC > 1 ? 1 : 0], gps[MAXOCC > 2 ? 2 : 0], gps[MAXOCC > 3 ? 3 : 0]);
    if ((lane & 15) == 0) { const int q = lane >> 4; const int t = q == 0 ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : 3)); if (t < Nt) s_out[t][8] += g0; }
    if (MAXOCC > 4 && Nt > 4) {
      const double g1 = wave_sum4(gps[MAXOCC > 4 ? 4 : 0], gps[MAXOCC > 5 ? 5 : 0], 0.0, 0.0);
      if (lane == 0) s_out[4][8] += g1;
      if (lane == 32 && Nt > 5) s_out[5][8] += g1;
    }
  }
  // ---- rare: SPH pairs of overflowed targets, streamed straight from the candidate ranges (no list)
  if (ovfmask) {
#pragma nounroll
    for (int i = 0; i < Nt; i++) {
      if (!((ovfmask >> i) & 1u)) continue;
      const TargetI ti = s_tg[i];
      Accum A;
      for (int k = 0; k < 3; k++) { A.a[k] = 0.0; A.at[k] = 0.0; }
      A.dudt = 0.0; A.div_v = 0.0; A.gpot = 0.0;
      unsigned long long npair = 0;
      const int mylevel = lv ? (int) d.f[D_LEVEL][first + i] : 0;
      int lnm = 0;
      auto slow_tile = [&](bool valid, int j, int) {
        if (valid) {
          const double4 *r = d.hrec + 4*(size_t) j;
          const double4 q0 = r[0], q1 = r[1];
          double dr[3] = {0.0, 0.0, 0.0};
          dr[0] = q0.x - ti.r[0];
          if (ND > 1) dr[1] = q0.y - ti.r[1];
          if (ND > 2) dr[2] = q0.z - ti.r[2];
          double r2 = dr[0]*dr[0];
          if (ND > 1) r2 += dr[1]*dr[1];
          if (ND > 2) r2 += dr[2]*dr[2];
          if (!(r2 >= ti.hr2 && r2 >= q1.w)) {
            const double4 q2 = r[2], q3 = r[3];
            Neib nb;
            nb.x = q0.x; nb.y = q0.y; nb.z = q0.z; nb.m = q0.w; nb.vx = q1.x; nb.vy = q1.y; nb.vz = q1.z; nb.hr2 = q1.w;
            nb.invh = q2.x; nb.hfac = q2.y; nb.pfac = q2.z; nb.invrho = q2.w; nb.sound = q3.x; nb.zeta = q3.y; nb.u = q3.z; nb.press = q3.w;
            sph_pair<ND, true, KT>(P, ti, A, nb, dr, r2);
            npair++;
            if (lv && ((actmask >> i) & 1u)) { lnm = max(lnm, (int) d.f[D_LEVEL][j]); raise_levelneib(d, j, mylevel); }
          }
        }
      };
      RangeState R; R.nrb = 0; R.nslots = 0;
      for (int c0 = 0; c0 < lenh; c0 += 64) {
        const int e = c0 + lane;
        int2 ent = make_int2(0, 0);
        if (e < lenh) ent = hydl[e];
        const unsigned long long vm = __ballot(ent.y > 0);
        if (ent.y > 0) { const int pos = R.nrb + __popcll(vm & lt); s_ring.first[pos] = ent.x; s_ring.cnt[pos] = ent.y; s_ring.tag[pos] = 0; }
        R.nrb += __popcll(vm);
        R.nslots += wave_sum_i(ent.y);
        range_drain_raw(s_ring.first, s_ring.cnt, s_ring.tag, s_pre, R, false, slow_tile);
      }
      range_drain_raw(s_ring.first, s_ring.cnt, s_ring.tag, s_pre, R, true, slow_tile);
      if (COUNT) n_pairs += wave_sum_u64(npair);
      if (lv && ((actmask >> i) & 1u)) {
        for (int off = 32; off > 0; off >>= 1) lnm = max(lnm, __shfl_xor(lnm, off));
        if (lane == 0) raise_levelneib(d, first + i, lnm);
      }
      double red[9];
      for (int k = 0; k < 3; k++) { red[k] = wave_sum_d(A.a[k]); red[3 + k] = wave_sum_d(A.at[k]); }
      red[6] = wave_sum_d(A.dudt); red[7] = wave_sum_d(A.div_v); red[8] = wave_sum_d(A.gpot);
      if (lane == 0) for (int k = 0; k < 9; k++) s_out[i][k] += red[k];
    }
  }
  __syncthreads();
  if (lane < Nt && ((actmask >> lane) & 1u)) {
    // GradhSph.cpp:577-578 then GradhSphTree.cpp:596-619
    const int i = first + lane;
    const TargetI &ti = s_tg[lane];
    double div_v = s_out[lane][7], dudt = s_out[lane][6];
    div_v *= ti.invrho;
    dudt -= ti.press*div_v*ti.invrho*d.f[D_INVOMEGA][i];
    for (int k = 0; k < ND; k++) {
      double a = d.f[D_AX + k][i];
      a += s_out[lane][k];
      a += s_out[lane][3 + k];
      d.f[D_AX + k][i] = a;
      d.f[D_ATX + k][i] += s_out[lane][3 + k];
    }
    d.f[D_GPOT][i] += s_out[lane][8];
    d.f[D_GPOT_HYDRO][i] += s_out[lane][8];
    d.f[D_DUDT][i] += dudt;
    d.f[D_DIV_V][i] += div_v;
  }
#ifdef GH_DEBUG_BLOCKTIME
  if (lane == 0 && G.dbge) G.dbge[gl] = (double) (wall_clock64() - dbg_t0);
#endif
  if (COUNT && lane == 0) {
    atomicAdd(&stats[ST_CELLS], n_cells);
    atomicAdd(&stats[ST_PAIRS], n_pairs);
  }
  if (COUNT) {
    const unsigned long long b = wave_sum_u64(n_direct), c = wave_sum_u64(n_ldir), e = wave_sum_u64(n_lcand);
    if (lane == 0) { atomicAdd(&stats[ST_DIRECT], b); atomicAdd(&stats[ST_ITER], n_lcell); atomicAdd(&stats[ST_RETRY], c); atomicAdd(&stats[ST_CAND], e); }
  }
}

// ================================================================================================
// host
// ================================================================================================
// Headroom of the per-leaf interaction lists.  A list that overflows makes the evaluation kernel stand back; the fused
// kernel (forces.hip) then does the call - but it has neither quadrupoles nor relative MACs, where an overflow is an error
// (GH_ERR_CAPACITY).  So the longest list of every kind is looked at whenever the host synchronises anyway
// (gh_sync_collect): one that has used more than half its capacity gets twice the room from the next pass on (up to 8 x:
// 288 GB of HBM hold it), long before a run drifts into an overflow.
__global__ __launch_bounds__(256) void k_list_max(const int *len, int nleaf, int *out)
{
  int m[3] = {0, 0, 0};
  for (int l = blockIdx.x*blockDim.x + threadIdx.x; l < nleaf; l += gridDim.x*blockDim.x)
    for (int k = 0; k < 3; k++) m[k] = max(m[k], len[(size_t) l*3 + k]);
  for (int k = 0; k < 3; k++) {
    for (int off = 32; off > 0; off >>= 1) m[k] = max(m[k], __shfl_xor(m[k], off, 64));
    if ((threadIdx.x & 63) == 0 && m[k] > 0) atomicMax(&out[k], m[k]);
  }
}
int gh_grav_list_headroom(gh_ctx *ctx)
{
  if (!ctx->gl_len || ctx->glist_leaves == 0 || ctx->glist_cap[0] == 0 || getenv("GH_GRAV_CAPS")) return GH_OK;
  if (ctx->Nsteps - ctx->glist_checked < 8 && ctx->glist_checked >= 0) return GH_OK;      // lists grow slowly: every eighth step is often enough
  ctx->glist_checked = ctx->Nsteps;
  int *out = ctx->gl_len + ctx->glist_leaves*3 + 1;           // three spare words behind the fallback flag
  GH_CHECK(ctx, hipMemsetAsync(out, 0, 3*sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_list_max, dim3(64), dim3(256), 0, ctx->stream, ctx->gl_len, (int) ctx->glist_leaves, out);
  int mx[3] = {0, 0, 0};
  GH_CHECK(ctx, hipMemcpyAsync(mx, out, sizeof(mx), hipMemcpyDeviceToHost, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 3; k++) {
    ctx->glist_max[k] = mx[k];
    if (2*mx[k] > ctx->glist_cap[k] && ctx->glist_mul[k] < 8) ctx->glist_mul[k] *= 2;
  }
  if (getenv("GH_GRAV_DEBUG"))
    fprintf(stderr, "[lists] step %d longest %d %d %d capacity %d %d %d next x%d x%d x%d\n", ctx->Nsteps, mx[0], mx[1], mx[2],
            ctx->glist_cap[0], ctx->glist_cap[1], ctx->glist_cap[2], ctx->glist_mul[0], ctx->glist_mul[1], ctx->glist_mul[2]);
  return GH_OK;
}

int gh_grav_lists_impl(gh_ctx *ctx, bool count)
{
  // capacities per leaf: accepted cells, direct-only leaves, hydro-candidate leaves
  int cap_c = 4096, cap_d = 256, cap_h = 1024;
  // small trees: a strict relative MAC (gadget2, small macerror) degenerates towards a direct sum, so let a leaf
  // list every other leaf while that is cheap (<= 8192 leaves x 2048 entries x 4 B = 64 MB)
  if (ctx->gtot <= 8192) cap_d = std::max(256, std::min(ctx->gtot, 2048));
  const int cap_g = 4096;
  if (const char *e = getenv("GH_GRAV_CAPS0")) {      // test hook: starting capacities (the headroom rule below still applies)
    int a = 0, b = 0, c = 0;
    if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { cap_c = a; cap_d = b; cap_h = c; }
  }
  // lists that came within a factor two of their capacity at the last look (gh_grav_list_headroom) get twice the room
  cap_c *= ctx->glist_mul[0]; cap_d *= ctx->glist_mul[1]; cap_h *= ctx->glist_mul[2];
  if (const char *e = getenv("GH_GRAV_CAPS")) {       // test hook: tiny capacities force the overflow fallback
    int a = 0, b = 0, c = 0;
    if (sscanf(e, "%d,%d,%d", &a, &b, &c) == 3 && a > 0 && b > 0 && c > 0) { cap_c = a; cap_d = b; cap_h = c; }
  }
  ctx->glist_cap[0] = cap_c; ctx->glist_cap[1] = cap_d; ctx->glist_cap[2] = cap_h;
  const size_t nleaf = (size_t) ctx->gtot;
  if (ctx->glist_leaves != nleaf || ctx->glist_caps != cap_c + 7*cap_d + 31*cap_h) {
    for (void *p : {(void*) ctx->gl_cells, (void*) ctx->gl_dirl, (void*) ctx->gl_hydl, (void*) ctx->gl_len}) if (p) (void) hipFree(p);
    ctx->gl_cells = ctx->gl_dirl = ctx->gl_hydl = ctx->gl_len = nullptr;
    ctx->glist_leaves = 0;
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_cells, sizeof(int)*nleaf*cap_c));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_dirl, sizeof(int)*nleaf*cap_d));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_hydl, sizeof(int2)*nleaf*cap_h));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_len, sizeof(int)*(nleaf*3 + 4)));
    if (ctx->gl_gcells) (void) hipFree(ctx->gl_gcells);
    if (ctx->gl_glen) (void) hipFree(ctx->gl_glen);
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_gcells, sizeof(int)*(size_t) ctx->ngroups*cap_g));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->gl_glen, sizeof(int)*(size_t) ctx->ngroups));
    ctx->glist_leaves = nleaf;
    ctx->glist_caps = cap_c + 7*cap_d + 31*cap_h;
  }
  GravLists G;
  G.cells = ctx->gl_cells; G.dirl = ctx->gl_dirl; G.hydl = (int2*) ctx->gl_hydl; G.len = ctx->gl_len;
  G.fallback = ctx->gl_len + nleaf*3;
  GH_CHECK(ctx, hipMemsetAsync(G.fallback, 0, sizeof(int), ctx->stream));
  G.cap_c = cap_c; G.cap_d = cap_d; G.cap_h = cap_h; G.cap_g = cap_g;
  G.gcells = ctx->gl_gcells; G.glen = ctx->gl_glen;
#ifdef GH_DEBUG_BLOCKTIME
  static double *dbgw = nullptr, *dbge = nullptr;
  if (!dbgw) { (void) hipMalloc((void**) &dbgw, sizeof(double)*8*(size_t) ctx->ngroups); (void) hipMalloc((void**) &dbge, sizeof(double)*(size_t) ctx->gtot); }
  (void) hipMemset(dbgw, 0, sizeof(double)*8*(size_t) ctx->ngroups); (void) hipMemset(dbge, 0, sizeof(double)*(size_t) ctx->gtot);
  G.dbgw = dbgw; G.dbge = dbge;
#endif
  DevicePtrs d = gh_dev(ctx);
  ForceParams P;
  gh_fill_domain(ctx, P.dom);
  gh_fill_eos(ctx, P.eos);
  P.alpha_visc = ctx->cfg.alpha_visc; P.beta_visc = ctx->cfg.beta_visc; P.alpha_visc_min = ctx->cfg.alpha_visc_min;
  P.avisc = ctx->cfg.avisc; P.acond = ctx->cfg.acond; P.ktab = ctx->ktab;
  if (ctx->cfg.avisc == GH_AVISC_MON97MM97) { P.avisc = GH_AVISC_MON97; P.alpha_visc = ctx->cfg.alpha_visc_min; }   // see sph_pair
  P.macerror = ctx->cfg.macerror; P.mac = ctx->mac_bootstrap ? GH_MAC_GEOMETRIC : ctx->cfg.gravity_mac;
  P.fastquad = ctx->cfg.multipole == GH_MULTIPOLE_FAST_QUADRUPOLE ? 1 : 0;
  P.stale = ctx->tree_stale ? 1 : 0;
  const int mpole = ctx->cfg.multipole;
  const bool lists_only = mpole != GH_MULTIPOLE_MONOPOLE || ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC;   // the fused fallback has neither
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  P.group0 = g0;
  const int ngroups = g1 - g0;
  const int nl = 1 << (ctx->ltot - ctx->lgroup);
  hipStream_t s = ctx->stream;
  { const int rc = gh_force_halo(ctx, GH_HALO_GRAVITY); if (rc) return rc; }
  gh_phase_begin(ctx, GH_T_GRAV_WALK);
  if (ngroups > 0) {
#define LAUNCH(ND_, KT_) \
    if (P.mac == GH_MAC_GADGET2) hipLaunchKernelGGL((k_grav_walk<ND_, KT_, GH_MAC_GADGET2>), dim3(ngroups), dim3(64), 0, s, d, P, G, ctx->d_flags); \
    else if (P.mac == GH_MAC_EIGENMAC) hipLaunchKernelGGL((k_grav_walk<ND_, KT_, GH_MAC_EIGENMAC>), dim3(ngroups), dim3(64), 0, s, d, P, G, ctx->d_flags); \
    else hipLaunchKernelGGL((k_grav_walk<ND_, KT_, GH_MAC_GEOMETRIC>), dim3(ngroups), dim3(64), 0, s, d, P, G, ctx->d_flags);
    GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  }
  gh_phase_end(ctx, GH_T_GRAV_WALK);
  gh_phase_begin(ctx, GH_T_SPH_FORCES);
  const int nchunk = cdiv(ctx->leafocc, ctx->leafocc <= 4 ? 4 : GH_MAXOCC);      // waves per leaf: 1 unless Nleafmax > 6
  if (ngroups > 0) {
#define LAUNCHM(ND_, KT_, MP_)                                                                                   \
    if (ctx->leafocc <= 4) { \
      if (count) hipLaunchKernelGGL((k_grav_eval<ND_, true, 4, KT_, MP_>), dim3(ngroups*nl, nchunk), dim3(64), 0, s, d, P, G, g0*nl, ctx->d_stats, ctx->d_flags); \
      else hipLaunchKernelGGL((k_grav_eval<ND_, false, 4, KT_, MP_>), dim3(ngroups*nl, nchunk), dim3(64), 0, s, d, P, G, g0*nl, ctx->d_stats, ctx->d_flags); \
    } else { \
      if (count) hipLaunchKernelGGL((k_grav_eval<ND_, true, GH_MAXOCC, KT_, MP_>), dim3(ngroups*nl, nchunk), dim3(64), 0, s, d, P, G, g0*nl, ctx->d_stats, ctx->d_flags); \
      else hipLaunchKernelGGL((k_grav_eval<ND_, false, GH_MAXOCC, KT_, MP_>), dim3(ngroups*nl, nchunk), dim3(64), 0, s, d, P, G, g0*nl, ctx->d_stats, ctx->d_flags); \
    }
#define LAUNCH(ND_, KT_) if (mpole == GH_MULTIPOLE_QUADRUPOLE) { LAUNCHM(ND_, KT_, 1) } else if (mpole == GH_MULTIPOLE_FAST_MONOPOLE || mpole == GH_MULTIPOLE_FAST_QUADRUPOLE) { LAUNCHM(ND_, KT_, 2) } else { LAUNCHM(ND_, KT_, 0) }
    GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCHM
#undef LAUNCH
  }
  // if a list overflowed the evaluation kernel did nothing; the fused kernel (forces.hip) then does the
  // whole call.  It checks the same word and returns at once otherwise - no host round trip.
#ifdef GH_DEBUG_BLOCKTIME
  if (const char *f = getenv("GH_DEBUG_BLOCKTIME_FILE")) {
    std::vector<double> hw((size_t) 8*ctx->ngroups), he((size_t) ctx->gtot);
    (void) hipStreamSynchronize(ctx->stream);
    (void) hipMemcpy(hw.data(), dbgw, sizeof(double)*hw.size(), hipMemcpyDeviceToHost);
    (void) hipMemcpy(he.data(), dbge, sizeof(double)*he.size(), hipMemcpyDeviceToHost);
    std::string fn = std::string(f) + ".walk";
    FILE *fp = fopen(fn.c_str(), "wb"); if (fp) { fwrite(hw.data(), 8, hw.size(), fp); fclose(fp); }
    fn = std::string(f) + ".eval";
    fp = fopen(fn.c_str(), "wb"); if (fp) { fwrite(he.data(), 8, he.size(), fp); fclose(fp); }
  }
#endif
  int rc = lists_only ? GH_OK : gh_grav_fused_launch(ctx, count, G.fallback);
  gh_phase_end(ctx, GH_T_SPH_FORCES);
  if (rc) return rc;
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}
