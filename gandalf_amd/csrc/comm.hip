// comm.hip -- multi-GPU: domain decomposition by the top cells of the global KD-tree, halo (locally essential
// tree) exchange, multipole all-gather.  One process per GPU; the collectives themselves are supplied by the host
// through gh_comm_ops (torch.distributed over RCCL in bench.py / multigpu.py; MPI or RCCL directly in a C++ host).
//
// Replaces the reference's MPI layer for the hot path: MpiKDTreeDecomposition (src/Mpi/MpiKDTreeDecomposition.cpp:56-135),
// MpiControl::UpdateAllBoundingBoxes / SendReceiveGhosts / ExportParticlesBeforeForceLoop (src/Mpi/MpiControl.cpp:329-337,
// 745-1150) and HydroTree's pruned-tree exchange (src/Tree/HydroTree.cpp:1044-1230).
//
// Design.  nranks = 2^L.  Rank r OWNS level-L cell r of the global KD-tree: its particle range
// [cfirst, cfirst + cN) of the global tree-order index space is static (it depends only on N), so every rank keeps the
// single-GPU index space and cell numbering and simply leaves the slots it neither owns nor imports untouched - with
// 288 GB of HBM per GPU the address space costs nothing and no index translation is ever needed.
//
//   gh_dd_decompose   the L shared top levels: every level's median splits are found EXACTLY over the distributed
//                     particles (two histogram refinements + a gather of the few candidates in the final bin, all
//                     cells of a level in the same collectives), so that the union of the ranks' subtrees IS the tree a
//                     single GPU builds; particles that changed cells migrate (all-to-all-v).
//   gh_dd_publish     all-gather of the top P levels of every rank's subtree (boxes, h-boxes, multipoles: the
//                     "pruned trees" of HydroTree.cpp:1044-1230), then the shared levels are stocked from them.
//   gh_exchange_halo  per phase (density / forces): every rank marks, for every other rank, the cells of its subtree
//                     that rank's walks can reach (conservative restatement of the walks' opening tests against that
//                     rank's published cells) and the leaves whose particles they can touch, packs cell records and
//                     particle records, one all-to-all-v, scatter into the global slots.  A walk that reaches a cell
//                     that was not imported raises FLAG_LET_MISS instead of computing with stale data.
//
// Because the walks then run unchanged on the same tree with the same cell records, every rank computes for its own
// particles exactly what a single GPU computes for them.
#include "gh_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>

#define DD_G 2048            /* histogram bins per refinement of a median search */
#define DD_CAPL 2048         /* candidates per rank and cell gathered in the final bin */
#define DD_PMAX 5            /* published levels per rank subtree (2^(P+1) - 1 cells) */
#define DD_VIS 16384         /* leaves below one published bottom cell that k_let_mark can flag (1M particles on 2 ranks: 4 096) */
#define DD_FMAX 6            /* halo selection: 2^F fine geometry entries per published bottom cell */
#define DD_REC 43            /* doubles per migrating particle: D_COUNT_BASE fields + iorig */

double *gh_time_dev(gh_ctx *ctx);
void gh_rootbox_local(gh_ctx *ctx, int node0);
void gh_stock_top_levels(gh_ctx *ctx, int ltop, int hmax_only);

struct DDCell {                      // one cell of the level being split (device, replicated on all ranks)
  double loA, scA, loB, scB;         // bin(x) = clamp((int) floor((x - lo)*sc), 0, DD_G - 1), coarse (A) and fine (B)
  int binA, binB;                    // the bins that hold the median (-1: not known yet)
  int kd, node;
  long long base, target;            // particles before the current bracket; rank of the median (= cN/2)
  double rdiv; int rdiv_id, pad;     // the split: first particle of the right half in (coordinate, id) order
};
struct DDCand { double key; int id, pad; };
struct LetGeomF;

struct gh_dd {
  gh_comm_ops ops;
  int P = 0;                         // published levels below a rank's cell
  int *topcell = nullptr;            // [own_count] cell index (within its level) of every own particle during gh_dd_decompose
  DDCell *cells = nullptr;           // [nranks]
  int *hist = nullptr, *hist_all = nullptr;          // [nranks/2][DD_G], [nranks][nranks/2][DD_G]
  DDCand *cand = nullptr, *cand_all = nullptr;       // [nranks/2][1 + DD_CAPL], [nranks][...]  (slot 0: count in .id)
  double *box6 = nullptr, *box6_all = nullptr;       // local extent, all ranks' extents
  // migration
  int *mig_cnt = nullptr;            // [2*nranks + 4]: leavers per destination, arrivals per source, cursor words
  int *mig_slot = nullptr;           // [own_count] position of every leaver in its destination's block
  int *mig_hole = nullptr;           // [own_count] positions freed by leavers
  double *mig_send = nullptr, *mig_recv = nullptr;   // [own_count][DD_REC]
  // published subtree tops
  char *pub_send = nullptr, *pub_recv = nullptr; size_t pub_bytes = 0;
  int F = 0;                         // fine entries: 2^F per published bottom cell
  const char *fine_base = nullptr; size_t fine_stride = 0;   // where the gathered fine tables currently are
  char *comb_send = nullptr, *comb_recv = nullptr;           // subtree tops + fine tables in one all-gather
  double fine_widen = 1.0;           // the density widening the gathered fine table was built with
  LetGeomF *fine = nullptr, *fine_all = nullptr;      // [2^(P+F)] own, [nranks][2^(P+F)] everybody's
  // locally essential tree
  int *let_cnt = nullptr;            // [2*MAX] cells / leaves marked per destination, then [2*MAX] received per source
  long long *let_off = nullptr;      // [2*MAX] send / receive block offsets in doubles
  int *let_cells = nullptr, *let_leaves = nullptr;   // [nranks][cap] marked cell ids, leaf ids
  size_t let_cellcap = 0, let_leafcap = 0;
  char *let_send = nullptr, *let_recv = nullptr; size_t let_send_bytes = 0, let_recv_bytes = 0;
  long long migrated = 0;            // particles this rank sent away in the last decomposition
  long long held_particles = 0;      // own + imported (last force-phase exchange): what this rank actually holds
  double dt_local[2];
  double *dt_all = nullptr;
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
#define DD_OP(ctx, call)                                                                         \
  do { const int rc__ = (call); if (rc__) return gh_fail(ctx, GH_ERR_HIP, "collective failed: " #call); } while (0)

static int dd_allgather(gh_ctx *ctx, const void *send, void *recv, size_t bytes)
{
  gh_dd *D = ctx->dd;
  return D->ops.allgather(D->ops.user, send, recv, (int64_t) bytes, (void*) ctx->stream);
}

__device__ __forceinline__ int dd_bin(double x, double lo, double sc)
{
  const double t = floor((x - lo)*sc);
  return t < 0.0 ? 0 : (t > (double) (DD_G - 1) ? DD_G - 1 : (int) t);
}

// ------------------------------------------------------------------------------------------------
// root box: all-gather of the ranks' extents (KDTree.cpp:269-280 over the distributed particles)
// ------------------------------------------------------------------------------------------------
__global__ void k_dd_box_pack(const double *dbbmin, const double *dbbmax, double *box6)
{
  if (threadIdx.x < 3) { box6[threadIdx.x] = dbbmin[threadIdx.x]; box6[3 + threadIdx.x] = dbbmax[threadIdx.x]; }
}
__global__ void k_dd_box_merge(const double *all, int nranks, double *dbbmin, double *dbbmax)
{
  if (threadIdx.x < 3) {
    double mn = 9.9e20, mx = -9.9e20;
    for (int r = 0; r < nranks; r++) { mn = fmin(mn, all[r*6 + threadIdx.x]); mx = fmax(mx, all[r*6 + 3 + threadIdx.x]); }
    dbbmin[threadIdx.x] = mn; dbbmax[threadIdx.x] = mx;
  }
}

// ------------------------------------------------------------------------------------------------
// exact distributed median splits of the shared top levels (KDTree::DivideTreeCell / QuickSelect,
// KDTree.cpp:442-595, 682-750: left child = the cN/2 particles with the smallest coordinate along the longest axis of
// the inherited box; ties between equal coordinates are ordered by particle id here)
// ------------------------------------------------------------------------------------------------
__global__ void k_dd_level_init(DDCell *cells, int level, const double *dbbmin, const double *dbbmax, const int *cN, int ndim)
{
  const int c = threadIdx.x;
  if (c >= (1 << level)) return;
  const int n = (1 << level) - 1 + c;
  double rkmax = 0.0; int kd = 0;
  for (int k = 0; k < ndim; k++) { const double ext = dbbmax[n*3 + k] - dbbmin[n*3 + k]; if (ext > rkmax) { rkmax = ext; kd = k; } }
  DDCell q;
  q.kd = kd; q.node = n;
  q.loA = dbbmin[n*3 + kd];
  q.scA = rkmax > 0.0 ? (double) DD_G/rkmax : 0.0;
  q.loB = 0.0; q.scB = 0.0; q.binA = -1; q.binB = -1;
  q.base = 0; q.target = cN[n]/2;
  q.rdiv = dbbmin[n*3 + kd]; q.rdiv_id = -1; q.pad = 0;
  cells[c] = q;
}

__global__ void k_dd_hist(DevicePtrs d, const int *topcell, const DDCell *cells, int round, int *hist)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  const int c = topcell[i];
  const DDCell q = cells[c];
  const double x = d.f[D_RX + q.kd][i];
  const int a = dd_bin(x, q.loA, q.scA);
  if (round == 0) atomicAdd(&hist[c*DD_G + a], 1);
  else if (a == q.binA) atomicAdd(&hist[c*DD_G + dd_bin(x, q.loB, q.scB)], 1);
}

// one workgroup per cell: sum the ranks' histograms, find the bin that holds the median
__global__ __launch_bounds__(1024) void k_dd_scan(DDCell *cells, const int *hist_all, int ncells, int nranks, int round)
{
  __shared__ long long s_cum[DD_G];
  const int c = blockIdx.x;
  for (int b = threadIdx.x; b < DD_G; b += blockDim.x) {
    long long v = 0;
    for (int r = 0; r < nranks; r++) v += hist_all[((size_t) r*ncells + c)*DD_G + b];
    s_cum[b] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    DDCell q = cells[c];
    long long run = q.base;
    int bsel = DD_G - 1;
    for (int b = 0; b < DD_G; b++) {
      if (q.target < run + s_cum[b]) { bsel = b; break; }
      run += s_cum[b];
    }
    if (bsel == DD_G - 1 && !(q.target < run + s_cum[bsel])) { /* empty cell: keep the last bin */ }
    q.base = run;
    if (round == 0) {
      q.binA = bsel;
      q.loB = q.scA > 0.0 ? q.loA + (double) bsel/q.scA : q.loA;
      q.scB = q.scA*(double) DD_G;
    }
    else q.binB = bsel;
    cells[c] = q;
  }
}

__global__ void k_dd_collect(DevicePtrs d, const int *topcell, const DDCell *cells, DDCand *cand, int *flags)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  const int c = topcell[i];
  const DDCell q = cells[c];
  const double x = d.f[D_RX + q.kd][i];
  if (dd_bin(x, q.loA, q.scA) != q.binA || dd_bin(x, q.loB, q.scB) != q.binB) return;
  DDCand *cc = cand + (size_t) c*(1 + DD_CAPL);
  const int slot = atomicAdd(&cc[0].id, 1);
  if (slot < DD_CAPL) { cc[1 + slot].key = x; cc[1 + slot].id = d.iorig[i]; cc[1 + slot].pad = 0; }
  else atomicOr(flags, FLAG_DD_SPLIT);
}

// one workgroup per cell: the candidate of rank (target - base) in (coordinate, id) order is the split; the
// children's inherited boxes follow (KDTree.cpp:508-527)
__global__ __launch_bounds__(1024) void k_dd_select(DDCell *cells, const DDCand *cand_all, int ncells, int nranks,
                                                    double *dbbmin, double *dbbmax, int *kdiv, int *flags)
{
  const int c = blockIdx.x;
  __shared__ int s_off[GH_MAX_RANKS + 1];
  __shared__ int s_found;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int r = 0; r < nranks; r++) { s_off[r] = run; run += min(cand_all[((size_t) r*ncells + c)*(1 + DD_CAPL)].id, DD_CAPL); }
    s_off[nranks] = run;
    s_found = -1;
  }
  __syncthreads();
  const DDCell q = cells[c];
  const int ntot = s_off[nranks];
  const long long want = q.target - q.base;
  auto entry = [&](int e) -> DDCand {
    int r = 0;
    while (r + 1 < nranks && e >= s_off[r + 1]) r++;
    return cand_all[((size_t) r*ncells + c)*(1 + DD_CAPL) + 1 + (e - s_off[r])];
  };
  for (int e = threadIdx.x; e < ntot; e += blockDim.x) {
    const DDCand me = entry(e);
    long long rank = 0;
    for (int r = 0; r < nranks; r++) {
      const DDCand *p = cand_all + ((size_t) r*ncells + c)*(1 + DD_CAPL) + 1;
      const int nr = s_off[r + 1] - s_off[r];
      for (int t = 0; t < nr; t++) { const DDCand o = p[t]; if (o.key < me.key || (o.key == me.key && o.id < me.id)) rank++; }
    }
    if (rank == want) s_found = e;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    DDCell w = q;
    const int n = q.node, c1 = 2*n + 1, c2 = 2*n + 2, kd = q.kd;
    if (s_found >= 0) { const DDCand m = entry(s_found); w.rdiv = m.key; w.rdiv_id = m.id; }
    else if (ntot > 0 || q.target > 0) atomicOr(flags, FLAG_DD_SPLIT);       // the bracket lost the median: never expected
    for (int k = 0; k < 3; k++) {
      dbbmin[c1*3 + k] = dbbmin[n*3 + k]; dbbmax[c1*3 + k] = (k == kd) ? w.rdiv : dbbmax[n*3 + k];
      dbbmin[c2*3 + k] = (k == kd) ? w.rdiv : dbbmin[n*3 + k]; dbbmax[c2*3 + k] = dbbmax[n*3 + k];
    }
    kdiv[n] = kd;
    cells[c] = w;
  }
}

__global__ void k_dd_assign(DevicePtrs d, int *topcell, const DDCell *cells)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  const int c = topcell[i];
  const DDCell q = cells[c];
  const double x = d.f[D_RX + q.kd][i];
  const int right = (q.rdiv_id >= 0 && (x > q.rdiv || (x == q.rdiv && d.iorig[i] >= q.rdiv_id))) ? 1 : 0;
  topcell[i] = 2*c + right;
}

// ------------------------------------------------------------------------------------------------
// migration
// ------------------------------------------------------------------------------------------------
__global__ void k_mig_count(int n, const int *dest, int self, int *cnt, int *slot, int *hole)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = dest[i];
  if (t == self) return;
  slot[i] = atomicAdd(&cnt[t], 1);
  hole[atomicAdd(&cnt[2*GH_MAX_RANKS], 1)] = i;
}

struct MigTab { double *fld[DD_REC]; int off[GH_MAX_RANKS]; };

__global__ void k_mig_pack(MigTab t, const int *iorig, int n, const int *dest, int self, const int *slot, double *send)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int to = dest[i];
  if (to == self) return;
  double *o = send + (size_t) (t.off[to] + slot[i])*DD_REC;
  for (int f = 0; f < DD_REC - 1; f++) o[f] = t.fld[f][i];
  o[DD_REC - 1] = (double) iorig[i];
}

__global__ void k_mig_unpack(MigTab t, int *iorig, int narr, const int *hole, const double *recv)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= narr) return;
  const int i = hole[e];
  const double *o = recv + (size_t) e*DD_REC;
  for (int f = 0; f < DD_REC - 1; f++) t.fld[f][i] = o[f];
  iorig[i] = (int) o[DD_REC - 1];
}

// ------------------------------------------------------------------------------------------------
// published subtree tops: levels L .. L+P of every rank's subtree, all four cell records (+ quadrupoles)
// ------------------------------------------------------------------------------------------------
struct PubRec { CellBox b; CellH h; CellGeo g; CellCom c; CellQuad q; };      // 288 bytes

// Geometry of one cell of the destination rank as the halo selection sees it: the cell's boxes plus the largest leaf
// reach below it.  Every rank publishes this for the 2^(P+F) cells F levels below its published bottom cells ("fine"
// entries, all-gathered with the subtree tops); the opening tests run against a published bottom cell first and, where
// that says "open", against its 2^F fine entries.  The gravity walk tests every LEAF with the leaf's own rmax and hmax
// (Tree.cpp:659-672), so the bounds that matter are maxima over leaves - and they must be local: in a Plummer halo one
// 1000-particle cell spans smoothing lengths from 0.3 to 2, and its innermost corner combined with its outermost
// leaf's reach would "need" the whole core.  Hence the two boxes around the leaves' own balls (rb*, cb*).
struct LetGeom {
  double bbmin[3], bbmax[3], hbmin[3], hbmax[3];
  double rbmin[3], rbmax[3];          // box around the leaves' balls of radius rmax + kernrange*hmax about their centres
  double cbmin[3], cbmax[3];          // ... of radius rmax
  double dbmin[3], dbmax[3];          // box around the density search boxes of its particle groups (group box -/+ kernrange*1.05^2*hmax*widen)
  int N, pad;
};
// as it travels: single precision, every box face rounded OUTWARD (the tests only have to be conservative), 128 bytes
struct LetGeomF {
  float b[5][6];                      // bb, hb, rb, cb, db: min[3] then max[3]
  int N, pad;
};
__device__ __forceinline__ float f_down(double x) { float f = (float) x; return (double) f > x ? nextafterf(f, -INFINITY) : f; }
__device__ __forceinline__ float f_up(double x) { float f = (float) x; return (double) f < x ? nextafterf(f, INFINITY) : f; }
__device__ __forceinline__ void let_expand(const LetGeomF &e, LetGeom &q)
{
  for (int k = 0; k < 3; k++) {
    q.bbmin[k] = e.b[0][k]; q.bbmax[k] = e.b[0][3 + k]; q.hbmin[k] = e.b[1][k]; q.hbmax[k] = e.b[1][3 + k];
    q.rbmin[k] = e.b[2][k]; q.rbmax[k] = e.b[2][3 + k]; q.cbmin[k] = e.b[3][k]; q.cbmax[k] = e.b[3][3 + k];
    q.dbmin[k] = e.b[4][k]; q.dbmax[k] = e.b[4][3 + k];
  }
  q.N = e.N; q.pad = 0;
}

__global__ __launch_bounds__(256) void k_pub_fine(DevicePtrs d, int L, int PF, int rank, double kernrange, double widen, LetGeomF *out)
{
  __shared__ double s_red[4][18];
  const int j = blockIdx.x;
  const int n = (1 << (L + PF)) - 1 + (rank << PF) + j;
  const int nl = 1 << (d.ltot - L - PF);                // leaves below fine cell j of this rank
  const int leaf0 = (d.gtot - 1) + ((rank << PF) + j)*nl;
  double v[18];                                         // rbmin, cbmin (as minima), -rbmax, -cbmax (maxima, negated), dbmin, -dbmax
  for (int k = 0; k < 18; k++) v[k] = 9.9e20;
  {
    // density search boxes of the particle groups below this cell (k_dens_walk: group box -/+ kernrange * 1.05 * hmax,
    // one retry of the reference's 1.05 growth included; `widen` grows with the retries of gh_density_impl)
    const int ng = 1 << (d.lgroup - L - PF);
    const int g0 = (1 << d.lgroup) - 1 + ((rank << PF) + j)*ng;
    for (int t = threadIdx.x; t < ng; t += blockDim.x) {
      const CellBox gb = d.cbox[g0 + t];
      if (gb.N > 0) {
        const double rs = kernrange*d.ch[g0 + t].hmax*(1.05*1.05)*widen*(1.0 + 1e-12);
        for (int k = 0; k < 3; k++) { v[12 + k] = fmin(v[12 + k], gb.bbmin[k] - rs); v[15 + k] = fmin(v[15 + k], -(gb.bbmax[k] + rs)); }
      }
    }
  }
  for (int t = threadIdx.x; t < nl; t += blockDim.x) {
    const CellGeo g = d.cgeo[leaf0 + t];
    if (g.N > 0) {
      const double r1 = g.rmax + kernrange*g.hmax, r2 = g.rmax;
      for (int k = 0; k < 3; k++) {
        v[k] = fmin(v[k], g.rcell[k] - r1); v[3 + k] = fmin(v[3 + k], g.rcell[k] - r2);
        v[6 + k] = fmin(v[6 + k], -(g.rcell[k] + r1)); v[9 + k] = fmin(v[9 + k], -(g.rcell[k] + r2));
      }
    }
  }
  for (int k = 0; k < 18; k++) {
    double x = v[k];
    for (int off = 32; off > 0; off >>= 1) x = fmin(x, __shfl_xor(x, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const CellBox cb = d.cbox[n]; const CellH ch = d.ch[n];
    LetGeomF q;
    auto red = [&](int c) { return fmin(fmin(s_red[0][c], s_red[1][c]), fmin(s_red[2][c], s_red[3][c])); };
    for (int k = 0; k < 3; k++) {
      q.b[0][k] = f_down(cb.bbmin[k]); q.b[0][3 + k] = f_up(cb.bbmax[k]);
      q.b[1][k] = f_down(ch.hbmin[k]); q.b[1][3 + k] = f_up(ch.hbmax[k]);
      q.b[2][k] = f_down(red(k)); q.b[2][3 + k] = f_up(-red(6 + k));
      q.b[3][k] = f_down(red(3 + k)); q.b[3][3 + k] = f_up(-red(9 + k));
      q.b[4][k] = f_down(red(12 + k)); q.b[4][3 + k] = f_up(-red(15 + k));
    }
    q.N = cb.N; q.pad = 0;
    out[j] = q;
  }
}

__global__ void k_pub_pack(DevicePtrs d, int L, int P, int rank, PubRec *out)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  const int ncell = (2 << P) - 1;
  if (e >= ncell) return;
  int p = 0;
  while (e >= (2 << p) - 1) p++;                       // level L+p holds entries [2^p - 1, 2^(p+1) - 1)
  const int j = e - ((1 << p) - 1);
  const int n = (1 << (L + p)) - 1 + (rank << p) + j;
  PubRec r;
  r.b = d.cbox[n]; r.h = d.ch[n]; r.g = d.cgeo[n]; r.c = d.ccom[n];
  if (d.cquad) r.q = d.cquad[n]; else { for (int k = 0; k < 5; k++) r.q.q[k] = 0.0; for (int k = 0; k < 3; k++) r.q.pad[k] = 0.0; }
  out[e] = r;
}

__global__ void k_pub_unpack(DevicePtrs d, int L, int P, int self, int nranks, const char *all, size_t stride)
{
  const int ncell = (2 << P) - 1;
  const int t = blockIdx.x*blockDim.x + threadIdx.x;
  if (t >= ncell*nranks) return;
  const int r = t/ncell, e = t - r*ncell;
  int p = 0;
  while (e >= (2 << p) - 1) p++;
  const int j = e - ((1 << p) - 1);
  if (r == self) return;
  const int n = (1 << (L + p)) - 1 + (r << p) + j;
  const PubRec &q = ((const PubRec*) (all + (size_t) r*stride))[e];
  d.cbox[n] = q.b; d.ch[n] = q.h; d.cgeo[n] = q.g; d.ccom[n] = q.c;
  if (d.cquad) d.cquad[n] = q.q;
}

// ------------------------------------------------------------------------------------------------
// locally essential tree
// ------------------------------------------------------------------------------------------------
// Could a walk of any leaf cell / particle group below the destination's published cell Q open cell Y (visit its
// children; for a leaf: touch its particles)?  Conservative restatements of
//   density : Tree::ComputeGatherNeighbourList (Tree.cpp:319-328) with the search radius kernrange*hmax*1.05^2
//             (first try and one retry of GradhSphTree.cpp:141-226; a walk that needs more raises FLAG_LET_MISS);
//   hydro   : Tree::ComputeNeighbourAndGhostList (Tree.cpp:579-580): overlap(bb, other.hbox) || overlap(hbox, other.bb);
//   gravity : Tree::ComputeGravityInteractionAndGhostList (Tree.cpp:659-700) + open_cell_for_gravity (Tree.h:413-432,
//             geometric MAC): a leaf opens Y if their centres are within Y.rmax + leaf.rmax + kernrange*max(leaf.hmax, Y.hmax)
//             - tested as "Y's ball reaches the box around the leaves' balls" - or closer than Y's opening distance
//             (leaf centres lie inside Q's box).
template <int PHASE>
__device__ __forceinline__ bool let_may_open(const LetGeom &Q, const CellBox &yb, const CellH &yh, const CellGeo &yg,
                                             int ndim, double kernrange, double widen)
{
  if (Q.N <= 0) return false;
  if (PHASE == GH_HALO_DENSITY) {
    for (int k = 0; k < ndim; k++) if (Q.dbmin[k] > yb.bbmax[k] || yb.bbmin[k] > Q.dbmax[k]) return false;
    return true;
  }
  if (PHASE == GH_HALO_HYDRO) {
    bool o1 = true, o2 = true;
    for (int k = 0; k < ndim; k++) {
      if (Q.bbmin[k] > yh.hbmax[k] || yh.hbmin[k] > Q.bbmax[k]) o1 = false;
      if (Q.hbmin[k] > yb.bbmax[k] || yb.bbmin[k] > Q.hbmax[k]) o2 = false;
    }
    return o1 || o2;
  }
  double dr2 = 0.0, dc2 = 0.0, db2 = 0.0;            // squared distance of Y's centre from the three boxes
  for (int k = 0; k < ndim; k++) {
    const double gr = fmax(fmax(Q.rbmin[k] - yg.rcell[k], yg.rcell[k] - Q.rbmax[k]), 0.0);
    const double gc = fmax(fmax(Q.cbmin[k] - yg.rcell[k], yg.rcell[k] - Q.cbmax[k]), 0.0);
    const double gb = fmax(fmax(Q.bbmin[k] - yg.rcell[k], yg.rcell[k] - Q.bbmax[k]), 0.0);
    dr2 += gr*gr; dc2 += gc*gc; db2 += gb*gb;
  }
  const double e = 1.0 + 1e-12;
  const double r1 = yg.rmax*e, r2 = (yg.rmax + kernrange*yg.hmax)*e;
  return dr2 <= r1*r1*e || dc2 <= r2*r2*e || db2*(1.0 - 1e-12) < yg.cdistsqd;
}

// periodic domains: the destination's walks also meet Y as its periodic images (density: image codes of the walk,
// k_dens_walk; hydro: GhostNeighbourFinder::PeriodicBoxOverlap, GhostNeighbours.hpp:202-226) - Y is needed if any image
// within one box length can be opened.  len[k] = box size of a periodic dimension, 0 otherwise.
struct LetPeriod { double len[3]; };
template <int PHASE>
__device__ __forceinline__ bool let_may_open_img(const LetGeom &Q, const CellBox &yb, const CellH &yh, const CellGeo &yg,
                                                 int ndim, double kernrange, double widen, const LetPeriod &per)
{
  if (let_may_open<PHASE>(Q, yb, yh, yg, ndim, kernrange, widen)) return true;
  if (per.len[0] == 0.0 && per.len[1] == 0.0 && per.len[2] == 0.0) return false;
  if (Q.N <= 0) return false;
  const int n0 = per.len[0] > 0.0 ? 1 : 0, n1 = (ndim > 1 && per.len[1] > 0.0) ? 1 : 0, n2 = (ndim > 2 && per.len[2] > 0.0) ? 1 : 0;
  for (int i0 = -n0; i0 <= n0; i0++)
    for (int i1 = -n1; i1 <= n1; i1++)
      for (int i2 = -n2; i2 <= n2; i2++) {
        if (i0 == 0 && i1 == 0 && i2 == 0) continue;
        const double sh[3] = {i0*per.len[0], i1*per.len[1], i2*per.len[2]};
        CellBox b = yb; CellH h = yh; CellGeo g = yg;
        for (int k = 0; k < 3; k++) {
          b.bbmin[k] += sh[k]; b.bbmax[k] += sh[k]; h.hbmin[k] += sh[k]; h.hbmax[k] += sh[k]; g.rcell[k] += sh[k];
        }
        if (let_may_open<PHASE>(Q, b, h, g, ndim, kernrange, widen)) return true;
      }
  return false;
}

// grid (2^P, nranks): workgroup (j, r) marks what rank r needs of the subtree below this rank's published bottom
// cell j, level by level with the visit flags in LDS
template <int PHASE>
__global__ __launch_bounds__(1024) void k_let_mark(DevicePtrs d, int L, int P, int F, int self, double kernrange, double widen, const char *fine_base, size_t fine_stride,
                                                  int *cnt, int *cells, int *leaves, size_t cellcap, size_t leafcap, int *flags, LetPeriod per)
{
  const int r = blockIdx.y;
  if (r == self) return;
  __shared__ LetGeom s_q[1 << DD_PMAX];
  __shared__ unsigned char s_vis[2][DD_VIS];
  const int nq = 1 << P, nf = 1 << F;
  const LetGeomF *fine = (const LetGeomF*) (fine_base + (size_t) r*fine_stride);
  if ((int) threadIdx.x < nq) {
    // published bottom cell of the destination = union of its fine entries
    LetGeom q;
    let_expand(fine[(size_t) threadIdx.x*nf], q);
    for (int f = 1; f < nf; f++) {
      LetGeom e;
      let_expand(fine[(size_t) threadIdx.x*nf + f], e);
      if (e.N <= 0) continue;
      if (q.N <= 0) { q = e; continue; }
      for (int k = 0; k < 3; k++) {
        q.bbmin[k] = fmin(q.bbmin[k], e.bbmin[k]); q.bbmax[k] = fmax(q.bbmax[k], e.bbmax[k]);
        q.hbmin[k] = fmin(q.hbmin[k], e.hbmin[k]); q.hbmax[k] = fmax(q.hbmax[k], e.hbmax[k]);
        q.rbmin[k] = fmin(q.rbmin[k], e.rbmin[k]); q.rbmax[k] = fmax(q.rbmax[k], e.rbmax[k]);
        q.cbmin[k] = fmin(q.cbmin[k], e.cbmin[k]); q.cbmax[k] = fmax(q.cbmax[k], e.cbmax[k]);
        q.dbmin[k] = fmin(q.dbmin[k], e.dbmin[k]); q.dbmax[k] = fmax(q.dbmax[k], e.dbmax[k]);
      }
      q.N += e.N;
    }
    s_q[threadIdx.x] = q;
  }
  const int depth = d.ltot - (L + P);                    // levels below the published bottom cell
  const int y0 = (1 << (L + P)) - 1 + (self << P) + blockIdx.x;
  if (threadIdx.x == 0) s_vis[0][0] = 1;
  __syncthreads();
  int cur = 0;
  for (int t = 0; t <= depth; t++) {
    const int nn = 1 << t;
    const int nbase = ((y0 + 1) << t) - 1;
    const bool leaflevel = t == depth;
    for (int k0 = 0; k0 < nn; k0 += blockDim.x) {
      const int k = k0 + threadIdx.x;
      bool vis = false, open = false;
      int n = 0;
      bool test = false;
      if (k < nn && s_vis[cur][k]) {
        vis = true;
        n = nbase + k;
        // a visited leaf travels whether or not its particles would be touched (below): no opening test on the last level,
        // where half of all cells sit and where an exhaustive "no" is the usual answer
        test = d.cbox[n].N > 0 && !(leaflevel && t > 0);
      }
      // the opening test of one node is shared by the wave: lane = published cell of the destination for the coarse test,
      // lane = fine entry of one published cell for the fine test (one coalesced read per round).  One thread per node with
      // the fine entries in a serial early-exit loop was latency-bound: 75 ms per launch for 1M particles on 2 ranks.
      {
        const int lanei = threadIdx.x & 63;
        unsigned long long tm = __ballot(test);
        while (tm) {
          const int j = __ffsll((long long) tm) - 1;
          tm &= tm - 1ull;
          const int nj = __shfl(n, j, 64);
          const CellBox yb = d.cbox[nj]; const CellH yh = d.ch[nj]; const CellGeo yg = d.cgeo[nj];
          bool cpass = false;
          for (int q0 = 0; q0 < nq; q0 += 64) {
            const int qi = q0 + lanei;
            if (qi < nq) cpass = let_may_open_img<PHASE>(s_q[qi], yb, yh, yg, d.ndim, kernrange, widen, per);
            unsigned long long cm = __ballot(cpass);
            bool any = false;
            while (cm && !any) {
              const int qq = q0 + __ffsll((long long) cm) - 1;
              cm &= cm - 1ull;
              bool o = false;
              for (int f0 = 0; f0 < nf; f0 += 64) {
                const int f = f0 + lanei;
                if (f < nf) { LetGeom e; let_expand(fine[(size_t) qq*nf + f], e); o = o || let_may_open_img<PHASE>(e, yb, yh, yg, d.ndim, kernrange, widen, per); }
              }
              any = __any(o);
            }
            if (any) { if (lanei == j) open = true; break; }
          }
        }
      }
      if (!leaflevel && k < nn) { s_vis[cur ^ 1][2*k] = open; s_vis[cur ^ 1][2*k + 1] = open; }
      // visited cells below the published levels travel as records; opened leaves (and every visited leaf: a leaf
      // with one particle enters the gravity lists as a particle, Tree.cpp:713) travel with their particles
      const bool sendcell = vis && t > 0;
      const unsigned long long cm = __ballot(sendcell);
      if (cm) {
        int base = 0;
        const int lanei = threadIdx.x & 63;
        if (lanei == 0) base = atomicAdd(&cnt[2*r], __popcll(cm));
        base = __shfl(base, 0, 64);
        if (sendcell) {
          const size_t pos = (size_t) base + __popcll(cm & (lanei ? ((~0ull) >> (64 - lanei)) : 0ull));
          if (pos < cellcap) cells[(size_t) r*cellcap + pos] = n; else atomicOr(flags, FLAG_LEAFLIST_OVERFLOW);
        }
      }
      const bool sendleaf = vis && leaflevel && (t == 0 ? open : true);
      const unsigned long long lm = __ballot(sendleaf);
      if (lm) {
        int base = 0;
        const int lanei = threadIdx.x & 63;
        if (lanei == 0) base = atomicAdd(&cnt[2*r + 1], __popcll(lm));
        base = __shfl(base, 0, 64);
        if (sendleaf) {
          const size_t pos = (size_t) base + __popcll(lm & (lanei ? ((~0ull) >> (64 - lanei)) : 0ull));
          if (pos < leafcap) leaves[(size_t) r*leafcap + pos] = n; else atomicOr(flags, FLAG_LEAFLIST_OVERFLOW);
        }
      }
    }
    __syncthreads();
    cur ^= 1;
  }
}

// record sizes in doubles: cell = id + the records of the phase; leaf = id + occ x particle record
struct LetLayout { int cell_dbl, part_dbl, leaf_dbl, occ, phase, quad; };

__global__ void k_let_pack(DevicePtrs d, LetLayout lay, int nranks, int self, const int *cnt, const int *cells, const int *leaves,
                           size_t cellcap, size_t leafcap, const long long *off /* [nranks] doubles */, double *send)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int nc = cnt[2*r], nl = cnt[2*r + 1];
  double *base = send + off[r];
  for (int e = blockIdx.x*blockDim.x + threadIdx.x; e < nc + nl; e += gridDim.x*blockDim.x) {
    if (e < nc) {
      const int n = cells[(size_t) r*cellcap + e];
      double *o = base + (size_t) e*lay.cell_dbl;
      o[0] = (double) n;
      const double *b = (const double*) &d.cbox[n];
      for (int k = 0; k < 8; k++) o[1 + k] = b[k];
      if (lay.phase != GH_HALO_DENSITY) {
        const double *h = (const double*) &d.ch[n], *g = (const double*) &d.cgeo[n], *c = (const double*) &d.ccom[n];
        for (int k = 0; k < 8; k++) { o[9 + k] = h[k]; o[17 + k] = g[k]; }
        for (int k = 0; k < 4; k++) o[25 + k] = c[k];
        if (lay.quad) { const double *qq = (const double*) &d.cquad[n]; for (int k = 0; k < 5; k++) o[29 + k] = qq[k]; }
      }
    }
    else {
      const int n = leaves[(size_t) r*leafcap + (e - nc)];
      double *o = base + (size_t) nc*lay.cell_dbl + (size_t) (e - nc)*lay.leaf_dbl;
      o[0] = (double) n;
      const int first = d.cfirst[n], cn = d.cN[n];
      for (int t = 0; t < lay.occ; t++) {
        double *po = o + 1 + (size_t) t*lay.part_dbl;
        if (t < cn) {
          const double *pm = (const double*) &d.posm[first + t];
          for (int k = 0; k < 4; k++) po[k] = pm[k];
          if (lay.phase != GH_HALO_DENSITY) {
            const double *hr = (const double*) &d.hrec[4*(size_t) (first + t)];
            for (int k = 0; k < 16; k++) po[4 + k] = hr[k];
          }
        }
      }
    }
  }
}

__global__ void k_let_unpack(DevicePtrs d, LetLayout lay, int nranks, int self, const int *rcnt, const long long *roff, const double *recv)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int nc = rcnt[2*r], nl = rcnt[2*r + 1];
  const double *base = recv + roff[r];
  for (int e = blockIdx.x*blockDim.x + threadIdx.x; e < nc + nl; e += gridDim.x*blockDim.x) {
    if (e < nc) {
      const double *o = base + (size_t) e*lay.cell_dbl;
      const int n = (int) o[0];
      double *b = (double*) &d.cbox[n];
      for (int k = 0; k < 8; k++) b[k] = o[1 + k];
      if (lay.phase != GH_HALO_DENSITY) {
        double *h = (double*) &d.ch[n], *g = (double*) &d.cgeo[n], *c = (double*) &d.ccom[n];
        for (int k = 0; k < 8; k++) { h[k] = o[9 + k]; g[k] = o[17 + k]; }
        for (int k = 0; k < 4; k++) c[k] = o[25 + k];
        if (lay.quad) { double *qq = (double*) &d.cquad[n]; for (int k = 0; k < 5; k++) qq[k] = o[29 + k]; }
      }
    }
    else {
      const double *o = base + (size_t) nc*lay.cell_dbl + (size_t) (e - nc)*lay.leaf_dbl;
      const int n = (int) o[0];
      const int first = d.cfirst[n], cn = d.cN[n];
      for (int t = 0; t < cn; t++) {
        const double *po = o + 1 + (size_t) t*lay.part_dbl;
        double *pm = (double*) &d.posm[first + t];
        for (int k = 0; k < 4; k++) pm[k] = po[k];
        if (lay.phase != GH_HALO_DENSITY) {
          double *hr = (double*) &d.hrec[4*(size_t) (first + t)];
          for (int k = 0; k < 16; k++) hr[k] = po[4 + k];
        }
      }
    }
  }
}

// cells of the other ranks' subtrees below the published levels: "not imported" until an exchange brings them
__global__ void k_let_invalidate(DevicePtrs d, int L, int P, int self, int Ncell)
{
  const int n = blockIdx.x*blockDim.x + threadIdx.x;
  if (n >= Ncell) return;
  int lev = 0;
  while (n >= (2 << lev) - 1) lev++;                    // level of heap node n
  if (lev <= L + P) return;
  const int owner = (n - ((1 << lev) - 1)) >> (lev - L);
  if (owner == self) return;
  d.cbox[n].N = -1;
  d.cgeo[n].N = -1;
}

__global__ void k_dd_min_dt(const double *all, int nranks, double *time)
{
  double v = 9.9e50;
  for (int r = 0; r < nranks; r++) v = fmin(v, all[r]);
  time[1] = v;
}

// ================================================================================================
// host
// ================================================================================================
extern "C" int gh_comm_init(gh_ctx *ctx, int rank, int nranks, const gh_comm_ops *ops)
{
  if (!ctx || nranks < 1 || rank < 0 || rank >= nranks) return GH_ERR_INVALID;
  if (nranks > GH_MAX_RANKS) return gh_fail(ctx, GH_ERR_INVALID, "gh_comm_init: more ranks than GH_MAX_RANKS");
  if (nranks & (nranks - 1)) return gh_fail(ctx, GH_ERR_INVALID, "gh_comm_init: the rank count must be a power of two (one top-level KD cell per rank)");
  if (nranks > 1 && (!ops || !ops->allgather || !ops->alltoallv)) return gh_fail(ctx, GH_ERR_INVALID, "gh_comm_init: collectives missing");
  if (ctx->N > 0) return gh_fail(ctx, GH_ERR_INVALID, "gh_comm_init: call before gh_upload_particles");
  if (nranks > 1) {
    const gh_config &c = ctx->cfg;
    bool ok = true;                                        // open or periodic (both faces of a dimension alike); no mirrors
    for (int k = 0; k < ctx->ndim; k++)
      ok = ok && c.boundary_lhs[k] == c.boundary_rhs[k] && (c.boundary_lhs[k] == GH_BOUNDARY_OPEN || c.boundary_lhs[k] == GH_BOUNDARY_PERIODIC);
    if (!ok) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: open or periodic boundaries only");
    if (c.Nlevels > 1) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: global timestep only (Nlevels = 1)");
    if (c.self_gravity && c.gravity_mac != GH_MAC_GEOMETRIC) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: gravity_mac = geometric only");
    if (c.avisc == GH_AVISC_MON97CD2010 || c.avisc == GH_AVISC_MON97MM97) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: no time-dependent viscosity");
    if (c.ntreebuildstep > 1) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: the tree is rebuilt every step (ntreebuildstep = 1)");
  }
  if (!ctx->dd) ctx->dd = new gh_dd();
  if (ops) ctx->dd->ops = *ops;
  ctx->rank = rank; ctx->nranks = nranks;
  return GH_OK;
}

void gh_dd_free(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  if (!D) return;
  void *ptrs[] = {D->topcell, D->cells, D->hist, D->hist_all, D->cand, D->cand_all, D->box6, D->box6_all, D->mig_cnt, D->mig_slot,
                  D->mig_hole, D->mig_send, D->mig_recv, D->pub_send, D->pub_recv, D->comb_send, D->comb_recv, D->fine, D->fine_all, D->let_cnt, D->let_off, D->let_cells, D->let_leaves,
                  D->let_send, D->let_recv, D->dt_all};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  delete D;
  ctx->dd = nullptr;
}

static int dd_alloc(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  if (D->topcell) return GH_OK;
  const int W = ctx->nranks, half = std::max(W/2, 1);
  const size_t n = (size_t) ctx->own_count + 1;
  D->P = std::min(DD_PMAX, ctx->lgroup - ctx->L);
  GH_CHECK(ctx, hipMalloc((void**) &D->topcell, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->cells, sizeof(DDCell)*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->hist, sizeof(int)*(size_t) half*DD_G));
  GH_CHECK(ctx, hipMalloc((void**) &D->hist_all, sizeof(int)*(size_t) W*half*DD_G));
  GH_CHECK(ctx, hipMalloc((void**) &D->cand, sizeof(DDCand)*(size_t) half*(1 + DD_CAPL)));
  GH_CHECK(ctx, hipMalloc((void**) &D->cand_all, sizeof(DDCand)*(size_t) W*half*(1 + DD_CAPL)));
  GH_CHECK(ctx, hipMalloc((void**) &D->box6, sizeof(double)*8));
  GH_CHECK(ctx, hipMalloc((void**) &D->box6_all, sizeof(double)*8*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_cnt, sizeof(int)*(4*GH_MAX_RANKS)));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_slot, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_hole, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_send, sizeof(double)*n*DD_REC));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_recv, sizeof(double)*n*DD_REC));
  D->pub_bytes = sizeof(PubRec)*(size_t) ((2 << D->P) - 1);
  GH_CHECK(ctx, hipMalloc((void**) &D->pub_send, D->pub_bytes));
  GH_CHECK(ctx, hipMalloc((void**) &D->pub_recv, D->pub_bytes*W));
  D->F = std::max(0, std::min(DD_FMAX, ctx->lgroup - ctx->L - D->P));
  GH_CHECK(ctx, hipMalloc((void**) &D->fine, sizeof(LetGeomF)*((size_t) 1 << (D->P + D->F))));
  GH_CHECK(ctx, hipMalloc((void**) &D->fine_all, sizeof(LetGeomF)*((size_t) W << (D->P + D->F))));
  {
    const size_t blk = D->pub_bytes + sizeof(LetGeomF)*((size_t) 1 << (D->P + D->F));
    GH_CHECK(ctx, hipMalloc((void**) &D->comb_send, blk));
    GH_CHECK(ctx, hipMalloc((void**) &D->comb_recv, blk*W));
  }
  GH_CHECK(ctx, hipMalloc((void**) &D->let_cnt, sizeof(int)*8*GH_MAX_RANKS));
  GH_CHECK(ctx, hipMalloc((void**) &D->let_off, sizeof(long long)*2*GH_MAX_RANKS));
  // a rank can need, at most, all of another rank's subtree
  D->let_cellcap = (size_t) 2*(ctx->gtot >> ctx->L); D->let_leafcap = (size_t) (ctx->gtot >> ctx->L);
  GH_CHECK(ctx, hipMalloc((void**) &D->let_cells, sizeof(int)*D->let_cellcap*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->let_leaves, sizeof(int)*D->let_leafcap*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->dt_all, sizeof(double)*(W + 2)));
  return GH_OK;
}

// the L shared top levels and the migration (see the header comment); leaves dbbmin/dbbmax of this rank's cell set
int gh_dd_decompose(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  if (!D) return gh_fail(ctx, GH_ERR_INVALID, "multi-GPU: gh_comm_init was not called");
  int rc = dd_alloc(ctx);
  if (rc) return rc;
  const int W = ctx->nranks, L = ctx->L;
  hipStream_t s = ctx->stream;
  const int pn = (int) ctx->own_count;
  const int nb = cdiv(pn, 256);
  DevicePtrs own = gh_dev_own(ctx);

  // global root box
  gh_rootbox_local(ctx, (1 << L) - 1 + ctx->rank);
  hipLaunchKernelGGL(k_dd_box_pack, dim3(1), dim3(64), 0, s, ctx->dbbmin, ctx->dbbmax, D->box6);
  DD_OP(ctx, dd_allgather(ctx, D->box6, D->box6_all, sizeof(double)*6));
  hipLaunchKernelGGL(k_dd_box_merge, dim3(1), dim3(64), 0, s, D->box6_all, W, ctx->dbbmin, ctx->dbbmax);

  // level by level: exact medians over all ranks' particles
  GH_CHECK(ctx, hipMemsetAsync(D->topcell, 0, sizeof(int)*(size_t) pn, s));
  for (int l = 0; l < L; l++) {
    const int nc = 1 << l;
    hipLaunchKernelGGL(k_dd_level_init, dim3(1), dim3(64), 0, s, D->cells, l, ctx->dbbmin, ctx->dbbmax, ctx->cN, ctx->ndim);
    for (int round = 0; round < 2; round++) {
      GH_CHECK(ctx, hipMemsetAsync(D->hist, 0, sizeof(int)*(size_t) nc*DD_G, s));
      hipLaunchKernelGGL(k_dd_hist, dim3(nb), dim3(256), 0, s, own, D->topcell, D->cells, round, D->hist);
      DD_OP(ctx, dd_allgather(ctx, D->hist, D->hist_all, sizeof(int)*(size_t) nc*DD_G));
      hipLaunchKernelGGL(k_dd_scan, dim3(nc), dim3(1024), 0, s, D->cells, D->hist_all, nc, W, round);
    }
    GH_CHECK(ctx, hipMemsetAsync(D->cand, 0, sizeof(DDCand)*(size_t) nc*(1 + DD_CAPL), s));
    hipLaunchKernelGGL(k_dd_collect, dim3(nb), dim3(256), 0, s, own, D->topcell, D->cells, D->cand, ctx->d_flags);
    DD_OP(ctx, dd_allgather(ctx, D->cand, D->cand_all, sizeof(DDCand)*(size_t) nc*(1 + DD_CAPL)));
    hipLaunchKernelGGL(k_dd_select, dim3(nc), dim3(1024), 0, s, D->cells, D->cand_all, nc, W, ctx->dbbmin, ctx->dbbmax, ctx->kdiv, ctx->d_flags);
    hipLaunchKernelGGL(k_dd_assign, dim3(nb), dim3(256), 0, s, own, D->topcell, D->cells);
  }

  // migration: topcell is now the destination rank
  GH_CHECK(ctx, hipMemsetAsync(D->mig_cnt, 0, sizeof(int)*4*GH_MAX_RANKS, s));
  hipLaunchKernelGGL(k_mig_count, dim3(nb), dim3(256), 0, s, pn, D->topcell, ctx->rank, D->mig_cnt, D->mig_slot, D->mig_hole);
  // counts: leavers per destination -> every rank learns its arrivals per source
  DD_OP(ctx, dd_allgather(ctx, D->mig_cnt, D->hist_all, sizeof(int)*GH_MAX_RANKS));
  std::vector<int> all((size_t) W*GH_MAX_RANKS);
  GH_CHECK(ctx, hipMemcpyAsync(all.data(), D->hist_all, sizeof(int)*all.size(), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  int64_t sb[GH_MAX_RANKS], rb[GH_MAX_RANKS];
  MigTab tab;
  long long nsend = 0, nrecv = 0;
  for (int r = 0; r < W; r++) {
    const int out = all[(size_t) ctx->rank*GH_MAX_RANKS + r], in = all[(size_t) r*GH_MAX_RANKS + ctx->rank];
    tab.off[r] = (int) nsend;
    sb[r] = (int64_t) out*DD_REC*sizeof(double); rb[r] = (int64_t) in*DD_REC*sizeof(double);
    nsend += out; nrecv += in;
  }
  if (nsend != nrecv) return gh_fail(ctx, GH_ERR_INVALID, "multi-GPU: unbalanced migration (equal coordinates at a top-level split?)");
  for (int f = 0; f < DD_REC - 1; f++) tab.fld[f] = own.f[f];
  static_assert(DD_REC - 1 == D_COUNT_BASE, "migration record = the fields of a global-timestep run + iorig");
  if (nsend > 0) hipLaunchKernelGGL(k_mig_pack, dim3(nb), dim3(256), 0, s, tab, own.iorig, pn, D->topcell, ctx->rank, D->mig_slot, D->mig_send);
  DD_OP(ctx, D->ops.alltoallv(D->ops.user, D->mig_send, sb, D->mig_recv, rb, (void*) s));     // collective: every rank calls it
  if (nrecv > 0) hipLaunchKernelGGL(k_mig_unpack, dim3(cdiv(nrecv, 256)), dim3(256), 0, s, tab, own.iorig, (int) nrecv, D->mig_hole, D->mig_recv);
  D->migrated = nsend;
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

// the ranks' fine geometry tables (what the halo selection tests against)
static int dd_publish_fine(gh_ctx *ctx, double widen)
{
  gh_dd *D = ctx->dd;
  const double kr = (ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0;
  const int PF = D->P + D->F;
  hipLaunchKernelGGL(k_pub_fine, dim3(1 << PF), dim3(256), 0, ctx->stream, gh_dev(ctx), ctx->L, PF, ctx->rank, kr, widen, D->fine);
  DD_OP(ctx, dd_allgather(ctx, D->fine, D->fine_all, sizeof(LetGeomF)*((size_t) 1 << PF)));
  D->fine_widen = widen;
  D->fine_base = (const char*) D->fine_all; D->fine_stride = sizeof(LetGeomF)*((size_t) 1 << PF);
  return GH_OK;
}

// all-gather of the top P levels of every rank's subtree, then the levels above the ranks' cells
int gh_dd_publish(gh_ctx *ctx, int hmax_only)
{
  gh_dd *D = ctx->dd;
  const int W = ctx->nranks, L = ctx->L, P = D->P;
  const int ncell = (2 << P) - 1;
  DevicePtrs d = gh_dev(ctx);
  const double kr = (ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0;
  // subtree tops and fine geometry tables travel in ONE all-gather: per rank [PubRec x ncell | LetGeom x 2^(P+F)]
  const int PF = P + D->F;
  const size_t fine_bytes = sizeof(LetGeomF)*((size_t) 1 << PF), blk = D->pub_bytes + fine_bytes;
  hipLaunchKernelGGL(k_pub_pack, dim3(cdiv(ncell, 64)), dim3(64), 0, ctx->stream, d, L, P, ctx->rank, (PubRec*) D->comb_send);
  hipLaunchKernelGGL(k_pub_fine, dim3(1 << PF), dim3(256), 0, ctx->stream, d, L, PF, ctx->rank, kr, 1.0, (LetGeomF*) (D->comb_send + D->pub_bytes));
  DD_OP(ctx, dd_allgather(ctx, D->comb_send, D->comb_recv, blk));
  D->fine_widen = 1.0; D->fine_base = D->comb_recv + D->pub_bytes; D->fine_stride = blk;
  hipLaunchKernelGGL(k_pub_unpack, dim3(cdiv(ncell*W, 256)), dim3(256), 0, ctx->stream, d, L, P, ctx->rank, W, D->comb_recv, blk);
  gh_stock_top_levels(ctx, L - 1, hmax_only);
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

extern "C" int gh_allgather_multipoles(gh_ctx *ctx)
{
  if (!ctx || !ctx->tree_valid) return GH_ERR_INVALID;
  if (ctx->nranks == 1) return GH_OK;
  int rc = gh_dd_publish(ctx, 0);
  if (rc) return rc;
  return gh_sync_collect(ctx, "gh_allgather_multipoles");
}

int gh_dd_exchange(gh_ctx *ctx, int phase) { return gh_dd_exchange_margin(ctx, phase, 1.0); }

int gh_dd_any(gh_ctx *ctx, const unsigned int *count_dev, int *any)
{
  gh_dd *D = ctx->dd;
  DD_OP(ctx, dd_allgather(ctx, count_dev, D->hist_all, sizeof(int)));
  std::vector<int> all((size_t) ctx->nranks);
  GH_CHECK(ctx, hipMemcpyAsync(all.data(), D->hist_all, sizeof(int)*all.size(), hipMemcpyDeviceToHost, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  *any = 0;
  for (int v : all) if (v) *any = 1;
  return GH_OK;
}

int gh_dd_exchange_margin(gh_ctx *ctx, int phase, double widen)
{
  if (ctx->nranks == 1) return GH_OK;
  gh_dd *D = ctx->dd;
  const int W = ctx->nranks, L = ctx->L, P = D->P;
  hipStream_t s = ctx->stream;
  DevicePtrs d = gh_dev(ctx);
  const double kr = (ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0;
  if (((size_t) 1 << (ctx->ltot - (L + P))) > (size_t) DD_VIS)
    return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU: more than DD_VIS leaves below a published cell (k_let_mark keeps one visit flag per cell of a level in LDS)");
  if (phase == GH_HALO_DENSITY && widen != D->fine_widen) { const int rc = dd_publish_fine(ctx, widen); if (rc) return rc; }
  hipLaunchKernelGGL(k_let_invalidate, dim3(cdiv(ctx->Ncell, 256)), dim3(256), 0, s, d, L, P, ctx->rank, ctx->Ncell);
  GH_CHECK(ctx, hipMemsetAsync(D->let_cnt, 0, sizeof(int)*8*GH_MAX_RANKS, s));
  const dim3 grid(1 << P, W);
  LetPeriod per;
  for (int k = 0; k < 3; k++)
    per.len[k] = (k < ctx->ndim && ctx->cfg.boundary_lhs[k] == GH_BOUNDARY_PERIODIC) ? ctx->cfg.boxmax[k] - ctx->cfg.boxmin[k] : 0.0;
  if (phase == GH_HALO_DENSITY) hipLaunchKernelGGL((k_let_mark<GH_HALO_DENSITY>), grid, dim3(1024), 0, s, d, L, P, D->F, ctx->rank, kr, widen, D->fine_base, D->fine_stride, D->let_cnt, D->let_cells, D->let_leaves, D->let_cellcap, D->let_leafcap, ctx->d_flags, per);
  else if (phase == GH_HALO_HYDRO) hipLaunchKernelGGL((k_let_mark<GH_HALO_HYDRO>), grid, dim3(1024), 0, s, d, L, P, D->F, ctx->rank, kr, widen, D->fine_base, D->fine_stride, D->let_cnt, D->let_cells, D->let_leaves, D->let_cellcap, D->let_leafcap, ctx->d_flags, per);
  else hipLaunchKernelGGL((k_let_mark<GH_HALO_GRAVITY>), grid, dim3(1024), 0, s, d, L, P, D->F, ctx->rank, kr, widen, D->fine_base, D->fine_stride, D->let_cnt, D->let_cells, D->let_leaves, D->let_cellcap, D->let_leafcap, ctx->d_flags, per);
  // counts to everybody (2 ints per pair), then sizes on the host
  DD_OP(ctx, dd_allgather(ctx, D->let_cnt, D->hist_all, sizeof(int)*2*GH_MAX_RANKS));
  std::vector<int> all((size_t) W*2*GH_MAX_RANKS);
  GH_CHECK(ctx, hipMemcpyAsync(all.data(), D->hist_all, sizeof(int)*all.size(), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  LetLayout lay;
  lay.phase = phase; lay.quad = ctx->cquad ? 1 : 0; lay.occ = ctx->leafocc;
  lay.cell_dbl = phase == GH_HALO_DENSITY ? 9 : (lay.quad ? 34 : 29);
  lay.part_dbl = phase == GH_HALO_DENSITY ? 4 : 20;
  lay.leaf_dbl = 1 + lay.occ*lay.part_dbl;
  int64_t sb[GH_MAX_RANKS], rb[GH_MAX_RANKS];
  long long soff[GH_MAX_RANKS], roff[GH_MAX_RANKS], stot = 0, rtot = 0, nimp = 0;
  int rcnt[2*GH_MAX_RANKS];
  for (int r = 0; r < W; r++) {
    const int oc = r == ctx->rank ? 0 : all[(size_t) ctx->rank*2*GH_MAX_RANKS + 2*r], ol = r == ctx->rank ? 0 : all[(size_t) ctx->rank*2*GH_MAX_RANKS + 2*r + 1];
    const int ic = r == ctx->rank ? 0 : all[(size_t) r*2*GH_MAX_RANKS + 2*ctx->rank], il = r == ctx->rank ? 0 : all[(size_t) r*2*GH_MAX_RANKS + 2*ctx->rank + 1];
    soff[r] = stot; roff[r] = rtot;
    const long long sd = (long long) oc*lay.cell_dbl + (long long) ol*lay.leaf_dbl, rd = (long long) ic*lay.cell_dbl + (long long) il*lay.leaf_dbl;
    sb[r] = sd*8; rb[r] = rd*8; stot += sd; rtot += rd;
    rcnt[2*r] = ic; rcnt[2*r + 1] = il;
    nimp += (long long) il*lay.occ;
  }
  if (phase != GH_HALO_DENSITY) D->held_particles = ctx->own_count + nimp;
  if (getenv("GH_DD_DEBUG")) {
    fprintf(stderr, "[dd] rank %d phase %d widen %.1f:", ctx->rank, phase, widen);
    for (int r = 0; r < W; r++) if (r != ctx->rank) fprintf(stderr, "  to %d: %d cells %d leaves | from %d: %d cells %d leaves", r,
        all[(size_t) ctx->rank*2*GH_MAX_RANKS + 2*r], all[(size_t) ctx->rank*2*GH_MAX_RANKS + 2*r + 1], r, rcnt[2*r], rcnt[2*r + 1]);
    fprintf(stderr, "  (subtree: %d cells, %d leaves)\n", 2*(ctx->gtot >> L) - 1, ctx->gtot >> L);
  }
  auto grow = [&](char **p, size_t *have, size_t need) -> hipError_t {
    if (need <= *have) return hipSuccess;
    if (*p) (void) hipFree(*p);
    *p = nullptr; *have = 0;
    const size_t cap = need + need/4 + 4096;
    hipError_t e = hipMalloc((void**) p, cap);
    if (e == hipSuccess) *have = cap;
    return e;
  };
  GH_CHECK(ctx, grow(&D->let_send, &D->let_send_bytes, (size_t) stot*8 + 8));
  GH_CHECK(ctx, grow(&D->let_recv, &D->let_recv_bytes, (size_t) rtot*8 + 8));
  // received counts and block offsets (in doubles) for the pack / unpack kernels
  GH_CHECK(ctx, hipMemcpyAsync(D->let_cnt + 2*GH_MAX_RANKS, rcnt, sizeof(int)*2*GH_MAX_RANKS, hipMemcpyHostToDevice, s));
  long long hoff[2*GH_MAX_RANKS];
  for (int r = 0; r < GH_MAX_RANKS; r++) { hoff[r] = r < W ? soff[r] : 0; hoff[GH_MAX_RANKS + r] = r < W ? roff[r] : 0; }
  long long *d_offs = D->let_off;
  GH_CHECK(ctx, hipMemcpyAsync(d_offs, hoff, sizeof(hoff), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_let_pack, dim3(256, W), dim3(256), 0, s, d, lay, W, ctx->rank, D->let_cnt, D->let_cells, D->let_leaves,
                     D->let_cellcap, D->let_leafcap, d_offs, (double*) D->let_send);
  DD_OP(ctx, D->ops.alltoallv(D->ops.user, D->let_send, sb, D->let_recv, rb, (void*) s));
  hipLaunchKernelGGL(k_let_unpack, dim3(256, W), dim3(256), 0, s, d, lay, W, ctx->rank, D->let_cnt + 2*GH_MAX_RANKS, d_offs + GH_MAX_RANKS, (const double*) D->let_recv);
  GH_CHECK(ctx, hipStreamSynchronize(s));                // hoff / rcnt are stack arrays of this call
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

extern "C" int gh_exchange_halo(gh_ctx *ctx, int phase)
{
  if (!ctx || !ctx->tree_valid || phase < GH_HALO_DENSITY || phase > GH_HALO_GRAVITY) return GH_ERR_INVALID;
  int rc = gh_dd_exchange(ctx, phase);
  if (rc) return rc;
  return gh_sync_collect(ctx, "gh_exchange_halo");
}

// global timestep: minimum over the ranks' minima (Simulation.cpp:1738)
int gh_dd_min_dt(gh_ctx *ctx)
{
  if (ctx->nranks == 1) return GH_OK;
  gh_dd *D = ctx->dd;
  DD_OP(ctx, dd_allgather(ctx, gh_time_dev(ctx) + 1, D->dt_all, sizeof(double)));
  hipLaunchKernelGGL(k_dd_min_dt, dim3(1), dim3(1), 0, ctx->stream, D->dt_all, ctx->nranks, gh_time_dev(ctx));
  return GH_OK;
}

extern "C" int gh_comm_info(gh_ctx *ctx, int64_t *own_first, int64_t *own_count, int64_t *held)
{
  if (!ctx) return GH_ERR_INVALID;
  if (own_first) *own_first = ctx->own_first;
  if (own_count) *own_count = ctx->own_count;
  if (held) *held = ctx->dd && ctx->nranks > 1 ? ctx->dd->held_particles : ctx->N;
  return GH_OK;
}
