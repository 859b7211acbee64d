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
#define DD_WCAP 512          /* speculative splits: candidates per rank and cell inside the window around last step's median */
#define DD_WLDS 2048         /* ... candidates of one cell ranked out of LDS (more: out of the gathered blocks) */
#define DD_WHDR 4            /* ... header slots of a rank's block (the rank's particle extent rides in them) */
#define DD_PMAX 5            /* published levels per rank subtree (2^(P+1) - 1 cells) */
#define DD_FMAX 6            /* halo selection: 2^F fine geometry entries per published bottom cell */
#define DD_RECMAX (D_COUNT + 1)   /* doubles per migrating particle: the fields that are live (D_COUNT_BASE, or all with block timesteps) + iorig */

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
struct DDWin { double lo, hi; int kd, pad; };         // window of one cell of the level being split: lo <= x <= hi travel as candidates
struct LetGeomF;

struct gh_dd {
  gh_comm_ops ops;
  int P = 0;                         // published levels below a rank's cell
  int *topcell = nullptr;            // [own_count] cell index (within its level) of every own particle during gh_dd_decompose
  DDCell *cells = nullptr;           // [nranks]
  int *hist = nullptr, *hist_all = nullptr;          // [nranks/2][DD_G], [nranks][nranks/2][DD_G]
  DDCand *cand = nullptr, *cand_all = nullptr;       // [nranks/2][1 + DD_CAPL], [nranks][...]  (slot 0: count in .id)
  double *box6 = nullptr, *box6_all = nullptr;       // local extent, all ranks' extents
  // speculative splits (one collective per level): last step's median, window half-width and split axis of every
  // shared top cell; per-level window table; the ranks' window blocks; status word (non-zero: redo in exact mode)
  double *spl_prev = nullptr, *spl_win = nullptr; int *spl_kd = nullptr, *spl_fail = nullptr;
  struct DDWin *wins = nullptr;
  DDCand *wnd = nullptr, *wnd_all = nullptr;
  // the last force-phase exchange, for the way back of levelneib: leaves sent to / cells and leaves received from every rank,
  // receive offsets (doubles), record layout
  int fwd_ol[GH_MAX_RANKS] = {0}, fwd_ic[GH_MAX_RANKS] = {0}, fwd_il[GH_MAX_RANKS] = {0};
  long long fwd_roff[GH_MAX_RANKS] = {0};
  int fwd_lay_cell = 0, fwd_lay_leaf = 0, fwd_occ = 0;
  char *ln_send = nullptr, *ln_recv = nullptr; size_t ln_send_bytes = 0, ln_recv_bytes = 0;
  long long *ln_roff = nullptr;                       // device copy of fwd_roff
  bool have_splits = false;
  long long n_spec = 0, n_exact = 0;                  // decompositions done speculatively / with the three-collective search
  // pinned host staging of the halo exchange (sizes in, counts and offsets out): no stack array is read by an async copy
  int *h_cnt = nullptr; long long *h_off = nullptr; int *h_all = nullptr;
  // migration
  int *mig_cnt = nullptr;            // [2*nranks + 4]: leavers per destination, arrivals per source, cursor words
  int *mig_slot = nullptr;           // [own_count] position of every leaver in its destination's block
  int *mig_hole = nullptr;           // [own_count] positions freed by leavers
  double *mig_send = nullptr, *mig_recv = nullptr;   // [own_count][DD_RECMAX]
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
  struct LetWork *let_work = nullptr;                  // [nranks << (P + F)] work items of the marking walk
  unsigned char *let_vis = nullptr; size_t vis_stride = 0;   // [nranks][cells of the own subtree] visit flags of k_let_walk
  size_t let_cellcap = 0, let_leafcap = 0;
  char *let_send = nullptr, *let_recv = nullptr; size_t let_send_bytes = 0, let_recv_bytes = 0;
  long long migrated = 0;            // particles this rank sent away in the last decomposition
  long long held_particles = 0;      // own + imported (last force-phase exchange): what this rank actually holds
  double dt_local[2];
  double *dt_all = nullptr;
  // gh_dd_gatherv: ragged all-gather of small host-bound records (sink runs)
  char *gv_send = nullptr, *gv_recv = nullptr; size_t gv_send_bytes = 0, gv_recv_bytes = 0;
  long long *gv_cnt = nullptr;       // [1 + nranks] device
  int alloc_gtot = 0, alloc_lgroup = 0; size_t alloc_n = 0;            // what the buffers below were sized for (sink runs: N shrinks)
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
    else {
      q.binB = bsel;
      // particles in the coarse bin (the fine histogram counts only those): the density the next step's window is sized by
      long long tot = 0;
      for (int b = 0; b < DD_G; b++) tot += s_cum[b];
      q.pad = (int) (tot > 0x7fffffff ? 0x7fffffff : tot);
    }
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
                                                    double *dbbmin, double *dbbmax, int *kdiv, int *flags,
                                                    double *spl_prev, double *spl_win, int *spl_kd, int *tie)
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
  // equal coordinates on both sides of the split: which of them go left is decided here by particle id, in the
  // reference by the dynamics of its quick-select (KDTree.cpp:682-750) - reported, see gh_sync_collect
  if (s_found >= 0) {
    const DDCand m = entry(s_found);
    for (int e = threadIdx.x; e < ntot; e += blockDim.x) { const DDCand o = entry(e); if (o.key == m.key && o.id < m.id) atomicOr(tie, 1); }
  }
  if (threadIdx.x == 0) {
    DDCell w = q;
    const int n = q.node, c1 = 2*n + 1, c2 = 2*n + 2, kd = q.kd;
    if (s_found >= 0) { const DDCand m = entry(s_found); w.rdiv = m.key; w.rdiv_id = m.id; }
    else if (ntot > 0 || q.target > 0) atomicOr(flags, FLAG_DD_SPLIT);       // the bracket lost the median: never expected
    // what next step's speculative split starts from: the median, the axis, and a window that holds ~256 candidates at
    // the density of the final bin (ntot candidates in a bin of width 1/scB)
    spl_prev[n] = w.rdiv; spl_kd[n] = kd;
    spl_win[n] = (q.scB > 0.0 && ntot > 0) ? 128.0/(q.scB*(double) ntot) : 0.0;
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
// speculative splits: ONE collective per shared level (and none for the root box).  The median of a top cell moves by a
// few dozen particle ranks per step, so every rank sends, per cell, the number of its particles below a narrow window
// around LAST step's median and the (coordinate, id) of those inside it; every rank then finds the exact median among
// the gathered candidates - the same element the histogram search below finds.  If a window misses its median, a rank
// overflows its block or a cell's longest axis changed, a status word is raised, rides in the migration counts to every
// rank, and all ranks redo the decomposition with the exact search (gh_dd_decompose).  Windows adapt: wide enough for
// four times the last shift and for ~256 candidates at the density seen.
// ------------------------------------------------------------------------------------------------
__global__ void k_dd_win_init(DDCell *cells, DDWin *wins, int level, const double *dbbmin, const double *dbbmax, const int *cN, int ndim,
                              const double *spl_prev, const double *spl_win, const int *spl_kd, int *fail, double winscale)
{
  const int c = threadIdx.x;
  if (c >= (1 << level)) return;
  const int n = (1 << level) - 1 + c;
  int kd = spl_kd[n];
  double rkmax = 0.0;
  if (level > 0) {                                     // the root's box is only known after the gather (k_dd_wselect checks it)
    int k2 = 0;
    for (int k = 0; k < ndim; k++) { const double ext = dbbmax[n*3 + k] - dbbmin[n*3 + k]; if (ext > rkmax) { rkmax = ext; k2 = k; } }
    if (k2 != kd) { atomicOr(fail, 1); kd = k2; }
  }
  DDCell q;
  q.kd = kd; q.node = n;
  q.loA = 0.0; q.scA = 0.0; q.loB = 0.0; q.scB = 0.0; q.binA = -1; q.binB = -1;
  q.base = 0; q.target = cN[n]/2;
  q.rdiv = spl_prev[n]; q.rdiv_id = -1; q.pad = 0;
  cells[c] = q;
  DDWin w;
  w.lo = spl_prev[n] - winscale*spl_win[n]; w.hi = spl_prev[n] + winscale*spl_win[n]; w.kd = kd; w.pad = 0;
  wins[c] = w;
}

// block layout per rank: [DD_WHDR header slots | per cell: 1 count slot (key bits = particles below the window, id =
// particles inside) + DD_WCAP candidates]
__global__ __launch_bounds__(256) void k_dd_window(DevicePtrs d, const int *topcell, const DDWin *wins, int ncells, DDCand *blk, int *fail)
{
  __shared__ unsigned int s_below[GH_MAX_RANKS];
  if (threadIdx.x < GH_MAX_RANKS) s_below[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < d.N) {
    const int c = topcell[i];
    const DDWin w = wins[c];
    const double x = d.f[D_RX + w.kd][i];
    if (x < w.lo) atomicAdd(&s_below[c], 1u);
    else if (x <= w.hi) {
      DDCand *cc = blk + DD_WHDR + (size_t) c*(1 + DD_WCAP);
      const int slot = atomicAdd(&cc[0].id, 1);
      if (slot < DD_WCAP) { cc[1 + slot].key = x; cc[1 + slot].id = d.iorig[i]; cc[1 + slot].pad = 0; }
    }
  }
  __syncthreads();
  if ((int) threadIdx.x < ncells && s_below[threadIdx.x])
    atomicAdd((unsigned long long*) &blk[DD_WHDR + (size_t) threadIdx.x*(1 + DD_WCAP)].key, (unsigned long long) s_below[threadIdx.x]);
}

__global__ void k_dd_win_box(const double *dbbmin, const double *dbbmax, DDCand *blk)
{
  if (threadIdx.x < 3) { ((double*) blk)[threadIdx.x] = dbbmin[threadIdx.x]; ((double*) blk)[3 + threadIdx.x] = dbbmax[threadIdx.x]; }
}

// one workgroup per cell of the level: exact median among the gathered window candidates
__global__ __launch_bounds__(1024) void k_dd_wselect(DDCell *cells, const DDCand *all, size_t stride /* DDCand per rank */, int level, int nranks, int ndim,
                                                     double *dbbmin, double *dbbmax, int *kdiv, double *spl_prev, double *spl_win, int *spl_kd, int *fail, int *tie)
{
  const int c = blockIdx.x;
  __shared__ int s_off[GH_MAX_RANKS + 1];
  __shared__ long long s_base;
  __shared__ int s_found, s_bad;
  __shared__ int s_cnt[GH_MAX_RANKS];
  __shared__ long long s_below[GH_MAX_RANKS];
  __shared__ double s_box[6];
  if ((int) threadIdx.x < nranks) {                      // the ranks' headers side by side: one memory latency, not nranks
    const DDCand h = all[(size_t) threadIdx.x*stride + DD_WHDR + (size_t) c*(1 + DD_WCAP)];
    s_cnt[threadIdx.x] = h.id; s_below[threadIdx.x] = (long long) __double_as_longlong(h.key);
  }
  if (level == 0 && threadIdx.x >= 64 && threadIdx.x < 70) {
    // global root box from the ranks' extents (KDTree.cpp:269-280)
    const int k = threadIdx.x - 64;
    double v = k < 3 ? 9.9e20 : -9.9e20;
    for (int r = 0; r < nranks; r++) { const double x = ((const double*) (all + (size_t) r*stride))[k]; v = k < 3 ? fmin(v, x) : fmax(v, x); }
    s_box[k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0; long long base = 0; int bad = 0;
    for (int r = 0; r < nranks; r++) {
      s_off[r] = run;
      if (s_cnt[r] > DD_WCAP) bad = 2;                       // (status bits: 1 axis changed, 2 a rank's window overflowed, 4 root axis, 8 median outside the windows)
      run += min(s_cnt[r], DD_WCAP);
      base += s_below[r];
    }
    s_off[nranks] = run; s_base = base; s_found = -1; s_bad = bad;
    if (level == 0) {
      // ... then the root's longest axis against the guess
      double rkmax = 0.0; int kd = 0;
      for (int k = 0; k < 3; k++) {
        dbbmin[k] = s_box[k]; dbbmax[k] = s_box[3 + k];
        if (k < ndim && s_box[3 + k] - s_box[k] > rkmax) { rkmax = s_box[3 + k] - s_box[k]; kd = k; }
      }
      if (kd != cells[0].kd) s_bad |= 4;
    }
  }
  __syncthreads();
  const DDCell q = cells[c];
  const int ntot = s_off[nranks];
  const long long want = q.target - s_base;
  auto entry = [&](int e) -> DDCand {
    int r = 0;
    while (r + 1 < nranks && e >= s_off[r + 1]) r++;
    return all[(size_t) r*stride + DD_WHDR + (size_t) c*(1 + DD_WCAP) + 1 + (e - s_off[r])];
  };
  // rank of every candidate among all of them: the candidates (a few hundred as a rule) are first copied to LDS, where
  // the ntot^2 comparisons are broadcast reads
  __shared__ double s_key[DD_WLDS];
  __shared__ int s_id[DD_WLDS];
  const bool inlds = ntot <= DD_WLDS;
  if (!s_bad && want >= 0 && want < ntot && inlds) {
    for (int e = threadIdx.x; e < ntot; e += blockDim.x) { const DDCand me = entry(e); s_key[e] = me.key; s_id[e] = me.id; }
  }
  __syncthreads();
  if (!s_bad && want >= 0 && want < ntot) {
    for (int e = threadIdx.x; e < ntot; e += blockDim.x) {
      long long rank = 0;
      if (inlds) {
        const double mk = s_key[e]; const int mi = s_id[e];
        for (int t = 0; t < ntot; t++) { const double ok = s_key[t]; if (ok < mk || (ok == mk && s_id[t] < mi)) rank++; }
      }
      else {
        const DDCand me = entry(e);
        for (int r = 0; r < nranks; r++) {
          const DDCand *p = all + (size_t) r*stride + DD_WHDR + (size_t) c*(1 + DD_WCAP) + 1;
          const int nr = s_off[r + 1] - s_off[r];
          for (int t = 0; t < nr; t++) { const DDCand o = p[t]; if (o.key < me.key || (o.key == me.key && o.id < me.id)) rank++; }
        }
      }
      if (rank == want) s_found = e;
    }
  }
  __syncthreads();
  // equal coordinates on both sides of the split (see k_dd_select): all threads look, each at its candidates - one thread
  // walking the gathered blocks entry by entry was 150 us of dependent global loads per level
  if (s_found >= 0) {
    const DDCand m = entry(s_found);
    for (int e = threadIdx.x; e < ntot; e += blockDim.x) {
      const double ok = inlds ? s_key[e] : entry(e).key;
      const int oi = inlds ? s_id[e] : entry(e).id;
      if (ok == m.key && oi < m.id) atomicOr(tie, 1);
    }
  }
  if (threadIdx.x == 0) {
    DDCell w = q;
    const int n = q.node, c1 = 2*n + 1, c2 = 2*n + 2, kd = q.kd;
    if (s_found >= 0) {
      const DDCand m = entry(s_found);
      w.rdiv = m.key; w.rdiv_id = m.id;
      const double wold = spl_win[n], shift = fabs(m.key - spl_prev[n]);
      double wnew = fmax(4.0*shift, 256.0*wold/(double) ntot);
      wnew = fmin(wnew, 0.25*(dbbmax[n*3 + kd] - dbbmin[n*3 + kd]));
      spl_prev[n] = m.key; spl_win[n] = wnew;
    }
    else atomicOr(fail, s_bad ? s_bad : 8);
    for (int k = 0; k < 3; k++) {
      dbbmin[c1*3 + k] = dbbmin[n*3 + k]; dbbmax[c1*3 + k] = (k == kd) ? w.rdiv : dbbmax[n*3 + k];
      dbbmin[c2*3 + k] = (k == kd) ? w.rdiv : dbbmin[n*3 + k]; dbbmax[c2*3 + k] = dbbmax[n*3 + k];
    }
    kdiv[n] = kd;
    cells[c] = w;
  }
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

struct MigTab { double *fld[DD_RECMAX]; int off[GH_MAX_RANKS]; int nf; };      // nf fields travel, then iorig

__global__ void k_mig_pack(MigTab t, const int *iorig, int n, const int *dest, int self, const int *slot, double *send)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int to = dest[i];
  if (to == self) return;
  double *o = send + (size_t) (t.off[to] + slot[i])*(t.nf + 1);
  for (int f = 0; f < t.nf; f++) o[f] = t.fld[f][i];
  o[t.nf] = (double) iorig[i];
}

// arrival e takes the place of leaver e; arrivals beyond the leavers (a sink run after accretion elsewhere: this rank's
// cell grew) line up behind the particles held so far
__global__ void k_mig_unpack(MigTab t, int *iorig, int narr, const int *hole, const double *recv, int nholes, int held)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= narr) return;
  const int i = e < nholes ? hole[e] : held + (e - nholes);
  const double *o = recv + (size_t) e*(t.nf + 1);
  for (int f = 0; f < t.nf; f++) t.fld[f][i] = o[f];
  iorig[i] = (int) o[t.nf];
}

// fewer arrivals than leavers: the particles behind the new end move into the holes that stayed open
__global__ void k_mig_move(MigTab t, int *iorig, const int *from, const int *to, int n)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int a = from[e], b = to[e];
  for (int f = 0; f < t.nf; f++) t.fld[f][b] = t.fld[f][a];
  iorig[b] = iorig[a];
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

// (the spare word of the first record carries this rank's "a median split separated equal coordinates" flag, so that
// every rank learns of a tie anywhere and all of them stop together in gh_sync_collect)
__global__ void k_pub_pack(DevicePtrs d, int L, int P, int rank, PubRec *out, const int *tie)
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
  if (e == 0) r.q.pad[0] = (tie[0] | tie[1]) ? 1.0 : 0.0;
  out[e] = r;
}

__global__ void k_pub_unpack(DevicePtrs d, int L, int P, int self, int nranks, const char *all, size_t stride, int *tie)
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
  if (e == 0 && q.q.pad[0] != 0.0) atomicOr(&tie[1], 1);
  if (d.cquad) { CellQuad cq = q.q; cq.pad[0] = 0.0; d.cquad[n] = cq; }
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
    bool o = true;                                       // (fixed trip count + predicate: Q stays in registers)
#pragma unroll
    for (int k = 0; k < 3; k++) if (k < ndim && (Q.dbmin[k] > yb.bbmax[k] || yb.bbmin[k] > Q.dbmax[k])) o = false;
    return o;
  }
  if (PHASE == GH_HALO_HYDRO) {
    bool o1 = true, o2 = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k >= ndim) continue;
      if (Q.bbmin[k] > yh.hbmax[k] || yh.hbmin[k] > Q.bbmax[k]) o1 = false;
      if (Q.hbmin[k] > yb.bbmax[k] || yb.bbmin[k] > Q.hbmax[k]) o2 = false;
    }
    return o1 || o2;
  }
  double dr2 = 0.0, dc2 = 0.0, db2 = 0.0;            // squared distance of Y's centre from the three boxes
#pragma unroll
  for (int k = 0; k < 3; k++) {
    if (k >= ndim) continue;
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

// Marking = the destination's own walks, run on the sender at the granularity of the destination's fine table
// (2^(P+F) entries per rank: its particle groups, or small bundles of them).
//   k_let_prepass  one thread per (destination rank, fine entry): which of THIS rank's 2^P published bottom cells can
//                  that entry's walks open?  Entries that open none (most of them: the far side of the other domains)
//                  cost nothing further; the others become work items (rank, entry, start-cell mask).
//   k_let_walk     persistent wavefronts take work items off a queue: a depth-first walk of this rank's subtree below the
//                  item's start cells (stack in LDS, up to 64 nodes popped per round, lane = node) with the conservative
//                  opening test of the phase.  An entry only ever follows the nodes it opened itself, so no (entry, node)
//                  pair is tested unless the destination's walks really can get near that node - the level-by-level
//                  version this replaces tested every visited node against the destination's whole table (32 published
//                  cells x 64 fine entries, 75 ms per launch for 1M particles on two ranks).
// Measured on the way (8 ranks, 1M Plummer): one wave per (rank, entry) without the prepass cost 180 / 275 us (density /
// gravity phase) - the sum of 14 336 wave lifetimes over the ~3 000 waves the device holds, whatever the walks did; more,
// smaller walks (one per start-cell chunk) scaled with the wave count (x8 waves: 520 / 960 us), bundles of 8 entries per
// wave with a union pre-test lost in the density phase what they won in the gravity phase (union boxes are loose).
// Visited cells get a flag byte in vis[r][subtree-local heap index] (1 = cell record travels, 2 = the leaf's particles
// travel; plain idempotent byte stores, every writer of a byte writes the same value); k_let_compact turns the flags into
// the per-destination id lists.
#define LW_CAP 2048
struct LetWork { int r, e; unsigned int mask; int pad; };
__device__ __forceinline__ int let_loc(int n, int L, int self)
{
  const int lev = 31 - __clz(n + 1), rl = lev - L;
  return (1 << rl) - 1 + (n - ((1 << lev) - 1) - (self << rl));
}
template <int PHASE, bool PER>
__device__ __forceinline__ bool let_test(const LetGeom &Q, const DevicePtrs &d, int n, double kernrange, double widen, const LetPeriod &per)
{
  CellBox yb; CellH yh; CellGeo yg;
  if (PHASE == GH_HALO_GRAVITY) { yg = d.cgeo[n]; yb.N = yg.N; }
  else { yb = d.cbox[n]; if (PHASE == GH_HALO_HYDRO) yh = d.ch[n]; }
  if (yb.N <= 0) return false;
  if (PER) return let_may_open_img<PHASE>(Q, yb, yh, yg, d.ndim, kernrange, widen, per);
  return let_may_open<PHASE>(Q, yb, yh, yg, d.ndim, kernrange, widen);
}

template <int PHASE, bool PER>
__global__ __launch_bounds__(256) void k_let_prepass(DevicePtrs d, int L, int P, int PF, int self, int nranks, double kernrange, double widen,
                                                     const char *fine_base, size_t fine_stride, LetPeriod per, LetWork *work, int *nwork)
{
  const int t = blockIdx.x*blockDim.x + threadIdx.x;
  const int r = t >> PF, e = t & ((1 << PF) - 1);
  if (r >= nranks || r == self) return;
  LetGeom Q;
  let_expand(((const LetGeomF*) (fine_base + (size_t) r*fine_stride))[e], Q);
  if (Q.N <= 0) return;
  const int y0 = (1 << (L + P)) - 1 + (self << P);
  unsigned int mask = 0;
  for (int c = 0; c < (1 << P); c++)
    if (let_test<PHASE, PER>(Q, d, y0 + c, kernrange, widen, per)) mask |= 1u << c;
  if (mask) { LetWork w; w.r = r; w.e = e; w.mask = mask; w.pad = 0; work[atomicAdd(nwork, 1)] = w; }
}

template <int PHASE, bool PER, int NPL /* nodes per lane and round */, int CAP /* stack entries */>
__global__ __launch_bounds__(64) void k_let_walk(DevicePtrs d, int L, int P, int self, double kernrange, double widen,
                                                 const char *fine_base, size_t fine_stride, unsigned char *vis, size_t vis_stride, LetPeriod per,
                                                 const LetWork *work, const int *nwork, int *head)
{
  __shared__ int s_stack[CAP + 128*NPL];
  __shared__ int s_item;
  const int lane = threadIdx.x;
  const int nw = *nwork;
  const int depth = d.ltot - (L + P);                    // levels below the published bottom cells
  const int y0 = (1 << (L + P)) - 1 + (self << P);
  for (;;) {
    if (lane == 0) s_item = atomicAdd(head, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= nw) break;                               // every wave ends here: the queue is finite and only ever drained
    const LetWork w = work[item];
    LetGeom Q;
    let_expand(((const LetGeomF*) (fine_base + (size_t) w.r*fine_stride))[w.e], Q);
    unsigned char *v = vis + (size_t) w.r*vis_stride;
    const bool mine = lane < (1 << P) && ((w.mask >> lane) & 1u);
    const unsigned long long sm = __ballot(mine);
    if (mine) s_stack[__popcll(sm & ((1ull << lane) - 1ull))] = y0 + lane;
    int top = __popcll(sm);
    __syncthreads();
    while (top > 0) {
      // pops narrow when the stack is nearly full: one node at a time it grows by at most one entry per level
      const int take = top > CAP - 128*NPL ? 1 : min(top, 64*NPL);
      int n[NPL];
#pragma unroll
      for (int k = 0; k < NPL; k++) n[k] = k*64 + lane < take ? s_stack[top - 1 - (k*64 + lane)] : -1;
      top -= take;
      __syncthreads();
      bool open[NPL];
#pragma unroll
      for (int k = 0; k < NPL; k++) open[k] = n[k] >= 0 && let_test<PHASE, PER>(Q, d, n[k] < 0 ? 0 : n[k], kernrange, widen, per);
#pragma unroll
      for (int k = 0; k < NPL; k++) {
        bool push = false;
        if (open[k]) {
          if (depth == 0) v[1 + let_loc(n[k], L, self)] = 2;   // the published bottom cell is a leaf: its particles travel
          else {
            // a visited leaf travels with its particles whether or not they would be touched (a leaf with one particle enters
            // the gravity lists as a particle, Tree.cpp:713): no walk below the last-but-one level
            // (the two children's flags are one aligned 16-bit word - the array is shifted by a byte; most visits find
            // them set already by another entry's walk and skip the store: the byte stores of thousands of walks into
            // the same few cache lines were what this kernel waited for)
            const bool childleaf = 31 - __clz(n[k] + 1) + 1 == d.ltot;
            const int l1 = let_loc(2*n[k] + 1, L, self);
            const unsigned short val = childleaf ? 0x0303 : 0x0101;
            unsigned short *pv = (unsigned short*) (v + 1 + l1);
            if (*pv != val) *pv = val;
            push = !childleaf;
          }
        }
        const unsigned long long pm = __ballot(push);
        if (push) {
          const int at = top + 2*__popcll(pm & ((1ull << lane) - 1ull));
          s_stack[at] = 2*n[k] + 1; s_stack[at + 1] = 2*n[k] + 2;
        }
        top += 2*__popcll(pm);
      }
      __syncthreads();
    }
  }
}

// flags -> per-destination lists of cell ids (records travel) and leaf ids (particles travel)
__global__ __launch_bounds__(256) void k_let_compact(int L, int self, const unsigned char *vis, size_t vis_stride, int nloc,
                                                     int *cnt, int *cells, int *leaves, size_t cellcap, size_t leafcap, int *flags)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int lanei = threadIdx.x & 63;
  const unsigned long long below = (1ull << lanei) - 1ull;
  for (int l0 = blockIdx.x*blockDim.x; l0 < nloc; l0 += gridDim.x*blockDim.x) {
    const int loc = l0 + threadIdx.x;
    const unsigned char val = loc < nloc ? vis[(size_t) r*vis_stride + 1 + loc] : 0;
    int n = 0;
    if (val) { const int rl = 31 - __clz(loc + 1); n = (1 << (L + rl)) - 1 + (self << rl) + (loc - ((1 << rl) - 1)); }
    const unsigned long long cm = __ballot(val & 1);
    if (cm) {
      int base = 0;
      if (lanei == 0) base = atomicAdd(&cnt[2*r], __popcll(cm));
      base = __shfl(base, 0, 64);
      if (val & 1) {
        const size_t pos = (size_t) base + __popcll(cm & below);
        if (pos < cellcap) cells[(size_t) r*cellcap + pos] = n; else atomicOr(flags, FLAG_LEAFLIST_OVERFLOW);
      }
    }
    const unsigned long long lm = __ballot(val & 2);
    if (lm) {
      int base = 0;
      if (lanei == 0) base = atomicAdd(&cnt[2*r + 1], __popcll(lm));
      base = __shfl(base, 0, 64);
      if (val & 2) {
        const size_t pos = (size_t) base + __popcll(lm & below);
        if (pos < leafcap) leaves[(size_t) r*leafcap + pos] = n; else atomicOr(flags, FLAG_LEAFLIST_OVERFLOW);
      }
    }
  }
}

// record sizes in doubles: cell = id + the records of the phase; leaf = id + occ x particle record
struct LetLayout { int cell_dbl, part_dbl, leaf_dbl, occ, phase, quad, lev, gpot; };     // lev: force records end with the particle's timestep level; gpot: density records end with last step's potential (sink search)

// pack / unpack: a thread moves ONE double of one record (cell records: up to 34 doubles; leaf records: id + occ particle
// records), so that consecutive threads touch consecutive words of the message and of the 32-byte packs
__device__ __forceinline__ const double *let_cell_word(const DevicePtrs &d, int n, int w, bool dens)
{
  // word w >= 1 of a cell record: CellBox [1, 9), then (force phases) CellH [9, 17), CellGeo [17, 25), CellCom [25, 29), CellQuad [29, 34)
  if (w < 9) return (const double*) &d.cbox[n] + (w - 1);
  if (w < 17) return (const double*) &d.ch[n] + (w - 9);
  if (w < 25) return (const double*) &d.cgeo[n] + (w - 17);
  if (w < 29) return (const double*) &d.ccom[n] + (w - 25);
  return (const double*) &d.cquad[n] + (w - 29);
}
__global__ __launch_bounds__(256) void k_let_pack(DevicePtrs d, LetLayout lay, int nranks, int self, const int *cnt, const int *cells, const int *leaves,
                                                  size_t cellcap, size_t leafcap, const long long *off /* [nranks] doubles */, double *send)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int nc = cnt[2*r], nl = cnt[2*r + 1];
  double *base = send + off[r];
  const long long ncw = (long long) nc*lay.cell_dbl, tot = ncw + (long long) nl*lay.leaf_dbl;
  for (long long e = (long long) blockIdx.x*blockDim.x + threadIdx.x; e < tot; e += (long long) gridDim.x*blockDim.x) {
    if (e < ncw) {
      const int c = (int) (e/lay.cell_dbl), w = (int) (e - (long long) c*lay.cell_dbl);
      const int n = cells[(size_t) r*cellcap + c];
      base[e] = w == 0 ? (double) n : *let_cell_word(d, n, w, lay.phase == GH_HALO_DENSITY);
    }
    else {
      const long long q = e - ncw;
      const int c = (int) (q/lay.leaf_dbl), w = (int) (q - (long long) c*lay.leaf_dbl);
      const int n = leaves[(size_t) r*leafcap + c];
      double val = 0.0;
      if (w == 0) val = (double) n;
      else {
        const int t = (w - 1)/lay.part_dbl, k = (w - 1) - t*lay.part_dbl;
        const int first = d.cfirst[n], cn = d.cN[n];
        if (t < cn) val = k < 4 ? ((const double*) &d.posm[first + t])[k] : (lay.gpot ? d.f[D_GPOT][first + t] : (k < 20 ? ((const double*) &d.hrec[4*(size_t) (first + t)])[k - 4] : d.f[D_LEVEL][first + t]));
      }
      base[e] = val;
    }
  }
}

__global__ __launch_bounds__(256) void k_let_unpack(DevicePtrs d, LetLayout lay, int nranks, int self, const int *rcnt, const long long *roff, const double *recv)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int nc = rcnt[2*r], nl = rcnt[2*r + 1];
  const double *base = recv + roff[r];
  const long long ncw = (long long) nc*lay.cell_dbl, tot = ncw + (long long) nl*lay.leaf_dbl;
  for (long long e = (long long) blockIdx.x*blockDim.x + threadIdx.x; e < tot; e += (long long) gridDim.x*blockDim.x) {
    if (e < ncw) {
      const int c = (int) (e/lay.cell_dbl), w = (int) (e - (long long) c*lay.cell_dbl);
      if (w == 0) continue;
      const int n = (int) base[(long long) c*lay.cell_dbl];
      *const_cast<double*>(let_cell_word(d, n, w, lay.phase == GH_HALO_DENSITY)) = base[e];
    }
    else {
      const long long q = e - ncw;
      const int c = (int) (q/lay.leaf_dbl), w = (int) (q - (long long) c*lay.leaf_dbl);
      if (w == 0) continue;
      const int n = (int) base[ncw + (long long) c*lay.leaf_dbl];
      const int t = (w - 1)/lay.part_dbl, k = (w - 1) - t*lay.part_dbl;
      const int first = d.cfirst[n], cn = d.cN[n];
      if (t < cn) {
        if (k < 4) ((double*) &d.posm[first + t])[k] = base[e];
        else if (lay.gpot) d.f[D_GPOT][first + t] = base[e];             // the potential-minimum test reads the neighbours' potential of the last force pass (GradhSph.cpp:270-280)
        else if (k < 20) {
          ((double*) &d.hrec[4*(size_t) (first + t)])[k - 4] = base[e];
          if (k == 4 + 7) d.f[D_HRANGESQD][first + t] = base[e];        // hrangesqd: the candidate classification of k_grav_eval reads the SoA array
        }
        else { d.f[D_LEVEL][first + t] = base[e]; d.f[D_LEVELNEIB][first + t] = 0.0; }     // what this rank's active particles raise goes back (gh_dd_return_levelneib)
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
    if (c.Nlevels > 1 && c.sph_single_timestep) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: block timesteps without sph_single_timestep");
    if (c.self_gravity && c.gravity_mac != GH_MAC_GEOMETRIC) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: gravity_mac = geometric only");
    if (c.avisc == GH_AVISC_MON97CD2010 || c.avisc == GH_AVISC_MON97MM97) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: no time-dependent viscosity");
    if (c.ntreebuildstep > 1) return gh_fail(ctx, GH_ERR_UNSUPPORTED, "multi-GPU runs: the tree is rebuilt every step (ntreebuildstep = 1)");
  }
  if (!ctx->dd) ctx->dd = new gh_dd();
  if (ops) ctx->dd->ops = *ops;
  ctx->rank = rank; ctx->nranks = nranks;
  return GH_OK;
}

// the buffers dd_alloc sizes by the tree (released when a sink run's shrinking N changes the tree's depth)
static void dd_release(gh_dd *D)
{
  void **ptrs[] = {(void**) &D->topcell, (void**) &D->cells, (void**) &D->hist, (void**) &D->hist_all, (void**) &D->cand, (void**) &D->cand_all, (void**) &D->box6,
                   (void**) &D->box6_all, (void**) &D->mig_cnt, (void**) &D->mig_slot, (void**) &D->mig_hole, (void**) &D->mig_send, (void**) &D->mig_recv,
                   (void**) &D->pub_send, (void**) &D->pub_recv, (void**) &D->comb_send, (void**) &D->comb_recv, (void**) &D->fine, (void**) &D->fine_all,
                   (void**) &D->let_cnt, (void**) &D->let_off, (void**) &D->let_cells, (void**) &D->let_leaves, (void**) &D->ln_roff, (void**) &D->dt_all,
                   (void**) &D->let_vis, (void**) &D->let_work, (void**) &D->spl_prev, (void**) &D->spl_win, (void**) &D->spl_kd, (void**) &D->wins,
                   (void**) &D->wnd, (void**) &D->wnd_all};
  for (void **p : ptrs) { if (*p) (void) hipFree(*p); *p = nullptr; }
  D->spl_fail = nullptr;
  for (void **p : {(void**) &D->h_cnt, (void**) &D->h_off, (void**) &D->h_all}) { if (*p) (void) hipHostFree(*p); *p = nullptr; }
  D->have_splits = false;
}

void gh_dd_free(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  if (!D) return;
  dd_release(D);
  void *ptrs[] = {D->let_send, D->let_recv, D->ln_send, D->ln_recv, D->gv_send, D->gv_recv, D->gv_cnt};
  for (void *p : ptrs) if (p) (void) hipFree(p);
  delete D;
  ctx->dd = nullptr;
}

static int dd_alloc(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  const size_t n = (size_t) std::max(ctx->own_count, ctx->own_held) + 1;
  if (D->topcell && D->alloc_gtot == ctx->gtot && D->alloc_lgroup == ctx->lgroup && D->alloc_n >= n) return GH_OK;
  if (D->topcell) { GH_CHECK(ctx, hipStreamSynchronize(ctx->stream)); dd_release(D); }
  D->alloc_gtot = ctx->gtot; D->alloc_lgroup = ctx->lgroup; D->alloc_n = n;
  const int W = ctx->nranks, half = std::max(W/2, 1);
  D->P = std::min(DD_PMAX, ctx->lgroup - ctx->L);
  GH_CHECK(ctx, hipMalloc((void**) &D->topcell, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->cells, sizeof(DDCell)*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->hist, sizeof(int)*(size_t) half*DD_G));
  GH_CHECK(ctx, hipMalloc((void**) &D->hist_all, sizeof(int)*(size_t) W*half*DD_G));
  GH_CHECK(ctx, hipMalloc((void**) &D->cand, sizeof(DDCand)*(size_t) half*(1 + DD_CAPL)));
  GH_CHECK(ctx, hipMalloc((void**) &D->cand_all, sizeof(DDCand)*(size_t) W*half*(1 + DD_CAPL)));
  GH_CHECK(ctx, hipMalloc((void**) &D->box6, sizeof(double)*8));
  GH_CHECK(ctx, hipMalloc((void**) &D->spl_prev, sizeof(double)*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->spl_win, sizeof(double)*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->spl_kd, sizeof(int)*(W + 2)));
  D->spl_fail = D->spl_kd + W;
  GH_CHECK(ctx, hipMemset(D->spl_prev, 0, sizeof(double)*W));
  GH_CHECK(ctx, hipMemset(D->spl_win, 0, sizeof(double)*W));
  GH_CHECK(ctx, hipMemset(D->spl_kd, 0, sizeof(int)*(W + 2)));
  GH_CHECK(ctx, hipMalloc((void**) &D->wins, sizeof(DDWin)*GH_MAX_RANKS));
  GH_CHECK(ctx, hipMalloc((void**) &D->wnd, sizeof(DDCand)*(DD_WHDR + (size_t) half*(1 + DD_WCAP))));
  GH_CHECK(ctx, hipMalloc((void**) &D->wnd_all, sizeof(DDCand)*(DD_WHDR + (size_t) half*(1 + DD_WCAP))*W));
  GH_CHECK(ctx, hipHostMalloc((void**) &D->h_cnt, sizeof(int)*2*GH_MAX_RANKS));
  GH_CHECK(ctx, hipHostMalloc((void**) &D->h_off, sizeof(long long)*2*GH_MAX_RANKS));
  GH_CHECK(ctx, hipHostMalloc((void**) &D->h_all, sizeof(int)*GH_MAX_RANKS*(2*GH_MAX_RANKS + 2)));
  GH_CHECK(ctx, hipMalloc((void**) &D->box6_all, sizeof(double)*8*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_cnt, sizeof(int)*(4*GH_MAX_RANKS)));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_slot, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_hole, sizeof(int)*n));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_send, sizeof(double)*n*DD_RECMAX));
  GH_CHECK(ctx, hipMalloc((void**) &D->mig_recv, sizeof(double)*n*DD_RECMAX));
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
  GH_CHECK(ctx, hipMalloc((void**) &D->ln_roff, sizeof(long long)*GH_MAX_RANKS));
  // a rank can need, at most, all of another rank's subtree
  D->let_cellcap = (size_t) 2*(ctx->gtot >> ctx->L); D->let_leafcap = (size_t) (ctx->gtot >> ctx->L);
  GH_CHECK(ctx, hipMalloc((void**) &D->let_cells, sizeof(int)*D->let_cellcap*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->let_leaves, sizeof(int)*D->let_leafcap*W));
  GH_CHECK(ctx, hipMalloc((void**) &D->dt_all, sizeof(double)*(W + 2)));
  D->vis_stride = ((size_t) 2*(ctx->gtot >> ctx->L) + 2 + 255) & ~(size_t) 255;     // flags of cell `loc` at byte 1 + loc
  GH_CHECK(ctx, hipMalloc((void**) &D->let_vis, D->vis_stride*(size_t) W));
  GH_CHECK(ctx, hipMalloc((void**) &D->let_work, sizeof(LetWork)*((size_t) W << (D->P + D->F))));
  return GH_OK;
}

// the L shared top levels and the migration (see the header comment); leaves dbbmin/dbbmax of this rank's cell set
static int dd_levels_exact(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  const int W = ctx->nranks, L = ctx->L;
  hipStream_t s = ctx->stream;
  const int pn = (int) (ctx->own_held >= 0 ? ctx->own_held : ctx->own_count);
  const int nb = cdiv(std::max(pn, 1), 256);
  DevicePtrs own = gh_dev_own(ctx);
  own.N = pn;
  // global root box
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
    hipLaunchKernelGGL(k_dd_select, dim3(nc), dim3(1024), 0, s, D->cells, D->cand_all, nc, W, ctx->dbbmin, ctx->dbbmax, ctx->kdiv, ctx->d_flags,
                       D->spl_prev, D->spl_win, D->spl_kd, ctx->d_blk + 13);
    hipLaunchKernelGGL(k_dd_assign, dim3(nb), dim3(256), 0, s, own, D->topcell, D->cells);
  }
  return GH_OK;
}

// the same splits from windows around last step's medians: one collective per level, none for the root box
static int dd_levels_speculative(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  const int W = ctx->nranks, L = ctx->L;
  hipStream_t s = ctx->stream;
  const int pn = (int) (ctx->own_held >= 0 ? ctx->own_held : ctx->own_count);
  const int nb = cdiv(std::max(pn, 1), 256);
  DevicePtrs own = gh_dev_own(ctx);
  own.N = pn;
  GH_CHECK(ctx, hipMemsetAsync(D->topcell, 0, sizeof(int)*(size_t) pn, s));
  GH_CHECK(ctx, hipMemsetAsync(D->spl_fail, 0, sizeof(int), s));
  // test hook: GH_DD_WINSCALE=0 empties every window, so each step's speculative attempt misses its medians and the
  // collective fallback to the exact search runs (test_speculative_splits_fall_back_collectively)
  const double winscale = getenv("GH_DD_WINSCALE") ? atof(getenv("GH_DD_WINSCALE")) : 1.0;
  for (int l = 0; l < L; l++) {
    const int nc = 1 << l;
    const size_t stride = DD_WHDR + (size_t) nc*(1 + DD_WCAP);
    hipLaunchKernelGGL(k_dd_win_init, dim3(1), dim3(64), 0, s, D->cells, D->wins, l, ctx->dbbmin, ctx->dbbmax, ctx->cN, ctx->ndim,
                       D->spl_prev, D->spl_win, D->spl_kd, D->spl_fail, winscale);
    GH_CHECK(ctx, hipMemsetAsync(D->wnd, 0, sizeof(DDCand)*stride, s));
    if (l == 0) hipLaunchKernelGGL(k_dd_win_box, dim3(1), dim3(64), 0, s, ctx->dbbmin, ctx->dbbmax, D->wnd);   // this rank's extent (gh_rootbox_local)
    hipLaunchKernelGGL(k_dd_window, dim3(nb), dim3(256), 0, s, own, D->topcell, D->wins, nc, D->wnd, D->spl_fail);
    DD_OP(ctx, dd_allgather(ctx, D->wnd, D->wnd_all, sizeof(DDCand)*stride));
    hipLaunchKernelGGL(k_dd_wselect, dim3(nc), dim3(1024), 0, s, D->cells, D->wnd_all, stride, l, W, ctx->ndim, ctx->dbbmin, ctx->dbbmax, ctx->kdiv,
                       D->spl_prev, D->spl_win, D->spl_kd, D->spl_fail, ctx->d_blk + 13);
    hipLaunchKernelGGL(k_dd_assign, dim3(nb), dim3(256), 0, s, own, D->topcell, D->cells);
  }
  return GH_OK;
}

__global__ void k_mig_status(int *mig_cnt, const int *fail, int held) { mig_cnt[GH_MAX_RANKS] = fail ? *fail : 0; mig_cnt[GH_MAX_RANKS + 1] = held; }

int gh_dd_decompose(gh_ctx *ctx)
{
  gh_dd *D = ctx->dd;
  if (!D) return gh_fail(ctx, GH_ERR_INVALID, "multi-GPU: gh_comm_init was not called");
  int rc = dd_alloc(ctx);
  if (rc) return rc;
  const int W = ctx->nranks, L = ctx->L;
  hipStream_t s = ctx->stream;
  // the particles this rank holds: its cell's static count, except right after a sink run removed accreted particles
  // (gh_sinks_delete_dead): then every cell's count follows the new N and the migration below evens the ranks out
  const int pn = (int) (ctx->own_held >= 0 ? ctx->own_held : ctx->own_count);
  const int nb = cdiv(std::max(pn, 1), 256);
  DevicePtrs own = gh_dev_own(ctx);
  own.N = pn;

  // this rank's extent of r -/+ kernrange*h; every own particle starts in this rank's cell
  gh_rootbox_local(ctx, (1 << L) - 1 + ctx->rank);
  bool spec = D->have_splits && !getenv("GH_DD_EXACT");
  const int CWM = GH_MAX_RANKS + 2;                         // leavers per destination, status word, particles held
  std::vector<int> all((size_t) W*CWM);
  for (;;) {
    if ((rc = spec ? dd_levels_speculative(ctx) : dd_levels_exact(ctx))) return rc;
    // migration: topcell is now the destination rank.  Leavers per destination -> every rank learns its arrivals per
    // source; the status word of the speculative splits rides along (identical on all ranks: it is computed from
    // gathered data only - the OR over the ranks below is belt and braces)
    GH_CHECK(ctx, hipMemsetAsync(D->mig_cnt, 0, sizeof(int)*4*GH_MAX_RANKS, s));
    hipLaunchKernelGGL(k_mig_count, dim3(nb), dim3(256), 0, s, pn, D->topcell, ctx->rank, D->mig_cnt, D->mig_slot, D->mig_hole);
    hipLaunchKernelGGL(k_mig_status, dim3(1), dim3(1), 0, s, D->mig_cnt, spec ? D->spl_fail : nullptr, pn);
    DD_OP(ctx, dd_allgather(ctx, D->mig_cnt, D->hist_all, sizeof(int)*CWM));
    GH_CHECK(ctx, hipMemcpyAsync(all.data(), D->hist_all, sizeof(int)*all.size(), hipMemcpyDeviceToHost, s));
    GH_CHECK(ctx, hipStreamSynchronize(s));
    int failed = 0;
    for (int r = 0; r < W; r++) failed |= all[(size_t) r*CWM + GH_MAX_RANKS];
    if (spec && failed && getenv("GH_DD_DEBUG")) fprintf(stderr, "[dd] rank %d: speculative splits failed, status %d\n", ctx->rank, failed);
    if (spec && failed) { spec = false; continue; }       // collective decision: every rank sees the same words
    break;
  }
  if (spec) D->n_spec++; else D->n_exact++;
  D->have_splits = true;
  if (getenv("GH_DD_DEBUG")) {
    fprintf(stderr, "[dd] rank %d held %d cell count %d spec %d:", ctx->rank, pn, ctx->h_cN[(1 << L) - 1 + ctx->rank], (int) spec);
    for (int r = 0; r < W; r++) fprintf(stderr, " ->%d: %d <-%d: %d", r, all[(size_t) ctx->rank*CWM + r], r, all[(size_t) r*CWM + ctx->rank]);
    fprintf(stderr, "\n");
  }
  int64_t sb[GH_MAX_RANKS], rb[GH_MAX_RANKS];
  MigTab tab;
  // block timesteps: level, levelneib, nstep, nlast, flags travel too; sink runs: sinkid and the flags (dead, potmin)
  tab.nf = (ctx->cfg.Nlevels > 1 || ctx->cfg.sink_particles) ? D_COUNT : D_COUNT_BASE;
  const int nrec = tab.nf + 1;
  long long nsend = 0, nrecv = 0;
  for (int r = 0; r < W; r++) {
    const int out = all[(size_t) ctx->rank*CWM + r], in = all[(size_t) r*CWM + ctx->rank];
    tab.off[r] = (int) nsend;
    sb[r] = (int64_t) out*nrec*sizeof(double); rb[r] = (int64_t) in*nrec*sizeof(double);
    nsend += out; nrecv += in;
  }
  // every rank must end up with its cell's count: held - leavers + arrivals = cN on EVERY rank or on none - a split that
  // left a cell with the wrong count (equal coordinates beyond DD_CAPL) shows up on all ranks, and all of them stop here together
  bool unbalanced = false;
  for (int r = 0; r < W; r++) {
    long long o = 0, in = 0;
    for (int q = 0; q < W; q++) { o += all[(size_t) r*CWM + q]; in += all[(size_t) q*CWM + r]; }
    if ((long long) all[(size_t) r*CWM + GH_MAX_RANKS + 1] - o + in != (long long) ctx->h_cN[(1 << L) - 1 + r]) unbalanced = true;
  }
  if (unbalanced) return gh_fail(ctx, GH_ERR_INVALID, "multi-GPU: unbalanced migration (equal coordinates at a top-level split?)");
  for (int f = 0; f < tab.nf; f++) tab.fld[f] = own.f[f];
  if (nsend > 0) hipLaunchKernelGGL(k_mig_pack, dim3(nb), dim3(256), 0, s, tab, own.iorig, pn, D->topcell, ctx->rank, D->mig_slot, D->mig_send);
  DD_OP(ctx, D->ops.alltoallv(D->ops.user, D->mig_send, sb, D->mig_recv, rb, (void*) s));     // collective: every rank calls it
  if (nrecv > 0) hipLaunchKernelGGL(k_mig_unpack, dim3(cdiv(nrecv, 256)), dim3(256), 0, s, tab, own.iorig, (int) nrecv, D->mig_hole, D->mig_recv, (int) nsend, pn);
  if (nsend > nrecv) {
    // holes nrecv .. nsend stayed open: the new end is pn - (nsend - nrecv); what lies behind it moves into the open holes before it
    const int nopen = (int) (nsend - nrecv), target = pn - nopen;
    std::vector<int> open((size_t) nopen), from, to;
    GH_CHECK(ctx, hipMemcpyAsync(open.data(), D->mig_hole + nrecv, sizeof(int)*(size_t) nopen, hipMemcpyDeviceToHost, s));
    GH_CHECK(ctx, hipStreamSynchronize(s));
    std::sort(open.begin(), open.end());
    for (int q : open) if (q < target) to.push_back(q);
    for (int q = target; q < pn; q++) if (!std::binary_search(open.begin(), open.end(), q)) from.push_back(q);
    if (from.size() != to.size()) return gh_fail(ctx, GH_ERR_INVALID, "multi-GPU: migration bookkeeping out of step");
    if (!from.empty()) {
      // (topcell / mig_slot have done their work: scratch for the two index lists)
      GH_CHECK(ctx, hipMemcpyAsync(D->topcell, from.data(), sizeof(int)*from.size(), hipMemcpyHostToDevice, s));
      GH_CHECK(ctx, hipMemcpyAsync(D->mig_slot, to.data(), sizeof(int)*to.size(), hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(k_mig_move, dim3(cdiv(from.size(), 256)), dim3(256), 0, s, tab, own.iorig, D->topcell, D->mig_slot, (int) from.size());
      GH_CHECK(ctx, hipStreamSynchronize(s));              // from / to live on this stack
    }
  }
  ctx->own_held = -1;
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
  hipLaunchKernelGGL(k_pub_pack, dim3(cdiv(ncell, 64)), dim3(64), 0, ctx->stream, d, L, P, ctx->rank, (PubRec*) D->comb_send, ctx->d_blk + 13);
  hipLaunchKernelGGL(k_pub_fine, dim3(1 << PF), dim3(256), 0, ctx->stream, d, L, PF, ctx->rank, kr, 1.0, (LetGeomF*) (D->comb_send + D->pub_bytes));
  DD_OP(ctx, dd_allgather(ctx, D->comb_send, D->comb_recv, blk));
  D->fine_widen = 1.0; D->fine_base = D->comb_recv + D->pub_bytes; D->fine_stride = blk;
  hipLaunchKernelGGL(k_pub_unpack, dim3(cdiv(ncell*W, 256)), dim3(256), 0, ctx->stream, d, L, P, ctx->rank, W, D->comb_recv, blk, ctx->d_blk + 13);
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
  if (phase == GH_HALO_DENSITY && widen != D->fine_widen) { const int rc = dd_publish_fine(ctx, widen); if (rc) return rc; }
  hipLaunchKernelGGL(k_let_invalidate, dim3(cdiv(ctx->Ncell, 256)), dim3(256), 0, s, d, L, P, ctx->rank, ctx->Ncell);
  GH_CHECK(ctx, hipMemsetAsync(D->let_cnt, 0, sizeof(int)*8*GH_MAX_RANKS, s));
  GH_CHECK(ctx, hipMemsetAsync(D->let_vis, 0, D->vis_stride*(size_t) W, s));
  LetPeriod per;
  bool anyper = false;
  for (int k = 0; k < 3; k++) {
    per.len[k] = (k < ctx->ndim && ctx->cfg.boundary_lhs[k] == GH_BOUNDARY_PERIODIC) ? ctx->cfg.boxmax[k] - ctx->cfg.boxmin[k] : 0.0;
    anyper = anyper || per.len[k] > 0.0;
  }
  {
    const int PF = P + D->F;
    int *nwork = D->let_cnt + 6*GH_MAX_RANKS, *head = nwork + 1;       // zeroed with let_cnt above
    const int nthr = W << PF;
    const int nwaves = 256*16;                             // persistent marking waves: what the device holds of them
    // (nodes per lane and round 1 / 2, stack 1024 / 2048 entries, 16 / 28 waves per CU all measured the same 220 us)
#define LW_LAUNCH(PH, PR) { \
    hipLaunchKernelGGL((k_let_prepass<PH, PR>), dim3(cdiv(nthr, 256)), dim3(256), 0, s, d, L, P, PF, ctx->rank, W, kr, widen, D->fine_base, D->fine_stride, per, D->let_work, nwork); \
    hipLaunchKernelGGL((k_let_walk<PH, PR, 1, 2048>), dim3(nwaves), dim3(64), 0, s, d, L, P, ctx->rank, kr, widen, D->fine_base, D->fine_stride, D->let_vis, D->vis_stride, per, D->let_work, nwork, head); }
    if (anyper) { if (phase == GH_HALO_DENSITY) LW_LAUNCH(GH_HALO_DENSITY, true) else if (phase == GH_HALO_HYDRO) LW_LAUNCH(GH_HALO_HYDRO, true) else LW_LAUNCH(GH_HALO_GRAVITY, true) }
    else { if (phase == GH_HALO_DENSITY) LW_LAUNCH(GH_HALO_DENSITY, false) else if (phase == GH_HALO_HYDRO) LW_LAUNCH(GH_HALO_HYDRO, false) else LW_LAUNCH(GH_HALO_GRAVITY, false) }
#undef LW_LAUNCH
  }
  {
    const int nloc = 2*(ctx->gtot >> L) - 1;              // cells of this rank's subtree
    hipLaunchKernelGGL(k_let_compact, dim3(std::min(cdiv(nloc, 256), 512), W), dim3(256), 0, s, L, ctx->rank, D->let_vis, D->vis_stride, nloc,
                       D->let_cnt, D->let_cells, D->let_leaves, D->let_cellcap, D->let_leafcap, ctx->d_flags);
  }
  // counts to everybody (2 ints per pair), then sizes on the host.  The miss counter of a density pass whose check was
  // deferred (gh_step: the force-phase exchange is the next collective anyway) rides in the same block.
  const int CW = 2*GH_MAX_RANKS + 2;
  const bool check_miss = phase != GH_HALO_DENSITY && ctx->dd_miss_pending;
  if (check_miss) GH_CHECK(ctx, hipMemcpyAsync(D->let_cnt + 2*GH_MAX_RANKS, ctx->d_blk + 12, sizeof(int), hipMemcpyDeviceToDevice, s));
  DD_OP(ctx, dd_allgather(ctx, D->let_cnt, D->hist_all, sizeof(int)*CW));
  int *all = D->h_all;
  GH_CHECK(ctx, hipMemcpyAsync(all, D->hist_all, sizeof(int)*(size_t) W*CW, hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  if (check_miss) {
    ctx->dd_miss_pending = false;
    int any = 0;
    for (int r = 0; r < W; r++) any |= all[(size_t) r*CW + 2*GH_MAX_RANKS];
    if (any) return GH_DD_REDO_DENSITY;                    // collective decision; the caller widens the density import and redoes those groups
  }
  LetLayout lay;
  lay.phase = phase; lay.quad = ctx->cquad ? 1 : 0; lay.occ = ctx->leafocc;
  lay.cell_dbl = phase == GH_HALO_DENSITY ? 9 : (lay.quad ? 34 : 29);
  lay.lev = (phase != GH_HALO_DENSITY && ctx->cfg.Nlevels > 1) ? 1 : 0;
  lay.gpot = (phase == GH_HALO_DENSITY && ctx->cfg.sink_particles && ctx->cfg.create_sinks == 1) ? 1 : 0;
  lay.part_dbl = phase == GH_HALO_DENSITY ? 4 + lay.gpot : 20 + lay.lev;
  lay.leaf_dbl = 1 + lay.occ*lay.part_dbl;
  int64_t sb[GH_MAX_RANKS], rb[GH_MAX_RANKS];
  long long soff[GH_MAX_RANKS], roff[GH_MAX_RANKS], stot = 0, rtot = 0, nimp = 0;
  int *rcnt = D->h_cnt;
  for (int r = 0; r < W; r++) {
    const int oc = r == ctx->rank ? 0 : all[(size_t) ctx->rank*CW + 2*r], ol = r == ctx->rank ? 0 : all[(size_t) ctx->rank*CW + 2*r + 1];
    const int ic = r == ctx->rank ? 0 : all[(size_t) r*CW + 2*ctx->rank], il = r == ctx->rank ? 0 : all[(size_t) r*CW + 2*ctx->rank + 1];
    soff[r] = stot; roff[r] = rtot;
    const long long sd = (long long) oc*lay.cell_dbl + (long long) ol*lay.leaf_dbl, rd = (long long) ic*lay.cell_dbl + (long long) il*lay.leaf_dbl;
    sb[r] = sd*8; rb[r] = rd*8; stot += sd; rtot += rd;
    rcnt[2*r] = ic; rcnt[2*r + 1] = il;
    nimp += (long long) il*lay.occ;
  }
  if (phase != GH_HALO_DENSITY) {
    D->held_particles = ctx->own_count + nimp;
    for (int r = 0; r < W; r++) {
      D->fwd_ol[r] = r == ctx->rank ? 0 : all[(size_t) ctx->rank*CW + 2*r + 1];
      D->fwd_ic[r] = rcnt[2*r]; D->fwd_il[r] = rcnt[2*r + 1]; D->fwd_roff[r] = roff[r];
    }
    D->fwd_lay_cell = lay.cell_dbl; D->fwd_lay_leaf = lay.leaf_dbl; D->fwd_occ = lay.occ;
  }
  if (getenv("GH_DD_DEBUG")) {
    fprintf(stderr, "[dd] rank %d phase %d widen %.1f:", ctx->rank, phase, widen);
    for (int r = 0; r < W; r++) if (r != ctx->rank) fprintf(stderr, "  to %d: %d cells %d leaves | from %d: %d cells %d leaves", r,
        all[(size_t) ctx->rank*CW + 2*r], all[(size_t) ctx->rank*CW + 2*r + 1], r, rcnt[2*r], rcnt[2*r + 1]);
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
  // received counts and block offsets (in doubles) for the pack / unpack kernels: staged in pinned host memory that
  // lives as long as the decomposition (the next exchange synchronises before it rewrites them)
  GH_CHECK(ctx, hipMemcpyAsync(D->let_cnt + 4*GH_MAX_RANKS, rcnt, sizeof(int)*2*GH_MAX_RANKS, hipMemcpyHostToDevice, s));
  long long *hoff = D->h_off;
  for (int r = 0; r < GH_MAX_RANKS; r++) { hoff[r] = r < W ? soff[r] : 0; hoff[GH_MAX_RANKS + r] = r < W ? roff[r] : 0; }
  long long *d_offs = D->let_off;
  GH_CHECK(ctx, hipMemcpyAsync(d_offs, hoff, sizeof(long long)*2*GH_MAX_RANKS, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_let_pack, dim3(1024, W), dim3(256), 0, s, d, lay, W, ctx->rank, D->let_cnt, D->let_cells, D->let_leaves,
                     D->let_cellcap, D->let_leafcap, d_offs, (double*) D->let_send);
  DD_OP(ctx, D->ops.alltoallv(D->ops.user, D->let_send, sb, D->let_recv, rb, (void*) s));
  hipLaunchKernelGGL(k_let_unpack, dim3(1024, W), dim3(256), 0, s, d, lay, W, ctx->rank, D->let_cnt + 4*GH_MAX_RANKS, d_offs + GH_MAX_RANKS, (const double*) D->let_recv);
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

// ------------------------------------------------------------------------------------------------
// block timesteps on more than one rank: the reductions of Simulation::ComputeBlockTimesteps (the MPI_Allreduce calls of
// Simulation.cpp:1843-1847, 2016-2080: minimum timestep, highest occupied level) and of MainLoop's wake-up count
// (SphSimulation.cpp:753), on one device word each
// ------------------------------------------------------------------------------------------------
__global__ void k_dd_reduce_int(const int *all, int nranks, int *word, int op)
{
  int v = all[0];
  for (int r = 1; r < nranks; r++) v = op == 0 ? max(v, all[r]) : v + all[r];
  *word = v;
}
int gh_dd_reduce_int(gh_ctx *ctx, int *word_dev, int op /* 0 max, 1 sum */)
{
  if (ctx->nranks == 1) return GH_OK;
  gh_dd *D = ctx->dd;
  DD_OP(ctx, dd_allgather(ctx, word_dev, D->hist_all, sizeof(int)));
  hipLaunchKernelGGL(k_dd_reduce_int, dim3(1), dim3(1), 0, ctx->stream, D->hist_all, ctx->nranks, word_dev, op);
  return GH_OK;
}

// levelneib of the imported particles -> their owners.  An active particle raises levelneib of every SPH neighbour to
// its own level (GradhSph.cpp:455, 569; GradhSphTree.cpp:376-417); for a neighbour this rank only holds a copy of, the
// raise lands in the copy (zeroed at import).  The copies' values go back along the lists of the force-phase exchange -
// no count exchange, the sizes are those of the way out - and the owners take the maximum.  (The reference's MPI layer
// returns levelneib with the exported particles' accelerations, MpiControl.cpp:910-990.)
__global__ void k_ln_pack(DevicePtrs d, int self, const double *recv, const long long *roff, const int *rcnt, int cell_dbl, int leaf_dbl, int occ,
                          const long long *boff, double *out)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int ic = rcnt[2*r], il = rcnt[2*r + 1];
  const double *leaf0 = recv + roff[r] + (long long) ic*cell_dbl;
  for (int e = blockIdx.x*blockDim.x + threadIdx.x; e < il*occ; e += gridDim.x*blockDim.x) {
    const int l = e/occ, t = e - l*occ;
    const int n = (int) leaf0[(long long) l*leaf_dbl];
    out[boff[r] + e] = t < d.cN[n] ? d.f[D_LEVELNEIB][d.cfirst[n] + t] : 0.0;
  }
}
__global__ void k_ln_merge(DevicePtrs d, int self, const int *cnt, const int *leaves, size_t leafcap, int occ, const long long *boff, const double *in)
{
  const int r = blockIdx.y;
  if (r == self) return;
  const int ol = cnt[2*r + 1];
  for (int e = blockIdx.x*blockDim.x + threadIdx.x; e < ol*occ; e += gridDim.x*blockDim.x) {
    const int l = e/occ, t = e - l*occ;
    const int n = leaves[(size_t) r*leafcap + l];
    if (t < d.cN[n]) {
      const double v = in[boff[r] + e];
      double *p = &d.f[D_LEVELNEIB][d.cfirst[n] + t];
      if (*p < v) atomicMax((unsigned long long*) p, (unsigned long long) __double_as_longlong(v));     // non-negative doubles order like their bit patterns
    }
  }
}
int gh_dd_return_levelneib(gh_ctx *ctx)
{
  if (ctx->nranks == 1 || ctx->cfg.Nlevels <= 1) return GH_OK;
  gh_dd *D = ctx->dd;
  const int W = ctx->nranks, occ = D->fwd_occ;
  hipStream_t s = ctx->stream;
  int64_t sb[GH_MAX_RANKS], rb[GH_MAX_RANKS];
  // (h_off is pinned staging shared with the forward exchange, whose copy out of it may still be queued: drain first)
  GH_CHECK(ctx, hipStreamSynchronize(s));
  long long *hoff = D->h_off;                             // [0, MAX): send offsets, [MAX, 2 MAX): receive offsets
  long long stot = 0, rtot = 0;
  for (int r = 0; r < GH_MAX_RANKS; r++) {
    const long long sd = r < W ? (long long) D->fwd_il[r]*occ : 0, rd = r < W ? (long long) D->fwd_ol[r]*occ : 0;
    hoff[r] = stot; hoff[GH_MAX_RANKS + r] = rtot;
    if (r < W) { sb[r] = sd*8; rb[r] = rd*8; }
    stot += sd; rtot += rd;
  }
  auto grow = [&](char **p, size_t *have, size_t need) -> hipError_t {
    if (need <= *have) return hipSuccess;
    if (*p) (void) hipFree(*p);
    *p = nullptr; *have = 0;
    const size_t cap = need + need/4 + 4096;
    const hipError_t e = hipMalloc((void**) p, cap);
    if (e == hipSuccess) *have = cap;
    return e;
  };
  GH_CHECK(ctx, grow(&D->ln_send, &D->ln_send_bytes, (size_t) stot*8 + 8));
  GH_CHECK(ctx, grow(&D->ln_recv, &D->ln_recv_bytes, (size_t) rtot*8 + 8));
  long long *d_offs = D->let_off;                         // device [2 MAX]
  GH_CHECK(ctx, hipMemcpyAsync(d_offs, hoff, sizeof(long long)*2*GH_MAX_RANKS, hipMemcpyHostToDevice, s));
  GH_CHECK(ctx, hipMemcpyAsync(D->ln_roff, D->fwd_roff, sizeof(long long)*GH_MAX_RANKS, hipMemcpyHostToDevice, s));
  DevicePtrs d = gh_dev(ctx);
  hipLaunchKernelGGL(k_ln_pack, dim3(256, W), dim3(256), 0, s, d, ctx->rank, (const double*) D->let_recv, D->ln_roff, D->let_cnt + 4*GH_MAX_RANKS,
                     D->fwd_lay_cell, D->fwd_lay_leaf, occ, d_offs, (double*) D->ln_send);
  DD_OP(ctx, D->ops.alltoallv(D->ops.user, D->ln_send, sb, D->ln_recv, rb, (void*) s));
  hipLaunchKernelGGL(k_ln_merge, dim3(256, W), dim3(256), 0, s, d, ctx->rank, D->let_cnt, D->let_leaves, D->let_leafcap, occ,
                     d_offs + GH_MAX_RANKS, (const double*) D->ln_recv);
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

// ragged all-gather of small records to the hosts of all ranks (sink runs: dead slots, sink candidates, the rows of the
// gas inside the sink radii - what the reference moves with MPI_Allgatherv / MPI_Bcast in Sinks.cpp and MpiControl.cpp).
// `src` may be device or host memory; out = the ranks' blocks in rank order, sizes[r] = bytes of rank r.  Two all-gathers:
// the sizes, then blocks padded to the largest.  One rank: a plain copy.
int gh_dd_gatherv(gh_ctx *ctx, const void *src, size_t bytes, std::vector<char> &out, std::vector<size_t> &sizes)
{
  const int W = ctx->nranks;
  hipStream_t s = ctx->stream;
  sizes.assign((size_t) W, 0);
  if (W == 1) {
    out.resize(bytes); sizes[0] = bytes;
    if (bytes) GH_CHECK(ctx, hipMemcpyAsync(out.data(), src, bytes, hipMemcpyDefault, s));
    GH_CHECK(ctx, hipStreamSynchronize(s));
    return GH_OK;
  }
  gh_dd *D = ctx->dd;
  if (!D->gv_cnt) GH_CHECK(ctx, hipMalloc((void**) &D->gv_cnt, sizeof(long long)*(GH_MAX_RANKS + 1)));
  long long mine = (long long) bytes;
  std::vector<long long> cnt((size_t) W);
  GH_CHECK(ctx, hipMemcpyAsync(D->gv_cnt, &mine, sizeof(long long), hipMemcpyHostToDevice, s));
  DD_OP(ctx, dd_allgather(ctx, D->gv_cnt, D->gv_cnt + 1, sizeof(long long)));
  GH_CHECK(ctx, hipMemcpyAsync(cnt.data(), D->gv_cnt + 1, sizeof(long long)*(size_t) W, hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  size_t mx = 0, tot = 0;
  for (int r = 0; r < W; r++) { sizes[r] = (size_t) cnt[r]; mx = std::max(mx, sizes[r]); tot += sizes[r]; }
  out.resize(tot);
  if (mx == 0) return GH_OK;
  mx = (mx + 15) & ~(size_t) 15;
  auto grow = [&](char **p, size_t *have, size_t need) -> hipError_t {
    if (need <= *have) return hipSuccess;
    if (*p) (void) hipFree(*p);
    *p = nullptr; *have = 0;
    const hipError_t e = hipMalloc((void**) p, need + need/2 + 4096);
    if (e == hipSuccess) *have = need + need/2 + 4096;
    return e;
  };
  GH_CHECK(ctx, grow(&D->gv_send, &D->gv_send_bytes, mx));
  GH_CHECK(ctx, grow(&D->gv_recv, &D->gv_recv_bytes, mx*(size_t) W));
  if (bytes) GH_CHECK(ctx, hipMemcpyAsync(D->gv_send, src, bytes, hipMemcpyDefault, s));
  DD_OP(ctx, dd_allgather(ctx, D->gv_send, D->gv_recv, mx));
  std::vector<char> pad(mx*(size_t) W);
  GH_CHECK(ctx, hipMemcpyAsync(pad.data(), D->gv_recv, pad.size(), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  size_t o = 0;
  for (int r = 0; r < W; r++) { if (sizes[r]) memcpy(out.data() + o, pad.data() + (size_t) r*mx, sizes[r]); o += sizes[r]; }
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
