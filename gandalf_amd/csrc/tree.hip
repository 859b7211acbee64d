// tree.hip -- GPU construction of GANDALF's balanced KD-tree.
//
// Produces exactly the tree of the reference's KDTree::BuildTree / DivideTreeCell / StockCellProperties
// (reference src/Tree/KDTree.cpp:220-313, 442-595, 808-1083): the cell at level l that holds N particles
// is split along the longest axis of its *inherited* box (root box = extent of r -/+ kernrange*h,
// KDTree.cpp:277-278; children inherit it cut at the split value, :508-527) so that the N/2 particles
// with the smallest coordinate go left and the rest go right (QuickSelect, :682-750).  Only the SET of
// particles per cell is defined by that (ties between equal coordinates aside); the reference finds it
// with a serial recursive quick-select on an index array, here it is found level by level for all
// cells at once:
//
//   1. argsort the particles once per axis (radix sort of the fp64 coordinates);
//   2. per level: every cell picks its split axis and reads the median straight out of the list sorted
//      on that axis; every particle is marked left/right; the per-axis lists are stably partitioned
//      inside each cell segment (ballot words + popcount prefix), which keeps them sorted;
//   3. the final list is the tree order; all particle arrays are gathered into it (one pass), so that
//      every cell at every level owns a CONTIGUOUS range of particles from then on;
//   4. cells are stocked bottom-up, one launch per level.
//
// Cells live in heap order (children of n: 2n+1, 2n+2) - levels are contiguous, which is what the
// level-synchronous kernels want; gh_export_tree renumbers to the reference's pre-order ids.
#include "gh_internal.hpp"
#include <rocprim/rocprim.hpp>

#define KERNRANGE_OF(cfg) (((cfg).kernel == GH_KERNEL_QUINTIC || (cfg).kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0)

static const double BIG = 9.9e20;   // reference Constants.h:72 big_number

// ------------------------------------------------------------------------------------------------
// root box: min/max over particles of r -/+ kernrange*h           (KDTree.cpp:269-280)
// ------------------------------------------------------------------------------------------------
#define GH_RB_THREADS 1024       /* 256 blocks of 16 waves (4 waves per block left one wave per SIMD: latency-bound) */
__global__ __launch_bounds__(GH_RB_THREADS) void k_rootbox_partial(DevicePtrs d, double kernrange, double *out /* [nblk][6] */, int *cellnode0, int node0)
{
  // d is the view of this rank's own particles (all particles on one rank); cellnode0 is shifted likewise
  __shared__ double s[6][GH_RB_THREADS];
  double mn[3] = {BIG, BIG, BIG}, mx[3] = {-BIG, -BIG, -BIG};
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < d.N; i += gridDim.x*blockDim.x) {
    cellnode0[i] = node0;              // every particle starts in the root cell (of this rank's subtree)
    const double hr = kernrange*d.f[D_H][i];
    _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
      const double x = d.f[D_RX + k][i];
      mx[k] = fmax(mx[k], x + hr);
      mn[k] = fmin(mn[k], x - hr);
    }
  }
  for (int k = 0; k < 3; k++) { s[k][threadIdx.x] = mn[k]; s[3 + k][threadIdx.x] = mx[k]; }
  __syncthreads();
  for (int off = GH_RB_THREADS/2; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off)
      for (int k = 0; k < 3; k++) {
        s[k][threadIdx.x] = fmin(s[k][threadIdx.x], s[k][threadIdx.x + off]);
        s[3 + k][threadIdx.x] = fmax(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + off]);
      }
    __syncthreads();
  }
  if (threadIdx.x < 6) out[blockIdx.x*6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void k_rootbox_final(const double *part, int nblk, double *dbbmin, double *dbbmax)
{
  // one wave: lanes stride over the block partials, then a shuffle reduction per component
  const int lane = threadIdx.x;
  for (int k = 0; k < 3; k++) {
    double mn = BIG, mx = -BIG;
    for (int b = lane; b < nblk; b += 64) { mn = fmin(mn, part[b*6 + k]); mx = fmax(mx, part[b*6 + 3 + k]); }
    for (int off = 32; off > 0; off >>= 1) { mn = fmin(mn, __shfl_xor(mn, off, 64)); mx = fmax(mx, __shfl_xor(mx, off, 64)); }
    if (lane == 0) { dbbmin[k] = mn; dbbmax[k] = mx; }
  }
}

__global__ void k_iota(int *v, int n, int base)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) v[i] = base + i;
}

__global__ void k_fill_int(int *v, int n, int val)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) v[i] = val;
}

// ------------------------------------------------------------------------------------------------
// level step 1: per cell, split axis + median + children's inherited boxes   (KDTree.cpp:490-533)
// ------------------------------------------------------------------------------------------------
struct LevelArgs {
  const int *P[3];
  int *Pn[3];
  const int *cellnode; int *cellnode_next;
  unsigned char *side;
  unsigned long long *W[3];
  unsigned int *Wpre[3];
  double *dbbmin, *dbbmax;
  int *kdiv;
  int *tie;              // set when a median split separates equal coordinates (the reference's quick-select order then matters)
  int level, nwords;
  int p0, pn;            // particle range [p0, p0 + pn) the build works on: everything on one rank, the rank's own cell otherwise
  int jbase;             // k_build_subtree: first level-L0 cell of the range
  const int *cleft;      // per cell: lefts (floor(N/2) each) of the cells before it on its level
  int cleft0;            // ... of the first cell of this rank's range on the level being split
};

// One level of the build above the LDS-resident subtrees, two kernels.
//
// k_level_flags: per cell, split axis + median + children's inherited boxes (KDTree.cpp:490-533); per position of each
// of the three presorted lists, whether its particle goes left.  A cell's segment of every list is sorted by
// (coordinate, position in the argsort's input) - the argsort is stable and the partitions below keep the order - so
// "left" is "(x_kd, id) < (x_kd, id) of the median element", read straight from the coordinates: no scatter of side
// marks through particle ids.  Output: one 64-bit word of left-flags per wavefront and list, and per 1024-position block
// and list the number of lefts.
// k_level_scatter: stable partition of every cell segment of every list.  The rank of a left-going element among the
// lefts of its cell = (lefts before it in the whole list) - (lefts before the cell's first position); the second term
// is static - every cell sends exactly floor(N/2) left - and precomputed per cell (`cleft`); the first is the block's
// prefix (each block sums the counts of the blocks before it: at most 1024 values) + words + bits.
#define GH_LB 1024
__global__ __launch_bounds__(GH_LB) void k_level_flags(DevicePtrs d, LevelArgs a)
{
  __shared__ unsigned int s_cnt[3][GH_LB/64];
  const int pl = blockIdx.x*GH_LB + threadIdx.x;
  const bool in = pl < a.pn;
  const int p = a.p0 + (in ? pl : 0);
  const int n = a.cellnode[p];
  const int first = d.cfirst[n], cnt = d.cN[n], half = cnt/2;
  double bmin[3], bmax[3], rkmax = 0.0;
  int kd = 0;
  for (int k = 0; k < 3; k++) { bmin[k] = a.dbbmin[n*3 + k]; bmax[k] = a.dbbmax[n*3 + k]; }
  _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
    const double ext = bmax[k] - bmin[k];
    if (ext > rkmax) { rkmax = ext; kd = k; }
  }
  const double *xk = d.f[D_RX + kd];
  const int idm = a.P[kd][first + half];
  const double xm = xk[idm];
  _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
    const int id = a.P[k][p];
    const double x = xk[id];
    const bool left = in && (x < xm || (x == xm && id < idm));
    const unsigned long long w = __ballot(left);
    if ((threadIdx.x & 63) == 0) {
      if ((pl >> 6) < a.nwords) a.W[k][pl >> 6] = w;
      s_cnt[k][threadIdx.x >> 6] = __popcll(w);
    }
  }
  if (in) {
    a.cellnode_next[p] = 2*n + 1 + ((p - first >= half) ? 1 : 0);
    if (p == first) {
      // the cell's first position also writes the children's inherited boxes (median = first of the right half)
      if (half > 0 && xk[a.P[kd][first + half - 1]] == xm) *a.tie = 1;
      const int c1 = 2*n + 1, c2 = 2*n + 2;
      for (int k = 0; k < 3; k++) {
        a.dbbmin[c1*3 + k] = bmin[k]; a.dbbmax[c1*3 + k] = (k == kd) ? xm : bmax[k];
        a.dbbmin[c2*3 + k] = (k == kd) ? xm : bmin[k]; a.dbbmax[c2*3 + k] = bmax[k];
      }
      a.kdiv[n] = kd;
    }
  }
  __syncthreads();
  if ((int) threadIdx.x < d.ndim) {
    unsigned int c = 0;
    for (int w = 0; w < GH_LB/64; w++) c += s_cnt[threadIdx.x][w];
    a.Wpre[threadIdx.x][blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(GH_LB) void k_level_scatter(DevicePtrs d, LevelArgs a)
{
  __shared__ unsigned int s_base[3];
  __shared__ unsigned int s_wpre[3][GH_LB/64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < d.ndim) {
    // lefts in all blocks before this one (wave k sums list k) and the running count of this block's words
    unsigned int sum = 0;
    for (int b = lane; b < (int) blockIdx.x; b += 64) sum += a.Wpre[wave][b];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    const int w0 = blockIdx.x*(GH_LB/64);
    unsigned int c = (lane < GH_LB/64 && w0 + lane < a.nwords) ? (unsigned int) __popcll(a.W[wave][w0 + lane]) : 0u;
    unsigned int inc = c;
    for (int off = 1; off < GH_LB/64; off <<= 1) { const unsigned int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane < GH_LB/64) s_wpre[wave][lane] = inc - c;
    if (lane == 0) s_base[wave] = sum;
  }
  __syncthreads();
  const int pl = blockIdx.x*GH_LB + threadIdx.x;
  if (pl >= a.pn) return;
  const int p = a.p0 + pl;
  const int n = a.cellnode[p];
  const int first = d.cfirst[n], half = d.cN[n]/2;
  const unsigned int lbase = (unsigned int) (a.cleft[n] - a.cleft0);        // lefts before the cell's first position
  _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
    const int id = a.P[k][p];
    const unsigned long long w = a.W[k][pl >> 6];
    const bool left = (w >> lane) & 1ull;
    const unsigned int rl = s_base[k] + s_wpre[k][wave] + (unsigned int) __popcll(lane ? (w & ((1ull << lane) - 1ull)) : 0ull);
    const int nleft_before = (int) (rl - lbase);
    const int np = left ? first + nleft_before : first + half + ((p - first) - nleft_before);
    a.Pn[k][np] = id;
  }
}


// ------------------------------------------------------------------------------------------------
// Levels >= L0 (cells of <= GH_SEG particles): the whole subtree below one level-L0 cell is built by ONE
// workgroup with the three presorted lists held in LDS as 16-bit local indices - the same
// divide / mark / ballot / prefix / stable-partition sequence as the level kernels above, with barriers in
// place of launches.  Cuts ~5 launches per level off the (launch-bound) lower half of the build.
// ------------------------------------------------------------------------------------------------
#define GH_SEG 2048
__global__ __launch_bounds__(1024) void k_build_subtree(DevicePtrs d, LevelArgs a, int L0, int ltot, int *perm_out, int *inv)
{
  __shared__ unsigned short s_L[2][3][GH_SEG];
  __shared__ int s_gid[GH_SEG];
  __shared__ unsigned short s_cnode[2][GH_SEG];
  __shared__ unsigned char s_side[GH_SEG];
  __shared__ unsigned long long s_W[3][GH_SEG/64];
  __shared__ unsigned int s_pre[3][GH_SEG/64 + 1];
  const int tid = threadIdx.x;
  const int j0 = a.jbase + blockIdx.x;
  const int n0 = (1 << L0) - 1 + j0;
  const int first0 = d.cfirst[n0], cnt = d.cN[n0];
  const int ndim = d.ndim;
  for (int i = tid; i < cnt; i += 1024) {
    const int g = a.P[0][first0 + i];
    s_gid[i] = g;
    inv[g] = i;
    s_L[0][0][i] = (unsigned short) i;
    s_cnode[0][i] = 0;
  }
  __syncthreads();
  for (int k = 1; k < ndim; k++)
    for (int i = tid; i < cnt; i += 1024) s_L[0][k][i] = (unsigned short) inv[a.P[k][first0 + i]];
  __syncthreads();
  int cur = 0;
  const int nw = (cnt + 63) >> 6;
  for (int l = L0; l < ltot; l++) {
    const int sh = l - L0;
    const int nbase = (1 << l) - 1 + (j0 << sh);
    // divide the cells of this level (KDTree.cpp:490-533)
    for (int t = tid; t < (1 << sh); t += 1024) {
      const int n = nbase + t;
      const int first = d.cfirst[n], c = d.cN[n];
      double rkmax = 0.0;
      int kd = 0;
      for (int k = 0; k < ndim; k++) {
        const double ext = a.dbbmax[n*3 + k] - a.dbbmin[n*3 + k];
        if (ext > rkmax) { rkmax = ext; kd = k; }
      }
      double rdiv = a.dbbmin[n*3 + kd];
      if (c > 0) rdiv = d.f[D_RX + kd][s_gid[s_L[cur][kd][first - first0 + c/2]]];
      if (c > 1 && d.f[D_RX + kd][s_gid[s_L[cur][kd][first - first0 + c/2 - 1]]] == rdiv) *a.tie = 1;
      const int c1 = 2*n + 1, c2 = 2*n + 2;
      for (int k = 0; k < 3; k++) {
        a.dbbmin[c1*3 + k] = a.dbbmin[n*3 + k]; a.dbbmax[c1*3 + k] = a.dbbmax[n*3 + k];
        a.dbbmin[c2*3 + k] = a.dbbmin[n*3 + k]; a.dbbmax[c2*3 + k] = a.dbbmax[n*3 + k];
      }
      a.dbbmax[c1*3 + kd] = rdiv;
      a.dbbmin[c2*3 + kd] = rdiv;
      a.kdiv[n] = kd;
    }
    __syncthreads();
    // mark left / right from the split-axis list
    for (int p = tid; p < cnt; p += 1024) {
      const int jl = s_cnode[cur][p];
      const int n = nbase + jl;
      const int lf = d.cfirst[n] - first0, half = d.cN[n]/2;
      const int right = (p - lf >= half) ? 1 : 0;
      s_side[s_L[cur][a.kdiv[n]][p]] = (unsigned char) right;
      s_cnode[cur ^ 1][p] = (unsigned short) (2*jl + right);
    }
    __syncthreads();
    for (int k = 0; k < ndim; k++)
      for (int pb = 0; pb < nw*64; pb += 1024) {
        const int p = pb + tid;
        if (p < nw*64) {        // whole waves take this branch together (nw*64 is a multiple of 64)
          const int flag = (p < cnt) ? (s_side[s_L[cur][k][p]] == 0) : 0;
          const unsigned long long w = __ballot(flag);
          if ((tid & 63) == 0) s_W[k][p >> 6] = w;
        }
      }
    __syncthreads();
    if (tid < 64*ndim) {
      const int k = tid >> 6, lane = tid & 63;
      const unsigned int c = lane < nw ? (unsigned int) __popcll(s_W[k][lane]) : 0u;
      unsigned int inc = c;
      for (int off = 1; off < 64; off <<= 1) { const unsigned int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
      if (lane < nw) s_pre[k][lane] = inc - c;
    }
    __syncthreads();
    for (int p = tid; p < cnt; p += 1024) {
      const int n = nbase + s_cnode[cur][p];
      const int lf = d.cfirst[n] - first0, half = d.cN[n]/2;
      for (int k = 0; k < ndim; k++) {
        const unsigned short id = s_L[cur][k][p];
        const int right = s_side[id];
        const int bp = p & 63, bf = lf & 63;
        const unsigned int rp = s_pre[k][p >> 6] + (bp ? __popcll(s_W[k][p >> 6] & ((1ull << bp) - 1ull)) : 0);
        const unsigned int rf = s_pre[k][lf >> 6] + (bf ? __popcll(s_W[k][lf >> 6] & ((1ull << bf) - 1ull)) : 0);
        const int nleft = (int) (rp - rf);
        const int np = right ? lf + half + ((p - lf) - nleft) : lf + nleft;
        s_L[cur ^ 1][k][np] = id;
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  for (int i = tid; i < cnt; i += 1024) perm_out[first0 + i] = s_gid[s_L[cur][0][i]];
}


// ------------------------------------------------------------------------------------------------
// Exact mode: KDTree::QuickSelect restated (KDTree.cpp:682-750).
//
// The level-wise build above puts the cN/2 particles with the smallest coordinate on the left; WHICH of several
// particles with exactly the median coordinate go left is decided, in the reference, by the dynamics of its in-place
// quick-select on the index array `ids` (pivot = middle element of the current range, moved to the end; one Lomuto pass
// that keeps the "<= pivot" elements in order and rotates the "> pivot" ones; repeat on the side that holds the
// median).  Lattice initial conditions (particle_distribution = cubic_lattice / hexagonal_lattice in a box) have
// thousands of equal coordinates, and the tree - hence the tree force - depends on that order.  When the fast build
// reports such a tie (LevelArgs::tie) the build is redone by this kernel: one workgroup per cell of a level performs
// the SAME sequence of passes and swaps on the SAME initial order (ids[i] = i in the caller's particle order,
// KDTree.cpp:281), a block of up to 1024 consecutive elements per step.  A Lomuto step "ids[j] <= pivot: swap with
// ids[jguess++]" touches position j and the front of the region of "> pivot" elements [jguess, j); as long as a block
// is no longer than that region, its swaps are disjoint and can be done side by side - the result is the serial
// algorithm's, element for element.
// ------------------------------------------------------------------------------------------------
__global__ void k_qsel_init(DevicePtrs d, int *ids, const int *gate)
{
  if (gate && !*gate) return;
  const int p = blockIdx.x*blockDim.x + threadIdx.x;
  if (p < d.N) ids[d.iorig[p]] = p;                 // caller order -> current storage position
}

#define GH_QCAP 2048       /* elements of a quick-select range held in LDS (two buffers) */
// One Lomuto pass in closed form.  "for j: if a[j] <= pivot: swap(a[j], a[jguess++])" leaves (i) everything before the first
// element > pivot (position f0) where it is, (ii) the later "<=" elements packed behind it in their order, (iii) the ">"
// elements as a queue that every later "<=" element rotates by one (its front goes to position j).  Number the operations
// from f0 (op i acts on position f0 + i - 1): every op leaves a token at its position - a push its own element, the k-th
// rotation the element of token k (tokens are consumed in the order they were made) - and the tokens R+1 .. M survive in
// place (R rotations, M ops).  So the element that ends at a position >= f0 + R is found by following "rotation at op i ->
// token (number of rotations up to i)" until a push is reached: a chain of strictly decreasing positions, O(log M) long.
// All positions are independent: the pass is a prefix sum and a gather instead of M/64 dependent block steps.
// state of one cell's quick-select while its range is wider than a workgroup's LDS (k_qw_*, below)
struct QSel {
  int left, right;                   // current range (the pivot search continues in it)
  int cur, done, conv;               // buffer the range lives in; range fits LDS or converged; converged (pivot landed on the median)
  int f0, nb4, jguess, none;         // this pass: first "> pivot" position, "<=" elements before it, where the pivot lands, no element > pivot
  int kg, fewg;                      // ... number of "> pivot" elements; few enough for the segment walk of k_qw_gather
  int nleft, nright, nconv;          // the range after this pass
  double rpiv;                       // last pivot value
};
struct QWide { const QSel *st; double *k0, *k1; int *i0, *i1; };

__global__ __launch_bounds__(1024) void k_qselect_level(DevicePtrs d, int level, int *ids, double *keys, double *dbbmin, double *dbbmax,
                                                        int *kdiv, const int *gate, int qcap, QWide wide)
{
  if (gate && !*gate) return;
  __shared__ int s_wave[16];
  __shared__ int s_tot;
  __shared__ double s_piv;
  // LDS buffers sized by the launch: qcap = min(largest cell of the level, GH_QCAP) elements (28 bytes each), so that the many
  // small cells of the deep levels do not each reserve the 56 KB the top levels need
  extern __shared__ double s_dyn[];
  double *const s_qk[2] = {s_dyn, s_dyn + qcap};
  int *const s_qi[2] = {(int*) (s_dyn + 2*qcap), (int*) (s_dyn + 2*qcap) + qcap};
  int *const s_rk = (int*) (s_dyn + 2*qcap) + 2*qcap;
  const int n = (1 << level) - 1 + blockIdx.x;
  const int first = d.cfirst[n], cnt = d.cN[n];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
  double rkmax = 0.0;
  int kd = 0;
  for (int k = 0; k < d.ndim; k++) { const double ext = dbbmax[n*3 + k] - dbbmin[n*3 + k]; if (ext > rkmax) { rkmax = ext; kd = k; } }
  if (!wide.st) for (int p = first + tid; p < first + cnt; p += nt) keys[p] = d.f[D_RX + kd][ids[p]];
  __syncthreads();
  int left = first, right = first + cnt - 1;
  const int jpivot = first + cnt/2;
  double rpivot = dbbmin[n*3 + kd];
  // the wide passes (k_qw_*) have narrowed the range (as a rule to what fits the LDS buffers; a range of many equal
  // coordinates, which the reference's quick-select peels one element per pass, is handed over as it is after a few dozen
  // passes) or finished it: go on from their state, in their buffer, and copy what was left of the range to `ids` at the end
  int *const ids_out = ids;
  bool wconv = false;
  int wleft = 0, wright = -1;
  if (wide.st) {
    const QSel S = wide.st[blockIdx.x];
    left = S.left; right = S.right; rpivot = S.rpiv; wconv = S.conv != 0;
    wleft = S.left; wright = S.conv ? S.left - 1 : S.right;
    keys = S.cur ? wide.k1 : wide.k0; ids = S.cur ? wide.i1 : wide.i0;
  }
  // block-wide: exclusive prefix of a flag over the threads, and the total
  auto scan = [&](bool f, int &pre) -> int {
    const unsigned long long m = __ballot(f);
    if (lane == 0) s_wave[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) { int run = 0; for (int w = 0; w < nw; w++) { const int c = s_wave[w]; s_wave[w] = run; run += c; } s_tot = run; }
    __syncthreads();
    pre = s_wave[wv] + __popcll(m & (lane ? ((~0ull) >> (64 - lane)) : 0ull));
    const int tot = s_tot;
    __syncthreads();
    return tot;
  };
  // ... of an integer
  auto iscan = [&](int v, int &pre) -> int {
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int w = 0; w < nw; w++) { const int c = s_wave[w]; s_wave[w] = run; run += c; } s_tot = run; }
    __syncthreads();
    pre = s_wave[wv] + inc - v;
    const int tot = s_tot;
    __syncthreads();
    return tot;
  };
  // closed-form Lomuto pass over the LDS positions [lo, hi) of buffer A against pivot rp (see above): the result goes to B
  // (returns true) unless no element exceeds the pivot, when nothing moves (returns false); jguess = where the "> pivot" run starts
  auto cf_pass = [&](const double *A, const int *Ai, double *B, int *Bi, int lo, int hi, double rp, int &jguess) -> bool {
    const int mc = hi - lo;
    const int E = (mc + nt - 1)/nt;                          // consecutive positions per thread
    const int q0 = lo + tid*E, q1 = min(q0 + E, hi);
    int nle = 0, fgt = 0x7fffffff;
    for (int q = q0; q < q1; q++) { if (A[q] <= rp) nle++; else if (fgt == 0x7fffffff) fgt = q; }
    int pre;
    const int tot_le = iscan(nle, pre);
    { int run = pre; for (int q = q0; q < q1; q++) { if (A[q] <= rp) run++; s_rk[q] = run; } }      // inclusive count of "<=" in [lo, q]
    for (int off = 32; off > 0; off >>= 1) fgt = min(fgt, __shfl_xor(fgt, off, 64));                  // first element > pivot
    if (lane == 0) s_wave[wv] = fgt;
    __syncthreads();
    int f0 = 0x7fffffff;
    for (int w = 0; w < nw; w++) f0 = min(f0, s_wave[w]);
    __syncthreads();
    if (f0 == 0x7fffffff) { jguess = hi; return false; }     // nothing > pivot: every swap was with itself
    const int nb4 = f0 - lo;                                 // all "<=": untouched
    const int R = tot_le - nb4;                              // rotations
    jguess = f0 + R;
    for (int q = q0; q < q1; q++) {
      if (q < f0) { B[q] = A[q]; Bi[q] = Ai[q]; }
      else if (A[q] <= rp) { const int t = f0 + (s_rk[q] - nb4) - 1; B[t] = A[q]; Bi[t] = Ai[q]; }
      if (q >= jguess) {
        int src = q;
        while (A[src] <= rp) src = f0 + (s_rk[src] - nb4) - 1;
        B[q] = A[src]; Bi[q] = Ai[src];
      }
    }
    return true;
  };
  if (cnt > 0 && !wconv) {
    bool converged = false;
    // ---- ranges larger than the LDS buffers: block steps on the global arrays
    while (right - left + 1 > qcap) {
      const int jg0 = (left + right)/2;                       // pivot guess: the middle element ...
      if (tid == 0) {
        s_piv = keys[jg0];
        const int ti = ids[jg0]; ids[jg0] = ids[right]; ids[right] = ti;        // ... parked at the end of the range
        const double tk = keys[jg0]; keys[jg0] = keys[right]; keys[right] = tk;
      }
      __syncthreads();
      rpivot = s_piv;
      int jguess = left, j = left;
      while (j < right) {
        const int g = j - jguess;                             // elements > pivot seen so far: they occupy [jguess, j)
        if (g >= nt) {
          // a block step: the next nt elements against a run of "> pivot" elements at least as long - the k-th "<=" element
          // of the block swaps with the k-th element of the run, all swaps disjoint
          const int w = min(nt, right - j);
          bool le = false; int idb = 0; double kb = 0.0;
          if (tid < w) { kb = keys[j + tid]; idb = ids[j + tid]; le = kb <= rpivot; }
          int pre;
          const int c = scan(le, pre);
          int idt = 0; double kt = 0.0;
          if (le) { idt = ids[jguess + pre]; kt = keys[jguess + pre]; }
          __syncthreads();
          if (le) { ids[jguess + pre] = idb; keys[jguess + pre] = kb; ids[j + tid] = idt; keys[j + tid] = kt; }
          __syncthreads();
          jguess += c; j += w;
        }
        else {
          // a short run (always at the start of a pass, and throughout a pass with few "> pivot" elements) would make the
          // block steps as narrow as the run: take the run and what follows it, up to qcap positions, in closed form
          // instead (the run's elements are "pushes" like any other "> pivot" element)
          const int hn = min(qcap, right - jguess);
          for (int r = tid; r < hn; r += nt) { s_qk[0][r] = keys[jguess + r]; s_qi[0][r] = ids[jguess + r]; }
          __syncthreads();
          int jg;
          const bool moved = cf_pass(s_qk[0], s_qi[0], s_qk[1], s_qi[1], 0, hn, rpivot, jg);
          __syncthreads();
          if (moved) for (int r = tid; r < hn; r += nt) { keys[jguess + r] = s_qk[1][r]; ids[jguess + r] = s_qi[1][r]; }
          __syncthreads();
          j = jguess + hn; jguess = jguess + jg;
        }
      }
      if (tid == 0) {                                          // the pivot goes between the two sides
        const int ti = ids[right]; ids[right] = ids[jguess]; ids[jguess] = ti;
        const double tk = keys[right]; keys[right] = keys[jguess]; keys[jguess] = tk;
      }
      __syncthreads();
      if (jguess < jpivot) left = jguess + 1;
      else if (jguess > jpivot) right = jguess - 1;
      else { converged = true; break; }
    }
    // ---- the range fits: stage it, closed-form passes between two LDS buffers
    if (!converged) {
      const int base = left, m0 = right - left + 1;
      for (int r = tid; r < m0; r += nt) { s_qk[0][r] = keys[base + r]; s_qi[0][r] = ids[base + r]; }
      __syncthreads();
      int cur = 0;
      int lo = 0, hi = m0 - 1;
      const int jp = jpivot - base;
      for (;;) {
        double *A = s_qk[cur], *B = s_qk[cur ^ 1];
        int *Ai = s_qi[cur], *Bi = s_qi[cur ^ 1];
        const int jg0 = (lo + hi)/2;
        if (tid == 0) {
          s_piv = A[jg0];
          const int ti = Ai[jg0]; Ai[jg0] = Ai[hi]; Ai[hi] = ti;
          const double tk = A[jg0]; A[jg0] = A[hi]; A[hi] = tk;
        }
        __syncthreads();
        rpivot = s_piv;
        int jguess;
        if (cf_pass(A, Ai, B, Bi, lo, hi, rpivot, jguess)) {
          if (tid == 0) { B[hi] = A[hi]; Bi[hi] = Ai[hi]; }
          __syncthreads();
          cur ^= 1;
        }
        {
          double *C = s_qk[cur]; int *Ci = s_qi[cur];
          if (tid == 0) {                                      // the pivot goes between the two sides
            const int ti = Ci[hi]; Ci[hi] = Ci[jguess]; Ci[jguess] = ti;
            const double tk = C[hi]; C[hi] = C[jguess]; C[jguess] = tk;
          }
        }
        __syncthreads();
        // (positions outside [lo, hi] of the other buffer are stale: carry the settled ones along when the buffers swap)
        if (jguess < jp) {
          const int nlo = jguess + 1;
          for (int q = lo + tid; q < nlo; q += nt) { s_qk[cur ^ 1][q] = s_qk[cur][q]; s_qi[cur ^ 1][q] = s_qi[cur][q]; }
          lo = nlo;
        }
        else if (jguess > jp) {
          const int nhi = jguess - 1;
          for (int q = nhi + 1 + tid; q <= hi; q += nt) { s_qk[cur ^ 1][q] = s_qk[cur][q]; s_qi[cur ^ 1][q] = s_qi[cur][q]; }
          hi = nhi;
        }
        else break;
        __syncthreads();
      }
      __syncthreads();
      for (int r = tid; r < m0; r += nt) { keys[base + r] = s_qk[cur][r]; ids[base + r] = s_qi[cur][r]; }
    }
  }
  if (wide.st) {
    __syncthreads();
    for (int p = wleft + tid; p <= wright; p += nt) ids_out[p] = ids[p];
  }
  if (tid == 0) {
    const int c1 = 2*n + 1, c2 = 2*n + 2;
    for (int k = 0; k < 3; k++) {
      dbbmin[c1*3 + k] = dbbmin[n*3 + k]; dbbmax[c1*3 + k] = (k == kd) ? rpivot : dbbmax[n*3 + k];
      dbbmin[c2*3 + k] = (k == kd) ? rpivot : dbbmin[n*3 + k]; dbbmax[c2*3 + k] = dbbmax[n*3 + k];
    }
    kdiv[n] = kd;
  }
}


// ------------------------------------------------------------------------------------------------
// Wide quick-select passes: the top levels of an exact-mode build (every build of a sink run).  One workgroup per cell
// walks a range of a million elements in two thousand dependent block steps per pass (37 of the 77 ms of a 2 000 000
// particle sink step); the closed form of a Lomuto pass (above) has no such dependence, so here the WHOLE DEVICE does one
// pass of every cell of a level at a time: count "<= pivot" per block (k_qw_count), prefix over the blocks and the pass's
// outcome (k_qw_scan), ranks (k_qw_rank), scatter / chain gather into the other buffer (k_qw_gather).  Element for element
// what k_qselect_level's block steps - and the reference's serial loop - produce, including the pivot's trip to the end of
// the range and back (done virtually: a(q) below).  Positions that leave the range are final and go straight to `out`;
// when a range fits a workgroup's LDS, k_qselect_level finishes it from the state left here.
// ------------------------------------------------------------------------------------------------
#define GH_QW_BLOCK 1024
struct QWArgs {
  QSel *st_cur, *st_next;            // state read by this pass / written for the next one
  double *k[2]; int *i[2];           // the two buffers (keys, ids)
  int *rk;                           // "<=" rank of every position of the pass ("> pivot": minus its number among them)
  int *gp;                           // gp[lo + j - 1]: op index of the j-th "> pivot" element of the cell's pass
  int *blkcnt, *blkfirst, *blkpre;   // per (cell, block)
  int *out;                          // final ids
  int nb, level, qcap;
};

__global__ void k_qw_init(DevicePtrs d, QWArgs a, const int *ids, const double *dbbmin, const double *dbbmax)
{
  const int c = blockIdx.y, n = (1 << a.level) - 1 + c;
  const int first = d.cfirst[n], cnt = d.cN[n];
  double rkmax = 0.0; int kd = 0;
  for (int k = 0; k < d.ndim; k++) { const double ext = dbbmax[n*3 + k] - dbbmin[n*3 + k]; if (ext > rkmax) { rkmax = ext; kd = k; } }
  const int q = blockIdx.x*blockDim.x + threadIdx.x;
  if (q < cnt) { const int id = ids[first + q]; a.i[0][first + q] = id; a.k[0][first + q] = d.f[D_RX + kd][id]; }
  if (q == 0) {
    QSel S;
    S.left = first; S.right = first + cnt - 1; S.cur = 0; S.conv = 0;
    S.done = cnt <= a.qcap ? 1 : 0;
    S.f0 = S.nb4 = S.jguess = S.none = 0; S.kg = 0; S.fewg = 0; S.nleft = S.left; S.nright = S.right; S.nconv = 0;
    S.rpiv = dbbmin[n*3 + kd];
    a.st_cur[c] = S;
  }
}

// a(q): the range as the pass sees it - the pivot guess (middle element) has changed places with the last element
__device__ __forceinline__ double qw_key(const double *A, int q, int jg0, int hi) { return A[q == jg0 ? hi : q]; }
__device__ __forceinline__ int qw_id(const int *Ai, int q, int jg0, int hi) { return Ai[q == jg0 ? hi : q]; }

__global__ __launch_bounds__(GH_QW_BLOCK) void k_qw_count(QWArgs a)
{
  __shared__ int s_c[GH_QW_BLOCK/64], s_f[GH_QW_BLOCK/64];
  const int c = blockIdx.y;
  const QSel S = a.st_cur[c];
  if (S.done) return;
  const int lo = S.left, hi = S.right;
  const int q = lo + blockIdx.x*GH_QW_BLOCK + threadIdx.x;
  if (lo + (int) blockIdx.x*GH_QW_BLOCK >= hi) return;
  const double *A = a.k[S.cur];
  const int jg0 = (lo + hi)/2;
  const double rp = A[jg0];
  const bool in = q < hi;
  const bool le = in && qw_key(A, q, jg0, hi) <= rp;
  const unsigned long long m = __ballot(le), g = __ballot(in && !le);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { s_c[wv] = __popcll(m); s_f[wv] = g ? q + (__ffsll((long long) g) - 1) : 0x7fffffff; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int cnt = 0, f = 0x7fffffff;
    for (int w = 0; w < GH_QW_BLOCK/64; w++) { cnt += s_c[w]; f = min(f, s_f[w]); }
    a.blkcnt[(size_t) c*a.nb + blockIdx.x] = cnt; a.blkfirst[(size_t) c*a.nb + blockIdx.x] = f;
  }
}

// one workgroup per cell: prefix over the blocks' counts, outcome of the pass, the state of the next pass
__global__ __launch_bounds__(1024) void k_qw_scan(DevicePtrs d, QWArgs a)
{
  __shared__ int s_w[16], s_fw[16];
  const int c = blockIdx.x, n = (1 << a.level) - 1 + c;
  QSel S = a.st_cur[c];
  if (S.done) { if (threadIdx.x == 0) a.st_next[c] = S; return; }
  const int lo = S.left, hi = S.right;
  const int nbl = (hi - lo + GH_QW_BLOCK - 1)/GH_QW_BLOCK;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int run = 0, f0 = 0x7fffffff;
  for (int b0 = 0; b0 < nbl; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const int v = b < nbl ? a.blkcnt[(size_t) c*a.nb + b] : 0;
    if (b < nbl) f0 = min(f0, a.blkfirst[(size_t) c*a.nb + b]);
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) s_w[wv] = inc;
    __syncthreads();
    int wpre = 0, tot = 0;
    for (int w = 0; w < 16; w++) { if (w < wv) wpre += s_w[w]; tot += s_w[w]; }
    if (b < nbl) a.blkpre[(size_t) c*a.nb + b] = run + wpre + inc - v;
    run += tot;
    __syncthreads();
  }
  for (int off = 32; off > 0; off >>= 1) f0 = min(f0, __shfl_xor(f0, off, 64));
  if (lane == 0) s_fw[wv] = f0;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 0; w < 16; w++) f0 = min(f0, s_fw[w]);
    const int tot_le = run;
    const double *A = a.k[S.cur];
    const int jpivot = d.cfirst[n] + d.cN[n]/2;
    S.rpiv = A[(lo + hi)/2];
    S.none = f0 == 0x7fffffff ? 1 : 0;
    if (S.none) { S.f0 = hi; S.nb4 = hi - lo; S.jguess = hi; }
    else { S.f0 = f0; S.nb4 = f0 - lo; S.jguess = f0 + (tot_le - S.nb4); }
    S.kg = (hi - lo) - tot_le;
    // a chain of k_qw_gather is ~M/K hops long with K "> pivot" elements among M: with few of them (a pivot near the top of
    // the range - one pass in a few dozen, but then thousands of dependent hops: the 20 longest of 1 248 launches were half
    // of the kernel's time) it walks segment by segment instead, at most K steps
    S.fewg = (!S.none && (long long) S.kg*S.kg < (long long) (hi - lo)) ? 1 : 0;
    S.nleft = lo; S.nright = hi; S.nconv = 0;
    if (S.jguess < jpivot) S.nleft = S.jguess + 1;
    else if (S.jguess > jpivot) S.nright = S.jguess - 1;
    else S.nconv = 1;
    a.st_cur[c] = S;
    QSel T = S;
    T.left = S.nleft; T.right = S.nright; T.cur = S.cur ^ 1; T.conv = S.nconv;
    T.done = (S.nconv || S.nright - S.nleft + 1 <= a.qcap) ? 1 : 0;
    a.st_next[c] = T;
  }
}

__global__ __launch_bounds__(GH_QW_BLOCK) void k_qw_rank(QWArgs a)
{
  __shared__ int s_c[GH_QW_BLOCK/64];
  const int c = blockIdx.y;
  const QSel S = a.st_cur[c];
  if (S.done || S.none) return;
  const int lo = S.left, hi = S.right;
  if (lo + (int) blockIdx.x*GH_QW_BLOCK >= hi) return;
  const int q = lo + blockIdx.x*GH_QW_BLOCK + threadIdx.x;
  const double *A = a.k[S.cur];
  const int jg0 = (lo + hi)/2;
  const double rp = S.rpiv;
  const bool le = q < hi && qw_key(A, q, jg0, hi) <= rp;
  const unsigned long long m = __ballot(le);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_c[wv] = __popcll(m);
  __syncthreads();
  int pre = a.blkpre[(size_t) c*a.nb + blockIdx.x];
  for (int w = 0; w < wv; w++) pre += s_c[w];
  // "<=" elements: inclusive count of "<=" in [lo, q]; "> pivot" elements: minus their number among the "> pivot" ones (the
  // chains of k_qw_gather stop at a negative word, and one 4-byte word per hop is all they read)
  if (q < hi) {
    const int linc = pre + __popcll(m & ((2ull << lane) - 1ull));
    if (le) a.rk[q] = linc;
    else {
      const int i = q - S.f0 + 1, j = i - (linc - S.nb4);     // op number, number among the "> pivot" elements (both from 1)
      a.rk[q] = -j;
      if (S.fewg) a.gp[lo + j - 1] = i;
    }
  }
}

__global__ __launch_bounds__(GH_QW_BLOCK) void k_qw_gather(QWArgs a)
{
  const int c = blockIdx.y;
  const QSel S = a.st_cur[c];
  if (S.done) return;
  const int lo = S.left, hi = S.right;
  const int q = lo + blockIdx.x*GH_QW_BLOCK + threadIdx.x;
  if (q > hi) return;
  const double *A = a.k[S.cur]; const int *Ai = a.i[S.cur];
  double *B = a.k[S.cur ^ 1]; int *Bi = a.i[S.cur ^ 1];
  const int jg0 = (lo + hi)/2;
  const double rp = S.rpiv;
  // a position that stays in the range goes to the other buffer, one that leaves it is final
  auto put = [&](int p, double key, int id) {
    if (!S.nconv && p >= S.nleft && p <= S.nright) { B[p] = key; Bi[p] = id; }
    else a.out[p] = id;
  };
  if (q == hi) { put(S.jguess, rp, Ai[jg0]); return; }       // the pivot lands between the two sides
  const double aq = qw_key(A, q, jg0, hi);
  const int iq = qw_id(Ai, q, jg0, hi);
  if (S.none || q < S.f0) { put(q, aq, iq); return; }
  int cd = a.rk[q];
  if (cd >= 0) put(S.f0 + (cd - S.nb4) - 1, aq, iq);
  if (q >= S.jguess) {
    int src = q;
    if (!S.fewg) while (cd >= 0) { src = S.f0 + (cd - S.nb4) - 1; cd = a.rk[src]; }
    else if (cd >= 0) {
      // ops between the j-th and the (j+1)-th "> pivot" element are all "<=" and each of them sends the chain j ops back:
      // the whole stretch in one step - landing on the j-th "> pivot" op itself ends the chain, otherwise it goes on below it
      int i = cd - S.nb4;                                  // op the first hop leads to
      for (;;) {
        src = S.f0 + i - 1;
        const int c2 = a.rk[src];
        if (c2 < 0) break;
        const int j = i - (c2 - S.nb4);
        const int gj = a.gp[lo + j - 1];
        const int dd = i - gj, nn = dd/j;
        if (dd - nn*j == 0) { src = S.f0 + gj - 1; break; }
        i -= (nn + 1)*j;
      }
    }
    put(q == S.jguess ? hi : q, qw_key(A, src, jg0, hi), qw_id(Ai, src, jg0, hi));            // (the element at jguess changes places with the pivot)
  }
}

__global__ void k_qw_pending(const QSel *st, int ncells, int *out)
{
  int n = 0;
  for (int c = threadIdx.x; c < ncells; c += blockDim.x) n += st[c].done ? 0 : 1;
  for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

__global__ void k_tie_reset(int *t) { t[1] |= t[0]; t[0] = 0; }

__global__ void k_copy_if(const int *src, int *dst, int n, const int *gate)
{
  if (gate && !*gate) return;
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// enqueue the exact build behind the fast one; every kernel returns at once unless *gate (the fast build's tie flag) is set.
// Leaves the reference's final `ids` order in perm_out.
static int exact_build_gated(gh_ctx *ctx, int *perm_out, const int *gate)
{
  const int N = (int) ctx->N;
  if (!ctx->qs_ids) {
    GH_CHECK(ctx, hipMalloc((void**) &ctx->qs_ids, sizeof(int)*(size_t) ctx->Ncap));
    GH_CHECK(ctx, hipMalloc((void**) &ctx->qs_keys, sizeof(double)*(size_t) ctx->Ncap));
  }
  DevicePtrs d = gh_dev(ctx);
  hipStream_t s = ctx->stream;
  hipLaunchKernelGGL(k_qsel_init, dim3(cdiv(N, 256)), dim3(256), 0, s, d, ctx->qs_ids, gate);
  // levels whose cells hold more than `wide_min` elements run their passes device-wide (ungated builds only: the host reads a
  // "cells still going" word between batches of passes)
  const int wide_min = getenv("GH_QSEL_WIDE_MIN") ? atoi(getenv("GH_QSEL_WIDE_MIN")) : 16384;
  for (int l = 0; l < ctx->ltot; l++) {
    int mx = 1;
    for (int j = 0; j < (1 << l); j += std::max(1, (1 << l)/64)) mx = std::max(mx, ctx->h_cN[(1 << l) - 1 + j]);   // cells of a level differ by at most 1
    mx += 1;
    int bs = 64;
    while (bs < mx && bs < 1024) bs <<= 1;
    const int qcap = std::min(std::max(mx, 2), GH_QCAP);
    QWide wide = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (!gate && mx > wide_min && mx > GH_QCAP && wide_min > 0) {
      const int ncells = 1 << l;
      if (!ctx->qw_k[0]) {
        for (int b = 0; b < 2; b++) {
          GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_k[b], sizeof(double)*(size_t) ctx->Ncap));
          GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_i[b], sizeof(int)*(size_t) ctx->Ncap));
        }
        GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_rk, sizeof(int)*(size_t) ctx->Ncap));
        GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_gp, sizeof(int)*(size_t) ctx->Ncap));
        // per (cell, block) words: at most Ncap/1024 + one per cell; states: two per cell of the deepest wide level
        ctx->qw_words = (size_t) ctx->Ncap/GH_QW_BLOCK + 2*(size_t) (ctx->Ncap/std::max(wide_min, 1) + 2) + 1024;
        GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_blk, sizeof(int)*3*ctx->qw_words));
        GH_CHECK(ctx, hipMalloc((void**) &ctx->qw_st, sizeof(QSel)*2*ctx->qw_words));
      }
      QWArgs a;
      a.k[0] = ctx->qw_k[0]; a.k[1] = ctx->qw_k[1]; a.i[0] = ctx->qw_i[0]; a.i[1] = ctx->qw_i[1];
      a.rk = ctx->qw_rk; a.gp = ctx->qw_gp; a.out = ctx->qs_ids;
      a.nb = cdiv(mx, GH_QW_BLOCK); a.level = l; a.qcap = GH_QCAP;
      if ((size_t) a.nb*ncells > ctx->qw_words || (size_t) ncells > ctx->qw_words) return gh_fail(ctx, GH_ERR_CAPACITY, "exact tree build: wide quick-select scratch too small");
      a.blkcnt = ctx->qw_blk; a.blkfirst = ctx->qw_blk + ctx->qw_words; a.blkpre = ctx->qw_blk + 2*ctx->qw_words;
      QSel *st[2] = {(QSel*) ctx->qw_st, (QSel*) ctx->qw_st + ctx->qw_words};
      int par = 0;
      a.st_cur = st[0]; a.st_next = st[1];
      hipLaunchKernelGGL(k_qw_init, dim3(a.nb, ncells), dim3(GH_QW_BLOCK), 0, s, d, a, ctx->qs_ids, ctx->dbbmin, ctx->dbbmax);
      int *pending = ctx->d_blk + 18;
      // (8 rounds of 8 passes: a range still wider than the LDS buffers after that is one with many equal coordinates, which
      //  every pass shortens by a single element; k_qselect_level's in-place block steps take it from there)
      for (int round = 0; round < 8; round++) {
        for (int pass = 0; pass < 8; pass++) {
          a.st_cur = st[par]; a.st_next = st[par ^ 1];
          hipLaunchKernelGGL(k_qw_count, dim3(a.nb, ncells), dim3(GH_QW_BLOCK), 0, s, a);
          hipLaunchKernelGGL(k_qw_scan, dim3(ncells), dim3(1024), 0, s, d, a);
          hipLaunchKernelGGL(k_qw_rank, dim3(a.nb, ncells), dim3(GH_QW_BLOCK), 0, s, a);
          hipLaunchKernelGGL(k_qw_gather, dim3(a.nb, ncells), dim3(GH_QW_BLOCK), 0, s, a);
          par ^= 1;
        }
        GH_CHECK(ctx, hipMemsetAsync(pending, 0, sizeof(int), s));
        hipLaunchKernelGGL(k_qw_pending, dim3(1), dim3(256), 0, s, st[par], ncells, pending);
        int np = 0;
        GH_CHECK(ctx, hipMemcpyAsync(&np, pending, sizeof(int), hipMemcpyDeviceToHost, s));
        GH_CHECK(ctx, hipStreamSynchronize(s));
        if (np == 0) break;
      }
      wide.st = st[par]; wide.k0 = a.k[0]; wide.k1 = a.k[1]; wide.i0 = a.i[0]; wide.i1 = a.i[1];
    }
    hipLaunchKernelGGL(k_qselect_level, dim3(1 << l), dim3(bs), (size_t) 28*qcap, s, d, l, ctx->qs_ids, ctx->qs_keys, ctx->dbbmin, ctx->dbbmax, ctx->kdiv, gate, qcap, wide);
  }
  hipLaunchKernelGGL(k_copy_if, dim3(cdiv(N, 256)), dim3(256), 0, s, ctx->qs_ids, perm_out, N, gate);
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// gather all particle arrays into tree order
// ------------------------------------------------------------------------------------------------
__global__ void k_permute(double **tab /* [2*D_COUNT]: src then dst */, const int *perm, int p0, int pn, unsigned long long live)
{
  const int f = blockIdx.y;
  if (!((live >> f) & 1ull)) return;                     // an array every step rewrites before it reads it (gh_tree_build_impl)
  const double *src = tab[f];
  double *dst = tab[D_COUNT + f];
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < pn; i += gridDim.x*blockDim.x) dst[p0 + i] = src[perm[p0 + i]];
}

__global__ void k_permute_int(const int *src, int *dst, const int *perm, int p0, int pn)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < pn) dst[p0 + i] = src[perm[p0 + i]];
}

__global__ void k_pack_posm(DevicePtrs d)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  double4 v;
  v.x = d.f[D_RX][i];
  v.y = d.ndim > 1 ? d.f[D_RY][i] : 0.0;
  v.z = d.ndim > 2 ? d.f[D_RZ][i] : 0.0;
  v.w = d.f[D_M][i];
  d.posm[i] = v;
}

// ------------------------------------------------------------------------------------------------
// stocking                                                    (KDTree.cpp:808-1083, 1128-1208)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void finish_cell(const DevicePtrs &d, const CellBox &b, CellGeo &g, double hmax, double thetamaxsqd)
{
  double dr2 = 0.0;
  _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
    g.rcell[k] = 0.5*(b.bbmin[k] + b.bbmax[k]);
    const double dr = 0.5*(b.bbmax[k] - b.bbmin[k]);
    dr2 += dr*dr;
  }
  g.cdistsqd = fmax(dr2, hmax*hmax)/thetamaxsqd;
  g.rmax = sqrt(dr2);
}

// q += m (3 dr dr - |dr|^2 1): the independent components the reference keeps (KDTree.cpp:929-944)
__device__ __forceinline__ void add_quad(double *q, double mi, const double *dr, int nd)
{
  double drsqd = dr[0]*dr[0];
  if (nd > 1) drsqd += dr[1]*dr[1];
  if (nd > 2) drsqd += dr[2]*dr[2];
  q[0] += mi*(3.0*dr[0]*dr[0] - drsqd);
  if (nd > 1) { q[1] += mi*3.0*dr[0]*dr[1]; q[2] += mi*(3.0*dr[1]*dr[1] - drsqd); }
  if (nd > 2) { q[3] += mi*3.0*dr[2]*dr[0]; q[4] += mi*3.0*dr[2]*dr[1]; }
}

// cell.mac of the eigenvalue MAC from the cell's quadrupole (KDTree.cpp:1054-1076)
__device__ __forceinline__ double eigen_mac(const double *q, int nd, double macerror)
{
  double lambda;
  if (nd == 3) {
    const double p = q[0]*q[2] - (q[0] + q[2])*(q[0] + q[2]) - q[1]*q[1] - q[3]*q[3] - q[4]*q[4];
    lambda = p >= 0.0 ? 0.0 : 2.0*sqrt(-p/3.0);
  }
  else if (nd == 2) {
    const double p = (q[0] - q[2])*(q[0] - q[2]) + 4*q[1]*q[1];
    lambda = 0.5*fmax(q[0] + q[2] + sqrt(p), 0.);
  }
  else lambda = fabs(q[0]);
  return pow(0.5*lambda/macerror, 0.66666666666666);
}

// what a parent needs of a child (held in LDS between the levels of one fused launch)
struct SRec {
  double hbmin[3], hbmax[3], hmax;
  double bbmin[3], bbmax[3];
  double com[3], m;
  double q[5];
  int N, pad;
};

__device__ __forceinline__ void srec_load(const DevicePtrs &d, int c, int hmax_only, SRec &r)
{
  const CellH h = d.ch[c];
  for (int k = 0; k < 3; k++) { r.hbmin[k] = h.hbmin[k]; r.hbmax[k] = h.hbmax[k]; }
  r.hmax = h.hmax;
  r.N = d.cN[c];
  if (hmax_only) return;
  const CellBox b = d.cbox[c];
  const CellCom m = d.ccom[c];
  for (int k = 0; k < 3; k++) { r.bbmin[k] = b.bbmin[k]; r.bbmax[k] = b.bbmax[k]; r.com[k] = m.com[k]; }
  r.m = m.m;
  if (d.cquad) { const CellQuad cq = d.cquad[c]; for (int k = 0; k < 5; k++) r.q[k] = cq.q[k]; }
}

// leaf cell from its particles (StockCellProperties KDTree.cpp:808-930); hmax_only = 1 restates
// KDTree::UpdateHmaxValues: only hmax and hbox are refreshed
__device__ __forceinline__ void stock_leaf(const DevicePtrs &d, int n, double kernrange, double thetamaxsqd, int hmax_only, SRec &o)
{
  const int first = d.cfirst[n], cnt = d.cN[n];
  CellH hh;
  hh.hmax = 0.0; hh.pad = 0.0;
  for (int k = 0; k < 3; k++) { hh.hbmin[k] = BIG; hh.hbmax[k] = -BIG; }
  CellBox b; CellGeo c; CellCom cm;
  if (!hmax_only) {
    cm.m = 0.0; c.rmax = 0.0; c.cdistsqd = BIG; c.mac = 0.0; b.pad = 0.0;
    for (int k = 0; k < 3; k++) { cm.com[k] = 0.0; c.rcell[k] = 0.0; b.bbmin[k] = BIG; b.bbmax[k] = -BIG; }
    c.first = first; c.N = cnt; b.first = first; b.N = cnt;
  }
  for (int i = first; i < first + cnt; i++) {
    const double h = d.f[D_H][i];
    hh.hmax = fmax(hh.hmax, h);
    const double m = d.f[D_M][i];
    if (!hmax_only) cm.m += m;
    _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) {
      const double x = d.f[D_RX + k][i];
      if (!hmax_only) {
        cm.com[k] += m*x;
        if (x < b.bbmin[k]) b.bbmin[k] = x;
        if (x > b.bbmax[k]) b.bbmax[k] = x;
      }
      if (x - kernrange*h < hh.hbmin[k]) hh.hbmin[k] = x - kernrange*h;
      if (x + kernrange*h > hh.hbmax[k]) hh.hbmax[k] = x + kernrange*h;
    }
  }
  d.ch[n] = hh;
  if (!hmax_only) {
    if (cm.m > 0) _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) cm.com[k] /= cm.m;
    if (cnt > 0) finish_cell(d, b, c, hh.hmax, thetamaxsqd);
    c.hmax = hh.hmax;
    d.cbox[n] = b; d.cgeo[n] = c; d.ccom[n] = cm;
    if (d.cquad) {                                   // KDTree.cpp:921-950
      CellQuad cq;
      for (int k = 0; k < 5; k++) cq.q[k] = 0.0;
      for (int k = 0; k < 3; k++) cq.pad[k] = 0.0;
      for (int i = first; i < first + cnt; i++) {
        double dr[3] = {0.0, 0.0, 0.0};
        _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) dr[k] = d.f[D_RX + k][i] - cm.com[k];
        add_quad(cq.q, d.f[D_M][i], dr, d.ndim);
      }
      d.cquad[n] = cq;
      for (int k = 0; k < 5; k++) o.q[k] = cq.q[k];
      if (d.mac_stock == GH_MAC_EIGENMAC) d.cgeo[n].mac = eigen_mac(cq.q, d.ndim, d.macerror);
    }
    if (d.leaf_amin && d.mac_stock == GH_MAC_GADGET2) {   // cell.amin, KDTree.cpp:899-901 (read by the walk of THIS leaf)
      double amin = BIG;
      for (int i = first; i < first + cnt; i++) {
        double a2 = d.f[D_ATX][i]*d.f[D_ATX][i];
        if (d.ndim > 1) a2 += d.f[D_ATX + 1][i]*d.f[D_ATX + 1][i];
        if (d.ndim > 2) a2 += d.f[D_ATX + 2][i]*d.f[D_ATX + 2][i];
        amin = fmin(amin, sqrt(a2));
      }
      d.leaf_amin[n - (d.gtot - 1)] = amin;
    }
    if (d.leaf_amin && d.mac_stock == GH_MAC_EIGENMAC) {  // cell.macfactor, KDTree.cpp:902-903
      double mf = 0.0;
      for (int i = first; i < first + cnt; i++) mf = fmax(mf, pow(d.f[D_GPOT][i], -0.66666666666666666666666));
      d.leaf_amin[n - (d.gtot - 1)] = mf;
    }
  }
  else d.cgeo[n].hmax = hh.hmax;
  for (int k = 0; k < 3; k++) { o.hbmin[k] = hh.hbmin[k]; o.hbmax[k] = hh.hbmax[k]; }
  o.hmax = hh.hmax; o.N = cnt;
  if (!hmax_only) {
    for (int k = 0; k < 3; k++) { o.bbmin[k] = b.bbmin[k]; o.bbmax[k] = b.bbmax[k]; o.com[k] = cm.com[k]; }
    o.m = cm.m;
  }
}

// parent n from its two children (StockCellProperties, KDTree.cpp:931-1083; UpdateHmaxValues :1170-1200);
// writes the parent's global records and returns what the next level up needs
__device__ __forceinline__ void stock_combine(const DevicePtrs &d, int n, const SRec &r1, const SRec &r2, SRec &o,
                                              double thetamaxsqd, int hmax_only)
{
  CellH hh;
  hh.hmax = 0.0; hh.pad = 0.0;
  for (int k = 0; k < 3; k++) { hh.hbmin[k] = BIG; hh.hbmax[k] = -BIG; }
  if (r1.N > 0) {
    _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) { hh.hbmin[k] = fmin(r1.hbmin[k], hh.hbmin[k]); hh.hbmax[k] = fmax(r1.hbmax[k], hh.hbmax[k]); }
    hh.hmax = fmax(hh.hmax, r1.hmax);
  }
  if (r2.N > 0) {
    _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) { hh.hbmin[k] = fmin(r2.hbmin[k], hh.hbmin[k]); hh.hbmax[k] = fmax(r2.hbmax[k], hh.hbmax[k]); }
    hh.hmax = fmax(hh.hmax, r2.hmax);
  }
  d.ch[n] = hh;
  for (int k = 0; k < 3; k++) { o.hbmin[k] = hh.hbmin[k]; o.hbmax[k] = hh.hbmax[k]; }
  o.hmax = hh.hmax;
  o.N = d.cN[n];
  if (hmax_only) { d.cgeo[n].hmax = hh.hmax; return; }
  CellBox b; CellGeo c; CellCom cm;
  cm.m = 0.0; c.rmax = 0.0; c.cdistsqd = BIG; c.mac = 0.0; b.pad = 0.0;
  for (int k = 0; k < 3; k++) { cm.com[k] = 0.0; c.rcell[k] = 0.0; b.bbmin[k] = BIG; b.bbmax[k] = -BIG; }
  c.first = d.cfirst[n]; c.N = o.N; b.first = c.first; b.N = c.N;
  if (r1.N > 0) _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) { b.bbmin[k] = fmin(r1.bbmin[k], b.bbmin[k]); b.bbmax[k] = fmax(r1.bbmax[k], b.bbmax[k]); }
  if (r2.N > 0) _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) { b.bbmin[k] = fmin(r2.bbmin[k], b.bbmin[k]); b.bbmax[k] = fmax(r2.bbmax[k], b.bbmax[k]); }
  cm.m = r1.m + r2.m;
  if (cm.m > 0) _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) cm.com[k] = (r1.m*r1.com[k] + r2.m*r2.com[k])/cm.m;
  if (c.N > 0) finish_cell(d, b, c, hh.hmax, thetamaxsqd);
  c.hmax = hh.hmax;
  d.cbox[n] = b; d.cgeo[n] = c; d.ccom[n] = cm;
  for (int k = 0; k < 3; k++) { o.bbmin[k] = b.bbmin[k]; o.bbmax[k] = b.bbmax[k]; o.com[k] = cm.com[k]; }
  o.m = cm.m;
  if (d.cquad) {                                     // KDTree.cpp:1004-1052: children's moments shifted to the parent's COM
    CellQuad cq;
    for (int k = 0; k < 5; k++) cq.q[k] = 0.0;
    for (int k = 0; k < 3; k++) cq.pad[k] = 0.0;
    const int nq = d.ndim == 3 ? 5 : (d.ndim == 2 ? 3 : 1);
    if (r1.m > 0) {
      double dr[3] = {0.0, 0.0, 0.0};
      _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) dr[k] = r1.com[k] - cm.com[k];
      _Pragma("unroll") for (int k = 0; k < 5; k++) if (k < nq) cq.q[k] += r1.q[k];
      add_quad(cq.q, r1.m, dr, d.ndim);
    }
    if (r2.m > 0) {
      double dr[3] = {0.0, 0.0, 0.0};
      _Pragma("unroll") for (int k = 0; k < 3; k++) if (k < d.ndim) dr[k] = r2.com[k] - cm.com[k];
      _Pragma("unroll") for (int k = 0; k < 5; k++) if (k < nq) cq.q[k] += r2.q[k];
      add_quad(cq.q, r2.m, dr, d.ndim);
    }
    d.cquad[n] = cq;
    for (int k = 0; k < 5; k++) o.q[k] = cq.q[k];
    if (d.mac_stock == GH_MAC_EIGENMAC) d.cgeo[n].mac = eigen_mac(cq.q, d.ndim, d.macerror);
  }
}

__global__ void k_stock_level(DevicePtrs d, int level, double thetamaxsqd, int hmax_only, int j0, int nj)
{
  // cells [j0, j0 + nj) of the level: all of it on one rank, the rank's own subtree otherwise
  const int jl = blockIdx.x*blockDim.x + threadIdx.x;
  if (jl >= nj) return;
  const int j = j0 + jl;
  const int n = (1 << level) - 1 + j;
  SRec r1, r2, o;
  srec_load(d, 2*n + 1, hmax_only, r1);
  srec_load(d, 2*n + 2, hmax_only, r2);
  stock_combine(d, n, r1, r2, o, thetamaxsqd, hmax_only);
}

// Leaves plus the nlev levels above them in one launch: a workgroup owns the 2^nlev leaves below one cell of
// level ltot-nlev and passes child records between levels through LDS (a barrier per level instead of a launch).
#define GH_STOCK_NLEV 7
__global__ __launch_bounds__(1 << GH_STOCK_NLEV) void k_stock_bottom(DevicePtrs d, double kernrange, double thetamaxsqd,
                                                                      int hmax_only, int nlev, int boff)
{
  __shared__ SRec s_a[1 << GH_STOCK_NLEV];
  __shared__ SRec s_b[1 << (GH_STOCK_NLEV - 1)];
  const int t = threadIdx.x;
  const int bx = boff + blockIdx.x;               // boff: first workgroup of this rank's subtree
  if (t < (1 << nlev)) {
    const int g = (bx << nlev) + t;
    const int n = d.gtot - 1 + g;
    SRec r;
    stock_leaf(d, n, kernrange, thetamaxsqd, hmax_only, r);
    s_a[t] = r;
  }
  SRec *src = s_a, *dst = s_b;
  for (int s = 1; s <= nlev; s++) {
    __syncthreads();
    const int level = d.ltot - s;
    if (t < (1 << (nlev - s))) {
      const int n = (1 << level) - 1 + (bx << (nlev - s)) + t;
      SRec o;
      stock_combine(d, n, src[2*t], src[2*t + 1], o, thetamaxsqd, hmax_only);
      dst[t] = o;
    }
    SRec *tmp = src; src = dst; dst = tmp;
  }
}

// levels ltop .. 0 in one launch of one workgroup: the top of the tree is latency, not bandwidth.  Levels
// wider than 128 cells hand their results on through global memory, the rest through LDS.
// levels lbase + ltop .. lbase of the subtree below cell `sub` of level lbase (the whole tree: lbase = sub = 0)
__global__ __launch_bounds__(1024) void k_stock_top(DevicePtrs d, int ltop, double thetamaxsqd, int hmax_only, int lbase, int sub)
{
  __shared__ SRec s_a[128];
  __shared__ SRec s_b[64];
  SRec *src = s_a, *dst = s_b;
  bool in_lds = false;
  for (int level = ltop; level >= 0; level--) {
    const int nc = 1 << level;
    const bool out_lds = nc <= 128;
    SRec *o_buf = in_lds ? dst : s_a;
    for (int j = threadIdx.x; j < nc; j += blockDim.x) {
      const int n = (1 << (lbase + level)) - 1 + (sub << level) + j;
      SRec r1, r2, o;
      if (in_lds) { r1 = src[2*j]; r2 = src[2*j + 1]; }
      else { srec_load(d, 2*n + 1, hmax_only, r1); srec_load(d, 2*n + 2, hmax_only, r2); }
      stock_combine(d, n, r1, r2, o, thetamaxsqd, hmax_only);
      if (out_lds) o_buf[j] = o;
    }
    if (in_lds) { SRec *tmp = src; src = dst; dst = tmp; }
    else if (out_lds) { src = s_a; dst = s_b; }
    in_lds = out_lds;
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
DevicePtrs gh_dev(gh_ctx *ctx)
{
  DevicePtrs d;
  for (int f = 0; f < D_COUNT; f++) d.f[f] = ctx->fbuf[ctx->cur][f];
  d.iorig = ctx->iorig[ctx->cur];
  d.posm = ctx->posm; d.hrec = ctx->hrec;
  d.cbox = ctx->cbox; d.ch = ctx->ch; d.cgeo = ctx->cgeo; d.ccom = ctx->ccom; d.cquad = ctx->cquad; d.leaf_amin = ctx->leaf_amin;
  d.macerror = ctx->cfg.macerror; d.mac_stock = ctx->cfg.self_gravity ? ctx->cfg.gravity_mac : GH_MAC_GEOMETRIC;
  d.cfirst = ctx->cfirst; d.cN = ctx->cN;
  d.N = (int) ctx->N; d.ndim = ctx->ndim; d.ltot = ctx->ltot; d.gtot = ctx->gtot;
  d.lgroup = ctx->lgroup; d.ngroups = ctx->ngroups; d.leafocc = ctx->leafocc;
  d.levels = ctx->cfg.Nlevels > 1 ? 1 : 0;
  d.leafact = (d.levels && ctx->tree_stale) ? ctx->leafact : nullptr;
  d.sinks = ctx->cfg.sink_particles ? 1 : 0;
  d.pm_invhsqd = ctx->pm_invhsqd; d.pm_cullsqd = ctx->pm_cullsqd;
  return d;
}

// the same view restricted to this rank's own particles [own_first, own_first + own_count): for the elementwise kernels
// (drift, kick, timestep, packs).  Cell tables are NOT shifted - kernels that walk the tree take gh_dev().
DevicePtrs gh_dev_own(gh_ctx *ctx)
{
  DevicePtrs d = gh_dev(ctx);
  const size_t o = (size_t) ctx->own_first;
  for (int f = 0; f < D_COUNT; f++) d.f[f] += o;
  d.iorig += o; d.posm += o; d.hrec += 4*o;
  if (d.pm_invhsqd) { d.pm_invhsqd += o; d.pm_cullsqd += o; }
  d.N = (int) (ctx->own_held >= 0 ? ctx->own_held : ctx->own_count);     // (own_held: between gh_sinks_delete_dead and the migration)
  return d;
}

// tree size (KDTree::ComputeTreeSize, KDTree.cpp:322-352) and the static particle ranges of all cells
int gh_alloc_tree(gh_ctx *ctx)
{
  const int64_t N = ctx->N;
  if (ctx->tree_layout_N == N) return GH_OK;
  int ltot = 0;
  while ((int64_t) ctx->cfg.Nleafmax*(1ll << ltot) < N) ltot++;
  const int gtot = 1 << ltot, Ncell = 2*gtot - 1;
  ctx->ltot = ltot; ctx->gtot = gtot; ctx->Ncell = Ncell;
  ctx->h_cfirst.assign(Ncell, 0); ctx->h_cN.assign(Ncell, 0);
  ctx->h_cN[0] = (int) N;
  for (int n = 0; n < gtot - 1; n++) {
    const int half = ctx->h_cN[n]/2;
    ctx->h_cfirst[2*n + 1] = ctx->h_cfirst[n];        ctx->h_cN[2*n + 1] = half;
    ctx->h_cfirst[2*n + 2] = ctx->h_cfirst[n] + half; ctx->h_cN[2*n + 2] = ctx->h_cN[n] - half;
  }
  ctx->h_cleft.assign(Ncell, 0);
  for (int l = 0; l <= ltot; l++) {
    int run = 0;
    for (int j = 0; j < (1 << l); j++) { const int n = (1 << l) - 1 + j; ctx->h_cleft[n] = run; run += ctx->h_cN[n]/2; }
  }
  int occ = 0;
  for (int g = 0; g < gtot; g++) occ = std::max(occ, ctx->h_cN[gtot - 1 + g]);
  ctx->leafocc = occ;
  // a group = the subtree whose particles one wavefront handles: as many leaves as fit 64 lanes
  // ... and at most 16 (the gravity walk keeps a 16-bit leaf mask per stack entry): Nleafmax < 4 leaves lanes idle
  int G = 0;
  while (G < ltot && G < 4 && occ*(2 << G) <= GH_WAVE) G++;
  ctx->lgroup = ltot - G;
  ctx->ngroups = 1 << ctx->lgroup;
  // first level whose cells all fit the LDS-resident subtree kernel
  int lsub = 0;
  for (; lsub < ltot; lsub++) {
    int mx = 0;
    for (int j = 0; j < (1 << lsub); j++) mx = std::max(mx, ctx->h_cN[(1 << lsub) - 1 + j]);
    if (mx <= GH_SEG) break;
  }
  ctx->lsub = lsub;
  // multi-GPU: rank r owns level-L cell r of the global tree (L = log2 nranks) - a static particle range
  {
    int L = 0;
    while ((1 << L) < ctx->nranks) L++;
    if ((1 << L) != ctx->nranks || L > ctx->lgroup) return gh_fail(ctx, GH_ERR_INVALID, "the rank count must be a power of two, at most the number of particle groups");
    ctx->L = L;
    const int T = (1 << L) - 1 + ctx->rank;
    ctx->own_first = ctx->h_cfirst[T]; ctx->own_count = ctx->h_cN[T];
    if (ctx->lsub < L) ctx->lsub = L;
  }

  // buffers are kept while they are large enough: in a sink run N shrinks by a few particles whenever gas is accreted, the
  // cell count stays, and a hipFree + hipMalloc per array (each a device synchronisation) cost 1.4 ms of such a step
  auto re = [&](void **p, size_t bytes) -> hipError_t {
    size_t &have = ctx->tree_bytes[(void*) p];
    if (*p && have >= bytes) return hipSuccess;
    if (*p) (void) hipFree(*p);
    *p = nullptr; have = 0;
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) have = bytes;
    return e;
  };
  GH_CHECK(ctx, re((void**) &ctx->cfirst, sizeof(int)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->cN, sizeof(int)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->cleft, sizeof(int)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->cbox, sizeof(CellBox)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->ch, sizeof(CellH)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->cgeo, sizeof(CellGeo)*Ncell));
  GH_CHECK(ctx, re((void**) &ctx->ccom, sizeof(CellCom)*Ncell));
  if (ctx->cfg.ntreestockstep > 1) GH_CHECK(ctx, re((void**) &ctx->cvel, sizeof(double)*3*Ncell));
  if (ctx->cfg.ntreestockstep > 1 && ctx->cfg.Nlevels > 1) GH_CHECK(ctx, re((void**) &ctx->leafact, sizeof(int)*gtot));
  if (ctx->cfg.gravity_mac != GH_MAC_GEOMETRIC && ctx->cfg.self_gravity) {
    GH_CHECK(ctx, re((void**) &ctx->leaf_amin, sizeof(double)*gtot));
    GH_CHECK(ctx, hipMemsetAsync(ctx->leaf_amin, 0, sizeof(double)*gtot, ctx->stream));
  }
  if ((ctx->cfg.multipole == GH_MULTIPOLE_QUADRUPOLE || ctx->cfg.multipole == GH_MULTIPOLE_FAST_QUADRUPOLE || ctx->cfg.gravity_mac == GH_MAC_EIGENMAC) && ctx->cfg.self_gravity) {   // KDTree.cpp:823-824
    GH_CHECK(ctx, re((void**) &ctx->cquad, sizeof(CellQuad)*Ncell));
    GH_CHECK(ctx, hipMemsetAsync(ctx->cquad, 0, sizeof(CellQuad)*Ncell, ctx->stream));
  }
  GH_CHECK(ctx, re((void**) &ctx->dbbmin, sizeof(double)*3*(Ncell + 2)));
  GH_CHECK(ctx, re((void**) &ctx->dbbmax, sizeof(double)*3*(Ncell + 2)));
  GH_CHECK(ctx, re((void**) &ctx->kdiv, sizeof(int)*Ncell));
  GH_CHECK(ctx, hipMemcpyAsync(ctx->cfirst, ctx->h_cfirst.data(), sizeof(int)*Ncell, hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipMemcpyAsync(ctx->cN, ctx->h_cN.data(), sizeof(int)*Ncell, hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipMemcpyAsync(ctx->cleft, ctx->h_cleft.data(), sizeof(int)*Ncell, hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipMemsetAsync(ctx->cbox, 0, sizeof(CellBox)*Ncell, ctx->stream));
  GH_CHECK(ctx, hipMemsetAsync(ctx->ch, 0, sizeof(CellH)*Ncell, ctx->stream));
  GH_CHECK(ctx, hipMemsetAsync(ctx->cgeo, 0, sizeof(CellGeo)*Ncell, ctx->stream));
  GH_CHECK(ctx, hipMemsetAsync(ctx->ccom, 0, sizeof(CellCom)*Ncell, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ctx->tree_layout_N = N;
  return GH_OK;
}

int gh_pack_posm(gh_ctx *ctx)
{
  DevicePtrs d = gh_dev_own(ctx);
  hipLaunchKernelGGL(k_pack_posm, dim3(cdiv(d.N, 256)), dim3(256), 0, ctx->stream, d);
  return GH_OK;
}

int gh_dd_publish(gh_ctx *ctx, int hmax_only);      // comm.hip: all-gather of the top of every rank's subtree

// k_stock_top for the levels above the ranks' cells (comm.hip calls it after the all-gather)
void gh_stock_top_levels(gh_ctx *ctx, int ltop, int hmax_only)
{
  if (ltop >= 0)
    hipLaunchKernelGGL(k_stock_top, dim3(1), dim3(1024), 0, ctx->stream, gh_dev(ctx), ltop, ctx->cfg.thetamaxsqd, hmax_only, 0, 0);
}

static int stock_tree(gh_ctx *ctx, int hmax_only)
{
  DevicePtrs d = gh_dev(ctx);
  const double kr = KERNRANGE_OF(ctx->cfg);
  const int L = ctx->L;                            // 0 on one rank: the subtree is the tree
  const int nlev = std::min(ctx->ltot - L, GH_STOCK_NLEV);
  const int nblk = (ctx->gtot >> L) >> nlev;
  hipLaunchKernelGGL(k_stock_bottom, dim3(nblk), dim3(1 << GH_STOCK_NLEV), 0, ctx->stream, d, kr,
                     ctx->cfg.thetamaxsqd, hmax_only, nlev, ctx->rank*nblk);
  const int lnext = ctx->ltot - nlev - 1;        // highest level not stocked yet
  if (ctx->nranks > 1) {
    const int ltoprel = std::min(lnext - L, 9);    // the top levels of the rank's own subtree in one launch
    for (int l = lnext; l > L + ltoprel; l--)
      hipLaunchKernelGGL(k_stock_level, dim3(cdiv(1 << (l - L), 256)), dim3(256), 0, ctx->stream, d, l,
                         ctx->cfg.thetamaxsqd, hmax_only, ctx->rank << (l - L), 1 << (l - L));
    if (ltoprel >= 0)
      hipLaunchKernelGGL(k_stock_top, dim3(1), dim3(1024), 0, ctx->stream, d, ltoprel, ctx->cfg.thetamaxsqd, hmax_only, L, ctx->rank);
    return gh_dd_publish(ctx, hmax_only);          // remote subtree tops, then the shared levels above the ranks' cells
  }
  const int ltop = std::min(lnext, 9);
  for (int l = lnext; l > ltop; l--)
    hipLaunchKernelGGL(k_stock_level, dim3(cdiv(1 << l, 256)), dim3(256), 0, ctx->stream, d, l,
                       ctx->cfg.thetamaxsqd, hmax_only, 0, 1 << l);
  gh_stock_top_levels(ctx, ltop, hmax_only);
  return GH_OK;
}

int gh_update_hmax_impl(gh_ctx *ctx) { return stock_tree(ctx, 1); }

void stock_cell_velocities_fwd(gh_ctx *ctx);
int gh_dd_decompose(gh_ctx *ctx);                    // comm.hip: global root box, top-level splits, particle migration

// local extent of r -/+ kernrange*h over this rank's particles -> ctx->dbbmin/dbbmax[0..2]; every own particle starts in `node0`
void gh_rootbox_local(gh_ctx *ctx, int node0)
{
  const int nblk = 256;
  hipLaunchKernelGGL(k_rootbox_partial, dim3(nblk), dim3(GH_RB_THREADS), 0, ctx->stream, gh_dev_own(ctx), KERNRANGE_OF(ctx->cfg), ctx->redbuf,
                     ctx->cellnode[0] + ctx->own_first, node0);
  hipLaunchKernelGGL(k_rootbox_final, dim3(1), dim3(64), 0, ctx->stream, ctx->redbuf, nblk, ctx->dbbmin, ctx->dbbmax);
}

// 32-bit sort keys: the coordinate's position in the build's box, truncated - monotone (not strictly) in x
__global__ void k_sortkeys(DevicePtrs d, const double *bmin, const double *bmax, unsigned int *keys, int p0, int pn, int N)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= pn) return;
  const double lo = bmin[k], ext = bmax[k] - lo;
  const double scale = ext > 0.0 ? 4294967295.0/ext : 0.0;
  double u = (d.f[D_RX + k][p0 + i] - lo)*scale;
  u = u < 0.0 ? 0.0 : (u > 4294967295.0 ? 4294967295.0 : u);
  keys[(size_t) k*N + i] = (unsigned int) u;
}
// runs of equal 32-bit keys -> (coordinate, position) order: the first element of every run sorts it (insertion sort;
// runs are one or two elements long unless coordinates coincide, and then they are already in order)
__global__ void k_sortfix(const double *x, const unsigned int *keys, int *ids, int n)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= n || i + 1 >= n) return;
  const unsigned int u = keys[i];
  if ((i > 0 && keys[i - 1] == u) || keys[i + 1] != u) return;
  int e = i + 2;
  while (e < n && keys[e] == u) e++;
  for (int j = i + 1; j < e; j++) {
    const int idj = ids[j];
    const double xj = x[idj];
    int q = j - 1;
    while (q >= i) {
      const int idq = ids[q];
      const double xq = x[idq];
      if (xq < xj || (xq == xj && idq < idj)) break;
      ids[q + 1] = idq;
      q--;
    }
    ids[q + 1] = idj;
  }
}
typedef rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0> SortCfg;

int gh_tree_build_impl(gh_ctx *ctx)
{
  int rc = gh_alloc_tree(ctx);
  if (rc) return rc;
  const int L = ctx->L;
  if (ctx->nranks > 1) {
    // root box, the L shared top levels (distributed median splits) and the migration of the particles that changed
    // cells; leaves every own particle in cell (1 << L) - 1 + rank with that cell's inherited box in dbbmin/dbbmax
    const bool uneven = ctx->own_held >= 0;               // a sink run after accretion: this rank's range may have grown
    if ((rc = gh_dd_decompose(ctx))) return rc;
    if (uneven)                                            // ... beyond the particles gh_rootbox_local put into the rank's cell
      hipLaunchKernelGGL(k_fill_int, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, ctx->cellnode[0] + ctx->own_first, (int) ctx->own_count, (1 << L) - 1 + ctx->rank);
  }
  else gh_rootbox_local(ctx, 0);
  const int p0 = (int) ctx->own_first, pn = (int) ctx->own_count;
  const int N = (int) ctx->N;
  const int nb = cdiv(pn, 256);
  DevicePtrs d = gh_dev(ctx);
  hipStream_t s = ctx->stream;
  if (getenv("GH_DD_DEBUG")) {
    // this rank's particles after the migration: ids and coordinates must be distinct
    GH_CHECK(ctx, hipStreamSynchronize(s));
    std::vector<int> ids((size_t) pn); std::vector<double> x((size_t) pn), m((size_t) pn);
    GH_CHECK(ctx, hipMemcpy(ids.data(), ctx->iorig[ctx->cur] + p0, sizeof(int)*(size_t) pn, hipMemcpyDeviceToHost));
    GH_CHECK(ctx, hipMemcpy(x.data(), ctx->fbuf[ctx->cur][D_RX] + p0, sizeof(double)*(size_t) pn, hipMemcpyDeviceToHost));
    GH_CHECK(ctx, hipMemcpy(m.data(), ctx->fbuf[ctx->cur][D_M] + p0, sizeof(double)*(size_t) pn, hipMemcpyDeviceToHost));
    std::vector<int> si(ids); std::sort(si.begin(), si.end());
    std::vector<double> sx(x); std::sort(sx.begin(), sx.end());
    int dupi = 0, dupx = 0, zm = 0, bad = 0;
    for (int i = 1; i < pn; i++) { if (si[i] == si[i - 1]) dupi++; if (sx[i] == sx[i - 1]) dupx++; }
    for (int i = 0; i < pn; i++) { if (m[i] == 0.0) zm++; if (ids[i] < 0 || ids[i] >= N) bad++; }
    fprintf(stderr, "[build] rank %d N %d p0 %d pn %d: duplicate ids %d, equal x %d, zero masses %d, ids out of range %d\n", ctx->rank, N, p0, pn, dupi, dupx, zm, bad);
  }

  // One argsort per axis, on three streams.  The keys are 32-bit fixed-point coordinates relative to the build's box
  // (monotone in x), sorted with rocPRIM's onesweep radix sort (4 passes; its merge sort, which it would pick for <= 2^20
  // fp64 keys, needs 10 merge passes of two launches each); particles whose 32-bit keys collide are then put into
  // (x, position) order by k_sortfix - the result is the stable fp64 sort's, element for element.
  if (ctx->iota_N != N || ctx->iota_p0 != p0) {             // identity values for the argsorts: never modified
    hipLaunchKernelGGL(k_iota, dim3(nb), dim3(256), 0, s, ctx->sortvals, pn, p0);
    ctx->iota_N = N; ctx->iota_p0 = p0;
  }
  {
    unsigned int *kin = (unsigned int*) ctx->sortkeys_out, *kout = kin + (size_t) 3*N;      // 2 x 3N x 4 B = the 24N bytes allocated
    const int node0 = (1 << L) - 1 + ctx->rank;
    hipLaunchKernelGGL(k_sortkeys, dim3(nb, ctx->ndim), dim3(256), 0, s, d, ctx->dbbmin + 3*node0, ctx->dbbmax + 3*node0, kin, p0, pn, N);
    size_t need = 0;
    GH_CHECK(ctx, rocprim::radix_sort_pairs<SortCfg>(nullptr, need, kin, kout, ctx->sortvals, ctx->P[0][0] + p0, (size_t) pn, 0, 32, s));
    need = (need + 255) & ~(size_t) 255;
    if (3*need > ctx->sorttemp_bytes) {
      GH_CHECK(ctx, hipStreamSynchronize(s));
      if (ctx->sorttemp) (void) hipFree(ctx->sorttemp);
      GH_CHECK(ctx, hipMalloc(&ctx->sorttemp, 3*need));
      ctx->sorttemp_bytes = 3*need;
    }
    GH_CHECK(ctx, hipEventRecord(ctx->ev_fork, s));
    for (int k = 0; k < ctx->ndim; k++) {
      hipStream_t sk = k == 0 ? s : ctx->aux[k - 1];
      if (k > 0) GH_CHECK(ctx, hipStreamWaitEvent(sk, ctx->ev_fork, 0));
      GH_CHECK(ctx, rocprim::radix_sort_pairs<SortCfg>((char*) ctx->sorttemp + k*need, need, kin + (size_t) k*N, kout + (size_t) k*N, ctx->sortvals,
                                                       ctx->P[0][k] + p0, (size_t) pn, 0, 32, sk));
      hipLaunchKernelGGL(k_sortfix, dim3(nb), dim3(256), 0, sk, d.f[D_RX + k], kout + (size_t) k*N, ctx->P[0][k] + p0, pn);
      if (k > 0) GH_CHECK(ctx, hipEventRecord(ctx->ev_join[k - 1], sk));
    }
    for (int k = 1; k < ctx->ndim; k++) GH_CHECK(ctx, hipStreamWaitEvent(s, ctx->ev_join[k - 1], 0));
  }

  // d_blk[13]: this build split equal coordinates; d_blk[14]: ... some build since the last look did (sticky)
  hipLaunchKernelGGL(k_tie_reset, dim3(1), dim3(1), 0, s, ctx->d_blk + 13);
  int pb = 0;
  const int nwords = (pn + 63)/64;
  auto level_args = [&](int l) {
    LevelArgs a;
    for (int k = 0; k < 3; k++) { a.P[k] = ctx->P[pb][k]; a.Pn[k] = ctx->P[pb ^ 1][k]; a.W[k] = ctx->W[k]; a.Wpre[k] = ctx->Wpre[k]; }
    a.cellnode = ctx->cellnode[pb]; a.cellnode_next = ctx->cellnode[pb ^ 1];
    a.side = ctx->side; a.dbbmin = ctx->dbbmin; a.dbbmax = ctx->dbbmax; a.kdiv = ctx->kdiv;
    a.tie = ctx->d_blk + 13;
    a.level = l; a.nwords = nwords;
    a.p0 = p0; a.pn = pn; a.jbase = 0;
    a.cleft = ctx->cleft; a.cleft0 = 0;
    return a;
  };
  const int lsub = ctx->lsub;
  for (int l = L; l < lsub; l++) {
    LevelArgs a = level_args(l);
    a.cleft0 = ctx->h_cleft[(1 << l) - 1 + (ctx->rank << (l - L))];
    hipLaunchKernelGGL(k_level_flags, dim3(cdiv(pn, GH_LB)), dim3(GH_LB), 0, s, d, a);
    hipLaunchKernelGGL(k_level_scatter, dim3(cdiv(pn, GH_LB)), dim3(GH_LB), 0, s, d, a);
    pb ^= 1;
  }
  {
    // remaining levels: one workgroup per level-lsub cell (cellnode[pb^1] is free: used as id -> local index map)
    LevelArgs a = level_args(lsub);
    a.jbase = ctx->rank << (lsub - L);
    hipLaunchKernelGGL(k_build_subtree, dim3(1 << (lsub - L)), dim3(1024), 0, s, d, a, lsub, ctx->ltot, ctx->P[pb ^ 1][0],
                       ctx->cellnode[pb ^ 1]);
    pb ^= 1;
  }

  // gather every particle array into tree order (perm[new] = old position).  The two pointer tables
  // (buffer 0 -> 1 and 1 -> 0) live in device memory since allocation: no host synchronisation here.
  // equal coordinates at a median: the reference's own quick-select order decides (exact mode, armed once a tie was seen)
  // (sink runs: always - the potential-minimum flag and the accretion order depend on the reference's order of the
  // particles inside a leaf cell, which only its own quick-select produces)
  // ... that is: once a particle is within a factor two of the sink density, or a sink or star exists (ctx->sink_exact,
  // decided in gh_sync_collect from the count the potential-minimum stage leaves).  Before that - the whole early collapse of
  // a Boss-Bodenheimer run, where this build was 19 of the 29 ms of a 2 000 000-particle step - nothing reads that order, and
  // the run takes the fast build like any other (same cells; inside a leaf the coordinate order, i.e. sums in another order).
  if (ctx->cfg.sink_particles && ctx->nranks == 1 && ctx->sink_exact) { if ((rc = exact_build_gated(ctx, ctx->P[pb][0], nullptr))) return rc; }
  else if (ctx->exact_armed && ctx->nranks == 1) { if ((rc = exact_build_gated(ctx, ctx->P[pb][0], ctx->d_blk + 13))) return rc; }
  const int *perm = ctx->P[pb][0];
  // Inside a global-timestep gh_step every particle gets new density-pass outputs (rho, invomega, zeta, hfactor, hrangesqd,
  // sound, pressure, div_v) and new accelerations (a, atree, gpot, gpot_hydro, dudt: zeroed, then summed) before anything
  // reads them: those 17 of the 42 arrays need not be carried into the new tree order.  Not with block timesteps (inactive
  // particles keep theirs), sinks, the time-dependent viscosities (cd2010 gathers the neighbours' a), relative MACs (the
  // stocking reads atree / gpot), nor for builds the caller asks for itself (gh_build_tree: it may look at anything).
  unsigned long long live = ~0ull;
  if (ctx->in_step && ctx->cfg.Nlevels <= 1 && !ctx->cfg.sink_particles && (ctx->cfg.avisc == GH_AVISC_NONE || ctx->cfg.avisc == GH_AVISC_MON97) &&
      (!ctx->cfg.self_gravity || ctx->cfg.gravity_mac == GH_MAC_GEOMETRIC) && ctx->nstars == 0) {
    for (int f : {D_RHO, D_INVOMEGA, D_ZETA, D_HFACTOR, D_HRANGESQD, D_SOUND, D_PRESSURE, D_DIV_V, D_AX, D_AY, D_AZ, D_ATX, D_ATY, D_ATZ,
                  D_GPOT, D_GPOT_HYDRO, D_DUDT}) live &= ~(1ull << f);
  }
  hipLaunchKernelGGL(k_permute, dim3(std::min(nb, 2048), (ctx->cfg.Nlevels > 1 || ctx->cfg.sink_particles) ? D_COUNT : D_COUNT_BASE), dim3(256), 0, s, ctx->d_ptrtab + (size_t) ctx->cur*2*D_COUNT, perm, p0, pn, live);
  hipLaunchKernelGGL(k_permute_int, dim3(nb), dim3(256), 0, s, ctx->iorig[ctx->cur], ctx->iorig[ctx->cur ^ 1], perm, p0, pn);
  ctx->cur ^= 1;
  ctx->tree_valid = true;
  ctx->tree_stale = false;
  gh_pack_posm(ctx);
  if ((rc = stock_tree(ctx, 0))) return rc;
  stock_cell_velocities_fwd(ctx);
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

// ---- cell mean velocities + Tree::ExtrapolateCellProperties (Tree.cpp:172-198), ntreestockstep > 1 only ----------------
// leaf: v = sum m_i v_i / m (KDTree.cpp stocking of leaf cells); parent: (m1 v1 + m2 v2)/m
__global__ void k_cellv_leaf(DevicePtrs d, double *cvel)
{
  const int l = blockIdx.x*blockDim.x + threadIdx.x;
  if (l >= d.gtot) return;
  const int n = d.gtot - 1 + l;
  const int first = d.cfirst[n], cn = d.cN[n];
  double m = 0.0, v[3] = {0.0, 0.0, 0.0};
  for (int t = 0; t < cn; t++) {
    const double mi = d.f[D_M][first + t];
    m += mi;
    for (int k = 0; k < d.ndim; k++) v[k] += mi*d.f[D_VX + k][first + t];
  }
  for (int k = 0; k < 3; k++) cvel[(size_t) 3*n + k] = m > 0.0 ? v[k]/m : 0.0;
}
__global__ void k_cellv_level(DevicePtrs d, double *cvel, int level)
{
  const int j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= (1 << level)) return;
  const int n = (1 << level) - 1 + j, c1 = 2*n + 1, c2 = 2*n + 2;
  const double m1 = d.ccom[c1].m, m2 = d.ccom[c2].m, m = d.ccom[n].m;
  for (int k = 0; k < 3; k++)
    cvel[(size_t) 3*n + k] = m > 0.0 ? (m1*cvel[(size_t) 3*c1 + k] + m2*cvel[(size_t) 3*c2 + k])/m : 0.0;
}
__global__ void k_extrapolate_cells(DevicePtrs d, const double *cvel, const double *time, int Ncell)
{
  const int n = blockIdx.x*blockDim.x + threadIdx.x;
  if (n >= Ncell) return;
  const double dt = time[1];
  for (int k = 0; k < d.ndim; k++) {
    const double dx = cvel[(size_t) 3*n + k]*dt;
    d.cbox[n].bbmin[k] += dx; d.cbox[n].bbmax[k] += dx;
    d.ch[n].hbmin[k] += dx; d.ch[n].hbmax[k] += dx;
    d.cgeo[n].rcell[k] += dx;
    d.ccom[n].com[k] += dx;
  }
}
// cell.Nactive of the leaf cells (StockCellProperties KDTree.cpp:856 / UpdateActiveParticleCounters :1217-1254).  The
// reference refreshes it when it builds or stocks the tree and at the top of a repeated pass of MainLoop's
// do ... while (activecount > 0) - NOT on a step that only extrapolates the tree (SphSimulation.cpp:663), where the
// active cells are therefore those of the last stocking; restated as it is.
__global__ void k_leaf_nactive(DevicePtrs d, int *leafact)
{
  const int l = blockIdx.x*blockDim.x + threadIdx.x;
  if (l >= d.gtot) return;
  const int n = d.gtot - 1 + l;
  const int first = d.cfirst[n], cn = d.cN[n];
  int na = 0;
  for (int t = 0; t < cn; t++) na += (int) d.f[D_FLAGS][first + t] & 1;
  leafact[l] = na;
}
int gh_leaf_active_counters(gh_ctx *ctx)
{
  if (!ctx->leafact) return GH_OK;
  hipLaunchKernelGGL(k_leaf_nactive, dim3(cdiv(ctx->gtot, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), ctx->leafact);
  return GH_OK;
}

void stock_cell_velocities_fwd(gh_ctx *ctx) ;
static void stock_cell_velocities(gh_ctx *ctx)
{
  gh_leaf_active_counters(ctx);
  if (ctx->cfg.ntreestockstep <= 1 || !ctx->cvel) return;
  DevicePtrs d = gh_dev(ctx);
  hipLaunchKernelGGL(k_cellv_leaf, dim3(cdiv(ctx->gtot, 256)), dim3(256), 0, ctx->stream, d, ctx->cvel);
  for (int l = ctx->ltot - 1; l >= 0; l--)
    hipLaunchKernelGGL(k_cellv_level, dim3(cdiv(1 << l, 256)), dim3(256), 0, ctx->stream, d, ctx->cvel, l);
}
void stock_cell_velocities_fwd(gh_ctx *ctx) { stock_cell_velocities(ctx); }
extern double *gh_time_dev(gh_ctx *ctx);
int gh_tree_extrapolate_impl(gh_ctx *ctx)
{
  if (!ctx->tree_valid || !ctx->cvel) return gh_tree_build_impl(ctx);
  hipLaunchKernelGGL(k_extrapolate_cells, dim3(cdiv(ctx->Ncell, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), ctx->cvel, gh_time_dev(ctx), ctx->Ncell);
  ctx->tree_stale = true;
  gh_pack_posm(ctx);
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}

int gh_tree_restock_impl(gh_ctx *ctx)
{
  if (!ctx->tree_valid && ctx->tree_layout_N != ctx->N) return gh_tree_build_impl(ctx);
  stock_tree(ctx, 0);
  stock_cell_velocities(ctx);
  gh_pack_posm(ctx);
  GH_CHECK(ctx, hipGetLastError());
  ctx->tree_valid = true;
  ctx->tree_stale = false;
  return GH_OK;
}
