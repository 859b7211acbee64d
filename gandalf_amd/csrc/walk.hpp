// walk.hpp -- wave-synchronous breadth-first walks of the heap-ordered KD-tree.
//
// One wavefront (= one 64-thread workgroup) owns one "group" of up to 64 target particles (a small
// subtree).  The walk keeps a frontier of tree nodes in LDS; every lane tests one frontier node per
// pass against the group's search volume, hits are compacted with ballot + popcount.  This replaces
// the reference's per-leaf-cell stackless pointer-chasing walks (copen/cnext, Tree.cpp:291-381,
// 562-617) by something the 64-wide hardware can execute without divergence.
#pragma once
#include "gh_internal.hpp"

#define GH_NODE_BITS 26
#define GH_NODE_MASK ((1 << GH_NODE_BITS) - 1)

struct Domain {              // reference DomainBox (DomainBox.h): periodic flags and sizes, mirror walls
  int periodic[3];
  int mirror[3][2];          // mirror wall at the lhs / rhs face
  double bmin[3], bmax[3], size[3], half[3];
};

__device__ __forceinline__ unsigned long long lanemask_lt()
{
  const unsigned lane = threadIdx.x & 63;
  return lane ? ((~0ull) >> (64 - lane)) : 0ull;
}

__device__ __forceinline__ double wave_max(double v)
{
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ double wave_min(double v)
{
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// periodic image code c in [0,27): per dimension 0 = no shift, 1 = +L, 2 = -L
// Image transform of code c (one base-3 digit per dimension): x' = sg*x + sh.  Digit 1 = the image seen beyond the
// rhs face (periodic: x + L; mirror wall: 2*bmax - x), digit 2 = beyond the lhs face (x - L; 2*bmin - x).
// Velocities of mirror images change sign in that dimension (Hydrodynamics.cpp:232-246, Particle.h:601-607).
__device__ __forceinline__ void code_xform(const Domain &dom, int code, double sg[3], double sh[3])
{
  const int c[3] = {code % 3, (code/3) % 3, code/9};
  for (int k = 0; k < 3; k++) {
    sg[k] = 1.0; sh[k] = 0.0;
    if (c[k] == 1) { if (dom.periodic[k]) sh[k] = dom.size[k]; else { sg[k] = -1.0; sh[k] = 2.0*dom.bmax[k]; } }
    else if (c[k] == 2) { if (dom.periodic[k]) sh[k] = -dom.size[k]; else { sg[k] = -1.0; sh[k] = 2.0*dom.bmin[k]; } }
  }
}
// image of the interval [bmin, bmax]
__device__ __forceinline__ void image_interval(double sg, double sh, double bmin, double bmax, double &omin, double &omax)
{
  if (sg > 0.0) { omin = bmin + sh; omax = bmax + sh; }
  else { omin = sh - bmax; omax = sh - bmin; }
}

// ------------------------------------------------------------------------------------------------
// Streaming depth-first walk.
//
// The wave keeps a LIFO stack of tree nodes in LDS.  Each step pops up to 64 nodes (one per lane),
// classifies them, pushes the children of the nodes that have to be opened and emits "ranges" -
// contiguous runs of particles in tree order (a leaf, or a whole subtree that lies inside the search
// volume) - into a small ring in LDS.  As soon as the ring holds 64 particle slots, they are mapped onto
// the 64 lanes (prefix sum over the range lengths + binary search) and handed to the kernel's tile
// function.  Nothing grows with the size of the search volume: the stack is bounded by ~128 entries per
// tree level, the ring by its drain threshold.  (A breadth-first frontier or a stored candidate list
// overflows for the outlier groups of a Plummer halo, whose search volume covers the whole system.)
// ------------------------------------------------------------------------------------------------
#define GH_SCAP 768          /* stack capacity; pops narrow to GH_NARROW lanes near the top, see pop_width */
#define GH_NARROW 4
#define GH_RBCAP 160         /* range ring capacity */

// A 64-wide pop can grow the stack by 64 entries per step; a GH_NARROW-wide pop is nearly a true
// depth-first descent that needs at most GH_NARROW extra entries per tree level (<= 4*32).  Switching to
// narrow pops once fewer than 192 slots are free therefore bounds the stack for any search volume.
__device__ __forceinline__ int pop_width(int top)
{
  const int w = top > GH_SCAP - 192 ? GH_NARROW : 64;
  return top < w ? top : w;
}

struct RangeRing { int first[GH_RBCAP], cnt[GH_RBCAP], tag[GH_RBCAP]; };

template <typename StackT> struct WalkLDS {
  StackT stack[GH_SCAP];
  int rb_first[GH_RBCAP], rb_cnt[GH_RBCAP], rb_tag[GH_RBCAP];
  int pre[64];
};

__device__ __forceinline__ int wave_sum_i(int v)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// State of the range ring between drains
struct RangeState { int nrb, nslots; };

// Hand complete tiles (all of them if `final`) to tile(valid, j, tag): lane's slot holds particle j.
template <class Tile>
__device__ void range_drain_raw(int *rb_first, int *rb_cnt, int *rb_tag, int *pre, RangeState &R, bool final, Tile tile);

template <typename StackT, class Tile>
__device__ void range_drain(WalkLDS<StackT> &L, RangeState &R, bool final, Tile tile)
{
  range_drain_raw(L.rb_first, L.rb_cnt, L.rb_tag, L.pre, R, final, tile);
}

template <class Tile>
__device__ void range_drain_raw(int *rb_first, int *rb_cnt, int *rb_tag, int *pre, RangeState &R, bool final, Tile tile)
{
  struct { int *rb_first, *rb_cnt, *rb_tag, *pre; } L = {rb_first, rb_cnt, rb_tag, pre};
  const int lane = threadIdx.x & 63;
  __syncthreads();
  int pos = 0, used = 0;
  while (R.nslots >= 64 || (final && R.nslots > 0)) {
    int c = (pos + lane < R.nrb) ? L.rb_cnt[pos + lane] : 0;
    if (lane == 0) c -= used;
    int inc = c;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
    L.pre[lane] = inc;
    __syncthreads();
    const int tot = __shfl(inc, 63, 64);
    const int take = tot < 64 ? tot : 64;
    int lo = 0, hi = 64;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.pre[mid] <= lane) lo = mid + 1; else hi = mid; }
    const int idx = lo < 63 ? lo : 63;
    const bool valid = lane < take;
    const int excl = idx > 0 ? L.pre[idx - 1] : 0;
    const int off = lane - excl + (idx == 0 ? used : 0);
    const int j = valid ? L.rb_first[pos + idx] + off : 0;
    const int tag = valid ? L.rb_tag[pos + idx] : 0;
    tile(valid, j, tag);
    const int full = __popcll(__ballot((pos + lane < R.nrb) && inc <= take));
    int nused = 0;
    if (pos + full < R.nrb && full < 64) {
      const int before = full > 0 ? L.pre[full - 1] : 0;
      nused = take - before + (full == 0 ? used : 0);
    }
    pos += full; used = nused; R.nslots -= take;
    __syncthreads();
  }
  // move what is left to the front of the ring
  const int rem = R.nrb - pos;
  if (pos > 0 && rem > 0) {
    for (int base = 0; base < rem; base += 64) {
      const int k = base + lane;
      int a = 0, b = 0, c = 0;
      if (k < rem) { a = L.rb_first[pos + k]; b = L.rb_cnt[pos + k]; c = L.rb_tag[pos + k]; }
      __syncthreads();
      if (k < rem) { L.rb_first[k] = a; L.rb_cnt[k] = b; L.rb_tag[k] = c; }
      __syncthreads();
    }
  }
  if (rem > 0 && used > 0 && lane == 0) { L.rb_first[0] += used; L.rb_cnt[0] -= used; }
  R.nrb = rem;
  __syncthreads();
}

// Depth-first walk over int stack entries (node | image code << 26).
// cls(node, code, open, emit, first, cnt): classification of one popped node.
template <class Classify, class Tile>
__device__ void walk_dfs_stream(const DevicePtrs &d, WalkLDS<int> &L, unsigned int codes, Classify cls, Tile tile, int *flags)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lanemask_lt();
  int top = 0;
  for (int c = 0; c < 27; c++) {
    if (codes & (1u << c)) { if (lane == 0) L.stack[top] = 0 | (c << GH_NODE_BITS); top++; }
  }
  RangeState R; R.nrb = 0; R.nslots = 0;
  __syncthreads();
  while (top > 0) {
    const int p = pop_width(top);
    const int newtop = top - p;
    bool open = false, emit = false;
    int n = 0, code = 0, first = 0, cnt = 0;
    if (lane < p) {
      const int e = L.stack[top - 1 - lane];
      n = e & GH_NODE_MASK; code = e >> GH_NODE_BITS;
      cls(n, code, open, emit, first, cnt);
    }
    const unsigned long long om = __ballot(open), em = __ballot(emit);
    __syncthreads();
    if (open) {
      const int pos = newtop + 2*__popcll(om & lt);
      if (pos + 1 < GH_SCAP) {
        L.stack[pos] = (2*n + 1) | (code << GH_NODE_BITS);
        L.stack[pos + 1] = (2*n + 2) | (code << GH_NODE_BITS);
      }
    }
    top = newtop + 2*__popcll(om);
    if (top > GH_SCAP) { if (lane == 0) atomicOr(flags, FLAG_FRONTIER_OVERFLOW); top = GH_SCAP; }
    if (emit) {
      const int pos = R.nrb + __popcll(em & lt);
      L.rb_first[pos] = first; L.rb_cnt[pos] = cnt; L.rb_tag[pos] = code;
    }
    R.nrb += __popcll(em);
    R.nslots += wave_sum_i(emit ? cnt : 0);
    if (R.nslots >= 64 || R.nrb > GH_RBCAP - 64) range_drain(L, R, false, tile);
    else __syncthreads();
  }
  range_drain(L, R, true, tile);
}


// Depth-first walk whose stack entries carry a 16-bit mask of the group's LEAF cells that still descend through the node.
// The reference walks the tree once per leaf cell (Tree.cpp:291-381, 562-617).  With a freshly stocked tree a group-level
// walk plus distance tests finds the same neighbours; with an EXTRAPOLATED tree (Tree::ExtrapolateCellProperties: every
// cell drifted with its own mean velocity) the boxes of a parent and its children no longer nest and particles may have
// left their cell's box, so the reference loses neighbours - exactly those whose leaf, or any ancestor of it, fails the
// box test of the target leaf.  To lose the same ones every (node, leaf) decision is taken with that leaf's own box:
// cls(node, code, inmask, first, cnt) returns the mask of leaves whose test the node passes.  Ranges are emitted with
// tag = image code | leaf mask << 5.
template <class Classify, class Tile>
__device__ void walk_dfs_stream_masked(const DevicePtrs &d, WalkLDS<int> &L, unsigned short *smask, unsigned int codes,
                                       unsigned int mask0, Classify cls, Tile tile, int *flags)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lanemask_lt();
  int top = 0;
  for (int c = 0; c < 27; c++) {
    if (codes & (1u << c)) { if (lane == 0) { L.stack[top] = 0 | (c << GH_NODE_BITS); smask[top] = (unsigned short) mask0; } top++; }
  }
  RangeState R; R.nrb = 0; R.nslots = 0;
  __syncthreads();
  while (top > 0) {
    const int p = pop_width(top);
    const int newtop = top - p;
    unsigned int om = 0;
    int n = 0, code = 0, first = 0, cnt = 0;
    if (lane < p) {
      const int e = L.stack[top - 1 - lane];
      n = e & GH_NODE_MASK; code = e >> GH_NODE_BITS;
      om = cls(n, code, (unsigned int) smask[top - 1 - lane], first, cnt);
    }
    const bool open = om != 0 && n < d.gtot - 1, emit = om != 0 && n >= d.gtot - 1 && cnt > 0;
    const unsigned long long opm = __ballot(open), em = __ballot(emit);
    __syncthreads();
    if (open) {
      const int pos = newtop + 2*__popcll(opm & lt);
      if (pos + 1 < GH_SCAP) {
        L.stack[pos] = (2*n + 1) | (code << GH_NODE_BITS); smask[pos] = (unsigned short) om;
        L.stack[pos + 1] = (2*n + 2) | (code << GH_NODE_BITS); smask[pos + 1] = (unsigned short) om;
      }
    }
    top = newtop + 2*__popcll(opm);
    if (top > GH_SCAP) { if (lane == 0) atomicOr(flags, FLAG_FRONTIER_OVERFLOW); top = GH_SCAP; }
    if (emit) {
      const int pos = R.nrb + __popcll(em & lt);
      L.rb_first[pos] = first; L.rb_cnt[pos] = cnt; L.rb_tag[pos] = code | ((int) om << 5);
    }
    R.nrb += __popcll(em);
    R.nslots += wave_sum_i(emit ? cnt : 0);
    if (R.nslots >= 64 || R.nrb > GH_RBCAP - 64) range_drain(L, R, false, tile);
    else __syncthreads();
  }
  range_drain(L, R, true, tile);
}

// which periodic image codes a search box [lo,hi] needs
__device__ __forceinline__ unsigned int image_codes(const Domain &dom, int ndim, const double lo[3], const double hi[3])
{
  int opt[3][3]; int nopt[3];
  for (int k = 0; k < 3; k++) {
    nopt[k] = 0; opt[k][nopt[k]++] = 0;
    if (k < ndim) {
      if ((dom.periodic[k] || dom.mirror[k][1]) && hi[k] > dom.bmax[k]) opt[k][nopt[k]++] = 1;   // images at x+L / 2 bmax - x
      if ((dom.periodic[k] || dom.mirror[k][0]) && lo[k] < dom.bmin[k]) opt[k][nopt[k]++] = 2;   // images at x-L / 2 bmin - x
    }
  }
  unsigned int codes = 0;
  for (int a = 0; a < nopt[0]; a++)
    for (int b = 0; b < nopt[1]; b++)
      for (int c = 0; c < nopt[2]; c++) codes |= 1u << (opt[0][a] + 3*opt[1][b] + 9*opt[2][c]);
  return codes;
}

// XCD-aware block -> group map: blocks are dealt round-robin over the 8 XCDs, so give each XCD a
// contiguous (= spatially compact, L2-sharing) slice of the groups (MI355X_MICROARCH: blocks b and
// b+8 share an XCD).  Speed only, never correctness.
__device__ __forceinline__ int block_to_group(int b, int nblocks)
{
  if ((nblocks & 7) == 0) return (b & 7)*(nblocks >> 3) + (b >> 3);
  return b;
}
