// walk.hpp -- wave-synchronous breadth-first walks of the heap-ordered KD-tree.
//
// One wavefront (= one 64-thread workgroup) owns one "group" of up to 64 target particles (a small
// subtree).  The walk keeps a frontier of tree nodes in LDS; every lane tests one frontier node per
// pass against the group's search volume, hits are compacted with ballot + popcount.  This replaces
// the reference's per-leaf-cell stackless pointer-chasing walks (copen/cnext, Tree.cpp:291-381,
// 562-617) by something the 64-wide hardware can execute without divergence.
#pragma once
#include "gh_internal.hpp"

#define GH_FCAP 512          /* frontier capacity (nodes) per buffer */
#define GH_LCAP 1024         /* leaf-list capacity (leaves) */
#define GH_NODE_BITS 26
#define GH_NODE_MASK ((1 << GH_NODE_BITS) - 1)

struct Domain {              // reference DomainBox (DomainBox.h): periodic flags and sizes
  int periodic[3];
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
__device__ __forceinline__ void code_shift(const Domain &dom, int code, double s[3])
{
  const int c0 = code % 3, c1 = (code/3) % 3, c2 = code/9;
  s[0] = c0 == 0 ? 0.0 : (c0 == 1 ? dom.size[0] : -dom.size[0]);
  s[1] = c1 == 0 ? 0.0 : (c1 == 1 ? dom.size[1] : -dom.size[1]);
  s[2] = c2 == 0 ? 0.0 : (c2 == 1 ? dom.size[2] : -dom.size[2]);
}

// leaf-list entry: first particle (32 bit) | count << 32 | image code << 40
__device__ __forceinline__ unsigned long long make_leaf_entry(int first, int cnt, int code)
{
  return (unsigned long long) (unsigned) first | ((unsigned long long) cnt << 32) | ((unsigned long long) code << 40);
}

// Breadth-first walk: collects, in tree order, all non-empty leaves whose (image-shifted) cell passes
// `pred(node, shift)`; interior nodes that pass are opened.  `codes` = bit mask of the image codes to
// walk.  Returns the number of leaves (clamped to GH_LCAP; an overflow raises a flag).
template <class Pred>
__device__ int walk_collect_leaves(const DevicePtrs &d, const Domain &dom, unsigned int codes, Pred pred,
                                   int (*s_front)[GH_FCAP], unsigned long long *s_leaf, int *flags)
{
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lanemask_lt();
  int nfront = 0;
  // seed: the root once per image code (wave-uniform loop)
  for (int c = 0; c < 27; c++) {
    if (codes & (1u << c)) {
      if (lane == 0) s_front[0][nfront] = 0 | (c << GH_NODE_BITS);
      nfront++;
    }
  }
  __syncthreads();
  int cur = 0, nleaf = 0;
  const int leaf0 = d.gtot - 1;
  while (nfront > 0) {
    int nnext = 0;
    for (int base = 0; base < nfront; base += 64) {
      const int idx = base + lane;
      bool hit = false, isleaf = false;
      int n = 0, code = 0, first = 0, cnt = 0;
      if (idx < nfront) {
        const int e = s_front[cur][idx];
        n = e & GH_NODE_MASK; code = e >> GH_NODE_BITS;
        cnt = d.cN[n];
        if (cnt > 0) {
          double sh[3];
          code_shift(dom, code, sh);
          hit = pred(n, sh);
        }
        isleaf = n >= leaf0;
        if (hit && isleaf) first = d.cfirst[n];
      }
      const unsigned long long lm = __ballot(hit && isleaf);
      const unsigned long long om = __ballot(hit && !isleaf);
      if (hit && isleaf) {
        const int pos = nleaf + __popcll(lm & lt);
        if (pos < GH_LCAP) s_leaf[pos] = make_leaf_entry(first, cnt, code);
      }
      nleaf += __popcll(lm);
      if (hit && !isleaf) {
        const int pos = nnext + 2*__popcll(om & lt);
        if (pos + 1 < GH_FCAP) {
          s_front[cur ^ 1][pos] = (2*n + 1) | (code << GH_NODE_BITS);
          s_front[cur ^ 1][pos + 1] = (2*n + 2) | (code << GH_NODE_BITS);
        }
      }
      nnext += 2*__popcll(om);
    }
    if (nnext > GH_FCAP) { if (lane == 0) atomicOr(flags, FLAG_FRONTIER_OVERFLOW); nnext = GH_FCAP; }
    __syncthreads();
    cur ^= 1;
    nfront = nnext;
  }
  if (nleaf > GH_LCAP) { if (lane == 0) atomicOr(flags, FLAG_LEAFLIST_OVERFLOW); nleaf = GH_LCAP; }
  __syncthreads();
  return nleaf;
}

// which periodic image codes a search box [lo,hi] needs
__device__ __forceinline__ unsigned int image_codes(const Domain &dom, int ndim, const double lo[3], const double hi[3])
{
  int opt[3][3]; int nopt[3];
  for (int k = 0; k < 3; k++) {
    nopt[k] = 0; opt[k][nopt[k]++] = 0;
    if (k < ndim && dom.periodic[k]) {
      if (hi[k] > dom.bmax[k]) opt[k][nopt[k]++] = 1;   // images at x+L
      if (lo[k] < dom.bmin[k]) opt[k][nopt[k]++] = 2;   // images at x-L
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
