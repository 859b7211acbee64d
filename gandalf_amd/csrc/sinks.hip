// sinks.hip -- sink particles on the device-resident path.
//
// Replaces Sinks::SearchForNewSinkParticles / CreateNewSinkParticle / AccreteMassToSinks (reference
// src/Nbody/Sinks.cpp:118-273, 282-356, 365-770), the potential-minimum flag of GradhSph::ComputeH (GradhSph.cpp:270-280)
// and Hydrodynamics::DoDeleteDeadParticles (Hydrodynamics.h:158-202).
//
// What runs where.  Everything that touches all particles is a kernel: the candidate search (all formation criteria per
// particle, block arg-max of the density), the gas inside every sink radius (brute force over the particles - a few sinks
// against N particles is one streaming pass; the reference notes itself that it "should really use the tree"), the
// potential-minimum test (an ordered depth-first gather per dense particle), the removal of accreted particles (prefix
// sum + one scatter pass over all arrays).  What the reference does serially on a few dozen particles per sink - the
// distance sort, the energy sums with their pow() / exp() chains, the sequential mass transfer - runs on the host on the
// packed records of exactly those particles, in the reference's order and with the host's libm, and the result is
// scattered back (mass, dead flag).  Nothing else leaves the device.
//
// Order.  The reference walks its tree in cell order and a cell's particles in the order its quick-select left them;
// sink runs therefore always build the tree in exact mode (tree.hip), so that ascending device index IS that order.  The
// potential-minimum test needs it (the distance it tests for neighbour j is that of neighbour j-1, GradhSph.cpp:274-279),
// and so does the stable distance sort of the accretion list (ties).
#include "gh_internal.hpp"
#include "host_kernels.hpp"
#include <rocprim/rocprim.hpp>
#include <algorithm>
#include <cmath>
#include <unordered_set>

#define SK_SMALL 1.0e-20      /* small_number, Constants.h:73 */

// ------------------------------------------------------------------------------------------------
// potential-minimum flag (GradhSph.cpp:270-280), for the particles the sink search can select: rho >= rho_sink
// ------------------------------------------------------------------------------------------------
// The reference's neighbour list of particle i is every live particle j with |r_j - r_i|^2 + small <= (kernrange*hmax)^2 in
// tree order (GradhSphTree.cpp:200-219), hmax being the cell's 1.05^k hmax of the successful ComputeH call.  The flag is
// cleared if some neighbour j has gpot_j > 1.000000001 gpot_i while the distance of the neighbour BEFORE it in the list
// (for j = 0: of the last one) lies inside kernrange*h of the last h iteration.
//
// Several ranks: the candidates are this rank's own particles [p0, p0 + pn); their neighbours may be imported copies, of
// which the density-phase halo carries position, mass and - in sink-creating runs - last force pass's potential (comm.hip,
// LetLayout::gpot), so positions are read from the (x, y, z, m) pack everywhere.  The cells this test opens are among those
// the particle's own density walk opened (its cull radius is that walk's), so they were imported; one that was not raises
// FLAG_LET_MISS like in every other walk.
__device__ void potmin_serial(const DevicePtrs &d, int i, double kernrangesqd, int *flags);
__global__ void k_potmin(DevicePtrs d, double rho_sink, double kernrangesqd, int p0, int pn, int *flags)
{
  const int i = p0 + blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= p0 + pn) return;
  const int fl = (int) d.f[D_FLAGS][i];
  if (fl & GH_FLAG_DEAD) return;
  if (d.levels && !(fl & GH_FLAG_ACTIVE)) return;
  if (!(d.f[D_RHO][i] >= rho_sink)) return;
  potmin_serial(d, i, kernrangesqd, flags);
}
__device__ void potmin_serial(const DevicePtrs &d, int i, double kernrangesqd, int *flags)
{
  const int fl = (int) d.f[D_FLAGS][i];
  const double cull = d.pm_cullsqd[i], invhsqd = d.pm_invhsqd[i];
  const double thr = 1.000000001*d.f[D_GPOT][i];
  const double4 pi4 = d.posm[i];
  const double ri[3] = {pi4.x, pi4.y, pi4.z};
  int stack[40];
  int sp = 0;
  stack[sp++] = 0;
  bool pm = true, have_first = false;
  double d2_prev = 0.0, g_first = 0.0;
  const int leaf0 = d.gtot - 1;
  while (sp > 0) {
    const int n = stack[--sp];
    const CellBox b = d.cbox[n];
    if (b.N < 0) atomicOr(flags, FLAG_LET_MISS);
    if (b.N <= 0) continue;
    double md = 0.0;
    for (int k = 0; k < d.ndim; k++) {
      const double x = ri[k] < b.bbmin[k] ? b.bbmin[k] - ri[k] : (ri[k] > b.bbmax[k] ? ri[k] - b.bbmax[k] : 0.0);
      md += x*x;
    }
    if (md*(1.0 - 1e-12) > cull) continue;
    if (n >= leaf0) {
      for (int t = 0; t < b.N; t++) {
        const int j = b.first + t;
        const double4 pj = d.posm[j];
        const double rj[3] = {pj.x, pj.y, pj.z};
        double d2 = 0.0;
        for (int k = 0; k < d.ndim; k++) { const double dx = rj[k] - ri[k]; d2 += dx*dx; }
        if (!(d2 + SK_SMALL <= cull)) continue;
        const double gj = d.f[D_GPOT][j];
        if (!have_first) { have_first = true; g_first = gj; }
        else if (gj > thr && d2_prev*invhsqd < kernrangesqd) pm = false;
        d2_prev = d2;
      }
    }
    else if (sp + 2 <= 40) { stack[sp++] = 2*n + 2; stack[sp++] = 2*n + 1; }
  }
  if (have_first && g_first > thr && d2_prev*invhsqd < kernrangesqd) pm = false;
  d.f[D_FLAGS][i] = (double) (pm ? (fl | GH_FLAG_POTMIN) : (fl & ~GH_FLAG_POTMIN));
}

// The same test with one wavefront per dense particle.  The serial walk above costs one memory latency per node and per
// candidate (3.4 of the 15 ms of a 262 144-particle sink step); here the ordered list of leaves inside the cull radius is
// built level by level - lane = node, culled nodes dropped, the others replaced by their two children IN ORDER through a
// prefix sum (all leaves of the balanced tree sit on the last level) - and the candidates are then taken 8 leaves x 8
// slots at a time: "the neighbour before me in the list" is the nearest accepted lower lane (ballot), or the last accepted
// one of the previous chunk.  Same arithmetic, same decisions; a particle whose frontier outgrows the LDS list is left
// to the serial kernel (mask).
#define PM_CAP 1536
__global__ void k_potmin_collect(DevicePtrs d, double rho_sink, int *list, int *count, int p0, int pn)
{
  const int i = p0 + blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= p0 + pn) return;
  const int fl = (int) d.f[D_FLAGS][i];
  if (fl & GH_FLAG_DEAD) return;
  if (d.levels && !(fl & GH_FLAG_ACTIVE)) return;
  if (!(d.f[D_RHO][i] >= rho_sink)) return;
  list[atomicAdd(count, 1)] = i;
}

__global__ __launch_bounds__(64) void k_potmin_wave(DevicePtrs d, double kernrangesqd, const int *list, const int *count, int *redo, int *nredo, int *flags)
{
  __shared__ int s_a[PM_CAP], s_b[PM_CAP];
  const int lane = threadIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int leaf0 = d.gtot - 1;
  const int ncand = *count;
  for (int c = blockIdx.x; c < ncand; c += gridDim.x) {
    const int i = list[c];
    const int fl = (int) d.f[D_FLAGS][i];
    const double cull = d.pm_cullsqd[i], invhsqd = d.pm_invhsqd[i];
    const double thr = 1.000000001*d.f[D_GPOT][i];
    const double4 pi4 = d.posm[i];
    const double ri[3] = {pi4.x, pi4.y, pi4.z};
    int *cur = s_a, *nxt = s_b;
    int ncur = 1;
    bool overflow = false;
    __syncthreads();
    if (lane == 0) cur[0] = 0;
    __syncthreads();
    // levels 0 .. ltot: cull, expand in order; after the last round `cur` holds the kept leaves
    for (int lev = 0; lev <= d.ltot && !overflow; lev++) {
      int nout = 0;
      for (int c0 = 0; c0 < ncur; c0 += 64) {
        const int e = c0 + lane;
        int n = 0, cnt = 0;
        if (e < ncur) {
          n = cur[e];
          const CellBox b = d.cbox[n];
          if (b.N < 0) atomicOr(flags, FLAG_LET_MISS);
          if (b.N > 0) {
            double md = 0.0;
            for (int k = 0; k < d.ndim; k++) {
              const double x = ri[k] < b.bbmin[k] ? b.bbmin[k] - ri[k] : (ri[k] > b.bbmax[k] ? ri[k] - b.bbmax[k] : 0.0);
              md += x*x;
            }
            if (!(md*(1.0 - 1e-12) > cull)) cnt = n >= leaf0 ? 1 : 2;
          }
        }
        int inc = cnt;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (lane >= off) inc += t; }
        const int tot = __shfl(inc, 63, 64);
        if (nout + tot > PM_CAP) { overflow = true; break; }
        const int pos = nout + inc - cnt;
        if (cnt == 1) nxt[pos] = n;
        else if (cnt == 2) { nxt[pos] = 2*n + 1; nxt[pos + 1] = 2*n + 2; }
        nout += tot;
      }
      __syncthreads();
      int *t = cur; cur = nxt; nxt = t;
      ncur = nout;
    }
    if (overflow) {
      if (lane == 0) redo[atomicAdd(nredo, 1)] = i;
      continue;
    }
    // candidates in list order: 8 leaves x 8 slots per round
    bool pm = true, have_first = false;
    double d2_prev = 0.0, g_first = 0.0;
    for (int l0 = 0; l0 < ncur; l0 += 8) {
      const int li = l0 + (lane >> 3), t = lane & 7;
      bool acc = false;
      double d2 = 0.0, gj = 0.0;
      if (li < ncur) {
        const CellBox b = d.cbox[cur[li]];
        if (t < b.N) {
          const int j = b.first + t;
          const double4 pj = d.posm[j];
          const double rj[3] = {pj.x, pj.y, pj.z};
          for (int k = 0; k < d.ndim; k++) { const double dx = rj[k] - ri[k]; d2 += dx*dx; }
          acc = d2 + SK_SMALL <= cull;
          gj = d.f[D_GPOT][j];
        }
      }
      const unsigned long long am = __ballot(acc);
      if (am) {
        const unsigned long long below = am & lt;
        const int prevlane = below ? 63 - __clzll((long long) below) : 0;
        const double d2p_lane = __shfl(d2, prevlane, 64);
        const double d2p = below ? d2p_lane : d2_prev;
        const int firstlane = __ffsll((long long) am) - 1;
        const bool is_first = !have_first && lane == firstlane;
        const bool fail = acc && !is_first && gj > thr && d2p*invhsqd < kernrangesqd;
        if (__any(fail)) pm = false;
        if (!have_first) { g_first = __shfl(gj, firstlane, 64); have_first = true; }
        d2_prev = __shfl(d2, 63 - __clzll((long long) am), 64);
      }
    }
    if (have_first && g_first > thr && d2_prev*invhsqd < kernrangesqd) pm = false;
    if (lane == 0) d.f[D_FLAGS][i] = (double) (pm ? (fl | GH_FLAG_POTMIN) : (fl & ~GH_FLAG_POTMIN));
  }
}

// the serial test for the particles the wave kernel left (frontier larger than its list)
__global__ void k_potmin_redo(DevicePtrs d, double kernrangesqd, const int *redo, const int *nredo, int *flags)
{
  const int n = *nredo;
  for (int c = blockIdx.x*blockDim.x + threadIdx.x; c < n; c += gridDim.x*blockDim.x) potmin_serial(d, redo[c], kernrangesqd, flags);
}

// particles within a factor two of the sink density: while there are none (and no sink or star exists) the tree builds need
// not keep the reference's order inside the leaves (gh_tree_build_impl); the count is read at the next host synchronisation
__global__ void k_count_near_sink_density(DevicePtrs d, double rho_half, int p0, int pn, int *count)
{
  const int i = p0 + blockIdx.x*blockDim.x + threadIdx.x;
  const bool dense = i < p0 + pn && !((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD) && d.f[D_RHO][i] >= rho_half;
  const unsigned long long m = __ballot(dense);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, __popcll(m));
}

int gh_sinks_potmin(gh_ctx *ctx)
{
  if (!ctx->cfg.sink_particles || ctx->cfg.create_sinks != 1) return GH_OK;
  {
    int *word = ctx->d_blk + 19;
    GH_CHECK(ctx, hipMemsetAsync(word, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_count_near_sink_density, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), 0.5*ctx->cfg.rho_sink,
                       (int) ctx->own_first, (int) ctx->own_count, word);
  }
  const double krs = (ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 9.0 : 4.0;
  // scratch: candidate list and redo list in the (otherwise idle) sort-value buffer, two counters behind the block clock
  int *list = ctx->P[0][0], *redo = ctx->P[0][1], *cnt = ctx->d_blk + 16;
  // k_potmin_wave takes its candidates 8 leaves x 8 slots at a time: leaves wider than 8 particles (Nleafmax up to 32 is
  // accepted) go through the serial kernel, which walks every slot of a leaf
  const int p0 = (int) ctx->own_first, pn = (int) ctx->own_count;
  if (!list || !redo || ctx->leafocc > 8 || getenv("GH_POTMIN_SERIAL")) {
    hipLaunchKernelGGL(k_potmin, dim3(cdiv(pn, 64)), dim3(64), 0, ctx->stream, gh_dev(ctx), ctx->cfg.rho_sink, krs, p0, pn, ctx->d_flags);
    return GH_OK;
  }
  GH_CHECK(ctx, hipMemsetAsync(cnt, 0, 2*sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_potmin_collect, dim3(cdiv(pn, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), ctx->cfg.rho_sink, list, cnt, p0, pn);
  hipLaunchKernelGGL(k_potmin_wave, dim3(8192), dim3(64), 0, ctx->stream, gh_dev(ctx), krs, list, cnt, redo, cnt + 1, ctx->d_flags);
  hipLaunchKernelGGL(k_potmin_redo, dim3(256), dim3(64), 0, ctx->stream, gh_dev(ctx), krs, redo, cnt + 1, ctx->d_flags);
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// removal of dead particles (Hydrodynamics::DoDeleteDeadParticles) before a tree build
// ------------------------------------------------------------------------------------------------
__global__ void k_dead_collect(DevicePtrs d, int *count, int *slots, int *alive)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  const bool dead = ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD) != 0;
  alive[i] = dead ? 0 : 1;
  if (dead) slots[atomicAdd(count, 1)] = d.iorig[i];
}

// stable compaction of every particle array; a survivor whose slot lies beyond the new end takes the hole the
// reference moves it to (from[] sorted ascending).  src_off / dst_off: where this rank's own range starts before and
// after (several ranks: the ranges follow the new N, see gh_sinks_delete_dead)
__global__ void k_dead_compact(double **tab, const int *iorig_in, int *iorig_out, const int *alive, const int *newidx, int N,
                               const int *from, const int *to, int nmove, int nfields, size_t src_off, size_t dst_off)
{
  const int f = blockIdx.y;
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < N; i += gridDim.x*blockDim.x) {
    if (!alive[i]) continue;
    const int o = newidx[i];
    if (f < nfields) tab[D_COUNT + f][dst_off + o] = tab[f][src_off + i];
    else {
      int slot = iorig_in[src_off + i];
      int lo = 0, hi = nmove - 1;
      while (lo <= hi) { const int mid = (lo + hi) >> 1; if (from[mid] == slot) { slot = to[mid]; break; } if (from[mid] < slot) lo = mid + 1; else hi = mid - 1; }
      iorig_out[dst_off + o] = slot;
    }
  }
}

// Several ranks.  The reference's loop works on ITS array of all particles, whose slots are what `iorig` holds here
// (caller order, global): every rank gathers every rank's dead slots and runs the same loop, so the survivors' new slots
// are the single-rank run's.  The new particle count changes every tree cell's count and with it every rank's own range
// [own_first, own_first + own_count) (gh_alloc_tree): each rank compacts its survivors to the start of its NEW range and
// reports how many it holds (own_held); the migration of the tree build that follows evens the ranks out to the new cell
// counts (gh_dd_decompose: arrivals need not equal leavers then).
int gh_sinks_delete_dead(gh_ctx *ctx)
{
  if (!ctx->cfg.sink_particles || ctx->N <= 0) return GH_OK;
  const int N = (int) ctx->N;
  const int pn = (int) ctx->own_count;
  const size_t old_first = (size_t) ctx->own_first;
  hipStream_t s = ctx->stream;
  int *cnt = ctx->d_blk + 15;
  GH_CHECK(ctx, hipMemsetAsync(cnt, 0, sizeof(int), s));
  // scratch: P[0..2] of the idle build buffers are free between builds
  int *slots = ctx->P[0][0], *alive = ctx->P[0][1], *newidx = ctx->P[0][2];
  hipLaunchKernelGGL(k_dead_collect, dim3(cdiv(pn, 256)), dim3(256), 0, s, gh_dev_own(ctx), cnt, slots, alive);
  int nmine = 0;
  GH_CHECK(ctx, hipMemcpyAsync(&nmine, cnt, sizeof(int), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  std::vector<int> dslots;
  {
    std::vector<char> all; std::vector<size_t> sizes;
    int rc = gh_dd_gatherv(ctx, slots, sizeof(int)*(size_t) nmine, all, sizes);      // (one rank: the copy to the host)
    if (rc) return rc;
    dslots.assign((const int*) all.data(), (const int*) all.data() + all.size()/sizeof(int));
  }
  const int ndead = (int) dslots.size();
  if (ndead == 0) return GH_OK;
  std::sort(dslots.begin(), dslots.end());
  // the reference's loop, restricted to the slots it acts on (Hydrodynamics.h:171-185): every dead slot i < ilast takes
  // the particle of slot --ilast, again if that one was dead too
  std::unordered_set<int> dead(dslots.begin(), dslots.end());
  std::vector<std::pair<int, int> > moves;                  // (from slot, to slot) of live particles
  int ilast = N, Ndead = 0;
  for (size_t q = 0; q < dslots.size(); q++) {
    const int i = dslots[q];
    if (i >= ilast) break;                                   // the loop has ended (its exit test is i >= ilast - 1 after slot i - 1)
    bool idead = true;
    while (idead) {
      Ndead++; ilast--;
      if (i < ilast) { idead = dead.count(ilast) != 0; if (!idead) moves.push_back(std::make_pair(ilast, i)); }
      else break;
    }
  }
  if (Ndead != ndead) return gh_fail(ctx, GH_ERR_INVALID, "gh_sinks_delete_dead: dead-particle bookkeeping out of step with the reference's loop");
  const int Nnew = N - Ndead;
  std::sort(moves.begin(), moves.end());
  std::vector<int> from(moves.size() + 1, 0), to(moves.size() + 1, 0);
  for (size_t q = 0; q < moves.size(); q++) { from[q] = moves[q].first; to[q] = moves[q].second; }
  int *d_from = ctx->P[1][0], *d_to = ctx->P[1][1];
  GH_CHECK(ctx, hipMemcpyAsync(d_from, from.data(), sizeof(int)*from.size(), hipMemcpyHostToDevice, s));
  GH_CHECK(ctx, hipMemcpyAsync(d_to, to.data(), sizeof(int)*to.size(), hipMemcpyHostToDevice, s));
  size_t need = 0;
  GH_CHECK(ctx, rocprim::exclusive_scan(nullptr, need, alive, newidx, 0, (size_t) pn, rocprim::plus<int>(), s));
  if (need > ctx->sorttemp_bytes) {
    if (ctx->sorttemp) (void) hipFree(ctx->sorttemp);
    ctx->sorttemp = nullptr; ctx->sorttemp_bytes = 0;
    GH_CHECK(ctx, hipMalloc(&ctx->sorttemp, need));
    ctx->sorttemp_bytes = need;
  }
  GH_CHECK(ctx, rocprim::exclusive_scan(ctx->sorttemp, need, alive, newidx, 0, (size_t) pn, rocprim::plus<int>(), s));
  // the new particle count fixes the new cell counts and this rank's new range
  ctx->N = Nnew;
  if (ctx->nranks > 1) {
    GH_CHECK(ctx, hipStreamSynchronize(s));
    const int rc = gh_alloc_tree(ctx);
    if (rc) return rc;
  }
  else { ctx->own_first = 0; ctx->own_count = Nnew; }
  const size_t new_first = (size_t) ctx->own_first;
  hipLaunchKernelGGL(k_dead_compact, dim3(std::min(cdiv(pn, 256), 1024), D_COUNT + 1), dim3(256), 0, s,
                     ctx->d_ptrtab + (size_t) ctx->cur*2*D_COUNT, ctx->iorig[ctx->cur], ctx->iorig[ctx->cur ^ 1], alive, newidx, pn,
                     d_from, d_to, (int) moves.size(), D_COUNT, old_first, new_first);
  ctx->cur ^= 1;
  GH_CHECK(ctx, hipStreamSynchronize(s));                   // from / to live on the host stack until the kernel has read them
  const int held = pn - nmine;
  if (ctx->nranks > 1) ctx->own_held = held;
  ctx->tree_valid = false;
  ctx->rebuild_tree = true;
  // a global-timestep run keeps only "dead" and "potmin" in the flag word: both start afresh
  if (ctx->cfg.Nlevels <= 1 && held > 0) GH_CHECK(ctx, hipMemsetAsync(ctx->fbuf[ctx->cur][D_FLAGS] + new_first, 0, sizeof(double)*(size_t) held, s));
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// rows of particle data for the host side of the sink routines
// ------------------------------------------------------------------------------------------------
#define SK_ROW (D_COUNT + 1)          /* every field + the caller-order slot */
__global__ void k_pack_rows(DevicePtrs d, const int *idx, int n, double *out)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int j = idx[e];
  for (int f = 0; f < D_COUNT; f++) out[(size_t) e*SK_ROW + f] = d.f[f][j];
  out[(size_t) e*SK_ROW + D_COUNT] = (double) d.iorig[j];
}

struct SinkTab { const double *r, *v, *a, *rad; int n, ndim; };   // [n][3] star position / velocity / acceleration, [n] sink radius

// formation criteria of one particle against every sink (Sinks.cpp:158-197); block arg-max of the density over the
// particles that pass (ties: the lowest slot, as the reference's ascending scan with "rho > rho_max" keeps the first)
__global__ __launch_bounds__(256) void k_sink_search(DevicePtrs d, SinkTab S, double rho_sink, double sink_radius, int n, double *best_rho, int *best_slot, int *best_idx)
{
  __shared__ double s_rho[256];
  __shared__ int s_slot[256], s_idx[256];
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  double rho = -1.0; int slot = 0x7fffffff, idx = -1;
  if (i < d.N) {
    const int fl = (int) d.f[D_FLAGS][i];
    const double rh = d.f[D_RHO][i];
    bool ok = !(fl & GH_FLAG_DEAD) && (fl & GH_FLAG_POTMIN) && !(rh < rho_sink);
    if (ok && d.levels && n%(int) d.f[D_NSTEP][i] != 0) ok = false;      // the candidate is at the end of its own step (:172)
    if (ok) {
      const double h = d.f[D_H][i];
      for (int s = 0; s < S.n && ok; s++) {
        double drsqd = 0.0, dadr = 0.0, dvdr = 0.0;
        for (int k = 0; k < d.ndim; k++) {
          const double dr = d.f[D_RX + k][i] - S.r[3*s + k];
          drsqd += dr*dr; dadr += dr*(d.f[D_AX + k][i] - S.a[3*s + k]); dvdr += dr*(d.f[D_VX + k][i] - S.v[3*s + k]);
        }
        const double tff = 0.5/sqrt(rh);
        if (tff > drsqd/dvdr && dvdr > 0) ok = false;
        if (rh < -dadr/drsqd) ok = false;
        const double rr = sink_radius*h + S.rad[s];
        if (drsqd < rr*rr) ok = false;
      }
    }
    if (ok) { rho = rh; slot = d.iorig[i]; idx = i; }
  }
  s_rho[threadIdx.x] = rho; s_slot[threadIdx.x] = slot; s_idx[threadIdx.x] = idx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) {
      const int o = threadIdx.x + off;
      if (s_rho[o] > s_rho[threadIdx.x] || (s_rho[o] == s_rho[threadIdx.x] && s_slot[o] < s_slot[threadIdx.x])) {
        s_rho[threadIdx.x] = s_rho[o]; s_slot[threadIdx.x] = s_slot[o]; s_idx[threadIdx.x] = s_idx[o];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { best_rho[blockIdx.x] = s_rho[0]; best_slot[blockIdx.x] = s_slot[0]; best_idx[blockIdx.x] = s_idx[0]; }
}

// live particles strictly inside `radsqd` of a point (Tree.cpp:246-256 as Sinks.cpp uses it), appended in any order
__global__ void k_sink_within(DevicePtrs d, const double *centre, double radsqd, int *count, int *list, int cap)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD) return;
  double d2 = 0.0;
  for (int k = 0; k < d.ndim; k++) { const double dr = centre[k] - d.f[D_RX + k][i]; d2 += dr*dr; }
  if (d2 < radsqd) { const int p = atomicAdd(count, 1); if (p < cap) list[p] = i; }
}

// Sinks.cpp:377-468: sinkid of every particle (the LAST sink whose radius holds it), the gas count of every sink and the
// (sink, particle) pairs
__global__ void k_sink_assign(DevicePtrs d, SinkTab S, int *ngas, int *count, int2 *pairs, int cap)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  int sid = -1;
  if (!((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD)) {
    for (int s = 0; s < S.n; s++) {
      double d2 = 0.0;
      for (int k = 0; k < d.ndim; k++) { const double dr = d.f[D_RX + k][i] - S.r[3*s + k]; d2 += dr*dr; }
      if (d2 < S.rad[s]*S.rad[s]) {
        sid = s;
        atomicAdd(&ngas[s], 1);
        const int p = atomicAdd(count, 1);
        if (p < cap) pairs[p] = make_int2(s, i);
      }
    }
  }
  d.f[D_SINKID][i] = (double) sid;
}

// mass left to a particle after accretion; 0 = wholly accreted: dead (Sinks.cpp:696-704)
__global__ void k_sink_apply(DevicePtrs d, const int *idx, const double *mnew, int n)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int j = idx[e];
  d.f[D_M][j] = mnew[e];
  if (mnew[e] == 0.0) d.f[D_FLAGS][j] = (double) ((((int) d.f[D_FLAGS][j]) | GH_FLAG_DEAD) & ~GH_FLAG_ACTIVE);
}

// part.levelneib = max(part.levelneib, star->level) for the gas a sink accretes from (Sinks.cpp:516)
__global__ void k_sink_levelneib(DevicePtrs d, const int *idx, const int *lvl, int n)
{
  const int e = blockIdx.x*blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int j = idx[e];
  if ((int) d.f[D_LEVELNEIB][j] < lvl[e]) d.f[D_LEVELNEIB][j] = (double) lvl[e];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// one field of every particle in caller (slot) order
__global__ void k_slot_order(DevicePtrs d, int field, double *out)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < d.N) out[d.iorig[i]] = d.f[field][i];
}

struct SinkScratch {
  double *d_slot = nullptr, *h_slot = nullptr; size_t cap_slot = 0;     // masses in slot order (device, pinned host): mmean
  double *d_st = nullptr;        // star table: r[3n] v[3n] a[3n] rad[n] + centre[3]
  int *d_i = nullptr;            // counters / lists
  double *d_rows = nullptr, *h_rows = nullptr;
  size_t cap_st = 0, cap_i = 0, cap_rows = 0;
};

static int sk_reserve(gh_ctx *ctx, SinkScratch &W, size_t nstar, size_t nint, size_t nrows)
{
  // (a failed allocation leaves pointer and capacity cleared)
  if (10*nstar + 8 > W.cap_st) { if (W.d_st) (void) hipFree(W.d_st); W.d_st = nullptr; W.cap_st = 0; GH_CHECK(ctx, hipMalloc((void**) &W.d_st, sizeof(double)*(10*nstar + 64))); W.cap_st = 10*nstar + 64; }
  if (nint > W.cap_i) { if (W.d_i) (void) hipFree(W.d_i); W.d_i = nullptr; W.cap_i = 0; GH_CHECK(ctx, hipMalloc((void**) &W.d_i, sizeof(int)*(nint + 1024))); W.cap_i = nint + 1024; }
  if (nrows*SK_ROW > W.cap_rows) { if (W.d_rows) (void) hipFree(W.d_rows); W.d_rows = nullptr; W.cap_rows = 0; GH_CHECK(ctx, hipMalloc((void**) &W.d_rows, sizeof(double)*(nrows + 256)*SK_ROW)); W.cap_rows = (nrows + 256)*SK_ROW; }
  return GH_OK;
}

static SinkScratch &sk_scratch(gh_ctx *ctx)
{
  if (!ctx->sink_scratch) ctx->sink_scratch = new SinkScratch;
  return *(SinkScratch*) ctx->sink_scratch;
}
void gh_sinks_free(gh_ctx *ctx)
{
  SinkScratch *W = (SinkScratch*) ctx->sink_scratch;
  if (!W) return;
  if (W->d_st) (void) hipFree(W->d_st);
  if (W->d_i) (void) hipFree(W->d_i);
  if (W->d_rows) (void) hipFree(W->d_rows);
  if (W->d_slot) (void) hipFree(W->d_slot);
  if (W->h_slot) (void) hipHostFree(W->h_slot);
  delete W;
  ctx->sink_scratch = nullptr;
}

// the sinks' stars as the kernels see them
static int sk_upload_stars(gh_ctx *ctx, SinkScratch &W, const gh_host_stars &S, SinkTab &T)
{
  const int ns = (int) ctx->sinks.size();
  std::vector<double> t((size_t) 10*ns + 1, 0.0);
  for (int s = 0; s < ns; s++) {
    const int is = ctx->sinks[s].istar;
    for (int k = 0; k < 3; k++) { t[3*s + k] = S.r[3*is + k]; t[3*ns + 3*s + k] = S.v[3*is + k]; t[6*ns + 3*s + k] = S.a[3*is + k]; }
    t[9*ns + s] = ctx->sinks[s].radius;
  }
  GH_CHECK(ctx, hipMemcpyAsync(W.d_st, t.data(), sizeof(double)*(size_t) (10*ns + 1), hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  T.r = W.d_st; T.v = W.d_st + 3*ns; T.a = W.d_st + 6*ns; T.rad = W.d_st + 9*ns; T.n = ns; T.ndim = ctx->ndim;
  return GH_OK;
}

// rows (all fields + slot) of the particles idx[0..n) - indices into this rank's own range, like everywhere below
static int sk_fetch_rows(gh_ctx *ctx, SinkScratch &W, const std::vector<int> &idx, std::vector<double> &rows)
{
  const size_t n = idx.size();
  rows.assign(n*SK_ROW, 0.0);
  if (n == 0) return GH_OK;
  int rc = sk_reserve(ctx, W, ctx->sinks.size(), n + 16, n);
  if (rc) return rc;
  GH_CHECK(ctx, hipMemcpyAsync(W.d_i, idx.data(), sizeof(int)*n, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_pack_rows, dim3(cdiv(n, 64)), dim3(64), 0, ctx->stream, gh_dev_own(ctx), W.d_i, (int) n, W.d_rows);
  GH_CHECK(ctx, hipMemcpyAsync(rows.data(), W.d_rows, sizeof(double)*n*SK_ROW, hipMemcpyDeviceToHost, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return GH_OK;
}

static inline double dot3(const double *a, const double *b, int nd) { double s = 0.0; for (int k = 0; k < nd; k++) s += a[k]*b[k]; return s; }

// kernel functions of the accretion sums as hydro->kernp evaluates them: the analytic functions, or the 1000-entry
// piecewise-constant tables of tabulated_kernel = 1 (TabulatedKernel.cpp:57-100, SmoothingKernel.h:620-640)
struct SinkKernel {
  bool quintic, tab; int nd; double kernrange, invkernrange;
  std::vector<double> tW0, tWpot;
  SinkKernel(const gh_ctx *ctx) {
    quintic = ctx->cfg.kernel == GH_KERNEL_QUINTIC || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB;
    tab = ctx->cfg.kernel == GH_KERNEL_M4_TAB || ctx->cfg.kernel == GH_KERNEL_QUINTIC_TAB;
    nd = ctx->ndim; kernrange = quintic ? 3.0 : 2.0; invkernrange = quintic ? 1.0/3.0 : 0.5;
    if (tab) {
      tW0.resize(GH_TAB_RES); tWpot.resize(GH_TAB_RES);
      const double step = kernrange/GH_TAB_RES;
      for (int i = 0; i < GH_TAB_RES; i++) { tW0[i] = a_w0(step*i); tWpot[i] = a_wpot(step*i); }
    }
  }
  double a_w0(double s) const { return quintic ? HostQuintic(nd).w0(s) : HostM4(nd).w0(s); }
  double a_wpot(double s) const { return quintic ? HostQuintic(nd).wpot(s) : HostM4(nd).wpot(s); }
  double w0(double s) const { if (!tab) return a_w0(s); if (s >= kernrange) return 0.0; return tW0[(int) (s*(GH_TAB_RES/kernrange))]; }
  double wpot(double s) const { if (!tab) return a_wpot(s); if (s >= kernrange) return 1.0/s; return tWpot[(int) (s*(GH_TAB_RES/kernrange))]; }
};

// Sinks::CreateNewSinkParticle (Sinks.cpp:282-356) from the packed row of the chosen particle + the mass inside the new
// sink (Sinks.cpp:243-253).  Several ranks: `owner` holds the particle (own index idx) and hands its row to everybody;
// every rank appends the same star and sink; the mass inside the radius is summed in slot order over all ranks' particles.
static int sk_create(gh_ctx *ctx, SinkScratch &W, gh_host_stars &S, int idx, int owner, double t)
{
  const int nd = ctx->ndim;
  const bool mine = owner == ctx->rank;
  std::vector<double> row;
  int rc;
  {
    std::vector<int> one;
    if (mine) one.push_back(idx);
    std::vector<double> own_row;
    if ((rc = sk_fetch_rows(ctx, W, one, own_row))) return rc;
    std::vector<char> all; std::vector<size_t> sizes;
    if ((rc = gh_dd_gatherv(ctx, own_row.data(), sizeof(double)*own_row.size(), all, sizes))) return rc;
    if (all.size() != sizeof(double)*SK_ROW) return gh_fail(ctx, GH_ERR_INVALID, "sink creation: the chosen particle's row did not arrive");
    row.assign((const double*) all.data(), (const double*) all.data() + SK_ROW);
  }
  const SinkKernel K(ctx);
  gh_sink_rec sk = gh_sink_rec();
  const double h = row[D_H];
  if (ctx->cfg.sink_radius_mode == 0) sk.radius = ctx->cfg.sink_radius;
  else if (ctx->cfg.sink_radius_mode == 1) sk.radius = ctx->cfg.sink_radius*h;
  else sk.radius = K.kernrange*h;
  sk.invh = 1.0/h;
  sk.istar = (int) S.n;
  S.append();
  const size_t o = (size_t) 3*sk.istar;
  for (int k = 0; k < nd; k++) {
    S.r[o + k] = row[D_RX + k]; S.v[o + k] = row[D_VX + k]; S.a[o + k] = row[D_AX + k]; S.adot[o + k] = 0.0;
    S.r0[o + k] = row[D_R0X + k]; S.v0[o + k] = row[D_V0X + k]; S.a0[o + k] = row[D_A0X + k];
  }
  S.m[sk.istar] = row[D_M]; S.h[sk.istar] = K.invkernrange*sk.radius; S.gpot[sk.istar] = row[D_GPOT];
  S.tlast[sk.istar] = t; S.dti[sk.istar] = 9.9e20;
  if (ctx->cfg.Nlevels > 1) { S.level[sk.istar] = (int) row[D_LEVEL]; S.nstep[sk.istar] = (int) row[D_NSTEP]; S.nlast[sk.istar] = (int) row[D_NLAST]; }
  // the particle is gone: m = 0, dead (W.d_i[0] still holds its index from the row fetch)
  if (mine) {
    const double zero = 0.0;
    GH_CHECK(ctx, hipMemcpyAsync(W.d_rows, &zero, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sink_apply, dim3(1), dim3(64), 0, ctx->stream, gh_dev_own(ctx), W.d_i, W.d_rows, 1);
  }
  // mmax: masses of the live particles inside the radius, summed in slot order
  double c[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < nd; k++) c[k] = S.r[o + k];
  const int pn = (int) ctx->own_count;
  if ((rc = sk_reserve(ctx, W, ctx->sinks.size() + 1, (size_t) pn + 16, 1))) return rc;
  GH_CHECK(ctx, hipMemcpyAsync(W.d_st, c, sizeof(c), hipMemcpyHostToDevice, ctx->stream));
  GH_CHECK(ctx, hipMemsetAsync(W.d_i, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_sink_within, dim3(cdiv(pn, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), W.d_st, sk.radius*sk.radius, W.d_i, W.d_i + 1, pn);
  int cnt = 0;
  GH_CHECK(ctx, hipMemcpyAsync(&cnt, W.d_i, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  std::vector<int> in((size_t) cnt);
  if (cnt) GH_CHECK(ctx, hipMemcpy(in.data(), W.d_i + 1, sizeof(int)*(size_t) cnt, hipMemcpyDeviceToHost));
  std::vector<double> rows;
  if ((rc = sk_fetch_rows(ctx, W, in, rows))) return rc;
  std::vector<double> mine_sm((size_t) 2*cnt);              // (slot, mass) of this rank's particles inside
  for (int e = 0; e < cnt; e++) { mine_sm[2*e] = rows[(size_t) e*SK_ROW + D_COUNT]; mine_sm[2*e + 1] = rows[(size_t) e*SK_ROW + D_M]; }
  std::vector<char> all; std::vector<size_t> sizes;
  if ((rc = gh_dd_gatherv(ctx, mine_sm.data(), sizeof(double)*mine_sm.size(), all, sizes))) return rc;
  const size_t ntot = all.size()/(2*sizeof(double));
  const double *q = (const double*) all.data();
  std::vector<std::pair<int, double> > sm(ntot);
  for (size_t e = 0; e < ntot; e++) sm[e] = std::make_pair((int) q[2*e], q[2*e + 1]);
  std::sort(sm.begin(), sm.end());
  sk.mmax = 0.0;
  for (size_t e = 0; e < ntot; e++) sk.mmax += sm[e].second;
  ctx->sinks.push_back(sk);
  return GH_OK;
}

// Sinks::SearchForNewSinkParticles (Sinks.cpp:118-273); the global timestep makes "n % nstep == 0" true for everyone.
// Several ranks: every rank's best candidate (density, slot) is gathered; the densest wins, the lowest slot among equals -
// the particle the reference's ascending scan over all slots keeps (its MPI build reduces the same way, Sinks.cpp:207-241).
static int sk_search(gh_ctx *ctx, SinkScratch &W, gh_host_stars &S, double t)
{
  if ((int) ctx->sinks.size() >= ctx->cfg.Nsinkfixed && ctx->cfg.Nsinkfixed != -1) return GH_OK;
  const int pn = (int) ctx->own_count;
  const int nblk = cdiv(pn, 256);
  for (;;) {
    int rc = sk_reserve(ctx, W, ctx->sinks.size() + 1, (size_t) 2*nblk + 16, (size_t) nblk/SK_ROW + 2);
    if (rc) return rc;
    SinkTab T;
    if ((rc = sk_upload_stars(ctx, W, S, T))) return rc;
    double *b_rho = W.d_rows; int *b_slot = W.d_i, *b_idx = W.d_i + nblk;
    hipLaunchKernelGGL(k_sink_search, dim3(nblk), dim3(256), 0, ctx->stream, gh_dev_own(ctx), T, ctx->cfg.rho_sink, ctx->cfg.sink_radius, ctx->n, b_rho, b_slot, b_idx);
    std::vector<double> hr((size_t) nblk); std::vector<int> hs((size_t) 2*nblk);
    GH_CHECK(ctx, hipMemcpyAsync(hr.data(), b_rho, sizeof(double)*(size_t) nblk, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK(ctx, hipMemcpyAsync(hs.data(), b_slot, sizeof(int)*(size_t) 2*nblk, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double rho = 0.0; int slot = 0x7fffffff, idx = -1;               // rho_max starts at 0: a candidate needs rho > 0
    for (int b = 0; b < nblk; b++)
      if (hs[nblk + b] >= 0 && (hr[b] > rho || (hr[b] == rho && idx >= 0 && hs[b] < slot))) { rho = hr[b]; slot = hs[b]; idx = hs[nblk + b]; }
    int owner = ctx->rank;
    if (ctx->nranks > 1) {
      const double mine[2] = {idx >= 0 ? rho : 0.0, idx >= 0 ? (double) slot : -1.0};
      std::vector<char> all; std::vector<size_t> sizes;
      if ((rc = gh_dd_gatherv(ctx, mine, sizeof(mine), all, sizes))) return rc;
      const double *q = (const double*) all.data();
      double brho = 0.0; int bslot = 0x7fffffff; owner = -1;
      for (int r = 0; r < ctx->nranks; r++) {
        const double rr = q[2*r]; const int rs = (int) q[2*r + 1];
        if (rs >= 0 && (rr > brho || (rr == brho && owner >= 0 && rs < bslot))) { brho = rr; bslot = rs; owner = r; }
      }
      if (owner < 0) return GH_OK;
    }
    else if (idx < 0) return GH_OK;
    if ((rc = sk_create(ctx, W, S, idx, owner, t))) return rc;
  }
}

// Sinks::AccreteMassToSinks (Sinks.cpp:365-770).  Several ranks: every rank finds its own particles inside the sink
// radii and all ranks gather all of their rows (a few dozen per sink), rank by rank = in global tree order; the serial part
// below then runs on every rank alike, and each rank writes back the masses of its own particles.
static int sk_accrete(gh_ctx *ctx, SinkScratch &W, gh_host_stars &S, double timestep)
{
  const int nd = ctx->ndim;
  const int ns = (int) ctx->sinks.size();
  const int N = (int) ctx->own_count;
  int rc;
  // slots: [0] pair count, [1 .. ns] Ngas, then the pairs
  const int cap = std::max(4096, N/4);
  if ((rc = sk_reserve(ctx, W, ns, (size_t) 2*cap + ns + 16, 1))) return rc;
  SinkTab T;
  if ((rc = sk_upload_stars(ctx, W, S, T))) return rc;
  GH_CHECK(ctx, hipMemsetAsync(W.d_i, 0, sizeof(int)*(size_t) (ns + 2), ctx->stream));
  int2 *d_pairs = (int2*) (W.d_i + ((ns + 2 + 1) & ~1));
  hipLaunchKernelGGL(k_sink_assign, dim3(cdiv(N, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), T, W.d_i + 1, W.d_i, d_pairs, cap);
  std::vector<int> head((size_t) ns + 1);
  GH_CHECK(ctx, hipMemcpyAsync(head.data(), W.d_i, sizeof(int)*(size_t) (ns + 1), hipMemcpyDeviceToHost, ctx->stream));
  GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  // (a rank whose pair buffer overflowed says so in its block of the gather below: all ranks stop together)
  const bool toomany = head[0] > cap;
  const int np = toomany ? 0 : head[0];
  std::vector<int2> pairs((size_t) np);
  if (np) GH_CHECK(ctx, hipMemcpy(pairs.data(), d_pairs, sizeof(int2)*(size_t) np, hipMemcpyDeviceToHost));
  // a particle accretes to the last sink that holds it; per sink the list in tree order (= the reference's neiblist order)
  std::sort(pairs.begin(), pairs.end(), [](const int2 &a, const int2 &b) { return a.y != b.y ? a.y < b.y : a.x < b.x; });
  std::vector<std::vector<int> > lists((size_t) ns);
  for (int e = 0; e < np; e++) if (e + 1 == np || pairs[e + 1].y != pairs[e].y) lists[pairs[e].x].push_back(pairs[e].y);
  std::vector<int> mine;                                             // this rank's particles, sink by sink
  for (int s = 0; s < ns; s++) mine.insert(mine.end(), lists[s].begin(), lists[s].end());
  std::vector<double> myrows;
  if ((rc = sk_fetch_rows(ctx, W, mine, myrows))) return rc;
  // every rank's block: [Ngas of every sink | rows per sink | rows ...]; `who` = (rank, own index) of every gathered row
  std::vector<double> rows;
  std::vector<std::pair<int, int> > who;
  std::vector<size_t> off((size_t) ns + 1, 0);
  {
    std::vector<double> blk((size_t) 2*ns + 1 + myrows.size());
    for (int s = 0; s < ns; s++) { blk[s] = (double) head[1 + s]; blk[ns + s] = (double) lists[s].size(); }
    blk[2*ns] = toomany ? 1.0 : 0.0;
    std::copy(myrows.begin(), myrows.end(), blk.begin() + 2*ns + 1);
    std::vector<char> all; std::vector<size_t> sizes;
    if ((rc = gh_dd_gatherv(ctx, blk.data(), sizeof(double)*blk.size(), all, sizes))) return rc;
    const int Wn = ctx->nranks;
    std::vector<const double*> base((size_t) Wn);
    { size_t o = 0; for (int r = 0; r < Wn; r++) { base[r] = (const double*) (all.data() + o); o += sizes[r]; } }
    for (int r = 0; r < Wn; r++)
      if (base[r][2*ns] != 0.0) return gh_fail(ctx, GH_ERR_CAPACITY, "sink accretion: more than N/4 particles inside sink radii");
    for (int s = 0; s < ns; s++) ctx->sinks[s].Ngas = 0;
    std::vector<size_t> cursor((size_t) Wn, 0), local((size_t) Wn, 0);      // rows / own-list entries of rank r consumed so far
    for (int s = 0; s < ns; s++) {
      off[s] = who.size();
      for (int r = 0; r < Wn; r++) {
        ctx->sinks[s].Ngas += (int) base[r][s];
        const size_t cnt = (size_t) base[r][ns + s];
        const double *src = base[r] + 2*ns + 1 + cursor[r]*SK_ROW;
        rows.insert(rows.end(), src, src + cnt*SK_ROW);
        for (size_t e = 0; e < cnt; e++) who.push_back(std::make_pair(r, r == ctx->rank ? mine[local[r] + e] : -1));
        cursor[r] += cnt; local[r] += cnt;
      }
    }
    off[ns] = who.size();
  }
  std::vector<double> mnew(who.size());
  for (size_t e = 0; e < who.size(); e++) mnew[e] = rows[e*SK_ROW + D_M];
  const SinkKernel K(ctx);
  const double small_number = SK_SMALL, pi = 3.14159265358979, twopi = 6.28318530717959;
  std::vector<int> lnb_idx, lnb_lvl;                                 // levelneib of the particles a sink accretes from (:516)

  for (int s = 0; s < ns; s++) {
    gh_sink_rec &sk = ctx->sinks[s];
    const int is = sk.istar;
    double *sr = &S.r[3*is], *sv = &S.v[3*is], *sa = &S.a[3*is];
    double &sm = S.m[is];
    if (sk.Ngas == 0 || ctx->n%S.nstep[is] != 0) continue;
    double wnorm = 0.0;
    sk.menc = 0.0; sk.trad = 0.0; sk.tvisc = 1.0; sk.ketot = 0.0; sk.rotketot = 0.0; sk.gpetot = 0.0;
    // particles of this sink inside its radius, sorted by distance (InsertionSortIds, InlineFuncs.h:226-248: stable)
    const size_t e0 = off[s], e1 = off[s + 1];
    std::vector<size_t> il; std::vector<double> rs;
    double dr[3], dv[3], dvtang[3];
    for (size_t e = e0; e < e1; e++) {
      const double *p = &rows[e*SK_ROW];
      for (int k = 0; k < nd; k++) dr[k] = p[D_RX + k] - sr[k];
      const double drsqd = dot3(dr, dr, nd);
      if (drsqd > sk.radius*sk.radius) continue;
      il.push_back(e); rs.push_back(drsqd);
      if (ctx->cfg.Nlevels > 1 && who[e].first == ctx->rank) { lnb_idx.push_back(who[e].second); lnb_lvl.push_back(S.level[is]); }
    }
    const int Nneib = (int) il.size();
    for (int j = 1; j < Nneib; j++) {
      const double raux = rs[j]; const size_t iaux = il[j];
      int i;
      for (i = j - 1; i >= 0; i--) { if (rs[i] <= raux) break; rs[i + 1] = rs[i]; il[i + 1] = il[i]; }
      rs[i + 1] = raux; il[i + 1] = iaux;
    }
    const double invh = sk.invh;
    for (int j = 0; j < Nneib; j++) {                                  // Sinks.cpp:524-561
      const double *p = &rows[il[j]*SK_ROW];
      const double pm = mnew[il[j]];
      for (int k = 0; k < nd; k++) dr[k] = p[D_RX + k] - sr[k];
      const double drsqd = dot3(dr, dr, nd);
      const double drmag = sqrt(drsqd) + small_number;
      for (int k = 0; k < nd; k++) dr[k] /= drmag;
      sk.menc += pm;
      wnorm += pm*K.w0(drmag*invh)*pow(invh, nd)/p[D_RHO];
      sk.gpetot += 0.5*pm*(sm + sk.menc)*invh*K.wpot(drmag*invh);
      for (int k = 0; k < nd; k++) dv[k] = p[D_VX + k] - sv[k];
      for (int k = 0; k < nd; k++) dvtang[k] = dv[k] - dot3(dv, dr, nd)*dr[k];
      sk.ketot += pm*dot3(dv, dv, nd)*K.w0(drmag*invh)*pow(invh, nd)/p[D_RHO];
      sk.rotketot += pm*dot3(dvtang, dvtang, nd)*K.w0(drmag*invh)*pow(invh, nd)/p[D_RHO];
      sk.tvisc *= pow(sqrt(drmag)/p[D_SOUND]/p[D_SOUND], pm);
      sk.trad += fabs(4.0*pi*drsqd*pm*dot3(dv, dr, nd)*K.w0(drmag*invh)*pow(invh, nd));
    }
    sk.ketot *= 0.5*sk.menc/wnorm;
    sk.rotketot *= 0.5*sk.menc/wnorm;
    double macc, dt;
    if (ctx->cfg.smooth_accretion == 1) {                              // Sinks.cpp:573-602
      const double efrac = std::min(2.0*sk.rotketot/sk.gpetot, 1.0);
      sk.tvisc = (sqrt(sm + sk.menc)*pow(sk.tvisc, 1.0/sk.menc))/ctx->cfg.alpha_ss;
      sk.trad = sk.menc/sk.trad;
      sk.trot = twopi*sqrt(pow(sk.radius, 3)/(sk.menc + sm));
      sk.taccrete = pow(sk.trad, 1.0 - efrac)*pow(sk.tvisc, efrac);
      if (sk.mmax > small_number && sk.menc > sk.mmax) sk.taccrete *= pow(sk.mmax/sk.menc, 2);
      dt = (double) S.nstep[is]*timestep;                              // star->nstep * timestep
      macc = sk.menc*std::max(1.0 - exp(-dt/sk.taccrete), 0.0);
      sk.dmdt = macc/dt;
    }
    else { macc = sk.menc; sk.dmdt = macc/timestep; }
    double macc_temp = macc, rold[3], vold[3];
    for (int k = 0; k < nd; k++) { rold[k] = sr[k]; vold[k] = sv[k]; }
    const double mold = sm;
    for (int k = 0; k < nd; k++) { sr[k] *= sm; sv[k] *= sm; sa[k] *= sm; }
    for (int j = 0; j < Nneib; j++) {                                  // Sinks.cpp:626-654
      const double *p = &rows[il[j]*SK_ROW];
      const double pm = mnew[il[j]];
      double mtemp = std::min(pm, macc_temp);
      dt = p[D_DT];
      if (ctx->cfg.smooth_accretion == 0 || pm - mtemp < ctx->cfg.smooth_accrete_frac*ctx->mmean || dt < ctx->cfg.smooth_accrete_dt*sk.trot) mtemp = pm;
      macc_temp -= mtemp;
      sm += mtemp;
      for (int k = 0; k < nd; k++) { sr[k] += mtemp*p[D_RX + k]; sv[k] += mtemp*p[D_VX + k]; sa[k] += mtemp*p[D_AX + k]; }
      sk.utot += mtemp*p[D_U];
      if (macc_temp < small_number) break;
    }
    for (int k = 0; k < nd; k++) { sr[k] /= sm; sv[k] /= sm; sa[k] /= sm; }
    for (int k = 0; k < nd; k++) { S.r0[3*is + k] = sr[k]; S.v0[3*is + k] = sv[k]; S.a0[3*is + k] = sa[k]; }
    for (int k = 0; k < nd; k++) { dr[k] = rold[k] - sr[k]; dv[k] = vold[k] - sv[k]; }
    if (nd == 3) {
      sk.angmom[0] += mold*(dr[1]*dv[2] - dr[2]*dv[1]);
      sk.angmom[1] += mold*(dr[2]*dv[0] - dr[0]*dv[2]);
      sk.angmom[2] += mold*(dr[0]*dv[1] - dr[1]*dv[0]);
    }
    else if (nd == 2) sk.angmom[2] += mold*(dr[0]*dv[1] - dr[1]*dv[0]);
    for (int j = 0; j < Nneib; j++) {                                  // Sinks.cpp:685-729
      const double *p = &rows[il[j]*SK_ROW];
      const double pm = mnew[il[j]];
      double mtemp = std::min(pm, macc);
      dt = p[D_DT];
      if (ctx->cfg.smooth_accretion == 0 || pm - mtemp < ctx->cfg.smooth_accrete_frac*ctx->mmean || dt < ctx->cfg.smooth_accrete_dt*sk.trot) { mtemp = pm; mnew[il[j]] = 0.0; }
      else mnew[il[j]] = pm - mtemp;                                   // Sph::AccreteMassFromParticle, Sph.h:108
      macc -= mtemp;
      for (int k = 0; k < nd; k++) { dr[k] = p[D_RX + k] - sr[k]; dv[k] = p[D_VX + k] - sv[k]; }
      if (nd == 3) {
        sk.angmom[0] += mtemp*(dr[1]*dv[2] - dr[2]*dv[1]);
        sk.angmom[1] += mtemp*(dr[2]*dv[0] - dr[0]*dv[2]);
        sk.angmom[2] += mtemp*(dr[0]*dv[1] - dr[1]*dv[0]);
      }
      else if (nd == 2) sk.angmom[2] += mtemp*(dr[0]*dv[1] - dr[1]*dv[0]);
      if (macc < small_number) break;
    }
    const double asqd = dot3(sa, sa, nd);
    S.dti[is] = 0.4*sqrt(sk.radius/(sqrt(asqd) + small_number));
  }
  // masses and dead flags back to the device (this rank's particles)
  {
    std::vector<int> aidx; std::vector<double> am;
    for (size_t e = 0; e < who.size(); e++) if (who[e].first == ctx->rank) { aidx.push_back(who[e].second); am.push_back(mnew[e]); }
    if (!aidx.empty()) {
      if ((rc = sk_reserve(ctx, W, ns, aidx.size() + 16, aidx.size()))) return rc;
      GH_CHECK(ctx, hipMemcpyAsync(W.d_i, aidx.data(), sizeof(int)*aidx.size(), hipMemcpyHostToDevice, ctx->stream));
      GH_CHECK(ctx, hipMemcpyAsync(W.d_rows, am.data(), sizeof(double)*aidx.size(), hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_sink_apply, dim3(cdiv(aidx.size(), 64)), dim3(64), 0, ctx->stream, gh_dev_own(ctx), W.d_i, W.d_rows, (int) aidx.size());
      GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  if (!lnb_idx.empty()) {
    const size_t nl = lnb_idx.size();
    if ((rc = sk_reserve(ctx, W, ns, 2*nl + 16, 1))) return rc;
    GH_CHECK(ctx, hipMemcpyAsync(W.d_i, lnb_idx.data(), sizeof(int)*nl, hipMemcpyHostToDevice, ctx->stream));
    GH_CHECK(ctx, hipMemcpyAsync(W.d_i + nl, lnb_lvl.data(), sizeof(int)*nl, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_sink_levelneib, dim3(cdiv(nl, 64)), dim3(64), 0, ctx->stream, gh_dev_own(ctx), W.d_i, W.d_i + nl, (int) nl);
    GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return GH_OK;
}

// the sink part of MainLoop (SphSimulation.cpp:820-838).  S = the stars after their correction terms; it comes back with the
// new sinks appended and the accreting sinks' r, v, a, m, r0, v0, a0 and dt_internal updated.
int gh_sinks_step(gh_ctx *ctx, gh_host_stars &S, double t, double timestep)
{
  if (!ctx->cfg.sink_particles) return GH_OK;
  SinkScratch &W = sk_scratch(ctx);
  int rc;
  if (ctx->cfg.create_sinks == 1 && (rc = sk_search(ctx, W, S, t))) return rc;
  if (ctx->sinks.empty()) return GH_OK;
  // mmean: masses summed in slot order, dead particles (m = 0) included in the count (SphSimulation.cpp:826-828).
  // Several ranks: every rank sums its own particles (tree order), the partial sums are added in rank order - mmean only
  // enters the "what is left is less than smooth_accrete_frac*mmean" test of the smooth accretion
  if (ctx->nranks == 1) {
    // (the masses are put into slot order on the device and come over in one pinned copy; the sum itself stays the
    //  reference's serial loop - through gh_download, with its pageable copies and the scatter on the host, this was ~12 ms
    //  of a 2 000 000-particle sink step)
    const size_t n = (size_t) ctx->N;
    if (W.cap_slot < n) {
      if (W.d_slot) (void) hipFree(W.d_slot);
      if (W.h_slot) (void) hipHostFree(W.h_slot);
      W.d_slot = nullptr; W.h_slot = nullptr; W.cap_slot = 0;
      GH_CHECK(ctx, hipMalloc((void**) &W.d_slot, sizeof(double)*(n + n/8 + 64)));
      GH_CHECK(ctx, hipHostMalloc((void**) &W.h_slot, sizeof(double)*(n + n/8 + 64)));
      W.cap_slot = n + n/8 + 64;
    }
    hipLaunchKernelGGL(k_slot_order, dim3(cdiv(n, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), D_M, W.d_slot);
    GH_CHECK(ctx, hipMemcpyAsync(W.h_slot, W.d_slot, sizeof(double)*n, hipMemcpyDeviceToHost, ctx->stream));
    GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (size_t i = 0; i < n; i++) sum += W.h_slot[i];
    ctx->mmean = sum/(double) ctx->N;
  }
  else {
    std::vector<double> m((size_t) ctx->own_count);
    GH_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    GH_CHECK(ctx, hipMemcpy(m.data(), ctx->fbuf[ctx->cur][D_M] + ctx->own_first, sizeof(double)*m.size(), hipMemcpyDeviceToHost));
    double part = 0.0;
    for (size_t i = 0; i < m.size(); i++) part += m[i];
    std::vector<char> all; std::vector<size_t> sizes;
    if ((rc = gh_dd_gatherv(ctx, &part, sizeof(double), all, sizes))) return rc;
    double sum = 0.0;
    for (int r = 0; r < ctx->nranks; r++) sum += ((const double*) all.data())[r];
    ctx->mmean = sum/(double) ctx->N;
  }
  return sk_accrete(ctx, W, S, timestep);
}

extern "C" int gh_get_sinks(gh_ctx *ctx, int *nsinks, double *rec, int *irec)
{
  if (!ctx || !nsinks) return GH_ERR_INVALID;
  *nsinks = (int) ctx->sinks.size();
  for (size_t s = 0; s < ctx->sinks.size(); s++) {
    const gh_sink_rec &k = ctx->sinks[s];
    if (rec) {
      const double v[GH_SINK_NREC] = {k.radius, k.mmax, k.menc, k.dmdt, k.ketot, k.gpetot, k.rotketot, k.utot, k.taccrete, k.trad, k.trot, k.tvisc,
                                      k.angmom[0], k.angmom[1], k.angmom[2], k.invh, ctx->mmean};
      for (int q = 0; q < GH_SINK_NREC; q++) rec[GH_SINK_NREC*s + q] = v[q];
    }
    if (irec) { irec[2*s] = k.istar; irec[2*s + 1] = k.Ngas; }
  }
  return GH_OK;
}
