// integrate.hip -- leapfrog kick-drift-kick glue and the global timestep, as streaming kernels.
//
// Replaces SphLeapfrogKDK::AdvanceParticles / EndTimestep, TimeIntegration::CheckBoundaries,
// SphIntegration::Timestep, Simulation::ComputeGlobalTimestep and Sph::ZeroAccelerations
// (reference src/Hydrodynamics/SphLeapfrogKDK.cpp:76-127, 219-272, src/Common/Integration.cpp,
//  src/Hydrodynamics/SphIntegration.cpp:81-134, src/Common/Simulation.cpp:1669-1754,
//  src/Hydrodynamics/Sph.cpp:126-140); global timestep (Nlevels = 1) here, hierarchical block timesteps further down.
// The simulation time and timestep live in device memory (ctx->d_time) so that a run of steps can be
// enqueued without a host round trip.
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

__global__ void k_zero_acc(DevicePtrs d)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if (d.levels) {
    // block timesteps: active particles only (Sph.cpp:130).  The reference zeroes levelneib here and the particle's own
    // pair (always inside its kernel) raises it to its level again (GradhSph.cpp:445): start from the level
    if (!((int) d.f[D_FLAGS][i] & 1)) return;
    d.f[D_LEVELNEIB][i] = d.f[D_LEVEL][i];
  }
  d.f[D_DIV_V][i] = 0.0; d.f[D_DUDT][i] = 0.0; d.f[D_GPOT][i] = 0.0; d.f[D_GPOT_HYDRO][i] = 0.0;
  for (int k = 0; k < 3; k++) { d.f[D_AX + k][i] = 0.0; d.f[D_ATX + k][i] = 0.0; }
}

// time[0] = t, time[1] = timestep.  advance: t <- t + timestep first (SphSimulation.cpp:587)
__global__ void k_advance_time(double *time) { time[0] = time[0] + time[1]; }

__global__ void k_kdk_advance(DevicePtrs d, Domain dom, const double *time, int energy_integration, int tdavisc, int *blk)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  int nact = 0;
  if (i < d.N && !(d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD))) {      // dead particles stay where they are (SphLeapfrogKDK.cpp:99)
  const double t = time[0];
  const double dt = t - d.f[D_TLAST][i];
  for (int k = 0; k < d.ndim; k++) {
    const double r0 = d.f[D_R0X + k][i], v0 = d.f[D_V0X + k][i], a0 = d.f[D_A0X + k][i];
    double r = r0 + v0*dt + 0.5*a0*dt*dt;
    const double v = v0 + a0*dt;
    // TimeIntegration::CheckBoundaries: periodic wrap of r and r0
    double vv = v;
    if (dom.periodic[k]) {
      if (r < dom.bmin[k]) { r += dom.size[k]; d.f[D_R0X + k][i] = r0 + dom.size[k]; }
      if (r > dom.bmax[k]) { r -= dom.size[k]; d.f[D_R0X + k][i] = d.f[D_R0X + k][i] - dom.size[k]; }
    }
    // mirror walls: reflect r, r0 and flip v, v0, a, a0 (Integration.cpp:44-51, 66-73)
    if (dom.mirror[k][0] && r < dom.bmin[k]) {
      r = 2.0*dom.bmin[k] - r; d.f[D_R0X + k][i] = 2.0*dom.bmin[k] - d.f[D_R0X + k][i];
      vv = -vv; d.f[D_V0X + k][i] = -d.f[D_V0X + k][i]; d.f[D_AX + k][i] = -d.f[D_AX + k][i]; d.f[D_A0X + k][i] = -d.f[D_A0X + k][i];
    }
    if (dom.mirror[k][1] && r > dom.bmax[k]) {
      r = 2.0*dom.bmax[k] - r; d.f[D_R0X + k][i] = 2.0*dom.bmax[k] - d.f[D_R0X + k][i];
      vv = -vv; d.f[D_V0X + k][i] = -d.f[D_V0X + k][i]; d.f[D_AX + k][i] = -d.f[D_AX + k][i]; d.f[D_A0X + k][i] = -d.f[D_A0X + k][i];
    }
    d.f[D_RX + k][i] = r;
    d.f[D_VX + k][i] = vv;
  }
  if (tdavisc) d.f[D_ALPHA][i] += d.f[D_DALPHADT][i]*time[1];              // SphLeapfrogKDK.cpp:111 (the global timestep)
  if (energy_integration) d.f[D_U][i] = d.f[D_U0][i] + d.f[D_DUDT0][i]*dt;
  if (d.levels) {                                                         // SphLeapfrogKDK.cpp:117-118
    const int fl = (int) d.f[D_FLAGS][i];
    const bool active = blk[0] - (int) d.f[D_NLAST][i] == (int) d.f[D_NSTEP][i];
    d.f[D_FLAGS][i] = (double) (active ? (fl | 1) : (fl & ~1));
    nact = active ? 1 : 0;
  }
  }
  if (d.levels) {                                                         // running count of force evaluations (blk[8..9])
    const unsigned long long m = __ballot(nact);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd((unsigned long long*) (blk + 8), (unsigned long long) __popcll(m));
  }
}

struct TimestepParams { double courant_mult, accel_mult, energy_mult; int energy_integration, hydro_forces; };

// SphIntegration::Timestep, SphIntegration.cpp:81-134
__device__ __forceinline__ double particle_timestep(const DevicePtrs &d, const TimestepParams &tp, int i)
{
  const double h = d.f[D_H][i];
  const double divv = fabs(d.f[D_DIV_V][i]);
  double ts;
  if (tp.hydro_forces) ts = tp.courant_mult*h/(d.f[D_SOUND][i] + h*divv + GH_SMALL_DP);
  else ts = tp.courant_mult*h/(h*divv + GH_SMALL_DP);
  double a2 = 0.0;
  for (int k = 0; k < d.ndim; k++) { const double a = d.f[D_AX + k][i]; a2 += a*a; }
  const double amag = sqrt(a2);
  ts = fmin(ts, tp.accel_mult*sqrt(h/(amag + GH_SMALL_DP)));
  if (tp.energy_integration) ts = fmin(ts, tp.energy_mult*(d.f[D_U][i]/(fabs(d.f[D_DUDT][i]) + GH_SMALL)));
  return ts;
}

#define GH_TS_THREADS 1024      /* 256 blocks of 16 waves: one wave per SIMD left the kernel latency-bound */
__global__ __launch_bounds__(GH_TS_THREADS) void k_timestep_partial(DevicePtrs d, TimestepParams tp, double *partial)
{
  __shared__ double s[GH_TS_THREADS];
  double dtmin = 9.9e50;                                   // big_number_dp
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < d.N; i += gridDim.x*blockDim.x) {
    if (d.levels && d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD)) continue;      // Simulation.cpp:1811 (the global-timestep loop, :1696, has no such skip)
    const double ts = particle_timestep(d, tp, i);
    if (d.levels) d.f[D_DT_NEXT][i] = ts;                  // resynchronisation of the block structure, Simulation.cpp:1816
    dtmin = fmin(dtmin, ts);
  }
  s[threadIdx.x] = dtmin;
  __syncthreads();
  for (int off = GH_TS_THREADS/2; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) s[threadIdx.x] = fmin(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = s[0];
}

__global__ void k_timestep_final(const double *partial, int nblk, double *time)
{
  __shared__ double s[256];
  double v = 9.9e50;
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) v = fmin(v, partial[b]);
  s[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) s[threadIdx.x] = fmin(s[threadIdx.x], s[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) time[1] = s[0];
}

__global__ void k_set_dt_next(DevicePtrs d, const double *time)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < d.N) d.f[D_DT_NEXT][i] = time[1];
}

__global__ void k_kdk_end(DevicePtrs d, const double *time, int energy_integration, const int *blk)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if (d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD)) return;          // SphLeapfrogKDK.cpp:238
  if (d.levels) {                                          // only particles at the end of their step (SphLeapfrogKDK.cpp:241)
    if (!((int) d.f[D_FLAGS][i] & 2)) return;
    d.f[D_NLAST][i] = (double) blk[0];
    d.f[D_FLAGS][i] = 0.0;
  }
  const double dt = d.f[D_DT][i];
  for (int k = 0; k < d.ndim; k++) {
    const double a = d.f[D_AX + k][i];
    const double v = d.f[D_VX + k][i] + 0.5*dt*(a - d.f[D_A0X + k][i]);
    d.f[D_VX + k][i] = v;
    d.f[D_R0X + k][i] = d.f[D_RX + k][i];
    d.f[D_V0X + k][i] = v;
    d.f[D_A0X + k][i] = a;
  }
  if (energy_integration) {
    const double dudt = d.f[D_DUDT][i], dudt0 = d.f[D_DUDT0][i], u0 = d.f[D_U0][i];
    double u = d.f[D_U][i] + 0.5*(dudt - dudt0)*dt;
    if (u <= 0.0) u = u0 + dudt0*dt;
    d.f[D_U][i] = u;
    d.f[D_U0][i] = u;
    d.f[D_DUDT0][i] = dudt;
  }
  d.f[D_TLAST][i] = time[0];
  d.f[D_DT][i] = d.f[D_DT_NEXT][i];
  d.f[D_DT_NEXT][i] = 0.0;
}

int gh_zero_acc_impl(gh_ctx *ctx)
{
  hipLaunchKernelGGL(k_zero_acc, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx));
  return GH_OK;
}

extern double *gh_time_dev(gh_ctx *ctx);

int gh_kdk_advance_impl(gh_ctx *ctx, int, double, double)
{
  Domain dom;
  gh_fill_domain(ctx, dom);
  hipLaunchKernelGGL(k_kdk_advance, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), dom,
                     gh_time_dev(ctx), ctx->cfg.energy_integration,
                     (ctx->cfg.avisc == GH_AVISC_MON97MM97 || ctx->cfg.avisc == GH_AVISC_MON97CD2010) ? 1 : 0, ctx->d_blk);
  return GH_OK;
}

int gh_advance_time_impl(gh_ctx *ctx)
{
  hipLaunchKernelGGL(k_advance_time, dim3(1), dim3(1), 0, ctx->stream, gh_time_dev(ctx));
  return GH_OK;
}

int gh_timestep_impl_extra(gh_ctx *ctx, int nextra);
int gh_timestep_impl(gh_ctx *ctx) { return gh_timestep_impl_extra(ctx, 0); }

// nextra further candidates for the minimum sit behind the block minima in ctx->redbuf (the stars of a hybrid run)
int gh_timestep_impl_extra(gh_ctx *ctx, int nextra)
{
  TimestepParams tp;
  tp.courant_mult = ctx->cfg.courant_mult; tp.accel_mult = ctx->cfg.accel_mult; tp.energy_mult = ctx->cfg.energy_mult;
  tp.energy_integration = ctx->cfg.energy_integration; tp.hydro_forces = ctx->cfg.hydro_forces;
  const int nblk = 256;
  hipLaunchKernelGGL(k_timestep_partial, dim3(nblk), dim3(GH_TS_THREADS), 0, ctx->stream, gh_dev_own(ctx), tp, ctx->redbuf);
  hipLaunchKernelGGL(k_timestep_final, dim3(1), dim3(256), 0, ctx->stream, ctx->redbuf, nblk + nextra, gh_time_dev(ctx));
  { const int rc = gh_dd_min_dt(ctx); if (rc) return rc; }      // multi-GPU: minimum over the ranks (Simulation.cpp:1738)
  hipLaunchKernelGGL(k_set_dt_next, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), gh_time_dev(ctx));
  return GH_OK;
}

int gh_kdk_end_impl(gh_ctx *ctx, int, double, double)
{
  hipLaunchKernelGGL(k_kdk_end, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), gh_time_dev(ctx),
                     ctx->cfg.energy_integration, ctx->d_blk);
  return GH_OK;
}

// ================================================================================================
// hierarchical block timesteps (Nlevels > 1)
//   Simulation::ComputeBlockTimesteps (Simulation.cpp:1764-2200, hydro particles only),
//   SphLeapfrogKDK::CheckTimesteps (SphLeapfrogKDK.cpp:284-330) and the all-particle thermal refresh of the main loop
//   (SphSimulation.cpp:665-679).  The integer clock lives on the device:
//     blk = {n, nresync, level_max, level_step, level_max_new, activecount, nfactor_mul, nfactor_div};  time[2] = dt_max.
// ================================================================================================
// (int) pow(2, e) of the reference: 0 for e < 0 (a particle that moves to a level created in the same call gets
// nstep = 0 and therefore dt_next = 0 until the next pass refreshes nstep, Simulation.cpp:1989-1990 with :2141-2146)
__device__ __forceinline__ int ipow2(int e) { return e >= 0 ? (1 << e) : 0; }

enum { B_N = 0, B_NRESYNC, B_LMAX, B_LSTEP, B_LMAXNEW, B_ACTIVE, B_MUL, B_DIV, B_CNT0, B_CNT1, B_LMH, B_LMSTAR /* highest star level (sink runs) */ };   // B_CNT*: 64-bit active counter

// ComputeTimestepLevel, InlineFuncs.h:550-558
__device__ __forceinline__ int timestep_level(double dt, double dt_max)
{
  const int l = (int) (1.44269504088896*log(dt_max/dt)) + 1;
  return l > 0 ? l : 0;
}

__global__ void k_thermal_all(DevicePtrs d, EosParams eos)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  double u = d.f[D_U][i], sound, press;
  eos_eval(eos, d.f[D_RHO][i], u, sound, press);
  d.f[D_U][i] = u; d.f[D_SOUND][i] = sound; d.f[D_PRESSURE][i] = press;
}

// resynchronisation (:1795-1924): time[1] holds the minimum timestep found by k_timestep_partial/final
__global__ void k_block_resync_clock(int *blk, double *time, int Nlevels)
{
  blk[B_N] = 0;
  blk[B_LMAX] = Nlevels - 1;
  blk[B_LSTEP] = Nlevels - 1;                                // level_max + integration_step - 1, integration_step = 1 (lfkdk)
  time[2] = time[1]*(double) (1 << (Nlevels - 1));           // dt_max = timestep*2^level_max
  const int lmh = timestep_level(time[1], time[2]);          // level_max_hydro (:1857): the minimum timestep is a gas one
  blk[B_LMH] = lmh < Nlevels - 1 ? lmh : Nlevels - 1;
}
__global__ void k_block_resync_assign(DevicePtrs d, const int *blk, const double *time, int single)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if (d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD)) return;
  const int level_max = blk[B_LMAX], level_step = blk[B_LSTEP];
  int level = timestep_level(d.f[D_DT_NEXT][i], time[2]);
  level = level < level_max ? level : level_max;
  if (single) level = blk[B_LMH];                            // sph_single_timestep (:1890-1900)
  const int nstep = ipow2(level_step - level);
  d.f[D_LEVEL][i] = (double) level; d.f[D_LEVELNEIB][i] = (double) level;
  d.f[D_NSTEP][i] = (double) nstep; d.f[D_NLAST][i] = 0.0;
  d.f[D_DT_NEXT][i] = (double) nstep*time[1];                // with the minimum timestep, :1913
  d.f[D_FLAGS][i] = (double) ((int) d.f[D_FLAGS][i] | 2);
}
__global__ void k_block_resync_finish(int *blk, double *time)
{
  blk[B_NRESYNC] = 1 << blk[B_LSTEP];
  time[1] = time[2]/(double) blk[B_NRESYNC];
}

// no resynchronisation (:1929-2150), pass 1: particles that end their step pick their new level
__global__ void k_block_levels(DevicePtrs d, TimestepParams tp, int *blk, const double *time, int level_diff_max)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  int mylevel = 0;
  if (i < d.N && !(d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD))) {
    const int n = blk[B_N], level_step = blk[B_LSTEP];
    const int nlast = (int) d.f[D_NLAST][i], nstep = (int) d.f[D_NSTEP][i];
    int level_i = (int) d.f[D_LEVEL][i];
    if (n - nlast == nstep) {
      const int levelneib = (int) d.f[D_LEVELNEIB][i];
      const double dt = particle_timestep(d, tp, i);
      int level = timestep_level(dt, time[2]);
      level = max(level, levelneib - level_diff_max);
      if (nstep != ipow2(level_step - level_i)) {          // step cut short by CheckTimesteps (:1956-1966)
        level_i = max(level_i, level);
        d.f[D_LEVELNEIB][i] = (double) level_i;
      }
      else {                                                 // natural end of the step (:1968-1992)
        const int last_level = level_i;
        if (level < last_level && last_level > 1 && n%(2*nstep) == 0) level_i = last_level - 1;
        else if (level > last_level) level_i = level;
        d.f[D_LEVELNEIB][i] = (double) level;
      }
      const int ns = ipow2(level_step - level_i);
      d.f[D_LEVEL][i] = (double) level_i;
      d.f[D_NLAST][i] = (double) n;
      d.f[D_NSTEP][i] = (double) ns;
      d.f[D_DT_NEXT][i] = (double) ns*time[1];
      d.f[D_FLAGS][i] = (double) ((int) d.f[D_FLAGS][i] | 2);
    }
    mylevel = level_i;
  }
  int wm = mylevel;
  for (int off = 32; off > 0; off >>= 1) wm = max(wm, __shfl_xor(wm, off));
  if ((threadIdx.x & 63) == 0) atomicMax(&blk[B_LMAXNEW], wm);
}
// pass 2: the clock (:2097-2136)
__global__ void k_block_clock(int *blk, double *time)
{
  const int level_max_old = blk[B_LMAX], level_step_old = blk[B_LSTEP];
  int level_max = max(blk[B_LMAXNEW], blk[B_LMSTAR]), n = blk[B_N];      // over gas and stars (:2058, 2068)
  blk[B_LMSTAR] = 0;
  const int istep = 1 << (level_step_old - level_max_old + 1);
  int mul = 1, div = 1;
  if (level_max > level_max_old) { mul = 1 << (level_max - level_max_old); n *= mul; }
  else if (level_max <= level_max_old - 1 && level_max_old > 1 && n%istep == 0) { level_max = level_max_old - 1; div = 2; n /= 2; }
  else level_max = level_max_old;
  blk[B_N] = n; blk[B_LMAX] = level_max; blk[B_LSTEP] = level_max;
  blk[B_NRESYNC] = 1 << level_max;
  blk[B_MUL] = mul; blk[B_DIV] = div; blk[B_LMH] = blk[B_LMAXNEW]; blk[B_LMAXNEW] = 0;   // hydro only: level_max_hydro = max level
  time[1] = time[2]/(double) blk[B_NRESYNC];
}
// pass 3: rescale the integer times, refresh nstep of the particles that just ended their step (:2101-2146)
__global__ void k_block_rescale(DevicePtrs d, const int *blk, int single)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  if (d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD)) return;
  const int mul = blk[B_MUL], div = blk[B_DIV];
  int nstep = (int) d.f[D_NSTEP][i], nlast = (int) d.f[D_NLAST][i];
  nstep = nstep*mul/div; nlast = nlast*mul/div;
  if (nlast == blk[B_N]) {
    if (single) d.f[D_LEVEL][i] = (double) blk[B_LMH];      // sph_single_timestep (:2090-2096)
    nstep = ipow2(blk[B_LSTEP] - (int) d.f[D_LEVEL][i]);
  }
  d.f[D_NSTEP][i] = (double) nstep; d.f[D_NLAST][i] = (double) nlast;
}

// all active flags off, then CheckTimesteps: particles whose neighbours run on much shorter steps end theirs early
__global__ void k_check_timesteps(DevicePtrs d, int *blk, int level_diff_max)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  int woke = 0;
  if (i < d.N && !(d.sinks && ((int) d.f[D_FLAGS][i] & GH_FLAG_DEAD))) {      // SphLeapfrogKDK.cpp:307
    int fl = (int) d.f[D_FLAGS][i] & ~1;
    const int dn = blk[B_N] - (int) d.f[D_NLAST][i];
    const int level = (int) d.f[D_LEVEL][i], levelneib = (int) d.f[D_LEVELNEIB][i];
    if (dn != (int) d.f[D_NSTEP][i] && levelneib - level > level_diff_max) {
      const int level_new = levelneib - level_diff_max;
      const int nnewstep = ipow2(blk[B_LSTEP] - level_new);
      if (dn%nnewstep == 0) {
        if (dn > 0) d.f[D_NSTEP][i] = (double) dn;
        d.f[D_LEVEL][i] = (double) level_new;
        fl |= 1; woke = 1;
      }
    }
    d.f[D_FLAGS][i] = (double) fl;
  }
  const unsigned long long m = __ballot(woke);
  if ((threadIdx.x & 63) == 0 && m) { atomicAdd(&blk[B_ACTIVE], __popcll(m)); atomicAdd((unsigned long long*) (blk + 8), (unsigned long long) __popcll(m)); }
}

static TimestepParams fill_tp(gh_ctx *ctx)
{
  TimestepParams tp;
  tp.courant_mult = ctx->cfg.courant_mult; tp.accel_mult = ctx->cfg.accel_mult; tp.energy_mult = ctx->cfg.energy_mult;
  tp.energy_integration = ctx->cfg.energy_integration; tp.hydro_forces = ctx->cfg.hydro_forces;
  return tp;
}

int gh_thermal_all_impl(gh_ctx *ctx)
{
  EosParams eos;
  gh_fill_eos(ctx, eos);
  hipLaunchKernelGGL(k_thermal_all, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), eos);
  return GH_OK;
}

// Simulation::ComputeBlockTimesteps; the host knows n and nresync (read back after every step)
int gh_block_timesteps_impl(gh_ctx *ctx)
{
  hipStream_t s = ctx->stream;
  const int nb = cdiv(ctx->own_count, 256);
  DevicePtrs d = gh_dev_own(ctx);
  double *time = gh_time_dev(ctx);
  if (ctx->n == ctx->nresync) {
    const int nblk = 256;
    hipLaunchKernelGGL(k_timestep_partial, dim3(nblk), dim3(GH_TS_THREADS), 0, s, d, fill_tp(ctx), ctx->redbuf);
    hipLaunchKernelGGL(k_timestep_final, dim3(1), dim3(256), 0, s, ctx->redbuf, nblk, time);
    { const int rc = gh_dd_min_dt(ctx); if (rc) return rc; }            // multi-GPU: the minimum over all ranks (Simulation.cpp:1843-1847)
    hipLaunchKernelGGL(k_block_resync_clock, dim3(1), dim3(1), 0, s, ctx->d_blk, time, ctx->cfg.Nlevels);
    hipLaunchKernelGGL(k_block_resync_assign, dim3(nb), dim3(256), 0, s, d, ctx->d_blk, time, ctx->cfg.sph_single_timestep);
    hipLaunchKernelGGL(k_block_resync_finish, dim3(1), dim3(1), 0, s, ctx->d_blk, time);
  }
  else {
    hipLaunchKernelGGL(k_block_levels, dim3(nb), dim3(256), 0, s, d, fill_tp(ctx), ctx->d_blk, time, ctx->cfg.level_diff_max);
    { const int rc = gh_dd_reduce_int(ctx, ctx->d_blk + B_LMAXNEW, 0); if (rc) return rc; }   // highest occupied level of all ranks (:2016-2080)
    hipLaunchKernelGGL(k_block_clock, dim3(1), dim3(1), 0, s, ctx->d_blk, time);
    hipLaunchKernelGGL(k_block_rescale, dim3(nb), dim3(256), 0, s, d, ctx->d_blk, ctx->cfg.sph_single_timestep);
  }
  return GH_OK;
}

// Simulation::ComputeBlockTimesteps of a sink run: the gas as above, the stars (a handful, host mirror S) with the star
// branches of the reference (Simulation.cpp:1820-1873, 2024-2060, 2111-2150) - they sit on levels >= the highest gas level
static inline int host_timestep_level(double dt, double dt_max) { const int l = (int) (1.44269504088896*log(dt_max/dt)) + 1; return l > 0 ? l : 0; }
static inline int host_ipow2(int e) { return (int) pow(2.0, e); }
int gh_block_timesteps_hybrid(gh_ctx *ctx, gh_host_stars &S, double nbody_mult)
{
  hipStream_t s = ctx->stream;
  const int nb = cdiv(ctx->own_count, 256);
  DevicePtrs d = gh_dev_own(ctx);
  double *time = gh_time_dev(ctx);
  const int ns = (int) S.n;
  auto star_dt = [&](int i) {                               // NbodyLeapfrogKDK::Timestep, :387-400
    const double amag = sqrt(S.a[3*i]*S.a[3*i] + S.a[3*i + 1]*S.a[3*i + 1] + S.a[3*i + 2]*S.a[3*i + 2]);
    return std::min(nbody_mult*sqrt(S.h[i]/(amag + GH_SMALL_DP)), S.dti[i]);
  };
  double tt[3];
  int blk[12];
  if (ctx->n == ctx->nresync) {
    const int nblk = 256;
    hipLaunchKernelGGL(k_timestep_partial, dim3(nblk), dim3(GH_TS_THREADS), 0, s, d, fill_tp(ctx), ctx->redbuf);
    hipLaunchKernelGGL(k_timestep_final, dim3(1), dim3(256), 0, s, ctx->redbuf, nblk, time);
    { const int rc = gh_dd_min_dt(ctx); if (rc) return rc; }            // several ranks: the gas minimum of all of them (the stars are everybody's)
    GH_CHECK(ctx, hipMemcpyAsync(tt, time, sizeof(tt), hipMemcpyDeviceToHost, s));
    GH_CHECK(ctx, hipStreamSynchronize(s));
    const double dt_min_hydro = tt[1];
    double timestep = dt_min_hydro;
    std::vector<double> sdt((size_t) ns);
    for (int i = 0; i < ns; i++) { sdt[i] = star_dt(i); timestep = std::min(timestep, sdt[i]); }
    GH_CHECK(ctx, hipMemcpyAsync(time + 1, &timestep, sizeof(double), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_block_resync_clock, dim3(1), dim3(1), 0, s, ctx->d_blk, time, ctx->cfg.Nlevels);
    const int level_max = ctx->cfg.Nlevels - 1, level_step = level_max;
    const double dt_max = timestep*pow(2.0, level_max);
    const int lmh = std::min(host_timestep_level(dt_min_hydro, dt_max), level_max);
    GH_CHECK(ctx, hipMemcpyAsync(ctx->d_blk + B_LMH, &lmh, sizeof(int), hipMemcpyHostToDevice, s));
    for (int i = 0; i < ns; i++) {
      const int level = std::min(host_timestep_level(sdt[i], dt_max), level_max);
      S.level[i] = std::max(level, lmh);
      S.nlast[i] = 0; S.nstep[i] = host_ipow2(level_step - S.level[i]); S.tlast[i] = tt[0]; S.endflag[i] = 1;
    }
    hipLaunchKernelGGL(k_block_resync_assign, dim3(nb), dim3(256), 0, s, d, ctx->d_blk, time, ctx->cfg.sph_single_timestep);
    hipLaunchKernelGGL(k_block_resync_finish, dim3(1), dim3(1), 0, s, ctx->d_blk, time);
    GH_CHECK(ctx, hipStreamSynchronize(s));                  // timestep / lmh live on this stack
    return GH_OK;
  }
  hipLaunchKernelGGL(k_block_levels, dim3(nb), dim3(256), 0, s, d, fill_tp(ctx), ctx->d_blk, time, ctx->cfg.level_diff_max);
  { const int rc = gh_dd_reduce_int(ctx, ctx->d_blk + B_LMAXNEW, 0); if (rc) return rc; }     // highest occupied gas level of all ranks
  GH_CHECK(ctx, hipMemcpyAsync(blk, ctx->d_blk, sizeof(blk), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipMemcpyAsync(tt, time, sizeof(tt), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  const int n = blk[B_N], level_step_old = blk[B_LSTEP], lmh = blk[B_LMAXNEW];
  const double dt_max = tt[2];
  int lmstar = 0;
  for (int i = 0; i < ns; i++) {
    if (n - S.nlast[i] == S.nstep[i]) {
      const int nstep = S.nstep[i], last_level = S.level[i];
      const int level = std::max(host_timestep_level(star_dt(i), dt_max), lmh);
      if (level < last_level && level > lmh && last_level > 1 && n%(2*nstep) == 0) S.level[i] = last_level - 1;
      else if (level > last_level) S.level[i] = level;
      S.nlast[i] = n; S.nstep[i] = host_ipow2(level_step_old - S.level[i]); S.tlast[i] = tt[0]; S.endflag[i] = 1;
    }
    lmstar = std::max(lmstar, S.level[i]);
  }
  GH_CHECK(ctx, hipMemcpyAsync(ctx->d_blk + B_LMSTAR, &lmstar, sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_block_clock, dim3(1), dim3(1), 0, s, ctx->d_blk, time);
  hipLaunchKernelGGL(k_block_rescale, dim3(nb), dim3(256), 0, s, d, ctx->d_blk, ctx->cfg.sph_single_timestep);
  GH_CHECK(ctx, hipMemcpyAsync(blk, ctx->d_blk, sizeof(blk), hipMemcpyDeviceToHost, s));
  GH_CHECK(ctx, hipStreamSynchronize(s));
  const int mul = blk[B_MUL], div = blk[B_DIV];
  for (int i = 0; i < ns; i++) {
    S.nstep[i] = S.nstep[i]*mul/div; S.nlast[i] = S.nlast[i]*mul/div;
    if (S.nlast[i] == blk[B_N]) S.nstep[i] = host_ipow2(blk[B_LSTEP] - S.level[i]);
  }
  return GH_OK;
}

int gh_check_timesteps_impl(gh_ctx *ctx)
{
  hipLaunchKernelGGL(k_check_timesteps, dim3(cdiv(ctx->own_count, 256)), dim3(256), 0, ctx->stream, gh_dev_own(ctx), ctx->d_blk, ctx->cfg.level_diff_max);
  return gh_dd_reduce_int(ctx, ctx->d_blk + B_ACTIVE, 1);               // multi-GPU: particles woken on any rank repeat the passes on all (SphSimulation.cpp:753)
}
