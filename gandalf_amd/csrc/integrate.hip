// integrate.hip -- leapfrog kick-drift-kick glue and the global timestep, as streaming kernels.
//
// Replaces SphLeapfrogKDK::AdvanceParticles / EndTimestep, TimeIntegration::CheckBoundaries,
// SphIntegration::Timestep, Simulation::ComputeGlobalTimestep and Sph::ZeroAccelerations
// (reference src/Hydrodynamics/SphLeapfrogKDK.cpp:76-127, 219-272, src/Common/Integration.cpp,
//  src/Hydrodynamics/SphIntegration.cpp:81-134, src/Common/Simulation.cpp:1669-1754,
//  src/Hydrodynamics/Sph.cpp:126-140) for Nlevels = 1 (every particle active every step).
// The simulation time and timestep live in device memory (ctx->d_time) so that a run of steps can be
// enqueued without a host round trip.
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

__global__ void k_zero_acc(DevicePtrs d)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
  d.f[D_DIV_V][i] = 0.0; d.f[D_DUDT][i] = 0.0; d.f[D_GPOT][i] = 0.0; d.f[D_GPOT_HYDRO][i] = 0.0;
  for (int k = 0; k < 3; k++) { d.f[D_AX + k][i] = 0.0; d.f[D_ATX + k][i] = 0.0; }
}

// time[0] = t, time[1] = timestep.  advance: t <- t + timestep first (SphSimulation.cpp:587)
__global__ void k_advance_time(double *time) { time[0] = time[0] + time[1]; }

__global__ void k_kdk_advance(DevicePtrs d, Domain dom, const double *time, int energy_integration, int tdavisc)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
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
}

struct TimestepParams { double courant_mult, accel_mult, energy_mult; int energy_integration, hydro_forces; };

__global__ void k_timestep_partial(DevicePtrs d, TimestepParams tp, double *partial)
{
  __shared__ double s[256];
  double dtmin = 9.9e50;                                   // big_number_dp
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < d.N; i += gridDim.x*blockDim.x) {
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
    dtmin = fmin(dtmin, ts);
  }
  s[threadIdx.x] = dtmin;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
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

__global__ void k_kdk_end(DevicePtrs d, const double *time, int energy_integration)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= d.N) return;
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
  hipLaunchKernelGGL(k_zero_acc, dim3(cdiv(ctx->N, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx));
  return GH_OK;
}

extern double *gh_time_dev(gh_ctx *ctx);

int gh_kdk_advance_impl(gh_ctx *ctx, int, double, double)
{
  Domain dom;
  gh_fill_domain(ctx, dom);
  hipLaunchKernelGGL(k_kdk_advance, dim3(cdiv(ctx->N, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), dom,
                     gh_time_dev(ctx), ctx->cfg.energy_integration, ctx->cfg.avisc == GH_AVISC_MON97MM97 ? 1 : 0);
  return GH_OK;
}

int gh_advance_time_impl(gh_ctx *ctx)
{
  hipLaunchKernelGGL(k_advance_time, dim3(1), dim3(1), 0, ctx->stream, gh_time_dev(ctx));
  return GH_OK;
}

int gh_timestep_impl(gh_ctx *ctx)
{
  TimestepParams tp;
  tp.courant_mult = ctx->cfg.courant_mult; tp.accel_mult = ctx->cfg.accel_mult; tp.energy_mult = ctx->cfg.energy_mult;
  tp.energy_integration = ctx->cfg.energy_integration; tp.hydro_forces = ctx->cfg.hydro_forces;
  const int nblk = 256;
  hipLaunchKernelGGL(k_timestep_partial, dim3(nblk), dim3(256), 0, ctx->stream, gh_dev(ctx), tp, ctx->redbuf);
  hipLaunchKernelGGL(k_timestep_final, dim3(1), dim3(256), 0, ctx->stream, ctx->redbuf, nblk, gh_time_dev(ctx));
  hipLaunchKernelGGL(k_set_dt_next, dim3(cdiv(ctx->N, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), gh_time_dev(ctx));
  return GH_OK;
}

int gh_kdk_end_impl(gh_ctx *ctx, int, double, double)
{
  hipLaunchKernelGGL(k_kdk_end, dim3(cdiv(ctx->N, 256)), dim3(256), 0, ctx->stream, gh_dev(ctx), gh_time_dev(ctx),
                     ctx->cfg.energy_integration);
  return GH_OK;
}
