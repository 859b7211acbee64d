// nbody.hip -- star-star direct summation and the stars' leapfrog KDK on the GPU.
//
// Replaces Nbody<ndim>::CalculateDirectGravForces (reference src/Nbody/Nbody.cpp:233-287),
// NbodyLeapfrogKDK::CalculateDirectSmoothedGravForces (src/Nbody/NbodyLeapfrogKDK.cpp:78-142) and the
// elementwise integrator members (:253-400).  All pairs, no tree: N is a few thousand at most in GANDALF
// runs (sinks / stars), so the kernel is organised for latency, not bandwidth: a workgroup owns 64 target
// stars and splits the source loop over its 4 wavefronts (lane = target, wave = source slice); source
// tiles of 64 stars go through LDS (8 doubles per star) and are read as broadcasts.  Each lane sums its
// slice in source order, the four slices are added through LDS.  Jerk (adot) is summed with the forces.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <algorithm>
#include "../../include/gandalf_hip.h"
#include "sph_kernels.hpp"

#define NB_SMALL 1.0e-20
#define NB_SMALL_DP 1.0e-50
#define NB_BIG 9.9e20
#define NB_TWOPI 6.28318530717959       /* reference Constants.h:62 */

struct gh_nbody {
  int ndim = 3, softening = 0, device = 0;
  double nbody_mult = 0.1;
  int64_t N = 0, Ncap = 0;
  hipStream_t stream = nullptr;
  // SoA, component-major: r[k*N + i]
  double *r = nullptr, *v = nullptr, *a = nullptr, *adot = nullptr, *r0 = nullptr, *v0 = nullptr, *a0 = nullptr;
  double *m = nullptr, *h = nullptr, *gpot = nullptr, *tlast = nullptr;
  bool act_on = false;
  int *act = nullptr;               // sink runs on the block-timestep ladder: 1 = the star ends its step now (NbodyParticle active flag); nullptr = all
  std::vector<int> h_level, h_nstep, h_nlast;   // ... and the stars' level / nstep / nlast (host: a handful of stars)
  double *dti = nullptr;            // dt_internal (Sinks.cpp:735: an accreting sink limits its own timestep), else big_number
  double *tdt = nullptr;            // device {t, timestep, scratch min}
  double *red = nullptr;            // block minima
  double *stage = nullptr;          // host<->device transposition buffer
  std::string err;
};

#define NB_CHECK(nb, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (nb)->err = std::string(#call) + ": " + hipGetErrorString(e__); return GH_ERR_HIP; } } while (0)

struct NbPtrs { double *r, *v, *a, *adot, *r0, *v0, *a0, *m, *h, *gpot, *tlast, *tdt, *dti; const int *act; int N, ndim; };

static NbPtrs nb_ptrs(gh_nbody *nb)
{
  NbPtrs p = {nb->r, nb->v, nb->a, nb->adot, nb->r0, nb->v0, nb->a0, nb->m, nb->h, nb->gpot, nb->tlast, nb->tdt, nb->dti, nb->act_on ? nb->act : nullptr, (int) nb->N, nb->ndim};
  return p;
}

// powf(invhmean, ndim) as the reference evaluates it (NbodyLeapfrogKDK.cpp:118): single precision
__device__ __forceinline__ double powf_ref(double x, int nd)
{
  const float xf = (float) x;
  const double xd = (double) xf;
  const float p = nd == 1 ? xf : (nd == 2 ? (float) (xd*xd) : (float) (xd*xd*xd));
  return (double) p;
}

template <int ND, bool SOFT>
__global__ __launch_bounds__(256) void k_nbody_forces(NbPtrs p)
{
  typedef M4<ND> K;
  __shared__ double s_src[4][8][64];          // per wave: x,y,z,vx,vy,vz,m,h of a 64-star tile
  __shared__ double s_acc[4][7][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x*64 + lane;
  const bool live = i < p.N;
  double ri[3] = {0, 0, 0}, vi[3] = {0, 0, 0}, hi = 1.0;
  if (live) { for (int k = 0; k < ND; k++) { ri[k] = p.r[(size_t) k*p.N + i]; vi[k] = p.v[(size_t) k*p.N + i]; } hi = p.h[i]; }
  double a[3] = {0, 0, 0}, ad[3] = {0, 0, 0}, gp = 0.0;
  const int ntile = (p.N + 63)/64;
  // wave w takes a contiguous quarter of the tiles, so that every partial sum runs in source order
  const int per = (ntile + 3)/4;
  const int t0 = wave*per, t1 = min(ntile, t0 + per);
  for (int t = t0; t < t1; t++) {
    const int j = t*64 + lane;
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 1.0};
    if (j < p.N) {
      for (int k = 0; k < ND; k++) { q[k] = p.r[(size_t) k*p.N + j]; q[3 + k] = p.v[(size_t) k*p.N + j]; }
      q[6] = p.m[j]; q[7] = p.h[j];
    }
    for (int c = 0; c < 8; c++) s_src[wave][c][lane] = q[c];
    // a wave only reads its own tile: no workgroup barrier needed (LDS operations of one wave are in order)
    const int cnt = min(64, p.N - t*64);
    for (int jj = 0; jj < cnt; jj++) {
      if (t*64 + jj == i) continue;                                            // i == j
      double dr[3] = {0, 0, 0}, dv[3] = {0, 0, 0};
      for (int k = 0; k < ND; k++) { dr[k] = s_src[wave][k][jj] - ri[k]; dv[k] = s_src[wave][3 + k][jj] - vi[k]; }
      const double mj = s_src[wave][6][jj];
      double drsqd = dr[0]*dr[0];
      if (ND > 1) drsqd += dr[1]*dr[1];
      if (ND > 2) drsqd += dr[2]*dr[2];
      double dvdr = dv[0]*dr[0];
      if (ND > 1) dvdr += dv[1]*dr[1];
      if (ND > 2) dvdr += dv[2]*dr[2];
      if (!SOFT) {                                                             // Nbody.cpp:264-272
        const double invdrmag = 1.0/sqrt(drsqd);
        const double drdt = dvdr*invdrmag;
        const double inv3 = invdrmag*invdrmag*invdrmag;
        gp += mj*invdrmag;
        for (int k = 0; k < ND; k++) a[k] += mj*dr[k]*inv3;
        for (int k = 0; k < ND; k++) ad[k] += mj*inv3*(dv[k] - 3.0*drdt*invdrmag*dr[k]);
      }
      else {                                                                   // NbodyLeapfrogKDK.cpp:111-126
        const double hj = s_src[wave][7][jj];
        const double drmag = sqrt(drsqd) + NB_SMALL;
        const double invdrmag = 1.0/drmag;
        const double invhmean = 2.0/(hi + hj);
        const double drdt = dvdr*invdrmag;
        const double paux = mj*invhmean*invhmean*K::wgrav(drmag*invhmean)*invdrmag;
        const double wmean = K::w0(drmag*invhmean)*powf_ref(invhmean, ND);
        gp += mj*invhmean*K::wpot(drmag*invhmean);
        for (int k = 0; k < ND; k++) a[k] += paux*dr[k];
        for (int k = 0; k < ND; k++)
          ad[k] += paux*dv[k] - 3.0*paux*drdt*invdrmag*dr[k] + 2.0*NB_TWOPI*mj*drdt*wmean*invdrmag*dr[k];
      }
    }
  }
  for (int k = 0; k < 3; k++) { s_acc[wave][k][lane] = a[k]; s_acc[wave][3 + k][lane] = ad[k]; }
  s_acc[wave][6][lane] = gp;
  __syncthreads();
  if (wave == 0 && live && (!p.act || p.act[i])) {
    for (int k = 0; k < ND; k++) {
      p.a[(size_t) k*p.N + i] = ((s_acc[0][k][lane] + s_acc[1][k][lane]) + s_acc[2][k][lane]) + s_acc[3][k][lane];
      p.adot[(size_t) k*p.N + i] = ((s_acc[0][3 + k][lane] + s_acc[1][3 + k][lane]) + s_acc[2][3 + k][lane]) + s_acc[3][3 + k][lane];
    }
    p.gpot[i] = ((s_acc[0][6][lane] + s_acc[1][6][lane]) + s_acc[2][6][lane]) + s_acc[3][6][lane];
  }
}

// t += timestep, then NbodyLeapfrogKDK::AdvanceParticles (:253-291)
__global__ void k_nbody_advance(NbPtrs p, int first_thread_advances_clock)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  const double t = p.tdt[0] + p.tdt[1];
  if (i < p.N) {
    const double dt = t - p.tlast[i];
    for (int k = 0; k < p.ndim; k++) {
      const size_t o = (size_t) k*p.N + i;
      p.r[o] = p.r0[o] + p.v0[o]*dt + 0.5*p.a0[o]*dt*dt;
      p.v[o] = p.v0[o] + p.a0[o]*dt;
    }
  }
}
__global__ void k_nbody_clock(NbPtrs p) { p.tdt[0] = p.tdt[0] + p.tdt[1]; }

// CorrectionTerms (:301-331) and the per-star timestep (:387-400) with a block min
__global__ __launch_bounds__(256) void k_nbody_correct_dt(NbPtrs p, double nbody_mult, double *red, int correct)
{
  __shared__ double s_min[256];
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  double ts = 9.9e50;
  if (i < p.N) {
    double amag2 = 0.0;
    for (int k = 0; k < p.ndim; k++) {
      const size_t o = (size_t) k*p.N + i;
      const double ak = p.a[o];
      if (correct && (!p.act || p.act[i])) p.v[o] += 0.5*(ak - p.a0[o])*(p.tdt[0] - p.tlast[i]);
      amag2 += ak*ak;
    }
    const double amag = sqrt(amag2);
    ts = nbody_mult*sqrt(p.h[i]/(amag + NB_SMALL_DP));
    ts = fmin(ts, p.dti[i]);                                // dt_internal: big_number, or what the sink accretion set
  }
  s_min[threadIdx.x] = ts;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + off]);
    __syncthreads();
  }
  if (threadIdx.x == 0) red[blockIdx.x] = s_min[0];
}

// min over blocks -> timestep, then EndTimestep (:341-377)
__global__ __launch_bounds__(256) void k_nbody_end(NbPtrs p, const double *red, int nblk)
{
  __shared__ double s_min[256];
  double mn = 9.9e50;
  for (int b = threadIdx.x; b < nblk; b += 256) mn = fmin(mn, red[b]);
  s_min[threadIdx.x] = mn;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int) threadIdx.x < off) s_min[threadIdx.x] = fmin(s_min[threadIdx.x], s_min[threadIdx.x + off]);
    __syncthreads();
  }
  const double t = p.tdt[0];
  for (int i = blockIdx.x*256 + threadIdx.x; i < p.N; i += gridDim.x*256) {
    for (int k = 0; k < p.ndim; k++) {
      const size_t o = (size_t) k*p.N + i;
      p.r0[o] = p.r[o]; p.v0[o] = p.v[o]; p.a0[o] = p.a[o];
    }
    p.tlast[i] = t;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) p.tdt[2] = s_min[0];      // published as the timestep by k_nbody_publish
}
__global__ void k_nbody_publish(NbPtrs p) { p.tdt[1] = p.tdt[2]; }

static void nb_free(gh_nbody *nb)
{
  double **ptrs[] = {&nb->r, &nb->v, &nb->a, &nb->adot, &nb->r0, &nb->v0, &nb->a0, &nb->m, &nb->h, &nb->gpot, &nb->tlast, &nb->dti, &nb->red, &nb->stage};
  for (double **q : ptrs) { if (*q) (void) hipFree(*q); *q = nullptr; }
  if (nb->act) (void) hipFree(nb->act);
  nb->act = nullptr;
  nb->Ncap = 0;
}

// arrays for up to N stars (component-major: the stride is the CURRENT star count, not the capacity)
static int nb_reserve(gh_nbody *nb, int64_t N)
{
  if (N <= nb->Ncap) return GH_OK;
  const int64_t cap = N + 16;
  nb_free(nb);
  double **vec[] = {&nb->r, &nb->v, &nb->a, &nb->adot, &nb->r0, &nb->v0, &nb->a0};
  for (double **q : vec) NB_CHECK(nb, hipMalloc((void**) q, sizeof(double)*3*cap));
  NB_CHECK(nb, hipMalloc((void**) &nb->stage, sizeof(double)*4*cap));          // 3N transposition buffer / hybrid: gas a + gpot
  double **sca[] = {&nb->m, &nb->h, &nb->gpot, &nb->tlast, &nb->dti};
  for (double **q : sca) NB_CHECK(nb, hipMalloc((void**) q, sizeof(double)*cap));
  NB_CHECK(nb, hipMalloc((void**) &nb->red, sizeof(double)*((cap + 255)/256 + 1)));
  NB_CHECK(nb, hipMalloc((void**) &nb->act, sizeof(int)*cap));
  nb->Ncap = cap;
  return GH_OK;
}

extern "C" int gh_nbody_create(int ndim, int softening, double nbody_mult, int device, gh_nbody **out)
{
  if (!out) return GH_ERR_INVALID;
  *out = nullptr;
  if (ndim < 1 || ndim > 3) return GH_ERR_INVALID;
  gh_nbody *nb = new gh_nbody;
  nb->ndim = ndim; nb->softening = softening ? 1 : 0; nb->nbody_mult = nbody_mult; nb->device = device;
  *out = nb;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device >= ndev) {
    nb->err = "no HIP device: libgandalf_hip has no CPU path";
    return GH_ERR_HIP;
  }
  NB_CHECK(nb, hipSetDevice(device));
  NB_CHECK(nb, hipStreamCreate(&nb->stream));
  NB_CHECK(nb, hipMalloc((void**) &nb->tdt, 4*sizeof(double)));
  NB_CHECK(nb, hipMemset(nb->tdt, 0, 4*sizeof(double)));
  return GH_OK;
}

extern "C" void gh_nbody_destroy(gh_nbody *nb)
{
  if (!nb) return;
  if (nb->stream) (void) hipStreamSynchronize(nb->stream);
  nb_free(nb);
  if (nb->tdt) (void) hipFree(nb->tdt);
  if (nb->stream) (void) hipStreamDestroy(nb->stream);
  delete nb;
}

extern "C" const char *gh_nbody_last_error(const gh_nbody *nb) { return nb ? nb->err.c_str() : "null context"; }

extern "C" int gh_nbody_upload(gh_nbody *nb, int64_t N, const double *r, const double *v, const double *m, const double *h)
{
  if (!nb || N <= 0 || !r || !v || !m || !h) return GH_ERR_INVALID;
  const int nd = nb->ndim;
  int rca = nb_reserve(nb, N);
  if (rca) return rca;
  nb->N = N;
  { std::vector<double> big((size_t) N, NB_BIG); NB_CHECK(nb, hipMemcpy(nb->dti, big.data(), sizeof(double)*N, hipMemcpyHostToDevice)); }
  std::string tmp;
  std::vector<double> t((size_t) 3*N, 0.0);
  auto up = [&](double *dst, const double *src) -> hipError_t {
    for (int64_t i = 0; i < N; i++) for (int k = 0; k < nd; k++) t[(size_t) k*N + i] = src[(size_t) i*nd + k];
    return hipMemcpy(dst, t.data(), sizeof(double)*3*N, hipMemcpyHostToDevice);
  };
  NB_CHECK(nb, up(nb->r, r));
  NB_CHECK(nb, up(nb->v, v));
  NB_CHECK(nb, hipMemcpy(nb->r0, nb->r, sizeof(double)*3*N, hipMemcpyDeviceToDevice));
  NB_CHECK(nb, hipMemcpy(nb->v0, nb->v, sizeof(double)*3*N, hipMemcpyDeviceToDevice));
  NB_CHECK(nb, hipMemcpy(nb->m, m, sizeof(double)*N, hipMemcpyHostToDevice));
  NB_CHECK(nb, hipMemcpy(nb->h, h, sizeof(double)*N, hipMemcpyHostToDevice));
  double **zero3[] = {&nb->a, &nb->adot, &nb->a0};
  for (double **q : zero3) NB_CHECK(nb, hipMemset(*q, 0, sizeof(double)*3*N));
  NB_CHECK(nb, hipMemset(nb->gpot, 0, sizeof(double)*N));
  NB_CHECK(nb, hipMemset(nb->tlast, 0, sizeof(double)*N));
  NB_CHECK(nb, hipMemset(nb->tdt, 0, 4*sizeof(double)));
  return GH_OK;
}

extern "C" int gh_nbody_download(gh_nbody *nb, int field, double *out)
{
  if (!nb || !out || nb->N <= 0 || field < 0 || field >= GH_NB_ALLFIELDS) return GH_ERR_INVALID;
  const int64_t N = nb->N;
  const int nd = nb->ndim;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  if (field == GH_NB_GPOT) { NB_CHECK(nb, hipMemcpy(out, nb->gpot, sizeof(double)*N, hipMemcpyDeviceToHost)); return GH_OK; }
  if (field == GH_NB_TLAST) { NB_CHECK(nb, hipMemcpy(out, nb->tlast, sizeof(double)*N, hipMemcpyDeviceToHost)); return GH_OK; }
  const double *src = field == GH_NB_R ? nb->r : field == GH_NB_V ? nb->v : field == GH_NB_A ? nb->a : field == GH_NB_R0 ? nb->r0 :
                      field == GH_NB_V0 ? nb->v0 : field == GH_NB_A0 ? nb->a0 : nb->adot;
  std::vector<double> t((size_t) 3*N);
  NB_CHECK(nb, hipMemcpy(t.data(), src, sizeof(double)*3*N, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < N; i++) for (int k = 0; k < nd; k++) out[(size_t) i*nd + k] = t[(size_t) k*N + i];
  return GH_OK;
}

static int nb_launch_forces(gh_nbody *nb)
{
  NbPtrs p = nb_ptrs(nb);
  const dim3 grid((unsigned) ((nb->N + 63)/64)), block(256);
#define NB_LAUNCH(ND_) \
  if (nb->softening) hipLaunchKernelGGL((k_nbody_forces<ND_, true>), grid, block, 0, nb->stream, p); \
  else hipLaunchKernelGGL((k_nbody_forces<ND_, false>), grid, block, 0, nb->stream, p);
  if (nb->ndim == 1) { NB_LAUNCH(1) } else if (nb->ndim == 2) { NB_LAUNCH(2) } else { NB_LAUNCH(3) }
#undef NB_LAUNCH
  NB_CHECK(nb, hipGetLastError());
  return GH_OK;
}

extern "C" int gh_nbody_forces(gh_nbody *nb)
{
  if (!nb || nb->N <= 0) return GH_ERR_INVALID;
  int rc = nb_launch_forces(nb);
  if (rc) return rc;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  return GH_OK;
}

static int nb_timestep_end(gh_nbody *nb, int correct)
{
  NbPtrs p = nb_ptrs(nb);
  const int nblk = (int) ((nb->N + 255)/256);
  hipLaunchKernelGGL(k_nbody_correct_dt, dim3(nblk), dim3(256), 0, nb->stream, p, nb->nbody_mult, nb->red, correct);
  hipLaunchKernelGGL(k_nbody_end, dim3(std::min(nblk, 256)), dim3(256), 0, nb->stream, p, nb->red, nblk);
  hipLaunchKernelGGL(k_nbody_publish, dim3(1), dim3(1), 0, nb->stream, p);
  NB_CHECK(nb, hipGetLastError());
  return GH_OK;
}

extern "C" int gh_nbody_setup(gh_nbody *nb, double *timestep)
{
  if (!nb || nb->N <= 0) return GH_ERR_INVALID;
  int rc = nb_launch_forces(nb);
  if (rc) return rc;
  rc = nb_timestep_end(nb, 0);
  if (rc) return rc;
  double td[2];
  NB_CHECK(nb, hipMemcpyAsync(td, nb->tdt, sizeof(td), hipMemcpyDeviceToHost, nb->stream));
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  if (timestep) *timestep = td[1];
  return GH_OK;
}

extern "C" int gh_nbody_step(gh_nbody *nb, int nsteps, double *t, double *timestep)
{
  if (!nb || nb->N <= 0 || nsteps < 0) return GH_ERR_INVALID;
  NbPtrs p = nb_ptrs(nb);
  const int nblk = (int) ((nb->N + 255)/256);
  for (int s = 0; s < nsteps; s++) {
    hipLaunchKernelGGL(k_nbody_advance, dim3(nblk), dim3(256), 0, nb->stream, p, 0);
    hipLaunchKernelGGL(k_nbody_clock, dim3(1), dim3(1), 0, nb->stream, p);
    int rc = nb_launch_forces(nb);
    if (rc) return rc;
    rc = nb_timestep_end(nb, 1);
    if (rc) return rc;
  }
  double td[2];
  NB_CHECK(nb, hipMemcpyAsync(td, nb->tdt, sizeof(td), hipMemcpyDeviceToHost, nb->stream));
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  if (t) *t = td[0];
  if (timestep) *timestep = td[1];
  return GH_OK;
}

extern "C" int gh_nbody_upload_field(gh_nbody *nb, int field, const double *src)
{
  if (!nb || !src || nb->N <= 0 || field < 0 || field >= GH_NB_ALLFIELDS) return GH_ERR_INVALID;
  const int64_t N = nb->N;
  const int nd = nb->ndim;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  if (field == GH_NB_GPOT) { NB_CHECK(nb, hipMemcpy(nb->gpot, src, sizeof(double)*N, hipMemcpyHostToDevice)); return GH_OK; }
  if (field == GH_NB_TLAST) { NB_CHECK(nb, hipMemcpy(nb->tlast, src, sizeof(double)*N, hipMemcpyHostToDevice)); return GH_OK; }
  double *dst = field == GH_NB_R ? nb->r : field == GH_NB_V ? nb->v : field == GH_NB_A ? nb->a : field == GH_NB_R0 ? nb->r0 :
                field == GH_NB_V0 ? nb->v0 : field == GH_NB_A0 ? nb->a0 : nb->adot;
  std::vector<double> t((size_t) 3*N, 0.0);
  for (int64_t i = 0; i < N; i++) for (int k = 0; k < nd; k++) t[(size_t) k*N + i] = src[(size_t) i*nd + k];
  NB_CHECK(nb, hipMemcpy(dst, t.data(), sizeof(double)*3*N, hipMemcpyHostToDevice));
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// hybrid gas + stars stepping (SphSimulation::MainLoop with Nstar > 0, global timestep, Npec = 1)
// ------------------------------------------------------------------------------------------------
#include "gh_internal.hpp"
int gh_advance_time_impl(gh_ctx *ctx);
double *gh_time_dev(gh_ctx *ctx);
int gh_hybrid_gas_passes(gh_ctx *ctx);                       // api.hip: tree, density, zero, forces (enqueue only)
int gh_timestep_impl_extra(gh_ctx *ctx, int nextra);         // integrate.hip: global min incl. nextra values behind the block minima

// a += gas part, gpot += gas part (component-major star arrays; ga is [N][ndim] as gh_star_gas_forces returns it)
__global__ void k_nbody_add_gas(NbPtrs p, const double *ga, const double *gg)
{
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i >= p.N || (p.act && !p.act[i])) return;
  for (int k = 0; k < p.ndim; k++) p.a[(size_t) k*p.N + i] += ga[(size_t) i*p.ndim + k];
  p.gpot[i] += gg[i];
}

static int sink_hybrid_step(gh_ctx *gas, gh_nbody *nb);
static int sink_hybrid_setup(gh_ctx *gas, gh_nbody *nb, int initial_h_provided);

extern "C" int gh_hybrid_step(gh_ctx *gas, gh_nbody *nb, int nsteps, double *t_out, double *timestep_out)
{
  if (gas && nb && (gas->cfg.sink_particles || gas->cfg.Nlevels > 1)) {      // sink runs (the star list may be empty and grows), stars on the block-timestep ladder
    if (gas->N <= 0 || nsteps < 0) return GH_ERR_INVALID;
    for (int s = 0; s < nsteps; s++) { const int rc = sink_hybrid_step(gas, nb); if (rc) return rc; }
    if (t_out) *t_out = gas->t;
    if (timestep_out) *timestep_out = gas->timestep;
    return GH_OK;
  }
  if (!gas || !nb || nb->N <= 0 || gas->N <= 0 || nsteps < 0) return GH_ERR_INVALID;
  // (several ranks: stars go through the sink / block-timestep path above, which is the one the multi-rank tests exercise)
  if (gas->nranks > 1) return gh_fail(gas, GH_ERR_UNSUPPORTED, "hybrid runs with a global timestep and no sinks: one rank");
  if (!gas->cfg.self_gravity) return gh_fail(gas, GH_ERR_UNSUPPORTED, "hybrid runs need self_gravity = 1");
  const int64_t Ns = nb->N;
  const int nd = nb->ndim;
  NbPtrs p = nb_ptrs(nb);
  const int nblk = (int) ((Ns + 255)/256);
  std::vector<double> sr((size_t) Ns*nd), sm((size_t) Ns), shh((size_t) Ns), ga((size_t) Ns*nd), gg((size_t) Ns);
  NB_CHECK(nb, hipMemcpy(sm.data(), nb->m, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  NB_CHECK(nb, hipMemcpy(shh.data(), nb->h, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  int rc;
  for (int s = 0; s < nsteps; s++) {
    // the stars follow the gas clock
    double td[2] = {gas->t, gas->timestep};
    NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
    gas->n++; gas->Nsteps++;
    gh_advance_time_impl(gas);                                               // t = t + timestep
    gh_kdk_advance_impl(gas, gas->n, 0.0, 0.0);                              // hydroint->AdvanceParticles
    hipLaunchKernelGGL(k_nbody_advance, dim3(nblk), dim3(256), 0, nb->stream, p, 0);   // nbody->AdvanceParticles
    hipLaunchKernelGGL(k_nbody_clock, dim3(1), dim3(1), 0, nb->stream, p);
    if ((rc = gh_nbody_download(nb, GH_NB_R, sr.data()))) return rc;
    if ((rc = gh_set_stars(gas, Ns, sr.data(), sm.data(), shh.data(), nb->softening))) return rc;
    if ((rc = gh_hybrid_gas_passes(gas))) return rc;                         // tree, density (+ zeta), forces (+ gas <- stars)
    if ((rc = gh_star_gas_forces(gas, ga.data(), gg.data()))) return rc;    // UpdateAllStarGasForces
    if ((rc = nb_launch_forces(nb))) return rc;                              // star-star direct sum
    NB_CHECK(nb, hipMemcpyAsync(nb->stage, ga.data(), sizeof(double)*Ns*nd, hipMemcpyHostToDevice, nb->stream));
    NB_CHECK(nb, hipMemcpyAsync(nb->stage + (size_t) Ns*nd, gg.data(), sizeof(double)*Ns, hipMemcpyHostToDevice, nb->stream));
    hipLaunchKernelGGL(k_nbody_add_gas, dim3(nblk), dim3(256), 0, nb->stream, p, nb->stage, nb->stage + (size_t) Ns*nd);
    // CorrectionTerms, the stars' timestep minimum and EndTimestep
    hipLaunchKernelGGL(k_nbody_correct_dt, dim3(nblk), dim3(256), 0, nb->stream, p, nb->nbody_mult, nb->red, 1);
    hipLaunchKernelGGL(k_nbody_end, dim3(std::min(nblk, 256)), dim3(256), 0, nb->stream, p, nb->red, nblk);
    double star_min = 0.0;
    NB_CHECK(nb, hipMemcpyAsync(&star_min, nb->tdt + 2, sizeof(double), hipMemcpyDeviceToHost, nb->stream));
    NB_CHECK(nb, hipStreamSynchronize(nb->stream));
    // ComputeGlobalTimestep over both species, then the gas' EndTimestep
    GH_CHECK(gas, hipMemcpyAsync(gas->redbuf + 256, &star_min, sizeof(double), hipMemcpyHostToDevice, gas->stream));
    gh_timestep_impl_extra(gas, 1);
    gh_kdk_end_impl(gas, 0, 0.0, 0.0);
    gas->n = 0;
    if ((rc = gh_sync_collect(gas, "gh_hybrid_step"))) return rc;
    double tt[2];
    GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
    gas->t = tt[0]; gas->timestep = tt[1];
  }
  double td[2] = {gas->t, gas->timestep};
  NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
  if (t_out) *t_out = gas->t;
  if (timestep_out) *timestep_out = gas->timestep;
  return GH_OK;
}

int gh_setup_passes(gh_ctx *ctx, int initial_h_provided);    // api.hip

// PostInitialConditionsSetup of a hybrid run (SphSimulation.cpp:204-565): gas passes with the stars present, star forces
// (gas tree part + direct sum, :500-514), first timestep = minimum over both species (:538), EndTimestep of both (:551-553)
extern "C" int gh_hybrid_setup(gh_ctx *gas, gh_nbody *nb, int initial_h_provided, double *timestep_out)
{
  if (gas && nb && (gas->cfg.sink_particles || gas->cfg.Nlevels > 1)) {
    if (gas->N <= 0) return GH_ERR_INVALID;
    const int rc = sink_hybrid_setup(gas, nb, initial_h_provided);
    if (rc) return rc;
    if (timestep_out) *timestep_out = gas->timestep;
    return GH_OK;
  }
  if (!gas || !nb || nb->N <= 0 || gas->N <= 0) return GH_ERR_INVALID;
  if (gas->nranks > 1) return gh_fail(gas, GH_ERR_UNSUPPORTED, "hybrid runs with a global timestep and no sinks: one rank");
  if (!gas->cfg.self_gravity) return gh_fail(gas, GH_ERR_UNSUPPORTED, "hybrid runs need self_gravity = 1");
  const int64_t Ns = nb->N;
  const int nd = nb->ndim;
  NbPtrs p = nb_ptrs(nb);
  const int nblk = (int) ((Ns + 255)/256);
  std::vector<double> sr((size_t) Ns*nd), sm((size_t) Ns), shh((size_t) Ns), ga((size_t) Ns*nd), gg((size_t) Ns);
  NB_CHECK(nb, hipMemcpy(sm.data(), nb->m, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  NB_CHECK(nb, hipMemcpy(shh.data(), nb->h, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  int rc;
  if ((rc = gh_nbody_download(nb, GH_NB_R, sr.data()))) return rc;
  if ((rc = gh_set_stars(gas, Ns, sr.data(), sm.data(), shh.data(), nb->softening))) return rc;
  if ((rc = gh_setup_passes(gas, initial_h_provided))) return rc;
  if ((rc = gh_star_gas_forces(gas, ga.data(), gg.data()))) return rc;
  double td[2] = {gas->t, 0.0};
  NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
  if ((rc = nb_launch_forces(nb))) return rc;
  NB_CHECK(nb, hipMemcpyAsync(nb->stage, ga.data(), sizeof(double)*Ns*nd, hipMemcpyHostToDevice, nb->stream));
  NB_CHECK(nb, hipMemcpyAsync(nb->stage + (size_t) Ns*nd, gg.data(), sizeof(double)*Ns, hipMemcpyHostToDevice, nb->stream));
  hipLaunchKernelGGL(k_nbody_add_gas, dim3(nblk), dim3(256), 0, nb->stream, p, nb->stage, nb->stage + (size_t) Ns*nd);
  hipLaunchKernelGGL(k_nbody_correct_dt, dim3(nblk), dim3(256), 0, nb->stream, p, nb->nbody_mult, nb->red, 0);
  hipLaunchKernelGGL(k_nbody_end, dim3(std::min(nblk, 256)), dim3(256), 0, nb->stream, p, nb->red, nblk);
  double star_min = 0.0;
  NB_CHECK(nb, hipMemcpyAsync(&star_min, nb->tdt + 2, sizeof(double), hipMemcpyDeviceToHost, nb->stream));
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  gas->timestep = 0.0; gas->n = 0;
  double tt0[2] = {gas->t, 0.0};
  GH_CHECK(gas, hipMemcpyAsync(gh_time_dev(gas), tt0, sizeof(tt0), hipMemcpyHostToDevice, gas->stream));
  GH_CHECK(gas, hipMemcpyAsync(gas->redbuf + 256, &star_min, sizeof(double), hipMemcpyHostToDevice, gas->stream));
  gh_timestep_impl_extra(gas, 1);
  gh_kdk_end_impl(gas, 0, 0.0, 0.0);
  if ((rc = gh_sync_collect(gas, "gh_hybrid_setup"))) return rc;
  double tt[2];
  GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
  gas->t = tt[0]; gas->timestep = tt[1];
  td[0] = gas->t; td[1] = gas->timestep;
  NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
  if (timestep_out) *timestep_out = gas->timestep;
  return GH_OK;
}

// ------------------------------------------------------------------------------------------------
// sink runs: the stars are the sinks; their number grows, and the sink routines work on a host mirror
// ------------------------------------------------------------------------------------------------
static int nb_pull(gh_nbody *nb, gh_host_stars &S)
{
  const size_t N = (size_t) nb->N;
  S.resize(N);
  nb->h_level.resize(N, 0); nb->h_nstep.resize(N, 1); nb->h_nlast.resize(N, 0);
  S.level = nb->h_level; S.nstep = nb->h_nstep; S.nlast = nb->h_nlast;
  if (N == 0) return GH_OK;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  std::vector<double> t(3*N);
  struct { const double *src; std::vector<double> *dst; } v3[] = {{nb->r, &S.r}, {nb->v, &S.v}, {nb->a, &S.a}, {nb->adot, &S.adot}, {nb->r0, &S.r0}, {nb->v0, &S.v0}, {nb->a0, &S.a0}};
  for (auto &q : v3) {
    NB_CHECK(nb, hipMemcpy(t.data(), q.src, sizeof(double)*3*N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < N; i++) for (int k = 0; k < 3; k++) (*q.dst)[3*i + k] = t[(size_t) k*N + i];
  }
  struct { const double *src; std::vector<double> *dst; } v1[] = {{nb->m, &S.m}, {nb->h, &S.h}, {nb->gpot, &S.gpot}, {nb->tlast, &S.tlast}, {nb->dti, &S.dti}};
  for (auto &q : v1) NB_CHECK(nb, hipMemcpy(q.dst->data(), q.src, sizeof(double)*N, hipMemcpyDeviceToHost));
  return GH_OK;
}

static int nb_push(gh_nbody *nb, const gh_host_stars &S)
{
  const size_t N = S.n;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  int rc = nb_reserve(nb, (int64_t) N);
  if (rc) return rc;
  nb->N = (int64_t) N;
  nb->h_level = S.level; nb->h_nstep = S.nstep; nb->h_nlast = S.nlast;
  if (N == 0) return GH_OK;
  std::vector<double> t(3*N);
  struct { double *dst; const std::vector<double> *src; } v3[] = {{nb->r, &S.r}, {nb->v, &S.v}, {nb->a, &S.a}, {nb->adot, &S.adot}, {nb->r0, &S.r0}, {nb->v0, &S.v0}, {nb->a0, &S.a0}};
  for (auto &q : v3) {
    for (size_t i = 0; i < N; i++) for (int k = 0; k < 3; k++) t[(size_t) k*N + i] = (*q.src)[3*i + k];
    NB_CHECK(nb, hipMemcpy(q.dst, t.data(), sizeof(double)*3*N, hipMemcpyHostToDevice));
  }
  struct { double *dst; const std::vector<double> *src; } v1[] = {{nb->m, &S.m}, {nb->h, &S.h}, {nb->gpot, &S.gpot}, {nb->tlast, &S.tlast}, {nb->dti, &S.dti}};
  for (auto &q : v1) NB_CHECK(nb, hipMemcpy(q.dst, q.src->data(), sizeof(double)*N, hipMemcpyHostToDevice));
  return GH_OK;
}

// star positions, masses and softening lengths as the gas passes see them
static int sink_stars_to_gas(gh_ctx *gas, gh_nbody *nb)
{
  const int64_t Ns = nb->N;
  if (Ns == 0) return gh_set_stars(gas, 0, nullptr, nullptr, nullptr, nb->softening);
  const int nd = nb->ndim;
  std::vector<double> sr((size_t) Ns*nd), sm((size_t) Ns), shh((size_t) Ns);
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  NB_CHECK(nb, hipMemcpy(sm.data(), nb->m, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  NB_CHECK(nb, hipMemcpy(shh.data(), nb->h, sizeof(double)*Ns, hipMemcpyDeviceToHost));
  int rc = gh_nbody_download(nb, GH_NB_R, sr.data());
  if (rc) return rc;
  return gh_set_stars(gas, Ns, sr.data(), sm.data(), shh.data(), nb->softening);
}

// forces on the stars: gas tree part + star-star direct sum (SphSimulation.cpp:771-800)
static int sink_star_forces(gh_ctx *gas, gh_nbody *nb)
{
  const int64_t Ns = nb->N;
  if (Ns == 0) return GH_OK;
  const int nd = nb->ndim;
  std::vector<double> ga((size_t) Ns*nd), gg((size_t) Ns);
  int rc;
  if ((rc = gh_star_gas_forces(gas, ga.data(), gg.data()))) return rc;
  if ((rc = nb_launch_forces(nb))) return rc;
  NB_CHECK(nb, hipMemcpyAsync(nb->stage, ga.data(), sizeof(double)*Ns*nd, hipMemcpyHostToDevice, nb->stream));
  NB_CHECK(nb, hipMemcpyAsync(nb->stage + (size_t) Ns*nd, gg.data(), sizeof(double)*Ns, hipMemcpyHostToDevice, nb->stream));
  hipLaunchKernelGGL(k_nbody_add_gas, dim3((unsigned) ((Ns + 255)/256)), dim3(256), 0, nb->stream, nb_ptrs(nb), nb->stage, nb->stage + (size_t) Ns*nd);
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));         // ga / gg are on this function's stack
  return GH_OK;
}

// the stars' timestep minimum and EndTimestep; returns the minimum (big when there are no stars)
static int sink_star_end(gh_nbody *nb, double *star_min)
{
  *star_min = 9.9e50;
  const int64_t Ns = nb->N;
  if (Ns == 0) return GH_OK;
  NbPtrs p = nb_ptrs(nb);
  const int nblk = (int) ((Ns + 255)/256);
  hipLaunchKernelGGL(k_nbody_correct_dt, dim3(nblk), dim3(256), 0, nb->stream, p, nb->nbody_mult, nb->red, 0);
  hipLaunchKernelGGL(k_nbody_end, dim3(std::min(nblk, 256)), dim3(256), 0, nb->stream, p, nb->red, nblk);
  NB_CHECK(nb, hipMemcpyAsync(star_min, nb->tdt + 2, sizeof(double), hipMemcpyDeviceToHost, nb->stream));
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  return GH_OK;
}

static int sink_finish_step(gh_ctx *gas, gh_nbody *nb, double star_min, const char *where)
{
  int rc;
  GH_CHECK(gas, hipMemcpyAsync(gas->redbuf + 256, &star_min, sizeof(double), hipMemcpyHostToDevice, gas->stream));
  gh_timestep_impl_extra(gas, 1);                            // ComputeGlobalTimestep over gas (dead particles included) and stars
  gh_kdk_end_impl(gas, 0, 0.0, 0.0);
  gas->n = 0;
  if ((rc = gh_sync_collect(gas, where))) return rc;
  double tt[2];
  GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
  gas->t = tt[0]; gas->timestep = tt[1];
  NB_CHECK(nb, hipMemcpy(nb->tdt, tt, sizeof(tt), hipMemcpyHostToDevice));
  return GH_OK;
}

// EndTimestep of the stars that ended their step (NbodyLeapfrogKDK.cpp:341-377), on the host mirror
static void sink_star_end_host(gh_host_stars &S, int n, double t)
{
  for (size_t i = 0; i < S.n; i++) {
    if (!S.endflag[i]) continue;
    for (int k = 0; k < 3; k++) { S.r0[3*i + k] = S.r[3*i + k]; S.v0[3*i + k] = S.v[3*i + k]; S.a0[3*i + k] = S.a[3*i + k]; }
    S.nlast[i] = n; S.tlast[i] = t; S.endflag[i] = 0;
  }
}

// which stars end their step now (NbodyLeapfrogKDK::AdvanceParticles, :284: active if n - nlast == nstep)
static int sink_star_active(gh_ctx *gas, gh_nbody *nb)
{
  const size_t Ns = (size_t) nb->N;
  nb->act_on = gas->cfg.Nlevels > 1;
  if (!nb->act_on || Ns == 0) return GH_OK;
  std::vector<int> act(Ns);
  for (size_t i = 0; i < Ns; i++) act[i] = (gas->n - nb->h_nlast[i] == nb->h_nstep[i]) ? 1 : 0;
  NB_CHECK(nb, hipMemcpy(nb->act, act.data(), sizeof(int)*Ns, hipMemcpyHostToDevice));
  return GH_OK;
}

// one MainLoop call of a sink run (SphSimulation.cpp:574-880)
static int sink_hybrid_step(gh_ctx *gas, gh_nbody *nb)
{
  int rc;
  const bool levels = gas->cfg.Nlevels > 1;
  double td[2] = {gas->t, gas->timestep};
  NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
  if (levels) gh_block_begin_step(gas);
  else {
    gas->n++; gas->Nsteps++;
    gh_advance_time_impl(gas);
    gh_kdk_advance_impl(gas, gas->n, 0.0, 0.0);
  }
  if ((rc = sink_star_active(gas, nb))) return rc;
  if (nb->N > 0) hipLaunchKernelGGL(k_nbody_advance, dim3((unsigned) ((nb->N + 255)/256)), dim3(256), 0, nb->stream, nb_ptrs(nb), 0);
  hipLaunchKernelGGL(k_nbody_clock, dim3(1), dim3(1), 0, nb->stream, nb_ptrs(nb));
  if ((rc = sink_stars_to_gas(gas, nb))) return rc;
  // dead particles out, tree, density (+ zeta, potmin), forces (+ gas <- stars); block timesteps: repeated while CheckTimesteps wakes particles
  if (levels) { if ((rc = gh_block_gas_passes(gas))) return rc; }
  else if ((rc = gh_hybrid_gas_passes(gas))) return rc;
  if ((rc = sink_star_forces(gas, nb))) return rc;
  if (nb->N > 0)                                             // CorrectionTerms (its timestep output is redone below)
    hipLaunchKernelGGL(k_nbody_correct_dt, dim3((unsigned) ((nb->N + 255)/256)), dim3(256), 0, nb->stream, nb_ptrs(nb), nb->nbody_mult, nb->red, 1);
  if ((rc = gh_sync_collect(gas, "gh_hybrid_step/sinks"))) return rc;
  double tt[2];
  GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
  // search for new sinks, accrete (SphSimulation.cpp:820-838)
  gh_host_stars S;
  if ((rc = nb_pull(nb, S))) return rc;
  if ((rc = gh_sinks_step(gas, S, tt[0], tt[1]))) return rc;
  if (levels) {
    // ComputeBlockTimesteps over gas and stars, EndTimestep of both (:842, 868-872)
    if ((rc = gh_block_timesteps_hybrid(gas, S, nb->nbody_mult))) return rc;
    gh_kdk_end_impl(gas, 0, 0.0, 0.0);
    if ((rc = gh_sync_collect(gas, "gh_hybrid_step/sinks"))) return rc;
    gas->rebuild_tree = false;
    if ((rc = gh_block_pull(gas))) return rc;
    GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
    gas->t = tt[0]; gas->timestep = tt[1];
    sink_star_end_host(S, gas->n, tt[0]);
    if ((rc = nb_push(nb, S))) return rc;
    NB_CHECK(nb, hipMemcpy(nb->tdt, tt, sizeof(tt), hipMemcpyHostToDevice));
    return GH_OK;
  }
  if ((rc = nb_push(nb, S))) return rc;
  double star_min;
  if ((rc = sink_star_end(nb, &star_min))) return rc;
  return sink_finish_step(gas, nb, star_min, "gh_hybrid_step/sinks");
}

static int sink_hybrid_setup(gh_ctx *gas, gh_nbody *nb, int initial_h_provided)
{
  int rc;
  // SphSimulation.cpp:229-237: stars present at the start are sinks of radius kernrange*h
  gas->sinks.clear();
  if (nb->N > 0 && gas->cfg.sink_particles) {
    gh_host_stars S;
    if ((rc = nb_pull(nb, S))) return rc;
    const double kr = (gas->cfg.kernel == GH_KERNEL_QUINTIC || gas->cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0;
    for (size_t i = 0; i < S.n; i++) { gh_sink_rec k; k.istar = (int) i; k.radius = kr*S.h[i]; k.invh = 1.0/S.h[i]; gas->sinks.push_back(k); }
  }
  if ((rc = sink_stars_to_gas(gas, nb))) return rc;
  if ((rc = gh_setup_passes(gas, initial_h_provided))) return rc;
  double td[2] = {gas->t, 0.0};
  NB_CHECK(nb, hipMemcpy(nb->tdt, td, sizeof(td), hipMemcpyHostToDevice));
  nb->act_on = false;                                        // every star takes part in the setup
  if ((rc = sink_star_forces(gas, nb))) return rc;
  gas->timestep = 0.0; gas->n = 0;
  double tt0[2] = {gas->t, 0.0};
  GH_CHECK(gas, hipMemcpyAsync(gh_time_dev(gas), tt0, sizeof(tt0), hipMemcpyHostToDevice, gas->stream));
  if (gas->cfg.Nlevels > 1) {
    gas->nresync = 0;
    const int blk[8] = {0, 0, 0, 0, 0, 0, 1, 1};
    GH_CHECK(gas, hipMemcpyAsync(gas->d_blk, blk, sizeof(blk), hipMemcpyHostToDevice, gas->stream));
    gh_host_stars S;
    if ((rc = nb_pull(nb, S))) return rc;
    if ((rc = gh_block_timesteps_hybrid(gas, S, nb->nbody_mult))) return rc;      // n == nresync == 0: resynchronise (:539)
    gh_kdk_end_impl(gas, 0, 0.0, 0.0);
    if ((rc = gh_sync_collect(gas, "gh_hybrid_setup/sinks"))) return rc;
    if ((rc = gh_block_pull(gas))) return rc;
    double tt[2];
    GH_CHECK(gas, hipMemcpy(tt, gh_time_dev(gas), sizeof(tt), hipMemcpyDeviceToHost));
    gas->t = tt[0]; gas->timestep = tt[1];
    sink_star_end_host(S, gas->n, tt[0]);
    if ((rc = nb_push(nb, S))) return rc;
    NB_CHECK(nb, hipMemcpy(nb->tdt, tt, sizeof(tt), hipMemcpyHostToDevice));
    return GH_OK;
  }
  double star_min;
  if ((rc = sink_star_end(nb, &star_min))) return rc;
  return sink_finish_step(gas, nb, star_min, "gh_hybrid_setup/sinks");
}

extern "C" int64_t gh_nbody_num_stars(const gh_nbody *nb) { return nb ? nb->N : 0; }

// scalar star fields of sink runs: 0 m, 1 h, 2 dt_internal
extern "C" int gh_nbody_download_scalar(gh_nbody *nb, int which, double *out)
{
  if (!nb || !out || which < 0 || which > 2) return GH_ERR_INVALID;
  if (nb->N == 0) return GH_OK;
  NB_CHECK(nb, hipStreamSynchronize(nb->stream));
  NB_CHECK(nb, hipMemcpy(out, which == 0 ? nb->m : which == 1 ? nb->h : nb->dti, sizeof(double)*nb->N, hipMemcpyDeviceToHost));
  return GH_OK;
}
