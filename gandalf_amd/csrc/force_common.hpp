// force_common.hpp -- pieces shared by the force kernels (forces.hip, gravity.hip): parameter block,
// neighbour record, target record, the SPH pair term and the point-mass term.
#pragma once
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

struct ForceParams {
  Domain dom;
  EosParams eos;
  double alpha_visc, beta_visc;
  double alpha_visc_min;   // time_dependent_avisc = mm97
  int avisc, acond;
  const double *ktab;   // kernel tables (tabulated_kernel = 1) or nullptr
  double macerror;      // gravity_mac = gadget2
  int mac;              // GH_MAC_*
  int fastquad;         // multipole = fast_quadrupole: the fast_monopole kernel also adds the cells' quadrupole terms
  int group0;
  int stale;            // the tree has been extrapolated since its last stocking (Tree::ExtrapolateCellProperties): see k_density<.., STALE>
};

// per-neighbour record of the force tiles (reference HydroForcesParticle, Particle.h:313-364)
enum { T_X = 0, T_Y, T_Z, T_M, T_VX, T_VY, T_VZ, T_HR2, T_INVH, T_HFAC, T_PFAC, T_INVRHO, T_SOUND, T_ZETA, T_U, T_PRESS, T_NF,
       T_ALPHA = T_NF, T_NFA };     // T_ALPHA: extra row of the hydro-only kernel's tile (time-dependent viscosity)

struct TargetI {
  double r[3], v[3];
  double invh, hfactor, pfac, invrho, sound, zeta, hr2, u, press, invhsqd;
  double alpha;            // only read by the hydro-only kernel with mm97 viscosity
};

struct Accum {
  double a[3], at[3];
  double dudt, div_v, gpot;
};

// neighbour record in registers (same 16 fields, same order as the T_* tile / hrec layout)
struct Neib { double x, y, z, m, vx, vy, vz, hr2, invh, hfac, pfac, invrho, sound, zeta, u, press, alpha; };

__device__ __forceinline__ void neib_from_tile(Neib &n, const double (*s_t)[64], int c)
{
  n.x = s_t[T_X][c]; n.y = s_t[T_Y][c]; n.z = s_t[T_Z][c]; n.m = s_t[T_M][c];
  n.vx = s_t[T_VX][c]; n.vy = s_t[T_VY][c]; n.vz = s_t[T_VZ][c]; n.hr2 = s_t[T_HR2][c];
  n.invh = s_t[T_INVH][c]; n.hfac = s_t[T_HFAC][c]; n.pfac = s_t[T_PFAC][c]; n.invrho = s_t[T_INVRHO][c];
  n.sound = s_t[T_SOUND][c]; n.zeta = s_t[T_ZETA][c]; n.u = s_t[T_U][c]; n.press = s_t[T_PRESS][c];
}

// one SPH pair, particle i <- neighbour j               (GradhSph.cpp:384-448 / 498-572)
// TDAV: per-particle alpha in a self-gravity kernel (cd2010: ComputeH updates alpha whatever the force driver; only the fused
// kernel is instantiated with it)
template <int ND, bool GRAV, int KT, bool TDAV = false>
__device__ __forceinline__ void sph_pair(const ForceParams &P, const TargetI &ti, Accum &A, const Neib &nb,
                                         const double dr_in[3], double r2)
{
#pragma clang fp contract(fast)
  typedef typename KSel<ND, KT>::type K;
  double dr[3] = {dr_in[0], dr_in[1], dr_in[2]};
  double drmag;
  // |dr| and 1/|dr| from one rsqrt (the reference: sqrt then a division, GradhSph.cpp:399-400, 517-518)
  if (GRAV) {
    const double x = r2 + GH_SMALL;
    const double inv = fast_rsqrt(x);
    drmag = x*inv;
    for (int k = 0; k < ND; k++) dr[k] *= inv;
  }
  else {
    const double inv = r2 > 0.0 ? fast_rsqrt(r2) : 0.0;
    drmag = r2*inv;
    for (int k = 0; k < ND; k++) dr[k] *= inv;
  }
  const double mj = nb.m;
  const double invh_j = nb.invh;
  const double wkerni = ti.hfactor*K::t_w1(drmag*ti.invh, P.ktab);
  const double wkernj = nb.hfac*K::t_w1(drmag*invh_j, P.ktab);
  double dvdr = 0.0;
  {
    dvdr = (nb.vx - ti.v[0])*dr[0];
    if (ND > 1) dvdr += (nb.vy - ti.v[1])*dr[1];
    if (ND > 2) dvdr += (nb.vz - ti.v[2])*dr[2];
  }
  A.div_v -= mj*dvdr*wkerni;
  double paux = ti.pfac*wkerni + nb.pfac*wkernj;
  if (dvdr < 0.0) {
    const double invrho_j = nb.invrho;
    const double winvrho = 0.25*(wkerni + wkernj)*(ti.invrho + invrho_j);
    // mon97mm97 (GradhSph.cpp:419-424): the pair's mean alpha.  Only the hydro-only driver ever updates alpha
    // (GradhSphTree.cpp:403; UpdateAllSphForces never writes dalphadt back), so with self-gravity every alpha
    // stays alpha_visc_min and the host passes that as alpha_visc with avisc = mon97.
    if ((!GRAV || TDAV) && (P.avisc == GH_AVISC_MON97MM97 || P.avisc == GH_AVISC_MON97CD2010)) {
      const double alpha_mean = 0.5*(ti.alpha + nb.alpha);
      const double vsignal = ti.sound + nb.sound - P.beta_visc*alpha_mean*dvdr;
      paux -= alpha_mean*vsignal*dvdr*winvrho;
      A.dudt -= 0.5*mj*alpha_mean*vsignal*dvdr*dvdr*winvrho;
    }
    else if (P.avisc == GH_AVISC_MON97) {
      const double vsignal = ti.sound + nb.sound - P.beta_visc*P.alpha_visc*dvdr;
      paux -= P.alpha_visc*vsignal*dvdr*winvrho;
      A.dudt -= 0.5*mj*P.alpha_visc*vsignal*dvdr*dvdr*winvrho;
    }
    if (P.acond == GH_ACOND_WADSLEY2008)
      A.dudt += mj*dvdr*(nb.u - ti.u)*(ti.invrho*wkerni + invrho_j*wkernj);
    else if (P.acond == GH_ACOND_PRICE2008)
      A.dudt += 0.5*mj*(ti.u - nb.u)*winvrho*(ti.invrho + invrho_j)*sqrt(fabs(ti.press - nb.press));
  }
  for (int k = 0; k < ND; k++) A.a[k] += mj*dr[k]*paux;
  if (GRAV) {
    const double si = drmag*ti.invh, sj = drmag*invh_j;
    const double invsi = gh_fast_rcp(si), invsj = gh_fast_rcp(sj);
    const double pg = 0.5*(ti.invhsqd*K::t_wgrav(si, invsi, P.ktab) + ti.zeta*wkerni +
                           invh_j*invh_j*K::t_wgrav(sj, invsj, P.ktab) + nb.zeta*wkernj);
    for (int k = 0; k < ND; k++) A.at[k] += mj*dr[k]*pg;
    A.gpot += 0.5*mj*(ti.invh*K::t_wpot(si, invsi, P.ktab) + invh_j*K::t_wpot(sj, invsj, P.ktab));
  }
}

__device__ __forceinline__ void load_target(const DevicePtrs &d, int i, int ND, TargetI &t)
{
  for (int k = 0; k < 3; k++) { t.r[k] = k < ND ? d.f[D_RX + k][i] : 0.0; t.v[k] = k < ND ? d.f[D_VX + k][i] : 0.0; }
  const double h = d.f[D_H][i], rho = d.f[D_RHO][i];
  t.invh = 1.0/h; t.invhsqd = t.invh*t.invh;
  t.hfactor = d.f[D_HFACTOR][i];
  t.press = d.f[D_PRESSURE][i];
  t.pfac = (t.press*d.f[D_INVOMEGA][i])/(rho*rho);
  t.invrho = 1.0/rho;
  t.sound = d.f[D_SOUND][i];
  t.zeta = d.f[D_ZETA][i];
  t.hr2 = d.f[D_HRANGESQD][i];
  t.u = d.f[D_U][i];
}

// The neighbour record of the force tiles, 16 doubles = one 128-byte line per particle, in T_* order.
// Built once per force pass (k_pack_hydro) from the SoA arrays so that staging a tile is four 32-byte
// loads per lane instead of sixteen scattered ones plus three divisions.
static __global__ void k_pack_hydro(DevicePtrs d)
{
  const int j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= d.N) return;
  const double h = d.f[D_H][j], rho = d.f[D_RHO][j], press = d.f[D_PRESSURE][j];
  double4 q0, q1, q2, q3;
  q0.x = d.f[D_RX][j]; q0.y = d.ndim > 1 ? d.f[D_RY][j] : 0.0; q0.z = d.ndim > 2 ? d.f[D_RZ][j] : 0.0; q0.w = d.f[D_M][j];
  q1.x = d.f[D_VX][j]; q1.y = d.ndim > 1 ? d.f[D_VY][j] : 0.0; q1.z = d.ndim > 2 ? d.f[D_VZ][j] : 0.0; q1.w = d.f[D_HRANGESQD][j];
  q2.x = 1.0/h; q2.y = d.f[D_HFACTOR][j]; q2.z = (press*d.f[D_INVOMEGA][j])/(rho*rho); q2.w = 1.0/rho;
  q3.x = d.f[D_SOUND][j]; q3.y = d.f[D_ZETA][j]; q3.z = d.f[D_U][j]; q3.w = press;
  d.hrec[4*(size_t) j + 0] = q0; d.hrec[4*(size_t) j + 1] = q1; d.hrec[4*(size_t) j + 2] = q2; d.hrec[4*(size_t) j + 3] = q3;
}

// stage particle j (tree-order index) with image shift sh into tile slot `slot`
__device__ __forceinline__ void stage_neib(const DevicePtrs &d, int ND, double (*s_t)[64], int slot, int j, const double sg[3], const double sh[3], bool valid)
{
  double4 q0, q1, q2, q3;
  q0.x = 1e30; q0.y = 1e30; q0.z = 1e30; q0.w = 0.0;
  q1.x = 0.0; q1.y = 0.0; q1.z = 0.0; q1.w = 0.0;
  q2.x = 1.0; q2.y = 0.0; q2.z = 0.0; q2.w = 1.0;
  q3.x = 0.0; q3.y = 0.0; q3.z = 0.0; q3.w = 0.0;
  if (valid) {
    const double4 *r = d.hrec + 4*(size_t) j;
    q0 = r[0]; q1 = r[1]; q2 = r[2]; q3 = r[3];
    q0.x = sg[0]*q0.x + sh[0]; q0.y = sg[1]*q0.y + sh[1]; q0.z = sg[2]*q0.z + sh[2];     // image position
    q1.x *= sg[0]; q1.y *= sg[1]; q1.z *= sg[2];                                             // mirror images: v -> -v
  }
  s_t[T_X][slot] = q0.x; s_t[T_Y][slot] = q0.y; s_t[T_Z][slot] = q0.z; s_t[T_M][slot] = q0.w;
  s_t[T_VX][slot] = q1.x; s_t[T_VY][slot] = q1.y; s_t[T_VZ][slot] = q1.z; s_t[T_HR2][slot] = q1.w;
  s_t[T_INVH][slot] = q2.x; s_t[T_HFAC][slot] = q2.y; s_t[T_PFAC][slot] = q2.z; s_t[T_INVRHO][slot] = q2.w;
  s_t[T_SOUND][slot] = q3.x; s_t[T_ZETA][slot] = q3.y; s_t[T_U][slot] = q3.z; s_t[T_PRESS][slot] = q3.w;
}

// 1/sqrt(x) for x > 0: hardware estimate (v_rsq_f64, ~2^-26) + two Newton steps in FMA form; ends within
// 1-2 ulp.  (ocml's rsqrt also handles denormals/inf/nan, which cannot occur here: x >= 1e-20.)
// block timesteps: a target of level `mylevel` raises levelneib of neighbour j to its level (GradhSph.cpp:446 with
// GradhSphTree.cpp:375-417: integer max, propagated to the real particle whatever image was used).  levelneib only
// grows during a force pass, so a plain read that already shows >= mylevel makes the atomic unnecessary.
__device__ __forceinline__ void raise_levelneib(const DevicePtrs &d, int j, int mylevel)
{
  if ((int) d.f[D_LEVELNEIB][j] < mylevel)
    atomicMax((unsigned long long*) &d.f[D_LEVELNEIB][j], (unsigned long long) __double_as_longlong((double) mylevel));
}

// far-field entry evaluation: a += m dr/(dr^2+eps)^(3/2), gpot += m/(dr^2+eps)^(1/2).  The reference
// writes this once with 1/x and sqrt (cells, NeighbourSearch.h:364-372) and once with 1/sqrt(x) (direct
// particles, GradhSph.cpp:675-681); both are evaluated here with one rsqrt (<= 2 ulp from either).
// Branch-free: a lane that does not take the entry passes m = 0.
template <int ND>
__device__ __forceinline__ void point_mass(const TargetI &ti, Accum &A, double x, double y, double z, double m)
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
  const double invdrmag = fast_rsqrt(drsqd);
  const double minvdr3 = m*(invdrmag*invdrmag*invdrmag);
  A.gpot += m*invdrmag;
  for (int k = 0; k < ND; k++) A.at[k] += dr[k]*minvdr3;
}

