// sph_kernels.hpp -- device restatement of the reference's M4 (cubic spline) kernel functions
// (reference src/Headers/SmoothingKernel.h:131-240, normalisation src/Hydrodynamics/M4Kernel.cpp:39-53)
// and of the closed-form EOS used by the configs (src/Thermal/AdiabaticEOS.cpp:69, IsothermalEOS.cpp).
// Powers are written as products (the reference calls pow(s,n)); the difference is < 1 ulp per term.
// The kernel polynomials may be contracted to FMAs (they only feed sums; membership tests - which must
// round like the reference's x86 build - are made by the callers on uncontracted sums of squares).
#pragma once
#include <hip/hip_runtime.h>

#define GH_INVPI 0.31830988618379      /* reference Constants.h:63 (truncated on purpose) */
#define GH_TWOTHIRDS 0.66666666666666666666666
#define GH_SMALL 1.0e-20               /* Constants.h:73 small_number */
#define GH_SMALL_DP 1.0e-50            /* Constants.h:90 small_number_dp */

// 1/x from v_rcp_f64 and two Newton steps (<= 1 ulp): used where the reference divides inside a pair loop
__device__ __forceinline__ double gh_fast_rcp(double x)
{
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rcp(x);
  y = y + y*(1.0 - x*y);
  y = y + y*(1.0 - x*y);
  return y;
}

// 1/sqrt(x): v_rsq_f64 (~2^-26) + Newton steps
__device__ __forceinline__ double fast_rsqrt(double x)
{
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5*x;
  y = y*(1.5 - hx*y*y);
  y = y*(1.5 - hx*y*y);
  return y;
}

// one Newton step: v_rsq_f64 is good to ~2^-26, one step gives ~2^-51 - far inside the 1e-11 force tolerance
__device__ __forceinline__ double fast_rsqrt1(double x)
{
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rsq(x);
  y = y*(1.5 - (0.5*x)*y*y);
  return y;
}


// sqrt(x) as x*rsqrt(x) with two Newton steps on the rsqrt (<= 2 ulp; IEEE sqrt costs ~4x as many instructions)
__device__ __forceinline__ double gh_fast_sqrt(double x)
{
#pragma clang fp contract(fast)
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5*x;
  y = y*(1.5 - hx*y*y);
  y = y*(1.5 - hx*y*y);
  return x > 0.0 ? x*y : 0.0;
}

template <int ND> struct M4 {
  static constexpr double kernrange = 2.0;
  static constexpr double kernrangesqd = 4.0;
  __host__ __device__ static constexpr double norm()
  {
    return ND == 1 ? GH_TWOTHIRDS : (ND == 2 ? GH_INVPI*(10.0/7.0) : GH_INVPI);
  }
  __device__ static __forceinline__ double w0(double s)
  {
#pragma clang fp contract(fast)
    if (s < 1.0) return norm()*(1.0 - 1.5*s*s + 0.75*s*s*s);
    else if (s < 2.0) { const double t = 2.0 - s; return 0.25*norm()*(t*t*t); }
    return 0.0;
  }
  __device__ static __forceinline__ double w1(double s)
  {
#pragma clang fp contract(fast)
    if (s < 1.0) return norm()*(-3.0*s + 2.25*s*s);
    else if (s < 2.0) return -0.75*norm()*(2.0 - s)*(2.0 - s);
    return 0.0;
  }
  __device__ static __forceinline__ double womega(double s)
  {
#pragma clang fp contract(fast)
    const double nd = (double) ND;
    if (s < 1.0) return norm()*(-nd + 1.5*(nd + 2.0)*s*s - 0.75*(nd + 3.0)*(s*s*s));
    else if (s < 2.0)
      return norm()*(-2.0*nd + 3.0*(nd + 1.0)*s - 1.50*(nd + 2.0)*s*s + 0.25*(nd + 3.0)*(s*s*s));
    return 0.0;
  }
  __device__ static __forceinline__ double wzeta(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s;
    if (s < 1.0) return 1.4 - 2.0*s2 + 1.5*(s2*s2) - 0.6*(s2*s2*s);
    else if (s < 2.0) return 1.6 - 4.0*s2 + 4.0*(s2*s) - 1.5*(s2*s2) + 0.2*(s2*s2*s);
    return 0.0;
  }
  __device__ static __forceinline__ double wgrav(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s;
    if (s < 1.0) return 1.333333333333333333333*s - 1.2*(s2*s) + 0.5*(s2*s2);
    else if (s < 2.0)
      return 2.6666666666666666667*s - 3.0*s2 + 1.2*(s2*s) - 0.166666666666666666667*(s2*s2) -
             0.06666666666666666667/s2;
    return 1.0/s2;
  }
  // wgrav / wpot with 1/s supplied by the caller (one reciprocal shared by both, no division in any branch)
  __device__ static __forceinline__ double wgrav_i(double s, double invs)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, invs2 = invs*invs;
    if (s < 1.0) return 1.333333333333333333333*s - 1.2*(s2*s) + 0.5*(s2*s2);
    else if (s < 2.0)
      return 2.6666666666666666667*s - 3.0*s2 + 1.2*(s2*s) - 0.166666666666666666667*(s2*s2) -
             0.06666666666666666667*invs2;
    return invs2;
  }
  __device__ static __forceinline__ double wpot_i(double s, double invs)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s;
    if (s < 1.0) return 1.4 - 0.666666666666666666666666*s2 + 0.3*(s2*s2) - 0.1*(s2*s2*s);
    else if (s < 2.0)
      return -(1.0/15.0)*invs + 1.6 - 1.33333333333333333333333333*s2 + (s2*s) - 0.3*(s2*s2) +
             (1.0/30.0)*(s2*s2*s);
    return invs;
  }
  __device__ static __forceinline__ double wpot(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s;
    if (s < 1.0) return 1.4 - 0.666666666666666666666666*s2 + 0.3*(s2*s2) - 0.1*(s2*s2*s);
    else if (s < 2.0)
      return -1.0/(15.0*s) + 1.6 - 1.33333333333333333333333333*s2 + (s2*s) - 0.3*(s2*s2) +
             (1.0/30.0)*(s2*s2*s);
    return 1.0/s;
  }
  // table-aware entry points (tab is ignored by the analytic kernels; see TabK below)
  __device__ static __forceinline__ double t_w0s2(double s2, const double *) { return w0(gh_fast_sqrt(s2)); }
  __device__ static __forceinline__ double t_womegas2(double s2, const double *) { return womega(gh_fast_sqrt(s2)); }
  __device__ static __forceinline__ double t_wzetas2(double s2, const double *) { return wzeta(gh_fast_sqrt(s2)); }
  // The three density-pass functions of one s^2 together, branch-free: with a = max(2 - s, 0), b = max(1 - s, 0) the
  // piecewise polynomials above are exactly
  //   w0 = norm (a^3/4 - b^3),   w1 = norm (3 b^2 - 3/4 a^2),   womega = -(ND w0 + s w1),
  //   wzeta = a^4 (0.1 + 0.2 s) - b^4 (0.2 + 0.8 s)
  // (expand for s < 1 and 1 <= s < 2).  One square root, ~20 fused operations and no divergent branch per pair instead
  // of three two-sided branches; values agree with the forms above to a few ulp of the kernel's maximum.
  __device__ static __forceinline__ void t_dens3(double s2, const double *, double &o_w0, double &o_womega, double &o_wzeta)
  {
#pragma clang fp contract(fast)
    const double s = gh_fast_sqrt(s2);
    const double a = fmax(2.0 - s, 0.0), b = fmax(1.0 - s, 0.0);
    const double a2 = a*a, b2 = b*b;
    const double w = norm()*(0.25*(a2*a) - b2*b);
    const double w1_ = norm()*(3.0*b2 - 0.75*a2);
    o_w0 = w;
    o_womega = -((double) ND*w + s*w1_);
    o_wzeta = (a2*a2)*(0.1 + 0.2*s) - (b2*b2)*(0.2 + 0.8*s);
  }
  __device__ static __forceinline__ double t_w1(double s, const double *) { return w1(s); }
  __device__ static __forceinline__ double t_wgrav(double s, double invs, const double *) { return wgrav_i(s, invs); }
  __device__ static __forceinline__ double t_wpot(double s, double invs, const double *) { return wpot_i(s, invs); }
  __device__ static __forceinline__ double t_wpot0(const double *) { return wpot(0.0); }
};


// Quintic spline (reference src/Headers/SmoothingKernel.h:281-408, norms src/Hydrodynamics/QuinticKernel.cpp:39-60).
// Powers are written as products (the reference calls pow(s,n)); wzeta keeps the reference's constants as written.
template <int ND> struct Quintic {
  static constexpr double kernrange = 3.0;
  static constexpr double kernrangesqd = 9.0;
  __host__ __device__ static constexpr double norm()
  {
    return ND == 1 ? (1.0/120.0) : (ND == 2 ? GH_INVPI*(7.0/478.0) : GH_INVPI*(1.0/120.0));
  }
  __device__ static __forceinline__ double w0(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2, s5 = s4*s;
    if (s < 1.0) return norm()*(66.0 - 60.0*s2 + 30.0*s4 - 10.0*s5);
    else if (s < 2.0) return norm()*(51.0 + 75.0*s - 210.0*s2 + 150.0*s3 - 45.0*s4 + 5.0*s5);
    else if (s < 3.0) return norm()*(243.0 - 405.0*s + 270.0*s2 - 90.0*s3 + 15.0*s4 - s5);
    return 0.0;
  }
  __device__ static __forceinline__ double w1(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2;
    if (s < 1.0) return norm()*(-120.0*s + 120.0*s3 - 50.0*s4);
    else if (s < 2.0) return norm()*(75.0 - 420.0*s + 450.0*s2 - 180.0*s3 + 25.0*s4);
    else if (s < 3.0) return norm()*(-405.0 + 540.0*s - 270.0*s2 + 60.0*s3 - 5.0*s4);
    return 0.0;
  }
  __device__ static __forceinline__ double womega(double s)
  {
#pragma clang fp contract(fast)
    const double nd = (double) ND;
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2, s5 = s4*s;
    if (s < 1.0) return norm()*(-66.0*nd + 60.0*(nd + 2.0)*s2 - 30.0*(nd + 4.0)*s4 + 10.0*(nd + 5.0)*s5);
    else if (s < 2.0)
      return norm()*(-51.0*nd - 75.0*(nd + 1.0)*s + 210.0*(nd + 2.0)*s2 - 150.0*(nd + 3.0)*s3 + 45.0*(nd + 4.0)*s4 -
                     5.0*(nd + 5.0)*s5);
    else if (s < 3.0)
      return norm()*(-243.0*nd + 405.0*(nd + 1.0)*s - 270.0*(nd + 2.0)*s2 + 90.0*(nd + 3.0)*s3 - 15.0*(nd + 4.0)*s4 +
                     (nd + 5.0)*s5);
    return 0.0;
  }
  __device__ static __forceinline__ double wzeta(double s)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2, s5 = s4*s, s6 = s3*s3, s7 = s6*s;
    if (s < 1.0) return 33.0*s2 - 15.0*s4 + 5.0*s6 - 1.42857142857*s7 - 34.14285714;
    else if (s < 2.0) return 25.5*s2 + 25.0*s3 - 52.5*s4 + 30.0*s5 - 7.5*s6 + 0.7142857143*s7 - 33.785714286;
    else if (s < 3.0) return 121.5*s2 - 135.0*s3 + 67.5*s4 - 18.0*s5 + 2.5*s6 - 0.142857143*s7 - 52.07142857;
    return 0.0;
  }
  __device__ static __forceinline__ double wgrav_i(double s, double invs)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2, s5 = s4*s, s6 = s3*s3, invs2 = invs*invs;
    if (s < 1.0) return (12.0/359.0)*(22.0*s - 12.0*s3 + (30.0/7.0)*s5 - (5.0/4.0)*s6);
    else if (s < 2.0)
      return (12.0/359.0)*(17.0*s + (75.0/4.0)*s2 - 42.0*s3 + 25.0*s4 - (45.0/7.0)*s5 + (5.0/8.0)*s6 + (5.0/56.0)*invs2);
    else if (s < 3.0)
      return (12.0/359.0)*(81.0*s - (405.0/4.0)*s2 + 54.0*s3 - 15.0*s4 + (15.0/7.0)*s5 - (1.0/8.0)*s6 - (507.0/56.0)*invs2);
    return invs2;
  }
  __device__ static __forceinline__ double wpot_i(double s, double invs)
  {
#pragma clang fp contract(fast)
    const double s2 = s*s, s3 = s2*s, s4 = s2*s2, s5 = s4*s, s6 = s3*s3, s7 = s6*s;
    if (s < 1.0) return (12.0/359.0)*(-11.0*s2 + 3.0*s4 - (5.0/7.0)*s6 + (5.0/28.0)*s7 + (478.0/14.0));
    else if (s < 2.0)
      return (12.0/359.0)*(-(17.0/2.0)*s2 - (25.0/4.0)*s3 + (21.0/2.0)*s4 - 5.0*s5 + (15.0/14.0)*s6 - (5.0/56.0)*s7 +
                           (473.0/14.0) + (5.0/56.0)*invs);
    else if (s < 3.0)
      return (12.0/359.0)*(-(81.0/2.0)*s2 + (135.0/4.0)*s3 - (27.0/2.0)*s4 + 3.0*s5 - (5.0/14.0)*s6 + (1.0/56.0)*s7 +
                           (729.0/14.0) - (507.0/56.0)*invs);
    return invs;
  }
  __device__ static __forceinline__ double wgrav(double s) { return wgrav_i(s, 1.0/s); }
  __device__ static __forceinline__ double wpot(double s) { return s > 0.0 ? wpot_i(s, 1.0/s) : (12.0/359.0)*(478.0/14.0); }
  // table-aware entry points (tab is ignored by the analytic kernels; see TabK below)
  __device__ static __forceinline__ double t_w0s2(double s2, const double *) { return w0(gh_fast_sqrt(s2)); }
  __device__ static __forceinline__ double t_womegas2(double s2, const double *) { return womega(gh_fast_sqrt(s2)); }
  __device__ static __forceinline__ double t_wzetas2(double s2, const double *) { return wzeta(gh_fast_sqrt(s2)); }
  __device__ static __forceinline__ void t_dens3(double s2, const double *, double &o_w0, double &o_womega, double &o_wzeta)
  {
    const double s = gh_fast_sqrt(s2);
    o_w0 = w0(s); o_womega = womega(s); o_wzeta = wzeta(s);
  }
  __device__ static __forceinline__ double t_w1(double s, const double *) { return w1(s); }
  __device__ static __forceinline__ double t_wgrav(double s, double invs, const double *) { return wgrav_i(s, invs); }
  __device__ static __forceinline__ double t_wpot(double s, double invs, const double *) { return wpot_i(s, invs); }
  __device__ static __forceinline__ double t_wpot0(const double *) { return wpot(0.0); }
};


// Tabulated kernel (reference TabulatedKernel, SmoothingKernel.h:547-756, TabulatedKernel.cpp:57-100): 1000-entry
// PIECEWISE-CONSTANT tables of the base kernel, indexed (int)(s*res/R) or (int)(s^2*res/R^2); gravity tables fall
// back to 1/s^2, 1/s outside the kernel.  The tables are built on the host (api.hip) and passed by pointer.
#define GH_TAB_RES 1000
enum { GH_TAB_W1 = 0, GH_TAB_WGRAV, GH_TAB_WPOT, GH_TAB_W0S2, GH_TAB_WOMEGAS2, GH_TAB_WZETAS2, GH_TAB_COUNT };
template <int ND, class Base> struct TabK {
  static constexpr double kernrange = Base::kernrange;
  static constexpr double kernrangesqd = Base::kernrangesqd;
  __device__ static __forceinline__ double look(const double *tab, int which, double s)
  {
    if (s >= kernrange) return 0.0;
    return tab[which*GH_TAB_RES + (int) (s*((double) GH_TAB_RES/kernrange))];
  }
  __device__ static __forceinline__ double looksqd(const double *tab, int which, double s2)
  {
    if (s2 >= kernrangesqd) return 0.0;
    return tab[which*GH_TAB_RES + (int) (s2*((double) GH_TAB_RES/kernrangesqd))];
  }
  __device__ static __forceinline__ double t_w0s2(double s2, const double *tab) { return looksqd(tab, GH_TAB_W0S2, s2); }
  __device__ static __forceinline__ double t_womegas2(double s2, const double *tab) { return looksqd(tab, GH_TAB_WOMEGAS2, s2); }
  __device__ static __forceinline__ double t_wzetas2(double s2, const double *tab) { return looksqd(tab, GH_TAB_WZETAS2, s2); }
  __device__ static __forceinline__ void t_dens3(double s2, const double *tab, double &o_w0, double &o_womega, double &o_wzeta)
  {
    o_w0 = looksqd(tab, GH_TAB_W0S2, s2); o_womega = looksqd(tab, GH_TAB_WOMEGAS2, s2); o_wzeta = looksqd(tab, GH_TAB_WZETAS2, s2);
  }
  __device__ static __forceinline__ double t_w1(double s, const double *tab) { return look(tab, GH_TAB_W1, s); }
  __device__ static __forceinline__ double t_wgrav(double s, double invs, const double *tab)
  {
    if (s >= kernrange) return invs*invs;
    return tab[GH_TAB_WGRAV*GH_TAB_RES + (int) (s*((double) GH_TAB_RES/kernrange))];
  }
  __device__ static __forceinline__ double t_wpot(double s, double invs, const double *tab)
  {
    if (s >= kernrange) return invs;
    return tab[GH_TAB_WPOT*GH_TAB_RES + (int) (s*((double) GH_TAB_RES/kernrange))];
  }
  __device__ static __forceinline__ double t_wpot0(const double *tab) { return tab[GH_TAB_WPOT*GH_TAB_RES]; }
};

// kernel selector: KT = gh_config::kernel (GH_KERNEL_M4 = 0, GH_KERNEL_QUINTIC = 1, GH_KERNEL_M4_TAB = 2, GH_KERNEL_QUINTIC_TAB = 3)
template <int ND, int KT> struct KSel { typedef M4<ND> type; };
template <int ND> struct KSel<ND, 1> { typedef Quintic<ND> type; };
template <int ND> struct KSel<ND, 2> { typedef TabK<ND, M4<ND> > type; };
template <int ND> struct KSel<ND, 3> { typedef TabK<ND, Quintic<ND> > type; };

// pow(x, ND) / pow(x, ND+1) as the reference writes hfactor (GradhSph.cpp:192, 264)
template <int ND> __device__ __forceinline__ double powN(double x)
{
  return ND == 1 ? x : (ND == 2 ? x*x : x*x*x);
}

struct EosParams { int kind; double gamma, gammam1, temp0, mu_bar, rho_bary; };

// u, sound, pressure of a particle with density rho and energy u (GradhSph::ComputeThermalProperties,
// GradhSph.cpp:335-347).  energy_eqn: u is kept, c = sqrt(gamma (gamma-1) u), P = (gamma-1) rho u.
__device__ __forceinline__ void eos_eval(const EosParams &e, double rho, double &u, double &sound, double &press)
{
  if (e.kind == GH_EOS_ENERGY_EQN) {
    sound = sqrt(e.gamma*e.gammam1*u);
    press = e.gammam1*rho*u;
  }
  else {
    // dimensionless units.  isothermal: u = temp0/gammam1/mu_bar (IsothermalEOS.cpp:72-87); barotropic:
    // u = temp0 (1 + (rho/rho_bary)^(gamma-1))/gammam1/mu_bar (BarotropicEOS.cpp:78-91); c = sqrt(gammam1 u), P = gammam1 rho u
    if (e.kind == GH_EOS_BAROTROPIC) u = e.temp0*(1.0 + pow(rho*(1.0/e.rho_bary), e.gammam1))/e.gammam1/e.mu_bar;
    else u = e.temp0/e.gammam1/e.mu_bar;
    sound = sqrt(e.gammam1*u);
    press = e.gammam1*rho*u;
  }
}
