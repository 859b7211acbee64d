// oracle/gandalf_oracle.cpp -- TEST INFRASTRUCTURE, not product code.
//
// CPU restatement of the reference's algorithm for the hot path (GANDALF v0.4.0, double precision,
// M4 kernel, tabulated_kernel = 0, mon97 viscosity, Nlevels = 1, no stars):
//   KD-tree build + stocking, periodic ghosts + ghost tree, the density / h pass, the hydro force pass,
//   the hydro + self-gravity force pass (geometric MAC, monopole), leapfrog KDK and the global timestep.
// Every function cites the reference file:line it follows and keeps its loop order and arithmetic order
// (same operand order, pow() where the reference calls pow(), no FMA: build with -ffp-contract=off).
// It is pinned against the reference's own outputs (tests/golden/*.npz, written from the compiled
// reference by scripts/make_golden.py) in tests/test_oracle.py.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (gandalf_amd/) never does.
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double FLOAT;
static const FLOAT big_number = 9.9e20;         // Constants.h:72
static const FLOAT small_number = 1.0e-20;      // Constants.h:73
static const double small_number_dp = 1.0e-50;  // Constants.h:90
static const double big_number_dp = 9.9e50;     // Constants.h:89
static const FLOAT pi_const = 3.14159265358979; // Constants.h:60
static const FLOAT twopi = 6.28318530717959;    // Constants.h:61
static const FLOAT invpi = 0.31830988618379;    // Constants.h:63
static const FLOAT invlogetwo = 1.44269504088896;  // Constants.h:64
static const FLOAT twothirds = 0.66666666666666666666666;
static const FLOAT ghost_range = 2.5;           // Hydrodynamics.h:52

enum { F_DEAD = 1, F_ACTIVE = 2, F_END = 4, F_POTMIN = 8 };      // dead / active / end_timestep / potmin (Flags.h:29-35)

struct Part {                                   // Particle.h:133-223 + GradhSphParticle :285-368 (hot fields)
  int flags, iorig;
  int sinkid;                                     // sink the particle lies inside, -1 = none (Particle.h:139)
  int level, levelneib, nstep, nlast;             // block timesteps (Particle.h:137-142)
  FLOAT r[3], v[3], a[3], atree[3], r0[3], v0[3], a0[3];
  FLOAT m, h, hrangesqd, hfactor, sound, rho, pressure, u, u0, dudt0, dudt, gpot, gpot_hydro;
  double dt, dt_next, tlast;
  FLOAT div_v, invomega, zeta;
  FLOAT alpha, dalphadt;                          // time-dependent viscosity (mm97)
};

struct Cell {                                   // TreeCellBase, TreeCell.h:16-49 (+ KDTreeCell c1, c2)
  int cnext, copen, level, ifirst, ilast, N, Nactive, c1, c2;
  FLOAT cdistsqd, bbmin[3], bbmax[3], hbmin[3], hbmax[3], rcell[3], r[3], v[3], m, rmax, hmax;
  FLOAT q[5];                                   // traceless quadrupole about r (multipole = quadrupole)
  FLOAT amin;                                   // min |atree| of the cell's particles (gravity_mac = gadget2)
  FLOAT macfactor, mac;                         // eigenmac: max gpot^(-2/3) of the particles; eigenvalue length^2
};

struct Params {
  int ndim, Nleafmax, self_gravity, periodic[3], energy_integration, nthreads, kernel, multipole, acond, gravity_mac, tdavisc;
  int mirror[3][2];               // mirror wall at the lhs / rhs face of dimension k
  FLOAT macerror, alpha_visc_min;
  int Nlevels, level_diff_max, sph_single_timestep;   // block timesteps (Simulation.cpp:1209-1223)
  int gas_eos; FLOAT temp0, mu_bar, rho_bary;          // 0 energy_eqn, 1 isothermal, 2 barotropic
  int ntreebuildstep, ntreestockstep;
  FLOAT boxmin[3], boxmax[3], boxsize[3], boxhalf[3];
  FLOAT h_fac, h_converge, alpha_visc, beta_visc, gamma, thetamaxsqd, courant_mult, accel_mult, energy_mult;
  // sink particles (SphSimulation.cpp:116-136): sink_radius_mode 0 fixed, 1 hmult, 2 anything else (kernrange*h)
  int sink_particles = 0, create_sinks = 0, smooth_accretion = 0, sink_radius_mode = 1, Nsinkfixed = -1;
  FLOAT rho_sink = 0.0, sink_radius = 0.0, alpha_ss = 0.0, smooth_accrete_frac = 0.0, smooth_accrete_dt = 0.0;
};

struct Sink {                                   // SinkParticle, Sinks.h:48-100 (star = NbodyOracle::s[istar])
  int istar, Ngas;
  FLOAT radius, dmdt, menc, mmax, ketot, gpetot, rotketot, utot, taccrete, trad, trot, tvisc, angmom[3];
};

// ---------------------------------------------------------------------------------------------
// M4 kernel, SmoothingKernel.h:131-240, M4Kernel.cpp:39-53
// ---------------------------------------------------------------------------------------------
// pow(x, n) with a compile-time integer exponent as the reference's templates see it: g++ folds
// pow(x,1) -> x and pow(x,2) -> x*x (exactly rounded) even without -ffast-math, higher powers stay libm calls
static inline FLOAT pow_ref(FLOAT x, int n) { return n == 1 ? x : (n == 2 ? x*x : pow(x, (FLOAT) n)); }

struct M4 {            // kernel object: type 0 = M4, 1 = quintic (SmoothingKernel.h:281-408, QuinticKernel.cpp:39-60)
  int ndim, type, tabulated; FLOAT kernnorm, kernrange, kernrangesqd, invkernrange;
  // TabulatedKernel (SmoothingKernel.h:547-756, TabulatedKernel.cpp:57-100): piecewise-constant tables
  static const int res = 1000;
  FLOAT resinvkernrange, resinvkernrangesqd;
  std::vector<FLOAT> tW0, tW1, tWomega, tWzeta, tWgrav, tWpot, tW0_s2, tWomega_s2, tWzeta_s2;
  explicit M4(int nd, int type_ = 0, int tab_ = 0) : ndim(nd), type(type_ & 1), tabulated(tab_) {
    if (type == 0) {
      kernrange = 2.0; kernrangesqd = 4.0;
      kernnorm = nd == 1 ? twothirds : (nd == 2 ? invpi*(FLOAT) (10.0/7.0) : invpi);
    }
    else {
      kernrange = 3.0; kernrangesqd = 9.0;
      kernnorm = nd == 1 ? (FLOAT) (1.0/120.0) : (nd == 2 ? invpi*(FLOAT) (7.0/478.0) : invpi*(FLOAT) (1/120.));
    }
    invkernrange = type == 0 ? (FLOAT) 0.5 : (FLOAT) (1.0/3.0);        // M4Kernel.cpp:45, QuinticKernel.cpp:48
    resinvkernrange = res/kernrange; resinvkernrangesqd = res/kernrangesqd;
    if (tabulated) {
      const FLOAT step = kernrange/res, stepsq = kernrangesqd/res;
      tW0.resize(res); tW1.resize(res); tWomega.resize(res); tWzeta.resize(res); tWgrav.resize(res); tWpot.resize(res);
      tW0_s2.resize(res); tWomega_s2.resize(res); tWzeta_s2.resize(res);
      for (int i = 0; i < res; i++) {
        tW0[i] = a_w0(step*i); tW1[i] = a_w1(step*i); tWomega[i] = a_womega(step*i); tWzeta[i] = a_wzeta(step*i);
        tWgrav[i] = a_wgrav(step*i); tWpot[i] = a_wpot(step*i);
        tW0_s2[i] = a_w0(sqrt(stepsq*i)); tWomega_s2[i] = a_womega(sqrt(stepsq*i)); tWzeta_s2[i] = a_wzeta(sqrt(stepsq*i));
      }
    }
  }
  FLOAT look(const std::vector<FLOAT> &t, FLOAT s) const { if (s >= kernrange) return 0.0; return t[(int) (s*resinvkernrange)]; }
  FLOAT looksqd(const std::vector<FLOAT> &t, FLOAT s2) const { if (s2 >= kernrangesqd) return 0.0; return t[(int) (s2*resinvkernrangesqd)]; }
  FLOAT w0(FLOAT s) const { return tabulated ? look(tW0, s) : a_w0(s); }
  FLOAT w1(FLOAT s) const { return tabulated ? look(tW1, s) : a_w1(s); }
  FLOAT womega(FLOAT s) const { return tabulated ? look(tWomega, s) : a_womega(s); }
  FLOAT wzeta(FLOAT s) const { return tabulated ? look(tWzeta, s) : a_wzeta(s); }
  FLOAT wgrav(FLOAT s) const { if (!tabulated) return a_wgrav(s); if (s >= kernrange) return (FLOAT) 1.0/(s*s); return tWgrav[(int) (s*resinvkernrange)]; }
  FLOAT wpot(FLOAT s) const { if (!tabulated) return a_wpot(s); if (s >= kernrange) return (FLOAT) 1.0/s; return tWpot[(int) (s*resinvkernrange)]; }
  // w0_s2 etc.: base kernels take the square root (SmoothingKernel.h:78-92), the tabulated one indexes by s^2
  FLOAT w0_s2(FLOAT s2) const { return tabulated ? looksqd(tW0_s2, s2) : a_w0(sqrt(s2)); }
  FLOAT womega_s2(FLOAT s2) const { return tabulated ? looksqd(tWomega_s2, s2) : a_womega(sqrt(s2)); }
  FLOAT wzeta_s2(FLOAT s2) const { return tabulated ? looksqd(tWzeta_s2, s2) : a_wzeta(sqrt(s2)); }
  FLOAT a_w0(FLOAT s) const {
    if (type == 1) {
      if (s < 1.0) return (kernnorm)*(66.0 - 60.0*s*s + 30.0*pow(s,4) - 10.0*pow(s,5));
      else if (s < 2.0) return (kernnorm)*(51.0 + 75.0*s - 210.0*s*s + 150.0*pow(s,3) - 45.0*pow(s,4) + 5.0*pow(s,5));
      else if (s < 3.0) return (kernnorm)*(243.0 - 405*s + 270.0*s*s - 90.0*pow(s,3) + 15.0*pow(s,4) - pow(s,5));
      else return 0.0;
    }
    if (s < 1.0) return kernnorm*(1.0 - 1.5*s*s + 0.75*s*s*s);
    else if (s < 2.0) return 0.25*kernnorm*pow(2.0 - s, 3);
    else return 0.0;
  }
  FLOAT a_w1(FLOAT s) const {
    if (type == 1) {
      if (s < 1.0) return (kernnorm)*(-120.0*s + 120.0*pow(s,3) - 50.0*pow(s,4));
      else if (s < 2.0) return (kernnorm)*(75.0 - 420.0*s + 450.0*s*s - 180.0*pow(s,3) + 25.0*pow(s,4));
      else if (s < 3.0) return (kernnorm)*(-405.0 + 540.0*s - 270.0*s*s + 60.0*pow(s,3) - 5.0*pow(s,4));
      else return 0.0;
    }
    if (s < 1.0) return kernnorm*(-3.0*s + 2.25*s*s);
    else if (s < 2.0) return -0.75*kernnorm*(2.0 - s)*(2.0 - s);
    else return 0.0;
  }
  FLOAT a_womega(FLOAT s) const {
    if (type == 1) {
      if (s < 1.0)
        return (kernnorm)*(-66.0*(ndim) + 60.0*((ndim) + 2.0)*s*s - 30.0*((ndim) + 4.0)*pow(s,4) + 10.0*((ndim) + 5.0)*pow(s,5));
      else if (s < 2.0)
        return (kernnorm)*(-51.0*(ndim) - 75.0*((ndim) + 1.0)*s + 210.0*((ndim) + 2.0)*s*s - 150.0*((ndim) + 3.0)*pow(s,3) +
                           45.0*((ndim) + 4.0)*pow(s,4) - 5.0*((ndim) + 5.0)*pow(s,5));
      else if (s < 3.0)
        return (kernnorm)*(-243.0*(ndim) + 405.0*((ndim) + 1.0)*s - 270.0*((ndim) + 2.0)*s*s + 90.0*((ndim) + 3.0)*pow(s,3) -
                           15.0*((ndim) + 4.0)*pow(s,4) + ((ndim) + 5.0)*pow(s,5));
      else return 0.0;
    }
    if (s < 1.0) return kernnorm*(-ndim + 1.5*(ndim + 2.0)*s*s - 0.75*(ndim + 3.0)*pow(s, 3));
    else if (s < 2.0)
      return kernnorm*(-2.0*ndim + 3.0*(ndim + 1.0)*s - 1.50*(ndim + 2.0)*s*s + 0.25*(ndim + 3.0)*pow(s, 3));
    else return 0.0;
  }
  FLOAT a_wzeta(FLOAT s) const {
    if (type == 1) {
      if (s < (FLOAT) 1.0)
        return (FLOAT) 33.0*s*s - (FLOAT) 15.0*pow(s,4) + (FLOAT) 5.0*pow(s,6) - (FLOAT) 1.42857142857*pow(s,7) - (FLOAT) 34.14285714;
      else if (s < 2.0)
        return (FLOAT) 25.5*s*s + (FLOAT) 25.0*pow(s,3) - (FLOAT) 52.5*pow(s,4) + (FLOAT) 30.0*pow(s,5) - (FLOAT) 7.5*pow(s,6) +
               (FLOAT) 0.7142857143*pow(s,7) - 33.785714286;
      else if (s < (FLOAT) 3.0)
        return (FLOAT) 121.5*s*s - (FLOAT) 135.0*pow(s,3) + (FLOAT) 67.5*pow(s,4) - (FLOAT) 18.0*pow(s,5) + (FLOAT) 2.5*pow(s,6) -
               (FLOAT) 0.142857143*pow(s,7) - (FLOAT) 52.07142857;
      else return 0.0;
    }
    if (s < 1.0) return 1.4 - 2.0*s*s + 1.5*pow(s, 4) - 0.6*pow(s, 5);
    else if (s < 2.0) return 1.6 - 4.0*s*s + 4.0*pow(s, 3) - 1.5*pow(s, 4) + 0.2*pow(s, 5);
    else return 0.0;
  }
  FLOAT a_wgrav(FLOAT s) const {
    if (type == 1) {
      if (s < 1.0) return (12.0/359.0)*(22.0*s - 12.0*pow(s,3) + (30.0/7.0)*pow(s,5) - (5.0/4.0)*pow(s,6));
      else if (s < 2.0)
        return (12.0/359.0)*(17.0*s + (75.0/4.0)*s*s - 42.0*pow(s,3) + 25.0*pow(s,4) - (45.0/7.0)*pow(s,5) + (5.0/8.0)*pow(s,6) +
                             (5.0/56.0)/(s*s));
      else if (s < 3.0)
        return (12.0/359.0)*(81.0*s - (405.0/4.0)*pow_ref(s,2) + 54.0*pow(s,3) - 15.0*pow(s,4) + (15.0/7.0)*pow(s,5) -
                             (1.0/8.0)*pow(s,6) - (507.0/56.0)/(s*s));
      else return 1.0/(s*s);
    }
    if (s < 1.0) return 1.333333333333333333333*s - 1.2*pow(s, 3) + 0.5*pow(s, 4);
    else if (s < 2.0)
      return 2.6666666666666666667*s - 3.0*s*s + 1.2*pow(s, 3) - 0.166666666666666666667*pow(s, 4) -
             0.06666666666666666667/(s*s);
    else return 1.0/(s*s);
  }
  FLOAT a_wpot(FLOAT s) const {
    if (type == 1) {
      if (s < 1.0)
        return (12.0/359.0)*(-11.0*s*s + 3.0*pow(s,4) - (5.0/7.0)*pow(s,6) + (5.0/28.0)*pow(s,7) + (478.0/14.0));
      else if (s < 2.0)
        return (12.0/359.0)*(-(17.0/2.0)*s*s - (25.0/4.0)*pow(s,3) + (21.0/2.0)*pow(s,4) - 5.0*pow(s,5) + (15.0/14.0)*pow(s,6) -
                             (5.0/56.0)*pow(s,7) + (473.0/14.0) + (5.0/56.0)/s);
      else if (s < 3.0)
        return (12.0/359.0)*(-(81.0/2.0)*s*s + (135.0/4.0)*pow(s,3) - (27.0/2.0)*pow(s,4) + 3.0*pow(s,5) - (5.0/14.0)*pow(s,6) +
                             (1.0/56.0)*pow(s,7) + (729.0/14.0) - (507.0/56.0)/s);
      else return 1.0/s;
    }
    if (s < 1.0) return 1.4 - 0.666666666666666666666666*s*s + 0.3*pow(s, 4) - 0.1*pow(s, 5);
    else if (s < 2.0)
      return -1.0/(15.0*s) + 1.6 - 1.33333333333333333333333333*s*s + pow(s, 3) - 0.3*pow(s, 4) +
             (1.0/30.0)*pow(s, 5);
    else return 1.0/s;
  }
};

static inline FLOAT Dot(const FLOAT *a, const FLOAT *b, int nd)   // InlineFuncs.h:46-54
{
  if (nd == 1) return a[0]*b[0];
  else if (nd == 2) return a[0]*b[0] + a[1]*b[1];
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
}

static inline bool BoxOverlap(int nd, const FLOAT *b1min, const FLOAT *b1max, const FLOAT *b2min, const FLOAT *b2max)
{                                                                   // InlineFuncs.h:362-390
  for (int k = 0; k < nd; k++) {
    if (b1min[k] > b2max[k]) return false;
    if (b2min[k] > b1max[k]) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// KD-tree, KDTree.cpp
// ---------------------------------------------------------------------------------------------
struct KDTree {
  const Params *P; FLOAT kernrange;
  int Ntot = 0, ltot = 0, gtot = 0, Ncell = 0, ifirst = -1, ilast = -1;
  std::vector<Cell> cell;
  std::vector<int> ids, inext;

  void ComputeTreeSize() {                                          // KDTree.cpp:322-352
    ltot = 0;
    while (P->Nleafmax*pow(2, ltot) < Ntot) ltot++;
    gtot = (int) pow(2, ltot);
    Ncell = 2*gtot - 1;
  }
  void CreateTreeStructure() {                                      // KDTree.cpp:362-433
    std::vector<int> c2L(ltot + 1), cNL(ltot + 1);
    for (int l = 0; l < ltot; l++) { c2L[l] = (int) pow(2, ltot - l); cNL[l] = 2*c2L[l] - 1; }
    cell.assign(Ncell, Cell());
    for (int c = 0; c < Ncell; c++) {
      Cell &x = cell[c];
      memset(&x, 0, sizeof(Cell));
      x.copen = -1; x.cnext = -1; x.c1 = -1; x.c2 = -1; x.ifirst = -1; x.ilast = -1;
    }
    cell[0].level = 0;
    for (int c = 0; c < Ncell; c++) {
      if (cell[c].level == ltot) cell[c].cnext = c + 1;
      else {
        cell[c + 1].level = cell[c].level + 1;
        cell[c].copen = c + 1; cell[c].c1 = c + 1; cell[c].c2 = c + c2L[cell[c].level];
        cell[cell[c].c2].level = cell[c].level + 1;
        cell[c].cnext = c + cNL[cell[c].level];
      }
    }
  }
  FLOAT QuickSelect(int left, int right, int jpivot, int k, const std::vector<Part> &p) {   // KDTree.cpp:682-750
    int j, jguess, jtemp; FLOAT rpivot;
    do {
      jguess = (left + right)/2;
      rpivot = p[ids[jguess]].r[k];
      jtemp = ids[jguess]; ids[jguess] = ids[right]; ids[right] = jtemp;
      jguess = left;
      for (j = left; j < right; j++) {
        if (p[ids[j]].r[k] <= rpivot) { jtemp = ids[j]; ids[j] = ids[jguess]; ids[jguess] = jtemp; jguess++; }
      }
      jtemp = ids[right]; ids[right] = ids[jguess]; ids[jguess] = jtemp;
      if (jguess < jpivot) left = jguess + 1;
      else if (jguess > jpivot) right = jguess - 1;
    } while (jguess != jpivot);
    return rpivot;
  }
  // q += m (3 dr dr - |dr|^2 1), the five (3 / 1) independent components kept by KDTree.cpp:929-944
  static void AddQuad(FLOAT *q, FLOAT mi, const FLOAT *dr, int nd) {
    const FLOAT drsqd = Dot(dr, dr, nd);
    if (nd == 3) {
      q[0] += mi*((FLOAT) 3.0*dr[0]*dr[0] - drsqd);
      q[1] += mi*(FLOAT) 3.0*dr[0]*dr[1];
      q[2] += mi*((FLOAT) 3.0*dr[1]*dr[1] - drsqd);
      q[3] += mi*(FLOAT) 3.0*dr[2]*dr[0];
      q[4] += mi*(FLOAT) 3.0*dr[2]*dr[1];
    }
    else if (nd == 2) {
      q[0] += mi*((FLOAT) 3.0*dr[0]*dr[0] - drsqd);
      q[1] += mi*(FLOAT) 3.0*dr[0]*dr[1];
      q[2] += mi*((FLOAT) 3.0*dr[1]*dr[1] - drsqd);
    }
    else q[0] += mi*((FLOAT) 3.0*dr[0]*dr[0] - drsqd);
  }
  void StockCellProperties(Cell &c, const std::vector<Part> &p) {   // KDTree.cpp:808-1083 (geometric MAC)
    const int nd = P->ndim;
    const bool need_quad = P->multipole == 1 || P->multipole == 3 || P->gravity_mac == 2;      // KDTree.cpp:823-824
    FLOAT dr[3];
    for (int k = 0; k < 5; k++) c.q[k] = 0.0;
    c.amin = big_number; c.macfactor = 0; c.mac = 0.0;
    c.Nactive = 0; c.N = 0; c.m = 0.0; c.hmax = 0.0; c.rmax = 0.0; c.cdistsqd = big_number;
    for (int k = 0; k < nd; k++) { c.r[k] = 0.0; c.v[k] = 0.0; c.rcell[k] = 0.0; c.bbmin[k] = big_number; c.bbmax[k] = -big_number;
                                   c.hbmin[k] = big_number; c.hbmax[k] = -big_number; }
    if (c.level == ltot) {
      int i = c.ifirst;
      while (i != -1) {
        c.N++;
        if (p[i].flags & F_ACTIVE) c.Nactive++;
        c.hmax = std::max(c.hmax, p[i].h);
        c.m += p[i].m;
        if (P->gravity_mac == 1) c.amin = std::min(c.amin, (FLOAT) sqrt(Dot(p[i].atree, p[i].atree, nd)));   // KDTree.cpp:899-901
        else if (P->gravity_mac == 2) c.macfactor = std::max(c.macfactor, (FLOAT) pow(p[i].gpot, -twothirds));
        for (int k = 0; k < nd; k++) c.r[k] += p[i].m*p[i].r[k];
        for (int k = 0; k < nd; k++) c.v[k] += p[i].m*p[i].v[k];
        for (int k = 0; k < nd; k++) {
          if (p[i].r[k] < c.bbmin[k]) c.bbmin[k] = p[i].r[k];
          if (p[i].r[k] > c.bbmax[k]) c.bbmax[k] = p[i].r[k];
          if (p[i].r[k] - kernrange*p[i].h < c.hbmin[k]) c.hbmin[k] = p[i].r[k] - kernrange*p[i].h;
          if (p[i].r[k] + kernrange*p[i].h > c.hbmax[k]) c.hbmax[k] = p[i].r[k] + kernrange*p[i].h;
        }
        if (i == c.ilast) break;
        i = inext[i];
      }
      if (c.m > 0) { for (int k = 0; k < nd; k++) c.r[k] /= c.m; for (int k = 0; k < nd; k++) c.v[k] /= c.m; }
      if (c.N > 0) {
        for (int k = 0; k < nd; k++) c.rcell[k] = 0.5*(c.bbmin[k] + c.bbmax[k]);
        for (int k = 0; k < nd; k++) dr[k] = 0.5*(c.bbmax[k] - c.bbmin[k]);
        c.cdistsqd = std::max(Dot(dr, dr, nd), c.hmax*c.hmax)/P->thetamaxsqd;
        c.rmax = sqrt(Dot(dr, dr, nd));
      }
      if (need_quad) {                                               // KDTree.cpp:921-950
        int i = c.ifirst;
        while (i != -1) {
          for (int k = 0; k < nd; k++) dr[k] = p[i].r[k] - c.r[k];
          AddQuad(c.q, p[i].m, dr, nd);
          if (i == c.ilast) break;
          i = inext[i];
        }
      }
    }
    else {
      const Cell &c1 = cell[c.copen], &c2 = cell[cell[c.copen].cnext];
      const Cell *ch[2] = {&c1, &c2};
      for (int q = 0; q < 2; q++) {
        if (ch[q]->N > 0) {
          for (int k = 0; k < nd; k++) { c.bbmin[k] = std::min(ch[q]->bbmin[k], c.bbmin[k]); c.bbmax[k] = std::max(ch[q]->bbmax[k], c.bbmax[k]);
                                         c.hbmin[k] = std::min(ch[q]->hbmin[k], c.hbmin[k]); c.hbmax[k] = std::max(ch[q]->hbmax[k], c.hbmax[k]); }
          c.hmax = std::max(c.hmax, ch[q]->hmax);
          c.amin = std::min(c.amin, ch[q]->amin);                    // KDTree.cpp:968, 982
          c.macfactor = std::max(c.macfactor, ch[q]->macfactor);
        }
      }
      c.N = c1.N + c2.N; c.Nactive = c1.Nactive + c2.Nactive; c.m = c1.m + c2.m;
      if (c.m > 0) {
        for (int k = 0; k < nd; k++) c.r[k] = (c1.m*c1.r[k] + c2.m*c2.r[k])/c.m;
        for (int k = 0; k < nd; k++) c.v[k] = (c1.m*c1.v[k] + c2.m*c2.v[k])/c.m;
      }
      if (c.N > 0) {
        for (int k = 0; k < nd; k++) c.rcell[k] = 0.5*(c.bbmin[k] + c.bbmax[k]);
        for (int k = 0; k < nd; k++) dr[k] = 0.5*(c.bbmax[k] - c.bbmin[k]);
        c.cdistsqd = std::max(Dot(dr, dr, nd), c.hmax*c.hmax)/P->thetamaxsqd;
        c.rmax = sqrt(Dot(dr, dr, nd));
      }
      if (need_quad) {                                               // KDTree.cpp:1004-1052
        for (int q = 0; q < 2; q++) {
          if (!(ch[q]->m > 0)) continue;
          for (int k = 0; k < nd; k++) dr[k] = ch[q]->r[k] - c.r[k];
          const int nq = nd == 3 ? 5 : (nd == 2 ? 3 : 1);
          for (int k = 0; k < nq; k++) c.q[k] += ch[q]->q[k];
          AddQuad(c.q, ch[q]->m, dr, nd);
        }
      }
    }
    if (P->gravity_mac == 2) {                                       // KDTree.cpp:1054-1076
      FLOAT lambda, pp;
      if (nd == 3) {
        pp = c.q[0]*c.q[2] - (c.q[0] + c.q[2])*(c.q[0] + c.q[2]) - c.q[1]*c.q[1] - c.q[3]*c.q[3] - c.q[4]*c.q[4];
        if (pp >= (FLOAT) 0.0) lambda = 0; else lambda = (FLOAT) 2.0*sqrt(-pp/(FLOAT) 3.0);
      }
      else if (nd == 2) { pp = (c.q[0] - c.q[2])*(c.q[0] - c.q[2]) + 4*c.q[1]*c.q[1]; lambda = 0.5*std::max(c.q[0] + c.q[2] + sqrt(pp), 0.); }
      else lambda = fabs(c.q[0]);
      c.mac = pow((FLOAT) 0.5*lambda/P->macerror, (FLOAT) 0.66666666666666);
    }
  }
  void DivideTreeCell(int first, int last, const std::vector<Part> &p, Cell &c) {   // KDTree.cpp:442-595
    const int nd = P->ndim;
    if (c.level == ltot) {
      if (c.N > 0) {
        for (int j = c.ifirst; j < c.ilast; j++) inext[ids[j]] = ids[j + 1];
        c.ifirst = ids[c.ifirst]; c.ilast = ids[c.ilast];
      }
      else { c.ifirst = -1; c.ilast = -1; }
      StockCellProperties(c, p);
      return;
    }
    int k_divide = 0; FLOAT rkmax = 0.0;
    for (int k = 0; k < nd; k++) if (c.bbmax[k] - c.bbmin[k] > rkmax) { rkmax = c.bbmax[k] - c.bbmin[k]; k_divide = k; }
    const FLOAT rdivide = QuickSelect(c.ifirst, c.ilast, c.ifirst + c.N/2, k_divide, p);
    Cell &a = cell[c.c1], &b = cell[c.c2];
    for (int k = 0; k < nd; k++) { a.bbmin[k] = c.bbmin[k]; a.bbmax[k] = c.bbmax[k]; b.bbmin[k] = c.bbmin[k]; b.bbmax[k] = c.bbmax[k]; }
    a.bbmax[k_divide] = rdivide; a.N = c.N/2;
    if (a.N != 0) { a.ifirst = first; a.ilast = first + c.N/2 - 1; }
    b.bbmin[k_divide] = rdivide; b.N = c.N - a.N;
    if (b.N != 0) { b.ifirst = first + c.N/2; b.ilast = last; }
    DivideTreeCell(first, first + c.N/2 - 1, p, a);
    DivideTreeCell(first + c.N/2, last, p, b);
    if (a.N > 0) { c.ifirst = a.ifirst; inext[a.ilast] = b.ifirst; }
    else c.ifirst = b.ifirst;
    c.ilast = b.ilast;
    StockCellProperties(c, p);
  }
  // KDTree::StockTree, KDTree.cpp:760-798: re-stock every cell bottom-up, membership unchanged
  void StockTree(Cell &c, const std::vector<Part> &p) {
    if (c.level != ltot) { StockTree(cell[c.c1], p); StockTree(cell[c.c2], p); }
    StockCellProperties(c, p);
  }
  void BuildTree(int _ifirst, int Npart, const std::vector<Part> &p) {              // KDTree.cpp:220-313
    const int nd = P->ndim;
    Ntot = Npart;
    ComputeTreeSize();
    CreateTreeStructure();
    if ((int) ids.size() < (int) p.size()) { ids.resize(p.size()); inext.resize(p.size()); }
    FLOAT bbmin[3], bbmax[3];
    for (int k = 0; k < nd; k++) { bbmin[k] = big_number; bbmax[k] = -big_number; }
    if (Npart > 0) {
      ifirst = _ifirst; ilast = _ifirst + Npart - 1;
      for (int i = ifirst; i <= ilast; i++)
        for (int k = 0; k < nd; k++) {
          bbmax[k] = std::max(bbmax[k], p[i].r[k] + kernrange*p[i].h);
          bbmin[k] = std::min(bbmin[k], p[i].r[k] - kernrange*p[i].h);
        }
      for (int i = ifirst; i <= ilast; i++) ids[i] = i;
      for (int i = ifirst; i < ilast; i++) inext[i] = i + 1;
      inext[ilast] = -1;
    }
    else { ifirst = -1; ilast = -1; }
    cell[0].N = Ntot; cell[0].ifirst = ifirst; cell[0].ilast = ilast; cell[0].hmax = 0;
    for (int k = 0; k < nd; k++) { cell[0].bbmin[k] = bbmin[k]; cell[0].bbmax[k] = bbmax[k]; }
    if (Ntot > 0) DivideTreeCell(ifirst, ilast, p, cell[0]);
  }
  void UpdateHmaxValues(Cell &c, const std::vector<Part> &p) {      // KDTree.cpp:1128-1208
    const int nd = P->ndim;
    if (c.level != ltot) { UpdateHmaxValues(cell[c.c1], p); UpdateHmaxValues(cell[c.c2], p); }
    c.hmax = 0.0;
    for (int k = 0; k < nd; k++) { c.hbmin[k] = big_number; c.hbmax[k] = -big_number; }
    if (c.level == ltot) {
      int i = c.ifirst;
      while (i != -1) {
        c.hmax = std::max(c.hmax, p[i].h);
        for (int k = 0; k < nd; k++) {
          if (p[i].r[k] - kernrange*p[i].h < c.hbmin[k]) c.hbmin[k] = p[i].r[k] - kernrange*p[i].h;
          if (p[i].r[k] + kernrange*p[i].h > c.hbmax[k]) c.hbmax[k] = p[i].r[k] + kernrange*p[i].h;
        }
        if (i == c.ilast) break;
        i = inext[i];
      }
    }
    else {
      const Cell *ch[2] = {&cell[c.c1], &cell[c.c2]};
      for (int q = 0; q < 2; q++) if (ch[q]->N > 0) {
        c.hmax = std::max(c.hmax, ch[q]->hmax);
        for (int k = 0; k < nd; k++) { c.hbmin[k] = std::min(ch[q]->hbmin[k], c.hbmin[k]); c.hbmax[k] = std::max(ch[q]->hbmax[k], c.hbmax[k]); }
      }
    }
  }
  // Tree::ComputeGatherNeighbourList (cell version), Tree.cpp:291-381
  void GatherList(const Cell &c, const std::vector<Part> &p, FLOAT hmax, std::vector<int> &list) const {
    if (Ncell == 0 || Ntot == 0) return;
    const int nd = P->ndim;
    const size_t start = list.size();
    FLOAT gmin[3], gmax[3];
    const FLOAT hrangemaxsqd = pow(c.rmax + kernrange*hmax, 2);
    for (int k = 0; k < nd; k++) { gmin[k] = c.bbmin[k] - kernrange*hmax; gmax[k] = c.bbmax[k] + kernrange*hmax; }
    int cc = 0;
    while (cc < Ncell) {
      if (BoxOverlap(nd, gmin, gmax, cell[cc].bbmin, cell[cc].bbmax)) {
        if (cell[cc].copen != -1) cc = cell[cc].copen;
        else if (cell[cc].N == 0) cc = cell[cc].cnext;
        else {
          int i = cell[cc].ifirst;
          while (i != -1) { list.push_back(i); if (i == cell[cc].ilast) break; i = inext[i]; }
          cc = cell[cc].cnext;
        }
      }
      else cc = cell[cc].cnext;
    }
    size_t keep = start;
    FLOAT dr[3];
    for (size_t j = start; j < list.size(); j++) {
      const int i = list[j];
      for (int k = 0; k < nd; k++) dr[k] = p[i].r[k] - c.rcell[k];
      if (Dot(dr, dr, nd) < hrangemaxsqd) list[keep++] = i;
    }
    list.resize(keep);
  }
  // Tree::ComputeGatherNeighbourList (point version), Tree.cpp:208-280
  void GatherPoint(const FLOAT *rp, FLOAT rsearch, const std::vector<Part> &p, std::vector<int> &list) const {
    if (Ncell == 0 || Ntot == 0) return;
    const int nd = P->ndim;
    const FLOAT rsearchsqd = rsearch*rsearch;
    FLOAT dr[3];
    int cc = 0;
    while (cc < Ncell) {
      for (int k = 0; k < nd; k++) dr[k] = cell[cc].rcell[k] - rp[k];
      if (Dot(dr, dr, nd) < (rsearch + cell[cc].rmax)*(rsearch + cell[cc].rmax)) {
        if (cell[cc].copen != -1) cc = cell[cc].copen;
        else if (cell[cc].N == 0) cc = cell[cc].cnext;
        else {
          int i = cell[cc].ifirst;
          while (i != -1) {
            for (int k = 0; k < nd; k++) dr[k] = p[i].r[k] - rp[k];
            if (Dot(dr, dr, nd) < rsearchsqd) list.push_back(i);
            if (i == cell[cc].ilast) break;
            i = inext[i];
          }
          cc = cell[cc].cnext;
        }
      }
      else cc = cell[cc].cnext;
    }
  }
};

// ---------------------------------------------------------------------------------------------
// star particle as the gas force routines see it (hybrid gas + N-body runs; NbodyParticle.h)
struct GasStar { FLOAT r[3], a[3], m, h, gpot; };

struct Oracle {
  std::vector<GasStar> stars;
  int star_softening = 1;                         // nbody_softening
  std::vector<Sink> sinks;                        // Sinks::sink
  FLOAT mmean = 0.0;                              // Hydrodynamics::mmean
  Params P; M4 kern; FLOAT invndim;
  int mac_now;                          // MAC in force (geometric during the setup bootstrap, SphSimulation.cpp:381-388)
  std::vector<Part> p;          // [0,Nhydro) real, then periodic ghosts
  int Nhydro = 0, Nghost = 0;
  KDTree tree, ghosttree;
  int n = 0, Nsteps = 0; double t = 0.0, timestep = 0.0;
  bool rebuild_tree = true;
  int nresync = 0, level_max = 0, level_step = 1, integration_step = 1; double dt_max = 0.0;   // Simulation.cpp:159-198
  std::string err;
  explicit Oracle(const Params &pp) : P(pp), kern(pp.ndim, pp.kernel & 1, (pp.kernel >> 1) & 1), invndim(1.0/pp.ndim) {
    tree.P = &P; ghosttree.P = &P; tree.kernrange = kern.kernrange; ghosttree.kernrange = kern.kernrange;
    mac_now = P.gravity_mac;
  }
  FLOAT h_rho_func(FLOAT m, FLOAT rho) const { return P.h_fac*pow(m/rho, invndim); }   // Sph.h:259
  FLOAT h_rho_deriv(FLOAT h, FLOAT rho) const { return -invndim*h/rho; }               // Sph.h:264
  bool any_periodic() const { return P.periodic[0] || P.periodic[1] || P.periodic[2]; }
  bool any_mirror() const { for (int k = 0; k < P.ndim; k++) if (P.mirror[k][0] || P.mirror[k][1]) return true; return false; }
  bool any_special() const { return any_periodic() || any_mirror(); }

  // ---- ghosts: HydroTree::SearchBoundaryGhostParticles HydroTree.cpp:495-543, Tree.cpp:1098-1149,
  //      Hydrodynamics.cpp:217-289
  void CheckBoundaryGhostParticle(int i, int j) {
    const FLOAT r = p[i].r[j], h = p[i].h;
    const FLOAT v = p[i].v[j];                                       // Hydrodynamics.cpp:217-250 (tghost = 0)
    if (r < P.boxmin[j] + ghost_range*kern.kernrange*h) {
      if (P.periodic[j]) CreateGhost(i, j, r + P.boxsize[j], v);
      if (P.mirror[j][0]) CreateGhost(i, j, 2*P.boxmin[j] - r, -v);
    }
    if (r > P.boxmax[j] - ghost_range*kern.kernrange*h) {
      if (P.periodic[j]) CreateGhost(i, j, r - P.boxsize[j], v);
      if (P.mirror[j][1]) CreateGhost(i, j, 2*P.boxmax[j] - r, -v);
    }
  }
  void CreateGhost(int i, int k, FLOAT rk, FLOAT vk) {
    Part g = p[i];
    g.r[k] = rk; g.v[k] = vk; g.flags &= ~F_ACTIVE; g.iorig = i;
    p.push_back(g);
    Nghost++;
  }
  void SearchBoundaryGhostParticles() {
    p.resize(Nhydro); Nghost = 0;
    if (!any_special()) return;
    const FLOAT grange = ghost_range*kern.kernrange;
    int Ntot = Nhydro;
    for (int j = 0; j < P.ndim; j++) {
      if (!P.periodic[j] && !P.mirror[j][0] && !P.mirror[j][1]) continue;
      int c = 0;
      while (c < tree.Ncell) {                                       // Tree::GenerateBoundaryGhostParticles (tghost = 0)
        const Cell &x = tree.cell[c];
        if (x.bbmin[j] < P.boxmin[j] + grange*x.hmax || x.bbmax[j] > P.boxmax[j] - grange*x.hmax) {
          if (x.copen != -1) c = x.copen;
          else if (x.N == 0) c = x.cnext;
          else {
            int i = x.ifirst;
            while (i != -1) { CheckBoundaryGhostParticle(i, j); if (i == x.ilast) break; i = tree.inext[i]; }
            c = x.cnext;
          }
        }
        else c = x.cnext;
      }
      if (j > 0) for (int i = Nhydro; i < Ntot; i++) CheckBoundaryGhostParticle(i, j);
      Ntot = Nhydro + Nghost;
    }
  }
  // Hydrodynamics::DoDeleteDeadParticles, Hydrodynamics.h:158-202: every dead slot takes the last particle of the array
  int DeleteDeadParticles() {
    int Ndead = 0, ilast = Nhydro;
    for (int i = 0; i < Nhydro; i++) {
      int itype = p[i].flags;
      while (itype & F_DEAD) {
        Ndead++; ilast--;
        if (i < ilast) { p[i] = p[ilast]; p[ilast].flags |= F_DEAD; p[ilast].m = 0.0; }
        else break;
        itype = p[i].flags;
      }
      if (i >= ilast - 1) break;
    }
    if (Ndead == 0) return 0;
    Nhydro -= Ndead;
    return Ndead;
  }
  void BuildTree() {                                                 // HydroTree::BuildTree, HydroTree.cpp:332-343
    p.resize(Nhydro); Nghost = 0;
    if (P.sink_particles) { DeleteDeadParticles(); p.resize(Nhydro); }
    tree.BuildTree(0, Nhydro, p);
  }
  // HydroTree::BuildTree inside MainLoop (HydroTree.cpp:310-372): rebuild every ntreebuildstep steps, re-stock otherwise
  // (ntreestockstep = 1; ExtrapolateCellProperties is not restated)
  void StepTree() {
    // rebuild_tree is raised by the setup and only lowered at the end of a MainLoop call (SphSimulation.cpp:866): the
    // first step after the setup always rebuilds
    if (P.ntreebuildstep <= 1 || Nsteps%P.ntreebuildstep == 0 || rebuild_tree) { BuildTree(); return; }
    p.resize(Nhydro); Nghost = 0;
    if (Nsteps%P.ntreestockstep == 0) { tree.StockTree(tree.cell[0], p); return; }
    // Tree::ExtrapolateCellProperties, Tree.cpp:172-198: cells drift with their stocked mean velocity
    for (int c = 0; c < tree.Ncell; c++) {
      Cell &x = tree.cell[c];
      for (int k = 0; k < P.ndim; k++) {
        const FLOAT dx = x.v[k]*timestep;
        x.r[k] += dx; x.rcell[k] += dx; x.bbmin[k] += dx; x.bbmax[k] += dx; x.hbmin[k] += dx; x.hbmax[k] += dx;
      }
    }
  }
  void BuildGhostTree() { ghosttree.BuildTree(Nhydro, Nghost, p); }

  // ---- GradhSph::ComputeH, GradhSph.cpp:142-326 (no sinks, no stars) + ComputeThermalProperties :335-347
  int ComputeH(Part &pi, FLOAT hmax, const std::vector<int> &ngb2) const {
    const int nd = P.ndim;
    int iteration = 0; const int iteration_max = 30;
    FLOAT dr[3], h_lower_bound = 0.0, h_upper_bound = hmax, invh, invhsqd, ssqd;
    if (P.sink_particles) {                                            // GradhSph.cpp:163-169
      h_lower_bound = P.h_fac*pow(pi.m/P.rho_sink, invndim);
      if (hmax < h_lower_bound) return -1;
    }
    const int Nneib = (int) ngb2.size();
    do {
      iteration++;
      invh = 1.0/pi.h;
      pi.rho = 0.0; pi.invomega = 0.0; pi.zeta = 0.0;
      pi.hfactor = pow_ref(invh, nd);
      invhsqd = invh*invh;
      for (int j = 0; j < Nneib; j++) {
        const Part &ngb = p[ngb2[j]];
        for (int k = 0; k < nd; k++) dr[k] = ngb.r[k] - pi.r[k];
        ssqd = invhsqd*Dot(dr, dr, nd);
        pi.rho += ngb.m*kern.w0_s2(ssqd);                           // GradhSph.cpp:201-203
        pi.invomega += ngb.m*invh*kern.womega_s2(ssqd);
        pi.zeta += ngb.m*kern.wzeta_s2(ssqd);
      }
      pi.rho *= pi.hfactor; pi.invomega *= pi.hfactor; pi.zeta *= invhsqd;
      if (pi.rho > 0.0 && pi.h > h_lower_bound && fabs(pi.h - h_rho_func(pi.m, pi.rho))*invh < P.h_converge) break;
      if (iteration < iteration_max) pi.h = h_rho_func(pi.m, pi.rho);
      else if (iteration == iteration_max) pi.h = 0.5*(h_lower_bound + h_upper_bound);
      else if (iteration < 5*iteration_max) {
        if (pi.rho < small_number || pi.h > h_rho_func(pi.m, pi.rho)) h_upper_bound = pi.h;
        else h_lower_bound = pi.h;
        pi.h = 0.5*(h_lower_bound + h_upper_bound);
      }
      else return -2;
      if (pi.h > hmax) return 0;
    } while (pi.h > h_lower_bound && pi.h < h_upper_bound);
    pi.h = std::max(h_rho_func(pi.m, pi.rho), h_lower_bound);
    invh = 1/pi.h;
    pi.hfactor = pow_ref(invh, nd + 1);
    pi.hrangesqd = kern.kernrangesqd*pi.h*pi.h;
    pi.div_v = 0.0;
    // potential-minimum flag for the sink search, GradhSph.cpp:270-280 - as written there: the distance tested for
    // neighbour j is that of the neighbour before it (dr is refreshed after drsqd is taken; for j = 0 it is what the
    // density loop left behind), and invhsqd is still that of the last h iteration
    if (P.create_sinks == 1) {
      pi.flags |= F_POTMIN;
      for (int j = 0; j < Nneib; j++) {
        const Part &ngb = p[ngb2[j]];
        const FLOAT drsqd = Dot(dr, dr, nd);
        for (int k = 0; k < nd; k++) dr[k] = ngb.r[k] - pi.r[k];
        if (ngb.gpot > (FLOAT) 1.000000001*pi.gpot && drsqd*invhsqd < kern.kernrangesqd) pi.flags &= ~F_POTMIN;
      }
    }
    if (pi.sinkid != -1) {                                             // inside a sink: GradhSph.cpp:309-312
      pi.invomega = (FLOAT) 1.0; pi.zeta = (FLOAT) 0.0;
      Thermal(pi);
      if (P.tdavisc == 2) CullenDehnen(pi, ngb2);
      return pi.h <= hmax ? 1 : -1;
    }
    pi.invomega = 1.0 - h_rho_deriv(pi.h, pi.rho)*pi.invomega;
    pi.invomega = 1.0/pi.invomega;
    // Hubber et al. (2013) SPH-star conservative-gravity term in zeta (conservative_sph_star_gravity = 1, the default),
    // GradhSph.cpp:288-307: mean-h softening for kernel-softened stars, h/2 otherwise
    if (!stars.empty()) {
      FLOAT invhsqd_s = (FLOAT) 4.0*invh*invh;
      for (size_t j = 0; j < stars.size(); j++) {
        if (star_softening == 1) invhsqd_s = pow((FLOAT) 2.0/(pi.h + stars[j].h), 2);
        for (int k = 0; k < nd; k++) dr[k] = stars[j].r[k] - pi.r[k];
        ssqd = Dot(dr, dr, nd)*invhsqd_s;
        pi.zeta += stars[j].m*invhsqd_s*kern.wzeta_s2(ssqd);
      }
    }
    pi.zeta = h_rho_deriv(pi.h, pi.rho)*pi.zeta*pi.invomega;
    // energy_eqn EOS: AdiabaticEOS.cpp:69-82, EOS.h:156
    Thermal(pi);
    if (P.tdavisc == 2) CullenDehnen(pi, ngb2);                        // GradhSph.cpp:319-321
    return pi.h <= hmax ? 1 : -1;
  }

  // GradhSph::ComputeThermalProperties (GradhSph.cpp:335-347) with the three closed-form EOS in dimensionless units:
  // energy_eqn (AdiabaticEOS.cpp:69-82), isothermal (IsothermalEOS.cpp:72-87), barotropic (BarotropicEOS.cpp:78-91)
  void Thermal(Part &pi) const {
    const FLOAT gammam1 = P.gamma - 1.0;
    if (P.gas_eos == 0) { pi.sound = sqrt(P.gamma*gammam1*pi.u); pi.pressure = gammam1*pi.rho*pi.u; return; }
    if (P.gas_eos == 2) pi.u = P.temp0*(1.0 + pow(pi.rho*((FLOAT) 1.0/P.rho_bary), gammam1))/gammam1/P.mu_bar;
    else pi.u = P.temp0/gammam1/P.mu_bar;
    pi.sound = sqrt(gammam1*pi.u);
    pi.pressure = gammam1*pi.rho*pi.u;
  }

  // Sph::ComputeCullenAndDehnenViscosity, Sph.h:364-456 (time_dependent_avisc = cd2010): integral-gradient estimates of
  // grad v and grad a over the gather list, d(div v)/dt, Balsara-type limiter, alpha_loc; InvertMatrix InlineFuncs.h:577-606
  void CullenDehnen(Part &pi, const std::vector<int> &ngb2) const {
    const int nd = P.ndim;
    FLOAT dv[3][3], da[3][3], rr[3][3], dvdx[3][3], dadx[3][3], T[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rr[i][j] = da[i][j] = dv[i][j] = dadx[i][j] = dvdx[i][j] = T[i][j] = 0;
    const FLOAT invh = 1/pi.h;
    const FLOAT hfac = invh*pi.hfactor/pi.rho;
    for (size_t n = 0; n < ngb2.size(); n++) {
      const Part &g = p[ngb2[n]];
      FLOAT dr[3];
      for (int j = 0; j < nd; j++) dr[j] = g.r[j] - pi.r[j];
      const FLOAT w = g.m*hfac*kern.w1(invh*sqrt(Dot(dr, dr, nd)));
      for (int j = 0; j < nd; j++)
        for (int k = 0; k < nd; k++) {
          rr[j][k] += w*dr[j]*dr[k];
          dv[j][k] += w*dr[j]*(g.v[k] - pi.v[k]);
          da[j][k] += w*dr[j]*(g.a[k] - pi.a[k]);
        }
    }
    const FLOAT (*A)[3] = rr;
    if (nd == 1) T[0][0] = 1/A[0][0];
    else if (nd == 2) {
      const FLOAT invdet = (FLOAT) 1.0/(A[0][0]*A[1][1] - A[0][1]*A[1][0]);
      T[0][0] = invdet*A[1][1]; T[0][1] = -invdet*A[0][1]; T[1][0] = -invdet*A[1][0]; T[1][1] = invdet*A[0][0];
    }
    else {
      const FLOAT invdet = (FLOAT) 1.0/(A[0][0]*(A[1][1]*A[2][2] - A[2][1]*A[1][2]) - A[0][1]*(A[1][0]*A[2][2] - A[1][2]*A[2][0]) +
                                        A[0][2]*(A[1][0]*A[2][1] - A[1][1]*A[2][0]));
      T[0][0] = (A[1][1]*A[2][2] - A[2][1]*A[1][2])*invdet;
      T[0][1] = (A[0][2]*A[2][1] - A[0][1]*A[2][2])*invdet;
      T[0][2] = (A[0][1]*A[1][2] - A[0][2]*A[1][1])*invdet;
      T[1][0] = (A[1][2]*A[2][0] - A[1][0]*A[2][2])*invdet;
      T[1][1] = (A[0][0]*A[2][2] - A[0][2]*A[2][0])*invdet;
      T[1][2] = (A[1][0]*A[0][2] - A[0][0]*A[1][2])*invdet;
      T[2][0] = (A[1][0]*A[2][1] - A[2][0]*A[1][1])*invdet;
      T[2][1] = (A[2][0]*A[0][1] - A[0][0]*A[2][1])*invdet;
      T[2][2] = (A[0][0]*A[1][1] - A[1][0]*A[0][1])*invdet;
    }
    double modR = 0, modT = 0;
    for (int i = 0; i < nd; i++) for (int j = 0; j < nd; j++) { modR += rr[i][j]*rr[i][j]; modT += T[i][j]*T[i][j]; }
    const double sqd_condition_number = modR*modT/(nd*nd);
    FLOAT alpha_loc = 0;
    if (sqd_condition_number > 1e4) alpha_loc = P.alpha_visc;
    else {
      for (int i = 0; i < nd; i++) for (int j = 0; j < nd; j++) for (int k = 0; k < nd; k++) {
        dvdx[i][j] += T[j][k]*dv[k][i];
        dadx[i][j] += T[j][k]*da[k][i];
      }
      FLOAT ddivdt = 0, divv2 = 0;
      for (int i = 0; i < nd; ++i) {
        ddivdt += dadx[i][i];
        for (int j = 0; j < nd; ++j) ddivdt -= dvdx[i][j]*dvdx[j][i];
        divv2 += dvdx[i][i];
      }
      divv2 *= divv2;
      FLOAT curlv2 = 0;                                              // CurlVelSqd, Sph.h:344-358
      if (nd == 2) { const FLOAT c = dvdx[1][0] - dvdx[0][1]; curlv2 = c*c; }
      else if (nd == 3) {
        const FLOAT c[3] = {dvdx[1][2] - dvdx[2][1], dvdx[2][0] - dvdx[0][2], dvdx[0][1] - dvdx[1][0]};
        curlv2 = Dot(c, c, 3);
      }
      FLOAT f_balsara = 1;
      if (curlv2 > 0) f_balsara = divv2/(divv2 + curlv2);
      if (ddivdt < 0) {
        alpha_loc = (10*pi.h*pi.h/(pi.sound*pi.sound))*f_balsara*(-ddivdt);
        alpha_loc = std::min(alpha_loc, P.alpha_visc);
      }
    }
    if (alpha_loc > pi.alpha) pi.alpha = alpha_loc;
    pi.dalphadt = (FLOAT) 0.1*pi.sound*(std::max(P.alpha_visc_min, alpha_loc) - pi.alpha)*invh;
  }

  std::vector<int> ActiveLeafCells() const {                         // Tree::ComputeActiveCellList, Tree.cpp:91-115
    std::vector<int> out;
    for (int c = 0; c < tree.Ncell; c++)
      if (tree.cell[c].N <= P.Nleafmax && tree.cell[c].copen == -1 && tree.cell[c].Nactive > 0) out.push_back(c);
    return out;
  }
  int ActiveParticles(const Cell &c, int *list) const {              // Tree.cpp:60-81
    int i = c.ifirst, N = 0;
    while (i != -1) {
      if (i < Nhydro && (p[i].flags & F_ACTIVE)) list[N++] = i;
      if (i == c.ilast) break;
      i = tree.inext[i];
    }
    return N;
  }

  // ---- GradhSphTree::UpdateAllSphProperties, GradhSphTree.cpp:83-271
  int UpdateAllSphProperties() {
    const std::vector<int> cl = ActiveLeafCells();
    const int nd = P.ndim;
    int bad = 0;
#pragma omp parallel for schedule(guided) reduction(+:bad) num_threads(P.nthreads)
    for (int cc = 0; cc < (int) cl.size(); cc++) {
      const Cell cellc = tree.cell[cl[cc]];
      std::vector<int> neiblist, ngb2;
      int activelist[64]; Part activepart[64];
      FLOAT hmax = cellc.hmax;
      int celldone, Nactive;
      do {
        hmax = 1.05*hmax;
        celldone = 1;
        Nactive = ActiveParticles(cellc, activelist);
        for (int j = 0; j < Nactive; j++) activepart[j] = p[activelist[j]];
        neiblist.clear();
        tree.GatherList(cellc, p, hmax, neiblist);
        ghosttree.GatherList(cellc, p, hmax, neiblist);
        for (int j = 0; j < Nactive; j++) {
          const FLOAT hrangesqd = kern.kernrangesqd*hmax*hmax;
          ngb2.clear();
          FLOAT draux[3];
          for (size_t jj = 0; jj < neiblist.size(); jj++) {
            for (int k = 0; k < nd; k++) draux[k] = p[neiblist[jj]].r[k] - activepart[j].r[k];
            const FLOAT drsqdaux = Dot(draux, draux, nd) + small_number;
            if (drsqdaux <= hrangesqd) ngb2.push_back(neiblist[jj]);
          }
          const int ok = ComputeH(activepart[j], hmax, ngb2);
          if (ok == -2) bad++;
          if (ok == 0) { celldone = 0; break; }
        }
      } while (celldone == 0);
      // the write-back only touches fields ComputeH changes (the reference copies whole particles, :244)
      for (int j = 0; j < Nactive; j++) p[activelist[j]] = activepart[j];
    }
    tree.UpdateHmaxValues(tree.cell[0], p);                         // :268
    return bad;
  }

  void ZeroAccelerations() {                                        // Sph.cpp:126-140
    for (int i = 0; i < Nhydro; i++) {
      if (!(p[i].flags & F_ACTIVE)) continue;
      p[i].levelneib = 0;
      p[i].div_v = 0.0; p[i].dudt = 0.0; p[i].gpot = 0.0; p[i].gpot_hydro = 0.0;
      for (int k = 0; k < 3; k++) { p[i].a[k] = 0.0; p[i].atree[k] = 0.0; }
    }
  }

  // nearest periodic image helpers, GhostNeighbours.hpp:119-166, 343-357
  bool NearestPeriodicVector(FLOAT *dr) const {
    bool any = false;
    for (int k = 0; k < P.ndim; k++) if (P.periodic[k]) {
      if (dr[k] > P.boxhalf[k]) { dr[k] -= P.boxsize[k]; any = true; }
      else if (dr[k] < -P.boxhalf[k]) { dr[k] += P.boxsize[k]; any = true; }
    }
    return any;
  }
  // GhostNeighbourFinder::ConstructGhostsScatterGather + _MakeReflectedScatterGatherGhosts, GhostNeighbours.hpp:253-268, 404-450
  void ConstructGhostsScatterGather(const Part &src, const Cell &cellc, std::vector<Part> &ngbs) const {
    ngbs.push_back(src);
    if (any_periodic()) MakePeriodicGhost(ngbs.back(), cellc.rcell);
    if (!any_mirror()) return;
    int nc = 1;
    const size_t old_size = ngbs.size() - 1;
    const Part real_particle = ngbs.back();
    const FLOAT h2 = real_particle.hrangesqd;
    for (int k = 0; k < P.ndim; k++) {
      const int Nghost = nc;
      if (P.mirror[k][0]) {
        const FLOAT x = 2*P.boxmin[k] - real_particle.r[k];
        const FLOAT dx = x - cellc.bbmin[k];
        if (dx*dx < h2 || x > cellc.hbmin[k]) {
          for (int n = 0; n < Nghost; n++) {
            ngbs.push_back(ngbs[n + old_size]);
            Part &g = ngbs.back();
            g.r[k] = 2*P.boxmin[k] - g.r[k]; g.v[k] *= -1; g.a[k] *= -1;      // reflect(), Particle.h:601-607
            nc++;
          }
        }
      }
      if (P.mirror[k][1]) {
        const FLOAT x = 2*P.boxmax[k] - real_particle.r[k];
        const FLOAT dx = x - cellc.bbmax[k];
        if (dx*dx < h2 || x < cellc.hbmax[k]) {
          for (int n = 0; n < Nghost; n++) {
            ngbs.push_back(ngbs[n + old_size]);
            Part &g = ngbs.back();
            g.r[k] = 2*P.boxmax[k] - g.r[k]; g.v[k] *= -1; g.a[k] *= -1;
            nc++;
          }
        }
      }
    }
  }
  void MakePeriodicGhost(Part &q, const FLOAT *centre) const {
    FLOAT dr[3];
    for (int k = 0; k < P.ndim; k++) dr[k] = q.r[k] - centre[k];
    if (NearestPeriodicVector(dr)) for (int k = 0; k < P.ndim; k++) q.r[k] = centre[k] + dr[k];
  }
  void ApplyPeriodicDistanceCorrection(FLOAT *r, FLOAT *dr) const {
    for (int k = 0; k < P.ndim; k++) if (P.periodic[k]) {
      if (dr[k] > P.boxhalf[k]) { dr[k] += -P.boxsize[k]; r[k] += -P.boxsize[k]; }
      else if (dr[k] < -P.boxhalf[k]) { dr[k] += P.boxsize[k]; r[k] += P.boxsize[k]; }
    }
  }
  bool PeriodicBoxOverlap(const FLOAT *b1min, const FLOAT *b1max, const FLOAT *b2min, const FLOAT *b2max) const {
    const int nd = P.ndim;                                          // GhostNeighbours.hpp:202-226
    if (!any_periodic()) return BoxOverlap(nd, b1min, b1max, b2min, b2max);
    FLOAT dr[3], corr[3] = {0, 0, 0}; bool any = false;
    for (int k = 0; k < nd; k++) dr[k] = 0.5*((b2max[k] + b2min[k]) - (b1max[k] + b1min[k]));
    for (int k = 0; k < nd; k++) if (P.periodic[k]) {
      if (dr[k] > P.boxhalf[k]) { corr[k] = -P.boxsize[k]; any = true; }
      else if (dr[k] < -P.boxhalf[k]) { corr[k] = P.boxsize[k]; any = true; }
      else corr[k] = 0;
    }
    if (any) {
      FLOAT pmin[3], pmax[3];
      for (int k = 0; k < nd; k++) { pmin[k] = b2min[k] + corr[k]; pmax[k] = b2max[k] + corr[k]; }
      return BoxOverlap(nd, b1min, b1max, pmin, pmax);
    }
    return BoxOverlap(nd, b1min, b1max, b2min, b2max);
  }

  // ---- one SPH pair of ComputeSphHydroForces (GradhSph.cpp:384-448) / ComputeSphHydroGravForces (:498-572)
  template <bool GRAV> void SphPair(Part &pi, const Part &nb, const FLOAT *r_nb) const {
    const int nd = P.ndim;
    const FLOAT invh_i = 1/pi.h, invrho_i = 1/pi.rho, invh_j = 1/nb.h, invrho_j = 1/nb.rho;
    FLOAT dr[3], dv[3], drmag, dvdr, wkerni, wkernj, paux;
    for (int k = 0; k < nd; k++) dr[k] = r_nb[k] - pi.r[k];
    if (GRAV) {
      for (int k = 0; k < nd; k++) dv[k] = nb.v[k] - pi.v[k];
      drmag = sqrt(Dot(dr, dr, nd) + small_number);
      const FLOAT invdrmag = 1.0/drmag;
      for (int k = 0; k < nd; k++) dr[k] *= invdrmag;
      dvdr = Dot(dv, dr, nd);
    }
    else {
      drmag = sqrt(Dot(dr, dr, nd));
      if (drmag > 0) for (int k = 0; k < nd; k++) dr[k] /= drmag;
      dvdr = Dot(nb.v, dr, nd);
      dvdr -= Dot(pi.v, dr, nd);
    }
    wkerni = pi.hfactor*kern.w1(drmag*invh_i);
    wkernj = nb.hfactor*kern.w1(drmag*invh_j);
    if (!GRAV) pi.div_v -= nb.m*dvdr*wkerni;
    paux = ((pi.pressure*pi.invomega)/(pi.rho*pi.rho))*wkerni + ((nb.pressure*nb.invomega)/(nb.rho*nb.rho))*wkernj;
    if (dvdr < 0.0) {
      const FLOAT winvrho = 0.25*(wkerni + wkernj)*(invrho_i + invrho_j);
      if (P.tdavisc) {                                                 // mon97mm97, GradhSph.cpp:419-424 / 533-538
        const FLOAT alpha_mean = (FLOAT) 0.5*(pi.alpha + nb.alpha);
        const FLOAT vsignal = pi.sound + nb.sound - P.beta_visc*alpha_mean*dvdr;
        paux -= alpha_mean*vsignal*dvdr*winvrho;
        pi.dudt -= 0.5*nb.m*alpha_mean*vsignal*dvdr*dvdr*winvrho;
      }
      else {
      const FLOAT vsignal = pi.sound + nb.sound - P.beta_visc*P.alpha_visc*dvdr;      // mon97
      paux -= P.alpha_visc*vsignal*dvdr*winvrho;
      pi.dudt -= 0.5*nb.m*P.alpha_visc*vsignal*dvdr*dvdr*winvrho;
      }
      // artificial conductivity, GradhSph.cpp:426-435 / 541-549
      if (P.acond == 1) pi.dudt += nb.m*dvdr*(nb.u - pi.u)*(invrho_i*wkerni + invrho_j*wkernj);
      else if (P.acond == 2)
        pi.dudt += (FLOAT) 0.5*nb.m*(pi.u - nb.u)*winvrho*(invrho_i + invrho_j)*sqrt(fabs(pi.pressure - nb.pressure));
    }
    for (int k = 0; k < nd; k++) pi.a[k] += nb.m*dr[k]*paux;
    if (GRAV) {
      const FLOAT invhsqdi = invh_i*invh_i;
      paux = 0.5*(invhsqdi*kern.wgrav(drmag*invh_i) + pi.zeta*wkerni + invh_j*invh_j*kern.wgrav(drmag*invh_j) + nb.zeta*wkernj);
      for (int k = 0; k < nd; k++) pi.atree[k] += nb.m*dr[k]*paux;
      pi.gpot += 0.5*nb.m*(invh_i*kern.wpot(drmag*invh_i) + invh_j*kern.wpot(drmag*invh_j));
      pi.div_v -= nb.m*dvdr*wkerni;
    }
  }

  // ---- GradhSphTree::UpdateAllSphHydroForces (GradhSphTree.cpp:280-435) and UpdateAllSphForces (:444-657)
  template <bool GRAV> void UpdateForces() {
    const std::vector<int> cl = ActiveLeafCells();
    const int nd = P.ndim;
#pragma omp parallel for schedule(guided) num_threads(P.nthreads)
    for (int cc = 0; cc < (int) cl.size(); cc++) {
      const Cell cellc = tree.cell[cl[cc]];
      int activelist[64]; Part activepart[64];
      const int Nactive = ActiveParticles(cellc, activelist);
      for (int j = 0; j < Nactive; j++) {
        activepart[j] = p[activelist[j]];
        activepart[j].div_v = 0.0; activepart[j].dudt = 0.0; activepart[j].dalphadt = 0.0;   // GradhSphTree.cpp:334-342
        activepart[j].levelneib = 0;
        activepart[j].gpot = GRAV ? (activepart[j].m/activepart[j].h)*kern.wpot(0.0) : 0.0;
        for (int k = 0; k < 3; k++) { activepart[j].a[k] = 0.0; if (GRAV) activepart[j].atree[k] = 0.0; }
      }
      // ---- walks: Tree::ComputeNeighbourAndGhostList Tree.cpp:562-617 /
      //             Tree::ComputeGravityInteractionAndGhostList Tree.cpp:628-735
      std::vector<int> tempperneib, tempdirectneib;
      struct MP { FLOAT r[3], m, q[5]; };
      std::vector<MP> gravcell;
      {
        const FLOAT hrangemax = kern.kernrange*cellc.hmax, rmax = cellc.rmax;
        int c = 0; FLOAT dr[3];
        while (c < tree.Ncell) {
          const Cell &o = tree.cell[c];
          bool near;
          FLOAT drsqd = 0.0;
          if (GRAV) {
            for (int k = 0; k < nd; k++) dr[k] = o.rcell[k] - cellc.rcell[k];
            NearestPeriodicVector(dr);
            drsqd = Dot(dr, dr, nd);
            near = drsqd <= pow(o.rmax + rmax + hrangemax, 2) || drsqd <= pow(rmax + o.rmax + kern.kernrange*o.hmax, 2);
          }
          else near = PeriodicBoxOverlap(cellc.bbmin, cellc.bbmax, o.hbmin, o.hbmax) ||
                      PeriodicBoxOverlap(cellc.hbmin, cellc.hbmax, o.bbmin, o.bbmax);
          if (near) {
            if (o.copen != -1) c = o.copen;
            else if (o.N == 0) c = o.cnext;
            else {
              int i = o.ifirst;
              while (i != -1) { tempperneib.push_back(i); if (i == o.ilast) break; i = tree.inext[i]; }
              c = o.cnext;
            }
          }
          else if (!GRAV) c = o.cnext;
          else if (o.N == 0) c = o.cnext;
          else if (!(drsqd < o.cdistsqd) &&                          // !open_cell_for_gravity, Tree.h:413-432
                   !(mac_now == 1 && drsqd*drsqd*cellc.amin*P.macerror < o.rmax*o.rmax*o.m) &&
                   !(mac_now == 2 && drsqd < o.mac*cellc.macfactor)) {
            if (o.copen == -1 && o.N == 1) tempdirectneib.push_back(o.ifirst);
            else { MP m; for (int k = 0; k < 3; k++) m.r[k] = o.r[k]; m.m = o.m; for (int k = 0; k < 5; k++) m.q[k] = o.q[k]; gravcell.push_back(m); }
            c = o.cnext;
          }
          else {
            if (o.copen != -1) c = o.copen;
            else {
              int i = o.ifirst;
              while (i != -1) { tempdirectneib.push_back(i); if (i == o.ilast) break; i = tree.inext[i]; }
              c = o.cnext;
            }
          }
        }
      }
      // ---- NeighbourManager::_EndSearch, NeighbourManager.h:368-474
      std::vector<Part> neibdata; std::vector<int> neiblist, directlist, neib_idx;
      {
        const FLOAT hrangemaxsqd = pow(cellc.rmax + kern.kernrange*cellc.hmax, 2), rmax = cellc.rmax;
        FLOAT dr[3];
        if (GRAV) for (size_t ii = 0; ii < tempdirectneib.size(); ii++) {
          ConstructGhostsScatterGather(p[tempdirectneib[ii]], cellc, neibdata);
          directlist.push_back((int) neibdata.size() - 1);
          neib_idx.resize(neibdata.size(), tempdirectneib[ii]);
        }
        size_t Nneib = directlist.size();
        for (size_t ii = 0; ii < tempperneib.size(); ii++) {
          ConstructGhostsScatterGather(p[tempperneib[ii]], cellc, neibdata);     // the particle (nearest image) + mirror copies
          while (Nneib < neibdata.size()) {                                      // NeighbourManager.h:423-441
            size_t Nmax = neibdata.size();
            for (int k = 0; k < nd; k++) dr[k] = neibdata[Nneib].r[k] - cellc.rcell[k];
            const FLOAT drsqd = Dot(dr, dr, nd);
            const FLOAT h2 = rmax + kern.kernrange*neibdata[Nneib].h;
            if (drsqd < hrangemaxsqd || drsqd < h2*h2) { neiblist.push_back((int) Nneib); Nneib++; }
            else if (GRAV) { directlist.push_back((int) Nneib); Nneib++; }
            else {
              Nmax--;
              if (Nmax > Nneib) neibdata[Nneib] = neibdata[Nmax];
              neibdata.resize(neibdata.size() - 1);
            }
          }
          neib_idx.resize(neibdata.size(), tempperneib[ii]);             // NeighbourManager.h:427,432: index of the real particle
        }
        for (size_t ii = 0; ii < neibdata.size(); ii++) neibdata[ii].levelneib = 0;       // HydroForcesParticle ctor, Particle.h:322
      }
      const size_t NCellDirectNeib = directlist.size();
      // ---- per particle: TrimNeighbourLists NeighbourManager.h:483-543, then the force operators
      for (int j = 0; j < Nactive; j++) {
        Part &pi = activepart[j];
        std::vector<int> culled;
        directlist.resize(NCellDirectNeib);
        FLOAT draux[3];
        for (size_t jj = 0; jj < neiblist.size(); jj++) {
          Part &nb = neibdata[neiblist[jj]];
          for (int k = 0; k < nd; k++) draux[k] = nb.r[k] - pi.r[k];
          if (any_periodic()) ApplyPeriodicDistanceCorrection(nb.r, draux);
          const FLOAT drsqd = Dot(draux, draux, nd);
          if (drsqd >= pi.hrangesqd && drsqd >= nb.hrangesqd) { if (GRAV) directlist.push_back(neiblist[jj]); }
          else culled.push_back(neiblist[jj]);
        }
        for (size_t jj = 0; jj < culled.size(); jj++) {
          Part &nb = neibdata[culled[jj]];
          SphPair<GRAV>(pi, nb, nb.r);
          pi.levelneib = std::max(pi.levelneib, nb.level);               // GradhSph.cpp:445-446 / 569-570
          nb.levelneib = std::max(nb.levelneib, pi.level);
        }
        const FLOAT invrho_i = 1/pi.rho;
        pi.div_v *= invrho_i;                                         // GradhSph.cpp:452-453 / 577-578
        pi.dudt -= pi.pressure*pi.div_v*invrho_i*pi.invomega;
        if (P.tdavisc == 1) {                                          // GradhSph.cpp:454-457 / 579-582 (note the sign quirk)
          const FLOAT invh_i = 1/pi.h;
          if (!GRAV) pi.dalphadt = (FLOAT) 0.1*pi.sound*(P.alpha_visc_min - pi.alpha)*invh_i + std::max(-pi.div_v, (FLOAT) 0.0)*(P.alpha_visc - pi.alpha);
          else pi.dalphadt = (FLOAT) 0.1*pi.sound*(P.alpha_visc_min - pi.alpha)*invh_i + std::max(pi.div_v, (FLOAT) 0.0)*(P.alpha_visc - pi.alpha);
        }
        if (GRAV) {
          // ComputeDirectGravForces, GradhSph.cpp:657-690
          for (size_t jj = 0; jj < directlist.size(); jj++) {
            const Part &g = neibdata[directlist[jj]];
            FLOAT dr[3];
            for (int k = 0; k < nd; k++) dr[k] = g.r[k] - pi.r[k];
            const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
            const FLOAT invdrmag = 1.0/sqrt(drsqd);
            const FLOAT invdr3 = invdrmag*invdrmag*invdrmag;
            for (int k = 0; k < nd; k++) pi.atree[k] += g.m*dr[k]*invdr3;
            pi.gpot += g.m*invdrmag;
          }
          // ComputeCellQuadrupoleForces / ComputeQuadropole, NeighbourSearch.h:384-475
          if (P.multipole == 1) for (size_t jj = 0; jj < gravcell.size(); jj++) {
            const MP &cl = gravcell[jj];
            FLOAT dr[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < nd; k++) dr[k] = pi.r[k] - cl.r[k];
            const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
            const FLOAT invdrsqd = (FLOAT) 1.0/drsqd;
            const FLOAT invdrmag = sqrt(invdrsqd);
            const FLOAT invdr5 = invdrsqd*invdrsqd*invdrmag;
            for (int k = 0; k < nd; k++) pi.atree[k] -= cl.m*dr[k]*invdrsqd*invdrmag;
            if (nd == 3) {
              const FLOAT qscalar = cl.q[0]*dr[0]*dr[0] + cl.q[2]*dr[1]*dr[1] - (cl.q[0] + cl.q[2])*dr[2]*dr[2] +
                                    2.0*(cl.q[1]*dr[0]*dr[1] + cl.q[3]*dr[0]*dr[2] + cl.q[4]*dr[1]*dr[2]);
              const FLOAT qfactor = 2.5*qscalar*invdr5*invdrsqd;
              pi.atree[0] += (cl.q[0]*dr[0] + cl.q[1]*dr[1] + cl.q[3]*dr[2])*invdr5 - qfactor*dr[0];
              pi.atree[1] += (cl.q[1]*dr[0] + cl.q[2]*dr[1] + cl.q[4]*dr[2])*invdr5 - qfactor*dr[1];
              pi.atree[2] += (cl.q[3]*dr[0] + cl.q[4]*dr[1] - (cl.q[0] + cl.q[2])*dr[2])*invdr5 - qfactor*dr[2];
              pi.gpot += cl.m*invdrmag + 0.5*qscalar*invdr5;
            }
            else if (nd == 2) {
              const FLOAT qscalar = cl.q[0]*dr[0]*dr[0] + cl.q[2]*dr[1]*dr[1] + 2.0*cl.q[1]*dr[0]*dr[1];
              const FLOAT qfactor = 2.5*qscalar*invdr5*invdrsqd;
              pi.atree[0] += (cl.q[0]*dr[0] + cl.q[1]*dr[1])*invdr5 - qfactor*dr[0];
              pi.atree[1] += (cl.q[1]*dr[0] + cl.q[2]*dr[1])*invdr5 - qfactor*dr[1];
              pi.gpot += cl.m*invdrmag + 0.5*qscalar*invdr5;
            }
            else {
              const FLOAT qscalar = cl.q[0]*dr[0]*dr[0];
              const FLOAT qfactor = 2.5*qscalar*invdr5*invdrsqd;
              pi.atree[0] += (cl.q[0]*dr[0])*invdr5 - qfactor*dr[0];
              pi.gpot += cl.m*invdrmag + 0.5*qscalar*invdr5;
            }
          }
          // ComputeCellMonopoleForces, NeighbourSearch.h:350-377 (the fast_* modes add their cell terms after the loop)
          else if (P.multipole == 0) for (size_t jj = 0; jj < gravcell.size(); jj++) {
            FLOAT dr[3];
            for (int k = 0; k < nd; k++) dr[k] = gravcell[jj].r[k] - pi.r[k];
            const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
            const FLOAT invdrsqd = 1.0/drsqd;
            const FLOAT invdrmag = sqrt(invdrsqd);
            const FLOAT invdr3 = invdrsqd*invdrmag;
            pi.gpot += gravcell[jj].m*invdrmag;
            for (int k = 0; k < nd; k++) pi.atree[k] += gravcell[jj].m*dr[k]*invdr3;
          }
        }
      }
      if (GRAV && (P.multipole == 2 || P.multipole == 3)) {
        // ComputeFastMonopoleForces, NeighbourSearch.h:768-794: field, gradient and potential of all cells at the leaf's
        // COM (AddMonopoleContribution :561-583), first-order Taylor expansion to the particles (:737-745)
        FLOAT rc[3] = {0, 0, 0}, ac[3] = {0, 0, 0}, dphi[3] = {0, 0, 0}, q[6] = {0, 0, 0, 0, 0, 0}, pot = 0;
        for (int k = 0; k < nd; k++) rc[k] = cellc.r[k];
        for (size_t cc2 = 0; cc2 < gravcell.size(); cc2++) {
          FLOAT dr[3] = {0, 0, 0};
          for (int k = 0; k < nd; k++) dr[k] = gravcell[cc2].r[k] - rc[k];
          const FLOAT invdrmag = sqrt((FLOAT) 1.0/Dot(dr, dr, nd));
          const FLOAT invdrsqd = invdrmag*invdrmag;
          const FLOAT invdr3 = invdrsqd*invdrmag;
          FLOAT mc = gravcell[cc2].m;
          pot += mc*invdrmag;
          mc *= invdr3;
          for (int k = 0; k < nd; k++) ac[k] += mc*dr[k];
          for (int k = 0; k < nd; k++) dphi[k] += mc*dr[k];
          q[0] += mc*(3.0*dr[0]*dr[0]*invdrsqd - 1);
          if (nd > 1) { q[1] += mc*(3.0*dr[0]*dr[1]*invdrsqd); q[2] += mc*(3.0*dr[1]*dr[1]*invdrsqd - 1); }
          if (nd > 2) { q[3] += mc*(3.0*dr[2]*dr[0]*invdrsqd); q[4] += mc*(3.0*dr[2]*dr[1]*invdrsqd); q[5] += mc*(3.0*dr[2]*dr[2]*invdrsqd - 1); }
          if (P.multipole == 3) {
            // fast_quadrupole: FastMultipoleForces::AddQuadrupoleContribution, NeighbourSearch.h:601-720 (dr = rc - cell.r)
            const FLOAT *cq = gravcell[cc2].q;
            FLOAT e[3] = {0, 0, 0};
            for (int k = 0; k < nd; k++) e[k] = rc[k] - gravcell[cc2].r[k];
            const FLOAT drsqd = Dot(e, e, nd) + small_number;
            const FLOAT i2 = (FLOAT) 1.0/drsqd;
            const FLOAT im = sqrt(i2);
            const FLOAT i5 = i2*i2*im;
            FLOAT qscalar, qx[3] = {0, 0, 0};
            if (nd == 3) {
              qscalar = cq[0]*e[0]*e[0] + cq[2]*e[1]*e[1] - (cq[0] + cq[2])*e[2]*e[2] + 2.0*(cq[1]*e[0]*e[1] + cq[3]*e[0]*e[2] + cq[4]*e[1]*e[2]);
              qx[0] = (cq[0]*e[0] + cq[1]*e[1] + cq[3]*e[2])*i5;
              qx[1] = (cq[1]*e[0] + cq[2]*e[1] + cq[4]*e[2])*i5;
              qx[2] = (cq[3]*e[0] + cq[4]*e[1] - (cq[0] + cq[2])*e[2])*i5;
            }
            else if (nd == 2) {
              qscalar = cq[0]*e[0]*e[0] + cq[2]*e[1]*e[1] + 2.0*cq[1]*e[0]*e[1];
              qx[0] = (cq[0]*e[0] + cq[1]*e[1])*i5;
              qx[1] = (cq[1]*e[0] + cq[2]*e[1])*i5;
            }
            else { qscalar = cq[0]*e[0]*e[0]; qx[0] = (cq[0]*e[0])*i5; }
            const FLOAT qfactor = 2.5*qscalar*i5*i2;
            pot += 0.5*qscalar*i5;
            for (int k = 0; k < nd; k++) ac[k] += qx[k] - qfactor*e[k];
            for (int k = 0; k < nd; k++) dphi[k] += qx[k] - qfactor*e[k];
            for (int k = 0; k < nd; k++) qx[k] *= 5*i2;
            q[0] += qfactor*(7*e[0]*e[0]*i2 - 1);
            if (nd > 1) { q[1] += qfactor*(7*e[0]*e[1]*i2); q[2] += qfactor*(7*e[1]*e[1]*i2 - 1); }
            if (nd > 2) { q[3] += qfactor*(7*e[0]*e[2]*i2); q[4] += qfactor*(7*e[1]*e[2]*i2); q[5] += qfactor*(7*e[2]*e[2]*i2 - 1); }
            q[0] -= qx[0]*e[0] + qx[0]*e[0] - cq[0]*i5;
            if (nd > 1) { q[1] -= qx[0]*e[1] + qx[1]*e[0] - cq[1]*i5; q[2] -= qx[1]*e[1] + qx[1]*e[1] - cq[2]*i5; }
            if (nd > 2) {
              q[3] -= qx[0]*e[2] + qx[2]*e[0] - cq[3]*i5;
              q[4] -= qx[1]*e[2] + qx[2]*e[1] - cq[4]*i5;
              q[5] -= qx[2]*e[2] + qx[2]*e[2] + (cq[0] + cq[2])*i5;
            }
          }
        }
        for (int j = 0; j < Nactive; j++) {
          Part &pi = activepart[j];
          FLOAT dr[3] = {0, 0, 0};
          for (int k = 0; k < nd; k++) dr[k] = pi.r[k] - rc[k];
          if (nd == 3) {
            pi.atree[0] += ac[0] + q[0]*dr[0] + q[1]*dr[1] + q[3]*dr[2];
            pi.atree[1] += ac[1] + q[1]*dr[0] + q[2]*dr[1] + q[4]*dr[2];
            pi.atree[2] += ac[2] + q[3]*dr[0] + q[4]*dr[1] + q[5]*dr[2];
            pi.gpot += pot + dphi[0]*dr[0] + dphi[1]*dr[1] + dphi[2]*dr[2];
          }
          else if (nd == 2) {
            pi.atree[0] += ac[0] + q[0]*dr[0] + q[1]*dr[1];
            pi.atree[1] += ac[1] + q[1]*dr[0] + q[2]*dr[1];
            pi.gpot += pot + dphi[0]*dr[0] + dphi[1]*dr[1];
          }
          else { pi.atree[0] += ac[0] + q[0]*dr[0]; pi.gpot += pot + dphi[0]*dr[0]; }
        }
      }
      if (GRAV) for (int j = 0; j < Nactive; j++) activepart[j].gpot_hydro = activepart[j].gpot;   // GradhSphTree.cpp:595-598
      // gas <- stars: GradhSph::ComputeStarGravForces, GradhSph.cpp:699-743 (mean-h kernel softening), GradhSphTree.cpp:600-607
      if (GRAV && !stars.empty()) for (int j = 0; j < Nactive; j++) {
        Part &pi = activepart[j];
        for (size_t s = 0; s < stars.size(); s++) {
          FLOAT dr[3];
          for (int k = 0; k < nd; k++) dr[k] = stars[s].r[k] - pi.r[k];
          const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
          const FLOAT drmag = sqrt(drsqd);
          const FLOAT invdrmag = (FLOAT) 1.0/drmag;
          const FLOAT invhmean = (FLOAT) 2.0/(pi.h + stars[s].h);
          const FLOAT paux = stars[s].m*invhmean*invhmean*kern.wgrav(drmag*invhmean)*invdrmag;
          for (int k = 0; k < nd; k++) pi.atree[k] += paux*dr[k];
          pi.gpot += stars[s].m*invhmean*kern.wpot(drmag*invhmean);
        }
      }
      for (int j = 0; j < Nactive; j++) {                             // GradhSphTree.cpp:396-404 / 610-619
        const int i = activelist[j];
        for (int k = 0; k < nd; k++) p[i].a[k] += activepart[j].a[k];
        if (GRAV) {
          for (int k = 0; k < nd; k++) p[i].a[k] += activepart[j].atree[k];
          for (int k = 0; k < nd; k++) p[i].atree[k] += activepart[j].atree[k];
          p[i].gpot_hydro += activepart[j].gpot_hydro;
        }
        p[i].gpot += activepart[j].gpot;
        p[i].dudt += activepart[j].dudt;
        p[i].div_v += activepart[j].div_v;
        if (!GRAV) p[i].dalphadt += activepart[j].dalphadt;          // GradhSphTree.cpp:403 (hydro driver only; never zeroed)
      }
      if (P.Nlevels > 1) {
        // levelneib of the active particles and of every neighbour copy back to the real particle it was made from
        // (GradhSphTree.cpp:375-382, 405, 412-417 / 619-640; integer max, so the per-thread buffers of the reference
        // reduce to one max per particle)
#pragma omp critical(levelneib)
        {
          for (int j = 0; j < Nactive; j++) p[activelist[j]].levelneib = std::max(p[activelist[j]].levelneib, activepart[j].levelneib);
          for (size_t ii = 0; ii < neibdata.size(); ii++) p[neib_idx[ii]].levelneib = std::max(p[neib_idx[ii]].levelneib, neibdata[ii].levelneib);
        }
      }
    }
  }
  void Forces() { if (P.self_gravity) UpdateForces<true>(); else UpdateForces<false>(); }

  // ---- stars <- gas: HydroTree::UpdateAllStarGasForces (HydroTree.cpp:552-657), Tree::ComputeStarGravityInteractionList
  //      (Tree.cpp:748-885; always the geometric opening criterion), NbodyLeapfrogKDK::CalculateDirectHydroForces
  //      (NbodyLeapfrogKDK.cpp:151-239), ComputeCellMonopoleForces / ComputeCellQuadrupoleForces (NeighbourSearch.h:350-475)
  void UpdateAllStarGasForces() {
    const int nd = P.ndim;
    for (size_t si = 0; si < stars.size(); si++) {
      GasStar &st = stars[si];
      std::vector<int> neiblist, directlist, cells;
      const FLOAT hrangemax = kern.kernrange*st.h;
      int cc = 0;
      while (cc < tree.Ncell) {
        const Cell &o = tree.cell[cc];
        FLOAT dr[3];
        for (int k = 0; k < nd; k++) dr[k] = o.rcell[k] - st.r[k];
        const FLOAT drsqd = Dot(dr, dr, nd);
        if (drsqd < pow((FLOAT) 0.5*hrangemax + o.rmax + (FLOAT) 0.5*kern.kernrange*o.hmax, 2)) {
          if (o.copen != -1) cc = o.copen;
          else {
            int i = o.ifirst;
            while (i != -1) { neiblist.push_back(i); if (i == o.ilast) break; i = tree.inext[i]; }
            cc = o.cnext;
          }
        }
        else if (drsqd > o.cdistsqd && o.N > 0) {
          if (o.copen == -1 && o.N == 1) directlist.push_back(o.ifirst);
          else cells.push_back(cc);
          cc = o.cnext;
        }
        else if (drsqd <= o.cdistsqd && o.N > 0) {
          if (o.copen != -1) cc = o.copen;
          else {
            int i = o.ifirst;
            while (i != -1) { directlist.push_back(i); if (i == o.ilast) break; i = tree.inext[i]; }
            cc = o.cnext;
          }
        }
        else cc = o.cnext;
      }
      for (size_t jj = 0; jj < neiblist.size(); jj++) {
        const Part &g = p[neiblist[jj]];
        FLOAT dr[3];
        for (int k = 0; k < nd; k++) dr[k] = g.r[k] - st.r[k];
        const FLOAT drsqd = Dot(dr, dr, nd);
        const FLOAT drmag = sqrt(drsqd);
        const FLOAT invdrmag = 1.0/drmag;
        const FLOAT invhmean = 2.0/(st.h + g.h);
        const FLOAT paux = g.m*invhmean*invhmean*kern.wgrav(drmag*invhmean)*invdrmag;
        for (int k = 0; k < nd; k++) st.a[k] += paux*dr[k];
        st.gpot += g.m*invhmean*kern.wpot(drmag*invhmean);
      }
      for (size_t jj = 0; jj < directlist.size(); jj++) {
        const Part &g = p[directlist[jj]];
        FLOAT dr[3];
        for (int k = 0; k < nd; k++) dr[k] = g.r[k] - st.r[k];
        const FLOAT drsqd = Dot(dr, dr, nd);
        const FLOAT drmag = sqrt(drsqd);
        const FLOAT invdrmag = 1.0/drmag;
        const FLOAT paux = g.m*pow(invdrmag, 3);
        for (int k = 0; k < nd; k++) st.a[k] += paux*dr[k];
        st.gpot += g.m*invdrmag;
      }
      const bool quad = P.multipole == 1 || P.multipole == 3;
      for (size_t jj = 0; jj < cells.size(); jj++) {
        const Cell &cl = tree.cell[cells[jj]];
        if (!quad) {                                                   // NeighbourSearch.h:350-377
          FLOAT dr[3];
          for (int k = 0; k < nd; k++) dr[k] = cl.r[k] - st.r[k];
          const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
          const FLOAT invdrsqd = 1.0/drsqd;
          const FLOAT invdrmag = sqrt(invdrsqd);
          const FLOAT invdr3 = invdrsqd*invdrmag;
          st.gpot += cl.m*invdrmag;
          for (int k = 0; k < nd; k++) st.a[k] += cl.m*dr[k]*invdr3;
        }
        else {                                                         // NeighbourSearch.h:384-475 (3-D)
          FLOAT dr[3] = {0.0, 0.0, 0.0};
          for (int k = 0; k < nd; k++) dr[k] = st.r[k] - cl.r[k];
          const FLOAT drsqd = Dot(dr, dr, nd) + small_number;
          const FLOAT invdrsqd = (FLOAT) 1.0/drsqd;
          const FLOAT invdrmag = sqrt(invdrsqd);
          const FLOAT invdr5 = invdrsqd*invdrsqd*invdrmag;
          for (int k = 0; k < nd; k++) st.a[k] -= cl.m*dr[k]*invdrsqd*invdrmag;
          const FLOAT qscalar = cl.q[0]*dr[0]*dr[0] + cl.q[2]*dr[1]*dr[1] - (cl.q[0] + cl.q[2])*dr[2]*dr[2] +
                                2.0*(cl.q[1]*dr[0]*dr[1] + cl.q[3]*dr[0]*dr[2] + cl.q[4]*dr[1]*dr[2]);
          const FLOAT qfactor = 2.5*qscalar*invdr5*invdrsqd;
          st.a[0] += (cl.q[0]*dr[0] + cl.q[1]*dr[1] + cl.q[3]*dr[2])*invdr5 - qfactor*dr[0];
          st.a[1] += (cl.q[1]*dr[0] + cl.q[2]*dr[1] + cl.q[4]*dr[2])*invdr5 - qfactor*dr[1];
          st.a[2] += (cl.q[3]*dr[0] + cl.q[4]*dr[1] - (cl.q[0] + cl.q[2])*dr[2])*invdr5 - qfactor*dr[2];
          st.gpot += cl.m*invdrmag + 0.5*qscalar*invdr5;
        }
      }
    }
  }

  // ---- time integration: SphLeapfrogKDK.cpp:76-127, Integration.cpp (CheckBoundaries), SphIntegration.cpp:81-134,
  //      Simulation.cpp:1669-1754, SphLeapfrogKDK.cpp:219-272
  void AdvanceParticles() {
    for (int i = 0; i < Nhydro; i++) {
      Part &q = p[i];
      if (q.flags & F_DEAD) continue;                                  // SphLeapfrogKDK.cpp:99
      const FLOAT dt = t - q.tlast;
      for (int k = 0; k < P.ndim; k++) q.r[k] = q.r0[k] + q.v0[k]*dt + 0.5*q.a0[k]*dt*dt;
      for (int k = 0; k < P.ndim; k++) q.v[k] = q.v0[k] + q.a0[k]*dt;
      if (P.tdavisc) q.alpha += q.dalphadt*timestep;                 // SphLeapfrogKDK.cpp:111
      if (P.energy_integration) q.u = q.u0 + q.dudt0*dt;
      if (P.Nlevels == 1 || n - q.nlast == q.nstep) q.flags |= F_ACTIVE; else q.flags &= ~F_ACTIVE;   // SphLeapfrogKDK.cpp:117-118
    }
    for (int i = 0; i < Nhydro; i++) for (int k = 0; k < P.ndim; k++) {       // TimeIntegration::CheckBoundaries
      Part &q = p[i];
      if (q.r[k] < P.boxmin[k]) {
        if (P.periodic[k]) { q.r[k] += P.boxsize[k]; q.r0[k] += P.boxsize[k]; }
        if (P.mirror[k][0]) {
          q.r[k] = (FLOAT) 2.0*P.boxmin[k] - q.r[k]; q.r0[k] = (FLOAT) 2.0*P.boxmin[k] - q.r0[k];
          q.v[k] = -q.v[k]; q.v0[k] = -q.v0[k]; q.a[k] = -q.a[k]; q.a0[k] = -q.a0[k];
        }
      }
      if (q.r[k] > P.boxmax[k]) {
        if (P.periodic[k]) { q.r[k] -= P.boxsize[k]; q.r0[k] -= P.boxsize[k]; }
        if (P.mirror[k][1]) {
          q.r[k] = (FLOAT) 2.0*P.boxmax[k] - q.r[k]; q.r0[k] = (FLOAT) 2.0*P.boxmax[k] - q.r0[k];
          q.v[k] = -q.v[k]; q.v0[k] = -q.v0[k]; q.a[k] = -q.a[k]; q.a0[k] = -q.a0[k];
        }
      }
    }
  }
  double Timestep(const Part &q) const {
    double ts = P.courant_mult*q.h/(q.sound + q.h*fabs(q.div_v) + small_number_dp);
    const double amag = sqrt(Dot(q.a, q.a, P.ndim));
    ts = std::min(ts, P.accel_mult*sqrt(q.h/(amag + small_number_dp)));
    if (P.energy_integration) ts = std::min(ts, P.energy_mult*(double) (q.u/(fabs(q.dudt) + small_number)));
    return ts;
  }
  void ComputeGlobalTimestep() {
    double dt_min = big_number_dp;
    for (int i = 0; i < Nhydro; i++) {
      p[i].level = 0; p[i].levelneib = 0; p[i].nstep = 1;             // Simulation.cpp:1700-1704 (level_step = 0)
      p[i].dt_next = Timestep(p[i]); dt_min = std::min(dt_min, p[i].dt_next);
    }
    timestep = dt_min; n = 0;
    for (int i = 0; i < Nhydro; i++) p[i].dt_next = timestep;
  }
  // InlineFuncs.h:550-558
  static int ComputeTimestepLevel(double dt, double dt_max) { return std::max((int) (invlogetwo*log(dt_max/dt)) + 1, 0); }
  static int ipow2(int e) { return (int) pow(2.0, e); }
  // Simulation::ComputeBlockTimesteps, Simulation.cpp:1764-2200 (hydro particles only: no stars, no sinks)
  void ComputeBlockTimesteps() {
    if (n == nresync) {                                                // resynchronise: rebuild the level structure
      n = 0; timestep = big_number_dp;
      double dt_min_hydro = big_number_dp;
      for (int i = 0; i < Nhydro; i++) {
        const double dt = Timestep(p[i]);
        timestep = std::min(timestep, dt); dt_min_hydro = std::min(dt_min_hydro, dt);
        p[i].dt_next = dt;
      }
      level_max = P.Nlevels - 1;
      level_step = level_max + integration_step - 1;
      dt_max = timestep*pow(2.0, level_max);
      const int level_max_hydro = std::min(ComputeTimestepLevel(dt_min_hydro, dt_max), level_max);
      for (int i = 0; i < Nhydro; i++) {
        Part &q = p[i];
        const int level = P.sph_single_timestep ? level_max_hydro : std::min(ComputeTimestepLevel(q.dt_next, dt_max), level_max);
        q.level = level; q.levelneib = level;
        q.nstep = ipow2(level_step - q.level);
        q.nlast = n;
        q.dt_next = q.nstep*timestep;                                  // NB: 'timestep' is still the minimum dt here (:1913)
        q.flags |= F_END;
      }
      nresync = ipow2(level_step);
      timestep = dt_max/(double) nresync;
    }
    else {
      const int level_max_old = level_max;
      level_max = 0;
      int level_max_hydro = 0;
      for (int i = 0; i < Nhydro; i++) {
        Part &q = p[i];
        if (n - q.nlast == q.nstep && q.nstep != ipow2(level_step - q.level)) {   // step cut short by CheckTimesteps (:1956-1966)
          const double dt = Timestep(q);
          const int level = std::max(ComputeTimestepLevel(dt, dt_max), q.levelneib - P.level_diff_max);
          q.level = std::max(q.level, level);
          q.levelneib = q.level;
          q.nlast = n;
          q.nstep = ipow2(level_step - q.level);
          q.dt_next = q.nstep*timestep;
          q.flags |= F_END;
        }
        else if (n - q.nlast == q.nstep) {                             // natural end of step (:1968-1992)
          const int nstep = q.nstep, last_level = q.level;
          const double dt = Timestep(q);
          const int level = std::max(ComputeTimestepLevel(dt, dt_max), q.levelneib - P.level_diff_max);
          if (level < last_level && last_level > 1 && n%(2*nstep) == 0) q.level = last_level - 1;
          else if (level > last_level) q.level = level;
          else q.level = last_level;
          q.levelneib = level;
          q.nlast = n;
          q.nstep = ipow2(level_step - q.level);
          q.dt_next = q.nstep*timestep;
          q.flags |= F_END;
        }
        level_max_hydro = std::max(level_max_hydro, q.level);
        level_max = std::max(level_max, q.level);
      }
      if (P.sph_single_timestep) for (int i = 0; i < Nhydro; i++) if (p[i].nlast == n) p[i].level = level_max_hydro;
      const int istep = ipow2(level_step - level_max_old + 1);
      if (level_max > level_max_old) {                                 // levels added: refine the integer clock (:2101-2112)
        const int nfactor = ipow2(level_max - level_max_old);
        n *= nfactor;
        for (int i = 0; i < Nhydro; i++) { p[i].nstep *= nfactor; p[i].nlast *= nfactor; }
      }
      else if (level_max <= level_max_old - 1 && level_max_old > 1 && n%istep == 0) {   // one level removed (:2113-2127)
        level_max = level_max_old - 1;
        const int nfactor = ipow2(level_max_old - level_max);
        n /= nfactor;
        for (int i = 0; i < Nhydro; i++) { p[i].nlast /= nfactor; p[i].nstep /= nfactor; }
      }
      else level_max = level_max_old;
      level_step = level_max + integration_step - 1;
      nresync = ipow2(level_step);
      timestep = dt_max/(double) nresync;
      for (int i = 0; i < Nhydro; i++) if (p[i].nlast == n) p[i].nstep = ipow2(level_step - p[i].level);
    }
  }
  // SphLeapfrogKDK::CheckTimesteps, SphLeapfrogKDK.cpp:284-330
  int CheckTimesteps() {
    int activecount = 0;
    for (int i = 0; i < Nhydro; i++) {
      Part &q = p[i];
      if (q.flags & F_DEAD) continue;                                  // SphLeapfrogKDK.cpp:307
      const int dn = n - q.nlast;
      if (dn == q.nstep) continue;
      if (q.levelneib - q.level > P.level_diff_max) {
        const int level_new = q.levelneib - P.level_diff_max;
        const int nnewstep = ipow2(level_step - level_new);
        if (dn%nnewstep == 0) {
          if (dn > 0) q.nstep = dn;
          q.level = level_new;
          q.flags |= F_ACTIVE;
          activecount++;
        }
      }
    }
    return activecount;
  }
  void UpdateActiveParticleCounters() {                              // KDTree.cpp:1217-1254 (leaf cells only)
    for (int c = 0; c < tree.Ncell; c++) {
      Cell &x = tree.cell[c];
      x.Nactive = 0;
      if (x.level != tree.ltot) continue;
      int i = x.ifirst;
      while (i != -1) {
        if (i < Nhydro && (p[i].flags & F_ACTIVE) && !(p[i].flags & F_DEAD)) x.Nactive++;
        if (i == x.ilast) break;
        i = tree.inext[i];
      }
    }
  }
  void EndTimestepBlock() {                                          // SphLeapfrogKDK.cpp:219-272
    for (int i = 0; i < Nhydro; i++) {
      Part &q = p[i];
      if (q.flags & F_DEAD) continue;
      if (!(q.flags & F_END)) continue;
      for (int k = 0; k < P.ndim; k++) q.v[k] += 0.5*q.dt*(q.a[k] - q.a0[k]);
      for (int k = 0; k < P.ndim; k++) { q.r0[k] = q.r[k]; q.v0[k] = q.v[k]; q.a0[k] = q.a[k]; }
      if (P.energy_integration) {
        q.u += 0.5*(q.dudt - q.dudt0)*q.dt;
        if (q.u <= 0.0) q.u = q.u0 + q.dudt0*q.dt;
        q.u0 = q.u; q.dudt0 = q.dudt;
      }
      q.nlast = n; q.tlast = t; q.dt = q.dt_next; q.dt_next = 0; q.flags &= ~(F_ACTIVE | F_END);
    }
  }
  void EndTimestep() {
    if (P.Nlevels > 1) { EndTimestepBlock(); return; }
    for (int i = 0; i < Nhydro; i++) {
      Part &q = p[i];
      if (q.flags & F_DEAD) { q.flags |= F_END; continue; }            // SphLeapfrogKDK.cpp:238 (end_timestep was set by ComputeGlobalTimestep)
      for (int k = 0; k < P.ndim; k++) q.v[k] += 0.5*q.dt*(q.a[k] - q.a0[k]);
      for (int k = 0; k < P.ndim; k++) { q.r0[k] = q.r[k]; q.v0[k] = q.v[k]; q.a0[k] = q.a[k]; }
      if (P.energy_integration) {
        q.u += 0.5*(q.dudt - q.dudt0)*q.dt;
        if (q.u <= 0.0) q.u = q.u0 + q.dudt0*q.dt;
        q.u0 = q.u; q.dudt0 = q.dudt;
      }
      q.tlast = t; q.dt = q.dt_next; q.dt_next = 0; q.flags &= ~F_ACTIVE;
    }
  }
  void DensityPass() { SearchBoundaryGhostParticles(); BuildGhostTree(); UpdateAllSphProperties(); }
  void Setup(int h_provided) {                                       // SphSimulation.cpp:204-565 (see gh_setup)
    SetupPasses(h_provided);
    t = 0.0; n = 0; nresync = 0;
    if (P.Nlevels > 1) ComputeBlockTimesteps(); else ComputeGlobalTimestep();
    EndTimestep();
  }
  void SetupPasses(int h_provided) {                                 // the density / force passes of the setup (:266-473)
    for (int i = 0; i < Nhydro; i++) p[i].flags |= F_ACTIVE;
    const int npass = h_provided ? 2 : 3;
    for (int q = 0; q < npass; q++) { BuildTree(); DensityPass(); }
    // relative MAC: first force pass geometric, tree rebuilt (stocks amin), second pass with the MAC (SphSimulation.cpp:381-473)
    const bool relmac = P.self_gravity && P.gravity_mac != 0;
    mac_now = 0;
    ZeroAccelerations(); Forces();
    mac_now = P.gravity_mac;
    if (relmac) { BuildTree(); ZeroAccelerations(); Forces(); }
  }
  void MainLoop() {                                                  // SphSimulation.cpp:574-880
    n++; Nsteps++; t = t + timestep;
    AdvanceParticles();
    if (P.Nlevels > 1) {
      StepTree(); SearchBoundaryGhostParticles(); BuildGhostTree();
      int activecount = 0;
      do {                                                             // SphSimulation.cpp:654-755
        if (activecount > 0) UpdateActiveParticleCounters();
        UpdateAllSphProperties();
        ZeroAccelerations();
        // SphSimulation.cpp:665-679 (nradstep = 1): thermal properties of ALL particles from the predicted u - a no-op with
        // a global timestep, but with block timesteps this is what refreshes pressure / sound of the inactive neighbours
        for (int i = 0; i < Nhydro; i++) {
          Thermal(p[i]);
        }
        Forces();
        for (int i = 0; i < Nhydro; i++) p[i].flags &= ~F_ACTIVE;
        activecount = CheckTimesteps();
      } while (activecount > 0);
      ComputeBlockTimesteps(); EndTimestep();
      rebuild_tree = false;
      return;
    }
    StepTree(); DensityPass();
    ZeroAccelerations(); Forces();
    ComputeGlobalTimestep(); EndTimestep();
    rebuild_tree = false;
  }
};


// ---------------------------------------------------------------------------------------------
// N-body direct sum + leapfrog KDK for stars (global timestep, all stars active)
// ---------------------------------------------------------------------------------------------
struct Star {
  FLOAT r[3], v[3], a[3], adot[3], r0[3], v0[3], a0[3], adot0[3];
  FLOAT m, h, gpot, dt, dt_next, tlast, dt_internal;
  FLOAT invh;                                     // 1/h of the gas particle a sink was made from (Sinks.cpp:317)
  int nstep, nlast, level; bool active, end_timestep;
};

struct NbodyOracle {
  int N, softening;
  FLOAT nbody_mult, t, timestep;
  M4 kern;
  std::vector<Star> s;
  NbodyOracle(int N_, int soft, FLOAT mult) : N(N_), softening(soft), nbody_mult(mult), t(0.0), timestep(0.0), kern(3), s(N_) {}

  // Nbody::CalculateDirectGravForces, Nbody.cpp:233-287 (open boundaries, no Ewald)
  void DirectGrav() {
    const int ndim = 3;
    for (int i = 0; i < N; i++) {
      if (!s[i].active) continue;
      for (int j = 0; j < N; j++) {
        if (i == j) continue;
        FLOAT dr[3], dv[3];
        for (int k = 0; k < ndim; k++) dr[k] = s[j].r[k] - s[i].r[k];
        for (int k = 0; k < ndim; k++) dv[k] = s[j].v[k] - s[i].v[k];
        const FLOAT drsqd = Dot(dr, dr, ndim);
        const FLOAT invdrmag = (FLOAT) 1.0/sqrt(drsqd);
        const FLOAT drdt = Dot(dv, dr, ndim)*invdrmag;
        s[i].gpot += s[j].m*invdrmag;
        for (int k = 0; k < ndim; k++) s[i].a[k] += s[j].m*dr[k]*pow(invdrmag, 3);
        for (int k = 0; k < ndim; k++) s[i].adot[k] += s[j].m*pow(invdrmag, 3)*(dv[k] - 3.0*drdt*invdrmag*dr[k]);
      }
    }
  }
  // NbodyLeapfrogKDK::CalculateDirectSmoothedGravForces, NbodyLeapfrogKDK.cpp:78-142
  void DirectSmoothedGrav() {
    const int ndim = 3;
    for (int i = 0; i < N; i++) {
      if (!s[i].active) continue;
      for (int j = 0; j < N; j++) {
        if (i == j) continue;
        FLOAT dr[3], dv[3];
        for (int k = 0; k < ndim; k++) dr[k] = s[j].r[k] - s[i].r[k];
        for (int k = 0; k < ndim; k++) dv[k] = s[j].v[k] - s[i].v[k];
        const FLOAT drsqd = Dot(dr, dr, ndim);
        const FLOAT drmag = sqrt(drsqd) + small_number;
        const FLOAT invdrmag = (FLOAT) 1.0/drmag;
        const FLOAT invhmean = (FLOAT) 2.0/(s[i].h + s[j].h);
        const FLOAT drdt = Dot(dv, dr, ndim)*invdrmag;
        const FLOAT paux = s[j].m*invhmean*invhmean*kern.wgrav(drmag*invhmean)*invdrmag;
        const FLOAT wmean = kern.w0(drmag*invhmean)*powf(invhmean, ndim);       // float pow, :118
        s[i].gpot += s[j].m*invhmean*kern.wpot(drmag*invhmean);
        for (int k = 0; k < ndim; k++) s[i].a[k] += paux*dr[k];
        for (int k = 0; k < ndim; k++) s[i].adot[k] += paux*dv[k] - (FLOAT) 3.0*paux*drdt*invdrmag*dr[k] +
          (FLOAT) 2.0*twopi*s[j].m*drdt*wmean*invdrmag*dr[k];
      }
    }
  }
  void Zero() {                                  // NbodySimulation.cpp:332-340
    for (int i = 0; i < N; i++) if (s[i].active) { for (int k = 0; k < 3; k++) { s[i].a[k] = 0.0; s[i].adot[k] = 0.0; } s[i].gpot = 0.0; }
  }
  void Forces() { if (softening) DirectSmoothedGrav(); else DirectGrav(); }
  // NbodyLeapfrogKDK::AdvanceParticles, :253-291
  void Advance(int n) {
    for (int i = 0; i < N; i++) {
      const int dn = n - s[i].nlast;
      const FLOAT dt = t - s[i].tlast;
      for (int k = 0; k < 3; k++) s[i].r[k] = s[i].r0[k] + s[i].v0[k]*dt + (FLOAT) 0.5*s[i].a0[k]*dt*dt;
      for (int k = 0; k < 3; k++) s[i].v[k] = s[i].v0[k] + s[i].a0[k]*dt;
      s[i].active = (dn == s[i].nstep);
    }
  }
  // CorrectionTerms, :301-331
  void Correct(int n) {
    for (int i = 0; i < N; i++) {
      const int dn = n - s[i].nlast;
      if (dn == s[i].nstep) for (int k = 0; k < 3; k++) s[i].v[k] += (FLOAT) 0.5*(s[i].a[k] - s[i].a0[k])*(t - s[i].tlast);
    }
  }
  // NbodyLeapfrogKDK::Timestep :387-400 inside Simulation::ComputeGlobalTimestep (Simulation.cpp:1720-1745)
  void GlobalTimestep() {
    double dt_min = big_number_dp;
    for (int i = 0; i < N; i++) {
      s[i].end_timestep = true;
      s[i].nstep = 1;
      const double amag = sqrt(Dot(s[i].a, s[i].a, 3));
      double ts = nbody_mult*sqrt(s[i].h/(amag + small_number_dp));
      ts = std::min(ts, (double) s[i].dt_internal);
      s[i].dt_next = ts;
      dt_min = std::min(dt_min, ts);
    }
    timestep = dt_min;
    for (int i = 0; i < N; i++) s[i].dt_next = timestep;
  }
  // EndTimestep :341-377
  void EndTimestep(int n) {
    for (int i = 0; i < N; i++) if (s[i].end_timestep) {
      for (int k = 0; k < 3; k++) { s[i].r0[k] = s[i].r[k]; s[i].v0[k] = s[i].v[k]; s[i].a0[k] = s[i].a[k]; s[i].adot0[k] = s[i].adot[k]; }
      s[i].nlast = n; s[i].tlast = t; s[i].dt = s[i].dt_next; s[i].dt_next = 0;
      s[i].active = false; s[i].end_timestep = false;
    }
  }
  void Setup() {
    for (int i = 0; i < N; i++) {
      s[i].active = true;
      for (int k = 0; k < 3; k++) { s[i].r0[k] = s[i].r[k]; s[i].v0[k] = s[i].v[k]; }
      s[i].nlast = 0; s[i].tlast = 0.0; s[i].nstep = 1;
    }
    t = 0.0;
    Zero(); Forces(); GlobalTimestep(); EndTimestep(0);
  }
  void Step() {                                   // NbodySimulation::MainLoop, NbodySimulation.cpp:311-404
    t = t + timestep;
    Advance(1); Zero(); Forces(); Correct(1); GlobalTimestep(); EndTimestep(0);
  }
};

// ---------------------------------------------------------------------------------------------
// C ABI for ctypes
// ---------------------------------------------------------------------------------------------
extern "C" {

struct orc_params {
  int32_t ndim, Nleafmax, self_gravity, periodic[3], energy_integration, nthreads, kernel, multipole, acond, gravity_mac, tdavisc, Nlevels, level_diff_max, sph_single_timestep, gas_eos, ntreebuildstep, ntreestockstep, pad3_;
  double boxmin[3], boxmax[3], h_fac, h_converge, alpha_visc, beta_visc, gamma_eos, thetamaxsqd, courant_mult, accel_mult, energy_mult, macerror, alpha_visc_min, temp0, mu_bar, rho_bary;
};

Oracle *orc_create(const orc_params *q)
{
  Params P;
  P.ndim = q->ndim; P.Nleafmax = q->Nleafmax; P.self_gravity = q->self_gravity; P.energy_integration = q->energy_integration;
  P.nthreads = q->nthreads > 0 ? q->nthreads : 1;
  P.kernel = q->kernel; P.multipole = q->multipole; P.acond = q->acond; P.gravity_mac = q->gravity_mac; P.macerror = q->macerror; P.tdavisc = q->tdavisc; P.alpha_visc_min = q->alpha_visc_min;
  P.ntreebuildstep = q->ntreebuildstep > 1 ? q->ntreebuildstep : 1; P.ntreestockstep = q->ntreestockstep > 1 ? q->ntreestockstep : 1;
  P.gas_eos = q->gas_eos; P.temp0 = q->temp0; P.mu_bar = q->mu_bar; P.rho_bary = q->rho_bary;
  P.Nlevels = q->Nlevels > 1 ? q->Nlevels : 1; P.level_diff_max = q->level_diff_max; P.sph_single_timestep = q->sph_single_timestep;
  for (int k = 0; k < 3; k++) {
    P.periodic[k] = q->periodic[k] == 1; P.mirror[k][0] = (q->periodic[k] & 2) != 0; P.mirror[k][1] = (q->periodic[k] & 4) != 0;
    P.boxmin[k] = q->boxmin[k]; P.boxmax[k] = q->boxmax[k];
    P.boxsize[k] = q->boxmax[k] - q->boxmin[k]; P.boxhalf[k] = 0.5*P.boxsize[k];
  }
  P.h_fac = q->h_fac; P.h_converge = q->h_converge; P.alpha_visc = q->alpha_visc; P.beta_visc = q->beta_visc; P.gamma = q->gamma_eos;
  P.thetamaxsqd = q->thetamaxsqd; P.courant_mult = q->courant_mult; P.accel_mult = q->accel_mult; P.energy_mult = q->energy_mult;
  return new Oracle(P);
}
void orc_destroy(Oracle *o) { delete o; }

void orc_set_particles(Oracle *o, int N, const double *r, const double *v, const double *m, const double *h, const double *u)
{
  const int nd = o->P.ndim;
  o->p.assign(N, Part());
  o->Nhydro = N; o->Nghost = 0;
  for (int i = 0; i < N; i++) {
    Part &q = o->p[i];
    memset(&q, 0, sizeof(Part));
    q.flags = F_ACTIVE; q.iorig = i; q.sinkid = -1;
    for (int k = 0; k < nd; k++) { q.r[k] = r[i*nd + k]; q.r0[k] = q.r[k]; q.v[k] = v ? v[i*nd + k] : 0.0; q.v0[k] = q.v[k]; }
    q.m = m[i]; q.h = h[i]; q.u = u ? u[i] : 0.0; q.u0 = q.u;
    q.alpha = o->P.tdavisc ? o->P.alpha_visc_min : o->P.alpha_visc;   // SphSimulation.cpp:252-257
  }
  o->t = 0.0; o->timestep = 0.0; o->n = 0; o->Nsteps = 0;
}

static double *field_ptr(Part &q, const char *name, int *ncomp)
{
  *ncomp = 1;
#define V(nm) if (!strcmp(name, #nm)) { *ncomp = 3; return q.nm; }
#define S(nm) if (!strcmp(name, #nm)) return &q.nm;
  V(r) V(v) V(a) V(atree) V(r0) V(v0) V(a0)
  S(m) S(h) S(hrangesqd) S(hfactor) S(sound) S(rho) S(pressure) S(u) S(u0) S(dudt0) S(dudt) S(gpot) S(gpot_hydro)
  S(dt) S(dt_next) S(tlast) S(div_v) S(invomega) S(zeta) S(alpha) S(dalphadt)
#undef V
#undef S
  return nullptr;
}
int orc_get(Oracle *o, const char *name, double *out)
{
  const int nd = o->P.ndim;
  for (int i = 0; i < o->Nhydro; i++) {
    int nc; double *f = field_ptr(o->p[i], name, &nc);
    if (!f) return -1;
    if (nc == 3) for (int k = 0; k < nd; k++) out[i*nd + k] = f[k]; else out[i] = *f;
  }
  return 0;
}
int orc_set(Oracle *o, const char *name, const double *in)
{
  const int nd = o->P.ndim;
  for (int i = 0; i < o->Nhydro; i++) {
    int nc; double *f = field_ptr(o->p[i], name, &nc);
    if (!f) return -1;
    if (nc == 3) for (int k = 0; k < nd; k++) f[k] = in[i*nd + k]; else *f = in[i];
  }
  return 0;
}
void orc_set_time(Oracle *o, double t, double timestep) { o->t = t; o->timestep = timestep; }
// hybrid runs: the stars the gas sees (positions, masses, smoothing lengths); accelerations / potential start at zero
void orc_set_stars(Oracle *o, int N, const double *r, const double *m, const double *h, int nbody_softening)
{
  o->star_softening = nbody_softening;
  o->stars.assign(N, GasStar());
  for (int i = 0; i < N; i++) {
    GasStar &s = o->stars[i];
    for (int k = 0; k < 3; k++) { s.r[k] = k < o->P.ndim ? r[i*o->P.ndim + k] : 0.0; s.a[k] = 0.0; }
    s.m = m[i]; s.h = h[i]; s.gpot = 0.0;
  }
}
// stars <- gas (the tree must have been built); out_a [N][ndim], out_gpot [N]
void orc_star_gas_forces(Oracle *o, double *out_a, double *out_gpot)
{
  for (size_t i = 0; i < o->stars.size(); i++) { for (int k = 0; k < 3; k++) o->stars[i].a[k] = 0.0; o->stars[i].gpot = 0.0; }
  o->UpdateAllStarGasForces();
  for (size_t i = 0; i < o->stars.size(); i++) {
    for (int k = 0; k < o->P.ndim; k++) out_a[i*o->P.ndim + k] = o->stars[i].a[k];
    out_gpot[i] = o->stars[i].gpot;
  }
}
static int *ifield_ptr(Part &q, const char *name)
{
#define I(nm) if (!strcmp(name, #nm)) return &q.nm;
  I(level) I(levelneib) I(nstep) I(nlast) I(flags) I(sinkid) I(iorig)
#undef I
  return nullptr;
}
int orc_get_int(Oracle *o, const char *name, int *out)
{
  for (int i = 0; i < o->Nhydro; i++) { int *f = ifield_ptr(o->p[i], name); if (!f) return -1; out[i] = *f; }
  return 0;
}
int orc_set_int(Oracle *o, const char *name, const int *in)
{
  for (int i = 0; i < o->Nhydro; i++) { int *f = ifield_ptr(o->p[i], name); if (!f) return -1; *f = in[i]; }
  return 0;
}
// block-timestep clock: {n, nresync, level_max, level_step} and dt_max
void orc_set_block(Oracle *o, const int *v, double dt_max) { o->n = v[0]; o->nresync = v[1]; o->level_max = v[2]; o->level_step = v[3]; o->dt_max = dt_max; }
double orc_get_block(Oracle *o, int *v) { v[0] = o->n; v[1] = o->nresync; v[2] = o->level_max; v[3] = o->level_step; return o->dt_max; }
double orc_time(Oracle *o) { return o->t; }
double orc_timestep(Oracle *o) { return o->timestep; }
void orc_set_all_active(Oracle *o) { for (int i = 0; i < o->Nhydro; i++) o->p[i].flags |= F_ACTIVE; }

void orc_build_tree(Oracle *o) { o->BuildTree(); }
void orc_tree_size(Oracle *o, int *Ncell, int *ltot, int *gtot) { *Ncell = o->tree.Ncell; *ltot = o->tree.ltot; *gtot = o->tree.gtot; }
void orc_export_tree(Oracle *o, int *level, int *ifirst, int *ilast, int *N, int *inext, double *bbmin, double *bbmax,
                     double *hbmin, double *hbmax, double *rcell, double *com, double *m, double *rmax, double *hmax, double *cdistsqd)
{
  const int nd = o->P.ndim;
  for (int c = 0; c < o->tree.Ncell; c++) {
    const Cell &x = o->tree.cell[c];
    level[c] = x.level; ifirst[c] = x.ifirst; ilast[c] = x.ilast; N[c] = x.N;
    for (int k = 0; k < nd; k++) { bbmin[c*nd + k] = x.bbmin[k]; bbmax[c*nd + k] = x.bbmax[k]; hbmin[c*nd + k] = x.hbmin[k];
                                   hbmax[c*nd + k] = x.hbmax[k]; rcell[c*nd + k] = x.rcell[k]; com[c*nd + k] = x.r[k]; }
    m[c] = x.m; rmax[c] = x.rmax; hmax[c] = x.hmax; cdistsqd[c] = x.cdistsqd;
  }
  for (int i = 0; i < o->Nhydro; i++) inext[i] = o->tree.inext[i];
}
int orc_density(Oracle *o) { return o->DensityPass(), 0; }
void orc_zero_accelerations(Oracle *o) { o->ZeroAccelerations(); }
void orc_forces(Oracle *o) { o->Forces(); }
void orc_setup(Oracle *o, int h_provided) { o->Setup(h_provided); }
void orc_step(Oracle *o, int nsteps) { for (int s = 0; s < nsteps; s++) o->MainLoop(); }
int orc_num_ghosts(Oracle *o) { return o->Nghost; }

// HydroTree::GetGatherNeighbourList for every particle (real + ghost trees; ghost ids mapped to their real
// parent), CSR.  Needs the ghost tree of the last density pass.  Returns total count (ids may be NULL).
long orc_gather_neighbours(Oracle *o, long *offsets, int *ids)
{
  long tot = 0;
  std::vector<int> list;
  offsets[0] = 0;
  for (int i = 0; i < o->Nhydro; i++) {
    list.clear();
    o->tree.GatherPoint(o->p[i].r, o->kern.kernrange*o->p[i].h, o->p, list);
    o->ghosttree.GatherPoint(o->p[i].r, o->kern.kernrange*o->p[i].h, o->p, list);
    for (size_t k = 0; k < list.size(); k++) {
      int j = list[k];
      while (j >= o->Nhydro) j = o->p[j].iorig;
      if (ids) ids[tot + k] = j;
    }
    tot += (long) list.size();
    offsets[i + 1] = tot;
  }
  return tot;
}

NbodyOracle *orc_nbody_create(int N, int softening, double nbody_mult, const double *r, const double *v, const double *m, const double *h)
{
  NbodyOracle *o = new NbodyOracle(N, softening, nbody_mult);
  for (int i = 0; i < N; i++) {
    Star &q = o->s[i];
    memset(&q, 0, sizeof(Star));
    for (int k = 0; k < 3; k++) { q.r[k] = r[3*i + k]; q.v[k] = v[3*i + k]; }
    q.m = m[i]; q.h = h[i]; q.dt_internal = big_number;
    q.active = true;
  }
  return o;
}
void orc_nbody_destroy(NbodyOracle *o) { delete o; }
void orc_nbody_forces(NbodyOracle *o) { for (int i = 0; i < o->N; i++) o->s[i].active = true; o->Zero(); o->Forces(); }
void orc_nbody_setup(NbodyOracle *o) { o->Setup(); }
void orc_nbody_step(NbodyOracle *o, int n) { for (int i = 0; i < n; i++) o->Step(); }
double orc_nbody_time(NbodyOracle *o) { return o->t; }
double orc_nbody_timestep(NbodyOracle *o) { return o->timestep; }
// field: 0 r, 1 v, 2 a, 3 adot, 4 gpot, 5 r0, 6 v0, 7 a0
// post-setup state of the stars of a hybrid run: fields 0 r, 1 v, 2 a, 3 adot, 5 r0, 6 v0, 7 a0, 8 adot0 ([N][3]);
// 4 gpot, 9 dt, 10 tlast ([N]).  nstep = 1, nlast = 0 (global timestep)
void orc_nbody_set(NbodyOracle *o, int field, const double *in)
{
  for (int i = 0; i < o->N; i++) {
    Star &q = o->s[i];
    FLOAT *dst = field == 0 ? q.r : field == 1 ? q.v : field == 2 ? q.a : field == 3 ? q.adot : field == 5 ? q.r0 :
                 field == 6 ? q.v0 : field == 7 ? q.a0 : field == 8 ? q.adot0 : NULL;
    if (field == 4) q.gpot = in[i]; else if (field == 9) q.dt = in[i]; else if (field == 10) q.tlast = in[i];
    else for (int k = 0; k < 3; k++) dst[k] = in[3*i + k];
    q.nstep = 1; q.nlast = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Sink particles: Sinks::SearchForNewSinkParticles / CreateNewSinkParticle / AccreteMassToSinks, Sinks.cpp:118-777
// (one thread, no MPI).  Sinks are stars of the N-body oracle; the gas oracle holds the SinkParticle records.
// ---------------------------------------------------------------------------------------------
static void CreateNewSinkParticle(Oracle &g, NbodyOracle &nb, int isink, FLOAT t)      // Sinks.cpp:282-356
{
  const int nd = g.P.ndim;
  Part &part = g.p[isink];
  Sink sk;
  memset(&sk, 0, sizeof(Sink));
  Star st;
  memset(&st, 0, sizeof(Star));
  st.dt_internal = big_number;                                        // StarParticle constructor
  if (g.P.sink_radius_mode == 0) sk.radius = g.P.sink_radius;
  else if (g.P.sink_radius_mode == 1) sk.radius = g.P.sink_radius*part.h;
  else sk.radius = g.kern.kernrange*part.h;
  st.h = g.kern.invkernrange*sk.radius;
  st.invh = (FLOAT) 1.0/part.h;
  st.m = part.m; st.gpot = part.gpot;
  st.dt = part.dt; st.tlast = t; st.nstep = part.nstep; st.nlast = part.nlast; st.level = part.level;
  st.active = (part.flags & F_ACTIVE) != 0;
  for (int k = 0; k < nd; k++) {
    st.r[k] = part.r[k]; st.v[k] = part.v[k]; st.a[k] = part.a[k]; st.adot[k] = 0.0;
    st.r0[k] = part.r0[k]; st.v0[k] = part.v0[k]; st.a0[k] = part.a0[k]; st.adot0[k] = 0.0;
  }
  sk.istar = nb.N;
  part.m = 0.0;
  part.flags &= ~F_ACTIVE;
  part.flags |= F_DEAD;
  nb.s.push_back(st);                                                  // nbody->Nstar++ / Nnbody++ (:269-270) below
  g.sinks.push_back(sk);
}

static void SearchForNewSinkParticles(Oracle &g, NbodyOracle &nb, int n, FLOAT t)      // Sinks.cpp:118-273
{
  const int nd = g.P.ndim;
  if ((int) g.sinks.size() >= g.P.Nsinkfixed && g.P.Nsinkfixed != -1) return;
  int isink;
  do {
    isink = -1;
    FLOAT rho_max = 0.0;
    const int Nsink = (int) g.sinks.size();
    for (int i = 0; i < g.Nhydro; i++) {
      bool sink_flag = true;
      const Part &part = g.p[i];
      if (part.flags & F_DEAD) continue;
      if (!(part.flags & F_POTMIN)) continue;
      if (part.rho < g.P.rho_sink || part.rho < rho_max) continue;
      if (n%part.nstep != 0) continue;
      for (int s = 0; s < Nsink; s++) {
        const Star &st = nb.s[g.sinks[s].istar];
        FLOAT dr[3], da[3], dv[3];
        for (int k = 0; k < nd; k++) dr[k] = part.r[k] - st.r[k];
        for (int k = 0; k < nd; k++) da[k] = part.a[k] - st.a[k];
        for (int k = 0; k < nd; k++) dv[k] = part.v[k] - st.v[k];
        const FLOAT drsqd = Dot(dr, dr, nd), dadr = Dot(dr, da, nd), dvdr = Dot(dr, dv, nd);
        const FLOAT tff = (FLOAT) 0.5/sqrt(part.rho);
        if (tff > drsqd/dvdr && dvdr > 0) sink_flag = false;
        if (part.rho < -dadr/drsqd) sink_flag = false;
        if (drsqd < pow(g.P.sink_radius*part.h + g.sinks[s].radius, 2)) sink_flag = false;
        if (!sink_flag) break;
      }
      if (sink_flag && part.rho > rho_max) { isink = i; rho_max = part.rho; }
    }
    if (isink >= 0) {
      CreateNewSinkParticle(g, nb, isink, t);
      // total gas mass inside the new sink (direct sum, :246-253)
      Sink &sk = g.sinks.back();
      const Star &st = nb.s[sk.istar];
      sk.mmax = 0.0;
      for (int i = 0; i < g.Nhydro; i++) {
        const Part &part = g.p[i];
        if (part.flags & F_DEAD) continue;
        FLOAT dr[3];
        for (int k = 0; k < nd; k++) dr[k] = st.r[k] - part.r[k];
        if (Dot(dr, dr, nd) < pow(sk.radius, 2)) sk.mmax += part.m;
      }
      nb.N++;
    }
  } while (isink != -1);
}

static void AccreteMassToSinks(Oracle &g, NbodyOracle &nb, int n, FLOAT timestep)      // Sinks.cpp:365-770
{
  const int nd = g.P.ndim;
  const int Nsink = (int) g.sinks.size();
  for (int i = 0; i < g.Nhydro + g.Nghost; i++) g.p[i].sinkid = -1;
  for (int s = 0; s < Nsink; s++) g.sinks[s].Ngas = 0;
  std::vector<int> neiblist, ilist;
  std::vector<FLOAT> rsqdlist;
  FLOAT dr[3], dv[3], dvtang[3];
  // which sink each particle accretes to (:427-468; the last sink in whose radius it lies)
  for (int s = 0; s < Nsink; s++) {
    Sink &sk = g.sinks[s];
    const Star &st = nb.s[sk.istar];
    neiblist.clear();
    g.tree.GatherPoint(st.r, sk.radius, g.p, neiblist);
    for (size_t j = 0; j < neiblist.size(); j++) {
      Part &part = g.p[neiblist[j]];
      if (part.flags & F_DEAD) continue;
      for (int k = 0; k < nd; k++) dr[k] = part.r[k] - st.r[k];
      if (Dot(dr, dr, nd) <= sk.radius*sk.radius) { part.sinkid = s; sk.Ngas++; }
    }
  }
  for (int s = 0; s < Nsink; s++) {
    Sink &sk = g.sinks[s];
    Star &st = nb.s[sk.istar];
    if (sk.Ngas == 0 || n%st.nstep != 0) continue;
    FLOAT wnorm = 0.0;
    sk.menc = 0.0; sk.trad = 0.0; sk.tvisc = 1.0; sk.ketot = 0.0; sk.rotketot = 0.0; sk.gpetot = 0.0;
    neiblist.clear(); ilist.clear(); rsqdlist.clear();
    g.tree.GatherPoint(st.r, sk.radius, g.p, neiblist);
    for (size_t j = 0; j < neiblist.size(); j++) {
      const int i = neiblist[j];
      Part &part = g.p[i];
      if (part.flags & F_DEAD) continue;
      if (part.sinkid == s) {
        for (int k = 0; k < nd; k++) dr[k] = part.r[k] - st.r[k];
        const FLOAT drsqd = Dot(dr, dr, nd);
        if (drsqd > sk.radius*sk.radius) continue;
        ilist.push_back(i); rsqdlist.push_back(drsqd);
        part.levelneib = std::max(part.levelneib, st.level);
      }
    }
    const int Nneib = (int) ilist.size();
    for (int j = 1; j < Nneib; j++) {                                  // InsertionSortIds, InlineFuncs.h:226-248
      const FLOAT raux = rsqdlist[j]; const int iaux = ilist[j];
      int i;
      for (i = j - 1; i >= 0; i--) { if (rsqdlist[i] <= raux) break; rsqdlist[i + 1] = rsqdlist[i]; ilist[i + 1] = ilist[i]; }
      rsqdlist[i + 1] = raux; ilist[i + 1] = iaux;
    }
    for (int j = 0; j < Nneib; j++) {                                  // :524-561
      const Part &part = g.p[ilist[j]];
      if (part.flags & F_DEAD) continue;
      for (int k = 0; k < nd; k++) dr[k] = part.r[k] - st.r[k];
      const FLOAT drsqd = Dot(dr, dr, nd);
      const FLOAT drmag = sqrt(drsqd) + small_number;
      for (int k = 0; k < nd; k++) dr[k] /= drmag;
      sk.menc += part.m;
      wnorm += part.m*g.kern.w0(drmag*st.invh)*pow(st.invh, nd)/part.rho;
      sk.gpetot += (FLOAT) 0.5*part.m*(st.m + sk.menc)*st.invh*g.kern.wpot(drmag*st.invh);
      for (int k = 0; k < nd; k++) dv[k] = part.v[k] - st.v[k];
      for (int k = 0; k < nd; k++) dvtang[k] = dv[k] - Dot(dv, dr, nd)*dr[k];
      sk.ketot += part.m*Dot(dv, dv, nd)*g.kern.w0(drmag*st.invh)*pow(st.invh, nd)/part.rho;
      sk.rotketot += part.m*Dot(dvtang, dvtang, nd)*g.kern.w0(drmag*st.invh)*pow(st.invh, nd)/part.rho;
      sk.tvisc *= pow(sqrt(drmag)/part.sound/part.sound, part.m);
      sk.trad += fabs((FLOAT) 4.0*pi_const*drsqd*part.m*Dot(dv, dr, nd)*g.kern.w0(drmag*st.invh)*pow(st.invh, nd));
    }
    sk.ketot *= (FLOAT) 0.5*sk.menc/wnorm;
    sk.rotketot *= (FLOAT) 0.5*sk.menc/wnorm;
    FLOAT macc, dt;
    if (g.P.smooth_accretion == 1) {                                   // :573-602
      const FLOAT efrac = std::min((FLOAT) 2.0*sk.rotketot/sk.gpetot, (FLOAT) 1.0);
      sk.tvisc = (sqrt(st.m + sk.menc)*pow(sk.tvisc, (FLOAT) 1.0/sk.menc))/g.P.alpha_ss;
      sk.trad = sk.menc/sk.trad;
      sk.trot = twopi*sqrt(pow(sk.radius, 3)/(sk.menc + st.m));
      sk.taccrete = pow(sk.trad, (FLOAT) 1.0 - efrac)*pow(sk.tvisc, efrac);
      if (sk.mmax > small_number && sk.menc > sk.mmax) sk.taccrete *= pow(sk.mmax/sk.menc, 2);
      dt = (FLOAT) st.nstep*timestep;
      macc = sk.menc*std::max((FLOAT) 1.0 - (FLOAT) exp(-dt/sk.taccrete), (FLOAT) 0.0);
      sk.dmdt = macc/dt;
    }
    else { macc = sk.menc; sk.dmdt = macc/timestep; }
    FLOAT macc_temp = macc, rold[3], vold[3];
    for (int k = 0; k < nd; k++) { rold[k] = st.r[k]; vold[k] = st.v[k]; }
    const FLOAT mold = st.m;
    for (int k = 0; k < nd; k++) { st.r[k] *= st.m; st.v[k] *= st.m; st.a[k] *= st.m; }
    for (int j = 0; j < Nneib; j++) {                                  // :626-654
      const Part &part = g.p[ilist[j]];
      if (part.flags & F_DEAD) continue;
      FLOAT mtemp = std::min(part.m, macc_temp);
      dt = part.dt;
      if (g.P.smooth_accretion == 0 || part.m - mtemp < g.P.smooth_accrete_frac*g.mmean || dt < g.P.smooth_accrete_dt*sk.trot) mtemp = part.m;
      macc_temp -= mtemp;
      st.m += mtemp;
      for (int k = 0; k < nd; k++) { st.r[k] += mtemp*part.r[k]; st.v[k] += mtemp*part.v[k]; st.a[k] += mtemp*part.a[k]; }
      sk.utot += mtemp*part.u;
      if (macc_temp < small_number) break;
    }
    for (int k = 0; k < nd; k++) { st.r[k] /= st.m; st.v[k] /= st.m; st.a[k] /= st.m; }
    for (int k = 0; k < nd; k++) { st.r0[k] = st.r[k]; st.v0[k] = st.v[k]; st.a0[k] = st.a[k]; }
    for (int k = 0; k < nd; k++) { dr[k] = rold[k] - st.r[k]; dv[k] = vold[k] - st.v[k]; }
    if (nd == 3) {
      sk.angmom[0] += mold*(dr[1]*dv[2] - dr[2]*dv[1]);
      sk.angmom[1] += mold*(dr[2]*dv[0] - dr[0]*dv[2]);
      sk.angmom[2] += mold*(dr[0]*dv[1] - dr[1]*dv[0]);
    }
    else if (nd == 2) sk.angmom[2] += mold*(dr[0]*dv[1] - dr[1]*dv[0]);
    for (int j = 0; j < Nneib; j++) {                                  // :685-729
      Part &part = g.p[ilist[j]];
      if (part.flags & F_DEAD) continue;
      FLOAT mtemp = std::min(part.m, macc);
      dt = part.dt;
      if (g.P.smooth_accretion == 0 || part.m - mtemp < g.P.smooth_accrete_frac*g.mmean || dt < g.P.smooth_accrete_dt*sk.trot) {
        mtemp = part.m; part.m = 0.0; part.flags |= F_DEAD; part.flags &= ~F_ACTIVE;
      }
      else part.m -= mtemp;                                            // Sph::AccreteMassFromParticle, Sph.h:108
      macc -= mtemp;
      for (int k = 0; k < nd; k++) { dr[k] = part.r[k] - st.r[k]; dv[k] = part.v[k] - st.v[k]; }
      if (nd == 3) {
        sk.angmom[0] += mtemp*(dr[1]*dv[2] - dr[2]*dv[1]);
        sk.angmom[1] += mtemp*(dr[2]*dv[0] - dr[0]*dv[2]);
        sk.angmom[2] += mtemp*(dr[0]*dv[1] - dr[1]*dv[0]);
      }
      else if (nd == 2) sk.angmom[2] += mtemp*(dr[0]*dv[1] - dr[1]*dv[0]);
      if (macc < small_number) break;
    }
    const FLOAT asqd = Dot(st.a, st.a, nd);
    st.dt_internal = (FLOAT) 0.4*sqrt(sk.radius/(sqrt(asqd) + small_number));
  }
}

// Simulation::ComputeBlockTimesteps (Simulation.cpp:1764-2200) for gas + stars: the gas part as Oracle::ComputeBlockTimesteps,
// with the dead-particle skips and the star branches (stars sit on levels >= the highest gas level)
static double StarTimestep(const NbodyOracle &nb, const Star &s)       // NbodyLeapfrogKDK::Timestep, :387-400
{
  const double amag = sqrt(Dot(s.a, s.a, 3));
  double ts = nb.nbody_mult*sqrt(s.h/(amag + small_number_dp));
  return std::min(ts, (double) s.dt_internal);
}
static void ComputeBlockTimestepsHybrid(Oracle &g, NbodyOracle &nb)
{
  const Params &P = g.P;
  auto ipow2 = [](int e) { return (int) pow(2.0, e); };
  auto lvl = [](double dt, double dt_max) { return std::max((int) (invlogetwo*log(dt_max/dt)) + 1, 0); };
  if (g.n == g.nresync) {
    g.n = 0; g.timestep = big_number_dp;
    double dt_min_hydro = big_number_dp, dt_min_nbody = big_number_dp;
    for (int i = 0; i < g.Nhydro; i++) {
      Part &q = g.p[i];
      if (q.flags & F_DEAD) continue;
      const double dt = g.Timestep(q);
      g.timestep = std::min(g.timestep, dt); dt_min_hydro = std::min(dt_min_hydro, dt);
      q.dt_next = dt;
    }
    for (int i = 0; i < nb.N; i++) {
      const double dt = StarTimestep(nb, nb.s[i]);
      g.timestep = std::min(g.timestep, dt); dt_min_nbody = std::min(dt_min_nbody, dt);
      nb.s[i].dt_next = dt;
    }
    g.level_max = P.Nlevels - 1;
    g.level_step = g.level_max + g.integration_step - 1;
    g.dt_max = g.timestep*pow(2.0, g.level_max);
    int level_max_hydro = std::min(lvl(dt_min_hydro, g.dt_max), g.level_max);
    for (int i = 0; i < nb.N; i++) {                                   // :1862-1873
      Star &s = nb.s[i];
      const int level = std::min(lvl(s.dt_next, g.dt_max), g.level_max);
      s.level = std::max(level, level_max_hydro);
      s.nlast = g.n; s.nstep = ipow2(g.level_step - s.level); s.tlast = g.t;
      s.dt_next = s.nstep*g.timestep; s.end_timestep = true;
    }
    for (int i = 0; i < g.Nhydro; i++) {                               // sink neighbours, :1876-1887
      Part &q = g.p[i];
      if (q.sinkid != -1) {
        const int sl = nb.s[g.sinks[q.sinkid].istar].level;
        if (sl - q.level > P.level_diff_max) { q.level = sl - P.level_diff_max; q.levelneib = sl; level_max_hydro = std::max(level_max_hydro, q.level); }
      }
    }
    for (int i = 0; i < g.Nhydro; i++) {
      Part &q = g.p[i];
      if (q.flags & F_DEAD) continue;
      const int level = P.sph_single_timestep ? level_max_hydro : std::min(lvl(q.dt_next, g.dt_max), g.level_max);
      q.level = level; q.levelneib = level;
      q.nstep = ipow2(g.level_step - q.level);
      q.nlast = g.n;
      q.dt_next = q.nstep*g.timestep;
      q.flags |= F_END;
    }
    g.nresync = ipow2(g.level_step);
    g.timestep = g.dt_max/(double) g.nresync;
    nb.timestep = g.timestep;
    return;
  }
  const int level_max_old = g.level_max;
  g.level_max = 0;
  int level_max_hydro = 0;
  const int n = g.n;
  for (int i = 0; i < g.Nhydro; i++) {
    Part &q = g.p[i];
    if (q.flags & F_DEAD) continue;
    if (n - q.nlast == q.nstep && q.nstep != ipow2(g.level_step - q.level)) {
      const double dt = g.Timestep(q);
      const int level = std::max(lvl(dt, g.dt_max), q.levelneib - P.level_diff_max);
      q.level = std::max(q.level, level);
      q.levelneib = q.level;
      q.nlast = n; q.nstep = ipow2(g.level_step - q.level);
      q.dt_next = q.nstep*g.timestep; q.flags |= F_END;
    }
    else if (n - q.nlast == q.nstep) {
      const int nstep = q.nstep, last_level = q.level;
      const double dt = g.Timestep(q);
      const int level = std::max(lvl(dt, g.dt_max), q.levelneib - P.level_diff_max);
      if (level < last_level && last_level > 1 && n%(2*nstep) == 0) q.level = last_level - 1;
      else if (level > last_level) q.level = level;
      else q.level = last_level;
      q.levelneib = level;
      q.nlast = n; q.nstep = ipow2(g.level_step - q.level);
      q.dt_next = q.nstep*g.timestep; q.flags |= F_END;
    }
    level_max_hydro = std::max(level_max_hydro, q.level);
    g.level_max = std::max(g.level_max, q.level);
  }
  for (int i = 0; i < nb.N; i++) {                                     // :2024-2060
    Star &s = nb.s[i];
    if (n - s.nlast == s.nstep) {
      const int nstep = s.nstep, last_level = s.level;
      const double dt = StarTimestep(nb, s);
      const int level = std::max(lvl(dt, g.dt_max), level_max_hydro);
      if (level < last_level && level > level_max_hydro && last_level > 1 && n%(2*nstep) == 0) s.level = last_level - 1;
      else if (level > last_level) s.level = level;
      else s.level = last_level;
      s.nlast = n; s.nstep = ipow2(g.level_step - s.level); s.tlast = g.t;
      s.dt_next = s.nstep*g.timestep; s.end_timestep = true;
    }
    g.level_max = std::max(g.level_max, s.level);
  }
  if (P.sph_single_timestep) for (int i = 0; i < g.Nhydro; i++) if (!(g.p[i].flags & F_DEAD) && g.p[i].nlast == g.n) g.p[i].level = level_max_hydro;
  const int istep = ipow2(g.level_step - level_max_old + 1);
  if (g.level_max > level_max_old) {
    const int nfactor = ipow2(g.level_max - level_max_old);
    g.n *= nfactor;
    for (int i = 0; i < g.Nhydro; i++) { if (g.p[i].flags & F_DEAD) continue; g.p[i].nstep *= nfactor; g.p[i].nlast *= nfactor; }
    for (int i = 0; i < nb.N; i++) { nb.s[i].nstep *= nfactor; nb.s[i].nlast *= nfactor; }
  }
  else if (g.level_max <= level_max_old - 1 && level_max_old > 1 && g.n%istep == 0) {
    g.level_max = level_max_old - 1;
    const int nfactor = ipow2(level_max_old - g.level_max);
    g.n /= nfactor;
    for (int i = 0; i < g.Nhydro; i++) { if (g.p[i].flags & F_DEAD) continue; g.p[i].nlast /= nfactor; g.p[i].nstep /= nfactor; }
    for (int i = 0; i < nb.N; i++) { nb.s[i].nlast /= nfactor; nb.s[i].nstep /= nfactor; }
  }
  else g.level_max = level_max_old;
  g.level_step = g.level_max + g.integration_step - 1;
  g.nresync = ipow2(g.level_step);
  g.timestep = g.dt_max/(double) g.nresync;
  nb.timestep = g.timestep;
  for (int i = 0; i < g.Nhydro; i++) if (!(g.p[i].flags & F_DEAD) && g.p[i].nlast == g.n) g.p[i].nstep = ipow2(g.level_step - g.p[i].level);
  for (int i = 0; i < nb.N; i++) if (nb.s[i].nlast == g.n) nb.s[i].nstep = ipow2(g.level_step - nb.s[i].level);
}

// the sink part of MainLoop, SphSimulation.cpp:820-838 (ntreebuildstep = 1: the search runs on every step)
static void SinkStep(Oracle &g, NbodyOracle &nb)
{
  if (g.P.sink_particles != 1) return;
  if (g.P.create_sinks == 1) SearchForNewSinkParticles(g, nb, g.n, g.t);
  if (!g.sinks.empty()) {
    g.mmean = (FLOAT) 0.0;
    for (int i = 0; i < g.Nhydro; i++) g.mmean += g.p[i].m;
    g.mmean /= (FLOAT) g.Nhydro;
    AccreteMassToSinks(g, nb, g.n, g.timestep);
  }
}

// One SphSimulation::MainLoop call of a hybrid gas + stars run with a global timestep (SphSimulation.cpp:574-880, Npec = 1):
// both species advance, the gas passes see the stars (zeta term, ComputeStarGravForces), the stars get the gas' tree
// forces and their own direct sum, the timestep is the minimum over both (Simulation.cpp:1669-1754)
static void HybridMainLoop(Oracle &g, NbodyOracle &nb)
{
  g.n++; g.Nsteps++; g.t = g.t + g.timestep;
  nb.t = g.t; nb.timestep = g.timestep;
  g.AdvanceParticles();
  nb.Advance(g.n);
  g.stars.resize(nb.N);
  g.star_softening = nb.softening;
  for (int i = 0; i < nb.N; i++) { for (int k = 0; k < 3; k++) g.stars[i].r[k] = nb.s[i].r[k]; g.stars[i].m = nb.s[i].m; g.stars[i].h = nb.s[i].h; }
  if (g.P.Nlevels > 1) {                                             // block timesteps: SphSimulation.cpp:654-755
    g.StepTree(); g.SearchBoundaryGhostParticles(); g.BuildGhostTree();
    int activecount = 0;
    do {
      if (activecount > 0) g.UpdateActiveParticleCounters();
      g.UpdateAllSphProperties();
      g.ZeroAccelerations();
      for (int i = 0; i < g.Nhydro; i++) g.Thermal(g.p[i]);
      g.Forces();
      for (int i = 0; i < g.Nhydro; i++) g.p[i].flags &= ~F_ACTIVE;
      activecount = g.CheckTimesteps();
    } while (activecount > 0);
  }
  else {
    g.StepTree(); g.DensityPass();
    g.ZeroAccelerations(); g.Forces();
  }
  nb.Zero();                                                         // :773-784
  for (int i = 0; i < nb.N; i++) { for (int k = 0; k < 3; k++) g.stars[i].a[k] = 0.0; g.stars[i].gpot = 0.0; }
  g.UpdateAllStarGasForces();                                        // :787
  for (int i = 0; i < nb.N; i++) if (nb.s[i].active) { for (int k = 0; k < 3; k++) nb.s[i].a[k] = g.stars[i].a[k]; nb.s[i].gpot = g.stars[i].gpot; }
  nb.Forces();                                                       // :794-799 (adds the star-star sums)
  nb.Correct(g.n);                                                   // :811
  SinkStep(g, nb);                                                   // :820-838
  if (g.P.Nlevels > 1) {
    ComputeBlockTimestepsHybrid(g, nb);
    g.EndTimestep();
    nb.EndTimestep(g.n);
    g.rebuild_tree = false;
    return;
  }
  g.ComputeGlobalTimestep();                                         // minimum over gas and stars
  nb.GlobalTimestep();
  const double ts = std::min(g.timestep, (double) nb.timestep);
  g.timestep = ts; nb.timestep = ts;
  for (int i = 0; i < g.Nhydro; i++) g.p[i].dt_next = ts;
  for (int i = 0; i < nb.N; i++) nb.s[i].dt_next = ts;
  g.EndTimestep();
  nb.EndTimestep(0);
  g.rebuild_tree = false;
}
// PostInitialConditionsSetup of a hybrid run (SphSimulation.cpp:204-565): the gas passes see the stars, the stars get the
// gas' tree forces and their own direct sum (:500-514), the first timestep is the minimum over both (:538), both end it
static void HybridSetup(Oracle &g, NbodyOracle &nb, int h_provided)
{
  g.stars.resize(nb.N);
  g.star_softening = nb.softening;
  for (int i = 0; i < nb.N; i++) {
    Star &s = nb.s[i];
    for (int k = 0; k < 3; k++) { g.stars[i].r[k] = s.r[k]; g.stars[i].a[k] = 0.0; s.r0[k] = s.r[k]; s.v0[k] = s.v[k]; }
    g.stars[i].m = s.m; g.stars[i].h = s.h; g.stars[i].gpot = 0.0;
    s.active = true; s.nlast = 0; s.tlast = 0.0; s.nstep = 1;
  }
  g.SetupPasses(h_provided);
  g.t = 0.0; g.n = 0; g.nresync = 0; nb.t = 0.0;
  nb.Zero();
  g.UpdateAllStarGasForces();
  for (int i = 0; i < nb.N; i++) { for (int k = 0; k < 3; k++) nb.s[i].a[k] = g.stars[i].a[k]; nb.s[i].gpot = g.stars[i].gpot; }
  nb.Forces();
  if (g.P.Nlevels > 1) {
    ComputeBlockTimestepsHybrid(g, nb);
    g.EndTimestep();
    nb.EndTimestep(g.n);
    return;
  }
  g.ComputeGlobalTimestep();
  nb.GlobalTimestep();
  const double ts = std::min(g.timestep, (double) nb.timestep);
  g.timestep = ts; nb.timestep = ts;
  for (int i = 0; i < g.Nhydro; i++) g.p[i].dt_next = ts;
  for (int i = 0; i < nb.N; i++) nb.s[i].dt_next = ts;
  g.EndTimestep();
  nb.EndTimestep(0);
}

// sink parameters: v = {sink_particles, create_sinks, smooth_accretion, sink_radius_mode (0 fixed, 1 hmult, 2 other), Nsinkfixed,
//                       rho_sink, sink_radius, alpha_ss, smooth_accrete_frac, smooth_accrete_dt}
void orc_set_sink_params(Oracle *o, const double *v)
{
  o->P.sink_particles = (int) v[0]; o->P.create_sinks = (int) v[1]; o->P.smooth_accretion = (int) v[2];
  o->P.sink_radius_mode = (int) v[3]; o->P.Nsinkfixed = (int) v[4];
  o->P.rho_sink = v[5]; o->P.sink_radius = v[6]; o->P.alpha_ss = v[7]; o->P.smooth_accrete_frac = v[8]; o->P.smooth_accrete_dt = v[9];
}
int orc_num_particles(Oracle *o) { return o->Nhydro; }
int orc_num_sinks(Oracle *o) { return (int) o->sinks.size(); }
double orc_mmean(Oracle *o) { return o->mmean; }
// per sink: {radius, mmax, menc, dmdt, ketot, gpetot, rotketot, utot, taccrete, trad, trot, tvisc, angmom[3]} (15 doubles), ints {istar, Ngas}
void orc_get_sinks(Oracle *o, double *out, int *iout)
{
  for (size_t s = 0; s < o->sinks.size(); s++) {
    const Sink &k = o->sinks[s];
    const double v[15] = {k.radius, k.mmax, k.menc, k.dmdt, k.ketot, k.gpetot, k.rotketot, k.utot, k.taccrete, k.trad, k.trot, k.tvisc,
                          k.angmom[0], k.angmom[1], k.angmom[2]};
    for (int q = 0; q < 15; q++) out[15*s + q] = v[q];
    iout[2*s] = k.istar; iout[2*s + 1] = k.Ngas;
  }
}
int orc_nbody_count(NbodyOracle *o) { return o->N; }
// scalar star fields: 0 m, 1 h, 2 dt_internal, 3 invh, 4 gpot, 5 dt, 6 tlast
void orc_nbody_get_scalar(NbodyOracle *o, int field, double *out)
{
  for (int i = 0; i < o->N; i++) {
    const Star &q = o->s[i];
    out[i] = field == 0 ? q.m : field == 1 ? q.h : field == 2 ? q.dt_internal : field == 3 ? q.invh : field == 4 ? q.gpot : field == 5 ? q.dt : q.tlast;
  }
}
void orc_hybrid_setup(Oracle *g, NbodyOracle *nb, int h_provided) { HybridSetup(*g, *nb, h_provided); }
void orc_hybrid_step(Oracle *g, NbodyOracle *nb, int nsteps) { for (int s = 0; s < nsteps; s++) HybridMainLoop(*g, *nb); }

void orc_nbody_get(NbodyOracle *o, int field, double *out)
{
  for (int i = 0; i < o->N; i++) {
    const Star &q = o->s[i];
    const FLOAT *src = field == 0 ? q.r : field == 1 ? q.v : field == 2 ? q.a : field == 3 ? q.adot : field == 5 ? q.r0 :
                       field == 6 ? q.v0 : field == 7 ? q.a0 : NULL;
    if (field == 4) out[i] = q.gpot;
    else for (int k = 0; k < 3; k++) out[3*i + k] = src[k];
  }
}

}
