// host_kernels.hpp -- the SPH kernel functions on the host, written as the reference writes them (with libm's pow):
// used to build the tables of tabulated_kernel = 1 (api.hip) and by the sink accretion sums (sinks.hip).
#pragma once
#include <cmath>
#include "sph_kernels.hpp"

namespace {
struct HostM4 {            // M4Kernel functions as the reference writes them (SmoothingKernel.h:131-240)
  int nd; double norm;
  explicit HostM4(int nd_) : nd(nd_) { norm = nd == 1 ? GH_TWOTHIRDS : (nd == 2 ? GH_INVPI*(10.0/7.0) : GH_INVPI); }
  double w0(double s) const { return s < 1.0 ? norm*(1.0 - 1.5*s*s + 0.75*s*s*s) : (s < 2.0 ? 0.25*norm*std::pow(2.0 - s, 3) : 0.0); }
  double w1(double s) const { return s < 1.0 ? norm*(-3.0*s + 2.25*s*s) : (s < 2.0 ? -0.75*norm*(2.0 - s)*(2.0 - s) : 0.0); }
  double womega(double s) const {
    if (s < 1.0) return norm*(-nd + 1.5*(nd + 2.0)*s*s - 0.75*(nd + 3.0)*std::pow(s, 3));
    if (s < 2.0) return norm*(-2.0*nd + 3.0*(nd + 1.0)*s - 1.50*(nd + 2.0)*s*s + 0.25*(nd + 3.0)*std::pow(s, 3));
    return 0.0;
  }
  double wzeta(double s) const {
    if (s < 1.0) return 1.4 - 2.0*s*s + 1.5*std::pow(s, 4) - 0.6*std::pow(s, 5);
    if (s < 2.0) return 1.6 - 4.0*s*s + 4.0*std::pow(s, 3) - 1.5*std::pow(s, 4) + 0.2*std::pow(s, 5);
    return 0.0;
  }
  double wgrav(double s) const {
    if (s < 1.0) return 1.333333333333333333333*s - 1.2*std::pow(s, 3) + 0.5*std::pow(s, 4);
    if (s < 2.0) return 2.6666666666666666667*s - 3.0*s*s + 1.2*std::pow(s, 3) - 0.166666666666666666667*std::pow(s, 4) - 0.06666666666666666667/(s*s);
    return 1.0/(s*s);
  }
  double wpot(double s) const {
    if (s < 1.0) return 1.4 - 0.666666666666666666666666*s*s + 0.3*std::pow(s, 4) - 0.1*std::pow(s, 5);
    if (s < 2.0) return -1.0/(15.0*s) + 1.6 - 1.33333333333333333333333333*s*s + std::pow(s, 3) - 0.3*std::pow(s, 4) + (1.0/30.0)*std::pow(s, 5);
    return 1.0/s;
  }
};
struct HostQuintic {       // QuinticKernel functions as the reference writes them (SmoothingKernel.h:281-408)
  int nd; double norm;
  explicit HostQuintic(int nd_) : nd(nd_) { norm = nd == 1 ? (1.0/120.0) : (nd == 2 ? GH_INVPI*(7.0/478.0) : GH_INVPI*(1/120.)); }
  double w0(double s) const {
    if (s < 1.0) return norm*(66.0 - 60.0*s*s + 30.0*std::pow(s,4) - 10.0*std::pow(s,5));
    if (s < 2.0) return norm*(51.0 + 75.0*s - 210.0*s*s + 150.0*std::pow(s,3) - 45.0*std::pow(s,4) + 5.0*std::pow(s,5));
    if (s < 3.0) return norm*(243.0 - 405*s + 270.0*s*s - 90.0*std::pow(s,3) + 15.0*std::pow(s,4) - std::pow(s,5));
    return 0.0;
  }
  double w1(double s) const {
    if (s < 1.0) return norm*(-120.0*s + 120.0*std::pow(s,3) - 50.0*std::pow(s,4));
    if (s < 2.0) return norm*(75.0 - 420.0*s + 450.0*s*s - 180.0*std::pow(s,3) + 25.0*std::pow(s,4));
    if (s < 3.0) return norm*(-405.0 + 540.0*s - 270.0*s*s + 60.0*std::pow(s,3) - 5.0*std::pow(s,4));
    return 0.0;
  }
  double womega(double s) const {
    if (s < 1.0) return norm*(-66.0*nd + 60.0*(nd + 2.0)*s*s - 30.0*(nd + 4.0)*std::pow(s,4) + 10.0*(nd + 5.0)*std::pow(s,5));
    if (s < 2.0) return norm*(-51.0*nd - 75.0*(nd + 1.0)*s + 210.0*(nd + 2.0)*s*s - 150.0*(nd + 3.0)*std::pow(s,3) + 45.0*(nd + 4.0)*std::pow(s,4) - 5.0*(nd + 5.0)*std::pow(s,5));
    if (s < 3.0) return norm*(-243.0*nd + 405.0*(nd + 1.0)*s - 270.0*(nd + 2.0)*s*s + 90.0*(nd + 3.0)*std::pow(s,3) - 15.0*(nd + 4.0)*std::pow(s,4) + (nd + 5.0)*std::pow(s,5));
    return 0.0;
  }
  double wzeta(double s) const {
    if (s < 1.0) return 33.0*s*s - 15.0*std::pow(s,4) + 5.0*std::pow(s,6) - 1.42857142857*std::pow(s,7) - 34.14285714;
    if (s < 2.0) return 25.5*s*s + 25.0*std::pow(s,3) - 52.5*std::pow(s,4) + 30.0*std::pow(s,5) - 7.5*std::pow(s,6) + 0.7142857143*std::pow(s,7) - 33.785714286;
    if (s < 3.0) return 121.5*s*s - 135.0*std::pow(s,3) + 67.5*std::pow(s,4) - 18.0*std::pow(s,5) + 2.5*std::pow(s,6) - 0.142857143*std::pow(s,7) - 52.07142857;
    return 0.0;
  }
  double wgrav(double s) const {
    if (s < 1.0) return (12.0/359.0)*(22.0*s - 12.0*std::pow(s,3) + (30.0/7.0)*std::pow(s,5) - (5.0/4.0)*std::pow(s,6));
    if (s < 2.0) return (12.0/359.0)*(17.0*s + (75.0/4.0)*s*s - 42.0*std::pow(s,3) + 25.0*std::pow(s,4) - (45.0/7.0)*std::pow(s,5) + (5.0/8.0)*std::pow(s,6) + (5.0/56.0)/(s*s));
    if (s < 3.0) return (12.0/359.0)*(81.0*s - (405.0/4.0)*s*s + 54.0*std::pow(s,3) - 15.0*std::pow(s,4) + (15.0/7.0)*std::pow(s,5) - (1.0/8.0)*std::pow(s,6) - (507.0/56.0)/(s*s));
    return 1.0/(s*s);
  }
  double wpot(double s) const {
    if (s < 1.0) return (12.0/359.0)*(-11.0*s*s + 3.0*std::pow(s,4) - (5.0/7.0)*std::pow(s,6) + (5.0/28.0)*std::pow(s,7) + (478.0/14.0));
    if (s < 2.0) return (12.0/359.0)*(-(17.0/2.0)*s*s - (25.0/4.0)*std::pow(s,3) + (21.0/2.0)*std::pow(s,4) - 5.0*std::pow(s,5) + (15.0/14.0)*std::pow(s,6) - (5.0/56.0)*std::pow(s,7) + (473.0/14.0) + (5.0/56.0)/s);
    if (s < 3.0) return (12.0/359.0)*(-(81.0/2.0)*s*s + (135.0/4.0)*std::pow(s,3) - (27.0/2.0)*std::pow(s,4) + 3.0*std::pow(s,5) - (5.0/14.0)*std::pow(s,6) + (1.0/56.0)*std::pow(s,7) + (729.0/14.0) - (507.0/56.0)/s);
    return 1.0/s;
  }
};
}
