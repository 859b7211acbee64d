// cd2010.hip -- the Cullen & Dehnen (2010) viscosity switch (time_dependent_avisc = cd2010).
//
// Replaces Sph::ComputeCullenAndDehnenViscosity (reference src/Headers/Sph.h:364-456, called at the end of
// GradhSph::ComputeH, GradhSph.cpp:319-321; InvertMatrix InlineFuncs.h:577-606, CurlVelSqd Sph.h:344-358): integral-gradient
// estimates of grad v and grad a over the gather neighbours, d(div v)/dt, a Balsara-type limiter, and from them the local
// alpha and its decay rate.  It needs the converged h, so it is a pass of its own after the density kernel: same mapping
// (one wavefront per group of <= 64 particles, one lane per particle, streaming tree walk with broadcast LDS tiles), 24
// running sums per lane.
#include "gh_internal.hpp"
#include "sph_kernels.hpp"
#include "walk.hpp"

struct CdParams { Domain dom; const double *ktab; double alpha_visc, alpha_visc_min; int group0; };

enum { C_X = 0, C_Y, C_Z, C_M, C_VX, C_VY, C_VZ, C_AX, C_AY, C_AZ, C_NF };

template <int ND, int KT>
__global__ __launch_bounds__(64) void k_cullen_dehnen(DevicePtrs d, CdParams P, int *flags)
{
  typedef typename KSel<ND, KT>::type K;
  __shared__ WalkLDS<int> L;
  __shared__ double s_t[C_NF][64];
  const int lane = threadIdx.x;
  const int q = P.group0 + block_to_group(blockIdx.x, gridDim.x);
  const int gnode = (1 << d.lgroup) - 1 + q;
  const int gfirst = d.cfirst[gnode], gN = d.cN[gnode];
  if (gN == 0) return;
  const bool act = lane < gN && (!d.levels || ((int) d.f[D_FLAGS][gfirst + lane] & 1));
  if (!__any(act)) return;
  const int i = gfirst + (act ? lane : 0);
  double ri[3] = {0.0, 0.0, 0.0}, vi[3] = {0.0, 0.0, 0.0}, ai[3] = {0.0, 0.0, 0.0};
  for (int k = 0; k < ND; k++) { ri[k] = d.f[D_RX + k][i]; vi[k] = d.f[D_VX + k][i]; ai[k] = d.f[D_AX + k][i]; }
  const double h = d.f[D_H][i], rho = d.f[D_RHO][i];
  const double invh = 1.0/h;
  const double hfac = invh*d.f[D_HFACTOR][i]/rho;
  const double hr2 = K::kernrangesqd*h*h;
  double rr[3][3], dv[3][3], da[3][3];
  for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { rr[j][k] = 0.0; dv[j][k] = 0.0; da[j][k] = 0.0; }

  // candidate cells: every cell whose box comes within the group's largest kernel radius of the group's box
  const CellBox gb = d.cbox[gnode];
  const double hs = K::kernrange*wave_max(act ? h : 0.0);
  double lo[3], hi[3];
  for (int k = 0; k < 3; k++) { lo[k] = k < ND ? gb.bbmin[k] - hs : -1e300; hi[k] = k < ND ? gb.bbmax[k] + hs : 1e300; }
  const unsigned int codes = image_codes(P.dom, ND, lo, hi);
  auto cls = [&](int n, int code, bool &open, bool &emit, int &first, int &cnt) {
    const CellBox b = d.cbox[n];
    if (b.N == 0) return;
    double sg[3], sh[3];
    code_xform(P.dom, code, sg, sh);
    bool inside = true;
    for (int k = 0; k < ND; k++) {
      double bmin, bmax;
      image_interval(sg[k], sh[k], b.bbmin[k], b.bbmax[k], bmin, bmax);
      if (lo[k] > bmax || bmin > hi[k]) return;
      if (bmin < lo[k] || bmax > hi[k]) inside = false;
    }
    if (inside || n >= d.gtot - 1) { emit = true; first = b.first; cnt = b.N; }
    else open = true;
  };
  auto tile = [&](bool valid, int j, int code) {
    {
      double x = 1e30, y = 1e30, z = 1e30, m = 0.0, v[3] = {0.0, 0.0, 0.0}, a[3] = {0.0, 0.0, 0.0};
      if (valid) {
        double sg[3], sh[3];
        code_xform(P.dom, code, sg, sh);
        const double4 pm = d.posm[j];
        x = sg[0]*pm.x + sh[0]; y = sg[1]*pm.y + sh[1]; z = sg[2]*pm.z + sh[2]; m = pm.w;
        for (int k = 0; k < ND; k++) { v[k] = sg[k]*d.f[D_VX + k][j]; a[k] = sg[k]*d.f[D_AX + k][j]; }   // mirror images: v, a flip
      }
      s_t[C_X][lane] = x; s_t[C_Y][lane] = y; s_t[C_Z][lane] = z; s_t[C_M][lane] = m;
      s_t[C_VX][lane] = v[0]; s_t[C_VY][lane] = v[1]; s_t[C_VZ][lane] = v[2];
      s_t[C_AX][lane] = a[0]; s_t[C_AY][lane] = a[1]; s_t[C_AZ][lane] = a[2];
    }
    __syncthreads();
    if (act) {
      for (int c = 0; c < 64; c++) {
        double dr[3] = {0.0, 0.0, 0.0};
        dr[0] = s_t[C_X][c] - ri[0];
        if (ND > 1) dr[1] = s_t[C_Y][c] - ri[1];
        if (ND > 2) dr[2] = s_t[C_Z][c] - ri[2];
        const double r2 = dr[0]*dr[0] + dr[1]*dr[1] + dr[2]*dr[2];
        if (r2 < hr2) {                                          // w1 vanishes outside the kernel
          const double w = s_t[C_M][c]*hfac*K::t_w1(invh*sqrt(r2), P.ktab);
          for (int jj = 0; jj < ND; jj++) {
            const double wd = w*dr[jj];
            for (int k = 0; k < ND; k++) {
              rr[jj][k] += wd*dr[k];
              dv[jj][k] += wd*(s_t[C_VX + k][c] - vi[k]);
              da[jj][k] += wd*(s_t[C_AX + k][c] - ai[k]);
            }
          }
        }
      }
    }
    __syncthreads();
  };
  walk_dfs_stream(d, L, codes, cls, tile, flags);

  if (!act) return;
  double T[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  const double (*A)[3] = rr;
  if (ND == 1) T[0][0] = 1.0/A[0][0];
  else if (ND == 2) {
    const double invdet = 1.0/(A[0][0]*A[1][1] - A[0][1]*A[1][0]);
    T[0][0] = invdet*A[1][1]; T[0][1] = -invdet*A[0][1]; T[1][0] = -invdet*A[1][0]; T[1][1] = invdet*A[0][0];
  }
  else {
    const double invdet = 1.0/(A[0][0]*(A[1][1]*A[2][2] - A[2][1]*A[1][2]) - A[0][1]*(A[1][0]*A[2][2] - A[1][2]*A[2][0]) +
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
  double modR = 0.0, modT = 0.0;
  for (int j = 0; j < ND; j++) for (int k = 0; k < ND; k++) { modR += rr[j][k]*rr[j][k]; modT += T[j][k]*T[j][k]; }
  const double sqd_condition_number = modR*modT/(double) (ND*ND);
  double alpha_loc = 0.0;
  if (sqd_condition_number > 1e4) alpha_loc = P.alpha_visc;      // bad gradients
  else {
    double dvdx[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}}, dadx[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    for (int a_ = 0; a_ < ND; a_++) for (int b_ = 0; b_ < ND; b_++) for (int k = 0; k < ND; k++) {
      dvdx[a_][b_] += T[b_][k]*dv[k][a_];
      dadx[a_][b_] += T[b_][k]*da[k][a_];
    }
    double ddivdt = 0.0, divv2 = 0.0;
    for (int a_ = 0; a_ < ND; a_++) {
      ddivdt += dadx[a_][a_];
      for (int b_ = 0; b_ < ND; b_++) ddivdt -= dvdx[a_][b_]*dvdx[b_][a_];
      divv2 += dvdx[a_][a_];
    }
    divv2 *= divv2;
    double curlv2 = 0.0;
    if (ND == 2) { const double c = dvdx[1][0] - dvdx[0][1]; curlv2 = c*c; }
    else if (ND == 3) {
      const double c0 = dvdx[1][2] - dvdx[2][1], c1 = dvdx[2][0] - dvdx[0][2], c2 = dvdx[0][1] - dvdx[1][0];
      curlv2 = c0*c0 + c1*c1 + c2*c2;
    }
    double f_balsara = 1.0;
    if (curlv2 > 0.0) f_balsara = divv2/(divv2 + curlv2);
    if (ddivdt < 0.0) {
      const double sound = d.f[D_SOUND][i];
      alpha_loc = (10.0*h*h/(sound*sound))*f_balsara*(-ddivdt);
      alpha_loc = fmin(alpha_loc, P.alpha_visc);
    }
  }
  double alpha = d.f[D_ALPHA][i];
  if (alpha_loc > alpha) alpha = alpha_loc;
  d.f[D_ALPHA][i] = alpha;
  d.f[D_DALPHADT][i] = 0.1*d.f[D_SOUND][i]*(fmax(P.alpha_visc_min, alpha_loc) - alpha)*invh;
}

int gh_cullen_dehnen_impl(gh_ctx *ctx)
{
  DevicePtrs d = gh_dev(ctx);
  CdParams P;
  gh_fill_domain(ctx, P.dom);
  P.ktab = ctx->ktab; P.alpha_visc = ctx->cfg.alpha_visc; P.alpha_visc_min = ctx->cfg.alpha_visc_min;
  int g0, g1;
  gh_shard_groups(ctx, ctx->rank, g0, g1);
  P.group0 = g0;
  const int nblocks = g1 - g0;
  if (nblocks <= 0) return GH_OK;
#define LAUNCH(ND_, KT_) hipLaunchKernelGGL((k_cullen_dehnen<ND_, KT_>), dim3(nblocks), dim3(64), 0, ctx->stream, d, P, ctx->d_flags);
  GH_DISPATCH(ctx, LAUNCH)
#undef LAUNCH
  GH_CHECK(ctx, hipGetLastError());
  return GH_OK;
}
