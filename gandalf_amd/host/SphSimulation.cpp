#include "SphSimulation.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>

static const double pi = 3.14159265358979;        // reference Constants.h:61
static const double twopi = 6.28318530717959;     // Constants.h:62
static const double onethird = 0.33333333333333333333333;
static const double big_number = 9.9e20;

static void check(gh_ctx *ctx, int rc, const char *where)
{
  if (rc) throw GandalfError(std::string(where) + ": " + (ctx ? gh_last_error(ctx) : "no context"));
}

// ---------------------------------------------------------------------------------------------
void Sph::AllocateMemory(int N)
{
  part.N = N; part.ndim = ndim;
  part.r.assign((size_t) N*ndim, 0.0); part.v.assign((size_t) N*ndim, 0.0);
  part.m.assign(N, 0.0); part.h.assign(N, 0.0); part.u.assign(N, 0.0);
}

// One global h from the volume of the bounding box (reference Sph.cpp:76-119; the float pow/sqrt calls
// of the reference are kept because they set the number that seeds the first density pass)
void Sph::InitialSmoothingLengthGuess()
{
  double rmin[3], rmax[3];
  for (int k = 0; k < ndim; k++) { rmin[k] = big_number; rmax[k] = -big_number; }
  for (int i = 0; i < part.N; i++)
    for (int k = 0; k < ndim; k++) {
      rmin[k] = std::min(rmin[k], part.r[(size_t) i*ndim + k]);
      rmax[k] = std::max(rmax[k], part.r[(size_t) i*ndim + k]);
    }
  double h_guess, volume;
  if (ndim == 1) {
    Ngather = (int) (2.0*kernrange*h_fac);
    volume = rmax[0] - rmin[0];
    h_guess = (volume*(double) Ngather)/(4.0*(double) part.N);
  }
  else if (ndim == 2) {
    Ngather = (int) (pi*pow(kernrange*h_fac, 2));
    volume = (rmax[0] - rmin[0])*(rmax[1] - rmin[1]);
    h_guess = sqrtf((volume*(double) Ngather)/(4.0*(double) part.N));
  }
  else {
    Ngather = (int) (4.0*pi*pow(kernrange*h_fac, 3)/3.0);
    volume = (rmax[0] - rmin[0])*(rmax[1] - rmin[1])*(rmax[2] - rmin[2]);
    h_guess = powf((3.0*volume*(double) Ngather)/(32.0*pi*(double) part.N), onethird);
  }
  for (int i = 0; i < part.N; i++) part.h[i] = h_guess;
}

// ---------------------------------------------------------------------------------------------
void SphNeighbourSearch::BuildTree() { check(ctx, gh_build_tree(ctx), "BuildTree"); }
void SphNeighbourSearch::UpdateAllSphProperties(gh_stats *st) { check(ctx, gh_update_density(ctx, st), "UpdateAllSphProperties"); }
void SphNeighbourSearch::UpdateAllSphHydroForces(gh_stats *st) { check(ctx, gh_update_hydro_forces(ctx, st), "UpdateAllSphHydroForces"); }
void SphNeighbourSearch::UpdateAllSphForces(gh_stats *st) { check(ctx, gh_update_all_forces(ctx, st), "UpdateAllSphForces"); }

// ---------------------------------------------------------------------------------------------
SphSimulation *SphSimulation::SimulationFactory(int ndim, const std::string &simtype, Parameters *params)
{
  if (ndim < 1 || ndim > 3) throw GandalfError("Invalid simulation dimensionality chosen : ndim = " + std::to_string(ndim));
  if (simtype != "sph" && simtype != "gradhsph")
    throw GandalfError("Only sim = sph | gradhsph is built for the HIP hot path (got " + simtype + ")");
  return new SphSimulation(ndim, params);
}

SphSimulation::SphSimulation(int ndim, Parameters *params) : ndim(ndim), simparams(params) { memset(&cfg, 0, sizeof(cfg)); }

SphSimulation::~SphSimulation()
{
  if (ctx) gh_destroy(ctx);
  delete sph; delete sphneib; delete randnumb;
}

static int enum_of(const std::string &v, const char *const *names, int n, const std::string &key)
{
  for (int i = 0; i < n; i++) if (v == names[i]) return i;
  throw GandalfError("Unrecognised parameter : " + key + " = " + v);
}

void SphSimulation::ProcessParameters()
{
  auto &ip = simparams->intparams; auto &fp = simparams->floatparams; auto &sp = simparams->stringparams;
  if (sp["neib_search"] != "kdtree") throw GandalfError("Unrecognised parameter : neib_search = " + sp["neib_search"]);
  if (sp["sph_integration"] != "lfkdk") throw GandalfError("Unrecognised parameter : sph_integration = " + sp["sph_integration"]);
  static const char *kern[] = {"m4", "quintic"}, *eos[] = {"energy_eqn", "isothermal", "barotropic"};
  static const char *av[] = {"none", "mon97"}, *ac[] = {"none", "wadsley2008", "price2008"};
  static const char *mp[] = {"monopole", "quadrupole", "fast_monopole", "fast_quadrupole"}, *mac[] = {"geometric", "gadget2", "eigenmac"}, *bd[] = {"open", "periodic", "mirror"};
  cfg.ndim = ndim;
  cfg.kernel = enum_of(sp["kernel"], kern, 2, "kernel");
  if (ip["tabulated_kernel"] != 0)                                       // TabulatedKernel<ndim>(kernel), Sph constructor
    cfg.kernel = cfg.kernel == GH_KERNEL_QUINTIC ? GH_KERNEL_QUINTIC_TAB : GH_KERNEL_M4_TAB;
  cfg.gas_eos = enum_of(sp["gas_eos"], eos, 3, "gas_eos");
  cfg.avisc = enum_of(sp["avisc"], av, 2, "avisc");
  if (sp["time_dependent_avisc"] == "mm97" && cfg.avisc == GH_AVISC_MON97) cfg.avisc = GH_AVISC_MON97MM97;   // GradhSphSimulation.cpp:76-79
  else if (sp["time_dependent_avisc"] == "cd2010" && cfg.avisc == GH_AVISC_MON97) cfg.avisc = GH_AVISC_MON97CD2010;       // GradhSphSimulation.cpp:80-83
  else if (sp["time_dependent_avisc"] != "none")
    throw GandalfError("Unrecognised parameter : time_dependent_avisc = " + sp["time_dependent_avisc"] + " (built: none, mm97, cd2010)");
  cfg.alpha_visc_min = fp["alpha_visc_min"];
  cfg.acond = enum_of(sp["acond"], ac, 3, "acond");
  cfg.self_gravity = ip["self_gravity"];
  cfg.hydro_forces = ip["hydro_forces"];
  cfg.multipole = enum_of(sp["multipole"], mp, 4, "multipole");
  cfg.gravity_mac = enum_of(sp["gravity_mac"], mac, 3, "gravity_mac");
  cfg.macerror = fp["macerror"];
  cfg.Nleafmax = ip["Nleafmax"];
  cfg.Nlevels = ip["Nlevels"]; cfg.level_diff_max = ip["level_diff_max"];   // Simulation.cpp:1209-1211
  cfg.ntreebuildstep = ip["ntreebuildstep"]; cfg.ntreestockstep = ip["ntreestockstep"];
  cfg.sph_single_timestep = ip["sph_single_timestep"];
  cfg.device = ip["device"];
  cfg.energy_integration = sp["gas_eos"] == "energy_eqn" ? 1 : 0;     // GradhSphSimulation.cpp:114-122
  for (int k = 0; k < 3; k++) {
    const std::string idx = "[" + std::to_string(k) + "]";
    cfg.boundary_lhs[k] = enum_of(sp["boundary_lhs" + idx], bd, 3, "boundary_lhs" + idx);
    cfg.boundary_rhs[k] = enum_of(sp["boundary_rhs" + idx], bd, 3, "boundary_rhs" + idx);
    cfg.boxmin[k] = fp["boxmin" + idx]; cfg.boxmax[k] = fp["boxmax" + idx];
  }
  cfg.h_fac = fp["h_fac"]; cfg.h_converge = fp["h_converge"];
  cfg.alpha_visc = fp["alpha_visc"]; cfg.beta_visc = fp["beta_visc"];
  cfg.gamma_eos = fp["gamma_eos"]; cfg.temp0 = fp["temp0"]; cfg.mu_bar = fp["mu_bar"]; cfg.rho_bary = fp["rho_bary"];
  cfg.thetamaxsqd = fp["thetamaxsqd"];
  cfg.courant_mult = fp["courant_mult"]; cfg.accel_mult = fp["accel_mult"]; cfg.energy_mult = fp["energy_mult"];
  tend = fp["tend"]; Nstepsmax = ip["Nstepsmax"];
  delete sph; delete randnumb;
  sph = new Sph(ndim, cfg.h_fac, (cfg.kernel == GH_KERNEL_QUINTIC || cfg.kernel == GH_KERNEL_QUINTIC_TAB) ? 3.0 : 2.0);
  randnumb = new XorshiftRand((uint64_t) ip["randseed"]);
}

// the device context is created on first use, so that parameter handling and IC generation work
// (and are tested) on a machine without a GPU; the hot path itself has no CPU fallback
void SphSimulation::EnsureContext()
{
  if (ctx) return;
  int rc = gh_create(&cfg, &ctx);
  if (rc) { std::string m = ctx ? gh_last_error(ctx) : "gh_create failed"; if (ctx) gh_destroy(ctx); ctx = nullptr; throw GandalfError(m); }
  sphneib = new SphNeighbourSearch(ctx);
  // multi-GPU run (the reference's MpiControl): rank / size and the collectives were registered with InitComm
  if (comm_nranks > 1 && gh_comm_init(ctx, comm_rank, comm_nranks, &comm_ops)) throw GandalfError(gh_last_error(ctx));
}

// the reference's mpicontrol->InitialiseMpiProcess (MpiControl.cpp:94-150): one process per GPU, rank r owns the
// r-th top-level cell of the global KD-tree; the collectives come from the caller (gandalf_hip.h: gh_comm_ops)
void SphSimulation::InitComm(int rank, int nranks, const gh_comm_ops *ops)
{
  if (ctx) throw GandalfError("InitComm: call before the device context exists (before SetupSimulation)");
  comm_rank = rank; comm_nranks = nranks;
  if (ops) comm_ops = *ops;
}

// ---------------------------------------------------------------------------------------------
// Initial conditions: ic = box (UniformIc.cpp:50-131 with particle_distribution = random,
// Ic::AddRandomBox Ic.cpp:440-455) and ic = plummer (PlummerSphereIc.cpp:53-182, gas only).
void SphSimulation::GenerateIC()
{
  auto &ip = simparams->intparams; auto &fp = simparams->floatparams; auto &sp = simparams->stringparams;
  const std::string ic = sp["ic"];
  if (ic == "file") {
    // SimulationIC.hpp:88-92: particles from a snapshot (in_file, in_file_form = column | su); the smoothing lengths are
    // recomputed from scratch by the setup (initial_h_provided = false)
    Snapshot snap;
    ReadSnapshotFile(sp["in_file"], sp["in_file_form"], snap);
    if (snap.ndim != ndim) throw GandalfError("Incorrect no. of dimensions in file");
    sph->AllocateMemory(std::max(snap.N, 1));
    HydroParticles &q = sph->part;
    q.r = snap.r; q.v = snap.v; q.m = snap.m; q.h = snap.h; q.u = snap.u;
    t = snap.t;
    initial_h_provided = false;
    return;
  }
  const int N = ip["Nhydro"];
  if (N <= 0 && ic != "shocktube") throw GandalfError("Nhydro must be positive");
  sph->AllocateMemory(std::max(N, 1));
  HydroParticles &p = sph->part;
  if (ic == "box") {
    if (ip["dimensionless"] == 0) throw GandalfError("dimensionless units required");
    if (sp["particle_distribution"] != "random") throw GandalfError("Invalid particle distribution option");
    double volume = 1.0;
    for (int k = 0; k < ndim; k++) volume *= cfg.boxmax[k] - cfg.boxmin[k];
    for (int i = 0; i < N; i++)
      for (int k = 0; k < ndim; k++)
        p.r[(size_t) i*ndim + k] = cfg.boxmin[k] + (cfg.boxmax[k] - cfg.boxmin[k])*randnumb->floatrand();
    const double invndim = 1.0/ndim;
    for (int i = 0; i < N; i++) {
      p.m[i] = volume/(double) N;
      p.h[i] = cfg.h_fac*pow(volume/(double) N, invndim);
      p.u[i] = 1.5;
    }
    initial_h_provided = true;
  }
  else if (ic == "plummer") {
    if (ndim != 3) throw GandalfError("plummer needs ndim = 3");
    if (ip["Nstar"] != 0) throw GandalfError("star particles are not built (Nstar must be 0)");
    const double gamma_eos = fp["gamma_eos"];
    double gasfrac = fp["gasfrac"], starfrac = fp["starfrac"];
    const double mplummer = fp["mplummer"], rplummer = fp["rplummer"], radius = fp["radius"];
    const double raux = gasfrac + starfrac;
    gasfrac /= raux; starfrac /= raux;
    for (int j = 0; j < N; j++) {
      double x1, x2, x3, x4, x5, rad;
      bool flag;
      do {
        flag = false;
        x1 = randnumb->floatrand(); x2 = randnumb->floatrand(); x3 = randnumb->floatrand();
        if (x1 == 0.0 && x2 == 0.0 && x3 == 0.0) flag = true;
        rad = 1.0/sqrt(pow(x1, -2.0/3.0) - 1.0);
        if (rad > radius/rplummer) flag = true;
      } while (flag);
      const double z = (1.0 - 2.0*x2)*rad;
      p.r[(size_t) j*3 + 2] = z;
      p.r[(size_t) j*3 + 0] = sqrt(rad*rad - z*z)*cos(twopi*x3);
      p.r[(size_t) j*3 + 1] = sqrt(rad*rad - z*z)*sin(twopi*x3);
      p.m[j] = gasfrac/(double) N;
      // the velocity draws are made for gas particles too and thrown away (PlummerSphereIc.cpp:131-143)
      double t1, t2;
      do {
        x4 = randnumb->floatrand(); x5 = randnumb->floatrand();
        t1 = 0.1*x5;
        t2 = x4*x4*pow(1.0 - x4*x4, 3.5);
      } while (t1 > t2);
      (void) randnumb->floatrand(); (void) randnumb->floatrand();
      const double sound = sqrt(0.16666666666666666/sqrt(1.0 + rad*rad));
      p.u[j] = sound*sound/(gamma_eos - 1.0);
    }
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 3; k++) p.r[(size_t) i*3 + k] = p.r[(size_t) i*3 + k]*rplummer;
      p.m[i] = p.m[i]*mplummer;
      p.u[i] = p.u[i]*(mplummer/rplummer);
    }
    initial_h_provided = false;
  }
  else if (ic == "shocktube") {
    // ShocktubeIc::Generate (ShocktubeIc.cpp:55-200), 1-D: two cubic lattices (Ic::AddCubicLattice,
    // Ic.cpp:629-664) left and right of x = 0; u from the pressure (EOS::InternalEnergyFromPressure, EOS.h:164)
    if (ndim != 1) throw GandalfError("shocktube is built for ndim = 1");
    if (ip["dimensionless"] == 0) throw GandalfError("dimensionless units required");
    const int Nbox1 = ip["Nlattice1[0]"], Nbox2 = ip["Nlattice2[0]"];
    const double gammaone = fp["gamma_eos"] - 1.0;
    const double rho[2] = {fp["rhofluid1"], fp["rhofluid2"]}, press[2] = {fp["press1"], fp["press2"]};
    const double vf[2] = {fp["vfluid1[0]"], fp["vfluid2[0]"]};
    const double bmin[2] = {cfg.boxmin[0], 0.0}, bmax[2] = {0.0, cfg.boxmax[0]};
    const int Nb[2] = {Nbox1, Nbox2};
    sph->AllocateMemory(Nbox1 + Nbox2);
    HydroParticles &q = sph->part;
    int off = 0;
    for (int side = 0; side < 2; side++) {
      const double spacing = (bmax[side] - bmin[side])/(double) Nb[side];
      const double volume = bmax[side] - bmin[side];
      const double u = sp["gas_eos"] == "isothermal" ? fp["temp0"]/gammaone/fp["mu_bar"] : press[side]/(rho[side]*gammaone);
      for (int ii = 0; ii < Nb[side]; ii++) {
        const int i = off + ii;
        q.r[i] = bmin[side] + ((double) ii + 0.5)*spacing;
        q.v[i] = vf[side];
        q.m[i] = rho[side]*volume/(double) Nb[side];
        q.h[i] = cfg.h_fac*pow(q.m[i]/rho[side], 1.0);
        q.u[i] = u;
      }
      off += Nb[side];
    }
    initial_h_provided = true;
  }
  else throw GandalfError("Unrecognised parameter : ic = " + ic);
}

void SphSimulation::SetComFrame()
{
  HydroParticles &p = sph->part;
  double mtot = 0.0, rcom[3] = {0, 0, 0}, vcom[3] = {0, 0, 0};
  for (int i = 0; i < p.N; i++) {
    mtot += p.m[i];
    for (int k = 0; k < ndim; k++) { rcom[k] += p.m[i]*p.r[(size_t) i*ndim + k]; vcom[k] += p.m[i]*p.v[(size_t) i*ndim + k]; }
  }
  for (int k = 0; k < ndim; k++) { rcom[k] /= mtot; vcom[k] /= mtot; }
  for (int i = 0; i < p.N; i++)
    for (int k = 0; k < ndim; k++) { p.r[(size_t) i*ndim + k] -= rcom[k]; p.v[(size_t) i*ndim + k] -= vcom[k]; }
}

void SphSimulation::PostInitialConditionsSetup()
{
  HydroParticles &p = sph->part;
  if (!initial_h_provided) sph->InitialSmoothingLengthGuess();
  EnsureContext();
  check(ctx, gh_upload_particles(ctx, p.N, p.r.data(), p.v.data(), p.m.data(), p.h.data(), p.u.data()), "upload");
  // a run that starts from a snapshot keeps the snapshot's time (ReadColumnSnapshotFile / ReadSerenUnformSnapshotFile set
  // Simulation::t and nothing in the setup resets it); generated ICs start at t = 0
  if (t != 0.0) check(ctx, gh_set_time(ctx, t, 0.0), "set_time");
  check(ctx, gh_setup(ctx, initial_h_provided ? 1 : 0, &timestep), "PostInitialConditionsSetup");
  Nsteps = 0;
  setup = true;
}

void SphSimulation::SetupSimulation()
{
  if (setup) throw GandalfError("This simulation has been already set up");
  ProcessParameters();
  GenerateIC();
  if (simparams->intparams["com_frame"] == 1) SetComFrame();
  PostInitialConditionsSetup();
}

void SphSimulation::MainLoop(int nsteps)
{
  check(ctx, gh_step(ctx, nsteps, &t, &timestep), "MainLoop");
  Nsteps += nsteps;
}

void SphSimulation::Run(int Nadvance)
{
  const int Ntarget = Nadvance < 0 ? Nstepsmax : Nsteps + Nadvance;
  while (t < tend && Nsteps < Ntarget) MainLoop(1);
}

// SimulationBase::WriteSnapshotFile (SimulationIO.hpp:96-125): current device state in the caller's particle order
void SphSimulation::WriteSnapshotFile(const std::string &filename, const std::string &fileform)
{
  Snapshot s;
  s.ndim = ndim; s.N = sph->part.N; s.t = t; s.Nsteps = Nsteps; s.h_fac = cfg.h_fac;
  Download(GH_F_R, s.r); Download(GH_F_V, s.v); Download(GH_F_M, s.m); Download(GH_F_H, s.h);
  Download(GH_F_RHO, s.rho); Download(GH_F_U, s.u);
  s.iorig.resize(s.N);
  for (int i = 0; i < s.N; i++) s.iorig[i] = i;
  for (int i = 0; i < s.N; i++) s.mmean += s.m[i];                        // sph->mmean, SphSimulation.cpp:260-262
  if (s.N > 0) s.mmean /= (double) s.N;
  ::WriteSnapshotFile(filename, fileform, s);
}

void SphSimulation::Download(int field, std::vector<double> &out)
{
  const bool vec = field <= GH_F_A0;
  out.resize((size_t) sph->part.N*(vec ? ndim : 1));
  check(ctx, gh_download(ctx, field, out.data()), "Download");
}
