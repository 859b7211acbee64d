// HipSphTree.h -- the reference-side binding of libgandalf_hip: a SphNeighbourSearch<ndim> whose every virtual forwards
// to the C ABI of include/gandalf_hip.h.  A GANDALF maintainer drops this file into src/Headers/, adds
//     else if (stringparams["neib_search"] == "hip") sphneib = new HipSphTree<ndim>(simparams, &simbox, sph);
// next to the kdtree / octtree branches of GradhSphSimulation<ndim>::ProcessSphParameters
// (src/GradhSph/GradhSphSimulation.cpp:232-246) and links libgandalf_hip.so.
//
// It is compiled in the build container against the reference's own headers and linked against libgandalf_hip.so by
// `make -f oracle/ref.mk hipshell` (compile + link check only; nothing of the reference travels anywhere).
//
// This shell keeps GANDALF's AoS particle array authoritative: every UpdateAll* call uploads what changed on the host
// and downloads what the call produced (the "seam replaced, integrator kept" mode of INTEGRATION.md).  The
// device-resident mode - gh_setup / gh_step replacing PostInitialConditionsSetup / MainLoop - is what
// gandalf_amd/host/SphSimulation.cpp does.
#ifndef HIP_SPH_TREE_H
#define HIP_SPH_TREE_H

#include <string>
#include <vector>
#include "Precision.h"
#include "Constants.h"
#include "Exception.h"
#include "Parameters.h"
#include "DomainBox.h"
#include "Hydrodynamics.h"
#include "Sph.h"
#include "Nbody.h"
#include "Ewald.h"
#include "NeighbourSearch.h"
#include "SphNeighbourSearch.h"
#include "Particle.h"
#include "gandalf_hip.h"

template <int ndim>
class HipSphTree : public SphNeighbourSearch<ndim>
{
  gh_ctx *ctx;
  Sph<ndim> *sph0;
  MAC_Type mac;
  bool uploaded;
  int Nlevels;                                      // > 1: block timesteps - the integrator's level / active state goes down, levelneib comes back
  int avisc;                                        // GH_AVISC_*: time-dependent viscosity needs alpha on the device
  std::vector<double> r, v, m, h, u, buf;           // SoA staging in the caller's (iorig) order

  void raise() { ExceptionHandler::getIstance().raise(std::string(gh_last_error(ctx))); }   // Exception.cpp:50-70
  void check(int rc) { if (rc < 0) raise(); }

  GradhSphParticle<ndim>* parts(Hydrodynamics<ndim> *hydro) {
    return static_cast<GradhSphParticle<ndim>*>(static_cast<Sph<ndim>*>(hydro)->GetSphParticleArray());
  }

  // GradhSphParticle[] -> device (positions, velocities, masses, h, u; everything else the device derives itself)
  void Upload(Hydrodynamics<ndim> *hydro) {
    GradhSphParticle<ndim> *p = parts(hydro);
    const int N = hydro->Nhydro;
    r.resize((size_t) N*ndim); v.resize((size_t) N*ndim); m.resize(N); h.resize(N); u.resize(N);
    for (int i=0; i<N; i++) {
      for (int k=0; k<ndim; k++) { r[(size_t) i*ndim+k] = p[i].r[k]; v[(size_t) i*ndim+k] = p[i].v[k]; }
      m[i] = p[i].m; h[i] = p[i].h; u[i] = p[i].u;
    }
    if (!uploaded || gh_num_particles(ctx) != N) {
      check(gh_upload_particles(ctx, N, &r[0], &v[0], &m[0], &h[0], &u[0]));
      uploaded = true;
    }
    else {
      check(gh_upload_field(ctx, GH_F_R, &r[0])); check(gh_upload_field(ctx, GH_F_V, &v[0]));
      check(gh_upload_field(ctx, GH_F_H, &h[0])); check(gh_upload_field(ctx, GH_F_U, &u[0]));
    }
    UploadLevels(hydro);
  }
  // what the block-timestep integrator decides on the host (SphLeapfrogKDK::AdvanceParticles / CheckTimesteps,
  // Simulation::ComputeBlockTimesteps): active flag, level, levelneib - the device masks its passes with them
  void UploadLevels(Hydrodynamics<ndim> *hydro) {
    if (Nlevels <= 1) return;
    GradhSphParticle<ndim> *p = parts(hydro);
    const int N = hydro->Nhydro;
    buf.resize(N);
    for (int i=0; i<N; i++) buf[i] = (double) ((p[i].flags.check(active) ? 1 : 0) | (p[i].flags.check(end_timestep) ? 2 : 0));
    check(gh_upload_field(ctx, GH_F_FLAGS, &buf[0]));
    for (int i=0; i<N; i++) buf[i] = (double) p[i].level;
    check(gh_upload_field(ctx, GH_F_LEVEL, &buf[0]));
    for (int i=0; i<N; i++) buf[i] = (double) p[i].levelneib;
    check(gh_upload_field(ctx, GH_F_LEVELNEIB, &buf[0]));
  }
  // what MainLoop refreshes on the host for ALL particles between the density and the force pass (pressure and sound
  // speed of the drifted inactive neighbours, SphSimulation.cpp:665-679), and alpha of the time-dependent viscosities
  void UploadThermal(Hydrodynamics<ndim> *hydro) {
    if (Nlevels <= 1 && avisc == GH_AVISC_MON97) return;
    GradhSphParticle<ndim> *p = parts(hydro);
    const int N = hydro->Nhydro;
    buf.resize(N);
    if (Nlevels > 1) {
      for (int i=0; i<N; i++) buf[i] = p[i].u;         check(gh_upload_field(ctx, GH_F_U, &buf[0]));
      for (int i=0; i<N; i++) buf[i] = p[i].sound;     check(gh_upload_field(ctx, GH_F_SOUND, &buf[0]));
      for (int i=0; i<N; i++) buf[i] = p[i].pressure;  check(gh_upload_field(ctx, GH_F_PRESSURE, &buf[0]));
    }
    if (avisc != GH_AVISC_MON97 && avisc != GH_AVISC_NONE) {
      for (int i=0; i<N; i++) buf[i] = p[i].alpha;     check(gh_upload_field(ctx, GH_F_ALPHA, &buf[0]));
    }
  }
  // one scalar / vector field back into the AoS array
  void DownloadScalar(Hydrodynamics<ndim> *hydro, int field, FLOAT GradhSphParticle<ndim>::*member) {
    GradhSphParticle<ndim> *p = parts(hydro);
    const int N = hydro->Nhydro;
    buf.resize(N);
    check(gh_download(ctx, field, &buf[0]));
    for (int i=0; i<N; i++) p[i].*member = buf[i];
  }
  void DownloadVector(Hydrodynamics<ndim> *hydro, int field, FLOAT (GradhSphParticle<ndim>::*member)[ndim]) {
    GradhSphParticle<ndim> *p = parts(hydro);
    const int N = hydro->Nhydro;
    buf.resize((size_t) N*ndim);
    check(gh_download(ctx, field, &buf[0]));
    for (int i=0; i<N; i++) for (int k=0; k<ndim; k++) (p[i].*member)[k] = buf[(size_t) i*ndim+k];
  }

 public:

  HipSphTree(Parameters *params, DomainBox<ndim> *box, Sph<ndim> *sph) : ctx(0), sph0(sph), mac(geometric), uploaded(false), Nlevels(1), avisc(0)
  {
    std::map<std::string, int> &ip = params->intparams;
    std::map<std::string, double> &fp = params->floatparams;
    std::map<std::string, std::string> &sp = params->stringparams;
    gh_config c = gh_config();                         // filled from the same parameter maps the kd-tree reads
    c.ndim = ndim;
    const bool tab = ip["tabulated_kernel"] == 1;
    c.kernel = sp["kernel"] == "quintic" ? (tab ? GH_KERNEL_QUINTIC_TAB : GH_KERNEL_QUINTIC) : (tab ? GH_KERNEL_M4_TAB : GH_KERNEL_M4);
    c.gas_eos = sp["gas_eos"] == "isothermal" ? GH_EOS_ISOTHERMAL : (sp["gas_eos"] == "barotropic" ? GH_EOS_BAROTROPIC : GH_EOS_ENERGY_EQN);
    c.avisc = sp["avisc"] == "none" ? GH_AVISC_NONE : GH_AVISC_MON97;
    if (c.avisc == GH_AVISC_MON97 && sp["time_dependent_avisc"] == "mm97") c.avisc = GH_AVISC_MON97MM97;
    if (c.avisc == GH_AVISC_MON97 && sp["time_dependent_avisc"] == "cd2010") c.avisc = GH_AVISC_MON97CD2010;
    c.acond = sp["acond"] == "wadsley2008" ? GH_ACOND_WADSLEY2008 : (sp["acond"] == "price2008" ? GH_ACOND_PRICE2008 : GH_ACOND_NONE);
    c.self_gravity = ip["self_gravity"]; c.hydro_forces = ip["hydro_forces"];
    c.multipole = sp["multipole"] == "quadrupole" ? GH_MULTIPOLE_QUADRUPOLE : (sp["multipole"] == "fast_monopole" ? GH_MULTIPOLE_FAST_MONOPOLE :
                  (sp["multipole"] == "fast_quadrupole" ? GH_MULTIPOLE_FAST_QUADRUPOLE : GH_MULTIPOLE_MONOPOLE));
    c.gravity_mac = sp["gravity_mac"] == "gadget2" ? GH_MAC_GADGET2 : (sp["gravity_mac"] == "eigenmac" ? GH_MAC_EIGENMAC : GH_MAC_GEOMETRIC);
    c.Nleafmax = ip["Nleafmax"];
    c.energy_integration = (sp["gas_eos"] == "energy_eqn" && sp["energy_integration"] != "none") ? 1 : 0;
    c.device = 0;
    for (int k=0; k<ndim; k++) {
      c.boundary_lhs[k] = box->boundary_lhs[k] == periodicBoundary ? GH_BOUNDARY_PERIODIC : (box->boundary_lhs[k] == mirrorBoundary ? GH_BOUNDARY_MIRROR : GH_BOUNDARY_OPEN);
      c.boundary_rhs[k] = box->boundary_rhs[k] == periodicBoundary ? GH_BOUNDARY_PERIODIC : (box->boundary_rhs[k] == mirrorBoundary ? GH_BOUNDARY_MIRROR : GH_BOUNDARY_OPEN);
      c.boxmin[k] = box->min[k]; c.boxmax[k] = box->max[k];
    }
    c.Nlevels = ip["Nlevels"]; c.level_diff_max = ip["level_diff_max"];
    c.ntreebuildstep = ip["ntreebuildstep"]; c.ntreestockstep = ip["ntreestockstep"];
    c.sph_single_timestep = ip["sph_single_timestep"];
    c.h_fac = fp["h_fac"]; c.h_converge = fp["h_converge"];
    c.alpha_visc = fp["alpha_visc"]; c.beta_visc = fp["beta_visc"]; c.alpha_visc_min = fp["alpha_visc_min"];
    c.gamma_eos = fp["gamma_eos"]; c.temp0 = fp["temp0"]; c.mu_bar = fp["mu_bar"]; c.rho_bary = fp["rho_bary"];
    c.thetamaxsqd = fp["thetamaxsqd"]; c.macerror = fp["macerror"];
    c.courant_mult = fp["courant_mult"]; c.accel_mult = fp["accel_mult"]; c.energy_mult = fp["energy_mult"];
    // sink particles: the shell keeps GANDALF's own Sinks object (it reaches the device through GetGatherNeighbourList below);
    // the device needs the parameters for ComputeH's rho_sink floor, in-sink branch and potential-minimum flag
    // (GradhSph.cpp:163-169, 270-280, 309-312) and builds its tree in the reference's particle order for them
    c.sink_particles = ip["sink_particles"]; c.create_sinks = c.sink_particles ? ip["create_sinks"] : 0;
    c.rho_sink = sph->rho_sink;                      // already in code units (SphSimulation.cpp:128-129)
    if (gh_create(&c, &ctx) != GH_OK) {
      std::string msg = ctx ? gh_last_error(ctx) : "gh_create failed";
      ExceptionHandler::getIstance().raise(msg);
    }
    mac = c.gravity_mac == GH_MAC_GADGET2 ? gadget2 : (c.gravity_mac == GH_MAC_EIGENMAC ? eigenmac : geometric);
    Nlevels = c.Nlevels; avisc = c.avisc;
  }
  virtual ~HipSphTree() { gh_destroy(ctx); }

  // ---- NeighbourSearch<ndim> --------------------------------------------------------------------------------------
  // HydroTree::BuildTree (HydroTree.cpp:310-372): the particles the integrator has just moved go down, then the
  // reference's own schedule - rebuild / re-stock / extrapolate - with the reference's own arguments
  virtual void BuildTree(const bool rebuild_tree, const int n, const int ntreebuildstep, const int ntreestockstep,
                         const FLOAT timestep, Hydrodynamics<ndim> *hydro)
  { Upload(hydro); check(gh_build_tree_scheduled(ctx, rebuild_tree ? 1 : 0, n, ntreebuildstep, ntreestockstep, timestep)); }
  // periodic / mirror images are made on the fly inside gh_update_density / gh_update_*_forces: no ghost particles, no ghost tree
  virtual void BuildGhostTree(const bool, const int, const int, const int, const FLOAT, Hydrodynamics<ndim> *) {}
  virtual void SearchBoundaryGhostParticles(FLOAT, const DomainBox<ndim> &, Hydrodynamics<ndim> *) {}
  // Tree::ComputeGatherNeighbourList(part, rp, rsearch, ...) (Tree.cpp:208-280): -1 = the caller's list is too short
  virtual int GetGatherNeighbourList(FLOAT *rp, FLOAT rsearch, Particle<ndim> *, int, int Nneibmax, int *neiblist)
  {
    double p[3] = {0.0, 0.0, 0.0};
    for (int k=0; k<ndim; k++) p[k] = rp[k];
    std::vector<int32_t> ids(Nneibmax > 0 ? Nneibmax : 1);
    const int n = gh_gather_neighbours_at(ctx, p, rsearch, &ids[0], Nneibmax);
    if (n < -1) raise();
    for (int j=0; j<n; j++) neiblist[j] = ids[j];
    return n;
  }
  // KDTree::UpdateActiveParticleCounters (KDTree.cpp:1217): MainLoop calls it when CheckTimesteps has woken particles for
  // another pass (SphSimulation.cpp:663) - the new active flags go down
  virtual void UpdateActiveParticleCounters(Hydrodynamics<ndim> *hydro) { UploadLevels(hydro); }
  virtual void UpdateAllStarGasForces(Hydrodynamics<ndim> *, Nbody<ndim> *nbody, DomainBox<ndim> &, Ewald<ndim> *)
  {
    // HydroTree::UpdateAllStarGasForces (HydroTree.cpp:552-657): gh_set_stars with the stars' r, m, h, then gh_star_gas_forces
    const int Ns = nbody->Nnbody;
    if (Ns <= 0) return;
    std::vector<double> sr((size_t) Ns*ndim), sm(Ns), sh(Ns), sa((size_t) Ns*ndim), sg(Ns);
    for (int i=0; i<Ns; i++) {
      for (int k=0; k<ndim; k++) sr[(size_t) i*ndim+k] = nbody->nbodydata[i]->r[k];
      sm[i] = nbody->nbodydata[i]->m; sh[i] = nbody->nbodydata[i]->h;
    }
    check(gh_set_stars(ctx, Ns, &sr[0], &sm[0], &sh[0], nbody->nbody_softening));
    check(gh_star_gas_forces(ctx, &sa[0], &sg[0]));
    for (int i=0; i<Ns; i++) {
      for (int k=0; k<ndim; k++) nbody->nbodydata[i]->a[k] += sa[(size_t) i*ndim+k];
      nbody->nbodydata[i]->gpot += sg[i];
    }
  }
  virtual double GetMaximumSmoothingLength() const
  {
    const int64_t N = gh_num_particles(ctx);
    std::vector<double> hh(N > 0 ? N : 1);
    if (N > 0 && gh_download(ctx, GH_F_H, &hh[0]) < 0) return 0.0;
    double hmax = 0.0;
    for (int64_t i=0; i<N; i++) if (hh[i] > hmax) hmax = hh[i];
    return hmax;
  }
  virtual TreeBase<ndim>* GetTree() const { return 0; }          // the tree lives on the device: gh_export_tree for inspection
  virtual TreeBase<ndim>* GetGhostTree() const { return 0; }
  virtual void SetTimingObject(CodeTiming*) {}                   // device phase timers: gh_get_timers
  virtual void ToggleNeighbourCheck(bool) {}
  virtual void UpdateTimestepsLimitsFromDistantParticles(Hydrodynamics<ndim>*, const bool) {}
  virtual MAC_Type GetOpeningCriterion() const { return mac; }
  virtual void SetOpeningCriterion(MAC_Type) {}                  // fixed at gh_create (gravity_mac); gh_setup does the geometric bootstrap itself

  // ---- SphNeighbourSearch<ndim> ---------------------------------------------------------------------------------
  virtual void UpdateAllSphProperties(Sph<ndim> *sph, Nbody<ndim> *)                                  // GradhSphTree.cpp:83-271
  {
    if (sph->sink_particles) {                                           // Particle::sinkid, set by Sinks::AccreteMassToSinks
      GradhSphParticle<ndim> *p = parts(sph);
      buf.resize(sph->Nhydro);
      for (int i = 0; i < sph->Nhydro; i++) buf[i] = (double) p[i].sinkid;
      check(gh_upload_field(ctx, GH_F_SINKID, &buf[0]));
    }
    check(gh_update_density(ctx, 0));
    if (sph->create_sinks == 1) {                                        // the potmin flag Sinks::SearchForNewSinkParticles reads
      GradhSphParticle<ndim> *p = parts(sph);
      buf.resize(sph->Nhydro);
      check(gh_download(ctx, GH_F_FLAGS, &buf[0]));
      for (int i = 0; i < sph->Nhydro; i++) { if ((int) buf[i] & 8) p[i].flags.set(potmin); else p[i].flags.unset(potmin); }
    }
    DownloadScalar(sph, GH_F_H, &GradhSphParticle<ndim>::h);             DownloadScalar(sph, GH_F_RHO, &GradhSphParticle<ndim>::rho);
    DownloadScalar(sph, GH_F_INVOMEGA, &GradhSphParticle<ndim>::invomega); DownloadScalar(sph, GH_F_ZETA, &GradhSphParticle<ndim>::zeta);
    DownloadScalar(sph, GH_F_HFACTOR, &GradhSphParticle<ndim>::hfactor);  DownloadScalar(sph, GH_F_HRANGESQD, &GradhSphParticle<ndim>::hrangesqd);
    DownloadScalar(sph, GH_F_SOUND, &GradhSphParticle<ndim>::sound);      DownloadScalar(sph, GH_F_PRESSURE, &GradhSphParticle<ndim>::pressure);
    DownloadScalar(sph, GH_F_U, &GradhSphParticle<ndim>::u);              DownloadScalar(sph, GH_F_DIV_V, &GradhSphParticle<ndim>::div_v);
  }
  void DownloadForces(Sph<ndim> *sph)
  {
    DownloadVector(sph, GH_F_A, &GradhSphParticle<ndim>::a);             DownloadVector(sph, GH_F_ATREE, &GradhSphParticle<ndim>::atree);
    DownloadScalar(sph, GH_F_DUDT, &GradhSphParticle<ndim>::dudt);        DownloadScalar(sph, GH_F_DIV_V, &GradhSphParticle<ndim>::div_v);
    DownloadScalar(sph, GH_F_GPOT, &GradhSphParticle<ndim>::gpot);        DownloadScalar(sph, GH_F_GPOT_HYDRO, &GradhSphParticle<ndim>::gpot_hydro);
    DownloadScalar(sph, GH_F_DALPHADT, &GradhSphParticle<ndim>::dalphadt);
    if (Nlevels > 1) {                                                    // levelneib: raised by active neighbours (GradhSph.cpp:455, 569)
      GradhSphParticle<ndim> *p = parts(sph);
      buf.resize(sph->Nhydro);
      check(gh_download(ctx, GH_F_LEVELNEIB, &buf[0]));
      for (int i = 0; i < sph->Nhydro; i++) p[i].levelneib = (int) buf[i];
    }
  }
  virtual void UpdateAllSphHydroForces(Sph<ndim> *sph, Nbody<ndim> *, DomainBox<ndim> &)              // GradhSphTree.cpp:280-435
  { UploadThermal(sph); check(gh_zero_accelerations(ctx)); check(gh_update_hydro_forces(ctx, 0)); DownloadForces(sph); }
  virtual void UpdateAllSphForces(Sph<ndim> *sph, Nbody<ndim> *, DomainBox<ndim> &, Ewald<ndim> *)    // GradhSphTree.cpp:444-657
  { UploadThermal(sph); check(gh_zero_accelerations(ctx)); check(gh_update_all_forces(ctx, 0)); DownloadForces(sph); }
};
#endif
