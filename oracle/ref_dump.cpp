// oracle/ref_dump.cpp -- TEST INFRASTRUCTURE, not product code.
//
// Driver that is linked against the *reference's own objects* (oracle/ref.mk builds them
// from /root/reference/src where they lie) and dumps the state the parity tests are pinned
// to.  It calls only public entry points of the reference, in the order its own main()
// (reference src/Common/gandalf.cpp:40-190) and SphSimulation::MainLoop
// (src/Hydrodynamics/SphSimulation.cpp:574-880) call them; it contains no algorithmic
// code of its own.
//
//   ref_dump passes <params.dat> <out_prefix>
//       SetupSimulation(), dump "setup"; then one extra density pass and one extra force
//       pass in MainLoop order (BuildTree -> ghosts -> UpdateAllSphProperties ->
//       ZeroAccelerations -> UpdateAllSph(Hydro)Forces), dumping after each, so that each
//       pass has an (input state, output state) pair.
//   ref_dump steps  <params.dat> <out_prefix> <nsteps>
//       SetupSimulation(), dump "setup"; nsteps x MainLoop(); dump "final".
//   ref_dump run    <params.dat> <out_prefix> <nsteps>
//       SetupSimulation() (REF_RESTART=1: as a restart) and SimulationBase::Run(nsteps) with its regular snapshots.
//   ref_dump snap   <params.dat> <out_prefix> [nsteps]
//       SetupSimulation(), nsteps x MainLoop(), then the reference's column, SEREN-unformatted and SEREN-formatted snapshot writers.
//   ref_dump time   <params.dat> <nsteps> [warmup]
//       SetupSimulation(); warmup x MainLoop(); time nsteps x MainLoop(); prints one JSON line
//       (the CPU baseline of bench.py, kind "reference").
//   ref_dump nbody  <N> <softening 0|1> <nsteps> <out_prefix>
//       star cluster through the reference's own NbodyLeapfrogKDK<3,M4Kernel>: direct-sum forces
//       (Nbody::CalculateDirectGravForces / CalculateDirectSmoothedGravForces), then nsteps of
//       AdvanceParticles -> forces -> CorrectionTerms -> Timestep -> EndTimestep in the order of
//       NbodySimulation::MainLoop (NbodySimulation.cpp:258-405).  Star ICs come from a 64-bit LCG
//       written out below (inputs are part of the dump).  The only glue restated here is the
//       min-reduction of Simulation::ComputeGlobalTimestep (Simulation.cpp:1720-1745), which needs a
//       full Simulation object in the reference.
//
// Dump format ("GDMP1"): records of
//   u32 name_len, name bytes, u8 dtype ('d' f64, 'i' i32), u32 ndim, u64 dims[ndim], raw data.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>
#include <iostream>
#include <fstream>
#include <sys/time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "Exception.h"
#include "Parameters.h"
#include "Simulation.h"
#include "Sph.h"
#include "Nbody.h"
#include "Particle.h"
#include "KDTree.h"
#include "NeighbourSearch.h"
#include "SphNeighbourSearch.h"
#include "Sinks.h"
#ifdef REF_HIPSHELL
// the reference-side binding of libgandalf_hip (include/reference_shell/HipSphTree.h): ref_hipshell = this driver with
// the reference's own GradhSphSimulation running on a HipSphTree instead of its KD-tree (modes prefixed "hip")
#include "HipSphTree.h"
#endif

using namespace std;

static double wall()
{
  struct timeval tv;
  gettimeofday(&tv, 0);
  return tv.tv_sec + 1e-6*tv.tv_usec;
}

struct Dump {
  FILE *f;
  explicit Dump(const string &path) { f = fopen(path.c_str(), "wb"); if (!f) { perror(path.c_str()); exit(2);} fwrite("GDMP1\0\0\0", 1, 8, f); }
  ~Dump() { fclose(f); }
  void head(const char *name, char dt, const vector<uint64_t> &dims) {
    uint32_t n = strlen(name); fwrite(&n, 4, 1, f); fwrite(name, 1, n, f);
    fwrite(&dt, 1, 1, f);
    uint32_t nd = dims.size(); fwrite(&nd, 4, 1, f);
    for (size_t i = 0; i < dims.size(); i++) fwrite(&dims[i], 8, 1, f);
  }
  void d(const char *name, const vector<double> &v, uint64_t cols = 0) {
    vector<uint64_t> dims;
    if (cols) { dims.push_back(v.size()/cols); dims.push_back(cols); } else dims.push_back(v.size());
    head(name, 'd', dims); fwrite(v.data(), 8, v.size(), f);
  }
  void i(const char *name, const vector<int> &v, uint64_t cols = 0) {
    vector<uint64_t> dims;
    if (cols) { dims.push_back(v.size()/cols); dims.push_back(cols); } else dims.push_back(v.size());
    head(name, 'i', dims); fwrite(v.data(), 4, v.size(), f);
  }
};

#define PSCAL(field)  { vector<double> v(N); for (int i=0;i<N;i++) v[i]=p[i].field; out.d(#field, v); }
#define PINT(field)   { vector<int> v(N); for (int i=0;i<N;i++) v[i]=p[i].field; out.i(#field, v); }
#define PVEC(field)   { vector<double> v((size_t)N*ndim); for (int i=0;i<N;i++) for (int k=0;k<ndim;k++) v[(size_t)i*ndim+k]=p[i].field[k]; out.d(#field, v, ndim); }

template <int ndim>
static void dump_particles(Dump &out, Simulation<ndim> *sim)
{
  Sph<ndim> *sph = static_cast<Sph<ndim>*>(sim->hydro);
  GradhSphParticle<ndim> *p = static_cast<GradhSphParticle<ndim>*>(sph->GetSphParticleArray());
  const int N = sph->Nhydro;
  { vector<int> v(1, N); out.i("Nhydro", v); }
  { vector<int> v(1, sph->Ntot); out.i("Ntot", v); }
  { vector<int> v(1, ndim); out.i("ndim", v); }
  { vector<int> v(N); for (int i=0;i<N;i++) v[i]=(int)p[i].flags.get(); out.i("flags", v); }
  PINT(ptype) PINT(iorig) PINT(levelneib) PINT(nstep) PINT(nlast) PINT(level)
  PVEC(r) PVEC(v) PVEC(a) PVEC(atree) PVEC(r0) PVEC(v0) PVEC(a0)
  PSCAL(m) PSCAL(h) PSCAL(hrangesqd) PSCAL(hfactor) PSCAL(sound) PSCAL(rho) PSCAL(pressure)
  PSCAL(u) PSCAL(u0) PSCAL(dudt0) PSCAL(dudt) PSCAL(gpot) PSCAL(gpot_hydro)
  PSCAL(dt) PSCAL(dt_next) PSCAL(tlast) PSCAL(div_v) PSCAL(alpha) PSCAL(dalphadt)
  PSCAL(invomega) PSCAL(zeta)
  { vector<double> v; v.push_back(sim->t); v.push_back(sim->timestep); out.d("t_timestep", v); }
  // star particles of a hybrid gas + N-body run (NbodyParticle.h)
  if (sim->nbody && sim->nbody->Nstar > 0) {
    const int Ns = sim->nbody->Nstar;
    StarParticle<ndim> *st = sim->nbody->stardata;
    { vector<int> v(1, Ns); out.i("Nstar", v); }
#define SVEC(field) { vector<double> v((size_t)Ns*ndim); for (int i=0;i<Ns;i++) for (int k=0;k<ndim;k++) v[(size_t)i*ndim+k]=st[i].field[k]; out.d("star_" #field, v, ndim); }
#define SSCAL(field) { vector<double> v(Ns); for (int i=0;i<Ns;i++) v[i]=st[i].field; out.d("star_" #field, v); }
    SVEC(r) SVEC(v) SVEC(a) SVEC(adot) SVEC(r0) SVEC(v0) SVEC(a0) SVEC(adot0)
    SSCAL(m) SSCAL(h) SSCAL(gpot) SSCAL(dt) SSCAL(tlast) SSCAL(invh) SSCAL(radius) SSCAL(dt_internal)
    { vector<int> v(Ns); for (int i=0;i<Ns;i++) v[i]=st[i].level; out.i("star_level", v); }
    { vector<int> v(Ns); for (int i=0;i<Ns;i++) v[i]=st[i].nstep; out.i("star_nstep", v); }
    { vector<int> v(Ns); for (int i=0;i<Ns;i++) v[i]=st[i].nlast; out.i("star_nlast", v); }
  }
  // sink particles (Sinks.h: SinkParticle): every sink is star number istar
  PINT(sinkid)
  { vector<double> v; v.push_back(sph->mmean); v.push_back(sph->hmin_sink); out.d("mmean_hminsink", v); }
  if (sim->sinks && sim->sink_particles == 1) {
    const int Nk = sim->sinks->Nsink;
    SinkParticle<ndim> *sk = sim->sinks->sink;
    { vector<int> v(1, Nk); out.i("Nsink", v); }
#define KSCAL(field) { vector<double> v(Nk); for (int i=0;i<Nk;i++) v[i]=sk[i].field; out.d("sink_" #field, v); }
    KSCAL(radius) KSCAL(mmax) KSCAL(menc) KSCAL(dmdt) KSCAL(ketot) KSCAL(gpetot) KSCAL(rotketot) KSCAL(utot)
    KSCAL(taccrete) KSCAL(trad) KSCAL(trot) KSCAL(tvisc)
    { vector<double> v((size_t)Nk*3); for (int i=0;i<Nk;i++) for (int k=0;k<3;k++) v[(size_t)i*3+k]=sk[i].angmom[k]; out.d("sink_angmom", v, 3); }
    { vector<int> v(Nk); for (int i=0;i<Nk;i++) v[i]=sk[i].Ngas; out.i("sink_Ngas", v); }
    { vector<int> v(Nk); for (int i=0;i<Nk;i++) v[i]=(int)(sk[i].star - sim->nbody->stardata); out.i("sink_istar", v); }
  }
  { vector<int> v; v.push_back(sim->n); v.push_back(sim->Nsteps); v.push_back(sim->nresync); out.i("n_Nsteps_nresync", v); }
  { vector<int> v; v.push_back(sim->level_max); v.push_back(sim->level_step); v.push_back(sim->Nlevels); v.push_back(sim->level_diff_max); out.i("levelmax_levelstep_Nlevels_diffmax", v); }
  { vector<double> v; v.push_back(sim->dt_max); out.d("dt_max", v); }
}

#define CSCAL(field)  { vector<double> v(Nc); for (int c=0;c<Nc;c++) v[c]=cd[c].field; out.d("cell_" #field, v); }
#define CINT(field)   { vector<int> v(Nc); for (int c=0;c<Nc;c++) v[c]=cd[c].field; out.i("cell_" #field, v); }
#define CVEC(name,expr) { vector<double> v((size_t)Nc*ndim); for (int c=0;c<Nc;c++) for (int k=0;k<ndim;k++) v[(size_t)c*ndim+k]=cd[c].expr[k]; out.d("cell_" name, v, ndim); }

template <int ndim>
static void dump_tree(Dump &out, Simulation<ndim> *sim)
{
  typedef KDTree<ndim, GradhSphParticle, KDTreeCell> TreeT;
  TreeT *tree = static_cast<TreeT*>(sim->sphneib->GetTree());
  if (!tree) return;                                  // HipSphTree: the tree lives on the device
  KDTreeCell<ndim> *cd = tree->celldata;
  const int Nc = tree->Ncell;
  { vector<int> v; v.push_back(tree->Ncell); v.push_back(tree->ltot); v.push_back(tree->gtot); v.push_back(tree->Ntot); v.push_back(tree->Nleafmax); out.i("tree_Ncell_ltot_gtot_Ntot_Nleafmax", v); }
  CINT(cnext) CINT(copen) CINT(level) CINT(ifirst) CINT(ilast) CINT(N) CINT(Nactive)
  CSCAL(cdistsqd) CSCAL(m) CSCAL(rmax) CSCAL(hmax) CSCAL(maxsound)
  CVEC("bbmin", bb.min) CVEC("bbmax", bb.max) CVEC("hboxmin", hbox.min) CVEC("hboxmax", hbox.max)
  CVEC("rcell", rcell) CVEC("r", r) CVEC("v", v)
  { vector<double> v((size_t)Nc*5); for (int c=0;c<Nc;c++) for (int k=0;k<5;k++) v[(size_t)c*5+k]=cd[c].q[k]; out.d("cell_q", v, 5); }
  const int Ntot = tree->Ntot;
  { vector<int> v(tree->inext, tree->inext + Ntot); out.i("inext", v); }
}

// gather neighbour ids of every particle inside kernrange*h_i, through the reference's own
// point search (HydroTree::GetGatherNeighbourList, HydroTree.cpp:451-471: real + periodic-ghost trees).
template <int ndim>
static void dump_gather_lists(Dump &out, Simulation<ndim> *sim)
{
  Sph<ndim> *sph = static_cast<Sph<ndim>*>(sim->hydro);
  GradhSphParticle<ndim> *p = static_cast<GradhSphParticle<ndim>*>(sph->GetSphParticleArray());
  const int N = sph->Nhydro;
  vector<int> offs(N+1, 0), ids;
  int cap = 4096;
  vector<int> buf(cap);
  for (int i = 0; i < N; i++) {
    FLOAT rp[ndim];
    for (int k = 0; k < ndim; k++) rp[k] = p[i].r[k];
    int nn;
    while ((nn = sim->sphneib->GetGatherNeighbourList(rp, sph->kernp->kernrange*p[i].h, p, N, cap, buf.data())) < 0) {
      cap *= 2; buf.resize(cap);
    }
    for (int k = 0; k < nn; k++) {
      int j = buf[k];
      while (j >= N) j = p[j].iorig;          // periodic ghosts (and ghosts of ghosts) -> real parent
      ids.push_back(j);
    }
    offs[i+1] = (int) ids.size();
  }
  out.i("gather_offsets", offs);
  out.i("gather_ids", ids);
}

template <int ndim>
static void set_all_active(Simulation<ndim> *sim)
{
  Sph<ndim> *sph = static_cast<Sph<ndim>*>(sim->hydro);
  for (int i = 0; i < sph->Nhydro; i++) sph->GetSphParticlePointer(i).flags.set(active);
}

template <int ndim>
static int run(const string &mode_in, Parameters *params, SimulationBase *simbase, int argc, char **argv)
{
  const bool hip = mode_in.compare(0, 3, "hip") == 0;          // hipsteps, hippasses, hiptime: the same modes through HipSphTree
  const string mode = hip ? mode_in.substr(3) : mode_in;
#ifndef REF_HIPSHELL
  if (hip) { fprintf(stderr, "ref_dump: hip* modes need the ref_hipshell build (make -f oracle/ref.mk hipshell)\n"); return 1; }
#endif
  Simulation<ndim> *sim = static_cast<Simulation<ndim>*>(simbase);
  Sph<ndim> *sph;
#ifdef REF_HIPSHELL
  if (hip) {
    // SimulationBase::SetupSimulation (Simulation.cpp:639-694) call for call, with the neighbour-search object the
    // parameter processing made (GradhSphSimulation.cpp:232-246) replaced by the HIP binding before the IC is generated:
    // PostInitialConditionsSetup and every MainLoop then run the reference's own code against libgandalf_hip.so
    sim->ProcessParameters();
    HipSphTree<ndim> *shell = new HipSphTree<ndim>(sim->simparams, &sim->simbox, static_cast<Sph<ndim>*>(sim->hydro));
    sim->sphneib = shell; sim->neib = shell;
    sim->GenerateIC();
    if (sim->simparams->intparams["com_frame"] == 1) sim->SetComFrame();
    sim->PostInitialConditionsSetup();
    sim->Output();
  }
  else
#endif
  if (getenv("REF_H_PROVIDED")) {
    // SimulationBase::SetupSimulation (Simulation.cpp:639-694) call for call, declaring the smoothing lengths of the IC
    // file as provided: ic = file always leaves that flag false (SimulationIC.hpp:91) although the readers read h,
    // and without it PostInitialConditionsSetup replaces h by one global
    // guess (Sph.cpp:76-119) and spends ~500 s converging it at 1M Plummer particles.
    // bench.py's same-size CPU baseline starts the reference from the GPU run's state this way.
    sim->ProcessParameters();
    sim->GenerateIC();
    sim->initial_h_provided = true;
    if (sim->simparams->intparams["com_frame"] == 1) sim->SetComFrame();
    sim->PostInitialConditionsSetup();
    sim->Output();
  }
  else {
    if (getenv("REF_RESTART")) sim->restart = true;          // gandalf.cpp -r: continue from the snapshot named in <run_id>.restart
    sim->SetupSimulation();
  }
  sph = static_cast<Sph<ndim>*>(sim->hydro);

  if (mode == "time") {
    const int nsteps = atoi(argv[3]);
    const int warm   = argc > 4 ? atoi(argv[4]) : 0;
    for (int s = 0; s < warm; s++) sim->MainLoop();
    const double t0 = wall();
    for (int s = 0; s < nsteps; s++) sim->MainLoop();
    const double dt = wall() - t0;
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    printf("{\"N\": %d, \"steps\": %d, \"warmup\": %d, \"seconds\": %.6f, \"particle_steps_per_s\": %.6e, \"threads\": %d}\n",
           sph->Nhydro, nsteps, warm, dt, (double) sph->Nhydro*nsteps/dt, nthreads);
    return 0;
  }

  const string prefix = argv[3];
  { Dump out(prefix + "_setup.gdmp"); dump_particles<ndim>(out, sim); dump_tree<ndim>(out, sim); dump_gather_lists<ndim>(out, sim); }

  if (mode == "snap") {
    // the reference's own snapshot writers on the post-setup state (SimulationIO.hpp:274-540, 2009-2254)
    const int nsteps = argc > 4 ? atoi(argv[4]) : 0;
    for (int s = 0; s < nsteps; s++) sim->MainLoop();
    sim->WriteSnapshotFile(prefix + ".column", "column");
    sim->WriteSnapshotFile(prefix + ".su", "su");
    sim->WriteSnapshotFile(prefix + ".sf", "sf");
    { Dump out(prefix + "_snap.gdmp"); dump_particles<ndim>(out, sim);
      vector<double> v; v.push_back(sim->t); v.push_back(sim->tsnaplast); v.push_back(sph->mmean); v.push_back(sim->tlitesnaplast); v.push_back(sph->h_fac); out.d("snap_t_tsnaplast_mmean_tlitesnaplast_hfac", v);
      vector<int> w; w.push_back(sim->Noutsnap); w.push_back(sim->Nsteps); w.push_back(sim->Noutlitesnap); out.i("snap_Noutsnap_Nsteps_Noutlitesnap", w); }
    return 0;
  }
  if (mode == "passes") {
    // MainLoop order (SphSimulation.cpp:634-709) at fixed positions, all particles active
    set_all_active<ndim>(sim);
    sim->sphneib->BuildTree(true, 0, sim->ntreebuildstep, sim->ntreestockstep, sim->timestep, sph);
    sim->sphneib->SearchBoundaryGhostParticles((FLOAT) 0.0, sim->simbox, sph);
    sim->sphneib->BuildGhostTree(true, 0, sim->ntreebuildstep, sim->ntreestockstep, sim->timestep, sph);
    { Dump out(prefix + "_tree.gdmp"); dump_tree<ndim>(out, sim); }
    sim->sphneib->UpdateAllSphProperties(sph, sim->nbody);
    { Dump out(prefix + "_density.gdmp"); dump_particles<ndim>(out, sim); dump_tree<ndim>(out, sim); dump_gather_lists<ndim>(out, sim); }
    sph->ZeroAccelerations();
    if (sph->self_gravity == 1) sim->sphneib->UpdateAllSphForces(sph, sim->nbody, sim->simbox, sim->ewald);
    else sim->sphneib->UpdateAllSphHydroForces(sph, sim->nbody, sim->simbox);
    { Dump out(prefix + "_forces.gdmp"); dump_particles<ndim>(out, sim); }
    if (sim->nbody->Nstar > 0) {
      // the star part of MainLoop (SphSimulation.cpp:771-812) at fixed positions: gas -> star tree forces, then star-star
      Nbody<ndim> *nb = sim->nbody;
      for (int i = 0; i < nb->Nnbody; i++) {
        nb->nbodydata[i]->flags.set(active);
        for (int k = 0; k < ndim; k++) { nb->nbodydata[i]->a[k] = 0.0; nb->nbodydata[i]->adot[k] = 0.0; nb->nbodydata[i]->a2dot[k] = 0.0; nb->nbodydata[i]->a3dot[k] = 0.0; }
        nb->nbodydata[i]->gpot = 0.0; nb->nbodydata[i]->gpe = 0.0;
      }
      sim->sphneib->UpdateAllStarGasForces(sph, nb, sim->simbox, sim->ewald);
      { Dump out(prefix + "_stargas.gdmp"); dump_particles<ndim>(out, sim); }
      if (nb->nbody_softening == 1) nb->CalculateDirectSmoothedGravForces(nb->Nnbody, nb->nbodydata, sim->simbox, sim->ewald);
      else nb->CalculateDirectGravForces(nb->Nnbody, nb->nbodydata, sim->simbox, sim->ewald);
      { Dump out(prefix + "_starall.gdmp"); dump_particles<ndim>(out, sim); }
    }
  }
  else if (mode == "steps") {
    const int nsteps = atoi(argv[4]);
    for (int s = 0; s < nsteps; s++) sim->MainLoop();
    { Dump out(prefix + "_final.gdmp"); dump_particles<ndim>(out, sim); }
  }
  else if (mode == "run") {
    // SimulationBase::Run: MainLoop + Output (regular snapshots and <run_id>.restart in the working directory)
    const int nsteps = atoi(argv[4]);
    sim->Run(nsteps);
    Dump out(prefix + "_final.gdmp"); dump_particles<ndim>(out, sim);
    vector<double> v; v.push_back(sim->t); v.push_back(sim->tsnaplast); v.push_back(sim->tsnapnext); out.d("run_t_tsnaplast_tsnapnext", v);
    vector<int> w; w.push_back(sim->Noutsnap); w.push_back(sim->Nsteps); out.i("run_Noutsnap_Nsteps", w);
  }
  return 0;
}

// ---- N-body direct sum -------------------------------------------------------------------------
#include "CodeTiming.h"
#include "DomainBox.h"
#include "SmoothingKernel.h"
#include "StarParticle.h"

static void dump_stars(Dump &out, int N, StarParticle<3> *p, double t, double dt)
{
  const int ndim = 3;
  PVEC(r) PVEC(v) PVEC(a) PVEC(adot) PVEC(r0) PVEC(v0) PVEC(a0) PSCAL(m) PSCAL(h) PSCAL(gpot) PSCAL(dt)
  vector<double> tt(2); tt[0] = t; tt[1] = dt; out.d("t_dt", tt);
}

static int run_nbody(int argc, char **argv)
{
  if (argc < 6) { fprintf(stderr, "usage: ref_dump nbody <N> <softening> <nsteps> <prefix>\n"); return 1; }
  const int N = atoi(argv[2]), soft = atoi(argv[3]), nsteps = atoi(argv[4]);
  const string prefix = argv[5];
  const double nbody_mult = 0.1;
  ExceptionHandler::makeExceptionHandler(cplusplus);
  NbodyLeapfrogKDK<3, M4Kernel> nb(soft, 0, 0, nbody_mult, "m4");
  CodeTiming timing;
  nb.timing = &timing;
  DomainBox<3> box;
  for (int k = 0; k < 3; k++) {
    box.boundary_lhs[k] = openBoundary; box.boundary_rhs[k] = openBoundary;
    box.min[k] = -1e30; box.max[k] = 1e30; box.size[k] = 2e30; box.half[k] = 1e30;
  }
  box.PeriodicGravity = false;

  // synthetic cluster: LCG x <- x*6364136223846793005 + 1442695040888963407, u = (x >> 11) * 2^-53
  uint64_t x = 88172645463325252ull;
  auto u01 = [&]() { x = x*6364136223846793005ull + 1442695040888963407ull; return (double) (x >> 11)*(1.0/9007199254740992.0); };
  StarParticle<3> *star = new StarParticle<3>[N];
  NbodyParticle<3> **ptr = new NbodyParticle<3>*[N];
  for (int i = 0; i < N; i++) {
    for (int k = 0; k < 3; k++) star[i].r[k] = u01();
    for (int k = 0; k < 3; k++) star[i].v[k] = 0.2*(u01() - 0.5);
    star[i].m = (0.5 + u01())/N;
    star[i].h = 0.02*(1.0 + u01());
    star[i].invh = 1.0/star[i].h;
    star[i].dt_internal = big_number;
    star[i].istar = i;
    ptr[i] = &star[i];
  }
  auto zero = [&]() {
    for (int i = 0; i < N; i++) if (ptr[i]->flags.check(active)) {
      for (int k = 0; k < 3; k++) { ptr[i]->a[k] = 0.0; ptr[i]->adot[k] = 0.0; ptr[i]->a2dot[k] = 0.0; ptr[i]->a3dot[k] = 0.0; }
      ptr[i]->gpot = 0.0;
    }
  };
  auto forces = [&]() {
    if (soft) nb.CalculateDirectSmoothedGravForces(N, ptr, box, NULL);
    else nb.CalculateDirectGravForces(N, ptr, box, NULL);
  };
  double timestep = 0.0;
  auto global_timestep = [&]() {          // Simulation.cpp:1720-1745, stars only, Nlevels = 1
    double dt_min = big_number_dp;
    for (int i = 0; i < N; i++) {
      ptr[i]->flags.set(end_timestep);
      ptr[i]->level = 0;
      ptr[i]->nstep = 1;
      ptr[i]->dt_next = nb.Timestep(ptr[i]);
      dt_min = min(dt_min, ptr[i]->dt_next);
    }
    timestep = dt_min;
    for (int i = 0; i < N; i++) ptr[i]->dt_next = timestep;
  };
  // setup (NbodySimulation::PostInitialConditionsSetup, NbodySimulation.cpp:150-248)
  for (int i = 0; i < N; i++) {
    ptr[i]->flags.set(active);
    for (int k = 0; k < 3; k++) { ptr[i]->r0[k] = ptr[i]->r[k]; ptr[i]->v0[k] = ptr[i]->v[k]; }
    ptr[i]->nlast = 0; ptr[i]->tlast = 0.0; ptr[i]->nstep = 1;
  }
  double t = 0.0;
  zero(); forces();
  global_timestep();
  nb.EndTimestep(0, N, t, timestep, ptr);
  { Dump out(prefix + "_setup.gdmp"); dump_stars(out, N, star, t, timestep); }
  for (int s = 0; s < nsteps; s++) {
    t = t + timestep;
    nb.AdvanceParticles(1, N, t, timestep, ptr);
    zero(); forces();
    nb.CorrectionTerms(1, N, t, timestep, ptr);
    global_timestep();
    nb.EndTimestep(0, N, t, timestep, ptr);
  }
  { Dump out(prefix + "_final.gdmp"); dump_stars(out, N, star, t, timestep); }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc >= 2 && string(argv[1]) == "nbody") return run_nbody(argc, argv);
  if (argc < 4) {
    fprintf(stderr, "usage: ref_dump passes|steps|time <params.dat> ...\n");
    return 1;
  }
  const string mode = argv[1];
  Parameters *params = new Parameters();
  ExceptionHandler::makeExceptionHandler(cplusplus);
  params->ReadParamsFile(string(argv[2]));
  SimulationBase *sim = SimulationBase::SimulationFactory(params->intparams["ndim"],
                                                          params->stringparams["sim"], params);
  sim->restart = false;
  const int ndim = params->intparams["ndim"];
  if (ndim == 1) return run<1>(mode, params, sim, argc, argv);
  if (ndim == 2) return run<2>(mode, params, sim, argc, argv);
  return run<3>(mode, params, sim, argc, argv);
}
