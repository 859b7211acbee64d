// SphSimulation.h -- host shell that keeps GANDALF's driver / operator surface and forwards the hot
// path to libgandalf_hip through the C ABI (include/gandalf_hip.h).
//
// Class and method names follow the reference so that a maintainer finds the same seams:
//   SimulationBase::SimulationFactory / SetupSimulation / MainLoop / Run  (Simulation.h:95-255, Simulation.cpp:63, 382, 639)
//   Sph::InitialSmoothingLengthGuess / ZeroAccelerations                   (Sph.cpp:76-140)
//   SphNeighbourSearch::BuildTree / UpdateAllSphProperties / UpdateAllSphHydroForces / UpdateAllSphForces
//                                                                          (SphNeighbourSearch.h:76-94)
//   SphLeapfrogKDK::AdvanceParticles / EndTimestep                          (SphLeapfrogKDK.cpp:76, 219)
//   Nbody::CalculateDirectGravForces                                        (Nbody.cpp:233-287)
// Only what the SPH + tree-gravity path needs is here (+ the column / su snapshot formats, SnapshotIO.h); everything
// else GANDALF does (units, radiation, dust, MFV, sinks, MPI) is out of scope.
#pragma once
#include <string>
#include <vector>
#include "Parameters.h"
#include "RandomNumber.h"
#include "SnapshotIO.h"
#include "../../include/gandalf_hip.h"

class GandalfError : public std::exception {
 public:
  explicit GandalfError(const std::string &m) : msg(m) {}
  const char *what() const noexcept override { return msg.c_str(); }
  std::string msg;
};

// Host copy of the particle data in the caller's order (the reference's AoS GradhSphParticle array,
// here as SoA vectors: the hot fields only)
struct HydroParticles {
  int N = 0, ndim = 3;
  std::vector<double> r, v, m, h, u;       // r, v: [N][ndim]
};

class Sph {
 public:
  Sph(int ndim, double h_fac, double kernrange) : ndim(ndim), h_fac(h_fac), kernrange(kernrange) {}
  void AllocateMemory(int N);
  void InitialSmoothingLengthGuess();      // Sph.cpp:76-119
  HydroParticles part;
  int ndim;
  double h_fac, kernrange;
  int Ngather = 0;
};

// SimUnits (SimUnits.cpp:825-1118) for the quantities this path reads or writes.  Code units ARE the output units of length
// and mass (outscale = 1) with G = 1; every other quantity follows: x[code] = x[output unit] / outscale.  A dimensionless
// run has every factor 1.  outSI: the output unit in SI; outcgs: in cgs (rho_sink and rho_bary are given in g cm^-3
// whatever rhooutunit is, SphSimulation.cpp:129, BarotropicEOS.cpp:42).
struct SimUnit { double outscale = 1.0, outSI = 1.0, outcgs = 1.0; std::string outunit; };
struct SimUnits {
  bool dimensionless = true;
  SimUnit r, m, t, v, a, rho, u, temp, angvel;
  void SetupUnits(Parameters *params);
  std::vector<std::string> unit_strings(Parameters *params) const;      // the 21 unit ids of a snapshot header (SimulationIO.hpp:1040-1064)
};

class SphNeighbourSearch {                 // neib_search = kdtree on the GPU
 public:
  explicit SphNeighbourSearch(gh_ctx *ctx) : ctx(ctx) {}
  void BuildTree();
  void UpdateAllSphProperties(gh_stats *st = nullptr);
  void UpdateAllSphHydroForces(gh_stats *st = nullptr);
  void UpdateAllSphForces(gh_stats *st = nullptr);
  gh_ctx *ctx;
};

class SphSimulation {
 public:
  static SphSimulation *SimulationFactory(int ndim, const std::string &simtype, Parameters *params);
  SphSimulation(int ndim, Parameters *params);
  ~SphSimulation();

  void ProcessParameters();                // SphSimulation.cpp:67 / GradhSphSimulation.cpp:54
  void EnsureContext();
  void InitComm(int rank, int nranks, const gh_comm_ops *ops);
  int comm_rank = 0, comm_nranks = 1;
  gh_comm_ops comm_ops = {};
  void GenerateIC();                       // SimulationIC.hpp:51 (ic = box | plummer)
  void SetComFrame();                      // Simulation.cpp:1621
  void PostInitialConditionsSetup();       // SphSimulation.cpp:204
  void SetupSimulation();                  // Simulation.cpp:639
  void MainLoop(int nsteps = 1);           // SphSimulation.cpp:574
  void Run(int Nadvance = -1);             // Simulation.cpp:382 (MainLoop + Output until tend / Nstepsmax)
  std::string Output();                    // Simulation.cpp:500-600: regular snapshots <run_id>.<form>.NNNNN + <run_id>.restart
  void Download(int field, std::vector<double> &out);
  void WriteSnapshotFile(const std::string &filename, const std::string &fileform);   // SimulationIO.hpp:96
  void CalculateDiagnostics(double *out29);        // SimAnalysis.hpp:52-200 (returns the numbers of one .diag line)
  void RecordDiagnostics(const std::string &filename);   // SimAnalysis.hpp:262-300: appends one line of <run_id>.diag
  void WriteTimingStatistics(const std::string &filename);   // CodeTiming::ComputeTimingStatistics (CodeTiming.cpp:238-420): <run_id>.timing
  double wall_start = 0.0;

  int ndim;
  Parameters *simparams;
  SimUnits simunits;
  gh_config cfg;
  gh_ctx *ctx = nullptr;
  gh_nbody *nbody = nullptr;               // sink runs: the stars the sinks are (Nbody, Sinks; SphSimulation.cpp:116-136)
  Sph *sph = nullptr;
  SphNeighbourSearch *sphneib = nullptr;
  XorshiftRand *randnumb = nullptr;
  bool initial_h_provided = false;
  bool setup = false;
  // snapshot cadence and restarts (SimulationBase: restart, Noutsnap, tsnapnext, tsnaplast, dt_snap, run_id, out_file_form)
  bool write_output = false;               // Output() writes files (the executable: yes; an embedding host decides - gah_set_output)
  bool restart = false;                    // continue from the snapshot named in <run_id>.restart (gandalf.cpp -r)
  bool restarted_ids = false;              // ... whose particle ids (porig) this run keeps writing
  std::vector<int> restart_iorig;
  int Noutsnap = 0, nrestartstep = 512, nlastrestart = 0;
  void RestartSnapshot();                  // Simulation.cpp:609-632: <run_id>.<form>.tmp + <run_id>.restart naming it
  double tsnapnext = 0.0, tsnaplast = 0.0, dt_snap = 0.2;
  std::string run_id, out_file_form = "su";
  int Nsteps = 0, Nstepsmax = 0;
  double t = 0.0, timestep = 0.0, tend = 0.0;
};
