// gandalf_main.cpp -- `gandalf_hip <params.dat> [nsteps] [-r]`: the reference's command-line flow
// (gandalf.cpp:40-190: ReadParamsFile -> SimulationFactory -> SetupSimulation -> Run) on the HIP path,
// with a per-phase timing table that uses the reference's block names (CodeTiming).
#include "SphSimulation.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "No parameter file specified, aborting...\n"); return 1; }
  Parameters params;
  try {
    params.ReadParamsFile(argv[1]);
    SphSimulation *sim = SphSimulation::SimulationFactory(params.intparams["ndim"], params.stringparams["sim"], &params);
    sim->write_output = true;              // regular snapshots + <run_id>.restart, like the reference's executable
    for (int a = 2; a < argc; a++) if (std::string(argv[a]) == "-r") sim->restart = true;      // gandalf.cpp:108-111
    auto t0 = std::chrono::steady_clock::now();
    sim->SetupSimulation();
    auto t1 = std::chrono::steady_clock::now();
    gh_reset_timers(sim->ctx);
    const int nsteps = (argc > 2 && std::string(argv[2]) != "-r") ? atoi(argv[2]) : -1;
    sim->Run(nsteps);
    auto t2 = std::chrono::steady_clock::now();
    double ms[GH_T_COUNT];
    gh_get_timers(sim->ctx, ms, nullptr, nullptr);
    const double setup_s = std::chrono::duration<double>(t1 - t0).count(), run_s = std::chrono::duration<double>(t2 - t1).count();
    printf("t : %.10g   dt : %.10g   Nsteps : %d\n", sim->t, sim->timestep, sim->Nsteps);
    printf("SETUP %.4f s   RUN %.4f s   (%.4g particle-steps/s)\n", setup_s, run_s, (double) sim->sph->part.N*sim->Nsteps/run_s);
    printf("BUILD_TREE %.3f ms   SPH_PROPERTIES %.3f ms   %s %.3f ms   KDK+GLOBAL_TIMESTEPS %.3f ms\n", ms[GH_T_BUILD_TREE],
           ms[GH_T_SPH_PROPERTIES], params.intparams["self_gravity"] ? "SPH_ALL_FORCES" : "SPH_HYDRO_FORCES", ms[GH_T_SPH_FORCES], ms[GH_T_KDK]);
    delete sim;
  }
  catch (const std::exception &e) {
    fprintf(stderr, "%s\n", e.what());
    return -1;
  }
  return 0;
}
