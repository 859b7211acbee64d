// host_capi.cpp -- small extern "C" surface of libgandalf_host.so so that tests / bench.py can drive the
// C++ host shell (parameter files, IC generators, SetupSimulation / MainLoop) from Python.
#include "SphSimulation.h"
#include <cstring>
#include <string>

struct gah_sim { Parameters params; SphSimulation *sim = nullptr; std::string err; bool restart = false, output = false; };

#define GAH_TRY(s, body) try { body; return 0; } catch (const std::exception &e) { (s)->err = e.what(); return -1; }

extern "C" {

gah_sim *gah_create(void) { return new gah_sim(); }
void gah_destroy(gah_sim *s) { if (s) { delete s->sim; delete s; } }
const char *gah_last_error(gah_sim *s) { return s->err.c_str(); }
int gah_read_params(gah_sim *s, const char *file) { GAH_TRY(s, s->params.ReadParamsFile(file)) }
int gah_set_param(gah_sim *s, const char *key, const char *value) { GAH_TRY(s, s->params.SetParameter(key, value)) }
int gah_get_param(gah_sim *s, const char *key, char *out, int cap)
{
  const std::string v = s->params.GetParameter(key);
  if ((int) v.size() + 1 > cap) return -1;
  memcpy(out, v.c_str(), v.size() + 1);
  return 0;
}
// ProcessParameters + GenerateIC (+ SetComFrame): host-side only, particles stay on the host
int gah_generate_ic(gah_sim *s)
{
  GAH_TRY(s, {
    if (!s->sim) s->sim = SphSimulation::SimulationFactory(s->params.intparams["ndim"], s->params.stringparams["sim"], &s->params);
    s->sim->restart = s->restart; s->sim->write_output = s->output;
    s->sim->ProcessParameters();
    s->sim->GenerateIC();
    if (s->params.intparams["com_frame"] == 1) s->sim->SetComFrame();
    if (!s->sim->initial_h_provided) s->sim->sph->InitialSmoothingLengthGuess();
  })
}
int gah_num_particles(gah_sim *s) { return s->sim ? s->sim->sph->part.N : 0; }
int gah_initial_h_provided(gah_sim *s) { return s->sim && s->sim->initial_h_provided ? 1 : 0; }
int gah_get_ic(gah_sim *s, double *r, double *v, double *m, double *h, double *u)
{
  if (!s->sim) return -1;
  const HydroParticles &p = s->sim->sph->part;
  if (r) memcpy(r, p.r.data(), sizeof(double)*p.r.size());
  if (v) memcpy(v, p.v.data(), sizeof(double)*p.v.size());
  if (m) memcpy(m, p.m.data(), sizeof(double)*p.m.size());
  if (h) memcpy(h, p.h.data(), sizeof(double)*p.h.size());
  if (u) memcpy(u, p.u.data(), sizeof(double)*p.u.size());
  return 0;
}
int gah_post_ic_setup(gah_sim *s) { GAH_TRY(s, s->sim->PostInitialConditionsSetup()) }
// multi-GPU: register rank, size and the collectives (gh_comm_ops*) before the setup creates the device context
int gah_init_comm(gah_sim *s, int rank, int nranks, const void *ops)
{
  GAH_TRY(s, {
    if (!s->sim) s->sim = SphSimulation::SimulationFactory(s->params.intparams["ndim"], s->params.stringparams["sim"], &s->params);
    s->sim->InitComm(rank, nranks, (const gh_comm_ops*) ops);
  })
}
// create the device context and upload the host particles without running the setup passes
// (the multi-GPU runner drives those itself, with an exchange after every sliced pass)
int gah_upload_ic(gah_sim *s)
{
  GAH_TRY(s, {
    s->sim->EnsureContext();
    const HydroParticles &p = s->sim->sph->part;
    if (gh_upload_particles(s->sim->ctx, p.N, p.r.data(), p.v.data(), p.m.data(), p.h.data(), p.u.data()))
      throw GandalfError(gh_last_error(s->sim->ctx));
    s->sim->setup = true;
  })
}
int gah_setup(gah_sim *s)
{
  GAH_TRY(s, {
    if (!s->sim) s->sim = SphSimulation::SimulationFactory(s->params.intparams["ndim"], s->params.stringparams["sim"], &s->params);
    s->sim->restart = s->restart; s->sim->write_output = s->output;
    s->sim->SetupSimulation();
  })
}
int gah_main_loop(gah_sim *s, int nsteps) { GAH_TRY(s, s->sim->MainLoop(nsteps)) }
// SimulationBase::Run (MainLoop + Output until tend / Nstepsmax / nsteps more steps); restart: gandalf.cpp's -r
int gah_run(gah_sim *s, int nsteps) { GAH_TRY(s, s->sim->Run(nsteps)) }
int gah_set_restart(gah_sim *s, int on) { s->restart = on != 0; if (s->sim) s->sim->restart = s->restart; return 0; }
int gah_set_output(gah_sim *s, int on) { s->output = on != 0; if (s->sim) s->sim->write_output = s->output; return 0; }
// output scale of a quantity (r, m, t, v, a, rho, u, temp, angvel) after gah_generate_ic / gah_setup: x[code] = x[output unit]/scale
double gah_unit_outscale(gah_sim *s, const char *q)
{
  if (!s->sim) return 0.0;
  const SimUnits &u = s->sim->simunits;
  const std::string k(q);
  const SimUnit *p = k == "r" ? &u.r : k == "m" ? &u.m : k == "t" ? &u.t : k == "v" ? &u.v : k == "a" ? &u.a : k == "rho" ? &u.rho : k == "u" ? &u.u :
                     k == "temp" ? &u.temp : k == "angvel" ? &u.angvel : nullptr;
  return p ? p->outscale : 0.0;
}
int gah_nsteps(gah_sim *s) { return s->sim ? s->sim->Nsteps : 0; }
int gah_noutsnap(gah_sim *s) { return s->sim ? s->sim->Noutsnap : 0; }
double gah_time(gah_sim *s) { return s->sim->t; }
double gah_timestep(gah_sim *s) { return s->sim->timestep; }
gh_ctx *gah_ctx(gah_sim *s) { return s->sim ? s->sim->ctx : nullptr; }
int gah_write_snapshot(gah_sim *s, const char *filename, const char *fileform) { GAH_TRY(s, s->sim->WriteSnapshotFile(filename, fileform)) }
// Simulation::CalculateDiagnostics + RecordDiagnostics (<run_id>.diag line; out29 may be NULL, filename may be NULL)
int gah_diagnostics(gah_sim *s, double *out29, const char *filename)
{
  GAH_TRY(s, {
    double tmp[29];
    if (filename) s->sim->RecordDiagnostics(filename);
    if (out29 || !filename) { s->sim->CalculateDiagnostics(out29 ? out29 : tmp); }
  })
}
// CodeTiming::ComputeTimingStatistics (<run_id>.timing)
int gah_write_timing(gah_sim *s, const char *filename) { GAH_TRY(s, s->sim->WriteTimingStatistics(filename)) }

// ---- snapshot files without a simulation object (SnapshotIO.h); header = {Noutsnap, Nsteps, Noutlitesnap},
//      hd = {tsnaplast, mmean, tlitesnaplast, h_fac}
static std::string g_snap_err;
const char *gah_snapshot_error(void) { return g_snap_err.c_str(); }
int gah_snapshot_write(const char *filename, const char *fileform, int ndim, int N, double t, const double *r, const double *v,
                       const double *m, const double *h, const double *rho, const double *u, const int *iorig,
                       const long *hl, const double *hd)
{
  try {
    Snapshot s;
    s.ndim = ndim; s.N = N; s.t = t;
    s.r.assign(r, r + (size_t) N*ndim); s.v.assign(v, v + (size_t) N*ndim);
    s.m.assign(m, m + N); s.h.assign(h, h + N); s.rho.assign(rho, rho + N); s.u.assign(u, u + N);
    if (iorig) s.iorig.assign(iorig, iorig + N);
    if (hl) { s.Noutsnap = hl[0]; s.Nsteps = hl[1]; s.Noutlitesnap = hl[2]; }
    if (hd) { s.tsnaplast = hd[0]; s.mmean = hd[1]; s.tlitesnaplast = hd[2]; s.h_fac = hd[3]; }
    WriteSnapshotFile(filename, fileform, s);
    return 0;
  } catch (const std::exception &e) { g_snap_err = e.what(); return -1; }
}
Snapshot *gah_snapshot_open(const char *filename, const char *fileform)
{
  Snapshot *s = new Snapshot();
  try { ReadSnapshotFile(filename, fileform, *s); return s; }
  catch (const std::exception &e) { g_snap_err = e.what(); delete s; return nullptr; }
}
void gah_snapshot_info(Snapshot *s, int *ndim, int *N, double *t, long *hl, double *hd)
{
  *ndim = s->ndim; *N = s->N; *t = s->t;
  if (hl) { hl[0] = s->Noutsnap; hl[1] = s->Nsteps; hl[2] = s->Noutlitesnap; }
  if (hd) { hd[0] = s->tsnaplast; hd[1] = s->mmean; hd[2] = s->tlitesnaplast; hd[3] = s->h_fac; }
}
void gah_snapshot_data(Snapshot *s, double *r, double *v, double *m, double *h, double *rho, double *u, int *iorig)
{
  memcpy(r, s->r.data(), sizeof(double)*s->r.size()); memcpy(v, s->v.data(), sizeof(double)*s->v.size());
  memcpy(m, s->m.data(), sizeof(double)*s->m.size()); memcpy(h, s->h.data(), sizeof(double)*s->h.size());
  memcpy(rho, s->rho.data(), sizeof(double)*s->rho.size()); memcpy(u, s->u.data(), sizeof(double)*s->u.size());
  memcpy(iorig, s->iorig.data(), sizeof(int)*s->iorig.size());
}
void gah_snapshot_close(Snapshot *s) { delete s; }

}
