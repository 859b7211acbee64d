#include "Parameters.h"
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

static std::string trim(const std::string &s)
{
  std::string out;
  for (char c : s) if (c != ' ' && c != '\t' && c != '\r' && c != '\n') out += c;   // TrimWhiteSpace removes ALL blanks
  return out;
}

void Parameters::SetDefaultValues()
{
  // values: reference Parameters.cpp:160-520 (hot-path subset)
  intparams["ndim"] = 3;
  stringparams["sim"] = "sph";
  stringparams["nbody"] = "hermite4";
  stringparams["ic"] = "box";
  stringparams["run_id"] = "";
  floatparams["tend"] = 1.0;
  intparams["Nstepsmax"] = 99999999;
  intparams["noutputstep"] = 128;
  intparams["nrestartstep"] = 512;
  floatparams["dt_snap"] = 0.2;
  floatparams["tsnapfirst"] = 0.0;
  intparams["dimensionless"] = 0;
  // output (= code) units of a run with physical units (Parameters.cpp:214-235); the input units default to them
  stringparams["routunit"] = "pc"; stringparams["moutunit"] = "m_sun"; stringparams["toutunit"] = "myr";
  stringparams["voutunit"] = "km_s"; stringparams["aoutunit"] = "km_s2"; stringparams["rhooutunit"] = "g_cm3";
  stringparams["sigmaoutunit"] = "m_sun_pc2"; stringparams["pressoutunit"] = "Pa"; stringparams["foutunit"] = "N";
  stringparams["Eoutunit"] = "J"; stringparams["momoutunit"] = "m_sunkm_s"; stringparams["angmomoutunit"] = "m_sunkm2_s";
  stringparams["angveloutunit"] = "rad_s"; stringparams["dmdtoutunit"] = "m_sun_yr"; stringparams["Loutunit"] = "L_sun";
  stringparams["kappaoutunit"] = "m2_kg"; stringparams["Boutunit"] = "tesla"; stringparams["Qoutunit"] = "C";
  stringparams["Jcuroutunit"] = "C_s_m2"; stringparams["uoutunit"] = "J_kg"; stringparams["dudtoutunit"] = "J_kg_s";
  stringparams["tempoutunit"] = "K";
  floatparams["accel_mult"] = 0.3;
  floatparams["courant_mult"] = 0.15;
  intparams["Nlevels"] = 1;
  intparams["level_diff_max"] = 1;
  stringparams["in_file"] = "";
  stringparams["in_file_form"] = "su";
  stringparams["out_file_form"] = "su";
  intparams["sph_single_timestep"] = 0;
  stringparams["sph_integration"] = "lfkdk";
  stringparams["kernel"] = "m4";
  intparams["tabulated_kernel"] = 1;
  floatparams["h_fac"] = 1.2;
  floatparams["h_converge"] = 0.01;
  intparams["hydro_forces"] = 1;
  stringparams["gas_eos"] = "energy_eqn";
  stringparams["energy_integration"] = "null";
  floatparams["energy_mult"] = 0.4;
  floatparams["gamma_eos"] = 1.66666666666666;
  floatparams["temp0"] = 1.0;
  floatparams["mu_bar"] = 1.0;
  floatparams["rho_bary"] = 1.0e-14;
  stringparams["avisc"] = "mon97";
  stringparams["acond"] = "none";
  stringparams["time_dependent_avisc"] = "none";
  floatparams["alpha_visc"] = 1.0;
  floatparams["alpha_visc_min"] = 0.1;
  floatparams["beta_visc"] = 2.0;
  intparams["self_gravity"] = 0;
  stringparams["neib_search"] = "kdtree";
  stringparams["gravity_mac"] = "geometric";
  stringparams["multipole"] = "quadrupole";
  intparams["Nleafmax"] = 6;
  intparams["ntreebuildstep"] = 1;
  intparams["ntreestockstep"] = 1;
  floatparams["thetamaxsqd"] = 0.1;
  floatparams["macerror"] = 0.0001;
  for (int k = 0; k < 3; k++) {
    const std::string idx = "[" + std::to_string(k) + "]";
    stringparams["boundary_lhs" + idx] = "open";
    stringparams["boundary_rhs" + idx] = "open";
    floatparams["boxmin" + idx] = 0.0;
    floatparams["boxmax" + idx] = 0.0;
    intparams["Nlattice1" + idx] = 1;
    intparams["Nlattice2" + idx] = 1;
    floatparams["vfluid1" + idx] = 0.0;
    floatparams["vfluid2" + idx] = 0.0;
  }
  // initial conditions
  intparams["Nhydro"] = 0;
  intparams["Nstar"] = 0;
  stringparams["particle_distribution"] = "cubic_lattice";
  floatparams["rhofluid1"] = 1.0;
  floatparams["rhofluid2"] = 1.0;
  floatparams["press1"] = 1.0;
  floatparams["press2"] = 1.0;
  floatparams["mplummer"] = 1.0;
  floatparams["rplummer"] = 1.0;
  floatparams["radius"] = 1.0;
  floatparams["rstar"] = 0.1;
  floatparams["gasfrac"] = 0.0;
  floatparams["starfrac"] = 0.0;
  intparams["com_frame"] = 0;
  // Boss-Bodenheimer cloud (ic = bb) and sink particles (Parameters.cpp of the reference: same keys and defaults)
  floatparams["mcloud"] = 1.0;
  floatparams["angvel"] = 0.0;
  floatparams["amp"] = 0.1;
  intparams["sink_particles"] = 0;
  intparams["create_sinks"] = 0;
  intparams["smooth_accretion"] = 0;
  intparams["Nsinkfixed"] = -1;
  intparams["nbody_softening"] = 1;
  floatparams["nbody_mult"] = 0.1;
  floatparams["rho_sink"] = 1.0e-12;
  floatparams["sink_radius"] = 2.0;
  floatparams["alpha_ss"] = 0.01;
  floatparams["smooth_accrete_frac"] = 0.01;
  floatparams["smooth_accrete_dt"] = 0.01;
  stringparams["sink_radius_mode"] = "hmult";
  stringparams["rand_algorithm"] = "xorshift";
  intparams["randseed"] = 1;
  intparams["device"] = 0;               // (ours) HIP device ordinal of this process
}

void Parameters::SetParameter(const std::string &key, const std::string &value)
{
  if (intparams.count(key)) intparams[key] = std::atoi(value.c_str());
  else if (floatparams.count(key)) floatparams[key] = std::atof(value.c_str());
  else stringparams[key] = value;          // known string keys and keys of out-of-scope subsystems
}

std::string Parameters::GetParameter(const std::string &key) const
{
  std::ostringstream ss;
  ss.precision(17);
  if (intparams.count(key)) ss << intparams.at(key);
  else if (floatparams.count(key)) ss << floatparams.at(key);
  else if (stringparams.count(key)) ss << stringparams.at(key);
  return ss.str();
}

void Parameters::ParseLine(std::string line)
{
  line = trim(line);
  const size_t len = line.length();
  const size_t colon = line.find(':'), equal = line.find('='), hash = line.find('#');
  if (hash == 0 || len == 0) return;
  if (equal == std::string::npos || (colon != std::string::npos && colon > equal)) return;
  const size_t start = colon == std::string::npos ? 0 : colon + 1;
  SetParameter(line.substr(start, equal - start), line.substr(equal + 1));
}

void Parameters::ReadParamsFile(const std::string &filename)
{
  SetDefaultValues();
  std::ifstream f(filename.c_str());
  if (!f.is_open()) throw std::runtime_error("The specified parameter file: " + filename + " does not exist, aborting");
  std::string line;
  while (std::getline(f, line)) ParseLine(line);
  if (stringparams["run_id"] == "")
    throw std::runtime_error("The parameter file: " + filename + " does not contain a run id string, aborting");
}
