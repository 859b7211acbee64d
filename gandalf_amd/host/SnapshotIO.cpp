// SnapshotIO.cpp -- see SnapshotIO.h
#include "SnapshotIO.h"
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <map>
#include "SphSimulation.h"

static const char *kSerenTag = "SERENBINARYDUMPV3";
static const int kStr = 20;                 // string_length, SimulationIO.hpp:44

// ---- column -------------------------------------------------------------------------------------
void WriteColumnSnapshotFile(const std::string &filename, const Snapshot &s)
{
  std::ofstream out(filename.c_str());
  if (!out) throw GandalfError("Cannot open snapshot file for writing : " + filename);
  const int nd = s.ndim;
  out << s.N << std::endl << 0 << std::endl << nd << std::endl << s.t << std::endl;      // Nhydro, Nstar, ndim, t
  const char *sep = "   ";
  for (int i = 0; i < s.N; i++) {
    for (int k = 0; k < nd; k++) out << s.r[(size_t) i*nd + k] << sep;
    for (int k = 0; k < nd; k++) out << s.v[(size_t) i*nd + k] << sep;
    out << s.m[i] << sep << s.h[i] << sep << s.rho[i] << sep << s.u[i];
    if (nd == 1) out << sep;                 // the 1-D branch of the reference ends its line with a separator (:466)
    out << std::endl;
  }
}

void ReadColumnSnapshotFile(const std::string &filename, Snapshot &s)
{
  std::ifstream in(filename.c_str());
  if (!in) throw GandalfError("Cannot open snapshot file : " + filename);
  int Nstar = 0, nd = 0;
  in >> s.N >> Nstar >> nd >> s.t;
  if (nd < 1 || nd > 3) throw GandalfError("Incorrect no. of dimensions in file : " + filename);
  s.ndim = nd;
  s.r.assign((size_t) s.N*nd, 0.0); s.v.assign((size_t) s.N*nd, 0.0);
  s.m.assign(s.N, 0.0); s.h.assign(s.N, 0.0); s.rho.assign(s.N, 0.0); s.u.assign(s.N, 0.0);
  s.iorig.resize(s.N);
  for (int i = 0; i < s.N && in.good(); i++) {
    for (int k = 0; k < nd; k++) in >> s.r[(size_t) i*nd + k];
    for (int k = 0; k < nd; k++) in >> s.v[(size_t) i*nd + k];
    in >> s.m[i] >> s.h[i] >> s.rho[i] >> s.u[i];
    s.iorig[i] = i;
  }
  if (!in) throw GandalfError("Truncated column snapshot : " + filename);
}

// ---- SEREN unformatted ------------------------------------------------------------------------------
template <class T> static void put(std::ofstream &o, T v) { o.write(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <class T> static T get(std::ifstream &f) { T v; f.read(reinterpret_cast<char*>(&v), sizeof(T)); return v; }
static void put_str(std::ofstream &o, const std::string &x)
{
  char buf[kStr];
  std::memset(buf, ' ', kStr);
  std::memcpy(buf, x.data(), x.size() < (size_t) kStr ? x.size() : (size_t) kStr);
  o.write(buf, kStr);
}
static std::string get_str(std::ifstream &f)
{
  char buf[kStr];
  f.read(buf, kStr);
  std::string x(buf, kStr);
  const size_t e = x.find_last_not_of(' ');
  return e == std::string::npos ? std::string() : x.substr(0, e + 1);
}

void WriteSerenUnformSnapshotFile(const std::string &filename, const Snapshot &s)
{
  std::ofstream out(filename.c_str(), std::ios::binary);
  if (!out) throw GandalfError("Cannot open snapshot file for writing : " + filename);
  const int nd = s.ndim, N = s.N;
  // data arrays: id, {width, 1, count, type code, unit id}   (SimulationIO.hpp:2078-2122)
  struct Arr { const char *id; int t[5]; };
  const Arr arrs[7] = {{"porig", {1, 1, N, 2, 0}}, {"r", {nd, 1, N, 4, 1}}, {"m", {1, 1, N, 4, 2}}, {"h", {1, 1, N, 4, 1}},
                       {"v", {nd, 1, N, 4, 4}}, {"rho", {1, 1, N, 4, 6}}, {"u", {1, 1, N, 4, 20}}};
  const int ndata = N > 0 ? 7 : 0;
  int32_t idata[50] = {0}; int64_t ilpdata[50] = {0}; double rdata[50] = {0.0}, ddata[50] = {0.0};
  idata[0] = N; idata[1] = 0;               // Nhydro (live), Nstar
  idata[4] = N;                             // particles per type: icm, GAS, cdm, dust (:2151-2156)
  idata[19] = (int32_t) s.units.size(); idata[20] = ndata;      // unit strings: none when dimensionless, 21 otherwise (:2066-2090)
  ilpdata[0] = s.Noutsnap; ilpdata[1] = s.Nsteps; ilpdata[10] = s.Noutlitesnap;
  rdata[0] = s.h_fac;
  ddata[0] = s.t; ddata[1] = s.tsnaplast; ddata[2] = s.mmean; ddata[10] = s.tlitesnaplast;
  put_str(out, kSerenTag);
  put<int32_t>(out, 8);                     // double precision build
  for (int k = 0; k < 3; k++) put<int32_t>(out, nd);
  for (int i = 0; i < 50; i++) put<int32_t>(out, idata[i]);
  for (int i = 0; i < 50; i++) put<int64_t>(out, ilpdata[i]);
  for (int i = 0; i < 50; i++) put<double>(out, rdata[i]);
  for (int i = 0; i < 50; i++) put<double>(out, ddata[i]);
  for (size_t i = 0; i < s.units.size(); i++) put_str(out, s.units[i]);
  for (int a = 0; a < ndata; a++) put_str(out, arrs[a].id);
  for (int a = 0; a < ndata; a++) for (int j = 0; j < 5; j++) put<int32_t>(out, arrs[a].t[j]);
  if (N > 0) {
    for (int i = 0; i < N; i++) put<int32_t>(out, s.iorig.empty() ? i : s.iorig[i]);
    out.write(reinterpret_cast<const char*>(s.r.data()), sizeof(double)*(size_t) N*nd);
    out.write(reinterpret_cast<const char*>(s.m.data()), sizeof(double)*(size_t) N);
    out.write(reinterpret_cast<const char*>(s.h.data()), sizeof(double)*(size_t) N);
    out.write(reinterpret_cast<const char*>(s.v.data()), sizeof(double)*(size_t) N*nd);
    out.write(reinterpret_cast<const char*>(s.rho.data()), sizeof(double)*(size_t) N);
    out.write(reinterpret_cast<const char*>(s.u.data()), sizeof(double)*(size_t) N);
  }
}

void ReadSerenUnformSnapshotFile(const std::string &filename, Snapshot &s)
{
  std::ifstream in(filename.c_str(), std::ios::binary);
  if (!in) throw GandalfError("Cannot open snapshot file : " + filename);
  if (get_str(in) != kSerenTag) throw GandalfError("Incorrect format of IC file : " + filename);
  const int prec = get<int32_t>(in);
  if (prec != 8) throw GandalfError("Incorrect precision in snapshot (single-precision files are not read) : " + filename);
  const int nd = get<int32_t>(in);
  get<int32_t>(in); get<int32_t>(in);
  if (nd < 1 || nd > 3) throw GandalfError("Incorrect no. of dimensions in file : " + filename);
  int32_t idata[50]; int64_t ilpdata[50]; double rdata[50], ddata[50];
  for (int i = 0; i < 50; i++) idata[i] = get<int32_t>(in);
  for (int i = 0; i < 50; i++) ilpdata[i] = get<int64_t>(in);
  for (int i = 0; i < 50; i++) rdata[i] = get<double>(in);
  for (int i = 0; i < 50; i++) ddata[i] = get<double>(in);
  const int N = idata[0], Nstar = idata[1], nunit = idata[19], ndata = idata[20];
  if (Nstar != 0) throw GandalfError("snapshots with stars / sinks are not read on this path : " + filename);
  // the header comes from a file: bound everything that sizes a buffer before using it
  if (N < 0 || ndata < 0 || ndata > 50 || nunit < 0 || nunit > 50) throw GandalfError("Corrupt snapshot header : " + filename);
  for (int i = 0; i < nunit; i++) get_str(in);
  std::vector<std::string> ids(ndata);
  for (int a = 0; a < ndata; a++) ids[a] = get_str(in);
  std::vector<int> typ((size_t) ndata*5);
  for (int a = 0; a < ndata; a++) for (int j = 0; j < 5; j++) typ[(size_t) a*5 + j] = get<int32_t>(in);
  s.ndim = nd; s.N = N; s.t = ddata[0]; s.tsnaplast = ddata[1]; s.mmean = ddata[2]; s.tlitesnaplast = ddata[10];
  s.h_fac = rdata[0]; s.Noutsnap = ilpdata[0]; s.Nsteps = ilpdata[1]; s.Noutlitesnap = ilpdata[10];
  s.r.assign((size_t) N*nd, 0.0); s.v.assign((size_t) N*nd, 0.0);
  s.m.assign(N, 0.0); s.h.assign(N, 0.0); s.rho.assign(N, 0.0); s.u.assign(N, 0.0); s.iorig.assign(N, 0);
  for (int a = 0; a < ndata; a++) {
    const int width = typ[(size_t) a*5], count = typ[(size_t) a*5 + 2], code = typ[(size_t) a*5 + 3];
    if (width < 0 || count < 0) throw GandalfError("Corrupt array descriptor in snapshot : " + ids[a]);
    std::vector<double> *dst = nullptr;
    if (ids[a] == "r") dst = &s.r; else if (ids[a] == "v") dst = &s.v; else if (ids[a] == "m") dst = &s.m;
    else if (ids[a] == "h") dst = &s.h; else if (ids[a] == "rho") dst = &s.rho; else if (ids[a] == "u") dst = &s.u;
    if (ids[a] == "porig" && count == N) in.read(reinterpret_cast<char*>(s.iorig.data()), sizeof(int32_t)*(size_t) N);
    else if (dst && code == 4 && count == N) {
      // the declared width must be the one the destination was sized for (ndim for r and v, 1 for the scalars)
      if ((size_t) N*width != dst->size()) throw GandalfError("Array '" + ids[a] + "' has the wrong width in snapshot : " + filename);
      in.read(reinterpret_cast<char*>(dst->data()), sizeof(double)*(size_t) N*width);
    }
    else {                                   // an array this path does not use: skip it by its declared size
      const size_t el = code == 2 ? 4 : (code == 4 ? 8 : (code == 3 ? 8 : 0));
      if (!el) throw GandalfError("unknown array type in snapshot : " + ids[a]);
      in.seekg((std::streamoff) (el*(size_t) width*count), std::ios::cur);
    }
  }
  if (!in) throw GandalfError("Truncated snapshot : " + filename);
}

// ---- SEREN formatted (ASCII) --------------------------------------------------------------------------
// The same header and arrays as the unformatted file, as text (SimulationIO.hpp:993-1232 write, :601-925 read): scientific
// notation, floats 18 wide with 10 decimals, integers 2 wide for the five words that open the file and 10 wide after them
// (formatted_output.h), one header word / one particle per line.  The tag is "SERENASCIIDUMPV2", the precision word 4
// whatever the build, the header's time words are written as they are (dimensionless runs: outscale = 1).
static const char *kSerenAsciiTag = "SERENASCIIDUMPV2";

void WriteSerenFormSnapshotFile(const std::string &filename, const Snapshot &s)
{
  std::ofstream out(filename.c_str());
  if (!out) throw GandalfError("Cannot open snapshot file for writing : " + filename);
  out.setf(std::ios::scientific, std::ios::floatfield);
  const int nd = s.ndim, N = s.N;
  int wi = 2;
  auto pi = [&](long long v) { out << std::setw(wi) << v; };
  auto pf = [&](double v) { out << std::setw(18) << std::setprecision(10) << v; };
  struct Arr { const char *id; int t[5]; };
  const Arr arrs[7] = {{"porig", {1, 1, N, 2, 0}}, {"r", {nd, 1, N, 4, 1}}, {"m", {1, 1, N, 4, 2}}, {"h", {1, 1, N, 4, 1}},
                       {"v", {nd, 1, N, 4, 4}}, {"rho", {1, 1, N, 4, 6}}, {"u", {1, 1, N, 4, 20}}};
  const int ndata = N > 0 ? 7 : 0;
  int idata[50] = {0}, ilpdata[50] = {0}; double rdata[50] = {0.0}, ddata[50] = {0.0};
  idata[0] = N; idata[1] = 0; idata[4] = N; idata[19] = (int) s.units.size(); idata[20] = ndata;
  ilpdata[0] = s.Noutsnap; ilpdata[1] = s.Nsteps; ilpdata[10] = s.Noutlitesnap;       // (the reference's ilpdata are ints here)
  rdata[0] = s.h_fac;
  ddata[0] = s.t; ddata[1] = s.tsnaplast; ddata[2] = s.mmean; ddata[10] = s.tlitesnaplast;
  out << kSerenAsciiTag << std::endl;
  pi(4); out << std::endl;
  for (int k = 0; k < 3; k++) { pi(nd); out << std::endl; }
  wi = 10;
  for (int i = 0; i < 50; i++) { pi(idata[i]); out << std::endl; }
  for (int i = 0; i < 50; i++) { pi(ilpdata[i]); out << std::endl; }
  for (int i = 0; i < 50; i++) { pf(rdata[i]); out << std::endl; }
  for (int i = 0; i < 50; i++) { pf(ddata[i]); out << std::endl; }
  for (size_t i = 0; i < s.units.size(); i++) out << s.units[i] << std::endl;
  for (int a = 0; a < ndata; a++) out << arrs[a].id << std::endl;
  for (int a = 0; a < ndata; a++) { for (int j = 0; j < 5; j++) pi(arrs[a].t[j]); out << std::endl; }
  if (N > 0) {
    for (int i = 0; i < N; i++) { pi(s.iorig.empty() ? i : s.iorig[i]); out << std::endl; }
    auto vec = [&](const std::vector<double> &a) { for (int i = 0; i < N; i++) { for (int k = 0; k < nd; k++) pf(a[(size_t) i*nd + k]); out << std::endl; } };
    auto sca = [&](const std::vector<double> &a) { for (int i = 0; i < N; i++) { pf(a[i]); out << std::endl; } };
    vec(s.r); sca(s.m); sca(s.h); vec(s.v); sca(s.rho); sca(s.u);
  }
}

void ReadSerenFormSnapshotFile(const std::string &filename, Snapshot &s)
{
  std::ifstream in(filename.c_str());
  if (!in) throw GandalfError("Cannot open snapshot file : " + filename);
  std::string tag;
  in >> tag;
  if (tag != kSerenAsciiTag && tag != "SERENASCIIDUMPV3") throw GandalfError("Incorrect format of snapshot file " + filename + ": " + tag);
  int prec = 0, nd = 0, nd2 = 0, nd3 = 0;
  in >> prec >> nd >> nd2 >> nd3;
  if (!in || nd < 1 || nd > 3 || nd2 != nd || nd3 != nd) throw GandalfError("Incorrect no. of dimensions in file : " + filename);
  long long idata[50], ilpdata[50]; double rdata[50], ddata[50];
  for (int i = 0; i < 50; i++) in >> idata[i];
  for (int i = 0; i < 50; i++) in >> ilpdata[i];
  for (int i = 0; i < 50; i++) in >> rdata[i];
  for (int i = 0; i < 50; i++) in >> ddata[i];
  if (!in) throw GandalfError("Truncated snapshot header : " + filename);
  const long long N = idata[0], Nstar = idata[1], nunit = idata[19], ndata = idata[20];
  if (Nstar != 0) throw GandalfError("snapshots with stars / sinks are not read on this path : " + filename);
  if (N < 0 || N > 0x7fffffff || ndata < 0 || ndata > 50 || nunit < 0 || nunit > 50) throw GandalfError("Corrupt snapshot header : " + filename);
  std::string word;
  for (int i = 0; i < nunit; i++) in >> word;
  std::vector<std::string> ids((size_t) ndata);
  for (int a = 0; a < ndata; a++) in >> ids[a];
  std::vector<long long> typ((size_t) ndata*5);
  for (int a = 0; a < ndata; a++) for (int j = 0; j < 5; j++) in >> typ[(size_t) a*5 + j];
  if (!in) throw GandalfError("Truncated snapshot header : " + filename);
  s.ndim = nd; s.N = (int) N; s.t = ddata[0]; s.tsnaplast = ddata[1]; s.mmean = ddata[2]; s.tlitesnaplast = ddata[10];
  s.h_fac = rdata[0]; s.Noutsnap = ilpdata[0]; s.Nsteps = ilpdata[1]; s.Noutlitesnap = ilpdata[10];
  s.r.assign((size_t) N*nd, 0.0); s.v.assign((size_t) N*nd, 0.0);
  s.m.assign(N, 0.0); s.h.assign(N, 0.0); s.rho.assign(N, 0.0); s.u.assign(N, 0.0); s.iorig.assign(N, 0);
  for (int a = 0; a < ndata; a++) {
    const long long width = typ[(size_t) a*5], ifirst = typ[(size_t) a*5 + 1], ilast = typ[(size_t) a*5 + 2];
    std::vector<double> *dst = nullptr;
    if (ids[a] == "r") dst = &s.r; else if (ids[a] == "v") dst = &s.v; else if (ids[a] == "m") dst = &s.m;
    else if (ids[a] == "h") dst = &s.h; else if (ids[a] == "rho") dst = &s.rho; else if (ids[a] == "u") dst = &s.u;
    if (ids[a] == "porig") { for (long long i = 0; i < N; i++) in >> s.iorig[i]; }
    else if (dst) { for (size_t q = 0; q < dst->size(); q++) in >> (*dst)[q]; }      // N x ndim (r, v) or N values, as the reference reads them
    else if (width >= 1 && ilast >= ifirst - 1 && ilast - ifirst < 0x7fffffff) {     // an array this path does not use: skip its words
      for (long long i = ifirst - 1; i < ilast; i++) for (long long k = 0; k < width; k++) in >> word;
    }
    else throw GandalfError("Corrupt array descriptor in snapshot : " + ids[a]);
    if (!in) throw GandalfError("Truncated snapshot : " + filename);
  }
}

void WriteSnapshotFile(const std::string &filename, const std::string &fileform, const Snapshot &s)
{
  if (fileform == "column") WriteColumnSnapshotFile(filename, s);
  else if (fileform == "su" || fileform == "seren_unform") WriteSerenUnformSnapshotFile(filename, s);
  else if (fileform == "sf" || fileform == "seren_form") WriteSerenFormSnapshotFile(filename, s);
  else throw GandalfError("Unrecognised file format : " + fileform + " (built: column, su, sf)");
}

void ReadSnapshotFile(const std::string &filename, const std::string &fileform, Snapshot &s)
{
  if (fileform == "column") ReadColumnSnapshotFile(filename, s);
  else if (fileform == "su" || fileform == "seren_unform") ReadSerenUnformSnapshotFile(filename, s);
  else if (fileform == "sf" || fileform == "seren_form") ReadSerenFormSnapshotFile(filename, s);
  else throw GandalfError("Unrecognised file format : " + fileform + " (built: column, su, sf)");
}
