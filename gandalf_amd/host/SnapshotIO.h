// SnapshotIO.h -- GANDALF's `column`, SEREN unformatted (`su`) and SEREN formatted (`sf`) snapshot formats for the gas particles of the hot
// path, so that runs can start from and be compared with the reference's own files.
//
// File layouts restated from the reference's writers / readers:
//   column : SimulationIO.hpp:193-266 (read), :444-540 (write)
//   su     : SimulationIO.hpp:1244-1650 (read), :2009-2254 (write); tag "SERENBINARYDUMPV3", 20-byte strings
// Dimensionless units only (all out-scales are 1); stars / sinks are not on this path (Nstar = 0).
#pragma once
#include <string>
#include <vector>

struct Snapshot {
  int ndim = 3, N = 0;
  double t = 0.0;
  std::vector<double> r, v;                 // [N][ndim]
  std::vector<double> m, h, rho, u;         // [N]
  std::vector<int> iorig;                   // [N]   (su only: "porig")
  std::vector<std::string> units;           // su / sf: the 21 unit ids of a run with physical units (none: dimensionless)
  // su header words the reference fills from its Simulation object (SimulationIO.hpp:2151-2168)
  long Noutsnap = 0, Nsteps = 0, Noutlitesnap = 0;
  double h_fac = 1.2, tsnaplast = 0.0, mmean = 0.0, tlitesnaplast = 0.0;
};

void WriteColumnSnapshotFile(const std::string &filename, const Snapshot &s);
void ReadColumnSnapshotFile(const std::string &filename, Snapshot &s);
void WriteSerenUnformSnapshotFile(const std::string &filename, const Snapshot &s);
void ReadSerenUnformSnapshotFile(const std::string &filename, Snapshot &s);
//   sf     : SimulationIO.hpp:601-925 (read), :993-1232 (write); tag "SERENASCIIDUMPV2", the su layout as text (10 decimals)
void WriteSerenFormSnapshotFile(const std::string &filename, const Snapshot &s);
void ReadSerenFormSnapshotFile(const std::string &filename, Snapshot &s);
// fileform = column | su | seren_unform | sf | seren_form   (SimulationBase::Read/WriteSnapshotFile, SimulationIO.hpp:59-125)
void WriteSnapshotFile(const std::string &filename, const std::string &fileform, const Snapshot &s);
void ReadSnapshotFile(const std::string &filename, const std::string &fileform, Snapshot &s);
