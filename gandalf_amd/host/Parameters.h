// Parameters.h -- host-side mirror of the reference's Parameters class for the hot-path keys.
//
// Same three public maps, same file grammar ("Comment : key = value", '#' comments) and the same
// defaults as the reference for every key the SPH + tree-gravity path reads
// (reference src/Headers/Parameters.h, src/Common/Parameters.cpp:75-152 grammar, :160-520 defaults).
// Keys of subsystems that are out of scope (units, radiation, dust, MFV, ...) are accepted and stored
// but have no default here.
#pragma once
#include <map>
#include <string>

class Parameters {
 public:
  Parameters() { SetDefaultValues(); }
  void ReadParamsFile(const std::string &filename);          // throws std::runtime_error
  void ParseLine(std::string line);
  void SetParameter(const std::string &key, const std::string &value);
  void SetDefaultValues();
  std::string GetParameter(const std::string &key) const;

  std::map<std::string, int> intparams;
  std::map<std::string, double> floatparams;
  std::map<std::string, std::string> stringparams;
};
