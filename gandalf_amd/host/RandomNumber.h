// RandomNumber.h -- the reference's default generator (rand_algorithm = xorshift, randseed = 1):
// 64-bit xorshift (21,35,4) followed by an MLCG multiply, ten warm-up draws at construction
// (reference src/Headers/RandomNumber.h:71-109, src/Common/RandomNumber.cpp).  Needed bit for bit:
// it defines the synthetic inputs of the benchmark configs.
#pragma once
#include <cstdint>

class XorshiftRand {
 public:
  explicit XorshiftRand(uint64_t seed) : x(seed) { for (int k = 0; k < 10; k++) xorshiftrand(); }
  inline uint64_t xorshiftrand()
  {
    x ^= x >> 21;
    x ^= x << 35;
    x ^= x >> 4;
    return x*4768777513237032717ull;
  }
  inline double floatrand() { return invrandmax*(double) xorshiftrand(); }
  uint64_t x;
 private:
  static constexpr double invrandmax = 1.0/1.84467440737095e19;
};
