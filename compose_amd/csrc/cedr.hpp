// Basic CEDR types. Mirrors reference cedr/cedr.hpp:11-39 (no Kokkos).
#ifndef COMPOSE_AMD_CEDR_HPP
#define COMPOSE_AMD_CEDR_HPP

#include <cstddef>
#include <cstdint>

namespace cedr {
typedef int Int;
typedef long int Long;
typedef std::size_t Size;
typedef double Real;

// Execution-space tag. The reference templates its classes on a Kokkos
// execution space; this build has exactly one backend, MI355X via HIP.
struct Hip {};

// reference cedr/cedr.hpp:29-39
struct ProblemType {
  enum : Int {
    conserve = 1, shapepreserve = 1 << 1, consistent = 1 << 2,
    nonnegative = 1 << 3
  };
};
}  // namespace cedr

#endif
