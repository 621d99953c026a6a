// Error conventions of reference cedr/cedr_util.hpp:53-81: cedr_throw_if is
// always on and throws std::logic_error with file:line; cedr_assert is compiled
// out under NDEBUG.
#ifndef COMPOSE_AMD_CEDR_UTIL_HPP
#define COMPOSE_AMD_CEDR_UTIL_HPP

#include <sstream>
#include <stdexcept>
#include <string>

#define cedr_throw_if(condition, message) do {                          \
    if (condition) {                                                    \
      std::stringstream _ss_;                                           \
      _ss_ << __FILE__ << ":" << __LINE__ << ": The condition:\n"       \
           << #condition "\nled to the exception\n" << message << "\n"; \
      throw std::logic_error(_ss_.str());                               \
    }                                                                   \
  } while (0)

#ifndef NDEBUG
# define cedr_assert(condition) do {                                    \
    if ( ! (condition)) {                                               \
      std::stringstream _ss_;                                           \
      _ss_ << __FILE__ << ":" << __LINE__ << ": FAIL:\n" << #condition  \
           << "\n";                                                     \
      throw std::logic_error(_ss_.str());                               \
    }                                                                   \
  } while (0)
#else
# define cedr_assert(condition)
#endif

// HIP runtime calls made by host code: failure is an exception, never ignored.
#define cedr_hip_check(call) do {                                       \
    const hipError_t _e_ = (call);                                      \
    cedr_throw_if(_e_ != hipSuccess, "HIP error " << hipGetErrorString(_e_) \
                  << " from " #call);                                   \
  } while (0)

#endif
