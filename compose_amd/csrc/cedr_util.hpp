// Error conventions of reference cedr/cedr_util.hpp:53-81: cedr_throw_if is
// always on and throws std::logic_error with file:line; cedr_assert is compiled
// out under NDEBUG.
#ifndef COMPOSE_AMD_CEDR_UTIL_HPP
#define COMPOSE_AMD_CEDR_UTIL_HPP

#include <algorithm>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

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

namespace cedr {
namespace util {
// Global cell id -> local cell index. The reference keeps a std::map
// (cedr_qlt.cpp:220-227); at 6M cells a sorted array is ~10x faster to build and
// 6x smaller, with the same lookups.
class IdMap {
  std::vector<std::pair<int,int> > v_;
  bool sorted_ = true;
public:
  void reserve (size_t n) { v_.reserve(n); }
  void add (int id, int idx) {
    if ( ! v_.empty() && id <= v_.back().first) sorted_ = false;
    v_.emplace_back(id, idx);
  }
  void finish () { if ( ! sorted_) { std::sort(v_.begin(), v_.end()); sorted_ = true; } }
  size_t size () const { return v_.size(); }
  // -1 if absent
  int find (int id) const {
    const auto it = std::lower_bound(v_.begin(), v_.end(), std::make_pair(id, -1));
    return (it != v_.end() && it->first == id) ? it->second : -1;
  }
};
}  // namespace util
}  // namespace cedr

// HIP runtime calls made by host code: failure is an exception, never ignored.
#define cedr_hip_check(call) do {                                       \
    const hipError_t _e_ = (call);                                      \
    cedr_throw_if(_e_ != hipSuccess, "HIP error " << hipGetErrorString(_e_) \
                  << " from " #call);                                   \
  } while (0)

#endif
