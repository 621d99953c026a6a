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
  std::vector<int> dense_;  // a rank's ids are usually one block with few gaps: then a direct table
  long lo_ = 0;             // ... of ids lo_ .. lo_ + dense_.size() - 1
  size_t n_ = 0;            // entries added
  bool sorted_ = true;
public:
  void reserve (size_t n) { v_.reserve(n); }
  void add (int id, int idx) {
    if ( ! v_.empty() && id <= v_.back().first) sorted_ = false;
    v_.emplace_back(id, idx);
    ++n_;
  }
  void finish () {
    if (v_.empty()) return;
    long lo = v_[0].first, hi = lo;
    for (const auto& e : v_) { lo = std::min<long>(lo, e.first); hi = std::max<long>(hi, e.first); }
    if (hi - lo < 4*static_cast<long>(v_.size()) + 1024) {
      lo_ = lo;
      dense_.assign(static_cast<size_t>(hi - lo) + 1, -1);
      // (a duplicate id keeps its FIRST index, as the sorted form's lower_bound finds it)
      for (const auto& e : v_) if (dense_[e.first - lo] < 0) dense_[e.first - lo] = e.second;
      std::vector<std::pair<int,int> >().swap(v_);
      sorted_ = true;
      return;
    }
    // (stable, by id only: among duplicates of an id the first one added stays first)
    if ( ! sorted_) {
      std::stable_sort(v_.begin(), v_.end(), [] (const std::pair<int,int>& a, const std::pair<int,int>& b) {
        return a.first < b.first; });
      sorted_ = true;
    }
  }
  size_t size () const { return n_; }  // entries added (not the span of the ids)
  // -1 if absent
  int find (int id) const {
    if ( ! dense_.empty()) {
      const long k = static_cast<long>(id) - lo_;
      return (k >= 0 && static_cast<size_t>(k) < dense_.size()) ? dense_[k] : -1;
    }
    const auto it = std::lower_bound(v_.begin(), v_.end(), id,
                                     [] (const std::pair<int,int>& a, const int b) { return a.first < b; });
    return (it != v_.end() && it->first == id) ? it->second : -1;
  }
};
// Setup-phase timing for tools/setup_time.py: CEDR_SETUP_TIMES=1 prints "label seconds" lines to stderr.
class PhaseTimer {
  bool on_;
  double t0_;
  static double now ();
public:
  PhaseTimer ();
  void lap (const char* label);
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
