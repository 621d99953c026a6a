// Small RAII helpers over hipMalloc/hipMemcpy for the host classes.
#ifndef COMPOSE_AMD_CEDR_DEVICE_HPP
#define COMPOSE_AMD_CEDR_DEVICE_HPP

#include <hip/hip_runtime.h>

#include <vector>

#include "cedr_util.hpp"

namespace cedr {
namespace dev {

template <typename T>
class Buffer {
  T* p_ = nullptr;
  size_t n_ = 0;
public:
  Buffer () {}
  Buffer (const Buffer&) = delete;
  Buffer& operator= (const Buffer&) = delete;
  ~Buffer () { release(); }
  void release () { if (p_) (void) hipFree(p_); p_ = nullptr; n_ = 0; }
  void swap (Buffer& o) { T* const p = p_; p_ = o.p_; o.p_ = p; const size_t n = n_; n_ = o.n_; o.n_ = n; }
  void alloc (size_t n) {
    release();
    n_ = n;
    if (n == 0) return;
    cedr_hip_check(hipMalloc(reinterpret_cast<void**>(&p_), n*sizeof(T)));
  }
  void alloc_zero (size_t n) {
    alloc(n);
    if (n) cedr_hip_check(hipMemset(p_, 0, n*sizeof(T)));
  }
  void upload (const std::vector<T>& h) {
    alloc(h.size());
    if ( ! h.empty())
      cedr_hip_check(hipMemcpy(p_, h.data(), h.size()*sizeof(T), hipMemcpyHostToDevice));
  }
  T* data () const { return p_; }
  size_t size () const { return n_; }
};

// A CDR's work moves from stream `from` to stream `to` (set_stream): whatever the object enqueued on the old stream --
// a set_Qm's packing kernels, say -- is ordered before what it enqueues on the new one. Streams made non-blocking (every
// torch stream is) do not wait for the null stream by themselves.
inline void follow_stream (hipStream_t from, hipStream_t to) {
  if (from == to) return;
  hipEvent_t ev = nullptr;
  cedr_hip_check(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, from);
  if (e == hipSuccess) e = hipStreamWaitEvent(to, ev, 0);
  (void) hipEventDestroy(ev);  // (released once the recorded work has completed)
  cedr_hip_check(e);
}

}  // namespace dev
}  // namespace cedr

#endif
