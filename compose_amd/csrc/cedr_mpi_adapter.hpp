// Opt-in MPI adapter: the MPI-typed half of reference cedr/cedr_mpi.hpp (:17-85) for host
// applications that keep their MPI communicator -- E3SM's SL transport calls
// `cedr::mpi::make_parallel(comm)` and hands `p->comm()` to MPI. Include this header (after
// <mpi.h>) in the caller's translation units; libcedr_hip.so itself never links MPI.
//
//   Parallel::Ptr make_parallel(MPI_Comm)      cedr_mpi.hpp:37, cedr_mpi.cpp:10-12
//   MPI_Comm c = p->comm();                    cedr_mpi.hpp:22   (through CommRef / NativeComm)
//   Request, get_type<T>, reduce, all_reduce, isend, irecv, waitany, waitall, gather, gatherv,
//   all_ok, Op                                 cedr_mpi.hpp:29-85, cedr_mpi_inl.hpp:10-66
//
// The CDRs' own collectives (one all-gather per QLT::run -- or, for partial trees, one round of
// isend/irecv per tree level -- and one all-reduce per CAAS::run) go through MpiComm below: MPI called on DEVICE pointers (GPU-aware MPI, what the reference requires too:
// cedr/cedr_test_mpi_device_ptr.cpp) after a stream synchronisation -- or, with
// CEDR_MPI_HOST_STAGED defined (or MpiComm::host_staged = true), staged through host buffers.
// For 8 GPUs of one node the built-in RCCL communicator (make_parallel_rccl) is the fast path;
// this adapter is for drop-in use and for multi-node runs.
#ifndef COMPOSE_AMD_CEDR_MPI_ADAPTER_HPP
#define COMPOSE_AMD_CEDR_MPI_ADAPTER_HPP

#include <mpi.h>

#include <hip/hip_runtime_api.h>

#include <vector>

#include "cedr_mpi.hpp"

namespace cedr {
namespace mpi {

struct MpiComm : public Comm {
  MPI_Comm comm;
  bool host_staged;
  std::vector<Real> hs, hr, hx;
  explicit MpiComm (MPI_Comm c)
    : comm(c),
#ifdef CEDR_MPI_HOST_STAGED
      host_staged(true)
#else
      host_staged(false)
#endif
  {}
  int rank () const override { int r = 0; MPI_Comm_rank(comm, &r); return r; }
  int size () const override { int s = 0; MPI_Comm_size(comm, &s); return s; }
  const void* native_handle () const override { return &comm; }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    const int n = static_cast<int>(count);
    if (host_staged) {
      hs.resize(count); hr.resize(count*size());
      if (hipMemcpyAsync(hs.data(), send, count*sizeof(Real), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) return -1;
      const int e = MPI_Allgather(hs.data(), n, MPI_DOUBLE, hr.data(), n, MPI_DOUBLE, comm);
      if (e != MPI_SUCCESS) return e;
      return hipMemcpyAsync(recv, hr.data(), hr.size()*sizeof(Real), hipMemcpyHostToDevice, s) == hipSuccess &&
        hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return -1;  // MPI is not stream ordered
    return MPI_Allgather(const_cast<Real*>(send), n, MPI_DOUBLE, recv, n, MPI_DOUBLE, comm);
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    const int n = static_cast<int>(count);
    if (host_staged) {
      hs.resize(count); hr.resize(count);
      if (hipMemcpyAsync(hs.data(), send, count*sizeof(Real), hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) return -1;
      const int e = MPI_Allreduce(hs.data(), hr.data(), n, MPI_DOUBLE, MPI_SUM, comm);
      if (e != MPI_SUCCESS) return e;
      return hipMemcpyAsync(recv, hr.data(), count*sizeof(Real), hipMemcpyHostToDevice, s) == hipSuccess &&
        hipStreamSynchronize(s) == hipSuccess ? 0 : -1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    if (send == recv) return MPI_Allreduce(MPI_IN_PLACE, recv, n, MPI_DOUBLE, MPI_SUM, comm);
    return MPI_Allreduce(const_cast<Real*>(send), recv, n, MPI_DOUBLE, MPI_SUM, comm);
  }
  // One tree level's messages of a level-mode plan (partial trees): the reference's isend / irecv /
  // waitall, cedr_qlt.cpp:327-337,432-439, with its tag.
  bool has_exchange () const override { return true; }
  int exchange (const P2P* ops, int nops, hipStream_t s) override {
    if (nops == 0) return 0;
    enum { tag = 42 };
    std::vector<MPI_Request> reqs(nops);
    std::vector<size_t> off(nops + 1, 0);
    for (int i = 0; i < nops; ++i) off[i+1] = off[i] + ops[i].count;
    if (host_staged) {
      hx.resize(off[nops]);
      for (int i = 0; i < nops; ++i)
        if (ops[i].send && hipMemcpyAsync(hx.data() + off[i], ops[i].buf, ops[i].count*sizeof(Real),
                                          hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    for (int i = 0; i < nops; ++i) {
      Real* const b = host_staged ? hx.data() + off[i] : ops[i].buf;
      const int n = static_cast<int>(ops[i].count);
      const int e = ops[i].send ? MPI_Isend(b, n, MPI_DOUBLE, ops[i].peer, tag, comm, &reqs[i])
                                : MPI_Irecv(b, n, MPI_DOUBLE, ops[i].peer, tag, comm, &reqs[i]);
      if (e != MPI_SUCCESS) return e;
    }
    const int e = MPI_Waitall(nops, reqs.data(), MPI_STATUSES_IGNORE);
    if (e != MPI_SUCCESS) return e;
    if (host_staged) {
      for (int i = 0; i < nops; ++i)
        if ( ! ops[i].send && hipMemcpyAsync(ops[i].buf, hx.data() + off[i], ops[i].count*sizeof(Real),
                                             hipMemcpyHostToDevice, s) != hipSuccess) return -1;
      if (hipStreamSynchronize(s) != hipSuccess) return -1;
    }
    return 0;
  }
};

// `MPI_Comm c = p->comm();`: any Parallel made by make_parallel(MPI_Comm) hands its communicator
// back; one made over another transport (RCCL, callbacks, self) has none: MPI_COMM_NULL.
template <> struct NativeComm<MPI_Comm> {
  static MPI_Comm get (const Comm& c) {
    const void* h = c.native_handle();
    return h ? *static_cast<const MPI_Comm*>(h) : MPI_COMM_NULL;
  }
};

inline Parallel::Ptr make_parallel (MPI_Comm comm) {
  return make_parallel(std::static_pointer_cast<Comm>(std::make_shared<MpiComm>(comm)));
}

// ---- typed wrappers (host-side API; the buffers are whatever the MPI library accepts)
struct Request { MPI_Request request; };

template <typename T> MPI_Datatype get_type();
template <> inline MPI_Datatype get_type<int> () { return MPI_INT; }
template <> inline MPI_Datatype get_type<double> () { return MPI_DOUBLE; }
// (the reference maps long to MPI_LONG_INT, the (long, int) pair type -- cedr_mpi.cpp:46, reached
// only by its -w writer; this adapter uses the type that matches)
template <> inline MPI_Datatype get_type<long> () { return MPI_LONG; }

template <typename T>
int reduce (const Parallel& p, const T* sendbuf, T* rcvbuf, int count, MPI_Op op, int root) {
  return MPI_Reduce(const_cast<T*>(sendbuf), rcvbuf, count, get_type<T>(), op, root, p.comm());
}

template <typename T>
int all_reduce (const Parallel& p, const T* sendbuf, T* rcvbuf, int count, MPI_Op op) {
  return MPI_Allreduce(const_cast<T*>(sendbuf), rcvbuf, count, get_type<T>(), op, p.comm());
}

// ireq == nullptr: the request is freed at once (fire and forget), as cedr_mpi_inl.hpp:24-36.
template <typename T>
int isend (const Parallel& p, const T* buf, int count, int dest, int tag, Request* ireq = nullptr) {
  MPI_Request ureq;
  MPI_Request* const req = ireq ? &ireq->request : &ureq;
  const int ret = MPI_Isend(const_cast<T*>(buf), count, get_type<T>(), dest, tag, p.comm(), req);
  if ( ! ireq) MPI_Request_free(req);
  return ret;
}

template <typename T>
int irecv (const Parallel& p, T* buf, int count, int src, int tag, Request* ireq = nullptr) {
  MPI_Request ureq;
  MPI_Request* const req = ireq ? &ireq->request : &ureq;
  const int ret = MPI_Irecv(buf, count, get_type<T>(), src, tag, p.comm(), req);
  if ( ! ireq) MPI_Request_free(req);
  return ret;
}

inline int waitany (int count, Request* reqs, int* index, MPI_Status* stats = nullptr) {
  static_assert(sizeof(Request) == sizeof(MPI_Request), "Request must be layout compatible with MPI_Request");
  return MPI_Waitany(count, reinterpret_cast<MPI_Request*>(reqs), index, stats ? stats : MPI_STATUS_IGNORE);
}

inline int waitall (int count, Request* reqs, MPI_Status* stats = nullptr) {
  return MPI_Waitall(count, reinterpret_cast<MPI_Request*>(reqs), stats ? stats : MPI_STATUSES_IGNORE);
}

template <typename T>
int gather (const Parallel& p, const T* sendbuf, int sendcount, T* recvbuf, int recvcount, int root) {
  return MPI_Gather(const_cast<T*>(sendbuf), sendcount, get_type<T>(), recvbuf, recvcount, get_type<T>(), root,
                    p.comm());
}

template <typename T>
int gatherv (const Parallel& p, const T* sendbuf, int sendcount, T* recvbuf, const int* recvcounts,
             const int* displs, int root) {
  return MPI_Gatherv(const_cast<T*>(sendbuf), sendcount, get_type<T>(), recvbuf, const_cast<int*>(recvcounts),
                     const_cast<int*>(displs), get_type<T>(), root, p.comm());
}

inline bool all_ok (const Parallel& p, bool im_ok) {
  int ok = im_ok, msg = 0;
  all_reduce<int>(p, &ok, &msg, 1, MPI_LAND);
  return msg != 0;
}

struct Op {
  typedef std::shared_ptr<Op> Ptr;
  Op (MPI_User_function* function, bool commute) { MPI_Op_create(function, static_cast<int>(commute), &op_); }
  ~Op () { MPI_Op_free(&op_); }
  const MPI_Op& get () const { return op_; }
private:
  MPI_Op op_;
};

}  // namespace mpi
}  // namespace cedr

#endif
