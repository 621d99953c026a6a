// cedr::mpi::Parallel -- the communicator handle every CDR takes, as in
// reference cedr/cedr_mpi.hpp:17-27 (Parallel: comm(), size(), rank(), root(),
// amroot()). The reference wraps an MPI_Comm and moves device pointers with
// MPI point-to-point messages per tree level (cedr_qlt.cpp:327-337,432-439).
// Here a Parallel wraps a Comm: a stream-ordered collective backend over
// DEVICE buffers. One process per GPU; the built-in backend is RCCL over xGMI.
// A callback backend lets a host application plug in its own transport (MPI
// with GPU-aware pointers, or a host-staged one) without this library linking it.
#ifndef COMPOSE_AMD_CEDR_MPI_HPP
#define COMPOSE_AMD_CEDR_MPI_HPP

#include <memory>
#include <utility>

#include "cedr.hpp"

typedef struct ihipStream_t* hipStream_t;

namespace cedr {
namespace mpi {

// Collective backend. All buffers are device memory; every call is enqueued on
// (or ordered with respect to) 'stream' and returns 0 on success.
struct Comm {
  virtual ~Comm () {}
  // The communicator of the transport underneath, if it has one an adapter can hand back
  // (cedr_mpi_adapter.hpp: the address of the MPI_Comm); nullptr otherwise.
  virtual const void* native_handle () const { return nullptr; }
  virtual int rank () const = 0;
  virtual int size () const = 0;
  // recv[r*count : (r+1)*count] = rank r's send[0:count], for every r.
  virtual int allgather (const Real* send, Real* recv, size_t count, hipStream_t stream) = 0;
  // recv[i] = sum over ranks of send[i]. send and recv may alias.
  virtual int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t stream) = 0;
  // One round of point-to-point messages (the reference's isend / irecv / waitall of one tree
  // level, cedr_qlt.cpp:327-337,432-439): every op of the round is posted together (one RCCL group;
  // MPI: nonblocking + waitall), at most one send and one receive per peer and round, and the
  // matching op is in the peer's round of the same number. Only the per-level exchange of partial
  // trees / scattered ownership needs it (plan::Plan::level_mode); a transport without it keeps
  // the default (an error if such a plan is run).
  struct P2P { int peer; Real* buf; size_t count; int send; };
  virtual int exchange (const P2P* /*ops*/, int nops, hipStream_t /*stream*/) { return nops == 0 ? 0 : -2; }
  // Whether exchange() is implemented. REQUIRED OVERRIDE for a subclass that implements exchange(): the
  // default (false) tells TreeEngine to plan every full tree for the all-gather (with a warning on stderr
  // where ownership is so scattered that the planner would have preferred per-level messages) and to refuse,
  // when the CDR is constructed and on every rank alike, what only per-level messages can do: partial
  // trees (tree::Node::level) and nodes pinned to a rank that does not own the cells below them
  // (tree::Node::rank) -- instead of failing inside run() or silently computing the node elsewhere.
  virtual bool has_exchange () const { return false; }
};

// What Parallel::comm() returns: the Comm of this build, usable as the shared_ptr it is -- and,
// when an adapter for a native communicator type is in scope (cedr_mpi_adapter.hpp defines
// NativeComm<MPI_Comm>), convertible to that type, so that reference-style code such as
// `MPI_Comm c = p->comm();` or `MPI_Barrier(p->comm())` (cedr/cedr_mpi.hpp:22) compiles unchanged.
template <typename T> struct NativeComm;  // specialised by adapters: static T get(const Comm&)
class CommRef {
  const std::shared_ptr<Comm>& c_;
public:
  explicit CommRef (const std::shared_ptr<Comm>& c) : c_(c) {}
  Comm* operator-> () const { return c_.get(); }
  Comm& operator* () const { return *c_; }
  operator const std::shared_ptr<Comm>& () const { return c_; }
  const std::shared_ptr<Comm>& ptr () const { return c_; }
  template <typename T, typename = decltype(NativeComm<T>::get(std::declval<const Comm&>()))>
  operator T () const { return NativeComm<T>::get(*c_); }
};

class Parallel {
  std::shared_ptr<Comm> comm_;
public:
  typedef std::shared_ptr<Parallel> Ptr;
  explicit Parallel (const std::shared_ptr<Comm>& comm) : comm_(comm) {}
  CommRef comm () const { return CommRef(comm_); }
  Int size () const { return comm_->size(); }
  Int rank () const { return comm_->rank(); }
  Int root () const { return 0; }
  bool amroot () const { return rank() == root(); }
};

// Single-rank communicator (the reference's MPI_COMM_SELF use case).
Parallel::Ptr make_parallel ();
Parallel::Ptr make_parallel (const std::shared_ptr<Comm>& comm);

// RCCL communicator. 'unique_id' is the 128-byte ncclUniqueId produced by
// rccl_get_unique_id on one rank and distributed by the launcher (bench.py uses
// torch.distributed for that). librccl is dlopen'ed on first use.
int rccl_get_unique_id (void* unique_id_128B);
Parallel::Ptr make_parallel_rccl (const void* unique_id_128B, int rank, int size);

// Rehearsal of an N-rank job on ONE GPU: rank 0 of 'pretended_size' ranks over 'inner', a communicator of one real
// rank (make_parallel() or make_parallel_rccl(id, 0, 1)). Every collective runs through 'inner' for real -- with
// RCCL inside: librccl's enqueue path and this library's cross-stream ordering of the shared communicator -- and the
// other ranks' blocks of an all-gather are copies of this rank's. For measurements and tests only.
Parallel::Ptr make_parallel_rehearsal (const Parallel::Ptr& inner, int pretended_size);

// PEER FLAGS (behind a knob; RCCL stays the default): a transport for the two KB-sized collectives of a step that
// uses nothing but stream-ordered writes into the peers' memory -- no communicator library, no proxy thread, no
// kernel that spins. Every rank owns a MAILBOX in fine-grained device memory (per source rank two payload slots, and
// two arrival counters), exported with hipIpcGetMemHandle and mapped by every peer of the node over xGMI. A collective
// is then, on the caller's stream: stores of this rank's block into its slot of every mailbox and, after a system-scope
// release, an add of 1 on each mailbox's counter; a wait until this rank's own counter says that all blocks have
// arrived; a copy (all-gather) or a sum in rank order (all-reduce: the same bits on every rank) of the P slots into
// recv -- as ONE launch that waits in the kernel, bounded (the default), or as a put launch, ONE hipStreamWaitValue64
// and a take launch (CEDR_HIP_PEER_WAIT=stream; the same choice on every rank). In the one-launch form send and recv
// must not overlap. Slots and counters alternate between two parities by collective number, which
// is what makes reuse safe without a second handshake (cedr_mpi.cpp). Replaces, for these two collectives, the
// reference's per-level MPI messages (cedr_qlt.cpp:327-337,432-439) and MPI_Allreduce (cedr_caas.cpp:203-209), like
// the RCCL backend. Point-to-point rounds (exchange: partial trees, scattered ownership) run over the same mailboxes
// with per-pair message slots, flags and acks, one launch per round, for up to 32 ranks.
// Setup: every rank creates its mailbox, the launcher gathers the kPeerHandleBytes-byte handles of all ranks (rank
// order) by any means it has, and make_parallel_peer_flags maps them. One node only; at most kPeerFlagsMaxRanks ranks.
struct PeerMailbox;
constexpr size_t kPeerHandleBytes = 96;
constexpr int kPeerFlagsMaxRanks = 64;
// 'slot_doubles': capacity of one payload slot; longer collectives run as several rounds.
std::shared_ptr<PeerMailbox> peer_mailbox_create (int size, size_t slot_doubles = 4096);
void peer_mailbox_handle (const PeerMailbox& mb, void* handle_96B);
Parallel::Ptr make_parallel_peer_flags (const std::shared_ptr<PeerMailbox>& mine, const void* handles_in_rank_order,
                                        int rank, int size);

// Callback communicator: the host application supplies the two collectives.
typedef int (*AllgatherFn) (void* user, const Real* send, Real* recv, size_t count, hipStream_t);
typedef int (*AllreduceSumFn) (void* user, const Real* send, Real* recv, size_t count, hipStream_t);
// Optional third callback: one round of point-to-point messages (Comm::exchange). peers / bufs /
// counts / sends are arrays of nops entries.
typedef int (*ExchangeFn) (void* user, int nops, const int* peers, Real* const* bufs, const size_t* counts,
                           const int* sends, hipStream_t);
Parallel::Ptr make_parallel_callbacks (int rank, int size, AllgatherFn ag, AllreduceSumFn ar,
                                       void* user, ExchangeFn xch = nullptr);

}  // namespace mpi
}  // namespace cedr

#endif
