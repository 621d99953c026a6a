// Communicator backends behind cedr::mpi::Parallel (see cedr_mpi.hpp).
#include "cedr_mpi.hpp"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "cedr_util.hpp"

namespace cedr {
namespace mpi {
namespace {

// RehearsalComm: blocks 1 .. P-1 of an all-gather's result = copies of block 0, in ONE launch (P - 1 separate
// hipMemcpyAsync calls cost a rehearsed step ~1.5 us each, which no real all-gather would). Host-only builds of this
// file (tests/cpp/plan_check.cpp under g++ with sanitizers) have no kernels: they take the copies.
#ifdef __HIPCC__
__global__ void rehearsal_replicate_kernel (Real* __restrict__ recv, const size_t count, const int nblocks) {
  const size_t i = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
  if (i >= count) return;
  const Real v = recv[i];
  for (int r = 1; r < nblocks; ++r) recv[static_cast<size_t>(r)*count + i] = v;
}
#endif

// PeerFlagComm's kernels. A mailbox is [2 counters, padded to kMailboxHead bytes][2 parities][P sources][cap doubles].
constexpr size_t kMailboxHead = 128;
struct PeerTable { unsigned char* base[kPeerFlagsMaxRanks]; };
#ifdef __HIPCC__
// Workgroup r stores this rank's block into its slot of rank r's mailbox, then announces it: every thread's stores are
// released to system scope before thread 0 adds 1 to that mailbox's counter of this parity.
__global__ void peer_put_kernel (const PeerTable t, const Real* __restrict__ send, const size_t count,
                                 const size_t slot_off, const int parity) {
  unsigned char* const mb = t.base[blockIdx.x];
  unsigned long long* const dst = reinterpret_cast<unsigned long long*>(mb + kMailboxHead) + slot_off;
  const unsigned long long* const src = reinterpret_cast<const unsigned long long*>(send);
  for (size_t i = threadIdx.x; i < count; i += blockDim.x)
    __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(mb) + parity, 1ull, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}
// After the wait on this rank's counter: recv = the P slots one after the other (sum == 0), or their sum in rank
// order (sum == 1: ((s0 + s1) + s2) + ..., the same expression on every rank, hence the same bits).
__global__ void peer_take_kernel (const unsigned char* __restrict__ mb, Real* __restrict__ recv, const size_t count,
                                  const size_t parity_off, const size_t cap, const int nranks, const int sum,
                                  const size_t block /* all-gather: doubles between two ranks' blocks in recv */) {
  const size_t i = static_cast<size_t>(blockIdx.x)*blockDim.x + threadIdx.x;
  if (i >= count) return;
  const unsigned long long* const slots = reinterpret_cast<const unsigned long long*>(mb + kMailboxHead) + parity_off;
  Real acc = 0;
  for (int r = 0; r < nranks; ++r) {
    const unsigned long long bits = __hip_atomic_load(slots + static_cast<size_t>(r)*cap + i, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_SYSTEM);
    const Real v = __longlong_as_double(static_cast<long long>(bits));
    if (sum) acc = r == 0 ? v : acc + v;
    else recv[static_cast<size_t>(r)*block + i] = v;
  }
  if (sum) recv[i] = acc;
}
// Point-to-point rounds (Comm::exchange) over the same mailboxes: behind the collectives' region every mailbox has, per
// peer and parity, a message slot, a FLAG (number of the message in the slot, + 1) written by the sender, and an ACK
// (number of my message the peer has consumed, + 1) written back by the receiver into the SENDER's mailbox. Messages of
// one ordered pair are numbered 0, 1, 2, ... (a round has at most one per pair and direction, a longer one goes in pieces)
// and use parity = number & 1; a sender reuses a slot only after the ack of the message two before. One launch per
// round, one workgroup per op, which moves all pieces of its message in order -- every wait is for the workgroup of the
// MATCHING op in the peer's launch, all workgroups of a launch are resident (at most 2 (P - 1) of them), and the waits
// are bounded as in peer_xchg_kernel (the receive buffer is then filled with NaN).
struct PeerOp { Real* buf; unsigned long long count, first_msg; int peer, send; };
constexpr int kPeerMaxOps = 62;  // per round; ranks up to 32 (the kernel argument block is 4 KB)
struct PeerOps { PeerOp op[kPeerMaxOps]; };
__device__ inline bool peer_wait_ge (const unsigned long long* p, const unsigned long long v, const unsigned long long max_polls) {
  unsigned long long polls = 0;
  while (__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
    if (++polls > max_polls) return false;
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}
__global__ void peer_p2p_kernel (const PeerTable t, const PeerOps ops, const int me, const int nranks, const size_t cap,
                                 const size_t p2p_off /* bytes from a mailbox's start to its point-to-point region */,
                                 const unsigned long long max_polls) {
  const PeerOp o = ops.op[blockIdx.x];
  const size_t P = static_cast<size_t>(nranks);
  // region: flag[2][P], ack[2][P] (u64), then slots[2][P][cap]
  auto flag = [&] (unsigned char* mb, const int par, const int src) {
    return reinterpret_cast<unsigned long long*>(mb + p2p_off) + par*P + src; };
  auto ack = [&] (unsigned char* mb, const int par, const int dst) {
    return reinterpret_cast<unsigned long long*>(mb + p2p_off) + 2*P + par*P + dst; };
  auto slot = [&] (unsigned char* mb, const int par, const int src) {
    return reinterpret_cast<unsigned long long*>(mb + p2p_off) + 4*P + (par*P + src)*cap; };
  unsigned char* const mine = t.base[me];
  unsigned char* const theirs = t.base[o.peer];
  __shared__ int ok;
  unsigned long long* const buf = reinterpret_cast<unsigned long long*>(o.buf);
  unsigned long long msg = o.first_msg;
  for (size_t off = 0; off < o.count; off += cap, ++msg) {
    const size_t n = o.count - off < cap ? o.count - off : cap;
    const int par = static_cast<int>(msg & 1);
    if (o.send) {
      // the slot is free once the peer has consumed my message msg - 2 (its ack, in MY mailbox, is then msg - 1)
      if (threadIdx.x == 0) ok = msg < 2 || peer_wait_ge(ack(mine, par, o.peer), msg - 1, max_polls);
      __syncthreads();
      if ( ! ok) return;  // (the peer will give up as well)
      unsigned long long* const dst = slot(theirs, par, me);
      for (size_t i = threadIdx.x; i < n; i += blockDim.x)
        __hip_atomic_store(dst + i, buf[off + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0)
        __hip_atomic_store(flag(theirs, par, me), msg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      if (threadIdx.x == 0) ok = peer_wait_ge(flag(mine, par, o.peer), msg + 1, max_polls);
      __syncthreads();
      __threadfence_system();
      if ( ! ok) {
        for (size_t i = off + threadIdx.x; i < o.count; i += blockDim.x) buf[i] = 0x7ff8000000000000ull;
        if (threadIdx.x == 0)
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine) + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      const unsigned long long* const src = slot(mine, par, o.peer);
      for (size_t i = threadIdx.x; i < n; i += blockDim.x)
        buf[off + i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __syncthreads();  // every thread has read the slot before it is handed back
      if (threadIdx.x == 0)
        __hip_atomic_store(ack(theirs, par, me), msg + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
  }
}

// The three operations as ONE launch (the default form; CEDR_HIP_PEER_WAIT=stream: the three above): workgroup r stores this rank's block into rank r's
// mailbox and announces it (nothing to itself: its own block is read from `send`), then waits IN THE KERNEL for the
// P - 1 other ranks' arrivals in this rank's mailbox, then takes -- source r's block (all-gather), or the r-th share of
// the elements summed over the ranks in rank order (all-reduce; the same expression as peer_take_kernel's). The wait
// depends on the OTHER ranks' launches only, never on a workgroup of this launch; it gives up after `max_polls` polls of
// about a microsecond (2^28 by default, minutes: a peer that never arrives; CEDR_HIP_PEER_GIVE_UP_LOG2), raises the
// mailbox's error word and fills recv with NaN, so that every wave ends whatever the peers do.
__global__ void peer_xchg_kernel (const PeerTable t, const int me, const int nranks, const Real* __restrict__ send,
                                  Real* __restrict__ recv, const size_t count, const size_t cap, const int parity,
                                  const unsigned long long target, const int sum, const size_t block,
                                  const unsigned long long max_polls) {
  const int r = blockIdx.x;
  unsigned char* const mine = t.base[me];
  const unsigned long long* const src = reinterpret_cast<const unsigned long long*>(send);
  const size_t P = static_cast<size_t>(nranks);
  if (r != me) {
    unsigned char* const mb = t.base[r];
    unsigned long long* const dst = reinterpret_cast<unsigned long long*>(mb + kMailboxHead) + (parity*P + me)*cap;
    for (size_t i = threadIdx.x; i < count; i += blockDim.x)
      __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(mb) + parity, 1ull, __ATOMIC_RELEASE,
                             __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __shared__ int gave_up;
  if (threadIdx.x == 0) {
    gave_up = 0;
    const unsigned long long* const ctr = reinterpret_cast<const unsigned long long*>(mine) + parity;
    unsigned long long polls = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
      if (++polls > max_polls) { gave_up = 1; break; }
      __builtin_amdgcn_s_sleep(8);
    }
    if (gave_up)
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(mine) + 2, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  __threadfence_system();
  const bool bad = gave_up != 0;
  const unsigned long long* const slots = reinterpret_cast<const unsigned long long*>(mine + kMailboxHead) + parity*P*cap;
  auto value = [&] (const int q, const size_t i) -> Real {
    const unsigned long long bits = q == me ? src[i] :
      __hip_atomic_load(slots + static_cast<size_t>(q)*cap + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return bad ? __longlong_as_double(0x7ff8000000000000ll) : __longlong_as_double(static_cast<long long>(bits));
  };
  if ( ! sum) {
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) recv[static_cast<size_t>(r)*block + i] = value(r, i);
  } else {
    const size_t lo = count*static_cast<size_t>(r)/P, hi = count*(static_cast<size_t>(r) + 1)/P;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
      Real acc = value(0, i);
      for (int q = 1; q < nranks; ++q) acc = acc + value(q, i);
      recv[i] = acc;
    }
  }
}
#endif

// Collectives of ONE communicator must execute in the same order on every rank. Callers issue
// them in program order, but possibly on different streams (QLT and CAAS each on its own): every
// collective therefore waits for the previous one's completion event when the stream changes,
// so the order of execution is the order of issue whatever the streams are.
struct CrossStreamOrder {
  hipEvent_t done_ = nullptr;
  hipStream_t last_ = nullptr;
  bool used_ = false;
  ~CrossStreamOrder () { if (done_) (void) hipEventDestroy(done_); }
  // Both return 0 or a HIP error: a collective whose ordering could not be set up must fail, not run
  // concurrently with another one on the same communicator.
  int before (hipStream_t s) {
    if (used_ && s != last_) return static_cast<int>(hipStreamWaitEvent(s, done_, 0));
    return 0;
  }
  int after (hipStream_t s) {
    if ( ! done_) {
      const hipError_t e = hipEventCreateWithFlags(&done_, hipEventDisableTiming);
      if (e != hipSuccess) { done_ = nullptr; return static_cast<int>(e); }
    }
    const hipError_t e = hipEventRecord(done_, s);
    if (e != hipSuccess) return static_cast<int>(e);
    last_ = s;
    used_ = true;
    return 0;
  }
};

// ------------------------------------------------------------------ single rank
struct SelfComm : public Comm {
  int rank () const override { return 0; }
  int size () const override { return 1; }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    if (send == recv || count == 0) return 0;
    return static_cast<int>(hipMemcpyAsync(recv, send, count*sizeof(Real),
                                           hipMemcpyDeviceToDevice, s));
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return allgather(send, recv, count, s);
  }
  bool has_exchange () const override { return true; }  // no peers: there is never a message
};

// ------------------------------------------------------------------------ RCCL
// Resolved at run time so that the library loads on machines without RCCL and
// shares whichever librccl the process already has (PyTorch ships its own).
struct ncclUniqueIdBlob { char internal[128]; };
typedef void* ncclComm_t;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat64 = 8 };

struct RcclApi {
  int (*GetUniqueId)(ncclUniqueIdBlob*);
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueIdBlob, int);
  int (*CommDestroy)(ncclComm_t);
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t);
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t);
  int (*GroupStart)();
  int (*GroupEnd)();
  bool ok, p2p;
};

const RcclApi& rccl () {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] () {
    api.ok = api.p2p = false;
    // CEDR_HIP_RCCL_LIB: the library to bind instead (a site's own librccl build; the tests' double).
    const char* const given = std::getenv("CEDR_HIP_RCCL_LIB");
    void* h = (given && *given) ? dlopen(given, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (given && *given && ! h) return;  // named but not loadable: an error, not a silent fallback
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // RTLD_LOCAL: a process that later loads ANOTHER librccl (PyTorch bundles its own) must not
    // have that copy's symbols resolved against this one (seen as a double free at exit).
    for (size_t i = 0; ! h && i < sizeof(names)/sizeof(*names); ++i)
      h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if ( ! h) return;
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(h, "ncclAllGather"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
    api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(h, "ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(h, "ncclRecv"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(dlsym(h, "ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather &&
      api.AllReduce;
    api.p2p = api.Send && api.Recv && api.GroupStart && api.GroupEnd;
  });
  return api;
}

struct RcclComm : public Comm {
  ncclComm_t comm_;
  int rank_, size_;
  CrossStreamOrder ord_;  // collectives issued on different streams still execute in the order of issue
  int order_before (hipStream_t s) { return ord_.before(s); }
  int order_after (hipStream_t s) { return ord_.after(s); }
  RcclComm (const void* id, int rank, int size) : comm_(nullptr), rank_(rank), size_(size) {
    const RcclApi& a = rccl();
    cedr_throw_if( ! a.ok, "librccl could not be loaded");
    ncclUniqueIdBlob blob;
    std::memcpy(&blob, id, sizeof(blob));
    const int e = a.CommInitRank(&comm_, size, blob, rank);
    cedr_throw_if(e != kNcclSuccess, "ncclCommInitRank returned " << e);
  }
  ~RcclComm () override {
    if (comm_) rccl().CommDestroy(comm_);
  }
  int rank () const override { return rank_; }
  int size () const override { return size_; }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    if (const int eo = order_before(s)) return eo;
    const int e = rccl().AllGather(send, recv, count, kNcclFloat64, comm_, s);
    const int ea = order_after(s);
    return e != kNcclSuccess ? e : ea;
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    if (const int eo = order_before(s)) return eo;
    const int e = rccl().AllReduce(send, recv, count, kNcclFloat64, kNcclSum, comm_, s);
    const int ea = order_after(s);
    return e != kNcclSuccess ? e : ea;
  }
  bool has_exchange () const override { return rccl().p2p; }
  int exchange (const P2P* ops, int nops, hipStream_t s) override {
    if (nops == 0) return 0;
    const RcclApi& a = rccl();
    if ( ! a.p2p) return -2;
    if (const int eo = order_before(s)) return eo;
    int e = a.GroupStart();
    for (int i = 0; i < nops && e == kNcclSuccess; ++i)
      e = ops[i].send ? a.Send(ops[i].buf, ops[i].count, kNcclFloat64, ops[i].peer, comm_, s)
                      : a.Recv(ops[i].buf, ops[i].count, kNcclFloat64, ops[i].peer, comm_, s);
    const int e2 = a.GroupEnd();
    const int ea = order_after(s);
    return e != kNcclSuccess ? e : (e2 != kNcclSuccess ? e2 : ea);
  }
};

// -------------------------------------------------------------------- callbacks
struct CallbackComm : public Comm {
  int rank_, size_;
  AllgatherFn ag_;
  AllreduceSumFn ar_;
  void* user_;
  ExchangeFn xch_;
  CallbackComm (int r, int s, AllgatherFn ag, AllreduceSumFn ar, void* u, ExchangeFn xch)
    : rank_(r), size_(s), ag_(ag), ar_(ar), user_(u), xch_(xch) {}
  int rank () const override { return rank_; }
  int size () const override { return size_; }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return ag_(user_, send, recv, count, s);
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return ar_(user_, send, recv, count, s);
  }
  bool has_exchange () const override { return xch_ != nullptr; }
  int exchange (const P2P* ops, int nops, hipStream_t s) override {
    if (nops == 0) return 0;
    if ( ! xch_) return -2;
    std::vector<int> peers(nops), sends(nops);
    std::vector<Real*> bufs(nops);
    std::vector<size_t> counts(nops);
    for (int i = 0; i < nops; ++i) {
      peers[i] = ops[i].peer; sends[i] = ops[i].send; bufs[i] = ops[i].buf; counts[i] = ops[i].count;
    }
    return xch_(user_, nops, peers.data(), bufs.data(), counts.data(), sends.data(), s);
  }
};

// ------------------------------------------------------------------ peer flags
struct PeerHandle {  // what a rank publishes: kPeerHandleBytes
  hipIpcMemHandle_t ipc;
  int64_t pid;
  uint64_t ptr, nranks, cap;
};
static_assert(sizeof(PeerHandle) == kPeerHandleBytes, "PeerHandle layout");

}  // namespace

struct PeerMailbox {
  unsigned char* mem = nullptr;
  int nranks = 0;
  size_t cap = 0;
  bool taken = false;  // a mailbox serves ONE communicator: its counters count that communicator's collectives
  PeerHandle handle;
  ~PeerMailbox () { if (mem) (void) hipFree(mem); }
  // [head: 2 counters, error word][collectives: 2 x P slots][point to point: 2 x P flags, 2 x P acks, 2 x P slots]
  size_t p2p_off () const { return kMailboxHead + 2*static_cast<size_t>(nranks)*cap*sizeof(Real); }
  size_t bytes () const { return p2p_off() + 4*static_cast<size_t>(nranks)*8 + 2*static_cast<size_t>(nranks)*cap*sizeof(Real); }
};

namespace {

// Why two parities are enough. Collective number e (the same on every rank: collectives are issued in the same order)
// uses slots and counter e & 1 and waits for that counter to reach (e/2 + 1) P: P arrivals per collective of that
// parity. A rank can only issue its put of collective e + 1 after its own wait of e, i.e. after EVERY rank's put of e
// has landed in its mailbox; so when a put of e + 2 (the next user of parity e & 1) reaches my mailbox, its sender has
// passed its wait of e + 1, hence my put of e + 1 ran, and that is ordered on my side (stream order, or the
// cross-stream events) after my take of e: neither a slot of e nor the count of e can be disturbed by e + 2, and
// arrivals of e + 1 go to the other counter. The one-launch form counts P - 1 arrivals per collective (a rank sends
// nothing to itself) and puts at the START of its launch of e + 1, which its stream begins after its launch of e --
// wait and take included -- has ended: the same argument.
struct PeerFlagComm : public Comm {
  std::shared_ptr<PeerMailbox> mine_;
  int rank_, size_;
  PeerTable tab_;
  struct Mapped {  // other processes' mailboxes; unmapped also when the constructor throws half way
    std::vector<void*> p;
    ~Mapped () { for (void* q : p) (void) hipIpcCloseMemHandle(q); }
  } mapped_;
  uint64_t ncoll_ = 0;
  CrossStreamOrder ord_;
  // Put, wait and take as ONE launch that waits for the arrivals itself (peer_xchg_kernel; the default: 9 us per
  // collective back to back between two processes against 15 us, profiles/r05_peer_flags.txt), or
  // CEDR_HIP_PEER_WAIT=stream: two launches around a hipStreamWaitValue64. The same on every rank of a communicator: the
  // two forms count arrivals differently (without / with the rank's own block).
  bool in_kernel_ = true;
  unsigned long long max_polls_ = 1ull << 28;
  PeerFlagComm (const std::shared_ptr<PeerMailbox>& mine, const void* handles, int rank, int size)
    : mine_(mine), rank_(rank), size_(size) {
    cedr_throw_if( ! mine || mine->nranks != size || rank < 0 || rank >= size, "peer flags: the mailbox was made for "
                   << (mine ? mine->nranks : 0) << " ranks; asked for rank " << rank << " of " << size);
    cedr_throw_if(mine->taken, "peer flags: this mailbox already serves a communicator");
    std::memset(&tab_, 0, sizeof(tab_));
    const PeerHandle* const h = static_cast<const PeerHandle*>(handles);
    for (int r = 0; r < size; ++r) {
      PeerHandle hr;
      std::memcpy(&hr, h + r, sizeof(hr));
      cedr_throw_if(static_cast<int>(hr.nranks) != size || hr.cap != mine->cap,
                    "peer flags: rank " << r << "'s mailbox has another shape than this rank's");
      if (r == rank) {
        cedr_throw_if(hr.ptr != reinterpret_cast<uint64_t>(mine->mem), "peer flags: handle " << r << " is not this rank's");
        tab_.base[r] = mine->mem;
      } else if (hr.pid == static_cast<int64_t>(getpid())) {
        tab_.base[r] = reinterpret_cast<unsigned char*>(hr.ptr);  // ranks as threads of one process: the pointer itself
      } else {
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, hr.ipc, hipIpcMemLazyEnablePeerAccess);
        cedr_throw_if(e != hipSuccess, "peer flags: hipIpcOpenMemHandle of rank " << r << "'s mailbox: "
                      << hipGetErrorName(e));
        mapped_.p.push_back(p);
        tab_.base[r] = static_cast<unsigned char*>(p);
      }
    }
    mine->taken = true;
    if (const char* e = std::getenv("CEDR_HIP_PEER_WAIT")) in_kernel_ = std::strcmp(e, "stream") != 0;
    if (const char* e = std::getenv("CEDR_HIP_PEER_GIVE_UP_LOG2")) max_polls_ = 1ull << std::max(10, std::min(40, std::atoi(e)));
  }
  static bool overlap (const Real* a, const size_t na, const Real* b, const size_t nb) { return a < b + nb && b < a + na; }
  int rank () const override { return rank_; }
  int size () const override { return size_; }
  int round (const Real* send, Real* recv, const size_t count, const int sum, const size_t block, hipStream_t s) {
#ifdef __HIPCC__
    const uint64_t e = ncoll_++;
    const int parity = static_cast<int>(e & 1);
    const size_t cap = mine_->cap, P = static_cast<size_t>(size_);
    if (in_kernel_) {
      hipLaunchKernelGGL(peer_xchg_kernel, dim3(size_), dim3(256), 0, s, tab_, rank_, size_, send, recv, count, cap, parity,
                         static_cast<unsigned long long>((e/2 + 1)*(P - 1)), sum, block, max_polls_);
      return static_cast<int>(hipGetLastError());
    }
    hipLaunchKernelGGL(peer_put_kernel, dim3(size_), dim3(256), 0, s, tab_, send, count,
                       (parity*P + rank_)*cap, parity);
    hipError_t h = hipGetLastError();
    if (h != hipSuccess) return static_cast<int>(h);
    h = hipStreamWaitValue64(s, reinterpret_cast<uint64_t*>(mine_->mem) + parity, (e/2 + 1)*P, hipStreamWaitValueGte,
                             ~0ull);
    if (h != hipSuccess) return static_cast<int>(h);
    hipLaunchKernelGGL(peer_take_kernel, dim3(static_cast<unsigned>((count + 255)/256)), dim3(256), 0, s,
                       mine_->mem, recv, count, parity*P*cap, cap, size_, sum, block);
    return static_cast<int>(hipGetLastError());
#else
    (void) send; (void) recv; (void) count; (void) sum; (void) block; (void) s;
    return -3;  // a host-only build of this file has no kernels
#endif
  }
  int collective (const Real* send, Real* recv, const size_t count, const int sum, hipStream_t s) {
    if (count == 0) return 0;
    // (the one-launch form reads `send` while other workgroups may already write `recv`)
    if (in_kernel_ && overlap(send, count, recv, sum ? count : count*static_cast<size_t>(size_))) return -5;
    if (const int eo = ord_.before(s)) return eo;
    int e = 0;
    const size_t cap = mine_->cap;
    // (longer than a slot: in pieces, each a round of its own; an all-gather's pieces land at their place in every block)
    for (size_t o = 0; o < count && ! e; o += cap) e = round(send + o, recv + o, std::min(cap, count - o), sum, count, s);
    const int ea = ord_.after(s);
    return e ? e : ea;
  }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return collective(send, recv, count, 0, s);
  }
  // messages sent to / received from every peer so far (pieces of a long message count one each)
  std::vector<unsigned long long> nsent_, nrecvd_;
  bool has_exchange () const override {
#ifdef __HIPCC__
    return size_ <= kPeerMaxOps/2 + 1;
#else
    return false;
#endif
  }
  int exchange (const P2P* ops, int nops, hipStream_t s) override {
    if (nops == 0) return 0;
#ifdef __HIPCC__
    if ( ! has_exchange() || nops > kPeerMaxOps) return -2;
    if (nsent_.empty()) { nsent_.assign(size_, 0); nrecvd_.assign(size_, 0); }
    PeerOps po;
    std::memset(&po, 0, sizeof(po));
    const size_t cap = mine_->cap;
    int n = 0;
    for (int i = 0; i < nops; ++i) {
      if (ops[i].peer < 0 || ops[i].peer >= size_ || ops[i].peer == rank_) return -2;
      if (ops[i].count == 0) continue;  // (both sides know the count: nothing travels)
      unsigned long long& seq = ops[i].send ? nsent_[ops[i].peer] : nrecvd_[ops[i].peer];
      po.op[n++] = PeerOp{ops[i].buf, ops[i].count, seq, ops[i].peer, ops[i].send ? 1 : 0};
      seq += (ops[i].count + cap - 1)/cap;
    }
    if (n == 0) return 0;
    if (const int eo = ord_.before(s)) return eo;
    hipLaunchKernelGGL(peer_p2p_kernel, dim3(n), dim3(256), 0, s, tab_, po, rank_, size_, cap, mine_->p2p_off(), max_polls_);
    const int e = static_cast<int>(hipGetLastError());
    const int ea = ord_.after(s);
    return e ? e : ea;
#else
    (void) ops; (void) s;
    return -3;
#endif
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return collective(send, recv, count, 1, s);
  }
};

// ------------------------------------------------------------------- rehearsal
// Rank 0 of 'size' pretended ranks over a communicator of ONE real rank: what a single GPU can run of an N-GPU
// job. A collective goes through the inner communicator for real (RcclComm of one rank: librccl's own enqueue
// path and the cross-stream ordering events of this file) and the other ranks' blocks of an all-gather are copies
// of this rank's. Measurement and test tool (bench.py --pretend-ranks, tools/): the RESULTS are those of a job in
// which every rank holds this rank's data, not of any real decomposition.
struct RehearsalComm : public Comm {
  std::shared_ptr<Comm> inner_;
  int size_;
  RehearsalComm (const std::shared_ptr<Comm>& inner, int size) : inner_(inner), size_(size) {}
  int rank () const override { return 0; }
  int size () const override { return size_; }
  int allgather (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    const int e = inner_->allgather(send, recv, count, s);  // block 0, through the real communicator
    if (e || size_ == 1 || count == 0) return e;
#ifdef __HIPCC__
    hipLaunchKernelGGL(rehearsal_replicate_kernel, dim3(static_cast<unsigned>((count + 255)/256)), dim3(256), 0, s,
                       recv, count, size_);
    return static_cast<int>(hipGetLastError());
#else
    for (int r = 1; r < size_; ++r) {
      const hipError_t h = hipMemcpyAsync(recv + static_cast<size_t>(r)*count, recv, count*sizeof(Real),
                                          hipMemcpyDeviceToDevice, s);
      if (h != hipSuccess) return static_cast<int>(h);
    }
    return 0;
#endif
  }
  int allreduce_sum (const Real* send, Real* recv, size_t count, hipStream_t s) override {
    return inner_->allreduce_sum(send, recv, count, s);
  }
};

}  // namespace

Parallel::Ptr make_parallel_rehearsal (const Parallel::Ptr& inner, int pretended_size) {
  cedr_throw_if( ! inner || inner->size() != 1, "a rehearsal communicator wraps a communicator of one rank");
  cedr_throw_if(pretended_size < 1, "pretended_size must be at least 1");
  return std::make_shared<Parallel>(std::make_shared<RehearsalComm>(inner->comm().ptr(), pretended_size));
}

std::shared_ptr<PeerMailbox> peer_mailbox_create (int size, size_t slot_doubles) {
  cedr_throw_if(size < 1 || size > kPeerFlagsMaxRanks, "peer flags: 1 to " << kPeerFlagsMaxRanks << " ranks (one node)");
  cedr_throw_if(slot_doubles < 1, "peer flags: empty slots");
  int can = 0, dev = 0;
  (void) hipGetDevice(&dev);
  (void) hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev);
  cedr_throw_if( ! can, "peer flags: this device has no hipStreamWaitValue64");
  auto mb = std::make_shared<PeerMailbox>();
  mb->nranks = size;
  mb->cap = slot_doubles;
  void* p = nullptr;
  // fine-grained: stores from another GPU (and from another process on this one) are seen without a cache flush
  hipError_t e = hipExtMallocWithFlags(&p, mb->bytes(), hipDeviceMallocFinegrained);
  cedr_throw_if(e != hipSuccess, "peer flags: fine-grained allocation of " << mb->bytes() << " bytes: " << hipGetErrorName(e));
  mb->mem = static_cast<unsigned char*>(p);
  e = hipMemset(p, 0, mb->bytes());
  if (e == hipSuccess) e = hipDeviceSynchronize();
  cedr_throw_if(e != hipSuccess, "peer flags: clearing the mailbox: " << hipGetErrorName(e));
  std::memset(&mb->handle, 0, sizeof(mb->handle));
  e = hipIpcGetMemHandle(&mb->handle.ipc, p);
  cedr_throw_if(e != hipSuccess, "peer flags: hipIpcGetMemHandle: " << hipGetErrorName(e));
  mb->handle.pid = static_cast<int64_t>(getpid());
  mb->handle.ptr = reinterpret_cast<uint64_t>(p);
  mb->handle.nranks = static_cast<uint64_t>(size);
  mb->handle.cap = slot_doubles;
  return mb;
}

void peer_mailbox_handle (const PeerMailbox& mb, void* handle_96B) {
  std::memcpy(handle_96B, &mb.handle, sizeof(mb.handle));
}

Parallel::Ptr make_parallel_peer_flags (const std::shared_ptr<PeerMailbox>& mine, const void* handles, int rank, int size) {
  return std::make_shared<Parallel>(std::make_shared<PeerFlagComm>(mine, handles, rank, size));
}

Parallel::Ptr make_parallel () {
  return std::make_shared<Parallel>(std::make_shared<SelfComm>());
}

Parallel::Ptr make_parallel (const std::shared_ptr<Comm>& comm) {
  return std::make_shared<Parallel>(comm);
}

int rccl_get_unique_id (void* unique_id_128B) {
  const RcclApi& a = rccl();
  if ( ! a.ok) return -1;
  ncclUniqueIdBlob blob;
  const int e = a.GetUniqueId(&blob);
  if (e != kNcclSuccess) return e;
  std::memcpy(unique_id_128B, &blob, sizeof(blob));
  return 0;
}

Parallel::Ptr make_parallel_rccl (const void* unique_id_128B, int rank, int size) {
  return std::make_shared<Parallel>(std::make_shared<RcclComm>(unique_id_128B, rank, size));
}

Parallel::Ptr make_parallel_callbacks (int rank, int size, AllgatherFn ag, AllreduceSumFn ar,
                                       void* user, ExchangeFn xch) {
  cedr_throw_if( ! ag || ! ar, "null collective callback");
  return std::make_shared<Parallel>(std::make_shared<CallbackComm>(rank, size, ag, ar, user, xch));
}

}  // namespace mpi
}  // namespace cedr
