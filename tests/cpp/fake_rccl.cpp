// Test double of the nine RCCL entry points cedr::mpi's RcclComm binds (compose_amd/csrc/cedr_mpi.cpp),
// for ranks that are THREADS of one process on ONE device -- the only way more than one rank of that
// class can run on a one-GPU box (RCCL itself refuses two ranks on a device). Test infrastructure: the
// product loads it only when CEDR_HIP_RCCL_LIB names it (tests/test_gpu_rccl_double.py).
//
// What it checks that the other multi-rank harnesses (callbacks, gloo, MPI) cannot: the calls RcclComm
// itself makes -- datatype and reduction codes, counts, peers, the pairing of sends and receives of a
// group, that a rank never posts two messages to one peer in a group -- and that the engine's buffers are
// laid out as ncclAllGather lays them out (rank r's piece at r*count). What it cannot: RCCL's asynchronous
// execution. Every call here completes before it returns (stream synchronised, host rendezvous).
#include <hip/hip_runtime.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

enum { kSuccess = 0, kUnhandled = 1, kSystem = 2, kInternal = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
const int kFloat64 = 8, kSum = 0;
const std::chrono::seconds kPatience(60);

struct World {
  int size = 0, joined = 0;
  std::mutex m;
  std::condition_variable cv;
  // barrier
  int waiting = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<std::vector<double> > slot;
  std::map<std::pair<int, int>, std::deque<std::vector<double> > > mail;  // (source, destination)

  bool barrier () {
    std::unique_lock<std::mutex> lk(m);
    if (broken) return false;
    const uint64_t g = generation;
    if (++waiting == size) {
      waiting = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if ( ! cv.wait_for(lk, kPatience, [&] { return generation != g || broken; })) broken = true;
    if (broken) cv.notify_all();
    return ! broken;
  }
};

struct Comm {
  std::shared_ptr<World> w;
  int rank;
};

std::mutex g_m;
std::map<uint64_t, std::shared_ptr<World> > g_worlds;
uint64_t g_next_id = 1;
struct Stats { long allgather, allreduce, groups, sends, recvs, errors; } g_stats = {0, 0, 0, 0, 0, 0};

struct Op { bool send; void* buf; size_t count; int peer; Comm* c; hipStream_t s; };
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

int fail (const int code, const char* what) {
  std::fprintf(stderr, "fake_rccl: %s\n", what);
  std::lock_guard<std::mutex> lk(g_m);
  ++g_stats.errors;
  return code;
}

bool d2h (std::vector<double>& h, const void* d, const size_t count, hipStream_t s) {
  h.resize(count);
  if (count == 0) return hipStreamSynchronize(s) == hipSuccess;
  return hipMemcpyAsync(h.data(), d, count*sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess &&
    hipStreamSynchronize(s) == hipSuccess;
}
bool h2d (void* d, const double* h, const size_t count, hipStream_t s) {
  if (count == 0) return true;
  return hipMemcpyAsync(d, h, count*sizeof(double), hipMemcpyHostToDevice, s) == hipSuccess &&
    hipStreamSynchronize(s) == hipSuccess;
}

int run_ops (std::vector<Op>& ops) {
  // one message per (peer, direction) and group, as cedr::mpi::Comm::exchange promises
  for (size_t i = 0; i < ops.size(); ++i) {
    const Op& a = ops[i];
    if (a.peer < 0 || a.peer >= a.c->w->size) return fail(kInvalidArgument, "peer out of range");
    if (a.peer == a.c->rank && a.c->w->size > 1) return fail(kInvalidUsage, "message to self in a multi-rank group");
    for (size_t j = i + 1; j < ops.size(); ++j)
      if (ops[j].send == a.send && ops[j].peer == a.peer && ops[j].c == a.c)
        return fail(kInvalidUsage, "two messages in one direction to one peer in a group");
  }
  for (const Op& a : ops) {
    if ( ! a.send) continue;
    std::vector<double> h;
    if ( ! d2h(h, a.buf, a.count, a.s)) return fail(kUnhandled, "copy from device failed");
    World& w = *a.c->w;
    std::lock_guard<std::mutex> lk(w.m);
    w.mail[std::make_pair(a.c->rank, a.peer)].push_back(std::move(h));
    w.cv.notify_all();
  }
  for (const Op& a : ops) {
    if (a.send) continue;
    World& w = *a.c->w;
    std::vector<double> h;
    {
      std::unique_lock<std::mutex> lk(w.m);
      std::deque<std::vector<double> >& q = w.mail[std::make_pair(a.peer, a.c->rank)];
      if ( ! w.cv.wait_for(lk, kPatience, [&] { return ! q.empty() || w.broken; }) || q.empty()) {
        w.broken = true;
        w.cv.notify_all();
        lk.unlock();
        return fail(kSystem, "a receive was never matched by a send");
      }
      h.swap(q.front());
      q.pop_front();
    }
    if (h.size() != a.count) return fail(kInternal, "send and receive counts differ");
    if ( ! h2d(a.buf, h.data(), a.count, a.s)) return fail(kUnhandled, "copy to device failed");
  }
  return kSuccess;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };
typedef Comm* ncclComm_t;

int ncclGetUniqueId (ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  std::lock_guard<std::mutex> lk(g_m);
  const uint64_t v = g_next_id++;
  std::memcpy(id->internal, &v, sizeof(v));
  return kSuccess;
}

int ncclCommInitRank (ncclComm_t* comm, int size, ncclUniqueId id, int rank) {
  if (size < 1 || rank < 0 || rank >= size) return fail(kInvalidArgument, "rank / size");
  uint64_t v;
  std::memcpy(&v, id.internal, sizeof(v));
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_m);
    std::shared_ptr<World>& e = g_worlds[v];
    if ( ! e) {
      e = std::make_shared<World>();
      e->size = size;
      e->slot.resize(size);
    }
    w = e;
  }
  if (w->size != size) return fail(kInvalidArgument, "ranks of one id disagree on the size");
  {
    std::unique_lock<std::mutex> lk(w->m);
    ++w->joined;
    w->cv.notify_all();
    if ( ! w->cv.wait_for(lk, kPatience, [&] { return w->joined >= size; }))
      return fail(kSystem, "not every rank joined the communicator");
  }
  *comm = new Comm{w, rank};
  return kSuccess;
}

int ncclCommDestroy (ncclComm_t comm) {
  delete comm;
  return kSuccess;
}

int ncclAllGather (const void* send, void* recv, size_t count, int dtype, ncclComm_t c, hipStream_t s) {
  if (dtype != kFloat64) return fail(kInvalidArgument, "all-gather: datatype is not float64");
  if (t_depth) return fail(kInvalidUsage, "collective inside a group");
  World& w = *c->w;
  if ( ! d2h(w.slot[c->rank], send, count, s)) return fail(kUnhandled, "copy from device failed");
  if ( ! w.barrier()) return fail(kSystem, "all-gather: a rank is missing");
  std::vector<double> all;
  for (int r = 0; r < w.size; ++r) {
    if (w.slot[r].size() != count) return fail(kInternal, "all-gather: ranks disagree on the count");
    all.insert(all.end(), w.slot[r].begin(), w.slot[r].end());
  }
  if ( ! w.barrier()) return fail(kSystem, "all-gather: a rank is missing");
  if ( ! h2d(recv, all.data(), all.size(), s)) return fail(kUnhandled, "copy to device failed");
  { std::lock_guard<std::mutex> lk(g_m); ++g_stats.allgather; }
  return kSuccess;
}

int ncclAllReduce (const void* send, void* recv, size_t count, int dtype, int op, ncclComm_t c, hipStream_t s) {
  if (dtype != kFloat64 || op != kSum) return fail(kInvalidArgument, "all-reduce: not a float64 sum");
  if (t_depth) return fail(kInvalidUsage, "collective inside a group");
  World& w = *c->w;
  if ( ! d2h(w.slot[c->rank], send, count, s)) return fail(kUnhandled, "copy from device failed");
  if ( ! w.barrier()) return fail(kSystem, "all-reduce: a rank is missing");
  std::vector<double> sum(count, 0.0);
  for (int r = 0; r < w.size; ++r) {
    if (w.slot[r].size() != count) return fail(kInternal, "all-reduce: ranks disagree on the count");
    for (size_t i = 0; i < count; ++i) sum[i] = r == 0 ? w.slot[0][i] : sum[i] + w.slot[r][i];
  }
  if ( ! w.barrier()) return fail(kSystem, "all-reduce: a rank is missing");
  if ( ! h2d(recv, sum.data(), count, s)) return fail(kUnhandled, "copy to device failed");
  { std::lock_guard<std::mutex> lk(g_m); ++g_stats.allreduce; }
  return kSuccess;
}

int ncclGroupStart () {
  ++t_depth;
  return kSuccess;
}

int ncclGroupEnd () {
  if (t_depth <= 0) return fail(kInvalidUsage, "group end without a start");
  if (--t_depth) return kSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  { std::lock_guard<std::mutex> lk(g_m); ++g_stats.groups; }
  return run_ops(ops);
}

static int post (const bool send, void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  if (dtype != kFloat64) return fail(kInvalidArgument, "message: datatype is not float64");
  {
    std::lock_guard<std::mutex> lk(g_m);
    ++(send ? g_stats.sends : g_stats.recvs);
  }
  t_ops.push_back(Op{send, buf, count, peer, c, s});
  if (t_depth) return kSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_ops(ops);
}

int ncclSend (const void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, c, s);
}

int ncclRecv (void* buf, size_t count, int dtype, int peer, ncclComm_t c, hipStream_t s) {
  return post(false, buf, count, dtype, peer, c, s);
}

// For the test: how often each entry point ran (so that a run that silently used another transport fails).
void fake_rccl_stats (long* out6) {
  std::lock_guard<std::mutex> lk(g_m);
  out6[0] = g_stats.allgather; out6[1] = g_stats.allreduce; out6[2] = g_stats.groups;
  out6[3] = g_stats.sends; out6[4] = g_stats.recvs; out6[5] = g_stats.errors;
}

}  // extern "C"
