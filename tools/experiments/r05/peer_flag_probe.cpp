// Round 5 feasibility probe for a stream-ordered peer-flag exchange: which kinds of device memory can
// carry a flag that (a) hipStreamWaitValue64 accepts and (b) another PROCESS can map (hipIpcGetMemHandle /
// hipIpcOpenMemHandle) and write with hipStreamWriteValue64. No kernels are launched; every wait that is enqueued has a
// host deadline after which this process releases it itself, so a stream never stays blocked.
//
//   peer_flag_probe owner DIR     allocates, exports, waits for the writer
//   peer_flag_probe writer DIR    opens the owner's handles, writes a payload and the flag
//
// build: hipcc -O2 -std=c++17 --offload-arch=gfx950 -o peer_flag_probe peer_flag_probe.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

namespace {
using clk = std::chrono::steady_clock;
double since (const clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }
const char* es (const hipError_t e) { return hipGetErrorName(e); }

// Poll a stream until it is idle or 'ms' milliseconds have passed. Returns the time it took, or -1 at the deadline.
double drain (hipStream_t s, const double ms) {
  const auto t0 = clk::now();
  while (since(t0) < ms) {
    if (hipStreamQuery(s) == hipSuccess) return since(t0);
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return -1;
}

// Enqueue wait(flag >= v) + a 4-byte marker copy on sw, release it from sr with hipStreamWriteValue64 after 'delay' ms.
void local_wait_write (const char* kind, void* flag, hipStream_t sw, hipStream_t sr, uint32_t* marker_d, const uint64_t v) {
  uint32_t one = 1;
  hipError_t e = hipStreamWaitValue64(sw, flag, v, hipStreamWaitValueGte, ~0ull);
  std::printf("  [%s] hipStreamWaitValue64 -> %s\n", kind, es(e));
  if (e != hipSuccess) { (void) hipGetLastError(); return; }
  (void) hipMemcpyAsync(marker_d, &one, 4, hipMemcpyHostToDevice, sw);
  std::this_thread::sleep_for(std::chrono::milliseconds(20));
  std::printf("  [%s] stream idle while the flag is unset: %s\n", kind, hipStreamQuery(sw) == hipSuccess ? "YES (wait did not hold!)" : "no (held)");
  (void) hipGetLastError();
  const auto t0 = clk::now();
  e = hipStreamWriteValue64(sr, flag, v, 0);
  std::printf("  [%s] hipStreamWriteValue64 from a second stream -> %s\n", kind, es(e));
  double t = drain(sw, 3000);
  if (t < 0) {  // still held: last resort, a host-side copy into the flag
    const uint64_t vv = v;
    std::printf("  [%s] NOT released by the write within 3 s; releasing by hipMemcpy -> %s\n", kind,
                es(hipMemcpy(flag, &vv, 8, hipMemcpyHostToDevice)));
    t = drain(sw, 3000);
    std::printf("  [%s] after the copy: %s\n", kind, t < 0 ? "STILL HELD" : "released");
  } else {
    std::printf("  [%s] released %.3f ms after the write was enqueued\n", kind, t);
  }
}
}  // namespace

int main (int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s owner|writer DIR\n", argv[0]); return 2; }
  const std::string role = argv[1], dir = argv[2];
  const std::string hfile = dir + "/handles.bin", dfile = dir + "/writer_done";
  if (hipSetDevice(0) != hipSuccess) { std::printf("no device\n"); return 1; }
  struct Exported { hipIpcMemHandle_t h[4]; int ok[4]; } ex;
  std::memset(&ex, 0, sizeof(ex));
  const char* kinds[4] = {"payload: hipMalloc", "flag: hipMallocSignalMemory", "flag: hipMalloc", "flag: hipDeviceMallocFinegrained"};
  hipStream_t s1, s2;
  (void) hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  (void) hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);

  if (role == "owner") {
    int can = -1;
    hipError_t e = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    std::printf("hipDeviceAttributeCanUseStreamWaitValue -> %s, %d\n", es(e), can);
    void* p[4] = {nullptr, nullptr, nullptr, nullptr};
    std::printf("alloc payload           -> %s\n", es(hipMalloc(&p[0], 4096)));
    std::printf("alloc signal memory     -> %s\n", es(hipExtMallocWithFlags(&p[1], 8, hipMallocSignalMemory)));
    std::printf("alloc plain flag        -> %s\n", es(hipMalloc(&p[2], 8)));
    std::printf("alloc fine-grained flag -> %s\n", es(hipExtMallocWithFlags(&p[3], 8, hipDeviceMallocFinegrained)));
    (void) hipGetLastError();
    for (int i = 0; i < 4; ++i) if (p[i] && i != 1) (void) hipMemset(p[i], 0, i == 0 ? 4096 : 8);
    (void) hipDeviceSynchronize();
    uint32_t* marker = nullptr;
    (void) hipMalloc(&marker, 4);
    std::printf("one process, two streams:\n");
    for (int i = 1; i < 4; ++i) if (p[i]) local_wait_write(kinds[i], p[i], s1, s2, marker, 1);
    std::printf("export:\n");
    for (int i = 0; i < 4; ++i) {
      if ( ! p[i]) continue;
      e = hipIpcGetMemHandle(&ex.h[i], p[i]);
      ex.ok[i] = e == hipSuccess;
      (void) hipGetLastError();
      std::printf("  hipIpcGetMemHandle(%s) -> %s\n", kinds[i], es(e));
    }
    if (FILE* f = std::fopen(hfile.c_str(), "wb")) { std::fwrite(&ex, sizeof(ex), 1, f); std::fclose(f); }
    // Two processes: wait on every exportable flag the wait call accepts, for the writer's value 7.
    std::printf("two processes (the writer maps the handles):\n");
    bool waiting[4] = {false, false, false, false};
    hipStream_t sw[4];
    for (int i = 1; i < 4; ++i) {
      (void) hipStreamCreateWithFlags(&sw[i], hipStreamNonBlocking);
      if ( ! p[i] || ! ex.ok[i]) continue;
      e = hipStreamWaitValue64(sw[i], p[i], 7, hipStreamWaitValueGte, ~0ull);
      (void) hipGetLastError();
      std::printf("  [%s] wait for 7 enqueued -> %s\n", kinds[i], es(e));
      waiting[i] = e == hipSuccess;
    }
    const auto t0 = clk::now();
    bool done = false;
    while (since(t0) < 20000 && ! done) {  // the writer says when it is through
      done = access(dfile.c_str(), F_OK) == 0;
      std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    std::printf("  writer %s after %.0f ms\n", done ? "finished" : "DID NOT FINISH", since(t0));
    for (int i = 1; i < 4; ++i) {
      if ( ! waiting[i]) continue;
      double t = drain(sw[i], 2000);
      std::printf("  [%s] released by the other process: %s\n", kinds[i], t < 0 ? "NO" : "yes");
      if (t < 0) {  // let go of the stream ourselves
        (void) hipStreamWriteValue64(s2, p[i], 7, 0);
        t = drain(sw[i], 3000);
        if (t < 0) { const uint64_t v = 7; (void) hipMemcpy(p[i], &v, 8, hipMemcpyHostToDevice); t = drain(sw[i], 3000); }
        std::printf("  [%s] released by this process instead: %s\n", kinds[i], t < 0 ? "NO" : "yes");
      }
    }
    uint64_t got[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) if (p[i] && i != 1) (void) hipMemcpy(&got[i], p[i], 8, hipMemcpyDeviceToHost);
    std::printf("  payload word %llx (writer puts c0ffee), plain flag %llu, fine-grained flag %llu\n",
                (unsigned long long) got[0], (unsigned long long) got[2], (unsigned long long) got[3]);
    (void) hipDeviceSynchronize();
    std::printf("owner done\n");
    return 0;
  }

  // writer
  const auto t0 = clk::now();
  FILE* f = nullptr;
  while (since(t0) < 20000 && ! (f = std::fopen(hfile.c_str(), "rb"))) std::this_thread::sleep_for(std::chrono::milliseconds(20));
  if ( ! f) { std::printf("writer: no handles\n"); return 1; }
  std::this_thread::sleep_for(std::chrono::milliseconds(200));  // let the owner finish the file and enqueue its waits
  const size_t nr = std::fread(&ex, sizeof(ex), 1, f);
  std::fclose(f);
  if (nr != 1) { std::printf("writer: short handle file\n"); return 1; }
  void* q[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < 4; ++i) {
    if ( ! ex.ok[i]) continue;
    const hipError_t e = hipIpcOpenMemHandle(&q[i], ex.h[i], hipIpcMemLazyEnablePeerAccess);
    (void) hipGetLastError();
    std::printf("writer: hipIpcOpenMemHandle(%s) -> %s\n", kinds[i], es(e));
    if (e != hipSuccess) q[i] = nullptr;
  }
  const uint64_t word = 0xc0ffee;
  if (q[0]) std::printf("writer: payload copy -> %s\n", es(hipMemcpyAsync(q[0], &word, 8, hipMemcpyHostToDevice, s1)));
  for (int i = 1; i < 4; ++i) {
    if ( ! q[i]) continue;
    const hipError_t e = hipStreamWriteValue64(s1, q[i], 7, 0);
    (void) hipGetLastError();
    std::printf("writer: hipStreamWriteValue64(%s) -> %s\n", kinds[i], es(e));
  }
  std::printf("writer: stream drained: %s\n", drain(s1, 5000) < 0 ? "NO" : "yes");
  if (FILE* g = std::fopen(dfile.c_str(), "w")) std::fclose(g);
  std::this_thread::sleep_for(std::chrono::milliseconds(500));  // keep the mappings until the owner has looked
  for (int i = 0; i < 4; ++i) if (q[i]) (void) hipIpcCloseMemHandle(q[i]);
  std::printf("writer done\n");
  return 0;
}
