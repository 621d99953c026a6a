// Microbenchmark (diagnostic, not product): HBM read bandwidth of the tile sweeps' access patterns,
// without any of their LDS / barrier / arithmetic structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) Pair { double x, y; };

// (a) contiguous stream, 16 B per lane, grid-stride
__global__ __launch_bounds__(256) void k_stream (const d2* __restrict__ p, size_t n2, double* out, int nt) {
  double s = 0;
  for (size_t i = blockIdx.x*256ull + threadIdx.x; i < n2; i += gridDim.x*256ull) {
    const d2 v = nt ? __builtin_nontemporal_load(p + i) : p[i];
    s += v.x + v.y;
  }
  if (s == 1.2345e-300) out[0] = s;
}

// (b) tile pattern: workgroup = tile of ncell cells; 'nfields' rows 'stride' doubles apart; a wave
// takes rows wave, wave+4, ...; lane i reads cells 3i, 3i+1 (16 B) and 3i+2 (8 B) of its row.
// mode 1: the row segment as contiguous 16-byte pieces instead (lane i reads bytes 16i..16i+15).
template <int MODE>
__global__ __launch_bounds__(256) void k_tile (const double* __restrict__ p, int64_t stride, int nfields, int ncell,
                                               int ntiles, double* out) {
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* base = p + static_cast<size_t>(t)*ncell;
  double s = 0;
  for (int f = wave; f < nfields; f += 4) {
    const double* row = base + static_cast<size_t>(f)*stride;
    if (MODE == 0) {
      if (3*lane + 2 < ncell + 3) {
        const Pair ab = *reinterpret_cast<const Pair*>(row + 3*lane);
        s += ab.x + ab.y + row[3*lane + 2];
      }
    } else {
      for (int i = lane; 2*i < ncell; i += 64) {
        const Pair ab = *reinterpret_cast<const Pair*>(row + 2*i);
        s += ab.x + ab.y;
      }
    }
  }
  if (s == 1.2345e-300) out[0] = s;
}

// (c) the up sweep's issue pattern: per chunk of 16 rows a wave requests its 4 rows, the workgroup
// meets at a barrier, and only then is the next chunk requested (DEPTH 0) or the next chunk is
// requested before this one is consumed (DEPTH 1). Dynamic LDS bytes set the occupancy.
template <int DEPTH>
__global__ __launch_bounds__(256) void k_chunks (const double* __restrict__ p, int64_t stride, int nfields, int ncell,
                                                 int ntiles, double* out) {
  extern __shared__ double lds[];
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double* base = p + static_cast<size_t>(t)*ncell + 3*lane;
  const int nchunks = (nfields + 15)/16;
  double s = 0;
  Pair ab[2][4]; double c[2][4];
  auto load = [&] (int ch, int b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int f = ch*16 + wave + 4*r; if (f > nfields - 1) f = nfields - 1;
      const double* row = base + static_cast<size_t>(f)*stride;
      ab[b][r] = *reinterpret_cast<const Pair*>(row); c[b][r] = row[2];
    }
  };
  if (DEPTH == 1) load(0, 0);
  for (int ch = 0; ch < nchunks; ++ch) {
    const int b = DEPTH == 1 ? (ch & 1) : 0;
    if (DEPTH == 1) { if (ch + 1 < nchunks) load(ch + 1, b ^ 1); } else load(ch, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) s += ab[b][r].x + ab[b][r].y + c[b][r];
    lds[threadIdx.x] = s;
    __syncthreads();
    s += lds[(threadIdx.x + 64) & 255];
    __syncthreads();
  }
  if (s == 1.2345e-300) out[0] = s;
}

int main () {
  const int ncells = 1500000, nfields = 161, ncell = 183, ntiles = 8192;
  const int64_t stride = (ncells + 65536 + 15)/16*16;  // field-major rows, as the QLT buffer
  const size_t n = static_cast<size_t>(stride)*nfields;
  double *d, *out;
  CK(hipMalloc(&d, n*8)); CK(hipMalloc(&out, 64)); CK(hipMemset(d, 0, n*8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&] (const char* name, double bytes, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    printf("%-64s %.4f ms  %.0f GB/s\n", name, ms, bytes/ms/1e6);
    return 0;
  };
  const double bytes = static_cast<double>(ncells)*nfields*8;
  const size_t n2 = static_cast<size_t>(ncells)*nfields/2;
  timeit("contiguous 16 B/lane, 2048 WGs", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const d2*>(d), n2, out, 0); });
  timeit("contiguous 16 B/lane, 2048 WGs, nontemporal", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const d2*>(d), n2, out, 1); });
  timeit("contiguous 16 B/lane, 8192 WGs, nontemporal", bytes, [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const d2*>(d), n2, out, 1); });
  const double tb = static_cast<double>(ntiles)*ncell*nfields*8;
  timeit("tile pattern: lane = 3 cells (16 B + 8 B), 161 rows", tb, [&] { hipLaunchKernelGGL(k_tile<0>, dim3(8192), dim3(256), 0, 0, d, stride, nfields, ncell, ntiles, out); });
  timeit("tile pattern: row segment as contiguous 16-B pieces", tb, [&] { hipLaunchKernelGGL(k_tile<1>, dim3(8192), dim3(256), 0, 0, d, stride, nfields, ncell, ntiles, out); });
  timeit("tile pattern 3-cell lanes, 366-cell tiles (4096 WGs)", tb, [&] { hipLaunchKernelGGL(k_tile<1>, dim3(4096), dim3(256), 0, 0, d, stride, nfields, 2*ncell, ntiles/2, out); });
  for (int ldskb : {8, 20, 26, 40}) {
    char nm[128];
    snprintf(nm, sizeof nm, "chunked + barriers, no prefetch, %d KB LDS/WG", ldskb);
    timeit(nm, tb, [&] { hipLaunchKernelGGL(k_chunks<0>, dim3(8192), dim3(256), ldskb*1024, 0, d, stride, nfields, ncell, ntiles, out); });
    snprintf(nm, sizeof nm, "chunked + barriers, next chunk in flight, %d KB LDS/WG", ldskb);
    timeit(nm, tb, [&] { hipLaunchKernelGGL(k_chunks<1>, dim3(8192), dim3(256), ldskb*1024, 0, d, stride, nfields, ncell, ntiles, out); });
  }
  return 0;
}
