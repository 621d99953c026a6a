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

// (d) the root->leaves sweep's traffic without its arithmetic: per chunk of 4 tracers wave w reads its tracer's rows lo, orig, hi
// (lane = 3 cells: 16 B + 8 B per row) and 16 output masses from the result buffer, the workgroup meets at three barriers, and
// wave w writes its tracer's result row (3 cells per lane). 28 KB of dynamic LDS: five workgroups per CU, as the kernel.
// WRITE 0: no result stores.
template <int WRITE>
__global__ __launch_bounds__(256, 5) void k_down (const double* __restrict__ l2r, double* __restrict__ r2l, int64_t stride,
                                                   int nt, int ncell, int ntiles, int64_t out0, double* out, int pitch = 0) {
  extern __shared__ double lds[];
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (pitch == 0) pitch = ncell;
  const size_t cell0 = static_cast<size_t>(t)*pitch + 3*lane;
  double s = 0;
  for (int ch = 0; ch < (nt + 3)/4; ++ch) {
    int tr = 4*ch + wave; if (tr > nt - 1) tr = nt - 1;
    double v[3][3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
      const double* row = l2r + static_cast<size_t>(1 + 4*tr + f)*stride + cell0;
      const Pair ab = *reinterpret_cast<const Pair*>(row);
      v[f][0] = ab.x; v[f][1] = ab.y; v[f][2] = row[2];
    }
    const double x = r2l[static_cast<size_t>(tr)*stride + out0 + 16*t + (lane & 15)];
    double r0 = x, r1 = x, r2 = x;
#pragma unroll
    for (int f = 0; f < 3; ++f) { r0 += v[f][0]; r1 += v[f][1]; r2 += v[f][2]; }
    lds[threadIdx.x] = r0;
    __syncthreads();
    r1 += lds[(threadIdx.x + 64) & 255];
    __syncthreads();
    lds[threadIdx.x] = r1;
    __syncthreads();
    r2 += lds[(threadIdx.x + 128) & 255];
    if (WRITE == 1 && 3*lane < ncell) {
      double* row = r2l + static_cast<size_t>(tr)*stride + cell0;
      Pair ab; ab.x = r0; ab.y = r1;
      *reinterpret_cast<Pair*>(row) = ab;
      row[2] = r2;
    }
    if ((WRITE == 3 || WRITE == 4) && 3*lane < ncell) {
      // 3: IN PLACE over the 'orig' row this wave has just read; 4: into the tracer's fourth row of the same buffer
      double* row = const_cast<double*>(l2r) + static_cast<size_t>(1 + 4*tr + (WRITE == 3 ? 1 : 3))*stride + cell0;
      Pair ab; ab.x = r0; ab.y = r1;
      *reinterpret_cast<Pair*>(row) = ab;
      row[2] = r2;
    }
    if (WRITE == 2) {
      // the same row segment written as contiguous 16-byte pieces (lane i: cells 2i, 2i + 1), through LDS: interior lines are
      // written whole by one store instruction
      __syncthreads();
      double* stg = lds + 256 + wave*200;
      stg[3*lane] = r0; stg[3*lane + 1] = r1; stg[3*lane + 2] = r2;
      __builtin_amdgcn_wave_barrier();
      double* row = r2l + static_cast<size_t>(tr)*stride + static_cast<size_t>(t)*pitch;
      for (int i = lane; 2*i < ncell; i += 64) {
        Pair ab; ab.x = stg[2*i]; ab.y = stg[2*i + 1];
        if (2*i + 1 < ncell) *reinterpret_cast<Pair*>(row + 2*i) = ab;
        else row[2*i] = ab.x;
      }
    }
    s += r2;
  }
  if (s == 1.2345e-300) out[0] = s;
}

// (d2) the same with a workgroup taking a PAIR of adjacent tiles per chunk: tile A's rows read, then tile B's, results of both
// written together at the end of the chunk (2.9 KB contiguous per row and chunk instead of 1.4 KB).
__global__ __launch_bounds__(256, 5) void k_down_pair (const double* __restrict__ l2r, double* __restrict__ r2l, int64_t stride,
                                                        int nt, int ncell, int npairs, int64_t out0, double* out) {
  extern __shared__ double lds[];
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (t >= npairs) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double s = 0;
  for (int ch = 0; ch < (nt + 3)/4; ++ch) {
    int tr = 4*ch + wave; if (tr > nt - 1) tr = nt - 1;
    double r[2][3];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const size_t cell0 = (static_cast<size_t>(2*t + sub))*ncell + 3*lane;
      double v[3][3];
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const double* row = l2r + static_cast<size_t>(1 + 4*tr + f)*stride + cell0;
        const Pair ab = *reinterpret_cast<const Pair*>(row);
        v[f][0] = ab.x; v[f][1] = ab.y; v[f][2] = row[2];
      }
      const double x = r2l[static_cast<size_t>(tr)*stride + out0 + 16*(2*t + sub) + (lane & 15)];
      r[sub][0] = x; r[sub][1] = x; r[sub][2] = x;
#pragma unroll
      for (int f = 0; f < 3; ++f) { r[sub][0] += v[f][0]; r[sub][1] += v[f][1]; r[sub][2] += v[f][2]; }
      lds[threadIdx.x] = r[sub][0];
      __syncthreads();
      r[sub][1] += lds[(threadIdx.x + 64) & 255];
      __syncthreads();
      lds[threadIdx.x] = r[sub][1];
      __syncthreads();
      r[sub][2] += lds[(threadIdx.x + 128) & 255];
    }
    if (3*lane < ncell) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        double* row = r2l + static_cast<size_t>(tr)*stride + (static_cast<size_t>(2*t + sub))*ncell + 3*lane;
        Pair ab; ab.x = r[sub][0]; ab.y = r[sub][1];
        *reinterpret_cast<Pair*>(row) = ab;
        row[2] = r[sub][2];
      }
    }
    s += r[0][2] + r[1][2];
  }
  if (s == 1.2345e-300) out[0] = s;
}

// (e) the leaves->root sweep's traffic: per chunk of 16 rows a wave reads its 4 rows, and the
// workgroup writes one 128-byte line per row -- the tile's 16 outputs -- to slot out0 + 16 t of that row. WRITE 0: no stores;
// 2: the tile's lines as one contiguous block [tile][row][16] in a separate buffer instead.
template <int WRITE>
__global__ __launch_bounds__(256) void k_up (double* __restrict__ p, int64_t stride, int nfields, int ncell, int ntiles,
                                             int64_t out0, double* __restrict__ blk, double* out) {
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
  for (int ch = 0; ch < nchunks; ++ch) {
    const int b = 0;
    load(ch, 0);  // (no chunk in flight: the plain form of this loop streams at the copy rate, pattern (c))
    double q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { q[r] = ab[b][r].x + ab[b][r].y + c[b][r]; s += q[r]; }
    lds[threadIdx.x] = s;
    __syncthreads();
    s += lds[(threadIdx.x + 64) & 255];
    __syncthreads();
    // lane -> (row lane & 3, output lane >> 2): one store instruction = four full 128-byte lines
    const int f = ch*16 + wave + 4*(lane & 3);
    if (WRITE == 1 && f < nfields) p[static_cast<size_t>(f)*stride + out0 + 16*t + (lane >> 2)] = q[lane & 3];
    if (WRITE == 2 && f < nfields) blk[(static_cast<size_t>(t)*nfields + f)*16 + (lane >> 2)] = q[lane & 3];
  }
  if (s == 1.2345e-300) out[0] = s;
}

// (f) the result stores alone: per chunk of 4 tracers wave w writes its tracer's row segment of the tile. MODE 0: lane = 3 cells
// (16 B + 8 B); 1: contiguous 16-byte pieces; tiles of 'ncell' cells 'pitch' cells apart (pitch 192 = line-aligned segments).
template <int MODE>
__global__ __launch_bounds__(256, 5) void k_wronly (double* __restrict__ r2l, int64_t stride, int nt, int ncell, int pitch,
                                                     int ntiles) {
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  if (t >= ntiles) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int ch = 0; ch < (nt + 3)/4; ++ch) {
    int tr = 4*ch + wave; if (tr > nt - 1) tr = nt - 1;
    double* row = r2l + static_cast<size_t>(tr)*stride + static_cast<size_t>(t)*pitch;
    if (MODE == 0) {
      if (3*lane < ncell) {
        Pair ab; ab.x = lane; ab.y = ch;
        *reinterpret_cast<Pair*>(row + 3*lane) = ab;
        row[3*lane + 2] = tr;
      }
    } else {
      for (int i = lane; 2*i + 1 < ncell; i += 64) {
        Pair ab; ab.x = lane; ab.y = ch;
        *reinterpret_cast<Pair*>(row + 2*i) = ab;
      }
    }
    __syncthreads();
  }
}

// (g) the same bytes in CAAS's geometry: a workgroup takes 512 contiguous cells of ONE tracer, reads its rows lo / orig / hi
// (one 16-byte load per row and thread) and writes a result row: out of place (r2l) or in place over 'orig'.
template <int INPLACE>
__global__ __launch_bounds__(256) void k_caaslike (double* __restrict__ l2r, double* __restrict__ r2l, int64_t stride, int n) {
  const int tr = blockIdx.y;
  const size_t i = (static_cast<size_t>(blockIdx.x)*256 + threadIdx.x)*2;
  if (i + 1 >= static_cast<size_t>(n)) return;
  const d2* lo = reinterpret_cast<const d2*>(l2r + static_cast<size_t>(1 + 4*tr)*stride + i);
  const d2* og = reinterpret_cast<const d2*>(l2r + static_cast<size_t>(2 + 4*tr)*stride + i);
  const d2* hi = reinterpret_cast<const d2*>(l2r + static_cast<size_t>(3 + 4*tr)*stride + i);
  const d2 a = __builtin_nontemporal_load(lo), b = __builtin_nontemporal_load(og), c = __builtin_nontemporal_load(hi);
  d2 r; r.x = a.x + b.x + c.x; r.y = a.y + b.y + c.y;
  d2* dst = reinterpret_cast<d2*>((INPLACE ? l2r + static_cast<size_t>(2 + 4*tr)*stride : r2l + static_cast<size_t>(tr)*stride) + i);
  *dst = r;
}

int main () {
  const int ncells = 1500000, nfields = 161, ncell = 183, ntiles = 8192;
  const int64_t stride = (ncells + 8192*16 + 2 + 15)/16*16;  // field-major rows, as the QLT buffer: the cells, then 16 output slots per tile
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
  {
    // QLT buffers of the bench configuration: l2r 161 rows, r2l 40 rows, rows = cells + stage outputs
    const int nt = 40;
    double* r2l;
    CK(hipMalloc(&r2l, static_cast<size_t>(stride)*nt*8)); CK(hipMemset(r2l, 0, static_cast<size_t>(stride)*nt*8));
    double* blk;
    CK(hipMalloc(&blk, static_cast<size_t>(ntiles)*nfields*16*8));
    const double db = static_cast<double>(ntiles)*ncell*(32.0*nt + 8);
    timeit("root->leaves traffic: 3 rows read + 1 written per tracer, 5 WGs/CU", db, [&] { hipLaunchKernelGGL(k_down<1>, dim3(8192), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles, static_cast<int64_t>(ncells), out); });
    timeit("root->leaves traffic, a PAIR of tiles per workgroup (4096 WGs), results written together", db, [&] { hipLaunchKernelGGL(k_down_pair, dim3(4096), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles/2, static_cast<int64_t>(ncells), out); });
    timeit("root->leaves traffic, results as contiguous 16-B pieces via LDS", db, [&] { hipLaunchKernelGGL(k_down<2>, dim3(8192), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles, static_cast<int64_t>(ncells), out); });
    timeit("root->leaves traffic, results IN PLACE over the 'orig' row just read", db, [&] { hipLaunchKernelGGL(k_down<3>, dim3(8192), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles, static_cast<int64_t>(ncells), out); });
    timeit("root->leaves traffic, results into the tracer's 4th row of l2r", db, [&] { hipLaunchKernelGGL(k_down<4>, dim3(8192), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles, static_cast<int64_t>(ncells), out); });
    timeit("root->leaves traffic, 192-cell line-aligned tiles (7812), lane = 3 cells", static_cast<double>(7812)*192*(32.0*nt + 8), [&] { hipLaunchKernelGGL(k_down<1>, dim3(7816), dim3(256), 28*1024, 0, d, r2l, stride, nt, 192, 7812, static_cast<int64_t>(ncells), out, 192); });
    timeit("root->leaves traffic, 192-cell line-aligned tiles, results as contiguous pieces", static_cast<double>(7812)*192*(32.0*nt + 8), [&] { hipLaunchKernelGGL(k_down<2>, dim3(7816), dim3(256), 28*1024, 0, d, r2l, stride, nt, 192, 7812, static_cast<int64_t>(ncells), out, 192); });
    timeit("the same bytes in CAAS's geometry (512 cells of one tracer per WG), out of place", static_cast<double>(ncells)*32.0*nt, [&] { hipLaunchKernelGGL(k_caaslike<0>, dim3((ncells/2 + 255)/256, nt), dim3(256), 0, 0, d, r2l, stride, ncells); });
    timeit("the same bytes in CAAS's geometry, in place over 'orig'", static_cast<double>(ncells)*32.0*nt, [&] { hipLaunchKernelGGL(k_caaslike<1>, dim3((ncells/2 + 255)/256, nt), dim3(256), 0, 0, d, r2l, stride, ncells); });
    const double wb = static_cast<double>(ntiles)*ncell*8.0*nt;
    timeit("result stores alone: lane = 3 cells, 183-cell segments", wb, [&] { hipLaunchKernelGGL(k_wronly<0>, dim3(8192), dim3(256), 28*1024, 0, r2l, stride, nt, ncell, ncell, ntiles); });
    timeit("result stores alone: contiguous 16-B pieces, 183-cell segments", wb, [&] { hipLaunchKernelGGL(k_wronly<1>, dim3(8192), dim3(256), 28*1024, 0, r2l, stride, nt, ncell, ncell, ntiles); });
    timeit("result stores alone: contiguous pieces, 192-cell (line-aligned) segments, 7812 tiles", static_cast<double>(7812)*192*8.0*nt, [&] { hipLaunchKernelGGL(k_wronly<1>, dim3(7816), dim3(256), 28*1024, 0, r2l, stride, nt, 192, 192, 7812); });
    timeit("root->leaves traffic without the result stores", db*24/32, [&] { hipLaunchKernelGGL(k_down<0>, dim3(8192), dim3(256), 28*1024, 0, d, r2l, stride, nt, ncell, ntiles, static_cast<int64_t>(ncells), out); });
    timeit("leaves->root traffic: 161 rows read, no output lines", tb, [&] { hipLaunchKernelGGL(k_up<0>, dim3(8192), dim3(256), 16*1024, 0, d, stride, nfields, ncell, ntiles, static_cast<int64_t>(ncells), blk, out); });
    timeit("leaves->root traffic + one 128-B output line per row and tile", tb, [&] { hipLaunchKernelGGL(k_up<1>, dim3(8192), dim3(256), 16*1024, 0, d, stride, nfields, ncell, ntiles, static_cast<int64_t>(ncells), blk, out); });
    timeit("leaves->root traffic + the tile's lines as one 20 KB block", tb, [&] { hipLaunchKernelGGL(k_up<2>, dim3(8192), dim3(256), 16*1024, 0, d, stride, nfields, ncell, ntiles, static_cast<int64_t>(ncells), blk, out); });
  }
  return 0;
}
