// Hand-written HIP kernels for the CEDR limiter hot path on MI355X (gfx950).
// 64-lane wavefronts, LDS-resident subtrees, field-major (SoA) global layout so
// every global access is a unit-stride stream over cells. No MFMA: the path is
// HBM-bound fp64 elementwise / tree-reduction work. Built with
// -ffp-contract=off: the node problems branch on exact comparisons, so every
// product and sum must round exactly as the reference's (unfused) arithmetic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <type_traits>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <cstdio>

#include "../../include/cedr_hip.h"
#include "cedr_local.hpp"
#include "cedr_node_math.hpp"

namespace {

#ifndef CEDR_XCD_REMAP
# define CEDR_XCD_REMAP 1
#endif
constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kWave = 64;
using namespace cedr_dev;

#ifndef CEDR_REC_STRIDE
# define CEDR_REC_STRIDE 5
#endif
// Doubles per LDS record {lo, orig, hi, X} of the shape-preserving kernel. 5 = one double of padding: with 4 the lanes
// of a wave on consecutive nodes are 32 bytes apart and every fourth collides on an LDS bank (bank conflicts in 45 % of
// the LDS-active cycles, round 2's counters); 40 bytes apart they do not: first-stage sweep 0.457 -> 0.445 ms. (Round 2
// tried four separate arrays instead: slower, it doubles the LDS instructions.)
constexpr int kRS = CEDR_REC_STRIDE;

// ------------------------------------------------------------- tile sweeps
// One workgroup sweeps one tile. It keeps a CHUNK of FT fields (leaves->root)
// or TT tracers (root->leaves) of that tile in LDS and spreads the (node, field)
// or (node, tracer) items of every tree level over all 256 threads, so that the
// small upper levels of a tile still fill the lanes. The workgroup loops over
// chunks; the next chunk's input loads are in flight in registers while the
// current one is processed. The tile's kid indices and level offsets are staged
// in LDS once. LDS rows are pitched (== 1 mod 32 doubles) so that lanes that
// differ in field/tracer hit different banks.
constexpr int kNW = kBlock/kWave;

// Stage kid pairs (packed k0 | k1 << 16) and level offsets of a tile in LDS.
__device__ __forceinline__ void stage_tile_meta (const cedr_hip_stage& st, const cedr_hip_tile& tile,
                                                 uint32_t* kidsL, int32_t* lvlL) {
  const uint32_t* __restrict__ kp =
    reinterpret_cast<const uint32_t*>(st.kids + 2*static_cast<size_t>(tile.kids_off));
  for (int j = threadIdx.x; j < tile.nint; j += kBlock) kidsL[j] = kp[j];
  const int32_t* __restrict__ lp = st.lvlptr + tile.lvl_off;
  for (int j = threadIdx.x; j <= tile.nlvl; j += kBlock) lvlL[j] = lp[j];
}

// Workgroups are dealt round-robin to the 8 XCDs (blockIdx.x % 8), each with its own L2.
// Consecutive tiles share the cache lines in which their row segments meet (a tile's
// ~1.5 KB segment is not line aligned), so every XCD gets a CONTIGUOUS range of tiles:
// the grid has 8*ceil(n/8) workgroups and workgroup b takes tile (b % 8)*ceil(n/8) + b/8
// (the up to 7 workgroups that fall past the last tile exit at once).
__device__ __forceinline__ int xcd_tile (const int ntiles) {
#if CEDR_XCD_REMAP
  const int per = gridDim.x >> 3;
  const int t = (blockIdx.x & 7)*per + (blockIdx.x >> 3);
  return t < ntiles ? t : -1;
#else
  return blockIdx.x;
#endif
}
// The same with the LAST tiles of every XCD's range dealt to several workgroups each. All workgroups of a first-stage
// launch take about the same time (one tile, all chunks: ~60 us at 1.5M cells x 40 tracers) and there are 6.4 rounds of
// them (5.3 for the leaves->root kernel), so towards the end the CUs drain one by one with nothing left to start.
// Workgroups are dispatched in index order; when the tiles past the last FULL round (tail.nf per XCD) are split by
// chunks over tail.k workgroups each, the launch ends with short workgroups that fill the draining CUs. Measured
// (round 4, profiles/r04_experiments.md): root->leaves first stage -1.5 to -2 %, leaves->root -0.3 %.
struct TailSplit { int nf, k; };
struct WorkItem { int tix, c0, cstep; };
__device__ __forceinline__ WorkItem work_item (const int ntiles, const TailSplit tail) {
  WorkItem w{-1, static_cast<int>(blockIdx.y), static_cast<int>(gridDim.y)};
#if CEDR_XCD_REMAP
  const int per = (ntiles + 7) >> 3;
  int t = blockIdx.x >> 3;
  if (tail.k > 1 && t >= tail.nf) {
    const int j = t - tail.nf;
    t = tail.nf + j/tail.k;
    w.c0 = j - (t - tail.nf)*tail.k;
    w.cstep = tail.k;
  }
  const int g = (blockIdx.x & 7)*per + t;
  if (t < per && g < ntiles) w.tix = g;
#else
  w.tix = blockIdx.x;
#endif
  return w;
}
inline unsigned xcd_grid (const int ntiles) {
#if CEDR_XCD_REMAP
  return 8u*static_cast<unsigned>((ntiles + 7)/8);
#else
  return static_cast<unsigned>(ntiles);
#endif
}
inline unsigned xcd_grid_tail (const int ntiles, const TailSplit tail) {
#if CEDR_XCD_REMAP
  const int per = (ntiles + 7)/8;
  if (tail.k > 1 && tail.nf < per) return 8u*static_cast<unsigned>(tail.nf + (per - tail.nf)*tail.k);
#else
  (void) tail;
#endif
  return xcd_grid(ntiles);
}

// Diagnostic builds only (wrong results): -DCEDR_DIAG_L2 makes every tile read its cell records from
// the first 4096 cells of each row, i.e. from cache-resident memory -- the sweep without its HBM reads.
__device__ __forceinline__ int diag_in0 (const int in0) {
#ifdef CEDR_DIAG_L2
  return in0 & 4095;
#else
  return in0;
#endif
}

// LDS accesses of ONE wave execute in program order, so data a lane wrote can be read by another
// lane of the same wave without a workgroup barrier; this only keeps the COMPILER from moving LDS
// accesses across the point (no instruction is generated).
__device__ __forceinline__ void wave_lds_order () {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N> struct Log2 { enum { value = 1 + Log2<N/2>::value }; };
template <> struct Log2<1> { enum { value = 0 }; };

// Leaves -> root. FT fields per chunk (power of 2), NI input loads per lane and
// row (tiles have at most 64*NI inputs), pitch = LDS doubles per field row.
template <int FT, int NI>
__global__ __launch_bounds__(kBlock)
void tree_up_kernel (const cedr_hip_stage st, const int nfields, const uint8_t* __restrict__ op,
                     const int64_t stride, double* __restrict__ data, const int zero_init,
                     const int pitch) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[blockIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  double* const v = lds + (st.max_nint + 1);
  const int nleaf = tile.nleaf;
  const int nchunks = (nfields + FT - 1)/FT;
  const double* __restrict__ src0 = data + tile.in0;
  constexpr int NR = FT/kNW;  // field rows a wave loads per chunk
  // Combine rule per field, staged in LDS (see tree_up_micro_kernel).
  int32_t* const opL = reinterpret_cast<int32_t*>(v + static_cast<size_t>(FT)*pitch);
  for (int i = threadIdx.x; i < nfields; i += kBlock) opL[i] = op[i];

  auto load = [&] (const int c, double (&buf)[NR][NI]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int f = c*FT + wave + kNW*r;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int i = lane + kWave*it;
        buf[r][it] = (f < nfields && i < nleaf) ? src0[static_cast<size_t>(f)*stride + i] : 0.0;
      }
    }
  };

  double cur[NR][NI];
  if (static_cast<int>(blockIdx.y) < nchunks) load(blockIdx.y, cur);
  stage_tile_meta(st, tile, kidsL, lvlL);
  const int ff = threadIdx.x & (FT - 1);  // this thread's field within a chunk
  double* const vf = v + ff*pitch;
  __syncthreads();  // opL (and the tile metadata) staged

  for (int c = blockIdx.y; c < nchunks; c += gridDim.y) {
    double nxt[NR][NI];
    const int cn = c + gridDim.y;
    if (cn < nchunks) load(cn, nxt);
    const int f = c*FT + ff;
    const int o = f < nfields ? opL[f] : 0;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int i = lane + kWave*it;
        if (i < nleaf) v[(wave + kNW*r)*pitch + i] = cur[r][it];
      }
    __syncthreads();
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<FT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<FT>::value);
        const uint32_t kk = kidsL[j];
        const double k0 = vf[kk & 0xffff], k1 = vf[kk >> 16];
        double r;
        if (o == 0) {
          if (zero_init) { r = 0; r += k0; r += k1; }
          else r = k0 + k1;
        } else if (o == 1) r = rmin(k0, k1);
        else r = rmax(k0, k1);
        vf[nleaf + j] = r;
      }
      __syncthreads();
    }
    if (f < nfields)
      for (int k = threadIdx.x >> Log2<FT>::value; k < tile.nout; k += kBlock/FT)
        data[static_cast<size_t>(f)*stride + tile.out + k] = vf[st.outs[tile.outs_off + k]];
    __syncthreads();  // the next chunk overwrites v
    __builtin_amdgcn_sched_barrier(0);  // see tree_up_micro_kernel
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int it = 0; it < NI; ++it) cur[r][it] = nxt[r][it];
  }
}

// Root -> leaves. TT tracers (bulk order) per chunk; per tracer and node an LDS
// record {lo, orig, hi, X} of 4 doubles, tracer rows 'pitch' doubles apart.
// Shared per tile: rhom, 1/rhom and 1/(1/rhom) of every node -- the only
// divisions of the common path besides one per node problem.
template <int TT, int NI>
__global__ __launch_bounds__(kBlock)
void qlt_down_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st,
                      double* __restrict__ l2r, double* __restrict__ r2l, const int pitch) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[blockIdx.x];
  const int MN = st.max_nodes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  double* const rho = lds + (st.max_nint + 1);
  double* const rw = rho + MN;
  double* const rq = rw + MN;
  double* const recs = rq + MN;
  const int nleaf = tile.nleaf;
  const int64_t S = L.stride;
  const bool prefer = L.prefer_mass_con != 0;
  const int nt = L.nt;
  const int nchunks = (nt + TT - 1)/TT;
  constexpr int NR = 3*TT/kNW;  // (tracer, field) rows a wave loads per chunk

  // Row rr = 3*tt + fld of a chunk: fld 0/1/2 = lo/orig/hi of a shapepreserve
  // tracer. Other tracers only have 'orig' (consistent-only: l2r field 1, its q
  // bounds come from the root; nonnegative: l2r field 0).
  auto load = [&] (const int c, double (&buf)[NR][NI]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int rr = wave + kNW*r;
      const int bi = c*TT + rr/3, fld = rr % 3;
      bool want = bi < nt;
      int field = 0;
      if (want) {
        const int pt = L.bidx_prob[bi];
        const bool sp = (pt & SHAPEPRESERVE) != 0;
        want = sp || fld == 1;
        field = L.bidx_l2r[bi] + (sp ? fld : ((pt & CONSISTENT) ? 1 : 0));
      }
      const double* __restrict__ s0 = l2r + static_cast<size_t>(field)*S + tile.in0;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int i = lane + kWave*it;
        buf[r][it] = (want && i < nleaf) ? s0[i] : 0.0;
      }
    }
  };

  double cur[NR][NI];
  if (static_cast<int>(blockIdx.y) < nchunks) load(blockIdx.y, cur);
  {
    const double* __restrict__ src = l2r + tile.in0;
    for (int i = threadIdx.x; i < nleaf; i += kBlock) rho[i] = src[i];
  }
  stage_tile_meta(st, tile, kidsL, lvlL);
  __syncthreads();
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlL[lv], b = lvlL[lv+1];
    for (int j = a + threadIdx.x; j < b; j += kBlock) {
      const uint32_t kk = kidsL[j];
      rho[nleaf + j] = rho[kk & 0xffff] + rho[kk >> 16];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nleaf + tile.nint; i += kBlock) {
    const double w = 1.0/rho[i];
    rw[i] = w;
    rq[i] = 1.0/w;
  }
  const int tt = threadIdx.x & (TT - 1);  // this thread's tracer within a chunk
  double* const R = recs + tt*pitch;

  for (int c = blockIdx.y; c < nchunks; c += gridDim.y) {
    double nxt[NR][NI];
    const int cn = c + gridDim.y;
    if (cn < nchunks) load(cn, nxt);
    const int bi = c*TT + tt;
    const bool live = bi < nt;
    const int pt = live ? L.bidx_prob[bi] : (SHAPEPRESERVE | CONSISTENT);
    const int rb = live ? L.bidx_r2l[bi] : 0;
    const bool sp = (pt & SHAPEPRESERVE) != 0;
    const bool t_only = (pt & CONSISTENT) && ! sp;
    // The tile root's assigned mass, and for a t-only tracer the global q
    // bounds, which travel root->leaves unchanged (cedr_qlt.cpp:549-560).
    double qlo = 0, qhi = 0;
    if (live) {
      if (t_only) {
        qlo = r2l[static_cast<size_t>(rb + 1)*S + tile.out];
        qhi = r2l[static_cast<size_t>(rb + 2)*S + tile.out];
      }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int rr = wave + kNW*r;
      double* const Rr = recs + (rr/3)*pitch + rr % 3;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int i = lane + kWave*it;
        if (i < nleaf) Rr[4*i] = cur[r][it];
      }
    }
    __syncthreads();  // also orders rho/rw/rq before their first use
    // Phase A: re-form the sums of the tile's internal nodes.
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<TT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<TT>::value);
        const uint32_t kk = kidsL[j];
        const double* K0 = R + 4*(kk & 0xffff);
        const double* K1 = R + 4*(kk >> 16);
        double* me = R + 4*(nleaf + j);
        if (sp) { me[0] = K0[0] + K1[0]; me[2] = K0[2] + K1[2]; }
        me[1] = K0[1] + K1[1];
      }
      __syncthreads();
    }
    // Phase B: node problems, level by level from the tile root.
    if (live)
      for (int k = threadIdx.x >> Log2<TT>::value; k < tile.nout; k += kBlock/TT)
        R[4*st.outs[tile.outs_off + k] + 3] = r2l[static_cast<size_t>(rb)*S + tile.out + k];
    __syncthreads();
    for (int lv = tile.nlvl - 1; lv >= 0; --lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<TT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<TT>::value);
        const uint32_t kk = kidsL[j];
        const int c0 = kk & 0xffff, c1 = kk >> 16, me = nleaf + j;
        const KidRho r0{rho[c0], rw[c0], rq[c0]}, r1{rho[c1], rw[c1], rq[c1]};
        const double* P = R + 4*me;
        double* K0 = R + 4*c0;
        double* K1 = R + 4*c1;
        double x0, x1;
        if (t_only)
          solve_node(pt, rho[me], qlo, P[1], qhi, P[3], r0, qlo, K0[1], qhi,
                     r1, qlo, K1[1], qhi, x0, x1, prefer);
        else
          solve_node(pt, rho[me], P[0], P[1], P[2], P[3], r0, K0[0], K0[1], K0[2],
                     r1, K1[0], K1[1], K1[2], x0, x1, prefer);
        K0[3] = x0;
        K1[3] = x1;
      }
      __syncthreads();
    }
    // Phase C: inputs' masses back to HBM, one tracer row per wave pass.
    for (int t2 = wave; t2 < TT; t2 += kNW) {
      const int b2 = c*TT + t2;
      if (b2 >= nt) break;
      const int pt2 = L.bidx_prob[b2], rb2 = L.bidx_r2l[b2];
      const double* R2 = recs + t2*pitch;
      double* __restrict__ dst = r2l + static_cast<size_t>(rb2)*S + tile.in0;
      for (int i = lane; i < nleaf; i += kWave) dst[i] = R2[4*i + 3];
      if ((pt2 & CONSISTENT) && ! (pt2 & SHAPEPRESERVE)) {
        const double ql = r2l[static_cast<size_t>(rb2 + 1)*S + tile.out];
        const double qh = r2l[static_cast<size_t>(rb2 + 2)*S + tile.out];
        double* __restrict__ l0 = l2r + static_cast<size_t>(L.bidx_l2r[b2])*S + tile.in0;
        for (int i = lane; i < nleaf; i += kWave) {
          dst[S + i] = ql; dst[2*S + i] = qh; l0[i] = ql; l0[2*S + i] = qh;
        }
      }
    }
    __syncthreads();  // the next chunk overwrites the records
    __builtin_amdgcn_sched_barrier(0);  // see tree_up_micro_kernel
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int it = 0; it < NI; ++it) cur[r][it] = nxt[r][it];
  }
}

// ------------------------------------------------------------ micro tiles
// First-stage tiles over real cells: lane i of a wave owns micro-subtree i of
// the tile (<= 3 consecutive cells: a, b, c) and combines / solves it in
// registers; only the micro-subtrees' roots and the nodes above them live in
// LDS. That cuts the LDS footprint per cell by ~3 and removes the two largest
// tree levels from the barrier-separated LDS phases.
struct Micro { int start, shape, size; };

// 16-byte access at 8-byte alignment (gfx950 global memory allows it): one
// dwordx4 instruction instead of two dwordx2. The vector-memory pipe's cost is
// per instruction, so halving the instruction count is what matters.
struct __attribute__((packed, aligned(8))) Pair { double x, y; };

// Cells a, b, c of a micro-subtree from a field row: b rides with a in one load.
// The loads are UNCONDITIONAL (cells beyond 'size' are read and discarded; every
// field row is followed by at least two slots of padding, see TreeEngine::stride):
// with no control flow around them the compiler can count them and wait with
// vmcnt(N) for exactly the loads it needs, instead of vmcnt(0) for all in flight.
__device__ __forceinline__ void load_cells (const double* __restrict__ p, const int size,
                                            double& a, double& b, double& c) {
  // Cells beyond 'size' come back as whatever follows in the row; no caller
  // uses them (the shape decides which cells enter a result), and masking them
  // here would make the loads' first use immediate, defeating the prefetch.
  (void) size;
  const Pair ab = *reinterpret_cast<const Pair*>(p);
  a = ab.x;
  b = ab.y;
  c = p[2];
}

__device__ __forceinline__ void store_cells (double* __restrict__ p, const int size,
                                             const double a, const double b, const double c) {
  if (size >= 2) {
    Pair ab; ab.x = a; ab.y = b;
    *reinterpret_cast<Pair*>(p) = ab;
  } else if (size == 1) {
    p[0] = a;
  }
  if (size > 2) p[2] = c;
}

__device__ __forceinline__ Micro decode_micro (const uint32_t d, const bool valid) {
  Micro m{0, -1, 0};
  if (valid) {
    m.start = d & 0xff;
    m.shape = d >> 8;
    m.size = m.shape == 0 ? 1 : (m.shape == 1 ? 2 : 3);
  }
  return m;
}

__device__ __forceinline__ Micro get_micro (const cedr_hip_stage& st, const cedr_hip_tile& tile,
                                            const int lane) {
  Micro m{0, -1, 0};
  if (lane < tile.nleaf) {
    const uint32_t d = st.micro[tile.micro_off + lane];
    m.start = d & 0xff;
    m.shape = d >> 8;
    m.size = m.shape == 0 ? 1 : (m.shape == 1 ? 2 : 3);
  }
  return m;
}

__device__ __forceinline__ double combine (const int o, const int zero_init, const double k0,
                                           const double k1) {
  if (o == 0) {
    if (zero_init) { double r = 0; r += k0; r += k1; return r; }
    return k0 + k1;
  }
  return o == 1 ? rmin(k0, k1) : rmax(k0, k1);
}

// Root value of a micro-subtree: shape 1 = (a,b), 2 = ((a,b),c), 3 = (a,(b,c)).
__device__ __forceinline__ double micro_up (const int shape, const int o, const int zi,
                                            const double a, const double b, const double c) {
  const double t = combine(o, zi, shape == 3 ? b : a, shape == 3 ? c : b);
  if (shape <= 0) return a;
  if (shape == 1) return t;
  return shape == 2 ? combine(o, zi, t, c) : combine(o, zi, a, t);
}

#ifdef CEDR_STAMPS
// Diagnostic build only: per-phase shader-clock totals of the micro kernels
// (down: slots 0-31, up: slots 32-63).
__device__ unsigned long long g_stamps[64];
# define STAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); \
    __builtin_amdgcn_s_waitcnt(0xC07F); acc[i] += _t - t_prev; t_prev = _t; } while (0)
#elif defined(CEDR_MARKS)
// Diagnostic build only: comments in the ISA at the phase boundaries (count instructions between them).
# define STAMP(i) asm volatile("; MARK " #i ::: "memory")
#else
# define STAMP(i)
#endif

typedef double d2v __attribute__((ext_vector_type(2)));

// Software prefetch with inline asm. hipcc sinks ordinary prefetch loads down
// to their first use (the loop-carried copy at the end of an iteration), i.e.
// into the NEXT chunk, where they are waited for at once. asm volatile loads
// stay where they are written; the matching wait ties the destination registers
// ("+v") so that no use can be scheduled before it.
__device__ __forceinline__ void asm_load_cells (d2v& ab, double& c, const double* p) {
  asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16"
               : "=&v"(ab), "=&v"(c) : "v"(p) : "memory");
}
#define CEDR_WAIT_VM0_2(A, C) \
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(C[0]), "+v"(C[1]) :: "memory")
#define CEDR_WAIT_VM0_4(A, C) \
  asm volatile("s_waitcnt vmcnt(0)" \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]) \
               :: "memory")
#define CEDR_WAIT_VM0_8(A, C) \
  asm volatile("s_waitcnt vmcnt(0)" \
               : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(A[4]), "+v"(A[5]), "+v"(A[6]), \
                 "+v"(A[7]), "+v"(C[0]), "+v"(C[1]), "+v"(C[2]), "+v"(C[3]), "+v"(C[4]), "+v"(C[5]), \
                 "+v"(C[6]), "+v"(C[7]) :: "memory")

// The same for a plain sum (the rule of almost every field): one add for the
// inner pair, one for the root, operands picked by selects -- no branches.
// fp addition commutes, so ((a,b),c) and (a,(b,c)) share the second add.
__device__ __forceinline__ double micro_up_sum (const int shape, const double a, const double b,
                                                const double c) {
  const double t = (shape == 3 ? b : a) + (shape == 3 ? c : b);
  const double r = t + (shape == 2 ? c : a);
  return shape <= 0 ? a : (shape == 1 ? t : r);
}

// The same when every node is 0 + kid0 + kid1 (BfbTreeAllReducer, cedr_bfb_tree_allreduce.cpp:117-123): the leading
// zero stays -- 0 + (-0) is +0, so dropping it would change the sign of a zero.
__device__ __forceinline__ double micro_up_sum_zi (const int shape, const double a, const double b,
                                                   const double c) {
  const double t = (0.0 + (shape == 3 ? b : a)) + (shape == 3 ? c : b);
  const double r = (0.0 + (shape == 2 ? t : a)) + (shape == 2 ? c : t);
  return shape <= 0 ? a : (shape == 1 ? t : r);
}

// ROWS: the tile's INPUT rows come from a table of per-field base pointers
// (in_rows[f] + slot) instead of data + f*stride + slot -- the caller's own
// [tracer][cell] arrays in zero-copy mode. Outputs always go to 'data'.
// An unoptimised build keeps values in memory between statements: do not leave asm loads in flight there.
#ifdef CEDR_DIAG_UP_BLOCKSTORE2
__device__ double* g_diag_blk;
#endif
#if ! defined(__OPTIMIZE__) && ! defined(CEDR_UP_NOPREFETCH)
# define CEDR_UP_NOPREFETCH
#endif
// ZI (with SUMS): every node is 0 + kid0 + kid1 instead (the BFB reducer's sums).
// SUMS: the caller guarantees that every field's rule is a plain sum and zero_init is 0 (QLT without
// consistent-only tracers -- E3SM's case): no rule table, no per-field branches in the chunk loop.
template <int FT, bool ROWS, bool SUMS, bool ZI = false>
__global__ __launch_bounds__(kBlock)
void tree_up_micro_kernel (const cedr_hip_stage st, const int nfields,
                           const uint8_t* __restrict__ op, const int64_t stride,
                           double* __restrict__ data, const double* const* __restrict__ in_rows,
                           const int nrow, const int zero_init, const int pitch, const TailSplit tail) {
  static_assert(FT == 8*kNW || FT == 4*kNW || FT == 2*kNW,
                "the prefetch wait macros are written for 2, 4 or 8 rows per wave");
  extern __shared__ double lds[];
  const WorkItem wi = work_item(st.ntiles, tail);
  const int tix = wi.tix;
  if (tix < 0) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  // (uniform stages: the micro-subtree descriptor does not wait for the tile descriptor)
  const bool uni = (st.flags & CEDR_STAGE_UNIFORM64) != 0;
  const uint32_t dpre = uni ? st.micro[64*static_cast<size_t>(tix) + lane] : 0;
  const cedr_hip_tile tile = st.tiles[tix];
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  double* const v = lds + (st.max_nint + 1);
  const int nleaf = tile.nleaf;
  const int nchunks = (nfields + FT - 1)/FT;
  const Micro m = uni ? decode_micro(dpre, true) : get_micro(st, tile, lane);
  // ROWS: a caller's row (nrow cells) has no padding behind its last cell, so a lane whose
  // 3-cell load window would reach past it reads the window cshift cells earlier and
  // picks its cells from the later window positions (fix_window).
  const int cshift = ROWS ? max(0, tile.in0 + m.start + 2 - (nrow - 1)) : 0;
  const int64_t cell0 = diag_in0(tile.in0) + m.start - cshift;
  const double* __restrict__ src0 = data + cell0;
  constexpr int NR = FT/kNW;
  // The per-field combine rule, the tile's output node ids and (ROWS) the row
  // table, staged in LDS: any ordinary global load inside the chunk loop would
  // make the chunk wait (vmcnt is in-order) for the next chunk's prefetch issued
  // just before.
  const double** const rowsL = reinterpret_cast<const double**>(v + static_cast<size_t>(FT)*pitch);
  int32_t* const opL = reinterpret_cast<int32_t*>(rowsL + (ROWS ? nfields : 0));
  int32_t* const outL = opL + nfields;
  for (int i = threadIdx.x; i < nfields; i += kBlock) opL[i] = op[i];
  for (int i = threadIdx.x; i < tile.nout; i += kBlock) outL[i] = st.outs[tile.outs_off + i];
  if (ROWS) for (int i = threadIdx.x; i < nfields; i += kBlock) rowsL[i] = in_rows[i];

  // Cells past a micro-subtree's size are loaded and ignored (the shape decides
  // what enters a result); rows past the last field re-read the last one.
  const double* const plast = src0 + static_cast<size_t>(nfields - 1)*stride;
  // 'staged': the LDS copy of the row table is visible (not yet for the first chunk)
  // 'more' false (no next chunk): the loads still happen -- the count of loads in flight stays
  // static -- but all from field 0's row, a single row of this tile that is in cache, instead of
  // re-reading a whole chunk (that was +9 % load traffic through the CU's load path at 11 chunks).
  auto load = [&] (const int c, d2v (&ab)[NR], double (&cc)[NR], const bool staged, const bool more) {
    if (ROWS) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        int f = c*FT + wave + kNW*r;
        if (f > nfields - 1) f = nfields - 1;
        if ( ! more) f = 0;
        asm_load_cells(ab[r], cc[r], (staged ? rowsL[f] : in_rows[f]) + cell0);
      }
    } else {
      const double* prow = src0 + static_cast<size_t>(c*FT + wave)*stride;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const double* p = prow <= plast ? prow : plast;
        asm_load_cells(ab[r], cc[r], more ? p : src0);
        prow += static_cast<size_t>(kNW)*stride;
      }
    }
  };

#ifdef CEDR_STAMPS
  unsigned long long acc[8] = {0,0,0,0,0,0,0,0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  d2v cab[NR], nab[NR];
  double ccc[NR], ncc[NR];
  load(wi.c0, cab, ccc, false, true);  // wi.c0 < nchunks by construction of the grid
  stage_tile_meta(st, tile, kidsL, lvlL);
  if constexpr (FT == 8*kNW) { CEDR_WAIT_VM0_8(cab, ccc); }
  else if constexpr (FT == 4*kNW) { CEDR_WAIT_VM0_4(cab, ccc); }
  else { CEDR_WAIT_VM0_2(cab, ccc); }
  __syncthreads();  // opL, outL and the tile metadata staged
  STAMP(0);  // setup + first chunk's loads

  for (int c = wi.c0; c < nchunks; c += wi.cstep) {
    const int cn = c + wi.cstep;
#ifdef CEDR_UP_NOPREFETCH  // experiment: no chunk in flight during the combines
    if (c != wi.c0) {
      load(c, cab, ccc, true, true);
      if constexpr (FT == 8*kNW) { CEDR_WAIT_VM0_8(cab, ccc); }
      else if constexpr (FT == 4*kNW) { CEDR_WAIT_VM0_4(cab, ccc); }
      else { CEDR_WAIT_VM0_2(cab, ccc); }
    }
    (void) cn;
#else
    load(cn < nchunks ? cn : c, nab, ncc, true, cn < nchunks);
#endif
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int fr = c*FT + wave + kNW*r;
      const int orow = (SUMS || fr >= nfields) ? 0 : opL[fr];
      double wa = cab[r].x, wb = cab[r].y;
      const double wc = ccc[r];
      if (ROWS) {  // window shifted back by cshift: cell a sits at position cshift, b at cshift + 1
        wa = cshift == 0 ? wa : (cshift == 1 ? wb : wc);
        wb = cshift == 0 ? wb : wc;
      }
      const double val = (SUMS && ZI) ? micro_up_sum_zi(m.shape, wa, wb, wc) :
        ((SUMS || (orow == 0 && ! zero_init)) ?
         micro_up_sum(m.shape, wa, wb, wc) : micro_up(m.shape, orow, zero_init, wa, wb, wc));
      if (lane < nleaf) v[(wave + kNW*r)*pitch + lane] = val;
    }
    STAMP(1);  // issue prefetch + micro combines
    // The LDS tree above the micro-subtree roots, for the NR rows this wave loaded, by this wave
    // alone: (node, row) items over its 64 lanes, level after level in program order -- a wave's
    // LDS accesses execute in order, so no workgroup barrier is needed anywhere in the chunk loop
    // and the four waves of the workgroup run independently of each other.
    constexpr int LR = Log2<NR>::value;
    const int rr = lane & (NR - 1);                  // this lane's row among the wave's NR
    double* const vr = v + (wave + kNW*rr)*pitch;
    const int fl = c*FT + wave + kNW*rr;             // its field
    const int o = (SUMS || fl >= nfields) ? 0 : opL[fl];
#ifdef CEDR_DIAG_UP_NOLEVELS  // (diagnostic builds: the sweep without its LDS levels)
    for (int lv = 0; lv < 0; ++lv) {
#else
    for (int lv = 0; lv < tile.nlvl; ++lv) {
#endif
      wave_lds_order();
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << LR;
      for (int q = lane; q < nitems; q += kWave) {
        const int j = a + (q >> LR);
        const uint32_t kk = kidsL[j];
        if (SUMS && ZI) vr[nleaf + j] = (0.0 + vr[kk & 0xffff]) + vr[kk >> 16];
        else if (SUMS) vr[nleaf + j] = vr[kk & 0xffff] + vr[kk >> 16];
        else vr[nleaf + j] = combine(o, zero_init, vr[kk & 0xffff], vr[kk >> 16]);
      }
    }
    wave_lds_order();
    STAMP(3);  // LDS levels
    // The wait for the prefetched chunk comes BEFORE this chunk's output stores are issued: vmcnt
    // counts stores too, so waiting after them would also wait for their write acknowledgements.
#ifndef CEDR_UP_NOPREFETCH
    if constexpr (FT == 8*kNW) { CEDR_WAIT_VM0_8(nab, ncc); }
    else if constexpr (FT == 4*kNW) { CEDR_WAIT_VM0_4(nab, ncc); }
    else { CEDR_WAIT_VM0_2(nab, ncc); }
#endif
    STAMP(5);  // wait for the prefetched chunk
#ifdef CEDR_DIAG_UP_BLOCKSTORE2  // (diagnostic builds, wrong results: outputs tile-major into their own 169 MB buffer)
    if (fl < nfields)
      for (int k = lane >> LR; k < tile.nout; k += kWave/NR)
        g_diag_blk[(static_cast<size_t>(tix)*nfields + fl)*16 + k] = vr[outL[k]];
#elif defined(CEDR_DIAG_UP_BLOCKSTORE)  // (the same into a 13 MB window that wraps around: the stores stay in cache)
    if (fl < nfields)
      for (int k = lane >> LR; k < tile.nout; k += kWave/NR)
        data[static_cast<size_t>(160)*stride + static_cast<size_t>(tix)*nfields*16 % (stride - 4096) + fl*16 + k] = vr[outL[k]];
#elif defined(CEDR_DIAG_UP_ONELINE)  // (diagnostic builds, wrong results: every tile writes the SAME 161 lines: the stores without their memory traffic)
    if (fl < nfields)
      for (int k = lane >> LR; k < tile.nout; k += kWave/NR)
        data[static_cast<size_t>(fl)*stride + st.tiles[0].out + 16*(blockIdx.x & 7) + k] = vr[outL[k]];
#elif defined(CEDR_DIAG_UP_EIGHTH)  // (diagnostic builds, wrong results: one tile in eight writes its outputs)
    if (fl < nfields && (tix & 7) == 0)
      for (int k = lane >> LR; k < tile.nout; k += kWave/NR)
        data[static_cast<size_t>(fl)*stride + tile.out + k] = vr[outL[k]];
#elif ! defined(CEDR_DIAG_UP_NOSTORE)  // (diagnostic builds: the sweep without its output stores)
    if (fl < nfields)
      for (int k = lane >> LR; k < tile.nout; k += kWave/NR)
        data[static_cast<size_t>(fl)*stride + tile.out + k] = vr[outL[k]];
#else
    if (fl < nfields && vr[outL[0]] == 1.2345e-300) data[0] = 0;
#endif
    wave_lds_order();  // the next chunk's combines overwrite these rows
    STAMP(4);  // outputs
#ifndef CEDR_UP_NOPREFETCH
#pragma unroll
    for (int r = 0; r < NR; ++r) { cab[r] = nab[r]; ccc[r] = ncc[r]; }
#endif
  }
#ifdef CEDR_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[32 + wave*8 + i], acc[i]);
#endif
}

// A micro-subtree by ROLES instead of cell positions: t = (u, v) is its inner
// pair and p the third cell, so ((a,b),c) has (u,v,p) = (a,b,c) and root kids
// (t,p); (a,(b,c)) has (u,v,p) = (b,c,a) and root kids (p,t); (a,b) is t alone;
// a single cell is u alone. Addressing cells by role (offsets oUV / oP from the
// tile's first cell) keeps every per-lane shape decision out of the data path:
// no selects on loaded values, and the inner problem is the same code for all
// shapes.
struct Roles { int oUV, oP, shape; };

__device__ __forceinline__ Roles get_roles (const Micro& m) {
  Roles r;
  r.shape = m.shape;
  r.oUV = m.start + (m.shape == 3 ? 1 : 0);
  r.oP = m.shape == 3 ? m.start : (m.shape == 2 ? m.start + 2 : m.start);
  return r;
}

// u and v ride in one 16-byte load, p in an 8-byte one; unconditional like
// load_cells (roles a shape does not have read a neighbouring slot and are ignored).
__device__ __forceinline__ void load_roles (const double* __restrict__ row, const Roles& r,
                                            double& u, double& v, double& p) {
  const Pair uv = *reinterpret_cast<const Pair*>(row + r.oUV);
  u = uv.x;
  v = uv.y;
  p = row[r.oP];
}

// The same on a CALLER's row, which unlike the internal rows has no padding behind
// its last cell: a lane whose (u, v) pair would reach past the row reads the pair
// one cell earlier and takes its second half (ushift = 1). Still unconditional.
__device__ __forceinline__ void load_roles_edge (const double* __restrict__ row, const Roles& r,
                                                 const int ushift, double& u, double& v, double& p) {
  const Pair uv = *reinterpret_cast<const Pair*>(row + r.oUV - ushift);
  u = ushift ? uv.y : uv.x;
  v = uv.y;
  p = row[r.oP];
}

__device__ __forceinline__ void store_roles (double* __restrict__ row, const Roles& r,
                                             const double u, const double v, const double p) {
  if (r.shape >= 1) {
    Pair uv; uv.x = u; uv.y = v;
    *reinterpret_cast<Pair*>(row + r.oUV) = uv;
  } else if (r.shape == 0) {
    row[r.oUV] = u;
  }
  if (r.shape >= 2) row[r.oP] = p;
}

// The node problems of a micro-subtree, in registers: the root's (shapes 2, 3),
// then the inner pair's (shapes 1, 2, 3; for shape 1 the pair IS the root).
// cur[role u,v,p][lo, orig, hi]; T and RR are the records of t and of the root.
// any2 / any3: whether any lane of the wave has shape 2 / 3 (wave-uniform), so a
// tree that only produces one of them pays for one root problem, not two.
// TONLY (a tracer that carries the global q bounds instead of its cells' own) is
// a compile-time flag; callers branch on it wave-uniformly.
template <bool TONLY>
__device__ __forceinline__
void micro_solves (const int shape, const bool any2, const bool any3, const int pt, const double rr,
                   const KidRho ku, const KidRho kv, const KidRho kp, const KidRho kt,
                   const double (&cur)[3][3], const double (&T)[3], const double (&RR)[3],
                   const double Xr, const double qlw, const double qhw, const bool prefer,
                   double& Xu, double& Xv, double& Xp) {
#define CEDR_LO(x) (TONLY ? qlw : (x))
#define CEDR_HI(x) (TONLY ? qhw : (x))
  double Xt = Xr;  // shape 1: the pair is the root
  if (any2 && shape == 2)
    solve_node(pt, rr, CEDR_LO(RR[0]), RR[1], CEDR_HI(RR[2]), Xr,
               kt, CEDR_LO(T[0]), T[1], CEDR_HI(T[2]),
               kp, CEDR_LO(cur[2][0]), cur[2][1], CEDR_HI(cur[2][2]), Xt, Xp, prefer);
  if (any3 && shape == 3)
    solve_node(pt, rr, CEDR_LO(RR[0]), RR[1], CEDR_HI(RR[2]), Xr,
               kp, CEDR_LO(cur[2][0]), cur[2][1], CEDR_HI(cur[2][2]),
               kt, CEDR_LO(T[0]), T[1], CEDR_HI(T[2]), Xp, Xt, prefer);
  Xu = Xr;  // shape 0
  if (shape >= 1)
    solve_node(pt, kt.rh, CEDR_LO(T[0]), T[1], CEDR_HI(T[2]), Xt,
               ku, CEDR_LO(cur[0][0]), cur[0][1], CEDR_HI(cur[0][2]),
               kv, CEDR_LO(cur[1][0]), cur[1][1], CEDR_HI(cur[1][2]), Xu, Xv, prefer);
#undef CEDR_LO
#undef CEDR_HI
}


#ifndef CEDR_PRIO_LOADS
# define CEDR_PRIO_LOADS 1
#endif
#ifndef CEDR_PRIO_LDS
# define CEDR_PRIO_LDS 2
#endif
constexpr int kPrioLoads = CEDR_PRIO_LOADS, kPrioLdsLevels = CEDR_PRIO_LDS;  // (both 0: the experiment's baseline)
// Root -> leaves over micro tiles. 4 tracers per chunk, one per wave for the
// register part; the LDS part spreads (node, tracer) items over all threads.
#ifndef CEDR_MICRO_WAVES
# define CEDR_MICRO_WAVES 3
#endif
#ifndef CEDR_MICRO_TW
# define CEDR_MICRO_TW 2  // tracers per wave and chunk in the micro down sweep
#endif
constexpr int kCellRho = 12;  // LDS doubles per micro-subtree: {rh, w, q} of u, v, p, t
// ROWS (zero-copy mode, first stage): the cells' input rows come from in_rows[l2r field]
// and their results go to out_rows[bulk tracer index] instead of l2r / r2l rows.
#ifndef CEDR_MICRO_WAVES1
# define CEDR_MICRO_WAVES1 3  // ... and for its one-tracer-per-wave build
#endif
template <int TW, bool ROWS>
__global__ __launch_bounds__(kBlock, TW == 1 ? CEDR_MICRO_WAVES1 : CEDR_MICRO_WAVES)
void qlt_down_micro_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st,
                            double* __restrict__ l2r, double* __restrict__ r2l,
                            const double* const* __restrict__ in_rows,
                            double* const* __restrict__ out_rows, const int nrow, const int pitch) {
  constexpr int TT = kNW*TW;  // tracers per chunk: TW per wave for the register part
  extern __shared__ double lds[];
  const int tix = xcd_tile(st.ntiles);
  if (tix < 0) return;
  const cedr_hip_tile tile = st.tiles[tix];
  const int MN = st.max_nodes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  double* const rho = lds + (st.max_nint + 1);
  double* const rw = rho + MN;
  double* const rq = rw + MN;
  double* const cellrho = rq + MN;            // [kCellRho][kWave]
  double* const recs = cellrho + kCellRho*kWave;
  // row tables staged in LDS (after the records), see tree_up_micro_kernel
  const double** const inL = reinterpret_cast<const double**>(recs + static_cast<size_t>(TT)*pitch);
  double** const outL = reinterpret_cast<double**>(recs + static_cast<size_t>(TT)*pitch) + (ROWS ? L.l2rndps : 0);
  const int nleaf = tile.nleaf;
  const int64_t S = L.stride;
  const bool prefer = L.prefer_mass_con != 0;
  const int nt = L.nt;
  const int nchunks = (nt + TT - 1)/TT;
  const Micro m = get_micro(st, tile, lane);
  const Roles ro = get_roles(m);
  const bool any2 = __ballot(m.shape == 2) != 0, any3 = __ballot(m.shape == 3) != 0;
  // ROWS: nrow = cells per caller row; see load_roles_edge
  const int ushift = (ROWS && tile.in0 + ro.oUV + 1 > nrow - 1) ? 1 : 0;

  // Input records {lo, orig, hi} of this lane's cells for tracer bi (see
  // qlt_down_kernel for which l2r fields they are).
  // 'staged': the LDS copy of the row table is visible (not yet for the first chunk)
  auto load = [&] (const int bi, double (&buf)[3][3], const bool staged) {
    bool sp = false;
    int field = 0;
    if (bi < nt) {
      const int pt = L.bidx_prob[bi];
      sp = (pt & SHAPEPRESERVE) != 0;
      field = L.bidx_l2r[bi] + (sp ? 0 : ((pt & CONSISTENT) ? 1 : 0));
    }
    if (ROWS) {
      const double* const* rows = staged ? inL : in_rows;
      // (__restrict__ locals: the rows are never written by this kernel's stores, which lets
      // the compiler schedule these loads as freely as those from l2r in the packed variant)
      const double* __restrict__ const r0 = rows[field] + tile.in0;
      if (sp) {
        const double* __restrict__ const r1 = rows[field + 1] + tile.in0;
        const double* __restrict__ const r2 = rows[field + 2] + tile.in0;
        load_roles_edge(r0, ro, ushift, buf[0][0], buf[1][0], buf[2][0]);
        load_roles_edge(r1, ro, ushift, buf[0][1], buf[1][1], buf[2][1]);
        load_roles_edge(r2, ro, ushift, buf[0][2], buf[1][2], buf[2][2]);
      } else {
        load_roles_edge(r0, ro, ushift, buf[0][1], buf[1][1], buf[2][1]);
        buf[0][0] = buf[1][0] = buf[2][0] = 0.0;
        buf[0][2] = buf[1][2] = buf[2][2] = 0.0;
      }
      return;
    }
    const double* __restrict__ s0 = l2r + static_cast<size_t>(field)*S + diag_in0(tile.in0);
    if (sp) {
      load_roles(s0, ro, buf[0][0], buf[1][0], buf[2][0]);
      load_roles(s0 + S, ro, buf[0][1], buf[1][1], buf[2][1]);
      load_roles(s0 + 2*S, ro, buf[0][2], buf[1][2], buf[2][2]);
    } else {
      load_roles(s0, ro, buf[0][1], buf[1][1], buf[2][1]);
      buf[0][0] = buf[1][0] = buf[2][0] = 0.0;
      buf[0][2] = buf[1][2] = buf[2][2] = 0.0;
    }
  };

#ifdef CEDR_STAMPS
  unsigned long long acc[8] = {0,0,0,0,0,0,0,0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  double cur[TW][3][3];  // [tracer of this wave][role u,v,p][lo, orig, hi]
#pragma unroll
  for (int s = 0; s < TW; ++s) load(blockIdx.y*TT + wave + kNW*s, cur[s], false);
  if (ROWS) {
    for (int i = threadIdx.x; i < L.l2rndps; i += kBlock) inL[i] = in_rows[i];
    for (int i = threadIdx.x; i < nt; i += kBlock) outL[i] = out_rows[i];
  }

  // Densities of this lane's micro-subtree: roles u, v, p, the inner pair t and
  // the root, with 1/rhom and 1/(1/rhom) of everything that can be a kid.
  // They are the same for every wave and tracer: wave 0 derives them once per tile
  // and parks them in LDS, so they hold registers only while a micro-subtree is
  // being solved: 156 VGPRs and no spills instead of 168 plus spills (3 waves/SIMD
  // either way), and the eight divisions per micro-subtree happen once, not per wave.
  if (wave == 0) {
    double rh[4];
    if (ROWS) load_roles_edge(in_rows[0] + tile.in0, ro, ushift, rh[0], rh[1], rh[2]);
    else load_roles(l2r + tile.in0, ro, rh[0], rh[1], rh[2]);
    if (m.shape < 0) rh[0] = 1.0;
    if (m.shape < 1) rh[1] = 1.0;
    if (m.shape < 2) rh[2] = 1.0;
    rh[3] = rh[0] + rh[1];
    // (a,b)+c and a+(b,c) are the same sum: fp addition commutes
    const double rr = m.shape <= 0 ? rh[0] : (m.shape == 1 ? rh[3] : rh[3] + rh[2]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double w = 1.0/rh[r];
      cellrho[(3*r    )*kWave + lane] = rh[r];
      cellrho[(3*r + 1)*kWave + lane] = w;
      cellrho[(3*r + 2)*kWave + lane] = 1.0/w;
    }
    if (lane < nleaf) rho[lane] = rr;
  }
  stage_tile_meta(st, tile, kidsL, lvlL);
  __syncthreads();
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlL[lv], b = lvlL[lv+1];
    for (int j = a + threadIdx.x; j < b; j += kBlock) {
      const uint32_t kk = kidsL[j];
      rho[nleaf + j] = rho[kk & 0xffff] + rho[kk >> 16];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nleaf + tile.nint; i += kBlock) {
    const double w = 1.0/rho[i];
    rw[i] = w;
    rq[i] = 1.0/w;
  }
  const int tt = threadIdx.x & (TT - 1);  // tracer of this thread in the LDS phases
  double* const R = recs + tt*pitch;      // its record row
  STAMP(0);  // tile setup (rho, meta)

  for (int c = blockIdx.y; c < nchunks; c += gridDim.y) {
    if (c != static_cast<int>(blockIdx.y)) {
#pragma unroll
      for (int s = 0; s < TW; ++s) load(c*TT + wave + kNW*s, cur[s], true);
    }

    // (wave priorities by phase as in qlt_down_sp_kernel: sums and level sums 1, LDS node problems 2, register phase 0)
    __builtin_amdgcn_s_setprio(kPrioLoads);
    // ---- this wave's tracers: micro-subtree sums in registers, root records to LDS
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      double* const Rw = recs + (wave + kNW*s)*pitch;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        const double t = cur[s][0][f] + cur[s][1][f];
        const double r = m.shape <= 0 ? cur[s][0][f] : (m.shape == 1 ? t : t + cur[s][2][f]);
        if (lane < nleaf) Rw[kRS*lane + f] = r;
      }
    }
    STAMP(1);  // loads arrived + micro sums

    // ---- the LDS phases' tracer of this thread
    const int bi = c*TT + tt;
    const bool live = bi < nt;
    const int pt = live ? L.bidx_prob[bi] : (SHAPEPRESERVE | CONSISTENT);
    const int rb = live ? L.bidx_r2l[bi] : 0;
    const bool sp = (pt & SHAPEPRESERVE) != 0;
    const bool t_only = (pt & CONSISTENT) && ! sp;
    double qlo = 0, qhi = 0;
    if (live) {
      if (t_only) {
        qlo = r2l[static_cast<size_t>(rb + 1)*S + tile.out];
        qhi = r2l[static_cast<size_t>(rb + 2)*S + tile.out];
      }
    }
    // This thread's output masses (a micro tile has at most 64 outputs),
    // requested now so that their latency is covered by phase A.
    constexpr int NO = (kWave*TT + kBlock - 1)/kBlock;
    const int kout = threadIdx.x >> Log2<TT>::value;
    double Xout[NO];
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int k = kout + i*(kBlock/TT);
      Xout[i] = (live && k < tile.nout) ? r2l[static_cast<size_t>(rb)*S + tile.out + k] : 0.0;
    }
    __syncthreads();  // also orders rho/rw/rq before their first use
    STAMP(2);  // first barrier
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<TT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<TT>::value);
        const uint32_t kk = kidsL[j];
        const double* K0 = R + kRS*(kk & 0xffff);
        const double* K1 = R + kRS*(kk >> 16);
        double* me = R + kRS*(nleaf + j);
        if (sp) { me[0] = K0[0] + K1[0]; me[2] = K0[2] + K1[2]; }
        me[1] = K0[1] + K1[1];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int k = kout + i*(kBlock/TT);
      if (live && k < tile.nout) R[kRS*st.outs[tile.outs_off + k] + 3] = Xout[i];
    }
    __syncthreads();
    STAMP(3);  // phase A + outputs' masses
    __builtin_amdgcn_s_setprio(kPrioLdsLevels);
    for (int lv = tile.nlvl - 1; lv >= 0; --lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<TT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<TT>::value);
        const uint32_t kk = kidsL[j];
        const int c0 = kk & 0xffff, c1 = kk >> 16, me = nleaf + j;
        const KidRho r0{rho[c0], rw[c0], rq[c0]}, r1{rho[c1], rw[c1], rq[c1]};
        const double* P = R + kRS*me;
        double* K0 = R + kRS*c0;
        double* K1 = R + kRS*c1;
        double x0, x1;
        if (t_only)
          solve_node(pt, rho[me], qlo, P[1], qhi, P[3], r0, qlo, K0[1], qhi,
                     r1, qlo, K1[1], qhi, x0, x1, prefer);
        else
          solve_node(pt, rho[me], P[0], P[1], P[2], P[3], r0, K0[0], K0[1], K0[2],
                     r1, K1[0], K1[1], K1[2], x0, x1, prefer);
        K0[3] = x0;
        K1[3] = x1;
      }
      __syncthreads();
    }

    __builtin_amdgcn_s_setprio(0);
    STAMP(4);  // phase B
    // ---- this wave's tracers again: node problems of the micro-subtree in
    // registers, then the cells' masses to HBM.
    KidRho ku, kv, kp, kt;
    ku.rh = cellrho[0*kWave + lane]; ku.w = cellrho[1*kWave + lane]; ku.q = cellrho[2*kWave + lane];
    kv.rh = cellrho[3*kWave + lane]; kv.w = cellrho[4*kWave + lane]; kv.q = cellrho[5*kWave + lane];
    kp.rh = cellrho[6*kWave + lane]; kp.w = cellrho[7*kWave + lane]; kp.q = cellrho[8*kWave + lane];
    kt.rh = cellrho[9*kWave + lane]; kt.w = cellrho[10*kWave + lane]; kt.q = cellrho[11*kWave + lane];
    const double rr = rho[lane < nleaf ? lane : 0];
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      const int biw = c*TT + wave + kNW*s;
      if (biw >= nt || lane >= nleaf) continue;
      const int ptw = L.bidx_prob[biw], rbw = L.bidx_r2l[biw];
      const bool t_onlyw = (ptw & CONSISTENT) && ! (ptw & SHAPEPRESERVE);
      double qlw = 0, qhw = 0;
      if (t_onlyw) {
        qlw = r2l[static_cast<size_t>(rbw + 1)*S + tile.out];
        qhw = r2l[static_cast<size_t>(rbw + 2)*S + tile.out];
      }
      double T[3], RR[3];  // inner pair t and micro root: {lo, orig, hi} sums, re-formed
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        T[f] = cur[s][0][f] + cur[s][1][f];
        RR[f] = m.shape <= 0 ? cur[s][0][f] : (m.shape == 1 ? T[f] : T[f] + cur[s][2][f]);
      }
      const double Xr = recs[(wave + kNW*s)*pitch + kRS*lane + 3];
      double Xu = Xr, Xv = 0, Xp = 0;
      if (t_onlyw)
        micro_solves<true>(m.shape, any2, any3, ptw, rr, ku, kv, kp, kt, cur[s], T, RR, Xr, qlw, qhw, prefer,
                           Xu, Xv, Xp);
      else
        micro_solves<false>(m.shape, any2, any3, ptw, rr, ku, kv, kp, kt, cur[s], T, RR, Xr, qlw, qhw, prefer,
                            Xu, Xv, Xp);
      {  // (a cell's result is never read back by this kernel, also when Qm_out is Qm itself)
        double* __restrict__ const orow = (ROWS ? outL[biw] : r2l + static_cast<size_t>(rbw)*S) + tile.in0;
        store_roles(orow, ro, Xu, Xv, Xp);
      }
      if (t_onlyw) {
        double* __restrict__ dst = r2l + static_cast<size_t>(rbw)*S + tile.in0 + m.start;
        double* __restrict__ l0p = l2r + static_cast<size_t>(L.bidx_l2r[biw])*S + tile.in0 + m.start;
        for (int k = 0; k < m.size; ++k) {
          dst[S + k] = qlw; dst[2*S + k] = qhw; l0p[k] = qlw; l0p[2*S + k] = qhw;
        }
      }
    }
    STAMP(5);  // register node problems + stores issued
  }
#ifdef CEDR_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[wave*8 + i], acc[i]);
#endif
}

// ---------------------------------------------------------------------------
// Root -> leaves over micro tiles when EVERY tracer is shape preserving (E3SM's
// case: all cst). Same tile walk and the same node problems as
// qlt_down_micro_kernel; what differs follows from what was measured on that kernel
// (profiles/r02_down_experiments.md):
//  * With the cell records read from cache instead of HBM the sweep is as slow: it is
//    bound by the node problems' instruction issue (a wave alone issues one fp64
//    instruction per ~6-8 cycles; the SIMD can take one per 4), not by memory. So the
//    kernel is built for OCCUPANCY: one tracer per wave, no prefetch registers, the
//    problem type fixed at compile time (half the code, ~115 VGPRs: 4 waves per SIMD).
//    A register prefetch of the next chunk (inline asm, as tree_up_micro_kernel has) cost
//    more in registers than the memory latency it hid: 0.64 vs 0.56 ms.
//  * The chain of dependent tree levels per chunk sets how well workgroups overlap, so
//    the sums of the LDS levels are formed by the wave that owns the tracer, without
//    workgroup barriers (a wave's LDS accesses execute in program order), and the plan
//    cuts first-stage tiles to two LDS levels.
//  * The densities of a lane's micro-subtree are derived by all four waves, one role each.
#ifndef CEDR_DOWN_SP_WAVES2
# define CEDR_DOWN_SP_WAVES2 3  // waves per SIMD the 2-tracers-per-wave build is compiled for
#endif
#ifndef CEDR_DOWN_SP_WAVES1
# define CEDR_DOWN_SP_WAVES1 5  // ... and the 1-tracer-per-wave build
#endif

// The node problem of the shape-preserving root->leaves kernel: the form written for how a wave executes it
// (node_problem_mass_wave), in the zero-copy (ROWS) build too since that form got by with divergent branches
// (94 VGPRs, no spills; with the ballot form that build spilled 8 VGPRs in its loop). <false> = the branchy routine.

template <bool WAVE>
__device__ __forceinline__
void node_problem_sel (const double p_lo, const double p_orig, const double p_hi, const double Qm,
                       const KidRho r0, const double lo0, const double or0, const double hi0,
                       const KidRho r1, const double lo1, const double or1, const double hi1,
                       double& x0, double& x1, const bool prefer) {
  if (WAVE) node_problem_mass_wave(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
  else node_problem_mass(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
}

// CLS (round 5): the same kernel for the two other problem classes, whose tracers carry ONE field down the tree -- their
// mass -- instead of {lo, orig, hi}: 1 = t-only (consistent without shapepreserve: every node's bounds are the global
// q_min / q_max times its density, cedr_qlt.cpp:490-523,549-560; q bounds from the r2l rows behind the mass, as
// qlt_down_micro_kernel reads them), 2 = nonnegative (cedr_qlt_inl.hpp:190-198). A mixed tracer list is swept class by
// class, each launch over its range of the bulk order (tracers are grouped by type there, cedr_qlt.cpp:118-133), with the
// problem class a compile-time constant: the node problem of each is what solve_node evaluates for it in
// qlt_down_micro_kernel -- the same bits -- without the other classes' code and registers.
#ifndef CEDR_DOWN_CLS1_WAVES
# define CEDR_DOWN_CLS1_WAVES 5  // waves per SIMD the t-only form is compiled for
#endif
template <int TW, bool ROWS, int CLS = 0>
__global__ __launch_bounds__(kBlock, TW == 1 ? (CLS == 1 ? CEDR_DOWN_CLS1_WAVES : CEDR_DOWN_SP_WAVES1) : CEDR_DOWN_SP_WAVES2)
void qlt_down_sp_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st,
                         double* __restrict__ l2r, double* __restrict__ r2l,
                         const double* const* __restrict__ in_rows,
                         double* const* __restrict__ out_rows, const int nrow, const int pitch, const TailSplit tail) {
  constexpr int TT = kNW*TW;
  extern __shared__ double lds[];
  const WorkItem wi = work_item(st.ntiles, tail);
  const int tix = wi.tix;
  if (tix < 0) return;
  const int MN = st.max_nodes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  // (uniform stages could fetch the micro-subtree descriptor without waiting for the tile descriptor, as
  // tree_up_micro_kernel does)
  // (measured on this kernel: 6 more spilled VGPRs at its 96-register limit, 0.465 -> 0.527 ms -- not here)
  const bool uni = false;
  const uint32_t dpre = uni ? st.micro[64*static_cast<size_t>(tix) + lane] : 0;
  const cedr_hip_tile tile = st.tiles[tix];
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  // density records {rhom, 1/rhom, 1/(1/rhom)} of the tile's nodes, three doubles apart: one address per kid in the
  // node problems of the LDS levels (and 3 x node mod 16 is a permutation of the bank pairs)
  double* const rho3 = lds + (st.max_nint + 1);
  double* const cellrho = rho3 + 3*MN;        // [kCellRho][kWave]
  double* const recs = cellrho + kCellRho*kWave;
  // after the records: row tables (ROWS), then the tracers' l2r field / r2l row and the outputs' node ids
  const double** const inL = reinterpret_cast<const double**>(recs + static_cast<size_t>(TT)*pitch);
  double** const outL = reinterpret_cast<double**>(recs + static_cast<size_t>(TT)*pitch) + (ROWS ? L.l2rndps : 0);
  int32_t* const fldL = reinterpret_cast<int32_t*>(outL + (ROWS ? L.nt : 0));
  int32_t* const r2lL = fldL + L.nt;
  int32_t* const onodeL = r2lL + L.nt;        // [<= kWave]
  double* const qL = reinterpret_cast<double*>(onodeL + kWave);  // CLS 1: {q_min, q_max} of the chunk's tracers [TT][2]
  const int nleaf = tile.nleaf;
  const int64_t S = L.stride;
  const bool prefer = L.prefer_mass_con != 0;
  const int nt = L.nt;
  const int nchunks = (nt + TT - 1)/TT;
  const Micro m = uni ? decode_micro(dpre, true) : get_micro(st, tile, lane);
  const Roles ro = get_roles(m);
  const bool any2 = __ballot(m.shape == 2) != 0, any3 = __ballot(m.shape == 3) != 0;
  const int ushift = (ROWS && tile.in0 + ro.oUV + 1 > nrow - 1) ? 1 : 0;
  const int tt = threadIdx.x & (TT - 1);      // tracer of this thread in the LDS phases

  // The records {lo, orig, hi} of this lane's cells for this wave's tracers of chunk c, and the
  // assigned mass of tile output 'lane' (a micro tile has <= 64 outputs).
  // 'staged': the LDS copies of the tables are visible (not yet for the first chunk).
  auto load = [&] (const int c, double (&cur)[TW][3][3], double (&X)[TW], double (&Q)[TW][2], const bool staged) {
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      int bi = c*TT + wave + kNW*s;
      if (bi > nt - 1) bi = nt - 1;  // past the last tracer: re-read it (discarded)
      // (one-field classes: the mass is the second field of a t-only tracer's record, the first of a nonnegative one's)
      const int field = (staged ? fldL[bi] : L.bidx_l2r[bi]) - (CLS == 2 ? 1 : 0);
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        if (CLS != 0 && f != 1) { cur[s][0][f] = cur[s][1][f] = cur[s][2][f] = 0.0; continue; }
        const double* __restrict__ const row =
          (ROWS ? (staged ? inL[field + f] : in_rows[field + f]) : l2r + static_cast<size_t>(field + f)*S) +
          diag_in0(tile.in0);
        if (ROWS) load_roles_edge(row, ro, ushift, cur[s][0][f], cur[s][1][f], cur[s][2][f]);
        else load_roles(row, ro, cur[s][0][f], cur[s][1][f], cur[s][2][f]);
      }
      const int rb = staged ? r2lL[bi] : L.bidx_r2l[bi];
      X[s] = r2l[static_cast<size_t>(rb)*S + tile.out + (lane < tile.nout ? lane : tile.nout - 1)];
      if (CLS == 1) {
        Q[s][0] = r2l[static_cast<size_t>(rb + 1)*S + tile.out];
        Q[s][1] = r2l[static_cast<size_t>(rb + 2)*S + tile.out];
      } else {
        Q[s][0] = Q[s][1] = 0.0;
      }
    }
  };
  // The node problem of this kernel's class: rh = the parent's density (t-only: its bounds are q x density)
  auto node_problem = [&] (const double rh, const double ql, const double qh, const double p_lo, const double p_orig,
                           const double p_hi, const double Qm, const KidRho r0, const double lo0, const double or0,
                           const double hi0, const KidRho r1, const double lo1, const double or1, const double hi1,
                           double& x0, double& x1) {
    if (CLS == 0)
      node_problem_sel<true>(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
    else if (CLS == 1)
      solve_node(CONSISTENT, rh, ql, p_orig, qh, Qm, r0, ql, or0, qh, r1, ql, or1, qh, x0, x1, prefer);
    else
      solve_node(NONNEGATIVE, rh, 0.0, p_orig, 0.0, Qm, r0, 0.0, or0, 0.0, r1, 0.0, or1, 0.0, x0, x1, prefer);
  };

#ifdef CEDR_STAMPS
  unsigned long long acc[8] = {0,0,0,0,0,0,0,0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
  double cur[TW][3][3], Xout[TW], Qb[TW][2];
  load(wi.c0, cur, Xout, Qb, false);
  if (ROWS) {
    for (int i = threadIdx.x; i < L.l2rndps; i += kBlock) inL[i] = in_rows[i];
    for (int i = threadIdx.x; i < nt; i += kBlock) outL[i] = out_rows[i];
  }
  for (int i = threadIdx.x; i < nt; i += kBlock) { fldL[i] = L.bidx_l2r[i]; r2lL[i] = L.bidx_r2l[i]; }
  for (int i = threadIdx.x; i < tile.nout; i += kBlock) onodeL[i] = st.outs[tile.outs_off + i];

#ifdef CEDR_DIAG_NOSETUP  // (diagnostic builds, wrong results: the tile's densities are constants -- no loads, divisions or level sums)
  for (int i = threadIdx.x; i < 3*MN; i += kBlock) rho3[i] = 1.0;
  for (int i = threadIdx.x; i < kCellRho*kWave; i += kBlock) cellrho[i] = 1.0;
  stage_tile_meta(st, tile, kidsL, lvlL);
  __syncthreads();
#else
  // Densities: wave r derives role r of every micro-subtree (u, v, p, inner pair t):
  // two divisions per thread. Wave 3 also has the micro-subtree's root density.
  {
    double rh[3];
    if (ROWS) load_roles_edge(in_rows[0] + tile.in0, ro, ushift, rh[0], rh[1], rh[2]);
    else load_roles(l2r + tile.in0, ro, rh[0], rh[1], rh[2]);
    if (m.shape < 0) rh[0] = 1.0;
    if (m.shape < 1) rh[1] = 1.0;
    if (m.shape < 2) rh[2] = 1.0;
    const double t = rh[0] + rh[1];
    const double mine = wave == 0 ? rh[0] : (wave == 1 ? rh[1] : (wave == 2 ? rh[2] : t));
    const double w = 1.0/mine;
    cellrho[(3*wave    )*kWave + lane] = mine;
    cellrho[(3*wave + 1)*kWave + lane] = w;
    cellrho[(3*wave + 2)*kWave + lane] = 1.0/w;
    // (a,b)+c and a+(b,c) are the same sum: fp addition commutes
    if (wave == 3 && lane < nleaf) rho3[3*lane] = m.shape <= 0 ? rh[0] : (m.shape == 1 ? t : t + rh[2]);
  }
  stage_tile_meta(st, tile, kidsL, lvlL);
  __syncthreads();
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlL[lv], b = lvlL[lv+1];
    for (int j = a + threadIdx.x; j < b; j += kBlock) {
      const uint32_t kk = kidsL[j];
      rho3[3*(nleaf + j)] = rho3[3*(kk & 0xffff)] + rho3[3*(kk >> 16)];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nleaf + tile.nint; i += kBlock) {
    const double w = 1.0/rho3[3*i];
    rho3[3*i + 1] = w;
    rho3[3*i + 2] = 1.0/w;
  }
#endif
  double* const R = recs + tt*pitch;          // this thread's record row in the LDS phases
  STAMP(0);  // setup

  for (int c = wi.c0; c < nchunks; c += wi.cstep) {
    // Wave priority by phase (s_setprio; 0.481 -> 0.459 ms, profiles/r02_down_experiments.md 1): a SIMD holds
    // five waves of five workgroups in different phases. The register phase at the end of a chunk is
    // where all four waves of a workgroup are busy and nobody waits for anybody: lowest. A chunk's
    // loads and sums lead to the first barrier: above it. The LDS levels' node problems are done by
    // one or two waves while the others of the workgroup wait at a barrier: highest.
    __builtin_amdgcn_s_setprio(kPrioLoads);
    if (c != wi.c0) load(c, cur, Xout, Qb, true);

#ifdef CEDR_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(cur[0][0][0]), "+v"(cur[0][1][1]), "+v"(cur[0][2][2]), "+v"(Xout[0]) :: "memory");
#endif
    STAMP(1);  // loads arrived
    // ---- this wave's tracers: micro-subtree sums in registers, root records to LDS, the sums
    // of the LDS tree's nodes above them and the outputs' assigned masses -- all by this wave
    // alone, so no workgroup barrier separates the levels. (The previous chunk's last reads of
    // these rows were this wave's own, or lie before that chunk's last barrier.)
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      double* const Rw = recs + (wave + kNW*s)*pitch;
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        if (CLS != 0 && f != 1) continue;
        const double t = cur[s][0][f] + cur[s][1][f];
        const double r = m.shape <= 0 ? cur[s][0][f] : (m.shape == 1 ? t : t + cur[s][2][f]);
        if (lane < nleaf) Rw[kRS*lane + f] = r;
      }
    }
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      wave_lds_order();
      const int a = lvlL[lv], b = lvlL[lv+1];
#pragma unroll
      for (int s = 0; s < TW; ++s) {
        double* const Rw = recs + (wave + kNW*s)*pitch;
        for (int j = a + lane; j < b; j += kWave) {
          const uint32_t kk = kidsL[j];
          const double* K0 = Rw + kRS*(kk & 0xffff);
          const double* K1 = Rw + kRS*(kk >> 16);
          double* me = Rw + kRS*(nleaf + j);
          // (all six reads before the first write: the compiler must assume me aliases K0 / K1 and would
          // otherwise wait for LDS three times)
          if (CLS != 0) {
            me[1] = K0[1] + K1[1];
          } else {
            const double a0 = K0[0], a1 = K0[1], a2 = K0[2], b0 = K1[0], b1 = K1[1], b2 = K1[2];
            me[0] = a0 + b0; me[1] = a1 + b1; me[2] = a2 + b2;
          }
        }
      }
    }
    wave_lds_order();
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      const int biw = c*TT + wave + kNW*s;
      if (biw < nt && lane < tile.nout) recs[(wave + kNW*s)*pitch + kRS*onodeL[lane] + 3] = Xout[s];
      if (CLS == 1 && lane < 2) qL[2*(wave + kNW*s) + lane] = Qb[s][lane];
    }
    STAMP(2);  // sums + level sums
    __syncthreads();
    STAMP(3);  // first barrier
    // ---- node problems of the LDS levels, from the tile's outputs down, (node, tracer) items
    // spread over all threads
    __builtin_amdgcn_s_setprio(kPrioLdsLevels);
    for (int lv = tile.nlvl - 1; lv >= 0; --lv) {
      const int a = lvlL[lv];
      const int nitems = (lvlL[lv+1] - a) << Log2<TT>::value;
      for (int q = threadIdx.x; q < nitems; q += kBlock) {
        const int j = a + (q >> Log2<TT>::value);
        const uint32_t kk = kidsL[j];
        const int c0 = kk & 0xffff, c1 = kk >> 16, me = nleaf + j;
        const KidRho r0{rho3[3*c0], rho3[3*c0 + 1], rho3[3*c0 + 2]}, r1{rho3[3*c1], rho3[3*c1 + 1], rho3[3*c1 + 2]};
        const double* P = R + kRS*me;
        double* K0 = R + kRS*c0;
        double* K1 = R + kRS*c1;
        double x0, x1;
        node_problem(rho3[3*me], CLS == 1 ? qL[2*tt] : 0.0, CLS == 1 ? qL[2*tt + 1] : 0.0, P[0], P[1], P[2], P[3],
                     r0, K0[0], K0[1], K0[2], r1, K1[0], K1[1], K1[2], x0, x1);
        K0[3] = x0;
        K1[3] = x1;
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    STAMP(4);  // LDS levels' node problems + their barriers

    // ---- this wave's tracers again: the micro-subtree's node problems in registers
#pragma unroll
    for (int s = 0; s < TW; ++s) {
      const int biw = c*TT + wave + kNW*s;
      if (biw >= nt || lane >= nleaf) continue;
      const double Xr = recs[(wave + kNW*s)*pitch + kRS*lane + 3];
      double T[3];  // inner pair t: {lo, orig, hi} sums, re-formed
#pragma unroll
      for (int f = 0; f < 3; ++f) T[f] = cur[s][0][f] + cur[s][1][f];
      double Xt = Xr, Xu = Xr, Xv = 0, Xp = 0;
      const KidRho kt{cellrho[9*kWave + lane], cellrho[10*kWave + lane], cellrho[11*kWave + lane]};
      if ((any2 && m.shape == 2) || (any3 && m.shape == 3)) {
        // the micro root's sums, re-formed: three cells here, so t + p whatever the shape (no selects)
        const double RR[3] = {T[0] + cur[s][2][0], T[1] + cur[s][2][1], T[2] + cur[s][2][2]};
        const KidRho kp{cellrho[6*kWave + lane], cellrho[7*kWave + lane], cellrho[8*kWave + lane]};
        const double rr = CLS == 1 ? rho3[3*lane] : 0.0;  // the micro root's density
        if (any2 && m.shape == 2)
          node_problem(rr, Qb[s][0], Qb[s][1], RR[0], RR[1], RR[2], Xr, kt, T[0], T[1], T[2],
                       kp, cur[s][2][0], cur[s][2][1], cur[s][2][2], Xt, Xp);
        if (any3 && m.shape == 3)
          node_problem(rr, Qb[s][0], Qb[s][1], RR[0], RR[1], RR[2], Xr, kp, cur[s][2][0], cur[s][2][1], cur[s][2][2],
                       kt, T[0], T[1], T[2], Xp, Xt);
      }
      if (m.shape >= 1) {
        const KidRho ku{cellrho[0*kWave + lane], cellrho[1*kWave + lane], cellrho[2*kWave + lane]};
        const KidRho kv{cellrho[3*kWave + lane], cellrho[4*kWave + lane], cellrho[5*kWave + lane]};
        node_problem(kt.rh, Qb[s][0], Qb[s][1], T[0], T[1], T[2], Xt, ku, cur[s][0][0], cur[s][0][1], cur[s][0][2],
                     kv, cur[s][1][0], cur[s][1][1], cur[s][1][2], Xu, Xv);
      }
      double* __restrict__ const orow = (ROWS ? outL[biw] : r2l + static_cast<size_t>(r2lL[biw])*S) + tile.in0;
      store_roles(orow, ro, Xu, Xv, Xp);
      // The q bounds go down with the masses (cedr_qlt.cpp:549-560): the stage below reads them at its tiles' outputs.
      // Below the CELLS there is no stage: nothing reads a cell's bound rows again (get_Qm reads the mass row, the next
      // step's set_Qm rewrites the l2r bounds), so the launch over the cells leaves them out unless asked
      // (CEDR_QLT_LEAF_BOUNDS in the layout's flags; they are 4 of the 5 rows such a tracer's launch would write).
      if (CLS == 1 && ( ! (L.flags & CEDR_QLT_LEAF_STAGE) || (L.flags & CEDR_QLT_LEAF_BOUNDS))) {
        double* __restrict__ dst = r2l + static_cast<size_t>(r2lL[biw])*S + tile.in0 + m.start;
        double* __restrict__ l0p = l2r + static_cast<size_t>(fldL[biw])*S + tile.in0 + m.start;
        const double ql = Qb[s][0], qh = Qb[s][1];
        store_cells(dst + S, m.size, ql, ql, ql); store_cells(dst + 2*S, m.size, qh, qh, qh);
        store_cells(l0p, m.size, ql, ql, ql); store_cells(l0p + 2*S, m.size, qh, qh, qh);
      }
    }
    STAMP(5);  // register node problems + stores issued
  }
#ifdef CEDR_STAMPS
  if (lane == 0)
    for (int i = 0; i < 6; ++i) atomicAdd(&g_stamps[wave*8 + i], acc[i]);
#endif
}

// reference cedr_qlt.cpp:453-473 (root_compute), one thread per bulk index.
__global__ void qlt_root_kernel (const cedr_hip_qlt_layout L, const int root_slot,
                                 const double* __restrict__ l2r, double* __restrict__ r2l) {
  const int bi = blockIdx.x*blockDim.x + threadIdx.x;
  if (bi >= L.nt) return;
  const int pt = L.bidx_prob[bi], lb = L.bidx_l2r[bi], rb = L.bidx_r2l[bi];
  const int64_t S = L.stride;
  const int nf = (pt & NONNEGATIVE) ? ((pt & CONSERVE) ? 2 : 1) : ((pt & CONSERVE) ? 4 : 3);
  const int os = (pt & CONSERVE) ? nf - 1 : ((pt & NONNEGATIVE) ? 0 : 1);
  r2l[static_cast<size_t>(rb)*S + root_slot] = l2r[static_cast<size_t>(lb + os)*S + root_slot];
  if ((pt & CONSISTENT) && ! (pt & SHAPEPRESERVE)) {
    r2l[static_cast<size_t>(rb + 1)*S + root_slot] = l2r[static_cast<size_t>(lb)*S + root_slot];
    r2l[static_cast<size_t>(rb + 2)*S + root_slot] = l2r[static_cast<size_t>(lb + 2)*S + root_slot];
  }
}

// ------------------------------------------------------------ level mode
// The nodes a rank owns above its pure subtrees when the tree is partial or ownership is
// scattered (plan::LevelStep): one launch per tree level, one thread per (node, field) or
// (node, tracer), records in HBM -- the reference's own schedule (cedr_qlt.cpp:339-430, 525-604).
// nodes[3*j] = slot of node j, then its kids' slots. Consecutive threads take consecutive nodes.
__global__ void level_up_kernel (const int32_t* __restrict__ nodes, const int n, const int nfields,
                                 const uint8_t* __restrict__ op, const int64_t stride,
                                 double* __restrict__ data, const int zero_init) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x)*blockDim.x + threadIdx.x;
  if (idx >= static_cast<int64_t>(n)*nfields) return;
  const int j = static_cast<int>(idx % n), f = static_cast<int>(idx / n);
  double* const row = data + static_cast<size_t>(f)*stride;
  row[nodes[3*j]] = combine(op[f], zero_init, row[nodes[3*j + 1]], row[nodes[3*j + 2]]);
}

// r2l_solve_qp for one level (cedr_qlt.cpp:525-604): the node's assigned mass (and, for a tracer
// that is consistent only, the global q bounds that travel with it) is in r2l at the node's slot;
// the kids' masses go to r2l at theirs. The node's own sums are re-formed from the kids' records,
// as the tile kernels do.
__global__ void qlt_level_down_kernel (const cedr_hip_qlt_layout L, const int32_t* __restrict__ nodes,
                                       const int n, double* __restrict__ l2r, double* __restrict__ r2l) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x)*blockDim.x + threadIdx.x;
  if (idx >= static_cast<int64_t>(n)*L.nt) return;
  const int j = static_cast<int>(idx % n), bi = static_cast<int>(idx / n);
  const int me = nodes[3*j], c0 = nodes[3*j + 1], c1 = nodes[3*j + 2];
  const size_t S = static_cast<size_t>(L.stride);
  const int pt = L.bidx_prob[bi], lb = L.bidx_l2r[bi], rb = L.bidx_r2l[bi];
  const bool sp = (pt & SHAPEPRESERVE) != 0;
  const bool t_only = (pt & CONSISTENT) && ! sp;
  const bool prefer = L.prefer_mass_con != 0;
  const double rh0 = l2r[c0], rh1 = l2r[c1];
  const double w0 = 1.0/rh0, w1 = 1.0/rh1;
  const KidRho r0{rh0, w0, 1.0/w0}, r1{rh1, w1, 1.0/w1};
  const double rh = rh0 + rh1;
  const int fo = lb + (sp ? 1 : ((pt & CONSISTENT) ? 1 : 0));  // the 'orig' field
  const double o0 = l2r[fo*S + c0], o1 = l2r[fo*S + c1];
  const double X = r2l[rb*S + me];
  double x0, x1;
  if (t_only) {
    const double qlo = r2l[(rb + 1)*S + me], qhi = r2l[(rb + 2)*S + me];
    solve_node(pt, rh, qlo, o0 + o1, qhi, X, r0, qlo, o0, qhi, r1, qlo, o1, qhi, x0, x1, prefer);
    r2l[(rb + 1)*S + c0] = qlo; r2l[(rb + 2)*S + c0] = qhi;
    r2l[(rb + 1)*S + c1] = qlo; r2l[(rb + 2)*S + c1] = qhi;
    l2r[lb*S + c0] = qlo; l2r[(lb + 2)*S + c0] = qhi;
    l2r[lb*S + c1] = qlo; l2r[(lb + 2)*S + c1] = qhi;
  } else {
    double lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
    if (sp) {
      lo0 = l2r[lb*S + c0]; hi0 = l2r[(lb + 2)*S + c0];
      lo1 = l2r[lb*S + c1]; hi1 = l2r[(lb + 2)*S + c1];
    }
    solve_node(pt, rh, lo0 + lo1, o0 + o1, hi0 + hi1, X, r0, lo0, o0, hi0, r1, lo1, o1, hi1, x0, x1, prefer);
  }
  r2l[rb*S + c0] = x0;
  r2l[rb*S + c1] = x1;
}

// ------------------------------------------------------------------- apex
// Multi-rank runs: the part of the tree above the ranks' pure subtrees is tiny (P boundary
// roots for a contiguous P-way sharding) and replicated on every rank. After the all-gather of
// the boundary roots' records ONE workgroup does everything that part needs, in one launch:
// unpack the gathered records, sweep them up to the global root, root_compute
// (cedr_qlt.cpp:441-476), sweep the node problems down (:525-604), and write the assigned masses
// of THIS rank's boundary roots to their local r2l slots. It replaces five launches (scatter, top
// leaves->root sweep, root, top root->leaves sweep, copy) -- latency, not bandwidth.
// Record slots per (node, tracer): 0 lo / q_min, 1 orig (a nonnegative tracer's mass), 2 hi / q_max,
// 3 Qm_prev sum on the way up (conserving tracers), the assigned mass X on the way down.
// TT tracers per pass (runtime: all of them when they fit in LDS).
__global__ __launch_bounds__(kBlock)
void qlt_apex_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st,
                      const double* __restrict__ recv, const int32_t* __restrict__ src,
                      const int nmine, const int32_t* __restrict__ my_in,
                      const int32_t* __restrict__ my_slot, double* __restrict__ r2l,
                      const int TT, const int pitch) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[0];
  const int nleaf = tile.nleaf, MN = st.max_nodes, nf = L.l2rndps, nt = L.nt;
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  double* const rho = lds + (st.max_nint + 1);
  double* const rw = rho + MN;
  double* const rq = rw + MN;
  double* const recs = rq + MN;
  const int64_t S = L.stride;
  const bool prefer = L.prefer_mass_con != 0;
  const int root = st.outs[tile.outs_off];
  for (int i = threadIdx.x; i < nleaf; i += kBlock) rho[i] = recv[static_cast<size_t>(src[i])*nf];
  stage_tile_meta(st, tile, kidsL, lvlL);
  __syncthreads();
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlL[lv], b = lvlL[lv+1];
    for (int j = a + threadIdx.x; j < b; j += kBlock) {
      const uint32_t kk = kidsL[j];
      rho[nleaf + j] = rho[kk & 0xffff] + rho[kk >> 16];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nleaf + tile.nint; i += kBlock) {
    const double w = 1.0/rho[i];
    rw[i] = w;
    rq[i] = 1.0/w;
  }
  for (int c0 = 0; c0 < nt; c0 += TT) {
    const int ntc = min(TT, nt - c0);
    // gathered records -> LDS
    for (int q = threadIdx.x; q < nleaf*ntc*4; q += kBlock) {
      const int k = q & 3, i = (q >> 2) % nleaf, tt = (q >> 2)/nleaf;
      const int pt = L.bidx_prob[c0 + tt], lb = L.bidx_l2r[c0 + tt];
      int field = -1;
      if (pt & NONNEGATIVE) { if (k == 1) field = lb; else if (k == 3 && (pt & CONSERVE)) field = lb + 1; }
      else if (k < 3 || (pt & CONSERVE)) field = lb + k;
      recs[tt*pitch + 4*i + k] = field >= 0 ? recv[static_cast<size_t>(src[i])*nf + field] : 0.0;
    }
    __syncthreads();
    // leaves -> root: sums, except q_min / q_max of a consistent-only tracer (cedr_qlt.cpp:375-383)
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      const int a = lvlL[lv], nn = lvlL[lv+1] - a;
      for (int q = threadIdx.x; q < nn*ntc*4; q += kBlock) {
        const int k = q & 3, j = a + (q >> 2) % nn, tt = (q >> 2)/nn;
        const int pt = L.bidx_prob[c0 + tt];
        const bool t_only = (pt & CONSISTENT) && ! (pt & SHAPEPRESERVE);
        const uint32_t kk = kidsL[j];
        double* const R = recs + tt*pitch;
        const double k0 = R[4*(kk & 0xffff) + k], k1 = R[4*(kk >> 16) + k];
        R[4*(nleaf + j) + k] = (t_only && k == 0) ? rmin(k0, k1) : ((t_only && k == 2) ? rmax(k0, k1) : k0 + k1);
      }
      __syncthreads();
    }
    // root_compute: the root's assigned mass is its Qm_prev sum (conserving) or its Qm sum
    for (int tt = threadIdx.x; tt < ntc; tt += kBlock) {
      double* const R = recs + tt*pitch;
      if ( ! (L.bidx_prob[c0 + tt] & CONSERVE)) R[4*root + 3] = R[4*root + 1];
    }
    __syncthreads();
    // root -> leaves
    for (int lv = tile.nlvl - 1; lv >= 0; --lv) {
      const int a = lvlL[lv], nn = lvlL[lv+1] - a;
      for (int q = threadIdx.x; q < nn*ntc; q += kBlock) {
        const int j = a + q % nn, tt = q/nn;
        const int pt = L.bidx_prob[c0 + tt];
        const bool t_only = (pt & CONSISTENT) && ! (pt & SHAPEPRESERVE);
        double* const R = recs + tt*pitch;
        const uint32_t kk = kidsL[j];
        const int c0n = kk & 0xffff, c1n = kk >> 16, me = nleaf + j;
        const KidRho r0{rho[c0n], rw[c0n], rq[c0n]}, r1{rho[c1n], rw[c1n], rq[c1n]};
        const double* P = R + 4*me;
        double* K0 = R + 4*c0n;
        double* K1 = R + 4*c1n;
        double x0, x1;
        if (t_only) {
          const double qlo = R[4*root], qhi = R[4*root + 2];
          solve_node(pt, rho[me], qlo, P[1], qhi, P[3], r0, qlo, K0[1], qhi, r1, qlo, K1[1], qhi, x0, x1, prefer);
        } else {
          solve_node(pt, rho[me], P[0], P[1], P[2], P[3], r0, K0[0], K0[1], K0[2], r1, K1[0], K1[1], K1[2],
                     x0, x1, prefer);
        }
        K0[3] = x0;
        K1[3] = x1;
      }
      __syncthreads();
    }
    // this rank's boundary roots: assigned mass (and a consistent-only tracer's global q bounds,
    // which travel root->leaves unchanged) to their local slots
    for (int q = threadIdx.x; q < nmine*ntc; q += kBlock) {
      const int j = q % nmine, tt = q/nmine;
      const int pt = L.bidx_prob[c0 + tt], rb = L.bidx_r2l[c0 + tt];
      const double* const R = recs + tt*pitch;
      r2l[static_cast<size_t>(rb)*S + my_slot[j]] = R[4*my_in[j] + 3];
      if ((pt & CONSISTENT) && ! (pt & SHAPEPRESERVE)) {
        r2l[static_cast<size_t>(rb + 1)*S + my_slot[j]] = R[4*root];
        r2l[static_cast<size_t>(rb + 2)*S + my_slot[j]] = R[4*root + 2];
      }
    }
    __syncthreads();
  }
}

// The same for a plain sum of 'nfields' fields (BfbTreeAllReducer): one thread per field walks the
// whole (tiny) tree in its own LDS column -- no barriers -- and writes the root's value.
__global__ __launch_bounds__(kBlock)
void tree_apex_sum_kernel (const cedr_hip_stage st, const double* __restrict__ recv,
                           const int32_t* __restrict__ src, const int nfields, const int zero_init,
                           double* __restrict__ out, const int pitch, const int ncols) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[0];
  const int nleaf = tile.nleaf;
  const uint16_t* __restrict__ kids = st.kids + 2*static_cast<size_t>(tile.kids_off);
  const int root = st.outs[tile.outs_off];
  if (static_cast<int>(threadIdx.x) >= ncols) return;  // LDS holds ncols columns
  for (int f = threadIdx.x; f < nfields; f += ncols) {
    double* const v = lds + static_cast<size_t>(threadIdx.x)*pitch;
    for (int i = 0; i < nleaf; ++i) v[i] = recv[static_cast<size_t>(src[i])*nfields + f];
    for (int j = 0; j < tile.nint; ++j) {  // kids come before parents in the tile's node order
      const double k0 = v[kids[2*j]], k1 = v[kids[2*j + 1]];
      double r;
      if (zero_init) { r = 0; r += k0; r += k1; } else r = k0 + k1;
      v[nleaf + j] = r;
    }
    out[f] = v[root];
  }
}

// -------------------------------------------------------------- fused top
// The last, latency-bound stages of a rank's forest as ONE plain tile (plan::Stage::top: up to 2048
// inputs = the outputs of the last micro stage, all tree levels above them, one output per subtree
// root of the rank). Tracers (fields) are independent, so the tile is swept tracer-parallel: one
// workgroup of 512 threads per tracer -- or per group of FPB fields on the way up -- walks ALL levels
// in one launch with workgroup barriers between them; no grid barrier, no launch per stage. The
// inputs stay in HBM / L2 (only the lowest internal level reads them, coalesced); the internal
// nodes live in LDS. A kid with local id < nleaf is an input, else internal node id - nleaf.
// Replaces, per direction, the launches of stages 3-4 (cedr_qlt.cpp:339-430, 525-604 for those
// levels) and -- fused in -- gather_slots (:432-439), root_compute (:441-476) or the apex launch.
constexpr int kTopBlock = 1024;

#ifdef CEDR_TOP_STAMPS
// Diagnostic build only: wall-clock stamps (10 ns units) of workgroup 0's thread 0 at the phase
// boundaries of qlt_top_kernel (slots 0..31) and tree_top_up_kernel (32..63).
__device__ unsigned long long g_top_stamps[64];
// (slots 60 / 61: the shader clock at stamps 0 / 30 of qlt_top_kernel: clock rate = cycles / wall time)
# define TSTAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { g_top_stamps[i] = wall_clock64(); \
    if ((i) == 0) g_top_stamps[60] = clock64(); if ((i) == 30) g_top_stamps[61] = clock64(); } } while (0)
#else
# define TSTAMP(i)
#endif

// First level of the narrow suffix of a top tile's levels (every level from there up has <= 64 nodes),
// at least 1; the host passes it + 1 in bits 8..15 of the stage's flags (0 = not given: no narrow levels).
__device__ __forceinline__ int top_narrow_from (const cedr_hip_stage& st, const cedr_hip_tile& tile) {
  const int v = (st.flags >> 8) & 0xff;
  return v ? min(max(v - 1, 1), tile.nlvl) : tile.nlvl;
}

struct TopMeta {
  uint32_t* kidsL;
  int32_t* lvlL;
  double* rest;  // first double after the metadata
};
// The tile descriptor comes by value (the host knows it), so the first thing the kernel waits for is
// already the kid table itself.
__device__ __forceinline__ TopMeta top_stage_meta (const cedr_hip_stage& st, const cedr_hip_tile& tile,
                                                   double* lds) {
  TopMeta m;
  m.kidsL = reinterpret_cast<uint32_t*>(lds);
  m.lvlL = reinterpret_cast<int32_t*>(m.kidsL + st.max_nint);
  m.rest = lds + (2*static_cast<size_t>(st.max_nint) + 2 + 1)/2;
  const uint32_t* __restrict__ kp = reinterpret_cast<const uint32_t*>(st.kids + 2*static_cast<size_t>(tile.kids_off));
  for (int j = threadIdx.x; j < tile.nint; j += kTopBlock) m.kidsL[j] = kp[j];
  const int32_t* __restrict__ lp = st.lvlptr + tile.lvl_off;
  for (int j = threadIdx.x; j <= tile.nlvl; j += kTopBlock) m.lvlL[j] = lp[j];
  return m;
}

// Leaves->root, any combine rules. grid.x = ceil(nfields/FPB); lane -> (node, field) with the field
// fastest. Outputs go to their slots (data[f*stride + tile.out + k]); with sendbuf != NULL the
// records of this rank's nroots boundary roots are packed as [root][field] for the exchange as well
// (a root that is not an output of this tile was finished by an earlier stage: read from its slot).
// The lowest level (both kids of every node are inputs) takes its <= 1024 x FPB items in one pass of
// eight per thread, so that all of the kernel's HBM / L2 reads are in flight together.
template <int FPB>
__global__ __launch_bounds__(kTopBlock)
void tree_top_up_kernel (const cedr_hip_stage st, const cedr_hip_tile tile, const int nfields,
                         const uint8_t* __restrict__ op, const int64_t stride, double* __restrict__ data,
                         const int zero_init, const int nroots, const int32_t* __restrict__ root_slots,
                         double* __restrict__ sendbuf) {
  extern __shared__ double lds[];
  const int nleaf = tile.nleaf;
  const int pitch = st.max_nint | 1;
  constexpr int LF = Log2<FPB>::value;
  const int ff = threadIdx.x & (FPB - 1);
  const int f = blockIdx.x*FPB + ff;
  const bool live = f < nfields;
  const double* __restrict__ const row = data + static_cast<size_t>(live ? f : 0)*stride + tile.in0;
  // level 0 straight from the kid table in global memory: its input loads can go out before the
  // table has been staged in LDS
  const uint32_t* __restrict__ const kp = reinterpret_cast<const uint32_t*>(st.kids + 2*static_cast<size_t>(tile.kids_off));
  // (n0 is only needed once the loads are under way: nodes past level 0 among the first 1024 just
  // load an input for nothing). With CEDR_STAGE_LVL0_PAIRS node j's kids are known: inputs 2j, 2j + 1.
  const bool pairs = (st.flags & CEDR_STAGE_LVL0_PAIRS) != 0;
  const int n0 = pairs ? nleaf/2 : (tile.nlvl > 0 ? st.lvlptr[tile.lvl_off + 1] : 0);
  constexpr int NB0 = 1024*FPB/kTopBlock;  // items of level 0 per thread
  double k0[NB0], k1[NB0];
  {
    uint32_t kk[NB0];
#pragma unroll
    for (int u = 0; u < NB0; ++u) {
      const int j = (threadIdx.x + u*kTopBlock) >> LF;
      kk[u] = pairs ? (2*j + 1 < nleaf ? (2*j) | ((2*j + 1) << 16) : 0) : (j < tile.nint ? kp[j] : 0);
    }
#pragma unroll
    for (int u = 0; u < NB0; ++u) {
      const int c0 = kk[u] & 0xffff, c1 = kk[u] >> 16;
      k0[u] = row[c0 < nleaf ? c0 : 0];
      k1[u] = row[c1 < nleaf ? c1 : 0];
    }
  }
  const TopMeta M = top_stage_meta(st, tile, lds);
  double* const v = M.rest;  // [FPB][pitch]
  const int o = live ? op[f] : 0;
  double* const vf = v + ff*pitch;
  auto val = [&] (const int c) -> double { return c < nleaf ? row[c] : vf[c - nleaf]; };
  const int narrow = top_narrow_from(st, tile);
  TSTAMP(32);
#pragma unroll
  for (int u = 0; u < NB0; ++u) {
    const int q = threadIdx.x + u*kTopBlock;
    if ((q >> LF) < n0 && live) vf[q >> LF] = combine(o, zero_init, k0[u], k1[u]);
  }
  __syncthreads();
  TSTAMP(33);
  for (int lv = 1; lv < narrow; ++lv) {
    const int a = M.lvlL[lv];
    const int nitems = (M.lvlL[lv+1] - a) << LF;
    for (int q0 = threadIdx.x; q0 < nitems; q0 += 2*kTopBlock) {
      double c0[2], c1[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u*kTopBlock;
        c0[u] = c1[u] = 0;
        if (q < nitems && live) {
          const uint32_t kk = M.kidsL[a + (q >> LF)];
          c0[u] = val(kk & 0xffff);
          c1[u] = val(kk >> 16);
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = q0 + u*kTopBlock;
        if (q < nitems && live) vf[a + (q >> LF)] = combine(o, zero_init, c0[u], c1[u]);
      }
    }
    __syncthreads();
  }
  // Narrow levels (<= 64 nodes each): wave w takes field w of the workgroup's FPB, lane = node, level
  // after level in program order -- no workgroup barrier.
  {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
    const int fw = blockIdx.x*FPB + wave;
    if (wave < FPB && fw < nfields) {
      const int ow = op[fw];
      const double* __restrict__ const roww = data + static_cast<size_t>(fw)*stride + tile.in0;
      double* const vw = v + wave*pitch;
      for (int lv = narrow; lv < tile.nlvl; ++lv) {
        const int a = M.lvlL[lv], n = M.lvlL[lv+1] - a;
        if (lane < n) {
          const uint32_t kk = M.kidsL[a + lane];
          const int c0 = kk & 0xffff, c1 = kk >> 16;
          vw[a + lane] = combine(ow, zero_init, c0 < nleaf ? roww[c0] : vw[c0 - nleaf],
                                 c1 < nleaf ? roww[c1] : vw[c1 - nleaf]);
        }
        wave_lds_order();
      }
    }
  }
  __syncthreads();
  TSTAMP(34);
  if ( ! live) return;
  for (int k = threadIdx.x >> LF; k < tile.nout; k += kTopBlock/FPB)
    data[static_cast<size_t>(f)*stride + tile.out + k] = val(st.outs[tile.outs_off + k]);
  if (sendbuf)
    for (int j = threadIdx.x >> LF; j < nroots; j += kTopBlock/FPB) {
      const int s = root_slots[j], k = s - tile.out;
      sendbuf[static_cast<size_t>(j)*nfields + f] =
        (k >= 0 && k < tile.nout) ? val(st.outs[tile.outs_off + k]) : data[static_cast<size_t>(f)*stride + s];
    }
}

// QLT over the top tile, one workgroup per tracer (bulk index blockIdx.x): re-forms the sums of the
// tile's internal nodes from its inputs, obtains the assigned masses of the tile's outputs, and sweeps
// the node problems (cedr_qlt.cpp:525-604, cedr_qlt_inl.hpp:120-203) down to the inputs' r2l slots.
// mode 0: the outputs' masses (and a consistent-only tracer's global q bounds) are in their r2l slots
//         (staged replicated top, level mode).
// mode 1: single rank: the tile's one output is the global root; root_compute (cedr_qlt.cpp:441-476)
//         happens here, on sums that then include Qm_prev -- no leaves->root launch over the top tile
//         is needed at all.
// mode 2: multi-rank with an apex tile: what qlt_apex_kernel does for this tracer (gathered boundary
//         records -> up, root_compute, down) gives this rank's boundary roots their masses first.
// LDS record per internal node and tracer: {lo / q_min, orig, hi / q_max, Qm_prev sum going up ->
// assigned mass X going down}, plus rhom, 1/rhom, 1/(1/rhom) per node.
// 1024 threads: a level has at most nleaf/2 <= 1024 nodes, one per thread. The lowest level -- both kids
// of each node are inputs -- keeps its kids' records in REGISTERS from the sums to the node problems: the
// tile's inputs are read from HBM / L2 once, all loads in flight together at the start, and nothing at
// the end of the dependent chain of levels waits for memory. The root->leaves part is ONE loop over the
// apex tile's levels (mode 2) and then the top tile's, with a single inlined copy of the node problem:
// the kernel runs once per step on a cold instruction cache, so its code size is launch latency.
#ifndef CEDR_TOP_REC_STRIDE
# define CEDR_TOP_REC_STRIDE 5
#endif
constexpr int kTRS = CEDR_TOP_REC_STRIDE;  // doubles per LDS record of qlt_top_kernel (4 + padding, as kRS)
struct TopApex {
  cedr_hip_stage st;
  cedr_hip_tile tile;
  const double* recv;
  const int32_t* src;
  int32_t nmine;
  const int32_t* my_in;
  const int32_t* my_slot;
};

__global__ __launch_bounds__(kTopBlock)
void qlt_top_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st, const cedr_hip_tile tile,
                     double* __restrict__ l2r, double* __restrict__ r2l, const int mode, const TopApex A) {
  extern __shared__ double lds[];
  const int nleaf = tile.nleaf, nint = tile.nint, MI = st.max_nint;
  const size_t S = static_cast<size_t>(L.stride);
  const bool prefer = L.prefer_mass_con != 0;
  const int bi = blockIdx.x;
  const int pt = L.bidx_prob[bi], lb = L.bidx_l2r[bi], rb = L.bidx_r2l[bi];
  const bool sp = (pt & SHAPEPRESERVE) != 0;
  const bool t_only = (pt & CONSISTENT) && ! sp;
  const bool cons = (pt & CONSERVE) != 0;
  const int in0 = tile.in0;
  const int tid = threadIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  // l2r field of record entry k (-1: the tracer has none), as in qlt_apex_kernel
  auto field_of = [&] (const int k) -> int {
    if (pt & NONNEGATIVE) return k == 1 ? lb : ((k == 3 && cons) ? lb + 1 : -1);
    return (k < 3 || cons) ? lb + k : -1;
  };
  auto rule_of = [&] (const int k) -> int { return (t_only && k == 0) ? 1 : ((t_only && k == 2) ? 2 : 0); };
  const double* __restrict__ const rrow = l2r + in0;
  // NARROW levels: the levels from 'narrow' up have at most 64 nodes each (host-computed, >= 1). One wave
  // walks them alone -- per quantity on the way up (waves 0-3: record entry = wave; wave 4: densities),
  // wave 0 on the way down -- in program order, with no workgroup barrier between them.
  const int narrow = top_narrow_from(st, tile);
  TSTAMP(0);

  // ---- level 0 in registers: kid indices straight from global memory (or known: inputs 2j, 2j + 1),
  // then every load at once. (n0 is only needed once the loads are under way: a node past level 0 just
  // loads an input for nothing)
  const uint32_t* __restrict__ const kp = reinterpret_cast<const uint32_t*>(st.kids + 2*static_cast<size_t>(tile.kids_off));
  const bool pairs = (st.flags & CEDR_STAGE_LVL0_PAIRS) != 0;
  const int n0 = pairs ? nleaf/2 : (tile.nlvl > 0 ? st.lvlptr[tile.lvl_off + 1] : 0);
  int c0r[2];               // this thread's level-0 node: its kids' input indices
  double f0[2][4], r0[2];   // their records and densities
  {
    if (pairs) { c0r[0] = 2*tid < nleaf ? 2*tid : 0; c0r[1] = 2*tid + 1 < nleaf ? 2*tid + 1 : 0; }
    else {
      const uint32_t kk = tid < nint ? kp[tid] : 0;
      const int a = kk & 0xffff, b = kk >> 16;
      c0r[0] = a < nleaf ? a : 0; c0r[1] = b < nleaf ? b : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int fk = field_of(k);
      const bool want = fk >= 0 && (k < 3 || mode == 1);
      const double* __restrict__ const frow = l2r + static_cast<size_t>(fk < 0 ? 0 : fk)*S + in0;
#pragma unroll
      for (int e = 0; e < 2; ++e) f0[e][k] = want ? frow[c0r[e]] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) r0[e] = rrow[c0r[e]];
  }
  // Mode 2: everything the apex part will need from global memory is requested now as well -- the
  // gathered records (through src; wave w < 4 takes entry w of every apex input, wave 4 its density), this
  // thread's boundary root and where it goes -- so that it arrives while the tile's sums are formed.
  const int an = A.tile.nleaf, nf = L.l2rndps;
  double av = 0;
  int my_in_r = 0, my_slot_r = -1, my_node_r = -1;
  const int root_node = mode == 1 ? st.outs[tile.outs_off] : 0;
  if (mode == 2) {
    if (wave < 5 && lane < an) {
      const int fk = wave < 4 ? field_of(wave) : 0;
      av = fk >= 0 ? A.recv[static_cast<size_t>(A.src[lane])*nf + fk] : 0.0;
    }
    if (tid < A.nmine) {
      my_in_r = A.my_in[tid];
      my_slot_r = A.my_slot[tid];
      const int k = my_slot_r - tile.out;
      my_node_r = (k >= 0 && k < tile.nout) ? st.outs[tile.outs_off + k] : -1;
    }
  }
  const TopMeta M = top_stage_meta(st, tile, lds);
  double* const rho = M.rest;
  double* const rw = rho + MI;
  double* const rq = rw + MI;
  double* const recs = rq + MI;          // [nint][4]
  double* const apx = recs + kTRS*static_cast<size_t>(MI);
  // apex part of the LDS (mode 2): every node of the apex tile, inputs first
  const int AM = A.st.max_nodes;
  uint32_t* const akids = reinterpret_cast<uint32_t*>(apx);
  int32_t* const alvl = reinterpret_cast<int32_t*>(akids + A.st.max_nint);
  double* const arho = apx + (2*static_cast<size_t>(A.st.max_nint) + 2 + 1)/2;
  double* const arw = arho + AM;
  double* const arq = arw + AM;
  double* const arec = arq + AM;         // [an + ai][4]
  int aroot = 0;
  if (mode == 2) {
    const uint32_t* __restrict__ akp = reinterpret_cast<const uint32_t*>(A.st.kids + 2*static_cast<size_t>(A.tile.kids_off));
    for (int j = tid; j < A.tile.nint; j += kTopBlock) akids[j] = akp[j];
    const int32_t* __restrict__ lp = A.st.lvlptr + A.tile.lvl_off;
    for (int j = tid; j <= A.tile.nlvl; j += kTopBlock) alvl[j] = lp[j];
    aroot = A.st.outs[A.tile.outs_off];
  }
  if (tid < n0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) recs[kTRS*tid + k] = combine(rule_of(k), 0, f0[0][k], f0[1][k]);
    rho[tid] = r0[0] + r0[1];
  }
  __syncthreads();
  TSTAMP(1);

  // ---- sums of the other internal nodes: entries 0..2 (and 3 = Qm_prev when the root is computed
  // here) of every node's record, and its density. A kid may still be an input.
  // Wide levels: (node, entry) items over all threads, a barrier per level.
  {
    const int k = tid & 3;
    const int fk = field_of(k);
    const bool want = fk >= 0 && (k < 3 || mode == 1);
    const double* __restrict__ const frow = l2r + static_cast<size_t>(fk < 0 ? 0 : fk)*S + in0;
    const int ok = rule_of(k);
    for (int lv = 1; lv < narrow; ++lv) {
      const int a = M.lvlL[lv], n = M.lvlL[lv+1] - a;
      for (int q = tid; q < 4*n; q += kTopBlock) {
        double r = 0.0;
        if (want) {
          const uint32_t kk = M.kidsL[a + (q >> 2)];
          const int c0 = kk & 0xffff, c1 = kk >> 16;
          r = combine(ok, 0, c0 < nleaf ? frow[c0] : recs[kTRS*(c0 - nleaf) + k],
                      c1 < nleaf ? frow[c1] : recs[kTRS*(c1 - nleaf) + k]);
        }
        recs[kTRS*(a + (q >> 2)) + k] = r;
      }
      if (tid < n) {
        const uint32_t kk = M.kidsL[a + tid];
        const int c0 = kk & 0xffff, c1 = kk >> 16;
        rho[a + tid] = (c0 < nleaf ? rrow[c0] : rho[c0 - nleaf]) + (c1 < nleaf ? rrow[c1] : rho[c1 - nleaf]);
      }
      __syncthreads();
    }
  }
  // Narrow levels (and, mode 2, the apex tile leaves->root, all of whose levels are narrow): one wave
  // per quantity, lane = node.
  if (wave < 5) {
    const int k = wave;  // 0..3: record entry; 4: density
    const int fk = k < 4 ? field_of(k) : 0;
    const bool want = k == 4 || (fk >= 0 && (k < 3 || mode == 1));
    const bool awant = k == 4 || fk >= 0;  // the apex sums include Qm_prev: its root is computed here
    const double* __restrict__ const frow = l2r + static_cast<size_t>(fk < 0 ? 0 : fk)*S + in0;
    const int ok = k < 4 ? rule_of(k) : 0;
    // (a level's offsets and kid indices do not depend on the sums: they are read one level ahead, so that
    // only the kids' values and the write are on the chain from level to level)
    int a = 0, n = 0;
    uint32_t kk = 0;
    if (narrow < tile.nlvl) {
      a = M.lvlL[narrow]; n = M.lvlL[narrow+1] - a;
      kk = lane < n ? M.kidsL[a + lane] : 0;
    }
    for (int lv = narrow; lv < tile.nlvl; ++lv) {
      int a2 = 0, n2 = 0;
      uint32_t kk2 = 0;
      if (lv + 1 < tile.nlvl) {
        a2 = M.lvlL[lv+1]; n2 = M.lvlL[lv+2] - a2;
        kk2 = lane < n2 ? M.kidsL[a2 + lane] : 0;
      }
      if (lane < n) {
        const int j = a + lane;
        double r = 0.0;
        if (want) {
          const int c0 = kk & 0xffff, c1 = kk >> 16;
          if (k < 4) r = combine(ok, 0, c0 < nleaf ? frow[c0] : recs[kTRS*(c0 - nleaf) + k],
                                 c1 < nleaf ? frow[c1] : recs[kTRS*(c1 - nleaf) + k]);
          else r = (c0 < nleaf ? rrow[c0] : rho[c0 - nleaf]) + (c1 < nleaf ? rrow[c1] : rho[c1 - nleaf]);
        }
        if (k < 4) recs[kTRS*j + k] = r;
        else rho[j] = r;
      }
      wave_lds_order();
      a = a2; n = n2; kk = kk2;
    }
    if (mode == 2) {
      if (lane < an) {
        if (k < 4) arec[kTRS*lane + k] = av;
        else arho[lane] = av;
      }
      wave_lds_order();
      for (int lv = 0; lv < A.tile.nlvl; ++lv) {
        const int a = alvl[lv], n = alvl[lv+1] - a;
        if (lane < n) {
          const uint32_t kk = akids[a + lane];
          const int c0 = kk & 0xffff, c1 = kk >> 16, me = an + a + lane;
          if (k < 4) arec[kTRS*me + k] = awant ? combine(ok, 0, arec[kTRS*c0 + k], arec[kTRS*c1 + k]) : 0.0;
          else arho[me] = arho[c0] + arho[c1];
        }
        wave_lds_order();
      }
    }
  }
  // reciprocals of the densities (kept off the narrow levels' chain: two dependent divisions per level)
  __syncthreads();
  for (int j = tid; j < nint; j += kTopBlock) {
    const double w = 1.0/rho[j];
    rw[j] = w;
    rq[j] = 1.0/w;
  }
  if (mode == 2)
    for (int i = tid; i < an + A.tile.nint; i += kTopBlock) {
      const double w = 1.0/arho[i];
      arw[i] = w;
      arq[i] = 1.0/w;
    }
  __syncthreads();
  TSTAMP(2);

  // ---- assigned masses of the tile's outputs; the global q bounds of a consistent-only tracer
  double qlo = 0, qhi = 0;
  // an output that is an input of the tile (a subtree root the previous stage finished): pass on
  auto put_input = [&] (const int c, const double x, const double ql, const double qh) {
    r2l[rb*S + in0 + c] = x;
    if (t_only) {  // the q bounds travel with the mass (cedr_qlt.cpp:549-560)
      r2l[(rb + 1)*S + in0 + c] = ql; r2l[(rb + 2)*S + in0 + c] = qh;
      l2r[lb*S + in0 + c] = ql; l2r[(lb + 2)*S + in0 + c] = qh;
    }
  };
  if (mode == 0) {
    if (t_only) { qlo = r2l[(rb + 1)*S + tile.out]; qhi = r2l[(rb + 2)*S + tile.out]; }
    for (int k = tid; k < tile.nout; k += kTopBlock) {
      const int node = st.outs[tile.outs_off + k];
      const double x = r2l[rb*S + tile.out + k];
      if (node >= nleaf) recs[kTRS*(node - nleaf) + 3] = x;
      else put_input(node, x, qlo, qhi);
    }
  } else if (mode == 1) {
    // root_compute: the root's mass is its Qm_prev sum (conserving tracers) or its Qm sum; the
    // record's entry 3 already holds the former
    const int node = root_node;
    if (node >= nleaf) {
      const double* const R = recs + kTRS*(node - nleaf);
      if (t_only) { qlo = R[0]; qhi = R[2]; }
      if (tid == 0 && ! cons) recs[kTRS*(node - nleaf) + 3] = R[1];
    } else {  // a one-cell tree: the root is the input itself
      const int fo = field_of(cons ? 3 : 1);
      if (t_only) { qlo = l2r[lb*S + in0 + node]; qhi = l2r[(lb + 2)*S + in0 + node]; }
      if (tid == 0) put_input(node, l2r[fo*S + in0 + node], qlo, qhi);
    }
  } else {
    if (t_only) { qlo = arec[kTRS*aroot]; qhi = arec[kTRS*aroot + 2]; }
    if (tid == 0 && ! cons) arec[kTRS*aroot + 3] = arec[kTRS*aroot + 1];
  }
  __syncthreads();

  // ---- node problems: the apex tile's levels from its root down (mode 2), the hand-over of this
  // rank's boundary roots, then the top tile's levels from its outputs down. One node per thread and
  // level. What a node problem needs of an INPUT kid of the top tile comes from global memory (level 0:
  // from the registers) and does not depend on the level above: it is requested before the results of
  // that level are waited for. Everything else it needs but its own assigned mass -- the kids' sums and
  // densities in LDS -- is read before that point, too. The steps over narrow levels are wave 0's alone,
  // in program order; the wide ones are separated by workgroup barriers.
  const int fo = sp ? lb + 1 : ((pt & CONSISTENT) ? lb + 1 : lb);  // the 'orig' field
  const double* __restrict__ const orow = l2r + static_cast<size_t>(fo)*S + in0;
  const double* __restrict__ const lrow = l2r + static_cast<size_t>(lb)*S + in0;        // lo (sp)
  const double* __restrict__ const hrow = l2r + static_cast<size_t>(lb + 2)*S + in0;    // hi (sp)
  const int asteps = mode == 2 ? A.tile.nlvl : 0;
  const int nsteps = asteps + tile.nlvl;
  const int nnarrow = asteps + (tile.nlvl - narrow);  // the first steps: narrow
  TSTAMP(3);
  // a step's offsets and this thread's kid indices do not depend on the step before: read one step ahead
  auto describe = [&] (const int s, int& a, int& n, uint32_t& kk) {
    const bool ap = s < asteps;
    const int lv = ap ? asteps - 1 - s : tile.nlvl - 1 - (s - asteps);
    const int32_t* const LV = ap ? alvl : M.lvlL;
    a = LV[lv]; n = LV[lv+1] - a;
    kk = (tid < n && (ap || lv > 0)) ? (ap ? akids : M.kidsL)[a + tid] : 0;
  };
  int a_nx = 0, n_nx = 0;
  uint32_t kk_nx = 0;
  if (nsteps > 0) describe(0, a_nx, n_nx, kk_nx);
  for (int s = 0; s < nsteps; ++s) {
    if (s < 24) TSTAMP(4 + s);
    const bool narrow_step = s < nnarrow;
    const int a = a_nx, n = n_nx;
    const uint32_t kk_c = kk_nx;
    if (s + 1 < nsteps && (wave == 0 || s + 1 >= nnarrow)) describe(s + 1, a_nx, n_nx, kk_nx);
    if (narrow_step && wave != 0) continue;
    const bool ap = s < asteps;
    const int lv = ap ? asteps - 1 - s : tile.nlvl - 1 - (s - asteps);
    const int thr = ap ? 0 : nleaf;      // kid ids below thr are inputs in global memory
    const int off = ap ? 0 : nleaf;      // LDS index of kid c: c - off
    const int pb = ap ? an : 0;          // LDS index of this level's node j: pb + j
    double* const RC = ap ? arec : recs;
    const double* const RH = ap ? arho : rho;
    const double* const RW = ap ? arw : rw;
    const double* const RQ = ap ? arq : rq;
    const bool regs = ! ap && lv == 0;   // the top tile's lowest level: kids in this thread's registers
    if (mode == 2 && s == asteps) {
      // this rank's boundary roots: to their r2l slots (roots an earlier stage finished are read from
      // there by that stage) and, for this tile's outputs, into the tile. (<= 64 of them: wave 0's lanes.)
      if (narrow_step) wave_lds_order(); else __syncthreads();
      if (tid < A.nmine) {
        const double x = arec[kTRS*my_in_r + 3];
        r2l[rb*S + my_slot_r] = x;
        if (t_only) { r2l[(rb + 1)*S + my_slot_r] = qlo; r2l[(rb + 2)*S + my_slot_r] = qhi; }
        if (my_node_r >= nleaf) recs[kTRS*(my_node_r - nleaf) + 3] = x;
        else if (my_node_r >= 0) put_input(my_node_r, x, qlo, qhi);
      }
    }
    const bool act = tid < n;
    int cc[2] = {off, off};
    double krh[2] = {1, 1}, klo[2] = {0, 0}, kor[2] = {0, 0}, khi[2] = {0, 0};
    if (regs) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        cc[e] = c0r[e]; krh[e] = r0[e]; klo[e] = f0[e][0]; kor[e] = f0[e][1]; khi[e] = f0[e][2];
      }
    } else if (act) {
      cc[0] = kk_c & 0xffff; cc[1] = kk_c >> 16;
#pragma unroll
      for (int e = 0; e < 2; ++e)
        if (cc[e] < thr) {
          krh[e] = rrow[cc[e]];
          kor[e] = orow[cc[e]];
          if (sp) { klo[e] = lrow[cc[e]]; khi[e] = hrow[cc[e]]; }
        }
    }
    KidRho kr[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (regs || cc[e] < thr) {
        const double w = 1.0/krh[e];
        kr[e] = KidRho{krh[e], w, 1.0/w};
      } else if (act) {
        const int ci = cc[e] - off;
        kr[e] = KidRho{RH[ci], RW[ci], RQ[ci]};
        klo[e] = RC[kTRS*ci]; kor[e] = RC[kTRS*ci + 1]; khi[e] = RC[kTRS*ci + 2];
      }
    }
    const int me = pb + a + (act ? tid : 0);
    const double* const P = RC + kTRS*me;
    const double plo = P[0], por = P[1], phi = P[2], prh = RH[me];
    // the results of the level above: wave 0's own (narrow steps), or everybody's
    if (narrow_step) wave_lds_order(); else __syncthreads();
    if ( ! act) continue;
    double xs[2];
    solve_node(pt, prh, t_only ? qlo : plo, por, t_only ? qhi : phi, P[3],
               kr[0], t_only ? qlo : klo[0], kor[0], t_only ? qhi : khi[0],
               kr[1], t_only ? qlo : klo[1], kor[1], t_only ? qhi : khi[1], xs[0], xs[1], prefer);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if (regs || cc[e] < thr) put_input(cc[e], xs[e], qlo, qhi);
      else RC[kTRS*(cc[e] - off) + 3] = xs[e];
    }
  }
  TSTAMP(30);
  if (mode == 2 && tile.nlvl == 0) {  // (a top tile of copies only: the hand-over still has to happen)
    __syncthreads();
    if (tid < A.nmine) {
      const double x = arec[kTRS*my_in_r + 3];
      r2l[rb*S + my_slot_r] = x;
      if (t_only) { r2l[(rb + 1)*S + my_slot_r] = qlo; r2l[(rb + 2)*S + my_slot_r] = qhi; }
      if (my_node_r >= 0) put_input(my_node_r, x, qlo, qhi);
    }
  }
}

// ----------------------------------------------------------- slot shuffles
__global__ void gather_slots_kernel (const double* __restrict__ data, const int64_t stride,
                                     const int nfields, const int32_t* __restrict__ slots,
                                     const int n, double* __restrict__ out) {
  const int idx = blockIdx.x*blockDim.x + threadIdx.x;
  if (idx >= n*nfields) return;
  const int j = idx / nfields, f = idx % nfields;
  out[idx] = data[static_cast<size_t>(f)*stride + slots[j]];
}

__global__ void scatter_slots_kernel (double* __restrict__ data, const int64_t stride,
                                      const int nfields, const int32_t* __restrict__ slots,
                                      const int n, const double* __restrict__ in) {
  const int idx = blockIdx.x*blockDim.x + threadIdx.x;
  if (idx >= n*nfields) return;
  const int q = idx / nfields, f = idx % nfields;
  const int s = slots[q];
  if (s >= 0) data[static_cast<size_t>(f)*stride + s] = in[idx];
}

// The same for many slots (the BFB reducer's leaf-major sends): a 32 x 32 tile goes
// through LDS so that both the reads (along fields) and the writes (along slots, which
// are consecutive for consecutive q in practice) are coalesced.
__global__ __launch_bounds__(kBlock)
void scatter_slots_tiled_kernel (double* __restrict__ data, const int64_t stride,
                                 const int nfields, const int32_t* __restrict__ slots,
                                 const int n, const double* __restrict__ in) {
  __shared__ double tile[32][33];
  const int q0 = blockIdx.x*32, f0 = blockIdx.y*32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int q = q0 + ty + 8*r, f = f0 + tx;
    if (q < n && f < nfields) tile[ty + 8*r][tx] = in[static_cast<size_t>(q)*nfields + f];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int f = f0 + ty + 8*r, q = q0 + tx;
    if (q < n && f < nfields) {
      const int s = slots[q];
      if (s >= 0) data[static_cast<size_t>(f)*stride + s] = tile[tx][ty + 8*r];
    }
  }
}

__global__ void scatter_slots_fm_kernel (double* __restrict__ data, const int64_t stride,
                                         const int nfields, const int32_t* __restrict__ slots,
                                         const int n, const double* __restrict__ in) {
  const int idx = blockIdx.x*blockDim.x + threadIdx.x;
  if (idx >= n*nfields) return;
  const int f = idx / n, q = idx % n;
  const int s = slots[q];
  if (s >= 0) data[static_cast<size_t>(f)*stride + s] = in[idx];
}

__global__ void copy_slots_kernel (double* __restrict__ data, const int64_t stride,
                                   const int nfields, const int32_t* __restrict__ src,
                                   const int32_t* __restrict__ dst, const int n) {
  const int idx = blockIdx.x*blockDim.x + threadIdx.x;
  if (idx >= n*nfields) return;
  const int j = idx / nfields, f = idx % nfields;
  data[static_cast<size_t>(f)*stride + dst[j]] = data[static_cast<size_t>(f)*stride + src[j]];
}

// ------------------------------------------------------------ bulk pack/unpack
__global__ void qlt_set_rhom_kernel (const int n, const double* __restrict__ rhom,
                                     double* __restrict__ l2r) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) l2r[i] = rhom[i];
}

// reference cedr_qlt_inl.hpp:21-58; grid.y = caller tracer index.
__global__ void qlt_set_Qm_kernel (const int32_t* __restrict__ trcr_prob,
                                   const int32_t* __restrict__ trcr_l2r, const int64_t S,
                                   const int n, const double* __restrict__ Qm,
                                   const double* __restrict__ Qm_min,
                                   const double* __restrict__ Qm_max,
                                   const double* __restrict__ Qm_prev, double* __restrict__ l2r) {
  const int ti = blockIdx.y;
  const int pt = trcr_prob[ti], lb = trcr_l2r[ti];
  const size_t src = static_cast<size_t>(ti)*n;
  double* __restrict__ bd = l2r + static_cast<size_t>(lb)*S;
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x) {
    int next;
    if (pt & SHAPEPRESERVE) {
      bd[i] = Qm_min[src + i]; bd[S + i] = Qm[src + i]; bd[2*S + i] = Qm_max[src + i];
      next = 3;
    } else if (pt & CONSISTENT) {
      const double rhom = l2r[i];
      bd[i] = Qm_min[src + i]/rhom; bd[S + i] = Qm[src + i]; bd[2*S + i] = Qm_max[src + i]/rhom;
      next = 3;
    } else {
      bd[i] = Qm[src + i];
      next = 1;
    }
    if (pt & CONSERVE) bd[static_cast<size_t>(next)*S + i] = Qm_prev[src + i];
  }
}

__global__ void qlt_get_Qm_kernel (const int32_t* __restrict__ trcr_r2l, const int64_t S,
                                   const int n, const double* __restrict__ r2l,
                                   double* __restrict__ out) {
  const int ti = blockIdx.y;
  const double* __restrict__ src = r2l + static_cast<size_t>(trcr_r2l[ti])*S;
  for (int i = blockIdx.x*blockDim.x + threadIdx.x; i < n; i += gridDim.x*blockDim.x)
    out[static_cast<size_t>(ti)*n + i] = src[i];
}

// ------------------------------------------------------------------- CAAS
// Cells per workgroup. Round 3: SMALL chunks -- 512 cells = ONE 16-byte load per array and thread, no loop -- beat
// long ones: the streams are kept in flight by the number of resident workgroups, not by a loop's iterations
// (reduce 0.416 -> 0.385 ms, finish 0.253 -> 0.222 ms at 1.5M cells x 40 tracers; 256 cells per workgroup
// leave half the threads idle: reduce 0.51 ms; 1024-16384: monotonically slower).
#ifndef CEDR_CAAS_CHUNK
# define CEDR_CAAS_CHUNK 512
#endif
#ifndef CEDR_CAAS_UNROLL
# define CEDR_CAAS_UNROLL 1
#endif
constexpr int kCaasChunk = CEDR_CAAS_CHUNK;  // cells per workgroup, reduce pass (one partial sum each)
#ifndef CEDR_CAAS_FINISH_CHUNK
# define CEDR_CAAS_FINISH_CHUNK CEDR_CAAS_CHUNK
#endif
constexpr int kCaasFinishChunk = CEDR_CAAS_FINISH_CHUNK;  // ... finish pass

// CAAS streams every array exactly once per pass and nothing is re-read before
// ~2 GB of other data has gone by, so its accesses are non-temporal: +12 % on
// MI355X (5.3 -> 5.9 TB/s on the reduce pass).
#ifndef CEDR_CAAS_NT
# define CEDR_CAAS_NT 1
#endif
// (Measured and rejected: non-temporal accesses in the QLT kernels, -10 %, whose up and down passes re-read
// each other's lines out of MALL/L2; a non-temporal store of the clipped Qm in the reduce pass, neutral.)

__device__ __forceinline__ double wave_sum (double v) {
#pragma unroll
  for (int off = kWave/2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Workgroup sum of four values in a fixed order: lanes by shuffle, then the
// four waves through LDS. Result valid in thread 0.
__device__ __forceinline__ void block_sum4 (double v[4], double* sm /* [4][4] */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v[q] = wave_sum(v[q]);
    if (lane == 0) sm[q*4 + wave] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = ((sm[q*4] + sm[q*4+1]) + sm[q*4+2]) + sm[q*4+3];
  }
}

// reference cedr_caas.cpp:171-199 fused into one pass: clip, and the four sums.
// grid = (nchunks, nt). VEC = 2 uses 16-byte accesses (needs n even).
template <int VEC>
__global__ __launch_bounds__(kBlock)
void caas_reduce_kernel (const int32_t* __restrict__ probs, const int nt, const int n,
                         const int need_conserve, const double* __restrict__ qm_in,
                         double* __restrict__ qm_out, const double* __restrict__ qm_min,
                         const double* __restrict__ qm_max, const double* __restrict__ qm_prev,
                         double* __restrict__ partials) {
  __shared__ double sm[16];
  const int k = blockIdx.y, c = blockIdx.x, nchunks = gridDim.x;
  const size_t N = n;
  // Arrays are [tracer][cell]: the reference's SoA buffer rows (cedr_caas_inl.hpp:22-34)
  // or the caller's own arrays, bound in place.
  double* __restrict__ qm = qm_out + k*N;
  const double* __restrict__ qin = qm_in + k*N;
  const double* __restrict__ qlo = qm_min + k*N;
  const double* __restrict__ qhi = qm_max + k*N;
  const bool conserve = (probs[k] & CONSERVE) != 0;
  const double* __restrict__ qpv = (need_conserve && conserve) ? qm_prev + k*N : nullptr;
  const int beg = c*kCaasChunk, end = min(n, beg + kCaasChunk);
  double acc[4] = {0, 0, 0, 0};
  if (VEC == 2) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    const d2* __restrict__ vin = reinterpret_cast<const d2*>(qin);
    const d2* __restrict__ vlo = reinterpret_cast<const d2*>(qlo);
    const d2* __restrict__ vhi = reinterpret_cast<const d2*>(qhi);
    const d2* __restrict__ vpv = reinterpret_cast<const d2*>(qpv);
    d2* __restrict__ vout = reinterpret_cast<d2*>(qm);
#pragma unroll CEDR_CAAS_UNROLL
    for (int i = beg/2 + threadIdx.x; i < end/2; i += kBlock) {
#if CEDR_CAAS_NT
      const d2 q = __builtin_nontemporal_load(vin + i), lo = __builtin_nontemporal_load(vlo + i),
        hi = __builtin_nontemporal_load(vhi + i);
#else
      const d2 q = vin[i], lo = vlo[i], hi = vhi[i];
#endif
      d2 cl;
      cl.x = rmin(hi.x, rmax(lo.x, q.x));
      cl.y = rmin(hi.y, rmax(lo.y, q.y));
      vout[i] = cl;
#if CEDR_CAAS_NT
      const d2 term = qpv ? __builtin_nontemporal_load(vpv + i) : q;
#else
      const d2 term = qpv ? vpv[i] : q;
#endif
      acc[0] += cl.x; acc[0] += cl.y;
      acc[1] += term.x; acc[1] += term.y;
      acc[2] += lo.x; acc[2] += lo.y;
      acc[3] += hi.x; acc[3] += hi.y;
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += kBlock) {
      const double q = qin[i], lo = qlo[i], hi = qhi[i];
      const double cl = rmin(hi, rmax(lo, q));
      qm[i] = cl;
      acc[0] += cl;
      acc[1] += qpv ? qpv[i] : q;
      acc[2] += lo;
      acc[3] += hi;
    }
  }
  block_sum4(acc, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      partials[(static_cast<size_t>(q)*nt + k)*nchunks + c] = acc[q];
  }
}

// One workgroup per (sum kind, tracer): add the chunk partials in a fixed order -- thread t takes partials
// t, t + 256, ... (independent loads, all in flight), then lanes by shuffle and the four waves in order.
__global__ __launch_bounds__(kBlock)
void caas_sum_partials_kernel (const int nchunks, const double* __restrict__ partials,
                               double* __restrict__ send) {
  __shared__ double sm[kNW];
  const double* __restrict__ p = partials + static_cast<size_t>(blockIdx.x)*nchunks;
  double v = 0;
  for (int c = threadIdx.x; c < nchunks; c += kBlock) v += p[c];
  v = wave_sum(v);
  if ((threadIdx.x & (kWave - 1)) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) send[blockIdx.x] = ((sm[0] + sm[1]) + sm[2]) + sm[3];
}

// reference cedr_caas.cpp:140-168 (user-reducer branch): thread per (field, group).
__global__ void caas_reduce_accum_kernel (const int32_t* __restrict__ probs, const int nt,
                                          const int n, const int need_conserve, const int na,
                                          double* __restrict__ d, const double* __restrict__ qm_src,
                                          double* __restrict__ send) {
  const int nacc = n/na;
  const int j = blockIdx.x*blockDim.x + threadIdx.x;
  const size_t N = n;
  if (j < nt*nacc) {
    const int k = j/nacc, bi = j % nacc;
    const size_t os = (k + 1)*N;
    const bool conserve = (probs[k] & CONSERVE) != 0;
    double ac = 0, at = 0;
    for (int ai = 0; ai < na; ++ai) {
      const size_t i = static_cast<size_t>(na)*bi + ai;
      const double Qm = qm_src ? qm_src[static_cast<size_t>(k)*N + i] : d[os + i];
      const double term = (need_conserve && conserve) ? d[os + N*3*nt + i] : Qm;
      const double clip = rmin(d[os + N*2*nt + i], rmax(d[os + N*nt + i], Qm));
      d[os + i] = clip;
      ac += clip; at += term;
    }
    send[static_cast<size_t>(nacc)*k + bi] = ac;
    send[static_cast<size_t>(nacc)*(nt + k) + bi] = at;
  }
  if (j < 2*nt*nacc) {
    const int k = 2*nt + j/nacc, bi = j % nacc;
    const size_t os = (k - nt + 1)*N;
    double ae = 0;
    for (int ai = 0; ai < na; ++ai) ae += d[os + static_cast<size_t>(na)*bi + ai];
    send[static_cast<size_t>(nacc)*k + bi] = ae;
  }
}

// The same sums with coalesced traffic (the kernel above mirrors the reference's
// thread-per-group loop: at 16 cells per group it runs at a few % of HBM speed). A
// workgroup streams sub-blocks of whole groups through LDS: every thread loads cells
// the way caas_reduce_kernel does, clips, writes Qm back and parks {clip, term, min,
// max} in LDS; then one thread per (quantity, group) adds the group's cells in the
// reference's order, 0 + c0 + c1 + ..., so the partial sums a UserAllReducer sees
// are bit-identical. LDS rows are padded by one double per group, which keeps the
// group-strided reads of the second phase off a single bank.
#ifndef CEDR_CAAS_ACCUM_SUB
# define CEDR_CAAS_ACCUM_SUB 1024
#endif
constexpr int kAccumSub = CEDR_CAAS_ACCUM_SUB;  // cells per sub-block (upper bound; a multiple of n_accum is used)
#ifndef CEDR_CAAS_ACCUM_NSB
# define CEDR_CAAS_ACCUM_NSB 1
#endif
constexpr int kAccumNsb = CEDR_CAAS_ACCUM_NSB;     // sub-blocks per workgroup
constexpr size_t kBfbFusedMaxLds = 80*1024;        // caas_reduce_bfb_kernel: at least two workgroups per CU
constexpr int kBfbFusedWgs = 8192;                 // ... workgroups wanted per launch (tracers are dealt to reach it)
constexpr int kBfbFusedWgsP2 = 1 << 30;            // caas_reduce_bfb_p2_kernel: one tracer per workgroup, like caas_reduce_kernel
                                                   // (1.5M x 40, 2048 tiles: 0.390 ms; two / five / ten per workgroup: 0.423 / 0.439 / 0.445)
template <int VEC>  // 2: 16-byte accesses, two cells per thread and pass (needs n and sb even, 16-byte aligned rows)
__global__ __launch_bounds__(kBlock)
void caas_reduce_accum_lds_kernel (const int32_t* __restrict__ probs, const int nt, const int n,
                                   const int need_conserve, const int na, const int sb,
                                   double* __restrict__ d, const double* __restrict__ qm_src,
                                   double* __restrict__ send) {
  extern __shared__ double sm[];  // [4][sb + sb/na]
  const int k = blockIdx.y;
  const size_t N = n;
  const int nacc = n/na;
  double* __restrict__ qm = d + (1 + k)*N;
  const double* __restrict__ qin = qm_src ? qm_src + k*N : qm;
  const double* __restrict__ qlo = d + (1 + nt + k)*N;
  const double* __restrict__ qhi = d + (1 + 2*static_cast<size_t>(nt) + k)*N;
  const bool conserve = (probs[k] & CONSERVE) != 0;
  const double* __restrict__ qpv = (need_conserve && conserve) ?
    d + (1 + 3*static_cast<size_t>(nt) + k)*N : nullptr;
  const int rowlen = sb + sb/na;
  for (int b = 0; b < kAccumNsb; ++b) {
    const int64_t beg = (static_cast<int64_t>(blockIdx.x)*kAccumNsb + b)*sb;
    if (beg >= n) break;
    const int cnt = static_cast<int>(min(static_cast<int64_t>(sb), n - beg));  // whole groups
    if (VEC == 2) {
      typedef double __attribute__((ext_vector_type(2))) d2;
      const d2* __restrict__ vin = reinterpret_cast<const d2*>(qin + beg);
      const d2* __restrict__ vlo = reinterpret_cast<const d2*>(qlo + beg);
      const d2* __restrict__ vhi = reinterpret_cast<const d2*>(qhi + beg);
      const d2* __restrict__ vpv = reinterpret_cast<const d2*>(qpv ? qpv + beg : nullptr);
      d2* __restrict__ vout = reinterpret_cast<d2*>(qm + beg);
      for (int i2 = threadIdx.x; i2 < cnt/2; i2 += kBlock) {
        const d2 q = __builtin_nontemporal_load(vin + i2), lo = __builtin_nontemporal_load(vlo + i2),
          hi = __builtin_nontemporal_load(vhi + i2);
        d2 cl;
        cl.x = rmin(hi.x, rmax(lo.x, q.x));
        cl.y = rmin(hi.y, rmax(lo.y, q.y));
        vout[i2] = cl;
        const d2 term = qpv ? __builtin_nontemporal_load(vpv + i2) : q;
        const int i = 2*i2, j0 = i + i/na, j1 = (i + 1) + (i + 1)/na;
        sm[j0] = cl.x; sm[j1] = cl.y;
        sm[rowlen + j0] = term.x; sm[rowlen + j1] = term.y;
        sm[2*rowlen + j0] = lo.x; sm[2*rowlen + j1] = lo.y;
        sm[3*rowlen + j0] = hi.x; sm[3*rowlen + j1] = hi.y;
      }
    } else
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
      // streamed once: non-temporal, as in caas_reduce_kernel
      const double q = __builtin_nontemporal_load(qin + beg + i), lo = __builtin_nontemporal_load(qlo + beg + i),
        hi = __builtin_nontemporal_load(qhi + beg + i);
      const double cl = rmin(hi, rmax(lo, q));
      qm[beg + i] = cl;
      const int j = i + i/na;
      sm[j] = cl;
      sm[rowlen + j] = qpv ? __builtin_nontemporal_load(qpv + beg + i) : q;
      sm[2*rowlen + j] = lo;
      sm[3*rowlen + j] = hi;
    }
    __syncthreads();
    const int ng = cnt/na;
    const int g0 = static_cast<int>(beg/na);
    for (int t = threadIdx.x; t < 4*ng; t += kBlock) {
      const int q = t/ng, g = t - q*ng;
      const double* __restrict__ v = sm + q*rowlen + g*(na + 1);
      double acc = 0;
      for (int ai = 0; ai < na; ++ai) acc += v[ai];
      send[static_cast<size_t>(nacc)*(q*nt + k) + g0 + g] = acc;
    }
    __syncthreads();
  }
}

// THE PRODUCTION PAIRING, FUSED. CAAS whose UserAllReducer is this library's BFB tree reducer
// (caas::BfbUserAllReducer): reference cedr_caas.cpp:137-168 emits one partial sum per (field, group of
// n_accum cells) -- 32/n_accum bytes per cell-tracer written to HBM -- and cedr_bfb_tree_allreduce.cpp:91-124
// reads them all back as the leaves of its tree. Here the workgroup that clips the cells under one FIRST-STAGE
// TILE of the reducer's tree (micro tile: <= 64 micro-subtrees of <= 3 leaves, leaf = group) also forms the
// group sums (0 + c0 + c1 + ... in the reference's order), the micro-subtrees and the tile's LDS levels
// (0 + kid0 + kid1, cedr_bfb_tree_allreduce.cpp:117-123) and writes only the tile's outputs into the reducer's
// field-major buffer bd: no partial-sum buffer, no first leaves->root launch. Same operands in the same order
// as the two separate kernels, so the bits are the same.
// grid = (tiles dealt to XCDs, nt). The cells of a tile (nl leaves x na cells, contiguous: the reducer's leaf
// order is the cell-group order) are streamed like caas_reduce_kernel does; {clip, term, min, max} are parked
// in LDS (row q, cell i at i + (i >> sh): a power-of-two skew that keeps the group-strided reads of the second
// phase off one bank); after ONE workgroup barrier wave q owns quantity q: lane i sums the groups of
// micro-subtree i, combines them in registers and walks the tile's LDS levels alone (a wave's LDS accesses
// execute in order: no further barrier). Field of (quantity q, tracer k) = q*nt + k, CAAS's send order.
// NP > 0: every tile has at most 256*VEC*NP cells -- all of a thread's loads are issued before the first use.
// A workgroup takes 'tper' consecutive tracers of its tile (blockIdx.y-th share) and, in the NP > 0 builds, has the
// NEXT tracer's loads in flight while it clips, parks and sums the current one: a workgroup's life is otherwise
// descriptor fetch -> stream loads -> LDS phase -> tree, with loads in flight for a third of it, and at the 4-5
// workgroups a CU's LDS holds that leaves too few bytes in flight to fill the memory system (first version, one
// tracer per workgroup: 0.474 ms for the 2.4 GB of the 1.5M x 40 case, against 0.384 ms for caas_reduce_kernel).
template <int VEC, int NP>
__global__ __launch_bounds__(kBlock)
void caas_reduce_bfb_kernel (const int32_t* __restrict__ probs, const int nt, const int n,
                             const int need_conserve, const int na, const int sh,
                             double* __restrict__ d, const double* __restrict__ qm_src,
                             const cedr_hip_stage st, const int32_t* __restrict__ tile_nl,
                             const int64_t stride, double* __restrict__ bd, const int rowlen,
                             const int pitch, const int tper) {
  extern __shared__ double lds[];
  const int tix = xcd_tile(st.ntiles);
  if (tix < 0) return;
  const int k0 = blockIdx.y*tper, k1 = min(nt, k0 + tper);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  const cedr_hip_tile tile = st.tiles[tix];
  const int nl = tile_nl[tix];
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  int32_t* const outL = reinterpret_cast<int32_t*>(lds + (st.max_nint + 1));
  double* const v = lds + (st.max_nint + 1) + kWave/2;
  double* const park = v + static_cast<size_t>(kNW)*pitch;
  const size_t N = n;
#ifdef CEDR_DIAG_BFB_GEOM  // (diagnostic builds, wrong results: every workgroup streams CEDR_DIAG_BFB_GEOM cells, back to back)
  const int64_t cell0 = static_cast<int64_t>(tix)*CEDR_DIAG_BFB_GEOM;
  const int cnt = CEDR_DIAG_BFB_GEOM + 0*nl;
#else
  const int64_t cell0 = static_cast<int64_t>(tile.in0)*na;
  const int cnt = nl*na;
#endif
  typedef double __attribute__((ext_vector_type(2))) d2;
  // rows of tracer k, at the tile's first cell
  auto row_qm = [&] (const int k) { return d + (1 + k)*N + cell0; };
  auto row_in = [&] (const int k) { return qm_src ? qm_src + k*N + cell0 : d + (1 + k)*N + cell0; };
  auto row_lo = [&] (const int k) { return d + (1 + nt + k)*N + cell0; };
  auto row_hi = [&] (const int k) { return d + (1 + 2*static_cast<size_t>(nt) + k)*N + cell0; };
  auto row_pv = [&] (const int k) -> const double* {
    return (need_conserve && (probs[k] & CONSERVE) != 0) ? d + (1 + 3*static_cast<size_t>(nt) + k)*N + cell0 : nullptr;
  };
  auto park2 = [&] (const int i, const d2 cl, const d2 term, const d2 lo, const d2 hi) {
    const int j0 = i + (i >> sh), j1 = (i + 1) + ((i + 1) >> sh);
    park[j0] = cl.x; park[j1] = cl.y;
    park[rowlen + j0] = term.x; park[rowlen + j1] = term.y;
    park[2*rowlen + j0] = lo.x; park[2*rowlen + j1] = lo.y;
    park[3*rowlen + j0] = hi.x; park[3*rowlen + j1] = hi.y;
  };
  auto park1 = [&] (const int i, const double cl, const double term, const double lo, const double hi) {
    const int j = i + (i >> sh);
    park[j] = cl; park[rowlen + j] = term; park[2*rowlen + j] = lo; park[3*rowlen + j] = hi;
  };
  constexpr int NPR = NP > 0 ? NP : 1;
  struct Regs { d2 q[NPR], lo[NPR], hi[NPR], pv[NPR]; };
  const int cnt2 = cnt >> 1;
  auto load = [&] (const int k, Regs& r) {  // (VEC == 2, NP > 0)
    const d2* __restrict__ vin = reinterpret_cast<const d2*>(row_in(k));
    const d2* __restrict__ vlo = reinterpret_cast<const d2*>(row_lo(k));
    const d2* __restrict__ vhi = reinterpret_cast<const d2*>(row_hi(k));
    const d2* __restrict__ vpv = reinterpret_cast<const d2*>(row_pv(k));
    // Inline-asm loads: the compiler does not count them, so it cannot make the current tracer's work wait for them
    // (with ordinary loads it waited with vmcnt(4..0) at the first use of the PREVIOUS loads' registers, i.e. for the
    // prefetch as well -- vmcnt retires in order); wait_regs() is the matching wait. UNCONDITIONAL: threads past the
    // tile's last cell pair re-read it; a tracer without Qm_prev re-reads its Qm row (cache hits).
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      const int i2 = min(static_cast<int>(threadIdx.x) + kBlock*p, cnt2 - 1);
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r.q[p]) : "v"(vin + i2) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r.lo[p]) : "v"(vlo + i2) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r.hi[p]) : "v"(vhi + i2) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(r.pv[p]) : "v"((vpv ? vpv : vin) + i2) : "memory");
    }
  };
  auto wait_regs = [&] (Regs& r) {
    if constexpr (NPR == 1)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.q[0]), "+v"(r.lo[0]), "+v"(r.hi[0]), "+v"(r.pv[0]) :: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.q[0]), "+v"(r.lo[0]), "+v"(r.hi[0]), "+v"(r.pv[0]),
                   "+v"(r.q[NPR-1]), "+v"(r.lo[NPR-1]), "+v"(r.hi[NPR-1]), "+v"(r.pv[NPR-1]) :: "memory");
  };
  // clip (cedr_caas_inl.hpp:45-57), write Qm back, park {clip, term, min, max} of this thread's cells
  auto clip_park = [&] (const int k, const Regs& r) {
    d2* __restrict__ vout = reinterpret_cast<d2*>(row_qm(k));
    const bool has_pv = row_pv(k) != nullptr;
    (void) has_pv;
#pragma unroll
    for (int p = 0; p < NPR; ++p) {
      const int i2 = threadIdx.x + kBlock*p;
      if (i2 < cnt2) {
        d2 cl;
        cl.x = rmin(r.hi[p].x, rmax(r.lo[p].x, r.q[p].x));
        cl.y = rmin(r.hi[p].y, rmax(r.lo[p].y, r.q[p].y));
#ifndef CEDR_DIAG_BFB_NOCLIPSTORE  // (diagnostic builds, wrong results: parts of the kernel left out)
        vout[i2] = cl;
#endif
#ifndef CEDR_DIAG_BFB_NOPARK
        park2(2*i2, cl, r.pv[p], r.lo[p], r.hi[p]);  // (pv holds Qm itself for a tracer without Qm_prev)
#else
        if (cl.x + cl.y + r.pv[p].x + r.lo[p].x + r.hi[p].y == 1.2345e-300) park[0] = 0;
#endif
      }
    }
  };
  auto clip_park_loop = [&] (const int k) {  // (any tile size, 16- or 8-byte accesses; nothing in flight across tracers)
    double* __restrict__ qm = row_qm(k);
    const double* __restrict__ qin = row_in(k);
    const double* __restrict__ qlo = row_lo(k);
    const double* __restrict__ qhi = row_hi(k);
    const double* __restrict__ qpv = row_pv(k);
    if (VEC == 2) {
      const d2* __restrict__ vin = reinterpret_cast<const d2*>(qin);
      const d2* __restrict__ vlo = reinterpret_cast<const d2*>(qlo);
      const d2* __restrict__ vhi = reinterpret_cast<const d2*>(qhi);
      const d2* __restrict__ vpv = reinterpret_cast<const d2*>(qpv);
      d2* __restrict__ vout = reinterpret_cast<d2*>(qm);
      for (int i2 = threadIdx.x; i2 < cnt2; i2 += kBlock) {
        const d2 q = __builtin_nontemporal_load(vin + i2), lo = __builtin_nontemporal_load(vlo + i2),
          hi = __builtin_nontemporal_load(vhi + i2);
        d2 cl;
        cl.x = rmin(hi.x, rmax(lo.x, q.x));
        cl.y = rmin(hi.y, rmax(lo.y, q.y));
        vout[i2] = cl;
        park2(2*i2, cl, qpv ? __builtin_nontemporal_load(vpv + i2) : q, lo, hi);
      }
    } else {
      for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const double q = __builtin_nontemporal_load(qin + i), lo = __builtin_nontemporal_load(qlo + i),
          hi = __builtin_nontemporal_load(qhi + i);
        const double cl = rmin(hi, rmax(lo, q));
        qm[i] = cl;
        park1(i, cl, qpv ? __builtin_nontemporal_load(qpv + i) : q, lo, hi);
      }
    }
  };

  static_assert(NP <= 2, "wait_regs ties the registers of at most two passes");
  Regs cur, nxt;
  if constexpr (NP > 0) load(k0, cur);
  stage_tile_meta(st, tile, kidsL, lvlL);
  for (int i = threadIdx.x; i < tile.nout; i += kBlock) outL[i] = st.outs[tile.outs_off + i];
  const Micro m = get_micro(st, tile, lane);
  if constexpr (NP > 0) wait_regs(cur);
  const int nleaf = tile.nleaf;
  const double* __restrict__ pq = park + wave*rowlen;
  double* const vr = v + wave*pitch;
  for (int k = k0; k < k1; ++k) {
    if constexpr (NP > 0) {
      if (k + 1 < k1) load(k + 1, nxt);
      clip_park(k, cur);
    } else {
      clip_park_loop(k);
    }
    __syncthreads();
    // wave 'wave' = quantity: group sums of this lane's micro-subtree, 0 + c0 + c1 + ... (cedr_caas.cpp:147-156,161-165)
    double leaf[3] = {0, 0, 0};
#if defined(CEDR_DIAG_BFB_NOTAIL) || defined(CEDR_DIAG_BFB_NOPARK)
    if (na < 0)
#endif
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (s < m.size) {
        const int base = (m.start + s)*na;
        double acc = 0;
        for (int a = 0; a < na; ++a) { const int i = base + a; acc += pq[i + (i >> sh)]; }
        leaf[s] = acc;
      }
    }
    if (lane < nleaf) vr[lane] = micro_up_sum_zi(m.shape, leaf[0], leaf[1], leaf[2]);
#if defined(CEDR_DIAG_BFB_NOTAIL) || defined(CEDR_DIAG_BFB_NOPARK)
    if (na < 0)
#endif
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      wave_lds_order();
      const int a = lvlL[lv], b = lvlL[lv+1];
      for (int j = a + lane; j < b; j += kWave) {
        const uint32_t kk = kidsL[j];
        vr[nleaf + j] = (0.0 + vr[kk & 0xffff]) + vr[kk >> 16];
      }
    }
    wave_lds_order();
    // (the wait for the next tracer's cells comes BEFORE this tracer's output stores are issued: vmcnt counts
    // stores too, so waiting after them would also wait for their write acknowledgements)
    // (unconditional, also after the last tracer, when nothing was loaded: tools/check_prefetch_isa.py follows paths,
    // and a path that skips the wait would reach the copies below with the registers still marked in flight)
    if constexpr (NP > 0) wait_regs(nxt);
    double* __restrict__ orow = bd + static_cast<size_t>(wave*nt + k)*stride + tile.out;
#if defined(CEDR_DIAG_BFB_NOTAIL) || defined(CEDR_DIAG_BFB_NOPARK) || defined(CEDR_DIAG_BFB_NOOUT)
    if (na < 0)
#endif
    for (int o = lane; o < tile.nout; o += kWave) orow[o] = vr[outL[o]];
    if (k + 1 < k1) __syncthreads();  // the next tracer's cells overwrite the parked ones
    if constexpr (NP > 0) cur = nxt;
  }
}

// A first-stage BFB tile's tables as ONE element per thread, held in registers: fetched together with the first tracer's
// cells and written to LDS once those cells have arrived. (The tables hang off the tile descriptor; fetched in front
// of the stream loads, or behind them, they put one more dependent memory round trip into a workgroup's life, and a
// workgroup that is not streaming is occupancy lost: the launch took 0.45 ms either way.) Inline-asm loads like the
// cells': hipcc waits for an ordinary load at its first use, and it hoists the first uses of these -- loop invariants
// -- to the top of the kernel. tie_tile_tables() after the cells' wait marks them arrived.
struct TileTables { uint32_t kid, mic; int32_t lvl, out; };
__device__ __forceinline__ void load_tile_tables (const cedr_hip_stage& st, const cedr_hip_tile& tile, TileTables& r) {
  const int t = threadIdx.x, lane = threadIdx.x & (kWave - 1);
  const uint32_t* const kp = reinterpret_cast<const uint32_t*>(st.kids + 2*static_cast<size_t>(tile.kids_off));
  if (t < tile.nint) asm volatile("global_load_dword %0, %1, off" : "=&v"(r.kid) : "v"(kp + t) : "memory");
  if (t <= tile.nlvl) asm volatile("global_load_dword %0, %1, off" : "=&v"(r.lvl) : "v"(st.lvlptr + tile.lvl_off + t) : "memory");
  if (t < tile.nout) asm volatile("global_load_ushort %0, %1, off" : "=&v"(r.out) : "v"(st.outs + tile.outs_off + t) : "memory");
  if (lane < tile.nleaf) asm volatile("global_load_ushort %0, %1, off" : "=&v"(r.mic) : "v"(st.micro + tile.micro_off + lane) : "memory");
}
__device__ __forceinline__ void tie_tile_tables (TileTables& r) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.kid), "+v"(r.mic), "+v"(r.lvl), "+v"(r.out) :: "memory");
}
__device__ __forceinline__ void store_tile_tables (const TileTables& r, const cedr_hip_tile& tile, uint32_t* kidsL,
                                                   int32_t* lvlL, int32_t* outL) {
  const int t = threadIdx.x;
  if (t < tile.nint) kidsL[t] = r.kid;
  if (t <= tile.nlvl) lvlL[t] = r.lvl;
  if (t < tile.nout) outL[t] = r.out & 0xffff;
}

// Lane L-1's value within a row of 16 lanes (DPP row_shr:1: a VALU move, no LDS traffic); lanes 0, 16, 32, 48 keep x.
__device__ __forceinline__ double row_shr1 (const double x) {
  const int lo = __double2loint(x), hi = __double2hiint(x);
  return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false));
}

// The same fusion for groups of na = 2, 4, 8, 16 or 32 cells (16-byte accesses): the group sums are formed IN REGISTERS.
// Thread i2 of a tile holds the cell pair (2 i2, 2 i2 + 1); a group is g = na/2 consecutive lanes, never split
// over two DPP rows (16 % g == 0, tiles start on a group boundary). The reference's sequential sum
// 0 + c0 + c1 + ... (cedr_caas.cpp:147-156) runs along the lanes of a group: lane j takes lane j-1's partial sum
// (row_shr:1) and adds its own two cells, g - 1 steps; the group's last lane holds the result and writes it to LDS.
// Only the leaf values (4 quantities x <= 192 leaves) touch LDS, double-buffered so that one barrier per tracer is
// enough, and nothing is parked: ~10 KB of LDS and ~50 VGPRs per workgroup instead of 35 KB and 112, i.e. the
// occupancy of caas_reduce_kernel, whose streaming rate this kernel is built to keep (at 4 workgroups per CU
// that kernel itself drops from 0.383 to 0.423 ms, at 3 to 0.50: profiles/r05_experiments.md).
#ifndef CEDR_BFB_P2_WAVES
# define CEDR_BFB_P2_WAVES 7  // waves per SIMD the register allocation aims at (8 = 64 VGPRs: two spills, +4 %)
#endif
template <int NP>
__global__ __launch_bounds__(kBlock, CEDR_BFB_P2_WAVES)
void caas_reduce_bfb_p2_kernel (const int32_t* __restrict__ probs, const int nt, const int n,
                                const int need_conserve, const int lg,
                                double* __restrict__ d, const double* __restrict__ qm_src,
                                const cedr_hip_stage st, const int32_t* __restrict__ tile_nl,
                                const int64_t stride, double* __restrict__ bd, const int leafpitch,
                                const int pitch, const int tper) {
  static_assert(NP == 1 || NP == 2, "wait_regs ties the registers of one or two passes");
  extern __shared__ double lds[];
  const int tix = xcd_tile(st.ntiles);
  if (tix < 0) return;
  const int k0 = blockIdx.y*tper, k1 = min(nt, k0 + tper);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
#ifdef CEDR_DIAG_P2_NODESC  // (diagnostic builds, wrong results: no descriptor fetch in front of the stream loads)
  const cedr_hip_tile tile = {0, 64, 0, 48, 0, 0, 2, 0, 16, 0};
  const int nl = 128;
#else
  const cedr_hip_tile tile = st.tiles[tix];
  const int nl = tile_nl[tix];
#endif
  uint32_t* const kidsL = reinterpret_cast<uint32_t*>(lds);
  int32_t* const lvlL = reinterpret_cast<int32_t*>(kidsL + st.max_nint);
  int32_t* const outL = reinterpret_cast<int32_t*>(lds + (st.max_nint + 1));
  double* const v = lds + (st.max_nint + 1) + kWave/2;
  double* const leafL = v + static_cast<size_t>(kNW)*pitch;  // [2][4][leafpitch]
  const size_t N = n;
  const int g = 1 << lg, j = lane & (g - 1);
#ifdef CEDR_DIAG_BFB_GEOM  // (diagnostic builds, wrong results)
  const int64_t cell0 = static_cast<int64_t>(tix)*CEDR_DIAG_BFB_GEOM;
  const int cnt2 = CEDR_DIAG_BFB_GEOM/2 + 0*nl;
#else
  const int64_t cell0 = static_cast<int64_t>(tile.in0) << (lg + 1);
  const int cnt2 = nl << lg;  // cell pairs of the tile
#endif
  typedef double __attribute__((ext_vector_type(2))) d2;
  const int nleaf = tile.nleaf;
  double* const vr = v + wave*pitch;
  struct Regs { d2 q[NP], lo[NP], hi[NP], pv[NP]; };
  // The cells of tracer k: inline-asm loads (see caas_reduce_bfb_kernel), UNCONDITIONAL -- threads past the tile's last
  // pair re-read it -- so that both passes' loads are issued back to back (with ordinary loads under a per-pass branch
  // the compiler waited for the first pass before it asked for the second). A tracer without Qm_prev re-reads its Qm
  // row for the second sum (cache hits).
  auto load = [&] (const int k, const int b0, Regs& r) {
    // (scalar row bases + one 32-bit byte offset per pass: no 64-bit address arithmetic per load, fewer registers)
    const double* const vin = (qm_src ? qm_src + k*N : d + (1 + k)*N) + cell0;
    const double* const vlo = d + (1 + nt + k)*N + cell0;
    const double* const vhi = d + (1 + 2*static_cast<size_t>(nt) + k)*N + cell0;
    const double* const vpv = (need_conserve && (probs[k] & CONSERVE) != 0) ?
      d + (1 + 3*static_cast<size_t>(nt) + k)*N + cell0 : vin;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#ifdef CEDR_BFB_P2_CLAMP  // (experiment: threads past the tile's last pair re-read it instead of skipping the loads)
      const int i2 = min(b0 + static_cast<int>(threadIdx.x) + kBlock*p, cnt2 - 1);
      const bool on = true;
#else
      const int i2 = b0 + static_cast<int>(threadIdx.x) + kBlock*p;
      const bool on = i2 < cnt2;
#endif
      const int off = i2 << 4;
      if (on) {  // (the compiler does not see the loads: no wait ends up between the passes)
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r.q[p]) : "v"(off), "s"(vin) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r.lo[p]) : "v"(off), "s"(vlo) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r.hi[p]) : "v"(off), "s"(vhi) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(r.pv[p]) : "v"(off), "s"(vpv) : "memory");
      }
    }
  };
  auto wait_regs = [&] (Regs& r) {
    if constexpr (NP == 1)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.q[0]), "+v"(r.lo[0]), "+v"(r.hi[0]), "+v"(r.pv[0]) :: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(r.q[0]), "+v"(r.lo[0]), "+v"(r.hi[0]), "+v"(r.pv[0]),
                   "+v"(r.q[NP-1]), "+v"(r.lo[NP-1]), "+v"(r.hi[NP-1]), "+v"(r.pv[NP-1]) :: "memory");
  };
  // one pass: the pair i2 of tracer k -- clip, write back, the running sums of its group along the lanes
  auto pass = [&] (const int i2, const d2 q, const d2 lo, const d2 hi, const d2 term, d2* __restrict__ vout,
                   double* __restrict__ lf) {
    d2 cl;
    cl.x = rmin(hi.x, rmax(lo.x, q.x));
    cl.y = rmin(hi.y, rmax(lo.y, q.y));
    vout[i2] = cl;
    double s0 = (0.0 + cl.x) + cl.y, s1 = (0.0 + term.x) + term.y, s2 = (0.0 + lo.x) + lo.y, s3 = (0.0 + hi.x) + hi.y;
    for (int step = 1; step < g; ++step) {
      const double t0 = row_shr1(s0), t1 = row_shr1(s1), t2 = row_shr1(s2), t3 = row_shr1(s3);
      if (j == step) {
        s0 = (t0 + cl.x) + cl.y; s1 = (t1 + term.x) + term.y; s2 = (t2 + lo.x) + lo.y; s3 = (t3 + hi.x) + hi.y;
      }
    }
    if (j == g - 1) {
      const int leaf = i2 >> lg;
      lf[leaf] = s0; lf[leafpitch + leaf] = s1; lf[2*leafpitch + leaf] = s2; lf[3*leafpitch + leaf] = s3;
    }
  };
  TileTables tt{0u, 0u, 0, 0};
  Regs r;
  load_tile_tables(st, tile, tt);
  load(k0, 0, r);
  wait_regs(r);
  tie_tile_tables(tt);
  store_tile_tables(tt, tile, kidsL, lvlL, outL);
  const Micro m = decode_micro(tt.mic, lane < nleaf);
  for (int k = k0; k < k1; ++k) {
    d2* __restrict__ vout = reinterpret_cast<d2*>(d + (1 + k)*N + cell0);
    double* __restrict__ lf = leafL + static_cast<size_t>((k - k0) & 1)*4*leafpitch;
    // (a tile of more than NP*512 cells -- groups of 8 cells and more -- takes several rounds of NP passes)
    for (int b0 = 0; b0 < cnt2; b0 += NP*kBlock) {
      if (k != k0 || b0 != 0) { load(k, b0, r); wait_regs(r); }
      // (whole groups are active or not: cnt2 is a multiple of g, and so is every pass's first pair)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i2 = b0 + threadIdx.x + kBlock*p;
        if (i2 < cnt2) pass(i2, r.q[p], r.lo[p], r.hi[p], r.pv[p], vout, lf);
      }
    }
    __syncthreads();  // (the leaf values; before the first tracer also the tile's tables)
#ifdef CEDR_DIAG_P2_NOTAIL
    if (lg >= 0) continue;
#endif
    // wave 'wave' = quantity: this lane's micro-subtree, then the tile's LDS levels, 0 + kid0 + kid1
    const double* __restrict__ lq = lf + wave*leafpitch + m.start;
    if (lane < nleaf) vr[lane] = micro_up_sum_zi(m.shape, lq[0], lq[1], lq[2]);
    for (int lv = 0; lv < tile.nlvl; ++lv) {
      wave_lds_order();
      const int a = lvlL[lv], b = lvlL[lv+1];
      for (int jn = a + lane; jn < b; jn += kWave) {
        const uint32_t kk = kidsL[jn];
        vr[nleaf + jn] = (0.0 + vr[kk & 0xffff]) + vr[kk >> 16];
      }
    }
    wave_lds_order();
    double* __restrict__ orow = bd + static_cast<size_t>(wave*nt + k)*stride + tile.out;
    for (int o = lane; o < tile.nout; o += kWave) orow[o] = vr[outL[o]];
    wave_lds_order();  // this wave's next micro roots overwrite vr
  }
}

// reference cedr_caas.cpp:217-252 (adjust_Qm). grid = (nchunks, nt).
template <int VEC>
__global__ __launch_bounds__(kBlock)
void caas_finish_kernel (const int nt, const int n, double* __restrict__ qm_io,
                         const double* __restrict__ qm_min, const double* __restrict__ qm_max,
                         const double* __restrict__ recv) {
  const int k = blockIdx.y, c = blockIdx.x;
  const double clip_sum = recv[k], sum = recv[nt + k];
  const double m = sum - clip_sum;
  bool toward_max;
  double fac;
  if (m < 0) {
    fac = clip_sum - recv[2*nt + k];
    toward_max = false;
  } else if (m > 0) {
    fac = recv[3*nt + k] - clip_sum;
    toward_max = true;
  } else return;
  if ( ! (fac > 0)) return;
  fac = m/fac;
  const size_t N = n;
  double* __restrict__ qm = qm_io + k*N;
  const double* __restrict__ bnd = (toward_max ? qm_max : qm_min) + k*N;
  const int beg = c*kCaasFinishChunk, end = min(n, beg + kCaasFinishChunk);
  if (VEC == 2) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    d2* __restrict__ vq = reinterpret_cast<d2*>(qm);
    const d2* __restrict__ vb = reinterpret_cast<const d2*>(bnd);
#pragma unroll CEDR_CAAS_UNROLL
    for (int i = beg/2 + threadIdx.x; i < end/2; i += kBlock) {
#if CEDR_CAAS_NT
      d2 q = __builtin_nontemporal_load(vq + i);
      const d2 b = __builtin_nontemporal_load(vb + i);
#else
      d2 q = vq[i];
      const d2 b = vb[i];
#endif
      if (toward_max) {
        q.x += fac*(b.x - q.x); q.x = rmin(b.x, q.x);
        q.y += fac*(b.y - q.y); q.y = rmin(b.y, q.y);
      } else {
        q.x += fac*(q.x - b.x); q.x = rmax(b.x, q.x);
        q.y += fac*(q.y - b.y); q.y = rmax(b.y, q.y);
      }
#if CEDR_CAAS_NT
      __builtin_nontemporal_store(q, vq + i);
#else
      vq[i] = q;
#endif
    }
  } else {
    for (int i = beg + threadIdx.x; i < end; i += kBlock) {
      double q = qm[i];
      const double b = bnd[i];
      if (toward_max) { q += fac*(b - q); q = rmin(b, q); }
      else            { q += fac*(q - b); q = rmax(b, q); }
      qm[i] = q;
    }
  }
}


// ---------------------------------------------------------- cedr::local batch
// One problem per thread through the public cedr::local functions (cedr_local.hpp).
__device__ __host__ inline int local_dispatch (const int kind, const int n, const int flags, const int max_its,
                                               const double* w, const double* a, const double b,
                                               const double* xlo, const double* xhi, const double* y,
                                               double* x) {
  namespace cl = cedr::local;
  switch (kind) {
  case CEDR_LOCAL_BC_QP: return cl::solve_1eq_bc_qp(n, w, a, b, xlo, xhi, y, x, max_its);
  case CEDR_LOCAL_BC_QP_2D: return cl::solve_1eq_bc_qp_2d(w, a, b, xlo, xhi, y, x, (flags & 1) != 0, (flags & 2) != 0);
  case CEDR_LOCAL_CAAS: cl::caas(n, a, b, xlo, xhi, y, x, (flags & 1) != 0); return 0;
  case CEDR_LOCAL_NONNEG_LS: return cl::solve_1eq_nonneg(n, a, b, y, x, w, cl::Method::least_squares);
  case CEDR_LOCAL_NONNEG_CAAS: return cl::solve_1eq_nonneg(n, a, b, y, x, w, cl::Method::caas);
  default: return -3;
  }
}

__global__ void local_batch_kernel (const int kind, const int nprob, const int n, const int flags,
                                    const int max_its, const double* __restrict__ in,
                                    double* __restrict__ xout, int32_t* __restrict__ info) {
  const int p = blockIdx.x*blockDim.x + threadIdx.x;
  if (p >= nprob) return;
  double v[5][16], x[16];
  const double* q = in + static_cast<size_t>(p)*(5*n + 1);
  for (int k = 0; k < 5; ++k)
    for (int i = 0; i < n; ++i) v[k][i] = q[k*n + i];
  for (int i = 0; i < n; ++i) x[i] = 0;
  info[p] = local_dispatch(kind, n, flags, max_its, v[0], v[1], q[5*n], v[2], v[3], v[4], x);
  for (int i = 0; i < n; ++i) xout[static_cast<size_t>(p)*n + i] = x[i];
}

__global__ void node_problem_batch_kernel (const int pt, const int nprob, const int prefer, const int pair,
                                           const double* __restrict__ in, double* __restrict__ out) {
  const int t = blockIdx.x*blockDim.x + threadIdx.x;
  auto kid = [] (const double rh) { const double w = 1.0/rh; return KidRho{rh, w, 1.0/w}; };
  if (pair) {
    if (2*t + 1 >= nprob) return;
    double plo[2], por[2], phi[2], qm[2], l0[2], o0[2], h0[2], l1[2], o1[2], h1[2], x0[2], x1[2];
    KidRho k0[2], k1[2];
    for (int j = 0; j < 2; ++j) {
      const double* p = in + 13*static_cast<size_t>(2*t + j);
      plo[j] = p[1]; por[j] = p[2]; phi[j] = p[3]; qm[j] = p[4];
      k0[j] = kid(p[5]); l0[j] = p[6]; o0[j] = p[7]; h0[j] = p[8];
      k1[j] = kid(p[9]); l1[j] = p[10]; o1[j] = p[11]; h1[j] = p[12];
    }
    node_problems_mass<2>(plo, por, phi, qm, k0, l0, o0, h0, k1, l1, o1, h1, x0, x1, prefer != 0);
    for (int j = 0; j < 2; ++j) { out[2*(2*t + j)] = x0[j]; out[2*(2*t + j) + 1] = x1[j]; }
    return;
  }
  if (t >= nprob) return;
  const double* p = in + 13*static_cast<size_t>(t);
  double x0, x1;
  // a nonnegative tracer's record is its mass alone, k*d[0] (cedr_qlt_inl.hpp:190-198); the kernels
  // keep it in the 'orig' position
  const bool nn = (pt & NONNEGATIVE) != 0;
  solve_node(pt, p[0], p[1], p[2], p[3], p[4], kid(p[5]), p[6], nn ? p[6] : p[7], p[8],
             kid(p[9]), p[10], nn ? p[10] : p[11], p[12], x0, x1, prefer != 0);
  out[2*t] = x0;
  out[2*t + 1] = x1;
}

// ------------------------------------------------------------ launch profiler
// Optional HIP-event timing of every layer-K launch, on the stream the kernel
// is launched on. Off by default; bench.py switches it on for its roofline leg.
enum KernelId : int { K_TREE_UP = 0, K_QLT_DOWN, K_QLT_ROOT, K_GATHER, K_SCATTER, K_COPY,
                      K_SET_RHOM, K_SET_QM, K_GET_QM, K_CAAS_REDUCE, K_CAAS_SUM, K_CAAS_ACCUM,
                      K_CAAS_FINISH, K_APEX, K_LEVEL, K_TOP_UP, K_QLT_TOP, K_CAAS_BFB, K_COUNT };
const char* const kKernelNames[K_COUNT] = {
  "tree_up_kernel", "qlt_down_kernel", "qlt_root_kernel", "gather_slots_kernel",
  "scatter_slots_kernel", "copy_slots_kernel", "qlt_set_rhom_kernel", "qlt_set_Qm_kernel",
  "qlt_get_Qm_kernel", "caas_reduce_kernel", "caas_sum_partials_kernel",
  "caas_reduce_accum_kernel", "caas_finish_kernel", "apex_kernel", "level_kernel", "tree_top_up_kernel",
  "qlt_top_kernel", "caas_reduce_bfb_kernel" };

struct ProfRec { int id; int64_t aux; hipEvent_t a, b; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get () {
    if ( ! pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    (void) hipEventCreate(&e);
    return e;
  }
};
Profiler g_prof;

struct ProfScope {
  hipStream_t s;
  bool on;
  size_t idx;  // (its own record: scopes may nest)
  ProfScope (int id, int64_t aux, hipStream_t stream) : s(stream), on(g_prof.on), idx(0) {
    if ( ! on) return;
    ProfRec r{id, aux, g_prof.get(), g_prof.get()};
    (void) hipEventRecord(r.a, s);
    idx = g_prof.recs.size();
    g_prof.recs.push_back(r);
  }
  ~ProfScope () { if (on && idx < g_prof.recs.size()) (void) hipEventRecord(g_prof.recs[idx].b, s); }
};

inline int launch_status () { return static_cast<int>(hipGetLastError()); }

template <typename K>
int set_lds_limit (K kernel, size_t bytes) {
  if (bytes <= 64*1024) return 0;
  return static_cast<int>(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(bytes)));
}

// Fields per chunk, leaves->root. 16 (4 rows in flight per wave, 74 VGPRs, 6 waves/SIMD) beat 32
// (125 VGPRs, 4 waves/SIMD) and 8 on hardware: 0.426 vs 0.441 / 0.428 ms at 1.5M cells x 161 fields.
#ifndef CEDR_UP_FT
# define CEDR_UP_FT 16
#endif
constexpr int kUpFT = CEDR_UP_FT;
constexpr int kUpFT32MaxTiles = 0;  // stages of at most this many tiles take 32 fields per chunk (0: none)
constexpr int kDownTT = 8;   // tracers per chunk, root->leaves
constexpr int kClsMinTiles = 2048;  // mixed tracer lists: smallest launch that is swept class by class

// kid pairs (4 B each) + level offsets (4 B each), rounded up to doubles
inline size_t meta_doubles (const cedr_hip_stage* st) { return static_cast<size_t>(st->max_nint) + 1; }
// smallest p >= n with p == 1 (mod 32)
inline int pitch_for (int n) { return (n + 30)/32*32 + 1; }
// Spread the chunks of a stage with few tiles over more workgroups, chunks
// dealt evenly. 'target' = workgroups wanted: the leaves->root kernels pipeline
// their loads across the chunks of a workgroup and prefer long loops (one
// workgroup per CU slot is enough); the root->leaves kernels are limited by the
// last, partly idle round of workgroups and prefer ~4 rounds (measured at 187 500
// cells x 40 tracers, the 8-GPU shard: 0.117 -> 0.103 ms).
// (Round 2, five workgroups per CU: 2048 instead of 3072 -- a 1024-tile stage of 10 chunks then runs 2 workgroups of
// 5 chunks per tile instead of 4 + 4 + 2: 0.0549 -> 0.0534 ms at 1.5M cells' second stage, 0.0706 -> 0.0695 ms at the
// shard's first; 5120 and 10240 are slower.)
// (Round 4, later stages cut at three LDS levels: a stage of 1024 tiles is best left unsplit -- its root->leaves launch 0.052 ->
// 0.048 ms at 1.5M cells, 0.060 -> 0.058 ms as the shard's first stage -- while stages of 128 tiles or fewer want all their chunks
// in workgroups of their own: kSplitDown is the number of workgroups a launch may GROW to, 1280 = five per CU, rounded down.)
constexpr int kSplitUp = 1024, kSplitDown = 1280;  // kSplitDown: micro tiles only
#ifndef CEDR_TAIL_K
# define CEDR_TAIL_K 5
#endif
constexpr int kTailK = CEDR_TAIL_K;
// (Round 2: splitting the first stage's tiles too -- 9000 or 17000 workgroups wanted -- does not shorten
// the last, partly idle round: 0.488 -> 0.492 / 0.514 ms.)
inline int chunk_split (int ntiles, int nchunks, int target, bool round_down = false) {
  int y = round_down ? target/ntiles : (target + ntiles - 1)/ntiles;
  if (y > nchunks) y = nchunks;
  if (y < 1) y = 1;
  const int per = (nchunks + y - 1)/y;  // chunks per workgroup
  return (nchunks + per - 1)/per;
}

// The tail split of a first-stage launch (work_item): with 'wgs' resident workgroups per XCD (occupancy x 32 CUs), the
// tiles of an XCD's range that lie past its last full round go to k workgroups each. Only for stages of at least two
// full rounds whose chunks are not split already; CEDR_HIP_TAIL = k (default 5; 0 or 1 = off), CEDR_HIP_TAIL_ROUNDS = r
// (default 0): r more full rounds are split as well. A full round is sized for a kernel that has the whole device; on a
// CU-masked stream, or next to another stream's kernels, the split can fall mid-round (performance only, not results).
template <typename K>
TailSplit tail_split (K kernel, const size_t lds, const cedr_hip_stage* st, const int nchunks, const int ysplit) {
  // The knobs travel with the stage (DeviceStage::upload reads the environment once, bits 16-31 of flags): no getenv
  // on the launch path, which run() takes on every step and which ranks-as-threads tests take concurrently.
  const int ntiles = st->ntiles;
  const int k_env = (st->flags & CEDR_STAGE_TAIL_SET) ? ((st->flags >> 16) & 0x7f) : kTailK;
  const int r_env = (st->flags >> 24) & 0x7f;
  TailSplit t{0, 1};
  if (k_env < 2 || ysplit != 1 || nchunks < 2) return t;
  // (resident workgroups per XCD for this kernel and LDS size: asked once per thread, the answer is kept -- the query
  // is not free; a handful of (kernel, lds) pairs per process. Thread-local: no lock on the launch path.)
  struct Known { const void* k; size_t lds; int dev, wgs; };
  static thread_local std::vector<Known> known;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void) hipGetLastError(); return t; }
  int wgs = -1;
  for (const Known& kn : known)
    if (kn.k == reinterpret_cast<const void*>(kernel) && kn.lds == lds && kn.dev == dev) wgs = kn.wgs;
  if (wgs < 0) {
    int occ = 0, ncu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kBlock, lds) != hipSuccess || occ < 1 ||
        hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) {
      (void) hipGetLastError();
      wgs = 0;
    } else {
      wgs = occ*(ncu/8);
    }
    static const bool verbose = std::getenv("CEDR_HIP_TAIL_VERBOSE") != nullptr;
    if (verbose) std::fprintf(stderr, "cedr: tail split: %d workgroups per CU, %d CUs, lds %zu\n", occ, ncu, lds);
    known.push_back(Known{reinterpret_cast<const void*>(kernel), lds, dev, wgs});
  }
  if (wgs < 1) return t;
  const int per = (ntiles + 7)/8;
  const int rounds = per/wgs - r_env;
  if (rounds < 2 || per % wgs == 0) return t;
  t.nf = rounds*wgs;
  t.k = std::min(k_env, nchunks);
  return t;
}

}  // namespace

extern "C" {

int cedr_hip_tree_up (const cedr_hip_stage* st, int32_t nfields, const uint8_t* op,
                      int64_t stride, double* data, int zero_init, cedr_stream_t stream) {
  return cedr_hip_tree_up_rows(st, nfields, op, stride, data, nullptr, 0, zero_init, stream);
}

int cedr_hip_tree_up_rows (const cedr_hip_stage* st, int32_t nfields, const uint8_t* op,
                           int64_t stride, double* data, const double* const* in_rows,
                           int32_t nrow, int zero_init, cedr_stream_t stream) {
  return cedr_hip_tree_up_rows_sums(st, nfields, op, stride, data, in_rows, nrow, zero_init, 0, stream);
}

int cedr_hip_tree_up_rows_sums (const cedr_hip_stage* st, int32_t nfields, const uint8_t* op,
                                int64_t stride, double* data, const double* const* in_rows,
                                int32_t nrow, int zero_init, int all_sums, cedr_stream_t stream) {
  ProfScope ps(K_TREE_UP, st->ntiles, stream);
  if (st->ntiles == 0 || nfields == 0) return 0;
  if (st->max_nleaf > 256) return static_cast<int>(hipErrorInvalidValue);
  if (in_rows && ( ! st->micro || nrow < 3)) return static_cast<int>(hipErrorInvalidValue);
  const int pitch = pitch_for(st->max_nodes);
  // Stages of few tiles (one round of workgroups or less) are bound by the latency of a chunk, not by
  // occupancy: they take 32 fields per chunk -- half the chunks, twice the loads in flight per wave.
  static const int ft32_max = [] { const char* v = std::getenv("CEDR_HIP_UP_FT32"); return v ? std::atoi(v) : kUpFT32MaxTiles; }();
  const bool ft32 = st->micro && ! in_rows && st->ntiles <= ft32_max;
  const int FTr = ft32 ? 32 : kUpFT;
  const size_t lds0 = sizeof(double)*(meta_doubles(st) + static_cast<size_t>(FTr)*pitch) +
    (in_rows ? sizeof(double*)*static_cast<size_t>(nfields) : 0) +   // row table
    sizeof(int32_t)*static_cast<size_t>(nfields + 2 + kWave);  // + op table, + output node ids
  static const size_t lds_pad = [] { const char* v = std::getenv("CEDR_HIP_UP_LDS_PAD"); return v ? static_cast<size_t>(std::atol(v)) : 0; }();
  const size_t lds = lds0 + lds_pad;  // (experiments: extra LDS bytes per workgroup lower the occupancy)
  if (lds > 160*1024) return static_cast<int>(hipErrorInvalidValue);
  static const int split_up = [] { const char* v = std::getenv("CEDR_HIP_SPLIT_UP"); return v ? std::atoi(v) : kSplitUp; }();
  const dim3 grid(st->ntiles, chunk_split(st->ntiles, (nfields + FTr - 1)/FTr, split_up));
  const dim3 gridx(xcd_grid(st->ntiles), grid.y);  // micro kernels: XCD-contiguous tile ranges
#ifdef CEDR_DIAG_UP_BLOCKSTORE2
  {
    static double* blk = nullptr;
    static size_t have = 0;
    const size_t need = static_cast<size_t>(st->ntiles)*nfields*16;
    if (need > have) {
      if (blk) (void) hipFree(blk);
      if (hipMalloc(reinterpret_cast<void**>(&blk), need*sizeof(double)) != hipSuccess) return -1;
      have = need;
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_diag_blk), &blk, sizeof(blk));
    }
  }
#endif
  if (st->micro) {
    if (st->max_nleaf > kWave) return static_cast<int>(hipErrorInvalidValue);
    auto go = [&] (auto kernel) -> int {
      int e = set_lds_limit(kernel, lds);
      if (e) return e;
      const TailSplit tail = tail_split(kernel, lds, st, (nfields + FTr - 1)/FTr, gridx.y);
      hipLaunchKernelGGL(kernel, dim3(xcd_grid_tail(st->ntiles, tail), gridx.y), dim3(kBlock), lds, stream,
                         *st, nfields, op, stride, data, in_rows, nrow, zero_init, pitch, tail);
      return 0;
    };
    const bool sums = all_sums && ! zero_init;
    const bool sums_zi = all_sums && zero_init;  // (the BFB reducer: every node is 0 + kid0 + kid1)
    int e;
    if (sums_zi && ! ft32)
      e = in_rows ? go(tree_up_micro_kernel<kUpFT, true, true, true>) : go(tree_up_micro_kernel<kUpFT, false, true, true>);
    else if (ft32) e = sums ? go(tree_up_micro_kernel<32, false, true>) : go(tree_up_micro_kernel<32, false, false>);
    else if (in_rows) e = sums ? go(tree_up_micro_kernel<kUpFT, true, true>) : go(tree_up_micro_kernel<kUpFT, true, false>);
    else e = sums ? go(tree_up_micro_kernel<kUpFT, false, true>) : go(tree_up_micro_kernel<kUpFT, false, false>);
    if (e) return e;
  } else if (st->max_nleaf <= 128) {
    int e = set_lds_limit(tree_up_kernel<kUpFT, 2>, lds);
    if (e) return e;
    hipLaunchKernelGGL((tree_up_kernel<kUpFT, 2>), grid, dim3(kBlock), lds, stream,
                       *st, nfields, op, stride, data, zero_init, pitch);
  } else {
    int e = set_lds_limit(tree_up_kernel<kUpFT, 4>, lds);
    if (e) return e;
    hipLaunchKernelGGL((tree_up_kernel<kUpFT, 4>), grid, dim3(kBlock), lds, stream,
                       *st, nfields, op, stride, data, zero_init, pitch);
  }
  return launch_status();
}

int cedr_hip_qlt_down (const cedr_hip_qlt_layout* lay, const cedr_hip_stage* st,
                       double* l2r, double* r2l, cedr_stream_t stream) {
  return cedr_hip_qlt_down_rows(lay, st, l2r, r2l, nullptr, nullptr, 0, stream);
}

int cedr_hip_qlt_down_rows (const cedr_hip_qlt_layout* lay, const cedr_hip_stage* st,
                            double* l2r, double* r2l, const double* const* in_rows,
                            double* const* out_rows, int32_t nrow, cedr_stream_t stream) {
  ProfScope ps(K_QLT_DOWN, st->ntiles, stream);
  if (st->ntiles == 0) return 0;
  if (st->max_nleaf > 256) return static_cast<int>(hipErrorInvalidValue);
  if (( ! in_rows) != ( ! out_rows) || (in_rows && ( ! st->micro || nrow < 3)))
    return static_cast<int>(hipErrorInvalidValue);
  const size_t mn = st->max_nodes;
  const int pitch = pitch_for(kRS*st->max_nodes);
  if (st->micro) {
    if (st->max_nleaf > kWave) return static_cast<int>(hipErrorInvalidValue);
    // Two builds: 2 tracers per wave (8 per chunk) or 1 (4 per chunk). A chunk costs the
    // same whether it is full or not, and a 4-tracer chunk costs ~0.7 of an 8-tracer one
    // (measured at 1.5M cells: 0.092-0.12 vs 0.137 ms): pick what is cheaper for this
    // tracer count -- in effect 1 per wave only for <= 4 tracers.
    const int nt = lay->nt;
    // Every tracer shape preserving: the specialised kernel, which is fastest with one tracer per
    // wave at any tracer count (0.55 vs 0.62 ms at 1.5M cells x 40 tracers: 90 VGPRs, 5 waves/SIMD).
    static const bool sp_on = [] { const char* v = std::getenv("CEDR_HIP_DOWN_SP"); return ! v || std::atoi(v) != 0; }();
    const bool sp = (lay->flags & CEDR_QLT_ALL_SHAPEPRESERVE) && sp_on;
    int tw = (sp || 0.7*((nt + 3)/4) < 1.0*((nt + 7)/8)) ? 1 : 2;
    // (experiment, round 5: launches of at most CEDR_HIP_SP_TW2_MAX_TILES tiles -- one round of workgroups, bound by a
    // chunk's chain and not by occupancy -- take two tracers per wave, i.e. half the chunks; default 0 = none)
    static const int tw2_max = [] { const char* v = std::getenv("CEDR_HIP_SP_TW2_MAX_TILES"); return v ? std::atoi(v) : 0; }();
    if (sp && st->ntiles <= tw2_max) tw = 2;
    int tw_env = 0;  // (CEDR_HIP_MICRO_TW set: the general kernel with that many tracers per wave, also for mixed lists)
    if (const char* e = std::getenv("CEDR_HIP_MICRO_TW")) { tw = std::atoi(e) == 1 ? 1 : 2; tw_env = tw; }
    const int TTm = kNW*tw;
    const size_t ldsm = sizeof(double)*(meta_doubles(st) + 3*mn + kCellRho*kWave +
                                        static_cast<size_t>(TTm)*pitch) +
      (in_rows ? sizeof(double*)*static_cast<size_t>(lay->l2rndps + lay->nt) : 0);
    if (ldsm > 160*1024) return static_cast<int>(hipErrorInvalidValue);
    static const int split_down = [] { const char* v = std::getenv("CEDR_HIP_SPLIT_DOWN"); return v ? std::atoi(v) : kSplitDown; }();
    const dim3 gridm(xcd_grid(st->ntiles), chunk_split(st->ntiles, (nt + TTm - 1)/TTm, split_down, true));
    auto go = [&] (auto kernel) -> int {
      int e = set_lds_limit(kernel, ldsm);
      if (e) return e;
      hipLaunchKernelGGL(kernel, gridm, dim3(kBlock), ldsm, stream,
                         *lay, *st, l2r, r2l, in_rows, out_rows, nrow, pitch);
      return 0;
    };
    int e;
    // Mixed tracer lists (round 5): class by class through the class-specialised forms of the shape-preserving kernel
    // (qlt_down_sp_kernel<1, ROWS, CLS>), each over its range of the bulk order, when the layout says how many tracers
    // of each class there are (class_count; CEDR_HIP_DOWN_CLS=0: the general kernel instead).
    // (the two knobs travel in the layout's flags, read from the environment when the QLT is set up: CEDR_HIP_DOWN_CLS=0,
    // CEDR_HIP_DOWN_CLS_MIN_TILES)
    const bool cls_on = ! (lay->flags & CEDR_QLT_NO_CLASS_FORMS);
    // (only launches of at least kClsMinTiles tiles: three launches over a small stage cost more in launch gaps and
    // per-tile setup than the general kernel's slower chunks -- 1024 tiles: 0.094 against 0.077 ms, 64: 0.055 against 0.014)
    const int cls_min_f = (lay->flags >> 8) & 0xffff;
    const int cls_min = cls_min_f ? cls_min_f : kClsMinTiles;
    const bool by_class = ! sp && sp_on && cls_on && tw_env == 0 && st->ntiles >= cls_min &&
      lay->class_count[0] + lay->class_count[1] + lay->class_count[2] == nt &&
      lay->class_count[0] >= 0 && lay->class_count[1] >= 0 && lay->class_count[2] >= 0;
    if (sp || by_class) {
      const int pitchs = pitch;
      auto gosp = [&] (auto kernel, const cedr_hip_qlt_layout& Lc, double* const* out_rows_c, const int twc) -> int {
        const int TTc = kNW*twc;
        const size_t ldss = sizeof(double)*(meta_doubles(st) + 3*mn + kCellRho*kWave + static_cast<size_t>(TTc)*pitch) +
          (in_rows ? sizeof(double*)*static_cast<size_t>(Lc.l2rndps + Lc.nt) : 0) +
          sizeof(int32_t)*(2*static_cast<size_t>(Lc.nt) + kWave) + sizeof(double)*2*static_cast<size_t>(TTc);
        if (ldss > 160*1024) return static_cast<int>(hipErrorInvalidValue);
        int e2 = set_lds_limit(kernel, ldss);
        if (e2) return e2;
        const int nch = (Lc.nt + TTc - 1)/TTc;
        const unsigned gy = chunk_split(st->ntiles, nch, split_down, true);
        const TailSplit tail = tail_split(kernel, ldss, st, nch, gy);
        hipLaunchKernelGGL(kernel, dim3(xcd_grid_tail(st->ntiles, tail), gy), dim3(kBlock), ldss, stream,
                           Lc, *st, l2r, r2l, in_rows, out_rows_c, nrow, pitchs, tail);
        return 0;
      };
      if (sp) {
        if (tw == 1) e = in_rows ? gosp(qlt_down_sp_kernel<1, true>, *lay, out_rows, 1) : gosp(qlt_down_sp_kernel<1, false>, *lay, out_rows, 1);
        else e = in_rows ? gosp(qlt_down_sp_kernel<2, true>, *lay, out_rows, 2) : gosp(qlt_down_sp_kernel<2, false>, *lay, out_rows, 2);
        if (e) return e;
        return launch_status();
      }
      int off = 0;
      for (int cls = 0; cls < 3; ++cls) {
        const int cnt = lay->class_count[cls];
        if (cnt == 0) continue;
#ifdef CEDR_PROF_BY_CLASS  // (diagnostic builds: one profiler record per class launch, aux = tiles + (class + 1) * 2^24)
        ProfScope psc(K_QLT_DOWN, st->ntiles + (static_cast<int64_t>(cls + 1) << 24), stream);
#endif
        cedr_hip_qlt_layout Lc = *lay;
        Lc.nt = cnt;
        Lc.bidx_prob += off; Lc.bidx_l2r += off; Lc.bidx_r2l += off;
        double* const* const orc = out_rows ? out_rows + off : nullptr;
        if (cls == 0) e = in_rows ? gosp(qlt_down_sp_kernel<1, true, 0>, Lc, orc, 1) : gosp(qlt_down_sp_kernel<1, false, 0>, Lc, orc, 1);
        else if (cls == 1) e = in_rows ? gosp(qlt_down_sp_kernel<1, true, 1>, Lc, orc, 1) : gosp(qlt_down_sp_kernel<1, false, 1>, Lc, orc, 1);
        else e = in_rows ? gosp(qlt_down_sp_kernel<1, true, 2>, Lc, orc, 1) : gosp(qlt_down_sp_kernel<1, false, 2>, Lc, orc, 1);
        if (e) return e;
        off += cnt;
      }
      return launch_status();
    }
    if (tw == 1) e = in_rows ? go(qlt_down_micro_kernel<1, true>) : go(qlt_down_micro_kernel<1, false>);
    else e = in_rows ? go(qlt_down_micro_kernel<2, true>) : go(qlt_down_micro_kernel<2, false>);
    if (e) return e;
    return launch_status();
  }
  // (plain tiles hold up to 511 nodes: their records stay four doubles, padded ones would not fit the LDS)
  const int pitchp = pitch_for(4*st->max_nodes);
  const size_t lds = sizeof(double)*(meta_doubles(st) + 3*mn + static_cast<size_t>(kDownTT)*pitchp);
  if (lds > 160*1024) return static_cast<int>(hipErrorInvalidValue);
  const dim3 grid(st->ntiles, chunk_split(st->ntiles, (lay->nt + kDownTT - 1)/kDownTT, kSplitUp));
  if (st->max_nleaf <= 128) {
    int e = set_lds_limit(qlt_down_kernel<kDownTT, 2>, lds);
    if (e) return e;
    hipLaunchKernelGGL((qlt_down_kernel<kDownTT, 2>), grid, dim3(kBlock), lds, stream,
                       *lay, *st, l2r, r2l, pitchp);
  } else {
    int e = set_lds_limit(qlt_down_kernel<kDownTT, 4>, lds);
    if (e) return e;
    hipLaunchKernelGGL((qlt_down_kernel<kDownTT, 4>), grid, dim3(kBlock), lds, stream,
                       *lay, *st, l2r, r2l, pitchp);
  }
  return launch_status();
}

int cedr_hip_qlt_apex (const cedr_hip_qlt_layout* lay, const cedr_hip_stage* st, const double* recv,
                       const int32_t* src, int32_t nmine, const int32_t* my_in, const int32_t* my_slot,
                       double* r2l, cedr_stream_t stream) {
  ProfScope ps(K_APEX, st->ntiles, stream);
  if (st->ntiles != 1 || st->micro || st->max_nleaf > 64) return static_cast<int>(hipErrorInvalidValue);
  const size_t mn = st->max_nodes;
  const int pitch = static_cast<int>(4*mn) | 1;
  // all tracers in one pass if their records fit in 48 KB of LDS, else as many as do
  const size_t per = sizeof(double)*static_cast<size_t>(pitch);
  int tt = static_cast<int>((48*1024)/per);
  if (tt < 1) return static_cast<int>(hipErrorInvalidValue);
  if (tt > lay->nt) tt = lay->nt;
  const size_t lds = sizeof(double)*(meta_doubles(st) + 3*mn) + per*tt;
  int e = set_lds_limit(qlt_apex_kernel, lds);
  if (e) return e;
  hipLaunchKernelGGL(qlt_apex_kernel, dim3(1), dim3(kBlock), lds, stream, *lay, *st, recv, src, nmine, my_in,
                     my_slot, r2l, tt, pitch);
  return launch_status();
}

constexpr int kTopFPB = 4;  // fields per workgroup, leaves->root over the top tile

// tile: HOST copy of the stage's one tile descriptor (the kernels take it by value)
static bool top_stage_ok (const cedr_hip_stage* st, const cedr_hip_tile* t) {
  return st->ntiles == 1 && ! st->micro && st->max_nleaf <= 2048 && st->max_nodes <= 4095 && t &&
    t->nleaf >= 1 && t->nleaf <= st->max_nleaf && t->nint >= 0 && t->nint <= st->max_nint && t->nlvl >= 0 &&
    t->nlvl <= t->nint && t->nout >= 1;
}

int cedr_hip_tree_top_up (const cedr_hip_stage* st, const cedr_hip_tile* tile, int32_t nfields, const uint8_t* op,
                          int64_t stride, double* data, int zero_init, int32_t nroots, const int32_t* root_slots,
                          double* sendbuf, cedr_stream_t stream) {
  ProfScope ps(K_TOP_UP, st->max_nleaf, stream);
  if (nfields == 0) return 0;
  if ( ! top_stage_ok(st, tile) || (sendbuf && ! root_slots)) return static_cast<int>(hipErrorInvalidValue);
  const size_t lds = sizeof(double)*(static_cast<size_t>(st->max_nint) + 1 +
                                     static_cast<size_t>(kTopFPB)*(st->max_nint | 1));
  int e = set_lds_limit(tree_top_up_kernel<kTopFPB>, lds);
  if (e) return e;
  hipLaunchKernelGGL(tree_top_up_kernel<kTopFPB>, dim3((nfields + kTopFPB - 1)/kTopFPB), dim3(kTopBlock), lds,
                     stream, *st, *tile, nfields, op, stride, data, zero_init, nroots, root_slots, sendbuf);
  return launch_status();
}

int cedr_hip_qlt_top (const cedr_hip_qlt_layout* lay, const cedr_hip_stage* st, const cedr_hip_tile* tile,
                      double* l2r, double* r2l, int mode, const cedr_hip_stage* apex,
                      const cedr_hip_tile* apex_tile, const double* recv, const int32_t* src,
                      int32_t nmine, const int32_t* my_in, const int32_t* my_slot, cedr_stream_t stream) {
  ProfScope ps(K_QLT_TOP, st->max_nleaf, stream);
  if (lay->nt == 0) return 0;
  if ( ! top_stage_ok(st, tile) || mode < 0 || mode > 2 || (mode == 1 && tile->nout != 1))
    return static_cast<int>(hipErrorInvalidValue);
  size_t lds = sizeof(double)*((4 + kTRS)*static_cast<size_t>(st->max_nint) + 1);
  TopApex A;
  A.st = *st; A.tile = *tile; A.recv = nullptr; A.src = nullptr; A.nmine = 0; A.my_in = nullptr; A.my_slot = nullptr;
  if (mode == 2) {
    if ( ! apex || ! apex_tile || apex->ntiles != 1 || apex->micro || apex->max_nleaf > 64 || ! recv || ! src ||
        apex_tile->nleaf > apex->max_nleaf || apex_tile->nint > apex->max_nint || nmine > apex_tile->nleaf ||
        (nmine > 0 && ( ! my_in || ! my_slot)))
      return static_cast<int>(hipErrorInvalidValue);
    A.st = *apex; A.tile = *apex_tile; A.recv = recv; A.src = src; A.nmine = nmine; A.my_in = my_in; A.my_slot = my_slot;
    lds += sizeof(double)*(static_cast<size_t>(apex->max_nint) + 1 + (3 + kTRS)*static_cast<size_t>(apex->max_nodes));
  }
  if (lds > 160*1024) return static_cast<int>(hipErrorInvalidValue);
  int e = set_lds_limit(qlt_top_kernel, lds);
  if (e) return e;
  hipLaunchKernelGGL(qlt_top_kernel, dim3(lay->nt), dim3(kTopBlock), lds, stream, *lay, *st, *tile, l2r, r2l, mode, A);
  return launch_status();
}

int cedr_hip_tree_apex_sum (const cedr_hip_stage* st, const double* recv, const int32_t* src, int32_t nfields,
                            int zero_init, double* out, cedr_stream_t stream) {
  ProfScope ps(K_APEX, st->ntiles, stream);
  if (st->ntiles != 1 || st->micro || st->max_nleaf > 64) return static_cast<int>(hipErrorInvalidValue);
  const int pitch = st->max_nodes | 1;
  int ncols = static_cast<int>((48*1024)/(sizeof(double)*pitch));
  if (ncols > kBlock) ncols = kBlock;
  const size_t lds = sizeof(double)*static_cast<size_t>(pitch)*ncols;
  hipLaunchKernelGGL(tree_apex_sum_kernel, dim3(1), dim3(kBlock), lds, stream, *st, recv, src, nfields, zero_init,
                     out, pitch, ncols);
  return launch_status();
}

int cedr_hip_qlt_root (const cedr_hip_qlt_layout* lay, int32_t root_slot, const double* l2r,
                       double* r2l, cedr_stream_t stream) {
  ProfScope ps(K_QLT_ROOT, 0, stream);
  hipLaunchKernelGGL(qlt_root_kernel, dim3((lay->nt + 63)/64), dim3(64), 0, stream,
                     *lay, root_slot, l2r, r2l);
  return launch_status();
}

int cedr_hip_level_up (const int32_t* nodes, int32_t n, int32_t nfields, const uint8_t* op, int64_t stride,
                       double* data, int zero_init, cedr_stream_t stream) {
  ProfScope ps(K_LEVEL, n, stream);
  if (n <= 0 || nfields <= 0) return 0;
  const int64_t items = static_cast<int64_t>(n)*nfields;
  hipLaunchKernelGGL(level_up_kernel, dim3(static_cast<unsigned>((items + kBlock - 1)/kBlock)), dim3(kBlock), 0,
                     stream, nodes, n, nfields, op, stride, data, zero_init);
  return launch_status();
}

int cedr_hip_qlt_level_down (const cedr_hip_qlt_layout* lay, const int32_t* nodes, int32_t n, double* l2r,
                             double* r2l, cedr_stream_t stream) {
  ProfScope ps(K_LEVEL, n, stream);
  if (n <= 0 || lay->nt <= 0) return 0;
  const int64_t items = static_cast<int64_t>(n)*lay->nt;
  hipLaunchKernelGGL(qlt_level_down_kernel, dim3(static_cast<unsigned>((items + kBlock - 1)/kBlock)), dim3(kBlock),
                     0, stream, *lay, nodes, n, l2r, r2l);
  return launch_status();
}

int cedr_hip_gather_slots (const double* data, int64_t stride, int32_t nfields,
                           const int32_t* slots, int32_t n, double* out, cedr_stream_t stream) {
  ProfScope ps(K_GATHER, n, stream);
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(gather_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, data, stride, nfields, slots, n, out);
  return launch_status();
}

int cedr_hip_scatter_slots (double* data, int64_t stride, int32_t nfields, const int32_t* slots,
                            int32_t n, const double* in, cedr_stream_t stream) {
  ProfScope ps(K_SCATTER, n, stream);
  if (n*nfields == 0) return 0;
  if (n >= 1024)
    hipLaunchKernelGGL(scatter_slots_tiled_kernel, dim3((n + 31)/32, (nfields + 31)/32), dim3(kBlock), 0,
                       stream, data, stride, nfields, slots, n, in);
  else
    hipLaunchKernelGGL(scatter_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                       stream, data, stride, nfields, slots, n, in);
  return launch_status();
}

int cedr_hip_scatter_slots_fm (double* data, int64_t stride, int32_t nfields,
                               const int32_t* slots, int32_t n, const double* in,
                               cedr_stream_t stream) {
  ProfScope ps(K_SCATTER, n, stream);
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(scatter_slots_fm_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock),
                     0, stream, data, stride, nfields, slots, n, in);
  return launch_status();
}

int cedr_hip_copy_slots (double* data, int64_t stride, int32_t nfields, const int32_t* src,
                         const int32_t* dst, int32_t n, cedr_stream_t stream) {
  ProfScope ps(K_COPY, n, stream);
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(copy_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, data, stride, nfields, src, dst, n);
  return launch_status();
}

int cedr_hip_qlt_set_rhom (int64_t, int32_t n, const double* rhom, double* l2r,
                           cedr_stream_t stream) {
  ProfScope ps(K_SET_RHOM, n, stream);
  hipLaunchKernelGGL(qlt_set_rhom_kernel, dim3((n + kBlock - 1)/kBlock), dim3(kBlock), 0, stream,
                     n, rhom, l2r);
  return launch_status();
}

int cedr_hip_qlt_set_Qm (int32_t nt, const int32_t* trcr_prob, const int32_t* trcr_l2r,
                         int64_t stride, int32_t n, const double* Qm, const double* Qm_min,
                         const double* Qm_max, const double* Qm_prev, double* l2r,
                         cedr_stream_t stream) {
  ProfScope ps(K_SET_QM, n, stream);
  const int gx = (n + kBlock - 1)/kBlock;
  hipLaunchKernelGGL(qlt_set_Qm_kernel, dim3(gx < 1024 ? gx : 1024, nt), dim3(kBlock), 0, stream,
                     trcr_prob, trcr_l2r, stride, n, Qm, Qm_min, Qm_max, Qm_prev, l2r);
  return launch_status();
}

int cedr_hip_qlt_get_Qm (int32_t nt, const int32_t* trcr_r2l, int64_t stride, int32_t n,
                         const double* r2l, double* Qm_out, cedr_stream_t stream) {
  ProfScope ps(K_GET_QM, n, stream);
  const int gx = (n + kBlock - 1)/kBlock;
  hipLaunchKernelGGL(qlt_get_Qm_kernel, dim3(gx < 1024 ? gx : 1024, nt), dim3(kBlock), 0, stream,
                     trcr_r2l, stride, n, r2l, Qm_out);
  return launch_status();
}

int32_t cedr_hip_caas_nchunks (int32_t n) { return (n + kCaasChunk - 1)/kCaasChunk; }

int cedr_hip_caas_reduce_local_arrays (const int32_t* probs, int32_t nt, int32_t n,
                                       int need_conserve, const double* qm_in, double* qm_out,
                                       const double* qm_min, const double* qm_max,
                                       const double* qm_prev, double* partials,
                                       cedr_stream_t stream) {
  ProfScope ps(K_CAAS_REDUCE, n, stream);
  const dim3 grid(cedr_hip_caas_nchunks(n), nt);
  bool vec = n % 2 == 0;
  for (const void* p : {static_cast<const void*>(qm_in), static_cast<const void*>(qm_out),
                        static_cast<const void*>(qm_min), static_cast<const void*>(qm_max),
                        static_cast<const void*>(qm_prev)})
    if (p && reinterpret_cast<uintptr_t>(p) % 16) vec = false;
  // (experiments: CEDR_HIP_CAAS_LDS_PAD bytes of unused LDS per workgroup lower the occupancy)
  static const size_t lds_pad = [] { const char* v = std::getenv("CEDR_HIP_CAAS_LDS_PAD"); return v ? static_cast<size_t>(std::atol(v)) : 0; }();
  if (vec)
    hipLaunchKernelGGL(caas_reduce_kernel<2>, grid, dim3(kBlock), lds_pad, stream, probs, nt, n,
                       need_conserve, qm_in, qm_out, qm_min, qm_max, qm_prev, partials);
  else
    hipLaunchKernelGGL(caas_reduce_kernel<1>, grid, dim3(kBlock), lds_pad, stream, probs, nt, n,
                       need_conserve, qm_in, qm_out, qm_min, qm_max, qm_prev, partials);
  return launch_status();
}

int cedr_hip_caas_reduce_local (const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                                double* d, const double* qm_src, double* partials,
                                cedr_stream_t stream) {
  const size_t N = n;
  double* const qm = d + N;
  return cedr_hip_caas_reduce_local_arrays(probs, nt, n, need_conserve, qm_src ? qm_src : qm, qm,
                                           d + (1 + static_cast<size_t>(nt))*N,
                                           d + (1 + 2*static_cast<size_t>(nt))*N,
                                           need_conserve ? d + (1 + 3*static_cast<size_t>(nt))*N : nullptr,
                                           partials, stream);
}

int cedr_hip_caas_sum_partials (int32_t nt, int32_t n, const double* partials, double* send,
                                cedr_stream_t stream) {
  ProfScope ps(K_CAAS_SUM, n, stream);
  hipLaunchKernelGGL(caas_sum_partials_kernel, dim3(4*nt), dim3(kBlock), 0, stream,
                     cedr_hip_caas_nchunks(n), partials, send);
  return launch_status();
}

int cedr_hip_caas_reduce_local_accum (const int32_t* probs, int32_t nt, int32_t n,
                                      int need_conserve, int32_t na, double* d,
                                      const double* qm_src, double* send, cedr_stream_t stream) {
  ProfScope ps(K_CAAS_ACCUM, n, stream);
  if (na < 1 || n % na) return static_cast<int>(hipErrorInvalidValue);
  if (na <= kAccumSub) {
    const int sb = kAccumSub/na*na;
    const size_t lds = sizeof(double)*4*static_cast<size_t>(sb + sb/na);
    const int64_t per = static_cast<int64_t>(sb)*kAccumNsb;
    // 16-byte accesses when every row and every sub-block starts on a 16-byte boundary
    const bool vec = n % 2 == 0 && sb % 2 == 0 && reinterpret_cast<uintptr_t>(d) % 16 == 0 &&
      ( ! qm_src || reinterpret_cast<uintptr_t>(qm_src) % 16 == 0);
    if (vec)
      hipLaunchKernelGGL(caas_reduce_accum_lds_kernel<2>, dim3(static_cast<unsigned>((n + per - 1)/per), nt),
                         dim3(kBlock), lds, stream, probs, nt, n, need_conserve, na, sb, d, qm_src, send);
    else
      hipLaunchKernelGGL(caas_reduce_accum_lds_kernel<1>, dim3(static_cast<unsigned>((n + per - 1)/per), nt),
                         dim3(kBlock), lds, stream, probs, nt, n, need_conserve, na, sb, d, qm_src, send);
    return launch_status();
  }
  const int total = 2*nt*(n/na);
  hipLaunchKernelGGL(caas_reduce_accum_kernel, dim3((total + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, probs, nt, n, need_conserve, na, d, qm_src, send);
  return launch_status();
}

// LDS bytes caas_reduce_bfb_kernel needs for tiles of at most max_tile_leaves leaves of n_accum cells; the skew shift.
static inline int bfb_skew_shift (const int na) { int sh = 0; while ((2 << sh) <= na) ++sh; return sh; }
static inline size_t bfb_fused_lds (const cedr_hip_stage* st, const int na, const int max_tile_leaves, int* rowlen,
                                    int* pitch) {
  const int64_t cells = static_cast<int64_t>(max_tile_leaves)*na;
  *rowlen = static_cast<int>(cells + (cells >> bfb_skew_shift(na))) + 1;
  *pitch = pitch_for(st->max_nodes);
  return sizeof(double)*(meta_doubles(st) + kWave/2 + static_cast<size_t>(kNW)*(*pitch) + 4*static_cast<size_t>(*rowlen));
}
// Groups of 2, 4, 8, 16 or 32 cells take the kernel that sums them in registers (caas_reduce_bfb_p2_kernel; its
// 16-byte accesses need 16-byte aligned rows, which the caller checks); lg = log2(lanes per group).
static inline bool bfb_p2_group (const int na, int* lg) {
  for (int e = 0; e <= 4; ++e) if (na == (2 << e)) { *lg = e; return true; }
  return false;
}

int cedr_hip_caas_bfb_fusable (const cedr_hip_stage* st, int32_t n_accum, int32_t max_tile_leaves, int aligned16,
                               int flags) {
  static_assert(kNW == 4, "one wave per quantity {clip, term, min, max}");
  if ( ! st || ! st->micro || st->ntiles < 1 || st->max_nleaf > kWave || n_accum < 1 || max_tile_leaves < 1 ||
      max_tile_leaves > 3*kWave || static_cast<int64_t>(max_tile_leaves)*n_accum > 8192)
    return 0;
  int rowlen, pitch, lg;
  if (aligned16 && bfb_p2_group(n_accum, &lg) && ! (flags & CEDR_CAAS_BFB_PARK)) return 1;
  return bfb_fused_lds(st, n_accum, max_tile_leaves, &rowlen, &pitch) <= kBfbFusedMaxLds ? 1 : 0;
}

int cedr_hip_caas_reduce_local_bfb (const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                                    int32_t na, double* d, const double* qm_src,
                                    const cedr_hip_stage* st, const int32_t* tile_nleaves,
                                    int32_t max_tile_leaves, int64_t stride, double* bd, int flags,
                                    cedr_stream_t stream) {
  ProfScope ps(K_CAAS_BFB, n, stream);
  // 16-byte accesses when every row and every tile's first cell sit on a 16-byte boundary
  const bool vec = n % 2 == 0 && na % 2 == 0 && reinterpret_cast<uintptr_t>(d) % 16 == 0 &&
    ( ! qm_src || reinterpret_cast<uintptr_t>(qm_src) % 16 == 0);
  if ( ! cedr_hip_caas_bfb_fusable(st, na, max_tile_leaves, vec, flags) || n % na || nt < 1)
    return static_cast<int>(hipErrorInvalidValue);
  int rowlen, pitch;
  const size_t lds = bfb_fused_lds(st, na, max_tile_leaves, &rowlen, &pitch);
  const int sh = bfb_skew_shift(na);
#ifdef CEDR_DIAG_BFB_GEOM
  const int maxcells = CEDR_DIAG_BFB_GEOM;
#else
  const int maxcells = max_tile_leaves*na;
#endif
  int lg = 0;
  bool p2 = vec && bfb_p2_group(na, &lg) && ! (flags & CEDR_CAAS_BFB_PARK);
  if ( ! p2 && lds > kBfbFusedMaxLds) return static_cast<int>(hipErrorInvalidValue);
  // Tracers per workgroup: enough workgroups for several rounds, and otherwise as many tracers as possible
  // behind one descriptor fetch (CEDR_HIP_CAAS_BFB_WGS: workgroups wanted).
  static const int wgs_env = [] { const char* v = std::getenv("CEDR_HIP_CAAS_BFB_WGS"); return v ? std::atoi(v) : 0; }();
  const int wgs_wanted = wgs_env > 0 ? wgs_env : (p2 ? kBfbFusedWgsP2 : kBfbFusedWgs);
  int ysplit = (wgs_wanted + st->ntiles - 1)/st->ntiles;
  if (ysplit > nt) ysplit = nt;
  if (ysplit < 1) ysplit = 1;
  const int tper = (nt + ysplit - 1)/ysplit;
#ifdef CEDR_DIAG_BFB_GEOM
  cedr_hip_stage stg = *st;
  stg.ntiles = std::min(st->ntiles, n/(CEDR_DIAG_BFB_GEOM));
  st = &stg;
#endif
  const dim3 grid(xcd_grid(st->ntiles), (nt + tper - 1)/tper);
  if (p2) {
    const int leafpitch = (max_tile_leaves + 3) | 1;
    const size_t ldsp = sizeof(double)*(meta_doubles(st) + kWave/2 + static_cast<size_t>(kNW)*pitch + 8*static_cast<size_t>(leafpitch));
    auto gop = [&] (auto kernel) -> int {
      const int e = set_lds_limit(kernel, ldsp);
      if (e) return e;
      hipLaunchKernelGGL(kernel, grid, dim3(kBlock), ldsp, stream, probs, nt, n, need_conserve, lg, d, qm_src,
                         *st, tile_nleaves, stride, bd, leafpitch, pitch, tper);
      return 0;
    };
    const int e = maxcells <= 2*kBlock ? gop(caas_reduce_bfb_p2_kernel<1>) : gop(caas_reduce_bfb_p2_kernel<2>);
    if (e) return e;
    return launch_status();
  }
  auto go = [&] (auto kernel) -> int {
    const int e = set_lds_limit(kernel, lds);
    if (e) return e;
    hipLaunchKernelGGL(kernel, grid, dim3(kBlock), lds, stream, probs, nt, n, need_conserve, na, sh, d, qm_src,
                       *st, tile_nleaves, stride, bd, rowlen, pitch, tper);
    return 0;
  };
  int e;
  if (vec) e = maxcells <= 2*kBlock ? go(caas_reduce_bfb_kernel<2, 1>) :
         (maxcells <= 4*kBlock ? go(caas_reduce_bfb_kernel<2, 2>) : go(caas_reduce_bfb_kernel<2, 0>));
  else e = go(caas_reduce_bfb_kernel<1, 0>);
  if (e) return e;
  return launch_status();
}

int cedr_hip_caas_finish_local_arrays (int32_t nt, int32_t n, double* qm, const double* qm_min,
                                       const double* qm_max, const double* recv,
                                       cedr_stream_t stream) {
  ProfScope ps(K_CAAS_FINISH, n, stream);
  const dim3 grid((n + kCaasFinishChunk - 1)/kCaasFinishChunk, nt);
  const bool vec = n % 2 == 0 && reinterpret_cast<uintptr_t>(qm) % 16 == 0 &&
    reinterpret_cast<uintptr_t>(qm_min) % 16 == 0 && reinterpret_cast<uintptr_t>(qm_max) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL(caas_finish_kernel<2>, grid, dim3(kBlock), 0, stream, nt, n, qm, qm_min,
                       qm_max, recv);
  else
    hipLaunchKernelGGL(caas_finish_kernel<1>, grid, dim3(kBlock), 0, stream, nt, n, qm, qm_min,
                       qm_max, recv);
  return launch_status();
}

int cedr_hip_caas_finish_local (int32_t nt, int32_t n, double* d, const double* recv,
                                cedr_stream_t stream) {
  const size_t N = n;
  return cedr_hip_caas_finish_local_arrays(nt, n, d + N, d + (1 + static_cast<size_t>(nt))*N,
                                           d + (1 + 2*static_cast<size_t>(nt))*N, recv, stream);
}

int cedr_local_solve (int kind, int n, int flags, int max_its, const double* w, const double* a, double b,
                      const double* xlo, const double* xhi, const double* y, double* x) {
  if (n < 1 || n > 16 || (kind == CEDR_LOCAL_BC_QP_2D && n != 2)) return -3;
  return local_dispatch(kind, n, flags, max_its, w, a, b, xlo, xhi, y, x);
}

int cedr_hip_local_batch (int kind, int nprob, int n, int flags, int max_its, const double* in_dev,
                          double* x_dev, int32_t* info_dev, cedr_stream_t stream) {
  if (n < 1 || n > 16 || (kind == CEDR_LOCAL_BC_QP_2D && n != 2) || kind < 0 || kind > 4)
    return static_cast<int>(hipErrorInvalidValue);
  if (nprob <= 0) return 0;
  hipLaunchKernelGGL(local_batch_kernel, dim3((nprob + 63)/64), dim3(64), 0, stream, kind, nprob, n, flags,
                     max_its, in_dev, x_dev, info_dev);
  return launch_status();
}

int cedr_hip_node_problem_batch (int problem_type, int nprob, int prefer_mass_con, int pair, const double* in_dev,
                                 double* out_dev, cedr_stream_t stream) {
  if (nprob <= 0) return 0;
  if (pair && ( ! (problem_type & SHAPEPRESERVE) || nprob % 2)) return static_cast<int>(hipErrorInvalidValue);
  const int nthr = pair ? nprob/2 : nprob;
  hipLaunchKernelGGL(node_problem_batch_kernel, dim3((nthr + 63)/64), dim3(64), 0, stream, problem_type, nprob,
                     prefer_mass_con, pair, in_dev, out_dev);
  return launch_status();
}

void cedr_hip_profile_start (void) {
  for (auto& r : g_prof.recs) { g_prof.pool.push_back(r.a); g_prof.pool.push_back(r.b); }
  g_prof.recs.clear();
  g_prof.on = true;
}

int cedr_hip_profile_stop (void) {
  g_prof.on = false;
  (void) hipDeviceSynchronize();
  return static_cast<int>(g_prof.recs.size());
}

int cedr_hip_profile_get (int i, int* id, int64_t* aux, float* ms) {
  if (i < 0 || i >= static_cast<int>(g_prof.recs.size())) return -1;
  const ProfRec& r = g_prof.recs[i];
  *id = r.id;
  *aux = r.aux;
  return static_cast<int>(hipEventElapsedTime(ms, r.a, r.b));
}

#ifdef CEDR_TOP_STAMPS
int cedr_hip_debug_top_stamps (unsigned long long* out) {
  return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_top_stamps), sizeof(unsigned long long)*64));
}
#endif

#ifdef CEDR_STAMPS
int cedr_hip_debug_stamps (unsigned long long* out, int reset) {
  int e = static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long)*64));
  if (reset) {
    unsigned long long z[64] = {0};
    e |= static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)));
  }
  return e;
}
#endif

#ifndef CEDR_KERNELS_SHA
# define CEDR_KERNELS_SHA "unknown"
#endif
const char* cedr_hip_kernels_sha256 (void) { return CEDR_KERNELS_SHA; }

const char* cedr_hip_profile_name (int id) {
  return (id >= 0 && id < K_COUNT) ? kKernelNames[id] : "";
}

}  // extern "C"
