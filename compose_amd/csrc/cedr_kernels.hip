// Hand-written HIP kernels for the CEDR limiter hot path on MI355X (gfx950).
// 64-lane wavefronts, LDS-resident subtrees, field-major (SoA) global layout so
// every global access is a unit-stride stream over cells. No MFMA: the path is
// HBM-bound fp64 elementwise / tree-reduction work. Built with
// -ffp-contract=off: the node problems branch on exact comparisons, so every
// product and sum must round exactly as the reference's (unfused) arithmetic.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdint>

#include "../../include/cedr_hip.h"

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kWave = 64;
constexpr double kEps = DBL_EPSILON;

enum : int { CONSERVE = 1, SHAPEPRESERVE = 2, CONSISTENT = 4, NONNEGATIVE = 8 };

// reference cedr_kokkos.hpp:136-139: comparisons, not fmin/fmax.
__device__ __forceinline__ double rmin (double a, double b) { return a < b ? a : b; }
__device__ __forceinline__ double rmax (double a, double b) { return a > b ? a : b; }

// ------------------------------------------------------------------ node math
// reference cedr_local_inl.hpp:69-165 (solve_1eq_bc_qp_2d) with a = (1, 1).
// Multiplications and divisions by the constant 1 are dropped: they are exact,
// so every remaining operation sees the same operands in the same order.
__device__ void qp_2d (const double w0, const double w1, const double b,
                       const double lo0, const double lo1, const double hi0, const double hi1,
                       const double y0, const double y1, double& x0, double& x1,
                       const bool clip, const bool early_exit_on_tol) {
  if (early_exit_on_tol) {
    // calc_r_tol (:14-18) and check_lu (:23-41)
    double ab = fabs(b);
    ab = rmax(ab, fabs(y0));
    ab = rmax(ab, fabs(y1));
    const double r_tol = 1e1*kEps*fabs(ab);
    double r = -b;
    r += lo0; r += lo1;
    if ( ! (fabs(r) <= r_tol)) {
      if (r > 0) { x0 = lo0; x1 = lo1; return; }
      r = -b;
      r += hi0; r += hi1;
      if ( ! (fabs(r) <= r_tol) && r < 0) { x0 = hi0; x1 = hi1; return; }
    }
  }
  const double q0 = 1.0/w0, q1 = 1.0/w1;
  {
    double qmass = 0;
    qmass += q0; qmass += q1;
    double dm = b;
    dm -= y0; dm -= y1;
    const double lambda = dm/qmass;
    x0 = y0 + lambda*q0;
    if ( ! (x0 < lo0 || x0 > hi0)) {
      x1 = y1 + lambda*q1;
      if ( ! (x1 < lo1 || x1 > hi1)) return;
    }
  }
  // Line a'x = b as base + alpha*dir, dir = (-1, 1); intersect the box edges.
  const double base = 0.5*b;
  double al[4];
  al[0] = lo1 - base;      // bottom
  al[1] = -(hi0 - base);   // right
  al[2] = hi1 - base;      // top
  al[3] = -(lo0 - base);   // left
  double mn = al[0], mx = al[0];
  int imin = 0, imax = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    if (al[i] < mn) { mn = al[i]; imin = i; }
    if (al[i] > mx) { mx = al[i]; imax = i; }
  }
  int m0 = -1, m1 = -1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i != imin && i != imax) {
      if (m0 < 0) m0 = i; else if (m1 < 0) m1 = i;
    }
  const double a0 = m0 == 0 ? al[0] : m0 == 1 ? al[1] : m0 == 2 ? al[2] : al[3];
  const double a1 = m1 == 0 ? al[0] : m1 == 1 ? al[1] : m1 == 2 ? al[2] : al[3];
  double obj0, obj1;
  {
    const double t0 = base - a0, t1 = base + a0;
    const double d0 = y0 - t0, d1 = y1 - t1;
    obj0 = 0; obj0 += w0*(d0*d0); obj0 += w1*(d1*d1);
  }
  {
    const double t0 = base - a1, t1 = base + a1;
    const double d0 = y0 - t0, d1 = y1 - t1;
    obj1 = 0; obj1 += w0*(d0*d0); obj1 += w1*(d1*d1);
  }
  const int ai = obj0 <= obj1 ? m0 : m1;
  if (ai == 0 || ai == 2) {
    x1 = ai == 0 ? lo1 : hi1;
    x0 = b - x1;
    if (clip) x0 = rmin(hi0, rmax(lo0, x0));
  } else {
    x0 = ai == 1 ? hi0 : lo0;
    x1 = b - x0;
    if (clip) x1 = rmin(hi1, rmax(lo1, x1));
  }
}

// reference cedr_qlt_inl.hpp:69-99 (r2l_nl_adjust_bounds)
__device__ void adjust_bounds (double& b0, double& b1, const double rh0, const double rh1,
                               const double extra) {
  const double q0 = b0/rh0, q1 = b1/rh1;
  if (extra < 0) {
    const bool first = q0 >= q1;  // i0 = 0 if first
    const double gap = first ? (q1 - q0)*rh0 : (q0 - q1)*rh1;
    if (gap <= extra) { if (first) b0 += extra; else b1 += extra; return; }
  } else {
    const bool first = q0 <= q1;
    const double gap = first ? (q1 - q0)*rh0 : (q0 - q1)*rh1;
    if (gap >= extra) { if (first) b0 += extra; else b1 += extra; return; }
  }
  const double tot = b0 + b1 + extra;
  const double rtot = rh0 + rh1;
  const double qt = tot/rtot;
  b0 = qt*rh0;
  b1 = qt*rh1;
}

// reference cedr_qlt_inl.hpp:120-173 (mass form of the node problem)
__device__ void node_problem_mass (const double p_lo, const double p_orig, const double p_hi,
                                   const double Qm,
                                   const double rh0, double lo0, const double or0, double hi0,
                                   const double rh1, double lo1, const double or1, double hi1,
                                   double& x0, double& x1, const bool prefer) {
  const bool below = Qm < p_lo, above = Qm > p_hi;
  if (below || above) {
    const double tol = 10*kEps;
    const double disc = below ? p_lo - Qm : Qm - p_hi;
    if (disc > tol*(p_hi - p_lo)) {
      if (below) adjust_bounds(lo0, lo1, rh0, rh1, Qm - p_lo);
      else       adjust_bounds(hi0, hi1, rh0, rh1, Qm - p_hi);
    }
  } else if (Qm == p_orig && or0 >= lo0 && or0 <= hi0 && or1 >= lo1 && or1 <= hi1) {
    x0 = or0;
    x1 = or1;
    return;
  }
  x0 = or0;
  x1 = or1;
  qp_2d(1.0/rh0, 1.0/rh1, Qm, lo0, lo1, hi0, hi1, or0, or1, x0, x1, ! prefer, ! prefer);
}

// reference cedr_qlt_inl.hpp:175-203 (dispatch on problem type)
__device__ void solve_node (const int pt, const double rh, const double p_lo,
                            const double p_orig, const double p_hi, const double Qm,
                            const double rh0, const double lo0, const double or0, const double hi0,
                            const double rh1, const double lo1, const double or1, const double hi1,
                            double& x0, double& x1, const bool prefer) {
  if ((pt & CONSISTENT) && ! (pt & SHAPEPRESERVE)) {
    node_problem_mass(p_lo*rh, p_orig, p_hi*rh, Qm,
                      rh0, lo0*rh0, or0, hi0*rh0, rh1, lo1*rh1, or1, hi1*rh1, x0, x1, prefer);
  } else if (pt & NONNEGATIVE) {
    // solve_1eq_nonneg(2, ones, Qm, y, x, w, least_squares), cedr_local_inl.hpp:309-330
    x0 = or0;
    x1 = or1;
    if (Qm < 0) return;
    qp_2d(1.0/rh0, 1.0/rh1, Qm, 0.0, 0.0, Qm, Qm, or0, or1, x0, x1, true, true);
  } else {
    node_problem_mass(p_lo, p_orig, p_hi, Qm, rh0, lo0, or0, hi0, rh1, lo1, or1, hi1,
                      x0, x1, prefer);
  }
}

// --------------------------------------------------------- leaves -> root sweep
// One workgroup = one tile x a group of F fields. A wave owns whole fields;
// its lanes stride over the nodes of a level.
template <int F>
__global__ __launch_bounds__(kBlock)
void tree_up_kernel (const cedr_hip_stage st, const int nfields, const uint8_t* __restrict__ op,
                     const int64_t stride, double* __restrict__ data, const int zero_init) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[blockIdx.x];
  const int MN = st.max_nodes;
  const int f0 = blockIdx.y*F;
  const int nf = min(F, nfields - f0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  const uint16_t* __restrict__ kids = st.kids + 2*static_cast<size_t>(tile.kids_off);
  const int32_t* __restrict__ lvlptr = st.lvlptr + tile.lvl_off;
  const int nleaf = tile.nleaf;

  for (int f = wave; f < nf; f += kBlock/kWave) {
    const double* __restrict__ src = data + static_cast<size_t>(f0 + f)*stride + tile.in0;
    double* v = lds + f*MN;
    for (int i = lane; i < nleaf; i += kWave) v[i] = src[i];
  }
  __syncthreads();
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlptr[lv], b = lvlptr[lv+1];
    for (int f = wave; f < nf; f += kBlock/kWave) {
      double* v = lds + f*MN;
      const int o = op[f0 + f];
      for (int j = a + lane; j < b; j += kWave) {
        const double k0 = v[kids[2*j]], k1 = v[kids[2*j+1]];
        double r;
        if (o == 0) {
          if (zero_init) { r = 0; r += k0; r += k1; }
          else r = k0 + k1;
        } else if (o == 1) r = rmin(k0, k1);
        else r = rmax(k0, k1);
        v[nleaf + j] = r;
      }
    }
    __syncthreads();
  }
  const int root = tile.nint ? nleaf + tile.nint - 1 : 0;
  if (threadIdx.x < nf)
    data[static_cast<size_t>(f0 + threadIdx.x)*stride + tile.out] = lds[threadIdx.x*MN + root];
}

// --------------------------------------------------------- root -> leaves sweep
// One workgroup = one tile x T tracers (bulk order). LDS per tracer: a record
// {lo, orig, hi, X} per node (RS doubles apart), plus rhom per node.
template <int T, int RS>
__global__ __launch_bounds__(kBlock)
void qlt_down_kernel (const cedr_hip_qlt_layout L, const cedr_hip_stage st,
                      double* __restrict__ l2r, double* __restrict__ r2l) {
  extern __shared__ double lds[];
  const cedr_hip_tile tile = st.tiles[blockIdx.x];
  const int MN = st.max_nodes;
  double* const rho = lds;
  double* const recs = lds + ((MN + 1) & ~1);
  const int bi0 = blockIdx.y*T;
  const int ntr = min(T, L.nt - bi0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & (kWave - 1);
  constexpr int NW = kBlock/kWave;
  const uint16_t* __restrict__ kids = st.kids + 2*static_cast<size_t>(tile.kids_off);
  const int32_t* __restrict__ lvlptr = st.lvlptr + tile.lvl_off;
  const int nleaf = tile.nleaf;
  const int64_t S = L.stride;
  const bool prefer = L.prefer_mass_con != 0;

  // Phase 0: inputs' records from HBM. Row ntr is rhom.
  for (int row = wave; row <= ntr; row += NW) {
    if (row == ntr) {
      const double* __restrict__ src = l2r + tile.in0;
      for (int i = lane; i < nleaf; i += kWave) rho[i] = src[i];
      continue;
    }
    const int pt = L.bidx_prob[bi0 + row], lb = L.bidx_l2r[bi0 + row];
    double* R = recs + static_cast<size_t>(row)*MN*RS;
    if (pt & SHAPEPRESERVE) {
      const double* __restrict__ s0 = l2r + static_cast<size_t>(lb)*S + tile.in0;
      const double* __restrict__ s1 = s0 + S;
      const double* __restrict__ s2 = s1 + S;
      for (int i = lane; i < nleaf; i += kWave) {
        R[i*RS] = s0[i]; R[i*RS+1] = s1[i]; R[i*RS+2] = s2[i];
      }
    } else {
      // consistent-only: orig is field 1, q bounds come from the root;
      // nonnegative: orig is field 0.
      const int of = (pt & CONSISTENT) ? 1 : 0;
      const double* __restrict__ s1 = l2r + static_cast<size_t>(lb + of)*S + tile.in0;
      for (int i = lane; i < nleaf; i += kWave) R[i*RS+1] = s1[i];
    }
  }
  __syncthreads();

  // Phase A: re-form the sums of the tile's internal nodes (what the
  // leaves->root sweep computed but did not store).
  for (int lv = 0; lv < tile.nlvl; ++lv) {
    const int a = lvlptr[lv], b = lvlptr[lv+1];
    for (int row = wave; row <= ntr; row += NW) {
      if (row == ntr) {
        for (int j = a + lane; j < b; j += kWave)
          rho[nleaf + j] = rho[kids[2*j]] + rho[kids[2*j+1]];
        continue;
      }
      const int pt = L.bidx_prob[bi0 + row];
      double* R = recs + static_cast<size_t>(row)*MN*RS;
      if (pt & SHAPEPRESERVE) {
        for (int j = a + lane; j < b; j += kWave) {
          const double* k0 = R + kids[2*j]*RS;
          const double* k1 = R + kids[2*j+1]*RS;
          double* me = R + (nleaf + j)*RS;
          me[0] = k0[0] + k1[0];
          me[1] = k0[1] + k1[1];
          me[2] = k0[2] + k1[2];
        }
      } else {
        for (int j = a + lane; j < b; j += kWave)
          R[(nleaf + j)*RS + 1] = R[kids[2*j]*RS + 1] + R[kids[2*j+1]*RS + 1];
      }
    }
    __syncthreads();
  }

  // Phase B: the tile root's assigned mass, then node problems level by level.
  const int root = tile.nint ? nleaf + tile.nint - 1 : 0;
  if (threadIdx.x < ntr) {
    const int rb = L.bidx_r2l[bi0 + threadIdx.x];
    recs[(static_cast<size_t>(threadIdx.x)*MN + root)*RS + 3] =
      r2l[static_cast<size_t>(rb)*S + tile.out];
  }
  __syncthreads();
  for (int lv = tile.nlvl - 1; lv >= 0; --lv) {
    const int a = lvlptr[lv], b = lvlptr[lv+1];
    for (int row = wave; row < ntr; row += NW) {
      const int pt = L.bidx_prob[bi0 + row];
      double* R = recs + static_cast<size_t>(row)*MN*RS;
      double qlo = 0, qhi = 0;
      const bool t_only = (pt & CONSISTENT) && ! (pt & SHAPEPRESERVE);
      if (t_only) {
        // Global q_min/q_max travel root->leaves unchanged (cedr_qlt.cpp:549-560).
        const int rb = L.bidx_r2l[bi0 + row];
        qlo = r2l[static_cast<size_t>(rb + 1)*S + tile.out];
        qhi = r2l[static_cast<size_t>(rb + 2)*S + tile.out];
      }
      for (int j = a + lane; j < b; j += kWave) {
        const int c0 = kids[2*j], c1 = kids[2*j+1], me = nleaf + j;
        const double* P = R + me*RS;
        double* K0 = R + c0*RS;
        double* K1 = R + c1*RS;
        double x0, x1;
        if (t_only)
          solve_node(pt, rho[me], qlo, P[1], qhi, P[3], rho[c0], qlo, K0[1], qhi,
                     rho[c1], qlo, K1[1], qhi, x0, x1, prefer);
        else
          solve_node(pt, rho[me], P[0], P[1], P[2], P[3], rho[c0], K0[0], K0[1], K0[2],
                     rho[c1], K1[0], K1[1], K1[2], x0, x1, prefer);
        K0[3] = x0;
        K1[3] = x1;
      }
    }
    __syncthreads();
  }

  // Phase C: inputs' masses back to HBM.
  for (int row = wave; row < ntr; row += NW) {
    const int pt = L.bidx_prob[bi0 + row], rb = L.bidx_r2l[bi0 + row];
    const double* R = recs + static_cast<size_t>(row)*MN*RS;
    double* __restrict__ dst = r2l + static_cast<size_t>(rb)*S + tile.in0;
    for (int i = lane; i < nleaf; i += kWave) dst[i] = R[i*RS + 3];
    if ((pt & CONSISTENT) && ! (pt & SHAPEPRESERVE)) {
      const int lb = L.bidx_l2r[bi0 + row];
      const double qlo = r2l[static_cast<size_t>(rb + 1)*S + tile.out];
      const double qhi = r2l[static_cast<size_t>(rb + 2)*S + tile.out];
      double* __restrict__ r1 = dst + S;
      double* __restrict__ r2 = r1 + S;
      double* __restrict__ l0 = l2r + static_cast<size_t>(lb)*S + tile.in0;
      double* __restrict__ l2 = l0 + 2*S;
      for (int i = lane; i < nleaf; i += kWave) {
        r1[i] = qlo; r2[i] = qhi; l0[i] = qlo; l2[i] = qhi;
      }
    }
  }
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
constexpr int kCaasChunk = 4096;  // cells per workgroup

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
                         const int need_conserve, double* __restrict__ d,
                         const double* __restrict__ qm_src, double* __restrict__ partials) {
  __shared__ double sm[16];
  const int k = blockIdx.y, c = blockIdx.x, nchunks = gridDim.x;
  const size_t N = n;
  double* __restrict__ qm = d + (1 + k)*N;
  const double* __restrict__ qin = qm_src ? qm_src + k*N : qm;
  const double* __restrict__ qlo = d + (1 + nt + k)*N;
  const double* __restrict__ qhi = d + (1 + 2*static_cast<size_t>(nt) + k)*N;
  const bool conserve = (probs[k] & CONSERVE) != 0;
  const double* __restrict__ qpv = (need_conserve && conserve) ?
    d + (1 + 3*static_cast<size_t>(nt) + k)*N : nullptr;
  const int beg = c*kCaasChunk, end = min(n, beg + kCaasChunk);
  double acc[4] = {0, 0, 0, 0};
  if (VEC == 2) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    const d2* __restrict__ vin = reinterpret_cast<const d2*>(qin);
    const d2* __restrict__ vlo = reinterpret_cast<const d2*>(qlo);
    const d2* __restrict__ vhi = reinterpret_cast<const d2*>(qhi);
    const d2* __restrict__ vpv = reinterpret_cast<const d2*>(qpv);
    d2* __restrict__ vout = reinterpret_cast<d2*>(qm);
    for (int i = beg/2 + threadIdx.x; i < end/2; i += kBlock) {
      const d2 q = vin[i], lo = vlo[i], hi = vhi[i];
      d2 cl;
      cl.x = rmin(hi.x, rmax(lo.x, q.x));
      cl.y = rmin(hi.y, rmax(lo.y, q.y));
      vout[i] = cl;
      const d2 term = qpv ? vpv[i] : q;
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

// One wave per (sum kind, tracer): add the chunk partials in a fixed order.
__global__ __launch_bounds__(kWave)
void caas_sum_partials_kernel (const int nchunks, const double* __restrict__ partials,
                               double* __restrict__ send) {
  const double* __restrict__ p = partials + static_cast<size_t>(blockIdx.x)*nchunks;
  double v = 0;
  for (int c = threadIdx.x; c < nchunks; c += kWave) v += p[c];
  v = wave_sum(v);
  if (threadIdx.x == 0) send[blockIdx.x] = v;
}

// reference cedr_caas.cpp:140-168 (user-reducer branch): thread per (field, group).
__global__ void caas_reduce_accum_kernel (const int32_t* __restrict__ probs, const int nt,
                                          const int n, const int need_conserve, const int na,
                                          double* __restrict__ d, double* __restrict__ send) {
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
      const double Qm = d[os + i];
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

// reference cedr_caas.cpp:217-252 (adjust_Qm). grid = (nchunks, nt).
template <int VEC>
__global__ __launch_bounds__(kBlock)
void caas_finish_kernel (const int nt, const int n, double* __restrict__ d,
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
  double* __restrict__ qm = d + (1 + k)*N;
  const double* __restrict__ bnd = d + (1 + (toward_max ? 2 : 1)*static_cast<size_t>(nt) + k)*N;
  const int beg = c*kCaasChunk, end = min(n, beg + kCaasChunk);
  if (VEC == 2) {
    typedef double __attribute__((ext_vector_type(2))) d2;
    d2* __restrict__ vq = reinterpret_cast<d2*>(qm);
    const d2* __restrict__ vb = reinterpret_cast<const d2*>(bnd);
    for (int i = beg/2 + threadIdx.x; i < end/2; i += kBlock) {
      d2 q = vq[i];
      const d2 b = vb[i];
      if (toward_max) {
        q.x += fac*(b.x - q.x); q.x = rmin(b.x, q.x);
        q.y += fac*(b.y - q.y); q.y = rmin(b.y, q.y);
      } else {
        q.x += fac*(q.x - b.x); q.x = rmax(b.x, q.x);
        q.y += fac*(q.y - b.y); q.y = rmax(b.y, q.y);
      }
      vq[i] = q;
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

inline int launch_status () { return static_cast<int>(hipGetLastError()); }

template <typename K>
int set_lds_limit (K kernel, size_t bytes) {
  if (bytes <= 64*1024) return 0;
  return static_cast<int>(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize,
                                              static_cast<int>(bytes)));
}

constexpr int kUpFields = 16;    // fields per workgroup, leaves->root
constexpr int kDownTracers = 4;  // tracers per workgroup, root->leaves
constexpr int kRecStride = 5;    // doubles between node records in LDS (odd: spreads banks)

}  // namespace

extern "C" {

int cedr_hip_tree_up (const cedr_hip_stage* st, int32_t nfields, const uint8_t* op,
                      int64_t stride, double* data, int zero_init, cedr_stream_t stream) {
  if (st->ntiles == 0 || nfields == 0) return 0;
  const size_t lds = sizeof(double)*kUpFields*static_cast<size_t>(st->max_nodes);
  if (lds > 160*1024) return static_cast<int>(hipErrorInvalidValue);
  int e = set_lds_limit(tree_up_kernel<kUpFields>, lds);
  if (e) return e;
  const dim3 grid(st->ntiles, (nfields + kUpFields - 1)/kUpFields);
  hipLaunchKernelGGL(tree_up_kernel<kUpFields>, grid, dim3(kBlock), lds, stream,
                     *st, nfields, op, stride, data, zero_init);
  return launch_status();
}

int cedr_hip_qlt_down (const cedr_hip_qlt_layout* lay, const cedr_hip_stage* st,
                       double* l2r, double* r2l, cedr_stream_t stream) {
  if (st->ntiles == 0) return 0;
  const size_t mn = st->max_nodes;
  const size_t lds = sizeof(double)*(((mn + 1) & ~static_cast<size_t>(1)) +
                                     kDownTracers*mn*kRecStride);
  if (lds > 160*1024) return static_cast<int>(hipErrorInvalidValue);
  int e = set_lds_limit(qlt_down_kernel<kDownTracers, kRecStride>, lds);
  if (e) return e;
  const dim3 grid(st->ntiles, (lay->nt + kDownTracers - 1)/kDownTracers);
  hipLaunchKernelGGL((qlt_down_kernel<kDownTracers, kRecStride>), grid, dim3(kBlock), lds, stream,
                     *lay, *st, l2r, r2l);
  return launch_status();
}

int cedr_hip_qlt_root (const cedr_hip_qlt_layout* lay, int32_t root_slot, const double* l2r,
                       double* r2l, cedr_stream_t stream) {
  hipLaunchKernelGGL(qlt_root_kernel, dim3((lay->nt + 63)/64), dim3(64), 0, stream,
                     *lay, root_slot, l2r, r2l);
  return launch_status();
}

int cedr_hip_gather_slots (const double* data, int64_t stride, int32_t nfields,
                           const int32_t* slots, int32_t n, double* out, cedr_stream_t stream) {
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(gather_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, data, stride, nfields, slots, n, out);
  return launch_status();
}

int cedr_hip_scatter_slots (double* data, int64_t stride, int32_t nfields, const int32_t* slots,
                            int32_t n, const double* in, cedr_stream_t stream) {
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(scatter_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, data, stride, nfields, slots, n, in);
  return launch_status();
}

int cedr_hip_scatter_slots_fm (double* data, int64_t stride, int32_t nfields,
                               const int32_t* slots, int32_t n, const double* in,
                               cedr_stream_t stream) {
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(scatter_slots_fm_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock),
                     0, stream, data, stride, nfields, slots, n, in);
  return launch_status();
}

int cedr_hip_copy_slots (double* data, int64_t stride, int32_t nfields, const int32_t* src,
                         const int32_t* dst, int32_t n, cedr_stream_t stream) {
  if (n*nfields == 0) return 0;
  hipLaunchKernelGGL(copy_slots_kernel, dim3((n*nfields + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, data, stride, nfields, src, dst, n);
  return launch_status();
}

int cedr_hip_qlt_set_rhom (int64_t, int32_t n, const double* rhom, double* l2r,
                           cedr_stream_t stream) {
  hipLaunchKernelGGL(qlt_set_rhom_kernel, dim3((n + kBlock - 1)/kBlock), dim3(kBlock), 0, stream,
                     n, rhom, l2r);
  return launch_status();
}

int cedr_hip_qlt_set_Qm (int32_t nt, const int32_t* trcr_prob, const int32_t* trcr_l2r,
                         int64_t stride, int32_t n, const double* Qm, const double* Qm_min,
                         const double* Qm_max, const double* Qm_prev, double* l2r,
                         cedr_stream_t stream) {
  const int gx = (n + kBlock - 1)/kBlock;
  hipLaunchKernelGGL(qlt_set_Qm_kernel, dim3(gx < 1024 ? gx : 1024, nt), dim3(kBlock), 0, stream,
                     trcr_prob, trcr_l2r, stride, n, Qm, Qm_min, Qm_max, Qm_prev, l2r);
  return launch_status();
}

int cedr_hip_qlt_get_Qm (int32_t nt, const int32_t* trcr_r2l, int64_t stride, int32_t n,
                         const double* r2l, double* Qm_out, cedr_stream_t stream) {
  const int gx = (n + kBlock - 1)/kBlock;
  hipLaunchKernelGGL(qlt_get_Qm_kernel, dim3(gx < 1024 ? gx : 1024, nt), dim3(kBlock), 0, stream,
                     trcr_r2l, stride, n, r2l, Qm_out);
  return launch_status();
}

int32_t cedr_hip_caas_nchunks (int32_t n) { return (n + kCaasChunk - 1)/kCaasChunk; }

int cedr_hip_caas_reduce_local (const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                                double* d, const double* qm_src, double* partials,
                                cedr_stream_t stream) {
  const dim3 grid(cedr_hip_caas_nchunks(n), nt);
  const bool vec = n % 2 == 0 && reinterpret_cast<uintptr_t>(d) % 16 == 0 &&
    ( ! qm_src || reinterpret_cast<uintptr_t>(qm_src) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL(caas_reduce_kernel<2>, grid, dim3(kBlock), 0, stream, probs, nt, n,
                       need_conserve, d, qm_src, partials);
  else
    hipLaunchKernelGGL(caas_reduce_kernel<1>, grid, dim3(kBlock), 0, stream, probs, nt, n,
                       need_conserve, d, qm_src, partials);
  return launch_status();
}

int cedr_hip_caas_sum_partials (int32_t nt, int32_t n, const double* partials, double* send,
                                cedr_stream_t stream) {
  hipLaunchKernelGGL(caas_sum_partials_kernel, dim3(4*nt), dim3(kWave), 0, stream,
                     cedr_hip_caas_nchunks(n), partials, send);
  return launch_status();
}

int cedr_hip_caas_reduce_local_accum (const int32_t* probs, int32_t nt, int32_t n,
                                      int need_conserve, int32_t na, double* d, double* send,
                                      cedr_stream_t stream) {
  if (na < 1 || n % na) return static_cast<int>(hipErrorInvalidValue);
  const int total = 2*nt*(n/na);
  hipLaunchKernelGGL(caas_reduce_accum_kernel, dim3((total + kBlock - 1)/kBlock), dim3(kBlock), 0,
                     stream, probs, nt, n, need_conserve, na, d, send);
  return launch_status();
}

int cedr_hip_caas_finish_local (int32_t nt, int32_t n, double* d, const double* recv,
                                cedr_stream_t stream) {
  const dim3 grid(cedr_hip_caas_nchunks(n), nt);
  const bool vec = n % 2 == 0 && reinterpret_cast<uintptr_t>(d) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL(caas_finish_kernel<2>, grid, dim3(kBlock), 0, stream, nt, n, d, recv);
  else
    hipLaunchKernelGGL(caas_finish_kernel<1>, grid, dim3(kBlock), 0, stream, nt, n, d, recv);
  return launch_status();
}

}  // extern "C"
