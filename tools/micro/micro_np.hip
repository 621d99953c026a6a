// Microbenchmark (diagnostic, not product): cycles per QLT node problem and per
// dependent fp64 instruction on one SIMD, at 1..3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../compose_amd/csrc/cedr_node_math.hpp"
using namespace cedr_dev;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: dependent v_add_f64 chain; 1: two independent chains; 2: dependent cmp+select chain;
// 3: dependent division chain; 4: node problems, one per iteration, result feeds the next;
// 5: two independent node problems per iteration.
template <int MODE>
__global__ void k (const double* in, double* out, unsigned long long* cyc, int reps) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  const double* p = in + 16*(i % 4096);
  double a = p[0], b = p[1], c = p[2], d = p[3];
  KidRho r0{p[0], 1.0/p[0], 1.0/(1.0/p[0])}, r1{p[1], 1.0/p[1], 1.0/(1.0/p[1])};
  double lo0 = p[4], or0 = p[5], hi0 = p[6], lo1 = p[7], or1 = p[8], hi1 = p[9];
  double lo0b = p[10], or0b = p[11], hi0b = p[12], lo1b = p[13], or1b = p[14], hi1b = p[15];
  double Qm = p[5] + p[8] + 0.01*(p[6] - p[5]);
  double Qmb = p[11] + p[14] - 0.01*(p[11] - p[10]);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a = a + b;
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { a = a + b; c = c + d; }
    } else if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 16; ++u) { a = a < c ? a + b : a - d; }
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a = b/(a + c);
    } else if (MODE == 4) {
      double x0, x1;
      node_problem_mass(lo0 + lo1, or0 + or1, hi0 + hi1, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, false);
      or0 = 0.5*(or0 + x1*0.999); or1 = 0.5*(or1 + x0*1.001);  // keeps the problems changing
      a += x0;
    } else if (MODE == 6) {
      const double plo[2] = {lo0 + lo1, lo0b + lo1b}, por[2] = {or0 + or1, or0b + or1b}, phi[2] = {hi0 + hi1, hi0b + hi1b};
      const double qm[2] = {Qm, Qmb};
      const KidRho k0[2] = {r0, r0}, k1[2] = {r1, r1};
      double l0[2] = {lo0, lo0b}, l1[2] = {lo1, lo1b}, h0[2] = {hi0, hi0b}, h1[2] = {hi1, hi1b};
      const double o0[2] = {or0, or0b}, o1[2] = {or1, or1b};
      double x0[2], x1[2];
      node_problems_mass<2>(plo, por, phi, qm, k0, l0, o0, h0, k1, l1, o1, h1, x0, x1, false);
      or0 = 0.5*(or0 + x1[0]*0.999); or1 = 0.5*(or1 + x0[0]*1.001);
      or0b = 0.5*(or0b + x1[1]*0.999); or1b = 0.5*(or1b + x0[1]*1.001);
      a += x0[0] + x0[1];
    } else {
      double x0, x1, y0, y1;
      node_problem_mass(lo0 + lo1, or0 + or1, hi0 + hi1, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, false);
      node_problem_mass(lo0b + lo1b, or0b + or1b, hi0b + hi1b, Qmb, r0, lo0b, or0b, hi0b, r1, lo1b, or1b, hi1b, y0, y1, false);
      or0 = 0.5*(or0 + x1*0.999); or1 = 0.5*(or1 + x0*1.001);
      or0b = 0.5*(or0b + y1*0.999); or1b = 0.5*(or1b + y0*1.001);
      a += x0 + y0;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[i] = a + c;
  if ((threadIdx.x & 63) == 0) cyc[i >> 6] = t1 - t0;
}

// Bitwise comparison of the lockstep form with two scalar calls, over scenarios: feasible interior,
// needs edges, quick exit, infeasible (bounds adjusted), corner exits; both values of prefer.
__global__ void verify (const double* in, int nprob, int prefer, unsigned* bad) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (2*i + 1 >= nprob) return;
  double plo[2], por[2], phi[2], qm[2], l0[2], l1[2], h0[2], h1[2], o0[2], o1[2], x0[2], x1[2], s0[2], s1[2];
  KidRho k0[2], k1[2];
  for (int j = 0; j < 2; ++j) {
    const double* p = in + 16*(2*i + j);
    k0[j] = KidRho{p[0], 1.0/p[0], 1.0/(1.0/p[0])};
    k1[j] = KidRho{p[1], 1.0/p[1], 1.0/(1.0/p[1])};
    l0[j] = p[4]; o0[j] = p[5]; h0[j] = p[6]; l1[j] = p[7]; o1[j] = p[8]; h1[j] = p[9];
    plo[j] = l0[j] + l1[j]; por[j] = o0[j] + o1[j]; phi[j] = h0[j] + h1[j];
    qm[j] = p[10];
    double a0 = l0[j], a1 = l1[j], b0 = h0[j], b1 = h1[j];
    node_problem_mass(plo[j], por[j], phi[j], qm[j], k0[j], a0, o0[j], b0, k1[j], a1, o1[j], b1, s0[j], s1[j], prefer);
  }
  node_problems_mass<2>(plo, por, phi, qm, k0, l0, o0, h0, k1, l1, o1, h1, x0, x1, prefer);
  for (int j = 0; j < 2; ++j)
    if (__double_as_longlong(x0[j]) != __double_as_longlong(s0[j]) ||
        __double_as_longlong(x1[j]) != __double_as_longlong(s1[j])) atomicAdd(bad, 1u);
}

template <int MODE>
int run (const char* name, int per_iter, const double* din, double* dout, unsigned long long* dcyc, int reps) {
  // (waves per workgroup, workgroups): 1 wave on the chip; 1 per SIMD on one CU; 2 and 3 per SIMD on every CU
  const int cfg[4][2] = {{1, 1}, {4, 1}, {4, 512}, {4, 768}};
  for (int c = 0; c < 4; ++c) {
    const int nw = cfg[c][0], nb = cfg[c][1];
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(64*nw), 0, 0, din, dout, dcyc, reps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw*nb);
    CK(hipMemcpy(h.data(), dcyc, sizeof(unsigned long long)*h.size(), hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += v;
    printf("%-34s waves/WG %d WGs %4d : %8.1f cycles per item (wave average)\n", name, nw, nb,
           s/h.size()/reps/per_iter);
  }
  return 0;
}

int main () {
  const int N = 4096;
  std::vector<double> h(16*N);
  unsigned long long sd = 12345;
  auto u = [&] () { sd = sd*6364136223846793005ULL + 1442695040888963407ULL; return (sd >> 11)*(1.0/9007199254740992.0); };
  for (int i = 0; i < N; ++i) {
    double* p = &h[16*i];
    p[0] = 0.5*(1 + u()); p[1] = 0.5*(1 + u()); p[2] = u(); p[3] = u();
    for (int k = 0; k < 4; ++k) {  // lo, orig, hi for 4 kids
      const double lo = -0.75 + u(), hi = lo + u();
      p[4 + 3*k] = lo; p[5 + 3*k] = lo + (hi - lo)*u(); p[6 + 3*k] = hi;
    }
  }
  double *din, *dout; unsigned long long* dcyc;
  CK(hipMalloc(&din, sizeof(double)*h.size())); CK(hipMalloc(&dout, sizeof(double)*64*4*1024));
  CK(hipMalloc(&dcyc, sizeof(unsigned long long)*4*1024));
  CK(hipMemcpy(din, h.data(), sizeof(double)*h.size(), hipMemcpyHostToDevice));
  const int reps = 2000;
  run<0>("dependent v_add_f64", 16, din, dout, dcyc, reps);
  run<1>("2 independent v_add_f64 chains (per add)", 32, din, dout, dcyc, reps);
  run<2>("dependent cmp+select+add", 16, din, dout, dcyc, reps);
  run<3>("dependent fp64 division (+1 add)", 4, din, dout, dcyc, reps);
  run<4>("node problem (serial)", 1, din, dout, dcyc, reps);
  run<5>("node problem (2 independent, per problem)", 2, din, dout, dcyc, reps);
  run<6>("node problems, lockstep pair (per problem)", 2, din, dout, dcyc, reps);
  {  // verification
    const int M = 1 << 20;
    std::vector<double> v(16*(size_t)M);
    for (int i = 0; i < M; ++i) {
      double* p = &v[16*(size_t)i];
      p[0] = 0.5*(1 + u()); p[1] = 0.5*(1 + u());
      for (int k = 0; k < 2; ++k) { const double lo = -0.75 + u(), hi = lo + u(); p[4+3*k] = lo; p[5+3*k] = lo + (hi-lo)*u(); p[6+3*k] = hi; }
      const double slo = p[4] + p[7], sor = p[5] + p[8], shi = p[6] + p[9];
      const int sc = i % 8;
      if (sc == 0) p[10] = sor;                                   // quick exit
      else if (sc == 1) p[10] = slo - 0.3*u();                    // infeasible below
      else if (sc == 2) p[10] = shi + 0.3*u();                    // infeasible above
      else if (sc == 3) p[10] = slo;                              // lower corner
      else if (sc == 4) p[10] = shi;                              // upper corner
      else if (sc == 5) { p[5] = p[4] - u(); p[10] = slo + (shi - slo)*u(); }  // y outside the box
      else if (sc == 6) p[10] = sor + 1e-3*(u() - 0.5);           // near the original: interior
      else p[10] = slo + (shi - slo)*u();                         // anywhere in range: edges common
    }
    double* dv; unsigned* dbad; unsigned hb;
    CK(hipMalloc(&dv, sizeof(double)*v.size())); CK(hipMalloc(&dbad, 4));
    CK(hipMemcpy(dv, v.data(), sizeof(double)*v.size(), hipMemcpyHostToDevice));
    for (int prefer = 0; prefer < 2; ++prefer) {
      CK(hipMemset(dbad, 0, 4));
      hipLaunchKernelGGL(verify, dim3(M/2/256), dim3(256), 0, 0, dv, M, prefer, dbad);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(&hb, dbad, 4, hipMemcpyDeviceToHost));
      printf("lockstep pair vs two scalar calls, %d problems, prefer=%d: %u mismatching results\n", M, prefer, hb);
    }
  }
  return 0;
}
