// Microbenchmark (diagnostic): issue cost of the fp64-path instructions the node problem is made of,
// per SIMD with 1 / 2 / 3 waves resident (independent instruction streams, 8 chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define BODY8(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7)
template <int MODE>
__global__ void k (const double* in, double* out, unsigned long long* cyc, int reps) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  double a[8], b[8];
  for (int j = 0; j < 8; ++j) { a[j] = in[(i + j) % 4096]; b[j] = in[(i + 7*j + 3) % 4096] + 0.5; }
  unsigned long long m = 0x5555555555555555ull;
  int x = i;
  int xi[8] = {i, i+1, i+2, i+3, i+4, i+5, i+6, i+7};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
#define S(j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 1) {
#define S(j) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 2) {
#define S(j) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 3) {
#define S(j) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(a[j]), "v"(b[j]) : "vcc");
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 4) {
#define S(j) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(a[j]), "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 5) {
#define S(j) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(i) : "vcc");
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 6) {
#define S(j) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 7) {
#define S(j) asm volatile("v_rcp_f64 %0, %1" : "=v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 8) {
#define S(j) asm volatile("v_div_scale_f64 %0, vcc, %1, %1, %2" : "=v"(a[j]) : "v"(b[j]), "v"(b[(j+1)&7]) : "vcc");
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 9) {
#define S(j) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b[j]), "v"(b[(j+1)&7]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 10) {
#define S(j) asm volatile("v_mov_b64 %0, %1" : "=v"(a[j]) : "v"(b[j]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 11) {
#define S(j) asm volatile("s_and_b64 %0, %0, exec" : "+s"(m) :: "scc");
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 13) {  // independent destinations, mask in vcc
#define S(j) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(xi[j]) : "v"(i), "v"(x) : "vcc");
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 14) {  // independent destinations, mask in an SGPR pair (VOP3)
#define S(j) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(xi[j]) : "v"(i), "v"(x), "s"(m));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 15) {
#define S(j) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(xi[j]) : "v"(i), "v"(x), "v"(xi[(j+1)&7]));
      BODY8(S) BODY8(S)
#undef S
    } else if (MODE == 16) {  // a 64-bit select as the compiler writes it: two v_cndmask on the halves
#define S(j) asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\n\tv_cndmask_b32_e64 %1, %2, %3, %4" : "=v"(xi[j]), "=v"(xi[(j+4)&7]) : "v"(i), "v"(x), "s"(m));
      BODY8(S)
#undef S
    } else if (MODE == 17) {  // the compiler's usual 64-bit select through VCC: 3 instructions per group
#define S(j) asm volatile("v_cmp_lt_f64 vcc, %2, %3\n\tv_cndmask_b32 %0, %4, %5, vcc\n\tv_cndmask_b32 %1, %5, %4, vcc" : "=&v"(xi[j]), "=&v"(xi[(j+4)&7]) : "v"(a[j]), "v"(b[j]), "v"(i), "v"(x) : "vcc");
      BODY8(S)
#undef S
    } else if (MODE == 18) {  // the same through an SGPR pair
#define S(j) asm volatile("v_cmp_lt_f64 %6, %2, %3\n\tv_cndmask_b32_e64 %0, %4, %5, %6\n\tv_cndmask_b32_e64 %1, %5, %4, %6" : "=&v"(xi[j]), "=&v"(xi[(j+4)&7]) : "v"(a[j]), "v"(b[j]), "v"(i), "v"(x), "s"(m));
      BODY8(S)
#undef S
    } else if (MODE == 12) {
#define S(j) asm volatile("v_cmp_lt_i32 vcc, %0, %1" :: "v"(x), "v"(i) : "vcc");
      BODY8(S) BODY8(S)
#undef S
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for (int j = 0; j < 8; ++j) s += a[j];
  for (int j = 0; j < 8; ++j) x += xi[j];
  out[i] = s + x + (double) m;
  if ((threadIdx.x & 63) == 0) cyc[i >> 6] = t1 - t0;
}

template <int MODE>
int run (const char* name, const double* din, double* dout, unsigned long long* dcyc) {
  const int reps = 500;
  const int cfg[3][2] = {{4, 1}, {4, 512}, {4, 768}};
  printf("%-36s", name);
  for (int c = 0; c < 3; ++c) {
    const int nw = cfg[c][0], nb = cfg[c][1];
    hipLaunchKernelGGL(k<MODE>, dim3(nb), dim3(64*nw), 0, 0, din, dout, dcyc, reps);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw*nb);
    CK(hipMemcpy(h.data(), dcyc, sizeof(unsigned long long)*h.size(), hipMemcpyDeviceToHost));
    double s = 0; for (auto v : h) s += v;
    const double per_wave = s/h.size()/reps/16;
    const int wps = c == 0 ? 1 : (c == 1 ? 2 : 3);
    printf("  %dw/SIMD: %6.2f per wave = %5.2f per SIMD", wps, per_wave, per_wave/wps);
  }
  printf("\n");
  fflush(stdout);
  return 0;
}

int main () {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0.5 + (i % 97)/97.0;
  double *din, *dout; unsigned long long* dcyc;
  CK(hipMalloc(&din, 8*4096)); CK(hipMalloc(&dout, 8*64*4*1024)); CK(hipMalloc(&dcyc, 8*4*1024));
  CK(hipMemcpy(din, h.data(), 8*4096, hipMemcpyHostToDevice));
  printf("cycles per wave-instruction (s_memtime), 16 independent instructions per loop pass\n");
  run<0>("v_add_f64", din, dout, dcyc);
  run<1>("v_mul_f64", din, dout, dcyc);
  run<2>("v_fma_f64", din, dout, dcyc);
  run<3>("v_cmp_lt_f64 -> vcc", din, dout, dcyc);
  run<4>("v_cmp_lt_f64 -> sgpr pair", din, dout, dcyc);
  run<5>("v_cndmask_b32", din, dout, dcyc);
  run<6>("v_max_f64", din, dout, dcyc);
  run<7>("v_rcp_f64", din, dout, dcyc);
  run<8>("v_div_scale_f64", din, dout, dcyc);
  run<9>("v_div_fixup_f64", din, dout, dcyc);
  run<10>("v_mov_b64", din, dout, dcyc);
  run<11>("s_and_b64", din, dout, dcyc);
  run<12>("v_cmp_lt_i32 -> vcc", din, dout, dcyc);
  run<13>("v_cndmask_b32 vcc, independent", din, dout, dcyc);
  run<14>("v_cndmask_b32_e64 sgpr, independent", din, dout, dcyc);
  run<15>("v_bfi_b32 independent", din, dout, dcyc);
  run<16>("2 x v_cndmask_b32_e64 (per instr)", din, dout, dcyc);
  run<17>("cmp->vcc + 2 cndmask vcc (per 1.5 instr)", din, dout, dcyc);
  run<18>("cmp->sgpr + 2 cndmask e64 (per 1.5 instr)", din, dout, dcyc);
  return 0;
}
