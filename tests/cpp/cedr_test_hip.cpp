// C++ drop-in check: a caller written the way the reference's own harness is
// (cedr_test.cpp + cedr_test_randomized_inl.hpp:12-65), using the C++ classes with
// the reference's names -- cedr::qlt::QLT<>, cedr::caas::CAAS<>,
// cedr::tree::make_tree_over_1d_mesh, CDR::DeviceOp copied by value into the
// caller's own device kernels -- against libcedr_hip.so. Exit code 0 and "PASS"
// if the property checks of cedr_test_randomized.cpp:293-418 hold.
//
//   cedr_test_hip [-nc ncells] [--imbalanced] [--prefer-mass-con]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#include "cedr_bfb_tree_allreduce.hpp"
#include "cedr_caas.hpp"
#include "cedr_qlt.hpp"

using cedr::Int;
using cedr::Real;
typedef cedr::ProblemType PT;

#define HIP_OK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s:%d\n", __FILE__, __LINE__); std::exit(2); } } while (0)

// ---- the caller's own device kernels: one thread per (cell) or (tracer, cell)
template <typename Op> __global__ void k_set_rhom (Op op, int n, const Real* rhom) {
  const int i = blockIdx.x*blockDim.x + threadIdx.x;
  if (i < n) op.set_rhom(i, 0, rhom[i]);
}
template <typename Op> __global__ void k_set_Qm (Op op, int n, int nt, const Real* Qm, const Real* lo,
                                                 const Real* hi, const Real* prev) {
  const int j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= n*nt) return;
  const int ti = j / n, i = j % n;
  op.set_Qm(i, ti, Qm[j], lo[j], hi[j], prev[j]);
}
template <typename Op> __global__ void k_get_Qm (Op op, int n, int nt, Real* Qm) {
  const int j = blockIdx.x*blockDim.x + threadIdx.x;
  if (j >= n*nt) return;
  Qm[j] = op.get_Qm(j % n, j / n);
}

struct Tracer { int pt, perturb; bool local_should_hold; };

// Uniform in [0,1) on the 2^-31 grid like the reference's rand()/(RAND_MAX+1.0)
// (cedr_test_randomized.cpp:53), from a private generator: std::rand's state is
// shared with the HIP runtime's threads, which made the inputs differ run to run.
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static Real urand () {
  g_rng = g_rng*6364136223846793005ull + 1442695040888963407ull;
  return (Real) (g_rng >> 33) / 2147483648.0;
}

struct Values {
  int n, nt;
  std::vector<Real> rhom, lo, Qm, hi, prev;  // [tracer][cell]
  Values (int n_, int nt_) : n(n_), nt(nt_), rhom(n_), lo((size_t) n_*nt_), Qm(lo), hi(lo), prev(lo) {}
};

static void permute (Real* q, int n) {
  std::vector<int> p(n);
  for (int i = 0; i < n; ++i) p[i] = i;
  for (int i = 0; i < n; ++i) std::swap(p[(int) (urand()*n)], p[(int) (urand()*n)]);
  std::vector<Real> o(q, q + n);
  for (int i = 0; i < n; ++i) q[i] = o[p[i]];
}

// Inputs and perturbations in the manner of cedr_test_randomized.cpp:55-197.
static void generate (const std::vector<Tracer>& ts, Values& v) {
  const int n = v.n;
  for (int i = 0; i < n; ++i) v.rhom[i] = 0.5*(1 + urand());
  for (size_t t = 0; t < ts.size(); ++t) {
    Real* lo = &v.lo[t*n], * Qm = &v.Qm[t*n], * hi = &v.hi[t*n], * prev = &v.prev[t*n];
    const bool nn = ts[t].pt & PT::nonnegative;
    for (int i = 0; i < n; ++i) {
      if (nn) {
        Qm[i] = ts[t].perturb == 0 ? urand() : ((i % 2 == 0) ? 0.75 : -0.75) + urand();
        lo[i] = 0; hi[i] = 10;
      } else {
        const Real qmin = -0.75 + urand(), qmax = qmin + urand(), q = qmin + (qmax - qmin)*urand();
        lo[i] = qmin*v.rhom[i]; hi[i] = qmax*v.rhom[i];
        Qm[i] = std::max(lo[i], std::min(hi[i], q*v.rhom[i]));
      }
      prev[i] = Qm[i];
    }
    const int p = ts[t].perturb;
    if (p == 1) permute(Qm, n);
    if (p >= 2) {
      const bool safety = p >= 4, cm = ! (ts[t].pt & PT::conserve);
      const Real alpha = (p % 2 == 0) ? 0.5 : 1 - n*std::numeric_limits<Real>::epsilon();
      Real srho = 0, sQ = 0, smax = 0, qsafe = hi[0]/v.rhom[0];
      for (int i = 0; i < n; ++i) {
        srho += v.rhom[i]; sQ += Qm[i]; smax += hi[i];
        qsafe = std::max(qsafe, hi[i]/v.rhom[i]);
      }
      const Real msafe = qsafe*srho;
      const Real dQ = safety ? ((smax - sQ) + alpha*(msafe - smax))/n : alpha*(smax - sQ)/n;
      for (int i = 0; i < n; ++i) Qm[i] += dQ;
      permute(Qm, n);
      const Real relax = 0.9;
      const Real dprev = cm ? dQ : (safety ? ((smax - sQ) + relax*alpha*(msafe - smax))/n
                                           : relax*alpha*(smax - sQ)/n);
      for (int i = 0; i < n; ++i) prev[i] += dprev;
    }
  }
}

// Property checks in the manner of cedr_test_randomized.cpp:293-418.
static int check (const char* name, const std::vector<Tracer>& ts, const Values& v,
                  const std::vector<Real>& out, bool prefer) {
  const Real eps = std::numeric_limits<Real>::epsilon();
  const Real lv_tol = prefer ? 100*eps : 0, safety_tol = prefer ? 100*eps : 3*eps;
  int nerr = 0;
  const int n = v.n;
  for (size_t t = 0; t < ts.size(); ++t) {
    const Real* lo = &v.lo[t*n], * hi = &v.hi[t*n], * prev = &v.prev[t*n], * Qm = &out[t*n];
    const bool nn = ts[t].pt & PT::nonnegative, safe_only = ! ts[t].local_should_hold;
    Real qmin = 1e3, qmax = -1e3;
    // The reference sums the masses in double; at ~1000 cells that sum's own roundoff reaches its
    // 100 eps tolerance for about one seed in three (oracle included), so sum in extended precision.
    long double m0 = 0, m1 = 0, den = 0;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
      const bool lv = nn ? Qm[i] < 0 : (Qm[i] < lo[i] - lv_tol || Qm[i] > hi[i] + lv_tol);
      if ( ! safe_only && lv) {
        ++bad;
        std::printf("  %s tracer %zu cell %d: local bound violated: %.17g not in [%.17g, %.17g]\n", name, t, i,
                    Qm[i], nn ? 0.0 : lo[i], nn ? 1e300 : hi[i]);
      }
      if (ts[t].perturb == 0 && Qm[i] != prev[i]) {
        ++bad;
        std::printf("  %s tracer %zu cell %d: changed %.17g -> %.17g\n", name, t, i, prev[i], Qm[i]);
      }
      m0 += prev[i]; m1 += Qm[i]; den += std::abs(prev[i]);
      qmin = std::min(qmin, lo[i]/v.rhom[i]); qmax = std::max(qmax, hi[i]/v.rhom[i]);
    }
    if (safe_only) {
      if (nn) qmin = 0;
      const Real delta = (qmax - qmin)*safety_tol;
      for (int i = 0; i < n; ++i)
        if (nn ? Qm[i] < -3*eps : (Qm[i] < qmin*v.rhom[i] - delta || Qm[i] > qmax*v.rhom[i] + delta)) {
          ++bad;
          std::printf("  %s tracer %zu cell %d: safety bound violated: %.17g\n", name, t, i, Qm[i]);
        }
    }
    const Real rd = (Real) (std::fabs(m1 - m0)/den);
    if ( ! (rd <= 100*eps)) {
      ++bad;
      std::printf("  %s tracer %zu: mass %.17g -> %.17g, relative error %.3e\n", name, t, (Real) m0,
                  (Real) m1, rd);
    }
    if (bad) std::printf("FAIL %s: tracer %zu (type %d, perturbation %d): %d violations\n", name, t,
                         ts[t].pt, ts[t].perturb, bad);
    nerr += bad;
  }
  return nerr;
}

// bind_arrays of the two CDRs: QLT writes to a separate output array, CAAS limits Qm in place.
static void bind (cedr::qlt::QLT<>& q, Real* const* d) { q.bind_arrays(d[0], d[1], d[2], d[3], d[4], d[5]); }
static void bind (cedr::caas::CAAS<>& c, Real* const* d) { c.bind_arrays(d[1], d[2], d[3], d[4]); }
static const Real* result_of (cedr::qlt::QLT<>&, Real* const* d) { return d[5]; }
static const Real* result_of (cedr::caas::CAAS<>&, Real* const* d) { return d[1]; }
// The bulk pack / unpack kernels (QLT only: CAAS's layout coincides with the caller's arrays, its bulk
// form is a plain copy). Returns false if the CDR has none.
static bool run_bulk (cedr::qlt::QLT<>& q, Real* const* d, Real* out) {
  q.bind_arrays(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  q.set_rhom_bulk(d[0]);
  q.set_Qm_bulk(d[1], d[2], d[3], d[4]);
  q.run();
  q.get_Qm_bulk(out);
  return true;
}
static bool run_bulk (cedr::caas::CAAS<>&, Real* const*, Real*) { return false; }

template <typename CDRT>
static int drive (const char* name, CDRT& cdr, const std::vector<Tracer>& ts, int n, bool external_memory,
                  bool prefer) {
  const int nt = (int) ts.size();
  for (const auto& t : ts) cdr.declare_tracer(t.pt, 0);
  cdr.end_tracer_declarations();
  Real* b1 = nullptr, * b2 = nullptr;
  if (external_memory) {
    size_t s1, s2;
    cdr.get_buffers_sizes(s1, s2);
    HIP_OK(hipMalloc((void**) &b1, s1*sizeof(Real)));
    HIP_OK(hipMalloc((void**) &b2, s2*sizeof(Real)));
    cdr.set_buffers(b1, b2);
  }
  cdr.finish_setup();
  Values v(n, nt);
  generate(ts, v);
  Real* d[6];
  const size_t N = (size_t) n*nt*sizeof(Real);
  HIP_OK(hipMalloc((void**) &d[0], n*sizeof(Real)));
  for (int k = 1; k < 6; ++k) HIP_OK(hipMalloc((void**) &d[k], N));
  HIP_OK(hipMemcpy(d[0], v.rhom.data(), n*sizeof(Real), hipMemcpyHostToDevice));
  const Real* src[] = {v.Qm.data(), v.lo.data(), v.hi.data(), v.prev.data()};
  for (int k = 0; k < 4; ++k) HIP_OK(hipMemcpy(d[k+1], src[k], N, hipMemcpyHostToDevice));
  // As the reference's callers do (cedr_test_randomized_inl.hpp:27-28): hold the CDR through its base,
  // ask THAT for the DeviceOp (the virtual of cedr_cdr.hpp:104) and cast to the concrete type; by value
  // into the kernels below.
  cedr::CDR& base = cdr;
  const auto op = static_cast<const typename CDRT::DeviceOp&>(base.get_device_op());
  static_assert(std::is_trivially_copyable<typename CDRT::DeviceOp>::value, "DeviceOp is copied into kernels");
  const int B = 256;
  k_set_rhom<<<(n + B - 1)/B, B>>>(op, n, d[0]);
  for (int trial = 0; trial < 2; ++trial) {  // the second run repeats the first
    k_set_Qm<<<(n*nt + B - 1)/B, B>>>(op, n, nt, d[1], d[2], d[3], d[4]);
    cdr.run();
  }
  k_get_Qm<<<(n*nt + B - 1)/B, B>>>(op, n, nt, d[5]);
  HIP_OK(hipDeviceSynchronize());
  std::vector<Real> out((size_t) n*nt);
  HIP_OK(hipMemcpy(out.data(), d[5], N, hipMemcpyDeviceToHost));
  int nzc = 0;
  {  // the bulk pack / unpack path must reproduce the inline-DeviceOp result bit for bit
    Real* ob = nullptr;
    HIP_OK(hipMalloc((void**) &ob, N));
    HIP_OK(hipMemset(ob, 0, N));
    if (run_bulk(cdr, d, ob)) {
      HIP_OK(hipDeviceSynchronize());
      std::vector<Real> bk((size_t) n*nt);
      HIP_OK(hipMemcpy(bk.data(), ob, N, hipMemcpyDeviceToHost));
      int nb = 0;
      for (size_t i = 0; i < bk.size(); ++i) if (std::memcmp(&bk[i], &out[i], sizeof(Real))) ++nb;
      if (nb) std::printf("FAIL %s: the bulk path differs from the DeviceOp path in %d values\n", name, nb);
      nzc += nb;
    }
    (void) hipFree(ob);
  }
  {  // zero-copy mode on the same arrays must reproduce the DeviceOp-driven result exactly
    HIP_OK(hipMemset(d[5], 0, N));
    bind(cdr, d);
    cdr.run();
    HIP_OK(hipDeviceSynchronize());
    std::vector<Real> zc((size_t) n*nt);
    HIP_OK(hipMemcpy(zc.data(), result_of(cdr, d), N, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < zc.size(); ++i) if (zc[i] != out[i]) ++nzc;
    if (nzc) std::printf("FAIL %s: bind_arrays differs from the DeviceOp path in %d values\n", name, nzc);
  }
  for (int k = 0; k < 6; ++k) (void) hipFree(d[k]);
  if (const char* dump = std::getenv("CEDR_TEST_DUMP")) {  // debugging aid: raw results of this run
    const std::string fn = std::string(dump) + "." + name;
    if (FILE* fp = std::fopen(fn.c_str(), "wb")) {
      std::fwrite(out.data(), sizeof(Real), out.size(), fp);
      std::fclose(fp);
    }
  }
  const int nerr = check(name, ts, v, out, prefer) + nzc;
  if (b1) { (void) hipFree(b1); (void) hipFree(b2); }
  return nerr;
}

// THE PRODUCTION PAIRING from C++: CAAS whose UserAllReducer hands its per-group partial sums to a BfbTreeAllReducer.
// (a) as a host model writes it around the reference (cedr_caas.hpp:27-49): its own UserAllReducer subclass calling
// allreduce(send, recv, /*transpose*/ true); (b) with this build's caas::BfbUserAllReducer, which a CAAS fuses with the
// first stage of the reducer's tree. Same inputs through the DeviceOp kernels: the masses must agree bit for bit and pass
// the harness's checks.
struct HandWrittenReducer : public cedr::caas::CAAS<>::UserAllReducer {
  std::shared_ptr<cedr::BfbTreeAllReducer<> > bfb; int nacc;
  HandWrittenReducer (const std::shared_ptr<cedr::BfbTreeAllReducer<> >& b, int n) : bfb(b), nacc(n) {}
  int operator() (const cedr::mpi::Parallel&, Real* sendbuf, Real* rcvbuf, int, int, MPI_Op) const override {
    bfb->allreduce(sendbuf, rcvbuf, true);
    return 0;
  }
  int n_accum_in_place () const override { return nacc; }
};

static int pairing (const cedr::mpi::Parallel::Ptr& p, const std::vector<Tracer>& ts, const int ngroups, const int nacc) {
  const int n = ngroups*nacc, nt = (int) ts.size();
  Values v(n, nt);
  generate(ts, v);
  const size_t N = (size_t) n*nt*sizeof(Real);
  Real* d[6];
  HIP_OK(hipMalloc((void**) &d[0], n*sizeof(Real)));
  for (int k = 1; k < 6; ++k) HIP_OK(hipMalloc((void**) &d[k], N));
  HIP_OK(hipMemcpy(d[0], v.rhom.data(), n*sizeof(Real), hipMemcpyHostToDevice));
  const Real* src[] = {v.Qm.data(), v.lo.data(), v.hi.data(), v.prev.data()};
  for (int k = 0; k < 4; ++k) HIP_OK(hipMemcpy(d[k+1], src[k], N, hipMemcpyHostToDevice));
  std::vector<Real> out[2];
  bool fused[2] = {false, false};
  for (int form = 0; form < 2; ++form) {
    auto bfb = std::make_shared<cedr::BfbTreeAllReducer<> >(p, cedr::tree::make_tree_over_1d_mesh(p, ngroups), ngroups, 4*nt);
    cedr::caas::CAAS<>::UserAllReducer::Ptr red;
    if (form == 0) red = std::make_shared<HandWrittenReducer>(bfb, nacc);
    else red = std::make_shared<cedr::caas::BfbUserAllReducer>(bfb, nacc);
    cedr::caas::CAAS<> caas(p, n, red);
    for (const auto& t : ts) caas.declare_tracer(t.pt, 0);
    caas.end_tracer_declarations();
    caas.finish_setup();
    fused[form] = caas.fused_with_bfb_reducer();
    const auto op = caas.get_device_op();
    const int B = 256;
    k_set_rhom<<<(n + B - 1)/B, B>>>(op, n, d[0]);
    k_set_Qm<<<(n*nt + B - 1)/B, B>>>(op, n, nt, d[1], d[2], d[3], d[4]);
    caas.run();
    k_get_Qm<<<(n*nt + B - 1)/B, B>>>(op, n, nt, d[5]);
    HIP_OK(hipDeviceSynchronize());
    out[form].resize((size_t) n*nt);
    HIP_OK(hipMemcpy(out[form].data(), d[5], N, hipMemcpyDeviceToHost));
  }
  for (int k = 0; k < 6; ++k) (void) hipFree(d[k]);
  int nerr = 0;
  if (fused[0]) { std::printf("FAIL pairing: a foreign UserAllReducer must not be fused\n"); ++nerr; }
  size_t nd = 0;
  for (size_t i = 0; i < out[0].size(); ++i) if (std::memcmp(&out[0][i], &out[1][i], sizeof(Real))) ++nd;
  if (nd) { std::printf("FAIL pairing: caas::BfbUserAllReducer (fused: %d) differs from the hand-written reducer in %zu values\n",
                        (int) fused[1], nd); ++nerr; }
  return nerr + check("CAAS+BFB", ts, v, out[1], false);
}

int main (int argc, char** argv) {
  int ncells = 777;
  bool imbalanced = false, prefer = false;
  for (int i = 1; i < argc; ++i) {
    if ( ! std::strcmp(argv[i], "-nc") && i + 1 < argc) ncells = std::atoi(argv[++i]);
    else if ( ! std::strcmp(argv[i], "--imbalanced")) imbalanced = true;
    else if ( ! std::strcmp(argv[i], "--prefer-mass-con")) prefer = true;
  }
  int nerr = 0;
  try {
    auto p = cedr::mpi::make_parallel();
    // cedr_test_randomized.cpp:26-51: 6 perturbations x 6 problem types
    const int pts[] = { PT::conserve | PT::shapepreserve | PT::consistent, PT::shapepreserve,
                        PT::conserve | PT::consistent, PT::consistent, PT::nonnegative,
                        PT::nonnegative | PT::conserve };
    std::vector<Tracer> ts, cts;
    for (int perturb = 0; perturb < 6; ++perturb)
      for (int pt : pts) {
        const bool lsh = perturb < 4 && ((pt & PT::shapepreserve) || (pt & PT::nonnegative));
        ts.push_back(Tracer{pt, perturb, lsh});
        if ((pt & PT::shapepreserve) && lsh) cts.push_back(Tracer{pt, perturb, lsh});  // cedr_caas.cpp:338-351
      }
    {
      cedr::tree::Node::Ptr tree = cedr::tree::make_tree_over_1d_mesh(p, ncells, imbalanced);
      cedr::CDR::Options o;
      o.prefer_numerical_mass_conservation_to_numerical_bounds = prefer;
      cedr::qlt::QLT<> qlt(p, ncells, tree, o);
      tree = nullptr;  // the caller may drop its tree (cedr_tree.cpp:478)
      if (qlt.nlclcells() != ncells) { std::printf("FAIL nlclcells\n"); ++nerr; }
      std::vector<cedr::Long> gcis;
      qlt.get_owned_glblcells(gcis);
      for (int i = 0; i < ncells; ++i) if (qlt.gci2lci((Int) gcis[i]) != i) ++nerr;
      nerr += drive("QLT", qlt, ts, ncells, imbalanced, prefer);
      for (size_t i = 0; i < ts.size(); ++i) {  // canonical types, cedr_qlt.cpp:707-716
        const int pt = qlt.get_problem_type((Int) i);
        if (pt != ((ts[i].pt | PT::consistent) & ~PT::nonnegative) &&
            pt != ((ts[i].pt | PT::nonnegative) & ~PT::consistent)) ++nerr;
      }
    }
    {
      cedr::caas::CAAS<> caas(p, ncells);
      nerr += drive("CAAS", caas, cts, ncells, imbalanced, false);
    }
    nerr += pairing(p, cts, std::max(ncells, 2), 4);   // small trees: the two launches; from a few thousand groups on: fused
    nerr += pairing(p, cts, 3000 + ncells, 4);
    bool threw = false;  // error convention: std::logic_error (cedr_util.hpp:70-77)
    try { cedr::caas::CAAS<> bad(p, 0); } catch (const std::logic_error&) { threw = true; }
    if ( ! threw) { std::printf("FAIL: CAAS with 0 cells did not throw\n"); ++nerr; }
  } catch (const std::exception& e) {
    std::printf("exception: %s\n", e.what());
    return 2;
  }
  std::printf(nerr ? "FAIL\n" : "PASS\n");
  return nerr ? 1 : 0;
}
