// Host-side check (no GPU): cedr_dev::node_problem_mass_wave -- the straight-line form of the node
// problem the root->leaves kernels run -- against cedr_dev::node_problem_mass, the branchy form that is
// compared bitwise with the oracle elsewhere (tests/test_gpu_hand_vectors.py). Compiled by hipcc for the
// HOST, where a "wave" is one lane: every problem either takes the straight-line code, whose result must
// equal the general routine's bit for bit, or is handed to the general routine (counted). Inputs: random
// feasible and infeasible nodes, kids outside their bounds, exact corners, degenerate boxes (lo == hi),
// equal edge intersections, zero masses, huge and tiny scales.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>

static long g_path[5];  // 0 handed to the general routine, 1 quick, 2 corner exit, 3 interior, 4 box edge
#if ! defined(__HIP_DEVICE_COMPILE__)
# define CEDR_NP_TRACE(code) ++g_path[code]
#endif
#include "cedr_node_math.hpp"

using namespace cedr_dev;

static bool same (double a, double b) { return std::memcmp(&a, &b, sizeof(a)) == 0; }

int main (int argc, char** argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 4000000;
  std::mt19937_64 g(20261004);
  std::uniform_real_distribution<double> U(0, 1);
  long bad = 0, nfast = 0, nquick = 0, ntot = 0;
  for (long it = 0; it < n; ++it) {
    const int kind = it % 16;
    const double scale = (kind == 13) ? 1e-200 : (kind == 14 ? 1e150 : 1.0);
    double rh0 = 0.5*(1 + U(g)), rh1 = 0.5*(1 + U(g));
    if (kind == 12) rh1 = rh0;
    double qlo0 = -0.75 + U(g), qlo1 = -0.75 + U(g);
    double qhi0 = qlo0 + U(g), qhi1 = qlo1 + U(g);
    if (kind == 9) { qhi0 = qlo0; }                    // degenerate box in x0
    if (kind == 10) { qhi0 = qlo0; qhi1 = qlo1; }      // a point
    if (kind == 11) { qlo1 = qlo0; qhi1 = qhi0; }      // equal mixing-ratio bounds
    double lo0 = qlo0*rh0*scale, hi0 = qhi0*rh0*scale, lo1 = qlo1*rh1*scale, hi1 = qhi1*rh1*scale;
    double or0 = (qlo0 + (qhi0 - qlo0)*U(g))*rh0*scale, or1 = (qlo1 + (qhi1 - qlo1)*U(g))*rh1*scale;
    if (kind == 1 || kind == 2) or0 = (qlo0 - 0.5 + 2*U(g))*rh0*scale;  // kid outside its bounds
    if (kind == 2) or1 = (qlo1 - 0.5 + 2*U(g))*rh1*scale;
    const double p_lo = lo0 + lo1, p_hi = hi0 + hi1, p_orig = or0 + or1;
    double Qm = p_lo + (p_hi - p_lo)*U(g);
    if (kind == 3) Qm = p_orig;                         // quick exit candidates
    if (kind == 4) Qm = p_lo;                           // exact corners
    if (kind == 5) Qm = p_hi;
    if (kind == 6) Qm = p_lo - (p_hi - p_lo)*U(g);      // infeasible below
    if (kind == 7) Qm = p_hi + (p_hi - p_lo)*U(g);      // infeasible above
    if (kind == 8) Qm = p_lo + (p_hi - p_lo)*(U(g) < 0.5 ? 1e-17 : 1 - 1e-17);
    if (kind == 15) { lo0 = 0; lo1 = 0; hi0 = Qm; hi1 = Qm; }  // the nonnegative problem's box
    const double w0 = 1.0/rh0, w1 = 1.0/rh1;
    const KidRho k0{rh0, w0, 1.0/w0}, k1{rh1, w1, 1.0/w1};
    for (int prefer = 0; prefer < 2; ++prefer) {
      double a0, a1, b0, b1;
      node_problem_mass(p_lo, p_orig, p_hi, Qm, k0, lo0, or0, hi0, k1, lo1, or1, hi1, a0, a1, prefer != 0);
      node_problem_mass_wave(p_lo, p_orig, p_hi, Qm, k0, lo0, or0, hi0, k1, lo1, or1, hi1, b0, b1, prefer != 0);
      ++ntot;
      if ( ! same(a0, b0) || ! same(a1, b1)) {
        if (bad < 10)
          std::printf("MISMATCH kind %d prefer %d: general (%.17g, %.17g) wave (%.17g, %.17g)\n", kind, prefer,
                      a0, a1, b0, b1);
        ++bad;
      }
    }
  }
  std::printf("%ld problems, %ld mismatches; straight-line code: quick %ld, corner exit %ld, interior %ld, "
              "box edge %ld; handed to the general routine %ld\n", ntot, bad, g_path[1], g_path[2], g_path[3],
              g_path[4], g_path[0]);
  (void) nfast; (void) nquick;
  // the check means nothing unless the straight-line code took most of the problems, every kind of them
  const bool covered = g_path[1] > 1000 && g_path[2] > 1000 && g_path[3] > 1000 && g_path[4] > 1000 &&
    g_path[0] < ntot/2;
  if ( ! covered) std::printf("FAIL: the straight-line paths were not all exercised\n");
  return (bad || ! covered) ? 1 : 0;
}
