// cedr::local -- the scalar 1-equality solvers a caller may use on its own,
// host and device (reference cedr/cedr_local.hpp:24-60, cedr_local_inl.hpp:14-330):
//
//     min_x norm(x - y; w)   s.t.  a'x = b,  xlo <= x <= xhi,   a, w > 0
//
//   solve_1eq_bc_qp     N-D, safeguarded Newton on the dual        (:168-270)
//   solve_1eq_bc_qp_2d  2-D special case                           (:69-165)
//   caas                local ClipAndAssuredSum (1-norm, w = 1)    (:273-306)
//   solve_1eq_nonneg    x >= 0 wrapper                             (:309-330)
//
// The QLT kernels do not call these: they use the a = (1,1) specialisation in
// cedr_node_math.hpp, which shares per-node quantities between tracers. These
// general forms keep the reference's operation order, so that results are the
// same bits as the reference's (build with -ffp-contract=off, as this library is).
// Every function is `__host__ __device__` under hipcc and plain inline otherwise.
#ifndef COMPOSE_AMD_CEDR_LOCAL_HPP
#define COMPOSE_AMD_CEDR_LOCAL_HPP

#include <cmath>
#include <limits>

#include "cedr.hpp"

#if defined(__HIPCC__)
# include <hip/hip_runtime.h>
# define CEDR_LOCAL_FN __host__ __device__ inline
#else
# define CEDR_LOCAL_FN inline
#endif

namespace cedr {
namespace local {

struct Method { enum Enum { least_squares, caas }; };

namespace impl {
// comparisons, not fmin/fmax (reference cedr_kokkos.hpp:136-139)
CEDR_LOCAL_FN Real rmin (const Real a, const Real b) { return a < b ? a : b; }
CEDR_LOCAL_FN Real rmax (const Real a, const Real b) { return a > b ? a : b; }

// Residual tolerance 10 eps max(|b|, |a_i y_i|)   (:14-18)
CEDR_LOCAL_FN Real calc_r_tol (const Real b, const Real* a, const Real* y, const Int n) {
  Real ab = std::fabs(b);
  for (Int i = 0; i < n; ++i) ab = rmax(ab, std::fabs(a[i]*y[i]));
  return 1e1*std::numeric_limits<Real>::epsilon()*std::fabs(ab);
}

// a'x - b at a corner of the box; x is left at that corner.
CEDR_LOCAL_FN Real corner_residual (const Int n, const Real* a, const Real b, const Real* corner,
                                    Real* x) {
  Real r = -b;
  for (Int i = 0; i < n; ++i) {
    x[i] = corner[i];
    r += a[i]*x[i];
  }
  return r;
}

// Feasibility at the two extreme corners (:23-41): -1 infeasible, 1 a corner
// solves the problem (x holds it), 0 feasible and no corner does (x = xhi).
CEDR_LOCAL_FN Int check_lu (const Int n, const Real* a, const Real& b, const Real* xlo,
                            const Real* xhi, const Real& r_tol, Real* x) {
  Real r = corner_residual(n, a, b, xlo, x);
  if (std::fabs(r) <= r_tol) return 1;
  if (r > 0) return -1;
  r = corner_residual(n, a, b, xhi, x);
  if (std::fabs(r) <= r_tol) return 1;
  if (r < 0) return -1;
  return 0;
}

// x(lambda), r = a'x - b and dr/dlambda (:44-64)
CEDR_LOCAL_FN void calc_r (const Int n, const Real* w, const Real* a, const Real b, const Real* xlo,
                           const Real* xhi, const Real* y, const Real& lambda, Real* x, Real& r,
                           Real& r_lambda) {
  r = 0;
  r_lambda = 0;
  for (Int i = 0; i < n; ++i) {
    const Real q = a[i]/w[i];
    const Real trial = y[i] + lambda*q;
    if (trial < xlo[i]) x[i] = xlo[i];
    else if (trial > xhi[i]) x[i] = xhi[i];
    else {
      x[i] = trial;
      r_lambda += a[i]*q;
    }
    r += a[i]*x[i];
  }
  r -= b;
}
}  // namespace impl

// Returns -1 if infeasible (only with early_exit_on_tol), else 1.
CEDR_LOCAL_FN Int solve_1eq_bc_qp_2d (const Real* w, const Real* a, const Real b, const Real* xlo,
                                      const Real* xhi, const Real* y, Real* x,
                                      const bool clip = true, const bool early_exit_on_tol = true) {
  if (early_exit_on_tol) {
    const Real r_tol = impl::calc_r_tol(b, a, y, 2);
    if (impl::check_lu(2, a, b, xlo, xhi, r_tol, x) == -1) return -1;
  }
  {  // the optimum of the problem without bounds, if it is inside the box
    Real qmass = 0, dm = b;
    for (Int i = 0; i < 2; ++i) {
      const Real qi = a[i]/w[i];
      qmass += a[i]*qi;
      dm -= a[i]*y[i];
    }
    const Real lambda = dm/qmass;
    bool inside = true;
    for (Int i = 0; i < 2 && inside; ++i) {
      x[i] = y[i] + lambda*(a[i]/w[i]);
      inside = ! (x[i] < xlo[i] || x[i] > xhi[i]);
    }
    if (inside) return 1;
  }
  // The constraint is the line base + alpha*dir; alpha at its crossings of the
  // box edges 0 bottom, 1 right, 2 top, 3 left. The solution sits at one of the two
  // middle crossings.
  const Real base[2] = {0.5*b/a[0], 0.5*b/a[1]};
  const Real dir[2] = {-a[1], a[0]};
  const Real alpha[4] = {(xlo[1] - base[1])/dir[1], (xhi[0] - base[0])/dir[0],
                         (xhi[1] - base[1])/dir[1], (xlo[0] - base[0])/dir[0]};
  Int imin = 0, imax = 0;  // first minimum, first maximum
  for (Int i = 1; i < 4; ++i) {
    if (alpha[i] < alpha[imin]) imin = i;
    if (alpha[i] > alpha[imax]) imax = i;
  }
  Int mid[2] = {0, 0}, nmid = 0;
  for (Int i = 0; i < 4 && nmid < 2; ++i)
    if (i != imin && i != imax) mid[nmid++] = i;
  Real obj[2];
  for (Int j = 0; j < 2; ++j) {
    Real o = 0;
    for (Int i = 0; i < 2; ++i) {
      x[i] = base[i] + alpha[mid[j]]*dir[i];
      const Real d = y[i] - x[i];
      o += w[i]*(d*d);
    }
    obj[j] = o;
  }
  const Int edge = mid[obj[0] <= obj[1] ? 0 : 1];
  // the coordinate on its bound, then the other one from the constraint
  const Int i0 = (edge == 0 || edge == 2) ? 1 : 0, i1 = 1 - i0;
  x[i0] = (edge == 0 || edge == 3) ? xlo[i0] : xhi[i0];
  x[i1] = (b - a[i0]*x[i0])/a[i1];
  if (clip) x[i1] = impl::rmin(xhi[i1], impl::rmax(xlo[i1], x[i1]));
  return 1;
}

// Returns 0 on success with x == y, 1 on success with x != y, -1 if infeasible,
// -2 if max_its was hit.
CEDR_LOCAL_FN Int solve_1eq_bc_qp (const Int n, const Real* w, const Real* a, const Real b,
                                   const Real* xlo, const Real* xhi, const Real* y, Real* x,
                                   const Int max_its = 100) {
  const Real r_tol = impl::calc_r_tol(b, a, y, n);
  Int info = impl::check_lu(n, a, b, xlo, xhi, r_tol, x);
  if (info != 0) return info;
  for (Int i = 0; i < n; ++i)
    if (x[i] != y[i]) {
      info = 1;
      x[i] = y[i];
    }
  // Bracket of the multiplier: every coordinate is clipped outside it.
  Real lamlo = 0, lamhi = 0;
  for (Int i = 0; i < n; ++i) {
    const Real rq = w[i]/a[i];
    const Real lo_i = rq*(xlo[i] - y[i]), hi_i = rq*(xhi[i] - y[i]);
    lamlo = i == 0 ? lo_i : impl::rmin(lamlo, lo_i);
    lamhi = i == 0 ? hi_i : impl::rmax(lamhi, hi_i);
  }
  const Real lamlo_feas = lamlo, lamhi_feas = lamhi;
  Real lambda = (lamlo <= 0 && lamhi >= 0) ? 0 : lamlo;
  const Real wall_dist = 1e-3;
  bool prev_step_bisect = false;
  Int nbisect = 0;
  info = -2;
  for (Int it = 0; it < max_its; ++it) {
    Real r, r_lambda;
    impl::calc_r(n, w, a, b, xlo, xhi, y, lambda, x, r, r_lambda);
    if (std::fabs(r) <= r_tol) { info = 1; break; }
    if (nbisect > 64) {
      // the bracket has collapsed: infeasible if it still touches its initial end
      info = (lamhi == lamhi_feas || lamlo == lamlo_feas) ? -1 : 1;
      break;
    }
    if (r > 0) lamhi = lambda; else lamlo = lambda;
    if (r_lambda != 0) lambda -= r/r_lambda;  // Newton
    else lambda = lamlo;                      // forces the bisection below
    // Safeguard; the wall distance applies to every other potential bisection only.
    const Real D = prev_step_bisect ? 0 : wall_dist*(lamhi - lamlo);
    if (lambda - lamlo < D || lamhi - lambda < D) {
      lambda = 0.5*(lamlo + lamhi);
      ++nbisect;
      prev_step_bisect = true;
    } else {
      prev_step_bisect = false;
    }
  }
  return info;
}

// Clip, then hand the mass discrepancy out in proportion to the remaining capacity.
CEDR_LOCAL_FN void caas (const Int n, const Real* a, const Real b, const Real* xlo, const Real* xhi,
                         const Real* y, Real* x, const bool clip = true) {
  Real dm = b;
  for (Int i = 0; i < n; ++i) {
    x[i] = impl::rmax(xlo[i], impl::rmin(xhi[i], y[i]));
    dm -= a[i]*x[i];
  }
  if (dm == 0) return;
  if (dm > 0) {
    Real fac = 0;
    for (Int i = 0; i < n; ++i) fac += a[i]*(xhi[i] - x[i]);
    if (fac > 0) {
      fac = dm/fac;
      for (Int i = 0; i < n; ++i) x[i] += fac*(xhi[i] - x[i]);
    }
  } else if (dm < 0) {
    Real fac = 0;
    for (Int i = 0; i < n; ++i) fac += a[i]*(x[i] - xlo[i]);
    if (fac > 0) {
      fac = dm/fac;
      for (Int i = 0; i < n; ++i) x[i] += fac*(x[i] - xlo[i]);
    }
  }
  if (clip)
    for (Int i = 0; i < n; ++i) x[i] = impl::rmax(xlo[i], impl::rmin(xhi[i], x[i]));
}

// x >= 0 (n <= 16): upper bounds b/a_i, i.e. one entry may take all of the mass.
CEDR_LOCAL_FN Int solve_1eq_nonneg (const Int n, const Real* a, const Real b, const Real* y, Real* x,
                                    const Real* w, const Method::Enum method) {
  if (n > 16) return -2;
  if (b < 0) return -1;
  Real zero[16], xhi[16];
  for (Int i = 0; i < n; ++i) {
    zero[i] = 0;
    xhi[i] = b/a[i];
  }
  if (method == Method::caas) {
    caas(n, a, b, zero, xhi, y, x);
    return 1;
  }
  return n == 2 ? solve_1eq_bc_qp_2d(w, a, b, zero, xhi, y, x) :
    solve_1eq_bc_qp(n, w, a, b, zero, xhi, y, x);
}

}  // namespace local
}  // namespace cedr

#endif
