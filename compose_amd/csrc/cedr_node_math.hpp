// Device-side node problems of QLT (reference cedr/cedr_local_inl.hpp and
// cedr/cedr_qlt_inl.hpp), written for one lane = one (node, tracer). Compiled
// with -ffp-contract=off: every operation must round exactly like the
// reference's unfused x86-64 arithmetic because the algorithm branches on exact
// comparisons.
#ifndef COMPOSE_AMD_CEDR_NODE_MATH_HPP
#define COMPOSE_AMD_CEDR_NODE_MATH_HPP

#include <hip/hip_runtime.h>

#include <cfloat>

namespace cedr_dev {

constexpr double kEps = DBL_EPSILON;
enum : int { CONSERVE = 1, SHAPEPRESERVE = 2, CONSISTENT = 4, NONNEGATIVE = 8 };

// Whether any lane of the wave (that is active here) has p set. Host builds (tests/cpp/np_wave_check.cpp)
// see a wave of one lane.
#ifndef CEDR_NP_TRACE  // (tests/cpp/np_wave_check.cpp counts which way a problem went)
# define CEDR_NP_TRACE(code)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// (the builtin takes the predicate as a lane mask; __ballot(int) would first turn it into 0 / 1 per lane and
// compare that again: two vector instructions per use)
# define CEDR_WAVE_ANY(p) (__builtin_amdgcn_ballot_w64(p) != 0)
#else
# define CEDR_WAVE_ANY(p) (p)
#endif

// p ? a : b for three predicates that are each the result of ONE comparison. On the device the three lane masks
// are combined by scalar instructions; written as `p ? a : b` (or with & and |, which the compiler turns back into
// it) the choice is made on 0 / 1 values in vector registers: five vector instructions.
__host__ __device__ __forceinline__ bool mask_select (const bool p, const bool a, const bool b) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned long long mp = __builtin_amdgcn_ballot_w64(p);
  return __builtin_amdgcn_inverse_ballot_w64((mp & __builtin_amdgcn_ballot_w64(a)) |
                                             (~mp & __builtin_amdgcn_ballot_w64(b)));
#else
  return p ? a : b;
#endif
}

// max(|a|, |b|, |c|) of three numbers that are not NaN. -DCEDR_ASM_ABSMAX: two v_max_f64 with |.| source
// modifiers (hipcc's fmax(fabs(x), ...) quiets every operand first: three more v_max_f64 x, x, x).
__host__ __device__ __forceinline__ double absmax3 (const double a, const double b, const double c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(CEDR_ASM_ABSMAX)
  double t, r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(t) : "v"(a), "v"(b));
  asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(t), "v"(c));
  return r;
#else
  return fmax(fmax(fabs(a), fabs(b)), fabs(c));
#endif
}

// reference cedr_kokkos.hpp:136-139: comparisons, not fmin/fmax.
__host__ __device__ __forceinline__ double rmin (double a, double b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ double rmax (double a, double b) { return a > b ? a : b; }

// ------------------------------------------------------------------ node math
// The part of solve_1eq_bc_qp_2d (reference cedr_local_inl.hpp:99-165, a = (1, 1)) that runs when
// the optimum of the unconstrained problem is outside the box: the solution sits where the
// line x0 + x1 = b crosses one of the box edges.
__host__ __device__ __forceinline__
void qp_edges (const double w0, const double w1, const double b,
               const double lo0, const double lo1, const double hi0, const double hi1,
               const double y0, const double y1, const bool clip, double& x0, double& x1) {
  // Line a'x = b as base + alpha*dir, dir = (-1, 1); intersect the box edges:
  // 0 bottom (x1 = lo1), 1 right (x0 = hi0), 2 top (x1 = hi1), 3 left (x0 = lo0).
  // The reference scans the four alphas for the FIRST minimum and the FIRST
  // maximum, takes the two remaining edges in index order (m0 < m1; edges 1, 2
  // if all four alphas are equal) and keeps the one with the smaller objective,
  // m0 on a tie. Here the same choice is made without branches or indexed
  // arrays: the twelve ordered comparisons give wave masks (scalar registers),
  // everything up to the final selects is scalar mask logic.
  const double base = 0.5*b;
  const double al0 = lo1 - base;      // bottom
  const double al1 = -(hi0 - base);   // right
  const double al2 = hi1 - base;      // top
  const double al3 = -(lo0 - base);   // left
  const bool L01 = al0 < al1, L02 = al0 < al2, L03 = al0 < al3;
  const bool L10 = al1 < al0, L12 = al1 < al2, L13 = al1 < al3;
  const bool L20 = al2 < al0, L21 = al2 < al1, L23 = al2 < al3;
  const bool L30 = al3 < al0, L31 = al3 < al1, L32 = al3 < al2;
  // first minimum: smaller than everything before, not larger than anything after
  const bool mn0 = ! L10 && ! L20 && ! L30;
  const bool mn1 = L10 && ! L21 && ! L31;
  const bool mn2 = L20 && L21 && ! L32;
  const bool mn3 = L30 && L31 && L32;
  // first maximum
  const bool mx0 = ! L01 && ! L02 && ! L03;
  const bool mx1 = L01 && ! L12 && ! L13;
  const bool mx2 = L02 && L12 && ! L23;
  const bool mx3 = L03 && L13 && L23;
  const bool e0 = mn0 || mx0, e1 = mn1 || mx1, e2 = mn2 || mx2, e3 = mn3 || mx3;
  // m0: first edge that is neither; m1: second one
  const bool s00 = ! e0, s01 = e0 && ! e1;                      // else m0 == 2
  const bool s11 = ! e0 && ! e1, s12 = ! e2 && (e0 != e1);      // else m1 == 3
  (void) e3;
  const double a0 = s00 ? al0 : (s01 ? al1 : al2);
  const double a1 = s11 ? al1 : (s12 ? al2 : al3);
  double obj0, obj1;
  {
    const double t0 = base - a0, t1 = base + a0;
    const double d0 = y0 - t0, d1 = y1 - t1;
    obj0 = w0*(d0*d0) + w1*(d1*d1);
  }
  {
    const double t0 = base - a1, t1 = base + a1;
    const double d0 = y0 - t0, d1 = y1 - t1;
    obj1 = w0*(d0*d0) + w1*(d1*d1);
  }
  const bool pick0 = obj0 <= obj1;
  // chosen edge ai as masks
  const bool c0 = pick0 && s00;                                 // ai == 0 (only m0 can be 0)
  const bool c1 = pick0 ? s01 : s11;                            // ai == 1
  const bool c2 = pick0 ? (! s00 && ! s01) : (! s11 && s12);    // ai == 2
  const bool fixed1 = c0 || c2;                                 // x1 sits on its bound, else x0 does
  const double v = c0 ? lo1 : (c2 ? hi1 : (c1 ? hi0 : lo0));
  double other = b - v;
  if (clip) {
    const double lo_o = fixed1 ? lo0 : lo1, hi_o = fixed1 ? hi0 : hi1;
    other = rmin(hi_o, rmax(lo_o, other));
  }
  x0 = fixed1 ? other : v;
  x1 = fixed1 ? v : other;
}

// reference cedr_local_inl.hpp:69-165 (solve_1eq_bc_qp_2d) with a = (1, 1).
// Multiplications and divisions by the constant 1 are dropped: they are exact,
// so every remaining operation sees the same operands in the same order.
// w_i = 1/rhom_i and q_i = a_i/w_i = 1/w_i depend only on the node, not on the
// tracer, so the caller computes them once per node and passes them in.
__host__ __device__ inline void qp_2d (const double w0, const double w1, const double q0, const double q1,
                       const double b,
                       const double lo0, const double lo1, const double hi0, const double hi1,
                       const double y0, const double y1, double& x0, double& x1,
                       const bool clip, const bool early_exit_on_tol) {
  if (early_exit_on_tol) {
    // calc_r_tol (:14-18) and check_lu (:23-41)
    // max(|b|, |y0|, |y1|): for operands that are not NaN, fmax returns the value the reference's
    // chain of comparisons does (all operands are >= +0, so there is no signed-zero case)
    const double ab = fmax(fmax(fabs(b), fabs(y0)), fabs(y1));
    const double r_tol = 1e1*kEps*ab;
    double r = -b;
    r += lo0; r += lo1;
    if ( ! (fabs(r) <= r_tol)) {
      if (r > 0) { x0 = lo0; x1 = lo1; return; }
      r = -b;
      r += hi0; r += hi1;
      if ( ! (fabs(r) <= r_tol) && r < 0) { x0 = hi0; x1 = hi1; return; }
    }
  }
  {
    // The reference starts its sums at 0: 0 + q0 is q0 bit for bit unless q0 is -0, and
    // q0 = 1/(1/rhom) with rhom > 0; likewise the objectives below (w*(d*d) is never -0).
    const double qmass = q0 + q1;
    double dm = b;
    dm -= y0; dm -= y1;
    const double lambda = dm/qmass;
    x0 = y0 + lambda*q0;
    if ( ! (x0 < lo0 || x0 > hi0)) {
      x1 = y1 + lambda*q1;
      if ( ! (x1 < lo1 || x1 > hi1)) return;
    }
  }
  qp_edges(w0, w1, b, lo0, lo1, hi0, hi1, y0, y1, clip, x0, x1);
}

// reference cedr_qlt_inl.hpp:69-99 (r2l_nl_adjust_bounds)
__host__ __device__ inline void adjust_bounds (double& b0, double& b1, const double rh0, const double rh1,
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
struct KidRho { double rh, w, q; };  // rhom, 1/rhom, 1/(1/rhom) of a kid

__host__ __device__ inline void node_problem_mass (const double p_lo, const double p_orig, const double p_hi,
                                   const double Qm,
                                   const KidRho r0, double lo0, const double or0, double hi0,
                                   const KidRho r1, double lo1, const double or1, double hi1,
                                   double& x0, double& x1, const bool prefer) {
#ifdef CEDR_DIAG_NOQP  // diagnostic builds only (wrong results): the sweep without its node problems
  x0 = or0; x1 = or1; return;
#endif
  const double rh0 = r0.rh, rh1 = r1.rh;
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
  qp_2d(r0.w, r1.w, r0.q, r1.q, Qm, lo0, lo1, hi0, hi1, or0, or1, x0, x1, ! prefer, ! prefer);
}

// The branchy routine as the wave-oriented form below calls it (rarely). Inlined by default.
#ifdef CEDR_NP_COLD_NOINLINE  // (measured: a real call costs 29 spilled VGPRs around it in the kernel)
__host__ __device__ __attribute__((noinline))
#else
__host__ __device__ inline
#endif
void node_problem_mass_cold (const double p_lo, const double p_orig, const double p_hi, const double Qm,
                             const KidRho r0, const double lo0, const double or0, const double hi0,
                             const KidRho r1, const double lo1, const double or1, const double hi1,
                             double& x0, double& x1, const bool prefer) {
  node_problem_mass(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
}

// The same node problem (bit-identical results) written for how a wave executes it. The lanes of a
// wave almost never agree on which way node_problem_mass branches, so the wave runs both sides of
// nearly every `if`, and every divergent branch costs a saveexec / branch / restore sequence and a
// fetch bubble on top. Here the COMMON case -- the node's mass inside its bounds, which is every node
// of a feasible problem -- is straight-line code: the unconstrained optimum and the corner exits'
// residuals are computed unconditionally. Lanes part ways in three places only, by ordinary divergent
// branches that are skipped when no lane takes them: lanes that keep their kids' masses are done; the
// box-edge part runs for the lanes whose optimum is outside the box; and the reference's own control
// flow (node_problem_mass) runs for the lanes in a case the straight-line code does not cover (an
// infeasible node, which adjusts bounds; a degenerate box whose edge intersections do not separate).
// Predicates are combined with & and | on purpose and a select between lane masks goes through
// mask_select: this compiler keeps `a && b`, `p ? a : b` and ballots of combined predicates as 0 / 1
// values in vector registers (profiles/r03_experiments.md 11).
// (-DCEDR_NP_WAVE_UNIFORM: the earlier form, which asked the wave before each part -- a ballot -- and
// ran the part for all lanes.)
//
// The box-edge part replaces qp_edges' twelve comparisons by four, using what the reference's scan
// (cedr_local_inl.hpp:121-147: first minimum and first maximum of the four alphas, the two remaining
// edges in index order) gives whenever the minimum is al0 or al1 (al0 if equal: the FIRST minimum) and
// the maximum is al2 or al3 (al2 if equal): the remaining edges are m0 = (al0 > al1 ? 0 : 1) and
// m1 = (al2 < al3 ? 2 : 3), m0 < m1. Any box the line x0 + x1 = Qm really crosses is like that;
// otherwise: the general routine.
__host__ __device__ __forceinline__
void node_problem_mass_wave (const double p_lo, const double p_orig, const double p_hi, const double Qm,
                             const KidRho r0, const double lo0, const double or0, const double hi0,
                             const KidRho r1, const double lo1, const double or1, const double hi1,
                             double& x0, double& x1, const bool prefer) {
#ifdef CEDR_DIAG_NOQP
  x0 = or0; x1 = or1; return;
#endif
#ifdef CEDR_NP_GENERAL  // (experiments: the branchy routine everywhere)
  node_problem_mass(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
  return;
#endif
  const bool outside = Qm < p_lo || Qm > p_hi;
  const bool quick = Qm == p_orig && or0 >= lo0 && or0 <= hi0 && or1 >= lo1 && or1 <= hi1;
#ifndef CEDR_NP_WAVE_UNIFORM
  // The three places where lanes part ways -- done already; the box-edge part; the general routine -- are
  // ordinary divergent branches: lanes that do not need a part sit it out (exec mask) and the part is skipped
  // when no lane needs it (s_cbranch_execz). Asking the wave first (a ballot) so that every lane runs the
  // part costs two vector instructions per question and one more select level at the end.
  x0 = or0; x1 = or1;
  if ( ! outside & quick) {  // keeps its kids' masses
    CEDR_NP_TRACE(1);
    return;
  }
  // unconstrained optimum
  const double qmass = r0.q + r1.q;
  double dm = Qm;
  dm -= or0; dm -= or1;
  const double lambda = dm/qmass;
  const double xa = or0 + lambda*r0.q;
  const double xb = or1 + lambda*r1.q;
  const bool inside = ! (xa < lo0 || xa > hi0) && ! (xb < lo1 || xb > hi1);
  double s0 = xa, s1 = xb;
  // check_lu's corner exits (only with early_exit_on_tol = ! prefer), which qp_2d takes BEFORE it looks at the
  // optimum. The two residuals are straight-line code (nearly every lane gets as far as the second); the exits
  // themselves are rare and set their results in a branch (kept a branch by the empty asm: as selects every lane
  // would pay eight instructions for them).
  // (computed whatever `prefer` is and masked with it: under `if ( ! prefer)` the flag would live in a vector
  // register as 0 / 1 across that branch)
  const double ab = absmax3(Qm, or0, or1);
  const double r_tol = 1e1*kEps*ab;
  double r = -Qm;
  r += lo0; r += lo1;
  double r2 = -Qm;
  r2 += hi0; r2 += hi1;
  const bool out = ! (fabs(r) <= r_tol);
  const bool exl = out & (r > 0);
  const bool exh = out & ! (r > 0) & ! (fabs(r2) <= r_tol) & (r2 < 0);
  const bool corner = ! prefer & (exl | exh);
  if (corner) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("");
#endif
    s0 = exl ? lo0 : hi0;
    s1 = exl ? lo1 : hi1;
  }
  const bool done = inside | corner;
  bool general = outside;
  if ( ! done & ! outside) {
    const double base = 0.5*Qm;
    const double al0 = lo1 - base;      // bottom: x1 = lo1
    const double al1 = -(hi0 - base);   // right:  x0 = hi0
    const double al2 = hi1 - base;      // top:    x1 = hi1
    const double al3 = -(lo0 - base);   // left:   x0 = lo0
    const bool c01 = al0 > al1, c23 = al2 < al3;
    // the kept pair a_lo = c01 ? al0 : al1, a_hi = c23 ? al2 : al3 and the dropped pair a_min = c01 ? al1 : al0,
    // a_max = c23 ? al3 : al2, as one max / min each instead of two selects: the same values (none is NaN); where
    // the operands are +0 and -0 the sign may differ, which no use below can see (sums with base, squares,
    // comparisons)
    const double a_lo = fmax(al0, al1), a_hi = fmin(al2, al3);
    const double a_min = fmin(al0, al1), a_max = fmax(al2, al3);
    // The scan drops exactly these two iff the FIRST minimum of the four is the dropped lower one
    // (a_min <= a_hi; a_min <= a_lo and a_hi <= a_max hold by construction) and the FIRST maximum is the
    // dropped upper one (a_max > a_lo, strictly: a tie would make index 0 or 1 the first maximum).
    // A node sitting exactly on a bound -- common: every box-edge solution puts a kid there -- has
    // a_lo == a_hi up to rounding and still passes.
    // (bitwise on purpose, here and for fixed1: lane masks combined by scalar and / or; `||` and `?:` between
    // masks become selects on 0 / 1 vector registers)
    general = ! (a_min <= a_hi) | ! (a_max > a_lo);
    double obj0, obj1;
    {
      const double t0 = base - a_lo, t1 = base + a_lo;
      const double d0 = or0 - t0, d1 = or1 - t1;
      obj0 = r0.w*(d0*d0) + r1.w*(d1*d1);
    }
    {
      const double t0 = base - a_hi, t1 = base + a_hi;
      const double d0 = or0 - t0, d1 = or1 - t1;
      obj1 = r0.w*(d0*d0) + r1.w*(d1*d1);
    }
    const bool pick0 = obj0 <= obj1;
    const bool fixed1 = mask_select(pick0, c01, c23);  // x1 sits on its bound, else x0 does
    const double v_lo = c01 ? lo1 : hi0, v_hi = c23 ? hi1 : lo0;
    const double v = pick0 ? v_lo : v_hi;
    double other = Qm - v;
    if ( ! prefer) {
      const double lo_o = fixed1 ? lo0 : lo1, hi_o = fixed1 ? hi0 : hi1;
      other = rmin(hi_o, rmax(lo_o, other));
    }
    s0 = fixed1 ? other : v;
    s1 = fixed1 ? v : other;
  }
  if (general) {  // an infeasible node (adjusts bounds), a degenerate box: the reference's own control flow
    CEDR_NP_TRACE(0);
    node_problem_mass_cold(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, s0, s1, prefer);
  } else {
    CEDR_NP_TRACE(corner ? 2 : (inside ? 3 : 4));
  }
  x0 = s0; x1 = s1;
#else
  if ( ! CEDR_WAVE_ANY(outside || ! quick)) {  // every lane keeps its kids' masses (unperturbed data)
    x0 = or0; x1 = or1;
    CEDR_NP_TRACE(1);
    return;
  }
  // check_lu's corner exits (only with early_exit_on_tol = ! prefer)
  bool exl = false, exh = false;
  if ( ! prefer) {
    const double ab = fmax(fmax(fabs(Qm), fabs(or0)), fabs(or1));
    const double r_tol = 1e1*kEps*ab;
    double r = -Qm;
    r += lo0; r += lo1;
    double r2 = -Qm;
    r2 += hi0; r2 += hi1;
    const bool out = ! (fabs(r) <= r_tol);
    exl = out && r > 0;
    exh = out && ! (r > 0) && ! (fabs(r2) <= r_tol) && r2 < 0;
  }
  // unconstrained optimum
  const double qmass = r0.q + r1.q;
  double dm = Qm;
  dm -= or0; dm -= or1;
  const double lambda = dm/qmass;
  const double xa = or0 + lambda*r0.q;
  const double xb = or1 + lambda*r1.q;
  const bool inside = ! (xa < lo0 || xa > hi0) && ! (xb < lo1 || xb > hi1);
  const bool done = quick || exl || exh || inside;
  double e0 = 0, e1 = 0;
  bool general = outside;
  if (CEDR_WAVE_ANY( ! done && ! outside)) {
    const double base = 0.5*Qm;
    const double al0 = lo1 - base;      // bottom: x1 = lo1
    const double al1 = -(hi0 - base);   // right:  x0 = hi0
    const double al2 = hi1 - base;      // top:    x1 = hi1
    const double al3 = -(lo0 - base);   // left:   x0 = lo0
    const bool c01 = al0 > al1, c23 = al2 < al3;
    // the kept pair a_lo = c01 ? al0 : al1, a_hi = c23 ? al2 : al3 and the dropped pair a_min = c01 ? al1 : al0,
    // a_max = c23 ? al3 : al2, as one max / min each instead of two selects: the same values (none is NaN); where
    // the operands are +0 and -0 the sign may differ, which no use below can see (sums with base, squares,
    // comparisons)
    const double a_lo = fmax(al0, al1), a_hi = fmin(al2, al3);
    const double a_min = fmin(al0, al1), a_max = fmax(al2, al3);
    // The scan drops exactly these two iff the FIRST minimum of the four is the dropped lower one
    // (a_min <= a_hi; a_min <= a_lo and a_hi <= a_max hold by construction) and the FIRST maximum is the
    // dropped upper one (a_max > a_lo, strictly: a tie would make index 0 or 1 the first maximum).
    // A node sitting exactly on a bound -- common: every box-edge solution puts a kid there -- has
    // a_lo == a_hi up to rounding and still passes.
    // (bitwise on purpose, here and for fixed1: lane masks combined by scalar and / or; `||` and `?:` between
    // masks become selects on 0 / 1 vector registers)
    const bool invalid = ! (a_min <= a_hi) | ! (a_max > a_lo);
    general = general | ( ! done & invalid);
    double obj0, obj1;
    {
      const double t0 = base - a_lo, t1 = base + a_lo;
      const double d0 = or0 - t0, d1 = or1 - t1;
      obj0 = r0.w*(d0*d0) + r1.w*(d1*d1);
    }
    {
      const double t0 = base - a_hi, t1 = base + a_hi;
      const double d0 = or0 - t0, d1 = or1 - t1;
      obj1 = r0.w*(d0*d0) + r1.w*(d1*d1);
    }
    const bool pick0 = obj0 <= obj1;
    const bool fixed1 = (pick0 & c01) | ( ! pick0 & c23);  // x1 sits on its bound, else x0 does
    const double v_lo = c01 ? lo1 : hi0, v_hi = c23 ? hi1 : lo0;
    const double v = pick0 ? v_lo : v_hi;
    double other = Qm - v;
    if ( ! prefer) {
      const double lo_o = fixed1 ? lo0 : lo1, hi_o = fixed1 ? hi0 : hi1;
      other = rmin(hi_o, rmax(lo_o, other));
    }
    e0 = fixed1 ? other : v;
    e1 = fixed1 ? v : other;
  }
  if (CEDR_WAVE_ANY(general)) {
    CEDR_NP_TRACE(0);
    node_problem_mass_cold(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1, x0, x1, prefer);
    return;
  }
  CEDR_NP_TRACE(quick ? 1 : (exl || exh ? 2 : (inside ? 3 : 4)));
  // in the order in which node_problem_mass / qp_2d decide
  x0 = quick ? or0 : (exl ? lo0 : (exh ? hi0 : (inside ? xa : e0)));
  x1 = quick ? or1 : (exl ? lo1 : (exh ? hi1 : (inside ? xb : e1)));
#endif
}

// N independent node problems of one lane advanced in LOCKSTEP: the same operations on the same
// operands as N calls of node_problem_mass (bit-identical results), but written as straight-line
// code over the N problems so that their dependent chains (a dependent fp64 operation waits ~8
// cycles, an independent one issues every ~4) interleave. The rare paths -- an infeasible node,
// the corner exits of check_lu -- stay per-problem branches.
template <int N>
__device__ __forceinline__
void node_problems_mass (const double (&p_lo)[N], const double (&p_orig)[N], const double (&p_hi)[N],
                         const double (&Qm)[N],
                         const KidRho (&r0)[N], double (&lo0)[N], const double (&or0)[N], double (&hi0)[N],
                         const KidRho (&r1)[N], double (&lo1)[N], const double (&or1)[N], double (&hi1)[N],
                         double (&x0)[N], double (&x1)[N], const bool prefer) {
#ifdef CEDR_DIAG_NOQP
  for (int i = 0; i < N; ++i) { x0[i] = or0[i]; x1[i] = or1[i]; }
  return;
#endif
  bool quick[N];
  bool all_quick = true;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const bool below = Qm[i] < p_lo[i], above = Qm[i] > p_hi[i];
    quick[i] = false;
    if (below || above) {
      const double tol = 10*kEps;
      const double disc = below ? p_lo[i] - Qm[i] : Qm[i] - p_hi[i];
      if (disc > tol*(p_hi[i] - p_lo[i])) {
        if (below) adjust_bounds(lo0[i], lo1[i], r0[i].rh, r1[i].rh, Qm[i] - p_lo[i]);
        else       adjust_bounds(hi0[i], hi1[i], r0[i].rh, r1[i].rh, Qm[i] - p_hi[i]);
      }
    } else {
      quick[i] = Qm[i] == p_orig[i] && or0[i] >= lo0[i] && or0[i] <= hi0[i] && or1[i] >= lo1[i] &&
        or1[i] <= hi1[i];
    }
    all_quick = all_quick && quick[i];
    x0[i] = or0[i];
    x1[i] = or1[i];
  }
  if (all_quick) return;
  // solve_1eq_bc_qp_2d for every problem (the quick ones discard it)
  const bool clip = ! prefer, early = ! prefer;
  bool exl[N], exh[N], inside[N];
  double xa[N], xb[N];
  bool need_edges = false;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    exl[i] = exh[i] = false;
    if (early) {
      const double ab = fmax(fmax(fabs(Qm[i]), fabs(or0[i])), fabs(or1[i]));
      const double r_tol = 1e1*kEps*ab;
      double r = -Qm[i];
      r += lo0[i]; r += lo1[i];
      const bool out = ! (fabs(r) <= r_tol);
      exl[i] = out && r > 0;
      double r2 = -Qm[i];
      r2 += hi0[i]; r2 += hi1[i];
      exh[i] = out && ! (r > 0) && ! (fabs(r2) <= r_tol) && r2 < 0;
    }
    const double qmass = r0[i].q + r1[i].q;
    double dm = Qm[i];
    dm -= or0[i]; dm -= or1[i];
    const double lambda = dm/qmass;
    xa[i] = or0[i] + lambda*r0[i].q;
    xb[i] = or1[i] + lambda*r1[i].q;
    inside[i] = ! (xa[i] < lo0[i] || xa[i] > hi0[i]) && ! (xb[i] < lo1[i] || xb[i] > hi1[i]);
    need_edges = need_edges || ! (quick[i] || exl[i] || exh[i] || inside[i]);
  }
  double e0[N], e1[N];
  if (need_edges) {
#pragma unroll
    for (int i = 0; i < N; ++i)
      qp_edges(r0[i].w, r1[i].w, Qm[i], lo0[i], lo1[i], hi0[i], hi1[i], or0[i], or1[i], clip, e0[i], e1[i]);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) e0[i] = e1[i] = 0;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (quick[i]) continue;
    x0[i] = inside[i] ? xa[i] : e0[i];
    x1[i] = inside[i] ? xb[i] : e1[i];
    if (exl[i]) { x0[i] = lo0[i]; x1[i] = lo1[i]; }
    if (exh[i]) { x0[i] = hi0[i]; x1[i] = hi1[i]; }
  }
}

// reference cedr_qlt_inl.hpp:175-203 (dispatch on problem type)
__host__ __device__ inline void solve_node (const int pt, const double rh, const double p_lo,
                            const double p_orig, const double p_hi, const double Qm,
                            const KidRho r0, const double lo0, const double or0, const double hi0,
                            const KidRho r1, const double lo1, const double or1, const double hi1,
                            double& x0, double& x1, const bool prefer) {
  if ((pt & CONSISTENT) && ! (pt & SHAPEPRESERVE)) {
    node_problem_mass_wave(p_lo*rh, p_orig, p_hi*rh, Qm,
                           r0, lo0*r0.rh, or0, hi0*r0.rh, r1, lo1*r1.rh, or1, hi1*r1.rh, x0, x1, prefer);
  } else if (pt & NONNEGATIVE) {
    // solve_1eq_nonneg(2, ones, Qm, y, x, w, least_squares), cedr_local_inl.hpp:309-330
    x0 = or0;
    x1 = or1;
    if (Qm < 0) return;
    qp_2d(r0.w, r1.w, r0.q, r1.q, Qm, 0.0, 0.0, Qm, Qm, or0, or1, x0, x1, true, true);
  } else {
    node_problem_mass_wave(p_lo, p_orig, p_hi, Qm, r0, lo0, or0, hi0, r1, lo1, or1, hi1,
                           x0, x1, prefer);
  }
}

}  // namespace cedr_dev

#endif
