// Device-callable pack/unpack of one (cell, tracer), reference
// cedr/cedr_qlt_inl.hpp:14-66, on the field-major layout of this build.
#ifndef COMPOSE_AMD_CEDR_QLT_INL_HPP
#define COMPOSE_AMD_CEDR_QLT_INL_HPP

namespace cedr {
namespace qlt {

CEDR_DEVICE_INLINE void QLTHip::DeviceOp::
set_rhom (const Int& lclcellidx, const Int& /*rhomidx*/, const Real& rhom) const {
  l2r_data[lclcellidx] = rhom;
}

CEDR_DEVICE_INLINE void QLTHip::DeviceOp::
set_Qm (const Int& lclcellidx, const Int& tracer_idx, const Real& Qm, const Real& Qm_min,
        const Real& Qm_max, const Real Qm_prev) const {
  Real* const bd = l2r_data + static_cast<int64_t>(trcr2bl2r[tracer_idx])*stride + lclcellidx;
  const Int problem_type = trcr2prob[tracer_idx];
  Int next = 0;
  if (problem_type & ProblemType::shapepreserve) {
    bd[0] = Qm_min; bd[stride] = Qm; bd[2*stride] = Qm_max;
    next = 3;
  } else if (problem_type & ProblemType::consistent) {
    const Real rhom = l2r_data[lclcellidx];
    bd[0] = Qm_min/rhom; bd[stride] = Qm; bd[2*stride] = Qm_max/rhom;
    next = 3;
  } else {
    bd[0] = Qm;
    next = 1;
  }
  if (problem_type & ProblemType::conserve) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (Qm_prev == std::numeric_limits<Real>::infinity()) __builtin_trap();
#endif
    bd[next*stride] = Qm_prev;
  }
}

CEDR_DEVICE_INLINE Real QLTHip::DeviceOp::
get_Qm (const Int& lclcellidx, const Int& tracer_idx) const {
  return r2l_data[static_cast<int64_t>(trcr2br2l[tracer_idx])*stride + lclcellidx];
}

}  // namespace qlt
}  // namespace cedr

#endif
