// Constrained Density Reconstructor plugin interface, source compatible with
// reference cedr/cedr_cdr.hpp:16-112. Find Qm with e'Qm = Qm_global and
// Qm_min <= Qm <= Qm_max.
#ifndef COMPOSE_AMD_CEDR_CDR_HPP
#define COMPOSE_AMD_CEDR_CDR_HPP

#include <iostream>
#include <limits>
#include <memory>

#include "cedr_mpi.hpp"

// Device-callable marker for the DeviceOp methods (the reference's
// KOKKOS_INLINE_FUNCTION). Plain inline when not compiled by hipcc.
#if defined(__HIPCC__)
# define CEDR_DEVICE_INLINE __host__ __device__ inline
#else
# define CEDR_DEVICE_INLINE inline
#endif

namespace cedr {

struct CDR {
  typedef std::shared_ptr<CDR> Ptr;

  // reference cedr_cdr.hpp:19-31
  struct Options {
    bool prefer_numerical_mass_conservation_to_numerical_bounds;
    Options () : prefer_numerical_mass_conservation_to_numerical_bounds(false) {}
  };

  CDR (const Options options = Options()) : options_(options) {}
  virtual ~CDR () {}

  virtual void print (std::ostream&) const {}
  const Options& get_options () const { return options_; }

  // Call in caller tracer order; an error after end_tracer_declarations.
  virtual void declare_tracer(int problem_type, const Int& rhomidx) = 0;
  virtual void end_tracer_declarations() = 0;
  // Sizes in units of Real. Valid after end_tracer_declarations.
  virtual void get_buffers_sizes(size_t& buf1, size_t& buf2) = 0;
  // DEVICE memory of at least those sizes, owned by the caller.
  virtual void set_buffers(Real* buf1, Real* buf2) = 0;
  virtual void finish_setup() = 0;
  virtual int get_problem_type(const Int& tracer_idx) const = 0;
  virtual Int get_num_tracers() const = 0;

  // Concrete CDRs derive their DeviceOp from this tag. It has no virtual
  // functions: a DeviceOp is copied by value into device kernels (the
  // reference does the same through a static_cast to the concrete type,
  // cedr_test_randomized_inl.hpp:27-28), so it must stay trivially copyable.
  // Each concrete DeviceOp provides
  //   set_rhom(lclcellidx, rhomidx, rhom)
  //   set_Qm(lclcellidx, tracer_idx, Qm, Qm_min, Qm_max, Qm_prev = inf)
  //   get_Qm(lclcellidx, tracer_idx)
  struct DeviceOp {};

  // reference cedr_cdr.hpp:104. Through a CDR::Ptr this returns the tag base; callers
  // static_cast it to the concrete CDR's DeviceOp before copying it into a kernel, as the
  // reference's callers do (cedr_test_randomized_inl.hpp:27-28). QLT and CAAS override it
  // with a covariant return type, so `qlt.get_device_op()` yields the concrete POD directly.
  virtual const DeviceOp& get_device_op() = 0;

  // Enqueue the algorithm on the CDR's HIP stream over the values set through
  // the DeviceOp; results are read through the DeviceOp afterwards, stream
  // ordered. Not to be called from a parallel region.
  virtual void run() = 0;

protected:
  Options options_;
};

}  // namespace cedr

#endif
