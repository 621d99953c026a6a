// CAAS: ClipAndAssuredSum. Public surface follows reference
// cedr/cedr_caas.hpp:16-121; the kernels are in cedr_kernels.hip.
#ifndef COMPOSE_AMD_CEDR_CAAS_HPP
#define COMPOSE_AMD_CEDR_CAAS_HPP

#include <memory>
#include <vector>

#include "cedr_cdr.hpp"

typedef struct ihipStream_t* hipStream_t;

// The reference's UserAllReducer takes an MPI_Op. Builds without MPI get a
// stand-in type so caller code keeps compiling; only MPI_SUM is ever passed.
#if ! defined(MPI_VERSION) && ! defined(CEDR_NO_MPI_OP_STANDIN)
typedef int MPI_Op;
# ifndef MPI_SUM
#  define MPI_SUM 1
# endif
#endif

namespace cedr {
class BfbTreeAllReducerHip;
namespace caas {

class CAASHip : public CDR {
public:
  typedef std::shared_ptr<CAASHip> Ptr;

  // reference cedr_caas.hpp:27-49. sendbuf/rcvbuf are DEVICE pointers, as in
  // the reference's GPU build.
  struct UserAllReducer {
    typedef std::shared_ptr<const UserAllReducer> Ptr;
    virtual ~UserAllReducer () {}
    virtual int operator()(const mpi::Parallel& p, Real* sendbuf, Real* rcvbuf,
                           int nlocal, int nfld, MPI_Op op) const = 0;
    virtual int n_accum_in_place () const { return 1; }
  };

  CAASHip(const mpi::Parallel::Ptr& p, const Int nlclcells,
          const UserAllReducer::Ptr& r = nullptr);
  ~CAASHip();

  void declare_tracer(int problem_type, const Int& rhomidx) override;
  void end_tracer_declarations() override;
  void get_buffers_sizes(size_t& buf1, size_t& buf2) override;
  void set_buffers(Real* buf1, Real* buf2) override;
  void finish_setup() override;
  int get_problem_type(const Int& tracer_idx) const override;
  Int get_num_tracers() const override;

  // reference cedr_caas_inl.hpp:14-42: d is SoA, d[(1 + a*nt + tracer)*n + cell].
  struct DeviceOp : public CDR::DeviceOp {
    CEDR_DEVICE_INLINE void set_rhom (const Int& lclcellidx, const Int&, const Real& rhom) const {
      d_[lclcellidx] = rhom;
    }
    CEDR_DEVICE_INLINE void set_Qm (const Int& lclcellidx, const Int& tracer_idx, const Real& Qm,
                                    const Real& Qm_min, const Real& Qm_max,
                                    const Real Qm_prev = std::numeric_limits<Real>::infinity()) const {
      const int64_t n = nlclcells_;
      d_[(1 +         tracer_idx)*n + lclcellidx] = Qm;
      d_[(1 +   nt_ + tracer_idx)*n + lclcellidx] = Qm_min;
      d_[(1 + 2*nt_ + tracer_idx)*n + lclcellidx] = Qm_max;
      if (need_conserve_) d_[(1 + 3*nt_ + tracer_idx)*n + lclcellidx] = Qm_prev;
    }
    CEDR_DEVICE_INLINE Real get_Qm (const Int& lclcellidx, const Int& tracer_idx) const {
      return d_[(1 + tracer_idx)*static_cast<int64_t>(nlclcells_) + lclcellidx];
    }
    Int nlclcells_, nrhomidxs_, nt_;
    bool need_conserve_;
    Real* d_;
  };
  const DeviceOp& get_device_op() override;

  void run() override;

  // ---- additions of this build ----
  void set_stream(hipStream_t stream);
  // If set, reduce_locally reads the unclipped Qm from this device array
  // ([tracer][cell]) instead of from the data buffer; the clipped and final
  // values still land in the data buffer. Same arithmetic and traffic as the
  // in-place form; lets a caller keep its own Qm untouched.
  void set_Qm_source(const Real* qm_src);
  // Zero-copy mode: run() limits the caller's own [tracer][cell] device arrays
  // in place -- Qm is overwritten with the result -- instead of the internal SoA buffer, so no
  // set_Qm / get_Qm pass is needed. Qm == nullptr returns to the internal buffer.
  void bind_arrays(Real* Qm, const Real* Qm_min, const Real* Qm_max, const Real* Qm_prev);
  const Real* send_data () const;
  const Real* recv_data () const;
  // Whether run() takes the fused CAAS + BFB reducer path (decided in finish_setup).
  bool fused_with_bfb_reducer () const;

  // Exposed for tests: the three steps of run(), reference cedr_caas.cpp:129-253.
  void reduce_locally();
  void reduce_globally();
  void finish_locally();

private:
  struct Impl;
  mpi::Parallel::Ptr p_;
  UserAllReducer::Ptr user_reducer_;
  std::unique_ptr<std::vector<int> > tracer_decls_;
  std::vector<Int> probs_h_;
  bool finished_setup_;
  std::unique_ptr<Impl> m_;
  DeviceOp o;

  void get_buffers_sizes(size_t& buf1, size_t& buf2, size_t& buf3);
};

template <typename ExeSpace = cedr::Hip> using CAAS = CAASHip;

// THE PRODUCTION PAIRING as a type: a UserAllReducer that hands CAAS's per-group partial sums
// (cedr_caas.cpp:137-168: send(field, group), i.e. the reducer's transposed layout) to this library's
// BfbTreeAllReducer (cedr_bfb_tree_allreduce.cpp:79-159) -- what a host model writes by hand around the
// reference. Used through operator() it is exactly that; but a CAAS that finds this type as its reducer
// fuses its reduce pass with the first stage of the reducer's tree (cedr_hip_caas_reduce_local_bfb): the
// partial sums stay on chip, same bits. CEDR_HIP_CAAS_BFB_FUSE=0 keeps the two separate launches.
// The reducer's leaf i must be the group of cells [i*n_accum_in_place, (i+1)*n_accum_in_place) and it must
// reduce 4*nt fields; any other tree / leaf order still works, unfused.
struct BfbUserAllReducer : public CAASHip::UserAllReducer {
  BfbUserAllReducer(const std::shared_ptr<BfbTreeAllReducerHip>& bfb, int n_accum_in_place);
  int operator()(const mpi::Parallel& p, Real* sendbuf, Real* rcvbuf, int nlocal, int nfld,
                 MPI_Op op) const override;
  int n_accum_in_place () const override { return nacc_; }
  const std::shared_ptr<BfbTreeAllReducerHip>& reducer () const { return bfb_; }
private:
  std::shared_ptr<BfbTreeAllReducerHip> bfb_;
  int nacc_;
};

}  // namespace caas
}  // namespace cedr

#endif
