// QLT: quasi-local tree-based non-iterative tracer density reconstructor.
// Public surface follows reference cedr/cedr_qlt.hpp:28-218; the execution
// underneath is the tile/stage engine of cedr_plan.hpp on MI355X.
#ifndef COMPOSE_AMD_CEDR_QLT_HPP
#define COMPOSE_AMD_CEDR_QLT_HPP

#include <memory>
#include <string>
#include <vector>

#include "cedr_cdr.hpp"
#include "cedr_tree.hpp"
#include "cedr_util.hpp"

typedef struct ihipStream_t* hipStream_t;

typedef struct ihipEvent_t* hipEvent_t;

namespace cedr {
namespace engine { class TreeEngine; }
namespace qlt {
using cedr::mpi::Parallel;

class QLTHip : public cedr::CDR {
public:
  typedef std::shared_ptr<QLTHip> Ptr;

  // reference cedr_qlt.hpp:46-109
  struct MetaData {
    enum : Int { nprobtypes = 6 };
    struct CPT {
      enum {
        s  = ProblemType::shapepreserve,
        st = ProblemType::shapepreserve | ProblemType::consistent,
        cs  = ProblemType::conserve | s,
        cst = ProblemType::conserve | st,
        t = ProblemType::consistent,
        ct = ProblemType::conserve | t,
        nn = ProblemType::nonnegative,
        cnn = ProblemType::conserve | nn
      };
    };
    std::vector<Int> trcr2prob, bidx2trcr, trcr2bidx, trcr2bl2r, trcr2br2l;
    Int prob2trcrptr[nprobtypes+1], prob2bl2r[nprobtypes+1], prob2br2l[nprobtypes+1];

    static int get_problem_type(const int& idx);
    static int get_problem_type_idx(const int& mask);  // throws on an invalid mask
    static int get_problem_type_l2r_bulk_size(const int& mask);
    static int get_problem_type_r2l_bulk_size(const int& mask);
    void init(const std::vector<int>& trcr2prob);
  };

  // Both ncells and tree refer to the global mesh (reference cedr_qlt.hpp:129-132).
  QLTHip(const Parallel::Ptr& p, const Int& ncells, const tree::Node::Ptr& tree,
         CDR::Options options = Options());
  // Same, from the array form of the tree.
  QLTHip(const Parallel::Ptr& p, const Int& ncells, const tree::FlatTree& tree,
         CDR::Options options = Options());
  ~QLTHip();

  void print(std::ostream& os) const override;
  Int nlclcells() const;
  void get_owned_glblcells(std::vector<Long>& gcis) const;
  Int gci2lci(const Int& gci) const;

  void declare_tracer(int problem_type, const Int& rhomidx) override;
  void end_tracer_declarations() override;
  void get_buffers_sizes(size_t& buf1, size_t& buf2) override;
  // What this build really touches: ndps * stride Reals per buffer, about half of the reference's
  // figure above (internal tree nodes other than tile outputs never reach HBM here). A caller that
  // brings its own memory may size it by this instead; set_buffers accepts either.
  void get_buffers_sizes_tight(size_t& buf1, size_t& buf2);
  void set_buffers(Real* buf1, Real* buf2) override;
  void finish_setup() override;
  int get_problem_type(const Int& tracer_idx) const override;
  Int get_num_tracers() const override;

  // Trivially copyable; pass by value into device kernels. Bulk data are
  // field-major: l2r[field*stride + cell] (see include/cedr_hip.h).
  struct DeviceOp : public CDR::DeviceOp {
    CEDR_DEVICE_INLINE void set_rhom(const Int& lclcellidx, const Int& rhomidx,
                                     const Real& rhom) const;
    CEDR_DEVICE_INLINE void set_Qm(const Int& lclcellidx, const Int& tracer_idx,
                                   const Real& Qm, const Real& Qm_min, const Real& Qm_max,
                                   const Real Qm_prev = std::numeric_limits<Real>::infinity()) const;
    CEDR_DEVICE_INLINE Real get_Qm(const Int& lclcellidx, const Int& tracer_idx) const;

    Real* l2r_data;
    Real* r2l_data;
    int64_t stride;
    const Int* trcr2prob;  // device arrays, caller tracer order
    const Int* trcr2bl2r;
    const Int* trcr2br2l;
  };
  const DeviceOp& get_device_op() override;

  void run() override;

  // ---- additions of this build ----
  // All work is enqueued on this stream (default: the null stream).
  void set_stream(hipStream_t stream);
  // An event run() records on its stream just before the first-stage root->leaves sweep, the one
  // long kernel of a run that is bound by fp64 issue and leaves HBM bandwidth unused: a caller
  // with independent memory-bound work (CAAS on other tracers, a halo copy) lets another stream
  // wait for it (hipStreamWaitEvent) so that the two run side by side. Created on first call.
  hipEvent_t sweep_down_event();
  // The same for the point where run() has finished its first-stage leaves->root sweep, i.e. the start of its
  // short middle part of small, latency-bound launches (stages 2.., top), during which most of the GPU is idle.
  hipEvent_t sweep_mid_event();
  // Bulk forms of the DeviceOp calls over all local cells; device arrays laid
  // out [tracer][cell] in caller tracer order (rhom: [cell]).
  void set_rhom_bulk(const Real* rhom);
  void set_Qm_bulk(const Real* Qm, const Real* Qm_min, const Real* Qm_max, const Real* Qm_prev);
  void get_Qm_bulk(Real* Qm_out) const;
  // Zero-copy mode: run() sweeps the caller's own device arrays -- rhom[cell];
  // Qm, Qm_min, Qm_max, Qm_prev [tracer][cell] in caller tracer order -- and writes the limited
  // masses to Qm_out[tracer][cell] (may be Qm): no set_rhom / set_Qm / get_Qm pass at all. Only
  // tracers whose bounds are stored as mixing ratios (consistent without shapepreserve) are
  // still packed, by a small kernel inside run(). rhom == nullptr switches back.
  void bind_arrays(const Real* rhom, const Real* Qm, const Real* Qm_min, const Real* Qm_max,
                   const Real* Qm_prev, Real* Qm_out);
  // What finish_setup did about the placement of the buffers it allocated itself (CEDR_HIP_PLACE_TRIES): "tries N;
  // l2r ms: <time of both first-stage sweeps over every candidate> kept <index>; r2l ms: ... kept <index>".
  const std::string& placement_report () const;
  const MetaData& metadata () const { return md_; }
  const tree::NodeSets& nodesets () const { return *ns_; }
  const engine::TreeEngine& tree_engine () const { return *eng_; }

private:
  struct Impl;
  Parallel::Ptr p_;
  std::shared_ptr<const tree::NodeSets> ns_;
  std::unique_ptr<engine::TreeEngine> eng_;
  std::unique_ptr<tree::FlatTree> tree_keep_;  // the caller's tree, until end_tracer_declarations has chosen the plan's cuts
  util::IdMap gci2lci_;
  std::unique_ptr<std::vector<int> > mdb_;  // tracer declarations in progress
  MetaData md_;
  std::unique_ptr<Impl> d_;
  DeviceOp o;

  void init(const Parallel::Ptr& p, const Int& ncells, const tree::FlatTree& tree);
};

// The reference's class is a template on the Kokkos execution space; any
// argument maps to the one HIP implementation so `QLT<>` callers compile.
template <typename ExeSpace = cedr::Hip> using QLT = QLTHip;

}  // namespace qlt
}  // namespace cedr

#include "cedr_qlt_inl.hpp"

#endif
