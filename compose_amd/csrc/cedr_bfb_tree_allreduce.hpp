// All-reduce (sum) of nfield scalars per leaf along the caller's tree, so that
// the result is bit-for-bit independent of the rank decomposition. Public
// surface follows reference cedr/cedr_bfb_tree_allreduce.hpp:14-57. The
// reference copies device data to the host and reduces there with per-level
// MPI messages (cedr_bfb_tree_allreduce.cpp:55-66,104-156); this build keeps
// everything on the GPU and uses the tile/stage engine with one all-gather.
#ifndef COMPOSE_AMD_CEDR_BFB_TREE_ALLREDUCE_HPP
#define COMPOSE_AMD_CEDR_BFB_TREE_ALLREDUCE_HPP

#include <memory>

#include "../../include/cedr_hip.h"
#include "cedr_tree.hpp"

typedef struct ihipStream_t* hipStream_t;

namespace cedr {
namespace engine { class TreeEngine; }

class BfbTreeAllReducerHip {
public:
  typedef std::shared_ptr<BfbTreeAllReducerHip> Ptr;

  BfbTreeAllReducerHip(const mpi::Parallel::Ptr& p, const tree::Node::Ptr& tree,
                       const Int nleaf, const Int nfield);
  BfbTreeAllReducerHip(const mpi::Parallel::Ptr& p, const tree::FlatTree& tree,
                       const Int nleaf, const Int nfield);
  ~BfbTreeAllReducerHip();

  // No host staging exists in this build: both sizes are 0 and set_host_buffers
  // is accepted and ignored, so reference call sequences keep working.
  void get_host_buffers_sizes(size_t& buf1, size_t& buf2);
  void set_host_buffers(Real* buf1, Real* buf2);
  void finish_setup();

  // DEVICE pointers. recv(nfield); send(nfield, nlocal) -- fastest index last --
  // if transpose, else send(nlocal, nfield). send and recv may alias.
  void allreduce(const Real* send, Real* recv, const bool transpose = false) const;

  Int nlocal () const { return nlocal_; }
  Int nfield () const { return nfield_; }
  void set_stream(hipStream_t stream);

  // ---- additions of this build: the reducer's first stage swept by the CALLER's kernel ----
  // CAAS paired with this reducer (caas::BfbUserAllReducer) forms its per-group partial sums and the first
  // leaves->root stage of this tree in one kernel (cedr_hip_caas_reduce_local_bfb), so the partial sums never reach
  // HBM. first_stage_for_fusion: false if the tree does not allow it (the local leaves are not the first stage's
  // inputs in order: tiny trees, a leaf that is a rank boundary by itself, non-contiguous leaf order); otherwise
  // the stage, the leaves under each of its tiles (device array), their maximum, and where the tiles' outputs go:
  // data[field*stride + slot]. allreduce_after_first_stage: the rest of allreduce() once that kernel has run on
  // 'stream' (further stages, the exchange between ranks, the top), recv as in allreduce.
  struct FirstStage {
    const cedr_hip_stage* stage; const int32_t* tile_nleaves; int32_t max_tile_leaves; int64_t stride; Real* data;
  };
  bool first_stage_for_fusion(FirstStage& fs) const;
  void allreduce_after_first_stage(Real* recv, hipStream_t stream) const;

private:
  struct Impl;
  mpi::Parallel::Ptr p_;
  Int nlocal_, nfield_;
  std::shared_ptr<const tree::NodeSets> ns_;
  std::unique_ptr<engine::TreeEngine> eng_;
  std::unique_ptr<Impl> m_;
  void init(const mpi::Parallel::Ptr& p, const tree::FlatTree& tree, const Int nleaf,
            const Int nfield);
  void finish(const Real* const* in_rows, Real* recv, hipStream_t s, bool first_done) const;
};

template <typename ExeSpace = cedr::Hip> using BfbTreeAllReducer = BfbTreeAllReducerHip;

}  // namespace cedr

#endif
