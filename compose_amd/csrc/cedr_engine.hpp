// Device-resident form of a plan::Plan plus the leaves->root driver shared by
// QLT and BfbTreeAllReducer: local stages, ONE all-gather of the pure
// subtrees' root records, replicated top stages.
#ifndef COMPOSE_AMD_CEDR_ENGINE_HPP
#define COMPOSE_AMD_CEDR_ENGINE_HPP

#include <memory>
#include <vector>

#include "../../include/cedr_hip.h"
#include "cedr_device.hpp"
#include "cedr_plan.hpp"

namespace cedr {
namespace engine {

struct DeviceStage {
  dev::Buffer<cedr_hip_tile> tiles;
  dev::Buffer<uint16_t> kids, micro, outs;
  dev::Buffer<int32_t> lvlptr;
  cedr_hip_stage h;
  cedr_hip_tile tile0;  // host copy of the first tile's descriptor (the single-tile stages: top, apex)
  void upload(const plan::Stage& st);
};

// The planner's options as every CDR of this build sets them: the tuning / testing knobs of the environment
// (CEDR_HIP_TILE_LEAVES, CEDR_HIP_MICRO, CEDR_HIP_MICRO_CUT, CEDR_HIP_MICRO_CUT2, CEDR_HIP_EXCHANGE, CEDR_HIP_TOP,
// CEDR_HIP_TOP_MAX) and, when a GPU is present, the fused top tile bounded by what the device grants a workgroup in
// LDS (the top kernels keep 72 B per internal node there + the apex: 147 KB + 9 KB for 2048 inputs, which fits
// gfx950's 160 KB; a device with less gets a smaller top tile, one with < 24 KB none -- the per-stage launches).
void plan_options (plan::Options& o);

class TreeEngine {
public:
  TreeEngine(const mpi::Parallel::Ptr& p, const tree::FlatTree& t, const tree::NodeSets& ns,
             const plan::Options& opt);

  // The constructor only builds the host plan; upload() puts it on the GPU
  // (called from the owners' finish_setup). Needs a device.
  void upload();
  bool uploaded () const { return uploaded_; }
  // Size the exchange buffers for sweeps of up to 'nfields' fields now (owners call it from
  // finish_setup), so that no run() allocates or frees device memory.
  void reserve_exchange(int32_t nfields);

  const plan::Plan& get_plan () const { return plan_; }
  // Doubles between consecutive fields: the plan's slot count rounded up to a
  // multiple of 16 so that every field row starts on a 128-byte line.
  int64_t stride () const { return stride_; }
  int32_t nslots () const { return plan_.nslots; }

  // Leaves->root over 'nfields' fields of 'data' (field-major, stride()).
  // After it returns (stream ordered) data[f*stride + root_slot] holds the
  // global root's record on every rank.
  // in_rows (optional, device table of nfields row pointers): the first local stage reads
  // its inputs from in_rows[f][slot] instead of data (zero-copy mode).
  // skip_top: leave out the fused top tile (has_top()); the caller sweeps it itself (QLT on one
  // rank: cedr_hip_qlt_top mode 1 re-forms the tile's sums and computes the root in its one launch).
  // root_out (device, nfields doubles; single rank with a fused top tile only -- see takes_root_out()): the
  // top tile's launch also writes the root's record there, which saves the caller a gather launch.
  // first_done: the caller's own kernel has swept the first local stage already -- its tiles' outputs are in
  // data (CAAS fused with the BFB reducer, cedr_hip_caas_reduce_local_bfb); the sweep starts at the second.
  void up(int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init,
          hipStream_t stream, const double* const* in_rows = nullptr, bool skip_top = false,
          double* root_out = nullptr, bool first_done = false);
  bool takes_root_out () const { return has_top() && plan_.nranks == 1 && ! plan_.level_mode; }
  // Milliseconds the first local leaves->root stage takes over 'data' (best of three launches; it only sums, and
  // writes the stage's tile-output slots of data), or a negative value when that stage is too small for the answer
  // to matter. For owners that allocate 'data' themselves: the same kernel over the same bytes runs up to 12 %
  // faster or slower depending on which device memory the allocation got (profiles/r03_experiments.md 12), so an
  // owner may try a few allocations and keep the best.
  float time_first_stage(int32_t nfields, const uint8_t* op_dev, double* data, hipStream_t stream);
  // The same for the first local root->leaves stage of QLT (cedr_hip_qlt_down over l2r / r2l).
  float time_first_stage_down(const cedr_hip_qlt_layout* lay, double* l2r, double* r2l, hipStream_t stream);
  // The fused top tile (plan::Stage::top): the last local stage, swept by the tracer-parallel top
  // kernels in one launch per direction.
  // Recorded (if set) on the sweep's stream right after the first local leaves->root stage of every up().
  void set_after_first_stage_event (hipEvent_t e) { first_ev_ = e; }
  bool has_top () const { return ! plan_.local.empty() && plan_.local.back().top; }
  const cedr_hip_stage* top_stage () const { return &local_.back()->h; }
  const cedr_hip_tile* top_tile () const { return &local_.back()->tile0; }  // host copy of its descriptor
  const cedr_hip_tile* apex_tile () const { return &apex_->tile0; }
  // The owner's promise that every field's combine rule is a plain sum (op_dev all zero).
  void set_all_sums (bool v) { all_sums_ = v; }
  // Multi-rank runs whose replicated top fits one apex tile (plan::Plan::apex): the leaves->root
  // sweep stops at the exchange -- local stages, gather of this rank's boundary roots, ONE
  // all-gather -- and returns the gathered records [record][nfields] for the apex kernels
  // (cedr_hip_qlt_apex / cedr_hip_tree_apex_sum), which do the rest in one launch.
  bool has_apex () const { return apex_ != nullptr; }
  const double* up_to_exchange(int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init,
                               hipStream_t stream, const double* const* in_rows = nullptr,
                               bool first_done = false);
  const cedr_hip_stage* apex_stage () const { return &apex_->h; }
  const int32_t* apex_src () const { return apex_src_.data(); }       // record of every apex input
  const int32_t* apex_my_in () const { return apex_my_in_.data(); }   // apex inputs that are my roots
  const int32_t* my_root_slots_dev () const { return my_root_slots_.data(); }
  int32_t n_my_roots () const { return static_cast<int32_t>(plan_.my_root_slots.size()); }
  // Whether the first local stage can read through a row table (it must be a micro stage).
  bool first_stage_takes_rows () const;

  // Level mode (plan::LevelStep): up() runs the local stages and then the steps in ascending
  // level order (my nodes of the level, then the round of messages of that level); the root's
  // record ends up on its owner only unless bcast_root is set (one small all-gather). The
  // root->leaves direction is driven by the owner of the data (QLT::run): for every step from the
  // last to the first, level_exchange_down() and then its own kernel over level_nodes().
  bool level_mode () const { return plan_.level_mode; }
  bool owns_root () const { return ! plan_.level_mode || plan_.root_owner == plan_.my_rank; }
  int32_t n_level_steps () const { return static_cast<int32_t>(plan_.steps.size()); }
  int32_t level_nnodes (int32_t i) const { return static_cast<int32_t>(plan_.steps[i].nodes.size()/3); }
  const int32_t* level_nodes (int32_t i) const { return lvl_nodes_.data() + lvl_off_[i].nodes; }
  void level_exchange_down(int32_t i, int32_t nfields, double* data, hipStream_t stream);
  void bcast_root(int32_t nfields, double* data, hipStream_t stream);

  const std::vector<std::unique_ptr<DeviceStage> >& local_stages () const { return local_; }
  const std::vector<std::unique_ptr<DeviceStage> >& top_stages () const { return top_; }
  // Copy 'nfields' fields of my boundary roots from their top-stage input
  // slots to their local slots (root->leaves direction).
  void top_to_local(int32_t nfields, double* data, hipStream_t stream);

private:
  mpi::Parallel::Ptr p_;
  plan::Plan plan_;
  int64_t stride_;
  std::vector<std::unique_ptr<DeviceStage> > local_, top_;
  std::unique_ptr<DeviceStage> apex_;
  dev::Buffer<int32_t> my_root_slots_, my_top_slots_, gather_slots_, apex_src_, apex_my_in_;
  void exchange(int32_t nfields, double* data, hipStream_t stream, bool gathered = false);
  void size_exchange_buffers(int32_t nfields);
  void up_local(int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init, hipStream_t stream,
                const double* const* in_rows, bool skip_top, bool pack_roots, double* root_out = nullptr,
                bool first_done = false);
  struct StepOff { size_t nodes, outs, ins; };
  std::vector<StepOff> lvl_off_;
  dev::Buffer<int32_t> lvl_nodes_, lvl_outs_, lvl_ins_;
  void level_exchange(int32_t i, bool down, int32_t nfields, double* data, hipStream_t stream);
  void up_levels(int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init, hipStream_t stream);
  dev::Buffer<double> sendbuf_, recvbuf_, bcast_, bcast_send_;
  dev::Buffer<int32_t> root_slot_dev_;
  hipEvent_t first_ev_ = nullptr;
  int32_t xfields_ = 0;
  std::vector<mpi::Comm::P2P> ops_;  // scratch of level_exchange (reused: no allocation per round)
  bool uploaded_ = false;
  bool all_sums_ = false;
};

}  // namespace engine
}  // namespace cedr

#endif
