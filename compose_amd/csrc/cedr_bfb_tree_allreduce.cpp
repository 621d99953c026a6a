#include "cedr_bfb_tree_allreduce.hpp"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/cedr_hip.h"
#include "cedr_device.hpp"
#include "cedr_engine.hpp"
#include "cedr_util.hpp"

namespace cedr {

#define cedr_k_check(call) do {                                         \
    const int _e_ = (call);                                             \
    cedr_throw_if(_e_ != 0, "kernel launch failed with HIP error " << _e_ << " in " #call); \
  } while (0)

struct BfbTreeAllReducerHip::Impl {
  hipStream_t stream = nullptr;
  dev::Buffer<Real> bd;          // [nfield][stride]
  dev::Buffer<int32_t> leaf_slots, root_slot;
  dev::Buffer<uint8_t> op;       // all sums
  // Transposed sends (field, leaf) whose leaf order is the slot order are swept in place:
  // the first stage reads send's rows through a row table instead of a scattered copy.
  bool identity = false;             // local leaf id == slot
  // [nfield] row pointers into a send buffer, one table per buffer seen (callers alternate between a
  // few buffers; a table is built once, the first time its buffer shows up: no host stall afterwards)
  struct RowTable { const Real* send = nullptr; std::unique_ptr<dev::Buffer<const Real*> > rows; };
  std::vector<RowTable> tables;
  size_t next_evict = 0;
  // For CAAS fused with this reducer (first_stage_for_fusion): leaves under every tile of the first local stage
  // and whether those tiles cover the local leaves 0 .. nlocal-1 exactly once, in leaf order.
  dev::Buffer<int32_t> tile_nleaves;
  int32_t max_tile_leaves = 0;
  bool fusable = false;
};

void BfbTreeAllReducerHip::init (const mpi::Parallel::Ptr& p, const tree::FlatTree& tree,
                                 const Int nleaf, const Int nfield) {
  p_ = p;
  ns_ = tree::analyze(p->rank(), nleaf, tree);
  cedr_throw_if(ns_->levels.empty() || ns_->levels[0].nodes.empty(),
                "BfbTreeAllReducer does not support 0 leaves on a rank.");
  nlocal_ = static_cast<Int>(ns_->levels[0].nodes.size());
  nfield_ = nfield;
  plan::Options popt;
  engine::plan_options(popt);
  // The first stage's tiles: QLT cuts them to two LDS levels because its root->leaves sweep is bound by the chain of
  // dependent levels; this reducer only sweeps leaves->root, where a level costs a wave one LDS round trip, and a cut
  // first stage means one more (small, latency-bound) stage = launch before the top tile (round 5, 375k leaves x 160
  // fields: CAAS::run with this reducer 0.654 -> 0.620 ms fused, 0.889 -> 0.870 ms as separate launches; as rank 0 of 8
  // 0.113 -> 0.108 ms). CEDR_HIP_BFB_CUT = LDS levels kept (default 0 = all).
  popt.micro_cut = 0;
  if (const char* e = std::getenv("CEDR_HIP_BFB_CUT")) popt.micro_cut = std::atoi(e);
  eng_.reset(new engine::TreeEngine(p, tree, *ns_, popt));
  m_.reset(new Impl());
}

BfbTreeAllReducerHip::BfbTreeAllReducerHip (const mpi::Parallel::Ptr& p,
                                            const tree::Node::Ptr& tree, const Int nleaf,
                                            const Int nfield) {
  init(p, tree::flatten(tree), nleaf, nfield);
}

BfbTreeAllReducerHip::BfbTreeAllReducerHip (const mpi::Parallel::Ptr& p,
                                            const tree::FlatTree& tree, const Int nleaf,
                                            const Int nfield) {
  init(p, tree, nleaf, nfield);
}

BfbTreeAllReducerHip::~BfbTreeAllReducerHip () {}

void BfbTreeAllReducerHip::get_host_buffers_sizes (size_t& buf1, size_t& buf2) { buf1 = buf2 = 0; }
void BfbTreeAllReducerHip::set_host_buffers (Real*, Real*) {}
void BfbTreeAllReducerHip::set_stream (hipStream_t stream) {
  dev::follow_stream(m_->stream, stream);
  m_->stream = stream;
}

void BfbTreeAllReducerHip::finish_setup () {
  if (m_->bd.size()) return;
  eng_->upload();
  eng_->reserve_exchange(nfield_);
  m_->bd.alloc_zero(static_cast<size_t>(nfield_)*eng_->stride());
  // send's i-th leaf is levels[0].nodes[i] (cedr_bfb_tree_allreduce.cpp:91-102).
  std::vector<int32_t> ls(nlocal_);
  for (Int i = 0; i < nlocal_; ++i) ls[i] = ns_->node_h(ns_->levels[0].nodes[i])->offset;
  m_->leaf_slots.upload(ls);
  m_->identity = true;
  for (Int i = 0; i < nlocal_; ++i) if (ls[i] != i) m_->identity = false;
  m_->root_slot.upload(std::vector<int32_t>(1, eng_->get_plan().root_slot));
  m_->op.upload(std::vector<uint8_t>(nfield_, 0));
  eng_->set_all_sums(true);  // every field is a sum (0 + kid0 + kid1: the zero-init flavour of the sum kernels)
  // May a caller's kernel sweep the first stage itself? Micro tiles whose leaves are the local leaves in order,
  // every local leaf under exactly one tile (a leaf that is itself a rank's boundary root is under none).
  const plan::Plan& pl = eng_->get_plan();
  if (m_->identity && eng_->first_stage_takes_rows() && ! pl.local.empty() && ! pl.local[0].top) {
    const plan::Stage& st = pl.local[0];
    std::vector<int32_t> nl(st.tiles.size());
    int64_t next = 0;
    bool ok = true;
    for (size_t t = 0; t < st.tiles.size() && ok; ++t) {
      const plan::Tile& tl = st.tiles[t];
      if (tl.nleaf < 1 || tl.in0 != next) { ok = false; break; }
      int32_t covered = 0;  // the micro-subtrees of a tile follow each other without gaps
      for (int32_t i = 0; i < tl.nleaf && ok; ++i) {
        const uint16_t dl = st.micro[tl.micro_off + i];
        const int shape = dl >> 8;
        if ((dl & 0xff) != covered) ok = false;
        covered += shape == 0 ? 1 : (shape == 1 ? 2 : 3);
      }
      nl[t] = covered;
      next += nl[t];
      m_->max_tile_leaves = std::max(m_->max_tile_leaves, nl[t]);
    }
    if (ok && next == nlocal_) {
      m_->tile_nleaves.upload(nl);
      m_->fusable = true;
    }
  }
}

bool BfbTreeAllReducerHip::first_stage_for_fusion (FirstStage& fs) const {
  if ( ! m_->bd.size()) const_cast<BfbTreeAllReducerHip*>(this)->finish_setup();
  if ( ! m_->fusable) return false;
  fs.stage = &eng_->local_stages()[0]->h;
  fs.tile_nleaves = m_->tile_nleaves.data();
  fs.max_tile_leaves = m_->max_tile_leaves;
  fs.stride = eng_->stride();
  fs.data = m_->bd.data();
  return true;
}

void BfbTreeAllReducerHip::allreduce_after_first_stage (Real* recv, hipStream_t s) const {
  cedr_throw_if( ! m_->fusable, "allreduce_after_first_stage: this reducer's first stage cannot be fused");
  finish(nullptr, recv, s, true);
}

// Everything after the leaf fill: the local stages (from the second if first_done), the exchange, the top.
void BfbTreeAllReducerHip::finish (const Real* const* in_rows, Real* recv, hipStream_t s, const bool first_done) const {
  Real* const bd = m_->bd.data();
  const int64_t S = eng_->stride();
  // Every node is 0 + kid0 + kid1, as in cedr_bfb_tree_allreduce.cpp:117-123.
  if (eng_->has_apex()) {  // multi-rank, small replicated top: one launch after the all-gather
    const Real* const gathered = eng_->up_to_exchange(nfield_, m_->op.data(), bd, true, s, in_rows, first_done);
    cedr_k_check(cedr_hip_tree_apex_sum(eng_->apex_stage(), gathered, eng_->apex_src(), nfield_, 1, recv, s));
    return;
  }
  if (eng_->takes_root_out()) {  // one rank: the top tile's launch writes the root's record to recv itself
    eng_->up(nfield_, m_->op.data(), bd, true, s, in_rows, false, recv, first_done);
    return;
  }
  eng_->up(nfield_, m_->op.data(), bd, true, s, in_rows, false, nullptr, first_done);
  eng_->bcast_root(nfield_, bd, s);  // level mode only: cedr_bfb_tree_allreduce.cpp:68-76, the way back down
  // The root's record is the result on every rank (:68-76 read it back from
  // the first leaf after the root->leaves copies; the value is the same).
  cedr_k_check(cedr_hip_gather_slots(bd, S, nfield_, m_->root_slot.data(), 1, recv, s));
}

void BfbTreeAllReducerHip::allreduce (const Real* send, Real* recv, const bool transpose) const {
  if ( ! m_->bd.size()) const_cast<BfbTreeAllReducerHip*>(this)->finish_setup();
  hipStream_t s = m_->stream;
  Real* const bd = m_->bd.data();
  const int64_t S = eng_->stride();
  const Real* const* in_rows = nullptr;
  if (transpose && m_->identity && eng_->first_stage_takes_rows()) {
    Impl::RowTable* tab = nullptr;
    for (auto& t : m_->tables) if (t.send == send) tab = &t;
    if ( ! tab) {  // first time this buffer is seen: rows of 'send' are nlocal apart
      std::vector<const Real*> r(nfield_);
      for (Int f = 0; f < nfield_; ++f) r[f] = send + static_cast<size_t>(f)*nlocal_;
      constexpr size_t kMaxTables = 8;
      if (m_->tables.size() < kMaxTables) {
        m_->tables.emplace_back();
        tab = &m_->tables.back();
      } else {  // recycle the oldest table; a sweep may still be reading it
        tab = &m_->tables[m_->next_evict++ % kMaxTables];
        cedr_hip_check(hipStreamSynchronize(s));
      }
      tab->rows.reset(new dev::Buffer<const Real*>());
      tab->rows->upload(r);
      tab->send = send;
    }
    in_rows = tab->rows->data();
  } else if (transpose) {
    cedr_k_check(cedr_hip_scatter_slots_fm(bd, S, nfield_, m_->leaf_slots.data(), nlocal_, send, s));
  } else {
    cedr_k_check(cedr_hip_scatter_slots(bd, S, nfield_, m_->leaf_slots.data(), nlocal_, send, s));
  }
  finish(in_rows, recv, s, false);
}

}  // namespace cedr
