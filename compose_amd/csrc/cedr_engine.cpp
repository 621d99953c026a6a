#include "cedr_engine.hpp"

#include <algorithm>
#include <atomic>
#include <string>
#include <cstdio>
#include <cstdlib>


namespace cedr {
namespace engine {

#define cedr_k_check(call) do {                                         \
    const int _e_ = (call);                                             \
    cedr_throw_if(_e_ != 0, "kernel launch failed with HIP error " << _e_ << " in " #call); \
  } while (0)

void plan_options (plan::Options& o) {
  if (const char* e = std::getenv("CEDR_HIP_TILE_LEAVES")) o.max_tile_leaves = std::atoi(e);
  if (const char* e = std::getenv("CEDR_HIP_MICRO")) {  // 0 = off, 1 = first stage only, 2 = all stages
    o.micro = std::atoi(e) != 0;
    o.micro_all = std::atoi(e) >= 2;
  }
  if (const char* e = std::getenv("CEDR_HIP_MICRO_CUT")) o.micro_cut = std::atoi(e);
  if (const char* e = std::getenv("CEDR_HIP_MICRO_CUT2")) o.micro_cut2 = std::atoi(e);
  // 1 = all-gather + replicated top, 2 = per-level point-to-point; default: decided from the tree
  if (const char* e = std::getenv("CEDR_HIP_EXCHANGE")) o.exchange = std::atoi(e);
  // fused top tile (plan::Stage::top): 0 = off (every stage its own launch, as before round 3)
  if (const char* e = std::getenv("CEDR_HIP_TOP")) o.fused_top = std::atoi(e) != 0;
  if (const char* e = std::getenv("CEDR_HIP_TOP_MAX")) o.top_max_inputs = std::atoi(e);
  // The top kernels' LDS request must be grantable: (4 + kTRS = 9) doubles per internal node of the tile
  // (inputs - 1 of them) + up to ~9 KB for the apex tile and the tables (cedr_hip_qlt_top). Every rank of a run
  // sits on the same kind of GPU, so all ranks decide alike. Without a GPU (host-only planning) nothing is capped.
  int dev = 0, lds = 0, optin = 0;
  if (o.fused_top && hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && lds > 0) {
    // (what hipFuncSetAttribute can raise a kernel's dynamic LDS to, where the runtime tells)
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && optin > lds)
      lds = optin;
    else
      (void) hipGetLastError();
    // (CEDR_HIP_LDS_LIMIT: plan as for a device that grants this many bytes -- how tests/test_gpu_qlt.py reaches the branches
    // below on a GPU that has 160 KB; it can only lower the figure)
    if (const char* e = std::getenv("CEDR_HIP_LDS_LIMIT")) { const int v = std::atoi(e); if (v > 0 && v < lds) lds = v; }
    if (std::getenv("CEDR_HIP_PLAN_VERBOSE")) std::fprintf(stderr, "cedr: device %d grants %d B of LDS per workgroup\n", dev, lds);
    const int64_t room = static_cast<int64_t>(lds) - 10*1024;
    int32_t cap = 2;
    while (cap < 2048 && 72*static_cast<int64_t>(2*cap - 1) <= room) cap *= 2;
    const int32_t asked = o.top_max_inputs;
    if (72*static_cast<int64_t>(cap - 1) > room) o.fused_top = false;           // not even a 2-input tile
    else if (cap < 256) o.fused_top = false;                                     // too small to replace a stage
    else if (o.top_max_inputs > cap) o.top_max_inputs = cap;
    // A smaller or missing top tile changes every plan of the process: say so once, not only under a verbose knob
    // (a gfx950 grants 160 KB; a runtime that reports 64 KB there would otherwise change the plans silently).
    static std::atomic<bool> said(false);
    if (( ! o.fused_top || o.top_max_inputs < asked) && ! said.exchange(true))
      std::fprintf(stderr, "cedr (hip): device %d grants %d B of LDS per workgroup: the fused top tile is %s "
                   "(asked for up to %d inputs)\n", dev, lds,
                   o.fused_top ? ("limited to " + std::to_string(o.top_max_inputs) + " inputs").c_str() : "off", asked);
  } else {
    (void) hipGetLastError();
  }
}

void DeviceStage::upload (const plan::Stage& st) {
  static_assert(sizeof(plan::Tile) == sizeof(cedr_hip_tile), "tile layouts must agree");
  std::vector<cedr_hip_tile> tl(st.tiles.size());
  for (size_t i = 0; i < tl.size(); ++i) {
    const plan::Tile& s = st.tiles[i];
    tl[i] = cedr_hip_tile{s.in0, s.nleaf, s.out, s.nint, s.kids_off, s.lvl_off, s.nlvl, s.micro_off,
                          s.nout, s.outs_off};
  }
  tiles.upload(tl);
  tile0 = tl.empty() ? cedr_hip_tile{0, 0, 0, 0, 0, 0, 0, 0, 0, 0} : tl[0];
  kids.upload(st.kids);
  lvlptr.upload(st.lvlptr);
  h.ntiles = static_cast<int32_t>(st.tiles.size());
  h.max_nodes = st.max_nodes;
  h.max_nint = st.max_nint;
  h.max_nleaf = st.max_nleaf;
  h.tiles = tiles.data();
  h.kids = kids.data();
  h.lvlptr = lvlptr.data();
  micro.upload(st.micro);
  h.micro = st.micro.empty() ? nullptr : micro.data();
  outs.upload(st.outs);
  h.outs = outs.data();
  h.flags = 0;
  if ( ! st.micro.empty()) {
    bool uni = true;
    for (size_t i = 0; i < st.tiles.size(); ++i)
      if (st.tiles[i].nleaf != 64 || st.tiles[i].micro_off != static_cast<int32_t>(64*i)) uni = false;
    if (uni) h.flags |= CEDR_STAGE_UNIFORM64;
  }
  // Tail-split knobs of the first-stage launches (cedr_kernels.hip tail_split), read here once per stage
  if (const char* e = std::getenv("CEDR_HIP_TAIL"))
    h.flags |= CEDR_STAGE_TAIL_SET | ((std::max(0, std::min(127, std::atoi(e)))) << 16);
  if (const char* e = std::getenv("CEDR_HIP_TAIL_ROUNDS")) h.flags |= (std::max(0, std::min(127, std::atoi(e)))) << 24;
  if (st.top && st.tiles.size() == 1) {
    const plan::Tile& t = st.tiles[0];
    const int32_t* const lp = st.lvlptr.data() + t.lvl_off;
    const uint16_t* const kd = st.kids.data() + 2*static_cast<size_t>(t.kids_off);
    if (t.nlvl > 0) {
      const int32_t n0 = lp[1] - lp[0];
      bool pairs = t.nleaf % 2 == 0 && 2*n0 == t.nleaf;
      for (int32_t j = 0; pairs && j < n0; ++j) pairs = kd[2*j] == 2*j && kd[2*j + 1] == 2*j + 1;
      if (pairs) h.flags |= CEDR_STAGE_LVL0_PAIRS;
    }
    int32_t narrow = t.nlvl;  // first level of the suffix of levels with <= 64 nodes each
    while (narrow > 0 && lp[narrow] - lp[narrow - 1] <= 64) --narrow;
    if (narrow < 1) narrow = 1;
    if (narrow < 255) h.flags |= (narrow + 1) << 8;
  }
}

TreeEngine::TreeEngine (const mpi::Parallel::Ptr& p, const tree::FlatTree& t,
                        const tree::NodeSets& ns, const plan::Options& opt)
  : p_(p)
{
  util::PhaseTimer ptm;
  // Local cell index of every owned leaf = its NodeSets offset (the reference's
  // numbering, visible to callers through get_owned_glblcells).
  util::IdMap gci2lci;
  gci2lci.reserve(ns.levels[0].nodes.size());
  for (const Int idx : ns.levels[0].nodes) gci2lci.add(ns.node_h(idx)->id, ns.node_h(idx)->offset);
  gci2lci.finish();
  std::vector<int32_t> leaf_slot(t.size(), -1);
  const Int me = p->rank();
  for (Int n = 0; n < t.size(); ++n)
    if (t.nkids[n] == 0 && t.rank[n] == me) {
      const int lci = gci2lci.find(static_cast<Int>(t.cellidx[n]));
      cedr_throw_if(lci < 0, "owned leaf " << t.cellidx[n] << " missing from NodeSets");
      leaf_slot[n] = lci;
    }
  // A transport without point-to-point messages (Comm::has_exchange) cannot run a level-mode plan.
  // Full trees: plan for the all-gather instead, which works for any ownership (every rank sees the
  // same communicator type, so all decide alike). Partial trees need the messages: refuse here, in
  // the constructor and on every rank, not at the first level step inside run().
  plan::Options popt = opt;
  if (p->size() > 1 && ! p->comm()->has_exchange()) {
    bool partial = false;
    for (Int n = 0; n < t.size(); ++n)
      if (t.level[n] >= 0 || (t.nkids[n] != 2 && t.nkids[n] != 0)) partial = true;
    cedr_throw_if(partial || popt.exchange == 2,
                  "this tree is partial (tree::Node::level is set), or per-level exchange was requested, "
                  "which needs one round of point-to-point messages per tree level, but the communicator has "
                  "no point-to-point exchange (mpi::Comm::has_exchange): pass make_parallel_callbacks its "
                  "third callback, use RCCL, or the MPI adapter");
    // exchange stays as the caller set it (0 = decide): plan::build takes the all-gather wherever level mode
    // would have been chosen for size reasons (with a warning), and throws if a tree::Node::rank pin would be dropped.
    popt.no_p2p = true;
  }
  ptm.lap("engine: leaf slots");
  plan_ = plan::build(t, me, p->size(), leaf_slot, popt);
  ptm.lap("engine: plan::build");
  // Rounded up to 16 doubles (128-byte rows) with at least two spare slots per
  // row: the kernels' 16-byte cell loads may read up to two slots past a tile.
  stride_ = (static_cast<int64_t>(plan_.nslots) + 2 + 15)/16*16;
}

void TreeEngine::upload () {
  if (uploaded_) return;
  for (const auto& st : plan_.local) {
    local_.emplace_back(new DeviceStage());
    local_.back()->upload(st);
  }
  for (const auto& st : plan_.top) {
    top_.emplace_back(new DeviceStage());
    top_.back()->upload(st);
  }
  if ( ! plan_.top.empty()) {
    my_root_slots_.upload(plan_.my_root_slots);
    my_top_slots_.upload(plan_.my_top_slots);
    gather_slots_.upload(plan_.gather_slots);
  }
  if (plan_.level_mode) {
    my_root_slots_.upload(plan_.my_root_slots);
    std::vector<int32_t> nodes, outs, ins;
    for (const auto& st : plan_.steps) {
      lvl_off_.push_back(StepOff{nodes.size(), outs.size(), ins.size()});
      nodes.insert(nodes.end(), st.nodes.begin(), st.nodes.end());
      outs.insert(outs.end(), st.out_slots.begin(), st.out_slots.end());
      ins.insert(ins.end(), st.in_slots.begin(), st.in_slots.end());
    }
    lvl_nodes_.upload(nodes);
    lvl_outs_.upload(outs);
    lvl_ins_.upload(ins);
  }
  static const bool apex_on = [] { const char* v = std::getenv("CEDR_HIP_APEX"); return ! v || std::atoi(v) != 0; }();
  if (apex_on && ! plan_.apex.tiles.empty()) {
    apex_.reset(new DeviceStage());
    apex_->upload(plan_.apex);
    // input i of the apex tile is top slot in0 + i; which gathered record holds it?
    const plan::Tile& t = plan_.apex.tiles[0];
    std::vector<int32_t> src(t.nleaf, -1), mine;
    for (size_t q = 0; q < plan_.gather_slots.size(); ++q) {
      const int32_t s = plan_.gather_slots[q];
      if (s < 0) continue;
      cedr_throw_if(s < t.in0 || s >= t.in0 + t.nleaf, "apex tile does not cover a boundary root");
      src[s - t.in0] = static_cast<int32_t>(q);
    }
    for (const int32_t v : src) cedr_throw_if(v < 0, "apex input without a gathered record");
    for (const int32_t s : plan_.my_top_slots) mine.push_back(s - t.in0);
    apex_src_.upload(src);
    apex_my_in_.upload(mine);
  }
  if (takes_root_out()) root_slot_dev_.upload(std::vector<int32_t>(1, plan_.root_slot));
  uploaded_ = true;
}

bool TreeEngine::first_stage_takes_rows () const {
  if (plan_.local.empty() || plan_.local[0].micro.empty() || plan_.nlcl < 3) return false;
  // A cell that is itself a boundary root (its sibling belongs to another rank) is never the
  // input of a local tile: the exchange reads its record straight from the internal buffer,
  // which a row table bypasses. Rare (ownership cutting through a pair of cells): such ranks
  // keep the packed path.
  for (const int32_t s : plan_.my_root_slots) if (s < plan_.nlcl) return false;
  return true;
}

// The local stages, leaves->root. pack_roots: the fused top tile's launch also packs this rank's
// boundary roots' records into sendbuf_ (the gather launch fused in).
void TreeEngine::up_local (int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init,
                           hipStream_t stream, const double* const* in_rows, bool skip_top, bool pack_roots,
                           double* root_out, bool first_done) {
  cedr_throw_if( ! uploaded_, "TreeEngine::upload was not called");
  for (size_t i = 0; i < local_.size(); ++i) {
    if (i == 0 && first_done) {
      // (the caller's own kernel swept the first stage: its tile outputs are in 'data' already)
    } else if (plan_.local[i].top) {
      if (skip_top) continue;
      if (root_out)  // single rank: the root's record straight to the caller's buffer
        cedr_k_check(cedr_hip_tree_top_up(&local_[i]->h, &local_[i]->tile0, nfields, op_dev, stride_, data, zero_init,
                                          1, root_slot_dev_.data(), root_out, stream));
      else
        cedr_k_check(cedr_hip_tree_top_up(&local_[i]->h, &local_[i]->tile0, nfields, op_dev, stride_, data, zero_init,
                                          pack_roots ? static_cast<int32_t>(plan_.my_root_slots.size()) : 0,
                                          pack_roots ? my_root_slots_.data() : nullptr,
                                          pack_roots ? sendbuf_.data() : nullptr, stream));
    } else {
      cedr_k_check(cedr_hip_tree_up_rows_sums(&local_[i]->h, nfields, op_dev, stride_, data,
                                              i == 0 ? in_rows : nullptr, plan_.nlcl, zero_init, all_sums_, stream));
    }
    if (i == 0 && first_ev_) cedr_throw_if(hipEventRecord(first_ev_, stream) != hipSuccess, "hipEventRecord failed");
  }
  if (local_.empty() && first_ev_) cedr_throw_if(hipEventRecord(first_ev_, stream) != hipSuccess, "hipEventRecord failed");
}

namespace {
// Best of three timed launches after a warm-up one; negative on any failure.
template <typename Launch>
float best_ms (hipStream_t stream, const Launch& launch) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess) return -1;
  if (hipEventCreate(&e1) != hipSuccess) { (void) hipEventDestroy(e0); return -1; }
  float best = -1;
  for (int rep = 0; rep < 4; ++rep) {
    bool ok = hipEventRecord(e0, stream) == hipSuccess;
    ok = ok && launch() == 0;
    ok = ok && hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float ms = 0;
    if ( ! ok || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { best = -1; break; }
    if (rep > 0 && (best < 0 || ms < best)) best = ms;
  }
  (void) hipEventDestroy(e0);
  (void) hipEventDestroy(e1);
  return best;
}
const int32_t kMinTilesToTime = 2048;  // below that a stage takes a few tens of microseconds
}  // namespace

float TreeEngine::time_first_stage (int32_t nfields, const uint8_t* op_dev, double* data, hipStream_t stream) {
  if ( ! uploaded_ || local_.empty() || plan_.local[0].top || local_[0]->h.ntiles < kMinTilesToTime) return -1;
  return best_ms(stream, [&] {
    return cedr_hip_tree_up_rows_sums(&local_[0]->h, nfields, op_dev, stride_, data, nullptr, plan_.nlcl, 0,
                                      all_sums_, stream);
  });
}

float TreeEngine::time_first_stage_down (const cedr_hip_qlt_layout* lay, double* l2r, double* r2l,
                                         hipStream_t stream) {
  if ( ! uploaded_ || local_.empty() || plan_.local[0].top || local_[0]->h.ntiles < kMinTilesToTime) return -1;
  return best_ms(stream, [&] { return cedr_hip_qlt_down(lay, &local_[0]->h, l2r, r2l, stream); });
}

void TreeEngine::up (int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init,
                     hipStream_t stream, const double* const* in_rows, bool skip_top, double* root_out,
                     bool first_done) {
  const bool xch = ! plan_.level_mode && ! top_.empty();
  if (xch) size_exchange_buffers(nfields);
  cedr_throw_if(root_out && ! takes_root_out(), "root_out needs a single-rank plan with a fused top tile");
  cedr_throw_if(first_done && (local_.empty() || plan_.local[0].top), "first_done needs a first stage of tiles");
  up_local(nfields, op_dev, data, zero_init, stream, in_rows, skip_top, xch && has_top(), root_out, first_done);
  if (plan_.level_mode) { up_levels(nfields, op_dev, data, zero_init, stream); return; }
  if (top_.empty()) return;
  exchange(nfields, data, stream, has_top());
  cedr_k_check(cedr_hip_scatter_slots(data, stride_, nfields, gather_slots_.data(),
                                      plan_.nranks*plan_.nb_max, recvbuf_.data(), stream));
  for (const auto& st : top_)
    cedr_k_check(cedr_hip_tree_up(&st->h, nfields, op_dev, stride_, data, zero_init, stream));
}

void TreeEngine::size_exchange_buffers (int32_t nfields) {
  if (nfields <= xfields_) return;
  if (plan_.level_mode) {
    size_t most = 1;
    for (const auto& s : plan_.steps) most = std::max(most, std::max(s.out_slots.size(), s.in_slots.size()));
    sendbuf_.alloc_zero(most*nfields);
    recvbuf_.alloc_zero(most*nfields);
    ops_.reserve(2*static_cast<size_t>(plan_.nranks));
  } else {
    const int32_t P = plan_.nranks, nb = plan_.nb_max;
    sendbuf_.alloc_zero(static_cast<size_t>(nb)*nfields);
    recvbuf_.alloc_zero(static_cast<size_t>(P)*nb*nfields);
  }
  xfields_ = nfields;
}

void TreeEngine::reserve_exchange (int32_t nfields) {
  if (plan_.nranks <= 1) return;
  size_exchange_buffers(nfields);
  if (plan_.level_mode && static_cast<size_t>(nfields)*plan_.nranks > bcast_.size()) {  // bcast_root's buffers
    bcast_.alloc_zero(static_cast<size_t>(nfields)*plan_.nranks);
    bcast_send_.alloc_zero(nfields);
    root_slot_dev_.upload(std::vector<int32_t>(1, plan_.root_slot));
  }
}

// Gather this rank's boundary roots (unless the top tile's launch already packed them) and
// all-gather them: recvbuf_ = [rank][nb_max][nfields].
void TreeEngine::exchange (int32_t nfields, double* data, hipStream_t stream, bool gathered) {
  const int32_t nb = plan_.nb_max;
  size_exchange_buffers(nfields);
  if ( ! gathered)
    cedr_k_check(cedr_hip_gather_slots(data, stride_, nfields, my_root_slots_.data(),
                                       static_cast<int32_t>(plan_.my_root_slots.size()),
                                       sendbuf_.data(), stream));
  const int err = p_->comm()->allgather(sendbuf_.data(), recvbuf_.data(),
                                        static_cast<size_t>(nb)*nfields, stream);
  cedr_throw_if(err != 0, "allgather of subtree roots returned " << err);
}

const double* TreeEngine::up_to_exchange (int32_t nfields, const uint8_t* op_dev, double* data,
                                          bool zero_init, hipStream_t stream,
                                          const double* const* in_rows, bool first_done) {
  cedr_throw_if( ! uploaded_ || ! apex_, "up_to_exchange needs an uploaded plan with an apex tile");
  cedr_throw_if(first_done && (local_.empty() || plan_.local[0].top), "first_done needs a first stage of tiles");
  size_exchange_buffers(nfields);
  up_local(nfields, op_dev, data, zero_init, stream, in_rows, false, has_top(), nullptr, first_done);
  exchange(nfields, data, stream, has_top());
  return recvbuf_.data();
}

// ---- level mode
void TreeEngine::up_levels (int32_t nfields, const uint8_t* op_dev, double* data, bool zero_init,
                            hipStream_t stream) {
  for (int32_t i = 0; i < n_level_steps(); ++i) {
    if (level_nnodes(i))
      cedr_k_check(cedr_hip_level_up(level_nodes(i), level_nnodes(i), nfields, op_dev, stride_, data,
                                     zero_init, stream));
    level_exchange(i, false, nfields, data, stream);
  }
}

void TreeEngine::level_exchange_down (int32_t i, int32_t nfields, double* data, hipStream_t stream) {
  level_exchange(i, true, nfields, data, stream);
}

// One round of messages of step i. Leaves->root: records at out_slots go to the parents' owners,
// records of in_slots arrive from the kids' owners; root->leaves: the same lists, roles swapped.
void TreeEngine::level_exchange (int32_t i, bool down, int32_t nfields, double* data, hipStream_t stream) {
  const plan::LevelStep& st = plan_.steps[i];
  const auto& send_slots = down ? st.in_slots : st.out_slots;
  const auto& recv_slots = down ? st.out_slots : st.in_slots;
  const auto& send_msgs = down ? st.in_msgs : st.out_msgs;
  const auto& recv_msgs = down ? st.out_msgs : st.in_msgs;
  if (send_msgs.empty() && recv_msgs.empty()) return;
  size_exchange_buffers(nfields);  // (a no-op after reserve_exchange)
  const int32_t* const send_dev = (down ? lvl_ins_.data() + lvl_off_[i].ins : lvl_outs_.data() + lvl_off_[i].outs);
  const int32_t* const recv_dev = (down ? lvl_outs_.data() + lvl_off_[i].outs : lvl_ins_.data() + lvl_off_[i].ins);
  if ( ! send_slots.empty())
    cedr_k_check(cedr_hip_gather_slots(data, stride_, nfields, send_dev, static_cast<int32_t>(send_slots.size()),
                                       sendbuf_.data(), stream));
  std::vector<mpi::Comm::P2P>& ops = ops_;
  ops.clear();
  for (const auto& m : send_msgs)
    ops.push_back(mpi::Comm::P2P{m.peer, sendbuf_.data() + static_cast<size_t>(m.first)*nfields,
                                 static_cast<size_t>(m.count)*nfields, 1});
  for (const auto& m : recv_msgs)
    ops.push_back(mpi::Comm::P2P{m.peer, recvbuf_.data() + static_cast<size_t>(m.first)*nfields,
                                 static_cast<size_t>(m.count)*nfields, 0});
  const int err = p_->comm()->exchange(ops.data(), static_cast<int>(ops.size()), stream);
  cedr_throw_if(err == -2, "this plan exchanges tree levels point to point (a partial tree, or ownership "
                "that does not follow subtrees) but the communicator has no point-to-point exchange");
  cedr_throw_if(err != 0, "point-to-point exchange of tree level " << st.level << " returned " << err);
  if ( ! recv_slots.empty())
    cedr_k_check(cedr_hip_scatter_slots(data, stride_, nfields, recv_dev, static_cast<int32_t>(recv_slots.size()),
                                        recvbuf_.data(), stream));
}

// The root's record from its owner to every rank's root_slot. An all-gather and a pick: bitwise
// (a sum with zeros would lose the sign of a zero).
void TreeEngine::bcast_root (int32_t nfields, double* data, hipStream_t stream) {
  if ( ! plan_.level_mode) return;
  if (static_cast<size_t>(nfields)*plan_.nranks > bcast_.size()) {
    bcast_.alloc_zero(static_cast<size_t>(nfields)*plan_.nranks);
    bcast_send_.alloc_zero(nfields);
    root_slot_dev_.upload(std::vector<int32_t>(1, plan_.root_slot));
  }
  cedr_k_check(cedr_hip_gather_slots(data, stride_, nfields, root_slot_dev_.data(), 1, bcast_send_.data(), stream));
  const int err = p_->comm()->allgather(bcast_send_.data(), bcast_.data(), nfields, stream);
  cedr_throw_if(err != 0, "allgather of the root record returned " << err);
  cedr_k_check(cedr_hip_scatter_slots(data, stride_, nfields, root_slot_dev_.data(), 1,
                                      bcast_.data() + static_cast<size_t>(plan_.root_owner)*nfields, stream));
}

void TreeEngine::top_to_local (int32_t nfields, double* data, hipStream_t stream) {
  if (top_.empty()) return;
  cedr_k_check(cedr_hip_copy_slots(data, stride_, nfields, my_top_slots_.data(),
                                   my_root_slots_.data(),
                                   static_cast<int32_t>(plan_.my_root_slots.size()), stream));
}

}  // namespace engine
}  // namespace cedr
