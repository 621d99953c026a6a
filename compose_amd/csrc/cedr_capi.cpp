// Object-level C ABI (layer O of include/cedr_hip.h) over the C++ host classes.
// Nothing throws across this boundary: exceptions become -1 / NULL and a
// message retrievable with cedr_last_error().
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <string>

#include "../../include/cedr_hip.h"
#include "cedr_bfb_tree_allreduce.hpp"
#include "cedr_caas.hpp"
#include "cedr_engine.hpp"
#include "cedr_qlt.hpp"
#include "cedr_tree.hpp"
#include "cedr_util.hpp"

using namespace cedr;

namespace {
thread_local std::string g_err;

struct ParallelBox { mpi::Parallel::Ptr p; };
struct TreeBox { tree::FlatTree t; };
struct NsBox { tree::NodeSets::ConstPtr owned; const tree::NodeSets* ns; };
struct PlanBox { plan::Plan owned; const plan::Plan* p; int64_t stride; };
struct QltBox { std::unique_ptr<qlt::QLTHip> q; NsBox ns; PlanBox plan; };
struct BfbBox { std::shared_ptr<BfbTreeAllReducerHip> b; };
struct MailboxBox { std::shared_ptr<mpi::PeerMailbox> m; };

struct CbReducer : public caas::CAASHip::UserAllReducer {
  cedr_user_allreduce_fn fn; int nacc; void* user;
  CbReducer (cedr_user_allreduce_fn f, int n, void* u) : fn(f), nacc(n), user(u) {}
  int operator() (const mpi::Parallel&, Real* s, Real* r, int nlocal, int nfld, MPI_Op) const override {
    return fn(user, s, r, nlocal, nfld);
  }
  int n_accum_in_place () const override { return nacc; }
};
struct CaasBox { std::unique_ptr<caas::CAASHip> c; };
}  // namespace

#define CEDR_TRY try {
#define CEDR_CATCH(ret) } catch (const std::exception& e) { g_err = e.what(); return ret; } \
  catch (...) { g_err = "unknown exception"; return ret; }

extern "C" {

const char* cedr_last_error (void) { return g_err.c_str(); }

// ------------------------------------------------------------------ Parallel
void* cedr_parallel_create_self (void) {
  CEDR_TRY return new ParallelBox{mpi::make_parallel()}; CEDR_CATCH(nullptr)
}
int cedr_rccl_get_unique_id (void* id) { return mpi::rccl_get_unique_id(id); }
void* cedr_parallel_create_rccl (const void* id, int rank, int size) {
  CEDR_TRY return new ParallelBox{mpi::make_parallel_rccl(id, rank, size)}; CEDR_CATCH(nullptr)
}
void* cedr_parallel_create_rehearsal (void* inner, int pretended_size) {
  CEDR_TRY
  cedr_throw_if( ! inner, "cedr_parallel_create_rehearsal needs an inner Parallel");
  return new ParallelBox{mpi::make_parallel_rehearsal(static_cast<ParallelBox*>(inner)->p, pretended_size)};
  CEDR_CATCH(nullptr)
}
void* cedr_peer_mailbox_create (int size, size_t slot_doubles, void* handle_out) {
  CEDR_TRY
  static_assert(CEDR_PEER_HANDLE_BYTES == mpi::kPeerHandleBytes, "handle size");
  cedr_throw_if( ! handle_out, "cedr_peer_mailbox_create needs room for the handle");
  auto mb = mpi::peer_mailbox_create(size, slot_doubles);
  mpi::peer_mailbox_handle(*mb, handle_out);
  return new MailboxBox{mb};
  CEDR_CATCH(nullptr)
}
void cedr_peer_mailbox_destroy (void* mailbox) { delete static_cast<MailboxBox*>(mailbox); }
void* cedr_parallel_create_peer_flags (void* my_mailbox, const void* handles, int rank, int size) {
  CEDR_TRY
  cedr_throw_if( ! my_mailbox || ! handles, "cedr_parallel_create_peer_flags needs a mailbox and the ranks' handles");
  return new ParallelBox{mpi::make_parallel_peer_flags(static_cast<MailboxBox*>(my_mailbox)->m, handles, rank, size)};
  CEDR_CATCH(nullptr)
}
void* cedr_parallel_create_callbacks (int rank, int size, cedr_allgather_fn ag,
                                      cedr_allreduce_sum_fn ar, void* user) {
  CEDR_TRY
  return new ParallelBox{mpi::make_parallel_callbacks(rank, size, ag, ar, user)};
  CEDR_CATCH(nullptr)
}
void* cedr_parallel_create_callbacks_p2p (int rank, int size, cedr_allgather_fn ag,
                                          cedr_allreduce_sum_fn ar, cedr_exchange_fn xch, void* user) {
  CEDR_TRY
  return new ParallelBox{mpi::make_parallel_callbacks(rank, size, ag, ar, user, xch)};
  CEDR_CATCH(nullptr)
}
int cedr_parallel_exchange (void* p, int nops, const int* peers, double* const* bufs, const size_t* counts,
                            const int* sends, cedr_stream_t stream) {
  CEDR_TRY
  std::vector<mpi::Comm::P2P> ops(nops);
  for (int i = 0; i < nops; ++i) ops[i] = mpi::Comm::P2P{peers[i], bufs[i], counts[i], sends[i]};
  return static_cast<ParallelBox*>(p)->p->comm()->exchange(ops.data(), nops, static_cast<hipStream_t>(stream));
  CEDR_CATCH(-1)
}
int cedr_parallel_allgather (void* p, const double* send, double* recv, size_t count, cedr_stream_t stream) {
  CEDR_TRY
  return static_cast<ParallelBox*>(p)->p->comm()->allgather(send, recv, count, static_cast<hipStream_t>(stream));
  CEDR_CATCH(-1)
}
int cedr_parallel_allreduce_sum (void* p, const double* send, double* recv, size_t count, cedr_stream_t stream) {
  CEDR_TRY
  return static_cast<ParallelBox*>(p)->p->comm()->allreduce_sum(send, recv, count, static_cast<hipStream_t>(stream));
  CEDR_CATCH(-1)
}
int cedr_parallel_rank (void* p) { return static_cast<ParallelBox*>(p)->p->rank(); }
int cedr_parallel_size (void* p) { return static_cast<ParallelBox*>(p)->p->size(); }
void cedr_parallel_free (void* p) { delete static_cast<ParallelBox*>(p); }

// ---------------------------------------------------------------------- trees
void* cedr_tree_create (int nnodes, int root, const int* kid0, const int* kid1, const int* rank,
                        const long* cellidx) {
  return cedr_tree_create_levels(nnodes, root, kid0, kid1, rank, cellidx, nullptr);
}

void* cedr_tree_prune (void* tree, int my_rank) {
  CEDR_TRY
  return new TreeBox{tree::prune_for_rank(static_cast<TreeBox*>(tree)->t, my_rank)};
  CEDR_CATCH(nullptr)
}
int cedr_tree_nnodes (void* tree) { return static_cast<TreeBox*>(tree)->t.size(); }

void* cedr_tree_create_levels (int nnodes, int root, const int* kid0, const int* kid1, const int* rank,
                               const long* cellidx, const int* level) {
  CEDR_TRY
  cedr_throw_if(nnodes < 1 || root < 0 || root >= nnodes, "bad tree size or root");
  TreeBox* tb = new TreeBox();
  tree::FlatTree& t = tb->t;
  struct Item { int src, fparent, which; };
  std::vector<Item> stack(1, Item{root, -1, 0});
  int visited = 0;
  while ( ! stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    cedr_throw_if(++visited > nnodes, "tree arrays contain a cycle");
    const Int me = t.add(it.fparent);
    if (it.fparent >= 0) (it.which == 0 ? t.kid0 : t.kid1)[it.fparent] = me;
    t.rank[me] = rank[it.src];
    t.cellidx[me] = cellidx[it.src];
    if (level) t.level[me] = level[it.src];
    int k0 = kid0[it.src], k1 = kid1[it.src];
    if (level && level[it.src] > 0 && k0 < 0 && k1 >= 0) { k0 = k1; k1 = -1; }  // partial: the one kid is kid 0
    cedr_throw_if((k0 < 0) != (k1 < 0) && ! (level && level[it.src] > 0),
                  "a node has exactly one kid; only binary trees are supported");
    cedr_throw_if(k0 >= nnodes || k1 >= nnodes, "kid index out of range");
    if (k0 >= 0 && k1 >= 0) {
      t.nkids[me] = 2;
      stack.push_back(Item{k1, me, 1});
      stack.push_back(Item{k0, me, 0});
    } else if (k0 >= 0) {
      t.nkids[me] = 1;
      stack.push_back(Item{k0, me, 0});
    }
  }
  return tb;
  CEDR_CATCH(nullptr)
}

void* cedr_tree_make_1d (int ncells, int nranks, int pseudorandom, int imbalanced) {
  CEDR_TRY
  typedef tree::oned::Mesh Mesh;
  const Mesh m(ncells, nranks, pseudorandom ? Mesh::ParallelDecomp::pseudorandom
                                            : Mesh::ParallelDecomp::contiguous);
  return new TreeBox{tree::oned::make_flat_tree(m, imbalanced != 0)};
  CEDR_CATCH(nullptr)
}
void cedr_tree_free (void* t) { delete static_cast<TreeBox*>(t); }

// ------------------------------------------------------------------- NodeSets
void* cedr_tree_analyze (void* tree, int ncells, int my_rank) {
  CEDR_TRY
  NsBox* b = new NsBox();
  b->owned = tree::analyze(my_rank, ncells, static_cast<TreeBox*>(tree)->t);
  b->ns = b->owned.get();
  return b;
  CEDR_CATCH(nullptr)
}
void cedr_ns_free (void* ns) { delete static_cast<NsBox*>(ns); }
#define NS(h) (*static_cast<NsBox*>(h)->ns)
int cedr_ns_nslots (void* h) { return NS(h).nslots; }
int cedr_ns_nnode (void* h) { return NS(h).nnode(); }
int cedr_ns_nlevels (void* h) { return static_cast<int>(NS(h).levels.size()); }
int cedr_ns_level_size (void* h, int il) { return static_cast<int>(NS(h).levels[il].nodes.size()); }
int cedr_ns_level_nme (void* h, int il) { return static_cast<int>(NS(h).levels[il].me.size()); }
int cedr_ns_level_nkids (void* h, int il) { return static_cast<int>(NS(h).levels[il].kids.size()); }
void cedr_ns_nodes (void* h, int* out) {
  const tree::NodeSets& ns = NS(h);
  for (Int i = 0; i < ns.nnode(); ++i) {
    const auto n = ns.node_h(i);
    int* o = out + 7*i;
    o[0] = n->rank; o[1] = n->id; o[2] = n->parent; o[3] = n->nkids;
    o[4] = n->kids[0]; o[5] = n->kids[1]; o[6] = n->offset;
  }
}
void cedr_ns_level_nodes (void* h, int il, int* out) {
  const auto& v = NS(h).levels[il].nodes;
  std::copy(v.begin(), v.end(), out);
}
void cedr_ns_level_msgs (void* h, int il, int which, int* out) {
  const auto& lvl = NS(h).levels[il];
  const auto& v = which == 0 ? lvl.me : lvl.kids;
  for (size_t i = 0; i < v.size(); ++i) {
    out[3*i] = v[i].rank; out[3*i+1] = v[i].offset; out[3*i+2] = v[i].size;
  }
}
int cedr_ns_check_comm (void* h) { return tree::check_comm(NS(h)); }

// ------------------------------------------------------------------------ QLT
#define Q(h) (*static_cast<QltBox*>(h)->q)
void* cedr_qlt_create (void* parallel, int ncells, void* tree, int prefer) {
  CEDR_TRY
  CDR::Options o;
  o.prefer_numerical_mass_conservation_to_numerical_bounds = prefer != 0;
  QltBox* b = new QltBox();
  b->q.reset(new qlt::QLTHip(static_cast<ParallelBox*>(parallel)->p, ncells,
                             static_cast<TreeBox*>(tree)->t, o));
  b->ns.ns = &b->q->nodesets();
  return b;
  CEDR_CATCH(nullptr)
}
void cedr_qlt_free (void* h) { delete static_cast<QltBox*>(h); }
int cedr_qlt_declare_tracer (void* h, int pt, int rhomidx) {
  CEDR_TRY Q(h).declare_tracer(pt, rhomidx); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_end_tracer_declarations (void* h) {
  CEDR_TRY Q(h).end_tracer_declarations(); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_get_buffers_sizes (void* h, size_t* b1, size_t* b2) {
  CEDR_TRY Q(h).get_buffers_sizes(*b1, *b2); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_get_buffers_sizes_tight (void* h, size_t* b1, size_t* b2) {
  CEDR_TRY Q(h).get_buffers_sizes_tight(*b1, *b2); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_set_buffers (void* h, double* b1, double* b2) {
  CEDR_TRY Q(h).set_buffers(b1, b2); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_finish_setup (void* h) { CEDR_TRY Q(h).finish_setup(); return 0; CEDR_CATCH(-1) }
int cedr_qlt_get_problem_type (void* h, int ti) {
  CEDR_TRY return Q(h).get_problem_type(ti); CEDR_CATCH(-1)
}
int cedr_qlt_get_num_tracers (void* h) { return Q(h).get_num_tracers(); }
int cedr_qlt_nlclcells (void* h) { return Q(h).nlclcells(); }
int cedr_qlt_get_owned_glblcells (void* h, long* gcis) {
  CEDR_TRY
  std::vector<Long> g;
  Q(h).get_owned_glblcells(g);
  std::copy(g.begin(), g.end(), gcis);
  return 0;
  CEDR_CATCH(-1)
}
int cedr_qlt_gci2lci (void* h, int gci) { CEDR_TRY return Q(h).gci2lci(gci); CEDR_CATCH(-1) }
int cedr_qlt_run (void* h) { CEDR_TRY Q(h).run(); return 0; CEDR_CATCH(-1) }
int cedr_qlt_set_stream (void* h, cedr_stream_t s) { Q(h).set_stream(s); return 0; }
void* cedr_qlt_sweep_down_event (void* h) { CEDR_TRY return Q(h).sweep_down_event(); CEDR_CATCH(nullptr) }
const char* cedr_qlt_placement_report (void* h) { return Q(h).placement_report().c_str(); }
void* cedr_qlt_sweep_mid_event (void* h) { CEDR_TRY return Q(h).sweep_mid_event(); CEDR_CATCH(nullptr) }
int cedr_stream_wait_event (cedr_stream_t s, void* ev) {
  return static_cast<int>(hipStreamWaitEvent(static_cast<hipStream_t>(s), static_cast<hipEvent_t>(ev), 0));
}
int cedr_qlt_get_device_op (void* h, cedr_qlt_device_op* out) {
  const auto& o = Q(h).get_device_op();
  out->l2r_data = o.l2r_data; out->r2l_data = o.r2l_data; out->stride = o.stride;
  out->trcr2prob = o.trcr2prob; out->trcr2bl2r = o.trcr2bl2r; out->trcr2br2l = o.trcr2br2l;
  return 0;
}
int cedr_qlt_set_rhom_bulk (void* h, const double* rhom) {
  CEDR_TRY Q(h).set_rhom_bulk(rhom); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_set_Qm_bulk (void* h, const double* Qm, const double* lo, const double* hi,
                          const double* prev) {
  CEDR_TRY Q(h).set_Qm_bulk(Qm, lo, hi, prev); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_get_Qm_bulk (void* h, double* out) {
  CEDR_TRY Q(h).get_Qm_bulk(out); return 0; CEDR_CATCH(-1)
}
int cedr_qlt_metadata (void* h, int* trcr2prob, int* bidx2trcr, int* trcr2bidx, int* trcr2bl2r,
                       int* trcr2br2l, int* prob2trcrptr, int* prob2bl2r, int* prob2br2l) {
  const auto& m = Q(h).metadata();
  const size_t nt = m.trcr2prob.size();
  for (size_t i = 0; i < nt; ++i) {
    trcr2prob[i] = m.trcr2prob[i]; bidx2trcr[i] = m.bidx2trcr[i]; trcr2bidx[i] = m.trcr2bidx[i];
    trcr2bl2r[i] = m.trcr2bl2r[i]; trcr2br2l[i] = m.trcr2br2l[i];
  }
  for (int i = 0; i < 7; ++i) {
    prob2trcrptr[i] = m.prob2trcrptr[i]; prob2bl2r[i] = m.prob2bl2r[i]; prob2br2l[i] = m.prob2br2l[i];
  }
  return 0;
}
int cedr_qlt_bind_arrays (void* h, const double* rhom, const double* Qm, const double* Qm_min,
                          const double* Qm_max, const double* Qm_prev, double* Qm_out) {
  CEDR_TRY
  Q(h).bind_arrays(rhom, Qm, Qm_min, Qm_max, Qm_prev, Qm_out);
  return 0;
  CEDR_CATCH(1)
}
void* cedr_qlt_nodesets (void* h) { return &static_cast<QltBox*>(h)->ns; }

// ----------------------------------------------------------------------- plan
#define PL(h) (*static_cast<PlanBox*>(h)->p)

void* cedr_plan_create (void* tree, int ncells, int my_rank, int nranks, int max_tile_leaves) {
  CEDR_TRY
  const tree::FlatTree& t = static_cast<TreeBox*>(tree)->t;
  const auto ns = tree::analyze(my_rank, ncells, t);
  cedr_throw_if(ns->levels.empty() || ns->levels[0].nodes.empty(), "0 cells on this rank");
  util::IdMap g2l;
  g2l.reserve(ns->levels[0].nodes.size());
  for (const Int idx : ns->levels[0].nodes) g2l.add(ns->node_h(idx)->id, ns->node_h(idx)->offset);
  g2l.finish();
  std::vector<int32_t> leaf_slot(t.size(), -1);
  for (Int n = 0; n < t.size(); ++n)
    if (t.nkids[n] == 0 && t.rank[n] == my_rank) {
      leaf_slot[n] = g2l.find(static_cast<Int>(t.cellidx[n]));
      cedr_throw_if(leaf_slot[n] < 0, "owned leaf missing from NodeSets");
    }
  plan::Options o;
  if (max_tile_leaves > 0) o.max_tile_leaves = max_tile_leaves;
  if (max_tile_leaves < 0) { o.micro = false; if (max_tile_leaves < -1) o.max_tile_leaves = -max_tile_leaves; }
  if (const char* e = std::getenv("CEDR_HIP_MICRO_CUT")) o.micro_cut = std::atoi(e);
  if (const char* e = std::getenv("CEDR_HIP_MICRO_CUT2")) o.micro_cut2 = std::atoi(e);
  if (const char* e = std::getenv("CEDR_HIP_MICRO")) o.micro_all = std::atoi(e) >= 2;
  if (const char* e = std::getenv("CEDR_HIP_EXCHANGE")) o.exchange = std::atoi(e);
  // fused top tile (plan::Stage::top): 0 = off (every stage its own launch, as before round 3)
  if (const char* e = std::getenv("CEDR_HIP_TOP")) o.fused_top = std::atoi(e) != 0;
  if (const char* e = std::getenv("CEDR_HIP_TOP_MAX")) o.top_max_inputs = std::atoi(e);
  PlanBox* b = new PlanBox();
  b->owned = plan::build(t, my_rank, nranks, leaf_slot, o);
  b->p = &b->owned;
  b->stride = (static_cast<int64_t>(b->owned.nslots) + 2 + 15)/16*16;
  return b;
  CEDR_CATCH(nullptr)
}
void cedr_plan_free (void* h) { delete static_cast<PlanBox*>(h); }
void* cedr_qlt_plan (void* h) {
  QltBox* b = static_cast<QltBox*>(h);
  b->plan.p = &b->q->tree_engine().get_plan();
  b->plan.stride = b->q->tree_engine().stride();
  return &b->plan;
}

int cedr_plan_summary (void* h, int64_t* s) {
  const plan::Plan& p = PL(h);
  s[0] = p.nlcl; s[1] = p.nslots; s[2] = static_cast<int64_t>(p.local.size());
  s[3] = static_cast<int64_t>(p.top.size()); s[4] = p.root_slot; s[5] = p.nb_max;
  s[6] = static_cast<int64_t>(p.my_root_slots.size()); s[7] = static_cast<PlanBox*>(h)->stride;
  return 0;
}
static const plan::Stage* get_stage (void* h, int top, int idx) {
  const plan::Plan& p = PL(h);
  if (top == 2) return (idx == 0 && ! p.apex.tiles.empty()) ? &p.apex : nullptr;  // the apex tile
  const auto& v = top ? p.top : p.local;
  if (idx < 0 || idx >= static_cast<int>(v.size())) return nullptr;
  return &v[idx];
}
int cedr_plan_stage_sizes (void* h, int top, int idx, int* sz) {
  const plan::Stage* st = get_stage(h, top, idx);
  if ( ! st) { g_err = "no such stage"; return -1; }
  sz[0] = static_cast<int>(st->tiles.size()); sz[1] = st->max_nodes;
  sz[2] = static_cast<int>(st->kids.size()/2); sz[3] = static_cast<int>(st->lvlptr.size());
  sz[4] = static_cast<int>(st->micro.size());
  sz[5] = static_cast<int>(st->outs.size());
  return 0;
}
int cedr_plan_stage_is_top (void* h, int top, int idx) {
  const plan::Stage* st = get_stage(h, top, idx);
  if ( ! st) { g_err = "no such stage"; return -1; }
  return st->top ? 1 : 0;
}
int cedr_plan_stage_micro (void* h, int top, int idx, uint16_t* micro, uint16_t* outs) {
  const plan::Stage* st = get_stage(h, top, idx);
  if ( ! st) { g_err = "no such stage"; return -1; }
  std::copy(st->micro.begin(), st->micro.end(), micro);
  std::copy(st->outs.begin(), st->outs.end(), outs);
  return 0;
}
int cedr_plan_stage (void* h, int top, int idx, int* tiles, uint16_t* kids, int* lvlptr) {
  const plan::Stage* st = get_stage(h, top, idx);
  if ( ! st) { g_err = "no such stage"; return -1; }
  std::memcpy(tiles, st->tiles.data(), st->tiles.size()*sizeof(plan::Tile));
  std::copy(st->kids.begin(), st->kids.end(), kids);
  std::copy(st->lvlptr.begin(), st->lvlptr.end(), lvlptr);
  return 0;
}
int cedr_plan_exchange (void* h, int* my_root_slots, int* my_top_slots, int* gather_slots) {
  const plan::Plan& p = PL(h);
  std::copy(p.my_root_slots.begin(), p.my_root_slots.end(), my_root_slots);
  std::copy(p.my_top_slots.begin(), p.my_top_slots.end(), my_top_slots);
  std::copy(p.gather_slots.begin(), p.gather_slots.end(), gather_slots);
  return 0;
}

int cedr_plan_levels (void* h, int* info3) {
  const plan::Plan& p = PL(h);
  info3[0] = p.level_mode ? 1 : 0; info3[1] = static_cast<int>(p.steps.size()); info3[2] = p.root_owner;
  return 0;
}
int cedr_plan_level_sizes (void* h, int i, int* sz) {
  const plan::Plan& p = PL(h);
  if (i < 0 || i >= static_cast<int>(p.steps.size())) { g_err = "no such level step"; return -1; }
  const plan::LevelStep& st = p.steps[i];
  sz[0] = st.level; sz[1] = static_cast<int>(st.nodes.size()/3);
  sz[2] = static_cast<int>(st.out_slots.size()); sz[3] = static_cast<int>(st.in_slots.size());
  sz[4] = static_cast<int>(st.out_msgs.size()); sz[5] = static_cast<int>(st.in_msgs.size());
  return 0;
}
int cedr_plan_level (void* h, int i, int* nodes, int* out_slots, int* in_slots, int* out_msgs, int* in_msgs) {
  const plan::Plan& p = PL(h);
  if (i < 0 || i >= static_cast<int>(p.steps.size())) { g_err = "no such level step"; return -1; }
  const plan::LevelStep& st = p.steps[i];
  std::copy(st.nodes.begin(), st.nodes.end(), nodes);
  std::copy(st.out_slots.begin(), st.out_slots.end(), out_slots);
  std::copy(st.in_slots.begin(), st.in_slots.end(), in_slots);
  for (size_t k = 0; k < st.out_msgs.size(); ++k) {
    out_msgs[3*k] = st.out_msgs[k].peer; out_msgs[3*k+1] = st.out_msgs[k].first; out_msgs[3*k+2] = st.out_msgs[k].count;
  }
  for (size_t k = 0; k < st.in_msgs.size(); ++k) {
    in_msgs[3*k] = st.in_msgs[k].peer; in_msgs[3*k+1] = st.in_msgs[k].first; in_msgs[3*k+2] = st.in_msgs[k].count;
  }
  return 0;
}

// ----------------------------------------------------------------------- CAAS
#define CA(h) (*static_cast<CaasBox*>(h)->c)
void* cedr_caas_create (void* parallel, int nlclcells, cedr_user_allreduce_fn reducer,
                        int nacc, void* user) {
  CEDR_TRY
  caas::CAASHip::UserAllReducer::Ptr r;
  if (reducer) r = std::make_shared<CbReducer>(reducer, nacc < 1 ? 1 : nacc, user);
  CaasBox* b = new CaasBox();
  b->c.reset(new caas::CAASHip(static_cast<ParallelBox*>(parallel)->p, nlclcells, r));
  return b;
  CEDR_CATCH(nullptr)
}
void* cedr_caas_create_bfb (void* parallel, int nlclcells, void* bfb, int nacc) {
  CEDR_TRY
  cedr_throw_if( ! bfb, "cedr_caas_create_bfb needs a BfbTreeAllReducer handle");
  caas::CAASHip::UserAllReducer::Ptr r =
    std::make_shared<caas::BfbUserAllReducer>(static_cast<BfbBox*>(bfb)->b, nacc);
  CaasBox* b = new CaasBox();
  b->c.reset(new caas::CAASHip(static_cast<ParallelBox*>(parallel)->p, nlclcells, r));
  return b;
  CEDR_CATCH(nullptr)
}
int cedr_caas_fused_with_bfb (void* h) { return CA(h).fused_with_bfb_reducer() ? 1 : 0; }
void cedr_caas_free (void* h) { delete static_cast<CaasBox*>(h); }
int cedr_caas_declare_tracer (void* h, int pt, int rhomidx) {
  CEDR_TRY CA(h).declare_tracer(pt, rhomidx); return 0; CEDR_CATCH(-1)
}
int cedr_caas_end_tracer_declarations (void* h) {
  CEDR_TRY CA(h).end_tracer_declarations(); return 0; CEDR_CATCH(-1)
}
int cedr_caas_get_buffers_sizes (void* h, size_t* b1, size_t* b2) {
  CEDR_TRY CA(h).get_buffers_sizes(*b1, *b2); return 0; CEDR_CATCH(-1)
}
int cedr_caas_set_buffers (void* h, double* b1, double* b2) {
  CEDR_TRY CA(h).set_buffers(b1, b2); return 0; CEDR_CATCH(-1)
}
int cedr_caas_finish_setup (void* h) { CEDR_TRY CA(h).finish_setup(); return 0; CEDR_CATCH(-1) }
int cedr_caas_get_problem_type (void* h, int ti) {
  CEDR_TRY return CA(h).get_problem_type(ti); CEDR_CATCH(-1)
}
int cedr_caas_get_num_tracers (void* h) { return CA(h).get_num_tracers(); }
int cedr_caas_run (void* h) { CEDR_TRY CA(h).run(); return 0; CEDR_CATCH(-1) }
int cedr_caas_set_stream (void* h, cedr_stream_t s) { CA(h).set_stream(s); return 0; }
int cedr_caas_set_Qm_source (void* h, const double* src) { CA(h).set_Qm_source(src); return 0; }
int cedr_caas_bind_arrays (void* h, double* Qm, const double* Qm_min, const double* Qm_max,
                           const double* Qm_prev) {
  CEDR_TRY
  CA(h).bind_arrays(Qm, Qm_min, Qm_max, Qm_prev);
  return 0;
  CEDR_CATCH(1)
}
int cedr_caas_get_device_op (void* h, cedr_caas_device_op* out) {
  const auto& o = CA(h).get_device_op();
  out->d = o.d_; out->nlclcells = o.nlclcells_; out->nt = o.nt_;
  out->need_conserve = o.need_conserve_;
  return 0;
}
int cedr_caas_get_recv (void* h, const double** recv) { *recv = CA(h).recv_data(); return 0; }

// ------------------------------------------------------------------------ BFB
void* cedr_bfb_create (void* parallel, void* tree, int nleaf, int nfield) {
  CEDR_TRY
  BfbBox* b = new BfbBox();
  b->b.reset(new BfbTreeAllReducerHip(static_cast<ParallelBox*>(parallel)->p,
                                      static_cast<TreeBox*>(tree)->t, nleaf, nfield));
  return b;
  CEDR_CATCH(nullptr)
}
void cedr_bfb_free (void* h) { delete static_cast<BfbBox*>(h); }
int cedr_bfb_nlocal (void* h) { return static_cast<BfbBox*>(h)->b->nlocal(); }
int cedr_bfb_allreduce (void* h, const double* send, double* recv, int transpose) {
  CEDR_TRY static_cast<BfbBox*>(h)->b->allreduce(send, recv, transpose != 0); return 0; CEDR_CATCH(-1)
}
int cedr_bfb_set_stream (void* h, cedr_stream_t stream) {
  CEDR_TRY
  static_cast<BfbBox*>(h)->b->set_stream(static_cast<hipStream_t>(stream));
  return 0;
  CEDR_CATCH(1)
}

// -------------------------------------------------------------- device helpers
int cedr_device_count (void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
void* cedr_malloc (size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) { g_err = "hipMalloc failed"; return nullptr; }
  return p;
}
void cedr_free (void* p) { (void) hipFree(p); }
int cedr_memcpy_h2d (void* d, const void* s, size_t n) {
  return static_cast<int>(hipMemcpy(d, s, n, hipMemcpyHostToDevice));
}
int cedr_memcpy_d2h (void* d, const void* s, size_t n) {
  return static_cast<int>(hipMemcpy(d, s, n, hipMemcpyDeviceToHost));
}
int cedr_memcpy_d2d_async (void* d, const void* s, size_t n, cedr_stream_t st) {
  return static_cast<int>(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st));
}
int cedr_device_synchronize (void) { return static_cast<int>(hipDeviceSynchronize()); }
int cedr_stream_synchronize (cedr_stream_t st) { return static_cast<int>(hipStreamSynchronize(st)); }
int cedr_memcpy_h2d_async (void* d, const void* s, size_t n, cedr_stream_t st) {
  return static_cast<int>(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st));
}
int cedr_memcpy_d2h_async (void* d, const void* s, size_t n, cedr_stream_t st) {
  return static_cast<int>(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st));
}
int cedr_stream_create_cu_mask (const uint32_t* mask, int32_t nwords, cedr_stream_t* st) {
  if ( ! mask || nwords < 1 || ! st) return static_cast<int>(hipErrorInvalidValue);
  return static_cast<int>(hipExtStreamCreateWithCUMask(st, static_cast<uint32_t>(nwords), mask));
}
int cedr_stream_destroy (cedr_stream_t st) { return static_cast<int>(hipStreamDestroy(st)); }

}  // extern "C"
