// Host-side tree analysis for the MI355X CEDR build. Integer results match
// reference cedr/cedr_tree.cpp (cited per function); the code is iterative and
// array based so that a 6M-cell tree neither recurses 6M shared_ptr frames nor
// overflows the stack on deep (imbalanced) trees.
#include "cedr_tree.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <sstream>
#include <utility>

#include "cedr_util.hpp"

namespace cedr {
namespace util {
double PhaseTimer::now () {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
PhaseTimer::PhaseTimer () : on_(std::getenv("CEDR_SETUP_TIMES") != nullptr), t0_(on_ ? now() : 0) {}
void PhaseTimer::lap (const char* label) {
  if ( ! on_) return;
  const double t = now();
  std::fprintf(stderr, "  [setup] %-28s %.3f s\n", label, t - t0_);
  t0_ = t;
}
}  // namespace util
namespace tree {

Int FlatTree::add (Int parent_idx) {
  parent.push_back(parent_idx); rank.push_back(-1); nkids.push_back(0);
  kid0.push_back(-1); kid1.push_back(-1); level.push_back(-1); cellidx.push_back(-1);
  return size() - 1;
}

FlatTree flatten (const Node::Ptr& tree) {
  cedr_throw_if( ! tree, "null tree");
  FlatTree ft;
  // (caller node, index of its flat parent, which kid of that parent)
  struct Item { const Node* n; Int fparent; Int which; };
  std::vector<Item> stack;
  stack.push_back(Item{tree.get(), -1, 0});
  while ( ! stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const Int me = ft.add(it.fparent);
    if (it.fparent >= 0) (it.which == 0 ? ft.kid0 : ft.kid1)[it.fparent] = me;
    ft.rank[me] = it.n->rank;
    ft.cellidx[me] = it.n->cellidx;
    ft.nkids[me] = it.n->nkids;
    ft.level[me] = it.n->level;
    // Push kid 1 first so kid 0 is numbered (and later visited) first.
    for (Int i = it.n->nkids - 1; i >= 0; --i) {
      cedr_throw_if( ! it.n->kids[i], "tree node has nkids > 0 but a null kid");
      stack.push_back(Item{it.n->kids[i].get(), me, i});
    }
  }
  return ft;
}

std::vector<Int> post_order (const FlatTree& t) {
  std::vector<Int> order;
  order.reserve(t.size());
  std::vector<std::pair<Int,Int> > stack;  // (node, next kid to descend into)
  stack.push_back(std::make_pair(0, 0));
  while ( ! stack.empty()) {
    std::pair<Int,Int>& top = stack.back();
    if (top.second < t.nkids[top.first]) {
      const Int k = t.kid(top.first, top.second++);
      stack.push_back(std::make_pair(k, 0));
    } else {
      order.push_back(top.first);
      stack.pop_back();
    }
  }
  return order;
}

FlatTree prune_for_rank (const FlatTree& full, const Int me) {
  const Int nn = full.size();
  cedr_throw_if(nn == 0, "empty tree");
  std::vector<Int> rank(full.rank), lvl(nn, 0);
  std::vector<char> keep(nn, 0);
  const std::vector<Int> order = post_order(full);
  for (const Int n : order) {
    Int h = -1;
    for (Int i = 0; i < full.nkids[n]; ++i) h = std::max(h, lvl[full.kid(n, i)]);
    lvl[n] = full.level[n] >= 0 ? full.level[n] : h + 1;
    if (rank[n] < 0 && full.nkids[n] > 0) rank[n] = rank[full.kid0[n]];
    // mine, or above something of mine
    if (rank[n] == me) keep[n] = 1;
    for (Int i = 0; i < full.nkids[n]; ++i) if (keep[full.kid(n, i)]) keep[n] = 1;
  }
  for (Int n = 0; n < nn; ++n)  // kids of my nodes (kept so far means: mine or an ancestor)
    if (rank[n] == me)
      for (Int i = 0; i < full.nkids[n]; ++i) if ( ! keep[full.kid(n, i)]) keep[full.kid(n, i)] = 2;
  keep[0] = keep[0] ? keep[0] : 1;
  FlatTree out;
  std::vector<std::pair<Int,Int> > stack(1, std::make_pair(0, -1));  // (full node, parent in out)
  while ( ! stack.empty()) {
    const std::pair<Int,Int> it = stack.back();
    stack.pop_back();
    const Int n = it.first, o = out.add(it.second);
    if (it.second >= 0) {
      if (out.kid0[it.second] < 0) out.kid0[it.second] = o; else out.kid1[it.second] = o;
      ++out.nkids[it.second];
    }
    out.rank[o] = rank[n];
    out.level[o] = lvl[n];
    out.cellidx[o] = full.nkids[n] == 0 ? full.cellidx[n] : -1;
    if (keep[n] == 2 && rank[n] != me) continue;  // somebody else's kid of mine: nothing below it matters
    for (Int i = full.nkids[n] - 1; i >= 0; --i)
      if (keep[full.kid(n, i)]) stack.push_back(std::make_pair(full.kid(n, i), o));
  }
  return out;
}

void NodeSets::print (std::ostream& os) const {  // reference cedr_tree.cpp:8-34
  std::stringstream ss;
  if (levels.empty()) return;
  const Int myrank = node_h(levels[0].nodes[0])->rank;
  ss << "pid " << myrank << ": #levels " << levels.size();
  for (size_t i = 0; i < levels.size(); ++i) {
    const auto& lvl = levels[i];
    ss << "\n  " << i << ": " << lvl.nodes.size();
    std::set<Int> ps, ks;
    for (size_t j = 0; j < lvl.nodes.size(); ++j) {
      const auto n = node_h(lvl.nodes[j]);
      for (Int k = 0; k < n->nkids; ++k)
        if (node_h(n->kids[k])->rank != myrank) ks.insert(node_h(n->kids[k])->rank);
      if (n->parent >= 0 && node_h(n->parent)->rank != myrank) ps.insert(node_h(n->parent)->rank);
    }
    ss << " |";
    for (const auto& e : ks) ss << " " << e;
    if ( ! lvl.kids.empty()) ss << " (" << lvl.kids.size() << ") |";
    for (const auto& e : ps) ss << " " << e;
    if ( ! lvl.me.empty()) ss << " (" << lvl.me.size() << ")";
  }
  ss << "\n";
  os << ss.str();
}

namespace {
typedef std::pair<Int,Int> RankNode;  // (rank, NodeSets node index)

// reference cedr_tree.cpp:148-180. The reference sorts pair<rank, Node*> with
// the default pair order, so ties in rank fall back to the node's address,
// i.e. its allocation index; sorting (rank, index) pairs is the same order.
void init_offsets (const Int my_rank, NodeSets& ns, std::vector<RankNode>& rns,
                   std::vector<NodeSets::Level::MPIMetaData>& mmds, Int& offset) {
  for (auto& rn : rns) if (rn.first == my_rank) rn.first = -1;
  // (a rank's own nodes are listed in allocation order, i.e. already sorted: on one rank, and for
  // every big level of many, the O(n) check replaces the sort)
  if ( ! std::is_sorted(rns.begin(), rns.end())) std::sort(rns.begin(), rns.end());
  Int prev_rank = -1;
  for (const auto& rn : rns) {
    NodeSets::Node* const nd = ns.node_h(rn.second);
    if (rn.first == -1) {
      if (nd->offset == -1) nd->offset = offset++;
      continue;
    }
    if (rn.first != prev_rank) {
      prev_rank = rn.first;
      NodeSets::Level::MPIMetaData mmd;
      mmd.rank = rn.first; mmd.offset = offset; mmd.size = 0;
      mmds.push_back(mmd);
    }
    ++mmds.back().size;
    nd->offset = offset++;
  }
}

// reference cedr_tree.cpp:185-213
void init_comm (const Int my_rank, NodeSets& ns) {
  ns.nslots = 0;
  std::vector<RankNode> me, kids;
  for (auto& lvl : ns.levels) {
    me.clear(); kids.clear();
    for (const Int idx : lvl.nodes) {
      const NodeSets::Node* const n = ns.node_h(idx);
      me.push_back(RankNode(n->parent >= 0 ? ns.node_h(n->parent)->rank : my_rank, idx));
      for (Int k = 0; k < n->nkids; ++k)
        kids.push_back(RankNode(ns.node_h(n->kids[k])->rank, n->kids[k]));
    }
    init_offsets(my_rank, ns, me, lvl.me, ns.nslots);
    init_offsets(my_rank, ns, kids, lvl.kids, ns.nslots);
  }
}
}  // namespace

NodeSets::ConstPtr analyze (const Int my_rank, const Int& ncells, FlatTree t) {
  cedr_throw_if(t.size() == 0, "empty tree");
  cedr_throw_if(t.parent[0] != -1, "tree root has a parent");
  util::PhaseTimer pt;
  const auto nodesets = std::make_shared<NodeSets>();
  NodeSets& ns = *nodesets;
  const std::vector<Int> order = post_order(t);
  const Int nn = t.size();
  pt.lap("analyze: post_order");

  // reference cedr_tree.cpp:37-53 (init_tree): depth, ranks of internal nodes,
  // ids of internal nodes in post-order starting at ncells.
  std::vector<Int> height(nn, 1);
  Int id = ncells;
  for (const Int n : order) {
    if (t.nkids[n]) {
      Int h = 0;
      for (Int i = 0; i < t.nkids[n]; ++i) h = std::max(h, height[t.kid(n, i)]);
      height[n] = h + 1;
      if (t.rank[n] < 0 && t.nkids[n] > 0) t.rank[n] = t.rank[t.kid0[n]];
      t.cellidx[n] = id++;
    } else {
      cedr_throw_if(t.rank[n] == my_rank && (t.cellidx[n] < 0 || t.cellidx[n] >= id),
                    "cellidx is " << t.cellidx[n] << " but should be between " << 0 <<
                    " and " << id);
    }
  }
  Int depth = height[0];
  if (t.level[0] >= 0) depth = t.level[0] + 1;
  ns.levels.resize(depth);
  pt.lap("analyze: init_tree");

  // reference cedr_tree.cpp:55-129 (level_schedule_and_collect), post-order.
  {  // one allocation for the node pool: owned nodes + one placeholder per remote kid at most
    size_t nown = 0;
    for (Int n = 0; n < nn; ++n) if (t.rank[n] == my_rank) ++nown;
    ns.reserve(std::min<size_t>(static_cast<size_t>(nn), 3*nown + 16));
  }
  std::vector<Int> lvl(nn, 0), reserved(nn, -1);
  std::vector<char> need_parent(nn, 0);
  for (const Int n : order) {
    Int level = -1;
    bool make_ns_node = false;
    for (Int i = 0; i < t.nkids[n]; ++i) {
      const Int k = t.kid(n, i);
      level = std::max(level, lvl[k]);
      if (need_parent[k]) make_ns_node = true;
    }
    ++level;
    if (t.level[n] >= 0) {
      level = t.level[n];
      cedr_throw_if(level >= depth, "user-provided level exceeds the root's level");
    }
    lvl[n] = level;
    cedr_throw_if(t.rank[n] == -1, "a tree node has no rank");
    const bool owned = t.rank[n] == my_rank;
    need_parent[n] = owned;
    if ( ! (owned || make_ns_node)) continue;
    const Int idx = ns.alloc();
    if (owned) ns.levels[level].nodes.push_back(idx);
    reserved[n] = idx;
    ns.node_h(idx)->rank = t.rank[n];
    if (t.nkids[n] == 0) ns.node_h(idx)->id = static_cast<Int>(t.cellidx[n]);
    if (owned) {
      ns.node_h(idx)->nkids = t.nkids[n];
      for (Int i = 0; i < t.nkids[n]; ++i) {
        const Int k = t.kid(n, i);
        if (reserved[k] == -1) {
          // A kid on another rank: keep a slot for what it sends us.
          const Int kidx = ns.alloc();
          reserved[k] = kidx;
          ns.node_h(idx)->kids[i] = kidx;
          ns.node_h(kidx)->rank = t.rank[k];
          ns.node_h(kidx)->id = static_cast<Int>(t.cellidx[k]);
          ns.node_h(kidx)->nkids = 0;
        } else {
          ns.node_h(idx)->kids[i] = reserved[k];
          ns.node_h(reserved[k])->parent = idx;
        }
      }
    } else {
      // Not ours, but a parent of something of ours: record it as the partner.
      NodeSets::Node* const nd = ns.node_h(idx);
      nd->nkids = 0;
      for (Int i = 0; i < t.nkids[n]; ++i) {
        const Int k = t.kid(n, i);
        if (reserved[k] >= 0 && t.rank[k] == my_rank) {
          nd->kids[nd->nkids++] = reserved[k];
          ns.node_h(reserved[k])->parent = idx;
        }
      }
    }
  }

  pt.lap("analyze: level schedule");
  {  // reference cedr_tree.cpp:138-144 (consolidate)
    std::vector<NodeSets::Level> keep;
    for (auto& level : ns.levels)
      if ( ! level.nodes.empty()) keep.push_back(std::move(level));
    ns.levels.swap(keep);
  }
  init_comm(my_rank, ns);
  pt.lap("analyze: init_comm");
  return nodesets;
}

NodeSets::ConstPtr analyze (const Parallel::Ptr& p, const Int& ncells,
                            const tree::Node::Ptr& tree) {
  cedr_throw_if(tree->parent, "analyze must be given the root");
  return analyze(p->rank(), ncells, flatten(tree));
}

Int check_comm (const NodeSets& ns) {
  Int nerr = 0;
  std::vector<Int> offsets(ns.nslots, 0);
  for (const auto& lvl : ns.levels)
    for (const Int idx : lvl.nodes) {
      const auto n = ns.node_h(idx);
      if (n->offset < 0 || n->offset >= ns.nslots) { ++nerr; continue; }
      ++offsets[n->offset];
      for (Int i = 0; i < n->nkids; ++i) {
        const auto kid = ns.node_h(n->kids[i]);
        if (kid->rank != n->rank) ++offsets[kid->offset];
      }
    }
  for (const Int e : offsets) if (e != 1) ++nerr;
  return nerr;
}

Node::Ptr make_tree_over_1d_mesh (const Parallel::Ptr& p, const Int& ncells,
                                  const bool imbalanced) {
  return oned::make_tree(oned::Mesh(ncells, p), imbalanced);
}

namespace oned {
Mesh::Mesh (const Int nc, const Parallel::Ptr& p, const ParallelDecomp::Enum& pd)
  : nc_(nc), nranks_(p->size()), pd_(pd) {
  cedr_throw_if(nranks_ > nc_, "#GIDs < #ranks is not supported.");
}

Mesh::Mesh (const Int nc, const Int nranks, const ParallelDecomp::Enum& pd)
  : nc_(nc), nranks_(nranks), pd_(pd) {
  cedr_throw_if(nranks_ > nc_, "#GIDs < #ranks is not supported.");
}

Int Mesh::rank (const Int& ci) const {  // reference cedr_tree.cpp:366-375
  if (pd_ == ParallelDecomp::contiguous)
    return std::min(nranks_ - 1, ci / (nc_ / nranks_));
  const Int chunk = ci / nranks_;
  return (ci + chunk) % nranks_;
}

// Split rule of reference cedr_tree.cpp:391-412: [cs, ce) -> [cs, cs+cn0) and
// [cs+cn0, ce) with cn0 = cn/2, or cn/3 when imbalanced and cn > 2.
FlatTree make_flat_tree (const Mesh& m, const bool imbalanced) {
  FlatTree t;
  struct Item { Int cs, ce, fparent, which; };
  std::vector<Item> stack;
  stack.push_back(Item{0, m.ncell(), -1, 0});
  while ( ! stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const Int me = t.add(it.fparent);
    if (it.fparent >= 0) (it.which == 0 ? t.kid0 : t.kid1)[it.fparent] = me;
    const Int cn = it.ce - it.cs;
    if (cn == 1) {
      t.rank[me] = m.rank(it.cs);
      t.cellidx[me] = it.cs;
      continue;
    }
    const Int cn0 = (imbalanced && cn > 2) ? cn/3 : cn/2;
    t.nkids[me] = 2;
    stack.push_back(Item{it.cs + cn0, it.ce, me, 1});
    stack.push_back(Item{it.cs, it.cs + cn0, me, 0});
  }
  return t;
}

tree::Node::Ptr make_tree (const Mesh& m, const bool imbalanced) {
  const FlatTree ft = make_flat_tree(m, imbalanced);
  std::vector<tree::Node::Ptr> nodes(ft.size());
  for (Int i = 0; i < ft.size(); ++i) nodes[i] = std::make_shared<tree::Node>();
  for (Int i = 0; i < ft.size(); ++i) {
    tree::Node& n = *nodes[i];
    n.parent = ft.parent[i] >= 0 ? nodes[ft.parent[i]].get() : nullptr;
    n.nkids = ft.nkids[i];
    if (n.nkids) {
      n.kids[0] = nodes[ft.kid0[i]];
      n.kids[1] = nodes[ft.kid1[i]];
    } else {
      n.rank = ft.rank[i];
      n.cellidx = ft.cellidx[i];
    }
  }
  return nodes[0];
}

tree::Node::Ptr make_tree (const Parallel::Ptr& p, const Int& ncells, const bool imbalanced) {
  return make_tree(Mesh(ncells, p), imbalanced);
}
}  // namespace oned

}  // namespace tree
}  // namespace cedr
