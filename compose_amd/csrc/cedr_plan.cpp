// Builds the tile/stage plan described in cedr_plan.hpp. Pure host integer work.
#include "cedr_plan.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>

#include "cedr_util.hpp"

namespace cedr {
namespace plan {
namespace {

struct Builder {
  const tree::FlatTree& t;
  const Options opt;
  std::vector<int32_t> slot;   // per tree node: slot of its record once it is an input
  std::vector<int32_t> cnt;    // scratch: # current inputs below a node
  std::vector<int32_t> lid;    // scratch: local id inside the tile being built
  std::vector<int32_t> llvl;   // scratch: local level inside the tile being built
  std::vector<int32_t> nin;    // scratch: # current inputs (nodes holding a slot) below a node
  std::vector<int32_t> first_in; // scratch: local id of a node's first input
  std::vector<char> has_parent;  // scratch: node has a kept parent in the tile
  bool micro_mode = false;     // inputs are micro-subtrees (<= 3 leaves) instead of slots
  int32_t stage_cut = 0;       // local levels the tiles of the stage being built keep (0 = all).
                               // Only the first two stages may be cut: later ones are small, and
                               // a cut there would add a stage = a launch in each direction.
  int32_t next_slot;

  bool is_input (const int32_t n) const {
    return micro_mode ? nin[n] <= 3 : slot[n] >= 0;
  }

  Builder (const tree::FlatTree& t_, const Options& o)
    : t(t_), opt(o), slot(t_.size(), -1), cnt(t_.size(), 0), lid(t_.size(), -1),
      llvl(t_.size(), 0), nin(t_.size(), 0), first_in(t_.size(), 0), has_parent(t_.size(), 0),
      next_slot(0) {}

  // nin[n] for every node reachable from r without passing through a slot holder.
  void count_slots (const int32_t r) {
    std::vector<std::pair<int32_t,int32_t> > st;
    st.push_back(std::make_pair(r, 0));
    while ( ! st.empty()) {
      auto& top = st.back();
      const int32_t n = top.first;
      if (slot[n] >= 0) { nin[n] = 1; st.pop_back(); continue; }
      if (top.second < 2) {
        const int32_t k = t.kid(n, top.second++);
        st.push_back(std::make_pair(k, 0));
      } else {
        nin[n] = nin[t.kid0[n]] + nin[t.kid1[n]];
        st.pop_back();
      }
    }
  }

  // cnt[n] for every node reachable from r without passing through an input.
  void count_inputs (const int32_t r) {
    std::vector<std::pair<int32_t,int32_t> > st;
    st.push_back(std::make_pair(r, 0));
    while ( ! st.empty()) {
      auto& top = st.back();
      const int32_t n = top.first;
      if (is_input(n)) { cnt[n] = 1; st.pop_back(); continue; }
      if (top.second < 2) {
        const int32_t k = t.kid(n, top.second++);
        st.push_back(std::make_pair(k, 0));
      } else {
        cnt[n] = cnt[t.kid0[n]] + cnt[t.kid1[n]];
        st.pop_back();
      }
    }
  }

  // Append the tile rooted at tree node r to st. Inputs are the nodes with
  // slot >= 0 met by the DFS, in DFS order; their slots must be consecutive.
  void build_tile (Stage& st, const int32_t r) { build_tile(st, std::vector<int32_t>(1, r)); }

  // Several roots (DFS order): the fused top tile over a forest. Its outputs are the roots.
  void build_tile (Stage& st, const std::vector<int32_t>& rts) {
    Tile tile;
    tile.micro_off = micro_mode ? static_cast<int32_t>(st.micro.size()) : 0;
    int32_t nreal = 0;  // real leaves seen so far (micro mode)
    tile.kids_off = static_cast<int32_t>(st.kids.size()/2);
    tile.lvl_off = static_cast<int32_t>(st.lvlptr.size());
    std::vector<int32_t> internals;  // post-order
    std::vector<int32_t> inputs;     // DFS order
    int32_t nleaf = 0, in0 = -1;
    {
      std::vector<std::pair<int32_t,int32_t> > stack;
      for (auto it = rts.rbegin(); it != rts.rend(); ++it) stack.push_back(std::make_pair(*it, 0));
      while ( ! stack.empty()) {
        auto& top = stack.back();
        const int32_t n = top.first;
        if (is_input(n)) {
          if (micro_mode) {
            // Inputs (slot holders: cells in the first stage, outputs of the
            // previous stage later) of the micro-subtree in DFS order, and its shape.
            int32_t lv[3], nl = 0;
            int32_t shape = 0;
            if (slot[n] >= 0) lv[nl++] = n;
            else {
              const int32_t a = t.kid0[n], b = t.kid1[n];
              if (slot[a] >= 0) lv[nl++] = a; else { lv[nl++] = t.kid0[a]; lv[nl++] = t.kid1[a]; }
              if (slot[b] >= 0) lv[nl++] = b; else { lv[nl++] = t.kid0[b]; lv[nl++] = t.kid1[b]; }
              shape = nl == 2 ? 1 : (slot[a] < 0 ? 2 : 3);
            }
            for (int32_t i = 0; i < nl; ++i) {
              cedr_throw_if(slot[lv[i]] < 0, "micro-subtree leaf without a slot");
              if (nleaf == 0 && i == 0) in0 = slot[lv[0]];
              cedr_throw_if(slot[lv[i]] != in0 + nreal + i,
                            "The local cell order does not keep each owned subtree's cells "
                            "contiguous; this ordering is not supported yet.");
            }
            cedr_throw_if(nreal > 255, "micro tile spans more than 255 leaves");
            st.micro.push_back(static_cast<uint16_t>(nreal | (shape << 8)));
            nreal += nl;
          } else {
            if (nleaf == 0) in0 = slot[n];
            cedr_throw_if(slot[n] != in0 + nleaf,
                          "The local cell order does not keep each owned subtree's cells "
                          "contiguous; this ordering is not supported yet.");
          }
          lid[n] = nleaf++;
          llvl[n] = 0;
          inputs.push_back(n);
          stack.pop_back();
          continue;
        }
        if (top.second < 2) {
          const int32_t k = t.kid(n, top.second++);
          stack.push_back(std::make_pair(k, 0));
        } else {
          llvl[n] = 1 + std::max(llvl[t.kid0[n]], llvl[t.kid1[n]]);
          internals.push_back(n);
          stack.pop_back();
        }
      }
    }
    // Cut tiles keep only the internal nodes of local level <= cut.
    const int32_t cut = (micro_mode && stage_cut > 0) ? stage_cut : 0;
    std::vector<int32_t> dropped;
    if (cut > 0) {
      std::vector<int32_t> keep;
      for (const int32_t n : internals) (llvl[n] <= cut ? keep : dropped).push_back(n);
      internals.swap(keep);
    }
    tile.in0 = in0;
    tile.nleaf = nleaf;
    tile.nint = static_cast<int32_t>(internals.size());
    cedr_throw_if(tile.nleaf + tile.nint > 65535, "tile too large for 16-bit local ids");
    cedr_throw_if(rts.size() > 1 && cut > 0, "a tile over several roots cannot be cut");
    // Kids before parents: order internal nodes by local level, DFS order within.
    std::stable_sort(internals.begin(), internals.end(),
                     [&] (int32_t a, int32_t b) { return llvl[a] < llvl[b]; });
    for (size_t i = 0; i < internals.size(); ++i)
      lid[internals[i]] = nleaf + static_cast<int32_t>(i);
    // lvlptr: start offsets of each local level's internal nodes, then nint.
    tile.nlvl = 0;
    st.lvlptr.push_back(0);
    for (size_t i = 0; i < internals.size(); ++i) {
      const int32_t n = internals[i];
      if (i > 0 && llvl[n] != llvl[internals[i-1]]) {
        st.lvlptr.push_back(static_cast<int32_t>(i));
        ++tile.nlvl;
      }
      st.kids.push_back(static_cast<uint16_t>(lid[t.kid0[n]]));
      st.kids.push_back(static_cast<uint16_t>(lid[t.kid1[n]]));
    }
    if ( ! internals.empty()) { st.lvlptr.push_back(tile.nint); ++tile.nlvl; }
    // Outputs: kept nodes (inputs or internal) without a kept parent, in DFS
    // order. lid is increasing along the DFS for inputs; an internal node sits
    // where its first input sits.
    std::vector<int32_t> outs;
    if (dropped.empty()) {
      outs = rts;
    } else {
      for (const int32_t n : inputs) has_parent[n] = 0;
      for (const int32_t n : internals) has_parent[n] = 0;
      for (const int32_t n : internals) has_parent[t.kid0[n]] = has_parent[t.kid1[n]] = 1;
      for (const int32_t n : inputs) { first_in[n] = lid[n]; if ( ! has_parent[n]) outs.push_back(n); }
      for (const int32_t n : internals) {  // level order: kids are done before parents
        first_in[n] = first_in[t.kid0[n]];
        if ( ! has_parent[n]) outs.push_back(n);
      }
      std::sort(outs.begin(), outs.end(),
                [&] (int32_t a, int32_t b) { return first_in[a] < first_in[b]; });
    }
    tile.out = next_slot;
    tile.nout = static_cast<int32_t>(outs.size());
    tile.outs_off = static_cast<int32_t>(st.outs.size());
    for (const int32_t n : outs) {
      st.outs.push_back(static_cast<uint16_t>(lid[n]));
      pending.push_back(std::make_pair(n, next_slot++));
    }
    st.max_nodes = std::max(st.max_nodes, tile.nleaf + tile.nint);
    st.max_nint = std::max(st.max_nint, tile.nint);
    st.max_nleaf = std::max(st.max_nleaf, tile.nleaf);
    st.tiles.push_back(tile);
  }

  std::vector<std::pair<int32_t,int32_t> > pending;  // (tile root, out slot) of the stage

  // Reduce every root in 'roots' to a single input slot, appending stages.
  // micro: 0 = no stage uses micro tiles, 1 = the first stage does, 2 = all do.
  void tile_forest (const std::vector<int32_t>& roots, std::vector<Stage>& stages,
                    const int micro, const bool cut_first, const bool fuse_top = false) {
    bool first = true;
    std::vector<int32_t> prev_active;  // the roots the previous stage still worked on
    for (int istage = 0;; ++istage) {
      micro_mode = micro == 2 || (micro == 1 && first);
      // With a fused top every stage from the second on is cut like the second: the levels a cut
      // leaves out land in the top tile's launch. Without, only the first two may be cut: later
      // stages are small, and a cut there would add a stage = a launch in each direction.
      stage_cut = ! cut_first ? 0 : (istage == 0 ? opt.micro_cut :
                                     ((istage == 1 || fuse_top) ? opt.micro_cut2 : 0));
      first = false;
      std::vector<int32_t> active;
      for (const int32_t r : roots) if (slot[r] < 0) active.push_back(r);
      if (active.empty()) break;
      for (const int32_t r : active) count_slots(r);
      if (fuse_top && istage > 0) {
        // What is left above the previous stage's outputs, which are consecutive slots: the roots
        // that stage finished (one slot each; copies in the top tile) and everything below the
        // roots still open. Few enough: ONE plain tile over all of them ends the forest.
        int64_t left = 0;
        for (const int32_t r : prev_active) left += slot[r] >= 0 ? 1 : nin[r];
        if (left <= opt.top_max_inputs) {
          micro_mode = false;
          stage_cut = 0;
          Stage st;
          st.top = true;
          pending.clear();
          build_tile(st, prev_active);
          for (const auto& p : pending) slot[p.first] = p.second;
          stages.push_back(std::move(st));
          break;
        }
      }
      for (const int32_t r : active) count_inputs(r);
      Stage st;
      pending.clear();
      std::vector<int32_t> stack(active.rbegin(), active.rend());
      while ( ! stack.empty()) {
        const int32_t n = stack.back();
        stack.pop_back();
        if (is_input(n) || cnt[n] <= (micro_mode ? 64 : opt.max_tile_leaves)) {
          // An input hanging directly off a too-large node becomes a 1-input
          // tile (a copy) so that the next stage's inputs stay contiguous.
          build_tile(st, n);
        } else {
          stack.push_back(t.kid1[n]);
          stack.push_back(t.kid0[n]);
        }
      }
      for (const auto& p : pending) slot[p.first] = p.second;
      stages.push_back(std::move(st));
      prev_active.swap(active);
    }
    micro_mode = false;
  }
};

}  // namespace

// Level mode (cedr_plan.hpp, LevelStep). Works from what one rank knows: the nodes it owns, their
// kids, and the chain of ancestors, with explicit ranks and levels wherever the tree is pruned.
static Plan build_level_mode (const tree::FlatTree& t, const int32_t me, const int32_t nranks,
                       const std::vector<int32_t>& leaf_slot, const Options& opt) {
  const int32_t nn = t.size();
  Plan p;
  p.my_rank = me;
  p.nranks = nranks;
  p.level_mode = true;

  // Ranks of internal nodes default to kid 0's (cedr_tree.cpp:45); levels to the height, unless
  // the caller gives them; pure[n]: n and everything below it is mine and present.
  std::vector<int32_t> rank(t.rank.begin(), t.rank.end()), lvl(nn, 0);
  std::vector<char> pure(nn, 0);
  for (const Int n : tree::post_order(t)) {
    int32_t h = -1;
    for (Int i = 0; i < t.nkids[n]; ++i) h = std::max(h, lvl[t.kid(n, i)]);
    lvl[n] = t.level[n] >= 0 ? t.level[n] : h + 1;
    if (rank[n] < 0 && t.nkids[n] > 0) rank[n] = rank[t.kid0[n]];
    cedr_throw_if(rank[n] < 0 || rank[n] >= nranks, "a tree node has no valid rank: " << rank[n]);
    if (rank[n] != me) continue;
    if (t.nkids[n] == 0) {
      cedr_throw_if(lvl[n] != 0, "an owned node without kids is a cell and must be at level 0");
      pure[n] = 1;
    } else {
      cedr_throw_if(t.nkids[n] != 2, "Only binary trees are supported: an owned node has "
                    << t.nkids[n] << " kids.");
      for (Int i = 0; i < 2; ++i)
        cedr_throw_if(lvl[t.kid(n, i)] >= lvl[n], "a kid's level is not below its parent's");
      pure[n] = pure[t.kid0[n]] && pure[t.kid1[n]];
    }
  }
  p.root_owner = rank[0];

  Builder b(t, opt);
  for (int32_t n = 0; n < nn; ++n)
    if (t.nkids[n] == 0 && rank[n] == me) {
      cedr_throw_if(leaf_slot[n] < 0, "owned leaf without a local index");
      b.slot[n] = leaf_slot[n];
      ++p.nlcl;
    }
  b.next_slot = p.nlcl;

  // Roots of my maximal pure subtrees, DFS order (node indices are pre-order): tile stages.
  std::vector<int32_t> my_roots;
  for (int32_t n = 0; n < nn; ++n)
    if (pure[n] && (t.parent[n] < 0 || ! pure[t.parent[n]])) my_roots.push_back(n);
  b.tile_forest(my_roots, p.local, opt.micro ? (opt.micro_all ? 2 : 1) : 0, true, opt.fused_top);
  for (const int32_t r : my_roots) p.my_root_slots.push_back(b.slot[r]);

  // Slots of the rest of my nodes and of their kids elsewhere.
  for (int32_t n = 0; n < nn; ++n) {
    if (rank[n] != me || pure[n]) continue;
    if (b.slot[n] < 0) b.slot[n] = b.next_slot++;
    for (Int i = 0; i < 2; ++i) {
      const int32_t c = t.kid(n, i);
      if (b.slot[c] < 0) b.slot[c] = b.next_slot++;
    }
  }
  p.root_slot = b.slot[0] >= 0 ? b.slot[0] : b.next_slot++;
  p.nslots = b.next_slot;

  // Steps: (peer, slot) lists grouped by level and peer.
  struct Ev { int32_t level, peer, slot, node; };
  std::vector<Ev> outs, ins;
  std::vector<std::pair<int32_t,int32_t> > mine;  // (level, node)
  for (int32_t n = 0; n < nn; ++n) {
    if (rank[n] != me) continue;
    const int32_t par = t.parent[n];
    if (par >= 0 && rank[par] != me) outs.push_back(Ev{lvl[n], rank[par], b.slot[n], n});
    if (pure[n]) continue;
    mine.push_back(std::make_pair(lvl[n], n));
    for (Int i = 0; i < 2; ++i) {
      const int32_t c = t.kid(n, i);
      if (rank[c] != me) ins.push_back(Ev{lvl[c], rank[c], b.slot[c], c});
    }
  }
  // Within a message the records are in the nodes' own DFS (pre-order) order -- NOT their parents':
  // a kid of a node can come after a same-level kid of one of that node's descendants.
  const auto by_level_peer = [] (const Ev& a, const Ev& c) {
    if (a.level != c.level) return a.level < c.level;
    return a.peer != c.peer ? a.peer < c.peer : a.node < c.node;
  };
  std::stable_sort(outs.begin(), outs.end(), by_level_peer);
  std::stable_sort(ins.begin(), ins.end(), by_level_peer);
  std::stable_sort(mine.begin(), mine.end(),
                   [] (const std::pair<int32_t,int32_t>& a, const std::pair<int32_t,int32_t>& c) {
                     return a.first < c.first; });
  std::vector<int32_t> levels;
  for (const Ev& e : outs) levels.push_back(e.level);
  for (const Ev& e : ins) levels.push_back(e.level);
  for (const auto& m : mine) levels.push_back(m.first);
  std::sort(levels.begin(), levels.end());
  levels.erase(std::unique(levels.begin(), levels.end()), levels.end());
  size_t io = 0, ii = 0, im = 0;
  const auto fill = [] (const std::vector<Ev>& ev, size_t& i, const int32_t L, std::vector<int32_t>& slots,
                        std::vector<LevelStep::Msg>& msgs) {
    for (; i < ev.size() && ev[i].level == L; ++i) {
      if (msgs.empty() || msgs.back().peer != ev[i].peer)
        msgs.push_back(LevelStep::Msg{ev[i].peer, static_cast<int32_t>(slots.size()), 0});
      ++msgs.back().count;
      slots.push_back(ev[i].slot);
    }
  };
  for (const int32_t L : levels) {
    LevelStep st;
    st.level = L;
    for (; im < mine.size() && mine[im].first == L; ++im) {
      const int32_t n = mine[im].second;
      st.nodes.push_back(b.slot[n]);
      st.nodes.push_back(b.slot[t.kid0[n]]);
      st.nodes.push_back(b.slot[t.kid1[n]]);
    }
    fill(outs, io, L, st.out_slots, st.out_msgs);
    fill(ins, ii, L, st.in_slots, st.in_msgs);
    p.steps.push_back(std::move(st));
  }
  return p;
}

Plan build (const tree::FlatTree& t, int32_t my_rank, int32_t nranks,
            const std::vector<int32_t>& leaf_slot, const Options& opt) {
  const int32_t nn = t.size();
  cedr_throw_if(opt.max_tile_leaves < 2 || opt.max_tile_leaves > 256,
                "max_tile_leaves must be in [2, 256]");
  cedr_throw_if(opt.fused_top && (opt.top_max_inputs < 2 || opt.top_max_inputs > 2048),
                "top_max_inputs must be in [2, 2048]");
  Plan p;
  p.my_rank = my_rank;
  p.nranks = nranks;

  // "If providing only partial trees, set level" (cedr_tree_caller.hpp:20-22): a caller that sets
  // levels says its trees may be partial. Whether THIS rank's tree has anything missing cannot
  // decide it -- a rank whose part happens to be complete must still pick the same protocol as the
  // others.
  bool partial = false;
  for (int32_t n = 0; n < nn; ++n)
    if (t.level[n] >= 0 || (t.nkids[n] != 2 && t.nkids[n] != 0)) partial = true;
  cedr_throw_if(partial && opt.exchange == 1,
                "a partial tree (Node::level) cannot use the all-gather exchange");
  cedr_throw_if(nranks > 1 && opt.no_p2p && (partial || opt.exchange == 2),
                "a partial tree (tree::Node::level), or exchange == 2, needs one round of point-to-point messages "
                "per tree level, which this communicator does not have (mpi::Comm::has_exchange)");
  if (nranks > 1 && (partial || opt.exchange == 2)) return build_level_mode(t, my_rank, nranks, leaf_slot, opt);

  // own[n]: the rank owning every leaf below n, or -1 if more than one does.
  std::vector<int32_t> own(nn, -1), rk(t.rank.begin(), t.rank.end());
  int64_t ncells = 0;
  bool pinned_inside = false;  // a node pinned to another rank than the one owning everything below it
  for (const Int n : tree::post_order(t)) {
    if (t.nkids[n] == 0) {
      cedr_throw_if(t.rank[n] < 0 || t.rank[n] >= nranks, "leaf rank out of range");
      own[n] = t.rank[n];
      ++ncells;
    } else {
      cedr_throw_if(t.nkids[n] != 2, "Only binary trees are supported: a node has "
                    << t.nkids[n] << " kids.");
      const int32_t a = own[t.kid0[n]], b = own[t.kid1[n]];
      own[n] = (a >= 0 && a == b) ? a : -1;
      if (rk[n] < 0) rk[n] = rk[t.kid0[n]];
      if (own[n] >= 0 && rk[n] != own[n]) pinned_inside = true;
    }
  }
  // The replicated-top design takes ownership from the leaves. A caller that pins a node inside a
  // single-rank subtree to ANOTHER rank (tree::Node::rank) wants that node computed there; level
  // mode does exactly that (every rank sees the same full tree here, so all decide alike).
  cedr_throw_if(pinned_inside && opt.exchange == 0 && nranks > 1 && opt.no_p2p,
                "tree::Node::rank pins a node to another rank than the one owning every cell below it; computing "
                "it there needs point-to-point messages per tree level, which this communicator does not have "
                "(mpi::Comm::has_exchange): override has_exchange() / exchange(), or drop the pin");
  if (pinned_inside && opt.exchange == 0 && nranks > 1) return build_level_mode(t, my_rank, nranks, leaf_slot, opt);
  // exchange == 1 (the caller insists on the all-gather design) cannot honour such a pin either: the node would be
  // computed by the owner of the cells below it. Refused like the case above, not done silently (cedr_mpi.hpp).
  cedr_throw_if(pinned_inside && opt.exchange == 1 && nranks > 1,
                "tree::Node::rank pins a node to another rank than the one owning every cell below it, and exchange = 1 "
                "(CEDR_HIP_EXCHANGE=1: one all-gather + replicated top) computes every node where its cells are: drop "
                "the pin, or leave exchange at 0 / 2 so that the plan exchanges tree levels point to point");

  // Boundary roots = roots of maximal single-rank subtrees, in DFS order.
  std::vector<int32_t> broots;
  {
    std::vector<int32_t> stack(1, 0);
    while ( ! stack.empty()) {
      const int32_t n = stack.back();
      stack.pop_back();
      if (own[n] >= 0) { broots.push_back(n); continue; }
      stack.push_back(t.kid1[n]);
      stack.push_back(t.kid0[n]);
    }
  }

  if (opt.exchange == 0 && nranks > 1) {
    // The all-gather hands every rank nranks*nb_max records and the sweeps over them; a rank's own
    // share of the tree is ~2*ncells/nranks nodes. Scattered ownership (nb_max ~ ncells/nranks)
    // makes the first nranks/2 times the second: per-level messages instead, unless the whole
    // thing is so small that ~2*depth rounds of messages cost more than the replicated sweeps.
    std::vector<int64_t> per_rank(nranks, 0);
    for (const int32_t r : broots) ++per_rank[own[r]];
    const int64_t nb_max = *std::max_element(per_rank.begin(), per_rank.end());
    if (nranks*nb_max > 16384 + 2*ncells/nranks) {
      if ( ! opt.no_p2p) return build_level_mode(t, my_rank, nranks, leaf_slot, opt);
      static std::atomic<bool> warned(false);  // (ranks may be threads of one process)
      if (my_rank == 0 && ! warned.exchange(true)) {
        std::fprintf(stderr, "cedr (hip): ownership is scattered (%lld boundary subtrees per rank at most, %lld cells "
                     "on %d ranks) and the communicator has no point-to-point exchange (mpi::Comm::has_exchange): "
                     "falling back to one all-gather of %lld records; every rank then holds and sweeps about the "
                     "whole tree.\n", static_cast<long long>(nb_max), static_cast<long long>(ncells), nranks,
                     static_cast<long long>(nranks*nb_max));
      }
    }
  }

  Builder b(t, opt);
  for (int32_t n = 0; n < nn; ++n)
    if (t.nkids[n] == 0 && t.rank[n] == my_rank) {
      cedr_throw_if(leaf_slot[n] < 0, "owned leaf without a local index");
      b.slot[n] = leaf_slot[n];
      ++p.nlcl;
    }
  b.next_slot = p.nlcl;

  std::vector<int32_t> my_roots;
  for (const int32_t r : broots) if (own[r] == my_rank) my_roots.push_back(r);
  b.tile_forest(my_roots, p.local, opt.micro ? (opt.micro_all ? 2 : 1) : 0, true, opt.fused_top);
  for (const int32_t r : my_roots) p.my_root_slots.push_back(b.slot[r]);

  if (broots.size() > 1) {
    std::vector<int32_t> per_rank(nranks, 0);
    for (const int32_t r : broots) ++per_rank[own[r]];
    p.nb_max = *std::max_element(per_rank.begin(), per_rank.end());
    p.gather_slots.assign(static_cast<size_t>(nranks)*p.nb_max, -1);
    std::fill(per_rank.begin(), per_rank.end(), 0);
    for (const int32_t r : broots) {
      const int32_t ts = b.next_slot++;
      b.slot[r] = ts;
      p.gather_slots[static_cast<size_t>(own[r])*p.nb_max + per_rank[own[r]]++] = ts;
      if (own[r] == my_rank) p.my_top_slots.push_back(ts);
    }
    if (static_cast<int32_t>(broots.size()) <= Plan::kApexMaxInputs) {
      // build_tile does not touch 'slot' (tile_forest applies 'pending' afterwards): undo the rest
      const int32_t keep_next = b.next_slot;
      b.micro_mode = false;
      b.stage_cut = 0;
      b.pending.clear();
      b.build_tile(p.apex, 0);
      b.pending.clear();
      b.next_slot = keep_next;
    }
    b.tile_forest(std::vector<int32_t>(1, 0), p.top, (opt.micro && opt.micro_all) ? 2 : 0, false);
  }
  p.root_slot = b.slot[0];
  p.nslots = b.next_slot;
  return p;
}

}  // namespace plan
}  // namespace cedr
