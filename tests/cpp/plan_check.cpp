// Host-side check of the tile plans the kernels consume, meant to be built with
// -fsanitize=address,undefined (CPU only): for a sweep of trees, ownerships and
// planner options it builds NodeSets + plan exactly as TreeEngine does and verifies
// every precondition the kernel launchers and kernels rely on (index ranges, LDS
// bounds, micro-subtree shapes, padding reads). A violated precondition on the GPU
// would be an out-of-bounds access; here it is a message and a non-zero exit.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <set>
#include <vector>

#include "cedr_plan.hpp"
#include "cedr_tree.hpp"
#include "cedr_util.hpp"

using namespace cedr;

static int g_fail = 0;
#define REQUIRE(c) do { if ( ! (c)) { std::printf("FAIL %s:%d: %s [%s]\n", __FILE__, __LINE__, #c, ctx); ++g_fail; \
                                     if (g_fail > 20) std::exit(1); } } while (0)

static void check_stage (const plan::Stage& st, const plan::Plan& pl, const int64_t stride, const bool first,
                         const char* ctx) {
  if (st.top) {
    // the top kernels' preconditions (cedr_kernels.hip top_stage_ok): one plain tile, <= 2048 inputs,
    // <= 4095 nodes, a level has at most nleaf/2 <= 1024 nodes (two per thread in qlt_top_kernel)
    REQUIRE(st.tiles.size() == 1 && st.micro.empty() && ! first);
    REQUIRE(st.max_nleaf <= 2048 && st.max_nodes <= 4095);
    const auto& t = st.tiles[0];
    for (int l = 0; l < t.nlvl; ++l)
      REQUIRE(st.lvlptr[t.lvl_off + l + 1] - st.lvlptr[t.lvl_off + l] <= 1024);
    REQUIRE(t.nint == t.nleaf - t.nout);  // a forest with one output per tree
  } else {
    REQUIRE(st.max_nleaf <= 256);
  }
  const bool micro = ! st.micro.empty();
  if (micro) REQUIRE(st.max_nleaf <= 64);
  std::set<int> out_slots;
  for (const auto& t : st.tiles) {
    REQUIRE(t.nleaf >= 1 && t.nleaf <= st.max_nleaf);
    REQUIRE(t.nint >= 0 && t.nint <= st.max_nint);
    REQUIRE(t.nleaf + t.nint <= st.max_nodes);
    REQUIRE(t.nleaf + t.nint <= 0xffff);
    REQUIRE(t.nlvl >= 0 && t.lvl_off + t.nlvl < (int) st.lvlptr.size());
    REQUIRE(2*((size_t) t.kids_off + t.nint) <= st.kids.size());
    REQUIRE(t.nout >= 1 && (size_t) t.outs_off + t.nout <= st.outs.size());
    REQUIRE(t.out >= 0 && t.out + t.nout <= pl.nslots);
    // levels: contiguous prefix over the tile's internal nodes, kids before parents
    int prev = 0;
    for (int l = 0; l <= t.nlvl; ++l) {
      const int v = st.lvlptr[t.lvl_off + l];
      REQUIRE(v >= prev && v <= t.nint);
      prev = v;
    }
    REQUIRE(st.lvlptr[t.lvl_off] == 0 && st.lvlptr[t.lvl_off + t.nlvl] == t.nint);
    for (int l = 0; l < t.nlvl; ++l)
      for (int j = st.lvlptr[t.lvl_off + l]; j < st.lvlptr[t.lvl_off + l + 1]; ++j) {
        const int k0 = st.kids[2*((size_t) t.kids_off + j)], k1 = st.kids[2*((size_t) t.kids_off + j) + 1];
        // a kid is an input or an internal node of a LOWER level
        REQUIRE(k0 < t.nleaf + st.lvlptr[t.lvl_off + l] && k1 < t.nleaf + st.lvlptr[t.lvl_off + l]);
        REQUIRE(k0 != k1);
      }
    for (int k = 0; k < t.nout; ++k) {
      REQUIRE(st.outs[t.outs_off + k] < t.nleaf + t.nint);
      REQUIRE(out_slots.insert(t.out + k).second);  // nobody else writes this slot
    }
    if (micro) {
      REQUIRE(t.nout <= 64);
      REQUIRE((size_t) t.micro_off + t.nleaf <= st.micro.size());
      int next = 0;
      for (int i = 0; i < t.nleaf; ++i) {
        const int d = st.micro[t.micro_off + i], start = d & 0xff, shape = d >> 8;
        REQUIRE(shape >= 0 && shape <= 3);
        const int size = shape == 0 ? 1 : (shape == 1 ? 2 : 3);
        REQUIRE(start == next);  // micro-subtrees tile the input range in order
        next = start + size;
      }
      REQUIRE(next <= 192);
      REQUIRE(t.in0 >= 0 && t.in0 + next <= pl.nslots);
      if (first) REQUIRE(t.in0 + next <= pl.nlcl);  // first-stage inputs are cells
      // unmasked 16 + 8 byte reads may touch two slots past the tile: inside the row
      REQUIRE((int64_t) t.in0 + next + 2 <= stride);
    } else {
      REQUIRE(t.in0 >= 0 && t.in0 + t.nleaf <= pl.nslots);
    }
  }
}

static void check (const tree::FlatTree& t, const int ncells, const int nranks, const plan::Options& opt,
                   const char* ctx) {
  for (int me = 0; me < nranks; ++me) {
    const auto ns = tree::analyze(me, ncells, t);
    REQUIRE(tree::check_comm(*ns) == 0 || nranks > 1);
    util::IdMap g2l;
    for (const Int idx : ns->levels[0].nodes) g2l.add(ns->node_h(idx)->id, ns->node_h(idx)->offset);
    g2l.finish();
    std::vector<int32_t> leaf_slot(t.size(), -1);
    int nown = 0;
    for (Int n = 0; n < t.size(); ++n)
      if (t.nkids[n] == 0 && t.rank[n] == me) { leaf_slot[n] = g2l.find((Int) t.cellidx[n]); REQUIRE(leaf_slot[n] >= 0); ++nown; }
    plan::Plan pl;
    try { pl = plan::build(t, me, nranks, leaf_slot, opt); }
    catch (const std::logic_error&) { continue; }  // orders the planner rejects by design
    REQUIRE(pl.nlcl == nown);
    const int64_t stride = ((int64_t) pl.nslots + 2 + 15)/16*16;
    for (size_t i = 0; i < pl.local.size(); ++i) {
      REQUIRE( ! pl.local[i].top || i + 1 == pl.local.size());  // the fused top tile is the last local stage
      check_stage(pl.local[i], pl, stride, i == 0, ctx);
    }
    if ( ! pl.local.empty() && pl.local.back().top) {
      // its inputs are exactly the previous stage's outputs; on one rank its one output is the root
      const plan::Stage& prev = pl.local[pl.local.size() - 2];
      const plan::Tile& tt = pl.local.back().tiles[0];
      int lo = pl.nslots, hi = -1;
      for (const auto& q : prev.tiles) { lo = std::min(lo, q.out); hi = std::max(hi, q.out + q.nout); }
      REQUIRE(tt.in0 == lo && tt.in0 + tt.nleaf == hi);
      if (nranks == 1) REQUIRE(tt.nout == 1 && tt.out == pl.root_slot);
      // every output of the tile is one of my boundary roots' slots
      if (nranks > 1)
        for (int k = 0; k < tt.nout; ++k) {
          bool found = false;
          for (const int s : pl.my_root_slots) found = found || s == tt.out + k;
          REQUIRE(found);
        }
    }
    for (size_t i = 0; i < pl.top.size(); ++i) check_stage(pl.top[i], pl, stride, false, ctx);
    REQUIRE(pl.root_slot >= 0 && pl.root_slot < pl.nslots);
    if ( ! pl.apex.tiles.empty()) {
      // the apex kernels' preconditions: one plain tile, <= 64 inputs = exactly the boundary roots'
      // top slots (every one covered by a gathered record), one output
      REQUIRE(pl.apex.tiles.size() == 1 && pl.apex.micro.empty());
      const plan::Tile& a = pl.apex.tiles[0];
      REQUIRE(a.nleaf <= plan::Plan::kApexMaxInputs && a.nout == 1 && a.nint == a.nleaf - 1);
      check_stage(pl.apex, pl, stride, false, ctx);
      std::set<int> covered;
      for (const int s : pl.gather_slots) if (s >= 0) { REQUIRE(s >= a.in0 && s < a.in0 + a.nleaf); covered.insert(s); }
      REQUIRE((int) covered.size() == a.nleaf);
      for (const int s : pl.my_top_slots) REQUIRE(s >= a.in0 && s < a.in0 + a.nleaf);
      REQUIRE(pl.apex.outs[a.outs_off] == a.nleaf + a.nint - 1 || a.nint == 0);
    } else if (nranks > 1 && ! pl.top.empty()) {
      int nb = 0;
      for (const int s : pl.gather_slots) nb += s >= 0;
      REQUIRE(nb > plan::Plan::kApexMaxInputs);  // the apex tile is only left out when it would be too large
    }
    if (nranks > 1 && ! pl.level_mode) {
      REQUIRE((int) pl.gather_slots.size() == nranks*pl.nb_max);
      REQUIRE(pl.my_root_slots.size() == pl.my_top_slots.size());
      REQUIRE((int) pl.my_root_slots.size() <= pl.nb_max);
      for (const int s : pl.gather_slots) REQUIRE(s >= -1 && s < pl.nslots);
      for (const int s : pl.my_root_slots) REQUIRE(s >= 0 && s < pl.nslots);
      for (const int s : pl.my_top_slots) REQUIRE(s >= 0 && s < pl.nslots);
    }
  }
}

// Level mode (partial trees / per-level point-to-point): every rank plans from its own tree; the
// ranks' message lists must pair up, every slot a step reads must have been produced before it.
static void check_level (const tree::FlatTree& full, const int ncells, const int nranks, plan::Options opt,
                         const bool partial, const char* ctx) {
  if (nranks < 2) return;
  if ( ! partial) opt.exchange = 2;
  std::vector<plan::Plan> pls(nranks);
  for (int me = 0; me < nranks; ++me) {
    const tree::FlatTree t = partial ? tree::prune_for_rank(full, me) : full;
    const auto ns = tree::analyze(me, ncells, t);
    util::IdMap g2l;
    for (const Int idx : ns->levels[0].nodes) g2l.add(ns->node_h(idx)->id, ns->node_h(idx)->offset);
    g2l.finish();
    std::vector<int32_t> leaf_slot(t.size(), -1);
    for (Int n = 0; n < t.size(); ++n)
      if (t.nkids[n] == 0 && t.rank[n] == me) leaf_slot[n] = g2l.find((Int) t.cellidx[n]);
    plan::Plan& pl = pls[me];
    try { pl = plan::build(t, me, nranks, leaf_slot, opt); }
    catch (const std::logic_error& e) { std::printf("FAIL %s: rank %d: %s\n", ctx, me, e.what()); ++g_fail; return; }
    REQUIRE(pl.level_mode && pl.top.empty() && pl.apex.tiles.empty());
    const int64_t stride = ((int64_t) pl.nslots + 2 + 15)/16*16;
    for (size_t i = 0; i < pl.local.size(); ++i) {
      REQUIRE( ! pl.local[i].top || i + 1 == pl.local.size());  // the fused top tile is the last local stage
      check_stage(pl.local[i], pl, stride, i == 0, ctx);
    }
    if ( ! pl.local.empty() && pl.local.back().top) {
      // its inputs are exactly the previous stage's outputs; on one rank its one output is the root
      const plan::Stage& prev = pl.local[pl.local.size() - 2];
      const plan::Tile& tt = pl.local.back().tiles[0];
      int lo = pl.nslots, hi = -1;
      for (const auto& q : prev.tiles) { lo = std::min(lo, q.out); hi = std::max(hi, q.out + q.nout); }
      REQUIRE(tt.in0 == lo && tt.in0 + tt.nleaf == hi);
      if (nranks == 1) REQUIRE(tt.nout == 1 && tt.out == pl.root_slot);
      // every output of the tile is one of my boundary roots' slots
      if (nranks > 1)
        for (int k = 0; k < tt.nout; ++k) {
          bool found = false;
          for (const int s : pl.my_root_slots) found = found || s == tt.out + k;
          REQUIRE(found);
        }
    }
    REQUIRE(pl.root_slot >= 0 && pl.root_slot < pl.nslots && pl.root_owner >= 0 && pl.root_owner < nranks);
    std::vector<char> have(pl.nslots, 0);
    for (int i = 0; i < pl.nlcl; ++i) have[i] = 1;
    for (const int s : pl.my_root_slots) { REQUIRE(s >= 0 && s < pl.nslots); have[s] = 1; }
    int prev = -1;
    for (const auto& st : pl.steps) {
      REQUIRE(st.level > prev);
      prev = st.level;
      REQUIRE(st.nodes.size() % 3 == 0);
      for (size_t j = 0; j < st.nodes.size(); j += 3) {
        for (int k = 0; k < 3; ++k) REQUIRE(st.nodes[j+k] >= 0 && st.nodes[j+k] < pl.nslots);
        REQUIRE(have[st.nodes[j+1]] && have[st.nodes[j+2]] && ! have[st.nodes[j]]);
      }
      for (size_t j = 0; j < st.nodes.size(); j += 3) have[st.nodes[j]] = 1;
      for (const int s : st.out_slots) REQUIRE(s >= 0 && s < pl.nslots && have[s]);
      for (const int s : st.in_slots) { REQUIRE(s >= 0 && s < pl.nslots && ! have[s]); have[s] = 1; }
      for (const auto* ml : {&st.out_msgs, &st.in_msgs}) {
        const size_t n = ml == &st.out_msgs ? st.out_slots.size() : st.in_slots.size();
        int at = 0, last_peer = -1;
        for (const auto& m : *ml) {
          REQUIRE(m.peer > last_peer && m.peer < nranks && m.peer != me && m.first == at && m.count > 0);
          last_peer = m.peer;
          at += m.count;
        }
        REQUIRE(at == (int) n);
      }
    }
    if (pl.root_owner == me) REQUIRE(have[pl.root_slot]);
  }
  for (int a = 0; a < nranks; ++a) {
    REQUIRE(pls[a].root_owner == pls[0].root_owner);
    for (const auto& st : pls[a].steps)
      for (const auto& m : st.out_msgs) {  // a -> m.peer at st.level: the peer expects exactly that
        int found = 0;
        for (const auto& sb : pls[m.peer].steps)
          if (sb.level == st.level)
            for (const auto& mb : sb.in_msgs) if (mb.peer == a) { REQUIRE(mb.count == m.count); ++found; }
        REQUIRE(found == 1);
      }
    for (const auto& st : pls[a].steps)
      for (const auto& m : st.in_msgs) {
        int found = 0;
        for (const auto& sb : pls[m.peer].steps)
          if (sb.level == st.level)
            for (const auto& mb : sb.out_msgs) if (mb.peer == a) ++found;
        REQUIRE(found == 1);
      }
  }
}

// The same partial tree two ways: pruned from the full tree, and built part by part as tree::Node objects the
// way a host model would (tests/cpp/cedr_test_mpi.cpp has the same builder): the plans must agree.
static tree::Node::Ptr my_tree_part (const int nc, const int nranks, const int me, const int cs, const int ce,
                                     const tree::Node* parent) {
  const auto owner = [&] (const int c) { return std::min(nranks - 1, c/(nc/nranks)); };
  auto n = std::make_shared<tree::Node>();
  n->parent = parent;
  n->rank = owner(cs);
  const int cn = ce - cs;
  if (cn == 1) { n->nkids = 0; n->level = 0; n->cellidx = n->rank == me ? cs : -1; return n; }
  const auto k0 = my_tree_part(nc, nranks, me, cs, cs + cn/2, n.get());
  const auto k1 = my_tree_part(nc, nranks, me, cs + cn/2, ce, n.get());
  n->level = 1 + std::max(k0->level, k1->level);
  if (n->rank == me) { n->nkids = 2; n->kids[0] = k0; n->kids[1] = k1; }
  else {
    n->nkids = 0;
    for (const auto& k : {k0, k1}) if (k->rank == me || k->nkids > 0) n->kids[n->nkids++] = k;
    if (n->nkids == 0) n->nkids = -1;
  }
  return n;
}

static plan::Plan plan_of (const tree::FlatTree& t, const int nc, const int me, const int nranks) {
  const auto ns = tree::analyze(me, nc, t);
  util::IdMap g2l;
  for (const Int idx : ns->levels[0].nodes) g2l.add(ns->node_h(idx)->id, ns->node_h(idx)->offset);
  g2l.finish();
  std::vector<int32_t> leaf_slot(t.size(), -1);
  for (Int n = 0; n < t.size(); ++n)
    if (t.nkids[n] == 0 && t.rank[n] == me) leaf_slot[n] = g2l.find((Int) t.cellidx[n]);
  return plan::build(t, me, nranks, leaf_slot, plan::Options());
}

static void check_node_built_partial_trees () {
  typedef tree::oned::Mesh Mesh;
  const char* ctx = "node-built partial trees";
  for (const int nc : {2, 3, 8, 17, 100, 777, 3001})
    for (const int P : {2, 3, 4, 7}) {
      if (P > nc) continue;
      const tree::FlatTree full = tree::oned::make_flat_tree(Mesh(nc, P, Mesh::ParallelDecomp::contiguous), false);
      for (int me = 0; me < P; ++me) {
        const plan::Plan a = plan_of(tree::prune_for_rank(full, me), nc, me, P);
        const plan::Plan b = plan_of(tree::flatten(my_tree_part(nc, P, me, 0, nc, nullptr)), nc, me, P);
        REQUIRE(a.level_mode && b.level_mode && a.nlcl == b.nlcl && a.root_owner == b.root_owner);
        REQUIRE(a.local.size() == b.local.size() && a.steps.size() == b.steps.size());
        for (size_t i = 0; i < std::min(a.steps.size(), b.steps.size()); ++i) {
          const auto& x = a.steps[i]; const auto& y = b.steps[i];
          REQUIRE(x.level == y.level && x.nodes.size() == y.nodes.size());
          REQUIRE(x.out_slots.size() == y.out_slots.size() && x.in_slots.size() == y.in_slots.size());
          REQUIRE(x.out_msgs.size() == y.out_msgs.size() && x.in_msgs.size() == y.in_msgs.size());
          for (size_t k = 0; k < std::min(x.out_msgs.size(), y.out_msgs.size()); ++k)
            REQUIRE(x.out_msgs[k].peer == y.out_msgs[k].peer && x.out_msgs[k].count == y.out_msgs[k].count);
          for (size_t k = 0; k < std::min(x.in_msgs.size(), y.in_msgs.size()); ++k)
            REQUIRE(x.in_msgs[k].peer == y.in_msgs[k].peer && x.in_msgs[k].count == y.in_msgs[k].count);
        }
      }
    }
}

int main () {
  typedef tree::oned::Mesh Mesh;
  char ctx[160];
  long ncase = 0;
  const int sizes[] = {1, 2, 3, 4, 5, 7, 8, 21, 63, 64, 65, 111, 127, 128, 129, 191, 192, 193, 255, 256, 257, 777,
                       1000, 1024, 4097};
  const int ranks[] = {1, 2, 3, 8};
  for (const int nc : sizes)
    for (const int P : ranks) {
      if (P > nc || (nc > 1024 && P != 1 && P != 8)) continue;
      for (int pr = 0; pr < 2; ++pr)
        for (int imb = 0; imb < 2; ++imb) {
          const Mesh m(nc, P, pr ? Mesh::ParallelDecomp::pseudorandom : Mesh::ParallelDecomp::contiguous);
          const tree::FlatTree t = tree::oned::make_flat_tree(m, imb != 0);
          for (const int tile : {2, 3, 16, 128, 256})
            for (const int micro : {0, 1})
              for (const int cut : {0, 1, 2, 3, 5}) {
                if ( ! micro && cut != 3) continue;
                plan::Options o;
                o.max_tile_leaves = tile; o.micro = micro != 0; o.micro_cut = cut;
                o.micro_cut2 = cut == 2 ? 2 : (cut == 5 ? 3 : (cut == 3 ? 4 : 0));
                std::snprintf(ctx, sizeof(ctx), "nc %d P %d pseudorandom %d imbalanced %d tile %d micro %d cut %d",
                              nc, P, pr, imb, tile, micro, cut);
                check(t, nc, P, o, ctx);
                ++ncase;
                if (P > 1 && (cut == 2 || ! micro) && (tile == 3 || tile == 128)) {
                  check_level(t, nc, P, o, false, ctx);
                  check_level(t, nc, P, o, true, ctx);
                  ncase += 2;
                }
              }
        }
    }
  check_node_built_partial_trees();
  {  // util::IdMap's contract (the engine's gci2lci and the C ABI's g2l rely on it), dense and sparse form:
     // a duplicate id returns its FIRST index, size() is the number of adds, an absent id gives -1
    for (const int stride : {1, 100000}) {  // ids 0, s, 2s, ...: one block -> direct table; far apart -> sorted pairs
      util::IdMap m;
      const int ids[] = {5, 2, 9, 2, 7, 5};
      for (int i = 0; i < 6; ++i) m.add(ids[i]*stride, 10 + i);
      m.finish();
      bool ok = m.size() == 6 && m.find(2*stride) == 11 && m.find(5*stride) == 10 && m.find(9*stride) == 12 &&
        m.find(7*stride) == 14 && m.find(3*stride) == -1 && m.find(-1) == -1 && m.find(10*stride) == -1 &&
        m.find(2*stride + 1) == -1;
      util::IdMap e;
      e.finish();
      ok = ok && e.size() == 0 && e.find(0) == -1;
      if ( ! ok) { std::printf("IdMap contract broken (stride %d)\n", stride); ++g_fail; }
      ++ncase;
    }
  }
  std::printf("%ld plans checked, %d failures\n", ncase, g_fail);
  std::printf(g_fail ? "FAIL\n" : "PASS\n");
  return g_fail ? 1 : 0;
}
