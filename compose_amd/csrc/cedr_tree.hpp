// Tree analysis: caller tree -> per-rank NodeSets (level schedule, slot
// offsets, per-level message metadata). Integer output is bit-exact with
// reference cedr/cedr_tree.{hpp,cpp}; the implementation is array based and
// iterative (no shared_ptr recursion) so it scales to millions of cells.
#ifndef COMPOSE_AMD_CEDR_TREE_HPP
#define COMPOSE_AMD_CEDR_TREE_HPP

#include <iostream>
#include <vector>

#include "cedr_tree_caller.hpp"

namespace cedr {
namespace tree {
using cedr::mpi::Parallel;

// Array form of the caller's tree. Node 0 is the root; node indices are in
// pre-order, kids before later siblings.
struct FlatTree {
  std::vector<Int> parent, rank, nkids, kid0, kid1, level;
  std::vector<Long> cellidx;
  Int size () const { return static_cast<Int>(parent.size()); }
  Int add (Int parent_idx);
  Int kid (Int n, Int i) const { return i == 0 ? kid0[n] : kid1[n]; }
};

FlatTree flatten(const Node::Ptr& tree);
// Node indices in post-order: kid 0's subtree, kid 1's subtree, then the node.
std::vector<Int> post_order(const FlatTree& t);

// The part of a full tree rank 'my_rank' has to know, as a PARTIAL tree (cedr_tree_caller.hpp:20-22):
// the nodes it owns, their kids, and every ancestor of those, with ranks and levels filled in and
// everything else pruned away. O(cells/ranks + depth) nodes.
FlatTree prune_for_rank(const FlatTree& full, Int my_rank);

// reference cedr/cedr_tree.hpp:16-97
struct NodeSets {
  typedef std::shared_ptr<const NodeSets> ConstPtr;
  enum : int { mpitag = 42 };

  struct Node {
    Int rank, id, parent, nkids, kids[2], offset;
    Node () : rank(-1), id(-1), parent(-1), nkids(0), offset(-1) { kids[0] = kids[1] = -1; }
  };

  struct Level {
    struct MPIMetaData { Int rank, offset, size; };
    std::vector<Int> nodes;
    std::vector<MPIMetaData> me, kids;
  };

  std::vector<Level> levels;
  Int nslots;

  Int alloc () { node_mem_.push_back(Node()); return static_cast<Int>(node_mem_.size()) - 1; }
  void reserve (size_t n) { node_mem_.reserve(n); }
  Int nnode () const { return static_cast<Int>(node_mem_.size()); }
  Node* node_h (const Int& idx) { return &node_mem_[idx]; }
  const Node* node_h (const Int& idx) const { return &node_mem_[idx]; }
  void print(std::ostream& os) const;

private:
  std::vector<Node> node_mem_;
};

// reference cedr/cedr_tree.cpp:215-231
NodeSets::ConstPtr analyze(const Parallel::Ptr& p, const Int& ncells, const tree::Node::Ptr& tree);
// Array-based entry used by everything in this build.
NodeSets::ConstPtr analyze(const Int my_rank, const Int& ncells, FlatTree tree);

// reference cedr/cedr_tree.cpp:234-251 (per-rank self consistency of offsets)
Int check_comm(const NodeSets& ns);

// Test / benchmark trees over a 1-D mesh, reference cedr/cedr_tree.hpp:126-171.
namespace oned {
struct Mesh {
  struct ParallelDecomp { enum Enum { contiguous, pseudorandom }; };
  Mesh (const Int nc, const Parallel::Ptr& p,
        const ParallelDecomp::Enum& parallel_decomp = ParallelDecomp::contiguous);
  Mesh (const Int nc, const Int nranks, const ParallelDecomp::Enum& parallel_decomp);
  Int ncell () const { return nc_; }
  Int rank(const Int& ci) const;
private:
  Int nc_, nranks_;
  ParallelDecomp::Enum pd_;
};

tree::Node::Ptr make_tree(const Parallel::Ptr& p, const Int& ncells, const bool imbalanced);
tree::Node::Ptr make_tree(const Mesh& m, const bool imbalanced);
// Same tree, directly in array form (no shared_ptr graph).
FlatTree make_flat_tree(const Mesh& m, const bool imbalanced);
}  // namespace oned

}  // namespace tree
}  // namespace cedr

#endif
