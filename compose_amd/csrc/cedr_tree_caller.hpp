// The caller-supplied tree, source compatible with reference
// cedr/cedr_tree_caller.hpp:12-30.
#ifndef COMPOSE_AMD_CEDR_TREE_CALLER_HPP
#define COMPOSE_AMD_CEDR_TREE_CALLER_HPP

#include <memory>

#include "cedr_mpi.hpp"

namespace cedr {
namespace tree {

struct Node {
  typedef std::shared_ptr<Node> Ptr;
  const Node* parent;
  Int rank;      // owning rank
  Long cellidx;  // at a leaf: the cell this node stands for
  Int nkids;     // 0 at a leaf, 2 otherwise
  Node::Ptr kids[2];
  Int reserved;
  Int level;     // partial trees only (not supported by this build)
  Node () : parent(nullptr), rank(-1), cellidx(-1), nkids(0), reserved(-1), level(-1) {}
};

Node::Ptr make_tree_over_1d_mesh(const mpi::Parallel::Ptr& p, const Int& ncells,
                                 const bool imbalanced = false);

}  // namespace tree
}  // namespace cedr

#endif
