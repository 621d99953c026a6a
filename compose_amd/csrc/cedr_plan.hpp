// Execution plan for tree sweeps on MI355X.
//
// The reference walks the tree one level per kernel launch with every node's
// record in global memory (cedr_qlt.cpp:339-430, 525-604) and exchanges one
// MPI message per (level, partner rank). This build instead cuts the tree into
// TILES -- subtrees small enough for one workgroup to hold in LDS -- and runs
// a few STAGES: stage 1 tiles sit on the leaf cells, the roots of stage-k tiles
// are the "leaves" of stage k+1. Only tile roots ever touch HBM besides the
// leaves themselves. Ranks own whole subtrees ("pure" subtrees: every leaf on
// one rank); the part of the tree above them is small, is replicated on every
// rank, and is fed by ONE all-gather of the pure subtrees' root records.
//
// Every node still computes exactly the reference's expression on exactly the
// reference's operands, so results are bit-identical whatever the tiling and
// whatever the rank count.
#ifndef COMPOSE_AMD_CEDR_PLAN_HPP
#define COMPOSE_AMD_CEDR_PLAN_HPP

#include <cstdint>
#include <vector>

#include "cedr_tree.hpp"

namespace cedr {
namespace plan {

// One tile = one subtree handled by one workgroup. Input slots are contiguous.
// Local node numbering inside a tile: inputs 0..nleaf-1 in slot order, then
// internal nodes nleaf..nleaf+nint-1 ordered by local level (kids before
// parents); the tile root is the last node (or input 0 if nint == 0).
//
// MICRO tiles: the inputs of the tile's LDS tree are not single slots but
// "micro-subtrees" of at most 3 consecutive inputs (cells in the first stage,
// outputs of the previous stage later), one per lane, combined/solved in registers. Then nleaf counts the
// micro-subtrees (<= 64), micro_off points at their descriptors in
// Stage::micro, and in0 is the slot of the first leaf of the first micro-subtree.
struct Tile {
  int32_t in0;       // first input slot
  int32_t nleaf;     // number of inputs of the LDS tree (slots, or micro-subtrees)
  int32_t out;       // first output slot: output k of the tile lives in slot out + k
  int32_t nint;      // number of internal nodes
  int32_t kids_off;  // offset of this tile's first internal node in Stage::kids (pairs)
  int32_t lvl_off;   // offset into Stage::lvlptr
  int32_t nlvl;      // number of local levels of internal nodes
  int32_t micro_off; // micro tiles: offset into Stage::micro; else 0
  int32_t nout;      // number of outputs (1 = the tile root, unless the tile is cut)
  int32_t outs_off;  // offset into Stage::outs of the outputs' local node ids
};

struct Stage {
  std::vector<Tile> tiles;
  std::vector<uint16_t> kids;   // 2 local node ids per internal node
  std::vector<int32_t> lvlptr;  // per tile: nlvl+1 prefix offsets over its internal nodes
  int32_t max_nodes = 0;        // max over tiles of nleaf + nint
  int32_t max_nint = 0;
  int32_t max_nleaf = 0;        // max over tiles of nleaf
  // Micro tiles only (empty otherwise): per micro-subtree, first leaf relative
  // to the tile's in0 (low byte) and shape (high byte): 0 = one leaf a,
  // 1 = (a,b), 2 = ((a,b),c), 3 = (a,(b,c)).
  std::vector<uint16_t> micro;
  // Local node ids of every tile's outputs, in DFS order. A tile normally has
  // one output, its root. A CUT tile keeps only the lowest 'cut' local levels
  // of its tree in this stage; the subtrees' roots at that height are its
  // outputs and the nodes above them are left to the next stage, which sees
  // the outputs as consecutive input slots. Cutting shortens the chain of
  // dependent tree levels a workgroup has to walk per tile.
  std::vector<uint16_t> outs;
  // FUSED TOP (plan::Options::fused_top): the stage is ONE plain tile over everything that is left
  // of this rank's forest above the previous stage's outputs -- up to Options::top_max_inputs
  // inputs, several tree levels, one output per subtree root of the rank (an input that is itself
  // such a root is an output too: a copy). It is swept by the tracer-parallel top kernels
  // (cedr_hip_tree_top_up / cedr_hip_qlt_top): one workgroup per tracer (or group of fields) walks
  // ALL its levels in one launch, instead of one launch per stage for the last, latency-bound
  // stages. Always the last local stage.
  bool top = false;
};

// LEVEL MODE. When the caller hands every rank only its part of the tree (tree::Node::level,
// cedr_tree_caller.hpp:20-22) the part above a rank's pure subtrees cannot be replicated, and when
// ownership does not follow subtrees (oned::Mesh pseudorandom) replicating it costs O(N) per rank.
// Then a rank keeps the tile stages for the pure subtrees it owns and walks the REST of what it
// owns -- nodes with at least one descendant it does not own -- one tree level per step, as the
// reference does (cedr_qlt.cpp:618-640), with one round of point-to-point messages per step:
// leaves->root, the records of my level-L nodes whose parent another rank owns go out and those
// of the level-L kids of my nodes come in; root->leaves the same messages run backwards. Between
// two ranks and one step there is at most one message per direction; its records are in DFS
// order, which both sides can compute from their own (partial) trees.
struct LevelStep {
  int32_t level = 0;               // tree level (leaf = 0) of this step's nodes and messages
  std::vector<int32_t> nodes;      // 3 per node of mine at this level: its slot, kid 0's, kid 1's
  struct Msg { int32_t peer, first, count; };  // 'count' records from entry 'first' of the slot list
  std::vector<int32_t> out_slots;  // my nodes of this level with a parent elsewhere, grouped by peer
  std::vector<int32_t> in_slots;   // other ranks' nodes of this level whose parent is mine
  std::vector<Msg> out_msgs, in_msgs;
};

struct Plan {
  int32_t my_rank = 0, nranks = 1;
  int32_t nlcl = 0;       // cells owned by this rank; their slots are 0..nlcl-1 in
                          // the reference's local cell order (NodeSets leaf offsets)
  int32_t nslots = 0;     // slots used by this plan
  std::vector<Stage> local;  // stages over this rank's pure subtrees, leaf-most first
  std::vector<Stage> top;    // replicated stages above the pure subtrees (empty if 1 rank)
  // The same replicated part as ONE plain tile over all boundary roots (inputs = the top input
  // slots in order, one output = the global root), when there are at most kApexMaxInputs of them:
  // the single-workgroup "apex" kernels sweep it up and down in one launch. Empty otherwise.
  Stage apex;
  static constexpr int32_t kApexMaxInputs = 64;
  int32_t root_slot = 0;     // slot of the global root's record on this rank

  // Boundary exchange (unused when nranks == 1). A boundary root is the root
  // of a maximal pure subtree.
  int32_t nb_max = 0;                    // max # boundary roots on any rank
  std::vector<int32_t> my_root_slots;    // slots of my boundary roots, my DFS order
  std::vector<int32_t> my_top_slots;     // the same roots' slots as inputs of 'top'
  std::vector<int32_t> gather_slots;     // [nranks*nb_max] top input slot of rank r's
                                         // j-th boundary root, or -1 for padding

  // Level mode (see LevelStep): 'top', 'apex' and the gather tables are empty; my_root_slots are
  // the roots of this rank's pure subtrees; root_slot holds the root's record on root_owner only
  // (the BFB reducer broadcasts it).
  bool level_mode = false;
  std::vector<LevelStep> steps;          // ascending level
  int32_t root_owner = 0;
};

struct Options {
  int32_t max_tile_leaves = 128;  // inputs per tile (bounded by the kernels' LDS use)
  bool micro = true;              // first local stage uses micro tiles (64 x <= 3 leaves)
  bool micro_all = true;          // ... and so does every later stage, over the previous one's outputs
  int32_t micro_cut = 3;          // local levels kept per micro tile of the FIRST stage (0 = keep all). Round 5: 3 instead of 2,
                                  // with micro_cut2 = 4 (profiles/r05_experiments.md): at 1.5M cells x 40 tracers the first-stage
                                  // root->leaves launch walks one level more per chunk (+0.03-0.04 ms), the leaves->root one writes
                                  // half the output lines (-0.01 to -0.02 ms), the second stage halves (-0.04 ms) and the third
                                  // disappears into the top tile: QLT::run -1 to -2 %, the two-stream step -1.5 to -5 %; neutral
                                  // at the 187.5k-cell shard
  int32_t micro_cut2 = 4;         // the same for the second stage (micro_all only; 0 = keep all; round 4: 3 instead of 2
                                  // -- the later stages' tiles keep one level more, the top tile one level fewer: QLT::run
                                  // -0.3 % at 1.5M cells, -2.5 % at the 187.5k-cell shard, profiles/r04_experiments.md 8): cutting
                                  // it moves tree levels from the second to the third stage, which
                                  // otherwise has only a handful of tiles, at no cost in stages
  // How the part of the tree above the ranks' pure subtrees is handled on more than one rank:
  // 0 = decide (level mode for partial trees and when the gathered boundary roots would outnumber
  // a rank's fair share of the cells several times over), 1 = one all-gather + replicated top,
  // 2 = level mode. Every rank must use the same value; the automatic rule only looks at numbers
  // every rank computes alike.
  int32_t exchange = 0;
  // The transport has no point-to-point messages (mpi::Comm::has_exchange() is false), so level mode
  // cannot run: a plan that would have chosen it (exchange == 0) takes the all-gather instead where
  // that is still correct (scattered ownership of a full tree: with a warning, once per process, since
  // every rank then holds and sweeps about the whole tree) and throws where it is not (tree::Node::rank
  // pins a node to another rank than the owner of everything below it; partial trees; exchange == 2).
  bool no_p2p = false;
  // The last local stages as one fused top tile (Stage::top) once at most top_max_inputs slots
  // are left below this rank's unfinished roots; every micro stage before it is then cut
  // (micro_cut2 from the second on): the levels a cut leaves out end up in the top tile's single
  // launch, not in an extra stage.
  bool fused_top = true;
  int32_t top_max_inputs = 2048;  // bounded by the top kernels' LDS use (4095 nodes of 56 B)
};

// 'leaf_slot[n]' gives, for every tree node n that is a leaf owned by my_rank,
// its local cell index (NodeSets offset); -1 elsewhere.
Plan build(const tree::FlatTree& t, int32_t my_rank, int32_t nranks,
           const std::vector<int32_t>& leaf_slot, const Options& opt);

}  // namespace plan
}  // namespace cedr

#endif
