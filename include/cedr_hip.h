/* cedr_hip.h -- C ABI of the MI355X (gfx950) CEDR limiter library, libcedr_hip.so.
 *
 * Two layers, both plain C (pointers, sizes, opaque handles; no C++ or torch
 * types cross this boundary; nothing throws across it):
 *
 *  (K) kernel level  cedr_hip_*  : the hand-written HIP kernels, one entry point
 *      per device lambda of the reference. These are what a
 *      `QLT<ExeSpace>` / `CAAS<ExeSpace>` specialisation inside the reference
 *      would call instead of its Kokkos::parallel_for bodies. All pointers are
 *      DEVICE pointers unless a parameter says "host". Every call enqueues work
 *      on `stream` and returns immediately.
 *
 *  (O) object level  cedr_*      : the cedr::CDR plugin protocol
 *      (reference cedr/cedr_cdr.hpp:16-112) flattened to C for non-C++ callers
 *      (E3SM's Fortran side binds COMPOSE this way) and for the Python tests.
 *      The C++ classes with the reference's own names live in
 *      compose_amd/csrc/cedr_{cdr,qlt,caas,bfb_tree_allreduce,tree,mpi}.hpp.
 *
 * Return value: 0 on success; a positive hipError_t from layer K; -1 from layer
 * O with the text available from cedr_last_error().
 */
#ifndef CEDR_HIP_H
#define CEDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* cedr_stream_t; /* == hipStream_t */

/* ------------------------------------------------------------------ layer K */

/* One tile: a subtree one workgroup sweeps in LDS (see DESIGN.md "Tiles").
 * Output k (k < nout) is local node outs[outs_off + k] and lives in slot out + k;
 * normally nout = 1 (the tile root). */
typedef struct {
  int32_t in0, nleaf, out, nint, kids_off, lvl_off, nlvl, micro_off, nout, outs_off;
} cedr_hip_tile;

/* One stage: tiles that are swept by one kernel launch (max_nleaf <= 256; the fused top tile, a stage of one plain
 * tile for cedr_hip_tree_top_up / cedr_hip_qlt_top: <= 2048).
 * Arrays are device memory. kids: 2 local node ids per internal node; lvlptr: per tile nlvl+1
 * offsets. micro != NULL marks a MICRO stage: the nleaf (<= 64) inputs of a
 * tile's LDS tree are subtrees of <= 3 consecutive cells, one per lane, handled
 * in registers; micro[tile.micro_off + i] = first cell relative to in0 (low
 * byte) | shape << 8, shape 0 = a, 1 = (a,b), 2 = ((a,b),c), 3 = (a,(b,c)). */
typedef struct {
  int32_t ntiles, max_nodes, max_nint, max_nleaf;
  const cedr_hip_tile* tiles;
  const uint16_t* kids;
  const int32_t* lvlptr;
  const uint16_t* micro;
  const uint16_t* outs;
  int32_t flags;  /* CEDR_STAGE_*: properties of the whole stage that let kernels skip a dependent load */
} cedr_hip_stage;
/* micro stage whose every tile has exactly 64 micro-subtrees: tile t's descriptors are micro[64*t ...], so a
 * workgroup can fetch them without waiting for its tile descriptor */
enum { CEDR_STAGE_UNIFORM64 = 1,
/* fused top tile: the first level's node j has the inputs 2j and 2j + 1 as its kids and there are nleaf/2 of
 * them, so its loads need not wait for the kid table */
       CEDR_STAGE_LVL0_PAIRS = 2 };
/* bits 16..31 of flags: launch-shape knobs read from the environment when the stage is put on the device (so that no
 * launch reads the environment): CEDR_STAGE_TAIL_SET says bits 16..22 hold CEDR_HIP_TAIL (workgroups per tail tile;
 * unset = the built-in default), bits 24..30 CEDR_HIP_TAIL_ROUNDS */
enum { CEDR_STAGE_TAIL_SET = 1 << 23 };
/* fused top tile, bits 8..15 of flags: 1 + the first level of the NARROW suffix of the tile's levels (every
 * level from there up has at most 64 nodes; one wave walks those without workgroup barriers); 0 = not given */

/* QLT record layout. Bulk data are stored field-major:
 *   l2r[field*stride + slot], r2l[field*stride + slot]
 * where `field` is the reference's index inside a slot record
 * (cedr_qlt_inl.hpp:14-58: 0 = rhom, trcr2bl2r(ti)+{0,1,2,3} ...). The
 * per-tracer arrays are indexed by BULK index (reference bidx2trcr order). */
typedef struct {
  int32_t nt, l2rndps, r2lndps, prefer_mass_con;
  int64_t stride;
  const int32_t* bidx_prob;  /* [nt] canonical problem type            */
  const int32_t* bidx_l2r;   /* [nt] first l2r field of the tracer     */
  const int32_t* bidx_r2l;   /* [nt] first r2l field of the tracer     */
  const uint8_t* l2r_op;     /* [l2rndps] 0 sum, 1 min, 2 max per field */
  int32_t flags;             /* bit 0: every tracer is shape preserving (s, st, cs, cst): the
                              * root->leaves sweep may use the kernel specialised for that */
  int32_t class_count[3];    /* tracers per problem class in bulk order -- shape preserving first, then t-only
                              * (consistent without shapepreserve), then nonnegative (the reference's bulk order,
                              * cedr_qlt.cpp:118-133): a mixed list is then swept class by class with
                              * class-specialised kernels. All zero = not given (one general kernel) */
} cedr_hip_qlt_layout;
enum { CEDR_QLT_ALL_SHAPEPRESERVE = 1,
/* set by the caller for the launch whose inputs are the CELLS (a plan's first local stage): below them no stage reads the
 * q bounds a t-only tracer's sweep hands down (cedr_qlt.cpp:549-560), so the class-specialised kernel does not write the
 * cells' bound rows -- unless CEDR_QLT_LEAF_BOUNDS asks for the reference's full post-run state of l2r / r2l */
       CEDR_QLT_LEAF_STAGE = 2, CEDR_QLT_LEAF_BOUNDS = 4,
/* mixed tracer lists: never sweep class by class (the general kernel for every launch); bits 8..23 of flags, if not 0:
 * the smallest launch, in tiles, that is swept class by class (default 2048) */
       CEDR_QLT_NO_CLASS_FORMS = 8 };

/* Leaves->root sweep of `nfields` independent fields over one stage: every
 * internal node gets op(kid0, kid1); only tile roots are written back.
 * Replaces QLT::l2r_combine_kid_data (cedr_qlt.cpp:339-430) for all levels
 * inside the stage's tiles, and the combine loop of
 * BfbTreeAllReducer::allreduce (cedr_bfb_tree_allreduce.cpp:114-124) when
 * zero_init != 0 (then a node is 0 + kid0 + kid1, as the reference computes). */
int cedr_hip_tree_up(const cedr_hip_stage* stage, int32_t nfields, const uint8_t* op,
                     int64_t stride, double* data, int zero_init, cedr_stream_t stream);
/* The same with the stage's INPUT rows taken from a device table of per-field base
 * pointers, in_rows[f][slot], instead of data[f*stride + slot] (outputs still go to
 * data): the first stage reading the caller's own [tracer][cell] arrays, i.e. no
 * set_Qm pack pass. Micro stages only. nrow (>= 3) = valid cells per
 * row: nothing is read outside [row, row + nrow). in_rows == NULL is cedr_hip_tree_up. */
int cedr_hip_tree_up_rows(const cedr_hip_stage* stage, int32_t nfields, const uint8_t* op,
                          int64_t stride, double* data, const double* const* in_rows, int32_t nrow,
                          int zero_init, cedr_stream_t stream);
/* The same with a promise: all_sums != 0 says every op[f] is 0 (a plain sum), which lets micro
 * stages run a build without the per-field rule (QLT without consistent-only tracers). */
int cedr_hip_tree_up_rows_sums(const cedr_hip_stage* stage, int32_t nfields, const uint8_t* op,
                               int64_t stride, double* data, const double* const* in_rows, int32_t nrow,
                               int zero_init, int all_sums, cedr_stream_t stream);

/* Root->leaves sweep over one stage: re-forms the tile's node sums in LDS,
 * takes the tile root's assigned mass from r2l, solves every node problem
 * (cedr_qlt_inl.hpp:120-203 via cedr_qlt.cpp:525-604) and writes the inputs'
 * masses. Replaces QLT::r2l_solve_qp for all levels inside the stage. */
int cedr_hip_qlt_down(const cedr_hip_qlt_layout* lay, const cedr_hip_stage* stage,
                      double* l2r, double* r2l, cedr_stream_t stream);
/* The same for the first stage in zero-copy mode: the cells' l2r rows are
 * in_rows[l2r field][cell] and their new masses go to out_rows[bulk index][cell]
 * (no get_Qm unpack pass). Both tables are device arrays; nrow as above; both NULL
 * is cedr_hip_qlt_down. */
int cedr_hip_qlt_down_rows(const cedr_hip_qlt_layout* lay, const cedr_hip_stage* stage,
                           double* l2r, double* r2l, const double* const* in_rows,
                           double* const* out_rows, int32_t nrow, cedr_stream_t stream);

/* Level mode (partial caller trees, ownership that does not follow subtrees): the nodes a rank
 * owns above its pure subtrees, one launch per tree level as in the reference (cedr_qlt.cpp:339-430
 * l2r_combine_kid_data, :525-604 r2l_solve_qp). nodes[3*j] = slot of node j, nodes[3*j+1..2] = its
 * kids' slots; data is field-major with `stride` doubles per field. */
int cedr_hip_level_up(const int32_t* nodes, int32_t n, int32_t nfields, const uint8_t* op, int64_t stride,
                      double* data, int zero_init, cedr_stream_t stream);
int cedr_hip_qlt_level_down(const cedr_hip_qlt_layout* lay, const int32_t* nodes, int32_t n, double* l2r,
                            double* r2l, cedr_stream_t stream);

/* QLT::root_compute (cedr_qlt.cpp:441-476). */
int cedr_hip_qlt_root(const cedr_hip_qlt_layout* lay, int32_t root_slot,
                      const double* l2r, double* r2l, cedr_stream_t stream);

/* Multi-rank runs, after the all-gather of the boundary roots' records: everything the replicated
 * top of the tree needs, in ONE single-workgroup launch (it replaces scatter + the top sweeps +
 * cedr_hip_qlt_root + copy; reference: the levels above the rank boundaries in QLT::run,
 * cedr_qlt.cpp:618-640, and root_compute :441-476). `apex` is a stage of exactly one plain tile
 * whose nleaf (<= 64) inputs are the boundary roots; recv holds the gathered records
 * [record][l2rndps]; src[i] = record of input i; this rank's nmine boundary roots are inputs
 * my_in[j] and get their assigned masses written to r2l slot my_slot[j]. */
int cedr_hip_qlt_apex(const cedr_hip_qlt_layout* lay, const cedr_hip_stage* apex, const double* recv,
                      const int32_t* src, int32_t nmine, const int32_t* my_in, const int32_t* my_slot,
                      double* r2l, cedr_stream_t stream);
/* The same for a plain sum of nfields fields (BfbTreeAllReducer): out[f] = the root's value. */
int cedr_hip_tree_apex_sum(const cedr_hip_stage* apex, const double* recv, const int32_t* src, int32_t nfields,
                           int zero_init, double* out, cedr_stream_t stream);

/* FUSED TOP. The last, latency-bound stages of a rank's forest as one plain tile `top` (a stage of
 * exactly one tile with up to 2048 inputs, several levels, one output per subtree root of the rank),
 * swept tracer-parallel in ONE launch per direction instead of one launch per stage.
 * cedr_hip_tree_top_up: leaves->root over `nfields` fields (any combine rules), one workgroup per 4
 * fields; writes the tile's outputs to their slots and, if sendbuf != NULL, packs the records of this
 * rank's nroots boundary roots (slots root_slots[j]) as sendbuf[j*nfields + f] for the all-gather --
 * the gather_slots launch fused in (cedr_qlt.cpp:432-439).
 * cedr_hip_qlt_top: root->leaves, one workgroup per tracer: re-forms the tile's sums, gets the outputs'
 * masses -- mode 0: from their r2l slots; mode 1 (single rank, the tile's one output is the global root):
 * by root_compute (cedr_qlt.cpp:441-476) on sums that include Qm_prev, so that no leaves->root launch over
 * the top tile is needed at all; mode 2: from the gathered boundary records through the apex tile, i.e.
 * cedr_hip_qlt_apex fused in (apex, recv, src, nmine, my_in, my_slot as there) -- and solves every node
 * problem of the tile (cedr_qlt.cpp:525-604), writing the inputs' masses to their r2l slots.
 * top_tile / apex_tile: HOST copies of the stages' single tile descriptors (top->tiles[0] is device memory;
 * the kernels take the descriptor by value so that their first memory access is already the kid table). */
int cedr_hip_tree_top_up(const cedr_hip_stage* top, const cedr_hip_tile* top_tile, int32_t nfields,
                         const uint8_t* op, int64_t stride, double* data, int zero_init, int32_t nroots,
                         const int32_t* root_slots, double* sendbuf, cedr_stream_t stream);
int cedr_hip_qlt_top(const cedr_hip_qlt_layout* lay, const cedr_hip_stage* top, const cedr_hip_tile* top_tile,
                     double* l2r, double* r2l, int mode, const cedr_hip_stage* apex,
                     const cedr_hip_tile* apex_tile, const double* recv, const int32_t* src,
                     int32_t nmine, const int32_t* my_in, const int32_t* my_slot, cedr_stream_t stream);

/* Boundary exchange helpers (replace the per-level MPI messages of
 * cedr_qlt.cpp:327-337,432-439,478-488,606-613):
 * gather : out[j*nfields + f] = data[f*stride + slots[j]], j < n
 * scatter: data[f*stride + slots[q]] = in[q*nfields + f] for slots[q] >= 0, q < n
 * copy   : data[f*stride + dst[j]] = data[f*stride + src[j]], j < n            */
int cedr_hip_gather_slots(const double* data, int64_t stride, int32_t nfields,
                          const int32_t* slots, int32_t n, double* out, cedr_stream_t stream);
int cedr_hip_scatter_slots(double* data, int64_t stride, int32_t nfields,
                           const int32_t* slots, int32_t n, const double* in, cedr_stream_t stream);
/* scatter with a field-major source: data[f*stride + slots[q]] = in[f*n + q] */
int cedr_hip_scatter_slots_fm(double* data, int64_t stride, int32_t nfields,
                              const int32_t* slots, int32_t n, const double* in,
                              cedr_stream_t stream);
int cedr_hip_copy_slots(double* data, int64_t stride, int32_t nfields, const int32_t* src,
                        const int32_t* dst, int32_t n, cedr_stream_t stream);

/* Bulk forms of QLT::DeviceOp::set_rhom / set_Qm / get_Qm
 * (cedr_qlt_inl.hpp:14-66) for n local cells; caller arrays are [tracer][cell]
 * in the CALLER's tracer order; trcr_* arrays are indexed by caller tracer. */
int cedr_hip_qlt_set_rhom(int64_t stride, int32_t n, const double* rhom, double* l2r,
                          cedr_stream_t stream);
int cedr_hip_qlt_set_Qm(int32_t nt, const int32_t* trcr_prob, const int32_t* trcr_l2r,
                        int64_t stride, int32_t n, const double* Qm, const double* Qm_min,
                        const double* Qm_max, const double* Qm_prev, double* l2r,
                        cedr_stream_t stream);
int cedr_hip_qlt_get_Qm(int32_t nt, const int32_t* trcr_r2l, int64_t stride, int32_t n,
                        const double* r2l, double* Qm_out, cedr_stream_t stream);

/* CAAS::reduce_locally (cedr_caas.cpp:129-201), own-reducer branch, fused:
 * clips Qm (read from qm_src, or from d if NULL; written to d), and forms the
 * four per-tracer sums. Stage 1 writes per-chunk partials, stage 2
 * (cedr_hip_caas_sum_partials) adds them in fixed order into send[4*nt] =
 * [sum clip | sum Qm_prev or Qm | sum Qm_min | sum Qm_max].
 * d is the reference's SoA buffer (cedr_caas_inl.hpp:22-34).
 * partials must hold 4*nt*cedr_hip_caas_nchunks(n) doubles. */
int32_t cedr_hip_caas_nchunks(int32_t n);
int cedr_hip_caas_reduce_local(const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                               double* d, const double* qm_src, double* partials,
                               cedr_stream_t stream);
/* The same on separate [tracer][cell] arrays (row stride n) instead of the one SoA
 * buffer: a caller that already holds Qm, Qm_min, Qm_max, Qm_prev in that layout is
 * limited IN PLACE (qm_in == qm_out), with no pack / unpack pass.
 * qm_prev may be NULL when need_conserve is 0. */
int cedr_hip_caas_reduce_local_arrays(const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                                      const double* qm_in, double* qm_out, const double* qm_min,
                                      const double* qm_max, const double* qm_prev, double* partials,
                                      cedr_stream_t stream);
int cedr_hip_caas_sum_partials(int32_t nt, int32_t n, const double* partials, double* send,
                               cedr_stream_t stream);
/* User-reducer branch (cedr_caas.cpp:137-168): per-group partial sums in the
 * reference's send layout, send[nlclaccum*f + bi]. */
int cedr_hip_caas_reduce_local_accum(const int32_t* probs, int32_t nt, int32_t n,
                                     int need_conserve, int32_t n_accum, double* d,
                                     const double* qm_src, double* send, cedr_stream_t stream);
/* The production pairing, fused: CAAS whose UserAllReducer is the library's own BFB tree reducer over groups
 * of n_accum cells. The user-reducer reduce pass (cedr_caas.cpp:137-168) AND the leaf fill + first
 * leaves->root stage of BfbTreeAllReducer::allreduce (cedr_bfb_tree_allreduce.cpp:91-124) in one launch:
 * the group sums never reach HBM; the tile outputs of `bfb_stage` (the reducer's first local stage: micro
 * tiles whose leaf i is the cell group i, fields in CAAS's send order q*nt + tracer, q = clip, term, min,
 * max) are written to bfb_data[field*bfb_stride + slot]. tile_nleaves[t] = leaves under tile t (device
 * array); max_tile_leaves their maximum. The tiles must cover the n/n_accum local leaves exactly once.
 * cedr_hip_caas_bfb_fusable: 1 if such a stage / group size can be launched. Groups of 2, 4, 8, 16 or 32 cells on
 * 16-byte aligned rows (aligned16: n even, d and qm_src 16-byte aligned) are summed in registers; any other
 * group size parks a tile's cells in LDS (32 B per cell: up to ~2000 cells per tile); flags: CEDR_CAAS_BFB_PARK
 * takes the LDS form for every group size (tests). */
enum { CEDR_CAAS_BFB_PARK = 1 };
int cedr_hip_caas_bfb_fusable(const cedr_hip_stage* bfb_stage, int32_t n_accum, int32_t max_tile_leaves,
                              int aligned16, int flags);
int cedr_hip_caas_reduce_local_bfb(const int32_t* probs, int32_t nt, int32_t n, int need_conserve,
                                   int32_t n_accum, double* d, const double* qm_src,
                                   const cedr_hip_stage* bfb_stage, const int32_t* tile_nleaves,
                                   int32_t max_tile_leaves, int64_t bfb_stride, double* bfb_data, int flags,
                                   cedr_stream_t stream);
/* CAAS::finish_locally (cedr_caas.cpp:211-253). recv[4*nt]. */
int cedr_hip_caas_finish_local(int32_t nt, int32_t n, double* d, const double* recv,
                               cedr_stream_t stream);
int cedr_hip_caas_finish_local_arrays(int32_t nt, int32_t n, double* qm, const double* qm_min,
                                      const double* qm_max, const double* recv, cedr_stream_t stream);
/* Bulk CAAS::DeviceOp::set_Qm / get_Qm (cedr_caas_inl.hpp:22-42) are plain
 * copies into/out of d because the layouts coincide: use hipMemcpyAsync. */

/* Optional per-launch timing with HIP events recorded on the launch stream
 * (the analogue of the reference's COMPOSE_QLT_TIME timers, cedr_qlt.cpp:19-76).
 * start clears earlier records; stop synchronises and returns the record count;
 * get returns one record: kernel id, an int describing the launch (tiles or
 * cells) and the elapsed milliseconds. */
void cedr_hip_profile_start(void);
int cedr_hip_profile_stop(void);
int cedr_hip_profile_get(int i, int* id, int64_t* aux, float* ms);
const char* cedr_hip_profile_name(int id);
/* sha256 (hex) of the kernel source file this library was built from (compose_amd/csrc/cedr_kernels.hip):
 * stamped into benchmark lines and profiles so that a measurement can be matched to its code. */
const char* cedr_hip_kernels_sha256(void);

/* ------------------------------------------------------------------ layer O */
/* Handles are opaque. Creation functions return NULL on failure. */
const char* cedr_last_error(void);

/* -- cedr::mpi::Parallel (reference cedr/cedr_mpi.hpp:17-37) ---------------- */
void* cedr_parallel_create_self(void);
/* One process per GPU over RCCL; unique_id from cedr_rccl_get_unique_id on one
 * rank, distributed by the launcher. */
int cedr_rccl_get_unique_id(void* unique_id_128B);
void* cedr_parallel_create_rccl(const void* unique_id_128B, int rank, int size);
/* Rehearsal of an N-rank job on ONE GPU (measurements, tests): rank 0 of pretended_size ranks over `inner`, a
 * Parallel of one real rank (cedr_parallel_create_self or cedr_parallel_create_rccl(id, 0, 1)); collectives go
 * through `inner` for real, the other ranks' blocks of an all-gather are copies of this rank's. */
void* cedr_parallel_create_rehearsal(void* inner_parallel, int pretended_size);
/* Peer flags (behind a knob; RCCL stays the default): the two KB-sized collectives of a step as stream-ordered stores
 * into the peers' IPC-mapped mailboxes + one hipStreamWaitValue64 per collective (compose_amd/csrc/cedr_mpi.hpp; replaces
 * the same reference calls as the RCCL backend: cedr_qlt.cpp:327-337,432-439, cedr_caas.cpp:203-209). Every rank (one
 * process per GPU of ONE node, at most 64) creates its mailbox -- `handle_out` receives CEDR_PEER_HANDLE_BYTES bytes --
 * the launcher gathers all ranks' handles in rank order, and cedr_parallel_create_peer_flags maps them. The Parallel
 * keeps the mailbox alive; cedr_peer_mailbox_destroy only drops the caller's reference. */
#define CEDR_PEER_HANDLE_BYTES 96
void* cedr_peer_mailbox_create(int size, size_t slot_doubles, void* handle_out);
void cedr_peer_mailbox_destroy(void* mailbox);
void* cedr_parallel_create_peer_flags(void* my_mailbox, const void* handles_in_rank_order, int rank, int size);
/* Host-application transport (e.g. GPU-aware MPI). Buffers are device memory. */
typedef int (*cedr_allgather_fn)(void* user, const double* send, double* recv, size_t count,
                                 cedr_stream_t stream);
typedef int (*cedr_allreduce_sum_fn)(void* user, const double* send, double* recv, size_t count,
                                     cedr_stream_t stream);
void* cedr_parallel_create_callbacks(int rank, int size, cedr_allgather_fn ag,
                                     cedr_allreduce_sum_fn ar, void* user);
/* The same plus one round of point-to-point messages (the reference's isend/irecv/waitall of one
 * tree level, cedr_qlt.cpp:327-337,432-439): op i sends (sends[i] != 0) or receives counts[i]
 * doubles at device pointer bufs[i] to / from rank peers[i]; all ops of a call are posted together;
 * at most one send and one receive per peer. Needed only by plans in level mode (partial trees). */
typedef int (*cedr_exchange_fn)(void* user, int nops, const int* peers, double* const* bufs,
                                const size_t* counts, const int* sends, cedr_stream_t stream);
void* cedr_parallel_create_callbacks_p2p(int rank, int size, cedr_allgather_fn ag,
                                         cedr_allreduce_sum_fn ar, cedr_exchange_fn xch, void* user);
/* One round of point-to-point messages through a Parallel's own transport (what level-mode plans
 * use between tree levels); arguments as cedr_exchange_fn. */
int cedr_parallel_exchange(void* p, int nops, const int* peers, double* const* bufs, const size_t* counts,
                           const int* sends, cedr_stream_t stream);
/* The two collectives of a Parallel's transport, enqueued on `stream` over device buffers (what QLT::run / CAAS::run /
 * the BFB reducer call between their launches; for transport tests and latency measurements): recv[r*count ..] = rank r's
 * send[0 .. count) / recv[i] = sum over ranks of send[i]. */
int cedr_parallel_allgather(void* p, const double* send, double* recv, size_t count, cedr_stream_t stream);
int cedr_parallel_allreduce_sum(void* p, const double* send, double* recv, size_t count, cedr_stream_t stream);
int cedr_parallel_rank(void* p);
int cedr_parallel_size(void* p);
void cedr_parallel_free(void* p);

/* -- caller tree (reference cedr/cedr_tree_caller.hpp:12-30) ---------------- */
/* Binary tree from arrays: kid0/kid1 = -1 at leaves; rank and cellidx are read
 * at leaves (rank >= 0 at an internal node pins that node's rank). */
void* cedr_tree_create(int nnodes, int root, const int* kid0, const int* kid1, const int* rank,
                       const long* cellidx);
/* The same with tree::Node::level (cedr_tree_caller.hpp:20-22) for PARTIAL trees: a rank passes only
 * the nodes it owns, their kids and the chain of ancestors. level[i] >= 0 is the node's level (leaf
 * = 0), required wherever a subtree is left out; a node may then have one kid (kid1 = -1) or none
 * although it is not a cell; every node needs its rank. level == NULL: cedr_tree_create. */
void* cedr_tree_create_levels(int nnodes, int root, const int* kid0, const int* kid1, const int* rank,
                              const long* cellidx, const int* level);
/* What rank `my_rank` needs of a full tree, as a new partial tree with levels and ranks filled in
 * (what E3SM builds directly for large rank counts). */
void* cedr_tree_prune(void* tree, int my_rank);
int cedr_tree_nnodes(void* tree);
/* tree::make_tree_over_1d_mesh / oned::make_tree (cedr_tree.cpp:350-420). */
void* cedr_tree_make_1d(int ncells, int nranks, int pseudorandom, int imbalanced);
void cedr_tree_free(void* tree);

/* -- tree::analyze (reference cedr/cedr_tree.cpp:215-231), host only -------- */
void* cedr_tree_analyze(void* tree, int ncells, int my_rank);
void cedr_ns_free(void* ns);
int cedr_ns_nslots(void* ns);
int cedr_ns_nnode(void* ns);
int cedr_ns_nlevels(void* ns);
int cedr_ns_level_size(void* ns, int il);
int cedr_ns_level_nme(void* ns, int il);
int cedr_ns_level_nkids(void* ns, int il);
void cedr_ns_nodes(void* ns, int* out_nnode_x7); /* rank,id,parent,nkids,kid0,kid1,offset */
void cedr_ns_level_nodes(void* ns, int il, int* out);
void cedr_ns_level_msgs(void* ns, int il, int which /*0 me, 1 kids*/, int* out_n_x3);
int cedr_ns_check_comm(void* ns);

/* -- cedr::qlt::QLT (reference cedr/cedr_qlt.hpp:129-197) ------------------- */
void* cedr_qlt_create(void* parallel, int ncells, void* tree, int prefer_mass_con);
void cedr_qlt_free(void* qlt);
int cedr_qlt_declare_tracer(void* qlt, int problem_type, int rhomidx);
int cedr_qlt_end_tracer_declarations(void* qlt);
int cedr_qlt_get_buffers_sizes(void* qlt, size_t* buf1, size_t* buf2); /* in doubles */
/* What this build touches (ndps * stride doubles, about half of the reference's figure above):
 * enough for set_buffers; the reference-sized query stays for callers written against it. */
int cedr_qlt_get_buffers_sizes_tight(void* qlt, size_t* buf1, size_t* buf2);
int cedr_qlt_set_buffers(void* qlt, double* buf1, double* buf2);       /* device memory */
int cedr_qlt_finish_setup(void* qlt);
int cedr_qlt_get_problem_type(void* qlt, int tracer_idx); /* < 0 on error */
int cedr_qlt_get_num_tracers(void* qlt);
int cedr_qlt_nlclcells(void* qlt);
int cedr_qlt_get_owned_glblcells(void* qlt, long* gcis);
int cedr_qlt_gci2lci(void* qlt, int gci); /* < 0 on error */
int cedr_qlt_run(void* qlt);
int cedr_qlt_set_stream(void* qlt, cedr_stream_t stream);
/* A hipEvent_t that run() records on its stream just before its first-stage root->leaves sweep (the
 * long kernel of a run that is bound by fp64 issue, not by HBM): let another stream wait for it
 * (cedr_stream_wait_event = hipStreamWaitEvent) to run independent memory-bound work, e.g. CAAS,
 * side by side with that sweep. Owned by the QLT. */
void* cedr_qlt_sweep_down_event(void* qlt);
/* the same for the end of run()'s first-stage leaves->root sweep: what follows until the first-stage root->leaves
 * sweep is a short series of small, latency-bound launches that leave most of the GPU idle */
void* cedr_qlt_sweep_mid_event(void* qlt);
int cedr_stream_wait_event(cedr_stream_t stream, void* event);
/* What finish_setup did about where its own l2r / r2l buffers live (CEDR_HIP_PLACE_TRIES = allocations tried, default 1):
 * "tries N; l2r ms: t0 t1 ... kept i; r2l ms: ... kept j" (time of the first-stage sweeps over each candidate). Valid
 * while the QLT lives. */
const char* cedr_qlt_placement_report(void* qlt);
/* DeviceOp for C callers: the values a device kernel needs to do set/get itself. */
typedef struct {
  double* l2r_data; double* r2l_data; int64_t stride;
  const int* trcr2prob; const int* trcr2bl2r; const int* trcr2br2l;
} cedr_qlt_device_op;
int cedr_qlt_get_device_op(void* qlt, cedr_qlt_device_op* out);
/* Bulk set_rhom / set_Qm / get_Qm over all local cells; device arrays
 * [tracer][cell] in caller tracer order. */
int cedr_qlt_set_rhom_bulk(void* qlt, const double* rhom);
int cedr_qlt_set_Qm_bulk(void* qlt, const double* Qm, const double* Qm_min,
                         const double* Qm_max, const double* Qm_prev);
int cedr_qlt_get_Qm_bulk(void* qlt, double* Qm_out);
/* QLT::MetaData (cedr_qlt.hpp:46-109): five arrays of nt ints, three of 7. */
int cedr_qlt_metadata(void* qlt, int* trcr2prob, int* bidx2trcr, int* trcr2bidx,
                      int* trcr2bl2r, int* trcr2br2l, int* prob2trcrptr, int* prob2bl2r,
                      int* prob2br2l);
/* Zero-copy mode: run() reads the caller's device arrays (rhom[cell]; the others
 * [tracer][cell], caller tracer order) and writes the limited masses to Qm_out, which
 * may be Qm itself. No set_rhom / set_Qm / get_Qm pass; NULL rhom switches back. */
int cedr_qlt_bind_arrays(void* qlt, const double* rhom, const double* Qm, const double* Qm_min,
                         const double* Qm_max, const double* Qm_prev, double* Qm_out);
void* cedr_qlt_nodesets(void* qlt); /* borrowed NodeSets handle for cedr_ns_* */
/* Execution plan (host copy), for inspection and for the CPU multi-rank tests.
 * cedr_plan_create builds one without touching a GPU; cedr_qlt_plan borrows
 * the one a QLT runs (valid while the QLT lives; do not free it).
 * summary[8] = {nlcl, nslots, #local stages, #top stages, root_slot, nb_max,
 *               #my boundary roots, stride}. */
/* max_tile_leaves: 0 = defaults; > 0 sets the tile size; < 0 disables micro
 * tiles (and -n, n > 1, sets the tile size n as well). */
void* cedr_plan_create(void* tree, int ncells, int my_rank, int nranks, int max_tile_leaves);
void cedr_plan_free(void* plan);
void* cedr_qlt_plan(void* qlt);
int cedr_plan_summary(void* plan, int64_t* summary8);
/* sizes[6] = {ntiles, max_nodes, #internal nodes, #lvlptr entries, #micro entries,
 *             #outs entries}; tiles are 10 ints each (cedr_hip_tile).
 * top: 0 = local stages, 1 = replicated top stages, 2 = the apex tile (idx 0; an error if the
 * plan has none). */
int cedr_plan_stage_sizes(void* plan, int top, int idx, int* sizes6);
int cedr_plan_stage_micro(void* plan, int top, int idx, uint16_t* micro, uint16_t* outs);
/* 1 if the stage is the fused top tile (one plain tile of up to 2048 inputs over what is left of the
 * rank's forest, swept by cedr_hip_tree_top_up / cedr_hip_qlt_top), 0 if not, -1 if there is no such stage */
int cedr_plan_stage_is_top(void* plan, int top, int idx);
int cedr_plan_stage(void* plan, int top, int idx, int* tiles_x10, uint16_t* kids, int* lvlptr);
/* my_root_slots[#my roots], my_top_slots[#my roots], gather_slots[nranks*nb_max] */
int cedr_plan_exchange(void* plan, int* my_root_slots, int* my_top_slots, int* gather_slots);

/* Level mode (partial trees / scattered ownership; plan::LevelStep in csrc/cedr_plan.hpp).
 * info[3] = {level mode?, #steps, rank owning the root}; sizes[6] of step i = {tree level, #nodes,
 * #out slots, #in slots, #out messages, #in messages}; nodes are (slot, kid 0 slot, kid 1 slot),
 * messages (peer, first entry of the slot list, count). */
int cedr_plan_levels(void* plan, int* info3);
int cedr_plan_level_sizes(void* plan, int i, int* sizes6);
int cedr_plan_level(void* plan, int i, int* nodes_x3, int* out_slots, int* in_slots, int* out_msgs_x3,
                    int* in_msgs_x3);

/* -- cedr::caas::CAAS (reference cedr/cedr_caas.hpp:51-96) ------------------ */
/* UserAllReducer (cedr_caas.hpp:27-49) as a callback; NULL = own all-reduce. */
typedef int (*cedr_user_allreduce_fn)(void* user, double* sendbuf, double* rcvbuf, int nlocal,
                                      int nfld);
void* cedr_caas_create(void* parallel, int nlclcells, cedr_user_allreduce_fn reducer,
                       int n_accum_in_place, void* user);
/* The production pairing: CAAS whose UserAllReducer is the library's own BFB tree reducer `bfb` (a
 * cedr_bfb_create handle over a tree whose leaf i is the group of cells [i*n_accum, (i+1)*n_accum) of this
 * rank, nfield = 4 * #tracers; it stays the caller's to free, the CAAS shares it): C++ callers pass a
 * caas::BfbUserAllReducer. run() then fuses the reduce pass with the reducer's first tree stage
 * (cedr_hip_caas_reduce_local_bfb) where the tree allows it -- cedr_caas_fused_with_bfb says, after
 * finish_setup -- and otherwise runs the two one after the other; the results are bit-identical either way
 * and to a cedr_caas_create whose callback calls cedr_bfb_allreduce(bfb, send, recv, 1). */
void* cedr_caas_create_bfb(void* parallel, int nlclcells, void* bfb, int n_accum_in_place);
int cedr_caas_fused_with_bfb(void* caas);
void cedr_caas_free(void* caas);
int cedr_caas_declare_tracer(void* caas, int problem_type, int rhomidx);
int cedr_caas_end_tracer_declarations(void* caas);
int cedr_caas_get_buffers_sizes(void* caas, size_t* buf1, size_t* buf2);
int cedr_caas_set_buffers(void* caas, double* buf1, double* buf2);
int cedr_caas_finish_setup(void* caas);
int cedr_caas_get_problem_type(void* caas, int tracer_idx);
int cedr_caas_get_num_tracers(void* caas);
int cedr_caas_run(void* caas);
int cedr_caas_set_stream(void* caas, cedr_stream_t stream);
int cedr_caas_set_Qm_source(void* caas, const double* qm_src);
/* Zero-copy mode: run() limits the caller's [tracer][cell] arrays in place (Qm is
 * overwritten with the result); NULL Qm returns to the internal buffer. */
int cedr_caas_bind_arrays(void* caas, double* Qm, const double* Qm_min, const double* Qm_max,
                          const double* Qm_prev);
typedef struct { double* d; int nlclcells; int nt; int need_conserve; } cedr_caas_device_op;
int cedr_caas_get_device_op(void* caas, cedr_caas_device_op* out);
int cedr_caas_get_recv(void* caas, const double** recv_dev);

/* -- cedr::BfbTreeAllReducer (reference cedr/cedr_bfb_tree_allreduce.hpp) ---- */
void* cedr_bfb_create(void* parallel, void* tree, int nleaf, int nfield);
void cedr_bfb_free(void* bfb);
int cedr_bfb_nlocal(void* bfb);
int cedr_bfb_allreduce(void* bfb, const double* send, double* recv, int transpose);
/* All work of allreduce is enqueued on this stream (default: the null stream); a reducer
 * used inside CAAS::run must use CAAS's stream. */
int cedr_bfb_set_stream(void* bfb, cedr_stream_t stream);

/* -- cedr::local scalar solvers (reference cedr/cedr_local.hpp:24-60) --------- */
/* The same `__host__ __device__` functions of compose_amd/csrc/cedr_local.hpp, evaluated on
 * the HOST (cedr_local_*) or, one problem per GPU thread, on the DEVICE (cedr_hip_local_batch).
 * kind: 0 solve_1eq_bc_qp, 1 solve_1eq_bc_qp_2d, 2 caas, 3 solve_1eq_nonneg(least_squares),
 * 4 solve_1eq_nonneg(caas). flags (kinds 1, 2): bit 0 = clip, bit 1 = early_exit_on_tol.
 * Problem p of a batch: in[p*(5n+1) ..] = w[n], a[n], xlo[n], xhi[n], y[n], b;
 * x[p*n ..] receives the solution and info[p] the function's return value (0 for caas). */
enum { CEDR_LOCAL_BC_QP = 0, CEDR_LOCAL_BC_QP_2D = 1, CEDR_LOCAL_CAAS = 2,
       CEDR_LOCAL_NONNEG_LS = 3, CEDR_LOCAL_NONNEG_CAAS = 4 };
int cedr_local_solve(int kind, int n, int flags, int max_its, const double* w, const double* a, double b,
                     const double* xlo, const double* xhi, const double* y, double* x);
int cedr_hip_local_batch(int kind, int nprob, int n, int flags, int max_its, const double* in_dev,
                         double* x_dev, int32_t* info_dev, cedr_stream_t stream);

/* The node problem of the root->leaves sweep (reference cedr_qlt_inl.hpp:120-203 solve_node_problem),
 * exactly as the sweep kernels evaluate it (cedr_node_math.hpp), one problem per GPU thread: a direct
 * handle for parity tests. Problem p: in[13p ..] = rhom, pd[3], Qm, rhom0, k0d[3], rhom1, k1d[3];
 * out[2p ..] = Qm0, Qm1. pair != 0 solves problems 2i and 2i+1 with the lockstep form
 * (node_problems_mass<2>; shape-preserving problem types only). */
int cedr_hip_node_problem_batch(int problem_type, int nprob, int prefer_mass_con, int pair, const double* in_dev,
                                double* out_dev, cedr_stream_t stream);

/* -- device memory helpers for hosts without their own HIP bindings ---------- */
int cedr_device_count(void);
void* cedr_malloc(size_t bytes);
void cedr_free(void* p);
int cedr_memcpy_h2d(void* dst, const void* src, size_t bytes);
int cedr_memcpy_d2h(void* dst, const void* src, size_t bytes);
int cedr_memcpy_d2d_async(void* dst, const void* src, size_t bytes, cedr_stream_t stream);
int cedr_device_synchronize(void);
/* Stream-ordered forms: a host application that stages a collective through host memory must work
 * on the stream it is handed (cedr_allgather_fn / cedr_allreduce_sum_fn), not on the null stream. */
int cedr_stream_synchronize(cedr_stream_t stream);
int cedr_memcpy_h2d_async(void* dst, const void* src, size_t bytes, cedr_stream_t stream);
int cedr_memcpy_d2h_async(void* dst, const void* src, size_t bytes, cedr_stream_t stream);
/* A stream whose kernels run only on the compute units whose bit is set in mask[0 .. nwords) (bit i of word w =
 * compute unit 32w + i; hipExtStreamCreateWithCUMask). For a caller that runs two CDRs of a step side by
 * side -- QLT::run's root->leaves sweep is bound by instruction issue, CAAS::run by HBM -- and wants each on
 * its own share of the machine (set_stream of either CDR takes the handle). */
int cedr_stream_create_cu_mask(const uint32_t* mask, int32_t nwords, cedr_stream_t* stream);
int cedr_stream_destroy(cedr_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* CEDR_HIP_H */
