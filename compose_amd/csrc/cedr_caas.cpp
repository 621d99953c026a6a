// Host side of CAAS for MI355X; reference cedr/cedr_caas.cpp cited per function.
#include "cedr_caas.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include <cstdlib>

#include "../../include/cedr_hip.h"
#include "cedr_bfb_tree_allreduce.hpp"
#include "cedr_device.hpp"
#include "cedr_util.hpp"

namespace cedr {
namespace caas {

#define cedr_k_check(call) do {                                         \
    const int _e_ = (call);                                             \
    cedr_throw_if(_e_ != 0, "kernel launch failed with HIP error " << _e_ << " in " #call); \
  } while (0)

struct CAASHip::Impl {
  hipStream_t stream = nullptr;
  dev::Buffer<Int> probs;
  dev::Buffer<Real> d_own, buf2_own, partials;
  Real* send = nullptr;
  Real* recv = nullptr;
  size_t send_sz = 0, recv_sz = 0;
  const Real* qm_src = nullptr;
  // bind_arrays: the caller's [tracer][cell] arrays, limited in place (null = internal buffer)
  Real* b_qm = nullptr;
  const Real* b_min = nullptr;
  const Real* b_max = nullptr;
  const Real* b_prev = nullptr;
  // Fused with the library's BFB reducer (caas::BfbUserAllReducer): the reducer and its first stage
  const BfbTreeAllReducerHip* bfb = nullptr;
  BfbTreeAllReducerHip::FirstStage fs{};
  int bfb_flags = 0;  // CEDR_CAAS_BFB_*
};

BfbUserAllReducer::BfbUserAllReducer (const std::shared_ptr<BfbTreeAllReducerHip>& bfb, int n_accum_in_place)
  : bfb_(bfb), nacc_(n_accum_in_place)
{
  cedr_throw_if( ! bfb, "BfbUserAllReducer needs a BfbTreeAllReducer");
  cedr_throw_if(n_accum_in_place < 1, "n_accum_in_place must be at least 1");
}

int BfbUserAllReducer::operator() (const mpi::Parallel&, Real* sendbuf, Real* rcvbuf, int nlocal, int nfld,
                                   MPI_Op) const {
  if (nlocal != bfb_->nlocal() || nfld != bfb_->nfield()) return -1;
  bfb_->allreduce(sendbuf, rcvbuf, true);  // CAAS emits (field, group): the transposed layout
  return 0;
}

CAASHip::CAASHip (const mpi::Parallel::Ptr& p, const Int nlclcells,
                  const UserAllReducer::Ptr& uar)
  : p_(p), user_reducer_(uar), finished_setup_(false), m_(new Impl())
{
  o.nlclcells_ = nlclcells;
  o.nrhomidxs_ = 0;
  o.nt_ = 0;
  o.need_conserve_ = false;
  o.d_ = nullptr;
  cedr_throw_if(nlclcells == 0, "CAAS does not support 0 cells on a rank.");  // :46
  tracer_decls_.reset(new std::vector<int>());
}

CAASHip::~CAASHip () {}

void CAASHip::declare_tracer (int problem_type, const Int& rhomidx) {  // :50-59
  cedr_throw_if( ! tracer_decls_, "end_tracer_declarations was already called; "
                 "it is an error to call declare_tracer now.");
  cedr_throw_if( ! (problem_type & ProblemType::shapepreserve),
                "CAAS does not support ! shapepreserve yet.");
  cedr_throw_if(rhomidx > 0, "rhomidx > 0 is not supported yet.");
  tracer_decls_->push_back(problem_type);
  if (problem_type & ProblemType::conserve) o.need_conserve_ = true;
  o.nrhomidxs_ = std::max(o.nrhomidxs_, rhomidx + 1);
}

void CAASHip::end_tracer_declarations () {  // :61-74
  cedr_throw_if( ! tracer_decls_, "end_tracer_declarations was already called.");
  cedr_throw_if(tracer_decls_->size() == 0, "#tracers is 0.");
  cedr_throw_if(o.nrhomidxs_ == 0, "#rhomidxs is 0.");
  probs_h_.assign(tracer_decls_->begin(), tracer_decls_->end());
  o.nt_ = static_cast<Int>(probs_h_.size());
  tracer_decls_.reset();
}

void CAASHip::get_buffers_sizes (size_t& buf1, size_t& buf2, size_t& buf3) {  // :76-84
  cedr_throw_if(tracer_decls_ != nullptr, "call end_tracer_declarations first");
  const size_t e = o.need_conserve_ ? 1 : 0;
  const size_t nt = probs_h_.size();
  const size_t nslots = 4*nt;
  buf1 = o.nlclcells_*((3 + e)*nt + 1);
  cedr_throw_if(user_reducer_ && o.nlclcells_ % user_reducer_->n_accum_in_place() != 0,
                "nlclcells must be a multiple of n_accum_in_place");
  buf2 = nslots*(user_reducer_ ? (o.nlclcells_/user_reducer_->n_accum_in_place()) : 1);
  buf3 = nslots;
}

void CAASHip::get_buffers_sizes (size_t& buf1, size_t& buf2) {  // :86-91
  size_t buf3;
  get_buffers_sizes(buf1, buf2, buf3);
  buf2 += buf3;
}

void CAASHip::set_buffers (Real* buf1, Real* buf2) {  // :93-100
  size_t s1, s2, s3;
  get_buffers_sizes(s1, s2, s3);
  o.d_ = buf1;
  m_->send = buf2;
  m_->send_sz = s2;
  m_->recv = buf2 + s2;
  m_->recv_sz = s3;
}

void CAASHip::finish_setup () {  // :102-116
  cedr_throw_if(tracer_decls_ != nullptr, "call end_tracer_declarations first");
  if ( ! m_->probs.size()) m_->probs.upload(probs_h_);
  // The production pairing: fuse the reduce pass with the first stage of the reducer's tree when the reducer is the
  // library's own, its leaves are this rank's cell groups in order, and the tiles fit the kernel.
  // (decided once the buffers are known: here for caller memory, below for the CDR's own)
  const bool own_memory = m_->recv_sz == 0;
  const bool aligned16 = o.nlclcells_ % 2 == 0 && (own_memory || reinterpret_cast<uintptr_t>(o.d_) % 16 == 0);
  m_->bfb = nullptr;
  if (const BfbUserAllReducer* const br = dynamic_cast<const BfbUserAllReducer*>(user_reducer_.get())) {
    const char* const e = std::getenv("CEDR_HIP_CAAS_BFB_FUSE");
    // CEDR_HIP_CAAS_BFB_P2=0: the LDS form of the fused kernel also for groups of 2, 4, ..., 32 cells
    const char* const e2 = std::getenv("CEDR_HIP_CAAS_BFB_P2");
    m_->bfb_flags = (e2 && std::atoi(e2) == 0) ? CEDR_CAAS_BFB_PARK : 0;
    const BfbTreeAllReducerHip& b = *br->reducer();
    const int na = br->n_accum_in_place();
    if (( ! e || std::atoi(e) != 0) && b.nfield() == 4*static_cast<Int>(probs_h_.size()) &&
        static_cast<int64_t>(b.nlocal())*na == o.nlclcells_ && b.first_stage_for_fusion(m_->fs) &&
        cedr_hip_caas_bfb_fusable(m_->fs.stage, na, m_->fs.max_tile_leaves, aligned16, m_->bfb_flags))
      m_->bfb = &b;
  }
  if (m_->recv_sz == 0) {
    size_t s1, s2, s3;
    get_buffers_sizes(s1, s2, s3);
    if (m_->bfb) s2 = 0;  // fused: the per-group partial sums are never stored
    m_->d_own.alloc_zero(s1);
    m_->buf2_own.alloc_zero(s2 + s3);
    o.d_ = m_->d_own.data();
    m_->send = m_->buf2_own.data();
    m_->send_sz = s2;
    m_->recv = m_->send + s2;
    m_->recv_sz = s3;
  }
  if ( ! user_reducer_)
    m_->partials.alloc_zero(4*probs_h_.size()*static_cast<size_t>(cedr_hip_caas_nchunks(o.nlclcells_)));
  finished_setup_ = true;
}

int CAASHip::get_problem_type (const Int& tracer_idx) const {
  cedr_throw_if(tracer_idx < 0 || tracer_idx >= static_cast<Int>(probs_h_.size()),
                "tracer_idx is out of bounds: " << tracer_idx);
  return probs_h_[tracer_idx];
}

Int CAASHip::get_num_tracers () const { return static_cast<Int>(probs_h_.size()); }

const CAASHip::DeviceOp& CAASHip::get_device_op () { return o; }

void CAASHip::set_stream (hipStream_t stream) {
  dev::follow_stream(m_->stream, stream);
  m_->stream = stream;
}
void CAASHip::set_Qm_source (const Real* qm_src) {
  // (the fused CAAS + BFB reducer kernel for groups of 2..32 cells uses 16-byte accesses)
  cedr_throw_if(m_->bfb && qm_src && reinterpret_cast<uintptr_t>(qm_src) % 16 != 0 &&
                ! cedr_hip_caas_bfb_fusable(m_->fs.stage, user_reducer_->n_accum_in_place(), m_->fs.max_tile_leaves, 0,
                                            m_->bfb_flags),
                "set_Qm_source: with this reducer and group size the source array must be 16-byte aligned");
  m_->qm_src = qm_src;
}

void CAASHip::bind_arrays (Real* Qm, const Real* Qm_min, const Real* Qm_max, const Real* Qm_prev) {
  cedr_throw_if(Qm && ( ! Qm_min || ! Qm_max), "bind_arrays needs Qm_min and Qm_max with Qm");
  cedr_throw_if(Qm && o.need_conserve_ && ! Qm_prev,
                "bind_arrays needs Qm_prev: a declared tracer conserves");
  cedr_throw_if(Qm && user_reducer_, "bind_arrays is not supported with a UserAllReducer");
  m_->b_qm = Qm; m_->b_min = Qm_min; m_->b_max = Qm_max; m_->b_prev = Qm_prev;
}
const Real* CAASHip::send_data () const { return m_->send; }
const Real* CAASHip::recv_data () const { return m_->recv; }
bool CAASHip::fused_with_bfb_reducer () const { return m_->bfb != nullptr; }

void CAASHip::reduce_locally () {  // :129-201
  const Int nt = get_num_tracers();
  if (m_->bfb) {  // fused with the reducer's first stage: the group sums go straight into its tree
    cedr_k_check(cedr_hip_caas_reduce_local_bfb(m_->probs.data(), nt, o.nlclcells_, o.need_conserve_,
                                                user_reducer_->n_accum_in_place(), o.d_, m_->qm_src,
                                                m_->fs.stage, m_->fs.tile_nleaves, m_->fs.max_tile_leaves,
                                                m_->fs.stride, m_->fs.data, m_->bfb_flags, m_->stream));
    return;
  }
  if (user_reducer_) {
    cedr_k_check(cedr_hip_caas_reduce_local_accum(m_->probs.data(), nt, o.nlclcells_,
                                                  o.need_conserve_,
                                                  user_reducer_->n_accum_in_place(), o.d_,
                                                  m_->qm_src, m_->send, m_->stream));
    return;
  }
  if (m_->b_qm)
    cedr_k_check(cedr_hip_caas_reduce_local_arrays(m_->probs.data(), nt, o.nlclcells_, o.need_conserve_,
                                                   m_->qm_src ? m_->qm_src : m_->b_qm, m_->b_qm,
                                                   m_->b_min, m_->b_max, m_->b_prev,
                                                   m_->partials.data(), m_->stream));
  else
    cedr_k_check(cedr_hip_caas_reduce_local(m_->probs.data(), nt, o.nlclcells_, o.need_conserve_,
                                            o.d_, m_->qm_src, m_->partials.data(), m_->stream));
  cedr_k_check(cedr_hip_caas_sum_partials(nt, o.nlclcells_, m_->partials.data(), m_->send,
                                          m_->stream));
}

void CAASHip::reduce_globally () {  // :203-209
  const int err = p_->comm()->allreduce_sum(m_->send, m_->recv, m_->recv_sz, m_->stream);
  cedr_throw_if(err != 0, "CAAS::reduce_globally all-reduce returned " << err);
}

void CAASHip::finish_locally () {  // :211-253
  if (m_->b_qm)
    cedr_k_check(cedr_hip_caas_finish_local_arrays(get_num_tracers(), o.nlclcells_, m_->b_qm, m_->b_min,
                                                   m_->b_max, m_->recv, m_->stream));
  else
    cedr_k_check(cedr_hip_caas_finish_local(get_num_tracers(), o.nlclcells_, o.d_, m_->recv,
                                            m_->stream));
}

void CAASHip::run () {  // :258-270
  cedr_throw_if( ! finished_setup_, "finish_setup was not called");
  reduce_locally();
  if (m_->bfb) {
    m_->bfb->allreduce_after_first_stage(m_->recv, m_->stream);
  } else if (user_reducer_) {
    const int err = (*user_reducer_)(*p_, m_->send, m_->recv,
                                     o.nlclcells_/user_reducer_->n_accum_in_place(),
                                     static_cast<int>(m_->recv_sz), MPI_SUM);
    cedr_throw_if(err != 0, "UserAllReducer returned " << err);
  } else {
    reduce_globally();
  }
  finish_locally();
}

}  // namespace caas
}  // namespace cedr
