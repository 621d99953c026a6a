// Host side of QLT for MI355X. Setup mirrors reference cedr/cedr_qlt.cpp
// (cited per function); run() drives the tile/stage kernels instead of the
// reference's per-level launch + fence + MPI loop (cedr_qlt.cpp:618-640).
#include "cedr_qlt.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <cstdio>
#include <cstdlib>

#include "../../include/cedr_hip.h"
#include "cedr_device.hpp"
#include "cedr_engine.hpp"
#include "cedr_util.hpp"

namespace cedr {
namespace qlt {

#define cedr_k_check(call) do {                                         \
    const int _e_ = (call);                                             \
    cedr_throw_if(_e_ != 0, "kernel launch failed with HIP error " << _e_ << " in " #call); \
  } while (0)

struct QLTHip::Impl {
  hipStream_t stream = nullptr;
  bool inited = false;
  Real* l2r = nullptr;
  Real* r2l = nullptr;
  dev::Buffer<Real> l2r_own, r2l_own;
  // Tracer tables on the device: caller order (DeviceOp, bulk pack) and bulk order (kernels).
  dev::Buffer<Int> trcr2prob, trcr2bl2r, trcr2br2l, bidx_prob, bidx_l2r, bidx_r2l;
  dev::Buffer<uint8_t> l2r_op;
  cedr_hip_qlt_layout lay;
  // bind_arrays state: row tables on the device, the caller's arrays, tracers still packed
  bool bound = false;
  dev::Buffer<const Real*> in_rows;
  dev::Buffer<Real*> out_rows;
  const Real* b_rhom = nullptr;
  const Real* b_qm = nullptr;
  const Real* b_min = nullptr;
  const Real* b_max = nullptr;
  const Real* b_prev = nullptr;
  Real* b_out = nullptr;
  bool bound_rows = false;  // the kernels read / write the caller's rows directly
  std::vector<Int> packed_tracers;  // caller indices of the t-only tracers
  hipEvent_t down_ev = nullptr;     // see QLTHip::sweep_down_event
  hipEvent_t mid_ev = nullptr;      // see QLTHip::sweep_mid_event
  std::string placement;            // see QLTHip::placement_report
  // CEDR_HIP_LEAF_BOUNDS=1: the cells' q-bound rows of t-only tracers are written as the reference leaves them
  int leaf_flags = [] { const char* v = std::getenv("CEDR_HIP_LEAF_BOUNDS"); return (v && std::atoi(v) != 0) ? CEDR_QLT_LEAF_BOUNDS : 0; }();
  ~Impl () { if (down_ev) (void) hipEventDestroy(down_ev); if (mid_ev) (void) hipEventDestroy(mid_ev); }
};

// ---- MetaData: reference cedr_qlt.cpp:85-157, cedr_qlt_inl.hpp:101-117 ----
int QLTHip::MetaData::get_problem_type (const int& idx) {
  static const Int problem_type[] = { CPT::st, CPT::cst, CPT::t, CPT::ct, CPT::nn, CPT::cnn };
  return problem_type[idx];
}

int QLTHip::MetaData::get_problem_type_idx (const int& mask) {
  switch (mask) {
  case CPT::s:   case CPT::st:  return 0;
  case CPT::cs:  case CPT::cst: return 1;
  case CPT::t:   return 2;
  case CPT::ct:  return 3;
  case CPT::nn:  return 4;
  case CPT::cnn: return 5;
  default: cedr_throw_if(true, "Invalid problem type."); return -1;
  }
}

int QLTHip::MetaData::get_problem_type_l2r_bulk_size (const int& mask) {
  if (mask & ProblemType::nonnegative) return (mask & ProblemType::conserve) ? 2 : 1;
  return (mask & ProblemType::conserve) ? 4 : 3;
}

int QLTHip::MetaData::get_problem_type_r2l_bulk_size (const int& mask) {
  return (mask & ProblemType::shapepreserve) ? 1 : 3;
}

void QLTHip::MetaData::init (const std::vector<int>& t2p) {
  const Int ntracers = static_cast<Int>(t2p.size());
  trcr2prob.assign(t2p.begin(), t2p.end());
  bidx2trcr.assign(ntracers, -1);
  trcr2bidx.assign(ntracers, -1);
  trcr2bl2r.assign(ntracers, -1);
  trcr2br2l.assign(ntracers, -1);
  prob2trcrptr[0] = 0;
  prob2bl2r[0] = 1;  // rhom is field 0
  prob2br2l[0] = 0;
  for (Int pi = 0; pi < nprobtypes; ++pi) {
    const Int pt = get_problem_type(pi);
    const Int l2rsz = get_problem_type_l2r_bulk_size(pt), r2lsz = get_problem_type_r2l_bulk_size(pt);
    Int cnt = 0;
    for (Int ti = 0; ti < ntracers; ++ti) {
      if (trcr2prob[ti] != pt) continue;
      trcr2bl2r[ti] = prob2bl2r[pi] + cnt*l2rsz;
      trcr2br2l[ti] = prob2br2l[pi] + cnt*r2lsz;
      bidx2trcr[prob2trcrptr[pi] + cnt] = ti;
      ++cnt;
    }
    prob2trcrptr[pi+1] = prob2trcrptr[pi] + cnt;
    prob2bl2r[pi+1] = prob2bl2r[pi] + cnt*l2rsz;
    prob2br2l[pi+1] = prob2br2l[pi] + cnt*r2lsz;
  }
  for (Int bi = 0; bi < ntracers; ++bi) trcr2bidx[bidx2trcr[bi]] = bi;
  cedr_assert(prob2trcrptr[nprobtypes] == ntracers);
}

// ---- construction: reference cedr_qlt.cpp:206-236 ----
void QLTHip::init (const Parallel::Ptr& p, const Int& ncells, const tree::FlatTree& tree) {
  p_ = p;
  ns_ = tree::analyze(p->rank(), ncells, tree);
  cedr_throw_if(ns_->levels.empty() || ns_->levels[0].nodes.empty(),
                "QLT does not support 0 cells on a rank.");
  gci2lci_.reserve(ns_->levels[0].nodes.size());
  for (const Int idx : ns_->levels[0].nodes) {
    const auto n = ns_->node_h(idx);
    gci2lci_.add(n->id, n->offset);
  }
  gci2lci_.finish();
  plan::Options popt;
  engine::plan_options(popt);  // the environment's knobs; the fused top tile bounded by the device's LDS
  eng_.reset(new engine::TreeEngine(p, tree, *ns_, popt));
  // How deep the tiles are cut pays or not by the tracer list (end_tracer_declarations), which is not known yet:
  // the tree is kept until then unless the environment fixes the cuts.
  if ( ! std::getenv("CEDR_HIP_MICRO_CUT") && ! std::getenv("CEDR_HIP_MICRO_CUT2") && popt.micro && popt.fused_top)
    tree_keep_.reset(new tree::FlatTree(tree));
  mdb_.reset(new std::vector<int>());
  d_.reset(new Impl());
  o.l2r_data = o.r2l_data = nullptr;
  o.stride = eng_->stride();
  o.trcr2prob = o.trcr2bl2r = o.trcr2br2l = nullptr;
}

QLTHip::QLTHip (const Parallel::Ptr& p, const Int& ncells, const tree::Node::Ptr& tree,
                Options options)
  : CDR(options)
{
  init(p, ncells, tree::flatten(tree));
}

QLTHip::QLTHip (const Parallel::Ptr& p, const Int& ncells, const tree::FlatTree& tree,
                Options options)
  : CDR(options)
{
  init(p, ncells, tree);
}

QLTHip::~QLTHip () {}

void QLTHip::print (std::ostream& os) const { ns_->print(os); }

Int QLTHip::nlclcells () const { return static_cast<Int>(ns_->levels[0].nodes.size()); }

void QLTHip::get_owned_glblcells (std::vector<Long>& gcis) const {
  gcis.resize(ns_->levels[0].nodes.size());
  for (const Int idx : ns_->levels[0].nodes) {
    const auto n = ns_->node_h(idx);
    gcis[n->offset] = n->id;
  }
}

Int QLTHip::gci2lci (const Int& gci) const {
  const int lci = gci2lci_.find(gci);
  cedr_throw_if(lci < 0, "gci " << gci << " not in gci2lci map.");
  return lci;
}

// ---- tracer declaration: reference cedr_qlt.cpp:276-325 ----
void QLTHip::declare_tracer (int problem_type, const Int& rhomidx) {
  cedr_throw_if( ! mdb_, "end_tracer_declarations was already called; "
                 "it is an error to call declare_tracer now.");
  cedr_throw_if(rhomidx > 0, "rhomidx > 0 is not supported yet.");
  problem_type = MetaData::get_problem_type(MetaData::get_problem_type_idx(problem_type));
  mdb_->push_back(problem_type);
}

void QLTHip::end_tracer_declarations () {
  cedr_throw_if( ! mdb_, "end_tracer_declarations was already called.");
  md_.init(*mdb_);
  mdb_.reset();
  // The planner's default cuts (first stage 3 LDS levels, later stages 4) trade one more dependent level per chunk of
  // the first-stage root->leaves sweep -- a cost per tracer -- for smaller, fewer small stages -- a saving per run
  // (profiles/r05_experiments.md 3: QLT::run -1 to -2 % at 40 shape-preserving tracers, but +3 % at 128, and +6 % of the
  // first stage for mixed types, whose general kernel walks a level more slowly). Many tracers, or any tracer that is
  // not shape preserving: round 4's cuts (2 / 3). One more plan build, here, for those lists only.
  if (tree_keep_) {
    bool all_sp = true;
    for (const int pt : md_.trcr2prob) if ( ! (pt & ProblemType::shapepreserve)) all_sp = false;
    if ( ! all_sp || md_.trcr2prob.size() > 64) {
      plan::Options popt;
      engine::plan_options(popt);
      popt.micro_cut = 2;
      popt.micro_cut2 = 3;
      eng_.reset(new engine::TreeEngine(p_, *tree_keep_, *ns_, popt));
      o.stride = eng_->stride();
    }
    tree_keep_.reset();
  }
}

// Sizes in Reals. The reference returns ndps * nslots (cedr_qlt.cpp:293-298);
// this build needs ndps * stride, which is smaller except for degenerate
// trees, so the reference's figure is reported whenever it suffices.
void QLTHip::get_buffers_sizes (size_t& buf1, size_t& buf2) {
  cedr_throw_if(mdb_ != nullptr, "call end_tracer_declarations first");
  const size_t need = static_cast<size_t>(eng_->stride());
  const size_t nslots = std::max(need, static_cast<size_t>(ns_->nslots));
  buf1 = md_.prob2bl2r[MetaData::nprobtypes]*nslots;
  buf2 = md_.prob2br2l[MetaData::nprobtypes]*nslots;
}

void QLTHip::get_buffers_sizes_tight (size_t& buf1, size_t& buf2) {
  cedr_throw_if(mdb_ != nullptr, "call end_tracer_declarations first");
  const size_t stride = static_cast<size_t>(eng_->stride());
  buf1 = md_.prob2bl2r[MetaData::nprobtypes]*stride;
  buf2 = md_.prob2br2l[MetaData::nprobtypes]*stride;
}

void QLTHip::set_buffers (Real* buf1, Real* buf2) {
  d_->l2r = buf1;
  d_->r2l = buf2;
  d_->inited = true;
}

void QLTHip::finish_setup () {
  cedr_throw_if(mdb_ != nullptr, "call end_tracer_declarations first");
  Impl& d = *d_;
  if (o.l2r_data) return;  // already set up
  eng_->upload();
  eng_->reserve_exchange(std::max(md_.prob2bl2r[MetaData::nprobtypes], md_.prob2br2l[MetaData::nprobtypes]));
  {  // tracer tables: caller order for the DeviceOp, bulk order for the kernels
    const Int nt = get_num_tracers();
    d.trcr2prob.upload(md_.trcr2prob);
    d.trcr2bl2r.upload(md_.trcr2bl2r);
    d.trcr2br2l.upload(md_.trcr2br2l);
    std::vector<Int> bp(nt), bl(nt), br(nt);
    for (Int bi = 0; bi < nt; ++bi) {
      const Int ti = md_.bidx2trcr[bi];
      bp[bi] = md_.trcr2prob[ti]; bl[bi] = md_.trcr2bl2r[ti]; br[bi] = md_.trcr2br2l[ti];
    }
    d.bidx_prob.upload(bp);
    d.bidx_l2r.upload(bl);
    d.bidx_r2l.upload(br);
    // Combine rule of every l2r field, cedr_qlt.cpp:375-383: sums, except that a
    // consistent-only tracer carries min(q_min) and max(q_max).
    const Int l2rndps = md_.prob2bl2r[MetaData::nprobtypes];
    std::vector<uint8_t> op(l2rndps, 0);
    for (Int ti = 0; ti < nt; ++ti) {
      const Int pt = md_.trcr2prob[ti];
      if ((pt & ProblemType::consistent) && ! (pt & ProblemType::shapepreserve)) {
        op[md_.trcr2bl2r[ti]] = 1;
        op[md_.trcr2bl2r[ti] + 2] = 2;
      }
    }
    d.l2r_op.upload(op);
    bool sums = true;
    for (const uint8_t o1 : op) if (o1) sums = false;
    eng_->set_all_sums(sums);
    o.trcr2prob = d.trcr2prob.data();
    o.trcr2bl2r = d.trcr2bl2r.data();
    o.trcr2br2l = d.trcr2br2l.data();
  }
  d.lay.nt = get_num_tracers();
  d.lay.l2rndps = md_.prob2bl2r[MetaData::nprobtypes];
  d.lay.r2lndps = md_.prob2br2l[MetaData::nprobtypes];
  d.lay.prefer_mass_con = options_.prefer_numerical_mass_conservation_to_numerical_bounds;
  d.lay.stride = eng_->stride();
  d.lay.bidx_prob = d.bidx_prob.data();
  d.lay.bidx_l2r = d.bidx_l2r.data();
  d.lay.bidx_r2l = d.bidx_r2l.data();
  d.lay.l2r_op = d.l2r_op.data();
  d.lay.flags = 0;
  // tracers per class in bulk order: problem type indices st, cst | t, ct | nn, cnn (cedr_qlt_inl.hpp:103-105)
  d.lay.class_count[0] = md_.prob2trcrptr[2];
  d.lay.class_count[1] = md_.prob2trcrptr[4] - md_.prob2trcrptr[2];
  d.lay.class_count[2] = md_.prob2trcrptr[6] - md_.prob2trcrptr[4];
  {
    bool all_sp = get_num_tracers() > 0;
    for (const Int pt : md_.trcr2prob) if ( ! (pt & ProblemType::shapepreserve)) all_sp = false;
    if (all_sp) d.lay.flags |= CEDR_QLT_ALL_SHAPEPRESERVE;
  }
  // mixed lists: the class-specialised root->leaves forms (cedr_hip_qlt_down_rows), knobs read here once per QLT
  if (const char* e = std::getenv("CEDR_HIP_DOWN_CLS")) if (std::atoi(e) == 0) d.lay.flags |= CEDR_QLT_NO_CLASS_FORMS;
  if (const char* e = std::getenv("CEDR_HIP_DOWN_CLS_MIN_TILES"))
    d.lay.flags |= (std::max(1, std::min(0xffff, std::atoi(e)))) << 8;
  if ( ! d.inited) {
    // Own allocation: only what this build's layout touches (fields x stride). That is about
    // half of the reference's figure, which get_buffers_sizes reports for callers who bring
    // their own memory: internal tree nodes other than tile outputs never reach HBM here.
    const size_t stride = static_cast<size_t>(eng_->stride());
    const size_t nl2r = md_.prob2bl2r[MetaData::nprobtypes]*stride;
    d.r2l_own.alloc_zero(md_.prob2br2l[MetaData::nprobtypes]*stride);
    d.l2r_own.alloc_zero(nl2r);
    // Which device memory an allocation gets decides how fast the first-stage sweeps stream it (the leaves->root
    // kernel 0.385-0.44 ms over the same 1.9 GB at 1.5M cells x 40 tracers, the root->leaves kernel +-3 %, not
    // always the same way; nothing in the virtual address tells; profiles/r03_experiments.md 12). With a first
    // stage big enough for it to matter a few more allocations of the l2r buffer are tried -- the earlier ones
    // kept meanwhile, so that each gets other memory -- and the one over which the two first-stage sweeps together
    // run fastest is kept (on zeros: the sweeps' memory traffic does not depend on the data); then the same for the
    // r2l buffer with the root->leaves sweep.
    // CEDR_HIP_PLACE_TRIES = allocations tried (default 1 = take the first: a library that grabs several transient
    // copies of its largest buffer and times ~30 launches in finish_setup, for an effect whose cause is unknown, should
    // be asked to; bench.py asks for 4 and reports what happened, placement_report()). Callers who bring their own
    // memory are not affected.
    const char* const tv = std::getenv("CEDR_HIP_PLACE_TRIES");
    const int tries = tv ? std::atoi(tv) : 1;
    d.placement = "tries " + std::to_string(tries < 1 ? 1 : tries);
    if (tries > 1) {
      const int32_t nf = md_.prob2bl2r[MetaData::nprobtypes];
      const size_t nr2l = d.r2l_own.size();
      const bool verbose = std::getenv("CEDR_HIP_PLACE_VERBOSE") != nullptr;
      // both first-stage sweeps for an l2r candidate; the root->leaves one alone for an r2l candidate
      const auto score = [&] (Real* l2r, Real* r2l, const bool both) -> float {
        const float up = both ? eng_->time_first_stage(nf, d.l2r_op.data(), l2r, d.stream) : 0.0f;
        if (both && up <= 0) return -1;
        const float down = eng_->time_first_stage_down(&d.lay, l2r, r2l, d.stream);
        return down > 0 ? up + down : -1;
      };
      bool timed = false;  // a timed sweep has written into the buffers: zero them again afterwards, whatever happens
      const auto pick = [&] (dev::Buffer<Real>& own, const size_t n, const bool is_l2r) -> bool {
        timed = true;
        float best = is_l2r ? score(own.data(), d.r2l_own.data(), true) : score(d.l2r_own.data(), own.data(), false);
        if (verbose) std::fprintf(stderr, "cedr: %s placement try 0: %.4f ms\n", is_l2r ? "l2r" : "r2l", best);
        if (best <= 0) { d.placement += is_l2r ? "; l2r: first stage too small to time" : "; r2l: not timed"; return false; }
        char buf[64];
        std::snprintf(buf, sizeof(buf), "; %s ms: %.4f", is_l2r ? "l2r" : "r2l", best);
        d.placement += buf;
        int kept = 0;
        std::vector<std::unique_ptr<dev::Buffer<Real> > > parked;
        for (int t = 1; t < tries; ++t) {
          // Another candidate only while the device keeps twice its size free afterwards: the tries are a transient
          // grab of up to (tries - 1) more copies of the buffer, and a GPU shared with a host model (or its caching
          // allocator) must not be starved by them.
          size_t free_b = 0, total_b = 0;
          if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); break; }
          if (free_b < 3*n*sizeof(Real)) {
            if (verbose) std::fprintf(stderr, "cedr: %s placement: %.1f GB free, no further tries\n",
                                      is_l2r ? "l2r" : "r2l", free_b*1e-9);
            break;
          }
          std::unique_ptr<dev::Buffer<Real> > c(new dev::Buffer<Real>());
          try { c->alloc_zero(n); }
          catch (const std::exception&) { (void) hipGetLastError(); break; }  // out of memory: keep what we have
          const float ms = is_l2r ? score(c->data(), d.r2l_own.data(), true) : score(d.l2r_own.data(), c->data(), false);
          if (verbose) std::fprintf(stderr, "cedr: %s placement try %d: %.4f ms (best so far %.4f)\n",
                                    is_l2r ? "l2r" : "r2l", t, ms, best);
          std::snprintf(buf, sizeof(buf), " %.4f", ms);
          d.placement += buf;
          if (ms > 0 && ms < 0.995f*best) { best = ms; c->swap(own); kept = t; }
          parked.push_back(std::move(c));
        }
        d.placement += " kept " + std::to_string(kept);
        return true;
      };
      // (The sweeps are timed on zeros: every node problem then takes its quick exit, so the root->leaves time is
      // its memory side alone -- which is what an allocation changes; the 0.5 % threshold only avoids swaps for nothing.)
      if (pick(d.l2r_own, nl2r, true)) pick(d.r2l_own, nr2l, false);
      if (timed) {
        cedr_hip_check(hipMemsetAsync(d.l2r_own.data(), 0, nl2r*sizeof(Real), d.stream));
        cedr_hip_check(hipMemsetAsync(d.r2l_own.data(), 0, nr2l*sizeof(Real), d.stream));
        cedr_hip_check(hipStreamSynchronize(d.stream));
      }
    }
    d.l2r = d.l2r_own.data();
    d.r2l = d.r2l_own.data();
    d.inited = true;
  }
  o.l2r_data = d.l2r;
  o.r2l_data = d.r2l;
}

int QLTHip::get_problem_type (const Int& tracer_idx) const {
  cedr_throw_if(tracer_idx < 0 || tracer_idx >= static_cast<Int>(md_.trcr2prob.size()),
                "tracer_idx is out of bounds: " << tracer_idx);
  return md_.trcr2prob[tracer_idx];
}

Int QLTHip::get_num_tracers () const { return static_cast<Int>(md_.trcr2prob.size()); }

const QLTHip::DeviceOp& QLTHip::get_device_op () { return o; }

void QLTHip::set_stream (hipStream_t stream) {
  dev::follow_stream(d_->stream, stream);  // e.g. set_Qm on the old stream, run() on the new one
  d_->stream = stream;
}
const std::string& QLTHip::placement_report () const { return d_->placement; }

hipEvent_t QLTHip::sweep_down_event () {
  if ( ! d_->down_ev) cedr_hip_check(hipEventCreateWithFlags(&d_->down_ev, hipEventDisableTiming));
  return d_->down_ev;
}

hipEvent_t QLTHip::sweep_mid_event () {
  if ( ! d_->mid_ev) cedr_hip_check(hipEventCreateWithFlags(&d_->mid_ev, hipEventDisableTiming));
  eng_->set_after_first_stage_event(d_->mid_ev);
  return d_->mid_ev;
}

void QLTHip::set_rhom_bulk (const Real* rhom) {
  cedr_throw_if( ! o.l2r_data, "finish_setup was not called");
  cedr_k_check(cedr_hip_qlt_set_rhom(o.stride, nlclcells(), rhom, o.l2r_data, d_->stream));
}

void QLTHip::set_Qm_bulk (const Real* Qm, const Real* Qm_min, const Real* Qm_max,
                          const Real* Qm_prev) {
  cedr_throw_if( ! o.l2r_data, "finish_setup was not called");
  cedr_throw_if( ! Qm || ! Qm_min || ! Qm_max, "set_Qm_bulk: null array");
  if ( ! Qm_prev)  // the kernel would dereference it (the inline DeviceOp traps: cedr_qlt_inl.hpp:53-54)
    for (const Int pt : md_.trcr2prob)
      cedr_throw_if(pt & ProblemType::conserve, "set_Qm_bulk: Qm_prev was not provided but a tracer conserves");
  cedr_k_check(cedr_hip_qlt_set_Qm(get_num_tracers(), o.trcr2prob, o.trcr2bl2r, o.stride,
                                   nlclcells(), Qm, Qm_min, Qm_max, Qm_prev, o.l2r_data,
                                   d_->stream));
}

void QLTHip::get_Qm_bulk (Real* Qm_out) const {
  cedr_throw_if( ! o.r2l_data, "finish_setup was not called");
  cedr_k_check(cedr_hip_qlt_get_Qm(get_num_tracers(), o.trcr2br2l, o.stride, nlclcells(),
                                   o.r2l_data, Qm_out, d_->stream));
}

void QLTHip::bind_arrays (const Real* rhom, const Real* Qm, const Real* Qm_min, const Real* Qm_max,
                          const Real* Qm_prev, Real* Qm_out) {
  Impl& d = *d_;
  cedr_throw_if( ! d.inited || ! o.l2r_data, "finish_setup was not called");
  if ( ! rhom) { d.bound = false; return; }
  cedr_throw_if( ! Qm || ! Qm_min || ! Qm_max || ! Qm_out, "bind_arrays: null array");
  d.b_rhom = rhom; d.b_qm = Qm; d.b_min = Qm_min; d.b_max = Qm_max; d.b_prev = Qm_prev;
  d.b_out = Qm_out;
  d.bound = true;
  // Without a micro first stage (a one-cell rank, or micro tiles switched off) run() packs
  // and unpacks through the internal buffers instead: same interface, same results.
  d.bound_rows = eng_->first_stage_takes_rows();
  if ( ! d.bound_rows) return;
  const Int nt = get_num_tracers();
  const size_t n = nlclcells();
  std::vector<const Real*> in(d.lay.l2rndps, nullptr);
  std::vector<Real*> out(nt, nullptr);
  in[0] = rhom;
  d.packed_tracers.clear();
  for (Int ti = 0; ti < nt; ++ti) {
    const Int pt = md_.trcr2prob[ti], bl = md_.trcr2bl2r[ti];
    const Real* const q = Qm + ti*n;
    Int next;
    if (pt & ProblemType::shapepreserve) {
      in[bl] = Qm_min + ti*n; in[bl+1] = q; in[bl+2] = Qm_max + ti*n;
      next = 3;
    } else if (pt & ProblemType::consistent) {
      // bounds are stored divided by rhom (cedr_qlt_inl.hpp:37-44): packed inside run()
      for (Int k = 0; k < 3; ++k) in[bl+k] = o.l2r_data + static_cast<int64_t>(bl+k)*o.stride;
      d.packed_tracers.push_back(ti);
      next = 3;
    } else {
      in[bl] = q;
      next = 1;
    }
    if (pt & ProblemType::conserve) {
      cedr_throw_if( ! Qm_prev, "bind_arrays needs Qm_prev: tracer " << ti << " conserves");
      const bool packed = ! (pt & ProblemType::shapepreserve) && (pt & ProblemType::consistent);
      in[bl+next] = packed ? o.l2r_data + static_cast<int64_t>(bl+next)*o.stride : Qm_prev + ti*n;
    }
    out[md_.trcr2bidx[ti]] = Qm_out + ti*n;
  }
  for (const Real* p : in) cedr_throw_if( ! p, "bind_arrays: an l2r field has no source row");
  d.in_rows.upload(in);
  d.out_rows.upload(out);
}

// Leaves->root over the local stages, one all-gather, the replicated top
// stages, root_compute, then the same stages root->leaves.
void QLTHip::run () {
  Impl& d = *d_;
  cedr_throw_if( ! d.inited || ! o.l2r_data, "finish_setup was not called");
  hipStream_t s = d.stream;
  const bool rows_mode = d.bound && d.bound_rows;
  if (d.bound && ! d.bound_rows) {
    set_rhom_bulk(d.b_rhom);
    set_Qm_bulk(d.b_qm, d.b_min, d.b_max, d.b_prev);
  }
  if (rows_mode && ! d.packed_tracers.empty()) {
    // t-only tracers: their rows stay internal; pack them (they read rhom from field 0)
    const size_t n = nlclcells();
    cedr_k_check(cedr_hip_qlt_set_rhom(o.stride, nlclcells(), d.b_rhom, o.l2r_data, s));
    for (const Int ti : d.packed_tracers)
      cedr_k_check(cedr_hip_qlt_set_Qm(1, o.trcr2prob + ti, o.trcr2bl2r + ti, o.stride, nlclcells(),
                                       d.b_qm + ti*n, d.b_min + ti*n, d.b_max + ti*n,
                                       d.b_prev ? d.b_prev + ti*n : nullptr, o.l2r_data, s));
  }
  // The fused top tile (plan::Stage::top), if the plan has one, is the last local stage; the
  // root->leaves launch over it, cedr_hip_qlt_top, also takes in what precedes it:
  // root_compute on one rank (mode 1), the apex on several (mode 2).
  const bool has_top = eng_->has_top();
  int top_mode = 0;
  const Real* recv = nullptr;
  if (eng_->level_mode()) {
    // Partial tree / scattered ownership: what this rank owns above its pure subtrees goes level by
    // level with one round of messages per level (reference QLT::run, cedr_qlt.cpp:618-640).
    eng_->up(d.lay.l2rndps, d.lay.l2r_op, d.l2r, false, s, rows_mode ? d.in_rows.data() : nullptr);
    if (eng_->owns_root())
      cedr_k_check(cedr_hip_qlt_root(&d.lay, eng_->get_plan().root_slot, d.l2r, d.r2l, s));
    for (int32_t i = eng_->n_level_steps(); i > 0; --i) {
      eng_->level_exchange_down(i-1, d.lay.r2lndps, d.r2l, s);
      cedr_k_check(cedr_hip_qlt_level_down(&d.lay, eng_->level_nodes(i-1), eng_->level_nnodes(i-1), d.l2r, d.r2l, s));
    }
  } else if (eng_->has_apex()) {
    // Ranks own whole subtrees and the tree above them is small: after the local sweeps and the
    // ONE all-gather, one launch does the replicated top (both directions and root_compute) and
    // hands this rank's boundary roots their masses -- the top tile's own launch if there is one.
    recv = eng_->up_to_exchange(d.lay.l2rndps, d.lay.l2r_op, d.l2r, false, s,
                                rows_mode ? d.in_rows.data() : nullptr);
    if (has_top) top_mode = 2;
    else
      cedr_k_check(cedr_hip_qlt_apex(&d.lay, eng_->apex_stage(), recv, eng_->apex_src(), eng_->n_my_roots(),
                                     eng_->apex_my_in(), eng_->my_root_slots_dev(), d.r2l, s));
  } else {
    const auto& top = eng_->top_stages();
    // One rank: the top tile's root->leaves launch re-forms the tile's sums and computes the root
    // itself, so the leaves->root sweep stops below it.
    const bool fused_root = has_top && top.empty();
    eng_->up(d.lay.l2rndps, d.lay.l2r_op, d.l2r, false, s, rows_mode ? d.in_rows.data() : nullptr, fused_root);
    if (fused_root) top_mode = 1;
    else cedr_k_check(cedr_hip_qlt_root(&d.lay, eng_->get_plan().root_slot, d.l2r, d.r2l, s));
    for (size_t i = top.size(); i > 0; --i)
      cedr_k_check(cedr_hip_qlt_down(&d.lay, &top[i-1]->h, d.l2r, d.r2l, s));
    eng_->top_to_local(d.lay.r2lndps, d.r2l, s);
  }
  const auto& loc = eng_->local_stages();
  // sweep_down_event(): recorded exactly once per run(), before the launch over the cells
  if (loc.empty() && d.down_ev) cedr_hip_check(hipEventRecord(d.down_ev, s));
  for (size_t i = loc.size(); i > 0; --i) {
    if (i == 1 && d.down_ev) cedr_hip_check(hipEventRecord(d.down_ev, s));
    if (has_top && i == loc.size()) {
      const bool apx = top_mode == 2;
      cedr_k_check(cedr_hip_qlt_top(&d.lay, eng_->top_stage(), eng_->top_tile(), d.l2r, d.r2l, top_mode,
                                    apx ? eng_->apex_stage() : nullptr, apx ? eng_->apex_tile() : nullptr,
                                    recv, apx ? eng_->apex_src() : nullptr,
                                    apx ? eng_->n_my_roots() : 0, apx ? eng_->apex_my_in() : nullptr,
                                    apx ? eng_->my_root_slots_dev() : nullptr, s));
      continue;
    }
    const bool rows = rows_mode && i == 1;
    cedr_hip_qlt_layout lay = d.lay;
    if (i == 1) lay.flags |= CEDR_QLT_LEAF_STAGE | d.leaf_flags;  // the launch over the cells
    cedr_k_check(cedr_hip_qlt_down_rows(&lay, &loc[i-1]->h, d.l2r, d.r2l,
                                        rows ? d.in_rows.data() : nullptr,
                                        rows ? d.out_rows.data() : nullptr, nlclcells(), s));
  }
  if (d.bound && ! d.bound_rows) get_Qm_bulk(d.b_out);
}

}  // namespace qlt
}  // namespace cedr
