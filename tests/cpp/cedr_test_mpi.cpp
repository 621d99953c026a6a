// The CDRs over a real MPI communicator (mpiexec, N processes sharing GPU 0): the opt-in adapter
// cedr_mpi_adapter.hpp carries QLT's all-gather, CAAS's all-reduce and the BFB reducer's all-gather
// (host staged: the image's MPICH is not GPU aware); QLT also runs from per-rank PARTIAL trees, whose
// tree levels travel as isend/irecv rounds. Every rank also runs the same global problem
// alone (cedr::mpi::make_parallel(), one rank) and compares its own cells: QLT and BFB bit for
// bit, CAAS to 1e-13 per tracer (N partial sums are added in another order than one rank's).
// The reference's counterpart: `mpirun -np 8 cedr_test -t` (cedr/CMakeLists.txt:14-16).
#define CEDR_MPI_HOST_STAGED
#include <mpi.h>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "cedr_mpi_adapter.hpp"
#include "cedr_bfb_tree_allreduce.hpp"
#include "cedr_caas.hpp"
#include "cedr_engine.hpp"
#include "cedr_qlt.hpp"

using cedr::Int;
using cedr::Real;
typedef cedr::ProblemType PT;
#define HIP_OK(x) do { if ((x) != hipSuccess) { std::printf("HIP error at %s:%d\n", __FILE__, __LINE__); MPI_Abort(MPI_COMM_WORLD, 2); } } while (0)

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static Real urand () { g_rng = g_rng*6364136223846793005ull + 1442695040888963407ull; return (Real) (g_rng >> 11)/9007199254740992.0; }

struct Dev {
  Real* p = nullptr;
  explicit Dev (size_t n) { HIP_OK(hipMalloc((void**) &p, (n ? n : 1)*sizeof(Real))); }
  Dev (const std::vector<Real>& h) : Dev(h.size()) { HIP_OK(hipMemcpy(p, h.data(), h.size()*sizeof(Real), hipMemcpyHostToDevice)); }
  ~Dev () { (void) hipFree(p); }
  std::vector<Real> get (size_t n) const { std::vector<Real> h(n); HIP_OK(hipMemcpy(h.data(), p, n*sizeof(Real), hipMemcpyDeviceToHost)); return h; }
};

// rows [tracer][cell] of the cells 'g' of a global [tracer][ncells] array
static std::vector<Real> rows (const std::vector<Real>& glob, int nt, int nc, const std::vector<cedr::Long>& g) {
  std::vector<Real> o((size_t) nt*g.size());
  for (int t = 0; t < nt; ++t) for (size_t i = 0; i < g.size(); ++i) o[t*g.size() + i] = glob[(size_t) t*nc + g[i]];
  return o;
}

// A rank's part of the bisection tree over cells [cs, ce), built directly as tree::Node objects the way a
// host model does for large rank counts (cedr_tree_caller.hpp:12-30): every node gets its rank (its first
// cell's owner) and its level; a node of another rank keeps only the kids that lead to something of this
// rank's or that this rank's nodes need, and nkids = -1 marks "not a cell, nothing kept below".
static cedr::tree::Node::Ptr my_tree_part (const int nc, const int nranks, const int me, const int cs, const int ce,
                                           const cedr::tree::Node* parent) {
  const auto owner = [&] (const int c) { return std::min(nranks - 1, c/(nc/nranks)); };  // oned::Mesh, contiguous
  auto n = std::make_shared<cedr::tree::Node>();
  n->parent = parent;
  n->rank = owner(cs);
  const int cn = ce - cs;
  if (cn == 1) { n->nkids = 0; n->level = 0; n->cellidx = n->rank == me ? cs : -1; return n; }
  const auto k0 = my_tree_part(nc, nranks, me, cs, cs + cn/2, n.get());
  const auto k1 = my_tree_part(nc, nranks, me, cs + cn/2, ce, n.get());
  n->level = 1 + std::max(k0->level, k1->level);
  if (n->rank == me) {  // mine: both kids, whoever owns them
    n->nkids = 2; n->kids[0] = k0; n->kids[1] = k1;
  } else {
    n->nkids = 0;
    for (const auto& k : {k0, k1})
      if (k->rank == me || k->nkids > 0) n->kids[n->nkids++] = k;
    if (n->nkids == 0) n->nkids = -1;
  }
  return n;
}

int main (int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank, size, nerr = 0;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  HIP_OK(hipSetDevice(0));
  try {
    const int nc = 3001;
    const int pts[] = {7, 7, 2, 5, 4, 8, 9};  // cst, cst, s, ct, t, nn, cnn
    const int nt = 7;
    std::vector<Real> rhom(nc), Qm((size_t) nt*nc), lo(Qm), hi(Qm), prev(Qm);
    for (int i = 0; i < nc; ++i) rhom[i] = 0.5*(1 + urand());
    for (int t = 0; t < nt; ++t) {
      for (int i = 0; i < nc; ++i) {
        const size_t k = (size_t) t*nc + i;
        if (pts[t] & PT::nonnegative) { Qm[k] = urand(); lo[k] = 0; hi[k] = 10; prev[k] = Qm[k]; }
        else {
          const Real a = -0.75 + urand(), b = a + urand();
          lo[k] = a*rhom[i]; hi[k] = b*rhom[i]; prev[k] = std::max(lo[k], std::min(hi[k], (a + (b - a)*urand())*rhom[i]));
        }
      }
      for (int i = 0; i < nc; ++i) Qm[(size_t) t*nc + i] = prev[(size_t) t*nc + (i*7919 + 13) % nc];  // permuted
      if (pts[t] & PT::nonnegative) for (int i = 0; i < nc; ++i) prev[(size_t) t*nc + i] = Qm[(size_t) t*nc + i];
    }
    // partial: the rank passes only its part of the tree (tree::Node::level) -> per-level isend/irecv
    // partial 1: prune_for_rank of the full tree; 2: tree::Node objects built part by part (my_tree_part)
    auto run_qlt = [&] (const cedr::mpi::Parallel::Ptr& p, std::vector<cedr::Long>& g, const int partial = 0) {
      cedr::tree::FlatTree ft = cedr::tree::flatten(partial == 2 ? my_tree_part(nc, p->size(), p->rank(), 0, nc, nullptr)
                                                    : cedr::tree::make_tree_over_1d_mesh(p, nc, false));
      if (partial == 1) ft = cedr::tree::prune_for_rank(ft, p->rank());
      cedr::qlt::QLT<> q(p, nc, ft);
      if (partial && p->size() > 1 && ! q.tree_engine().level_mode()) throw std::runtime_error("expected level mode");
      for (int t = 0; t < nt; ++t) q.declare_tracer(pts[t], 0);
      q.end_tracer_declarations();
      q.finish_setup();
      q.get_owned_glblcells(g);
      std::vector<Real> r(g.size());
      for (size_t i = 0; i < g.size(); ++i) r[i] = rhom[g[i]];
      Dev dr(r), dq(rows(Qm, nt, nc, g)), dl(rows(lo, nt, nc, g)), dh(rows(hi, nt, nc, g)), dp(rows(prev, nt, nc, g)),
        out((size_t) nt*g.size());
      q.set_rhom_bulk(dr.p);
      q.set_Qm_bulk(dq.p, dl.p, dh.p, dp.p);
      q.run();
      q.get_Qm_bulk(out.p);
      HIP_OK(hipDeviceSynchronize());
      return out.get((size_t) nt*g.size());
    };
    // --peer-flags: the ranks' communicator is the library's peer-flag transport, set up as INTEGRATION.md shows: the
    // mailbox handles go round with one MPI_Allgather, after which MPI carries nothing of the CDRs' data.
    const bool peer = argc > 1 && ! std::strcmp(argv[1], "--peer-flags");
    cedr::mpi::Parallel::Ptr pm;
    if (peer) {
      auto mb = cedr::mpi::peer_mailbox_create(size);
      char mine[cedr::mpi::kPeerHandleBytes];
      std::vector<char> all((size_t) size*cedr::mpi::kPeerHandleBytes);
      cedr::mpi::peer_mailbox_handle(*mb, mine);
      MPI_Allgather(mine, (int) sizeof(mine), MPI_BYTE, all.data(), (int) sizeof(mine), MPI_BYTE, MPI_COMM_WORLD);
      pm = cedr::mpi::make_parallel_peer_flags(mb, all.data(), rank, size);
      MPI_Barrier(MPI_COMM_WORLD);  // every rank has mapped every mailbox before the first run()
    } else {
      pm = cedr::mpi::make_parallel(MPI_COMM_WORLD);
    }
    const auto ps = cedr::mpi::make_parallel();
    if (pm->size() != size || pm->rank() != rank) ++nerr;
    std::vector<cedr::Long> g1, gm;
    const std::vector<Real> ref = run_qlt(ps, g1), mine = run_qlt(pm, gm);
    std::vector<Real> refg((size_t) nt*nc);
    for (int t = 0; t < nt; ++t) for (size_t i = 0; i < g1.size(); ++i) refg[(size_t) t*nc + g1[i]] = ref[t*g1.size() + i];
    int nq = 0;
    for (int t = 0; t < nt; ++t) for (size_t i = 0; i < gm.size(); ++i)
      if (std::memcmp(&mine[t*gm.size() + i], &refg[(size_t) t*nc + gm[i]], sizeof(Real))) ++nq;
    if (nq) { std::printf("rank %d FAIL QLT over MPI differs from the single-rank run in %d values\n", rank, nq); nerr += nq; }
    for (int form = 1; form <= 2; ++form) {
      std::vector<cedr::Long> gp;
      if (peer && ! pm->comm()->has_exchange()) {  // (more than 32 ranks: no point-to-point rounds; refused at construction)
        bool refused = false;
        try { run_qlt(pm, gp, form); } catch (const std::exception&) { refused = true; }
        if ( ! refused && size > 1) { std::printf("rank %d FAIL a partial tree was not refused over peer flags\n", rank); ++nerr; }
        continue;
      }
      const std::vector<Real> part = run_qlt(pm, gp, form);
      int np = gp.size() == gm.size() ? 0 : 1;
      for (int t = 0; t < nt && ! np; ++t) for (size_t i = 0; i < gp.size(); ++i)
        if (std::memcmp(&part[t*gp.size() + i], &refg[(size_t) t*nc + gp[i]], sizeof(Real))) ++np;
      if (np) { std::printf("rank %d FAIL QLT over partial trees (form %d) differs in %d values\n", rank, form, np); nerr += np; }
    }

    {  // CAAS, contiguous split, shape-preserving tracers only (cedr_caas.cpp:52-53)
      const int ct = 3;  // the first three tracers
      const int b0 = (int) ((long) nc*rank/size), b1 = (int) ((long) nc*(rank + 1)/size);
      auto run_caas = [&] (const cedr::mpi::Parallel::Ptr& p, int c0, int c1) {
        std::vector<cedr::Long> g(c1 - c0);
        for (int i = c0; i < c1; ++i) g[i - c0] = i;
        cedr::caas::CAAS<> c(p, c1 - c0);
        for (int t = 0; t < ct; ++t) c.declare_tracer(pts[t], 0);
        c.end_tracer_declarations();
        c.finish_setup();
        Dev dq(rows(Qm, ct, nc, g)), dl(rows(lo, ct, nc, g)), dh(rows(hi, ct, nc, g)), dp(rows(prev, ct, nc, g));
        c.bind_arrays(dq.p, dl.p, dh.p, dp.p);
        c.run();
        HIP_OK(hipDeviceSynchronize());
        return dq.get((size_t) ct*g.size());
      };
      const std::vector<Real> cref = run_caas(ps, 0, nc), cm = run_caas(pm, b0, b1);
      for (int t = 0; t < ct; ++t) {
        Real dmax = 0, vmax = 0;
        for (int i = b0; i < b1; ++i) {
          dmax = std::max(dmax, std::fabs(cm[(size_t) t*(b1 - b0) + i - b0] - cref[(size_t) t*nc + i]));
          vmax = std::max(vmax, std::fabs(cref[(size_t) t*nc + i]));
        }
        if ( ! (dmax <= 1e-13*vmax)) { std::printf("rank %d FAIL CAAS tracer %d: %.3e\n", rank, t, dmax/vmax); ++nerr; }
      }
    }
    {  // BFB tree all-reduce: every rank gets the bits of the single-rank sum. 3072 leaves: with 2 or 4
       // ranks the contiguous blocks of oned::Mesh (cedr_tree.cpp:366-375) are whole subtrees, so a
       // rank's leaves are its block in order.
      const int nf = 3, nl = 3072;
      std::vector<Real> vals((size_t) nl*nf);
      for (auto& x : vals) x = urand() - 0.5;
      auto run_bfb = [&] (const cedr::mpi::Parallel::Ptr& p) {
        cedr::BfbTreeAllReducer<> b(p, cedr::tree::make_tree_over_1d_mesh(p, nl, false), nl, nf);
        const int first = p->rank()*(nl/p->size());
        std::vector<Real> s(vals.begin() + (size_t) first*nf, vals.begin() + (size_t) (first + b.nlocal())*nf);
        Dev ds(s), dr((size_t) nf);
        b.allreduce(ds.p, dr.p, false);
        HIP_OK(hipDeviceSynchronize());
        return dr.get(nf);
      };
      if (nl % size == 0 && (size & (size - 1)) == 0) {
        const std::vector<Real> r1 = run_bfb(ps), rm = run_bfb(pm);
        if (std::memcmp(r1.data(), rm.data(), nf*sizeof(Real))) { std::printf("rank %d FAIL BFB over MPI differs\n", rank); ++nerr; }
      }
    }
  } catch (const std::exception& e) {
    std::printf("rank %d exception: %s\n", rank, e.what());
    MPI_Abort(MPI_COMM_WORLD, 3);
  }
  int tot = 0;
  MPI_Allreduce(&nerr, &tot, 1, MPI_INT, MPI_SUM, MPI_COMM_WORLD);
  if (rank == 0) std::printf(tot ? "CEDR_MPI_FAIL %d\n" : "CEDR_MPI_OK %d ranks\n", tot ? tot : size);
  MPI_Finalize();
  return tot ? 1 : 0;
}
