// ORACLE -- TEST INFRASTRUCTURE ONLY. Flat C entry points over cedr_oracle.hpp
// so that tests/ and bench.py's cpu_baseline leg can drive the CPU restatement
// through ctypes. Every function returns 0 on success, -1 on a caught
// exception (message via or_last_error), unless it returns a count or handle.
#include "cedr_oracle.hpp"

#include <chrono>
#include <cstring>
#include <exception>
#include <string>

#ifdef _OPENMP
# include <omp.h>
#endif

using namespace oracle;

static thread_local std::string g_err;
#define OR_TRY try {
#define OR_CATCH(ret) } catch (const std::exception& e) { g_err = e.what(); return ret; }

extern "C" {

const char* or_last_error() { return g_err.c_str(); }

// Compiler and flags this library was built with (oracle/Makefile passes them in).
#ifndef OR_BUILD_FLAGS
# define OR_BUILD_FLAGS "unknown"
#endif
const char* or_build_flags() { return OR_BUILD_FLAGS; }

int or_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// ------------------------------------------------------------------ trees
void* or_tree_make_1d(int ncells, int nranks, int pseudorandom, int imbalanced) {
  OR_TRY
  return new CallerTree(make_tree_1d(ncells, nranks, pseudorandom != 0, imbalanced != 0));
  OR_CATCH(nullptr)
}

// Generic binary tree from arrays; node 'root' has no parent. kid0/kid1 = -1 at
// leaves. rank/cellidx matter at leaves; rank < 0 at internal nodes means
// "derive", user_level < 0 means "derive".
void* or_tree_from_arrays(int nnodes, int root, const int* kid0, const int* kid1,
                          const int* rank, const long* cellidx, const int* user_level) {
  OR_TRY
  CallerTree* t = new CallerTree();
  for (int i = 0; i < nnodes; ++i) t->add_node();
  t->root = root;
  for (int i = 0; i < nnodes; ++i) {
    t->rank[i] = rank[i];
    t->cellidx[i] = cellidx[i];
    if (user_level) t->level[i] = user_level[i];
    if (kid0[i] >= 0) {
      t->nkids[i] = kid1[i] >= 0 ? 2 : 1;
      t->kid0[i] = kid0[i]; t->parent[kid0[i]] = i;
      if (kid1[i] >= 0) { t->kid1[i] = kid1[i]; t->parent[kid1[i]] = i; }
    }
  }
  return t;
  OR_CATCH(nullptr)
}

int or_tree_nnodes(void* t) { return ((CallerTree*) t)->size(); }
// Export the caller tree (for handing the same tree to the product library).
int or_tree_export(void* tv, int* root, int* kid0, int* kid1, int* rank, long* cellidx) {
  CallerTree* t = (CallerTree*) tv;
  *root = t->root;
  for (int i = 0; i < t->size(); ++i) {
    kid0[i] = t->nkids[i] > 0 ? t->kid0[i] : -1;
    kid1[i] = t->nkids[i] > 1 ? t->kid1[i] : -1;
    rank[i] = t->rank[i];
    cellidx[i] = t->cellidx[i];
  }
  return 0;
}
void or_tree_free(void* t) { delete (CallerTree*) t; }

// ---------------------------------------------------------------- NodeSets
void* or_analyze(void* tree, int ncells, int my_rank) {
  OR_TRY
  return new NodeSets(analyze(*(CallerTree*) tree, ncells, my_rank));
  OR_CATCH(nullptr)
}
void or_ns_free(void* ns) { delete (NodeSets*) ns; }
int or_ns_nslots(void* ns) { return ((NodeSets*) ns)->nslots; }
int or_ns_nnode(void* ns) { return (int) ((NodeSets*) ns)->nodes.size(); }
int or_ns_nlevels(void* ns) { return (int) ((NodeSets*) ns)->levels.size(); }
int or_ns_level_size(void* ns, int il) { return (int) ((NodeSets*) ns)->levels[il].nodes.size(); }
int or_ns_level_nme(void* ns, int il) { return (int) ((NodeSets*) ns)->levels[il].me.size(); }
int or_ns_level_nkids(void* ns, int il) { return (int) ((NodeSets*) ns)->levels[il].kids.size(); }
// out: nnode x 7 ints {rank,id,parent,nkids,kid0,kid1,offset} (cedr_tree.hpp:22-41)
void or_ns_nodes(void* nsv, int* out) {
  NodeSets* ns = (NodeSets*) nsv;
  for (size_t i = 0; i < ns->nodes.size(); ++i) {
    const NsNode& n = ns->nodes[i];
    int* o = out + 7*i;
    o[0] = n.rank; o[1] = n.id; o[2] = n.parent; o[3] = n.nkids;
    o[4] = n.kids[0]; o[5] = n.kids[1]; o[6] = n.offset;
  }
}
void or_ns_level_nodes(void* nsv, int il, int* out) {
  const auto& v = ((NodeSets*) nsv)->levels[il].nodes;
  std::copy(v.begin(), v.end(), out);
}
// which: 0 = me, 1 = kids; out: n x 3 {rank, offset, size}
void or_ns_level_msgs(void* nsv, int il, int which, int* out) {
  const auto& lvl = ((NodeSets*) nsv)->levels[il];
  const auto& v = which == 0 ? lvl.me : lvl.kids;
  for (size_t i = 0; i < v.size(); ++i) {
    out[3*i] = v[i].rank; out[3*i+1] = v[i].offset; out[3*i+2] = v[i].size;
  }
}
int or_ns_check_comm(void* ns) { return check_comm(*(NodeSets*) ns); }
int or_ns_count_leaves(void* ns, int ncells) { return count_leaves_checked(*(NodeSets*) ns, ncells); }

// All ranks at once: analyze + check_comm + leaf count + test_comm_pattern,
// i.e. unittest_NodeSets (cedr_tree.cpp:448-464). Returns #errors.
int or_tree_unittest(void* tree, int ncells, int nranks, int ntrial) {
  OR_TRY
  std::vector<NodeSets> all;
  int nerr = 0, nleaves = 0;
  for (int r = 0; r < nranks; ++r) {
    all.push_back(analyze(*(CallerTree*) tree, ncells, r));
    nerr += check_comm(all.back());
    const int nl = count_leaves_checked(all.back(), ncells);
    if (nl < 0) ++nerr; else nleaves += nl;
  }
  if (nleaves != ncells) ++nerr;
  for (int t = 0; t < ntrial; ++t) nerr += test_comm_pattern(all, ncells);
  return nerr;
  OR_CATCH(-1)
}

// ------------------------------------------------------------------- local
int or_solve_1eq_bc_qp_2d(const double* w, const double* a, double b, const double* xlo,
                          const double* xhi, const double* y, double* x, int clip, int early) {
  return solve_1eq_bc_qp_2d(w, a, b, xlo, xhi, y, x, clip != 0, early != 0);
}
int or_solve_1eq_bc_qp(int n, const double* w, const double* a, double b, const double* xlo,
                       const double* xhi, const double* y, double* x, int max_its) {
  return solve_1eq_bc_qp(n, w, a, b, xlo, xhi, y, x, max_its);
}
void or_caas_local(int n, const double* a, double b, const double* xlo, const double* xhi,
                   const double* y, double* x, int clip) {
  caas_local(n, a, b, xlo, xhi, y, x, clip != 0);
}
int or_solve_1eq_nonneg(int n, const double* a, double b, const double* y, double* x,
                        const double* w, int method) {
  OR_TRY
  return solve_1eq_nonneg(n, a, b, y, x, w, method);
  OR_CATCH(-99)
}
// io: pd[3..4], k0d, k1d as in cedr_qlt_inl.hpp:175-180; out[2] = kids' Qm
void or_solve_node_problem(int pt, double rhom, const double* pd, double Qm,
                           double rhom0, const double* k0d, double rhom1, const double* k1d,
                           int prefer, double* out) {
  solve_node_problem(pt, rhom, pd, Qm, rhom0, k0d, out, rhom1, k1d, out + 1, prefer != 0);
}

// --------------------------------------------------------------------- QLT
int or_canonical_problem_type(int mask) { return canonical_problem_type(mask); }

void* or_qlt_create(void* tree, int ncells, int nranks, int prefer_mass_con) {
  OR_TRY
  return new Qlt(*(CallerTree*) tree, ncells, nranks, prefer_mass_con != 0);
  OR_CATCH(nullptr)
}
void or_qlt_free(void* q) { delete (Qlt*) q; }
int or_qlt_declare(void* q, const int* types, int nt) {
  OR_TRY
  ((Qlt*) q)->declare_tracers(std::vector<int>(types, types + nt));
  return 0;
  OR_CATCH(-1)
}
void or_qlt_set_threads(void* q, int n) { ((Qlt*) q)->nthreads = n; }
int or_qlt_nlclcells(void* q, int rank) { return ((Qlt*) q)->nlclcells(rank); }
int or_qlt_nslots(void* q, int rank) { return ((Qlt*) q)->r[rank].ns.nslots; }
int or_qlt_l2rndps(void* q) { return ((Qlt*) q)->md.l2rndps(); }
int or_qlt_r2lndps(void* q) { return ((Qlt*) q)->md.r2lndps(); }
void* or_qlt_nodesets(void* q, int rank) { return &((Qlt*) q)->r[rank].ns; }  // borrowed
void or_qlt_owned_glblcells(void* q, int rank, long* out) {
  std::vector<long> g;
  ((Qlt*) q)->owned_glblcells(rank, g);
  std::copy(g.begin(), g.end(), out);
}
// out arrays have nt entries each, then the three 7-entry tables.
void or_qlt_meta(void* qv, int* trcr2prob, int* bidx2trcr, int* trcr2bidx, int* trcr2bl2r,
                 int* trcr2br2l, int* prob2trcrptr, int* prob2bl2r, int* prob2br2l) {
  const QltMeta& m = ((Qlt*) qv)->md;
  for (int i = 0; i < m.nt; ++i) {
    trcr2prob[i] = m.trcr2prob[i]; bidx2trcr[i] = m.bidx2trcr[i]; trcr2bidx[i] = m.trcr2bidx[i];
    trcr2bl2r[i] = m.trcr2bl2r[i]; trcr2br2l[i] = m.trcr2br2l[i];
  }
  for (int i = 0; i < 7; ++i) {
    prob2trcrptr[i] = m.prob2trcrptr[i]; prob2bl2r[i] = m.prob2bl2r[i]; prob2br2l[i] = m.prob2br2l[i];
  }
}
// Bulk pack for one rank, arrays laid out [tracer][local cell] (rhom: [cell]).
// Mirrors the set_rhom / set_Qm kernels of cedr_test_randomized_inl.hpp:32-48.
int or_qlt_set(void* qv, int rank, const double* rhom, const double* Qm, const double* Qm_min,
               const double* Qm_max, const double* Qm_prev) {
  OR_TRY
  Qlt* q = (Qlt*) qv;
  const int n = q->nlclcells(rank), nt = q->md.nt;
  if (rhom) for (int i = 0; i < n; ++i) q->set_rhom(rank, i, rhom[i]);
  for (int t = 0; t < nt; ++t)
    for (int i = 0; i < n; ++i) {
      const size_t k = (size_t) t*n + i;
      q->set_Qm(rank, i, t, Qm[k], Qm_min[k], Qm_max[k], Qm_prev[k]);
    }
  return 0;
  OR_CATCH(-1)
}
int or_qlt_run(void* q) {
  OR_TRY
  ((Qlt*) q)->run();
  return 0;
  OR_CATCH(-1)
}
void or_qlt_get(void* qv, int rank, double* Qm_out) {
  Qlt* q = (Qlt*) qv;
  const int n = q->nlclcells(rank), nt = q->md.nt;
  for (int t = 0; t < nt; ++t)
    for (int i = 0; i < n; ++i) Qm_out[(size_t) t*n + i] = q->get_Qm(rank, i, t);
}
// Seconds for nrep calls of run() (cst tracers do not modify l2r, so repeating
// without re-packing is the same work every time).
double or_qlt_time_run(void* q, int nrep) {
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < nrep; ++i) ((Qlt*) q)->run();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// -------------------------------------------------------------------- CAAS
void* or_caas_create(const int* nlcl, int nranks, const int* types, int nt, int accum) {
  OR_TRY
  return new Caas(std::vector<int>(nlcl, nlcl + nranks), std::vector<int>(types, types + nt), accum);
  OR_CATCH(nullptr)
}
void or_caas_free(void* c) { delete (Caas*) c; }
void or_caas_set_threads(void* c, int n) { ((Caas*) c)->nthreads = n; }
int or_caas_set(void* cv, int rank, const double* rhom, const double* Qm, const double* Qm_min,
                const double* Qm_max, const double* Qm_prev) {
  Caas* c = (Caas*) cv;
  const int n = c->nlcl[rank];
  if (rhom) for (int i = 0; i < n; ++i) c->set_rhom(rank, i, rhom[i]);
  for (int t = 0; t < c->nt; ++t)
    for (int i = 0; i < n; ++i) {
      const size_t k = (size_t) t*n + i;
      c->set_Qm(rank, i, t, Qm[k], Qm_min[k], Qm_max[k], Qm_prev ? Qm_prev[k] : 0.0);
    }
  return 0;
}
// Re-pack only Qm (what a timing loop must restore between runs).
void or_caas_set_Qm_only(void* cv, int rank, const double* Qm) {
  Caas* c = (Caas*) cv;
  const size_t n = c->nlcl[rank];
  std::memcpy(&c->d[rank][n], Qm, sizeof(double)*n*c->nt);
}
int or_caas_run(void* c) {
  OR_TRY
  ((Caas*) c)->run();
  return 0;
  OR_CATCH(-1)
}
void or_caas_get(void* cv, int rank, double* Qm_out) {
  Caas* c = (Caas*) cv;
  const size_t n = c->nlcl[rank];
  std::memcpy(Qm_out, &c->d[rank][n], sizeof(double)*n*c->nt);
}
void or_caas_recv(void* cv, double* out) {
  Caas* c = (Caas*) cv;
  std::copy(c->recv.begin(), c->recv.end(), out);
}
// Seconds spent inside run() over nrep repetitions; Qm is restored from 'Qm'
// before each run outside the timed region (rank 0 only; single-rank use).
double or_caas_time_run(void* cv, const double* Qm, int nrep) {
  Caas* c = (Caas*) cv;
  double tot = 0;
  for (int i = 0; i < nrep; ++i) {
    or_caas_set_Qm_only(cv, 0, Qm);
    const auto t0 = std::chrono::steady_clock::now();
    c->run();
    tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return tot;
}

// --------------------------------------------------------------------- BFB
// send: concatenation over ranks of nlocal[r]*nfield; recv: nranks*nfield.
int or_bfb_allreduce(void* tree, int ncells, int nranks, int nfield, int transpose,
                     const double* send, double* recv) {
  OR_TRY
  std::vector<NodeSets> all;
  std::vector<std::vector<double> > s(nranks), rv;
  size_t pos = 0;
  for (int r = 0; r < nranks; ++r) {
    all.push_back(analyze(*(CallerTree*) tree, ncells, r));
    const size_t n = all.back().levels[0].nodes.size()*nfield;
    s[r].assign(send + pos, send + pos + n);
    pos += n;
  }
  bfb_tree_allreduce(all, nfield, transpose != 0, s, rv);
  for (int r = 0; r < nranks; ++r) std::copy(rv[r].begin(), rv[r].end(), recv + (size_t) r*nfield);
  return 0;
  OR_CATCH(-1)
}

}  // extern "C"
