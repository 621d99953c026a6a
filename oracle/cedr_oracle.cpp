// ORACLE -- TEST INFRASTRUCTURE ONLY (see cedr_oracle.hpp for the rules and the
// pinning status). Plain C++ restatement; compiled with -ffp-contract=off so
// that no a*b+c is fused (the reference's x86-64 build does not fuse either).
#include "cedr_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>

namespace oracle {

static void fail(const std::string& m) { throw std::logic_error("oracle: " + m); }

// cedr_kokkos.hpp:136-139 -- comparisons, not std::fmin/fmax.
static inline double rmin(double a, double b) { return a < b ? a : b; }
static inline double rmax(double a, double b) { return a > b ? a : b; }
static const double EPS = std::numeric_limits<double>::epsilon();

// ===================================================================== tree
int CallerTree::add_node() {
  parent.push_back(-1); rank.push_back(-1); nkids.push_back(0);
  kid0.push_back(-1); kid1.push_back(-1); reserved.push_back(-1);
  level.push_back(-1); cellidx.push_back(-1);
  return size() - 1;
}

int mesh_rank(int nc, int nranks, bool pseudorandom, int ci) {
  if (!pseudorandom) return std::min(nranks - 1, ci / (nc / nranks));
  const int chunk = ci / nranks;
  return (ci + chunk) % nranks;
}

static int build_1d(CallerTree& t, int nc, int nranks, bool prand, bool imb,
                    int cs, int ce, int parent) {
  const int cn = ce - cs;
  const int cn0 = (imb && cn > 2) ? cn/3 : cn/2;
  const int me = t.add_node();
  t.parent[me] = parent;
  if (cn == 1) {
    t.rank[me] = mesh_rank(nc, nranks, prand, cs);
    t.cellidx[me] = cs;
    return me;
  }
  t.nkids[me] = 2;
  const int a = build_1d(t, nc, nranks, prand, imb, cs, cs + cn0, me);
  t.kid0[me] = a;
  const int b = build_1d(t, nc, nranks, prand, imb, cs + cn0, ce, me);
  t.kid1[me] = b;
  return me;
}

CallerTree make_tree_1d(int ncells, int nranks, bool pseudorandom, bool imbalanced) {
  if (nranks > ncells) fail("#GIDs < #ranks is not supported.");  // cedr_tree.cpp:363
  CallerTree t;
  t.root = build_1d(t, ncells, nranks, pseudorandom, imbalanced, 0, ncells, -1);
  return t;
}

static int kid_of(const CallerTree& t, int n, int i) { return i == 0 ? t.kid0[n] : t.kid1[n]; }

// cedr_tree.cpp:37-53
static int init_tree(CallerTree& t, int my_rank, int n, int& id) {
  t.reserved[n] = -1;
  int depth = 0;
  for (int i = 0; i < t.nkids[n]; ++i)
    depth = std::max(depth, init_tree(t, my_rank, kid_of(t, n, i), id));
  if (t.nkids[n]) {
    if (t.rank[n] < 0) t.rank[n] = t.rank[t.kid0[n]];
    t.cellidx[n] = id++;
  } else if (t.rank[n] == my_rank && (t.cellidx[n] < 0 || t.cellidx[n] >= id)) {
    fail("cellidx out of range");
  }
  return depth + 1;
}

// cedr_tree.cpp:55-129
static void schedule(CallerTree& t, NodeSets& ns, int my_rank, int n, int& level,
                     bool& need_parent) {
  level = -1;
  bool make = false;
  for (int i = 0; i < t.nkids[n]; ++i) {
    int kl; bool kn;
    schedule(t, ns, my_rank, kid_of(t, n, i), kl, kn);
    level = std::max(level, kl);
    if (kn) make = true;
  }
  ++level;
  if (t.level[n] >= 0) level = t.level[n];
  const bool owned = t.rank[n] == my_rank;
  need_parent = owned;
  if (!(owned || make)) return;
  const int idx = (int) ns.nodes.size();
  ns.nodes.push_back(NsNode());
  if (owned) ns.levels[level].nodes.push_back(idx);
  t.reserved[n] = idx;
  ns.nodes[idx].rank = t.rank[n];
  if (t.nkids[n] == 0) ns.nodes[idx].id = (int) t.cellidx[n];
  if (owned) {
    ns.nodes[idx].nkids = t.nkids[n];
    for (int i = 0; i < t.nkids[n]; ++i) {
      const int kid = kid_of(t, n, i);
      if (t.reserved[kid] == -1) {
        const int kidx = (int) ns.nodes.size();
        ns.nodes.push_back(NsNode());
        t.reserved[kid] = kidx;
        ns.nodes[idx].kids[i] = kidx;
        ns.nodes[kidx].rank = t.rank[kid];
        ns.nodes[kidx].id = (int) t.cellidx[kid];
        ns.nodes[kidx].nkids = 0;
      } else {
        ns.nodes[idx].kids[i] = t.reserved[kid];
        ns.nodes[t.reserved[kid]].parent = idx;
      }
    }
  } else {
    ns.nodes[idx].nkids = 0;
    for (int i = 0; i < t.nkids[n]; ++i) {
      const int kid = kid_of(t, n, i);
      if (t.reserved[kid] >= 0 && t.rank[kid] == my_rank) {
        ns.nodes[idx].kids[ns.nodes[idx].nkids++] = t.reserved[kid];
        ns.nodes[t.reserved[kid]].parent = idx;
      }
    }
  }
}

// cedr_tree.cpp:148-180. The reference stable-sorts pair<rank, Node*>; the
// default pair ordering compares the pointer second, i.e. the node's index in
// node_mem_, so sorting (rank, node index) pairs reproduces it.
static void init_offsets(int my_rank, NodeSets& ns, std::vector<std::pair<int,int> >& rns,
                         std::vector<Msg>& mmds, int& offset) {
  for (auto& rn : rns) if (rn.first == my_rank) rn.first = -1;
  std::stable_sort(rns.begin(), rns.end());
  int prev = -1;
  for (auto& rn : rns) {
    NsNode& nd = ns.nodes[rn.second];
    if (rn.first == -1) {
      if (nd.offset == -1) nd.offset = offset++;
      continue;
    }
    if (rn.first != prev) {
      prev = rn.first;
      Msg m; m.rank = rn.first; m.offset = offset; m.size = 0;
      mmds.push_back(m);
    }
    ++mmds.back().size;
    nd.offset = offset++;
  }
}

// cedr_tree.cpp:185-213
static void init_comm(int my_rank, NodeSets& ns) {
  ns.nslots = 0;
  for (auto& lvl : ns.levels) {
    std::vector<std::pair<int,int> > me, kids;
    for (int idx : lvl.nodes) {
      const NsNode& n = ns.nodes[idx];
      me.push_back(std::make_pair(n.parent >= 0 ? ns.nodes[n.parent].rank : my_rank, idx));
      for (int k = 0; k < n.nkids; ++k)
        kids.push_back(std::make_pair(ns.nodes[n.kids[k]].rank, n.kids[k]));
    }
    init_offsets(my_rank, ns, me, lvl.me, ns.nslots);
    init_offsets(my_rank, ns, kids, lvl.kids, ns.nslots);
  }
}

NodeSets analyze(CallerTree t, int ncells, int my_rank) {
  NodeSets ns;
  int id = ncells;
  int depth = init_tree(t, my_rank, t.root, id);
  if (t.level[t.root] >= 0) depth = t.level[t.root] + 1;
  ns.levels.resize(depth);
  int lv; bool np;
  schedule(t, ns, my_rank, t.root, lv, np);
  // consolidate, cedr_tree.cpp:138-144
  std::vector<NsLevel> keep;
  for (auto& l : ns.levels) if (!l.nodes.empty()) keep.push_back(l);
  ns.levels.swap(keep);
  init_comm(my_rank, ns);
  return ns;
}

int check_comm(const NodeSets& ns) {
  int nerr = 0;
  std::vector<int> cnt(ns.nslots, 0);
  for (const auto& lvl : ns.levels)
    for (int idx : lvl.nodes) {
      const NsNode& n = ns.nodes[idx];
      if (n.offset < 0 || n.offset >= ns.nslots) { ++nerr; continue; }
      ++cnt[n.offset];
      for (int i = 0; i < n.nkids; ++i) {
        const NsNode& k = ns.nodes[n.kids[i]];
        if (k.rank != n.rank) ++cnt[k.offset];
      }
    }
  for (int c : cnt) if (c != 1) ++nerr;
  return nerr;
}

int count_leaves_checked(const NodeSets& ns, int ncells) {
  if (ns.levels.empty() || ns.levels[0].nodes.empty()) return -1;
  const int nl = (int) ns.levels[0].nodes.size();
  for (int idx : ns.levels[0].nodes) {
    const NsNode& n = ns.nodes[idx];
    if (n.nkids) return -1;
    if (n.offset >= nl || n.id >= ncells) return -1;
  }
  return nl;
}

// ================================================================ simulated net
void SimNet::send(int src, int dst, const double* p, size_t n) {
  q[std::make_pair(src, dst)].push_back(std::vector<double>(p, p + n));
}
bool SimNet::can_recv(int src, int dst) const {
  auto it = q.find(std::make_pair(src, dst));
  return it != q.end() && !it->second.empty();
}
void SimNet::recv(int src, int dst, double* p, size_t n) {
  auto& dq = q[std::make_pair(src, dst)];
  if (dq.empty()) fail("recv on empty queue");
  if (dq.front().size() != n) fail("message size mismatch");
  std::copy(dq.front().begin(), dq.front().end(), p);
  dq.pop_front();
}

// Drives the per-level communication pattern shared by QLT::run
// (cedr_qlt.cpp:618-640), BfbTreeAllReducer::allreduce
// (cedr_bfb_tree_allreduce.cpp:104-156) and test_comm_pattern
// (cedr_tree.cpp:279-348) over all simulated ranks. A rank advances until it
// reaches a receive whose message has not been sent yet.
struct SweepOps {
  virtual ~SweepOps() {}
  virtual double* up_buf(int rank) = 0;     // buffer l2r messages are cut from
  virtual double* down_buf(int rank) = 0;   // buffer r2l messages are cut from
  virtual int up_ndps() const = 0;
  virtual int down_ndps() const = 0;
  virtual void up(int rank, int il) = 0;
  virtual void root(int rank) = 0;
  virtual void down(int rank, int il) = 0;
};

static void run_sweep(const std::vector<const NodeSets*>& all, SweepOps& ops) {
  const int P = (int) all.size();
  SimNet net;
  std::vector<int> pc(P, 0);
  bool progress = true;
  int done = 0;
  while (done < P) {
    if (!progress) fail("simulated communication deadlocked");
    progress = false;
    done = 0;
    for (int r = 0; r < P; ++r) {
      const NodeSets& ns = *all[r];
      const int L = (int) ns.levels.size();
      const int nstep = 6*L + 1;
      for (;;) {
        if (pc[r] >= nstep) break;
        const int s = pc[r];
        if (s < 3*L) {
          const int il = s / 3, kind = s % 3;
          const NsLevel& lvl = ns.levels[il];
          if (kind == 0) {
            bool ok = true;
            for (const Msg& m : lvl.kids) if (!net.can_recv(m.rank, r)) ok = false;
            if (!ok) break;
            for (const Msg& m : lvl.kids)
              net.recv(m.rank, r, ops.up_buf(r) + (size_t) m.offset*ops.up_ndps(),
                       (size_t) m.size*ops.up_ndps());
          } else if (kind == 1) {
            ops.up(r, il);
          } else {
            for (const Msg& m : lvl.me)
              net.send(r, m.rank, ops.up_buf(r) + (size_t) m.offset*ops.up_ndps(),
                       (size_t) m.size*ops.up_ndps());
          }
        } else if (s == 3*L) {
          ops.root(r);
        } else {
          const int t = s - 3*L - 1;
          const int il = L - 1 - t / 3, kind = t % 3;
          const NsLevel& lvl = ns.levels[il];
          if (kind == 0) {
            bool ok = true;
            for (const Msg& m : lvl.me) if (!net.can_recv(m.rank, r)) ok = false;
            if (!ok) break;
            for (const Msg& m : lvl.me)
              net.recv(m.rank, r, ops.down_buf(r) + (size_t) m.offset*ops.down_ndps(),
                       (size_t) m.size*ops.down_ndps());
          } else if (kind == 1) {
            ops.down(r, il);
          } else {
            for (const Msg& m : lvl.kids)
              net.send(r, m.rank, ops.down_buf(r) + (size_t) m.offset*ops.down_ndps(),
                       (size_t) m.size*ops.down_ndps());
          }
        }
        ++pc[r];
        progress = true;
      }
      if (pc[r] >= nstep) ++done;
    }
  }
}

// cedr_tree.cpp:279-348 (the data type there is Int; doubles hold the same
// integers exactly for every size the tests use).
struct CommPatternOps : SweepOps {
  const std::vector<NodeSets>& all;
  std::vector<std::vector<double> > data;
  CommPatternOps(const std::vector<NodeSets>& a) : all(a) {
    for (const auto& ns : all) {
      data.push_back(std::vector<double>(ns.nslots, 0));
      for (int idx : ns.levels[0].nodes)
        data.back()[ns.nodes[idx].offset] = ns.nodes[idx].id;
    }
  }
  double* up_buf(int r) override { return data[r].data(); }
  double* down_buf(int r) override { return data[r].data(); }
  int up_ndps() const override { return 1; }
  int down_ndps() const override { return 1; }
  void up(int r, int il) override {
    const NodeSets& ns = all[r];
    for (int idx : ns.levels[il].nodes) {
      const NsNode& n = ns.nodes[idx];
      if (!n.nkids) continue;
      data[r][n.offset] = 0;
      for (int i = 0; i < n.nkids; ++i) data[r][n.offset] += data[r][ns.nodes[n.kids[i]].offset];
    }
  }
  void root(int) override {}
  void down(int r, int il) override {
    const NodeSets& ns = all[r];
    for (int idx : ns.levels[il].nodes) {
      const NsNode& n = ns.nodes[idx];
      if (!n.nkids) continue;
      for (int i = 0; i < n.nkids; ++i) data[r][ns.nodes[n.kids[i]].offset] = data[r][n.offset];
    }
  }
};

int test_comm_pattern(const std::vector<NodeSets>& all, int ncells) {
  CommPatternOps ops(all);
  std::vector<const NodeSets*> p;
  for (const auto& ns : all) p.push_back(&ns);
  run_sweep(p, ops);
  int nerr = 0;
  const double want = ((double) ncells*(ncells - 1)) / 2;
  for (size_t r = 0; r < all.size(); ++r)
    for (int idx : all[r].levels[0].nodes)
      if (ops.data[r][all[r].nodes[idx].offset] != want) ++nerr;
  return nerr;
}

// ===================================================================== local
// cedr_local_inl.hpp:14-18
static double calc_r_tol(double b, const double* a, const double* y, int n) {
  double ab = std::abs(b);
  for (int i = 0; i < n; ++i) ab = rmax(ab, std::abs(a[i]*y[i]));
  return 1e1*EPS*std::abs(ab);
}

// cedr_local_inl.hpp:23-41
static int check_lu(int n, const double* a, double b, const double* xlo, const double* xhi,
                    double r_tol, double* x) {
  double r = -b;
  for (int i = 0; i < n; ++i) { x[i] = xlo[i]; r += a[i]*x[i]; }
  if (std::abs(r) <= r_tol) return 1;
  if (r > 0) return -1;
  r = -b;
  for (int i = 0; i < n; ++i) { x[i] = xhi[i]; r += a[i]*x[i]; }
  if (std::abs(r) <= r_tol) return 1;
  if (r < 0) return -1;
  return 0;
}

// cedr_local_inl.hpp:69-165
int solve_1eq_bc_qp_2d(const double* w, const double* a, double b, const double* xlo,
                       const double* xhi, const double* y, double* x,
                       bool clip, bool early_exit_on_tol) {
  if (early_exit_on_tol) {
    const double r_tol = calc_r_tol(b, a, y, 2);
    if (check_lu(2, a, b, xlo, xhi, r_tol, x) == -1) return -1;
  }
  {
    double qmass = 0, dm = b;
    for (int i = 0; i < 2; ++i) {
      const double qi = a[i]/w[i];
      qmass += a[i]*qi;
      dm -= a[i]*y[i];
    }
    const double lambda = dm/qmass;
    bool inside = true;
    for (int i = 0; i < 2; ++i) {
      x[i] = y[i] + lambda*(a[i]/w[i]);
      if (x[i] < xlo[i] || x[i] > xhi[i]) { inside = false; break; }
    }
    if (inside) return 1;
  }
  double base[2];
  for (int i = 0; i < 2; ++i) base[i] = 0.5*b/a[i];
  const double dir[2] = {-a[1], a[0]};
  double al[4];
  al[0] = (xlo[1] - base[1])/dir[1];
  al[1] = (xhi[0] - base[0])/dir[0];
  al[2] = (xhi[1] - base[1])/dir[1];
  al[3] = (xlo[0] - base[0])/dir[0];
  double mn = al[0], mx = al[0];
  int imin = 0, imax = 0;
  for (int i = 1; i < 4; ++i) {
    if (al[i] < mn) { mn = al[i]; imin = i; }
    if (al[i] > mx) { mx = al[i]; imax = i; }
  }
  int mid[2] = {0, 0}, cnt = 0;
  for (int i = 0; i < 4 && cnt < 2; ++i)
    if (i != imin && i != imax) mid[cnt++] = i;
  double obj[2];
  for (int j = 0; j < 2; ++j) {
    const double alpha = al[mid[j]];
    double o = 0;
    for (int i = 0; i < 2; ++i) {
      x[i] = base[i] + alpha*dir[i];
      const double d = y[i] - x[i];
      o += w[i]*(d*d);
    }
    obj[j] = o;
  }
  const int ai = mid[obj[0] <= obj[1] ? 0 : 1];
  int i0;
  if (ai == 0 || ai == 2) { x[1] = ai == 0 ? xlo[1] : xhi[1]; i0 = 1; }
  else                    { x[0] = ai == 1 ? xhi[0] : xlo[0]; i0 = 0; }
  const int i1 = (i0 + 1) % 2;
  x[i1] = (b - a[i0]*x[i0])/a[i1];
  if (clip) x[i1] = rmin(xhi[i1], rmax(xlo[i1], x[i1]));
  return 1;
}

// cedr_local_inl.hpp:44-64
static void calc_r(int n, const double* w, const double* a, double b, const double* xlo,
                   const double* xhi, const double* y, double lambda, double* x,
                   double& r, double& r_lambda) {
  r = 0; r_lambda = 0;
  for (int i = 0; i < n; ++i) {
    const double q = a[i]/w[i];
    const double xt = y[i] + lambda*q;
    if (xt < xlo[i]) x[i] = xlo[i];
    else if (xt > xhi[i]) x[i] = xhi[i];
    else { x[i] = xt; r_lambda += a[i]*q; }
    r += a[i]*x[i];
  }
  r -= b;
}

// cedr_local_inl.hpp:168-270
int solve_1eq_bc_qp(int n, const double* w, const double* a, double b, const double* xlo,
                    const double* xhi, const double* y, double* x, int max_its) {
  const double r_tol = calc_r_tol(b, a, y, n);
  int info = check_lu(n, a, b, xlo, xhi, r_tol, x);
  if (info != 0) return info;
  for (int i = 0; i < n; ++i)
    if (x[i] != y[i]) { info = 1; x[i] = y[i]; }
  const double wall_dist = 1e-3;
  double lamlo = 0, lamhi = 0;
  for (int i = 0; i < n; ++i) {
    const double rq = w[i]/a[i];
    const double lo_i = rq*(xlo[i] - y[i]), hi_i = rq*(xhi[i] - y[i]);
    if (i == 0) { lamlo = lo_i; lamhi = hi_i; }
    else { lamlo = rmin(lamlo, lo_i); lamhi = rmax(lamhi, hi_i); }
  }
  const double lamlo_feas = lamlo, lamhi_feas = lamhi;
  double lambda = (lamlo <= 0 && lamhi >= 0) ? 0 : lamlo;
  bool prev_bisect = false;
  int nbisect = 0;
  info = -2;
  for (int it = 0; it < max_its; ++it) {
    double r, r_lambda;
    calc_r(n, w, a, b, xlo, xhi, y, lambda, x, r, r_lambda);
    if (std::abs(r) <= r_tol) { info = 1; break; }
    if (nbisect > 64) {
      if (lamhi == lamhi_feas || lamlo == lamlo_feas) { info = -1; break; }
      info = 1; break;
    }
    if (r > 0) lamhi = lambda; else lamlo = lambda;
    if (r_lambda != 0) lambda -= r/r_lambda; else lambda = lamlo;
    const double D = prev_bisect ? 0 : wall_dist*(lamhi - lamlo);
    if (lambda - lamlo < D || lamhi - lambda < D) {
      lambda = 0.5*(lamlo + lamhi);
      ++nbisect;
      prev_bisect = true;
    } else {
      prev_bisect = false;
    }
  }
  return info;
}

// cedr_local_inl.hpp:273-306
void caas_local(int n, const double* a, double b, const double* xlo, const double* xhi,
                const double* y, double* x, bool clip) {
  double dm = b;
  for (int i = 0; i < n; ++i) {
    x[i] = rmax(xlo[i], rmin(xhi[i], y[i]));
    dm -= a[i]*x[i];
  }
  if (dm == 0) return;
  if (dm > 0) {
    double fac = 0;
    for (int i = 0; i < n; ++i) fac += a[i]*(xhi[i] - x[i]);
    if (fac > 0) {
      fac = dm/fac;
      for (int i = 0; i < n; ++i) x[i] += fac*(xhi[i] - x[i]);
    }
  } else if (dm < 0) {
    double fac = 0;
    for (int i = 0; i < n; ++i) fac += a[i]*(x[i] - xlo[i]);
    if (fac > 0) {
      fac = dm/fac;
      for (int i = 0; i < n; ++i) x[i] += fac*(x[i] - xlo[i]);
    }
  }
  if (clip)
    for (int i = 0; i < n; ++i) x[i] = rmax(xlo[i], rmin(xhi[i], x[i]));
}

// cedr_local_inl.hpp:309-330
int solve_1eq_nonneg(int n, const double* a, double b, const double* y, double* x,
                     const double* w, int method) {
  if (n > 16) fail("solve_1eq_nonneg: n > 16");
  if (b < 0) return -1;
  double zero[16], xhi[16];
  for (int i = 0; i < n; ++i) { zero[i] = 0; xhi[i] = b/a[i]; }
  if (method == 1) { caas_local(n, a, b, zero, xhi, y, x, true); return 1; }
  if (n == 2) return solve_1eq_bc_qp_2d(w, a, b, zero, xhi, y, x, true, true);
  return solve_1eq_bc_qp(n, w, a, b, zero, xhi, y, x, 100);
}

// cedr_qlt_inl.hpp:69-99
static void adjust_bounds(double bnd[2], const double rhom[2], double extra) {
  double q[2];
  for (int i = 0; i < 2; ++i) q[i] = bnd[i]/rhom[i];
  if (extra < 0) {
    int i0, i1;
    if (q[0] >= q[1]) { i0 = 0; i1 = 1; } else { i0 = 1; i1 = 0; }
    const double gap = (q[i1] - q[i0])*rhom[i0];
    if (gap <= extra) { bnd[i0] += extra; return; }
  } else {
    int i0, i1;
    if (q[0] <= q[1]) { i0 = 0; i1 = 1; } else { i0 = 1; i1 = 0; }
    const double gap = (q[i1] - q[i0])*rhom[i0];
    if (gap >= extra) { bnd[i0] += extra; return; }
  }
  const double tot = bnd[0] + bnd[1] + extra;
  const double rtot = rhom[0] + rhom[1];
  const double qt = tot/rtot;
  for (int i = 0; i < 2; ++i) bnd[i] = qt*rhom[i];
}

// cedr_qlt_inl.hpp:120-173
static void node_problem_mass(double rhom, const double* pd, double Qm,
                              double rhom0, const double* k0d, double* Qm0,
                              double rhom1, const double* k1d, double* Qm1, bool prefer) {
  (void) rhom;
  double lo[2] = {k0d[0], k1d[0]};
  double orig[2] = {k0d[1], k1d[1]};
  double hi[2] = {k0d[2], k1d[2]};
  {
    const double Qm_min = pd[0], Qm_max = pd[2];
    const bool below = Qm < Qm_min, above = Qm > Qm_max;
    if (below || above) {
      const double tol = 10*EPS;
      const double disc = below ? Qm_min - Qm : Qm - Qm_max;
      if (disc > tol*(Qm_max - Qm_min)) {
        const double rk[2] = {rhom0, rhom1};
        adjust_bounds(below ? lo : hi, rk, Qm - (below ? Qm_min : Qm_max));
      }
    } else if (Qm == pd[1] &&
               orig[0] >= lo[0] && orig[0] <= hi[0] &&
               orig[1] >= lo[1] && orig[1] <= hi[1]) {
      *Qm0 = orig[0];
      *Qm1 = orig[1];
      return;
    }
  }
  const double ones[2] = {1, 1};
  const double w[2] = {1/rhom0, 1/rhom1};
  double xk[2] = {k0d[1], k1d[1]};
  solve_1eq_bc_qp_2d(w, ones, Qm, lo, hi, orig, xk, !prefer, !prefer);
  *Qm0 = xk[0];
  *Qm1 = xk[1];
}

// cedr_qlt_inl.hpp:175-203
void solve_node_problem(int pt, double rhom, const double* pd, double Qm,
                        double rhom0, const double* k0d, double* Qm0,
                        double rhom1, const double* k1d, double* Qm1, bool prefer) {
  if ((pt & CONSISTENT) && !(pt & SHAPEPRESERVE)) {
    double mpd[3], mk0[3], mk1[3];
    mpd[0] = pd[0]*rhom;   mpd[1] = pd[1];  mpd[2] = pd[2]*rhom;
    mk0[0] = k0d[0]*rhom0; mk0[1] = k0d[1]; mk0[2] = k0d[2]*rhom0;
    mk1[0] = k1d[0]*rhom1; mk1[1] = k1d[1]; mk1[2] = k1d[2]*rhom1;
    node_problem_mass(rhom, mpd, Qm, rhom0, mk0, Qm0, rhom1, mk1, Qm1, prefer);
  } else if (pt & NONNEGATIVE) {
    const double ones[2] = {1, 1};
    const double w[2] = {1/rhom0, 1/rhom1};
    double orig[2] = {k0d[0], k1d[0]};
    double xk[2] = {k0d[0], k1d[0]};
    solve_1eq_nonneg(2, ones, Qm, orig, xk, w, 0);
    *Qm0 = xk[0];
    *Qm1 = xk[1];
  } else {
    node_problem_mass(rhom, pd, Qm, rhom0, k0d, Qm0, rhom1, k1d, Qm1, prefer);
  }
}

// ======================================================================= QLT
// cedr_qlt.cpp:85-96 + cedr_qlt_inl.hpp:101-107
int canonical_problem_type(int mask) {
  switch (mask) {
    case SHAPEPRESERVE: case SHAPEPRESERVE|CONSISTENT: return SHAPEPRESERVE|CONSISTENT;
    case CONSERVE|SHAPEPRESERVE: case CONSERVE|SHAPEPRESERVE|CONSISTENT:
      return CONSERVE|SHAPEPRESERVE|CONSISTENT;
    case CONSISTENT: return CONSISTENT;
    case CONSERVE|CONSISTENT: return CONSERVE|CONSISTENT;
    case NONNEGATIVE: return NONNEGATIVE;
    case CONSERVE|NONNEGATIVE: return CONSERVE|NONNEGATIVE;
    default: return -1;
  }
}

static const int kProbOrder[6] = {
  SHAPEPRESERVE|CONSISTENT, CONSERVE|SHAPEPRESERVE|CONSISTENT, CONSISTENT,
  CONSERVE|CONSISTENT, NONNEGATIVE, CONSERVE|NONNEGATIVE };

static int l2r_bulk(int m) {  // cedr_qlt_inl.hpp:109-117
  if (m & NONNEGATIVE) return (m & CONSERVE) ? 2 : 1;
  return (m & CONSERVE) ? 4 : 3;
}
static int r2l_bulk(int m) { return (m & SHAPEPRESERVE) ? 1 : 3; }  // cedr_qlt.cpp:98-102

// cedr_qlt.cpp:104-157
QltMeta make_qlt_meta(const std::vector<int>& declared) {
  QltMeta m;
  m.nt = (int) declared.size();
  for (int t : declared) {
    const int c = canonical_problem_type(t);
    if (c < 0) fail("Invalid problem type.");
    m.trcr2prob.push_back(c);
  }
  m.bidx2trcr.assign(m.nt, -1); m.trcr2bidx.assign(m.nt, -1);
  m.trcr2bl2r.assign(m.nt, -1); m.trcr2br2l.assign(m.nt, -1);
  m.prob2trcrptr[0] = 0; m.prob2bl2r[0] = 1; m.prob2br2l[0] = 0;
  for (int pi = 0; pi < 6; ++pi) {
    m.prob2trcrptr[pi+1] = m.prob2trcrptr[pi];
    const int lsz = l2r_bulk(kProbOrder[pi]), rsz = r2l_bulk(kProbOrder[pi]);
    for (int ti = 0; ti < m.nt; ++ti) {
      if (m.trcr2prob[ti] != kProbOrder[pi]) continue;
      const int cnt = m.prob2trcrptr[pi+1] - m.prob2trcrptr[pi];
      m.trcr2bl2r[ti] = m.prob2bl2r[pi] + cnt*lsz;
      m.trcr2br2l[ti] = m.prob2br2l[pi] + cnt*rsz;
      m.bidx2trcr[m.prob2trcrptr[pi+1]++] = ti;
    }
    const int ni = m.prob2trcrptr[pi+1] - m.prob2trcrptr[pi];
    m.prob2bl2r[pi+1] = m.prob2bl2r[pi] + ni*lsz;
    m.prob2br2l[pi+1] = m.prob2br2l[pi] + ni*rsz;
  }
  for (int bi = 0; bi < m.nt; ++bi) m.trcr2bidx[m.bidx2trcr[bi]] = bi;
  return m;
}

Qlt::Qlt(const CallerTree& tree, int ncells, int nranks_, bool prefer)
  : nranks(nranks_), prefer_mass_con(prefer) {
  r.resize(nranks);
  for (int k = 0; k < nranks; ++k) {
    r[k].rank = k;
    r[k].ns = analyze(tree, ncells, k);
    if (r[k].ns.levels.empty() || r[k].ns.levels[0].nodes.empty())
      fail("QLT does not support 0 cells on a rank.");  // cedr_qlt.cpp:235
  }
}

void Qlt::declare_tracers(const std::vector<int>& types) {
  md = make_qlt_meta(types);
  for (auto& k : r) {
    k.l2r.assign((size_t) k.ns.nslots*md.l2rndps(), 0.0);
    k.r2l.assign((size_t) k.ns.nslots*md.r2lndps(), 0.0);
  }
}

void Qlt::owned_glblcells(int rank, std::vector<long>& g) const {
  const NodeSets& ns = r[rank].ns;
  g.resize(ns.levels[0].nodes.size());
  for (int idx : ns.levels[0].nodes) g[ns.nodes[idx].offset] = ns.nodes[idx].id;
}

void Qlt::set_rhom(int rank, int lci, double rhom) {
  r[rank].l2r[(size_t) md.l2rndps()*lci] = rhom;
}

void Qlt::set_Qm(int rank, int lci, int ti, double Qm, double Qm_min, double Qm_max,
                 double Qm_prev) {
  const int ndps = md.l2rndps();
  double* bd = &r[rank].l2r[(size_t) ndps*lci + md.trcr2bl2r[ti]];
  const int pt = md.trcr2prob[ti];
  int next = 0;
  if (pt & SHAPEPRESERVE) {
    bd[0] = Qm_min; bd[1] = Qm; bd[2] = Qm_max; next = 3;
  } else if (pt & CONSISTENT) {
    const double rhom = r[rank].l2r[(size_t) ndps*lci];
    bd[0] = Qm_min/rhom; bd[1] = Qm; bd[2] = Qm_max/rhom; next = 3;
  } else if (pt & NONNEGATIVE) {
    bd[0] = Qm; next = 1;
  } else fail("set_Q: invalid problem_type.");
  if (pt & CONSERVE) {
    if (Qm_prev == std::numeric_limits<double>::infinity())
      fail("Qm_prev was not provided to set_Q.");
    bd[next] = Qm_prev;
  }
}

double Qlt::get_Qm(int rank, int lci, int ti) const {
  return r[rank].r2l[(size_t) md.r2lndps()*lci + md.trcr2br2l[ti]];
}

struct QltOps : SweepOps {
  Qlt& q;
  QltOps(Qlt& q_) : q(q_) {}
  double* up_buf(int k) override { return q.r[k].l2r.data(); }
  double* down_buf(int k) override { return q.r[k].r2l.data(); }
  int up_ndps() const override { return q.md.l2rndps(); }
  int down_ndps() const override { return q.md.r2lndps(); }

  // cedr_qlt.cpp:388-429 (host branch)
  void up(int k, int il) override {
    QltRank& R = q.r[k];
    const NodeSets& ns = R.ns;
    const QltMeta& md = q.md;
    const int ndps = md.l2rndps();
    const std::vector<int>& nodes = ns.levels[il].nodes;
    const int nn = (int) nodes.size();
    double* const l2r = R.l2r.data();
#ifdef _OPENMP
#   pragma omp parallel for num_threads(q.nthreads) if (q.nthreads > 1)
#endif
    for (int ni = 0; ni < nn; ++ni) {
      const NsNode& n = ns.nodes[nodes[ni]];
      if (!n.nkids) continue;
      const size_t o = (size_t) n.offset*ndps;
      const size_t o0 = (size_t) ns.nodes[n.kids[0]].offset*ndps;
      const size_t o1 = (size_t) ns.nodes[n.kids[1]].offset*ndps;
      l2r[o] = l2r[o0] + l2r[o1];
      for (int pti = 0; pti < 6; ++pti) {
        const int pt = kProbOrder[pti];
        const bool nn_ = pt & NONNEGATIVE, sp = pt & SHAPEPRESERVE, cs = pt & CONSERVE;
        for (int bi = md.prob2trcrptr[pti]; bi < md.prob2trcrptr[pti+1]; ++bi) {
          const int bdi = md.trcr2bl2r[md.bidx2trcr[bi]];
          double* me = l2r + o + bdi;
          const double* k0 = l2r + o0 + bdi;
          const double* k1 = l2r + o1 + bdi;
          if (nn_) {
            me[0] = k0[0] + k1[0];
            if (cs) me[1] = k0[1] + k1[1];
          } else {
            me[0] = sp ? k0[0] + k1[0] : rmin(k0[0], k1[0]);
            me[1] = k0[1] + k1[1];
            me[2] = sp ? k0[2] + k1[2] : rmax(k0[2], k1[2]);
            if (cs) me[3] = k0[3] + k1[3];
          }
        }
      }
    }
  }

  // cedr_qlt.cpp:441-476
  void root(int k) override {
    QltRank& R = q.r[k];
    const NodeSets& ns = R.ns;
    if (ns.levels.empty() || ns.levels.back().nodes.size() != 1 ||
        ns.nodes[ns.levels.back().nodes[0]].parent >= 0) return;
    const QltMeta& md = q.md;
    const NsNode& n = ns.nodes[ns.levels.back().nodes[0]];
    const int l2n = md.l2rndps(), r2n = md.r2lndps();
    for (int bi = 0; bi < md.nt; ++bi) {
      const int ti = md.bidx2trcr[bi];
      const int pt = md.trcr2prob[ti];
      const int lb = md.trcr2bl2r[ti], rb = md.trcr2br2l[ti];
      const int os = (pt & CONSERVE) ? l2r_bulk(pt) - 1 : ((pt & NONNEGATIVE) ? 0 : 1);
      R.r2l[(size_t) n.offset*r2n + rb] = R.l2r[(size_t) n.offset*l2n + lb + os];
      if ((pt & CONSISTENT) && !(pt & SHAPEPRESERVE)) {
        R.r2l[(size_t) n.offset*r2n + rb + 1] = R.l2r[(size_t) n.offset*l2n + lb + 0];
        R.r2l[(size_t) n.offset*r2n + rb + 2] = R.l2r[(size_t) n.offset*l2n + lb + 2];
      }
    }
  }

  // cedr_qlt.cpp:567-603 (host branch) with :490-523
  void down(int k, int il) override {
    QltRank& R = q.r[k];
    const NodeSets& ns = R.ns;
    const QltMeta& md = q.md;
    const int l2n = md.l2rndps(), r2n = md.r2lndps();
    const std::vector<int>& nodes = ns.levels[il].nodes;
    const int nn = (int) nodes.size();
    double* const l2r = R.l2r.data();
    double* const r2l = R.r2l.data();
    const bool prefer = q.prefer_mass_con;
#ifdef _OPENMP
#   pragma omp parallel for num_threads(q.nthreads) if (q.nthreads > 1)
#endif
    for (int ni = 0; ni < nn; ++ni) {
      const NsNode& n = ns.nodes[nodes[ni]];
      if (!n.nkids) continue;
      const size_t no = n.offset;
      const size_t k0o = ns.nodes[n.kids[0]].offset, k1o = ns.nodes[n.kids[1]].offset;
      for (int pti = 0; pti < 6; ++pti) {
        const int pt = kProbOrder[pti];
        for (int bi = md.prob2trcrptr[pti]; bi < md.prob2trcrptr[pti+1]; ++bi) {
          const int ti = md.bidx2trcr[bi];
          const int lb = md.trcr2bl2r[ti], rb = md.trcr2br2l[ti];
          if ((pt & CONSISTENT) && !(pt & SHAPEPRESERVE)) {
            const double q_min = r2l[no*r2n + rb + 1], q_max = r2l[no*r2n + rb + 2];
            l2r[no*l2n + lb + 0] = q_min;
            l2r[no*l2n + lb + 2] = q_max;
            const size_t ko[2] = {k0o, k1o};
            for (int c = 0; c < 2; ++c) {
              l2r[ko[c]*l2n + lb + 0] = q_min;
              l2r[ko[c]*l2n + lb + 2] = q_max;
              r2l[ko[c]*r2n + rb + 1] = q_min;
              r2l[ko[c]*r2n + rb + 2] = q_max;
            }
          }
          solve_node_problem(pt, l2r[no*l2n], l2r + no*l2n + lb, r2l[no*r2n + rb],
                             l2r[k0o*l2n], l2r + k0o*l2n + lb, r2l + k0o*r2n + rb,
                             l2r[k1o*l2n], l2r + k1o*l2n + lb, r2l + k1o*r2n + rb,
                             prefer);
        }
      }
    }
  }
};

void Qlt::run() {
  QltOps ops(*this);
  std::vector<const NodeSets*> p;
  for (auto& k : r) p.push_back(&k.ns);
  run_sweep(p, ops);
}

// ====================================================================== CAAS
Caas::Caas(const std::vector<int>& nl, const std::vector<int>& types, int accum_)
  : nranks((int) nl.size()), nt((int) types.size()), accum(accum_), probs(types), nlcl(nl) {
  if (nt == 0) fail("#tracers is 0.");  // cedr_caas.cpp:61
  for (int t : types) {
    if (!(t & SHAPEPRESERVE)) fail("CAAS does not support ! shapepreserve yet.");  // :52
    if (t & CONSERVE) need_conserve = true;
  }
  d.resize(nranks); send.resize(nranks);
  for (int k = 0; k < nranks; ++k) {
    if (nlcl[k] == 0) fail("CAAS does not support 0 cells on a rank.");  // :46
    const int e = need_conserve ? 1 : 0;
    d[k].assign((size_t) nlcl[k]*((3 + e)*nt + 1), 0.0);  // :79
    if (accum > 0 && nlcl[k] % accum) fail("nlclcells % n_accum_in_place != 0");
    send[k].assign((size_t) 4*nt*(accum > 0 ? nlcl[k]/accum : 1), 0.0);  // :81
  }
  recv.assign(4*nt, 0.0);
}

void Caas::set_Qm(int k, int i, int ti, double Qm, double Qm_min, double Qm_max, double Qm_prev) {
  const size_t n = nlcl[k];
  d[k][(1 + ti)*n + i] = Qm;
  d[k][(1 + nt + ti)*n + i] = Qm_min;
  d[k][(1 + 2*nt + ti)*n + i] = Qm_max;
  if (need_conserve) d[k][(1 + 3*nt + ti)*n + i] = Qm_prev;
}

// cedr_caas.cpp:129-201. Serial team of size 1 => cells are summed in order.
void Caas::reduce_locally(int k) {
  const size_t n = nlcl[k];
  double* const D = d[k].data();
  double* const S = send[k].data();
  if (accum > 0) {
    const int na = accum;
    const size_t nacc = n/na;
    for (size_t j = 0; j < (size_t) nt*nacc; ++j) {
      const size_t t = j/nacc, bi = j % nacc, os = (t + 1)*n;
      double ac = 0, at = 0;
      for (int ai = 0; ai < na; ++ai) {
        const size_t i = na*bi + ai;
        const double Qm = D[os + i];
        const double term = (probs[t] & CONSERVE) ? D[os + n*3*nt + i] : Qm;
        const double clip = rmin(D[os + n*2*nt + i], rmax(D[os + n*nt + i], Qm));
        D[os + i] = clip;
        ac += clip; at += term;
      }
      S[nacc*t + bi] = ac;
      S[nacc*(nt + t) + bi] = at;
    }
    for (size_t j = 0; j < (size_t) 2*nt*nacc; ++j) {
      const size_t t = 2*nt + j/nacc, bi = j % nacc, os = (t - nt + 1)*n;
      double ae = 0;
      for (int ai = 0; ai < na; ++ai) ae += D[os + na*bi + ai];
      S[nacc*t + bi] = ae;
    }
    return;
  }
#ifdef _OPENMP
# pragma omp parallel for num_threads(nthreads) if (nthreads > 1)
#endif
  for (int t = 0; t < nt; ++t) {
    const size_t os = (size_t) (t + 1)*n;
    double ac = 0, at = 0;
    for (size_t i = 0; i < n; ++i) {
      const double Qm = D[os + i];
      const double term = (probs[t] & CONSERVE) ? D[os + n*3*nt + i] : Qm;
      const double clip = rmin(D[os + n*2*nt + i], rmax(D[os + n*nt + i], Qm));
      D[os + i] = clip;
      ac += clip; at += term;
    }
    S[t] = ac;
    S[nt + t] = at;
  }
#ifdef _OPENMP
# pragma omp parallel for num_threads(nthreads) if (nthreads > 1)
#endif
  for (int r2 = 0; r2 < 2*nt; ++r2) {
    const size_t os = (size_t) (nt + r2 + 1)*n;
    double a = 0;
    for (size_t i = 0; i < n; ++i) a += D[os + i];
    S[2*nt + r2] = a;
  }
}

// cedr_caas.cpp:203-209, or the test's UserAllReducer (:304-321) if accum > 0.
// MPI leaves the cross-rank order unspecified; ranks are summed in rank order.
void Caas::reduce_globally() {
  std::fill(recv.begin(), recv.end(), 0.0);
  for (int k = 0; k < nranks; ++k) {
    if (accum > 0) {
      const size_t nl = nlcl[k]/accum;
      std::vector<double>& s = send[k];
      for (int f = 0; f < 4*nt; ++f) {
        s[f] = s[nl*f];
        for (size_t i = 1; i < nl; ++i) s[f] += s[nl*f + i];
      }
    }
    for (int f = 0; f < 4*nt; ++f)
      recv[f] = (k == 0) ? send[k][f] : recv[f] + send[k][f];
  }
}

// cedr_caas.cpp:211-253
void Caas::finish_locally(int k) {
  const size_t n = nlcl[k];
  double* const D = d[k].data();
#ifdef _OPENMP
# pragma omp parallel for num_threads(nthreads) if (nthreads > 1)
#endif
  for (int t = 0; t < nt; ++t) {
    const size_t os = (size_t) (t + 1)*n;
    const double clip_sum = recv[t], sum = recv[nt + t];
    const double m = sum - clip_sum;
    if (m < 0) {
      double fac = clip_sum - recv[2*nt + t];
      if (fac > 0) {
        fac = m/fac;
        for (size_t i = 0; i < n; ++i) {
          const double lo = D[os + n*nt + i];
          double& Qm = D[os + i];
          Qm += fac*(Qm - lo);
          Qm = rmax(lo, Qm);
        }
      }
    } else if (m > 0) {
      double fac = recv[3*nt + t] - clip_sum;
      if (fac > 0) {
        fac = m/fac;
        for (size_t i = 0; i < n; ++i) {
          const double hi = D[os + n*2*nt + i];
          double& Qm = D[os + i];
          Qm += fac*(hi - Qm);
          Qm = rmin(hi, Qm);
        }
      }
    }
  }
}

void Caas::run() {
  for (int k = 0; k < nranks; ++k) reduce_locally(k);
  reduce_globally();
  for (int k = 0; k < nranks; ++k) finish_locally(k);
}

// =========================================================== BFB tree all-reduce
struct BfbOps : SweepOps {
  const std::vector<NodeSets>& all;
  int nf;
  std::vector<std::vector<double> > bd;
  BfbOps(const std::vector<NodeSets>& a, int nf_) : all(a), nf(nf_) {
    for (const auto& ns : all) bd.push_back(std::vector<double>((size_t) ns.nslots*nf, 0.0));
  }
  double* up_buf(int r) override { return bd[r].data(); }
  double* down_buf(int r) override { return bd[r].data(); }
  int up_ndps() const override { return nf; }
  int down_ndps() const override { return nf; }
  void up(int r, int il) override {  // cedr_bfb_tree_allreduce.cpp:114-124
    const NodeSets& ns = all[r];
    for (int idx : ns.levels[il].nodes) {
      const NsNode& n = ns.nodes[idx];
      if (!n.nkids) continue;
      double* dd = &bd[r][(size_t) n.offset*nf];
      for (int j = 0; j < nf; ++j) dd[j] = 0;
      for (int i = 0; i < n.nkids; ++i) {
        const double* s = &bd[r][(size_t) ns.nodes[n.kids[i]].offset*nf];
        for (int j = 0; j < nf; ++j) dd[j] += s[j];
      }
    }
  }
  void root(int) override {}
  void down(int r, int il) override {  // :141-149
    const NodeSets& ns = all[r];
    for (int idx : ns.levels[il].nodes) {
      const NsNode& n = ns.nodes[idx];
      if (!n.nkids) continue;
      const double* s = &bd[r][(size_t) n.offset*nf];
      for (int i = 0; i < n.nkids; ++i) {
        double* dd = &bd[r][(size_t) ns.nodes[n.kids[i]].offset*nf];
        for (int j = 0; j < nf; ++j) dd[j] = s[j];
      }
    }
  }
};

void bfb_tree_allreduce(const std::vector<NodeSets>& all, int nf, bool transpose,
                        const std::vector<std::vector<double> >& send,
                        std::vector<std::vector<double> >& recv) {
  BfbOps ops(all, nf);
  for (size_t r = 0; r < all.size(); ++r) {  // :91-102
    const NodeSets& ns = all[r];
    const int nl = (int) ns.levels[0].nodes.size();
    if ((int) send[r].size() != nl*nf) fail("bfb send size");
    for (int id = 0; id < nl; ++id) {
      const NsNode& n = ns.nodes[ns.levels[0].nodes[id]];
      double* dd = &ops.bd[r][(size_t) n.offset*nf];
      for (int j = 0; j < nf; ++j)
        dd[j] = transpose ? send[r][(size_t) nl*j + id] : send[r][(size_t) id*nf + j];
    }
  }
  std::vector<const NodeSets*> p;
  for (const auto& ns : all) p.push_back(&ns);
  run_sweep(p, ops);
  recv.resize(all.size());
  for (size_t r = 0; r < all.size(); ++r) {  // :68-76
    const NsNode& n = all[r].nodes[all[r].levels[0].nodes[0]];
    recv[r].assign(&ops.bd[r][(size_t) n.offset*nf], &ops.bd[r][(size_t) n.offset*nf] + nf);
  }
}

}  // namespace oracle
