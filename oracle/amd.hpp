// ORACLE / CPU baseline — test infrastructure only (see bal_model.hpp header).
//
// Approximate minimum degree ordering — the fill-reducing ordering Eigen::SimplicialLDLT applies by default
// (Eigen 3.4 `AMDOrdering`, Eigen/src/OrderingMethods/Amd.h; the reference's EigenLDLTSolver / EigenSchurLDLTSolver use
// SimplicialLDLT with its default template arguments: /root/reference/src/eigen_solver.cpp:10-13, so every factorisation of the
// "eigen_solver CPU path" is AMD-ordered).  Eigen is an un-vendored dependency (CMakeLists.txt:25, Dockerfile: libeigen3-dev 3.4.0)
// and absent here; its AMD is the algorithm of Amestoy, Davis & Duff, "An approximate minimum degree ordering algorithm"
// (SIAM J. Matrix Anal. Appl. 17, 1996) / "Algorithm 837: AMD" (ACM TOMS 30, 2004), in the compact form T. Davis published in
// "Direct Methods for Sparse Linear Systems" (SIAM 2006, section 7.1).  That published algorithm is restated below:
//   * quotient graph kept IN PLACE in one index array (variables list adjacent elements first, then adjacent variables;
//     elements list their variables), compacted when it runs out of room;
//   * pivot = a variable of minimum APPROXIMATE external degree  d_i = min(n - k, d_i + |Lk \ i|, |Ai \ i| + |Lk \ i| + sum_e |Le \ Lk|),
//     the set differences |Le \ Lk| for all elements at once by the w[] counters (one pass over the pivot's variables);
//   * aggressive element absorption (an element with Le \ Lk empty is absorbed into k), mass elimination (a variable whose
//     new degree is 0 is eliminated with the pivot), supervariables (variables of the new element with identical adjacency,
//     found by hashing, merge: nv counts the merged variables — the nine scalars of a camera become one supervariable at once);
//   * rows denser than max(16, 10 sqrt(n)) are set aside and ordered last;
//   * the pivot order is the postorder of the assembly tree.
// Output: perm[new] = old, as sparse_ldlt.hpp's analyze() takes it.  Not pinned by a reference vector (no Eigen here, SURVEY
// section 8c: "parity unpinned"): tests/test_oracle_solvers.py checks that it is a permutation, that its fill is within a few
// per cent of EXACT minimum degree on BAL-shaped reduced systems, far below the natural order's, and that solves agree.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace gro {

namespace amd_detail {
inline int64_t flip(int64_t i) { return -i - 2; }
// reset the work counters when the mark is about to overflow
inline int64_t wclear(int64_t mark, int64_t lemax, std::vector<int64_t> &w, int64_t n) {
  if (mark < 2 || mark + lemax < 0) {
    for (int64_t k = 0; k < n; ++k) if (w[k] != 0) w[k] = 1;
    mark = 2;
  }
  return mark;
}
// postorder of the tree rooted at j (children lists head / next), appended to post from position k
inline int64_t tdfs(int64_t j, int64_t k, std::vector<int64_t> &head, const std::vector<int64_t> &next, std::vector<int64_t> &post, std::vector<int64_t> &stack) {
  int64_t top = 0;
  stack[0] = j;
  while (top >= 0) {
    const int64_t p = stack[top], i = head[p];
    if (i == -1) { --top; post[k++] = p; }
    else { head[p] = next[i]; stack[++top] = i; }
  }
  return k;
}
} // namespace amd_detail

// Ap / Ai: scalar CSC of the UPPER triangle (row <= col) of a symmetric n x n matrix (any diagonal entries are ignored).
inline std::vector<int64_t> amd_order(int64_t n, const int64_t *Ap, const int64_t *Ai) {
  using namespace amd_detail;
  std::vector<int64_t> P(n + 1, 0);
  if (n <= 0) return {};
  int64_t dense = std::max<int64_t>(16, (int64_t)(10.0 * std::sqrt((double)n)));
  dense = std::min<int64_t>(n - 2, dense);
  // ---- C = pattern of A + A^T without the diagonal, with elbow room ------------------------------------------------------
  std::vector<int64_t> Cp(n + 1, 0), len(n + 1, 0);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) { const int64_t i = Ai[p]; if (i != j) { len[i]++; len[j]++; } }
  int64_t cnz = 0;
  for (int64_t j = 0; j < n; ++j) { Cp[j] = cnz; cnz += len[j]; }
  Cp[n] = cnz;
  const int64_t nzmax = cnz + cnz / 5 + 2 * n + 16;
  std::vector<int64_t> Ci(nzmax, 0);
  {
    std::vector<int64_t> at(Cp.begin(), Cp.end() - 1);
    for (int64_t j = 0; j < n; ++j)
      for (int64_t p = Ap[j]; p < Ap[j + 1]; ++p) { const int64_t i = Ai[p]; if (i != j) { Ci[at[i]++] = j; Ci[at[j]++] = i; } }
  }
  std::vector<int64_t> nv(n + 1, 1), next(n + 1, -1), head(n + 1, -1), elen(n + 1, 0), degree(n + 1, 0), w(n + 1, 1), hhead(n + 1, -1), last(n + 1, -1);
  for (int64_t i = 0; i < n; ++i) degree[i] = len[i];
  int64_t lemax = 0, mindeg = 0, nel = 0;
  int64_t mark = wclear(0, 0, w, n);
  elen[n] = -2; Cp[n] = -1; w[n] = 0; // n: the root that collects the dense rows
  for (int64_t i = 0; i < n; ++i) {
    const int64_t d = degree[i];
    if (d == 0) { elen[i] = -2; nel++; Cp[i] = -1; w[i] = 0; }                         // no neighbour: an element at once
    else if (d > dense) { nv[i] = 0; elen[i] = -1; nel++; Cp[i] = flip(n); nv[n]++; }  // dense: ordered last
    else { if (head[d] != -1) last[head[d]] = i; next[i] = head[d]; head[d] = i; }     // degree list d
  }
  while (nel < n) {
    // ---- a variable of minimum approximate degree becomes the pivot k ---------------------------------------------------
    int64_t k = -1;
    for (; mindeg < n && (k = head[mindeg]) == -1; ++mindeg) {}
    if (next[k] != -1) last[next[k]] = -1;
    head[mindeg] = next[k];
    const int64_t elenk = elen[k];
    int64_t nvk = nv[k];
    nel += nvk;
    // ---- compaction of the index array when the new element may not fit ---------------------------------------------------
    if (elenk > 0 && cnz + mindeg >= nzmax) {
      for (int64_t j = 0; j < n; ++j) { const int64_t p = Cp[j]; if (p >= 0) { Cp[j] = Ci[p]; Ci[p] = flip(j); } }
      int64_t q = 0;
      for (int64_t p = 0; p < cnz;) {
        const int64_t j = flip(Ci[p++]);
        if (j >= 0) {
          Ci[q] = Cp[j]; Cp[j] = q++;
          for (int64_t k3 = 0; k3 < len[j] - 1; ++k3) Ci[q++] = Ci[p++];
        }
      }
      cnz = q;
    }
    // ---- the new element Lk = (Ak U union of the Le of k's elements) \ k ---------------------------------------------------
    int64_t dk = 0;
    nv[k] = -nvk;
    int64_t p = Cp[k];
    const int64_t pk1 = (elenk == 0) ? p : cnz;
    int64_t pk2 = pk1;
    for (int64_t k1 = 1; k1 <= elenk + 1; ++k1) {
      int64_t e, pj, ln;
      if (k1 > elenk) { e = k; pj = p; ln = len[k] - elenk; }
      else { e = Ci[p++]; pj = Cp[e]; ln = len[e]; }
      for (int64_t k2 = 1; k2 <= ln; ++k2) {
        const int64_t i = Ci[pj++];
        const int64_t nvi = nv[i];
        if (nvi <= 0) continue;            // dead or already in Lk
        dk += nvi; nv[i] = -nvi; Ci[pk2++] = i;
        if (next[i] != -1) last[next[i]] = last[i];
        if (last[i] != -1) next[last[i]] = next[i]; else head[degree[i]] = next[i];
      }
      if (e != k) { Cp[e] = flip(k); w[e] = 0; } // absorbed into k
    }
    if (elenk != 0) cnz = pk2;
    degree[k] = dk; Cp[k] = pk1; len[k] = pk2 - pk1; elen[k] = -2;
    // ---- |Le \ Lk| for every element adjacent to a variable of Lk -----------------------------------------------------------
    mark = wclear(mark, lemax, w, n);
    for (int64_t pk = pk1; pk < pk2; ++pk) {
      const int64_t i = Ci[pk], eln = elen[i];
      if (eln <= 0) continue;
      const int64_t nvi = -nv[i], wnvi = mark - nvi;
      for (int64_t q = Cp[i]; q <= Cp[i] + eln - 1; ++q) {
        const int64_t e = Ci[q];
        if (w[e] >= mark) w[e] -= nvi;
        else if (w[e] != 0) w[e] = degree[e] + wnvi;
      }
    }
    // ---- approximate degrees of the variables of Lk -------------------------------------------------------------------------
    for (int64_t pk = pk1; pk < pk2; ++pk) {
      const int64_t i = Ci[pk], p1 = Cp[i], p2 = p1 + elen[i] - 1;
      int64_t pn = p1, h = 0, d = 0;
      for (int64_t q = p1; q <= p2; ++q) {
        const int64_t e = Ci[q];
        if (w[e] != 0) {
          const int64_t dext = w[e] - mark;
          if (dext > 0) { d += dext; Ci[pn++] = e; h += e; }
          else { Cp[e] = flip(k); w[e] = 0; } // aggressive absorption: Le \ Lk is empty
        }
      }
      elen[i] = pn - p1 + 1;
      const int64_t p3 = pn, p4 = p1 + len[i];
      for (int64_t q = p2 + 1; q < p4; ++q) {
        const int64_t j = Ci[q], nvj = nv[j];
        if (nvj <= 0) continue;
        d += nvj; Ci[pn++] = j; h += j;
      }
      if (d == 0) { // mass elimination: i has no neighbour outside Lk
        Cp[i] = flip(k);
        const int64_t nvi = -nv[i];
        dk -= nvi; nvk += nvi; nel += nvi; nv[i] = 0; elen[i] = -1;
      } else {
        degree[i] = std::min(degree[i], d);
        Ci[pn] = Ci[p3]; Ci[p3] = Ci[p1]; Ci[p1] = k; // k leads i's element list
        len[i] = pn - p1 + 1;
        h = (h < 0 ? -h : h) % n;
        next[i] = hhead[h]; hhead[h] = i; last[i] = h; // hash bucket (supervariable detection)
      }
    }
    degree[k] = dk;
    lemax = std::max(lemax, dk);
    mark = wclear(mark + lemax, lemax, w, n);
    // ---- supervariables: variables of Lk with the same adjacency -------------------------------------------------------------
    for (int64_t pk = pk1; pk < pk2; ++pk) {
      int64_t i = Ci[pk];
      if (nv[i] >= 0) continue;
      const int64_t h = last[i];
      i = hhead[h]; hhead[h] = -1;
      for (; i != -1 && next[i] != -1; i = next[i], ++mark) {
        const int64_t ln = len[i], eln = elen[i];
        for (int64_t q = Cp[i] + 1; q <= Cp[i] + ln - 1; ++q) w[Ci[q]] = mark;
        int64_t jlast = i;
        for (int64_t j = next[i]; j != -1;) {
          bool ok = (len[j] == ln) && (elen[j] == eln);
          for (int64_t q = Cp[j] + 1; ok && q <= Cp[j] + ln - 1; ++q) if (w[Ci[q]] != mark) ok = false;
          if (ok) { Cp[j] = flip(i); nv[i] += nv[j]; nv[j] = 0; elen[j] = -1; j = next[j]; next[jlast] = j; }
          else { jlast = j; j = next[j]; }
        }
      }
    }
    // ---- the surviving variables of Lk go back to the degree lists ------------------------------------------------------------
    int64_t pe = pk1;
    for (int64_t pk = pk1; pk < pk2; ++pk) {
      const int64_t i = Ci[pk], nvi = -nv[i];
      if (nvi <= 0) continue;
      nv[i] = nvi;
      int64_t d = degree[i] + dk - nvi;
      d = std::min(d, n - nel - nvi);
      if (head[d] != -1) last[head[d]] = i;
      next[i] = head[d]; last[i] = -1; head[d] = i;
      mindeg = std::min(mindeg, d);
      degree[i] = d;
      Ci[pe++] = i;
    }
    nv[k] = nvk;
    if ((len[k] = pe - pk1) == 0) { Cp[k] = -1; w[k] = 0; }
    if (elenk != 0) cnz = pe;
  }
  // ---- postorder of the assembly tree ---------------------------------------------------------------------------------------
  for (int64_t i = 0; i < n; ++i) Cp[i] = flip(Cp[i]);
  for (int64_t j = 0; j <= n; ++j) head[j] = -1;
  for (int64_t j = n; j >= 0; --j) { if (nv[j] > 0) continue; next[j] = head[Cp[j]]; head[Cp[j]] = j; }                     // merged variables hang under their representative
  for (int64_t e = n; e >= 0; --e) { if (nv[e] <= 0) continue; if (Cp[e] != -1) { next[e] = head[Cp[e]]; head[Cp[e]] = e; } } // elements under their parent
  std::vector<int64_t> stack(n + 1, 0);
  int64_t k = 0;
  for (int64_t i = 0; i <= n; ++i) if (Cp[i] == -1) k = tdfs(i, k, head, next, P, stack);
  // drop the root placeholder n
  std::vector<int64_t> perm;
  perm.reserve(n);
  for (int64_t q = 0; q <= n && (int64_t)perm.size() < n; ++q) if (P[q] != n) perm.push_back(P[q]);
  return perm;
}

} // namespace gro
