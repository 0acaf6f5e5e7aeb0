// Rectangular linear sum assignment (shortest augmenting path, Crouse 2016), usable from host
// and device code.
//
// Replaces scipy.optimize.linear_sum_assignment as called by the reference at
// sleap/nn/utils.py:79-98 (through tf.numpy_function from paf_grouping.py:633). SciPy is a
// third-party dependency that is not vendored in the reference tree (pinned there as
// scipy >=1.4.1,<=1.9.0); this restates its published algorithm -- D. F. Crouse, "On implementing
// 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016 -- with the tie-breaking rules of
// SciPy >= 1.6 (columns scanned in reverse "remaining" order; among equal shortest paths prefer an
// unassigned column), so that results agree with SciPy also on ties. Checked against SciPy 1.7.1
// and 1.15.3 in tests/test_lsa.py.
#pragma once

#ifdef __HIPCC__
#define SA_HD __host__ __device__
#else
#define SA_HD
#endif

namespace sa {

// Work arrays live wherever the caller puts them (host heap, device global workspace): the problem size is a
// run-time value, the reference's ragged tensors have no upper bound on it.
struct LsaWork {
  double *u, *v, *sp;
  int *path, *col4row, *row4col, *remaining;
  unsigned char *SR, *SC;
  // bytes needed for an n x n (or smaller) problem, 8-byte aligned
  SA_HD static inline unsigned long bytes(int n) { return (unsigned long)n * (3 * 8 + 4 * 4 + 2) + 16; }
  SA_HD inline void bind(void* mem, int n) {
    unsigned char* p = (unsigned char*)mem;
    u = (double*)p;
    v = u + n;
    sp = v + n;
    path = (int*)(sp + n);
    col4row = path + n;
    row4col = col4row + n;
    remaining = row4col + n;
    SR = (unsigned char*)(remaining + n);
    SC = SR + n;
  }
};

// cost(i, j) accessor returns the (possibly transposed) cost as double.
// Solves for nr <= nc (caller transposes). Returns false if infeasible.
template <typename CostFn>
SA_HD inline bool lsa_solve(int nr, int nc, CostFn cost, LsaWork& w) {
  const double INF = __builtin_huge_val();
  for (int i = 0; i < nr; ++i) {
    w.u[i] = 0.0;
    w.col4row[i] = -1;
  }
  for (int j = 0; j < nc; ++j) {
    w.v[j] = 0.0;
    w.path[j] = -1;
    w.row4col[j] = -1;
  }
  for (int cur = 0; cur < nr; ++cur) {
    // ---- augmenting path from row `cur`
    double minVal = 0.0;
    int num_remaining = nc;
    for (int it = 0; it < nc; ++it) w.remaining[it] = nc - it - 1;
    for (int i = 0; i < nr; ++i) w.SR[i] = false;
    for (int j = 0; j < nc; ++j) {
      w.SC[j] = false;
      w.sp[j] = INF;
    }
    int sink = -1;
    int i = cur;
    while (sink == -1) {
      int index = -1;
      double lowest = INF;
      w.SR[i] = true;
      for (int it = 0; it < num_remaining; ++it) {
        const int j = w.remaining[it];
        const double r = minVal + cost(i, j) - w.u[i] - w.v[j];
        if (r < w.sp[j]) {
          w.path[j] = i;
          w.sp[j] = r;
        }
        if (w.sp[j] < lowest || (w.sp[j] == lowest && w.row4col[j] == -1)) {
          lowest = w.sp[j];
          index = it;
        }
      }
      minVal = lowest;
      if (minVal == INF) return false;  // infeasible
      const int j = w.remaining[index];
      if (w.row4col[j] == -1)
        sink = j;
      else
        i = w.row4col[j];
      w.SC[j] = true;
      w.remaining[index] = w.remaining[--num_remaining];
    }
    // ---- dual update
    w.u[cur] += minVal;
    for (int r = 0; r < nr; ++r)
      if (w.SR[r] && r != cur) w.u[r] += minVal - w.sp[w.col4row[r]];
    for (int j = 0; j < nc; ++j)
      if (w.SC[j]) w.v[j] -= minVal - w.sp[j];
    // ---- augment
    int j = sink;
    while (true) {
      const int r = w.path[j];
      w.row4col[j] = r;
      const int t = w.col4row[r];
      w.col4row[r] = j;
      j = t;
      if (r == cur) break;
    }
  }
  return true;
}

}  // namespace sa
