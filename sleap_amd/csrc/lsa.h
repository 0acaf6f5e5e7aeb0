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

// ------------------------------------------------------------------------------------------------------------------------
// The same algorithm run by ONE WAVEFRONT (64 lanes) instead of one thread: the column scan of every augmenting-path step,
// the dual update and the initialisation are spread over the lanes; the (short) path walk stays scalar.
//
// SciPy's tie rule lives in the scan: sequentially, `index` ends up at the LAST scanned column that attains the minimum
// and is still unassigned, or -- if no minimal column is unassigned -- at the FIRST scanned column that attains it
// (`sp < lowest || (sp == lowest && row4col == -1)` walked over `remaining`). Each lane keeps that triple (min, first
// position, last free position) for the positions it owns; merging two triples is associative and commutative, so a
// butterfly reduction reproduces the sequential result exactly.
//
// Device: call with all 64 lanes of a wave converged; work arrays must be in LDS (DS operations of one wave are ordered).
// Host (tests): SA_LSA_HOST_LANES virtual lanes are looped, through the same per-lane and merge functions.
// ------------------------------------------------------------------------------------------------------------------------
struct ScanPartial {
  double m;    // smallest shortest-path cost seen
  int first;   // first position (in `remaining` order) attaining m
  int lfree;   // last position attaining m whose column is unassigned, or -1
};

SA_HD inline void scan_take(ScanPartial& p, double val, bool is_free, int it) {
  if (val < p.m) {
    p.m = val;
    p.first = it;
    p.lfree = is_free ? it : -1;
  } else if (val == p.m) {
    if (it < p.first) p.first = it;
    if (is_free && it > p.lfree) p.lfree = it;
  }
}

SA_HD inline ScanPartial scan_merge(const ScanPartial& a, const ScanPartial& b) {
  if (a.m < b.m) return a;
  if (b.m < a.m) return b;
  ScanPartial r;
  r.m = a.m;
  r.first = a.first < b.first ? a.first : b.first;
  r.lfree = a.lfree > b.lfree ? a.lfree : b.lfree;
  return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline ScanPartial scan_wave_reduce(ScanPartial p) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    ScanPartial q;
    q.m = __shfl_xor(p.m, off, 64);
    q.first = __shfl_xor(p.first, off, 64);
    q.lfree = __shfl_xor(p.lfree, off, 64);
    p = scan_merge(p, q);
  }
  return p;
}
// Lanes of a wave exchange data through the LDS work arrays: DS operations of one wave execute in order, but the COMPILER must
// be told that another lane may have written what this lane reads next (a wavefront-scope fence; it costs no instruction
// beyond the lgkmcnt wait the read needs anyway).
#define SA_WAVE_SYNC()                                        \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)
#else
#define SA_WAVE_SYNC() \
  do {                 \
  } while (0)
#endif
#define SA_LSA_LANES 64

template <typename CostFn>
SA_HD inline bool lsa_solve_wave(int nr, int nc, CostFn cost, LsaWork& w) {
  const double INF = __builtin_huge_val();
  constexpr int L = SA_LSA_LANES;
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane0 = (int)(threadIdx.x & 63), lane1 = lane0 + 1;  // this lane only
#else
  const int lane0 = 0, lane1 = L;  // every virtual lane in turn
#endif
  for (int lane = lane0; lane < lane1; ++lane) {
    for (int i = lane; i < nr; i += L) {
      w.u[i] = 0.0;
      w.col4row[i] = -1;
    }
    for (int j = lane; j < nc; j += L) {
      w.v[j] = 0.0;
      w.path[j] = -1;
      w.row4col[j] = -1;
    }
  }
  SA_WAVE_SYNC();
  for (int cur = 0; cur < nr; ++cur) {
    double minVal = 0.0;
    int num_remaining = nc;
    for (int lane = lane0; lane < lane1; ++lane) {
      for (int it = lane; it < nc; it += L) w.remaining[it] = nc - it - 1;
      for (int i = lane; i < nr; i += L) w.SR[i] = false;
      for (int j = lane; j < nc; j += L) {
        w.SC[j] = false;
        w.sp[j] = INF;
      }
    }
    SA_WAVE_SYNC();
    int sink = -1;
    int i = cur;
    while (sink == -1) {
      w.SR[i] = true;  // every lane writes the same value
      const double ui = w.u[i];
      ScanPartial tot = {INF, 0x7fffffff, -1};
      for (int lane = lane0; lane < lane1; ++lane) {
        ScanPartial p = {INF, 0x7fffffff, -1};
        for (int it = lane; it < num_remaining; it += L) {
          const int j = w.remaining[it];
          const double r = minVal + cost(i, j) - ui - w.v[j];
          if (r < w.sp[j]) {
            w.path[j] = i;
            w.sp[j] = r;
          }
          const double val = w.sp[j];
          const bool is_free = w.row4col[j] == -1;
          scan_take(p, val, is_free, it);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        tot = scan_wave_reduce(p);
#else
        tot = scan_merge(tot, p);
#endif
      }
      SA_WAVE_SYNC();
      minVal = tot.m;
      if (minVal == INF) return false;  // infeasible
      const int index = tot.lfree >= 0 ? tot.lfree : tot.first;
      const int j = w.remaining[index];
      const int r4c = w.row4col[j];
      if (r4c == -1)
        sink = j;
      else
        i = r4c;
      const int last = w.remaining[num_remaining - 1];
      --num_remaining;
      // single-writer updates (same values from every lane on the device)
      w.SC[j] = true;
      w.remaining[index] = last;
      SA_WAVE_SYNC();
    }
    // ---- dual update
    for (int lane = lane0; lane < lane1; ++lane) {
      for (int r = lane; r < nr; r += L)
        if (w.SR[r] && r != cur) w.u[r] += minVal - w.sp[w.col4row[r]];
      for (int j = lane; j < nc; j += L)
        if (w.SC[j]) w.v[j] -= minVal - w.sp[j];
    }
    SA_WAVE_SYNC();
    {  // u[cur] is touched by this statement only (r != cur above); every lane reads the old value, then writes the same sum
      const double ucur = w.u[cur];
      SA_WAVE_SYNC();
      w.u[cur] = ucur + minVal;
    }
    // ---- augment (scalar walk, the same on every lane; lane-identical single writes, every read precedes the writes of its
    // step on all lanes)
    int j = sink;
    while (true) {
      const int r = w.path[j];
      const int t = w.col4row[r];
      SA_WAVE_SYNC();
      w.row4col[j] = r;
      w.col4row[r] = j;
      SA_WAVE_SYNC();
      j = t;
      if (r == cur) break;
    }
  }
  return true;
}

}  // namespace sa
