// Cross-frame identity tracker (host code, no GIL): the `simple` / `simplemaxtracks` candidate makers of
// sleap/nn/tracking.py:442-507, the `flow` / `flowmaxtracks` ones (:108-440: candidates are the queued instances SHIFTED into
// the current frame by sparse pyramidal Lucas-Kanade optical flow -- cv2.calcOpticalFlowPyrLK in the reference, the device
// kernels of csrc/flow.hip here, one launch per frame for the points of every queued frame), Tracker.track / spawn_for_untracked_instances (:642-814), FrameMatches
// (sleap/nn/tracker/components.py:469-640), the similarity functions (:33-196), greedy / Hungarian matching (:199-226),
// pre-cull (nms_fast / cull_frame_instances, :229-417) and connect_single_track_breaks (:419-466).
//
// All arithmetic is float64 like the reference (sleap.instance.Point stores x, y as f8); sums follow NumPy's reduction
// order (first element + pairwise sum of the rest) so that scores agree with the NumPy restatement to the last bits that
// exp() allows. Tracks are integers: index into the list of spawned tracks (reference name: f"track_{index}").
// This is a sequential per-video state machine: it runs on the host, after the device path, exactly where the reference
// runs it (inference.py:3306-3313); whole batches of frames go through one call.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <map>
#include <numeric>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sa_common.h"

extern "C" int sa_lsa_host(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind);

namespace {

const double NaN = std::numeric_limits<double>::quiet_NaN();
const double Inf = std::numeric_limits<double>::infinity();

struct Inst {
  std::vector<double> pts;     // [N][2], NaN = missing
  std::vector<double> scores;  // [N]
  double score = 0.0;
  int track = -1;
  int nvis = 0;
  int src = -1;       // index in the caller's frame
  long fserial = -1;  // serial number of the Tracker.track call (= frame) it was detected in
};

inline bool row_nan(const Inst& a, int k) { return std::isnan(a.pts[2 * k]) || std::isnan(a.pts[2 * k + 1]); }

// np.add.reduce over a contiguous 1-D double array: first element + pairwise_sum(rest) (numpy/core/src/umath/loops)
double np_pairwise(const double* a, long n) {
  if (n < 8) {
    double r = 0.0;
    for (long i = 0; i < n; ++i) r += a[i];
    return r;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  long n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}
double np_sum(const std::vector<double>& a) {
  if (a.empty()) return 0.0;
  return a[0] + np_pairwise(a.data() + 1, (long)a.size() - 1);
}
double np_nansum(std::vector<double> a) {
  for (double& v : a)
    if (std::isnan(v)) v = 0.0;
  return np_sum(a);
}

inline double pymax(double a, double b) { return (b > a) ? b : a; }  // Python's max(a, b) including its NaN behaviour
inline double pymin(double a, double b) { return (b < a) ? b : a; }

struct Config {
  int max_tracks_mode, similarity, match, track_window;
  double robust;
  int min_new_track_points, min_match_points, target_instance_count, pre_cull_to_target;
  double pre_cull_iou_threshold;
  int max_tracks, max_tracking;
  std::vector<double> kp_precision;  // 1 / (2 err^2); size 1 = scalar
  int oks_score_weighting, oks_normalization;
  int flow = 0, of_window_size = 21, of_max_levels = 3;  // FlowCandidateMaker (tracking.py:108-137)
  double img_scale = 1.0;                                // frames resized by cv2.resize before the flow (tracking.py:311-314)
  int save_shifted = 0;  // FlowCandidateMaker.save_shifted_instances: chain the flow through the latest shifted copy (:146-166)
};

// Device side of the flow candidate makers: one image pyramid per queued time step (the reference keeps `img_t` in every
// MatchedFrameInstances), the pyramid of the frame being tracked, and scratch for one Lucas-Kanade launch.
struct FlowState {
  int H = 0, W = 0;    // frame size
  int Hs = 0, Ws = 0;  // size of the pyramids' level 0: the frame resized by img_scale (== H, W without)
  std::map<int, void*> pyr;  // t -> device pyramid of that frame
  void* cur = nullptr;       // pyramid of the current frame (set by sa_tracker_set_image, consumed by the next track call)
  std::vector<void*> pool;   // free pyramid buffers of the current (H, W)
  void* scratch = nullptr;  // Lucas-Kanade job tables and results of one launch
  size_t scratch_bytes = 0;
  void* ptab = nullptr;     // the pyramid buffers of a run of frames (sa_flow_pyramid_build_batch reads them from the device)
  size_t ptab_bytes = 0;
  hipStream_t stream = nullptr;

  void* take(size_t bytes) {
    if (!pool.empty()) {
      void* b = pool.back();
      pool.pop_back();
      return b;
    }
    void* b = nullptr;
    return hipMalloc(&b, bytes) == hipSuccess ? b : nullptr;
  }
  void clear() {
    for (auto& kv : pyr) pool.push_back(kv.second);
    pyr.clear();
    if (cur) pool.push_back(cur);
    cur = nullptr;
  }
  void release_all() {
    clear();
    for (void* b : pool) (void)hipFree(b);
    pool.clear();
    if (scratch) (void)hipFree(scratch);
    scratch = nullptr;
    scratch_bytes = 0;
    if (ptab) (void)hipFree(ptab);
    ptab = nullptr;
    ptab_bytes = 0;
  }
};

struct Tracker {
  Config c;
  std::deque<std::pair<int, std::vector<Inst>>> queue;             // (t, tracked instances)
  std::vector<std::pair<int, std::deque<std::pair<int, Inst>>>> qdict;  // insertion-ordered {track: deque[(t, inst)]}
  int n_spawned = 0;
  long serial = 0;  // Tracker.track calls so far
  // FlowCandidateMaker.shifted_instances (tracking.py:136-138): (reference time step, time step shifted to) -> the shifted
  // instances; their image is the pyramid of the second time step
  std::map<std::pair<int, int>, std::vector<Inst>> saved;
  int last_first_choice = -1;  // FrameMatches.has_only_first_choice_matches of the last frame that had instances (-1: none yet)
  FlowState fs;
  // Shifts of a whole run of frames computed ahead (sa_tracker_track_frames_images, flow without max-tracks): the optical flow
  // of a queued instance into a later frame depends on the detections only, never on the track assignments, so all (queued
  // frame, target frame) pairs of the run go through ONE Lucas-Kanade launch and the frame-by-frame matching that follows only
  // looks results up: (frame serial, instance index, target frame of the run) -> first point of the instance.
  struct FlowBatch {
    bool active = false;
    long first_serial = 0;
    std::unordered_map<uint64_t, size_t> at;
    std::vector<float> shifted;
    std::vector<uint8_t> st;
    static uint64_t key(long fserial, int src, int f) { return ((uint64_t)(uint32_t)fserial << 32) | ((uint64_t)(uint16_t)src << 16) | (uint16_t)f; }
  } fb;
  ~Tracker() { fs.release_all(); }

  int find_track(int tr) const {
    for (size_t i = 0; i < qdict.size(); ++i)
      if (qdict[i].first == tr) return (int)i;
    return -1;
  }
};

// ---------------------------------------------------------------------------------------------- similarities
double sim_instance(const Inst& ref, const Inst& q, double nx, double ny) {
  const int N = (int)ref.scores.size();
  std::vector<double> e(N);
  int vis = 0;
  for (int k = 0; k < N; ++k) {
    if (!row_nan(ref, k)) ++vis;
    const double dx = q.pts[2 * k] / nx - ref.pts[2 * k] / nx, dy = q.pts[2 * k + 1] / ny - ref.pts[2 * k + 1] / ny;
    e[k] = std::exp(-(dx * dx + dy * dy));
  }
  return np_nansum(e) / (double)vis;  // 0 visible reference nodes -> 0/0 = NaN, as NumPy
}

double sim_oks(const Tracker& t, const Inst& ref, const Inst& q) {
  const Config& c = t.c;
  const int N = (int)ref.scores.size();
  int max_n = N;
  if (c.oks_normalization != 0) {
    max_n = 0;
    for (int k = 0; k < N; ++k) {
      const bool rv = !row_nan(ref, k);
      if (c.oks_normalization == 1 ? rv : (rv && !row_nan(q, k))) ++max_n;
    }
  }
  if (max_n == 0) return 0.0;
  std::vector<double> e(N);
  for (int k = 0; k < N; ++k) {
    double prec = c.kp_precision.size() == 1 ? c.kp_precision[0]
                  : c.kp_precision[std::min<size_t>(k, c.kp_precision.size() - 1)];  // truncate / pad with the last value
    const double dx = q.pts[2 * k] - ref.pts[2 * k], dy = q.pts[2 * k + 1] - ref.pts[2 * k + 1];
    const double d = (dx * dx + dy * dy) * prec;
    const double w = c.oks_score_weighting ? ref.scores[k] * q.scores[k] : 1.0;
    e[k] = w * std::exp(-d);
  }
  return np_nansum(e) / (double)max_n;
}

double nanmedian_axis(const Inst& a, int axis) {
  std::vector<double> v;
  for (size_t k = 0; k < a.scores.size(); ++k)
    if (!std::isnan(a.pts[2 * k + axis])) v.push_back(a.pts[2 * k + axis]);
  if (v.empty()) return NaN;
  std::sort(v.begin(), v.end());
  const size_t n = v.size();
  return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0;
}

double sim_centroid(const Inst& ref, const Inst& q) {
  const double dx = nanmedian_axis(ref, 0) - nanmedian_axis(q, 0), dy = nanmedian_axis(ref, 1) - nanmedian_axis(q, 1);
  return -std::sqrt(dx * dx + dy * dy);
}

void bbox(const Inst& a, double* b) {  // [y1, x1, y2, x2]
  double mn[2] = {Inf, Inf}, mx[2] = {-Inf, -Inf};
  bool any = false;
  for (size_t k = 0; k < a.scores.size(); ++k)
    for (int ax = 0; ax < 2; ++ax) {
      const double v = a.pts[2 * k + ax];
      if (!std::isnan(v)) {
        any = true;
        mn[ax] = std::min(mn[ax], v);
        mx[ax] = std::max(mx[ax], v);
      }
    }
  if (!any) {
    b[0] = b[1] = b[2] = b[3] = NaN;
    return;
  }
  for (int ax = 0; ax < 2; ++ax) {  // an axis that is NaN everywhere stays NaN (np.nanmin of an all-NaN slice)
    if (mn[ax] == Inf) mn[ax] = NaN;
    if (mx[ax] == -Inf) mx[ax] = NaN;
  }
  b[0] = mn[1];
  b[1] = mn[0];
  b[2] = mx[1];
  b[3] = mx[0];
}

double iou(const double* a, const double* b) {  // utils.py:45-76, Python max/min semantics
  const double iy1 = pymax(a[0], b[0]), ix1 = pymax(a[1], b[1]), iy2 = pymin(a[2], b[2]), ix2 = pymin(a[3], b[3]);
  const double inter = pymax(ix2 - ix1 + 1, 0) * pymax(iy2 - iy1 + 1, 0);
  const double a1 = (a[3] - a[1] + 1) * (a[2] - a[0] + 1), a2 = (b[3] - b[1] + 1) * (b[2] - b[0] + 1);
  return inter / (a1 + a2 - inter);
}

double similarity(const Tracker& t, const Inst& ref, const Inst& q, int img_h, int img_w) {
  switch (t.c.similarity) {
    case SA_SIM_INSTANCE: return sim_instance(ref, q, 1.0, 1.0);
    case SA_SIM_NORMALIZED_INSTANCE: return sim_instance(ref, q, (double)img_w, (double)img_h);
    case SA_SIM_CENTROID: return sim_centroid(ref, q);
    case SA_SIM_IOU: {
      double a[4], b[4];
      bbox(ref, a);
      bbox(q, b);
      return iou(a, b);
    }
    default: return sim_oks(t, ref, q);
  }
}

// np.quantile(v, q) (method "linear") / np.max with NaN propagation
double np_quantile(std::vector<double> v, double q) {
  for (double x : v)
    if (std::isnan(x)) return NaN;
  std::sort(v.begin(), v.end());
  const double idx = q * (double)(v.size() - 1);
  const double lo = std::floor(idx);
  const size_t i0 = (size_t)lo, i1 = std::min(i0 + 1, v.size() - 1);
  const double g = idx - lo, a = v[i0], b = v[i1];
  double r = a + (b - a) * g;  // numpy _lerp
  if (g >= 0.5) r = b - (b - a) * (1 - g);
  if (g == 0) r = a;  // (b - a) * 0 with infinities would give NaN; numpy's where(t == 0 ...) path is not needed for finite data
  return r;
}
double np_max(const std::vector<double>& v) {
  double m = v[0];
  for (double x : v) {
    if (std::isnan(x)) return NaN;
    if (x > m) m = x;
  }
  return m;
}

// ---------------------------------------------------------------------------------------------- pre-cull
std::vector<int> nms_fast(const std::vector<std::vector<double>>& boxes, const std::vector<double>& scores, double thr,
                          int target) {
  const int n = (int)boxes.size();
  std::vector<int> picked, nms;
  if (n == 0) return picked;
  if (target && n < target) {
    picked.resize(n);
    std::iota(picked.begin(), picked.end(), 0);
    return picked;
  }
  std::vector<double> area(n);
  for (int i = 0; i < n; ++i) area[i] = (boxes[i][2] - boxes[i][0] + 1) * (boxes[i][3] - boxes[i][1] + 1);
  std::vector<int> idxs(n);
  std::iota(idxs.begin(), idxs.end(), 0);
  std::stable_sort(idxs.begin(), idxs.end(), [&](int a, int b) { return scores[a] < scores[b]; });
  while (!idxs.empty()) {
    const int p = idxs.back();
    picked.push_back(p);
    std::vector<int> rest;
    for (size_t k = 0; k + 1 < idxs.size(); ++k) {
      const int i = idxs[k];
      const double xx1 = std::max(boxes[p][0], boxes[i][0]), yy1 = std::max(boxes[p][1], boxes[i][1]);
      const double xx2 = std::min(boxes[p][2], boxes[i][2]), yy2 = std::min(boxes[p][3], boxes[i][3]);
      const double w = std::max(0.0, xx2 - xx1 + 1), h = std::max(0.0, yy2 - yy1 + 1);
      if ((w * h) / area[i] > thr)
        nms.push_back(i);
      else
        rest.push_back(i);
    }
    idxs.swap(rest);
  }
  if (target && !nms.empty() && (int)picked.size() < target) {
    std::stable_sort(nms.begin(), nms.end(), [&](int a, int b) { return -scores[a] < -scores[b]; });
    long add_back = std::min<long>((long)nms.size(), (long)picked.size() - target);  // negative, as in the reference:
    if (add_back < 0) add_back = std::max<long>(0, (long)nms.size() + add_back);     // list[:negative] drops from the end
    for (long k = 0; k < add_back; ++k) picked.push_back(nms[k]);
  }
  return picked;
}

void cull_frame_instances(std::vector<Inst>& lst, int count, double iou_thr) {
  if (lst.empty() || (int)lst.size() <= count) return;
  std::vector<int> keep((int)lst.size());
  std::iota(keep.begin(), keep.end(), 0);
  std::vector<char> removed(lst.size(), 0);
  if (iou_thr > 0) {
    std::vector<std::vector<double>> boxes;
    std::vector<double> scores;
    for (const Inst& a : lst) {
      double b[4];
      bbox(a, b);
      boxes.push_back({b[0], b[1], b[2], b[3]});  // nms_fast reads columns 0..3 of the [y1,x1,y2,x2] box as "x1,y1,x2,y2"
      scores.push_back(a.score);
    }
    const std::vector<int> picks = nms_fast(boxes, scores, iou_thr, count);
    keep.clear();
    for (int i = 0; i < (int)lst.size(); ++i) {
      if (std::find(picks.begin(), picks.end(), i) != picks.end())
        keep.push_back(i);
      else
        removed[i] = 1;
    }
  }
  if ((int)keep.size() > count) {
    std::vector<int> order = keep;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lst[a].score < lst[b].score; });
    for (size_t k = 0; k + count < order.size(); ++k) removed[order[k]] = 1;
  }
  std::vector<Inst> out;
  for (size_t i = 0; i < lst.size(); ++i)
    if (!removed[i]) out.push_back(std::move(lst[i]));
  lst.swap(out);
}

// ---------------------------------------------------------------------------------------------- matching
int greedy_matching(const std::vector<double>& cost, int nr, int nc, std::vector<std::pair<int, int>>& out) {
  std::vector<int> order(nr * nc);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] < cost[b]; });
  std::vector<char> ru(nr, 0), cu(nc, 0);
  for (int e : order) {
    const int r = e / nc, c = e % nc;
    if (ru[r] || cu[c]) continue;
    ru[r] = cu[c] = 1;
    out.emplace_back(r, c);
  }
  return SA_OK;
}

int hungarian_matching(const std::vector<double>& cost, int nr, int nc, std::vector<std::pair<int, int>>& out) {
  std::vector<int64_t> ri(std::min(nr, nc)), ci(std::min(nr, nc));
  const int n = sa_lsa_host(cost.data(), nr, nc, ri.data(), ci.data());
  if (n < 0) return sa::fail(SA_ERR_INVALID_ARG, "tracker: cost matrix is infeasible");
  for (int i = 0; i < n; ++i) out.emplace_back((int)ri[i], (int)ci[i]);
  return SA_OK;
}

// FlowCandidateMaker.flow_shift_instances (tracking.py:258-356) for the instance lists of several reference frames at once:
// refs[g] = (t of the frame, its instances). ONE Lucas-Kanade launch for all points, then per reference instance (in order): a
// candidate when MORE than min_points of its points were found (`found.sum() > min_shifted_points`), lost points NaN, the
// reference instance's track. out[g] = the shifted instances of group g.
int flow_shift(Tracker& T, const std::vector<std::pair<int, std::vector<const Inst*>>>& refs, std::vector<std::vector<Inst>>& out) {
  FlowState& F = T.fs;
  const Config& c = T.c;
  out.assign(refs.size(), {});
  size_t n = 0;
  for (const auto& g : refs)
    for (const Inst* a : g.second) n += a->pts.size() / 2;
  if (n == 0) return SA_OK;
  if (!F.cur) return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: no image for the current frame (sa_tracker_set_image / img=...)");
  // scratch: [n] pyramid pointers | [n][2] points | [n][2] shifted | [n] err | [n] status
  const size_t need = n * (sizeof(void*) + 2 * sizeof(float) * 2 + sizeof(float) + 1) + 256;
  if (need > F.scratch_bytes) {
    if (F.scratch) (void)hipFree(F.scratch);
    F.scratch = nullptr;
    F.scratch_bytes = 0;
    SA_HIP_CHECK(hipMalloc(&F.scratch, need * 2));
    F.scratch_bytes = need * 2;
  }
  std::vector<const void*> ptrs(n);
  std::vector<float> pts(2 * n);
  const float fscale = (float)c.img_scale;  // (1 without: x * 1.0f and x / 1.0f are exact)
  size_t k = 0;
  for (const auto& g : refs) {
    const auto it = F.pyr.find(g.first);
    if (it == F.pyr.end() && !g.second.empty())
      return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: the frame of time step %d was tracked without an image", g.first);
    for (const Inst* a : g.second)
      for (size_t j = 0; j < a->pts.size() / 2; ++j, ++k) {
        ptrs[k] = it->second;
        pts[2 * k] = (float)a->pts[2 * j] * fscale;  // `.astype("float32") * scale`; NaN points stay NaN and come back "not found"
        pts[2 * k + 1] = (float)a->pts[2 * j + 1] * fscale;
      }
  }
  unsigned char* d = static_cast<unsigned char*>(F.scratch);
  void* d_ptrs = d;
  float* d_pts = reinterpret_cast<float*>(d + n * sizeof(void*));
  float* d_out = d_pts + 2 * n;
  float* d_err = d_out + 2 * n;
  uint8_t* d_st = reinterpret_cast<uint8_t*>(d_err + n);
  SA_HIP_CHECK(hipMemcpyAsync(d_ptrs, ptrs.data(), n * sizeof(void*), hipMemcpyHostToDevice, F.stream));
  SA_HIP_CHECK(hipMemcpyAsync(d_pts, pts.data(), 2 * n * sizeof(float), hipMemcpyHostToDevice, F.stream));
  const int rc = sa_flow_lk(static_cast<const void* const*>(d_ptrs), F.cur, F.Hs, F.Ws, c.of_window_size, c.of_max_levels, (int)n, d_pts,
                            d_out, d_st, d_err, 30, 0.01f, F.stream);
  if (rc != SA_OK) return rc;
  std::vector<float> shifted(2 * n);
  std::vector<uint8_t> st(n);
  SA_HIP_CHECK(hipMemcpyAsync(shifted.data(), d_out, 2 * n * sizeof(float), hipMemcpyDeviceToHost, F.stream));
  SA_HIP_CHECK(hipMemcpyAsync(st.data(), d_st, n, hipMemcpyDeviceToHost, F.stream));
  SA_HIP_CHECK(hipStreamSynchronize(F.stream));
  k = 0;
  for (size_t g = 0; g < refs.size(); ++g)
    for (const Inst* a : refs[g].second) {
      const size_t m = a->pts.size() / 2;
      int found = 0;
      for (size_t j = 0; j < m; ++j) found += st[k + j] ? 1 : 0;
      if (found > c.min_match_points) {
        Inst b;
        b.pts.resize(2 * m);
        b.scores = a->scores;
        b.score = a->score;
        b.track = a->track;
        b.src = a->src;
        for (size_t j = 0; j < m; ++j) {
          const bool ok = st[k + j] != 0;
          b.pts[2 * j] = ok ? (double)(shifted[2 * (k + j)] / fscale) : NaN;  // `shifted_pts /= scale` (float32)
          b.pts[2 * j + 1] = ok ? (double)(shifted[2 * (k + j) + 1] / fscale) : NaN;
          if (ok) ++b.nvis;
        }
        out[g].push_back(std::move(b));
      }
      k += m;
    }
  return SA_OK;
}

int track_one(Tracker& T, std::vector<Inst> untracked, int img_h, int img_w, int t, std::vector<Inst>& tracked,
              std::vector<double>& tscore) {
  const Config& c = T.c;
  const long my_serial = T.serial++;
  for (Inst& u : untracked) u.fserial = my_serial;
  if (t < 0) {
    t = 0;
    if (c.max_tracks_mode) {
      if (!T.qdict.empty()) {
        size_t best = 0;  // Python max(): first track with the most queued instances
        for (size_t i = 1; i < T.qdict.size(); ++i)
          if (T.qdict[i].second.size() > T.qdict[best].second.size()) best = i;
        // an empty deque can not occur: deques are created together with their first element
        t = T.qdict[best].second.back().first + 1;
      }
    } else if (!T.queue.empty()) {
      t = T.queue.back().first + 1;
    }
  }
  tracked.clear();
  tscore.clear();
  if (!untracked.empty()) {
    if (c.target_instance_count && c.pre_cull_to_target)
      cull_frame_instances(untracked, c.target_instance_count, c.pre_cull_iou_threshold);
    // candidates (SimpleCandidateMaker / SimpleMaxTracksCandidateMaker, or their optical-flow counterparts)
    std::vector<const Inst*> cand;
    std::vector<std::vector<Inst>> shifted;  // flow: owns the shifted instances `cand` points to
    if (c.flow) {
      // FlowCandidateMaker.get_candidates (tracking.py:210-237): every queued frame's instances shifted into this frame, oldest
      // frame first. FlowMaxTracksCandidateMaker.get_candidates (:1194-1240): per track (the first max_tracks when max_tracking)
      // and per queued item of that track, ALL queued instances of the item's time step are shifted -- an instance appears once
      // per track that holds an item of its time step, exactly as there.
      std::vector<std::pair<int, std::vector<const Inst*>>> groups;  // distinct reference time steps
      std::vector<size_t> order;                                      // candidate groups in the reference's order (indices into `groups`)
      std::vector<int> saved_ref_t;                                   // save_shifted: the queued frame every group stands for
      auto group_of = [&](int tr) {
        for (size_t g = 0; g < groups.size(); ++g)
          if (groups[g].first == tr) return g;
        groups.emplace_back(tr, std::vector<const Inst*>());
        return groups.size() - 1;
      };
      if (c.max_tracks_mode) {
        int n_tracks = 0;
        for (const auto& kv : T.qdict) {
          if (!c.max_tracking || n_tracks < c.max_tracks) {
            ++n_tracks;
            for (const auto& ti : kv.second) {
              const size_t before = groups.size();
              const size_t g = group_of(ti.first);
              if (groups.size() != before)  // get_ref_instances: every track's items of that time step, dictionary order
                for (const auto& kv2 : T.qdict)
                  for (const auto& tj : kv2.second)
                    if (tj.first == ti.first) groups[g].second.push_back(&tj.second);
              order.push_back(g);
            }
          }
        }
      } else if (c.save_shifted) {
        // prune_shifted_instances (:239-256), then per queued frame get_shifted_instances_from_earlier_time (:146-166): the latest
        // non-empty shifted copy of that frame's instances (and ITS image) is what gets shifted into the current frame
        for (auto it = T.saved.begin(); it != T.saved.end();)
          it = (t - it->first.first > c.track_window) ? T.saved.erase(it) : std::next(it);
        for (const auto& fr : T.queue) {
          int img_t = fr.first;
          const std::vector<Inst>* refs = &fr.second;
          for (int ti = t - 1; ti >= fr.first; --ti) {
            const auto it = T.saved.find({fr.first, ti});
            if (it != T.saved.end() && !it->second.empty()) {
              img_t = ti;
              refs = &it->second;
              break;
            }
          }
          if (refs->empty()) continue;
          groups.emplace_back(img_t, std::vector<const Inst*>());  // one group per queued frame (two may share an image)
          for (const Inst& a : *refs) groups.back().second.push_back(&a);
          order.push_back(groups.size() - 1);
          saved_ref_t.push_back(fr.first);
        }
      } else {
        for (const auto& fr : T.queue) {
          if (fr.second.empty()) continue;
          const size_t g = group_of(fr.first);
          for (const Inst& a : fr.second) groups[g].second.push_back(&a);
          order.push_back(g);
        }
      }
      if (T.fb.active) {
        // the shifts were computed for the whole run of frames (see FlowBatch): look them up. A group none of whose shifts were
        // computed ahead (max-tracks mode: a track that went unmatched keeps items older than the run looked back) is shifted now.
        const int f = (int)(my_serial - T.fb.first_serial);
        shifted.assign(groups.size(), {});
        std::vector<std::pair<int, std::vector<const Inst*>>> miss;
        std::vector<size_t> miss_g;
        for (size_t g = 0; g < groups.size(); ++g) {
          bool all = true;
          for (const Inst* a : groups[g].second) all = all && T.fb.at.count(Tracker::FlowBatch::key(a->fserial, a->src, f)) != 0;
          if (!all) {
            miss.push_back(groups[g]);
            miss_g.push_back(g);
            continue;
          }
          for (const Inst* a : groups[g].second) {
            const size_t k = T.fb.at.find(Tracker::FlowBatch::key(a->fserial, a->src, f))->second, m = a->pts.size() / 2;
            int found = 0;
            for (size_t j = 0; j < m; ++j) found += T.fb.st[k + j] ? 1 : 0;
            if (found <= c.min_match_points) continue;
            Inst b;
            b.pts.resize(2 * m);
            b.scores = a->scores;
            b.score = a->score;
            b.track = a->track;
            b.src = a->src;
            for (size_t j = 0; j < m; ++j) {
              const bool ok = T.fb.st[k + j] != 0;
              b.pts[2 * j] = ok ? (double)(T.fb.shifted[2 * (k + j)] / (float)c.img_scale) : NaN;
              b.pts[2 * j + 1] = ok ? (double)(T.fb.shifted[2 * (k + j) + 1] / (float)c.img_scale) : NaN;
              if (ok) ++b.nvis;
            }
            shifted[g].push_back(std::move(b));
          }
        }
        if (!miss.empty()) {
          std::vector<std::vector<Inst>> late;
          const int rc = flow_shift(T, miss, late);
          if (rc != SA_OK) return rc;
          for (size_t i = 0; i < miss_g.size(); ++i) shifted[miss_g[i]] = std::move(late[i]);
        }
      } else {
        const int rc = flow_shift(T, groups, shifted);
        if (rc != SA_OK) return rc;
      }
      if (c.save_shifted && !c.max_tracks_mode)  // get_shifted_instances (:197-206): kept for the frames to come
        for (size_t g = 0; g < groups.size(); ++g) T.saved[{saved_ref_t[g], t}] = shifted[g];
      for (size_t g : order)
        for (const Inst& a : shifted[g]) cand.push_back(&a);
    } else if (c.max_tracks_mode) {
      int n_tracks = 0;
      for (const auto& kv : T.qdict) {
        if (!c.max_tracking || n_tracks < c.max_tracks) {
          ++n_tracks;
          for (const auto& ti : kv.second)
            if (ti.second.nvis >= c.min_match_points) cand.push_back(&ti.second);
        }
      }
    } else {
      for (const auto& fr : T.queue)
        for (const Inst& a : fr.second)
          if (a.nvis >= c.min_match_points) cand.push_back(&a);
    }
    std::vector<int> tracks;  // candidate tracks in order of first appearance
    std::vector<std::vector<const Inst*>> by_track;
    for (const Inst* a : cand) {
      size_t j = 0;
      for (; j < tracks.size(); ++j)
        if (tracks[j] == a->track) break;
      if (j == tracks.size()) {
        tracks.push_back(a->track);
        by_track.emplace_back();
      }
      by_track[j].push_back(a);
    }
    const int nr = (int)untracked.size(), nc = (int)tracks.size();
    std::vector<double> cost((size_t)nr * nc);
    std::vector<std::pair<int, int>> matches;
    if (nc > 0) {
      for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) {
          std::vector<double> vals;
          for (const Inst* q : by_track[j]) vals.push_back(similarity(T, untracked[i], *q, img_h, img_w));
          const double best = (c.robust > 0 && c.robust < 1) ? np_quantile(vals, c.robust) : np_max(vals);
          cost[(size_t)i * nc + j] = std::isnan(best) ? Inf : -best;
        }
      const int rc = c.match == SA_MATCH_HUNGARIAN ? hungarian_matching(cost, nr, nc, matches)
                                                    : greedy_matching(cost, nr, nc, matches);
      if (rc != SA_OK) return rc;
    }
    {  // Tracker.last_matches.has_only_first_choice_matches (components.py:478-480, 593-607): every match is its row's argmin
      bool all_first = true;
      for (const auto& m : matches) {
        int best = 0;
        for (int j = 1; j < nc; ++j)
          if (cost[(size_t)m.first * nc + j] < cost[(size_t)m.first * nc + best]) best = j;
        all_first = all_first && best == m.second;
      }
      T.last_first_choice = all_first ? 1 : 0;
    }
    std::vector<char> matched(nr, 0);
    for (const auto& m : matches) {
      matched[m.first] = 1;
      Inst a = untracked[m.first];
      a.track = tracks[m.second];
      tracked.push_back(std::move(a));
      tscore.push_back(-cost[(size_t)m.first * nc + m.second]);
    }
    for (int i = 0; i < nr; ++i) {  // spawn_for_untracked_instances
      if (matched[i]) continue;
      if (untracked[i].nvis < c.min_new_track_points) continue;
      if (c.max_tracks_mode && c.max_tracking && (int)T.qdict.size() >= c.max_tracks) break;
      Inst a = untracked[i];
      a.track = T.n_spawned++;
      tracked.push_back(std::move(a));
      tscore.push_back(0.0);
    }
  }
  if (c.max_tracks_mode) {
    for (const Inst& a : tracked) {
      int k = T.find_track(a.track);
      if (k < 0 && (!c.max_tracking || (int)T.qdict.size() < c.max_tracks)) {
        T.qdict.emplace_back(a.track, std::deque<std::pair<int, Inst>>());
        k = (int)T.qdict.size() - 1;
      }
      if (k >= 0) {
        auto& dq = T.qdict[k].second;
        dq.emplace_back(t, a);
        while ((int)dq.size() > c.track_window) dq.pop_front();
      }
    }
  } else {
    T.queue.emplace_back(t, tracked);
    while ((int)T.queue.size() > c.track_window) T.queue.pop_front();
  }
  if (c.flow) {
    // the frame's pyramid is kept while a queued item refers to its time step (MatchedFrameInstance(s).img_t)
    FlowState& F = T.fs;
    if (F.cur) {
      auto it = F.pyr.find(t);
      if (it != F.pyr.end()) F.pool.push_back(it->second);
      F.pyr[t] = F.cur;
      F.cur = nullptr;
    }
    for (auto it = F.pyr.begin(); it != F.pyr.end();) {
      bool used = false;
      if (c.max_tracks_mode) {
        for (const auto& kv : T.qdict)
          for (const auto& ti : kv.second) used = used || ti.first == it->first;
      } else {
        for (const auto& fr : T.queue) used = used || (fr.first == it->first && !fr.second.empty());
        for (const auto& kv : T.saved) used = used || kv.first.second == it->first;  // the image of a saved shifted copy
      }
      if (used) {
        ++it;
      } else {
        F.pool.push_back(it->second);
        it = F.pyr.erase(it);
      }
    }
  }
  return SA_OK;
}

Inst make_inst(const float* pts, const float* ps, float score, int N, int src) {
  Inst a;
  a.pts.resize(2 * N);
  a.scores.resize(N);
  for (int k = 0; k < N; ++k) {
    const bool nan = std::isnan(pts[2 * k]) || std::isnan(pts[2 * k + 1]);  // from_arrays skips such nodes entirely
    a.pts[2 * k] = nan ? NaN : (double)pts[2 * k];
    a.pts[2 * k + 1] = nan ? NaN : (double)pts[2 * k + 1];
    a.scores[k] = ps ? (double)ps[k] : 1.0;
    if (!nan) ++a.nvis;
  }
  a.score = score;
  a.src = src;
  return a;
}

// One run of frames of a flow tracker without max-tracks: pyramids of all frames, one Lucas-Kanade launch for every (queued frame,
// target frame) pair -- the queue a frame will see is known in advance: the last track_window entries of (what is queued now,
// then the earlier frames of the run), and a frame of the run can only queue instances that are among its detections -- then the
// per-frame matching (track_one) with the shifts looked up.
int flow_batch_run(Tracker& T, int n_frames, int max_inst, int n_nodes, const float* points, const float* point_scores,
                   const float* inst_scores, const int* n_valid, int img_h, int img_w, int t0, const int* frame_t,
                   const uint8_t* images, int frame_h, int frame_w, int C, sa_stream_t stream, int* out_track, double* out_score,
                   int* out_order) {
  FlowState& F = T.fs;
  const Config& c = T.c;
  F.stream = (hipStream_t)stream;
  if (F.H != frame_h || F.W != frame_w) {
    if (!F.pyr.empty())
      return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: frame size changed from %dx%d to %dx%d with frames still queued (reset first)",
                      F.H, F.W, frame_h, frame_w);
    F.release_all();
    F.H = frame_h;
    F.W = frame_w;
    const int rs = sa_flow_scaled_size(frame_h, frame_w, c.img_scale, &F.Hs, &F.Ws);
    if (rs != SA_OK) return rs;
  }
  if (F.cur) {  // an image handed over by sa_tracker_set_image and never used
    F.pool.push_back(F.cur);
    F.cur = nullptr;
  }
  const size_t pbytes = sa_flow_pyramid_bytes(F.Hs, F.Ws, c.of_window_size, c.of_max_levels);
  std::vector<void*> bp((size_t)n_frames, nullptr);
  struct PoolGuard {  // every exit path hands the pyramid buffers it did not consume back to the tracker's pool
    std::vector<void*>& v;
    std::vector<void*>& pool;
    ~PoolGuard() {
      for (void*& b : v)
        if (b) pool.push_back(b), b = nullptr;
    }
  } bp_guard{bp, F.pool};
  for (int f = 0; f < n_frames; ++f) {
    SA_REQUIRE(n_valid[f] >= 0 && n_valid[f] <= max_inst && max_inst < 65536, "sa_tracker_track_frames_images: n_valid[%d] out of range", f);
    bp[(size_t)f] = F.take(pbytes);
    if (!bp[(size_t)f]) return sa::fail(SA_ERR_HIP, "flow tracker: out of device memory for the frame pyramids");
  }
  {  // all pyramids of the run in a handful of launches; no synchronisation here -- the Lucas-Kanade job tables below are built
     // on the host while these run (the table of buffer pointers has a device area of its own)
    const size_t need = (size_t)n_frames * sizeof(void*);
    if (need > F.ptab_bytes) {
      if (F.ptab) (void)hipFree(F.ptab);  // (hipFree waits for the device: nothing still reads the old table)
      F.ptab = nullptr;
      F.ptab_bytes = 0;
      SA_HIP_CHECK(hipMalloc(&F.ptab, need * 2));
      F.ptab_bytes = need * 2;
    }
    SA_HIP_CHECK(hipMemcpyAsync(F.ptab, bp.data(), need, hipMemcpyHostToDevice, F.stream));
    const int rc = c.img_scale != 1.0
                       ? sa_flow_pyramid_build_scaled(images, n_frames, frame_h, frame_w, C, c.img_scale, c.of_window_size,
                                                      c.of_max_levels, nullptr, static_cast<void* const*>(F.ptab), stream)
                       : sa_flow_pyramid_build_batch(images, n_frames, frame_h, frame_w, C, c.of_window_size, c.of_max_levels,
                                                     static_cast<void* const*>(F.ptab), stream);
    if (rc != SA_OK) return rc;
  }
  // ---- jobs
  const int Wq = c.track_window, q0 = (int)T.queue.size();
  std::vector<const void*> prev, next;
  std::vector<float> pts;
  Tracker::FlowBatch& B = T.fb;
  B.at.clear();
  B.first_serial = T.serial;
  const float fscale = (float)c.img_scale;
  auto add_point = [&](const void* pp, const void* pn, float x, float y) {
    prev.push_back(pp);
    next.push_back(pn);
    pts.push_back(x * fscale);
    pts.push_back(y * fscale);
  };
  for (int f = 0; f < n_frames; ++f) {
    const int nb = f < Wq ? f : Wq;
    if (c.max_tracks_mode) {
      // per-track queues: which earlier frames a track still holds when frame f arrives depends on the matching; computed ahead
      // are the likely ones -- everything queued now, into the first track_window frames of the run, and the detections of the
      // track_window frames before f -- anything else is shifted on demand (track_one)
      if (f < Wq)
        for (const auto& kv : T.qdict)
          for (const auto& ti : kv.second) {
            const auto it = F.pyr.find(ti.first);
            if (it == F.pyr.end()) return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: the frame of time step %d was tracked without an image", ti.first);
            const Inst& a = ti.second;
            B.at[Tracker::FlowBatch::key(a.fserial, a.src, f)] = prev.size();
            for (size_t j = 0; j < a.pts.size() / 2; ++j) add_point(it->second, bp[(size_t)f], (float)a.pts[2 * j], (float)a.pts[2 * j + 1]);
          }
    }
    const int npre = c.max_tracks_mode ? 0 : ((Wq - nb) < q0 ? (Wq - nb) : q0);
    for (int e = q0 - npre; e < q0; ++e) {
      const auto& fr = T.queue[(size_t)e];
      if (fr.second.empty()) continue;
      const auto it = F.pyr.find(fr.first);
      if (it == F.pyr.end()) return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: the frame of time step %d was tracked without an image", fr.first);
      for (const Inst& a : fr.second) {
        B.at[Tracker::FlowBatch::key(a.fserial, a.src, f)] = prev.size();
        for (size_t j = 0; j < a.pts.size() / 2; ++j) add_point(it->second, bp[(size_t)f], (float)a.pts[2 * j], (float)a.pts[2 * j + 1]);
      }
    }
    for (int fbk = f - nb; fbk < f; ++fbk)
      for (int i = 0; i < n_valid[fbk]; ++i) {
        B.at[Tracker::FlowBatch::key(B.first_serial + fbk, i, f)] = prev.size();
        const float* q = points + ((size_t)fbk * max_inst + i) * n_nodes * 2;
        for (int j = 0; j < n_nodes; ++j) {
          const bool nan = std::isnan(q[2 * j]) || std::isnan(q[2 * j + 1]);  // make_inst: such nodes are NaN in both coordinates
          add_point(bp[(size_t)fbk], bp[(size_t)f], nan ? std::numeric_limits<float>::quiet_NaN() : q[2 * j],
                    nan ? std::numeric_limits<float>::quiet_NaN() : q[2 * j + 1]);
        }
      }
  }
  const size_t n = prev.size();
  B.shifted.assign(2 * n, 0.0f);
  B.st.assign(n, 0);
  if (n) {
    const size_t need = n * (2 * sizeof(void*) + 2 * sizeof(float) * 2 + sizeof(float) + 1) + 256;
    if (need > F.scratch_bytes) {
      if (F.scratch) (void)hipFree(F.scratch);
      F.scratch = nullptr;
      F.scratch_bytes = 0;
      SA_HIP_CHECK(hipMalloc(&F.scratch, need * 2));
      F.scratch_bytes = need * 2;
    }
    unsigned char* d = static_cast<unsigned char*>(F.scratch);
    void* d_prev = d;
    void* d_next = d + n * sizeof(void*);
    float* d_pts = reinterpret_cast<float*>(d + 2 * n * sizeof(void*));
    float* d_out = d_pts + 2 * n;
    float* d_err = d_out + 2 * n;
    uint8_t* d_st = reinterpret_cast<uint8_t*>(d_err + n);
    SA_HIP_CHECK(hipMemcpyAsync(d_prev, prev.data(), n * sizeof(void*), hipMemcpyHostToDevice, F.stream));
    SA_HIP_CHECK(hipMemcpyAsync(d_next, next.data(), n * sizeof(void*), hipMemcpyHostToDevice, F.stream));
    SA_HIP_CHECK(hipMemcpyAsync(d_pts, pts.data(), 2 * n * sizeof(float), hipMemcpyHostToDevice, F.stream));
    const int rc = sa_flow_lk_pairs(static_cast<const void* const*>(d_prev), static_cast<const void* const*>(d_next), F.Hs, F.Ws,
                                    c.of_window_size, c.of_max_levels, (int)n, d_pts, d_out, d_st, d_err, 30, 0.01f, F.stream);
    if (rc != SA_OK) return rc;
    SA_HIP_CHECK(hipMemcpyAsync(B.shifted.data(), d_out, 2 * n * sizeof(float), hipMemcpyDeviceToHost, F.stream));
    SA_HIP_CHECK(hipMemcpyAsync(B.st.data(), d_st, n, hipMemcpyDeviceToHost, F.stream));
  }
  SA_HIP_CHECK(hipStreamSynchronize(F.stream));
  // ---- frame-by-frame matching on the looked-up shifts
  B.active = true;
  int rc = SA_OK;
  std::vector<Inst> tracked;
  std::vector<double> ts;
  for (int f = 0; f < n_frames && rc == SA_OK; ++f) {
    F.cur = bp[(size_t)f];
    bp[(size_t)f] = nullptr;
    std::vector<Inst> untracked;
    for (int i = 0; i < n_valid[f]; ++i) {
      const size_t o = (size_t)f * max_inst + i;
      untracked.push_back(make_inst(points + o * n_nodes * 2, point_scores ? point_scores + o * n_nodes : nullptr,
                                    inst_scores ? inst_scores[o] : 0.0f, n_nodes, i));
    }
    for (int i = 0; i < max_inst; ++i) {
      const size_t o = (size_t)f * max_inst + i;
      out_track[o] = -1;
      if (out_score) out_score[o] = NaN;
      if (out_order) out_order[o] = -1;
    }
    rc = track_one(T, std::move(untracked), img_h, img_w, frame_t ? frame_t[f] : (t0 < 0 ? -1 : t0 + f), tracked, ts);
    if (rc != SA_OK) break;
    for (size_t k = 0; k < tracked.size(); ++k) {
      const size_t o = (size_t)f * max_inst + tracked[k].src;
      out_track[o] = tracked[k].track;
      if (out_score) out_score[o] = ts[k];
      if (out_order) out_order[o] = (int)k;
    }
  }
  B.active = false;
  B.at.clear();
  return rc;  // (bp_guard returns what is left of bp)
}

}  // namespace

extern "C" {

void* sa_tracker_create(const sa_tracker_config* cfg) {
  if (!cfg || cfg->track_window <= 0 || cfg->similarity < 0 || cfg->similarity > SA_SIM_OBJECT_KEYPOINT ||
      (cfg->match != SA_MATCH_GREEDY && cfg->match != SA_MATCH_HUNGARIAN) ||
      (cfg->max_tracks_mode && cfg->max_tracking && cfg->max_tracks <= 0)) {
    sa::fail(SA_ERR_INVALID_ARG, "sa_tracker_create: invalid configuration");
    return nullptr;
  }
  Tracker* t = new Tracker();
  Config& c = t->c;
  c.max_tracks_mode = cfg->max_tracks_mode;
  c.similarity = cfg->similarity;
  c.match = cfg->match;
  c.track_window = cfg->track_window;
  c.robust = cfg->robust;
  c.min_new_track_points = cfg->min_new_track_points;
  c.min_match_points = cfg->min_match_points;
  c.target_instance_count = cfg->target_instance_count;
  c.pre_cull_to_target = cfg->pre_cull_to_target;
  c.pre_cull_iou_threshold = cfg->pre_cull_iou_threshold;
  c.max_tracks = cfg->max_tracks;
  c.max_tracking = cfg->max_tracks > 0 ? cfg->max_tracking : 0;
  c.oks_score_weighting = cfg->oks_score_weighting;
  c.oks_normalization = cfg->oks_normalization;
  c.flow = cfg->flow ? 1 : 0;
  c.of_window_size = cfg->of_window_size > 0 ? cfg->of_window_size : 21;
  c.of_max_levels = cfg->of_max_levels >= 0 ? cfg->of_max_levels : 3;
  c.img_scale = cfg->img_scale > 0.0 ? cfg->img_scale : 1.0;
  c.save_shifted = cfg->save_shifted_instances != 0 && c.flow && !c.max_tracks_mode;  // (the reference configures it for "flow" only)
  if (c.flow && (c.of_window_size < 3 || c.of_window_size > 31)) {
    delete t;
    sa::fail(SA_ERR_UNSUPPORTED, "sa_tracker_create: of_window_size must be in 3..31");
    return nullptr;
  }
  if (cfg->oks_n_errors > 0 && cfg->oks_errors) {
    for (int i = 0; i < cfg->oks_n_errors; ++i) c.kp_precision.push_back(1.0 / (2.0 * cfg->oks_errors[i] * cfg->oks_errors[i]));
  } else {
    c.kp_precision.push_back(0.5);  // keypoint_errors = 1
  }
  return t;
}

void sa_tracker_destroy(void* h) { delete static_cast<Tracker*>(h); }

int sa_tracker_n_tracks(void* h) { return h ? static_cast<Tracker*>(h)->n_spawned : -1; }

int sa_tracker_last_first_choice(void* h) { return h ? static_cast<Tracker*>(h)->last_first_choice : -1; }

int sa_tracker_reset(void* h) {
  SA_REQUIRE(h, "sa_tracker_reset: NULL handle");
  Tracker* T = static_cast<Tracker*>(h);
  // Tracker.reset_candidates (tracking.py:615-620): queues are emptied, spawned tracks are kept
  T->queue.clear();
  T->saved.clear();
  for (auto& kv : T->qdict) kv.second.clear();
  T->fs.clear();
  return SA_OK;
}

int sa_tracker_set_image(void* h, const void* image, int H, int W, int C, sa_stream_t stream) {
  SA_REQUIRE(h && image && H > 0 && W > 0 && (C == 1 || C == 3), "sa_tracker_set_image: bad arguments");
  Tracker* T = static_cast<Tracker*>(h);
  if (!T->c.flow) return SA_OK;  // the simple candidate makers do not look at frames (uses_image == False)
  FlowState& F = T->fs;
  F.stream = (hipStream_t)stream;
  if (F.H != H || F.W != W) {
    if (!F.pyr.empty())
      return sa::fail(SA_ERR_INVALID_ARG, "flow tracker: frame size changed from %dx%d to %dx%d with frames still queued (reset first)",
                      F.H, F.W, H, W);
    F.release_all();
    F.H = H;
    F.W = W;
    const int rs = sa_flow_scaled_size(H, W, T->c.img_scale, &F.Hs, &F.Ws);
    if (rs != SA_OK) return rs;
  }
  if (!F.cur) F.cur = F.take(sa_flow_pyramid_bytes(F.Hs, F.Ws, T->c.of_window_size, T->c.of_max_levels));
  if (!F.cur) return sa::fail(SA_ERR_HIP, "flow tracker: out of device memory for the frame pyramid");
  if (T->c.img_scale != 1.0)
    return sa_flow_pyramid_build_scaled(image, 1, H, W, C, T->c.img_scale, T->c.of_window_size, T->c.of_max_levels, F.cur, nullptr, stream);
  return sa_flow_pyramid_build(image, H, W, C, T->c.of_window_size, T->c.of_max_levels, F.cur, stream);
}

int sa_tracker_track(void* h, int n, int n_nodes, const float* points, const float* point_scores, const float* inst_scores,
                     int img_h, int img_w, int t, int* out_index, int* out_track, double* out_score, int* n_out) {
  SA_REQUIRE(h && n >= 0 && n_nodes > 0 && (n == 0 || points) && n_out, "sa_tracker_track: bad arguments");
  Tracker* T = static_cast<Tracker*>(h);
  std::vector<Inst> untracked;
  for (int i = 0; i < n; ++i)
    untracked.push_back(make_inst(points + (size_t)i * n_nodes * 2, point_scores ? point_scores + (size_t)i * n_nodes : nullptr,
                                  inst_scores ? inst_scores[i] : 0.0f, n_nodes, i));
  std::vector<Inst> tracked;
  std::vector<double> ts;
  const int rc = track_one(*T, std::move(untracked), img_h, img_w, t, tracked, ts);
  if (rc != SA_OK) return rc;
  *n_out = (int)tracked.size();
  for (size_t k = 0; k < tracked.size(); ++k) {
    if (out_index) out_index[k] = tracked[k].src;
    if (out_track) out_track[k] = tracked[k].track;
    if (out_score) out_score[k] = ts[k];
  }
  return SA_OK;
}

int sa_tracker_track_frames(void* h, int n_frames, int max_inst, int n_nodes, const float* points, const float* point_scores,
                            const float* inst_scores, const int* n_valid, int img_h, int img_w, int t0, int* out_track,
                            double* out_score, int* out_order) {
  SA_REQUIRE(h && n_frames >= 0 && max_inst >= 0 && n_nodes > 0 && n_valid && out_track, "sa_tracker_track_frames: bad arguments");
  Tracker* T = static_cast<Tracker*>(h);
  std::vector<Inst> tracked;
  std::vector<double> ts;
  for (int f = 0; f < n_frames; ++f) {
    const int n = n_valid[f];
    SA_REQUIRE(n >= 0 && n <= max_inst, "sa_tracker_track_frames: n_valid[%d] = %d out of range", f, n);
    std::vector<Inst> untracked;
    for (int i = 0; i < n; ++i) {
      const size_t o = (size_t)f * max_inst + i;
      untracked.push_back(make_inst(points + o * n_nodes * 2, point_scores ? point_scores + o * n_nodes : nullptr,
                                    inst_scores ? inst_scores[o] : 0.0f, n_nodes, i));
    }
    for (int i = 0; i < max_inst; ++i) {
      const size_t o = (size_t)f * max_inst + i;
      out_track[o] = -1;
      if (out_score) out_score[o] = NaN;
      if (out_order) out_order[o] = -1;
    }
    const int rc = track_one(*T, std::move(untracked), img_h, img_w, t0 < 0 ? -1 : t0 + f, tracked, ts);
    if (rc != SA_OK) return rc;
    for (size_t k = 0; k < tracked.size(); ++k) {
      const size_t o = (size_t)f * max_inst + tracked[k].src;
      out_track[o] = tracked[k].track;
      if (out_score) out_score[o] = ts[k];
      if (out_order) out_order[o] = (int)k;
    }
  }
  return SA_OK;
}

int sa_tracker_track_frames_images(void* h, int n_frames, int max_inst, int n_nodes, const float* points, const float* point_scores,
                                   const float* inst_scores, const int* n_valid, int img_h, int img_w, int t0,
                                   const int* frame_t, const void* images, int frame_h, int frame_w, int C, sa_stream_t stream,
                                   int* out_track, double* out_score, int* out_order) {
  SA_REQUIRE(h && n_frames >= 0 && max_inst >= 0 && n_nodes > 0 && n_valid && out_track, "sa_tracker_track_frames_images: bad arguments");
  Tracker* T = static_cast<Tracker*>(h);
  SA_REQUIRE(!T->c.flow || images, "sa_tracker_track_frames_images: a flow tracker needs the frames");
  const size_t stride_f = (size_t)max_inst;
  if (T->c.flow && n_frames > 1 && !T->c.save_shifted) {  // (chained flow depends on the earlier frames' results: frame by frame)
    // ---- the flow of every (queued frame, target frame) pair of this run in ONE Lucas-Kanade launch (FlowBatch), in runs of
    // <= 128 frames (a 1024 x 1024 pyramid is 6.7 MB)
    constexpr int RUN = 128;
    for (int f0 = 0; f0 < n_frames; f0 += RUN) {
      const int nf = n_frames - f0 < RUN ? n_frames - f0 : RUN;
      const int rc = flow_batch_run(*T, nf, max_inst, n_nodes, points + (size_t)f0 * stride_f * n_nodes * 2,
                                    point_scores ? point_scores + (size_t)f0 * stride_f * n_nodes : nullptr,
                                    inst_scores ? inst_scores + (size_t)f0 * stride_f : nullptr, n_valid + f0, img_h, img_w,
                                    t0 < 0 ? -1 : t0 + f0, frame_t ? frame_t + f0 : nullptr,
                                    static_cast<const uint8_t*>(images) + (size_t)f0 * frame_h * frame_w * C, frame_h, frame_w, C, stream,
                                    out_track + (size_t)f0 * stride_f, out_score ? out_score + (size_t)f0 * stride_f : nullptr,
                                    out_order ? out_order + (size_t)f0 * stride_f : nullptr);
      if (rc != SA_OK) return rc;
    }
    return SA_OK;
  }
  for (int f = 0; f < n_frames; ++f) {
    if (T->c.flow) {
      const int rc = sa_tracker_set_image(h, static_cast<const uint8_t*>(images) + (size_t)f * frame_h * frame_w * C, frame_h, frame_w, C,
                                          stream);
      if (rc != SA_OK) return rc;
    }
    const int t = frame_t ? frame_t[f] : (t0 < 0 ? -1 : t0 + f);
    const int rc = sa_tracker_track_frames(h, 1, max_inst, n_nodes, points + (size_t)f * stride_f * n_nodes * 2,
                                           point_scores ? point_scores + (size_t)f * stride_f * n_nodes : nullptr,
                                           inst_scores ? inst_scores + (size_t)f * stride_f : nullptr, n_valid + f, img_h, img_w, t,
                                           out_track + (size_t)f * stride_f, out_score ? out_score + (size_t)f * stride_f : nullptr,
                                           out_order ? out_order + (size_t)f * stride_f : nullptr);
    if (rc != SA_OK) return rc;
  }
  return SA_OK;
}

int sa_connect_single_track_breaks(int n_frames, int max_inst, const int* order, int* track, int instance_count) {
  // components.py:419-466 on the [F, I] track table of sa_tracker_track_frames (-1 = no instance); `order` gives the
  // position of each instance in its frame's list (the reference iterates lf.instances in list order), or NULL = slot order
  SA_REQUIRE(n_frames >= 0 && max_inst >= 0 && track, "sa_connect_single_track_breaks: bad arguments");
  if (n_frames == 0) return SA_OK;
  auto frame_list = [&](int f) {
    std::vector<int> idx;
    for (int i = 0; i < max_inst; ++i)
      if (track[(size_t)f * max_inst + i] >= 0) idx.push_back(i);
    if (order)
      std::stable_sort(idx.begin(), idx.end(),
                       [&](int a, int b) { return order[(size_t)f * max_inst + a] < order[(size_t)f * max_inst + b]; });
    return idx;
  };
  auto tracks_of = [&](int f, const std::vector<int>& idx) {
    std::vector<int> s;
    for (int i : idx) s.push_back(track[(size_t)f * max_inst + i]);
    std::sort(s.begin(), s.end());
    s.erase(std::unique(s.begin(), s.end()), s.end());
    return s;
  };
  auto has = [](const std::vector<int>& s, int v) { return std::binary_search(s.begin(), s.end(), v); };
  std::vector<std::pair<int, int>> fix;  // old -> new
  auto fix_find = [&](int tr) {
    for (auto& kv : fix)
      if (kv.first == tr) return kv.second;
    return -1;
  };
  std::vector<int> last_good = tracks_of(0, frame_list(0));
  for (int f = 0; f < n_frames; ++f) {
    const std::vector<int> idx = frame_list(f);
    std::vector<int> ft = tracks_of(f, idx);
    bool fixed_before = false;
    for (int tr : ft)
      if (fix_find(tr) >= 0) fixed_before = true;
    if (fixed_before) {
      for (int i : idx) {
        int& tr = track[(size_t)f * max_inst + i];
        const int nt = fix_find(tr);
        if (nt >= 0 && !has(ft, nt)) {
          tr = nt;
          ft = tracks_of(f, idx);
        }
      }
    }
    std::vector<int> extra, missing;
    for (int tr : ft)
      if (!has(last_good, tr)) extra.push_back(tr);
    for (int tr : last_good)
      if (!has(ft, tr)) missing.push_back(tr);
    if (extra.size() == 1 && missing.size() == 1) {
      for (int i : idx) {
        int& tr = track[(size_t)f * max_inst + i];
        if (tr == extra[0]) {
          // the reference keeps a stale entry when the same old track is fixed twice: dict assignment overwrites
          bool found = false;
          for (auto& kv : fix)
            if (kv.first == tr) {
              kv.second = missing[0];
              found = true;
            }
          if (!found) fix.emplace_back(tr, missing[0]);
          tr = missing[0];
          break;
        }
      }
    } else if ((int)ft.size() == instance_count) {
      last_good = ft;
    }
  }
  return SA_OK;
}

}  // extern "C"
