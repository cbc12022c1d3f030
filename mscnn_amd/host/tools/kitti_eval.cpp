// kitti_eval: the KITTI 2-D object benchmark's precision / recall evaluation for the detections this runtime writes
// (mscnn_amd/kitti.py::write_kitti_labels).  Behaviour restated from the devkit evaluator the reference ships
// (examples/kitti_result/eval/evaluate_object.cpp): difficulty filters (:25-37), neighbouring classes and DontCare areas
// (cleanData :267-347), the two-pass assignment (computeStatistics :349-484: best SCORE per ground truth to collect the
// recall thresholds, then best OVERLAP per ground truth at every threshold), the recall sampling quirk of getThresholds
// (:231-265), 41 precision samples made monotone from the right (:540-556), optional orientation similarity.
// Not restated: gnuplot / pdf output.  Usage:  kitti_eval <gt_dir> <result_dir> <list_file>
//   reads <gt_dir>/<id>.txt and <result_dir>/data/<id>.txt for every id in the list, writes
//   <result_dir>/stats_<class>_detection.txt (3 lines: easy / moderate / hard, 41 values, "%f ") like the devkit, and prints
//   the 11-point AP (mean of the precision at recall 0, 0.1, ..., 1) per class and difficulty.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <strings.h>
#include <vector>

namespace {

constexpr int kSamples = 41;
const int kMinHeight[3] = {40, 25, 25};
const int kMaxOcclusion[3] = {0, 1, 2};
const double kMaxTruncation[3] = {0.15, 0.3, 0.5};
const char* const kClassName[3] = {"car", "pedestrian", "cyclist"};
const char* const kNeighbour[3] = {"van", "person_sitting", ""};
const double kMinOverlap[3] = {0.7, 0.5, 0.5};

struct Obj {
  std::string type;
  double x1, y1, x2, y2, alpha;
  double truncation = -1;      // ground truth only
  int occlusion = -1;
  double score = -1000;        // detections only
};

struct Frame { std::vector<Obj> gt, det; };

bool same(const std::string& a, const char* b) { return strcasecmp(a.c_str(), b) == 0; }

bool read_objects(const std::string& path, bool detections, std::vector<Obj>* out) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char name[256];
  while (fscanf(f, "%255s", name) == 1) {
    double v[15];
    const int n = detections ? 15 : 14;          // a detection line carries the score as 16th field
    int got = 0;
    for (; got < n; ++got)
      if (fscanf(f, "%lf", &v[got]) != 1) break;
    if (got != n) break;                          // ragged tail: the devkit's fscanf loop stops here as well
    Obj o;
    o.type = name;
    o.truncation = v[0]; o.occlusion = (int)v[1]; o.alpha = v[2];
    o.x1 = v[3]; o.y1 = v[4]; o.x2 = v[5]; o.y2 = v[6];
    if (detections) o.score = v[14];
    out->push_back(o);
  }
  fclose(f);
  return true;
}

// criterion: -1 intersection over union, 0 intersection over a's area
double overlap(const Obj& a, const Obj& b, int criterion) {
  const double w = std::min(a.x2, b.x2) - std::max(a.x1, b.x1), h = std::min(a.y2, b.y2) - std::max(a.y1, b.y1);
  if (w <= 0 || h <= 0) return 0;
  const double inter = w * h, aa = (a.x2 - a.x1) * (a.y2 - a.y1), ab = (b.x2 - b.x1) * (b.y2 - b.y1);
  return criterion == 0 ? inter / aa : inter / (aa + ab - inter);
}

// role of every ground-truth box for (class, difficulty): 0 counted, 1 ignored (neighbour class / too hard), -1 other class
struct Roles { std::vector<int> gt, det; std::vector<int> dontcare; int counted = 0; };

Roles classify(const Frame& f, int cls, int diff) {
  Roles r;
  for (size_t i = 0; i < f.gt.size(); ++i) {
    const Obj& g = f.gt[i];
    const int valid = same(g.type, kClassName[cls]) ? 1 : (kNeighbour[cls][0] && same(g.type, kNeighbour[cls])) ? 0 : -1;
    const bool hard = g.occlusion > kMaxOcclusion[diff] || g.truncation > kMaxTruncation[diff] || (g.y2 - g.y1) < kMinHeight[diff];
    int role = -1;
    if (valid == 1 && !hard) { role = 0; ++r.counted; }
    else if (valid == 0 || (valid == 1 && hard)) role = 1;
    r.gt.push_back(role);
    if (same(g.type, "DontCare")) r.dontcare.push_back((int)i);
  }
  for (const Obj& d : f.det) r.det.push_back(same(d.type, kClassName[cls]) ? 0 : -1);
  return r;
}

struct Counts { int tp = 0, fp = 0, fn = 0; double similarity = 0; bool has_similarity = false; std::vector<double> scores; };

// One frame at one score threshold.  with_fp == false: the recall pass (no threshold, best score wins, scores collected).
Counts assign(const Frame& f, const Roles& r, int cls, bool with_fp, bool with_aos, double thresh) {
  Counts c;
  const size_t nd = f.det.size();
  std::vector<char> taken(nd, 0), low(nd, 0);
  std::vector<double> delta;
  if (with_fp)
    for (size_t j = 0; j < nd; ++j) low[j] = f.det[j].score < thresh;
  for (size_t i = 0; i < f.gt.size(); ++i) {
    if (r.gt[i] == -1) continue;
    int best = -1;
    double best_key = with_fp ? 0.0 : -1e7;          // greatest overlap (> 0) resp. greatest score (> -10000000)
    for (size_t j = 0; j < nd; ++j) {
      if (r.det[j] == -1 || taken[j] || low[j]) continue;
      const double o = overlap(f.det[j], f.gt[i], -1);
      if (!(o > kMinOverlap[cls])) continue;
      const double key = with_fp ? o : f.det[j].score;
      if (key > best_key) { best_key = key; best = (int)j; }      // strict: the first of equal candidates stays
    }
    if (best < 0) { if (r.gt[i] == 0) ++c.fn; continue; }
    taken[best] = 1;
    if (r.gt[i] == 1) continue;                       // matched an ignored ground truth: neither TP nor FP
    ++c.tp;
    c.scores.push_back(f.det[best].score);
    if (with_aos) delta.push_back(f.gt[i].alpha - f.det[best].alpha);
  }
  if (!with_fp) return c;
  for (size_t j = 0; j < nd; ++j)
    if (!taken[j] && r.det[j] == 0 && !low[j]) ++c.fp;
  // detections that lie (mostly) inside a DontCare area are not false positives
  for (int gi : r.dontcare)
    for (size_t j = 0; j < nd; ++j) {
      if (taken[j] || r.det[j] != 0 || low[j]) continue;
      if (overlap(f.det[j], f.gt[gi], 0) > kMinOverlap[cls]) { taken[j] = 1; --c.fp; }
    }
  if (with_aos && (c.tp > 0 || c.fp > 0)) {
    c.has_similarity = true;
    for (double d : delta) c.similarity += (1.0 + std::cos(d)) / 2.0;      // false positives contribute 0
  }
  return c;
}

// Scores at which the recall curve is sampled (kSamples points): a score is kept when the recall it yields is the
// closest available approximation of the next sample recall.
std::vector<double> recall_thresholds(std::vector<double> scores, double n_gt) {
  std::sort(scores.begin(), scores.end(), [](double a, double b) { return a > b; });
  std::vector<double> t;
  double current = 0;
  for (size_t i = 0; i < scores.size(); ++i) {
    const double left = (double)(i + 1) / n_gt;
    const bool last = i + 1 == scores.size();
    const double right = last ? left : (double)(i + 2) / n_gt;
    if (!last && (right - current) < (current - left)) continue;
    t.push_back(scores[i]);
    current += 1.0 / (kSamples - 1.0);
  }
  return t;
}

void evaluate(const std::vector<Frame>& frames, int cls, int diff, bool with_aos, std::vector<double>* precision, std::vector<double>* aos) {
  std::vector<Roles> roles;
  std::vector<double> scores;
  int n_gt = 0;
  for (const Frame& f : frames) {
    roles.push_back(classify(f, cls, diff));
    n_gt += roles.back().counted;
    const Counts c = assign(f, roles.back(), cls, false, false, 0);
    scores.insert(scores.end(), c.scores.begin(), c.scores.end());
  }
  const std::vector<double> thr = recall_thresholds(scores, n_gt);
  precision->assign(kSamples, 0.0);
  aos->assign(with_aos ? kSamples : 0, 0.0);
  for (size_t t = 0; t < thr.size(); ++t) {
    long tp = 0, fp = 0;
    double sim = 0;
    for (size_t k = 0; k < frames.size(); ++k) {
      const Counts c = assign(frames[k], roles[k], cls, true, with_aos, thr[t]);
      tp += c.tp; fp += c.fp;
      if (c.has_similarity) sim += c.similarity;
    }
    (*precision)[t] = tp / (double)(tp + fp);
    if (with_aos) (*aos)[t] = sim / (double)(tp + fp);
  }
  for (size_t t = 0; t < thr.size(); ++t) {           // monotone from the right: max over [t, end)
    (*precision)[t] = *std::max_element(precision->begin() + t, precision->end());
    if (with_aos) (*aos)[t] = *std::max_element(aos->begin() + t, aos->end());
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s gt_dir result_dir list_file\n", argv[0]); return 1; }
  const std::string gt_dir = argv[1], res_dir = argv[2];
  std::vector<std::string> ids;
  { std::ifstream in(argv[3]); std::string line; while (std::getline(in, line)) ids.push_back(line); }
  std::vector<Frame> frames(ids.size());
  bool with_aos = true, seen[3] = {false, false, false};
  for (size_t i = 0; i < ids.size(); ++i) {
    if (!read_objects(gt_dir + "/" + ids[i] + ".txt", false, &frames[i].gt)) { fprintf(stderr, "cannot read ground truth %s\n", ids[i].c_str()); return 2; }
    if (!read_objects(res_dir + "/data/" + ids[i] + ".txt", true, &frames[i].det)) { fprintf(stderr, "cannot read detections %s\n", ids[i].c_str()); return 2; }
    for (const Obj& d : frames[i].det) {
      if (d.alpha == -10) with_aos = false;            // one unknown orientation switches AOS off for the submission
      for (int c = 0; c < 3; ++c) if (same(d.type, kClassName[c])) seen[c] = true;
    }
  }
  static const char* const kDiff[3] = {"easy", "moderate", "hard"};
  for (int c = 0; c < 3; ++c) {
    if (!seen[c]) continue;                            // a class is evaluated only if it was detected at least once
    FILE* fd = fopen((res_dir + "/stats_" + kClassName[c] + "_detection.txt").c_str(), "w");
    FILE* fo = with_aos ? fopen((res_dir + "/stats_" + kClassName[c] + "_orientation.txt").c_str(), "w") : nullptr;
    if (!fd) { fprintf(stderr, "cannot write into %s\n", res_dir.c_str()); return 3; }
    for (int diff = 0; diff < 3; ++diff) {
      std::vector<double> p, a;
      evaluate(frames, c, diff, with_aos, &p, &a);
      for (double v : p) fprintf(fd, "%f ", v);
      fprintf(fd, "\n");
      if (fo) { for (double v : a) fprintf(fo, "%f ", v); fprintf(fo, "\n"); }
      double ap = 0;
      for (int i = 0; i < kSamples; i += 4) ap += p[i];
      printf("%s %s AP(11-point) %.4f\n", kClassName[c], kDiff[diff], 100.0 * ap / 11.0);
    }
    fclose(fd);
    if (fo) fclose(fo);
  }
  return 0;
}
