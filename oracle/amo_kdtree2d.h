/*
 * oracle/amo_kdtree2d.h -- TEST INFRASTRUCTURE ONLY (CPU oracle, "port").
 *
 * A from-scratch restatement of the 2-D kd-tree fixed-radius search the
 * reference's DSM runs through its vendored nanoflann v1.2.2
 * (aerial_mapper_thirdparty/include/aerial-mapper-thirdparty/nanoflann.hpp,
 * NANOFLANN_VERSION 0x122 at :74), specialised to what dsm.h:57-65 / dsm.cc
 * instantiate: DIM = 2, double coordinates, squared-L2 metric, leaf size 10,
 * SearchParams() (eps = 0).  It exists so that the oracle can be (re)built
 * and timed on a box that has no /root/reference; oracle/_ref/ holds the same
 * driver compiled against the vendored header itself, and
 * tests/test_oracle.py requires the two to agree BIT FOR BIT (same
 * neighbours, same visiting order, hence same double sums).
 *
 * To get the same visiting order the build has to make the same choices:
 *   - root bounding box over all points           (nanoflann.hpp:1045-1066)
 *   - leaf when count <= 10, leaf bbox recomputed  (:1079-1097)
 *   - split dimension: the version vendored by the reference measures the
 *     point spread along the CURRENT best dimension rather than along the
 *     candidate one (:1152), which reduces to
 *        cut = (span0 > (1 - 1e-5) * max_span) ? 0 : 1
 *   - cut value = bbox mid-point clamped to the points' [min,max] (:1162-1171)
 *   - three-way partition (<, ==, >) and the balanced pick of the split
 *     position (:1173-1181, :1195-1228)
 *   - children shrink the box they were handed; parent keeps
 *     divlow = left.high, divhigh = right.low and the union box (:1104-1117)
 * and the search has to descend into the nearer child first, then the other
 * one iff  mindistsq <= radius  (:1254-1302), testing leaf points with a
 * strict  d2 < radius  (:156-158, :1265).
 */
#ifndef AMO_KDTREE2D_H_
#define AMO_KDTREE2D_H_

#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace amo {

struct KdPoint {
  double x, y, z;
};

class KdTree2D {
 public:
  static const size_t kLeafMax = 10;  // dsm.h:57

  explicit KdTree2D(const std::vector<KdPoint>& pts) : pts_(pts) {}

  void build() {
    const size_t n = pts_.size();
    order_.resize(n);
    for (size_t i = 0; i < n; ++i) order_[i] = i;
    nodes_.clear();
    root_ = -1;
    if (n == 0) return;
    nodes_.reserve(n / 4 + 16);
    Box box;
    box.lo[0] = box.hi[0] = pts_[0].x;
    box.lo[1] = box.hi[1] = pts_[0].y;
    for (size_t k = 1; k < n; ++k) {
      for (int d = 0; d < 2; ++d) {
        const double c = coord(k, d);
        if (c < box.lo[d]) box.lo[d] = c;
        if (c > box.hi[d]) box.hi[d] = c;
      }
    }
    root_box_ = box;
    root_ = divide(0, n, &box);
  }

  /* Appends (index, d2) for every point with d2 < radius_sq, in tree
   * visiting order.  `out` is NOT cleared here (the caller mirrors
   * RadiusResultSet's constructor, which does). */
  void radius_search(double qx, double qy, double radius_sq,
                     std::vector<std::pair<int, double> >* out) const {
    if (root_ < 0) return;
    const double q[2] = {qx, qy};
    double side[2] = {0.0, 0.0};
    double mind = 0.0;
    for (int d = 0; d < 2; ++d) {
      if (q[d] < root_box_.lo[d]) {
        side[d] = (q[d] - root_box_.lo[d]) * (q[d] - root_box_.lo[d]);
        mind += side[d];
      }
      if (q[d] > root_box_.hi[d]) {
        side[d] = (q[d] - root_box_.hi[d]) * (q[d] - root_box_.hi[d]);
        mind += side[d];
      }
    }
    descend(root_, q, mind, side, radius_sq, out);
  }

  size_t size() const { return pts_.size(); }

 private:
  struct Box {
    double lo[2], hi[2];
  };
  struct Node {
    int32_t left, right;  // children (node ids); -1/-1 marks a leaf
    // leaf: [first, last) into order_;  inner: cut dimension + the two planes
    size_t first, last;
    int dim;
    double divlow, divhigh;
  };

  double coord(size_t idx, int d) const {
    return d == 0 ? pts_[idx].x : pts_[idx].y;
  }

  void min_max(const size_t* ind, size_t count, int d, double* mn,
               double* mx) const {
    *mn = *mx = coord(ind[0], d);
    for (size_t i = 1; i < count; ++i) {
      const double v = coord(ind[i], d);
      if (v < *mn) *mn = v;
      if (v > *mx) *mx = v;
    }
  }

  /* Three-way partition around `cut` along `d`; afterwards
   *   [0,lim1) < cut, [lim1,lim2) == cut, [lim2,count) > cut.
   * Uses unsigned indices with the same "stop when right reaches 0" guard
   * as the vendored code so the resulting permutation is identical. */
  void partition3(size_t* ind, size_t count, int d, double cut, size_t* lim1,
                  size_t* lim2) const {
    size_t lo = 0;
    size_t hi = count - 1;
    while (true) {
      while (lo <= hi && coord(ind[lo], d) < cut) ++lo;
      while (hi != 0 && lo <= hi && coord(ind[hi], d) >= cut) --hi;
      if (lo > hi || hi == 0) break;
      std::swap(ind[lo], ind[hi]);
      ++lo;
      --hi;
    }
    *lim1 = lo;
    hi = count - 1;
    while (true) {
      while (lo <= hi && coord(ind[lo], d) <= cut) ++lo;
      while (hi != 0 && lo <= hi && coord(ind[hi], d) > cut) --hi;
      if (lo > hi || hi == 0) break;
      std::swap(ind[lo], ind[hi]);
      ++lo;
      --hi;
    }
    *lim2 = lo;
  }

  int32_t divide(size_t first, size_t last, Box* box) {
    const int32_t id = static_cast<int32_t>(nodes_.size());
    nodes_.push_back(Node());
    if (last - first <= kLeafMax) {
      Node& nd = nodes_[id];
      nd.left = nd.right = -1;
      nd.first = first;
      nd.last = last;
      for (int d = 0; d < 2; ++d)
        box->lo[d] = box->hi[d] = coord(order_[first], d);
      for (size_t k = first + 1; k < last; ++k) {
        for (int d = 0; d < 2; ++d) {
          const double c = coord(order_[k], d);
          if (box->lo[d] > c) box->lo[d] = c;
          if (box->hi[d] < c) box->hi[d] = c;
        }
      }
      return id;
    }

    size_t* ind = &order_[0] + first;
    const size_t count = last - first;

    // choose the cut dimension (see header comment for the quirk)
    const double eps = 0.00001;
    const double span0 = box->hi[0] - box->lo[0];
    const double span1 = box->hi[1] - box->lo[1];
    double max_span = span0;
    if (span1 > max_span) max_span = span1;
    int dim = 0;
    {
      double best_spread = -1.0;
      for (int d = 0; d < 2; ++d) {
        const double span = (d == 0) ? span0 : span1;
        if (span > (1 - eps) * max_span) {
          double mn, mx;
          min_max(ind, count, dim, &mn, &mx);  // along the current best dim
          const double spread = mx - mn;
          if (spread > best_spread) {
            dim = d;
            best_spread = spread;
          }
        }
      }
    }

    const double mid = (box->lo[dim] + box->hi[dim]) / 2;
    double mn, mx;
    min_max(ind, count, dim, &mn, &mx);
    double cut;
    if (mid < mn)
      cut = mn;
    else if (mid > mx)
      cut = mx;
    else
      cut = mid;

    size_t lim1, lim2;
    partition3(ind, count, dim, cut, &lim1, &lim2);
    size_t split;
    if (lim1 > count / 2)
      split = lim1;
    else if (lim2 < count / 2)
      split = lim2;
    else
      split = count / 2;

    Box lbox = *box;
    lbox.hi[dim] = cut;
    const int32_t lchild = divide(first, first + split, &lbox);
    Box rbox = *box;
    rbox.lo[dim] = cut;
    const int32_t rchild = divide(first + split, last, &rbox);

    Node& nd = nodes_[id];  // (re-fetch: the vector may have grown)
    nd.left = lchild;
    nd.right = rchild;
    nd.dim = dim;
    nd.divlow = lbox.hi[dim];
    nd.divhigh = rbox.lo[dim];
    for (int d = 0; d < 2; ++d) {
      box->lo[d] = lbox.lo[d] < rbox.lo[d] ? lbox.lo[d] : rbox.lo[d];
      box->hi[d] = lbox.hi[d] > rbox.hi[d] ? lbox.hi[d] : rbox.hi[d];
    }
    return id;
  }

  void descend(int32_t id, const double* q, double mind, double* side,
               double radius_sq,
               std::vector<std::pair<int, double> >* out) const {
    const Node& nd = nodes_[id];
    if (nd.left < 0) {
      for (size_t k = nd.first; k < nd.last; ++k) {
        const size_t idx = order_[k];
        // L2_Adaptor with size == 2: result = 0; += dx*dx; += dy*dy
        const double dx = q[0] - pts_[idx].x;
        const double dy = q[1] - pts_[idx].y;
        double d2 = 0.0;
        d2 += dx * dx;
        d2 += dy * dy;
        if (d2 < radius_sq)
          out->push_back(std::make_pair(static_cast<int>(idx), d2));
      }
      return;
    }
    const int d = nd.dim;
    const double val = q[d];
    const double diff_lo = val - nd.divlow;
    const double diff_hi = val - nd.divhigh;
    int32_t near_child, far_child;
    double cut_dist;
    if ((diff_lo + diff_hi) < 0) {
      near_child = nd.left;
      far_child = nd.right;
      cut_dist = (val - nd.divhigh) * (val - nd.divhigh);
    } else {
      near_child = nd.right;
      far_child = nd.left;
      cut_dist = (val - nd.divlow) * (val - nd.divlow);
    }
    descend(near_child, q, mind, side, radius_sq, out);
    const double saved = side[d];
    mind = mind + cut_dist - saved;
    side[d] = cut_dist;
    // epsError = 1 + eps = 1.0f
    if (mind * 1.0f <= radius_sq)
      descend(far_child, q, mind, side, radius_sq, out);
    side[d] = saved;
  }

  const std::vector<KdPoint>& pts_;
  std::vector<size_t> order_;
  std::vector<Node> nodes_;
  Box root_box_;
  int32_t root_;
};

}  // namespace amo

#endif  // AMO_KDTREE2D_H_
