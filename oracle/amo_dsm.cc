/*
 * oracle/amo_dsm.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle for dsm::Dsm).
 *
 * CPU restatement of the reference's point-cloud -> DSM path:
 *   Dsm::initializeAndFillKdTree          aerial_mapper_dsm/src/dsm.cc:36-52
 *   Dsm::updateElevationLayer             dsm.cc:54-111   (single thread)
 *   Dsm::updateElevationLayerMultiThreaded dsm.cc:113-184 (utils::parFor)
 *   Dsm::process                          dsm.cc:186-201
 *   utils::parFor                         aerial_mapper_utils/include/
 *                                         aerial-mapper-utils/utils-common.h:29-59
 *
 * Two builds of this one file (oracle/Makefile):
 *   liboracle.so           kd-tree = oracle/amo_kdtree2d.h (own restatement)
 *   _ref/liboracle_ref.so  kd-tree = the reference's VENDORED nanoflann.hpp,
 *                          included unchanged from /root/reference
 *                          (-DAMO_USE_VENDORED_NANOFLANN), driven exactly the
 *                          way dsm.cc drives it (RadiusResultSet +
 *                          findNeighbors + the shared result vector).
 *
 * Pinning: PARITY UNPINNED, except the kd-tree.  The reference has NO tests and
 * no golden vectors, and dsm.cc cannot be built here (it
 * needs Eigen, grid_map_core, glog, ROS: absent from the image and from
 * /root/reference).  What IS the reference's own code run here: the kd-tree
 * build and the radius search -- the vendored nanoflann.hpp compiles by
 * itself, the _ref build of this file drives it exactly as dsm.cc does.  The
 * loops around it are a restatement; grid_map_core's arithmetic behind
 * setGeometry / getPosition is adopted (amo_compat.h).
 * Consistency check (NOT a pin): oracle/refkit/ holds builder-written stand-in
 * headers over which the text of dsm.cc / ortho-from-pcl.cc compiles unchanged
 * (_ref/libref_loops_*.so; tests/test_reference_loops.py holds this file to it
 * bit for bit).  By the task's rules a build over stand-ins is not a reference
 * build: it catches a mis-read of the loops' control flow, nothing more, and
 * is neither the timed CPU baseline nor bench.py's parity checker.
 */
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <memory>
#include <thread>
#include <utility>
#include <vector>

#include "amo_compat.h"
#include "amo_types.h"

#ifdef AMO_USE_VENDORED_NANOFLANN
#include <aerial-mapper-thirdparty/nanoflann.hpp>
#else
#include "amo_kdtree2d.h"
#endif

namespace amo {

#ifdef AMO_USE_VENDORED_NANOFLANN
// Dataset + adaptor with the member names nanoflann expects
// (reference: utils-nearest-neighbor.h:24-76, Eigen include dropped).
struct RefCloud {
  struct Point {
    double x, y, z;
  };
  std::vector<Point> pts;
};
struct RefAdaptor {
  const RefCloud& obj;
  explicit RefAdaptor(const RefCloud& o) : obj(o) {}
  inline size_t kdtree_get_point_count() const { return obj.pts.size(); }
  inline double kdtree_get_pt(const size_t idx, int dim) const {
    if (dim == 0) return obj.pts[idx].x;
    if (dim == 1) return obj.pts[idx].y;
    return obj.pts[idx].z;
  }
  template <class BBOX>
  bool kdtree_get_bbox(BBOX&) const {
    return false;
  }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<
    nanoflann::L2_Adaptor<double, RefAdaptor>, RefAdaptor, 2>
    RefTree;
#endif

class DsmOracle {
 public:
  DsmOracle(const amo_grid& g, int radius_sq, double center_e, double center_n, int knn_k = 0)
      : grid_(g),
        radius_(radius_sq),
        center_e_(center_e),
        center_n_(center_n),
        knn_k_(knn_k),
        error_(AMO_OK) {}

  // dsm.cc:36-52 -- note x uses center_NORTHING and y center_EASTING.
  void fill_and_build(const double* xyz, size_t n) {
#ifdef AMO_USE_VENDORED_NANOFLANN
    cloud_.pts.resize(n);
    for (size_t i = 0; i < n; ++i) {
      cloud_.pts[i].x = xyz[3 * i + 0] - center_n_;
      cloud_.pts[i].y = xyz[3 * i + 1] - center_e_;
      cloud_.pts[i].z = xyz[3 * i + 2];
    }
    adaptor_.reset(new RefAdaptor(cloud_));
    tree_.reset(new RefTree(
        2, *adaptor_, nanoflann::KDTreeSingleIndexAdaptorParams(10)));
    tree_->buildIndex();
#else
    pts_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      pts_[i].x = xyz[3 * i + 0] - center_n_;
      pts_[i].y = xyz[3 * i + 1] - center_e_;
      pts_[i].z = xyz[3 * i + 2];
    }
    tree_.reset(new KdTree2D(pts_));
    tree_->build();
#endif
  }

  // One cell of dsm.cc:119-174 (identical to :58-106).
  void cell(int i, int j, float* elevation) {
    double qx, qy;
    cell_position(grid_, i, j, &qx, &qy);

    std::vector<std::pair<int, double> > hits;
#ifdef AMO_USE_VENDORED_NANOFLANN
    nanoflann::RadiusResultSet<double, int> result_set(radius_, hits);
    const double query_pt[3] = {qx, qy, 0.0};
    tree_->findNeighbors(result_set, query_pt, nanoflann::SearchParams());
    {
      double lambda = 1.0;
      while (result_set.size() == 0u) {
        nanoflann::RadiusResultSet<double, int> tmp(lambda * radius_, hits);
        tree_->findNeighbors(tmp, query_pt, nanoflann::SearchParams());
        lambda *= 1.1;
        if (lambda * radius_ > 7.0) break;
      }
    }
#else
    // RadiusResultSet(radius, vec) clears vec, then the tree appends.
    hits.clear();
    tree_->radius_search(qx, qy, static_cast<double>(radius_), &hits);
    {
      // Expanding-radius fallback: the temporary result set shares `hits`,
      // so the loop condition sees what the retry found (dsm.cc:133-144).
      double lambda = 1.0;
      while (hits.size() == 0u) {
        hits.clear();
        tree_->radius_search(qx, qy, lambda * radius_, &hits);
        lambda *= 1.1;
        if (lambda * radius_ > 7.0) break;
      }
    }
#endif
    if (hits.empty()) return;  // cell left untouched

    // OPTIONAL capped mode (BASELINE.json "IDW k=4"; NOT a reference code path -- the
    // reference weights ALL points of the radius search): only the k
    // nearest of the search's result take part, in nanoflann::KNNResultSet's order
    // (nanoflann.hpp:80-131: ascending distance, a later arrival never displaces an equal
    // distance).  Vendored build: KNNResultSet itself on the same tree (the k nearest of the
    // whole cloud are the k nearest of the search result as soon as that holds more than k
    // points); port: a stable sort of the search result.  Parity of this mode is UNPINNED
    // by the reference; it is reported separately and never graded.
    if (knn_k_ > 0) {
      if (static_cast<int>(hits.size()) > knn_k_) {
#ifdef AMO_USE_VENDORED_NANOFLANN
        std::vector<int> idx(knn_k_);
        std::vector<double> d2s(knn_k_);
        nanoflann::KNNResultSet<double, int> knn(knn_k_);
        knn.init(idx.data(), d2s.data());
        tree_->findNeighbors(knn, query_pt, nanoflann::SearchParams());
        hits.clear();
        for (int k = 0; k < knn_k_; ++k) hits.push_back(std::make_pair(idx[k], d2s[k]));
#else
        std::stable_sort(hits.begin(), hits.end(),
                         [](const std::pair<int, double>& a, const std::pair<int, double>& b) {
                           return a.second < b.second;
                         });
        hits.resize(knn_k_);
#endif
      } else {
        std::stable_sort(hits.begin(), hits.end(),
                         [](const std::pair<int, double>& a, const std::pair<int, double>& b) {
                           return a.second < b.second;
                         });
      }
    }

    double num = 0.0, den = 0.0;
    for (size_t k = 0; k < hits.size(); ++k) {
      const double d2 = hits[k].second;
      const double h = height(hits[k].first);
      if (!(d2 > 0.0)) {  // CHECK(distances[i] > 0.0), dsm.cc:165
        error_ = AMO_ERR_EXACT_HIT;
        return;
      }
      num += h / d2;
      den += 1.0 / d2;
    }
    const double idw = num / den;
    elevation[static_cast<size_t>(i) +
              static_cast<size_t>(j) * static_cast<size_t>(grid_.rows)] =
        static_cast<float>(idw);
  }

  int error() const { return error_; }
  const amo_grid& grid() const { return grid_; }

 private:
  double height(int idx) const {
#ifdef AMO_USE_VENDORED_NANOFLANN
    return cloud_.pts[idx].z;
#else
    return pts_[idx].z;
#endif
  }

  amo_grid grid_;
  int radius_;  // dsm::Settings::interpolation_radius is an int (dsm.h:27)
  double center_e_, center_n_;
  int knn_k_;  // 0: the reference's behaviour (every point of the radius search)
  volatile int error_;
#ifdef AMO_USE_VENDORED_NANOFLANN
  RefCloud cloud_;
  std::unique_ptr<RefAdaptor> adaptor_;
  std::unique_ptr<RefTree> tree_;
#else
  std::vector<KdPoint> pts_;
  std::unique_ptr<KdTree2D> tree_;
#endif
};

// utils::parFor (utils-common.h:29-59): ceil(n/threads) items per block,
// one std::thread per block, contiguous index ranges.
template <typename F>
static void par_for(size_t num_items, const F& fn, size_t num_threads) {
  if (num_threads == 0) num_threads = 1;
  const size_t per_block = static_cast<size_t>(
      std::ceil(static_cast<double>(num_items) /
                static_cast<double>(num_threads)));
  if (per_block == 0) return;
  const size_t num_blocks = static_cast<size_t>(std::ceil(
      static_cast<double>(num_items) / static_cast<double>(per_block)));
  std::vector<std::thread> threads;
  for (size_t b = 0; b < num_blocks; ++b) {
    const size_t lo = b * per_block;
    const size_t hi = (lo + per_block < num_items) ? lo + per_block : num_items;
    threads.push_back(std::thread([&fn, lo, hi]() { fn(lo, hi); }));
  }
  for (size_t b = 0; b < threads.size(); ++b) threads[b].join();
}

}  // namespace amo

static double now_s() {
  using namespace std::chrono;
  return duration_cast<duration<double> >(
             steady_clock::now().time_since_epoch())
      .count();
}

extern "C" {

/* Which kd-tree this build carries: 1 = vendored nanoflann, 0 = port. */
int amo_uses_vendored_nanoflann(void) {
#ifdef AMO_USE_VENDORED_NANOFLANN
  return 1;
#else
  return 0;
#endif
}

/*
 * dsm::Dsm(settings, map) + Dsm::process(point_cloud, map).
 *   xyz          n points, AoS x,y,z doubles (the memory layout of
 *                std::vector<Eigen::Vector3d, aligned_allocator>)
 *   elevation    in/out, float32 column-major rows*cols ("elevation" layer)
 *   multi_thread dsm::Settings::use_multi_threads
 *   num_threads  0 -> std::thread::hardware_concurrency() (dsm.cc:178)
 *   timing       optional double[2]: kd-tree fill+build seconds, cell loop s
 * Empty cloud is the reference's soft no-op (dsm.cc:189-192).
 */
static int dsm_process_impl(const double* xyz, size_t n, const amo_grid* grid,
                    int radius_sq, double center_easting,
                    double center_northing, int multi_thread, int num_threads,
                    float* elevation, double* timing, int knn_k) {
  if (!grid || !elevation || (n && !xyz)) return AMO_ERR_ARG;
  if (timing) timing[0] = timing[1] = 0.0;
  if (n == 0) return AMO_OK;

  amo::DsmOracle dsm(*grid, radius_sq, center_easting, center_northing, knn_k);
  const double t0 = now_s();
  dsm.fill_and_build(xyz, n);
  const double t1 = now_s();

  const size_t cells =
      static_cast<size_t>(grid->rows) * static_cast<size_t>(grid->cols);
  auto range = [&](size_t lo, size_t hi) {
    for (size_t lin = lo; lin < hi; ++lin) {
      int i, j;
      amo::linear_to_index(dsm.grid(), lin, &i, &j);
      dsm.cell(i, j, elevation);
    }
  };
  if (multi_thread) {
    size_t nt = num_threads > 0 ? static_cast<size_t>(num_threads)
                                : std::thread::hardware_concurrency();
    amo::par_for(cells, range, nt);
  } else {
    range(0, cells);
  }
  const double t2 = now_s();
  if (timing) {
    timing[0] = t1 - t0;
    timing[1] = t2 - t1;
  }
  return dsm.error();
}

int amo_dsm_process(const double* xyz, size_t n, const amo_grid* grid,
                    int radius_sq, double center_easting,
                    double center_northing, int multi_thread, int num_threads,
                    float* elevation, double* timing) {
  return dsm_process_impl(xyz, n, grid, radius_sq, center_easting, center_northing, multi_thread,
                          num_threads, elevation, timing, 0);
}

/* The OPTIONAL capped mode: as amo_dsm_process, with only the knn_k nearest points of every
 * cell's search result (see DsmOracle::cell).  Not a reference code path. */
int amo_dsm_process_knn(const double* xyz, size_t n, const amo_grid* grid, int radius_sq,
                        double center_easting, double center_northing, int knn_k,
                        int multi_thread, int num_threads, float* elevation) {
  if (knn_k < 1) return AMO_ERR_ARG;
  return dsm_process_impl(xyz, n, grid, radius_sq, center_easting, center_northing, multi_thread,
                          num_threads, elevation, nullptr, knn_k);
}

/* Neighbour probe used by the known-answer tests: radius search around one
 * query, returns the hit count and (optionally) the first `cap` squared
 * distances / indices in visiting order. */
int amo_dsm_radius_probe(const double* xyz, size_t n, double qx, double qy,
                         double radius_sq, int cap, int* idx_out,
                         double* d2_out) {
  amo_grid g = amo::make_grid(1.0, 1.0, 1.0, 0.0, 0.0);
  (void)g;
  std::vector<std::pair<int, double> > hits;
#ifdef AMO_USE_VENDORED_NANOFLANN
  amo::RefCloud cloud;
  cloud.pts.resize(n);
  for (size_t i = 0; i < n; ++i) {
    cloud.pts[i].x = xyz[3 * i];
    cloud.pts[i].y = xyz[3 * i + 1];
    cloud.pts[i].z = xyz[3 * i + 2];
  }
  amo::RefAdaptor ad(cloud);
  amo::RefTree tree(2, ad, nanoflann::KDTreeSingleIndexAdaptorParams(10));
  tree.buildIndex();
  nanoflann::RadiusResultSet<double, int> rs(radius_sq, hits);
  const double q[3] = {qx, qy, 0.0};
  tree.findNeighbors(rs, q, nanoflann::SearchParams());
#else
  std::vector<amo::KdPoint> pts(n);
  for (size_t i = 0; i < n; ++i) {
    pts[i].x = xyz[3 * i];
    pts[i].y = xyz[3 * i + 1];
    pts[i].z = xyz[3 * i + 2];
  }
  amo::KdTree2D tree(pts);
  tree.build();
  tree.radius_search(qx, qy, radius_sq, &hits);
#endif
  for (int k = 0; k < cap && k < static_cast<int>(hits.size()); ++k) {
    if (idx_out) idx_out[k] = hits[k].first;
    if (d2_out) d2_out[k] = hits[k].second;
  }
  return static_cast<int>(hits.size());
}

/* amo::make_grid for ctypes callers (grid_map_core setGeometry). */
void amo_make_grid(double length_x, double length_y, double resolution,
                   double pos_x, double pos_y, amo_grid* out) {
  *out = amo::make_grid(length_x, length_y, resolution, pos_x, pos_y);
}

/* Cell-centre position (grid_map_core getPosition). */
void amo_cell_position(const amo_grid* g, int i, int j, double* x, double* y) {
  amo::cell_position(*g, i, j, x, y);
}

}  // extern "C"

// ---------------------------------------------------------------------------
// ortho::OrthoFromPcl::process  
//   aerial_mapper_ortho/src/ortho-from-pcl.cc:20-113
// Same kd-tree radius search + inverse-squared-distance weighting as the DSM,
// but: the interpolated quantity is the point's INTENSITY (an int), there is no
// centre offset (:30-31), an exact hit (d2 == 0) short-circuits to that
// point's value (:91-96), the fallback only exists with
// use_adaptive_interpolation and multiplies the squared radius by an INT
// lambda = 10, 100, ... without upper bound (:63-71), the loop is always
// single-threaded and the result goes to the "ortho" layer.
// ---------------------------------------------------------------------------
extern "C" int amo_ortho_from_pcl_process(const double* xyz, const int32_t* intensities,
                                          size_t n, const amo_grid* grid, int radius_sq,
                                          int adaptive, float* ortho) {
  if (!grid || !ortho || !xyz || !intensities || n == 0) return AMO_ERR_ARG;  // CHECK(!empty)
#ifdef AMO_USE_VENDORED_NANOFLANN
  amo::RefCloud cloud;
  cloud.pts.resize(n);
  for (size_t i = 0; i < n; ++i) {
    cloud.pts[i].x = xyz[3 * i + 0];
    cloud.pts[i].y = xyz[3 * i + 1];
    cloud.pts[i].z = double(intensities[i]);
  }
  amo::RefAdaptor ad(cloud);
  amo::RefTree tree(2, ad, nanoflann::KDTreeSingleIndexAdaptorParams(10));
  tree.buildIndex();
#else
  std::vector<amo::KdPoint> pts(n);
  for (size_t i = 0; i < n; ++i) {
    pts[i].x = xyz[3 * i + 0];
    pts[i].y = xyz[3 * i + 1];
    pts[i].z = double(intensities[i]);
  }
  amo::KdTree2D tree(pts);
  tree.build();
#endif
  const size_t cells = static_cast<size_t>(grid->rows) * static_cast<size_t>(grid->cols);
  for (size_t lin = 0; lin < cells; ++lin) {
    int i, j;
    amo::linear_to_index(*grid, lin, &i, &j);
    double qx, qy;
    amo::cell_position(*grid, i, j, &qx, &qy);
    std::vector<std::pair<int, double> > hits;
#ifdef AMO_USE_VENDORED_NANOFLANN
    nanoflann::RadiusResultSet<double, int> result_set(radius_sq, hits);
    const double query_pt[3] = {qx, qy, 0.0};
    tree.findNeighbors(result_set, query_pt, nanoflann::SearchParams());
    if (adaptive) {
      int lambda = 10;
      while (result_set.size() == 0u) {
        nanoflann::RadiusResultSet<double, int> tmp(lambda * radius_sq, hits);
        tree.findNeighbors(tmp, query_pt, nanoflann::SearchParams());
        lambda *= 10;
      }
    }
#else
    hits.clear();
    tree.radius_search(qx, qy, static_cast<double>(radius_sq), &hits);
    if (adaptive) {
      int lambda = 10;
      while (hits.size() == 0u) {
        hits.clear();
        tree.radius_search(qx, qy, static_cast<double>(lambda * radius_sq), &hits);
        lambda *= 10;
      }
    }
#endif
    if (hits.empty()) continue;
    double num = 0.0, den = 0.0;
    bool perfect = false;
    for (size_t k = 0; k < hits.size(); ++k) {
      const double d2 = hits[k].second;
#ifdef AMO_USE_VENDORED_NANOFLANN
      const double h = cloud.pts[hits[k].first].z;
#else
      const double h = pts[hits[k].first].z;
#endif
      if (d2 == 0.0) {  // perfect match, no interpolation needed
        num = h;
        den = 1.0;
        perfect = true;
      }
      if (!perfect) {
        num += h / d2;
        den += 1.0 / d2;
      }
    }
    ortho[static_cast<size_t>(i) + static_cast<size_t>(j) * static_cast<size_t>(grid->rows)] =
        static_cast<float>(num / den);
  }
  return AMO_OK;
}
