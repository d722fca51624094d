// oracle/refkit: stand-in for <grid_map_core/iterators/GridMapIterator.hpp> (see
// ../../refkit.h): every cell once, linear index -> (i, j) column-major like the oracle's
// amo::linear_to_index.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GRID_MAP_ITERATOR_HPP_
#define ORACLE_REFKIT_GRID_MAP_ITERATOR_HPP_

#include <grid_map_core/GridMap.hpp>

namespace grid_map {

class GridMapIterator {
 public:
  explicit GridMapIterator(const GridMap& map)
      : g_(map.geometry()), lin_(0),
        end_(static_cast<size_t>(g_.rows) * static_cast<size_t>(g_.cols)) {
    update();
  }
  bool isPastEnd() const { return lin_ >= end_; }
  GridMapIterator& operator++() {
    ++lin_;
    update();
    return *this;
  }
  const Index& operator*() const { return index_; }

 private:
  void update() {
    if (lin_ < end_) {
      int i, j;
      amo::linear_to_index(g_, lin_, &i, &j);
      index_ = Index(i, j);
    }
  }
  amo_grid g_;
  size_t lin_, end_;
  Index index_;
};

}  // namespace grid_map

#endif  // ORACLE_REFKIT_GRID_MAP_ITERATOR_HPP_
