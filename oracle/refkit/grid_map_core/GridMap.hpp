// oracle/refkit: stand-in for <grid_map_core/GridMap.hpp> (see ../refkit.h): a layer
// container with the geometry calls the three files make.  The arithmetic of
// setGeometry / getPosition / colorVectorToValue is the oracle's adopted definition
// (../../amo_compat.h) -- NOT pinned by this build.  TEST INFRASTRUCTURE ONLY.
#ifndef ORACLE_REFKIT_GRID_MAP_CORE_GRIDMAP_HPP_
#define ORACLE_REFKIT_GRID_MAP_CORE_GRIDMAP_HPP_

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include <Eigen/Core>

#include "../../amo_compat.h"

namespace grid_map {

typedef Eigen::MatrixXf Matrix;
typedef Eigen::Vector2d Position;
typedef Eigen::Array2d Length;
typedef Eigen::Array2i Index;
typedef Eigen::Array2i Size;

class GridMap {
 public:
  GridMap() { std::memset(&g_, 0, sizeof(g_)); }
  explicit GridMap(const std::vector<std::string>& layers) {
    for (const std::string& name : layers) data_[name] = Matrix();
    std::memset(&g_, 0, sizeof(g_));
  }
  void setFrameId(const std::string& frame) { frame_ = frame; }
  const std::string& getFrameId() const { return frame_; }
  void setTimestamp(uint64_t) {}
  Length getLength() const { return Length(g_.length_x, g_.length_y); }
  Position getPosition() const { return Position(g_.pos_x, g_.pos_y); }
  bool exists(const std::string& layer) const { return data_.count(layer) != 0; }
  void setGeometry(const Length& length, double resolution, const Position& position) {
    g_ = amo::make_grid(length(0), length(1), resolution, position(0), position(1));
    size_ = Size(g_.rows, g_.cols);
    for (auto& kv : data_) kv.second.resize(g_.rows, g_.cols);
  }
  const Size& getSize() const { return size_; }
  double getResolution() const { return g_.resolution; }
  Index getStartIndex() const { return Index(0, 0); }  // (never moved)
  const amo_grid& geometry() const { return g_; }
  Matrix& operator[](const std::string& layer) { return get(layer); }
  const Matrix& operator[](const std::string& layer) const { return get(layer); }
  Matrix& get(const std::string& layer) {
    auto it = data_.find(layer);
    if (it == data_.end()) throw std::out_of_range("no layer '" + layer + "'");
    return it->second;
  }
  const Matrix& get(const std::string& layer) const {
    auto it = data_.find(layer);
    if (it == data_.end()) throw std::out_of_range("no layer '" + layer + "'");
    return it->second;
  }
  float& at(const std::string& layer, const Index& index) { return get(layer)(index(0), index(1)); }
  bool getPosition(const Index& index, Position& position) const {
    if (index(0) < 0 || index(1) < 0 || index(0) >= g_.rows || index(1) >= g_.cols) return false;
    amo::cell_position(g_, index(0), index(1), &position(0), &position(1));
    return true;
  }

 private:
  std::map<std::string, Matrix> data_;
  std::string frame_;
  amo_grid g_;
  Size size_;
};

inline bool colorVectorToValue(const Eigen::Vector3f& color, float& value) {
  value = amo::color_vector_to_value(color(0), color(1), color(2));
  return true;
}

}  // namespace grid_map

#endif  // ORACLE_REFKIT_GRID_MAP_CORE_GRIDMAP_HPP_
