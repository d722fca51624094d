// grid_map_lite.h -- stand-in for grid_map_core's GridMap (external dependency
// of the reference, absent from this image): layer container + geometry with
// the member names the hot path and the demos call
// (aerial-mapper-grid-map.cc:25-48, dsm.cc:28,116,125,
// ortho-backward-grid.cc:48-58).  Arithmetic = the formulas adopted in
// SURVEY.md section 8c.
#ifndef AERIAL_MAPPER_COMPAT_GRID_MAP_LITE_H_
#define AERIAL_MAPPER_COMPAT_GRID_MAP_LITE_H_

#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "aerial-mapper-compat/eigen_lite.h"

namespace grid_map {

typedef Eigen::MatrixXf Matrix;
typedef Eigen::Vector2d Position;
typedef Eigen::Array2d Length;
typedef Eigen::Array2i Index;
typedef Eigen::Array2i Size;

class GridMap {
 public:
  GridMap() : resolution_(0.0) {}
  explicit GridMap(const std::vector<std::string>& layers) : resolution_(0.0) {
    for (const std::string& l : layers) data_[l] = Matrix();
    layers_ = layers;
  }
  void setFrameId(const std::string& id) { frame_ = id; }
  const std::string& getFrameId() const { return frame_; }
  void setTimestamp(unsigned long long) {}

  void setGeometry(const Length& length, const double resolution,
                   const Position& position) {
    size_(0) = static_cast<int>(std::round(length(0) / resolution));
    size_(1) = static_cast<int>(std::round(length(1) / resolution));
    for (auto& kv : data_) kv.second.resize(size_(0), size_(1));
    resolution_ = resolution;
    length_(0) = static_cast<double>(size_(0)) * resolution_;
    length_(1) = static_cast<double>(size_(1)) * resolution_;
    position_ = position;
  }
  const Size& getSize() const { return size_; }
  double getResolution() const { return resolution_; }
  const Length& getLength() const { return length_; }
  const Position& getPosition() const { return position_; }
  Index getStartIndex() const { return Index(0, 0); }

  bool exists(const std::string& layer) const { return data_.count(layer) != 0; }
  const std::vector<std::string>& getLayers() const { return layers_; }
  void add(const std::string& layer, float value = NAN) {
    if (!exists(layer)) layers_.push_back(layer);
    data_[layer].resize(size_(0), size_(1));
    data_[layer].setConstant(value);
  }
  Matrix& operator[](const std::string& layer) { return get(layer); }
  const Matrix& operator[](const std::string& layer) const { return get(layer); }
  Matrix& get(const std::string& layer) {
    auto it = data_.find(layer);
    if (it == data_.end()) throw std::out_of_range("GridMap::get(...) : No map layer '" + layer + "'");
    return it->second;
  }
  const Matrix& get(const std::string& layer) const {
    auto it = data_.find(layer);
    if (it == data_.end()) throw std::out_of_range("GridMap::get(...) : No map layer '" + layer + "'");
    return it->second;
  }
  float& at(const std::string& layer, const Index& index) {
    return get(layer)(index(0), index(1));
  }

  bool getPosition(const Index& index, Position& position) const {
    if (index(0) < 0 || index(1) < 0 || index(0) >= size_(0) || index(1) >= size_(1))
      return false;
    const double off_x = 0.5 * length_(0) - 0.5 * resolution_;
    const double off_y = 0.5 * length_(1) - 0.5 * resolution_;
    position(0) = (position_(0) + off_x) + resolution_ * (-static_cast<double>(index(0)));
    position(1) = (position_(1) + off_y) + resolution_ * (-static_cast<double>(index(1)));
    return true;
  }

 private:
  std::map<std::string, Matrix> data_;
  std::vector<std::string> layers_;
  std::string frame_;
  Size size_;
  Length length_;
  Position position_;
  double resolution_;
};

}  // namespace grid_map

#endif  // AERIAL_MAPPER_COMPAT_GRID_MAP_LITE_H_
