// The four typedefs of aerial-mapper-io.h:17-20 that appear in
// ortho::OrthoBackwardGrid's signature.  (The file/pose/image loaders of
// io::AerialMapperIO are outside the hot path and stay the reference's.)
#ifndef AERIAL_MAPPER_HIP_IO_TYPES_H_
#define AERIAL_MAPPER_HIP_IO_TYPES_H_

#include <vector>

#include "aerial-mapper-deps.h"
#include "aerial-mapper-utils/utils-nearest-neighbor.h"

typedef kindr::minimal::QuatTransformation Pose;
typedef std::vector<Pose> Poses;
typedef cv::Mat Image;
typedef std::vector<Image> Images;

#endif  // AERIAL_MAPPER_HIP_IO_TYPES_H_
