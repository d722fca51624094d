// AlignedType: the container typedef of the point-cloud argument of
// dsm::Dsm::process (reference: aerial_mapper_utils/include/
// aerial-mapper-utils/utils-nearest-neighbor.h:18-21).  The kd-tree adaptor
// that shares the reference's header is not needed on the GPU path.
#ifndef AERIAL_MAPPER_HIP_UTILS_NEAREST_NEIGHBOR_H_
#define AERIAL_MAPPER_HIP_UTILS_NEAREST_NEIGHBOR_H_

#include "aerial-mapper-deps.h"

template <template <typename, typename> class Container, typename Type>
struct AlignedType {
  typedef Container<Type, Eigen::aligned_allocator<Type> > type;
};

#endif  // AERIAL_MAPPER_HIP_UTILS_NEAREST_NEIGHBOR_H_
