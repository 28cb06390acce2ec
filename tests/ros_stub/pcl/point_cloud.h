#pragma once
#include <cstddef>
namespace pcl {
template <typename PointT>
class PointCloud {
 public:
  void reserve(std::size_t n);
  void push_back(const PointT& p);
  std::size_t size() const;
  const PointT& operator[](std::size_t k) const;
  PointT& operator[](std::size_t k);
};
}  // namespace pcl
