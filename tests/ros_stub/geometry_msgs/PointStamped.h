#pragma once
#include <memory>
#include <sensor_msgs/PointCloud2.h>
namespace geometry_msgs {
struct Point { double x, y, z; };
struct PointStamped {
  std_msgs::Header header;
  Point point;
  typedef std::shared_ptr<const PointStamped> ConstPtr;
};
}  // namespace geometry_msgs
