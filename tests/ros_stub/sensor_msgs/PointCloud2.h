#pragma once
#include <memory>
#include <string>
#include <ros/ros.h>
namespace std_msgs { struct Header { uint32_t seq; ros::Time stamp; std::string frame_id; }; }
namespace sensor_msgs {
struct PointCloud2 {
  std_msgs::Header header;
  typedef std::shared_ptr<PointCloud2> Ptr;
  typedef std::shared_ptr<const PointCloud2> ConstPtr;
};
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}  // namespace sensor_msgs
