#pragma once
#include <pcl/point_cloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
template <typename T> void toROSMsg(const pcl::PointCloud<T>& cloud, sensor_msgs::PointCloud2& msg);
template <typename T> void fromROSMsg(const sensor_msgs::PointCloud2& msg, pcl::PointCloud<T>& cloud);
}  // namespace pcl
