// get_lidar_corners_node.cpp -- the reference's ROS1 node surface over libilcc_hip.so.
// NOT compiled in this repository's CI: ROS1 / PCL / rosbag are absent from the build image.
// It is what a maintainer drops into ilcc2/test/ in place of get_lidar_corners.cpp (see
// INTEGRATION.md): identical node name, private params, subscribed/advertised topics, frame id,
// bag handling and output file as /root/reference/ilcc2/test/get_lidar_corners.cpp:89-216; only the
// four LidarCornersEst calls go to the GPU and the two viewer confirmations are automatic.
#include <ros/package.h>
#include <ros/ros.h>
#include <rosbag/bag.h>
#include <rosbag/view.h>

#include <boost/foreach.hpp>
#define foreach BOOST_FOREACH

#include <geometry_msgs/PointStamped.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <sensor_msgs/PointCloud2.h>

#include "LidarCornersEst.h"

using ilcc_host::LidarCornersEst;
using ilcc_host::myPointCloudPtr;
using ilcc_host::PointXYZI;

static ros::Publisher pubLaserCloud, pubLaserChessBoard, pubLaserPCA, pubLaserOptim, pubLaserCorners;
static bool received_click_point = false;
static PointXYZI click_point;

static void clickedPointHandler(const geometry_msgs::PointStamped::ConstPtr& msg) {
  std::cout << "received a clicked point: " << msg->point.x << "," << msg->point.y << "," << msg->point.z << std::endl;
  click_point.x = msg->point.x;
  click_point.y = msg->point.y;
  click_point.z = msg->point.z;
  received_click_point = true;
}

static void publish_cloud(ros::Publisher& pub, const myPointCloudPtr& cloud) {
  pcl::PointCloud<pcl::PointXYZI> pc;
  for (const PointXYZI& p : *cloud) {
    pcl::PointXYZI q;
    q.x = p.x; q.y = p.y; q.z = p.z; q.intensity = p.intensity;
    pc.push_back(q);
  }
  sensor_msgs::PointCloud2 msg;
  pcl::toROSMsg(pc, msg);
  msg.header.stamp = ros::Time::now();
  msg.header.frame_id = "/velodyne";
  pub.publish(msg);
}

int main(int argc, char** argv) {
  ros::init(argc, argv, "lidar_corners");
  ros::NodeHandle nh;
  std::string package_path = ros::package::getPath("ilcc2");
  int bag_num;
  std::string bag_path_prefix, lidar_topic, yaml_path, camera_name;
  ros::NodeHandle nh_private("~");
  nh_private.param<std::string>("bag_path_prefix", bag_path_prefix, "20181101_");
  nh_private.param<int>("bag_num", bag_num, 1);
  nh_private.param<std::string>("lidar_topic", lidar_topic, "/velodyne_points");
  nh_private.param<std::string>("camera_name", camera_name, "front");
  nh_private.param<std::string>("yaml_path", yaml_path, "front.yaml");

  LidarCornersEst::Ptr lidar_corners_est(new LidarCornersEst);
  lidar_corners_est->register_viewer();
  lidar_corners_est->set_chessboard_param(package_path + "/config/" + yaml_path);

  ros::Subscriber sub = nh.subscribe<geometry_msgs::PointStamped>("/clicked_point", 100, clickedPointHandler);
  pubLaserCloud = nh.advertise<sensor_msgs::PointCloud2>("/velodyne_points", 10);
  pubLaserChessBoard = nh.advertise<sensor_msgs::PointCloud2>("/ChessBoard", 10);
  pubLaserPCA = nh.advertise<sensor_msgs::PointCloud2>("/pca_cloud", 10);
  pubLaserOptim = nh.advertise<sensor_msgs::PointCloud2>("/Optim_cloud", 10);
  pubLaserCorners = nh.advertise<sensor_msgs::PointCloud2>("/lidar_corners", 10);

  for (int bag_idx = 1; bag_idx <= bag_num; bag_idx++) {
    std::string bag_path = bag_path_prefix + std::to_string(bag_idx) + ".bag";
    rosbag::Bag bag_read;
    bag_read.open(bag_path, rosbag::bagmode::Read);
    std::vector<std::string> topics(1, lidar_topic);
    rosbag::View view(bag_read, rosbag::TopicQuery(topics));
    sensor_msgs::PointCloud2ConstPtr msg_cloud_last = NULL;
    foreach (rosbag::MessageInstance const m, view) {
      sensor_msgs::PointCloud2ConstPtr msg_cloud = m.instantiate<sensor_msgs::PointCloud2>();
      if (msg_cloud != NULL) msg_cloud_last = msg_cloud;
      if (msg_cloud_last != NULL) break;
    }
    bag_read.close();
    if (msg_cloud_last == NULL) {
      ROS_WARN("can't read lidar topic");
      continue;
    }
    pcl::PointCloud<pcl::PointXYZI> pointcloud;
    pcl::fromROSMsg(*msg_cloud_last, pointcloud);
    myPointCloudPtr cloud(new ilcc_host::myPointCloud);
    for (const pcl::PointXYZI& q : pointcloud.points) cloud->push_back(PointXYZI{q.x, q.y, q.z, q.intensity});
    publish_cloud(pubLaserCloud, cloud);

    ROS_INFO_STREAM("please public topic /click_point.....");
    ros::Rate loop_rate(10);
    std::vector<std::array<double, 3>> lidar_corner;
    received_click_point = false;
    while (ros::ok()) {
      if (received_click_point) {
        received_click_point = false;
        lidar_corners_est->setROI(cloud, click_point);
        if (lidar_corners_est->EuclideanCluster()) {
          lidar_corner.clear();
          lidar_corners_est->PCA();
          publish_cloud(pubLaserChessBoard, lidar_corners_est->m_cloud_chessboard);
          publish_cloud(pubLaserPCA, lidar_corners_est->m_cloud_PCA);
          if (lidar_corners_est->get_corners(lidar_corner)) {
            ROS_WARN("add_corner");
            std::string savepath =
                package_path + "/process_data/" + camera_name + "_lidar_" + std::to_string(bag_idx) + ".txt";
            ilcc_host::save_corners2txt(lidar_corners_est->m_cloud_corners, savepath);
            publish_cloud(pubLaserOptim, lidar_corners_est->m_cloud_optim);
            publish_cloud(pubLaserCorners, lidar_corners_est->m_cloud_corners);
          }
          break;
        }
      }
      publish_cloud(pubLaserCloud, cloud);
      ros::spinOnce();
      loop_rate.sleep();
    }
  }
  return 0;
}
