// get_lidar_corners_node.cpp -- ROS1 front end for libilcc_hip.so with the external surface of the
// reference's get_lidar_corners node (/root/reference/ilcc2/test/get_lidar_corners.cpp:89-216,
// launch/lidar_corners.launch:4-16), so that calib_lidar_cam.launch consumes its output unchanged:
//
//   node name            lidar_corners (ros::init), launched as type get_lidar_corners
//   private parameters   ~bag_path_prefix ("20181101_"), ~bag_num (1), ~lidar_topic
//                        ("/velodyne_points"), ~camera_name ("front"), ~yaml_path ("front.yaml")
//                        added (the operator's 'r' key made automatic, one switch per signal, both default false):
//                        ~accept_ambiguous, ~accept_low_coverage
//   subscribes           /clicked_point  geometry_msgs/PointStamped, queue 100
//   advertises           /velodyne_points /ChessBoard /pca_cloud /Optim_cloud /lidar_corners
//                        sensor_msgs/PointCloud2, queue 10, frame_id "/velodyne"
//   reads                the first PointCloud2 on ~lidar_topic of <prefix><idx>.bag, idx = 1..bag_num
//   writes               <ilcc2 package>/process_data/<camera_name>_lidar_<idx>.txt
//
// Not BUILT in this repository (ROS1, rosbag and PCL are absent from the image) but type-checked on every test run against
// declaration-only headers (tests/ros_stub/, tests/test_host_logic.py::test_ros_node_source_type_checks); it is the
// file a maintainer adds to the reference's catkin package, see INTEGRATION.md.  All computation is
// one call chain into the C-ABI through ilcc_host::LidarCornersEst; the two PCL-viewer
// confirmations of the reference are automatic.
#include <geometry_msgs/PointStamped.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <ros/package.h>
#include <ros/ros.h>
#include <rosbag/bag.h>
#include <rosbag/view.h>
#include <sensor_msgs/PointCloud2.h>

#include <array>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "LidarCornersEst.h"

namespace {

using ilcc_host::myPointCloud;
using ilcc_host::myPointCloudPtr;
using ilcc_host::PointXYZI;

struct NodeConfig {
  std::string bag_prefix = "20181101_";
  int bag_count = 1;
  std::string lidar_topic = "/velodyne_points";
  std::string camera = "front";
  std::string yaml = "front.yaml";
  std::string package_dir;
  bool accept_ambiguous = false, accept_low_coverage = false;

  void load(ros::NodeHandle& priv) {
    priv.param<bool>("accept_ambiguous", accept_ambiguous, accept_ambiguous);
    priv.param<bool>("accept_low_coverage", accept_low_coverage, accept_low_coverage);
    priv.param<std::string>("bag_path_prefix", bag_prefix, bag_prefix);
    priv.param<int>("bag_num", bag_count, bag_count);
    priv.param<std::string>("lidar_topic", lidar_topic, lidar_topic);
    priv.param<std::string>("camera_name", camera, camera);
    priv.param<std::string>("yaml_path", yaml, yaml);
    package_dir = ros::package::getPath("ilcc2");
  }
  std::string bag_file(int idx) const { return bag_prefix + std::to_string(idx) + ".bag"; }
  std::string corner_file(int idx) const {
    return package_dir + "/process_data/" + camera + "_lidar_" + std::to_string(idx) + ".txt";
  }
};

// PCL <-> packed XYZI at the node boundary only
sensor_msgs::PointCloud2 to_message(const myPointCloud& cloud) {
  pcl::PointCloud<pcl::PointXYZI> pc;
  pc.reserve(cloud.size());
  for (size_t k = 0; k < cloud.size(); ++k) {
    pcl::PointXYZI q;
    q.x = cloud[k].x;
    q.y = cloud[k].y;
    q.z = cloud[k].z;
    q.intensity = cloud[k].intensity;
    pc.push_back(q);
  }
  sensor_msgs::PointCloud2 msg;
  pcl::toROSMsg(pc, msg);
  msg.header.frame_id = "/velodyne";
  msg.header.stamp = ros::Time::now();
  return msg;
}

myPointCloudPtr from_message(const sensor_msgs::PointCloud2& msg) {
  pcl::PointCloud<pcl::PointXYZI> pc;
  pcl::fromROSMsg(msg, pc);
  myPointCloudPtr out(new myPointCloud);
  out->reserve(pc.size());
  for (size_t k = 0; k < pc.size(); ++k) out->push_back(PointXYZI{pc[k].x, pc[k].y, pc[k].z, pc[k].intensity});
  return out;
}

// first PointCloud2 of the bag on the given topic, or null
sensor_msgs::PointCloud2ConstPtr first_cloud_of(const std::string& bag_path, const std::string& topic) {
  rosbag::Bag bag;
  bag.open(bag_path, rosbag::bagmode::Read);
  rosbag::View view(bag, rosbag::TopicQuery(std::vector<std::string>(1, topic)));
  sensor_msgs::PointCloud2ConstPtr found;
  for (rosbag::View::iterator it = view.begin(); it != view.end() && !found; ++it)
    found = it->instantiate<sensor_msgs::PointCloud2>();
  bag.close();
  return found;
}

class CornerNode {
 public:
  explicit CornerNode(ros::NodeHandle& nh) {
    ros::NodeHandle priv("~");
    cfg_.load(priv);
    ROS_INFO("ilcc2 package at %s, %d bag(s)", cfg_.package_dir.c_str(), cfg_.bag_count);
    estimator_.register_viewer();
    estimator_.accept_ambiguous = cfg_.accept_ambiguous;
    estimator_.accept_low_coverage = cfg_.accept_low_coverage;
    estimator_.set_chessboard_param(cfg_.package_dir + "/config/" + cfg_.yaml);
    click_sub_ = nh.subscribe<geometry_msgs::PointStamped>("/clicked_point", 100, &CornerNode::on_click, this);
    const char* topics[] = {"/velodyne_points", "/ChessBoard", "/pca_cloud", "/Optim_cloud", "/lidar_corners"};
    for (const char* t : topics) pubs_[t] = nh.advertise<sensor_msgs::PointCloud2>(t, 10);
  }

  void run() {
    for (int idx = 1; idx <= cfg_.bag_count && ros::ok(); ++idx) handle_bag(idx);
  }

 private:
  void on_click(const geometry_msgs::PointStamped::ConstPtr& msg) {
    click_ = PointXYZI{(float)msg->point.x, (float)msg->point.y, (float)msg->point.z, 0.f};
    have_click_ = true;
    ROS_INFO("clicked point %.3f %.3f %.3f", click_.x, click_.y, click_.z);
  }

  void publish(const char* topic, const myPointCloudPtr& cloud) { pubs_[topic].publish(to_message(*cloud)); }

  // one bag: show its first cloud, wait for a click that yields a board, write the corner file
  void handle_bag(int idx) {
    const sensor_msgs::PointCloud2ConstPtr msg = first_cloud_of(cfg_.bag_file(idx), cfg_.lidar_topic);
    if (!msg) {
      ROS_WARN("can't read lidar topic in %s", cfg_.bag_file(idx).c_str());
      return;
    }
    const myPointCloudPtr cloud = from_message(*msg);
    ROS_INFO_STREAM("bag " << idx << ": " << cloud->size() << " points; publish /clicked_point on the board");
    have_click_ = false;
    ros::Rate rate(10);
    bool done = false;
    while (ros::ok() && !done) {
      publish("/velodyne_points", cloud);
      ros::spinOnce();
      if (have_click_) {
        have_click_ = false;
        done = try_click(idx, cloud);
      }
      rate.sleep();
    }
  }

  // the reference's per-click sequence: setROI -> EuclideanCluster -> PCA -> get_corners -> file
  bool try_click(int idx, const myPointCloudPtr& cloud) {
    estimator_.setROI(cloud, click_);
    if (!estimator_.EuclideanCluster()) return false;   // keep waiting for a better click
    estimator_.PCA();
    publish("/ChessBoard", estimator_.m_cloud_chessboard);
    publish("/pca_cloud", estimator_.m_cloud_PCA);
    std::vector<std::array<double, 3>> corners;
    if (estimator_.get_corners(corners)) {
      ROS_WARN("add_corner");
      ilcc_host::save_corners2txt(estimator_.m_cloud_corners, cfg_.corner_file(idx));
      publish("/Optim_cloud", estimator_.m_cloud_optim);
      publish("/lidar_corners", estimator_.m_cloud_corners);
    }
    return true;   // like the reference, one accepted cluster ends the bag whether or not corners came out
  }

  NodeConfig cfg_;
  ilcc_host::LidarCornersEst estimator_;
  ros::Subscriber click_sub_;
  std::map<std::string, ros::Publisher> pubs_;
  PointXYZI click_{};
  bool have_click_ = false;
};

}  // namespace

int main(int argc, char** argv) {
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);   // host-side, before the HIP runtime starts (include/ilcc_hip.h)
  ros::init(argc, argv, "lidar_corners");
  ros::NodeHandle nh;
  CornerNode node(nh);
  node.run();
  return 0;
}
