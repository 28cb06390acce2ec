// LidarCornersEst.h -- C++ host mirror of the reference class for the corner-extraction path,
// implemented over the C-ABI of libilcc_hip.so (include/ilcc_hip.h).  Same public surface as
// /root/reference/ilcc2/include/ilcc2/LidarCornersEst.h:12-84 as far as the node
// ilcc2/test/get_lidar_corners.cpp:112-201 uses it; the PCL cloud type is replaced by a plain
// vector of the PointXYZI payload so that this header needs neither PCL nor ROS.  A ROS build
// converts with pcl::fromROSMsg / toROSMsg at the node boundary (get_lidar_corners_node.cpp).
#pragma once

#include <array>
#include <memory>
#include <string>
#include <vector>

#include "ilcc_hip.h"

namespace ilcc_host {

struct PointXYZI {   // pcl::PointXYZI payload (ilcc2/include/ilcc2/config.h:8)
  float x, y, z, intensity;
};
typedef std::vector<PointXYZI> myPointCloud;
typedef std::shared_ptr<myPointCloud> myPointCloudPtr;

class LidarCornersEst {
 public:
  typedef std::shared_ptr<LidarCornersEst> Ptr;

  explicit LidarCornersEst(int device = -1, uint32_t max_points = 200000);
  ~LidarCornersEst();
  LidarCornersEst(const LidarCornersEst&) = delete;
  LidarCornersEst& operator=(const LidarCornersEst&) = delete;

  // LidarCornersEst.h:26-37 opened two PCLVisualizer windows; candidates are accepted
  // automatically here (keys 'o' / 'k'), so this only records that the caller asked.
  void register_viewer() {}

  bool set_chessboard_param(std::string cam_yaml);            // LidarCornersEst.cpp:20-46
  void setROI(myPointCloudPtr cloud, PointXYZI point);        // :48-70
  bool EuclideanCluster();                                    // :124-186
  void PCA();                                                 // :366-372
  bool get_corners(std::vector<std::array<double, 3>>& corners);   // :374-450

  // The automatic stand-ins for the operator who would press 'r' at the viewer, one switch per signal:
  // ILCC_AMBIGUOUS scans (GRID solver: a basin one square away costs about the same): get_corners returns false unless
  // accept_ambiguous; ILCC_OK scans flagged ILCC_FLAG_LOW_COVERAGE (fewer than params().min_cell_coverage of the squares
  // hold a labelled point -- a far board on a 16-ring sensor): false unless accept_low_coverage.  result().flags keeps
  // the flag either way, so a caller can also take the decision itself.
  bool accept_ambiguous = false;
  bool accept_low_coverage = false;

  PointXYZI m_click_point{};
  myPointCloudPtr m_cloud_ROI, m_cloud_chessboard, m_cloud_PCA, m_cloud_optim, m_cloud_corners;

  const ilcc_result& result() const { return m_result; }
  ilcc_params& params() { return m_params; }
  std::string last_error() const;

 private:
  bool run();
  myPointCloudPtr fetch(int32_t which);

  ilcc_params m_params{};
  ilcc_handle* m_handle = nullptr;
  ilcc_result m_result{};
  myPointCloudPtr m_input;
  bool m_done = false;
  int m_device;
  uint32_t m_max_points;
};

// get_lidar_corners.cpp:27-36
bool save_corners2txt(const myPointCloudPtr& cloud, const std::string& filename);

}  // namespace ilcc_host
