// LidarCornersEst.cpp -- see LidarCornersEst.h.  Plumbing only: every computation is a call into
// libilcc_hip.so; there is no CPU path behind these methods.
#include "LidarCornersEst.h"

#include <iostream>

namespace ilcc_host {

LidarCornersEst::LidarCornersEst(int device, uint32_t max_points) : m_device(device), m_max_points(max_points) {
  ilcc_default_params(&m_params);
  m_cloud_ROI.reset(new myPointCloud);
  m_cloud_chessboard.reset(new myPointCloud);
  m_cloud_PCA.reset(new myPointCloud);
  m_cloud_optim.reset(new myPointCloud);
  m_cloud_corners.reset(new myPointCloud);
}

LidarCornersEst::~LidarCornersEst() { ilcc_destroy(m_handle); }

std::string LidarCornersEst::last_error() const { return ilcc_last_error(m_handle); }

bool LidarCornersEst::set_chessboard_param(std::string cam_yaml) {
  if (ilcc_set_chessboard_param(&m_params, cam_yaml.c_str()) != ILCC_OK) {
    std::cerr << "can not open " << cam_yaml << std::endl;
    return false;
  }
  if (m_handle && ilcc_set_params(m_handle, &m_params) != ILCC_OK) return false;
  std::cout << "grid_length: " << m_params.grid_length << std::endl;
  std::cout << "grid_in_x: " << m_params.board_w << std::endl;
  std::cout << "grid_in_y: " << m_params.board_h << std::endl;
  return true;
}

void LidarCornersEst::setROI(myPointCloudPtr cloud, PointXYZI point) {
  m_input = cloud;
  m_click_point = point;
  m_done = false;
}

bool LidarCornersEst::run() {
  if (m_done) return true;
  if (!m_input) return false;
  if (!m_handle) {
    if (ilcc_abi_version() != ILCC_ABI_VERSION) {   // a stale libilcc_hip.so: ilcc_result / ilcc_params would be read at the wrong offsets
      std::cerr << "libilcc_hip.so implements ABI " << ilcc_abi_version() << ", this host was compiled against ABI "
                << ILCC_ABI_VERSION << std::endl;
      return false;
    }
    m_handle = ilcc_create(m_device, &m_params, 1, m_max_points);
    if (!m_handle) {
      std::cerr << "ilcc_create failed: " << ilcc_last_error(nullptr) << std::endl;
      return false;
    }
  }
  const float click[3] = {m_click_point.x, m_click_point.y, m_click_point.z};
  static_assert(sizeof(PointXYZI) == 16, "packed XYZI");
  const int32_t st = ilcc_extract(m_handle, reinterpret_cast<const float*>(m_input->data()),
                                  (uint32_t)m_input->size(), click, &m_result);
  if (st != ILCC_OK) {
    std::cerr << "ilcc_extract: " << ilcc_strerror(st) << " " << ilcc_last_error(m_handle) << std::endl;
    return false;
  }
  m_done = true;
  return true;
}

myPointCloudPtr LidarCornersEst::fetch(int32_t which) {
  myPointCloudPtr out(new myPointCloud);
  const int64_t n = ilcc_fetch_cloud(m_handle, 0, which, nullptr, 0);
  if (n > 0) {
    out->resize((size_t)n);
    ilcc_fetch_cloud(m_handle, 0, which, reinterpret_cast<float*>(out->data()), (uint64_t)n);
  }
  return out;
}

bool LidarCornersEst::EuclideanCluster() {
  if (!run()) return false;
  m_cloud_ROI = fetch(ILCC_CLOUD_ROI);
  if (m_result.status == ILCC_NO_ROI_POINTS || m_result.status == ILCC_NO_CLUSTER ||
      m_result.status == ILCC_NO_PLANE) {
    std::cout << "change /click_point ..." << std::endl;   // LidarCornersEst.cpp:179
    return false;
  }
  m_cloud_chessboard = fetch(ILCC_CLOUD_CHESSBOARD);
  std::cout << "chessboard plane size: " << m_cloud_chessboard->size() << std::endl;   // :184
  return true;
}

void LidarCornersEst::PCA() {
  if (!run()) return;
  m_cloud_PCA = fetch(ILCC_CLOUD_PCA);
  std::cout << "rate: " << m_params.gray_rate << ", gray_zone: " << m_result.gray_zone[0] << " "
            << m_result.gray_zone[1] << std::endl;   // :325
}

bool LidarCornersEst::get_corners(std::vector<std::array<double, 3>>& corners) {
  if (run() && m_result.status == ILCC_AMBIGUOUS && !accept_ambiguous) {
    // what the reference leaves to the human at the viewer (keys 'd' / 'r', :415-437): a board position one square
    // away explains the intensities (almost) equally well -- reject unless the caller asked to keep such scans
    std::cout << "reject this scan (ambiguous: basin margin " << m_result.basin_margin << ")" << std::endl;
    return false;
  }
  if (!run() || (m_result.status != ILCC_OK && m_result.status != ILCC_AMBIGUOUS)) {
    std::cout << "reject this scan" << std::endl;   // :439
    return false;
  }
  if ((m_result.flags & ILCC_FLAG_LOW_COVERAGE) && !accept_low_coverage) {
    // the second half of the operator's look at the viewer: the virtual board must sit ON the points.  With more than
    // 10 % of the squares empty a one-square slip can fit better than the truth (include/ilcc_hip.h "accepting a frame")
    std::cout << "reject this scan (pattern under-sampled: " << m_result.cells_hit << " of "
              << m_params.board_w * m_params.board_h << " squares hold points)" << std::endl;
    return false;
  }
  m_cloud_optim = fetch(ILCC_CLOUD_OPTIM);
  m_cloud_corners.reset(new myPointCloud);
  for (int32_t i = 0; i < m_result.n_corners; ++i) {
    const float* c = m_result.corners + 3 * i;
    m_cloud_corners->push_back(PointXYZI{c[0], c[1], c[2], 50.0f});   // :531
    corners.push_back({(double)c[0], (double)c[1], (double)c[2]});     // cornerCloud2vector :559-570
  }
  return true;
}

bool save_corners2txt(const myPointCloudPtr& cloud, const std::string& filename) {
  std::vector<float> xyz;
  for (const PointXYZI& p : *cloud) {
    xyz.push_back(p.x);
    xyz.push_back(p.y);
    xyz.push_back(p.z);
  }
  return ilcc_save_corners2txt(xyz.data(), (uint32_t)cloud->size(), filename.c_str()) == ILCC_OK;
}

}  // namespace ilcc_host
