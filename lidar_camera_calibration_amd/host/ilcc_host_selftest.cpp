// ilcc_host_selftest -- drives ilcc_host::LidarCornersEst exactly like the reference node's per-bag loop
// (/root/reference/ilcc2/test/get_lidar_corners.cpp:130-211: one estimator object, for every bag: setROI ->
// EuclideanCluster -> PCA -> get_corners -> save_corners2txt, members read in between) over several frames with ONE
// estimator, and prints what a test needs to compare with the Python mirror: per frame one line
//   frame <i> ok <0|1> status <s> roi <n> chessboard <n> pca <n> optim <n> corners <n> file <path>
// Usage: ilcc_host_selftest <yaml> <out_prefix> <solver grid|reference> {<cloud.bin> <cx> <cy> <cz>}...
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "LidarCornersEst.h"

using namespace ilcc_host;

static myPointCloudPtr load(const std::string& path) {
  myPointCloudPtr cloud(new myPointCloud);
  std::ifstream in(path, std::ios::binary | std::ios::ate);
  if (!in.is_open()) return cloud;
  const std::streamsize bytes = in.tellg();
  in.seekg(0);
  cloud->resize((size_t)bytes / sizeof(PointXYZI));
  in.read(reinterpret_cast<char*>(cloud->data()), (std::streamsize)(cloud->size() * sizeof(PointXYZI)));
  return cloud;
}

int main(int argc, char** argv) {
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);   // host-side, before the HIP runtime starts (include/ilcc_hip.h)
  if (argc < 8 || (argc - 4) % 4 != 0) {
    std::fprintf(stderr, "usage: %s <yaml> <out_prefix> <grid|reference> {<cloud.bin> <cx> <cy> <cz>}...\n", argv[0]);
    return 2;
  }
  const std::string yaml = argv[1], prefix = argv[2], solver = argv[3];
  LidarCornersEst::Ptr lidar_corners_est(new LidarCornersEst(-1, 200000));     // get_lidar_corners.cpp:112
  lidar_corners_est->register_viewer();                                        // :113
  lidar_corners_est->params().solver = solver == "reference" ? ILCC_SOLVER_REFERENCE_LOCAL : ILCC_SOLVER_GRID;
  if (!lidar_corners_est->set_chessboard_param(yaml)) return 1;                // :114
  const int n_frames = (argc - 4) / 4;
  for (int bag_idx = 1; bag_idx <= n_frames; ++bag_idx) {                      // :130
    const char* const* a = argv + 4 + 4 * (bag_idx - 1);
    myPointCloudPtr pointcloud = load(a[0]);
    PointXYZI clicked_point{(float)std::atof(a[1]), (float)std::atof(a[2]), (float)std::atof(a[3]), 0.f};
    const std::string file = prefix + "_lidar_" + std::to_string(bag_idx) + ".txt";   // :197
    bool ok = false;
    lidar_corners_est->setROI(pointcloud, clicked_point);                      // :183
    size_t n_chess = 0, n_pca = 0, n_optim = 0, n_corner = 0;
    if (lidar_corners_est->EuclideanCluster()) {                               // :188
      lidar_corners_est->PCA();                                                // :191
      n_chess = lidar_corners_est->m_cloud_chessboard->size();                 // :192 (published as /ChessBoard)
      n_pca = lidar_corners_est->m_cloud_PCA->size();                          // :193
      std::vector<std::array<double, 3>> lidar_corner;
      if (lidar_corners_est->get_corners(lidar_corner)) {                      // :194
        ok = save_corners2txt(lidar_corners_est->m_cloud_corners, file);       // :197-198
        n_optim = lidar_corners_est->m_cloud_optim->size();                    // :200
        n_corner = lidar_corners_est->m_cloud_corners->size();                 // :201
        if (n_corner != lidar_corner.size()) return 3;
      }
    }
    std::printf("frame %d ok %d status %d roi %zu chessboard %zu pca %zu optim %zu corners %zu file %s\n", bag_idx, ok ? 1 : 0,
                lidar_corners_est->result().status, lidar_corners_est->m_cloud_ROI->size(), n_chess, n_pca, n_optim, n_corner,
                file.c_str());
  }
  return 0;
}
