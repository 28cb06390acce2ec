// ilcc_corners -- ROS-free driver of the get_lidar_corners path for machines without ROS (the GPU
// box): one frame (raw float32 x,y,z,intensity records) + one click -> process_data-format file.
//   ilcc_corners --cloud frame.bin --click x y z --yaml pointgrey.yaml --out pointgrey_lidar_1.txt
//                [--solver grid|reference] [--device N] [--accept-ambiguous] [--accept-low-coverage]
//   ilcc_corners --bag 20181101_1.bag [--topic /velodyne_points] --click x y z ... (first PointCloud2
//                of the bag, read without ROS: include/ilcc_ingest.h)
// Mirrors the per-bag body of /root/reference/ilcc2/test/get_lidar_corners.cpp:133-204.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "LidarCornersEst.h"
#include "ilcc_ingest.h"

using namespace ilcc_host;

int main(int argc, char** argv) {
  // the library keeps up to four batches in flight on four HIP streams: more hardware queues than HIP's default of 4,
  // exported by the HOST before the runtime starts (include/ilcc_hip.h, ilcc_submit_batch_device); never overrides the user
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
  std::string cloud_path, bag_path, topic = "/velodyne_points", yaml_path, out_path, solver = "grid";
  PointXYZI click{0, 0, 0, 0};
  int device = -1;
  bool have_click = false, accept_ambiguous = false, accept_low_coverage = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--cloud" && i + 1 < argc) cloud_path = argv[++i];
    else if (a == "--bag" && i + 1 < argc) bag_path = argv[++i];
    else if (a == "--topic" && i + 1 < argc) topic = argv[++i];
    else if (a == "--yaml" && i + 1 < argc) yaml_path = argv[++i];
    else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
    else if (a == "--solver" && i + 1 < argc) solver = argv[++i];
    else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "--accept-ambiguous") accept_ambiguous = true;
    else if (a == "--accept-low-coverage") accept_low_coverage = true;
    else if (a == "--click" && i + 3 < argc) {
      click.x = (float)std::atof(argv[++i]);
      click.y = (float)std::atof(argv[++i]);
      click.z = (float)std::atof(argv[++i]);
      have_click = true;
    } else {
      std::fprintf(stderr, "unknown or incomplete argument: %s\n", a.c_str());
      return 2;
    }
  }
  if ((cloud_path.empty() == bag_path.empty()) || out_path.empty() || !have_click) {
    std::fprintf(stderr, "usage: ilcc_corners (--cloud frame.bin | --bag file.bag [--topic /velodyne_points]) --click x y z "
                         "[--yaml board.yaml] --out file.txt [--solver grid|reference] [--device N] [--accept-ambiguous] [--accept-low-coverage]\n");
    return 2;
  }
  myPointCloudPtr cloud(new myPointCloud);
  if (!bag_path.empty()) {   // get_lidar_corners.cpp:136-164
    uint32_t n = 0;
    int32_t st = ilcc_bag_first_cloud(device < 0 ? 0 : device, bag_path.c_str(), topic.c_str(), nullptr, 0, &n);
    if (st == ILCC_CAPACITY || st == ILCC_OK) {
      cloud->resize(n);
      st = ilcc_bag_first_cloud(device < 0 ? 0 : device, bag_path.c_str(), topic.c_str(),
                                reinterpret_cast<float*>(cloud->data()), n, &n);
    }
    if (st != ILCC_OK) {
      std::fprintf(stderr, "can't read lidar topic: %s\n", ilcc_last_error(nullptr));   // :157-161
      return 1;
    }
  } else {
    std::ifstream in(cloud_path, std::ios::binary | std::ios::ate);
    if (!in.is_open()) {
      std::fprintf(stderr, "can not open %s\n", cloud_path.c_str());
      return 1;
    }
    const std::streamsize bytes = in.tellg();
    in.seekg(0);
    cloud->resize((size_t)bytes / sizeof(PointXYZI));
    in.read(reinterpret_cast<char*>(cloud->data()), (std::streamsize)(cloud->size() * sizeof(PointXYZI)));
  }

  LidarCornersEst est(device, (uint32_t)cloud->size() + 1);
  est.params().solver = (solver == "reference") ? ILCC_SOLVER_REFERENCE_LOCAL : ILCC_SOLVER_GRID;
  est.accept_ambiguous = accept_ambiguous;
  est.accept_low_coverage = accept_low_coverage;
  est.register_viewer();
  if (!yaml_path.empty() && !est.set_chessboard_param(yaml_path)) return 1;

  est.setROI(cloud, click);                          // get_lidar_corners.cpp:183
  if (!est.EuclideanCluster()) return 3;             // :188
  est.PCA();                                         // :191
  std::vector<std::array<double, 3>> lidar_corner;
  if (!est.get_corners(lidar_corner)) return 4;      // :194
  std::cout << "add_corner" << std::endl;            // :196
  if (!save_corners2txt(est.m_cloud_corners, out_path)) {   // :197-198
    std::fprintf(stderr, "can not write %s\n", out_path.c_str());
    return 1;
  }
  const ilcc_result& r = est.result();
  std::printf("theta_t %.6f %.6f %.6f phase %d cost %.6f margin %.3f status %d corners %d -> %s\n", r.theta_t[0], r.theta_t[1],
              r.theta_t[2], r.phase, r.sel_cost, r.basin_margin, r.status, r.n_corners, out_path.c_str());
  return 0;
}
