// ilcc_calib.h -- consumer closure: the offline part of the reference's
// calib_lidar_cam node that turns our corner files into the LiDAR->camera extrinsic.
//
//   reference                                                         here
//   ---------------------------------------------------------------   -------------------------------
//   ImageCornersEst::read_cam_corners   (src/ImageCornersEst.cpp:213-279)   ilcc_read_cam_corners
//   ImageCornersEst::read_lidar_corners (:281-299)                           ilcc_read_lidar_corners (libilcc_hip)
//   get_lidar2cam_axis_roughly          (test/calib_lidar_cam.cpp:50-69)     ilcc_lidar2cam_axis_roughly
//   check_order_lidar / check_order_cam (src/ImageCornersEst.cpp:430-488)    ilcc_check_order_lidar / _cam
//   Optimization::solvePose3d2dError    (src/Optimization.cpp:13-91,
//                                        include/ilcc2/Optimization.h:126-189) ilcc_solve_pose_3d2d
//   extrinsic2txt / txt2extrinsic       (src/ImageCornersEst.cpp:301-306,352-371) ilcc_extrinsic_write / _read
//   calib_lidar_cam main                (test/calib_lidar_cam.cpp:72-165)    ilcc_calib_lidar_cam
//
// Plain C-ABI, host only: a 210 x 6 dense robust least-squares problem has no data parallelism
// worth a kernel.  All paths are relative to /root/reference/ilcc2/.
#ifndef ILCC_CALIB_H_
#define ILCC_CALIB_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pointgreyN.txt: an X block (one text line per row of the detector's corner matrix) followed by a Y
 * block; returns the number of corners written to out_xy (x0 y0 x1 y1 ...), ordered like
 * read_cam_corners: column-major over the file's matrix unless it has board_h rows. */
int32_t ilcc_read_cam_corners(const char* filename, int32_t num, int32_t board_w, int32_t board_h,
                              double* out_xy);

/* rough axis alignment by camera name (row-major 4x4); returns 0 for an unknown name (identity) */
int32_t ilcc_lidar2cam_axis_roughly(const char* camera_name, double T[16]);

/* in-place row/column flips so that both sensors list the corners in the same order */
void ilcc_check_order_lidar(double* xyz, int32_t board_w, int32_t board_h);
void ilcc_check_order_cam(double* xy, int32_t board_w, int32_t board_h);

/* Ceres-style robust solve of the 6-DoF pose from 3-D/2-D pairs: angle-axis r and translation t,
 * in/out (the reference starts from zero); camera = fx, cx, fy, cy.  Returns iterations used. */
int32_t ilcc_solve_pose_3d2d(const double* pts3d, const double* pts2d, int32_t n, const double camera[4],
                             double r[3], double t[3], double* final_cost);

/* raw Eigen::Matrix4d dump: 16 doubles, column-major, 128 bytes */
int32_t ilcc_extrinsic_write(const char* filename, const double T_rowmajor[16]);
int32_t ilcc_extrinsic_read(const char* filename, double T_rowmajor[16]);

/* the whole node for bag_num corner-file pairs <dir>/<camera>_lidar_<i>.txt + <dir>/<camera><i>.txt:
 * K (fx, cx, fy, cy) given directly.  T_lidar2cam (row-major) out; mean reprojection error (px) out. */
int32_t ilcc_calib_lidar_cam(const char* process_data_dir, const char* camera_name, int32_t bag_num,
                             int32_t board_w, int32_t board_h, const double camera[4], double T_lidar2cam[16],
                             double* mean_reproj_px);

#ifdef __cplusplus
}
#endif
#endif
