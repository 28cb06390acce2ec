/*
 * ilcc_project.h -- the step AFTER calibration: per-point LiDAR -> image
 * projection with the calibrated extrinsic.  Implemented in libilcc_hip.so (K8, gfx950 kernels).
 *
 *   reference                                                            here
 *   -------------------------------------------------------------------  -------------------------------
 *   ImageCornersEst::spaceToPlane   (ilcc2/src/ImageCornersEst.cpp:135-155)  both entries, per point
 *   ImageCornersEst::HSVtoRGB       (:373-428)                               ilcc_project_intensity_device
 *   pcd2image processData loop      (ilcc2/test/pcd2image.cpp:40-82)         ilcc_project_intensity_device
 *   rgblidar  processData loop      (ilcc2/test/rgblidar.cpp:45-78)          ilcc_colourise_device
 *
 * Not here: cv::undistort of the image, cv::circle / imshow (display), the ROS subscribers.  Both
 * entries keep the reference's quirks: spaceToPlane accepts P_c.z == 0 (division by zero -> inf/NaN
 * fails the image test), pixel = (int) truncation, rgblidar samples the image it was GIVEN (the
 * reference passes the distorted one, rgblidar.cpp:62-64), pcd2image's fixed colour range 0..60.
 */
#ifndef ILCC_PROJECT_H_
#define ILCC_PROJECT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ilcc_camera_model {
  double R[9];            /* m_R, row-major: lidar -> camera rotation (ImageCornersEst::setRt) */
  double t[3];            /* m_t */
  double fx, cx, fy, cy;  /* m_fx ... from camK */
  int32_t width, height;  /* m_image_size */
} ilcc_camera_model;

/* one projected point: integer pixel, colour, index of the source point */
typedef struct ilcc_pixel_hit {
  int32_t x, y;
  uint8_t r, g, b, pad;
  uint32_t index;
} ilcc_pixel_hit;

/* pcd2image: hits (input order) of the points that pass spaceToPlane, coloured by
 * HSVtoRGB((intensity - inten_low) / (inten_high - inten_low) * 255, 100, 100); the reference fixes
 * inten_low = 0, inten_high = 60.  d_hits has room for n_points records.  Synchronous on return
 * (*n_hits is a host value).  hip_stream: hipStream_t or NULL. */
int32_t ilcc_project_intensity_device(const void* d_xyzi, uint32_t n_points, const ilcc_camera_model* cam,
                                      double distance_valid, double inten_low, double inten_high, void* d_hits,
                                      uint32_t* n_hits, void* hip_stream);

/* rgblidar: XYZRGB cloud (input order) of the points that pass spaceToPlane, colour = the BGR pixel
 * at (int)u, (int)v of d_image_bgr (rows image_step bytes apart).  Output records are 16 bytes:
 * float x, y, z and PCL's packed rgb (uint32 r << 16 | g << 8 | b, stored in the 4th float's bits). */
int32_t ilcc_colourise_device(const void* d_xyzi, uint32_t n_points, const ilcc_camera_model* cam,
                              double distance_valid, const void* d_image_bgr, uint32_t image_step, void* d_xyzrgb,
                              uint32_t* n_out, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif
