/*
 * ilcc_ingest.h -- the on-disk / on-wire step BEFORE the corner path: what the
 * reference does with rosbag + pcl_conversions in ilcc2/test/get_lidar_corners.cpp:136-164,
 * without ROS.  Implemented in libilcc_hip.so.
 *
 *   reference                                                             here
 *   --------------------------------------------------------------------  ---------------------------
 *   rosbag::Bag::open + View(TopicQuery(topic)) + first message that      ilcc_bag_first_message
 *     instantiates as sensor_msgs/PointCloud2            (:136-155)
 *   ros::serialization of sensor_msgs/PointCloud2 (message layout)        ilcc_pointcloud2_parse
 *   pcl::fromROSMsg(msg, pcl::PointCloud<pcl::PointXYZI>) (:163-164)      ilcc_pointcloud2_unpack_device
 *                                                                         (K0, gfx950 kernel)
 *   all three, host buffers in and out                                    ilcc_bag_first_cloud
 *
 * rosbag and pcl_conversions are third-party and absent from /root/reference; the bag format is the
 * published "ROS bag format 2.0", the message layout is sensor_msgs/PointCloud2.msg (md5
 * 1158d486dd51d683ce2f1be655c3c181), and the field matching is pcl::fromPCLPointCloud2's: a field of
 * the message feeds a PointXYZI member only when name, datatype (FLOAT32) and count (1) all agree;
 * unmatched members stay 0.
 */
#ifndef ILCC_INGEST_H_
#define ILCC_INGEST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILCC_FIELD_ABSENT 0xFFFFFFFFu

typedef struct ilcc_pointcloud2_layout {
  uint32_t height, width;       /* points = height * width */
  uint32_t point_step, row_step;
  uint32_t off_x, off_y, off_z, off_intensity; /* byte offset inside a point, ILCC_FIELD_ABSENT if unmatched */
  uint32_t is_bigendian, is_dense;
  uint32_t stamp_sec, stamp_nsec, seq;
  uint32_t n_fields;
  uint64_t data_offset;         /* of data[] inside the serialized message */
  uint64_t data_bytes;
  char frame_id[64];
} ilcc_pointcloud2_layout;

/* First message (bag time order) on `topic` whose connection carries `md5sum` (NULL: PointCloud2's).
 * Copies the serialized message into msg (cap bytes); *msg_bytes = its size even when cap is too
 * small (ILCC_CAPACITY).  ILCC_IO_ERROR: unreadable / not a V2.0 bag / unindexed / unsupported
 * chunk compression; ILCC_NO_ROI_POINTS is never used here: "no such message" is ILCC_BAD_ARGUMENT
 * with the last-error text "no message of that type on topic" -- the reference prints "can't read lidar topic" and skips the bag. */
int32_t ilcc_bag_first_message(const char* bag_path, const char* topic, const char* md5sum, uint8_t* msg,
                               uint64_t cap, uint64_t* msg_bytes);

/* layout of a serialized sensor_msgs/PointCloud2 */
int32_t ilcc_pointcloud2_parse(const uint8_t* msg, uint64_t msg_bytes, ilcc_pointcloud2_layout* out);

/* K0: data[] (device pointer) -> packed float4 {x, y, z, intensity} (device pointer, height*width
 * records), on hip_stream (a hipStream_t, NULL = default stream); asynchronous. */
int32_t ilcc_pointcloud2_unpack_device(const void* d_data, const ilcc_pointcloud2_layout* layout, void* d_xyzi,
                                       void* hip_stream);

/* bag -> host XYZI (uses device `device` for K0).  *n_points = points in the message even when
 * cap_points is too small (ILCC_CAPACITY). */
int32_t ilcc_bag_first_cloud(int32_t device, const char* bag_path, const char* topic, float* xyzi,
                             uint32_t cap_points, uint32_t* n_points);

#ifdef __cplusplus
}
#endif
#endif
