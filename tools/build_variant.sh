# build a variant of libilcc_hip.so with extra compile-time knobs into build/ab/ (selected through ILCC_HIP_LIB):
#   tools/build_variant.sh NAME -DILCC_K6_TIMING [...]
set -e
NAME=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
D=$R/build/ab/obj_$NAME
mkdir -p $D
cd $R/lidar_camera_calibration_amd/csrc
COMMON="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -I../../include -I. -Wall -Wno-unused-function $@"
for f in k0_unpack_pointcloud2 k1_roi_crop k2_cluster k3_ransac_plane k45_plane_frame_hist k7_refine_corners k8_project; do
  /opt/rocm/bin/hipcc $COMMON -ffp-contract=off -c $f.hip -o $D/$f.o &
done
/opt/rocm/bin/hipcc $COMMON -fno-honor-nans -c k6_grid_cost.hip -o $D/k6_grid_cost.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off -x hip -c ilcc_api.cpp -o $D/ilcc_api.o &
/opt/rocm/bin/hipcc $COMMON -x hip -c bag_reader.cpp -o $D/bag_reader.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/ab/libilcc_hip_$NAME.so $D/*.o -ldl
echo built $R/build/ab/libilcc_hip_$NAME.so
