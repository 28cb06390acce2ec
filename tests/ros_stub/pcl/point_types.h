#pragma once
namespace pcl { struct PointXYZI { float x, y, z, intensity; }; }
