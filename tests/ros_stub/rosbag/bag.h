#pragma once
#include <cstdint>
#include <string>
namespace rosbag {
namespace bagmode { enum BagMode { Write = 1, Read = 2, Append = 4 }; }
class Bag {
 public:
  Bag();
  void open(const std::string& filename, uint32_t mode = bagmode::Read);
  void close();
};
}  // namespace rosbag
