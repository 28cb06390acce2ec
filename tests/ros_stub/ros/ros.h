// declaration-only subset of roscpp for -fsyntax-only checks (see ../README.md)
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <sstream>
#include <string>
namespace ros {
void init(int& argc, char** argv, const std::string& name, uint32_t options = 0);
bool ok();
void spinOnce();
struct Time {
  static Time now();
};
class Rate {
 public:
  explicit Rate(double hz);
  bool sleep();
};
class Publisher {
 public:
  template <typename M>
  void publish(const M& message) const;
};
class Subscriber {};
class NodeHandle {
 public:
  explicit NodeHandle(const std::string& ns = std::string());
  template <typename T>
  bool param(const std::string& name, T& value, const T& default_value) const;
  template <typename M, typename T>
  Subscriber subscribe(const std::string& topic, uint32_t queue, void (T::*callback)(const std::shared_ptr<const M>&), T* object);
  template <typename M>
  Publisher advertise(const std::string& topic, uint32_t queue, bool latch = false);
};
}  // namespace ros
#define ROS_INFO(...) ((void)std::printf(__VA_ARGS__))
#define ROS_WARN(...) ((void)std::printf(__VA_ARGS__))
#define ROS_INFO_STREAM(args) do { std::ostringstream ros_stub_ss_; ros_stub_ss_ << args; } while (0)
