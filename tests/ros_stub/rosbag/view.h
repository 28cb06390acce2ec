#pragma once
#include <memory>
#include <string>
#include <vector>
#include <rosbag/bag.h>
namespace rosbag {
class TopicQuery {
 public:
  explicit TopicQuery(const std::vector<std::string>& topics);
};
class MessageInstance {
 public:
  template <class T> std::shared_ptr<const T> instantiate() const;
};
class View {
 public:
  class iterator {
   public:
    iterator& operator++();
    bool operator==(const iterator& o) const;
    bool operator!=(const iterator& o) const;
    MessageInstance* operator->() const;
    MessageInstance& operator*() const;
  };
  View(const Bag& bag, const TopicQuery& query);
  iterator begin();
  iterator end();
};
}  // namespace rosbag
