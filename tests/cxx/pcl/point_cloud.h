// TEST-ONLY stand-in for <pcl/point_cloud.h> (see point_types.h next to it).
#pragma once
#include <cstddef>
#include <utility>
#include <vector>
namespace pcl
{
template <typename PointT> struct PointCloud
{
    std::vector<PointT> points;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); }
    void reserve(std::size_t n) { points.reserve(n); }
    void resize(std::size_t n) { points.resize(n); }  // (pcl::PointCloud::resize also keeps width x height == n)
    void push_back(const PointT &p) { points.push_back(p); }
    template <typename... A> PointT &emplace_back(A &&...a) { return points.emplace_back(std::forward<A>(a)...); }
    const PointT &operator[](std::size_t i) const { return points[i]; }
    PointT &operator[](std::size_t i) { return points[i]; }
    auto begin() const { return points.begin(); }
    auto end() const { return points.end(); }
};
} // namespace pcl
