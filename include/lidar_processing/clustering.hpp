// clustering.hpp -- drop-in replacement of the reference's src/clustering.hpp.
//
// Same names, namespace, defaults, sentinels and call surface as the reference
// (src/clustering.hpp:40-90), so src/processor.cpp compiles against it unchanged
// (#include "clustering.hpp", :27; member :132; call :178; sentinels used at :186-190).
// cluster() works for every PCL point type whose record starts with float x, y, z
// (the reference instantiates PointXYZ, PointXYZI, PointXYZL, PointXYZRGB, PointXYZRGBL).
#ifndef LIDAR_PROCESSING__CLUSTERING_HPP
#define LIDAR_PROCESSING__CLUSTERING_HPP

#include "lpx_context.hpp"

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <cmath>
#include <cstdint>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace lidar_processing
{
using ClusteringLabel = std::int32_t;

struct ClusteringConfiguration final
{
    float distance_squared{0.18F};
    float cluster_quality{0.5F};
    std::uint32_t min_cluster_size{4U};
    std::uint32_t max_cluster_size{std::numeric_limits<std::uint32_t>::max()};
};

class Clusterer final
{
  public:
    static constexpr ClusteringLabel UNDEFINED{std::numeric_limits<std::int32_t>::lowest()};
    static constexpr ClusteringLabel INVALID{-1};

    // the process-wide context, shared with a default-constructed Segmenter (lpx_context.hpp: default_context): the
    // obstacle cloud segment() left on the device is clustered where it lies
    Clusterer() : context_{detail::default_context()}, configuration_{}
    {
        reserve_memory();
    }
    explicit Clusterer(std::shared_ptr<detail::LpxContext> context) : context_{std::move(context)}, configuration_{}
    {
        reserve_memory();
    }
    ~Clusterer() = default;

    void update_configuration(const ClusteringConfiguration &configuration)
    {
        configuration_ = configuration;
    }

    void reserve_memory(std::uint32_t number_of_points = 200'000U)
    {
        std::lock_guard<std::mutex> lock(context_->mutex());
        lpx_reserve(context_->get(), number_of_points, 0U);
    }

    // Like the reference's, this call does not fail on a non-empty cloud (src/clustering.cpp:47-125: the only throw on
    // its path, KDTree::rebuild's, is unreachable after the empty check at :51-54).  Should the device report an error
    // (workspace, a transient HIP status), the call is repeated once on a re-reserved workspace; if that fails too the
    // object degrades in the SAFE direction, the one the reference's Segmenter takes when a plane fit fails
    // (src/segmentation.cpp:251-259: everything is an obstacle): a line on std::cerr and a defined result -- every point
    // in ONE cluster, label 0, so the unchanged caller (src/processor.cpp:180-200) publishes the whole obstacle cloud as
    // one obstacle instead of "nothing there" -- and failed() says so.  Never an exception one layer above the C-ABI,
    // never UNDEFINED, never an empty obstacle set for a non-empty cloud.
    template <typename PointT>
    void cluster(const pcl::PointCloud<PointT> &cloud_in, std::vector<ClusteringLabel> &labels)
    {
        labels.assign(cloud_in.size(), UNDEFINED);
        last_size_ = 0U;
        last_clusters_ = 0U;
        last_epoch_ = 0U;
        failed_ = false;
        if (cloud_in.empty())
        {
            return;
        }
        std::lock_guard<std::mutex> lock(context_->mutex());
        context_->cluster_owner = nullptr;

        lpx_clu_cfg cfg{};
        cfg.distance_squared = configuration_.distance_squared;
        cfg.cluster_quality = configuration_.cluster_quality;
        cfg.min_cluster_size = configuration_.min_cluster_size;
        cfg.max_cluster_size = configuration_.max_cluster_size;

        const std::uint32_t number_of_points = static_cast<std::uint32_t>(cloud_in.size());
        std::uint32_t number_of_clusters = 0U;
        int rc = lpx_cluster(context_->get(), detail::points_base(cloud_in.points.data()), sizeof(PointT),
                             number_of_points, &cfg, labels.data(), &number_of_clusters);
        if (rc != LPX_OK && rc != LPX_ERR_ARG && rc != LPX_ERR_RANGE)
        {
            std::cerr << "Clustering: " << lpx_last_error(context_->get()) << " -- retrying once" << std::endl;
            lpx_reserve(context_->get(), number_of_points > 100'000U ? 2U * number_of_points : 200'000U, 0U);
            labels.assign(cloud_in.size(), UNDEFINED);
            rc = lpx_cluster(context_->get(), detail::points_base(cloud_in.points.data()), sizeof(PointT),
                             number_of_points, &cfg, labels.data(), &number_of_clusters);
        }
        if (rc != LPX_OK)
        {
            std::cerr << "Failed clustering: " << lpx_last_error(context_->get())
                      << " -- the whole cloud is reported as one cluster" << std::endl;
            labels.assign(cloud_in.size(), ClusteringLabel{0});
            failed_ = true;
            last_size_ = number_of_points;
            return;
        }
        last_size_ = number_of_points;
        last_clusters_ = number_of_clusters;
        last_epoch_ = lpx_cluster_epoch(context_->get());
        context_->cluster_owner = this;
    }

    // true when the last cluster() call could not be served by the device even after its retry and returned the
    // degraded result (every point in cluster 0)
    bool failed() const noexcept
    {
        return failed_;
    }

    // Optional fast path for the regrouping the reference's caller does right after cluster()
    // (src/processor.cpp:180-200): one cloud per valid cluster, clusters in label order, points in
    // index order, INVALID dropped.  The grouping is computed on the device from the labels of the last
    // cluster() call; only the points are gathered here.  `cloud_in` must be the cloud just clustered, and nothing else
    // may have used the (possibly shared) context since: call it right after cluster(), as processor.cpp does its own
    // regrouping -- otherwise it throws (check_resident).
    template <typename PointT, typename PointOutT>
    void regroup(const pcl::PointCloud<PointT> &cloud_in, std::vector<pcl::PointCloud<PointOutT>> &clustered_cloud)
    {
        clustered_cloud.clear();
        if (failed_ && cloud_in.size() == last_size_)
        {
            // the degraded result of cluster(): one group holding the whole cloud, like the caller's own loop builds it
            clustered_cloud.resize(1U);
            clustered_cloud[0].reserve(cloud_in.size());
            for (const auto &point : cloud_in.points)
            {
                clustered_cloud[0].emplace_back(point.x, point.y, point.z);
            }
            return;
        }
        if (last_clusters_ == 0U || cloud_in.size() != last_size_)
        {
            return;
        }
        group_offsets_.resize(last_clusters_ + 1U);
        group_indices_.resize(last_size_);
        std::uint32_t number_of_valid = 0U;
        std::lock_guard<std::mutex> lock(context_->mutex());
        check_resident("cluster regrouping");
        const int rc = lpx_cluster_groups(context_->get(), last_size_, last_clusters_, group_offsets_.data(),
                                          group_indices_.data(), &number_of_valid);
        if (rc != LPX_OK)
        {
            throw std::runtime_error(std::string("cluster regrouping failed: ") + lpx_last_error(context_->get()));
        }
        clustered_cloud.resize(last_clusters_);
        for (std::uint32_t c = 0U; c < last_clusters_; ++c)
        {
            auto &cloud = clustered_cloud[c];
            cloud.reserve(group_offsets_[c + 1U] - group_offsets_[c]);
            for (std::uint32_t p = group_offsets_[c]; p < group_offsets_[c + 1U]; ++p)
            {
                const auto &point = cloud_in.points[group_indices_[p]];
                cloud.emplace_back(point.x, point.y, point.z);
            }
        }
    }

    // Optional fast path for the convex branch of findOrderedConcaveOutlines (src/polygon_simplification.cpp:96-115):
    // one counter-clockwise outline per valid cluster with fewer than `max_points` points (20 in the reference),
    // clusters in label order, computed on the device for the labels of the last cluster() call.  Larger
    // clusters are skipped here (the reference sends them to its concave-hull submodule).  PointOutT needs
    // public x and y members and a (x, y) constructor, like geom::Point<float>.
    template <typename PointOutT>
    void convex_outlines(std::vector<std::vector<PointOutT>> &outlines, std::uint32_t max_points = 20U)
    {
        outlines.clear();
        if (last_clusters_ == 0U)
        {
            return;
        }
        hull_offsets_.resize(last_clusters_ + 1U);
        hull_indices_.resize(last_size_);
        hull_xy_.resize(2U * static_cast<std::size_t>(last_size_));
        std::uint32_t number_of_hull_points = 0U;
        std::lock_guard<std::mutex> lock(context_->mutex());
        check_resident("convex outlines");
        const int rc = lpx_cluster_hulls(context_->get(), last_size_, last_clusters_, max_points, hull_offsets_.data(),
                                         hull_indices_.data(), hull_xy_.data(), &number_of_hull_points);
        if (rc != LPX_OK)
        {
            throw std::runtime_error(std::string("convex outlines failed: ") + lpx_last_error(context_->get()));
        }
        for (std::uint32_t c = 0U; c < last_clusters_; ++c)
        {
            if (hull_offsets_[c + 1U] == hull_offsets_[c])
            {
                continue;
            }
            std::vector<PointOutT> outline;
            outline.reserve(hull_offsets_[c + 1U] - hull_offsets_[c]);
            for (std::uint32_t p = hull_offsets_[c]; p < hull_offsets_[c + 1U]; ++p)
            {
                outline.emplace_back(hull_xy_[2U * p], hull_xy_[2U * p + 1U]);
            }
            outlines.push_back(std::move(outline));
        }
    }

  private:
    // regroup() / convex_outlines() read the labels cluster() left on the device.  On a shared context (the default) any
    // call of another object in between -- a Segmenter::segment, whose look-ahead clusters the NEXT cloud, or another
    // Clusterer -- replaces them: the context's owner mark and the library's clustering epoch (0 once anything else has
    // run) must still be this object's, or the caller would silently get groups of a different cloud.
    void check_resident(const char *what) const
    {
        if (context_->cluster_owner != this || lpx_cluster_epoch(context_->get()) != last_epoch_)
        {
            throw std::runtime_error(std::string(what) +
                                     " failed: the shared context has been used since this object's cluster() call");
        }
    }

    std::shared_ptr<detail::LpxContext> context_;
    ClusteringConfiguration configuration_;
    std::uint64_t last_epoch_{0U};
    bool failed_{false};
    std::vector<std::uint32_t> hull_offsets_;
    std::vector<std::uint32_t> hull_indices_;
    std::vector<float> hull_xy_;
    std::uint32_t last_size_{0U};
    std::uint32_t last_clusters_{0U};
    std::vector<std::uint32_t> group_offsets_;
    std::vector<std::uint32_t> group_indices_;
};

} // namespace lidar_processing

#endif // LIDAR_PROCESSING__CLUSTERING_HPP
