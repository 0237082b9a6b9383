// segmentation.hpp -- drop-in replacement of the reference's src/segmentation.hpp.
//
// Same names, namespace, defaults and call surface as the reference (src/segmentation.hpp:41-139), so
// src/processor.cpp compiles against it unchanged (#include "segmentation.hpp", :30; members :129;
// call :150).  The work is done on the MI355X through the C-ABI of include/lpx.h; this header only
// gathers the points of the two output clouds on the host, in the reference's order
// (src/segmentation.cpp:331-343).
#ifndef LIDAR_PROCESSING__SEGMENTATION_HPP
#define LIDAR_PROCESSING__SEGMENTATION_HPP

#include "lpx_context.hpp"

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

#include <cstdint>
#include <vector>

namespace lidar_processing
{
enum class SegmentationLabel : std::uint32_t
{
    UNKNOWN = 0U,
    GROUND,
    OBSTACLE
};

struct SegmentationConfiguration final
{
    float sensor_height_m{1.73F};
    float orthogonal_distance_threshold{0.3F};
    float initial_seed_threshold{0.6F};
    std::uint32_t number_of_iterations{3U};
    std::uint32_t number_of_planar_partitions{2U};
    std::uint32_t number_of_lower_point_representatives{5000U};
};

class Segmenter final
{
  public:
    // the process-wide context, shared with a default-constructed Clusterer (lpx_context.hpp: default_context)
    Segmenter() : context_{detail::default_context()}, configuration_{}
    {
        reserve_memory();
    }
    explicit Segmenter(std::shared_ptr<detail::LpxContext> context) : context_{std::move(context)}, configuration_{}
    {
        reserve_memory();
    }
    ~Segmenter() = default;

    void update_configuration(const SegmentationConfiguration &configuration)
    {
        configuration_ = configuration;
        reserve_memory();
    }

    void reserve_memory(std::uint32_t number_of_points = 200'000U)
    {
        std::lock_guard<std::mutex> lock(context_->mutex());
        lpx_reserve(context_->get(), number_of_points, 0U);
        ground_indices_.reserve(number_of_points);
        obstacle_indices_.reserve(number_of_points);
    }

    // Plane coefficients a, b, c, d (ax + by + cz = d) of the last plane fitted in every segment; the
    // reference keeps them private (src/segmentation.cpp:245).
    const std::vector<float> &planes() const noexcept
    {
        return planes_;
    }

    // shared GPU context, to keep the obstacle cloud on the device for the Clusterer
    const std::shared_ptr<detail::LpxContext> &context() const noexcept
    {
        return context_;
    }

    template <typename PointT>
    void segment(const pcl::PointCloud<PointT> &cloud_in, std::vector<SegmentationLabel> &labels,
                 pcl::PointCloud<PointT> &ground_cloud, pcl::PointCloud<PointT> &obstacle_cloud)
    {
        static_assert(sizeof(SegmentationLabel) == sizeof(std::uint32_t), "label layout");
        std::lock_guard<std::mutex> lock(context_->mutex());
        context_->segment_owner = this;
        context_->cluster_owner = nullptr;  // the look-ahead may cluster THIS cloud into the label buffer
        labels.resize(cloud_in.size(), SegmentationLabel::UNKNOWN);
        ground_cloud.clear();
        obstacle_cloud.clear();

        const std::uint32_t number_of_points = static_cast<std::uint32_t>(cloud_in.points.size());
        last_ground_ = 0U;
        last_obstacle_ = 0U;
        if (number_of_points == 0)
        {
            return;
        }

        ground_indices_.resize(number_of_points);
        obstacle_indices_.resize(number_of_points);
        planes_.assign(4U * configuration_.number_of_planar_partitions, 0.0F);

        lpx_seg_cfg cfg{};
        cfg.sensor_height_m = configuration_.sensor_height_m;
        cfg.orthogonal_distance_threshold = configuration_.orthogonal_distance_threshold;
        cfg.initial_seed_threshold = configuration_.initial_seed_threshold;
        cfg.number_of_iterations = configuration_.number_of_iterations;
        cfg.number_of_planar_partitions = configuration_.number_of_planar_partitions;
        cfg.number_of_lower_point_representatives = configuration_.number_of_lower_point_representatives;

        std::uint32_t number_of_ground = 0U;
        std::uint32_t number_of_obstacle = 0U;
        const int rc = lpx_segment(context_->get(), detail::points_base(cloud_in.points.data()), sizeof(PointT),
                                   number_of_points, &cfg, reinterpret_cast<std::uint32_t *>(labels.data()),
                                   ground_indices_.data(), &number_of_ground, obstacle_indices_.data(),
                                   &number_of_obstacle, planes_.data());
        if (rc != LPX_OK)
        {
            // the reference never throws from segment(): it reports on std::cerr and degrades to
            // "everything is an obstacle" (src/segmentation.cpp:251-259)
            std::cerr << "Failed ground segmentation: " << lpx_last_error(context_->get()) << std::endl;
            for (std::uint32_t i = 0U; i < number_of_points; ++i)
            {
                labels[i] = SegmentationLabel::OBSTACLE;
                obstacle_cloud.push_back(cloud_in[i]);
            }
            return;
        }

        last_ground_ = number_of_ground;
        last_obstacle_ = number_of_obstacle;
        // the two output clouds in the order of the index lists (src/segmentation.cpp:331-343): one resize and an indexed
        // copy each, on a few threads for large clouds (lpx_context.hpp: gather_cloud)
        detail::gather_cloud(cloud_in, ground_indices_.data(), number_of_ground, ground_cloud);
        detail::gather_cloud(cloud_in, obstacle_indices_.data(), number_of_obstacle, obstacle_cloud);
    }

    // Optional fast path for what the reference's caller does right after segment() (src/processor.cpp:152-163):
    // the ground cloud recoloured as PointXYZRGBL(x, y, z, 220, 220, 220, label 0) and the obstacle cloud as
    // (x, y, z, 0, 255, 0, label 1), whose records the node then memcpy's into the published PointCloud2 messages
    // (src/conversions.cpp:164-193).  The records are written on the device from the points that are still
    // resident there; valid after segment() and before the next call on the (possibly shared) context: with a
    // Clusterer on the same LpxContext, call it before cluster() -- as processor.cpp does -- or it throws.
    template <typename PointOutT>
    void coloured_clouds(pcl::PointCloud<PointOutT> &ground_cloud, pcl::PointCloud<PointOutT> &obstacle_cloud)
    {
        static_assert(sizeof(PointOutT) == 32U, "pcl::PointXYZRGBL records are 32 bytes");
        std::lock_guard<std::mutex> lock(context_->mutex());
        if (context_->segment_owner != this)
        {
            throw std::runtime_error("coloured clouds failed: another Segmenter has used the shared context since segment()");
        }
        ground_cloud.points.resize(last_ground_);
        obstacle_cloud.points.resize(last_obstacle_);
        std::uint32_t number_of_ground = 0U;
        std::uint32_t number_of_obstacle = 0U;
        PointOutT scratch{};  // a null destination is an argument error; an empty cloud has no data()
        const int rc = lpx_coloured_clouds(context_->get(), last_ground_ ? static_cast<void *>(ground_cloud.points.data()) : &scratch,
                                           last_obstacle_ ? static_cast<void *>(obstacle_cloud.points.data()) : &scratch,
                                           &number_of_ground, &number_of_obstacle);
        if (rc != LPX_OK || number_of_ground != last_ground_ || number_of_obstacle != last_obstacle_)
        {
            throw std::runtime_error(std::string("coloured clouds failed: ") + lpx_last_error(context_->get()));
        }
    }

  private:
    std::shared_ptr<detail::LpxContext> context_;
    SegmentationConfiguration configuration_;
    std::uint32_t last_ground_{0U};
    std::uint32_t last_obstacle_{0U};
    std::vector<std::uint32_t> ground_indices_;
    std::vector<std::uint32_t> obstacle_indices_;
    std::vector<float> planes_;
};

} // namespace lidar_processing

#endif // LIDAR_PROCESSING__SEGMENTATION_HPP
