// Drives the drop-in headers exactly like Processor::process does (reference src/processor.cpp:135-200):
// segment -> recolour into PointXYZRGBL clouds -> cluster -> regroup.  Reads a raw float32 x y z i
// file, writes labels so that the Python test can compare them with the oracle.
#include "clustering.hpp"
#include "segmentation.hpp"

#include <algorithm>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <vector>

using namespace lidar_processing;

int main(int argc, char **argv)
{
    if (argc < 3)
    {
        std::fprintf(stderr, "usage: dropin_main in.f32 out.bin\n");
        return 2;
    }
    std::FILE *f = std::fopen(argv[1], "rb");
    if (!f)
        return 2;
    pcl::PointCloud<pcl::PointXYZI> cloud_in_;
    float rec[4];
    while (std::fread(rec, sizeof rec, 1, f) == 1)
    {
        pcl::PointXYZI p;
        p.x = rec[0];
        p.y = rec[1];
        p.z = rec[2];
        p.intensity = rec[3];
        cloud_in_.push_back(p);
    }
    std::fclose(f);

    Segmenter segmenter_;
    Clusterer clusterer_;
    pcl::PointCloud<pcl::PointXYZI> ground_points_;
    pcl::PointCloud<pcl::PointXYZI> obstacle_points_;
    std::vector<SegmentationLabel> segmentation_labels_;

    segmenter_.segment(cloud_in_, segmentation_labels_, ground_points_, obstacle_points_);

    auto obstacle_cloud = std::make_unique<pcl::PointCloud<pcl::PointXYZRGBL>>();
    obstacle_cloud->reserve(obstacle_points_.size());
    for (const auto &obstacle_point : obstacle_points_)
    {
        obstacle_cloud->emplace_back(obstacle_point.x, obstacle_point.y, obstacle_point.z, 0, 255, 0, 1);
    }

    std::vector<pcl::PointCloud<pcl::PointXYZ>> clustered_obstacle_cloud;
    std::vector<ClusteringLabel> cluster_labels;
    clusterer_.cluster(*obstacle_cloud, cluster_labels);

    const auto max_label = *std::max_element(cluster_labels.cbegin(), cluster_labels.cend());
    clustered_obstacle_cloud.resize(max_label + 1);
    for (std::size_t i = 0; i < obstacle_cloud->size(); ++i)
    {
        auto label = cluster_labels[i];
        if (label == Clusterer::UNDEFINED)
        {
            throw std::runtime_error("Undefined label found (clustering)");
        }
        if (label != Clusterer::INVALID)
        {
            const auto &point = obstacle_cloud->points[i];
            clustered_obstacle_cloud[label].emplace_back(point.x, point.y, point.z);
        }
    }

    // optional device-side regrouping must equal the caller's own loop above
    std::vector<pcl::PointCloud<pcl::PointXYZ>> regrouped;
    clusterer_.regroup(*obstacle_cloud, regrouped);
    if (regrouped.size() != clustered_obstacle_cloud.size())
    {
        throw std::runtime_error("regroup: cluster count differs");
    }
    for (std::size_t c = 0; c < regrouped.size(); ++c)
    {
        if (regrouped[c].size() != clustered_obstacle_cloud[c].size())
        {
            throw std::runtime_error("regroup: cluster size differs");
        }
        for (std::size_t p = 0; p < regrouped[c].size(); ++p)
        {
            if (regrouped[c][p].x != clustered_obstacle_cloud[c][p].x || regrouped[c][p].y != clustered_obstacle_cloud[c][p].y ||
                regrouped[c][p].z != clustered_obstacle_cloud[c][p].z)
            {
                throw std::runtime_error("regroup: point differs");
            }
        }
    }

    std::FILE *o = std::fopen(argv[2], "wb");
    const std::uint32_t n = static_cast<std::uint32_t>(cloud_in_.size());
    const std::uint32_t ng = static_cast<std::uint32_t>(ground_points_.size());
    const std::uint32_t no = static_cast<std::uint32_t>(obstacle_points_.size());
    const std::uint32_t nc = static_cast<std::uint32_t>(clustered_obstacle_cloud.size());
    std::fwrite(&n, 4, 1, o);
    std::fwrite(&ng, 4, 1, o);
    std::fwrite(&no, 4, 1, o);
    std::fwrite(&nc, 4, 1, o);
    std::fwrite(segmentation_labels_.data(), 4, n, o);
    std::fwrite(cluster_labels.data(), 4, no, o);
    for (const auto &p : obstacle_points_)
    {
        std::fwrite(&p.x, 4, 3, o);
    }
    std::fclose(o);
    std::printf("points %u ground %u obstacle %u clusters %u\n", n, ng, no, nc);
    return 0;
}
