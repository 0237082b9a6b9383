// Test harness for the drop-in headers: makes the two hot-path calls the reference's processor node makes
// (Segmenter::segment on a PointXYZI cloud, then Clusterer::cluster on the obstacle cloud recoloured as
// PointXYZRGBL; reference src/processor.cpp:150 and :178) and dumps what the Python test compares with the
// oracle.  It also checks Clusterer::regroup against a per-label bucket count computed here.
//
// in : raw float32 records x y z intensity
// out: u32 n, n_ground, n_obstacle, n_clusters | u32 seg labels[n] | i32 cluster labels[n_obstacle] |
//      f32 obstacle xyz[n_obstacle * 3]
#include "clustering.hpp"
#include "segmentation.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace lp = lidar_processing;

static int fail(const char *what)
{
    std::fprintf(stderr, "dropin_main: %s\n", what);
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3)
        return fail("usage: dropin_main in.f32 out.bin");
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in)
        return fail("cannot open the input");
    pcl::PointCloud<pcl::PointXYZI> cloud;
    for (float r[4]; std::fread(r, sizeof r, 1, in) == 1;)
    {
        pcl::PointXYZI p;
        p.x = r[0], p.y = r[1], p.z = r[2], p.intensity = r[3];
        cloud.push_back(p);
    }
    std::fclose(in);

    lp::Segmenter segmenter;
    lp::Clusterer clusterer;
    std::vector<lp::SegmentationLabel> seg_labels;
    pcl::PointCloud<pcl::PointXYZI> ground, obstacles;
    segmenter.segment(cloud, seg_labels, ground, obstacles);  // call 1

    pcl::PointCloud<pcl::PointXYZRGBL> coloured;  // the point type the node hands to the clusterer
    coloured.reserve(obstacles.size());
    for (std::size_t i = 0; i < obstacles.size(); ++i)
        coloured.emplace_back(obstacles[i].x, obstacles[i].y, obstacles[i].z, 0, 255, 0, 1);

    // optional device-side recolouring: same records as the copy above (and the ground cloud's counterpart)
    pcl::PointCloud<pcl::PointXYZRGBL> ground_rgbl, obstacle_rgbl;
    segmenter.coloured_clouds(ground_rgbl, obstacle_rgbl);
    if (ground_rgbl.size() != ground.size() || obstacle_rgbl.size() != coloured.size())
        return fail("coloured_clouds: sizes");
    for (std::size_t i = 0; i < coloured.size(); ++i)
        if (std::memcmp(&obstacle_rgbl[i], &coloured[i], 24) != 0)  // x y z 1 | b g r a | label
            return fail("coloured_clouds: an obstacle record differs");
    for (std::size_t i = 0; i < ground.size(); ++i)
    {
        const pcl::PointXYZRGBL want(ground[i].x, ground[i].y, ground[i].z, 220, 220, 220, 0);
        if (std::memcmp(&ground_rgbl[i], &want, 24) != 0)
            return fail("coloured_clouds: a ground record differs");
    }

    std::vector<lp::ClusteringLabel> clu_labels;
    clusterer.cluster(coloured, clu_labels);  // call 2
    if (clu_labels.size() != coloured.size())
        return fail("one cluster label per obstacle point expected");

    // bucket sizes per label, and the running position of every point inside its bucket
    std::int64_t top = -1;
    for (const lp::ClusteringLabel l : clu_labels)
    {
        if (l == lp::Clusterer::UNDEFINED)
            return fail("a point was left UNDEFINED");
        if (l > top)
            top = l;
    }
    std::vector<std::uint32_t> bucket(static_cast<std::size_t>(top + 1), 0U);
    std::vector<std::uint32_t> slot(clu_labels.size(), 0U);
    for (std::size_t i = 0; i < clu_labels.size(); ++i)
        if (clu_labels[i] != lp::Clusterer::INVALID)
            slot[i] = bucket[static_cast<std::size_t>(clu_labels[i])]++;

    // the device-side regrouping must put point i of label l at position slot[i] of cloud l
    std::vector<pcl::PointCloud<pcl::PointXYZ>> groups;
    clusterer.regroup(coloured, groups);
    if (groups.size() != bucket.size())
        return fail("regroup: number of clusters");
    for (std::size_t c = 0; c < groups.size(); ++c)
        if (groups[c].size() != bucket[c] || bucket[c] == 0U)
            return fail("regroup: cluster size (or an empty cluster)");
    for (std::size_t i = 0; i < clu_labels.size(); ++i)
    {
        if (clu_labels[i] == lp::Clusterer::INVALID)
            continue;
        const pcl::PointXYZ &g = groups[static_cast<std::size_t>(clu_labels[i])][slot[i]];
        if (g.x != coloured[i].x || g.y != coloured[i].y || g.z != coloured[i].z)
            return fail("regroup: a point is out of place");
    }

    std::FILE *out = std::fopen(argv[2], "wb");
    if (!out)
        return fail("cannot open the output");
    const std::uint32_t head[4] = {static_cast<std::uint32_t>(cloud.size()), static_cast<std::uint32_t>(ground.size()),
                                   static_cast<std::uint32_t>(obstacles.size()),
                                   static_cast<std::uint32_t>(bucket.size())};
    std::fwrite(head, sizeof head, 1, out);
    std::fwrite(seg_labels.data(), sizeof(std::uint32_t), seg_labels.size(), out);
    std::fwrite(clu_labels.data(), sizeof(std::int32_t), clu_labels.size(), out);
    for (std::size_t i = 0; i < obstacles.size(); ++i)
    {
        const float xyz[3] = {obstacles[i].x, obstacles[i].y, obstacles[i].z};
        std::fwrite(xyz, sizeof xyz, 1, out);
    }
    std::fclose(out);
    std::printf("points %u ground %u obstacle %u clusters %u\n", head[0], head[1], head[2], head[3]);
    return 0;
}
