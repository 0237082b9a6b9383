// Test harness for the failure and ownership behaviour of the drop-in Clusterer (include/lidar_processing/clustering.hpp):
//   * a device error inside cluster() is retried once and otherwise degrades to "the whole cloud is ONE cluster" (the
//     safe direction: the caller publishes everything as an obstacle, never an empty set) with a line on std::cerr and
//     failed() == true -- no exception (the reference's cluster() cannot fail on a non-empty cloud, src/clustering.cpp:47-125);
//     run against liblpx_dev.so with LPX_FAIL_CLUSTER=1 (retry succeeds) / =2 (degrades);
//   * regroup() / convex_outlines() refuse to serve labels that another object's call has replaced on the shared
//     default context (a Segmenter::segment with its look-ahead, another Clusterer): std::runtime_error.
//
// in : raw float32 records x y z intensity;  prints one line of key=value pairs the Python test parses
#include "clustering.hpp"
#include "segmentation.hpp"

#include <cstdint>
#include <cstdio>
#include <vector>

namespace lp = lidar_processing;

struct XY  // like geom::Point<float>: public x, y and an (x, y) constructor
{
    float x, y;
    XY(float x_, float y_) : x(x_), y(y_) {}
};

template <typename F> static int throws(F &&f)
{
    try
    {
        f();
    }
    catch (const std::runtime_error &)
    {
        return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in)
        return 2;
    pcl::PointCloud<pcl::PointXYZI> cloud;
    for (float r[4]; std::fread(r, sizeof r, 1, in) == 1;)
    {
        pcl::PointXYZI p;
        p.x = r[0], p.y = r[1], p.z = r[2], p.intensity = r[3];
        cloud.push_back(p);
    }
    std::fclose(in);

    lp::Segmenter segmenter;
    lp::Clusterer clusterer, other;  // all three on the process-wide default context
    std::vector<lp::SegmentationLabel> seg_labels;
    pcl::PointCloud<pcl::PointXYZI> ground, obstacles;
    std::vector<lp::ClusteringLabel> labels, labels2;
    std::vector<pcl::PointCloud<pcl::PointXYZ>> groups;
    std::vector<std::vector<XY>> outlines;

    // 1. the node's sequence; with LPX_FAIL_CLUSTER the first cluster() call meets forced device errors
    segmenter.segment(cloud, seg_labels, ground, obstacles);
    int threw = throws([&] { clusterer.cluster(obstacles, labels); });
    std::size_t invalid = 0, undefined = 0;
    std::int64_t top = -1;
    for (const lp::ClusteringLabel l : labels)
    {
        invalid += l == lp::Clusterer::INVALID;
        undefined += l == lp::Clusterer::UNDEFINED;
        top = l > top ? l : top;
    }
    clusterer.regroup(obstacles, groups);  // a degraded call has one group, the whole cloud; must not throw either way
    std::printf("points=%zu obstacle=%zu labels=%zu invalid=%zu undefined=%zu clusters=%lld groups=%zu group0=%zu failed=%d "
                "threw=%d", cloud.size(), obstacles.size(), labels.size(), invalid, undefined, static_cast<long long>(top + 1),
                groups.size(), groups.empty() ? std::size_t{0} : groups[0].size(), clusterer.failed() ? 1 : 0, threw);

    // 2. frame after frame (the look-ahead arms itself on the second pair): regroup right after cluster() works ...
    int ok_pairs = 0;
    for (int k = 0; k < 3; ++k)
    {
        segmenter.segment(cloud, seg_labels, ground, obstacles);
        clusterer.cluster(obstacles, labels);
        ok_pairs += !throws([&] { clusterer.regroup(obstacles, groups); }) && !groups.empty();
    }
    // ... a segment() in between (its look-ahead clusters into the label buffer) makes regroup / outlines refuse
    segmenter.segment(cloud, seg_labels, ground, obstacles);
    const int after_segment = throws([&] { clusterer.regroup(obstacles, groups); }) +
                              throws([&] { clusterer.convex_outlines(outlines); });
    // ... and so does another Clusterer's call on the shared context, while that object is served
    clusterer.cluster(obstacles, labels);
    pcl::PointCloud<pcl::PointXYZI> half;
    for (std::size_t i = 0; i < obstacles.size() / 2; ++i)
        half.push_back(obstacles[i]);
    other.cluster(half, labels2);
    const int after_other = throws([&] { clusterer.regroup(obstacles, groups); }) +
                            throws([&] { clusterer.convex_outlines(outlines); });
    const int other_ok = !throws([&] { other.regroup(half, groups); }) && !groups.empty();
    std::printf(" ok_pairs=%d after_segment=%d after_other=%d other_ok=%d\n", ok_pairs, after_segment, after_other, other_ok);
    return 0;
}
