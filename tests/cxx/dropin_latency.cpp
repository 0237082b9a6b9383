// What the UNCHANGED processor node pays per message for the two hot-path calls, through the drop-in headers
// themselves (not the Python wrappers): Segmenter::segment on a PointXYZI cloud (reference src/processor.cpp:150),
// the node's own recolour copy of the obstacle cloud (:156-163, host code of the node, timed separately), then
// Clusterer::cluster on that cloud (:178).  Default-constructed objects, i.e. the shared context of
// lpx_context.hpp -- and, for comparison, two objects with a context each (the round-3 default).
//
// in : raw float32 records x y z intensity;  out (stdout): one JSON object, milliseconds (median of `reps`)
#include "clustering.hpp"
#include "segmentation.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace lp = lidar_processing;
using clk = std::chrono::steady_clock;

static double ms(clk::time_point a, clk::time_point b)
{
    return std::chrono::duration<double, std::milli>(b - a).count();
}

static double median(std::vector<double> v)
{
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

struct Timing
{
    double segment, recolour, cluster, callback;  // callback: segment() entered -> cluster() returned
    std::size_t obstacles, clusters;
};

static Timing run(lp::Segmenter &segmenter, lp::Clusterer &clusterer, const pcl::PointCloud<pcl::PointXYZI> &cloud,
                  int reps, const lp::SegmentationConfiguration &scfg, const lp::ClusteringConfiguration &ccfg)
{
    segmenter.update_configuration(scfg);
    clusterer.update_configuration(ccfg);
    std::vector<lp::SegmentationLabel> seg_labels;
    std::vector<lp::ClusteringLabel> clu_labels;
    pcl::PointCloud<pcl::PointXYZI> ground, obstacles;
    pcl::PointCloud<pcl::PointXYZRGBL> coloured;
    std::vector<double> t_seg, t_col, t_clu, t_all;
    Timing out{};
    for (int r = 0; r < reps + 3; ++r)
    {
        const auto t0 = clk::now();
        segmenter.segment(cloud, seg_labels, ground, obstacles);
        const auto t1 = clk::now();
        coloured.clear();
        coloured.reserve(obstacles.size());
        for (std::size_t i = 0; i < obstacles.size(); ++i)
            coloured.emplace_back(obstacles[i].x, obstacles[i].y, obstacles[i].z, 0, 255, 0, 1);
        const auto t2 = clk::now();
        clusterer.cluster(coloured, clu_labels);
        const auto t3 = clk::now();
        if (r >= 3)  // the first iterations allocate
        {
            t_seg.push_back(ms(t0, t1));
            t_col.push_back(ms(t1, t2));
            t_clu.push_back(ms(t2, t3));
            t_all.push_back(ms(t0, t3));
        }
        out.obstacles = obstacles.size();
        out.clusters = clu_labels.empty() ? 0 : (std::size_t)(*std::max_element(clu_labels.begin(), clu_labels.end()) + 1);
    }
    out.segment = median(t_seg);
    out.recolour = median(t_col);
    out.cluster = median(t_clu);
    out.callback = median(t_all);
    return out;
}

int main(int argc, char **argv)
{
    if (argc < 2)
    {
        std::fprintf(stderr, "usage: dropin_latency in.f32 [reps] [partitions iterations distance_squared]\n");
        return 1;
    }
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in)
        return 1;
    pcl::PointCloud<pcl::PointXYZI> cloud;
    for (float r[4]; std::fread(r, sizeof r, 1, in) == 1;)
    {
        pcl::PointXYZI p;
        p.x = r[0], p.y = r[1], p.z = r[2], p.intensity = r[3];
        cloud.push_back(p);
    }
    std::fclose(in);
    const int reps = argc > 2 ? std::atoi(argv[2]) : 15;
    lp::SegmentationConfiguration scfg;
    lp::ClusteringConfiguration ccfg;
    if (argc > 5)
    {
        scfg.number_of_planar_partitions = (std::uint32_t)std::atoi(argv[3]);
        scfg.number_of_iterations = (std::uint32_t)std::atoi(argv[4]);
        ccfg.distance_squared = (float)std::atof(argv[5]);
    }
    Timing shared{}, plain{}, separate{};
    {
        lp::Segmenter segmenter;  // as the node constructs them: one shared context
        lp::Clusterer clusterer;
        shared = run(segmenter, clusterer, cloud, reps, scfg, ccfg);
    }
    {
        auto context = std::make_shared<lp::detail::LpxContext>();
        lpx_set_lookahead(context->get(), 0);
        lp::Segmenter segmenter{context};
        lp::Clusterer clusterer{context};
        plain = run(segmenter, clusterer, cloud, reps, scfg, ccfg);
    }
    {
        lp::Segmenter segmenter{std::make_shared<lp::detail::LpxContext>()};
        lp::Clusterer clusterer{std::make_shared<lp::detail::LpxContext>()};
        separate = run(segmenter, clusterer, cloud, reps, scfg, ccfg);
    }
    std::printf("{\"points\": %zu, \"obstacle_points\": %zu, \"clusters\": %zu, \"reps\": %d, "
                "\"segment_ms\": %.4f, \"cluster_ms\": %.4f, \"segment_plus_cluster_ms\": %.4f, \"callback_ms\": %.4f, "
                "\"node_recolour_copy_ms\": %.4f, "
                "\"lookahead_off\": {\"segment_ms\": %.4f, \"cluster_ms\": %.4f, \"segment_plus_cluster_ms\": %.4f, "
                "\"callback_ms\": %.4f}, "
                "\"separate_contexts\": {\"segment_ms\": %.4f, \"cluster_ms\": %.4f, "
                "\"segment_plus_cluster_ms\": %.4f, \"callback_ms\": %.4f}}\n",
                cloud.size(), shared.obstacles, shared.clusters, reps, shared.segment, shared.cluster,
                shared.segment + shared.cluster, shared.callback, shared.recolour, plain.segment, plain.cluster,
                plain.segment + plain.cluster, plain.callback, separate.segment, separate.cluster,
                separate.segment + separate.cluster, separate.callback);
    return shared.clusters == separate.clusters && shared.clusters == plain.clusters ? 0 : 2;
}
