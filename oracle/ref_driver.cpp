// ref_driver.cpp -- TEST INFRASTRUCTURE.  Thin C-ABI driver around the REFERENCE's own, unmodified,
// std-only headers (src/kdtree.hpp, src/queue.hpp, src/vector.hpp, src/stack.hpp), compiled from
// where they lie under /root/reference by oracle/Makefile into oracle/_ref/libkdref.so.
// No reference source is copied into this repository; nothing here runs on the product path.
//
// What it pins: KDTree<float,3>::rebuild (kdtree.hpp:174-225, i.e. libstdc++'s std::nth_element
// tie placement), KDTree::radius_search emission order (kdtree.hpp:292-341), dist_sqr (:145-163)
// and containers::Queue FIFO behaviour (queue.hpp:121-169).
//
// src/clustering.cpp itself cannot be built here (it includes PCL headers, which this image
// lacks, and writing stand-ins for them is not allowed), so ref_fec() below drives the
// reference's KDTree and Queue objects with a restatement of the loop at clustering.cpp:47-125.
#include "kdtree.hpp"
#include "queue.hpp"
#include "vector.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

using lidar_processing::KDTree;
using lidar_processing::Point;


namespace
{
void fill_points(const float *xyz, std::uint32_t m, containers::Vector<Point<float, 3>> &points)
{
    points.clear();
    points.reserve(m);
    for (std::uint32_t i = 0; i < m; ++i)
    {
        points.push_back({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    }
}
} // namespace

extern "C"
{
    // Pre-order sequence of original indices: an unbounded radius search visits every node,
    // left before right (kdtree.hpp:324-328).
    int ref_kd_preorder(const float *xyz, std::uint32_t m, std::uint32_t *out)
    {
        if (m == 0)
            return 0;
        KDTree<float, 3> tree;
        containers::Vector<Point<float, 3>> points;
        fill_points(xyz, m, points);
        tree.rebuild(points);
        containers::Vector<KDTree<float, 3>::RetT> neigh;
        neigh.reserve(m);
        tree.radius_search(points[0], std::numeric_limits<float>::infinity(), neigh);
        if (neigh.size() != m)
            return -1;
        for (std::uint32_t i = 0; i < m; ++i)
            out[i] = neigh[i].first;
        return 0;
    }

    int ref_radius_search(const float *xyz, std::uint32_t m, const float *target, float r2, std::uint32_t *out_idx,
                          float *out_dist, std::uint32_t *count)
    {
        *count = 0;
        if (m == 0)
            return 0;
        KDTree<float, 3> tree;
        containers::Vector<Point<float, 3>> points;
        fill_points(xyz, m, points);
        tree.rebuild(points);
        containers::Vector<KDTree<float, 3>::RetT> neigh;
        neigh.reserve(m);
        tree.radius_search({target[0], target[1], target[2]}, r2, neigh);
        for (std::uint32_t i = 0; i < neigh.size(); ++i)
        {
            out_idx[i] = neigh[i].first;
            out_dist[i] = neigh[i].second;
        }
        *count = static_cast<std::uint32_t>(neigh.size());
        return 0;
    }

    // many queries against one tree: targets are points of the cloud (query_idx), results as CSR
    int ref_radius_search_many(const float *xyz, std::uint32_t m, const std::uint32_t *query_idx, std::uint32_t nq,
                               float r2, std::uint64_t *offsets, std::uint32_t *out_idx, float *out_dist,
                               std::uint64_t capacity)
    {
        offsets[0] = 0;
        if (m == 0)
            return 0;
        KDTree<float, 3> tree;
        containers::Vector<Point<float, 3>> points;
        fill_points(xyz, m, points);
        tree.rebuild(points);
        containers::Vector<KDTree<float, 3>::RetT> neigh;
        neigh.reserve(m);
        std::uint64_t total = 0;
        for (std::uint32_t q = 0; q < nq; ++q)
        {
            tree.radius_search(points[query_idx[q]], r2, neigh);
            if (total + neigh.size() > capacity)
                return -2;
            for (std::uint32_t i = 0; i < neigh.size(); ++i)
            {
                out_idx[total + i] = neigh[i].first;
                out_dist[total + i] = neigh[i].second;
            }
            total += neigh.size();
            offsets[q + 1] = total;
        }
        return 0;
    }

    // std::nth_element of this toolchain's libstdc++ on (key, payload) pairs, comparator on key only
    void ref_nth_element_u32(float *keys, std::uint32_t *payload, std::uint32_t first, std::uint32_t nth,
                             std::uint32_t last)
    {
        struct KP
        {
            float k;
            std::uint32_t p;
        };
        std::vector<KP> v(last - first);
        for (std::uint32_t i = first; i < last; ++i)
            v[i - first] = {keys[i], payload[i]};
        std::nth_element(v.begin(), v.begin() + (nth - first), v.end(),
                         [](const KP &a, const KP &b) { return a.k < b.k; });
        for (std::uint32_t i = first; i < last; ++i)
        {
            keys[i] = v[i - first].k;
            payload[i] = v[i - first].p;
        }
    }

    // The loop of Clusterer::cluster (clustering.cpp:47-125) over the reference KDTree + Queue.
    int ref_fec(const float *xyz, std::uint32_t m, float distance_squared, float cluster_quality,
                std::uint32_t min_cluster_size, std::uint32_t max_cluster_size, std::int32_t *labels,
                std::uint32_t *n_clusters)
    {
        constexpr std::int32_t UNDEFINED = std::numeric_limits<std::int32_t>::lowest();
        constexpr std::int32_t INVALID = -1;
        *n_clusters = 0;
        if (m == 0)
            return 0;
        std::fill(labels, labels + m, UNDEFINED);

        KDTree<float, 3> tree;
        containers::Vector<Point<float, 3>> points;
        fill_points(xyz, m, points);
        tree.rebuild(points);

        containers::Vector<KDTree<float, 3>::RetT> neigh;
        neigh.reserve(m);
        containers::Vector<std::uint32_t> indices;
        indices.reserve(m);
        std::vector<bool> removed(m, false);
        containers::Queue<std::uint32_t> queue;
        queue.reserve(m);

        const auto threshold = std::pow(1.0 - cluster_quality, 2) * distance_squared;

        std::int32_t label = 0;
        for (std::uint32_t i = 0; i < m; ++i)
        {
            if (removed[i])
                continue;
            queue.push(i);
            indices.clear();
            while (queue.size() > 0U)
            {
                const auto j = queue.front();
                queue.pop();
                if (removed[j])
                    continue;
                tree.radius_search(points[j], distance_squared, neigh);
                for (const auto &[k, dist] : neigh)
                {
                    if (removed[k])
                        continue;
                    labels[k] = label;
                    indices.push_back(k);
                    if (dist <= threshold)
                        removed[k] = true;
                    else
                        queue.push(k);
                }
            }
            if ((indices.size() < min_cluster_size) || (indices.size() > max_cluster_size))
            {
                for (const auto &index : indices)
                    labels[index] = INVALID;
            }
            else
            {
                ++label;
            }
        }
        *n_clusters = static_cast<std::uint32_t>(label);
        return 0;
    }
}
