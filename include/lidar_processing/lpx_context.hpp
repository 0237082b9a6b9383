// lpx_context.hpp -- RAII holder of one lpx_ctx shared by the drop-in Segmenter / Clusterer classes.
#ifndef LIDAR_PROCESSING__LPX_CONTEXT_HPP
#define LIDAR_PROCESSING__LPX_CONTEXT_HPP

#include "lpx.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <type_traits>
#include <vector>

namespace lidar_processing
{
namespace detail
{
class LpxContext final
{
  public:
    explicit LpxContext(int device = 0)
    {
        const int rc = lpx_create(device, &ctx_);
        if (rc != LPX_OK)
        {
            // no CPU fallback: the object cannot work without the GPU
            throw std::runtime_error("lpx_create failed (" + std::to_string(rc) + "): no usable MI355X/HIP device");
        }
    }
    ~LpxContext()
    {
        lpx_destroy(ctx_);
    }
    LpxContext(const LpxContext &) = delete;
    LpxContext &operator=(const LpxContext &) = delete;

    lpx_ctx *get() const noexcept
    {
        return ctx_;
    }

    // One call at a time per context (lpx.h: a context is not thread-safe).  The drop-in classes share ONE context by
    // default (default_context below), so two objects used from two threads serialise here instead of racing.
    std::mutex &mutex() noexcept
    {
        return mutex_;
    }

    // which object's segmentation the resident clouds belong to (Segmenter::coloured_clouds checks it)
    const void *segment_owner{nullptr};
    // which Clusterer's labels are resident (Clusterer::regroup / convex_outlines check it; Segmenter::segment and every
    // Clusterer::cluster reset it)
    const void *cluster_owner{nullptr};

  private:
    lpx_ctx *ctx_{nullptr};
    std::mutex mutex_;
};

// The context default-constructed Segmenter and Clusterer objects share.  The reference's node constructs one of each
// as members (src/processor.cpp:129-132) and calls segment() then cluster() on the cloud segment() produced (:150,
// :178): on one context the obstacle cloud is still on the device when cluster() arrives, and lpx_cluster recognises it
// (size + checksum) and skips the upload -- and the two objects hold ONE workspace instead of two.  Alive as long as
// any object that uses it.
inline std::shared_ptr<LpxContext> default_context()
{
    static std::mutex guard;
    static std::weak_ptr<LpxContext> shared;
    std::lock_guard<std::mutex> lock(guard);
    std::shared_ptr<LpxContext> context = shared.lock();
    if (!context)
    {
        context = std::make_shared<LpxContext>();
        shared = context;
    }
    return context;
}

// byte offset of x inside a PCL point record and the record size; x, y, z are the first three floats
// of every pcl::PointXYZ* type (PCL_ADD_POINT4D)
template <typename PointT> inline const void *points_base(const PointT *p) noexcept
{
    return static_cast<const void *>(&p->x);
}

// cloud_out = the points of cloud_in named by indices[0 .. count), in that order: what the reference builds with one
// push_back per point (src/segmentation.cpp:331-343).  push_back re-checks the capacity and bumps the size for every
// point, and a 5M-point frame spent 45 of its 59 ms there -- the indices arrive x-sorted, so every source record is a
// cache miss that one core takes one at a time.  Here: one resize, then an indexed copy, split over a few threads from
// kParallelGatherFrom points on (each thread owns a contiguous slice of the OUTPUT; nothing is shared but the read-only
// input).  Same records in the same order; a thread that cannot be started just leaves its slice to the caller.
constexpr std::uint32_t kParallelGatherFrom = 250'000U;
constexpr unsigned kGatherThreads = 4U;

template <typename CloudT>
inline void gather_cloud(const CloudT &cloud_in, const std::uint32_t *indices, std::uint32_t count, CloudT &cloud_out)
{
    using PointT = typename std::remove_cv<typename std::remove_reference<decltype(cloud_in.points[0])>::type>::type;
    static_assert(std::is_trivially_copyable<PointT>::value, "PCL point records are plain data");
    cloud_out.resize(count);
    if (count == 0U)
    {
        return;
    }
    const PointT *const source = cloud_in.points.data();
    PointT *const target = cloud_out.points.data();
    const auto copy_slice = [source, target, indices](std::uint32_t begin, std::uint32_t end) {
        for (std::uint32_t i = begin; i < end; ++i)
        {
            target[i] = source[indices[i]];
        }
    };
    unsigned workers = 1U;
    if (count >= kParallelGatherFrom)
    {
        const unsigned cores = std::thread::hardware_concurrency();
        workers = std::max(1U, std::min(kGatherThreads, cores == 0U ? 1U : cores));
    }
    const std::uint32_t slice = (count + workers - 1U) / workers;
    std::vector<std::thread> helpers;
    std::uint32_t done_to = slice < count ? slice : count;  // the caller's own slice is the first
    for (unsigned w = 1U; w < workers; ++w)
    {
        const std::uint32_t begin = std::min(count, w * slice), end = std::min(count, begin + slice);
        try
        {
            helpers.emplace_back(copy_slice, begin, end);
        }
        catch (const std::system_error &)
        {
            copy_slice(begin, end);  // no thread to be had: do it here
        }
    }
    copy_slice(0U, done_to);
    for (std::thread &helper : helpers)
    {
        helper.join();
    }
}
} // namespace detail
} // namespace lidar_processing

#endif // LIDAR_PROCESSING__LPX_CONTEXT_HPP
