// lpx_context.hpp -- RAII holder of one lpx_ctx shared by the drop-in Segmenter / Clusterer classes.
#ifndef LIDAR_PROCESSING__LPX_CONTEXT_HPP
#define LIDAR_PROCESSING__LPX_CONTEXT_HPP

#include "lpx.h"

#include <cstdint>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>

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
} // namespace detail
} // namespace lidar_processing

#endif // LIDAR_PROCESSING__LPX_CONTEXT_HPP
