// lpx_context.hpp -- RAII holder of one lpx_ctx shared by the drop-in Segmenter / Clusterer classes.
#ifndef LIDAR_PROCESSING__LPX_CONTEXT_HPP
#define LIDAR_PROCESSING__LPX_CONTEXT_HPP

#include "lpx.h"

#include <cstdint>
#include <iostream>
#include <memory>
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

  private:
    lpx_ctx *ctx_{nullptr};
};

// byte offset of x inside a PCL point record and the record size; x, y, z are the first three floats
// of every pcl::PointXYZ* type (PCL_ADD_POINT4D)
template <typename PointT> inline const void *points_base(const PointT *p) noexcept
{
    return static_cast<const void *>(&p->x);
}
} // namespace detail
} // namespace lidar_processing

#endif // LIDAR_PROCESSING__LPX_CONTEXT_HPP
