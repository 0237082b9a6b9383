// TEST-ONLY stand-in for <pcl/point_types.h> so that the drop-in headers can be compiled and run in
// an image without PCL.  Record sizes and the x, y, z prefix follow PCL (16-byte PointXYZ, 32-byte
// PointXYZI / PointXYZL / PointXYZRGB / PointXYZRGBL).  Not part of the product.
#pragma once
#include <cstdint>
namespace pcl
{
struct alignas(16) PointXYZ
{
    float x{0}, y{0}, z{0}, pad{1};
    PointXYZ() = default;
    PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) PointXYZI
{
    float x{0}, y{0}, z{0}, pad{1};
    float intensity{0};
    float pad2[3]{};
};
struct alignas(16) PointXYZL
{
    float x{0}, y{0}, z{0}, pad{1};
    std::uint32_t label{0};
    float pad2[3]{};
};
struct alignas(16) PointXYZRGB
{
    float x{0}, y{0}, z{0}, pad{1};
    std::uint32_t rgba{0};
    float pad2[3]{};
};
struct alignas(16) PointXYZRGBL
{
    float x{0}, y{0}, z{0}, pad{1};
    std::uint32_t rgba{0};
    std::uint32_t label{0};
    float pad2[2]{};
    PointXYZRGBL() = default;
    PointXYZRGBL(float x_, float y_, float z_, std::uint8_t r, std::uint8_t g, std::uint8_t b, std::uint32_t l)
        : x(x_), y(y_), z(z_), rgba((255U << 24) | (std::uint32_t(r) << 16) | (std::uint32_t(g) << 8) | b), label(l)
    {
    }
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointXYZI) == 32 && sizeof(PointXYZRGBL) == 32, "PCL record sizes");
} // namespace pcl
