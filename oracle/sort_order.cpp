// sort_order.cpp -- TEST INFRASTRUCTURE (tests/golden/make_sort_bound.py).  The index order the reference's x sort
// (src/segmentation.cpp:114-122: std::sort(std::execution::par, indices, [x[i] < x[j]])) leaves for points with EQUAL x,
// which the C++ standard does not specify, as the two builds of it a user can have produce it with this image's
// libstdc++ (GCC 11.4, a third-party dependency of the reference):
//   mode 1  no TBB headers at build time: libstdc++'s PSTL falls back to the serial backend and the call is plain
//           std::sort -- introsort + final insertion sort (bits/stl_algo.h:1940-1975), run here as it is;
//   mode 2  TBB backend: pstl/parallel_backend_tbb.h:1116-1180 -- the range is halved recursively while it holds more
//           than _PSTL_STABLE_SORT_CUT_OFF = 500 elements, every leaf is sorted with std::sort (algorithm_impl.h:2116-2121)
//           and the halves are merged STABLY (__serial_move_merge takes the left element on ties); restated with
//           std::sort on the same leaves and std::inplace_merge.
// The canonical order of this repository is (x, index), i.e. a fully stable sort.  Nothing here is on the product path.
#include <algorithm>
#include <cstdint>
#include <numeric>

namespace
{
template <class Cmp> void tbb_like(std::uint32_t *b, std::uint32_t *e, Cmp cmp)
{
    const std::ptrdiff_t n = e - b;
    if (n <= 500)
    {
        std::sort(b, e, cmp);
        return;
    }
    std::uint32_t *m = b + n / 2;
    tbb_like(b, m, cmp);
    tbb_like(m, e, cmp);
    std::inplace_merge(b, m, e, cmp);
}
}  // namespace

extern "C" int so_order(const float *x, std::uint32_t n, std::uint32_t *idx, int mode)
{
    std::iota(idx, idx + n, 0u);
    auto cmp = [x](std::uint32_t a, std::uint32_t b) -> bool { return x[a] < x[b]; };
    if (mode == 1)
        std::sort(idx, idx + n, cmp);
    else if (mode == 2)
        tbb_like(idx, idx + n, cmp);
    else
        std::stable_sort(idx, idx + n, cmp);
    return 0;
}
