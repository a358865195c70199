// compat/include/utils.hpp -- diff::utils::matsz as the reference server uses it
// (server/include/utils.hpp:7-16): two ints, height first, plus area().
#ifndef MI355_COMPAT_UTILS_HPP_
#define MI355_COMPAT_UTILS_HPP_

namespace diff {
namespace utils {

struct matsz {
    int height;
    int width;
    matsz(int h, int w) : height(h), width(w) {}
    matsz() : height(0), width(0) {}
    int area();
};

}  // namespace utils
}  // namespace diff
#endif
