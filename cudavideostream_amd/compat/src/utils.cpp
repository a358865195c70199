// compat/src/utils.cpp -- diff::utils::matsz::area (reference server/src/utils.cpp:5-7).
#include "../include/utils.hpp"

int diff::utils::matsz::area() { return width * height; }
