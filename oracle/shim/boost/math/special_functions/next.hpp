#pragma once
#include <cmath>
namespace boost { namespace math {
template <typename T> inline T nextafter(T a, T b) { return std::nextafter(a, b); }
} }
