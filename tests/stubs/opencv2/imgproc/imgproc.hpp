// tests/stubs/opencv2/imgproc/imgproc.hpp -- declarations only
#include "../core/core.hpp"
