// tests/stubs/opencv2/highgui/highgui.hpp -- declarations only
#include "../core/core.hpp"
