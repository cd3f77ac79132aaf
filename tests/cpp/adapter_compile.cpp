// compile-only check of the C++ adapter (flat-array form) and a host-only run of the getters
#include <cstdio>
#include "ORBextractor_hip.hpp"
int main()
{
    try {
        iORB_SLAM::ORBextractor ex(1000, 1.2f, 8, 20, 7, 640, 480, /*device=*/-1);  // host-only handle
        if (ex.GetLevels() != 8) return 2;
        if (ex.GetScaleFactors()[1] != 1.2f) return 3;
        std::vector<OrbxKeyPoint> k; std::vector<uint8_t> d;
        std::vector<uint8_t> img(640 * 480, 7);
        ex(img.data(), 640, 480, 640, k, d);  // refuses to compute without a device, returns 0 keypoints
        if (!k.empty()) return 4;
        std::puts("adapter ok");
        return 0;
    } catch (const std::exception& e) { std::fprintf(stderr, "%s\n", e.what()); return 1; }
}
