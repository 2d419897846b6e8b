// Host timing of the Telea front march (openfx-opencv_amd/csrc/telea_march.h) on a BASELINE-like mask: 1920x1080, twelve
// ellipses, ~4-5 % of the frame.  g++ -O2 -std=c++17 -pthread -I openfx-opencv_amd/csrc tools/march_bench.cpp -o /tmp/march_bench
#include <chrono>
#include <cstdio>
#include <random>

#include "telea_march.h"

using namespace ofxcv_telea;

int main(int argc, char **argv) {
    const int w = argc > 1 ? atoi(argv[1]) : 1920, h = argc > 2 ? atoi(argv[2]) : 1080;
    std::mt19937 rng(42);
    std::vector<uint8_t> mask((size_t)w * h, 0);
    for (int b = 0; b < 12; b++) {
        const int cx = rng() % w, cy = rng() % h, rx = w / 60 + rng() % (w / 26), ry = h / 60 + rng() % (h / 15);
        for (int y = std::max(0, cy - ry); y <= std::min(h - 1, cy + ry); y++)
            for (int x = std::max(0, cx - rx); x <= std::min(w - 1, cx + rx); x++) {
                const double dx = (x - cx) / (double)rx, dy = (y - cy) / (double)ry;
                if (dx * dx + dy * dy <= 1.0) mask[(size_t)y * w + x] = 255;
            }
    }
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    March m;
    double best_b = 1e9, best_m = 1e9;
    unsigned long long sig = 0;
    for (int rep = 0; rep < 9; rep++) {
        m.prepare(w, h, 3);
        const double t0 = now();
        march_begin(mask.data(), true, m);
        const double t1 = now();
        m.pix.reserve(m.holes.size());
        while (march_advance(m, 8192) > 0) {}
        const double t2 = now();
        best_b = std::min(best_b, t1 - t0);
        best_m = std::min(best_m, t2 - t1);
        sig = 0;
        for (size_t k = 0; k < m.pix.size(); k++) sig = sig * 1000003ull + (unsigned)m.pix[k];
        for (int p : m.pix) {
            unsigned b;
            std::memcpy(&b, &m.t[p], 4);
            sig = sig * 1000003ull + b;
        }
    }
    std::printf("%dx%d: %zu hole pixels, %zu filled; set-up + ring %.2f ms, inward march %.2f ms (%.1f ns per pixel); order/T signature %016llx\n", w, h,
                m.holes.size(), m.pix.size(), best_b, best_m, best_m * 1e6 / m.pix.size(), sig);
    return 0;
}
