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
    const bool dust = argc > 3 && !strcmp(argv[3], "dust"), scratches = argc > 3 && !strcmp(argv[3], "scratches");
    if (dust)  // 3000 specks of 3x3 .. 7x7 pixels
        for (int b = 0; b < 3000; b++) {
            const int x = 4 + rng() % (w - 8), y = 4 + rng() % (h - 8), r = 1 + rng() % 3;
            for (int yy = y - r; yy <= y + r; yy++)
                for (int xx = x - r; xx <= x + r; xx++) mask[(size_t)yy * w + xx] = 255;
        }
    if (scratches)  // 40 lines, 3 pixels wide
        for (int b = 0; b < 40; b++) {
            const double x0 = rng() % w, y0 = rng() % h, ang = (rng() % 3141) / 1000.0;
            const int L = 200 + rng() % 700;
            for (int t = 0; t < L; t++)
                for (int d = 0; d < 3; d++) {
                    const int x = (int)(x0 + t * std::cos(ang)), y = (int)(y0 + t * std::sin(ang)) + d;
                    if (x >= 0 && y >= 0 && x < w && y < h) mask[(size_t)y * w + x] = 255;
                }
        }
    for (int b = 0; b < 12 && !dust && !scratches; b++) {
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
