// CPU test of the component-parallel Telea front march (openfx-opencv_amd/csrc/telea_march.h): on random hole masks the merged
// fill order and the distance map must equal the serial march's, pixel for pixel.  Built and run by tests/test_march_host.py.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "telea_march.h"

using namespace ofxcv_telea;

static std::vector<uint8_t> random_mask(int w, int h, unsigned seed, int blobs, int rmax, bool touching) {
    std::mt19937 rng(seed);
    std::vector<uint8_t> m((size_t)w * h, 0);
    for (int b = 0; b < blobs; b++) {
        const int cx = rng() % w, cy = rng() % h, rx = 2 + rng() % rmax, ry = 2 + rng() % rmax;
        for (int y = std::max(0, cy - ry); y <= std::min(h - 1, cy + ry); y++)
            for (int x = std::max(0, cx - rx); x <= std::min(w - 1, cx + rx); x++) {
                const double dx = (x - cx) / (double)rx, dy = (y - cy) / (double)ry;
                if (dx * dx + dy * dy <= 1.0) m[(size_t)y * w + x] = 255;
            }
    }
    if (touching) {  // thin walls: one-pixel gaps between holes (band seeds shared by two components), diagonal contacts
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++)
                if ((x % 17 == 5 || y % 23 == 7) && m[(size_t)y * w + x]) m[(size_t)y * w + x] = 0;
    }
    return m;
}

static int run_case(int w, int h, unsigned seed, int blobs, int rmax, bool touching, bool ring, int range) {
    const std::vector<uint8_t> mask = random_mask(w, h, seed, blobs, rmax, touching);
    March a, b;
    a.prepare(w, h, range);
    b.prepare(w, h, range);
    const bool any_a = march_begin(mask.data(), ring, a), any_b = march_begin(mask.data(), ring, b);
    if (any_a != any_b) return 1;
    if (!any_a) return 0;
    while (march_advance(a, 1 << 30) > 0) {}
    const bool par = march_parallel_run(b, 1);
    int got = 0;
    while ((got = march_advance(b, 1000 + seed % 777)) > 0) {}  // handed out in odd portions like the pipelined fill does
    if (a.pix != b.pix) {
        size_t k = 0;
        while (k < a.pix.size() && k < b.pix.size() && a.pix[k] == b.pix[k]) k++;
        std::printf("case %dx%d seed %u: fill order differs at %zu of %zu / %zu (parallel %d, components %zu)\n", w, h, seed, k, a.pix.size(), b.pix.size(),
                    (int)par, b.par ? b.par->comps.size() : 0);
        return 1;
    }
    // (the parallel form leaves the host order map at "not yet": the order is the position in pix)
    if (a.t != b.t || a.filled != b.filled) {
        std::printf("case %dx%d seed %u: maps differ\n", w, h, seed);
        return 1;
    }
    return par ? 0 : -1;  // -1: one piece, the serial form ran
}

// a front that lives beyond the last bucket (distances >= 2048: a hole more than 4 000 pixels across): every push then belongs to the current
// bucket and goes through the queue's overflow heap -- same pop order as a binary heap, and no quadratic inserts (200 000 pushes here)
static int queue_far_case(unsigned seed) {
    std::mt19937 rng(seed);
    FrontQueue q;
    std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> ref;
    std::vector<std::pair<int, int>> who;
    float last = 2040.f;
    auto push = [&](float T) {
        uint32_t bits;
        std::memcpy(&bits, &T, 4);
        ref.push(((uint64_t)bits << 32) | (uint32_t)who.size());
        const int i = (int)(rng() % 5000), j = (int)(rng() % 5000);
        who.push_back({i, j});
        q.push(i, j, T);
    };
    for (int k = 0; k < 64; k++) push(last + (float)(rng() % 100) / 10.f);
    for (int step = 0; step < 200000; step++) {
        const uint64_t k = ref.top();
        ref.pop();
        int i = -1, j = -1;
        if (!q.pop(i, j) || i != who[(uint32_t)k].first || j != who[(uint32_t)k].second) return 1;
        const uint32_t bits = (uint32_t)(k >> 32);
        std::memcpy(&last, &bits, 4);
        push(last + 0.707f + (float)(rng() % 300) / 1000.f);
        if (rng() % 8 == 0) push(last);  // ties with what just popped
    }
    while (!ref.empty()) {
        const uint64_t k = ref.top();
        ref.pop();
        int i, j;
        if (!q.pop(i, j) || i != who[(uint32_t)k].first || j != who[(uint32_t)k].second) return 1;
    }
    int i, j;
    return q.pop(i, j) ? 1 : 0;
}

// the bucket queue against a binary heap on arbitrary push / pop sequences: equal distances (push order decides), pushes
// below the distance that popped last (never produced by a front, served all the same), distances beyond the last bucket
static int queue_case(unsigned seed) {
    std::mt19937 rng(seed);
    FrontQueue q;
    std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> ref;
    std::vector<std::pair<int, int>> who;
    float last = 0.f;
    for (int step = 0; step < 20000; step++) {
        const unsigned r = rng() % 100;
        if (r < 55 || ref.empty()) {
            float T;
            const unsigned kind = rng() % 20;
            if (kind == 0) T = last * (float)(rng() % 1000) / 1000.f;               // behind the front
            else if (kind == 1) T = last;                                            // a tie with what just popped
            else if (kind == 2) T = 1.0e6f + (float)(rng() % 3);                      // the last bucket
            else if (kind == 3) T = (float)(rng() % 64) / 16.f;                       // exact bucket edges, many ties
            else T = last + (float)(rng() % 4000) / 1000.f;
            const int i = (int)(rng() % 5000), j = (int)(rng() % 5000);
            uint32_t bits;
            std::memcpy(&bits, &T, 4);
            ref.push(((uint64_t)bits << 32) | (uint32_t)who.size());
            who.push_back({i, j});
            q.push(i, j, T);
        } else {
            const uint64_t k = ref.top();
            ref.pop();
            int i = -1, j = -1;
            if (!q.pop(i, j) || i != who[(uint32_t)k].first || j != who[(uint32_t)k].second) return 1;
            const uint32_t bits = (uint32_t)(k >> 32);
            std::memcpy(&last, &bits, 4);
            if (last > 1000.f) last = 0.f;
        }
    }
    while (!ref.empty()) {
        const uint64_t k = ref.top();
        ref.pop();
        int i, j;
        if (!q.pop(i, j) || i != who[(uint32_t)k].first || j != who[(uint32_t)k].second) return 1;
    }
    int i, j;
    if (q.pop(i, j)) return 1;
    q.clear();  // and again on the cleared queue
    q.push(3, 4, 2.5f);
    q.push(5, 6, 0.25f);
    if (!q.pop(i, j) || i != 5 || !q.pop(i, j) || i != 3 || q.pop(i, j)) return 1;
    return 0;
}

int main() {
    int bad = 0, parallel = 0, cases = 0;
    for (unsigned seed = 1; seed <= 40; seed++) {
        cases++;
        if (queue_case(seed)) {
            std::printf("bucket queue differs from the heap, seed %u\n", seed);
            bad++;
        }
    }
    for (unsigned seed = 1; seed <= 3; seed++) {
        cases++;
        if (queue_far_case(seed)) {
            std::printf("bucket queue (front beyond the last bucket) differs from the heap, seed %u\n", seed);
            bad++;
        }
    }
    for (unsigned seed = 1; seed <= 60; seed++) {
        const int w = 40 + (seed * 37) % 300, h = 30 + (seed * 53) % 200;
        for (int touching = 0; touching < 2; touching++) {
            const int r = run_case(w, h, seed, 3 + seed % 20, 4 + seed % 30, touching != 0, seed % 3 != 0, 1 + seed % 5);
            cases++;
            if (r > 0) bad++;
            if (r == 0) parallel++;
        }
    }
    // the second call on the same state (sparse resets) and a large frame
    {
        March m;
        for (int rep = 0; rep < 3; rep++) {
            const std::vector<uint8_t> mask = random_mask(640, 480, 100 + rep, 40, 30, rep == 1);
            March ref;
            ref.prepare(640, 480, 3);
            m.prepare(640, 480, 3);
            march_begin(mask.data(), true, ref);
            march_begin(mask.data(), true, m);
            while (march_advance(ref, 1 << 30) > 0) {}
            march_parallel_run(m, 1);
            while (march_advance(m, 8192) > 0) {}
            cases++;
            if (ref.pix != m.pix || ref.t != m.t) {
                std::printf("repeat %d on one state differs\n", rep);
                bad++;
            } else
                parallel++;
        }
    }
    std::printf("%d cases, %d marched in parallel, %d failed\n", cases, parallel, bad);
    return bad ? 1 : 0;
}
