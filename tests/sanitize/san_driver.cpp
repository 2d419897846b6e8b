// san_driver.cpp -- drives the three sanitizer-built OFX plugins through the in-repo mock host (tests/mock_host/mock_host.cpp, linked in) the
// way tests/test_ofx_boundary.py does from Python: load, describe, create instances, render -- VectorGenerator from several threads at once on
// ONE instance (eRenderFullySafe, VectorGenerator.cpp:108) with named frames, so the per-device named-frame cache and the submission queue
// run -- destroy, unload.  The plugins are linked against tests/sanitize/hip_stub.cpp: no kernel runs, outputs are not looked at; the run is
// clean when the sanitizers report nothing and every action returns its expected status.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" {
void *mh_open(const char *path);
int mh_load(void *h, int set_host);
int mh_describe(void *h, const char *context);
void *mh_create_instance(void *h, int *status);
int mh_destroy_instance(void *h, void *inst);
int mh_set_image(void *inst, const char *clip, double time, void *data, int x1, int y1, int x2, int y2, int row_bytes, const char *depth, const char *components,
                 double rsx, double rsy);
int mh_set_image_id(void *inst, const char *clip, double time, const char *id);
int mh_render(void *h, void *inst, double time, int x1, int y1, int x2, int y2, double rsx, double rsy);
int mh_action_raw(void *h, const char *action);
int mh_clip_balance(void *inst, const char *clip);
void mh_close(void *h);
}

static int fails = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            std::printf("FAILED: %s (line %d)\n", #cond, __LINE__);    \
            fails++;                                                   \
        }                                                              \
    } while (0)

int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int threads = argc > 2 ? std::atoi(argv[2]) : 8, rounds = argc > 3 ? std::atoi(argv[3]) : 3;
    // ---- VectorGenerator: concurrent renders of consecutive frames on one instance, images named
    {
        void *h = mh_open((dir + "/VectorGenerator.ofx").c_str());
        EXPECT(h != nullptr);
        if (!h) return 1;
        EXPECT(mh_load(h, 1) == 0);
        EXPECT(mh_describe(h, "OfxImageEffectContextFilter") == 0);
        int st = -1;
        void *inst = mh_create_instance(h, &st);
        EXPECT(inst && st == 0);
        const int w = 128, hh = 96, nframes = threads + 2;
        std::vector<std::vector<float>> src(nframes, std::vector<float>((size_t)w * hh * 4)), dst(nframes, std::vector<float>((size_t)w * hh * 4));
        for (int t = 0; t < nframes; t++) {
            for (size_t i = 0; i < src[t].size(); i++) src[t][i] = (float)((i * 2654435761u + t * 40503u) % 1000) / 1000.f;
            EXPECT(mh_set_image(inst, "Source", (double)t, src[t].data(), 0, 0, w, hh, w * 16, "OfxBitDepthFloat", "OfxImageComponentRGBA", 1.0, 1.0) == 0);
            char id[32];
            std::snprintf(id, sizeof id, "frame-%d", t);
            EXPECT(mh_set_image_id(inst, "Source", (double)t, id) == 0);
            EXPECT(mh_set_image(inst, "Output", (double)t, dst[t].data(), 0, 0, w, hh, w * 16, "OfxBitDepthFloat", "OfxImageComponentRGBA", 1.0, 1.0) == 0);
        }
        std::atomic<int> bad{0};
        for (int r = 0; r < rounds; r++) {
            std::vector<std::thread> th;
            for (int k = 0; k < threads; k++)
                th.emplace_back([&, k] {
                    if (mh_render(h, inst, (double)(1 + k), 0, 0, w, hh, 1.0, 1.0) != 0) bad++;
                });
            for (auto &t : th) t.join();
        }
        EXPECT(bad.load() == 0);
        EXPECT(mh_clip_balance(inst, "Source") == 0 && mh_clip_balance(inst, "Output") == 0);
        EXPECT(mh_destroy_instance(h, inst) == 0);
        EXPECT(mh_action_raw(h, "OfxActionUnload") == 0);
        mh_close(h);
    }
    // ---- inpaint and segment: instance-safe plugins, one instance per thread
    for (const char *name : {"inpaint", "segment"}) {
        void *h = mh_open((dir + "/" + name + ".ofx").c_str());
        EXPECT(h != nullptr);
        if (!h) return 1;
        EXPECT(mh_load(h, 1) == 0);
        EXPECT(mh_describe(h, "OfxImageEffectContextFilter") == 0);
        const int w = 64, hh = 64;  // (segment needs multiples of 1 << level)
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        for (int k = 0; k < 4; k++)
            th.emplace_back([&, k] {
                int st = -1;
                void *inst = mh_create_instance(h, &st);
                if (!inst || st != 0) { bad++; return; }
                std::vector<unsigned char> a((size_t)w * hh * 4), b((size_t)w * hh * 4);
                for (size_t i = 0; i < a.size(); i++) a[i] = (unsigned char)(1 + (i * 7 + k) % 250);
                for (int y = 20; y < 30; y++)
                    for (int x = 20; x < 34; x++) std::memset(&a[((size_t)y * w + x) * 4], 0, 3);  // a hole
                if (mh_set_image(inst, "Source", 0.0, a.data(), 0, 0, w, hh, w * 4, "OfxBitDepthByte", "OfxImageComponentRGBA", 1.0, 1.0) != 0) bad++;
                if (mh_set_image(inst, "Output", 0.0, b.data(), 0, 0, w, hh, w * 4, "OfxBitDepthByte", "OfxImageComponentRGBA", 1.0, 1.0) != 0) bad++;
                for (int r = 0; r < 3; r++)
                    if (mh_render(h, inst, 0.0, 0, 0, w, hh, 1.0, 1.0) != 0) bad++;
                if (mh_destroy_instance(h, inst) != 0) bad++;
            });
        for (auto &t : th) t.join();
        EXPECT(bad.load() == 0);
        mh_close(h);
    }
    std::printf("san_driver: %d failed expectations\n", fails);
    return fails ? 1 : 0;
}
