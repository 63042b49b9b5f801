// abi_replay.cpp — replays, through the C ABI only, the call order a Swift host would issue
// (create -> process x N -> reset -> process -> destroy) and dumps the samples for the parity
// test to compare with the oracle.  Built with plain g++ (no HIP headers): proves the boundary
// needs nothing but include/airwave_hip.h.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/airwave_hip.hpp"

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: abi_replay <hrir.wav> <out.bin>\n"); return 2; }
    try {
        aw::Context ctx(0);
        // WAVLoader.load -> layout -> map -> resolve  (HRIRManager.activatePreset :349-379)
        aw_wav *wav = nullptr;
        aw::check(aw_wav_load(argv[1], &wav));
        aw_layout *layout = nullptr;
        aw::check(aw_layout_detect(2, &layout));
        aw_channel_map *map = nullptr;
        aw::check(aw_map_hesuvi14(layout, &map));
        std::vector<int32_t> lt(2), rt(2);
        aw::check(aw_map_resolve(map, layout, aw_wav_channel_count(wav), lt.data(), rt.data()));
        aw::HRIR hrir(ctx, aw_wav_planar(wav), aw_wav_channel_count(wav), aw_wav_frame_count(wav), aw_wav_sample_rate(wav));
        aw::RealtimeAudioProcessor proc(ctx, hrir, lt, rt, 512, 4096);
        auto engine = aw::ConvolutionEngine::make(ctx, std::vector<float>(aw_wav_channel(wav, 0), aw_wav_channel(wav, 0) + aw_wav_frame_count(wav)), 512);
        if (!engine) return 3;

        std::vector<float> dump;
        unsigned state = 12345u;
        auto rnd = [&]() { state = state * 1664525u + 1013904223u; return (float)(state >> 8) / 16777216.0f - 0.5f; };
        const int sizes[] = {128, 512, 700, 4096, 1};
        for (int pass = 0; pass < 2; ++pass) {                 // second pass after reset must equal the first
            state = 12345u;
            for (int n : sizes) {
                std::vector<float> l(n), r(n), ol(n), orr(n);
                for (int i = 0; i < n; ++i) { l[i] = rnd(); r[i] = rnd(); }
                proc.process(l.data(), r.data(), ol.data(), orr.data(), n);
                dump.insert(dump.end(), ol.begin(), ol.end());
                dump.insert(dump.end(), orr.begin(), orr.end());
            }
            std::vector<float> blk(512), out(512);
            for (auto &v : blk) v = rnd();
            engine->process(blk.data(), out.data());
            dump.insert(dump.end(), out.begin(), out.end());
            proc.reset();
            engine->reset();
        }
        // the EQ that follows the spatializer in the reference's graph: parse an APO preset, fade to it across
        // 512-frame callbacks, retarget to unity (ParametricEqualizerProcessor.setTarget / process / drainRetiredStates)
        if (argc > 3) {
            FILE *pf = std::fopen(argv[3], "rb");
            if (!pf) return 5;
            std::vector<char> text(1 << 16);
            const size_t len = std::fread(text.data(), 1, text.size(), pf);
            std::fclose(pf);
            aw::EqualizerDefinition def = aw::EqualizerDefinition::parse(text.data(), len);
            aw::ParametricEqualizerProcessor eq(ctx, 48000.0, 512);
            eq.setTarget(&def);
            state = 777u;
            for (int call = 0; call < 6; ++call) {
                if (call == 4) { eq.drainRetiredStates(); eq.setTarget(nullptr); }
                std::vector<float> l(512), r(512), ol(512), orr(512);
                for (int i = 0; i < 512; ++i) { l[i] = rnd(); r[i] = rnd(); }
                eq.process(l.data(), r.data(), ol.data(), orr.data(), 512);
                dump.insert(dump.end(), ol.begin(), ol.end());
                dump.insert(dump.end(), orr.begin(), orr.end());
            }
        }
        aw_map_destroy(map); aw_layout_destroy(layout); aw_wav_destroy(wav);
        FILE *f = std::fopen(argv[2], "wb");
        if (!f) return 4;
        std::fwrite(dump.data(), sizeof(float), dump.size(), f);
        std::fclose(f);
        std::printf("ok %zu floats\n", dump.size());
        return 0;
    } catch (const aw::Error &e) {
        std::fprintf(stderr, "aw::Error %d: %s\n", (int)e.status, e.what());
        return 1;
    }
}
