/* offline_batch.c — the C ABI from plain C99: an offline batch host.
 *
 *   N streams of interleaved 7.1 PCM in host memory  ->  HeSuVi HRIR preset (a 14-track .wav)  ->  N stereo streams in host memory.
 *
 * What the Swift product does per render callback (HRIRManager.activatePreset, HRIRManager.swift:347-446, then
 * StereoAudioProcessing.process, AudioPipeline.swift:3-11) done once for a whole batch: activate the preset, size every
 * buffer ahead of time (process then allocates nothing), hand over page-locked host buffers, read the result.
 *
 *   cc -std=c99 -O2 -Iinclude examples/offline_batch.c -Lairwave_amd -lairwave_hip -Wl,-rpath,$PWD/airwave_amd -Wl,-rpath,/opt/rocm/lib -lm -o offline_batch
 *   ./offline_batch tests/golden/hrtf/RoomSH1.0.wav [streams] [seconds] [out.f32]
 *
 * Prints one line: streams, frames, seconds of wall time, G stereo frames/s (PCIe inclusive) and a checksum of the output;
 * with a fourth argument the output samples are written as raw float32 (tests/test_example_host.py compares them with the
 * oracle's float64 convolution). */
#define _POSIX_C_SOURCE 199309L        /* clock_gettime under -std=c99 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <time.h>

#include "airwave_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        aw_status st_ = (call);                                                                       \
        if (st_ != AW_OK) {                                                                           \
            fprintf(stderr, "%s: %s (%s)\n", #call, aw_status_string(st_), aw_last_error_message()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: %s hrir.wav [streams] [seconds] [out.f32]\n", argv[0]);
        return 2;
    }
    const int streams = argc > 2 ? atoi(argv[2]) : 64;
    const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
    const double rate = 48000.0;
    const int64_t frames = (int64_t)(seconds * rate);
    const int channels = 8;                                           /* 7.1: FL FR FC LFE BL BR SL SR (VirtualSpeaker.swift:76-79) */
    if (streams < 1 || frames < 1) return 2;

    aw_context *ctx = NULL;
    aw_layout *layout = NULL;
    aw_spatializer *sp = NULL;
    CHECK(aw_context_create(0, &ctx));
    CHECK(aw_layout_detect(channels, &layout));
    /* no custom map: 14 tracks -> HRIRChannelMap.hesuvi14Channel, 7 tracks -> hesuvi7Channel (HRIRManager.swift:355-360) */
    CHECK(aw_preset_activate(ctx, argv[1], rate, layout, NULL, streams, &sp, NULL));
    CHECK(aw_spatializer_reserve_host(sp, frames));                   /* creation may block; process must not allocate */

    const size_t n_in = (size_t)streams * (size_t)frames * channels, n_out = (size_t)streams * (size_t)frames * 2;
    float *in = NULL, *out = NULL;
    CHECK(aw_host_alloc_pinned(ctx, n_in * sizeof(float), (void **)&in));
    CHECK(aw_host_alloc_pinned(ctx, n_out * sizeof(float), (void **)&out));
    uint32_t s = 12345u;                                              /* any PCM: a linear congruential sequence in [-0.5, 0.5) */
    for (size_t i = 0; i < n_in; ++i) {
        s = s * 1664525u + 1013904223u;
        in[i] = (float)(s >> 8) / 16777216.0f - 0.5f;
    }

    CHECK(aw_spatializer_process_host(sp, in, out, frames));         /* first call: also loads the kernels */
    CHECK(aw_spatializer_reset(sp));
    CHECK(aw_context_synchronize(ctx));
    const double t0 = now_s();
    CHECK(aw_spatializer_process_host(sp, in, out, frames));         /* synchronous: the samples are in `out` on return */
    const double dt = now_s() - t0;

    double sum = 0.0;
    for (size_t i = 0; i < n_out; ++i) sum += (double)out[i] * (double)((i % 251) + 1);
    printf("streams %d frames %lld wall %.4f s %.3f Gframes/s checksum %.9e window %lld hop %lld\n", streams, (long long)frames, dt,
           (double)streams * (double)frames / dt / 1e9, sum, (long long)aw_spatializer_info(sp, 0), (long long)aw_spatializer_info(sp, 1));
    if (argc > 4) {
        FILE *f = fopen(argv[4], "wb");
        if (!f || fwrite(out, sizeof(float), n_out, f) != n_out) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
        fclose(f);
    }
    CHECK(aw_host_free_pinned(ctx, in));
    CHECK(aw_host_free_pinned(ctx, out));
    aw_spatializer_destroy(sp);
    aw_layout_destroy(layout);
    aw_context_destroy(ctx);
    return 0;
}
