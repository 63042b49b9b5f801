/* offline_batch_eq.c — the C ABI from plain C99: spatializer AND equalizer of a batch in ONE pass over the audio.
 *
 * The reference's graph runs two effects per render callback — the spatial effect, then the parametric equalizer on its output
 * (AudioEffectGraph.swift:195-211).  A batch host knows the equalizer preset up front, and between two target changes the equalizer is a
 * linear time-invariant filter, so it folds it into the HRIR once, at activation (aw_eq_fold_hrir: EQ(x * h) = x * (h * g)), and the
 * convolution kernels apply both.  Steps: WAVLoader.load -> hesuvi14Channel map -> resolve (the per-speaker loop of
 * HRIRManager.activatePreset, HRIRManager.swift:347-446) -> EqualizerAPOParser.parse -> fold -> spatializer on the folded tracks.
 * If the equalizer's response does not decay within the allowed length (AW_ERR_EQ_NOT_FOLDABLE) the host falls back to the reference's
 * structure: spatializer on the plain HRIR, then aw_eq_state_process on its output (device buffers; not shown here).
 *
 *   cc -std=c99 -O2 -Iinclude examples/offline_batch_eq.c -Lairwave_amd -lairwave_hip -Wl,-rpath,$PWD/airwave_amd -Wl,-rpath,/opt/rocm/lib -lm -o offline_batch_eq
 *   ./offline_batch_eq tests/golden/hrtf/StageSH1.0.wav "tests/golden/eq/CCA CRA ParametricEq.txt" [streams] [seconds] [out.f32]
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include "airwave_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        aw_status st_ = (call);                                                                       \
        if (st_ != AW_OK) {                                                                           \
            fprintf(stderr, "%s: %s (%s)\n", #call, aw_status_string(st_), aw_last_error_message()); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s hrir.wav equalizer_apo.txt [streams] [seconds] [out.f32]\n", argv[0]);
        return 2;
    }
    const int streams = argc > 3 ? atoi(argv[3]) : 16;
    const double seconds = argc > 4 ? atof(argv[4]) : 1.0;
    const int channels = 8;                                           /* 7.1 */
    if (streams < 1 || seconds <= 0.0) return 2;

    /* the equalizer preset: the library's Equalizer APO parser (EqualizerAPOParser.swift:36-151) */
    FILE *f = fopen(argv[2], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
    char text[65536];
    const size_t n_text = fread(text, 1, sizeof text, f);
    fclose(f);
    aw_eq_definition *eq = NULL;
    char issues[1024];
    if (aw_eq_parse(text, n_text, &eq, issues, sizeof issues) != AW_OK) { fprintf(stderr, "equalizer preset: %s\n", issues); return 1; }

    /* the HRIR and the channel map of a 7.1 layout (host only so far) */
    aw_wav *wav = NULL;
    aw_layout *layout = NULL;
    aw_channel_map *map = NULL;
    int32_t lt[8], rt[8];
    CHECK(aw_wav_load(argv[1], &wav));
    CHECK(aw_layout_detect(channels, &layout));
    CHECK(aw_map_hesuvi14(layout, &map));
    CHECK(aw_map_resolve(map, layout, aw_wav_channel_count(wav), lt, rt));
    const double rate = aw_wav_sample_rate(wav);                      /* the batch runs at the HRIR's own rate: no resampling here */
    const int32_t n_tracks = aw_wav_channel_count(wav), taps = aw_wav_frame_count(wav);
    const int64_t frames = (int64_t)(seconds * rate);

    /* fold: the length first, then the tracks */
    int32_t folded_taps = 0, response = 0;
    double tail = 0.0;
    CHECK(aw_eq_fold_hrir(eq, rate, aw_wav_planar(wav), n_tracks, taps, 1e-7, 65536, NULL, &folded_taps, &response, &tail));
    float *folded = (float *)malloc((size_t)n_tracks * (size_t)folded_taps * sizeof(float));
    if (!folded) return 1;
    CHECK(aw_eq_fold_hrir(eq, rate, aw_wav_planar(wav), n_tracks, taps, 1e-7, 65536, folded, &folded_taps, &response, &tail));

    /* from here on a device is needed */
    aw_context *ctx = NULL;
    aw_hrir *hrir = NULL;
    aw_spatializer *sp = NULL;
    CHECK(aw_context_create(0, &ctx));
    CHECK(aw_hrir_create(ctx, folded, n_tracks, folded_taps, rate, &hrir));
    CHECK(aw_spatializer_create(ctx, hrir, channels, lt, rt, streams, 0, &sp));
    CHECK(aw_spatializer_reserve_host(sp, frames));

    const size_t n_in = (size_t)streams * (size_t)frames * channels, n_out = (size_t)streams * (size_t)frames * 2;
    float *in = NULL, *out = NULL;
    CHECK(aw_host_alloc_pinned(ctx, n_in * sizeof(float), (void **)&in));
    CHECK(aw_host_alloc_pinned(ctx, n_out * sizeof(float), (void **)&out));
    uint32_t s = 12345u;
    for (size_t i = 0; i < n_in; ++i) {
        s = s * 1664525u + 1013904223u;
        in[i] = (float)(s >> 8) / 16777216.0f - 0.5f;
    }
    CHECK(aw_spatializer_process_host(sp, in, out, frames));         /* spatializer + equalizer, one pass */

    double sum = 0.0;
    for (size_t i = 0; i < n_out; ++i) sum += (double)out[i] * (double)((i % 251) + 1);
    printf("streams %d frames %lld hrir taps %d + equalizer response %d -> %d taps (tail bound %.3e) checksum %.9e\n", streams, (long long)frames,
           (int)taps, (int)response, (int)folded_taps, tail, sum);
    if (argc > 5) {
        FILE *o = fopen(argv[5], "wb");
        if (!o || fwrite(out, sizeof(float), n_out, o) != n_out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
        fclose(o);
    }
    CHECK(aw_host_free_pinned(ctx, in));
    CHECK(aw_host_free_pinned(ctx, out));
    aw_spatializer_destroy(sp);
    aw_hrir_destroy(hrir);
    aw_context_destroy(ctx);
    free(folded);
    aw_map_destroy(map);
    aw_layout_destroy(layout);
    aw_wav_destroy(wav);
    aw_eq_definition_destroy(eq);
    return 0;
}
