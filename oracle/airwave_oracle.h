/*
 * airwave_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's HRIR convolution hot path
 * (sallliisa/Airwave, Swift + Apple vDSP).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may link or call this.
 * The shipped library (libairwave_hip.so) never does.
 *
 * Parity pin: the reference cannot be compiled or run here (Swift, closed
 * Accelerate.framework).  This restatement is pinned against every known-answer
 * test the reference holds for the path (AirwaveTests/ConvolutionEngineTests.swift:12-59,
 * AirwaveTests/RealtimeAudioProcessorTests.swift:59-126) and, for non-trivial
 * HRIRs where the reference holds no vector ("parity unpinned" by the reference),
 * against the mathematical definition: float64 direct convolution
 * (orc_direct_conv_f64).  See oracle/README.md.
 */
#ifndef AIRWAVE_ORACLE_H
#define AIRWAVE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ConvolutionEngine (Airwave/ConvolutionEngine.swift:14-408) ---------------- */
typedef struct orc_engine orc_engine;

/* init?(hrirSamples:blockSize:sharedFFTSetup:)  ConvolutionEngine.swift:68-197.
 * Returns NULL when blockSize is not a power of two >= 2 (the reference derives
 * log2n = Int(log2(2*blockSize)), :74, i.e. silently assumes one) or count <= 0. */
orc_engine *orc_engine_create(const float *hrir_samples, int count, int block_size);
void orc_engine_destroy(orc_engine *e);
/* process(input:output:)  :232-367 — exactly block_size frames in and out. */
void orc_engine_process(orc_engine *e, const float *input, float *output);
/* processAndAccumulate(input:outputAccumulator:)  :388-394 */
void orc_engine_process_accumulate(orc_engine *e, const float *input, float *accumulator);
/* reset()  :397-407 */
void orc_engine_reset(orc_engine *e);
int orc_engine_block_size(const orc_engine *e);
int orc_engine_partition_count(const orc_engine *e);

/* ---- RealtimeAudioProcessor (Airwave/RealtimeAudioProcessor.swift:11-191) ------- */
typedef struct orc_realtime orc_realtime;

/* renderers: n_renderers pairs of (left-ear IR, right-ear IR); every IR has ir_len
 * taps.  left_irs/right_irs are [n_renderers][ir_len] row-major.
 * two_renderer_cap != 0 reproduces processPendingBlock's min(renderers.count, 2)
 * cap fed from L/R (RealtimeAudioProcessor.swift:145-147). */
orc_realtime *orc_realtime_create(const float *left_irs, const float *right_irs,
                                  int n_renderers, int ir_len, int block_size,
                                  int max_frames_per_callback);
void orc_realtime_destroy(orc_realtime *p);
/* process(inputLeft:inputRight:leftOutput:rightOutput:frameCount:)  :77-119.
 * input_right may be NULL (mono duplication, :95-107).  Returns 0, or -1 when
 * frame_count exceeds max_frames_per_callback (the reference traps, :85). */
int orc_realtime_process(orc_realtime *p, const float *input_left, const float *input_right,
                         float *left_output, float *right_output, int frame_count);
void orc_realtime_reset(orc_realtime *p);

/* ---- N-speaker spatializer: the generalisation SURVEY.md §3.1 describes --------
 * out_L = sum_c conv(x_c, h[left_track[c]]),  out_R = sum_c conv(x_c, h[right_track[c]])
 * built from one orc_engine per (input channel, ear) exactly like
 * HRIRManager.activatePreset (HRIRManager.swift:366-418) builds renderers, summed
 * in renderer order with plain adds (RealtimeAudioProcessor.swift:152-163).
 * Channels whose track index is < 0 are skipped (HRIRManager.swift:370-372). */
typedef struct orc_spatializer orc_spatializer;

orc_spatializer *orc_spatializer_create(const float *tracks /* [n_tracks][taps] */, int n_tracks,
                                        int taps, int n_channels, const int32_t *left_track,
                                        const int32_t *right_track, int block_size);
void orc_spatializer_destroy(orc_spatializer *s);
/* One stream.  in: [frames][n_channels] interleaved, out: [frames][2] interleaved.
 * frames must be a multiple of block_size (returns -1 otherwise). */
int orc_spatializer_process(orc_spatializer *s, const float *in, float *out, int64_t frames);
void orc_spatializer_reset(orc_spatializer *s);

/* Batch helper for the CPU baseline: n_streams independent streams, one fresh
 * spatializer state per stream, run over `threads` OpenMP threads (one stream per
 * thread at a time).  in: [stream][frames][ch], out: [stream][frames][2]. */
int orc_spatializer_batch(const float *tracks, int n_tracks, int taps, int n_channels,
                          const int32_t *left_track, const int32_t *right_track, int block_size,
                          const float *in, float *out, int n_streams, int64_t frames, int threads);

/* ---- float64 truth ---------------------------------------------------------------
 * y[n] = sum_{k<taps} h[k] x[n-k], x[<0] = 0; n in [0, frames).  Accumulated in double. */
void orc_direct_conv_f64(const float *x, int64_t frames, int64_t x_stride, const float *h, int taps,
                         double *y, int accumulate);

/* ---- synthetic input: counter-based U(-0.5, 0.5) -------------------
 * value(stream, i) = top24(splitmix64((seed + stream) * GOLDEN + i)) / 2^24 - 0.5,
 * i = frame * n_channels + channel.  Same function as the HIP fill kernel. */
float orc_synth_value(uint64_t seed, uint64_t stream, uint64_t index);
void orc_synth_fill(float *dst, int n_streams, int64_t frames, int n_channels, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif

/* ==== Parametric EQ ("next" row) ==========================================
 * BiquadCoefficientBuilder.make (Airwave/BiquadCoefficientBuilder.swift:29-107), type: 0 peaking,
 * 1 lowShelf, 2 highShelf.  Returns 0 or the BiquadCoefficientError ordinal + 1
 * (1 invalidSampleRate, 2 invalidFrequency, 3 invalidQ, 4 nonFiniteInput, 5 nonFiniteCoefficients). */
#ifndef AIRWAVE_ORACLE_EQ_H
#define AIRWAVE_ORACLE_EQ_H
#ifdef __cplusplus
extern "C" {
#endif
typedef struct { double b0, b1, b2, a1, a2; } orc_biquad;
int orc_biquad_make(int type, double gain_db, double frequency_hz, double q, double sample_rate, orc_biquad *out);

/* ParametricEqualizerState (Airwave/ParametricEqualizerProcessor.swift:16-98): cascaded
 * transposed-DF2 biquads, Float64 state and arithmetic, preamp 10^(dB/20), |z| < 1e-30 flushed. */
typedef struct orc_eq_state orc_eq_state;
orc_eq_state *orc_eq_state_create(double sample_rate, double preamp_db, const orc_biquad *coefficients, int count);
void orc_eq_state_destroy(orc_eq_state *s);
void orc_eq_state_reset(orc_eq_state *s);
/* process(inputLeft:inputRight:leftOutput:rightOutput:frameCount:)  :58-91; input_right may be NULL;
 * in-place (output == input) is allowed, as the reference's canary test does. */
void orc_eq_state_process(orc_eq_state *s, const float *input_left, const float *input_right, float *left_output,
                          float *right_output, int frame_count);
#ifdef __cplusplus
}
#endif
#endif
