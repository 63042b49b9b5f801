/*
 * airwave_hip.h — C ABI of the MI355X-native batch HRIR spatializer (libairwave_hip.so).
 *
 * This is the drop-in boundary for ONE path of sallliisa/Airwave: the per-virtual-speaker
 * partitioned FFT convolution + stereo downmix (ConvolutionEngine / VirtualSpeaker /
 * RealtimeAudioProcessor / HRIRManager.activatePreset).  Every entry point cites the reference
 * interface it replaces (paths relative to the reference repository root).  Host code in any
 * language binds these symbols (Swift module map + wrapper: swift/; ctypes: airwave_amd/;
 * C++ RAII mirror: include/airwave_hip.hpp).  See INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; opaque handles; every function returns aw_status (0 = ok);
 *    no exceptions or callbacks cross the boundary; aw_last_error_message() gives detail.
 *  - a handle is single-threaded for process/reset (like ConvolutionEngine: "not thread-safe,
 *    one owner"); independent handles are independent.
 *  - creation may block and allocate (the reference builds engines on a background queue,
 *    HRIRManager.swift:347); aw_spatializer_process on device buffers does not allocate once
 *    aw_spatializer_reserve has sized the handle (or it has seen a call of that size class:
 *    scratch is grow-only).  The scratch of the multi-kernel paths is a pool of the CONTEXT,
 *    shared by its spatializers; a handle is still single-owner, handles of one context may be
 *    driven from different threads (their launch sequences are serialised).
 *  - there is NO CPU fallback: without a HIP device every create returns AW_ERR_NO_DEVICE.
 */
#ifndef AIRWAVE_HIP_H
#define AIRWAVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AW_API __attribute__((visibility("default")))

/* ---- status codes ---------------------------------------------------------------------------- */
typedef int32_t aw_status;
enum {
    AW_OK = 0,
    AW_ERR_INVALID_ARGUMENT = 1,         /* Swift precondition failures (RealtimeAudioProcessor.swift:35-36,85) */
    AW_ERR_OUT_OF_MEMORY = 2,
    AW_ERR_HIP = 3,                      /* any HIP runtime error; message has hipGetErrorString */
    AW_ERR_NO_DEVICE = 4,                /* no HIP device / ordinal out of range */
    AW_ERR_INVALID_CHANNEL_MAPPING = 5,  /* HRIRError.invalidChannelMapping  HRIRManager.swift:375-379 */
    AW_ERR_CONVOLUTION_SETUP_FAILED = 6, /* HRIRError.convolutionSetupFailed HRIRManager.swift:406-409,420-422 */
    AW_ERR_INVALID_CHANNEL_COUNT = 7,    /* WAVError/HRIRError.invalidChannelCount WAVLoader.swift:40-42, HRIRManager.swift:223 */
    AW_ERR_WAV_FILE_READ = 8,            /* WAVError.fileReadError   WAVLoader.swift:31-33,59-61 */
    AW_ERR_WAV_EMPTY_FILE = 9,           /* WAVError.emptyFile       WAVLoader.swift:44-46 */
    AW_ERR_WAV_UNSUPPORTED_FORMAT = 10,  /* WAVError.unsupportedFormat WAVLoader.swift:89-91 */
    AW_ERR_BLOCK_SIZE_MISMATCH = 11,     /* ConvolutionEngine.process(input:output:frameCount:) guard, ConvolutionEngine.swift:372 */
    AW_ERR_EQ_PARSE = 12,                /* EqualizerParseError                     EqualizerAPOParser.swift:8-21 */
    AW_ERR_EQ_INVALID_SAMPLE_RATE = 13,  /* ParametricEqualizerPreparationError.invalidSampleRate  ParametricEqualizerProcessor.swift:100-101 */
    AW_ERR_EQ_NON_FINITE_PREAMP = 14,    /* .nonFinitePreamp  :102 */
    AW_ERR_EQ_TOO_MANY_FILTERS = 15,     /* .tooManyFilters   :103 (also the maxFramesPerCallback guard, :148-150) */
    AW_ERR_EQ_INVALID_FILTER = 16,       /* .invalidFilter(index:error:)  :104; BiquadCoefficientError  BiquadCoefficientBuilder.swift:11-16 */
    AW_ERR_EQ_NOT_FOLDABLE = 17          /* aw_eq_fold_hrir: the equalizer's impulse response does not decay to the tolerance within the allowed length */
};
AW_API const char *aw_status_string(aw_status s);
AW_API const char *aw_last_error_message(void); /* thread-local, valid until the next failing call */
/* The structured part of the calling thread's last AW_ERR_EQ_INVALID_FILTER — what EqualizerRuntimeEffect.map builds
 * EqualizerAudioEffectError.invalidFilter(line:reason:) from (EqualizerRuntimeEffect.swift:80-100): the index among the ENABLED filters
 * (ParametricEqualizerPreparationError.invalidFilter(index:error:), ParametricEqualizerProcessor.swift:188-202), the
 * BiquadCoefficientError kind (aw_biquad_make) and that filter's sourceLine (0 if the definition carried none).  Returns 1 and fills
 * the non-NULL outputs, or 0 when the thread's last failing call was something else. */
AW_API int32_t aw_last_eq_filter_error(int32_t *enabled_index, int32_t *error_kind, int32_t *source_line);

/* ---- context: device + stream + twiddle tables -------------------------------------------------
 * Replaces FFTSetupManager.shared.getSetup(log2n:) (FFTSetupManager.swift:41-60): the twiddle
 * table is built once per context and shared by every spatializer/engine created on it. */
typedef struct aw_context aw_context;
AW_API aw_status aw_context_create(int32_t device_ordinal, aw_context **out);
/* Same, but launches on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
AW_API aw_status aw_context_create_on_stream(int32_t device_ordinal, void *hip_stream, aw_context **out);
AW_API void aw_context_destroy(aw_context *ctx);
AW_API aw_status aw_context_synchronize(aw_context *ctx);
AW_API void *aw_context_stream(aw_context *ctx);          /* the hipStream_t kernels are launched on */
/* HIP-event stopwatch on the context's stream (bench.py times the kernels with these). */
AW_API aw_status aw_context_timer_start(aw_context *ctx);
AW_API aw_status aw_context_timer_stop(aw_context *ctx, float *elapsed_ms); /* records, syncs, returns ms */

/* The HBM scratch of the multi-kernel paths (long HRIRs, long calls) is ONE grow-only pool per context, shared by the
 * spatializers created on it.  aw_spatializer_reserve grows it as needed; a host that knows its largest batch can size it at
 * start-up instead (one large hipMalloc, whose wall time varies from 0.2 ms to seconds on MI355X boxes).  The analogue of
 * allocating every engine buffer in ConvolutionEngine.init (ConvolutionEngine.swift:97-138), hoisted to the context. */
AW_API aw_status aw_context_reserve_scratch(aw_context *ctx, size_t bytes);
AW_API size_t aw_context_scratch_bytes(const aw_context *ctx);

/* Measured ceilings of the device (bench / diagnostics; blocks, allocates its own buffers, never on a process path).
 * aw_context_bandwidth_probe: a read-only, a write-only and a copy kernel over `bytes` (>= 64 MiB) of HBM each, best of
 * `repetitions`; GB/s, the copy's figure counting bytes read + bytes written.  SURVEY.md 8d asks for this next to the
 * vendor peak the roofline is priced on.  aw_context_pcie_probe: page-locked host memory to the device, back, and both at
 * once (GB/s; duplex = bytes both ways / time) — the yardstick for aw_spatializer_process_host. */
AW_API aw_status aw_context_bandwidth_probe(aw_context *ctx, size_t bytes, int32_t repetitions, double *read_gbs,
                                            double *write_gbs, double *copy_gbs);
AW_API aw_status aw_context_pcie_probe(aw_context *ctx, size_t bytes, int32_t repetitions, double *h2d_gbs, double *d2h_gbs,
                                       double *duplex_gbs);

/* Device memory helpers for hosts without their own HIP bindings (Swift, ctypes). */
AW_API aw_status aw_device_alloc(aw_context *ctx, size_t bytes, void **dptr);
AW_API aw_status aw_device_free(aw_context *ctx, void *dptr);
AW_API aw_status aw_memcpy_h2d(aw_context *ctx, void *dst_device, const void *src_host, size_t bytes);
AW_API aw_status aw_memcpy_d2h(aw_context *ctx, void *dst_host, const void *src_device, size_t bytes);
/* Page-locked host memory (hipHostMalloc): buffers from here cross PCIe by DMA, without the HIP runtime's staging copy, when
 * handed to the host entries below.  The analogue of the caller-owned render buffers of AudioPipeline.swift:3-11. */
AW_API aw_status aw_host_alloc_pinned(aw_context *ctx, size_t bytes, void **ptr);
AW_API aw_status aw_host_free_pinned(aw_context *ctx, void *ptr);

/* ---- HRIR set --------------------------------------------------------------------------------
 * The planar impulse responses a preset provides (WAVData.audioData, WAVLoader.swift:12-17):
 * tracks is [n_tracks][taps] float32 on the HOST. */
typedef struct aw_hrir aw_hrir;
AW_API aw_status aw_hrir_create(aw_context *ctx, const float *tracks, int32_t n_tracks, int32_t taps,
                                double sample_rate, aw_hrir **out);
AW_API void aw_hrir_destroy(aw_hrir *h);
AW_API int32_t aw_hrir_track_count(const aw_hrir *h);
AW_API int32_t aw_hrir_taps(const aw_hrir *h);
AW_API double aw_hrir_sample_rate(const aw_hrir *h);

/* ---- batch spatializer -----------------------------------------------------------------------
 * n_streams independent streams, each = the engine network HRIRManager.activatePreset builds
 * (one ConvolutionEngine per (input channel, ear), HRIRManager.swift:366-418) plus the downmix
 * of RealtimeAudioProcessor.processPendingBlock (RealtimeAudioProcessor.swift:141-164) without
 * its two-renderer cap:  out_L = sum_c x_c * h[left_track[c]],  out_R = sum_c x_c * h[right_track[c]].
 * A channel with left_track[c] < 0 or right_track[c] < 0 is skipped (HRIRManager.swift:370-372);
 * an index >= n_tracks is AW_ERR_INVALID_CHANNEL_MAPPING (:375-379); no mapped channel at all is
 * AW_ERR_CONVOLUTION_SETUP_FAILED (:420-422).  block_hint: 0 = automatic (results do not depend on it). */
typedef struct aw_spatializer aw_spatializer;
AW_API aw_status aw_spatializer_create(aw_context *ctx, const aw_hrir *hrir, int32_t n_in_channels,
                                       const int32_t *left_track, const int32_t *right_track,
                                       int32_t n_streams, int32_t block_hint, aw_spatializer **out);
AW_API void aw_spatializer_destroy(aw_spatializer *sp);
/* Offline/batch entry: DEVICE buffers.  in: [stream][frames][n_in_channels] interleaved float32,
 * out: [stream][frames][2].  Any frames >= 1; state (the convolution tail) carries over to the
 * next call exactly like consecutive ConvolutionEngine.process calls.  Asynchronous on the
 * context stream. */
AW_API aw_status aw_spatializer_process(aw_spatializer *sp, const float *in_device, float *out_device, int64_t frames);
/* Same with HOST buffers; synchronous.  A multi-stream batch crosses PCIe in chunks of streams, double buffered on three HIP
 * streams (H2D of chunk k+1 || kernels of chunk k || D2H of chunk k-1; streams are independent, so chunks are); page-locked
 * buffers (aw_host_alloc_pinned) move by DMA directly, pageable ones are bounced through page-locked chunks by host copy threads.  Small batches and
 * single streams (the plug-in shaped calls of aw_engine_* / aw_realtime_*) go in one piece.
 * The call holds the context's launch lock from entry to return (other handles of the context wait for its whole PCIe time); on an
 * error the contents of out_host are unspecified and the streams' state is that of a failed call: aw_spatializer_reset before reuse.
 * Without aw_spatializer_reserve_host the first call also creates the pipeline's streams, events and copy threads. */
AW_API aw_status aw_spatializer_process_host(aw_spatializer *sp, const float *in_host, float *out_host, int64_t frames);
/* aw_spatializer_reserve plus the device-side staging of the host entry for calls of up to max_frames frames (two chunks
 * each way): afterwards aw_spatializer_process_host does not allocate either.  (The scratch pool is sized as aw_spatializer_reserve
 * sizes it — for the whole batch — although the host entry only ever runs one staged chunk of streams at a time; AW_SPEC_SCRATCH_MB
 * bounds it for hosts that never use the device entry.) */
AW_API aw_status aw_spatializer_reserve_host(aw_spatializer *sp, int64_t max_frames);
/* StereoAudioProcessing.process shape (AudioPipeline.swift:3-11) for a 1-stream, 2-channel
 * spatializer: planar HOST buffers, input_right may be NULL (mono duplication). Zero latency. */
AW_API aw_status aw_spatializer_process_planar(aw_spatializer *sp, const float *input_left, const float *input_right,
                                               float *output_left, float *output_right, int32_t frame_count);
/* Sizes every internal device buffer for calls of up to max_frames frames, so that the process entries never
 * allocate afterwards (the reference allocates all engine state in ConvolutionEngine.init, ConvolutionEngine.swift:97-138,
 * and nothing in process; creation may block, process must not).  Optional: without it buffers grow on first use. */
AW_API aw_status aw_spatializer_reserve(aw_spatializer *sp, int64_t max_frames);
/* ConvolutionEngine.reset() for every engine of every stream (ConvolutionEngine.swift:397-407). */
AW_API aw_status aw_spatializer_reset(aw_spatializer *sp);
AW_API int32_t aw_spatializer_stream_count(const aw_spatializer *sp);
AW_API int32_t aw_spatializer_channel_count(const aw_spatializer *sp);
/* Introspection for benches/tests: 0 fft length, 1 hop, 2 partitions, 3 path (0 fused, 1 partitioned),
 * 4 history frames, 5 output frames produced by the launch that aw_spatializer_kernel_time() times
 * (the interior-tile launch of the last call; the few boundary tiles are a second, untimed launch),
 * 6 bytes of grow-only internal device buffers currently allocated (sized by aw_spatializer_reserve or by the largest call so far),
 * 7 rows R of the last call's windows when it ran on the long-window kernels (windows of R x 4096 frames; 0: it ran on the
 *   fused / partitioned kernels that 0-3 describe).  The kernel set is chosen per call; results do not depend on it.
 * 8 rows of the remainder window of a call that ran as two groups of windows, 9 long-window table sets built so far,
 * 10 / 11 / 12 microseconds the last aw_spatializer_reserve spent on the float64 table build (host threads) / the table upload /
 *   growing the context's scratch pool, 13 device or page-locked allocations and 14 blocking table uploads made so far on behalf
 *   of the context's handles (a reserved process path makes neither), 15 streams per staged chunk of the last host-entry call. */
AW_API int64_t aw_spatializer_info(const aw_spatializer *sp, int32_t what);
/* Average device time of the dominant kernel over the launches since the last call (HIP events
 * on the context stream); used for bench.py's roofline object.  Returns launches counted. */
AW_API aw_status aw_spatializer_set_profiling(aw_spatializer *sp, int32_t enabled);
AW_API int32_t aw_spatializer_kernel_time(aw_spatializer *sp, double *avg_ms, const char **kernel_name);
/* While profiling is on every kernel launch of the partitioned path (and the history carry of every path) is bracketed by
 * HIP events of its own; this iterates the per-kernel sums since profiling was switched on: index 0, 1, ... until it
 * returns 0.  total_ms / launches are over all calls; names are static strings. */
AW_API int32_t aw_spatializer_stage_time(aw_spatializer *sp, int32_t index, const char **name, double *total_ms, int32_t *launches);

/* Diagnostic builds only (library compiled with -DAW_STAMPS=1, see tools/archive/stamps.py): copies the
 * per-workgroup phase time stamps of the last fused-kernel launch to host_out as
 * [workgroup][16] uint64 shader-clock values.  The shipped library returns AW_ERR_INVALID_ARGUMENT. */
AW_API aw_status aw_spatializer_debug_stamps(aw_spatializer *sp, uint64_t *host_out, int64_t capacity_words,
                                             int64_t *n_workgroups);

/* ---- mono engine: ConvolutionEngine (ConvolutionEngine.swift:14-408) ---------------------------
 * init?(hrirSamples:blockSize:) :68 / process(input:output:) :232 / processAndAccumulate :388 /
 * reset() :397.  HOST buffers of exactly block_size frames. */
typedef struct aw_engine aw_engine;
AW_API aw_status aw_engine_create(aw_context *ctx, const float *hrir_samples, int32_t count, int32_t block_size,
                                  aw_engine **out);
AW_API void aw_engine_destroy(aw_engine *e);
AW_API aw_status aw_engine_process(aw_engine *e, const float *input, float *output);
/* frame_count != block_size returns AW_ERR_BLOCK_SIZE_MISMATCH and leaves output untouched
 * (the Swift array wrapper silently returns, ConvolutionEngine.swift:370-373). */
AW_API aw_status aw_engine_process_n(aw_engine *e, const float *input, float *output, int32_t frame_count);
AW_API aw_status aw_engine_process_accumulate(aw_engine *e, const float *input, float *output_accumulator);
AW_API aw_status aw_engine_reset(aw_engine *e);
AW_API int32_t aw_engine_block_size(const aw_engine *e);

/* ---- callback-size adapter: RealtimeAudioProcessor (RealtimeAudioProcessor.swift:11-191) --------
 * Renderer r convolves with hrir tracks (left_track[r], right_track[r]); like the reference only
 * the first min(n_renderers, 2) renderers run, fed from the left / right input (:145-147).
 * Pending-buffer + FIFO semantics, latency block_size - callback frames, silence on underflow. */
typedef struct aw_realtime aw_realtime;
AW_API aw_status aw_realtime_create(aw_context *ctx, const aw_hrir *hrir, int32_t n_renderers,
                                    const int32_t *left_track, const int32_t *right_track, int32_t block_size,
                                    int32_t max_frames_per_callback, aw_realtime **out);
AW_API void aw_realtime_destroy(aw_realtime *p);
AW_API aw_status aw_realtime_process(aw_realtime *p, const float *input_left, const float *input_right,
                                     float *left_output, float *right_output, int32_t frame_count);
AW_API aw_status aw_realtime_reset(aw_realtime *p);
/* Everything a callback needs is allocated by aw_realtime_create (RealtimeAudioProcessor.init, :30-62); this lets a host or a
 * test check it: 0 bytes of host buffer capacity held, 1 bytes of grow-only device buffers, 2 device allocations made so far
 * on the context.  None of them moves across aw_realtime_process calls. */
AW_API int64_t aw_realtime_info(const aw_realtime *p, int32_t what);

/* ---- host-side data model (no GPU needed) ----------------------------------------------------- */
/* WAVLoader.load (WAVLoader.swift:26-99): any RIFF/WAVE -> planar float32. */
typedef struct aw_wav aw_wav;
AW_API aw_status aw_wav_load(const char *path, aw_wav **out);
AW_API void aw_wav_destroy(aw_wav *w);
AW_API double aw_wav_sample_rate(const aw_wav *w);
AW_API int32_t aw_wav_channel_count(const aw_wav *w);
AW_API int32_t aw_wav_frame_count(const aw_wav *w);
AW_API const float *aw_wav_channel(const aw_wav *w, int32_t channel);   /* frame_count floats */
AW_API const float *aw_wav_planar(const aw_wav *w);                     /* [channels][frames] */

/* InputLayout (VirtualSpeaker.swift:59-100).  Speakers are identified by their case name
 * ("FL", "FR", "FC", "LFE", "BL", "BR", "SL", "SR", "TFL", ... ) or, for .custom, the custom name. */
typedef struct aw_layout aw_layout;
AW_API aw_status aw_layout_detect(int32_t channel_count, aw_layout **out);      /* InputLayout.detect :88-99 */
AW_API aw_status aw_layout_create(const char *const *speaker_names, int32_t count, const char *name, aw_layout **out);
AW_API void aw_layout_destroy(aw_layout *l);
AW_API int32_t aw_layout_count(const aw_layout *l);
AW_API const char *aw_layout_speaker(const aw_layout *l, int32_t index);
AW_API const char *aw_layout_name(const aw_layout *l);

/* HRIRChannelMap (VirtualSpeaker.swift:103-347): speaker -> (left-ear track, right-ear track). */
typedef struct aw_channel_map aw_channel_map;
AW_API aw_status aw_map_hesuvi14(const aw_layout *speakers, aw_channel_map **out);          /* :270-297 */
AW_API aw_status aw_map_hesuvi7(const aw_layout *speakers, aw_channel_map **out);           /* :224-250 */
AW_API aw_status aw_map_interleaved_pairs(const aw_layout *speakers, aw_channel_map **out); /* :126-159 */
AW_API aw_status aw_map_split_blocks(const aw_layout *speakers, aw_channel_map **out);      /* :200-209 */
AW_API aw_status aw_map_parse_text(const char *text, aw_channel_map **out);                 /* parseHeSuViFormat :301-346 */
AW_API void aw_map_destroy(aw_channel_map *m);
AW_API int32_t aw_map_count(const aw_channel_map *m);
/* getIndices(for:) :115-117 — returns 1 and fills the indices when the speaker is mapped, else 0. */
AW_API int32_t aw_map_get(const aw_channel_map *m, const char *speaker, int32_t *left_ear, int32_t *right_ear);
/* The per-speaker loop of activatePreset (HRIRManager.swift:366-379,420-422): fills
 * left_track/right_track[layout count] with -1 for unmapped speakers. */
AW_API aw_status aw_map_resolve(const aw_channel_map *m, const aw_layout *layout, int32_t n_tracks,
                                int32_t *left_track, int32_t *right_track);

/* Resampler.resampleHighQuality (Resampler.swift:31-68), the INTENDED linear interpolation
 * out[i] = lerp(input, i * fromRate / toRate); see DESIGN.md for the vDSP_vgenp divergence. */
AW_API int32_t aw_resample_output_count(int32_t count, double from_rate, double to_rate);
AW_API aw_status aw_resample(const float *input, int32_t count, double from_rate, double to_rate, float *output,
                             int32_t output_capacity, int32_t *output_count);
/* The LITERAL call the reference makes (Resampler.swift:56-65): vDSP_vramp(0, stride) into a control vector, then
 * vDSP_vgenp(A = input, B = control, C = output, N = outputCount, M = input.count), as Apple documents vgenp: the knots
 * (B[m], A[m]) define a piecewise-linear function that is evaluated at the INTEGERS n = 0..N-1 (C[n] = A[0] for n <= B[0],
 * A[M-1] past the last knot).  With B[m] = m * stride that is lerp(input, n / stride) — the inverse of the intended ratio:
 * 48 -> 96 kHz yields input[2n] then holds the last sample, 48 -> 44.1 kHz stretches the response instead of compressing it.
 * Same output length as aw_resample.  Optional; nothing selects it unless asked (aw_context_set_resampler). */
AW_API aw_status aw_resample_vgenp(const float *input, int32_t count, double from_rate, double to_rate, float *output,
                                   int32_t output_capacity, int32_t *output_count);
/* Which of the two aw_preset_activate uses when it resamples an HRIR: 0 = intended interpolation (default), 1 = literal vgenp. */
AW_API aw_status aw_context_set_resampler(aw_context *ctx, int32_t literal_vgenp);

/* HRIRManager.activatePreset body (HRIRManager.swift:347-446): load WAV -> choose map (7 tracks:
 * hesuvi7, else hesuvi14; or `custom_map`) -> resolve -> resample HRIR to target rate when it
 * differs by > 0.01 Hz -> build the spatializer.  *hrir_out (optional) receives the HRIR handle
 * the spatializer was built from (caller destroys both). */
AW_API aw_status aw_preset_activate(aw_context *ctx, const char *wav_path, double target_sample_rate,
                                    const aw_layout *input_layout, const aw_channel_map *custom_map,
                                    int32_t n_streams, aw_spatializer **spatializer_out, aw_hrir **hrir_out);

/* Seeded synthetic input on the device: U(-0.5,0.5) counter RNG,
 * value(stream, i) as oracle/airwave_oracle.h:orc_synth_value.  dst: [n_streams][frames][n_channels]. */
AW_API aw_status aw_synth_fill(aw_context *ctx, float *dst_device, int32_t n_streams, int64_t frames,
                               int32_t n_channels, uint64_t seed, uint64_t first_stream);

/* ==== Parametric EQ (the effect that follows the spatializer) ======================
 * Mirrors BiquadCoefficientBuilder, EqualizerAPOParser, ParametricEqualizerState and
 * ParametricEqualizerProcessor for a batch of independent stereo streams held in HBM
 * ([stream][frames][2] interleaved L,R float32 — the spatializer's output layout).  Arithmetic is
 * Float64 like the reference's; the time axis is evaluated chunk-parallel (device/eq_cascade.hpp), so
 * results agree with the sequential recurrence to Float64 rounding (<= 1 ulp of the Float32 output). */

/* BiquadCoefficientBuilder.make  BiquadCoefficientBuilder.swift:29-107.  type: 0 peaking, 1 lowShelf,
 * 2 highShelf.  coefficients_out = {b0, b1, b2, a1, a2}.  On AW_ERR_EQ_INVALID_FILTER *error_kind is the
 * BiquadCoefficientError: 1 invalidSampleRate, 2 invalidFrequency, 3 invalidQ, 4 nonFiniteInput,
 * 5 nonFiniteCoefficients (error_kind may be NULL). */
AW_API aw_status aw_biquad_make(int32_t type, double gain_db, double frequency_hz, double q, double sample_rate,
                                double coefficients_out[5], int32_t *error_kind);

/* EqualizerDefinition / EqualizerFilter  EqualizerPreset.swift:9-27 (host object). */
typedef struct aw_eq_definition aw_eq_definition;
AW_API aw_status aw_eq_definition_create(double preamp_db, aw_eq_definition **out);
AW_API aw_status aw_eq_definition_add_filter(aw_eq_definition *d, int32_t is_enabled, int32_t type, double frequency_hz,
                                             double gain_db, double q);
/* EqualizerFilter.sourceLine / sourceNumber of filter `index` (defaults: index + 1, none = -1). */
AW_API aw_status aw_eq_definition_set_source(aw_eq_definition *d, int32_t index, int32_t source_line, int64_t source_number);
AW_API void aw_eq_definition_destroy(aw_eq_definition *d);
AW_API double aw_eq_definition_preamp_db(const aw_eq_definition *d);
AW_API int32_t aw_eq_definition_filter_count(const aw_eq_definition *d);
/* source_number_out: -1 when the directive carried no number (nil).  Any out pointer may be NULL. */
AW_API aw_status aw_eq_definition_filter(const aw_eq_definition *d, int32_t index, int32_t *source_line, int64_t *source_number,
                                         int32_t *is_enabled, int32_t *type, double *frequency_hz, double *gain_db, double *q);
/* EqualizerAPOParser.parse(data:filename:)  EqualizerAPOParser.swift:36-151.  On AW_ERR_EQ_PARSE the
 * issues are written to issues_out as "line N: reason" / "reason" joined by "; " (the text of
 * EqualizerParseError.errorDescription without the filename prefix), truncated to issues_capacity. */
AW_API aw_status aw_eq_parse(const void *data, size_t size, aw_eq_definition **out, char *issues_out, size_t issues_capacity);

/* ParametricEqualizerState  ParametricEqualizerProcessor.swift:16-98, prepared by
 * ParametricEqualizerProcessor.prepare(definition:sampleRate:) :168-212, for n_streams streams.
 * definition may be NULL (unity).  On AW_ERR_EQ_INVALID_FILTER aw_last_error_message() names the filter. */
typedef struct aw_eq_state aw_eq_state;
AW_API aw_status aw_eq_state_create(aw_context *ctx, const aw_eq_definition *definition, double sample_rate,
                                    int32_t n_streams, aw_eq_state **out);
AW_API void aw_eq_state_destroy(aw_eq_state *s);
AW_API aw_status aw_eq_state_reset(aw_eq_state *s);                                     /* reset() :49-56 */
/* process(...) :58-91 on device buffers, in place allowed; asynchronous on the context stream. */
AW_API aw_status aw_eq_state_process(aw_eq_state *s, const float *in_device, float *out_device, int64_t frames);
AW_API int32_t aw_eq_state_filter_count(const aw_eq_state *s);
AW_API double aw_eq_state_preamp_linear(const aw_eq_state *s);

/* The equalizer FOLDED INTO THE HRIR (batch hosts; round 6).  In the reference's graph the equalizer follows the spatializer
 * (AudioEffectGraph.swift:195-211: spatial effect, then equalizer) and, between two setTarget calls, is a linear time-invariant filter
 * (preamp x biquad cascade, ParametricEqualizerProcessor.swift:58-91), so EQ(x * h) = x * (h * g) with g its impulse response.  This
 * entry convolves every HRIR track with g once, on the host, in Float64 with the reference's own recurrence; a spatializer created
 * from the folded tracks then applies both effects in its one pass over the audio — instead of ParametricEqualizerState.process
 * re-reading and re-writing every stereo frame with ten sequential Float64 sections.  g is cut after response_taps = the smallest L
 * with sum_{n >= L} |g[n]| <= tail_tolerance x max |g|; the result differs from EQ-after-spatializer by at most *tail_bound x the
 * spatializer output's peak (tail_bound <= tail_tolerance; 1e-7 is two orders below the 1e-5 parity tolerance).
 *   tracks: [n_tracks][taps] at sample_rate (resampled like HRIRManager.swift:389-403 does, if the device rate differs)
 *   out_tracks: NULL = only *out_taps is computed (= taps + response_taps - 1); else [n_tracks][*out_taps]
 * AW_ERR_EQ_NOT_FOLDABLE: the response does not decay to the tolerance within max_taps - taps + 1 frames (a narrow band at a few Hz):
 * the host keeps the cascade (aw_eq_state_process / aw_eq_process after the spatializer).  Validation and its errors as
 * aw_eq_state_create.  definition NULL = unity (the tracks come back unchanged).  Host only: no device work. */
AW_API aw_status aw_eq_fold_hrir(const aw_eq_definition *definition_or_null, double sample_rate, const float *tracks, int32_t n_tracks,
                                 int32_t taps, double tail_tolerance, int32_t max_taps, float *out_tracks, int32_t *out_taps,
                                 int32_t *response_taps, double *tail_bound);

/* ParametricEqualizerProcessor  :116-408: starts at unity; aw_eq_set_target prepares and publishes a
 * target that the next process call crossfades to over max(1, round(0.020 * sample_rate)) frames
 * (:155); newest target wins while a fade runs (:311-333); a finished fade parks the old state in a
 * one-slot retirement box that aw_eq_drain_retired empties, and a full box holds the next fade back
 * (:373-406).  Threads, as in the reference: aw_eq_set_target / aw_eq_reset / aw_eq_drain_retired may be called
 * from a control thread while ONE other thread is inside aw_eq_process*, which only tries the publication, reset and
 * retirement locks (:322,:342,:380,:393) and, when one is contended, keeps its prior target / applies the reset on a
 * later call / defers the retirement.  max_frames_per_callback: 0 = unlimited (batch); otherwise the reference's guard
 * 1..4096 (:148-150, AW_ERR_EQ_TOO_MANY_FILTERS as there) and process() rejects longer calls. */
typedef struct aw_eq aw_eq;
AW_API aw_status aw_eq_create(aw_context *ctx, double sample_rate, int32_t n_streams, int32_t max_frames_per_callback,
                              aw_eq **out);
AW_API void aw_eq_destroy(aw_eq *eq);
AW_API aw_status aw_eq_set_target(aw_eq *eq, const aw_eq_definition *definition_or_null);   /* setTarget :226-228 */
AW_API aw_status aw_eq_reset(aw_eq *eq);                                                     /* reset()   :230-234 */
AW_API aw_status aw_eq_drain_retired(aw_eq *eq);                                             /* drainRetiredStates :237-241 */
AW_API aw_status aw_eq_process(aw_eq *eq, const float *in_device, float *out_device, int64_t frames);
/* The StereoAudioProcessing surface (AudioPipeline.swift:3-11): HOST planar buffers, one stream,
 * in_right may be NULL (left is duplicated, :68); synchronous. */
AW_API aw_status aw_eq_process_planar(aw_eq *eq, const float *in_left, const float *in_right, float *out_left,
                                      float *out_right, int32_t frames);
/* withPublicationLockForTesting :229-233: hold != 0 takes the publication lock, 0 releases it (same thread). */
AW_API aw_status aw_eq_debug_hold_publication_lock(aw_eq *eq, int32_t hold);
AW_API int32_t aw_eq_transition_length(const aw_eq *eq);
AW_API int32_t aw_eq_is_transitioning(const aw_eq *eq);

AW_API const char *aw_version(void);

#ifdef __cplusplus
}
#endif
#endif
