// eq_kernels.hpp — host-callable launchers for the parametric EQ kernels (eq_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "eq_cascade.hpp"

namespace awk {

// Chunk-parallel cascade over p.frames (multiple of kEqChunk) frames of every stream: one workgroup
// of kEqThreads per stream.
hipError_t prepare_eq_kernels();     // dynamic-LDS attribute of the cascade kernels (more than 64 KB); once per context
hipError_t launch_eq_cascade(const EqParams &p, int n_streams, hipStream_t stream);
// The sequential recurrence, one thread per (stream, ear): short calls and tails.
hipError_t launch_eq_sequential(const EqParams &p, int n_streams, hipStream_t stream);
// Crossfade of two rendered segments (ParametricEqualizerProcessor.swift:296-304):
// out[s][f][e] = Float(Double(old) * (1 - g) + Double(new) * g),  g = (t_frame + f + 1) / length.
// old_seg/new_seg: [stream][seg_frames][2]; out: [stream][out_stride][2].
hipError_t launch_eq_blend(const float *old_seg, const float *new_seg, float *out, int n_streams, long long seg_frames,
                           long long out_stride, long long t_frame, long long length, hipStream_t stream);
// dst[s][f][:] = src[s][f][:] for f < frames with distinct strides (used by passthrough copies)
hipError_t launch_eq_copy(const float *src, long long src_stride, float *dst, long long dst_stride, int n_streams,
                          long long frames, hipStream_t stream);

}  // namespace awk
