//  BatchSpatializer.swift
//  Offline/batch entry that the macOS product does not have: N independent streams of interleaved
//  multichannel PCM resident on the GPU -> interleaved stereo, one call.

import Foundation
import CAirwaveHIP

public final class BatchSpatializer {
    private let handle: OpaquePointer
    public let streams: Int
    public let channels: Int

    /// The equalizer of the batch, if one was folded into the HRIR: samples of its impulse response that were kept and the bound on what
    /// the cut can change relative to the spatializer output's peak (aw_eq_fold_hrir).  nil: no equalizer, or not folded.
    public private(set) var foldedEqualizer: (responseTaps: Int, tailBound: Double)?

    /// tracks: planar HRIR ([track][tap]); leftTrack/rightTrack per input channel, -1 = speaker without a mapping
    /// (skipped like HRIRManager.swift:370-372).  Throws the reference's error categories as NSError codes.
    /// equalizer: the definition AudioEffectGraph would run AFTER the spatial effect (AudioEffectGraph.swift:195-211).  A batch host knows
    /// it up front, so it is folded into the HRIR here — EQ(x * h) = x * (h * g), one pass over the audio instead of two — when its impulse
    /// response decays within `equalizerMaxTaps`; otherwise (AW_ERR_EQ_NOT_FOLDABLE) the initialiser throws and the host runs
    /// `HIPEqualizerEffect` / aw_eq_process on the output as the reference's graph does.
    public init(context: HIPContext, tracks: [[Float]], sampleRate: Double, leftTrack: [Int32], rightTrack: [Int32], streams: Int,
                equalizer: HIPEqualizerDefinition? = nil, equalizerTailTolerance: Double = 1e-7, equalizerMaxTaps: Int32 = 65536) throws {
        precondition(leftTrack.count == rightTrack.count)
        var taps = tracks.first?.count ?? 0
        var flat = tracks.flatMap { $0 }
        if let definition = equalizer {
            guard let def = HIPEqualizerEffect.makeDefinitionHandle(definition) else { throw BatchSpatializer.error(AW_ERR_OUT_OF_MEMORY) }
            defer { aw_eq_definition_destroy(def) }
            var outTaps: Int32 = 0, response: Int32 = 0, bound = 0.0
            var st = flat.withUnsafeBufferPointer {
                aw_eq_fold_hrir(def, sampleRate, $0.baseAddress, Int32(tracks.count), Int32(taps), equalizerTailTolerance, equalizerMaxTaps, nil, &outTaps, &response, &bound)
            }
            guard st == AW_OK else { throw BatchSpatializer.error(st) }
            var folded = [Float](repeating: 0, count: tracks.count * Int(outTaps))
            st = flat.withUnsafeBufferPointer { src in
                folded.withUnsafeMutableBufferPointer { dst in
                    aw_eq_fold_hrir(def, sampleRate, src.baseAddress, Int32(tracks.count), Int32(taps), equalizerTailTolerance, equalizerMaxTaps, dst.baseAddress, &outTaps, &response, &bound)
                }
            }
            guard st == AW_OK else { throw BatchSpatializer.error(st) }
            flat = folded
            taps = Int(outTaps)
            foldedEqualizer = (Int(response), bound)
        }
        var hrir: OpaquePointer?
        var st = flat.withUnsafeBufferPointer { aw_hrir_create(context.handle, $0.baseAddress, Int32(tracks.count), Int32(taps), sampleRate, &hrir) }
        guard st == AW_OK, let h = hrir else { throw BatchSpatializer.error(st) }
        defer { aw_hrir_destroy(h) }
        var sp: OpaquePointer?
        st = aw_spatializer_create(context.handle, h, Int32(leftTrack.count), leftTrack, rightTrack, Int32(streams), 0, &sp)
        guard st == AW_OK, let s = sp else { throw BatchSpatializer.error(st) }
        handle = s
        self.streams = streams
        self.channels = leftTrack.count
    }
    deinit { aw_spatializer_destroy(handle) }

    /// Device pointers: input [stream][frame][channel], output [stream][frame][2].  Asynchronous on the context stream.
    public func process(deviceInput: UnsafePointer<Float>, deviceOutput: UnsafeMutablePointer<Float>, frames: Int64) throws {
        let st = aw_spatializer_process(handle, deviceInput, deviceOutput, frames)
        guard st == AW_OK else { throw BatchSpatializer.error(st) }
    }

    /// Sizes every internal device buffer for calls of up to `maxFrames` frames; `process` never allocates afterwards.
    public func reserve(maxFrames: Int64) throws {
        let st = aw_spatializer_reserve(handle, maxFrames)
        guard st == AW_OK else { throw BatchSpatializer.error(st) }
    }

    /// Host buffers: the batch crosses PCIe in chunks of streams, H2D of the next chunk and D2H of the previous one under the
    /// kernels of the current one.  Page-locked buffers (`aw_host_alloc_pinned`) move by DMA directly.  Synchronous.
    public func process(hostInput: UnsafePointer<Float>, hostOutput: UnsafeMutablePointer<Float>, frames: Int64) throws {
        let st = aw_spatializer_process_host(handle, hostInput, hostOutput, frames)
        guard st == AW_OK else { throw BatchSpatializer.error(st) }
    }

    /// `reserve` plus the host entry's device-side staging: `process(hostInput:…)` never allocates afterwards either.
    public func reserveHost(maxFrames: Int64) throws {
        let st = aw_spatializer_reserve_host(handle, maxFrames)
        guard st == AW_OK else { throw BatchSpatializer.error(st) }
    }

    public func reset() { _ = aw_spatializer_reset(handle) }

    static func error(_ st: aw_status) -> NSError {
        NSError(domain: "AirwaveHIP", code: Int(st), userInfo: [NSLocalizedDescriptionKey: String(cString: aw_last_error_message())])
    }
}
