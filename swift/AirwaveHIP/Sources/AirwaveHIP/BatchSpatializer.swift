//  BatchSpatializer.swift
//  Offline/batch entry that the macOS product does not have: N independent streams of interleaved
//  multichannel PCM resident on the GPU -> interleaved stereo, one call.

import Foundation
import CAirwaveHIP

public final class BatchSpatializer {
    private let handle: OpaquePointer
    public let streams: Int
    public let channels: Int

    /// tracks: planar HRIR ([track][tap]); leftTrack/rightTrack per input channel, -1 = speaker without a mapping
    /// (skipped like HRIRManager.swift:370-372).  Throws the reference's error categories as NSError codes.
    public init(context: HIPContext, tracks: [[Float]], sampleRate: Double, leftTrack: [Int32], rightTrack: [Int32], streams: Int) throws {
        precondition(leftTrack.count == rightTrack.count)
        let taps = tracks.first?.count ?? 0
        let flat = tracks.flatMap { $0 }
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
