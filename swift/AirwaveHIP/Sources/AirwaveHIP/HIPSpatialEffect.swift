//  HIPSpatialEffect.swift
//  Drop-in `AudioSpatialEffect` (Airwave/AudioEffectGraph.swift:47-49) backed by the MI355X engine.
//
//  The class has the exact `StereoAudioProcessing.process` signature (Airwave/AudioPipeline.swift:3-11):
//  planar Float32 buffers owned by the caller, `inputRight == nil` duplicates left, never throws.
//  Activation mirrors `HRIRManager.activatePreset` (Airwave/HRIRManager.swift:316-449): it may block
//  (call it off the render thread, as the reference does) and reports `HRIRActivationResult`.

import Foundation
import CAirwaveHIP

public enum HIPActivationResult: Equatable {
    case success
    case failure(String)
}

public final class HIPContext {
    let handle: OpaquePointer
    public init?(device: Int32 = 0) {
        var h: OpaquePointer?
        guard aw_context_create(device, &h) == AW_OK, let ctx = h else { return nil }
        handle = ctx
    }
    deinit { aw_context_destroy(handle) }
}

public final class HIPSpatialEffect /* : AudioSpatialEffect */ {
    private let context: HIPContext
    private var spatializer: OpaquePointer?   // published like HRIRManager.rendererState: replaced, never mutated

    public init(context: HIPContext) { self.context = context }
    deinit { if let s = spatializer { aw_spatializer_destroy(s) } }

    /// AudioSpatialEffect.isReady
    public var isReady: Bool { spatializer != nil }

    /// HRIRManager.activatePreset(_:targetSampleRate:inputLayout:)  — stereo layout, like the shipped product
    /// (DeviceProfileRuntimeCoordinator.swift:104-108).
    @discardableResult
    public func activatePreset(fileURL: URL, targetSampleRate: Double, channelCount: Int32 = 2) -> HIPActivationResult {
        var layout: OpaquePointer?
        guard aw_layout_detect(channelCount, &layout) == AW_OK else { return .failure(lastError()) }
        defer { aw_layout_destroy(layout) }
        var created: OpaquePointer?
        let status = fileURL.path.withCString {
            aw_preset_activate(context.handle, $0, targetSampleRate, layout, nil, 1, &created, nil)
        }
        guard status == AW_OK, let sp = created else {
            return .failure("Failed to activate preset: \(lastError())")   // HRIRManager.swift:441
        }
        let old = spatializer
        spatializer = sp
        if let o = old { aw_spatializer_destroy(o) }
        return .success
    }

    public func deactivatePreset() {
        if let s = spatializer { aw_spatializer_destroy(s) }
        spatializer = nil
    }

    /// StereoAudioProcessing.process(inputLeft:inputRight:outputLeft:outputRight:frameCount:)
    public func process(
        inputLeft: UnsafePointer<Float>,
        inputRight: UnsafePointer<Float>?,
        outputLeft: UnsafeMutablePointer<Float>,
        outputRight: UnsafeMutablePointer<Float>,
        frameCount: Int
    ) {
        guard frameCount > 0 else { return }
        guard let sp = spatializer else {
            // passthrough, HRIRManager.swift:550-559
            memcpy(outputLeft, inputLeft, frameCount * MemoryLayout<Float>.size)
            memcpy(outputRight, inputRight ?? inputLeft, frameCount * MemoryLayout<Float>.size)
            return
        }
        _ = aw_spatializer_process_planar(sp, inputLeft, inputRight, outputLeft, outputRight, Int32(frameCount))
    }

    /// HRIRManager.resetConvolutionState()
    public func resetConvolutionState() { if let s = spatializer { _ = aw_spatializer_reset(s) } }

    private func lastError() -> String { String(cString: aw_last_error_message()) }
}
