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

/// One activated renderer network.  Immutable once published; the C handle is destroyed when the LAST reference goes
/// away — the control thread's or the render thread's snapshot, whichever drops it later — exactly the lifetime rule of
/// `HRIRManager.RendererState` (Airwave/HRIRManager.swift:118-131), which ARC keeps alive while `audioThreadState` holds it.
/// The box RETAINS the context: `aw_spatializer_destroy` dereferences it (device, stream), and boxes can outlive the
/// effect's own `context` property during teardown (stored properties are released in declaration order).
final class SpatializerBox {
    let handle: OpaquePointer
    let context: HIPContext
    init(_ h: OpaquePointer, context: HIPContext) { handle = h; self.context = context }
    deinit { aw_spatializer_destroy(handle) }
}

public final class HIPSpatialEffect /* : AudioSpatialEffect */ {
    private let context: HIPContext
    // Writers publish immutable state under one lock; the render thread makes one non-blocking snapshot attempt per
    // callback and otherwise keeps the state it already has (HRIRManager.swift:133-147, 539-548).
    private let stateLock = TryLock<SpatializerBox?>(initialState: nil)      // TryLock.swift: pthread try-lock, Linux and Darwin
    private var audioThreadState: SpatializerBox?                            // touched by the render thread only
    // Destroying a handle frees device memory and synchronises a HIP stream: never on the render thread.  A state the
    // render thread lets go of is parked here (try-lock; kept one more callback under contention) and destroyed by the
    // next control-side call — the retirement scheme of ParametricEqualizerProcessor (ParametricEqualizerProcessor.swift:380-406).
    private let kRetireCapacity = 64       // boxes in flight between a publish and the next drain; the render thread never grows these arrays
    private let retiredLock = TryLock<[SpatializerBox]>(initialState: [])
    private var awaitingRetirement: [SpatializerBox] = []                    // render thread only; capacity reserved in init

    public init(context: HIPContext) {
        self.context = context
        awaitingRetirement.reserveCapacity(kRetireCapacity)       // the render thread appends without allocating
        retiredLock.withLock { $0.reserveCapacity(kRetireCapacity) }
    }

    /// AudioSpatialEffect.isReady
    public var isReady: Bool { stateLock.withLock { $0 != nil } }

    /// HRIRManager.activatePreset(_:targetSampleRate:inputLayout:)  — stereo layout, like the shipped product
    /// (DeviceProfileRuntimeCoordinator.swift:104-108).  Blocks (file I/O, table FFTs, device allocation): call it off
    /// the render thread, as the reference does on its background queue.
    @discardableResult
    public func activatePreset(fileURL: URL, targetSampleRate: Double, channelCount: Int32 = 2, maxFramesPerCallback: Int = 4096) -> HIPActivationResult {
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
        let fresh = SpatializerBox(sp, context: context)
        guard aw_spatializer_reserve(sp, Int64(maxFramesPerCallback)) == AW_OK else {      // process never allocates afterwards
            return .failure("Failed to activate preset: \(lastError())")                  // (`fresh` destroys the handle here, on the control thread)
        }
        // Publish.  The previous box is NOT destroyed here: the render thread may be inside `process` with it; it dies
        // when the render thread's next successful snapshot replaces `audioThreadState` (deferred destruction by ARC).
        stateLock.withLock { $0 = fresh }
        drainRetiredStates()
        return .success
    }

    public func deactivatePreset() {
        stateLock.withLock { $0 = nil }
        drainRetiredStates()
    }

    /// Control thread: destroys the states the render thread has let go of (their deinit runs here, not in `process`).
    /// Only boxes nobody else references are destroyed here: the render thread may still hold a handed-over box in a local for the rest
    /// of its callback, and the last release — `SpatializerBox.deinit` frees device memory and synchronises — must not happen there.
    /// Such a box waits for the next drain.
    public func drainRetiredStates() {
        var taken = retiredLock.withLock { list -> [SpatializerBox] in let d = list; list.removeAll(keepingCapacity: true); return d }
        var stillShared: [SpatializerBox] = []
        while var box = taken.popLast() {
            if !isKnownUniquelyReferenced(&box) { stillShared.append(box) }
            // else: `box` is the last reference and ends here, on the calling (control) thread
        }
        if !stillShared.isEmpty { retiredLock.withLock { $0.append(contentsOf: stillShared) } }
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
        // A writer can never stall the render thread.  A failed attempt keeps the prior immutable state.
        var state = audioThreadState
        enum StateRead { case available(SpatializerBox?) }
        if let read = stateLock.withLockIfAvailable({ StateRead.available($0) }) {
            if case .available(let published) = read {
                if let old = audioThreadState, old !== published { retire(old) }
                state = published
                audioThreadState = published
            }
        }
        if !awaitingRetirement.isEmpty { flushAwaitingRetirement() }
        guard let state else {
            // passthrough, HRIRManager.swift:550-559
            memcpy(outputLeft, inputLeft, frameCount * MemoryLayout<Float>.size)
            memcpy(outputRight, inputRight ?? inputLeft, frameCount * MemoryLayout<Float>.size)
            return
        }
        _ = aw_spatializer_process_planar(state.handle, inputLeft, inputRight, outputLeft, outputRight, Int32(frameCount))
    }

    /// Render thread: hand a state over for destruction without ever blocking or freeing here.
    private func retire(_ box: SpatializerBox) {
        awaitingRetirement.append(box)
        flushAwaitingRetirement()
    }

    private func flushAwaitingRetirement() {
        // try-lock only: under contention the states wait in the render thread's own list for a later callback
        // Never beyond the capacity reserved on the control side: an append that grows the array would allocate on the render thread.
        // What does not fit stays in the render thread's own list for a later callback (after the control side has drained).
        let moved: Int? = retiredLock.withLockIfAvailable { list in
            var n = 0
            while n < awaitingRetirement.count && list.count < list.capacity { list.append(awaitingRetirement[n]); n += 1 }
            return n
        }
        if let n = moved, n > 0 { awaitingRetirement.removeFirst(n) }     // keeps the storage (no reallocation)
    }

    /// HRIRManager.resetConvolutionState(): the reference resets through the control side's reference to the state; the
    /// engine handle is single-owner for `process`, so the host must call this with the render thread quiescent, as the
    /// reference does (device stop / start, AudioPipeline.swift).
    public func resetConvolutionState() {
        if let s = stateLock.withLock({ $0 }) { _ = aw_spatializer_reset(s.handle) }
    }

    private func lastError() -> String { String(cString: aw_last_error_message()) }
}
