//  HIPEqualizerEffect.swift
//  Drop-in `AudioEqualizerEffect` (Airwave/AudioEffectGraph.swift:51-54) backed by the MI355X EQ
//  (aw_eq_* in airwave_hip.h).  Same control/render split as Airwave/EqualizerRuntimeEffect.swift:
//  `prepare` / `setTarget` run on the control thread and may allocate, `process` has the
//  `StereoAudioProcessing` signature and passes audio through until a processor exists.
//  Not compiled in this repository's image (no swiftc); see INTEGRATION.md.

import Foundation
import CAirwaveHIP

public struct HIPEqualizerFilter {
    public var isEnabled: Bool
    public var type: Int32          // 0 peaking, 1 lowShelf, 2 highShelf (EqualizerFilterType)
    public var frequencyHz: Double
    public var gainDB: Double
    public var q: Double
}

public enum HIPEqualizerError: Error {
    case invalidSampleRate
    case invalidFilter(reason: String)
    case unavailable(String)
    case parse(String)
}

/// One sample-rate-specific processor (`aw_eq`).  Destroyed when the LAST reference goes away — the control side's or the
/// render thread's snapshot — and only ever on the control thread (see `retire`).  Retains the context `aw_eq_destroy` uses.
final class EqualizerBox {
    let handle: OpaquePointer
    let sampleRate: Double
    let context: HIPContext
    init(_ h: OpaquePointer, sampleRate: Double, context: HIPContext) { handle = h; self.sampleRate = sampleRate; self.context = context }
    deinit { aw_eq_destroy(handle) }
}

public final class HIPEqualizerEffect /* : AudioEqualizerEffect */ {
    private let context: HIPContext
    // The publication scheme of Airwave/EqualizerRuntimeEffect.swift:6-8,57-61: the control thread publishes a processor under
    // `processorLock`, the render thread takes one try-lock snapshot per callback into `audioThreadProcessor` and otherwise keeps
    // the one it has — it can never observe a processor that `prepare` is tearing down.
    private let processorLock = TryLock<EqualizerBox?>(initialState: nil)
    private var controlProcessor: EqualizerBox?                  // control thread only
    private var audioThreadProcessor: EqualizerBox?              // render thread only
    // `aw_eq_destroy` frees device memory and synchronises a stream: never on the render thread.  A processor the render
    // thread lets go of is parked (try-lock) and destroyed by the next control-side call, as in HIPSpatialEffect.
    private let kRetireCapacity = 64       // boxes in flight between a publish and the next drain; the render thread never grows these arrays
    private let retiredLock = TryLock<[EqualizerBox]>(initialState: [])
    private var awaitingRetirement: [EqualizerBox] = []          // render thread only; capacity reserved in init

    public init(context: HIPContext) {
        self.context = context
        awaitingRetirement.reserveCapacity(kRetireCapacity)
        retiredLock.withLock { $0.reserveCapacity(kRetireCapacity) }
    }

    /// EqualizerAPOParser.parse(data:filename:) -> definition handle (caller destroys with aw_eq_definition_destroy).
    public static func parse(data: Data) throws -> OpaquePointer {
        var def: OpaquePointer?
        var issues = [CChar](repeating: 0, count: 4096)
        let st = data.withUnsafeBytes { aw_eq_parse($0.baseAddress, data.count, &def, &issues, issues.count) }
        guard st == AW_OK, let d = def else { throw HIPEqualizerError.parse(String(cString: issues)) }
        return d
    }

    private func makeDefinition(preampDB: Double, filters: [HIPEqualizerFilter]) -> OpaquePointer? {
        var def: OpaquePointer?
        guard aw_eq_definition_create(preampDB, &def) == AW_OK, let d = def else { return nil }
        for f in filters {
            _ = aw_eq_definition_add_filter(d, f.isEnabled ? 1 : 0, f.type, f.frequencyHz, f.gainDB, f.q)
        }
        return d
    }

    /// AudioEqualizerEffect.prepare(definition:sampleRate:)   EqualizerRuntimeEffect.swift:10-34
    public func prepare(preampDB: Double?, filters: [HIPEqualizerFilter], sampleRate: Double) throws {
        guard sampleRate.isFinite, sampleRate > 0 else { throw HIPEqualizerError.invalidSampleRate }
        let box: EqualizerBox
        if let current = controlProcessor, current.sampleRate == sampleRate {
            box = current
        } else {
            var p: OpaquePointer?
            guard aw_eq_create(context.handle, sampleRate, 1, 4096, &p) == AW_OK, let created = p else {
                throw HIPEqualizerError.unavailable(String(cString: aw_last_error_message()))
            }
            box = EqualizerBox(created, sampleRate: sampleRate, context: context)
            controlProcessor = box
            // Publish.  The previous processor is NOT destroyed here: the render thread may be inside `process` with it; it is
            // released by the render thread's next successful snapshot and destroyed by a later control-side call.
            processorLock.withLock { $0 = box }
        }
        drainRetiredProcessors()
        try publish(box, preampDB: preampDB, filters: filters)
    }

    /// AudioEqualizerEffect.setTarget(definition:)   EqualizerRuntimeEffect.swift:36-48
    public func setTarget(preampDB: Double?, filters: [HIPEqualizerFilter]) throws {
        guard let box = controlProcessor else {
            throw HIPEqualizerError.unavailable("Equalizer has not been prepared for an output.")
        }
        drainRetiredProcessors()
        try publish(box, preampDB: preampDB, filters: filters)
    }

    /// `aw_eq_set_target` / `aw_eq_drain_retired` take the processor's own publication locks, `aw_eq_process_planar` only tries
    /// them (ParametricEqualizerProcessor.swift:322,342,380,393): safe against a concurrent `process` on the same handle.
    private func publish(_ box: EqualizerBox, preampDB: Double?, filters: [HIPEqualizerFilter]) throws {
        let def = preampDB.flatMap { makeDefinition(preampDB: $0, filters: filters) }   // nil = unity
        defer { if let d = def { aw_eq_definition_destroy(d) } }
        let st = aw_eq_set_target(box.handle, def)
        if st != AW_OK {                                           // :27-33: fall back to unity, then report
            let reason = String(cString: aw_last_error_message())
            _ = aw_eq_set_target(box.handle, nil)
            _ = aw_eq_drain_retired(box.handle)
            throw st == AW_ERR_EQ_INVALID_SAMPLE_RATE ? HIPEqualizerError.invalidSampleRate
                                                      : HIPEqualizerError.invalidFilter(reason: reason)
        }
        _ = aw_eq_drain_retired(box.handle)
    }

    /// Control thread: destroys the processors the render thread has let go of (their deinit runs here, not in `process`).
    /// Only boxes nobody else references are destroyed: the render thread may still hold a handed-over box in a local for the rest of its
    /// callback (after it has released the try-lock), and the last release must not happen there — `EqualizerBox.deinit` frees device memory
    /// and synchronises the stream.  Such a box waits for the next drain.
    public func drainRetiredProcessors() {
        var taken = retiredLock.withLock { list -> [EqualizerBox] in let d = list; list.removeAll(keepingCapacity: true); return d }
        var stillShared: [EqualizerBox] = []
        while var box = taken.popLast() {
            if !isKnownUniquelyReferenced(&box) { stillShared.append(box) }
            // else: `box` is the last reference and ends here, on the control thread
        }
        if !stillShared.isEmpty { retiredLock.withLock { $0.append(contentsOf: stillShared) } }
    }

    /// StereoAudioProcessing.process   EqualizerRuntimeEffect.swift:50-78
    public func process(inputLeft: UnsafePointer<Float>, inputRight: UnsafePointer<Float>?,
                        outputLeft: UnsafeMutablePointer<Float>, outputRight: UnsafeMutablePointer<Float>,
                        frameCount: Int) {
        var processor = audioThreadProcessor
        enum Read { case available(EqualizerBox?) }
        if let read = processorLock.withLockIfAvailable({ Read.available($0) }) {      // one non-blocking snapshot attempt
            if case .available(let published) = read {
                if let old = audioThreadProcessor, old !== published { retire(old) }
                processor = published
                audioThreadProcessor = published
            }
        }
        if !awaitingRetirement.isEmpty { flushAwaitingRetirement() }
        guard let p = processor,
              aw_eq_process_planar(p.handle, inputLeft, inputRight, outputLeft, outputRight, Int32(frameCount)) == AW_OK else {
            memcpy(outputLeft, inputLeft, frameCount * MemoryLayout<Float>.size)
            memcpy(outputRight, inputRight ?? inputLeft, frameCount * MemoryLayout<Float>.size)
            return
        }
    }

    /// Render thread: hand a processor over for destruction without ever blocking or freeing here.
    private func retire(_ box: EqualizerBox) {
        awaitingRetirement.append(box)
        flushAwaitingRetirement()
    }

    private func flushAwaitingRetirement() {
        // Never beyond the capacity reserved on the control side: an append that grows the array would allocate on the render thread.
        // What does not fit stays in the render thread's own list for a later callback (after the control side has drained).
        let moved: Int? = retiredLock.withLockIfAvailable { list in
            var n = 0
            while n < awaitingRetirement.count && list.count < list.capacity { list.append(awaitingRetirement[n]); n += 1 }
            return n
        }
        if let n = moved, n > 0 { awaitingRetirement.removeFirst(n) }     // keeps the storage (no reallocation)
    }
}
