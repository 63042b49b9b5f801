//  HIPEqualizerEffect.swift
//  Drop-in `AudioEqualizerEffect` (Airwave/AudioEffectGraph.swift:51-54) backed by the MI355X EQ
//  (aw_eq_* in airwave_hip.h).  Same control/render split as Airwave/EqualizerRuntimeEffect.swift:
//  `prepare(definition:sampleRate:)` / `setTarget(definition:)` — the protocol's exact shape — run on the control
//  thread and may allocate, `process` has the `StereoAudioProcessing` signature and passes audio through until a
//  processor exists.  The definition types mirror Airwave/EqualizerPreset.swift:9-27 field for field (the app's own
//  are internal to its module; INTEGRATION.md shows the two-line conversion and the conformance the app adds).
//  Not compiled in this repository's image (no swiftc); see INTEGRATION.md.

import Foundation
import CAirwaveHIP

/// EqualizerFilterType   EqualizerPreset.swift:3-7
public enum HIPEqualizerFilterType: Int32, Equatable {
    case peaking = 0, lowShelf = 1, highShelf = 2
}

/// EqualizerFilter   EqualizerPreset.swift:9-17 — `sourceLine` / `sourceNumber` travel with the filter: a rejected filter is
/// reported by the line of the Equalizer APO file it came from (EqualizerRuntimeEffect.swift:85-89).
public struct HIPEqualizerFilter: Equatable {
    public let sourceLine: Int
    public let sourceNumber: Int?
    public let isEnabled: Bool
    public let type: HIPEqualizerFilterType
    public let frequencyHz: Double
    public let gainDB: Double
    public let q: Double
    public init(sourceLine: Int, sourceNumber: Int?, isEnabled: Bool, type: HIPEqualizerFilterType, frequencyHz: Double, gainDB: Double, q: Double) {
        self.sourceLine = sourceLine; self.sourceNumber = sourceNumber; self.isEnabled = isEnabled; self.type = type
        self.frequencyHz = frequencyHz; self.gainDB = gainDB; self.q = q
    }
}

/// EqualizerDefinition   EqualizerPreset.swift:19-27
public struct HIPEqualizerDefinition: Equatable {
    public let preampDB: Double
    public let filters: [HIPEqualizerFilter]
    public init(preampDB: Double = 0, filters: [HIPEqualizerFilter] = []) { self.preampDB = preampDB; self.filters = filters }
}

/// EqualizerAudioEffectError   AudioEffectGraph.swift:27-46 (+ .parse for the static parser below)
public enum HIPEqualizerError: Error, Equatable {
    case invalidFilter(line: Int?, reason: String)
    case invalidSampleRate
    case unavailable(String)
    case parse(String)

    public var filterLine: Int? {
        if case .invalidFilter(let line, _) = self { return line }
        return nil
    }
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

public final class HIPEqualizerEffect {      // the app declares `extension HIPEqualizerEffectAdapter: AudioEqualizerEffect` (INTEGRATION.md 1b)
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

    /// EqualizerAPOParser.parse(data:filename:)   EqualizerAPOParser.swift:36-151 — the library's parser, read back into the value type
    /// (source lines and numbers included: aw_eq_definition_filter).
    public static func parse(data: Data) throws -> HIPEqualizerDefinition {
        var def: OpaquePointer?
        var issues = [CChar](repeating: 0, count: 4096)
        let st = data.withUnsafeBytes { aw_eq_parse($0.baseAddress, data.count, &def, &issues, issues.count) }
        guard st == AW_OK, let d = def else { throw HIPEqualizerError.parse(String(cString: issues)) }
        defer { aw_eq_definition_destroy(d) }
        var filters: [HIPEqualizerFilter] = []
        for i in 0..<aw_eq_definition_filter_count(d) {
            var line: Int32 = 0, number: Int64 = -1, enabled: Int32 = 0, type: Int32 = 0
            var f = 0.0, g = 0.0, q = 0.0
            guard aw_eq_definition_filter(d, i, &line, &number, &enabled, &type, &f, &g, &q) == AW_OK else { continue }
            filters.append(HIPEqualizerFilter(sourceLine: Int(line), sourceNumber: number < 0 ? nil : Int(number), isEnabled: enabled != 0,
                                              type: HIPEqualizerFilterType(rawValue: type) ?? .peaking, frequencyHz: f, gainDB: g, q: q))
        }
        return HIPEqualizerDefinition(preampDB: aw_eq_definition_preamp_db(d), filters: filters)
    }

    /// The definition as the C ABI's host object, `sourceLine` / `sourceNumber` included (aw_eq_definition_set_source): the library
    /// reports a rejected filter by them (aw_last_eq_filter_error).  nil definition = nil handle = unity.
    private func makeDefinition(_ definition: HIPEqualizerDefinition) -> OpaquePointer? { Self.makeDefinitionHandle(definition) }
    static func makeDefinitionHandle(_ definition: HIPEqualizerDefinition) -> OpaquePointer? {
        var def: OpaquePointer?
        guard aw_eq_definition_create(definition.preampDB, &def) == AW_OK, let d = def else { return nil }
        for (i, f) in definition.filters.enumerated() {
            _ = aw_eq_definition_add_filter(d, f.isEnabled ? 1 : 0, f.type.rawValue, f.frequencyHz, f.gainDB, f.q)
            _ = aw_eq_definition_set_source(d, Int32(i), Int32(f.sourceLine), Int64(f.sourceNumber ?? -1))
        }
        return d
    }

    /// AudioEqualizerEffect.prepare(definition:sampleRate:)   AudioEffectGraph.swift:52, EqualizerRuntimeEffect.swift:10-34
    public func prepare(definition: HIPEqualizerDefinition?, sampleRate: Double) throws {
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
        try publish(box, definition: definition)
    }

    /// AudioEqualizerEffect.setTarget(definition:)   AudioEffectGraph.swift:53, EqualizerRuntimeEffect.swift:36-48
    public func setTarget(definition: HIPEqualizerDefinition?) throws {
        guard let box = controlProcessor else {
            throw HIPEqualizerError.unavailable("Equalizer has not been prepared for an output.")
        }
        drainRetiredProcessors()
        try publish(box, definition: definition)
    }

    /// `aw_eq_set_target` / `aw_eq_drain_retired` take the processor's own publication locks, `aw_eq_process_planar` only tries
    /// them (ParametricEqualizerProcessor.swift:322,342,380,393): safe against a concurrent `process` on the same handle.
    private func publish(_ box: EqualizerBox, definition: HIPEqualizerDefinition?) throws {
        let def = definition.flatMap { makeDefinition($0) }        // nil = unity
        defer { if let d = def { aw_eq_definition_destroy(d) } }
        let st = aw_eq_set_target(box.handle, def)
        if st != AW_OK {                                           // :27-33: fall back to unity, then report
            let error = Self.map(st, definition: definition)       // (reads the thread's error state before the calls below replace it)
            _ = aw_eq_set_target(box.handle, nil)
            _ = aw_eq_drain_retired(box.handle)
            throw error
        }
        _ = aw_eq_drain_retired(box.handle)
    }

    /// EqualizerRuntimeEffect.map   EqualizerRuntimeEffect.swift:80-100: ParametricEqualizerPreparationError -> EqualizerAudioEffectError.
    /// `.invalidFilter(index:error:)` names the filter by its index among the ENABLED filters; the line is that filter's `sourceLine`
    /// (the library looked it up the same way and reports it with the index: aw_last_eq_filter_error), the reason the
    /// BiquadCoefficientError's description without the "Filter N is invalid:" prefix of the preparation error.
    static func map(_ status: aw_status, definition: HIPEqualizerDefinition?) -> HIPEqualizerError {
        switch status {
        case AW_ERR_EQ_INVALID_FILTER:
            var index: Int32 = 0, kind: Int32 = 0, line: Int32 = 0
            guard aw_last_eq_filter_error(&index, &kind, &line) == 1 else {
                return .invalidFilter(line: nil, reason: String(cString: aw_last_error_message()))
            }
            let enabled = (definition?.filters ?? []).filter(\.isEnabled)
            let sourceLine: Int? = enabled.indices.contains(Int(index)) ? enabled[Int(index)].sourceLine : nil
            return .invalidFilter(line: sourceLine, reason: biquadErrorDescription(kind))
        case AW_ERR_EQ_INVALID_SAMPLE_RATE:
            return .invalidSampleRate
        case AW_ERR_EQ_NON_FINITE_PREAMP:
            return .invalidFilter(line: nil, reason: "Preamp produces a non-finite gain.")
        case AW_ERR_EQ_TOO_MANY_FILTERS:
            let count = (definition?.filters ?? []).filter(\.isEnabled).count
            return .invalidFilter(line: nil, reason: "Equalizer supports at most 64 filters; received \(count).")
        default:
            return .unavailable(String(cString: aw_last_error_message()))
        }
    }

    /// BiquadCoefficientError.errorDescription   BiquadCoefficientBuilder.swift:18-26 (kinds as aw_biquad_make reports them)
    static func biquadErrorDescription(_ kind: Int32) -> String {
        switch kind {
        case 1: return "Sample rate must be finite and positive."
        case 2: return "Frequency must be finite, positive, and below Nyquist."
        case 3: return "Q must be finite and positive."
        case 4: return "Filter parameters must be finite."
        default: return "Filter coefficients must be finite."
        }
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
