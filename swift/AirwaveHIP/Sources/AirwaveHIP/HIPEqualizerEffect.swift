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

public final class HIPEqualizerEffect /* : AudioEqualizerEffect */ {
    private let context: HIPContext
    private var processor: OpaquePointer?
    private var sampleRate: Double = 0

    public init(context: HIPContext) { self.context = context }
    deinit { if let p = processor { aw_eq_destroy(p) } }

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
        if processor == nil || self.sampleRate != sampleRate {
            var p: OpaquePointer?
            guard aw_eq_create(context.handle, sampleRate, 1, 4096, &p) == AW_OK, let created = p else {
                throw HIPEqualizerError.unavailable(String(cString: aw_last_error_message()))
            }
            if let old = processor { aw_eq_destroy(old) }
            processor = created
            self.sampleRate = sampleRate
        }
        try publish(preampDB: preampDB, filters: filters)
    }

    /// AudioEqualizerEffect.setTarget(definition:)   EqualizerRuntimeEffect.swift:36-48
    public func setTarget(preampDB: Double?, filters: [HIPEqualizerFilter]) throws {
        guard processor != nil else {
            throw HIPEqualizerError.unavailable("Equalizer has not been prepared for an output.")
        }
        try publish(preampDB: preampDB, filters: filters)
    }

    private func publish(preampDB: Double?, filters: [HIPEqualizerFilter]) throws {
        guard let p = processor else { return }
        let def = preampDB.flatMap { makeDefinition(preampDB: $0, filters: filters) }   // nil = unity
        defer { if let d = def { aw_eq_definition_destroy(d) } }
        let st = aw_eq_set_target(p, def)
        if st != AW_OK {                                           // :27-33: fall back to unity, then report
            let reason = String(cString: aw_last_error_message())
            _ = aw_eq_set_target(p, nil)
            _ = aw_eq_drain_retired(p)
            throw st == AW_ERR_EQ_INVALID_SAMPLE_RATE ? HIPEqualizerError.invalidSampleRate
                                                      : HIPEqualizerError.invalidFilter(reason: reason)
        }
        _ = aw_eq_drain_retired(p)
    }

    /// StereoAudioProcessing.process   EqualizerRuntimeEffect.swift:50-78
    public func process(inputLeft: UnsafePointer<Float>, inputRight: UnsafePointer<Float>?,
                        outputLeft: UnsafeMutablePointer<Float>, outputRight: UnsafeMutablePointer<Float>,
                        frameCount: Int) {
        guard let p = processor,
              aw_eq_process_planar(p, inputLeft, inputRight, outputLeft, outputRight, Int32(frameCount)) == AW_OK else {
            memcpy(outputLeft, inputLeft, frameCount * MemoryLayout<Float>.size)
            memcpy(outputRight, inputRight ?? inputLeft, frameCount * MemoryLayout<Float>.size)
            return
        }
    }
}
