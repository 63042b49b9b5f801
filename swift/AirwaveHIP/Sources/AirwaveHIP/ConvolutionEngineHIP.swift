//  ConvolutionEngineHIP.swift
//  Same surface as `ConvolutionEngine` (Airwave/ConvolutionEngine.swift:14-408): failable init,
//  process(input:output:), process(input:output:frameCount:), processAndAccumulate, reset.

import Foundation
import CAirwaveHIP

public final class ConvolutionEngineHIP {
    private let handle: OpaquePointer
    public let blockSize: Int

    /// init?(hrirSamples:blockSize:)  ConvolutionEngine.swift:68 — nil when the engine cannot be built
    public init?(context: HIPContext, hrirSamples: [Float], blockSize: Int = 512) {
        var h: OpaquePointer?
        let st = hrirSamples.withUnsafeBufferPointer {
            aw_engine_create(context.handle, $0.baseAddress, Int32($0.count), Int32(blockSize), &h)
        }
        guard st == AW_OK, let e = h else { return nil }
        handle = e
        self.blockSize = blockSize
    }
    deinit { aw_engine_destroy(handle) }

    /// process(input:output:)  :232 — exactly blockSize frames
    public func process(input: UnsafePointer<Float>, output: UnsafeMutablePointer<Float>) {
        _ = aw_engine_process(handle, input, output)
    }

    /// process(input:output:frameCount:)  :370 — silently returns when count != blockSize
    public func process(input: [Float], output: inout [Float], frameCount: Int? = nil) {
        let count = frameCount ?? blockSize
        input.withUnsafeBufferPointer { i in
            output.withUnsafeMutableBufferPointer { o in
                guard let ia = i.baseAddress, let oa = o.baseAddress else { return }
                _ = aw_engine_process_n(handle, ia, oa, Int32(count))
            }
        }
    }

    /// processAndAccumulate(input:outputAccumulator:)  :388
    public func processAndAccumulate(input: UnsafePointer<Float>, outputAccumulator: UnsafeMutablePointer<Float>) {
        _ = aw_engine_process_accumulate(handle, input, outputAccumulator)
    }

    /// reset()  :397
    public func reset() { _ = aw_engine_reset(handle) }
}
