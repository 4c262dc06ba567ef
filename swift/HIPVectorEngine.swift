//  HIPVectorEngine.swift — the reference-side binding a Wax maintainer adds under
//  Sources/WaxVectorSearch/ to drop libwaxhip in beside MetalVectorEngine / USearchVectorEngine.
//
//  NOT compiled in this repository (the build image has no Swift toolchain); it is the literal
//  counterpart of wax_amd/engine.py, which IS exercised by the tests through the same C ABI.
//  Conforms to `VectorSearchEngine` (VectorSearchEngine.swift:10-18) and mirrors the concrete
//  surface callers use on MetalVectorEngine (isAvailable, init(metric:dimensions:), load(from:),
//  serialize/deserialize, addBatchStreaming).

#if canImport(CWaxHIP)
import CWaxHIP
import Foundation
import WaxCore

/// The engine handle as it crosses into `@Sendable` closures. Wax builds with Swift 6 strict concurrency
/// (Package.swift: tools-version 6.2, `StrictConcurrency` on every target) and `BlockingIOExecutor.run` takes a
/// `@Sendable` closure (BlockingIOExecutor.swift:19); an `OpaquePointer` is not `Sendable` (SE-0331), so capturing it
/// there does not compile. libwaxhip is internally synchronised (its own reader/writer lock, every entry point
/// re-entrant), which is what makes sharing the pointer sound — the same reasoning as
/// `extension USearchIndex: @retroactive @unchecked Sendable {}` (USearchSendable.swift:6). The wrapper also owns the
/// engine's lifetime: the actor needs no `deinit` (a nonisolated actor deinit may not touch non-Sendable state).
final class HIPEngineHandle: @unchecked Sendable {
    let raw: OpaquePointer
    init(_ raw: OpaquePointer) { self.raw = raw }
    deinit { wax_hip_engine_destroy(raw) }
}

public actor HIPVectorEngine {
    private let handle: HIPEngineHandle
    private let metric: VectorMetric
    public let dimensions: Int
    private var dirty = false
    // Blocking C calls hop onto a dedicated queue exactly like USearchVectorEngine does
    // (USearchVectorEngine.swift:66, 208); the library itself takes the reader/writer lock.
    private let io = BlockingIOExecutor(label: "wax.hip.vector", qos: .userInitiated)

    public static var isAvailable: Bool { wax_hip_available() != 0 }

    public init(metric: VectorMetric, dimensions: Int, device: Int32 = -1) throws {
        guard dimensions > 0 else { throw WaxError.invalidToc(reason: "dimensions must be > 0") }
        guard dimensions <= Constants.maxEmbeddingDimensions else {
            throw WaxError.capacityExceeded(limit: UInt64(Constants.maxEmbeddingDimensions), requested: UInt64(dimensions))
        }
        var h: OpaquePointer?
        try Self.check(wax_hip_engine_create(metric.toVecSimilarity().rawValue, UInt32(dimensions), device, &h))
        self.handle = HIPEngineHandle(h!)
        self.metric = metric
        self.dimensions = dimensions
    }

    /// One engine row-sharded over several GPUs of the node, inside the library (`wax_hip_engine_create_sharded`): same
    /// protocol, same results as one engine holding all the rows (tie order and `serialize()` bytes included). The
    /// scan of every query runs on all listed devices at once; only the per-shard top-k crosses xGMI.
    public init(metric: VectorMetric, dimensions: Int, devices: [Int32]) throws {
        guard dimensions > 0 else { throw WaxError.invalidToc(reason: "dimensions must be > 0") }
        guard dimensions <= Constants.maxEmbeddingDimensions else {
            throw WaxError.capacityExceeded(limit: UInt64(Constants.maxEmbeddingDimensions), requested: UInt64(dimensions))
        }
        var h: OpaquePointer?
        try Self.check(devices.withUnsafeBufferPointer {
            wax_hip_engine_create_sharded(metric.toVecSimilarity().rawValue, UInt32(dimensions), $0.baseAddress, Int32(devices.count), &h)
        })
        self.handle = HIPEngineHandle(h!)
        self.metric = metric
        self.dimensions = dimensions
    }

    /// Every gfx950 device of the node: pass it as `devices:` to opt into the row-sharded engine.
    public static var allDevices: [Int32] { (0..<wax_hip_device_count()).map { Int32($0) } }

    /// `devices: nil` (the default) = ONE device, like every other engine in this package. Row-sharding over several GPUs is
    /// opt-in (`devices: HIPVectorEngine.allDevices`, or an explicit list): the sharded handle has been verified on a single
    /// GPU only (every shard on device 0), and a loader must not silently take a multi-device code path no node has run yet.
    public static func load(from wax: Wax, metric: VectorMetric, dimensions: Int, devices: [Int32]? = nil) async throws -> HIPVectorEngine {
        let engine: HIPVectorEngine
        if let devices, devices.count > 1 {
            engine = try HIPVectorEngine(metric: metric, dimensions: dimensions, devices: devices)
        } else if let d = devices?.first {
            engine = try HIPVectorEngine(metric: metric, dimensions: dimensions, devices: [d])
        } else {
            engine = try HIPVectorEngine(metric: metric, dimensions: dimensions)
        }
        if let bytes = try await wax.readCommittedVecIndexBytes() { try await engine.deserialize(bytes) }
        for e in await wax.pendingEmbeddingMutations() { try await engine.add(frameId: e.frameId, vector: e.vector) }
        return engine
    }

    /// wax_hip_status -> WaxError (INTEGRATION.md §3)
    private static func check(_ rc: Int32) throws {
        guard rc != 0 else { return }
        let msg = String(cString: wax_hip_last_error())
        switch rc {
        case -1, -7: throw WaxError.encodingError(reason: msg)          // DIM_MISMATCH, INVALID_ARGUMENT
        case -2: throw WaxError.capacityExceeded(limit: UInt64(UInt32.max), requested: 0)
        default: throw WaxError.invalidToc(reason: msg)                 // NO_DEVICE, ALLOC, BAD_SEGMENT, METRIC, INTERNAL
        }
    }

    public func search(vector: [Float], topK: Int) async throws -> [(frameId: UInt64, score: Float)] {
        let h = handle
        let k32 = Int32(clamping: topK)
        // Sized by clampTopK alone and handed over as the capacity: the row count may grow between this line and the
        // search (another task's add runs while this one is suspended in io.run); the library never writes past `cap`.
        let cap = Int(wax_hip_result_capacity(k32))
        return try await io.run {
            var ids = [UInt64](repeating: 0, count: cap)
            var scores = [Float](repeating: 0, count: cap)
            var got: UInt32 = 0
            try Self.check(vector.withUnsafeBufferPointer { q in
                wax_hip_search(h.raw, q.baseAddress, UInt32(vector.count), k32, &ids, &scores, UInt32(cap), &got)
            })
            return (0..<Int(got)).map { (frameId: ids[$0], score: scores[$0]) }
        }
    }

    public func add(frameId: UInt64, vector: [Float]) async throws {
        let h = handle
        try await io.run {
            try Self.check(vector.withUnsafeBufferPointer { wax_hip_add(h.raw, frameId, $0.baseAddress, UInt32(vector.count)) })
        }
        dirty = true
    }

    public func addBatch(frameIds: [UInt64], vectors: [[Float]]) async throws {
        guard !frameIds.isEmpty else { return }
        guard frameIds.count == vectors.count else {
            throw WaxError.encodingError(reason: "addBatch: frameIds.count != vectors.count")
        }
        for v in vectors where v.count != dimensions {
            throw WaxError.encodingError(reason: "vector dimension mismatch: expected \(dimensions), got \(v.count)")
        }
        let flat = vectors.flatMap { $0 }   // row-major n x dims, one H2D copy inside the library
        let h = handle, d = UInt32(dimensions)
        try await io.run {
            try Self.check(flat.withUnsafeBufferPointer { rows in
                frameIds.withUnsafeBufferPointer { ids in
                    wax_hip_add_batch(h.raw, ids.baseAddress, rows.baseAddress, UInt64(frameIds.count), d)
                }
            })
        }
        dirty = true
    }

    public func addBatchStreaming(frameIds: [UInt64], vectors: [[Float]], chunkSize: Int = 256) async throws {
        for start in stride(from: 0, to: frameIds.count, by: chunkSize) {
            let end = min(start + chunkSize, frameIds.count)
            try await addBatch(frameIds: Array(frameIds[start..<end]), vectors: Array(vectors[start..<end]))
        }
    }

    /// Many queries in one call (`wax_hip_search_batch`): from 16 queries up the scan runs as a bf16 MFMA GEMM with an
    /// exact f32 re-score; every answer is certified exact or re-run on the single-query path, so the results equal
    /// `vectors.count` calls of `search(vector:topK:)`. For callers with several recalls pending.
    public func searchBatch(vectors: [[Float]], topK: Int) async throws -> [[(frameId: UInt64, score: Float)]] {
        guard !vectors.isEmpty else { return [] }
        for v in vectors where v.count != dimensions {
            throw WaxError.encodingError(reason: "vector dimension mismatch: expected \(dimensions), got \(v.count)")
        }
        let h = handle, d = UInt32(dimensions), nq = vectors.count
        let cap = Int(wax_hip_result_capacity(Int32(clamping: topK)))   // row stride of the result arrays
        let flat = vectors.flatMap { $0 }
        return try await io.run {
            var ids = [UInt64](repeating: 0, count: nq * cap)
            var scores = [Float](repeating: 0, count: nq * cap)
            var counts = [UInt32](repeating: 0, count: nq)
            try Self.check(flat.withUnsafeBufferPointer { q in
                wax_hip_search_batch(h.raw, q.baseAddress, UInt32(nq), d, Int32(clamping: topK), &ids, &scores, UInt32(cap), &counts)
            })
            return (0..<nq).map { i in (0..<Int(counts[i])).map { (frameId: ids[i * cap + $0], score: scores[i * cap + $0]) } }
        }
    }

    /// The vector lane's candidate filters on the device (`passesFrameFilter`, UnifiedSearch.swift:1241-1258):
    /// the best `topK` among the frames of `allowlist` (FrameFilter.frameIds), minus results below `minScore`.
    /// With this, UnifiedSearch no longer needs the 3x `candidateLimit` over-fetch (:1195-1200) for allow-listed requests.
    public func search(vector: [Float], topK: Int, allowlist: Set<UInt64>?, minScore: Float?) async throws -> [(frameId: UInt64, score: Float)] {
        let h = handle
        let limit = max(1, min(topK, 10_000))
        let allowed: [UInt64]? = allowlist.map(Array.init)
        return try await io.run {
            var ids = [UInt64](repeating: 0, count: limit)
            var scores = [Float](repeating: 0, count: limit)
            var n: UInt32 = 0
            try Self.check(vector.withUnsafeBufferPointer { q in
                (allowed ?? []).withUnsafeBufferPointer { a in
                    wax_hip_search_filtered(h.raw, q.baseAddress, UInt32(vector.count), Int32(clamping: topK),
                                            allowed == nil ? 0 : 1, a.baseAddress, UInt64(a.count),
                                            minScore == nil ? 0 : 1, minScore ?? 0, &ids, &scores, UInt32(limit), &n)
                }
            })
            return (0..<Int(n)).map { (frameId: ids[$0], score: scores[$0]) }
        }
    }

    /// Pending-embedding replay without the [[Float]] detour: WAL putEmbedding payloads (WALEntryCodec.encode,
    /// WALEntryCodec.swift:39-54) concatenated, validated and applied inside the library as one addBatch.
    /// UnifiedSearchEngineCache.applyPendingEmbeddingsIfNeeded (:252-283) can call this with the raw payloads.
    @discardableResult
    public func applyPutEmbeddings(payloads: Data) async throws -> Int {
        guard !payloads.isEmpty else { return 0 }
        let h = handle
        let applied: UInt64 = try await io.run {
            var n: UInt64 = 0
            try Self.check(payloads.withUnsafeBytes { raw in
                wax_hip_apply_put_embeddings(h.raw, raw.bindMemory(to: UInt8.self).baseAddress, UInt64(raw.count), &n)
            })
            return n
        }
        if applied > 0 { dirty = true }
        return Int(applied)
    }

    public func remove(frameId: UInt64) async throws {
        let h = handle
        try await io.run { try Self.check(wax_hip_remove(h.raw, frameId)) }
        dirty = true
    }

    public func serialize() async throws -> Data {
        let h = handle
        return try await io.run {
            var p: UnsafeMutablePointer<UInt8>?
            var len = 0
            try Self.check(wax_hip_serialize(h.raw, &p, &len))
            defer { wax_hip_free(p) }
            return Data(bytes: p!, count: len)   // "MV2V" encoding 2, byte-identical to MetalVectorEngine.serialize
        }
    }

    public func deserialize(_ data: Data) async throws {
        let h = handle
        try await io.run {
            try Self.check(data.withUnsafeBytes { wax_hip_deserialize(h.raw, $0.bindMemory(to: UInt8.self).baseAddress, data.count) })
        }
        dirty = false
    }

    public func stageForCommit(into wax: Wax) async throws {
        if !dirty { return }
        let blob = try await serialize()
        try await wax.stageVecIndexForNextCommit(bytes: blob, vectorCount: wax_hip_count(handle.raw),
                                                 dimension: UInt32(dimensions), similarity: metric.toVecSimilarity())
        dirty = false
    }
}

extension HIPVectorEngine: VectorSearchEngine {}
#endif
