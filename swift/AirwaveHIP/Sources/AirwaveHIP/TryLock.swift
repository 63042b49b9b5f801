//  TryLock.swift
//  A lock-guarded value with a non-blocking read for the render thread — the two operations the reference takes from
//  `OSAllocatedUnfairLock` (Airwave/HRIRManager.swift:134, Airwave/EqualizerRuntimeEffect.swift:6): `withLock` for
//  writers, `withLockIfAvailable` for the one try-lock snapshot per callback.  `OSAllocatedUnfairLock` is Darwin-only
//  (`import os`); libairwave_hip.so is a ROCm (Linux) library, so the package uses pthread mutexes, which both have.

#if canImport(Glibc)
import Glibc
#elseif canImport(Darwin)
import Darwin
#endif

final class TryLock<Value> {
    private let mutex: UnsafeMutablePointer<pthread_mutex_t>
    private var value: Value

    init(initialState: Value) {
        mutex = UnsafeMutablePointer<pthread_mutex_t>.allocate(capacity: 1)
        mutex.initialize(to: pthread_mutex_t())
        pthread_mutex_init(mutex, nil)
        value = initialState
    }

    deinit {
        pthread_mutex_destroy(mutex)
        mutex.deinitialize(count: 1)
        mutex.deallocate()
    }

    /// Control thread: may block.
    @discardableResult
    func withLock<R>(_ body: (inout Value) throws -> R) rethrows -> R {
        pthread_mutex_lock(mutex)
        defer { pthread_mutex_unlock(mutex) }
        return try body(&value)
    }

    /// Render thread: never blocks; nil when a writer holds the lock (the caller keeps what it already has).
    func withLockIfAvailable<R>(_ body: (inout Value) throws -> R) rethrows -> R? {
        guard pthread_mutex_trylock(mutex) == 0 else { return nil }
        defer { pthread_mutex_unlock(mutex) }
        return try body(&value)
    }
}
