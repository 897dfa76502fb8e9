// zl_handoff.h -- what crosses from the callers' threads to the real-time cycle, HIP-free and lock-free on the cycle's side.
//
// The reference hands control to audio through SPSC rings without locks (SamplerSynth.cpp:328-341, SyncTimer.cpp:553-558) and lets
// its setters write plain fields that the voice reads per block (libzl.cpp:230-302, SamplerSynthVoice.cpp:189-196).  Two primitives
// give the same "the audio thread never waits" with defined behaviour under concurrent callers:
//   ZlRequestQueue  a bounded multi-producer / single-consumer queue (per-cell sequence numbers): play / stop / queue / timer calls
//   ZlSnapshot      a sequence lock around one POD record with a dirty flag: a clip's parameters
// Used by zl_libzl.cpp (product) and by tests/cpu_harness/handoff_tsan.cpp (ThreadSanitizer, CPU tier).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>

template <typename T, size_t CAP>
struct ZlRequestQueue {
    static_assert(std::is_trivially_copyable<T>::value, "requests are POD");
    struct Cell { std::atomic<size_t> seq; T r; };
    Cell cells[CAP];
    std::atomic<size_t> head{0}, tail{0};
    ZlRequestQueue() { for (size_t i = 0; i < CAP; ++i) cells[i].seq.store(i, std::memory_order_relaxed); }
    bool push(const T &r)                                          // any thread; false = full
    {
        size_t pos = tail.load(std::memory_order_relaxed);
        for (;;) {
            Cell &c = cells[pos % CAP];
            const size_t sq = c.seq.load(std::memory_order_acquire);
            const intptr_t d = (intptr_t)sq - (intptr_t)pos;
            if (d == 0) { if (tail.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) { c.r = r; c.seq.store(pos + 1, std::memory_order_release); return true; } }
            else if (d < 0) return false;
            else pos = tail.load(std::memory_order_relaxed);
        }
    }
    bool pop(T &r)                                                 // the cycle only; never waits (a producer caught between claiming its
    {                                                              // cell and filling it ends the drain: its request is taken next cycle)
        const size_t pos = head.load(std::memory_order_relaxed);
        Cell &c = cells[pos % CAP];
        if (c.seq.load(std::memory_order_acquire) != pos + 1) return false;
        r = c.r;
        c.seq.store(pos + CAP, std::memory_order_release);
        head.store(pos + 1, std::memory_order_relaxed);
        return true;
    }
    void clear() { T r; while (pop(r)) { } }
};

// Writers (serialised among themselves by the owner -- a mutex the reader never takes): publish().  Reader (one thread): take()
// returns true with a consistent copy if the record changed since the last successful take; a record caught in the middle of a write
// stays dirty and is taken at the next call -- the reader never waits.  The payload is copied word by word through relaxed atomics
// (a sequence lock's data accesses race by design; this keeps them defined).
template <typename T>
struct ZlSnapshot {
    static_assert(std::is_trivially_copyable<T>::value && sizeof(T) % 4 == 0, "snapshots are POD, a whole number of 32-bit words");
    std::atomic<uint32_t> seq{0};
    std::atomic<bool> dirty{false};
    std::atomic<uint32_t> words[sizeof(T) / 4];
    ZlSnapshot() { for (auto &w : words) w.store(0, std::memory_order_relaxed); }
    void publish(const T &v)
    {
        uint32_t tmp[sizeof(T) / 4];
        std::memcpy(tmp, &v, sizeof(T));
        const uint32_t s0 = seq.load(std::memory_order_relaxed);
        seq.store(s0 + 1, std::memory_order_relaxed);
        std::atomic_thread_fence(std::memory_order_release);
        for (size_t i = 0; i < sizeof(T) / 4; ++i) words[i].store(tmp[i], std::memory_order_relaxed);
        seq.store(s0 + 2, std::memory_order_release);
        dirty.store(true, std::memory_order_release);
    }
    bool take(T &out)
    {
        if (!dirty.load(std::memory_order_acquire)) return false;          // (a plain load first: the cycle looks at every clip, every cycle)
        // cleared with a read-modify-write: the reads of the record below may not move in front of it (a publication that lands
        // after the clearing sets the flag again; one that landed before is the one being read)
        (void)dirty.exchange(false, std::memory_order_acq_rel);
        const uint32_t s1 = seq.load(std::memory_order_acquire);
        if (!(s1 & 1u)) {
            uint32_t tmp[sizeof(T) / 4];
            for (size_t i = 0; i < sizeof(T) / 4; ++i) tmp[i] = words[i].load(std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_acquire);
            if (seq.load(std::memory_order_relaxed) == s1) { std::memcpy(&out, tmp, sizeof(T)); return true; }
        }
        dirty.store(true, std::memory_order_release);
        return false;
    }
};
