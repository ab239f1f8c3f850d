// host_pin.h -- page-locked host staging buffers for the frame layer (frame.cpp), with a process-wide cache.
//
// A FrameEncoder / FrameDecoder stages whole batches of blocks on the host (the reference's `src` / `dst` vectors,
// src/frame/compress.rs:62-93, src/frame/decompress.rs:62-72, sized for a launch instead of a block).  As std::vectors
// those buffers cost more than the GPU work on them: every one-shot frame call allocated them anew (a page fault per
// 4 KiB on first touch: 0.3-0.5 ms per 4 MiB buffer) and every transfer from pageable memory goes through the runtime's
// own bounce buffers.  PinBuf is the part of std::vector<uint8_t> the frame layer uses (data / size / resize that keeps
// the contents / clear), on memory from hipHostMalloc: transfers are plain DMA, and released buffers go to a small cache
// (power-of-two sizes, at most CACHE_BYTES kept) so that the next frame finds its buffers mapped and pinned.
// Without a usable device (header-only callers, the CPU test suite) the memory comes from malloc.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <new>

namespace lz4flex {

// cap is rounded up to a power of two >= 64 KiB; *pinned tells release how the block was obtained.  nullptr: out of memory
uint8_t* pin_acquire(size_t min_cap, size_t* cap, bool* pinned);
void pin_release(uint8_t* p, size_t cap, bool pinned);

class PinBuf {
  public:
    PinBuf() = default;
    PinBuf(const PinBuf&) = delete;
    PinBuf& operator=(const PinBuf&) = delete;
    ~PinBuf() { if (p_) pin_release(p_, cap_, pinned_); }
    uint8_t* data() { return p_; }
    const uint8_t* data() const { return p_; }
    size_t size() const { return n_; }
    void clear() { n_ = 0; }
    // like vector::resize (std::bad_alloc included), except that new bytes are NOT initialised and growth doubles
    void resize(size_t n) {
        if (n > cap_) {
            size_t ncap = 0; bool npin = false;
            uint8_t* q = pin_acquire(n > 2 * cap_ ? n : 2 * cap_, &ncap, &npin);
            if (!q) throw std::bad_alloc();
            if (n_) memcpy(q, p_, n_);
            if (p_) pin_release(p_, cap_, pinned_);
            p_ = q; cap_ = ncap; pinned_ = npin;
        }
        n_ = n;
    }
  private:
    uint8_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool pinned_ = false;
};

}  // namespace lz4flex
