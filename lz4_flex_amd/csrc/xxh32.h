// xxh32.h -- XXH32 (one-shot and streaming) for the LZ4 frame format's header, block and
// content checksums.  The reference gets it from the third-party crate twox-hash 2.x
// (reference Cargo.toml:50; call sites src/frame/header.rs:266-268, src/frame/compress.rs:314,320,
// src/frame/decompress.rs:178-187).  Written from the public XXH32 specification.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace lz4flex {

class XxHash32 {
   public:
    explicit XxHash32(uint32_t seed = 0) { reset(seed); }
    void reset(uint32_t seed) {
        v_[0] = seed + kP1 + kP2; v_[1] = seed + kP2; v_[2] = seed; v_[3] = seed - kP1;
        seed_ = seed; total_ = 0; buffered_ = 0;
    }
    void write(const uint8_t* p, size_t len) {
        total_ += len;
        if (buffered_ + len < 16) { std::memcpy(buf_ + buffered_, p, len); buffered_ += len; return; }
        if (buffered_) {
            const size_t fill = 16 - buffered_;
            std::memcpy(buf_ + buffered_, p, fill);
            stripe(buf_);
            p += fill; len -= fill; buffered_ = 0;
        }
        while (len >= 16) { stripe(p); p += 16; len -= 16; }
        if (len) { std::memcpy(buf_, p, len); buffered_ = len; }
    }
    uint32_t finish() const {
        uint32_t h = total_ >= 16 ? rotl(v_[0], 1) + rotl(v_[1], 7) + rotl(v_[2], 12) + rotl(v_[3], 18) : seed_ + kP5;
        h += (uint32_t)total_;
        const uint8_t* p = buf_;
        size_t len = buffered_;
        while (len >= 4) { h = rotl(h + rd32(p) * kP3, 17) * kP4; p += 4; len -= 4; }
        while (len) { h = rotl(h + (*p) * kP5, 11) * kP1; ++p; --len; }
        h ^= h >> 15; h *= kP2; h ^= h >> 13; h *= kP3; h ^= h >> 16;
        return h;
    }
    static uint32_t oneshot(uint32_t seed, const uint8_t* p, size_t len) {
        XxHash32 h(seed);
        h.write(p, len);
        return h.finish();
    }

   private:
    static constexpr uint32_t kP1 = 2654435761u, kP2 = 2246822519u, kP3 = 3266489917u, kP4 = 668265263u,
                              kP5 = 374761393u;
    static uint32_t rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
    static uint32_t rd32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
    void stripe(const uint8_t* p) {
        for (int i = 0; i < 4; i++) v_[i] = rotl(v_[i] + rd32(p + 4 * i) * kP2, 13) * kP1;
    }
    uint32_t v_[4];
    uint32_t seed_;
    uint64_t total_;
    uint8_t buf_[16];
    size_t buffered_;
};

}  // namespace lz4flex
