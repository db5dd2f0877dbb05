// Host-side utilities of libemdenoise.so (no GPU work): CRC-32C for the TFRecord reader.
// TFRecord framing (the container misc_py/TFRecord_creator.py:57-85 writes through
// tf.python_io.TFRecordWriter): uint64 length | masked crc32c(length) | data | masked crc32c(data).
#include <nmmintrin.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

extern "C" uint32_t emd_crc32c(const void* data_host, size_t n, uint32_t crc) {
    const unsigned char* p = static_cast<const unsigned char*>(data_host);
    uint64_t c = crc ^ 0xffffffffu;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        c = _mm_crc32_u8((uint32_t)c, *p++);
        --n;
    }
    while (n >= 8) {
        uint64_t v;
        std::memcpy(&v, p, 8);
        c = _mm_crc32_u64(c, v);
        p += 8;
        n -= 8;
    }
    while (n--) c = _mm_crc32_u8((uint32_t)c, *p++);
    return (uint32_t)c ^ 0xffffffffu;
}
