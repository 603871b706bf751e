// Host-side decoder of the region wire format (include/mmgen.h, mmgen_pack_*): plain C++, no device code, so that it also builds
// under AddressSanitizer / UBSan on the CPU (`make -C oracle asan` fuzzes it with truncated and corrupted streams).
#include "../../include/mmgen.h"

extern "C" int mmgen_unpack_chunk_host(const uint8_t* packed, size_t packed_bytes, uint8_t* blocks)
{
    if (!packed || !blocks || packed_bytes < 512) return -1;
    size_t pos = 512;
    for (int col = 0; col < 256; ++col) {
        const unsigned runs = (unsigned)packed[2 * col] | ((unsigned)packed[2 * col + 1] << 8);      // u16 little endian, alignment-free
        int y = 0;
        for (unsigned r = 0; r < runs; ++r) {
            if (pos + 2 > packed_bytes) return -1;
            const uint8_t id = packed[pos];
            const int len = (int)packed[pos + 1] + 1;
            pos += 2;
            if (y + len > 384) return -1;
            for (int k = 0; k < len; ++k) blocks[384 * col + y++] = id;
        }
        if (y != 384) return -1;
    }
    return pos == packed_bytes ? 0 : -1;
}
