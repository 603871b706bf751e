// ORACLE — test infrastructure only.  Probe over the REAL reference code for the mesh build's data: compiled together with the
// reference's own src/terrain/block.cpp (which needs nothing but its vendored header-only glm) into oracle/_ref/libblockprobe.so.
// It exposes BlockUtils::init() / getBlockData() (block.cpp:11-159) and DirectionEnums::dirVecs (util/enums.hpp:43-50) so that
// the generated tables (oracle/mmo_blockdata.inc, csrc/mm_blockdata.cuh) are pinned against the reference itself.
#include "terrain/block.hpp"
#include "util/enums.hpp"

extern "C" int ref_num_blocks() { return numBlocks; }

// out[b][13]: side.u side.v top.u top.v bottom.u bottom.v, randRot side top bottom, randFlip side top bottom, transparency
extern "C" void ref_block_data(int* out)
{
    BlockUtils::init();
    for (int b = 0; b < numBlocks; ++b) {
        const BlockData d = BlockUtils::getBlockData((Block)b);
        int* o = out + 13 * b;
        o[0] = d.uvs.side.uv.x; o[1] = d.uvs.side.uv.y; o[2] = d.uvs.top.uv.x; o[3] = d.uvs.top.uv.y; o[4] = d.uvs.bottom.uv.x; o[5] = d.uvs.bottom.uv.y;
        o[6] = d.uvs.side.randRot; o[7] = d.uvs.top.randRot; o[8] = d.uvs.bottom.randRot;
        o[9] = d.uvs.side.randFlip; o[10] = d.uvs.top.randFlip; o[11] = d.uvs.bottom.randFlip;
        o[12] = (int)d.transparency;
    }
}

extern "C" void ref_dir_vecs(int* out18)
{
    for (int d = 0; d < 6; ++d) { out18[3 * d] = DirectionEnums::dirVecs[d].x; out18[3 * d + 1] = DirectionEnums::dirVecs[d].y; out18[3 * d + 2] = DirectionEnums::dirVecs[d].z; }
}
