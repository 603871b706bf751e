// ORACLE — test infrastructure only.  Probe over the REAL reference code for the mesh build's data: compiled together with the
// reference's own src/terrain/block.cpp (which needs nothing but its vendored header-only glm) into oracle/_ref/libblockprobe.so.
// It exposes BlockUtils::init() / getBlockData() (block.cpp:11-159) and DirectionEnums::dirVecs (util/enums.hpp:43-50) so that
// the generated tables (oracle/mmo_blockdata.inc, csrc/mm_blockdata.cuh) are pinned against the reference itself.
#include <cstddef>
#include <vector>          // biome.hpp uses std::vector and relies on its includer for the header
#include <unordered_set>
#include "terrain/block.hpp"
#include "terrain/biome.hpp"
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

// ABI layout and constants of the generation path as the reference's own biome.hpp / block.hpp define them (same order as
// mmo_abi_layout in oracle/mmo_mesh.cpp, which reports include/mmgen_types.h)
extern "C" int ref_abi_layout(int* out)
{
    int n = 0;
    out[n++] = (int)sizeof(CaveLayer); out[n++] = (int)offsetof(CaveLayer, start); out[n++] = (int)offsetof(CaveLayer, end);
    out[n++] = (int)offsetof(CaveLayer, bottomBiome); out[n++] = (int)offsetof(CaveLayer, topBiome);
    out[n++] = (int)sizeof(FeaturePlacement); out[n++] = (int)offsetof(FeaturePlacement, feature); out[n++] = (int)offsetof(FeaturePlacement, pos);
    out[n++] = (int)offsetof(FeaturePlacement, canReplaceBlocks);
    out[n++] = (int)sizeof(CaveFeaturePlacement); out[n++] = (int)offsetof(CaveFeaturePlacement, feature); out[n++] = (int)offsetof(CaveFeaturePlacement, pos);
    out[n++] = (int)offsetof(CaveFeaturePlacement, layerHeight); out[n++] = (int)offsetof(CaveFeaturePlacement, canReplaceBlocks);
    out[n++] = MAX_CAVE_LAYERS_PER_COLUMN; out[n++] = MAX_GATHERED_FEATURES_PER_CHUNK; out[n++] = MAX_GATHERED_CAVE_FEATURES_PER_CHUNK;
    out[n++] = SEA_LEVEL; out[n++] = LAVA_LEVEL;
    out[n++] = numBiomes; out[n++] = numOceanBiomes; out[n++] = numOceanAndBeachBiomes; out[n++] = numCaveBiomes;
    out[n++] = numMaterials; out[n++] = numStratifiedMaterials; out[n++] = numForwardMaterials; out[n++] = numErodedMaterials;
    out[n++] = numFeatures; out[n++] = numCaveFeatures; out[n++] = numBlocks; out[n++] = numNonSolidBlocks;
    // a few named enumerators across the ranges (the counts above pin the last ones)
    out[n++] = (int)Block::BEDROCK; out[n++] = (int)Block::STONE; out[n++] = (int)Block::DEEPSLATE; out[n++] = (int)Block::BLACKSTONE; out[n++] = (int)Block::QUARTZ;
    out[n++] = (int)Biome::BEACH; out[n++] = (int)Biome::MESA; out[n++] = (int)Biome::CRYSTALS; out[n++] = (int)Biome::ARCHIPELAGO;
    out[n++] = (int)CaveBiome::CRYSTAL_CAVES; out[n++] = (int)CaveBiome::LUSH_CAVES; out[n++] = (int)CaveBiome::WARPED_FOREST;
    out[n++] = (int)Material::DIRT; out[n++] = (int)Material::SANDSTONE; out[n++] = (int)Material::GRAVEL;
    out[n++] = (int)Feature::ICEBERG; out[n++] = (int)Feature::PURPLE_MUSHROOM; out[n++] = (int)Feature::PALM_TREE;
    out[n++] = (int)CaveFeature::GLOWSTONE_CLUSTER; out[n++] = (int)CaveFeature::CRYSTAL_PILLAR;
    return n;
}
