// ORACLE — test infrastructure only (see oracle/README.md).
// CPU restatement of the mesh build that follows the generation path: Chunk::createVBOs (src/terrain/chunk.cu:1778-2003) and its static
// tables (:1753-1776), with BlockUtils::getBlockData (src/terrain/block.cpp:11-159, table generated as data by tools/gen_block_data.py),
// Vertex / Mats (src/rendering/structs.hpp:7-31) and DirectionEnums::dirVecs (src/util/enums.hpp:43-50).
//
// Output order is the reference's: voxels z-major, then x, then y; an X-shaped block emits 8 vertices + 12 indices, a cube
// emits, per displayed face in dirVecs order, 4 vertices + 6 indices (indices are local to the chunk).
// Parity status: PINNED like the stage functions of mmo_stages.cpp - Chunk::createVBOs and the five static tables in front of it are
// stated token for token as the reference writes them (tests/golden/ref_skeletons.json holds the reference's digests,
// tests/test_ref_literals.py compares; tools/extract_ref_literals.py lists what the normaliser treats as equal), over a view struct with
// the reference's member names; the render DATA is pinned to the reference's block.cpp compiled in place (ref_block_probe.cpp).
// xShapedPosOffset = 0.5f * sinf(radians(45.f)) is a static initialiser the reference evaluates with its HOST libm (MSVC); it is stated the
// same way here and evaluated by this host's libm.  Any libm that rounds sinf correctly gives 0x1.6a09e6p-2f (the argument lies 0.46 ulp
// from that value, 0.54 from its neighbour); tests/test_mesh.py holds this oracle and the constant compiled into the device mesher to it.
// (The path's own deterministic sine, 2 ulp by contract, lands on the neighbour here - it is not what a host initialiser uses.)
#include <array>
#include <cstdint>
#include <cstddef>
#include <cmath>
#include <cstring>
#include <vector>
#include "mmo_vec.h"
#include "mmo_math.h"
#include "mmo_noise.h"
#include "mmo_biome.h"
#include "../include/mmgen_types.h"      // the product's ABI types: reported by mmo_abi_layout for the pinning test only

using namespace mmo;

namespace {

struct TileUv { int u, v; };
struct BlockRender { TileUv side, top, bottom; int rot[3]; int flip[3]; int trans; };      // rot/flip order: side, top, bottom
const BlockRender kBlockRender[] = {
#include "mmo_blockdata.inc"
};
static_assert(sizeof(kBlockRender) / sizeof(kBlockRender[0]) == (size_t)numBlocks, "one render-data row per Block");

// ---- the reference's types as createVBOs sees them (block.hpp:156-222, rendering/structs.hpp:7-31, util/enums.hpp:43-50)
enum class Mats : size_t { M_DIFFUSE, M_WATER, M_CRYSTAL, M_SMOOTH_MICRO, M_MICRO, M_ROUGH_MICRO };
enum class TransparencyType : unsigned char { T_OPAQUE, T_SEMI_TRANSPARENT, T_TRANSPARENT, T_X_SHAPED };
struct SideUv { ivec2 uv{0}; bool randRot{false}; bool randFlip{false}; };
struct BlockUvs { SideUv side, top, bottom; };
struct BlockData { BlockUvs uvs; TransparencyType transparency; };
struct Vertex { vec3 pos; vec3 nor; vec2 uv; Mats m{Mats::M_DIFFUSE}; };      // structs.hpp:25-31: vec3, vec3, vec2, Mats : size_t
static_assert(sizeof(Vertex) == 40, "Vertex layout");
typedef unsigned int GLuint;

namespace BlockUtils {
BlockData getBlockData(Block block)                     // block.cpp:156-159 over the generated table
{
    const BlockRender& d = kBlockRender[(int)block];
    BlockData out;
    out.uvs.side = SideUv{ivec2(d.side.u, d.side.v), d.rot[0] != 0, d.flip[0] != 0};
    out.uvs.top = SideUv{ivec2(d.top.u, d.top.v), d.rot[1] != 0, d.flip[1] != 0};
    out.uvs.bottom = SideUv{ivec2(d.bottom.u, d.bottom.v), d.rot[2] != 0, d.flip[2] != 0};
    out.transparency = (TransparencyType)d.trans;
    return out;
}
}  // namespace BlockUtils
namespace DirectionEnums {
const std::array<ivec3, 6> dirVecs = {ivec3(0, 0, 1), ivec3(1, 0, 0), ivec3(0, 0, -1), ivec3(-1, 0, 0), ivec3(0, 1, 0), ivec3(0, -1, 0)};      // enums.hpp:43-50
}
template <int xSize = 16> int posTo2dIndex(const int x, const int z) { return x + xSize * z; }
template <int xSize = 16, int ySize = 384> int posTo3dIndex(const ivec3 pos) { return pos.y + ySize * posTo2dIndex<xSize>(pos.x, pos.z); }      // biomeFuncs.hpp:25-37

// The reference's Chunk as createVBOs sees it (chunk.hpp:37-72): the same member names over the caller's block arrays.
struct BlocksView {
    const Block* p;
    Block operator[](size_t i) const { return p[i]; }
};
struct Chunk {
    ivec3 worldBlockPos;
    BlocksView blocks;
    std::array<Chunk*, 4> neighbors;
    std::vector<GLuint> idx;
    std::vector<Vertex> verts;
    int idxCount;
    void createVBOs();
};

// ---- chunk.cu:1753-1776, as written
static const float xShapedPosOffset = 0.5f * sinf(g_radians(45.f));      // (the HOST libm, like the reference: see the header)
static const std::array<vec3, 8> xShapedVertPositions = {
    vec3(xShapedPosOffset, 0.f, xShapedPosOffset),
    vec3(-xShapedPosOffset, 0.f, -xShapedPosOffset),
    vec3(-xShapedPosOffset, 1.f, -xShapedPosOffset),
    vec3(xShapedPosOffset, 1.f, xShapedPosOffset),

    vec3(-xShapedPosOffset, 0.f, xShapedPosOffset),
    vec3(xShapedPosOffset, 0.f, -xShapedPosOffset),
    vec3(xShapedPosOffset, 1.f, -xShapedPosOffset),
    vec3(-xShapedPosOffset, 1.f, xShapedPosOffset)
};
static const vec3 xShapedFaceNormal1 = g_normalize(vec3(1, 0, -1));
static const vec3 xShapedFaceNormal2 = g_normalize(vec3(1, 0, 1));

static const std::array<ivec3, 24> directionVertPositions = {
    ivec3(0, 0, 1), ivec3(1, 0, 1), ivec3(1, 1, 1), ivec3(0, 1, 1),
    ivec3(1, 0, 1), ivec3(1, 0, 0), ivec3(1, 1, 0), ivec3(1, 1, 1),
    ivec3(1, 0, 0), ivec3(0, 0, 0), ivec3(0, 1, 0), ivec3(1, 1, 0),
    ivec3(0, 0, 0), ivec3(0, 0, 1), ivec3(0, 1, 1), ivec3(0, 1, 0),
    ivec3(0, 1, 1), ivec3(1, 1, 1), ivec3(1, 1, 0), ivec3(0, 1, 0),
    ivec3(0, 0, 0), ivec3(1, 0, 0), ivec3(1, 0, 1), ivec3(0, 0, 1)
};

static const std::array<ivec2, 4> uvOffsets = {
    ivec2(0, 0), ivec2(1, 0), ivec2(1, 1), ivec2(0, 1)
};

// chunk.cu:1778-2003, as written.  (Like the reference's, the two switches leave `shouldDisplay` / `sideUv` to the cases that can occur: an
// X-shaped block never reaches the face loop.)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wswitch"
#pragma GCC diagnostic ignored "-Wmaybe-uninitialized"
void Chunk::createVBOs()
{
    idx.clear();
    verts.clear();

    idxCount = 0;

    for (int z = 0; z < 16; ++z)
    {
        for (int x = 0; x < 16; ++x)
        {
            for (int y = 0; y < 384; ++y)
            {
                ivec3 thisPos = ivec3(x, y, z);
                Block thisBlock = blocks[posTo3dIndex(thisPos)];

                Mats mat;
                switch (thisBlock)
                {
                case Block::AIR:
                    continue;
                case Block::WATER:
                    mat = Mats::M_WATER;
                    break;
                case Block::CYAN_CRYSTAL:
                case Block::GREEN_CRYSTAL:
                case Block::MAGENTA_CRYSTAL:
                    mat = Mats::M_CRYSTAL;
                    break;
                case Block::MARBLE:
                case Block::QUARTZ:
                case Block::ICE:
                case Block::PACKED_ICE:
                case Block::BLUE_ICE:
                    mat = Mats::M_SMOOTH_MICRO;
                    break;
                case Block::SNOW:
                case Block::SNOWY_GRASS_BLOCK:
                    mat = Mats::M_MICRO;
                    break;
                case Block::SAND:
                case Block::GRAVEL:
                    mat = Mats::M_ROUGH_MICRO;
                    break;
                default:
                    mat = Mats::M_DIFFUSE;
                    break;
                }

                BlockData thisBlockData = BlockUtils::getBlockData(thisBlock);
                const auto thisTrans = thisBlockData.transparency;

                if (thisTrans == TransparencyType::T_X_SHAPED)
                {
                    vec3 basePos = vec3(x + 0.5f, y, z + 0.5f);

                    vec2 worldBlockXZ = vec2(this->worldBlockPos.x + x, this->worldBlockPos.z + z);
                    vec2 randomOffset = 0.4f * (rand2From2(worldBlockXZ) - 0.5f);
                    basePos.x += randomOffset.x;
                    basePos.z += randomOffset.y;

                    int idx1 = verts.size();

                    for (int i = 0; i < 8; i++)
                    {
                        auto posOffset = xShapedVertPositions[i];

                        verts.emplace_back();
                        Vertex& vert = verts.back();

                        vert.pos = basePos + posOffset;
                        vert.nor = i < 4 ? xShapedFaceNormal1 : xShapedFaceNormal2;
                        vert.uv = vec2(thisBlockData.uvs.side.uv + uvOffsets[i % 4]) * 0.0625f;
                        vert.m = mat;
                    }

                    idx.push_back(idx1);
                    idx.push_back(idx1 + 1);
                    idx.push_back(idx1 + 2);
                    idx.push_back(idx1);
                    idx.push_back(idx1 + 2);
                    idx.push_back(idx1 + 3);

                    idx.push_back(idx1 + 4);
                    idx.push_back(idx1 + 5);
                    idx.push_back(idx1 + 6);
                    idx.push_back(idx1 + 4);
                    idx.push_back(idx1 + 6);
                    idx.push_back(idx1 + 7);

                    continue;
                }

                for (int dirIdx = 0; dirIdx < 6; ++dirIdx)
                {
                    const auto& direction = DirectionEnums::dirVecs[dirIdx];
                    ivec3 neighborPos = thisPos + direction;
                    Chunk* neighborPosChunk = this;
                    Block neighborBlock;

                    if (neighborPos.y >= 0 && neighborPos.y < 384)
                    {
                        if (neighborPos.x < 0)
                        {
                            neighborPosChunk = neighbors[3];
                            neighborPos.x += 16;
                        }
                        else if (neighborPos.x >= 16)
                        {
                            neighborPosChunk = neighbors[1];
                            neighborPos.x -= 16;
                        }
                        else if (neighborPos.z < 0)
                        {
                            neighborPosChunk = neighbors[2];
                            neighborPos.z += 16;
                        }
                        else if (neighborPos.z >= 16)
                        {
                            neighborPosChunk = neighbors[0];
                            neighborPos.z -= 16;
                        }

                        if (neighborPosChunk == nullptr)
                        {
                            continue;
                        }

                        neighborBlock = neighborPosChunk->blocks[posTo3dIndex(neighborPos)];

                        const auto neighborTrans = BlockUtils::getBlockData(neighborBlock).transparency;

                        // OPAQUE displays if neighbor is not OPAQUE
                        // SEMI_TRANSPARENT if neighbor is not OPAQUE
                        // TRANSPARENT (except AIR) displays if neighbor is AIR or SEMI_TRANSPARENT
                        // X_SHAPED displays no matter what (handled above)
                        bool shouldDisplay;
                        switch (thisTrans)
                        {
                        case TransparencyType::T_OPAQUE:
                        case TransparencyType::T_SEMI_TRANSPARENT:
                            shouldDisplay = neighborTrans != TransparencyType::T_OPAQUE;
                            break;
                        case TransparencyType::T_TRANSPARENT:
                            shouldDisplay = neighborBlock == Block::AIR || neighborTrans == TransparencyType::T_SEMI_TRANSPARENT;
                            break;
                        }

                        if (!shouldDisplay)
                        {
                            continue;
                        }
                    }

                    int idx1 = verts.size();

                    const auto& thisUvs = thisBlockData.uvs;
                    SideUv sideUv;
                    switch (direction.y)
                    {
                    case 1:
                        sideUv = thisUvs.top;
                        break;
                    case -1:
                        sideUv = thisUvs.bottom;
                        break;
                    case 0:
                        sideUv = thisUvs.side;
                        break;
                    }

                    int uvStartIdx = 0;
                    int uvFlipIdx = -1;
                    if (sideUv.randRot || sideUv.randFlip)
                    {
                        ivec3 worldPos = thisPos + this->worldBlockPos;
                        auto rng = makeSeededRandomEngine(worldPos.x, worldPos.y, worldPos.z, dirIdx);
                        uniform_real_distribution<float> u04(0, 4);
                        if (sideUv.randRot)
                        {
                            uvStartIdx = (int)u04(rng);
                        }
                        if (sideUv.randFlip)
                        {
                            uvFlipIdx = (int)u04(rng);
                        }
                    }

                    for (int j = 0; j < 4; ++j)
                    {
                        verts.emplace_back();
                        Vertex& vert = verts.back();

                        vert.pos = vec3(thisPos + directionVertPositions[dirIdx * 4 + j]);
                        vert.nor = direction;

                        ivec2 uvOffset = uvOffsets[(uvStartIdx + j) % 4];
                        if (uvFlipIdx != -1)
                        {
                            if (uvFlipIdx & 1)
                            {
                                uvOffset.x = 1 - uvOffset.x;
                            }
                            if (uvFlipIdx & 2)
                            {
                                uvOffset.y = 1 - uvOffset.y;
                            }
                        }
                        vert.uv = vec2(sideUv.uv + uvOffset) * 0.0625f;
                        vert.m = mat;
                    }

                    idx.push_back(idx1);
                    idx.push_back(idx1 + 1);
                    idx.push_back(idx1 + 2);
                    idx.push_back(idx1);
                    idx.push_back(idx1 + 2);
                    idx.push_back(idx1 + 3);
                }
            }
        }
    }
}

}  // namespace

// the oracle's render-data table and face directions, in the layout of oracle/ref_block_probe.cpp (tests pin them to the reference)
extern "C" void mmo_block_data(int* out)
{
    for (int b = 0; b < numBlocks; ++b) {
        const BlockRender& d = kBlockRender[b];
        int* o = out + 13 * b;
        o[0] = d.side.u; o[1] = d.side.v; o[2] = d.top.u; o[3] = d.top.v; o[4] = d.bottom.u; o[5] = d.bottom.v;
        for (int k = 0; k < 3; ++k) { o[6 + k] = d.rot[k]; o[9 + k] = d.flip[k]; }
        o[12] = d.trans;
    }
}
extern "C" void mmo_dir_vecs(int* out18) { for (int d = 0; d < 6; ++d) for (int k = 0; k < 3; ++k) out18[3 * d + k] = k == 0 ? DirectionEnums::dirVecs[d].x : (k == 1 ? DirectionEnums::dirVecs[d].y : DirectionEnums::dirVecs[d].z); }

// ABI layout and constants of include/mmgen_types.h, in the order of ref_abi_layout (oracle/ref_block_probe.cpp)
extern "C" int mmo_abi_layout(int* out)
{
    int n = 0;
    out[n++] = (int)sizeof(mmgen_cave_layer); out[n++] = (int)offsetof(mmgen_cave_layer, start); out[n++] = (int)offsetof(mmgen_cave_layer, end);
    out[n++] = (int)offsetof(mmgen_cave_layer, bottom_biome); out[n++] = (int)offsetof(mmgen_cave_layer, top_biome);
    out[n++] = (int)sizeof(mmgen_feature_placement); out[n++] = (int)offsetof(mmgen_feature_placement, feature); out[n++] = (int)offsetof(mmgen_feature_placement, pos);
    out[n++] = (int)offsetof(mmgen_feature_placement, can_replace_blocks);
    out[n++] = (int)sizeof(mmgen_cave_feature_placement); out[n++] = (int)offsetof(mmgen_cave_feature_placement, feature);
    out[n++] = (int)offsetof(mmgen_cave_feature_placement, pos); out[n++] = (int)offsetof(mmgen_cave_feature_placement, layer_height);
    out[n++] = (int)offsetof(mmgen_cave_feature_placement, can_replace_blocks);
    out[n++] = MMGEN_MAX_CAVE_LAYERS_PER_COLUMN; out[n++] = MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK; out[n++] = MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK;
    out[n++] = MMGEN_SEA_LEVEL; out[n++] = MMGEN_LAVA_LEVEL;
    out[n++] = MMGEN_NUM_BIOMES; out[n++] = MMGEN_NUM_OCEAN_BIOMES; out[n++] = MMGEN_NUM_OCEAN_AND_BEACH_BIOMES; out[n++] = MMGEN_NUM_CAVE_BIOMES;
    out[n++] = MMGEN_NUM_MATERIALS; out[n++] = MMGEN_NUM_STRATIFIED_MATERIALS; out[n++] = MMGEN_NUM_FORWARD_MATERIALS; out[n++] = MMGEN_NUM_ERODED_MATERIALS;
    out[n++] = MMGEN_NUM_FEATURES; out[n++] = MMGEN_NUM_CAVE_FEATURES; out[n++] = MMB_NUM_BLOCKS; out[n++] = MMB_NUM_NON_SOLID_BLOCKS;
    out[n++] = MMB_BEDROCK; out[n++] = MMB_STONE; out[n++] = MMB_DEEPSLATE; out[n++] = MMB_BLACKSTONE; out[n++] = MMB_QUARTZ;
    out[n++] = MMBIO_BEACH; out[n++] = MMBIO_MESA; out[n++] = MMBIO_CRYSTALS; out[n++] = MMBIO_ARCHIPELAGO;
    out[n++] = MMCB_CRYSTAL_CAVES; out[n++] = MMCB_LUSH_CAVES; out[n++] = MMCB_WARPED_FOREST;
    out[n++] = MMM_DIRT; out[n++] = MMM_SANDSTONE; out[n++] = MMM_GRAVEL;
    out[n++] = MMF_ICEBERG; out[n++] = MMF_PURPLE_MUSHROOM; out[n++] = MMF_PALM_TREE;
    out[n++] = MMCF_GLOWSTONE_CLUSTER; out[n++] = MMCF_CRYSTAL_PILLAR;
    return n;
}

#pragma GCC diagnostic pop

// Returns the number of vertices (and *nIdxOut indices) the chunk produces; writes at most capVerts / capIdx of them (either
// output may be null to count only).  neighbors: N(+z), E(+x), S(-z), W(-x) block arrays, null = chunk absent (faces skipped).
extern "C" long mmo_create_vbos(const uint8_t* blocks, const uint8_t* const neighbors[4], int worldBlockX, int worldBlockZ,
                                void* vertsOut, uint32_t* idxOut, long capVerts, long capIdx, long* nIdxOut)
{
    Chunk around[4];
    Chunk chunk;
    chunk.worldBlockPos = ivec3(worldBlockX, 0, worldBlockZ);
    chunk.blocks = BlocksView{(const Block*)blocks};
    for (int k = 0; k < 4; ++k) {
        around[k].blocks = BlocksView{(const Block*)neighbors[k]};
        chunk.neighbors[k] = neighbors[k] ? &around[k] : nullptr;
    }
    chunk.createVBOs();
    const long nv = (long)chunk.verts.size(), ni = (long)chunk.idx.size();
    if (vertsOut) std::memcpy(vertsOut, chunk.verts.data(), sizeof(Vertex) * (size_t)(nv < capVerts ? nv : capVerts));
    if (idxOut) std::memcpy(idxOut, chunk.idx.data(), sizeof(uint32_t) * (size_t)(ni < capIdx ? ni : capIdx));
    if (nIdxOut) *nIdxOut = ni;
    return nv;
}

// xShapedPosOffset as this oracle computes it (the product compiles the correctly rounded constant in: tests hold them together)
extern "C" float mmo_x_shaped_pos_offset() { return xShapedPosOffset; }
