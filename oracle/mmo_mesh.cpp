// ORACLE — test infrastructure only (see oracle/README.md).
// CPU restatement of the mesh build that follows the generation path: Chunk::createVBOs (src/terrain/chunk.cu:1751-2003),
// with BlockUtils::getBlockData (src/terrain/block.cpp:11-159, table generated as data by tools/gen_block_data.py),
// Vertex / Mats (src/rendering/structs.hpp:7-31) and DirectionEnums::dirVecs (src/util/enums.hpp:43-50).
//
// Output order is the reference's: voxels z-major, then x, then y; an X-shaped block emits 8 vertices + 12 indices, a cube
// emits, per displayed face in dirVecs order, 4 vertices + 6 indices (indices are local to the chunk).
// Parity status: unpinned (the reference has no fixture for it); xShapedPosOffset = 0.5f * sinf(radians(45.f)) is a host-libm
// value in the reference and is frozen here as the correctly rounded constant.
#include <cstdint>
#include <cstddef>
#include "mmo_vec.h"
#include "mmo_math.h"
#include "mmo_noise.h"
#include "mmo_biome.h"
#include "../include/mmgen_types.h"      // the product's ABI types: reported by mmo_abi_layout for the pinning test only

using namespace mmo;

namespace {

struct SideUv { int u, v; };
struct BlockRender { SideUv side, top, bottom; int rot[3]; int flip[3]; int trans; };      // rot/flip order: side, top, bottom
const BlockRender kBlockRender[] = {
#include "mmo_blockdata.inc"
};
enum { T_OPAQUE = 0, T_SEMI_TRANSPARENT = 1, T_TRANSPARENT = 2, T_X_SHAPED = 3 };
enum { M_DIFFUSE = 0, M_WATER, M_CRYSTAL, M_SMOOTH_MICRO, M_MICRO, M_ROUGH_MICRO };      // structs.hpp:7-14

static_assert(sizeof(kBlockRender) / sizeof(kBlockRender[0]) == (size_t)numBlocks, "one render-data row per Block");

// material class of a block (switch of chunk.cu:1797-1829)
int mat_of(Block b)
{
    switch (b) {
    case Block::WATER: return M_WATER;
    case Block::CYAN_CRYSTAL: case Block::GREEN_CRYSTAL: case Block::MAGENTA_CRYSTAL: return M_CRYSTAL;
    case Block::MARBLE: case Block::QUARTZ: case Block::ICE: case Block::PACKED_ICE: case Block::BLUE_ICE: return M_SMOOTH_MICRO;
    case Block::SNOW: case Block::SNOWY_GRASS_BLOCK: return M_MICRO;
    case Block::SAND: case Block::GRAVEL: return M_ROUGH_MICRO;
    default: return M_DIFFUSE;
    }
}

struct Vertex { float pos[3]; float nor[3]; float uv[2]; uint64_t m; };      // structs.hpp:25-31: vec3, vec3, vec2, Mats : size_t
static_assert(sizeof(Vertex) == 40, "Vertex layout");

const float kXOff = 0x1.6a09e6p-2f;                  // 0.5f * sin(radians(45)) correctly rounded (0.35355338f)
const float kXPos[8][3] = {                          // xShapedVertPositions, chunk.cu:1754-1764
    {kXOff, 0.f, kXOff}, {-kXOff, 0.f, -kXOff}, {-kXOff, 1.f, -kXOff}, {kXOff, 1.f, kXOff},
    {-kXOff, 0.f, kXOff}, {kXOff, 0.f, -kXOff}, {kXOff, 1.f, -kXOff}, {-kXOff, 1.f, kXOff}};
const int kDir[6][3] = {{0, 0, 1}, {1, 0, 0}, {0, 0, -1}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}};      // enums.hpp:43-50
const int kDirVert[24][3] = {                        // directionVertPositions, chunk.cu:1768-1775
    {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}, {1, 0, 1}, {1, 0, 0}, {1, 1, 0}, {1, 1, 1}, {1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 0},
    {0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {0, 1, 0}, {0, 1, 1}, {1, 1, 1}, {1, 1, 0}, {0, 1, 0}, {0, 0, 0}, {1, 0, 0}, {1, 0, 1}, {0, 0, 1}};
const int kUvOff[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};

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
extern "C" void mmo_dir_vecs(int* out18) { for (int d = 0; d < 6; ++d) for (int k = 0; k < 3; ++k) out18[3 * d + k] = kDir[d][k]; }

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

// Returns the number of vertices (and *nIdxOut indices) the chunk produces; writes at most capVerts / capIdx of them (either
// output may be null to count only).  neighbors: N(+z), E(+x), S(-z), W(-x) block arrays, null = chunk absent (faces skipped).
extern "C" long mmo_create_vbos(const uint8_t* blocks, const uint8_t* const neighbors[4], int worldBlockX, int worldBlockZ,
                                void* vertsOut, uint32_t* idxOut, long capVerts, long capIdx, long* nIdxOut)
{
    Vertex* verts = (Vertex*)vertsOut;
    long nv = 0, ni = 0;
    const vec3 xNor1 = g_normalize(vec3(1.f, 0.f, -1.f)), xNor2 = g_normalize(vec3(1.f, 0.f, 1.f));
    auto put = [&](const Vertex& v) { if (nv < capVerts && verts) verts[nv] = v; ++nv; };
    auto puti = [&](uint32_t i) { if (ni < capIdx && idxOut) idxOut[ni] = i; ++ni; };
    for (int z = 0; z < 16; ++z)
        for (int x = 0; x < 16; ++x)
            for (int y = 0; y < 384; ++y) {
                const uint8_t b = blocks[y + 384 * (x + 16 * z)];
                if (b == (uint8_t)Block::AIR) continue;
                const uint64_t mat = (uint64_t)mat_of((Block)b);
                const BlockRender& bd = kBlockRender[b];
                if (bd.trans == T_X_SHAPED) {
                    vec3 base((float)x + 0.5f, (float)y, (float)z + 0.5f);
                    const vec2 wxz((float)(worldBlockX + x), (float)(worldBlockZ + z));
                    const vec2 off = 0.4f * (rand2From2(wxz) - 0.5f);
                    base.x += off.x;
                    base.z += off.y;
                    const uint32_t i1 = (uint32_t)nv;
                    for (int i = 0; i < 8; ++i) {
                        Vertex v;
                        v.pos[0] = base.x + kXPos[i][0]; v.pos[1] = base.y + kXPos[i][1]; v.pos[2] = base.z + kXPos[i][2];
                        const vec3 n = i < 4 ? xNor1 : xNor2;
                        v.nor[0] = n.x; v.nor[1] = n.y; v.nor[2] = n.z;
                        v.uv[0] = (float)(bd.side.u + kUvOff[i % 4][0]) * 0.0625f;
                        v.uv[1] = (float)(bd.side.v + kUvOff[i % 4][1]) * 0.0625f;
                        v.m = mat;
                        put(v);
                    }
                    const uint32_t q[12] = {0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7};
                    for (int k = 0; k < 12; ++k) puti(i1 + q[k]);
                    continue;
                }
                for (int d = 0; d < 6; ++d) {
                    int nx = x + kDir[d][0], ny = y + kDir[d][1], nz = z + kDir[d][2];
                    if (ny >= 0 && ny < 384) {
                        const uint8_t* nb = blocks;
                        if (nx < 0) { nb = neighbors[3]; nx += 16; }
                        else if (nx >= 16) { nb = neighbors[1]; nx -= 16; }
                        else if (nz < 0) { nb = neighbors[2]; nz += 16; }
                        else if (nz >= 16) { nb = neighbors[0]; nz -= 16; }
                        if (!nb) continue;
                        const uint8_t n = nb[ny + 384 * (nx + 16 * nz)];
                        const int nt = kBlockRender[n].trans;
                        bool show = false;
                        switch (bd.trans) {
                        case T_OPAQUE: case T_SEMI_TRANSPARENT: show = nt != T_OPAQUE; break;
                        case T_TRANSPARENT: show = n == (uint8_t)Block::AIR || nt == T_SEMI_TRANSPARENT; break;
                        }
                        if (!show) continue;
                    }
                    const uint32_t i1 = (uint32_t)nv;
                    const int which = kDir[d][1] == 1 ? 1 : (kDir[d][1] == -1 ? 2 : 0);      // 0 side, 1 top, 2 bottom
                    const SideUv su = which == 1 ? bd.top : (which == 2 ? bd.bottom : bd.side);
                    int uvStart = 0, uvFlip = -1;
                    if (bd.rot[which] || bd.flip[which]) {
                        Rng rng = makeSeededRandomEngine(x + worldBlockX, y, z + worldBlockZ, d);
                        // thrust::uniform_real_distribution<float>(0, 4): (u01 * (4 - 0)) + 0
                        if (bd.rot[which]) uvStart = (int)((rng.u01() * (4.f - 0.f)) + 0.f);
                        if (bd.flip[which]) uvFlip = (int)((rng.u01() * (4.f - 0.f)) + 0.f);
                    }
                    for (int j = 0; j < 4; ++j) {
                        Vertex v;
                        v.pos[0] = (float)(x + kDirVert[4 * d + j][0]); v.pos[1] = (float)(y + kDirVert[4 * d + j][1]); v.pos[2] = (float)(z + kDirVert[4 * d + j][2]);
                        v.nor[0] = (float)kDir[d][0]; v.nor[1] = (float)kDir[d][1]; v.nor[2] = (float)kDir[d][2];
                        int ou = kUvOff[(uvStart + j) % 4][0], ov = kUvOff[(uvStart + j) % 4][1];
                        if (uvFlip != -1) {
                            if (uvFlip & 1) ou = 1 - ou;
                            if (uvFlip & 2) ov = 1 - ov;
                        }
                        v.uv[0] = (float)(su.u + ou) * 0.0625f;
                        v.uv[1] = (float)(su.v + ov) * 0.0625f;
                        v.m = mat;
                        put(v);
                    }
                    const uint32_t q[6] = {0, 1, 2, 0, 2, 3};
                    for (int k = 0; k < 6; ++k) puti(i1 + q[k]);
                }
            }
    if (nIdxOut) *nIdxOut = ni;
    return nv;
}
