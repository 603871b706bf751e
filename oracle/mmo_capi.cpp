// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  Never linked into or called from the product (mega-minecraft_amd/).
#include "mmo_stages.h"
#include <algorithm>
#include <thread>
#include <atomic>
#include <map>
#include <functional>
#include <cstring>
#include <cstdio>

using namespace mmo;

namespace {

void parallel_for(int n, int nthreads, const std::function<void(int)>& fn)
{
    if (nthreads <= 1 || n <= 1) {
        for (int i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&]() {
            for (;;) {
                int i = next.fetch_add(1);
                if (i >= n) break;
                fn(i);
            }
        });
    for (auto& t : th) t.join();
}

inline int floordiv(int a, int b) { int q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------- math probes (batch)
void mmo_sinf(int n, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = mm_sinf(x[i]); }
void mmo_cosf(int n, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = mm_cosf(x[i]); }
void mmo_powf(int n, const float* x, const float* y, float* out) { for (int i = 0; i < n; ++i) out[i] = mm_powf(x[i], y[i]); }
void mmo_atan2f(int n, const float* y, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = mm_atan2f(y[i], x[i]); }
void mmo_acosf(int n, const float* x, float* out) { for (int i = 0; i < n; ++i) out[i] = mm_acosf(x[i]); }
void mmo_hash(int n, const uint32_t* x, uint32_t* out) { for (int i = 0; i < n; ++i) out[i] = hash_u32(x[i]); }
// k draws of u01 from makeSeededRandomEngine(x,y,z[,w]) per seed tuple; use_w selects the 4-arg form
void mmo_rng_u01(int n, const int* xyzw, int use_w, int k, float* out)
{
    for (int i = 0; i < n; ++i) {
        const int* s = xyzw + 4 * i;
        Rng r = use_w ? makeSeededRandomEngine(s[0], s[1], s[2], s[3]) : makeSeededRandomEngine(s[0], s[1], s[2]);
        for (int j = 0; j < k; ++j) out[i * k + j] = r.u01();
    }
}
// the bare engine: raw outputs and u01 draws for arbitrary 32-bit seeds (pinned against the real thrust engine, tests/golden/thrust_probe.npz)
void mmo_minstd(int n, const uint32_t* seeds, int k, uint32_t* raw, float* u01)
{
    for (int i = 0; i < n; ++i) {
        Rng a(seeds[i]), b(seeds[i]);
        for (int j = 0; j < k; ++j) { raw[i * k + j] = a.next(); u01[i * k + j] = b.u01(); }
    }
}
void mmo_simplex2(int n, const float* xy, float* out) { for (int i = 0; i < n; ++i) out[i] = simplex(vec2(xy[2 * i], xy[2 * i + 1])); }
void mmo_simplex3(int n, const float* xyz, float* out) { for (int i = 0; i < n; ++i) out[i] = simplex(vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])); }
void mmo_fbm2(int n, int octaves, const float* xy, float* out)
{
    for (int i = 0; i < n; ++i) {
        vec2 p(xy[2 * i], xy[2 * i + 1]);
        out[i] = octaves == 3 ? fbm<3>(p) : octaves == 4 ? fbm<4>(p) : fbm<5>(p);
    }
}
void mmo_fbm3(int n, int octaves, const float* xyz, float* out)
{
    for (int i = 0; i < n; ++i) {
        vec3 p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        out[i] = octaves == 3 ? fbm<3>(p) : octaves == 4 ? fbm<4>(p) : fbm<5>(p);
    }
}
// out[i] = {dist, color.xyz, edgeDist}
void mmo_worley2(int n, const float* xy, float* out5)
{
    for (int i = 0; i < n; ++i) {
        vec3 c; float e;
        float d = worley(vec2(xy[2 * i], xy[2 * i + 1]), &c, &e);
        float* o = out5 + 5 * i; o[0] = d; o[1] = c.x; o[2] = c.y; o[3] = c.z; o[4] = e;
    }
}
void mmo_worley3(int n, const float* xyz, float* out5)
{
    for (int i = 0; i < n; ++i) {
        vec3 c; float e;
        float d = worley(vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]), &c, &e);
        float* o = out5 + 5 * i; o[0] = d; o[1] = c.x; o[2] = c.y; o[3] = c.z; o[4] = e;
    }
}
void mmo_special_cave_noise(int n, const float* xyz, float* out)
{
    for (int i = 0; i < n; ++i) out[i] = specialCaveNoise(vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}
void mmo_rand3from3(int n, const float* xyz, float* out3)
{
    for (int i = 0; i < n; ++i) {
        vec3 r = rand3From3(vec3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        out3[3 * i] = r.x; out3[3 * i + 1] = r.y; out3[3 * i + 2] = r.z;
    }
}
// per-biome height (DEBUG_BIOME_OVERRIDE-style probe)
void mmo_biome_height(int n, int biome, const float* xy, float* out) { for (int i = 0; i < n; ++i) out[i] = getHeight((Biome)biome, vec2(xy[2 * i], xy[2 * i + 1])); }
void mmo_cave_biome(int n, const int* xyz, const float* maxHeight, int seed, uint8_t* out)
{
    for (int i = 0; i < n; ++i) out[i] = (uint8_t)getCaveBiome(ivec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, maxHeight[i], seed);
}
void mmo_should_generate_cave(int n, const int* xyz, const float* maxHeight, const float* obw, uint8_t* out)
{
    for (int i = 0; i < n; ++i) out[i] = shouldGenerateCaveAtBlock(ivec3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}, maxHeight[i], obw[i]) ? 1 : 0;
}

// ---------------------------------------------------------------- constant tables (fixture 3 of SURVEY §8c)
// layout: [24*6 biome rules u8][5*4 cave rules u8][24 grass u8][20 * (block u8 as f32, thickness, amp/tan, scale/maxSlope)] ...
void mmo_tables_material_infos(float* out80)
{
    for (int i = 0; i < numMaterials; ++i) {
        const auto& m = T().materialInfos[i];
        out80[4 * i] = (float)(int)m.block; out80[4 * i + 1] = m.thickness;
        out80[4 * i + 2] = m.noiseAmplitudeOrTanAngleOfRepose; out80[4 * i + 3] = m.noiseScaleOrMaxSlope;
    }
}
void mmo_tables_biome_material_weights(float* out480) { std::memcpy(out480, T().biomeMaterialWeights, sizeof(float) * numBiomes * numMaterials); }
// The gen tables in the numeric layout of tools/extract_ref_tables.py (tests/golden/ref_tables.npz), so that a test can hold them to
// the literals of the reference's BiomeUtils::init.  Material infos: column 2 of the eroded materials is the TANGENT here (the reference
// converts at the end of init); the test compares it with tan(radians(degrees)) of the extracted literal.
void mmo_tables_gens(int* featureBounds /*[21][2]*/, int* caveFeatureBounds /*[10][2]*/, float* surfGens /*[24][4][11]*/, float* caveGens /*[5][3][9]*/,
                     float* decoGens /*[24][7][10]*/, float* caveDecoGens /*[5][6][10]*/)
{
    const Tables& t = T();
    for (int f = 0; f < numFeatures; ++f) { featureBounds[2 * f] = t.featureHeightBounds[f].x; featureBounds[2 * f + 1] = t.featureHeightBounds[f].y; }
    for (int f = 0; f < numCaveFeatures; ++f) { caveFeatureBounds[2 * f] = t.caveFeatureHeightBounds[f].x; caveFeatureBounds[2 * f + 1] = t.caveFeatureHeightBounds[f].y; }
    std::memset(surfGens, 0, sizeof(float) * numBiomes * 4 * 11);
    for (int b = 0; b < numBiomes; ++b)
        for (size_t k = 0; k < t.biomeFeatureGens[b].size(); ++k) {
            const FeatureGen& g = t.biomeFeatureGens[b][k];
            float* o = surfGens + (b * 4 + k) * 11;
            o[0] = 1; o[1] = (float)(int)g.feature; o[2] = (float)g.gridCellSize; o[3] = (float)g.gridCellPadding; o[4] = g.chancePerGridCell;
            o[5] = g.canReplaceBlocks ? 1.f : 0.f; o[6] = (float)g.possibleTopLayers.size();
            for (size_t j = 0; j < g.possibleTopLayers.size() && j < 2; ++j) { o[7 + 2 * j] = (float)(int)g.possibleTopLayers[j].material; o[8 + 2 * j] = g.possibleTopLayers[j].minThickness; }
        }
    std::memset(caveGens, 0, sizeof(float) * numCaveBiomes * 3 * 9);
    for (int b = 0; b < numCaveBiomes; ++b)
        for (size_t k = 0; k < t.caveBiomeFeatureGens[b].size(); ++k) {
            const CaveFeatureGen& g = t.caveBiomeFeatureGens[b][k];
            float* o = caveGens + (b * 3 + k) * 9;
            o[0] = 1; o[1] = (float)(int)g.caveFeature; o[2] = (float)g.gridCellSize; o[3] = (float)g.gridCellPadding; o[4] = g.chancePerGridCell;
            o[5] = (float)g.minLayerHeight; o[6] = g.canReplaceBlocks ? 1.f : 0.f; o[7] = g.generatesFromCeiling ? 1.f : 0.f; o[8] = g.canGenerateInLava ? 1.f : 0.f;
        }
    auto deco = [](const std::vector<DecoratorGen>& gens, float* base) {
        for (size_t k = 0; k < gens.size(); ++k) {
            const DecoratorGen& g = gens[k];
            float* o = base + k * 10;
            std::vector<int> under;
            for (Block u : g.possibleUnderBlocks.v) under.push_back((int)u);
            std::sort(under.begin(), under.end());
            o[0] = 1; o[1] = (float)(int)g.decoratorBlock; o[2] = g.chance; o[3] = (float)under.size();
            for (size_t j = 0; j < under.size() && j < 3; ++j) o[4 + j] = (float)under[j];
            o[7] = (float)(int)g.possibleReplaceBlocks.v[0]; o[8] = (float)(int)g.secondDecoratorBlock; o[9] = g.generatesFromCeiling ? 1.f : 0.f;
        }
    };
    std::memset(decoGens, 0, sizeof(float) * numBiomes * 7 * 10);
    for (int b = 0; b < numBiomes; ++b) deco(t.biomeDecoratorGens[b], decoGens + b * 70);
    std::memset(caveDecoGens, 0, sizeof(float) * numCaveBiomes * 6 * 10);
    for (int b = 0; b < numCaveBiomes; ++b) deco(t.caveBiomeDecoratorGens[b], caveDecoGens + b * 60);
}
void mmo_tables_rules(uint8_t* biome144, uint8_t* cave20, uint8_t* grass24)
{
    for (int b = 0; b < numBiomes; ++b) {
        const auto& w = T().biomeNoiseWeights[b];
        uint8_t* o = biome144 + 6 * b; o[0] = w.ocean; o[1] = w.beach; o[2] = w.rocky; o[3] = w.magic; o[4] = w.temperature; o[5] = w.moisture;
        grass24[b] = (uint8_t)T().biomeBlocks[b].grassBlock;
    }
    for (int b = 0; b < numCaveBiomes; ++b) {
        const auto& w = T().caveBiomeNoiseWeights[b];
        uint8_t* o = cave20 + 4 * b; o[0] = w.none; o[1] = w.shallow; o[2] = w.warped; o[3] = w.rocky;
    }
}

// ---------------------------------------------------------------- per-stage entry points (reference per-chunk layouts)
void mmo_heightfields(int n, const int* posXZ, float* hf, float* bw, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) { generateHeightfield(ivec2{posXZ[2 * i], posXZ[2 * i + 1]}, hf + 256 * i, bw + 6144 * i); });
}
void mmo_gather_heightfields(int n, const int* posXZ, const float* hf, float* gathered, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) { gatherHeightfield(ivec2{posXZ[2 * i], posXZ[2 * i + 1]}, hf + 256 * i, gathered + 324 * i); });
}
void mmo_layers(int n, const int* posXZ, const float* gathered, const float* bw, float* layers, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) { generateLayers(ivec2{posXZ[2 * i], posXZ[2 * i + 1]}, gathered + 324 * i, bw + 6144 * i, layers + 5120 * i); });
}
int mmo_erode_zone_planes(float* gathered9) { return erodeZonePlanes(gathered9); }
void mmo_fix_backward_layers(int n, float* layers) { for (int i = 0; i < n; ++i) fixBackwardStratifiedLayers(layers + 5120 * i); }
void mmo_caves(int n, const int* posXZ, const float* hf, const float* bw, void* caveLayers, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) {
        generateCaves(ivec2{posXZ[2 * i], posXZ[2 * i + 1]}, hf + 256 * i, bw + 6144 * i, (CaveLayer*)caveLayers + 8192 * i);
    });
}
// writes up to maxPer entries per chunk into fp/cfp (chunk-major, stride maxPer) and the true counts into counts[2*i], counts[2*i+1]
void mmo_feature_placements(int n, const int* posXZ, const float* hf, const float* bw, const float* layers, const void* caveLayers,
                            void* fp, void* cfp, int maxPer, int* counts, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) {
        std::vector<FeaturePlacement> a; std::vector<CaveFeaturePlacement> b;
        generateFeaturePlacements(ivec2{posXZ[2 * i], posXZ[2 * i + 1]}, hf + 256 * i, bw + 6144 * i, layers + 5120 * i,
                                  (const CaveLayer*)caveLayers + 8192 * i, a, b);
        counts[2 * i] = (int)a.size(); counts[2 * i + 1] = (int)b.size();
        if (!a.empty()) std::memcpy((FeaturePlacement*)fp + (size_t)maxPer * i, a.data(), sizeof(FeaturePlacement) * std::min((int)a.size(), maxPer));
        if (!b.empty()) std::memcpy((CaveFeaturePlacement*)cfp + (size_t)maxPer * i, b.data(), sizeof(CaveFeaturePlacement) * std::min((int)b.size(), maxPer));
    });
}
// fill one batch; features for chunk i are fp[fpOff[i] .. fpOff[i+1]) (gathered lists, un-truncated)
void mmo_fill(int n, const int* posXZ, const float* hf, const float* bw, const float* layers, const void* caveLayers,
              const void* fp, const int* fpOff, const void* cfp, const int* cfpOff, uint8_t* blocks, int decorators, int nthreads)
{
    parallel_for(n, nthreads, [&](int i) {
        ivec3 wp = {posXZ[2 * i], 0, posXZ[2 * i + 1]};
        const FeaturePlacement* f = fp ? (const FeaturePlacement*)fp + fpOff[i] : nullptr;
        const CaveFeaturePlacement* c = cfp ? (const CaveFeaturePlacement*)cfp + cfpOff[i] : nullptr;
        int nf = fp ? fpOff[i + 1] - fpOff[i] : 0, nc = cfp ? cfpOff[i + 1] - cfpOff[i] : 0;
        Block* b = (Block*)blocks + (size_t)98304 * i;
        fillChunk(wp, hf + 256 * i, bw + 6144 * i, layers + 5120 * i, (const CaveLayer*)caveLayers + 8192 * i, f, nf, c, nc, b);
        if (decorators) placeDecorators(wp, hf + 256 * i, bw + 6144 * i, (const CaveLayer*)caveLayers + 8192 * i, b);
    });
}
void mmo_decorators(int n, const int* posXZ, const float* hf, const float* bw, const void* caveLayers, uint8_t* blocks)
{
    for (int i = 0; i < n; ++i)
        placeDecorators(ivec3{posXZ[2 * i], 0, posXZ[2 * i + 1]}, hf + 256 * i, bw + 6144 * i, (const CaveLayer*)caveLayers + 8192 * i,
                        (Block*)blocks + (size_t)98304 * i);
}
// single-voxel feature probes (fixture 11 of SURVEY §8c): rasterise one placement into a box
void mmo_place_feature_box(int feature, const int* fpos, int canReplace, const int* boxMin, const int* boxSize, uint8_t* out)
{
    FeaturePlacement p; std::memset((void*)&p, 0, sizeof(p));
    p.feature = (Feature)feature; p.pos = ivec3{fpos[0], fpos[1], fpos[2]}; p.canReplaceBlocks = canReplace != 0;
    size_t k = 0;
    for (int z = 0; z < boxSize[2]; ++z) for (int x = 0; x < boxSize[0]; ++x) for (int y = 0; y < boxSize[1]; ++y, ++k) {
        Block b = Block::AIR;
        bool placed = placeFeature(p, ivec3{boxMin[0] + x, boxMin[1] + y, boxMin[2] + z}, &b);
        out[k] = placed ? (uint8_t)b : 255;
    }
}
void mmo_place_cave_feature_box(int feature, const int* fpos, int layerHeight, int canReplace, const int* boxMin, const int* boxSize, uint8_t* out)
{
    CaveFeaturePlacement p; std::memset((void*)&p, 0, sizeof(p));
    p.feature = (CaveFeature)feature; p.pos = ivec3{fpos[0], fpos[1], fpos[2]}; p.layerHeight = layerHeight; p.canReplaceBlocks = canReplace != 0;
    size_t k = 0;
    for (int z = 0; z < boxSize[2]; ++z) for (int x = 0; x < boxSize[0]; ++x) for (int y = 0; y < boxSize[1]; ++y, ++k) {
        Block b = Block::AIR;
        bool placed = placeCaveFeature(p, ivec3{boxMin[0] + x, boxMin[1] + y, boxMin[2] + z}, &b);
        out[k] = placed ? (uint8_t)b : 255;
    }
}

void mmo_ub_counters(long long* out3, int reset)
{
    out3[0] = g_ub.noLayerFound; out3[1] = g_ub.caveLayerOverflow; out3[2] = g_ub.decoratorOutOfRange;
    if (reset) g_ub = UbCounters{0, 0, 0};
}

// ---------------------------------------------------------------- region pipeline (the canonical "world" definition)
// Generates chunks [cx0, cx0+nx) x [cz0, cz0+nz) (chunk coordinates), chunk-major output in z-major order (i = cx + nx*cz).
// flags: bit0 erosion, bit1 features, bit2 decorators.  Without erosion only E3's fix-up runs (DEBUG_SKIP_EROSION semantics,
// chunk.cu:713-720).
// Canonical region semantics: a zone's 6-chunk erosion padding always uses RAW (pre-erosion) layers; feature
// placements of the 3-chunk ring around the region are generated from that ring's own eroded layers.
//
// Two phases so that the multi-process tiling tests can exchange ring placements between the phases:
//   mmo_region_begin   heightfield .. placements on the ring-extended grid P (ring cells with local_mask == 0 are skipped)
//   mmo_region_finish  gather + fill + decorators on the region, from the (possibly externally completed) placement arrays
struct RegionCtx {
    int cx0, cz0, nx, nz, flags, ring, px0, pz0, pnx, pnz, np;
    std::vector<int> ppos;
    std::vector<float> hf, bw, layers;
    std::vector<CaveLayer> cave;
};

void* mmo_region_create() { return new RegionCtx(); }
void mmo_region_destroy(void* c) { delete (RegionCtx*)c; }

// fp/cfp/counts: flat placement arrays over the P grid, capacities fpCap / cfpCap entries per cell (20 / 24 byte records)
void mmo_region_begin(void* ctx, int cx0, int cz0, int nx, int nz, int flags, const uint8_t* local_mask, void* fp_out, void* cfp_out,
                      int* counts_out, int fpCap, int cfpCap, int nthreads)
{
    RegionCtx& r = *(RegionCtx*)ctx;
    const bool doErosion = flags & 1, doFeatures = flags & 2;
    r.cx0 = cx0; r.cz0 = cz0; r.nx = nx; r.nz = nz; r.flags = flags;
    r.ring = doFeatures ? 3 : 0;
    const int ring = r.ring;
    const int px0 = r.px0 = cx0 - ring, pz0 = r.pz0 = cz0 - ring, pnx = r.pnx = nx + 2 * ring, pnz = r.pnz = nz + 2 * ring;
    const int np = r.np = pnx * pnz;

    std::vector<int>& ppos = r.ppos;
    ppos.assign(2 * np, 0);
    for (int z = 0; z < pnz; ++z) for (int x = 0; x < pnx; ++x) { ppos[2 * (x + pnx * z)] = (px0 + x) * 16; ppos[2 * (x + pnx * z) + 1] = (pz0 + z) * 16; }
    std::vector<float>&hf = r.hf, &bw = r.bw, &layers = r.layers;
    hf.assign((size_t)256 * np, 0.f); bw.assign((size_t)6144 * np, 0.f); layers.assign((size_t)5120 * np, 0.f);
    r.cave.assign((size_t)8192 * np, CaveLayer{});

    if (!doErosion) {
        std::vector<float> gathered((size_t)324 * np);
        mmo_heightfields(np, ppos.data(), hf.data(), bw.data(), nthreads);
        mmo_gather_heightfields(np, ppos.data(), hf.data(), gathered.data(), nthreads);
        mmo_layers(np, ppos.data(), gathered.data(), bw.data(), layers.data(), nthreads);
    } else {
        // raw area A = union of the 24x24 gathered areas of all zones intersecting P
        const int zx0 = floordiv(px0, ZONE_SIZE) * ZONE_SIZE, zz0 = floordiv(pz0, ZONE_SIZE) * ZONE_SIZE;
        const int zx1 = floordiv(px0 + pnx - 1, ZONE_SIZE) * ZONE_SIZE, zz1 = floordiv(pz0 + pnz - 1, ZONE_SIZE) * ZONE_SIZE;
        const int ax0 = zx0 - 6, az0 = zz0 - 6, anx = (zx1 - zx0) + 24, anz = (zz1 - zz0) + 24;
        const int na = anx * anz;
        std::vector<int> apos(2 * na);
        for (int z = 0; z < anz; ++z) for (int x = 0; x < anx; ++x) { apos[2 * (x + anx * z)] = (ax0 + x) * 16; apos[2 * (x + anx * z) + 1] = (az0 + z) * 16; }
        std::vector<float> ahf((size_t)256 * na), abw((size_t)6144 * na), alayers((size_t)5120 * na), agath((size_t)324 * na);
        mmo_heightfields(na, apos.data(), ahf.data(), abw.data(), nthreads);
        mmo_gather_heightfields(na, apos.data(), ahf.data(), agath.data(), nthreads);
        mmo_layers(na, apos.data(), agath.data(), abw.data(), alayers.data(), nthreads);
        agath.clear(); agath.shrink_to_fit();

        std::vector<float> eroded(alayers);      // eroded copy; raw stays in alayers for the other zones' padding
        std::vector<ivec2> zones;
        for (int zz = zz0; zz <= zz1; zz += ZONE_SIZE) for (int zx = zx0; zx <= zx1; zx += ZONE_SIZE) zones.push_back(ivec2{zx, zz});
        parallel_for((int)zones.size(), nthreads, [&](int zi) {
            const ivec2 zc = zones[zi];
            std::vector<float> g((size_t)9 * EROSION_GRID_NUM_COLS);
            // E1 copyLayers(to) chunk.cu:603-656: the zone's 24 x 24 gathered chunks, RAW layers + heightfields
            std::vector<int> gatheredIdx(4 * ZONE_SIZE * ZONE_SIZE), ownIdx(ZONE_SIZE * ZONE_SIZE);
            for (int cz = 0; cz < 24; ++cz) for (int cx = 0; cx < 24; ++cx) gatheredIdx[cx + 24 * cz] = (zc.x - 6 + cx - ax0) + anx * (zc.y - 6 + cz - az0);
            for (int cz = 0; cz < 12; ++cz) for (int cx = 0; cx < 12; ++cx) ownIdx[cx + 12 * cz] = (zc.x + cx - ax0) + anx * (zc.y + cz - az0);
            zoneCopyLayers(alayers.data(), ahf.data(), gatheredIdx.data(), g.data(), true);
            erodeZonePlanes(g.data());
            // E3 copyLayers(from): the zone's own 12 x 12 chunks, 8 eroded planes
            zoneCopyLayers(eroded.data(), nullptr, ownIdx.data(), g.data(), false);
        });
        for (int z = 0; z < pnz; ++z) for (int x = 0; x < pnx; ++x) {
            const int pi = x + pnx * z, ai = (px0 + x - ax0) + anx * (pz0 + z - az0);
            std::memcpy(hf.data() + (size_t)256 * pi, ahf.data() + (size_t)256 * ai, 256 * sizeof(float));
            std::memcpy(bw.data() + (size_t)6144 * pi, abw.data() + (size_t)6144 * ai, 6144 * sizeof(float));
            std::memcpy(layers.data() + (size_t)5120 * pi, eroded.data() + (size_t)5120 * ai, 5120 * sizeof(float));
        }
    }
    mmo_fix_backward_layers(np, layers.data());

    // cells to compute locally: the region itself + ring cells selected by the mask (null = all)
    std::vector<int> compute;
    for (int i = 0; i < np; ++i) {
        const int x = i % pnx - ring, z = i / pnx - ring;
        const bool inR = x >= 0 && x < nx && z >= 0 && z < nz;
        if (inR || !local_mask || local_mask[i]) compute.push_back(i);
    }
    parallel_for((int)compute.size(), nthreads, [&](int k) {
        const int i = compute[k];
        generateCaves(ivec2{ppos[2 * i], ppos[2 * i + 1]}, hf.data() + (size_t)256 * i, bw.data() + (size_t)6144 * i, r.cave.data() + (size_t)8192 * i);
    });
    if (doFeatures && counts_out) {
        std::memset(counts_out, 0, sizeof(int) * 2 * np);
        parallel_for((int)compute.size(), nthreads, [&](int k) {
            const int i = compute[k];
            std::vector<FeaturePlacement> a; std::vector<CaveFeaturePlacement> b;
            generateFeaturePlacements(ivec2{ppos[2 * i], ppos[2 * i + 1]}, hf.data() + (size_t)256 * i, bw.data() + (size_t)6144 * i,
                                      layers.data() + (size_t)5120 * i, r.cave.data() + (size_t)8192 * i, a, b);
            counts_out[2 * i] = (int)a.size(); counts_out[2 * i + 1] = (int)b.size();
            // (an empty vector's data() may be null: memcpy's arguments must not be, even for 0 bytes - found by the UBSan job)
            if (!a.empty()) std::memcpy((FeaturePlacement*)fp_out + (size_t)fpCap * i, a.data(), sizeof(FeaturePlacement) * std::min((int)a.size(), fpCap));
            if (!b.empty()) std::memcpy((CaveFeaturePlacement*)cfp_out + (size_t)cfpCap * i, b.data(), sizeof(CaveFeaturePlacement) * std::min((int)b.size(), cfpCap));
        });
    }
}

void mmo_region_finish(void* ctx, const void* fp_in, const void* cfp_in, const int* counts_in, int fpCap, int cfpCap, uint8_t* out_blocks,
                       float* out_hf, float* out_layers, void* out_cave, int nthreads)
{
    RegionCtx& r = *(RegionCtx*)ctx;
    const bool doFeatures = r.flags & 2, doDecor = r.flags & 4;
    const int nx = r.nx, nz = r.nz, ring = r.ring, pnx = r.pnx;
    parallel_for(nx * nz, nthreads, [&](int i) {
        const int x = i % nx, z = i / nx;
        const int pi = (x + ring) + pnx * (z + ring);
        std::vector<FeaturePlacement> gf; std::vector<CaveFeaturePlacement> gc;
        if (doFeatures)
            for (const ivec2& off : gatherFeaturePlacementsChunkOffsets) {
                const int ni = (x + ring + off.x) + pnx * (z + ring + off.y);
                const FeaturePlacement* a = (const FeaturePlacement*)fp_in + (size_t)fpCap * ni;
                const CaveFeaturePlacement* b = (const CaveFeaturePlacement*)cfp_in + (size_t)cfpCap * ni;
                gf.insert(gf.end(), a, a + std::min(counts_in[2 * ni], fpCap));
                gc.insert(gc.end(), b, b + std::min(counts_in[2 * ni + 1], cfpCap));
            }
        Block* b = (Block*)out_blocks + (size_t)98304 * i;
        ivec3 wp = {r.ppos[2 * pi], 0, r.ppos[2 * pi + 1]};
        fillChunk(wp, r.hf.data() + (size_t)256 * pi, r.bw.data() + (size_t)6144 * pi, r.layers.data() + (size_t)5120 * pi, r.cave.data() + (size_t)8192 * pi,
                  gf.data(), (int)gf.size(), gc.data(), (int)gc.size(), b);
        if (doDecor) placeDecorators(wp, r.hf.data() + (size_t)256 * pi, r.bw.data() + (size_t)6144 * pi, r.cave.data() + (size_t)8192 * pi, b);
        if (out_hf) std::memcpy(out_hf + (size_t)256 * i, r.hf.data() + (size_t)256 * pi, 256 * sizeof(float));
        if (out_layers) std::memcpy(out_layers + (size_t)5120 * i, r.layers.data() + (size_t)5120 * pi, 5120 * sizeof(float));
        if (out_cave) std::memcpy((CaveLayer*)out_cave + (size_t)8192 * i, r.cave.data() + (size_t)8192 * pi, 8192 * sizeof(CaveLayer));
    });
}

void mmo_generate_region(int cx0, int cz0, int nx, int nz, int flags, uint8_t* out_blocks, float* out_hf, float* out_layers,
                         void* out_cave, int nthreads, double* stage_seconds /*unused*/)
{
    (void)stage_seconds;
    RegionCtx r;
    const int ring = (flags & 2) ? 3 : 0;
    const size_t np = (size_t)(nx + 2 * ring) * (nz + 2 * ring);
    const int fpCap = 256, cfpCap = 4096;
    std::vector<FeaturePlacement> fp(flags & 2 ? np * fpCap : 0);
    std::vector<CaveFeaturePlacement> cfp(flags & 2 ? np * cfpCap : 0);
    std::vector<int> counts(2 * np, 0);
    mmo_region_begin(&r, cx0, cz0, nx, nz, flags, nullptr, fp.data(), cfp.data(), counts.data(), fpCap, cfpCap, nthreads);
    mmo_region_finish(&r, fp.data(), cfp.data(), counts.data(), fpCap, cfpCap, out_blocks, out_hf, out_layers, out_cave, nthreads);
}

}  // extern "C"
