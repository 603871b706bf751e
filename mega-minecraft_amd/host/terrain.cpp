// mmgen host side — the `Terrain` scheduler mirror.  Behavioural spec: src/terrain/terrain.cpp (budget + staging sizes :65-185,
// spiral :220-252, zones :259-298, updateChunk :301-420, erosion readiness :430-522, updateZones :524-567, tick :587-960).
#include "terrain.hpp"
#include <cmath>
#include <cstdlib>
#include <algorithm>

namespace mmhost {

namespace {

constexpr int imin(int a, int b) { return a < b ? a : b; }

// staging slot counts (terrain.cpp:111-129)
constexpr int numBlocksSlots = Terrain::maxActionTimePerFrame / Terrain::actionTimeFill;
constexpr int numHeightfieldSlots = Terrain::maxActionTimePerFrame / imin(imin(Terrain::actionTimeGenerateHeightfield, Terrain::actionTimeGenerateLayers),
                                                                          imin(Terrain::actionTimeGenerateCaves, Terrain::actionTimeFill));
constexpr int numPositionSlots = Terrain::maxActionTimePerFrame / imin(Terrain::actionTimeGenerateHeightfield, imin(Terrain::actionTimeGenerateLayers, Terrain::actionTimeGenerateCaves));
constexpr int numLayersSlots = Terrain::maxActionTimePerFrame / imin(Terrain::actionTimeGenerateLayers, Terrain::actionTimeFill);
constexpr int numCaveLayersSlots = Terrain::maxActionTimePerFrame / imin(Terrain::actionTimeGenerateCaves, Terrain::actionTimeFill);
constexpr int numGatheredLayersSlots = Terrain::maxActionTimePerFrame / Terrain::actionTimeErodeZone;

// 8-neighbour order N, NE, E, SE, S, SW, W, NW (util/enums.hpp:29-38) and the 4-neighbour order N, E, S, W (:40-47)
const ivec2 kDir8[8] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}};
const ivec2 kDir4[4] = {{0, 1}, {1, 0}, {0, -1}, {-1, 0}};

int floorDiv(int a, int b) { return (int)std::floor((float)a / (float)b); }
ivec2 zonePosFromChunkPos(ivec2 c) { return {floorDiv(c.x, ZONE_SIZE) * ZONE_SIZE, floorDiv(c.y, ZONE_SIZE) * ZONE_SIZE}; }

template <class T> T* pinned(size_t n) { void* p = nullptr; HipUtils::checkError("hipHostMalloc failed", (int)hipHostMalloc(&p, n * sizeof(T))); return (T*)p; }
template <class T> T* device(size_t n) { void* p = nullptr; HipUtils::checkError("hipMalloc failed", (int)hipMalloc(&p, n * sizeof(T))); return (T*)p; }

}  // namespace

Terrain::Terrain() { generateSpiral(); }
Terrain::~Terrain() { freeHip(); }
void Terrain::init() { initHip(); }

void Terrain::initHip()
{
    host_blocks = pinned<Block>((size_t)numBlocksSlots * devBlocksSize);
    dev_blocks = device<Block>((size_t)numBlocksSlots * devBlocksSize);
    dev_featurePlacements = device<FeaturePlacement>((size_t)numBlocksSlots * devFeaturePlacementsSize);
    dev_caveFeaturePlacements = device<CaveFeaturePlacement>((size_t)numBlocksSlots * devCaveFeaturePlacementsSize);
    host_heightfields = pinned<float>((size_t)numHeightfieldSlots * devHeightfieldSize);
    dev_heightfields = device<float>((size_t)numHeightfieldSlots * devHeightfieldSize);
    host_biomeWeights = pinned<float>((size_t)numHeightfieldSlots * devBiomeWeightsSize);
    dev_biomeWeights = device<float>((size_t)numHeightfieldSlots * devBiomeWeightsSize);
    host_chunkWorldBlockPositions = pinned<ivec2>(numPositionSlots);
    dev_chunkWorldBlockPositions = device<ivec2>(numPositionSlots);
    host_layers = pinned<float>((size_t)numLayersSlots * devLayersSize);
    dev_layers = device<float>((size_t)numLayersSlots * devLayersSize);
    host_caveLayers = pinned<CaveLayer>((size_t)numCaveLayersSlots * devCaveLayersSize);
    dev_caveLayers = device<CaveLayer>((size_t)numCaveLayersSlots * devCaveLayersSize);
    host_gatheredLayers = pinned<float>((size_t)numGatheredLayersSlots * devGatheredLayersSize);
    dev_gatheredLayers = device<float>((size_t)numGatheredLayersSlots * devGatheredLayersSize);
    dev_accumulatedHeights = device<float>((size_t)numGatheredLayersSlots * devAccumulatedHeightsSize);
    for (auto& s : streams) HipUtils::checkError("hipStreamCreate failed", (int)hipStreamCreate(&s));
}

void Terrain::freeHip()
{
    if (!dev_blocks) return;
    void* hostPtrs[] = {host_blocks, host_heightfields, host_biomeWeights, host_chunkWorldBlockPositions, host_layers, host_caveLayers, host_gatheredLayers};
    void* devPtrs[] = {dev_blocks, dev_featurePlacements, dev_caveFeaturePlacements, dev_heightfields, dev_biomeWeights, dev_chunkWorldBlockPositions,
                       dev_layers, dev_caveLayers, dev_gatheredLayers, dev_accumulatedHeights};
    for (void* p : hostPtrs) (void)hipHostFree(p);
    for (void* p : devPtrs) (void)hipFree(p);
    for (auto& s : streams) (void)hipStreamDestroy(s);
    dev_blocks = nullptr;
}

// square spiral outwards from the player, radius chunkMaxGenRadius (terrain.cpp:220-252)
void Terrain::generateSpiral()
{
    int x = 0, z = 0, step = 1, side = 1;
    for (;;) {
        for (; 2 * x * step < side; x += step) spiral.push_back({x, z});
        if (side > chunkMaxGenRadius * 2) return;
        for (; 2 * z * step < side; z += step) spiral.push_back({x, z});
        step = -step;
        ++side;
    }
}

Zone* Terrain::createZone(ivec2 pos)
{
    auto owned = std::make_unique<Zone>(pos);
    Zone* z = owned.get();
    zones[{pos.x, pos.y}] = std::move(owned);
    for (int i = 0; i < 8; ++i) {
        auto it = zones.find({pos.x + ZONE_SIZE * kDir8[i].x, pos.y + ZONE_SIZE * kDir8[i].y});
        if (it == zones.end()) continue;
        z->neighbors[i] = it->second.get();
        it->second->neighbors[(i + 4) % 8] = z;
    }
    return z;
}

void Terrain::updateChunk(int dx, int dz)
{
    const ivec2 cpos = {currentChunkPos.x + dx, currentChunkPos.y + dz};
    const ivec2 zpos = zonePosFromChunkPos(cpos);
    Zone* zone;
    if (lastUpdateZonePtr && lastUpdateZonePtr->worldChunkPos == zpos) zone = lastUpdateZonePtr;
    else {
        auto it = zones.find({zpos.x, zpos.y});
        zone = it == zones.end() ? createZone(zpos) : it->second.get();
        lastUpdateZonePtr = zone;
    }
    const ivec2 local = cpos - zpos;
    auto& slot = zone->chunks[local.x + ZONE_SIZE * local.y];
    if (!slot) {
        auto c = std::make_unique<Chunk>(cpos);
        c->zonePtr = zone;
        for (int i = 0; i < 4; ++i) {
            const ivec2 nl = local + kDir4[i];
            Zone* nz = zone;
            if (nl.x < 0 || nl.x >= ZONE_SIZE || nl.y < 0 || nl.y >= ZONE_SIZE) {
                nz = zone->neighbors[i * 2];
                if (!nz) continue;
            }
            auto& n = nz->chunks[((nl.x + ZONE_SIZE) % ZONE_SIZE) + ZONE_SIZE * ((nl.y + ZONE_SIZE) % ZONE_SIZE)];
            if (!n) continue;
            c->neighbors[i] = n.get();
            n->neighbors[(i + 2) % 4] = c.get();
        }
        slot = std::move(c);
    }
    Chunk* c = slot.get();
    if (!c->isReadyForQueue()) return;

    std::queue<Chunk*>* q = nullptr;
    switch (c->getState()) {
    case ChunkState::EMPTY: q = &chunksToGenerateHeightfield; break;
    case ChunkState::HAS_HEIGHTFIELD: q = &chunksToGatherHeightfield; break;
    case ChunkState::NEEDS_LAYERS: q = &chunksToGenerateLayers; break;
    case ChunkState::NEEDS_CAVES: q = &chunksToGenerateCaves; break;
    case ChunkState::NEEDS_FEATURE_PLACEMENTS: q = &chunksToGenerateFeaturePlacements; break;
    case ChunkState::NEEDS_GATHER_FEATURE_PLACEMENTS: q = &chunksToGatherFeaturePlacements; break;
    case ChunkState::READY_TO_FILL: q = &chunksToFill; break;
    case ChunkState::NEEDS_VBOS:
        if (std::max(std::abs(dx), std::abs(dz)) <= chunkVbosGenRadius) q = &chunksToCreateAndBufferVbos;
        break;
    default: break;
    }
    if (q) { c->setNotReadyForQueue(); q->push(c); }
}

void Terrain::updateChunks()
{
    for (const ivec2& d : spiral) updateChunk(d.x, d.y);
}

// a chunk that just got its layers may complete its own zone and the (up to 3) zones whose padding it lies in (terrain.cpp:430-454)
void Terrain::addZonesToTryErosionSet(Chunk* c)
{
    Zone* zone = c->zonePtr;
    zonesToTryErosion.insert(zone);
    const ivec2 local = c->worldChunkPos - zone->worldChunkPos;
    const int start = local.x < ZONE_SIZE / 2 ? (local.y < ZONE_SIZE / 2 ? 4 : 6) : (local.y < ZONE_SIZE / 2 ? 0 : 2);
    for (int i = 0; i < 3; ++i) {
        Zone* n = zone->neighbors[(start + i) % 8];
        if (n && !n->hasBeenQueuedForErosion) zonesToTryErosion.insert(n);
    }
}

// fills zone->gatheredChunks (24 x 24) and reports whether all of them have layers (terrain.cpp:456-522)
static bool isZoneReadyForErosion(Zone* zone)
{
    zone->gatheredChunks.assign(ZONE_SIZE * ZONE_SIZE * 4, nullptr);
    auto take = [&](Chunk* c) {
        if (!c || c->getState() < ChunkState::HAS_LAYERS) return false;
        const ivec2 g = c->worldChunkPos - zone->worldChunkPos + ivec2{ZONE_SIZE / 2, ZONE_SIZE / 2};
        zone->gatheredChunks[g.x + ZONE_SIZE * 2 * g.y] = c;
        return true;
    };
    for (auto& c : zone->chunks) if (!take(c.get())) return false;
    for (int i = 0; i < 8; ++i) {
        Zone* n = zone->neighbors[i];
        if (!n) return false;      // the reference skips a missing neighbour zone and later dereferences the hole; a zone without all 8 neighbours is simply not ready
        const int x0 = kDir8[i].x == -1 ? ZONE_SIZE / 2 : 0, x1 = kDir8[i].x == 1 ? ZONE_SIZE / 2 : ZONE_SIZE;
        const int z0 = kDir8[i].y == -1 ? ZONE_SIZE / 2 : 0, z1 = kDir8[i].y == 1 ? ZONE_SIZE / 2 : ZONE_SIZE;
        for (int z = z0; z < z1; ++z) for (int x = x0; x < x1; ++x) if (!take(n->chunks[x + ZONE_SIZE * z].get())) return false;
    }
    return true;
}

void Terrain::updateZones()
{
    for (Zone* z : zonesToTryErosion) {
        if (isZoneReadyForErosion(z)) { zonesToErode.push(z); z->hasBeenQueuedForErosion = true; }
        else z->gatheredChunks.clear();
    }
    zonesToTryErosion.clear();
}

static void promoteIfNeighboursFilled(Chunk* c)      // checkChunkAndNeighborsForNeedsVbos, terrain.cpp:569-585
{
    if (!c || c->getState() < ChunkState::FILLED) return;
    for (Chunk* n : c->neighbors) if (!n || n->getState() < ChunkState::FILLED) return;
    if (c->getState() == ChunkState::FILLED) c->setState(ChunkState::NEEDS_VBOS);
}

void Terrain::tick(float deltaTime)
{
    if (!(currentChunkPos == lastChunkPos)) { lastChunkPos = currentChunkPos; needsUpdateChunks = true; }
    if (needsUpdateChunks) { updateZones(); updateChunks(); needsUpdateChunks = false; }

    actionTimeLeft = std::min(actionTimeLeft + (int)(totalActionTimePerSecond * deltaTime), maxActionTimePerFrame);

    // running offsets into the shared staging buffers, exactly as terrain.cpp:623-641
    int blocksIdx = 0, heightfieldIdx = 0, biomeWeightsIdx = 0, positionIdx = 0, layersIdx = 0, caveLayersIdx = 0, gatheredIdx = 0, streamIdx = 0;
    auto takeBatch = [&](std::queue<Chunk*>& q, int cost, ChunkState next, bool notReady) {
        std::vector<Chunk*> batch;
        while (!q.empty() && actionTimeLeft >= cost) {
            needsUpdateChunks = true;
            Chunk* c = q.front(); q.pop();
            batch.push_back(c);
            c->setState(next);
            if (notReady) c->setNotReadyForQueue();
            actionTimeLeft -= cost;
        }
        return batch;
    };

    while (!chunksToCreateAndBufferVbos.empty() && actionTimeLeft >= actionTimeCreateAndBufferVbos) {
        needsUpdateChunks = true;
        Chunk* c = chunksToCreateAndBufferVbos.front(); chunksToCreateAndBufferVbos.pop();
        c->createVBOs();                                // terrain.cpp:650; bufferVBOs / buildChunkAccel are the renderer's
        drawableChunks.insert(c);
        c->setState(ChunkState::DRAWABLE);
        c->setNotReadyForQueue();
        actionTimeLeft -= actionTimeCreateAndBufferVbos;
    }

    {
        std::vector<Chunk*> chunks = takeBatch(chunksToFill, actionTimeFill, ChunkState::FILLED, true);
        const int n = (int)chunks.size();
        if (n > 0) {
            Chunk::fill(chunks, host_heightfields + (size_t)heightfieldIdx * devHeightfieldSize, dev_heightfields + (size_t)heightfieldIdx * devHeightfieldSize,
                        host_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize, dev_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize,
                        host_layers + (size_t)layersIdx * devLayersSize, dev_layers + (size_t)layersIdx * devLayersSize,
                        host_caveLayers + (size_t)caveLayersIdx * devCaveLayersSize, dev_caveLayers + (size_t)caveLayersIdx * devCaveLayersSize,
                        dev_featurePlacements + (size_t)blocksIdx * devFeaturePlacementsSize, dev_caveFeaturePlacements + (size_t)blocksIdx * devCaveFeaturePlacementsSize,
                        host_blocks + (size_t)blocksIdx * devBlocksSize, dev_blocks + (size_t)blocksIdx * devBlocksSize, streams[streamIdx]);
            blocksIdx += n; heightfieldIdx += n; biomeWeightsIdx += n; layersIdx += n; caveLayersIdx += n; ++streamIdx;
        }
        for (Chunk* c : chunks) { promoteIfNeighboursFilled(c); for (Chunk* nb : c->neighbors) promoteIfNeighboursFilled(nb); }
    }

    while (!chunksToGatherFeaturePlacements.empty() && actionTimeLeft >= actionTimeGatherFeaturePlacements) {
        needsUpdateChunks = true;
        Chunk* c = chunksToGatherFeaturePlacements.front(); chunksToGatherFeaturePlacements.pop();
        c->gatherFeaturePlacements();
        actionTimeLeft -= actionTimeGatherFeaturePlacements;
    }

    while (!chunksToGenerateFeaturePlacements.empty() && actionTimeLeft >= actionTimeGenerateFeaturePlacements) {
        needsUpdateChunks = true;
        Chunk* c = chunksToGenerateFeaturePlacements.front(); chunksToGenerateFeaturePlacements.pop();
        c->generateFeaturePlacements();
        c->setState(ChunkState::NEEDS_GATHER_FEATURE_PLACEMENTS);
        actionTimeLeft -= actionTimeGenerateFeaturePlacements;
    }

    {
        std::vector<Chunk*> chunks = takeBatch(chunksToGenerateCaves, actionTimeGenerateCaves, ChunkState::NEEDS_FEATURE_PLACEMENTS, false);
        const int n = (int)chunks.size();
        if (n > 0) {
            Chunk::generateCaves(chunks, host_heightfields + (size_t)heightfieldIdx * devHeightfieldSize, dev_heightfields + (size_t)heightfieldIdx * devHeightfieldSize,
                                 host_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize, dev_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize,
                                 host_chunkWorldBlockPositions + positionIdx, dev_chunkWorldBlockPositions + positionIdx,
                                 host_caveLayers + (size_t)caveLayersIdx * devCaveLayersSize, dev_caveLayers + (size_t)caveLayersIdx * devCaveLayersSize, streams[streamIdx]);
            heightfieldIdx += n; biomeWeightsIdx += n; positionIdx += n; caveLayersIdx += n; ++streamIdx;
        }
    }

    while (!zonesToErode.empty() && actionTimeLeft >= actionTimeErodeZone) {
        needsUpdateChunks = true;
        Zone* z = zonesToErode.front(); zonesToErode.pop();
        Chunk::erodeZone(z, host_gatheredLayers + (size_t)gatheredIdx * devGatheredLayersSize, dev_gatheredLayers + (size_t)gatheredIdx * devGatheredLayersSize,
                         dev_accumulatedHeights + (size_t)gatheredIdx * devAccumulatedHeightsSize, streams[streamIdx]);
        ++gatheredIdx; ++streamIdx;
        for (auto& c : z->chunks) c->setState(ChunkState::NEEDS_CAVES);
        actionTimeLeft -= actionTimeErodeZone;
    }

    {
        std::vector<Chunk*> chunks = takeBatch(chunksToGenerateLayers, actionTimeGenerateLayers, ChunkState::HAS_LAYERS, false);
        for (Chunk* c : chunks) addZonesToTryErosionSet(c);
        const int n = (int)chunks.size();
        if (n > 0) {
            Chunk::generateLayers(chunks, host_heightfields + (size_t)heightfieldIdx * devHeightfieldSize, dev_heightfields + (size_t)heightfieldIdx * devHeightfieldSize,
                                  host_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize, dev_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize,
                                  host_chunkWorldBlockPositions + positionIdx, dev_chunkWorldBlockPositions + positionIdx,
                                  host_layers + (size_t)layersIdx * devLayersSize, dev_layers + (size_t)layersIdx * devLayersSize, streams[streamIdx]);
            heightfieldIdx += n; biomeWeightsIdx += n; positionIdx += n; layersIdx += n; ++streamIdx;
        }
    }

    while (!chunksToGatherHeightfield.empty() && actionTimeLeft >= actionTimeGatherHeightfield) {
        needsUpdateChunks = true;
        Chunk* c = chunksToGatherHeightfield.front(); chunksToGatherHeightfield.pop();
        c->gatherHeightfield();
        actionTimeLeft -= actionTimeGatherHeightfield;
    }

    {
        std::vector<Chunk*> chunks = takeBatch(chunksToGenerateHeightfield, actionTimeGenerateHeightfield, ChunkState::HAS_HEIGHTFIELD, false);
        if (!chunks.empty()) {
            Chunk::generateHeightfields(chunks, host_chunkWorldBlockPositions + positionIdx, dev_chunkWorldBlockPositions + positionIdx,
                                        host_heightfields + (size_t)heightfieldIdx * devHeightfieldSize, dev_heightfields + (size_t)heightfieldIdx * devHeightfieldSize,
                                        host_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize, dev_biomeWeights + (size_t)biomeWeightsIdx * devBiomeWeightsSize,
                                        streams[streamIdx]);
        }
    }

    if (streamIdx > 0) HipUtils::checkError("hipDeviceSynchronize failed", (int)hipDeviceSynchronize());
}

std::unordered_set<Chunk*> Terrain::getDrawableChunks() { return drawableChunks; }
ivec2 Terrain::getCurrentChunkPos() const { return currentChunkPos; }
void Terrain::setCurrentChunkPos(ivec2 p) { currentChunkPos = p; }
int Terrain::getMaxNumDrawableChunks() { const int side = chunkVbosGenRadius * 2 + 1; return side * side; }

bool Terrain::allQueuesEmpty() const
{
    return chunksToGenerateHeightfield.empty() && chunksToGatherHeightfield.empty() && chunksToGenerateLayers.empty() && zonesToTryErosion.empty() &&
           zonesToErode.empty() && chunksToGenerateCaves.empty() && chunksToGenerateFeaturePlacements.empty() && chunksToGatherFeaturePlacements.empty() &&
           chunksToFill.empty() && chunksToCreateAndBufferVbos.empty();
}

Chunk* Terrain::findChunk(ivec2 c)
{
    const ivec2 z = zonePosFromChunkPos(c);
    auto it = zones.find({z.x, z.y});
    if (it == zones.end()) return nullptr;
    return it->second->chunks[(c.x - z.x) + ZONE_SIZE * (c.y - z.y)].get();
}

size_t Terrain::numChunks() const
{
    size_t n = 0;
    for (auto& kv : zones) for (auto& c : kv.second->chunks) n += c != nullptr;
    return n;
}

}  // namespace mmhost
