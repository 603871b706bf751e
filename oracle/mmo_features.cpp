// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// L2 — restatement of src/terrain/featurePlacement.hpp: SDF helpers :15-34, De Casteljau :40-66, isInRasterizedLine :68-74,
// jungleLeaves :80-90, crystal helpers :92-142, placeFeature :147-1107 (21 surface features), placeCaveFeature :1110-1379 (10).
// Line helpers from src/util/rng.hpp:9-63.
//
// CANONICAL: where the reference draws several random numbers inside one constructor call, e.g.
// vec3(u11(rng), u01(rng), u11(rng)) (featurePlacement.hpp:224,700,991,1066), C++ leaves the evaluation order of the
// arguments unspecified; the canonical order is left to right (x, then y, then z), spelled out below with temporaries.
#include "mmo_stages.h"

namespace mmo {

namespace {


#define dev_dirVecs2d (T().dirVecs2d)      // the reference's __constant__ copy of BiomeUtils' table (biomeFuncs.hpp:709-723)

// glm::angle(x, y) of two normalised vectors, gtx/vector_angle.inl:16-21
static inline float g_angle(vec3 x, vec3 y) { return mm_acosf(g_clamp(g_dot(x, y), -1.f, 1.f)); }

// ---- featurePlacement.hpp:15-34
static inline float sdSphere(vec3 p, float s) { return g_length(p) - s; }
static inline float sdCappedCylinder(vec3 p, float r, float h)
{
    vec2 d = g_abs(vec2(g_length(vec2(p.x, p.z)), p.y)) - vec2(r, h);
    return fminf(fmaxf(d.x, d.y), 0.0f) + g_length(g_max(d, vec2(0.f)));
}
static inline float opSubtraction(float d1, float d2) { return fmaxf(d1, -d2); }
static inline float opOnion(float sdf, float thickness) { return fabsf(sdf) - thickness; }

// ---- featurePlacement.hpp:40-66
template <int numCtrlPts, int splineSize>
static void deCasteljau(vec3* ctrlPts, vec3* spline)
{
    for (int i = 0; i < splineSize; ++i) {
        vec3 ctrlPtsCopy[numCtrlPts];
        for (int j = 0; j < numCtrlPts; ++j) ctrlPtsCopy[j] = ctrlPts[j];
        int points = numCtrlPts;
        float t = float(i) / (splineSize - 1);
        while (points > 1) {
            for (int j = 0; j < points - 1; ++j) ctrlPtsCopy[j] = g_mix(ctrlPtsCopy[j], ctrlPtsCopy[j + 1], t);
            --points;
        }
        spline[i] = ctrlPtsCopy[0];
    }
}

// ---- featurePlacement.hpp:68-74
static bool isInRasterizedLine(const ivec3 floorPos, const vec3 linePos1, const vec3 linePos2)
{
    float ratio;
    float distFromLine;
    bool inLine = calculateLineParams(vec3(floorPos) + vec3(0.5f), linePos1, linePos2, &ratio, &distFromLine);
    return inLine && distFromLine < 2.f && floorPos == ivec3(g_floor(g_mix(linePos1, linePos2, ratio)));
}

// ---- featurePlacement.hpp:80-142
static bool jungleLeaves(vec3 pos, float maxHeight, float minRadius, float maxRadius, float rand)
{
    float leavesRadiusMultiplier = 0.8f + 0.4f * rand;
    if (isInRange(pos.y, 0.f, maxHeight)) {
        float leavesRadius = g_mix(maxRadius, minRadius, pos.y / maxHeight) * leavesRadiusMultiplier;
        return g_length(vec2(pos.x, pos.z)) < leavesRadius;
    }
    return false;
}

static const float crystalConeStart = 0.8f;
static const float crystalConeN = 1.f / (1.f - crystalConeStart);

static float getCrystalRadius(float ratio)
{
    if (ratio < crystalConeStart) return 0.8f + 0.25f * ratio;
    else return crystalConeN * (1.f - ratio);
}

static bool isInCrystal(vec3 pos, vec3 pos1, vec3 pos2, float radiusMultiplier)
{
    float ratio;
    float distanceFromLine;
    bool inLine = calculateLineParams(pos, pos1, pos2, &ratio, &distanceFromLine);
    if (!inLine) return false;

    float crystalRadius = getCrystalRadius(ratio) * radiusMultiplier;
    constexpr float p = PI / 6.f;
    const vec3 line = pos2 - pos1;
    const vec3 pointPos = pos - (pos1 + ratio * line);
    const float posAngle = g_length(pointPos) == 0.f ? 0.f : (g_angle(g_normalize(pointPos), g_normalize(g_cross(line, vec3(1, 0, 0)))) + TWO_PI);
    crystalRadius *= mm_cosf(p) / mm_cosf(p - mm_fmodf(posAngle, 2.f * p));
    return distanceFromLine < crystalRadius;
}

static Block getRandomCrystalBlock(float rand)
{
    float crystalRand = rand * 3.f;
    if (crystalRand < 1.f) return Block::MAGENTA_CRYSTAL;
    else if (crystalRand < 2.f) return Block::CYAN_CRYSTAL;
    else return Block::GREEN_CRYSTAL;
}

}  // namespace

// ===================================================================================================
// placeFeature — featurePlacement.hpp:147-1107
// ===================================================================================================
bool placeFeature(const FeaturePlacement& featurePlacement, ivec3 worldBlockPos, Block* blockPtr)
{
    const ivec3& featurePos = featurePlacement.pos;
    ivec3 floorPos = worldBlockPos - featurePos;
    vec3 pos = floorPos;

    auto featureRng = makeSeededRandomEngine(featurePos.x, featurePos.y, featurePos.z, 1293012);
    auto blockRng = makeSeededRandomEngine(worldBlockPos.x, worldBlockPos.y, worldBlockPos.z, 57847812);
    uniform_real_distribution<float> u01(0, 1);
    uniform_real_distribution<float> u11(-1, 1);

    switch (featurePlacement.feature) {
    case Feature::NONE:
        return false;
    case Feature::SPHERE: {                                                    // :164-175
        vec3 diff = worldBlockPos - featurePos;
        if (g_dot(diff, diff) > 25.f) return false;
        *blockPtr = Block::GRAVEL;
        return true;
    }
    case Feature::CORAL: {                                                     // :176-267
        if (featurePos.y > SEA_LEVEL - 6) return false;
        if (g_length(vec2(pos.x, pos.z)) > 8.f) return false;

        int coralRand = (int)(u01(featureRng) * 5.f);
        switch (coralRand) {
        case 0: {
            pos.y *= 1.15f;
            float radius = 2.8f + 1.4f * u01(featureRng);
            radius += 0.4f * simplex(vec3(worldBlockPos) * 0.2f);
            if (g_length(pos) < radius) { *blockPtr = Block::BRAIN_CORAL_BLOCK; return true; }
            return false;
        }
        case 1: {
            pos.y *= 1.25f;
            float radius = 2.2f + 1.7f * u01(featureRng);
            radius += 1.2f * simplex(vec3(worldBlockPos) * 0.3f);
            if (g_length(pos) < radius) { *blockPtr = Block::BUBBLE_CORAL_BLOCK; return true; }
            return false;
        }
        case 2:
        case 3: {
            Block coralBlock = coralRand == 2 ? Block::FIRE_CORAL_BLOCK : Block::HORN_CORAL_BLOCK;
            const vec3 pos1 = vec3_ltr(u11(featureRng), u01(featureRng), u11(featureRng)) * vec3(2.5f, 3.5f, 2.5f);
            if (isInRasterizedLine(floorPos, vec3(0), pos1)) { *blockPtr = coralBlock; return true; }
            for (int i = 0; i < 5; ++i) {
                vec3 pos2 = pos1;
                pos2.x += 4.f * u11(featureRng);
                pos2.y += 2.f + 3.f * u01(featureRng);
                pos2.z += 4.f * u11(featureRng);
                if (isInRasterizedLine(floorPos, pos1, pos2)) { *blockPtr = coralBlock; return true; }
            }
            return false;
        }
        case 4: {
            float edgeDist;
            float height = worley(vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.7f, nullptr, &edgeDist);
            height = (1.f - height) + edgeDist;
            height *= 3.5f;
            height *= g_smoothstep(3.7f, 2.5f, g_length(vec2(pos.x, pos.z)));
            height -= 2.f;
            if (isInRange(pos.y, -1.f, height)) { *blockPtr = Block::TUBE_CORAL_BLOCK; return true; }
            return false;
        }
        }
        return false;
    }
    case Feature::KELP: {                                                      // :268-286
        if (floorPos.x != 0 || floorPos.z != 0) return false;
        int height = (int)(5.f + 15.f * u01(featureRng));
        height = g_min(height, SEA_LEVEL - featurePos.y - 1);
        if (!isInRange(floorPos.y, 0, height)) return false;
        bool isEnd = floorPos.y == height;
        *blockPtr = isEnd ? Block::KELP_END : Block::KELP_MAIN;
        return true;
    }
    case Feature::ICEBERG: {                                                   // :287-322
        if (featurePos.y > SEA_LEVEL - 32) return false;
        pos.y = worldBlockPos.y - SEA_LEVEL;
        float horizontalDistance = g_length(vec2(pos.x, pos.z));
        float icebergRadius = 20.f + 12.f * u01(featureRng);
        float icebergCenterRatio = 1.f - (horizontalDistance / icebergRadius);
        if (icebergCenterRatio > 1.15f) return false;

        vec2 noisePos = vec2((float)worldBlockPos.x, (float)worldBlockPos.z) * 0.0450f;
        float icebergStartHeight = -6.f - 34.f * icebergCenterRatio + 14.f * fbm<3>(noisePos);
        float icebergEndHeight = -4.f + 20.f * icebergCenterRatio + 8.f * fbm<3>(noisePos);
        if (icebergEndHeight < icebergStartHeight || !isInRange(pos.y, icebergStartHeight, icebergEndHeight)) return false;

        if (pos.y < -4.f) { *blockPtr = Block::BLUE_ICE; return true; }
        float packedIceHeight = -2.2f + 5.6f * icebergCenterRatio + 1.2f * simplex(noisePos * 0.8000f);
        *blockPtr = (pos.y > icebergEndHeight - packedIceHeight) ? Block::PACKED_ICE : Block::BLUE_ICE;
        return true;
    }
    case Feature::ACACIA_TREE: {                                               // :323-383
        if (g_max(std::abs(floorPos.x), std::abs(floorPos.z)) > 15) return false;

        int trunkBaseHeight = (int)(4.5f + 1.5f * u01(featureRng));
        if (floorPos.x == 0 && floorPos.z == 0 && isInRange(floorPos.y, 0, trunkBaseHeight)) { *blockPtr = Block::ACACIA_WOOD; return true; }

        float branchAngle = u01(featureRng) * TWO_PI;
        vec3 branchStartPos = vec3(0, (float)trunkBaseHeight, 0);
        vec3 branchEndPos = vec3(0);
        mm_sincosf(branchAngle, &branchEndPos.z, &branchEndPos.x);
        branchEndPos = branchStartPos + (2.f + 1.5f * u01(featureRng)) * branchEndPos;
        branchEndPos.y += 2.5f + 1.5f * u01(featureRng);
        if (isInRasterizedLine(floorPos, g_floor(branchStartPos), g_ceil(branchEndPos))) {
            *blockPtr = Block::ACACIA_WOOD;
            return true;
        }

        vec3 leavesPos = vec3(floorPos) - branchEndPos;
        leavesPos.y += 0.5f;
        if (jungleLeaves(leavesPos, 2.f, 2.f, 4.f, 0.5f + 0.5f * u01(featureRng))) { *blockPtr = Block::ACACIA_LEAVES; return true; }

        if (u01(featureRng) < 0.5f) return false;

        branchAngle += PI_OVER_TWO + u01(featureRng) * PI;
        branchStartPos = vec3(0, (float)trunkBaseHeight - 0.8f - 0.8f * u01(featureRng), 0);
        branchEndPos = vec3(0);
        mm_sincosf(branchAngle, &branchEndPos.z, &branchEndPos.x);
        branchEndPos = branchStartPos + (1.5f + 1.f * u01(featureRng)) * branchEndPos;
        branchEndPos.y += 2.f + 1.f * u01(featureRng);
        if (isInRasterizedLine(floorPos, g_floor(branchStartPos), g_ceil(branchEndPos))) {
            *blockPtr = Block::ACACIA_WOOD;
            return true;
        }

        leavesPos = vec3(floorPos) - branchEndPos;
        leavesPos.y += 0.5f;
        if (jungleLeaves(leavesPos, 2.001f, 1.5f, 3.5f, 0.5f + 0.5f * u01(featureRng))) { *blockPtr = Block::ACACIA_LEAVES; return true; }
        return false;
    }
    case Feature::REDWOOD_TREE: {                                              // :384-473
        pos *= (0.6f + 0.3f * u01(featureRng));

        float height = 27.f + 13.f * u01(featureRng);
        float horizontalDistance = g_length(vec2(pos.x, pos.z));
        float leavesStart = 10.f + 4.f * u01(featureRng);
        if (pos.y > height + 8.f || horizontalDistance > 12.f || (pos.y < leavesStart - 4.f && horizontalDistance > 3.f)) return false;

        float trunkRatio = getRatio(pos.y, -4.f, height);
        if (isSaturated(trunkRatio)) {
            float trunkRadius = 2.f / (trunkRatio + 2.f) + 0.08f / mm_powf(trunkRatio + 0.4f, 3.f);
            trunkRadius += 0.3f * simplex(vec3(worldBlockPos) * 0.1300f) * g_smoothstep(0.6f, 0.2f, trunkRatio);
            if (horizontalDistance < trunkRadius) { *blockPtr = Block::REDWOOD_WOOD; return true; }
        }

        float leavesEnd = height + 1.5f + 1.f * u01(featureRng);
        if (!isInRange(pos.y, leavesStart, leavesEnd)) return false;

        const int leavesCellBaseHeight = (int)floorf(pos.y * 0.5f) * 2;
        const float branchSeed = 593.23f * rand1From3(featurePos);
        const float leavesSeed = 412.39f * rand1From1(branchSeed);
        const float leavesSimplex = 1.1f * simplex(vec3(worldBlockPos) * 0.2000f);
        bool isInLeaves = false;
        for (int dy = -4; dy <= 4; dy += 2) {
            const int leavesCellHeight = leavesCellBaseHeight + dy;

            float leavesHeightRatio = getRatio((float)leavesCellHeight, leavesStart, leavesEnd);
            leavesHeightRatio = 1.1f - 0.5f * leavesHeightRatio;

            vec3 leavesCenter = rand3From2(vec2((float)leavesCellHeight, leavesSeed)) - 0.5f;
            leavesCenter *= vec3(7.5f, 1.3f, 7.5f) * leavesHeightRatio;
            leavesCenter.y = g_min(leavesCenter.y + (float)leavesCellHeight, height + 0.8f);

            vec3 branchStart = vec3(0, leavesCenter.y - 2.f - 1.5f * rand1From1((float)leavesCellHeight + branchSeed), 0);
            vec3 branchEnd = leavesCenter;
            float branchRatio;
            float distFromBranch;
            if (calculateLineParams(pos, branchStart, branchEnd, &branchRatio, &distFromBranch)) {
                if (isSaturated(branchRatio) && distFromBranch < 0.5f) { *blockPtr = Block::REDWOOD_WOOD; return true; }
            }

            if (isInLeaves) continue;

            vec3 leavesPos = pos - leavesCenter;
            leavesPos.y *= 1.7f;
            float leavesDistance = g_length(leavesPos);
            if (leavesDistance > 5.0f) continue;

            float leavesRadius = 2.5f + 0.5f * rand1From1((float)leavesCellHeight + leavesSeed) + leavesSimplex;
            leavesRadius *= leavesHeightRatio;
            if (leavesDistance < leavesRadius) isInLeaves = true;
        }
        if (isInLeaves) { *blockPtr = Block::REDWOOD_LEAVES; return true; }
        return false;
    }
    case Feature::CYPRESS_TREE: {                                              // :474-538
        float trunkHeight = 25.f + 12.f * u01(featureRng);
        float trunkDistance = g_length(vec2(pos.x, pos.z));
        if (pos.y > trunkHeight + 4.f || trunkDistance > 12.f) return false;

        float trunkRatio = getRatio(pos.y, -2.f, trunkHeight);
        if (isSaturated(trunkRatio)) {
            float trunkRadius = 0.5f * ((1.3f + trunkRatio) / mm_powf(0.73f + trunkRatio, 4.f)) + 0.5f;
            trunkRadius *= (1.f + (0.3f * simplex(vec3(worldBlockPos) * 0.1500f)) * g_smoothstep(0.55f, 0.15f, trunkRatio));
            if (trunkDistance < trunkRadius) { *blockPtr = Block::CYPRESS_WOOD; return true; }
        }

        if (jungleLeaves(pos - vec3(0, trunkHeight, 0), 2.f, 3.f, 4.5f, u01(featureRng))) { *blockPtr = Block::CYPRESS_LEAVES; return true; }

        int numBranches = 6 + (int)(u01(featureRng) * 5.f);
        float branchHeight = trunkHeight - 1.f;
        float branchAngle = u01(featureRng) * TWO_PI;
        for (int i = 0; i < numBranches; ++i) {
            branchHeight -= 1.f + 3.6f * u01(featureRng);
            branchAngle += PI_OVER_TWO + u01(featureRng) * PI;

            vec3 branchStart = vec3(0, branchHeight, 0);
            vec3 branchEnd;                                  // y is overwritten below before it is read
            mm_sincosf(branchAngle, &branchEnd.z, &branchEnd.x);
            branchEnd *= 4.f + 1.5f * u01(featureRng);
            branchEnd.y = 2.2f + 1.2f * u01(featureRng);
            branchEnd *= 1.f - 0.3f * getRatio(branchHeight, 0.f, trunkHeight);
            branchEnd += branchStart;

            if (isInRasterizedLine(pos, branchStart, branchEnd)) { *blockPtr = Block::CYPRESS_WOOD; return true; }

            vec3 leavesPos = pos - branchEnd + 0.3f;
            float leavesDroopRand = rand1From2(vec2((float)worldBlockPos.x, (float)worldBlockPos.z));
            if (leavesDroopRand < 0.2f && isInRange(leavesPos.y, g_max(-2.f, leavesDroopRand * -10.f), 0.f)) leavesPos.y = 0.f;

            if (jungleLeaves(leavesPos, 2.f, 2.5f, 4.f, u01(featureRng))) { *blockPtr = Block::CYPRESS_LEAVES; return true; }
        }
        return false;
    }
    case Feature::BIRCH_TREE: {                                                // :539-595
        int height = (int)(6.2f + 4.f * u01(featureRng));
        bool tall = u01(featureRng) < 0.08f;
        if (tall) height *= 1.9f;

        if (g_max(std::abs(floorPos.x), std::abs(floorPos.z)) > 8 || !isInRange(floorPos.y, 0, height + 6)) return false;
        if (floorPos.x == 0 && floorPos.z == 0 && isInRange(floorPos.y, 0, height)) { *blockPtr = Block::BIRCH_WOOD; return true; }

        float leavesTallMultiplier = tall ? 1.5f : 1.f;
        float leavesStart = (float)height - (3.0f - 2.2f * u01(featureRng)) * leavesTallMultiplier;
        float leavesEnd = (float)height + (4.2f + 1.2f * u01(featureRng)) * leavesTallMultiplier;
        float ratio = (pos.y - leavesStart) / (leavesEnd - leavesStart);
        if (!isInRange(ratio, 0.f, 1.f)) return false;

        float x = mm_powf(ratio, 0.8f);
        float leavesRadius = 5.f * (0.5f * x * x * x - 1.5f * x * x + x) * (2.8f + 0.8f * u01(featureRng));
        if (g_length(vec2(pos.x, pos.z)) > leavesRadius) return false;

        Block leafBlock;
        float leafRand = u01(featureRng);
        if (leafRand < 0.1f) leafBlock = Block::YELLOW_BIRCH_LEAVES;
        else if (leafRand < 0.2f) leafBlock = Block::ORANGE_BIRCH_LEAVES;
        else leafBlock = Block::BIRCH_LEAVES;
        *blockPtr = leafBlock;
        return true;
    }
    case Feature::PINE_TREE: {                                                 // :596-626
        int height = (int)(7.f + 4.f * u01(featureRng));
        if (floorPos.y < 0 || floorPos.y > height + 4 || g_max(std::abs(floorPos.x), std::abs(floorPos.z)) > 6) return false;
        if (floorPos.x == 0 && floorPos.z == 0 && floorPos.y <= height) { *blockPtr = Block::PINE_WOOD; return true; }

        float leavesStart = (float)height - 4.f - 2.5f * u01(featureRng);
        float leavesEnd = (float)height + 3.f;
        float leavesRatio = (pos.y - leavesStart) / (leavesEnd - leavesStart);
        if (!isInRange(leavesRatio, 0.f, 1.f)) return false;

        float leavesRadius = g_mix(3.f, 1.f, leavesRatio);
        if (g_length(vec2(pos.x, pos.z)) < leavesRadius) {
            *blockPtr = u01(featureRng) < 0.5f ? Block::PINE_LEAVES_1 : Block::PINE_LEAVES_2;
            return true;
        }
        return false;
    }
    case Feature::PINE_SHRUB: {                                                // :627-649
        int height = (int)(2.f + 2.f * u01(featureRng));
        if (floorPos.y < 0 || floorPos.y > height + 4 || g_max(std::abs(floorPos.x), std::abs(floorPos.z)) > 6) return false;
        if (floorPos.x == 0 && floorPos.z == 0 && floorPos.y <= height) { *blockPtr = Block::PINE_WOOD; return true; }

        vec3 leavesPos = pos - vec3(0, (float)height - 1.f, 0);
        if (jungleLeaves(leavesPos, 2.5f, 1.5f, 2.5f, u01(featureRng))) {
            *blockPtr = u01(featureRng) < 0.5f ? Block::PINE_LEAVES_1 : Block::PINE_LEAVES_2;
            return true;
        }
        return false;
    }
    case Feature::MEDIUM_PURPLE_MUSHROOM: {                                    // :650-672
        if (manhattanLength(ivec2(floorPos.x, floorPos.z)) > 8) return false;
        int height = (int)(1.5f + 2.3f * u01(featureRng));
        if (floorPos.x == 0 && isInRange(floorPos.y, 0, height) && floorPos.z == 0) { *blockPtr = Block::MUSHROOM_STEM; return true; }
        float radius = u01(featureRng) < 0.5f ? 1.8f : 2.5f;
        if (floorPos.y == height + 1 && g_length(vec2(pos.x, pos.z)) < radius) { *blockPtr = Block::PURPLE_MUSHROOM_CAP; return true; }
        return false;
    }
    case Feature::PURPLE_MUSHROOM: {                                           // :673-768
        float universalScale = 1.f + u01(featureRng) * 1.2f;
        pos *= universalScale;
        if (u01(featureRng) < 0.2f) pos *= 0.5f;

        float height = 25.f + u01(featureRng) * 30.f;
        if (pos.y < -1 || pos.y > height + 12
            || (g_length(vec2(pos.x, pos.z)) > 8 && (pos.y < height - 12 || g_length(pos - vec3(0, height, 0)) > 35)))
            return false;

        constexpr int numCtrlPts = 5;
        constexpr int splineSize = 7;
        vec3 endPoint = vec3(0, height, 0);
        vec3 ctrlPts[numCtrlPts];
        constexpr float lastCtrlPtIndex = numCtrlPts - 1.f;
        ctrlPts[0] = vec3(0);
        for (int i = 1; i < numCtrlPts; ++i) {
            vec3 offset = vec3_ltr(u11(featureRng), u11(featureRng), u11(featureRng)) * vec3(6, 2, 6);
            if (i == numCtrlPts - 1) offset *= 0.6f;
            ctrlPts[i] = (endPoint * ((float)i / lastCtrlPtIndex)) + offset;
        }

        constexpr int lastSplineIndex = splineSize - 1;
        vec3 spline[splineSize];
        deCasteljau<numCtrlPts, splineSize>(ctrlPts, spline);

        for (int i = 0; i < splineSize; ++i) {
            vec3 pos1 = spline[i];
            vec3 pos2;
            if (i < lastSplineIndex) {
                pos2 = spline[i + 1];
                if (pos.y < pos1.y - 3 || pos.y > pos2.y + 3) continue;
            } else {
                pos2 = pos1 + g_normalize(pos1 - spline[i - 1]) * (3.f + u01(featureRng) * 1.5f);
            }

            float ratio;
            float distFromLine;
            bool inRatio = calculateLineParams(pos, pos1, pos2, &ratio, &distFromLine);

            float radius;
            Block potentialBlock;
            if (i < lastSplineIndex) {
                float t = ((float)i + g_clamp(ratio, 0.f, 1.f)) / (float)lastSplineIndex;
                float x = t - 0.5f;
                radius = (4.f * x * x + 1.5f) * 1.2f;
                potentialBlock = Block::MUSHROOM_STEM;
            } else {
                radius = (7.f * u01(featureRng) + 12.f) * g_mix(0.8f, 1.2f, (height - 33.f) / 40.f);
                if (distFromLine < radius - 1.8f && ratio < 0.5f && universalScale < 1.4f) potentialBlock = Block::MUSHROOM_UNDERSIDE;
                else potentialBlock = Block::PURPLE_MUSHROOM_CAP;
            }

            if ((inRatio && distFromLine <= radius)
                || (i < lastSplineIndex && ratio < 0 && g_distance(pos, pos1) < radius)
                || (i < (splineSize - 2) && ratio > 1 && g_distance(pos, pos2) < radius)) {
                *blockPtr = potentialBlock;
                return true;
            }
        }
        return false;
    }
    case Feature::RAFFLESIA: {                                                 // :769-819
        if (pos.y > 10.f || g_length(pos) > 15.f) return false;
        pos *= 0.8f;

        vec3 centerSdfPos = pos;
        centerSdfPos.y -= 1.f;
        centerSdfPos.y *= 1.4f;
        if (sdSphere(centerSdfPos, 1.f) < 0) { *blockPtr = Block::RAFFLESIA_SPIKES; return true; }

        float centerSdf = opOnion(sdSphere(centerSdfPos - vec3(0, 1, 0), 2.0f), 0.8f);
        float centerHoleSdf = sdSphere(centerSdfPos - vec3(0, 1.8f, 0), 1.8f);
        centerSdf = opSubtraction(centerSdf, centerHoleSdf);
        if (centerSdf < 0.f) { *blockPtr = centerSdfPos.y > 1.f ? Block::RAFFLESIA_CENTER : Block::RAFFLESIA_STEM; return true; }

        const float petalStartAngle = u01(featureRng) * TWO_PI;
        for (int i = 0; i < 5; ++i) {
            const float petalAngle = petalStartAngle + ((float)i * TWO_PI * 0.2f);
            float sinTheta, cosTheta;
            mm_sincosf(-petalAngle, &sinTheta, &cosTheta);
            vec3 petalPos = vec3(pos.x * cosTheta + pos.z * sinTheta, pos.y - 3.2f, -pos.x * sinTheta + pos.z * cosTheta);
            petalPos.y -= (float)(i % 2) * 0.53f;
            petalPos.y += g_clamp((fabsf(petalPos.x - 3.f) - 1.5f) / 1.5f, 0.f, 1.f) * 1.3f;
            petalPos.x -= 3.8f;
            petalPos.z *= 1.2f;
            if (sdCappedCylinder(petalPos, 2.5f, 0.5f) < 0) { *blockPtr = Block::RAFFLESIA_PETAL; return true; }
        }
        return false;
    }
    case Feature::LARGE_JUNGLE_TREE: {                                         // :820-879
        float height = 18.f + 10.f * u01(featureRng);
        if (pos.y > height + 6.f || g_length(vec2(pos.x, pos.z)) > 15.f) return false;

        ivec2 trunkPos = ivec2(g_floor(vec2(pos.x, pos.z)));
        if (isInRange(pos.y, 0.f, height) && trunkPos.x >= 0 && trunkPos.x <= 1 && trunkPos.y >= 0 && trunkPos.y <= 1) {
            *blockPtr = Block::JUNGLE_WOOD;
            return true;
        }

        pos -= vec3(0.5f, 0, 0.5f);

        vec3 leavesPos = pos;
        leavesPos.y -= (height - 2.f);
        if (jungleLeaves(leavesPos, 4.f, 4.f, 7.f, u01(featureRng))) {
            *blockPtr = u01(blockRng) < 0.5f ? Block::JUNGLE_LEAVES_FRUITS : Block::JUNGLE_LEAVES_PLAIN;
            return true;
        }

        float numBranches = 0.5f + 2.5f * u01(featureRng);
        float branchHeight = height;
        for (int i = 0; (float)i < numBranches; ++i) {
            branchHeight -= (8.f + u01(featureRng) * 3.f) * (height / 30.f);
            float branchAngle = TWO_PI * u01(featureRng);

            vec3 branchStart = vec3(0, branchHeight, 0);
            vec3 branchEnd = vec3(0);
            mm_sincosf(-branchAngle, &branchEnd.z, &branchEnd.x);
            branchEnd = ((3.f + 1.5f * u01(featureRng)) * branchEnd) + branchStart;
            branchEnd.y += 1.f + 1.5f * u01(featureRng);

            float ratio;
            float distFromLine;
            bool inRatio = calculateLineParams(pos, branchStart, branchEnd, &ratio, &distFromLine);
            float branchRadius = 1.2f - (0.4f * ratio);
            if (inRatio && distFromLine < branchRadius) { *blockPtr = Block::JUNGLE_WOOD; return true; }

            leavesPos = pos - branchEnd + vec3(0, 0.2f, 0);
            if (jungleLeaves(leavesPos, 2.f, 2.5f, 3.5f, u01(featureRng))) {
                *blockPtr = u01(blockRng) < 0.25f ? Block::JUNGLE_LEAVES_FRUITS : Block::JUNGLE_LEAVES_PLAIN;
                return true;
            }
        }
        return false;
    }
    case Feature::SMALL_JUNGLE_TREE: {                                         // :880-903
        float height = 8.f + 4.f * u01(featureRng);
        float maxDist = pos.y < height - 2.f ? 2.f : 8.f;
        if (pos.y > height + 4.f || g_length(vec2(pos.x, pos.z)) > maxDist) return false;

        if (isInRange(pos.y, 0.f, height) && ivec2(g_floor(vec2(pos.x, pos.z))) == ivec2(0)) { *blockPtr = Block::JUNGLE_WOOD; return true; }

        vec3 leavesPos = pos - vec3(0, height - 1.f, 0);
        if (jungleLeaves(leavesPos, 3.f, 2.f, 4.f, u01(featureRng))) {
            *blockPtr = u01(blockRng) < 0.25f ? Block::JUNGLE_LEAVES_FRUITS : Block::JUNGLE_LEAVES_PLAIN;
            return true;
        }
        return false;
    }
    case Feature::TINY_JUNGLE_TREE: {                                          // :904-925
        if (compAdd(floorPos) > 8) return false;
        int height = (int)(0.5f + 2.5f * u01(featureRng));
        if (floorPos.x == 0 && isInRange(floorPos.y, 0, height) && floorPos.z == 0) { *blockPtr = Block::JUNGLE_WOOD; return true; }
        if (manhattanDistance(floorPos, ivec3(0, height, 0)) == 1) { *blockPtr = Block::JUNGLE_LEAVES_PLAIN; return true; }
        return false;
    }
    case Feature::CACTUS: {                                                    // :926-971
        if (std::abs(floorPos.x) > 5 || std::abs(floorPos.z) > 5) return false;
        int height = (int)(7.5f + u01(featureRng) * 6.0f);
        if (pos.y > (float)height + 2.f) return false;
        if (floorPos.x == 0 && isInRange(floorPos.y, 0, height) && floorPos.z == 0) { *blockPtr = Block::CACTUS; return true; }

        for (int armIdx = 0; armIdx < 4; ++armIdx) {
            if (u01(featureRng) >= 0.35f) continue;
            int armStartHeight = (int)(4.f + u01(featureRng) * (float)(height - 10));
            int armLength = (int)(2.f + u01(featureRng) * 1.f);
            int armHeight = (int)(3.f + u01(featureRng) * 3.f);
            armHeight = g_min(height - armStartHeight - 1, armHeight);

            ivec3 armPos1 = ivec3(0, armStartHeight, 0);
            const ivec2 armDirection = dev_dirVecs2d[armIdx * 2];
            ivec3 armPos2 = armPos1 + (ivec3(armDirection.x, 0, armDirection.y) * armLength);
            ivec3 armPos3 = armPos2 + ivec3(0, armHeight, 0);
            if (isPosInRange(floorPos, armPos1, armPos2) || isPosInRange(floorPos, armPos2, armPos3)) { *blockPtr = Block::CACTUS; return true; }
        }
        return false;
    }
    case Feature::PALM_TREE: {                                                 // :972-1043
        if (floorPos.y < -2 || floorPos.y > 28 || std::abs(floorPos.x) + std::abs(floorPos.z) > 24) return false;

        constexpr int numCtrlPts = 4;
        constexpr int splineSize = 5;
        vec3 minPos = vec3(0);
        vec3 maxPos = vec3(0);
        vec3 ctrlPts[numCtrlPts];
        vec3 currentPoint = vec3(0);
        ctrlPts[0] = currentPoint;
        for (int i = 1; i < numCtrlPts; ++i) {
            float randomWalkScale = 1.f + ((float)i / numCtrlPts) * 5.f;
            currentPoint += vec3_ltr(randomWalkScale * u11(featureRng), 3.f + 5.f * u01(featureRng), randomWalkScale * u11(featureRng));
            ctrlPts[i] = currentPoint;
            minPos = g_min(minPos, currentPoint);
            maxPos = g_max(maxPos, currentPoint);
        }
        if (!isPosInRange(pos, minPos - vec3(7, 1, 7), maxPos + vec3(7, 6, 7))) return false;

        vec3 spline[splineSize];
        deCasteljau<numCtrlPts, splineSize>(ctrlPts, spline);

        ivec3 trunkTop = ivec3(g_floor(spline[splineSize - 1]));
        ivec3 leavesPos = floorPos - trunkTop;
        float leavesDistance = g_length(vec2((float)leavesPos.x, (float)leavesPos.z));
        leavesDistance *= 0.6f + (0.3f * saturate((float)(20 - trunkTop.y) * 0.05f)) + (0.3f * u01(featureRng));
        if (isInRange(leavesPos.y, -1, 0) && leavesDistance < 3.9f
            && (leavesPos.x == 0 || leavesPos.z == 0 || std::abs(leavesPos.x) == std::abs(leavesPos.z))) {
            int leavesHeight = leavesDistance > 3.f ? -1 : 0;
            if (leavesPos.y == leavesHeight) { *blockPtr = Block::PALM_LEAVES; return true; }
        }

        for (int i = 0; i < splineSize - 1; ++i) {
            vec3 pos1 = spline[i];
            vec3 pos2 = spline[i + 1];
            vec3 padding = g_normalize(pos2 - pos1) * 0.5f;
            if (i > 0) pos1 -= padding;
            if (i + 1 < splineSize - 1) pos2 += padding;
            if (isInRasterizedLine(floorPos, pos1, pos2)) { *blockPtr = Block::PALM_WOOD; return true; }
        }
        return false;
    }
    case Feature::MEDIUM_CRYSTAL:
    case Feature::CRYSTAL: {                                                   // :1044-1102
        if (featurePos.y > 180) return false;

        pos += vec3(0, 2, 0);
        pos *= 0.55f + 0.4f * u01(featureRng);
        if (featurePlacement.feature == Feature::MEDIUM_CRYSTAL) pos *= 2.f;

        if (g_max(std::abs(floorPos.x), std::abs(floorPos.z)) > 25) return false;

        vec3 crystalEndPos = vec3_ltr(12.f * u11(featureRng), 18.f + 8.f * u01(featureRng), 12.f * u11(featureRng));
        if (pos.y > crystalEndPos.y + 2.f) return false;

        Block crystalBlock = getRandomCrystalBlock(u01(featureRng));
        if (isInCrystal(pos, vec3(0), crystalEndPos, 4.f + 1.2f * u01(featureRng))) { *blockPtr = crystalBlock; return true; }

        pos *= 0.8f;

        int numSmallCrystals = (int)(4.f + 2.f * u01(featureRng));
        float smallCrystalAngle = u01(featureRng) * TWO_PI;
        for (int i = 0; i < numSmallCrystals; ++i) {
            smallCrystalAngle += PI_OVER_TWO + PI * u01(featureRng);
            vec3 smallCrystalStartPos = vec3(0);
            mm_sincosf(smallCrystalAngle, &smallCrystalStartPos.z, &smallCrystalStartPos.x);
            vec3 smallCrystalEndPos = smallCrystalStartPos;
            smallCrystalStartPos *= 3.f;
            smallCrystalEndPos *= 6.f + 3.f * u01(featureRng);
            smallCrystalEndPos.y = 7.f + 5.f * u01(featureRng);
            if (isInCrystal(pos, vec3(0), smallCrystalEndPos, 1.5f + 1.5f * u01(featureRng))) { *blockPtr = crystalBlock; return true; }
        }
        return false;
    }
    }
    return false;
}

// ===================================================================================================
// placeCaveFeature — featurePlacement.hpp:1110-1379
// ===================================================================================================
bool placeCaveFeature(const CaveFeaturePlacement& caveFeaturePlacement, ivec3 worldBlockPos, Block* blockPtr)
{
    const ivec3& featurePos = caveFeaturePlacement.pos;
    const int layerHeight = caveFeaturePlacement.layerHeight;
    ivec3 floorPos = worldBlockPos - featurePos;
    ivec3 floorTopPos = worldBlockPos - (featurePos + ivec3(0, layerHeight, 0));
    vec3 pos = floorPos;
    vec3 topPos = floorTopPos;

    auto featureRng = makeSeededRandomEngine(featurePos.x, featurePos.y, featurePos.z, 398132);
    auto blockRng = makeSeededRandomEngine(worldBlockPos.x, worldBlockPos.y, worldBlockPos.z, 9322743);
    uniform_real_distribution<float> u01(0, 1);
    uniform_real_distribution<float> u11(-1, 1);

    switch (caveFeaturePlacement.feature) {
    case CaveFeature::NONE:
        return false;
    case CaveFeature::TEST_GLOWSTONE_PILLAR:
        if (floorPos.x == 0 && floorPos.z == 0 && isInRange(floorPos.y, 0, layerHeight)) { *blockPtr = Block::GLOWSTONE; return true; }
        return false;
    case CaveFeature::TEST_SHROOMLIGHT_PILLAR:
        if (floorPos.x == 0 && floorPos.z == 0 && isInRange(floorPos.y, 0, layerHeight)) { *blockPtr = Block::SHROOMLIGHT; return true; }
        return false;
    case CaveFeature::CAVE_VINE: {                                             // :1150-1177
        if (floorTopPos.x != 0 || floorTopPos.z != 0) return false;
        int height = (int)(3.f + 12.f * u01(featureRng));
        height = g_min(height, layerHeight);
        if (!isInRange(floorTopPos.y, -height, 0)) return false;

        bool glowing = u01(blockRng) < 0.2f;
        bool isEnd = floorTopPos.y == -height;
        if (isEnd) *blockPtr = glowing ? Block::CAVE_VINES_GLOW_END : Block::CAVE_VINES_END;
        else *blockPtr = glowing ? Block::CAVE_VINES_GLOW_MAIN : Block::CAVE_VINES_MAIN;
        return true;
    }
    case CaveFeature::GLOWSTONE_CLUSTER: {                                     // :1178-1198
        topPos.y *= 1.35f;
        topPos *= 1.f + 0.5f * u01(featureRng);
        float thisRadius = g_length(topPos);
        if (thisRadius > 6.f) return false;

        float xzAngle = mm_atan2f(pos.z, pos.x);
        float maxRadius = 3.5f + 2.f * simplex(vec2(xzAngle, (float)worldBlockPos.y) * 1.5f);
        if (thisRadius < maxRadius) { *blockPtr = Block::GLOWSTONE; return true; }
        return false;
    }
    case CaveFeature::STORMLIGHT_SPHERE:
    case CaveFeature::CEILING_STORMLIGHT_SPHERE: {                             // :1199-1222
        float radius = 3.5f + 4.f * u01(featureRng);
        float dist = caveFeaturePlacement.feature == CaveFeature::STORMLIGHT_SPHERE ? g_length(pos) : g_length(topPos);
        if (dist > radius) return false;

        float radiusRatio = dist / radius;
        float lightChance = g_smoothstep(0.4f, 0.2f, radiusRatio);
        if (u01(blockRng) < lightChance) *blockPtr = Block::GLOWSTONE;
        else *blockPtr = getRandomCrystalBlock(u01(featureRng));
        return true;
    }
    case CaveFeature::CRYSTAL_PILLAR: {                                        // :1223-1266
        if (pos.y < -8.f || topPos.y > 8.f) return false;
        float dist = g_length(vec2(pos.x, pos.z));
        if (dist > 7.f) return false;

        float heightRatio = pos.y / (float)layerHeight;
        if (heightRatio < 0.f) { heightRatio = 0.f; dist = g_length(pos); }
        else if (heightRatio > 1.f) { heightRatio = 1.f; dist = g_length(topPos); }

        float radius = heightRatio - 0.5f;
        radius = 4.f * (2.f * radius * radius + 0.5f);
        if (dist > radius) return false;

        float radiusRatio = dist / radius;
        if (radiusRatio < 0.4f) *blockPtr = Block::GLOWSTONE;
        else *blockPtr = getRandomCrystalBlock(u01(featureRng));
        return true;
    }
    case CaveFeature::WARPED_FUNGUS: {                                         // :1267-1317
        if (manhattanLength(ivec2(floorPos.x, floorPos.z)) > 6) return false;
        int height = (int)(2.5f + 3.0f * u01(featureRng));
        if (floorPos.y < -2 || floorPos.y > height + 3) return false;
        if (floorPos.x == 0 && floorPos.z == 0 && isInRange(floorPos.y, 0, height)) { *blockPtr = Block::WARPED_STEM; return true; }

        int shroomlightHeight = floorPos.y - (height - 1);
        if (isInRange(shroomlightHeight, 0, 1) && manhattanLength(ivec2(floorPos.x, floorPos.z)) == 1) {
            float shroomlightChance = shroomlightHeight == 0 ? 0.2f : 0.5f;
            if (u01(blockRng) < shroomlightChance) { *blockPtr = Block::SHROOMLIGHT; return true; }
        }

        float capRadius = g_length(vec2(pos.x, pos.z));
        if (capRadius > 3.7f) return false;

        int capHeightEnd = height + 1 - (int)(capRadius / 2.5f);
        int capHeightStart = capHeightEnd - (4.2f
            * simplex((vec2(worldBlockPos.x, worldBlockPos.z) + vec2(featurePos.y)) * 3.f)
            * fmaxf(capRadius - 2.3f, 0.f));
        if (isInRange(floorPos.y, capHeightStart, capHeightEnd)) { *blockPtr = Block::WARPED_WART; return true; }
        return false;
    }
    case CaveFeature::AMBER_FUNGUS: {                                          // :1318-1375
        int manhattanLength2d = manhattanLength(ivec2(floorPos.x, floorPos.z));
        if (manhattanLength2d > 4) return false;
        int height = (int)(4.5f + 4.5f * u01(featureRng));
        if (floorPos.y < -2 || floorPos.y > height + 3) return false;

        if (floorPos.x == 0 && floorPos.z == 0) {
            if (isInRange(floorPos.y, 0, height)) { *blockPtr = Block::AMBER_STEM; return true; }
            else if (floorPos.y == height + 1) { *blockPtr = Block::AMBER_WART; return true; }
        }

        int capStart = height / 2;
        if (simplex(vec2((float)worldBlockPos.x, (float)worldBlockPos.z)) < 0.f) capStart -= 1;

        if (isInRange(floorPos.y, capStart, height)) {
            int capManhattanDist = (floorPos.y - capStart) < (height / 4 + 1) ? 2 : 1;
            if (manhattanLength2d == capManhattanDist) {
                ivec3 shroomlightGridCorner = (worldBlockPos / 2) * 2;
                ivec3 shroomlightGridRandPos = shroomlightGridCorner + ivec3(rand3From3(shroomlightGridCorner) * 2.f);
                if (worldBlockPos == shroomlightGridRandPos && u01(blockRng) < 0.65f) *blockPtr = Block::SHROOMLIGHT;
                else *blockPtr = Block::AMBER_WART;
                return true;
            }
        }
        return false;
    }
    }
    return false;
}

}  // namespace mmo
