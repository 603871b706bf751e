"""Constants of the reference's arithmetic, section by section (tests/golden/ref_literals.json, extracted from the reference's own text by
tools/extract_ref_literals.py: numbers only).  The oracle restates biomeFuncs.hpp / featurePlacement.hpp / rng.hpp / chunk.cu function by
function with the same function and case names; the device code uses its own names (biome_height, MMBIO_*, MMF_*, ...).  Every numeric
value the reference writes in a function (or in one case of a switch) must also appear in the corresponding section of the oracle and of
the device code: a mistyped or dropped constant - the common-mode error HIP-vs-oracle parity cannot see - fails here."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from extract_ref_literals import (ANCHORS, ORACLE_FILES, REFERENCE_SECTIONS, canonical_blocks, literals, oracle_anchors, oracle_signature, sections,   # noqa: E402
                                  skeleton, skeleton_digest, strip_comments)

REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_literals.json")))



def _read(paths):
    return "\n".join(strip_comments(open(os.path.join(ROOT, p)).read()) for p in paths)


def _check(ours, key, allow=()):
    want = set(REF[key])
    missing = sorted(v for v in want - set(ours) if not any(abs(v - a) <= 1e-6 * max(1.0, abs(a)) for a in allow))
    assert not missing, f"{key}: values of the reference missing from our section: {missing}"


def _oracle_sections():
    out = {}
    cache = {}
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        base = os.path.basename(rel)
        if base not in cache:
            cache[base] = _read(ORACLE_FILES[base])
        secs = sections(cache[base], oracle_signature(f"{base}::{key}", sig), prefixes, oracle_anchors(key))
        for name, code in secs.items():
            out[f"{base}::{key}" + (f"::{name}" if name else "")] = literals(code)
    return out


# values the reference writes that our restatement legitimately does not (each with its reason)
ORACLE_ALLOW = {
    "featurePlacement.hpp::placeFeature": (0.0, 1.0),        # preamble before the switch: vec3 / ivec3 helpers
    "featurePlacement.hpp::placeCaveFeature": (1.0,),
}
ORACLE_EXTRA = {
    "chunk.cu::kernGenerateCaves.store": (3.0,),             # CANONICAL_CAVE_LAYER_OVERFLOW: 3 ints per layer slot
}


def test_oracle_sections_hold_every_reference_constant():
    ours = _oracle_sections()
    problems = []
    for key in sorted(REF):
        if key not in ours:
            problems.append(f"{key}: section not found in the oracle")
            continue
        try:
            _check(ours[key], key, ORACLE_ALLOW.get(key, ()))
        except AssertionError as e:
            problems.append(str(e))
        # and the other way round: the oracle is a near-verbatim restatement, so a value the reference's section does not contain is a
        # typo (a constant that is duplicated by scalarisation would hide one wrong copy from the inclusion test above)
        extra = sorted(set(ours[key]) - set(REF[key]) - set(ORACLE_EXTRA.get(key, ())))
        if extra:
            problems.append(f"{key}: values in the oracle's section that the reference's section does not contain: {extra}")
    assert not problems, "\n".join(problems)


# ---------------------------------------------------------------------------------------------------------------- device code
CS = "mega-minecraft_amd/csrc/"
# reference function -> (our files, our function signatures whose bodies together restate it, case prefix of ours or None = compare whole function)
DEVICE_MAP = {
    "biomeFuncs.hpp::getSingleBiomeNoise": ([CS + "mm_biome.cuh"], [r"float\s+single_biome_noise\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getBiomeNoise": ([CS + "mm_biome.cuh"], [r"BiomeNoise\s+biome_noise\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getCaveBiomeNoise": ([CS + "mm_biome.cuh"], [r"int\s+cave_biome_t\s*\([^)]*\)\s*\{", r"int\s+cave_biome_draw\s*\([^)]*\)\s*\{", r"bool\s+cave_biome_py\s*\([^)]*\)\s*\{", r"int\s+cave_biome_rest\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getHeight": ([CS + "mm_biome.cuh"], [r"float\s+biome_height\s*\([^)]*\)\s*\{"], "MMBIO_"),
    "biomeFuncs.hpp::biomeBlockPreProcess": ([CS + "mm_biome.cuh"], [r"bool\s+biome_block_pre\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::biomeBlockPostProcess": ([CS + "mm_biome.cuh"], [r"void\s+biome_block_post\s*\([^)]*\)\s*\{"], "MMBIO_"),
    "biomeFuncs.hpp::caveBiomeBlockPostProcess": ([CS + "mm_biome.cuh"], [r"void\s+cave_post_noise_pos\s*\([^)]*\)\s*\{", r"bool\s+cave_post_apply\s*\([^)]*\)\s*\{",
                                                                        r"uint8_t\s+lush_clay_or_moss\s*\([^)]*\)\s*\{", r"void\s+cave_biome_block_post\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::sdCappedCylinder": ([CS + "mm_features.cuh"], [r"float\s+sd_capped_cylinder\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::isInRasterizedLine": ([CS + "mm_features.cuh"], [r"bool\s+in_rasterized_line\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::jungleLeaves": ([CS + "mm_features.cuh"], [r"bool\s+jungle_leaves\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::getCrystalRadius": ([CS + "mm_features.cuh"], [r"float\s+crystal_radius\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::isInCrystal": ([CS + "mm_features.cuh"], [r"bool\s+in_crystal\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::getRandomCrystalBlock": ([CS + "mm_features.cuh"], [r"uint8_t\s+random_crystal_block\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::placeFeature": ([CS + "mm_features.cuh"], [r"bool\s+place_feature\s*\([^)]*\)\s*\{", r"uint32_t\s+surface_feature_stream\s*\([^)]*\)\s*\{"], "MMF_"),
    "featurePlacement.hpp::placeCaveFeature": ([CS + "mm_features.cuh"], [r"bool\s+place_cave_feature\s*\([^)]*\)\s*\{", r"uint32_t\s+cave_feature_stream\s*\([^)]*\)\s*\{"], "MMCF_"),
    "biomeFuncs.hpp::getRandomBiome": ([CS + "mm_biome.cuh"], [r"int\s+random_biome\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::applySingleBiomeNoise": ([CS + "mm_biome.cuh"], [r"float\s+biome_weight\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getBiomeWeight": ([CS + "mm_biome.cuh"], [r"float\s+biome_weight\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getCaveBiomeWeight": ([CS + "mm_biome.cuh"], [r"int\s+cave_biome_t\s*\([^)]*\)\s*\{", r"int\s+cave_biome_draw\s*\([^)]*\)\s*\{", r"bool\s+cave_biome_py\s*\([^)]*\)\s*\{", r"int\s+cave_biome_rest\s*\([^)]*\)\s*\{"], None),
    "biomeFuncs.hpp::getCaveBiome": ([CS + "mm_biome.cuh"], [r"int\s+cave_biome_t\s*\([^)]*\)\s*\{", r"int\s+cave_biome_draw\s*\([^)]*\)\s*\{", r"bool\s+cave_biome_py\s*\([^)]*\)\s*\{", r"int\s+cave_biome_rest\s*\([^)]*\)\s*\{"], None),
    "featurePlacement.hpp::deCasteljau": ([CS + "mm_features.cuh"], [r"void\s+de_casteljau\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::saturate": ([CS + "mm_features.cuh"], [r"bool\s+saturated\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::isSaturated": ([CS + "mm_features.cuh"], [r"bool\s+saturated\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::fbm": ([CS + "mm_noise.cuh"], [r"float\s+fbm2_loop\s*\([^)]*\)\s*\{", r"float\s+fbm3_loop\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::worley2": ([CS + "mm_noise.cuh"], [r"Worley2\s+worley2\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::worley3": ([CS + "mm_noise.cuh"], [r"Worley3\s+worley3\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::chunkFillPlaceBlock": ([CS + "mmgen_kernels.hip"], [r"BaseBlock\s+place_block_base\s*\([^)]*\)\s*\{", r"void\s+fill_body\s*\([^{]*\)\s*\{"], None),
    "chunk.cu::tryGenerateCaveFeaturePlacement": ([CS + "mmgen_features.hip"], [r"void\s+column_placements\s*\([^{]*\)\s*\{"], None),
    "chunk.cu::tryPlaceSingleDecorator": ([CS + "mmgen_features.hip"], [r"void\s+try_place_decorator\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::hash": ([CS + "mm_math.cuh"], [r"uint32_t\s+hash32\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::makeSeededRandomEngine": ([CS + "mm_math.cuh"], [r"MinStd\s+rng3\s*\([^)]*\)\s*\{", r"MinStd\s+rng4\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand1From1": ([CS + "mm_features.cuh", CS + "mm_noise.cuh"], [r"float\s+rand1from1\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand1From2": ([CS + "mm_noise.cuh"], [r"float\s+rand1from2\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand1From3": ([CS + "mm_noise.cuh"], [r"float\s+rand1from3\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand2From2": ([CS + "mm_noise.cuh"], [r"f2\s+rand2from2\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand2From3": ([CS + "mm_noise.cuh"], [r"f2\s+rand2from3\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand3From2": ([CS + "mm_noise.cuh"], [r"f3\s+rand3from2\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::rand3From3": ([CS + "mm_noise.cuh"], [r"f3\s+rand3from3\s*\([^)]*\)\s*\{", r"float\s+hash_unit\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::fbm2From2": ([CS + "mm_noise.cuh"], [r"f2\s+fbm2from2\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::fbm3From3": ([CS + "mm_noise.cuh"], [r"f3\s+fbm3from3\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::simplex2From2": ([CS + "mm_noise.cuh"], [r"f2\s+simplex2from2\s*\([^)]*\)\s*\{"], None),
    "rng.hpp::specialCaveNoise": ([CS + "mm_noise.cuh"], [r"float\s+special_cave_noise\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::shouldGenerateCaveAtBlock": ([CS + "mmgen_kernels.hip"], [r"\bk_cave_columns\s*\([^)]*\)\s*\{", r"\bcave_voxels_body\s*\([^{]*\)\s*\{", r"float\s+cave_huge\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::getStratifiedMaterialThickness": ([CS + "mmgen_kernels.hip"], [r"float\s+stratified_thickness\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::isFeaturePos": ([CS + "mmgen_features.hip"], [r"bool\s+is_feature_pos\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::generateColumnFeaturePlacements": ([CS + "mmgen_features.hip"], [r"void\s+column_placements\s*\([^{]*\)\s*\{"], None),
    "chunk.cu::placeDecorators": ([CS + "mmgen_features.hip"], [r"\bk_decorators\s*\([^{]*\)\s*\{"], None),
}
# values the device code may contain beyond the reference's: loop / layout constants of the scalarised, lane-mapped formulation
DEVICE_BENIGN = {0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 9.0, 16.0, 63.0, 64.0, 255.0, 256.0, 384.0, 65535.0, 2147483648.0, 48271.0,
                 float(np.float32(3.402823466e+38))}
DEVICE_EXTRA = {
    # fill_body: lane / batch geometry of the row workgroup (16 columns x 384 y per row, 13-bit positions, 5-bit depth codes 30 / 31, 2^32 queue keys) and the
    # "further than 7 blocks from every cave surface" test of the CRYSTAL-only evaluations (LUSH_CAVES converts within 1.5 + 4.5 |simplex3|)
    "chunk.cu::chunkFillPlaceBlock": (1.5, 4.5, 7.0, 8.0, 13.0, 15.0, 17.0, 30.0, 31.0, 45.0, 96.0, 4294967296.0),
    # special_cave_noise's staged search keeps the three smallest squared distances as unsigned integers: 0x7f7fffff = the bits of FLT_MAX
    "rng.hpp::specialCaveNoise": (float(np.float32(0x7f7fffff)),),
    "biomeFuncs.hpp::getBiomeNoise": (float(np.float32(0.32)),),         # overallBiomeScale, a file-level constant in the reference (biomeFuncs.hpp:105)
    # cave_huge: slack of its exact pruning; k_cave_voxels: kCaveFaMax = 0.9375 * MM_SIMPLEX3_BOUND (the octave amplitudes of fbm3<4> sum to
    # 0.9375), 1e30 = "no bound" outside the pruning domain (the bounds themselves are macros of mm_noise.cuh)
    # ... and 18: cave_huge tests the gradient tables' domain (|argument| < 2^18) once for its four octaves
    "chunk.cu::shouldGenerateCaveAtBlock": (float(np.float32(0.001)), 0.9375, float(np.float32(1e30)), 18.0,
                                            31.0, 32.0, 144.0, 383.0),      # bit positions in 32-bit words; the 144-voxel walk; the closed-form analytic bits
    # cave_biome: fbm3From3's component offsets (rng.hpp:188-191, rolled into the loop) and 0.875 = the octave amplitudes of fbm2<3>
    "biomeFuncs.hpp::getCaveBiomeNoise": tuple(float(np.float32(v)) for v in (0.875, 5923.45, 4129.42, 5790.48, 1765.68, 4704.36, 5692.12)),
}
DEVICE_ALLOW = {
    "rng.hpp::worley2": (0.5,),                                        # the edge distance (d2 - d1) * 0.5 is formed by the callers from the returned pair
    "rng.hpp::worley3": (0.5,),
    "featurePlacement.hpp::placeFeature": (0.0, 1.0),
    "featurePlacement.hpp::placeCaveFeature": (0.0, 1.0),
    "rng.hpp::makeSeededRandomEngine": (1.0, 31.0),                    # 0x80000000u
    "chunk.cu::placeDecorators": (16.0,),                                # the 16 x 16 column loop is the lane index here
    "chunk.cu::generateColumnFeaturePlacements": (16.0,),
}


# The sections cut out of the reference's kernels and host stages (tools/extract_ref_literals.py::ANCHORS and the plain functions beside
# them) write no constant of the world definition: their numbers are index strides (16, 256, 384), loop bounds (8 neighbours, 18 x 18
# gathered heights) and CUDA thread-block geometry (which thread of a 32 x 32 block loads which cell of the 34 x 34 shared tile).  The
# device kernels index and tile differently (lane = column, 44 x 44 LDS tiles, chunk lists), so there is no set of constants to hold
# them to; their ARITHMETIC is pinned statement for statement in the oracle (digests below) and the device is held to the oracle by
# the bit-exact stage tests (tests/test_gpu_parity.py: layers, one erosion zone incl. its pass count, fill, gathered lists and bounds).
DEVICE_NO_OWN_CONSTANTS = {"chunk.cu::kernGenerateHeightfield", "chunk.cu::kernGenerateLayers", "chunk.cu::kernDoErosion.stage", "chunk.cu::kernDoErosion.relax",
                           "chunk.cu::copyLayers", "chunk.cu::fixBackwardStratifiedLayers", "chunk.cu::kernFill", "chunk.cu::heightBoundsMinMax",
                           "chunk.cu::Chunk.fill.lists",
                           # the mesh build: the device mesher works on 64-voxel words of bit masks and packed quad records, not on
                           # the reference's loops - its own constants are mostly layout; the world constants it shares with createVBOs are
                           # held by test_device_mesher_holds_the_reference_constants below, its bytes by tests/test_mesh.py
                           "chunk.cu::createVBOs", "chunk.cu::createVBOs.xShapedPosOffset", "chunk.cu::createVBOs.xShapedVertPositions",
                           "chunk.cu::createVBOs.xShapedFaceNormals", "chunk.cu::createVBOs.directionVertPositions", "chunk.cu::createVBOs.uvOffsets"}


def test_device_sections_hold_every_reference_constant():
    problems = []
    covered = set()
    # one device function often restates several functions of the reference (cave_biome_t = getCaveBiomeNoise + getCaveBiomeWeight +
    # getCaveBiome, column_placements = generateColumnFeaturePlacements + tryGenerateCaveFeaturePlacement, ...): a value of the device
    # code is an "extra" only if none of the reference functions mapped onto that device function contains it
    shared_want = {}
    for ref_fn, (files, sigs, prefix) in DEVICE_MAP.items():
        vals = set(v for k in REF if k == ref_fn or k.startswith(ref_fn + "::") for v in REF[k])
        for sig in sigs:
            shared_want.setdefault(sig, set()).update(vals)
    for ref_fn, (files, sigs, prefix) in DEVICE_MAP.items():
        text = _read(files)
        ref_keys = [k for k in REF if k == ref_fn or k.startswith(ref_fn + "::")]
        if not ref_keys:
            continue                                            # the reference writes no number in this function
        covered.update(ref_keys)
        if prefix is None:
            code = ""
            for sig in sigs:
                secs = sections(text, sig, ())
                assert secs, (ref_fn, sig)
                code += secs[""]
            ours = set(literals(code))
            want = set(v for k in ref_keys for v in REF[k])
            allow = DEVICE_ALLOW.get(ref_fn, ())
            missing = sorted(v for v in want - ours if not any(abs(v - a) <= 1e-6 * max(1.0, abs(a)) for a in allow))
            if missing:
                problems.append(f"{ref_fn} -> {sigs}: values of the reference missing from the device code: {missing}")
            shared = set().union(*(shared_want[sig] for sig in sigs))
            extra = sorted(ours - want - shared - DEVICE_BENIGN - set(v for fn in DEVICE_MAP if set(DEVICE_MAP[fn][1]) & set(sigs) for v in DEVICE_EXTRA.get(fn, ())))
            if extra:
                problems.append(f"{ref_fn}: values in the device code that the reference's function does not contain: {extra}")
            continue
        secs = {}
        for sig in sigs:
            for name, code in sections(text, sig, (prefix,)).items():
                secs[name] = secs.get(name, "") + code
        common = literals(secs.get("", ""))                     # preamble / helper functions count for every case
        for k in ref_keys:
            case = k[len(ref_fn) + 2:] if k != ref_fn else ""
            if case not in secs:
                problems.append(f"{k}: case not found in the device code")
                continue
            ours = set(literals(secs[case])) | set(common)
            allow = DEVICE_ALLOW.get(k, DEVICE_ALLOW.get(ref_fn, ()) if case == "" else ())
            missing = sorted(v for v in set(REF[k]) - ours if not any(abs(v - a) <= 1e-6 * max(1.0, abs(a)) for a in allow))
            if missing:
                problems.append(f"{k}: values of the reference missing from the device code: {missing}")
            extra = sorted(set(literals(secs[case])) - set(REF[k]) - DEVICE_BENIGN - set(DEVICE_EXTRA.get(k, ())))
            if extra:
                problems.append(f"{k}: values in the device code that the reference's section does not contain: {extra}")
    not_mapped = sorted(k for k in REF if k not in covered and not k.startswith("chunk.cu::kernGenerateCaves") and k not in DEVICE_NO_OWN_CONSTANTS)
    assert not not_mapped, f"reference sections without a device counterpart in DEVICE_MAP: {not_mapped}"
    assert not problems, "\n".join(problems)


def test_device_mesher_holds_the_reference_constants():
    """Every number Chunk::createVBOs writes (chunk.cu:1778-2003: the 0.4 jitter, the 0.0625 tile size, the half-block centre, 16 / 384 ...)
    appears in the device mesher, and its X-shaped offset is the correctly rounded 0.5 * sin(45 degrees) of chunk.cu:1753."""
    ours = set(literals(_read([CS + "mmgen_mesh.hip", CS + "mm_blockdata.cuh"])))
    missing = sorted(set(REF["chunk.cu::createVBOs"]) - ours)
    assert not missing, missing
    assert REF["chunk.cu::createVBOs.xShapedPosOffset"] == [0.5, 45.0]
    assert float(np.float32(0.5 * np.sin(np.radians(45.0)))) in ours


# ---------------------------------------------------------------------------------------------------------------------------------
# statement skeletons: tests/golden/ref_skeletons.json holds one SHA-256 per reference section (the digest of its normalised token stream,
# tools/extract_ref_literals.py::skeleton; a digest, not text).  Where the oracle's same-named section has the same digest it IS the
# reference's code statement for statement - control flow, operation order, operands - up to the documented table of renamed helpers.
SKEL = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_skeletons.json")))
# sections whose oracle form differs in more than names (each with its reason); everything else must be token-identical.  Empty since
# round 4: the last one was the CUDA kernel kernGenerateCaves, which the oracle now states phase by phase like the reference (one function per
# barrier-separated piece; the block's threads are run through each in ascending y, which also fixes the order of its atomicAdd).
SKELETON_DIFFERS = set()
# What the normaliser (tools/extract_ref_literals.py::skeleton) treats as equal, all of it listed there: comments, qualifiers
# (const / static / inline / __device__ / __host__), OPTIONAL braces (canonical_blocks: every control statement's body gets exactly one
# pair, scope-only braces go - which statements a condition or loop governs IS part of the digest), (float) casts, namespaces (glm:: thrust:: std::), printf diagnostics,
# `default: break;`, `(void)x;`, `#pragma unroll`, the reference's own compile-time switches resolved as its `#define`s set them, the
# table of renamed helpers (g_* glm look-alikes, mm_* libm, hash_u32, Rng = default_random_engine), and four named oracle-only
# insertions: vec3_ltr (= vec3 with the canonical left-to-right evaluation of its arguments), CANONICAL_RETURN_FALSE,
# CANONICAL_DECORATOR_RANGE, CANONICAL_NO_LAYER_FOUND (where the reference's behaviour is undefined).


def _skeleton_report():
    cache, identical, differs = {}, [], []
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        base = os.path.basename(rel)
        if base not in cache:
            cache[base] = _read(ORACLE_FILES[base])
        secs = sections(cache[base], oracle_signature(f"{base}::{key}", sig), prefixes, oracle_anchors(key))
        for k in SKEL:
            parts = k.split("::")
            if parts[0] != base or parts[1] != key:
                continue
            mine = skeleton_digest(secs.get(parts[2] if len(parts) == 3 else "", ""))
            (identical if mine["sha256"] == SKEL[k]["sha256"] else differs).append(k)
    return identical, differs


def test_oracle_sections_are_token_identical_to_the_reference():
    identical, differs = _skeleton_report()
    unexpected = sorted(set(differs) - SKELETON_DIFFERS)
    assert not unexpected, "oracle sections that no longer match the reference's statement skeleton: " + ", ".join(unexpected)
    stale = sorted(SKELETON_DIFFERS - set(differs))
    assert not stale, "sections listed as different that are identical now (move them out of SKELETON_DIFFERS): " + ", ".join(stale)
    # what is pinned this way: every function and switch case on the path, and the arithmetic written inline in the kernels / host stages
    # (ANCHORS), kernGenerateCaves included
    assert len(SKEL) >= 148 and len(identical) == len(SKEL)
    count = lambda prefix: sum(k.startswith(prefix) for k in identical)
    assert count("biomeFuncs.hpp::getHeight::") == 24
    assert count("featurePlacement.hpp::placeFeature::") == 21 and count("featurePlacement.hpp::placeCaveFeature::") == 10
    assert count("rng.hpp::") == 24 and count("chunk.cu::") == 30 and count("chunk.cu::createVBOs") == 6 and count("biomeFuncs.hpp::") == 51 and count("featurePlacement.hpp::") == 43
    assert all(SKEL[k]["tokens"] > 0 for k in SKEL)


def test_block_structure_is_part_of_the_digest():
    """optional braces are canonical, block scope is not: `if (c) { a; b; }` and `if (c) a; b;` must differ (round 3's normaliser dropped
    every brace and could not tell them apart)"""
    sk = lambda code: " ".join(skeleton(code))
    assert sk("if (c) a = 1;") == sk("if (c) { a = 1; }")
    assert sk("if (c) { a = 1; b = 2; }") != sk("if (c) a = 1; b = 2;")
    assert sk("if (c) { a = 1; } b = 2;") == sk("if (c) a = 1; b = 2;")
    assert sk("for (int i = 0; i < n; ++i) { x += i; y += i; }") != sk("for (int i = 0; i < n; ++i) x += i; y += i;")
    assert sk("if (a) x = 1; else if (b) x = 2; else x = 3;") == sk("if (a) { x = 1; } else { if (b) { x = 2; } else { x = 3; } }")
    assert sk("if (a) { if (b) x = 1; } else x = 2;") != sk("if (a) if (b) x = 1; else x = 2;")         # the dangling else binds to the inner if
    assert sk("{ int t = f(); g(t); }") == sk("int t = f(); g(t);")                                       # a scope governs nothing
    assert sk("while (c) { a(); } b();") != sk("while (c) { a(); b(); }")
    assert sk("do { a(); } while (c); b();") != sk("do { a(); b(); } while (c);")
    assert sk("switch (k) { case 1: a(); break; case 2: b(); }") != sk("switch (k) { case 1: a(); break; } case 2: b();")
    assert sk("float v[3] = { 1.f, 2.f, 3.f };") != sk("float v[3] = 1.f, 2.f, 3.f;")                    # braces inside an expression stay
    assert canonical_blocks("if ( c ) return ; }".split()) == "if ( c ) { return ; }".split()               # stray closer of a cut section


# one mutation per family of sections: a change of operation order, of an operand or of control flow in the oracle must break the digest
MUTATIONS = [
    ("oracle/mmo_biome.cpp", "136.f + 6.f * fbm", "6.f * fbm + 136.f", "biomeFuncs.hpp::getHeight::"),                      # operand order
    ("oracle/mmo_features.cpp", "float icebergCenterRatio = 1.f - (horizontalDistance / icebergRadius);",
     "float icebergCenterRatio = (icebergRadius - horizontalDistance) / icebergRadius;", "featurePlacement.hpp::placeFeature::ICEBERG"),
    ("oracle/mmo_features.cpp", "vec3_ltr(u11(featureRng), u01(featureRng), u11(featureRng)) * vec3(2.5f, 3.5f, 2.5f)",
     "vec3_ltr(u11(featureRng), u11(featureRng), u01(featureRng)) * vec3(2.5f, 3.5f, 2.5f)", "featurePlacement.hpp::placeFeature::CORAL"),   # draw order
    ("oracle/mmo_features.cpp", "if (radiusRatio < 0.4f) *blockPtr = Block::GLOWSTONE;", "if (radiusRatio <= 0.4f) *blockPtr = Block::GLOWSTONE;",
     "featurePlacement.hpp::placeCaveFeature::CRYSTAL_PILLAR"),                                                                  # comparison
    ("oracle/mmo_noise.h", "g_dot(v, vec3(654.37f, 560.45f, 747.42f)),\n        g_dot(v, vec3(640.88f, 151.81f, 674.81f))",
     "g_dot(v, vec3(640.88f, 151.81f, 674.81f)),\n        g_dot(v, vec3(654.37f, 560.45f, 747.42f))", "rng.hpp::rand3From3"),        # component order
    ("oracle/mmo_noise.h", "return minDist3 / minDist1 - 1.f;", "return minDist3 / (minDist1 - 1.f);", "rng.hpp::specialCaveNoise"),
    ("oracle/mmo_noise.h", "int h = hash_u32((1 << 31) | (x << 22) | (y << 11) | w) ^ hash_u32(z);",
     "int h = hash_u32((1 << 31) | (x << 22) | (z << 11) | w) ^ hash_u32(y);", "rng.hpp::makeSeededRandomEngine"),
    ("oracle/mmo_stages.cpp", "+ ivec2(gridCellPadding)\n        + ivec2(g_floor(randPos * (float)gridCellInternalSideLength));",
     "+ ivec2(g_floor(randPos * (float)gridCellInternalSideLength + (float)gridCellPadding));", "chunk.cu::isFeaturePos"),
    ("oracle/mmo_stages.cpp", "if (!placedTop && (topRand -= gen.chance) < 0.f)", "if (!placedTop && (topRand -= gen.chance) <= 0.f)", "chunk.cu::placeDecorators"),
    ("oracle/mmo_stages.cpp", "if (u01(blockRng) >= featureGen.chancePerGridCell)", "if (u01(blockRng) > featureGen.chancePerGridCell)",
     "chunk.cu::generateColumnFeaturePlacements"),
    ("oracle/mmo_stages.cpp", "if (layerStart <= y && y < layerEnd)", "if (layerStart < y && y <= layerEnd)", "chunk.cu::chunkFillPlaceBlock"),
    ("oracle/mmo_biome.cpp", "if (rand <= 0.f) return caveBiome;", "if (rand < 0.f) return caveBiome;", "biomeFuncs.hpp::getCaveBiome"),
    # ---- block structure: a statement moved out of the block its condition governs (same tokens, different braces)
    ("oracle/mmo_stages.cpp", "            shared_didChange = true;\n\n            accumulatedHeights[globalIdx2d] += newLayerStart - thisLayerStart;\n        }",
     "            shared_didChange = true;\n        }\n            accumulatedHeights[globalIdx2d] += newLayerStart - thisLayerStart;\n", "chunk.cu::kernDoErosion.relax"),
    ("oracle/mmo_stages.cpp", "            caveBottomDepth = -384;\n            break;\n        }", "            caveBottomDepth = -384;\n        }\n            break;\n",
     "chunk.cu::chunkFillPlaceBlock"),
    # ---- the arithmetic written inline in the kernels and host stages
    ("oracle/mmo_stages.cpp", "if (weight > 0.f)\n        {\n            height += weight * getHeight(biome, worldPos);", "if (weight >= 0.f)\n        {\n            height += weight * getHeight(biome, worldPos);",
     "chunk.cu::kernGenerateHeightfield"),
    ("oracle/mmo_stages.cpp", "fabsf(neighborHeight - maxHeight) * (i % 2 == 1 ? SQRT_2 : 1)", "fabsf(neighborHeight - maxHeight) * (i % 2 == 0 ? SQRT_2 : 1)", "chunk.cu::kernGenerateLayers"),
    ("oracle/mmo_stages.cpp", "neighborLayerStart - tanAngleOfRepose * (i % 2 == 1 ? SQRT_2 : 1)", "neighborLayerStart + tanAngleOfRepose * (i % 2 == 1 ? SQRT_2 : 1)",
     "chunk.cu::kernDoErosion.relax"),
    ("oracle/mmo_stages.cpp", "loadPos = g_clamp(loadPos, 0, EROSION_GRID_SIDE_LENGTH_BLOCKS - 1);", "loadPos = g_clamp(loadPos, 1, EROSION_GRID_SIDE_LENGTH_BLOCKS - 1);",
     "chunk.cu::kernDoErosion.stage"),
    ("oracle/mmo_stages.cpp", "chunkBlockPos = (ivec2(chunkX, chunkZ) + ivec2(ZONE_SIZE / 2)) * 16;", "chunkBlockPos = (ivec2(chunkX, chunkZ) + ivec2(ZONE_SIZE / 4)) * 16;", "chunk.cu::copyLayers"),
    ("oracle/mmo_stages.cpp", "columnLayers[layerIdx256] = erodedStartHeights[idx2d] - columnLayers[layerIdx256];", "columnLayers[layerIdx256] = columnLayers[layerIdx256] - erodedStartHeights[idx2d];",
     "chunk.cu::fixBackwardStratifiedLayers"),
    ("oracle/mmo_stages.cpp", "if (block != Block::AIR && !featurePlacement.canReplaceBlocks)", "if (block != Block::AIR || !featurePlacement.canReplaceBlocks)", "chunk.cu::kernFill"),
    ("oracle/mmo_stages.cpp", "in[0] = g_min(in[0], v[0]);", "in[0] = g_max(in[0], v[0]);", "chunk.cu::heightBoundsMinMax"),
    ("oracle/mmo_stages.cpp", "if (numFeaturePlacements < MAX_GATHERED_FEATURES_PER_CHUNK)", "if (numFeaturePlacements <= MAX_GATHERED_FEATURES_PER_CHUNK)", "chunk.cu::Chunk.fill.lists"),
    # ---- kernGenerateCaves, one per phase that computes something
    ("oracle/mmo_stages.cpp", "int isNextFilled = y < 383 ? shared_isFilled[y + 1] : 0;", "int isNextFilled = y < 383 ? shared_isFilled[y + 1] : 1;", "chunk.cu::kernGenerateCaves.flips"),
    ("oracle/mmo_stages.cpp", "storeIdx += (storeIdx >> 1);", "storeIdx += (storeIdx >> 2);", "chunk.cu::kernGenerateCaves.store"),
    ("oracle/mmo_stages.cpp", "if (srcLane < y)\n            {\n                startStoreIdx += srcLaneNumFlips;", "if (srcLane <= y)\n            {\n                startStoreIdx += srcLaneNumFlips;",
     "chunk.cu::kernGenerateCaves.store"),
    ("oracle/mmo_stages.cpp", "getCaveBiome(ivec3(worldBlockPos2d.x, caveLayer.end + 1, worldBlockPos2d.y), shared_maxHeight, 4982921)",
     "getCaveBiome(ivec3(worldBlockPos2d.x, caveLayer.end, worldBlockPos2d.y), shared_maxHeight, 4982921)", "chunk.cu::kernGenerateCaves.biomes"),
    ("oracle/mmo_stages.cpp", "int isThisFilled = shouldGenerateCaveAtBlock(worldPos, shared_maxHeight, shared_oceanAndBeachWeight) ? 0 : 1;",
     "int isThisFilled = shouldGenerateCaveAtBlock(worldPos, shared_maxHeight, shared_oceanAndBeachWeight) ? 1 : 0;", "chunk.cu::kernGenerateCaves.filled"),
    ("oracle/mmo_stages.cpp", "if (y < numOceanAndBeachBiomes)\n    {\n        float biomeWeight = biomeWeights[devBiomeWeightsSize * chunkIdx + 256 * y + idx2d];",
     "if (y <= numOceanAndBeachBiomes)\n    {\n        float biomeWeight = biomeWeights[devBiomeWeightsSize * chunkIdx + 256 * y + idx2d];", "chunk.cu::kernGenerateCaves.weights"),
    # ---- Chunk::createVBOs and its static tables: the culling rule, the neighbour a face looks at, the uv draw order, a table entry,
    # and a statement moved out of the block its condition governs
    ("oracle/mmo_mesh.cpp", "shouldDisplay = neighborBlock == Block::AIR || neighborTrans == TransparencyType::T_SEMI_TRANSPARENT;",
     "shouldDisplay = neighborBlock == Block::AIR && neighborTrans == TransparencyType::T_SEMI_TRANSPARENT;", "chunk.cu::createVBOs"),
    ("oracle/mmo_mesh.cpp", "neighborPosChunk = neighbors[3];\n                            neighborPos.x += 16;", "neighborPosChunk = neighbors[1];\n                            neighborPos.x += 16;",
     "chunk.cu::createVBOs"),
    ("oracle/mmo_mesh.cpp", "if (sideUv.randRot)\n                        {\n                            uvStartIdx = (int)u04(rng);\n                        }\n                        if (sideUv.randFlip)\n                        {\n                            uvFlipIdx = (int)u04(rng);\n                        }",
     "if (sideUv.randFlip)\n                        {\n                            uvFlipIdx = (int)u04(rng);\n                        }\n                        if (sideUv.randRot)\n                        {\n                            uvStartIdx = (int)u04(rng);\n                        }", "chunk.cu::createVBOs"),
    ("oracle/mmo_mesh.cpp", "                            if (uvFlipIdx & 2)\n                            {\n                                uvOffset.y = 1 - uvOffset.y;\n                            }\n                        }\n                        vert.uv = vec2(sideUv.uv + uvOffset) * 0.0625f;",
     "                            if (uvFlipIdx & 2)\n                            {\n                                uvOffset.y = 1 - uvOffset.y;\n                            }\n                        vert.uv = vec2(sideUv.uv + uvOffset) * 0.0625f;\n                        }", "chunk.cu::createVBOs"),
    ("oracle/mmo_mesh.cpp", "ivec3(1, 0, 1), ivec3(1, 0, 0), ivec3(1, 1, 0), ivec3(1, 1, 1),", "ivec3(1, 0, 1), ivec3(1, 0, 0), ivec3(1, 1, 1), ivec3(1, 1, 0),",
     "chunk.cu::createVBOs.directionVertPositions"),
    ("oracle/mmo_mesh.cpp", "0.5f * sinf(g_radians(45.f))", "0.5f * cosf(g_radians(45.f))", "chunk.cu::createVBOs.xShapedPosOffset"),
]


@pytest.mark.parametrize("path,old,new,prefix", MUTATIONS, ids=[f"{i}-" + m[3].split("::", 1)[1] for i, m in enumerate(MUTATIONS)])
def test_skeleton_digest_sees_a_mutated_oracle_statement(path, old, new, prefix, monkeypatch):
    """the digests are not vacuous: each listed one-statement change of the oracle's TEXT (never built) moves a section of that family
    out of the identical set"""
    full = os.path.join(ROOT, path)
    text = open(full).read()
    assert text.count(old) >= 1, f"mutation target not found in {path}: {old!r}"
    real_open = open

    def fake_read(paths):
        return "\n".join(strip_comments(text.replace(old, new, 1) if os.path.join(ROOT, p) == full else real_open(os.path.join(ROOT, p)).read())
                         for p in paths)

    monkeypatch.setitem(globals(), "_read", fake_read)
    _, differs = _skeleton_report()
    hit = [k for k in differs if k.startswith(prefix)]
    assert hit, f"mutating {old!r} -> {new!r} left every {prefix}* digest unchanged"
