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
from extract_ref_literals import REFERENCE_SECTIONS, literals, sections, skeleton_digest, strip_comments   # noqa: E402

REF = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_literals.json")))

ORACLE_FILES = {"biomeFuncs.hpp": ["oracle/mmo_biome.cpp"], "featurePlacement.hpp": ["oracle/mmo_features.cpp"],
                "rng.hpp": ["oracle/mmo_noise.h"], "chunk.cu": ["oracle/mmo_stages.cpp"]}


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
        sig_o = sig.replace(r"void\s+Chunk::", r"void\s+").replace(r"void\s+kernGenerateCaves", r"void\s+generateCaves")
        secs = sections(cache[base], sig_o, prefixes)
        for name, code in secs.items():
            out[f"{base}::{key}" + (f"::{name}" if name else "")] = literals(code)
    return out


# values the reference writes that our restatement legitimately does not (each with its reason)
ORACLE_ALLOW = {
    "chunk.cu::kernGenerateCaves": (12.0, 32.0, 383.0, 4095.0),       # thread-block geometry of the CUDA kernel (12 threads, 32-layer shared arrays, y = 383 - 32 k, 0xFFF flip mask)
    "featurePlacement.hpp::placeFeature": (0.0, 1.0),        # preamble before the switch: vec3 / ivec3 helpers
    "featurePlacement.hpp::placeCaveFeature": (1.0,),
    "rng.hpp::makeSeededRandomEngine": (1.0, 31.0),           # (1 << 31) is written 0x80000000u
}
ORACLE_EXTRA = {
    "chunk.cu::kernGenerateCaves": (2.0, 385.0),              # isFilled[385] + the flip loop replace the shared-memory bit words
    "rng.hpp::makeSeededRandomEngine": (2147483648.0,),       # 0x80000000u = the reference's (1 << 31)
}
ORACLE_SIGNATURES = {"rng.hpp::hash": r"uint32_t\s+hash_u32\s*\([^)]*\)\s*\{"}


def test_oracle_sections_hold_every_reference_constant():
    ours = _oracle_sections()
    for key, sig in ORACLE_SIGNATURES.items():
        base = key.split("::")[0]
        secs = sections(_read(ORACLE_FILES[base]), sig, ())
        if secs:
            ours[key] = literals(secs[""])
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
    "biomeFuncs.hpp::getCaveBiomeNoise": ([CS + "mm_biome.cuh"], [r"int\s+cave_biome_t\s*\([^)]*\)\s*\{"], None),
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
    "chunk.cu::shouldGenerateCaveAtBlock": ([CS + "mmgen_kernels.hip"], [r"\bk_cave_columns\s*\([^)]*\)\s*\{", r"\bk_cave_voxels\s*\([^{]*\)\s*\{", r"float\s+cave_huge\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::getStratifiedMaterialThickness": ([CS + "mmgen_kernels.hip"], [r"float\s+stratified_thickness\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::isFeaturePos": ([CS + "mmgen_features.hip"], [r"bool\s+is_feature_pos\s*\([^)]*\)\s*\{"], None),
    "chunk.cu::generateColumnFeaturePlacements": ([CS + "mmgen_features.hip"], [r"void\s+column_placements\s*\([^{]*\)\s*\{"], None),
    "chunk.cu::placeDecorators": ([CS + "mmgen_features.hip"], [r"\bk_decorators\s*\([^{]*\)\s*\{"], None),
}
# values the device code may contain beyond the reference's: loop / layout constants of the scalarised, lane-mapped formulation
DEVICE_BENIGN = {0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 9.0, 16.0, 63.0, 64.0, 255.0, 256.0, 384.0, 65535.0, 2147483648.0, 48271.0,
                 float(np.float32(3.402823466e+38))}
DEVICE_EXTRA = {
    "biomeFuncs.hpp::getBiomeNoise": (float(np.float32(0.32)),),         # overallBiomeScale, a file-level constant in the reference (biomeFuncs.hpp:105)
    # cave_huge: slack of its exact pruning; k_cave_voxels: kCaveFaMax = 0.9375 * MM_SIMPLEX3_BOUND (the octave amplitudes of fbm3<4> sum to
    # 0.9375), 1e30 = "no bound" outside the pruning domain (the bounds themselves are macros of mm_noise.cuh)
    "chunk.cu::shouldGenerateCaveAtBlock": (float(np.float32(0.001)), 0.9375, float(np.float32(1e30))),
    # cave_biome: fbm3From3's component offsets (rng.hpp:188-191, rolled into the loop) and 0.875 = the octave amplitudes of fbm2<3>
    "biomeFuncs.hpp::getCaveBiomeNoise": tuple(float(np.float32(v)) for v in (0.875, 5923.45, 4129.42, 5790.48, 1765.68, 4704.36, 5692.12)),
}
DEVICE_ALLOW = {
    "featurePlacement.hpp::placeFeature": (0.0, 1.0),
    "featurePlacement.hpp::placeCaveFeature": (0.0, 1.0),
    "rng.hpp::makeSeededRandomEngine": (1.0, 31.0),                    # 0x80000000u
    "chunk.cu::placeDecorators": (16.0,),                                # the 16 x 16 column loop is the lane index here
    "chunk.cu::generateColumnFeaturePlacements": (16.0,),
}


def test_device_sections_hold_every_reference_constant():
    problems = []
    covered = set()
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
            extra = sorted(ours - want - DEVICE_BENIGN - set(DEVICE_EXTRA.get(ref_fn, ())))
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
    not_mapped = sorted(k for k in REF if k not in covered and not k.startswith("chunk.cu::kernGenerateCaves"))
    assert not not_mapped, f"reference sections without a device counterpart in DEVICE_MAP: {not_mapped}"
    assert not problems, "\n".join(problems)


# ---------------------------------------------------------------------------------------------------------------------------------
# statement skeletons: tests/golden/ref_skeletons.json holds one SHA-256 per reference section (the digest of its normalised token stream,
# tools/extract_ref_literals.py::skeleton; a digest, not text).  Where the oracle's same-named section has the same digest it IS the
# reference's code statement for statement - control flow, operation order, operands - up to the documented table of renamed helpers.
SKEL = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_skeletons.json")))
# sections whose oracle form differs in more than names (each group with its reason); everything else must be token-identical
SKELETON_DIFFERS = {
    # vector expressions written per component, explicit evaluation order of constructor arguments that draw from the stream
    # (float r0 = u11(rng), r1 = ..., r2 = ...; vec3(r0, r1, r2)), closures u01f() / u01b() for u01(featureRng) / u01(blockRng), integer
    # literals in float vectors written as floats, ivec helpers: every rasteriser
    "featurePlacement.hpp::placeFeature", "featurePlacement.hpp::placeCaveFeature", "featurePlacement.hpp::sdCappedCylinder",
    "featurePlacement.hpp::isInRasterizedLine", "featurePlacement.hpp::isInCrystal", "featurePlacement.hpp::getCrystalRadius",
    "featurePlacement.hpp::getRandomCrystalBlock",
    *("featurePlacement.hpp::placeFeature::" + f for f in (
        "SPHERE", "CORAL", "KELP", "ICEBERG", "ACACIA_TREE", "REDWOOD_TREE", "CYPRESS_TREE", "BIRCH_TREE", "PINE_TREE", "PINE_SHRUB",
        "RAFFLESIA", "LARGE_JUNGLE_TREE", "SMALL_JUNGLE_TREE", "TINY_JUNGLE_TREE", "MEDIUM_PURPLE_MUSHROOM", "PURPLE_MUSHROOM",
        "MEDIUM_CRYSTAL", "CRYSTAL", "PALM_TREE", "CACTUS")),
    *("featurePlacement.hpp::placeCaveFeature::" + f for f in (
        "CAVE_VINE", "GLOWSTONE_CLUSTER", "STORMLIGHT_SPHERE", "CEILING_STORMLIGHT_SPHERE", "CRYSTAL_PILLAR", "WARPED_FUNGUS", "AMBER_FUNGUS")),
    # the sin hashes return fract(sin(v) * c) per component; the engine is our own Rng over uint32_t; specialCaveNoise floors explicitly
    "rng.hpp::hash", "rng.hpp::makeSeededRandomEngine", "rng.hpp::rand2From2", "rng.hpp::rand2From3", "rng.hpp::rand3From2",
    "rng.hpp::rand3From3", "rng.hpp::specialCaveNoise",
    # host / kernel orchestration restated over plain arrays (tables through T(), no shared memory, no thread indices)
    "chunk.cu::getStratifiedMaterialThickness", "chunk.cu::isFeaturePos", "chunk.cu::generateColumnFeaturePlacements",
    "chunk.cu::placeDecorators", "chunk.cu::kernGenerateCaves",
    # isInRange() written as two comparisons on floats, explicit parentheses around int -> float operands, an unused local dropped
    "biomeFuncs.hpp::caveBiomeBlockPostProcess", "biomeFuncs.hpp::caveBiomeBlockPostProcess::CRYSTAL_CAVES",
    "biomeFuncs.hpp::caveBiomeBlockPostProcess::LUSH_CAVES", "biomeFuncs.hpp::caveBiomeBlockPostProcess::AMBER_FOREST",
    "biomeFuncs.hpp::biomeBlockPostProcess::CRYSTALS",
}


def test_oracle_sections_are_token_identical_to_the_reference():
    cache, identical, differs = {}, [], []
    for rel, key, sig, prefixes in REFERENCE_SECTIONS:
        base = os.path.basename(rel)
        if base not in cache:
            cache[base] = _read(ORACLE_FILES[base])
        sig_o = sig.replace(r"void\s+Chunk::", r"void\s+").replace(r"void\s+kernGenerateCaves", r"void\s+generateCaves")
        secs = sections(cache[base], sig_o, prefixes)
        for k in SKEL:
            parts = k.split("::")
            if parts[0] != base or parts[1] != key:
                continue
            mine = skeleton_digest(secs.get(parts[2] if len(parts) == 3 else "", ""))
            (identical if mine["sha256"] == SKEL[k]["sha256"] else differs).append(k)
    unexpected = sorted(set(differs) - SKELETON_DIFFERS)
    assert not unexpected, "oracle sections that no longer match the reference's statement skeleton: " + ", ".join(unexpected)
    stale = sorted(SKELETON_DIFFERS - set(differs))
    assert not stale, "sections listed as different that are identical now (move them out of SKELETON_DIFFERS): " + ", ".join(stale)
    # what is pinned this way: every getHeight case, both biome-noise functions, the surface block rules, fbm / simplex-from helpers, ...
    assert len(identical) >= 53 and sum(k.startswith("biomeFuncs.hpp::getHeight::") for k in identical) == 24
