#!/usr/bin/env python3
"""Extracts the literal rule tables of the reference's BiomeUtils::init (src/terrain/biomeFuncs.hpp:725-1256) and the enum orders of
src/terrain/biome.hpp / block.hpp as DATA (numbers only) into tests/golden/ref_tables.npz.

Run in the build container (it reads /root/reference; nothing under tests/ or on the GPU box does).  The init function cannot be compiled
here (cuda headers), but it is plain literal initialisation: a handful of macros (`biomeWeights(X) = { wP, ... }`, `setMaterialInfoSameBlock`,
`setBiomeMaterialWeight`, `setFeatureHeightBounds`, ...) and brace-initialised gen lists, which this script reads with regular expressions.
The result pins the oracle's tables (tests/test_oracle_math.py::test_tables_match_reference_literals) and the device's constant tables
(tests/test_gpu_parity.py::test_device_tables_match_reference_literals) to the reference's own source text instead of to our reading of it.

Layouts (all float32 unless noted; "present" = 1 for a used slot, unused slots are zero):
  biome_rules  u8 [24][6]   BiomeWeightType of (ocean, beach, rocky, magic, temperature, moisture)   (W_IGNORE / W_POSITIVE / W_NEGATIVE enum values)
  cave_rules   u8 [5][4]    (none, shallow, warped, rocky)
  grass        u8 [24]      Block of BiomeBlocks::grassBlock (default Block::DIRT)
  material_infos  [20][4]   (block, thickness, noise amplitude | angle of repose IN DEGREES as written, noise scale | max slope)
  biome_material_weights [24][20]
  feature_bounds i32 [21][2], cave_feature_bounds i32 [10][2]
  surf_gens    [24][4][11]  (present, feature, cell, padding, chance, canReplace, nTop, material0, minThickness0, material1, minThickness1)
  cave_gens    [5][3][9]    (present, caveFeature, cell, padding, chance, minLayerHeight, canReplace, fromCeiling, canLava)
  deco_gens    [24][7][10], cave_deco_gens [5][6][10]
                            (present, block, chance, nUnder, under0, under1, under2 (sorted ascending), replaceBlock, secondBlock, fromCeiling)
  gather_offsets i32 [49][2]  gatherFeaturePlacementsChunkOffsets (chunk.cu:1158-1167), (dx, dz) in gather order
  enum_counts  i32 [6]      (numBiomes, numCaveBiomes, numMaterials, numFeatures, numCaveFeatures, numBlocks)
"""
import os
import re
import sys

import numpy as np

REF = "/root/reference/src/terrain"


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def parse_enum(text, name):
    m = re.search(r"enum\s+class\s+" + name + r"\b[^{]*\{(.*?)\}", text, flags=re.S)
    assert m, name
    out, v = {}, 0
    for item in m.group(1).split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, e = [s.strip() for s in item.split("=")]
            v = int(e, 0)
        else:
            k = item
        out[k] = v
        v += 1
    return out


def fnum(s):
    s = s.strip()
    return float(s[:-1] if s.endswith("f") else s)


def split_top(s):
    """split at top-level commas (ignoring commas inside () and {})"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({":
            depth += 1
        elif ch in ")}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def gen_items(body, ctor):
    """every `ctor(...)....` item of a brace list, with its chained setters"""
    items = []
    for part in split_top(body):
        if not part.startswith(ctor + "("):
            continue
        depth, i0 = 0, len(ctor)
        for i in range(i0, len(part)):
            if part[i] == "(":
                depth += 1
            elif part[i] == ")":
                depth -= 1
                if depth == 0:
                    items.append((part[i0 + 1:i], part[i + 1:]))
                    break
    return items


def main(out_path):
    biome_hpp = strip_comments(open(os.path.join(REF, "biome.hpp")).read())
    block_hpp = strip_comments(open(os.path.join(REF, "block.hpp")).read())
    funcs = open(os.path.join(REF, "biomeFuncs.hpp")).read()
    init = strip_comments(funcs[funcs.index("void BiomeUtils::init()"):])

    Biome, CaveBiome, Material = parse_enum(biome_hpp, "Biome"), parse_enum(biome_hpp, "CaveBiome"), parse_enum(biome_hpp, "Material")
    Feature, CaveFeature = parse_enum(biome_hpp, "Feature"), parse_enum(biome_hpp, "CaveFeature")
    Block = parse_enum(block_hpp, "Block")
    WT = parse_enum(strip_comments(funcs), "BiomeWeightType") if "enum class BiomeWeightType" in funcs else parse_enum(biome_hpp, "BiomeWeightType")
    wmap = {"wI": WT["W_IGNORE"], "wP": WT["W_POSITIVE"], "wN": WT["W_NEGATIVE"]}
    nb, ncb, nm, nf, ncf = len(Biome), len(CaveBiome), len(Material), len(Feature), len(CaveFeature)
    blk = lambda s: Block[s.strip().replace("Block::", "")]

    biome_rules = np.zeros((nb, 6), np.uint8)
    for m in re.finditer(r"\bbiomeWeights\((\w+)\)\s*=\s*\{([^}]*)\}", init):
        biome_rules[Biome[m.group(1)]] = [wmap[t.strip()] for t in m.group(2).split(",")]
    cave_rules = np.zeros((ncb, 4), np.uint8)
    for m in re.finditer(r"\bcaveBiomeWeights\((\w+)\)\s*=\s*\{([^}]*)\}", init):
        cave_rules[CaveBiome[m.group(1)]] = [wmap[t.strip()] for t in m.group(2).split(",")]

    default_grass = re.search(r"struct\s+BiomeBlocks\s*\{[^}]*grassBlock\s*\{\s*Block::(\w+)", biome_hpp).group(1)
    grass = np.full(nb, Block[default_grass], np.uint8)
    for m in re.finditer(r"host_biomeBlocks\[\(int\)Biome::(\w+)\]\.grassBlock\s*=\s*Block::(\w+)", init):
        grass[Biome[m.group(1)]] = Block[m.group(2)]

    material_infos = np.zeros((nm, 4), np.float32)
    for m in re.finditer(r"\bsetMaterialInfoSameBlock\((\w+),([^,]+),([^,]+),([^)]+)\)", init):
        if m.group(1) == "material":
            continue                                  # the macro's own definition
        material_infos[Material[m.group(1)]] = [Block[m.group(1)], fnum(m.group(2)), fnum(m.group(3)), fnum(m.group(4))]
    for m in re.finditer(r"\bsetMaterialInfo\((\w+),\s*(\w+),([^,]+),([^,]+),([^)]+)\)", init):
        if m.group(1) == "material":
            continue
        material_infos[Material[m.group(1)]] = [Block[m.group(2)], fnum(m.group(3)), fnum(m.group(4)), fnum(m.group(5))]

    bmw = np.ones((nb, nm), np.float32)               # "host_biomeMaterialWeights[i] = 1"
    assert re.search(r"host_biomeMaterialWeights\[i\]\s*=\s*1\s*;", init)
    for m in re.finditer(r"\bsetCurrentBiomeMaterialWeight\((\w+),([^)]+)\)", init):
        if m.group(1) != "material":
            bmw[:, Material[m.group(1)]] = fnum(m.group(2))
    for m in re.finditer(r"\bsetBiomeMaterialWeight\((\w+),\s*(\w+),([^)]+)\)", init):
        if m.group(1) != "biome":
            bmw[Biome[m.group(1)], Material[m.group(2)]] = fnum(m.group(3))

    feature_bounds = np.zeros((nf, 2), np.int32)
    for m in re.finditer(r"\bsetFeatureHeightBounds\((\w+),\s*(-?\d+),\s*(-?\d+)\)", init):
        feature_bounds[Feature[m.group(1)]] = [int(m.group(2)), int(m.group(3))]
    cave_feature_bounds = np.zeros((ncf, 2), np.int32)
    for m in re.finditer(r"\bsetCaveFeatureHeightBounds\((\w+),\s*(-?\d+),\s*(-?\d+)\)", init):
        cave_feature_bounds[CaveFeature[m.group(1)]] = [int(m.group(2)), int(m.group(3))]

    # named initializer lists used as under-block sets
    named = {m.group(1): [blk(t) for t in m.group(2).split(",")] for m in re.finditer(r"auto\s+(\w+)\s*=\s*\{\s*(Block::[^}]*)\}\s*;", init)}

    surf_gens = np.zeros((nb, 4, 11), np.float32)
    for m in re.finditer(r"host_biomeFeatureGens\[\(int\)Biome::(\w+)\]\s*=\s*\{(.*?)\}\s*;", init, flags=re.S):
        for k, (args, tail) in enumerate(gen_items(m.group(2), "FeatureGen")):
            a = split_top(args)
            tops = re.findall(r"\{\s*Material::(\w+)\s*,\s*([^}]+)\}", a[4])
            row = [1, Feature[a[0].replace("Feature::", "")], int(a[1]), int(a[2]), fnum(a[3]), 0 if "setNotReplaceBlocks" in tail else 1, len(tops), 0, 0, 0, 0]
            for j, (mat, thr) in enumerate(tops):
                row[7 + 2 * j], row[8 + 2 * j] = Material[mat], fnum(thr)
            surf_gens[Biome[m.group(1)], k] = row

    cave_gens = np.zeros((ncb, 3, 9), np.float32)
    for m in re.finditer(r"host_caveBiomeFeatureGens\[\(int\)CaveBiome::(\w+)\]\s*=\s*\{(.*?)\}\s*;", init, flags=re.S):
        for k, (args, tail) in enumerate(gen_items(m.group(2), "CaveFeatureGen")):
            a = split_top(args)
            mlh = re.search(r"setMinLayerHeight\((\d+)\)", tail)
            cave_gens[CaveBiome[m.group(1)], k] = [1, CaveFeature[a[0].replace("CaveFeature::", "")], int(a[1]), int(a[2]), fnum(a[3]),
                                                   int(mlh.group(1)) if mlh else 0, 0 if "setNotReplaceBlocks" in tail else 1,
                                                   1 if "setGeneratesFromCeiling" in tail else 0, 1 if "setCanGenerateInLava" in tail else 0]

    def decos(pattern, enum, shape):
        out = np.zeros(shape, np.float32)
        for m in re.finditer(pattern, init, flags=re.S):
            for k, (args, tail) in enumerate(gen_items(m.group(2), "DecoratorGen")):
                a = split_top(args)
                u = a[2].strip()
                under = sorted(named[u] if u in named else [blk(t) for t in u.strip("{} ").split(",") if t.strip()])
                n_under = len(under)
                under = under + [0] * (3 - n_under)
                sec = re.search(r"setSecondDecoratorBlock\(Block::(\w+)\)", tail)
                rep = Block["WATER"] if "setWater()" in tail else Block["AIR"]
                assert "setPossibleReplaceBlocks" not in tail
                out[enum[m.group(1)], k] = [1, blk(a[0]), fnum(a[1]), n_under, under[0], under[1], under[2], rep,
                                            Block[sec.group(1)] if sec else Block["AIR"], 1 if "setGeneratesFromCeiling" in tail else 0]
        return out

    deco_gens = decos(r"host_biomeDecoratorGens\[\(int\)Biome::(\w+)\]\s*=\s*\{(.*?)\}\s*;", Biome, (nb, 7, 10))
    cave_deco_gens = decos(r"host_caveBiomeDecoratorGens\[\(int\)CaveBiome::(\w+)\]\s*=\s*\{(.*?)\}\s*;", CaveBiome, (ncb, 6, 10))

    # gather order of the 49 neighbour lists (chunk.cu:1158-1167)
    chunk_cu = strip_comments(open(os.path.join(REF, "chunk.cu")).read())
    body = re.search(r"gatherFeaturePlacementsChunkOffsets\s*=\s*\{(.*?)\};", chunk_cu, flags=re.S).group(1)
    gather_offsets = np.array([[int(a), int(b)] for a, b in re.findall(r"ivec2\(\s*(-?\d+)\s*,\s*(-?\d+)\s*\)", body)], np.int32)
    assert gather_offsets.shape == (49, 2)

    np.savez_compressed(out_path, gather_offsets=gather_offsets, biome_rules=biome_rules, cave_rules=cave_rules, grass=grass, material_infos=material_infos,
                        biome_material_weights=bmw, feature_bounds=feature_bounds, cave_feature_bounds=cave_feature_bounds, surf_gens=surf_gens,
                        cave_gens=cave_gens, deco_gens=deco_gens, cave_deco_gens=cave_deco_gens,
                        enum_counts=np.array([nb, ncb, nm, nf, ncf, len(Block)], np.int32))
    print(f"wrote {out_path}: {int(surf_gens[..., 0].sum())} surface gens, {int(cave_gens[..., 0].sum())} cave gens, "
          f"{int(deco_gens[..., 0].sum())} + {int(cave_deco_gens[..., 0].sum())} decorator gens, {int((bmw != 1).sum())} non-default material weights")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_tables.npz"))
