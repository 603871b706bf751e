#!/usr/bin/env python3
"""tests/golden/tile_checksums.json: the checksum (mega-minecraft_amd.distributed.tile_checksum) of every tile of bench.py's layouts at
N = 1, 2, 4, 8 GPUs, taken from each layout's world generated as ONE region on one GPU (mmgen_region_generate).  A multi-GPU run holds
every rank's tile to these (bench.py: tiles_bit_exact), the config-5 test holds the tiled path on one GPU to them.  Run on a GPU box:
    python tools/gen_tile_checksums.py gpurun_out/tile_checksums.json   (then copy it to tests/golden/)"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("mega-minecraft_amd")
d = importlib.import_module("mega-minecraft_amd.distributed")
gen = pkg.MMGen(0)
TILES = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
nx, nz = 64, 128
out = {"_about": "tile_checksum of every tile of bench.py's layouts, from the layout's world generated as one region (tools/gen_tile_checksums.py)"}
for n, (tx, tz) in TILES.items():
    lay = d.TileLayout(-(tx * nx) // 2, -(tz * nz) // 2, tx, tz, nx, nz)
    W, H = tx * nx, tz * nz
    world = gen.generate_region(lay.world_cx0, lay.world_cz0, W, H)["blocks"].view(H, W, 98304)
    sums = []
    for r in range(lay.world_size):
        cx0, cz0, _, _ = lay.region(r)
        tile = world[cz0 - lay.world_cz0:cz0 - lay.world_cz0 + nz, cx0 - lay.world_cx0:cx0 - lay.world_cx0 + nx].reshape(nx * nz, 98304)
        sums.append(f"{d.tile_checksum(tile, torch):016x}")
    out[d.layout_key(lay)] = sums
    print(d.layout_key(lay), sums, flush=True)
    del world
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "tile_checksums.json"), "w"), indent=1)
