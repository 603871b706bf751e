"""debug aid: config-4 tiling on one GPU, serial vs DAG schedule; where do tiles differ from the single region?"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
pkg = importlib.import_module("mega-minecraft_amd")
d = importlib.import_module("mega-minecraft_amd.distributed")
import test_gpu_features as T
side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
layout = d.TileLayout(-side, -side, 2, 2, side, side)
for serial in (1, 0):
    os.environ["MMGEN_REGION_SERIAL"] = str(serial)
    world, (wx0, wz0, W, H) = T._tiled_world_on_one_gpu(pkg, layout)
    torch.cuda.synchronize()
    single = pkg.MMGen(0).generate_region(wx0, wz0, W, H)["blocks"]
    torch.cuda.synchronize()
    bad = (world != single).any(dim=1).view(H, W).cpu().numpy()
    print("serial" if serial else "dag", "differing chunks:", int(bad.sum()), "of", H * W)
    if bad.any():
        zs, xs = np.nonzero(bad)
        print("  rows", sorted(set(zs.tolist()))[:40], "cols", sorted(set(xs.tolist()))[:40])
