import importlib, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
pkg = importlib.import_module("mega-minecraft_amd")
gen = pkg.MMGen(0)
if len(sys.argv) > 1 and sys.argv[1] == "region":
    out = gen.generate_region(-60, -60, 120, 120); torch.cuda.synchronize(); del out
packs = []
for z in [(0, 0), (12, 0), (24, 0)]:
    pos = gen.positions(gen.zone_area_coords(*z))
    hf, bw, g = gen.generate_heightfields(pos, gathered=True)
    packs.append(gen.pack_zone_planes(gen.generate_layers(g, bw, pos), hf))
singles = []
for p in packs[:2]:
    t0 = time.time(); out, passes = gen.erode_zones(p.clone()); torch.cuda.synchronize(); print("single", passes, time.time() - t0, flush=True)
    singles.append(out)
for rep in range(6):
  for n in (2,):
    both = torch.cat(packs[:n], dim=0).contiguous()
    t0 = time.time()
    try:
        out, passes = gen.erode_zones(both)
        torch.cuda.synchronize()
        print("batched", n, passes, time.time() - t0, [bool(torch.equal(out[i], singles[i][0])) for i in range(n)], flush=True)
    except Exception as e:
        print("batched", n, "FAILED", time.time() - t0, e, flush=True)
