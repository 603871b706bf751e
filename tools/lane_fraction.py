import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import importlib, torch
m = importlib.import_module("mega-minecraft_amd"); gen = m.MMGen(0)
for name,(x0,z0,n) in {"config2 (0,0,16)":(0,0,16), "full (-32,-32,32)":(-32,-32,32)}.items():
    coords=[(x0+x,z0+z) for z in range(n) for x in range(n)]
    out = gen.generate_chunks_no_erosion(gen.positions(coords))
    hf = out["hf"]; bw = out["bw"].view(-1,24,256)
    obw = bw[:, :8].sum(1)          # ocean + beach weights (first 8 biomes)
    y = torch.arange(144, device=hf.device).view(1,1,144).float()
    top = torch.clamp(hf.int(), min=128).unsqueeze(-1)
    inband = (y>0) & (y<=top)
    tr = ((y + 50*obw.unsqueeze(-1)) - 142)/(95-142); tr = tr.clamp(0,1)
    need = inband & (tr>0)
    print(name, "needThr fraction of evaluated voxels:", float(need.float().mean()), " mean obw", float(obw.mean()), "mean h", float(hf.mean()))
    # per-wave (64 consecutive voxels of the 4-col groups) any-need fraction
