import sys, time, importlib, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
mm = importlib.import_module('mega-minecraft_amd')
from oracle_binding import Oracle
g = mm.MMGen(0); o = Oracle()
coords = [(0,0),(1,0),(-3,7),(100,-250),(-700,333),(40,41),(1000,1000),(-64,-64)]
pos = g.positions(coords); opos = o.positions(coords)
t=time.time(); out = g.generate_chunks_no_erosion(pos); torch.cuda.synchronize(); print('gpu first', time.time()-t)
t=time.time(); out = g.generate_chunks_no_erosion(pos); torch.cuda.synchronize(); print('gpu second', time.time()-t)
t=time.time()
hf, bw = o.heightfields(opos); gh = o.gather_heightfields(opos, hf); lay = o.fix_backward(o.layers(opos, gh, bw)); cave = o.caves(opos, hf, bw); blocks = o.fill(opos, hf, bw, lay, cave)
print('cpu', time.time()-t)
def cmp(name, a, b):
    a = a.cpu().numpy().reshape(b.shape)
    if a.dtype == np.float32:
        bad = (a.view(np.uint32) != b.view(np.uint32)).sum(); print(name, 'bit mismatches', int(bad), 'of', a.size, 'maxabs', float(np.abs(a-b).max()))
    else:
        bad = (a != b).sum(); print(name, 'mismatches', int(bad), 'of', a.size)
cmp('hf', out['hf'], hf); cmp('bw', out['bw'], bw); cmp('gathered', out['gathered'], gh); cmp('layers', out['layers'], lay)
cmp('cave', out['cave'], cave); cmp('blocks', out['blocks'], blocks)
print(o.ub_counters())
