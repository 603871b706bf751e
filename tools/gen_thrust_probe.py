#!/usr/bin/env python3
"""Freezes draws of the REAL rocThrust engine (oracle/_ref/libthrustprobe.so, built by `make -C oracle ref` from oracle/thrust_probe.cpp
against /opt/rocm/include/thrust) into tests/golden/thrust_probe.npz:
  seeds u32 [S], raw u32 [S][4] (engine outputs), u01 f32 [S][6]                        the bare engine, incl. seed 0 (-> 1), m - 1, m, 2^32 - 1
  xyzw i32 [N][4], u01_3 f32 [N][4], u01_4 f32 [N][4], seed_3 / seed_4 u32 [N]          makeSeededRandomEngine(x, y, z[, w]) (rng.hpp:86-96) incl.
                                                                                         negative and far coordinates and the seeds the path uses
Tests hold oracle/mmo_math.h's MinStd (CPU) and the device's rng3 / rng4 (GPU) to these vectors."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libthrustprobe.so"))
vp = ctypes.c_void_p
P = lambda a: a.ctypes.data_as(vp)

m = 2 ** 31 - 1
rs = np.random.RandomState(20261002)
seeds = np.array([0, 1, 2, 48271, m - 1, m, m + 1, 2 ** 31, 2 ** 32 - 1, 2 ** 32 - 2] + rs.randint(0, 2 ** 32, 246, dtype=np.uint64).tolist(), dtype=np.uint32)
raw = np.zeros((len(seeds), 4), np.uint32)
u01 = np.zeros((len(seeds), 6), np.float32)
for i, s in enumerate(seeds):
    lib.thrust_minstd_raw(ctypes.c_uint(int(s)), 4, P(raw[i]))
    lib.thrust_minstd_u01(ctypes.c_uint(int(s)), 6, P(u01[i]))

path_seeds = [329828101, 7589341, 1293012, 57847812, 398132, 9322743, 329271348, 4982921, 190249401]
xyzw = np.zeros((1024, 4), np.int32)
xyzw[:, 0] = rs.randint(-3000, 3000, 1024)
xyzw[:, 1] = rs.randint(0, 384, 1024)
xyzw[:, 2] = rs.randint(-3000, 3000, 1024)
xyzw[:, 3] = [path_seeds[i % len(path_seeds)] for i in range(1024)]
xyzw[:64, 0] = rs.randint(-2 ** 30, 2 ** 30, 64); xyzw[:64, 2] = rs.randint(-2 ** 30, 2 ** 30, 64)      # far / negative coordinates
xyzw[64] = [0, 0, 0, 0]; xyzw[65] = [-1, -1, -1, -1]; xyzw[66] = [2 ** 31 - 1, 383, -2 ** 31, 7589341]
u3 = np.zeros((1024, 4), np.float32); u4 = np.zeros((1024, 4), np.float32)
s3 = np.zeros(1024, np.uint32); s4 = np.zeros(1024, np.uint32)
for i, (x, y, z, w) in enumerate(xyzw.tolist()):
    so = ctypes.c_uint(0)
    lib.thrust_seeded_u01(x, y, z, w, 0, 4, P(u3[i]), ctypes.byref(so)); s3[i] = so.value
    lib.thrust_seeded_u01(x, y, z, w, 1, 4, P(u4[i]), ctypes.byref(so)); s4[i] = so.value
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "thrust_probe.npz")
np.savez_compressed(out, seeds=seeds, raw=raw, u01=u01, xyzw=xyzw, u01_3=u3, u01_4=u4, seed_3=s3, seed_4=s4)
print("wrote", out, "first draws of seed 0:", raw[0], u01[0][:2])
