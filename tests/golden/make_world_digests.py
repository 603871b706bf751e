#!/usr/bin/env python3
"""tests/golden/world_digests.npz: one 64-bit digest (mega-minecraft_amd.distributed.chunk_digests) per chunk of the 65 536-chunk world
[-128, 128)^2 - BASELINE config 5, which contains config 4's world [-32, 32)^2 and every tile of bench.py's layouts at N = 1, 2, 4, 8 -
as the CPU ORACLE generates it (all stages: erosion, features, decorators).  Nothing of the HIP path runs here: the file is what the
full-size tests and bench.py's tiles_bit_exact hold the device to, chunk by chunk.

The world is generated in pieces (default 64 x 64 chunks; a chunk is a function of its position only, so the piece size is free and
--piece 32 / 128 must give the same file: tests/test_world_digests.py regenerates sampled chunks from 2 x 2 regions).
    python tests/golden/make_world_digests.py [--piece 64] [--threads N] [--world -128 -128 256 256] out.npz
About 29 000 core-seconds (caves 0.25, fill 0.11 core-s per chunk, 1.3 core-s per erosion zone, ring overhead 1.2 - 1.5 x)."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle_binding import Oracle          # noqa: E402  (the checker: this tool makes test fixtures)

K_WORD = np.int64(-7046029254386353131)    # distributed._K_WORD


def chunk_digests_np(blocks):
    words = blocks.view(np.int64).reshape(blocks.shape[0], -1)
    mw = (2 * np.arange(words.shape[1], dtype=np.int64) + 1) * K_WORD
    with np.errstate(over="ignore"):
        return (words * mw).sum(1, dtype=np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--piece", type=int, default=64)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--world", type=int, nargs=4, default=[-128, -128, 256, 256], metavar=("CX0", "CZ0", "NX", "NZ"))
    a = ap.parse_args()
    o = Oracle(a.threads or None)
    cx0, cz0, nx, nz = a.world
    dig = np.zeros((nz, nx), np.int64)
    t0 = time.time()
    done = 0
    for pz in range(0, nz, a.piece):
        for px in range(0, nx, a.piece):
            w, h = min(a.piece, nx - px), min(a.piece, nz - pz)
            t1 = time.time()
            blocks = o.generate_region(cx0 + px, cz0 + pz, w, h, erosion=True, features=True, decorators=True, lean=True)["blocks"]
            dig[pz:pz + h, px:px + w] = chunk_digests_np(blocks).reshape(h, w)
            done += w * h
            print(f"piece ({cx0 + px},{cz0 + pz}) {w}x{h}: {time.time() - t1:.1f} s, {done}/{nx * nz} chunks, {time.time() - t0:.0f} s elapsed, "
                  f"{o.nthreads} threads", flush=True)
            del blocks
    ub = o.ub_counters()
    np.savez_compressed(a.out, digests=dig, world=np.array([cx0, cz0, nx, nz], np.int32), piece=np.int32(a.piece),
                        ub_counters=np.array([ub["no_layer_found"], ub["cave_layer_overflow"], ub["decorator_out_of_range"]], np.int64))
    print(f"wrote {a.out}: {nx * nz} chunk digests from the oracle in {time.time() - t0:.0f} s; oracle UB counters {ub}", flush=True)


if __name__ == "__main__":
    main()
