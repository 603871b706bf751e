"""ctypes binding of libmmgen.so (include/mmgen.h) + a thin torch-tensor convenience layer.

Mirrors the static stage functions of the reference's ``Chunk`` class (src/terrain/chunk.hpp:99-172): same stage names,
same per-chunk staging layouts, same error convention at this level (a failing stage raises; the C++ wrapper in host/
prints and exits like src/cuda/cuda_utils.cpp:5-17).  There is NO CPU fallback: if the HIP library is missing or the
device is not gfx950 every call fails loudly.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMGEN_LIB", os.path.join(_HERE, "libmmgen.so"))   # MMGEN_LIB: A/B builds of the same ABI

HF, GHF, BW, LAYERS, CAVE, BLOCKS = 256, 324, 6144, 5120, 8192, 98304


def build(verbose=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libmmgen build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C mega-minecraft_amd/csrc` (no CPU fallback exists)")
    # torch first: libmmgen.so needs libamdhip64.so.N, and the loader gives it whichever copy is in the process already.  Loaded before
    # torch it binds /opt/rocm's runtime, torch then brings its own bundled copy, and the process ends up with two HIP runtimes of which
    # ours sees no device (hipErrorNoDevice from mmgen_init: found with build() followed by smoke() in one process, round 6).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    lib.mmgen_init.argtypes = [i32]
    lib.mmgen_error_string.restype = ctypes.c_char_p
    lib.mmgen_error_string.argtypes = [i32]
    lib.mmgen_reserve.argtypes = [i32, vp]
    lib.mmgen_generate_heightfields.argtypes = [vp, i32, vp, vp, vp]
    lib.mmgen_generate_heightfields_gathered.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.mmgen_generate_layers.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.mmgen_fix_backward_layers.argtypes = [vp, i32, vp]
    lib.mmgen_generate_caves.argtypes = [vp, vp, vp, i32, vp, vp]
    lib.mmgen_fill.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    return lib


class MMGen:
    """Stage-level access to libmmgen on one GPU.  All tensors are torch CUDA tensors in the reference's layouts."""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("mmgen needs a GPU (gfx950); no CPU path exists")
        self.lib = load_library()
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self._check(self.lib.mmgen_init(device), "mmgen_init")

    # ------------------------------------------------------------------ helpers
    def _check(self, code, what):
        if code != 0:
            raise RuntimeError(f"{what} failed: {self.lib.mmgen_error_string(code).decode()} ({code})")

    def _stream(self):
        return ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def _empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    def positions(self, chunk_coords):
        """[(cx, cz)] chunk coordinates -> device int32 [n,2] world block positions (Chunk::worldBlockPos, chunk.cu:18-20)."""
        t = self.torch.tensor([[c[0] * 16, c[1] * 16] for c in chunk_coords], dtype=self.torch.int32)
        return t.to(self.device)

    # ------------------------------------------------------------------ stages
    def generate_heightfields(self, pos, gathered=False):
        n = pos.shape[0]
        f32 = self.torch.float32
        hf, bw = self._empty((n, HF), f32), self._empty((n, 24, 256), f32)
        if gathered:
            g = self._empty((n, GHF), f32)
            self._check(self.lib.mmgen_generate_heightfields_gathered(self._p(pos), n, self._p(hf), self._p(bw), self._p(g), self._stream()),
                        "mmgen_generate_heightfields_gathered")
            return hf, bw, g
        self._check(self.lib.mmgen_generate_heightfields(self._p(pos), n, self._p(hf), self._p(bw), self._stream()), "mmgen_generate_heightfields")
        return hf, bw

    def generate_layers(self, gathered, bw, pos):
        n = pos.shape[0]
        layers = self._empty((n, 20, 256), self.torch.float32)
        self._check(self.lib.mmgen_generate_layers(self._p(gathered), self._p(bw), self._p(pos), n, self._p(layers), self._stream()), "mmgen_generate_layers")
        return layers

    def fix_backward_layers(self, layers):
        self._check(self.lib.mmgen_fix_backward_layers(self._p(layers), layers.shape[0], self._stream()), "mmgen_fix_backward_layers")
        return layers

    def erode_zones(self, gathered, want_acc=False):
        """gathered: [zones, MMGEN_GATHERED_LAYERS_SIZE] f32 packed zone planes (copyLayers(to), chunk.cu:603-656), eroded in place.
        Returns (gathered, max_passes[, accumulated])."""
        zones = gathered.shape[0]
        assert gathered.shape[1] == 1327105 and gathered.is_contiguous()
        acc = self._empty((zones, 147456), self.torch.float32) if want_acc else None
        mp = ctypes.c_int(0)
        self.lib.mmgen_erode_zones.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
        self._check(self.lib.mmgen_erode_zones(self._p(gathered), zones, self._p(acc), ctypes.byref(mp), self._stream()), "mmgen_erode_zones")
        return (gathered, mp.value, acc) if want_acc else (gathered, mp.value)

    def erosion_stalls(self):
        """(persistent relaxations that gave up, zones the rescue pass relaxed) - process-wide counters (mmgen_erosion_stalls)."""
        a, b = ctypes.c_longlong(0), ctypes.c_longlong(0)
        self.lib.mmgen_erosion_stalls.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
        self._check(self.lib.mmgen_erosion_stalls(ctypes.byref(a), ctypes.byref(b)), "mmgen_erosion_stalls")
        return a.value, b.value

    def debug_erosion_stall(self, missing_workgroups, timeout_ms):
        """Test hook (mmgen_debug_erosion_stall): the next persistent relaxations are launched short; (0, 0) restores the defaults."""
        self.lib.mmgen_debug_erosion_stall.argtypes = [ctypes.c_int, ctypes.c_int]
        self._check(self.lib.mmgen_debug_erosion_stall(missing_workgroups, timeout_ms), "mmgen_debug_erosion_stall")

    @staticmethod
    def zone_area_coords(zone_cx, zone_cz):
        """The 24x24 chunks a zone gathers for erosion: own 12x12 + 6-chunk padding (terrain.cpp:471-522), z-major."""
        return [(zone_cx - 6 + x, zone_cz - 6 + z) for z in range(24) for x in range(24)]

    def pack_zone_planes(self, layers, hf):
        """copyLayers(..., true) on device tensors: layers [576,20,256], hf [576,256] of zone_area_coords order -> [1, 1327105]."""
        t = self.torch
        planes = t.cat([layers[:, 12:20, :], hf.unsqueeze(1)], dim=1)              # [576, 9, 256]
        planes = planes.view(24, 24, 9, 16, 16).permute(2, 0, 3, 1, 4).reshape(9, 384 * 384)   # [plane][cz,z][cx,x]
        out = t.zeros((1, 1327105), dtype=t.float32, device=self.device)
        out[0, : 9 * 147456] = planes.reshape(-1)
        return out

    def generate_caves(self, hf, bw, pos):
        n = pos.shape[0]
        cave = self._empty((n, 256, 32, 3), self.torch.int32)      # 12-byte mmgen_cave_layer viewed as 3 int32
        self._check(self.lib.mmgen_generate_caves(self._p(hf), self._p(bw), self._p(pos), n, self._p(cave), self._stream()), "mmgen_generate_caves")
        return cave

    def fill(self, hf, bw, layers, cave, pos, features=None, cave_features=None, bounds=None):
        n = pos.shape[0]
        blocks = self._empty((n, BLOCKS), self.torch.uint8)
        self._check(self.lib.mmgen_fill(self._p(hf), self._p(bw), self._p(layers), self._p(cave), self._p(pos), n, self._p(features),
                                        self._p(cave_features), self._p(bounds), self._p(blocks), self._stream()), "mmgen_fill")
        return blocks

    # ------------------------------------------------------------------ feature stages
    def generate_feature_placements(self, hf, bw, layers, cave, pos):
        """-> (fp [n,256,5] int32 view of 20-byte records, cfp [n,1024,6] int32 view of 24-byte records, counts [n,2])."""
        n = pos.shape[0]
        i32 = self.torch.int32
        fp = self.torch.zeros((n, 256, 5), dtype=i32, device=self.device)
        cfp = self.torch.zeros((n, 1024, 6), dtype=i32, device=self.device)
        counts = self.torch.zeros((n, 2), dtype=i32, device=self.device)
        self.lib.mmgen_generate_feature_placements.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 4
        self._check(self.lib.mmgen_generate_feature_placements(self._p(hf), self._p(bw), self._p(layers), self._p(cave), self._p(pos), n,
                                                               self._p(fp), self._p(cfp), self._p(counts), self._stream()),
                    "mmgen_generate_feature_placements")
        return fp, cfp, counts

    def gather_feature_placements(self, fp, cfp, counts, targets, grid_w, grid_h):
        nt = targets.shape[0]
        i32 = self.torch.int32
        gfp = self.torch.zeros((nt, 2048, 5), dtype=i32, device=self.device)
        gcfp = self.torch.zeros((nt, 4096, 6), dtype=i32, device=self.device)
        bounds = self.torch.zeros((nt, 4), dtype=i32, device=self.device)
        self.lib.mmgen_gather_feature_placements.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4
        self._check(self.lib.mmgen_gather_feature_placements(self._p(fp), self._p(cfp), self._p(counts), self._p(targets), nt, grid_w, grid_h,
                                                             self._p(gfp), self._p(gcfp), self._p(bounds), self._stream()),
                    "mmgen_gather_feature_placements")
        return gfp, gcfp, bounds

    def place_decorators(self, blocks, hf, bw, cave, pos):
        self.lib.mmgen_place_decorators.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.mmgen_place_decorators(self._p(blocks), self._p(hf), self._p(bw), self._p(cave), self._p(pos), pos.shape[0],
                                                    self._stream()), "mmgen_place_decorators")
        return blocks

    # ------------------------------------------------------------------ mesh build (Chunk::createVBOs, chunk.cu:1778-2003)
    def create_vbos(self, blocks, world_block_pos, nx=None, nz=None, neighbor_idx=None):
        """blocks: uint8 [n, 98304] on the device.  Either `neighbor_idx` (int32 [n, 4]: N(+z), E(+x), S(-z), W(-x) index into the
        batch, -1 = absent) or a z-major nx x nz grid shape (neighbours = grid neighbours, absent beyond the grid) or neither (every
        chunk alone).  world_block_pos: int32 [n, 2] (x, z) world block origins.  Returns dict(verts float32 view [V, 10] (pos 3, nor 3,
        uv 2, material as 2 x 32-bit), idx int32 [3 V / 2], chunk_verts int32 [n], vert_offset int64 [n])."""
        t = self.torch
        n = blocks.shape[0]
        vp, i32 = ctypes.c_void_p, ctypes.c_int
        self.lib.mmgen_mesh_count.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        self.lib.mmgen_mesh_fill.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        if neighbor_idx is None and nx is not None:
            c = t.arange(n, dtype=t.int32, device=blocks.device)
            x, z = c % nx, c // nx
            neg = t.full_like(c, -1)
            neighbor_idx = t.stack([t.where(z < nz - 1, c + nx, neg), t.where(x < nx - 1, c + 1, neg), t.where(z > 0, c - nx, neg),
                                    t.where(x > 0, c - 1, neg)], dim=1).contiguous()
        colv = self._empty((n, 256), t.int32)
        chv = self._empty((n,), t.int32)
        self._check(self.lib.mmgen_mesh_count(self._p(blocks), None, self._p(neighbor_idx), n, self._p(colv), self._p(chv), self._stream()), "mmgen_mesh_count")
        incl = t.cumsum(chv.to(t.int64), 0)
        off = (incl - chv).contiguous()
        total = int(incl[-1].item()) if n else 0
        verts = self._empty((max(total, 1), 10), t.float32)
        idx = self._empty((max(total * 3 // 2, 1),), t.int32)
        if total:
            self._check(self.lib.mmgen_mesh_fill(self._p(blocks), None, self._p(neighbor_idx), self._p(world_block_pos), n, self._p(colv), self._p(off),
                                                 self._p(verts), self._p(idx), self._stream()), "mmgen_mesh_fill")
        return dict(verts=verts[:total], idx=idx[:total * 3 // 2], chunk_verts=chv, vert_offset=off)

    def create_vbos_capped(self, blocks, world_block_pos, nx, nz, capacity_verts, strip=False):
        """The round-trip-free form (mmgen_mesh_count -> mmgen_mesh_offsets -> mmgen_mesh_fill_capped): offsets scanned on the device, buffers
        sized for `capacity_verts` vertices and pre-filled with a sentinel.  Returns dict(verts [capacity, 10], idx, chunk_verts, vert_offset,
        total) - chunks that would end beyond the capacity are left untouched.  strip=True: mmgen_mesh_fill_strip, the offsets summed inside
        the fill (at most 256 chunks)."""
        t = self.torch
        n = blocks.shape[0]
        vp, i32, u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64
        self.lib.mmgen_mesh_count.argtypes = [vp, vp, vp, i32, vp, vp, vp]
        self.lib.mmgen_mesh_offsets.argtypes = [vp, i32, vp, vp, vp]
        self.lib.mmgen_mesh_fill_capped.argtypes = [vp, vp, vp, vp, i32, vp, vp, u64, vp, vp, vp]
        c = t.arange(n, dtype=t.int32, device=blocks.device)
        x, z = c % nx, c // nx
        neg = t.full_like(c, -1)
        nb = t.stack([t.where(z < nz - 1, c + nx, neg), t.where(x < nx - 1, c + 1, neg), t.where(z > 0, c - nx, neg), t.where(x > 0, c - 1, neg)], dim=1).contiguous()
        colv = self._empty((n, 256), t.int32)
        chv = self._empty((n,), t.int32)
        off = self._empty((n,), t.int64)
        total = self._empty((1,), t.int64)
        verts = t.full((capacity_verts, 10), -7.0, dtype=t.float32, device=blocks.device)
        idx = t.full((capacity_verts * 3 // 2,), -7, dtype=t.int32, device=blocks.device)
        self._check(self.lib.mmgen_mesh_count(self._p(blocks), None, self._p(nb), n, self._p(colv), self._p(chv), self._stream()), "mmgen_mesh_count")
        if strip:
            self.lib.mmgen_mesh_fill_strip.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, u64, vp, vp, vp]
            off.fill_(-1); total.fill_(-1)
            self._check(self.lib.mmgen_mesh_fill_strip(self._p(blocks), None, self._p(nb), self._p(world_block_pos), n, self._p(colv), self._p(chv), self._p(off),
                                                       self._p(total), capacity_verts, self._p(verts), self._p(idx), self._stream()), "mmgen_mesh_fill_strip")
            return dict(verts=verts, idx=idx, chunk_verts=chv, vert_offset=off, total=int(total.item()))
        self._check(self.lib.mmgen_mesh_offsets(self._p(chv), n, self._p(off), self._p(total), self._stream()), "mmgen_mesh_offsets")
        self._check(self.lib.mmgen_mesh_fill_capped(self._p(blocks), None, self._p(nb), self._p(world_block_pos), n, self._p(colv), self._p(off), capacity_verts,
                                                    self._p(verts), self._p(idx), self._stream()), "mmgen_mesh_fill_capped")
        return dict(verts=verts, idx=idx, chunk_verts=chv, vert_offset=off, total=int(total.item()))

    # ------------------------------------------------------------------ region wire format (run-length pairs per column)
    def pack(self, blocks):
        """blocks uint8 [n, 98304] on the device -> dict(data uint8 [total], chunk_bytes int32 [n], chunk_offset int64 [n])."""
        t = self.torch
        n = blocks.shape[0]
        vp, i32 = ctypes.c_void_p, ctypes.c_int
        self.lib.mmgen_pack_count.argtypes = [vp, vp, i32, vp, vp, vp]
        self.lib.mmgen_pack_fill.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        runs = self._empty((n, 256), t.int16)
        nbytes = self._empty((n,), t.int32)
        self._check(self.lib.mmgen_pack_count(self._p(blocks), None, n, self._p(runs), self._p(nbytes), self._stream()), "mmgen_pack_count")
        incl = t.cumsum(nbytes.to(t.int64), 0)
        off = (incl - nbytes).contiguous()
        total = int(incl[-1].item()) if n else 0
        data = self._empty((max(total, 1),), t.uint8)
        if n:
            self._check(self.lib.mmgen_pack_fill(self._p(blocks), None, n, self._p(runs), self._p(off), self._p(data), self._stream()), "mmgen_pack_fill")
        return dict(data=data[:total], chunk_bytes=nbytes, chunk_offset=off)

    def unpack(self, data, chunk_offset):
        t = self.torch
        n = chunk_offset.shape[0]
        self.lib.mmgen_unpack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        blocks = self._empty((n, BLOCKS), t.uint8)
        if n:
            self._check(self.lib.mmgen_unpack(self._p(data), self._p(chunk_offset), n, self._p(blocks), self._stream()), "mmgen_unpack")
        return blocks

    def unpack_chunk_host(self, packed_bytes):
        """Host decoder of one packed chunk (numpy uint8 in, numpy uint8 [98304] out); raises on a malformed stream."""
        import numpy as np
        self.lib.mmgen_unpack_chunk_host.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        src = np.ascontiguousarray(packed_bytes, dtype=np.uint8)
        out = np.zeros(BLOCKS, np.uint8)
        if self.lib.mmgen_unpack_chunk_host(src.ctypes.data, src.size, out.ctypes.data) != 0:
            raise ValueError("malformed packed chunk")
        return out

    # ------------------------------------------------------------------ region fast path (all stages, device resident)
    EROSION, FEATURES, DECORATORS = 1, 2, 4

    def close(self):
        """Destroys the region handle (its device buffers: ~0.4 MB per chunk of the largest region it generated).  Called by __del__; a
        long-lived process that creates many MMGen objects would otherwise hold every handle's memory until it exits."""
        h = getattr(self, "_region_handle", None)
        if h is not None:
            self._region_handle = None
            try:
                self.lib.mmgen_region_destroy.argtypes = [ctypes.c_void_p]
                self.lib.mmgen_region_destroy(h)
            except Exception:                     # (interpreter shutdown: the library may be gone)
                pass

    def __del__(self):
        self.close()

    def _region(self):
        if getattr(self, "_region_handle", None) is None:
            h = ctypes.c_void_p()
            self.lib.mmgen_region_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
            self._check(self.lib.mmgen_region_create(ctypes.byref(h)), "mmgen_region_create")
            self._region_handle = h
            vp, i32 = ctypes.c_void_p, ctypes.c_int
            self.lib.mmgen_region_begin.argtypes = [vp, i32, i32, i32, i32, ctypes.c_uint, vp, vp]
            self.lib.mmgen_region_finish.argtypes = [vp, vp, vp, vp, vp, vp]
            self.lib.mmgen_region_fill.argtypes = [vp, vp, vp]
            if hasattr(self.lib, "mmgen_region_set_output"):
                self.lib.mmgen_region_set_output.argtypes = [vp, vp]
            self.lib.mmgen_ring_header.argtypes = [vp, vp, i32, vp, vp]
            self.lib.mmgen_ring_offsets.argtypes = [vp, i32, vp, vp]
            self.lib.mmgen_ring_pack.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp]
            self.lib.mmgen_ring_unpack.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp]
            self.lib.mmgen_region_placement_buffers.argtypes = [vp] + [ctypes.POINTER(vp)] * 3 + [ctypes.POINTER(i32)] * 4
            self.lib.mmgen_region_last_erosion_passes.argtypes = [vp]
        return self._region_handle

    def region_set_zone_cache(self, max_zones):
        """Keep the eroded layers of up to max_zones zones across region calls (include/mmgen.h mmgen_region_set_zone_cache; 0 = off)."""
        self.lib.mmgen_region_set_zone_cache.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self._check(self.lib.mmgen_region_set_zone_cache(self._region(), max_zones), "mmgen_region_set_zone_cache")

    def region_zone_cache_stats(self):
        h, m = ctypes.c_longlong(0), ctypes.c_longlong(0)
        self.lib.mmgen_region_zone_cache_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_longlong)]
        self._check(self.lib.mmgen_region_zone_cache_stats(self._region(), ctypes.byref(h), ctypes.byref(m)), "mmgen_region_zone_cache_stats")
        return h.value, m.value

    def region_set_serial(self, serial, slices=0):
        """serial=True: every kernel of the region path on one stream (per-kernel attribution); False: the stage DAG (include/mmgen.h)."""
        self.lib.mmgen_region_set_serial.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        self._check(self.lib.mmgen_region_set_serial(self._region(), int(bool(serial)), int(slices)), "mmgen_region_set_serial")

    def region_max_cave_placements(self):
        """Largest cave placement list length seen by region_finish since the last call (synchronises); > CFP_CAP (1024) means entries were
        dropped (include/mmgen.h MMGEN_ERROR_PLACEMENT_OVERFLOW).  Clears the record."""
        m = ctypes.c_int(0)
        self.lib.mmgen_region_max_cave_placements.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
        self._check(self.lib.mmgen_region_max_cave_placements(self._region(), ctypes.byref(m), self._stream()), "mmgen_region_max_cave_placements")
        return m.value

    def region_max_gathered(self):
        """(surface, cave): the longest gathered, un-truncated placement lists of the finishes since the last call (synchronises; clears)."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        self.lib.mmgen_region_max_gathered.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
        self._check(self.lib.mmgen_region_max_gathered(self._region(), ctypes.byref(a), ctypes.byref(b), self._stream()), "mmgen_region_max_gathered")
        return a.value, b.value

    def region_begin(self, cx0, cz0, nx, nz, flags, local_mask=None):
        mask = None
        if local_mask is not None:
            mask = local_mask if isinstance(local_mask, ctypes.Array) else (ctypes.c_uint8 * len(local_mask))(*[int(m) for m in local_mask])
        # the output buffer is named up front (mmgen_region_set_output): begin then issues the base fill as soon as its inputs exist.  The
        # tensor is fresh from torch's stream-ordered allocator and nothing touches it before region_finish hands it out
        self._region_blocks = self._empty((nx * nz, BLOCKS), self.torch.uint8)
        if hasattr(self.lib, "mmgen_region_set_output"):      # (absent only in the older A/B builds MMGEN_LIB may point at)
            self._check(self.lib.mmgen_region_set_output(self._region(), self._p(self._region_blocks)), "mmgen_region_set_output")
        self._check(self.lib.mmgen_region_begin(self._region(), cx0, cz0, nx, nz, flags, mask, self._stream()), "mmgen_region_begin")

    def region_fill(self, nx, nz):
        """Base blocks of the rectangle (no feature lists needed): issued while the placement-ring exchange is in flight."""
        if getattr(self, "_region_blocks", None) is None:
            self._region_blocks = self._empty((nx * nz, BLOCKS), self.torch.uint8)
        self._check(self.lib.mmgen_region_fill(self._region(), self._p(self._region_blocks), self._stream()), "mmgen_region_fill")

    # compact ring exchange (include/mmgen.h mmgen_ring_*): all tensors int32 on the device
    def ring_header(self, bufs, cells):
        n = cells.shape[0]
        header = self._empty((n, 2), self.torch.int32)
        self._check(self.lib.mmgen_ring_header(self._p(bufs["counts"]), self._p(cells), n, self._p(header), self._stream()), "mmgen_ring_header")
        return header

    def ring_offsets(self, header):
        n = header.shape[0]
        off = self._empty((n + 1,), self.torch.int32)
        self._check(self.lib.mmgen_ring_offsets(self._p(header), n, self._p(off), self._stream()), "mmgen_ring_offsets")
        return off

    def ring_pack(self, bufs, cells, header, offsets, total_words):
        payload = self._empty((max(total_words, 1),), self.torch.int32)
        self._check(self.lib.mmgen_ring_pack(self._p(bufs["fp"]), self._p(bufs["cfp"]), self._p(cells), self._p(header), self._p(offsets),
                                             cells.shape[0], self._p(payload), self._stream()), "mmgen_ring_pack")
        return payload

    def ring_unpack(self, bufs, cells, header, offsets, payload):
        self._check(self.lib.mmgen_ring_unpack(self._p(payload), self._p(header), self._p(offsets), self._p(cells), cells.shape[0],
                                               self._p(bufs["fp"]), self._p(bufs["cfp"]), self._p(bufs["counts"]), self._stream()), "mmgen_ring_unpack")

    # one-phase exchange (include/mmgen.h mmgen_ring_pack_messages / _unpack_messages): fixed-size messages, lengths in-band, no host read
    def ring_pack_messages(self, bufs, cells, slots, scratch, messages, overflow):
        vp = ctypes.c_void_p
        self.lib.mmgen_ring_pack_messages.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp]
        self._check(self.lib.mmgen_ring_pack_messages(self._p(bufs["fp"]), self._p(bufs["cfp"]), self._p(bufs["counts"]), self._p(cells), self._p(slots),
                                                      cells.shape[0], self._p(scratch), self._p(messages), self._p(overflow), self._stream()),
                    "mmgen_ring_pack_messages")

    def ring_unpack_messages(self, bufs, cells, slots, scratch, messages, overflow):
        vp = ctypes.c_void_p
        self.lib.mmgen_ring_unpack_messages.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp]
        self._check(self.lib.mmgen_ring_unpack_messages(self._p(messages), self._p(cells), self._p(slots), cells.shape[0], self._p(scratch),
                                                        self._p(bufs["fp"]), self._p(bufs["cfp"]), self._p(bufs["counts"]), self._p(overflow), self._stream()),
                    "mmgen_ring_unpack_messages")

    def region_placement_buffers(self):
        """Torch views (no copy) of the region's ring-extended placement grid: fp [cells,256,5], cfp [cells,1024,6], counts [cells,2]."""
        vp, i32 = ctypes.c_void_p, ctypes.c_int
        fp, cfp, cnt = vp(), vp(), vp()
        x0, z0, w, h = i32(), i32(), i32(), i32()
        self._check(self.lib.mmgen_region_placement_buffers(self._region(), ctypes.byref(fp), ctypes.byref(cfp), ctypes.byref(cnt), ctypes.byref(x0),
                                                            ctypes.byref(z0), ctypes.byref(w), ctypes.byref(h)), "mmgen_region_placement_buffers")
        cells = w.value * h.value
        return dict(fp=self._view(fp.value, (cells, 256, 5)), cfp=self._view(cfp.value, (cells, 1024, 6)), counts=self._view(cnt.value, (cells, 2)),
                    x0=x0.value, z0=z0.value, w=w.value, h=h.value)

    def _view(self, ptr, shape):
        """int32 torch tensor aliasing device memory owned by the library (valid until the next region_begin)."""
        import numpy as np
        n = int(np.prod(shape))

        class _Holder:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}
        return self.torch.as_tensor(_Holder(), device=self.device).view(*shape)

    def region_finish(self, nx, nz, want=()):
        n = nx * nz
        t = self.torch
        blocks = self._region_blocks if getattr(self, "_region_blocks", None) is not None else self._empty((n, BLOCKS), t.uint8)
        self._region_blocks = None
        hf = self._empty((n, HF), t.float32)
        layers = self._empty((n, 20, 256), t.float32) if "layers" in want else None
        cave = self._empty((n, 256, 32, 3), t.int32) if "cave" in want else None
        self._check(self.lib.mmgen_region_finish(self._region(), self._p(blocks), self._p(hf), self._p(layers), self._p(cave), self._stream()),
                    "mmgen_region_finish")
        out = dict(blocks=blocks, hf=hf)
        if layers is not None:
            out["layers"] = layers
        if cave is not None:
            out["cave"] = cave
        return out

    def generate_region(self, cx0, cz0, nx, nz, erosion=True, features=True, decorators=True, want=(), lazy_ring=True):
        flags = (self.EROSION if erosion else 0) | (self.FEATURES if features else 0) | (self.DECORATORS if decorators else 0)
        # the ring's placement lists never leave this call: mask value 2 = computed lazily (include/mmgen.h)
        mask = (ctypes.c_uint8 * ((nx + 6) * (nz + 6)))(*([2] * ((nx + 6) * (nz + 6)))) if (features and lazy_ring) else None
        self.region_begin(cx0, cz0, nx, nz, flags, mask)
        out = self.region_finish(nx, nz, want)
        out["erosion_passes"] = self.lib.mmgen_region_last_erosion_passes(self._region())
        return out

    # ------------------------------------------------------------------ config-2 pipeline (no erosion, no features)
    def generate_chunks_no_erosion(self, pos):
        """K1 -> K2 -> E3 fix-up (DEBUG_SKIP_EROSION semantics, chunk.cu:713-720) -> K4 -> K6; everything stays in HBM."""
        hf, bw, g = self.generate_heightfields(pos, gathered=True)
        layers = self.fix_backward_layers(self.generate_layers(g, bw, pos))
        cave = self.generate_caves(hf, bw, pos)
        blocks = self.fill(hf, bw, layers, cave, pos)
        return dict(hf=hf, bw=bw, gathered=g, layers=layers, cave=cave, blocks=blocks)

    # ------------------------------------------------------------------ test-only device math probe
    PROBES = dict(sin=0, cos=1, pow=2, atan2=3, acos=4, simplex2=5, simplex3=6, fbm2_5=7, fbm3_4=8, rand3from3=9, worley2=10, worley3=11,
                  special_cave_noise=12, biome_height=13, cave_biome=14, hash=15, rng4_u01=16, simplex3_split=17)

    def debug_probe(self, name, packed_in, n, out_per_item=1):
        """packed_in: numpy float32/int32/uint32 array (ints are bit-cast); returns numpy float32 [n, out_per_item]."""
        import numpy as np
        t = self.torch.from_numpy(np.ascontiguousarray(packed_in).view(np.float32).copy()).to(self.device)
        out = self._empty((n, out_per_item), self.torch.float32)
        self.lib.mmgen_debug_probe.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.lib.mmgen_debug_probe(self.PROBES[name], self._p(t), n, self._p(out), self._stream()), "mmgen_debug_probe")
        return out.cpu().numpy()

    def debug_tables(self):
        """The library's constant rule tables as numpy arrays in the layout of tools/extract_ref_tables.py (+ its own reach tables)."""
        import numpy as np
        self.lib.mmgen_debug_tables.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        n = self.lib.mmgen_debug_tables(None, 0, None)
        out = self._empty((n,), self.torch.float32)
        self._check(self.lib.mmgen_debug_tables(self._p(out), n, self._stream()), "mmgen_debug_tables")
        a = out.cpu().numpy()
        t, o = {}, 0
        for name, shape, dt in (("biome_rules", (24, 6), np.uint8), ("grass", (24,), np.uint8), ("material_infos", (20, 4), np.float32),
                                ("biome_material_weights", (24, 20), np.float32), ("feature_bounds", (21, 2), np.int32),
                                ("cave_feature_bounds", (10, 2), np.int32), ("surf_gens", (24, 4, 11), np.float32), ("cave_gens", (5, 3, 9), np.float32),
                                ("deco_gens", (24, 7, 10), np.float32), ("cave_deco_gens", (5, 6, 10), np.float32), ("feature_reach", (21,), np.int32),
                                ("cave_feature_reach", (10,), np.int32), ("gather_offsets", (49, 2), np.int32)):
            k = int(np.prod(shape))
            t[name] = a[o:o + k].reshape(shape).astype(dt)
            o += k
        assert o == n
        return t

    def debug_feature_box(self, is_cave, feature, fpos, layer_height, box_min, box_size):
        i3 = ctypes.c_int32 * 3
        n = box_size[0] * box_size[1] * box_size[2]
        out = self._empty((n,), self.torch.uint8)
        self.lib.mmgen_debug_feature_box.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                                     ctypes.c_void_p, ctypes.c_void_p]
        self._check(self.lib.mmgen_debug_feature_box(int(is_cave), feature, i3(*fpos), layer_height, i3(*box_min), i3(*box_size), self._p(out),
                                                     self._stream()), "mmgen_debug_feature_box")
        return out.cpu().numpy()
