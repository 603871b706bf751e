"""Spatial multi-GPU tiling of a chunk world: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The world rectangle is cut into tiles_x x tiles_z tiles of tile_nx x tile_nz chunks, one tile per rank.  What crosses tile borders
(the reference itself is single-GPU, the neighbourhoods are those of terrain.cpp:471-522 and chunk.cu:1158-1196):
  * slope ring (1 column) and erosion padding (6 chunks of RAW layers): heights and raw layers are pure functions of position and
    cost ~2 % of a chunk, so every rank recomputes them for its own padding — no exchange, no cross-GPU iteration (canonical
    raw-padding semantics make zones independent of each other's results);
  * feature placements of the 3-chunk ring (chunk.cu:1158-1196): they depend on eroded layers AND cave layers, i.e. on the expensive
    stages, so ring cells that another rank owns are NOT recomputed: each rank sends the placement lists of its border strip to the
    (up to 8) neighbouring tiles, between region_begin and region_finish.  Ring cells outside the world rectangle have no owner and
    are computed locally.

Wire protocol (compact: "counts, then payload"), two batched point-to-point phases per step with every peer at once:
  1. headers: the two list lengths of every cell the peer needs (8 bytes per cell);
  2. payload: only the entries that exist (20 B per surface placement, 24 B per cave placement), packed on the device by
     mmgen_ring_pack; the receiver sizes its buffer from the header it got in phase 1.  A dense cell would be 29.7 KB; a typical
     cell carries 1 - 3 KB.
The base fill of the tile (kernFill without feature lists) needs nothing from the ring and is issued while phase 2 is in flight.
No all-reduce / all-gather on the data path.  Results are bit-identical to the single-process region pipeline (tests:
tests/test_distributed_cpu.py with gloo + the CPU oracle as backend, tests/test_gpu_features.py on the device).

`backend` is any object with region_begin / region_placement_buffers / region_fill / region_finish and ring_header / ring_offsets /
ring_pack / ring_unpack (MMGen on a GPU).
"""
import ctypes
from dataclasses import dataclass

RING = 3


@dataclass(frozen=True)
class TileLayout:
    world_cx0: int
    world_cz0: int
    tiles_x: int
    tiles_z: int
    tile_nx: int
    tile_nz: int

    @property
    def world_size(self):
        return self.tiles_x * self.tiles_z

    def region(self, rank):
        tx, tz = rank % self.tiles_x, rank // self.tiles_x
        return (self.world_cx0 + tx * self.tile_nx, self.world_cz0 + tz * self.tile_nz, self.tile_nx, self.tile_nz)

    def owner(self, cx, cz):
        """Rank whose tile contains chunk (cx, cz), or -1 outside the world rectangle."""
        x, z = cx - self.world_cx0, cz - self.world_cz0
        if x < 0 or z < 0 or x >= self.tiles_x * self.tile_nx or z >= self.tiles_z * self.tile_nz:
            return -1
        return (x // self.tile_nx) + self.tiles_x * (z // self.tile_nz)

    def ring_cells(self, rank):
        """Ring-extended grid of `rank`: list of (cell index, cx, cz, owner) for every cell, z-major."""
        cx0, cz0, nx, nz = self.region(rank)
        out = []
        for z in range(nz + 2 * RING):
            for x in range(nx + 2 * RING):
                cx, cz = cx0 - RING + x, cz0 - RING + z
                out.append((x + (nx + 2 * RING) * z, cx, cz, self.owner(cx, cz)))
        return out

    def _cell_of(self, rank, cx, cz):
        cx0, cz0, nx, _ = self.region(rank)
        return (cx - cx0 + RING) + (nx + 2 * RING) * (cz - cz0 + RING)

    def exchange_plan(self, rank):
        """{peer: (recv_cells, send_cells)}: P-grid cell indices of `rank` filled by peer / owned by rank and needed by peer, both in the
        same (cz, cx) order on the two sides."""
        plan = {}
        for cell, cx, cz, own in self._ring_only(rank):
            if own >= 0 and own != rank:
                plan.setdefault(own, ([], []))[0].append((cz, cx, cell))
        for peer in range(self.world_size):
            if peer == rank:
                continue
            for _, cx, cz, own in self._ring_only(peer):
                if own == rank:
                    plan.setdefault(peer, ([], []))[1].append((cz, cx, self._cell_of(rank, cx, cz)))
        return {p: ([c for _, _, c in sorted(r)], [c for _, _, c in sorted(s)]) for p, (r, s) in plan.items()}

    def _ring_only(self, rank):
        """ring_cells restricted to the ring itself (only ring cells can belong to another rank)."""
        cx0, cz0, nx, nz = self.region(rank)
        w, h = nx + 2 * RING, nz + 2 * RING
        for z in range(h):
            for x in range(w):
                if RING <= x < w - RING and RING <= z < h - RING:
                    continue
                cx, cz = cx0 - RING + x, cz0 - RING + z
                yield (x + w * z, cx, cz, self.owner(cx, cz))

    def local_mask(self, rank, lazy=True):
        """Per cell of the ring-extended grid (include/mmgen.h, mmgen_region_begin): 0 = arrives from the peer that owns it; 2 (1 with
        lazy=False) = no owner (beyond the world's border), computed here - lazily, since those lists go nowhere else; cells of the tile
        itself are always computed in full."""
        cx0, cz0, nx, nz = self.region(rank)
        w, h = nx + 2 * RING, nz + 2 * RING
        mask = [1] * (w * h)
        for cell, _, _, own in self._ring_only(rank):
            mask[cell] = 0 if (own >= 0 and own != rank) else (2 if lazy else 1)
        return mask


class TileContext:
    """Everything about one rank's tile that does not change from step to step: region rectangle, local mask, the exchange plan as
    device index tensors.  Built once; generate_tile() then does no host-side planning per step."""

    def __init__(self, layout, rank, torch, device, loopback=False):
        self.layout, self.rank = layout, rank
        self.region = layout.region(rank)
        self.multi = layout.world_size > 1
        self.loopback = bool(loopback)
        mask = layout.local_mask(rank)
        plan = layout.exchange_plan(rank) if self.multi else {}
        if loopback:
            # One-rank rehearsal of the transport (a box with a single GPU): every ring cell is computed here IN FULL (mask 1) and is
            # also shipped rank -> rank through the real two-phase exchange - header, pack, grouped send / recv to self over the
            # communicator, unpack - after its copy in the placement grid has been wiped (generate_tile).  The tile must come out equal
            # to the plain region: what the rasterisers read travelled over RCCL.
            if self.multi:
                raise ValueError("loopback is a single-tile mode")
            ring = [cell for cell, _, _, _ in layout._ring_only(rank)]
            mask = [1] * len(mask)
            plan = {rank: (ring, ring)}
            self.multi = True
        self.mask_list = mask
        self.mask = (ctypes.c_uint8 * len(mask))(*mask)
        self.peers = sorted(plan)
        send, recv, self.send_seg, self.recv_seg = [], [], [0], [0]
        for p in self.peers:
            r, s = plan[p]
            recv += r
            send += s
            self.send_seg.append(len(send))
            self.recv_seg.append(len(recv))
        self.send_cells = torch.tensor(send, dtype=torch.int32, device=device)
        self.recv_cells = torch.tensor(recv, dtype=torch.int32, device=device)
        self.bound_idx = torch.tensor(self.send_seg + [len(send) + 1 + i for i in self.recv_seg], dtype=torch.long, device=device)


def exchange_placements(backend, ctx, bufs, dist, torch, overlap=None):
    """Compact two-phase exchange of the ring placement lists (module docstring).  bufs: dict(fp, cfp, counts) int32 tensors aliasing the
    backend's placement grid (written in place).  overlap: callable issued while the payload is in flight.  Returns bytes received."""
    if not ctx.peers:
        if overlap:
            overlap()
        return 0
    ns, nr = ctx.send_cells.shape[0], ctx.recv_cells.shape[0]
    hdr_s = backend.ring_header(bufs, ctx.send_cells)
    hdr_r = torch.empty((nr, 2), dtype=torch.int32, device=hdr_s.device)
    ops = []
    for k, peer in enumerate(ctx.peers):
        a, b = ctx.send_seg[k], ctx.send_seg[k + 1]
        c, d = ctx.recv_seg[k], ctx.recv_seg[k + 1]
        if b > a:
            ops.append(dist.P2POp(dist.isend, hdr_s[a:b], peer))
        if d > c:
            ops.append(dist.P2POp(dist.irecv, hdr_r[c:d], peer))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    off_s, off_r = backend.ring_offsets(hdr_s), backend.ring_offsets(hdr_r)
    bounds = torch.cat([off_s, off_r])[ctx.bound_idx].tolist()        # the one host read of the step: message boundaries in words
    sb, rb = bounds[:len(ctx.send_seg)], bounds[len(ctx.send_seg):]
    payload_s = backend.ring_pack(bufs, ctx.send_cells, hdr_s, off_s, sb[-1])
    payload_r = torch.empty((max(rb[-1], 1),), dtype=torch.int32, device=hdr_s.device)
    ops = []
    for k, peer in enumerate(ctx.peers):
        if sb[k + 1] > sb[k]:
            ops.append(dist.P2POp(dist.isend, payload_s[sb[k]:sb[k + 1]], peer))
        if rb[k + 1] > rb[k]:
            ops.append(dist.P2POp(dist.irecv, payload_r[rb[k]:rb[k + 1]], peer))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if overlap:
        overlap()
    for req in reqs:
        req.wait()
    backend.ring_unpack(bufs, ctx.recv_cells, hdr_r, off_r, payload_r)
    return 8 * nr + 4 * rb[-1]


def generate_tile(backend, layout, rank, flags, dist=None, torch=None, want=(), ctx=None):
    """Generates this rank's tile of the world through all stages selected by `flags` (MMGEN_REGION_* bits).  Pass a TileContext to
    keep the per-layout planning out of the step."""
    if ctx is None:
        dev = getattr(backend, "device", "cpu")
        ctx = TileContext(layout, rank, torch if torch is not None else backend.torch, dev)
    cx0, cz0, nx, nz = ctx.region
    features = bool(flags & 2)
    exchange = features and ctx.multi
    backend.region_begin(cx0, cz0, nx, nz, flags, ctx.mask if features else None)
    halo_bytes = 0
    if exchange:
        bufs = backend.region_placement_buffers()

        def overlap():
            backend.region_fill(nx, nz)
            if ctx.loopback:                  # the packed payload is on its way: wipe the local copies, only the wire can restore them
                bufs["counts"][ctx.recv_cells.long()] = 0
        halo_bytes = exchange_placements(backend, ctx, bufs, dist, torch, overlap=overlap)
    out = backend.region_finish(nx, nz, want)
    out["halo_bytes_received"] = halo_bytes
    return out
