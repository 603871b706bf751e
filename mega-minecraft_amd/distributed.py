"""Spatial multi-GPU tiling of a chunk world: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The world rectangle is cut into tiles_x x tiles_z tiles of tile_nx x tile_nz chunks, one tile per rank.  What crosses tile borders
(SURVEY §8e):
  * slope ring (1 column) and erosion padding (6 chunks of RAW layers): heights and raw layers are pure functions of position and
    cost ~2 % of a chunk, so every rank recomputes them for its own padding — no exchange, no cross-GPU iteration (canonical
    raw-padding semantics make zones independent of each other's results);
  * feature placements of the 3-chunk ring (chunk.cu:1158-1196): they depend on eroded layers AND cave layers, i.e. on the expensive
    stages, so ring cells that another rank owns are NOT recomputed: each rank sends the placement lists of its border strip to the
    (up to 8) neighbouring tiles with one batched isend/irecv (RCCL point-to-point), between region_begin and region_finish.
    Ring cells outside the world rectangle have no owner and are computed locally.
No all-reduce / all-gather on the data path.  Results are bit-identical to the single-process region pipeline (tests:
tests/test_distributed_cpu.py with gloo + the CPU oracle as backend, tests/test_gpu_features.py on the device).

`backend` is any object with region_begin / region_placement_buffers / region_finish (MMGen on a GPU).
"""
from dataclasses import dataclass

RING = 3
FP_INTS, CFP_INTS = 256 * 5, 1024 * 6          # int32 words per cell of the two placement arrays (MMGEN_FP_CAP x 20 B, MMGEN_CFP_CAP x 24 B)
CELL_INTS = 2 + FP_INTS + CFP_INTS


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

    def exchange_plan(self, rank):
        """{peer: (recv_cells, send_cells)}: P-grid cell indices of `rank` filled by peer / owned by rank and needed by peer, both in the
        same (cz, cx) order on the two sides."""
        plan = {}
        for cell, cx, cz, own in self.ring_cells(rank):
            if own >= 0 and own != rank:
                plan.setdefault(own, ([], []))[0].append((cz, cx, cell))
        my_cells = {(cx, cz): cell for cell, cx, cz, own in self.ring_cells(rank) if own == rank}
        for peer in range(self.world_size):
            if peer == rank:
                continue
            for _, cx, cz, own in self.ring_cells(peer):
                if own == rank:
                    plan.setdefault(peer, ([], []))[1].append((cz, cx, my_cells[(cx, cz)]))
        return {p: ([c for _, _, c in sorted(r)], [c for _, _, c in sorted(s)]) for p, (r, s) in plan.items()}

    def local_mask(self, rank):
        """1 for ring cells this rank must compute itself (no owner), 0 for cells that arrive from a peer; tile cells are always local."""
        return [1 if (own < 0 or own == rank) else 0 for _, _, _, own in self.ring_cells(rank)]


def exchange_placements(bufs, plan, dist, torch):
    """One batched point-to-point exchange of ring placement lists.  bufs: dict(fp [cells,256,5], cfp [cells,1024,6], counts [cells,2])
    int32 tensors aliasing the backend's placement grid (written in place)."""
    if not plan:
        return 0
    dev = bufs["counts"].device
    ops, recvs = [], []
    for peer in sorted(plan):
        recv_cells, send_cells = plan[peer]
        if send_cells:
            idx = torch.tensor(send_cells, dtype=torch.long, device=dev)
            payload = torch.cat([bufs["counts"][idx].reshape(len(send_cells), -1), bufs["fp"][idx].reshape(len(send_cells), -1),
                                 bufs["cfp"][idx].reshape(len(send_cells), -1)], dim=1).contiguous()
            ops.append(dist.P2POp(dist.isend, payload, peer))
        if recv_cells:
            buf = torch.empty((len(recv_cells), CELL_INTS), dtype=torch.int32, device=dev)
            ops.append(dist.P2POp(dist.irecv, buf, peer))
            recvs.append((recv_cells, buf))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    moved = 0
    for recv_cells, buf in recvs:
        idx = torch.tensor(recv_cells, dtype=torch.long, device=dev)
        bufs["counts"][idx] = buf[:, :2]
        bufs["fp"][idx] = buf[:, 2:2 + FP_INTS].reshape(len(recv_cells), 256, 5)
        bufs["cfp"][idx] = buf[:, 2 + FP_INTS:].reshape(len(recv_cells), 1024, 6)
        moved += buf.numel() * 4
    return moved


def generate_tile(backend, layout, rank, flags, dist=None, torch=None, want=()):
    """Generates this rank's tile of the world through all stages selected by `flags` (MMGEN_REGION_* bits)."""
    cx0, cz0, nx, nz = layout.region(rank)
    features = bool(flags & 2)
    multi = layout.world_size > 1
    mask = layout.local_mask(rank) if (features and multi) else None
    backend.region_begin(cx0, cz0, nx, nz, flags, mask)
    halo_bytes = 0
    if features and multi:
        bufs = backend.region_placement_buffers()
        halo_bytes = exchange_placements(bufs, layout.exchange_plan(rank), dist, torch)
    out = backend.region_finish(nx, nz, want)
    out["halo_bytes_received"] = halo_bytes
    return out
