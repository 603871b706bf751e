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

Wire protocol: ONE batched point-to-point phase per step with every peer at once, and no host read anywhere in the step.  The cells a
peer needs travel as one message whose size follows from the layout alone (include/mmgen.h mmgen_ring_pack_messages):
  [2 words per cell: its two list lengths][only the entries that exist (20 B per surface placement, 24 B per cave placement), packed
  back to back on the device][slack up to words_per_cell words per cell on average].
A dense cell would be 29.7 KB (7 424 words); a typical cell carries 1 - 3 KB (250 - 750 words); the default budget is 2 048 words per
cell.  A message whose entries do not fit is seen by both sides from the lengths: the cells that did not fit arrive empty and the
context's overflow word is raised - TileContext.check() (one device read, after the step) raises, the caller retries with more slack
(7 424 can never overflow).  The base fill of the tile needs nothing from the ring: region_begin issues it as soon as the caves and the
eroded layers exist, and it runs while the messages are in flight.
No all-reduce / all-gather on the data path.  Results are bit-identical to the single-process region pipeline (tests:
tests/test_distributed_cpu.py with gloo + the CPU oracle as backend, tests/test_gpu_features.py on the device).

`backend` is any object with region_begin / region_placement_buffers / region_fill / region_finish and ring_pack_messages /
ring_unpack_messages (MMGen on a GPU).
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

    def __init__(self, layout, rank, torch, device, loopback=False, words_per_cell=2048):
        self.layout, self.rank = layout, rank
        self.words_per_cell = int(words_per_cell)
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
        # message layout (include/mmgen.h mmgen_ring_pack_messages): per peer [2 n lengths][n * words_per_cell payload words]
        self.send_msg, send_slots = message_layout(self.send_seg, self.words_per_cell)
        self.recv_msg, recv_slots = message_layout(self.recv_seg, self.words_per_cell)
        self.send_slots = torch.tensor(send_slots, dtype=torch.int32, device=device).reshape(-1, 4)
        self.recv_slots = torch.tensor(recv_slots, dtype=torch.int32, device=device).reshape(-1, 4)
        self.send_buf = torch.zeros((max(self.send_msg[-1], 1),), dtype=torch.int32, device=device)
        self.recv_buf = torch.zeros((max(self.recv_msg[-1], 1),), dtype=torch.int32, device=device)
        self.send_scratch = torch.zeros((3 * len(send) + 1,), dtype=torch.int32, device=device)
        self.recv_scratch = torch.zeros((3 * len(recv) + 1,), dtype=torch.int32, device=device)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=device)
        self._torch = torch
        self._host_overflow = None          # pinned copy of the overflow word of the last step + the event behind that copy (GPU)
        self._copied = None

    def _raise(self, need):
        self.overflow.zero_()
        raise RuntimeError(f"ring message overflow: a peer's cells needed {need} payload words, the budget is {self.words_per_cell} per cell "
                           f"on average - the tile of that step is INVALID (cells that did not fit arrived empty); rebuild the TileContext "
                           f"with a larger words_per_cell (7424 can never overflow)")

    def _verdict(self, dist):
        """The overflow word every rank will see: the MAXIMUM over the ranks.  An oversized message is only noticed by its sender and its
        receiver; if only those two raised, every other rank would walk into the next exchange with peers that have left the step and hang
        until the communicator's timeout.  One 4-byte all-reduce per step, enqueued behind the step (nccl: stream-ordered, the host does
        not wait; gloo: a CPU tensor) - bookkeeping, not data path.  Stand-in `dist` objects of one-process tests have no all_reduce."""
        if dist is None or self.loopback or not hasattr(dist, "all_reduce") or not hasattr(dist, "get_world_size") or dist.get_world_size() <= 1:
            return self.overflow
        v = self.overflow.clone()
        try:
            dist.all_reduce(v, op=dist.ReduceOp.MAX)
        except (RuntimeError, ValueError, TypeError) as e:           # a backend without integer MAX: every rank lands here alike, and keeps its own word
            if not getattr(self, "_verdict_warned", False):
                self._verdict_warned = True
                import sys
                print(f"mmgen: the ring-overflow verdict is not agreed over the ranks ({e!r}); a rank that overflows raises alone", file=sys.stderr)
            return self.overflow
        return v

    def check(self, dist=None):
        """Reads the overflow word (a device read: call it after the step, not inside it).  Raises - on EVERY rank when `dist` is the
        process group - when a ring message did not fit.  Collective: all ranks call it, or none."""
        self._copied = None
        need = int(self._verdict(dist).item())
        if need:
            self._raise(need)

    def note_step(self, dist=None):
        """After a step's region_finish: agrees on the step's overflow word with the other ranks (_verdict) and starts its copy to pinned
        memory behind everything the step enqueued, without waiting for it (generate_tile calls this; check_previous() looks at it at the
        start of the next step, on every rank in the same step)."""
        v = self._verdict(dist)
        if self.overflow.is_cuda:
            t = self._torch
            if self._host_overflow is None:
                self._host_overflow = t.zeros((1,), dtype=t.int32).pin_memory()
            self._host_overflow.copy_(v, non_blocking=True)
            self._copied = t.cuda.Event()
            self._copied.record()
        else:
            self._copied = int(v[0]) + 1              # (CPU backends: the agreed value itself, + 1 so that 0 still means "a step was noted")

    def check_previous(self):
        """Start of a step: the previous step's agreed overflow word.  On a GPU the wait is for an event recorded behind the previous step -
        over by the time the caller has consumed that step's tile - so the step itself still makes no device read."""
        if self._copied is None:
            return
        if isinstance(self._copied, int):
            need = self._copied - 1
        else:
            self._copied.synchronize()
            need = int(self._host_overflow[0])
        self._copied = None
        if need:
            self._raise(need)


_K_WORD = -7046029254386353131        # 0x9E3779B97F4A7C15 as int64
_K_CHUNK = -4417276706812531889       # 0xC2B2AE3D27D4EB4F as int64


def chunk_digests(blocks, torch):
    """One 64-bit digest per chunk of block ids ([chunks, 98304] uint8, device or host tensor) as int64 [chunks], computed where the tensor
    lives: sum over the chunk's 8-byte words w of word * (2 w + 1) K1 (mod 2^64).  Every multiplier is odd, so a changed, moved or swapped
    word changes it.  tests/golden/world_digests.npz holds these for every chunk of the [-128, 128)^2 world as the CPU ORACLE generates it
    (tests/golden/make_world_digests.py): BASELINE configs 4 and 5 at full size and every tile of bench.py's layouts are compared chunk by chunk."""
    n = blocks.shape[0]
    words = blocks.contiguous().view(torch.int64).view(n, -1)
    mw = (2 * torch.arange(words.shape[1], dtype=torch.int64, device=words.device) + 1) * _K_WORD
    out = torch.empty(n, dtype=torch.int64, device=words.device)
    for c0 in range(0, n, 1024):                      # (bounded temporaries: 100 MB per slab)
        out[c0:c0 + 1024] = (words[c0:c0 + 1024] * mw).sum(1)
    return out


def checksum_of_digests(digests, torch):
    """A tile's checksum from its chunks' digests in the tile's z-major order: sum over chunks c of digest_c * (2 c + 1) K2 (mod 2^64)."""
    digests = torch.as_tensor(digests).reshape(-1)
    mc = (2 * torch.arange(digests.shape[0], dtype=torch.int64, device=digests.device) + 1) * _K_CHUNK
    return int((digests * mc).sum().item()) & 0xFFFFFFFFFFFFFFFF


def tile_checksum(blocks, torch):
    """64-bit position-dependent checksum of a tile's block ids: checksum_of_digests(chunk_digests(blocks)) - linear, one pass."""
    return checksum_of_digests(chunk_digests(blocks, torch), torch)


def load_world_digests(path):
    """tests/golden/world_digests.npz -> (cx0, cz0, int64 array [nz, nx]): the ORACLE's digest of chunk (cx0 + x, cz0 + z) at [z, x]."""
    import numpy as np
    f = np.load(path)
    cx0, cz0, nx, nz = (int(v) for v in f["world"])
    d = f["digests"]
    assert d.shape == (nz, nx) and d.dtype == np.int64
    return cx0, cz0, d


def golden_tile_digests(world, cx0, cz0, nx, nz):
    """The digests of the rectangle [cx0, cx0 + nx) x [cz0, cz0 + nz) in a region's z-major chunk order, or None where the golden world
    (load_world_digests) does not cover it."""
    wx0, wz0, d = world
    if cx0 < wx0 or cz0 < wz0 or cx0 + nx > wx0 + d.shape[1] or cz0 + nz > wz0 + d.shape[0]:
        return None
    return d[cz0 - wz0:cz0 - wz0 + nz, cx0 - wx0:cx0 - wx0 + nx].reshape(-1).copy()


def layout_key(layout):
    return f"{layout.tiles_x}x{layout.tiles_z} tiles of {layout.tile_nx}x{layout.tile_nz} chunks at ({layout.world_cx0},{layout.world_cz0})"


def message_layout(seg, words_per_cell):
    """seg: the peers' segment boundaries in a cell list.  Returns (message boundaries in words, flat slots [4 per cell]): slot =
    (word of the cell's two lengths, first payload word of its peer's message, index of the peer's first cell, payload capacity)."""
    bounds, slots = [0], []
    for k in range(len(seg) - 1):
        a, b = seg[k], seg[k + 1]
        n = b - a
        base = bounds[-1]
        for i in range(n):
            slots += [base + 2 * i, base + 2 * n, a, words_per_cell * n]
        bounds.append(base + 2 * n + words_per_cell * n)
    return bounds, slots


def exchange_placements(backend, ctx, bufs, dist, torch, overlap=None, timing=None):
    """One-phase exchange of the ring placement lists (module docstring).  bufs: dict(fp, cfp, counts) int32 tensors aliasing the backend's
    placement grid (written in place).  overlap: callable issued while the messages are in flight.  timing: optional dict of event
    lists (bench.py) - records events around pack / wait / unpack on the current stream.  Returns bytes received.  No host reads."""
    if not ctx.peers:
        if overlap:
            overlap()
        return 0
    mark = (lambda name: timing[name].append(_event(torch))) if timing is not None else (lambda name: None)
    mark("pack0")
    backend.ring_pack_messages(bufs, ctx.send_cells, ctx.send_slots, ctx.send_scratch, ctx.send_buf, ctx.overflow)
    mark("pack1")
    ops = []
    for k, peer in enumerate(ctx.peers):
        if ctx.send_msg[k + 1] > ctx.send_msg[k]:
            ops.append(dist.P2POp(dist.isend, ctx.send_buf[ctx.send_msg[k]:ctx.send_msg[k + 1]], peer))
        if ctx.recv_msg[k + 1] > ctx.recv_msg[k]:
            ops.append(dist.P2POp(dist.irecv, ctx.recv_buf[ctx.recv_msg[k]:ctx.recv_msg[k + 1]], peer))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if overlap:
        overlap()
    for req in reqs:
        req.wait()                          # (orders the current stream behind the transfers; does not block the host under nccl)
    mark("arrived")
    backend.ring_unpack_messages(bufs, ctx.recv_cells, ctx.recv_slots, ctx.recv_scratch, ctx.recv_buf, ctx.overflow)
    mark("unpack1")
    return 4 * ctx.recv_msg[-1]


def _event(torch):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def generate_tile(backend, layout, rank, flags, dist=None, torch=None, want=(), ctx=None, timing=None, words_per_cell=2048):
    """Generates this rank's tile of the world through all stages selected by `flags` (MMGEN_REGION_* bits).  Pass a TileContext to
    keep the per-layout planning out of the step.

    Ring messages have a fixed budget (module docstring); cells that do not fit arrive EMPTY, which would silently drop features.  It
    cannot go unnoticed: with a caller-owned context the overflow word of step i - the maximum over ALL ranks, so that every rank raises
    in the same step instead of leaving its peers in the next exchange - is looked at when step i + 1 starts (RuntimeError; no device read
    inside a step) and by ctx.check(dist) whenever the caller wants certainty - the returned tile is valid once either has passed; without a
    context (one-off call) the word is read before returning."""
    own_ctx = ctx is None
    if own_ctx:
        dev = getattr(backend, "device", "cpu")
        ctx = TileContext(layout, rank, torch if torch is not None else backend.torch, dev, words_per_cell=words_per_cell)
    else:
        ctx.check_previous()
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
        halo_bytes = exchange_placements(backend, ctx, bufs, dist, torch, overlap=overlap, timing=timing)
    out = backend.region_finish(nx, nz, want)
    out["halo_bytes_received"] = halo_bytes
    if exchange:
        if own_ctx:
            ctx.check(dist)               # nobody else can: the context dies with this call
        else:
            ctx.note_step(dist)
    return out
