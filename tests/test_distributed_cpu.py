"""CPU tests of the N>1 path (mega-minecraft_amd/distributed.py): world_size-2 and -4 gloo process groups, the product's tiling +
halo-exchange orchestration driven with the CPU oracle as compute backend; the stitched tiles must equal the single-process region."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, layout_args, flags, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("mega-minecraft_amd.distributed")
    from oracle_binding import OracleBackend
    layout = d.TileLayout(*layout_args)
    out = d.generate_tile(OracleBackend(nthreads=max(1, (os.cpu_count() or 2) // world)), layout, rank, flags, dist=dist, torch=torch)
    np.save(os.path.join(outdir, f"blocks_{rank}.npy"), out["blocks"])
    np.save(os.path.join(outdir, f"halo_{rank}.npy"), np.array([out["halo_bytes_received"]]))
    dist.barrier()
    dist.destroy_process_group()


def _run(layout_args, flags, tmp_path):
    import torch.multiprocessing as mp
    world = layout_args[2] * layout_args[3]
    mp.spawn(_worker, args=(world, _free_port(), layout_args, flags, str(tmp_path)), nprocs=world, join=True)
    return [np.load(tmp_path / f"blocks_{r}.npy") for r in range(world)], [int(np.load(tmp_path / f"halo_{r}.npy")[0]) for r in range(world)]


def _stitch(tiles, layout_args):
    _, _, tx, tz, nx, nz = layout_args
    W = tx * nx
    world = np.zeros((tx * nx * tz * nz, 98304), np.uint8)
    for r, t in enumerate(tiles):
        ox, oz = (r % tx) * nx, (r // tx) * nz
        for z in range(nz):
            for x in range(nx):
                world[(ox + x) + W * (oz + z)] = t[x + nx * z]
    return world


def test_layout_plan_is_symmetric():
    sys.path.insert(0, ROOT)
    d = importlib.import_module("mega-minecraft_amd.distributed")
    lay = d.TileLayout(-5, 7, 2, 2, 4, 5)
    plans = [lay.exchange_plan(r) for r in range(4)]
    for r in range(4):
        assert set(plans[r]) == {0, 1, 2, 3} - {r}                  # 2x2 tiles: everyone neighbours everyone
        for p, (recv, send) in plans[r].items():
            assert len(recv) == len(plans[p][r][1]) and len(send) == len(plans[p][r][0])
        mask = lay.local_mask(r)
        remote = sum(len(v[0]) for v in plans[r].values())
        assert mask.count(0) == remote and set(mask) <= {0, 1, 2}
    assert lay.owner(-6, 7) == -1 and lay.owner(-5, 7) == 0 and lay.owner(-2, 12) == 2 and lay.owner(2, 16) == 3


def test_two_rank_tiling_matches_single_process(oracle, tmp_path):
    """world_size 2, all stages (erosion + features + decorators): 2x1 tiles of 2x2 chunks across a jungle/swamp border."""
    layout_args = (1487, -1111, 2, 1, 2, 2)
    tiles, halo = _run(layout_args, 7, tmp_path)
    ref = oracle.generate_region(1487, -1111, 4, 2, erosion=True, features=True, decorators=True)
    assert np.array_equal(_stitch(tiles, layout_args), ref["blocks"])
    assert all(h > 0 for h in halo)          # placements really travelled between the ranks


def test_four_rank_tiling_without_features_needs_no_exchange(oracle, tmp_path):
    """world_size 4 (2x2 tiles of 1x1), erosion only: no data-path communication at all, tiles still stitch to the region."""
    layout_args = (-1, -1, 2, 2, 1, 1)
    tiles, halo = _run(layout_args, 1, tmp_path)
    ref = oracle.generate_region(-1, -1, 2, 2, erosion=True, features=False, decorators=False)
    assert np.array_equal(_stitch(tiles, layout_args), ref["blocks"])
    assert halo == [0, 0, 0, 0]


def test_four_rank_2x2_tiling_with_features_compact_exchange(oracle, tmp_path):
    """world_size 4 (2x2 tiles of 2x2 chunks), ALL stages: every rank exchanges ring placements with its three neighbours (edge and
    corner peers) through the compact header + payload protocol; the stitched world equals the single-process region and the bytes
    that travelled are far below the dense 29.7 KB per ring cell."""
    layout_args = (1486, -1112, 2, 2, 2, 2)
    tiles, halo = _run(layout_args, 7, tmp_path)
    ref = oracle.generate_region(1486, -1112, 4, 4, erosion=True, features=True, decorators=True)
    assert np.array_equal(_stitch(tiles, layout_args), ref["blocks"])
    # each rank receives 2*2 + 2*2 + 2*2 = 12 remote ring cells... (tile 2x2: the 3 peers' whole tiles lie inside its ring)
    assert all(0 < h < 12 * 29704 // 2 for h in halo), halo


def test_compact_wire_format_round_trip():
    """ring_header / ring_offsets / ring_pack / ring_unpack of the CPU backend against a hand-built case (the device kernels are held to
    the same statement in tests/test_gpu_features.py)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleBackend
    b = OracleBackend.__new__(OracleBackend)
    b.torch = torch
    g = torch.Generator().manual_seed(5)
    cells = 7
    bufs = dict(fp=torch.randint(0, 1000, (cells, 256, 5), dtype=torch.int32, generator=g),
                cfp=torch.randint(0, 1000, (cells, 1024, 6), dtype=torch.int32, generator=g),
                counts=torch.tensor([[0, 0], [3, 0], [0, 5], [256, 1024], [2, 1500], [1, 1], [7, 9]], dtype=torch.int32))
    sel = torch.tensor([6, 1, 4, 2, 3], dtype=torch.int32)
    hdr = b.ring_header(bufs, sel)
    off = b.ring_offsets(hdr)
    assert hdr.tolist() == [[7, 9], [3, 0], [2, 1500], [0, 5], [256, 1024]]
    assert off.tolist() == [0, 89, 104, 104 + 10 + 6144, 6258 + 30, 6288 + 1280 + 6144]
    payload = b.ring_pack(bufs, sel, hdr, off, int(off[-1]))
    dst = dict(fp=torch.zeros_like(bufs["fp"]), cfp=torch.zeros_like(bufs["cfp"]), counts=torch.zeros_like(bufs["counts"]))
    b.ring_unpack(dst, sel, hdr, off, payload)
    for c in sel.tolist():
        n0, n1 = min(int(bufs["counts"][c, 0]), 256), min(int(bufs["counts"][c, 1]), 1024)
        assert torch.equal(dst["fp"][c, :n0], bufs["fp"][c, :n0]) and torch.equal(dst["cfp"][c, :n1], bufs["cfp"][c, :n1])
        assert torch.equal(dst["counts"][c], bufs["counts"][c])          # raw count travels, also beyond the cap
    assert int(dst["counts"][0].sum()) == 0 and int(dst["fp"][5].abs().sum()) == 0


def test_one_phase_messages_round_trip_and_overflow():
    """ring_pack_messages / ring_unpack_messages of the CPU backend (the statement the device kernels are held to in
    tests/test_gpu_features.py): two peers' cells in one call, lengths in-band, entries packed back to back; a message whose entries do
    not fit raises the overflow word on BOTH sides and the cells that did not fit arrive empty."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import OracleBackend
    d = importlib.import_module("mega-minecraft_amd.distributed")
    b = OracleBackend.__new__(OracleBackend)
    b.torch = torch
    g = torch.Generator().manual_seed(7)
    cells = 7
    bufs = dict(fp=torch.randint(0, 1000, (cells, 256, 5), dtype=torch.int32, generator=g),
                cfp=torch.randint(0, 1000, (cells, 1024, 6), dtype=torch.int32, generator=g),
                counts=torch.tensor([[0, 0], [3, 0], [0, 5], [40, 30], [2, 1500], [1, 1], [7, 9]], dtype=torch.int32))
    sel = torch.tensor([6, 1, 4, 2, 3], dtype=torch.int32)
    seg = [0, 2, 5]                                                    # peer A gets cells 6, 1; peer B gets 4, 2, 3

    def run(words_per_cell):
        bounds, slots = d.message_layout(seg, words_per_cell)
        slots = torch.tensor(slots, dtype=torch.int32).reshape(-1, 4)
        msg = torch.zeros(bounds[-1], dtype=torch.int32)
        of_s, of_r = torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
        b.ring_pack_messages(bufs, sel, slots, None, msg, of_s)
        dst = dict(fp=torch.zeros_like(bufs["fp"]), cfp=torch.zeros_like(bufs["cfp"]), counts=torch.full_like(bufs["counts"], -1))
        b.ring_unpack_messages(dst, sel, slots, None, msg, of_r)
        return bounds, msg, dst, int(of_s), int(of_r)

    bounds, msg, dst, of_s, of_r = run(3200)
    assert bounds == [0, 2 * 2 + 2 * 3200, 6404 + 2 * 3 + 3 * 3200] and of_s == 0 and of_r == 0
    assert msg[0:4].tolist() == [7, 9, 3, 0] and msg[6404:6410].tolist() == [2, 1500, 0, 5, 40, 30]       # raw lengths, also beyond the cap
    assert torch.equal(msg[4:4 + 35], bufs["fp"][6, :7].reshape(-1)) and torch.equal(msg[4 + 35:4 + 35 + 54], bufs["cfp"][6, :9].reshape(-1))
    for c in sel.tolist():
        n0, n1 = min(int(bufs["counts"][c, 0]), 256), min(int(bufs["counts"][c, 1]), 1024)
        assert torch.equal(dst["fp"][c, :n0], bufs["fp"][c, :n0]) and torch.equal(dst["cfp"][c, :n1], bufs["cfp"][c, :n1])
        assert torch.equal(dst["counts"][c], bufs["counts"][c])
    assert dst["counts"][0].tolist() == [-1, -1]                       # a cell that was not on the wire is not touched
    # peer B needs 10 + 6144 | 30 | 200 + 180 = 6564 payload words: at 2 100 words per cell (6 300) the last cell does not fit
    bounds, msg, dst, of_s, of_r = run(2100)
    assert of_s == 6564 and of_r == 6564
    assert dst["counts"][3].tolist() == [0, 0] and dst["counts"][2].tolist() == [0, 5] and dst["counts"][4].tolist() == [2, 1500]
    assert torch.equal(dst["cfp"][4, :1024], bufs["cfp"][4, :1024])


@pytest.mark.parametrize("layout_args", [(-5, 7, 2, 2, 4, 5), (-128, -128, 4, 2, 64, 128), (0, 0, 3, 1, 3, 3), (10, -20, 2, 1, 2, 2)])
def test_cpp_exchange_plan_equals_python_plan(layout_args, tmp_path):
    """The C++ host (mega-minecraft_amd/host/tile_layout.hpp, used by TiledWorld over RCCL) and distributed.py must enumerate the same
    cells in the same order on both ends of every link: the plan printed by host/tile_plan_dump (pure host C++) == TileContext's."""
    import subprocess
    import torch
    sys.path.insert(0, ROOT)
    exe = os.path.join(ROOT, "mega-minecraft_amd", "tile_plan_dump")
    if not os.path.exists(exe):
        subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "mega-minecraft_amd", "host", "tile_plan_dump.cpp")], check=True)
    out = subprocess.run([exe] + [str(a) for a in layout_args], capture_output=True, text=True, check=True).stdout.splitlines()
    d = importlib.import_module("mega-minecraft_amd.distributed")
    lay = d.TileLayout(*layout_args)
    it = iter(out)
    for rank in range(lay.world_size):
        ctx = d.TileContext(lay, rank, torch, "cpu")
        head = next(it).split()
        assert head[:2] == ["rank", str(rank)] and int(head[3]) == sum(1 for m in ctx.mask_list if m) and int(head[5]) == len(ctx.peers)
        for k, peer in enumerate(ctx.peers):
            tok = next(it).split()
            assert tok[0] == "peer" and int(tok[1]) == peer
            i_send = tok.index("send")
            assert [int(t) for t in tok[3:i_send]] == ctx.recv_cells[ctx.recv_seg[k]:ctx.recv_seg[k + 1]].tolist()
            assert [int(t) for t in tok[i_send + 1:]] == ctx.send_cells[ctx.send_seg[k]:ctx.send_seg[k + 1]].tolist()


class _SelfLoop:
    """in-process stand-in for a one-rank communicator: a send to self is matched with the next receive from self, in order"""
    isend, irecv = "isend", "irecv"

    class _Req:
        def wait(self):
            pass

    def P2POp(self, op, tensor, peer):
        return (op, tensor, peer)

    def batch_isend_irecv(self, ops):
        sends = [t for op, t, _ in ops if op == self.isend]
        recvs = [t for op, t, _ in ops if op == self.irecv]
        assert len(sends) == len(recvs)
        for s, r in zip(sends, recvs):
            r.copy_(s.clone())
        return [self._Req() for _ in ops]


def test_loopback_exchange_restores_the_wiped_ring(oracle):
    """TileContext(loopback=True): the one-rank rehearsal of the transport.  Every ring cell is computed in full, packed, its list lengths
    wiped in the placement grid, and restored from the payload that went rank -> rank; the tile equals the plain region, and a transport
    that drops the payload is noticed (the ring's features are missing)."""
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    d = importlib.import_module("mega-minecraft_amd.distributed")
    from oracle_binding import OracleBackend
    lay = d.TileLayout(1487, -1111, 1, 1, 2, 2)
    ref = oracle.generate_region(1487, -1111, 2, 2, erosion=True, features=True, decorators=True)
    ob = OracleBackend(nthreads=oracle.nthreads)
    ctx = d.TileContext(lay, 0, torch, "cpu", loopback=True)
    assert ctx.peers == [0] and ctx.send_cells.tolist() == ctx.recv_cells.tolist() and len(ctx.send_cells) == 8 * 8 - 2 * 2
    assert set(ctx.mask_list) == {1}
    out = d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, ctx=ctx)
    assert np.array_equal(out["blocks"], ref["blocks"]) and out["halo_bytes_received"] > 0

    class _Lossy(_SelfLoop):
        def batch_isend_irecv(self, ops):
            reqs = super().batch_isend_irecv(ops)
            for op, t, _ in ops:
                if op == self.irecv:                                                # the message: lengths and entries
                    t.zero_()
            return reqs
    bad = d.generate_tile(ob, lay, 0, 7, dist=_Lossy(), torch=torch, ctx=ctx)
    assert not np.array_equal(bad["blocks"], ref["blocks"]), "a payload of zeros went unnoticed: the loopback does not test the wire"
    with pytest.raises(ValueError):
        d.TileContext(d.TileLayout(0, 0, 2, 1, 2, 2), 0, torch, "cpu", loopback=True)


def test_generate_tile_surfaces_a_ring_message_overflow(oracle):
    """Cells that do not fit a ring message arrive EMPTY (features silently missing) - generate_tile must not let that pass: with a
    caller-owned context the NEXT step (or ctx.check()) raises, without a context the call itself does."""
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    d = importlib.import_module("mega-minecraft_amd.distributed")
    from oracle_binding import OracleBackend
    lay = d.TileLayout(1487, -1111, 1, 1, 2, 2)
    ob = OracleBackend(nthreads=oracle.nthreads)
    ctx = d.TileContext(lay, 0, torch, "cpu", loopback=True, words_per_cell=4)       # a jungle ring cell carries hundreds of words
    out = d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, ctx=ctx)      # the step itself makes no device read ...
    assert out["halo_bytes_received"] > 0
    with pytest.raises(RuntimeError, match="ring message overflow"):                   # ... the next one starts with the verdict
        d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, ctx=ctx)
    d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, ctx=ctx)              # (the raise consumed the record; this step overflows again)
    with pytest.raises(RuntimeError, match="ring message overflow"):
        ctx.check()
    ok = d.TileContext(lay, 0, torch, "cpu", loopback=True)
    d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, ctx=ok)
    ok.check()                                                                         # the default budget fits


def test_generate_tile_without_context_checks_before_returning(oracle, monkeypatch):
    import torch
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    d = importlib.import_module("mega-minecraft_amd.distributed")
    from oracle_binding import OracleBackend
    lay = d.TileLayout(1487, -1111, 1, 1, 2, 2)
    ob = OracleBackend(nthreads=oracle.nthreads)
    # a one-off call builds its own context; make that one a loopback with a budget that cannot fit
    real = d.TileContext
    monkeypatch.setattr(d, "TileContext", lambda layout, rank, t, dev, words_per_cell=2048: real(layout, rank, t, dev, loopback=True, words_per_cell=words_per_cell))
    with pytest.raises(RuntimeError, match="ring message overflow"):
        d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch, words_per_cell=4)
    out = d.generate_tile(ob, lay, 0, 7, dist=_SelfLoop(), torch=torch)
    ref = oracle.generate_region(1487, -1111, 2, 2, erosion=True, features=True, decorators=True)
    assert np.array_equal(out["blocks"], ref["blocks"])


def test_tile_checksum_is_position_dependent_and_goldens_cover_the_bench_layouts():
    """tile_checksum (bench.py: tiles_bit_exact) changes when a byte changes, when two chunks swap and when two words swap; the committed
    golden world digests (the CPU oracle's, one per chunk of [-128, 128)^2) cover every tile of bench.py's layouts at N = 1, 2, 4, 8."""
    import torch
    sys.path.insert(0, ROOT)
    d = importlib.import_module("mega-minecraft_amd.distributed")
    g = torch.Generator().manual_seed(3)
    blocks = torch.randint(0, 200, (5, 98304), dtype=torch.uint8, generator=g)
    base = d.tile_checksum(blocks, torch)
    assert base == d.tile_checksum(blocks.clone(), torch) and 0 <= base < 2 ** 64
    b = blocks.clone(); b[3, 77777] ^= 1
    assert d.tile_checksum(b, torch) != base
    b = blocks.clone(); b[[1, 2]] = blocks[[2, 1]]
    assert d.tile_checksum(b, torch) != base
    b = blocks.clone(); b[0, 0:8], b[0, 8:16] = blocks[0, 8:16].clone(), blocks[0, 0:8].clone()
    assert d.tile_checksum(b, torch) != base
    # slabs of 1 024 chunks: the sum over several slabs equals the one-slab definition
    many = torch.randint(0, 200, (1030, 64), dtype=torch.uint8, generator=g)
    words = many.view(torch.int64).view(1030, -1)
    mw = (2 * torch.arange(8, dtype=torch.int64) + 1) * d._K_WORD
    mc = (2 * torch.arange(1030, dtype=torch.int64) + 1) * d._K_CHUNK
    assert d.tile_checksum(many, torch) == int((((words * mw).sum(1)) * mc).sum().item()) & 0xFFFFFFFFFFFFFFFF
    # per-chunk digests: tile_checksum is checksum_of_digests(chunk_digests); the numpy statement tests/golden/make_world_digests.py uses agrees
    dig = d.chunk_digests(blocks, torch)
    assert d.checksum_of_digests(dig, torch) == base
    wnp = blocks.numpy().view(np.int64).reshape(5, -1)
    with np.errstate(over="ignore"):
        assert np.array_equal(dig.numpy(), (wnp * ((2 * np.arange(wnp.shape[1], dtype=np.int64) + 1) * np.int64(d._K_WORD))).sum(1, dtype=np.int64))
    # the golden world (tests/golden/world_digests.npz: the ORACLE's digest of every chunk of [-128, 128)^2) covers every tile of bench.py's
    # layouts at N = 1, 2, 4, 8 and BASELINE config 4's world; rectangles outside it have no golden
    world = d.load_world_digests(os.path.join(ROOT, "tests", "golden", "world_digests.npz"))
    assert (world[0], world[1], world[2].shape) == (-128, -128, (256, 256)) and len(np.unique(world[2])) == 65536
    tiles = {1: (1, 1), 2: (2, 1), 4: (2, 2), 8: (4, 2)}
    for n, (tx, tz) in tiles.items():
        lay = d.TileLayout(-(tx * 64) // 2, -(tz * 128) // 2, tx, tz, 64, 128)
        for r in range(n):
            g = d.golden_tile_digests(world, *lay.region(r))
            assert g is not None and g.shape == (64 * 128,)
    assert d.golden_tile_digests(world, -32, -32, 64, 64) is not None and d.golden_tile_digests(world, 100, 100, 64, 64) is None


def _verdict_worker(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = importlib.import_module("mega-minecraft_amd.distributed")
    ctx = d.TileContext(d.TileLayout(0, 0, world, 1, 2, 2), rank, torch, "cpu")
    raised = []
    for step in range(3):
        try:
            ctx.check_previous()
            raised.append(0)
        except RuntimeError as e:
            raised.append(1 if "ring message overflow" in str(e) else -1)
        if rank == 1 and step == 1:
            ctx.overflow[0] = 4242                 # what ring_pack / ring_unpack_messages leave on the two ranks of an oversized message
        ctx.note_step(dist)
    try:
        ctx.check(dist)                            # (the synchronous form agrees too: nothing is pending after step 2's clean verdict)
        raised.append(0)
    except RuntimeError:
        raised.append(1)
    with open(os.path.join(outdir, f"verdict_{rank}.txt"), "w") as f:
        f.write(" ".join(str(v) for v in raised))
    dist.barrier()
    dist.destroy_process_group()


def test_every_rank_raises_a_ring_overflow_in_the_same_step(tmp_path):
    """world_size 3 in a row (gloo): only rank 1 sees an oversized message in step 1 - rank 2 is neither its sender nor its receiver.  The
    step's overflow word is agreed over all ranks (TileContext._verdict: one MAX all-reduce behind the step), so ALL THREE raise at the
    start of step 2 - nobody is left waiting in the next exchange for peers that have gone - and are clean again afterwards."""
    import torch.multiprocessing as mp
    mp.spawn(_verdict_worker, args=(3, _free_port(), str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        assert open(tmp_path / f"verdict_{r}.txt").read().split() == ["0", "0", "1", "0"], r      # (the raise clears the word on the rank that held it)
