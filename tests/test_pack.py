"""Region wire format (include/mmgen.h, csrc/mmgen_pack.hip): per-column run-length pairs.  The format is ours (the reference has
none), so the checks are: the GPU encoder's bytes == a plain numpy restatement of the format, decode(encode(x)) == x on the device
and on the host decoder, edge cases (runs longer than 256, alternating ids, empty batch), and a malformed stream is rejected."""
import numpy as np
import pytest


def encode_chunk(blocks):
    """numpy restatement of the format: u16 runsOfColumn[256], then (id, len - 1) pairs, runs capped at 256 voxels."""
    counts = np.zeros(256, np.uint16)
    pairs = []
    cols = blocks.reshape(256, 384)
    for c in range(256):
        col = cols[c]
        change = np.flatnonzero(np.diff(col)) + 1
        starts = np.concatenate([[0], change]); ends = np.concatenate([change, [384]])
        n = 0
        for s, e in zip(starts, ends):
            length = int(e - s)
            while length > 0:
                take = min(length, 256)
                pairs.append((int(col[s]), take - 1)); n += 1
                length -= take
        counts[c] = n
    return np.concatenate([counts.view(np.uint8), np.array(pairs, np.uint8).reshape(-1)])


def test_numpy_format_round_trip():
    rng = np.random.default_rng(3)
    b = np.zeros(98304, np.uint8)
    b.reshape(256, 384)[:, :130] = 57
    b.reshape(256, 384)[7, :] = 58                       # one 384-run: 256 + 128
    b.reshape(256, 384)[9, :] = rng.integers(0, 140, 384)
    enc = encode_chunk(b)
    counts = enc[:512].view(np.uint16)
    assert counts[7] == 2 and counts[0] == 2 and enc.size == 512 + 2 * int(counts.sum())
    # decode by hand
    out = np.zeros(98304, np.uint8); pos = 512
    for c in range(256):
        y = 0
        for _ in range(counts[c]):
            n = int(enc[pos + 1]) + 1; out[384 * c + y:384 * c + y + n] = enc[pos]; y += n; pos += 2
        assert y == 384
    assert np.array_equal(out, b)


@pytest.mark.gpu
def test_pack_matches_format_and_round_trips(gen):
    import torch
    blocks = gen.generate_region(1488, -1110, 3, 2)["blocks"]                      # jungle: trees, plants, water, caves
    extra = torch.zeros((3, 98304), dtype=torch.uint8, device=blocks.device)
    extra[1] = 57                                                                   # 384-runs: 256 + 128 per column
    extra[2] = (torch.arange(98304, device=blocks.device) % 2).to(torch.uint8) * 5 # alternating: 384 runs per column (worst case)
    allb = torch.cat([blocks, extra])
    p = gen.pack(allb)
    data = p["data"].cpu().numpy(); off = p["chunk_offset"].cpu().numpy(); nb = p["chunk_bytes"].cpu().numpy()
    hb = allb.cpu().numpy()
    for c in range(allb.shape[0]):
        ref = encode_chunk(hb[c])
        assert nb[c] == ref.size, f"chunk {c}: {nb[c]} bytes vs {ref.size}"
        assert np.array_equal(data[off[c]:off[c] + nb[c]], ref), f"chunk {c} packed bytes"
        assert np.array_equal(gen.unpack_chunk_host(data[off[c]:off[c] + nb[c]]), hb[c])
    assert nb[6] == 512 + 2 * 2 * 256 and nb[7] == 512 + 2 * 2 * 256 and nb[8] == 512 + 2 * 384 * 256
    assert torch.equal(gen.unpack(p["data"], p["chunk_offset"]), allb)
    ratio = 98304 * 6 / nb[:6].sum()
    assert ratio > 4, ratio                                                         # jungle (leaves, plants, caves) is the hard case: still > 4x
    # malformed streams are rejected by the host decoder, never read out of bounds
    good = data[off[0]:off[0] + nb[0]].copy()
    with pytest.raises(ValueError):
        gen.unpack_chunk_host(good[:-2])
    bad = good.copy(); bad[513] = 255                                              # first run of column 0 now 256 long: column overflows
    with pytest.raises(ValueError):
        gen.unpack_chunk_host(bad)
    # empty batch
    e = gen.pack(torch.zeros((0, 98304), dtype=torch.uint8, device=blocks.device))
    assert e["data"].shape[0] == 0


@pytest.mark.gpu
def test_pack_full_size_round_trip(gen):
    """Config 4's per-GPU tile (1 024 chunks, all stages): unpack(pack(x)) == x on the device; compression reported."""
    import torch
    blocks = gen.generate_region(-32, -32, 32, 32)["blocks"]
    p = gen.pack(blocks)
    assert torch.equal(gen.unpack(p["data"], p["chunk_offset"]), blocks)
    print("packed bytes per chunk:", int(p["data"].shape[0]) // 1024)
    assert p["data"].shape[0] * 8 < blocks.numel()
