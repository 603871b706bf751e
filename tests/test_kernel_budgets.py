"""The register / LDS / scratch budgets the occupancy of the hot kernels rests on, read from the SHIPPED library's code objects (no GPU, no
recompilation): the uncompressed clang offload bundles inside libmmgen.so are cut out, `llvm-readelf --notes` prints each code object's
kernel metadata.  A change that silently costs a workgroup per CU (a few hundred bytes of LDS, eight VGPRs) fails here, not in a profile."""
import os
import re
import struct
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mega-minecraft_amd", "libmmgen.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """(triple, bytes) of every entry of every bundle: magic, u64 count, then per entry u64 offset, u64 size, u64 triple length, triple."""
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            yield triple, blob[base + off:base + off + size]


def kernel_metadata():
    blob = open(LIB, "rb").read()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for i, (triple, co) in enumerate(code_objects(blob)):
            if "gfx950" not in triple or not co.startswith(b"\x7fELF"):
                continue
            path = os.path.join(d, f"co{i}.elf")
            open(path, "wb").write(co)
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                num = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1))
                out[name.group(1)] = dict(lds=num("group_segment_fixed_size"), scratch=num("private_segment_fixed_size"), vgpr=num("vgpr_count"),
                                          sgpr_spill=num("sgpr_spill_count"), vgpr_spill=num("vgpr_spill_count"))
    return out


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(READELF)), reason="needs the built library and llvm-readelf")
def test_hot_kernels_keep_their_occupancy_budgets():
    md = kernel_metadata()
    assert len(md) > 30, sorted(md)

    def k(prefix):
        hits = [v for n, v in md.items() if re.match(r"_ZN2mm\d+" + prefix + r"E", n)]
        assert len(hits) == 1, (prefix, [n for n in md if prefix in n])
        return hits[0]

    cu_lds = 160 * 1024
    # six 4-wave workgroups per CU = six waves per SIMD: a sixth of the LDS (allocated in 1 280-byte granules), 512 / 6 -> 80 VGPRs
    for name in ("k_cave_voxels", "k_fill_cave", "k_cave_biomes"):
        m = k(name)
        assert m["lds"] <= cu_lds // 6 // 1280 * 1280 and m["vgpr"] <= 80, (name, m)
    assert k("k_cave_voxels")["scratch"] == 0 and k("k_cave_biomes")["scratch"] == 0
    # four dwords: the per-lane address and the limit of next_live_counter's probe, stored once in the prologue and reloaded only in the
    # block that runs when a wave's work counter has gone dry (a handful of times per wave); none in the noise loops
    assert k("k_fill_cave")["scratch"] <= 32
    # eight waves per SIMD
    m = k("k_fill_base")
    assert m["vgpr"] <= 64 and m["scratch"] == 0 and m["lds"] <= cu_lds // 8, m
    # four persistent 4-wave workgroups per CU at 128 VGPRs
    m = k("k_apply_features")
    assert m["vgpr"] <= 128 and m["scratch"] == 0 and m["lds"] <= cu_lds // 4, m
    # the relaxation: three workgroups per CU alone, one beside four cave workgroups
    m = k("k_erode_zones")
    assert m["scratch"] == 0 and 3 * m["lds"] <= cu_lds and m["lds"] + 4 * k("k_cave_voxels")["lds"] <= cu_lds, m


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernel_disassembly(prefix):
    """instruction lines (mnemonic + operands) of the one gfx950 kernel whose mangled name matches mm::<prefix>"""
    blob = open(LIB, "rb").read()
    with tempfile.TemporaryDirectory() as d:
        for i, (triple, co) in enumerate(code_objects(blob)):
            if "gfx950" not in triple or not co.startswith(b"\x7fELF"):
                continue
            path = os.path.join(d, f"co{i}.elf")
            open(path, "wb").write(co)
            text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
            m = re.search(r"^[0-9a-f]+ <(_ZN2mm\d+" + prefix + r"E[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", text, re.S | re.M)
            if m:
                return [re.sub(r"\s*//.*", "", l).strip() for l in m.group(2).splitlines() if l.strip()]
    raise AssertionError(prefix + " not found in the shipped code objects")


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built library and llvm-objdump")
def test_relaxation_barrier_has_release_ordering_in_the_shipped_isa():
    """k_erode_zones hands planes, masks and phases from workgroup to workgroup (of one XCD) inside one launch with stores to the XCD's L2
    and a counter.  The counter add must not overtake the stores: every storing wave waits for its own stores (s_waitcnt vmcnt(0)), THEN the
    workgroup barrier, THEN one lane publishes the mask, waits for that too, and arrives.  A workgroup-scope fence emits no vmcnt wait on
    gfx950 and s_barrier waits for no counter, so the waits are explicit - this test reads them out of the code object that ships."""
    ins = kernel_disassembly("k_erode_zones")
    idx = lambda pred: [i for i, l in enumerate(ins) if pred(l)]
    # two global atomic adds return nothing: the registration at the top of the kernel (before any store) and the arrive at a zone's barrier
    # (the per-XCD registration count returns its value: sc0)
    adds = idx(lambda l: l.startswith("global_atomic_add ") and "sc0" not in l)
    assert len(adds) == 2, [ins[i] for i in adds]
    a = adds[1]
    payload = [i for i in idx(lambda l: l.startswith("buffer_store_dwordx4")) if i < a]
    # every plane store is a 16-byte store to the XCD's L2 (no sc1: the zone's workgroups share that L2; csrc/mmgen_erosion.hip st4_dev)
    assert len(payload) >= 3 and adds[0] < min(payload), "the registration comes before the first plane store"
    assert not any("sc1" in ins[i] for i in payload), "plane stores are write-through again: st4_dev"
    last = max(payload)
    drains = idx(lambda l: l.startswith("s_waitcnt") and "vmcnt(0)" in l)
    barriers = idx(lambda l: l == "s_barrier")
    d1 = [i for i in drains if last < i < a]
    assert d1, "no s_waitcnt vmcnt(0) between the last payload store and the arrive"
    assert any(d1[0] < b < a for b in barriers), "no workgroup barrier between the storing waves' drain and the arrive"
    ors = [i for i in idx(lambda l: l.startswith("global_atomic_or ")) if last < i < a]
    assert len(ors) == 1, "the zone's changed mask is published once per round, ahead of the arrive"
    assert any(ors[0] < i < a for i in drains), "the mask's atomic is not waited for before the arrive"
    # every load of a handed-off plane is an sc1 load to registers; the spin reads the counter sc1 and sleeps, and consults the clock
    assert sum(1 for l in ins if l.startswith("buffer_load_dwordx4") and "sc1" in l) >= 3
    assert not any(l.startswith("flat_load") or l.startswith("flat_store") for l in ins)
    after = ins[a:]
    assert any(l.startswith("s_sleep") for l in after) and any(l.startswith("s_memrealtime") for l in after), "unbounded spin?"
