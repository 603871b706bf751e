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
    assert k("k_fill_cave")["scratch"] <= 16                       # eight dwords spilled in the range-draw block, none in the noise loops
    # eight waves per SIMD
    m = k("k_fill_base")
    assert m["vgpr"] <= 64 and m["scratch"] == 0 and m["lds"] <= cu_lds // 8, m
    # four persistent 4-wave workgroups per CU at 128 VGPRs
    m = k("k_apply_features")
    assert m["vgpr"] <= 128 and m["scratch"] == 0 and m["lds"] <= cu_lds // 4, m
    # the relaxation: three workgroups per CU alone, one beside four cave workgroups
    m = k("k_erode_zones")
    assert m["scratch"] == 0 and 3 * m["lds"] <= cu_lds and m["lds"] + 4 * k("k_cave_voxels")["lds"] <= cu_lds, m
