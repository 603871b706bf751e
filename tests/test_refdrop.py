"""the drop-in compile proof.  The reference's own src/terrain/terrain.cpp + terrain.hpp, UNMODIFIED and compiled
where they lie, type-check against mega-minecraft_amd/host/chunk.hpp standing in the reference tree for chunk.hpp (every call site of
Terrain::tick, terrain.cpp:643-937, against chunk.hpp:99-172: the five static stage functions with their 8 - 14 staging arguments and the
stream, gatherHeightfield / generateFeaturePlacements / gatherFeaturePlacements / createVBOs, the state accessors, the public members the
scheduler and its debug printers read) - and the whole thing links against libmmgen.so into oracle/_ref/ref_terrain_dropin, which
tests/test_gpu_schedulers.py runs on the MI355X.  Needs /root/reference (this container only; nothing of the reference travels as source)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "terrain", "terrain.cpp")), reason="needs the reference's sources (build container only)")
def test_reference_terrain_compiles_and_links_against_the_mmgen_chunk():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "refdrop"), "check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "type-checks against" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # the translation unit really is the reference's file (a symbolic link into /root/reference), and chunk.hpp really is ours
    tree = os.path.join(ROOT, "oracle", "_ref", "refdrop_tree", "terrain")
    assert os.path.realpath(os.path.join(tree, "terrain.cpp")) == os.path.join(REF, "src", "terrain", "terrain.cpp")
    assert os.path.realpath(os.path.join(tree, "terrain.hpp")) == os.path.join(REF, "src", "terrain", "terrain.hpp")
    assert "mega-minecraft_amd/host/chunk.hpp" in open(os.path.join(tree, "chunk.hpp")).read()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "refdrop")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_terrain_dropin")
    syms = subprocess.run(["nm", "-C", "--defined-only", exe], capture_output=True, text=True).stdout
    for name in ("Terrain::tick(float)", "Terrain::updateChunk(int, int)", "Chunk::fill(", "Chunk::erodeZone(", "Chunk::createVBOs()"):
        assert name in syms, name
    # ... and the generation path inside it is the C ABI's: the binary imports mmgen_* and defines no kernel of its own
    undefined = subprocess.run(["nm", "-C", "--undefined-only", exe], capture_output=True, text=True).stdout
    for name in ("mmgen_generate_heightfields", "mmgen_generate_layers", "mmgen_erode_zone", "mmgen_generate_caves", "mmgen_fill", "mmgen_mesh_fill"):
        assert name in undefined, name


def test_a_signature_change_breaks_the_reference_build(tmp_path):
    """The check has teeth: with one argument of Chunk::generateCaves swapped in a COPY of our header, the reference's terrain.cpp stops compiling."""
    if not os.path.exists(os.path.join(REF, "src", "terrain", "terrain.cpp")):
        pytest.skip("needs the reference's sources (build container only)")
    hdr = open(os.path.join(ROOT, "mega-minecraft_amd", "host", "chunk.hpp")).read()
    needle = "CaveLayer* host_caveLayers, CaveLayer* dev_caveLayers, mmhostStream stream);"
    assert hdr.count(needle) == 1
    bad = hdr.replace(needle, "mmhostStream stream, CaveLayer* host_caveLayers, CaveLayer* dev_caveLayers);").replace('"../../include/mmgen.h"', f'"{ROOT}/include/mmgen.h"')
    tree = tmp_path / "tree" / "terrain"
    tree.mkdir(parents=True)
    for f in ("terrain.cpp", "terrain.hpp", "biome.hpp", "block.hpp"):
        os.symlink(os.path.join(REF, "src", "terrain", f), tree / f)
    (tree / "chunk.hpp").write_text(bad)
    shim = os.path.join(ROOT, "tests", "refdrop", "shim")
    cmd = ["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-DMMHOST_REFERENCE_TREE", "-DGLM_ENABLE_EXPERIMENTAL", "-DGLEW_NO_GLU", "-w", f"-I{tmp_path / 'tree'}", f"-I{shim}",
           f"-I{REF}/src", f"-I{REF}/src/terrain", f"-I{REF}/external/include", "-I/opt/rocm/include", "-fsyntax-only", "-include", os.path.join(shim, "prelude.hpp"),
           str(tree / "terrain.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "generateCaves" in r.stderr
