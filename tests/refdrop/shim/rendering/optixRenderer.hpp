// tests/refdrop: the three entry points of the reference's OptixRenderer that Terrain::tick calls (terrain.cpp:600, 660, 672) - the real
// header needs D3D11 + OptiX + GLFW.  Stub: the ray tracer's acceleration structures are outside the generation path.
#pragma once
class Chunk;
class OptixRenderer {
public:
    int built = 0, destroyed = 0, rootBuilds = 0;
    void buildChunkAccel(const Chunk*) { ++built; }
    void destroyChunk(const Chunk*) { ++destroyed; }
    void buildRootAccel() { ++rootBuilds; }
};
