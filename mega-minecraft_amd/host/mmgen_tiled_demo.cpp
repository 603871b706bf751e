// mmgen_tiled_demo — the C++ multi-GPU host end to end: N processes (one per GPU) generate the tiles of one world over RCCL.
//   mmgen_tiled_demo [--gpus N] [--tile NX NZ] [--steps K] [--verify] [--loopback]
// --verify (N > 1): every rank regenerates the 4-chunk-wide strip of its tile along each border it shares with another tile as a plain
//   mmgen_region_generate call (which recomputes the neighbour's ring cells locally) and compares the blocks: equality means the lists
//   that arrived over RCCL are the right ones.
// --loopback (N = 1): a communicator of ONE rank; the whole ring is shipped rank 0 -> rank 0 through the real exchange (TiledWorld's
//   loopback mode) and the tile is compared with mmgen_region_generate.
// The parent forks its N ranks BEFORE any HIP / RCCL call (a process that has touched the GPU is never forked or exec'd); rank 0 creates
// the ncclUniqueId and publishes it through an anonymous shared mapping.  Every rank prints chunks/s and an FNV-1a checksum of its
// blocks; with --gpus 1 the checksum is that of the plain single-region call, which the demo verifies.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <vector>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include "tiled_world.hpp"

struct Shared { volatile int ready; ncclUniqueId id; volatile unsigned long long sum[64]; volatile double rate[64]; };

static unsigned long long fnv(const uint8_t* p, size_t n)
{
    unsigned long long h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

// blocks of the sub-rectangle [x0, x0 + w) x [z0, z0 + h) (tile-local chunk coordinates) regenerated as a plain region == the tile's?
static int verify_strip(const mmhost::TileLayout& lay, int rank, const std::vector<uint8_t>& tile, int x0, int z0, int w, int h, const char* what)
{
    const auto rg = lay.region(rank);
    const size_t n = (size_t)w * h;
    uint8_t* d = nullptr;
    if (hipMalloc((void**)&d, n * 98304) != hipSuccess) return 1;
    mmgen_region* r = nullptr;
    if (mmgen_region_create(&r)) return 1;
    std::vector<uint8_t> got(n * 98304);
    const int rc = mmgen_region_generate(r, rg[0] + x0, rg[1] + z0, w, h, 7, d, nullptr, nullptr);
    if (rc || hipDeviceSynchronize() != hipSuccess || hipMemcpy(got.data(), d, got.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    mmgen_region_destroy(r);
    (void)hipFree(d);
    size_t bad = 0;
    for (int z = 0; z < h; ++z) for (int x = 0; x < w; ++x)
        bad += std::memcmp(got.data() + ((size_t)x + (size_t)w * z) * 98304, tile.data() + ((size_t)(x0 + x) + (size_t)rg[2] * (z0 + z)) * 98304, 98304) != 0;
    std::printf("rank %d verify %s strip %dx%d at (%d,%d): %s\n", rank, what, w, h, x0, z0, bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}

static int run_rank(int rank, int world, int nx, int nz, int steps, Shared* sh, bool verify, bool loopback)
{
    static const int tiles[9][2] = {{0, 0}, {1, 1}, {2, 1}, {3, 1}, {2, 2}, {5, 1}, {3, 2}, {7, 1}, {4, 2}};
    const int tx = world <= 8 ? tiles[world][0] : world, tz = world <= 8 ? tiles[world][1] : 1;
    mmhost::TileLayout lay{-(tx * nx) / 2, -(tz * nz) / 2, tx, tz, nx, nz};
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);          // this pool's host driver only supports dmabuf IPC
    if (mmgen_init(rank) != 0) { std::fprintf(stderr, "rank %d: mmgen_init failed\n", rank); return 1; }
    ncclComm_t comm = nullptr;
    if (loopback) {
        ncclUniqueId id;
        if (ncclGetUniqueId(&id) != ncclSuccess || ncclCommInitRank(&comm, 1, id, 0) != ncclSuccess) { std::fprintf(stderr, "one-rank communicator failed\n"); return 1; }
    }
    if (world > 1) {
        if (rank == 0) { if (ncclGetUniqueId((ncclUniqueId*)&sh->id) != ncclSuccess) return 1; __sync_synchronize(); sh->ready = 1; }
        while (!sh->ready) usleep(1000);
        ncclUniqueId id; std::memcpy(&id, (const void*)&sh->id, sizeof(id));
        if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) { std::fprintf(stderr, "rank %d: ncclCommInitRank failed\n", rank); return 1; }
    }
    const size_t n = (size_t)nx * nz;
    uint8_t* d_blocks = nullptr;
    if (hipMalloc((void**)&d_blocks, n * 98304) != hipSuccess) return 1;
    int rc = 0;
    {
        mmhost::TiledWorld tw(lay, rank, comm, loopback);
        if (tw.status()) { std::fprintf(stderr, "rank %d: TiledWorld setup failed (%d)\n", rank, tw.status()); return 1; }
        rc = tw.generate(7, d_blocks, nullptr);                     // warm-up (allocations, layout upload)
        // the timed steps are enqueued back to back (no host synchronisation inside or between steps); one wait and one look at the
        // ring-overflow word at the end
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps && rc == 0; ++i) rc = tw.generateAsync(7, d_blocks, nullptr);
        if (rc == 0) rc = tw.finishStep();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc) { std::fprintf(stderr, "rank %d: generate failed (%d)\n", rank, rc); return 1; }
        std::vector<uint8_t> h(n * 98304);
        if (hipMemcpy(h.data(), d_blocks, h.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        sh->sum[rank] = fnv(h.data(), h.size());
        sh->rate[rank] = n * steps / dt;
        std::printf("rank %d/%d tile (%d,%d)+%dx%d: %.0f chunks/s, halo %zu B/step, checksum %016llx\n", rank, world, lay.region(rank)[0],
                    lay.region(rank)[1], nx, nz, sh->rate[rank], tw.lastHaloBytesReceived(), (unsigned long long)sh->sum[rank]);
        if (world == 1) {                                           // the tiled path with one tile == the plain region call
            mmgen_region* r = nullptr; mmgen_region_create(&r);
            const auto rg = lay.region(0);
            if (mmgen_region_generate(r, rg[0], rg[1], rg[2], rg[3], 7, d_blocks, nullptr, nullptr) || hipDeviceSynchronize() != hipSuccess) return 1;
            if (hipMemcpy(h.data(), d_blocks, h.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
            mmgen_region_destroy(r);
            if (fnv(h.data(), h.size()) != sh->sum[0]) { std::fprintf(stderr, "checksum differs from mmgen_region_generate\n"); return 1; }
            std::printf("single tile%s == mmgen_region_generate: ok\n", loopback ? " (ring shipped rank 0 -> rank 0 over RCCL)" : "");
            if (loopback && tw.lastHaloBytesReceived() == 0) { std::fprintf(stderr, "loopback moved no bytes\n"); return 1; }
        }
        if (verify && world > 1) {
            const int tX = rank % lay.tiles_x, tZ = rank / lay.tiles_x, sw = nx < 4 ? nx : 4, sh4 = nz < 4 ? nz : 4;
            int bad = 0;
            if (tX + 1 < lay.tiles_x) bad |= verify_strip(lay, rank, h, nx - sw, 0, sw, nz, "east");
            if (tX > 0) bad |= verify_strip(lay, rank, h, 0, 0, sw, nz, "west");
            if (tZ + 1 < lay.tiles_z) bad |= verify_strip(lay, rank, h, 0, nz - sh4, nx, sh4, "south");
            if (tZ > 0) bad |= verify_strip(lay, rank, h, 0, 0, nx, sh4, "north");
            if (bad) return 1;
        }
    }
    (void)hipFree(d_blocks);
    if (comm) ncclCommDestroy(comm);
    return 0;
}

int main(int argc, char** argv)
{
    int world = 1, nx = 24, nz = 24, steps = 3;
    bool verify = false, loopback = false;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) world = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--tile") && i + 2 < argc) { nx = std::atoi(argv[++i]); nz = std::atoi(argv[++i]); }
        else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) steps = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--verify")) verify = true;
        else if (!std::strcmp(argv[i], "--loopback")) loopback = true;
    }
    if (world < 1 || world > 64 || (loopback && world != 1)) return 2;
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) return 1;
    std::memset(sh, 0, sizeof(Shared));
    if (world == 1) {
        const int rc1 = run_rank(0, 1, nx, nz, steps, sh, verify, loopback);
        std::printf("world 1 GPUs: %.0f chunks/s aggregate%s\n", (double)sh->rate[0], rc1 ? " (FAILED)" : "");
        return rc1;
    }
    std::vector<pid_t> kids;
    for (int r = 0; r < world; ++r) {
        const pid_t p = fork();                                     // nothing has touched HIP yet
        if (p == 0) _exit(run_rank(r, world, nx, nz, steps, sh, verify, loopback));
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0); }
    double total = 0; for (int r = 0; r < world; ++r) total += sh->rate[r];
    std::printf("world %d GPUs: %.0f chunks/s aggregate%s\n", world, total, bad ? " (FAILED)" : "");
    return bad;
}
