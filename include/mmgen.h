/* mmgen — C ABI of the MI355X chunk-generation path (libmmgen.so).
 *
 * Drop-in boundary for the GPU stages of the reference's `Chunk` class (src/terrain/chunk.hpp:99-172, called only from
 * Terrain::tick, src/terrain/terrain.cpp:643-937).  Each entry point replaces the device part of one reference stage;
 * the reference's host part (pack / H2D / D2H / unpack) stays with the caller (see INTEGRATION.md and the C++ mirror in
 * mega-minecraft_amd/host/).  Conventions:
 *   - every pointer prefixed d_ is a DEVICE pointer owned by the caller, laid out exactly like the reference's staging
 *     buffers (sizes in mmgen_types.h); nothing is allocated or freed across the boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous on it;
 *   - return value: 0 on success, otherwise the hipError_t value (mmgen_error_string() names it).  The reference's
 *     convention "print + exit(EXIT_FAILURE)" (src/cuda/cuda_utils.cpp:5-17) is kept by the C++ wrapper, not here;
 *   - the library is stateless with respect to ChunkState (src/terrain/chunk.hpp:18-32): the caller advances it.
 */
#ifndef MMGEN_H
#define MMGEN_H
#include <stddef.h>
#include "mmgen_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* BiomeUtils::init() (src/terrain/biome.hpp:299-305, biomeFuncs.hpp:725-1256) + cudaSetDevice (src/main.cpp:31).
 * Rule tables are compile-time constants of the code object; this selects the device, checks it is gfx950 and builds the per-device
 * simplex-noise lookup image (one tiny launch + wait), so that no later call synchronises or allocates behind the caller's back.
 * Threading contract: calls may come from several host threads, streams and devices; the library-owned scratch of the per-stage
 * calls (mmgen_generate_caves, mmgen_erode_zone(s), mmgen_fill) is keyed by (device, stream), so calls on different streams never
 * share a buffer; calls that use the SAME stream from several threads must be serialised by the caller (the reference is
 * single-threaded, terrain.cpp:587-960).  A mmgen_region owns its scratch. */
int mmgen_init(int device);
const char* mmgen_error_string(int code);
/* pre-size the library-internal scratch of the per-stage calls on `stream` (per-column cave info, fill's voxel lists: 393 KB per chunk, at most 8 192 chunks' worth = 3.2 GB per (device, stream) for a full reserve) so that later
 * calls on that stream allocate nothing (graph capture).  Scratch is keyed by (device, stream). */
int mmgen_reserve(int max_chunks_per_call, void* stream);
/* frees the library-internal scratch kept for `stream` on the current device (synchronises it first): call before destroying a stream,
 * so that the entry does not outlive it and a recycled handle does not inherit its buffers.  mmgen_release_all: every stream, every
 * device entry (synchronises the current device).  Threading: per-stage calls on ONE (device, stream) and a release of that stream must
 * not run concurrently (a stage call launches with pointers into the entry the release frees); different streams are independent. */
int mmgen_release(void* stream);
int mmgen_release_all(void);

/* Chunk::generateHeightfields, device part (chunk.cu:150-185,207-213): kernGenerateHeightfield.
 * in : d_chunk_world_block_pos [n][2] int32 (x, z) world block position of each chunk's (0,0) column
 * out: d_heightfields [n][256] f32, d_biome_weights [n][24][256] f32 */
int mmgen_generate_heightfields(const int32_t* d_chunk_world_block_pos, int num_chunks, float* d_heightfields, float* d_biome_weights,
                                void* stream);

/* Same, fused with Chunk::gatherHeightfield (chunk.cu:231-302): additionally writes the 18x18 gathered heightfield
 * d_gathered [n][324] (ring columns recomputed from position — identical values to the neighbours' heightfields). */
int mmgen_generate_heightfields_gathered(const int32_t* d_chunk_world_block_pos, int num_chunks, float* d_heightfields, float* d_biome_weights,
                                         float* d_gathered, void* stream);

/* Chunk::generateLayers, device part (chunk.cu:308-415,448-455): kernGenerateLayers.
 * in : d_gathered_heightfields [n][324], d_biome_weights [n][24][256], positions; out: d_layers [n][20][256] */
int mmgen_generate_layers(const float* d_gathered_heightfields, const float* d_biome_weights, const int32_t* d_chunk_world_block_pos,
                          int num_chunks, float* d_layers, void* stream);

/* Chunk::fixBackwardStratifiedLayers (chunk.cu:725-749), run by Chunk::erodeZone after (or instead of) erosion. */
int mmgen_fix_backward_layers(float* d_layers, int num_chunks, void* stream);

/* Chunk::erodeZone, device part (chunk.cu:477-601 kernDoErosion + the relaxation loop :672-705).
 * d_gathered_layers: the packed zone buffer of copyLayers(..., true) (chunk.cu:603-656): 8 eroded-layer start planes + the heightfield
 * plane over the 384x384 zone grid (+1 trailing flag word, unused here), MMGEN_GATHERED_LAYERS_SIZE floats; eroded IN PLACE.
 * d_accumulated_heights: 147456 floats, overwritten with the accumulated lift (may be NULL).  Synchronous on return, like the reference.
 * Canonical semantics: synchronous Jacobi passes; the reference's in-place update races between thread blocks. */
int mmgen_erode_zone(float* d_gathered_layers, float* d_accumulated_heights, void* stream);
/* The relaxation loop runs on the device as one persistent launch whose workgroups wait for each other; no wait is unbounded.  If a zone's
 * workgroups do not meet within MMGEN_EROSION_TIMEOUT_MS (environment; default 2 000 ms, a thousand times the longest healthy wait) the
 * launch ends itself - starvation is the realistic cause: persistent kernels of other processes, or of this one on other streams, holding
 * its slots - and the RESCUE PASS enqueued right behind it relaxes every zone that is not done (abandoned, or never drawn because the
 * device handed the launch's workgroups to its XCDs differently than assumed) with single workgroups that wait for nobody: same planes,
 * same pass counts, only slower.  Like the reference's host loop (chunk.cu:682-705) the relaxation therefore cannot fail; a launch that
 * gave up is reported on stderr and counted (mmgen_erosion_stalls).  The code below is kept for ABI stability; no call returns it. */
#define MMGEN_ERROR_EROSION_STALL 20002
/* Batched form: num_zones buffers of MMGEN_GATHERED_LAYERS_SIZE floats back to back (ONE persistent launch relaxes all zones to
 * convergence: the host loop of chunk.cu:682-705 runs on the device, a zone's workgroups meet at a barrier after every block of passes);
 * *max_passes (nullable) receives the largest number of relaxation passes any zone needed. */
int mmgen_erode_zones(float* d_gathered_layers, int num_zones, float* d_accumulated_heights, int* max_passes, void* stream);

/* Chunk::generateCaves, device part (chunk.cu:755-937,970-981): default-fill + kernGenerateCaves.
 * out: d_cave_layers [n][256][32] mmgen_cave_layer */
int mmgen_generate_caves(const float* d_heightfields, const float* d_biome_weights, const int32_t* d_chunk_world_block_pos, int num_chunks,
                         mmgen_cave_layer* d_cave_layers, void* stream);

/* Chunk::generateFeaturePlacements (chunk.cu:999-1156; CPU in the reference, a device kernel here) for a batch of chunks.
 * in : heightfields, biome weights, (eroded, fixed-up) layers, cave layers, positions
 * out: d_feature_placements [n][MMGEN_FP_CAP], d_cave_feature_placements [n][MMGEN_CFP_CAP] in the reference's emission order
 *      (columns z-major, then emission order inside a column), d_counts [n][2] int32 = list lengths (cave count may exceed the cap:
 *      the surplus is dropped and reported by the count). */
int mmgen_generate_feature_placements(const float* d_heightfields, const float* d_biome_weights, const float* d_layers,
                                      const mmgen_cave_layer* d_cave_layers, const int32_t* d_chunk_world_block_pos, int num_chunks,
                                      mmgen_feature_placement* d_feature_placements, mmgen_cave_feature_placement* d_cave_feature_placements,
                                      int32_t* d_counts, void* stream);

/* Chunk::gatherFeaturePlacements (chunk.cu:1158-1196) + the host part of Chunk::fill (chunk.cu:1555-1601) on a rectangular chunk grid:
 * the per-chunk lists above are indexed by grid cell (cell = cx + grid_w * cz); for each of the num_targets cells named by
 * d_target_cells the 49 neighbour lists are concatenated in the reference's fixed offset order, truncated to 2048 / 4096 entries and
 * NONE-terminated; d_feature_bounds [num_targets][4] receives {allFeaturesHeightBounds, allCaveFeaturesHeightBounds} of the
 * un-truncated lists.  Neighbours outside the grid contribute nothing (the reference never fills such a chunk). */
int mmgen_gather_feature_placements(const mmgen_feature_placement* d_feature_placements, const mmgen_cave_feature_placement* d_cave_feature_placements,
                                    const int32_t* d_counts, const int32_t* d_target_cells, int num_targets, int grid_w, int grid_h,
                                    mmgen_feature_placement* d_gathered, mmgen_cave_feature_placement* d_gathered_cave, int32_t* d_feature_bounds,
                                    void* stream);

/* Chunk::placeDecorators (chunk.cu:1634-1747; CPU in the reference, after the D2H of the blocks), in place on d_blocks. */
int mmgen_place_decorators(uint8_t* d_blocks, const float* d_heightfields, const float* d_biome_weights, const mmgen_cave_layer* d_cave_layers,
                           const int32_t* d_chunk_world_block_pos, int num_chunks, void* stream);

/* Chunk::fill, device part (chunk.cu:1202-1510,1603-1616): kernFill for every chunk of the batch in ONE launch.
 * d_feature_placements / d_cave_feature_placements: [n][2048] / [n][4096] NONE-terminated gathered lists (may be NULL = empty),
 * d_feature_bounds [n][4] int32 = {allFeaturesHeightBounds.xy, allCaveFeaturesHeightBounds.xy} (chunk.cu:1555-1570; NULL with NULL lists).
 * out: d_blocks [n][98304] u8 */
int mmgen_fill(const float* d_heightfields, const float* d_biome_weights, const float* d_layers, const mmgen_cave_layer* d_cave_layers,
               const int32_t* d_chunk_world_block_pos, int num_chunks, const mmgen_feature_placement* d_feature_placements,
               const mmgen_cave_feature_placement* d_cave_feature_placements, const int32_t* d_feature_bounds, uint8_t* d_blocks, void* stream);

/* ---- Region fast path: a rectangle of chunks through ALL stages without leaving the device between stages -------------------
 * Replaces, for a batch world, the per-tick stage dispatch of Terrain::tick (src/terrain/terrain.cpp:643-937) and its host round
 * trips.  The region owns its scratch (grow-only device buffers); results go to caller-owned device buffers.
 * Chunks are ordered z-major: index = (cx - cx0) + nx * (cz - cz0).  flags select the stages that are optional in the reference
 * (DEBUG_SKIP_EROSION chunk.cu:12,665-715; features / decorators).  Canonical region semantics: DESIGN.md. */
#define MMGEN_REGION_EROSION 1u
#define MMGEN_REGION_FEATURES 2u
#define MMGEN_REGION_DECORATORS 4u
/* mmgen_region_generate only: compute the 3-chunk ring's placement lists in full (mask 1 below) instead of lazily (mask 2).  The blocks
 * are the same unless a chunk's gathered list exceeds the reference's truncation (2 048 surface / 4 096 cave entries, chunk.cu:1573-1601):
 * the reference truncates the FULL list, a lazy ring can only truncate its shortened one, so entries the reference drops could be kept.
 * The full lengths cannot be known without the work the lazy ring saves; mmgen_region_max_gathered reports what a run has seen. */
#define MMGEN_REGION_EXACT_RING 256u
typedef struct mmgen_region mmgen_region;
int mmgen_region_create(mmgen_region** out);
void mmgen_region_destroy(mmgen_region* region);
/* one call: begin + finish */
int mmgen_region_generate(mmgen_region* region, int cx0, int cz0, int nx, int nz, unsigned flags, uint8_t* d_blocks /*[nx*nz][98304]*/,
                          float* d_heightfields /*[nx*nz][256], nullable*/, void* stream);
/* two-phase form for spatial multi-GPU tiling: begin runs heightfield .. feature placements on the tile plus its 3-chunk ring;
 * h_local_mask (host, [(nx+6)*(nz+6)] bytes over the ring-extended grid, nullable = all 1) says for every RING cell who provides its
 * placement lists (cells of the rectangle itself are always computed in full):
 *   0  the caller writes them into the placement buffers before finish (RCCL halo exchange with the neighbouring tiles, a cache);
 *   1  computed here, complete (the lists may be kept and re-used as ring cells of later regions);
 *   2  computed here LAZILY: only the placements that can reach the rectangle are generated, and only the columns that can produce one
 *      get cave noise (a cave feature reaches <= 8 blocks; a surface feature only stands on its gen's jittered grid point, chunk.cu:999-1008,
 *      which is known before the caves are) - about 1/5 of the ring's cave work.  The blocks of the rectangle are identical; the
 *      ring's lists are a subset of the complete ones and must not be re-used elsewhere.  mmgen_region_generate uses 2 throughout. */
int mmgen_region_begin(mmgen_region* region, int cx0, int cz0, int nx, int nz, unsigned flags, const uint8_t* h_local_mask, void* stream);
int mmgen_region_placement_buffers(mmgen_region* region, mmgen_feature_placement** d_fp /*[grid][MMGEN_FP_CAP]*/,
                                   mmgen_cave_feature_placement** d_cfp /*[grid][MMGEN_CFP_CAP]*/, int32_t** d_counts /*[grid][2]*/,
                                   int* grid_cx0, int* grid_cz0, int* grid_w, int* grid_h);
/* optional middle step: the base blocks of the rectangle (kernFill without feature lists, chunk.cu:1202-1510).  It needs nothing from the
 * placement ring, so a tiling caller issues it while the ring exchange is in flight; finish (same d_blocks) then skips it. */
int mmgen_region_fill(mmgen_region* region, uint8_t* d_blocks, void* stream);
/* optional, BEFORE a begin: names the block buffer the region is going to be finished into, which lets that begin issue the base fill
 * itself as soon as the caves' extents and the eroded layers exist - beside the cave biomes and the placement stages, which only the
 * rasterisers wait for - instead of when mmgen_region_fill / _finish is called (then a no-op for the same pointer).  The fill is ordered
 * behind what the stream held when begin was CALLED: the caller promises that nothing it enqueues between that begin and the finish
 * reads or writes d_blocks.  One-shot (cleared by the begin); ignored in the serial schedule.  mmgen_region_generate does this itself. */
int mmgen_region_set_output(mmgen_region* region, uint8_t* d_blocks);
int mmgen_region_finish(mmgen_region* region, uint8_t* d_blocks, float* d_heightfields /*nullable*/, float* d_layers /*[n][20][256], nullable*/,
                        mmgen_cave_layer* d_cave_layers /*[n][256][32], nullable*/, void* stream);
/* waits for the erosion branch of the last begin and returns the largest pass count of its zones (-1: the wait itself failed) */
int mmgen_region_last_erosion_passes(const mmgen_region* region);
/* The longest gathered (un-truncated) surface / cave placement list any chunk had in the finishes since the last call (synchronises the
 * stream; clears the record).  With a full ring (mask 1 or peer-provided lists) these are the lengths the reference's Chunk::fill truncates
 * at MMGEN_MAX_GATHERED_FEATURES_PER_CHUNK / MMGEN_MAX_GATHERED_CAVE_FEATURES_PER_CHUNK; below them nothing is truncated and the lazy ring
 * gives the same blocks.  (A generated world stays below a third of either limit: tests/test_gpu_features.py.) */
int mmgen_region_max_gathered(mmgen_region* region, int* out_surface, int* out_cave, void* stream);
/* The one capacity of this library that the reference does not have: a chunk's cave placement list holds MMGEN_CFP_CAP entries (the
 * reference pushes into an unbounded vector, chunk.cu:1028-1038; a chunk of a generated world carries < 100, the worst case the
 * algorithm allows is 16 384).  Entries beyond the capacity are dropped, which would change blocks - so it is reported, not hidden:
 * mmgen_region_finish checks every list length it is about to gather (its own and the ones that arrived from other GPUs) on the device
 * and records the largest in host-visible memory; from then on mmgen_region_begin / _finish return MMGEN_ERROR_PLACEMENT_OVERFLOW
 * until the caller acknowledges it.  mmgen_region_max_cave_placements synchronises `stream`, returns the largest cave list length seen
 * since the last call (<= MMGEN_CFP_CAP means nothing was dropped) and clears the record.  (Surface lists cannot overflow: at most one
 * placement per column, MMGEN_FP_CAP = 256.  The per-stage call mmgen_generate_feature_placements returns the raw counts to its caller.) */
#define MMGEN_ERROR_PLACEMENT_OVERFLOW 20001
int mmgen_region_max_cave_placements(mmgen_region* region, int* out_max, void* stream);
/* How the region schedules its stages.  Default (serial = 0): a stage DAG over the caller's stream and three
 * internal ones - the erosion branch (one persistent relaxation launch per zone batch, no host read-backs, finish, fix-up) beside the
 * caves, which need none of it (the reference rotates five streams over its stages, terrain.cpp:129,179-182); the base fill on its own
 * stream, ordered behind whatever the caller's stream holds when d_blocks is first passed in - or, after mmgen_region_set_output, behind
 * what it held when begin was called, and then started by begin itself; the rectangle optionally cut into `slices` z slices (0 =
 * automatic = 1) with the rasterisers / decorators of slice i beside the fill of slice i + 1.  Every internal stream is joined into the
 * caller's stream before mmgen_region_finish returns, so callers order against that one stream only; no region call blocks the host
 * except the three queries (last_erosion_passes, max_cave_placements, max_gathered) and set_serial, which waits for the internal streams.
 * serial = 1: every kernel on the caller's stream in the reference's stage order (what per-kernel timing and the counters want); also
 * selected by MMGEN_REGION_SERIAL=1 in the environment.  Results are identical. */
int mmgen_region_set_serial(mmgen_region* region, int serial, int slices);
/* Streaming callers (host/region_terrain.cpp): keep the eroded layers of up to max_zones whole zones (1.18 MB each, device memory of the
 * region) across calls; 0 = off (the default) and frees them.  A region call then relaxes only the covering zones it has not seen, and
 * runs K1 / K2 on their gathered areas only - a thin strip of new chunks touches up to ten zones, each a 24 x 24-chunk gather and a full
 * relaxation in the reference (terrain.cpp:471-522, chunk.cu:658-723).  Results are identical: with the canonical raw padding (DESIGN.md
 * section 4) a zone's eroded layers are a pure function of its position.  Least recently used zones make room.  Synchronises the device.
 * Not for throughput measurements of a fixed rectangle (every call after the first would skip the erosion).
 * mmgen_region_zone_cache_stats: zones served from the cache / relaxed since the last set_zone_cache. */
int mmgen_region_set_zone_cache(mmgen_region* region, int max_zones);
int mmgen_region_zone_cache_stats(const mmgen_region* region, long long* hits, long long* misses);
/* Copies the placement lists of n whole cells between two placement grids with the per-cell layout of mmgen_region_placement_buffers
 * (fp [cells][MMGEN_FP_CAP], cfp [cells][MMGEN_CFP_CAP], counts [cells][2]): cell d_dst_idx[i] of dst <- cell d_src_idx[i] of src.  A streaming
 * caller keeps the lists of chunks it has generated in its own grid and feeds them back as ring cells of later regions (mask 0 in
 * mmgen_region_begin) instead of having their caves and placements recomputed. */
int mmgen_copy_placements(const mmgen_feature_placement* d_src_fp, const mmgen_cave_feature_placement* d_src_cfp, const int32_t* d_src_counts,
                          const int32_t* d_src_idx, mmgen_feature_placement* d_dst_fp, mmgen_cave_feature_placement* d_dst_cfp, int32_t* d_dst_counts,
                          const int32_t* d_dst_idx, int n, void* stream);

/* Compact wire form of placement-grid cells for the ring exchange between spatial tiles (SURVEY 8e "counts, then payload"; the reference
 * has no multi-GPU path, the lists are those of chunk.cu:1158-1196).  For a list of n grid cells d_cells:
 *   mmgen_ring_header : d_header [n][2] = the cells' two list lengths (as counted, a cave count may exceed MMGEN_CFP_CAP);
 *   mmgen_ring_offsets: d_offsets [n+1] = exclusive scan of the cells' payload words 5 min(c0, FP_CAP) + 6 min(c1, CFP_CAP)
 *                       (both sides run it: the receiver sizes its payload buffer from the received header);
 *   mmgen_ring_pack   : payload words of cell i at d_payload + d_offsets[i]: its surface entries (5 words each), then its cave entries (6);
 *   mmgen_ring_unpack : the inverse, into the placement grid of the receiver, lengths included.
 * Cells of several peers are handled in one call: concatenate the peers' cell lists, the peers' messages are contiguous slices. */
int mmgen_ring_header(const int32_t* d_counts, const int32_t* d_cells, int n, int32_t* d_header, void* stream);
int mmgen_ring_offsets(const int32_t* d_header, int n, int32_t* d_offsets, void* stream);
int mmgen_ring_pack(const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_cells, const int32_t* d_header,
                    const int32_t* d_offsets, int n, int32_t* d_payload, void* stream);
int mmgen_ring_unpack(const int32_t* d_payload, const int32_t* d_header, const int32_t* d_offsets, const int32_t* d_cells, int n,
                      mmgen_feature_placement* d_fp, mmgen_cave_feature_placement* d_cfp, int32_t* d_counts, void* stream);

/* The same exchange in ONE phase, with no host read between counting and sending (what a host that must never synchronise mid-step
 * wants: nothing it needs to know depends on the lists).  The cells of each peer form one message whose size follows from the layout alone:
 *   [2 words per cell: its two raw list lengths][the cells' entries packed back to back, as above][slack up to the message's capacity].
 * d_slots [n][4] (built once per layout by the host): word index of the cell's two lengths in d_messages, first payload word of its
 * peer's message, index (into d_cells) of that peer's first cell, the message's payload capacity in words.  A typical cell carries
 * 250 - 750 words; 5 * MMGEN_FP_CAP + 6 * MMGEN_CFP_CAP = 7 424 words per cell can never overflow.  If a message's entries do not fit,
 * sender and receiver both see it from the lengths: *d_overflow (device, cleared by the caller) is raised to the number of payload words
 * that message needed and the cells that did not fit arrive EMPTY - check it before trusting the step, and retry with more slack.
 * d_scratch: 3 n + 1 words.  Same kernels on both sides; cells of several peers in one call. */
int mmgen_ring_pack_messages(const mmgen_feature_placement* d_fp, const mmgen_cave_feature_placement* d_cfp, const int32_t* d_counts, const int32_t* d_cells,
                             const int32_t* d_slots, int n, int32_t* d_scratch, int32_t* d_messages, int32_t* d_overflow, void* stream);
int mmgen_ring_unpack_messages(const int32_t* d_messages, const int32_t* d_cells, const int32_t* d_slots, int n, int32_t* d_scratch,
                               mmgen_feature_placement* d_fp, mmgen_cave_feature_placement* d_cfp, int32_t* d_counts, int32_t* d_overflow, void* stream);

/* Measurement hooks (not part of the reference's interface): when enabled every kernel launch is bracketed by HIP events on its
 * launch stream; mmgen_profile_collect() waits for them and returns total milliseconds and launch counts per kernel id
 * (arrays of mmgen_profile_num_kernels() entries), then clears the records. */
void mmgen_profile_enable(int on);
int mmgen_profile_num_kernels(void);
const char* mmgen_profile_kernel_name(int id);
int mmgen_profile_collect(double* total_ms, long long* counts);

/* ---- mesh build that follows the path ------------------------------------------------------
 * Replaces Chunk::createVBOs (src/terrain/chunk.cu:1778-2003), the host loop over 98 304 voxels x 6 neighbours per chunk.
 * d_blocks: [*][98304] block ids (a batch, a whole region grid or a pool of chunk slots); d_chunk_idx: [n] which chunks of
 * d_blocks to mesh (NULL = chunks 0 .. n-1); every other per-chunk array is indexed by the position in that work list.
 * d_neighbor_idx: [n][4] index into d_blocks of the N (+z), E (+x), S (-z), W (-x) neighbour chunk, -1 = absent (faces towards an
 * absent chunk are not emitted, chunk.cu:1906), NULL = all absent.  Vertices are the reference's Vertex (40 bytes), indices are local to their chunk, order is the reference's
 * (z, x, y; faces in DirectionEnums::dirVecs order); a chunk with V vertices has exactly 3 V / 2 indices.
 *   mmgen_mesh_count: d_column_verts [n][256] and d_chunk_verts [n] receive the vertex counts.
 *   mmgen_mesh_fill : writes chunk c's vertices at d_verts[d_vert_offset[c] ...] and its indices at d_idx[d_vert_offset[c] * 3 / 2 ...]
 *                     (d_vert_offset: exclusive prefix of d_chunk_verts, computed by the caller who also sizes the buffers);
 *                     d_chunk_world_block_pos [n][2] = (x, z) world block origin of each chunk (X-shaped jitter and uv rotation seeds). */
int mmgen_mesh_count(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, int n, uint32_t* d_column_verts,
                     uint32_t* d_chunk_verts, void* stream);
int mmgen_mesh_fill(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                    const uint32_t* d_column_verts, const uint64_t* d_vert_offset, mmgen_vertex* d_verts, uint32_t* d_idx, void* stream);
/* The same without a host round trip between the count and the fill (a streaming tick: the host neither reads the counts nor uploads the
 * offsets before the fill is enqueued):
 *   mmgen_mesh_offsets      d_vert_offset[c] = exclusive prefix of d_chunk_verts on the device, d_total[0] = their sum;
 *   mmgen_mesh_fill_capped  mmgen_mesh_fill that writes nothing for a chunk whose vertices would end beyond capacity_verts (the caller sized
 *                           d_verts / d_idx for capacity_verts vertices from what earlier ticks needed; it reads d_total afterwards, and when
 *                           that exceeds the capacity it grows the buffers and repeats the three calls);
 *   mmgen_mesh_fill_strip   both in ONE launch for 1 <= n <= 256 chunks (a streaming tick's strip): every workgroup sums the counts of the
 *                           chunks before its own (d_chunk_verts from mmgen_mesh_count), d_vert_offset / d_total are OUTPUTS. */
int mmgen_mesh_offsets(const uint32_t* d_chunk_verts, int n, uint64_t* d_vert_offset, uint64_t* d_total, void* stream);
int mmgen_mesh_fill_strip(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                          const uint32_t* d_column_verts, const uint32_t* d_chunk_verts, uint64_t* d_vert_offset, uint64_t* d_total, uint64_t capacity_verts,
                          mmgen_vertex* d_verts, uint32_t* d_idx, void* stream);
int mmgen_mesh_fill_capped(const uint8_t* d_blocks, const int32_t* d_chunk_idx, const int32_t* d_neighbor_idx, const int32_t* d_chunk_world_block_pos, int n,
                           const uint32_t* d_column_verts, const uint64_t* d_vert_offset, uint64_t capacity_verts, mmgen_vertex* d_verts, uint32_t* d_idx,
                           void* stream);

/* ---- region wire format (the reference has none) ---------------------------------------------
 * A chunk's 98 304 block ids as per-column run-length pairs: u16 runsOfColumn[256], then for each column (x + 16 z, the blocks[]
 * order) its runs as (u8 blockId, u8 length - 1), 1 .. 256 voxels per pair.  512 + 2 R bytes per chunk (typically 5 - 9 KB).
 * d_chunk_idx as in the mesher (NULL = chunks 0 .. n-1 of d_blocks); d_chunk_offset = exclusive prefix of d_chunk_bytes, by the
 * caller, who also sizes d_out.  mmgen_unpack is the device inverse (d_blocks [n][98304] dense) for streams this library produced
 * (it never writes outside a chunk, but it trusts the run counts of the header when reading); mmgen_unpack_chunk_host decodes one
 * chunk on the host (plain C, no device), validates every count and length against packed_bytes and returns -1 on a malformed stream:
 * use it for data of unknown origin. */
int mmgen_pack_count(const uint8_t* d_blocks, const int32_t* d_chunk_idx, int n, uint16_t* d_col_runs /*[n][256]*/, uint32_t* d_chunk_bytes /*[n]*/, void* stream);
int mmgen_pack_fill(const uint8_t* d_blocks, const int32_t* d_chunk_idx, int n, const uint16_t* d_col_runs, const uint64_t* d_chunk_offset, uint8_t* d_out,
                    void* stream);
int mmgen_unpack(const uint8_t* d_packed, const uint64_t* d_chunk_offset, int n, uint8_t* d_blocks, void* stream);
int mmgen_unpack_chunk_host(const uint8_t* packed, size_t packed_bytes, uint8_t* blocks);

/* Test-only: evaluates device math function `fn` (MMGEN_PROBE_*) on n packed fp32 items (ints bit-cast); used by the parity
 * tests to pin the device math against golden vectors.  Not part of the reference's interface. */
int mmgen_debug_probe(int fn, const float* d_in, int n, float* d_out, void* stream);
/* Test-only: caps the queue of deferred clay / moss voxels of mmgen_fill / the region path at `entries` (0 = the library's own size, 2 048 per
 * chunk), so that a test can drive the path that evaluates them in place when a reservation does not fit.  Process-wide. */
int mmgen_debug_set_lush_queue_cap(int entries);
/* Test-only: the following persistent relaxation launches wait for `missing_workgroups` more workgroups than they have, so that the wait
 * can never complete, and give up after timeout_ms: drives the give-up + rescue path.  (0, 0) restores the defaults.  Process-wide. */
int mmgen_debug_erosion_stall(int missing_workgroups, int timeout_ms);
/* process-wide counters (either pointer may be NULL): persistent relaxations that gave up, as far as the host has learnt of them (the
 * synchronous per-stage calls at once, a region at its next call), and zones the rescue pass had to relax (synchronous calls only) */
int mmgen_erosion_stalls(long long* stalls, long long* zones_rescued);
/* Test-only: the library's constant rule tables (BiomeUtils::init, biomeFuncs.hpp:725-1256) as floats in the layout of
 * tools/extract_ref_tables.py, so that a test can hold them to the reference's literals.  d_out == NULL: returns the number of floats. */
int mmgen_debug_tables(float* d_out, int capacity_floats, void* stream);
/* Test-only: rasterises ONE (cave) feature placement into a box of voxels (placeFeature / placeCaveFeature per voxel,
 * featurePlacement.hpp:147,1110); d_out[box_size x*y*z] in z, x, y order (y fastest), 255 = voxel not claimed. */
int mmgen_debug_feature_box(int is_cave, int feature, const int32_t* h_feature_pos, int layer_height, const int32_t* h_box_min,
                            const int32_t* h_box_size, uint8_t* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMGEN_H */
