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
 * Rule tables are compile-time constants of the code object, so this only selects the device and checks it is gfx950. */
int mmgen_init(int device);
const char* mmgen_error_string(int code);
/* pre-size the library-internal scratch (per-column cave info) so that later calls allocate nothing (graph capture) */
int mmgen_reserve(int max_chunks_per_call);

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
/* Batched form: num_zones buffers of MMGEN_GATHERED_LAYERS_SIZE floats back to back (one launch per pass for all zones);
 * *max_passes (nullable) receives the largest number of relaxation passes any zone needed. */
int mmgen_erode_zones(float* d_gathered_layers, int num_zones, float* d_accumulated_heights, int* max_passes, void* stream);

/* Chunk::generateCaves, device part (chunk.cu:755-937,970-981): default-fill + kernGenerateCaves.
 * out: d_cave_layers [n][256][32] mmgen_cave_layer */
int mmgen_generate_caves(const float* d_heightfields, const float* d_biome_weights, const int32_t* d_chunk_world_block_pos, int num_chunks,
                         mmgen_cave_layer* d_cave_layers, void* stream);

/* Chunk::fill, device part (chunk.cu:1202-1510,1603-1616): kernFill for every chunk of the batch in ONE launch.
 * d_feature_placements / d_cave_feature_placements: [n][2048] / [n][4096] NONE-terminated gathered lists (may be NULL = empty),
 * d_feature_bounds [n][4] int32 = {allFeaturesHeightBounds.xy, allCaveFeaturesHeightBounds.xy} (chunk.cu:1555-1570; NULL with NULL lists).
 * out: d_blocks [n][98304] u8 */
int mmgen_fill(const float* d_heightfields, const float* d_biome_weights, const float* d_layers, const mmgen_cave_layer* d_cave_layers,
               const int32_t* d_chunk_world_block_pos, int num_chunks, const mmgen_feature_placement* d_feature_placements,
               const mmgen_cave_feature_placement* d_cave_feature_placements, const int32_t* d_feature_bounds, uint8_t* d_blocks, void* stream);

/* Measurement hooks (not part of the reference's interface): when enabled every kernel launch is bracketed by HIP events on its
 * launch stream; mmgen_profile_collect() waits for them and returns total milliseconds and launch counts per kernel id
 * (arrays of mmgen_profile_num_kernels() entries), then clears the records. */
void mmgen_profile_enable(int on);
int mmgen_profile_num_kernels(void);
const char* mmgen_profile_kernel_name(int id);
int mmgen_profile_collect(double* total_ms, long long* counts);

/* Test-only: evaluates device math function `fn` (MMGEN_PROBE_*) on n packed fp32 items (ints bit-cast); used by the parity
 * tests to pin the device math against golden vectors.  Not part of the reference's interface. */
int mmgen_debug_probe(int fn, const float* d_in, int n, float* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMGEN_H */
