/*
 * rayfinder_amd.h -- C ABI of the MI355X-native path-tracing core (librayfinder_amd.so).
 *
 * This is the drop-in boundary for rayfinder's hot path.  The reference (Nelarius/rayfinder) has
 * no FFI layer; its path sits behind three C++ seams.  Every entry point below names the reference
 * interface it replaces (paths relative to the reference repo).  Plain pointers and sizes only;
 * every call returns RF_OK (0) or an error code instead of throwing -- rf_last_error_message()
 * returns the text the reference would have put into its std::runtime_error.  One handle is used
 * from one host thread at a time (as in the reference, whose renderer lives on the GLFW thread).
 *
 * Record layouts are the reference's, byte for byte:
 *   BvhNode 48 B (src/common/bvh.hpp:14-21), Positions 36 B (src/common/triangle_attributes.hpp:7-12),
 *   PositionAttribute 48 B / VertexAttributes 80 B (src/pt-format/vertex_attributes.hpp:7-35),
 *   texture pixels u32 BGRA (src/common/texture.cpp:46), AlignedSkyState 160 B
 *   (src/pt/aligned_sky_state.hpp:34-41).
 */
#ifndef RAYFINDER_AMD_H
#define RAYFINDER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_API __attribute__((visibility("default")))

typedef enum rf_status
{
    RF_OK = 0,
    RF_ERROR_INVALID_ARGUMENT = 1,
    RF_ERROR_RUNTIME = 2,      /* what the reference reports with std::runtime_error */
    RF_ERROR_NO_DEVICE = 3,    /* no HIP device: this library has NO CPU fallback for rendering */
    RF_ERROR_OUT_OF_RANGE = 4, /* sky parameters out of range (sky_state_result != success) */
} rf_status;

/* Text of the last error on the calling thread ("" if none). */
RF_API const char* rf_last_error_message(void);
RF_API const char* rf_version(void);

/* ---------------------------------------------------------------------------------------------
 * Value types
 * ------------------------------------------------------------------------------------------ */
/* nlrs::Camera, src/common/camera.hpp:10-21 (19 floats). */
typedef struct rf_camera
{
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float up[3];
    float right[3];
    float lens_radius;
} rf_camera;

/* nlrs::Sky, src/pt/aligned_sky_state.hpp:15-23. */
typedef struct rf_sky
{
    float turbidity;          /* [1, 10] */
    float albedo[3];          /* [0, 1] */
    float sun_zenith_degrees; /* [0, 90] */
    float sun_azimuth_degrees;
} rf_sky;

/* nlrs::RenderParameters + SamplingParams, src/pt/reference_path_tracer.hpp:26-43. */
typedef struct rf_render_parameters
{
    uint32_t  width, height;          /* framebufferSize */
    rf_camera camera;
    uint32_t  num_samples_per_pixel;  /* default 128 */
    uint32_t  num_bounces;            /* default 4 */
    rf_sky    sky;
    float     exposure;               /* 1 / 2^stops (src/pt/main.cpp:367) */
} rf_render_parameters;

/* nlrs::Texture as a view, src/common/texture.hpp:10-47. */
typedef struct rf_texture
{
    const uint32_t* pixels; /* width*height BGRA (b | g<<8 | r<<16 | a<<24) */
    uint32_t        width, height;
} rf_texture;

/* nlrs::Scene, src/pt/reference_path_tracer.hpp:45-51: non-owning views; rf_renderer_create
 * copies everything to device memory and keeps nothing from these pointers. */
typedef struct rf_scene
{
    const void*       bvh_nodes;           /* 48-B BvhNode records */
    uint64_t          num_bvh_nodes;
    const void*       position_attributes; /* 48-B PositionAttribute records, BVH leaf order */
    const void*       vertex_attributes;   /* 80-B VertexAttributes records, same order */
    uint64_t          num_triangles;
    const rf_texture* base_color_textures;
    uint64_t          num_textures;
} rf_scene;

/* nlrs::RendererDescriptor, src/pt/reference_path_tracer.hpp:53-57, plus device placement. */
typedef struct rf_renderer_descriptor
{
    rf_render_parameters render_params;
    uint32_t             max_width, max_height; /* maxFramebufferSize; 0 = render_params size */
    int32_t              device_ordinal;
    uint64_t             max_paths_in_flight;   /* batch depth; 0 = default (1 Gi paths).  Default or chosen, a render() call whose batches do not fit the device memory
                                                   free at that moment traces the same samples in shallower batches (same image; said on stderr; the configured depth is
                                                   used again once the memory is back).  Path state is allocated on demand: samples x pixels x 148 B */
} rf_renderer_descriptor;

typedef struct rf_stats
{
    uint64_t primary_rays, closest_rays, shadow_rays;
    uint64_t closest_node_visits, closest_triangle_tests; /* only while counting is enabled */
    uint64_t shadow_node_visits, shadow_triangle_tests;
    uint64_t paths;
    uint32_t stack_high_water;
    uint32_t batch_samples_used; /* samples per pixel traced together in the MOST RECENT batch (x the shard's padded pixel count = the batch depth in paths, see
                                  * rf_renderer_memory_info).  Less than asked for when device memory was short at that moment: same image, shallower batches, shorter launches */
    double   ms_raygen, ms_closest, ms_shade, ms_shadow, ms_accumulate; /* while timing is enabled */
    uint32_t launches_raygen, launches_closest, launches_shade, launches_shadow, launches_accumulate;
    uint32_t batches_traced; /* batches since the last reset */
    uint64_t closest_record_fetches, shadow_record_fetches; /* 64-byte BVH records fetched (counting build) */
    /* Always counted.  abandoned_rays: traversals that needed more than 96 stack entries and were cut short (the
     * reference's 32-entry stack, ray_intersection.cpp:148,194 / wgsl:327,375, is overrun long before: undefined
     * there); 0 on every scene tested.  scalar_redo_rays: rays the packed traversal handed to the reference-ordered
     * scalar one (axis-parallel / non-finite rays, origins outside the conservative records' bound, more than 48 pending
     * entries: a full 12-entry LDS stack first evicts its oldest entries to scratch) -- results are identical. */
    uint64_t abandoned_rays, scalar_redo_rays;
    /* Of shadow_rays: rays whose occluder was found by the any-hit launch's first look (kShadowFirstLook: the leaves that stopped the last shadow rays
     * from the same cell of the scene, tested with the reference's box and triangle arithmetic) and that therefore never entered the BVH walk.
     * Same visibility bit; reported so that a rays-per-second figure can be read with and without them. */
    uint64_t shadow_rays_hint_answered;
    /* Of shadow_rays: rays stopped by the very triangle they start on -- the reference pushes the hit point off the surface along the GEOMETRIC normal whatever side the
     * path came from (wgsl:511-519), so wherever the sun stands behind that normal shadowRay (wgsl:321-368) finds the surface itself.  The shading stage tests exactly that
     * (the leaf's exact box, then the triangle, with the reference's arithmetic) and such a ray is never queued for an any-hit launch.  Same visibility bit. */
    uint64_t shadow_rays_self_answered;
} rf_stats;

typedef struct rf_renderer rf_renderer;

/* ---------------------------------------------------------------------------------------------
 * Renderer  (replaces class nlrs::ReferencePathTracer, src/pt/reference_path_tracer.hpp:59-76)
 * ------------------------------------------------------------------------------------------ */
/* ReferencePathTracer(const RendererDescriptor&, const GpuContext&, Scene)
 * (reference_path_tracer.cpp:131-481).  Fails with RF_ERROR_NO_DEVICE when no GPU is present. */
RF_API int rf_renderer_create(const rf_renderer_descriptor* desc, const rf_scene* scene, rf_renderer** out);
RF_API void rf_renderer_destroy(rf_renderer* r);

/* void setRenderParameters(const RenderParameters&) (reference_path_tracer.cpp:556-563):
 * any change resets the accumulation; frameCount keeps counting. */
RF_API int rf_renderer_set_render_parameters(rf_renderer* r, const rf_render_parameters* params);

/* void render(...) called num_frames times (reference_path_tracer.cpp:565-595 + fsMain
 * wgsl:34-57): frame f uses sample index frameCount % spp, adds one sample while fewer than spp
 * are accumulated.  Work is enqueued on the handle's HIP stream; returns without waiting. */
RF_API int rf_renderer_render(rf_renderer* r, uint32_t num_frames);
RF_API int rf_renderer_synchronize(rf_renderer* r);

/* float averageRenderpassDurationMs() const (reference_path_tracer.cpp:706-716): mean GPU time per
 * sample over the last 30 samples. */
RF_API float rf_renderer_average_renderpass_duration_ms(rf_renderer* r);
/* float renderProgressPercentage() const (reference_path_tracer.cpp:718-722). */
RF_API float rf_renderer_render_progress_percentage(const rf_renderer* r);

/* The reference's `imageBuffer` (wgsl:32; never read back there): row-major width*height*4 f32,
 * SUM of samples, 16-byte stride.  This is the parity surface. */
RF_API int rf_renderer_read_accumulation(rf_renderer* r, float* dst, uint32_t* accumulated_sample_count);
/* fsMain's return value (wgsl:59-63) as the BGRA8Unorm swap-chain texel, row-major. */
RF_API int rf_renderer_read_tonemapped(rf_renderer* r, uint32_t* dst_bgra8);

/* Deferred-lighting variant (replaces nlrs::DeferredRenderer's lighting + resolve passes, src/pt/deferred_renderer.hpp,
 * deferred_renderer_lighting_pass.wgsl:96-186 -- fixed 2-bounce surfaceColor, solar disk in the sky term, the
 * 1/16384 + 1024 offset constants :498-500 -- and deferred_renderer_resolve_pass.wgsl:33-54 -- 0.1 / 0.9 exponential
 * average).  The G-buffer comes from one primary ray per pixel through the jittered pixel centre
 * (deferred_renderer.cpp:309-315) instead of the reference's raster pass.  Camera, sky and exposure are the handle's
 * render parameters; the deferred frame counter starts at 0 and is separate from rf_renderer_render's. */
RF_API int rf_renderer_render_deferred(rf_renderer* r, uint32_t num_frames);
RF_API int rf_renderer_reset_deferred(rf_renderer* r);
/* sampleBuffer and accumulationBuffer (width*height*3 f32, row-major: array<array<f32, 3>>) and the resolve pass's
 * return value as BGRA8; any pointer may be NULL. */
RF_API int rf_renderer_read_deferred(rf_renderer* r, float* sample_rgb, float* accumulation_rgb, uint32_t* bgra8, uint32_t* frame_count);

/* Statistics (replaces the ImGui perf read-out, src/pt/main.cpp:251-257). */
RF_API int rf_renderer_set_counting(rf_renderer* r, int enabled);
RF_API int rf_renderer_set_timing(rf_renderer* r, int enabled);
/* Tuning knobs for A/B measurements; none of them changes a result (every combination is covered by the -m gpu parity tests).
 *   traversal_variant 0 | 2            one-ray-per-thread kernels over the 32-byte nodes | persistent kernels over the 64-byte records (default)
 *   quad_from_bounce, quad_shadow_from_bounce         first bounce whose closest-hit / shadow launch reads the 128-byte quad records (two
 *                                      levels of the tree per dependent fetch; defaults 1 / 1; 0 = never; takes precedence over the layouts below)
 *   quad_except_mask, quad_shadow_except_mask         ... except at bounce b when bit b-1 is set (default 0)
 *   quad_half_from_bounce, quad_half_shadow_from_bounce   first bounce whose quad launch reads the 64-byte half-precision quad records
 *                                      (conservative binary16 planes, exact boxes at the leaves; 0 = never; defaults chosen per scene:
 *                                      rf_wide_layout_stats)
 *   quad_local_from_bounce, quad_local_shadow_from_bounce   the same for the 64-byte local-grid quad records (8-bit planes on a
 *                                      per-record power-of-two grid); closest-hit launches: the half-precision records take precedence; shadow launches: these do
 *   compact_from_bounce, compact_shadow_from_bounce   first bounce whose closest-hit / shadow launch reads the compact-capable records
 *                                      (three loads per descending step; defaults 3 / 2; 0 = never)
 *   hot_from_bounce, hot_shadow_from_bounce           the same for the 32-byte records (two loads per step; default 0 = never)
 *   refill_min, refill_min_deep, refill_deep_from_bounce   idle lanes at which a wave refills (40; closest-hit launches from bounce 3 on: 22 on the 64-byte and the
 *                                      half-precision quad records, 40 on the exact quad records)
 *   leaf_vote                          descending lanes below which a wave processes its parked leaves (20)
 *   chunk, chunk_early, chunk_early_bounces   queue entries per cursor claim (128; 256 at bounces 1-2)
 *   shade_sort_from_bounce             first bounce whose shading stage appends each 1024-entry tile's surviving paths in the order of
 *                                      the triangles they hit (default 2; 0 = never: input order)
 *   uniform_fetch 0 | 1 | 2 | -1       scalar-cache fetch of wave-uniform records (1), and leaf triangles (2, default); -1: bounces 1-2 only
 *   shadow_nearest_first 0 | 1         any-hit child order: the reference's split-axis order | nearer slab entry first (default)
 *   shadow_record_order 0 | 1          shadow launches on the 64-byte quad layouts: nearest-first | entries in record order (default: the
 *                                      cheaper step wins where the VALU binds)
 *   packet_bounces n                   bounces 1..n traced by lockstep wave packets (default 0)
 *   occluder_cache_bounces n           the any-hit launches of bounces 1..n first visit the leaves that stopped the last shadow rays from the
 *                                      ray's cell of the scene (default 64; 0: off).  occluder_grid_cells c: cells along the longest extent of the
 *                                      scene (default 1024); occluder_grid_log2_cells n: table of 2^n cells x 16 bytes (default 22);
 *                                      shadow_first_look_from_bounce b: from this bounce on that first look is a dense pass of its own
 *                                      (kShadowFirstLook; default 2, 0: never).  Same image with any setting.
 *   transcendentals 0 | 1              THE ONE OPTION THAT CHANGES RESULTS (round 6; default 0).  0: sin / cos / acos / exp / pow(x, 1.5) of ray generation and the sky dome are
 *                                      the f32 rounding of a specified f64 evaluation (GPU == test oracle bit for bit); 1: the device math library's f32 functions (libm-grade) --
 *                                      WGSL's own builtins are f32 with implementation-defined ulps (wgsl:247-275,568-616).  Within SURVEY 8(d)'s tolerance of the default
 *                                      (>= 99.97 % of the pixels within 1e-3 |ref| + 1e-4 spp, image mean 5e-8), +0.2 % rays/s: off.  Set it before an accumulation's first sample.
 *   refill_min, refill_min_deep, refill_deep_from_bounce, leaf_vote, chunk, chunk_early ...
 *                                      scheduling of the persistent traversal kernel (idle lanes at which a wave refills: 40 at bounce 1 and in the any-hit
 *                                      launches, 12 from bounce 2 on -- 22 in scenes with leaves of 5 triangles or more; lanes that must still descend for the
 *                                      descend loop to go on; queue entries per cursor claim).  Same image with any setting.
 *   slot_group_shift, sample_sort, accumulate_runs, shade_blocks, reserve_samples, persistent_blocks, extra_lds
 *                                      path-slot order, accumulation kernel, grid sizes, occupancy experiments
 *   query_variant 0 | 2, query_compact 0 .. 5       kernels / record layout behind rf_renderer_intersect_rays / _occluded_rays (tests) */
RF_API int rf_renderer_set_option(rf_renderer* r, const char* name, int64_t value);
RF_API int rf_renderer_reset_stats(rf_renderer* r);
RF_API int rf_renderer_get_stats(rf_renderer* r, rf_stats* out);
/* Queue occupancy and traversal kernel time per bounce since the last reset (entry b = bounce b+1;
 * bounces past 32 are folded into entry 31).  Each array holds `capacity` entries (or is NULL);
 * *num_bounces receives the number of entries that are meaningful (the current numBounces, <= 32).
 * No reference counterpart: the megakernel has no queues. */
RF_API int rf_renderer_get_bounce_stats(rf_renderer* r, uint32_t capacity, uint64_t* closest_rays, uint64_t* shadow_rays,
                                        double* ms_closest, double* ms_shadow, uint32_t* num_bounces);

/* Multi-GPU tile sharding (no reference counterpart: the reference is single-device).  The image
 * is cut into 32x32 tiles dealt to ranks in a scrambled round-robin; each rank renders its tiles
 * into a compact tile-major float4 buffer; rf_renderer_gather_frame (below) brings the shards to one rank. */
RF_API int rf_renderer_set_tile_shard(rf_renderer* r, uint32_t rank, uint32_t world_size);
RF_API int rf_renderer_shard_tiles(rf_renderer* r, uint32_t* tile_ids /* may be NULL */, uint32_t* num_tiles);
RF_API int rf_renderer_accumulation_device_buffer(rf_renderer* r, void** device_ptr, uint64_t* bytes);
RF_API int rf_renderer_bind_accumulation_buffer(rf_renderer* r, void* device_ptr, uint64_t bytes);

/* Frame-end exchange behind the C ABI (no reference counterpart): one RCCL communicator per rank
 * (one process or host thread per GPU).  Rank 0 calls rf_comm_unique_id and hands the 128 bytes to the other ranks
 * through the host application's own channel; then every rank calls rf_comm_create (collective).
 * rf_renderer_gather_frame (collective, enqueued on the handle's stream behind the frame's kernels): every rank
 * ncclSend()s its compact tile buffer to `root`, the root posts all ncclRecv()s in one group (all xGMI ingress
 * links at once; no reduction, no ring) and un-tiles the shards into a row-major width*height float4 image in
 * device memory (*image_device_out on the root, NULL elsewhere; owned by the comm, valid until the next gather).
 * The renderer's tile shard must be (rank, world_size) of the comm.  SIDE EFFECT: what is sent is the accumulation of the
 * current parameters; if nothing has been rendered since the accumulation was last restarted (set_render_parameters, a new shard,
 * a newly bound buffer) the accumulation buffer is ZEROED on the handle's stream first (wgsl:47-49: a restarted frame starts from
 * zero), and that includes a caller-owned buffer bound with rf_renderer_bind_accumulation_buffer (its first
 * rf_renderer_accumulation_device_buffer() bytes).  RF_GATHER_LOOPBACK: the root's own shard also
 * goes through ncclSend/ncclRecv instead of being read in place (self-test of the RCCL path at world size 1). */
/* MI355X devices this process sees (hipGetDeviceCount; 0 without a GPU).  No reference counterpart (the reference asks Dawn for one adapter,
 * gpu_context.cpp); what a host application sizes `--gpus N` against. */
RF_API int rf_device_count(int32_t* count_out);
typedef struct rf_comm rf_comm;
#define RF_COMM_ID_BYTES 128
#define RF_GATHER_LOOPBACK 1u
RF_API int  rf_comm_unique_id(uint8_t id_out[RF_COMM_ID_BYTES]);
RF_API int  rf_comm_create(const uint8_t id[RF_COMM_ID_BYTES], uint32_t rank, uint32_t world_size, int32_t device_ordinal, rf_comm** out);
RF_API void rf_comm_destroy(rf_comm* c);
RF_API int  rf_renderer_gather_frame(rf_renderer* r, rf_comm* c, uint32_t root, uint32_t flags, void** image_device_out /* NULL ok */);
/* fsMain's display transform (wgsl:59-63) for a row-major float4 SUM image in device memory -- the frame
 * rf_renderer_gather_frame left on the root: num_pixels BGRA8 texels to the host, with the handle's exposure. */
RF_API int  rf_renderer_tonemap_device_image(rf_renderer* r, const void* image_device, uint64_t num_pixels, uint32_t samples, uint32_t* dst_bgra8);
/* Root: wait for the exchange and copy the gathered image to the host (width*height*4 floats, row-major, the
 * layout of rf_renderer_read_accumulation). */
RF_API int  rf_comm_read_frame(rf_comm* c, rf_renderer* r, float* dst);
/* Max over ranks of *value (timing plumbing for hosts without another collective layer; also a barrier). */
RF_API int  rf_comm_all_reduce_max(rf_comm* c, rf_renderer* r /* NULL: default stream */, double* value);
/* What RCCL reports for the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice); any pointer may be NULL.
 * A scaling line that says N GPUs carries rccl_ranks == N from here. */
RF_API int  rf_comm_info(const rf_comm* c, uint32_t* rccl_ranks, uint32_t* rccl_rank, int32_t* device_ordinal);
/* *local_out = 1: the communicator runs on the LOCAL TEST TRANSPORT, not on RCCL.  With RF_COMM_TRANSPORT=local in the environment rf_comm_unique_id makes an
 * id that rf_comm_create recognises: N communicators of ONE process (one host thread per rank, any number of them on one GPU) then execute the very plan
 * rf_gather_plan lists -- same staging offsets, same un-tile kernel -- with each ncclSend / ncclRecv pair replaced by a device-to-device copy between the ranks'
 * buffers.  It exists so that the multi-owner exchange runs on single-GPU boxes (RCCL refuses two ranks per device); rf_comm_info then reports the world the
 * caller asked for, and THIS call is how a measurement proves it did not come from such a communicator.  No reference counterpart
 * (single-device: reference_path_tracer.cpp:565-595). */
RF_API int  rf_comm_transport(const rf_comm* c, uint32_t* local_out);
/* Device time of this rank's LAST rf_renderer_gather_frame (HIP events on the handle's stream around the sends / receives and, on the root, the un-tile):
 * what the one exchange of the multi-GPU path costs once the rank's own frame has drained.  Waits for that exchange; -1 before the first.  (No reference
 * counterpart: the reference is single-device, reference_path_tracer.cpp:565-595.) */
RF_API int  rf_comm_last_exchange_ms(rf_comm* c, double* ms_out);
/* Device memory held by a handle: path state + queues (148 B per path slot: eight packed xyz streams, two float4 streams, five u32 queues / lists; allocated on demand for the largest batch traced),
 * the batch depth in use (lowered automatically when the device has less free memory than the default wants: same image,
 * more batches) and the resident scene.  Any pointer may be NULL. */
RF_API int  rf_renderer_memory_info(const rf_renderer* r, uint64_t* path_state_bytes, uint64_t* paths_allocated, uint64_t* max_paths_per_batch, uint64_t* scene_bytes);
/* Which BVH record layout the handle reads in the closest-hit / any-hit launch of each bounce -- what it picked BY ITSELF for this scene at upload (from the
 * binary16 surface-area ratio of the boxes, the tree's size against the Infinity Cache, the median leaf size against the sun disc) plus any option set since.
 * No reference counterpart (the reference has one node layout, bvh.hpp:14-21); lets a caller -- and the parity tests -- see that a result was produced by the
 * layouts the renderer would use on its own.  Layout codes: RF_LAYOUT_*. */
enum { RF_LAYOUT_BINARY = 0, RF_LAYOUT_COMPACT = 1, RF_LAYOUT_HOT = 2, RF_LAYOUT_QUAD = 3, RF_LAYOUT_QUAD_HALF = 4, RF_LAYOUT_QUAD_LOCAL = 5, RF_LAYOUT_OCT = 6,
       RF_LAYOUT_SCALAR = 7, RF_LAYOUT_PACKET = 8 };
typedef struct rf_layout_info
{
    uint32_t closest_layout[16];      /* bounce 1..16 */
    uint32_t shadow_layout[16];
    uint32_t shadow_cached[16];       /* 1: the any-hit launch of that bounce starts at the occluder cache's entries */
    uint32_t occluder_hint_levels;    /* 0: the cache remembers leaves; n: a record n quad levels above the leaf */
    uint32_t shadow_first_look_from_bounce;
    uint32_t dense_leaf_min;          /* leaf phases with a leaf of this many triangles or more run over dense (lane, triangle) pairs; 0: never */
    uint32_t legacy_layouts_compiled; /* 1: a build with RF_EXP_LEGACY_LAYOUTS (the compact-capable / 32-byte records and the packet kernel exist) */
    float    quad_half_area_ratio;
    float    reserved;
    uint64_t tree_bytes;              /* quad records + triangle records */
} rf_layout_info;
RF_API int  rf_renderer_layout_info(const rf_renderer* r, rf_layout_info* out);
/* Host helpers (no GPU needed). */
/* The point-to-point operations rank `rank` posts (one RCCL group) for a gather to `root`: exactly the list
 * rf_renderer_gather_frame executes.  Offsets / counts in tiles (1024 float4): a receive lands at offset_tiles of the root's
 * staging area (rf_gather_layout), a send starts at offset_tiles of the rank's own compact buffer.  ops == NULL: count query. */
typedef struct rf_gather_op
{
    uint32_t is_send, peer, offset_tiles, count_tiles;
} rf_gather_op;
RF_API int rf_gather_plan(uint32_t width, uint32_t height, uint32_t world_size, uint32_t rank, uint32_t root, uint32_t flags, rf_gather_op* ops, uint32_t* num_ops);
/* The staging layout the gather uses: shards rank after rank, each rank's tiles in ascending tile id.
 * rank_first_tile[world_size + 1], tile_slot[tiles] (staging position of a tile, in tiles), tile_owner[tiles]. */
RF_API int rf_gather_layout(uint32_t width, uint32_t height, uint32_t world_size, uint32_t* rank_first_tile, uint32_t* tile_slot, uint32_t* tile_owner);
RF_API int rf_tiles_for_rank(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, uint32_t* tile_ids, uint32_t* num_tiles);
RF_API int rf_untile(const float* compact, const uint32_t* tile_ids, uint32_t num_tiles, uint32_t width, uint32_t height, float* image);

/* ---------------------------------------------------------------------------------------------
 * BVH queries on the GPU
 * ------------------------------------------------------------------------------------------ */
/* The bvh-visualizer pixel loop (src/bvh-visualizer/main.cpp:60-78): pinhole camera
 * (generateCameraRay, src/common/camera.cpp:44-52), u = j/W, v = 1-(i+1)/H, tMax = FLT_MAX.
 * nodes_visited[W*H] row-major is bit-exact with the CPU reference's BvhStats::nodesVisited. */
RF_API int rf_renderer_trace_primary_stats(rf_renderer* r, const rf_camera* camera, uint32_t width, uint32_t height,
                                           uint32_t* nodes_visited, uint8_t* hit /* NULL ok */, float* t /* NULL ok */,
                                           uint32_t* triangle_tests /* NULL ok */);
/* bool rayIntersectBvh(const Ray&, span<BvhNode>, span<Positions>, float tMax, Intersection&,
 * BvhStats*) (src/common/ray_intersection.hpp:43-49) for a batch of rays (6 floats each: origin,
 * direction).  triangle[i] = 0xFFFFFFFF on a miss. */
RF_API int rf_renderer_intersect_rays(rf_renderer* r, const float* rays6, uint64_t num_rays, float t_max, uint32_t* triangle,
                                      float* t, float* uv, float* p, uint32_t* nodes_visited, uint32_t* triangle_tests);
/* shadowRay (wgsl:321-368): visibility[i] = 1.0 if nothing is hit, else 0.0. */
RF_API int rf_renderer_occluded_rays(rf_renderer* r, const float* rays6, uint64_t num_rays, float t_max, float* visibility);

/* ---------------------------------------------------------------------------------------------
 * BVH queries on the HOST (no GPU needed; re-entrant pure functions)
 * ------------------------------------------------------------------------------------------ */
/* nlrs::Intersection {p, t} (src/common/ray_intersection.hpp:15-19) plus the triangle hit and its barycentrics. */
typedef struct rf_intersection
{
    float    p[3];     /* offset hit point (offsetRay, ray_intersection.cpp:17-35) */
    float    t;
    uint32_t triangle; /* index into the triangle array (BVH leaf order); 0xFFFFFFFF on a miss */
    float    u, v;
} rf_intersection;
/* nlrs::BvhStats (ray_intersection.hpp:38-41) plus triangle tests and the high-water mark of the pending-node list. */
typedef struct rf_bvh_stats
{
    uint32_t nodes_visited, triangle_tests, stack_high_water;
} rf_bvh_stats;
/* bool rayIntersectBvh(const Ray&, span<const BvhNode>, span<const Positions>, float tMax, Intersection&, BvhStats* = nullptr)
 * (src/common/ray_intersection.hpp:43-49, .cpp:138-213): the reference's CPU query -- focus picking (src/pt/main.cpp:214-225),
 * bvh-visualizer (src/bvh-visualizer/main.cpp:60-78) -- on the host, bit-identical hit / t / p / nodesVisited.
 * positions: num_triangles records of position_stride_bytes = 36 (Positions, the .pt file's bvhPositionAttributes) or
 * 48 (PositionAttribute).  *hit_out = 1 / 0 replaces the bool; stats may be NULL.  Pending far children are kept in a list
 * that grows on demand (the reference's 32-entry array is overrun past depth 32).  Malformed links -> RF_ERROR_RUNTIME. */
RF_API int rf_intersect_bvh(const float ray6[6], const void* nodes48, uint64_t num_nodes, const void* positions, uint32_t position_stride_bytes,
                            uint64_t num_triangles, float t_max, rf_intersection* out, rf_bvh_stats* stats /* NULL ok */, int* hit_out);
/* The same for num_rays rays on num_threads host threads (0 = all hardware threads; static blocks of rays).
 * hit[i] = 1 / 0; any output array may be NULL. */
RF_API int rf_intersect_bvh_batch(const float* rays6, uint64_t num_rays, const void* nodes48, uint64_t num_nodes, const void* positions,
                                  uint32_t position_stride_bytes, uint64_t num_triangles, float t_max, uint32_t num_threads, uint8_t* hit,
                                  rf_intersection* out, rf_bvh_stats* stats);
/* The bvh-visualizer pixel loop on the host (src/bvh-visualizer/main.cpp:60-78; the CPU twin of
 * rf_renderer_trace_primary_stats): rows [row_begin, row_end) of a width x height grid, u = j/W, v = 1-(i+1)/H, tMax = FLT_MAX,
 * static blocks of scanlines over num_threads threads (0 = all; 1 = what the reference does).  Outputs are indexed
 * i*width + j over the whole grid; any of them may be NULL. */
RF_API int rf_bvh_visualizer_pass(const rf_camera* camera, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, const void* nodes48,
                                  uint64_t num_nodes, const void* positions, uint32_t position_stride_bytes, uint64_t num_triangles, uint32_t num_threads,
                                  uint32_t* nodes_visited, uint8_t* hit, float* t, uint32_t* triangle_tests);

/* ---------------------------------------------------------------------------------------------
 * CPU-side scene preparation (host code, runs without a GPU)
 * ------------------------------------------------------------------------------------------ */
/* Bvh buildBvh(std::span<const Positions>) (src/common/bvh.hpp:33, bvh.cpp:263-291).
 * nodes_out: room for 2*num_triangles 48-B nodes; triangle_indices_out[src] = leaf-order index. */
RF_API int rf_build_bvh(const float* positions36, uint64_t num_triangles, void* nodes_out, uint64_t* num_nodes_out,
                        uint64_t* triangle_indices_out, int32_t* depth_out /* NULL ok */);

/* The same build on the GPU (device_ordinal): identical node bytes and identical triangle_indices_out
 * (the order inside multi-triangle leaves decides closest-hit ties between coincident triangles; the
 * reference's is whatever its standard library's std::partition leaves -- rf_build_bvh uses libstdc++'s,
 * and the GPU builder reproduces that permutation).  Replaces the
 * single-threaded recursion of src/common/bvh.cpp:81-260 for large scenes.
 * build_ms_out (NULL ok): device time of the build, triangles already resident.  Fails without a
 * GPU (no CPU fallback: call rf_build_bvh). */
RF_API int rf_build_bvh_gpu(const float* positions36, uint64_t num_triangles, void* nodes_out, uint64_t* num_nodes_out,
                            uint64_t* triangle_indices_out, int32_t* depth_out /* NULL ok */, int32_t device_ordinal,
                            float* build_ms_out /* NULL ok */);

/* Host-only self-check of the render path's BVH record layouts for a flattened tree of 48-B nodes: the 64-byte
 * "children in the parent" records, the compact-capable records and the 32-byte records are built as rf_renderer_create builds
 * them and every variant must decode to the same child planes and child words.  Also checked: the leaf boxes and the occluder-cache
 * entries written into the triangle records (every entry is 0 or the index of a quad record that really lies above its leaf, 1 to 3 levels).  *flags_out: bit 0 = boxes regular (wide
 * layout usable), bit 1 = compact-capable records usable, bit 2 = 32-byte records usable.  No reference counterpart. */
RF_API int rf_check_wide_layouts(const void* nodes48, uint64_t num_nodes, uint32_t* flags_out);
/* The same check, plus (bit 3 = quad records usable, bit 4 = half-precision quad records usable) the figure the renderer's default
 * layout choice rests on: the surface area of the half-precision (binary16, conservative) child boxes relative to the exact ones,
 * summed over the tree -- the closest-hit launches read the half-precision records where it is <= 1.075 and the local-grid records
 * (bit 5, from bounce 2) beyond; the shadow launches the local-grid records from bounce 2 where it is <= 1.10 (bounce 1: the half-precision
 * records where they suit) and the exact quad records otherwise.  Either output may be NULL. */
RF_API int rf_wide_layout_stats(const void* nodes48, uint64_t num_nodes, uint32_t* flags_out, float* quad_half_area_ratio);

/* Camera createCamera(origin, lookAt, aperture, focusDistance, vfov, aspectRatio)
 * (src/common/camera.cpp:7-42); vfov in radians (Angle::asRadians). */
RF_API int rf_create_camera(const float origin[3], const float look_at[3], float aperture, float focus_distance,
                            float vfov_radians, float aspect_ratio, rf_camera* out);
/* FlyCameraController::getCamera (src/pt/fly_camera_controller.cpp:12-22,138-148). */
RF_API int rf_fly_camera(const float position[3], float yaw_degrees, float pitch_degrees, float vfov_degrees, float aperture,
                         float focus_distance, float aspect_ratio, rf_camera* out);
/* The camera lambda of src/bvh-visualizer/main.cpp:36-55 for a 48-B root node. */
RF_API int rf_bvh_visualizer_camera(const void* root_node48, float aspect_ratio, rf_camera* out);

/* sky_state_new / sky_state_radiance (src/hw-skymodel/hw_skymodel.h:35,44); state33 = params[27],
 * sky_radiances[3], solar_radiances[3].  Returns the reference's sky_state_result value. */
RF_API int   rf_sky_state_new(float elevation, float turbidity, const float albedo[3], float state33[33]);
RF_API float rf_sky_state_radiance(const float state33[33], float theta, float gamma, int channel);
/* AlignedSkyState(const Sky&) (src/pt/aligned_sky_state.hpp:44-70): 40 floats. */
RF_API int rf_aligned_sky_state(const rf_sky* sky, float out40[40]);

/* ---------------------------------------------------------------------------------------------
 * .pt scene files  (replaces nlrs::PtFormat + serialize/deserialize, src/pt-format/pt_format.hpp:18-43)
 * ------------------------------------------------------------------------------------------ */
typedef struct rf_pt_format rf_pt_format;

typedef struct rf_pt_format_view
{
    const void*     bvh_nodes;                    uint64_t num_bvh_nodes;
    const void*     bvh_position_attributes;      uint64_t num_bvh_position_attributes;      /* 36 B */
    const void*     triangle_position_attributes; uint64_t num_triangle_position_attributes; /* 48 B */
    const void*     triangle_vertex_attributes;   uint64_t num_triangle_vertex_attributes;   /* 80 B */
    const float*    vertex_positions;             uint64_t num_vertex_positions;             /* vec4 */
    const float*    vertex_normals;               uint64_t num_vertex_normals;               /* vec4 */
    const float*    vertex_tex_coords;            uint64_t num_vertex_tex_coords;            /* vec2 */
    const uint32_t* vertex_indices;               uint64_t num_vertex_indices;
    const uint64_t* model_vertex_positions;       uint64_t num_model_vertex_positions;       /* {offset,count} pairs */
    const uint64_t* model_vertex_normals;         uint64_t num_model_vertex_normals;
    const uint64_t* model_vertex_tex_coords;      uint64_t num_model_vertex_tex_coords;
    const uint64_t* model_vertex_indices;         uint64_t num_model_vertex_indices;
    const uint32_t* model_base_color_texture_indices; uint64_t num_model_base_color_texture_indices;
    uint64_t        num_textures;
} rf_pt_format_view;

/* PtFormat(std::filesystem::path gltfPath) (pt_format.cpp:20-151): glTF/GLB -> BVH + GPU arrays. */
/* BVH builder used by rf_pt_format_from_gltf / _from_triangles: -1 = host (default), >= 0 = rf_build_bvh_gpu on that device. */
RF_API int rf_pt_format_set_bvh_builder(int32_t gpu_device_or_minus_one);
RF_API int rf_pt_format_from_gltf(const char* gltf_path, rf_pt_format** out);
/* deserialize(InputStream&, PtFormat&) (pt_format.cpp:271-321) from a file / from memory.
 * Wrong magic -> RF_ERROR_RUNTIME with the reference's exact messages (src/tests/pt_format.cpp:192-210). */
RF_API int rf_pt_format_load(const char* pt_path, rf_pt_format** out);
RF_API int rf_pt_format_deserialize(const void* data, uint64_t size, rf_pt_format** out);
/* serialize(OutputStream&, const PtFormat&) (pt_format.cpp:240-269). */
RF_API int rf_pt_format_save(const rf_pt_format* f, const char* pt_path);
RF_API int rf_pt_format_serialize(const rf_pt_format* f, void* dst /* NULL = size query */, uint64_t* size);
/* Build a PtFormat from caller arrays (triangle soup in source order + textures): runs buildBvh +
 * reorderAttributes + the GPU-layout packing of pt_format.cpp:40-79.  Raster-mesh arrays are left
 * empty.  Used by the synthetic-scene generator. */
RF_API int rf_pt_format_from_triangles(const float* positions36, const float* normals36, const float* tex_coords24,
                                       const uint32_t* texture_indices, uint64_t num_triangles, const rf_texture* textures,
                                       uint64_t num_textures, rf_pt_format** out);
/* Texture::fromMemory (src/common/texture.hpp:41, texture.cpp:12-54): PNG or JPEG bytes -> width*height
 * BGRA8 texels packed as u32 (b | g<<8 | r<<16 | 255<<24).  pixels == NULL: size query. */
RF_API int  rf_texture_from_memory(const void* data, uint64_t size, uint32_t* width, uint32_t* height, uint32_t* pixels);
RF_API int  rf_pt_format_view_get(const rf_pt_format* f, rf_pt_format_view* out);
RF_API int  rf_pt_format_texture(const rf_pt_format* f, uint64_t index, rf_texture* out);
RF_API void rf_pt_format_destroy(rf_pt_format* f);
/* Fill an rf_scene (the four spans of nlrs::Scene, src/pt/main.cpp:150-157) from a PtFormat.
 * textures_out must hold num_textures entries. */
RF_API int rf_pt_format_scene(const rf_pt_format* f, rf_scene* scene_out, rf_texture* textures_out);

#ifdef __cplusplus
}
#endif
#endif /* RAYFINDER_AMD_H */
