// rf_c_api.cpp -- the extern "C" boundary declared in include/rayfinder_amd.h.
// Exceptions never cross it: they become status codes + a thread-local message.
#include "../../include/rayfinder_amd.h"

#include "rf_bvh.hpp"
#include "rf_bvh_gpu.hpp"
#include "rf_camera.hpp"
#include "rf_comm.hpp"
#include "rf_gltf.hpp"
#include "rf_pt_format.hpp"
#include "rf_query.hpp"
#include "rf_renderer.hpp"
#include "rf_sky.hpp"

#include <cstring>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

static_assert(sizeof(rf_camera) == sizeof(rf::Camera));

struct rf_renderer
{
    std::unique_ptr<rf::Renderer> impl;
};

struct rf_comm
{
    std::unique_ptr<rf::TileComm> impl;
};

struct rf_pt_format
{
    rf::PtFormat format;
};

namespace
{
thread_local std::string gLastError;

struct NoDevice : std::runtime_error
{
    using std::runtime_error::runtime_error;
};

template<typename F>
int guarded(F&& f)
{
    try
    {
        gLastError.clear();
        return f();
    }
    catch (const std::invalid_argument& e)
    {
        gLastError = e.what();
        return RF_ERROR_INVALID_ARGUMENT;
    }
    catch (const std::bad_alloc&)
    {
        gLastError = "out of host memory";
        return RF_ERROR_RUNTIME;
    }
    catch (const std::exception& e)
    {
        gLastError = e.what();
        if (gLastError.find("no HIP device") != std::string::npos) return RF_ERROR_NO_DEVICE;
        if (gLastError.find("sky parameters out of range") != std::string::npos) return RF_ERROR_OUT_OF_RANGE;
        return RF_ERROR_RUNTIME;
    }
    catch (...)
    {
        gLastError = "Unknown exception occurred.";
        return RF_ERROR_RUNTIME;
    }
}

void require(bool cond, const char* what)
{
    if (!cond) throw std::invalid_argument(what);
}

rf::Vec3 v3(const float* p) { return rf::vec3(p[0], p[1], p[2]); }

rf::Camera toCamera(const rf_camera& c)
{
    rf::Camera out;
    std::memcpy(&out, &c, sizeof out);
    return out;
}

rf::RenderParameters toParams(const rf_render_parameters& p)
{
    rf::RenderParameters out;
    out.width = p.width;
    out.height = p.height;
    out.camera = toCamera(p.camera);
    out.samplingParams.numSamplesPerPixel = p.num_samples_per_pixel;
    out.samplingParams.numBounces = p.num_bounces;
    out.sky.turbidity = p.sky.turbidity;
    std::memcpy(out.sky.albedo, p.sky.albedo, sizeof out.sky.albedo);
    out.sky.sunZenithDegrees = p.sky.sun_zenith_degrees;
    out.sky.sunAzimuthDegrees = p.sky.sun_azimuth_degrees;
    out.exposure = p.exposure;
    require(p.width > 0 && p.height > 0, "framebuffer size must be non-zero");
    require(p.num_samples_per_pixel > 0, "num_samples_per_pixel must be > 0");
    require(p.num_bounces > 0, "num_bounces must be > 0");
    return out;
}
} // namespace

extern "C" {

const char* rf_last_error_message(void) { return gLastError.c_str(); }
const char* rf_version(void) { return "rayfinder_amd 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------- renderer
int rf_renderer_create(const rf_renderer_descriptor* desc, const rf_scene* scene, rf_renderer** out)
{
    return guarded([&] {
        require(desc && scene && out, "null argument");
        require(scene->bvh_nodes && scene->num_bvh_nodes > 0, "scene has no BVH nodes");
        require(scene->num_triangles == 0 || (scene->position_attributes && scene->vertex_attributes), "null triangle arrays");
        rf::RendererDescriptor d;
        d.renderParams = toParams(desc->render_params);
        d.maxWidth = desc->max_width;
        d.maxHeight = desc->max_height;
        d.deviceOrdinal = desc->device_ordinal;
        d.maxPathsInFlight = desc->max_paths_in_flight;
        std::vector<rf::TextureView> textures;
        for (uint64_t i = 0; i < scene->num_textures; ++i)
        {
            const rf_texture& t = scene->base_color_textures[i];
            require(t.pixels && t.width && t.height, "empty texture");
            textures.push_back({t.pixels, t.width, t.height});
        }
        rf::SceneView view;
        view.bvhNodes = {static_cast<const rf::BvhNode*>(scene->bvh_nodes), static_cast<size_t>(scene->num_bvh_nodes)};
        view.positionAttributes = {static_cast<const rf::PositionAttribute*>(scene->position_attributes), static_cast<size_t>(scene->num_triangles)};
        view.vertexAttributes = {static_cast<const rf::VertexAttributes*>(scene->vertex_attributes), static_cast<size_t>(scene->num_triangles)};
        view.baseColorTextures = textures;
        auto handle = std::make_unique<rf_renderer>();
        handle->impl = std::make_unique<rf::Renderer>(d, view);
        *out = handle.release();
        return RF_OK;
    });
}

void rf_renderer_destroy(rf_renderer* r) { delete r; }

int rf_renderer_set_render_parameters(rf_renderer* r, const rf_render_parameters* params)
{
    return guarded([&] {
        require(r && params, "null argument");
        r->impl->setRenderParameters(toParams(*params));
        return RF_OK;
    });
}

int rf_renderer_render(rf_renderer* r, uint32_t num_frames)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->render(num_frames);
        return RF_OK;
    });
}

int rf_renderer_synchronize(rf_renderer* r)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->synchronize();
        return RF_OK;
    });
}

float rf_renderer_average_renderpass_duration_ms(rf_renderer* r)
{
    float v = 0.0f;
    guarded([&] {
        require(r, "null argument");
        v = r->impl->averageRenderpassDurationMs();
        return RF_OK;
    });
    return v;
}

float rf_renderer_render_progress_percentage(const rf_renderer* r) { return r ? r->impl->renderProgressPercentage() : 0.0f; }

int rf_renderer_read_accumulation(rf_renderer* r, float* dst, uint32_t* accumulated)
{
    return guarded([&] {
        require(r && dst, "null argument");
        r->impl->readAccumulation(dst);
        if (accumulated) *accumulated = r->impl->accumulatedSampleCount();
        return RF_OK;
    });
}

int rf_renderer_read_tonemapped(rf_renderer* r, uint32_t* dst)
{
    return guarded([&] {
        require(r && dst, "null argument");
        r->impl->readTonemapped(dst);
        return RF_OK;
    });
}

int rf_renderer_render_deferred(rf_renderer* r, uint32_t num_frames)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->renderDeferred(num_frames);
        return RF_OK;
    });
}

int rf_renderer_reset_deferred(rf_renderer* r)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->resetDeferred();
        return RF_OK;
    });
}

int rf_renderer_read_deferred(rf_renderer* r, float* sample_rgb, float* accumulation_rgb, uint32_t* bgra8, uint32_t* frame_count)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->readDeferred(sample_rgb, accumulation_rgb, bgra8);
        if (frame_count) *frame_count = r->impl->deferredFrameCount();
        return RF_OK;
    });
}

int rf_renderer_set_counting(rf_renderer* r, int enabled)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->setCounting(enabled != 0);
        return RF_OK;
    });
}

int rf_renderer_set_option(rf_renderer* r, const char* name, int64_t value)
{
    return guarded([&] {
        require(r && name, "null argument");
        r->impl->setOption(name, value);
        return RF_OK;
    });
}

int rf_renderer_set_timing(rf_renderer* r, int enabled)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->setTiming(enabled != 0);
        return RF_OK;
    });
}

int rf_renderer_reset_stats(rf_renderer* r)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->resetStats();
        return RF_OK;
    });
}

int rf_renderer_get_bounce_stats(rf_renderer* r, uint32_t capacity, uint64_t* closest_rays, uint64_t* shadow_rays, double* ms_closest,
                                 double* ms_shadow, uint32_t* num_bounces)
{
    return guarded([&] {
        require(r != nullptr, "null argument");
        const rf::RenderStats s = r->impl->stats();
        const uint32_t        n = std::min<uint32_t>(r->impl->numBounces(), rf::RenderStats::kMaxBounceStats);
        if (num_bounces) *num_bounces = n;
        for (uint32_t b = 0; b < capacity; ++b)
        {
            const bool in = b < rf::RenderStats::kMaxBounceStats;
            if (closest_rays) closest_rays[b] = in ? s.closestRaysByBounce[b] : 0;
            if (shadow_rays) shadow_rays[b] = in ? s.shadowRaysByBounce[b] : 0;
            if (ms_closest) ms_closest[b] = in ? s.msClosestByBounce[b] : 0.0;
            if (ms_shadow) ms_shadow[b] = in ? s.msShadowByBounce[b] : 0.0;
        }
        return RF_OK;
    });
}

int rf_renderer_get_stats(rf_renderer* r, rf_stats* out)
{
    return guarded([&] {
        require(r && out, "null argument");
        const rf::RenderStats s = r->impl->stats();
        std::memset(out, 0, sizeof *out);
        out->primary_rays = s.primaryRays;
        out->closest_rays = s.closestRays;
        out->shadow_rays = s.shadowRays;
        out->closest_node_visits = s.closestNodeVisits;
        out->closest_triangle_tests = s.closestTriangleTests;
        out->shadow_node_visits = s.shadowNodeVisits;
        out->shadow_triangle_tests = s.shadowTriangleTests;
        out->paths = s.paths;
        out->stack_high_water = s.stackHighWater;
        out->batch_samples_used = s.batchSamplesUsed;
        out->batches_traced = s.batchesTraced;
        out->ms_raygen = s.msRaygen;
        out->ms_closest = s.msClosest;
        out->ms_shade = s.msShade;
        out->ms_shadow = s.msShadow;
        out->ms_accumulate = s.msAccumulate;
        out->launches_raygen = s.launchesRaygen;
        out->launches_closest = s.launchesClosest;
        out->launches_shade = s.launchesShade;
        out->launches_shadow = s.launchesShadow;
        out->launches_accumulate = s.launchesAccumulate;
        out->closest_record_fetches = s.closestRecordFetches;
        out->shadow_record_fetches = s.shadowRecordFetches;
        out->abandoned_rays = s.abandonedRays;
        out->scalar_redo_rays = s.scalarRedoRays;
        out->shadow_rays_hint_answered = s.shadowRaysHintAnswered;
        out->shadow_rays_self_answered = s.shadowRaysSelfAnswered;
        return RF_OK;
    });
}

int rf_renderer_set_tile_shard(rf_renderer* r, uint32_t rank, uint32_t world_size)
{
    return guarded([&] {
        require(r, "null argument");
        require(world_size > 0 && rank < world_size, "invalid rank / world size");
        r->impl->setTileShard(rank, world_size);
        return RF_OK;
    });
}

int rf_renderer_shard_tiles(rf_renderer* r, uint32_t* tile_ids, uint32_t* num_tiles)
{
    return guarded([&] {
        require(r && num_tiles, "null argument");
        const auto tiles = r->impl->shardTiles();
        *num_tiles = static_cast<uint32_t>(tiles.size());
        if (tile_ids) std::memcpy(tile_ids, tiles.data(), tiles.size() * sizeof(uint32_t));
        return RF_OK;
    });
}

int rf_renderer_accumulation_device_buffer(rf_renderer* r, void** device_ptr, uint64_t* bytes)
{
    return guarded([&] {
        require(r && device_ptr && bytes, "null argument");
        *device_ptr = r->impl->accumulationDevicePointer();
        *bytes = r->impl->accumulationBytes();
        return RF_OK;
    });
}

int rf_renderer_bind_accumulation_buffer(rf_renderer* r, void* device_ptr, uint64_t bytes)
{
    return guarded([&] {
        require(r, "null argument");
        r->impl->bindAccumulationBuffer(device_ptr, bytes);
        return RF_OK;
    });
}

int rf_device_count(int32_t* count_out)
{
    return guarded([&] {
        require(count_out, "null argument");
        *count_out = rf::deviceCount();
        return RF_OK;
    });
}

int rf_comm_unique_id(uint8_t id_out[RF_COMM_ID_BYTES])
{
    return guarded([&] {
        require(id_out, "null argument");
        rf::TileComm::uniqueId(id_out);
        return RF_OK;
    });
}

int rf_comm_create(const uint8_t id[RF_COMM_ID_BYTES], uint32_t rank, uint32_t world_size, int32_t device_ordinal, rf_comm** out)
{
    return guarded([&] {
        require(id && out, "null argument");
        require(world_size > 0 && rank < world_size, "invalid rank / world size");
        auto h = std::make_unique<rf_comm>();
        h->impl = std::make_unique<rf::TileComm>(id, rank, world_size, device_ordinal);
        *out = h.release();
        return RF_OK;
    });
}

void rf_comm_destroy(rf_comm* c) { delete c; }

int rf_renderer_gather_frame(rf_renderer* r, rf_comm* c, uint32_t root, uint32_t flags, void** image_device_out)
{
    return guarded([&] {
        require(r && c, "null argument");
        require(root < c->impl->worldSize(), "gather root out of range");
        require(r->impl->shardRank() == c->impl->rank() && r->impl->shardWorldSize() == c->impl->worldSize(),
                "the renderer's tile shard differs from the communicator's rank / world size (call rf_renderer_set_tile_shard first)");
        // the exchange is enqueued on the renderer's stream with the communicator's device current: they must be one device
        require(r->impl->deviceOrdinal() == c->impl->deviceOrdinal(), "the renderer and the communicator are on different devices");
        r->impl->clearAccumulationIfStale(); // nothing rendered since the last reset: send zeros, not the previous frame
        const void* image = c->impl->gatherFrame(r->impl->accumulationDevicePointer(), r->impl->width(), r->impl->height(), root, r->impl->streamHandle(),
                                                 (flags & RF_GATHER_LOOPBACK) != 0);
        if (image_device_out) *image_device_out = const_cast<void*>(image);
        return RF_OK;
    });
}

int rf_renderer_tonemap_device_image(rf_renderer* r, const void* image_device, uint64_t num_pixels, uint32_t samples, uint32_t* dst)
{
    return guarded([&] {
        require(r && image_device && dst, "null argument");
        require(samples > 0, "samples must be > 0");
        r->impl->tonemapDeviceImage(image_device, num_pixels, samples, dst);
        return RF_OK;
    });
}

int rf_comm_read_frame(rf_comm* c, rf_renderer* r, float* dst)
{
    return guarded([&] {
        require(c && r && dst, "null argument");
        c->impl->readFrame(dst, r->impl->streamHandle());
        return RF_OK;
    });
}

int rf_comm_all_reduce_max(rf_comm* c, rf_renderer* r, double* value)
{
    return guarded([&] {
        require(c && value, "null argument");
        *value = c->impl->allReduceMax(*value, r ? r->impl->streamHandle() : nullptr);
        return RF_OK;
    });
}

namespace
{
rf::TriangleSpan triangleSpan(const void* positions, uint32_t stride, uint64_t count)
{
    require(stride == 36 || stride == 48, "position_stride_bytes must be 36 (Positions) or 48 (PositionAttribute)");
    require(positions != nullptr || count == 0, "null triangle array");
    return rf::TriangleSpan{static_cast<const uint8_t*>(positions), count, stride};
}
} // namespace

int rf_intersect_bvh(const float ray6[6], const void* nodes48, uint64_t num_nodes, const void* positions, uint32_t position_stride_bytes, uint64_t num_triangles,
                     float t_max, rf_intersection* out, rf_bvh_stats* stats, int* hit_out)
{
    return guarded([&] {
        require(ray6 && nodes48 && out && hit_out, "null argument");
        require(num_nodes > 0, "scene has no BVH nodes");
        static_assert(sizeof(rf_intersection) == sizeof(rf::HostIntersection) && sizeof(rf_bvh_stats) == sizeof(rf::HostBvhStats));
        rf::HostIntersection h{};
        rf::HostBvhStats     st{};
        const bool           hit = rf::intersectBvh(rf::vec3(ray6[0], ray6[1], ray6[2]), rf::vec3(ray6[3], ray6[4], ray6[5]),
                                                    std::span<const rf::BvhNode>(static_cast<const rf::BvhNode*>(nodes48), num_nodes),
                                                    triangleSpan(positions, position_stride_bytes, num_triangles), t_max, h, &st);
        std::memcpy(out, &h, sizeof h);
        if (stats) std::memcpy(stats, &st, sizeof st);
        *hit_out = hit ? 1 : 0;
        return RF_OK;
    });
}

int rf_intersect_bvh_batch(const float* rays6, uint64_t num_rays, const void* nodes48, uint64_t num_nodes, const void* positions, uint32_t position_stride_bytes,
                           uint64_t num_triangles, float t_max, uint32_t num_threads, uint8_t* hit, rf_intersection* out, rf_bvh_stats* stats)
{
    return guarded([&] {
        require((rays6 || num_rays == 0) && nodes48, "null argument");
        require(num_nodes > 0, "scene has no BVH nodes");
        rf::intersectBvhBatch(rays6, num_rays, std::span<const rf::BvhNode>(static_cast<const rf::BvhNode*>(nodes48), num_nodes),
                              triangleSpan(positions, position_stride_bytes, num_triangles), t_max, num_threads, hit, reinterpret_cast<rf::HostIntersection*>(out),
                              reinterpret_cast<rf::HostBvhStats*>(stats));
        return RF_OK;
    });
}

int rf_bvh_visualizer_pass(const rf_camera* camera, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end, const void* nodes48, uint64_t num_nodes,
                           const void* positions, uint32_t position_stride_bytes, uint64_t num_triangles, uint32_t num_threads, uint32_t* nodes_visited, uint8_t* hit,
                           float* t, uint32_t* triangle_tests)
{
    return guarded([&] {
        require(camera && nodes48, "null argument");
        require(num_nodes > 0, "scene has no BVH nodes");
        require(width > 0 && height > 0, "empty image");
        rf::Camera cam;
        static_assert(sizeof(rf::Camera) == sizeof(rf_camera));
        std::memcpy(&cam, camera, sizeof cam);
        rf::bvhVisualizerPass(cam, width, height, row_begin, row_end, std::span<const rf::BvhNode>(static_cast<const rf::BvhNode*>(nodes48), num_nodes),
                              triangleSpan(positions, position_stride_bytes, num_triangles), num_threads, nodes_visited, hit, t, triangle_tests);
        return RF_OK;
    });
}

int rf_comm_info(const rf_comm* c, uint32_t* rccl_ranks, uint32_t* rccl_rank, int32_t* device_ordinal)
{
    return guarded([&] {
        require(c, "null argument");
        uint32_t count = 0, user = 0;
        int      dev = 0;
        c->impl->rcclInfo(count, user, dev);
        if (rccl_ranks) *rccl_ranks = count;
        if (rccl_rank) *rccl_rank = user;
        if (device_ordinal) *device_ordinal = dev;
        return RF_OK;
    });
}

int rf_comm_transport(const rf_comm* c, uint32_t* local_out)
{
    return guarded([&] {
        require(c && local_out, "null argument");
        *local_out = c->impl->localTransport() ? 1u : 0u;
        return RF_OK;
    });
}

int rf_comm_last_exchange_ms(rf_comm* c, double* ms_out)
{
    return guarded([&] {
        require(c && ms_out, "null argument");
        *ms_out = c->impl->lastExchangeMs();
        return RF_OK;
    });
}

int rf_gather_plan(uint32_t width, uint32_t height, uint32_t world_size, uint32_t rank, uint32_t root, uint32_t flags, rf_gather_op* ops, uint32_t* num_ops)
{
    return guarded([&] {
        require(width > 0 && height > 0 && world_size > 0, "empty frame or world");
        require(rank < world_size && root < world_size, "rank / root out of range");
        require(num_ops, "null argument");
        static_assert(sizeof(rf_gather_op) == sizeof(rf::GatherOp));
        const rf::GatherLayout          g = rf::gatherLayout(width, height, world_size);
        const std::vector<rf::GatherOp> plan = rf::gatherPlan(g, world_size, rank, root, (flags & RF_GATHER_LOOPBACK) != 0);
        if (ops)
        {
            require(*num_ops >= plan.size(), "ops array too small");
            std::memcpy(ops, plan.data(), plan.size() * sizeof(rf::GatherOp));
        }
        *num_ops = static_cast<uint32_t>(plan.size());
        return RF_OK;
    });
}

int rf_renderer_layout_info(const rf_renderer* r, rf_layout_info* out)
{
    return guarded([&] {
        require(r && out, "null argument");
        uint32_t layouts[48], misc[4];
        float    ratio = 0.0f;
        uint64_t tree = 0;
        r->impl->layoutInfo(layouts, misc, ratio, tree);
        for (int i = 0; i < 16; ++i) out->closest_layout[i] = layouts[i], out->shadow_layout[i] = layouts[16 + i], out->shadow_cached[i] = layouts[32 + i];
        out->occluder_hint_levels = misc[0], out->shadow_first_look_from_bounce = misc[1], out->dense_leaf_min = misc[2], out->legacy_layouts_compiled = misc[3];
        out->quad_half_area_ratio = ratio, out->reserved = 0.0f, out->tree_bytes = tree;
        return RF_OK;
    });
}

int rf_renderer_memory_info(const rf_renderer* r, uint64_t* path_state_bytes, uint64_t* paths_allocated, uint64_t* max_paths_per_batch, uint64_t* scene_bytes)
{
    return guarded([&] {
        require(r, "null argument");
        uint64_t a = 0, b = 0, c = 0, d = 0;
        r->impl->memoryInfo(a, b, c, d);
        if (path_state_bytes) *path_state_bytes = a;
        if (paths_allocated) *paths_allocated = b;
        if (max_paths_per_batch) *max_paths_per_batch = c;
        if (scene_bytes) *scene_bytes = d;
        return RF_OK;
    });
}

int rf_gather_layout(uint32_t width, uint32_t height, uint32_t world_size, uint32_t* rank_first_tile, uint32_t* tile_slot, uint32_t* tile_owner)
{
    return guarded([&] {
        require(width > 0 && height > 0 && world_size > 0, "empty frame or world");
        const rf::GatherLayout g = rf::gatherLayout(width, height, world_size);
        if (rank_first_tile) std::memcpy(rank_first_tile, g.rankFirstTile.data(), g.rankFirstTile.size() * sizeof(uint32_t));
        if (tile_slot) std::memcpy(tile_slot, g.tileSlot.data(), g.tileSlot.size() * sizeof(uint32_t));
        if (tile_owner) std::memcpy(tile_owner, g.tileOwner.data(), g.tileOwner.size() * sizeof(uint32_t));
        return RF_OK;
    });
}

int rf_tiles_for_rank(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, uint32_t* tile_ids, uint32_t* num_tiles)
{
    return guarded([&] {
        require(num_tiles, "null argument");
        require(world_size > 0 && rank < world_size, "invalid rank / world size");
        const auto tiles = rf::tilesForRank(width, height, rank, world_size);
        *num_tiles = static_cast<uint32_t>(tiles.size());
        if (tile_ids) std::memcpy(tile_ids, tiles.data(), tiles.size() * sizeof(uint32_t));
        return RF_OK;
    });
}

int rf_untile(const float* compact, const uint32_t* tile_ids, uint32_t num_tiles, uint32_t width, uint32_t height, float* image)
{
    return guarded([&] {
        require((compact && tile_ids) || num_tiles == 0, "null argument");
        require(image, "null argument");
        rf::untileHost(compact, tile_ids, num_tiles, width, height, image);
        return RF_OK;
    });
}

// ---------------------------------------------------------------------------------------- queries
int rf_renderer_trace_primary_stats(rf_renderer* r, const rf_camera* camera, uint32_t width, uint32_t height, uint32_t* nodes_visited,
                                    uint8_t* hit, float* t, uint32_t* triangle_tests)
{
    return guarded([&] {
        require(r && camera && nodes_visited, "null argument");
        require(width > 0 && height > 0, "empty image");
        r->impl->tracePrimaryStats(toCamera(*camera), width, height, nodes_visited, hit, t, triangle_tests);
        return RF_OK;
    });
}

int rf_renderer_intersect_rays(rf_renderer* r, const float* rays6, uint64_t num_rays, float t_max, uint32_t* triangle, float* t, float* uv,
                               float* p, uint32_t* nodes_visited, uint32_t* triangle_tests)
{
    return guarded([&] {
        require(r && (num_rays == 0 || (rays6 && triangle)), "null argument");
        r->impl->intersectRays(rays6, num_rays, t_max, triangle, t, uv, p, nodes_visited, triangle_tests);
        return RF_OK;
    });
}

int rf_renderer_occluded_rays(rf_renderer* r, const float* rays6, uint64_t num_rays, float t_max, float* visibility)
{
    return guarded([&] {
        require(r && (num_rays == 0 || (rays6 && visibility)), "null argument");
        r->impl->occludedRays(rays6, num_rays, t_max, visibility);
        return RF_OK;
    });
}

// ---------------------------------------------------------------------------------------- CPU side
int rf_build_bvh(const float* positions36, uint64_t num_triangles, void* nodes_out, uint64_t* num_nodes_out,
                 uint64_t* triangle_indices_out, int32_t* depth_out)
{
    return guarded([&] {
        require(positions36 && nodes_out && num_nodes_out && triangle_indices_out, "null argument");
        require(num_triangles > 0, "buildBvh needs at least one triangle"); // bvh.cpp:265 asserts
        const rf::Bvh bvh = rf::buildBvh({reinterpret_cast<const rf::Positions*>(positions36), static_cast<size_t>(num_triangles)});
        std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(rf::BvhNode));
        *num_nodes_out = bvh.nodes.size();
        for (size_t i = 0; i < bvh.triangleIndices.size(); ++i) triangle_indices_out[i] = bvh.triangleIndices[i];
        if (depth_out) *depth_out = bvh.depth;
        return RF_OK;
    });
}

int rf_check_wide_layouts(const void* nodes48, uint64_t num_nodes, uint32_t* flags_out)
{
    return guarded([&] {
        require(nodes48 && flags_out, "null argument");
        std::span<const rf::BvhNode> nodes{static_cast<const rf::BvhNode*>(nodes48), static_cast<size_t>(num_nodes)};
        // (the same structural check rf_renderer_create runs first: the layout builder follows child links blindly)
        rf::validateScene(nodes, static_cast<size_t>(-1), {}, 0);
        *flags_out = rf::checkWideLayouts(nodes);
        return RF_OK;
    });
}

int rf_wide_layout_stats(const void* nodes48, uint64_t num_nodes, uint32_t* flags_out, float* quad_half_area_ratio)
{
    return guarded([&] {
        require(nodes48, "null argument");
        std::span<const rf::BvhNode> nodes{static_cast<const rf::BvhNode*>(nodes48), static_cast<size_t>(num_nodes)};
        rf::validateScene(nodes, static_cast<size_t>(-1), {}, 0);
        float          ratio = 0.0f;
        const uint32_t flags = rf::checkWideLayouts(nodes, &ratio);
        if (flags_out) *flags_out = flags;
        if (quad_half_area_ratio) *quad_half_area_ratio = ratio;
        return RF_OK;
    });
}

int rf_build_bvh_gpu(const float* positions36, uint64_t num_triangles, void* nodes_out, uint64_t* num_nodes_out, uint64_t* triangle_indices_out,
                     int32_t* depth_out, int32_t device_ordinal, float* build_ms_out)
{
    return guarded([&] {
        require(positions36 && nodes_out && num_nodes_out && triangle_indices_out, "null argument");
        require(num_triangles > 0, "buildBvh needs at least one triangle"); // bvh.cpp:265 asserts
        const rf::Bvh bvh = rf::buildBvhGpu({reinterpret_cast<const rf::Positions*>(positions36), static_cast<size_t>(num_triangles)}, device_ordinal,
                                            build_ms_out);
        std::memcpy(nodes_out, bvh.nodes.data(), bvh.nodes.size() * sizeof(rf::BvhNode));
        *num_nodes_out = bvh.nodes.size();
        for (size_t i = 0; i < bvh.triangleIndices.size(); ++i) triangle_indices_out[i] = bvh.triangleIndices[i];
        if (depth_out) *depth_out = bvh.depth;
        return RF_OK;
    });
}

int rf_create_camera(const float origin[3], const float look_at[3], float aperture, float focus_distance, float vfov_radians,
                     float aspect_ratio, rf_camera* out)
{
    return guarded([&] {
        require(origin && look_at && out, "null argument");
        const rf::Camera c = rf::createCamera(v3(origin), v3(look_at), aperture, focus_distance, vfov_radians, aspect_ratio);
        std::memcpy(out, &c, sizeof c);
        return RF_OK;
    });
}

int rf_fly_camera(const float position[3], float yaw_degrees, float pitch_degrees, float vfov_degrees, float aperture, float focus_distance,
                  float aspect_ratio, rf_camera* out)
{
    return guarded([&] {
        require(position && out, "null argument");
        const rf::Camera c = rf::flyCamera(v3(position), yaw_degrees, pitch_degrees, vfov_degrees, aperture, focus_distance, aspect_ratio);
        std::memcpy(out, &c, sizeof c);
        return RF_OK;
    });
}

int rf_bvh_visualizer_camera(const void* root_node48, float aspect_ratio, rf_camera* out)
{
    return guarded([&] {
        require(root_node48 && out, "null argument");
        const rf::Camera c = rf::bvhVisualizerCamera(static_cast<const rf::BvhNode*>(root_node48)->aabb, aspect_ratio);
        std::memcpy(out, &c, sizeof c);
        return RF_OK;
    });
}

int rf_sky_state_new(float elevation, float turbidity, const float albedo[3], float state33[33])
{
    if (!albedo || !state33) return -1;
    return static_cast<int>(rf::skyStateNew(elevation, turbidity, albedo, state33));
}

float rf_sky_state_radiance(const float state33[33], float theta, float gamma, int channel)
{
    return rf::skyStateRadiance(state33, theta, gamma, channel);
}

int rf_aligned_sky_state(const rf_sky* sky, float out40[40])
{
    return guarded([&] {
        require(sky && out40, "null argument");
        rf::Sky s;
        s.turbidity = sky->turbidity;
        std::memcpy(s.albedo, sky->albedo, sizeof s.albedo);
        s.sunZenithDegrees = sky->sun_zenith_degrees;
        s.sunAzimuthDegrees = sky->sun_azimuth_degrees;
        rf::SkyStateGpu g;
        const rf::SkyResult rc = rf::alignedSkyState(s, g);
        std::memcpy(out40, &g, sizeof g);
        if (rc != rf::SkyResult::Success)
        {
            gLastError = "sky parameters out of range";
            return static_cast<int>(RF_ERROR_OUT_OF_RANGE);
        }
        return static_cast<int>(RF_OK);
    });
}

// ---------------------------------------------------------------------------------------- .pt files
int rf_texture_from_memory(const void* data, uint64_t size, uint32_t* width, uint32_t* height, uint32_t* pixels)
{
    return guarded([&] {
        require(data && width && height, "null argument");
        const rf::Texture t = rf::textureFromMemory({static_cast<const uint8_t*>(data), static_cast<size_t>(size)});
        *width = t.width;
        *height = t.height;
        if (pixels) std::memcpy(pixels, t.pixels.data(), t.pixels.size() * sizeof(uint32_t));
        return RF_OK;
    });
}

int rf_pt_format_set_bvh_builder(int32_t gpu_device_or_minus_one)
{
    return guarded([&] {
        rf::setBakeBvhBuilder(gpu_device_or_minus_one);
        return RF_OK;
    });
}

int rf_pt_format_from_gltf(const char* gltf_path, rf_pt_format** out)
{
    return guarded([&] {
        require(gltf_path && out, "null argument");
        auto h = std::make_unique<rf_pt_format>();
        h->format = rf::ptFormatFromGltf(gltf_path);
        *out = h.release();
        return RF_OK;
    });
}

int rf_pt_format_load(const char* pt_path, rf_pt_format** out)
{
    return guarded([&] {
        require(pt_path && out, "null argument");
        auto h = std::make_unique<rf_pt_format>();
        h->format = rf::readPtFile(pt_path);
        *out = h.release();
        return RF_OK;
    });
}

int rf_pt_format_deserialize(const void* data, uint64_t size, rf_pt_format** out)
{
    return guarded([&] {
        require(data && out, "null argument");
        auto h = std::make_unique<rf_pt_format>();
        rf::deserializePt(static_cast<const uint8_t*>(data), static_cast<size_t>(size), h->format);
        *out = h.release();
        return RF_OK;
    });
}

int rf_pt_format_save(const rf_pt_format* f, const char* pt_path)
{
    return guarded([&] {
        require(f && pt_path, "null argument");
        rf::writePtFile(pt_path, f->format);
        return RF_OK;
    });
}

int rf_pt_format_serialize(const rf_pt_format* f, void* dst, uint64_t* size)
{
    return guarded([&] {
        require(f && size, "null argument");
        const std::vector<uint8_t> bytes = rf::serializePt(f->format);
        if (dst)
        {
            require(*size >= bytes.size(), "destination too small");
            std::memcpy(dst, bytes.data(), bytes.size());
        }
        *size = bytes.size();
        return RF_OK;
    });
}

int rf_pt_format_from_triangles(const float* positions36, const float* normals36, const float* tex_coords24, const uint32_t* texture_indices,
                                uint64_t num_triangles, const rf_texture* textures, uint64_t num_textures, rf_pt_format** out)
{
    return guarded([&] {
        require(positions36 && normals36 && tex_coords24 && texture_indices && out, "null argument");
        require(num_triangles > 0, "no triangles");
        const size_t n = static_cast<size_t>(num_triangles);
        std::vector<rf::Texture> tex;
        for (uint64_t i = 0; i < num_textures; ++i)
        {
            rf::Texture t;
            t.width = textures[i].width;
            t.height = textures[i].height;
            t.pixels.assign(textures[i].pixels, textures[i].pixels + static_cast<size_t>(t.width) * t.height);
            tex.push_back(std::move(t));
        }
        auto h = std::make_unique<rf_pt_format>();
        h->format = rf::ptFormatFromTriangles({reinterpret_cast<const rf::Positions*>(positions36), n},
                                              {reinterpret_cast<const rf::Normals*>(normals36), n},
                                              {reinterpret_cast<const rf::TexCoords*>(tex_coords24), n}, {texture_indices, n}, std::move(tex));
        *out = h.release();
        return RF_OK;
    });
}

int rf_pt_format_view_get(const rf_pt_format* f, rf_pt_format_view* v)
{
    return guarded([&] {
        require(f && v, "null argument");
        const rf::PtFormat& p = f->format;
        std::memset(v, 0, sizeof *v);
        v->bvh_nodes = p.bvhNodes.data();
        v->num_bvh_nodes = p.bvhNodes.size();
        v->bvh_position_attributes = p.bvhPositionAttributes.data();
        v->num_bvh_position_attributes = p.bvhPositionAttributes.size();
        v->triangle_position_attributes = p.trianglePositionAttributes.data();
        v->num_triangle_position_attributes = p.trianglePositionAttributes.size();
        v->triangle_vertex_attributes = p.triangleVertexAttributes.data();
        v->num_triangle_vertex_attributes = p.triangleVertexAttributes.size();
        v->vertex_positions = reinterpret_cast<const float*>(p.vertexPositions.data());
        v->num_vertex_positions = p.vertexPositions.size();
        v->vertex_normals = reinterpret_cast<const float*>(p.vertexNormals.data());
        v->num_vertex_normals = p.vertexNormals.size();
        v->vertex_tex_coords = reinterpret_cast<const float*>(p.vertexTexCoords.data());
        v->num_vertex_tex_coords = p.vertexTexCoords.size();
        v->vertex_indices = p.vertexIndices.data();
        v->num_vertex_indices = p.vertexIndices.size();
        static_assert(sizeof(rf::Slice) == 16);
        v->model_vertex_positions = reinterpret_cast<const uint64_t*>(p.modelVertexPositions.data());
        v->num_model_vertex_positions = p.modelVertexPositions.size();
        v->model_vertex_normals = reinterpret_cast<const uint64_t*>(p.modelVertexNormals.data());
        v->num_model_vertex_normals = p.modelVertexNormals.size();
        v->model_vertex_tex_coords = reinterpret_cast<const uint64_t*>(p.modelVertexTexCoords.data());
        v->num_model_vertex_tex_coords = p.modelVertexTexCoords.size();
        v->model_vertex_indices = reinterpret_cast<const uint64_t*>(p.modelVertexIndices.data());
        v->num_model_vertex_indices = p.modelVertexIndices.size();
        v->model_base_color_texture_indices = p.modelBaseColorTextureIndices.data();
        v->num_model_base_color_texture_indices = p.modelBaseColorTextureIndices.size();
        v->num_textures = p.baseColorTextures.size();
        return RF_OK;
    });
}

int rf_pt_format_texture(const rf_pt_format* f, uint64_t index, rf_texture* out)
{
    return guarded([&] {
        require(f && out, "null argument");
        require(index < f->format.baseColorTextures.size(), "texture index out of range");
        const rf::Texture& t = f->format.baseColorTextures[static_cast<size_t>(index)];
        out->pixels = t.pixels.data();
        out->width = t.width;
        out->height = t.height;
        return RF_OK;
    });
}

void rf_pt_format_destroy(rf_pt_format* f) { delete f; }

int rf_pt_format_scene(const rf_pt_format* f, rf_scene* scene, rf_texture* textures)
{
    return guarded([&] {
        require(f && scene, "null argument");
        const rf::PtFormat& p = f->format;
        require(textures || p.baseColorTextures.empty(), "null texture array");
        require(p.trianglePositionAttributes.size() == p.triangleVertexAttributes.size(), "attribute arrays differ in length");
        scene->bvh_nodes = p.bvhNodes.data();
        scene->num_bvh_nodes = p.bvhNodes.size();
        scene->position_attributes = p.trianglePositionAttributes.data();
        scene->vertex_attributes = p.triangleVertexAttributes.data();
        scene->num_triangles = p.trianglePositionAttributes.size();
        for (size_t i = 0; i < p.baseColorTextures.size(); ++i)
            require(p.baseColorTextures[i].pixels.size() == static_cast<size_t>(p.baseColorTextures[i].width) * p.baseColorTextures[i].height,
                    "texture pixel count differs from width * height");
        for (size_t i = 0; i < p.baseColorTextures.size(); ++i)
            textures[i] = rf_texture{p.baseColorTextures[i].pixels.data(), p.baseColorTextures[i].width, p.baseColorTextures[i].height};
        scene->base_color_textures = textures;
        scene->num_textures = p.baseColorTextures.size();
        return RF_OK;
    });
}
} // extern "C"
