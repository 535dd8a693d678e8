// rf-bvh-visualizer <input_gltf_or_pt_file> [width height]  ->  bvh-visualizer.png
// The reference tool (src/bvh-visualizer/main.cpp) traces 1280x720 primary rays on one CPU thread
// and writes the per-pixel node-visit count as a grey image; here the same pass runs on the GPU.
#include "cli_common.hpp"

#include <algorithm>

int main(int argc, char** argv)
{
    if (argc != 2 && argc != 4)
    {
        std::printf("Usage: rf-bvh-visualizer <input_gltf_or_pt_file> [width height]\n");
        return 0;
    }
    const uint32_t W = argc == 4 ? static_cast<uint32_t>(std::atoi(argv[2])) : 1280u;
    const uint32_t H = argc == 4 ? static_cast<uint32_t>(std::atoi(argv[3])) : 720u;
    rf_pt_format*  pt = loadScene(argv[1]);
    rf_pt_format_view v;
    rf_pt_format_view_get(pt, &v);
    std::vector<rf_texture> textures(std::max<uint64_t>(v.num_textures, 1));
    rf_scene                scene;
    rfCheck(rf_pt_format_scene(pt, &scene, textures.data()), "scene");

    rf_camera camera;
    rfCheck(rf_bvh_visualizer_camera(v.bvh_nodes, static_cast<float>(W) / static_cast<float>(H), &camera), "camera");

    rf_renderer_descriptor desc{};
    desc.render_params.width = W;
    desc.render_params.height = H;
    desc.render_params.camera = camera;
    desc.render_params.num_samples_per_pixel = 1;
    desc.render_params.num_bounces = 1;
    desc.render_params.sky = rf_sky{1.0f, {1.0f, 1.0f, 1.0f}, 30.0f, 0.0f};
    desc.render_params.exposure = 1.0f;
    desc.max_paths_in_flight = 1u << 20;
    rf_renderer* renderer = nullptr;
    rfCheck(rf_renderer_create(&desc, &scene, &renderer), "create renderer");

    std::vector<uint32_t> visited(static_cast<size_t>(W) * H);
    rfCheck(rf_renderer_trace_primary_stats(renderer, &camera, W, H, visited.data(), nullptr, nullptr, nullptr), "trace");

    std::vector<uint8_t> rgba(visited.size() * 4);
    unsigned long long   total = 0;
    for (size_t i = 0; i < visited.size(); ++i)
    {
        total += visited[i];
        const float   x = 0.01f * static_cast<float>(visited[i]);
        const uint8_t p = static_cast<uint8_t>(static_cast<uint32_t>(std::min(x, 1.0f) * 255.0f));
        rgba[4 * i] = rgba[4 * i + 1] = rgba[4 * i + 2] = p;
        rgba[4 * i + 3] = 255;
    }
    if (!writePngRgba("bvh-visualizer.png", rgba.data(), W, H)) return 1;
    std::printf("bvh-visualizer.png: %ux%u, %llu node visits (%.2f per ray)\n", W, H, total, static_cast<double>(total) / (static_cast<double>(W) * H));
    rf_renderer_destroy(renderer);
    rf_pt_format_destroy(pt);
    return 0;
}
