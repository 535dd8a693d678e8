// rf-bvh-visualizer [--cpu] [--threads N] [--out file.png] <input_gltf_or_pt_file> [width height]  ->  bvh-visualizer.png
// The reference tool (src/bvh-visualizer/main.cpp) traces 1280x720 primary rays on one CPU thread and writes the per-pixel
// node-visit count as a grey image (main.cpp:73-84: grey = u32(min(0.01 * nodesVisited, 1) * 255), alpha 255).
//   default   the same pass on the GPU (rf_renderer_trace_primary_stats); fails loudly without a device
//   --cpu     the same pass on the host (rf_bvh_visualizer_pass over the .pt file's 36-byte Positions, as the reference does):
//             no GPU needed -- BASELINE.json config 1.  --threads 1 is the reference's single thread; default: all cores.
// Both write identical files.
#include "cli_common.hpp"

#include <algorithm>
#include <chrono>

int main(int argc, char** argv)
{
    bool                     cpu = false;
    uint32_t                 threads = 0;
    std::string              outPath = "bvh-visualizer.png";
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i)
    {
        const std::string a = argv[i];
        if (a == "--cpu") cpu = true;
        else if (a == "--threads" && i + 1 < argc) threads = static_cast<uint32_t>(std::atoi(argv[++i]));
        else if (a == "--out" && i + 1 < argc) outPath = argv[++i];
        else pos.push_back(a);
    }
    if (pos.size() != 1 && pos.size() != 3)
    {
        std::printf("Usage: rf-bvh-visualizer [--cpu] [--threads N] [--out file.png] <input_gltf_or_pt_file> [width height]\n");
        return 0;
    }
    const uint32_t W = pos.size() == 3 ? static_cast<uint32_t>(std::atoi(pos[1].c_str())) : 1280u;
    const uint32_t H = pos.size() == 3 ? static_cast<uint32_t>(std::atoi(pos[2].c_str())) : 720u;
    if (W == 0 || H == 0)
    {
        std::fprintf(stderr, "invalid image size\n");
        return 1;
    }
    rf_pt_format*     pt = loadScene(pos[0].c_str());
    rf_pt_format_view v;
    rf_pt_format_view_get(pt, &v);

    rf_camera camera;
    rfCheck(rf_bvh_visualizer_camera(v.bvh_nodes, static_cast<float>(W) / static_cast<float>(H), &camera), "camera");

    std::vector<uint32_t> visited(static_cast<size_t>(W) * H);
    const auto            t0 = std::chrono::steady_clock::now();
    if (cpu)
    {
        rfCheck(rf_bvh_visualizer_pass(&camera, W, H, 0, H, v.bvh_nodes, v.num_bvh_nodes, v.bvh_position_attributes, 36, v.num_bvh_position_attributes, threads,
                                       visited.data(), nullptr, nullptr, nullptr),
                "trace (host)");
    }
    else
    {
        std::vector<rf_texture> textures(std::max<uint64_t>(v.num_textures, 1));
        rf_scene                scene;
        rfCheck(rf_pt_format_scene(pt, &scene, textures.data()), "scene");
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
        rfCheck(rf_renderer_trace_primary_stats(renderer, &camera, W, H, visited.data(), nullptr, nullptr, nullptr), "trace");
        rf_renderer_destroy(renderer);
    }
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

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
    if (!writePngRgba(outPath, rgba.data(), W, H)) return 1;
    std::printf("%s: %ux%u, %llu node visits (%.2f per ray), %s pass %.3f s (%.2f Mrays/s)\n", outPath.c_str(), W, H, total,
                static_cast<double>(total) / (static_cast<double>(W) * H), cpu ? "host" : "GPU", seconds, static_cast<double>(W) * H / seconds * 1e-6);
    rf_pt_format_destroy(pt);
    return 0;
}
