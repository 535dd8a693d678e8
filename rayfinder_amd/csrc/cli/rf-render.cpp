// rf-render <scene.pt|scene.glb> [--width W] [--height H] [--spp N] [--bounces B] [--vfov deg]
//           [--zenith deg] [--azimuth deg] [--turbidity t] [--exposure-stops s] [--out image.png]
//           [--pfm image.pfm]
// Offline counterpart of the interactive `pt` app (src/pt/main.cpp): same default camera pose,
// sky and exposure; renders all samples and writes the tonemapped image (and optionally the
// mean radiance as PFM).
#include "cli_common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::printf("Usage: rf-render <scene.pt|scene.glb> [--width W] [--height H] [--spp N] [--bounces B] [--vfov deg]\n"
                    "                 [--zenith deg] [--azimuth deg] [--turbidity t] [--exposure-stops s] [--out image.png] [--pfm image.pfm]\n");
        return 0;
    }
    uint32_t    W = 1920, H = 1080, spp = 64, bounces = 2; // UI defaults src/pt/main.cpp:46-60
    float       vfov = 70.0f, zenith = 30.0f, azimuth = 0.0f, turbidity = 1.0f;
    int         stops = 2;
    std::string out = "render.png", pfm;
    for (int i = 2; i + 1 < argc; i += 2)
    {
        const std::string k = argv[i];
        const char*       val = argv[i + 1];
        if (k == "--width") W = static_cast<uint32_t>(std::atoi(val));
        else if (k == "--height") H = static_cast<uint32_t>(std::atoi(val));
        else if (k == "--spp") spp = static_cast<uint32_t>(std::atoi(val));
        else if (k == "--bounces") bounces = static_cast<uint32_t>(std::atoi(val));
        else if (k == "--vfov") vfov = static_cast<float>(std::atof(val));
        else if (k == "--zenith") zenith = static_cast<float>(std::atof(val));
        else if (k == "--azimuth") azimuth = static_cast<float>(std::atof(val));
        else if (k == "--turbidity") turbidity = static_cast<float>(std::atof(val));
        else if (k == "--exposure-stops") stops = std::atoi(val);
        else if (k == "--out") out = val;
        else if (k == "--pfm") pfm = val;
        else
        {
            std::fprintf(stderr, "unknown option %s\n", k.c_str());
            return 1;
        }
    }
    rf_pt_format*     pt = loadScene(argv[1]);
    rf_pt_format_view v;
    rf_pt_format_view_get(pt, &v);
    std::vector<rf_texture> textures(std::max<uint64_t>(v.num_textures, 1));
    rf_scene                scene;
    rfCheck(rf_pt_format_scene(pt, &scene, textures.data()), "scene");

    rf_renderer_descriptor desc{};
    desc.render_params.width = W;
    desc.render_params.height = H;
    const float position[3] = {1.22f, 1.25f, -1.25f};
    rfCheck(rf_fly_camera(position, 129.64f, -13.73f, vfov, 0.0f, 10.0f, static_cast<float>(W) / static_cast<float>(H), &desc.render_params.camera), "camera");
    desc.render_params.num_samples_per_pixel = spp;
    desc.render_params.num_bounces = bounces;
    desc.render_params.sky = rf_sky{turbidity, {1.0f, 1.0f, 1.0f}, zenith, azimuth};
    desc.render_params.exposure = 1.0f / std::exp2(static_cast<float>(stops));
    rf_renderer* renderer = nullptr;
    rfCheck(rf_renderer_create(&desc, &scene, &renderer), "create renderer");

    const auto t0 = std::chrono::steady_clock::now();
    rfCheck(rf_renderer_render(renderer, spp), "render");
    rfCheck(rf_renderer_synchronize(renderer), "synchronize");
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    rf_stats     stats;
    rfCheck(rf_renderer_get_stats(renderer, &stats), "stats");
    const double rays = static_cast<double>(stats.closest_rays + stats.shadow_rays);
    std::printf("%ux%u, %u spp, %u bounces: %.3f s, %.1f Mrays/s (%llu closest + %llu shadow rays), %.1f%% done\n", W, H, spp, bounces, seconds,
                rays / seconds * 1e-6, (unsigned long long)stats.closest_rays, (unsigned long long)stats.shadow_rays,
                rf_renderer_render_progress_percentage(renderer));

    std::vector<uint32_t> bgra(static_cast<size_t>(W) * H);
    rfCheck(rf_renderer_read_tonemapped(renderer, bgra.data()), "tonemap");
    std::vector<uint8_t> rgba(bgra.size() * 4);
    for (size_t i = 0; i < bgra.size(); ++i)
    {
        rgba[4 * i] = static_cast<uint8_t>(bgra[i] >> 16);
        rgba[4 * i + 1] = static_cast<uint8_t>(bgra[i] >> 8);
        rgba[4 * i + 2] = static_cast<uint8_t>(bgra[i]);
        rgba[4 * i + 3] = 255;
    }
    if (!writePngRgba(out, rgba.data(), W, H)) return 1;
    if (!pfm.empty())
    {
        std::vector<float> acc(static_cast<size_t>(W) * H * 4);
        uint32_t           n = 0;
        rfCheck(rf_renderer_read_accumulation(renderer, acc.data(), &n), "read accumulation");
        writePfm(pfm, acc.data(), W, H, 1.0f / static_cast<float>(std::max(n, 1u)));
    }
    rf_renderer_destroy(renderer);
    rf_pt_format_destroy(pt);
    return 0;
}
