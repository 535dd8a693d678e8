// rf-render <scene.pt|scene.glb> [--width W] [--height H] [--spp N] [--bounces B] [--vfov deg]
//           [--zenith deg] [--azimuth deg] [--turbidity t] [--exposure-stops s] [--out image.png]
//           [--pfm image.pfm] [--gpus N]
// Offline counterpart of the interactive `pt` app (src/pt/main.cpp): same default camera pose,
// sky and exposure; renders all samples and writes the tonemapped image (and optionally the
// mean radiance as PFM).  --gpus N: one host thread per GPU, the image tile-sharded across them, one RCCL
// gather to GPU 0 at frame end (rf_renderer_gather_frame).
#include "cli_common.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

int main(int argc, char** argv)
{
    if (argc < 2)
    {
        std::printf("Usage: rf-render <scene.pt|scene.glb> [--width W] [--height H] [--spp N] [--bounces B] [--vfov deg]\n"
                    "                 [--zenith deg] [--azimuth deg] [--turbidity t] [--exposure-stops s] [--out image.png] [--pfm image.pfm] [--gpus N]\n");
        return 0;
    }
    uint32_t    W = 1920, H = 1080, spp = 64, bounces = 2; // UI defaults src/pt/main.cpp:46-60
    uint32_t    gpus = 1;
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
        else if (k == "--gpus") gpus = static_cast<uint32_t>(std::max(1, std::atoi(val)));
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
    // One host thread per GPU (gpus > 1: the image is tile-sharded; one RCCL gather to rank 0 at frame end).
    uint8_t commId[RF_COMM_ID_BYTES] = {};
    if (gpus > 1) rfCheck(rf_comm_unique_id(commId), "RCCL unique id");
    std::atomic<unsigned long long> closestRays{0}, shadowRays{0};
    std::vector<uint32_t>           bgra(static_cast<size_t>(W) * H);
    std::vector<float>              acc;
    if (!pfm.empty()) acc.resize(static_cast<size_t>(W) * H * 4);
    double       seconds = 0.0;
    // rank r runs on device r -- modulo the devices there are: with more ranks than GPUs RCCL refuses the communicator (two ranks on one device), the local TEST transport
    // (RF_COMM_TRANSPORT=local in the environment: rf_comm.hip) runs them all on what is there -- how the exchange is exercised with many owners on a single-GPU box
    int32_t deviceCount = 0;
    rfCheck(rf_device_count(&deviceCount), "device count");
    const auto   worker = [&](uint32_t rank) {
        rf_renderer_descriptor d = desc;
        d.device_ordinal = static_cast<int32_t>(deviceCount > 0 ? rank % static_cast<uint32_t>(deviceCount) : rank);
        rf_renderer* renderer = nullptr;
        rfCheck(rf_renderer_create(&d, &scene, &renderer), "create renderer");
        rf_comm* comm = nullptr;
        if (gpus > 1)
        {
            rfCheck(rf_renderer_set_tile_shard(renderer, rank, gpus), "tile shard");
            rfCheck(rf_comm_create(commId, rank, gpus, d.device_ordinal, &comm), "RCCL communicator");
        }
        const auto t0 = std::chrono::steady_clock::now();
        rfCheck(rf_renderer_render(renderer, spp), "render");
        void* gathered = nullptr;
        if (comm) rfCheck(rf_renderer_gather_frame(renderer, comm, 0, 0, &gathered), "gather");
        rfCheck(rf_renderer_synchronize(renderer), "synchronize");
        if (rank == 0) seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rf_stats stats;
        rfCheck(rf_renderer_get_stats(renderer, &stats), "stats");
        closestRays += stats.closest_rays;
        shadowRays += stats.shadow_rays;
        if (rank == 0)
        {
            if (comm)
            {
                rfCheck(rf_renderer_tonemap_device_image(renderer, gathered, static_cast<uint64_t>(W) * H, spp, bgra.data()), "tonemap");
                if (!acc.empty()) rfCheck(rf_comm_read_frame(comm, renderer, acc.data()), "read frame");
            }
            else
            {
                rfCheck(rf_renderer_read_tonemapped(renderer, bgra.data()), "tonemap");
                uint32_t n = 0;
                if (!acc.empty()) rfCheck(rf_renderer_read_accumulation(renderer, acc.data(), &n), "read accumulation");
            }
        }
        if (comm) rf_comm_destroy(comm);
        rf_renderer_destroy(renderer);
    };
    std::vector<std::thread> threads;
    for (uint32_t rank = 1; rank < gpus; ++rank) threads.emplace_back(worker, rank);
    worker(0);
    for (std::thread& t : threads) t.join();

    const double rays = static_cast<double>(closestRays.load() + shadowRays.load());
    std::printf("%ux%u, %u spp, %u bounces on %u GPU(s): %.3f s, %.1f Mrays/s (%llu closest + %llu shadow rays)\n", W, H, spp, bounces, gpus, seconds,
                rays / seconds * 1e-6, closestRays.load(), shadowRays.load());

    std::vector<uint8_t> rgba(bgra.size() * 4);
    for (size_t i = 0; i < bgra.size(); ++i)
    {
        rgba[4 * i] = static_cast<uint8_t>(bgra[i] >> 16);
        rgba[4 * i + 1] = static_cast<uint8_t>(bgra[i] >> 8);
        rgba[4 * i + 2] = static_cast<uint8_t>(bgra[i]);
        rgba[4 * i + 3] = 255;
    }
    if (!writePngRgba(out, rgba.data(), W, H)) return 1;
    if (!pfm.empty()) writePfm(pfm, acc.data(), W, H, 1.0f / static_cast<float>(std::max(spp, 1u)));
    rf_pt_format_destroy(pt);
    return 0;
}
