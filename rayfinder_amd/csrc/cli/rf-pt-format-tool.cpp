// rf-pt-format-tool <input_gltf_file>  ->  writes <input>.pt next to it.
// Command-line behaviour of the reference's src/pt-format-tool/main.cpp:16-44.
#include "cli_common.hpp"

#include <cstdlib>
#include <filesystem>
#include <string>

int main(int argc, char** argv)
{
    // extension over the reference's tool: --gpu-bvh[=device] builds the BVH with rf_build_bvh_gpu
    int gpuDevice = -1, argi = 1;
    if (argc >= 2 && std::string(argv[1]).rfind("--gpu-bvh", 0) == 0)
    {
        const std::string a = argv[1];
        gpuDevice = a.size() > 10 ? std::atoi(a.c_str() + 10) : 0;
        argi = 2;
    }
    if (argc != argi + 1)
    {
        std::printf("Usage:\n\trf-pt-format-tool [--gpu-bvh[=device]] <input_gltf_file>\n");
        return 0;
    }
    rfCheck(rf_pt_format_set_bvh_builder(gpuDevice), "select BVH builder");
    std::filesystem::path path = argv[argi];
    if (!std::filesystem::exists(path))
    {
        std::fprintf(stderr, "File %s does not exist\n", path.string().c_str());
        return 1;
    }
    rf_pt_format* pt = nullptr;
    rfCheck(rf_pt_format_from_gltf(path.string().c_str(), &pt), "bake glTF");
    path.replace_extension(".pt");
    rfCheck(rf_pt_format_save(pt, path.string().c_str()), "write .pt");
    rf_pt_format_view v;
    rf_pt_format_view_get(pt, &v);
    std::printf("%s: %llu nodes, %llu triangles, %llu textures\n", path.string().c_str(), (unsigned long long)v.num_bvh_nodes,
                (unsigned long long)v.num_triangle_position_attributes, (unsigned long long)v.num_textures);
    rf_pt_format_destroy(pt);
    return 0;
}
