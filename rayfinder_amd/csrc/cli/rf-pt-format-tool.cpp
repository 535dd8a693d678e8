// rf-pt-format-tool <input_gltf_file>  ->  writes <input>.pt next to it.
// Command-line behaviour of the reference's src/pt-format-tool/main.cpp:16-44.
#include "cli_common.hpp"

#include <filesystem>

int main(int argc, char** argv)
{
    if (argc != 2)
    {
        std::printf("Usage:\n\trf-pt-format-tool <input_gltf_file>\n");
        return 0;
    }
    std::filesystem::path path = argv[1];
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
