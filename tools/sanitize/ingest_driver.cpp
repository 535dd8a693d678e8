#include "rf_gltf.hpp"
#include "rf_pt_format.hpp"
#include "rf_bvh_gpu.hpp"
#include "rf_wide.hpp"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>
namespace rf { Bvh buildBvhGpu(std::span<const Positions>, int, float*) { throw std::runtime_error("no gpu"); } }
int main(int argc, char** argv)
{
    int ok = 0, err = 0;
    for (int i = 1; i < argc; ++i)
    {
        const std::string path = argv[i];
        try
        {
            if (path.size() > 3 && path.substr(path.size() - 3) == ".pt")
            {
                // what rf_renderer_create does on the host with a scene that passed the reader: the structural check,
                // the wide-record build (follows every child link) and the texture blob (width * height texels each)
                auto f = rf::readPtFile(path);
                if (f.bvhNodes.empty()) { ++err; continue; }
                rf::validateScene(f.bvhNodes, f.trianglePositionAttributes.size(), f.triangleVertexAttributes, f.baseColorTextures.size());
                const rf::WideBuild wb = rf::buildWide(f.bvhNodes.data(), f.bvhNodes.size());
                std::vector<uint32_t> blob;
                for (const rf::Texture& t : f.baseColorTextures) blob.insert(blob.end(), t.pixels.data(), t.pixels.data() + static_cast<size_t>(t.width) * t.height);
                ok += !wb.nodes.empty() + (blob.size() & 0);
            }
            else { auto f = rf::ptFormatFromGltf(path); ok += !f.bvhNodes.empty(); }
        }
        catch (const std::exception&) { ++err; }
    }
    std::printf("ok %d err %d\n", ok, err);
}
