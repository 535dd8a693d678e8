#include "rf_gltf.hpp"
#include "rf_pt_format.hpp"
#include "rf_bvh_gpu.hpp"
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
            if (path.size() > 3 && path.substr(path.size() - 3) == ".pt") { auto f = rf::readPtFile(path); ok += !f.bvhNodes.empty(); }
            else { auto f = rf::ptFormatFromGltf(path); ok += !f.bvhNodes.empty(); }
        }
        catch (const std::exception&) { ++err; }
    }
    std::printf("ok %d err %d\n", ok, err);
}
