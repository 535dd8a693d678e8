#include "rf_jpeg.hpp"
#include <cstdio>
#include <fstream>
#include <iterator>
#include <vector>
int main(int argc, char** argv)
{
    int ok = 0, err = 0;
    for (int i = 1; i < argc; ++i)
    {
        std::ifstream f(argv[i], std::ios::binary);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        try { auto img = rf::decodeJpeg(d); ok += img.width > 0; } catch (const std::exception&) { ++err; }
    }
    std::printf("ok %d err %d\n", ok, err);
}
