#include "rf_pt_format.hpp"

#include "rf_bvh.hpp"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string_view>

namespace rf
{
namespace
{
constexpr std::string_view kMagic = "PTFORMAT3";

class Writer
{
public:
    void raw(const void* p, std::size_t n)
    {
        const auto* b = static_cast<const uint8_t*>(p);
        mBytes.insert(mBytes.end(), b, b + n);
    }
    void u64(uint64_t v) { raw(&v, sizeof v); }
    template<typename T>
    void array(const std::vector<T>& v)
    {
        u64(v.size());
        raw(v.data(), v.size() * sizeof(T));
    }
    void slices(const std::vector<Slice>& s)
    {
        u64(s.size());
        for (const Slice& x : s)
        {
            u64(x.offset);
            u64(x.count);
        }
    }
    std::vector<uint8_t> take() { return std::move(mBytes); }

private:
    std::vector<uint8_t> mBytes;
};

class Reader
{
public:
    Reader(const uint8_t* p, std::size_t n) : mPtr(p), mLeft(n) {}
    void raw(void* dst, std::size_t n)
    {
        if (n > mLeft) throw std::runtime_error("Unexpected end of PtFormat data.");
        if (n) std::memcpy(dst, mPtr, n); // (an empty vector's data() may be null)
        mPtr += n;
        mLeft -= n;
    }
    uint64_t u64()
    {
        uint64_t v;
        raw(&v, sizeof v);
        return v;
    }
    template<typename T>
    void array(std::vector<T>& v)
    {
        const uint64_t n = u64();
        if (n > mLeft / sizeof(T)) throw std::runtime_error("Unexpected end of PtFormat data.");
        v.resize(n);
        raw(v.data(), n * sizeof(T));
    }
    void slices(std::vector<Slice>& s, std::size_t bufferSize)
    {
        const uint64_t n = u64();
        if (n > mLeft / 16) throw std::runtime_error("Unexpected end of PtFormat data.");
        s.resize(n);
        for (Slice& x : s)
        {
            x.offset = u64();
            x.count = u64();
            if (x.count > bufferSize || x.offset > bufferSize - x.count) throw std::runtime_error("PtFormat slice exceeds its buffer.");
        }
    }

private:
    const uint8_t* mPtr;
    std::size_t    mLeft;
};
} // namespace

std::vector<uint8_t> serializePt(const PtFormat& f)
{
    Writer w;
    w.raw(kMagic.data(), kMagic.size());
    w.array(f.bvhNodes);
    w.array(f.bvhPositionAttributes);
    w.array(f.trianglePositionAttributes);
    w.array(f.triangleVertexAttributes);
    w.array(f.vertexPositions);
    w.array(f.vertexNormals);
    w.array(f.vertexTexCoords);
    w.array(f.vertexIndices);
    w.slices(f.modelVertexPositions);
    w.slices(f.modelVertexNormals);
    w.slices(f.modelVertexTexCoords);
    w.slices(f.modelVertexIndices);
    w.array(f.modelBaseColorTextureIndices);
    w.u64(f.baseColorTextures.size());
    for (const Texture& t : f.baseColorTextures)
    {
        const uint32_t dims[2] = {t.width, t.height};
        w.raw(dims, sizeof dims);
        w.array(t.pixels);
    }
    return w.take();
}

void deserializePt(const uint8_t* data, std::size_t size, PtFormat& f)
{
    Reader r(data, size);
    std::string magic(kMagic.size(), '\0');
    r.raw(magic.data(), magic.size());
    if (magic != kMagic)
    {
        // the reference searches for the regex "PTFORMAT\d" anywhere in the 9 bytes read; with
        // 9 bytes the only possible match is at offset 0
        const bool otherVersion = magic.compare(0, 8, "PTFORMAT") == 0 && std::isdigit(static_cast<unsigned char>(magic[8]));
        if (otherVersion)
        {
            throw std::runtime_error(
                "Mismatching PtFormat file version. Invalid version in magic bytes: expected '" + std::string(kMagic) +
                "', got '" + magic + "'.");
        }
        throw std::runtime_error("Invalid file format: expected PtFormat file.");
    }
    r.array(f.bvhNodes);
    r.array(f.bvhPositionAttributes);
    r.array(f.trianglePositionAttributes);
    r.array(f.triangleVertexAttributes);
    r.array(f.vertexPositions);
    r.array(f.vertexNormals);
    r.array(f.vertexTexCoords);
    r.array(f.vertexIndices);
    r.slices(f.modelVertexPositions, f.vertexPositions.size());
    r.slices(f.modelVertexNormals, f.vertexNormals.size());
    r.slices(f.modelVertexTexCoords, f.vertexTexCoords.size());
    r.slices(f.modelVertexIndices, f.vertexIndices.size());
    r.array(f.modelBaseColorTextureIndices);
    const uint64_t numTextures = r.u64();
    f.baseColorTextures.clear();
    for (uint64_t i = 0; i < numTextures; ++i)
    {
        Texture  t;
        uint32_t dims[2];
        r.raw(dims, sizeof dims);
        t.width = dims[0];
        t.height = dims[1];
        r.array(t.pixels);
        if (t.pixels.size() != static_cast<uint64_t>(t.width) * t.height)
            throw std::runtime_error("PtFormat texture " + std::to_string(i) + " declares " + std::to_string(t.width) + "x" + std::to_string(t.height) +
                                     " but holds " + std::to_string(t.pixels.size()) + " pixels.");
        f.baseColorTextures.push_back(std::move(t));
    }
    // a file's hot-path arrays are checked once here; the renderer checks what it is handed again (rf_bvh.hpp)
    if (f.trianglePositionAttributes.size() != f.triangleVertexAttributes.size())
        throw std::runtime_error("PtFormat position and vertex attribute arrays differ in length.");
    if (!f.bvhNodes.empty())
        validateScene(f.bvhNodes, f.trianglePositionAttributes.size(), f.triangleVertexAttributes, f.baseColorTextures.size());
}

void writePtFile(const std::string& path, const PtFormat& format)
{
    const std::vector<uint8_t> bytes = serializePt(format);
    FILE*                      fp = std::fopen(path.c_str(), "wb");
    if (!fp) throw std::runtime_error("Cannot open " + path + " for writing.");
    const std::size_t n = std::fwrite(bytes.data(), 1, bytes.size(), fp);
    std::fclose(fp);
    if (n != bytes.size()) throw std::runtime_error("Short write to " + path + ".");
}

PtFormat readPtFile(const std::string& path)
{
    FILE* fp = std::fopen(path.c_str(), "rb");
    if (!fp) throw std::runtime_error("File " + path + " does not exist");
    std::fseek(fp, 0, SEEK_END);
    const long sz = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    std::vector<uint8_t> bytes(static_cast<std::size_t>(sz));
    const std::size_t    n = std::fread(bytes.data(), 1, bytes.size(), fp);
    std::fclose(fp);
    if (n != bytes.size()) throw std::runtime_error("Short read from " + path + ".");
    PtFormat f;
    deserializePt(bytes.data(), bytes.size(), f);
    return f;
}
} // namespace rf
