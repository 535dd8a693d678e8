// rf_comm.hip -- RCCL gather of tile shards + device un-tile (see rf_comm.hpp).
#include "rf_comm.hpp"

#include "rf_renderer.hpp" // tilesForRank, kTileSize

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <stdexcept>
#include <string>

namespace rf
{
namespace
{
#define RF_HIP(expr)                                                                                          \
    do                                                                                                        \
    {                                                                                                         \
        const hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " in " #expr);      \
    } while (0)
#define RF_NCCL(expr)                                                                                         \
    do                                                                                                        \
    {                                                                                                         \
        const ncclResult_t _r = (expr);                                                                       \
        if (_r != ncclSuccess)                                                                                \
            throw std::runtime_error(std::string("RCCL error: ") + ncclGetErrorString(_r) + " in " #expr);    \
    } while (0)

constexpr uint32_t kTilePixels = kTileSize * kTileSize; // 1024

// One workgroup per tile of the frame: tile-major (8x8-pixel blocks, one wave each) -> row-major.  Reads are
// 1-KiB contiguous per wave, writes 128-byte row segments.  `own`: the root's shard, read in place.
__global__ __launch_bounds__(256) void kUntile(const float4* __restrict__ staging, const float4* __restrict__ own, uint32_t ownRank, uint32_t ownFirstTile,
                                               const uint32_t* __restrict__ tileSlot, const uint32_t* __restrict__ tileOwner, uint32_t width, uint32_t height,
                                               uint32_t tilesX, float4* __restrict__ image)
{
    const uint32_t tile = blockIdx.x;
    const uint32_t slot = tileSlot[tile];
    const float4*  src = tileOwner[tile] == ownRank ? own + static_cast<size_t>(slot - ownFirstTile) * kTilePixels : staging + static_cast<size_t>(slot) * kTilePixels;
    const uint32_t x0 = (tile % tilesX) * kTileSize, y0 = (tile / tilesX) * kTileSize;
#pragma unroll
    for (uint32_t k = 0; k < kTilePixels / 256; ++k)
    {
        const uint32_t w = k * 256 + threadIdx.x;
        const uint32_t block = w >> 6, lane = w & 63u;
        const uint32_t x = x0 + (block & 3u) * 8u + (lane & 7u), y = y0 + (block >> 2) * 8u + (lane >> 3);
        if (x < width && y < height) image[static_cast<size_t>(y) * width + x] = src[w];
    }
}

template<typename T>
struct DevBuf
{
    T*     p = nullptr;
    size_t n = 0;
    void   ensure(size_t count)
    {
        if (count <= n) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        RF_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        n = count;
    }
    ~DevBuf()
    {
        if (p) (void)hipFree(p);
    }
};
} // namespace

GatherLayout gatherLayout(uint32_t width, uint32_t height, uint32_t worldSize)
{
    GatherLayout g;
    g.tilesX = (width + kTileSize - 1) / kTileSize;
    g.tilesY = (height + kTileSize - 1) / kTileSize;
    const uint32_t n = g.tilesX * g.tilesY;
    g.tileSlot.assign(n, 0);
    g.tileOwner.assign(n, 0);
    g.rankFirstTile.assign(worldSize + 1, 0);
    uint32_t at = 0;
    for (uint32_t r = 0; r < worldSize; ++r)
    {
        g.rankFirstTile[r] = at;
        const std::vector<uint32_t> tiles = tilesForRank(width, height, r, worldSize);
        for (uint32_t i = 0; i < tiles.size(); ++i)
        {
            g.tileSlot[tiles[i]] = at + i;
            g.tileOwner[tiles[i]] = r;
        }
        at += static_cast<uint32_t>(tiles.size());
    }
    g.rankFirstTile[worldSize] = at;
    return g;
}

struct TileComm::Impl
{
    ncclComm_t comm = nullptr;
    uint32_t   rank = 0, world = 1;
    int        device = 0;

    uint32_t         layoutW = 0, layoutH = 0;
    GatherLayout     layout;
    DevBuf<uint32_t> dTileSlot, dTileOwner;
    DevBuf<float4>   staging, image;
    DevBuf<double>   scalar;
    uint32_t         imageW = 0, imageH = 0;

    void ensureLayout(uint32_t w, uint32_t h)
    {
        if (w == layoutW && h == layoutH) return;
        layout = gatherLayout(w, h, world);
        dTileSlot.ensure(layout.tileSlot.size());
        dTileOwner.ensure(layout.tileOwner.size());
        RF_HIP(hipMemcpy(dTileSlot.p, layout.tileSlot.data(), layout.tileSlot.size() * 4, hipMemcpyHostToDevice));
        RF_HIP(hipMemcpy(dTileOwner.p, layout.tileOwner.data(), layout.tileOwner.size() * 4, hipMemcpyHostToDevice));
        layoutW = w;
        layoutH = h;
    }
};

void TileComm::uniqueId(uint8_t out[kCommIdBytes])
{
    static_assert(sizeof(ncclUniqueId) == kCommIdBytes);
    ncclUniqueId id;
    RF_NCCL(ncclGetUniqueId(&id));
    std::memcpy(out, &id, kCommIdBytes);
}

TileComm::TileComm(const uint8_t idBytes[kCommIdBytes], uint32_t rank, uint32_t worldSize, int deviceOrdinal) : mImpl(std::make_unique<Impl>())
{
    if (worldSize == 0 || rank >= worldSize) throw std::invalid_argument("invalid rank / world size");
    int deviceCount = 0;
    if (hipGetDeviceCount(&deviceCount) != hipSuccess || deviceCount == 0)
        throw std::runtime_error("rayfinder_amd: no HIP device available (the RCCL frame exchange needs one GPU per rank)");
    mImpl->rank = rank;
    mImpl->world = worldSize;
    mImpl->device = deviceOrdinal;
    RF_HIP(hipSetDevice(deviceOrdinal));
    ncclUniqueId id;
    std::memcpy(&id, idBytes, kCommIdBytes);
    RF_NCCL(ncclCommInitRank(&mImpl->comm, static_cast<int>(worldSize), id, static_cast<int>(rank)));
}

TileComm::~TileComm()
{
    if (mImpl && mImpl->comm)
    {
        (void)hipSetDevice(mImpl->device);
        (void)ncclCommDestroy(mImpl->comm);
    }
}

uint32_t TileComm::rank() const { return mImpl->rank; }
uint32_t TileComm::worldSize() const { return mImpl->world; }

const void* TileComm::gatherFrame(const void* compactDevice, uint32_t width, uint32_t height, uint32_t root, void* streamHandle, bool loopback)
{
    Impl& m = *mImpl;
    if (root >= m.world) throw std::invalid_argument("gather root out of range");
    if (width == 0 || height == 0) throw std::invalid_argument("empty frame");
    hipStream_t stream = static_cast<hipStream_t>(streamHandle);
    RF_HIP(hipSetDevice(m.device));
    m.ensureLayout(width, height);
    const GatherLayout& g = m.layout;
    const auto          tilesOf = [&](uint32_t r) { return g.rankFirstTile[r + 1] - g.rankFirstTile[r]; };
    const size_t        floatsPerTile = static_cast<size_t>(kTilePixels) * 4;
    const bool          isRoot = m.rank == root;
    if (tilesOf(m.rank) > 0 && compactDevice == nullptr) throw std::invalid_argument("null tile buffer");
    if (isRoot)
    {
        m.staging.ensure(static_cast<size_t>(g.rankFirstTile[m.world]) * kTilePixels);
        m.image.ensure(static_cast<size_t>(width) * height);
        m.imageW = width;
        m.imageH = height;
    }

    // one group: the root posts every receive at once, so all of its xGMI links carry data concurrently
    RF_NCCL(ncclGroupStart());
    try
    {
        if (isRoot)
        {
            for (uint32_t p = 0; p < m.world; ++p)
            {
                if (tilesOf(p) == 0 || (p == root && !loopback)) continue;
                RF_NCCL(ncclRecv(m.staging.p + static_cast<size_t>(g.rankFirstTile[p]) * kTilePixels, tilesOf(p) * floatsPerTile, ncclFloat, static_cast<int>(p), m.comm, stream));
            }
        }
        if (tilesOf(m.rank) > 0 && (!isRoot || loopback))
            RF_NCCL(ncclSend(compactDevice, tilesOf(m.rank) * floatsPerTile, ncclFloat, static_cast<int>(root), m.comm, stream));
    }
    catch (...)
    {
        (void)ncclGroupEnd(); // never leave the thread inside an open group
        throw;
    }
    RF_NCCL(ncclGroupEnd());
    if (!isRoot) return nullptr;

    const uint32_t numTiles = g.tilesX * g.tilesY;
    hipLaunchKernelGGL(kUntile, dim3(numTiles), dim3(256), 0, stream, m.staging.p, static_cast<const float4*>(compactDevice), loopback ? 0xFFFFFFFFu : m.rank,
                       g.rankFirstTile[m.rank], m.dTileSlot.p, m.dTileOwner.p, width, height, g.tilesX, m.image.p);
    RF_HIP(hipGetLastError());
    return m.image.p;
}

void TileComm::readFrame(float* dstHost, void* streamHandle)
{
    Impl& m = *mImpl;
    if (m.image.p == nullptr || m.imageW == 0) throw std::runtime_error("no gathered frame on this rank (only the gather root has one)");
    hipStream_t stream = static_cast<hipStream_t>(streamHandle);
    RF_HIP(hipSetDevice(m.device));
    RF_HIP(hipMemcpyAsync(dstHost, m.image.p, static_cast<size_t>(m.imageW) * m.imageH * sizeof(float4), hipMemcpyDeviceToHost, stream));
    RF_HIP(hipStreamSynchronize(stream));
}

double TileComm::allReduceMax(double value, void* streamHandle)
{
    Impl&       m = *mImpl;
    hipStream_t stream = static_cast<hipStream_t>(streamHandle);
    RF_HIP(hipSetDevice(m.device));
    m.scalar.ensure(2);
    RF_HIP(hipMemcpyAsync(m.scalar.p, &value, sizeof value, hipMemcpyHostToDevice, stream));
    RF_NCCL(ncclAllReduce(m.scalar.p, m.scalar.p + 1, 1, ncclDouble, ncclMax, m.comm, stream));
    double out = 0.0;
    RF_HIP(hipMemcpyAsync(&out, m.scalar.p + 1, sizeof out, hipMemcpyDeviceToHost, stream));
    RF_HIP(hipStreamSynchronize(stream));
    return out;
}
} // namespace rf
