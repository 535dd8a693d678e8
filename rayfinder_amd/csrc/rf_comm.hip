// rf_comm.hip -- RCCL gather of tile shards + device un-tile (see rf_comm.hpp).
#include "rf_comm.hpp"

#include "rf_renderer.hpp" // tilesForRank, kTileSize

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <future>
#include <map>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>

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

std::vector<GatherOp> gatherPlan(const GatherLayout& g, uint32_t worldSize, uint32_t rank, uint32_t root, bool loopback)
{
    std::vector<GatherOp> ops;
    const auto            tilesOf = [&](uint32_t r) { return g.rankFirstTile[r + 1] - g.rankFirstTile[r]; };
    // the root posts every receive at once (one group), so all of its xGMI ingress links carry data concurrently
    if (rank == root)
        for (uint32_t p = 0; p < worldSize; ++p)
        {
            if (tilesOf(p) == 0 || (p == root && !loopback)) continue;
            ops.push_back(GatherOp{0u, p, g.rankFirstTile[p], tilesOf(p)});
        }
    if (tilesOf(rank) > 0 && (rank != root || loopback)) ops.push_back(GatherOp{1u, root, 0u, tilesOf(rank)});
    return ops;
}

namespace
{
// ------------------------------------------------------------------------------------------------
// The LOCAL transport (round 6; a TEST transport, RF_COMM_TRANSPORT=local): the same exchange -- the same gatherPlan(), the same staging offsets, the same
// kUntile -- with every ncclSend / ncclRecv pair replaced by a device-to-device copy between the buffers of N communicators that live in ONE process, one host
// thread per rank.  RCCL refuses two ranks on one device, and the boxes this is developed on have one GPU: without this, TileComm::gatherFrame() and kUntile
// had only ever executed with one owner (world 1: every tile `own`, or every tile staging in loop-back).  With it, worlds 2 / 3 / 4 / 8 run on one GPU: the
// root's receives land at rankFirstTile[peer] of its staging area, kUntile picks own / staging per tile from tileOwner / tileSlot, and the image must equal
// the single-rank one bit for bit (tests/test_gpu_parity.py: test_local_transport_gather_*).  What it does NOT exercise: RCCL itself and xGMI.
//
// Semantics mirror a group of point-to-point operations: a send is matched with the receive the peer posts for it, in posting order per (source,
// destination); the receiver's stream waits for the sender's stream (an event recorded behind the sender's frame kernels), copies, and the sender's stream
// waits for the copy before anything queued behind the exchange may touch the buffer again.  Host side a rank posts all its sends, then performs its
// receives (each waits for its send to be posted), then waits until its sends have been taken -- dead-lock free for the gather (and for loop-back).  A
// peer that never arrives is an error after RF_COMM_TIMEOUT_S, as with RCCL.
// ------------------------------------------------------------------------------------------------
constexpr char kLocalMagic[8] = {'R', 'F', 'L', 'O', 'C', 'A', 'L', '1'};

struct LocalPost
{
    const void* src = nullptr;
    size_t      bytes = 0;
    int         srcDevice = 0;
    hipEvent_t  ready = nullptr;    // recorded on the sender's stream behind everything queued before the exchange
    hipEvent_t  consumed = nullptr; // recorded on the receiver's stream behind its copy (set when `taken`)
    bool        taken = false;
};

struct LocalFabric
{
    uint32_t                world = 0;
    std::mutex              mutex;
    std::condition_variable cv;
    std::map<std::pair<uint32_t, uint32_t>, std::deque<std::shared_ptr<LocalPost>>> mail; // (source, destination) -> sends not yet received, oldest first
    std::vector<hipEvent_t> freeEvents, allEvents;
    // allReduceMax: a generation barrier
    uint64_t generation = 0;
    uint32_t arrived = 0;
    double   running = 0.0, result = 0.0;
    uint32_t attached = 0;

    hipEvent_t takeEvent() // (mutex held)
    {
        if (!freeEvents.empty())
        {
            hipEvent_t e = freeEvents.back();
            freeEvents.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        RF_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        allEvents.push_back(e);
        return e;
    }
    ~LocalFabric()
    {
        for (hipEvent_t e : allEvents) (void)hipEventDestroy(e);
    }
};

std::mutex                                                                    gFabricMutex;
std::map<std::array<uint8_t, kCommIdBytes>, std::weak_ptr<LocalFabric>>        gFabrics;

bool localTransportRequested()
{
    const char* v = std::getenv("RF_COMM_TRANSPORT");
    return v != nullptr && std::strcmp(v, "local") == 0;
}
bool isLocalId(const uint8_t id[kCommIdBytes]) { return std::memcmp(id, kLocalMagic, sizeof kLocalMagic) == 0; }

std::shared_ptr<LocalFabric> attachFabric(const uint8_t id[kCommIdBytes], uint32_t world)
{
    std::array<uint8_t, kCommIdBytes> key;
    std::memcpy(key.data(), id, kCommIdBytes);
    std::lock_guard<std::mutex> lock(gFabricMutex);
    std::shared_ptr<LocalFabric> f = gFabrics[key].lock();
    if (!f)
    {
        f = std::make_shared<LocalFabric>();
        f->world = world;
        gFabrics[key] = f;
    }
    if (f->world != world) throw std::runtime_error("local transport: the ranks of one id disagree about the world size");
    for (auto it = gFabrics.begin(); it != gFabrics.end();) it = it->second.expired() ? gFabrics.erase(it) : std::next(it);
    return f;
}
} // namespace

namespace
{
// seconds a rank waits for its peers in ncclCommInitRank and in its FIRST gather before it gives up with an error
// instead of hanging (a missing rank, a wrong id, a dead xGMI link); RF_COMM_TIMEOUT_S overrides, 0 = wait forever
double commTimeoutSeconds()
{
    if (const char* v = std::getenv("RF_COMM_TIMEOUT_S")) return std::atof(v);
    return 180.0;
}
} // namespace

struct TileComm::Impl
{
    ncclComm_t comm = nullptr;
    std::shared_ptr<LocalFabric> fabric; // the local (test) transport instead of RCCL: see LocalFabric
    uint32_t   rank = 0, world = 1;
    int        device = 0;

    uint32_t         layoutW = 0, layoutH = 0;
    GatherLayout     layout;
    DevBuf<uint32_t> dTileSlot, dTileOwner;
    DevBuf<float4>   staging, image;
    DevBuf<double>   scalar;
    uint32_t         imageW = 0, imageH = 0;
    bool             firstGatherDone = false;
    // HIP events around the last exchange on the caller's stream (sends / receives + the root's un-tile): what the frame-end gather costs THIS rank
    // once its own frame kernels have drained (lastExchangeMs())
    hipEvent_t       exchangeStart = nullptr, exchangeStop = nullptr;
    bool             exchangeTimed = false;

    // A new frame size: the previous gather's kUntile (on the caller's non-blocking stream) may still be reading the tables and
    // the staging / image buffers that are about to be replaced -- wait for it, then upload on that same stream.
    void ensureLayout(uint32_t w, uint32_t h, hipStream_t stream)
    {
        if (w == layoutW && h == layoutH) return;
        RF_HIP(hipStreamSynchronize(stream));
        layout = gatherLayout(w, h, world);
        dTileSlot.ensure(layout.tileSlot.size());
        dTileOwner.ensure(layout.tileOwner.size());
        RF_HIP(hipMemcpyAsync(dTileSlot.p, layout.tileSlot.data(), layout.tileSlot.size() * 4, hipMemcpyHostToDevice, stream));
        RF_HIP(hipMemcpyAsync(dTileOwner.p, layout.tileOwner.data(), layout.tileOwner.size() * 4, hipMemcpyHostToDevice, stream));
        RF_HIP(hipStreamSynchronize(stream)); // (`layout` outlives the copy anyway; this keeps pageable-copy semantics out of the picture)
        layoutW = w;
        layoutH = h;
    }
};

int deviceCount()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

void TileComm::uniqueId(uint8_t out[kCommIdBytes])
{
    static_assert(sizeof(ncclUniqueId) == kCommIdBytes);
    if (localTransportRequested())
    {
        // the local transport's ids carry a magic prefix (TileComm's constructor recognises them whatever the environment says by then) + 120 random bytes
        std::memcpy(out, kLocalMagic, sizeof kLocalMagic);
        std::random_device rd;
        for (uint32_t i = sizeof kLocalMagic; i < kCommIdBytes; ++i) out[i] = static_cast<uint8_t>(rd());
        return;
    }
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
    if (isLocalId(idBytes))
    {
        mImpl->fabric = attachFabric(idBytes, worldSize); // (not collective: a rank's peers are waited for in the exchange itself)
        return;
    }
    ncclUniqueId id;
    std::memcpy(&id, idBytes, kCommIdBytes);
    // ncclCommInitRank blocks until every rank of the world has called it.  It runs on a helper thread so that a rank whose
    // peers never arrive reports that after RF_COMM_TIMEOUT_S instead of hanging the job (the helper is abandoned then: the
    // caller is expected to exit).
    const double timeout = commTimeoutSeconds();
    auto         task = std::make_shared<std::packaged_task<ncclResult_t()>>([deviceOrdinal, worldSize, rank, id, comm = &mImpl->comm]() -> ncclResult_t {
        if (hipSetDevice(deviceOrdinal) != hipSuccess) return ncclUnhandledCudaError;
        return ncclCommInitRank(comm, static_cast<int>(worldSize), id, static_cast<int>(rank));
    });
    std::future<ncclResult_t> done = task->get_future();
    std::thread([task] { (*task)(); }).detach();
    if (timeout > 0.0 && done.wait_for(std::chrono::duration<double>(timeout)) != std::future_status::ready)
    {
        mImpl.release(); // the helper still writes into it: leaked on purpose
        throw std::runtime_error("RCCL: ncclCommInitRank did not complete within " + std::to_string(static_cast<int>(timeout)) + " s on rank " + std::to_string(rank) + " of " +
                                 std::to_string(worldSize) + " (a rank is missing, or the ranks do not share one unique id); RF_COMM_TIMEOUT_S changes the limit");
    }
    RF_NCCL(done.get());
    int count = 0;
    RF_NCCL(ncclCommCount(mImpl->comm, &count));
    if (static_cast<uint32_t>(count) != worldSize) throw std::runtime_error("RCCL communicator spans " + std::to_string(count) + " ranks, expected " + std::to_string(worldSize));
}

TileComm::~TileComm()
{
    if (mImpl && mImpl->comm)
    {
        (void)hipSetDevice(mImpl->device);
        (void)ncclCommDestroy(mImpl->comm);
    }
    if (mImpl && mImpl->exchangeStart)
    {
        (void)hipEventDestroy(mImpl->exchangeStart);
        (void)hipEventDestroy(mImpl->exchangeStop);
    }
}

uint32_t TileComm::rank() const { return mImpl->rank; }
uint32_t TileComm::worldSize() const { return mImpl->world; }
int      TileComm::deviceOrdinal() const { return mImpl->device; }

bool TileComm::localTransport() const { return mImpl->fabric != nullptr; }

void TileComm::rcclInfo(uint32_t& count, uint32_t& userRank, int& device) const
{
    if (mImpl->fabric)
    {
        count = mImpl->world, userRank = mImpl->rank, device = mImpl->device; // (no RCCL communicator behind it: localTransport() says so)
        return;
    }
    int c = 0, r = 0, d = 0;
    RF_NCCL(ncclCommCount(mImpl->comm, &c));
    RF_NCCL(ncclCommUserRank(mImpl->comm, &r));
    RF_NCCL(ncclCommCuDevice(mImpl->comm, &d));
    count = static_cast<uint32_t>(c), userRank = static_cast<uint32_t>(r), device = d;
}

const void* TileComm::gatherFrame(const void* compactDevice, uint32_t width, uint32_t height, uint32_t root, void* streamHandle, bool loopback)
{
    Impl& m = *mImpl;
    if (m.comm == nullptr && !m.fabric) throw std::runtime_error("the RCCL communicator was aborted");
    if (root >= m.world) throw std::invalid_argument("gather root out of range");
    if (width == 0 || height == 0) throw std::invalid_argument("empty frame");
    hipStream_t stream = static_cast<hipStream_t>(streamHandle);
    RF_HIP(hipSetDevice(m.device));
    m.ensureLayout(width, height, stream);
    const GatherLayout& g = m.layout;
    const auto          tilesOf = [&](uint32_t r) { return g.rankFirstTile[r + 1] - g.rankFirstTile[r]; };
    const size_t        floatsPerTile = static_cast<size_t>(kTilePixels) * 4;
    const bool          isRoot = m.rank == root;
    if (tilesOf(m.rank) > 0 && compactDevice == nullptr) throw std::invalid_argument("null tile buffer");
    if (isRoot)
    {
        const size_t stagingWant = static_cast<size_t>(g.rankFirstTile[m.world]) * kTilePixels, imageWant = static_cast<size_t>(width) * height;
        if (stagingWant > m.staging.n || imageWant > m.image.n) RF_HIP(hipStreamSynchronize(stream)); // a consumer of the old buffers may still run
        m.staging.ensure(stagingWant);
        m.image.ensure(imageWant);
        m.imageW = width;
        m.imageH = height;
    }

    // one group: exactly the operations of gatherPlan() (the list the CPU tests check for every rank of a world)
    const std::vector<GatherOp> plan = gatherPlan(g, m.world, m.rank, root, loopback);
    // (first exchange only) marks the end of what was queued on the stream BEFORE the exchange -- this rank's frame kernels: the
    // watchdog below measures the exchange, not the render in front of it
    hipEvent_t queuedBefore = nullptr;
    if (!m.firstGatherDone)
    {
        RF_HIP(hipEventCreateWithFlags(&queuedBefore, hipEventDisableTiming));
        RF_HIP(hipEventRecord(queuedBefore, stream));
    }
    struct EventGuard
    {
        hipEvent_t& e;
        ~EventGuard()
        {
            if (e) (void)hipEventDestroy(e);
        }
    } eventGuard{queuedBefore};
    if (m.exchangeStart == nullptr)
    {
        RF_HIP(hipEventCreate(&m.exchangeStart));
        RF_HIP(hipEventCreate(&m.exchangeStop));
    }
    m.exchangeTimed = false;
    RF_HIP(hipEventRecord(m.exchangeStart, stream));
    if (m.fabric)
    {
        // ---- the local transport: the plan's operations as device-to-device copies between the ranks' buffers (see LocalFabric)
        LocalFabric&  f = *m.fabric;
        const double  timeout = commTimeoutSeconds();
        const auto    waitFor = [&](std::unique_lock<std::mutex>& lock, auto&& ready, const char* what) {
            if (timeout > 0.0)
            {
                if (!f.cv.wait_for(lock, std::chrono::duration<double>(timeout), ready))
                    throw std::runtime_error(std::string("local transport: ") + what + " did not happen within " + std::to_string(static_cast<int>(timeout)) + " s on rank " +
                                             std::to_string(m.rank) + " (a peer is missing or disagrees about frame size / root)");
            }
            else f.cv.wait(lock, ready);
        };
        std::vector<std::shared_ptr<LocalPost>> mine;
        for (const GatherOp& op : plan) // 1. post every send
        {
            if (!op.isSend) continue;
            auto post = std::make_shared<LocalPost>();
            post->src = static_cast<const float*>(compactDevice) + op.offsetTiles * floatsPerTile;
            post->bytes = op.countTiles * floatsPerTile * sizeof(float);
            post->srcDevice = m.device;
            {
                std::lock_guard<std::mutex> lock(f.mutex);
                post->ready = f.takeEvent();
            }
            RF_HIP(hipEventRecord(post->ready, stream));
            {
                std::lock_guard<std::mutex> lock(f.mutex);
                f.mail[{m.rank, op.peer}].push_back(post);
            }
            f.cv.notify_all();
            mine.push_back(post);
        }
        for (const GatherOp& op : plan) // 2. every receive: wait for the peer's send, copy behind it
        {
            if (op.isSend) continue;
            std::shared_ptr<LocalPost> post;
            {
                std::unique_lock<std::mutex> lock(f.mutex);
                auto&                        box = f.mail[{op.peer, m.rank}];
                waitFor(lock, [&] { return !box.empty(); }, "a peer's send");
                post = box.front();
                box.pop_front();
                post->consumed = f.takeEvent();
            }
            if (post->bytes != op.countTiles * floatsPerTile * sizeof(float))
                throw std::runtime_error("local transport: rank " + std::to_string(op.peer) + " sends " + std::to_string(post->bytes) + " bytes, rank " + std::to_string(m.rank) + " expects " +
                                         std::to_string(op.countTiles * floatsPerTile * sizeof(float)) + " (the ranks disagree about the frame)");
            RF_HIP(hipStreamWaitEvent(stream, post->ready, 0));
            RF_HIP(hipMemcpyAsync(m.staging.p + static_cast<size_t>(op.offsetTiles) * kTilePixels, post->src, post->bytes, hipMemcpyDeviceToDevice, stream));
            RF_HIP(hipEventRecord(post->consumed, stream));
            {
                std::lock_guard<std::mutex> lock(f.mutex);
                post->taken = true;
            }
            f.cv.notify_all();
        }
        for (const std::shared_ptr<LocalPost>& post : mine) // 3. my sends: the buffer is mine again once the receiver's copy has run
        {
            {
                std::unique_lock<std::mutex> lock(f.mutex);
                waitFor(lock, [&] { return post->taken; }, "the root's receive");
            }
            RF_HIP(hipStreamWaitEvent(stream, post->consumed, 0));
            std::lock_guard<std::mutex> lock(f.mutex);
            f.freeEvents.push_back(post->ready);
            f.freeEvents.push_back(post->consumed);
        }
        m.firstGatherDone = true;
    }
    else
    {
    RF_NCCL(ncclGroupStart());
    try
    {
        for (const GatherOp& op : plan)
        {
            if (op.isSend)
                RF_NCCL(ncclSend(static_cast<const float*>(compactDevice) + op.offsetTiles * floatsPerTile, op.countTiles * floatsPerTile, ncclFloat, static_cast<int>(op.peer), m.comm, stream));
            else
                RF_NCCL(ncclRecv(m.staging.p + static_cast<size_t>(op.offsetTiles) * kTilePixels, op.countTiles * floatsPerTile, ncclFloat, static_cast<int>(op.peer), m.comm, stream));
        }
    }
    catch (...)
    {
        (void)ncclGroupEnd(); // never leave the thread inside an open group
        throw;
    }
    RF_NCCL(ncclGroupEnd());
    if (!m.firstGatherDone)
    {
        // The first exchange of a communicator also sets up its peer connections; a peer that never posts its side (it died, it
        // disagrees about the frame size or the root) would leave this stream stuck forever.  Watch this one exchange: past the
        // time limit the communicator is aborted and the caller gets an error instead of a hang.  Later gathers are not watched
        // (they stay fully asynchronous).  NOTE: an exception thrown on one rank leaves its peers waiting in their own gather
        // until THEIR limit expires.  The clock starts when the work queued on this stream before the exchange (Renderer::render()
        // is asynchronous: the whole frame may still be in front of it) has drained, so a long first frame does not abort a healthy
        // job; what remains inside the limit is the connection set-up, the transfer, and the wait for the slowest peer to finish ITS
        // frame -- ranks own equal tile counts (+-1), so that wait is a fraction of a frame.
        // A second, much larger cap (kDrainFactor x the limit, from loop entry) covers the case the first clock never starts in: this rank's OWN queued kernels
        // never drain (a hung or faulted kernel keeps the event at hipErrorNotReady forever) -- the loop would otherwise spin without any limit.
        constexpr double kDrainFactor = 20.0;
        const double timeout = commTimeoutSeconds();
        const auto   entered = std::chrono::steady_clock::now();
        auto         t0 = entered;
        bool         drained = false;
        for (;;)
        {
            const hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) RF_HIP(q);
            if (!drained)
            {
                const hipError_t e = hipEventQuery(queuedBefore);
                if (e == hipSuccess)
                {
                    drained = true;
                    t0 = std::chrono::steady_clock::now();
                }
                else if (e != hipErrorNotReady) RF_HIP(e);
            }
            ncclResult_t async = ncclSuccess;
            RF_NCCL(ncclCommGetAsyncError(m.comm, &async));
            if (async != ncclSuccess && async != ncclInProgress) throw std::runtime_error(std::string("RCCL error during the first frame exchange: ") + ncclGetErrorString(async));
            if (drained && timeout > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout)
            {
                (void)ncclCommAbort(m.comm);
                m.comm = nullptr;
                throw std::runtime_error("RCCL: the first frame exchange did not complete within " + std::to_string(static_cast<int>(timeout)) + " s on rank " +
                                         std::to_string(m.rank) + " (a peer is missing or disagrees about frame size / root); communicator aborted");
            }
            if (!drained && timeout > 0.0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - entered).count() > kDrainFactor * timeout)
            {
                (void)ncclCommAbort(m.comm);
                m.comm = nullptr;
                throw std::runtime_error("RCCL: the work queued in front of the first frame exchange did not drain within " + std::to_string(static_cast<int>(kDrainFactor * timeout)) +
                                         " s on rank " + std::to_string(m.rank) + " (a kernel of this rank's own frame hangs); communicator aborted");
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        m.firstGatherDone = true;
    }
    } // (RCCL transport)
    if (!isRoot)
    {
        RF_HIP(hipEventRecord(m.exchangeStop, stream));
        m.exchangeTimed = true;
        return nullptr;
    }

    const uint32_t numTiles = g.tilesX * g.tilesY;
    hipLaunchKernelGGL(kUntile, dim3(numTiles), dim3(256), 0, stream, m.staging.p, static_cast<const float4*>(compactDevice), loopback ? 0xFFFFFFFFu : m.rank,
                       g.rankFirstTile[m.rank], m.dTileSlot.p, m.dTileOwner.p, width, height, g.tilesX, m.image.p);
    RF_HIP(hipGetLastError());
    RF_HIP(hipEventRecord(m.exchangeStop, stream));
    m.exchangeTimed = true;
    return m.image.p;
}

double TileComm::lastExchangeMs()
{
    Impl& m = *mImpl;
    if (!m.exchangeTimed) return -1.0;
    RF_HIP(hipSetDevice(m.device));
    RF_HIP(hipEventSynchronize(m.exchangeStop));
    float ms = 0.0f;
    RF_HIP(hipEventElapsedTime(&ms, m.exchangeStart, m.exchangeStop));
    return static_cast<double>(ms);
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
    if (m.fabric)
    {
        // local transport: a generation barrier over the fabric's ranks (host side; the caller's stream is drained first, as the RCCL path's read-back does)
        RF_HIP(hipStreamSynchronize(stream));
        LocalFabric&                 f = *m.fabric;
        std::unique_lock<std::mutex> lock(f.mutex);
        const uint64_t               gen = f.generation;
        f.running = f.arrived == 0 ? value : std::max(f.running, value);
        if (++f.arrived == f.world)
        {
            f.result = f.running;
            f.arrived = 0;
            ++f.generation;
            f.cv.notify_all();
            return f.result;
        }
        const double timeout = commTimeoutSeconds();
        const auto   released = [&] { return f.generation != gen; };
        if (timeout > 0.0)
        {
            if (!f.cv.wait_for(lock, std::chrono::duration<double>(timeout), released)) throw std::runtime_error("local transport: all-reduce timed out on rank " + std::to_string(m.rank));
        }
        else f.cv.wait(lock, released);
        return f.result;
    }
    m.scalar.ensure(2);
    RF_HIP(hipMemcpyAsync(m.scalar.p, &value, sizeof value, hipMemcpyHostToDevice, stream));
    RF_NCCL(ncclAllReduce(m.scalar.p, m.scalar.p + 1, 1, ncclDouble, ncclMax, m.comm, stream));
    double out = 0.0;
    RF_HIP(hipMemcpyAsync(&out, m.scalar.p + 1, sizeof out, hipMemcpyDeviceToHost, stream));
    RF_HIP(hipStreamSynchronize(stream));
    return out;
}
} // namespace rf
