// rf_comm.hpp -- the multi-GPU frame exchange: one RCCL gather of tile shards at frame end + a device un-tile.
//
// No reference counterpart (the reference is single-device: src/pt/reference_path_tracer.cpp:565-595 draws one
// full-screen quad on one WGPUDevice).  Contract: the image is cut into 32x32
// tiles dealt to ranks by tilesForRank(); every rank renders its tiles into a compact tile-major float4 buffer;
// at frame end every rank sends that buffer to the root over RCCL (point-to-point ncclSend / ncclRecv in one
// group: all of the root's xGMI ingress links are used at once; no reduction, no ring), and the root turns the
// shards into the row-major width x height float4 image with one kernel.  One process (or host thread) per GPU.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

namespace rf
{
constexpr uint32_t kCommIdBytes = 128; // NCCL_UNIQUE_ID_BYTES

// Where every tile of the frame lives in the root's staging area: shards are stored rank after rank, each
// rank's tiles in ascending tile id (the order of tilesForRank()).  Pure host arithmetic (tested on CPU).
struct GatherLayout
{
    uint32_t              tilesX = 0, tilesY = 0;
    std::vector<uint32_t> rankFirstTile; // world + 1 entries: staging offset of rank r's shard, in tiles
    std::vector<uint32_t> tileSlot;      // per tile id: staging position in tiles (rankFirstTile[owner] + index in the owner's list)
    std::vector<uint32_t> tileOwner;     // per tile id: owning rank
};
GatherLayout gatherLayout(uint32_t width, uint32_t height, uint32_t worldSize);

// The point-to-point operations ONE rank posts (inside one RCCL group) for a frame-end gather to `root`: what
// TileComm::gatherFrame() executes, as data.  Offsets and counts are in tiles (1024 float4 each): a receive lands
// at `offsetTiles` of the root's staging area, a send starts at `offsetTiles` (always 0) of the rank's own compact
// buffer.  Pure host arithmetic: the plan of every rank of a world can be checked without a GPU
// (tests/test_distributed_cpu.py: every send has its receive, the receives tile the staging area exactly).
struct GatherOp
{
    uint32_t isSend;      // 1: ncclSend to `peer`, 0: ncclRecv from `peer`
    uint32_t peer;
    uint32_t offsetTiles;
    uint32_t countTiles;
};
std::vector<GatherOp> gatherPlan(const GatherLayout& layout, uint32_t worldSize, uint32_t rank, uint32_t root, bool loopback);

// HIP devices this process sees (0 without a GPU or a driver; never throws)
int deviceCount();

class TileComm
{
public:
    // rank 0 calls uniqueId() and hands the 128 bytes to the other ranks through the host application's own
    // channel (file, socket, torch.distributed store ...); then every rank constructs its TileComm (collective).
    static void uniqueId(uint8_t out[kCommIdBytes]);
    TileComm(const uint8_t id[kCommIdBytes], uint32_t rank, uint32_t worldSize, int deviceOrdinal);
    ~TileComm();
    TileComm(const TileComm&) = delete;
    TileComm& operator=(const TileComm&) = delete;

    uint32_t rank() const;
    uint32_t worldSize() const;
    int      deviceOrdinal() const;
    // What RCCL itself reports for this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): proof of how
    // many ranks the exchange really spans.
    void rcclInfo(uint32_t& count, uint32_t& userRank, int& device) const;
    // true: this communicator runs on the LOCAL test transport (RF_COMM_TRANSPORT=local when its id was made: N ranks in one process, one host thread each,
    // ncclSend / ncclRecv replaced by device-to-device copies -- the same plan, staging offsets and un-tile; rf_comm.hip), not on RCCL
    bool localTransport() const;

    // Frame-end exchange, enqueued on `stream` (a hipStream_t: the renderer's, so the exchange is ordered behind
    // the frame's kernels).  compactDevice: this rank's tile-major buffer (tilesForRank(...).size() * 1024 float4).
    // On the root the row-major width * height float4 image is produced in device memory owned by this object
    // (returned; valid until the next gather); other ranks get nullptr.  loopback: the root's own shard also
    // travels through ncclSend / ncclRecv (to itself) instead of being read in place -- the world-size-1 self-test.
    const void* gatherFrame(const void* compactDevice, uint32_t width, uint32_t height, uint32_t root, void* stream, bool loopback = false);
    // Device time of the LAST gatherFrame() on the caller's stream, HIP events around it: from the moment the rank's queued frame kernels have drained and the
    // exchange starts to the end of its sends / receives (+ the un-tile on the root).  Waits for that exchange; -1 before the first one.
    double lastExchangeMs();
    // Root: wait for the stream and copy the gathered image to the host (width * height * 4 floats, row-major).
    void readFrame(float* dstHost, void* stream);
    // Max over ranks of a host double / barrier (timing plumbing for callers that have no other collective layer).
    double allReduceMax(double value, void* stream);

private:
    struct Impl;
    std::unique_ptr<Impl> mImpl;
};
} // namespace rf
