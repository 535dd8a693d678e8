// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access patterns of the traversal
// kernel (MI355X_MICROARCH.md "HBM": the x2 correction is calibrated for wide coalesced streaming reads only;
// "calibrate on a known byte count in your own access pattern").  Every kernel below moves a KNOWN number of
// bytes; run it once plainly (prints the byte counts and times as JSON lines) and once per counter under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ...    /    --pmc WRITE_SIZE ...
// (tools/roofline_pmc.sh) and divide.  Patterns:
//   stream16<T>   coalesced 16 B/lane reads of the whole table (the guide's calibrated case: expect 0.5)
//   gather56<T>   per lane: 3 x dwordx4 + 1 x dwordx2 = 56 of the 64 bytes of a random 64-B-aligned record,
//                 next record depends on the data (kTraceWide's record fetch).  T = log2(table bytes):
//                 25 = 32 MiB (the size of the atrium's BVH + triangles: L2 misses that hit the Infinity Cache),
//                 30 = 1 GiB, 33 = 8 GiB (past the 256-MiB Infinity Cache: every record comes from HBM)
//   gather12<T>   per lane: one dwordx3 of a random 16-B element (path-state streams read at refill)
//   scatter16<T>  per lane: one 16-B store to a random 16-B element (the hit record written per finished ray)
//   fill16<T>     coalesced 16-B stores of the whole table
//   gatherWide<T, BYTES, CHAINS>  (`fetch_calib wide`) random BYTES-byte records, whole record read, CHAINS chains per lane
// and two cooperative forms of the record fetch (design probes for kTraceWide: how many vector-L1 accesses does a
// record cost when the lanes of a pair / quad address the SAME line in one instruction?):
//   gatherPair<T>    lanes 2p, 2p+1 share one record: each loads its 32-byte half (2 x dwordx4), halves are
//                    exchanged with DPP quad_perm -- 32 records per wave-step
//   gatherPhased<T>  64 records per wave-step, fetched in two phases: in phase A both lanes of a pair load the two
//                    28-byte halves of the EVEN lane's record, in phase B those of the ODD lane's record (4 load
//                    instructions per lane as in gather56, but every instruction touches 32 lines instead of 64);
//                    per-half results are exchanged with DPP
//   gatherQuadLds<T> instruction k: the 4 lanes of a quad load the 4 pieces of quad-lane k's record (one line per
//                    quad per instruction), pieces parked in LDS, every lane reads its own 64 bytes back
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e = (x);                                                                        \
        if (e != hipSuccess)                                                                       \
        {                                                                                          \
            std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                     \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

typedef float v3f __attribute__((ext_vector_type(3)));

template<int T>
__global__ __launch_bounds__(256) void stream16(const float4* __restrict__ table, size_t n16, float* out)
{
    float acc = 0.0f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * 256ull)
    {
        const float4 v = table[i];
        acc += v.x + v.w;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

template<int T>
__global__ __launch_bounds__(256) void gather56(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    float          acc = 0.0f;
    uint32_t       idx = hash32(tid) & mask;
    for (int it = 0; it < iters; ++it)
    {
        const float4* r = table + 4 * static_cast<size_t>(idx);
        const float4  a = r[0], b = r[1], c = r[2];
        const uint2*  wp = reinterpret_cast<const uint2*>(r + 3);
        asm volatile("" : "+v"(wp));
        const uint2 w = *wp;
        acc += a.x + b.y + c.z;
        idx = hash32(idx + w.x + w.y + it) & mask; // dependent on the loaded data
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

__device__ __forceinline__ uint32_t dppSwapPair(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true));
}

template<int T>
__global__ __launch_bounds__(256) void gatherPair(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t half = threadIdx.x & 1u;
    float          acc = 0.0f;
    uint32_t       idx = hash32(tid >> 1) & mask; // the two lanes of a pair walk the same chain
    for (int it = 0; it < iters; ++it)
    {
        const float4* r = table + 4 * static_cast<size_t>(idx) + 2 * half;
        const float4  a = r[0], b = r[1];
        acc += a.x + b.y;
        const uint32_t mine = __float_as_uint(a.w) + __float_as_uint(b.w);
        const uint32_t other = dppSwapPair(mine);
        idx = hash32(idx + mine + other + it) & mask;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

template<int T>
__global__ __launch_bounds__(256) void gatherPhased(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t half = threadIdx.x & 1u;
    const bool     even = half == 0u;
    float          acc = 0.0f;
    uint32_t       idx = hash32(tid) & mask; // every lane walks its own chain
    for (int it = 0; it < iters; ++it)
    {
        const uint32_t partner = dppSwapPair(idx);
        const uint32_t recA = even ? idx : partner, recB = even ? partner : idx;
        const float4*  pa = table + 4 * static_cast<size_t>(recA) + 2 * half;
        const float4*  pb = table + 4 * static_cast<size_t>(recB) + 2 * half;
        const float4   a0 = pa[0];
        const v3f      a1 = *reinterpret_cast<const v3f*>(pa + 1);
        const float4   b0 = pb[0];
        const v3f      b1 = *reinterpret_cast<const v3f*>(pb + 1);
        acc += a0.x + b0.y;
        const uint32_t fromA = __float_as_uint(a0.w) + __float_as_uint(a1.z), fromB = __float_as_uint(b0.w) + __float_as_uint(b1.z);
        const uint32_t recvA = dppSwapPair(fromA), recvB = dppSwapPair(fromB);
        const uint32_t own = even ? fromA + recvA : fromB + recvB; // both halves of this lane's own record
        idx = hash32(idx + own + it) & mask;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

template<int T>
__global__ __launch_bounds__(256) void gatherQuadLds(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    // plane k holds piece j of the record of quad-lane k at [k][thread]; planes are 16 bytes longer than 256 float4
    // so that the four lanes of a quad (same column, different planes) fall into different banks on the read-back
    __shared__ float4 stage[4][256 + 1];
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, j = lane & 3u, qb = threadIdx.x & ~3u;
    float          acc = 0.0f;
    uint32_t       idx = hash32(tid) & mask;
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
            const uint32_t idxK = __shfl(idx, (lane & ~3u) + k);
            stage[k][threadIdx.x] = table[4 * static_cast<size_t>(idxK) + j];
        }
        __builtin_amdgcn_wave_barrier();
        const float4 a = stage[j][qb + 0], b = stage[j][qb + 1], c = stage[j][qb + 2], d = stage[j][qb + 3];
        __builtin_amdgcn_wave_barrier();
        acc += a.x + b.y + c.z;
        idx = hash32(idx + __float_as_uint(d.x) + __float_as_uint(d.y) + it) & mask;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

// Round 5: does a WIDER record raise the random-gather rate from HBM?  gatherWide<T, BYTES, CHAINS>: per lane, CHAINS independent
// dependent-fetch chains, each step reading all BYTES (16-B pieces: BYTES / 16 dwordx4) of a random BYTES-aligned record.  BYTES = 64 is gather56's
// pattern with the whole line read; 128 / 256 = what an 8-wide / 16-wide BVH record would ask for.  CHAINS = 2 doubles the requests in flight per
// lane (is the 64-B rate latency-bound or bound by the memory system?).
template<int T, int BYTES, int CHAINS>
__global__ __launch_bounds__(256) void gatherWide(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    constexpr int  kPieces = BYTES / 16;
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    float          acc = 0.0f;
    uint32_t       idx[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) idx[c] = hash32(tid * CHAINS + c) & mask;
    for (int it = 0; it < iters; ++it)
    {
        float4 v[CHAINS][kPieces];
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
        {
            const float4* r = table + kPieces * static_cast<size_t>(idx[c]);
#pragma unroll
            for (int k = 0; k < kPieces; ++k) v[c][k] = r[k];
        }
#pragma unroll
        for (int c = 0; c < CHAINS; ++c)
        {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < kPieces; ++k) acc += v[c][k].x, w += __float_as_uint(v[c][k].w);
            idx[c] = hash32(idx[c] + w + it) & mask;
        }
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

template<int T>
__global__ __launch_bounds__(256) void gather12(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    float          acc = 0.0f;
    uint32_t       idx = hash32(tid) & mask;
    for (int it = 0; it < iters; ++it)
    {
        const v3f v = *reinterpret_cast<const v3f*>(table + idx);
        acc += v.x;
        idx = hash32(idx + __float_as_uint(v.z) + it) & mask;
    }
    if (acc == 1.2345e-30f) out[0] = acc;
}

template<int T>
__global__ __launch_bounds__(256) void scatter16(float4* __restrict__ table, uint32_t mask, int iters)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    uint32_t       idx = hash32(tid) & mask;
    for (int it = 0; it < iters; ++it)
    {
        table[idx] = make_float4(1.0f, 2.0f, 3.0f, __uint_as_float(idx));
        idx = hash32(idx + it) & mask;
    }
}

template<int T>
__global__ __launch_bounds__(256) void fill16(float4* __restrict__ table, size_t n16)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += static_cast<size_t>(gridDim.x) * 256ull)
        table[i] = make_float4(1.0f, 2.0f, 3.0f, __uint_as_float(static_cast<uint32_t>(i * 2654435761u)));
}

struct Timer
{
    hipEvent_t a, b;
    Timer()
    {
        CHECK(hipEventCreate(&a));
        CHECK(hipEventCreate(&b));
    }
    void  start() { CHECK(hipEventRecord(a)); }
    float stop()
    {
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        return ms;
    }
};

template<int T>
void runTable(float4* table, float* out, Timer& tm)
{
    const size_t   bytes = 1ull << T;
    const size_t   n16 = bytes / 16;
    const uint32_t recMask = static_cast<uint32_t>(bytes / 64 - 1), elemMask = static_cast<uint32_t>(n16 - 1 > 0xFFFFFFFFull ? 0xFFFFFFFFull : n16 - 1);
    const int      blocks = 256 * 8, iters = 512;
    const double   threads = blocks * 256.0;
    tm.start(); hipLaunchKernelGGL(fill16<T>, dim3(blocks), dim3(256), 0, 0, table, n16); float ms = tm.stop();
    std::printf("{\"kernel\": \"fill16<%d>\", \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", T, (double)bytes, ms, bytes / ms * 1e-6);
    for (int rep = 0; rep < 2; ++rep) // second pass: steady state (for the 32-MiB table: Infinity-Cache resident)
    {
        tm.start(); hipLaunchKernelGGL(stream16<T>, dim3(blocks), dim3(256), 0, 0, table, n16, out); ms = tm.stop();
        std::printf("{\"kernel\": \"stream16<%d>\", \"rep\": %d, \"bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", T, rep, (double)bytes, ms, bytes / ms * 1e-6);
        tm.start(); hipLaunchKernelGGL(gather56<T>, dim3(blocks), dim3(256), 0, 0, table, recMask, iters, out); ms = tm.stop();
        std::printf("{\"kernel\": \"gather56<%d>\", \"rep\": %d, \"records\": %.0f, \"bytes_requested\": %.0f, \"bytes_lines\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, rep,
                    threads * iters, threads * iters * 56.0, threads * iters * 64.0, ms, threads * iters / ms * 1e-6);
        tm.start(); hipLaunchKernelGGL(gatherPair<T>, dim3(blocks), dim3(256), 0, 0, table, recMask, iters, out); ms = tm.stop();
        std::printf("{\"kernel\": \"gatherPair<%d>\", \"rep\": %d, \"records\": %.0f, \"bytes_lines\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, rep, threads * iters / 2,
                    threads * iters / 2 * 64.0, ms, threads * iters / 2 / ms * 1e-6);
        tm.start(); hipLaunchKernelGGL(gatherPhased<T>, dim3(blocks), dim3(256), 0, 0, table, recMask, iters, out); ms = tm.stop();
        std::printf("{\"kernel\": \"gatherPhased<%d>\", \"rep\": %d, \"records\": %.0f, \"bytes_lines\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, rep, threads * iters,
                    threads * iters * 64.0, ms, threads * iters / ms * 1e-6);
        tm.start(); hipLaunchKernelGGL(gatherQuadLds<T>, dim3(blocks), dim3(256), 0, 0, table, recMask, iters, out); ms = tm.stop();
        std::printf("{\"kernel\": \"gatherQuadLds<%d>\", \"rep\": %d, \"records\": %.0f, \"bytes_lines\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, rep, threads * iters,
                    threads * iters * 64.0, ms, threads * iters / ms * 1e-6);
        tm.start(); hipLaunchKernelGGL(gather12<T>, dim3(blocks), dim3(256), 0, 0, table, elemMask, iters, out); ms = tm.stop();
        std::printf("{\"kernel\": \"gather12<%d>\", \"rep\": %d, \"records\": %.0f, \"bytes_requested\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, rep, threads * iters,
                    threads * iters * 12.0, ms, threads * iters / ms * 1e-6);
    }
    tm.start(); hipLaunchKernelGGL(scatter16<T>, dim3(blocks), dim3(256), 0, 0, table, elemMask, iters); ms = tm.stop();
    std::printf("{\"kernel\": \"scatter16<%d>\", \"records\": %.0f, \"bytes_requested\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f}\n", T, threads * iters, threads * iters * 16.0, ms,
                threads * iters / ms * 1e-6);
    std::fflush(stdout);
}

template<int T, int BYTES, int CHAINS>
void runWideOne(float4* table, float* out, Timer& tm, int blocks)
{
    const size_t   bytes = 1ull << T;
    const uint32_t mask = static_cast<uint32_t>(bytes / BYTES - 1);
    const int      iters = 256 * 64 / BYTES * (CHAINS == 1 ? 2 : 1);
    const double   records = blocks * 256.0 * iters * CHAINS;
    float          best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
        tm.start(); hipLaunchKernelGGL((gatherWide<T, BYTES, CHAINS>), dim3(blocks), dim3(256), 0, 0, table, mask, iters, out); const float ms = tm.stop();
        best = ms < best ? ms : best;
    }
    std::printf("{\"kernel\": \"gatherWide<%d,%d,%d>\", \"blocks\": %d, \"records\": %.0f, \"bytes\": %.0f, \"ms\": %.4f, \"Grecords_per_s\": %.2f, \"GBps\": %.1f}\n", T, BYTES, CHAINS, blocks,
                records, records * BYTES, best, records / best * 1e-6, records * BYTES / best * 1e-6);
    std::fflush(stdout);
}
template<int T>
void runWide(float4* table, float* out, Timer& tm)
{
    const size_t n16 = (1ull << T) / 16;
    hipLaunchKernelGGL(fill16<T>, dim3(2048), dim3(256), 0, 0, table, n16);
    for (const int blocks : {2048, 4096}) // 8 / 16 waves per SIMD resident (the second: as many as the registers allow)
    {
        runWideOne<T, 64, 1>(table, out, tm, blocks);
        runWideOne<T, 64, 2>(table, out, tm, blocks);
        runWideOne<T, 128, 1>(table, out, tm, blocks);
        runWideOne<T, 128, 2>(table, out, tm, blocks);
        runWideOne<T, 256, 1>(table, out, tm, blocks);
    }
}

int main(int argc, char** argv)
{
    float4* table;
    float*  out;
    CHECK(hipMalloc(&table, 1ull << 33));
    CHECK(hipMalloc(&out, 64));
    Timer tm;
    if (argc > 1 && std::string(argv[1]) == "wide") // only the record-width probe (random gathers of 64 / 128 / 256-byte records)
    {
        runWide<25>(table, out, tm);
        runWide<31>(table, out, tm);
        runWide<33>(table, out, tm);
        CHECK(hipDeviceSynchronize());
        return 0;
    }
    runTable<25>(table, out, tm);
    runTable<30>(table, out, tm);
    runTable<33>(table, out, tm);
    CHECK(hipDeviceSynchronize());
    return 0;
}
