// Microbenchmark: cost of divergent 64-byte record gathers on gfx950, by access shape.
//   A: each lane issues 4 x dwordx4 at its own record (what kTraceWide does)
//   B: quad-cooperative: in instruction k, the 4 lanes of a quad load the 4 pieces of the record
//      of quad-lane k (each quad touches ONE 64-B line per instruction); data redistributed by DPP
//   C: each lane issues 2 x dwordx4 (32-byte records)
//   D: as A but only 50 % of the lanes active (exec-masked)
// Records are chosen by a hash of (lane, iteration): effectively random over a 16 MiB table.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template<int MODE>
__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, uint32_t mask, int iters, float* out)
{
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    float acc = 0.0f;
    uint32_t idx = hash32(tid) & mask;
    for (int it = 0; it < iters; ++it)
    {
        if (MODE == 0 || MODE == 3)
        {
            if (MODE == 3 && (lane & 1)) { idx = hash32(idx + it) & mask; continue; }
            const float4* r = table + 4 * (size_t)idx;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc += a.x + b.y + c.z + d.w;
            idx = hash32(idx + __float_as_uint(a.w) + it) & mask;   // dependent on the loaded data
        }
        else if (MODE == 1)
        {
            // quad-cooperative: lane j of the quad loads piece j of the records of quad-lanes 0..3
            const uint32_t j = lane & 3;
            float4 piece[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t idxK = __shfl(idx, (lane & ~3u) + k);
                piece[k] = table[4 * (size_t)idxK + j];
            }
            // lane j needs piece i (i = 0..3) of its own record = piece[j] held by quad-lane i
            float4 mine[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                float4 v;
                // select piece[j] on the source lane: every lane publishes, for each destination j', its piece[j'];
                // do it with 4 shuffles of the matching register
                float4 cand[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                {
                    cand[k].x = __shfl(piece[k].x, (lane & ~3u) + i);
                    cand[k].y = __shfl(piece[k].y, (lane & ~3u) + i);
                    cand[k].z = __shfl(piece[k].z, (lane & ~3u) + i);
                    cand[k].w = __shfl(piece[k].w, (lane & ~3u) + i);
                }
                v = j == 0 ? cand[0] : (j == 1 ? cand[1] : (j == 2 ? cand[2] : cand[3]));
                mine[i] = v;
            }
            acc += mine[0].x + mine[1].y + mine[2].z + mine[3].w;
            idx = hash32(idx + __float_as_uint(mine[0].w) + it) & mask;
        }
        else if (MODE == 2)
        {
            const float4* r = table + 2 * (size_t)idx;
            const float4 a = r[0], b = r[1];
            acc += a.x + b.y;
            idx = hash32(idx + __float_as_uint(a.w) + it) & mask;
        }
        else if (MODE == 5)
        {
            // 128-byte records (128-B aligned), lane-private 8 x dwordx4: a 4-wide BVH node
            const float4* r = table + 8 * (size_t)idx;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6], h = r[7];
            acc += a.x + b.y + c.z + d.w + e.x + f.y + g.z + h.w;
            idx = hash32(idx + __float_as_uint(a.w) + it) & mask;
        }
        else if (MODE == 6)
        {
            // 96 of the 128 bytes (6 x dwordx4)
            const float4* r = table + 8 * (size_t)idx;
            const float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
            acc += a.x + b.y + c.z + d.w + e.x + f.y;
            idx = hash32(idx + __float_as_uint(a.w) + it) & mask;
        }
        else if (MODE == 4)
        {
            // quad-cooperative through LDS: 4 loads land in LDS, each lane reads its 64 B back
            __shared__ float4 stage[4][256];
            const uint32_t j = lane & 3;
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
                const uint32_t idxK = __shfl(idx, (lane & ~3u) + k);
                stage[k][threadIdx.x] = table[4 * (size_t)idxK + j];
            }
            // stage[k][quadBase + i] = piece i of the record of quad-lane k
            const uint32_t qb = threadIdx.x & ~3u;
            const float4 a = stage[j][qb + 0], b = stage[j][qb + 1], c = stage[j][qb + 2], d = stage[j][qb + 3];
            acc += a.x + b.y + c.z + d.w;
            idx = hash32(idx + __float_as_uint(a.w) + it) & mask;
        }
    }
    out[tid] = acc;
}

int main(int argc, char** argv)
{
    const size_t records = argc > 1 ? (size_t)atol(argv[1]) : (1u << 18);   // default 256 Ki x 64 B = 16 MiB
    std::vector<float4> host(4 * records);
    for (size_t i = 0; i < host.size(); ++i) host[i] = make_float4(1.0f, 2.0f, 3.0f, __builtin_bit_cast(float, (uint32_t)(i * 2654435761u)));
    float4* table; float* out;
    hipMalloc(&table, host.size() * sizeof(float4));
    hipMemcpy(table, host.data(), host.size() * sizeof(float4), hipMemcpyHostToDevice);
    const int blocks = 256 * 5, iters = 2000;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("table: %zu records = %.1f KiB\n", records, records * 64 / 1024.0);
    const char* names[] = {"A lane-private 4x dwordx4 (64 B)", "B quad-cooperative + shuffles", "C lane-private 2x dwordx4 (32 B)", "D as A, half the lanes", "E quad-cooperative via LDS", "F lane-private 8x dwordx4 (128 B)", "G lane-private 6x dwordx4 of 128 B"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 7; ++mode)
        {
            hipEventRecord(e0);
            switch (mode)
            {
            case 0: hipLaunchKernelGGL(gather<0>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records - 1), iters, out); break;
            case 1: hipLaunchKernelGGL(gather<1>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records - 1), iters, out); break;
            case 2: hipLaunchKernelGGL(gather<2>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(2 * records - 1), iters, out); break;
            case 3: hipLaunchKernelGGL(gather<3>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records - 1), iters, out); break;
            case 4: hipLaunchKernelGGL(gather<4>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records - 1), iters, out); break;
            case 5: hipLaunchKernelGGL(gather<5>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records / 2 - 1), iters, out); break;
            case 6: hipLaunchKernelGGL(gather<6>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(records / 2 - 1), iters, out); break;
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gathers = (double)blocks * 256 * iters * (mode == 3 ? 0.5 : 1.0);
            if (rep) printf("%-40s %8.3f ms  %7.1f G records/s  %7.1f GB/s\n", names[mode], ms, gathers / ms * 1e-6, gathers * (mode == 2 ? 32 : (mode == 5 ? 128 : (mode == 6 ? 96 : 64))) / ms * 1e-6);
        }
    return 0;
}
