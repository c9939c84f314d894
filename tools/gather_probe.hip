// Random-gather reference for the roofline of the placement kernel: what this chip
// delivers on random W-byte reads from a table of T bytes at the placement kernel's occupancy (5 workgroups of 256
// threads per CU), (a) with every lane's reads independent (throughput) and (b) as one dependent chain per lane
// (latency-bound, the shape of the descent: the next address comes out of the previous read).
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe > gather_probe.json
// One JSON object on stdout; rates count REQUESTED bytes (W per read) and 64-byte lines (one per read).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int W> struct Word;
template <> struct Word<4> { using T = uint32_t; };
template <> struct Word<8> { using T = uint2; };
template <> struct Word<16> { using T = uint4; };
__device__ __forceinline__ uint32_t fold(uint32_t v) { return v; }
__device__ __forceinline__ uint32_t fold(uint2 v) { return v.x ^ v.y; }
__device__ __forceinline__ uint32_t fold(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// ILP independent reads in flight per lane and round; DEP: the next index depends on the value just read
template <int W, int ILP, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ table, uint32_t n_elems_mask, uint32_t rounds, uint32_t* __restrict__ sink) {
    using T = typename Word<W>::T;
    const T* t = reinterpret_cast<const T*>(table);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t idx[ILP], acc = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) idx[i] = mix(gid * ILP + i + 1);
    for (uint32_t r = 0; r < rounds; ++r) {
        T v[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) v[i] = t[idx[i] & n_elems_mask];
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            const uint32_t f = fold(v[i]);
            acc ^= f;
            idx[i] = mix(idx[i] + (DEP ? f : 0u) + 0x9e3779b9u);
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;  // keeps the loads alive
}

template <int W, int ILP, bool DEP>
static void run(const uint8_t* d_table, uint64_t table_bytes, int n_cu, int blocks_per_cu, uint32_t rounds, uint32_t* d_sink, bool first) {
    const uint32_t n_elems = (uint32_t)(table_bytes / W);  // power of two by construction
    const dim3 grid(n_cu * blocks_per_cu), block(256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather<W, ILP, DEP>), grid, block, 0, 0, d_table, n_elems - 1, rounds / 4 + 1, d_sink);  // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((gather<W, ILP, DEP>), grid, block, 0, 0, d_table, n_elems - 1, rounds, d_sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double reads = (double)grid.x * 256.0 * ILP * rounds;
    printf("%s  {\"table_mib\": %.0f, \"width\": %d, \"in_flight_per_lane\": %d, \"dependent\": %s, \"blocks_per_cu\": %d, \"ms\": %.3f, "
           "\"greads_per_s\": %.2f, \"requested_gb_per_s\": %.1f, \"line64_gb_per_s\": %.1f}",
           first ? "" : ",\n", table_bytes / 1048576.0, W, ILP, DEP ? "true" : "false", blocks_per_cu, best, reads / best / 1e6,
           reads * W / best / 1e6, reads * 64 / best / 1e6);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint64_t max_bytes = 1ull << 30;  // 1 GiB: beyond the 256 MiB Infinity Cache, like the 691 MB index of C3
    uint8_t* d_table;
    uint32_t* d_sink;
    CK(hipMalloc(&d_table, max_bytes));
    CK(hipMalloc(&d_sink, 64));
    {   // random contents (the dependent chains must not fall into short cycles)
        std::vector<uint32_t> h(max_bytes / 4);
        uint64_t s = 0x9e3779b97f4a7c15ull;
        for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
        CK(hipMemcpy(d_table, h.data(), max_bytes, hipMemcpyHostToDevice));
    }
    printf("{\"device\": \"%s\", \"cus\": %d, \"what\": \"random W-byte reads from a T-MiB table, 256-thread workgroups, rates over requested bytes and over 64-byte lines\",\n \"runs\": [\n",
           prop.gcnArchName, n_cu);
    bool first = true;
    const uint64_t sizes[] = {1ull << 30, 64ull << 20, 8ull << 20};  // HBM-resident / Infinity-Cache-resident / close to L2-resident (4 MiB per XCD)
    for (uint64_t sz : sizes) {
        run<16, 4, false>(d_table, sz, n_cu, 5, 256, d_sink, first); first = false;
        run<16, 1, false>(d_table, sz, n_cu, 5, 512, d_sink, false);
        run<16, 1, true>(d_table, sz, n_cu, 5, 256, d_sink, false);
        run<8, 4, false>(d_table, sz, n_cu, 5, 256, d_sink, false);
        run<8, 1, true>(d_table, sz, n_cu, 5, 256, d_sink, false);
        run<4, 4, false>(d_table, sz, n_cu, 5, 256, d_sink, false);
        run<4, 1, true>(d_table, sz, n_cu, 5, 256, d_sink, false);
        run<16, 4, false>(d_table, sz, n_cu, 8, 256, d_sink, false);
        run<16, 1, true>(d_table, sz, n_cu, 8, 256, d_sink, false);
    }
    printf("\n ]}\n");
    CK(hipFree(d_table)); CK(hipFree(d_sink));
    return 0;
}
