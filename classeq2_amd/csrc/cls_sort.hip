// Locality order of the reads (fast path, batches >= 4096 reads): a counting sort on the TOP bits of the 64-bit
// locality key -- histogram, exclusive scan, scatter; three small kernels, gfx950.
//
// The order is a heuristic that changes no record (tests/test_gpu_c3.py), and what it needs of the key is little: on C3 the
// placement kernel takes 5.36 ms with all 34 key bits sorted, 5.37 with the top 18, 5.40 with the top 12, 5.47 with the
// top 8 (CLS_ORDER_SKIP_BITS, round 3).  So the reads are binned by the top ORDER_BIN_BITS bits of their key (leaf
// neighbourhood = median set id of their specific k-mers; the MinHash below it only breaks ties) and keep whatever
// order the scatter gives them inside a bin.  Round 2 called hipcub::DeviceRadixSort here: 168 launches and 0.19 ms a
// step for a full 64-bit key-value sort of something that needs twelve bits.
#include "cls_sort.h"

namespace cls {

namespace {

constexpr int SORT_THREADS = 1024, SORT_PER_THREAD = 16;  // 16384 reads a workgroup: nearly every bin occurs in it, ONE global atomic per bin

__device__ __forceinline__ uint32_t bin_of(uint64_t key, int shift) {
    return key == ~0ull ? ORDER_BINS - 1 : (uint32_t)(key >> shift) & (ORDER_BINS - 1);  // (reads without a key: the last bin)
}

// Per workgroup a histogram in LDS; its flush reserves the workgroup's share of every bin: block_base[b][bin] = how many
// reads of that bin the workgroups that flushed earlier hold (whichever they are: the order inside a bin is free).
__global__ __launch_bounds__(SORT_THREADS) void order_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift, uint32_t* __restrict__ hist,
                                                                 uint32_t* __restrict__ block_base) {
    __shared__ uint32_t h[ORDER_BINS];
    for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += SORT_THREADS) h[i] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SORT_THREADS * SORT_PER_THREAD);
#pragma unroll
    for (int q = 0; q < SORT_PER_THREAD; ++q) {
        const uint32_t r = base + q * SORT_THREADS + threadIdx.x;
        if (r < n) atomicAdd(&h[bin_of(keys[r], shift)], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += SORT_THREADS)
        block_base[(size_t)blockIdx.x * ORDER_BINS + i] = h[i] ? atomicAdd(&hist[i], h[i]) : 0u;
}

// exclusive scan of the ORDER_BINS counts, in place: hist[i] becomes where bin i starts (one workgroup)
__global__ __launch_bounds__(1024) void order_scan_kernel(uint32_t* __restrict__ hist) {
    constexpr uint32_t PER = ORDER_BINS / 1024;
    __shared__ uint32_t wave_tot[16];
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) { v[q] = hist[threadIdx.x * PER + q]; sum += v[q]; }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = sum;  // inclusive scan over the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    uint32_t at = before + inc - sum;
#pragma unroll
    for (uint32_t q = 0; q < PER; ++q) { hist[threadIdx.x * PER + q] = at; at += v[q]; }
}

// read r goes to: start of its bin + its workgroup's share + its rank inside the workgroup (an LDS atomic)
__global__ __launch_bounds__(SORT_THREADS) void order_scatter_kernel(const uint64_t* __restrict__ keys, uint32_t n, int shift, const uint32_t* __restrict__ start,
                                                                    const uint32_t* __restrict__ block_base, uint32_t* __restrict__ idx_out) {
    __shared__ uint32_t at[ORDER_BINS];
    for (uint32_t i = threadIdx.x; i < ORDER_BINS; i += SORT_THREADS) at[i] = start[i] + block_base[(size_t)blockIdx.x * ORDER_BINS + i];
    __syncthreads();
    const uint32_t base = blockIdx.x * (SORT_THREADS * SORT_PER_THREAD);
#pragma unroll
    for (int q = 0; q < SORT_PER_THREAD; ++q) {
        const uint32_t r = base + q * SORT_THREADS + threadIdx.x;
        if (r < n) idx_out[atomicAdd(&at[bin_of(keys[r], shift)], 1u)] = r;
    }
}

}  // namespace

static uint32_t order_blocks(uint32_t n) { return (n + SORT_THREADS * SORT_PER_THREAD - 1) / (SORT_THREADS * SORT_PER_THREAD); }

size_t order_temp_bytes(uint32_t n) { return (1 + (size_t)order_blocks(n)) * ORDER_BINS * sizeof(uint32_t); }

hipError_t order_reads(void* tmp, const uint64_t* keys, uint32_t* idx_out, uint32_t n, int key_bits, hipStream_t stream) {
    uint32_t* hist = static_cast<uint32_t*>(tmp);
    uint32_t* block_base = hist + ORDER_BINS;
    const int shift = key_bits > ORDER_BIN_BITS ? key_bits - ORDER_BIN_BITS : 0;
    hipError_t e = hipMemsetAsync(hist, 0, ORDER_BINS * sizeof(uint32_t), stream);
    if (e != hipSuccess) return e;
    const uint32_t blocks = order_blocks(n);
    hipLaunchKernelGGL(order_hist_kernel, dim3(blocks), dim3(SORT_THREADS), 0, stream, keys, n, shift, hist, block_base);
    hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, stream, hist);
    hipLaunchKernelGGL(order_scatter_kernel, dim3(blocks), dim3(SORT_THREADS), 0, stream, keys, n, shift, (const uint32_t*)hist, (const uint32_t*)block_base, idx_out);
    return hipGetLastError();
}

}  // namespace cls
