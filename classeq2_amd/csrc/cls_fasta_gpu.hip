// FASTA input stage on the device: the record semantics of
// FileOrStdin::sequence_content_by_channel (core/src/domain/dtos/file_or_stdin.rs:76-116) and
// SequenceBody::remove_non_iupac_from_sequence (core/src/domain/dtos/sequence.rs:47-56), as data-parallel passes
// over the file text in HBM.  The filtered bases and their offsets stay on the device, where
// cls_place_batch_device() takes them: the reads never return to the host.
//
// The reference is a line-by-line state machine {header, sequence}.  Restated per header line j (h_j = the line
// with every '>' removed, seq_j = the kept bases up to the next header line, seq_-1 = those before the first):
//   * at header line j: h_{j-1} non-empty -> emit (h_{j-1}, seq_{j-1}), even if seq is empty (:92-95);
//                       h_{j-1} empty and seq_{j-1} non-empty -> "unexpected sequence without header": stop (:96-100)
//   * at end of file: emit (h_last, seq_last) iff both are non-empty (:111-113)
//   * a line that is not valid UTF-8 stops everything when it is reached (BufRead::lines -> `line?`), the pending
//     record is lost; empty lines (after "\n" / "\r\n" stripping) are skipped, which changes nothing
// so every per-byte decision needs only the first character of the byte's own line, and the stop conditions are
// minima over header lines.  Bytes of records that are not emitted always sit at the tail of the two output
// streams, so the streams need no second compaction.
//
// Passes (CH = 4096 bytes per workgroup, 16 per thread):
//   1. last newline inside every chunk                      -> exclusive prefix maximum = newline before the chunk
//   2. count per chunk: kept bases, kept header bytes, header lines; first byte that breaks UTF-8 (atomicMin)
//   3. exclusive sums of the three counts                   (hipCUB)           [host reads the three totals]
//   4. the same classification again, scattering bases / header bytes / per-header-line offsets
//   5. per header line: stop conditions, emitted flag -> exclusive sum -> record offset tables
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "cls_place.h"

extern "C" void cls_internal_set_error(const char* msg);  // cls_api.cpp: the thread-local text behind cls_last_error()

namespace {

constexpr int FA_THREADS = 256, FA_PER_THREAD = 16, FA_CHUNK = FA_THREADS * FA_PER_THREAD;
constexpr unsigned long long NO_POS = ~0ull;

int fa_fail(int code, const std::string& m) { cls_internal_set_error(m.c_str()); return code; }

struct MaxOp { __host__ __device__ long long operator()(long long a, long long b) const { return a > b ? a : b; } };

__device__ __forceinline__ uint8_t byte_at(const uint8_t* __restrict__ t, uint64_t len, uint64_t i) { return i < len ? t[i] : (uint8_t)'\n'; }

// pass 1: position of the last '\n' inside every chunk (-1: none)
__global__ __launch_bounds__(FA_THREADS) void fa_last_newline(const uint8_t* __restrict__ text, uint64_t len, long long* __restrict__ chunk_last_nl) {
    __shared__ long long best;
    if (threadIdx.x == 0) best = -1;
    __syncthreads();
    const uint64_t base = (uint64_t)blockIdx.x * FA_CHUNK + (uint64_t)threadIdx.x * FA_PER_THREAD;
    long long mine = -1;
    for (int k = 0; k < FA_PER_THREAD; ++k)
        if (base + k < len && text[base + k] == '\n') mine = (long long)(base + k);
    if (mine >= 0) atomicMax(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0) chunk_last_nl[blockIdx.x] = best;
}

// Is byte i part of a well-formed UTF-8 sequence?  Decided from the bytes around it (a sequence is at most 4 bytes
// and cannot cross a '\n', which is no continuation byte); the end of the text truncates like a line end.
__device__ __forceinline__ bool utf8_ok_at(const uint8_t* __restrict__ t, uint64_t len, uint64_t i) {
    const uint8_t b = t[i];
    if (b < 0x80) return true;
    auto cont = [&](uint64_t j) { return j < len && (t[j] & 0xC0) == 0x80; };
    if ((b & 0xC0) == 0x80) {  // continuation byte: a lead byte within the 3 bytes before must claim it
        for (uint64_t k = 1; k <= 3 && k <= i; ++k) {
            const uint8_t p = t[i - k];
            if ((p & 0xC0) == 0x80) continue;
            const uint32_t need = p >= 0xF0 && p <= 0xF4 ? 3 : p >= 0xE0 && p <= 0xEF ? 2 : p >= 0xC2 && p <= 0xDF ? 1 : 0;
            return need >= k;
        }
        return false;
    }
    if (b >= 0xC2 && b <= 0xDF) return cont(i + 1);
    if (b >= 0xE0 && b <= 0xEF) {
        if (!cont(i + 1) || !cont(i + 2)) return false;
        const uint8_t s = t[i + 1];
        return !(b == 0xE0 && s < 0xA0) && !(b == 0xED && s > 0x9F);  // overlong / surrogates
    }
    if (b >= 0xF0 && b <= 0xF4) {
        if (!cont(i + 1) || !cont(i + 2) || !cont(i + 3)) return false;
        const uint8_t s = t[i + 1];
        return !(b == 0xF0 && s < 0x90) && !(b == 0xF4 && s > 0x8F);  // overlong / beyond U+10FFFF
    }
    return false;  // 0xC0, 0xC1, 0xF5..0xFF
}

struct FaOut {
    uint8_t* bases;
    uint8_t* headers;
    uint64_t* line_pos;       // per header line: its position in the text,
    uint64_t* line_base_off;  // kept bases before it,
    uint64_t* line_hdr_off;   // kept header bytes before it
};

// passes 2 and 4: classify every byte of the chunk; count (WRITE = false) or scatter (WRITE = true)
template <bool WRITE>
__global__ __launch_bounds__(FA_THREADS) void fa_pass(const uint8_t* __restrict__ text, uint64_t len, const long long* __restrict__ prev_nl,
                                                      uint64_t* __restrict__ cnt_bases, uint64_t* __restrict__ cnt_hdr, uint64_t* __restrict__ cnt_lines,
                                                      unsigned long long* __restrict__ first_bad, FaOut out) {
    __shared__ long long s_nl[FA_THREADS];
    __shared__ uint32_t s_cnt[3][FA_THREADS];
    const uint32_t tid = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * FA_CHUNK + (uint64_t)tid * FA_PER_THREAD;
    uint8_t b[FA_PER_THREAD + 1];
    for (int k = 0; k <= FA_PER_THREAD; ++k) b[k] = byte_at(text, len, base + k);
    // last newline at or before the end of each thread's bytes: inclusive maximum over the threads before it
    long long mine = -1;
    for (int k = 0; k < FA_PER_THREAD; ++k) if (base + k < len && b[k] == '\n') mine = (long long)(base + k);
    s_nl[tid] = mine;
    __syncthreads();
    for (int o = 1; o < FA_THREADS; o <<= 1) {
        const long long v = tid >= (uint32_t)o ? s_nl[tid - o] : -1;
        __syncthreads();
        if (v > s_nl[tid]) s_nl[tid] = v;
        __syncthreads();
    }
    long long last_nl = tid ? s_nl[tid - 1] : -1;          // last newline before this thread's first byte, inside the chunk
    if (last_nl < 0) last_nl = prev_nl[blockIdx.x];        // ... or before the chunk (-1: none at all)
    bool is_hdr = byte_at(text, len, (uint64_t)(last_nl + 1)) == '>' && (uint64_t)(last_nl + 1) < len;
    uint32_t flags_base = 0, flags_hdr = 0, flags_line = 0;
    bool bad = false;
    uint64_t bad_pos = 0;
    for (int k = 0; k < FA_PER_THREAD; ++k) {
        const uint64_t i = base + k;
        if (i >= len) break;
        const uint8_t c = b[k];
        if ((uint64_t)(last_nl + 1) == i) {  // a line starts here
            is_hdr = c == '>';
            if (is_hdr) flags_line |= 1u << k;
        }
        if (c >= 0x80 && !bad && !utf8_ok_at(text, len, i)) { bad = true; bad_pos = i; }
        if (is_hdr) {
            // the header is the line without its terminator ("\n" or "\r\n") and without any '>' (:102)
            if (c != '>' && c != '\n' && !(c == '\r' && i + 1 < len && b[k + 1] == '\n')) flags_hdr |= 1u << k;
        } else {
            const uint8_t u = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
            if (u == 'A' || u == 'C' || u == 'G' || u == 'T') flags_base |= 1u << k;
        }
        if (c == '\n') last_nl = (long long)i;
    }
    if (bad && !WRITE) atomicMin(first_bad, (unsigned long long)bad_pos);
    s_cnt[0][tid] = __popc(flags_base);
    s_cnt[1][tid] = __popc(flags_hdr);
    s_cnt[2][tid] = __popc(flags_line);
    __syncthreads();
    for (int o = 1; o < FA_THREADS; o <<= 1) {  // inclusive sums over the threads of the chunk
        uint32_t v[3];
        for (int a = 0; a < 3; ++a) v[a] = tid >= (uint32_t)o ? s_cnt[a][tid - o] : 0u;
        __syncthreads();
        for (int a = 0; a < 3; ++a) s_cnt[a][tid] += v[a];
        __syncthreads();
    }
    if (!WRITE) {
        if (tid == FA_THREADS - 1) { cnt_bases[blockIdx.x] = s_cnt[0][tid]; cnt_hdr[blockIdx.x] = s_cnt[1][tid]; cnt_lines[blockIdx.x] = s_cnt[2][tid]; }
        return;
    }
    uint64_t ob = cnt_bases[blockIdx.x] + (tid ? s_cnt[0][tid - 1] : 0u);   // (the arrays now hold the exclusive sums over chunks)
    uint64_t oh = cnt_hdr[blockIdx.x] + (tid ? s_cnt[1][tid - 1] : 0u);
    uint64_t ol = cnt_lines[blockIdx.x] + (tid ? s_cnt[2][tid - 1] : 0u);
    for (int k = 0; k < FA_PER_THREAD; ++k) {
        const uint8_t c = b[k];
        if ((flags_line >> k) & 1u) { out.line_pos[ol] = base + k; out.line_base_off[ol] = ob; out.line_hdr_off[ol] = oh; ++ol; }
        if ((flags_base >> k) & 1u) out.bases[ob++] = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
        if ((flags_hdr >> k) & 1u) out.headers[oh++] = c;
    }
}

// pass 5a: where does the reference stop?  -> stop[0] = number of header lines it processes, stop[1] = truncated
__global__ void fa_stop(const uint8_t* __restrict__ text, uint64_t len, uint64_t n_lines, const uint64_t* __restrict__ line_pos,
                        const uint64_t* __restrict__ line_base_off, const uint64_t* __restrict__ line_hdr_off, uint64_t total_bases,
                        uint64_t total_hdr, unsigned long long first_bad, unsigned long long* __restrict__ stop) {
    // "unexpected sequence without header" at header line j: the header before it is empty, bases in between
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_lines) {
        const bool prev_empty = j == 0 ? true : line_hdr_off[j] == line_hdr_off[j - 1];
        const uint64_t seq = j == 0 ? line_base_off[0] : line_base_off[j] - line_base_off[j - 1];
        if (prev_empty && seq > 0) atomicMin(&stop[0], (unsigned long long)j);
    }
    if (j == 0 && first_bad != NO_POS) {
        // header lines that start before the line holding the first invalid byte
        uint64_t ls = first_bad;
        while (ls > 0 && text[ls - 1] != '\n') --ls;
        uint64_t lo = 0, hi = n_lines;  // first header line with line_pos >= ls
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (line_pos[mid] < ls) lo = mid + 1; else hi = mid; }
        atomicMin(&stop[0], (unsigned long long)lo);
        stop[1] = 1;
    }
    (void)len; (void)total_bases; (void)total_hdr;
}

// pass 5b: emitted flag per header line (record i is emitted when header line i+1 is processed, or at the end of file)
__global__ void fa_emit_flags(uint64_t n_lines, const uint64_t* __restrict__ line_base_off, const uint64_t* __restrict__ line_hdr_off,
                              uint64_t total_bases, uint64_t total_hdr, const unsigned long long* __restrict__ stop, uint64_t* __restrict__ flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const uint64_t n_proc = stop[0] < n_lines ? stop[0] : n_lines;       // header lines processed
    const bool stopped = stop[0] <= n_lines && (stop[0] < n_lines || stop[1] != 0);
    const uint64_t hlen = (i + 1 < n_lines ? line_hdr_off[i + 1] : total_hdr) - line_hdr_off[i];
    bool emit = false;
    if (i + 1 < n_proc) emit = hlen > 0;                                   // emitted at header line i + 1
    else if (i + 1 == n_lines && !stopped && n_proc == n_lines) emit = hlen > 0 && total_bases - line_base_off[i] > 0;  // at end of file
    flag[i] = emit ? 1 : 0;
}

// pass 5c: offset tables of the emitted records
__global__ void fa_records(uint64_t n_lines, const uint64_t* __restrict__ flag, const uint64_t* __restrict__ rank,
                           const uint64_t* __restrict__ line_base_off, const uint64_t* __restrict__ line_hdr_off, uint64_t total_bases,
                           uint64_t total_hdr, uint64_t* __restrict__ base_off, uint64_t* __restrict__ header_off) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines || !flag[i]) return;
    const uint64_t r = rank[i];
    base_off[r] = line_base_off[i];
    header_off[r] = line_hdr_off[i];
    // the end of record r = the start of the next header line (everything up to the next emitted record is empty)
    base_off[r + 1] = i + 1 < n_lines ? line_base_off[i + 1] : total_bases;
    header_off[r + 1] = i + 1 < n_lines ? line_hdr_off[i + 1] : total_hdr;
}

#define FA_HIP(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) { cleanup(); return fa_fail(e_ == hipErrorOutOfMemory ? CLS_E_NOMEM : CLS_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)

}  // namespace

extern "C" void cls_fasta_dev_free(cls_fasta_dev* f) {
    if (!f) return;
    if (f->d_headers) (void)hipFree(f->d_headers);
    if (f->d_header_off) (void)hipFree(f->d_header_off);
    if (f->d_bases) (void)hipFree(f->d_bases);
    if (f->d_base_off) (void)hipFree(f->d_base_off);
    memset(f, 0, sizeof *f);
}

extern "C" int cls_fasta_scan_device(const void* d_text, uint64_t len, cls_fasta_dev* out, void* hip_stream) {
    if (!out || (!d_text && len)) return fa_fail(CLS_E_INVALID_ARG, "cls_fasta_scan_device: null argument");
    memset(out, 0, sizeof *out);
    hipStream_t stream = (hipStream_t)hip_stream;
    const uint8_t* text = (const uint8_t*)d_text;
    const uint64_t n_chunks = (len + FA_CHUNK - 1) / FA_CHUNK;
    long long *d_last = nullptr, *d_prev = nullptr;
    uint64_t *d_cnt = nullptr, *d_line = nullptr, *d_flag = nullptr, *d_rank = nullptr;
    unsigned long long* d_misc = nullptr;  // [0] first invalid byte, [1..2] stop
    void* d_tmp = nullptr;
    bool ok = false;
    auto cleanup = [&]() {
        for (void* p : {(void*)d_last, (void*)d_prev, (void*)d_cnt, (void*)d_line, (void*)d_flag, (void*)d_rank, (void*)d_misc, d_tmp}) if (p) (void)hipFree(p);
        d_last = d_prev = nullptr; d_cnt = d_line = d_flag = d_rank = nullptr; d_misc = nullptr; d_tmp = nullptr;
        if (!ok) cls_fasta_dev_free(out);
    };
    auto fail_out = [&](int rc) { return rc; };
    try {
        FA_HIP(hipMalloc((void**)&d_misc, 4 * sizeof(unsigned long long)));
        const unsigned long long init[4] = {NO_POS, NO_POS, 0, 0};
        FA_HIP(hipMemcpyAsync(d_misc, init, sizeof init, hipMemcpyHostToDevice, stream));
        uint64_t totals[3] = {0, 0, 0};  // bases, header bytes, header lines
        unsigned long long first_bad = NO_POS;
        if (n_chunks) {
            FA_HIP(hipMalloc((void**)&d_last, n_chunks * 8));
            FA_HIP(hipMalloc((void**)&d_prev, n_chunks * 8));
            FA_HIP(hipMalloc((void**)&d_cnt, 3 * (n_chunks + 1) * 8));
            uint64_t* cnt[3] = {d_cnt, d_cnt + (n_chunks + 1), d_cnt + 2 * (n_chunks + 1)};
            hipLaunchKernelGGL(fa_last_newline, dim3((unsigned)n_chunks), dim3(FA_THREADS), 0, stream, text, len, d_last);
            size_t tmp_bytes = 0, need = 0;
            (void)hipcub::DeviceScan::ExclusiveScan(nullptr, need, d_last, d_prev, MaxOp(), (long long)-1, (int)n_chunks, stream);
            tmp_bytes = need;
            (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need, cnt[0], cnt[0], (int)(n_chunks + 1), stream);
            tmp_bytes = std::max(tmp_bytes, need);
            FA_HIP(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 16));
            FA_HIP(hipcub::DeviceScan::ExclusiveScan(d_tmp, tmp_bytes, d_last, d_prev, MaxOp(), (long long)-1, (int)n_chunks, stream));
            FA_HIP(hipMemsetAsync(d_cnt, 0, 3 * (n_chunks + 1) * 8, stream));
            hipLaunchKernelGGL((fa_pass<false>), dim3((unsigned)n_chunks), dim3(FA_THREADS), 0, stream, text, len, d_prev, cnt[0], cnt[1], cnt[2],
                               d_misc, FaOut{});
            for (int a = 0; a < 3; ++a) FA_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, cnt[a], cnt[a], (int)(n_chunks + 1), stream));
            for (int a = 0; a < 3; ++a) FA_HIP(hipMemcpyAsync(&totals[a], cnt[a] + n_chunks, 8, hipMemcpyDeviceToHost, stream));
            FA_HIP(hipMemcpyAsync(&first_bad, d_misc, 8, hipMemcpyDeviceToHost, stream));
            FA_HIP(hipStreamSynchronize(stream));
            // outputs (sized exactly) + the per-header-line tables
            const uint64_t nb = totals[0], nh = totals[1], nl = totals[2];
            FA_HIP(hipMalloc(&out->d_bases, nb ? nb : 16));
            FA_HIP(hipMalloc(&out->d_headers, nh ? nh : 16));
            FA_HIP(hipMalloc((void**)&d_line, 3 * (nl + 1) * 8));
            FaOut fo{(uint8_t*)out->d_bases, (uint8_t*)out->d_headers, d_line, d_line + (nl + 1), d_line + 2 * (nl + 1)};
            hipLaunchKernelGGL((fa_pass<true>), dim3((unsigned)n_chunks), dim3(FA_THREADS), 0, stream, text, len, d_prev, cnt[0], cnt[1], cnt[2],
                               d_misc, fo);
            uint64_t n_rec = 0;
            unsigned long long stop_h[2] = {NO_POS, 0};
            FA_HIP(hipMalloc((void**)&out->d_base_off, (nl + 2) * 8));
            FA_HIP(hipMalloc((void**)&out->d_header_off, (nl + 2) * 8));
            FA_HIP(hipMemsetAsync(out->d_base_off, 0, 8, stream));
            FA_HIP(hipMemsetAsync(out->d_header_off, 0, 8, stream));
            if (nl) {
                const unsigned blocks = (unsigned)((nl + 255) / 256);
                hipLaunchKernelGGL(fa_stop, dim3(blocks), dim3(256), 0, stream, text, len, nl, fo.line_pos, fo.line_base_off, fo.line_hdr_off, nb, nh,
                                   first_bad, d_misc + 1);
                FA_HIP(hipMalloc((void**)&d_flag, (nl + 1) * 8));
                FA_HIP(hipMalloc((void**)&d_rank, (nl + 1) * 8));
                FA_HIP(hipMemsetAsync(d_flag, 0, (nl + 1) * 8, stream));
                hipLaunchKernelGGL(fa_emit_flags, dim3(blocks), dim3(256), 0, stream, nl, fo.line_base_off, fo.line_hdr_off, nb, nh, d_misc + 1, d_flag);
                size_t need2 = 0;
                (void)hipcub::DeviceScan::ExclusiveSum(nullptr, need2, d_flag, d_rank, (int)(nl + 1), stream);
                if (need2 > tmp_bytes) { (void)hipFree(d_tmp); d_tmp = nullptr; FA_HIP(hipMalloc(&d_tmp, need2)); tmp_bytes = need2; }
                FA_HIP(hipcub::DeviceScan::ExclusiveSum(d_tmp, tmp_bytes, d_flag, d_rank, (int)(nl + 1), stream));
                hipLaunchKernelGGL(fa_records, dim3(blocks), dim3(256), 0, stream, nl, d_flag, d_rank, fo.line_base_off, fo.line_hdr_off, nb, nh,
                                   (uint64_t*)out->d_base_off, (uint64_t*)out->d_header_off);
                FA_HIP(hipMemcpyAsync(&n_rec, d_rank + nl, 8, hipMemcpyDeviceToHost, stream));
                FA_HIP(hipMemcpyAsync(stop_h, d_misc + 1, 16, hipMemcpyDeviceToHost, stream));
                FA_HIP(hipStreamSynchronize(stream));
            } else if (first_bad != NO_POS) {
                stop_h[1] = 1;
            } else if (nb > 0) {
                stop_h[0] = 0;  // bases but no header line at all: nothing is emitted, no error either (the end-of-file test needs a header)
            }
            if (n_rec > 0xFFFFFFFFull) { cleanup(); return fail_out(fa_fail(CLS_E_INVALID_ARG, "cls_fasta_scan_device: more than 2^32 records")); }
            out->n = (uint32_t)n_rec;
            out->truncated = (stop_h[1] != 0 || (nl && stop_h[0] < nl)) ? 1u : 0u;
            if (hipGetLastError() != hipSuccess) { cleanup(); return fail_out(fa_fail(CLS_E_HIP, "cls_fasta_scan_device: kernel launch failed")); }
            // the record tables say how much of the two streams belongs to emitted records
            uint64_t ends[2] = {0, 0};
            FA_HIP(hipMemcpyAsync(&ends[0], (uint64_t*)out->d_base_off + n_rec, 8, hipMemcpyDeviceToHost, stream));
            FA_HIP(hipMemcpyAsync(&ends[1], (uint64_t*)out->d_header_off + n_rec, 8, hipMemcpyDeviceToHost, stream));
            FA_HIP(hipStreamSynchronize(stream));
            out->n_bases = ends[0];
            out->n_header_bytes = ends[1];
        } else {
            FA_HIP(hipMalloc(&out->d_bases, 16));
            FA_HIP(hipMalloc(&out->d_headers, 16));
            FA_HIP(hipMalloc((void**)&out->d_base_off, 16));
            FA_HIP(hipMalloc((void**)&out->d_header_off, 16));
            FA_HIP(hipMemsetAsync(out->d_base_off, 0, 16, stream));
            FA_HIP(hipMemsetAsync(out->d_header_off, 0, 16, stream));
            FA_HIP(hipStreamSynchronize(stream));
        }
        ok = true;
        cleanup();
        return CLS_OK;
    } catch (...) {
        cleanup();
        return fail_out(fa_fail(CLS_E_INTERNAL, "cls_fasta_scan_device: unknown exception"));
    }
}

// Host text in, host records out, through the device passes (what the tests compare with cls_fasta_parse).
extern "C" int cls_fasta_parse_gpu(const char* text, size_t len, int device, cls_fasta* out) {
    if (!out || (!text && len)) return fa_fail(CLS_E_INVALID_ARG, "cls_fasta_parse_gpu: null argument");
    memset(out, 0, sizeof *out);
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fa_fail(CLS_E_NO_DEVICE, "cls_fasta_parse_gpu: no HIP device is visible");
    if (device >= 0 && hipSetDevice(device) != hipSuccess) return fa_fail(CLS_E_INVALID_ARG, "cls_fasta_parse_gpu: bad device ordinal");
    void* d_text = nullptr;
    cls_fasta_dev dv;
    memset(&dv, 0, sizeof dv);
    auto cleanup = [&]() { if (d_text) (void)hipFree(d_text); cls_fasta_dev_free(&dv); };
    FA_HIP(hipMalloc(&d_text, len ? len : 16));
    if (len) FA_HIP(hipMemcpy(d_text, text, len, hipMemcpyHostToDevice));
    int rc = cls_fasta_scan_device(d_text, len, &dv, nullptr);
    if (rc != CLS_OK) { cleanup(); return rc; }
    out->n = dv.n;
    out->truncated = dv.truncated;
    out->headers = (char*)malloc(dv.n_header_bytes + 1);
    out->bases = (char*)malloc(dv.n_bases + 1);
    out->header_off = (uint64_t*)malloc(((size_t)dv.n + 1) * 8);
    out->base_off = (uint64_t*)malloc(((size_t)dv.n + 1) * 8);
    if (!out->headers || !out->bases || !out->header_off || !out->base_off) { cleanup(); cls_fasta_free(out); return fa_fail(CLS_E_NOMEM, "cls_fasta_parse_gpu: out of host memory"); }
    if (dv.n_header_bytes) FA_HIP(hipMemcpy(out->headers, dv.d_headers, dv.n_header_bytes, hipMemcpyDeviceToHost));
    if (dv.n_bases) FA_HIP(hipMemcpy(out->bases, dv.d_bases, dv.n_bases, hipMemcpyDeviceToHost));
    FA_HIP(hipMemcpy(out->header_off, dv.d_header_off, ((size_t)dv.n + 1) * 8, hipMemcpyDeviceToHost));
    FA_HIP(hipMemcpy(out->base_off, dv.d_base_off, ((size_t)dv.n + 1) * 8, hipMemcpyDeviceToHost));
    cleanup();
    return CLS_OK;
}
