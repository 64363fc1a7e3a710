// Device-wide stable LSD radix sort of (u64 key, i32 value) pairs, 8-bit digits, for gfx950.
// This is the general sort behind gsplat.rendering.isect_tiles(sort=True)
// (street_gaussian/models/street_gaussian_renderer.py:243-252; SURVEY.md A.2: stable over bits
// [0, 32 + tile_bits + cam_bits)) and behind the Morton ordering of simple_knn (knn.hip).
// The tile-bucketed path in isect_bin.hip produces the same result with far less traffic;
// this one is the reference-shaped route, its fallback and its on-GPU cross-check.
//
// Per pass: (1) per-workgroup digit histogram, (2) per-digit exclusive scan across workgroups,
// (3) stable scatter.  Stability inside a workgroup comes from wave-ordered ranking: wave w owns
// a contiguous quarter of the workgroup's 4096 keys, walks it in 16 coalesced rounds of 64, and
// ranks the lanes of a round with a ballot-based match-any (wave64), so no per-thread counters
// are needed.  Integer-only; results are bit-exact by construction.
#include "sc_common.h"

namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ROUNDS = 16;
constexpr int RS_PER_WAVE = 64 * RS_ROUNDS;       // 1024
constexpr int RS_TILE = RS_PER_WAVE * RS_WAVES;   // 4096 keys per workgroup

__device__ __forceinline__ unsigned digit_of(unsigned long long k, int shift) {
    return (unsigned)(k >> shift) & 0xffu;
}

// lanes of this wave whose 8-bit digit equals mine, among `valid` lanes
__device__ __forceinline__ unsigned long long match_any8(unsigned d, bool valid) {
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const unsigned long long* __restrict__ keys,
                                                             unsigned n, int shift, unsigned nb,
                                                             unsigned* __restrict__ block_hist) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_TILE / RS_THREADS; ++r) {
        const unsigned i = base + r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(keys[i], shift)], 1u);
    }
    __syncthreads();
    block_hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// one workgroup per digit: exclusive scan of its row of nb block counts; row total to digit_tot.
__global__ __launch_bounds__(256) void rs_scan_rows_kernel(unsigned* __restrict__ block_hist, unsigned nb,
                                                           unsigned* __restrict__ digit_tot) {
    __shared__ unsigned wave_tot[4];
    __shared__ unsigned carry_s;
    unsigned* row = block_hist + (size_t)blockIdx.x * nb;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = sc_lane(), wave = threadIdx.x >> 6;
    for (unsigned base = 0; base < nb; base += 256) {
        const unsigned i = base + threadIdx.x;
        const unsigned v = (i < nb) ? row[i] : 0u;
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)v);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        unsigned pre = carry_s, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned s = wave_tot[w];
            if (w < wave) pre += s;
            tot += s;
        }
        if (i < nb) row[i] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) digit_tot[blockIdx.x] = carry_s;
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const unsigned long long* __restrict__ keys_in, const int* __restrict__ vals_in,
    unsigned long long* __restrict__ keys_out, int* __restrict__ vals_out, unsigned n, int shift,
    unsigned nb, const unsigned* __restrict__ block_hist, const unsigned* __restrict__ digit_tot) {
    __shared__ unsigned h[RS_WAVES][256];   // per-wave digit counts, then running write cursors
    __shared__ unsigned dscan[256];
    __shared__ unsigned wtot[4];
    const int t = threadIdx.x, lane = sc_lane(), wave = t >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) h[w][t] = 0;
    // exclusive scan of the 256 digit totals (every workgroup redoes it: 256 values)
    {
        const unsigned v = digit_tot[t];
        const unsigned incl = (unsigned)sc_wave_incl_scan((int)v);
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        unsigned pre = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) if (w < wave) pre += wtot[w];
        dscan[t] = pre + incl - v;
    }
    __syncthreads();

    const unsigned wbase = blockIdx.x * RS_TILE + wave * RS_PER_WAVE;
    unsigned long long k[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const unsigned i = wbase + r * 64 + lane;
        const bool valid = i < n;
        k[r] = valid ? keys_in[i] : ~0ull;
        if (valid) atomicAdd(&h[wave][digit_of(k[r], shift)], 1u);
    }
    __syncthreads();
    {
        // digit t: global base + this workgroup's prefix + earlier waves' counts
        unsigned run = dscan[t] + block_hist[(size_t)t * nb + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const unsigned c = h[w][t];
            h[w][t] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const unsigned i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const unsigned d = digit_of(k[r], shift);
        const unsigned long long peers = match_any8(d, valid);
        const unsigned rank = (unsigned)__popcll(peers & sc_lanemask_lt());
        unsigned pos = 0;
        if (valid) pos = h[wave][d] + rank;
        // all lanes of the wave have read the cursor before its leader bumps it (same wave, in order)
        if (valid && rank == 0) h[wave][d] += (unsigned)__popcll(peers);
        if (valid) {
            keys_out[pos] = k[r];
            vals_out[pos] = vals_in[i];
        }
    }
}

}  // namespace

static inline unsigned rs_num_blocks(int64_t n) { return (unsigned)((n + RS_TILE - 1) / RS_TILE); }

extern "C" size_t sc_radix_sort_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return sc_align_up(((size_t)rs_num_blocks(n) * 256 + 256) * sizeof(unsigned), 256);
}

extern "C" int sc_radix_sort_pairs_u64_i32(uint64_t* keys, int32_t* vals, uint64_t* tmp_keys,
                                           int32_t* tmp_vals, int64_t n, int end_bit,
                                           void* workspace, size_t ws_bytes, sc_stream_t stream) {
    if (n < 0 || end_bit < 0 || end_bit > 64) return SC_EINVAL;
    if (n <= 1 || end_bit == 0) return SC_OK;
    if (n > 0x7fffffffLL) return SC_EINVAL;
    if (!keys || !vals || !tmp_keys || !tmp_vals || !workspace) return SC_EINVAL;
    if (ws_bytes < sc_radix_sort_workspace_bytes(n)) return SC_EWORKSPACE;
    const unsigned nb = rs_num_blocks(n);
    unsigned* block_hist = (unsigned*)workspace;
    unsigned* digit_tot = block_hist + (size_t)nb * 256;
    unsigned long long* kin = (unsigned long long*)keys;
    unsigned long long* kout = (unsigned long long*)tmp_keys;
    int* vin = vals;
    int* vout = tmp_vals;
    const int passes = (end_bit + 7) / 8;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * 8;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(RS_THREADS), 0, sc_s(stream), kin,
                           (unsigned)n, shift, nb, block_hist);
        SC_LAUNCH_CHECK();
        hipLaunchKernelGGL(rs_scan_rows_kernel, dim3(256), dim3(256), 0, sc_s(stream), block_hist, nb,
                           digit_tot);
        SC_LAUNCH_CHECK();
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(RS_THREADS), 0, sc_s(stream), kin, vin,
                           kout, vout, (unsigned)n, shift, nb, block_hist, digit_tot);
        SC_LAUNCH_CHECK();
        unsigned long long* tk = kin; kin = kout; kout = tk;
        int* tv = vin; vin = vout; vout = tv;
    }
    if (kin != (unsigned long long*)keys) {
        SC_HIP(hipMemcpyAsync(keys, kin, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToDevice, sc_s(stream)));
        SC_HIP(hipMemcpyAsync(vals, vin, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, sc_s(stream)));
    }
    return SC_OK;
}
