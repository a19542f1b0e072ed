// Permutohedral lattice + DenseCRF mean-field kernels for gfx950.
//
// Reference semantics (third-party/densecrf/src):
//   Permutohedral::init  SSE branch   permutohedral.cpp:140-321   (elevate, round-half-even,
//                                     rank, barycentric, d+1 vertex keys per point, blur neighbours)
//   sseCompute / seqCompute           permutohedral.cpp:476-589   (splat, blur over d+1 axes, slice)
//   DenseKernel::initLattice/filter   pairwise.cpp:40-80          (symmetric normalisation)
//   PottsCompatibility::apply         labelcompatibility.cpp:46-48
//   expAndNormalize, inference        densecrf.cpp:98-131
//
// MI355X design notes
//   * One open-addressing hash table in HBM holds the lattice vertices of ALL frames of a chunk
//     (the frame index is an extra key coordinate), so every later kernel is a flat launch over
//     global vertex / point indices with no per-frame loop.
//   * Vertex numbering is whatever the atomics produce; no result depends on it.
//   * The reference splats sequentially over points, so a vertex's fp32 sum is ordered by point
//     index.  To stay bit-exact the splat is a GATHER: entries are stably sorted by vertex and each
//     (vertex, class) chain adds its contributions in ascending point order.  No float atomics.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "device_math.h"
#include "rvseg_crf.h"

namespace rvseg {

// ---------------------------------------------------------------------------------------------
// hash table
// ---------------------------------------------------------------------------------------------
constexpr int ST_EMPTY = -1, ST_LOCKED = -2, ST_FILLED = 0;

// d coordinates, then the frame index at k[7]; unused = 0.  A union so that the short / word /
// quad-word views alias legally (plain reinterpret_casts let the compiler drop the short stores).
struct Key8 {
    union {
        short k[8];
        unsigned w[4];
        unsigned long long q[2];
    };
};

__device__ __forceinline__ unsigned hash_key(const Key8& key) {
    unsigned h = 0x9E3779B9u;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        h ^= key.w[i];
        h *= 0x85EBCA6Bu;
        h ^= h >> 15;
    }
    return h;
}

__device__ __forceinline__ bool key_equal_at(const unsigned long long* tkeys, unsigned slot, const Key8& key) {
    const unsigned long long* mine = key.q;
    // agent-scope loads: another CU may have written the key after this CU cached the line
    const unsigned long long a = __hip_atomic_load(tkeys + 2 * (size_t)slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(tkeys + 2 * (size_t)slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a == mine[0] && b == mine[1];
}

// The table is partitioned by frame: frame f owns slots [f*cap_f, (f+1)*cap_f).  Slot order is
// therefore frame-major, and the scan-based compaction below numbers a frame's vertices
// contiguously (the fused update kernel stages one frame's vertex values in LDS).
// find-or-create; returns the global slot.  On overflow sets counters[1] and returns the region base.
__device__ __forceinline__ unsigned hash_insert(int* state, unsigned long long* tkeys, unsigned base, unsigned mask,
                                                int* counters, const Key8& key) {
    unsigned hl = hash_key(key) & mask;
    for (unsigned probes = 0; probes <= mask; ) {
        const unsigned h = base + hl;
        int st = __hip_atomic_load(state + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == ST_EMPTY) {
            int expected = ST_EMPTY;
            if (__hip_atomic_compare_exchange_strong(state + h, &expected, ST_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                const unsigned long long* mine = key.q;
                __hip_atomic_store(tkeys + 2 * (size_t)h, mine[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(tkeys + 2 * (size_t)h + 1, mine[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(state + h, ST_FILLED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                return h;
            }
            continue;  // lost the race: look at the same slot again
        }
        if (st == ST_LOCKED) continue;  // the owner publishes within its own loop iteration
        if (key_equal_at(tkeys, h, key)) return h;
        hl = (hl + 1) & mask;
        probes++;
    }
    counters[1] = 1;
    return base;
}

// read-only lookup (table complete, written by earlier kernels)
__device__ __forceinline__ int hash_lookup(const int* state, const unsigned long long* tkeys, unsigned base, unsigned mask,
                                           const Key8& key) {
    unsigned hl = hash_key(key) & mask;
    const unsigned long long* mine = key.q;
    for (unsigned probes = 0; probes <= mask; probes++) {
        const unsigned h = base + hl;
        if (state[h] == ST_EMPTY) return -1;
        if (tkeys[2 * (size_t)h] == mine[0] && tkeys[2 * (size_t)h + 1] == mine[1]) return (int)h;
        hl = (hl + 1) & mask;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------
// Permutohedral::init per point (SSE branch semantics, fp32, no contraction)
// ---------------------------------------------------------------------------------------------
constexpr int LP_SET = 512;        // block-local vertex set (LP_CHUNKS x 256 points x (d+1) keys, few distinct)
constexpr int LP_MAX_PROBE = 24;
constexpr int LP_CHUNKS = 8;       // consecutive 256-point chunks per block (at most): the set (and its global slots) carries over

template <int D>
__global__ void __launch_bounds__(256)
lattice_points_kernel(LatticeDev L, FeatureSource fs, int n_chunks) {
    // block-local vertex set: tag (EMPTY / LOCKED / FILLED), key, global slot; `lnew` lists the set
    // entries in creation order, so each chunk resolves only the entries it added
    __shared__ int ltag[LP_SET];
    __shared__ unsigned long long lkey[LP_SET][2];
    __shared__ unsigned lslot[LP_SET];
    __shared__ unsigned short lnew[LP_SET];
    __shared__ unsigned n_new;
    for (int t = threadIdx.x; t < LP_SET; t += 256) ltag[t] = ST_EMPTY;
    if (threadIdx.x == 0) n_new = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned resolved = 0;   // set entries [0, resolved) of lnew already have their global slot
    const long long per_frame = L.Npad;

  for (int chunk = 0; chunk < n_chunks; chunk++) {
    const long long gid0 = ((long long)blockIdx.x * n_chunks + chunk) * 256 + threadIdx.x;
    const bool active = gid0 < per_frame * L.n_frames;   // all threads stay for the barriers
    const long long gid = active ? gid0 : 0;
    const int frame = (int)(gid / per_frame);
    const int i = (int)(gid - (long long)frame * per_frame);
    const bool real = active && i < L.N;
    const long long gp = (long long)frame * L.N + i;  // global point index (valid when real)

    float f[D];
    if (!real) {
#pragma unroll
        for (int k = 0; k < D; k++) f[k] = 0.0f;  // padded lanes carry zero features (permutohedral.cpp:196)
    } else if (fs.mode == 0) {
#pragma unroll
        for (int k = 0; k < D; k++) f[k] = fs.feat[gp * D + k];
    } else {
        // frame mode (D == 6): segmenter.cpp:629-637 on the frame's own points
        const float4 c = fs.cloud[gp];
        float x = c.x, y = c.y, z = c.z;
        if (!(finite_f(x) && finite_f(y) && finite_f(z))) x = y = z = 0.0f;
        const uint8_t* px = fs.rgb + gp * 3;
        const float v[6] = {x * fs.xyz_kernel, y * fs.xyz_kernel, z * fs.xyz_kernel,
                            ((float)px[0] / 255.0f) * fs.rgb_kernel, ((float)px[1] / 255.0f) * fs.rgb_kernel,
                            ((float)px[2] / 255.0f) * fs.rgb_kernel};
#pragma unroll
        for (int k = 0; k < D; k++) f[k] = v[k < 6 ? k : 0];
    }

    const float invdplus1 = 1.0f / (D + 1), dplus1 = (float)(D + 1);
    float el[D + 1], rem0[D + 1], rank[D + 1];
    // elevate (permutohedral.cpp:201-207)
    float sm = 0.0f;
#pragma unroll
    for (int j = D; j > 0; j--) {
        const float cf = f[j - 1] * L.scale[j - 1];
        el[j] = sm - (float)j * cf;
        sm += cf;
    }
    el[0] = sm;
    // closest 0-coloured simplex (:210-220), cvtps_epi32 = round half to even
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k <= D; k++) {
        float v = invdplus1 * el[k];
        v = rintf(v);
        rem0[k] = v * dplus1;
        sum += v;
    }
    // rank (:223-233)
#pragma unroll
    for (int k = 0; k <= D; k++) rank[k] = 0.0f;
#pragma unroll
    for (int a = 0; a < D; a++) {
        const float di = el[a] - rem0[a];
#pragma unroll
        for (int b = a + 1; b <= D; b++) {
            const float dj = el[b] - rem0[b];
            const float c = di < dj ? 1.0f : 0.0f;
            rank[a] += c;
            rank[b] += 1.0f - c;
        }
    }
    // back onto the plane (:236-242)
#pragma unroll
    for (int k = 0; k <= D; k++) {
        rank[k] += sum;
        const float add = rank[k] < 0.0f ? dplus1 : 0.0f;
        const float sub = rank[k] >= dplus1 ? dplus1 : 0.0f;
        rank[k] += add - sub;
        rem0[k] += add - sub;
    }
    // barycentric (:245-263); the scatter index D - rank is data dependent, so walk it with
    // compile-time indices to keep the array in registers
    float bary[D + 2];
#pragma unroll
    for (int k = 0; k < D + 2; k++) bary[k] = 0.0f;
#pragma unroll
    for (int k = 0; k <= D; k++) {
        const float v = (el[k] - rem0[k]) * invdplus1;
        const int p = D - (int)rank[k];
#pragma unroll
        for (int q = 0; q <= D; q++) {
            if (q == p) { bary[q] += v; bary[q + 1] -= v; }
        }
    }
    bary[0] += 1.0f + bary[D + 1];
    // vertices (:266-275).  Neighbouring points share almost all their vertices, so the block first
    // collects its distinct keys in an LDS set (phase 1), then one lane per distinct key does the
    // global find-or-create -- all global round trips of a block overlap (phase 2) -- and finally
    // every point reads the global slots of its d+1 vertices back from LDS (phase 3).
    int lidx[D + 1];   // index into the LDS set, or -1 - (global slot) when the set was too full
#pragma unroll
    for (int r = 0; r <= D; r++) {
        Key8 key;
#pragma unroll
        for (int k = 0; k < 8; k++) key.k[k] = 0;
#pragma unroll
        for (int k = 0; k < D; k++) {
            const int rk = (int)rank[k];
            const int canon = rk <= D - r ? r : r - (D + 1);
            key.k[k] = (short)(rem0[k] + (float)canon);
        }
        key.k[7] = (short)frame;
        // equal keys inside the wave first (neighbouring points share most vertices): one lane per
        // distinct key goes to the LDS set, the others take its answer
        int leader_of = lane;
        {
            bool pending = active;
            for (;;) {
                const unsigned long long todo = __ballot(pending);
                if (!todo) break;
                const int ld = __ffsll((long long)todo) - 1;
                const unsigned a0 = __builtin_amdgcn_readlane(key.w[0], ld), a1 = __builtin_amdgcn_readlane(key.w[1], ld);
                const unsigned a2 = __builtin_amdgcn_readlane(key.w[2], ld), a3 = __builtin_amdgcn_readlane(key.w[3], ld);
                const bool same = pending && key.w[0] == a0 && key.w[1] == a1 && key.w[2] == a2 && key.w[3] == a3;
                if (same) leader_of = ld;
                pending = pending && !same;
            }
        }
        int found = 0;
        if (active && leader_of == lane) {
            unsigned h = (hash_key(key) >> 7) & (LP_SET - 1);
            found = -1;
            for (int probes = 0; probes < LP_MAX_PROBE; ) {
                const int t = __hip_atomic_load(&ltag[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (t == ST_EMPTY) {
                    int expected = ST_EMPTY;
                    if (__hip_atomic_compare_exchange_strong(&ltag[h], &expected, ST_LOCKED, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        __hip_atomic_store(&lkey[h][0], key.q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&lkey[h][1], key.q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&ltag[h], ST_FILLED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        lnew[atomicAdd(&n_new, 1u)] = (unsigned short)h;   // at most LP_SET entries are ever created
                        found = (int)h;
                        break;
                    }
                    continue;
                }
                if (t == ST_LOCKED) continue;
                const unsigned long long a = __hip_atomic_load(&lkey[h][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned long long b = __hip_atomic_load(&lkey[h][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (a == key.q[0] && b == key.q[1]) { found = (int)h; break; }
                h = (h + 1) & (LP_SET - 1);
                probes++;
            }
            if (found < 0)  // set too crowded around this key: go to the global table directly
                found = -1 - (int)hash_insert(L.state, L.tkeys, (unsigned)frame << L.cap_f_log2, L.cap_f_mask, L.counters, key);
        }
        lidx[r] = __shfl(found, leader_of, 64);
    }
    __syncthreads();
    const unsigned created = n_new;   // stable until the next chunk's phase 1, which starts after the barrier below
    for (unsigned t = resolved + threadIdx.x; t < created; t += 256) {
        const unsigned h = lnew[t];
        Key8 key;
        key.q[0] = lkey[h][0];
        key.q[1] = lkey[h][1];
        const unsigned fr = (unsigned)(unsigned short)key.k[7];
        lslot[h] = hash_insert(L.state, L.tkeys, fr << L.cap_f_log2, L.cap_f_mask, L.counters, key);
    }
    resolved = created;
    __syncthreads();
    if (real) {
        int so[D + 1];
#pragma unroll
        for (int r = 0; r <= D; r++) {
            const int li = lidx[r];
            so[r] = li >= 0 ? (int)lslot[li] : -1 - li;
        }
        store_row<D + 1>(L.offsets + gp * (D + 1), so);
        store_row<D + 1>(L.bary + gp * (D + 1), bary);
    }
  }
}

void launch_lattice_points(const LatticeDev& L, const FeatureSource& fs, hipStream_t s) {
    const long long total = (long long)L.Npad * L.n_frames;
    // chunks per block: as many as leave >= 1024 blocks (a single frame or a cloud is a latency case: 150 blocks of
    // 8 chunks kept three quarters of the chip idle for 106 us)
    int n_chunks = LP_CHUNKS;
    while (n_chunks > 1 && (total + 255) / 256 / n_chunks < 1024) n_chunks >>= 1;
    const long long per_block = 256ll * n_chunks;
    const dim3 grid((unsigned)((total + per_block - 1) / per_block)), block(256);
    switch (L.d) {
        case 1: lattice_points_kernel<1><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 2: lattice_points_kernel<2><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 3: lattice_points_kernel<3><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 4: lattice_points_kernel<4><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 5: lattice_points_kernel<5><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 6: lattice_points_kernel<6><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        case 7: lattice_points_kernel<7><<<grid, block, 0, s>>>(L, fs, n_chunks); break;
        default: break;
    }
    RV_LAUNCHED("lattice_points_kernel");
}

// ---------------------------------------------------------------------------------------------
// compaction: an exclusive scan over the slot occupancy numbers the vertices in slot order, i.e.
// deterministically and frame by frame
// ---------------------------------------------------------------------------------------------
struct SlotFilled {
    __device__ __forceinline__ int operator()(int st) const { return st == ST_FILLED ? 1 : 0; }
};

__global__ void __launch_bounds__(256)
lattice_compact_kernel(LatticeDev L) {
    const unsigned slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= L.cap_total) return;
    const bool filled = L.state[slot] == ST_FILLED;
    const int id = L.slot_to_id[slot];
    if (filled && (unsigned)id < (unsigned)L.m_bound) {
        L.vkeys[2 * (size_t)id] = L.tkeys[2 * (size_t)slot];
        L.vkeys[2 * (size_t)id + 1] = L.tkeys[2 * (size_t)slot + 1];
    }
    const unsigned cap_f = L.cap_f_mask + 1;
    if ((slot & L.cap_f_mask) == 0) L.fstart[slot >> L.cap_f_log2] = id;   // first vertex id of the frame
    if (slot == L.cap_total - 1) {
        const int M = id + (filled ? 1 : 0);
        L.counters[0] = M;
        L.fstart[L.n_frames] = M;
        if (M > L.m_bound) L.counters[1] = 1;
    }
    // load factor above 1/2 in a frame's region counts as overflow (checked at the region's last slot)
    if ((slot & L.cap_f_mask) == L.cap_f_mask) {
        const int first = L.slot_to_id[slot - L.cap_f_mask];
        const int Mf = id + (filled ? 1 : 0) - first;
        if ((unsigned)Mf > cap_f / 2) L.counters[1] = 1;
    }
}

// offsets: slot -> vertex id; sort keys / payloads for the stable vertex-major ordering
__global__ void __launch_bounds__(256)
lattice_remap_kernel(LatticeDev L, unsigned* __restrict__ sort_keys, unsigned* __restrict__ sort_vals, long long n_entries) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    int id = L.slot_to_id[L.offsets[e]];
    id = id < L.m_bound ? id : L.m_bound - 1;   // only after a (flagged) hash overflow
    L.offsets[e] = id;
    sort_keys[e] = (unsigned)id;
    sort_vals[e] = (unsigned)e;
}

// blur neighbours (permutohedral.cpp:296-318).  For axis j == d the +-d write of the reference
// lands on the coordinate that the d-length key ignores.
__global__ void __launch_bounds__(256)
lattice_neighbours_kernel(LatticeDev L) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int d = L.d;
    if (gid >= (long long)M * (d + 1)) return;
    const int j = (int)(gid / M);
    const int id = (int)(gid - (long long)j * M);
    Key8 key;
    const unsigned long long* src = L.vkeys + 2 * (size_t)id;
    key.q[0] = src[0];
    key.q[1] = src[1];
    Key8 n1 = key, n2 = key;
    for (int k = 0; k < d; k++) { n1.k[k] = (short)(key.k[k] - 1); n2.k[k] = (short)(key.k[k] + 1); }
    if (j < d) { n1.k[j] = (short)(key.k[j] + d); n2.k[j] = (short)(key.k[j] - d); }
    const unsigned fbase = (unsigned)(unsigned short)key.k[7] << L.cap_f_log2;
    const int s1 = hash_lookup(L.state, L.tkeys, fbase, L.cap_f_mask, n1);
    const int s2 = hash_lookup(L.state, L.tkeys, fbase, L.cap_f_mask, n2);
    L.nb1[(size_t)j * L.m_bound + id] = s1 < 0 ? -1 : L.slot_to_id[s1];
    L.nb2[(size_t)j * L.m_bound + id] = s2 < 0 ? -1 : L.slot_to_id[s2];
}

// CSR over the sorted entries: point index, barycentric weight, per-vertex [start, end)
__global__ void __launch_bounds__(256)
lattice_csr_kernel(LatticeDev L, const unsigned* __restrict__ keys_sorted, const unsigned* __restrict__ vals_sorted,
                   long long n_entries) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    const unsigned e = vals_sorted[k];
    const unsigned key = keys_sorted[k];
    L.csr_pw[k] = make_uint2(e / (unsigned)(L.d + 1), __float_as_uint(L.bary[e]));
    if (k == 0 || keys_sorted[k - 1] != key) L.vstart[key] = (unsigned)k;
    if (k == n_entries - 1 || keys_sorted[k + 1] != key) L.vend[key] = (unsigned)(k + 1);
}

void launch_vertex_order(const LatticeDev& L, SortBuffers& sb, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// Vertex-major ordering by a counting sort (fast path, used when a frame has at most CS_MCAP
// vertices -- the Segmenter kernel has ~300).  The entries of a frame are cut into wave-blocks of
// CS_PIX points; every wave walks its block in order, 64 entries at a time, and ranks equal vertex
// ids inside the chunk with ballots, so the order inside a vertex stays ascending in the point
// index without any comparison sort:
//   pass 1 (count)   per wave-block histogram over the frame's vertices (LDS) + slot -> id remap
//   scan             per frame: vertex start offsets and per-(wave-block, vertex) bases
//   pass 2 (scatter) same walk, entries land at base + rank
// ---------------------------------------------------------------------------------------------
constexpr int CS_PIX_MIN = 256;   // points per wave-block (LatticeDev::cs_pix): 256 .. 4096, a power of two
constexpr int CS_MCAP = 4096;   // 4 waves x 4096 counters = 64 KB of LDS at the largest fast-path capacity (2^13 slots per frame)

// bh holds, per frame, a dense [wave-block][vertex] matrix with row stride M_f; the frame's matrix
// starts at wbpf * fstart[frame] (so the whole array needs wbpf * M_total words).
template <bool SCATTER>
__global__ void __launch_bounds__(256)
csr_pass_kernel(LatticeDev L, unsigned* __restrict__ bh, int wbpf, int mcap) {
    extern __shared__ unsigned cs_cnt[];   // [4 waves][mcap]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long gwb = (long long)blockIdx.x * 4 + wave;
    const int frame = (int)(gwb / wbpf);
    if (frame >= L.n_frames) return;       // whole wave; no block-wide barrier below
    const int wb = (int)(gwb - (long long)frame * wbpf);
    // (clamps only matter after a flagged hash overflow; they keep every access in bounds)
    const int f0 = L.fstart[frame] < L.m_bound ? L.fstart[frame] : L.m_bound;
    const int f1 = L.fstart[frame + 1] < L.m_bound ? L.fstart[frame + 1] : L.m_bound;
    const int Mf = f1 - f0 < mcap ? f1 - f0 : mcap;
    const unsigned n_entries_total = (unsigned)((long long)L.n_frames * L.N * (L.d + 1));
    unsigned* my = cs_cnt + (size_t)wave * mcap;
    unsigned* row = bh + (size_t)wbpf * f0 + (size_t)wb * Mf;
    for (int lv = lane; lv < Mf; lv += 64) my[lv] = SCATTER ? row[lv] : 0u;
    if (Mf == 0 && lane == 0) my[0] = 0xFFFFFFFFu;   // no vertices (overflow only): positions fail the bound check
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int dp1 = L.d + 1;
    const long long p0 = (long long)wb * L.cs_pix;
    const long long p1 = p0 + L.cs_pix < L.N ? p0 + L.cs_pix : L.N;
    const long long ebeg = ((long long)frame * L.N + p0) * dp1, eend = ((long long)frame * L.N + p1) * dp1;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // software pipeline: the next chunk's loads are in flight while the current chunk is ranked
    auto fetch = [&](long long e, int& id, float& wgt) {
        id = 0; wgt = 0.f;
        if (e < eend) {
            // ids beyond the per-vertex arrays only occur after a (flagged) hash overflow: clamp so that
            // every later kernel stays in bounds; the host discards the result
            if (!SCATTER) { id = L.slot_to_id[L.offsets[e]]; id = id < L.m_bound ? id : L.m_bound - 1; }
            else { id = L.offsets[e]; wgt = L.bary[e]; }
        }
    };
    int id_n; float w_n;
    fetch(ebeg + lane, id_n, w_n);
    for (long long base = ebeg; base < eend; base += 64) {
        const long long e = base + lane;
        const bool valid = e < eend;
        const int id = id_n;
        const float wgt = w_n;
        fetch(e + 64, id_n, w_n);
        if (!SCATTER && valid) L.offsets[e] = id;    // slot -> vertex id, in place
        int lv = valid ? id - f0 : -1;
        if (lv >= Mf) lv = Mf - 1;                   // overflow case (flagged elsewhere): stay in bounds
        if (valid && lv < 0) lv = 0;
        if (!SCATTER) {
            // counting needs no order: LDS atomics (same-address lanes serialise in hardware, still an
            // order of magnitude cheaper than ranking the chunk with ballots)
            if (valid) atomicAdd(&my[lv], 1u);
            continue;
        }
        bool pending = valid;
        // distinct vertex ids of a chunk touch distinct counters, so the loop needs no ordering
        // inside a chunk; one fence per chunk orders the counters between chunks
        for (;;) {
            const unsigned long long todo = __ballot(pending);
            if (!todo) break;
            const int leader = __ffsll((long long)todo) - 1;
            const int k = __shfl(lv, leader, 64);
            const bool same = pending && lv == k;
            const unsigned long long m = __ballot(same);
            const unsigned c = (unsigned)__popcll(m);
            if (SCATTER) {
                const unsigned b = my[k];
                if (same) {
                    const unsigned pos = b + (unsigned)__popcll(m & lt);
                    if (pos < n_entries_total) {
                        L.csr_pw[pos] = make_uint2((unsigned)(e / dp1), __float_as_uint(wgt));   // one 8-byte store
                    }
                }
                if (lane == leader) my[k] = b + c;
            } else {
                if (lane == leader) my[k] = my[k] + c;
            }
            pending = pending && !same;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (!SCATTER) {
        for (int lv = lane; lv < Mf; lv += 64) row[lv] = my[lv];
    }
}

// Count pass, four entries per lane.  The generic pass above walks a wave-block 64 entries at a time with two DEPENDENT
// loads per step (slot, then slot -> vertex id): PMC showed its waves waiting 91 % of their cycles.  Counting needs no
// order, so a lane takes four consecutive entries at once -- one 16-byte load, four id gathers in flight together, one
// 16-byte store of the remapped ids -- and a wave-block is done in 7 steps instead of 28.
__global__ void __launch_bounds__(256)
csr_count_kernel(LatticeDev L, unsigned* __restrict__ bh, int wbpf, int mcap) {
    extern __shared__ unsigned cs_cnt[];   // [4 waves][mcap]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long gwb = (long long)blockIdx.x * 4 + wave;
    const int frame = (int)(gwb / wbpf);
    if (frame >= L.n_frames) return;       // whole wave; no block-wide barrier below
    const int wb = (int)(gwb - (long long)frame * wbpf);
    // (clamps only matter after a flagged hash overflow; they keep every access in bounds)
    const int f0 = L.fstart[frame] < L.m_bound ? L.fstart[frame] : L.m_bound;
    const int f1 = L.fstart[frame + 1] < L.m_bound ? L.fstart[frame + 1] : L.m_bound;
    const int Mf = f1 - f0 < mcap ? f1 - f0 : mcap;
    unsigned* my = cs_cnt + (size_t)wave * mcap;
    unsigned* row = bh + (size_t)wbpf * f0 + (size_t)wb * Mf;
    for (int lv = lane; lv < Mf; lv += 64) my[lv] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int dp1 = L.d + 1;
    const long long p0 = (long long)wb * L.cs_pix;
    const long long p1 = p0 + L.cs_pix < L.N ? p0 + L.cs_pix : L.N;
    const long long ebeg = ((long long)frame * L.N + p0) * dp1, eend = ((long long)frame * L.N + p1) * dp1;
    auto fetch = [&](long long e, int (&sl)[4]) {   // slots of entries e .. e + 3 (clamped into the wave-block)
        if (e + 4 <= eend) {
            load_row<4>(L.offsets + e, sl);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) sl[k] = e + k < eend ? L.offsets[e + k] : 0;
        }
    };
    int sl_n[4];
    fetch(ebeg + 4 * lane < eend ? ebeg + 4 * lane : ebeg, sl_n);
    for (long long base = ebeg; base < eend; base += 256) {
        const long long e = base + 4 * lane;
        int sl[4], id[4];
#pragma unroll
        for (int k = 0; k < 4; k++) sl[k] = sl_n[k];
        const bool any = e < eend;
        if (any) {
#pragma unroll
            for (int k = 0; k < 4; k++) id[k] = L.slot_to_id[sl[k]];   // four gathers in flight
        }
        const long long en = e + 256;
        fetch(en < eend ? en : ebeg, sl_n);                             // the next step's slots travel meanwhile
        if (!any) continue;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // ids beyond the per-vertex arrays only occur after a (flagged) hash overflow: clamp so that every later
            // kernel stays in bounds; the host discards the result
            id[k] = id[k] < L.m_bound ? id[k] : L.m_bound - 1;
        }
        if (e + 4 <= eend) {
            store_row<4>(L.offsets + e, id);                            // slot -> vertex id, in place
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) if (e + k < eend) L.offsets[e + k] = id[k];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (e + k >= eend) continue;
            int lv = id[k] - f0;
            lv = lv < Mf ? lv : Mf - 1;   // overflow case (flagged elsewhere): stay in bounds
            lv = lv < 0 ? 0 : lv;
            if (Mf > 0) atomicAdd(&my[lv], 1u);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int lv = lane; lv < Mf; lv += 64) row[lv] = my[lv];
}

// Scatter pass with one lane per POINT (the generic pass above has one lane per entry and ranks ~10
// distinct vertices per 64 entries; PMC: 438 vector instructions per 64 entries).  A chunk is 64
// consecutive points x DP1 entries.  For every distinct vertex k of the chunk the lanes that hold k
// -- in any of their DP1 slots, at most one per lane since a point's vertices are distinct -- are
// found with DP1 ballots; their union, masked to the lower lanes, is the rank of a point among the
// chunk's entries of k, i.e. ascending point order again.  Neighbouring points share their simplex,
// so a chunk has ~12-20 distinct vertices for 448 entries.
template <int DP1>
__global__ void __launch_bounds__(256)
csr_scatter_kernel(LatticeDev L, const unsigned* __restrict__ bh, int wbpf, int mcap) {
    extern __shared__ unsigned cs_cnt[];   // [4 waves][mcap]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long gwb = (long long)blockIdx.x * 4 + wave;
    const int frame = (int)(gwb / wbpf);
    if (frame >= L.n_frames) return;       // whole wave; no block-wide barrier below
    const int wb = (int)(gwb - (long long)frame * wbpf);
    // (clamps only matter after a flagged hash overflow; they keep every access in bounds)
    const int f0 = L.fstart[frame] < L.m_bound ? L.fstart[frame] : L.m_bound;
    const int f1 = L.fstart[frame + 1] < L.m_bound ? L.fstart[frame + 1] : L.m_bound;
    const int Mf = f1 - f0 < mcap ? f1 - f0 : mcap;
    const unsigned n_entries_total = (unsigned)((long long)L.n_frames * L.N * DP1);
    unsigned* my = cs_cnt + (size_t)wave * mcap;
    const unsigned* row = bh + (size_t)wbpf * f0 + (size_t)wb * Mf;
    for (int lv = lane; lv < Mf; lv += 64) my[lv] = row[lv];
    if (Mf == 0 && lane == 0) my[0] = 0xFFFFFFFFu;   // no vertices (overflow only): positions fail the bound check
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int p0 = wb * L.cs_pix;
    const int p1 = p0 + L.cs_pix < L.N ? p0 + L.cs_pix : L.N;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int pc = p0; pc < p1; pc += 64) {
        const int p = pc + lane;
        const bool valid = p < p1;
        const size_t gp = (size_t)frame * L.N + (valid ? p : p1 - 1);
        int lv[DP1];
        float w[DP1];
        load_row<DP1>(L.offsets + gp * DP1, lv);
        load_row<DP1>(L.bary + gp * DP1, w);
#pragma unroll
        for (int j = 0; j < DP1; j++) {
            lv[j] -= f0;
            lv[j] = lv[j] < Mf ? lv[j] : Mf - 1;   // overflow case (flagged elsewhere): stay in bounds
            lv[j] = lv[j] < 0 ? 0 : lv[j];
        }
        unsigned pend = valid ? (1u << DP1) - 1u : 0u;
#pragma unroll
        for (int j = 0; j < DP1; j++) {
            for (;;) {
                const unsigned long long todo = __ballot((pend >> j) & 1u);
                if (!todo) break;
                const int leader = __ffsll((long long)todo) - 1;
                const int k = __builtin_amdgcn_readlane(lv[j], leader);
                // slots below j are already empty for every lane; a vertex handled in an earlier pass
                // was removed from all slots then, so it cannot come up again in this chunk
                unsigned long long all = 0ull;
                float wsel = 0.0f;
                bool hit = false;
#pragma unroll
                for (int jj = j; jj < DP1; jj++) {
                    const bool same = ((pend >> jj) & 1u) && lv[jj] == k;
                    all |= __ballot(same);
                    if (same) { wsel = w[jj]; hit = true; pend &= ~(1u << jj); }
                }
                const unsigned b = my[k];
                if (hit) {
                    const unsigned pos = b + (unsigned)__popcll(all & lt);
                    if (pos < n_entries_total) L.csr_pw[pos] = make_uint2((unsigned)gp, __float_as_uint(wsel));
                }
                if (lane == leader) my[k] = b + (unsigned)__popcll(all);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
}

// per frame: bh[wb][lv] (counts) -> absolute base of (wave-block, vertex) in the csr arrays; vstart / vend per vertex.
// Two launches over (frame, group of 64 vertices) blocks of 1024 threads = 16 wave-block segments x 64 vertices: the
// first adds up the columns (per segment and whole), the second turns them into running bases.  The only thing a group
// needs from the others is the number of entries of the vertices before it, which it adds up itself from the column
// totals -- so the groups of a frame run side by side.  (Until round 3 one block per frame walked its groups one after
// the other: 139 us for a single 640x480 frame, 461 us for a 1.1 M-point cloud, on the critical path of both.)
constexpr int CS_SEGS = 16;
__device__ __forceinline__ size_t csr_aux_offset(const LatticeDev& L, int wbpf) { return (size_t)wbpf * ((size_t)L.m_bound + 64); }

__global__ void __launch_bounds__(1024)
csr_total_kernel(LatticeDev L, unsigned* __restrict__ bh, int wbpf, int mcap, int n_groups) {
    __shared__ unsigned sseg[CS_SEGS][64];
    const int frame = blockIdx.x / n_groups, grp = blockIdx.x - frame * n_groups;
    const int f0 = L.fstart[frame] < L.m_bound ? L.fstart[frame] : L.m_bound;
    const int f1 = L.fstart[frame + 1] < L.m_bound ? L.fstart[frame + 1] : L.m_bound;
    const int Mf = f1 - f0 < mcap ? f1 - f0 : mcap;
    const int lv0 = grp * 64;
    if (lv0 >= Mf) return;                   // whole block
    const int lvl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int sw = (wbpf + CS_SEGS - 1) / CS_SEGS;
    const int w0 = seg * sw, w1 = (w0 + sw < wbpf) ? w0 + sw : wbpf;
    const unsigned* fb = bh + (size_t)wbpf * f0;   // dense [wave-block][vertex] matrix, row stride Mf
    unsigned* vtot = bh + csr_aux_offset(L, wbpf);
    unsigned* segsum = vtot + L.m_bound;
    const int lv = lv0 + lvl;
    const bool ok = lv < Mf;
    unsigned sum = 0;
    if (ok) for (int w = w0; w < w1; w++) sum += fb[(size_t)w * Mf + lv];
    sseg[seg][lvl] = sum;
    if (ok) segsum[(size_t)(f0 + lv) * CS_SEGS + seg] = sum;
    __syncthreads();
    if (seg == 0 && ok) {
        unsigned total = 0;
#pragma unroll
        for (int q = 0; q < CS_SEGS; q++) total += sseg[q][lvl];
        vtot[f0 + lv] = total;
    }
}

__global__ void __launch_bounds__(1024)
csr_scan_kernel(LatticeDev L, unsigned* __restrict__ bh, int wbpf, int mcap, int n_groups) {
    __shared__ unsigned red[16];
    __shared__ unsigned vbase[64];
    const int frame = blockIdx.x / n_groups, grp = blockIdx.x - frame * n_groups;
    const int f0 = L.fstart[frame] < L.m_bound ? L.fstart[frame] : L.m_bound;
    const int f1 = L.fstart[frame + 1] < L.m_bound ? L.fstart[frame + 1] : L.m_bound;
    const int Mf = f1 - f0 < mcap ? f1 - f0 : mcap;
    const int lv0 = grp * 64;
    if (lv0 >= Mf) return;                   // whole block
    const int lvl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int sw = (wbpf + CS_SEGS - 1) / CS_SEGS;
    const int w0 = seg * sw, w1 = (w0 + sw < wbpf) ? w0 + sw : wbpf;
    unsigned* fb = bh + (size_t)wbpf * f0;
    const unsigned* vtot = bh + csr_aux_offset(L, wbpf);
    const unsigned* segsum = vtot + L.m_bound;
    const unsigned frame_base = (unsigned)((long long)frame * L.N * (L.d + 1));
    // entries of the frame's vertices before this group
    unsigned part = 0;
    for (int v = threadIdx.x; v < lv0; v += 1024) part += vtot[f0 + v];
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
    if (lvl == 0) red[seg] = part;
    __syncthreads();
    unsigned carry = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) carry += red[q];
    const int lv = lv0 + lvl;
    const bool ok = lv < Mf;
    if (seg == 0) {
        const unsigned total = ok ? vtot[f0 + lv] : 0u;
        unsigned incl = total;                // inclusive scan of the 64 column totals (one wave)
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (lvl >= off) incl += t;
        }
        vbase[lvl] = carry + incl - total;
        if (ok) {
            L.vstart[f0 + lv] = frame_base + carry + incl - total;
            L.vend[f0 + lv] = frame_base + carry + incl;
        }
    }
    __syncthreads();
    if (ok) {
        unsigned run = frame_base + vbase[lvl];
        for (int q = 0; q < seg; q++) run += segsum[(size_t)(f0 + lv) * CS_SEGS + q];
        for (int w = w0; w < w1; w++) {
            const unsigned t = fb[(size_t)w * Mf + lv];
            fb[(size_t)w * Mf + lv] = run;
            run += t;
        }
    }
}

bool csr_fast_path(const LatticeDev& L) { return ((L.cap_f_mask + 1) / 2) <= (unsigned)CS_MCAP; }
size_t csr_fast_bytes(const LatticeDev& L) {
    const size_t wbpf = ((size_t)L.N + L.cs_pix - 1) / L.cs_pix;
    // the [wave-block][vertex] matrices of all frames, then per vertex its column total and CS_SEGS segment sums
    return (wbpf * ((size_t)L.m_bound + 64) + (size_t)L.m_bound * (1 + CS_SEGS)) * sizeof(unsigned);
}

// phase 0: everything; 1: everything but the scatter of the counting-sort path (vertex numbering, neighbours, counts,
// list bounds, launch order: all the splat planner needs); 2: that scatter.  (The radix-sort path does it all in 0 / 1.)
void launch_lattice_finish(const LatticeDev& L, SortBuffers& sb, long long n_entries, hipStream_t s, int phase) {
    const bool fast = csr_fast_path(L) && sb.block_hist;
    if (phase == 2) {
        if (!fast) return;
        const int mcap = (int)((L.cap_f_mask + 1) / 2);
        const int wbpf = (L.N + L.cs_pix - 1) / L.cs_pix;
        const long long waves = (long long)wbpf * L.n_frames;
        const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
        const size_t lds = (size_t)4 * mcap * sizeof(unsigned);
        if (L.d == 6) csr_scatter_kernel<7><<<grid, block, lds, s>>>(L, sb.block_hist, wbpf, mcap);
        else if (L.d == 5) csr_scatter_kernel<6><<<grid, block, lds, s>>>(L, sb.block_hist, wbpf, mcap);
        else if (L.d == 2) csr_scatter_kernel<3><<<grid, block, lds, s>>>(L, sb.block_hist, wbpf, mcap);
        else csr_pass_kernel<true><<<grid, block, lds, s>>>(L, sb.block_hist, wbpf, mcap);
        RV_LAUNCHED("csr_scatter_kernel");
        return;
    }
    const unsigned cap = L.cap_total;
    {
        size_t temp = sb.scan_temp_bytes;
        auto in = rocprim::make_transform_iterator(L.state, SlotFilled());
        (void)rocprim::exclusive_scan(sb.scan_temp, temp, in, L.slot_to_id, 0, (size_t)cap, rocprim::plus<int>(), s);
    }
    lattice_compact_kernel<<<dim3((cap + 255) / 256), dim3(256), 0, s>>>(L);
    const long long nb_threads = (long long)L.m_bound * (L.d + 1);
    lattice_neighbours_kernel<<<dim3((unsigned)((nb_threads + 255) / 256)), dim3(256), 0, s>>>(L);
    RV_LAUNCHED("lattice_compact_kernel / lattice_neighbours_kernel");
    if (fast) {
        const int mcap = (int)((L.cap_f_mask + 1) / 2);
        const int wbpf = (L.N + L.cs_pix - 1) / L.cs_pix;
        const long long waves = (long long)wbpf * L.n_frames;
        const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
        const size_t lds = (size_t)4 * mcap * sizeof(unsigned);
        csr_count_kernel<<<grid, block, lds, s>>>(L, sb.block_hist, wbpf, mcap);
        const int n_groups = (mcap + 63) / 64;
        csr_total_kernel<<<dim3((unsigned)(L.n_frames * n_groups)), dim3(1024), 0, s>>>(L, sb.block_hist, wbpf, mcap, n_groups);
        csr_scan_kernel<<<dim3((unsigned)(L.n_frames * n_groups)), dim3(1024), 0, s>>>(L, sb.block_hist, wbpf, mcap, n_groups);
        RV_LAUNCHED("csr_count_kernel / csr_total_kernel / csr_scan_kernel");
        if (phase == 0) launch_lattice_finish(L, sb, n_entries, s, 2);
    } else {
        lattice_remap_kernel<<<dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, s>>>(L, sb.keys_in, sb.vals_in, n_entries);
        // stable radix sort by vertex id: equal keys keep ascending entry (= point) order
        size_t temp = sb.temp_bytes;
        (void)rocprim::radix_sort_pairs(sb.temp, temp, sb.keys_in, sb.keys_out, sb.vals_in, sb.vals_out, (size_t)n_entries, 0,
                                        (unsigned)sb.key_bits, s);
        lattice_csr_kernel<<<dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, s>>>(L, sb.keys_out, sb.vals_out, n_entries);
        RV_LAUNCHED("lattice_remap_kernel / radix sort / lattice_csr_kernel");
    }
    launch_vertex_order(L, sb, s);
}

// Launch order of the vertices for the splat: per frame, longest list first.  vorder[fstart[f] + k]
// is the k-th longest vertex of frame f (ids are frame-contiguous, so a sort by (frame, -length)
// keeps every frame in its own id range).  The splat forms its groups of G vertices inside a frame
// and starts all frames' heaviest groups first (LPT: the serial chains of the heaviest vertices
// start at t = 0).
__global__ void __launch_bounds__(256)
vertex_len_kernel(LatticeDev L, unsigned* __restrict__ key, unsigned* __restrict__ ids, int len_shift) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= L.m_bound) return;
    const int M = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    unsigned k = 0xFFFFFFFFu;  // unused ids sort to the end
    if (v < M) {
        atomicMax(&L.counters[3], (int)(L.vend[v] - L.vstart[v]));   // the longest chain of the chunk (rvseg_last_schedule)
        const unsigned len = (L.vend[v] - L.vstart[v]) >> len_shift;
        const unsigned frame = (unsigned)(unsigned short)(L.vkeys[2 * (size_t)v + 1] >> 48);
        k = ((frame < 1023u ? frame : 1022u) << 22) | (0x3FFFFFu - (len < 0x3FFFFFu ? len : 0x3FFFFFu));
    }
    key[v] = k;
    ids[v] = (unsigned)v;
}

void launch_vertex_order(const LatticeDev& L, SortBuffers& sb, hipStream_t s) {
    int len_shift = 0;   // a list has at most N entries
    while (((long long)L.N >> len_shift) >= (1 << 22)) len_shift++;
    vertex_len_kernel<<<dim3((unsigned)((L.m_bound + 255) / 256)), dim3(256), 0, s>>>(L, sb.keys_in, sb.vals_in, len_shift);
    size_t temp = sb.temp_bytes;
    (void)rocprim::radix_sort_pairs(sb.temp, temp, sb.keys_in, sb.keys_out, sb.vals_in, L.vorder, (size_t)L.m_bound, 0, 32, s);
    RV_LAUNCHED("vertex_len_kernel / radix sort");
}

size_t scan_temp_bytes(unsigned cap) {
    size_t temp = 0;
    int* nul = nullptr;
    auto in = rocprim::make_transform_iterator(nul, SlotFilled());
    (void)rocprim::exclusive_scan(nullptr, temp, in, nul, 0, (size_t)cap, rocprim::plus<int>(), (hipStream_t)0);
    return temp;
}

size_t sort_temp_bytes(long long n_entries, int key_bits) {
    size_t temp = 0;
    unsigned* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, temp, nul, nul, nul, nul, (size_t)n_entries, 0, (unsigned)key_bits, (hipStream_t)0);
    return temp;
}

// norm values gathered into CSR order once the normaliser exists
__global__ void __launch_bounds__(256)
csr_norm_kernel(const uint2* __restrict__ csr_pw, const float* __restrict__ norm, float* __restrict__ csr_nrm, long long n_entries,
                const int* __restrict__ counters) {
    if (counters[1]) return;   // hash overflow (flagged): the csr arrays are incomplete
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_entries) csr_nrm[k] = norm[csr_pw[k].x];
}

void launch_csr_norm(const LatticeDev& L, long long n_entries, hipStream_t s) {
    csr_norm_kernel<<<dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, s>>>(L.csr_pw, L.norm, L.csr_nrm, n_entries, L.counters);
    RV_LAUNCHED("csr_norm_kernel");
}

// ---------------------------------------------------------------------------------------------
// splat as an ordered gather: one lane per (vertex, class) chain.
//   values[v][c] = sum over the vertex's entries, ascending point index, of fl(w * in[p][c])
//   with in[p][c] = fl(Q[p][c] * norm[p]) when `scaled` (DenseKernel::filter, pairwise.cpp:66)
// Per tile of 64 list entries the lanes of a producer wave form the 64 x C products in parallel
// (coalesced reads of the CSR pairs, one gathered Q row per lane) and park them in LDS; a lane
// (vertex, class) of the adder wave then adds the tile's products in list order -- the only part
// that has to be sequential.
// MODE 0: in = src[p*C+c]; 1: in = fl(src*norm) (per-entry normaliser csr_nrm); 2: in = 1 (normaliser pass).
// CC = classes handled by a pass (compile time, so the body is branch-free and the compiler keeps
// counted vmcnt waits); classes [c0, c0 + n_store) are stored, n_store <= CC.
// ---------------------------------------------------------------------------------------------
// The first version ran one wave per vertex (products and adds in the same wave).  PMC showed it
// issue bound: of ~150 instructions per 64-entry tile, 64 adds + 16 LDS reads run with only C of 64
// lanes busy.  Here a block owns G = 64 / CC vertices of similar list length (neighbours in
// `vorder`): wave i < G forms the products of vertex i's tile t exactly as above, and ONE extra
// wave adds them for all G vertices at once -- lane (i, c) walks vertex i's class-c products in
// list order -- so the sequential phase costs 64 adds per G tiles.  Products are double buffered:
// the adder works on tile t while the producers write tile t + 1; one barrier per tile.
template <int CC> struct SplatGroup { static constexpr int G = 64 / CC > 8 ? 8 : 64 / CC; };
// A vertex's chain advances one tile per barrier, so the loads of a tile have to be in flight for
// many tiles: {point, weight} pairs are fetched SPLAT_RE - 1 tiles ahead and the Q rows they point at
// SPLAT_RR - 1 tiles ahead (a ring of 3 / 2 tiles stalled a memory round trip per tile).
constexpr int SPLAT_RE = 16, SPLAT_RR = 8;   // the stage list in the kernel is written out for RE == 16; RR must divide RE
static_assert(SPLAT_RE % SPLAT_RR == 0, "ring positions are compile-time: the row ring has to divide the entry ring");

// GV = vertices per block (<= G).  With GV = G the block is G producer waves + the adder as the last wave.  A block
// with fewer vertices (GV = 6 for C = 9: 7 waves) puts the adder at wave 3: a workgroup's waves go to the four SIMDs
// in cyclic order (MI355X_MICROARCH.md, LDS section), so waves w and w + 4 share a SIMD and wave 3 of a 7-wave block
// has one to itself -- its 64 dependent adds per step no longer compete with a producer for issue slots.  That
// shortens a step (the critical path of launches with few, long chains: a single frame, a 1280x960 chunk, a cloud)
// at the price of 7/6 as many block-steps; chunks with many frames are bandwidth bound and keep GV = G.
// FAST: the input is this library's own Q * norm (finite, >= 0) in one contiguous [point][C] matrix: a padding lane's
// product is 0 * x = +0 by itself (no select), and the row address needs no per-frame split.
// One item of the list-major walk: G vertices of one frame, whole lists.  The kernel below runs one item per block; the
// resident kernel falls back to a loop over these items when its planner gave up.
// NH = entries per producer lane and tile: 1 (64-entry tiles, rings of 16 / 8 tiles) or 2 (128-entry tiles, rings of 8 / 4:
// the same look-ahead in time).  A launch whose time is its longest chain (a single frame, a cloud, chunks of <= 16
// frames) pays the per-tile costs -- barrier, table reads, the wait for the first product row -- per 128 dependent adds
// instead of per 64 with NH = 2.
template <int MODE, int CC, bool FULL, int GV, bool FAST, int NH = 1>
__device__ __forceinline__ void splat_group_item(const LatticeDev& L, const ValueView& src, int C, int c0, int n_store, float* __restrict__ values,
                                                 unsigned item, float (*prod)[GV][CC][64 * NH + 4]) {
    constexpr int G = GV;
    constexpr int TE = 64 * NH;                       // entries per tile
    constexpr int RE_ = NH == 2 ? 8 : SPLAT_RE, RR_ = NH == 2 ? 4 : SPLAT_RR;
    static_assert(NH == 1 || NH == 2, "entries per lane");
    constexpr int AW = (GV < SplatGroup<CC>::G && GV >= 4) ? 3 : GV;   // the adder's wave index
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // item b -> XCD group b % n_groups (the dispatcher deals blocks round-robin over the 8 XCDs), item
    // b / n_groups inside it.  A group owns the frames f = g, g + n_groups, ...: all readers of a
    // frame's Q rows share one L2.  Item j = (rank r, frame slot q): the r-th heaviest G vertices of
    // that frame -- every frame's heaviest vertices are dispatched first.
    const unsigned g = item % (unsigned)L.n_groups, j = item / (unsigned)L.n_groups;
    const unsigned nfg = ((unsigned)L.n_frames - g + (unsigned)L.n_groups - 1u) / (unsigned)L.n_groups;   // frames of this group
    if (nfg == 0) return;
    const unsigned r = j / nfg, frame = g + (j - r * nfg) * (unsigned)L.n_groups;
    const int Mtot = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int fs0 = L.fstart[frame] < Mtot ? L.fstart[frame] : Mtot;
    const int fs1 = L.fstart[frame + 1] < Mtot ? L.fstart[frame + 1] : Mtot;
    const unsigned n_vert = (unsigned)(fs1 - fs0), gstart = (unsigned)fs0;
    if (r * G >= n_vert) return;
    const bool contig = FAST || (src.frame_stride == (size_t)L.N * (size_t)C && src.layer_off == 0);
    const int pw = wave < AW ? wave : wave - 1;   // producer index of this wave (unused by the adder)
    {
        // every wave reads the group's G list ranges itself (uniform): no broadcast step
        unsigned n_steps = 0, my_k0 = 0, my_k1 = 0;
        for (int i = 0; i < G; i++) {
            const unsigned idx = r * G + i;
            if (idx < n_vert) {
                const unsigned v = L.vorder[gstart + idx];
                const unsigned k0 = L.vstart[v], k1 = L.vend[v];
                if (idx < L.scan_ranks && k1 - k0 >= L.heavy_from) continue;   // a scan block's (splat_scan_item)
                const unsigned nt = (k1 - k0 + (unsigned)TE - 1u) / (unsigned)TE;
                n_steps = nt > n_steps ? nt : n_steps;
                if (i == pw && wave != AW) { my_k0 = k0; my_k1 = k1; }
            }
        }
        // the adder's 64 dependent adds are the critical path of every step: it wins issue arbitration
        // against the producers (which run a tile ahead and have slack)
        if (wave == AW) __builtin_amdgcn_s_setprio(3);
        else if (n_steps > 256u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        if (wave != AW) {
            // ---- producer of vertex `wave`: barriers 0 .. n_steps - 1 close its tiles, one more ends the item
            const bool has = my_k1 > my_k0;
            const unsigned kc0 = has ? my_k0 : 0u, kc1 = has ? my_k1 : 1u;   // clamp range of the loads (entry 0 exists)
            const unsigned n_tiles = has ? (my_k1 - my_k0 + (unsigned)TE - 1u) / (unsigned)TE : 0u;
            float x[RR_][NH][CC];
            float w[RE_][NH], nrm[RE_][NH];
            unsigned pix[RE_][NH];
#pragma unroll
            for (int r = 0; r < RE_; r++)
#pragma unroll
                for (int h = 0; h < NH; h++) { w[r][h] = 0.f; nrm[r][h] = 1.f; pix[r][h] = 0u; }
            // loads are unconditional (indices clamped into the list): no divergent branch, counted waits
            auto load_entries = [&](unsigned tile, int slot) {
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    unsigned k = kc0 + tile * (unsigned)TE + (unsigned)lane + 64u * h;
                    k = k < kc1 ? k : kc1 - 1u;
                    const uint2 pw = L.csr_pw[k];
                    w[slot][h] = __uint_as_float(pw.y);
                    pix[slot][h] = pw.x;
                    if (MODE == 1) nrm[slot][h] = L.csr_nrm[k];
                }
            };
            auto gather_rows = [&](int eslot, int rslot) {
                if (MODE == 2) return;
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    // one frame-contiguous [point][C] matrix (the single-layer case): no division by N
                    const size_t row = contig ? (size_t)pix[eslot][h] * (unsigned)C + (unsigned)c0 : src.index(pix[eslot][h], c0, C, L.N);
                    if (FULL) {
                        load_row<CC>(src.base + row, x[rslot][h]);
                    } else {
#pragma unroll
                        for (int c = 0; c < CC; c++) x[rslot][h][c] = src.base[row + (c < n_store ? c : n_store - 1)];
                    }
                }
            };
#pragma unroll
            for (int i = 0; i < RE_ - 1; i++) load_entries((unsigned)i, i);
#pragma unroll
            for (int i = 0; i < RR_ - 1; i++) gather_rows(i, i);
            // one tile: products -> LDS, refill the two rings, barrier.  `S` is a compile-time ring position
            // (the rings live in registers); a `false` return leaves the loop, so no stage is reachable
            // past a skipped one and the compiler keeps counted vmcnt waits.
            auto stage = [&](unsigned t, auto S) -> bool {
                constexpr int s = decltype(S)::value;
                if (t >= n_steps) return false;
                const unsigned base = my_k0 + t * (unsigned)TE;
                const unsigned n_valid = t < n_tiles ? (my_k1 - base < (unsigned)TE ? my_k1 - base : (unsigned)TE) : 0u;
                float (*pb)[TE + 4] = prod[t & 1u][pw];
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const bool in = (unsigned)lane + 64u * h < n_valid;
                    const float wl = in ? w[s][h] : 0.0f;
#pragma unroll
                    for (int c = 0; c < CC; c++) {
                        float xin = MODE == 2 ? 1.0f : x[s % RR_][h][c];
                        if (MODE == 1) xin = xin * nrm[s][h];
                        const float pr = wl * xin;
                        // +0 past the list: identity of the sum (FAST: wl is 0 there and the row is finite, so pr is +0 already)
                        pb[c][lane + 64 * h] = (FAST || in) ? pr : 0.0f;
                    }
                }
                load_entries(t + RE_ - 1, (s + RE_ - 1) % RE_);
                gather_rows((s + RR_ - 1) % RE_, (s + RR_ - 1) % RR_);
                __syncthreads();
                return true;
            };
#define RV_ST(i) if (!stage(t0 + i, std::integral_constant<int, i>())) break;
            for (unsigned t0 = 0;; t0 += RE_) {
                RV_ST(0) RV_ST(1) RV_ST(2) RV_ST(3) RV_ST(4) RV_ST(5) RV_ST(6) RV_ST(7)
                if constexpr (RE_ == 16) {
                    RV_ST(8) RV_ST(9) RV_ST(10) RV_ST(11) RV_ST(12) RV_ST(13) RV_ST(14) RV_ST(15)
                }
            }
#undef RV_ST
            __syncthreads();
        } else {
            // ---- adder: lane (i, c) owns the chain of vertex i, class c
            const int gi = lane < G * CC ? lane / CC : 0, c = lane < G * CC ? lane % CC : 0;
            const unsigned idx = r * G + gi;
            bool mine = lane < G * CC && idx < n_vert && c < n_store;
            const unsigned cv = mine ? L.vorder[gstart + idx] : 0u;
            if (mine && idx < L.scan_ranks && L.vend[cv] - L.vstart[cv] >= L.heavy_from) mine = false;   // a scan block's
            float acc = 0.0f;
            __syncthreads();
            for (unsigned t = 0; t < n_steps; t++) {
                const float* pr = prod[t & 1u][gi][c];
                float4 q[16 * NH];
#pragma unroll
                for (int i = 0; i < 16 * NH; i++) q[i] = reinterpret_cast<const float4*>(pr)[i];
#pragma unroll
                for (int i = 0; i < 16 * NH; i++) { acc += q[i].x; acc += q[i].y; acc += q[i].z; acc += q[i].w; }
                __syncthreads();
            }
            if (mine) values[(size_t)cv * C + c0 + c] = acc;
        }
    }
}

template <int MODE, int CC, bool FULL, int GV = SplatGroup<CC>::G, bool FAST = false, int NH = 1>   // FULL: all CC classes exist (n_store == CC): rows are fetched with wide loads
__global__ void __launch_bounds__((GV + 1) * 64)
splat_group_kernel(LatticeDev L, ValueView src, int C, int c0, int n_store, float* __restrict__ values) {
    __shared__ __attribute__((aligned(16))) float prod[2][GV][CC][64 * NH + 4];  // 16-B aligned rows, 4-bank skew
    if (L.counters[1]) return;   // hash overflow (flagged): the CSR arrays are incomplete, touch nothing
    splat_group_item<MODE, CC, FULL, GV, FAST, NH>(L, src, C, c0, n_store, values, blockIdx.x, prod);
}

template <int MODE, int CC, int GV, bool FAST, int NH = 1>
static void splat_group_launch_g(const LatticeDev& L, const ValueView& src, int C, int c0, int n, float* values, hipStream_t s) {
    constexpr int G = GV;
    // items per XCD group: (frames of the group) x (groups of G vertices a frame can have at most)
    const unsigned nfg = ((unsigned)L.n_frames + (unsigned)L.n_groups - 1u) / (unsigned)L.n_groups;
    const unsigned long long max_mf = std::min<unsigned long long>(((unsigned long long)L.cap_f_mask + 1) / 2 + 1, (unsigned long long)L.m_bound);
    const unsigned per_group = nfg * (unsigned)((max_mf + G - 1) / G);
    const dim3 grid(per_group * (unsigned)L.n_groups), block((G + 1) * 64);
    if (n == CC) splat_group_kernel<MODE, CC, true, GV, FAST, NH><<<grid, block, 0, s>>>(L, src, C, c0, n, values);
    else splat_group_kernel<MODE, CC, false, GV, false><<<grid, block, 0, s>>>(L, src, C, c0, n, values);
    RV_LAUNCHED("splat_group_kernel");
}

template <int MODE, int CC>
static void splat_group_launch(const LatticeDev& L, const ValueView& src, int C, int c0, int n, float* values, hipStream_t s) {
    splat_group_launch_g<MODE, CC, SplatGroup<CC>::G, false>(L, src, C, c0, n, values, s);
}

template <int MODE>
static void splat_group_pass(const LatticeDev& L, const ValueView& src, int C, int c0, int n, float* values, hipStream_t s) {
    if (n == 1) splat_group_launch<MODE, 1>(L, src, C, c0, n, values, s);
    else if (n == 2) splat_group_launch<MODE, 2>(L, src, C, c0, n, values, s);
    else if (n <= 4) splat_group_launch<MODE, 4>(L, src, C, c0, n, values, s);
    else if (n <= 8) splat_group_launch<MODE, 8>(L, src, C, c0, n, values, s);
    else if (n == 9) splat_group_launch<MODE, 9>(L, src, C, c0, n, values, s);
    else splat_group_launch<MODE, 16>(L, src, C, c0, n, values, s);
}


// ---------------------------------------------------------------------------------------------
// The normaliser's splat (C = 1: a vertex's value is the fp32 sum of its entries' barycentric weights, added in list
// order from +0) without the serial chain.  The ordered sum only LOOKS sequential: while the running sum s stays in one
// binade [2^E, 2^(E+1)) every partial sum is a multiple of u = 2^(E-23), and adding w >= 0 to it rounds s + w to the
// nearest multiple of u -- in units of u: n + k  ->  n + rne(k), k = w / u, unless k lies exactly half-way between two
// integers (then the direction depends on the parity of n).  So, as long as no addend of a tile is negative or such a
// tie and the tile does not leave the binade,
//       s_after = (n + sum_i rne(k_i)) * u ,
// a sum of integers below 2^24 -- exact in fp32 in ANY order.  A wave adds a tile of 128 or 256 addends (two or four
// per lane) with a few additions per lane and six DPP steps instead of that many dependent additions; a tile that breaks
// a condition (the first one, ~17 binade crossings and a few dozen ties per long list: 4-5 % of the 64-entry tiles of a
// bench frame's heaviest lists) is tried again in two halves, and a half that breaks one is added the reference's way,
// one addend after the other.  The result is bit-identical to the sequential sum by construction and is tested
// against it (tests/test_gpu_crf.py: test_normaliser_ordered_sums_...).
// The same idea does not pay for the C-class splats: there the serial adder already runs 54 chains in its 64 lanes.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {   // the DPP-selected lane's v, +0 where the pattern selects none
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// sum over the wave, valid in lane 63 (inclusive row scans, then the row totals travel up)
__device__ __forceinline__ float wave_total_lane63(float v) {
    v += dpp_term<0x111, 0xf>(v);   // row_shr:1
    v += dpp_term<0x112, 0xf>(v);   // row_shr:2
    v += dpp_term<0x114, 0xf>(v);   // row_shr:4
    v += dpp_term<0x118, 0xf>(v);   // row_shr:8
    v += dpp_term<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp_term<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_term_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, true);
}
// sum over the wave, the same value in every lane
__device__ __forceinline__ unsigned wave_total_u32(unsigned v) {
    v += dpp_term_u<0x111, 0xf>(v);
    v += dpp_term_u<0x112, 0xf>(v);
    v += dpp_term_u<0x114, 0xf>(v);
    v += dpp_term_u<0x118, 0xf>(v);
    v += dpp_term_u<0x142, 0xa>(v);
    v += dpp_term_u<0x143, 0xc>(v);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_term_keep(unsigned v) {   // the DPP-selected lane's v, ~0 where the pattern selects none
    return (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// minimum over the wave, the same value in every lane
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    unsigned t;
    t = dpp_term_keep<0x111, 0xf>(v); v = t < v ? t : v;
    t = dpp_term_keep<0x112, 0xf>(v); v = t < v ? t : v;
    t = dpp_term_keep<0x114, 0xf>(v); v = t < v ? t : v;
    t = dpp_term_keep<0x118, 0xf>(v); v = t < v ? t : v;
    t = dpp_term_keep<0x142, 0xa>(v); v = t < v ? t : v;
    t = dpp_term_keep<0x143, 0xc>(v); v = t < v ? t : v;
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// One attempt at the lanes `mine` (K addends per lane, lane l holding entries K l .. K l + K - 1 of the tile): true and
// s advanced if the conditions above hold for them, false and s untouched otherwise.
template <int K>
__device__ __forceinline__ bool ordered_try(float& s, const float (&w)[K], bool mine) {
    const unsigned sb = __float_as_uint(s);
    const unsigned e = (sb >> 23) & 0xffu;                              // biased exponent of the running sum
    const bool s_ok = ((int)sb > 0) & (e >= 24u) & (e <= 253u);        // positive, normal, scale factors representable
    const float scale = __uint_as_float((277u - (s_ok ? e : 127u)) << 23);      // 2^(23 - E) = 1 / u
    const float unscale = __uint_as_float(((s_ok ? e : 127u) - 23u) << 23);     // u
    float rs = 0.0f;
    bool bad = false;
#pragma unroll
    for (int h = 0; h < K; h++) {
        const float k = w[h] * scale;                 // exact (a power of two)
        const float r = __builtin_rintf(k);           // round half to even, like the addition itself
        bad |= !(w[h] >= 0.0f) | !(k < 16777216.0f) | (__builtin_fabsf(k - r) == 0.5f);
        rs += r;
    }
    rs = mine ? rs : 0.0f;
    bad &= mine;
    // (sums of integers: exact while below 2^24, and not below 2^24 once the true sum is not -- rounding is monotone)
    const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_total_lane63(rs)), 63));
    const float S = s * scale + total;
    const bool ok = s_ok & (__builtin_amdgcn_ballot_w64(bad) == 0ull) & (S < 16777216.0f);
    if (ok) s = S * unscale;
    return ok;
}
// the reference's way for lanes [LO, HI): one addition per addend, in list order
template <int K, int LO, int HI>
__device__ __forceinline__ float ordered_serial(float s, const float (&w)[K]) {
#pragma unroll
    for (int i = LO; i < HI; i++)
#pragma unroll
        for (int h = 0; h < K; h++) s = s + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[h]), i));
    return s;
}
// s + (a tile of 64 K addends), rounded after every addition like the sequential loop: the whole tile at once if it can
// be, else its two halves, each at once or serially
template <int K>
__device__ __forceinline__ float ordered_tile_sum(float s, const float (&w)[K]) {
    if (ordered_try<K>(s, w, true)) return s;
    const bool low = (threadIdx.x & 63) < 32;
    if (!ordered_try<K>(s, w, low)) s = ordered_serial<K, 0, 32>(s, w);
    if (!ordered_try<K>(s, w, !low)) s = ordered_serial<K, 32, 64>(s, w);
    return s;
}

// Two kernels share the vertices of a frame (`vorder`: longest list first).
//  * Lists of NS_HEAVY entries and more -- the chains that set the time of a launch with few frames: one block per
//    vertex; wave 0 sums, waves 1..3 bring the list's weights into LDS three batches ahead of it (a single wave cannot
//    keep enough loads in flight for itself: it sums 128 entries in ~0.1 us, a load takes 1-3 us to come back, and a
//    register ring deep enough for that defeated the compiler's wait counting -- every variant ended in s_waitcnt
//    vmcnt(0) or (1) per tile and ran at the speed of the serial chain).  One barrier per batch of 2 048 entries.
//  * The short lists, thousands of them: one wave per vertex straight from global memory; their loads' latency is
//    hidden by the other waves.
constexpr int NS_HEAVY = 8192;                 // entries from which a list gets a block of its own
constexpr int NS_BATCH = 2048;                 // entries per batch
constexpr int NS_RING = 4;                     // batches of floats in LDS (32 KB): one being summed, three on their way
constexpr int NS_K = 4;                        // entries per lane and tile on the heavy path: tiles of 256
constexpr int NS_SEGS = NS_BATCH / (64 * NS_K);   // tiles of a batch: one ordered_tile_sum each
constexpr int NS_PROD = 3;                     // producer waves
constexpr int NS_PER = (NS_SEGS + NS_PROD - 1) / NS_PROD;

struct NormItem { unsigned frame, r; int fs0; unsigned n_vert; bool ok; };
// item -> (rank r, frame) as in splat_group_item: every frame's heaviest vertices are dispatched first
__device__ __forceinline__ NormItem norm_item(const LatticeDev& L, unsigned item) {
    NormItem it{0u, 0u, 0, 0u, false};
    const unsigned g = item % (unsigned)L.n_groups, j = item / (unsigned)L.n_groups;
    const unsigned nfg = ((unsigned)L.n_frames - g + (unsigned)L.n_groups - 1u) / (unsigned)L.n_groups;
    if (nfg == 0) return it;
    it.r = j / nfg;
    it.frame = g + (j - it.r * nfg) * (unsigned)L.n_groups;
    const int Mtot = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    it.fs0 = L.fstart[it.frame] < Mtot ? L.fstart[it.frame] : Mtot;
    const int fs1 = L.fstart[it.frame + 1] < Mtot ? L.fstart[it.frame + 1] : Mtot;
    it.n_vert = (unsigned)(fs1 - it.fs0);
    it.ok = true;
    return it;
}

__device__ __forceinline__ void norm_sum_heavy(const LatticeDev& L, float* __restrict__ values, unsigned item, float (*buf)[NS_BATCH]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const NormItem it = norm_item(L, item);
    if (!it.ok || it.r >= it.n_vert) return;
    const unsigned v = L.vorder[(unsigned)it.fs0 + it.r];
    const unsigned k0 = L.vstart[v], k1 = L.vend[v];
    const unsigned len = k1 - k0;
    if (len < (unsigned)NS_HEAVY) return;   // (whole block) a light block's
    const unsigned n_batch = (len + NS_BATCH - 1u) / NS_BATCH;
    const float* wgt = reinterpret_cast<const float*>(L.csr_pw) + 1;   // the weight of entry k is wgt[2 k]
    // producer wave: its share of batch q -- issue (loads, all in flight together) and, an iteration or two later, commit
    auto issue = [&](unsigned q, float (&x)[NS_PER][NS_K]) {
        const unsigned base = k0 + q * NS_BATCH + (unsigned)(NS_K * lane);
#pragma unroll
        for (int i = 0; i < NS_PER; i++) {
            const unsigned e = base + (unsigned)(64 * NS_K) * (unsigned)((wave - 1) + NS_PROD * i);
#pragma unroll
            for (int h = 0; h < NS_K; h++) x[i][h] = wgt[2 * (size_t)__builtin_elementwise_min(e + h, k1 - 1u)];
        }
    };
    auto commit = [&](unsigned q, const float (&x)[NS_PER][NS_K]) {
        const unsigned base = k0 + q * NS_BATCH + (unsigned)(NS_K * lane);
#pragma unroll
        for (int i = 0; i < NS_PER; i++) {
            const int seg = (wave - 1) + NS_PROD * i;
            const unsigned e = base + (unsigned)(64 * NS_K) * (unsigned)seg;
            if (seg < NS_SEGS)   // +0 past the end of the list: the identity of the sum
                *reinterpret_cast<float4*>(&buf[q % NS_RING][seg * 64 * NS_K + NS_K * lane]) =
                    make_float4(e < k1 ? x[i][0] : 0.0f, e + 1u < k1 ? x[i][1] : 0.0f, e + 2u < k1 ? x[i][2] : 0.0f, e + 3u < k1 ? x[i][3] : 0.0f);
        }
    };
    // batches q + 1 and q + 2 travel in registers (xa: odd, xb: even batch numbers) while batch q is summed
    static_assert(NS_K == 4, "the LDS tiles are float4 per lane");
    float xa[NS_PER][NS_K], xb[NS_PER][NS_K];
    float s = 0.0f;
    if (wave > 0) {
        issue(0, xb); commit(0, xb);
        issue(1, xa); issue(2, xb);
    }
    __syncthreads();
    for (unsigned q = 0; q < n_batch; q += 2) {
        // even iteration: batch q is summed, q + 1 (xa) is committed, q + 3 issued into xa
        if (wave > 0) { commit(q + 1u, xa); issue(q + 3u, xa); }
        else {
            const unsigned here = len - q * NS_BATCH < (unsigned)NS_BATCH ? len - q * NS_BATCH : (unsigned)NS_BATCH;
            const float4* pb = reinterpret_cast<const float4*>(buf[q % NS_RING]) + lane;
            const unsigned n_seg = (here + (unsigned)(64 * NS_K - 1)) / (unsigned)(64 * NS_K);
            float4 w = pb[0];
            for (unsigned sg = 0; sg < n_seg; sg++) {   // the next tile's LDS read travels during this tile's sum
                const float4 wn = pb[64u * (sg + 1u < (unsigned)NS_SEGS ? sg + 1u : sg)];
                const float wk[NS_K] = {w.x, w.y, w.z, w.w};
                s = ordered_tile_sum<NS_K>(s, wk);
                w = wn;
            }
        }
        __syncthreads();
        if (q + 1u >= n_batch) break;
        // odd iteration: batch q + 1 is summed, q + 2 (xb) is committed, q + 4 issued into xb
        if (wave > 0) { commit(q + 2u, xb); issue(q + 4u, xb); }
        else {
            const unsigned q1 = q + 1u;
            const unsigned here = len - q1 * NS_BATCH < (unsigned)NS_BATCH ? len - q1 * NS_BATCH : (unsigned)NS_BATCH;
            const float4* pb = reinterpret_cast<const float4*>(buf[q1 % NS_RING]) + lane;
            const unsigned n_seg = (here + (unsigned)(64 * NS_K - 1)) / (unsigned)(64 * NS_K);
            float4 w = pb[0];
            for (unsigned sg = 0; sg < n_seg; sg++) {   // the next tile's LDS read travels during this tile's sum
                const float4 wn = pb[64u * (sg + 1u < (unsigned)NS_SEGS ? sg + 1u : sg)];
                const float wk[NS_K] = {w.x, w.y, w.z, w.w};
                s = ordered_tile_sum<NS_K>(s, wk);
                w = wn;
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane == 0) values[v] = s;
}

constexpr int NS_WAVES = NS_PROD + 1;   // vertices (waves) per light block

__device__ __forceinline__ void norm_sum_light(const LatticeDev& L, float* __restrict__ values, unsigned item, bool heavy_elsewhere) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const NormItem it = norm_item(L, item);
    if (!it.ok) return;
    const unsigned idx = it.r * NS_WAVES + (unsigned)wave;
    if (idx >= it.n_vert) return;   // whole wave; no block-wide barrier on this path
    const unsigned v = L.vorder[(unsigned)it.fs0 + idx];
    const unsigned k0 = L.vstart[v], k1 = L.vend[v];
    if (heavy_elsewhere && k1 - k0 >= (unsigned)NS_HEAVY) return;   // a heavy block's
    float s = 0.0f;
    const float* wgt = reinterpret_cast<const float*>(L.csr_pw) + 1;
    for (unsigned kb = k0; kb < k1; kb += 512u) {   // four tiles of 128 per step
        float x[4][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned e = kb + 128u * t + 2u * (unsigned)lane + h;
                const float y = wgt[2 * (size_t)__builtin_elementwise_min(e, k1 - 1u)];
                x[t][h] = e < k1 ? y : 0.0f;   // +0 past the end: the identity of the sum
            }
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (kb + 128u * t < k1) s = ordered_tile_sum<2>(s, x[t]);
    }
    if (lane == 0) values[v] = s;
}

// blocks [0, n_heavy_items): one long list each; the rest: NS_WAVES short lists each.  ONE launch, so that the short
// lists are summed beside the long ones
__global__ void __launch_bounds__(NS_WAVES * 64)
norm_sum_kernel(LatticeDev L, float* __restrict__ values, unsigned n_heavy_items) {
    __shared__ __attribute__((aligned(16))) float buf[NS_RING][NS_BATCH];
    if (L.counters[1]) return;   // hash overflow (flagged): the CSR arrays are incomplete, touch nothing
    if (blockIdx.x < n_heavy_items) norm_sum_heavy(L, values, blockIdx.x, buf);
    else norm_sum_light(L, values, blockIdx.x - n_heavy_items, n_heavy_items != 0u);
}
// the same without the LDS ring (chunks of many frames: no heavy blocks -- see launch_norm_sum)
__global__ void __launch_bounds__(NS_WAVES * 64)
norm_sum_light_kernel(LatticeDev L, float* __restrict__ values) {
    if (L.counters[1]) return;
    norm_sum_light(L, values, blockIdx.x, false);
}

static void launch_norm_sum(const LatticeDev& L, float* values, hipStream_t s) {
    const unsigned nfg = ((unsigned)L.n_frames + (unsigned)L.n_groups - 1u) / (unsigned)L.n_groups;
    const unsigned long long max_mf = std::min<unsigned long long>(((unsigned long long)L.cap_f_mask + 1) / 2 + 1, (unsigned long long)L.m_bound);
    const unsigned n_light = nfg * (unsigned)((max_mf + NS_WAVES - 1) / NS_WAVES) * (unsigned)L.n_groups;
    // A block (and 32 KB of LDS) per long list pays where the launch waits for its longest chains: a frame or two, a
    // cloud.  In a chunk of many frames the lists are summed beside the feature kernels, which need the LDS and the
    // wave slots more (measured at 64 frames: step 11.85 -> 12.2 ms with heavy blocks), and no single chain matters.
    if (L.n_frames <= 4) {
        // a frame of N points has at most 7 N / NS_HEAVY lists that long
        const unsigned long long max_heavy = std::min<unsigned long long>(max_mf, (unsigned long long)(L.d + 1) * L.N / NS_HEAVY + 1);
        const unsigned n_heavy_items = nfg * (unsigned)max_heavy * (unsigned)L.n_groups;
        norm_sum_kernel<<<dim3(n_heavy_items + n_light), dim3(NS_WAVES * 64), 0, s>>>(L, values, n_heavy_items);
    } else {
        norm_sum_light_kernel<<<dim3(n_light), dim3(NS_WAVES * 64), 0, s>>>(L, values);
    }
    RV_LAUNCHED("norm_sum_kernel");
}

// ---------------------------------------------------------------------------------------------
// Scan blocks of the list-major walk: the long lists of launches that wait for their longest chains (a frame or two, a
// cloud).  One vertex per block: two producer waves like those of splat_group_item (128-entry tiles of products into a
// double-buffered LDS tile, register rings of entries and rows; the tiles alternate between the two, so each has two
// steps per tile and its row gathers a lead of six steps -- with one producer the 0.2 us steps outran its rings), and CC
// waves that add one class each with ordered_tile_sum: a wave scan per tile where the serial adder spends 128
// dependent additions per class.  Mode 0 on
// the loop's own contiguous Q * norm only (the FAST producer: padding lanes have weight 0, the rows are finite).
// The same launch carries the regular blocks for the shorter lists (LatticeDev::heavy_from tells them which to leave).
// ---------------------------------------------------------------------------------------------
constexpr unsigned SPLAT_HEAVY = 16384;   // entries from which a list gets a scan block

constexpr int SCAN_PROD = 2;   // producer waves of a scan block (tiles alternate between them)
constexpr int SCAN_NH = 4;     // entries per producer lane and tile: tiles of 256 (the adders' lanes take four addends each)
constexpr int SCAN_TE = 64 * SCAN_NH;

// producer PI of a scan block: tiles PI, PI + 2, ... -- tile k goes into buffer k & 1 between barriers k - 1 and k
template <int CC, int CB, int PI>
__device__ __forceinline__ void splat_scan_producer(const LatticeDev& L, const ValueView& src, unsigned my_k0, unsigned my_k1, unsigned n_steps,
                                                    int c0, float (*prod)[CB][SCAN_TE + 4]) {
    constexpr int NH = SCAN_NH, TE = SCAN_TE, RE_ = 8, RR_ = 4;
    constexpr int LW = CB <= 2 ? 2 : 4;              // floats fetched per row: one load instruction
    static_assert(CB <= 4 && LW <= CC, "class part");
    const int lane = threadIdx.x & 63;
    const int ls = c0 + LW <= CC ? c0 : CC - LW;     // the load stays inside the row; the part starts at x[..][c0 - ls]
    const int xo = c0 - ls;
    float x[RR_][NH][LW];
    float w[RE_][NH];
    unsigned pix[RE_][NH];
#pragma unroll
    for (int r = 0; r < RE_; r++)
#pragma unroll
        for (int h = 0; h < NH; h++) { w[r][h] = 0.f; pix[r][h] = 0u; }
    // loads are unconditional (indices clamped into the list): no divergent branch, counted waits.  `j` counts this
    // producer's own tiles: tile 2 j + PI
    auto load_entries = [&](unsigned j, int slot) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            unsigned k = my_k0 + (2u * j + PI) * (unsigned)TE + (unsigned)lane + 64u * h;
            k = k < my_k1 ? k : my_k1 - 1u;
            const uint2 pw = L.csr_pw[k];
            w[slot][h] = __uint_as_float(pw.y);
            pix[slot][h] = pw.x;
        }
    };
    auto gather_rows = [&](int eslot, int rslot) {
#pragma unroll
        for (int h = 0; h < NH; h++) load_row<LW>(src.base + (size_t)pix[eslot][h] * (unsigned)CC + (unsigned)ls, x[rslot][h]);
    };
#pragma unroll
    for (int i = 0; i < RE_ - 1; i++) load_entries((unsigned)i, i);
#pragma unroll
    for (int i = 0; i < RR_ - 1; i++) gather_rows(i, i);
    // stage j: barriers 2 j and 2 j + 1 of the block (n_steps + 1 in all), this producer's tile before its own one
    auto stage = [&](unsigned j, auto S) -> bool {
        constexpr int s = decltype(S)::value;
        const unsigned t = 2u * j + PI;
        if (2u * j > n_steps) return false;
        if (PI == 1) {
            __syncthreads();                       // barrier 2 j
            if (t > n_steps) return false;
        }
        if (t < n_steps) {
            const unsigned base = my_k0 + t * (unsigned)TE;
            const unsigned n_valid = my_k1 - base < (unsigned)TE ? my_k1 - base : (unsigned)TE;
            float (*pb)[TE + 4] = prod[t & 1u];
#pragma unroll
            for (int h = 0; h < NH; h++) {
                const bool in = (unsigned)lane + 64u * h < n_valid;
                const float wl = in ? w[s][h] : 0.0f;   // +0 past the list: the product is +0, the identity of the sum
#pragma unroll
                for (int c = 0; c < CB; c++) {
                    // class c0 + c sits at x[c + xo], xo = 0 or 1: a select between two registers, no indexed access
                    const float xv = xo ? x[s % RR_][h][c + 1 < LW ? c + 1 : c] : x[s % RR_][h][c];
                    pb[c][lane + 64 * h] = wl * xv;
                }
            }
        }
        load_entries(j + RE_ - 1, (s + RE_ - 1) % RE_);
        gather_rows((s + RR_ - 1) % RE_, (s + RR_ - 1) % RR_);
        __syncthreads();                           // barrier t
        if (PI == 0) {
            if (t + 1u > n_steps) return false;
            __syncthreads();                       // barrier 2 j + 1
        }
        return true;
    };
#define RV_ST(i) if (!stage(j0 + i, std::integral_constant<int, i>())) break;
    for (unsigned j0 = 0;; j0 += RE_) {
        RV_ST(0) RV_ST(1) RV_ST(2) RV_ST(3) RV_ST(4) RV_ST(5) RV_ST(6) RV_ST(7)
    }
#undef RV_ST
}

// classes per scan block: the CC classes of a vertex are split over ceil(CC / CB) blocks -- nine adder waves on one CU
// were bound by instruction issue (0.35 us per 128-entry step); three per block leave a wave per SIMD
template <int CC> struct ScanPart { static constexpr int CB = CC == 9 ? 3 : 4; static constexpr int NP = (CC + CB - 1) / CB; };

template <int CC>
__device__ __forceinline__ void splat_scan_item(const LatticeDev& L, const ValueView& src, float* __restrict__ values, unsigned item,
                                                float (*prod)[ScanPart<CC>::CB][SCAN_TE + 4]) {
    constexpr int TE = SCAN_TE, CB = ScanPart<CC>::CB, NP = ScanPart<CC>::NP;
    static_assert(SCAN_NH == 4, "the adders read float4");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= CB + SCAN_PROD) return;
    const NormItem it = norm_item(L, item / NP);
    if (!it.ok || it.r >= it.n_vert || it.r >= L.scan_ranks) return;
    const unsigned v = L.vorder[(unsigned)it.fs0 + it.r];
    const unsigned my_k0 = L.vstart[v], my_k1 = L.vend[v];
    if (my_k1 - my_k0 < L.heavy_from) return;   // (whole block) a regular block's
    const unsigned n_steps = (my_k1 - my_k0 + (unsigned)TE - 1u) / (unsigned)TE;
    const int c0 = (int)(item % NP) * CB;       // this block's classes: c0 .. min(c0 + CB, CC) - 1
    // every wave passes barriers 0 .. n_steps: tile t is complete at barrier t and is summed between barriers t and t + 1
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(1);
        splat_scan_producer<CC, CB, 0>(L, src, my_k0, my_k1, n_steps, c0, prod);
    } else if (wave == 1) {
        __builtin_amdgcn_s_setprio(1);
        splat_scan_producer<CC, CB, 1>(L, src, my_k0, my_k1, n_steps, c0, prod);
    } else {
        // ---- adder of class c: entries 4 l .. 4 l + 3 of a tile in lane l
        __builtin_amdgcn_s_setprio(3);
        const int ci = wave - SCAN_PROD, c = c0 + ci;
        float acc = 0.0f;
        __syncthreads();
        for (unsigned t = 0; t < n_steps; t++) {
            const float4 q = *reinterpret_cast<const float4*>(&prod[t & 1u][ci][4 * lane]);
            const float wk[4] = {q.x, q.y, q.z, q.w};
            acc = ordered_tile_sum<4>(acc, wk);
            __syncthreads();
        }
        if (lane == 0 && c < CC) values[(size_t)v * CC + c] = acc;
    }
}

// blocks [0, n_scan_items): scan blocks; the rest: regular blocks of GV vertices (their waves beyond GV + 1 leave at once)
template <int CC, int GV>
__global__ void __launch_bounds__((GV + 1) * 64)
splat_mixed_kernel(LatticeDev L, ValueView src, float* __restrict__ values, unsigned n_scan_items) {
    static_assert(ScanPart<CC>::CB + SCAN_PROD <= GV + 1 && 2 * ScanPart<CC>::CB * (SCAN_TE + 4) <= 2 * GV * CC * (64 * 2 + 4), "block size, LDS");
    __shared__ __attribute__((aligned(16))) float prod[2][GV][CC][64 * 2 + 4];
    if (L.counters[1]) return;   // hash overflow (flagged): the CSR arrays are incomplete, touch nothing
    if (blockIdx.x < n_scan_items) {
        splat_scan_item<CC>(L, src, values, blockIdx.x, reinterpret_cast<float (*)[ScanPart<CC>::CB][SCAN_TE + 4]>(&prod[0][0][0][0]));
    } else {
        splat_group_item<0, CC, true, GV, true, 2>(L, src, CC, 0, CC, values, blockIdx.x - n_scan_items, prod);
    }
}

template <int CC, int GV>
static void splat_mixed_launch(const LatticeDev& L0, const ValueView& src, float* values, hipStream_t s) {
    LatticeDev L = L0;
    L.heavy_from = SPLAT_HEAVY;
    const unsigned nfg = ((unsigned)L.n_frames + (unsigned)L.n_groups - 1u) / (unsigned)L.n_groups;
    const unsigned long long max_mf = std::min<unsigned long long>(((unsigned long long)L.cap_f_mask + 1) / 2 + 1, (unsigned long long)L.m_bound);
    // a frame of N points has at most 7 N / SPLAT_HEAVY lists that long; and only as many ranks per frame as give every
    // scan block a CU of its own -- they are there to shorten the launch's longest chains, and cost ~60 instructions
    // per tile and class where the serial adder costs ~10, so a second round of them is a loss (measured: 8 - 16 frames
    // with a scan block for EVERY long list ran 4 - 28 % slower than without any)
    const unsigned long long max_heavy = std::min<unsigned long long>(max_mf, (unsigned long long)(L.d + 1) * L.N / SPLAT_HEAVY + 1);
    const unsigned ranks_by_cus = (unsigned)resident_cu_count() / ((unsigned)ScanPart<CC>::NP * (unsigned)std::max(1, L.n_frames));
    L.scan_ranks = (unsigned)std::min<unsigned long long>(max_heavy, std::max(1u, ranks_by_cus));
    const unsigned n_scan = nfg * L.scan_ranks * (unsigned)L.n_groups * (unsigned)ScanPart<CC>::NP;
    const unsigned n_regular = nfg * (unsigned)((max_mf + GV - 1) / GV) * (unsigned)L.n_groups;
    splat_mixed_kernel<CC, GV><<<dim3(n_scan + n_regular), dim3((GV + 1) * 64), 0, s>>>(L, src, values, n_scan);
    RV_LAUNCHED("splat_mixed_kernel");
}

// vertices per block of the list-major walk for C = 8, 9 (rvseg_schedule.group_vertices: 0 = by the chunk's shape)
static int splat_gv_choice(const LatticeDev& L) {
    if (L.group_vertices == 6 || L.group_vertices == 7) return L.group_vertices;
    // few frames: the launch waits for its longest chains (steps x step time), so the shorter step wins;
    // many frames: the launch is bound by the bytes it moves, the block count only adds overhead
    // (with scan blocks for the longest lists the six-vertex shape wins up to 24 frames: 20 frames 5.12 vs 5.48 ms per
    // step, 24 frames 5.49 vs 5.98; 32 frames 7.15 vs 7.37 -- but there the resident bands take 7.03)
    return L.n_frames <= 24 ? 6 : 7;
}

template <int CC>
static bool splat_resident_launch(const LatticeDev& L, const SplatResidentDev& R, const float* src, float* values, int slot, hipStream_t s);

void launch_splat(const LatticeDev& L, const ValueView& src, int C, int mode, float* values, hipStream_t s, bool own_q,
                  const SplatResidentDev* resident, int slot) {
    if (mode == 2) {
        if (L.ordered_sum_scan) launch_norm_sum(L, values, s);
        else splat_group_launch<2, 1>(L, src, 1, 0, 1, values, s);
        return;
    }
    const bool contig = src.frame_stride == (size_t)L.N * (size_t)C && src.layer_off == 0;
    if (mode == 0 && own_q && contig && (C == 9 || C == 8) && resident) {
        // resident band schedule (the kernel walks the lists the list-major way itself should the planner have given up)
        const bool ran = C == 9 ? splat_resident_launch<9>(L, *resident, src.base, values, slot, s)
                                : splat_resident_launch<8>(L, *resident, src.base, values, slot, s);
        if (ran) return;
    }
    if (mode == 0 && own_q && contig && (C == 9 || C == 8)) {
        // the mean-field loop's own input (Q * norm written by the previous update): the fast producer, and the
        // block shape chosen for the chunk
        const int gv = splat_gv_choice(L);
        // six vertices per block go with 128-entry tiles: both serve launches whose time is their longest chain
        // launches of <= 16 frames wait for their longest chains: those get scan blocks, as many as there are CUs
        if (gv == 6 && L.ordered_sum_scan) {
            if (C == 9) splat_mixed_launch<9, 6>(L, src, values, s);
            else splat_mixed_launch<8, 6>(L, src, values, s);
            return;
        }
        if (C == 9) {
            if (gv == 6) splat_group_launch_g<0, 9, 6, true, 2>(L, src, C, 0, 9, values, s);
            else splat_group_launch_g<0, 9, 7, true>(L, src, C, 0, 9, values, s);
        } else {
            if (gv == 6) splat_group_launch_g<0, 8, 6, true, 2>(L, src, C, 0, 8, values, s);
            else splat_group_launch_g<0, 8, 8, true>(L, src, C, 0, 8, values, s);
        }
        return;
    }
    for (int c0 = 0; c0 < C; c0 += 16) {
        const int n = C - c0 < 16 ? C - c0 : 16;
        if (mode == 0) splat_group_pass<0>(L, src, C, c0, n, values, s);
        else splat_group_pass<1>(L, src, C, c0, n, values, s);
    }
}

// ---------------------------------------------------------------------------------------------
// Resident band schedule of the ordered splat (SplatResidentDev, rvseg_crf.h).
// ---------------------------------------------------------------------------------------------
// piece of frame-local vertex lv inside band b: [k0, k1) of its list
__device__ __forceinline__ void resident_piece(const LatticeDev& L, int band_wb, int f0, int Mf, int lv, int b, unsigned& k0, unsigned& k1) {
    const unsigned* fb = L.bh + (size_t)L.wbpf * f0 + lv;
    const int w0 = b * band_wb, w1 = w0 + band_wb;
    k0 = fb[(size_t)w0 * Mf];
    k1 = w1 < L.wbpf ? fb[(size_t)w1 * Mf] : L.vend[f0 + lv];
}

// One block per frame: deals the frame's vertices to the B blocks and numbers the tiles.  A block's tile count is
// sum over bands of max(chunks of its longest vertex there, all its chunks there / 7), so the heavy vertices are placed
// one at a time, longest first, each into the block whose count ends up smallest -- evaluated exactly, band by band,
// by the whole workgroup (`profiles/analysis`: 1 111 tiles for the fullest block of a bench frame, which is the chain
// of its heaviest vertex, against 1 346 when the blocks are balanced by chunk totals).  The many short vertices
// that follow go to the block with the fewest tiles so far, priced at chunks / 7.
constexpr int RES_PLAN_THREADS = 1024;
constexpr int RES_HEAVY_MAX = 160;   // vertices placed exactly at most (numpy on a bench frame: 96 reach the result of 200; other frames need more)
// dynamic LDS: T[B][nb] words, then the heavy table [heavy_cap][nb] bytes
__global__ void __launch_bounds__(RES_PLAN_THREADS)
resident_plan_kernel(LatticeDev L, SplatResidentDev R, int heavy_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned plan_lds[];
    __shared__ unsigned chv[RES_MAX_VERTS];
    __shared__ unsigned short own[RES_MAXB][RES_MAX_OWNV];
    __shared__ unsigned nown[RES_MAXB], nheavy[RES_MAXB];
    __shared__ unsigned short lvo[RES_MAX_VERTS];   // the frame's vertices, longest list first (`vorder`, frame-local)
    __shared__ unsigned cur[RES_MAXB];
    __shared__ unsigned blk0[RES_MAXB + 1];
    __shared__ int bad, choice;
    if (L.counters[1]) return;
    const int frame = blockIdx.x;
    const int Mtot = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int f0 = L.fstart[frame] < Mtot ? L.fstart[frame] : Mtot;
    const int f1 = L.fstart[frame + 1] < Mtot ? L.fstart[frame + 1] : Mtot;
    const int Mf = f1 - f0, nb = R.n_bands, B = R.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned clog = (unsigned)R.chunk_log2, cmask = (1u << clog) - 1u;   // entries per chunk: 64 or 128
    unsigned* T = plan_lds;                                               // [B][nb]: greedy: (sum of chunks) << 8 | longest; then tiles, then their scan
    unsigned char* hc = reinterpret_cast<unsigned char*>(plan_lds + B * nb);   // [heavy_cap][nb]
    if (tid == 0) { bad = 0; choice = 0; }
    if (tid < RES_MAXB) { nown[tid] = 0; nheavy[tid] = 0; cur[tid] = 0; }
    for (int i = tid; i < B * nb; i += RES_PLAN_THREADS) T[i] = 0;
    for (int lv = tid; lv < RES_MAX_VERTS; lv += RES_PLAN_THREADS) chv[lv] = 0;
    __syncthreads();
    if (Mf > RES_MAX_VERTS) {
        if (tid == 0) atomicAdd(&R.flags[0], 1);
        return;
    }
    for (int k = tid; k < Mf; k += RES_PLAN_THREADS) {
        const int lv = (int)L.vorder[f0 + k] - f0;
        lvo[k] = (unsigned short)(lv < 0 ? 0 : (lv >= Mf ? Mf - 1 : lv));
    }
    // 64-entry chunks of every vertex, summed over the bands: one (vertex, band) piece per thread and step
    for (int i = tid; i < Mf * nb; i += RES_PLAN_THREADS) {
        const int b = i / Mf, lv = i - b * Mf;   // consecutive threads: consecutive vertices of one band (the table's row)
        unsigned k0, k1;
        resident_piece(L, R.band_wb, f0, Mf, lv, b, k0, k1);
        const unsigned ch = (k1 - k0 + cmask) >> clog;
        if (ch) atomicAdd(&chv[lv], ch);
    }
    __syncthreads();
    // heavy = at least 12 chunks; `vorder` is sorted by length, so they are (about) a prefix of it: its length is counted
    if (tid < heavy_cap && tid < Mf && chv[lvo[tid]] >= (12u >> (clog - 6))) atomicAdd(&choice, 1);
    __syncthreads();
    const int n_heavy = choice;
    __syncthreads();
    // their chunks per band, once, in LDS: the placement loop below touches no global memory
    for (int i = tid; i < n_heavy * nb; i += RES_PLAN_THREADS) {
        const int k = i / nb, b = i - k * nb;
        unsigned k0, k1;
        resident_piece(L, R.band_wb, f0, Mf, lvo[k], b, k0, k1);
        hc[i] = (unsigned char)((k1 - k0 + cmask) >> clog);
    }
    __syncthreads();
    // ---- the heavy vertices, exactly, by ONE wave (lane = band): the loop is a chain of up to 160 dependent decisions,
    // and with the whole workgroup on it every decision cost three block-wide barriers and an LDS round trip through
    // thread 0 (0.45 ms per launch, 88 % of the wave cycles waiting); a single wave needs no barrier at all, the tile
    // counts of the B candidate blocks are B independent DPP reductions
    if (wave == 0) {
        for (int k = 0; k < n_heavy; k++) {
            const unsigned char* cb = hc + (size_t)k * nb;
            int best = -1; unsigned best_t = 0, best_inc = 0;
            for (int j = 0; j < B; j++) {
                unsigned sum = 0;
                for (int b = lane; b < nb; b += 64) {
                    const unsigned pk = T[j * nb + b], c = cb[b];
                    const unsigned mx = (pk & 255u) > c ? (pk & 255u) : c, t7 = ((pk >> 8) + c + 6u) / 7u;
                    sum += mx > t7 ? mx : t7;
                }
                const unsigned t = wave_total_u32(sum);
                if (nown[j] >= (unsigned)RES_MAX_OWNV) continue;
                const unsigned inc = t - cur[j];
                if (best < 0 || t < best_t || (t == best_t && inc < best_inc)) { best = j; best_t = t; best_inc = inc; }
            }
            if (best < 0) { if (lane == 0) bad = 1; break; }
            for (int b = lane; b < nb; b += 64) {
                const unsigned pk = T[best * nb + b], c = cb[b];
                T[best * nb + b] = (((pk >> 8) + c) << 8) | ((pk & 255u) > c ? (pk & 255u) : c);
            }
            if (lane == 0) { cur[best] = best_t; own[best][nown[best]] = lvo[k]; nown[best] = nown[best] + 1u; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // one wave: LDS in program order; keeps the compiler from caching T / cur / nown
        }
    }
    __syncthreads();
    if (bad) { if (tid == 0) atomicAdd(&R.flags[0], 1); return; }
    // ---- the rest: fewest tiles so far, a vertex priced at chunks / 7 (sevenths of a tile).  One wave, lane j = block j:
    // the running prices live in registers and the cheapest block is a DPP minimum (until round 3 thread 0 walked a
    // private array -- scratch memory -- through this loop: 0.6 ms of the deep scene's 2.8 ms planner, 2 159 light vertices)
    if (wave == 0) {
        const bool blk = lane < B;
        unsigned price = blk ? cur[lane] * 7u : 0xFFFFFFFFu;
        unsigned cnt = blk ? nown[lane] : 0u;
        if (blk) nheavy[lane] = cnt;
        for (int k = n_heavy; k < Mf; k++) {
            const int lv = lvo[k];
            const bool room = blk && cnt < (unsigned)RES_MAX_OWNV;
            const unsigned cand = room ? price : 0xFFFFFFFFu;
            const unsigned best_price = wave_min_u32(cand);
            const unsigned long long at = __ballot(room && cand == best_price);
            if (!at || lv < 0 || lv >= Mf) { if (lane == 0) bad = 1; break; }
            const int best = __ffsll((long long)at) - 1;      // the lowest block among equals, as the serial loop chose
            if (lane == best) {
                price += chv[lv] + 1u;
                own[best][cnt] = (unsigned short)lv;
                cnt++;
            }
        }
        if (blk) nown[lane] = cnt;
    }
    __syncthreads();
    if (bad) { if (tid == 0) atomicAdd(&R.flags[0], 1); return; }
    // tiles of every (block, band): the light vertices' chunks join the packed sums, a wave per (block, band)
    for (int idx = wave; idx < B * nb; idx += RES_PLAN_THREADS / 64) {
        const int j = idx / nb, b = idx - j * nb;
        unsigned sum = 0, mx = 0;
        for (unsigned u = nheavy[j] + lane; u < nown[j]; u += 64) {
            unsigned k0, k1;
            resident_piece(L, R.band_wb, f0, Mf, own[j][u], b, k0, k1);
            const unsigned ch = (k1 - k0 + cmask) >> clog;
            sum += ch;
            mx = ch > mx ? ch : mx;
        }
        for (int o = 32; o > 0; o >>= 1) { sum += __shfl_xor(sum, o, 64); const unsigned m2 = __shfl_xor(mx, o, 64); mx = m2 > mx ? m2 : mx; }
        if (lane == 0) {
            const unsigned pk = T[idx];
            sum += pk >> 8; mx = (pk & 255u) > mx ? (pk & 255u) : mx;
            const unsigned t = (sum + 6u) / 7u;
            T[idx] = t > mx ? t : mx;
        }
    }
    __syncthreads();
    if (tid < B) {
        unsigned run = 0;
        for (int b = 0; b < nb; b++) { const unsigned t = T[tid * nb + b]; T[tid * nb + b] = run; run += t; }
        blk0[tid + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {
        blk0[0] = 0;
        for (int j = 0; j < B; j++) blk0[j + 1] += blk0[j];
        if (blk0[B] > R.cap_tiles) bad = 1;
    }
    __syncthreads();
    if (bad) { if (tid == 0) atomicAdd(&R.flags[0], 1); return; }
    unsigned* jb = R.jb_tile + (size_t)frame * RES_MAXB * (nb + 1);
    for (int idx = tid; idx < B * (nb + 1); idx += RES_PLAN_THREADS) {
        const int j = idx / (nb + 1), b = idx - j * (nb + 1);
        jb[idx] = b < nb ? blk0[j] + T[j * nb + b] : blk0[j + 1];
    }
    for (int j = tid; j <= B; j += RES_PLAN_THREADS) R.blk_tile0[(size_t)frame * (RES_MAXB + 1) + j] = blk0[j];
    for (int j = tid; j < B; j += RES_PLAN_THREADS) R.blk_nown[(size_t)frame * RES_MAXB + j] = nown[j];
    for (int idx = tid; idx < B * RES_MAX_OWNV; idx += RES_PLAN_THREADS) {
        const int j = idx / RES_MAX_OWNV, u = idx - j * RES_MAX_OWNV;
        R.blk_verts[((size_t)frame * RES_MAXB + j) * RES_MAX_OWNV + u] = (unsigned)u < nown[j] ? own[j][u] : (unsigned short)0;
    }
}

__global__ void resident_seal_kernel(LatticeDev L, SplatResidentDev R) {
    R.flags[1] = (R.flags[0] == 0 && L.counters[1] == 0) ? 1 : 0;
}

// One wave per (block, band): packs the chunks of the block's vertices into the band's T tiles x 7 slots by the
// wrap-around rule.  The cells are numbered slot after slot; a vertex takes the next `chunks` cells (a prefix sum over
// the block's vertices, kept in LDS), and when its cells run over the end of a slot it continues at the top of the next
// one.  Because a vertex has at most T chunks the two parts never share a tile, and its chunks are numbered by tile,
// so they are summed in list order whatever slot they sit in.  The cells are then written lane = cell (coalesced
// stores), each lane finding its vertex by bisection of the prefix sums.
__global__ void __launch_bounds__(256)
resident_fill_kernel(LatticeDev L, SplatResidentDev R) {
    __shared__ unsigned s_pre[4][RES_MAX_OWNV + 1], s_k0[4][RES_MAX_OWNV], s_len[4][RES_MAX_OWNV];
    if (!R.flags[1]) return;
    const int frame = blockIdx.y;
    const int nb = R.n_bands, B = R.B;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int idx = blockIdx.x * 4 + wv;
    if (idx >= B * nb) return;
    const int j = idx / nb, b = idx - j * nb;
    const int Mtot = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int f0 = L.fstart[frame] < Mtot ? L.fstart[frame] : Mtot;
    const int f1 = L.fstart[frame + 1] < Mtot ? L.fstart[frame + 1] : Mtot;
    const int Mf = f1 - f0;
    const unsigned* jb = R.jb_tile + ((size_t)frame * RES_MAXB + j) * (nb + 1);
    const unsigned t0 = jb[b], T = jb[b + 1] - t0;
    if (!T) return;
    const unsigned n_own = R.blk_nown[(size_t)frame * RES_MAXB + j];
    const unsigned short* verts = R.blk_verts + ((size_t)frame * RES_MAXB + j) * RES_MAX_OWNV;
    const unsigned base = (unsigned)frame * (unsigned)L.N * 7u;   // the frame's first entry (csr_scan_kernel)
    unsigned* pre = s_pre[wv]; unsigned* k0s = s_k0[wv]; unsigned* lens = s_len[wv];
    const unsigned clog = (unsigned)R.chunk_log2, cmask = (1u << clog) - 1u, CH = 1u << clog;
    unsigned run = 0, hmax = 0;
    unsigned any_k0 = 0xFFFFFFFFu;    // an entry of this band: what an unused cell points at (weight 0)
    for (unsigned u0 = 0; u0 < n_own; u0 += 64) {
        const unsigned u = u0 + lane;
        unsigned k0 = 0, len = 0;
        if (u < n_own) { unsigned k1; resident_piece(L, R.band_wb, f0, Mf, verts[u], b, k0, k1); len = k1 - k0; }
        const unsigned ch = (len + cmask) >> clog;
        unsigned incl = ch;
        for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
        if (u < n_own) { pre[u] = run + incl - ch; k0s[u] = k0; lens[u] = len; }
        hmax = len > hmax ? len : hmax;
        const unsigned long long has = __ballot(ch != 0u);
        if (any_k0 == 0xFFFFFFFFu && has) any_k0 = __shfl(k0, __ffsll((long long)has) - 1, 64);
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) pre[n_own] = run;
    for (int o = 32; o > 0; o >>= 1) { const unsigned m2 = __shfl_xor(hmax, o, 64); hmax = m2 > hmax ? m2 : hmax; }
    hmax = hmax > CH ? CH : hmax;   // height of the band's tiles for the adder (16 / 32 / 64 / 128 adds)
    __builtin_amdgcn_wave_barrier();   // (one wave: its LDS writes are in order before the reads below)
    unsigned* info = R.tinfo + (size_t)frame * R.cap_tiles + t0;
    for (unsigned t = lane; t < T; t += 64) info[t] = ((unsigned)b << 16) | hmax;
    for (unsigned q = lane; q < 7u * T; q += 64) {
        unsigned u = n_own, at = any_k0, n = 0;
        if (q < run) {
            unsigned lo = 0, hi = n_own;   // the vertex with pre[u] <= q < pre[u + 1]
            while (hi - lo > 1u) { const unsigned mid = (lo + hi) >> 1; if (pre[mid] <= q) lo = mid; else hi = mid; }
            u = lo;
            const unsigned p = pre[u], ch = pre[u + 1] - p, i = q - p;
            const unsigned room = T - p % T;                       // cells left in the slot where the vertex starts
            const unsigned wrap = ch > room ? ch - room : 0u;      // chunks at the top of the next slot: the FIRST ones (lower tiles)
            const unsigned chunk = i < room ? wrap + i : i - room;
            const unsigned len = lens[u];
            n = len - CH * chunk < CH ? len - CH * chunk : CH;
            at = k0s[u] + CH * chunk;
        }
        const unsigned s = q / T, t = q - s * T;
        R.tdesc[((size_t)frame * 7 + s) * R.cap_tiles + t0 + t] = ((at - base) << 8) | n;
        R.tvl[((size_t)frame * 7 + s) * R.cap_tiles + t0 + t] = (unsigned short)u;
    }
}

void launch_resident_plan(const LatticeDev& L, const SplatResidentDev& R, hipStream_t s) {
    (void)hipMemsetAsync(R.flags, 0, 2 * sizeof(int), s);
    // dynamic LDS of the planner: tile sums [B][n_bands] words + the heavy table, inside the 64 KB a block gets by default
    const size_t t_bytes = (size_t)R.B * R.n_bands * 4;
    const size_t fixed = (size_t)RES_MAX_VERTS * 4 + (size_t)RES_MAX_VERTS * 2 + RES_MAXB * RES_MAX_OWNV * 2 + 1024;
    int heavy_cap = RES_HEAVY_MAX;
    while (heavy_cap > 8 && fixed + t_bytes + (size_t)heavy_cap * R.n_bands > 60000) heavy_cap -= 8;
    const size_t dyn = t_bytes + (size_t)heavy_cap * R.n_bands + 16;
    resident_plan_kernel<<<dim3((unsigned)L.n_frames), dim3(RES_PLAN_THREADS), dyn, s>>>(L, R, heavy_cap);
    resident_seal_kernel<<<dim3(1), dim3(1), 0, s>>>(L, R);
    resident_fill_kernel<<<dim3((unsigned)((R.B * R.n_bands + 3) / 4), (unsigned)L.n_frames), dim3(256), 0, s>>>(L, R);
    RV_LAUNCHED("resident_plan_kernel / resident_seal_kernel / resident_fill_kernel");
}

// The splat over the schedule.  Block (frame, j): 7 producer waves + the adder, as in splat_group_kernel; the tile
// list replaces the walk down seven whole lists.  Producer i reads the descriptor stream of slot i 64 tiles at a time
// into one register (lane = tile), refreshed once per unrolled group of stages with an unconditional load, so the
// stage bodies index it with compile-time lanes and the loads of entries (RE - 1 tiles ahead) and rows (RR - 1 tiles
// ahead) run on across vertices and bands without a bubble.  The adder's lane (i, c) serves whatever vertex slot i
// holds in the tile: the running sums of the block's vertices sit in LDS, and a lane swaps its sum when the slot's
// vertex changes (store, then load: one wave, so a chain that moves to another slot in the next tile is handed over
// in order).  Pacing: at the first tile of a band the adder publishes the band and holds the block (by arriving late
// at the tile's barrier) while any block of its frame is more than `window` bands behind; the check uses the progress
// words fetched one band earlier, so it costs no round trip unless it waits, and the wait is bounded -- blocks never
// depend on each other for their results.
template <int CC, int RE, int RR, int TH>   // TH: entries per slot and tile (64 or 128)
__global__ void __launch_bounds__(512)
splat_resident_kernel(LatticeDev L, SplatResidentDev R, ValueView srcv, float* __restrict__ values, unsigned tag, int slot, unsigned n_items) {
    constexpr int G = 7, C = CC;
    static_assert(CC * G <= 64 && (RE == 16 || RE == 8) && RE % RR == 0 && (TH == 64 || TH == 128), "block shape");
    constexpr int NH = TH / 64;        // entries per lane and tile
    constexpr int ROW = TH + 4;        // product row: 16-B aligned, 4-bank skew
    // dynamic LDS (more than 64 KB for TH = 128): products [2][G][CC][ROW], running sums, the adder's table
    extern __shared__ __attribute__((aligned(16))) float res_lds[];
    float (*prod)[G][CC][ROW] = reinterpret_cast<float (*)[G][CC][ROW]>(res_lds);
    float* accs = res_lds + 2 * G * CC * ROW;
    unsigned (*ainfo)[64][8] = reinterpret_cast<unsigned (*)[64][8]>(accs + (RES_MAX_OWNV + 1) * CC);   // per tile and slot: vertex | height << 10 | band << 18
    if (L.counters[1]) return;
    if (!R.flags[1]) {
        // the planner gave up on some frame (more vertices or tiles than its tables hold): this grid walks the lists
        // the list-major way, G vertices per item
        for (unsigned item = blockIdx.x; item < n_items; item += gridDim.x) {
            splat_group_item<0, CC, true, G, true>(L, srcv, CC, 0, CC, values, item, reinterpret_cast<float (*)[G][CC][68]>(res_lds));
            __syncthreads();
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // block b -> XCD b % NG; all B blocks of a frame on one XCD (frame f lives on XCD f % NG)
    const unsigned NG = (unsigned)L.n_groups, B = (unsigned)R.B;
    const unsigned x = blockIdx.x % NG, jj = blockIdx.x / NG;
    const unsigned j = jj % B, frame = (jj / B) * NG + x;
    if (frame >= (unsigned)L.n_frames) return;
    const int Mtot = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int f0 = L.fstart[frame] < Mtot ? L.fstart[frame] : Mtot;
    const unsigned tb = R.blk_tile0[(size_t)frame * (RES_MAXB + 1) + j], te = R.blk_tile0[(size_t)frame * (RES_MAXB + 1) + j + 1];
    const unsigned n_t = te - tb;
    const unsigned n_own = R.blk_nown[(size_t)frame * RES_MAXB + j];
    for (unsigned e = threadIdx.x; e < (n_own + 1u) * CC; e += 512) accs[e] = 0.0f;
    unsigned* prog = R.prog + ((size_t)slot * L.n_frames + frame) * RES_MAXB;
    const unsigned long long t_start = R.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long c_start = R.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long t_spin = 0;
    if (wave == G) __builtin_amdgcn_s_setprio(3);
    else if (n_t > 256u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    if (wave < G) {
        if (n_t) {
            const unsigned* D = R.tdesc + ((size_t)frame * 7 + wave) * R.cap_tiles + tb;
            const unsigned base = frame * (unsigned)L.N * 7u;   // the frame's first entry (csr_scan_kernel: every point has d + 1 entries)
            const unsigned klast = base + (unsigned)L.N * 7u - 1u;
            float xr[RR][NH][CC];
            float w[RE][NH];
            unsigned pix[RE][NH];
#pragma unroll
            for (int r = 0; r < RE; r++)
#pragma unroll
                for (int h = 0; h < NH; h++) { w[r][h] = 0.f; pix[r][h] = 0u; }
            auto load_desc = [&](unsigned first) -> unsigned {
                const unsigned t = first + (unsigned)lane;
                return D[t < n_t ? t : n_t - 1u];
            };
            unsigned dcur = load_desc(0u), dnxt = dcur;
            auto load_entries = [&](unsigned d, int slot_e) {
                const unsigned n = d & 255u, last = n ? n - 1u : 0u;
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const unsigned i = (unsigned)lane + 64u * h;
                    unsigned k = base + (d >> 8) + (i < last ? i : last);
                    k = k < klast ? k : klast;
                    const uint2 e = L.csr_pw[k];
                    w[slot_e][h] = __uint_as_float(e.y);
                    pix[slot_e][h] = e.x;
                }
            };
            // (the rows come through a plain pointer: a const __restrict__ kernel argument makes the gathers invariant loads,
            // which the compiler then sinks to their use -- the whole prefetch distance lost)
            auto gather_rows = [&](int eslot, int rslot) {
#pragma unroll
                for (int h = 0; h < NH; h++) load_row<CC>(srcv.base + (size_t)pix[eslot][h] * (unsigned)C, xr[rslot][h]);
            };
#pragma unroll
            for (int i = 0; i < RE - 1; i++) load_entries(__builtin_amdgcn_readlane(dcur, i), i);
#pragma unroll
            for (int i = 0; i < RR - 1; i++) gather_rows(i, i);
            auto stage = [&](unsigned t, auto S) -> bool {
                constexpr int s = decltype(S)::value;
                if (t >= n_t) return false;
                const unsigned n = __builtin_amdgcn_readlane(dcur, s) & 255u;
                float (*pb)[ROW] = prod[t & 1u][wave];
#pragma unroll
                for (int h = 0; h < NH; h++) {
                    const float wl = (unsigned)lane + 64u * h < n ? w[s][h] : 0.0f;
#pragma unroll
                    for (int c = 0; c < CC; c++) pb[c][lane + 64 * h] = wl * xr[s % RR][h][c];   // +0 past the chunk (rows are finite)
                }
                load_entries(__builtin_amdgcn_readlane(dcur, s + RE - 1), (s + RE - 1) % RE);
                gather_rows((s + RR - 1) % RE, (s + RR - 1) % RR);
                __syncthreads();
                return true;
            };
#define RV_ST(i) if (!stage(t0 + i, std::integral_constant<int, i>())) break;
            for (unsigned t0 = 0;; t0 += RE) {
                dnxt = load_desc(t0 + RE);   // lanes 0 .. 2 RE - 2 of the next group's register
                RV_ST(0) RV_ST(1) RV_ST(2) RV_ST(3) RV_ST(4) RV_ST(5) RV_ST(6) RV_ST(7)
                if constexpr (RE == 16) {
                    RV_ST(8) RV_ST(9) RV_ST(10) RV_ST(11) RV_ST(12) RV_ST(13) RV_ST(14) RV_ST(15)
                }
                dcur = dnxt;
            }
#undef RV_ST
        }
        __syncthreads();
    } else {
        const bool live = lane < G * CC;
        const int gi = live ? lane / CC : 0;
        const int c = live ? lane % CC : 0;
        const unsigned* I = R.tinfo + (size_t)frame * R.cap_tiles + tb;
        const unsigned short* V = R.tvl + (size_t)frame * 7 * R.cap_tiles + tb;
        // What a tile needs besides its products -- height and band (uniform) and the vertex of the lane's slot -- comes
        // from an LDS table of 2 x 64 tiles x 8 words {info, vertex of slot 0 .. 6}.  The adder refills it 64 tiles at a
        // time: lane = tile, eight global loads that stay in flight for 64 tiles and are stored when their half of the
        // table comes up.  Per tile that leaves two LDS reads, issued a tile ahead: the loop's critical path is the 64
        // dependent adds and the barrier, nothing else may wait on it (no load per tile, no register shuffling).
        unsigned pend[8];
        auto batch_load = [&](unsigned first) {
            const unsigned t = first + (unsigned)lane;
            const unsigned tc = n_t ? (t < n_t ? t : n_t - 1u) : 0u;
            pend[0] = n_t ? I[tc] : 0u;
#pragma unroll
            for (int i = 0; i < G; i++) pend[1 + i] = n_t ? (unsigned)V[(size_t)i * R.cap_tiles + tc] : n_own;
        };
        auto batch_store = [&](unsigned first) {
            const bool in = first + (unsigned)lane < n_t;
            unsigned* row = &ainfo[(first >> 6) & 1u][lane][0];
            // one word per slot: vertex | height << 10 | band << 18 (a lane reads ONE word per tile)
            const unsigned hb = ((pend[0] & 255u) << 10) | ((pend[0] >> 16) << 18);
#pragma unroll
            for (int i = 0; i < G; i++) row[i] = (in ? pend[1 + i] : n_own) | hb;
        };
        auto poll = [&]() -> unsigned { return __hip_atomic_load(&prog[(unsigned)lane < B ? lane : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        const unsigned tagv = tag << 16;
        if (lane == 0) __hip_atomic_store(&prog[j], n_t ? tagv : (tagv | 0xFFFFu), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        batch_load(0u);
        batch_store(0u);
        batch_load(64u);
        unsigned polled = R.window >= 0 ? poll() : 0u;
        unsigned cur_vl = n_own, cur_band = 0u;
        bool pacing = R.window >= 0;
        float acc = 0.0f;
        unsigned w_n = n_own;
        // issues the table read of tile t (used after the next barrier)
        auto prep = [&](unsigned t) {
            if (t >= n_t) return;
            if ((t & 63u) == 0u && t) { batch_store(t); batch_load(t + 64u); }
            w_n = ainfo[(t >> 6) & 1u][t & 63u][gi];
        };
        auto pace = [&](unsigned band) {
            cur_band = band;
            if (lane == 0) __hip_atomic_store(&prog[j], tagv | band, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!pacing) return;
            auto behind = [&](unsigned pv) -> bool {
                // a word of another launch: that block has not started yet
                return (unsigned)lane < B && ((pv >> 16) != tag || (pv & 0xFFFFu) + (unsigned)R.window < band);
            };
            if (__ballot(behind(polled))) {
                const unsigned long long ts = R.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
                unsigned spins = 0;
                for (;;) {
                    polled = poll();
                    if (!__ballot(behind(polled))) break;
                    if (++spins > 256u) { pacing = false; break; }   // some block is not running: stop waiting for good
                    __builtin_amdgcn_s_sleep(8);
                }
                if (R.trace) t_spin += __builtin_amdgcn_s_memrealtime() - ts;
            }
            polled = poll();   // consumed at the next band
        };
        prep(0u);
        if (n_t) { const unsigned b0 = __builtin_amdgcn_readfirstlane(w_n) >> 18; if (b0 != cur_band) pace(b0); }
        __syncthreads();
        for (unsigned t = 0; t < n_t; t++) {
            const unsigned nmax = (__builtin_amdgcn_readfirstlane(w_n) >> 10) & 255u;
            const unsigned v_n = live ? (w_n & 1023u) : n_own;
            // the slot's vertex changed: park the sum, fetch the other one.  Store before load, one wave: a chain that
            // moved here from another slot of the previous tile is handed over in order.
            if (v_n != cur_vl) { accs[cur_vl * CC + c] = acc; acc = accs[v_n * CC + c]; cur_vl = v_n; }
            const float* pr = prod[t & 1u][gi][c];
            bool done = false;
            if constexpr (TH == 128) {
                if (nmax > 64u) {
                    float4 q[16], q2[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) q[i] = reinterpret_cast<const float4*>(pr)[i];
#pragma unroll
                    for (int i = 0; i < 16; i++) q2[i] = reinterpret_cast<const float4*>(pr)[16 + i];
                    prep(t + 1u);
#pragma unroll
                    for (int i = 0; i < 16; i++) { acc += q[i].x; acc += q[i].y; acc += q[i].z; acc += q[i].w; }
#pragma unroll
                    for (int i = 0; i < 16; i++) { acc += q2[i].x; acc += q2[i].y; acc += q2[i].z; acc += q2[i].w; }
                    done = true;
                }
            }
            if (done) {
            } else if (nmax > 32u) {
                float4 q[16];
#pragma unroll
                for (int i = 0; i < 16; i++) q[i] = reinterpret_cast<const float4*>(pr)[i];
                prep(t + 1u);
#pragma unroll
                for (int i = 0; i < 16; i++) { acc += q[i].x; acc += q[i].y; acc += q[i].z; acc += q[i].w; }
            } else if (nmax > 16u) {
                float4 q[8];
#pragma unroll
                for (int i = 0; i < 8; i++) q[i] = reinterpret_cast<const float4*>(pr)[i];
                prep(t + 1u);
#pragma unroll
                for (int i = 0; i < 8; i++) { acc += q[i].x; acc += q[i].y; acc += q[i].z; acc += q[i].w; }
            } else {
                float4 q[4];
#pragma unroll
                for (int i = 0; i < 4; i++) q[i] = reinterpret_cast<const float4*>(pr)[i];
                prep(t + 1u);
#pragma unroll
                for (int i = 0; i < 4; i++) { acc += q[i].x; acc += q[i].y; acc += q[i].z; acc += q[i].w; }
            }
            if (t + 1u < n_t) { const unsigned bn = __builtin_amdgcn_readfirstlane(w_n) >> 18; if (bn != cur_band) pace(bn); }
            __syncthreads();
        }
        accs[cur_vl * CC + c] = acc;
        if (lane == 0) __hip_atomic_store(&prog[j], tagv | 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (R.trace && lane == 0) {
            unsigned long long* tr = R.trace + ((size_t)frame * RES_MAXB + j) * 8;
            tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime(); tr[2] = ((unsigned long long)n_own << 32) | n_t; tr[3] = t_spin;
            tr[4] = __builtin_amdgcn_s_memtime() - c_start;   // shader clocks
        }
    }
    __syncthreads();
    // every vertex of the frame belongs to one block: all sums are written, empty lists as 0
    const unsigned short* verts = R.blk_verts + ((size_t)frame * RES_MAXB + j) * RES_MAX_OWNV;
    for (unsigned e = threadIdx.x; e < n_own * CC; e += 512) values[((size_t)f0 + verts[e / CC]) * C + e % CC] = accs[e];
}

static std::atomic<unsigned> g_resident_tag{0};   // launch tags of the pacing words (any two concurrent launches just need different ones)

static size_t resident_lds_bytes(int CC, int TH) {
    return ((size_t)2 * 7 * CC * (TH + 4) + (size_t)(RES_MAX_OWNV + 1) * CC) * sizeof(float) + 2 * 64 * 8 * sizeof(unsigned);
}

// more than 64 KB of dynamic LDS has to be asked for, once per instantiation AND device (one process may drive
// several GPUs, one context each: rvseg_comm.cpp)
constexpr int RES_MAX_DEVICES = 64;
template <int CC, int TH>
static bool resident_setup() {
    static std::atomic<int> state[RES_MAX_DEVICES];   // 0 = not tried, 1 = ok, 2 = refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RES_MAX_DEVICES) return false;
    int st = state[dev].load();
    if (st == 0) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(splat_resident_kernel<CC, 8, 4, TH>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)resident_lds_bytes(CC, TH));
        if (e != hipSuccess) (void)hipGetLastError();   // refused: the caller walks the lists the list-major way
        st = e == hipSuccess ? 1 : 2;
        state[dev].store(st);
    }
    return st == 1;
}

// blocks of the resident kernel that fit on the current device at once, and its CU count (cached per device)
static bool resident_device_info(int chunk, int* capacity, int* cus) {
    static std::atomic<int> cap[RES_MAX_DEVICES][2], ncu[RES_MAX_DEVICES];   // 0 = not known yet, -1 = failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RES_MAX_DEVICES) return false;
    const int which = chunk == 128 ? 1 : 0;
    if (cap[dev][which].load() == 0) {
        int per_cu = 0;
        hipDeviceProp_t pr;
        hipError_t e = hipGetDeviceProperties(&pr, dev);
        if (e == hipSuccess) {
            if (which) e = resident_setup<9, 128>() ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, splat_resident_kernel<9, 8, 4, 128>, 512, resident_lds_bytes(9, 128)) : hipErrorUnknown;
            else e = resident_setup<9, 64>() ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, splat_resident_kernel<9, 8, 4, 64>, 512, resident_lds_bytes(9, 64)) : hipErrorUnknown;
        }
        if (e != hipSuccess) (void)hipGetLastError();
        ncu[dev].store(e == hipSuccess ? pr.multiProcessorCount : -1);
        cap[dev][which].store(e == hipSuccess && per_cu > 0 ? per_cu * pr.multiProcessorCount : -1);
    }
    const int c = cap[dev][which].load(), n = ncu[dev].load();
    if (capacity) *capacity = c > 0 ? c : 0;
    if (cus) *cus = n > 0 ? n : 0;
    return c > 0;
}

int resident_cu_count() { int n = 0; (void)resident_device_info(128, nullptr, &n); return n; }
int resident_block_capacity(int chunk) { int c = 0; (void)resident_device_info(chunk, &c, nullptr); return c; }

// false: the kernel could not be set up or launched (its dynamic LDS was refused): the caller walks the lists the
// list-major way
template <int CC>
static bool splat_resident_launch(const LatticeDev& L, const SplatResidentDev& R, const float* src, float* values, int slot, hipStream_t s) {
    const unsigned NG = (unsigned)L.n_groups;
    const unsigned rounds = ((unsigned)L.n_frames + NG - 1u) / NG;
    const unsigned tag = (g_resident_tag.fetch_add(1u) % 0x7FFFu) + 1u;
    const ValueView sv{const_cast<float*>(src), (size_t)L.N * (unsigned)CC, 0};
    // items of the list-major walk, should the planner have given up (splat_group_launch_g's grid)
    const unsigned nfg = ((unsigned)L.n_frames + NG - 1u) / NG;
    const unsigned long long max_mf = std::min<unsigned long long>(((unsigned long long)L.cap_f_mask + 1) / 2 + 1, (unsigned long long)L.m_bound);
    const unsigned n_items = nfg * (unsigned)((max_mf + 6) / 7) * NG;
    const dim3 grid(rounds * (unsigned)R.B * NG), block(512);
    if (R.chunk_log2 == 7) {
        if (!resident_setup<CC, 128>()) return false;
        splat_resident_kernel<CC, 8, 4, 128><<<grid, block, resident_lds_bytes(CC, 128), s>>>(L, R, sv, values, tag, slot, n_items);
    } else {
        if (!resident_setup<CC, 64>()) return false;
        splat_resident_kernel<CC, 8, 4, 64><<<grid, block, resident_lds_bytes(CC, 64), s>>>(L, R, sv, values, tag, slot, n_items);
    }
    return hipGetLastError() == hipSuccess;   // a refused launch is a refused set-up: fall back
}

int csr_pix_min() { return CS_PIX_MIN; }

// ---------------------------------------------------------------------------------------------
// blur along one lattice axis (permutohedral.cpp:556-569 / :496-510)
// ---------------------------------------------------------------------------------------------
template <bool SEQ>
__global__ void __launch_bounds__(256)
blur_kernel(LatticeDev L, int axis, int C, const float* __restrict__ old_v, float* __restrict__ new_v) {
    if (L.counters[1]) return;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int M = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    if (gid >= (long long)M * C) return;
    const int v = (int)(gid / C), c = (int)(gid - (long long)v * C);
    const int n1 = L.nb1[(size_t)axis * L.m_bound + v], n2 = L.nb2[(size_t)axis * L.m_bound + v];
    const float a = n1 >= 0 ? old_v[(size_t)n1 * C + c] : 0.0f;
    const float b = n2 >= 0 ? old_v[(size_t)n2 * C + c] : 0.0f;
    const float o = old_v[gid];
    if (SEQ) {
        new_v[gid] = (float)((double)o + 0.5 * (double)(a + b));  // seqCompute :505
    } else {
        const float sum = a + b;
        const float h = 0.5f * sum;
        new_v[gid] = o + h;                                        // sseCompute :566
    }
}

// All d+1 axis passes of one frame in one block: the frame's vertex values (M_f x C, ~10 KB for the
// Segmenter kernel) ping-pong between two LDS tables, so a filter costs one launch instead of d+1
// launch-latency-bound ones.  Frames whose values do not fit go through global memory, still inside
// the block (a vertex's neighbours belong to its own frame).  The result lands in `b`.
constexpr int BLUR_LDS_FLOATS = 6144;   // per table

template <bool SEQ>
__global__ void __launch_bounds__(1024)
blur_frames_kernel(LatticeDev L, int C, int reverse, float* __restrict__ a, float* __restrict__ b) {
    __shared__ float tab[2][BLUR_LDS_FLOATS];
    if (L.counters[1]) return;
    const int frame = blockIdx.x;
    const int M = L.counters[0] < L.m_bound ? L.counters[0] : L.m_bound;
    const int f0 = L.fstart[frame] < M ? L.fstart[frame] : M, f1 = L.fstart[frame + 1] < M ? L.fstart[frame + 1] : M;
    const int n = (f1 - f0) * C;
    const bool lds = n <= BLUR_LDS_FLOATS;
    float* ga = a + (size_t)f0 * C;
    float* gb = b + (size_t)f0 * C;
    if (lds) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) tab[0][i] = ga[i];
        __syncthreads();
    }
    int cur = 0;
    for (int t = 0; t <= L.d; t++) {
        const int axis = reverse ? L.d - t : t;
        const int* n1p = L.nb1 + (size_t)axis * L.m_bound + f0;
        const int* n2p = L.nb2 + (size_t)axis * L.m_bound + f0;
        const float* old_v = lds ? tab[cur] : (cur ? gb : ga);
        float* new_v = lds ? tab[cur ^ 1] : (cur ? ga : gb);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int v = i / C, c = i - v * C;
            const int n1 = n1p[v], n2 = n2p[v];
            const float x = n1 >= 0 ? old_v[(n1 - f0) * C + c] : 0.0f;
            const float y = n2 >= 0 ? old_v[(n2 - f0) * C + c] : 0.0f;
            const float o = old_v[i];
            if (SEQ) {
                new_v[i] = (float)((double)o + 0.5 * (double)(x + y));  // seqCompute :505
            } else {
                const float sum = x + y;
                const float h = 0.5f * sum;
                new_v[i] = o + h;                                        // sseCompute :566
            }
        }
        __syncthreads();   // block-wide: also orders the global-memory path (one block owns the frame)
        cur ^= 1;
    }
    if (lds) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) gb[i] = tab[cur][i];
    } else if (cur == 0) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) gb[i] = ga[i];
    }
}

// runs the d+1 axis passes; returns the buffer that holds the result
float* launch_blur(const LatticeDev& L, int C, bool seq, bool reverse, float* a, float* b, hipStream_t s, bool small_blocks) {
    // small_blocks: the pass runs beside the feature kernels (lattice build on the side stream) and a
    // 1024-thread block would wait for a whole free CU
    if (L.cap_f_mask + 1 <= 8192u) {   // at most 4096 vertices per frame: one block per frame is enough
        if (seq) blur_frames_kernel<true><<<dim3((unsigned)L.n_frames), dim3(small_blocks ? 256 : 1024), 0, s>>>(L, C, reverse ? 1 : 0, a, b);
        else blur_frames_kernel<false><<<dim3((unsigned)L.n_frames), dim3(small_blocks ? 256 : 1024), 0, s>>>(L, C, reverse ? 1 : 0, a, b);
        RV_LAUNCHED("blur_frames_kernel");
        return b;
    }
    const long long total = (long long)L.m_bound * C;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    float *cur = a, *nxt = b;
    for (int t = 0; t <= L.d; t++) {
        const int axis = reverse ? L.d - t : t;
        if (seq) blur_kernel<true><<<grid, block, 0, s>>>(L, axis, C, cur, nxt);
        else blur_kernel<false><<<grid, block, 0, s>>>(L, axis, C, cur, nxt);
        float* tmp = cur; cur = nxt; nxt = tmp;
    }
    RV_LAUNCHED("blur_kernel");
    return cur;
}

// ---------------------------------------------------------------------------------------------
// slice (permutohedral.cpp:574-584 / :515-524), one thread per (point, class)
//   OUT_MODE 0: out[p][c] = sliced                      (plain filter, rvseg_lattice_filter)
//   OUT_MODE 1: norm[p]   = 1/sqrt(sliced + 1e-20)      (normaliser, pairwise.cpp:55-56; C == 1)
//   OUT_MODE 2: tmp[p][c] = tmp[p][c] - (-w) * (sliced * norm[p])   (filter + Potts + inference)
// ---------------------------------------------------------------------------------------------
template <bool SEQ, int OUT_MODE>
__global__ void __launch_bounds__(256)
slice_kernel(LatticeDev L, int C, const float* __restrict__ values, float alpha, float neg_w, float* __restrict__ out,
             long long n_points) {
    if (L.counters[1]) return;
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_points * C) return;
    const long long p = gid / C;
    const int c = (int)(gid - p * C);
    const int dp1 = L.d + 1;
    float acc = 0.0f;
    for (int j = 0; j < dp1; j++) {
        const int o = L.offsets[p * dp1 + j];
        const float bw = L.bary[p * dp1 + j];
        const float val = values[(size_t)o * C + c];
        if (SEQ) {
            const float t = bw * val;
            const float u = t * alpha;
            acc += u;
        } else {
            const float w = bw * alpha;
            const float prod = w * val;
            acc += prod;
        }
    }
    if (OUT_MODE == 0) {
        out[gid] = acc;
    } else if (OUT_MODE == 1) {
        out[gid] = (float)(1.0 / sqrt((double)acc + 1e-20));
    } else {
        const float t = acc * L.norm[p];   // out = out*norm_.asDiagonal(), pairwise.cpp:79
        const float m = neg_w * t;         // out = -w_*Q, labelcompatibility.cpp:47
        out[gid] = out[gid] - m;           // tmp1 -= tmp2, densecrf.cpp:126
    }
}

// The normaliser's slice (C == 1, seqCompute rounding, OUT_MODE 1) with the d+1 offsets and weights of a
// point fetched as two wide rows.
template <int DP1>
__global__ void __launch_bounds__(256)
slice_norm_kernel(LatticeDev L, const float* __restrict__ values, float alpha, float* __restrict__ out, long long n_points) {
    if (L.counters[1]) return;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    int offs[DP1];
    float wts[DP1];
    load_row<DP1>(L.offsets + p * DP1, offs);
    load_row<DP1>(L.bary + p * DP1, wts);
    float acc = 0.0f;
#pragma unroll
    for (int j = 0; j < DP1; j++) {
        const float t = wts[j] * values[offs[j]];
        const float u = t * alpha;   // seqCompute :520
        acc += u;
    }
    out[p] = (float)(1.0 / sqrt((double)acc + 1e-20));   // pairwise.cpp:55-56
}

void launch_slice(const LatticeDev& L, int C, bool seq, int out_mode, const float* values, float neg_w, float* out,
                  long long n_points, hipStream_t s) {
    const float alpha = 1.0f / (1 + powf(2, (float)-L.d));  // permutohedral.cpp:571
    if (seq && out_mode == 1 && C == 1 && (L.d == 6 || L.d == 5 || L.d == 2)) {
        const dim3 g1((unsigned)((n_points + 255) / 256)), b1(256);
        if (L.d == 6) slice_norm_kernel<7><<<g1, b1, 0, s>>>(L, values, alpha, out, n_points);
        else if (L.d == 5) slice_norm_kernel<6><<<g1, b1, 0, s>>>(L, values, alpha, out, n_points);
        else slice_norm_kernel<3><<<g1, b1, 0, s>>>(L, values, alpha, out, n_points);
        RV_LAUNCHED("slice_norm_kernel");
        return;
    }
    const long long total = n_points * C;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define RV_SLICE(SEQ, OM) slice_kernel<SEQ, OM><<<grid, block, 0, s>>>(L, C, values, alpha, neg_w, out, n_points)
    if (seq) {
        if (out_mode == 0) RV_SLICE(true, 0); else if (out_mode == 1) RV_SLICE(true, 1); else RV_SLICE(true, 2);
    } else {
        if (out_mode == 0) RV_SLICE(false, 0); else if (out_mode == 1) RV_SLICE(false, 1); else RV_SLICE(false, 2);
    }
#undef RV_SLICE
    RV_LAUNCHED("slice_kernel");
}

// ---------------------------------------------------------------------------------------------
// tmp = -U (densecrf.cpp:123) and expAndNormalize (densecrf.cpp:98-106)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
neg_unary_kernel(ValueView unary, int negate, int C, int N, float* __restrict__ tmp, long long n_points) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n_points * C) return;
    const long long p = gid / C;
    const int c = (int)(gid - p * C);
    const float u = unary.at((unsigned)p, c, C, N);
    tmp[gid] = negate ? -u : u;
}

void launch_neg_unary(const ValueView& unary, bool negate, int C, int N, float* tmp, long long n_points, hipStream_t s) {
    const long long total = n_points * C;
    neg_unary_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(unary, negate ? 1 : 0, C, N, tmp, n_points);
    RV_LAUNCHED("neg_unary_kernel");
}

__global__ void __launch_bounds__(256)
softmax_kernel(const float* __restrict__ tmp, int C, int N, ValueView q, long long n_points) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const float* b = tmp + p * C;
    float mx = b[0];
    for (int c = 1; c < C; c++) { const float v = b[c]; if (v > mx) mx = v; }
    float sum = 0.0f;
    for (int c = 0; c < C; c++) {
        const float e = exp_f32_dev(b[c] - mx);
        q.ref((unsigned)p, c, C, N) = e;
        sum += e;
    }
    for (int c = 0; c < C; c++) {
        float& r = q.ref((unsigned)p, c, C, N);
        r = r / sum;
    }
}

void launch_softmax(const float* tmp, int C, int N, const ValueView& q, long long n_points, hipStream_t s) {
    softmax_kernel<<<dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, s>>>(tmp, C, N, q, n_points);
    RV_LAUNCHED("softmax_kernel");
}

// ---------------------------------------------------------------------------------------------
// Fused mean-field update for ONE Potts kernel (the Segmenter's case, segmenter.cpp:641-644):
//   slice (permutohedral.cpp:574-584) -> * norm (pairwise.cpp:79) -> * -w (labelcompatibility.cpp:47)
//   -> tmp1 = -U - tmp2 (densecrf.cpp:123-126) -> expAndNormalize (densecrf.cpp:98-106)
// One thread per point, all C classes in registers.  Vertex ids are frame-contiguous, so a block
// (256 points of one frame) stages that frame's blurred vertex values in LDS when they fit and
// slices from there; otherwise it gathers from HBM/L2.  Same operation order as the unfused
// kernels, so the result is bit-identical.
// ---------------------------------------------------------------------------------------------
constexpr int MF_LDS_BYTES = 24 * 1024;   // frames with more vertices than fit read `values` from L2
constexpr int MF_PTS = 512;               // points per block (2 per thread; 256 / 1024 / 2048 / 4096 measured +0.16 / +0.05 / +0.11 / +0.17 ms per 64-frame step)

// inputs of one point of the update: fetched one point ahead of their use
template <int C, int DP1>
struct MfIn {
    int offs[DP1 > 0 ? DP1 : 1];
    float wts[DP1 > 0 ? DP1 : 1];
    float ur[C];
    float nrm;
};

template <int C, int DP1>
__device__ __forceinline__ void mf_load(const LatticeDev& L, const ValueView& unary, int f0, size_t p, MfIn<C, DP1>& in) {
    if (DP1 > 0) {
        load_row<(DP1 > 0 ? DP1 : 1)>(L.offsets + p * DP1, in.offs);
        load_row<(DP1 > 0 ? DP1 : 1)>(L.bary + p * DP1, in.wts);
    }
    in.nrm = L.norm[p];
    load_row<C>(unary.base + unary.index((unsigned)p, 0, C, L.N), in.ur);
}

template <bool SEQ, int C, int DP1, bool USE_LDS>
__device__ __forceinline__ void mf_points(const LatticeDev& L, const float* __restrict__ values, const float* tab, float alpha,
                                          float neg_w, const ValueView& unary, int negate, const ValueView& Q, int scale_out,
                                          const MfLabels& lab, int frame, int f0, int i0) {
    constexpr int CP = (C + 3) / 4 * 4;
    constexpr int PER_THREAD = MF_PTS / 256;
    const int dp1 = DP1 > 0 ? DP1 : L.d + 1;
    MfIn<C, DP1> cur, nxt;
    if (i0 < L.N) mf_load<C, DP1>(L, unary, f0, (size_t)frame * L.N + i0, cur);
#pragma unroll
    for (int k = 0; k < PER_THREAD; k++) {   // the staged table serves MF_PTS points
        const int i = i0 + 256 * k;
        if (i >= L.N) break;
        const size_t p = (size_t)frame * L.N + i;
        if (k + 1 < PER_THREAD && i + 256 < L.N) mf_load<C, DP1>(L, unary, f0, p + 256, nxt);   // next point's rows travel now
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; c++) acc[c] = 0.0f;
#pragma unroll
        for (int j = 0; j < (DP1 > 0 ? DP1 : 8); j++) {
            if (DP1 == 0 && j >= dp1) break;
            const int o = DP1 > 0 ? cur.offs[j] : L.offsets[p * dp1 + j];
            const float bw = DP1 > 0 ? cur.wts[j] : L.bary[p * dp1 + j];
            float val[C];
            if (USE_LDS) {
                const float* row = tab + (o - f0) * CP;
#pragma unroll
                for (int c = 0; c < C; c++) val[c] = row[c];
            } else {
                load_row<C>(values + (size_t)o * C, val);
            }
            if (SEQ) {
#pragma unroll
                for (int c = 0; c < C; c++) { const float t = bw * val[c]; const float u = t * alpha; acc[c] += u; }
            } else {
                const float w = bw * alpha;
#pragma unroll
                for (int c = 0; c < C; c++) { const float prod = w * val[c]; acc[c] += prod; }
            }
        }
        const float nrm = cur.nrm;
        float b[C];
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float t = acc[c] * nrm;
            const float m = neg_w * t;
            const float u = cur.ur[c];
            b[c] = (negate ? -u : u) - m;
        }
        float mx = b[0];
#pragma unroll
        for (int c = 1; c < C; c++) if (b[c] > mx) mx = b[c];
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < C; c++) { b[c] = exp_f32_dev(b[c] - mx); sum += b[c]; }
        const size_t qrow = Q.index((unsigned)p, 0, C, L.N);
#pragma unroll
        for (int c = 0; c < C; c++) b[c] = b[c] / sum;
        if (lab.labels) lab.labels[((size_t)frame * lab.n_layers + lab.layer) * L.N + i] = (int8_t)label_rule(b, C, lab.mode, lab.unknown);
        if (scale_out) {   // not the last iteration: hand the next splat its input Q * norm directly
#pragma unroll
            for (int c = 0; c < C; c++) b[c] = b[c] * nrm;
        }
        store_row<C>(Q.base + qrow, b);
        cur = nxt;
    }
}

template <bool SEQ, int C, int DP1>   // DP1 = d+1 at compile time (wide offset / weight loads), 0 = runtime d
__global__ void __launch_bounds__(256)
mf_update_kernel(LatticeDev L, const float* __restrict__ values, float alpha, float neg_w, ValueView unary, int negate,
                 ValueView Q, int scale_out, MfLabels lab) {
    extern __shared__ __attribute__((aligned(16))) float tab[];
    if (L.counters[1]) return;   // uniform: hash overflow (flagged)
    constexpr int CP = (C + 3) / 4 * 4;
    const int bpf = (L.N + MF_PTS - 1) / MF_PTS;
    const int frame = blockIdx.x / bpf;
    const int i0 = (blockIdx.x - frame * bpf) * MF_PTS + threadIdx.x;
    const int f0 = L.fstart[frame], f1 = L.fstart[frame + 1];
    const int Mf = f1 - f0;
    const bool use_lds = (size_t)Mf * CP * sizeof(float) <= (size_t)MF_LDS_BYTES;   // block-uniform
    if (use_lds) {
        for (int idx = threadIdx.x; idx < Mf * C; idx += 256) {
            const int r = idx / C, c = idx - r * C;
            tab[r * CP + c] = values[(size_t)(f0 + r) * C + c];
        }
        __syncthreads();
        mf_points<SEQ, C, DP1, true>(L, values, tab, alpha, neg_w, unary, negate, Q, scale_out, lab, frame, f0, i0);
    } else {
        mf_points<SEQ, C, DP1, false>(L, values, tab, alpha, neg_w, unary, negate, Q, scale_out, lab, frame, f0, i0);
    }
}

// class counts with a fused softmax_unary / mf_update instantiation (the switch lists below)
bool mf_fused_supported(int C) { return (C >= 2 && C <= 10) || C == 12 || C == 16 || C == 21; }

// returns false when C has no fused instantiation (the caller then runs the unfused kernels)
bool launch_mf_update(const LatticeDev& L, int C, const float* values, float neg_w, const ValueView& unary, bool negate,
                      const ValueView& Q, bool scale_out, const MfLabels& lab, hipStream_t s) {
    const float alpha = 1.0f / (1 + powf(2, (float)-L.d));
    const int bpf = (L.N + MF_PTS - 1) / MF_PTS;
    const dim3 grid((unsigned)(bpf * L.n_frames)), block(256);
#define RV_MF(SEQ, CC)                                                                                            \
    if (L.d == 6) mf_update_kernel<SEQ, CC, 7><<<grid, block, MF_LDS_BYTES, s>>>(L, values, alpha, neg_w, unary, negate ? 1 : 0, Q, scale_out ? 1 : 0, lab); \
    else mf_update_kernel<SEQ, CC, 0><<<grid, block, MF_LDS_BYTES, s>>>(L, values, alpha, neg_w, unary, negate ? 1 : 0, Q, scale_out ? 1 : 0, lab); \
    RV_LAUNCHED("mf_update_kernel"); \
    return true
    switch (C) {
        case 2: RV_MF(true, 2);
        case 3: RV_MF(false, 3);
        case 4: RV_MF(false, 4);
        case 5: RV_MF(false, 5);
        case 6: RV_MF(false, 6);
        case 7: RV_MF(false, 7);
        case 8: RV_MF(false, 8);
        case 9: RV_MF(false, 9);
        case 10: RV_MF(false, 10);
        case 12: RV_MF(false, 12);
        case 16: RV_MF(false, 16);
        case 21: RV_MF(false, 21);
        default: return false;
    }
#undef RV_MF
}

// Q0 = expAndNormalize(-U) straight from the unary (densecrf.cpp:120), one thread per point
// scale != nullptr: store fl(Q * scale[p]) instead of Q -- the input of the next splat
// (DenseKernel::filter, pairwise.cpp:66), so the splat needs no per-entry normaliser
template <int C>
__global__ void __launch_bounds__(256)
softmax_unary_kernel(ValueView unary, int negate, int N, ValueView q, long long n_points, const float* __restrict__ scale) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_points) return;
    const size_t urow = unary.index((unsigned)p, 0, C, N);
    float b[C];
    load_row<C>(unary.base + urow, b);
#pragma unroll
    for (int c = 0; c < C; c++) b[c] = negate ? -b[c] : b[c];
    float mx = b[0];
#pragma unroll
    for (int c = 1; c < C; c++) if (b[c] > mx) mx = b[c];
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < C; c++) { b[c] = exp_f32_dev(b[c] - mx); sum += b[c]; }
    const size_t qrow = q.index((unsigned)p, 0, C, N);
#pragma unroll
    for (int c = 0; c < C; c++) b[c] = b[c] / sum;
    if (scale) {
        const float sc = scale[p];
#pragma unroll
        for (int c = 0; c < C; c++) b[c] = b[c] * sc;
    }
    store_row<C>(q.base + qrow, b);
}

bool launch_softmax_unary(const ValueView& unary, bool negate, int C, int N, const ValueView& q, long long n_points,
                          const float* scale, hipStream_t s) {
    const dim3 grid((unsigned)((n_points + 255) / 256)), block(256);
#define RV_SU(CC) softmax_unary_kernel<CC><<<grid, block, 0, s>>>(unary, negate ? 1 : 0, N, q, n_points, scale); RV_LAUNCHED("softmax_unary_kernel"); return true
    switch (C) {
        case 2: RV_SU(2); case 3: RV_SU(3); case 4: RV_SU(4); case 5: RV_SU(5); case 6: RV_SU(6); case 7: RV_SU(7);
        case 8: RV_SU(8); case 9: RV_SU(9); case 10: RV_SU(10); case 12: RV_SU(12); case 16: RV_SU(16); case 21: RV_SU(21);
        default: return false;
    }
#undef RV_SU
}

__global__ void __launch_bounds__(256)
fill_int_kernel(int* p, int v, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_fill_int(int* p, int v, long long n, hipStream_t s) {
    if (n <= 0) return;
    fill_int_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(p, v, n);
    RV_LAUNCHED("fill_int_kernel");
}

}  // namespace rvseg
