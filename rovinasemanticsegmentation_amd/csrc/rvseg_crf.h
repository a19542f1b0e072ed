// Device-side views and launch interface of the lattice / mean-field kernels (kernels_crf.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "rvseg_internal.h"

namespace rvseg {

// Everything a lattice kernel needs, passed by value.
struct LatticeDev {
    int d;          // feature dimension (<= 7)
    int N;          // points per frame
    int Npad;       // N rounded up to a multiple of 4 (SSE block padding, permutohedral.cpp:196)
    int n_frames;
    unsigned cap_f_mask;   // per-frame hash capacity - 1 (power of two)
    unsigned cap_f_log2;
    unsigned cap_total;    // n_frames * per-frame capacity
    int m_bound;        // capacity of the per-vertex arrays
    float scale[8];     // diagonal of E, permutohedral.cpp:177-182
    int* state;                  // per slot: EMPTY / LOCKED / FILLED
    unsigned long long* tkeys;   // per slot: 8 x int16 key (d coordinates .. frame)
    int* slot_to_id;
    int* counters;               // [0] vertices M, [1] overflow flag, [2] (unused), [3] entries of the longest vertex list
    int* fstart;                 // n_frames + 1: first vertex id of each frame (ids are frame-contiguous)
    unsigned long long* vkeys;   // per vertex id: key
    int* offsets;                // P x (d+1): slot, later vertex id
    float* bary;                 // P x (d+1)
    int *nb1, *nb2;              // (d+1) x m_bound blur neighbours (-1 = none)
    uint2* csr_pw;               // entries sorted by vertex, ascending point index inside a vertex:
                                 // {point index, barycentric weight bits}
    float* csr_nrm;              // norm[point] per entry (multi-kernel path only, built on demand)
    unsigned *vstart, *vend;     // per vertex [start, end) into the csr arrays
    unsigned* vorder;            // vorder[fstart[f] + k] = k-th longest vertex of frame f (splat launch order)
    int n_groups;                // 8 when the chunk has >= 8 frames, else 1
    int group_vertices;          // list-major walk, C = 8 / 9: vertices per block (0 = by the chunk's shape, 6, 7)
    int ordered_sum_scan;        // normaliser: exact wave-scan sums (1) or the serial chain (0); same bits either way
    int cs_pix;                  // counting sort: points per wave-block (256 for launches of <= 8 frames, else 1024)
    unsigned heavy_from;         // list-major walk: lists of this many entries and more belong to scan blocks (0 = none) ...
    unsigned scan_ranks;         // ... if they are among the scan_ranks longest of their frame
    // counting-sort path: after the scan, bh[wave-block][vertex] is the position of the vertex's first entry at or
    // after that wave-block, i.e. the vertex-major lists can be cut at any multiple of CS_PIX points without
    // another sort (the resident band schedule does)
    const unsigned* bh;          // per frame a dense [wbpf][M_f] matrix starting at wbpf * fstart[frame]; null on the radix path
    int wbpf;                    // wave-blocks per frame
    float* norm;                 // per point, pairwise.cpp:55-56
};

// Band-interleaved resident schedule of the ordered splat ("resident bands").  A frame's
// vertices are dealt to B blocks that all stay on the chip for the whole launch; a block walks ITS vertices band after
// band of `band_wb` wave-blocks of points, so the d+1 readers of a point's row pass within a few bands of each other
// and meet in the XCD's L2, while every chain stays inside one block (sums never travel between blocks: no hand-off,
// no dependency -- the pacing is a hint, not a condition of correctness).  The walk is a list of tiles built once per
// lattice.  A tile is one step of the block: seven slots, each up to 64 (or 128) consecutive entries of one vertex.  Inside a
// band the chunks of the block's vertices are packed into the fewest tiles that keep a vertex's chunks in
// different tiles, in order (wrap-around rule: tiles = max(longest vertex, chunks / 7)); which vertex a slot serves
// changes from tile to tile, so the running sums live in LDS and a slot swaps its sum when its vertex changes.
constexpr int RES_MAXB = 16;          // blocks per frame at most
constexpr int RES_MAX_OWNV = 640;     // vertices a block can own (their running sums live in LDS): a 2 048-vertex frame on 4 blocks
constexpr int RES_MAX_VERTS = 4096;   // vertices of a frame the planner handles (= the counting-sort path's limit, CS_MCAP)
constexpr int RES_MAX_BANDS = 512;
struct SplatResidentDev {
    unsigned* tdesc;             // [n_frames][7][cap_tiles]: (first entry - frame's first entry) << 8 | entries (0..128)
    unsigned short* tvl;         // [n_frames][7][cap_tiles]: block-local vertex of the slot (n_own = nobody: entries is 0)
    unsigned* tinfo;             // [n_frames][cap_tiles]: band << 16 | most entries of a slot
    unsigned* blk_tile0;         // [n_frames][RES_MAXB + 1]: block j walks tiles [blk_tile0[j], blk_tile0[j + 1]) of its frame
    unsigned short* blk_verts;   // [n_frames][RES_MAXB][RES_MAX_OWNV]: frame-local vertex of each block-local one
    unsigned* blk_nown;          // [n_frames][RES_MAXB]
    unsigned* jb_tile;           // [n_frames][RES_MAXB][n_bands + 1]: first tile of (block, band), relative to the frame
    unsigned* prog;              // [2 scratch slots][n_frames][RES_MAXB]: launch tag << 16 | band reached
    int* flags;                  // [0] frames the planner could not handle, [1] = 1: schedule valid
    unsigned long long* trace;   // optional (rvseg_schedule.trace): per (frame, block) 8 words: start, end, tiles, ticks spent waiting for the pace (10 ns ticks), shader clocks
    int B, band_wb, n_bands, window;
    int chunk_log2;              // entries per slot and tile: 2^6 or 2^7
    unsigned cap_tiles;
};

// Where the lattice features come from.
struct FeatureSource {
    int mode;            // 0: feat array P x d; 1: frame mode (cloud + colours, d = 6)
    const float* feat;
    const float4* cloud;
    const uint8_t* rgb;
    float xyz_kernel, rgb_kernel;
};

// N x C values addressed either densely or inside the per-frame posterior layout
// (frame stride sumC*N floats, layer offset N*prefixC).
struct ValueView {
    float* base;
    size_t frame_stride;
    size_t layer_off;
    __host__ __device__ __forceinline__ size_t index(unsigned p, int c, int C, int N) const {
        const unsigned frame = p / (unsigned)N;
        const unsigned i = p - frame * (unsigned)N;
        return (size_t)frame * frame_stride + layer_off + (size_t)i * C + c;
    }
    __device__ __forceinline__ float at(unsigned p, int c, int C, int N) const { return base[index(p, c, C, N)]; }
    __device__ __forceinline__ float& ref(unsigned p, int c, int C, int N) const { return base[index(p, c, C, N)]; }
};

struct SortBuffers {
    unsigned *keys_in, *keys_out, *vals_in, *vals_out;
    void* temp;
    size_t temp_bytes;
    int key_bits;
    void* scan_temp;
    size_t scan_temp_bytes;
    unsigned* block_hist;   // counting-sort fast path: n_frames x wave-blocks x (cap_f/2), or null
};

void launch_fill_int(int* p, int v, long long n, hipStream_t s);
void launch_lattice_points(const LatticeDev& L, const FeatureSource& fs, hipStream_t s);
void launch_lattice_finish(const LatticeDev& L, SortBuffers& sb, long long n_entries, hipStream_t s, int phase = 0);
size_t sort_temp_bytes(long long n_entries, int key_bits);
size_t scan_temp_bytes(unsigned cap);
bool csr_fast_path(const LatticeDev& L);
size_t csr_fast_bytes(const LatticeDev& L);
int csr_pix_min();
void launch_csr_norm(const LatticeDev& L, long long n_entries, hipStream_t s);
// mode 0: in = src; 1: in = fl(src * norm); 2: in = 1
// own_q: src is the mean-field loop's own Q * norm (finite, non-negative): enables the select-free producer
void launch_splat(const LatticeDev& L, const ValueView& src, int C, int mode, float* values, hipStream_t s, bool own_q = false,
                  const SplatResidentDev* resident = nullptr, int slot = 0);
// builds the resident band schedule from the counting-sort table (after launch_lattice_finish)
void launch_resident_plan(const LatticeDev& L, const SplatResidentDev& r, hipStream_t s);
// how many blocks of the resident splat kernel fit on the device at once (0: unknown)
int resident_block_capacity(int chunk);
int resident_cu_count();
float* launch_blur(const LatticeDev& L, int C, bool seq, bool reverse, float* a, float* b, hipStream_t s, bool small_blocks = false);
// out_mode 0: plain, 1: normaliser, 2: inference update (tmp -= (-w) * (sliced * norm))
void launch_slice(const LatticeDev& L, int C, bool seq, int out_mode, const float* values, float neg_w, float* out,
                  long long n_points, hipStream_t s);
// fused slice + Potts + softmax for a single pairwise kernel; false if C is not instantiated
// labels of the final marginals straight from the last update (the values are in registers there):
// labels[(frame * n_layers + layer) * N + point]; labels == nullptr: none
struct MfLabels {
    int8_t* labels;
    int mode, unknown, n_layers, layer;
};
bool mf_fused_supported(int C);
bool launch_mf_update(const LatticeDev& L, int C, const float* values, float neg_w, const ValueView& unary, bool negate,
                      const ValueView& Q, bool scale_out, const MfLabels& lab, hipStream_t s);
void launch_neg_unary(const ValueView& unary, bool negate, int C, int N, float* tmp, long long n_points, hipStream_t s);
bool launch_softmax_unary(const ValueView& unary, bool negate, int C, int N, const ValueView& q, long long n_points,
                          const float* scale, hipStream_t s);
void launch_softmax(const float* tmp, int C, int N, const ValueView& q, long long n_points, hipStream_t s);

}  // namespace rvseg
