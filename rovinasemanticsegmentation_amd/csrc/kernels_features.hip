// Feature-side kernels for gfx950: Lab conversion + back-projection, the PCL-style smoothing-window
// map (depth-change map + two-pass chamfer distance), and the normal feature.
//
// Reference: include/feature_extractor.h:125-291.  The arithmetic of cvtColor / PCL normals lives
// in libraries the reference does not vendor; the definitions implemented here are the ones
// written down in DESIGN.md and restated independently by the CPU oracle.
#include <algorithm>
#include <cstdlib>

#include "device_math.h"
#include "rvseg_internal.h"
#include "rvseg_kernels.h"

namespace rvseg {

// ---------------------------------------------------------------------------------------------
// prep: one thread per pixel.
//   lab   : cvtColor(CV_BGR2Lab) on 8-bit data, integer LUT pipeline (feature_extractor.h:129),
//           stored as one dword (L | a<<8 | b<<16) so that the patch kernel fetches a tap with a
//           single load.
//   cloud : (R*Kinv)*(d*x, d*y, d) + t, NaN where depth is outside [d_min, d_max]
//           (feature_extractor.h:209-223), stored float4 (x,y,z,0).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool pair_fails(float z0, float z1) {
    const float thr = (0.02f * (fabsf(z0) + 1.0f) * 2.0f);
    return fabsf(z0 - z1) > thr || !finite_f(z0) || !finite_f(z1);
}

__global__ void __launch_bounds__(256)
prep_kernel(FrameGeom g, LabCoeffs lc, const uint16_t* __restrict__ gamma, const uint16_t* __restrict__ cbrt_tab,
            const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth,
            const float* __restrict__ calibA,  // n x 12: A = R*Kinv (row-major 9), t (3)
            uint32_t* __restrict__ lab, float4* __restrict__ cloud, uint8_t* __restrict__ change, int n_frames,
            uint2* __restrict__ lab2) {
    const size_t npix = (size_t)g.W * g.H;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= npix * (size_t)n_frames) return;
    const int frame = (int)(gid / npix);
    const int pix = (int)(gid - (size_t)frame * npix);
    const int y = pix / g.W, x = pix - y * g.W;

    if (lab) {
        const uint8_t* s = rgb + gid * 3;
        const int c0 = gamma[s[0]], c1 = gamma[s[1]], c2 = gamma[s[2]];
        const int fX = cbrt_tab[(c0 * lc.c[0] + c1 * lc.c[1] + c2 * lc.c[2] + (1 << 11)) >> 12];
        const int fY = cbrt_tab[(c0 * lc.c[3] + c1 * lc.c[4] + c2 * lc.c[5] + (1 << 11)) >> 12];
        const int fZ = cbrt_tab[(c0 * lc.c[6] + c1 * lc.c[7] + c2 * lc.c[8] + (1 << 11)) >> 12];
        const int Lscale = (116 * 255 + 50) / 100;
        const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
        int L = (Lscale * fY + Lshift + (1 << 14)) >> 15;
        int a = (500 * (fX - fY) + 128 * (1 << 15) + (1 << 14)) >> 15;
        int b = (200 * (fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15;
        L = min(max(L, 0), 255); a = min(max(a, 0), 255); b = min(max(b, 0), 255);
        const uint32_t v = (uint32_t)L | ((uint32_t)a << 8) | ((uint32_t)b << 16);
        lab[gid] = v;
        if (lab2) {   // {this pixel, the pixel below}: this thread writes its own top word and the bottom word of the row above
            uint32_t* l2 = reinterpret_cast<uint32_t*>(lab2);
            l2[2 * gid] = v;
            if (y > 0) l2[2 * (gid - (size_t)g.W) + 1] = v;
            if (y == g.H - 1) l2[2 * gid + 1] = v;
        }
    }
    if (cloud) {
        const float* A = calibA + (size_t)frame * 12;
        const float d = (float)depth[gid] / 1000.0f;
        float m0, m1, m2;
        if (d < g.depth_min || d > g.depth_max) {
            m0 = m1 = m2 = __int_as_float(0x7fc00000);
        } else {
            m0 = d * (float)x; m1 = d * (float)y; m2 = d;
        }
        float4 o;
        o.x = ((A[0] * m0 + A[1] * m1) + A[2] * m2) + A[9];
        o.y = ((A[3] * m0 + A[4] * m1) + A[5] * m2) + A[10];
        o.z = ((A[6] * m0 + A[7] * m1) + A[8] * m2) + A[11];
        o.w = 0.f;
        cloud[gid] = o;
        if (change) {
            // PCL's depth-change map (integral_image_normal.hpp, computeFeature): a pixel is marked when it
            // belongs to a horizontal or vertical neighbour pair whose depths differ by more than
            // 0.02*(|z|+1)*2 or contain a non-finite value.  The neighbours' z are recomputed from their depth
            // with the very expression above, so they are the values those pixels store.
            const uint16_t* dp = depth + (size_t)frame * npix;
            auto z_at = [&](int rr, int cc) -> float {
                const float dn = (float)dp[(size_t)rr * g.W + cc] / 1000.0f;
                float n0, n1, n2;
                if (dn < g.depth_min || dn > g.depth_max) {
                    n0 = n1 = n2 = __int_as_float(0x7fc00000);
                } else {
                    n0 = dn * (float)cc; n1 = dn * (float)rr; n2 = dn;
                }
                return ((A[6] * n0 + A[7] * n1) + A[8] * n2) + A[11];
            };
            const int W = g.W, H = g.H, r = y, c = x;
            const float z = o.z;
            bool ch = false;
            if (r <= H - 2 && c <= W - 2) ch = pair_fails(z, z_at(r, c + 1)) || pair_fails(z, z_at(r + 1, c));
            if (!ch && c >= 1 && r <= H - 2) ch = pair_fails(z_at(r, c - 1), z);
            if (!ch && r >= 1 && c <= W - 2) ch = pair_fails(z_at(r - 1, c), z);
            change[gid] = ch ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// smoothing-window map.  PCL computes a depth-change map, then a two-pass raster chamfer distance
// (steps 1.0 / 1.4 in float), then window = int(min(distance, 10)) (NaN normal when <= 2).
// The raster passes are sequential over the whole image; only distances below 10 matter and every
// step costs >= 1, so a pixel's value is decided inside its 19x19 neighbourhood: a value below 10 is the float sum
// along a path of at most 9 steps from a depth-change pixel -- forward steps (first pass: from the left / upper
// neighbours), then backward steps (second pass) -- and every pixel of such a path lies within 9 pixels of its end.
// Re-running both passes on a tile plus an apron therefore gives the image-wide result wherever that is below 10, and
// something >= 10 wherever it is not (restricting the region only removes paths).  Each workgroup (one wave) does
// that for its tile plus a 10-pixel apron held in LDS, sweeping anti-diagonals t = 2*row + col so that all four
// already-final neighbours of a cell are available.  The float additions follow the raster order exactly.
// (Until the end of round 3 the apron was 20 pixels -- the reach of a forward path plus that of a backward path, added
// up although they share the 9 steps: 25 % of a tile's cells were its own pixels, now 55 %: 0.74 -> 0.38 ms.)
// ---------------------------------------------------------------------------------------------
// Tile shape: 44 rows + apron = 64 table rows = one lane per row with no lane idle; 80 columns (26 KB of LDS: six
// waves per CU; 108 and 128 columns measured 0.54 and 0.47 ms against 0.38).
constexpr int DM_TW = 80, DM_TH = 44, DM_APRON = 10;
constexpr int DM_CW = DM_TW + 2 * DM_APRON;  // 100
// row pitch of the LDS table.  Lane l touches row l at column t - 2l, i.e. word (pitch - 2) * l + t: with pitch 104
// lanes l and l + 16 of a 32-lane group met in one bank (PMC: 62 % of the LDS-active cycles were conflict cycles);
// an odd multiplier spreads a group over all 32 banks
constexpr int DM_PITCH = DM_CW + 1;
constexpr int DM_CH = DM_TH + 2 * DM_APRON;  // 64
static_assert(DM_CH <= 64, "one lane per table row");

// value of the lane below in the wave (lane l takes lane l-1's); lane 0 takes `edge`
__device__ __forceinline__ float wave_shr1(float v, float edge) {
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __int_as_float(r);
}

__global__ void __launch_bounds__(64)
window_map_kernel(FrameGeom g, const uint8_t* __restrict__ change_all, uint8_t* __restrict__ rect_all) {
    __shared__ float dist[DM_CH * DM_PITCH];
    __shared__ int any_zero;
    const int W = g.W, H = g.H;
    const int tiles_x = (W + DM_TW - 1) / DM_TW;
    const int tiles_y = (H + DM_TH - 1) / DM_TH;
    const int frame = blockIdx.x / (tiles_x * tiles_y);
    const int tile = blockIdx.x - frame * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x0 = tx * DM_TW - DM_APRON, y0 = ty * DM_TH - DM_APRON;
    const uint8_t* change = change_all + (size_t)frame * W * H;
    uint8_t* rect = rect_all + (size_t)frame * W * H;
    const int lane = threadIdx.x;
    const float BIG = (float)(W + H);

    if (lane == 0) any_zero = 0;
    __syncthreads();
    int my_zero = 0;
    for (int idx = lane; idx < DM_CH * DM_CW; idx += 64) {
        const int lr = idx / DM_CW, lc = idx - lr * DM_CW;
        const int r = y0 + lr, c = x0 + lc;
        float v = BIG;
        if (r >= 0 && r < H && c >= 0 && c < W && change[(size_t)r * W + c]) { v = 0.0f; my_zero = 1; }
        dist[lr * DM_PITCH + lc] = v;
    }
    if (my_zero) any_zero = 1;
    __syncthreads();

    if (any_zero) {
        // Lane l owns one row of the tile and walks it left to right, two columns behind the lane
        // above (anti-diagonal t = 2*row + col).  The three upper neighbours are the last three
        // results of lane l-1 and arrive by a one-lane wave shift (DPP, no LDS round trip); the
        // left neighbour is the lane's own previous result.  Only the initial value of a cell is
        // read from LDS (prefetched one step ahead) and its final value written back.
        const int n_steps = 2 * (DM_CH - 1) + DM_CW;
        const bool row_ok = lane < DM_CH;
        // Both passes are written without data-dependent branches: whether a lane updates a cell at step t
        // depends only on its column index (lane-constant bounds), the result is a select.
        // ---- first pass: rows 1..H-1, columns 1..W-1 of the image, raster order
        {
            const int lr = lane;
            const int r = y0 + lr;
            const bool row_upd = row_ok && r >= 1 && r < H;
            const int lo = 1 - x0 > 0 ? 1 - x0 : 0, hi0 = W - x0 < DM_CW ? W - x0 : DM_CW;   // updated columns [lo, hi)
            const int hi = hi0 < lo ? lo : hi0;
            const int edge = W - 1 - x0;                  // tile column of the last image column
            float* rowp = dist + (row_ok ? lr : 0) * DM_PITCH;
            float o1 = BIG, o2 = BIG, o3 = BIG;          // this lane's results at steps t-1, t-2, t-3
            int lc = -2 * lane;
            float center_next = (row_ok && lc >= 0 && lc < DM_CW) ? rowp[lc] : BIG;
            for (int t = 0; t < n_steps; t++, lc++) {
                const float center = center_next;
                const int lcn = lc + 1;
                const int lcn_c = lcn < 0 ? 0 : (lcn >= DM_CW ? DM_CW - 1 : lcn);
                const float nxt = rowp[lcn_c];
                center_next = (row_ok && (unsigned)lcn < (unsigned)DM_CW) ? nxt : BIG;
                // results of the lane above: step t-1 -> column lc+1, t-2 -> lc, t-3 -> lc-1
                float upRight = wave_shr1(o1, BIG);
                const float up = wave_shr1(o2, BIG), upLeft = wave_shr1(o3, BIG);
                upRight = lc >= edge ? BIG : upRight;    // prev[W] belongs to the next image row
                const float mn = fminf(fminf(upLeft + 1.4f, up + 1.0f), fminf(o1 + 1.0f, upRight + 1.4f));
                const bool upd = row_upd && (unsigned)(lc - lo) < (unsigned)(hi - lo);
                const float out = (upd && mn < center) ? mn : center;   // center is BIG outside the tile
                if (upd) rowp[lc] = out;
                o3 = o2; o2 = o1; o1 = out;
            }
        }
        __syncthreads();
        // ---- second pass: rows H-2..0, columns W-2..0, reverse raster order (lane l <-> row CH-1-l)
        {
            const int lr = DM_CH - 1 - lane;
            const int r = y0 + lr;
            const bool row_upd = row_ok && r >= 0 && r <= H - 2;
            // mirrored column index cp = CW-1-lc; updated image columns 0..W-2  <=>  lc in [max(0,-x0), min(CW, W-1-x0))
            const int llo = -x0 > 0 ? -x0 : 0, lhi = W - 1 - x0 < DM_CW ? W - 1 - x0 : DM_CW;
            const int lo = DM_CW - lhi, hi1 = DM_CW - llo;   // the same range in cp
            const int hi = hi1 < lo ? lo : hi1;
            const int edge = DM_CW - 1 + x0;                 // cp of image column 0: cells with cp >= edge have no lowerLeft
            float* rowp = dist + (row_ok ? lr : 0) * DM_PITCH;
            float o1 = BIG, o2 = BIG, o3 = BIG;
            int cp = -2 * lane;                          // mirrored column index
            float center_next = (row_ok && cp >= 0 && cp < DM_CW) ? rowp[DM_CW - 1 - cp] : BIG;
            for (int t = 0; t < n_steps; t++, cp++) {
                const float center = center_next;
                const int cpn = cp + 1;
                const int cpn_c = cpn < 0 ? 0 : (cpn >= DM_CW ? DM_CW - 1 : cpn);
                const float nxt = rowp[DM_CW - 1 - cpn_c];
                center_next = (row_ok && (unsigned)cpn < (unsigned)DM_CW) ? nxt : BIG;
                // lane l-1 is the row below: its step t-1 is column lc-1, t-2 -> lc, t-3 -> lc+1
                float lowerLeft = wave_shr1(o1, BIG);
                const float lower = wave_shr1(o2, BIG), lowerRight = wave_shr1(o3, BIG);
                lowerLeft = cp >= edge ? BIG : lowerLeft;   // next[-1] belongs to the previous image row
                const float mn = fminf(fminf(lowerLeft + 1.4f, lower + 1.0f), fminf(o1 + 1.0f, lowerRight + 1.4f));
                const bool upd = row_upd && (unsigned)(cp - lo) < (unsigned)(hi - lo);
                const float out = (upd && mn < center) ? mn : center;
                if (upd) rowp[DM_CW - 1 - cp] = out;
                o3 = o2; o2 = o1; o1 = out;
            }
        }
        __syncthreads();
    }
    // window size per pixel of the tile: 0 <=> NaN normal (smoothing <= 2), else int(smoothing)
    for (int idx = lane; idx < DM_TH * DM_TW; idx += 64) {
        const int lr = idx / DM_TW + DM_APRON, lc = idx % DM_TW + DM_APRON;
        const int r = y0 + lr, c = x0 + lc;
        if (r < H && c < W) {
            const float dv = dist[lr * DM_PITCH + lc];
            const float smoothing = dv < 10.0f ? dv : 10.0f;
            rect[(size_t)r * W + c] = smoothing > 2.0f ? (uint8_t)(int)smoothing : (uint8_t)0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// normal feature at the stride-grid sample points: acos(|n_z|) or -2 (feature_extractor.h:275-283)
// with n = normalise(sum_window(dy) x sum_window(dx)), gradients summed as 2^-32 fixed-point int64
// (exact, order independent).  One thread per sample point.
// ---------------------------------------------------------------------------------------------
// v * 2^32 rounded to the nearest integer (ties to even), clamped to +-9e18 -- the oracle computes the same with
// llrint.  (An integer-only form built from the float's mantissa -- profiles/analysis/fix32_check.cpp proves it
// equal on 4e8 random and all boundary encodings -- measured 100 us SLOWER per 64-frame launch than this f64
// sequence: variable 64-bit shifts are not cheaper than the double-rate multiply and conversion.)
__device__ __forceinline__ long long to_fix32(float v) {
    double s = (double)v * 4294967296.0;
    // |s| < 2^51 (|v| < 2^19, any gradient of metric coordinates): s + 1.5 * 2^52 is rounded by the addition itself
    // (nearest, ties to even) and its low bits ARE the integer -- one f64 add and one 64-bit subtract instead of the
    // library's f64 -> i64 routine (gfx950 has no such conversion instruction)
    if (fabs(s) < 2251799813685248.0) {
        const double magic = 6755399441055744.0;   // 1.5 * 2^52
        return __double_as_longlong(s + magic) - __double_as_longlong(magic);
    }
    s = s > 9.0e18 ? 9.0e18 : s;
    s = s < -9.0e18 ? -9.0e18 : s;
    return __double2ll_rn(s);
}

__global__ void __launch_bounds__(256)
normal_feature_kernel(FrameGeom g, const float4* __restrict__ cloud_all, const uint8_t* __restrict__ rect_all,
                      float* __restrict__ nfeat_all, int n_frames) {
    const int per_frame = g.lw * g.lh;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= per_frame * n_frames) return;
    const int frame = gid / per_frame;
    const int p = gid - frame * per_frame;
    const int ly = p / g.lw, lx = p - ly * g.lw;
    const int ri = ly * g.stride, ci = lx * g.stride;
    const int W = g.W, H = g.H;
    const float4* cloud = cloud_all + (size_t)frame * W * H;
    const uint8_t* rect_map = rect_all + (size_t)frame * W * H;
    float out = -2.0f;
    const int border = 10;
    if (ri >= border && ri < H - border && ci >= border && ci < W - border) {
        const int rect = rect_map[(size_t)ri * W + ci];
        const float zc = cloud[(size_t)ri * W + ci].z;
        if (rect > 0 && finite_f(zc)) {
            const int rect2 = rect >> 1;
            const int sx = ci - rect2, sy = ri - rect2;
            long long gx0 = 0, gx1 = 0, gx2 = 0, gy0 = 0, gy1 = 0, gy2 = 0;
            int cnt_x = 0, cnt_y = 0;
            for (int y = sy; y < sy + rect; y++) {
                for (int x = sx; x < sx + rect; x++) {
                    float dx0 = 0.f, dx1 = 0.f, dx2 = 0.f, dy0 = 0.f, dy1 = 0.f, dy2 = 0.f;
                    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
                        const float4 r = cloud[(size_t)y * W + x + 1], l = cloud[(size_t)y * W + x - 1];
                        const float4 dn = cloud[(size_t)(y + 1) * W + x], up = cloud[(size_t)(y - 1) * W + x];
                        dx0 = r.x - l.x; dx1 = r.y - l.y; dx2 = r.z - l.z;
                        dy0 = dn.x - up.x; dy1 = dn.y - up.y; dy2 = dn.z - up.z;
                    }
                    if (finite_f(dx0) && finite_f(dx1) && finite_f(dx2)) {
                        cnt_x++; gx0 += to_fix32(dx0); gx1 += to_fix32(dx1); gx2 += to_fix32(dx2);
                    }
                    if (finite_f(dy0) && finite_f(dy1) && finite_f(dy2)) {
                        cnt_y++; gy0 += to_fix32(dy0); gy1 += to_fix32(dy1); gy2 += to_fix32(dy2);
                    }
                }
            }
            if (cnt_x > 0 && cnt_y > 0) {
                const double k = 1.0 / 4294967296.0;
                const double GX0 = (double)gx0 * k, GX1 = (double)gx1 * k, GX2 = (double)gx2 * k;
                const double GY0 = (double)gy0 * k, GY1 = (double)gy1 * k, GY2 = (double)gy2 * k;
                const double n0 = GY1 * GX2 - GY2 * GX1;
                const double n1 = GY2 * GX0 - GY0 * GX2;
                const double n2 = GY0 * GX1 - GY1 * GX0;
                const double len2 = (n0 * n0 + n1 * n1) + n2 * n2;
                if (len2 != 0.0) {
                    const float nz = (float)(n2 / sqrt(len2));
                    if (nz == nz) out = acos_f32_dev(fabsf(nz));
                }
            }
        }
    }
    nfeat_all[gid] = out;
}

// ---------------------------------------------------------------------------------------------
// LDS-tiled variant: a block owns 8 x 16 sample points.  The fixed-point gradients of its pixel tile
// (+ 5-px apron) are formed once, cooperatively, and turned into a two-dimensional prefix sum
// (summed-area table) IN PLACE: a window sum is then four corner reads per channel instead of up to
// 10 x 10 cells -- the integral-image evaluation PCL itself uses (feature_extractor.h:256-261), here
// on exact 2^-32 fixed-point int64 (order free, and exact modulo 2^64 even if a clamped outlier made
// a partial sum wrap), so the result is identical to the direct window walk of the gather kernel.
//   table: 7 planes x (th + 1) x (tw + 1) 8-byte cells, row 0 / column 0 are zero;
//          planes 0..5 = gx0 gx1 gx2 gy0 gy1 gy2, plane 6 = {count_x, count_y} as two int32
// ---------------------------------------------------------------------------------------------
constexpr int NF_TY = 8, NF_TX = 16, NF_APRON = 5;   // a window of <= 10 cells reaches 5 to the left / up of its sample point and 4 to the right / down
constexpr int NF_PARTS = 4;     // lanes per sample point: three take two gradient planes each, one the counts
constexpr int NF_PLANES = 7;
constexpr int NF_THREADS = NF_TY * NF_TX * NF_PARTS;

__global__ void __launch_bounds__(NF_THREADS)
normal_feature_tiled_kernel(FrameGeom g, const float4* __restrict__ cloud_all, const uint8_t* __restrict__ rect_all,
                            float* __restrict__ nfeat_all, int tiles_x, int tiles_y, int n_jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sat[];   // [NF_PLANES][th + 1][tw + 1]
    const int s = g.stride;
    const int tw = NF_TX * s + 2 * NF_APRON, th = NF_TY * s + 2 * NF_APRON;
    const int pw = tw + 1, ph = th + 1;           // padded plane: row 0 and column 0 stay zero
    const int plane = pw * ph;
    const int W = g.W, H = g.H;
    const int tid = threadIdx.x;
    // a block walks several tiles (grid = a few blocks per CU).  Measured per 64-frame launch with phases
    // switched off: cloud loads 215 us, fixed-point conversion 175, the two prefix passes 330, window
    // evaluation 105, the rest 160; 2 / 4 / 8 / 75 blocks per CU: 774 / 766 / 747 / 759 us -- turnover is not it
  for (int job = blockIdx.x; job < n_jobs; job += gridDim.x) {
    const int frame = job / (tiles_x * tiles_y);
    const int tile = job - frame * tiles_x * tiles_y;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int px0 = tx * NF_TX * s - NF_APRON, py0 = ty * NF_TY * s - NF_APRON;   // pixel of tile cell (0,0)
    const float4* cloud = cloud_all + (size_t)frame * W * H;
    // ---- gradients (PCL: central differences of the organised cloud), one padded cell per thread step
    for (int idx = tid; idx < plane; idx += NF_THREADS) {
        const int cy = idx / pw, cx = idx - cy * pw;
        long long gx0 = 0, gx1 = 0, gx2 = 0, gy0 = 0, gy1 = 0, gy2 = 0;
        unsigned cnt_x = 0, cnt_y = 0;
        const int y = py0 + cy - 1, x = px0 + cx - 1;
        if (cy > 0 && cx > 0 && y >= 0 && y < H && x >= 0 && x < W) {
            float dx0 = 0.f, dx1 = 0.f, dx2 = 0.f, dy0 = 0.f, dy1 = 0.f, dy2 = 0.f;
            if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
                const float4 r = cloud[(size_t)y * W + x + 1], l = cloud[(size_t)y * W + x - 1];
                const float4 dn = cloud[(size_t)(y + 1) * W + x], up = cloud[(size_t)(y - 1) * W + x];
                dx0 = r.x - l.x; dx1 = r.y - l.y; dx2 = r.z - l.z;
                dy0 = dn.x - up.x; dy1 = dn.y - up.y; dy2 = dn.z - up.z;
            }
            if (finite_f(dx0) && finite_f(dx1) && finite_f(dx2)) { cnt_x = 1; gx0 = to_fix32(dx0); gx1 = to_fix32(dx1); gx2 = to_fix32(dx2); }
            if (finite_f(dy0) && finite_f(dy1) && finite_f(dy2)) { cnt_y = 1; gy0 = to_fix32(dy0); gy1 = to_fix32(dy1); gy2 = to_fix32(dy2); }
        }
        sat[0 * plane + idx] = (unsigned long long)gx0; sat[1 * plane + idx] = (unsigned long long)gx1;
        sat[2 * plane + idx] = (unsigned long long)gx2; sat[3 * plane + idx] = (unsigned long long)gy0;
        sat[4 * plane + idx] = (unsigned long long)gy1; sat[5 * plane + idx] = (unsigned long long)gy2;
        sat[6 * plane + idx] = (unsigned long long)cnt_x | ((unsigned long long)cnt_y << 32);   // two int32 counters, no carry between them (<= 1232 each)
    }
    __syncthreads();
    // ---- prefix along x: one thread per (plane, row) walks its row (unsigned: wrap-around is harmless).
    //      (Reading a batch of cells into registers before writing any back, to keep several LDS reads in
    //      flight, measured 150 us slower per 64-frame launch than this plain loop.)
    for (int t = tid; t < NF_PLANES * th; t += NF_THREADS) {
        const int k = t / th, cy = t - k * th + 1;
        unsigned long long* row = sat + (size_t)k * plane + (size_t)cy * pw;
        unsigned long long run = 0;
#pragma unroll 4
        for (int cx = 1; cx < pw; cx++) { run += row[cx]; row[cx] = run; }
    }
    __syncthreads();
    // ---- prefix along y: one thread per (plane, column)
    for (int t = tid; t < NF_PLANES * tw; t += NF_THREADS) {
        const int k = t / tw, cx = t - k * tw + 1;
        unsigned long long* col = sat + (size_t)k * plane + cx;
        unsigned long long run = 0;
#pragma unroll 4
        for (int cy = 1; cy < ph; cy++) { run += col[(size_t)cy * pw]; col[(size_t)cy * pw] = run; }
    }
    __syncthreads();
    // ---- window sums: the NF_PARTS lanes of a sample are adjacent; lane `part` fetches two planes
    const int sample = tid / NF_PARTS, part = tid - sample * NF_PARTS;
    const int sy = sample / NF_TX, sx = sample - sy * NF_TX;
    const int ly_s = ty * NF_TY + sy, lx_s = tx * NF_TX + sx;   // sample grid coordinates
    const bool inside = ly_s < g.lh && lx_s < g.lw;
    const int ri = ly_s * s, ci = lx_s * s;
    const int border = 10;
    int rect = 0;
    if (inside && ri >= border && ri < H - border && ci >= border && ci < W - border) {
        rect = rect_all[(size_t)frame * W * H + (size_t)ri * W + ci];
        const float zc = cloud[(size_t)ri * W + ci].z;
        if (!finite_f(zc)) rect = 0;
    }
    unsigned long long a = 0, b = 0;     // this lane's two planes (the counter lane: a = {count_x, count_y})
    if (rect > 0) {
        const int rect2 = rect >> 1;
        // window = tile cells [wx, wx + rect) x [wy, wy + rect)  ->  padded cells +1; corners of the inclusive table
        const int wx = ci - rect2 - px0, wy = ri - rect2 - py0;
        const int x0 = wx, x1 = wx + rect, y0 = wy, y1 = wy + rect;     // padded coordinates: sum = S[y1][x1] - S[y0][x1] - S[y1][x0] + S[y0][x0]
        const int k0 = part < 3 ? 2 * part : 6;
        const unsigned long long* P = sat + (size_t)k0 * plane;
        a = P[(size_t)y1 * pw + x1] - P[(size_t)y0 * pw + x1] - P[(size_t)y1 * pw + x0] + P[(size_t)y0 * pw + x0];
        if (part < 3) {
            const unsigned long long* Q = P + plane;
            b = Q[(size_t)y1 * pw + x1] - Q[(size_t)y0 * pw + x1] - Q[(size_t)y1 * pw + x0] + Q[(size_t)y0 * pw + x0];
        } else {
            // the two packed int32 counters were subtracted as one 64-bit word: a borrow out of the low
            // counter's difference (never negative as a true count) cannot occur in the final sum, but
            // intermediate terms may borrow; recompute each half on its own
            const unsigned lo = (unsigned)P[(size_t)y1 * pw + x1] - (unsigned)P[(size_t)y0 * pw + x1] - (unsigned)P[(size_t)y1 * pw + x0] + (unsigned)P[(size_t)y0 * pw + x0];
            const unsigned hi = (unsigned)(P[(size_t)y1 * pw + x1] >> 32) - (unsigned)(P[(size_t)y0 * pw + x1] >> 32) -
                                (unsigned)(P[(size_t)y1 * pw + x0] >> 32) + (unsigned)(P[(size_t)y0 * pw + x0] >> 32);
            a = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
    }
    // gather the sample's seven sums into its first lane (the four lanes sit in one wave)
    const int lane = tid & 63, base = lane & ~(NF_PARTS - 1);
    const long long gx0 = (long long)__shfl(a, base + 0, 64), gx1 = (long long)__shfl(b, base + 0, 64);
    const long long gx2 = (long long)__shfl(a, base + 1, 64), gy0 = (long long)__shfl(b, base + 1, 64);
    const long long gy1 = (long long)__shfl(a, base + 2, 64), gy2 = (long long)__shfl(b, base + 2, 64);
    const unsigned long long cnt = __shfl(a, base + 3, 64);
    const int cnt_x = (int)(unsigned)cnt, cnt_y = (int)(unsigned)(cnt >> 32);
    if (inside && part == 0) {
        float out = -2.0f;
        if (rect > 0 && cnt_x > 0 && cnt_y > 0) {
            const double k = 1.0 / 4294967296.0;
            const double GX0 = (double)gx0 * k, GX1 = (double)gx1 * k, GX2 = (double)gx2 * k;
            const double GY0 = (double)gy0 * k, GY1 = (double)gy1 * k, GY2 = (double)gy2 * k;
            const double n0 = GY1 * GX2 - GY2 * GX1;
            const double n1 = GY2 * GX0 - GY0 * GX2;
            const double n2 = GY0 * GX1 - GY1 * GX0;
            const double len2 = (n0 * n0 + n1 * n1) + n2 * n2;
            if (len2 != 0.0) {
                const float nz = (float)(n2 / sqrt(len2));
                if (nz == nz) out = acos_f32_dev(fabsf(nz));
            }
        }
        nfeat_all[(size_t)frame * g.lw * g.lh + (size_t)ly_s * g.lw + lx_s] = out;
    }
    __syncthreads();   // the table is rewritten by the next tile
  }
}

// ---------------------------------------------------------------------------------------------
void launch_prep(const FrameGeom& g, const LabTables& lab, const uint8_t* d_rgb, const uint16_t* d_depth,
                 const float* d_calibA, uint32_t* d_lab, float4* d_cloud, uint8_t* d_change, int n, hipStream_t s, uint2* d_lab2) {
    LabCoeffs lc;
    for (int i = 0; i < 9; i++) lc.c[i] = lab.coeffs[i];
    const size_t total = (size_t)g.W * g.H * n;
    prep_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        g, lc, lab.gamma.as<uint16_t>(), lab.cbrt.as<uint16_t>(), d_rgb, d_depth, d_calibA, d_lab, d_cloud, d_change, n, d_lab ? d_lab2 : nullptr);
    RV_LAUNCHED("prep_kernel");
}

void launch_window_map(const FrameGeom& g, const float4* d_cloud, uint8_t* d_change, uint8_t* d_rect, int n, hipStream_t s) {
    const int tiles = ((g.W + DM_TW - 1) / DM_TW) * ((g.H + DM_TH - 1) / DM_TH);
    window_map_kernel<<<dim3((unsigned)(tiles * n)), dim3(64), 0, s>>>(g, d_change, d_rect);
    RV_LAUNCHED("window_map_kernel");
}

void launch_normal_feature(const FrameGeom& g, const float4* d_cloud, const uint8_t* d_rect, float* d_nfeat,
                           int n, hipStream_t s) {
    const int tw = NF_TX * g.stride + 2 * NF_APRON, th = NF_TY * g.stride + 2 * NF_APRON;
    // (A two-half variant -- x planes, then y planes, through one 36.5 KB table, four tiles per CU -- was built and
    // measured: the kernel's overlapped time fell from 2.0 to 1.2 ms, the lattice build beside it rose from 4.0 to
    // 5.5 ms and became the critical path: 13.74 vs 13.38 ms per step.  The two branches share the chip work for
    // work; only less work helps.  Removed.)
    const size_t lds = (size_t)(tw + 1) * (th + 1) * NF_PLANES * 8;
    if (lds <= 80 * 1024) {   // stride <= 2 (65 KB: two tiles per CU); larger strides use the gather kernel
        static bool attr_set[64] = {};   // per device: more than the default 64 KB of dynamic LDS has to be asked for
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(normal_feature_tiled_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        const int tiles_x = (g.lw + NF_TX - 1) / NF_TX, tiles_y = (g.lh + NF_TY - 1) / NF_TY;
        const int n_jobs = tiles_x * tiles_y * n;
        // one tile per block: a grid of 2..8 long-lived blocks per CU was measured to run this kernel alone as fast
        // (747-774 vs 759 us) but keeps the lattice build on the side stream off the CUs for its whole duration
        // (build 3.5 -> 5.3 ms, step +0.3 ms)
        normal_feature_tiled_kernel<<<dim3((unsigned)n_jobs), dim3(NF_THREADS), lds, s>>>(
            g, d_cloud, d_rect, d_nfeat, tiles_x, tiles_y, n_jobs);
        RV_LAUNCHED("normal_feature_tiled_kernel");
        return;
    }
    const int total = g.lw * g.lh * n;
    normal_feature_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(g, d_cloud, d_rect, d_nfeat, n);
    RV_LAUNCHED("normal_feature_kernel");
}

}  // namespace rvseg
