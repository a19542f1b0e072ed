// CRF orchestration: lattice construction, normaliser, mean-field loop; the C-ABI entry points
// rvseg_crf_infer / rvseg_crf_infer_multi / rvseg_lattice_build / rvseg_lattice_filter and the
// per-frame CRF stage of the frame pipeline.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rvseg_crf.h"
#include "rvseg_pipeline.h"

namespace rvseg {

struct LatticeBufs {
    DevBuf state, tkeys, slot_to_id, counters, vkeys, offsets, bary, nb1, nb2, csr_pw, csr_nrm, vstart, vend, norm;
    DevBuf keys_in, keys_out, vals_in, vals_out, sort_temp, scan_temp, fstart, vorder, block_hist;
    DevBuf r_desc, r_vl, r_info, r_small, r_verts, r_jb, r_trace;   // resident band schedule of the splat
    SplatResidentDev resident{};
    bool resident_on = false;
    LatticeDev dev{};
    SortBuffers sb{};
    long long n_entries = 0, n_points = 0;
    bool built = false;
    bool cleared = false;       // the build's memsets are already enqueued (crf_frames_build_begin)
    bool has_csr_nrm = false;   // per-entry normaliser (multi-kernel inference only)
};

struct CrfState {
    std::vector<LatticeBufs> lat;  // one per pairwise kernel
    // slot 1 of the scratch buffers: a second label layer's mean field runs beside the first on its own stream
    DevBuf val_a, val_b, tmp, q, qn, unary, feat, labels, val_a2, val_b2, tmp2, qn2;
    hipStream_t layer_stream = nullptr;       // the second layer's stream
    hipEvent_t layer_fork = nullptr, layer_join = nullptr;
    // pinned read-back of a build: [0] M, [1] overflow, [2] longest vertex list, [3] frames the splat planner gave up on.
    // Slot 0 (words 0..3) belongs to the asynchronous frame builds (consumed by crf_frames_status), slot 1 (words 4..7)
    // to the synchronous entry points -- a cloud or host CRF call on the same context must not overwrite a frame
    // build's status that nobody has polled yet.
    int* h_counters = nullptr;
    hipEvent_t counters_ev = nullptr;
    bool counters_pending = false;
    rvseg_schedule_info info{};    // what the last build ran with (rvseg_last_schedule)
    int frame_vertices_seen = 0;   // vertices per frame of the last frame build whose status was read (0: none yet)
    int pending_frames = 0;        // frames of the build whose status is pending
    bool info_async = false;       // info.vertices / planner_fallback still travel with the pending frame-build status
};

static rvseg_status crf_state(rvseg_ctx* ctx, Pipeline* im, CrfState** out) {
    if (!im->crf) {
        CrfState* cs = new CrfState();
        if (!hip_ok(ctx, hipHostMalloc((void**)&cs->h_counters, 8 * sizeof(int), hipHostMallocDefault), "hipHostMalloc(counters)") ||
            !hip_ok(ctx, hipEventCreateWithFlags(&cs->counters_ev, hipEventDisableTiming), "hipEventCreate(counters)")) {
            if (cs->h_counters) (void)hipHostFree(cs->h_counters);
            delete cs;
            return RVSEG_ERR_HIP;
        }
        for (int i = 0; i < 8; i++) cs->h_counters[i] = 0;
        im->crf = cs;
    }
    *out = im->crf;
    return RVSEG_OK;
}

static void lattice_free(LatticeBufs& b) {
    DevBuf* all[] = {&b.state, &b.tkeys, &b.slot_to_id, &b.counters, &b.vkeys, &b.offsets, &b.bary, &b.nb1, &b.nb2,
                     &b.csr_pw, &b.csr_nrm, &b.vstart, &b.vend, &b.norm, &b.keys_in, &b.keys_out, &b.vals_in,
                     &b.vals_out, &b.sort_temp, &b.scan_temp, &b.fstart, &b.vorder, &b.block_hist,
                     &b.r_desc, &b.r_vl, &b.r_info, &b.r_small, &b.r_verts, &b.r_jb, &b.r_trace};
    for (DevBuf* x : all) dev_free(*x);
}

void crf_state_free(Pipeline* im) {
    if (!im->crf) return;
    for (auto& l : im->crf->lat) lattice_free(l);
    DevBuf* all[] = {&im->crf->val_a, &im->crf->val_b, &im->crf->tmp, &im->crf->q, &im->crf->qn, &im->crf->unary, &im->crf->feat, &im->crf->labels,
                     &im->crf->val_a2, &im->crf->val_b2, &im->crf->tmp2, &im->crf->qn2};
    if (im->crf->layer_stream) (void)hipStreamDestroy(im->crf->layer_stream);
    if (im->crf->layer_fork) (void)hipEventDestroy(im->crf->layer_fork);
    if (im->crf->layer_join) (void)hipEventDestroy(im->crf->layer_join);
    for (DevBuf* x : all) dev_free(*x);
    if (im->crf->h_counters) (void)hipHostFree(im->crf->h_counters);
    if (im->crf->counters_ev) (void)hipEventDestroy(im->crf->counters_ev);
    delete im->crf;
    im->crf = nullptr;
}

static int ceil_log2(unsigned long long v) {
    int b = 0;
    while ((1ull << b) < v) b++;
    return b;
}

// per-frame hash capacity: params.lattice_capacity_log2 > 0 as given; 0 = 2^12; < 0 or `safe` =
// enough for every point to own d+1 private vertices at load factor 1/2.  Every overflow seen on this
// context has raised cap_boost by 3 (x8 slots), so a scene that does not fit is paid for once.
static int capacity_log2_per_frame(const rvseg_ctx* ctx, int Npad, int d, bool safe) {
    const int safe_log2 = ceil_log2(2ull * (unsigned long long)Npad * (d + 1));
    int want = ctx->params.lattice_capacity_log2;
    if (safe || want < 0) return safe_log2;
    if (want == 0) want = 12;   // the Segmenter kernel yields ~300 vertices per synthetic 640x480 frame
    if (ctx->impl) want += reinterpret_cast<const Pipeline*>(ctx->impl)->cap_boost;
    return want < safe_log2 ? want : safe_log2;
}

static bool capacity_is_worst_case(const rvseg_ctx* ctx, int N, int d) {
    const int Npad = (N + 3) / 4 * 4;
    return capacity_log2_per_frame(ctx, Npad, d, false) == capacity_log2_per_frame(ctx, Npad, d, true);
}

// Does the resident band schedule pay for a chunk of this shape?  It wins where the list-major walk is bound by the
// bytes it re-reads (many frames) and loses where a frame's chains set the time (few frames, a cloud) or where its
// planner -- whose work grows with the vertices of a frame -- costs more than the splat saves.  Measured crossovers
// (profiles/r03_schedule_sweep.json, step time of the whole frame path, 640x480 frames with holes): ~360 vertices per
// frame (the flat synthetic scene): between 16 and 32 frames (resident 5.14 / 7.53 / 12.59 ms against 4.85 / 7.94 /
// 13.97 at 16 / 32 / 64 frames); ~2 320 vertices per frame (the deep scene): between 32 and 64 frames (12.35 / 18.70
// against 10.52 / 19.22).  A line through the two crossovers: resident from 5.6 M points + 4 900 points per vertex of a
// frame -- moved to 6.9 M + 4 900 per vertex when the list-major walk got scan blocks for its longest lists (24 flat
// frames: list-major 5.49 ms against 6.04 resident; 32 frames: 7.15 against 7.03).  The vertex count is a MEASURED quantity: that of the context's previous lattice build (read back with its
// status); before any build has been seen, the flat scene's.
static bool resident_pays(int n_frames, int N, int vertices_per_frame_seen) {
    const long long vpf = vertices_per_frame_seen > 0 ? vertices_per_frame_seen : 360;
    return n_frames >= 2 && (long long)n_frames * N >= 6900000ll + 4900ll * vpf;
}

static rvseg_status lattice_prepare(rvseg_ctx* ctx, LatticeBufs& b, int d, int N, int n_frames, bool safe, int vertices_per_frame_seen = 0) {
    if (d < 1 || d > 7) { ctx->err = "feature dimension must be in [1,7]"; return RVSEG_ERR_INVALID_ARG; }
    if (n_frames > 1022) { ctx->err = "at most 1022 frames per chunk (lower max_batch)"; return RVSEG_ERR_INVALID_ARG; }   // 10-bit frame field of the launch-order sort key
    const int Npad = (N + 3) / 4 * 4;
    const int cap_f_log2 = capacity_log2_per_frame(ctx, Npad, d, safe);
    const unsigned long long cap = (unsigned long long)n_frames << cap_f_log2;
    if (cap >= (1ull << 31)) { ctx->err = "lattice hash capacity too large (lower max_batch or lattice_capacity_log2)"; return RVSEG_ERR_CAPACITY; }
    const unsigned long long worst = (unsigned long long)Npad * n_frames * (d + 1);
    const unsigned long long m_bound = std::min<unsigned long long>(cap / 2 + 2, worst);
    const long long P = (long long)N * n_frames;
    const long long E = P * (d + 1);
    if (E >= (1ll << 32) || m_bound >= (1ull << 31)) { ctx->err = "too many lattice entries for 32-bit indices"; return RVSEG_ERR_CAPACITY; }
    rvseg_status st;
#define RV_RES(buf, bytes) if ((st = dev_reserve(ctx, buf, (size_t)(bytes))) != RVSEG_OK) return st
    RV_RES(b.state, cap * 4);
    RV_RES(b.tkeys, cap * 16);
    RV_RES(b.slot_to_id, cap * 4);
    RV_RES(b.fstart, ((size_t)n_frames + 1) * 4);
    RV_RES(b.vkeys, m_bound * 16);
    RV_RES(b.offsets, E * 4);
    RV_RES(b.bary, E * 4);
    RV_RES(b.nb1, m_bound * (d + 1) * 4);
    RV_RES(b.nb2, m_bound * (d + 1) * 4);
    RV_RES(b.csr_pw, E * 8);
    RV_RES(b.vstart, (2 * m_bound + 4) * 4);   // vstart | vend | counters: one allocation, zeroed by ONE memset per build
    RV_RES(b.vorder, m_bound * 4);
    RV_RES(b.norm, P * 4);
    const unsigned long long SE = std::max<unsigned long long>((unsigned long long)E, m_bound);  // also sorts the vertex order
    RV_RES(b.keys_in, SE * 4);
    RV_RES(b.keys_out, SE * 4);
    RV_RES(b.vals_in, SE * 4);
    RV_RES(b.vals_out, SE * 4);
    const int key_bits = std::max(1, ceil_log2(m_bound));
    const size_t temp = std::max(sort_temp_bytes(E, key_bits), sort_temp_bytes((long long)m_bound, 32));
    RV_RES(b.sort_temp, temp);
    const size_t stemp = scan_temp_bytes((unsigned)cap);
    RV_RES(b.scan_temp, stemp);
#undef RV_RES
    LatticeDev& L = b.dev;
    L.d = d; L.N = N; L.Npad = Npad; L.n_frames = n_frames;
    L.cap_f_log2 = (unsigned)cap_f_log2;
    L.cap_f_mask = (1u << cap_f_log2) - 1u;
    L.cap_total = (unsigned)cap;
    L.m_bound = (int)m_bound;
    // diagonal of E (permutohedral.cpp:177-182): float inv_std_dev; scale = 1/sqrt((i+2)(i+1)) * inv_std_dev
    const float inv_std_dev = (float)(std::sqrt(2.0 / 3.0) * (d + 1));
    for (int i = 0; i < 8; i++) L.scale[i] = i < d ? (float)(1.0 / std::sqrt((double)((i + 2) * (i + 1))) * inv_std_dev) : 0.f;
    L.state = b.state.as<int>(); L.tkeys = b.tkeys.as<unsigned long long>(); L.slot_to_id = b.slot_to_id.as<int>();
    L.fstart = b.fstart.as<int>(); L.vkeys = b.vkeys.as<unsigned long long>();
    L.offsets = b.offsets.as<int>(); L.bary = b.bary.as<float>();
    L.nb1 = b.nb1.as<int>(); L.nb2 = b.nb2.as<int>();
    L.csr_pw = b.csr_pw.as<uint2>(); L.csr_nrm = nullptr;
    b.has_csr_nrm = false;
    L.vstart = b.vstart.as<unsigned>(); L.vend = L.vstart + m_bound; L.counters = reinterpret_cast<int*>(L.vend + m_bound);
    L.vorder = b.vorder.as<unsigned>(); L.norm = b.norm.as<float>();
    L.n_groups = n_frames >= 8 ? 8 : 1;
    b.sb.keys_in = b.keys_in.as<unsigned>(); b.sb.keys_out = b.keys_out.as<unsigned>();
    b.sb.vals_in = b.vals_in.as<unsigned>(); b.sb.vals_out = b.vals_out.as<unsigned>();
    b.sb.temp = b.sort_temp.p; b.sb.temp_bytes = temp; b.sb.key_bits = key_bits;
    b.sb.scan_temp = b.scan_temp.p; b.sb.scan_temp_bytes = stemp;
    b.sb.block_hist = nullptr;
    L.bh = nullptr; L.wbpf = 0;
    L.group_vertices = ctx->sched.group_vertices;
    L.ordered_sum_scan = ctx->sched.serial_chains ? 0 : 1;
    L.heavy_from = 0;
    L.scan_ranks = 0;
    // Wave-blocks of the counting sort.  Its [wave-block][vertex] count matrix is written once and read three times, and
    // its size is points / cs_pix x vertices: 1024-point blocks cut a 64-frame step by 0.15 ms (deep scene: 1.3 ms; 32
    // frames 0.16, 16 frames 0.11), but a single frame then has only 300 waves to sort with (+0.07 ms) and 8 frames
    // gain nothing, so small launches keep 256 (sweep of 256 .. 4096 over 8 - 64 frames: scratch-style script in
    // DESIGN.md section 4; results are identical for every size)
    L.cs_pix = ctx->sched.csr_block > 0 ? ctx->sched.csr_block : (n_frames <= 8 ? 256 : 1024);
    if (csr_fast_path(L)) {
        if ((st = dev_reserve(ctx, b.block_hist, csr_fast_bytes(L))) != RVSEG_OK) return st;
        b.sb.block_hist = b.block_hist.as<unsigned>();
        L.bh = b.sb.block_hist;
        L.wbpf = (N + L.cs_pix - 1) / L.cs_pix;
    }
    b.n_entries = E; b.n_points = P;
    b.built = false;
    b.cleared = false;
    // Resident band schedule of the mean-field splat: chunks of many frames, whose splat is
    // bound by the bytes the list-major walk re-reads.  All n_frames x B blocks have to be on the chip together.
    b.resident_on = false;
    {
        const rvseg_schedule& sc = ctx->sched;
        // Worth it where the list-major walk is bound by the bytes it re-reads rather than by its longest chain
        // ("where it pays"): see resident_pays().  sched.splat = 2 forces it, 1 forbids it.
        const bool wanted = sc.splat == 2 || (sc.splat == 0 && resident_pays(n_frames, N, vertices_per_frame_seen));
        const int chunk = sc.resident_chunk == 64 ? 64 : 128;
        const int capacity = resident_block_capacity(chunk);
        // one block per CU measured best (the tile loop is bound by its own barrier-coupled latencies, a second block on
        // the CU slows both): B = CUs / frames, at least 2, at most 12
        int B = sc.resident_blocks > 0 ? sc.resident_blocks : (n_frames > 0 ? resident_cu_count() / n_frames : 0);
        if (sc.resident_blocks <= 0) B = B < 2 ? 2 : (B > 12 ? 12 : B);
        B = B > RES_MAXB ? RES_MAXB : B;
        if (wanted && L.bh && d == 6 && B >= 2 && (long long)n_frames * B <= capacity &&
            7ll * N < (1ll << 24) && (long long)L.wbpf * L.cs_pix <= 8192ll * RES_MAX_BANDS && L.cs_pix <= 4096) {   // (bands stay under 16 384 points: chunk counts fit 8 bits)
            SplatResidentDev& R = b.resident;
            R.B = B;
            R.band_wb = sc.resident_band < 1 ? 1 : (sc.resident_band > 32 ? 32 : sc.resident_band);
            R.band_wb = std::max(1, R.band_wb * 256 / L.cs_pix);   // rvseg_schedule.resident_band counts 256 points
            R.n_bands = (L.wbpf + R.band_wb - 1) / R.band_wb;
            while (R.n_bands > RES_MAX_BANDS) { R.band_wb *= 2; R.n_bands = (L.wbpf + R.band_wb - 1) / R.band_wb; }
            R.window = sc.resident_window;
            R.chunk_log2 = chunk == 128 ? 7 : 6;
            R.cap_tiles = (unsigned)(N / 8 + 1024);
            if (sc.resident_cap_tiles > 0 && (unsigned)sc.resident_cap_tiles < R.cap_tiles) R.cap_tiles = (unsigned)sc.resident_cap_tiles;
            const size_t small = (16 + (size_t)n_frames * (RES_MAXB + 1) + (size_t)n_frames * RES_MAXB + 2 * (size_t)n_frames * RES_MAXB) * 4;
            if ((st = dev_reserve(ctx, b.r_desc, (size_t)n_frames * 7 * R.cap_tiles * 4)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, b.r_vl, (size_t)n_frames * 7 * R.cap_tiles * 2)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, b.r_info, (size_t)n_frames * R.cap_tiles * 4)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, b.r_small, small)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, b.r_verts, (size_t)n_frames * RES_MAXB * RES_MAX_OWNV * 2)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, b.r_jb, (size_t)n_frames * RES_MAXB * (R.n_bands + 1) * 4)) != RVSEG_OK) return st;
            R.tdesc = b.r_desc.as<unsigned>(); R.tvl = b.r_vl.as<unsigned short>(); R.tinfo = b.r_info.as<unsigned>();
            unsigned* sm = b.r_small.as<unsigned>();
            R.flags = reinterpret_cast<int*>(sm); sm += 16;
            R.blk_tile0 = sm; sm += (size_t)n_frames * (RES_MAXB + 1);
            R.blk_nown = sm; sm += (size_t)n_frames * RES_MAXB;
            R.prog = sm;
            R.blk_verts = b.r_verts.as<unsigned short>();
            R.jb_tile = b.r_jb.as<unsigned>();
            R.trace = nullptr;
            if (sc.trace) {
                if ((st = dev_reserve(ctx, b.r_trace, (size_t)n_frames * RES_MAXB * 64)) != RVSEG_OK) return st;
                R.trace = b.r_trace.as<unsigned long long>();
            }
            b.resident_on = true;
        }
    }
    return RVSEG_OK;
}

static rvseg_status values_reserve(rvseg_ctx* ctx, CrfState* cs, long long m_bound, int C, int slot = 0) {
    rvseg_status st;
    if ((st = dev_reserve(ctx, slot ? cs->val_a2 : cs->val_a, (size_t)m_bound * C * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, slot ? cs->val_b2 : cs->val_b, (size_t)m_bound * C * 4)) != RVSEG_OK) return st;
    return RVSEG_OK;
}

// the context's second CRF stream (the second label layer's mean field; the splat planner beside the normaliser)
static rvseg_status second_stream(rvseg_ctx* ctx, CrfState* cs) {
    if (cs->layer_stream) return RVSEG_OK;
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    RV_HIP(ctx, hipStreamCreateWithPriority(&cs->layer_stream, hipStreamNonBlocking, prio_hi));
    RV_HIP(ctx, hipEventCreateWithFlags(&cs->layer_fork, hipEventDisableTiming));
    RV_HIP(ctx, hipEventCreateWithFlags(&cs->layer_join, hipEventDisableTiming));
    return RVSEG_OK;
}

// Permutohedral::init + the normaliser of DenseKernel::initLattice (pairwise.cpp:40-56)
static rvseg_status lattice_clear(rvseg_ctx* ctx, LatticeBufs& b, hipStream_t s) {
    const LatticeDev& L = b.dev;
    RV_HIP(ctx, hipMemsetAsync(L.state, 0xFF, (size_t)L.cap_total * 4, s));
    RV_HIP(ctx, hipMemsetAsync(L.vstart, 0, (2 * (size_t)L.m_bound + 4) * 4, s));   // vstart, vend, counters
    b.cleared = true;
    return RVSEG_OK;
}

static rvseg_status lattice_build(rvseg_ctx* ctx, CrfState* cs, LatticeBufs& b, const FeatureSource& fs, hipStream_t s) {
    const LatticeDev& L = b.dev;
    if (!b.cleared) { rvseg_status stc = lattice_clear(ctx, b, s); if (stc != RVSEG_OK) return stc; }
    b.cleared = false;
    const bool trace = ctx->sched.trace >= 2;
    auto tr = [&](const char* what) {
        if (!trace) return;
        hipError_t e = hipStreamSynchronize(s);
        std::fprintf(stderr, "[rvseg] %s: %s (N=%d d=%d cap_f=%u m_bound=%d)\n", what, hipGetErrorString(e), L.N, L.d, L.cap_f_mask + 1, L.m_bound);
    };
    tr("memsets");
    launch_lattice_points(L, fs, s);
    tr("points");
    // the splat's band schedule needs the lists' bounds (count / scan) and the launch order, not the lists themselves: it
    // is planned beside the scatter and the normaliser
    launch_lattice_finish(L, b.sb, b.n_entries, s, b.resident_on ? 1 : 0);
    tr("finish");
    rvseg_status st;
    bool plan_forked = false;
    if (b.resident_on) {
        if ((st = second_stream(ctx, cs)) != RVSEG_OK) return st;
        RV_HIP(ctx, hipEventRecord(cs->layer_fork, s));
        RV_HIP(ctx, hipStreamWaitEvent(cs->layer_stream, cs->layer_fork, 0));
        RV_HIP(ctx, hipMemsetAsync(b.resident.prog, 0, 2 * (size_t)L.n_frames * RES_MAXB * 4, cs->layer_stream));
        launch_resident_plan(L, b.resident, cs->layer_stream);
        RV_HIP(ctx, hipEventRecord(cs->layer_join, cs->layer_stream));
        plan_forked = true;
        tr("resident plan");
        launch_lattice_finish(L, b.sb, b.n_entries, s, 2);   // the scatter (counting-sort path)
    }
    st = values_reserve(ctx, cs, L.m_bound, 1);
    if (st != RVSEG_OK) { if (plan_forked) (void)hipStreamWaitEvent(s, cs->layer_join, 0); return st; }
    // norm = lattice.compute(ones) through seqCompute (1 row), then 1/sqrt(norm + 1e-20)
    ValueView none{nullptr, 0, 0};
    launch_splat(L, none, 1, 2, cs->val_a.as<float>(), s);
    float* blurred = launch_blur(L, 1, true, false, cs->val_a.as<float>(), cs->val_b.as<float>(), s, true);
    launch_slice(L, 1, true, 1, blurred, 0.f, L.norm, b.n_points, s);
    tr("normaliser");
    if (plan_forked) RV_HIP(ctx, hipStreamWaitEvent(s, cs->layer_join, 0));
    RV_LAUNCH_OK(ctx);
    b.built = true;
    rvseg_schedule_info& inf = cs->info;
    inf.splat = b.resident_on ? 2 : 1;
    inf.planner_fallback = -1;
    inf.csr_path = L.bh ? 1 : 2;
    inf.n_frames = L.n_frames;
    inf.points_per_frame = L.N;
    inf.vertices = -1;
    inf.longest_list = -1;
    inf.resident_blocks = b.resident_on ? b.resident.B : 0;
    inf.resident_band = b.resident_on ? b.resident.band_wb : 0;
    inf.resident_chunk = b.resident_on ? (1 << b.resident.chunk_log2) : 0;
    inf.capacity_log2 = (int)L.cap_f_log2;
    return RVSEG_OK;
}

// enqueues the read-back of a build's counters (+ the planner's flag) into pinned slot `slot` (0: async frame builds,
// 1: synchronous entry points)
static rvseg_status counters_readback(rvseg_ctx* ctx, CrfState* cs, const LatticeBufs& b, int slot, hipStream_t s) {
    int* h = cs->h_counters + 4 * slot;
    RV_HIP(ctx, hipMemcpyAsync(h, b.dev.counters, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipMemcpyAsync(h + 2, b.dev.counters + 3, sizeof(int), hipMemcpyDeviceToHost, s));
    if (b.resident_on) RV_HIP(ctx, hipMemcpyAsync(h + 3, b.resident.flags, sizeof(int), hipMemcpyDeviceToHost, s));
    else h[3] = 0;   // (host write; no copy of this build touches the word)
    return RVSEG_OK;
}

// synchronous read of the build counters (host entry points)
static rvseg_status lattice_counters(rvseg_ctx* ctx, CrfState* cs, const LatticeBufs& b, hipStream_t s, int out[3]) {
    rvseg_status st = counters_readback(ctx, cs, b, 1, s);
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipStreamSynchronize(s));
    const int* h = cs->h_counters + 4;
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
    cs->info.vertices = h[0];
    cs->info.longest_list = h[2];
    cs->info.planner_fallback = h[3];
    cs->info_async = false;
    return RVSEG_OK;
}

// DenseKernel::filter + PottsCompatibility::apply folded into tmp (pairwise.cpp:63-80,173-178)
static void filter_into(rvseg_ctx* ctx, const LatticeBufs& b, CrfState* cs, const ValueView& Q, int C, float w, float* tmp, hipStream_t s) {
    const bool seq = C <= 2;  // Permutohedral::compute dispatch, permutohedral.cpp:600-603
    timer_mark(ctx, "splat", s);
    launch_splat(b.dev, Q, C, 1, cs->val_a.as<float>(), s);
    timer_mark(ctx, "blur", s);
    float* blurred = launch_blur(b.dev, C, seq, false, cs->val_a.as<float>(), cs->val_b.as<float>(), s);
    timer_mark(ctx, "slice", s);
    launch_slice(b.dev, C, seq, 2, blurred, -w, tmp, b.n_points, s);
}

// per-entry copy of the normaliser for the unfused splat (MODE 1); filled once per lattice
static rvseg_status ensure_csr_nrm(rvseg_ctx* ctx, LatticeBufs& b, hipStream_t s) {
    if (b.has_csr_nrm) return RVSEG_OK;
    rvseg_status st = dev_reserve(ctx, b.csr_nrm, (size_t)b.n_entries * 4);
    if (st != RVSEG_OK) return st;
    b.dev.csr_nrm = b.csr_nrm.as<float>();
    launch_csr_norm(b.dev, b.n_entries, s);
    b.has_csr_nrm = true;
    return RVSEG_OK;
}

// Layers that run side by side on two streams share the lattice's csr_nrm table: when any of them takes the unfused
// path (no fused instantiation for its class count) the table is filled on the PARENT stream before the fork, so the
// stream that did not enqueue the fill cannot read it early.
static rvseg_status csr_nrm_before_fork(rvseg_ctx* ctx, CrfState* cs, int n_layers, const int* class_counts, int iterations, hipStream_t s) {
    for (int l = 0; l < n_layers; l++)
        if (!(iterations > 0 && mf_fused_supported(class_counts[l]))) return ensure_csr_nrm(ctx, cs->lat[0], s);
    return RVSEG_OK;
}

// DenseCRF::inference (densecrf.cpp:115-131)
// `lab` (optional): where the last fused update may write the labels; *labels_done tells whether it did
// slot: which set of scratch buffers (0 / 1; two layers may run side by side on two streams);
// timed: record stage marks (only one of two concurrent loops may: the marks are a sequence on ONE stream)
static rvseg_status mean_field(rvseg_ctx* ctx, CrfState* cs, int n_kernels, const float* ws, const ValueView& unary,
                               bool unary_is_energy, int C, int N, long long n_points, int iterations,
                               const ValueView& Q, hipStream_t s, const MfLabels* lab = nullptr, bool* labels_done = nullptr,
                               int slot = 0, bool timed = true) {
    if (labels_done) *labels_done = false;
    rvseg_status st;
    DevBuf& b_tmp = slot ? cs->tmp2 : cs->tmp;
    DevBuf& b_qn = slot ? cs->qn2 : cs->qn;
    DevBuf& b_va = slot ? cs->val_a2 : cs->val_a;
    DevBuf& b_vb = slot ? cs->val_b2 : cs->val_b;
    auto mark = [&](const char* name) { if (timed) timer_mark(ctx, name, s); };
    if ((st = dev_reserve(ctx, b_tmp, (size_t)n_points * C * 4)) != RVSEG_OK) return st;
    long long mb = 0;
    for (int k = 0; k < n_kernels; k++) mb = std::max<long long>(mb, cs->lat[k].dev.m_bound);
    if ((st = values_reserve(ctx, cs, mb, C, slot)) != RVSEG_OK) return st;
    float* tmp = b_tmp.as<float>();
    // Single Potts kernel with a fused instantiation: between iterations Q holds fl(Q * norm), the
    // input of the next splat (pairwise.cpp:66), so the splat is a plain gather; only the last
    // update stores the marginals themselves.
    bool fused = n_kernels == 1 && iterations > 0;
    // Where the fused loop keeps fl(Q * norm) between iterations: Q itself when that is one contiguous
    // [point][C] matrix, else (a layer inside the reference's [layer][y][x][class] frames) a contiguous scratch
    // matrix -- the splat then addresses a row as point * C without splitting the point index per frame, and
    // its gathers stay inside one array.  The last update writes the marginals into Q in either case.
    ValueView Qs = Q;
    if (fused && !(Q.frame_stride == (size_t)N * (size_t)C && Q.layer_off == 0)) {
        if ((st = dev_reserve(ctx, b_qn, (size_t)n_points * C * 4)) != RVSEG_OK) return st;
        Qs = ValueView{b_qn.as<float>(), (size_t)N * (size_t)C, 0};
    }
    mark("softmax");
    if (!launch_softmax_unary(unary, unary_is_energy, C, N, fused ? Qs : Q, n_points, fused ? cs->lat[0].dev.norm : nullptr, s)) {
        fused = false;
        launch_neg_unary(unary, unary_is_energy, C, N, tmp, n_points, s);
        launch_softmax(tmp, C, N, Q, n_points, s);
    }
    if (!fused) {
        // the general path scales by the normaliser inside the splat: per-entry copy of norm
        for (int k = 0; k < n_kernels; k++)
            if ((st = ensure_csr_nrm(ctx, cs->lat[k], s)) != RVSEG_OK) return st;
    }
    for (int it = 0; it < iterations; it++) {
        if (n_kernels == 1) {
            // single Potts kernel: splat, blur, then one fused slice + update + softmax pass
            const LatticeBufs& b = cs->lat[0];
            const bool seq = C <= 2;
            mark("splat");
            launch_splat(b.dev, fused ? Qs : Q, C, fused ? 0 : 1, b_va.as<float>(), s, fused, b.resident_on ? &b.resident : nullptr, slot);
            mark("blur");
            float* blurred = launch_blur(b.dev, C, seq, false, b_va.as<float>(), b_vb.as<float>(), s);
            if (fused) {
                mark("mf_update");
                const bool last = it + 1 == iterations;
                MfLabels none{nullptr, 0, 0, 0, 0};
                launch_mf_update(b.dev, C, blurred, -ws[0], unary, unary_is_energy, last ? Q : Qs, !last, last && lab ? *lab : none, s);
                if (last && lab && labels_done) *labels_done = true;
                continue;
            }
            mark("softmax");
            launch_neg_unary(unary, unary_is_energy, C, N, tmp, n_points, s);
            mark("slice");
            launch_slice(b.dev, C, seq, 2, blurred, -ws[0], tmp, n_points, s);
            mark("softmax");
            launch_softmax(tmp, C, N, Q, n_points, s);
            continue;
        }
        mark("softmax");
        launch_neg_unary(unary, unary_is_energy, C, N, tmp, n_points, s);
        for (int k = 0; k < n_kernels; k++) filter_into(ctx, cs->lat[k], cs, Q, C, ws[k], tmp, s);
        mark("softmax");
        launch_softmax(tmp, C, N, Q, n_points, s);
    }
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

// The label layers of one lattice are independent mean fields (the reference runs one DenseCRF per layer,
// segmenter.cpp:639-644).  Their splats wait for their longest chains rather than for bandwidth whenever a chunk
// has few frames (a cloud, a 1280x960 chunk), so odd layers run on a second stream beside the even ones.
// li-th layer to enqueue: odd layers (second stream) first when the layers run on two streams
static int layer_enqueue_order(int li, int n_layers, bool two_streams) {
    if (!two_streams) return li;
    const int n_odd = n_layers / 2;
    return li < n_odd ? 2 * li + 1 : 2 * (li - n_odd);
}

static rvseg_status layer_stream_fork(rvseg_ctx* ctx, CrfState* cs, hipStream_t s, int n_layers, hipStream_t* s2) {
    *s2 = s;
    if (n_layers < 2 || !ctx->sched.overlap_layers) return RVSEG_OK;
    rvseg_status st = second_stream(ctx, cs);
    if (st != RVSEG_OK) return st;
    RV_HIP(ctx, hipEventRecord(cs->layer_fork, s));
    RV_HIP(ctx, hipStreamWaitEvent(cs->layer_stream, cs->layer_fork, 0));
    *s2 = cs->layer_stream;
    return RVSEG_OK;
}

static rvseg_status layer_stream_join(rvseg_ctx* ctx, CrfState* cs, hipStream_t s, hipStream_t s2) {
    if (s2 == s) return RVSEG_OK;
    RV_HIP(ctx, hipEventRecord(cs->layer_join, s2));
    RV_HIP(ctx, hipStreamWaitEvent(s, cs->layer_join, 0));
    return RVSEG_OK;
}

// ---------------------------------------------------------------------------------------------
// per-frame CRF stage of the frame pipeline
// ---------------------------------------------------------------------------------------------
rvseg_status crf_frames_status(rvseg_ctx* ctx, Pipeline* im, bool wait) {
    CrfState* cs = im->crf;
    if (!cs || !cs->counters_pending) return RVSEG_OK;
    if (wait) {
        RV_HIP(ctx, hipEventSynchronize(cs->counters_ev));
    } else {
        const hipError_t e = hipEventQuery(cs->counters_ev);
        if (e == hipErrorNotReady) return RVSEG_NOT_READY;
        RV_HIP(ctx, e);
    }
    cs->counters_pending = false;
    if (cs->pending_frames > 0 && !cs->h_counters[1]) cs->frame_vertices_seen = cs->h_counters[0] / cs->pending_frames;
    if (cs->info_async) {   // no other lattice has been built on this context since
        cs->info.vertices = cs->h_counters[0];
        cs->info.longest_list = cs->h_counters[2];
        cs->info.planner_fallback = cs->h_counters[3];
        cs->info_async = false;
    }
    if (cs->h_counters[1]) {
        const FrameGeom& g = im->geom;
        const bool was_worst = capacity_is_worst_case(ctx, g.W * g.H, 6);
        // x8 slots per step, but stop at 2^13 on the way up: the largest capacity the counting-sort CSR path serves
        // (real scenes with a deep range have ~2 000 vertices per frame; beyond it the radix-sort path takes over)
        {
            int base = ctx->params.lattice_capacity_log2 > 0 ? ctx->params.lattice_capacity_log2 : 12;
            const int cur = base + im->cap_boost;
            im->cap_boost += cur < 13 ? std::min(3, 13 - cur) : 3;
        }
        ctx->err = was_worst ? "lattice hash table overflowed at its worst-case capacity (internal error)"
                             : "lattice hash table overflowed: the outputs of that call are invalid; the context has raised its "
                               "capacity (x8 slots per frame), repeat the call (or set params.lattice_capacity_log2 = -1)";
        return RVSEG_ERR_CAPACITY;
    }
    return RVSEG_OK;
}

// First half of crf_frames_build: status of the previous build, buffers, and the memsets of the new one.  None of it
// needs the frames' cloud, so the frame path enqueues it on the build stream BEFORE that stream waits for prep_kernel
// (40 us of a single frame's 2 ms).  `s` must already be ordered behind the previous user of the lattice.
rvseg_status crf_frames_build_begin(rvseg_ctx* ctx, Pipeline* im, int n, hipStream_t s) {
    CrfState* cs;
    rvseg_status st = crf_state(ctx, im, &cs);
    if (st != RVSEG_OK) return st;
    const FrameGeom& g = im->geom;
    const int N = g.W * g.H;
    // status of the previous asynchronous build (an earlier chunk of this call, or an earlier call whose
    // status nobody polled): its outputs were invalid, so this call must not pass for a clean one
    if ((st = crf_frames_status(ctx, im, true)) != RVSEG_OK) return st;
    if (cs->lat.size() < 1) cs->lat.resize(1);
    LatticeBufs& lb = cs->lat[0];
    if ((st = lattice_prepare(ctx, lb, 6, N, n, false, cs->frame_vertices_seen)) != RVSEG_OK) return st;
    return lattice_clear(ctx, lb, s);
}

rvseg_status crf_frames_build(rvseg_ctx* ctx, Pipeline* im, int n, const uint8_t* d_rgb, hipStream_t s) {
    CrfState* cs;
    rvseg_status st = crf_state(ctx, im, &cs);
    if (st != RVSEG_OK) return st;
    const rvseg_params& p = ctx->params;
    if (cs->lat.size() < 1 || !cs->lat[0].cleared) {
        if ((st = crf_frames_build_begin(ctx, im, n, s)) != RVSEG_OK) return st;
    }
    LatticeBufs& lb = cs->lat[0];
    FeatureSource fs{};
    fs.mode = 1; fs.cloud = im->cloud.as<float4>(); fs.rgb = d_rgb;
    fs.xyz_kernel = p.dcrf_xyz_kernel; fs.rgb_kernel = p.dcrf_rgb_kernel;
    if ((st = lattice_build(ctx, cs, lb, fs, s)) != RVSEG_OK) return st;
    if ((st = counters_readback(ctx, cs, lb, 0, s)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipEventRecord(cs->counters_ev, s));
    cs->counters_pending = true;
    cs->pending_frames = n;
    cs->info_async = true;
    return RVSEG_OK;
}

rvseg_status crf_frames_infer(rvseg_ctx* ctx, Pipeline* im, int n, const float* d_post, float* d_marg, int8_t* d_labels,
                              hipStream_t s) {
    CrfState* cs;
    rvseg_status st0 = crf_state(ctx, im, &cs);
    if (st0 != RVSEG_OK) return st0;
    const FrameGeom& g = im->geom;
    const rvseg_params& p = ctx->params;
    const DeviceForest& f = ctx->forest;
    const int N = g.W * g.H;
    rvseg_status st;
    const size_t frame_stride = (size_t)N * f.sum_classes;
    float* marg = d_marg;
    if (!marg) {
        if ((st = dev_reserve(ctx, cs->q, frame_stride * 4 * n)) != RVSEG_OK) return st;
        marg = cs->q.as<float>();
    }
    int prefix = 0;
    const float w = p.dcrf_kernel_weight;
    bool all_labelled = true;   // the last fused update of every layer wrote its labels
    hipStream_t s2;
    if ((st = csr_nrm_before_fork(ctx, cs, f.n_layers, f.class_counts, p.dcrf_iterations, s)) != RVSEG_OK) return st;
    if ((st = layer_stream_fork(ctx, cs, s, f.n_layers, &s2)) != RVSEG_OK) return st;
    // With two streams the layers of the SECOND stream are enqueued first: enqueuing a layer's whole loop takes the host
    // a few hundred microseconds, during which the other stream has nothing to run, and the reference's second layer
    // is the one with more classes (8 and 9: the longer loop starts first)
    for (int li = 0; li < f.n_layers; li++) {
        const int l = layer_enqueue_order(li, f.n_layers, s2 != s);
        prefix = 0;
        for (int k = 0; k < l; k++) prefix += f.class_counts[k];
        const int C = f.class_counts[l];
        ValueView U{const_cast<float*>(d_post), frame_stride, (size_t)N * prefix};
        ValueView Q{marg, frame_stride, (size_t)N * prefix};
        // unary energy = -(log-posterior) (segmenter.cpp:642), so -U is the posterior itself
        MfLabels lab{d_labels, p.label_mode, p.unknown_label[l], f.n_layers, l};
        bool done = false;
        const int slot = l & 1;
        if ((st = mean_field(ctx, cs, 1, &w, U, false, C, N, (long long)N * n, p.dcrf_iterations, Q, slot ? s2 : s, d_labels ? &lab : nullptr,
                             &done, slot, slot == 0 || s2 == s)) != RVSEG_OK) { (void)layer_stream_join(ctx, cs, s, s2); return st; }
        all_labelled = all_labelled && done;
    }
    if ((st = layer_stream_join(ctx, cs, s, s2)) != RVSEG_OK) return st;
    if (d_labels && !all_labelled) {
        timer_mark(ctx, "labels", s);
        launch_labels_frames(marg, n, N, f, p.label_mode, p.unknown_label, d_labels, s);
    }
    return RVSEG_OK;
}

// Permutohedral::init on device-resident features with the overflow retry of the host entry points:
// the counters are read back (one stream synchronisation) before the mean field is enqueued
static rvseg_status lattice_build_retry(rvseg_ctx* ctx, CrfState* cs, LatticeBufs& lb, int d, int N, const float* d_feat, hipStream_t s) {
    rvseg_status st;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((st = lattice_prepare(ctx, lb, d, N, 1, attempt == 1)) != RVSEG_OK) return st;
        FeatureSource fs{};
        fs.mode = 0; fs.feat = d_feat;
        if ((st = lattice_build(ctx, cs, lb, fs, s)) != RVSEG_OK) return st;
        int cnt[3];
        if ((st = lattice_counters(ctx, cs, lb, s, cnt)) != RVSEG_OK) return st;
        if (!cnt[1]) return RVSEG_OK;
    }
    ctx->err = "lattice hash table overflow";
    return RVSEG_ERR_CAPACITY;
}

static rvseg_status crf_bare_state(rvseg_ctx* ctx, Pipeline** im_out, CrfState** cs_out) {
    if (!ctx->impl) {
        Pipeline* im = new Pipeline();
        ctx->impl = reinterpret_cast<rvseg_ctx::Impl*>(im);
        im->bare = true;
    }
    *im_out = reinterpret_cast<Pipeline*>(ctx->impl);
    return crf_state(ctx, *im_out, cs_out);
}

rvseg_status crf_cloud_layers(rvseg_ctx* ctx, int N, int n_layers, const int* class_counts, const float* d_unaries,
                              const float* d_features, float potts_w, int iterations, int label_mode, const int* unknown,
                              int8_t* d_labels, hipStream_t s) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_bare_state(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (cs->lat.size() < 1) cs->lat.resize(1);
    timer_mark(ctx, "lattice_build", s);
    if ((st = lattice_build_retry(ctx, cs, cs->lat[0], 6, N, d_features, s)) != RVSEG_OK) return st;
    int cmax = 0;
    for (int l = 0; l < n_layers; l++) cmax = std::max(cmax, class_counts[l]);
    // marginals of even / odd layers in two halves of cs->q (the odd layers run on the second stream)
    if ((st = dev_reserve(ctx, cs->q, (size_t)N * cmax * 4 * 2)) != RVSEG_OK) return st;
    hipStream_t s2;
    if ((st = csr_nrm_before_fork(ctx, cs, n_layers, class_counts, iterations, s)) != RVSEG_OK) return st;
    if ((st = layer_stream_fork(ctx, cs, s, n_layers, &s2)) != RVSEG_OK) return st;
    for (int li = 0; li < n_layers; li++) {
        const int l = layer_enqueue_order(li, n_layers, s2 != s);
        size_t prefix = 0;
        for (int k = 0; k < l; k++) prefix += (size_t)class_counts[k];
        const int C = class_counts[l];
        const int slot = l & 1;
        hipStream_t sl = slot ? s2 : s;
        float* q = cs->q.as<float>() + (size_t)slot * N * cmax;
        ValueView U{const_cast<float*>(d_unaries) + (size_t)N * prefix, (size_t)N * C, 0};
        ValueView Q{q, (size_t)N * C, 0};
        MfLabels lab{d_labels ? d_labels + (size_t)l * N : nullptr, label_mode, unknown[l], 1, 0};
        bool done = false;
        // crf.setUnaryEnergy(-unaries[l]) (segmenter.cpp:642): the accumulated posteriors ARE -energy
        if ((st = mean_field(ctx, cs, 1, &potts_w, U, false, C, N, N, iterations, Q, sl, d_labels ? &lab : nullptr, &done, slot,
                             slot == 0 || s2 == s)) != RVSEG_OK) { (void)layer_stream_join(ctx, cs, s, s2); return st; }
        if (d_labels && !done) {
            if (slot == 0 || s2 == s) timer_mark(ctx, "labels", sl);
            launch_labels(q, (size_t)N, C, label_mode, unknown[l], d_labels + (size_t)l * N, sl);
        }
    }
    if ((st = layer_stream_join(ctx, cs, s, s2)) != RVSEG_OK) return st;
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

}  // namespace rvseg

using namespace rvseg;

static rvseg_status crf_enter(rvseg_ctx* ctx, Pipeline** im_out, CrfState** cs_out) {
    if (!ctx) return RVSEG_ERR_INVALID_ARG;
    RV_HIP(ctx, hipSetDevice(ctx->params.device));
    if (!ctx->impl) {
        // the CRF entry points do not need the frame tables; create a bare pipeline object
        Pipeline* im = new Pipeline();
        ctx->impl = reinterpret_cast<rvseg_ctx::Impl*>(im);
        im->bare = true;
    }
    *im_out = reinterpret_cast<Pipeline*>(ctx->impl);
    return crf_state(ctx, *im_out, cs_out);
}

extern "C" {

rvseg_status rvseg_crf_infer_multi(rvseg_ctx* ctx, int32_t N, int32_t C, int32_t n_kernels, const int32_t* ds,
                                   const float* const* features, const float* ws, const float* unary_energy,
                                   int32_t iterations, float* Q_out, int8_t* map_out, int32_t label_mode, int32_t unknown_label) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_enter(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (N <= 0 || C <= 0 || C > 64 || n_kernels < 0 || n_kernels > 8 || iterations < 0 || !unary_energy || !Q_out ||
        (n_kernels > 0 && (!ds || !features || !ws)) || label_mode < 0 || label_mode > 3) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    hipStream_t s = ctx->stream;
    if ((int)cs->lat.size() < n_kernels) cs->lat.resize(n_kernels);
    for (int attempt = 0; attempt < 2; attempt++) {
        bool overflow = false;
        for (int k = 0; k < n_kernels && !overflow; k++) {
            LatticeBufs& lb = cs->lat[k];
            if ((st = lattice_prepare(ctx, lb, ds[k], N, 1, attempt == 1)) != RVSEG_OK) return st;
            if ((st = dev_reserve(ctx, cs->feat, (size_t)N * ds[k] * 4)) != RVSEG_OK) return st;
            RV_HIP(ctx, hipMemcpyAsync(cs->feat.p, features[k], (size_t)N * ds[k] * 4, hipMemcpyHostToDevice, s));
            FeatureSource fs{};
            fs.mode = 0; fs.feat = cs->feat.as<float>();
            if ((st = lattice_build(ctx, cs, lb, fs, s)) != RVSEG_OK) return st;
            int cnt[3];
            if ((st = lattice_counters(ctx, cs, lb, s, cnt)) != RVSEG_OK) return st;
            overflow = cnt[1] != 0;
        }
        if (!overflow) break;
        if (attempt == 1) { ctx->err = "lattice hash table overflow"; return RVSEG_ERR_CAPACITY; }
    }
    const size_t tot = (size_t)N * C;
    if ((st = dev_reserve(ctx, cs->unary, tot * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, cs->q, tot * 4)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(cs->unary.p, unary_energy, tot * 4, hipMemcpyHostToDevice, s));
    ValueView U{cs->unary.as<float>(), tot, 0}, Q{cs->q.as<float>(), tot, 0};
    timer_reset(ctx);
    if ((st = mean_field(ctx, cs, n_kernels, ws, U, true, C, N, N, iterations, Q, s)) != RVSEG_OK) return st;
    timer_mark(ctx, "end", s);
    RV_HIP(ctx, hipMemcpyAsync(Q_out, cs->q.p, tot * 4, hipMemcpyDeviceToHost, s));
    if (map_out) {
        if ((st = dev_reserve(ctx, cs->labels, (size_t)N)) != RVSEG_OK) return st;
        launch_labels(cs->q.as<float>(), (size_t)N, C, label_mode, unknown_label, cs->labels.as<int8_t>(), s);
        RV_HIP(ctx, hipMemcpyAsync(map_out, cs->labels.p, (size_t)N, hipMemcpyDeviceToHost, s));
    }
    RV_HIP(ctx, hipStreamSynchronize(s));
    return RVSEG_OK;
}

rvseg_status rvseg_crf_infer_device(rvseg_ctx* ctx, int32_t N, int32_t C, int32_t d, const float* d_unary, int32_t unary_is_energy,
                                    const float* d_features, float potts_w, int32_t iterations, float* d_Q_out, int8_t* d_map_out,
                                    int32_t label_mode, int32_t unknown_label, void* hip_stream) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_enter(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (N <= 0 || C <= 0 || C > 64 || d < 1 || d > 7 || iterations < 0 || !d_unary || !d_features || (!d_Q_out && !d_map_out) ||
        label_mode < 0 || label_mode > 3) {
        ctx->err = "bad arguments";
        return RVSEG_ERR_INVALID_ARG;
    }
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
    if (cs->lat.size() < 1) cs->lat.resize(1);
    timer_reset(ctx);
    timer_mark(ctx, "lattice_build", s);
    if ((st = lattice_build_retry(ctx, cs, cs->lat[0], d, N, d_features, s)) != RVSEG_OK) return st;
    float* q = d_Q_out;
    if (!q) {
        if ((st = dev_reserve(ctx, cs->q, (size_t)N * C * 4)) != RVSEG_OK) return st;
        q = cs->q.as<float>();
    }
    ValueView U{const_cast<float*>(d_unary), (size_t)N * C, 0}, Q{q, (size_t)N * C, 0};
    MfLabels lab{d_map_out, label_mode, unknown_label, 1, 0};
    bool done = false;
    if ((st = mean_field(ctx, cs, 1, &potts_w, U, unary_is_energy != 0, C, N, N, iterations, Q, s, d_map_out ? &lab : nullptr, &done)) != RVSEG_OK) return st;
    if (d_map_out && !done) {
        timer_mark(ctx, "labels", s);
        launch_labels(q, (size_t)N, C, label_mode, unknown_label, d_map_out, s);
    }
    timer_mark(ctx, "end", s);
    RV_LAUNCH_OK(ctx);
    return RVSEG_OK;
}

rvseg_status rvseg_crf_infer(rvseg_ctx* ctx, int32_t N, int32_t C, int32_t d, const float* unary_energy,
                             const float* features, float potts_w, int32_t iterations, float* Q_out, int8_t* map_out,
                             int32_t label_mode, int32_t unknown_label) {
    const float* feats[1] = {features};
    return rvseg_crf_infer_multi(ctx, N, C, 1, &d, feats, &potts_w, unary_energy, iterations, Q_out, map_out, label_mode, unknown_label);
}

rvseg_status rvseg_lattice_build(rvseg_ctx* ctx, const float* features, int32_t N, int32_t d, int32_t* offsets_out,
                                 float* bary_out, int16_t* keys_out, int32_t keys_capacity, int32_t* M_out) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_enter(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (!features || N <= 0 || !M_out) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    hipStream_t s = ctx->stream;
    if (cs->lat.size() < 1) cs->lat.resize(1);
    LatticeBufs& lb = cs->lat[0];
    int cnt[3] = {0, 0, 0};
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((st = lattice_prepare(ctx, lb, d, N, 1, attempt == 1)) != RVSEG_OK) return st;
        if ((st = dev_reserve(ctx, cs->feat, (size_t)N * d * 4)) != RVSEG_OK) return st;
        RV_HIP(ctx, hipMemcpyAsync(cs->feat.p, features, (size_t)N * d * 4, hipMemcpyHostToDevice, s));
        FeatureSource fs{};
        fs.mode = 0; fs.feat = cs->feat.as<float>();
        if ((st = lattice_build(ctx, cs, lb, fs, s)) != RVSEG_OK) return st;
        if ((st = lattice_counters(ctx, cs, lb, s, cnt)) != RVSEG_OK) return st;
        if (!cnt[1]) break;
        if (attempt == 1) { ctx->err = "lattice hash table overflow"; return RVSEG_ERR_CAPACITY; }
    }
    const int M = cnt[0];
    *M_out = M;
    const size_t E = (size_t)N * (d + 1);
    if (offsets_out) RV_HIP(ctx, hipMemcpyAsync(offsets_out, lb.dev.offsets, E * 4, hipMemcpyDeviceToHost, s));
    if (bary_out) RV_HIP(ctx, hipMemcpyAsync(bary_out, lb.dev.bary, E * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    if (keys_out) {
        if (keys_capacity < M) { ctx->err = "keys_out too small"; return RVSEG_ERR_INVALID_ARG; }
        std::vector<int16_t> k8((size_t)M * 8);
        RV_HIP(ctx, hipMemcpy(k8.data(), lb.dev.vkeys, (size_t)M * 16, hipMemcpyDeviceToHost));
        for (int i = 0; i < M; i++)
            for (int k = 0; k < d; k++) keys_out[(size_t)i * d + k] = k8[(size_t)i * 8 + k];
    }
    return RVSEG_OK;
}

rvseg_status rvseg_lattice_neighbours(rvseg_ctx* ctx, int32_t* n1_out, int32_t* n2_out, uint32_t* csr_point,
                                      uint32_t* vstart, uint32_t* vend) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_enter(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (cs->lat.empty() || !cs->lat[0].built || cs->lat[0].dev.n_frames != 1) { ctx->err = "no lattice built on this context"; return RVSEG_ERR_INVALID_ARG; }
    LatticeBufs& lb = cs->lat[0];
    int cnt[3];
    if ((st = lattice_counters(ctx, cs, lb, ctx->stream, cnt)) != RVSEG_OK) return st;
    const int M = cnt[0], d = lb.dev.d;
    for (int j = 0; j <= d; j++) {
        if (n1_out) RV_HIP(ctx, hipMemcpy(n1_out + (size_t)j * M, lb.dev.nb1 + (size_t)j * lb.dev.m_bound, (size_t)M * 4, hipMemcpyDeviceToHost));
        if (n2_out) RV_HIP(ctx, hipMemcpy(n2_out + (size_t)j * M, lb.dev.nb2 + (size_t)j * lb.dev.m_bound, (size_t)M * 4, hipMemcpyDeviceToHost));
    }
    if (csr_point) {
        std::vector<uint2> pw((size_t)lb.n_entries);
        RV_HIP(ctx, hipMemcpy(pw.data(), lb.dev.csr_pw, (size_t)lb.n_entries * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < pw.size(); i++) csr_point[i] = pw[i].x;
    }
    if (vstart) RV_HIP(ctx, hipMemcpy(vstart, lb.dev.vstart, (size_t)M * 4, hipMemcpyDeviceToHost));
    if (vend) RV_HIP(ctx, hipMemcpy(vend, lb.dev.vend, (size_t)M * 4, hipMemcpyDeviceToHost));
    return RVSEG_OK;
}

// debug (not in rvseg.h): per-block trace of the last resident splat of the frame path's lattice, and the tile ranges
extern "C" rvseg_status rvseg_debug_resident(rvseg_ctx* ctx, void* trace_out, size_t trace_cap, unsigned* tile0_out, size_t tile0_cap, int meta[6]) {
    if (!ctx || !ctx->impl) return RVSEG_ERR_INVALID_ARG;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    if (!im->crf || im->crf->lat.empty() || !im->crf->lat[0].resident_on) return RVSEG_ERR_INVALID_ARG;
    LatticeBufs& b = im->crf->lat[0];
    RV_HIP(ctx, hipDeviceSynchronize());
    const int nf = b.dev.n_frames;
    meta[0] = b.resident.B; meta[1] = b.resident.band_wb; meta[2] = b.resident.n_bands; meta[3] = nf; meta[4] = RES_MAXB;
    int fl[2] = {0, 0};
    RV_HIP(ctx, hipMemcpy(fl, b.resident.flags, 8, hipMemcpyDeviceToHost));
    meta[5] = fl[1];
    if (trace_out && b.resident.trace && trace_cap >= (size_t)nf * RES_MAXB * 64)
        RV_HIP(ctx, hipMemcpy(trace_out, b.resident.trace, (size_t)nf * RES_MAXB * 64, hipMemcpyDeviceToHost));
    if (tile0_out && tile0_cap >= (size_t)nf * (RES_MAXB + 1) * 4)
        RV_HIP(ctx, hipMemcpy(tile0_out, b.resident.blk_tile0, (size_t)nf * (RES_MAXB + 1) * 4, hipMemcpyDeviceToHost));
    return RVSEG_OK;
}

rvseg_status rvseg_last_schedule(rvseg_ctx* ctx, rvseg_schedule_info* out) {
    if (!ctx || !out) return RVSEG_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    out->planner_fallback = -1;
    out->vertices = -1;
    out->longest_list = -1;
    if (!ctx->impl) return RVSEG_OK;
    Pipeline* im = reinterpret_cast<Pipeline*>(ctx->impl);
    if (im->crf) *out = im->crf->info;
    return RVSEG_OK;
}

rvseg_status rvseg_lattice_filter(rvseg_ctx* ctx, const float* in, int32_t C, float* out) {
    Pipeline* im; CrfState* cs;
    rvseg_status st = crf_enter(ctx, &im, &cs);
    if (st != RVSEG_OK) return st;
    if (cs->lat.empty() || !cs->lat[0].built || cs->lat[0].dev.n_frames != 1) { ctx->err = "no lattice built on this context"; return RVSEG_ERR_INVALID_ARG; }
    if (!in || !out || C <= 0 || C > 64) { ctx->err = "bad arguments"; return RVSEG_ERR_INVALID_ARG; }
    LatticeBufs& lb = cs->lat[0];
    const int N = lb.dev.N;
    hipStream_t s = ctx->stream;
    const size_t tot = (size_t)N * C;
    if ((st = dev_reserve(ctx, cs->q, tot * 4)) != RVSEG_OK) return st;
    if ((st = dev_reserve(ctx, cs->tmp, tot * 4)) != RVSEG_OK) return st;
    if ((st = values_reserve(ctx, cs, lb.dev.m_bound, C)) != RVSEG_OK) return st;
    RV_HIP(ctx, hipMemcpyAsync(cs->q.p, in, tot * 4, hipMemcpyHostToDevice, s));
    const bool seq = C <= 2;
    ValueView V{cs->q.as<float>(), tot, 0};
    launch_splat(lb.dev, V, C, 0, cs->val_a.as<float>(), s);
    float* blurred = launch_blur(lb.dev, C, seq, false, cs->val_a.as<float>(), cs->val_b.as<float>(), s);
    launch_slice(lb.dev, C, seq, 0, blurred, 0.f, cs->tmp.as<float>(), N, s);
    RV_LAUNCH_OK(ctx);
    RV_HIP(ctx, hipMemcpyAsync(out, cs->tmp.p, tot * 4, hipMemcpyDeviceToHost, s));
    RV_HIP(ctx, hipStreamSynchronize(s));
    return RVSEG_OK;
}

}  // extern "C"
