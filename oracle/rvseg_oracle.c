/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the rvseg hot path (see rvseg_oracle.h for the
 * pinning status of each part).  Plain C, single-threaded, written from the cited reference
 * lines (paths relative to /root/reference).  Compile with -ffp-contract=off: the reference
 * build has no FMA (CMakeLists.txt:4-13).
 */
#define _GNU_SOURCE
#include "rvseg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* =========================================================================================
 * Forest model IO + evaluation
 * ========================================================================================= */

typedef struct {
    const unsigned char *p;
    size_t left;
    int bad;
} rd_t;

static int rd_i32(rd_t *r) {
    int v = 0;
    if (r->left < 4) { r->bad = 1; return 0; }
    memcpy(&v, r->p, 4); /* io.h:43-47: raw native-endian bytes */
    r->p += 4; r->left -= 4;
    return v;
}
static float rd_f32(rd_t *r) {
    float v = 0;
    if (r->left < 4) { r->bad = 1; return 0; }
    memcpy(&v, r->p, 4);
    r->p += 4; r->left -= 4;
    return v;
}
/* io.h:98-108: int32 N followed by N elements */
static int *rd_vec_i32(rd_t *r, int *n_out) {
    int n = rd_i32(r);
    if (r->bad || n < 0 || (size_t)n * 4 > r->left) { r->bad = 1; *n_out = 0; return NULL; }
    int *v = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) v[i] = rd_i32(r);
    *n_out = n;
    return v;
}
static float *rd_vec_f32(rd_t *r, int *n_out) {
    int n = rd_i32(r);
    if (r->bad || n < 0 || (size_t)n * 4 > r->left) { r->bad = 1; *n_out = 0; return NULL; }
    float *v = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) v[i] = rd_f32(r);
    *n_out = n;
    return v;
}

static void tree_free(orc_tree *t) {
    free(t->split_feature); free(t->threshold); free(t->left_child);
    free(t->hist_off); free(t->hist); free(t->mh_node_off); free(t->mh_off); free(t->mh);
    memset(t, 0, sizeof(*t));
}

/* DecisionTree::read, classifier.cpp:134-142 */
static int tree_read(rd_t *r, orc_tree *t) {
    int n1, n2, n3;
    memset(t, 0, sizeof(*t));
    t->split_feature = rd_vec_i32(r, &n1);
    t->threshold = rd_vec_f32(r, &n2);
    t->left_child = rd_vec_i32(r, &n3);
    if (r->bad || n1 != n2 || n1 != n3 || n1 < 1) { r->bad = 1; return -1; }
    t->n_nodes = n1;
    /* histograms: vector<vector<float>> */
    int nh = rd_i32(r);
    if (r->bad || nh != n1) { r->bad = 1; return -1; }
    t->hist_off = (int *)calloc((size_t)nh + 1, sizeof(int));
    size_t cap = 16, used = 0;
    t->hist = (float *)malloc(cap * sizeof(float));
    for (int i = 0; i < nh; i++) {
        int c = rd_i32(r);
        if (r->bad || c < 0 || (size_t)c * 4 > r->left) { r->bad = 1; return -1; }
        while (used + (size_t)c > cap) { cap *= 2; t->hist = (float *)realloc(t->hist, cap * sizeof(float)); }
        for (int k = 0; k < c; k++) t->hist[used++] = rd_f32(r);
        t->hist_off[i + 1] = (int)used;
    }
    /* multi_histograms: vector<vector<vector<float>>> */
    int nm = rd_i32(r);
    if (r->bad || nm != n1) { r->bad = 1; return -1; }
    t->mh_node_off = (int *)calloc((size_t)nm + 1, sizeof(int));
    size_t lcap = 16, lused = 0;
    t->mh_off = (int *)malloc((lcap + 1) * sizeof(int));
    t->mh_off[0] = 0;
    size_t mcap = 16, mused = 0;
    t->mh = (float *)malloc(mcap * sizeof(float));
    for (int i = 0; i < nm; i++) {
        int L = rd_i32(r);
        if (r->bad || L < 0 || (size_t)L * 4 > r->left) { r->bad = 1; return -1; }
        for (int l = 0; l < L; l++) {
            int c = rd_i32(r);
            if (r->bad || c < 0 || (size_t)c * 4 > r->left) { r->bad = 1; return -1; }
            while (mused + (size_t)c > mcap) { mcap *= 2; t->mh = (float *)realloc(t->mh, mcap * sizeof(float)); }
            for (int k = 0; k < c; k++) t->mh[mused++] = rd_f32(r);
            if (lused + 1 > lcap) { lcap *= 2; t->mh_off = (int *)realloc(t->mh_off, (lcap + 1) * sizeof(int)); }
            t->mh_off[++lused] = (int)mused;
        }
        t->mh_node_off[i + 1] = (int)lused;
    }
    return r->bad ? -1 : 0;
}

/* RandomForest::read, classifier.cpp:222-235 */
orc_forest *orc_forest_load_mem(const void *buf, size_t size) {
    rd_t r = { (const unsigned char *)buf, size, 0 };
    int T = rd_i32(&r);
    if (r.bad || T < 0 || T > 1 << 20) return NULL;
    orc_forest *f = (orc_forest *)calloc(1, sizeof(orc_forest));
    f->n_trees = T;
    f->trees = (orc_tree *)calloc((size_t)(T > 0 ? T : 1), sizeof(orc_tree));
    for (int t = 0; t < T; t++) {
        if (tree_read(&r, &f->trees[t]) != 0) { orc_forest_free(f); return NULL; }
    }
    return f;
}

orc_forest *orc_forest_load(const char *path) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    void *buf = malloc((size_t)(n > 0 ? n : 1));
    size_t got = fread(buf, 1, (size_t)n, fp);
    fclose(fp);
    orc_forest *f = got == (size_t)n ? orc_forest_load_mem(buf, (size_t)n) : NULL;
    free(buf);
    return f;
}

void orc_forest_free(orc_forest *f) {
    if (!f) return;
    for (int t = 0; t < f->n_trees; t++) tree_free(&f->trees[t]);
    free(f->trees);
    free(f);
}

static int first_leaf(const orc_tree *t) {
    for (int n = 0; n < t->n_nodes; n++)
        if (t->left_child[n] == 0) return n;
    return -1;
}

int orc_forest_single_classes(const orc_forest *f) {
    if (!f || f->n_trees < 1) return 0;
    int n = first_leaf(&f->trees[0]);
    if (n < 0) return 0;
    return f->trees[0].hist_off[n + 1] - f->trees[0].hist_off[n];
}

int orc_forest_layers(const orc_forest *f, int *class_counts, int max_layers) {
    if (!f || f->n_trees < 1) return 0;
    const orc_tree *t = &f->trees[0];
    int n = first_leaf(t);
    if (n < 0) return 0;
    int L = t->mh_node_off[n + 1] - t->mh_node_off[n];
    for (int l = 0; l < L && l < max_layers; l++) {
        int o = t->mh_node_off[n] + l;
        class_counts[l] = t->mh_off[o + 1] - t->mh_off[o];
    }
    return L;
}

/* classifier.cpp:97-117: strict '<' on float, right child = left + 1 */
int orc_tree_find_leaf(const orc_tree *t, const float *x) {
    int node = 0;
    while (t->left_child[node] != 0) {
        if (x[t->split_feature[node]] < t->threshold[node]) node = t->left_child[node];
        else node = t->left_child[node] + 1;
    }
    return node;
}

/* classifier.cpp:166-184: copy tree 0's histogram, then += trees 1..T-1 in order */
void orc_forest_class_log_posterior(const orc_forest *f, const float *x, float *out) {
    const orc_tree *t0 = &f->trees[0];
    int leaf = orc_tree_find_leaf(t0, x);
    int C = t0->hist_off[leaf + 1] - t0->hist_off[leaf];
    for (int c = 0; c < C; c++) out[c] = t0->hist[t0->hist_off[leaf] + c];
    for (int i = 1; i < f->n_trees; i++) {
        const orc_tree *t = &f->trees[i];
        int lf = orc_tree_find_leaf(t, x);
        int Ci = t->hist_off[lf + 1] - t->hist_off[lf];
        for (int c = 0; c < Ci; c++) out[c] += t->hist[t->hist_off[lf] + c];
    }
}

/* classifier.cpp:187-208 */
void orc_forest_multi_class_log_posterior(const orc_forest *f, const float *x, float *out) {
    const orc_tree *t0 = &f->trees[0];
    int leaf = orc_tree_find_leaf(t0, x);
    int b0 = t0->mh_off[t0->mh_node_off[leaf]];
    int e0 = t0->mh_off[t0->mh_node_off[leaf + 1]];
    for (int k = 0; k < e0 - b0; k++) out[k] = t0->mh[b0 + k];
    for (int i = 1; i < f->n_trees; i++) {
        const orc_tree *t = &f->trees[i];
        int lf = orc_tree_find_leaf(t, x);
        int b = t->mh_off[t->mh_node_off[lf]];
        int e = t->mh_off[t->mh_node_off[lf + 1]];
        for (int k = 0; k < e - b; k++) out[k] += t->mh[b + k];
    }
}

static int forest_sum_classes(const orc_forest *f, int multi) {
    if (!multi) return orc_forest_single_classes(f);
    int cc[64];
    int L = orc_forest_layers(f, cc, 64), s = 0;
    for (int l = 0; l < L; l++) s += cc[l];
    return s;
}

void orc_forest_eval(const orc_forest *f, const float *X, int P, int D, int multi, float *out) {
    int S = forest_sum_classes(f, multi);
    for (int i = 0; i < P; i++) {
        if (multi) orc_forest_multi_class_log_posterior(f, X + (size_t)i * D, out + (size_t)i * S);
        else orc_forest_class_log_posterior(f, X + (size_t)i * D, out + (size_t)i * S);
    }
}

/* =========================================================================================
 * Parameters
 * ========================================================================================= */
void orc_params_default(orc_params *p) {
    /* resources/config.json:32-44,81-90 */
    p->width = 640; p->height = 480; p->stride = 2;
    p->depth_min = 0.5f; p->depth_max = 15.0f;
    p->patch_size = 77; p->patch_size_reduce = 11;
    p->feature_color_patch = p->feature_depth = p->feature_height = p->feature_normal = 1;
    p->fill_value = 0.0f;
    p->dcrf_xyz_kernel = 0.5f; p->dcrf_rgb_kernel = 4.0f; p->dcrf_kernel_weight = 10.0f;
    p->dcrf_iterations = 10;
}

int orc_feature_length(const orc_params *p) {
    int n = 0; /* feature_extractor.h:46-51 */
    if (p->feature_color_patch) n += p->patch_size_reduce * p->patch_size_reduce * 3;
    if (p->feature_depth) n += 1;
    if (p->feature_height) n += 1;
    if (p->feature_normal) n += 1;
    return n;
}

/* =========================================================================================
 * Deterministic elementary functions.  Both are IEEE double sequences of +,-,*,/,sqrt,rint so
 * that the HIP path can evaluate the very same sequence and agree bit for bit.
 * ========================================================================================= */

/* fdlibm __ieee754_acos for x in [0,1], result rounded to float.  The reference calls libm
 * acos on fabs(normal_z) (feature_extractor.h:283); libm is correctly rounded to < 1 ulp double,
 * so after narrowing to float the two agree except in ~1e-9 of the cases. */
float orc_acos_f32(float xf) {
    static const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17,
        pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
        pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
        pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
        qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
        qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
    double x = (double)xf;
    if (x != x) return xf;
    if (x >= 1.0) return 0.0f;
    if (x < 0.0) x = 0.0;
    if (x < 0.5) {
        double z = x * x;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        return (float)(pio2_hi - (x - (pio2_lo - x * r)));
    } else {
        double z = (1.0 - x) * 0.5;
        double s = sqrt(z);
        uint64_t bits;
        memcpy(&bits, &s, 8);
        bits &= 0xFFFFFFFF00000000ull;
        double df;
        memcpy(&df, &bits, 8);
        double c = (z - df * df) / (s + df);
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        double w = r * s + c;
        return (float)(2.0 * (df + w));
    }
}

/* exp for expAndNormalize (densecrf.cpp:102: `b.array().exp()`; Eigen is not in the tree, version unpinned: PARITY
 * UNPINNED, build-owned definition).  Restates, as recollected, what Eigen 3's float packet path evaluates on an SSE
 * build without FMA (pexp<Packet4f>, after Cephes expf): clamp to +-88.3762626647949, fx = floor(x*log2(e) + 0.5),
 * two-step reduction by ln 2 = 0.693359375 - 2.12194440e-4, degree-5 polynomial (separately rounded multiply and add),
 * y * x^2 + x + 1, exact scaling by 2^fx.  All in float; this file is compiled with -ffp-contract=off.  (Eigen sends the
 * tail of a vector whose length is not a multiple of 4 through std::exp; here every class takes the packet formula.) */
float orc_exp_f32(float xf) {
    if (xf != xf) return xf;
    float x = xf > 88.3762626647949f ? 88.3762626647949f : xf;
    x = x < -88.3762626647949f ? -88.3762626647949f : x;
    float fx = x * 1.44269504088896341f;
    fx = fx + 0.5f;
    const float t = (float)(int)fx;
    fx = t > fx ? t - 1.0f : t;
    const float hi = fx * 0.693359375f;
    const float lo = fx * -2.12194440e-4f;
    x = x - hi;
    x = x - lo;
    const float z = x * x;
    float y = 1.9875691500E-4f;
    y = y * x; y = y + 1.3981999507E-3f;
    y = y * x; y = y + 8.3334519073E-3f;
    y = y * x; y = y + 4.1665795894E-2f;
    y = y * x; y = y + 1.6666665459E-1f;
    y = y * x; y = y + 5.0000001201E-1f;
    y = y * z; y = y + x;
    y = y + 1.0f;
    union { int32_t i; float f; } sc;
    sc.i = ((int32_t)fx + 127) << 23;
    return y * sc.f;
}

/* =========================================================================================
 * cvtColor(CV_BGR2Lab) on 8-bit data (feature_extractor.h:129).  OpenCV is not vendored and not
 * installed: PARITY UNPINNED.  Restated after OpenCV 2.4 imgproc/color.cpp RGB2Lab_b (integer
 * pipeline: sRGB gamma LUT to 11 bit, 12-bit fixed-point XYZ with the D65 white point folded in,
 * 16-bit cube-root LUT, L*255/100, a+128, b+128).  "B" is whatever sits in channel 0.
 * ========================================================================================= */
static unsigned short g_gamma_tab[256];
static unsigned short g_cbrt_tab[3072];
static int g_lab_coeffs[9];
static int g_lab_init = 0;

static int sat_u16(long v) { return v < 0 ? 0 : (v > 65535 ? 65535 : (int)v); }

static void lab_init(void) {
    if (g_lab_init) return;
    for (int i = 0; i < 256; i++) {
        float x = i * (1.f / 255.f);
        float g = x <= 0.04045f ? x * (1.f / 12.92f) : (float)pow((double)(x + 0.055) * (1. / 1.055), 2.4);
        g_gamma_tab[i] = (unsigned short)sat_u16(lrintf(255.f * 8 * g));
    }
    for (int i = 0; i < 3072; i++) {
        float x = i * (1.f / (255.f * 8));
        float v = x < 0.008856f ? x * 7.787f + 0.13793103448275862f : cbrtf(x);
        g_cbrt_tab[i] = (unsigned short)sat_u16(lrintf(32768.f * v));
    }
    static const float xyz[9] = { 0.412453f, 0.357580f, 0.180423f, 0.212671f, 0.715160f,
                                  0.072169f, 0.019334f, 0.119193f, 0.950227f };
    static const float white[3] = { 0.950456f, 1.f, 1.088754f };
    float scale[3] = { 4096.f / white[0], 4096.f, 4096.f / white[2] };
    for (int i = 0; i < 3; i++) { /* blueIdx = 0: R coefficient applies to channel 2 */
        g_lab_coeffs[i * 3 + 2] = (int)lrint((double)(xyz[i * 3] * scale[i]));
        g_lab_coeffs[i * 3 + 1] = (int)lrint((double)(xyz[i * 3 + 1] * scale[i]));
        g_lab_coeffs[i * 3 + 0] = (int)lrint((double)(xyz[i * 3 + 2] * scale[i]));
    }
    g_lab_init = 1;
}

#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void orc_bgr2lab_u8(const uint8_t *src, uint8_t *dst, int n_pixels) {
    lab_init();
    const int Lscale = (116 * 255 + 50) / 100;
    const int Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    const int *C = g_lab_coeffs;
    for (int i = 0; i < n_pixels; i++, src += 3, dst += 3) {
        int c0 = g_gamma_tab[src[0]], c1 = g_gamma_tab[src[1]], c2 = g_gamma_tab[src[2]];
        int fX = g_cbrt_tab[DESCALE(c0 * C[0] + c1 * C[1] + c2 * C[2], 12)];
        int fY = g_cbrt_tab[DESCALE(c0 * C[3] + c1 * C[4] + c2 * C[5], 12)];
        int fZ = g_cbrt_tab[DESCALE(c0 * C[6] + c1 * C[7] + c2 * C[8], 12)];
        int L = DESCALE(Lscale * fY + Lshift, 15);
        int a = DESCALE(500 * (fX - fY) + 128 * (1 << 15), 15);
        int b = DESCALE(200 * (fY - fZ) + 128 * (1 << 15), 15);
        dst[0] = sat_u8(L); dst[1] = sat_u8(a); dst[2] = sat_u8(b);
    }
}

/* Lab tables, exported so that tests can hand the very same tables to checks */
const unsigned short *orc_lab_gamma_tab(void) { lab_init(); return g_gamma_tab; }
const unsigned short *orc_lab_cbrt_tab(void) { lab_init(); return g_cbrt_tab; }
const int *orc_lab_coeffs(void) { lab_init(); return g_lab_coeffs; }

/* =========================================================================================
 * cv::resize INTER_LINEAR (OpenCV 2.4 imgproc/imgwarp.cpp), PARITY UNPINNED.
 * Index/weight rule: f = (float)((d+0.5)*scale - 0.5); s = floor(f); f -= s; x axis clamps
 * (s<0 -> s=0,f=0; s>=w-1 -> s=w-1,f=0); y axis keeps its weights and clips the two rows.
 * ========================================================================================= */
static void resize_coeffs(int ssize, int dsize, int clamp_weights, int *ofs, float *w0, float *w1) {
    double inv_scale = (double)dsize / ssize;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_weights) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        w0[d] = 1.f - f;
        w1[d] = f;
    }
}

static int reflect(int p, int len) { /* BORDER_REFLECT: fedcba|abcdefgh|hgfedcb */
    if (p < 0) return -p - 1;
    if (p >= len) return 2 * len - p - 1;
    return p;
}
static int clipi(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

void orc_resize_patch_u8(const uint8_t *lab, int W, int H, int x0, int y0, int size, int r, uint8_t *dst) {
    int xofs[64], yofs[64];
    float ax0[64], ax1[64], ay0[64], ay1[64];
    resize_coeffs(size, r, 1, xofs, ax0, ax1);
    resize_coeffs(size, r, 0, yofs, ay0, ay1);
    /* 8-bit path: weights quantised to 11 bits, saturate_cast<short>(w*2048) = round half even */
    short ia0[64], ia1[64], ib0[64], ib1[64];
    for (int d = 0; d < r; d++) {
        ia0[d] = (short)lrintf(ax0[d] * 2048.f); ia1[d] = (short)lrintf(ax1[d] * 2048.f);
        ib0[d] = (short)lrintf(ay0[d] * 2048.f); ib1[d] = (short)lrintf(ay1[d] * 2048.f);
    }
    for (int dy = 0; dy < r; dy++) {
        int sy0 = clipi(yofs[dy], 0, size), sy1 = clipi(yofs[dy] + 1, 0, size);
        int ry0 = reflect(y0 + sy0, H), ry1 = reflect(y0 + sy1, H);
        for (int dx = 0; dx < r; dx++) {
            int sx0 = xofs[dx];
            int sx1 = sx0 + 1 < size ? sx0 + 1 : sx0; /* weight is 0 there (xmax tail, S*ONE) */
            int rx0 = reflect(x0 + sx0, W), rx1 = reflect(x0 + sx1, W);
            for (int c = 0; c < 3; c++) {
                int r0 = lab[((size_t)ry0 * W + rx0) * 3 + c] * ia0[dx] + lab[((size_t)ry0 * W + rx1) * 3 + c] * ia1[dx];
                int r1 = lab[((size_t)ry1 * W + rx0) * 3 + c] * ia0[dx] + lab[((size_t)ry1 * W + rx1) * 3 + c] * ia1[dx];
                int v = (((ib0[dy] * (r0 >> 4)) >> 16) + ((ib1[dy] * (r1 >> 4)) >> 16) + 2) >> 2;
                dst[(dy * r + dx) * 3 + c] = sat_u8(v);
            }
        }
    }
}

void orc_resize_linear_f32(const float *src, int sw, int sh, int C, float *dst, int dw, int dh) {
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw), *yofs = (int *)malloc(sizeof(int) * (size_t)dh);
    float *ax0 = (float *)malloc(sizeof(float) * (size_t)dw), *ax1 = (float *)malloc(sizeof(float) * (size_t)dw);
    float *ay0 = (float *)malloc(sizeof(float) * (size_t)dh), *ay1 = (float *)malloc(sizeof(float) * (size_t)dh);
    resize_coeffs(sw, dw, 1, xofs, ax0, ax1);
    resize_coeffs(sh, dh, 0, yofs, ay0, ay1);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clipi(yofs[dy], 0, sh), sy1 = clipi(yofs[dy] + 1, 0, sh);
        const float *S0 = src + (size_t)sy0 * sw * C, *S1 = src + (size_t)sy1 * sw * C;
        for (int dx = 0; dx < dw; dx++) {
            int sx0 = xofs[dx];
            int tail = sx0 + 1 >= sw; /* dx >= xmax: D = S[sx]*ONE */
            for (int c = 0; c < C; c++) {
                float h0, h1;
                if (tail) {
                    h0 = S0[(size_t)sx0 * C + c] * 1.f;
                    h1 = S1[(size_t)sx0 * C + c] * 1.f;
                } else {
                    h0 = S0[(size_t)sx0 * C + c] * ax0[dx] + S0[(size_t)(sx0 + 1) * C + c] * ax1[dx];
                    h1 = S1[(size_t)sx0 * C + c] * ax0[dx] + S1[(size_t)(sx0 + 1) * C + c] * ax1[dx];
                }
                dst[((size_t)dy * dw + dx) * C + c] = h0 * ay0[dy] + h1 * ay1[dy];
            }
        }
    }
    free(xofs); free(yofs); free(ax0); free(ax1); free(ay0); free(ay1);
}

/* =========================================================================================
 * Back-projection (feature_extractor.h:200-232)
 * ========================================================================================= */
void orc_cloud(const orc_params *p, const uint16_t *depth, const float *calib, float *cloud) {
    const float *Kinv = calib, *R = calib + 9, *t = calib + 18;
    /* A = R * Kinv (Eigen fixed 3x3 product: row . column, accumulated left to right) */
    float A[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            A[i * 3 + j] = (R[i * 3 + 0] * Kinv[0 * 3 + j] + R[i * 3 + 1] * Kinv[1 * 3 + j]) + R[i * 3 + 2] * Kinv[2 * 3 + j];
    const int W = p->width, H = p->height;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t idx = (size_t)y * W + x;
            float d = (float)depth[idx] / 1000.0f; /* :209 */
            float m0, m1, m2;
            if (d < p->depth_min || d > p->depth_max) { m0 = m1 = m2 = NAN; } /* :210-213 */
            else { m0 = d * (float)x; m1 = d * (float)y; m2 = d; }              /* :215-217 */
            for (int i = 0; i < 3; i++)
                cloud[idx * 3 + i] = ((A[i * 3 + 0] * m0 + A[i * 3 + 1] * m1) + A[i * 3 + 2] * m2) + t[i]; /* :223 */
        }
}

/* =========================================================================================
 * pcl::IntegralImageNormalEstimation, AVERAGE_3D_GRADIENT (feature_extractor.h:254-261).
 * PCL is not vendored and not installed: PARITY UNPINNED.  Restated after PCL 1.7
 * features/impl/integral_image_normal.hpp: depth-change map, two-pass chamfer distance map
 * (1.0 / 1.4 steps, float, raster order), window = int(min(distance, 10)), 10-px NaN border,
 * central-difference 3-D gradients summed over the window, normal = normalise(sum_dy x sum_dx).
 * Build-owned deviation: PCL accumulates the integral images in double; this build sums the
 * gradients as 2^-32 fixed-point int64 so that the window sum is exact and independent of the
 * summation order (the GPU sums the window directly).
 * ========================================================================================= */
static int finite3(const float *v) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

static int64_t to_fix32(float v) {
    double s = (double)v * 4294967296.0;
    if (s > 9.0e18) s = 9.0e18;
    if (s < -9.0e18) s = -9.0e18;
    return (int64_t)llrint(s);
}

void orc_normals_nz(int W, int H, const float *cloud, float *nz_out, float *dist_out) {
    const size_t N = (size_t)W * H;
    unsigned char *change = (unsigned char *)malloc(N);
    float *dist = (float *)malloc(N * sizeof(float));
    memset(change, 255, N);
    const float factor = 0.02f;
    for (int ri = 0; ri < H - 1; ri++)
        for (int ci = 0; ci < W - 1; ci++) {
            size_t index = (size_t)ri * W + ci;
            const float depth = cloud[index * 3 + 2];
            const float depthR = cloud[(index + 1) * 3 + 2];
            const float depthD = cloud[(index + W) * 3 + 2];
            const float thr = (factor * (fabsf(depth) + 1.0f) * 2.0f);
            if (fabsf(depth - depthR) > thr || !isfinite(depth) || !isfinite(depthR)) {
                change[index] = 0; change[index + 1] = 0;
            }
            if (fabsf(depth - depthD) > thr || !isfinite(depth) || !isfinite(depthD)) {
                change[index] = 0; change[index + W] = 0;
            }
        }
    for (size_t i = 0; i < N; i++) dist[i] = change[i] == 0 ? 0.0f : (float)(W + H);
    /* first pass (flat indexing reproduces the row wrap of previous_row[ci+1] at ci = W-1) */
    for (int ri = 1; ri < H; ri++)
        for (int ci = 1; ci < W; ci++) {
            float *prev = dist + (size_t)(ri - 1) * W, *cur = dist + (size_t)ri * W;
            const float upLeft = prev[ci - 1] + 1.4f, up = prev[ci] + 1.0f, upRight = prev[ci + 1] + 1.4f;
            const float left = cur[ci - 1] + 1.0f, center = cur[ci];
            const float a = upLeft < up ? upLeft : up, b = left < upRight ? left : upRight;
            const float mn = a < b ? a : b;
            if (mn < center) cur[ci] = mn;
        }
    /* second pass */
    for (int ri = H - 2; ri >= 0; ri--)
        for (int ci = W - 2; ci >= 0; ci--) {
            float *next = dist + (size_t)(ri + 1) * W, *cur = dist + (size_t)ri * W;
            const float lowerLeft = next[ci - 1] + 1.4f, lower = next[ci] + 1.0f, lowerRight = next[ci + 1] + 1.4f;
            const float right = cur[ci + 1] + 1.0f, center = cur[ci];
            const float a = lowerLeft < lower ? lowerLeft : lower, b = right < lowerRight ? right : lowerRight;
            const float mn = a < b ? a : b;
            if (mn < center) cur[ci] = mn;
        }
    if (dist_out) memcpy(dist_out, dist, N * sizeof(float));

    const int border = 10;
    for (size_t i = 0; i < N; i++) nz_out[i] = NAN;
    for (int ri = border; ri < H - border; ri++)
        for (int ci = border; ci < W - border; ci++) {
            size_t index = (size_t)ri * W + ci;
            if (!isfinite(cloud[index * 3 + 2])) continue;
            float smoothing = dist[index] < 10.0f ? dist[index] : 10.0f;
            if (!(smoothing > 2.0f)) continue;
            int rect = (int)smoothing, rect2 = rect / 2;
            int sx = ci - rect2, sy = ri - rect2;
            int64_t gx[3] = { 0, 0, 0 }, gy[3] = { 0, 0, 0 };
            unsigned cnt_x = 0, cnt_y = 0;
            for (int y = sy; y < sy + rect; y++)
                for (int x = sx; x < sx + rect; x++) {
                    /* window never leaves the image: rect2 <= 5 < border */
                    float dx[3] = { 0, 0, 0 }, dy[3] = { 0, 0, 0 };
                    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
                        const float *r = cloud + ((size_t)y * W + x + 1) * 3, *l = cloud + ((size_t)y * W + x - 1) * 3;
                        const float *dn = cloud + ((size_t)(y + 1) * W + x) * 3, *up = cloud + ((size_t)(y - 1) * W + x) * 3;
                        for (int k = 0; k < 3; k++) { dx[k] = r[k] - l[k]; dy[k] = dn[k] - up[k]; }
                    }
                    if (finite3(dx)) { cnt_x++; for (int k = 0; k < 3; k++) gx[k] += to_fix32(dx[k]); }
                    if (finite3(dy)) { cnt_y++; for (int k = 0; k < 3; k++) gy[k] += to_fix32(dy[k]); }
                }
            if (cnt_x == 0 || cnt_y == 0) continue;
            double GX[3], GY[3];
            for (int k = 0; k < 3; k++) { GX[k] = (double)gx[k] * (1.0 / 4294967296.0); GY[k] = (double)gy[k] * (1.0 / 4294967296.0); }
            /* normal_vector = gradient_y.cross(gradient_x) */
            double n0 = GY[1] * GX[2] - GY[2] * GX[1];
            double n1 = GY[2] * GX[0] - GY[0] * GX[2];
            double n2 = GY[0] * GX[1] - GY[1] * GX[0];
            double len2 = (n0 * n0 + n1 * n1) + n2 * n2;
            if (len2 == 0.0) continue;
            nz_out[index] = (float)(n2 / sqrt(len2));
        }
    free(change); free(dist);
}

/* =========================================================================================
 * FeatureExtractor::extract, NO_LABEL branch (feature_extractor.h:41-291)
 * ========================================================================================= */
int orc_extract(const orc_params *p, const uint8_t *rgb, const uint16_t *depth, const float *calib,
                float *feat, int *x_v, int *y_v) {
    const int W = p->width, H = p->height, s = p->stride, D = orc_feature_length(p);
    const float d_min_mm = (float)(p->depth_min * 1000.0), d_max_mm = (float)(p->depth_max * 1000.0); /* :43-44 */
    int P = 0;
    for (int y = 0; y < H; y += s)       /* :56-71 */
        for (int x = 0; x < W; x += s) {
            float dv = (float)depth[(size_t)y * W + x];
            if (dv >= d_min_mm && dv <= d_max_mm) { x_v[P] = x; y_v[P] = y; P++; }
        }
    int pos = 0;
    if (p->feature_color_patch) { /* :125-175 */
        uint8_t *lab = (uint8_t *)malloc((size_t)W * H * 3);
        orc_bgr2lab_u8(rgb, lab, W * H);
        const int r = p->patch_size_reduce;
        uint8_t patch[64 * 64 * 3];
        for (int i = 0; i < P; i++) {
            int x = x_v[i], y = y_v[i];
            float dm = (float)depth[(size_t)y * W + x] / 1000.0f;  /* :139 */
            int half = (int)(p->patch_size / (2.0 * dm));          /* :140 */
            int size = half * 2 + 1;
            orc_resize_patch_u8(lab, W, H, x - half, y - half, size, r, patch); /* :142 */
            for (int k = 0; k < r * r * 3; k++) feat[(size_t)i * D + pos + k] = (float)patch[k]; /* :160-167 */
        }
        free(lab);
        pos += r * r * 3;
    }
    if (p->feature_depth) { /* :180-197 */
        for (int i = 0; i < P; i++) feat[(size_t)i * D + pos] = (float)depth[(size_t)y_v[i] * W + x_v[i]] / 1000.0f;
        pos++;
    }
    float *cloud = NULL;
    if (p->feature_height || p->feature_normal) {
        cloud = (float *)malloc((size_t)W * H * 3 * sizeof(float));
        orc_cloud(p, depth, calib, cloud);
    }
    if (p->feature_height) { /* :236-251 */
        for (int i = 0; i < P; i++) feat[(size_t)i * D + pos] = cloud[((size_t)y_v[i] * W + x_v[i]) * 3 + 2];
        pos++;
    }
    if (p->feature_normal) { /* :254-291 */
        float *nz = (float *)malloc((size_t)W * H * sizeof(float));
        orc_normals_nz(W, H, cloud, nz, NULL);
        for (int i = 0; i < P; i++) {
            float v = nz[(size_t)y_v[i] * W + x_v[i]];
            feat[(size_t)i * D + pos] = isnan(v) ? -2.0f : orc_acos_f32(fabsf(v)); /* :275-283 */
        }
        free(nz);
        pos++;
    }
    free(cloud);
    return P;
}

/* =========================================================================================
 * RF driver: scatter, up-sample, pack (segmenter.cpp:351-431)
 * ========================================================================================= */
int orc_rf_frame(const orc_params *p, const orc_forest *f, int multi, const uint8_t *rgb,
                 const uint16_t *depth, const float *calib, float *posteriors) {
    const int W = p->width, H = p->height, s = p->stride, D = orc_feature_length(p);
    const int lw = W / s, lh = H / s;
    int cc[64], L;
    if (multi) L = orc_forest_layers(f, cc, 64);
    else { L = 1; cc[0] = orc_forest_single_classes(f); }
    int S = 0;
    for (int l = 0; l < L; l++) S += cc[l];
    size_t cap = (size_t)(H / s + 1) * (size_t)(W / s + 1);
    float *feat = (float *)malloc(cap * D * sizeof(float));
    int *x_v = (int *)malloc(cap * sizeof(int)), *y_v = (int *)malloc(cap * sizeof(int));
    int P = orc_extract(p, rgb, depth, calib, feat, x_v, y_v);
    float **low = (float **)malloc(sizeof(float *) * (size_t)L);
    for (int l = 0; l < L; l++) {
        size_t n = (size_t)lw * lh * cc[l];
        low[l] = (float *)malloc(n * sizeof(float));
        for (size_t i = 0; i < n; i++) low[l][i] = p->fill_value; /* segmenter.cpp:358-362 */
    }
    float *post = (float *)malloc(sizeof(float) * (size_t)(S > 0 ? S : 1));
    for (int j = 0; j < P; j++) { /* segmenter.cpp:366-376 */
        if (multi) orc_forest_multi_class_log_posterior(f, feat + (size_t)j * D, post);
        else orc_forest_class_log_posterior(f, feat + (size_t)j * D, post);
        int o = 0;
        for (int l = 0; l < L; l++) {
            /* result_ims[layer].ptr<float>(y/stride) + C*x/stride  (integer division of C*x) */
            float *dst = low[l] + (size_t)(y_v[j] / s) * lw * cc[l] + (size_t)(cc[l] * x_v[j] / s);
            for (int c = 0; c < cc[l]; c++) dst[c] = post[o + c];
            o += cc[l];
        }
    }
    size_t off = 0;
    for (int l = 0; l < L; l++) { /* segmenter.cpp:380-382,413-431 */
        orc_resize_linear_f32(low[l], lw, lh, cc[l], posteriors + off, W, H);
        off += (size_t)W * H * cc[l];
        free(low[l]);
    }
    free(low); free(post); free(feat); free(x_v); free(y_v);
    return P;
}

/* src/segmenter.cpp:561-616.  label_distribution of one image is [layer][y][x][class] (:413-431);
 * `offset` advances by H*W*C_l per layer (:613), image_index counts pixels in raster order (:598-611). */
void orc_fuse_posteriors(int n_images, int W, int H, const int32_t *index_images, const float *posteriors, int n_layers,
                         const int *class_counts, int cloud_size, float *unaries) {
    size_t S = 0;
    for (int l = 0; l < n_layers; l++) S += (size_t)class_counts[l];
    for (size_t i = 0; i < (size_t)cloud_size * S; i++) unaries[i] = 0.0f;      /* Constant(label_count, cloud_size, 0.0), :566 */
    const size_t pixels = (size_t)W * H;
    for (int m = 0; m < n_images; m++) {
        const int32_t *index_image = index_images + (size_t)m * pixels;
        const float *label_distribution = posteriors + (size_t)m * pixels * S;
        size_t offset = 0, uoff = 0;
        for (int l = 0; l < n_layers; l++) {
            const size_t C = (size_t)class_counts[l];
            size_t image_index = 0;
            for (int y = 0; y < H; y++) {
                for (int x = 0; x < W; x++) {
                    const int index = index_image[(size_t)y * W + x];
                    if (index >= 0 && index < cloud_size) {
                        for (size_t c = 0; c < C; c++)
                            unaries[uoff + (size_t)index * C + c] += label_distribution[offset + image_index * C + c];   /* :607 */
                    }
                    image_index++;
                }
            }
            offset += pixels * C;
            uoff += (size_t)cloud_size * C;
        }
    }
}

void orc_labels(const float *values, int N, int C, int mode, int unknown_label, int8_t *labels) {
    for (int i = 0; i < N; i++) {
        const float *v = values + (size_t)i * C;
        int best;
        float mx;
        if (mode == 0) { /* test.cpp:160-175 */
            best = -1; mx = -1000.f;
            for (int c = 0; c < C; c++) if (v[c] > mx) { mx = v[c]; best = c; }
        } else if (mode == 1) { /* segmenter.cpp:646-657 */
            best = unknown_label; mx = (float)(2.0 / C);
            for (int c = 0; c < C; c++) if (v[c] > mx) { mx = v[c]; best = c; }
        } else if (mode == 2) { /* segmenter.cpp:664-679 */
            best = unknown_label; mx = -1000.f;
            float sum = 0;
            for (int c = 0; c < C; c++) { sum += v[c]; if (v[c] > mx) { mx = v[c]; best = c; } }
            if (!(sum != 0.0f)) best = unknown_label;
        } else { /* densecrf.cpp:202-211: Eigen maxCoeff = first maximum */
            best = 0; mx = v[0];
            for (int c = 1; c < C; c++) if (v[c] > mx) { mx = v[c]; best = c; }
        }
        labels[i] = (int8_t)best;
    }
}

/* =========================================================================================
 * Permutohedral lattice, SSE branch (permutohedral.cpp:54-321, 476-603).  PARITY UNPINNED.
 * ========================================================================================= */
typedef struct {
    size_t key_size, filled, capacity;
    short *keys;
    int *table;
} hash_t;

static size_t hash_key(const hash_t *h, const short *k) { /* :80-87 */
    size_t r = 0;
    for (size_t i = 0; i < h->key_size; i++) { r += (size_t)(long)k[i]; r *= 1664525; }
    return r;
}
static void hash_init(hash_t *h, int key_size, int n_elements) { /* :89-90 */
    h->key_size = (size_t)key_size; h->filled = 0; h->capacity = 2 * (size_t)n_elements;
    if (h->capacity < 2) h->capacity = 2;
    h->keys = (short *)calloc((h->capacity / 2 + 10) * h->key_size, sizeof(short));
    h->table = (int *)malloc(h->capacity * sizeof(int));
    for (size_t i = 0; i < h->capacity; i++) h->table[i] = -1;
}
static void hash_grow(hash_t *h) { /* :59-79 */
    size_t old_cap = h->capacity;
    h->capacity *= 2;
    h->keys = (short *)realloc(h->keys, (old_cap + 10) * h->key_size * sizeof(short));
    int *old = h->table;
    h->table = (int *)malloc(h->capacity * sizeof(int));
    for (size_t i = 0; i < h->capacity; i++) h->table[i] = -1;
    for (size_t i = 0; i < old_cap; i++)
        if (old[i] >= 0) {
            int e = old[i];
            size_t hh = hash_key(h, h->keys + (size_t)e * h->key_size) % h->capacity;
            for (; h->table[hh] >= 0; hh = hh < h->capacity - 1 ? hh + 1 : 0);
            h->table[hh] = e;
        }
    free(old);
}
static int hash_find(hash_t *h, const short *k, int create) { /* :99-127 */
    if (2 * h->filled >= h->capacity) hash_grow(h);
    size_t hh = hash_key(h, k) % h->capacity;
    for (;;) {
        int e = h->table[hh];
        if (e == -1) {
            if (!create) return -1;
            for (size_t i = 0; i < h->key_size; i++) h->keys[h->filled * h->key_size + i] = k[i];
            h->table[hh] = (int)h->filled;
            return (int)h->filled++;
        }
        int good = 1;
        for (size_t i = 0; i < h->key_size && good; i++)
            if (h->keys[(size_t)e * h->key_size + i] != k[i]) good = 0;
        if (good) return e;
        hh++;
        if (hh == h->capacity) hh = 0;
    }
}

orc_lattice *orc_lattice_init(const float *feature, int N, int d) {
    orc_lattice *L = (orc_lattice *)calloc(1, sizeof(orc_lattice));
    L->N = N; L->d = d;
    hash_t ht;
    hash_init(&ht, d, N); /* :144 */
    const int bs = 4;     /* blocksize = sizeof(__m128)/sizeof(float) */
    const float invdplus1 = 1.0f / (d + 1), dplus1 = (float)(d + 1);
    size_t tot = (size_t)(d + 1) * ((size_t)N + 16);
    L->offset = (int *)calloc(tot, sizeof(int));
    L->barycentric = (float *)calloc(tot, sizeof(float));
    L->rank = (float *)calloc(tot, sizeof(float));
    float *scale_factor = (float *)malloc(sizeof(float) * (size_t)(d + 1));
    float *elevated = (float *)malloc(sizeof(float) * (size_t)(d + 1));
    float *rem0 = (float *)malloc(sizeof(float) * (size_t)(d + 1));
    float *rank = (float *)malloc(sizeof(float) * (size_t)(d + 1));
    float *bary = (float *)malloc(sizeof(float) * (size_t)(d + 2));
    short *canonical = (short *)malloc(sizeof(short) * (size_t)(d + 1) * (d + 1));
    short *key = (short *)malloc(sizeof(short) * (size_t)(d + 1));
    for (int i = 0; i <= d; i++) { /* :169-174 */
        for (int j = 0; j <= d - i; j++) canonical[i * (d + 1) + j] = (short)i;
        for (int j = d - i + 1; j <= d; j++) canonical[i * (d + 1) + j] = (short)(i - (d + 1));
    }
    float inv_std_dev = (float)(sqrt(2.0 / 3.0) * (d + 1));                                  /* :177 */
    for (int i = 0; i < d; i++) scale_factor[i] = (float)(1.0 / sqrt((double)((i + 2) * (i + 1))) * inv_std_dev); /* :179-180 */

    int Npad = (N + bs - 1) / bs * bs; /* padded lanes carry zero features and ARE inserted (:196,261-275) */
    for (int k = 0; k < Npad; k++) {
        const float *f = feature + (size_t)k * d;
        /* elevate (:201-207) */
        float sm = 0;
        for (int j = d; j > 0; j--) {
            float fv = k < N ? f[j - 1] : 0.0f;
            float cf = fv * scale_factor[j - 1];
            elevated[j] = sm - (float)j * cf;
            sm += cf;
        }
        elevated[0] = sm;
        /* closest 0-coloured simplex (:210-220): round half to even of elevated * 1/(d+1) */
        float sum = 0;
        for (int i = 0; i <= d; i++) {
            float v = invdplus1 * elevated[i];
            v = rintf(v);
            rem0[i] = v * dplus1;
            sum += v;
        }
        /* rank (:223-233) */
        for (int i = 0; i <= d; i++) rank[i] = 0;
        for (int i = 0; i < d; i++) {
            float di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= d; j++) {
                float dj = elevated[j] - rem0[j];
                float c = di < dj ? 1.0f : 0.0f;
                rank[i] += c;
                rank[j] += 1.0f - c;
            }
        }
        /* bring back onto the plane (:236-242) */
        for (int i = 0; i <= d; i++) {
            rank[i] += sum;
            float add = rank[i] < 0 ? dplus1 : 0.0f;
            float sub = rank[i] >= dplus1 ? dplus1 : 0.0f;
            rank[i] += add - sub;
            rem0[i] += add - sub;
        }
        /* barycentric (:245-263) */
        for (int i = 0; i < d + 2; i++) bary[i] = 0;
        for (int i = 0; i <= d; i++) {
            float v = (elevated[i] - rem0[i]) * invdplus1;
            int pidx = d - (int)rank[i];
            bary[pidx] += v;
            bary[pidx + 1] -= v;
        }
        bary[0] += 1 + bary[d + 1];
        /* vertices and offsets (:266-275) */
        for (int remainder = 0; remainder <= d; remainder++) {
            for (int i = 0; i < d; i++)
                key[i] = (short)(rem0[i] + (float)canonical[remainder * (d + 1) + (int)rank[i]]);
            size_t o = (size_t)k * (d + 1) + remainder;
            L->offset[o] = hash_find(&ht, key, 1);
            L->rank[o] = rank[remainder];
            L->barycentric[o] = bary[remainder];
        }
    }
    L->M = (int)ht.filled;
    /* blur neighbours (:296-318); for j == d the +-d write lands on the ignored coordinate */
    L->blur_n1 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * (size_t)(L->M > 0 ? L->M : 1));
    L->blur_n2 = (int *)malloc(sizeof(int) * (size_t)(d + 1) * (size_t)(L->M > 0 ? L->M : 1));
    short *n1 = (short *)malloc(sizeof(short) * (size_t)(d + 1)), *n2 = (short *)malloc(sizeof(short) * (size_t)(d + 1));
    for (int j = 0; j <= d; j++)
        for (int i = 0; i < L->M; i++) {
            const short *kk = ht.keys + (size_t)i * d;
            for (int k = 0; k < d; k++) { n1[k] = (short)(kk[k] - 1); n2[k] = (short)(kk[k] + 1); }
            if (j < d) { n1[j] = (short)(kk[j] + d); n2[j] = (short)(kk[j] - d); }
            L->blur_n1[(size_t)j * L->M + i] = hash_find(&ht, n1, 0);
            L->blur_n2[(size_t)j * L->M + i] = hash_find(&ht, n2, 0);
        }
    L->keys = (short *)malloc(sizeof(short) * (size_t)d * (size_t)(L->M > 0 ? L->M : 1));
    memcpy(L->keys, ht.keys, sizeof(short) * (size_t)d * (size_t)L->M);
    free(n1); free(n2); free(scale_factor); free(elevated); free(rem0); free(rank); free(bary);
    free(canonical); free(key); free(ht.keys); free(ht.table);
    return L;
}

void orc_lattice_free(orc_lattice *l) {
    if (!l) return;
    free(l->offset); free(l->barycentric); free(l->rank); free(l->blur_n1); free(l->blur_n2); free(l->keys);
    free(l);
}

void orc_lattice_compute_seq(const orc_lattice *l, float *out, const float *in, int vs, int reverse) {
    const int N = l->N, d = l->d, M = l->M;
    size_t tot = (size_t)(M + 2) * vs;
    float *values = (float *)calloc(tot, sizeof(float)), *new_values = (float *)calloc(tot, sizeof(float));
    for (int i = 0; i < N; i++)          /* splat :487-494 */
        for (int j = 0; j <= d; j++) {
            int o = l->offset[(size_t)i * (d + 1) + j] + 1;
            float w = l->barycentric[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; k++) values[(size_t)o * vs + k] += w * in[(size_t)i * vs + k];
        }
    for (int j = reverse ? d : 0; j <= d && j >= 0; reverse ? j-- : j++) { /* blur :496-510 */
        for (int i = 0; i < M; i++) {
            float *old_val = values + (size_t)(i + 1) * vs, *new_val = new_values + (size_t)(i + 1) * vs;
            int n1 = l->blur_n1[(size_t)j * M + i] + 1, n2 = l->blur_n2[(size_t)j * M + i] + 1;
            float *n1v = values + (size_t)n1 * vs, *n2v = values + (size_t)n2 * vs;
            for (int k = 0; k < vs; k++) new_val[k] = (float)((double)old_val[k] + 0.5 * (double)(n1v[k] + n2v[k]));
        }
        float *t = values; values = new_values; new_values = t;
    }
    float alpha = 1.0f / (1 + powf(2, (float)-d)); /* :512 */
    for (int i = 0; i < N; i++) {                  /* slice :515-524 */
        for (int k = 0; k < vs; k++) out[(size_t)i * vs + k] = 0;
        for (int j = 0; j <= d; j++) {
            int o = l->offset[(size_t)i * (d + 1) + j] + 1;
            float w = l->barycentric[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; k++) out[(size_t)i * vs + k] += w * values[(size_t)o * vs + k] * alpha;
        }
    }
    free(values); free(new_values);
}

void orc_lattice_compute_sse(const orc_lattice *l, float *out, const float *in, int vs, int reverse) {
    const int N = l->N, d = l->d, M = l->M;
    size_t tot = (size_t)(M + 2) * vs;
    float *values = (float *)calloc(tot, sizeof(float)), *new_values = (float *)calloc(tot, sizeof(float));
    float *val = (float *)malloc(sizeof(float) * (size_t)vs);
    for (int i = 0; i < N; i++) {        /* splat :545-553 */
        memcpy(val, in + (size_t)i * vs, sizeof(float) * (size_t)vs);
        for (int j = 0; j <= d; j++) {
            int o = l->offset[(size_t)i * (d + 1) + j] + 1;
            float w = l->barycentric[(size_t)i * (d + 1) + j];
            for (int k = 0; k < vs; k++) { float prod = w * val[k]; values[(size_t)o * vs + k] += prod; }
        }
    }
    for (int j = reverse ? d : 0; j <= d && j >= 0; reverse ? j-- : j++) { /* blur :556-569 */
        for (int i = 0; i < M; i++) {
            float *old_val = values + (size_t)(i + 1) * vs, *new_val = new_values + (size_t)(i + 1) * vs;
            int n1 = l->blur_n1[(size_t)j * M + i] + 1, n2 = l->blur_n2[(size_t)j * M + i] + 1;
            float *n1v = values + (size_t)n1 * vs, *n2v = values + (size_t)n2 * vs;
            for (int k = 0; k < vs; k++) { float s = n1v[k] + n2v[k]; float h = 0.5f * s; new_val[k] = old_val[k] + h; }
        }
        float *t = values; values = new_values; new_values = t;
    }
    float alpha = 1.0f / (1 + powf(2, (float)-d)); /* :571 */
    for (int i = 0; i < N; i++) {                  /* slice :574-584 */
        for (int k = 0; k < vs; k++) val[k] = 0;
        for (int j = 0; j <= d; j++) {
            int o = l->offset[(size_t)i * (d + 1) + j] + 1;
            float w = l->barycentric[(size_t)i * (d + 1) + j] * alpha;
            for (int k = 0; k < vs; k++) { float prod = w * values[(size_t)o * vs + k]; val[k] += prod; }
        }
        memcpy(out + (size_t)i * vs, val, sizeof(float) * (size_t)vs);
    }
    free(values); free(new_values); free(val);
}

void orc_lattice_compute(const orc_lattice *l, float *out, const float *in, int vs, int reverse) {
    if (vs <= 2) orc_lattice_compute_seq(l, out, in, vs, reverse); /* :600-603 */
    else orc_lattice_compute_sse(l, out, in, vs, reverse);
}

void orc_kernel_norm(const orc_lattice *l, float *norm) { /* pairwise.cpp:40-56 */
    const int N = l->N;
    float *ones = (float *)malloc(sizeof(float) * (size_t)N);
    for (int i = 0; i < N; i++) ones[i] = 1.0f;
    orc_lattice_compute(l, norm, ones, 1, 0);
    for (int i = 0; i < N; i++) norm[i] = (float)(1.0 / sqrt((double)norm[i] + 1e-20));
    free(ones);
}

void orc_exp_and_normalize(float *out, const float *in, int N, int C) { /* densecrf.cpp:98-106 */
    for (int i = 0; i < N; i++) {
        const float *b = in + (size_t)i * C;
        float *o = out + (size_t)i * C;
        float mx = b[0];
        for (int c = 1; c < C; c++) if (b[c] > mx) mx = b[c];
        float sum = 0;
        for (int c = 0; c < C; c++) { o[c] = orc_exp_f32(b[c] - mx); sum += o[c]; }
        for (int c = 0; c < C; c++) o[c] = o[c] / sum;
    }
}

void orc_crf_inference_multi(int N, int C, int n_kernels, const int *ds, const float *const *features,
                             const float *ws, const float *unary_energy, int iterations, float *Q) {
    orc_lattice **lat = (orc_lattice **)malloc(sizeof(orc_lattice *) * (size_t)n_kernels);
    float **norm = (float **)malloc(sizeof(float *) * (size_t)n_kernels);
    for (int k = 0; k < n_kernels; k++) {
        lat[k] = orc_lattice_init(features[k], N, ds[k]);
        norm[k] = (float *)malloc(sizeof(float) * (size_t)N);
        orc_kernel_norm(lat[k], norm[k]);
    }
    size_t tot = (size_t)N * C;
    float *tmp1 = (float *)malloc(sizeof(float) * tot), *tmp2 = (float *)malloc(sizeof(float) * tot);
    for (size_t i = 0; i < tot; i++) tmp1[i] = -unary_energy[i];
    orc_exp_and_normalize(Q, tmp1, N, C); /* densecrf.cpp:120 */
    for (int it = 0; it < iterations; it++) {
        for (size_t i = 0; i < tot; i++) tmp1[i] = -unary_energy[i]; /* :123 */
        for (int k = 0; k < n_kernels; k++) {
            /* DenseKernel::filter, pairwise.cpp:63-80 */
            for (int i = 0; i < N; i++)
                for (int c = 0; c < C; c++) tmp2[(size_t)i * C + c] = Q[(size_t)i * C + c] * norm[k][i];
            orc_lattice_compute(lat[k], tmp2, tmp2, C, 0);
            for (int i = 0; i < N; i++)
                for (int c = 0; c < C; c++) tmp2[(size_t)i * C + c] = tmp2[(size_t)i * C + c] * norm[k][i];
            /* PottsCompatibility::apply, labelcompatibility.cpp:46-48 */
            float mw = -ws[k];
            for (size_t i = 0; i < tot; i++) tmp2[i] = mw * tmp2[i];
            for (size_t i = 0; i < tot; i++) tmp1[i] -= tmp2[i]; /* densecrf.cpp:126 */
        }
        orc_exp_and_normalize(Q, tmp1, N, C); /* :128 */
    }
    for (int k = 0; k < n_kernels; k++) { orc_lattice_free(lat[k]); free(norm[k]); }
    free(lat); free(norm); free(tmp1); free(tmp2);
}

void orc_crf_inference(int N, int C, int d, const float *unary_energy, const float *feature,
                       float potts_w, int iterations, float *Q) {
    const float *feats[1] = { feature };
    orc_crf_inference_multi(N, C, 1, &d, feats, &potts_w, unary_energy, iterations, Q);
}

void orc_frame_crf_features(const orc_params *p, const uint8_t *rgb, const float *cloud, float *feat) {
    const size_t N = (size_t)p->width * p->height;
    for (size_t i = 0; i < N; i++) { /* segmenter.cpp:629-637 */
        float x = cloud[i * 3], y = cloud[i * 3 + 1], z = cloud[i * 3 + 2];
        if (!(isfinite(x) && isfinite(y) && isfinite(z))) x = y = z = 0.0f;
        feat[i * 6 + 0] = x * p->dcrf_xyz_kernel;
        feat[i * 6 + 1] = y * p->dcrf_xyz_kernel;
        feat[i * 6 + 2] = z * p->dcrf_xyz_kernel;
        for (int c = 0; c < 3; c++) feat[i * 6 + 3 + c] = ((float)rgb[i * 3 + c] / 255.0f) * p->dcrf_rgb_kernel;
    }
}

int orc_segment_frame(const orc_params *p, const orc_forest *f, int multi, const uint8_t *rgb,
                      const uint16_t *depth, const float *calib, float *posteriors, float *marginals,
                      int8_t *labels, int label_mode, const int *unknown_labels) {
    const int W = p->width, H = p->height;
    const size_t N = (size_t)W * H;
    int cc[64], L;
    if (multi) L = orc_forest_layers(f, cc, 64);
    else { L = 1; cc[0] = orc_forest_single_classes(f); }
    int P = orc_rf_frame(p, f, multi, rgb, depth, calib, posteriors);
    float *cloud = (float *)malloc(N * 3 * sizeof(float));
    orc_cloud(p, depth, calib, cloud);
    float *feat = (float *)malloc(N * 6 * sizeof(float));
    orc_frame_crf_features(p, rgb, cloud, feat);
    size_t off = 0;
    for (int l = 0; l < L; l++) { /* one fresh DenseCRF per layer, segmenter.cpp:639-644 */
        size_t tot = N * cc[l];
        float *U = (float *)malloc(tot * sizeof(float));
        for (size_t i = 0; i < tot; i++) U[i] = -posteriors[off + i]; /* setUnaryEnergy(-unaries[l]) */
        orc_crf_inference((int)N, cc[l], 6, U, feat, p->dcrf_kernel_weight, p->dcrf_iterations, marginals + off);
        if (labels) orc_labels(marginals + off, (int)N, cc[l], label_mode, unknown_labels ? unknown_labels[l] : cc[l] - 1, labels + (size_t)l * N);
        free(U);
        off += tot;
    }
    free(cloud); free(feat);
    return P;
}
