/*
 * oracle/vpp_oracle.c -- CPU restatement of the reference's VPP scan kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * there only as the checker.  The product path is vppstereo_amd/csrc (HIP).
 *
 * Parity status: PINNED.  Checked bit-for-bit against the reference's own Cython
 * build of vpp_core/vpp_core_opt.pyx (tests/golden/make_vpp_golden.py generates the
 * fixtures; tests/test_oracle_vpp.py replays them).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference repo root).  The arithmetic follows the C that Cython 3.2.9 generates
 * from the .pyx, i.e. C usual-arithmetic-conversions on (uint8_t, float, double):
 *     uint8 * float  -> float product          (c, c_occ, beta are float32)
 *     (1.0 - float)  -> double
 *     uint8 * double -> double
 *     (uint8_t)(double) truncates toward zero
 * Compile with -ffp-contract=off (no FMA) and without -ffast-math.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* glibc 2.35 rand()/srand() (TYPE_3 additive feedback generator).           */
/* Reference: vpp_core/vpp_core_opt.pyx:33-35 (init_rand -> srand((int)seed)) */
/* and :93,:102 (rand() % 256).  The .pyx uses libc's global generator; this  */
/* is a bit-exact restatement of glibc's random_r.c / srandom_r so the stream */
/* can be reproduced on any host and on the GPU.                              */
/* ------------------------------------------------------------------------- */
typedef struct {
    int32_t r[31];
    int f, b; /* front / rear indices */
} vppo_rand_state;

static vppo_rand_state g_rs;
static int g_rs_init = 0;

void vppo_rand_seed(vppo_rand_state *s, unsigned int seed)
{
    int32_t word;
    int i;
    if (seed == 0)
        seed = 1;
    word = (int32_t)seed;
    s->r[0] = word;
    for (i = 1; i < 31; i++) {
        /* Schrage: word = 16807 * word % 2147483647 without overflow */
        long hi = word / 127773;
        long lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0)
            word += 2147483647;
        s->r[i] = word;
    }
    s->f = 3; /* SEP_3 */
    s->b = 0;
    for (i = 0; i < 310; i++) { /* discard 10 * DEG_3 outputs */
        uint32_t v = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
        s->r[s->f] = (int32_t)v;
        if (++s->f >= 31) s->f = 0;
        if (++s->b >= 31) s->b = 0;
    }
}

int vppo_rand_next(vppo_rand_state *s)
{
    uint32_t v = (uint32_t)s->r[s->f] + (uint32_t)s->r[s->b];
    s->r[s->f] = (int32_t)v;
    if (++s->f >= 31) s->f = 0;
    if (++s->b >= 31) s->b = 0;
    return (int)(v >> 1);
}

/* libc-like global interface used by the scans */
void vppo_srand(unsigned int seed)
{
    vppo_rand_seed(&g_rs, seed);
    g_rs_init = 1;
}

int vppo_rand(void)
{
    if (!g_rs_init)
        vppo_srand(1);
    return vppo_rand_next(&g_rs);
}

/* Fill out[0..n) with the first n outputs after srand(seed). */
void vppo_rand_stream(unsigned int seed, int64_t n, int32_t *out)
{
    vppo_rand_state s;
    int64_t i;
    vppo_rand_seed(&s, seed);
    for (i = 0; i < n; i++)
        out[i] = vppo_rand_next(&s);
}

/* ------------------------------------------------------------------------- */
/* helpers                                                                    */
/* ------------------------------------------------------------------------- */
/* Cython memoryview indexing with boundscheck=False, wraparound=True
 * (vpp_core_opt.pyx:1): a negative index i becomes i + dim. */
static inline int wrapx(int i, int w) { return i < 0 ? i + w : i; }

#define LIDX(y, x, j) (((size_t)(y) * (size_t)width + (size_t)(x)) * (size_t)channels + (size_t)(j))

/* numba-twin extras (vpp_standalone.py:7-11): per-hint patch radius.
 * round() there is Python round on a float64 -> half-to-even -> nearbyint. */
static int patch_radius_from_distance(float d_ref, float d_min, float d_max, int patch_size, double gamma)
{
    /* numba typing: float32 - float32 and float32 / float32 stay float32; ** (1/gamma) is float64 */
    volatile float num = d_ref - d_min;
    volatile float den = d_max - d_min;
    volatile float ratio = num / den;
    double gamma_weight = pow((double)ratio, 1.0 / gamma);
    double ws = nearbyint(gamma_weight * (double)(patch_size - 1) + 1.0);
    long long wsize = (long long)ws;
    /* Python floor division */
    long long a = wsize - 1;
    long long n = a >= 0 ? a / 2 : -((-a + 1) / 2);
    return (int)n;
}

int vppo_patch_radius(float d_ref, float d_min, float d_max, int patch_size, double gamma)
{
    return patch_radius_from_distance(d_ref, d_min, d_max, patch_size, gamma);
}

/* ------------------------------------------------------------------------- */
/* virtual_projection_scan_rnd -- vpp_core/vpp_core_opt.pyx:53-131           */
/* Extended (_ex) form adds the two numba-only gates of                       */
/* vpp_standalone.py:243-369 (filled_g gate :335, distance patch :315-318);  */
/* with filled_g==NULL and use_distance_patch==0 it is exactly the .pyx.      */
/* NOTE (vpp_standalone.py:339-340): in the numba twin the random draw sits   */
/* INSIDE the bilateral gate; the _ex form follows that when the gate is on.  */
/* ------------------------------------------------------------------------- */
int vppo_scan_rnd_ex(uint8_t *l, uint8_t *r, const float *g, int width, int height, int channels,
                     int uniform_color, int wsize, int direction, float c, float c_occ,
                     const uint8_t *g_occ, int discard_occluded, int interpolate,
                     const float *filled_g, int use_distance_patch, float dmin, float dmax,
                     double distance_gamma)
{
    int sample_i = 0;
    int x, y, j, xd1, d1, d0, xd0, xd, d;
    float d1_blending;
    int n = (wsize - 1) / 2; /* :60 (wsize >= 1) */
    int xw, yw;
    uint8_t rvalue = 0;

    for (y = 0; y < height; y++) {                                   /* :77 */
        x = (direction == 0) ? width - 1 : 0;                        /* :78 */
        while ((direction != 0 && x < width) || (direction == 0 && x >= 0)) { /* :80 */
            float gv = g[(size_t)y * width + x];
            if (gv > 0) {                                            /* :81 */
                int nn = n;
                d = (int)round((double)gv);                          /* :82 half away from zero */
                d0 = (int)floor((double)gv);                         /* :83 */
                d1 = (int)ceil((double)gv);                          /* :84 */
                d1_blending = gv - (float)d0;                        /* :85 float32 */
                xd = x - d;                                          /* :87 */
                xd0 = x - d0;                                        /* :88 */
                xd1 = x - d1;                                        /* :89 */
                if (use_distance_patch)                              /* vpp_standalone.py:315-318 */
                    nn = patch_radius_from_distance(gv, dmin, dmax, wsize, distance_gamma);
                for (j = 0; j < channels; j++) {                     /* :90 */
                    if (uniform_color)
                        rvalue = (uint8_t)(vppo_rand() % 256);       /* :92-93 */
                    for (yw = -nn; yw <= nn; yw++) {                 /* :97 */
                        for (xw = -nn; xw <= nn; xw++) {             /* :98 */
                            if (0 <= y + yw && y + yw <= height - 1 && 0 <= x + xw && x + xw <= width - 1) { /* :99 */
                                size_t li = LIDX(y + yw, x + xw, j);
                                if (filled_g) { /* vpp_standalone.py:335 */
                                    float fg = filled_g[(size_t)(y + yw) * width + (x + xw)];
                                    /* numba: filled_g is float64 there (np.where(.., f32, 0) promotes),
                                     * so the gate is |f32 - f64| < 0.1 evaluated in float64 */
                                    if (!(fabs((double)gv - (double)fg) < 0.1))
                                        continue;
                                }
                                if (!uniform_color)
                                    rvalue = (uint8_t)(vppo_rand() % 256); /* :101-102 */
                                if (0 <= xd0 + xw && xd0 + xw <= width - 1) { /* :104 */
                                    if (g_occ[(size_t)y * width + x] == 0) { /* :106 */
                                        l[li] = (uint8_t)(rvalue * c + l[li] * (1.0 - c)); /* :107 */
                                        if (interpolate) {
                                            size_t r0 = LIDX(y + yw, xd0 + xw, j);
                                            r[r0] = (uint8_t)(((rvalue * c + r[r0] * (1.0 - c)) * (1.0 - d1_blending)) + r[r0] * d1_blending); /* :109 */
                                            if (0 <= xd1 + xw && xd1 + xw <= width - 1) { /* :110 */
                                                size_t r1 = LIDX(y + yw, xd1 + xw, j);
                                                r[r1] = (uint8_t)(((rvalue * c + r[r1] * (1.0 - c)) * d1_blending) + r[r1] * (1.0 - d1_blending)); /* :111 */
                                            }
                                        } else {
                                            size_t rd = LIDX(y + yw, wrapx(xd + xw, width), j);
                                            r[rd] = (uint8_t)(rvalue * c + r[rd] * (1.0 - c)); /* :113 */
                                        }
                                    } else if (!discard_occluded) {  /* :114 */
                                        if (interpolate) {
                                            size_t r0 = LIDX(y + yw, xd0 + xw, j);
                                            size_t r1 = LIDX(y + yw, wrapx(xd1 + xw, width), j);
                                            r[r0] = (uint8_t)(((rvalue * c_occ + r[r0] * (1.0 - c_occ)) * (1.0 - d1_blending)) + r[r0] * d1_blending); /* :116 */
                                            if (0 <= xd1 + xw && xd1 + xw <= width - 1) /* :117 */
                                                r[r1] = (uint8_t)(((rvalue * c_occ + r[r1] * (1.0 - c_occ)) * d1_blending) + r[r1] * (1.0 - d1_blending)); /* :118 */
                                            l[li] = (uint8_t)((r[r0] * (1.0 - d1_blending) + r[r1] * d1_blending) * c + l[li] * (1.0 - c)); /* :119 (r1 unguarded) */
                                        } else {
                                            size_t rd = LIDX(y + yw, wrapx(xd + xw, width), j);
                                            r[rd] = (uint8_t)(rvalue * c_occ + r[rd] * (1.0 - c_occ)); /* :121 */
                                            l[li] = (uint8_t)(r[rd] * c + l[li] * (1.0 - c));          /* :122 */
                                        }
                                    }
                                } else {                             /* :123 left-side occlusion */
                                    l[li] = (uint8_t)(rvalue * c + l[li] * (1.0 - c)); /* :124 */
                                }
                            }
                        }
                    }
                }
                sample_i += 1;                                       /* :127 */
            }
            x = (direction == 0) ? x - 1 : x + 1;                    /* :129 */
        }
    }
    return sample_i;
}

int vppo_scan_rnd(uint8_t *l, uint8_t *r, const float *g, int width, int height, int channels,
                  int uniform_color, int wsize, int direction, float c, float c_occ,
                  const uint8_t *g_occ, int discard_occluded, int interpolate)
{
    return vppo_scan_rnd_ex(l, r, g, width, height, channels, uniform_color, wsize, direction, c, c_occ,
                            g_occ, discard_occluded, interpolate, NULL, 0, 0.f, 0.f, 0.3);
}

/* ------------------------------------------------------------------------- */
/* maxDistance colour search -- vpp_core/vpp_core_opt.pyx:216-260 (uniform)   */
/* and :269-313 (per patch pixel).  (cy,cx) is the window centre in the left  */
/* image, rcx the centre column in the right image (uses the ROUNDED d).      */
/* bins_inside != 0 reproduces the uniform branch, where the n_bins/used_bins */
/* bookkeeping sits inside the pa<p<pb test (:235-237,:248-250).              */
/* ------------------------------------------------------------------------- */
static void maxdist_search(const uint8_t *l, const uint8_t *r, int width, int height, int channels, int j,
                           int cy, int cx, int rcx, int n_agg_x, int n_agg_y, int occluded,
                           int bins_inside, uint16_t *ppa, uint16_t *ppb)
{
    uint16_t pa = 0, pb = 255;
    int used_bins[256];
    int n_bins = 256;
    int k, yw_agg, xw_agg;
    for (k = 0; k < 256; k++)
        used_bins[k] = 0;
    for (yw_agg = -n_agg_y; yw_agg <= n_agg_y; yw_agg++) {
        for (xw_agg = -n_agg_x; xw_agg <= n_agg_x; xw_agg++) {
            int yy = cy + yw_agg, xx = cx + xw_agg, rx = rcx + xw_agg;
            if (0 <= yy && yy <= height - 1 && 0 <= xx && xx <= width - 1) {
                int r_in = (0 <= rx && rx <= width - 1);
                if (!occluded || !r_in) {
                    int p = l[LIDX(yy, xx, j)];
                    if (p > pa && p < pb) {
                        if (p - pa > pb - p)
                            pb = (uint16_t)p;
                        else if (p - pa < pb - p)
                            pa = (uint16_t)p;
                        if (bins_inside) {
                            if (p == 0) n_bins -= 1;
                            used_bins[p] += 1;
                        }
                    }
                    if (!bins_inside) {
                        if (p == 0) n_bins -= 1;
                        used_bins[p] += 1;
                    }
                }
                if (r_in) {
                    int p = r[LIDX(yy, rx, j)];
                    if (p > pa && p < pb) {
                        if (p - pa > pb - p)
                            pb = (uint16_t)p;
                        else if (p - pa < pb - p)
                            pa = (uint16_t)p;
                        if (bins_inside) {
                            if (p == 0) n_bins -= 1;
                            used_bins[p] += 1;
                        }
                    }
                    if (!bins_inside) {
                        if (p == 0) n_bins -= 1;
                        used_bins[p] += 1;
                    }
                }
            }
        }
    }
    if (n_bins == 0) { /* :252-260 / :305-313 */
        int min_bin_value = used_bins[0];
        int min_bin = 0;
        for (k = 0; k < 256; k++) {
            if (min_bin_value > used_bins[k]) {
                min_bin = k;
                min_bin_value = used_bins[k];
            }
        }
        pa = (uint16_t)min_bin;
        pb = (uint16_t)min_bin;
    }
    *ppa = pa;
    *ppb = pb;
}

/* ------------------------------------------------------------------------- */
/* virtual_projection_scan_max_dist -- vpp_core/vpp_core_opt.pyx:133-341     */
/* _ex adds the numba-only gates (vpp_standalone.py:93-96,154).               */
/* ------------------------------------------------------------------------- */
int vppo_scan_max_dist_ex(uint8_t *l, uint8_t *r, const float *g, int width, int height, int channels,
                          int uniform_color, int wsize, int wsize_agg_x, int wsize_agg_y, int direction,
                          float c, float c_occ, const uint8_t *g_occ, int discard_occluded, int interpolate,
                          const float *filled_g, int use_distance_patch, float dmin, float dmax,
                          double distance_gamma)
{
    int sample_i = 0;
    int x, y, j, d, xd, xd1, d1, d0, xd0;
    float d1_blending;
    int n = (wsize - 1) / 2;             /* :177 */
    int n_agg_x = (wsize_agg_x - 1) / 2; /* :179 */
    int n_agg_y = (wsize_agg_y - 1) / 2; /* :180 */
    uint16_t pa = 0, pb = 255;           /* :196-197 */
    int xw, yw;

    for (y = 0; y < height; y++) {                                   /* :200 */
        x = (direction == 0) ? width - 1 : 0;
        while ((direction != 0 && x < width) || (direction == 0 && x >= 0)) {
            float gv = g[(size_t)y * width + x];
            if (gv > 0) {                                            /* :204 */
                int nn = n;
                int occluded = g_occ[(size_t)y * width + x] != 0;
                d = (int)round((double)gv);
                d0 = (int)floor((double)gv);
                d1 = (int)ceil((double)gv);
                d1_blending = gv - (float)d0;
                xd = x - d;
                xd0 = x - d0;
                xd1 = x - d1;
                if (use_distance_patch)
                    nn = patch_radius_from_distance(gv, dmin, dmax, wsize, distance_gamma);
                for (j = 0; j < channels; j++) {                     /* :214 */
                    if (uniform_color)                               /* :216-260 */
                        maxdist_search(l, r, width, height, channels, j, y, x, xd, n_agg_x, n_agg_y, occluded, 1,
                                       &pa, &pb);
                    for (yw = -nn; yw <= nn; yw++) {                 /* :265 */
                        for (xw = -nn; xw <= nn; xw++) {
                            if (0 <= y + yw && y + yw <= height - 1 && 0 <= x + xw && x + xw <= width - 1) { /* :267 */
                                size_t li = LIDX(y + yw, x + xw, j);
                                double V;
                                if (filled_g) { /* vpp_standalone.py:154 */
                                    float fg = filled_g[(size_t)(y + yw) * width + (x + xw)];
                                    if (!(fabs((double)gv - (double)fg) < 0.1))
                                        continue;
                                }
                                if (!uniform_color)                  /* :269-313 */
                                    maxdist_search(l, r, width, height, channels, j, y + yw, x + xw, xd + xw, n_agg_x,
                                                   n_agg_y, occluded, 0, &pa, &pb);
                                V = ((double)(pa + pb)) / 2.0;
                                if (0 <= xd0 + xw && xd0 + xw <= width - 1) { /* :315 */
                                    if (!occluded) {                 /* :317 */
                                        l[li] = (uint8_t)(V * c + l[li] * (1.0 - c)); /* :318 */
                                        if (interpolate) {
                                            size_t r0 = LIDX(y + yw, xd0 + xw, j);
                                            r[r0] = (uint8_t)(((V * c + r[r0] * (1.0 - c)) * (1.0 - d1_blending)) + r[r0] * d1_blending); /* :320 */
                                            if (0 <= xd1 + xw && xd1 + xw <= width - 1) {
                                                size_t r1 = LIDX(y + yw, xd1 + xw, j);
                                                r[r1] = (uint8_t)(((V * c + r[r1] * (1.0 - c)) * d1_blending) + r[r1] * (1.0 - d1_blending)); /* :322 */
                                            }
                                        } else {
                                            size_t rd = LIDX(y + yw, wrapx(xd + xw, width), j);
                                            r[rd] = (uint8_t)(V * c + r[rd] * (1.0 - c)); /* :324 */
                                        }
                                    } else if (!discard_occluded) {  /* :325 */
                                        if (interpolate) {
                                            size_t r0 = LIDX(y + yw, xd0 + xw, j);
                                            size_t r1 = LIDX(y + yw, wrapx(xd1 + xw, width), j);
                                            r[r0] = (uint8_t)(((V * c_occ + r[r0] * (1.0 - c_occ)) * (1.0 - d1_blending)) + r[r0] * d1_blending); /* :327 */
                                            if (0 <= xd1 + xw && xd1 + xw <= width - 1)
                                                r[r1] = (uint8_t)(((V * c_occ + r[r1] * (1.0 - c_occ)) * d1_blending) + r[r1] * (1.0 - d1_blending)); /* :329 */
                                            l[li] = (uint8_t)((r[r0] * (1.0 - d1_blending) + r[r1] * d1_blending) * c + l[li] * (1.0 - c)); /* :330 */
                                        } else {
                                            size_t rd = LIDX(y + yw, wrapx(xd + xw, width), j);
                                            r[rd] = (uint8_t)(V * c_occ + r[rd] * (1.0 - c_occ)); /* :332 */
                                            l[li] = (uint8_t)(r[rd] * c + l[li] * (1.0 - c));     /* :333 */
                                        }
                                    }
                                } else {                             /* :334 */
                                    l[li] = (uint8_t)(V * c + l[li] * (1.0 - c)); /* :335 */
                                }
                            }
                        }
                    }
                }
                sample_i += 1;                                       /* :337 */
            }
            x = (direction == 0) ? x - 1 : x + 1;
        }
    }
    return sample_i;
}

int vppo_scan_max_dist(uint8_t *l, uint8_t *r, const float *g, int width, int height, int channels,
                       int uniform_color, int wsize, int wsize_agg_x, int wsize_agg_y, int direction, float c,
                       float c_occ, const uint8_t *g_occ, int discard_occluded, int interpolate)
{
    return vppo_scan_max_dist_ex(l, r, g, width, height, channels, uniform_color, wsize, wsize_agg_x, wsize_agg_y,
                                 direction, c, c_occ, g_occ, discard_occluded, interpolate, NULL, 0, 0.f, 0.f, 0.3);
}

/* ------------------------------------------------------------------------- */
/* gt_reshape -- vpp_core/vpp_core_opt.pyx:352-371 (dense hints -> N x 4)     */
/* ------------------------------------------------------------------------- */
int vppo_gt_reshape(const float *gt, int height, int width, float *out /* [H*W,4] */)
{
    int i = 0, y, x;
    for (y = 0; y < height; y++)
        for (x = 0; x < width; x++) {
            float v = gt[(size_t)y * width + x];
            if (v > 0) {
                out[4 * i + 0] = (float)x;
                out[4 * i + 1] = (float)y;
                out[4 * i + 2] = v;
                out[4 * i + 3] = 1.f;
                i++;
            }
        }
    return i;
}

/* ------------------------------------------------------------------------- */
/* _bilateral_filling -- vpp_standalone.py:372-394 (numba semantics:          */
/* img uint8 promoted to int64 before the subtraction, weights float64,       */
/* cmap/aug float32 arrays -> the weight is rounded to float32 on store and    */
/* compared as float32-promoted-to-float64).                                   */
/* ------------------------------------------------------------------------- */
void vppo_bilateral_filling(const float *dmap, const uint8_t *img, int h, int w, int n, double o_xy, double o_i,
                            double th, float *aug /* out [H,W] */)
{
    float *cmap = (float *)calloc((size_t)h * w, sizeof(float));
    int y, x, yw, xw;
    memcpy(aug, dmap, (size_t)h * w * sizeof(float));
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            int i_ref = img[(size_t)y * w + x];
            float d_ref = dmap[(size_t)y * w + x];
            if (d_ref > 0) {
                for (yw = -n; yw <= n; yw++)
                    for (xw = -n; xw <= n; xw++) {
                        if (0 <= y + yw && y + yw <= h - 1 && 0 <= x + xw && x + xw <= w - 1) {
                            size_t q = (size_t)(y + yw) * w + (x + xw);
                            long long di = (long long)img[q] - (long long)i_ref;
                            double weight = exp(-(((double)(yw * yw + xw * xw)) / (2.0 * (o_xy * o_xy)) +
                                                  ((double)(di * di)) / (2.0 * (o_i * o_i))));
                            if ((double)cmap[q] < weight) {
                                cmap[q] = (float)weight;
                                aug[q] = d_ref;
                            }
                        }
                    }
            }
        }
    for (y = 0; y < h * w; y++)
        if (!((double)cmap[y] > th))
            aug[y] = 0.f;
    free(cmap);
}
