/*
 * ym_oracle.c -- CPU restatement of the reference's correlative scan matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see ym_oracle.h).  Plain C99, fp64, compiled with
 * -ffp-contract=off so every expression rounds exactly as written.
 *
 * ONE matcher code path serves both semantics; the eighteen places where the Karto matcher and the
 * reference's Python matcher differ are explicit switches (ORC_D*, ym_oracle.h).  The Python side of
 * each cites /root/reference/yag_slam/{helpers,scan_matching}.py lines; the Karto side the open_karto
 * function it restates (source not in /root/reference; reached by the reference through
 * karto_scanmatcher==1.0.0, /root/reference/setup.py:46).
 */
#define _POSIX_C_SOURCE 199309L
#include "ym_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define KT_PI 3.14159265358979323846
#define KT_2PI 6.28318530717958647692
#define KT_TOLERANCE 1e-06
#define MAX_VARIANCE 500.0
#define DISTANCE_PENALTY_GAIN 0.2
#define ANGLE_PENALTY_GAIN 0.2
#define GRID_OCCUPIED 100

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static char g_err[256];
const char *orc_last_error(void) { return g_err; }
static int fail(const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return -1;
}

/* ------------------------------------------------------------------ karto math:: helpers */
/* math::Round: half away from zero */
static double kt_round(double v) { return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5); }
static int kt_double_equal(double a, double b) {
    double d = a - b;
    return d < 0.0 ? d >= -KT_TOLERANCE : d <= KT_TOLERANCE;
}
static double kt_normalize_angle(double angle) {
    while (angle < -KT_PI) {
        if (angle < -KT_2PI)
            angle += (double)(unsigned int)(angle / -KT_2PI) * KT_2PI;
        else
            angle += KT_2PI;
    }
    while (angle > KT_PI) {
        if (angle > KT_2PI)
            angle -= (double)(unsigned int)(angle / KT_2PI) * KT_2PI;
        else
            angle -= KT_2PI;
    }
    return angle;
}
/* math::NormalizeAngleDifference returns the adjusted minuend */
static double kt_normalize_angle_difference(double minuend, double subtrahend) {
    while (minuend - subtrahend < -KT_PI) minuend += KT_2PI;
    while (minuend - subtrahend > KT_PI) minuend -= KT_2PI;
    return minuend;
}
static int align8(int v) { return (v + 7) & ~7; }

/* ------------------------------------------------------------------ context */
struct orc_ctx {
    orc_config cfg;
    /* grid */
    uint8_t *grid;   /* u8 storage */
    double  *gridf;  /* yagpy f64 grid */
    int gw, gh, pitch, roi_x, roi_y, roi_w, roi_h;
    double off_x, off_y; /* world coordinate of ROI cell (0,0) */
    double scale;        /* 1/resolution (CoordinateConverter::m_Scale) */
    double res;          /* cell size as the active converter holds it (D5) */
    unsigned mask;       /* bit n set: delta n (ORC_D*) follows the Python path */
    /* smear kernel: cell values, and the Python path's float values (D1) */
    uint8_t *kernel;
    double *kernelf;
    int ksize;
    /* per-pass volumes */
    uint32_t *sums[2];
    double   *resp[2];
    int dims[2][3];
    /* karto search-space probabilities (side x side doubles) */
    double *probs;
    int side;
    /* points */
    double *raster_pts; int n_raster;
    double *qlocal;     int n_qlocal;
    double serial_s; /* wall time of the last match's single-threaded part (grid clear + rasterisation) */
};

static void free_pass(orc_ctx *c, int p) {
    free(c->sums[p]); c->sums[p] = NULL;
    free(c->resp[p]); c->resp[p] = NULL;
    c->dims[p][0] = c->dims[p][1] = c->dims[p][2] = 0;
}

orc_ctx *orc_create(const orc_config *cfg) {
    if (!cfg) { fail("null config"); return NULL; }
    if (cfg->resolution <= 0 || cfg->search_size <= 0 || cfg->smear_deviation < 0 ||
        cfg->range_threshold <= 0) { fail("invalid matcher parameters"); return NULL; }
    /* helpers.py:370 / CorrelationGrid::CalculateKernel smear bounds */
    if (!(0.5 * cfg->resolution <= cfg->smear_deviation &&
          cfg->smear_deviation <= 10 * cfg->resolution)) {
        fail("smear deviation must be between 0.5*resolution and 10*resolution");
        return NULL;
    }
    if (cfg->semantics != ORC_SEM_KARTO && cfg->semantics != ORC_SEM_YAGPY && cfg->semantics != ORC_SEM_MIXED) {
        fail("unknown semantics"); return NULL;
    }
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    c->cfg = *cfg;
    if (c->cfg.threads < 1) c->cfg.threads = 1;
    c->mask = cfg->semantics == ORC_SEM_KARTO ? 0u : cfg->semantics == ORC_SEM_YAGPY ? ORC_ALL_PY : (cfg->delta_mask & ORC_ALL_PY);
    return c;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    free(c->grid); free(c->gridf); free(c->kernel); free(c->kernelf); free(c->probs);
    free(c->raster_pts); free(c->qlocal);
    free_pass(c, 0); free_pass(c, 1);
    free(c);
}

/* ------------------------------------------------------------------ point readings */
/* yagpy: helpers.py:58-68 (_get_point_readings): keep unless r > rt or NaN.
 * karto: LocalizedRangeScan::Update: keep iff InRange(r, min_range, range_threshold). */
int orc_point_readings(const orc_scan *s, int semantics, double *xs, double *ys) {
    int n = 0;
    for (int i = 0; i < s->n; i++) {
        double r = s->ranges[i];
        if (semantics == ORC_SEM_YAGPY) {
            if (r > s->range_threshold || isnan(r)) continue;
        } else {
            if (!(r >= s->min_range && r <= s->range_threshold)) continue;
        }
        double angle = s->pose[2] + s->min_angle + i * s->angle_increment;
        xs[n] = s->pose[0] + r * cos(angle);
        ys[n] = s->pose[1] + r * sin(angle);
        n++;
    }
    return n;
}

/* ------------------------------------------------------------------ valid-point filter */
/* yagpy: helpers.py:298-329 (validate_points): d = 0.2 m, keep run (f_k, f_{k+1}] iff ss > 0,
 *        point 0 and the trailing run never kept.
 * karto: ScanMatcher::FindValidPoints: d = 0.1 m, keep run [trail, current) unless ss < 0. */
int orc_valid_points(const double *xs, const double *ys, int n, double vpx, double vpy,
                     int semantics, uint8_t *keep) {
    int kept = 0;
    memset(keep, 0, (size_t)n);
    if (n == 0) return 0;
    if (semantics == ORC_SEM_YAGPY) {
        const double msd = 0.2 * 0.2;
        double fpx = xs[0], fpy = ys[0];
        int run_start = 1; /* first index of the pending run */
        for (int i = 1; i < n; i++) {
            double cpx = xs[i], cpy = ys[i];
            if ((fpx - cpx) * (fpx - cpx) + (fpy - cpy) * (fpy - cpy) > msd) {
                double a = vpy - fpy;
                double b = fpx - vpx;
                double cc = fpy * vpx - fpx * vpy;
                fpx = cpx; fpy = cpy;
                double ss = cpx * a + cpy * b + cc;
                if (ss > 0.0)
                    for (int j = run_start; j <= i; j++) { keep[j] = 1; kept++; }
                run_start = i + 1;
            }
        }
    } else {
        const double min_sq = 0.1 * 0.1;
        int trail = 0;
        double fx = 0, fy = 0;
        int first_time = 1;
        for (int i = 0; i < n; i++) {
            double cx = xs[i], cy = ys[i];
            if (first_time && !isnan(cx) && !isnan(cy)) { fx = cx; fy = cy; first_time = 0; }
            double dx = fx - cx, dy = fy - cy;
            if (dx * dx + dy * dy > min_sq) {
                double a = vpy - fy;
                double b = fx - vpx;
                double cc = fy * vpx - fx * vpy;
                double ss = cx * a + cy * b + cc;
                fx = cx; fy = cy;
                if (ss < 0.0) {
                    trail = i;
                } else {
                    for (; trail != i; ++trail) { keep[trail] = 1; kept++; }
                }
            }
        }
    }
    return kept;
}

/* ------------------------------------------------------------------ kernels */
/* karto: CorrelationGrid::CalculateKernel; half = Round(2*smear/res) */
int orc_kernel_karto(double resolution, double smear, uint8_t *k) {
    double scale = 1.0 / resolution;
    double res = 1.0 / scale; /* Grid::GetResolution() */
    int half = (int)kt_round(2.0 * smear / res);
    int size = 2 * half + 1;
    if (!k) return size;
    for (int i = -half; i <= half; i++)
        for (int j = -half; j <= half; j++) {
            double d = hypot(i * res, j * res);
            double z = exp(-0.5 * pow(d / smear, 2));
            unsigned int v = (unsigned int)kt_round(z * GRID_OCCUPIED);
            k[(j + half) + size * (i + half)] = (uint8_t)v;
        }
    return size;
}

/* yagpy: helpers.py:86-97 (calculate_kernel); size = int(4*np.round(smear/res)+1) */
int orc_kernel_yagpy(double resolution, double smear, double *k) {
    int size = (int)(4 * rint(smear / resolution) + 1);
    if (!k) return size;
    int half = size / 2;
    for (int i_ = 0; i_ < size; i_++) {
        int i = i_ - half;
        for (int j_ = 0; j_ < size; j_++) {
            int j = j_ - half;
            double a = i * resolution, b = j * resolution;
            double sqdist = a * a + b * b;
            k[i_ * size + j_] = exp(-0.5 * sqdist / (smear * smear));
        }
    }
    return size;
}

/* numpy.arange for float64: len = ceil((stop-start)/step); v[0]=start, v[1]=start+step,
 * v[i] = start + i*((start+step)-start)  (numpy DOUBLE_fill). */
int orc_arange(double start, double stop, double step, double *out, int cap) {
    double q = (stop - start) / step;
    int len = (int)ceil(q);
    if (len < 0) len = 0;
    if (!out) return len;
    if (len > cap) len = cap;
    if (len > 0) out[0] = start;
    if (len > 1) out[1] = start + step;
    if (len > 2) {
        double delta = out[1] - start;
        for (int i = 2; i < len; i++) out[i] = start + i * delta;
    }
    return len;
}

/* ================================================================== the matcher, one code path
 *
 * The Karto matcher and the reference's in-tree Python matcher are the same algorithm with the EIGHTEEN
 * differences listed in ym_oracle.h (SURVEY.md Appendix B's seventeen + the lookup construction).  Each is one
 * switch, PY(c, Dn): 0 = what Karto does, 1 = what the Python path does.  ORC_SEM_KARTO is all switches 0,
 * ORC_SEM_YAGPY all switches 1, ORC_SEM_MIXED takes cfg.delta_mask.  Every statement outside a switch is shared,
 * and therefore pinned by the reference-generated golden vectors (tests/golden) that the all-Py setting must
 * reproduce; what stays unpinned in Karto mode is exactly the Karto branch of each switch. */
#define PY(c, d) ((int)(((c)->mask >> (d)) & 1u))

/* D5: CoordinateConverter::WorldToGrid = Round((w - offset) * scale)  |  helpers.py:81-83,139-140 np.round((w - o) / res) */
static int conv(const orc_ctx *c, double w, double off) {
    if (PY(c, ORC_D5_ROUNDING)) return (int)rint((w - off) / c->res);
    return (int)kt_round((w - off) * c->scale);
}

/* ScanMatcher::Create / CorrelationGrid::CreateGrid  |  scan_matching.py:183-190 */
static int u_setup_grid(orc_ctx *c) {
    const orc_config *g = &c->cfg;
    /* D5: Karto keeps scale = 1/resolution and derives the resolution back from it */
    c->scale = 1.0 / g->resolution;
    c->res = PY(c, ORC_D5_ROUNDING) ? g->resolution : 1.0 / c->scale;
    /* D2 kernel half size: Round(2 sigma / res)  |  size = int(4 * np.round(sigma / res) + 1) */
    int half = PY(c, ORC_D2_KERNEL_HALF) ? (int)(4 * rint(g->smear_deviation / g->resolution) + 1) / 2
                                         : (int)kt_round(2.0 * g->smear_deviation / c->res);
    int ksz = 2 * half + 1;
    /* D1 cell values: (u8)Round(100 exp(-(hypot/sigma)^2 / 2))  |  float exp(-sqdist / 2 sigma^2), scored as int(100 v) */
    free(c->kernel); free(c->kernelf);
    c->kernel = (uint8_t *)malloc((size_t)ksz * ksz);
    c->kernelf = NULL;
    c->ksize = ksz;
    if (PY(c, ORC_D1_CELL_VALUE)) {
        c->kernelf = (double *)malloc(sizeof(double) * ksz * ksz);
        for (int i_ = 0; i_ < ksz; i_++)
            for (int j_ = 0; j_ < ksz; j_++) {
                int i = i_ - half, j = j_ - half;
                double a = i * g->resolution, b = j * g->resolution;
                double sqdist = a * a + b * b;
                double v = exp(-0.5 * sqdist / (g->smear_deviation * g->smear_deviation));
                c->kernelf[i_ * ksz + j_] = v;
                c->kernel[i_ * ksz + j_] = (uint8_t)(int)(100 * v);
            }
    } else {
        for (int i = -half; i <= half; i++)
            for (int j = -half; j <= half; j++) {
                double d = hypot(i * c->res, j * c->res);
                double z = exp(-0.5 * pow(d / g->smear_deviation, 2));
                unsigned int v = (unsigned int)kt_round(z * GRID_OCCUPIED);
                c->kernel[(j + half) + ksz * (i + half)] = (uint8_t)v;
            }
    }
    /* D3 grid size / border: Round(S/r) + 1 + 2 ceil(rt/r) cells inside a (half + 1)-cell border, rows aligned to 8
     *                      | int(S/r + 1 + 2 rt/r) cells, no border (taps are bounds-checked instead) */
    int W, gsize, border, pitch;
    c->side = (int)(kt_round(g->search_size / g->resolution) + 1);
    if (PY(c, ORC_D3_GRID_SIZE)) {
        gsize = (int)(g->search_size / g->resolution + 1 + 2 * g->range_threshold / g->resolution);
        if (gsize <= 0) return fail("bad grid size");
        border = 0;
        W = gsize;
        pitch = W;
    } else {
        int margin = (int)ceil(g->range_threshold / g->resolution);
        gsize = c->side + 2 * margin;
        border = half + 1;
        W = gsize + 2 * border;
        pitch = align8(W);
    }
    if (!c->grid || c->gw != W || c->pitch != pitch) {
        free(c->grid);
        c->grid = (uint8_t *)malloc((size_t)pitch * W);
        if (!c->grid) return fail("grid alloc failed");
    }
    if (PY(c, ORC_D1_CELL_VALUE)) {
        if (!c->gridf || c->gw != W) {
            free(c->gridf);
            c->gridf = (double *)malloc(sizeof(double) * (size_t)W * W);
            if (!c->gridf) return fail("grid alloc failed");
        }
    } else {
        free(c->gridf);
        c->gridf = NULL;
    }
    c->gw = c->gh = W;
    c->pitch = pitch;
    c->roi_x = c->roi_y = border;
    c->roi_w = c->roi_h = gsize;
    free(c->probs);
    c->probs = (double *)malloc(sizeof(double) * c->side * c->side);
    return 0;
}

/* ScanMatcher::AddScans / AddScan / CorrelationGrid::SmearPoint  |  helpers.py:105-131 add_scan_to_grid / smear_point */
static int u_add_scans(orc_ctx *c, const orc_scan *base, int n_base, double vpx, double vpy) {
    memset(c->grid, 0, (size_t)c->pitch * c->gh);
    if (c->gridf) memset(c->gridf, 0, sizeof(double) * (size_t)c->gw * c->gh);
    int maxn = 0, total = 0;
    for (int b = 0; b < n_base; b++) { if (base[b].n > maxn) maxn = base[b].n; total += base[b].n; }
    double *xs = (double *)malloc(sizeof(double) * (maxn + 1));
    double *ys = (double *)malloc(sizeof(double) * (maxn + 1));
    uint8_t *keep = (uint8_t *)malloc((size_t)maxn + 1);
    free(c->raster_pts);
    c->raster_pts = (double *)malloc(sizeof(double) * 2 * (total + 1));
    c->n_raster = 0;
    const int half = c->ksize / 2, ksz = c->ksize;
    for (int b = 0; b < n_base; b++) {
        /* D7 range gating, D6 valid-point filter: see orc_point_readings / orc_valid_points */
        int n = orc_point_readings(&base[b], PY(c, ORC_D7_RANGE_GATE) ? ORC_SEM_YAGPY : ORC_SEM_KARTO, xs, ys);
        orc_valid_points(xs, ys, n, vpx, vpy, PY(c, ORC_D6_VALID_FILTER) ? ORC_SEM_YAGPY : ORC_SEM_KARTO, keep);
        for (int i = 0; i < n; i++) {
            if (!keep[i]) continue;
            c->raster_pts[2 * c->n_raster] = xs[i];
            c->raster_pts[2 * c->n_raster + 1] = ys[i];
            c->n_raster++;
            int gx = conv(c, xs[i], c->off_x);
            int gy = conv(c, ys[i], c->off_y);
            if (gx < 0 || gx >= c->roi_w || gy < 0 || gy >= c->roi_h) continue; /* not in grid */
            uint8_t *cell = c->grid + (size_t)(gy + c->roi_y) * c->pitch + (gx + c->roi_x);
            /* D8 an occupied cell: Karto skips the point ("value already set")  |  the Python path stamps again */
            if (!PY(c, ORC_D8_RESTAMP) && *cell == GRID_OCCUPIED) continue;
            *cell = GRID_OCCUPIED;
            if (c->gridf) c->gridf[(size_t)(gy + c->roi_y) * c->gw + (gx + c->roi_x)] = 1.0;
            if (!PY(c, ORC_D3_GRID_SIZE) && !c->gridf) {
                /* CorrelationGrid::SmearPoint as Karto runs it: the border makes every tap land in storage */
                for (int j = -half; j <= half; j++) {
                    uint8_t *row = c->grid + (size_t)(gy + j + c->roi_y) * c->pitch + (gx + c->roi_x);
                    const uint8_t *krow = c->kernel + ksz * (j + half) + half;
                    for (int t = -half; t <= half; t++)
                        if (krow[t] > row[t]) row[t] = krow[t];
                }
                continue;
            }
            for (int j = -half; j <= half; j++) {
                for (int t = -half; t <= half; t++) {
                    int x = gx + t, y = gy + j;
                    /* D3: Karto's border makes every tap land in storage  |  helpers.py:115 tests every tap */
                    if (PY(c, ORC_D3_GRID_SIZE) && !(0 <= x && x < c->roi_w && 0 <= y && y < c->roi_h)) continue;
                    uint8_t kv = c->kernel[ksz * (j + half) + (t + half)];
                    uint8_t *p = c->grid + (size_t)(y + c->roi_y) * c->pitch + (x + c->roi_x);
                    if (kv > *p) *p = kv;
                    if (c->gridf) {
                        double kf = c->kernelf[ksz * (j + half) + (t + half)];
                        double *pf = c->gridf + (size_t)(y + c->roi_y) * c->gw + (x + c->roi_x);
                        if (kf > *pf) *pf = kf;
                    }
                }
            }
        }
    }
    free(xs); free(ys); free(keep);
    return 0;
}

/* one search lattice: per-axis hypothesis coordinates */
typedef struct {
    int nx, ny, nt;
    double *xoff, *yoff;   /* offsets from the search centre (Karto's x, y) */
    double *xabs, *yabs;   /* hypothesis positions (Karto's newPositionX/Y, the Python path's xvals/yvals) */
    double *tabs;          /* hypothesis headings before normalisation */
    double off_x, off_y, step_x, step_y, angle_off, angle_res; /* Karto's lattice parameters (covariances) */
} lattice;

static void lattice_free(lattice *L) {
    free(L->xoff); free(L->yoff); free(L->xabs); free(L->yabs); free(L->tabs);
    memset(L, 0, sizeof *L);
}

/* CorrelateScan's loop bounds: n = Round(2 off / step) + 1 points at centre - off + i step (inclusive, symmetric)
 *                            | helpers.py:177-179 np.arange(-search + c, search + c, step) (exclusive end) */
static void lattice_axis(int py_arange, double centre, double off, double step, int *n_out, double **offs, double **abss) {
    int n;
    double *o, *a;
    if (py_arange) {
        n = orc_arange(-off + centre, off + centre, step, NULL, 0);
        o = (double *)malloc(sizeof(double) * (n + 1));
        a = (double *)malloc(sizeof(double) * (n + 1));
        orc_arange(-off + centre, off + centre, step, a, n);
        for (int i = 0; i < n; i++) o[i] = a[i] - centre;
    } else {
        n = (int)(kt_round(off * 2.0 / step) + 1);
        o = (double *)malloc(sizeof(double) * (n + 1));
        a = (double *)malloc(sizeof(double) * (n + 1));
        double start = -off;
        for (int i = 0; i < n; i++) {
            o[i] = start + i * step;
            a[i] = centre + o[i];
        }
    }
    *n_out = n; *offs = o; *abss = a;
}

static void lattice_make(lattice *L, int py_arange, const double center[3], double off, double step, double angle_off,
                         double angle_res) {
    memset(L, 0, sizeof *L);
    L->off_x = L->off_y = off; L->step_x = L->step_y = step; L->angle_off = angle_off; L->angle_res = angle_res;
    lattice_axis(py_arange, center[0], off, step, &L->nx, &L->xoff, &L->xabs);
    lattice_axis(py_arange, center[1], off, step, &L->ny, &L->yoff, &L->yabs);
    if (py_arange) {
        double *dummy;
        lattice_axis(1, center[2], angle_off, angle_res, &L->nt, &dummy, &L->tabs);
        free(dummy);
    } else {
        L->nt = (int)(kt_round(angle_off * 2.0 / angle_res) + 1);
        L->tabs = (double *)malloc(sizeof(double) * (L->nt + 1));
        double start = center[2] - angle_off;
        for (int k = 0; k < L->nt; k++) L->tabs[k] = start + k * angle_res;
    }
}

/* ScanMatcher::GetResponse -> integer sum (the division is the caller's, exactly as Karto writes it) */
static uint32_t k_get_sum(const orc_ctx *c, const int32_t *lk, int np, int grid_index) {
    uint32_t sum = 0;
    int data_size = c->pitch * c->gh;
    const uint8_t *p = c->grid + grid_index;
    for (int i = 0; i < np; i++) {
        int idx = grid_index + lk[i];
        if (idx < 0 || idx >= data_size) continue; /* IsUpTo */
        sum += p[lk[i]];
    }
    return sum;
}

/* helpers.py:134-153 score_world_points_on_grid: every point of every hypothesis is rounded on its own */
static uint32_t y_score(const orc_ctx *c, const double *px, const double *py, int n, double shift_x, double shift_y) {
    uint32_t sum = 0;
    for (int l = 0; l < n; l++) {
        double x = shift_x + px[l];
        double y = shift_y + py[l];
        int _x = conv(c, x, c->off_x), _y = conv(c, y, c->off_y);
        if (_x >= 0 && _x < c->roi_w && _y >= 0 && _y < c->roi_h)
            sum += c->grid[(size_t)(_y + c->roi_y) * c->pitch + (_x + c->roi_x)];
    }
    return sum;
}

typedef struct {
    double best;        /* Karto: best response (clamped)  |  Python: out[argmax] */
    double mean[3];
    int ii, jj, kk;     /* Python: argmax indices */
} pass_result;

#define VOL(L, ix, iy, k) ((((size_t)(iy)) * (L)->nx + (ix)) * (L)->nt + (k))

/* ScanMatcher::CorrelateScan  |  helpers.py:156-295 find_best_pose.  Volumes are stored [iy][ix][it]. */
static int u_correlate(orc_ctx *c, int pass, const double center[3], const lattice *L, int penalize, pass_result *o,
                       double cov[9], int *err) {
    const orc_config *g = &c->cfg;
    const int np = c->n_qlocal, nx = L->nx, ny = L->ny, nt = L->nt;
    size_t nh = (size_t)nx * ny * nt;
    free_pass(c, pass);
    if (nh == 0) return fail("empty search lattice");
    c->sums[pass] = (uint32_t *)malloc(sizeof(uint32_t) * nh);
    c->resp[pass] = (double *)malloc(sizeof(double) * nh);
    c->dims[pass][0] = nx; c->dims[pass][1] = ny; c->dims[pass][2] = nt;
    uint32_t *sums = c->sums[pass];
    double *resp = c->resp[pass];

    /* D18 lookup construction.
     * Karto: GridIndexLookup::ComputeOffsets -- per angle, rotate the sensor-frame point, WorldToGrid the rotated offset
     *        ONCE (relative to the grid origin), keep the linear cell offset; a hypothesis adds it to its own cell.
     * Python: helpers.py:76-78,199 -- per angle rotate the points, then round EVERY (hypothesis + point) sum. */
    int32_t *lk = NULL;
    double *rx = NULL, *ry = NULL;
    if (PY(c, ORC_D18_LOOKUP)) {
        rx = (double *)malloc(sizeof(double) * (size_t)nt * (np + 1));
        ry = (double *)malloc(sizeof(double) * (size_t)nt * (np + 1));
        for (int k = 0; k < nt; k++) {
            double ca = cos(L->tabs[k]), sa = sin(L->tabs[k]);
            for (int l = 0; l < np; l++) {
                double lx = c->qlocal[2 * l], ly = c->qlocal[2 * l + 1];
                rx[(size_t)k * np + l] = lx * ca - ly * sa;
                ry[(size_t)k * np + l] = ly * ca + lx * sa;
            }
        }
    } else {
        lk = (int32_t *)malloc(sizeof(int32_t) * (size_t)nt * (np > 0 ? np : 1));
        for (int k = 0; k < nt; k++) {
            double angle = L->tabs[k];
            double cosine = cos(angle), sine = sin(angle);
            for (int i = 0; i < np; i++) {
                double px = c->qlocal[2 * i], py = c->qlocal[2 * i + 1];
                double ox = cosine * px - sine * py;
                double oy = sine * px + cosine * py;
                int gx = conv(c, ox + c->off_x, c->off_x); /* WorldToGrid(offset + gridOffset) */
                int gy = conv(c, oy + c->off_y, c->off_y);
                lk[(size_t)k * np + i] = gx + gy * c->pitch; /* base GridIndex, ROI ignored */
            }
        }
    }
    /* D12 (Python): penalty centre = grid corner + half the grid, helpers.py:173-174 */
    const double sx_ = c->off_x + c->roi_w * c->res / 2;
    const double sy_ = c->off_y + c->roi_w * c->res / 2;
    const int data_size = c->pitch * c->gh;
    int bad_index = 0;

#ifdef _OPENMP
#pragma omp parallel for collapse(2) num_threads(g->threads) schedule(static) if (g->threads > 1)
#endif
    for (int iy = 0; iy < ny; iy++) {
        for (int ix = 0; ix < nx; ix++) {
            double y = L->yoff[iy], x = L->xoff[ix];
            double new_y = L->yabs[iy], new_x = L->xabs[ix];
            double sq_y = y * y, sq_x = x * x;
            int grid_index = 0;
            if (!PY(c, ORC_D18_LOOKUP)) {
                int gx = conv(c, new_x, c->off_x) + c->roi_x;
                int gy = conv(c, new_y, c->off_y) + c->roi_y;
                grid_index = gx + gy * c->pitch;
                if (gx < 0 || gx >= c->gw || gy < 0 || gy >= c->gh || grid_index < 0 || grid_index >= data_size) {
                    bad_index = 1;
                    grid_index = 0;
                }
            }
            for (int k = 0; k < nt; k++) {
                double angle = L->tabs[k];
                size_t h = VOL(L, ix, iy, k);
                uint32_t s = PY(c, ORC_D18_LOOKUP) ? y_score(c, rx + (size_t)k * np, ry + (size_t)k * np, np, new_x, new_y)
                                                   : k_get_sum(c, lk + (size_t)k * np, np, grid_index);
                /* D12 penalty.  Karto: config variances, clamped from below, distance from the search centre, only for
                 * a non-zero response  |  Python: constants 0.5 / 1.0 divided by the resolution, no clamps, distance
                 * from the grid centre, always (helpers.py:181-184,200-210) */
                double pen = 1.0;
                int pen_on = 0;
                if (penalize) {
                    if (PY(c, ORC_D12_PENALTY)) {
                        const double dist_var_penalty = 0.5, ang_var_penalty = 1.0;
                        double sd = (new_x - sx_) * (new_x - sx_) + (new_y - sy_) * (new_y - sy_);
                        double dist_penalty = 1.0 - 0.2 * sd / (dist_var_penalty * c->res);
                        double sa2 = (angle - center[2]) * (angle - center[2]);
                        double ang_penalty = 1.0 - 0.2 * sa2 / (ang_var_penalty * c->res);
                        pen = dist_penalty * ang_penalty;
                        pen_on = 1;
                    } else {
                        double sq_dist = sq_x + sq_y;
                        double dp = 1.0 - (DISTANCE_PENALTY_GAIN * sq_dist / g->distance_variance_penalty);
                        if (dp < g->minimum_distance_penalty) dp = g->minimum_distance_penalty;
                        double sq_ang = (angle - center[2]) * (angle - center[2]);
                        double ap = 1.0 - (ANGLE_PENALTY_GAIN * sq_ang / g->angle_variance_penalty);
                        if (ap < g->minimum_angle_penalty) ap = g->minimum_angle_penalty;
                        pen = (dp * ap);
                        pen_on = 2; /* applied below, only to a non-zero response */
                    }
                }
                /* D11 normaliser: response = sum; response /= nPoints * 100  |  res / len(pts) * penalty / 100 */
                double response;
                if (PY(c, ORC_D11_NORMALISER)) {
                    double p = pen;
                    if (pen_on == 2) {
                        double r0 = np == 0 ? 0.0 : (double)s / (double)(np * GRID_OCCUPIED);
                        if (kt_double_equal(r0, 0.0)) p = 1.0;
                    }
                    response = (double)s / np * p / 100.0;
                } else {
                    response = 0.0;
                    if (np != 0) {
                        response = (double)s;
                        response /= (double)(np * GRID_OCCUPIED);
                    }
                    if (pen_on == 1) response *= pen;
                    else if (pen_on == 2 && !kt_double_equal(response, 0.0)) response *= pen;
                }
                sums[h] = s;
                resp[h] = response;
            }
        }
    }
    free(lk); free(rx); free(ry);
    if (bad_index) *err = fail("hypothesis grid index out of range");

    /* D13 best + tie set.  Karto: max; all hypotheses with DoubleEqual(response, best) (1e-6); mean position, circular
     * mean of the NORMALISED headings (D16); visited y, x, theta  |  Python: np.argmax (first maximum, x-major order),
     * all with response >= best - 1e-8, arithmetic means, visited x, y, theta (helpers.py:214-244) */
    double avg[3] = {0, 0, 0};
    double best = -1;
    o->ii = o->jj = o->kk = 0;
    if (PY(c, ORC_D13_TIES)) {
        int mi = 0, mj = 0, mk = 0, stop = 0;
        for (int i = 0; i < nx && !stop; i++)
            for (int j = 0; j < ny && !stop; j++)
                for (int k = 0; k < nt; k++) {
                    double cur = resp[VOL(L, mi, mj, mk)];
                    if (isnan(cur)) { stop = 1; break; } /* numpy: the first NaN is the maximum */
                    double v = resp[VOL(L, i, j, k)];
                    if (v > cur || isnan(v)) { mi = i; mj = j; mk = k; }
                }
        o->ii = mi; o->jj = mj; o->kk = mk;
        best = resp[VOL(L, mi, mj, mk)];
        double bx = 0, by = 0, bt = 0, norm_ = 0.0;
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++)
                for (int k = 0; k < nt; k++)
                    if (resp[VOL(L, i, j, k)] >= best - 0.00000001) {
                        bx += L->xabs[i]; by += L->yabs[j];
                        bt += PY(c, ORC_D16_EXPANSION_CLAMP) ? L->tabs[k] : kt_normalize_angle(L->tabs[k]);
                        norm_ += 1.0;
                    }
        avg[0] = bx / norm_; avg[1] = by / norm_; avg[2] = bt / norm_;
    } else {
        for (size_t h = 0; h < nh; h++)
            if (resp[h] > best) best = resp[h];
        double ax = 0, ay = 0, tx = 0, ty = 0;
        int cnt = 0;
        for (int iy = 0; iy < ny; iy++)
            for (int ix = 0; ix < nx; ix++)
                for (int k = 0; k < nt; k++)
                    if (kt_double_equal(resp[VOL(L, ix, iy, k)], best)) {
                        double hd = PY(c, ORC_D16_EXPANSION_CLAMP) ? L->tabs[k] : kt_normalize_angle(L->tabs[k]);
                        ax += L->xabs[ix]; ay += L->yabs[iy];
                        tx += cos(hd); ty += sin(hd);
                        cnt++;
                    }
        if (cnt > 0) {
            ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
            avg[0] = ax; avg[1] = ay; avg[2] = atan2(ty, tx);
        } else {
            *err = fail("Unable to find best position");
        }
    }

    if (pass == 0) {
        /* D14 positional covariance (always from the coarse pass).
         * Karto: ComputePositionalCovariance over m_pSearchSpaceProbs = max over theta per (x, y): every cell with
         *        response >= best - 0.1, floors of 0.1 step^2, times 1 / best, 500 when degenerate; cov[8] = 4 car^2
         * Python: the 11 x 11 window around the arg-max in its own theta slice, no threshold, / norm / response
         *        (helpers.py:266-282,295) */
        if (PY(c, ORC_D14_POS_COV)) {
            double XX = 0, YY = 0, XY = 0, norm = 0.0;
            int ii = o->ii, jj = o->jj, kk = o->kk;
            int xs = ii - 5 > 0 ? ii - 5 : 0;
            int ys = jj - 5 > 0 ? jj - 5 : 0;
            int xe = nx - 1 < ii + 6 ? nx - 1 : ii + 6;
            int ye = ny - 1 < jj + 6 ? ny - 1 : jj + 6;
            for (int i = xs; i < xe; i++)
                for (int j = ys; j < ye; j++) {
                    double r_ = resp[VOL(L, i, j, kk)];
                    double x_ = L->xabs[i], y_ = L->yabs[j];
                    norm += r_;
                    XX += r_ * ((x_ - avg[0]) * (x_ - avg[0]));
                    YY += r_ * ((y_ - avg[1]) * (y_ - avg[1]));
                    XY += (x_ - avg[0]) * (y_ - avg[1]) * r_;
                }
            memset(cov, 0, sizeof(double) * 9);
            cov[0] = XX / norm / best; cov[4] = YY / norm / best; cov[1] = cov[3] = XY / norm / best;
            cov[8] = 4 * L->angle_res; /* D15, coarse only: scan_matching.py:214 `4 * self.angle_res` */
        } else {
            for (int i = 0; i < c->side * c->side; i++) c->probs[i] = 0.0;
            double pox = center[0] - L->off_x, poy = center[1] - L->off_y;
            for (int iy = 0; iy < ny; iy++)
                for (int ix = 0; ix < nx; ix++) {
                    int gx = conv(c, L->xabs[ix], pox);
                    int gy = conv(c, L->yabs[iy], poy);
                    if (gx < 0 || gx >= c->side || gy < 0 || gy >= c->side) {
                        *err = fail("Index out of range in probability search");
                        continue;
                    }
                    double *p = &c->probs[gy * c->side + gx];
                    for (int k = 0; k < nt; k++)
                        if (resp[VOL(L, ix, iy, k)] > *p) *p = resp[VOL(L, ix, iy, k)];
                }
            memset(cov, 0, sizeof(double) * 9);
            cov[0] = cov[4] = cov[8] = 1.0;
            if (best < KT_TOLERANCE) {
                cov[0] = MAX_VARIANCE; cov[4] = MAX_VARIANCE; cov[8] = 4 * (L->angle_res * L->angle_res);
            } else {
                double axx = 0, axy = 0, ayy = 0, norm = 0;
                double dx = avg[0] - center[0];
                double dy = avg[1] - center[1];
                for (int iy = 0; iy < ny; iy++) {
                    double y = L->yoff[iy];
                    for (int ix = 0; ix < nx; ix++) {
                        double x = L->xoff[ix];
                        int gx = conv(c, center[0] + x, pox);
                        int gy = conv(c, center[1] + y, poy);
                        if (gx < 0 || gx >= c->side || gy < 0 || gy >= c->side) continue; /* reported above */
                        double response = c->probs[gy * c->side + gx];
                        if (response >= (best - 0.1)) {
                            norm += response;
                            axx += ((x - dx) * (x - dx)) * response;
                            axy += ((x - dx) * (y - dy) * response);
                            ayy += ((y - dy) * (y - dy)) * response;
                        }
                    }
                }
                if (norm > KT_TOLERANCE) {
                    double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
                    double vthth = 4 * (L->angle_res * L->angle_res);
                    double min_xx = 0.1 * (L->step_x * L->step_x), min_yy = 0.1 * (L->step_y * L->step_y);
                    if (vxx < min_xx) vxx = min_xx;
                    if (vyy < min_yy) vyy = min_yy;
                    double mult = 1.0 / best;
                    cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
                    cov[8] = vthth;
                }
                if (kt_double_equal(cov[0], 0.0)) cov[0] = MAX_VARIANCE;
                if (kt_double_equal(cov[4], 0.0)) cov[4] = MAX_VARIANCE;
            }
        }
    } else {
        /* D15 angular covariance (fine pass).
         * Karto: ComputeAngularCovariance -- GetResponse at the cell of the mean pose for every fine angle, those
         *        >= best - 0.1, (theta_k - bestAngle)^2 weights with theta_k absolute and bestAngle relative (upstream
         *        quirk), fallbacks  |  Python: the 11 angles around the arg-max at its (x, y) (helpers.py:284-293) */
        if (PY(c, ORC_D15_ANG_COV)) {
            double TH = 0, th_norm = 0.0;
            int ts = o->kk - 5 > 0 ? o->kk - 5 : 0;
            int te = nt - 1 < o->kk + 6 ? nt - 1 : o->kk + 6;
            for (int k = ts; k < te; k++) {
                double r_ = resp[VOL(L, o->ii, o->jj, k)];
                th_norm += r_;
                TH += r_ * ((L->tabs[k] - avg[2]) * (L->tabs[k] - avg[2]));
            }
            cov[8] = TH / th_norm;
        } else {
            double best_angle = kt_normalize_angle_difference(avg[2], center[2]);
            int gx = conv(c, avg[0], c->off_x) + c->roi_x;
            int gy = conv(c, avg[1], c->off_y) + c->roi_y;
            int grid_index = gx + gy * c->pitch;
            /* the lookup table of this pass, rebuilt (it was freed above): same expressions */
            double norm = 0.0, acc = 0.0;
            int32_t *row = (int32_t *)malloc(sizeof(int32_t) * (np > 0 ? np : 1));
            for (int k = 0; k < nt; k++) {
                double angle = L->tabs[k];
                double cosine = cos(angle), sine = sin(angle);
                for (int i = 0; i < np; i++) {
                    double px = c->qlocal[2 * i], py = c->qlocal[2 * i + 1];
                    double ox = cosine * px - sine * py;
                    double oy = sine * px + cosine * py;
                    row[i] = conv(c, ox + c->off_x, c->off_x) + conv(c, oy + c->off_y, c->off_y) * c->pitch;
                }
                double response = 0.0;
                if (np != 0) {
                    response = (double)k_get_sum(c, row, np, grid_index);
                    response /= (double)(np * GRID_OCCUPIED);
                }
                if (response >= (best - 0.1)) {
                    norm += response;
                    acc += ((angle - best_angle) * (angle - best_angle)) * response;
                }
            }
            free(row);
            if (norm > KT_TOLERANCE) {
                if (acc < KT_TOLERANCE) acc = L->angle_res * L->angle_res;
                acc /= norm;
            } else {
                acc = 1000 * (L->angle_res * L->angle_res);
            }
            cov[8] = acc;
        }
    }
    o->mean[0] = avg[0]; o->mean[1] = avg[1]; o->mean[2] = avg[2];
    /* D16: Karto clamps the returned response to 1 */
    if (!PY(c, ORC_D16_EXPANSION_CLAMP) && best > 1.0) best = 1.0;
    o->best = best;
    return 0;
}

/* ScanMatcher::MatchScan  |  scan_matching.py:175-222 Scan2DMatcherPy.match_scan */
static int u_match(orc_ctx *c, const orc_scan *query, const orc_scan *base, int n_base, int penalize, int refine,
                   orc_result *out) {
    const orc_config *g = &c->cfg;
    memset(out, 0, sizeof *out);
    if (u_setup_grid(c)) return -1;
    const double res = c->res;
    double pose[3] = {query->pose[0], query->pose[1], query->pose[2]};

    /* query point readings.  D18: Karto takes the readings at the scan's pose and moves them into the sensor frame with
     * Transform(pose).InverseTransformPose (identity when the pose is exactly 0)  |  Python: points_local() projects at
     * pose (0, 0, 0) directly (models.py:96-97).  D7 gates the ranges. */
    double *qx = (double *)malloc(sizeof(double) * (query->n + 1));
    double *qy = (double *)malloc(sizeof(double) * (query->n + 1));
    const int gate = PY(c, ORC_D7_RANGE_GATE) ? ORC_SEM_YAGPY : ORC_SEM_KARTO;
    int nq;
    free(c->qlocal);
    if (PY(c, ORC_D18_LOOKUP)) {
        orc_scan ql = *query;
        ql.pose[0] = ql.pose[1] = ql.pose[2] = 0.0;
        nq = orc_point_readings(&ql, gate, qx, qy);
        c->qlocal = (double *)malloc(sizeof(double) * 2 * (nq + 1));
        for (int i = 0; i < nq; i++) { c->qlocal[2 * i] = qx[i]; c->qlocal[2 * i + 1] = qy[i]; }
    } else {
        nq = orc_point_readings(query, gate, qx, qy);
        c->qlocal = (double *)malloc(sizeof(double) * 2 * (nq + 1));
        if (pose[0] == 0.0 && pose[1] == 0.0 && pose[2] == 0.0) {
            for (int i = 0; i < nq; i++) { c->qlocal[2 * i] = qx[i]; c->qlocal[2 * i + 1] = qy[i]; }
        } else {
            double cr = cos(0.0 - pose[2]), sr = sin(0.0 - pose[2]);
            /* FromAxisAngle(0,0,1,a): m00 = c, m01 = -s, m10 = s, m11 = c with a = -theta */
            for (int i = 0; i < nq; i++) {
                double dx = qx[i] - pose[0], dy = qy[i] - pose[1];
                c->qlocal[2 * i]     = cr * dx + (0.0 - sr) * dy;
                c->qlocal[2 * i + 1] = sr * dx + cr * dy;
            }
        }
    }
    free(qx); free(qy);
    c->n_qlocal = nq;
    out->n_query_points = nq;
    /* D16: Karto returns early for a scan without readings ("cannot do scan matching"); the Python path has no such test */
    if (nq == 0 && !PY(c, ORC_D16_EXPANSION_CLAMP)) {
        out->pose[0] = pose[0]; out->pose[1] = pose[1]; out->pose[2] = pose[2];
        out->cov[0] = MAX_VARIANCE; out->cov[4] = MAX_VARIANCE;
        out->cov[8] = 4 * (g->coarse_angle_resolution * g->coarse_angle_resolution);
        out->response = 0.0;
        free_pass(c, 0); free_pass(c, 1);
        c->n_raster = 0;
        return 0;
    }

    /* D4 grid origin: the query pose is the centre cell -- the same formula on both sides */
    c->off_x = pose[0] - (0.5 * (c->roi_w - 1) * res);
    c->off_y = pose[1] - (0.5 * (c->roi_h - 1) * res);

    double t_raster = now_s();
    u_add_scans(c, base, n_base, pose[0], pose[1]);
    c->serial_s = now_s() - t_raster;

    /* D9 coarse lattice: +-0.5 (side - 1) res at 2 res (inclusive), +-cao at car  |  np.arange(+-S/2, 2 res), +-cao/2 */
    const int py9 = PY(c, ORC_D9_COARSE_LATTICE);
    double coarse_off = py9 ? g->search_size * 0.5 : 0.5 * (c->side - 1) * res;
    double coarse_step = 2 * res;
    double angle_off = py9 ? g->coarse_search_angle_offset * 0.5 : g->coarse_search_angle_offset;
    int err = 0;
    lattice L;
    pass_result pr;
    lattice_make(&L, py9, pose, coarse_off, coarse_step, angle_off, g->coarse_angle_resolution);
    if (u_correlate(c, 0, pose, &L, penalize, &pr, out->cov, &err)) { lattice_free(&L); return -1; }
    out->hypotheses += (long long)L.nx * L.ny * L.nt;
    lattice_free(&L);
    /* D16 response expansion: up to three retries 20 degrees wider when nothing matched */
    if (!PY(c, ORC_D16_EXPANSION_CLAMP) && g->use_response_expansion && kt_double_equal(pr.best, 0.0)) {
        double new_off = angle_off;
        for (int i = 0; i < 3; i++) {
            new_off += 20.0 * KT_PI / 180.0; /* math::DegreesToRadians(20) */
            lattice_make(&L, py9, pose, coarse_off, coarse_step, new_off, g->coarse_angle_resolution);
            if (u_correlate(c, 0, pose, &L, penalize, &pr, out->cov, &err)) { lattice_free(&L); return -1; }
            out->expansions++;
            out->hypotheses += (long long)L.nx * L.ny * L.nt;
            lattice_free(&L);
            if (!kt_double_equal(pr.best, 0.0)) break;
        }
    }
    memcpy(out->coarse_dims, c->dims[0], sizeof out->coarse_dims);
    free_pass(c, 1);
    if (refine) {
        /* D10 fine lattice: +-res at res, +-car/2 at the fine resolution  |  +-2 res at res, +-0.01745 at 0.00349 (constants) */
        const int py10 = PY(c, ORC_D10_FINE_LATTICE);
        double center[3] = {pr.mean[0], pr.mean[1], pr.mean[2]};
        if (py10)
            lattice_make(&L, 1, center, res * 2, res, 0.0349 * 0.5, 0.00349);
        else
            lattice_make(&L, 0, center, coarse_step * 0.5, res, 0.5 * g->coarse_angle_resolution, g->fine_search_angle_resolution);
        if (u_correlate(c, 1, center, &L, penalize, &pr, out->cov, &err)) { lattice_free(&L); return -1; }
        memcpy(out->fine_dims, c->dims[1], sizeof out->fine_dims);
        out->hypotheses += (long long)L.nx * L.ny * L.nt;
        lattice_free(&L);
    }
    out->response = pr.best;
    out->pose[0] = pr.mean[0]; out->pose[1] = pr.mean[1]; out->pose[2] = pr.mean[2];
    return err ? -1 : 0;
}

/* ================================================================== occupancy-grid rendering
 * karto_scanmatcher.create_occupancy_grid(scans, resolution, range_threshold) -- /root/reference/yag_slam/graph_slam.py:341-342,
 * image codes as /root/reference/ros1/slam_node_ros1:199-202 reads them.  Restates open_karto OccupancyGrid::
 * {CreateFromScans, ComputeDimensions, AddScan, RayTrace, Update, UpdateCell}, Grid<T>::TraceLine and the bounding box of
 * LocalizedRangeScan::Update, sequentially, scan by scan and beam by beam as Karto does.  PARITY UNPINNED. */
static int occ_valid(int x, int y, int w, int h) { return x >= 0 && x < w && y >= 0 && y < h; }

int orc_occupancy_grid(const orc_scan *scans, const double *max_ranges, int n_scans, double resolution, double range_threshold,
                       int *width, int *height, double *off_x, double *off_y, uint8_t *image, long long image_bytes) {
    if (!scans || n_scans <= 0 || !width || !height) return fail("bad arguments");
    /* LocalizedRangeScan::Update: bounding box = sensor position + readings within [min_range, range_threshold] */
    double x0 = 1e300, y0 = 1e300, x1 = -1e300, y1 = -1e300;
    for (int s = 0; s < n_scans; s++) {
        const orc_scan *sc = &scans[s];
        double bx0 = sc->pose[0], by0 = sc->pose[1], bx1 = sc->pose[0], by1 = sc->pose[1];
        for (int i = 0; i < sc->n; i++) {
            double r = sc->ranges[i];
            if (!(r >= sc->min_range && r <= range_threshold)) continue;
            double angle = sc->pose[2] + sc->min_angle + i * sc->angle_increment;
            double px = sc->pose[0] + r * cos(angle), py = sc->pose[1] + r * sin(angle);
            if (px < bx0) bx0 = px;
            if (px > bx1) bx1 = px;
            if (py < by0) by0 = py;
            if (py > by1) by1 = py;
        }
        if (bx0 < x0) x0 = bx0;
        if (by0 < y0) y0 = by0;
        if (bx1 > x1) x1 = bx1;
        if (by1 > y1) y1 = by1;
    }
    /* OccupancyGrid::ComputeDimensions */
    double scale = 1.0 / resolution;
    int w = (int)kt_round((x1 - x0) * scale), h = (int)kt_round((y1 - y0) * scale);
    *width = w; *height = h;
    if (off_x) *off_x = x0;
    if (off_y) *off_y = y0;
    if (!image) return 0;
    if (w <= 0 || h <= 0 || image_bytes < (long long)w * h) return fail("image buffer too small");
    uint32_t *pass = (uint32_t *)calloc((size_t)w * h, sizeof(uint32_t));
    uint32_t *hits = (uint32_t *)calloc((size_t)w * h, sizeof(uint32_t));
    for (int s = 0; s < n_scans; s++) {
        const orc_scan *sc = &scans[s];
        for (int i = 0; i < sc->n; i++) {
            /* OccupancyGrid::AddScan */
            double r = sc->ranges[i];
            int end_valid = r < (range_threshold - KT_TOLERANCE);
            if (r <= sc->min_range || r >= max_ranges[s] || isnan(r)) continue;
            double angle = sc->pose[2] + sc->min_angle + i * sc->angle_increment;
            double px = sc->pose[0] + r * cos(angle), py = sc->pose[1] + r * sin(angle);
            if (r >= range_threshold) {
                double ratio = range_threshold / r;
                double dx = px - sc->pose[0], dy = py - sc->pose[1];
                px = sc->pose[0] + ratio * dx;
                py = sc->pose[1] + ratio * dy;
            }
            /* RayTrace + Grid<T>::TraceLine */
            int gx0 = (int)kt_round((sc->pose[0] - x0) * scale), gy0 = (int)kt_round((sc->pose[1] - y0) * scale);
            int gx1 = (int)kt_round((px - x0) * scale), gy1 = (int)kt_round((py - y0) * scale);
            int tx = gx1, ty = gy1;
            int steep = abs(gy1 - gy0) > abs(gx1 - gx0);
            if (steep) { int t = gx0; gx0 = gy0; gy0 = t; t = gx1; gx1 = gy1; gy1 = t; }
            if (gx0 > gx1) { int t = gx0; gx0 = gx1; gx1 = t; t = gy0; gy0 = gy1; gy1 = t; }
            int delta_x = gx1 - gx0, delta_y = abs(gy1 - gy0), error = 0, y = gy0;
            int ystep = gy0 < gy1 ? 1 : -1;
            for (int x = gx0; x <= gx1; x++) {
                int cx = steep ? y : x, cy = steep ? x : y;
                error += delta_y;
                if (2 * error >= delta_x) { y += ystep; error -= delta_x; }
                if (occ_valid(cx, cy, w, h)) pass[(size_t)cy * w + cx]++;
            }
            if (end_valid && occ_valid(tx, ty, w, h)) { pass[(size_t)ty * w + tx]++; hits[(size_t)ty * w + tx]++; }
        }
    }
    /* OccupancyGrid::Update / UpdateCell: MinPassThrough 2, OccupancyThreshold 0.1; codes 0 occupied, 200 unknown, 255 free */
    for (size_t c = 0; c < (size_t)w * h; c++) {
        uint8_t v = 200;
        if (pass[c] > 2u) {
            double ratio = (double)hits[c] / (double)pass[c];
            v = ratio > 0.1 ? 0 : 255;
        }
        image[c] = v;
    }
    free(pass); free(hits);
    return 0;
}

/* ================================================================== public entry + accessors */
int orc_match(orc_ctx *c, const orc_scan *query, const orc_scan *base, int n_base, int penalize,
              int refine, orc_result *out) {
    if (!c || !query || !out || (n_base > 0 && !base)) return fail("null argument");
    g_err[0] = 0;
    return u_match(c, query, base, n_base, penalize, refine, out);
}

const uint8_t *orc_grid_u8(const orc_ctx *c, int *width, int *height, int *pitch, int *roi_x,
                           int *roi_y, int *roi_w, int *roi_h) {
    if (width) *width = c->gw;
    if (height) *height = c->gh;
    if (pitch) *pitch = c->pitch;
    if (roi_x) *roi_x = c->roi_x;
    if (roi_y) *roi_y = c->roi_y;
    if (roi_w) *roi_w = c->roi_w;
    if (roi_h) *roi_h = c->roi_h;
    return c->grid;
}
const double *orc_grid_f64(const orc_ctx *c, int *size) {
    if (size) *size = c->gridf ? c->gw : 0;
    return c->gridf;
}
void orc_grid_offset(const orc_ctx *c, double *ox, double *oy) {
    if (ox) *ox = c->off_x;
    if (oy) *oy = c->off_y;
}
const uint32_t *orc_sums(const orc_ctx *c, int pass, int *nx, int *ny, int *nt) {
    if (nx) *nx = c->dims[pass][0];
    if (ny) *ny = c->dims[pass][1];
    if (nt) *nt = c->dims[pass][2];
    return c->sums[pass];
}
const double *orc_responses(const orc_ctx *c, int pass, int *nx, int *ny, int *nt) {
    if (nx) *nx = c->dims[pass][0];
    if (ny) *ny = c->dims[pass][1];
    if (nt) *nt = c->dims[pass][2];
    return c->resp[pass];
}
const double *orc_raster_points(const orc_ctx *c, int *n) {
    if (n) *n = c->n_raster;
    return c->raster_pts;
}
double orc_last_serial_seconds(const orc_ctx *c) { return c->serial_s; }
const double *orc_query_local(const orc_ctx *c, int *n) {
    if (n) *n = c->n_qlocal;
    return c->qlocal;
}
