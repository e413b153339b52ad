/*
 * ym_oracle.c -- CPU restatement of the reference's correlative scan matcher.
 *
 * TEST INFRASTRUCTURE ONLY (see ym_oracle.h).  Plain C99, fp64, compiled with
 * -ffp-contract=off so every expression rounds exactly as written.
 *
 * "yagpy" functions cite /root/reference/yag_slam/{helpers,scan_matching}.py line ranges.
 * "karto" functions cite the open_karto function they restate (source not in /root/reference;
 * reached by the reference through karto_scanmatcher==1.0.0, /root/reference/setup.py:46).
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
    double scale;        /* karto: 1/resolution */
    /* karto kernel */
    uint8_t *kernel;
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
    if (cfg->semantics != ORC_SEM_KARTO && cfg->semantics != ORC_SEM_YAGPY) {
        fail("unknown semantics"); return NULL;
    }
    orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
    c->cfg = *cfg;
    if (c->cfg.threads < 1) c->cfg.threads = 1;
    return c;
}

void orc_destroy(orc_ctx *c) {
    if (!c) return;
    free(c->grid); free(c->gridf); free(c->kernel); free(c->probs);
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

/* ================================================================== KARTO semantics */

/* CoordinateConverter::WorldToGrid (one axis): Round((w - offset) * scale) */
static int k_world_to_grid(double w, double off, double scale) {
    return (int)kt_round((w - off) * scale);
}

/* ScanMatcher::Create: grid geometry */
static int k_setup_grid(orc_ctx *c) {
    const orc_config *g = &c->cfg;
    c->scale = 1.0 / g->resolution;
    int side = (int)(kt_round(g->search_size / g->resolution) + 1);
    int margin = (int)ceil(g->range_threshold / g->resolution);
    int gsize = side + 2 * margin;
    int ksz = orc_kernel_karto(g->resolution, g->smear_deviation, NULL);
    int half = ksz / 2;
    int border = half + 1;
    int W = gsize + 2 * border;
    if (!c->kernel || c->ksize != ksz) {
        free(c->kernel);
        c->kernel = (uint8_t *)malloc((size_t)ksz * ksz);
        c->ksize = ksz;
        orc_kernel_karto(g->resolution, g->smear_deviation, c->kernel);
    }
    if (!c->grid || c->gw != W) {
        free(c->grid);
        c->gw = c->gh = W;
        c->pitch = align8(W);
        c->grid = (uint8_t *)malloc((size_t)c->pitch * c->gh);
        if (!c->grid) return fail("grid alloc failed");
    }
    c->roi_x = c->roi_y = border;
    c->roi_w = c->roi_h = gsize;
    if (!c->probs || c->side != side) {
        free(c->probs);
        c->side = side;
        c->probs = (double *)malloc(sizeof(double) * side * side);
    }
    return 0;
}

/* CorrelationGrid::SmearPoint */
static void k_smear_point(orc_ctx *c, int gx, int gy) {
    int half = c->ksize / 2;
    for (int j = -half; j <= half; j++) {
        uint8_t *row = c->grid + (size_t)(gy + j + c->roi_y) * c->pitch + (gx + c->roi_x);
        const uint8_t *krow = c->kernel + c->ksize * (j + half) + half;
        for (int i = -half; i <= half; i++)
            if (krow[i] > row[i]) row[i] = krow[i];
    }
}

/* ScanMatcher::AddScans / AddScan */
static int k_add_scans(orc_ctx *c, const orc_scan *base, int n_base, double vpx, double vpy) {
    memset(c->grid, 0, (size_t)c->pitch * c->gh);
    int maxn = 0, total = 0;
    for (int b = 0; b < n_base; b++) { if (base[b].n > maxn) maxn = base[b].n; total += base[b].n; }
    double *xs = (double *)malloc(sizeof(double) * (maxn + 1));
    double *ys = (double *)malloc(sizeof(double) * (maxn + 1));
    uint8_t *keep = (uint8_t *)malloc((size_t)maxn + 1);
    free(c->raster_pts);
    c->raster_pts = (double *)malloc(sizeof(double) * 2 * (total + 1));
    c->n_raster = 0;
    for (int b = 0; b < n_base; b++) {
        int n = orc_point_readings(&base[b], ORC_SEM_KARTO, xs, ys);
        orc_valid_points(xs, ys, n, vpx, vpy, ORC_SEM_KARTO, keep);
        for (int i = 0; i < n; i++) {
            if (!keep[i]) continue;
            c->raster_pts[2 * c->n_raster] = xs[i];
            c->raster_pts[2 * c->n_raster + 1] = ys[i];
            c->n_raster++;
            int gx = k_world_to_grid(xs[i], c->off_x, c->scale);
            int gy = k_world_to_grid(ys[i], c->off_y, c->scale);
            if (gx < 0 || gx >= c->roi_w || gy < 0 || gy >= c->roi_h) continue; /* not in grid */
            uint8_t *cell = c->grid + (size_t)(gy + c->roi_y) * c->pitch + (gx + c->roi_x);
            if (*cell == GRID_OCCUPIED) continue; /* value already set */
            *cell = GRID_OCCUPIED;
            k_smear_point(c, gx, gy);
        }
    }
    free(xs); free(ys); free(keep);
    return 0;
}

/* GridIndexLookup::ComputeOffsets -> lookup[k][i] linear offsets; angles[k] */
static int k_compute_offsets(orc_ctx *c, double center_theta, double angle_off, double angle_res,
                             int32_t **lookup_out, int *n_angles_out) {
    int na = (int)(kt_round(angle_off * 2.0 / angle_res) + 1);
    int np = c->n_qlocal;
    int32_t *lk = (int32_t *)malloc(sizeof(int32_t) * (size_t)na * (np > 0 ? np : 1));
    double start = center_theta - angle_off;
    for (int k = 0; k < na; k++) {
        double angle = start + k * angle_res;
        double cosine = cos(angle), sine = sin(angle);
        for (int i = 0; i < np; i++) {
            double px = c->qlocal[2 * i], py = c->qlocal[2 * i + 1];
            double ox = cosine * px - sine * py;
            double oy = sine * px + cosine * py;
            /* WorldToGrid(offset + gridOffset) */
            int gx = k_world_to_grid(ox + c->off_x, c->off_x, c->scale);
            int gy = k_world_to_grid(oy + c->off_y, c->off_y, c->scale);
            lk[(size_t)k * np + i] = gx + gy * c->pitch; /* base GridIndex, ROI ignored */
        }
    }
    *lookup_out = lk;
    *n_angles_out = na;
    return 0;
}

/* ScanMatcher::GetResponse -> integer sum (division done by caller exactly as karto) */
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

static double k_response_from_sum(uint32_t sum, int np) {
    if (np == 0) return 0.0;
    double response = (double)sum;
    response /= (double)(np * GRID_OCCUPIED);
    return response;
}

/* ScanMatcher::ComputePositionalCovariance */
static void k_positional_cov(orc_ctx *c, const double best_pose[3], double best, const double center[3],
                             double off_x, double off_y, double step_x, double step_y,
                             double angle_res, double cov[9]) {
    memset(cov, 0, sizeof(double) * 9);
    cov[0] = cov[4] = cov[8] = 1.0;
    if (best < KT_TOLERANCE) {
        cov[0] = MAX_VARIANCE; cov[4] = MAX_VARIANCE; cov[8] = 4 * (angle_res * angle_res);
        return;
    }
    double axx = 0, axy = 0, ayy = 0, norm = 0;
    double dx = best_pose[0] - center[0];
    double dy = best_pose[1] - center[1];
    int nx = (int)(kt_round(off_x * 2.0 / step_x) + 1);
    int ny = (int)(kt_round(off_y * 2.0 / step_y) + 1);
    double start_x = -off_x, start_y = -off_y;
    double pox = center[0] - off_x, poy = center[1] - off_y; /* probs grid offset */
    for (int iy = 0; iy < ny; iy++) {
        double y = start_y + iy * step_y;
        for (int ix = 0; ix < nx; ix++) {
            double x = start_x + ix * step_x;
            int gx = k_world_to_grid(center[0] + x, pox, c->scale);
            int gy = k_world_to_grid(center[1] + y, poy, c->scale);
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
        double vthth = 4 * (angle_res * angle_res);
        double min_xx = 0.1 * (step_x * step_x), min_yy = 0.1 * (step_y * step_y);
        if (vxx < min_xx) vxx = min_xx;
        if (vyy < min_yy) vyy = min_yy;
        double mult = 1.0 / best;
        cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
        cov[8] = vthth;
    }
    if (kt_double_equal(cov[0], 0.0)) cov[0] = MAX_VARIANCE;
    if (kt_double_equal(cov[4], 0.0)) cov[4] = MAX_VARIANCE;
}

/* ScanMatcher::CorrelateScan */
static double k_correlate(orc_ctx *c, const double center[3], double off_x, double off_y,
                          double step_x, double step_y, double angle_off, double angle_res,
                          int penalize, double mean[3], double cov[9], int fine, int *err) {
    const orc_config *g = &c->cfg;
    int pass = fine ? 1 : 0;
    int np = c->n_qlocal;
    int32_t *lk = NULL;
    int na = 0;
    k_compute_offsets(c, center[2], angle_off, angle_res, &lk, &na);

    double pox = 0, poy = 0;
    if (!fine) {
        for (int i = 0; i < c->side * c->side; i++) c->probs[i] = 0.0;
        pox = center[0] - off_x;
        poy = center[1] - off_y;
    }
    int nx = (int)(kt_round(off_x * 2.0 / step_x) + 1);
    int ny = (int)(kt_round(off_y * 2.0 / step_y) + 1);
    double start_x = -off_x, start_y = -off_y;

    free_pass(c, pass);
    size_t nh = (size_t)nx * ny * na;
    c->sums[pass] = (uint32_t *)malloc(sizeof(uint32_t) * (nh ? nh : 1));
    c->resp[pass] = (double *)malloc(sizeof(double) * (nh ? nh : 1));
    c->dims[pass][0] = nx; c->dims[pass][1] = ny; c->dims[pass][2] = na;
    double *hx = (double *)malloc(sizeof(double) * (nh ? nh : 1));
    double *hy = (double *)malloc(sizeof(double) * (nh ? nh : 1));
    double *ht = (double *)malloc(sizeof(double) * (nh ? nh : 1));
    uint32_t *sums = c->sums[pass];
    double *resp = c->resp[pass];
    double start_angle = center[2] - angle_off;
    int data_size = c->pitch * c->gh;
    int bad_index = 0;

#ifdef _OPENMP
#pragma omp parallel for collapse(2) num_threads(g->threads) schedule(static) if (g->threads > 1)
#endif
    for (int iy = 0; iy < ny; iy++) {
        for (int ix = 0; ix < nx; ix++) {
            double y = start_y + iy * step_y;
            double new_y = center[1] + y;
            double sq_y = y * y;
            double x = start_x + ix * step_x;
            double new_x = center[0] + x;
            double sq_x = x * x;
            int gx = k_world_to_grid(new_x, c->off_x, c->scale) + c->roi_x;
            int gy = k_world_to_grid(new_y, c->off_y, c->scale) + c->roi_y;
            int grid_index = gx + gy * c->pitch;
            if (gx < 0 || gx >= c->gw || gy < 0 || gy >= c->gh || grid_index < 0 ||
                grid_index >= data_size) { bad_index = 1; grid_index = 0; }
            for (int k = 0; k < na; k++) {
                double angle = start_angle + k * angle_res;
                size_t h = ((size_t)iy * nx + ix) * na + k;
                uint32_t s = k_get_sum(c, lk + (size_t)k * np, np, grid_index);
                double response = k_response_from_sum(s, np);
                if (penalize && !kt_double_equal(response, 0.0)) {
                    double sq_dist = sq_x + sq_y;
                    double dp = 1.0 - (DISTANCE_PENALTY_GAIN * sq_dist / g->distance_variance_penalty);
                    if (dp < g->minimum_distance_penalty) dp = g->minimum_distance_penalty;
                    double sq_ang = (angle - center[2]) * (angle - center[2]);
                    double ap = 1.0 - (ANGLE_PENALTY_GAIN * sq_ang / g->angle_variance_penalty);
                    if (ap < g->minimum_angle_penalty) ap = g->minimum_angle_penalty;
                    response *= (dp * ap);
                }
                sums[h] = s;
                resp[h] = response;
                hx[h] = new_x; hy[h] = new_y; ht[h] = kt_normalize_angle(angle);
            }
        }
    }
    if (bad_index) { *err = fail("hypothesis grid index out of range"); }

    double best = -1;
    for (size_t h = 0; h < nh; h++) {
        if (resp[h] > best) best = resp[h];
        if (!fine) {
            int gx = k_world_to_grid(hx[h], pox, c->scale);
            int gy = k_world_to_grid(hy[h], poy, c->scale);
            if (gx < 0 || gx >= c->side || gy < 0 || gy >= c->side) {
                *err = fail("Index out of range in probability search");
                continue;
            }
            double *p = &c->probs[gy * c->side + gx];
            if (resp[h] > *p) *p = resp[h];
        }
    }
    double ax = 0, ay = 0, tx = 0, ty = 0;
    int cnt = 0;
    for (size_t h = 0; h < nh; h++) {
        if (kt_double_equal(resp[h], best)) {
            ax += hx[h]; ay += hy[h];
            tx += cos(ht[h]); ty += sin(ht[h]);
            cnt++;
        }
    }
    double avg[3] = {0, 0, 0};
    if (cnt > 0) {
        ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
        avg[0] = ax; avg[1] = ay; avg[2] = atan2(ty, tx);
    } else {
        *err = fail("Unable to find best position");
    }
    free(hx); free(hy); free(ht);

    if (!fine) {
        k_positional_cov(c, avg, best, center, off_x, off_y, step_x, step_y, angle_res, cov);
    } else {
        /* ScanMatcher::ComputeAngularCovariance (does not reset the matrix) */
        double best_angle = kt_normalize_angle_difference(avg[2], center[2]);
        int gx = k_world_to_grid(avg[0], c->off_x, c->scale) + c->roi_x;
        int gy = k_world_to_grid(avg[1], c->off_y, c->scale) + c->roi_y;
        int grid_index = gx + gy * c->pitch;
        double norm = 0.0, acc = 0.0;
        for (int k = 0; k < na; k++) {
            double angle = start_angle + k * angle_res;
            double response =
                k_response_from_sum(k_get_sum(c, lk + (size_t)k * np, np, grid_index), np);
            if (response >= (best - 0.1)) {
                norm += response;
                acc += ((angle - best_angle) * (angle - best_angle)) * response;
            }
        }
        if (norm > KT_TOLERANCE) {
            if (acc < KT_TOLERANCE) acc = angle_res * angle_res;
            acc /= norm;
        } else {
            acc = 1000 * (angle_res * angle_res);
        }
        cov[8] = acc;
    }
    mean[0] = avg[0]; mean[1] = avg[1]; mean[2] = avg[2];
    free(lk);
    if (best > 1.0) best = 1.0;
    return best;
}

/* ScanMatcher::MatchScan */
static int k_match(orc_ctx *c, const orc_scan *query, const orc_scan *base, int n_base,
                   int penalize, int refine, orc_result *out) {
    const orc_config *g = &c->cfg;
    memset(out, 0, sizeof *out);
    if (k_setup_grid(c)) return -1;
    double res = 1.0 / c->scale;

    /* query point readings (at its current pose) and their sensor-frame coordinates */
    double *qx = (double *)malloc(sizeof(double) * (query->n + 1));
    double *qy = (double *)malloc(sizeof(double) * (query->n + 1));
    int nq = orc_point_readings(query, ORC_SEM_KARTO, qx, qy);
    out->n_query_points = nq;
    double pose[3] = {query->pose[0], query->pose[1], query->pose[2]};
    if (nq == 0) {
        /* scan has no readings; cannot do scan matching */
        out->pose[0] = pose[0]; out->pose[1] = pose[1]; out->pose[2] = pose[2];
        out->cov[0] = MAX_VARIANCE; out->cov[4] = MAX_VARIANCE;
        out->cov[8] = 4 * (g->coarse_angle_resolution * g->coarse_angle_resolution);
        out->response = 0.0;
        free(qx); free(qy);
        free_pass(c, 0); free_pass(c, 1);
        c->n_qlocal = 0; c->n_raster = 0;
        return 0;
    }
    /* Transform(pose).InverseTransformPose: R(-theta) * (p - t); identity when pose == 0 */
    free(c->qlocal);
    c->qlocal = (double *)malloc(sizeof(double) * 2 * nq);
    c->n_qlocal = nq;
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
    free(qx); free(qy);

    /* grid offset so that the query pose is the ROI centre */
    c->off_x = pose[0] - (0.5 * (c->roi_w - 1) * res);
    c->off_y = pose[1] - (0.5 * (c->roi_h - 1) * res);

    double t_raster = now_s();
    k_add_scans(c, base, n_base, pose[0], pose[1]);
    c->serial_s = now_s() - t_raster;

    double coarse_off = 0.5 * (c->side - 1) * res;
    double coarse_step = 2 * res;
    int err = 0;
    double mean[3];
    double best = k_correlate(c, pose, coarse_off, coarse_off, coarse_step, coarse_step,
                              g->coarse_search_angle_offset, g->coarse_angle_resolution,
                              penalize, mean, out->cov, 0, &err);
    out->hypotheses += (long long)c->dims[0][0] * c->dims[0][1] * c->dims[0][2];
    if (g->use_response_expansion && kt_double_equal(best, 0.0)) {
        double new_off = g->coarse_search_angle_offset;
        for (int i = 0; i < 3; i++) {
            new_off += 20.0 * KT_PI / 180.0; /* math::DegreesToRadians(20) */
            best = k_correlate(c, pose, coarse_off, coarse_off, coarse_step, coarse_step, new_off,
                               g->coarse_angle_resolution, penalize, mean, out->cov, 0, &err);
            out->expansions++;
            out->hypotheses += (long long)c->dims[0][0] * c->dims[0][1] * c->dims[0][2];
            if (!kt_double_equal(best, 0.0)) break;
        }
    }
    memcpy(out->coarse_dims, c->dims[0], sizeof out->coarse_dims);
    free_pass(c, 1);
    if (refine) {
        double fine_off = coarse_step * 0.5;
        double center[3] = {mean[0], mean[1], mean[2]};
        best = k_correlate(c, center, fine_off, fine_off, res, res,
                           0.5 * g->coarse_angle_resolution, g->fine_search_angle_resolution,
                           penalize, mean, out->cov, 1, &err);
        memcpy(out->fine_dims, c->dims[1], sizeof out->fine_dims);
        out->hypotheses += (long long)c->dims[1][0] * c->dims[1][1] * c->dims[1][2];
    }
    out->response = best;
    out->pose[0] = mean[0]; out->pose[1] = mean[1]; out->pose[2] = mean[2];
    return err ? -1 : 0;
}

/* ================================================================== YAGPY semantics */

/* helpers.py:105-131 add_scan_to_grid + smear_point on the f64 grid (bounds-checked taps) */
static void y_add_point(double *grid, int G, int gx, int gy, const double *kernel, int ksz) {
    if (!(0 <= gx && gx < G && 0 <= gy && gy < G)) return;
    grid[(size_t)gy * G + gx] = 1.0;
    int half = ksz / 2;
    for (int sx = 0; sx < ksz; sx++)
        for (int sy = 0; sy < ksz; sy++) {
            int x = gx + (sx - half), y = gy + (sy - half);
            if (0 <= x && x < G && 0 <= y && y < G) {
                double cand = kernel[sy * ksz + sx];
                if (cand > grid[(size_t)y * G + x]) grid[(size_t)y * G + x] = cand;
            }
        }
}

/* helpers.py:134-153 score_world_points_on_grid: per point rint((p-o)/res), bounds check,
 * int(100*cell) accumulate.  Uses the u8 image trunc(100*v), which is exact because
 * int(100*max(..)) == max(int(100*..)). */
static uint32_t y_score(const uint8_t *g8, int G, const double *px, const double *py, int n,
                        double shift_x, double shift_y, double ox, double oy, double res) {
    uint32_t sum = 0;
    for (int l = 0; l < n; l++) {
        double x = shift_x + px[l];
        double y = shift_y + py[l];
        double gx = rint((x - ox) / res);
        double gy = rint((y - oy) / res);
        int _x = (int)gx, _y = (int)gy;
        if (_x >= 0 && _x < G && _y >= 0 && _y < G) sum += g8[(size_t)_y * G + _x];
    }
    return sum;
}

typedef struct { double response, x, y, t, xx, yy, xy, th; } y_best;

/* helpers.py:156-295 find_best_pose */
static int y_find_best_pose(orc_ctx *c, int pass, const uint8_t *g8, int G, const double *lx,
                            const double *ly, int np, double cx, double cy, double ct, double ox,
                            double oy, double xy_search, double xy_res, double ang_search,
                            double ang_res, double grid_res, int penalize, y_best *o) {
    double sx_ = ox + G * grid_res / 2;
    double sy_ = oy + G * grid_res / 2;
    int nx = orc_arange(-xy_search + cx, xy_search + cx, xy_res, NULL, 0);
    int ny = orc_arange(-xy_search + cy, xy_search + cy, xy_res, NULL, 0);
    int nt = orc_arange(-ang_search + ct, ang_search + ct, ang_res, NULL, 0);
    double *xv = (double *)malloc(sizeof(double) * (nx + 1));
    double *yv = (double *)malloc(sizeof(double) * (ny + 1));
    double *tv = (double *)malloc(sizeof(double) * (nt + 1));
    orc_arange(-xy_search + cx, xy_search + cx, xy_res, xv, nx);
    orc_arange(-xy_search + cy, xy_search + cy, xy_res, yv, ny);
    orc_arange(-ang_search + ct, ang_search + ct, ang_res, tv, nt);
    const double dist_var_penalty = 0.5, ang_var_penalty = 1.0;

    free_pass(c, pass);
    size_t nh = (size_t)nx * ny * nt;
    if (nh == 0) { free(xv); free(yv); free(tv); return fail("empty search lattice"); }
    c->sums[pass] = (uint32_t *)malloc(sizeof(uint32_t) * nh);
    c->resp[pass] = (double *)malloc(sizeof(double) * nh);
    c->dims[pass][0] = nx; c->dims[pass][1] = ny; c->dims[pass][2] = nt;
    uint32_t *sums = c->sums[pass];
    double *out = c->resp[pass];

#ifdef _OPENMP
#pragma omp parallel for num_threads(c->cfg.threads) schedule(static) if (c->cfg.threads > 1)
#endif
    for (int k = 0; k < nt; k++) {
        double *rx = (double *)malloc(sizeof(double) * (np + 1));
        double *ry = (double *)malloc(sizeof(double) * (np + 1));
        double ca = cos(tv[k]), sa = sin(tv[k]);
        for (int l = 0; l < np; l++) { /* helpers.py:76-78 _rotate_points */
            rx[l] = lx[l] * ca - ly[l] * sa;
            ry[l] = ly[l] * ca + lx[l] * sa;
        }
        for (int i = 0; i < nx; i++)
            for (int j = 0; j < ny; j++) {
                uint32_t s = y_score(g8, G, rx, ry, np, xv[i], yv[j], ox, oy, grid_res);
                double res = (double)s;
                double penalty_val = 1.0;
                if (penalize) {
                    double sd = (xv[i] - sx_) * (xv[i] - sx_) + (yv[j] - sy_) * (yv[j] - sy_);
                    double dist_penalty = 1.0 - 0.2 * sd / (dist_var_penalty * grid_res);
                    double sa2 = (tv[k] - ct) * (tv[k] - ct);
                    double ang_penalty = 1.0 - 0.2 * sa2 / (ang_var_penalty * grid_res);
                    penalty_val = dist_penalty * ang_penalty;
                }
                size_t h = ((size_t)i * ny + j) * nt + k;
                sums[h] = s;
                out[h] = res / np * penalty_val / 100.0;
            }
        free(rx); free(ry);
    }
    /* np.argmax: first maximum in C order; NaN is treated as maximal by numpy */
    size_t m = 0;
    for (size_t h = 1; h < nh; h++) {
        if (isnan(out[m])) break;
        if (out[h] > out[m] || isnan(out[h])) m = h;
    }
    int ii = (int)(m / ((size_t)ny * nt));
    int jj = (int)((m % ((size_t)ny * nt)) / nt);
    int kk = (int)((m % ((size_t)ny * nt)) % nt);
    double response = out[m];
    double bx = 0, by = 0, bt = 0, norm_ = 0.0;
    for (int i = 0; i < nx; i++)
        for (int j = 0; j < ny; j++)
            for (int k = 0; k < nt; k++)
                if (out[((size_t)i * ny + j) * nt + k] >= response - 0.00000001) {
                    bx += xv[i]; by += yv[j]; bt += tv[k]; norm_ += 1.0;
                }
    bx /= norm_; by /= norm_; bt /= norm_;

    double XX = 0, YY = 0, XY = 0, TH = 0, norm = 0.0;
    int xs = ii - 5 > 0 ? ii - 5 : 0;
    int ys = jj - 5 > 0 ? jj - 5 : 0;
    int xe = nx - 1 < ii + 6 ? nx - 1 : ii + 6;
    int ye = ny - 1 < jj + 6 ? ny - 1 : jj + 6;
    for (int i = xs; i < xe; i++)
        for (int j = ys; j < ye; j++) {
            double r_ = out[((size_t)i * ny + j) * nt + kk];
            double x_ = xv[i], y_ = yv[j];
            norm += r_;
            XX += r_ * ((x_ - bx) * (x_ - bx));
            YY += r_ * ((y_ - by) * (y_ - by));
            XY += (x_ - bx) * (y_ - by) * r_;
        }
    double th_norm = 0.0;
    int ts = kk - 5 > 0 ? kk - 5 : 0;
    int te = nt - 1 < kk + 6 ? nt - 1 : kk + 6;
    for (int k = ts; k < te; k++) {
        double r_ = out[((size_t)ii * ny + jj) * nt + k];
        th_norm += r_;
        TH += r_ * ((tv[k] - bt) * (tv[k] - bt));
    }
    o->response = response; o->x = bx; o->y = by; o->t = bt;
    o->xx = XX / norm / response; o->yy = YY / norm / response; o->xy = XY / norm / response;
    o->th = TH / th_norm;
    free(xv); free(yv); free(tv);
    return 0;
}

/* scan_matching.py:175-222 Scan2DMatcherPy.match_scan */
static int y_match(orc_ctx *c, const orc_scan *query, const orc_scan *base, int n_base,
                   int penalize, int refine, orc_result *out) {
    const orc_config *g = &c->cfg;
    memset(out, 0, sizeof *out);
    double res = g->resolution;
    int G = (int)(g->search_size / res + 1 + 2 * g->range_threshold / res);
    if (G <= 0) return fail("bad grid size");
    if (!c->gridf || c->gw != G) {
        free(c->gridf); free(c->grid);
        c->gridf = (double *)malloc(sizeof(double) * (size_t)G * G);
        c->grid = (uint8_t *)malloc((size_t)G * G);
        if (!c->gridf || !c->grid) return fail("grid alloc failed");
    }
    c->gw = c->gh = c->pitch = G;
    c->roi_x = c->roi_y = 0; c->roi_w = c->roi_h = G;
    memset(c->gridf, 0, sizeof(double) * (size_t)G * G);
    double ox = query->pose[0] - 0.5 * (G - 1) * res;
    double oy = query->pose[1] - 0.5 * (G - 1) * res;
    c->off_x = ox; c->off_y = oy;

    int ksz = orc_kernel_yagpy(res, g->smear_deviation, NULL);
    double *kernel = (double *)malloc(sizeof(double) * ksz * ksz);
    orc_kernel_yagpy(res, g->smear_deviation, kernel);

    int maxn = query->n, total = 0;
    for (int b = 0; b < n_base; b++) { if (base[b].n > maxn) maxn = base[b].n; total += base[b].n; }
    double *xs = (double *)malloc(sizeof(double) * (maxn + 1));
    double *ys = (double *)malloc(sizeof(double) * (maxn + 1));
    uint8_t *keep = (uint8_t *)malloc((size_t)maxn + 1);
    free(c->raster_pts);
    c->raster_pts = (double *)malloc(sizeof(double) * 2 * (total + 1));
    c->n_raster = 0;
    for (int b = 0; b < n_base; b++) {
        int n = orc_point_readings(&base[b], ORC_SEM_YAGPY, xs, ys);
        orc_valid_points(xs, ys, n, query->pose[0], query->pose[1], ORC_SEM_YAGPY, keep);
        for (int i = 0; i < n; i++) {
            if (!keep[i]) continue;
            c->raster_pts[2 * c->n_raster] = xs[i];
            c->raster_pts[2 * c->n_raster + 1] = ys[i];
            c->n_raster++;
            /* helpers.py:81-83 world_to_grid (np.round = half-to-even) then astype(int32) */
            int gx = (int)rint((xs[i] - ox) / res);
            int gy = (int)rint((ys[i] - oy) / res);
            y_add_point(c->gridf, G, gx, gy, kernel, ksz);
        }
    }
    free(kernel); free(keep);
    for (size_t i = 0; i < (size_t)G * G; i++) c->grid[i] = (uint8_t)(int)(100 * c->gridf[i]);

    /* query.points_local(): models.py:96-97 -> points_for_pose2d(0,0,0) */
    orc_scan ql = *query;
    ql.pose[0] = ql.pose[1] = ql.pose[2] = 0.0;
    int np = orc_point_readings(&ql, ORC_SEM_YAGPY, xs, ys);
    out->n_query_points = np;
    free(c->qlocal);
    c->qlocal = (double *)malloc(sizeof(double) * 2 * (np + 1));
    c->n_qlocal = np;
    for (int i = 0; i < np; i++) { c->qlocal[2 * i] = xs[i]; c->qlocal[2 * i + 1] = ys[i]; }

    y_best co, fi;
    int rc = y_find_best_pose(c, 0, c->grid, G, xs, ys, np, query->pose[0], query->pose[1],
                              query->pose[2], ox, oy, g->search_size * 0.5, res * 2,
                              g->coarse_search_angle_offset * 0.5, g->coarse_angle_resolution, res,
                              penalize, &co);
    if (rc) { free(xs); free(ys); return rc; }
    memcpy(out->coarse_dims, c->dims[0], sizeof out->coarse_dims);
    out->hypotheses += (long long)c->dims[0][0] * c->dims[0][1] * c->dims[0][2];
    double th;
    y_best fin = co;
    free_pass(c, 1);
    if (refine) {
        rc = y_find_best_pose(c, 1, c->grid, G, xs, ys, np, co.x, co.y, co.t, ox, oy, res * 2, res,
                              0.0349 * 0.5, 0.00349, res, penalize, &fi);
        if (rc) { free(xs); free(ys); return rc; }
        memcpy(out->fine_dims, c->dims[1], sizeof out->fine_dims);
        out->hypotheses += (long long)c->dims[1][0] * c->dims[1][1] * c->dims[1][2];
        fin = fi;
        th = fi.th;
    } else {
        th = 4 * g->coarse_angle_resolution;
    }
    free(xs); free(ys);
    out->response = fin.response;
    out->pose[0] = fin.x; out->pose[1] = fin.y; out->pose[2] = fin.t;
    out->cov[0] = co.xx; out->cov[1] = co.xy; out->cov[2] = 0;
    out->cov[3] = co.xy; out->cov[4] = co.yy; out->cov[5] = 0;
    out->cov[6] = 0; out->cov[7] = 0; out->cov[8] = th;
    return 0;
}

/* ================================================================== public entry + accessors */
int orc_match(orc_ctx *c, const orc_scan *query, const orc_scan *base, int n_base, int penalize,
              int refine, orc_result *out) {
    if (!c || !query || !out || (n_base > 0 && !base)) return fail("null argument");
    g_err[0] = 0;
    if (c->cfg.semantics == ORC_SEM_KARTO) return k_match(c, query, base, n_base, penalize, refine, out);
    return y_match(c, query, base, n_base, penalize, refine, out);
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
