/*
 * ym_throughput.c -- CPU throughput baseline of bench.py: a farm of oracle matchers, one per physical core.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY (see ym_oracle.h): never part of the product.  Every thread owns one oracle context
 * (its own correlation grid, single-threaded, Karto semantics), is pinned to one cpu of the list it is given, and runs the
 * same cfg2 match again and again until the deadline; the program prints the number of matches completed.  No Python, no
 * shared allocator state beyond malloc's per-thread arenas: what N independent Karto processes on N cores would do.
 *
 *   ym_throughput <input file> <seconds> <cpu,cpu,...>
 * input file (doubles, written by bench.py): 12 config values in orc_config order (use_response_expansion as a double),
 * n_base, n_beams, min_angle, angle_increment, min_range, range_threshold, then 1 + n_base scans (the query first), each
 * pose[3] + ranges[n_beams].
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ym_oracle.h"

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

typedef struct {
    int cpu, n_base;
    orc_config cfg;
    const orc_scan *query, *base;
    pthread_barrier_t *start;
    const double *deadline;
    long matches;
    long long hyp;
    int failed;
} worker;

static void *run(void *arg) {
    worker *w = (worker *)arg;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(w->cpu, &set);
    pthread_setaffinity_np(pthread_self(), sizeof set, &set); /* (best effort: a refused mask leaves the thread where it is) */
    orc_ctx *c = orc_create(&w->cfg);
    orc_result r;
    memset(&r, 0, sizeof r);
    if (!c || orc_match(c, w->query, w->base, w->n_base, 1, 1, &r)) w->failed = 1; /* (first touch of the grid on this core's memory) */
    w->hyp = r.hypotheses;
    pthread_barrier_wait(w->start); /* every worker is ready */
    pthread_barrier_wait(w->start); /* the deadline is set */
    if (!w->failed)
        while (now() < *w->deadline) {
            if (orc_match(c, w->query, w->base, w->n_base, 1, 1, &r)) { w->failed = 1; break; }
            w->matches++;
        }
    if (c) orc_destroy(c);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s input seconds cpu,cpu,...\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    double *d = (double *)malloc((size_t)bytes);
    if (!d || fread(d, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    orc_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.angle_variance_penalty = d[0]; cfg.distance_variance_penalty = d[1]; cfg.coarse_search_angle_offset = d[2];
    cfg.coarse_angle_resolution = d[3]; cfg.fine_search_angle_resolution = d[4]; cfg.use_response_expansion = (int)d[5];
    cfg.range_threshold = d[6]; cfg.minimum_angle_penalty = d[7]; cfg.minimum_distance_penalty = d[8];
    cfg.search_size = d[9]; cfg.resolution = d[10]; cfg.smear_deviation = d[11];
    cfg.semantics = ORC_SEM_KARTO; cfg.threads = 1;
    const int n_base = (int)d[12], n_beams = (int)d[13];
    if ((size_t)bytes != sizeof(double) * (18 + (size_t)(1 + n_base) * (3 + (size_t)n_beams))) { fprintf(stderr, "input size mismatch\n"); return 2; }
    orc_scan *scans = (orc_scan *)calloc((size_t)n_base + 1, sizeof *scans);
    const double *p = d + 18;
    for (int i = 0; i <= n_base; i++, p += 3 + n_beams) {
        scans[i].ranges = p + 3; scans[i].n = n_beams; scans[i].min_angle = d[14]; scans[i].angle_increment = d[15];
        scans[i].min_range = d[16]; scans[i].range_threshold = d[17];
        scans[i].pose[0] = p[0]; scans[i].pose[1] = p[1]; scans[i].pose[2] = p[2];
    }
    const double seconds = atof(argv[2]);
    int cpus[4096], n = 0;
    for (char *tok = strtok(argv[3], ","); tok && n < 4096; tok = strtok(NULL, ",")) cpus[n++] = atoi(tok);
    if (n < 1) return 2;
    pthread_barrier_t start;
    pthread_barrier_init(&start, NULL, (unsigned)n + 1);
    double deadline = 1e300;
    worker *w = (worker *)calloc((size_t)n, sizeof *w);
    pthread_t *th = (pthread_t *)calloc((size_t)n, sizeof *th);
    for (int i = 0; i < n; i++) {
        w[i].cpu = cpus[i]; w[i].n_base = n_base; w[i].cfg = cfg; w[i].query = &scans[0]; w[i].base = &scans[1];
        w[i].start = &start; w[i].deadline = &deadline;
        pthread_create(&th[i], NULL, run, &w[i]);
    }
    pthread_barrier_wait(&start); /* every worker has created its context and run one match */
    const double t1 = now();
    deadline = t1 + seconds;
    pthread_barrier_wait(&start); /* go */
    for (int i = 0; i < n; i++) pthread_join(th[i], NULL);
    const double dt = now() - t1;
    long total = 0, lo = -1, hi = 0; int failed = 0;
    for (int i = 0; i < n; i++) {
        total += w[i].matches; failed |= w[i].failed;
        if (lo < 0 || w[i].matches < lo) lo = w[i].matches;
        if (w[i].matches > hi) hi = w[i].matches;
    }
    printf("{\"threads\": %d, \"matches\": %ld, \"seconds\": %.6f, \"hypotheses_per_match\": %lld, \"min_matches_per_thread\": %ld, "
           "\"max_matches_per_thread\": %ld, \"failed\": %d}\n", n, total, dt, w[0].hyp, lo, hi, failed);
    return failed ? 1 : 0;
}
