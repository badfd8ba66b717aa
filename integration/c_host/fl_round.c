/*
 * One federated round against librofl_zk.so from a plain C99 host -- no Python, no torch: the shape of the reference's processes
 * (rofl_service/src/flserver/client.rs:265-266 clients as tasks of one process; server.rs:379-384, 513-521, 656-687 one server process
 * that hands every client to a verification pool), written against include/rofl_zk.h only.  It is the compiled-host stand-in for the
 * Rust binding of INTEGRATION.md (no Rust toolchain in this image): tests/test_c_host.py builds it with gcc -std=c99 -pedantic -Werror,
 * runs it on the GPU box and compares every byte it wrote with the ctypes path and the oracle.
 *
 *   fl_round sizes
 *       no GPU: prints the size helpers for a few shapes (the library loads and links from C)
 *   fl_round bench <d> <prove_range> <n_partition> <iterations>
 *       one client, create + verify back to back from host buffers (BASELINE cfg 2 when called with 25000 32 4): the library's latency
 *       as a compiled host sees it, on the system HIP runtime, with no interpreter in the loop; prints the median and the extremes
 *   fl_round run <d> <prove_range> <n_partition> <n_clients> <n_devices> <out-file>
 *       client role : one pthread per device, bound with rofl_set_device; client i is proved on device i % n_devices
 *       server role : rofl_set_option("devices", mask) + ("verify_batch", 2), ONE rofl_verify_rangeproof_batch call for the round;
 *                     then the same call with one byte of client 1's first proof flipped (only that client may fail)
 *       out-file    : header (8 x u64: d, prove_range, n_partition, n_clients, n_proofs, proof_len, fp_bits, fp_frac), then per client
 *                     values (d f32), blindings (d x 32), nonce seed (32), proofs (n_proofs x proof_len), commitments (d x 32);
 *                     then verdicts of the clean round and of the tampered round (n_clients x i32 each)
 *   fl_round reject <d> <prove_range> <n_partition> <n_clients> <n_devices> <out-file>
 *       the same round, then what a batch verifier is defined by -- what it rejects (server.rs:474-484 fails the round on any bad client):
 *       five scenarios (clean | one late R flipped | a late L, two swapped commitments and a non-canonical scalar in three clients |
 *       a = 0 in every chunk of one client: scalars that collide in the sort | clean again), each verified twice: the server path
 *       (verify_batch = 2, devices = all) and the per-client path (verify_batch = 1, one device)
 *       out-file    : header as above, then 5 x 2 x n_clients verdicts (i32) and, per scenario, the index of the chunk that was touched
 *                     in each client (n_clients x i32, -1 = untouched), then per client proofs and commitments of scenarios 1..3 as verified
 *   fl_round split <d> <prove_range> <n_partition> <n_devices> <out-file>
 *       ONE client over several devices ("cfg 2/3 at > 1 GPU -> chunks over ranks"; the reference proves a client's chunks in
 *       parallel on its rayon pool, range_proof_vec/mod.rs:54-78, 168-181), three ways that must give the same bytes:
 *       (a) rofl_create_rangeproof on one device; (b) the same call with rofl_set_option("devices", mask): the library deals the chunks;
 *       (c) what a rank of a one-process-per-GPU host does: rofl_create_rangeproof_chunks per contiguous run, each run on its own thread
 *       and device, results placed by the caller -- then rofl_verify_rangeproof_chunks per run (clean, and with a byte of the LAST run's
 *       proof flipped: only that run may fail) and rofl_verify_rangeproof over the assembled set with the devices option.
 *       out-file    : header as above (n_clients = 1), values, blindings, nonce seed, proofs and commitments of (a); exit code 1 on any difference
 *   fl_round comm <id-file> <rank> <world>
 *       one process per GPU, no Python: rank 0 draws the RCCL unique id (rofl_comm_unique_id) and publishes it in <id-file>, the other ranks
 *       wait for the file; rofl_set_device(rank), rofl_comm_init, then one round's exchange -- each rank proves one small client, the ranks
 *       all-gather [verdict | proofs | commitments], every rank verifies the proofs of rank + 1, MIN of the verdicts -- and a barrier.
 *       (A world of one runs on the one-GPU test box; RCCL refuses two ranks on one GPU.)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "rofl_zk.h"

#define FP_BITS 32u
#define FP_FRAC 7u

static uint64_t lcg_state;
static uint32_t lcg(void) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(lcg_state >> 32);
}

static void die(const char *what, int rc) {
    char msg[512];
    msg[0] = 0;
    rofl_last_error(msg, sizeof msg);
    fprintf(stderr, "fl_round: %s failed with %d: %s\n", what, rc, msg);
    exit(1);
}

typedef struct {
    size_t d, prove_range, n_partition, n_clients, n_devices, n_proofs, proof_len;
    float **values;
    uint8_t **blindings, **proofs, **commits;
    rofl_nonce_t *nonces;
    int device;
    int rc;
} round_t;

static void *client_thread(void *arg) {
    round_t *r = (round_t *)arg;
    int bound = -1;
    size_t i;
    r->rc = rofl_set_device(r->device);
    if (r->rc) return NULL;
    rofl_get_device(&bound);
    if (bound != r->device) { r->rc = -2; return NULL; }
    for (i = (size_t)r->device; i < r->n_clients; i += r->n_devices) {
        size_t plen = 0, np = 0;
        r->rc = rofl_create_rangeproof(r->values[i], r->d, r->blindings[i], r->d, r->prove_range, r->n_partition, FP_BITS, FP_FRAC,
                                       &r->nonces[i], r->proofs[i], &plen, &np, r->commits[i]);
        if (r->rc) return NULL;
        if (plen != r->proof_len || np != r->n_proofs) { r->rc = -3; return NULL; }
    }
    return NULL;
}

static int sizes(void) {
    static const size_t shapes[][3] = {{5000, 8, 4}, {25000, 32, 4}, {25000, 32, 64}, {55000, 32, 4}, {3, 8, 1}};
    size_t k;
    for (k = 0; k < sizeof shapes / sizeof shapes[0]; k++) {
        size_t d = shapes[k][0], n = shapes[k][1], p = shapes[k][2];
        printf("%zu %zu %zu %zu %zu %zu\n", d, n, p, rofl_next_pow2(d), rofl_rangeproof_chunks(d, p), rofl_rangeproof_size(n, d, p));
    }
    return 0;
}

static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

static int bench(size_t d, size_t prove_range, size_t n_partition, size_t iters) {
    size_t n_proofs = rofl_rangeproof_chunks(d, n_partition), proof_len = rofl_rangeproof_size(prove_range, d, n_partition), j, it;
    float *values = (float *)malloc(d * sizeof(float)), lo = 0, hi = 0;
    uint8_t *blindings = (uint8_t *)malloc(d * 32), *proofs = (uint8_t *)malloc(n_proofs * proof_len), *commits = (uint8_t *)malloc(d * 32);
    double *tc = (double *)malloc(iters * sizeof(double)), *tv = (double *)malloc(iters * sizeof(double)), *ts = (double *)malloc(iters * sizeof(double));
    uint8_t seed[32];
    rofl_nonce_t nonce;
    int rc, ok = 0;
    if (!n_proofs || !proof_len || iters < 3) return 2;
    rc = rofl_get_clip_bounds(prove_range, FP_BITS, FP_FRAC, &lo, &hi);
    if (rc) die("rofl_get_clip_bounds", rc);
    lcg_state = 12345;
    for (j = 0; j < d; j++) values[j] = (float)((double)lo + (double)lcg() / 4294967296.0 * 0.999 * ((double)hi - (double)lo));
    for (j = 0; j < d * 32; j++) blindings[j] = (uint8_t)(lcg() >> 24);
    for (j = 0; j < d; j++) blindings[j * 32 + 31] &= 0x0f;
    memset(seed, 0x5a, sizeof seed);
    memset(&nonce, 0, sizeof nonce);
    nonce.mode = 1;
    rc = rofl_set_device(0);
    if (rc) die("rofl_set_device", rc);
    rc = rofl_bp_gens_prepare(prove_range, rofl_next_pow2(d) / n_partition);
    if (rc) die("rofl_bp_gens_prepare", rc);
    for (it = 0; it < iters + 3; it++) {                                    /* three warm-ups */
        size_t plen = 0, np = 0;
        double t0, t1, t2;
        nonce.seed[0] = (uint8_t)it;
        t0 = now_ms();
        rc = rofl_create_rangeproof(values, d, blindings, d, prove_range, n_partition, FP_BITS, FP_FRAC, &nonce, proofs, &plen, &np, commits);
        if (rc) die("rofl_create_rangeproof", rc);
        t1 = now_ms();
        rc = rofl_verify_rangeproof(proofs, plen, np, commits, d, prove_range, FP_BITS, FP_FRAC, seed, &ok);
        if (rc || !ok) die("rofl_verify_rangeproof", rc ? rc : -1);
        t2 = now_ms();
        if (it >= 3) { tc[it - 3] = t1 - t0; tv[it - 3] = t2 - t1; ts[it - 3] = t2 - t0; }
        if (getenv("FL_ROUND_EACH")) fprintf(stderr, "iteration %zu: create %.3f verify %.3f ms\n", it, t1 - t0, t2 - t1);
    }
    qsort(tc, iters, sizeof(double), cmp_double);
    qsort(tv, iters, sizeof(double), cmp_double);
    qsort(ts, iters, sizeof(double), cmp_double);
    printf("{\"host\": \"C99, no interpreter\", \"d\": %zu, \"prove_range\": %zu, \"n_partition\": %zu, \"iterations\": %zu, "
           "\"create_ms\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f}, \"verify_ms\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f}, "
           "\"create_plus_verify_ms\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f}, \"elements_per_s_at_median\": %.0f}\n",
           d, prove_range, n_partition, iters, tc[iters / 2], tc[0], tc[iters - 1], tv[iters / 2], tv[0], tv[iters - 1],
           ts[iters / 2], ts[0], ts[iters - 1], (double)d / (ts[iters / 2] * 1e-3));
    return 0;
}

static void setup_round(round_t *R, char **argv) {
    size_t i, j;
    float lo = 0, hi = 0;
    int rc;
    memset(R, 0, sizeof *R);
    R->d = (size_t)strtoul(argv[2], NULL, 10);
    R->prove_range = (size_t)strtoul(argv[3], NULL, 10);
    R->n_partition = (size_t)strtoul(argv[4], NULL, 10);
    R->n_clients = (size_t)strtoul(argv[5], NULL, 10);
    R->n_devices = (size_t)strtoul(argv[6], NULL, 10);
    if (!R->d || !R->n_clients || !R->n_devices || R->n_devices > 16) exit(2);
    R->n_proofs = rofl_rangeproof_chunks(R->d, R->n_partition);
    R->proof_len = rofl_rangeproof_size(R->prove_range, R->d, R->n_partition);
    if (!R->n_proofs || !R->proof_len) { fprintf(stderr, "fl_round: the size helpers reject this shape\n"); exit(2); }

    /* inputs: values inside the clip interval of (prove_range, fp), canonical blinding scalars, one nonce seed per client */
    rc = rofl_get_clip_bounds(R->prove_range, FP_BITS, FP_FRAC, &lo, &hi);
    if (rc) die("rofl_get_clip_bounds", rc);
    R->values = (float **)calloc(R->n_clients, sizeof *R->values);
    R->blindings = (uint8_t **)calloc(R->n_clients, sizeof *R->blindings);
    R->proofs = (uint8_t **)calloc(R->n_clients, sizeof *R->proofs);
    R->commits = (uint8_t **)calloc(R->n_clients, sizeof *R->commits);
    R->nonces = (rofl_nonce_t *)calloc(R->n_clients, sizeof *R->nonces);
    lcg_state = 0x726f666c5f7a6bull + R->d;
    for (i = 0; i < R->n_clients; i++) {
        R->values[i] = (float *)malloc(R->d * sizeof(float));
        R->blindings[i] = (uint8_t *)malloc(R->d * 32);
        R->proofs[i] = (uint8_t *)malloc(R->n_proofs * R->proof_len);
        R->commits[i] = (uint8_t *)malloc(R->d * 32);
        for (j = 0; j < R->d; j++) {
            double u = (double)lcg() / 4294967296.0;
            R->values[i][j] = (float)((double)lo + u * 0.999 * ((double)hi - (double)lo));
        }
        for (j = 0; j < R->d * 32; j++) R->blindings[i][j] = (uint8_t)(lcg() >> 24);
        for (j = 0; j < R->d; j++) R->blindings[i][j * 32 + 31] &= 0x0f;      /* < 2^252: canonical */
        R->nonces[i].mode = 1;
        for (j = 0; j < 32; j++) R->nonces[i].seed[j] = (uint8_t)(lcg() >> 24);
    }
}

/* client role: one thread per device */
static void prove_round(round_t *R) {
    round_t *per_dev = (round_t *)calloc(R->n_devices, sizeof *per_dev);
    pthread_t *th = (pthread_t *)calloc(R->n_devices, sizeof *th);
    size_t i;
    for (i = 0; i < R->n_devices; i++) {
        per_dev[i] = *R;
        per_dev[i].device = (int)i;
        if (pthread_create(&th[i], NULL, client_thread, &per_dev[i])) exit(1);
    }
    for (i = 0; i < R->n_devices; i++) {
        pthread_join(th[i], NULL);
        if (per_dev[i].rc) die("client thread (rofl_set_device / rofl_create_rangeproof)", per_dev[i].rc);
    }
    free(per_dev);
    free(th);
}

static void write_head(FILE *f, const round_t *R) {
    uint64_t head[8];
    head[0] = R->d; head[1] = R->prove_range; head[2] = R->n_partition; head[3] = R->n_clients;
    head[4] = R->n_proofs; head[5] = R->proof_len; head[6] = FP_BITS; head[7] = FP_FRAC;
    fwrite(head, sizeof head, 1, f);
}

/* the server role under attack: see the header comment */
#define N_SCEN 5
static int reject(char **argv) {
    round_t R;
    size_t i, s, lg = 0, chunk, plen, psz;
    long mask = 0;
    int rc, *verdict, *touched;
    uint8_t seed[32], **p, **c;
    FILE *f;
    setup_round(&R, argv);
    if (R.n_clients < 6 || R.d < 64) { fprintf(stderr, "fl_round reject: at least six clients of 64 values\n"); return 2; }
    prove_round(&R);
    plen = R.proof_len; psz = R.n_proofs * plen;
    lg = (plen / 32 - 9) / 2;
    chunk = rofl_next_pow2(R.d) / R.n_proofs;
    for (i = 0; i < R.n_devices; i++) mask |= 1L << i;
    rc = rofl_set_device(0);
    if (rc) die("rofl_set_device(0)", rc);
    memset(seed, 0x5a, sizeof seed);
    verdict = (int *)calloc(N_SCEN * 2 * R.n_clients, sizeof(int));
    touched = (int *)malloc(N_SCEN * R.n_clients * sizeof(int));
    for (i = 0; i < N_SCEN * R.n_clients; i++) touched[i] = -1;
    p = (uint8_t **)calloc(R.n_clients, sizeof *p);
    c = (uint8_t **)calloc(R.n_clients, sizeof *c);
    for (i = 0; i < R.n_clients; i++) { p[i] = (uint8_t *)malloc(psz); c[i] = (uint8_t *)malloc(R.d * 32); }
    f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 1; }
    write_head(f, &R);
    fseek(f, (long)(64 + (N_SCEN * 2 + N_SCEN) * R.n_clients * sizeof(int)), SEEK_SET);
    for (s = 0; s < N_SCEN; s++) {
        int path;
        size_t n = R.n_clients, last = R.n_proofs - 1;
        for (i = 0; i < n; i++) { memcpy(p[i], R.proofs[i], psz); memcpy(c[i], R.commits[i], R.d * 32); }
        if (s == 1) {                                  /* the last round's R of the last full chunk of the last client */
            size_t k = n - 1, ch = R.n_proofs > 1 ? last - 1 : 0;
            p[k][ch * plen + 7 * 32 + 64 * (lg - 1) + 32 + 5] ^= 0x10; touched[s * n + k] = (int)ch;
        } else if (s == 2) {
            uint8_t tmp[32];
            size_t a = 1, b = n / 2, e = n - 2, chb = R.n_proofs > 2 ? 1 : 0;
            p[a][7 * 32 + 64 * (lg - 2) + 9] ^= 0x01; touched[s * n + a] = 0;                               /* a late L of chunk 0 */
            memcpy(tmp, c[b] + (chb * chunk + 3) * 32, 32);                                                   /* two commitments swapped */
            memcpy(c[b] + (chb * chunk + 3) * 32, c[b] + (chb * chunk + 4) * 32, 32);
            memcpy(c[b] + (chb * chunk + 4) * 32, tmp, 32); touched[s * n + b] = (int)chb;
            memset(p[e] + 128, 0xff, 32); touched[s * n + e] = 0;                                             /* t_x >= l: not canonical */
        } else if (s == 3) {                           /* a = 0 in every chunk: one scalar for all G terms of the client's own check */
            size_t k = 2, ch;
            for (ch = 0; ch < R.n_proofs; ch++) memset(p[k] + ch * plen + plen - 64, 0, 32);
            touched[s * n + k] = 0;
        }
        for (path = 0; path < 2; path++) {
            rc = rofl_set_option("verify_batch", path == 0 ? 2 : 1);
            if (rc) die("rofl_set_option(verify_batch)", rc);
            rc = rofl_set_option("devices", path == 0 ? mask : 0);
            if (rc) die("rofl_set_option(devices)", rc);
            rc = rofl_verify_rangeproof_batch(n, (const uint8_t *const *)p, plen, R.n_proofs, (const uint8_t *const *)c, R.d, R.prove_range,
                                              FP_BITS, FP_FRAC, seed, verdict + (s * 2 + (size_t)path) * n);
            if (rc) die("rofl_verify_rangeproof_batch", rc);
        }
        if (s >= 1 && s <= 3) for (i = 0; i < n; i++) { fwrite(p[i], 1, psz, f); fwrite(c[i], 32, R.d, f); }
        printf("scenario %zu:", s);
        for (i = 0; i < n; i++) printf(" %d%d", verdict[(s * 2) * n + i], verdict[(s * 2 + 1) * n + i]);
        printf("\n");
    }
    fseek(f, 64, SEEK_SET);
    fwrite(verdict, sizeof(int), N_SCEN * 2 * R.n_clients, f);
    fwrite(touched, sizeof(int), N_SCEN * R.n_clients, f);
    fclose(f);
    return 0;
}

/* one process per GPU: the exchange of a round through the library's RCCL communicator */
/* ---- one client split by chunks ---- */
typedef struct {
    const float *values; const uint8_t *blindings; const rofl_nonce_t *nonce;
    size_t d, prove_range, n_partition, first, count, proof_len, m;
    uint8_t *proofs, *commits;      /* the CLIENT's arrays: the run writes its own part */
    int device, rc, ok, ok_bad;
    const uint8_t *seed;
} run_t;

static void *run_thread(void *arg) {
    run_t *r = (run_t *)arg;
    size_t plen = 0, nc = 0, lo = r->first * r->m, want;
    if (lo > r->d) lo = r->d;
    want = (r->first + r->count) * r->m;
    if (want > r->d) want = r->d;
    want -= lo;
    r->rc = rofl_set_device(r->device);
    if (r->rc) return NULL;
    r->rc = rofl_create_rangeproof_chunks(r->values, r->d, r->blindings, r->d, r->prove_range, r->n_partition, FP_BITS, FP_FRAC, r->nonce, r->first, r->count,
                                          r->proofs + r->first * r->proof_len, &plen, r->commits + lo * 32, &nc);
    if (r->rc) return NULL;
    if (plen != r->proof_len || nc != want) { r->rc = -3; return NULL; }
    r->rc = rofl_verify_rangeproof_chunks(r->proofs + r->first * r->proof_len, r->proof_len, rofl_rangeproof_chunks(r->d, r->n_partition), r->first, r->count,
                                          r->commits + lo * 32, r->d, r->prove_range, FP_BITS, FP_FRAC, r->seed, &r->ok);
    return NULL;
}

static int split_mode(char **argv) {
    size_t d = (size_t)strtoul(argv[2], NULL, 10), prove_range = (size_t)strtoul(argv[3], NULL, 10), n_partition = (size_t)strtoul(argv[4], NULL, 10);
    size_t n_devices = (size_t)strtoul(argv[5], NULL, 10), n_proofs = rofl_rangeproof_chunks(d, n_partition), proof_len = rofl_rangeproof_size(prove_range, d, n_partition);
    size_t m = rofl_next_pow2(d) / n_proofs, nruns = n_devices < n_proofs ? n_devices : n_proofs, j, k, plen = 0, np = 0;
    float *values = (float *)malloc(d * sizeof(float)), lim;
    uint8_t *blindings = (uint8_t *)malloc(d * 32), *pa = (uint8_t *)calloc(n_proofs, proof_len), *ca = (uint8_t *)calloc(d, 32);
    uint8_t *pb = (uint8_t *)calloc(n_proofs, proof_len), *cb = (uint8_t *)calloc(d, 32), *pc = (uint8_t *)calloc(n_proofs, proof_len), *cc = (uint8_t *)calloc(d, 32);
    run_t *runs = (run_t *)calloc(nruns, sizeof *runs);
    pthread_t *th = (pthread_t *)calloc(nruns, sizeof *th);
    rofl_nonce_t nonce;
    uint8_t seed[32];
    long mask = 0;
    int rc, ok = 0, bad = 0;
    FILE *f;
    round_t H;
    lcg_state = 0x5eed0000ull + d;
    lim = (float)((double)((1ull << (prove_range - 1)) - 1) / (double)(1u << FP_FRAC)) * 0.999f;
    for (j = 0; j < d; j++) values[j] = ((float)(lcg() >> 8) / 8388608.0f - 1.0f) * lim;
    for (j = 0; j < d * 32; j++) blindings[j] = (uint8_t)(lcg() >> 24);
    for (j = 0; j < d; j++) blindings[j * 32 + 31] &= 0x0f;
    memset(&nonce, 0, sizeof nonce);
    nonce.mode = 1;
    memset(nonce.seed, 0x42, 32);
    memset(seed, 0x5a, sizeof seed);
    /* (a) one device */
    rc = rofl_set_device(0);
    if (rc) die("rofl_set_device(0)", rc);
    rc = rofl_create_rangeproof(values, d, blindings, d, prove_range, n_partition, FP_BITS, FP_FRAC, &nonce, pa, &plen, &np, ca);
    if (rc) die("rofl_create_rangeproof", rc);
    if (plen != proof_len || np != n_proofs) { fprintf(stderr, "fl_round: unexpected proof geometry\n"); return 1; }
    /* (b) the library deals the chunks */
    for (j = 0; j < n_devices; j++) mask |= 1L << j;
    rc = rofl_set_option("devices", mask);
    if (rc) die("rofl_set_option(devices)", rc);
    rc = rofl_create_rangeproof(values, d, blindings, d, prove_range, n_partition, FP_BITS, FP_FRAC, &nonce, pb, &plen, &np, cb);
    if (rc) die("rofl_create_rangeproof (devices)", rc);
    rc = rofl_verify_rangeproof(pb, proof_len, n_proofs, cb, d, prove_range, FP_BITS, FP_FRAC, seed, &ok);
    if (rc) die("rofl_verify_rangeproof (devices)", rc);
    pb[(n_proofs - 1) * proof_len + 70] ^= 0x10;
    rc = rofl_verify_rangeproof(pb, proof_len, n_proofs, cb, d, prove_range, FP_BITS, FP_FRAC, seed, &bad);
    if (rc) die("rofl_verify_rangeproof (devices, tampered)", rc);
    pb[(n_proofs - 1) * proof_len + 70] ^= 0x10;
    rc = rofl_set_option("devices", 0);
    if (rc) die("rofl_set_option(devices, 0)", rc);
    if (memcmp(pa, pb, n_proofs * proof_len) || memcmp(ca, cb, d * 32) || !ok || bad) { fprintf(stderr, "fl_round: the devices split differs (ok %d, tampered %d)\n", ok, bad); return 1; }
    /* (c) the caller deals contiguous runs, one thread and device per run */
    for (k = 0; k < nruns; k++) {
        run_t *r = &runs[k];
        r->values = values; r->blindings = blindings; r->nonce = &nonce; r->d = d; r->prove_range = prove_range; r->n_partition = n_partition;
        r->first = k * n_proofs / nruns; r->count = (k + 1) * n_proofs / nruns - r->first; r->proof_len = proof_len; r->m = m;
        r->proofs = pc; r->commits = cc; r->device = (int)k; r->seed = seed;
        if (pthread_create(&th[k], NULL, run_thread, r)) { perror("pthread_create"); return 1; }
    }
    for (k = 0; k < nruns; k++) pthread_join(th[k], NULL);
    for (k = 0; k < nruns; k++) {
        if (runs[k].rc) die("a run of chunks", runs[k].rc);
        if (!runs[k].ok) { fprintf(stderr, "fl_round: run %zu did not verify\n", k); return 1; }
    }
    if (memcmp(pa, pc, n_proofs * proof_len) || memcmp(ca, cc, d * 32)) { fprintf(stderr, "fl_round: the runs' bytes differ from the whole call's\n"); return 1; }
    /* a flipped byte in the last run: that run fails, the first still verifies */
    pc[(n_proofs - 1) * proof_len + 70] ^= 0x10;
    for (k = 0; k < nruns; k += (nruns > 1 ? nruns - 1 : 1)) {
        run_t *r = &runs[k];
        size_t lo = r->first * m > d ? d : r->first * m;
        rc = rofl_verify_rangeproof_chunks(pc + r->first * proof_len, proof_len, n_proofs, r->first, r->count, cc + lo * 32, d, prove_range, FP_BITS, FP_FRAC, seed, &r->ok_bad);
        if (rc) die("rofl_verify_rangeproof_chunks (tampered)", rc);
    }
    if (runs[nruns - 1].ok_bad != 0 || (nruns > 1 && runs[0].ok_bad != 1)) { fprintf(stderr, "fl_round: tampered runs: first %d last %d\n", runs[0].ok_bad, runs[nruns - 1].ok_bad); return 1; }
    f = fopen(argv[6], "wb");
    if (!f) { perror(argv[6]); return 1; }
    memset(&H, 0, sizeof H);
    H.d = d; H.prove_range = prove_range; H.n_partition = n_partition; H.n_clients = 1; H.n_proofs = n_proofs; H.proof_len = proof_len;
    write_head(f, &H);
    fwrite(values, sizeof(float), d, f);
    fwrite(blindings, 32, d, f);
    fwrite(nonce.seed, 1, 32, f);
    fwrite(pa, proof_len, n_proofs, f);
    fwrite(ca, 32, d, f);
    fclose(f);
    printf("split ok: d=%zu chunks=%zu runs=%zu\n", d, n_proofs, nruns);
    return 0;
}

static int comm_mode(const char *id_file, int rank, int world) {
    enum { D = 600, NB = 32, PART = 4 };
    uint8_t id[128], seed[32], *mine, *all, blind[D * 32];
    float values[D];
    size_t n_proofs = rofl_rangeproof_chunks(D, PART), plen = rofl_rangeproof_size(NB, D, PART), rec, i, got_len = 0, got_np = 0;
    rofl_nonce_t nonce;
    int rc, ok = 0, r = -1, w = 0, ver = 0, src;
    double v[2];
    char lib[512];
    FILE *f;
    if (world < 1 || rank < 0 || rank >= world) return 2;
    rc = rofl_set_device(rank);
    if (rc) die("rofl_set_device", rc);
    if (rank == 0) {
        char tmp[1024];
        rc = rofl_comm_unique_id(id);
        if (rc) die("rofl_comm_unique_id", rc);
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, 128, f) != 128) { perror(tmp); return 1; }
        fclose(f);
        if (rename(tmp, id_file)) { perror(id_file); return 1; }      /* atomic: a reader never sees half an id */
    } else {
        for (i = 0; i < 6000; i++) {                                  /* up to a minute */
            struct timespec ts;
            f = fopen(id_file, "rb");
            if (f) { size_t k = fread(id, 1, 128, f); fclose(f); if (k == 128) break; }
            ts.tv_sec = 0; ts.tv_nsec = 10000000;
            nanosleep(&ts, NULL);
        }
        if (i == 6000) { fprintf(stderr, "fl_round comm: no unique id in %s\n", id_file); return 1; }
    }
    rc = rofl_comm_init(id, rank, world);
    if (rc) die("rofl_comm_init", rc);
    rc = rofl_comm_info(&r, &w, &ver, lib, sizeof lib);
    if (rc || r != rank || w != world) die("rofl_comm_info", rc ? rc : -1);
    /* this rank's client */
    lcg_state = 777u + (uint64_t)rank;
    for (i = 0; i < D; i++) values[i] = (float)((double)lcg() / 4294967296.0 * 200.0 - 100.0);
    for (i = 0; i < D * 32; i++) blind[i] = (uint8_t)(lcg() >> 24);
    for (i = 0; i < D; i++) blind[i * 32 + 31] &= 0x0f;
    memset(&nonce, 0, sizeof nonce);
    nonce.mode = 1;
    memset(nonce.seed, 0x30 + rank, 32);
    rec = 1 + n_proofs * plen + (size_t)D * 32;
    mine = (uint8_t *)malloc(rec);
    all = (uint8_t *)malloc(rec * (size_t)world);
    rc = rofl_create_rangeproof(values, D, blind, D, NB, PART, FP_BITS, FP_FRAC, &nonce, mine + 1, &got_len, &got_np, mine + 1 + n_proofs * plen);
    if (rc || got_len != plen || got_np != n_proofs) die("rofl_create_rangeproof", rc ? rc : -1);
    mine[0] = 1;
    rc = rofl_comm_allgather(mine, rec, all);
    if (rc) die("rofl_comm_allgather", rc);
    if (memcmp(all + (size_t)rank * rec, mine, rec) != 0) { fprintf(stderr, "fl_round comm: my own record came back changed\n"); return 1; }
    src = (rank + 1) % world;
    memset(seed, 0x5a, sizeof seed);
    rc = rofl_verify_rangeproof(all + (size_t)src * rec + 1, plen, n_proofs, all + (size_t)src * rec + 1 + n_proofs * plen, D, NB, FP_BITS, FP_FRAC, seed, &ok);
    if (rc) die("rofl_verify_rangeproof", rc);
    v[0] = ok ? 1.0 : 0.0; v[1] = 1.0;
    rc = rofl_comm_allreduce_f64(v, 1, 1);                             /* MIN of the verdicts */
    if (rc) die("rofl_comm_allreduce_f64(min)", rc);
    rc = rofl_comm_allreduce_f64(v + 1, 1, 0);                         /* how many ranks joined */
    if (rc) die("rofl_comm_allreduce_f64(sum)", rc);
    rc = rofl_comm_barrier();
    if (rc) die("rofl_comm_barrier", rc);
    rc = rofl_comm_destroy();
    if (rc) die("rofl_comm_destroy", rc);
    printf("rank %d of %d: round verified %d, ranks joined %.0f, rccl %d from %s\n", rank, world, (int)v[0], v[1], ver, lib);
    free(mine);
    free(all);
    return v[0] == 1.0 && (int)v[1] == world ? 0 : 1;
}

int main(int argc, char **argv) {
    round_t R;
    FILE *f;
    size_t i;
    int rc, *ok_clean, *ok_tampered;
    long mask = 0, got = 0;
    uint8_t seed[32];

    if (argc >= 2 && strcmp(argv[1], "sizes") == 0) return sizes();
    if (argc == 6 && strcmp(argv[1], "bench") == 0)
        return bench((size_t)strtoul(argv[2], NULL, 10), (size_t)strtoul(argv[3], NULL, 10), (size_t)strtoul(argv[4], NULL, 10), (size_t)strtoul(argv[5], NULL, 10));
    if (argc == 8 && strcmp(argv[1], "reject") == 0) return reject(argv);
    if (argc == 5 && strcmp(argv[1], "comm") == 0) return comm_mode(argv[2], atoi(argv[3]), atoi(argv[4]));
    if (argc == 7 && strcmp(argv[1], "split") == 0) return split_mode(argv);
    if (argc != 8 || strcmp(argv[1], "run") != 0) {
        fprintf(stderr, "usage: fl_round sizes | fl_round bench <d> <prove_range> <n_partition> <iterations> | fl_round run|reject <d> <prove_range> <n_partition> <n_clients> <n_devices> <out-file> | fl_round split <d> <prove_range> <n_partition> <n_devices> <out-file> | fl_round comm <id-file> <rank> <world>\n");
        return 2;
    }
    setup_round(&R, argv);
    prove_round(&R);

    /* server role: one call for the round, its clients spread over the devices by the library */
    for (i = 0; i < R.n_devices; i++) mask |= 1L << i;
    rc = rofl_set_device(0);
    if (rc) die("rofl_set_device(0)", rc);
    rc = rofl_set_option("devices", mask);
    if (rc) die("rofl_set_option(devices)", rc);
    rc = rofl_set_option("verify_batch", 2);
    if (rc) die("rofl_set_option(verify_batch)", rc);
    if (rofl_get_option("devices", &got) || got != mask) die("rofl_get_option(devices)", -1);
    memset(seed, 0x5a, sizeof seed);
    ok_clean = (int *)calloc(R.n_clients, sizeof(int));
    ok_tampered = (int *)calloc(R.n_clients, sizeof(int));
    rc = rofl_verify_rangeproof_batch(R.n_clients, (const uint8_t *const *)R.proofs, R.proof_len, R.n_proofs,
                                      (const uint8_t *const *)R.commits, R.d, R.prove_range, FP_BITS, FP_FRAC, seed, ok_clean);
    if (rc) die("rofl_verify_rangeproof_batch", rc);
    if (R.n_clients > 1) {
        uint8_t *bad = (uint8_t *)malloc(R.n_proofs * R.proof_len);
        const uint8_t **set = (const uint8_t **)malloc(R.n_clients * sizeof *set);
        memcpy(bad, R.proofs[1], R.n_proofs * R.proof_len);
        bad[4 * 32 + 5] ^= 0x01;                                            /* t_x of chunk 0 */
        for (i = 0; i < R.n_clients; i++) set[i] = i == 1 ? bad : R.proofs[i];
        rc = rofl_verify_rangeproof_batch(R.n_clients, set, R.proof_len, R.n_proofs, (const uint8_t *const *)R.commits, R.d, R.prove_range,
                                          FP_BITS, FP_FRAC, seed, ok_tampered);
        if (rc) die("rofl_verify_rangeproof_batch (tampered)", rc);
        free(bad);
        free(set);
    }

    f = fopen(argv[7], "wb");
    if (!f) { perror(argv[7]); return 1; }
    write_head(f, &R);
    for (i = 0; i < R.n_clients; i++) {
        fwrite(R.values[i], sizeof(float), R.d, f);
        fwrite(R.blindings[i], 32, R.d, f);
        fwrite(R.nonces[i].seed, 1, 32, f);
        fwrite(R.proofs[i], R.proof_len, R.n_proofs, f);
        fwrite(R.commits[i], 32, R.d, f);
    }
    fwrite(ok_clean, sizeof(int), R.n_clients, f);
    fwrite(ok_tampered, sizeof(int), R.n_clients, f);
    fclose(f);
    for (i = 0; i < R.n_clients; i++) printf("client %zu: clean %d tampered %d\n", i, ok_clean[i], ok_tampered[i]);
    return 0;
}
