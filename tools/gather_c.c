/* The sharded path driven from plain C over include/sxfir.h: every rank decimates its 8 channels (BASELINE config 4's
 * per-GPU shard) and the decimated blocks are gathered to rank 0 with sxfir_comm_gather (RCCL over xGMI), no Python,
 * no torch.distributed.  Two forms:
 *
 *   sx_gather_c ranks <nranks> <rank> <idfile> [log2_n_per_channel [steps]]
 *       one process per GPU (device = rank mod visible GPUs); rank 0 writes the RCCL id to <idfile>, the others wait
 *       for the file -- the "any means it has" of sxfir_comm_unique_id.
 *   sx_gather_c all <ndev> [log2_n_per_channel [steps]]
 *       one process drives ndev GPUs (ndev communicators from sxfir_comm_init_all, sxfir_comm_gather_all).
 *
 * The root checks that every rank's block in the gathered buffer carries that rank's channels: each rank also
 * sends an 8-byte sum of its output words behind its block, and the root recomputes the sums from what arrived.
 * Prints one line of JSON; exit status 1 on any error.
 * Built by sxxcvr_amd/build.py as sxxcvr_amd/lib/sx_gather_c (gcc -O2 -Iinclude tools/gather_c.c -lsxfir). */
#include <sxfir.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NCH 8
#define CK(x) do { int rc_ = (x); if (rc_) { printf("%s: %d %s\n", #x, rc_, sxfir_last_error()); return 1; } } while (0)

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

static uint64_t wordsum(const uint32_t *w, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i) s += w[i];
    return s;
}

struct shard {
    sxfir_plan *plan;
    void *in, *out, *stream;
    uint64_t sum;          /* of this shard's output words */
};

/* plan + buffers of one rank on the calling thread's current GPU; one decimation pass; host sum of its output */
static int shard_make(struct shard *s, int rank, size_t n_in, const float *taps)
{
    const size_t n_out = n_in / 4;
    CK(sxfir_stream_create(&s->stream));
    CK(sxfir_create(&s->plan, SXFIR_DECIMATE, taps, 128, 4, NCH, SXFIR_CF32, -1));
    CK(sxfir_malloc(&s->in, 8 * n_in * NCH));
    CK(sxfir_malloc(&s->out, 8 * n_out * NCH));
    CK(sxfir_synth_fill(s->in, n_in, n_in, NCH, 0x51255, (uint32_t)(NCH * rank), 0, SXFIR_CF32, s->stream));
    size_t got = 0;
    CK(sxfir_decimate(s->plan, s->in, n_in, n_in, s->out, n_out, &got, s->stream));
    if (got != n_out) { printf("decimate produced %zu of %zu\n", got, n_out); return 1; }
    uint32_t *h = malloc(8 * n_out * NCH);
    CK(sxfir_memcpy_d2h(h, s->out, 8 * n_out * NCH, s->stream));
    CK(sxfir_stream_sync(s->stream));
    s->sum = wordsum(h, 2 * n_out * NCH);
    free(h);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 3) { printf("usage: %s ranks <n> <rank> <idfile> [log2n [steps]] | all <ndev> [log2n [steps]]\n", argv[0]); return 1; }
    const int all = strcmp(argv[1], "all") == 0;
    const int nranks = atoi(argv[2]);
    const int rank = all ? 0 : atoi(argv[3]);
    const int a0 = all ? 3 : 5;
    const int log2n = argc > a0 ? atoi(argv[a0]) : 22;
    const int steps = argc > a0 + 1 ? atoi(argv[a0 + 1]) : 5;
    const size_t n_in = (size_t)1 << log2n, n_out = n_in / 4, bytes = 8 * n_out * NCH;
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) { printf("bad rank arguments\n"); return 1; }
    int ngpu = 0;
    CK(sxfir_device_count(&ngpu));
    float taps[128];
    CK(sxfir_design_lowpass(128, 4, 8.0, 1.0, taps));

    struct shard sh[64];
    sxfir_comm *comms[64];
    memset(sh, 0, sizeof sh);
    const int local = all ? nranks : 1;                      /* shards this process owns */
    if (all) {
        int devs[64];
        for (int i = 0; i < nranks; ++i) devs[i] = i % ngpu;
        if (nranks > ngpu) { printf("all: %d devices asked, %d visible\n", nranks, ngpu); return 1; }
        CK(sxfir_comm_init_all(comms, nranks, devs));
        for (int i = 0; i < nranks; ++i) {
            CK(sxfir_set_device(devs[i]));
            if (shard_make(&sh[i], i, n_in, taps)) return 1;
        }
    } else {
        CK(sxfir_set_device(rank % ngpu));
        unsigned char id[SXFIR_COMM_ID_BYTES];
        const char *idfile = argv[4];
        if (rank == 0) {
            char tmp[1024];
            CK(sxfir_comm_unique_id(id));
            snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { printf("cannot write %s\n", tmp); return 1; }
            fclose(f);
            if (rename(tmp, idfile)) { printf("cannot rename to %s\n", idfile); return 1; }
        } else {
            FILE *f = NULL;
            const struct timespec nap = {0, 10 * 1000 * 1000};
            for (int i = 0; i < 6000 && !(f = fopen(idfile, "rb")); ++i) nanosleep(&nap, NULL);
            if (!f || fread(id, 1, sizeof id, f) != sizeof id) { printf("no id in %s\n", idfile); return 1; }
            fclose(f);
        }
        CK(sxfir_comm_init_rank(&comms[0], id, nranks, rank, -1));
        if (shard_make(&sh[0], rank, n_in, taps)) return 1;
    }

    /* root's receive buffer: nranks blocks of `bytes`, then nranks 8-byte sums */
    void *recv = NULL, *sums_dev = NULL, *mysum_dev[64];
    const int root_here = all || rank == 0;
    if (root_here) {
        if (all) CK(sxfir_set_device(0 % ngpu));
        CK(sxfir_malloc(&recv, bytes * nranks));
        CK(sxfir_malloc(&sums_dev, 8 * (size_t)nranks));
    }
    for (int i = 0; i < local; ++i) {
        if (all) CK(sxfir_set_device(i % ngpu));
        CK(sxfir_malloc(&mysum_dev[i], 8));
        CK(sxfir_memcpy_h2d(mysum_dev[i], &sh[i].sum, 8, sh[i].stream));
        CK(sxfir_stream_sync(sh[i].stream));
    }

    const size_t chunk = 8 * n_out * 2;                      /* two channels per piece: four RCCL groups per gather */
    double best = 1e30, first = 0;
    for (int s = 0; s < steps + 1; ++s) {                    /* step 0 = warm-up (connection set-up) */
        const double t0 = now();
        if (all) {
            const void *sends[64]; void *streams[64]; const void *ssum[64];
            for (int i = 0; i < nranks; ++i) { sends[i] = sh[i].out; streams[i] = sh[i].stream; ssum[i] = mysum_dev[i]; }
            CK(sxfir_comm_gather_all(comms, nranks, sends, recv, bytes, bytes, 0, chunk, streams));
            CK(sxfir_comm_gather_all(comms, nranks, ssum, sums_dev, 8, 8, 0, 0, streams));
            for (int i = 0; i < nranks; ++i) CK(sxfir_stream_sync(sh[i].stream));
        } else {
            CK(sxfir_comm_gather(comms[0], sh[0].out, recv, bytes, bytes, 0, chunk, sh[0].stream));
            CK(sxfir_comm_gather(comms[0], mysum_dev[0], sums_dev, 8, 8, 0, 0, sh[0].stream));
            CK(sxfir_stream_sync(sh[0].stream));
        }
        const double dt = now() - t0;
        if (s == 0) first = dt; else if (dt < best) best = dt;
    }

    int bad = 0;
    if (root_here) {
        uint32_t *h = malloc(bytes);
        uint64_t sums[64];
        void *st = sh[0].stream;
        if (all) CK(sxfir_set_device(0 % ngpu));
        CK(sxfir_memcpy_d2h(sums, sums_dev, 8 * (size_t)nranks, st));
        for (int r = 0; r < nranks; ++r) {
            CK(sxfir_memcpy_d2h(h, (char *)recv + bytes * r, bytes, st));
            CK(sxfir_stream_sync(st));
            const uint64_t got = wordsum(h, 2 * n_out * NCH);
            if (got != sums[r] || (r < local && got != sh[r].sum)) { printf("rank %d's block: sum %llu, sent %llu\n", r, (unsigned long long)got, (unsigned long long)sums[r]); bad = 1; }
        }
        free(h);
        /* distinct channels => distinct sums: a block landing in the wrong slot cannot pass */
        for (int r = 1; r < nranks; ++r) if (sums[r] == sums[0]) { printf("ranks 0 and %d sent the same sum\n", r); bad = 1; }
        const double gb = (double)bytes * (nranks - 1) / 1e9;
        printf("{\"form\": \"%s\", \"nranks\": %d, \"gpus_visible\": %d, \"bytes_per_rank\": %zu, \"chunk_bytes\": %zu, \"gather_ms_best\": %.4f, "
               "\"gather_ms_first\": %.3f, \"root_inbound_GB/s\": %.1f, \"verified\": %s}\n", all ? "one process (init_all)" : "rank per process (init_rank)",
               nranks, ngpu, bytes, chunk, best * 1e3, first * 1e3, nranks > 1 ? gb / best : 0.0, bad ? "false" : "true");
    }
    for (int i = 0; i < local; ++i) {
        sxfir_comm_destroy(comms[i]);
        sxfir_destroy(sh[i].plan);
    }
    return bad;
}
