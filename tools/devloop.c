/* readStream / writeStream per-call cost from a C caller (no Python in the loop), through the flat C view of the
 * Device (include/sx_device.h): what the plugin itself costs per call at the reference's block sizes.
 * Built by sxxcvr_amd/build.py as sxxcvr_amd/lib/sx_devloop (gcc -O2 -Iinclude tools/devloop.c -lSXSupport); bench.py runs
 * it as a child process for through_device.c_caller.  Errors go to stdout as text and the exit status is 1. */
#include <sx_device.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}

int main(int argc, char **argv)
{
    /* --json: one JSON object on stdout (bench.py's through_device.c_caller) */
    const int json = argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'j';
    const size_t sizes[] = {256, 1024, 4096, 8192, 65536};
    if (json) printf("{");
    for (unsigned k = 0; k < sizeof(sizes) / sizeof(sizes[0]); ++k) {
        const size_t blk = sizes[k];
        sx_device *dev = sx_device_make("driver=sx,clock=virtual");
        if (!dev) { printf("make: %s\n", sx_device_last_error()); return 1; }
        sx_device_set_sample_rate(dev, 1, 0, 600000.0);
        sx_device_set_sample_rate(dev, 0, 0, 600000.0);
        char args[64];
        snprintf(args, sizeof(args), "period=%zu", blk > 65536 ? (size_t)65536 : blk);
        const size_t ch0 = 0;
        sx_stream *rx = sx_device_setup_stream(dev, 1, "CF32", &ch0, 1, args);
        sx_stream *tx = sx_device_setup_stream(dev, 0, "CF32", &ch0, 1, args);
        if (!rx || !tx) { printf("setup: %s\n", sx_device_last_error()); return 1; }
        sx_device_activate_stream(dev, rx, 0, 0, 0);
        sx_device_activate_stream(dev, tx, 0, 0, 0);
        float *buf = calloc(2 * blk, sizeof(float));
        void *buffs[1] = {buf};
        const void *cbuffs[1] = {buf};
        int flags = 0;
        long long t_ns = 0;
        const int n = blk <= 8192 ? 20000 : 2000;
        for (int i = 0; i < 100; ++i) sx_device_read_stream(dev, rx, buffs, blk, &flags, &t_ns, 100000);
        double t0 = now();
        for (int i = 0; i < n; ++i)
            if (sx_device_read_stream(dev, rx, buffs, blk, &flags, &t_ns, 100000) != (int)blk) { printf("read: %s\n", sx_device_last_error()); return 1; }
        const double rx_us = (now() - t0) / n * 1e6;
        for (int i = 0; i < 100; ++i) { flags = 0; sx_device_write_stream(dev, tx, cbuffs, blk, &flags, 0, 100000); }
        t0 = now();
        for (int i = 0; i < n; ++i) {
            flags = 0;
            if (sx_device_write_stream(dev, tx, cbuffs, blk, &flags, 0, 100000) != (int)blk) { printf("write: %s\n", sx_device_last_error()); return 1; }
        }
        const double tx_us = (now() - t0) / n * 1e6;
        if (json)
            printf("%s\"%zu\": {\"readStream_us_per_call\": %.3f, \"readStream_out_MS/s\": %.1f, \"writeStream_us_per_call\": %.3f, "
                   "\"writeStream_in_MS/s\": %.1f, \"calls\": %d}", k ? ", " : "", blk, rx_us, blk / rx_us, tx_us, blk / tx_us, n);
        else
            printf("block %6zu: readStream %.2f us/call (%.1f MS/s out) | writeStream %.2f us/call (%.1f MS/s in)\n", blk, rx_us,
                   blk / rx_us, tx_us, blk / tx_us);
        free(buf);
        sx_device_unmake(dev);
    }
    if (json) printf("}\n");
    return 0;
}
