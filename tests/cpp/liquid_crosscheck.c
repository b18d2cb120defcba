/* Cross-check against a REAL liquid-dsp, compiled and run only where one is installed (tests/test_libliquid_crosscheck.py;
 * SURVEY.md section 8(c)).  Reads an IQ file of interleaved float32 (re, im) -- frames made by this repo's generator -- and runs
 * liquid's own flexframesync over it in 256-sample calls, as /root/reference/lib/flex_rx_impl.cc:212-215 does; prints one
 * line per frame: "F <header_valid> <payload_valid> <payload_len> <hex payload>".  The test compares them with the bytes that
 * were sent.  (It has never been compiled in the build image, which has no liquid-dsp: the test skips, loudly, when the
 * header is missing or this file does not compile against the installed version.) */
#include <stdio.h>
#include <stdlib.h>
#include <complex.h>
#include <liquid/liquid.h>

static int on_frame(unsigned char *header, int header_valid, unsigned char *payload, unsigned int payload_len, int payload_valid,
                    framesyncstats_s stats, void *userdata)
{
    (void)header; (void)stats; (void)userdata;
    printf("F %d %d %u ", header_valid, payload_valid, payload_len);
    if (header_valid) for (unsigned int i = 0; i < payload_len; i++) printf("%02x", payload[i]);
    printf("\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END); long nbytes = ftell(f); fseek(f, 0, SEEK_SET);
    size_t n = (size_t)nbytes / (2 * sizeof(float));
    float complex *x = (float complex *)malloc((n + 256) * sizeof(float complex));
    if (fread(x, 2 * sizeof(float), n, f) != n) return 2;
    fclose(f);
    for (size_t i = n; i < n + 256; i++) x[i] = 0.0f;
    flexframesync fs = flexframesync_create(on_frame, NULL);
    for (size_t p = 0; p + 256 <= n + 256; p += 256) flexframesync_execute(fs, x + p, 256);
    flexframesync_destroy(fs);
    free(x);
    return 0;
}
