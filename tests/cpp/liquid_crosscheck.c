/* Cross-check against a REAL liquid-dsp, compiled and run only where one is installed (tests/test_libliquid_crosscheck.py;
 * SURVEY.md section 8(c)).  Two directions:
 *   liquid_crosscheck <iqfile>             own TX -> liquid RX (below)
 *   liquid_crosscheck tx <outfile> <seed>  liquid TX -> own RX: liquid's own flexframegen, with the reference's properties
 *       (/root/reference/lib/flex_tx_impl.cc:51-59: CRC-24, 14 zero header bytes; here PSK4 / CONV_V27 / none =
 *       flex_tx::make(1, 1, 0)), writes three frames (payload lengths 1024, 64, 300; bytes from the LCG
 *       b = (seed = seed * 1103515245 + 12345) >> 16) with 600 zero samples around them, exactly as send_pkt does
 *       (assemble, getframelen, write_samples: :198-201); the test runs the file through this repo's receiver -- the
 *       direction that decides whether the drop-in can receive a real liquid transmitter.
 * Reads an IQ file of interleaved float32 (re, im) -- frames made by this repo's generator -- and runs
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

static int make_frames(const char *path, unsigned seed)
{
    static const unsigned lens[3] = { 1024, 64, 300 };
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    flexframegenprops_s props;
    flexframegenprops_init_default(&props);
    props.check = LIQUID_CRC_24; props.fec0 = LIQUID_FEC_CONV_V27; props.fec1 = LIQUID_FEC_NONE; props.mod_scheme = LIQUID_MODEM_PSK4;
    flexframegen fg = flexframegen_create(&props);
    unsigned char header[14] = { 0 }, payload[1024];
    float complex gap[600] = { 0 };
    fwrite(gap, sizeof(float complex), 400, f);
    for (int k = 0; k < 3; k++) {
        for (unsigned i = 0; i < lens[k]; i++) { seed = seed * 1103515245u + 12345u; payload[i] = (unsigned char)(seed >> 16); }
        flexframegen_assemble(fg, header, payload, lens[k]);
        unsigned int n = flexframegen_getframelen(fg);
        float complex *buf = (float complex *)malloc(n * sizeof(float complex));
        flexframegen_write_samples(fg, buf, n);
        fwrite(buf, sizeof(float complex), n, f);
        fwrite(gap, sizeof(float complex), 600, f);
        free(buf);
    }
    flexframegen_destroy(fg);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (argc >= 4 && argv[1][0] == 't' && argv[1][1] == 'x') return make_frames(argv[2], (unsigned)strtoul(argv[3], NULL, 10));
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
