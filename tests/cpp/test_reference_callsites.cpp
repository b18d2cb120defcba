// Source-level drop-in proof: the liquid-dsp calls that gr::liquiddsp's three block implementations make, spelled with
// the argument types the reference uses (gr_complex*, gr_complex by value, framesyncstats_s by value, struct fields
// assigned from LIQUID_* enumerators), compiled as C++ against include/liquid/liquid.h and linked against libfxrx.so only.
//
//   rx_calls   <- /root/reference/lib/flex_rx_impl.cc:49 (create), :71 (destroy), :181-201 (callback), :213 (execute),
//                 :216-250 (what work() reads after the call); struct packet_info as /root/reference/lib/flex_rx_impl.h:27-37
//   det_calls  <- /root/reference/lib/frame_detector_cc_impl.cc:46-55 (p/n sequence, create_linear, set_threshold),
//                 :63 (destroy), :77 (execute per sample).  The reference writes `d_preamble_pn[i].real() = ...` (:49-50),
//                 which only compiles against a pre-C++11 std::complex; here the same values go through the C++11 setters.
//   tx_calls   <- /root/reference/lib/flex_tx_impl.cc:51-56 (props, create), :72 (destroy), :188 (setprops),
//                 :198-201 (assemble, getframelen, write_samples into a std::vector<gr_complex>)
//
// GNU Radio, pmt and Boost are absent from the build image: gr_complex is declared as GNU Radio declares it
// (std::complex<float>) and the PMT publication is replaced by plain std::vector copies.  Nothing here includes fxrx.h
// directly or casts a sample pointer: if this file compiles, the reference's call sites do.
//
// Built in the CPU suite (tests/test_cabi.py: compile + link), run on the GPU (tests/test_gpu_parity.py).
#include <complex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <liquid/liquid.h>

typedef std::complex<float> gr_complex;          // gnuradio/gr_complex.h

struct packet_info {                             // lib/flex_rx_impl.h:27-37
    unsigned char *_header;
    int _header_valid;
    unsigned char *_payload;
    unsigned int _payload_len;
    framesyncstats_s _stats;
    gr_complex *_frame_symbols;
    unsigned int _num_frames;
    int _payload_valid;
    bool _new_payload;
};

struct rx_calls {
    flexframesync d_fs;
    packet_info *d_info;
    static const unsigned int d_inbuf_len = 256;                              // lib/flex_rx_impl.h:47
    std::vector<std::vector<unsigned char> > payloads; std::vector<size_t> constellations; std::vector<int> valid;
    std::vector<unsigned> mods, fec0s, fec1s;

    static int callback(unsigned char *_header, int _header_valid, unsigned char *_payload, unsigned int _payload_len,
                        int _payload_valid, framesyncstats_s _stats, void *_userdata)
    {
        packet_info *info = (packet_info *)_userdata;
        info->_payload = _payload;
        info->_header = _header;
        info->_header_valid = _header_valid;
        info->_stats = _stats;
        info->_payload_valid = _payload_valid;
        info->_payload_len = _payload_len;
        info->_frame_symbols = _stats.framesyms;                              // gr_complex* <- liquid_float_complex*
        info->_num_frames++;
        info->_new_payload = true;
        return 0;                                                             // (the reference falls off the end here)
    }
    rx_calls()
    {
        d_info = (packet_info *)calloc(1, sizeof(packet_info));               // (malloc, not zeroed, in the reference)
        d_fs = flexframesync_create(callback, (void *)d_info);
        d_info->_header_valid = 0;
        d_info->_payload_valid = 0;
    }
    ~rx_calls() { if (d_fs) flexframesync_destroy(d_fs); free(d_info); }
    int work(int noutput_items, gr_complex *in)
    {
        unsigned int num_items = 0;
        while (num_items < (unsigned)noutput_items) {
            flexframesync_execute(d_fs, in, d_inbuf_len);
            num_items += d_inbuf_len;
            in += d_inbuf_len;
            if (d_info->_new_payload) {
                std::vector<gr_complex> constellation(d_info->_stats.framesyms, d_info->_stats.framesyms + d_info->_stats.num_framesyms);
                constellations.push_back(constellation.size());
                if (d_info->_header_valid) {
                    payloads.push_back(std::vector<unsigned char>(d_info->_payload, d_info->_payload + d_info->_payload_len));
                    valid.push_back(d_info->_payload_valid);
                    mods.push_back(d_info->_stats.mod_scheme); fec0s.push_back(d_info->_stats.fec0); fec1s.push_back(d_info->_stats.fec1);
                }
                d_info->_new_payload = false;
            }
        }
        return (int)num_items;
    }
};

struct det_calls {
    qdetector_cccf d_detector;
    const unsigned int d_k = 2;
    const unsigned int d_m = 7;
    const float d_beta = 0.3;
    gr_complex *d_preamble_pn;
    unsigned long int d_num_frames;
    det_calls() : d_num_frames(0)
    {
        d_preamble_pn = (gr_complex *)malloc(64 * sizeof(gr_complex));
        msequence ms = msequence_create(7, 0x0089, 1);
        for (unsigned int i = 0; i < 64; i++) {
            d_preamble_pn[i].real(msequence_advance(ms) ? M_SQRT1_2 : -M_SQRT1_2);
            d_preamble_pn[i].imag(msequence_advance(ms) ? M_SQRT1_2 : -M_SQRT1_2);
        }
        msequence_destroy(ms);
        d_detector = qdetector_cccf_create_linear(d_preamble_pn, 64, LIQUID_FIRFILT_ARKAISER, d_k, d_m, d_beta);
        qdetector_cccf_set_threshold(d_detector, 0.45);
    }
    ~det_calls() { qdetector_cccf_destroy(d_detector); free(d_preamble_pn); }
    int work(int noutput_items, const gr_complex *in, gr_complex *out)
    {
        for (unsigned long int i = 0; i < (unsigned long)noutput_items; i++) {
            void *v = qdetector_cccf_execute(d_detector, in[i]);
            if (v != NULL) d_num_frames++;
            out[i] = in[i];
        }
        return noutput_items;
    }
};

struct tx_calls {
    flexframegenprops_s d_fgprops;
    flexframegen d_fg;
    unsigned char *d_header;
    tx_calls()
    {
        flexframegenprops_init_default(&d_fgprops);
        d_fgprops.check = LIQUID_CRC_24;
        d_fgprops.fec0 = LIQUID_FEC_CONV_V27;                                 // set_inner_code(1), lib/flex_tx_impl.cc:124-126
        d_fgprops.fec1 = LIQUID_FEC_NONE;                                     // set_outer_code(0), :152-154
        d_fgprops.mod_scheme = LIQUID_MODEM_PSK4;                             // set_modulation(1), :81-83
        d_fg = flexframegen_create(&d_fgprops);
        d_header = (unsigned char *)malloc(14 * sizeof(unsigned char));
        memset(d_header, 0, 14);
    }
    ~tx_calls() { flexframegen_destroy(d_fg); free(d_header); }
    void configure(unsigned mod, unsigned inner, unsigned outer)
    {
        d_fgprops.mod_scheme = mod; d_fgprops.fec0 = inner; d_fgprops.fec1 = outer;
        flexframegen_setprops(d_fg, &d_fgprops);
    }
    std::vector<gr_complex> send_pkt(std::vector<unsigned char> payload)
    {
        flexframegen_assemble(d_fg, d_header, &payload.front(), payload.size());
        unsigned int frame_len = flexframegen_getframelen(d_fg);
        std::vector<gr_complex> vec(frame_len);
        flexframegen_write_samples(d_fg, &vec.front(), frame_len);
        return vec;
    }
};

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "--link-only")) { std::printf("LINKED\n"); return 0; }
    std::mt19937 rng(0xCA11);
    std::vector<std::vector<unsigned char> > sent; std::vector<gr_complex> x;
    {
        tx_calls tx;
        for (int f = 0; f < 5; f++) {
            if (f == 3) tx.configure(LIQUID_MODEM_QAM16, LIQUID_FEC_CONV_V27P23, LIQUID_FEC_GOLAY2412);
            std::vector<unsigned char> pl(f == 4 ? 333 : 1024); for (size_t i = 0; i < pl.size(); i++) pl[i] = (unsigned char)(rng() & 0xff);
            sent.push_back(pl);
            std::vector<gr_complex> fr = tx.send_pkt(pl);
            if (f == 0 && fr.size() != 17066) { std::printf("FAIL frame length %zu\n", fr.size()); return 1; }
            x.insert(x.end(), fr.begin(), fr.end()); x.insert(x.end(), 256, gr_complex(0, 0));
        }
    }
    std::normal_distribution<float> nd(0.0f, 0.04f);
    for (size_t n = 0; n < x.size(); n++) x[n] = x[n] * std::polar(1.0f, -0.015f * (float)n + 0.2f) + gr_complex(nd(rng), nd(rng));
    x.insert(x.end(), 70000, gr_complex(0, 0));                               // the per-sample detector drop-in searches 64 Ki-sample blocks: push the last one through
    while (x.size() % 256) x.push_back(gr_complex(0, 0));

    rx_calls rx;
    if (!rx.d_fs) { std::fprintf(stderr, "flexframesync_create failed: %s\n", fxrx_last_error()); return 2; }
    int consumed = rx.work((int)x.size(), x.data());
    fxrx_sync_flush(rx.d_fs);                                                 // the stream ends here: the drop-in runs whole blocks, push the last one through
    gr_complex zero[256] = {};
    for (int k = 0; k < 64 && fxrx_sync_pending(rx.d_fs); k++) rx.work(256, zero);   // one frame per call, as the reference's loop drains them

    det_calls det;
    std::vector<gr_complex> y(x.size());
    det.work((int)x.size(), x.data(), y.data());

    bool ok = consumed == (int)x.size() && rx.payloads == sent && rx.constellations.size() == 5 && rx.constellations[0] == 8224;
    for (size_t i = 0; i < rx.valid.size(); i++) ok = ok && rx.valid[i] == 1;
    ok = ok && rx.mods.size() == 5 && rx.mods[0] == LIQUID_MODEM_PSK4 && rx.fec0s[0] == LIQUID_FEC_CONV_V27 && rx.fec1s[0] == LIQUID_FEC_NONE &&
         rx.mods[3] == LIQUID_MODEM_QAM16 && rx.fec0s[3] == LIQUID_FEC_CONV_V27P23 && rx.fec1s[3] == LIQUID_FEC_GOLAY2412;
    ok = ok && det.d_num_frames >= 5 && memcmp(x.data(), y.data(), x.size() * sizeof(gr_complex)) == 0;
    ok = ok && rx.d_info->_num_frames == 5 && fxrx_sync_errors(rx.d_fs) == 0;
    std::printf("payloads %zu/%zu constellations %zu detector %lu -> %s\n", rx.payloads.size(), sent.size(), rx.constellations.size(),
                det.d_num_frames, ok ? "PASS" : "FAIL");
    return ok ? 0 : 1;
}
