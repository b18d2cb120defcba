// fx_blocks.hpp -- C++ block shells with the reference's class names, factories and work() signature, hosted on
// the C ABI of libfxrx.so (include/fxrx.h).  GNU Radio 3.7, Boost and pmt are absent from the build image, so the
// few runtime types the reference's headers take from them are declared here in the smallest form that keeps the
// *shape* of the interface (include/liquiddsp/flex_rx.h:40-50, frame_detector_cc.h:39-49, flex_tx.h:39-52):
//
//   gr_complex                         std::complex<float>
//   gr_vector_const_void_star / ..._void_star
//   sptr                               std::shared_ptr (boost::shared_ptr in GR 3.7)
//   message ports                      message_port_register_out / message_port_pub with a tiny variant `msg_t`
//                                      (c32vector, u8vector, dict<string,long>) instead of pmt::pmt_t
//
// With a real GNU Radio these classes would derive from gr::sync_block and publish pmt values; nothing else changes
// (INTEGRATION.md).  Header-only on purpose: a host application needs only this file and -lfxrx.
#pragma once
#include <complex>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <iostream>
#include <cstring>
#include "../../../include/fxrx.h"

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {
namespace liquiddsp {

// what the reference carries in pmt values on its message ports
struct msg_t {
    std::vector<gr_complex> c32;            // pmt::init_c32vector
    std::vector<uint8_t> u8;                // pmt::init_u8vector
    std::map<std::string, long> dict;       // pmt::make_dict of from_long
};
typedef std::function<void(const std::string &port, const msg_t &)> msg_sink_t;

class block_base {
public:
    explicit block_base(const std::string &name) : d_name(name) {}
    virtual ~block_base() {}
    const std::string &name() const { return d_name; }
    void set_msg_sink(msg_sink_t s) { d_sink = s; }
    const std::vector<std::string> &message_ports_out() const { return d_ports; }
    int output_multiple() const { return d_output_multiple; }
protected:
    void message_port_register_out(const std::string &p) { d_ports.push_back(p); }
    void message_port_pub(const std::string &p, const msg_t &m) { if (d_sink) d_sink(p, m); }
    void set_output_multiple(int m) { d_output_multiple = m; }
private:
    std::string d_name; std::vector<std::string> d_ports; msg_sink_t d_sink; int d_output_multiple = 1;
};

// ------------------------------------------------------------------ flex_rx (lib/flex_rx_impl.cc)
class flex_rx : public block_base {
public:
    typedef std::shared_ptr<flex_rx> sptr;
    static sptr make() { return sptr(new flex_rx()); }                       // include/liquiddsp/flex_rx.h:50
    ~flex_rx() { flexframesync_destroy(d_fs); }                              // lib/flex_rx_impl.cc:71

    // lib/flex_rx_impl.cc:203-254: the reference's loop as it stands -- 256 samples per call, at most one frame drained after each
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &)
    {
        fx_complex *in = (fx_complex *)input_items[0];                       // :208
        if (noutput_items % d_inbuf_len != 0) throw std::invalid_argument("flex_rx: noutput_items must be a multiple of 256");   // :210
        int num_items = 0;
        while (num_items < noutput_items) {                                  // :212
            flexframesync_execute(d_fs, in, d_inbuf_len);                    // :213
            num_items += d_inbuf_len; in += d_inbuf_len;                     // :214-215
            if (d_info._new_payload) { publish(); d_info._new_payload = false; }   // :216-250
        }
        return noutput_items;                                                // :253
    }
    // (additive; a GNU Radio shell would call it from stop()): run what is still queued, publish every frame that completes
    void flush()
    {
        fxrx_sync_flush(d_fs);
        while (fxrx_sync_pending(d_fs)) {
            flexframesync_execute(d_fs, nullptr, 0);
            if (d_info._new_payload) { publish(); d_info._new_payload = false; }
        }
    }
    flexframesync handle() const { return d_fs; }
    unsigned long num_frames() const { return d_info._num_frames; }

private:
    struct packet_info {                                                     // lib/flex_rx_impl.h:27-37
        unsigned char *_header = nullptr; int _header_valid = 0; unsigned char *_payload = nullptr; unsigned int _payload_len = 0;
        framesyncstats_s _stats{}; unsigned int _num_frames = 0; int _payload_valid = 0; bool _new_payload = false;
    };
    static const int d_inbuf_len = 256;                                      // lib/flex_rx_impl.h:47
    flexframesync d_fs; packet_info d_info;

    flex_rx() : block_base("flex_rx")
    {
        d_fs = flexframesync_create(callback, (void *)&d_info);             // :49
        if (!d_fs) throw std::runtime_error(std::string("flex_rx: ") + fxrx_last_error());
        set_output_multiple(d_inbuf_len);                                    // :50
        message_port_register_out("constellation");                          // :61-63
        message_port_register_out("payload_data");
        message_port_register_out("packet_info");
    }
    static int callback(unsigned char *h, int hv, unsigned char *p, unsigned int plen, int pv, framesyncstats_s st, void *ud)   // :181-201
    {
        packet_info *info = (packet_info *)ud;
        info->_payload = p; info->_header = h; info->_header_valid = hv; info->_stats = st; info->_payload_valid = pv;
        info->_payload_len = plen; info->_num_frames++; info->_new_payload = true;
        return 0;
    }
    void publish()
    {
        msg_t c;                                                             // constellation regardless of validity, :217-221
        if (d_info._stats.framesyms && d_info._stats.num_framesyms)
            c.c32.assign((gr_complex *)d_info._stats.framesyms, (gr_complex *)d_info._stats.framesyms + d_info._stats.num_framesyms);
        message_port_pub("constellation", c);
        if (!d_info._header_valid) return;                                   // :223
        msg_t pl; pl.u8.assign(d_info._payload, d_info._payload + d_info._payload_len);
        message_port_pub("payload_data", pl);                                // :224-229
        msg_t pi;                                                            // :236-247
        pi.dict["header_valid"] = 1; pi.dict["payload_valid"] = d_info._payload_valid;
        pi.dict["modulation"] = fxrx_mod_to_index(d_info._stats.mod_scheme);
        pi.dict["inner_code"] = fxrx_inner_to_index(d_info._stats.fec0);
        pi.dict["outer_code"] = fxrx_outer_to_index(d_info._stats.fec1);
        message_port_pub("packet_info", pi);
    }
};

// ------------------------------------------------------------------ frame_detector_cc (lib/frame_detector_cc_impl.cc)
class frame_detector_cc : public block_base {
public:
    typedef std::shared_ptr<frame_detector_cc> sptr;
    static sptr make() { return sptr(new frame_detector_cc()); }            // include/liquiddsp/frame_detector_cc.h:49
    ~frame_detector_cc() { fxrx_destroy(d_ctx); }

    // lib/frame_detector_cc_impl.cc:66-97: detect on the whole block, count, pass the samples through
    int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items)
    {
        const gr_complex *in = (const gr_complex *)input_items[0];
        gr_complex *out = (gr_complex *)output_items[0];
        const void *p = in; uint64_t n = (uint64_t)noutput_items;
        int nd = fxrx_process(d_ctx, &p, &n, 0);
        if (nd < 0) throw std::runtime_error(std::string("frame_detector_cc: ") + fxrx_last_error());
        for (int i = 0; i < nd; i++) {
            if (d_verbose) std::cout << "Detected " << d_num_frames << " frames!" << std::endl;   // :79
            d_num_frames++;                                                  // :80
        }
        std::memcpy(out, in, sizeof(gr_complex) * (size_t)noutput_items);    // :82
        return noutput_items;                                                // :96
    }
    unsigned long num_frames() const { return d_num_frames; }
    void set_verbose(bool v) { d_verbose = v; }

private:
    fxrx_ctx *d_ctx; unsigned long d_num_frames = 0; bool d_verbose = false;
    frame_detector_cc() : block_base("frame_detector_cc")
    {
        fxrx_config cfg{}; cfg.device = 0; cfg.mode = FXRX_MODE_DETECTOR; cfg.n_streams = 1; cfg.threshold = 0.45f;   // :55
        d_ctx = fxrx_create(&cfg);
        if (!d_ctx) throw std::runtime_error(std::string("frame_detector_cc: ") + fxrx_last_error());
    }
};

// ------------------------------------------------------------------ flex_tx (lib/flex_tx_impl.cc)
class flex_tx : public block_base {
public:
    typedef std::shared_ptr<flex_tx> sptr;
    static sptr make(unsigned int modulation, unsigned int inner_code, unsigned int outer_code)      // include/liquiddsp/flex_tx.h:49
    { return sptr(new flex_tx(modulation, inner_code, outer_code)); }
    ~flex_tx() { flexframegen_destroy(d_fg); }                               // :72

    void set_modulation(unsigned int m) { int v = fxrx_mod_from_index((int)m); d_props.mod_scheme = v < 0 ? (unsigned)fxrx_mod_from_index(0) : (unsigned)v; }   // :75-116
    void set_inner_code(unsigned int c) { int v = fxrx_inner_from_index((int)c); d_props.fec0 = v < 0 ? 1u : (unsigned)v; }                                    // :118-146
    void set_outer_code(unsigned int c) { int v = fxrx_outer_from_index((int)c); d_props.fec1 = v < 0 ? 1u : (unsigned)v; }                                    // :148-181
    void configure(const std::map<std::string, long> &cfg)                   // :183-189
    {
        auto it = cfg.find("modulation"); if (it != cfg.end()) set_modulation((unsigned)it->second);
        it = cfg.find("inner_code"); if (it != cfg.end()) set_inner_code((unsigned)it->second);
        it = cfg.find("outer_code"); if (it != cfg.end()) set_outer_code((unsigned)it->second);
        if (flexframegen_setprops(d_fg, &d_props) != 0) throw std::invalid_argument("flex_tx: unsupported configuration");
    }
    void send_pkt(const std::vector<uint8_t> &bytes)                         // :191-209
    {
        flexframegen_assemble(d_fg, d_header, bytes.data(), (unsigned)bytes.size());
        unsigned frame_len = flexframegen_getframelen(d_fg);
        msg_t m; m.c32.resize(frame_len);
        flexframegen_write_samples(d_fg, (fx_complex *)m.c32.data(), frame_len);
        message_port_pub("pdus", m);
        d_num_frames++;
    }
    int work(int, gr_vector_const_void_star &, gr_vector_void_star &) { throw std::runtime_error("This is not a stream block."); }   // :211-218

private:
    flexframegenprops_s d_props; flexframegen d_fg; unsigned char d_header[14]; unsigned long d_num_frames = 0;
    flex_tx(unsigned int modulation, unsigned int inner_code, unsigned int outer_code) : block_base("flex_tx")
    {
        flexframegenprops_init_default(&d_props);                            // :51
        d_props.check = 5;                                                   // LIQUID_CRC_24, :52
        set_inner_code(inner_code); set_outer_code(outer_code); set_modulation(modulation);
        d_fg = flexframegen_create(&d_props);                                // :56
        if (!d_fg) throw std::invalid_argument("flex_tx: unsupported configuration");
        message_port_register_out("pdus");
        std::memset(d_header, 0, 14);                                        // :58-59
    }
};

}  // namespace liquiddsp
}  // namespace gr
