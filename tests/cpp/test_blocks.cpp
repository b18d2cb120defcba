// C++ loopback through the block shells: flex_tx -> channel -> flex_rx / frame_detector_cc (BASELINE config 1
// plumbing on the GPU path).  Built and run by tests/test_gpu_parity.py::test_cpp_block_shells_loopback.
#include <cstdio>
#include <random>
#include "../../gr-liquiddsp_amd/csrc/blocks/fx_blocks.hpp"
using namespace gr::liquiddsp;

int main()
{
    std::vector<std::vector<uint8_t>> sent; std::vector<gr_complex> x;
    flex_tx::sptr tx = flex_tx::make(1, 1, 0);                       // PSK4, CONV_V27, none
    std::mt19937 rng(0x5EED);
    tx->set_msg_sink([&](const std::string &port, const msg_t &m) {
        if (port != "pdus") return;
        x.insert(x.end(), m.c32.begin(), m.c32.end());
        x.insert(x.end(), 256, gr_complex(0, 0));
    });
    for (int f = 0; f < 6; f++) {
        std::vector<uint8_t> pl(1024); for (auto &b : pl) b = (uint8_t)(rng() & 0xff);
        sent.push_back(pl); tx->send_pkt(pl);
    }
    std::normal_distribution<float> nd(0.0f, 0.05f);
    for (size_t n = 0; n < x.size(); n++) x[n] = x[n] * std::polar(1.0f, 0.02f * (float)n + 0.5f) + gr_complex(nd(rng), nd(rng));
    while (x.size() % 256) x.push_back(gr_complex(0, 0));

    flex_rx::sptr rx = flex_rx::make();
    std::vector<std::vector<uint8_t>> got; int n_info = 0, n_const = 0; bool info_ok = true;
    rx->set_msg_sink([&](const std::string &port, const msg_t &m) {
        if (port == "payload_data") got.push_back(m.u8);
        if (port == "constellation") n_const += (m.c32.size() == 8224);
        if (port == "packet_info") {
            n_info++;
            info_ok &= m.dict.at("header_valid") == 1 && m.dict.at("payload_valid") == 1 && m.dict.at("modulation") == 1 &&
                       m.dict.at("inner_code") == 1 && m.dict.at("outer_code") == 0;
        }
    });
    gr_vector_const_void_star in(1); gr_vector_void_star out;
    for (size_t i = 0; i < x.size(); i += 8192) {
        int n = (int)std::min<size_t>(8192, x.size() - i);
        in[0] = x.data() + i;
        if (rx->work(n, in, out) != n) { std::printf("FAIL work return\n"); return 1; }
    }
    rx->flush();                                                     // (the stream ends here: run what is still queued)
    frame_detector_cc::sptr det = frame_detector_cc::make();
    std::vector<gr_complex> y(x.size()); gr_vector_void_star outs(1); outs[0] = y.data(); in[0] = x.data();
    det->work((int)x.size(), in, outs);
    bool pass = got == sent && n_info == 6 && n_const == 6 && info_ok && rx->output_multiple() == 256 &&
                std::memcmp(x.data(), y.data(), x.size() * sizeof(gr_complex)) == 0 && det->num_frames() >= 6;
    bool threw = false;
    try { tx->work(0, in, out); } catch (const std::runtime_error &) { threw = true; }
    std::printf("frames %zu/%zu info %d const %d detector %lu passthrough %s tx.work throws %d -> %s\n", got.size(), sent.size(), n_info,
                n_const, det->num_frames(), std::memcmp(x.data(), y.data(), x.size() * sizeof(gr_complex)) == 0 ? "ok" : "BAD", (int)threw,
                (pass && threw) ? "PASS" : "FAIL");
    return (pass && threw) ? 0 : 1;
}
