// codec_sanitize.cpp -- AddressSanitizer / UndefinedBehaviorSanitizer run of the product's host-side generator
// (gr-liquiddsp_amd/csrc/fx_codec.hpp: FrameGen behind flexframegen_*, the packet encoder, interleaver tables, pulse design),
// compiled host-only with g++ (no HIP): tests/test_sanitizers.py.
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../gr-liquiddsp_amd/csrc/fx_codec.hpp"

int main()
{
    static const unsigned mods[] = { 1, 2, 3, 4, 9, 10, 11, 18, 27, 28, 29, 40 };
    static const unsigned fecs[] = { 1, 4, 5, 6, 7, 8, 9, 10, 11, 15, 16, 17, 18, 19, 20, 27 };
    static const unsigned lens[] = { 0, 1, 2, 7, 64, 223, 224, 1024, 4097 };
    unsigned long long s = 0x9E3779B97F4A7C15ull; auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 32); };
    (void)fx::host_tables(); (void)fx::block_codes();
    size_t total = 0; int runs = 0;
    for (unsigned m : mods) for (unsigned f0 : fecs) {
        const unsigned f1 = fecs[rnd() % 16], n = lens[rnd() % 9], check = 1 + rnd() % 6;
        std::vector<uint8_t> pl(n + 1); for (auto &b : pl) b = (uint8_t)rnd();
        uint8_t hd[14]; for (auto &b : hd) b = (uint8_t)rnd();
        fx::FrameGen g; g.ms = m; g.check = check; g.fec0 = f0; g.fec1 = f1; g.dt = (runs & 3) ? 0.0f : 0.3f;
        g.assemble(hd, pl.data(), n);
        std::vector<fx::cf> out(FX_K * g.syms.size());
        g.write(out.data());
        if (g.frame_len(n) != out.size()) { std::printf("FAIL frame_len %u vs %zu\n", g.frame_len(n), out.size()); return 1; }
        const fx::PacketPlan p = fx::packet_plan(n, check, f0, f1);
        if (p.l1 == 0 && n) { std::printf("FAIL plan\n"); return 1; }
        total += out.size(); runs++;
    }
    {   // maximum payload, heaviest chain
        std::vector<uint8_t> pl(65535, 0xA5); uint8_t hd[14] = { 0 };
        fx::FrameGen g; g.ms = FX_MODEM_PSK2; g.check = FX_CRC_32; g.fec0 = FX_FEC_CONV_V27; g.fec1 = FX_FEC_RS_M8;
        g.assemble(hd, pl.data(), 65535);
        std::vector<fx::cf> out(FX_K * g.syms.size()); g.write(out.data()); total += out.size(); runs++;
    }
    for (unsigned len : { 2u, 3u, 54u, 27u, 1027u, 2056u, 65535u }) { auto t = fx::Interleaver(len).decode_gather(); if (t.size() != 8u * len) { std::printf("FAIL interleaver %u\n", len); return 1; } }
    float taps[32]; for (float dt : { -0.5f, -0.1f, 0.0f, 0.49f }) fx::design_arkaiser(FX_K, FX_M, FX_BETA, dt, taps);
    float eq[16]; fx::design_eq_init(eq);
    std::printf("fx_codec.hpp under ASan/UBSan: %d frames, %zu samples\n", runs, total);
    return 0;
}
