// fx_tx.cpp -- batched frame generator on the GPU (the flex_tx counterpart: SURVEY section 8(f)-1).
//
// Replaces, for many frames at once, what /root/reference/lib/flex_tx_impl.cc:191-209 (send_pkt) does per PDU:
// flexframegen_assemble (:200) + flexframegen_write_samples (:203-205) with the properties set at :51-56 / :183-189.
// Modulation and the pulse-shaping interpolator -- all of the float work, 2 x 15 fused multiply-adds per output sample
// -- run in fx_txgen_kernel; the packet encoding (CRC, whitening, both code stages, interleavers, bit packing, Gray /
// DPSK index arithmetic) of header and payload packets in fx_txenc_kernel.  With FXTX_HOST_ENCODE=1 the packets are
// encoded by fx_codec.hpp on host threads instead and join at the symbol-index stage.  Output is bit-identical to the host generator behind flexframegen_* (fx_dropin.cpp) and to the oracle's
// fxr_gen_frame.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <memory>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "../../include/fxrx.h"
#include "fx_common.h"
#include "fx_codec.hpp"

extern "C" hipError_t fx_launch_txgen(unsigned ntiles, hipStream_t st, const FxTxJob *jobs, const uint32_t *tile_job, const uint32_t *tile_n0,
                                      const uint8_t *head_idx, const uint8_t *pay_idx, const FxTxTables *T, float2 *out);
extern "C" hipError_t fx_launch_txenc(unsigned njobs, hipStream_t st, const FxTxEncJob *jobs, const uint8_t *pay, const uint32_t *perm_arena,
                                      uint8_t *bufA, uint8_t *bufB, uint8_t *pay_idx, const FxTxTables *T);
extern "C" hipError_t fx_launch_channel(hipStream_t st, float2 *x, unsigned n_streams, unsigned long long n_per_stream, const FxChannel *chs, const FxTxTables *T);
extern "C" void fxrx_set_error(const char *msg);       // fx_host.cpp: thread-local message behind fxrx_last_error()

namespace {
template <class T> struct Dev {
    T *p = nullptr; size_t cap = 0;
    bool reserve(size_t n) { if (n <= cap) return true; if (p) (void)hipFree(p); p = nullptr; cap = 0; if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) return false; cap = n; return true; }
    ~Dev() { if (p) (void)hipFree(p); }
};
}  // namespace

struct fxtx_ctx_s {
    int device = 0;
    hipStream_t stream = nullptr;
    Dev<FxTxTables> d_tab; Dev<uint8_t> d_idx; Dev<FxTxJob> d_jobs; Dev<uint32_t> d_tiles;
    // packet encoder on the GPU: payload bytes, scratch, interleaver gather tables (one per coded length, append-only)
    Dev<uint8_t> d_pay, d_bufA, d_bufB; Dev<FxTxEncJob> d_ejobs; Dev<uint32_t> d_perm; Dev<FxChannel> d_chan;
    std::map<uint32_t, uint32_t> perm_off; std::vector<uint32_t> perm_host; size_t perm_uploaded = 0;
    bool host_encode_only = false;       // FXTX_HOST_ENCODE=1: every frame's packet encoding on the host
};

extern "C" {

fxtx_ctx *fxtx_create(int device)
{
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || device < 0 || device >= nd) { fxrx_set_error("fxtx_create: no usable HIP device (this library has no CPU path)"); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { fxrx_set_error("hipSetDevice failed"); return nullptr; }
    std::unique_ptr<fxtx_ctx_s> c(new fxtx_ctx_s);
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { fxrx_set_error("hipStreamCreate failed"); return nullptr; }
    const fx::HostTables &H = fx::host_tables();
    const fx::BlockCodes &B = fx::block_codes();
    if (const char *e = std::getenv("FXTX_HOST_ENCODE")) c->host_encode_only = std::atoi(e) != 0;
    std::unique_ptr<FxTxTables> t(new FxTxTables);
    std::memset(t.get(), 0, sizeof(FxTxTables));
    for (int i = 0; i < 1024; i++) t->sc[i] = make_float2(H.sc[i].re, H.sc[i].im);
    std::memcpy(t->golenc, B.gol_enc, sizeof t->golenc); std::memcpy(t->h128enc, B.h128_enc, sizeof t->h128enc);
    std::memcpy(t->h74enc, B.h74_enc, 16); std::memcpy(t->h84enc, B.h84_enc, 16);
    std::memcpy(t->sdcol, B.sd_col, 64); std::memcpy(t->sd22col, B.sd22_col, 16); std::memcpy(t->sd39col, B.sd39_col, 32);
    for (int i = 0; i < FX_PN_LEN; i++) t->pn[i] = make_float2(H.pn[i].re, H.pn[i].im);
    for (int i = 0; i < FX_HDR_PILOTS; i++) t->pilots[i] = make_float2(H.pilots[i].re, H.pilots[i].im);
    std::memcpy(t->rsexp, B.rs_exp, 512); std::memcpy(t->rslog, B.rs_log, 256); std::memcpy(t->rsgen, B.rs_gen, 33);
    if (!c->d_tab.reserve(1) || hipMemcpy(c->d_tab.p, t.get(), sizeof(FxTxTables), hipMemcpyHostToDevice) != hipSuccess) { fxrx_set_error("fxtx_create: table upload failed"); return nullptr; }
    return c.release();
}

void fxtx_destroy(fxtx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    delete c;
}

unsigned int fxtx_frame_len(const fxtx_frame *f)
{
    if (!f) return 0;
    fx::FrameGen g; g.ms = f->props.mod_scheme; g.check = f->props.check; g.fec0 = f->props.fec0; g.fec1 = f->props.fec1;
    return g.frame_len(f->payload_len);
}

int fxtx_generate(fxtx_ctx *c, const fxtx_frame *frames, unsigned int n_frames, void *out_device, unsigned long long out_len)
{
    if (!c || (!frames && n_frames) || !out_device) { fxrx_set_error("fxtx_generate: null argument"); return FXRX_ERR_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { fxrx_set_error("hipSetDevice failed"); return FXRX_ERR_HIP; }
    const unsigned nhead = FX_PN_LEN + FX_HDR_SYM;
    std::vector<FxTxJob> jobs(n_frames);
    std::vector<uint8_t> idx; std::vector<uint32_t> tile_job, tile_n0;
    const fx::HostTables &H = fx::host_tables();
    // pass 1 (serial, cheap): sizes, offsets, encoder jobs.  pass 2 (host threads): pulse design for fractional delays, and
    // the packet encoding of frames that do not go through the encoder kernel (FXTX_HOST_ENCODE=1, unsupported schemes).
    uint64_t idx_total = 0, buf_total = 0;
    std::vector<uint8_t> pay; std::vector<FxTxEncJob> ejobs; std::vector<char> on_gpu(n_frames, 0);
    // gather table of the interleaver of a given length: bit q of the output = bit g[q] of the input (inverse of the
    // de-interleaver's); built once per length, kept on the device
    auto perm_for = [&](uint32_t len) -> uint32_t {
        auto it = c->perm_off.find(len);
        if (it == c->perm_off.end()) {
            const std::vector<uint32_t> lab = fx::Interleaver(len).decode_gather();
            const uint32_t off = (uint32_t)c->perm_host.size();
            c->perm_host.resize(off + lab.size());
            for (size_t q = 0; q < lab.size(); q++) c->perm_host[off + lab[q]] = (uint32_t)q;
            it = c->perm_off.emplace(len, off).first;
        }
        return it->second;
    };
    for (unsigned i = 0; i < n_frames; i++) {
        const fxtx_frame &f = frames[i];
        if (fx::modem_bps(f.props.mod_scheme) == 0 || f.payload_len > 65535u || (!f.payload && f.payload_len)) { fxrx_set_error("fxtx_generate: bad frame description"); return FXRX_ERR_ARG; }
        fx::FrameGen g; g.ms = f.props.mod_scheme; g.check = f.props.check; g.fec0 = f.props.fec0; g.fec1 = f.props.fec1;
        FxTxJob &j = jobs[i];
        std::memset(&j, 0, sizeof j);
        j.npay = g.payload_syms(f.payload_len);
        j.nsym = nhead + j.npay + 2 * FX_M;
        j.ms = f.props.mod_scheme; j.out_off = f.out_offset;
        if (f.out_offset + 2ull * j.nsym > out_len) { fxrx_set_error("fxtx_generate: frame does not fit the output buffer"); return FXRX_ERR_ARG; }
        j.idx_off = (uint32_t)idx_total;
        idx_total += j.npay;
        // packet encoding on the GPU (fx_txenc_kernel) unless forced onto the host
        if (!c->host_encode_only && fx::fec_supported(f.props.fec0) && fx::fec_supported(f.props.fec1)) {
            const fx::PacketPlan pl = fx::packet_plan(f.payload_len, f.props.check, f.props.fec0, f.props.fec1);
            FxTxEncJob e{};
            e.pay_off = (uint32_t)pay.size(); e.n = f.payload_len; e.check = f.props.check; e.fec0 = f.props.fec0; e.fec1 = f.props.fec1;
            e.k = pl.k; e.l0 = pl.l0; e.l1 = pl.l1;
            e.perm0_off = perm_for(pl.l0); e.perm1_off = perm_for(pl.l1);
            e.buf_off = (uint32_t)buf_total; buf_total += (std::max(pl.l0, pl.l1) + 16 + 15) & ~15u;
            e.idx_off = j.idx_off; e.npay = j.npay; e.ms = j.ms;
            if (f.payload_len) pay.insert(pay.end(), f.payload, f.payload + f.payload_len);
            ejobs.push_back(e); on_gpu[i] = 1;
        }
        for (uint32_t n0 = 0; n0 < j.nsym; n0 += 256) { tile_job.push_back(i); tile_n0.push_back(n0); }
    }
    // the header symbol words (216 per frame) live behind the payload indices; the header packet (20 bytes, CRC-32,
    // SECDED(72,64), Hamming(8,4), QPSK) goes through the same encoder kernel as the payload packets
    const uint64_t head_base = (idx_total + 15) & ~15ull;
    if (head_base + (uint64_t)FX_HDR_MOD * n_frames >= (1ull << 32)) { fxrx_set_error("fxtx_generate: batch too large"); return FXRX_ERR_ARG; }
    idx.assign((size_t)(head_base + (uint64_t)FX_HDR_MOD * n_frames) + 16, 0);
    const fx::PacketPlan hp = fx::packet_plan(FX_HDR_DEC, FX_CRC_32, FX_FEC_SECDED7264, FX_FEC_HAMMING84);
    for (unsigned i = 0; i < n_frames; i++) {
        const fxtx_frame &f = frames[i];
        jobs[i].head_off = (uint32_t)(head_base + (uint64_t)FX_HDR_MOD * i);
        if (c->host_encode_only) continue;
        fx::FrameGen g; g.ms = f.props.mod_scheme; g.check = f.props.check; g.fec0 = f.props.fec0; g.fec1 = f.props.fec1;
        uint8_t hd[FX_HDR_DEC]; g.head_bytes(f.header, f.payload_len, hd);
        FxTxEncJob e{};
        e.pay_off = (uint32_t)pay.size(); e.n = FX_HDR_DEC; e.check = FX_CRC_32; e.fec0 = FX_FEC_SECDED7264; e.fec1 = FX_FEC_HAMMING84;
        e.k = hp.k; e.l0 = hp.l0; e.l1 = hp.l1;
        e.perm0_off = perm_for(hp.l0); e.perm1_off = perm_for(hp.l1);
        e.buf_off = (uint32_t)buf_total; buf_total += (std::max(hp.l0, hp.l1) + 16 + 15) & ~15u;
        e.idx_off = jobs[i].head_off; e.npay = FX_HDR_MOD; e.ms = FX_MODEM_QPSK;
        pay.insert(pay.end(), hd, hd + FX_HDR_DEC);
        ejobs.push_back(e);
    }
    (void)fx::block_codes();                                        // build the shared code tables before the threads start
    auto encode_range = [&](unsigned first, unsigned step) {
        for (unsigned i = first; i < n_frames; i += step) {
            const fxtx_frame &f = frames[i];
            fx::FrameGen g; g.ms = f.props.mod_scheme; g.check = f.props.check; g.fec0 = f.props.fec0; g.fec1 = f.props.fec1;
            FxTxJob &j = jobs[i];
            if (c->host_encode_only) g.head_words(f.header, f.payload_len, idx.data() + j.head_off);
            if (!on_gpu[i]) {
                const std::vector<uint8_t> w = g.payload_indices(f.payload, f.payload_len);
                std::memcpy(idx.data() + j.idx_off, w.data(), w.size());
            }
            if (f.dt != 0.0f) fx::design_arkaiser(FX_K, FX_M, FX_BETA, f.dt, j.taps); else std::memcpy(j.taps, H.txh, sizeof H.txh);
        }
    };
    const unsigned nthr = std::max(1u, std::min({ 16u, std::thread::hardware_concurrency(), n_frames / 8u }));
    if (nthr <= 1) encode_range(0, 1);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthr; t++) pool.emplace_back(encode_range, t, nthr);
        for (auto &t : pool) t.join();
    }
    if (n_frames == 0) return 0;
    const size_t nt = tile_job.size();
    if (!c->d_jobs.reserve(n_frames) || !c->d_idx.reserve(idx.size()) || !c->d_tiles.reserve(2 * nt)) { fxrx_set_error("hipMalloc failed"); return FXRX_ERR_HIP; }
    bool ok = hipMemcpyAsync(c->d_jobs.p, jobs.data(), n_frames * sizeof(FxTxJob), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->d_idx.p, idx.data(), idx.size(), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->d_tiles.p, tile_job.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    ok = ok && hipMemcpyAsync(c->d_tiles.p + nt, tile_n0.data(), nt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    if (!ejobs.empty()) {
        pay.resize(pay.size() + 16);
        if (c->perm_uploaded != c->perm_host.size()) {
            ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && c->d_perm.reserve(c->perm_host.size()) &&
                 hipMemcpy(c->d_perm.p, c->perm_host.data(), c->perm_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
            c->perm_uploaded = c->perm_host.size();
        }
        ok = ok && c->d_pay.reserve(pay.size()) && c->d_bufA.reserve(buf_total + 16) && c->d_bufB.reserve(buf_total + 16) && c->d_ejobs.reserve(ejobs.size());
        ok = ok && hipMemcpyAsync(c->d_pay.p, pay.data(), pay.size(), hipMemcpyHostToDevice, c->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(c->d_ejobs.p, ejobs.data(), ejobs.size() * sizeof(FxTxEncJob), hipMemcpyHostToDevice, c->stream) == hipSuccess;
        ok = ok && fx_launch_txenc((unsigned)ejobs.size(), c->stream, c->d_ejobs.p, c->d_pay.p, c->d_perm.p, c->d_bufA.p, c->d_bufB.p, c->d_idx.p, c->d_tab.p) == hipSuccess;
    }
    ok = ok && fx_launch_txgen((unsigned)nt, c->stream, c->d_jobs.p, c->d_tiles.p, c->d_tiles.p + nt, c->d_idx.p, c->d_idx.p, c->d_tab.p, (float2 *)out_device) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;       // the staging vectors above are pageable and die with this call
    if (!ok) { fxrx_set_error(std::string("fxtx_generate: ") .append(hipGetErrorString(hipGetLastError())).c_str()); return FXRX_ERR_HIP; }
    return 0;
}

// The channel of SURVEY 8(d) on the device: stream s = samples [s n_per_stream, (s + 1) n_per_stream) of iq_device, turned
// by its carrier offset and phase, scaled, and given white Gaussian noise (fx_channel_kernel).  Synchronous.
int fxtx_apply_channel(fxtx_ctx *c, void *iq_device, unsigned int n_streams, unsigned long long n_per_stream, const fxtx_channel *ch)
{
    if (!c || !iq_device || (!ch && n_streams)) { fxrx_set_error("fxtx_apply_channel: null argument"); return FXRX_ERR_ARG; }
    if (n_per_stream & 1ull) { fxrx_set_error("fxtx_apply_channel: n_per_stream must be even"); return FXRX_ERR_ARG; }
    if (n_streams > 65535u) { fxrx_set_error("fxtx_apply_channel: at most 65535 streams per call"); return FXRX_ERR_ARG; }
    if (hipSetDevice(c->device) != hipSuccess) { fxrx_set_error("hipSetDevice failed"); return FXRX_ERR_HIP; }
    std::vector<FxChannel> h(n_streams);
    auto units = [](double rad) { return (uint32_t)(long long)std::nearbyint(rad * (4294967296.0 / 6.283185307179586)); };
    for (unsigned s = 0; s < n_streams; s++) {
        h[s].th0 = units(ch[s].phase); h[s].dl = units(ch[s].cfo); h[s].gain = ch[s].gain; h[s].sigma = ch[s].sigma; h[s].seed = ch[s].seed;
    }
    bool ok = c->d_chan.reserve(n_streams ? n_streams : 1);
    ok = ok && hipMemcpyAsync(c->d_chan.p, h.data(), n_streams * sizeof(FxChannel), hipMemcpyHostToDevice, c->stream) == hipSuccess;
    ok = ok && fx_launch_channel(c->stream, (float2 *)iq_device, n_streams, n_per_stream, c->d_chan.p, c->d_tab.p) == hipSuccess;
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    if (!ok) { fxrx_set_error(std::string("fxtx_apply_channel: ").append(hipGetErrorString(hipGetLastError())).c_str()); return FXRX_ERR_HIP; }
    return 0;
}

}  // extern "C"
