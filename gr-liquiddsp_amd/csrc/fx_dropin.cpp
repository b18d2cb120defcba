// fx_dropin.cpp -- the liquid-dsp entry points that gr::liquiddsp's blocks call, re-hosted on the
// batched GPU context (include/fxrx.h, layer 1).  Each function cites the reference call site it serves.
#include <deque>
#include <memory>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../include/fxrx.h"
#include "fx_codec.hpp"

namespace {

struct HeldFrame {
    unsigned char header[20]; int header_valid = 0, payload_valid = 0;
    std::vector<unsigned char> payload; std::vector<fx_complex> syms;
    framesyncstats_s stats{};
};

}  // namespace

// ------------------------------------------------------------------------------------------ flexframesync
struct fxrx_sync_s {
    framesync_callback cb = nullptr; void *ud = nullptr;
    fxrx_ctx *ctx = nullptr; float threshold = 0.0f; int equalizer = 0, soft = 0;
    std::vector<fx_complex> queue; unsigned block = 1u << 16;
    std::deque<HeldFrame> pending; HeldFrame current;

    unsigned errors = 0;
    // returns false when the GPU call failed: the samples stay queued (they run again with the next call), the error
    // text stays in fxrx_last_error(), one line goes to stderr, `errors` counts
    bool run()
    {
        const void *p = queue.data(); uint64_t n = queue.size();
        int nr = ctx ? fxrx_process(ctx, &p, &n, 0) : FXRX_ERR_STATE;
        if (nr < 0) {
            if (errors++ == 0 || (errors & (errors - 1)) == 0)
                std::fprintf(stderr, "libfxrx: flexframesync_execute: block of %llu samples failed (%d): %s [%u failures so far]\n",
                             (unsigned long long)n, nr, ctx ? fxrx_last_error() : "no context", errors);
            if (queue.size() > (size_t)64 * block) queue.erase(queue.begin(), queue.end() - (std::ptrdiff_t)(32 * (size_t)block));   // bounded
            return false;
        }
        queue.clear();
        for (int i = 0; i < nr; i++) {
            fxrx_frame f; if (fxrx_result(ctx, (unsigned)i, &f) != 0) continue;
            HeldFrame h;
            std::memcpy(h.header, f.header, 20); h.header_valid = f.header_valid; h.payload_valid = f.payload_valid;
            if (f.payload && f.payload_len) h.payload.assign(f.payload, f.payload + f.payload_len);
            if (f.framesyms && f.num_framesyms) h.syms.assign(f.framesyms, f.framesyms + f.num_framesyms);
            h.stats.evm = f.header_valid ? f.evm_db : 0.0f; h.stats.rssi = f.rssi_db; h.stats.cfo = f.cfo;
            h.stats.mod_scheme = f.mod_scheme; h.stats.mod_bps = f.mod_bps; h.stats.check = f.check; h.stats.fec0 = f.fec0; h.stats.fec1 = f.fec1;
            pending.push_back(std::move(h));
        }
        return true;
    }
    void deliver_one()
    {
        if (pending.empty()) return;
        current = std::move(pending.front()); pending.pop_front();
        current.stats.framesyms = current.syms.empty() ? nullptr : current.syms.data();
        current.stats.num_framesyms = (unsigned)current.syms.size();
        if (cb)
            cb(current.header, current.header_valid, current.payload.empty() ? nullptr : current.payload.data(),
               (unsigned)current.payload.size(), current.payload_valid, current.stats, ud);
    }
};

extern "C" {

// /root/reference/lib/flex_rx_impl.cc:49
flexframesync flexframesync_create(framesync_callback callback, void *userdata)
{
    fxrx_config cfg{}; cfg.device = 0; cfg.mode = FXRX_MODE_FLEX_RX; cfg.n_streams = 1; cfg.want_framesyms = 1;
    if (const char *d = std::getenv("FXRX_DEVICE")) cfg.device = std::atoi(d);
    fxrx_ctx *ctx = fxrx_create(&cfg);
    if (!ctx) return nullptr;
    fxrx_sync_s *q = new fxrx_sync_s; q->cb = callback; q->ud = userdata; q->ctx = ctx;
    return q;
}
// /root/reference/lib/flex_rx_impl.cc:71
void flexframesync_destroy(flexframesync q) { if (!q) return; fxrx_destroy(q->ctx); delete q; }
void flexframesync_reset(flexframesync q) { if (!q) return; fxrx_reset(q->ctx); q->queue.clear(); q->pending.clear(); }
// /root/reference/lib/flex_rx_impl.cc:213
void flexframesync_execute(flexframesync q, fx_complex *x, unsigned int n)
{
    if (!q) return;
    if (n) q->queue.insert(q->queue.end(), x, x + n);
    if (q->queue.size() >= q->block) (void)q->run();
    q->deliver_one();
}
void fxrx_sync_flush(flexframesync q) { if (!q) return; if (!q->queue.empty()) (void)q->run(); }
void fxrx_sync_set_block(flexframesync q, unsigned int samples) { if (q) q->block = samples ? samples : 1; }
unsigned int fxrx_sync_pending(flexframesync q) { return q ? (unsigned)q->pending.size() : 0; }
unsigned int fxrx_sync_errors(flexframesync q) { return q ? q->errors : 0; }
// threshold and equaliser live in the context configuration: make a new context first, swap only if that worked (state is
// reset, as liquid's setters do not promise otherwise); on failure the old context stays and the error is reported
static void sync_recreate(flexframesync q, const char *what)
{
    fxrx_config cfg{}; cfg.device = 0; cfg.mode = FXRX_MODE_FLEX_RX; cfg.n_streams = 1; cfg.want_framesyms = 1;
    cfg.threshold = q->threshold; cfg.equalizer = q->equalizer; cfg.soft_decision = q->soft;
    if (const char *d = std::getenv("FXRX_DEVICE")) cfg.device = std::atoi(d);
    fxrx_ctx *nc = fxrx_create(&cfg);
    if (!nc) { q->errors++; std::fprintf(stderr, "libfxrx: %s: %s (setting unchanged)\n", what, fxrx_last_error()); return; }
    fxrx_destroy(q->ctx);
    q->ctx = nc;
}
void fxrx_sync_set_threshold(flexframesync q, float t) { if (!q) return; q->threshold = t; sync_recreate(q, "fxrx_sync_set_threshold"); }
void fxrx_sync_set_equalizer(flexframesync q, int on) { if (!q) return; q->equalizer = on ? 1 : 0; sync_recreate(q, "fxrx_sync_set_equalizer"); }
void fxrx_sync_set_soft(flexframesync q, int on) { if (!q) return; q->soft = on ? 1 : 0; sync_recreate(q, "fxrx_sync_set_soft"); }

}  // extern "C"

// ------------------------------------------------------------------------------------------ msequence
struct fxrx_mseq_s { fx::MSeq ms; fxrx_mseq_s(unsigned m, unsigned g, unsigned a) : ms(m, g, a) {} };
extern "C" {
// /root/reference/lib/frame_detector_cc_impl.cc:47
msequence msequence_create(unsigned int m, unsigned int g, unsigned int a) { if (m < 2 || m > 15) return nullptr; return new fxrx_mseq_s(m, g, a); }
// /root/reference/lib/frame_detector_cc_impl.cc:49-50
unsigned int msequence_advance(msequence ms) { return ms ? ms->ms.advance() : 0; }
// /root/reference/lib/frame_detector_cc_impl.cc:52
void msequence_destroy(msequence ms) { delete ms; }
}

// ------------------------------------------------------------------------------------------ qdetector_cccf
struct fxrx_qdet_s {
    fxrx_ctx *ctx = nullptr; float threshold = 0.5f;
    std::vector<fx_complex> hist;       // samples since hist_base (kept long enough to cut aligned windows)
    int64_t hist_base = 0; size_t fed = 0;     // `fed` samples of hist already given to the GPU
    unsigned block = 1u << 16;
    std::deque<fxrx_frame> pending;
    fx_complex window[FX_NFFT];
    float tau = 0, gamma = 0, dphi = 0, phi = 0;
    unsigned errors = 0;
    bool make_ctx()                     // a failed re-creation keeps the old context
    {
        fxrx_config cfg{}; cfg.device = 0; cfg.mode = FXRX_MODE_DETECTOR; cfg.n_streams = 1; cfg.threshold = threshold;
        if (const char *d = std::getenv("FXRX_DEVICE")) cfg.device = std::atoi(d);
        fxrx_ctx *nc = fxrx_create(&cfg);
        if (!nc) { errors++; std::fprintf(stderr, "libfxrx: qdetector_cccf: %s\n", fxrx_last_error()); return false; }
        if (ctx) fxrx_destroy(ctx);
        ctx = nc;
        return true;
    }
    void run()
    {
        const void *p = hist.data() + fed; uint64_t n = hist.size() - fed;
        int nr = fxrx_process(ctx, &p, &n, 0);
        if (nr < 0) {                   // samples stay un-fed: they run again with the next block
            if (errors++ == 0 || (errors & (errors - 1)) == 0)
                std::fprintf(stderr, "libfxrx: qdetector_cccf_execute: block of %llu samples failed (%d): %s [%u failures so far]\n",
                             (unsigned long long)n, nr, fxrx_last_error(), errors);
            return;
        }
        fed = hist.size();
        for (int i = 0; i < nr; i++) { fxrx_frame f; if (fxrx_result(ctx, (unsigned)i, &f) == 0) pending.push_back(f); }
    }
};
extern "C" {
// /root/reference/lib/frame_detector_cc_impl.cc:54
qdetector_cccf qdetector_cccf_create_linear(fx_complex *seq, unsigned int len, int ftype, unsigned int k, unsigned int m, float beta)
{
    const fx::HostTables &T = fx::host_tables();
    if (!seq || len != FX_PN_LEN || ftype != LIQUID_FIRFILT_ARKAISER || k != FX_K || m != FX_M || std::fabs(beta - FX_BETA) > 1e-6f) return nullptr;
    for (unsigned i = 0; i < len; i++)
        if (std::fabs(seq[i].re - T.pn[i].re) > 1e-6f || std::fabs(seq[i].im - T.pn[i].im) > 1e-6f) return nullptr;
    fxrx_qdet_s *q = new fxrx_qdet_s; q->make_ctx();
    if (!q->ctx) { delete q; return nullptr; }
    return q;
}
// /root/reference/lib/frame_detector_cc_impl.cc:63
void qdetector_cccf_destroy(qdetector_cccf q) { if (!q) return; fxrx_destroy(q->ctx); delete q; }
// /root/reference/lib/frame_detector_cc_impl.cc:55
void qdetector_cccf_set_threshold(qdetector_cccf q, float t)
{
    if (!q) return;
    const float old = q->threshold;
    q->threshold = t;
    if (!q->make_ctx()) { q->threshold = old; return; }
    q->hist.clear(); q->hist_base = 0; q->fed = 0; q->pending.clear();
}
unsigned int fxrx_qdet_errors(qdetector_cccf q) { return q ? q->errors : 0; }
// /root/reference/lib/frame_detector_cc_impl.cc:77
void *qdetector_cccf_execute(qdetector_cccf q, fx_complex x)
{
    if (!q || !q->ctx) return nullptr;
    q->hist.push_back(x);
    if (q->hist.size() - q->fed >= q->block) {
        q->run();
        // drop history no pending detection can need any more (keep 2 windows of slack)
        int64_t keep_from = q->hist_base + (int64_t)q->fed - 4 * FX_NFFT;
        for (const auto &f : q->pending) keep_from = std::min<int64_t>(keep_from, f.start);
        if (keep_from > q->hist_base) {
            size_t drop = (size_t)(keep_from - q->hist_base);
            q->hist.erase(q->hist.begin(), q->hist.begin() + (std::ptrdiff_t)drop);
            q->hist_base += (int64_t)drop; q->fed -= drop;
        }
    }
    if (q->pending.empty()) return nullptr;
    fxrx_frame f = q->pending.front(); q->pending.pop_front();
    q->tau = f.tau; q->gamma = f.gamma; q->dphi = f.dphi; q->phi = f.phi;
    for (int i = 0; i < FX_NFFT; i++) {
        int64_t p = f.start + i - q->hist_base;
        q->window[i] = (p >= 0 && p < (int64_t)q->hist.size() && f.start + i >= 0) ? q->hist[(size_t)p] : fx_complex{ 0, 0 };
    }
    return q->window;
}
// /root/reference/lib/frame_detector_cc_impl.cc:90-93 (commented-out getters)
float qdetector_cccf_get_tau(qdetector_cccf q) { return q ? q->tau : 0; }
float qdetector_cccf_get_gamma(qdetector_cccf q) { return q ? q->gamma : 0; }
float qdetector_cccf_get_dphi(qdetector_cccf q) { return q ? q->dphi : 0; }
float qdetector_cccf_get_phi(qdetector_cccf q) { return q ? q->phi : 0; }
unsigned int qdetector_cccf_get_buf_len(qdetector_cccf) { return FX_NFFT; }
}

// ------------------------------------------------------------------------------------------ flexframegen
struct fxrx_gen_s { fx::FrameGen g; bool assembled = false; };
extern "C" {
// /root/reference/lib/flex_tx_impl.cc:51
int flexframegenprops_init_default(flexframegenprops_s *p)
{
    if (!p) return -1;
    p->check = FX_CRC_32; p->fec0 = FX_FEC_NONE; p->fec1 = FX_FEC_NONE; p->mod_scheme = FX_MODEM_QPSK;
    return 0;
}
static int apply_props(fxrx_gen_s *q, const flexframegenprops_s *p)
{
    if (!fx::modem_bps(p->mod_scheme) || !fx::fec_supported(p->fec0) || !fx::fec_supported(p->fec1) ||
        p->check == FX_CRC_UNKNOWN || p->check > FX_CRC_32) return -1;
    q->g.check = p->check; q->g.fec0 = p->fec0; q->g.fec1 = p->fec1; q->g.ms = p->mod_scheme;
    return 0;
}
// /root/reference/lib/flex_tx_impl.cc:56
flexframegen flexframegen_create(flexframegenprops_s *props)
{
    fxrx_gen_s *q = new fxrx_gen_s;
    flexframegenprops_s d; flexframegenprops_init_default(&d);
    if (apply_props(q, props ? props : &d) != 0) { delete q; return nullptr; }
    return q;
}
// /root/reference/lib/flex_tx_impl.cc:72
void flexframegen_destroy(flexframegen q) { delete q; }
// /root/reference/lib/flex_tx_impl.cc:188
int flexframegen_setprops(flexframegen q, flexframegenprops_s *props) { if (!q || !props) return -1; return apply_props(q, props); }
// /root/reference/lib/flex_tx_impl.cc:198
int flexframegen_assemble(flexframegen q, const unsigned char *header, const unsigned char *payload, unsigned int n)
{
    if (!q || (!payload && n) || n > 65535) return -1;
    static const unsigned char z = 0;
    q->g.assemble(header, payload ? payload : &z, n); q->assembled = true;
    return 0;
}
// /root/reference/lib/flex_tx_impl.cc:199
unsigned int flexframegen_getframelen(flexframegen q) { return (q && q->assembled) ? FX_K * (unsigned)q->g.syms.size() : 0; }
// /root/reference/lib/flex_tx_impl.cc:201.  Returns 1 when the frame is complete (liquid's convention).
int flexframegen_write_samples(flexframegen q, fx_complex *buf, unsigned int len)
{
    if (!q || !q->assembled || !buf) return -1;
    unsigned need = FX_K * (unsigned)q->g.syms.size();
    if (len < need) return -1;
    q->g.write(reinterpret_cast<fx::cf *>(buf));
    for (unsigned i = need; i < len; i++) buf[i] = fx_complex{ 0, 0 };
    return 1;
}
void fxrx_gen_set_delay(flexframegen q, float dt) { if (q) q->g.dt = dt; }
}
