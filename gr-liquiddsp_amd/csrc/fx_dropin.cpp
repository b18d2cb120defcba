// fx_dropin.cpp -- the liquid-dsp entry points that gr::liquiddsp's blocks call, re-hosted on the
// batched GPU context (include/fxrx.h, layer 1).  Each function cites the reference call site it serves.
#include <algorithm>
#include <deque>
#include <memory>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <emmintrin.h>
#include "../../include/fxrx.h"
#include "fx_codec.hpp"

namespace {

// Samples go from the caller's (hot, pageable) buffer into a page-locked ring that the CPU never reads again: stream them past
// the cache with non-temporal stores (no read-for-ownership of the destination lines: half the DRAM traffic of a plain memcpy,
// which a single core's copy of tens of MB per second-fraction is bound by).  Falls back to memcpy for unaligned pieces.
inline void stream_copy(fx_complex *dst, const fx_complex *src, size_t n)
{
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) || n < 8) { std::memcpy(dst, src, n * sizeof(fx_complex)); return; }
    const size_t n16 = n / 2;                                  // 16-byte pieces (two samples)
    __m128i *d = reinterpret_cast<__m128i *>(dst); const __m128i *s = reinterpret_cast<const __m128i *>(src);
    size_t i = 0;
    for (; i + 4 <= n16; i += 4) {
        const __m128i a = _mm_loadu_si128(s + i), b = _mm_loadu_si128(s + i + 1), c = _mm_loadu_si128(s + i + 2), e = _mm_loadu_si128(s + i + 3);
        _mm_stream_si128(d + i, a); _mm_stream_si128(d + i + 1, b); _mm_stream_si128(d + i + 2, c); _mm_stream_si128(d + i + 3, e);
    }
    for (; i < n16; i++) _mm_stream_si128(d + i, _mm_loadu_si128(s + i));
    if (n & 1) dst[n - 1] = src[n - 1];
}

// A completed frame waiting for its callback.  Buffers are BORROWED from the context's result arenas (pinned host memory the
// kernels wrote into; valid until the next fxrx_collect on the context) unless `own` holds copies.
struct HeldFrame {
    unsigned char header[20]; int header_valid = 0, payload_valid = 0;
    const unsigned char *payload = nullptr; unsigned payload_len = 0;
    const fx_complex *syms = nullptr; unsigned nsyms = 0;
    bool owned = false; std::vector<unsigned char> own_payload; std::vector<fx_complex> own_syms;
    framesyncstats_s stats{};
    void materialise()                      // take copies: the arenas are about to be reused
    {
        if (owned) return;
        if (payload && payload_len) { own_payload.assign(payload, payload + payload_len); payload = own_payload.data(); }
        if (syms && nsyms) { own_syms.assign(syms, syms + nsyms); syms = own_syms.data(); }
        owned = true;
    }
};

constexpr unsigned kSyncBlockDefault = 1u << 20;    // samples per GPU block (fxrx_sync_set_block / FXRX_SYNC_BLOCK); see DESIGN.md section 7
constexpr unsigned kSyncDepthDefault = 3;           // blocks in flight (FXRX_SYNC_DEPTH)
constexpr unsigned kSyncPollEvery = 1u << 15;       // samples between two looks at whether the oldest block has finished (an event query costs about as much as copying 4000 samples)

}  // namespace

// ------------------------------------------------------------------------------------------ flexframesync
// The reference feeds 256 samples per call from pageable memory (/root/reference/lib/flex_rx_impl.cc:212-215).  Here a call
// copies them into the pinned buffer being filled; a full buffer is submitted as one block of the continuing stream
// (fxrx_submit: upload and the whole kernel chain are enqueued, nothing is waited for) and the next buffer of the ring is
// filled while up to `depth` blocks are in flight.  Finished blocks are collected when a slot is needed or when a poll
// (every few thousand samples) finds the oldest one done; their frames wait in `pending` and leave one per call.
struct fxrx_sync_s {
    framesync_callback cb = nullptr; void *ud = nullptr;
    fxrx_ctx *ctx = nullptr; float threshold = 0.0f; int equalizer = 0, soft = 0;
    unsigned block = kSyncBlockDefault, depth = kSyncDepthDefault;
    std::vector<fx_complex *> bufs; unsigned cur = 0; size_t fill = 0;     // ring of depth + 1 pinned input buffers of `block` samples
    std::deque<HeldFrame> pending; HeldFrame current;
    unsigned errors = 0, since_poll = 0;

    ~fxrx_sync_s() { free_bufs(); }
    void free_bufs() { for (auto p : bufs) fxrx_pinned_free(p); bufs.clear(); cur = 0; fill = 0; }
    bool alloc_bufs()
    {
        free_bufs();
        for (unsigned i = 0; i < depth + 1; i++) {
            fx_complex *p = (fx_complex *)fxrx_pinned_alloc((size_t)block * sizeof(fx_complex));
            if (!p) { free_bufs(); return false; }
            bufs.push_back(p);
        }
        return true;
    }
    void report(const char *what, int rc)
    {
        if (errors++ == 0 || (errors & (errors - 1)) == 0)
            std::fprintf(stderr, "libfxrx: %s failed (%d): %s [%u failures so far]\n", what, rc, ctx ? fxrx_last_error() : "no context", errors);
    }
    // the oldest block in flight: wait for it, queue its frames
    void collect_one()
    {
        for (auto &h : pending) h.materialise();           // (rare: frames are normally delivered long before the next block is due)
        current.materialise();
        const int nr = fxrx_collect(ctx);
        if (nr < 0) { report("flexframesync_execute: block", nr); return; }     // (every block in flight was dropped with it; the streams restart fresh)
        for (int i = 0; i < nr; i++) {
            fxrx_frame f; if (fxrx_result(ctx, (unsigned)i, &f) != 0) continue;
            pending.emplace_back();
            HeldFrame &h = pending.back();
            std::memcpy(h.header, f.header, 20); h.header_valid = f.header_valid; h.payload_valid = f.payload_valid;
            h.payload = f.payload_len ? f.payload : nullptr; h.payload_len = f.payload ? f.payload_len : 0;
            h.syms = f.num_framesyms ? f.framesyms : nullptr; h.nsyms = f.framesyms ? f.num_framesyms : 0;
            h.stats.evm = f.header_valid ? f.evm_db : 0.0f; h.stats.rssi = f.rssi_db; h.stats.cfo = f.cfo;
            h.stats.mod_scheme = f.mod_scheme; h.stats.mod_bps = f.mod_bps; h.stats.check = f.check; h.stats.fec0 = f.fec0; h.stats.fec1 = f.fec1;
        }
    }
    // hand the buffer being filled to the GPU as the stream's next block
    void submit_current()
    {
        if (!fill) return;
        if (fxrx_inflight(ctx) >= depth) collect_one();
        _mm_sfence();                                          // (the streamed samples are in memory before the upload is enqueued)
        const void *p = bufs[cur]; uint64_t n = fill;
        const int r = fxrx_submit(ctx, &p, &n, 0);
        if (r < 0) {
            // the context is as it was before the call; these samples are lost to it: restart the synchroniser behind the gap
            report("flexframesync_execute: submit", r);
            while (fxrx_inflight(ctx)) collect_one();
            fxrx_reset(ctx);
        }
        cur = (cur + 1) % (unsigned)bufs.size(); fill = 0;
    }
    void drain() { while (fxrx_inflight(ctx)) collect_one(); }
    void deliver_one()
    {
        if (pending.empty()) return;
        current = std::move(pending.front()); pending.pop_front();
        if (current.owned) { current.payload = current.own_payload.empty() ? nullptr : current.own_payload.data(); current.syms = current.own_syms.empty() ? nullptr : current.own_syms.data(); }
        current.stats.framesyms = const_cast<fx_complex *>(current.syms);
        current.stats.num_framesyms = current.nsyms;
        if (cb)
            cb(current.header, current.header_valid, const_cast<unsigned char *>(current.payload), current.payload_len, current.payload_valid, current.stats, ud);
    }
};

static fxrx_ctx *sync_make_ctx(const fxrx_sync_s *q)
{
    fxrx_config cfg{}; cfg.device = 0; cfg.mode = FXRX_MODE_FLEX_RX; cfg.n_streams = 1; cfg.want_framesyms = 1;
    if (q) { cfg.threshold = q->threshold; cfg.equalizer = q->equalizer; cfg.soft_decision = q->soft; }
    if (const char *d = std::getenv("FXRX_DEVICE")) cfg.device = std::atoi(d);
    fxrx_ctx *ctx = fxrx_create(&cfg);
    if (ctx && fxrx_set_depth(ctx, q ? q->depth : kSyncDepthDefault) != 0) { fxrx_destroy(ctx); return nullptr; }
    return ctx;
}

extern "C" {

// /root/reference/lib/flex_rx_impl.cc:49
flexframesync flexframesync_create(framesync_callback callback, void *userdata)
{
    std::unique_ptr<fxrx_sync_s> q(new fxrx_sync_s); q->cb = callback; q->ud = userdata;
    if (const char *e = std::getenv("FXRX_SYNC_BLOCK")) q->block = (unsigned)std::max(256, std::atoi(e));
    if (const char *e = std::getenv("FXRX_SYNC_DEPTH")) q->depth = (unsigned)std::min(15, std::max(1, std::atoi(e)));
    q->ctx = sync_make_ctx(q.get());
    if (!q->ctx) return nullptr;
    if (!q->alloc_bufs()) { fxrx_destroy(q->ctx); return nullptr; }
    return q.release();
}
// /root/reference/lib/flex_rx_impl.cc:71
void flexframesync_destroy(flexframesync q) { if (!q) return; fxrx_destroy(q->ctx); delete q; }
void flexframesync_reset(flexframesync q)
{
    if (!q) return;
    q->drain(); q->pending.clear(); q->fill = 0;
    fxrx_reset(q->ctx);
}
// /root/reference/lib/flex_rx_impl.cc:213
void flexframesync_execute(flexframesync q, fx_complex *x, unsigned int n)
{
    if (!q) return;
    q->since_poll += n;
    while (n) {
        const size_t take = std::min<size_t>(n, (size_t)q->block - q->fill);
        stream_copy(q->bufs[q->cur] + q->fill, x, take);
        q->fill += take; x += take; n -= (unsigned)take;
        if (q->fill == q->block) q->submit_current();
    }
    if (q->since_poll >= kSyncPollEvery) {
        q->since_poll = 0;
        if (q->pending.empty() && fxrx_ready(q->ctx) == 1) q->collect_one();
    }
    q->deliver_one();
}
void fxrx_sync_flush(flexframesync q) { if (!q) return; q->submit_current(); q->drain(); }
void fxrx_sync_set_block(flexframesync q, unsigned int samples)
{
    if (!q) return;
    const unsigned nb = samples < 256 ? 256u : samples;
    if (nb == q->block) return;
    fxrx_sync_flush(q);                      // what is queued runs at the old size (buffers move)
    for (auto &h : q->pending) h.materialise();
    const unsigned old = q->block;
    q->block = nb;
    if (!q->alloc_bufs()) { q->errors++; std::fprintf(stderr, "libfxrx: fxrx_sync_set_block: %s\n", fxrx_last_error()); q->block = old; (void)q->alloc_bufs(); }
}
unsigned int fxrx_sync_pending(flexframesync q) { return q ? (unsigned)q->pending.size() : 0; }
unsigned int fxrx_sync_errors(flexframesync q) { return q ? q->errors : 0; }
fxrx_ctx *fxrx_sync_context(flexframesync q) { return q ? q->ctx : nullptr; }
// threshold and equaliser live in the context configuration: make a new context first, swap only if that worked (state is
// reset, as liquid's setters do not promise otherwise); on failure the old context stays and the error is reported
static void sync_recreate(flexframesync q, const char *what)
{
    fxrx_sync_flush(q);
    for (auto &h : q->pending) h.materialise();
    q->current.materialise();
    fxrx_ctx *nc = sync_make_ctx(q);
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
