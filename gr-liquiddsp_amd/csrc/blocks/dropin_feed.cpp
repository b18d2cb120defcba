// dropin_feed.cpp -- bench / test driver for the boundary the reference really has: a host IQ buffer goes through the C++
// flex_rx block shell (gr-liquiddsp_amd/csrc/blocks/fx_blocks.hpp) the way GNU Radio's scheduler would feed it -- work() calls
// of `items_per_work` items from pageable memory, inside each the reference's own loop of flexframesync_execute(q, in, 256)
// calls (/root/reference/lib/flex_rx_impl.cc:212-215), one frame drained and published as three messages after each
// (:216-250).  Messages are counted and the payload bytes hashed so that the caller can check every frame.
// Built by csrc/Makefile into csrc/libdropin_feed.so; links against libfxrx.so only.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "fx_blocks.hpp"

extern "C" {

struct dropin_stats {
    double   seconds;               // wall time of the work() loop including the final flush
    uint64_t frames, header_valid, payload_valid, payload_bytes, constellation_syms, packet_infos;
    uint64_t payload_hash;          // FNV-1a over all payload_data bytes in order of arrival
    uint64_t errors;                // fxrx_sync_errors() at the end
    double   first_frame_seconds;   // wall time from the first work() call to the first published frame
};

// returns 0, or -1 when the block could not be made (no GPU: the text is in fxrx_last_error())
int dropin_feed(const float *iq, unsigned long long n_samples, unsigned items_per_work, unsigned repeats, dropin_stats *out)
{
    using namespace gr::liquiddsp;
    using clk = std::chrono::steady_clock;
    dropin_stats st{}; st.payload_hash = 14695981039346656037ull;
    flex_rx::sptr rx;
    try { rx = flex_rx::make(); } catch (const std::exception &e) { std::fprintf(stderr, "dropin_feed: %s\n", e.what()); return -1; }
    clk::time_point t0; bool first = true;
    rx->set_msg_sink([&](const std::string &port, const msg_t &m) {
        if (port == "constellation") {
            st.frames++; st.constellation_syms += m.c32.size();
            if (first) { first = false; st.first_frame_seconds = std::chrono::duration<double>(clk::now() - t0).count(); }
        }
        else if (port == "payload_data") {
            st.header_valid++; st.payload_bytes += m.u8.size();
            uint64_t h = st.payload_hash;
            for (uint8_t b : m.u8) { h ^= b; h *= 1099511628211ull; }
            st.payload_hash = h;
        } else if (port == "packet_info") { st.packet_infos++; st.payload_valid += (uint64_t)m.dict.at("payload_valid"); }
    });
    if (items_per_work < 256) items_per_work = 256;
    items_per_work -= items_per_work % 256;
    gr_vector_const_void_star in(1); gr_vector_void_star outv;
    t0 = clk::now();
    for (unsigned r = 0; r < (repeats ? repeats : 1u); r++) {
        if (r) flexframesync_reset(rx->handle());      // every repeat is an independent capture
        unsigned long long p = 0;
        while (p + 256 <= n_samples) {
            const unsigned long long left = n_samples - p;
            int n = (int)(left < items_per_work ? left - left % 256 : items_per_work);
            in[0] = iq + 2 * p;
            rx->work(n, in, outv);
            p += (unsigned long long)n;
        }
        rx->flush();
    }
    st.seconds = std::chrono::duration<double>(clk::now() - t0).count();
    st.errors = fxrx_sync_errors(rx->handle());
    *out = st;
    return 0;
}

// The bare C boundary without the block shell's messages: flexframesync_execute(q, x, 256) over the buffer, a callback that
// counts frames and hashes payload bytes where they lie (no copies) -- what the library itself costs behind the reference's
// calling convention, apart from what the caller does with a frame.
static int raw_cb(unsigned char *, int hv, unsigned char *p, unsigned int n, int pv, framesyncstats_s st, void *ud)
{
    dropin_stats *s = (dropin_stats *)ud;
    s->frames++; s->constellation_syms += st.num_framesyms;
    if (hv) {
        s->header_valid++; s->payload_bytes += n; s->payload_valid += (uint64_t)(pv != 0); s->packet_infos++;
        uint64_t h = s->payload_hash;
        for (unsigned i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
        s->payload_hash = h;
    }
    return 0;
}
int dropin_feed_raw(const float *iq, unsigned long long n_samples, unsigned repeats, dropin_stats *out)
{
    using clk = std::chrono::steady_clock;
    dropin_stats st{}; st.payload_hash = 14695981039346656037ull;
    flexframesync q = flexframesync_create(raw_cb, &st);
    if (!q) return -1;
    const auto t0 = clk::now();
    for (unsigned r = 0; r < (repeats ? repeats : 1u); r++) {
        if (r) flexframesync_reset(q);
        for (unsigned long long p = 0; p + 256 <= n_samples; p += 256) flexframesync_execute(q, (fx_complex *)(iq + 2 * p), 256);
        fxrx_sync_flush(q);
        while (fxrx_sync_pending(q)) flexframesync_execute(q, nullptr, 0);
    }
    st.seconds = std::chrono::duration<double>(clk::now() - t0).count();
    st.errors = fxrx_sync_errors(q);
    flexframesync_destroy(q);
    *out = st;
    return 0;
}

#ifndef DROPIN_NO_CEILING      /* (built without it against round 2's library, which has no fxrx_pinned_alloc) */
// What the calling convention alone costs: the same stream in the same 256-sample pieces, each piece only copied into a page-locked
// buffer of `block` samples (what any GPU implementation behind flexframesync_execute has to do with pageable caller memory
// before it can upload).  Returns samples per second of that loop -- the ceiling of the drop-in boundary on this host.
double dropin_copy_ceiling(const float *iq, unsigned long long n_samples, unsigned block, unsigned repeats)
{
    using clk = std::chrono::steady_clock;
    fx_complex *buf[4];                                         // a ring like the library's: the destination is never cache-resident
    for (auto &b : buf) { b = (fx_complex *)fxrx_pinned_alloc((size_t)block * sizeof(fx_complex)); if (!b) return 0.0; }
    volatile size_t sink = 0;
    const auto t0 = clk::now();
    for (unsigned r = 0; r < (repeats ? repeats : 1u); r++) {
        size_t fill = 0; unsigned cur = 0;
        for (unsigned long long p = 0; p + 256 <= n_samples; p += 256) {
            std::memcpy(buf[cur] + fill, iq + 2 * p, 256 * sizeof(fx_complex));
            fill += 256; if (fill + 256 > block) { sink = sink + fill; fill = 0; cur = (cur + 1) & 3u; }
        }
    }
    const double dt = std::chrono::duration<double>(clk::now() - t0).count();
    for (auto b : buf) fxrx_pinned_free(b);
    return (double)(repeats ? repeats : 1u) * (double)(n_samples - n_samples % 256) / dt;
}
#endif

// the same from `n_threads` threads at once, each with its own block instance (GNU Radio: one thread per block) and its own
// buffer; stats[t] per thread.  Returns the number of threads that failed to make their block.
int dropin_feed_threads(const float *const *iq, const unsigned long long *n_samples, unsigned n_threads, unsigned items_per_work, dropin_stats *stats)
{
    std::vector<std::thread> th; std::vector<int> rc(n_threads, 0);
    for (unsigned t = 0; t < n_threads; t++)
        th.emplace_back([&, t]() { rc[t] = dropin_feed(iq[t], n_samples[t], items_per_work, 1, stats + t); });
    for (auto &x : th) x.join();
    int bad = 0; for (int r : rc) bad += r != 0;
    return bad;
}

}  // extern "C"
