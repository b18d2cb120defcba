// fx_host.cpp -- host runtime of libfxrx.so: device tables, per-stream state carried across calls,
// speculative segment walking with exact stitching, payload job planning, result marshalling and the
// C ABI declared in include/fxrx.h.
//
// Why segments: liquid's synchroniser is one sequential state machine per stream (where the detector
// restarts after a frame depends on that frame's header).  Each stream is cut into segments that are
// walked concurrently from a freshly-reset detector; a segment's walker, once past its end, keeps
// seeking until its next detection (a, cfo_bin) -- the hand-off target.  If the next segment's
// speculative list contains that same (a, cfo_bin), everything after it is provably what the
// sequential machine would have produced (frame processing depends only on the aligned start and
// the coarse bin), so the lists are spliced; otherwise that segment is re-walked from the true state
// ("repair").  The result is identical to a single sequential walk, for any segment size.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include "../../include/fxrx.h"
#include "fx_device.h"
#include "fx_codec.hpp"

extern "C" hipError_t fx_launch_walk(unsigned mode, unsigned njobs, hipStream_t st, const FxWalkJob *jobs, FxWalkResult *results,
                                     FxFrame *frames, const FxTables *T);
extern "C" hipError_t fx_launch_seekverify(unsigned njobs, hipStream_t st, const FxVerifyJob *jobs, FxVerifyResult *results, const FxTables *T);
extern "C" __global__ void fx_paymf_kernel(const FxPayJob *, const uint32_t *, const uint32_t *, float2 *, const FxTables *);
extern "C" hipError_t fx_launch_paypll(unsigned ms, unsigned njobs, unsigned wg_skip, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx,
                                       const float2 *sym_raw, float2 *framesyms, uint8_t *hard, FxPayResult *res, const FxTables *T);
extern "C" hipError_t fx_launch_paydec(int with_rs, unsigned njobs, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx, const uint8_t *hard,
                                       const uint32_t *perm_arena, uint8_t *bufA, uint8_t *bufB, unsigned long long *dw_arena, uint8_t *out,
                                       FxPayResult *res, const FxTables *T);

namespace {

thread_local std::string g_err;
void set_err(const std::string &s) { g_err = s; }

#define HIP_OK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_err(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
            return FXRX_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// growable device / pinned-host buffers
template <class T> struct DevBuf {
    T *p = nullptr; size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        size_t nc = std::max(n, cap + cap / 2);
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc((void **)&p, nc * sizeof(T)) != hipSuccess) { set_err("hipMalloc failed"); return FXRX_ERR_HIP; }
        cap = nc; return 0;
    }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
template <class T> struct PinBuf {
    T *p = nullptr; size_t cap = 0;
    int reserve(size_t n)
    {
        if (n <= cap) return 0;
        size_t nc = std::max(n, cap + cap / 2);
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc((void **)&p, nc * sizeof(T), hipHostMallocDefault) != hipSuccess) { set_err("hipHostMalloc failed"); return FXRX_ERR_HIP; }
        cap = nc; return 0;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
};

struct StreamState {
    DevBuf<float2> carry[2]; int cur = 0;   // double-buffered tail of the previous block
    size_t carry_len = 0;
    int64_t total = 0;                      // absolute index of the next new sample (since the last reset)
    int64_t pos = 0, floor_ = 0; bool fresh = true;   // resume state, relative to carry sample 0
    // [headroom | carry | new] staging when a tail exists, input is on the host, or the block is walked speculatively.
    // Two buffers, used alternately: the payload MF of block k may still be reading one while block k+1 is staged.
    DevBuf<float2> work[2]; int wcur = 0;
    hipEvent_t work_mf[2] = { nullptr, nullptr };   // end of the last MF that read work[i] (borrowed from its slot)
    hipEvent_t work_rd[2] = { nullptr, nullptr };   // end of the last tail copy that read work[i] (likewise)
    hipEvent_t carry_ev = nullptr;                  // end of the copy that filled carry[cur] (borrowed from its slot)
    size_t max_keep = 0;                            // longest tail carried so far (sizes the speculative headroom)
};

struct PlanKey { unsigned n, check, fec0, fec1; bool operator<(const PlanKey &o) const { return std::tie(n, check, fec0, fec1) < std::tie(o.n, o.check, o.fec0, o.fec1); } };
struct PlanDev { fx::PacketPlan plan; uint32_t perm0_off, perm1_off; };

}  // namespace

struct Out { fxrx_frame f; int pjob; };

// One stream's stitched result for a block.  A chain also lists its seek spans: runs of hops [pos, end) on which the
// walkers reported no detection; with hop skipping enabled these are what fx_seekverify_kernel re-checks.
struct Span { int64_t pos, floor_, end; };
struct Chain { std::vector<FxFrame> frames; std::vector<Span> spans; int64_t pos = 0, floor_ = 0; bool fresh = true; };

// One in-flight block's payload stage: its own arenas, staging and result buffers, so that the PLL of
// block n, the packet decode of block n-1 and the walk of block n+1 can run concurrently on three streams.
struct Slot {
    std::vector<FxPayJob> pjobs; std::vector<uint32_t> blk_job, blk_c0, pll_idx;
    // payload-stage descriptors, one arena = one upload: [FxPayJob x NP | blk_job | blk_c0 | pll_idx | dec_idx]
    DevBuf<uint8_t> d_meta; PinBuf<uint8_t> hp_meta;
    DevBuf<float2> d_symraw, d_framesyms; DevBuf<uint8_t> d_hard, d_bufA, d_bufB, d_out; DevBuf<unsigned long long> d_dw;
    DevBuf<FxPayResult> d_pres; PinBuf<FxPayResult> h_pres; PinBuf<uint8_t> h_out; PinBuf<float2> h_framesyms;
    std::vector<Out> out;
    uint64_t n_syms = 0;
    fxrx_timing timing{};
    hipEvent_t ev_mf0 = nullptr, ev_mf1 = nullptr, ev_pll0 = nullptr, ev_pll1 = nullptr, ev_dec0 = nullptr, ev_dec1 = nullptr, ev_done = nullptr;
    hipStream_t stream_p = nullptr, stream_d = nullptr;   // borrowed from the context (see fxrx_ctx_s)
    bool busy = false;
    // walk stage.  Per block, because the walk of a block may be launched before the previous block's walk has been
    // stitched (see fxrx_submit): job descriptors, results and frame tables are pinned host memory the kernels address
    // directly; xs / ns / first_job / NJ are what the completion phase needs to know about the launch.
    std::vector<FxWalkJob> jobs; std::vector<uint32_t> job_stream;
    PinBuf<FxWalkResult> h_res; PinBuf<FxFrame> h_frames; PinBuf<FxWalkJob> hp_jobs;
    PinBuf<FxVerifyJob> hp_vjobs; PinBuf<FxVerifyResult> h_vres;
    hipEvent_t ev_w0 = nullptr, ev_w1 = nullptr, ev_v0 = nullptr, ev_v1 = nullptr, ev_carry = nullptr, ev_j0 = nullptr;
    bool wait_j0 = false;                // true walkers were launched on the priority stream: ev_j0 marks their end
    unsigned index = 0;                  // position in the ring of slots
    hipStream_t stream_w = nullptr;      // the walk stream this block uses (borrowed)
    std::vector<const float2 *> xs; std::vector<int64_t> ns; std::vector<size_t> first_job;
    size_t NJ = 0; uint32_t repair_base = 0, repair_cap = 256;
    uint64_t epoch = 0;                  // fxrx_reset generation the block was submitted in
    // a block's way through the host: walk_phase -> (WALKING) -> stitch_phase -> (VERIFYING) -> finish_phase -> (LAUNCHED)
    enum { IDLE = 0, WALKING = 1, VERIFYING = 2, LAUNCHED = 3 };
    int stage = IDLE;
    std::vector<Chain> chains;           // stitched per stream (stitch_phase), consumed by finish_phase
    std::vector<int64_t> base;           // absolute sample index of coordinate 0 of xs[s]
    std::vector<int> wbuf;               // which of the stream's work buffers xs[s] is (-1: the caller's buffer)
    // speculative block of a continuing stream (see fxrx_submit): the true walker of every stream (its job 0) is
    // launched by stitch_phase, once the previous block has left its resume state and tail behind
    bool late0 = false; int64_t headroom = 0;
    std::vector<int64_t> tot0;           // absolute index of the block's first new sample, per stream
    std::vector<const void *> in_ptr; std::vector<uint64_t> in_n; int in_dev = 0;   // the input, for a re-stage
    std::vector<unsigned> vj_stream;     // stream of every verification run launched by stitch_phase
};

struct fxrx_ctx_s {
    fxrx_config cfg{};
    // HIP multiplexes streams onto a few hardware queues (a kernel trace of this process shows three usable
    // ones); streams that share a queue serialise.  So exactly three: W, and two payload streams used
    // alternately by consecutive blocks, each running its block's PLL -> decode -> result copies in order.
    hipStream_t stream = nullptr;        // W: input staging, walker, payload MF, tail carry
    hipStream_t stream_p[16] = {};       // payload stage (MF -> PLL -> decode -> result copies), tied to the slots
    unsigned n_pstreams = 0;             // created so far
    bool pstreams_fixed = false;         // FXRX_PAYLOAD_STREAMS given: do not grow with the pipeline depth
    bool psplit = false; std::vector<uint32_t> pmask;   // CU mask of the payload streams (FXRX_PAYLOAD_SPLIT / FXRX_RESERVE_CUS)
    unsigned pll_waves = 1, dec_waves = 1;   // waves per workgroup of the PLL / decode grids (placement only)
    unsigned pll_stagger = 32;               // blocks in flight start their PLL grids this many workgroup slots apart (0 = off)
    int n_cus = 256;
    hipStream_t stream2 = nullptr;       // second walk stream: consecutive independent blocks alternate (see fxrx_submit)
    bool early_walk = true;              // FXRX_EARLY_WALK=0: never launch a walk before the previous block is stitched
    uint64_t epoch = 0;     // fxrx_reset generation; blocks submitted so far
    std::deque<struct Slot *> pending;   // blocks whose payload stage is not launched yet, oldest first
    FxTables *d_tables = nullptr;
    std::vector<StreamState> st;
    bool skip_seek = true;               // FXRX_SKIP_SEEK=0: walkers run the full detector on every hop (no verification pass)
    // packet plans (shared, append-only)
    std::map<PlanKey, PlanDev> plans; std::vector<uint32_t> perm_host; DevBuf<uint32_t> d_perm; size_t perm_uploaded = 0;
    // pipeline
    std::vector<std::unique_ptr<Slot>> slots; unsigned depth = 1, head = 0, tail = 0, inflight = 0;
    Slot *last = nullptr;                // slot whose results are currently exposed through fxrx_result
    uint64_t walk_stamp[4] = { 0, 0, 0, 0 };   // diagnostic builds: summed walker phase clocks of the last submit
    uint64_t walk_stamp_max = 0, walk_stamp_maxjob[4] = { 0, 0, 0, 0 }, walk_maxjob_hops = 0, walk_maxjob_cheap = 0, walk_maxjob_frames = 0;
};

namespace {

int upload_tables(fxrx_ctx_s *c)
{
    const fx::HostTables &H = fx::host_tables();
    const fx::BlockCodes &B = fx::block_codes();
    std::unique_ptr<FxTables> t(new FxTables);
    std::memset(t.get(), 0, sizeof(FxTables));
    for (int i = 0; i < 512; i++) { t->tw[i] = make_float2(H.tw[i].re, H.tw[i].im); t->S[i] = make_float2(H.S[i].re, H.S[i].im); }
    for (int i = 0; i < 1024; i++) t->sc[i] = make_float2(H.sc[i].re, H.sc[i].im);
    for (int i = 0; i < FX_S_LEN; i++) t->s[i] = make_float2(H.s[i].re, H.s[i].im);
    for (int i = 0; i < FX_HDR_PILOTS; i++) t->pilots[i] = make_float2(H.pilots[i].re, H.pilots[i].im);
    std::memcpy(t->proto, H.proto, sizeof H.proto);
    t->s2sum = H.s2sum;
    {   // differential template for the speculative walkers' coarse scan
        fx::cf td[512]; std::memset(td, 0, sizeof td);
        float e = 0.0f, mr = 0.0f, mi = 0.0f;
        const int nd = FX_S_LEN - 1;
        for (int k = 0; k < nd; k++) {
            td[k].re = H.s[k + 1].re * H.s[k].re + H.s[k + 1].im * H.s[k].im;
            td[k].im = H.s[k + 1].im * H.s[k].re - H.s[k + 1].re * H.s[k].im;
            mr += td[k].re; mi += td[k].im;
        }
        mr /= (float)nd; mi /= (float)nd;                       // zero-mean over its support (see the kernel)
        for (int k = 0; k < nd; k++) { td[k].re -= mr; td[k].im -= mi; e += td[k].re * td[k].re + td[k].im * td[k].im; }
        fx::cf TD[512]; fx::hfft::fft512(td, TD, H.tw);
        for (int i = 0; i < 512; i++) t->TD[i] = make_float2(TD[i].re, TD[i].im);
        t->td2sum = e;
    }
    for (size_t i = 0; i < H.perm54.size(); i++) t->perm54[i] = (uint16_t)H.perm54[i];
    for (size_t i = 0; i < H.perm27.size(); i++) t->perm27[i] = (uint16_t)H.perm27[i];
    std::memcpy(t->h84dec, B.h84_dec, 256); std::memcpy(t->sdcol, B.sd_col, 64);
    std::memcpy(t->sd22col, B.sd22_col, 16); std::memcpy(t->sd39col, B.sd39_col, 32);
    std::memcpy(t->h74dec, B.h74_dec, 128); std::memcpy(t->h128dec, B.h128_dec, 4096);
    std::memcpy(t->golenc, B.gol_enc, sizeof B.gol_enc); std::memcpy(t->golerr, B.gol_err, sizeof B.gol_err);
    std::memcpy(t->rsexp, B.rs_exp, 512); std::memcpy(t->rslog, B.rs_log, 256);
    HIP_OK(hipMalloc((void **)&c->d_tables, sizeof(FxTables)));
    HIP_OK(hipMemcpy(c->d_tables, t.get(), sizeof(FxTables), hipMemcpyHostToDevice));
    return 0;
}

const PlanDev &get_plan(fxrx_ctx_s *c, unsigned n, unsigned check, unsigned fec0, unsigned fec1)
{
    PlanKey k{ n, check, fec0, fec1 };
    auto it = c->plans.find(k);
    if (it != c->plans.end()) return it->second;
    PlanDev pd; pd.plan = fx::packet_plan(n, check, fec0, fec1);
    std::vector<uint32_t> g0 = fx::Interleaver(pd.plan.l0).decode_gather(), g1 = fx::Interleaver(pd.plan.l1).decode_gather();
    pd.perm0_off = (uint32_t)c->perm_host.size(); c->perm_host.insert(c->perm_host.end(), g0.begin(), g0.end());
    pd.perm1_off = (uint32_t)c->perm_host.size(); c->perm_host.insert(c->perm_host.end(), g1.begin(), g1.end());
    return c->plans.emplace(k, pd).first->second;
}

// frame-table slots per walk job: one per 2048 samples of segment (+8).  The table is copied to the host every
// block, so it is not sized for the densest legal traffic (a 650-sample frame); a segment with more detections than
// slots continues through the FX_EXIT_TABLE_FULL path.
inline uint32_t seg_frames_cap(uint64_t seg) { return (uint32_t)std::min<uint64_t>(seg / 2048 + 8, 512); }

int launch_walk(fxrx_ctx_s *c, Slot &sl, size_t first, size_t count, hipStream_t st = nullptr)
{
    if (!st) st = sl.stream_w;
    // Job descriptors, per-job results and frame tables live in pinned host memory that the kernel addresses
    // directly: each workgroup reads one descriptor and writes a handful of 128-byte records, so the PCIe hop costs
    // less than the three staging copies it replaces on the walk -> stitch critical path.
    std::memcpy(sl.hp_jobs.p + first, sl.jobs.data() + first, count * sizeof(FxWalkJob));
    HIP_OK(fx_launch_walk(sl.jobs[first].mode, (unsigned)count, st, sl.hp_jobs.p + first, sl.h_res.p + first, sl.h_frames.p, c->d_tables));
    return 0;
}

}  // namespace

// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and streams sharing a queue
// serialise; the pipeline wants one queue per stream (2 walk + one payload stream per block in flight).  The runtime
// reads the variable when it initialises, so this only helps when the library is loaded before the first HIP call.
__attribute__((constructor)) static void fxrx_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// ============================================================================ batched API
extern "C" {

const char *fxrx_last_error(void) { return g_err.c_str(); }
void fxrx_set_error(const char *msg) { set_err(msg ? msg : ""); }      // for the library's other translation units
const char *fxrx_version(void) { return "fxrx 0.1 (gfx950)"; }
int fxrx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

static int ensure_pstreams(fxrx_ctx_s *c, unsigned want)
{
    for (; c->n_pstreams < std::min(want, 16u); c->n_pstreams++) {
        hipStream_t *st = &c->stream_p[c->n_pstreams];
        HIP_OK(c->psplit ? hipExtStreamCreateWithCUMask(st, (uint32_t)c->pmask.size(), c->pmask.data())
                         : hipStreamCreateWithFlags(st, hipStreamNonBlocking));
    }
    return 0;
}

static int make_slot(fxrx_ctx_s *c)
{
    std::unique_ptr<Slot> s(new Slot);
    hipEvent_t *ev[13] = { &s->ev_mf0, &s->ev_mf1, &s->ev_pll0, &s->ev_pll1, &s->ev_dec0, &s->ev_dec1, &s->ev_done,
                           &s->ev_w0, &s->ev_w1, &s->ev_v0, &s->ev_v1, &s->ev_carry, &s->ev_j0 };
    for (auto e : ev) HIP_OK(hipEventCreate(e));
    s->index = (unsigned)c->slots.size();
    s->stream_p = c->stream_p[c->slots.size() % c->n_pstreams]; s->stream_d = s->stream_p;
    c->slots.push_back(std::move(s));
    return 0;
}

fxrx_ctx *fxrx_create(const fxrx_config *cfg)
{
    if (!cfg || cfg->n_streams == 0) { set_err("fxrx_create: bad config"); return nullptr; }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || cfg->device >= nd) {
        set_err("fxrx_create: no usable HIP device (this library has no CPU path)"); return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
    std::unique_ptr<fxrx_ctx_s> c(new fxrx_ctx_s);
    c->cfg = *cfg;
    if (c->cfg.threshold <= 0.0f) c->cfg.threshold = cfg->mode == FXRX_MODE_DETECTOR ? 0.45f : 0.5f;
    unsigned want_pstreams = 1;          // one per block in flight (fxrx_set_depth adds more) unless the environment says otherwise
    if (const char *e = std::getenv("FXRX_PAYLOAD_STREAMS")) { want_pstreams = (unsigned)std::min(16, std::max(1, std::atoi(e))); c->pstreams_fixed = true; }
    if (const char *e = std::getenv("FXRX_PLL_WAVES")) c->pll_waves = (unsigned)std::min(4, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_PLL_STAGGER")) c->pll_stagger = (unsigned)std::min(256, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_DEC_WAVES")) c->dec_waves = (unsigned)std::min(8, std::max(1, std::atoi(e)));
    {
        // Two flex_rx walker workgroups (4 waves x 256 VGPRs each) fill a CU's register file.  Keeping the walker off a few CUs
        // (FXRX_WALK_CUS=<n>; bench.py uses 224 of 256) leaves room where the latency-critical PLL / decode
        // waves of the blocks in flight always find a slot at once.
        hipDeviceProp_t prop; int ncu = 256;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ncu = prop.multiProcessorCount;
        c->n_cus = ncu;
        int want = ncu;                                              // default: no mask (best for large batches)
        if (const char *e = std::getenv("FXRX_WALK_CUS")) want = std::atoi(e);
        if (want > 0 && want < ncu) c->n_cus = want;
        hipError_t err;
        if (want > 0 && want < ncu) {
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int i = 0; i < want; i++) mask[(size_t)i / 32] |= 1u << (i % 32);
            err = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data());
            if (err == hipSuccess) err = hipExtStreamCreateWithCUMask(&c->stream2, (uint32_t)mask.size(), mask.data());
        } else {
            err = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            if (err == hipSuccess) err = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
        }
        if (err != hipSuccess) { set_err("hipStreamCreate failed"); return nullptr; }
    }
    {
        // FXRX_PAYLOAD_SPLIT=1 (needs FXRX_WALK_CUS): the payload streams get exactly the CUs the walker leaves alone.
        // A walker workgroup needs a whole, empty register file; decode waves of blocks in flight scattered over
        // every CU make it wait for CUs to drain, so a hard partition can beat sharing.
        const char *e = std::getenv("FXRX_PAYLOAD_SPLIT");
        int ncu = 256; hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) ncu = prop.multiProcessorCount;
        bool split = e && std::atoi(e) != 0 && c->n_cus < ncu;
        int first = c->n_cus;
        // FXRX_RESERVE_CUS=<r>: payload streams stay off the first r CUs, which the (unmasked) walker then always finds empty
        if (const char *r = std::getenv("FXRX_RESERVE_CUS")) { const int v = std::atoi(r); if (v > 0 && v < ncu) { split = true; first = v; } }
        c->psplit = split; c->pmask.assign((size_t)(ncu + 31) / 32, 0u);
        for (int i = first; i < ncu; i++) c->pmask[(size_t)i / 32] |= 1u << (i % 32);
        if (ensure_pstreams(c.get(), want_pstreams) != 0) return nullptr;
    }
    if (const char *e = std::getenv("FXRX_SKIP_SEEK")) c->skip_seek = std::atoi(e) != 0;
    if (const char *e = std::getenv("FXRX_EARLY_WALK")) c->early_walk = std::atoi(e) != 0;
    if (upload_tables(c.get()) != 0) return nullptr;
    c->st.resize(cfg->n_streams);
    if (make_slot(c.get()) != 0) return nullptr;
    return c.release();
}

static void sync_all(fxrx_ctx_s *c)
{
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    for (auto s : c->stream_p) if (s) (void)hipStreamSynchronize(s);
}

void fxrx_destroy(fxrx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    sync_all(c);
    for (auto &s : c->slots) {
        hipEvent_t ev[13] = { s->ev_mf0, s->ev_mf1, s->ev_pll0, s->ev_pll1, s->ev_dec0, s->ev_dec1, s->ev_done, s->ev_w0, s->ev_w1, s->ev_v0, s->ev_v1, s->ev_carry, s->ev_j0 };
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
    }
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (auto s : c->stream_p) if (s) (void)hipStreamDestroy(s);
    delete c;
}

static int advance_front(fxrx_ctx_s *c);

// forget all per-stream state (position, carried tail).  In-flight blocks are unaffected: a block submitted before the
// reset no longer writes its resume state back (the epoch tells), and a pending speculative block, whose true walker
// still needs the state its predecessor leaves behind, is run through its host phases first.
void fxrx_reset(fxrx_ctx *c)
{
    if (!c) return;
    bool needs_state = false;
    for (Slot *p : c->pending) if (p->late0) needs_state = true;
    if (needs_state) {
        (void)hipSetDevice(c->cfg.device);
        while (!c->pending.empty()) if (advance_front(c)) break;
    }
    c->epoch++;
    for (auto &s : c->st) { s.carry_len = 0; s.total = 0; s.pos = 0; s.floor_ = 0; s.fresh = true; s.carry_ev = nullptr; }
    // (work-buffer guards stay: blocks in flight may still be reading them)
}

int fxrx_set_depth(fxrx_ctx *c, unsigned int depth)
{
    if (!c || depth == 0 || depth > 16) return FXRX_ERR_ARG;
    if (c->inflight) { set_err("fxrx_set_depth: blocks in flight"); return FXRX_ERR_STATE; }
    HIP_OK(hipSetDevice(c->cfg.device));
    if (!c->pstreams_fixed && ensure_pstreams(c, depth) != 0) return FXRX_ERR_HIP;     // one payload stream per block in flight
    while (c->slots.size() < depth) if (make_slot(c) != 0) return FXRX_ERR_HIP;
    for (auto &sl : c->slots) sl->stream_p = sl->stream_d = c->stream_p[sl->index % c->n_pstreams];
    c->depth = depth; c->head = c->tail = 0;
    return 0;
}

void *fxrx_stream(const fxrx_ctx *c) { return c ? (void *)c->stream : nullptr; }

int fxrx_last_timing(const fxrx_ctx *c, fxrx_timing *t) { if (!c || !t || !c->last) return FXRX_ERR_ARG; *t = c->last->timing; return 0; }

const void *fxrx_device_framesyms(const fxrx_ctx *c, uint64_t *n)
{
    if (!c || !c->last) return nullptr;
    if (n) *n = c->last->n_syms;
    return c->last->n_syms ? c->last->d_framesyms.p : nullptr;
}

// ---- phase 1 of a block: stage its input, cut it into segments, launch the walkers (nothing is waited for) ----
enum { WALK_SERIAL = 0, WALK_SPEC = 1, WALK_RESTAGE = 2 };

// mode WALK_SERIAL: the streams' resume state is known (nothing pending, or the streams were reset): job 0 of every
//   stream is the true walker, everything is launched at once.
// mode WALK_SPEC: a continuing stream whose previous block is still pending.  The new samples are staged behind a
//   headroom that will take the tail once it is known, the speculative walkers of all segments are launched at once,
//   and job 0 is a placeholder that stitch_phase replaces by the true walker (a short walk up to its hand-off).
// mode WALK_RESTAGE: a WALK_SPEC block whose tail turned out longer than the headroom: staged and walked again, serially.
static int walk_phase(fxrx_ctx_s *c, Slot &sl, const void *const *iq, const uint64_t *n_samples, int on_device, int mode)
{
    const unsigned NS = c->cfg.n_streams;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    const auto t_enter = std::chrono::steady_clock::now();
    if (mode != WALK_RESTAGE) {
        sl.out.clear(); sl.timing = fxrx_timing{};
        // Streams are tied to slots, not to blocks: the HIP runtime tracks which queue last touched a buffer that takes part
        // in a hipMemcpyAsync and makes the next queue wait for the previous one, so handing a slot's arenas to a different
        // stream every time serialises the payload stages (measured: 10.7 instead of 17.8 Gsamples/s).
        sl.stream_w = (sl.index & 1u) ? c->stream2 : c->stream;
        sl.epoch = c->epoch;
        sl.in_ptr.assign(iq, iq + NS); sl.in_n.assign(n_samples, n_samples + NS); sl.in_dev = on_device;
    }
    const bool spec = mode == WALK_SPEC;
    sl.late0 = spec;
    std::vector<const float2 *> &xs = sl.xs; std::vector<int64_t> &ns = sl.ns; std::vector<size_t> &first_job = sl.first_job;
    const std::vector<int> wbuf_prev = sl.wbuf;
    xs.assign(NS, nullptr); ns.assign(NS, 0); first_job.assign(NS + 1, 0); sl.base.assign(NS, 0); sl.wbuf.assign(NS, -1);
    if (mode != WALK_RESTAGE) sl.tot0.assign(NS, 0);

    // ---- 1. per-stream work buffers: [headroom | tail of the previous block | new samples] ----
    uint64_t total_new = 0;
    int64_t H = 0;
    if (spec) {
        size_t mk = 0; for (const auto &S : c->st) mk = std::max(mk, std::max(S.max_keep, S.carry_len));
        H = (int64_t)std::max<size_t>(131072, 2 * mk + 4096);
    }
    sl.headroom = H;
    for (unsigned s = 0; s < NS; s++) {
        StreamState &S = c->st[s];
        const uint64_t nn = n_samples[s];
        total_new += nn;
        if (mode != WALK_RESTAGE) { sl.tot0[s] = S.total; S.total += (int64_t)nn; }
        const int64_t total_before = sl.tot0[s];
        if (!spec && S.carry_len == 0 && on_device) { xs[s] = (const float2 *)iq[s]; ns[s] = (int64_t)nn; sl.base[s] = total_before; continue; }
        const int b = (mode == WALK_RESTAGE && s < wbuf_prev.size() && wbuf_prev[s] >= 0) ? wbuf_prev[s] : (S.wcur ^= 1);
        DevBuf<float2> &W = S.work[b];
        // the payload MF and the tail copy of the block that used this buffer last may still be reading it (other streams)
        if (S.work_mf[b]) { HIP_OK(hipStreamWaitEvent(sl.stream_w, S.work_mf[b], 0)); S.work_mf[b] = nullptr; }
        if (S.work_rd[b]) { HIP_OK(hipStreamWaitEvent(sl.stream_w, S.work_rd[b], 0)); S.work_rd[b] = nullptr; }
        const size_t lead = spec ? (size_t)H : S.carry_len;
        if (W.reserve(lead + nn + 1)) return FXRX_ERR_HIP;
        if (!spec && S.carry_len) {
            if (S.carry_ev) HIP_OK(hipStreamWaitEvent(sl.stream_w, S.carry_ev, 0));
            HIP_OK(hipMemcpyAsync(W.p, S.carry[S.cur].p, S.carry_len * sizeof(float2), hipMemcpyDeviceToDevice, sl.stream_w));
        }
        if (nn) HIP_OK(hipMemcpyAsync(W.p + lead, iq[s], nn * sizeof(float2), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, sl.stream_w));
        xs[s] = W.p; ns[s] = (int64_t)(lead + nn); sl.base[s] = total_before - (int64_t)lead; sl.wbuf[s] = b;
    }
    sl.timing.samples = total_new; sl.timing.walk_mode = (uint64_t)mode;

    // ---- 2. walk jobs: cut every stream into segments ----
    uint64_t seg = c->cfg.segment_len;
    if (seg == 0) {
        uint64_t tot = 0;
        for (unsigned s = 0; s < NS; s++) tot += spec ? n_samples[s] : (uint64_t)std::max<int64_t>(0, ns[s] - c->st[s].pos);
        // Walker workgroups resident at once: two per CU for the flex_rx instance (4 waves x 256 VGPRs each), two for the
        // leaner detector-only instance.  Little work: a single round of workgroups with a small margin (the kernel
        // then lasts as long as its slowest segment).  Lots of work: ~4 rounds so that uneven segments even out.
        const uint64_t slots = (uint64_t)c->n_cus * 2u;
        if (tot / slots < 131072) seg = tot / (slots - slots / 16);
        else seg = std::max<uint64_t>(tot / (4 * slots), 65536);
        seg = std::max<uint64_t>(seg, 32768); seg = std::min<uint64_t>(seg, 1u << 20);
    }
    seg = std::max<uint64_t>(seg, 4096);
    sl.jobs.clear(); sl.job_stream.clear();
    uint32_t frame_slots = 0;
    for (unsigned s = 0; s < NS; s++) {
        StreamState &S = c->st[s];
        first_job[s] = sl.jobs.size();
        // speculative block: the true walker only has to reach its first hand-off, so its own segment is short
        int64_t p = spec ? H : S.pos;
        const int64_t seg0 = spec ? (int64_t)std::min<uint64_t>(seg, 8192) : (int64_t)seg;
        bool first = true;
        while (first || p < ns[s]) {
            FxWalkJob j{};
            j.x = xs[s]; j.n = ns[s]; j.start = p;
            j.stop = std::min<int64_t>(ns[s], p + (first ? seg0 : (int64_t)seg));
            if (ns[s] - j.stop < (int64_t)seg / 2) j.stop = ns[s];          // fold a short last segment in
            j.fresh = first ? (S.fresh ? 1u : 0u) : 1u;
            j.floor = first ? S.floor_ : p;
            j.mode = detect ? FX_MODE_DETECT : FX_MODE_FLEXRX;
            j.handoff = j.stop < ns[s] ? 1u : 0u;
            j.prelock = first ? 0u : 1u;
            j.frame_base = frame_slots; j.max_frames = seg_frames_cap((uint64_t)(j.stop - j.start) + (first && spec ? 65536u : 0u));
            frame_slots += j.max_frames;
            j.threshold = c->cfg.threshold;
            j.no_skip = (detect || !c->skip_seek) ? 1u : 0u;
            if (first && spec) { j.start = j.stop; j.fresh = 1u; j.floor = j.stop; j.handoff = 0u; }   // placeholder: exits at once
            sl.jobs.push_back(j); sl.job_stream.push_back(s);
            p = j.stop; first = false;
            if (p >= ns[s]) break;
        }
    }
    first_job[NS] = sl.jobs.size();
    const size_t NJ = sl.jobs.size();
    // one spare job slot + frame region for repairs
    sl.NJ = NJ; sl.repair_base = frame_slots; sl.repair_cap = 256;
    frame_slots += sl.repair_cap;
    if (sl.h_res.reserve(NJ + 1) || sl.hp_jobs.reserve(NJ + 1) || sl.h_frames.reserve(frame_slots)) return FXRX_ERR_HIP;
    sl.jobs.resize(NJ + 1);
    HIP_OK(hipEventRecord(sl.ev_w0, sl.stream_w));
    if (launch_walk(c, sl, 0, NJ)) return FXRX_ERR_HIP;
    HIP_OK(hipEventRecord(sl.ev_w1, sl.stream_w));
    sl.timing.walk_jobs = NJ;
    sl.stage = Slot::WALKING;
    const double dt = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    if (mode == WALK_RESTAGE) sl.timing.host_submit_ms += dt; else sl.timing.host_submit_ms = dt;
    return 0;
}

// ---- stitch one stream: splice the speculative lists into the sequential chain ----
static int stitch_stream(fxrx_ctx_s *c, Slot &sl, unsigned s)
{
    const std::vector<size_t> &first_job = sl.first_job;
    const size_t NJ = sl.NJ; const uint32_t repair_base = sl.repair_base, repair_cap = sl.repair_cap;
    {
        Chain &ch = sl.chains[s];
        ch.frames.clear(); ch.spans.clear();
        auto add_span = [&](int64_t p, int64_t fl, int64_t e) { if (e > p) ch.spans.push_back(Span{ p, fl, e }); };
        size_t cur = first_job[s]; uint32_t m = 0;
        FxWalkResult R = sl.h_res.p[cur]; const FxFrame *F = sl.h_frames.p + sl.jobs[cur].frame_base;
        std::vector<FxFrame> repair_frames; float splice_rxy = -1.0f;
        bool spliced = false; int64_t tpos = 0, tfloor = 0; bool tfresh = true;   // true-chain state at the last splice
        auto run_repair = [&](const FxWalkJob &tmpl, const FxWalkResult &from) -> int {
            FxWalkJob j = tmpl; j.start = from.pos; j.fresh = from.fresh; j.floor = from.floor; j.prelock = 0;
            j.frame_base = repair_base; j.max_frames = repair_cap;
            sl.jobs[NJ] = j;
            if (launch_walk(c, sl, NJ, 1)) return FXRX_ERR_HIP;
            HIP_OK(hipStreamSynchronize(sl.stream_w));
            sl.timing.repairs++;
            R = sl.h_res.p[NJ]; repair_frames.assign(sl.h_frames.p + repair_base, sl.h_frames.p + repair_base + R.n_frames);
            F = repair_frames.data(); m = 0;
            return 0;
        };
        for (;;) {
            sl.timing.hops += R.hops; sl.timing.hops_cheap += R.hops_cheap;
            for (int i = 0; i < 4; i++) c->walk_stamp[i] += R.stamp[i];
            { uint64_t tot = (uint64_t)R.stamp[0] + R.stamp[1] + R.stamp[2] + R.stamp[3]; if (tot > c->walk_stamp_max) { c->walk_stamp_max = tot; for (int i = 0; i < 4; i++) c->walk_stamp_maxjob[i] = R.stamp[i]; c->walk_maxjob_hops = R.hops; c->walk_maxjob_cheap = R.hops_cheap; c->walk_maxjob_frames = R.n_frames; } }
            uint32_t nf = R.n_frames;
            if (R.exit_code == FX_EXIT_PAYLOAD && nf > 0) nf--;            // incomplete frame: redo next call
            for (uint32_t i = m; i < nf; i++) {
                if (!(F[i].flags & FX_FLAG_EXACT)) continue;               // tentative pre-lock entries of a speculative walk
                ch.frames.push_back(F[i]);
                if (i == m && splice_rxy >= 0.0f) ch.frames.back().rxy = splice_rxy;   // coarse peak as the true chain saw it
                else add_span(F[i].seek_pos, F[i].seek_floor, F[i].det_pos);            // (a spliced frame's seek is the hand-off's)
            }
            // the seek in progress when this walker stopped (a spliced-in walker that contributed nothing is not on the chain)
            if (!(spliced && nf <= m)) add_span(R.tail_pos, R.tail_floor, R.has_handoff ? R.handoff_pos : R.pos);
            splice_rxy = -1.0f;
            const bool last = (cur + 1 == first_job[s + 1]);
            if (R.exit_code == FX_EXIT_TABLE_FULL) {                       // continue the same segment where the table filled up
                const FxWalkResult from = R;
                if (run_repair(sl.jobs[cur], from)) return FXRX_ERR_HIP;
                continue;
            }
            if (last || R.exit_code != FX_EXIT_STOP || !R.has_handoff) {
                if (spliced && nf <= m) {
                    // the spliced-in walker added nothing complete (its matched frame runs past the data):
                    // its own hop state is speculative, so resume from the true chain's hand-off hop instead
                    ch.pos = tpos; ch.floor_ = tfloor; ch.fresh = tfresh;
                } else { ch.pos = R.pos; ch.floor_ = R.floor; ch.fresh = R.fresh != 0; }
                break;
            }
            spliced = false;
            // hand-off: look the target up in the next segment's speculative list
            // (segments the true walker crossed without a detection cannot hold the target: skip them)
            size_t nxt = cur + 1;
            while (nxt + 1 < first_job[s + 1] && R.handoff_start >= sl.jobs[nxt].stop + FX_HOP) nxt++;
            const FxWalkResult &RN = sl.h_res.p[nxt]; const FxFrame *FN = sl.h_frames.p + sl.jobs[nxt].frame_base;
            uint32_t found = UINT32_MAX;
            // A frame is a function of (start, CFO bin) alone only if no sample it reads was masked by a zero-floor: the
            // true chain may detect with start < floor (window half zeros after a reset), a speculative walker reaches the
            // same (start, bin) with another floor.  Splice only when both floors lie at or below the start; else repair.
            for (uint32_t i = 0; R.handoff_clear && i < RN.n_frames; i++)
                if ((FN[i].flags & FX_FLAG_EXACT) && (FN[i].flags & FX_FLAG_FLOOR_CLEAR) && FN[i].start == R.handoff_start && FN[i].offset == R.handoff_offset) { found = i; break; }
            if (found != UINT32_MAX) {
                splice_rxy = R.handoff_rxy; spliced = true; tpos = R.pos; tfloor = R.floor; tfresh = R.fresh != 0;
                cur = nxt; m = found; R = RN; F = FN; continue;
            }
            // repair: walk the next segment from the true state
            const FxWalkResult from = R;
            if (run_repair(sl.jobs[nxt], from)) return FXRX_ERR_HIP;
            cur = nxt;
        }
        return 0;
    }
}

// Speculative block of a continuing stream, once the previous block is finished (its tail and resume state are known):
// put the tail in front of the new samples and launch every stream's true walker (job 0).  Called as early as possible
// -- before the next block's speculative walkers are launched -- so that these few workgroups find free CUs at once.
static int launch_true_walkers(fxrx_ctx_s *c, Slot &sl)
{
    if (!sl.late0) return 0;
    const unsigned NS = c->cfg.n_streams;
    std::vector<const float2 *> &xs = sl.xs; std::vector<int64_t> &ns = sl.ns;
    {
        // Speculative block of a continuing stream: the previous block is finished by now, so the tail and the resume
        // state are known.  Put the tail in front of the new samples and launch every stream's true walker (job 0).
        bool fits = true;
        for (unsigned s = 0; s < NS; s++) if ((int64_t)c->st[s].carry_len > sl.headroom) fits = false;
        if (!fits) {                                  // (rare) a tail longer than the headroom: stage and walk again, serially
            HIP_OK(hipEventSynchronize(sl.ev_w1));
            if (walk_phase(c, sl, sl.in_ptr.data(), sl.in_n.data(), sl.in_dev, WALK_RESTAGE)) return FXRX_ERR_HIP;
        } else {
            // (a separate highest-priority stream for these few workgroups was tried: no measurable difference)
            hipStream_t st = sl.stream_w;
            for (unsigned s = 0; s < NS; s++) {
                StreamState &S = c->st[s];
                const int64_t c0 = sl.headroom - (int64_t)S.carry_len;          // coordinate of tail sample 0
                if (S.carry_len) {
                    if (S.carry_ev) HIP_OK(hipStreamWaitEvent(st, S.carry_ev, 0));
                    HIP_OK(hipMemcpyAsync(const_cast<float2 *>(xs[s]) + c0, S.carry[S.cur].p, S.carry_len * sizeof(float2), hipMemcpyDeviceToDevice, st));
                }
                FxWalkJob &j = sl.jobs[sl.first_job[s]];
                j.start = c0 + S.pos; j.floor = c0 + S.floor_; j.fresh = S.fresh ? 1u : 0u;
                j.handoff = j.stop < ns[s] ? 1u : 0u;
                if (launch_walk(c, sl, sl.first_job[s], 1, st)) return FXRX_ERR_HIP;
            }
            HIP_OK(hipEventRecord(sl.ev_j0, st));
            sl.wait_j0 = true;
            sl.late0 = false;
        }
    }
    return 0;
}

// ---- phase 2: wait for the walkers, stitch every stream's chain, launch the seek verification ----
static int stitch_phase(fxrx_ctx_s *c, Slot &sl)
{
    const unsigned NS = c->cfg.n_streams;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    const auto t_enter = std::chrono::steady_clock::now();
    std::vector<const float2 *> &xs = sl.xs; std::vector<int64_t> &ns = sl.ns;
    for (auto &w : c->walk_stamp) w = 0;
    c->walk_stamp_max = 0;
    if (launch_true_walkers(c, sl)) return FXRX_ERR_HIP;
    {
        // (events, not the stream: the walk of the block after next may already be queued behind this one's)
        const auto tw = std::chrono::steady_clock::now();
        HIP_OK(hipEventSynchronize(sl.ev_w1));
        if (sl.wait_j0) { HIP_OK(hipEventSynchronize(sl.ev_j0)); HIP_OK(hipStreamWaitEvent(sl.stream_w, sl.ev_j0, 0)); sl.wait_j0 = false; }
        sl.timing.host_walkwait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
    }
    sl.chains.assign(NS, Chain{});
    for (unsigned s = 0; s < NS; s++) if (stitch_stream(c, sl, s)) return FXRX_ERR_HIP;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, sl.ev_w0, sl.ev_w1); sl.timing.walk_ms = ms;

    // seek verification: the full detector over every hop the chains' walkers skipped
    sl.vj_stream.clear();
    if (!detect && c->skip_seek) {
        const std::vector<Chain> &chains = sl.chains;
        uint64_t tot_hops = 0;
        for (unsigned s = 0; s < NS; s++) for (const Span &sp : chains[s].spans) tot_hops += (uint64_t)(sp.end - sp.pos) / FX_HOP;
        // runs of at most `per` hops: about four workgroups per CU, so that the grid drains evenly
        const uint64_t per = std::min<uint64_t>(16, std::max<uint64_t>(1, (tot_hops + 4ull * c->n_cus - 1) / (4ull * c->n_cus)));
        std::vector<FxVerifyJob> vj;
        for (unsigned s = 0; s < NS; s++)
            for (const Span &sp : chains[s].spans) {
                const uint64_t nh = (uint64_t)(sp.end - sp.pos) / FX_HOP;
                for (uint64_t h = 0; h < nh; h += per) {
                    FxVerifyJob j{};
                    j.x = xs[s]; j.n = ns[s]; j.pos = sp.pos + (int64_t)(h * FX_HOP); j.floor = sp.floor_;
                    j.nhops = (uint32_t)std::min<uint64_t>(per, nh - h); j.threshold = c->cfg.threshold;
                    vj.push_back(j); sl.vj_stream.push_back(s);
                }
            }
        sl.timing.verify_hops = tot_hops;
        if (!vj.empty()) {
            if (sl.hp_vjobs.reserve(vj.size()) || sl.h_vres.reserve(vj.size())) return FXRX_ERR_HIP;
            std::memcpy(sl.hp_vjobs.p, vj.data(), vj.size() * sizeof(FxVerifyJob));
            HIP_OK(hipEventRecord(sl.ev_v0, sl.stream_w));
            HIP_OK(fx_launch_seekverify((unsigned)vj.size(), sl.stream_w, sl.hp_vjobs.p, sl.h_vres.p, c->d_tables));
            HIP_OK(hipEventRecord(sl.ev_v1, sl.stream_w));
        }
    }
    sl.stage = Slot::VERIFYING;
    sl.timing.host_submit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    return 0;
}

// ---- phase 3: check the verification, launch the payload stage, carry the tails ----
static int finish_phase(fxrx_ctx_s *c, Slot &sl)
{
    const unsigned NS = c->cfg.n_streams;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    const auto t_enter = std::chrono::steady_clock::now();
    std::vector<const float2 *> &xs = sl.xs; std::vector<int64_t> &ns = sl.ns; std::vector<size_t> &first_job = sl.first_job;
    std::vector<Chain> &chains = sl.chains;
    float ms = 0;
    if (!sl.vj_stream.empty()) {
        {
            const auto tw = std::chrono::steady_clock::now();
            HIP_OK(hipEventSynchronize(sl.ev_v1));
            sl.timing.host_walkwait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
        }
        (void)hipEventElapsedTime(&ms, sl.ev_v0, sl.ev_v1); sl.timing.seekverify_ms = ms;
        // A skipped hop on which the detector does fire (a false alarm, or a preamble too weak for the coarse
        // scan): that stream's chain is void from there on.  Walk the stream again with skipping off -- exact
        // by itself, as in the first version of this walker -- and stitch it again.
        std::vector<char> bad(NS, 0); unsigned n_bad = 0;
        for (size_t i = 0; i < sl.vj_stream.size(); i++)
            if (sl.h_vres.p[i].det_hop != 0xFFFFFFFFu && !bad[sl.vj_stream[i]]) { bad[sl.vj_stream[i]] = 1; n_bad++; }
        if (n_bad) {
            sl.timing.verify_failures = n_bad;
            for (unsigned s = 0; s < NS; s++) {
                if (!bad[s]) continue;
                for (size_t j = first_job[s]; j < first_job[s + 1]; j++) sl.jobs[j].no_skip = 1u;
                if (launch_walk(c, sl, first_job[s], first_job[s + 1] - first_job[s])) return FXRX_ERR_HIP;
            }
            HIP_OK(hipStreamSynchronize(sl.stream_w));
            for (unsigned s = 0; s < NS; s++) if (bad[s] && stitch_stream(c, sl, s)) return FXRX_ERR_HIP;
        }
    }

    // ---- 4. payload jobs ----
    sl.pjobs.clear(); sl.blk_job.clear(); sl.blk_c0.clear();
    uint64_t sym_total = 0, byte_total = 0, dw_total = 0, out_total = 0;
    const size_t perm_before = c->perm_host.size();
    for (unsigned s = 0; s < NS; s++) {
        for (const FxFrame &f : chains[s].frames) {
            Out o{}; std::memset(&o.f, 0, sizeof o.f);
            o.f.stream = s; o.f.start = sl.base[s] + f.start; o.f.cfo_bin = f.offset;
            o.f.rxy = f.rxy; o.f.tau = f.tau; o.f.gamma = f.gamma; o.f.dphi = f.dphi; o.f.phi = f.phi; o.f.pfb_index = f.pfb;
            o.f.pilot_dphi = f.pilot_dphi; o.f.pilot_phi = f.pilot_phi; o.f.pilot_gain = f.pilot_gain;
            o.f.header_valid = (f.flags & FX_FLAG_HEADER_VALID) ? 1 : 0;
            std::memcpy(o.f.header, f.header, FX_HDR_DEC);
            o.f.rssi_db = 20.0f * log10f(f.gamma); o.f.cfo = f.dphi;
            o.pjob = -1;
            if (!detect && o.f.header_valid) {
                const PlanDev &pd = get_plan(c, f.pay_len, f.check, f.fec0, f.fec1);
                FxPayJob j{};
                j.x = xs[s]; j.start = f.start; j.mix_th = f.mix_th; j.mix_dl = f.mix_dl; j.mf_scale = f.mf_scale;
                j.pfb = f.pfb; j.mfc0 = f.mfc0; j.pll_th = f.pll_th; j.pll_f = f.pll_f; j.ms = f.ms; j.bps = fx::modem_bps(f.ms);
                j.nsym = f.pay_sym_len; j.sym_off = (uint32_t)sym_total;
                j.pay_len = f.pay_len; j.check = f.check; j.fec0 = f.fec0; j.fec1 = f.fec1;
                j.k = pd.plan.k; j.l0 = pd.plan.l0; j.l1 = pd.plan.l1; j.perm0_off = pd.perm0_off; j.perm1_off = pd.perm1_off;
                j.byte_off = (uint32_t)byte_total; j.dw_off = (uint32_t)dw_total; j.out_off = (uint32_t)out_total;
                o.pjob = (int)sl.pjobs.size();
                for (uint32_t c0 = 0; c0 < j.nsym; c0 += 1024) { sl.blk_job.push_back((uint32_t)o.pjob); sl.blk_c0.push_back(c0); }
                sym_total += (j.nsym + 7) & ~7u;                      // 8-symbol granules: 64-byte block I/O in the PLL kernel
                byte_total += (uint64_t)((std::max(j.l1, j.k) + 8 + 15) & ~15u);
                dw_total += ((8ull * std::max(j.l0, j.k) + 6 + 63) & ~63ull) + 64;   // whole 64-step chunks, lane-major
                out_total += (j.pay_len + 15) & ~15u;
                sl.pjobs.push_back(j);
                o.f.mod_scheme = f.ms; o.f.mod_bps = j.bps; o.f.check = f.check; o.f.fec0 = f.fec0; o.f.fec1 = f.fec1;
                o.f.payload_len = f.pay_len; o.f.num_framesyms = f.pay_sym_len;
            }
            sl.out.push_back(o);
        }
    }
    sl.timing.frames = sl.out.size(); sl.timing.payload_symbols = sym_total; sl.n_syms = sym_total;
    if (sym_total >= (1ull << 32) || byte_total >= (1ull << 32) || dw_total >= (1ull << 32)) { set_err("fxrx_submit: batch too large for 32-bit arena offsets"); return FXRX_ERR_ARG; }

    const size_t NP = sl.pjobs.size();
    if (NP) {
        if (c->perm_host.size() != perm_before || c->perm_uploaded != c->perm_host.size()) {
            // a packet configuration seen for the first time: the shared gather-table arena grows.  Rare; drain
            // everything so that no in-flight decode still reads the old allocation.
            sync_all(c);
            if (c->d_perm.reserve(c->perm_host.size())) return FXRX_ERR_HIP;
            HIP_OK(hipMemcpy(c->d_perm.p, c->perm_host.data(), c->perm_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            c->perm_uploaded = c->perm_host.size();
        }
        const size_t NB = sl.blk_job.size();
        auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
        const size_t o_blk = up16(NP * sizeof(FxPayJob)), o_c0 = o_blk + up16(NB * 4), o_pll = o_c0 + up16(NB * 4),
                     o_dec = o_pll + up16(NP * 4), meta_bytes = o_dec + up16(NP * 4);
        if (sl.d_meta.reserve(meta_bytes) || sl.hp_meta.reserve(meta_bytes) ||
            sl.d_symraw.reserve(sym_total + 8) || sl.d_framesyms.reserve(sym_total + 8) || sl.d_hard.reserve(sym_total + 64) ||
            sl.d_bufA.reserve(byte_total) || sl.d_bufB.reserve(byte_total) || sl.d_dw.reserve(dw_total) ||
            sl.d_out.reserve(out_total + 16) || sl.d_pres.reserve(NP) || sl.h_pres.reserve(NP) || sl.h_out.reserve(out_total + 16)) return FXRX_ERR_HIP;
        const FxPayJob *d_pjobs = reinterpret_cast<const FxPayJob *>(sl.d_meta.p);
        const uint32_t *d_blk_job = reinterpret_cast<const uint32_t *>(sl.d_meta.p + o_blk), *d_blk_c0 = reinterpret_cast<const uint32_t *>(sl.d_meta.p + o_c0),
                       *d_pll_idx = reinterpret_cast<const uint32_t *>(sl.d_meta.p + o_pll), *d_dec_idx = reinterpret_cast<const uint32_t *>(sl.d_meta.p + o_dec);
        // group PLL jobs by modulation scheme (one grid per scheme: demodulator resolved at compile time)
        std::map<unsigned, std::vector<uint32_t>> by_ms;
        for (size_t i = 0; i < NP; i++) by_ms[sl.pjobs[i].ms].push_back((uint32_t)i);
        sl.pll_idx.clear();
        std::vector<std::tuple<unsigned, size_t, size_t>> groups;
        for (auto &kv : by_ms) { groups.emplace_back(kv.first, sl.pll_idx.size(), kv.second.size()); sl.pll_idx.insert(sl.pll_idx.end(), kv.second.begin(), kv.second.end()); }
        std::memcpy(sl.hp_meta.p, sl.pjobs.data(), NP * sizeof(FxPayJob));
        std::memcpy(sl.hp_meta.p + o_blk, sl.blk_job.data(), NB * sizeof(uint32_t));
        std::memcpy(sl.hp_meta.p + o_c0, sl.blk_c0.data(), NB * sizeof(uint32_t));
        std::memcpy(sl.hp_meta.p + o_pll, sl.pll_idx.data(), NP * sizeof(uint32_t));
        // decode grids: frames without / with a Reed-Solomon stage (the latter use a heavier kernel instance)
        size_t n_plain = 0, n_rs = 0;
        {
            uint32_t *di = reinterpret_cast<uint32_t *>(sl.hp_meta.p + o_dec);
            for (size_t i = 0; i < NP; i++) if (sl.pjobs[i].fec0 != FX_FEC_RS_M8 && sl.pjobs[i].fec1 != FX_FEC_RS_M8) di[n_plain++] = (uint32_t)i;
            for (size_t i = 0; i < NP; i++) if (sl.pjobs[i].fec0 == FX_FEC_RS_M8 || sl.pjobs[i].fec1 == FX_FEC_RS_M8) di[n_plain + n_rs++] = (uint32_t)i;
        }
        // The whole payload stage runs on this block's payload stream, so that stream W is free for the next block's
        // walk as soon as this one's is stitched.  The MF is the only payload kernel that reads the IQ; its input was
        // staged on W before the walk, which the host has already waited for.
        HIP_OK(hipMemcpyAsync(sl.d_meta.p, sl.hp_meta.p, meta_bytes, hipMemcpyHostToDevice, sl.stream_p));
        HIP_OK(hipEventRecord(sl.ev_mf0, sl.stream_p));
        hipLaunchKernelGGL(fx_paymf_kernel, dim3((unsigned)NB), dim3(256), 0, sl.stream_p,
                           d_pjobs, d_blk_job, d_blk_c0, sl.d_symraw.p, c->d_tables);
        HIP_OK(hipGetLastError());
        HIP_OK(hipEventRecord(sl.ev_mf1, sl.stream_p));
        for (unsigned s = 0; s < NS; s++) if (sl.wbuf[s] >= 0) c->st[s].work_mf[sl.wbuf[s]] = sl.ev_mf1;   // guards the work buffers
        // P: payload PLL
        HIP_OK(hipEventRecord(sl.ev_pll0, sl.stream_p));
        // stagger concurrent blocks' PLL grids over different CUs (see the kernel): slot k skips k * (grid rounded to 32)
        const unsigned pll_wgs = (unsigned)((NP + 64 * c->pll_waves - 1) / (64 * c->pll_waves));
        const unsigned wg_skip = (c->pll_stagger && pll_wgs <= 128) ? (sl.index * ((pll_wgs + c->pll_stagger - 1u) / c->pll_stagger * c->pll_stagger)) % (unsigned)c->n_cus : 0u;
        for (auto &g : groups)
            HIP_OK(fx_launch_paypll(std::get<0>(g), (unsigned)std::get<2>(g), wg_skip, c->pll_waves, sl.stream_p, d_pjobs, d_pll_idx + std::get<1>(g),
                                    sl.d_symraw.p, sl.d_framesyms.p, sl.d_hard.p, sl.d_pres.p, c->d_tables));
        HIP_OK(hipEventRecord(sl.ev_pll1, sl.stream_p));
        // D: packet decode, results home
        HIP_OK(hipEventRecord(sl.ev_dec0, sl.stream_d));
        if (n_plain) HIP_OK(fx_launch_paydec(0, (unsigned)n_plain, c->dec_waves, sl.stream_d, d_pjobs, d_dec_idx, sl.d_hard.p, c->d_perm.p,
                                            sl.d_bufA.p, sl.d_bufB.p, sl.d_dw.p, sl.d_out.p, sl.d_pres.p, c->d_tables));
        if (n_rs) HIP_OK(fx_launch_paydec(1, (unsigned)n_rs, 1u, sl.stream_d, d_pjobs, d_dec_idx + n_plain, sl.d_hard.p, c->d_perm.p,
                                         sl.d_bufA.p, sl.d_bufB.p, sl.d_dw.p, sl.d_out.p, sl.d_pres.p, c->d_tables));
        HIP_OK(hipEventRecord(sl.ev_dec1, sl.stream_d));
        HIP_OK(hipMemcpyAsync(sl.h_pres.p, sl.d_pres.p, NP * sizeof(FxPayResult), hipMemcpyDeviceToHost, sl.stream_d));
        HIP_OK(hipMemcpyAsync(sl.h_out.p, sl.d_out.p, out_total, hipMemcpyDeviceToHost, sl.stream_d));
        if (c->cfg.want_framesyms) {
            if (sl.h_framesyms.reserve(sym_total)) return FXRX_ERR_HIP;
            HIP_OK(hipMemcpyAsync(sl.h_framesyms.p, sl.d_framesyms.p, sym_total * sizeof(float2), hipMemcpyDeviceToHost, sl.stream_d));
        }
    }
    HIP_OK(hipEventRecord(sl.ev_done, NP ? sl.stream_d : sl.stream_w));

    // ---- 5. carry the unconsumed tail of every stream into the next call (walk stream) ----
    // (not if the streams were reset after this block was submitted: its resume state is nobody's business then)
    for (unsigned s = 0; s < NS && sl.epoch == c->epoch; s++) {
        StreamState &S = c->st[s]; const Chain &ch = chains[s];
        int64_t keep_from = ch.fresh ? ch.pos : ch.pos - FX_HOP;
        keep_from = std::max<int64_t>(0, std::min<int64_t>(keep_from, ns[s]));
        const size_t keep = (size_t)(ns[s] - keep_from);
        const int nxt = S.cur ^ 1;
        if (keep) {
            if (S.carry[nxt].reserve(keep)) return FXRX_ERR_HIP;
            HIP_OK(hipMemcpyAsync(S.carry[nxt].p, xs[s] + keep_from, keep * sizeof(float2), hipMemcpyDeviceToDevice, sl.stream_w));
        }
        S.cur = nxt; S.carry_len = keep; S.max_keep = std::max(S.max_keep, keep); S.carry_ev = sl.ev_carry;
        if (sl.wbuf[s] >= 0) S.work_rd[sl.wbuf[s]] = sl.ev_carry;
        S.pos = ch.pos - keep_from; S.floor_ = ch.floor_ - keep_from; S.fresh = ch.fresh;
    }
    HIP_OK(hipEventRecord(sl.ev_carry, sl.stream_w));
    sl.stage = Slot::LAUNCHED;
    sl.timing.host_submit_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    return 0;
}

// A block goes through walk_phase, stitch_phase, finish_phase.  The last two need the host, and the true walker of
// block k+1 needs the stream state block k leaves behind.  The host is software-pipelined so that it does not wait for
// kernels it has just launched: a submit launches the new block's walkers, finishes the block whose verification was
// launched by the previous submit, and stitches the block whose walkers were launched by the previous submit
// (launching its verification).  That works when the new block is independent of the pending ones (the streams were
// reset in between: separate captures, the bench's passes) and, through cross-block speculation (WALK_SPEC), for big
// blocks of a continuing stream; small continuing blocks run the pending blocks to the end first.
static int advance_front(fxrx_ctx_s *c)          // one phase of the oldest pending block
{
    Slot *p = c->pending.front();
    if (p->stage == Slot::WALKING) return stitch_phase(c, *p);
    if (finish_phase(c, *p)) return FXRX_ERR_HIP;
    c->pending.pop_front();
    return 0;
}

int fxrx_submit(fxrx_ctx *c, const void *const *iq, const uint64_t *n_samples, int on_device)
{
    if (!c || !iq || !n_samples) { set_err("fxrx_submit: null argument"); return FXRX_ERR_ARG; }
    if (c->inflight >= c->depth) { set_err("fxrx_submit: pipeline full, call fxrx_collect first"); return FXRX_ERR_STATE; }
    HIP_OK(hipSetDevice(c->cfg.device));
    Slot &sl = *c->slots[c->head];
    const unsigned NS = c->cfg.n_streams;
    // independent of everything pending: the streams were reset after the newest pending block was submitted
    bool indep = c->early_walk && !c->pending.empty() && on_device && c->pending.back()->epoch != c->epoch;
    for (const auto &S : c->st) if (S.carry_len != 0 || !S.fresh) indep = false;
    // continuing the newest pending block, and big enough to be worth walking speculatively
    bool cont = c->early_walk && !indep && !c->pending.empty() && c->pending.back()->epoch == c->epoch && NS <= 8;
    for (unsigned s = 0; s < NS; s++) if (n_samples[s] < (1u << 18)) cont = false;
    auto finish_front = [&]() -> int { return (!c->pending.empty() && c->pending.front()->stage == Slot::VERIFYING) ? advance_front(c) : 0; };
    auto stitch_next = [&]() -> int {
        for (Slot *p : c->pending) if (p->stage == Slot::WALKING) return stitch_phase(c, *p);
        return 0;
    };
    if (indep) {                     // software pipeline: the new walk first, then one phase each of the two blocks behind it
        if (walk_phase(c, sl, iq, n_samples, on_device, WALK_SERIAL)) return FXRX_ERR_HIP;
        if (finish_front() || stitch_next()) return FXRX_ERR_HIP;
    } else if (cont) {               // same, but the block being verified is finished first: its payload MF is the last
        if (finish_front()) return FXRX_ERR_HIP;                                   // reader of the work buffer staged next
        for (Slot *p : c->pending) if (p->stage == Slot::WALKING) { if (launch_true_walkers(c, *p)) return FXRX_ERR_HIP; break; }
        if (walk_phase(c, sl, iq, n_samples, on_device, WALK_SPEC)) return FXRX_ERR_HIP;
        if (stitch_next()) return FXRX_ERR_HIP;
    } else {                         // state needed and not worth speculating: run the pending blocks to the end first
        while (!c->pending.empty()) if (advance_front(c)) return FXRX_ERR_HIP;
        if (walk_phase(c, sl, iq, n_samples, on_device, WALK_SERIAL)) return FXRX_ERR_HIP;
    }
    while (c->pending.size() >= 2) if (advance_front(c)) return FXRX_ERR_HIP;     // (never more than one block behind the one being verified)
    c->pending.push_back(&sl);
    sl.busy = true;
    c->head = (c->head + 1) % c->depth; c->inflight++;
    return 0;
}

int fxrx_collect(fxrx_ctx *c)
{
    if (!c) return FXRX_ERR_ARG;
    if (!c->inflight) { set_err("fxrx_collect: nothing in flight"); return FXRX_ERR_STATE; }
    HIP_OK(hipSetDevice(c->cfg.device));
    Slot &sl = *c->slots[c->tail];
    while (sl.stage != Slot::LAUNCHED) if (advance_front(c)) return FXRX_ERR_HIP;     // blocks are pending in order: the oldest is this one
    {
        const auto tw = std::chrono::steady_clock::now();
        HIP_OK(hipEventSynchronize(sl.ev_done));
        HIP_OK(hipEventSynchronize(sl.ev_carry));
        sl.timing.host_collectwait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
    }      // tail-carry copies of this block (cheap; keeps caller buffers reusable)
    for (auto &o : sl.out) {
        if (o.pjob < 0) continue;
        const FxPayJob &j = sl.pjobs[(size_t)o.pjob]; const FxPayResult &r = sl.h_pres.p[o.pjob];
        o.f.payload = sl.h_out.p + j.out_off; o.f.payload_valid = (int)r.payload_valid;
        o.f.evm_sum = r.evm_sum; o.f.evm_db = 10.0f * log10f(r.evm_sum / (float)(j.nsym ? j.nsym : 1));
        o.f.framesyms = c->cfg.want_framesyms ? (const fx_complex *)(sl.h_framesyms.p + j.sym_off) : nullptr;
    }
    if (!sl.pjobs.empty()) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, sl.ev_mf0, sl.ev_mf1); sl.timing.paymf_ms = ms;
        (void)hipEventElapsedTime(&ms, sl.ev_pll0, sl.ev_pll1); sl.timing.paypll_ms = ms;
        (void)hipEventElapsedTime(&ms, sl.ev_dec0, sl.ev_dec1); sl.timing.paydec_ms = ms;
    }
    sl.timing.total_ms = sl.timing.walk_ms + sl.timing.seekverify_ms + sl.timing.paymf_ms + sl.timing.paypll_ms + sl.timing.paydec_ms;
    sl.busy = false; sl.stage = Slot::IDLE; c->last = &sl;
    c->tail = (c->tail + 1) % c->depth; c->inflight--;
    return (int)sl.out.size();
}

int fxrx_process(fxrx_ctx *c, const void *const *iq, const uint64_t *n_samples, int on_device)
{
    if (!c) return FXRX_ERR_ARG;
    while (c->inflight) { int r = fxrx_collect(c); if (r < 0) return r; }      // drain anything a caller left in flight
    int r = fxrx_submit(c, iq, n_samples, on_device);
    if (r < 0) return r;
    return fxrx_collect(c);
}

int fxrx_debug_walk_stamps(const fxrx_ctx *c, uint64_t out[4]) { if (!c) return FXRX_ERR_ARG; std::memcpy(out, c->walk_stamp, sizeof c->walk_stamp); return 0; }
int fxrx_debug_walk_maxjob(const fxrx_ctx *c, uint64_t out[8]) { if (!c) return FXRX_ERR_ARG; for (int i = 0; i < 4; i++) out[i] = c->walk_stamp_maxjob[i]; out[4] = c->walk_maxjob_hops; out[5] = c->walk_maxjob_cheap; out[6] = c->walk_maxjob_frames; out[7] = c->walk_stamp_max; return 0; }

// diagnostic: decode-phase shader-clock deltas of payload job i (zeros unless built with -DFX_STAMPS)
int fxrx_debug_stamps(const fxrx_ctx *c, unsigned int i, uint32_t out[8])
{
    if (!c || !c->last || i >= c->last->pjobs.size()) return FXRX_ERR_ARG;
    std::memcpy(out, c->last->h_pres.p[i].stamp, 8 * sizeof(uint32_t)); return 0;
}

int fxrx_result(const fxrx_ctx *c, unsigned int i, fxrx_frame *out)
{
    if (!c || !out || !c->last || i >= c->last->out.size()) return FXRX_ERR_ARG;
    *out = c->last->out[i].f; return 0;
}

unsigned int fxrx_gen_frame_len(unsigned int ms, unsigned int check, unsigned int fec0, unsigned int fec1, unsigned int n)
{
    fx::FrameGen g; g.ms = ms; g.check = check; g.fec0 = fec0; g.fec1 = fec1;
    return g.frame_len(n);
}

// ---- block-API index maps (reference: lib/flex_tx_impl.cc:75-181, lib/flex_rx_impl.cc:74-179) ----
static const int kMod[11] = { FX_MODEM_PSK2, FX_MODEM_PSK4, FX_MODEM_PSK8, FX_MODEM_PSK16, FX_MODEM_DPSK2, FX_MODEM_DPSK4,
                              FX_MODEM_DPSK8, FX_MODEM_ASK4, FX_MODEM_QAM16, FX_MODEM_QAM32, FX_MODEM_QAM64 };
static const int kInner[7] = { FX_FEC_NONE, FX_FEC_CONV_V27, FX_FEC_CONV_V27P23, FX_FEC_CONV_V27P45, FX_FEC_CONV_V27P56,
                               FX_FEC_CONV_V27P67, FX_FEC_CONV_V27P78 };
static const int kOuter[8] = { FX_FEC_NONE, FX_FEC_GOLAY2412, FX_FEC_RS_M8, FX_FEC_HAMMING74, FX_FEC_HAMMING128,
                               FX_FEC_SECDED2216, FX_FEC_SECDED3932, FX_FEC_SECDED7264 };
int fxrx_mod_from_index(int i) { return (i >= 0 && i < 11) ? kMod[i] : -1; }
int fxrx_inner_from_index(int i) { return (i >= 0 && i < 7) ? kInner[i] : -1; }
int fxrx_outer_from_index(int i) { return (i >= 0 && i < 8) ? kOuter[i] : -1; }
int fxrx_mod_to_index(unsigned v) { for (int i = 0; i < 11; i++) if ((unsigned)kMod[i] == v) return i; return -1; }
int fxrx_inner_to_index(unsigned v) { for (int i = 0; i < 7; i++) if ((unsigned)kInner[i] == v) return i; return -1; }
int fxrx_outer_to_index(unsigned v) { for (int i = 0; i < 8; i++) if ((unsigned)kOuter[i] == v) return i; return -1; }

}  // extern "C"
