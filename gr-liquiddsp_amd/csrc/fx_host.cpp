// fx_host.cpp -- host runtime of libfxrx.so: device tables, block descriptors, the stream-ordered kernel chain of a
// block, result marshalling and the C ABI declared in include/fxrx.h.
//
// A block (one fxrx_submit: new samples of every stream) is ONE chain of kernels enqueued on the block's own HIP stream;
// the host never waits between them:
//
//   walkers (speculative segments)      -- fx_walk_kernel, launched at once
//   seek verification, phase 0          -- fx_seekverify_kernel over the runs those walkers emitted (continuing streams only)
//   [wait: chain kernel of the previous block]   -- an event, on the device; from here to the chain kernel a block of
//                                          continuing streams is on the context's high-priority stream
//   true walkers of continuing streams  -- fx_walk_kernel; their start state is read from device memory
//   seek verification (phase 1)         -- the remaining runs
//   chain                               -- fx_chainfast_kernel: stitch / resume state / carried tail, per stream
//   plan                                -- fx_plan_kernel + fx_planlists_kernel: payload jobs, work lists, host result records
//   payload MF -> PLL -> packet decode  -- results land in pinned host memory
//
// Slow paths, all at fxrx_collect (repair_and_replay, finish_decode): repair rounds (segments walked again in parallel:
// hand-off misses, fired skipped hops), the full-size fx_chain_kernel, a whole-block walk without hop skipping, carry
// buffers that have to grow, decode launches that the hint-sized grids did not cover.
//
// Why segments: liquid's synchroniser is one sequential state machine per stream (where the detector restarts after a
// frame depends on that frame's header).  Each stream is cut into segments that are walked concurrently from a
// freshly-reset detector; a segment's walker, once past its end, keeps seeking until its next detection (a, cfo_bin) --
// the hand-off target.  If the next segment's speculative list contains that same (a, cfo_bin), everything after it is
// provably what the sequential machine would have produced, so the lists are spliced; otherwise that segment is walked
// again from the true state.  The result is identical to a single sequential walk, for any segment size.
//
// What carries a stream from block to block lives on the device: FxStreamState (resume hop, zero-floor, freshness) in a
// ring indexed by block number, and the unconsumed tail, right-aligned in one of three carry buffers per stream (block b
// reads buffer b mod 3 and writes (b+1) mod 3).  The host learns both at fxrx_collect.  The one thing the device cannot
// do is grow a carry buffer: a tail longer than the buffer marks the state invalid, every block behind it exits at once,
// and fxrx_collect of the overflowing block enlarges the buffers, copies the tail itself and enqueues the blocks behind
// it again ("replay"; their inputs are still there: input buffers stay valid until their block is collected).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
#include "../../include/fxrx.h"
#include "fx_device.h"
#include "fx_codec.hpp"

extern "C" hipError_t fx_launch_walk(unsigned mode, int eq, unsigned njobs, hipStream_t st, const FxWalkJob *jobs, const uint32_t *job_list, FxWalkResult *results,
                                     FxFrame *frames, FxVerifyRun *runs, uint32_t run_cap, FxBlockHdr *hdr, const FxTables *T, int ext, uint32_t n_jobs_total,
                                     const uint32_t *n_list, uint32_t list_cap);
extern "C" hipError_t fx_launch_seekverify(unsigned grid, hipStream_t st, const FxVerifyRun *runs, uint32_t run_cap, const FxWalkJob *jobs, FxWalkResult *results,
                                           FxFrame *frames, FxBlockHdr *hdr, const FxTables *T, uint32_t phase);
extern "C" hipError_t fx_launch_chain(unsigned mode, int eq, unsigned nstreams, hipStream_t st, const FxStreamDesc *streams, const FxWalkJob *jobs, uint32_t n_jobs_total,
                                      FxWalkResult *results, FxFrame *frames, FxFrame *chain, uint32_t *chain_count, FxVerifyRun *runs, uint32_t run_cap,
                                      FxBlockHdr *hdr, uint32_t force_slow, const FxTables *T);
extern "C" hipError_t fx_launch_chainfast(unsigned nstreams, hipStream_t st, const FxStreamDesc *streams, const FxWalkJob *jobs, const FxWalkResult *results,
                                          const FxFrame *frames, FxFrame *chain, uint32_t *chain_count, FxBlockHdr *hdr, uint32_t force_repair,
                                          FxWalkJob *jobs_rw, uint32_t *req_list, uint32_t *stat, uint32_t pass);
extern "C" hipError_t fx_launch_plan(hipStream_t st, unsigned grid, const FxStreamDesc *streams, uint32_t nstreams, uint32_t detect, uint32_t eq, uint32_t vb_blk, const FxFrame *chain,
                                     const uint32_t *chain_count, uint32_t *stream_base, FxPayJob *pjobs, FxOutRec *recs, uint32_t *mf_job, uint32_t *mf_c0, uint32_t mf_cap,
                                     uint32_t *pll_list, uint32_t *dec_list, uint32_t list_cap, uint32_t *vb_items, uint32_t vb_cap, FxBlockHdr *hdr, FxBlockHdr *hdr_pay,
                                     FxBlockHdr *hdr_host, uint32_t *plan_ws);
extern "C" unsigned fx_plan_ws_words(void);
extern "C" hipError_t fx_launch_vbpre(unsigned first_wave, unsigned n_waves, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx, const FxBlockHdr *hdr,
                                      const uint8_t *hard, uint8_t *bufA, uint8_t *bufB, const FxTables *T);
extern "C" hipError_t fx_launch_vbitems(unsigned first_item, unsigned n_items, hipStream_t st, const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap,
                                        const FxBlockHdr *hdr, uint8_t *bufA, const uint8_t *bufB, unsigned long long *dwv, uint8_t *vec_arena, uint32_t *vb_st, uint32_t dbg, int packed,
                                        int with_fix);
extern "C" hipError_t fx_launch_vbfinish(unsigned first_wave, unsigned n_waves, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx, FxBlockHdr *hdr,
                                         uint8_t *bufA, uint8_t *bufB, unsigned long long *dwv, const uint8_t *vec_arena, const uint32_t *vb_st, uint32_t *fb_list, uint32_t list_cap,
                                         uint8_t *out, FxOutRec *recs, FxBlockHdr *hdr_host);
extern "C" hipError_t fx_launch_paymf(unsigned grid, int eq, hipStream_t st, const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr,
                                      const FxFrame *chain, float2 *sym_raw, const FxTables *T);
extern "C" hipError_t fx_launch_paypll(unsigned grid_waves, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs, const uint32_t *pll_list, const FxBlockHdr *hdr,
                                       const float2 *sym_raw, float2 *framesyms, uint8_t *hard, FxOutRec *recs, const FxTables *T);
extern "C" hipError_t fx_launch_paydec(int with_rs, int soft, unsigned first_wave, unsigned grid_waves, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs,
                                       const uint32_t *job_idx, const FxBlockHdr *hdr, const uint8_t *hard, uint8_t *bufA, uint8_t *bufB, uint8_t *soft_arena,
                                       unsigned long long *dw_arena, uint8_t *out, FxOutRec *recs, FxPayResult *res, const FxTables *T, FxBlockHdr *fallback_host);
extern "C" hipError_t fx_launch_symcopy(unsigned grid, hipStream_t st, const FxBlockHdr *hdr, const float2 *sym, float2 *host);
extern "C" hipError_t fx_launch_upload(hipStream_t st, const void *src, void *dst, size_t bytes, unsigned n_cus);
extern "C" hipError_t fx_launch_copy_u32(hipStream_t st, const uint32_t *src, uint32_t *dst);
extern "C" hipError_t fx_launch_softdemod(unsigned grid, hipStream_t st, const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr,
                                          const float2 *framesyms, const uint8_t *hard, uint8_t *soft_arena, const FxTables *T);

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

// growable device / pinned-host buffers (contents are NOT preserved across a growth)
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

constexpr size_t kUploadKernelMax = 64u << 20;      // host blocks up to this size are uploaded by fx_upload_kernel when their memory is page-locked
// is p page-locked host memory the device can read, and at which address?
static bool pinned_device_ptr(const void *p, const void **dev)
{
    if (!p || (reinterpret_cast<uintptr_t>(p) & 7u)) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }     // (pageable memory: not an error here)
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return false;
    *dev = a.devicePointer;
    return true;
}
constexpr unsigned kMaxDepth = 32;
constexpr unsigned kStateRing = kMaxDepth + 3;      // FxStreamState records per stream: one per block in flight and then some
constexpr uint32_t kFallbackWaves = 32;             // in-chain launch for frames the batch Viterbi path hands back (normally none)
constexpr uint32_t kRepairCap = 256;                // frame-table slots per stream for walks done by the chain kernel

struct StreamState {
    float2 *carry[3] = { nullptr, nullptr, nullptr };   // tails, right-aligned: block b reads carry[b % 3], writes carry[(b + 1) % 3]
    int64_t carry_cap = 0;                  // samples per buffer
    int64_t total = 0;                      // absolute index of the next new sample (since the last reset)
    bool fresh_start = true;                // the next block starts from a freshly reset synchroniser (no state to read)
    int64_t carry_bound = 0;                // upper bound of the tail the next block will find (sizes its arenas)
    unsigned noskip_left = 0;               // blocks this stream is still walked with the exact detector on every hop (a skipped hop fired recently; survives a reset)
};

// what a block needs to know about a stream at submit time (kept for a replay)
struct StreamSnap { int64_t tot0 = 0; bool fresh_start = true; int64_t carry_bound = 0; bool noskip = false; };

}  // namespace

struct Out { fxrx_frame f; };

// One block in flight: descriptors, device tables, arenas, result buffers and its own HIP stream.
struct Slot {
    hipStream_t st = nullptr;
    hipEvent_t ev[10] = {};                  // 0 start | 1 walk | 2 verify | 3 chain | 4 plan | 5 mf | 6 pll | 7 dec | 8 done
    unsigned index = 0;
    bool busy = false;
    uint64_t seq = 0;                        // block number since context creation
    // the input, as submitted (kept for a replay)
    std::vector<const float2 *> x; std::vector<uint64_t> n; std::vector<StreamSnap> snap;
    std::vector<DevBuf<float2>> d_in;        // staging of host inputs
    // descriptor arena, one upload: [FxWalkJob x NJ | job lists | FxStreamDesc x NS]
    PinBuf<uint8_t> hp_desc; DevBuf<uint8_t> d_desc;
    size_t NJ = 0, n_early = 0, n_late = 0, o_list = 0, o_streams = 0;
    // device tables
    DevBuf<FxWalkResult> d_wres; DevBuf<FxFrame> d_frames, d_chain; DevBuf<FxVerifyRun> d_runs;
    DevBuf<FxBlockHdr> d_hdr;                // [0] walk-phase counters (zero between blocks), [1] what the payload kernels read
    DevBuf<uint32_t> d_chain_count, d_stream_base, d_mf_job, d_mf_c0, d_pll_list, d_dec_list, d_vb_items;
    DevBuf<uint32_t> d_plan_ws;              // plan kernels: look-back state, list counts and cursors (zero between blocks)
    DevBuf<uint32_t> d_req;                  // repair rounds: segments to be walked again from their true start state
    DevBuf<uint32_t> d_cstat;                // in-chain repair round: per stream 1 stitched | 2 pending (segments queued) | 3 left to fxrx_collect
    DevBuf<uint8_t> d_vb_vec;                // batch Viterbi: metric differences at the start and end of every trellis block
    DevBuf<unsigned long long> d_vb_dw;      // its decision words, step-major within the 64 work items of a wave
    DevBuf<uint32_t> d_vb_st;                // traceback states and flags per work item
    uint32_t vb_cap = 0, vb_blk = 0, vb_pre_launched = 0, vb_items_launched = 0, fb_launched = 0;
    uint64_t vb_dw_words = 0, vb_slots_alloc = 128;
    DevBuf<FxPayJob> d_pjobs; DevBuf<FxPayResult> d_pres;
    uint32_t run_cap = 0, chain_cap = 0, mf_cap = 0, frame_slots = 0;
    uint64_t sym_cap = 0, byte_cap = 0, dw_cap = 0, out_cap = 0;
    // payload arenas
    DevBuf<float2> d_symraw;                 // matched-filter output; the PLL overwrites it in place with the carrier-recovered symbols
    DevBuf<uint8_t> d_hard, d_bufA, d_bufB, d_soft; DevBuf<unsigned long long> d_dw;
    // results (pinned host memory the kernels write into)
    PinBuf<FxBlockHdr> h_hdr; PinBuf<FxOutRec> h_recs; PinBuf<uint8_t> h_out, h_soft; PinBuf<float2> h_framesyms;
    std::vector<Out> out;
    uint64_t n_syms = 0;
    fxrx_timing timing{};
    double host_submit_ms = 0.0;
    bool any_late = false;
    uint32_t dec_launched = 0, rs_launched = 0;   // decode waves launched with the chain (lean / Reed-Solomon instance)
    double dbg_submit_ms = 0.0, dbg_collect_ms = 0.0, dbg_gpu_start_ms = 0.0, dbg_gpu_done_ms = 0.0;   // fxrx_debug_block_times
    bool force_noskip = false;               // walk this block with the exact detector on every hop (nothing to verify)
    int timing_level = 0;                    // which stage events this block recorded (fxrx_set_timing)
    bool inchain = false;                    // the repair round within the chain was enqueued with it
    uint32_t kept_hops = 0, kept_cheap = 0, kept_vhops = 0, kept_vfail = 0, kept_repairs = 0;   // walk-phase counters of a block whose back part was run again
};

struct fxrx_ctx_s {
    fxrx_config cfg{};
    FxTables *d_tables = nullptr;
    std::vector<StreamState> st;
    FxStreamState *d_state = nullptr; FxStreamState *h_state = nullptr;   // [kStateRing][n_streams], device / pinned mirror
    float taper = -1.0f;                 // FXRX_TAPER: segment lengths of a stream fall from (1 + t) to (1 - t) times the mean (-1: chosen per block)
    bool skip_seek = true;               // FXRX_SKIP_SEEK=0: walkers run the full detector on every hop (nothing to verify)
    bool chain_slow = false;             // FXRX_CHAIN_SLOW=1: every block goes through the full-size chain kernel's general (sequential) path
    unsigned pll_waves = 1, dec_waves = 1;   // waves per workgroup of the PLL / decode grids (placement only)
    int n_cus = 256;
    uint64_t seq = 0;                    // blocks submitted so far
    hipEvent_t prev_chain = nullptr;     // chain kernel of the newest submitted block (borrowed from its slot)
    hipEvent_t carry_reader[3] = { nullptr, nullptr, nullptr };   // payload MF of the newest block that reads carry[i]
    uint32_t verify_per = 4;             // hops per verification run (adapted to the traffic)
    uint64_t frames_hint = 0, rs_hint = 0;   // frames / Reed-Solomon frames of the last collected block (size the PLL / decode grids)
    uint64_t plain_hint = 0, batch_hint = 0, vb_items_hint = 0, vb_steps_hint = 0, vb_want_hint = 0;   // likewise: frames of the wave-per-frame / batch decoders, trellis blocks, trellis steps
    bool first_block = true;             // nothing collected yet: grids cover their lists' capacity
    bool batch_viterbi = true;           // FXRX_BATCH_VITERBI=0: every frame through the wave-per-frame decoder
    // The repair round within the chain costs two launches per block -- and launches are what the pipeline is short of (DESIGN.md
    // section 6) --, so it is only enqueued while blocks keep needing it: for the next 16 blocks after one that did.
    // FXRX_INCHAIN_REPAIR=0: never, 1: while needed (default), 2: always.
    int inchain_repair = 1; unsigned inchain_left = 0;
    int timing_level = -1;               // stage events per block: -1 auto (all stages one block at a time, none with blocks in flight) | 0 none | 1 the PLL only | 2 all stages
    hipEvent_t ref_event = nullptr; double ref_host_ms = 0.0;   // fxrx_debug_block_times: a common origin of GPU and host clocks
    int debug_walk_twice = 0;            // FXRX_DEBUG_WALK_TWICE: the speculative walkers are launched twice, the stage time is the second launch's (cold-start experiment)
    int debug_stop_after = 0;            // FXRX_DEBUG_STOP_AFTER (tools/dev/dev_stage_cost.py): 4 plan | 5 matched filter | 6 PLL -- the chain ends there, no results
    uint64_t vbfix_hint = 0;             // the last collected block's batch Viterbi path ran blocks again / handed frames back: the hand-over check is launched with the next one
    uint64_t fb_hint = 0;                // frames the last collected block's batch Viterbi path handed back (sizes the fallback launch; 0: none enqueued)
    uint32_t walk_per_cu = 2;            // walker workgroups resident per CU (FXRX_WALK_PER_CU; follows the kernel's register budget)
    hipStream_t st_chain = nullptr;      // highest priority: the state-dependent stretch of continuing blocks (true walkers, their verification, chain kernel)
    uint32_t verify_per_cu = 4;          // FXRX_VERIFY_PER_CU: workgroups of the seek verifier per CU (they stride over the runs)
    uint32_t mf_per_cu = 0;              // FXRX_MF_PER_CU: workgroups of the payload matched filter per CU (0: one per item of the last block)
    uint64_t mf_items_hint = 0;
    uint32_t plan_grid = 0;              // FXRX_PLAN_GRID: workgroups of the plan kernels (tests; default: from the last block's frame count)
    uint32_t vb_debug = 0, vb_blk_force = 0;   // tests: FXRX_VB_DEBUG (see fx_vbfix_kernel / fx_vbtrace_kernel), FXRX_VB_BLK (trellis steps per block)
    // pipeline: a ring of depth + 1 slots, so that the block whose results are exposed is never the one being refilled
    std::vector<std::unique_ptr<Slot>> slots; unsigned depth = 1, head = 0, tail = 0, inflight = 0;
    Slot *last = nullptr;                // slot whose results are currently exposed through fxrx_result
    uint64_t replays = 0, repairs_host = 0, late_decodes = 0;
    unsigned noskip_left = 0;            // blocks still to be walked with the exact detector on every hop (after a verification failure)
    unsigned debug_fail_submit = 0, debug_fail_collect = 0;   // tests (fxrx_debug_fail): the next n submits / collects report a failure
    uint64_t discarded = 0;              // blocks dropped by a failing fxrx_collect (see discard_inflight)
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
    for (int i = 0; i < FX_PN_LEN; i++) t->pn[i] = make_float2(H.pn[i].re, H.pn[i].im);
    fx::design_eq_init(t->eq0);
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

// frame-table slots per walk job: the densest legal traffic is a header-only frame (618 samples) after the other.  (Should a
// table fill up all the same -- FX_EXIT_TABLE_FULL -- the full-size chain kernel continues the segment.)
// Densest possible traffic: a flexframe is at least ~630 samples long (preamble, header, tail), so flex_rx finds at most one frame
// per 600 samples; the bare detector (frame_detector_cc) can fire again on the very next hop, 256 samples on.
inline uint32_t seg_frames_cap(uint64_t seg, bool detect) { return (uint32_t)std::min<uint64_t>(seg / (detect ? 256 : 600) + 8, 4000); }

int alloc_carry(StreamState &S, int64_t cap)
{
    for (auto &p : S.carry) {
        if (p) (void)hipFree(p);
        p = nullptr;
        if (hipMalloc((void **)&p, (size_t)cap * sizeof(float2)) != hipSuccess) { set_err("hipMalloc (carry buffer) failed"); return FXRX_ERR_HIP; }
    }
    S.carry_cap = cap;
    return 0;
}

}  // namespace

// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and streams sharing a queue
// serialise; the pipeline wants one queue per block in flight.  The runtime reads the variable when it initialises, so
// this only helps when the library is loaded before the first HIP call.  An existing value is never overridden.
__attribute__((constructor)) static void fxrx_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

// ============================================================================ batched API
extern "C" {

const char *fxrx_last_error(void) { return g_err.c_str(); }
void fxrx_set_error(const char *msg) { set_err(msg ? msg : ""); }      // for the library's other translation units
const char *fxrx_version(void) { return "fxrx 0.2 (gfx950)"; }
int fxrx_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
// page-locked host memory: uploads from it are asynchronous (fxrx_submit returns while the copy is still under way)
void *fxrx_pinned_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_err("fxrx_pinned_alloc: hipHostMalloc failed"); return nullptr; }
    return p;
}
void fxrx_pinned_free(void *p) { if (p) (void)hipHostFree(p); }

static int make_slot(fxrx_ctx_s *c)
{
    std::unique_ptr<Slot> s(new Slot);
    HIP_OK(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
    for (auto &e : s->ev) HIP_OK(hipEventCreate(&e));
    s->index = (unsigned)c->slots.size();
    if (s->h_hdr.reserve(1) || s->d_hdr.reserve(2)) return FXRX_ERR_HIP;
    std::memset(s->h_hdr.p, 0, sizeof(FxBlockHdr));
    HIP_OK(hipMemset(s->d_hdr.p, 0, 2 * sizeof(FxBlockHdr)));     // (fx_plan_kernel zeroes the counters again after every block)
    c->slots.push_back(std::move(s));
    return 0;
}

static void sync_all(fxrx_ctx_s *c) { if (c->st_chain) (void)hipStreamSynchronize(c->st_chain); for (auto &s : c->slots) if (s->st) (void)hipStreamSynchronize(s->st); }

// FXRX_DEBUG_SUBMIT_PROFILE: where the host time of fxrx_submit goes (printed when a context is destroyed)
static double g_prof[8]; static unsigned long g_prof_n;
static const bool g_prof_on = std::getenv("FXRX_DEBUG_SUBMIT_PROFILE") != nullptr;
static inline double prof_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void fxrx_destroy(fxrx_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    sync_all(c);
    if (g_prof_on && g_prof_n) {
        std::fprintf(stderr, "[fxrx] submit profile over %lu blocks (ms per block): upload + bookkeeping %.4f, segments %.4f, memory + descriptors %.4f, front launches %.4f, back launches %.4f\n",
                     g_prof_n, g_prof[0] / g_prof_n, g_prof[1] / g_prof_n, g_prof[2] / g_prof_n, g_prof[3] / g_prof_n, g_prof[4] / g_prof_n);
        for (auto &v : g_prof) v = 0.0; g_prof_n = 0;
    }
    for (auto &s : c->slots) {
        for (auto e : s->ev) if (e) (void)hipEventDestroy(e);
        if (s->st) (void)hipStreamDestroy(s->st);
    }
    if (c->st_chain) { (void)hipStreamSynchronize(c->st_chain); (void)hipStreamDestroy(c->st_chain); }
    for (auto &S : c->st) for (auto &p : S.carry) if (p) (void)hipFree(p);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_state) (void)hipFree(c->d_state);
    if (c->h_state) (void)hipHostFree(c->h_state);
    delete c;
}

fxrx_ctx *fxrx_create(const fxrx_config *cfg)
{
    if (!cfg || cfg->n_streams == 0) { set_err("fxrx_create: bad config"); return nullptr; }
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || nd <= 0 || cfg->device >= nd) {
        set_err("fxrx_create: no usable HIP device (this library has no CPU path)"); return nullptr;
    }
    if (hipSetDevice(cfg->device) != hipSuccess) { set_err("hipSetDevice failed"); return nullptr; }
    fxrx_ctx_s *c = new fxrx_ctx_s;
    auto fail = [&]() -> fxrx_ctx * { const std::string keep = g_err; fxrx_destroy(c); set_err(keep); return nullptr; };   // frees whatever was created
    c->cfg = *cfg;
    if (c->cfg.threshold <= 0.0f) c->cfg.threshold = cfg->mode == FXRX_MODE_DETECTOR ? 0.45f : 0.5f;
    if (const char *e = std::getenv("FXRX_PLL_WAVES")) c->pll_waves = (unsigned)std::min(4, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_DEC_WAVES")) c->dec_waves = (unsigned)std::min(8, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_SKIP_SEEK")) c->skip_seek = std::atoi(e) != 0;
    if (const char *e = std::getenv("FXRX_TAPER")) c->taper = std::min(0.95f, (float)std::atof(e));
    if (const char *e = std::getenv("FXRX_CHAIN_SLOW")) c->chain_slow = std::atoi(e) != 0;
    if (const char *e = std::getenv("FXRX_BATCH_VITERBI")) c->batch_viterbi = std::atoi(e) != 0;
    if (const char *e = std::getenv("FXRX_INCHAIN_REPAIR")) c->inchain_repair = std::min(2, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_DEBUG_STOP_AFTER")) c->debug_stop_after = std::atoi(e);
    if (const char *e = std::getenv("FXRX_DEBUG_WALK_TWICE")) c->debug_walk_twice = std::atoi(e);
    if (const char *e = std::getenv("FXRX_TIMING")) c->timing_level = std::min(2, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_WALK_PER_CU")) c->walk_per_cu = (uint32_t)std::min(8, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_VERIFY_PER_CU")) c->verify_per_cu = (uint32_t)std::min(64, std::max(1, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_MF_PER_CU")) c->mf_per_cu = (uint32_t)std::min(64, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_PLAN_GRID")) c->plan_grid = (uint32_t)std::min(256, std::max(0, std::atoi(e)));
    if (const char *e = std::getenv("FXRX_VB_DEBUG")) c->vb_debug = (uint32_t)std::atoi(e);
    // (a block is at least as long as the warm-up of the next one: 128 steps)
    if (const char *e = std::getenv("FXRX_VB_BLK")) if (std::atoi(e) > 0) c->vb_blk_force = (uint32_t)std::min(4096, std::max(128, (std::atoi(e) + 63) / 64 * 64));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    if (upload_tables(c) != 0) return fail();
    const unsigned NS = cfg->n_streams;
    c->st.resize(NS);
    // carry buffers: the unconsumed tail is an incomplete frame (plus a detector window); 2 MB covers everything up to
    // 260 k samples per stream, longer tails grow the buffers through the replay path
    int64_t cap0 = std::max<int64_t>(1 << 15, std::min<int64_t>(1 << 18, (int64_t)(1 << 24) / (int64_t)NS));
    if (const char *e = std::getenv("FXRX_CARRY_SAMPLES")) cap0 = std::max<int64_t>(1024, std::atoll(e));
    for (auto &S : c->st) if (alloc_carry(S, cap0) != 0) return fail();
    if (hipMalloc((void **)&c->d_state, sizeof(FxStreamState) * kStateRing * NS) != hipSuccess ||
        hipHostMalloc((void **)&c->h_state, sizeof(FxStreamState) * kStateRing * NS, hipHostMallocDefault) != hipSuccess) { set_err("fxrx_create: state ring allocation failed"); return fail(); }
    if (hipMemset(c->d_state, 0, sizeof(FxStreamState) * kStateRing * NS) != hipSuccess) { set_err("hipMemset failed"); return fail(); }
    std::memset(c->h_state, 0, sizeof(FxStreamState) * kStateRing * NS);
    for (unsigned i = 0; i < 2; i++) if (make_slot(c) != 0) return fail();
    return c;
}

// forget all per-stream state (position, carried tail).  Blocks in flight are unaffected: what they leave behind on the
// device is simply never read.
void fxrx_reset(fxrx_ctx *c)
{
    if (!c) return;
    for (auto &S : c->st) { S.total = 0; S.fresh_start = true; S.carry_bound = 0; }
}

int fxrx_set_depth(fxrx_ctx *c, unsigned int depth)
{
    if (!c || depth == 0 || depth > kMaxDepth) return FXRX_ERR_ARG;
    if (c->inflight) { set_err("fxrx_set_depth: blocks in flight"); return FXRX_ERR_STATE; }
    HIP_OK(hipSetDevice(c->cfg.device));
    while (c->slots.size() < depth + 1) if (make_slot(c) != 0) return FXRX_ERR_HIP;
    c->depth = depth; c->head = c->tail = 0; c->last = nullptr;
    return 0;
}

// diagnostics (timing level 2 set before the first submit): of the last collected block, in ms since a common origin --
// host time when fxrx_submit was entered, GPU time of its first stage event, GPU time of its last event, host time when
// fxrx_collect had it
int fxrx_debug_block_times(const fxrx_ctx *c, double out[4])
{
    if (!c || !c->last || !out) return FXRX_ERR_ARG;
    out[0] = c->last->dbg_submit_ms; out[1] = c->last->dbg_gpu_start_ms; out[2] = c->last->dbg_gpu_done_ms; out[3] = c->last->dbg_collect_ms;
    return 0;
}

int fxrx_set_timing(fxrx_ctx *c, int level)
{
    if (!c || level < -1 || level > 2) return FXRX_ERR_ARG;
    c->timing_level = level;
    return 0;
}

void *fxrx_stream(const fxrx_ctx *c) { return (c && !c->slots.empty()) ? (void *)c->slots[0]->st : nullptr; }

int fxrx_last_timing(const fxrx_ctx *c, fxrx_timing *t) { if (!c || !t || !c->last) return FXRX_ERR_ARG; *t = c->last->timing; return 0; }

const void *fxrx_device_framesyms(const fxrx_ctx *c, uint64_t *n)
{
    if (!c || !c->last) return nullptr;
    if (n) *n = c->last->n_syms;
    return c->last->n_syms ? c->last->d_symraw.p : nullptr;
}

enum { kChainFast = 0, kChainFull = 1, kChainDone = 2 };
static int enqueue_back(fxrx_ctx_s *c, Slot &sl, int chain_mode, hipStream_t chain_st = nullptr);

// ---- enqueue the whole kernel chain of the block in `sl` (descriptors are rebuilt: a replay calls this again) ----
static int enqueue_block(fxrx_ctx_s *c, Slot &sl)
{
    const unsigned NS = c->cfg.n_streams;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    const uint64_t b = sl.seq;
    hipStream_t st = sl.st;
    const double tp0 = g_prof_on ? prof_now() : 0.0;

    // ---- 1. segments ----
    uint64_t seg = c->cfg.segment_len;
    if (seg == 0) {
        uint64_t tot = 0;
        for (unsigned s = 0; s < NS; s++) tot += sl.n[s];
        // Walker workgroups resident at once: two per CU for the flex_rx instance (4 waves x 256 VGPRs each), two for the
        // leaner detector-only instance.  One block at a time (latency): little work -> a single round of workgroups with a
        // small margin (the kernel then lasts as long as its slowest segment); lots of work -> ~4 rounds so that uneven
        // segments even out.  Several blocks in flight (throughput): every segment start costs a speculative walker a
        // pre-lock scan of half a frame on average -- about as much as two frames' worth of real work -- and the chip is
        // kept busy by the other blocks anyway, so fewer, longer segments: down to one workgroup per two CUs.
        const uint64_t slots = (uint64_t)c->n_cus * c->walk_per_cu;
        // (about three blocks' walkers overlap at any time: keep that many workgroups' worth of segments between them)
        if (c->depth > 1 && !detect) seg = tot / std::max<uint64_t>(std::max<uint64_t>(1, c->n_cus / 2), std::min<uint64_t>(slots, 3 * slots / c->depth));
        else if (tot / slots < 131072) seg = tot / (slots - slots / 16);
        else seg = std::max<uint64_t>(tot / (4 * slots), 65536);
        seg = std::max<uint64_t>(seg, 32768); seg = std::min<uint64_t>(seg, 1u << 20);
    }
    seg = std::max<uint64_t>(seg, 4096);
    // (detector-only mode with more jobs than fit on the chip at once: config 3 alone 28.7 -> 24.7 ms; the flex_rx walker, whose launches
    // overlap the rest of the chain, gains nothing from it: configs 4, 5 and 2 measured equal or worse)
    float taper = c->taper;
    if (taper < 0.0f) {
        uint64_t tot = 0;
        for (unsigned s = 0; s < NS; s++) tot += sl.n[s];
        taper = (detect && tot / seg > 4ull * (uint64_t)c->n_cus) ? 0.75f : 0.0f;
    }
    std::vector<FxWalkJob> jobs; std::vector<uint32_t> early, late; std::vector<FxStreamDesc> sds(NS);
    uint32_t frame_slots = 0, chain_slots = 0; uint64_t n_total = 0, carry_total = 0;
    for (unsigned s = 0; s < NS; s++) {
        StreamState &S = c->st[s]; const StreamSnap &sn = sl.snap[s];
        const int64_t ns = (int64_t)sl.n[s];
        const bool cont = !sn.fresh_start;
        FxStreamDesc &sd = sds[s];
        sd.x = sl.x[s]; sd.xa_end = cont ? S.carry[b % 3] + S.carry_cap : nullptr; sd.n = ns;
        sd.first_job = (uint32_t)jobs.size();
        sd.state_in = cont ? c->d_state + ((b + kStateRing - 1) % kStateRing) * NS + s : nullptr;
        sd.state_out = c->d_state + (b % kStateRing) * NS + s;
        sd.state_out_host = c->h_state + (b % kStateRing) * NS + s;
        sd.carry_out_end = S.carry[(b + 1) % 3] + S.carry_cap; sd.carry_cap = S.carry_cap;
        sd.abs_base = sn.tot0;
        // a continuing stream's true walker only has to reach its first hand-off, so its own segment is short: it starts
        // late (after the previous block's chain kernel) while the speculative walkers of the other segments are long under way
        int64_t p = 0; bool first = true;
        const int64_t seg0 = cont ? (int64_t)std::min<uint64_t>(seg, 8192) : (int64_t)seg;
        // Tapered segments: what is launched last is short.  A walker workgroup alone on its CU runs at well under half the rate four
        // of them reach together (tools/dev/dev_walk_timeline.py), so the last jobs of a launch set the length of its drain.
        const int64_t p_t = cont ? std::min<int64_t>(ns, seg0) : 0;                       // the taper covers [p_t, ns)
        const int64_t k_t = taper > 0.0f ? std::max<int64_t>(1, ((ns - p_t) + (int64_t)seg / 2) / (int64_t)seg) : 0;
        int64_t i_t = 0;
        while (first || p < ns) {
            FxWalkJob j{};
            j.x = sd.x; j.xa_end = sd.xa_end; j.n = ns; j.start = p;
            j.stop = std::min<int64_t>(ns, p + (first ? seg0 : (int64_t)seg));
            if (ns - j.stop < (int64_t)seg / 2) j.stop = ns;          // fold a short last segment in
            if (k_t > 1 && p >= p_t) {
                // piece i of k: boundary at the integral of the falling line (1 + t) -> (1 - t)
                const double u = (double)(i_t + 1) / (double)k_t;
                const int64_t e = p_t + (int64_t)((double)(ns - p_t) * (u * (1.0 + taper) - taper * u * u));
                j.stop = (i_t + 1 >= k_t) ? ns : std::min<int64_t>(ns, std::max<int64_t>(p + 4096, e & ~(int64_t)255));
                i_t++;
            }
            j.fresh = 1u; j.floor = p;
            j.mode = detect ? FX_MODE_DETECT : FX_MODE_FLEXRX;
            j.handoff = j.stop < ns ? 1u : 0u;
            j.prelock = first ? 0u : 1u;
            j.frame_base = frame_slots; j.max_frames = seg_frames_cap((uint64_t)(j.stop - j.start) + (first && cont ? (uint64_t)sn.carry_bound : 0u), detect);
            frame_slots += j.max_frames;
            j.threshold = c->cfg.threshold;
            j.no_skip = (detect || !c->skip_seek || sl.force_noskip || sn.noskip) ? 1u : 0u;
            j.state_in = (first && cont) ? sd.state_in : nullptr;
            j.stream = s; j.verify_per = c->verify_per; j.eq = (!detect && c->cfg.equalizer) ? 1u : 0u;
            (first && cont ? late : early).push_back((uint32_t)jobs.size());
            jobs.push_back(j);
            p = j.stop; first = false;
            if (p >= ns) break;
        }
        sd.n_jobs = (uint32_t)jobs.size() - sd.first_job;
        sd.chain_base = chain_slots; sd.chain_cap = (uint32_t)((uint64_t)(ns + (cont ? sn.carry_bound : 0)) / (detect ? 256u : 600u) + 8u);
        chain_slots += sd.chain_cap;
        n_total += (uint64_t)ns; carry_total += cont ? (uint64_t)sn.carry_bound : 0u;
    }
    for (unsigned s = 0; s < NS; s++) { sds[s].repair_base = frame_slots; sds[s].repair_cap = kRepairCap; frame_slots += kRepairCap; }
    if (taper > 0.0f)
        std::stable_sort(early.begin(), early.end(), [&](uint32_t a, uint32_t b2) { return jobs[a].stop - jobs[a].start > jobs[b2].stop - jobs[b2].start; });
    const size_t NJ = jobs.size();
    sl.NJ = NJ; sl.n_early = early.size(); sl.n_late = late.size(); sl.any_late = !late.empty();
    sl.frame_slots = frame_slots; sl.chain_cap = chain_slots;
    sl.run_cap = (uint32_t)std::min<uint64_t>((n_total + carry_total) / FX_HOP + 2 * NJ + 1024, 0x7fffffffu);
    const uint64_t span = n_total + carry_total;                     // samples the block's frames can occupy
    sl.sym_cap = span / 2 + 8ull * chain_slots + 64;
    sl.byte_cap = (span / 2) * 3 / 4 + 48ull * chain_slots + 64;
    sl.dw_cap = detect ? 0 : (span / 2) * 6 + 200ull * chain_slots + 64;
    sl.out_cap = (span / 2) * 3 / 4 + 16ull * chain_slots + 64;
    sl.mf_cap = (uint32_t)std::min<uint64_t>(span / 2 / 1024 + chain_slots + 16, 0x7fffffffu);
    // batch Viterbi: trellis steps per block from the traffic of the last block.  The forward pass is arithmetic bound, three
    // waves to a SIMD: some six waves per SIMD spread evenly over the chip, fewer and the slowest SIMD sets the time; blocks
    // shorter than a couple of warm-ups waste their work.  (Longer blocks with several blocks in flight -- less warm-up, the
    // chip is full anyway -- measured worse: 192 steps 30.0, 256: 29.8, 384: 28.8, 512: 27.6, 1024: 24.6 Gsamples/s on config 2.)
    {
        const uint64_t want_items = 64ull * 4ull * (uint64_t)c->n_cus * 6ull;
        sl.vb_blk = (detect || c->cfg.soft_decision || !c->batch_viterbi) ? 0u
                  : (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(192, ((c->vb_steps_hint / want_items + 63) / 64) * 64));
    }
    if (sl.vb_blk && c->vb_blk_force) sl.vb_blk = c->vb_blk_force;
    // (arena: 1.5 trellis steps per sample of span -- QPSK r1/2 has 0.47, QAM16 r2/3 1.17 -- or what the last block's traffic
    // asked for plus a quarter, in work items of the block length chosen above; frames whose work items do not fit are
    // decoded by the wave-per-frame kernel, and the next block's arena is larger.  Sized in steps, not items, so that a
    // change of block length does not reallocate gigabytes.)
    {
        const uint64_t steps_cap = std::min<uint64_t>(std::max<uint64_t>(span + span / 2, c->vb_steps_hint + c->vb_steps_hint / 4), 4 * span);
        sl.vb_cap = sl.vb_blk ? (uint32_t)((steps_cap / sl.vb_blk + 2048 + 127) & ~127ull) : 128u;
        // (the decision-word arena itself: the same number of words whatever the block length -- the 2048 extra work items
        // priced at the longest block --, or a change of block length by one notch would reallocate it, 1.5 x larger)
        sl.vb_dw_words = sl.vb_blk ? std::max<uint64_t>((uint64_t)sl.vb_cap * sl.vb_blk, steps_cap + 2176ull * 4096ull) : 0;
        sl.vb_slots_alloc = sl.vb_blk ? std::max<uint64_t>(sl.vb_cap, steps_cap / 192 + 2176) : 128;     // likewise the per-item arenas: priced at the shortest block
    }
    if (sl.sym_cap >= (1ull << 32) || sl.dw_cap >= (1ull << 32)) { set_err("fxrx_submit: batch too large for 32-bit arena offsets"); return FXRX_ERR_ARG; }

    const double tp1 = g_prof_on ? prof_now() : 0.0;
    // ---- 2. memory ----
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_list = up16(NJ * sizeof(FxWalkJob)), o_streams = o_list + up16(NJ * sizeof(uint32_t)), desc_bytes = o_streams + up16(NS * sizeof(FxStreamDesc));
    sl.o_list = o_list; sl.o_streams = o_streams;
    const uint32_t list_cap = chain_slots + 64 * FX_PLL_CLASSES;
    if (sl.hp_desc.reserve(desc_bytes) || sl.d_desc.reserve(desc_bytes) || sl.d_wres.reserve(NJ + NS) || sl.d_frames.reserve(frame_slots) ||
        sl.d_runs.reserve(sl.run_cap) || sl.d_req.reserve(2 * NJ + 16) || sl.d_cstat.reserve(NS) || sl.d_chain.reserve(chain_slots) || sl.d_chain_count.reserve(NS) || sl.d_stream_base.reserve(NS + 1) ||
        sl.d_pjobs.reserve(chain_slots) || sl.h_recs.reserve(chain_slots) || sl.d_mf_job.reserve(sl.mf_cap) || sl.d_mf_c0.reserve(sl.mf_cap) ||
        sl.d_pll_list.reserve(list_cap) || sl.d_dec_list.reserve(4 * (size_t)list_cap)) return FXRX_ERR_HIP;
    if (!detect && (sl.d_symraw.reserve(sl.sym_cap) || sl.d_hard.reserve(sl.sym_cap + 64) ||
                    sl.d_bufA.reserve(sl.byte_cap + 256) || sl.d_bufB.reserve(sl.byte_cap + 256) || sl.d_dw.reserve(sl.dw_cap) || sl.h_out.reserve(sl.out_cap))) return FXRX_ERR_HIP;
    if (!detect && c->cfg.want_framesyms && sl.h_framesyms.reserve(sl.sym_cap)) return FXRX_ERR_HIP;
    if (!sl.d_plan_ws.p) {
        if (sl.d_plan_ws.reserve(fx_plan_ws_words())) return FXRX_ERR_HIP;
        HIP_OK(hipMemsetAsync(sl.d_plan_ws.p, 0, fx_plan_ws_words() * sizeof(uint32_t), st));
    }
    if (sl.d_vb_items.reserve(2 * (size_t)sl.vb_slots_alloc) ||
        (sl.vb_blk && (sl.d_vb_vec.reserve(128 * (size_t)sl.vb_slots_alloc) || sl.d_vb_dw.reserve((size_t)sl.vb_dw_words) || sl.d_vb_st.reserve(sl.vb_slots_alloc)))) return FXRX_ERR_HIP;
    if (!detect && c->cfg.soft_decision && (sl.d_soft.reserve(8 * sl.byte_cap) || (c->cfg.want_framesyms && sl.h_soft.reserve(8 * sl.byte_cap)))) return FXRX_ERR_HIP;
#ifdef FX_STAMPS
    if (!detect && sl.d_pres.reserve(chain_slots)) return FXRX_ERR_HIP;
#endif
    std::memcpy(sl.hp_desc.p, jobs.data(), NJ * sizeof(FxWalkJob));
    uint32_t *hl = reinterpret_cast<uint32_t *>(sl.hp_desc.p + o_list);
    if (!early.empty()) std::memcpy(hl, early.data(), early.size() * sizeof(uint32_t));
    if (!late.empty()) std::memcpy(hl + early.size(), late.data(), late.size() * sizeof(uint32_t));
    std::memcpy(sl.hp_desc.p + o_streams, sds.data(), NS * sizeof(FxStreamDesc));
    const FxWalkJob *d_jobs = reinterpret_cast<const FxWalkJob *>(sl.d_desc.p);
    const uint32_t *d_list = reinterpret_cast<const uint32_t *>(sl.d_desc.p + o_list);
    const unsigned mode = detect ? FX_MODE_DETECT : FX_MODE_FLEXRX;

    const double tp2 = g_prof_on ? prof_now() : 0.0;
    // ---- 3. the chain, front part: walkers and seek verification ----
    HIP_OK(fx_launch_upload(st, sl.hp_desc.p, sl.d_desc.p, desc_bytes, (unsigned)c->n_cus));     // (descriptors: our own page-locked buffer)
    const int tl = c->timing_level >= 0 ? c->timing_level : (c->depth > 1 ? 0 : 2);      // stage events: see fxrx_set_timing
    sl.timing_level = tl;
    for (int rep = 0; rep < c->debug_walk_twice; rep++)
        HIP_OK(fx_launch_walk(mode, c->cfg.equalizer ? 1 : 0, (unsigned)early.size(), st, d_jobs, d_list, sl.d_wres.p, sl.d_frames.p, sl.d_runs.p, sl.run_cap, sl.d_hdr.p, c->d_tables, 0, (uint32_t)NJ, nullptr, 0u));
    if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[0], st));
    HIP_OK(fx_launch_walk(mode, c->cfg.equalizer ? 1 : 0, (unsigned)early.size(), st, d_jobs, d_list, sl.d_wres.p, sl.d_frames.p, sl.d_runs.p, sl.run_cap, sl.d_hdr.p, c->d_tables, 0, (uint32_t)NJ, nullptr, 0u));
    // the true walkers of continuing streams read the state the previous block's chain kernel leaves.  What the speculative
    // walkers skipped is verified before that wait -- it does not depend on the state --, so that the chain of dependencies
    // from one block's chain kernel to the next one's is just: true walkers, their few verification runs, chain kernel
    const bool verify = !detect && c->skip_seek && !sl.force_noskip;
    hipStream_t cst = st;                                     // the stream the rest of the front part and the chain kernel go to
    if (!late.empty()) {
        if (verify) {
            HIP_OK(fx_launch_copy_u32(st, &sl.d_hdr.p->n_runs, &sl.d_hdr.p->runs_done));
            HIP_OK(fx_launch_seekverify(c->verify_per_cu * (unsigned)c->n_cus, st, sl.d_runs.p, sl.run_cap, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_hdr.p, c->d_tables, 0u));
        }
        // (from here to the chain kernel the block is on the chain of dependencies that runs through all blocks of a continuing
        // stream: a single workgroup or two that must not queue behind the chip-filling kernels of the other blocks in flight
        // -- so this stretch goes to the context's high-priority stream, which also keeps the chain kernels in block order)
        if (!c->st_chain) {
            int lo = 0, hi = 0;
            HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIP_OK(hipStreamCreateWithPriority(&c->st_chain, hipStreamNonBlocking, hi));
        }
        HIP_OK(hipEventRecord(sl.ev[9], st));
        cst = c->st_chain;
        HIP_OK(hipStreamWaitEvent(cst, sl.ev[9], 0));
        if (c->prev_chain) HIP_OK(hipStreamWaitEvent(cst, c->prev_chain, 0));
        HIP_OK(fx_launch_walk(mode, c->cfg.equalizer ? 1 : 0, (unsigned)late.size(), cst, d_jobs, d_list + early.size(), sl.d_wres.p, sl.d_frames.p, sl.d_runs.p, sl.run_cap, sl.d_hdr.p, c->d_tables, 0, (uint32_t)NJ, nullptr, 0u));
    }
    if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[1], cst));
    if (verify)
        HIP_OK(fx_launch_seekverify(late.empty() ? c->verify_per_cu * (unsigned)c->n_cus : (unsigned)std::min<size_t>((size_t)c->verify_per_cu * (size_t)c->n_cus, std::max<size_t>(64, 16 * late.size())), cst, sl.d_runs.p,
                                    sl.run_cap, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_hdr.p, c->d_tables, 1u));
    if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[2], cst));
    // chain kernels run in block order in any case (they write the carry buffers in rotation); this one writes
    // carry[(b + 1) % 3], which the payload MF of block b - 2 may still be reading
    if (late.empty() && c->prev_chain) HIP_OK(hipStreamWaitEvent(cst, c->prev_chain, 0));
    if (c->carry_reader[(b + 1) % 3]) HIP_OK(hipStreamWaitEvent(cst, c->carry_reader[(b + 1) % 3], 0));
    const double tp3 = g_prof_on ? prof_now() : 0.0;
    const int rb = enqueue_back(c, sl, kChainFast, cst);
    if (g_prof_on) { const double tp4 = prof_now(); g_prof[1] += tp1 - tp0; g_prof[2] += tp2 - tp1; g_prof[3] += tp3 - tp2; g_prof[4] += tp4 - tp3; g_prof_n++; }
    return rb;
}

// ---- back part of the chain: chain kernel, plan, payload stage.  `full`: the full-size chain kernel, which can walk
// (fxrx_collect runs it for a block whose lean chain kernel reported FX_BLK_NEEDS_REPAIR) ----
static int enqueue_back(fxrx_ctx_s *c, Slot &sl, int chain_mode, hipStream_t chain_st)
{
    const unsigned NS = c->cfg.n_streams;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    const unsigned mode = detect ? FX_MODE_DETECT : FX_MODE_FLEXRX;
    const uint64_t b = sl.seq;
    hipStream_t st = sl.st;
    const FxWalkJob *d_jobs = reinterpret_cast<const FxWalkJob *>(sl.d_desc.p);
    const FxStreamDesc *d_streams = reinterpret_cast<const FxStreamDesc *>(sl.d_desc.p + sl.o_streams);
    const uint32_t chain_slots = sl.chain_cap, list_cap = chain_slots + 64 * FX_PLL_CLASSES;
    FxBlockHdr *hdr = sl.d_hdr.p, *hdr_pay = sl.d_hdr.p + 1;
    if (chain_mode == kChainFull)
        HIP_OK(fx_launch_chain(mode, c->cfg.equalizer ? 1 : 0, NS, st, d_streams, d_jobs, (uint32_t)sl.NJ, sl.d_wres.p, sl.d_frames.p, sl.d_chain.p, sl.d_chain_count.p, sl.d_runs.p, sl.run_cap,
                               hdr, c->chain_slow ? 1u : 0u, c->d_tables));
    else if (chain_mode == kChainFast) {
        // stitch; one repair round within the chain for what only takes a walk from a hand-off state (a hand-off target missing
        // from the next list, a skipped hop on which the exact detector fires): its walkers read the number of queued segments
        // on the device -- normally none: two launches that leave at once --; stitch the streams concerned again.  Whatever
        // is still open after that is flagged and mended when the block is collected (repair_and_replay).
        hipStream_t cs = chain_st ? chain_st : st;
        FxWalkJob *d_jobs_rw = reinterpret_cast<FxWalkJob *>(sl.d_desc.p);
        const uint32_t req_cap = (uint32_t)(2 * sl.NJ + 16);
        sl.inchain = !c->chain_slow && (c->inchain_repair == 2 || (c->inchain_repair == 1 && c->inchain_left > 0));
        if (c->inchain_left) c->inchain_left--;
        if (!sl.inchain)
            HIP_OK(fx_launch_chainfast(NS, cs, d_streams, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_chain.p, sl.d_chain_count.p, hdr, c->chain_slow ? 1u : 0u, nullptr, nullptr, nullptr, 0u));
        else {
            HIP_OK(fx_launch_chainfast(NS, cs, d_streams, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_chain.p, sl.d_chain_count.p, hdr, 0u, d_jobs_rw, sl.d_req.p, sl.d_cstat.p, 1u));
            HIP_OK(fx_launch_walk(mode, c->cfg.equalizer ? 1 : 0, (unsigned)std::min<size_t>(2 * sl.NJ, 64), cs, d_jobs, sl.d_req.p, sl.d_wres.p, sl.d_frames.p, sl.d_runs.p, sl.run_cap, hdr,
                                  c->d_tables, 1, (uint32_t)sl.NJ, &hdr->n_repair_req, req_cap));
            HIP_OK(fx_launch_chainfast(NS, cs, d_streams, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_chain.p, sl.d_chain_count.p, hdr, 0u, nullptr, nullptr, sl.d_cstat.p, 2u));
        }
    }
    // (kChainDone: the repair rounds have stitched the block already)
    HIP_OK(hipEventRecord(sl.ev[3], chain_mode == kChainFast && chain_st ? chain_st : st));
    if (chain_mode == kChainFast && chain_st && chain_st != st) HIP_OK(hipStreamWaitEvent(st, sl.ev[3], 0));   // back onto the block's own stream
    c->prev_chain = sl.ev[3];
    // (the plan kernels' workgroups take contiguous ranges of the chain's frames: about 2048 each, from the last block's count)
    HIP_OK(fx_launch_plan(st, c->plan_grid ? c->plan_grid : (unsigned)std::min<uint64_t>(64, c->frames_hint / 2048 + 1), d_streams, NS, detect ? 1u : 0u, c->cfg.equalizer ? 1u : 0u, sl.vb_blk, sl.d_chain.p,
                          sl.d_chain_count.p, sl.d_stream_base.p, sl.d_pjobs.p, sl.h_recs.p, sl.d_mf_job.p, sl.d_mf_c0.p, sl.mf_cap, sl.d_pll_list.p, sl.d_dec_list.p, list_cap,
                          sl.d_vb_items.p, sl.vb_cap, hdr, hdr_pay, sl.h_hdr.p, sl.d_plan_ws.p));
    const int tl = sl.timing_level;
    if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[4], st));
    if (c->debug_stop_after == 4) { c->carry_reader[b % 3] = nullptr; HIP_OK(hipEventRecord(sl.ev[8], st)); return 0; }
    if (!detect) {
        // grids stride over lists whose lengths only the device knows; size them from what the last block held
        const uint64_t fh = c->frames_hint ? std::min<uint64_t>(chain_slots, c->frames_hint + c->frames_hint / 2 + 64) : chain_slots;
        // (one workgroup per item of 1024 symbols when the previous block's count is anything to go by -- measured better than
        // fewer workgroups striding over the items: 8 per CU 34.3, 18: 35.1, one per item (36 per CU on config 2): 35.3 Gsamples/s)
        const unsigned mf_grid = (unsigned)std::min<uint64_t>(sl.mf_cap, c->mf_per_cu ? (uint64_t)c->mf_per_cu * (uint64_t)c->n_cus
                                                                                     : std::max<uint64_t>(8ull * (uint64_t)c->n_cus, c->mf_items_hint + c->mf_items_hint / 8 + 64));
        HIP_OK(fx_launch_paymf(mf_grid, c->cfg.equalizer ? 1 : 0, st, sl.d_pjobs.p, sl.d_mf_job.p, sl.d_mf_c0.p, hdr_pay, sl.d_chain.p, sl.d_symraw.p, c->d_tables));
        HIP_OK(hipEventRecord(sl.ev[5], st));
        c->carry_reader[b % 3] = sl.ev[5];
        if (c->debug_stop_after == 5) { HIP_OK(hipEventRecord(sl.ev[8], st)); return 0; }
        HIP_OK(fx_launch_paypll((unsigned)(fh / 64 + FX_PLL_CLASSES), c->pll_waves, st, sl.d_pjobs.p, sl.d_pll_list.p, hdr_pay, sl.d_symraw.p, sl.d_symraw.p,
                                sl.d_hard.p, sl.h_recs.p, c->d_tables));
        if (tl >= 1) HIP_OK(hipEventRecord(sl.ev[6], st));
        if (c->debug_stop_after == 6) { HIP_OK(hipEventRecord(sl.ev[8], st)); return 0; }
        FxPayResult *pres = nullptr;
#ifdef FX_STAMPS
        pres = sl.d_pres.p;
#endif
        // decode: one wave per frame.  The lean instance has no loop (it would double its registers): its grid covers what
        // the last block held plus a margin, and a second, usually empty launch covers the rest of the list's capacity in
        // big workgroups (few of them).  The Reed-Solomon instance strides.
        const int soft = c->cfg.soft_decision ? 1 : 0;
        if (soft) {
            // per-bit soft values from the carrier-recovered symbols (data parallel, off the PLL's recurrence); the decoder
            // de-interleaves them in place, so a caller that wants to see them gets a copy first
            HIP_OK(fx_launch_softdemod(mf_grid, st, sl.d_pjobs.p, sl.d_mf_job.p, sl.d_mf_c0.p, hdr_pay, sl.d_symraw.p, sl.d_hard.p, sl.d_soft.p, c->d_tables));
            if (c->cfg.want_framesyms) HIP_OK(hipMemcpyAsync(sl.h_soft.p, sl.d_soft.p, 8 * sl.byte_cap, hipMemcpyDeviceToHost, st));
        }
        // decode: one wave per frame.  The lean instance has no loop (it would double its registers) and an empty workgroup
        // still has to be given its registers before it can leave: so the grid covers what the last block held plus a margin,
        // not the list's capacity, and the Reed-Solomon instance (256 registers a wave) is only launched while such frames
        // keep turning up.  What a launch did not cover is decoded when the block is collected (finish_decode).
        // (the plain instance too is only launched while such frames keep turning up: with the batch path on, a block normally has none)
        sl.dec_launched = c->first_block ? chain_slots : (c->plain_hint || !sl.vb_blk ? (unsigned)std::min<uint64_t>(chain_slots, c->plain_hint + c->plain_hint / 2 + 64) : 0u);
        sl.rs_launched = c->rs_hint ? (unsigned)std::min<uint64_t>(chain_slots, c->rs_hint + c->rs_hint / 2 + 64) : 0u;
        HIP_OK(fx_launch_paydec(0, soft, 0, sl.dec_launched, c->dec_waves, st, sl.d_pjobs.p, sl.d_dec_list.p, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p, sl.d_soft.p,
                                sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, nullptr));
        HIP_OK(fx_launch_paydec(1, soft, 0, sl.rs_launched, 1u, st, sl.d_pjobs.p, sl.d_dec_list.p + list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p,
                                sl.d_bufB.p, sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, nullptr));
        // batch Viterbi path (most frames: hard decisions, convolutional fec0): front part, forward pass over trellis blocks, back part
        if (sl.vb_blk) {
            sl.vb_pre_launched = c->first_block ? chain_slots : (unsigned)std::min<uint64_t>(chain_slots, c->batch_hint + c->batch_hint / 2 + 64);
            // (two work items per lane once the items fill the chip or other blocks do; one per lane for a lone small block)
            const int vb_packed = (c->depth > 1 || c->vb_items_hint > 64ull * 4ull * (uint64_t)c->n_cus * 3ull / 2ull) ? 1 : 0;
            sl.vb_items_launched = c->first_block ? sl.vb_cap : (unsigned)std::min<uint64_t>(sl.vb_cap, c->vb_items_hint + c->vb_items_hint / 2 + 1024);
            HIP_OK(fx_launch_vbpre(0, sl.vb_pre_launched, st, sl.d_pjobs.p, sl.d_dec_list.p + 2 * (size_t)list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p, c->d_tables));
            HIP_OK(fx_launch_vbitems(0, sl.vb_items_launched, st, sl.d_pjobs.p, sl.d_vb_items.p, sl.vb_cap, hdr_pay, sl.d_bufA.p, sl.d_bufB.p, sl.d_vb_dw.p, sl.d_vb_vec.p,
                                     sl.d_vb_st.p, c->vb_debug, vb_packed, (c->first_block || c->vbfix_hint || c->vb_debug) ? 1 : 0));
            HIP_OK(fx_launch_vbfinish(0, sl.vb_pre_launched, st, sl.d_pjobs.p, sl.d_dec_list.p + 2 * (size_t)list_cap, hdr_pay, sl.d_bufA.p, sl.d_bufB.p, sl.d_vb_dw.p,
                                      sl.d_vb_vec.p, sl.d_vb_st.p, sl.d_dec_list.p + 3 * (size_t)list_cap, list_cap, sl.h_out.p, sl.h_recs.p, sl.h_hdr.p));
            // frames whose hand-overs could not be verified (a block that ran again and ended differently): the wave-per-frame decoder,
            // launched with the chain only while such frames keep turning up (the count reaches the host either way; what a launch
            // did not cover is decoded when the block is collected)
            sl.fb_launched = (c->first_block || c->fb_hint || c->vb_debug) ? (uint32_t)std::min<uint64_t>(chain_slots, std::max<uint64_t>(kFallbackWaves, c->fb_hint + c->fb_hint / 2 + 16)) : 0u;
            HIP_OK(fx_launch_paydec(0, 0, 0, sl.fb_launched, 1u, st, sl.d_pjobs.p, sl.d_dec_list.p + 3 * (size_t)list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p,
                                    sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, sl.h_hdr.p));
        } else sl.vb_pre_launched = sl.vb_items_launched = sl.fb_launched = 0;
        if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[7], st));
        // (a copy kernel: it knows how many symbols the block really holds)
        if (c->cfg.want_framesyms)
            HIP_OK(fx_launch_symcopy(2u * (unsigned)c->n_cus, st, hdr_pay, sl.d_symraw.p, sl.h_framesyms.p));
    } else {
        HIP_OK(hipEventRecord(sl.ev[5], st));
        if (tl >= 1) HIP_OK(hipEventRecord(sl.ev[6], st));
        if (tl >= 2) HIP_OK(hipEventRecord(sl.ev[7], st));
        c->carry_reader[b % 3] = sl.ev[5];
    }
    HIP_OK(hipEventRecord(sl.ev[8], st));
    return 0;
}

int fxrx_submit(fxrx_ctx *c, const void *const *iq, const uint64_t *n_samples, int on_device)
{
    if (!c || !iq || !n_samples) { set_err("fxrx_submit: null argument"); return FXRX_ERR_ARG; }
    if (c->inflight >= c->depth) { set_err("fxrx_submit: pipeline full, call fxrx_collect first"); return FXRX_ERR_STATE; }
    HIP_OK(hipSetDevice(c->cfg.device));
    const auto t_enter = std::chrono::steady_clock::now();
    const unsigned NS = c->cfg.n_streams;
    const unsigned nslots = c->depth + 1;
    Slot &sl = *c->slots[c->head];
    sl.out.clear(); sl.timing = fxrx_timing{};
    sl.kept_hops = sl.kept_cheap = sl.kept_vhops = sl.kept_vfail = sl.kept_repairs = 0;
    const unsigned noskip_before = c->noskip_left;
    sl.force_noskip = c->noskip_left > 0; if (c->noskip_left) c->noskip_left--;
    sl.seq = c->seq;
    sl.x.assign(NS, nullptr); sl.n.assign(n_samples, n_samples + NS); sl.snap.assign(NS, StreamSnap{});
    if (sl.d_in.size() < NS) sl.d_in.resize(NS);
    // A submit that fails leaves the context as it found it: per-stream positions, freshness and tail bounds are restored, so
    // that the next block -- the same samples again, or others -- continues from the state before the call.  (Caller-recoverable
    // failures happen before anything is launched: argument checks, arena sizing, allocations.)
    struct Undo { fxrx_ctx_s *c; std::vector<StreamState> st; unsigned noskip; bool armed = true;
                  ~Undo() { if (armed) { c->st = st; c->noskip_left = noskip; } } } undo{ c, c->st, noskip_before };
    uint64_t total_new = 0;
    for (unsigned s = 0; s < NS; s++) {
        StreamState &S = c->st[s];
        const uint64_t nn = n_samples[s];
        total_new += nn;
        if (on_device) sl.x[s] = (const float2 *)iq[s];
        else {
            if (sl.d_in[s].reserve(nn + 1)) return FXRX_ERR_HIP;
            if (nn) {
                // page-locked source, moderate size: the shader cores fetch it (fx_upload_kernel: the launch costs microseconds, the
                // copy call held this thread for hundreds); pageable memory and very large blocks go through the runtime's copy engines
                const void *dsrc = nullptr;
                if (nn * sizeof(float2) <= kUploadKernelMax && pinned_device_ptr(iq[s], &dsrc)) HIP_OK(fx_launch_upload(sl.st, dsrc, sl.d_in[s].p, nn * sizeof(float2), (unsigned)c->n_cus));
                else HIP_OK(hipMemcpyAsync(sl.d_in[s].p, iq[s], nn * sizeof(float2), hipMemcpyHostToDevice, sl.st));
            }
            sl.x[s] = sl.d_in[s].p;
        }
        sl.snap[s].tot0 = S.total; sl.snap[s].fresh_start = S.fresh_start; sl.snap[s].carry_bound = S.fresh_start ? 0 : S.carry_bound;
        sl.snap[s].noskip = S.noskip_left > 0; if (S.noskip_left) S.noskip_left--;
        S.total += (int64_t)nn; S.fresh_start = false;
        S.carry_bound = std::min<int64_t>(S.carry_cap, sl.snap[s].carry_bound + (int64_t)nn);
    }
    if (c->timing_level >= 2 && !c->ref_event) {
        if (hipEventCreate(&c->ref_event) == hipSuccess) { (void)hipEventRecord(c->ref_event, sl.st); (void)hipEventSynchronize(c->ref_event); c->ref_host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    }
    sl.dbg_submit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->ref_host_ms;
    sl.timing.samples = total_new;
    if (c->debug_fail_submit) { c->debug_fail_submit--; set_err("fxrx_submit: injected failure (fxrx_debug_fail)"); return FXRX_ERR_STATE; }
    if (g_prof_on) g_prof[0] += prof_now() - std::chrono::duration<double, std::milli>(t_enter.time_since_epoch()).count();
    int r = enqueue_block(c, sl);
    if (r) return r;
    undo.armed = false;
    c->seq++;
    sl.busy = true;
    c->head = (c->head + 1) % nslots; c->inflight++;
    sl.host_submit_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    return 0;
}

// Two things the device cannot finish by itself, both rare, both handled here when the block concerned is collected:
//  * FX_BLK_NEEDS_REPAIR: the lean chain kernel met a stretch that has to be walked again (a hand-off target missing from
//    the next list, a skipped hop on which the exact detector fires, a full frame table).  The full-size chain kernel does
//    that; the block's back part (chain, plan, payload) is enqueued again with it.
//  * FX_BLK_CARRY_OVERFLOW: a tail did not fit its carry buffer.  The block's own results are good; the buffers are grown
//    and the tail is copied by hand.
// Either way the state the block left was marked invalid, so every block behind it did nothing: they are enqueued again,
// in order, once the state is there.
// Repair rounds.  The lean chain kernel, asked to, rewrites the job of every segment whose predecessor's hand-off target it
// cannot find -- start / floor / fresh := the predecessor's hand-off hop state, exact detector on every hop -- and lists
// those segments; they are walked again, all at once, and the chain kernel runs again: a re-walked segment is entered plainly
// when its predecessor still hands over that state (it does unless the predecessor was itself re-walked and ended
// differently: then the next round mends it).  Low SNR is where this matters: marginal detections depend on where the
// detector's hops happen to fall, so speculative lists and true chain disagree dozens of times per block, and the
// full-size chain kernel would walk those stretches one after the other.  Returns 0: stitched, 1: leave it to that kernel.
static int repair_rounds(fxrx_ctx_s *c, Slot &sl)
{
    if (c->chain_slow) return 1;
    const unsigned NS = c->cfg.n_streams;
    const unsigned mode = c->cfg.mode == FXRX_MODE_DETECTOR ? FX_MODE_DETECT : FX_MODE_FLEXRX;
    hipStream_t st = sl.st;
    FxWalkJob *d_jobs = reinterpret_cast<FxWalkJob *>(sl.d_desc.p);
    const FxStreamDesc *d_streams = reinterpret_cast<const FxStreamDesc *>(sl.d_desc.p + sl.o_streams);
    FxBlockHdr *hdr = sl.d_hdr.p;
    for (int round = 0; round < 64; round++) {
        HIP_OK(hipMemsetAsync(hdr, 0, sizeof(FxBlockHdr), st));
        HIP_OK(fx_launch_chainfast(NS, st, d_streams, d_jobs, sl.d_wres.p, sl.d_frames.p, sl.d_chain.p, sl.d_chain_count.p, hdr, 0u, d_jobs, sl.d_req.p, nullptr, 0u));
        FxBlockHdr hh;
        HIP_OK(hipMemcpyAsync(&hh, hdr, sizeof hh, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        if (std::getenv("FXRX_DEBUG_ROUNDS")) std::fprintf(stderr, "[fxrx] repair round %d: flags %x requests %u\n", round, hh.flags, hh.n_repair_req);
        if (!(hh.flags & FX_BLK_NEEDS_REPAIR)) return 0;          // (hdr keeps what the chain kernel left there -- flags, stamps -- for the plan kernel)
        if ((hh.flags & FX_BLK_NEEDS_SLOW) || hh.n_repair_req == 0 || hh.n_repair_req > 2 * sl.NJ) break;   // (a segment is queued at most twice: as a miss target and for a fired hop)
        {   // streams in which a skipped hop fired: their next blocks are walked with the exact detector on every hop from the start
            std::vector<uint32_t> req(hh.n_repair_req);
            HIP_OK(hipMemcpyAsync(req.data(), sl.d_req.p, req.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_OK(hipStreamSynchronize(st));
            const FxWalkJob *hj = reinterpret_cast<const FxWalkJob *>(sl.hp_desc.p);
            for (uint32_t r : req) if ((r & 0x80000000u) && (r & 0x7fffffffu) < sl.NJ) c->st[hj[r & 0x7fffffffu].stream].noskip_left = 32;
        }
        HIP_OK(hipMemsetAsync(hdr, 0, sizeof(FxBlockHdr), st));
        HIP_OK(fx_launch_walk(mode, c->cfg.equalizer ? 1 : 0, hh.n_repair_req, st, d_jobs, sl.d_req.p, sl.d_wres.p, sl.d_frames.p, sl.d_runs.p, sl.run_cap, hdr, c->d_tables, 1, (uint32_t)sl.NJ, nullptr, 0u));
        HIP_OK(hipMemcpyAsync(&hh, hdr, sizeof hh, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        sl.kept_hops += hh.hops; sl.kept_cheap += hh.hops_cheap; sl.kept_repairs += hh.walk_jobs_run ? hh.walk_jobs_run : 0;
    }
    HIP_OK(hipMemsetAsync(hdr, 0, sizeof(FxBlockHdr), st));
    return 1;
}

static int repair_and_replay(fxrx_ctx_s *c, Slot &sl)
{
    const unsigned NS = c->cfg.n_streams;
    const unsigned nslots = c->depth + 1;
    // Blocks behind this one that continue its streams started from the state it failed to leave: they are enqueued again
    // afterwards (and are drained first: a true walker must not read a state entry while it is being rewritten).  Blocks whose
    // streams all start afresh (separate captures) never looked at it; they, and whatever continues *them*, run on untouched.
    unsigned n_dep = 0;
    for (unsigned k = 1; k < c->inflight; k++) {
        const Slot &nx = *c->slots[(c->tail + k) % nslots];
        bool dep = false;
        for (unsigned s = 0; s < NS; s++) dep = dep || !nx.snap[s].fresh_start;
        if (!dep) break;
        n_dep++;
    }
    const hipEvent_t newest_chain = c->prev_chain;
    bool carry_moved = false;
    for (int round = 0; round < 6; round++) {
        HIP_OK(hipStreamSynchronize(sl.st));
        for (unsigned k = 1; k <= n_dep; k++) HIP_OK(hipStreamSynchronize(c->slots[(c->tail + k) % nslots]->st));
        const uint32_t flags = sl.h_hdr.p->flags;
        if (std::getenv("FXRX_DEBUG_ROUNDS")) std::fprintf(stderr, "[fxrx] block %llu at collect: flags %x verify_failures %u (pass %d)\n", (unsigned long long)sl.seq, flags, sl.h_hdr.p->verify_failures, round);
        if (flags & FX_BLK_NEEDS_REPAIR) {
            c->repairs_host++; c->inchain_left = 16;
            const FxBlockHdr &h0 = *sl.h_hdr.p;         // the walk-phase counters are zeroed with the first plan kernel: keep them
            sl.kept_hops += h0.hops; sl.kept_cheap += h0.hops_cheap; sl.kept_vhops += h0.verify_hops; sl.kept_vfail += h0.verify_failures;
            if (!sl.force_noskip && c->cfg.mode != FXRX_MODE_DETECTOR && c->skip_seek && h0.verify_failures > std::max<uint32_t>(16u, (uint32_t)sl.NJ / 8u)) {
                // Skipped hops on which the exact detector fires, all over the block (weak preambles, false alarms: low SNR; a
                // few of them are mended segment by segment in the repair rounds below).  Walk the block
                // again, all segments in parallel, with the exact detector on every hop -- exact by itself, nothing to
                // verify -- and stay in that mode for the next blocks: the channel will not have improved meanwhile.
                sl.force_noskip = true; c->noskip_left = 32;
                if (enqueue_block(c, sl)) return FXRX_ERR_HIP;
            } else {
                // What is left: hand-off targets that the next segment's list does not hold.  Repair rounds mend those in parallel;
                // anything else (or a block that does not settle) goes through the full-size chain kernel, one workgroup per stream.
                const int rr = repair_rounds(c, sl);
                if (rr < 0 || enqueue_back(c, sl, rr == 0 ? kChainDone : kChainFull)) return FXRX_ERR_HIP;
            }
            continue;                                   // (its tail may in turn overflow the carry buffer)
        }
        if (!(flags & FX_BLK_CARRY_OVERFLOW)) break;
        const uint64_t b = sl.seq;
        FxStreamState *hs = c->h_state + (b % kStateRing) * NS, *ds = c->d_state + (b % kStateRing) * NS;
        for (auto &o : c->slots) if (o->busy) HIP_OK(hipStreamSynchronize(o->st));      // (carry buffers are about to move under every block in flight)
        carry_moved = true;
        for (unsigned s = 0; s < NS; s++) {
            if (!hs[s].overflow) continue;
            StreamState &S = c->st[s];
            const int64_t keep = hs[s].carry_len, old_cap = S.carry_cap;
            float2 *old[3] = { S.carry[0], S.carry[1], S.carry[2] };
            S.carry[0] = S.carry[1] = S.carry[2] = nullptr;
            int64_t cap = std::max<int64_t>(2 * old_cap, 2 * keep); cap = (cap + 4095) & ~(int64_t)4095;
            if (alloc_carry(S, cap)) return FXRX_ERR_HIP;
            // tail = logical samples [n - keep, n) of block b: a piece of its own carried tail (old buffer b % 3), then its new samples
            const int64_t nb = (int64_t)sl.n[s], from = nb - keep;
            float2 *dst = S.carry[(b + 1) % 3] + cap - keep;
            if (from < 0) HIP_OK(hipMemcpy(dst, old[b % 3] + old_cap + from, (size_t)(-from) * sizeof(float2), hipMemcpyDeviceToDevice));
            const int64_t n_new = std::min<int64_t>(keep, nb);
            if (n_new > 0) HIP_OK(hipMemcpy(dst + (keep - n_new), sl.x[s] + (nb - n_new), (size_t)n_new * sizeof(float2), hipMemcpyDeviceToDevice));
            for (auto p : old) if (p) (void)hipFree(p);
            hs[s].overflow = 0; hs[s].invalid = 0;
            HIP_OK(hipMemcpy(ds + s, hs + s, sizeof(FxStreamState), hipMemcpyHostToDevice));
        }
        sl.h_hdr.p->flags &= ~(uint32_t)FX_BLK_CARRY_OVERFLOW;
        c->replays++;
        break;
    }
    if (sl.h_hdr.p->flags & (FX_BLK_NEEDS_REPAIR | FX_BLK_CARRY_OVERFLOW)) { set_err("fxrx_collect: block could not be repaired"); return FXRX_ERR_STATE; }
    if (sl.h_hdr.p->flags & FX_BLK_CHAIN_FULL) { set_err("fxrx_collect: chain table overflow (internal sizing error)"); return FXRX_ERR_STATE; }
    // the blocks behind it, oldest first (descriptors are rebuilt: carry buffers may have moved)
    c->prev_chain = sl.ev[3];
    const unsigned n_replay = carry_moved ? c->inflight - 1 : n_dep;
    for (unsigned k = 1; k <= n_replay; k++) {
        Slot &nx = *c->slots[(c->tail + k) % nslots];
        for (unsigned s = 0; s < NS; s++) if (!nx.snap[s].fresh_start) nx.snap[s].carry_bound = c->st[s].carry_cap;
        int r = enqueue_block(c, nx);
        if (r) return r;
    }
    if (n_replay + 1 < c->inflight) c->prev_chain = newest_chain;      // (the newest block in flight was not touched: the next block waits for its chain)
    for (auto &S : c->st) S.carry_bound = S.carry_cap;
    return 0;
}

// frames the decode launches of the chain did not cover (more frames than the grid sized from the previous block, or
// Reed-Solomon frames turning up unannounced): decode them now, on the block's stream, and wait
static int finish_decode(fxrx_ctx_s *c, Slot &sl)
{
    if (sl.vb_blk && sl.h_hdr.p->vb_ticket) {          // frames were handed back: how many (the block is done: a plain copy)
        uint32_t n_fb = 0;
        HIP_OK(hipMemcpy(&n_fb, &(sl.d_hdr.p + 1)->n_vb_fallback, sizeof n_fb, hipMemcpyDeviceToHost));
        sl.h_hdr.p->n_vb_fallback = n_fb; sl.h_hdr.p->vb_ticket = 0;
    }
    const FxBlockHdr &h = *sl.h_hdr.p;
    const bool more_plain = h.n_dec_plain > sl.dec_launched, more_rs = h.n_dec_rs > 0 && sl.rs_launched == 0;   // (the Reed-Solomon instance strides: any launch covers all)
    const bool more_batch = h.n_dec_batch > sl.vb_pre_launched || h.n_vb_items > sl.vb_items_launched;
    const bool more_fb = sl.vb_blk && h.n_vb_fallback > sl.fb_launched;
    if (!more_plain && !more_rs && !more_batch && !more_fb) return 0;
    const uint32_t list_cap = sl.chain_cap + 64 * FX_PLL_CLASSES;
    FxBlockHdr *hdr_pay = sl.d_hdr.p + 1;
    const int soft = c->cfg.soft_decision ? 1 : 0;
    FxPayResult *pres = nullptr;
#ifdef FX_STAMPS
    pres = sl.d_pres.p;
#endif
    if (more_plain)
        HIP_OK(fx_launch_paydec(0, soft, sl.dec_launched, h.n_dec_plain - sl.dec_launched, c->dec_waves, sl.st, sl.d_pjobs.p, sl.d_dec_list.p, hdr_pay, sl.d_hard.p,
                                sl.d_bufA.p, sl.d_bufB.p, sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, nullptr));
    if (more_rs)
        HIP_OK(fx_launch_paydec(1, soft, 0, h.n_dec_rs, 1u, sl.st, sl.d_pjobs.p, sl.d_dec_list.p + list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p,
                                sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, nullptr));
    if (more_batch) {       // (all parts again, for all of the path's frames: they are idempotent -- but for the fallback list, which starts over)
        HIP_OK(hipMemsetAsync(&hdr_pay->n_vb_fallback, 0, sizeof(uint32_t), sl.st));
        HIP_OK(fx_launch_vbpre(0, h.n_dec_batch, sl.st, sl.d_pjobs.p, sl.d_dec_list.p + 2 * (size_t)list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p, c->d_tables));
        HIP_OK(fx_launch_vbitems(0, h.n_vb_items, sl.st, sl.d_pjobs.p, sl.d_vb_items.p, sl.vb_cap, hdr_pay, sl.d_bufA.p, sl.d_bufB.p, sl.d_vb_dw.p, sl.d_vb_vec.p, sl.d_vb_st.p, c->vb_debug, 1, 1));
        HIP_OK(fx_launch_vbfinish(0, h.n_dec_batch, sl.st, sl.d_pjobs.p, sl.d_dec_list.p + 2 * (size_t)list_cap, hdr_pay, sl.d_bufA.p, sl.d_bufB.p, sl.d_vb_dw.p,
                                  sl.d_vb_vec.p, sl.d_vb_st.p, sl.d_dec_list.p + 3 * (size_t)list_cap, list_cap, sl.h_out.p, sl.h_recs.p, sl.h_hdr.p));
        HIP_OK(fx_launch_paydec(0, 0, 0, kFallbackWaves, 1u, sl.st, sl.d_pjobs.p, sl.d_dec_list.p + 3 * (size_t)list_cap, hdr_pay, sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p,
                                sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, sl.h_hdr.p));
        sl.vb_pre_launched = h.n_dec_batch; sl.vb_items_launched = h.n_vb_items; sl.fb_launched = kFallbackWaves;
        HIP_OK(hipStreamSynchronize(sl.st));
        { uint32_t nfb = 0; HIP_OK(hipMemcpy(&nfb, &hdr_pay->n_vb_fallback, sizeof nfb, hipMemcpyDeviceToHost)); sl.h_hdr.p->n_vb_fallback = nfb; sl.h_hdr.p->vb_ticket = 0; }
    }
    const uint32_t n_fb = *(volatile const uint32_t *)&sl.h_hdr.p->n_vb_fallback;
    if (sl.vb_blk && n_fb > sl.fb_launched) {
        HIP_OK(fx_launch_paydec(0, 0, sl.fb_launched, n_fb - sl.fb_launched, c->dec_waves, sl.st, sl.d_pjobs.p, sl.d_dec_list.p + 3 * (size_t)list_cap, hdr_pay,
                                sl.d_hard.p, sl.d_bufA.p, sl.d_bufB.p, sl.d_soft.p, sl.d_dw.p, sl.h_out.p, sl.h_recs.p, pres, c->d_tables, sl.h_hdr.p));
        sl.fb_launched = n_fb;
    }
    HIP_OK(hipStreamSynchronize(sl.st));
    sl.dec_launched = std::max(sl.dec_launched, h.n_dec_plain); if (more_rs) sl.rs_launched = h.n_dec_rs;
    c->late_decodes++;
    return 0;
}

// A block that cannot be completed takes the blocks in flight behind it along (they continue its streams, or at least share its
// arenas' fate): everything in flight is dropped, the slots' device-side counters are cleared, and every stream restarts from a
// freshly reset synchroniser with its next block -- sample positions keep counting, the dropped samples are simply never
// searched.  The context stays usable (unless the HIP runtime itself is gone: then every later call fails the same way).
static void discard_inflight(fxrx_ctx_s *c)
{
    const std::string keep = g_err;
    sync_all(c);
    const unsigned nslots = c->depth + 1;
    while (c->inflight) {
        Slot &s = *c->slots[c->tail];
        s.busy = false; s.out.clear();
        c->tail = (c->tail + 1) % nslots; c->inflight--; c->discarded++;
    }
    for (auto &s : c->slots) {
        if (s->d_hdr.p) (void)hipMemset(s->d_hdr.p, 0, 2 * sizeof(FxBlockHdr));
        if (s->d_plan_ws.p) (void)hipMemset(s->d_plan_ws.p, 0, fx_plan_ws_words() * sizeof(uint32_t));
        if (s->h_hdr.p) std::memset(s->h_hdr.p, 0, sizeof(FxBlockHdr));
    }
    for (auto &S : c->st) { S.fresh_start = true; S.carry_bound = 0; }
    c->last = nullptr; c->prev_chain = nullptr;
    for (auto &e : c->carry_reader) e = nullptr;
    set_err(keep);
}

static int collect_block(fxrx_ctx_s *c);

int fxrx_collect(fxrx_ctx *c)
{
    if (!c) return FXRX_ERR_ARG;
    if (!c->inflight) { set_err("fxrx_collect: nothing in flight"); return FXRX_ERR_STATE; }
    const int r = collect_block(c);
    if (r < 0) discard_inflight(c);
    return r;
}

int fxrx_ready(const fxrx_ctx *c)
{
    if (!c) return FXRX_ERR_ARG;
    if (!c->inflight) return 0;
    const hipError_t e = hipEventQuery(c->slots[c->tail]->ev[8]);
    return e == hipSuccess ? 1 : (e == hipErrorNotReady ? 0 : FXRX_ERR_HIP);
}

unsigned int fxrx_inflight(const fxrx_ctx *c) { return c ? c->inflight : 0; }

int fxrx_debug_fail(fxrx_ctx *c, unsigned int submits, unsigned int collects)
{
    if (!c) return FXRX_ERR_ARG;
    c->debug_fail_submit = submits; c->debug_fail_collect = collects;
    return 0;
}

static int collect_block(fxrx_ctx_s *c)
{
    HIP_OK(hipSetDevice(c->cfg.device));
    const unsigned NS = c->cfg.n_streams;
    const unsigned nslots = c->depth + 1;
    Slot &sl = *c->slots[c->tail];
    const auto tw = std::chrono::steady_clock::now();
    HIP_OK(hipEventSynchronize(sl.ev[8]));
    if (c->debug_fail_collect) { c->debug_fail_collect--; set_err("fxrx_collect: injected failure (fxrx_debug_fail)"); return FXRX_ERR_STATE; }
    {
        const uint32_t flags = sl.h_hdr.p->flags;
        if (flags & FX_BLK_CHAIN_FULL) { set_err("fxrx_collect: chain table overflow (internal sizing error)"); return FXRX_ERR_STATE; }
        // (an invalid block sits behind an overflowing one, whose collect replays it before it is collected itself)
        if (flags & FX_BLK_INVALID) { set_err("fxrx_collect: block has no valid start state"); return FXRX_ERR_STATE; }
        if (flags & (FX_BLK_CARRY_OVERFLOW | FX_BLK_NEEDS_REPAIR)) { int r = repair_and_replay(c, sl); if (r) return r; }
    }
    sl.dbg_collect_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - c->ref_host_ms;
    if (sl.timing_level >= 2 && c->ref_event) { float m = 0; (void)hipEventElapsedTime(&m, c->ref_event, sl.ev[0]); sl.dbg_gpu_start_ms = m; (void)hipEventElapsedTime(&m, c->ref_event, sl.ev[8]); sl.dbg_gpu_done_ms = m; }
    if (c->debug_stop_after) { sl.out.clear(); sl.busy = false; c->last = &sl; c->tail = (c->tail + 1) % nslots; c->inflight--; return 0; }
    if (c->cfg.mode != FXRX_MODE_DETECTOR) { int r = finish_decode(c, sl); if (r) return r; }
    if (sl.h_hdr.p->n_repair_req) c->inchain_left = 16;
    if (sl.h_hdr.p->n_repair_req && sl.h_hdr.p->verify_failures) {
        // the chain mended something by itself: streams in which a skipped hop fired are walked with the exact detector on every
        // hop for their next blocks (the list says which; a rare, small copy)
        std::vector<uint32_t> req(std::min<size_t>(sl.h_hdr.p->n_repair_req, 2 * sl.NJ + 16));
        HIP_OK(hipMemcpy(req.data(), sl.d_req.p, req.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        const FxWalkJob *hj = reinterpret_cast<const FxWalkJob *>(sl.hp_desc.p);
        for (uint32_t r : req) if ((r & 0x80000000u) && (r & 0x7fffffffu) < sl.NJ) c->st[hj[r & 0x7fffffffu].stream].noskip_left = 32;
    }
    sl.timing.host_collectwait_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw).count();
    const FxBlockHdr &h = *sl.h_hdr.p;
    const bool detect = c->cfg.mode == FXRX_MODE_DETECTOR;
    sl.out.resize(h.n_frames);
    uint64_t vb_rep = 0;                       // trellis blocks the batch Viterbi path had to run again (hand-over check failed)
    for (uint32_t i = 0; i < h.n_frames; i++) {
        const FxOutRec &r = sl.h_recs.p[i];
        fxrx_frame &f = sl.out[i].f; std::memset(&f, 0, sizeof f);
        f.stream = r.stream; f.start = r.start; f.cfo_bin = r.offset;
        f.rxy = r.rxy; f.tau = r.tau; f.gamma = r.gamma; f.dphi = r.dphi; f.phi = r.phi; f.pfb_index = r.pfb;
        f.pilot_dphi = r.pilot_dphi; f.pilot_phi = r.pilot_phi; f.pilot_gain = r.pilot_gain;
        f.header_valid = (r.flags & FX_FLAG_HEADER_VALID) ? 1 : 0;
        std::memcpy(f.header, r.header, FX_HDR_DEC);
        f.rssi_db = 20.0f * log10f(r.gamma); f.cfo = r.dphi;
        if (!detect && f.header_valid) {
            f.mod_scheme = r.ms; f.mod_bps = r.bps; f.check = r.check; f.fec0 = r.fec0; f.fec1 = r.fec1;
            f.payload_len = r.pay_len; f.num_framesyms = r.nsym;
            f.payload = sl.h_out.p + r.out_off; f.payload_valid = (int)r.payload_valid;
            vb_rep += r.status >> 8;
            f.evm_sum = r.evm_sum; f.evm_db = 10.0f * log10f(r.evm_sum / (float)(r.nsym ? r.nsym : 1));
            f.framesyms = c->cfg.want_framesyms ? (const fx_complex *)(sl.h_framesyms.p + r.sym_off) : nullptr;
            if (c->cfg.soft_decision && c->cfg.want_framesyms) {
                f.soft_bits = sl.h_soft.p + 8 * (size_t)r.byte_off;
                f.num_soft_bits = 8u * fx::packet_plan(r.pay_len, r.check, r.fec0, r.fec1).l1;
            }
        }
    }
    sl.n_syms = h.sym_total;
    // what the streams carry now (bounds the next blocks' arenas), and the traffic hints for the next launches
    const FxStreamState *hs = c->h_state + (sl.seq % kStateRing) * NS;
    for (unsigned s = 0; s < NS; s++) {
        StreamState &S = c->st[s];
        const int64_t end_total = sl.snap[s].tot0 + (int64_t)sl.n[s];
        if (!S.fresh_start && S.total >= end_total)
            S.carry_bound = std::min<int64_t>(S.carry_bound, std::min<int64_t>(S.carry_cap, hs[s].carry_len + (S.total - end_total)));
    }
    c->frames_hint = h.n_frames; c->plain_hint = h.n_dec_plain; c->batch_hint = h.n_dec_batch; c->vb_items_hint = h.n_vb_items;
    c->fb_hint = h.n_vb_fallback; c->vbfix_hint = h.n_vb_fallback + vb_rep; c->mf_items_hint = h.n_mfblk; c->vb_want_hint = h.vb_want; c->vb_steps_hint = (uint64_t)h.vb_want * (h.vb_blk ? h.vb_blk : 1u); c->first_block = false;
    c->rs_hint = h.n_dec_rs ? h.n_dec_rs : c->rs_hint - std::min<uint64_t>(c->rs_hint, std::max<uint64_t>(1, c->rs_hint / 8));   // (fades out, to zero, over a few dozen blocks without such frames)
    if (h.verify_hops) c->verify_per = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(1, ((uint64_t)h.verify_hops + 4ull * c->n_cus - 1) / (4ull * c->n_cus)));
    fxrx_timing &t = sl.timing;
    float ms = 0;
    if (sl.timing_level >= 2) {
        (void)hipEventElapsedTime(&ms, sl.ev[0], sl.ev[1]); t.walk_ms = ms;
        (void)hipEventElapsedTime(&ms, sl.ev[1], sl.ev[2]); t.seekverify_ms = ms;
        (void)hipEventElapsedTime(&ms, sl.ev[2], sl.ev[4]); t.chain_ms = ms;
        if (!detect) {
            (void)hipEventElapsedTime(&ms, sl.ev[4], sl.ev[5]); t.paymf_ms = ms;
            (void)hipEventElapsedTime(&ms, sl.ev[6], sl.ev[7]); t.paydec_ms = ms;
        }
    }
    if (sl.timing_level >= 1 && !detect) { (void)hipEventElapsedTime(&ms, sl.ev[5], sl.ev[6]); t.paypll_ms = ms; }
    t.total_ms = t.walk_ms + t.seekverify_ms + t.chain_ms + t.paymf_ms + t.paypll_ms + t.paydec_ms;
    t.hops = h.hops + sl.kept_hops; t.hops_cheap = h.hops_cheap + sl.kept_cheap; t.walk_jobs = sl.NJ; t.repairs = h.repairs + sl.kept_repairs + h.n_repair_req; t.frames = h.n_frames;
    t.payload_symbols = h.sym_total; t.verify_hops = h.verify_hops + sl.kept_vhops; t.verify_failures = h.verify_failures + sl.kept_vfail;
    t.host_submit_ms = sl.host_submit_ms; t.host_walkwait_ms = 0.0; t.walk_mode = sl.any_late ? 1 : 0; t.replays = c->replays + c->repairs_host;
    t.vb_blocks = h.n_vb_items; t.vb_repairs = vb_rep; t.late_decodes = c->late_decodes; t.vb_fallbacks = h.n_vb_fallback;
    sl.busy = false; c->last = &sl;
    c->tail = (c->tail + 1) % nslots; c->inflight--;
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

// diagnostic builds (-DFX_STAMPS): walker phase clocks of the last collected block, summed over its walk jobs / of the slowest job
static int walk_stamps(const fxrx_ctx *c, uint64_t sum[4], uint64_t maxjob[8])
{
    for (int i = 0; i < 4; i++) sum[i] = 0;
    for (int i = 0; i < 8; i++) maxjob[i] = 0;
#ifdef FX_STAMPS
    if (!c->last || !c->last->NJ) return 0;
    std::vector<FxWalkResult> r(c->last->NJ);
    if (hipMemcpy(r.data(), c->last->d_wres.p, r.size() * sizeof(FxWalkResult), hipMemcpyDeviceToHost) != hipSuccess) return FXRX_ERR_HIP;
    for (const auto &w : r) {
        uint64_t tot = 0;
        for (int i = 0; i < 4; i++) { sum[i] += w.stamp[i]; tot += w.stamp[i]; }
        if (tot > maxjob[7]) { for (int i = 0; i < 4; i++) maxjob[i] = w.stamp[i]; maxjob[4] = w.hops; maxjob[5] = w.hops_cheap; maxjob[6] = w.n_frames; maxjob[7] = tot; }
    }
#endif
    return 0;
}
// diagnostic builds: per walk job of the last collected block 6 words -- stamp[0..3], hops, frames (detector-only mode: start / end on the
// 100 MHz wall clock, CU id, 0); returns the number of jobs
int fxrx_debug_walk_jobs(const fxrx_ctx *c, uint32_t *out, unsigned int cap_jobs)
{
    if (!c || !out || !c->last) return FXRX_ERR_ARG;
    const size_t nj = std::min<size_t>(c->last->NJ, cap_jobs);
    std::vector<FxWalkResult> r(nj);
    if (nj && hipMemcpy(r.data(), c->last->d_wres.p, nj * sizeof(FxWalkResult), hipMemcpyDeviceToHost) != hipSuccess) return FXRX_ERR_HIP;
    for (size_t i = 0; i < nj; i++) { for (int k = 0; k < 4; k++) out[6 * i + k] = r[i].stamp[k]; out[6 * i + 4] = r[i].hops; out[6 * i + 5] = r[i].n_frames; }
    return (int)nj;
}
// shader clocks of the chain kernel's phases (stream 0) of the last collected block: whole fast path, -, pointer chase, kernel total
int fxrx_debug_chain_stamps(const fxrx_ctx *c, uint32_t out[8]) { if (!c || !c->last) return FXRX_ERR_ARG; std::memcpy(out, c->last->h_hdr.p->stamp, 8 * sizeof(uint32_t)); return 0; }
int fxrx_debug_walk_stamps(const fxrx_ctx *c, uint64_t out[4]) { if (!c) return FXRX_ERR_ARG; uint64_t mj[8]; return walk_stamps(c, out, mj); }
int fxrx_debug_walk_maxjob(const fxrx_ctx *c, uint64_t out[8]) { if (!c) return FXRX_ERR_ARG; uint64_t sm[4]; return walk_stamps(c, sm, out); }

// diagnostic: decode-phase shader-clock deltas of frame i of the last collected block (zeros unless built with -DFX_STAMPS)
int fxrx_debug_stamps(const fxrx_ctx *c, unsigned int i, uint32_t out[8])
{
    if (!c || !c->last || i >= c->last->out.size()) return FXRX_ERR_ARG;
    std::memset(out, 0, 8 * sizeof(uint32_t));
#ifdef FX_STAMPS
    if (c->last->d_pres.p && hipMemcpy(out, c->last->d_pres.p[i].stamp, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return FXRX_ERR_HIP;
#endif
    return 0;
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
