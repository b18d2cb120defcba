// fx_common.h -- constants and plain-data records shared by host code and HIP kernels.
//
// Frame geometry of liquid-dsp's flexframe as used by gr::liquiddsp::flex_rx / frame_detector_cc
// (reference parameters: /root/reference/lib/frame_detector_cc_impl.h:34-36,
// /root/reference/lib/frame_detector_cc_impl.cc:46-55, /root/reference/lib/flex_tx_impl.cc:52,58).
#pragma once
#include <stdint.h>
#include <hip/hip_vector_types.h>

#define FX_K             2
#define FX_M             7
#define FX_BETA          0.3f
#define FX_NPFB          32
#define FX_MF_TAPS       28
#define FX_PROTO_LEN     897
#define FX_PN_LEN        64
#define FX_S_LEN         156
#define FX_NFFT          512
#define FX_HOP           256
#define FX_RANGE         24
#define FX_HDR_USER      14
#define FX_HDR_DEC       20
#define FX_HDR_CRC       24      /* 20 + CRC32 */
#define FX_HDR_E0        27      /* after SECDED(72,64) */
#define FX_HDR_ENC       54      /* after Hamming(8,4) */
#define FX_HDR_MOD       216
#define FX_HDR_PILOTS    15
#define FX_HDR_SYM       231
#define FX_PILOT_SPACING 16
#define FX_PROTOCOL      102
#define FX_SYM0_HDR      78      /* 2m + 64 */
#define FX_SYM0_PAY      309
#define FX_EQ_TAPS       13      /* optional equaliser: 2 k p + 1 taps at 2 samples/symbol, p = 3 */
#define FX_EQ_DELAY      3       /* its delay: every symbol instant moves this many symbols later */
#define FX_EQ_MU         0.05f

// Waves per workgroup of the detector kernels (4 or 8).  8 halves the latency of a detector hop (49 CFO-sweep
// transforms in 7 rounds instead of 13): right for the dense detector-only walker.  The flex_rx
// walker spends most of its time in short serial phases at 256 VGPRs per wave; with 4 waves a workgroup holds half
// a CU's register file instead of all of it, and payload waves of the blocks in flight fit beside it (measured:
// +9 % throughput with blocks in flight, same latency alone).  The seek verifier has no serial chain to shorten;
// small workgroups pack better.
#ifndef FX_FLEX_WAVES
#define FX_FLEX_WAVES 4
#endif
#ifndef FX_DETECT_WAVES
#define FX_DETECT_WAVES 4
#endif
#ifndef FX_VERIFY_WAVES
#define FX_VERIFY_WAVES 4
#endif

// header-stored enums (liquid.h v1.3.x numbering, recalled; see include/fxrx.h)
enum { FX_CRC_UNKNOWN = 0, FX_CRC_NONE, FX_CRC_CHECKSUM, FX_CRC_8, FX_CRC_16, FX_CRC_24, FX_CRC_32 };
enum {
    FX_FEC_UNKNOWN = 0, FX_FEC_NONE = 1, FX_FEC_HAMMING74 = 4, FX_FEC_HAMMING84 = 5, FX_FEC_HAMMING128 = 6,
    FX_FEC_GOLAY2412 = 7, FX_FEC_SECDED2216 = 8, FX_FEC_SECDED3932 = 9, FX_FEC_SECDED7264 = 10,
    FX_FEC_CONV_V27 = 11, FX_FEC_CONV_V27P23 = 15, FX_FEC_CONV_V27P34 = 16, FX_FEC_CONV_V27P45 = 17,
    FX_FEC_CONV_V27P56 = 18, FX_FEC_CONV_V27P67 = 19, FX_FEC_CONV_V27P78 = 20, FX_FEC_RS_M8 = 27
};
enum {
    FX_MODEM_UNKNOWN = 0, FX_MODEM_PSK2 = 1, FX_MODEM_PSK4 = 2, FX_MODEM_PSK8 = 3, FX_MODEM_PSK16 = 4,
    FX_MODEM_DPSK2 = 9, FX_MODEM_DPSK4 = 10, FX_MODEM_DPSK8 = 11, FX_MODEM_ASK4 = 18,
    FX_MODEM_QAM16 = 27, FX_MODEM_QAM32 = 28, FX_MODEM_QAM64 = 29, FX_MODEM_QPSK = 40
};

// ---- walker job / result records ----
// Coordinates: every block of a stream has its own origin, logical index 0 = the first NEW sample of the block.  The
// tail the previous block left unconsumed sits at negative indices, right-aligned in a carry buffer: index p < 0 reads
// xa_end[p].  Nothing below the zero-floor (samples before the last synchroniser reset) is ever read.
enum { FX_MODE_FLEXRX = 0, FX_MODE_DETECT = 1 };
enum {
    FX_EXIT_STOP = 0,        // reached job.stop, hand-off target recorded (or not requested)
    FX_EXIT_NEED_DATA = 1,   // ran out of samples in SEEK/ALIGN/header: resume at (pos, fresh, floor)
    FX_EXIT_PAYLOAD = 2,     // last frame descriptor is incomplete: payload runs past the end of data
    FX_EXIT_TABLE_FULL = 3,  // frame table exhausted: resume at (pos, fresh, floor)
    FX_EXIT_INVALID = 4      // the state this walker was to start from does not exist (see FxStreamState.invalid)
};
enum { FX_FLAG_HEADER_VALID = 1, FX_FLAG_INCOMPLETE = 2, FX_FLAG_EXACT = 4 /* walker was in exact (locked) mode */,
       FX_FLAG_FLOOR_CLEAR = 8 /* the walker's zero-floor was <= start at detection: no sample the frame reads was masked */,
       FX_FLAG_SPAN_BAD = 16   /* set by fx_seekverify_kernel: the exact detector fires on a hop this frame's seek skipped */,
       FX_FLAG_SEEK_FRESH = 32 /* the seek that led here started from a freshly reset detector */,
       FX_FLAG_SPAN_EXACT = 64 /* that seek ran the exact detector on every hop: nothing to verify */ };

// per-stream state a block leaves to its successor (device memory; the chain kernel also mirrors it to the host)
struct FxStreamState {
    int64_t  pos, floor;    // resume hop / zero-floor in the NEXT block's coordinates (<= 0)
    int64_t  carry_len;     // samples kept, right-aligned at the end of the carry buffer the next block reads
    uint32_t fresh;         // the detector restarts freshly reset at pos
    uint32_t invalid;       // no state: the tail did not fit the carry buffer (overflow, set by this block) or the state
    uint32_t overflow;      //   this block started from was itself invalid.  The host replays the blocks behind it.
    uint32_t pad_;
};

struct FxWalkJob {
    const float2 *x;        // new samples of the block; logical index 0 == x[0]
    const float2 *xa_end;   // end of the carried tail (index p < 0 reads xa_end[p]); NULL: none
    int64_t  n;             // new samples
    int64_t  start;         // first new-half position (detector restarts / resumes here)
    int64_t  stop;          // segment end: no new detection is *started* at pos >= stop
    int64_t  floor;         // samples below this index read as zero (last synchroniser reset)
    uint32_t fresh;         // 1: overlap half is zeros (just reset); 0: overlap = x[start-256, start)
    uint32_t mode;          // FX_MODE_*
    uint32_t handoff;       // 1: past stop, keep seeking until the next detection and record it
    uint32_t prelock;       // 1: speculative start: anything found before the first trustworthy frame is
                            //    tentative (see fx_host.cpp); 0: state is the true sequential state
    uint32_t frame_base;    // first slot of this job in the frame table
    uint32_t max_frames;    // slots available
    float    threshold;
    uint32_t no_skip;       // 1: a locked flex_rx walker runs the full detector on every hop (exact by itself);
                            // 0: it may skip hops its coarse scan finds empty -- every skipped hop is then re-checked
                            //    by fx_seekverify_kernel before fx_chain_kernel trusts the span
    const FxStreamState *state_in;   // non-NULL: true walker of a continuing stream: start / floor / fresh come from here
    uint32_t stream;
    uint32_t verify_per;    // hops per verification run
    uint32_t eq;            // 1: equaliser stage on (fxrx_config.equalizer)
    uint32_t pad_;
};

struct FxFrameHead {        // 152 bytes: everything the walker keeps in registers while it builds a frame record
    int64_t  start;         // index of aligned sample 0 (may be < floor: zeros there)
    int64_t  next;          // restart position after this frame (flex_rx) / next new-half (detect)
    int64_t  seek_pos, seek_floor;   // the seek that led here started at this hop / with this floor ...
    int64_t  det_pos;                // ... and detected at this hop: hops [seek_pos, det_pos) saw nothing
    int32_t  offset;        // CFO bin of the coarse search
    float    rxy, tau, gamma, dphi, phi;
    uint32_t pfb; int32_t mfc0;
    uint32_t mix_th, mix_dl; float mf_scale;
    float    pilot_dphi, pilot_phi, pilot_gain;
    uint32_t pll_th; float pll_f;
    uint32_t flags;
    uint32_t pay_len, ms, check, fec0, fec1, pay_sym_len;
    uint8_t  header[FX_HDR_DEC];
};
struct FxFrame : FxFrameHead {   // 256 bytes
    float2   eq[FX_EQ_TAPS];    // equaliser taps after training on the p/n symbols (equaliser stage on; else untouched)
};

struct FxWalkResult {
    uint32_t n_frames;      // descriptors written
    uint32_t exit_code;     // FX_EXIT_*
    int64_t  pos;           // resume position (new-half start)
    int64_t  floor;
    uint32_t fresh;
    uint32_t has_handoff;   // hand-off target valid
    int64_t  handoff_start; int32_t handoff_offset; uint32_t hops;
    float    handoff_rxy; uint32_t hops_cheap;
    int64_t  tail_pos, tail_floor;   // the seek in progress at exit started here: hops [tail_pos, end) saw nothing,
    int64_t  handoff_pos;            // end = handoff_pos (hop of the hand-off detection) or pos
    uint32_t handoff_clear;          // the hand-off detection saw nothing masked by the floor (floor <= handoff_start)
    uint32_t tail_flags;             // FX_FLAG_SEEK_FRESH / FX_FLAG_SPAN_EXACT of the seek in progress at exit;
                                     // FX_FLAG_SPAN_BAD is or-ed in by fx_seekverify_kernel
    uint32_t stamp[4];      // diagnostic builds: shader clocks in coarse scan / exact seek / align / header
};

// ---- seek verification (fx_seekverify_kernel): a run of consecutive detector hops that must all come up empty.  The
// walkers emit the runs themselves as they close a seek span (atomic counter in the block header). ----
struct FxVerifyRun {
    int64_t  pos;           // first hop (new-half start); hop h sits at pos + 256 h
    int64_t  floor;
    uint32_t job;           // walk job the span belongs to (sample pointers, threshold)
    uint32_t owner;         // frame-table slot whose seek span this is; 0x80000000 | job: that job's tail span
    uint32_t nhops, pad_;
};

// ---- one stream of a block, as fx_chain_kernel sees it ----
struct FxStreamDesc {
    const float2 *x, *xa_end; int64_t n;
    uint32_t first_job, n_jobs;
    uint32_t chain_base, chain_cap;      // this stream's region of the chain table
    uint32_t repair_base, repair_cap;    // frame-table region for walks the chain kernel does itself
    const FxStreamState *state_in;       // NULL: the stream starts freshly reset at 0
    FxStreamState *state_out, *state_out_host;
    float2  *carry_out_end;              // the tail goes to carry_out_end[-carry_len, 0)
    int64_t  carry_cap;
    int64_t  abs_base;                   // absolute index (since the last reset) of logical sample 0
};

enum { FX_BLK_INVALID = 1 /* some stream started from an invalid state: nothing in this block counts */,
       FX_BLK_CARRY_OVERFLOW = 2, FX_BLK_CHAIN_FULL = 4 /* chain table exhausted (sizing bug) */,
       FX_BLK_NEEDS_REPAIR = 8 /* fx_chainfast_kernel met something that takes a walk: a stream is left without chain and state */,
       FX_BLK_NEEDS_SLOW = 16 /* ... and not just hand-off misses (which repair rounds mend in parallel): the full-size chain kernel has to */ };
#define FX_PLL_CLASSES 12
struct FxBlockHdr {                      // device memory, zeroed at submit; mirrored to the host by fx_plan_kernel
    uint32_t n_runs;                     // verification runs emitted (may exceed the capacity: the excess spans are marked bad)
    uint32_t flags;
    uint32_t n_frames, n_pjobs, n_mfblk, n_dec_plain, n_dec_rs, n_dec_batch;   // n_dec_batch: frames decoded by the batch Viterbi path
    uint32_t n_vb_items, vb_blk;         // its forward-pass work items (frame, trellis block) / trellis steps per block
    uint32_t vb_want;                    // work items the block's traffic asked for (those beyond the arena go the wave-per-frame way)
    uint32_t n_vb_fallback;              // frames the batch path could not verify: decoded again by the wave-per-frame decoder
    uint32_t n_repair_req;               // repair rounds: segments queued for a walk from their true start state
    uint32_t runs_done;                  // verification runs emitted by the speculative walkers (verified before the true walkers start)
    uint32_t pll_cnt[FX_PLL_CLASSES];    // frames per modulation class
    uint32_t pll_base[FX_PLL_CLASSES + 1];   // first list slot of each class (multiples of 64: a wave never mixes classes)
    uint64_t sym_total, byte_total, dw_total, out_total;
    uint32_t hops, hops_cheap, repairs, verify_hops, verify_failures, walk_jobs_run;
    uint32_t vb_ticket;                  // host mirror only: fx_vbfinish_kernel handed frames back to the wave-per-frame decoder (n_vb_fallback on the device says how many)
    uint32_t done;                       // host mirror only: written last
    uint32_t stamp[8];
};

// ---- one result record per chain frame, written straight into pinned host memory ----
struct FxOutRec {
    int64_t  start;                      // absolute sample index of aligned sample 0
    uint32_t stream; int32_t offset;
    float    rxy, tau, gamma, dphi, phi;
    uint32_t pfb;
    float    pilot_dphi, pilot_phi, pilot_gain;
    uint32_t flags;                      // FX_FLAG_HEADER_VALID
    uint32_t pay_len, ms, check, fec0, fec1, nsym, bps;
    uint32_t sym_off, out_off;           // payload symbols / decoded bytes of this frame in the block's arenas
    float    evm_sum;                    // fx_paypll_kernel
    uint32_t payload_valid;              // fx_paydec_kernel
    uint32_t status;
    uint8_t  header[FX_HDR_DEC];
    uint32_t byte_off;                   // the frame's offset in the byte arenas (soft values: 8 x this)
};

// ---- frame generator (fx_txgen_kernel) ----
struct FxTxJob {
    uint64_t out_off;       // first output sample of the frame in the destination buffer
    uint32_t nsym;          // symbols incl. the 2m flush zeros (output = 2 nsym samples)
    uint32_t npay;          // payload symbols
    uint32_t ms;            // payload modulation
    uint32_t head_off;      // offset of the frame's 216 header symbol indices (QPSK words), one byte each
    uint32_t idx_off;       // offset of its payload symbol indices (one byte each)
    uint32_t pad_;
    float    taps[32];      // transmit pulse (29 taps; designed with the frame's fractional delay)
};

// packet encoder on the GPU (fx_txenc_kernel), one wave per frame; the mirror image of the decoder's front end
struct FxTxEncJob {
    uint32_t pay_off;       // payload bytes in the input arena
    uint32_t n, check, fec0, fec1;   // payload length, CRC scheme, the two codes of the chain
    uint32_t k, l0, l1;     // n + crc_len; bytes after fec0; bytes after fec1
    uint32_t perm0_off, perm1_off;   // interleaver gather tables (bit i of the output = bit perm[i] of the input)
    uint32_t buf_off;       // scratch (two buffers of stride >= max(l0, l1) + 16)
    uint32_t idx_off;       // output: payload symbol indices for fx_txgen_kernel
    uint32_t npay, ms;
};
// constant tables of the generator kernels
struct FxTxTables {
    float2   sc[1024];
    uint32_t golenc[4096];
    uint16_t h128enc[256];
    uint8_t  h74enc[16], h84enc[16], sdcol[64], sd22col[16], sd39col[32];
    uint8_t  rsexp[512], rslog[256], rsgen[40];
    float2   pn[FX_PN_LEN], pilots[16];
};

// ---- synthetic channel of the generator (fx_channel_kernel): per stream, y[n] = gain x[n] exp(j (th0 + n dl)) + sigma w[n] ----
struct FxChannel {
    uint32_t th0, dl;       // carrier phase / phase increment per sample, 2^32 = one turn
    float    gain, sigma;   // sigma: standard deviation of the noise per real dimension
    uint64_t seed;          // key of the counter-based noise generator (Philox-4x32-10, counter = sample index / 2)
};

// ---- payload stage records (built on the device by fx_plan_kernel) ----
struct FxPayJob {           // one per chain frame; nsym == 0: no payload stage (header invalid / detector mode)
    const float2 *x, *xa_end;   // stream samples (two pieces, see FxWalkJob)
    int64_t  start;         // aligned sample 0
    uint32_t mix_th, mix_dl; float mf_scale;
    uint32_t pfb; int32_t mfc0;
    uint32_t pll_th; float pll_f;
    uint32_t ms, bps;
    uint32_t nsym;          // payload symbols
    uint32_t sym_off;       // offset (symbols) of this frame in the symbol arenas
    // decode plan
    uint32_t pay_len, check, fec0, fec1;
    uint32_t k;             // pay_len + crc_len
    uint32_t l0, l1;        // bytes after fec0 / after fec1
    uint32_t byte_off;      // offset of this frame's scratch in the byte arenas (stride >= l1+8)
    uint32_t dw_off;        // offset in the decision-word arena (u64 units)
    uint32_t out_off;       // offset of decoded payload in the output arena
    uint32_t pad_;          // 1: the frame has a payload stage
    uint32_t eq;            // 1: equaliser on: taps at chain[chain_idx].eq, symbol instants FX_EQ_DELAY later
    uint32_t chain_idx;     // the frame's slot in the chain table
    uint32_t vb_off, vb_nblk;   // batch Viterbi path: first work item / number of trellis blocks of this frame
};

// batch Viterbi: trellis steps of warm-up a block runs before its own region (survivor paths merge within a few
// constraint lengths; whether they did is verified, see fx_vbpost_kernel)
#ifndef FX_VB_WARM
#define FX_VB_WARM 96
#endif

struct FxPayResult {        // diagnostic builds (-DFX_STAMPS) only: shader-clock deltas of the decode phases
    uint32_t stamp[8];
};
