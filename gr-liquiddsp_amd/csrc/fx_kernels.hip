// fx_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the flexframe receive path.
//
// What liquid-dsp does one sample at a time inside flexframesync_execute / qdetector_cccf_execute
// (call sites /root/reference/lib/flex_rx_impl.cc:213 and /root/reference/lib/frame_detector_cc_impl.cc:77)
// is regrouped here into the kernels of one stream-ordered chain per block of input (fx_host.cpp enqueues it):
//
//   fx_walk_kernel      one workgroup walks one stream segment through the synchroniser's state
//                       machine: hop-wise FFT cross-correlation with the +-24-bin CFO sweep
//                       (qdetector SEEK), ALIGN estimates (tau, gamma, dphi, phi), then -- flex_rx mode --
//                       NCO mix + polyphase MF of the preamble/header span, pilot sync, header decode.
//                       Emits one FxFrame per detection, the hand-off to the next segment, and the runs of hops it skipped.
//   fx_seekverify_kernel  the exact detector on every skipped hop (thousands of independent workgroups).
//   fx_chainfast_kernel   stitches the segments' lists into the sequential machine's list; resume state, carried tail
//                       (fx_chain_kernel: the full-size fallback that can walk by itself).
//   fx_plan_kernel, fx_planlists_kernel   arena offsets, jobs, result records, work lists of the payload stage.
//   fx_paymf_kernel     per frame, data parallel: closed-form NCO mix (32-bit phase) + fixed-branch
//                       polyphase matched filter + decimate-by-2 over the payload span.
//   fx_paypll_kernel    one lane per frame: the decision-directed payload PLL (the only true
//                       sample-to-sample recurrence of the path), hard demod, EVM.
//   fx_vbpre / fx_vbfwd / fx_vbfix / fx_vbtrace / fx_vbfinish   packet decode with the K=7 Viterbi run a lane per trellis
//                       block (two blocks per lane, packed 16-bit metrics), hand-overs speculated and verified.
//   fx_paydec_kernel    one wavefront per frame (lane = trellis state): soft decisions, Reed-Solomon, frames without a
//                       convolutional inner code, and whatever the batch path hands back.
//   fx_softdemod_kernel, fx_symcopy_kernel, fx_txgen_kernel, fx_txenc_kernel   optional stage / results / frame generator.
//
// No MFMA anywhere: nothing on this path is a dense contraction.  Taps, windows, spectra and the
// FFT exchange buffers live in LDS; IQ is read from HBM as coalesced float2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "fx_device.h"

// 8 waves: the 49 CFO-sweep transforms of a hop take 7 rounds instead of 13.  The 512-sample window is still
// handled by the first 256 threads (HALF); reductions keep the 4-wave tree order of the canonical arithmetic.
#define HALF         256

template <int WW> struct WalkLdsT {
    static constexpr int WAVES = WW;
    static constexpr bool SWEEP_PAIRS = WW == 4;   // the detector sweep takes two bins per wave at a time (seek_sweep)
    float2 win[FX_NFFT];            // time window of the current hop / aligned window
    float2 X[FX_NFFT];              // its spectrum
    float2 S[FX_NFFT];              // template spectrum
    float2 scr[WW][576];            // per-wave FFT exchange buffers
    float2 v[640];                  // mixed-down samples of the preamble+header span  } dead during a detector sweep: v, P, m2 together
    float2 P[256];                  // x conj(s) products (ALIGN)                      } are exactly two more exchange buffers,
    float  m2[FX_NFFT];             //                                                 } cw another two (scr2)
    float2 hdr[FX_HDR_SYM];
    float2 cw[(WW + 1) * FX_HOP + 8];   // coarse scan: overlap half + one new hop per wave
    __device__ __forceinline__ float2 *scr2(int wave) { return wave < 2 ? v + 576 * wave : cw + 576 * (wave - 2); }
    float2 pb[16];                  // de-rotated pilots
    float2 eqw[16];                 // equaliser taps (equaliser stage on)
    float  taps[FX_MF_TAPS];
    float  redf[WW < 4 ? 4 : WW]; float2 redc[WW < 4 ? 4 : WW];
    float  redv[WW]; uint32_t redk[WW];
    float  f[16];                   // scalar broadcast slots
    uint32_t u[16];
    uint32_t cand[WW];
    uint8_t hs[FX_HDR_MOD];
    uint8_t b0[64], b1[64];
};

// what one detector hop needs, and nothing else: the seek verifier's workgroups (30 KB) leave LDS for the single-wave
// PLL workgroups of the blocks in flight -- with the walker's 53-KB layout three verifier workgroups fill a CU's LDS
// and PLL grids, the longest stage of a block, queue behind them.
template <int WW> struct SeekLdsT {
    static constexpr int WAVES = WW;
    static constexpr bool SWEEP_PAIRS = false;     // (four verifier workgroups share a CU: nothing to gain, 18 KB of LDS to lose)
    __device__ __forceinline__ float2 *scr2(int) { return nullptr; }
    float2 win[FX_NFFT], X[FX_NFFT], S[FX_NFFT];
    float2 scr[WW][576];
    float  redf[WW < 4 ? 4 : WW]; float2 redc[WW < 4 ? 4 : WW];
    float  redv[WW]; uint32_t redk[WW];
};

// ---- workgroup reductions (all 256 threads call; result on every thread) ----
template <class LDS> __device__ __forceinline__ float block_sum256(float v, LDS &L, int lane, int wave)
{
    v = wave_sum(v);                       // callers pass 0 from threads >= HALF; only waves 0-3 enter the tree
    __syncthreads();
    if (lane == 0) L.redf[wave] = v;
    __syncthreads();
    return (L.redf[0] + L.redf[1]) + (L.redf[2] + L.redf[3]);
}
template <class LDS> __device__ __forceinline__ float2 block_csum256(float2 v, LDS &L, int lane, int wave)
{
    v.x = wave_sum(v.x); v.y = wave_sum(v.y);
    __syncthreads();
    if (lane == 0) L.redc[wave] = v;
    __syncthreads();
    float2 a = cadd(L.redc[0], L.redc[1]), b = cadd(L.redc[2], L.redc[3]);
    return cadd(a, b);
}

__device__ __forceinline__ int64_t uniform64(int64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// A stream as a kernel sees it: the block's new samples at logical indices [0, n), the tail carried over from the
// previous block right-aligned below 0 (index p < 0 reads xa_end[p]).  Samples below `floor` (before the last
// synchroniser reset) and beyond the data read as zero; a floor is never below the start of the carried tail.
struct XSrc { const float2 *x, *xa_end; int64_t n; };
// the two pointers are the same for the whole workgroup: say so.  (Scalar registers instead of a vector pair each -- and, read
// straight out of a descriptor in memory where they sit side by side, the compiler turned xld()'s choice between them into an
// indexed load from a copy of the descriptor in SCRATCH: the 32 bytes of private memory every walker instance had.)
__device__ __forceinline__ XSrc make_xsrc(const float2 *x, const float2 *xa_end, int64_t n)
{
    XSrc s;
    s.x = reinterpret_cast<const float2 *>(uniform64((int64_t)reinterpret_cast<uintptr_t>(x)));
    s.xa_end = reinterpret_cast<const float2 *>(uniform64((int64_t)reinterpret_cast<uintptr_t>(xa_end)));
    s.n = n;
    return s;
}
__device__ __forceinline__ float2 xld(const XSrc &s, int64_t p) { const float2 *a = s.x, *b = s.xa_end; return (p < 0 ? b : a)[p]; }
__device__ __forceinline__ float2 xv(const XSrc &s, int64_t p, int64_t floor_)
{
    return (p >= floor_ && p < s.n && (p >= 0 || s.xa_end)) ? xld(s, p) : make_float2(0.0f, 0.0f);
}

// One hop of the exact detector (qdetector SEEK): L.win holds the 512-sample window (overlap half + new half,
// published by a barrier), x2_0 / x2_1 the energies of its halves.  Forward FFT, 49-bin CFO sweep, first maximum in
// (bin, lag) order.  Shared by the walker and by fx_seekverify_kernel so that both take bit-identical decisions.
// Called by the whole workgroup; result on every thread.
template <class LDS> __device__ __forceinline__ bool seek_sweep(LDS &L, float x2_0, float x2_1, float threshold, float s2sum, const float2 (&twA)[7],
                                           const float2 (&twB)[7], int lane, int wave, uint32_t &bidx, int &boff, float &peak)
{
    const float g0 = sqrtf(x2_0 + x2_1) * sqrtf((float)FX_S_LEN / (float)FX_NFFT);
    bidx = 0; boff = 0; peak = 0.0f;
    if (g0 < 1e-10f) return false;
    float2 a[8];
    if (wave == 0) {                                          // forward FFT of the window
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = L.win[lane + 64 * q];
        fft512_wave(a, L.scr[0], lane, twA, twB);
        const int kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
        for (int t = 0; t < 8; t++) L.X[kb + 64 * t] = a[t];
    }
    __syncthreads();
    // CFO sweep: offsets -24..24 dealt round-robin to the waves.  Only the maximum |R|^2 and the bin it belongs to are
    // tracked here; the lag of the winner is needed only when the peak clears the threshold, and is then read off one
    // more transform of the winning bin (same arithmetic, so the value is found again exactly).  Tie rule as before:
    // first maximum in (bin, lag) order.
    float bv = -1.0f; uint32_t bo = 0xFFFFFFFFu;
    if constexpr (LDS::SWEEP_PAIRS) {
        // two bins of this wave at a time (off, off + WAVES), their transforms side by side; the odd one out is done twice
        float2 *scrB = L.scr2(wave);
        for (int off = -FX_RANGE + wave; off <= FX_RANGE; off += 2 * LDS::WAVES) {
            const bool two = off + LDS::WAVES <= FX_RANGE;
            const int off2 = two ? off + LDS::WAVES : off;
            fx_v2 a[8], b[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = lane + 64 * q;
                const fx_v2 x = to_v2(L.X[i]);
                a[q] = pk_cmulc_swap(x, to_v2(L.S[(i - off) & (FX_NFFT - 1)]));                 // swap: inverse via forward FFT
                b[q] = pk_cmulc_swap(x, to_v2(L.S[(i - off2) & (FX_NFFT - 1)]));
            }
            fft512_wave2(a, b, L.scr[wave], scrB, lane, twA, twB);
            float mo = fmaf(a[0].y, a[0].y, a[0].x * a[0].x), mo2 = fmaf(b[0].y, b[0].y, b[0].x * b[0].x);
#pragma unroll
            for (int t = 1; t < 8; t++) { mo = fmaxf(mo, fmaf(a[t].y, a[t].y, a[t].x * a[t].x)); mo2 = fmaxf(mo2, fmaf(b[t].y, b[t].y, b[t].x * b[t].x)); }
            if (mo > bv) { bv = mo; bo = (uint32_t)(off + FX_RANGE); }       // bins ascend per wave: first maximum kept
            if (two && mo2 > bv) { bv = mo2; bo = (uint32_t)(off2 + FX_RANGE); }
            __builtin_amdgcn_wave_barrier();
        }
    } else
    for (int off = -FX_RANGE + wave; off <= FX_RANGE; off += LDS::WAVES) {
        fx_v2 a[8];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = lane + 64 * q;
            a[q] = pk_cmulc_swap(to_v2(L.X[i]), to_v2(L.S[(i - off) & (FX_NFFT - 1)]));        // swap: inverse via forward FFT
        }
        fft512_wave(a, L.scr[wave], lane, twA, twB);
        float mo = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);     // |R|^2, R = (a.y, a.x)
#pragma unroll
        for (int t = 1; t < 8; t++) mo = fmaxf(mo, fmaf(a[t].y, a[t].y, a[t].x * a[t].x));
        if (mo > bv) { bv = mo; bo = (uint32_t)(off + FX_RANGE); }          // bins ascend per wave: first maximum kept
    }
    wave_argmax(bv, bo);
    if (lane == 0) { L.redv[wave] = bv; L.redk[wave] = bo; }
    __syncthreads();
    bv = L.redv[0]; bo = L.redk[0];
#pragma unroll
    for (int w = 1; w < LDS::WAVES; w++) {
        float ov = L.redv[w]; uint32_t ok = L.redk[w];
        bool take = (ov > bv) || (ov == bv && ok < bo);
        bv = take ? ov : bv; bo = take ? ok : bo;
    }
    const float g = 1.0f / ((float)FX_NFFT * g0 * sqrtf(s2sum));
    peak = sqrtf(bv) * g;
    boff = (int)bo - FX_RANGE;
    if (!(peak > threshold)) return false;
    // the winner's lag: first lag of bin bo whose |R|^2 equals the maximum
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int i = lane + 64 * q;
            float2 y = cmulc(L.X[i], L.S[(i - boff) & (FX_NFFT - 1)]);
            a[q] = make_float2(y.y, y.x);
        }
        fft512_wave(a, L.scr[0], lane, twA, twB);
        uint32_t kmin = 0xFFFFFFFFu;
        const uint32_t kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
        for (int t = 7; t >= 0; t--) if (fmaf(a[t].y, a[t].y, a[t].x * a[t].x) == bv) kmin = kb + 64 * t;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m, 64));
        if (lane == 0) L.redk[0] = kmin;
    }
    __syncthreads();
    bidx = L.redk[0] & (FX_NFFT - 1);
    return (peak > threshold) && (bidx < FX_NFFT - FX_S_LEN);
}

// header: 54 received bytes (L.b0) -> 20 header bytes (L.b1[0..19]) + CRC verdict (L.u[1]).
// Called by the whole workgroup; every stage is spread over threads (a one-thread version of this cost more
// than the rest of the header span together).
template <class LDS> __device__ __forceinline__ void decode_header_bytes(LDS &L, const FxTables *T, int tid)
{
    uint8_t *b0 = L.b0, *b1 = L.b1;
    auto gather_byte = [&](const uint8_t *src, const uint16_t *perm, int j) -> uint8_t {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) { const unsigned s_ = perm[8 * j + b]; v = (v << 1) | ((src[s_ >> 3] >> (7 - (s_ & 7))) & 1u); }
        return (uint8_t)v;
    };
    if (tid < FX_HDR_ENC) b1[tid] = gather_byte(b0, T->perm54, tid);                       // de-interleave (54)
    __syncthreads();
    if (tid < FX_HDR_E0) b0[tid] = (uint8_t)((T->h84dec[b1[2 * tid]] << 4) | T->h84dec[b1[2 * tid + 1]]);   // Hamming(8,4)
    __syncthreads();
    if (tid < FX_HDR_E0) b1[tid] = gather_byte(b0, T->perm27, tid);                        // de-interleave (27)
    __syncthreads();
    if (tid < 192) {                                                                       // SECDED(72,64): 3 blocks x 64 bits
        const int blk = tid >> 6, j = tid & 63;
        const uint8_t *e = b1 + 9 * blk;
        const unsigned bit = (e[1 + (j >> 3)] >> (7 - (j & 7))) & 1u;
        unsigned par = bit ? T->sdcol[j] : 0u;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) par ^= (unsigned)__shfl_xor((int)par, m, 64);
        const unsigned syn = (e[0] ^ par) & 0xffu;
        // single data-bit error: the (unique) lane whose column equals the syndrome flips its bit
        const bool fix = syn != 0 && __popc(syn) != 1 && T->sdcol[j] == syn;
        const unsigned long long word = __ballot(bit ^ (fix ? 1u : 0u));                   // corrected 64 data bits, bit j = lane j
        if (j < 8) {
            unsigned v = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) v = (v << 1) | (unsigned)((word >> (8 * j + b)) & 1ull);
            b0[8 * blk + j] = (uint8_t)v;
        }
    }
    __syncthreads();
    if (tid < FX_HDR_CRC) { const uint8_t mask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 }; b0[tid] ^= mask[tid & 3]; }   // de-whiten
    __syncthreads();
    if (tid < FX_HDR_DEC) b1[tid] = b0[tid];
    if (tid == 0) {
        uint32_t key = 0xFFFFFFFFu;
        for (int j = 0; j < FX_HDR_DEC; j++) {
            key ^= b0[j];
#pragma unroll
            for (int b = 0; b < 8; b++) key = (key >> 1) ^ (0xEDB88320u & (0u - (key & 1u)));
        }
        key = ~key;
        const uint32_t rx = ((uint32_t)b0[20] << 24) | ((uint32_t)b0[21] << 16) | ((uint32_t)b0[22] << 8) | b0[23];
        L.u[1] = key == rx;
    }
    __syncthreads();
}

// ===================================================================== detector-only walker (frame_detector_cc): a wave per hop
// In detector mode every hop costs the full sweep (1 forward + 49 inverse FFT-512 + the winner's lag), and a hop is a pure
// function of (pos, floor).  Between two detections the hop grid is known in advance, so a workgroup takes as many hops as it has
// waves at a time and every wave runs ONE WHOLE HOP by itself: its own forward transform (no wave idles meanwhile), the window's
// spectrum kept in registers in natural order for all 49 bins (no LDS round trip for it), all 49 bins (no 49 = 6 x 8 + 1 tail),
// no workgroup barrier between the window load and the verdicts.  The first hop of the round that fires is aligned by the whole
// workgroup exactly as before; the hops behind it are discarded (the grid restarts at the detection).  Same arithmetic per
// transform, same tie rules (first maximum in (bin, lag) order) as seek_sweep(): bit-identical decisions.
template <int WW> struct DetLdsT {
    static constexpr int WAVES = WW;
    float2 S[FX_NFFT];                  // template spectrum
    float2 cw[(WW + 1) * FX_HOP];       // samples [pos - 256, pos + 256 WW); ALIGN: win = cw[0, 512), X = cw[512, 1024), P = cw[1024, 1280)
    float2 scr[WW][576];                // per-wave FFT exchange buffers (ALIGN: m2 borrows scr[1])
    float2 redc[4];
    float  f[16]; uint32_t u[16];
    float  pk[WW]; uint32_t bi[WW]; int bo[WW]; uint32_t dt[WW];
};
static_assert(FX_DETECT_WAVES >= 4, "ALIGN borrows cw[0, 1280) and scr[1]");

// the canonical half-window energy (block_sum256's order) by one wave: elements 64 q + lane, q = 0..3
__device__ __forceinline__ float half_energy_wave(const float2 *h, int lane)
{
    const float s0 = wave_sum(cm2(h[lane])), s1 = wave_sum(cm2(h[64 + lane])), s2 = wave_sum(cm2(h[128 + lane])), s3 = wave_sum(cm2(h[192 + lane]));
    return (s0 + s1) + (s2 + s3);
}

// one detector hop by ONE wave: w = its 512-sample window in LDS, scr = its exchange buffer (576 float2)
__device__ __forceinline__ bool seek_sweep_wave(const float2 *w, const float2 *S, float2 *scr, float threshold, float s2sum, const float2 (&twA)[7],
                                                const float2 (&twB)[7], int lane, uint32_t &bidx, int &boff, float &peak)
{
    const float x2_0 = half_energy_wave(w, lane), x2_1 = half_energy_wave(w + FX_HOP, lane);
    const float g0 = sqrtf(x2_0 + x2_1) * sqrtf((float)FX_S_LEN / (float)FX_NFFT);
    bidx = 0; boff = 0; peak = 0.0f;
    if (g0 < 1e-10f) return false;
    float2 a[8], xn[8];
#pragma unroll
    for (int q = 0; q < 8; q++) a[q] = w[lane + 64 * q];
    fft512_wave(a, scr, lane, twA, twB);
    {   // spectrum into natural order, into registers: lane j keeps X[j + 64 q]
        const int kb = (lane >> 3) + 8 * (lane & 7);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 8; t++) scr[kb + 64 * t] = a[t];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 8; q++) xn[q] = scr[lane + 64 * q];
        __builtin_amdgcn_wave_barrier();
    }
    float bv = -1.0f; uint32_t bo = 0xFFFFFFFFu;
    fx_v2 xv2[8];
#pragma unroll
    for (int q = 0; q < 8; q++) xv2[q] = to_v2(xn[q]);
#pragma unroll 1
    for (int off = -FX_RANGE; off <= FX_RANGE; off++) {
        fx_v2 a[8];
#pragma unroll
        for (int q = 0; q < 8; q++) a[q] = pk_cmulc_swap(xv2[q], to_v2(S[(lane + 64 * q - off) & (FX_NFFT - 1)]));    // swap: inverse via forward FFT
        fft512_wave(a, scr, lane, twA, twB);
        float mo = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);     // |R|^2, R = (a.y, a.x)
#pragma unroll
        for (int t = 1; t < 8; t++) mo = fmaxf(mo, fmaf(a[t].y, a[t].y, a[t].x * a[t].x));
        if (mo > bv) { bv = mo; bo = (uint32_t)(off + FX_RANGE); }          // bins ascend: first maximum kept
        __builtin_amdgcn_wave_barrier();
    }
    wave_argmax(bv, bo);
    const float g = 1.0f / ((float)FX_NFFT * g0 * sqrtf(s2sum));
    peak = sqrtf(bv) * g;
    boff = (int)bo - FX_RANGE;
    if (!(peak > threshold)) return false;
    // the winner's lag: first lag of bin bo whose |R|^2 equals the maximum
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float2 y = cmulc(xn[q], S[(lane + 64 * q - boff) & (FX_NFFT - 1)]);
        a[q] = make_float2(y.y, y.x);
    }
    fft512_wave(a, scr, lane, twA, twB);
    uint32_t kmin = 0xFFFFFFFFu;
    const uint32_t kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
    for (int t = 7; t >= 0; t--) if (fmaf(a[t].y, a[t].y, a[t].x * a[t].x) == bv) kmin = kb + 64 * t;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, m, 64));
    bidx = kmin & (FX_NFFT - 1);
    return bidx < FX_NFFT - FX_S_LEN;
}

template <int WW, bool EXT>
__device__ __forceinline__ void detect_run(const FxWalkJob &job, uint32_t job_index, FxWalkResult *result, FxFrame *frames, FxBlockHdr *hdr, const FxTables *T,
                                           DetLdsT<WW> &L, const float2 (&twA)[7], const float2 (&twB)[7],
                                           const FxWalkJob *all_jobs, const FxWalkResult *all_results, uint32_t n_jobs_total)
{
    constexpr int NT = 64 * WW;
    const int tid_in = threadIdx.x;
    const int tid = tid_in;
#ifdef FX_STAMPS
    const uint32_t wall0_ = (uint32_t)wall_clock64();           // job timeline (tools/dev/dev_walk_timeline.py): start, end (100 MHz), CU
#endif
    const XSrc xs = make_xsrc(job.x, job.xa_end, job.n);
    const int64_t n = job.n;
    int64_t pos = job.start, floor_ = job.floor, stop = job.stop;
    int ext_left = 6;
    bool fresh = job.fresh != 0, in_handoff = false, locked = job.prelock == 0;
    uint32_t nfr = 0, hops = 0, exit_code = FX_EXIT_STOP, has_handoff = 0;
    int64_t ho_start = 0, ho_pos = 0; int32_t ho_off = 0; float ho_rxy = 0.0f; uint32_t ho_clear = 0;
    if (job.state_in) {
        const FxStreamState st = *job.state_in;
        if (st.invalid) {
            if (tid == 0) {
                FxWalkResult r; r.n_frames = 0; r.exit_code = FX_EXIT_INVALID; r.pos = 0; r.floor = 0; r.fresh = 1; r.has_handoff = 0;
                r.handoff_start = 0; r.handoff_offset = 0; r.hops = 0; r.handoff_rxy = 0.0f; r.hops_cheap = 0; r.tail_pos = 0; r.tail_floor = 0;
                r.handoff_pos = 0; r.handoff_clear = 0; r.tail_flags = FX_FLAG_SPAN_EXACT;
                for (int i = 0; i < 4; i++) r.stamp[i] = 0;
                *result = r;
            }
            return;
        }
        pos = st.pos; floor_ = st.floor; fresh = st.fresh != 0;
    }
    int64_t span_pos = pos, span_floor = floor_;
    uint32_t span_flags = (fresh ? FX_FLAG_SEEK_FRESH : 0u) | FX_FLAG_SPAN_EXACT;
    const float s2sum = T->s2sum;
    const float2 *sc = T->sc;
    float2 *win = L.cw, *X = L.cw + FX_NFFT, *P = L.cw + 2 * FX_NFFT; float *m2 = reinterpret_cast<float *>(L.scr[1]);
    // Re-sync (walks of a repair round only): this segment was walked before, speculatively, and is walked again because the true
    // chain enters it in a state its list did not anticipate.  From the first detection the two walks have in common on, the old
    // list IS what the sequential machine produces (same rule as a splice: a locked detection at the same start and CFO bin, no
    // zero-floor in the way), so the new walk only has to bridge the gap: it writes its frames into the upper half of the segment's
    // table, looks every detection up in the old list below, and on the first hit appends the old list's remainder, moves the
    // whole to the front and takes over the old walk's result.  A segment re-walk costs a frame or two instead of the whole segment.
    uint32_t fbase = 0, n_old = 0; FxWalkResult old_res;
    bool resync = false;
    if constexpr (EXT) {
        if (job.pad_ && all_results) {
            old_res = all_results[job_index];
            n_old = old_res.n_frames;
            if (n_old > 0 && n_old <= job.max_frames / 2 && old_res.exit_code != FX_EXIT_INVALID) { resync = true; fbase = job.max_frames / 2; }
        }
    }
    const uint32_t fcap = resync ? job.max_frames - fbase : job.max_frames;

    for (;;) {
        // (thread index opaque per round: lane-dependent indices are computed where they are used instead of being kept, see walk_run)
        int tid = tid_in; asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool lo = tid < HALF;
        pos = uniform64(pos); floor_ = uniform64(floor_);
        nfr = __builtin_amdgcn_readfirstlane(nfr); hops = __builtin_amdgcn_readfirstlane(hops);
        if (pos >= stop && !in_handoff) {
            if (job.handoff && locked) in_handoff = true; else { exit_code = FX_EXIT_STOP; break; }
        }
        if (pos + FX_HOP > n) { exit_code = FX_EXIT_NEED_DATA; break; }
        // hops of this round: on the grid pos + 256 h, inside the data, and -- unless this is the hand-off seek -- before the segment's end
        int nh = (int)min((int64_t)WW, (n - pos) / FX_HOP);
        if (!in_handoff) nh = (int)min((int64_t)nh, (stop - pos + FX_HOP - 1) / FX_HOP);
        const int64_t fl0 = fresh ? max(floor_, pos) : floor_;        // (a freshly reset detector's overlap half is zeros)
        __syncthreads();
        {
            constexpr int NLD = ((WW + 1) * FX_HOP + NT - 1) / NT;
            float2 ld[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int i = tid + NT * k; ld[k] = i < (nh + 1) * FX_HOP ? xv(xs, pos - FX_HOP + i, i < FX_HOP ? fl0 : floor_) : make_float2(0.0f, 0.0f); }
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int i = tid + NT * k; if (i < (nh + 1) * FX_HOP) L.cw[i] = ld[k]; }
        }
        __syncthreads();
        if (wave < nh) {
            uint32_t bidx; int boff; float peak;
            const bool d = seek_sweep_wave(L.cw + FX_HOP * wave, L.S, L.scr[wave], job.threshold, s2sum, twA, twB, lane, bidx, boff, peak);
            if (lane == 0) { L.dt[wave] = d ? 1u : 0u; L.bi[wave] = bidx; L.bo[wave] = boff; L.pk[wave] = peak; }
        }
        __syncthreads();
        bool moved = false, leave = false;
        for (int h = 0; h < nh; h++) {
            hops++;
            if (!L.dt[h]) continue;
            const int64_t hp = pos + (int64_t)FX_HOP * h;              // this hop's position; what came before it in the round saw nothing
            const uint32_t bidx = L.bi[h]; const int boff = L.bo[h]; const float peak = L.pk[h];
            const int64_t a0 = hp - FX_HOP + (int64_t)bidx;
            if (in_handoff) {
                bool extend = false;
                if constexpr (EXT) {
                    if (ext_left > 0 && all_jobs) {
                        uint32_t k = job_index + 1;                            // the segment the detection falls in (as fx_chain_kernel picks it)
                        while (k + 1 < n_jobs_total && all_jobs[k + 1].stream == job.stream && a0 >= all_jobs[k].stop + FX_HOP) k++;
                        if (k < n_jobs_total && all_jobs[k].stream == job.stream) {
                            const FxFrame *FN = frames + all_jobs[k].frame_base;
                            const uint32_t nfn = all_results[k].n_frames;
                            int hit = 0;
                            if (floor_ <= a0)
                                for (uint32_t i = tid; i < nfn; i += NT) {
                                    const uint32_t fl = FN[i].flags;
                                    if ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_FLOOR_CLEAR) && FN[i].start == a0 && FN[i].offset == boff) hit = 1;
                                }
                            hit = __syncthreads_or(hit);
                            const int64_t room = (int64_t)job.max_frames - (int64_t)nfr - 8;
                            if (!hit && room * FX_HOP > all_jobs[k].stop - a0) { extend = true; stop = all_jobs[k].stop; ext_left--; in_handoff = false; }
                        }
                    }
                }
                if (!extend) { has_handoff = 1; ho_start = a0; ho_off = boff; ho_rxy = peak; ho_pos = hp; ho_clear = floor_ <= a0 ? 1u : 0u; exit_code = FX_EXIT_STOP; pos = hp; fresh = fresh && h == 0; leave = true; break; }
            }
            if constexpr (EXT) {
                if (resync && !in_handoff && floor_ <= a0) {
                    const FxFrame *FO = frames + job.frame_base;
                    __syncthreads();
                    if (tid == 0) L.u[15] = 0xFFFFFFFFu;
                    __syncthreads();
                    for (uint32_t i = tid; i < n_old; i += NT) {
                        const uint32_t fl = FO[i].flags;
                        if ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_FLOOR_CLEAR) && FO[i].start == a0 && FO[i].offset == boff) atomicMin(&L.u[15], i);
                    }
                    __syncthreads();
                    const uint32_t hit = L.u[15];
                    if (hit != 0xFFFFFFFFu) {
                        const uint32_t q = hit, n_tail = n_old - q;
                        if (nfr + n_tail <= fcap) {
                            FxFrame *FW = frames + job.frame_base;
                            // old[q, n_old) behind the new frames, then everything to the front (the two ranges may overlap: staged moves, a frame per thread and pass)
                            for (uint32_t i0 = 0; i0 < n_tail; i0 += NT) {
                                const uint32_t i = i0 + tid; FxFrame t;
                                if (i < n_tail) t = FW[q + i];
                                __syncthreads();
                                if (i < n_tail) FW[fbase + nfr + i] = t;
                                __syncthreads();
                            }
                            __threadfence(); __syncthreads();
                            const uint32_t n_all = nfr + n_tail;
                            for (uint32_t i0 = 0; i0 < n_all; i0 += NT) {
                                const uint32_t i = i0 + tid; FxFrame t;
                                if (i < n_all) t = FW[fbase + i];
                                __syncthreads();
                                if (i < n_all) {
                                    if (i == nfr) {                     // the frame both walks found: its seek is this walk's (exact), its coarse peak the true chain's
                                        t.rxy = peak; t.seek_pos = span_pos; t.seek_floor = span_floor; t.det_pos = hp;
                                        t.flags = (t.flags & ~(uint32_t)(FX_FLAG_SPAN_BAD | FX_FLAG_SEEK_FRESH)) | span_flags;
                                    }
                                    FW[i] = t;
                                }
                                __syncthreads();
                            }
                            if (tid == 0) {
                                FxWalkResult r = old_res;
                                r.n_frames = n_all; r.hops = hops;
                                *result = r;
                                atomicAdd(&hdr->hops, hops); atomicAdd(&hdr->walk_jobs_run, 1u);
                            }
                            return;
                        }
                    }
                }
            }
            if (nfr >= fcap) { exit_code = FX_EXIT_TABLE_FULL; pos = hp; fresh = fresh && h == 0; leave = true; break; }
            if (a0 + FX_NFFT > n) { exit_code = FX_EXIT_NEED_DATA; pos = hp; fresh = fresh && h == 0; leave = true; break; }
            // Speculative walker not yet locked: a weak peak may be a false alarm the sequential chain never sees (its hop grid
            // differs).  Ignore it and keep the grid; lock on a strong one.
            if (!locked && !(peak > 0.7f)) continue;

            // ------------------------------------------------------------ ALIGN on x[a0, a0+512): the whole workgroup
            __syncthreads();
            for (int i = tid; i < FX_NFFT; i += NT) win[i] = xv(xs, a0 + i, floor_);
            __syncthreads();
            if (wave == 0) {
                float2 a[8];
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = win[lane + 64 * q];
                fft512_wave(a, L.scr[0], lane, twA, twB);
                const int kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
                for (int t = 0; t < 8; t++) X[kb + 64 * t] = a[t];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int i = lane + 64 * q;
                    float2 y = cmulc(X[i], L.S[(i - boff) & (FX_NFFT - 1)]);
                    a[q] = make_float2(y.y, y.x);
                }
                fft512_wave(a, L.scr[0], lane, twA, twB);
                if (lane == 0)  L.f[0] = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);      // lags 0, +1, -1
                if (lane == 8)  L.f[1] = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);
                if (lane == 63) L.f[2] = fmaf(a[7].y, a[7].y, a[7].x * a[7].x);
            }
            __syncthreads();
            float tau, gamma;
            {
                float y0 = sqrtf(sqrtf(L.f[0])), ypos = sqrtf(sqrtf(L.f[1])), yneg = sqrtf(sqrtf(L.f[2]));
                float qa = 0.5f * (ypos + yneg) - y0, qb = 0.5f * (ypos - yneg);
                tau = qa == 0.0f ? 0.0f : -qb / (2.0f * qa);
                if (!(fabsf(tau) < 1.0f)) tau = 0.0f;
                float gh = fmaf(fmaf(qa, tau, qb), tau, y0);
                gamma = gh * gh / ((float)FX_NFFT * s2sum);
            }
            {
                float2 p = make_float2(0.0f, 0.0f);
                if (tid < FX_S_LEN) p = cmulc(win[tid], T->s[tid]);
                if (lo) P[tid] = p;
            }
            __syncthreads();
            if (wave == 0) {
                float2 a[8];
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = (q < 4) ? P[lane + 64 * q] : make_float2(0.0f, 0.0f);
                fft512_wave(a, L.scr[0], lane, twA, twB);
                const uint32_t kb = (lane >> 3) + 8 * (lane & 7);
                float bv = -1.0f; uint32_t bk = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    float m = cm2(a[t]);
                    m2[kb + 64 * t] = m;
                    if (m > bv) { bv = m; bk = kb + 64 * t; }
                }
                wave_argmax(bv, bk);
                if (lane == 0) { L.f[3] = bv; L.u[0] = bk; }
            }
            __syncthreads();
            float dphi;
            {
                const uint32_t i0 = L.u[0];
                float v0 = sqrtf(L.f[3]);
                float vneg = sqrtf(m2[(i0 + FX_NFFT - 1) & (FX_NFFT - 1)]);
                float vpos = sqrtf(m2[(i0 + 1) & (FX_NFFT - 1)]);
                float qa = 0.5f * (vpos + vneg) - v0, qb = 0.5f * (vpos - vneg);
                float idx = qa == 0.0f ? 0.0f : -qb / (2.0f * qa);
                float index = (float)i0 + idx;
                dphi = (i0 > FX_NFFT / 2 ? index - (float)FX_NFFT : index) * (6.28318531f / (float)FX_NFFT);
            }
            const uint32_t mix_dl = rad2u32(dphi);
            float phi;
            {
                float2 term = make_float2(0.0f, 0.0f);
                if (tid < FX_S_LEN) term = derot(P[tid], mix_dl * (uint32_t)tid, sc);
                float2 metric = block_csum256(term, L, lane, wave);
                phi = atan2c(metric.y, metric.x);
            }
            FxFrameHead fr;
            fr.start = a0; fr.offset = boff; fr.rxy = peak; fr.tau = tau; fr.gamma = gamma; fr.dphi = dphi; fr.phi = phi;
            fr.seek_pos = span_pos; fr.seek_floor = span_floor; fr.det_pos = hp;
            fr.pfb = 0; fr.mfc0 = 0; fr.mix_th = rad2u32(phi); fr.mix_dl = mix_dl; fr.mf_scale = 0.0f;
            fr.pilot_dphi = fr.pilot_phi = fr.pilot_gain = 0.0f; fr.pll_th = 0; fr.pll_f = 0.0f;
            fr.flags = (floor_ <= a0 ? FX_FLAG_FLOOR_CLEAR : 0u) | span_flags | FX_FLAG_EXACT;
            fr.pay_len = fr.ms = fr.check = fr.fec0 = fr.fec1 = fr.pay_sym_len = 0;
#pragma unroll
            for (int j = 0; j < FX_HDR_DEC; j++) fr.header[j] = 0;
            locked = true;
            fr.next = a0 + FX_NFFT;                                       // back to SEEK with the second half of the aligned window as overlap
            if (tid == 0) static_cast<FxFrameHead &>(frames[job.frame_base + fbase + nfr]) = fr;
            nfr++;
            span_pos = a0 + FX_NFFT; span_floor = floor_; span_flags = FX_FLAG_SPAN_EXACT;
            pos = a0 + FX_NFFT; fresh = false; moved = true;
            break;
        }
        if (leave) break;
        if (!moved) { pos += (int64_t)FX_HOP * nh; fresh = false; }
    }

    if (fbase) {                                                        // a re-walk that never met its old list: its frames to the front
        __syncthreads();
        FxFrame *FW = frames + job.frame_base;
        for (uint32_t i0 = 0; i0 < nfr; i0 += NT) {
            const uint32_t i = i0 + tid; FxFrame t;
            if (i < nfr) t = FW[fbase + i];
            __syncthreads();
            if (i < nfr) FW[i] = t;
            __syncthreads();
        }
    }
    if (tid == 0) {
        FxWalkResult r;
        r.n_frames = nfr; r.exit_code = exit_code; r.pos = pos; r.floor = floor_; r.fresh = fresh ? 1u : 0u;
        r.has_handoff = has_handoff; r.handoff_start = ho_start; r.handoff_offset = ho_off; r.hops = hops;
        r.handoff_rxy = ho_rxy; r.hops_cheap = 0;
        r.tail_pos = span_pos; r.tail_floor = span_floor; r.handoff_pos = ho_pos; r.handoff_clear = ho_clear;
        r.tail_flags = span_flags;
        for (int i = 0; i < 4; i++) r.stamp[i] = 0;
#ifdef FX_STAMPS
        r.stamp[0] = wall0_; r.stamp[1] = (uint32_t)wall_clock64(); r.stamp[2] = __smid();
#endif
        *result = r;
        atomicAdd(&hdr->hops, hops); atomicAdd(&hdr->walk_jobs_run, 1u);
    }
}

// MODE is a template parameter so that the detector-only instance (frame_detector_cc) carries none of the
// header-recovery code or its registers.
#ifndef FX_DETECT_OCC
#define FX_DETECT_OCC 4      // waves per SIMD the detector-only instance (and the seek verifier) is compiled for: 128 VGPRs, 4 x 39 KB of LDS per CU
#endif
// register budget of the walker instances ("amdgpu-num-vgpr" counts half of gfx950's unified file: 72 -> 144 VGPRs; the flex_rx
// instance the bench runs needs no private memory at that, its equaliser / repair-round variants spill a little)
#ifndef FX_FLEX_VGPRS
#define FX_FLEX_VGPRS 72
#endif
#define FX_WALK_VGPR_ATTR __attribute__((amdgpu_num_vgpr(FX_FLEX_VGPRS)))
#ifndef FX_VERIFY_OCC
#define FX_VERIFY_OCC 5      // the seek verifier: 96 VGPRs, 5 x 31 KB of LDS per CU
#endif
#ifndef FX_FLEX_OCC
#define FX_FLEX_OCC 2        // same for the flex_rx instance
#endif
// One walk: the synchroniser's state machine from (start, floor, fresh) of `job` until the job's stop / hand-off / end of
// data.  Called by the whole workgroup (fx_walk_kernel: once; fx_chain_kernel: for every repair).  L.S (template spectrum)
// must be loaded; the result record and the frames go to global memory (thread 0), verification runs to `runs`.
// EXT (walks of the repair rounds, fx_host.cpp): a hand-off target that the list of the segment it falls in does not hold
// would only be the next round's repair -- the walker looks it up itself and, if it is not there, carries on through that
// segment as well (a few times at most, and while its frame table has room).
template <int MODE, int WW, bool EQ, bool EXT = false>
__device__ __forceinline__ void walk_run(const FxWalkJob &job, uint32_t job_index, FxWalkResult *result, FxFrame *frames, FxVerifyRun *runs,
                                         uint32_t run_cap, FxBlockHdr *hdr, const FxTables *T_in, WalkLdsT<WW> &L,
                                         const float2 (&twA)[7], const float2 (&twB)[7],
                                         const FxWalkJob *all_jobs = nullptr, const FxWalkResult *all_results = nullptr, uint32_t n_jobs_total = 0)
{
    constexpr int WALK_WAVES = WW, WALK_THREADS = 64 * WW;
    const FxTables *T = T_in;
    const int tid_in = threadIdx.x;
    const int tid = tid_in, lane = tid & 63, wave = tid >> 6;
    const XSrc xs = make_xsrc(job.x, job.xa_end, job.n);
    const int64_t n = job.n;

    int64_t pos = job.start, floor_ = job.floor, stop = job.stop;
    int ext_left = 6;
    bool fresh = job.fresh != 0, in_handoff = false, locked = job.prelock == 0;
    uint32_t nfr = 0, hops = 0, hops_cheap = 0, exact_left = 0, exit_code = FX_EXIT_STOP, has_handoff = 0;
    int64_t ho_start = 0, ho_pos = 0; int32_t ho_off = 0; float ho_rxy = 0.0f; uint32_t ho_clear = 0;
    float x2_0 = 0.0f;
    if (job.state_in) {
        // true walker of a continuing stream: the previous block's chain kernel left the resume state on the device
        const FxStreamState st = *job.state_in;
        if (st.invalid) {
            if (tid == 0) {
                FxWalkResult r; r.n_frames = 0; r.exit_code = FX_EXIT_INVALID; r.pos = 0; r.floor = 0; r.fresh = 1; r.has_handoff = 0;
                r.handoff_start = 0; r.handoff_offset = 0; r.hops = 0; r.handoff_rxy = 0.0f; r.hops_cheap = 0; r.tail_pos = 0; r.tail_floor = 0;
                r.handoff_pos = 0; r.handoff_clear = 0; r.tail_flags = FX_FLAG_SPAN_EXACT;
                for (int i = 0; i < 4; i++) r.stamp[i] = 0;
                *result = r;
            }
            return;
        }
        pos = st.pos; floor_ = st.floor; fresh = st.fresh != 0;
    }
    // A locked flex_rx walker may skip hops its coarse scan finds empty (job.no_skip == 0): it stays on the true hop
    // grid and runs the exact detector only around coarse-scan candidates; every hop between span_pos and the next
    // detection is re-checked by fx_seekverify_kernel (the walker emits the runs itself when the span closes).
    const bool may_skip = MODE == FX_MODE_FLEXRX && job.no_skip == 0;
    int64_t span_pos = pos, span_floor = floor_;
    // the span is the chain's business only if the walker was locked when the seek began
    uint32_t span_flags = (fresh ? FX_FLAG_SEEK_FRESH : 0u) | ((!may_skip || !locked) ? FX_FLAG_SPAN_EXACT : 0u);
    auto emit_runs = [&](int64_t end, uint32_t owner) {          // thread 0: hops [span_pos, end) to the verifier
        if (span_flags & FX_FLAG_SPAN_EXACT) return false;
        const int64_t nh = (end - span_pos) / FX_HOP;
        if (nh <= 0) return false;
        const uint32_t per = job.verify_per ? job.verify_per : 4u;
        const uint32_t nr = (uint32_t)((nh + per - 1) / per);
        const uint32_t base = atomicAdd(&hdr->n_runs, nr);
        atomicAdd(&hdr->verify_hops, (uint32_t)nh);
        if ((uint64_t)base + nr > run_cap) return true;          // no room: the caller marks the span bad (walked again, exactly)
        for (uint32_t r = 0; r < nr; r++) {
            FxVerifyRun v; v.pos = span_pos + (int64_t)r * per * FX_HOP; v.floor = span_floor; v.job = job_index; v.owner = owner;
            v.nhops = (uint32_t)min((int64_t)per, nh - (int64_t)r * per); v.pad_ = 0;
            runs[base + r] = v;
        }
        return false;
    };
#ifdef FX_STAMPS
    unsigned long long wt_ = __builtin_readcyclecounter(); uint32_t wst_[4] = { 0, 0, 0, 0 };
#define WSTAMP(i) do { unsigned long long t2_ = __builtin_readcyclecounter(); wst_[i] += (uint32_t)(t2_ - wt_); wt_ = t2_; } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
    const float s2sum = T->s2sum;
    const float2 *sc = T->sc;

    const bool lo = tid < HALF;            // threads that own one sample of a 256-sample half
    __syncthreads();
    if (fresh) { if (lo) L.win[tid] = make_float2(0.0f, 0.0f); }
    else { float2 w = lo ? xv(xs, pos - FX_HOP + tid, floor_) : make_float2(0.0f, 0.0f); if (lo) L.win[tid] = w; x2_0 = block_sum256(cm2(w), L, lane, wave); }
    __syncthreads();

    for (;;) {
        // The table pointer is made opaque once per turn of the state machine: left alone, the compiler computes every lane-dependent
        // table address of every phase (some forty 64-bit pointers: T->TD[k + 64 t], the gather tables, ...) once, in front of this
        // loop, and keeps them in vector registers through all phases -- the detector sweep below was left with so few registers that
        // it read the template spectrum one element at a time, a full LDS round trip each.
        // Likewise the thread index: every lane-dependent index and byte offset any phase uses (lane | 64 q, 8 (lane + 64 q), ...:
        // fifty registers) would be computed once and kept; they cost an instruction each where they are used.
        const FxTables *T = T_in; asm volatile("" : "+s"(T));
        int tid = tid_in; asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool lo = tid < HALF;
        // the walk state is uniform across the workgroup: say so, so that it lives in scalar registers
        pos = uniform64(pos); floor_ = uniform64(floor_); span_pos = uniform64(span_pos); span_floor = uniform64(span_floor);
        x2_0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x2_0)));
        nfr = __builtin_amdgcn_readfirstlane(nfr); hops = __builtin_amdgcn_readfirstlane(hops); hops_cheap = __builtin_amdgcn_readfirstlane(hops_cheap);
        exact_left = __builtin_amdgcn_readfirstlane(exact_left);
        if (pos >= stop && !in_handoff) {
            // a speculative walker that never locked has nothing to hand off (its state is not the chain's)
            if (job.handoff && locked) in_handoff = true; else { exit_code = FX_EXIT_STOP; break; }
        }
        if (pos + FX_HOP > n) { exit_code = FX_EXIT_NEED_DATA; break; }

        // ------------------------------------------------------------ pre-lock coarse scan, four hops at a time
        // (same differential correlator as the single-hop form below, one window per wave, no block barriers
        // inside; used while at least four hops remain before the segment end / end of data)
        if ((!locked || may_skip) && MODE == FX_MODE_FLEXRX && exact_left == 0 && pos + WALK_WAVES * FX_HOP <= n &&
            (in_handoff || pos + (WALK_WAVES - 1) * FX_HOP < stop)) {
            __syncthreads();
            {   // (all of a thread's loads in flight at once: a rolled load / store loop pays the memory latency once per pass)
                constexpr int NLD = ((WALK_WAVES + 1) * FX_HOP + WALK_THREADS - 1) / WALK_THREADS;
                float2 ld[NLD];
#pragma unroll
                for (int k = 0; k < NLD; k++) { const int i = tid + WALK_THREADS * k; ld[k] = i < (WALK_WAVES + 1) * FX_HOP ? xv(xs, pos - FX_HOP + i, floor_) : make_float2(0.0f, 0.0f); }
#pragma unroll
                for (int k = 0; k < NLD; k++) { const int i = tid + WALK_THREADS * k; if (i < (WALK_WAVES + 1) * FX_HOP) L.cw[i] = ld[k]; }
            }
            __syncthreads();
            hops_cheap += WALK_WAVES;
            {
                const float2 *w = L.cw + FX_HOP * wave;               // this wave's 512-sample window
                float2 a[8]; float e = 0.0f; float2 sm = make_float2(0.0f, 0.0f);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int i = lane + 64 * q;
                    a[q] = (i < FX_NFFT - 1) ? cmulc(w[i + 1], w[i]) : make_float2(0.0f, 0.0f);
                    e += cm2(a[q]); sm = cadd(sm, a[q]);
                }
                e = wave_sum(e); sm.x = wave_sum(sm.x); sm.y = wave_sum(sm.y);
                e -= cm2(sm) * (1.0f / (float)FX_NFFT);
                fft512_wave(a, L.scr[wave], lane, twA, twB);
                const int kb = (lane >> 3) + 8 * (lane & 7);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 8; t++) { float2 y = cmulc(a[t], T->TD[kb + 64 * t]); L.scr[wave][kb + 64 * t] = make_float2(y.y, y.x); }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = L.scr[wave][lane + 64 * q];
                __builtin_amdgcn_wave_barrier();
                fft512_wave(a, L.scr[wave], lane, twA, twB);
                float bv = -1.0f; uint32_t bk = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t k = (uint32_t)kb + 64 * t;
                    const float m = cm2(a[t]);
                    if (k < FX_NFFT - FX_S_LEN && m > bv) { bv = m; bk = k; }
                }
                wave_argmax(bv, bk);
                const bool hit = e > 0.0f && bv > 0.06f * e * T->td2sum * (float)FX_NFFT * (float)FX_NFFT;
                if (lane == 0) L.cand[wave] = hit ? bk : 0xFFFFFFFFu;
            }
            __syncthreads();
            int hw = -1;
#pragma unroll
            for (int w = WALK_WAVES - 1; w >= 0; w--) if (L.cand[w] != 0xFFFFFFFFu) hw = w;
            if (hw >= 0 && locked) {
                // candidate in the window of grid hop hw: resume the exact detector there (one hop earlier when the
                // candidate sits at the very start of the window -- the two correlators may disagree by a sample)
#ifdef FX_EXACT_HOPS_OLD
                exact_left = 3;
                if (L.cand[hw] < 8u && hw > 0) { hw--; exact_left = 4; }
#else
                // The exact detector accepts a peak at lags below 512 - 156 only, like this one: a candidate well inside that range is
                // this very hop's detection or nobody's (a false alarm of the cheap correlator costs ONE sweep of 50 transforms, not
                // three); near either end of the range the two correlators may place it in the neighbouring hop.  Whatever this misjudges,
                // the seek verifier finds (every hop of the span is checked there anyway) and the chain kernel repairs.
                const uint32_t cl_ = L.cand[hw];
                exact_left = cl_ >= FX_NFFT - FX_S_LEN - 12u ? 2 : 1;
                if (cl_ < 8u && hw > 0) { hw--; exact_left = 2; }
#endif
                hops_cheap -= WALK_WAVES - hw;
                if (hw > 0) {
                    const float2 w = lo ? L.cw[FX_HOP * hw + tid] : make_float2(0.0f, 0.0f);
                    if (lo) L.win[tid] = w;
                    x2_0 = block_sum256(cm2(w), L, lane, wave);
                    pos += (int64_t)FX_HOP * hw; fresh = false;
                }
            } else if (hw >= 0) {
                // candidate preamble at p: the exact detector takes over on the hop whose new half starts at p, behind a
                // synchroniser reset at p - 256 (the hop before it -- zeros and x[p-256, p) -- cannot hold the preamble)
                const int64_t p = pos - FX_HOP + (int64_t)FX_HOP * hw + (int64_t)L.cand[hw];
                pos = p; floor_ = p - FX_HOP; fresh = false; exact_left = 2;
                const float2 w = lo ? xv(xs, pos - FX_HOP + tid, floor_) : make_float2(0.0f, 0.0f);
                if (lo) L.win[tid] = w;
                x2_0 = block_sum256(cm2(w), L, lane, wave);
            } else {
                const float2 w = lo ? L.cw[WALK_WAVES * FX_HOP + tid] : make_float2(0.0f, 0.0f);
                if (lo) L.win[tid] = w;                                  // last hop becomes the overlap half
                if (locked) x2_0 = block_sum256(cm2(w), L, lane, wave);  // (a walker not yet locked re-arms fresh on a hit)
                pos += WALK_WAVES * FX_HOP; fresh = false;
            }
            __syncthreads();
            WSTAMP(0);
            continue;
        }

        float2 nw = lo ? xv(xs, pos + tid, floor_) : make_float2(0.0f, 0.0f);
        if (lo) L.win[FX_HOP + tid] = nw;

        // ------------------------------------------------------------ pre-lock coarse scan (speculative walkers only)
        // Until a speculative walker has locked onto the chain, nothing it produces is kept, so it may look
        // for its first preamble any way it likes.  A CFO-blind differential correlator needs 2 FFTs per hop
        // instead of the 50 of the real detector: d[i] = w[i+1] conj(w[i]) against the zero-mean td[k] = s[k+1] conj(s[k]).
        // A hit at lag l re-arms the exact detector (fresh) one hop before the candidate.
        bool coarse_hit = false;
        if ((!locked || may_skip) && MODE == FX_MODE_FLEXRX && exact_left == 0) {
            __syncthreads();
            hops_cheap++;
            float2 d0 = lo ? cmulc(L.win[tid + 1], L.win[tid]) : make_float2(0.0f, 0.0f);
            float2 d1 = (tid < 255) ? cmulc(L.win[tid + 257], L.win[tid + 256]) : make_float2(0.0f, 0.0f);
            if (lo) { L.X[tid] = d0; L.X[tid + 256] = d1; }
            // energy of d about its mean: oversampled signals give d a large DC term that is not information
            float ed = block_sum256(cm2(d0) + cm2(d1), L, lane, wave);
            const float2 dsum = block_csum256(cadd(d0, d1), L, lane, wave);
            ed -= cm2(dsum) * (1.0f / (float)FX_NFFT);
            if (wave == 0) {
                float2 a[8];
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = L.X[lane + 64 * q];
                fft512_wave(a, L.scr[0], lane, twA, twB);
                const int kb = (lane >> 3) + 8 * (lane & 7);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 8; t++) { float2 y = cmulc(a[t], T->TD[kb + 64 * t]); L.X[kb + 64 * t] = make_float2(y.y, y.x); }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 8; q++) a[q] = L.X[lane + 64 * q];
                fft512_wave(a, L.scr[0], lane, twA, twB);
                float bv = -1.0f; uint32_t bk = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) {
                    const uint32_t k = (uint32_t)kb + 64 * t;
                    const float m = cm2(a[t]);
                    if (k < FX_NFFT - FX_S_LEN && m > bv) { bv = m; bk = k; }
                }
                wave_argmax(bv, bk);
                if (lane == 0) { L.f[4] = bv; L.u[8] = bk; }
            }
            __syncthreads();
            const float cpk = L.f[4]; const uint32_t cl = L.u[8];
            if (cpk > 0.06f * ed * T->td2sum * (float)FX_NFFT * (float)FX_NFFT && ed > 0.0f) {
#ifdef FX_EXACT_HOPS_OLD
                if (locked) { coarse_hit = true; exact_left = 3; hops_cheap--; }   // run the exact detector on this very hop
#else
                if (locked) { coarse_hit = true; exact_left = cl >= FX_NFFT - FX_S_LEN - 12u ? 2 : 1; hops_cheap--; }   // run the exact detector on this very hop
#endif
                else {
                    // candidate preamble at p: the exact detector takes over on the hop whose new half starts at p (see above)
                    const int64_t p = pos - FX_HOP + (int64_t)cl;
                    pos = p; floor_ = p - FX_HOP; fresh = false; exact_left = 2;
                    __syncthreads();
                    const float2 w = lo ? xv(xs, pos - FX_HOP + tid, floor_) : make_float2(0.0f, 0.0f);
                    if (lo) L.win[tid] = w;
                    x2_0 = block_sum256(cm2(w), L, lane, wave);
                    __syncthreads();
                    continue;
                }
            }
            if (!coarse_hit) {
                if (lo) L.win[tid] = nw;
                if (locked) x2_0 = block_sum256(cm2(nw), L, lane, wave);
                pos += FX_HOP; fresh = false;
                __syncthreads();
                continue;
            }
        }
        if (exact_left) exact_left--;

        // ------------------------------------------------------------ SEEK: one 256-sample hop
        const float x2_1 = block_sum256(cm2(nw), L, lane, wave);     // barriers inside publish win[]
        hops++;
        bool det; uint32_t bidx; int boff; float peak;
        det = seek_sweep(L, x2_0, x2_1, job.threshold, s2sum, twA, twB, lane, wave, bidx, boff, peak);
        WSTAMP(1);
        if (!det) {                                                    // slide the window by one hop
            __syncthreads();
            if (lo) L.win[tid] = nw;
            x2_0 = x2_1; pos += FX_HOP; fresh = false;
            __syncthreads();
            continue;
        }

        const int64_t a0 = pos - FX_HOP + (int64_t)bidx;
        if (in_handoff) {
            bool extend = false;
            if constexpr (EXT) {
                if (ext_left > 0 && all_jobs) {
                    uint32_t k = job_index + 1;                            // the segment the detection falls in (as fx_chain_kernel picks it)
                    while (k + 1 < n_jobs_total && all_jobs[k + 1].stream == job.stream && a0 >= all_jobs[k].stop + FX_HOP) k++;
                    if (k < n_jobs_total && all_jobs[k].stream == job.stream) {
                        const FxFrame *FN = frames + all_jobs[k].frame_base;
                        const uint32_t nfn = all_results[k].n_frames;
                        int hit = 0;
                        if (floor_ <= a0)
                            for (uint32_t i = tid; i < nfn; i += WALK_THREADS) {
                                const uint32_t fl = FN[i].flags;
                                if ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_FLOOR_CLEAR) && FN[i].start == a0 && FN[i].offset == boff) hit = 1;
                            }
                        hit = __syncthreads_or(hit);
                        const int64_t room = (int64_t)job.max_frames - (int64_t)nfr - 8;
                        if (!hit && room * 600 > all_jobs[k].stop - a0) { extend = true; stop = all_jobs[k].stop; ext_left--; in_handoff = false; }
                    }
                }
            }
            if (!extend) { has_handoff = 1; ho_start = a0; ho_off = boff; ho_rxy = peak; ho_pos = pos; ho_clear = floor_ <= a0 ? 1u : 0u; exit_code = FX_EXIT_STOP; break; }
        }
        if (nfr >= job.max_frames) { exit_code = FX_EXIT_TABLE_FULL; break; }
        if (a0 + FX_NFFT > n) { exit_code = FX_EXIT_NEED_DATA; break; }

        // ------------------------------------------------------------ ALIGN on x[a0, a0+512)
        __syncthreads();
        {
            constexpr int NLD = (FX_NFFT + WALK_THREADS - 1) / WALK_THREADS;
            float2 ld[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int i = tid + WALK_THREADS * k; ld[k] = i < FX_NFFT ? xv(xs, a0 + i, floor_) : make_float2(0.0f, 0.0f); }
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int i = tid + WALK_THREADS * k; if (i < FX_NFFT) L.win[i] = ld[k]; }
        }
        __syncthreads();
        if (wave == 0) {
            float2 a[8];
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = L.win[lane + 64 * q];
            fft512_wave(a, L.scr[0], lane, twA, twB);
            const int kb = (lane >> 3) + 8 * (lane & 7);
#pragma unroll
            for (int t = 0; t < 8; t++) L.X[kb + 64 * t] = a[t];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int i = lane + 64 * q;
                float2 y = cmulc(L.X[i], L.S[(i - boff) & (FX_NFFT - 1)]);
                a[q] = make_float2(y.y, y.x);
            }
            fft512_wave(a, L.scr[0], lane, twA, twB);
            // lags 0, +1, -1 sit in lanes 0 (t=0), 8 (t=0), 63 (t=7)
            if (lane == 0)  L.f[0] = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);
            if (lane == 8)  L.f[1] = fmaf(a[0].y, a[0].y, a[0].x * a[0].x);
            if (lane == 63) L.f[2] = fmaf(a[7].y, a[7].y, a[7].x * a[7].x);
        }
        __syncthreads();
        float tau, gamma;
        {
            float y0 = sqrtf(sqrtf(L.f[0])), ypos = sqrtf(sqrtf(L.f[1])), yneg = sqrtf(sqrtf(L.f[2]));
            float qa = 0.5f * (ypos + yneg) - y0, qb = 0.5f * (ypos - yneg);
            tau = qa == 0.0f ? 0.0f : -qb / (2.0f * qa);
            if (!(fabsf(tau) < 1.0f)) tau = 0.0f;
            float gh = fmaf(fmaf(qa, tau, qb), tau, y0);
            gamma = gh * gh / ((float)FX_NFFT * s2sum);
        }
        // carrier frequency: spectral peak of x conj(s)
        {
            float2 p = make_float2(0.0f, 0.0f);
            if (tid < FX_S_LEN) p = cmulc(L.win[tid], T->s[tid]);
            if (lo) L.P[tid] = p;
        }
        __syncthreads();
        if (wave == 0) {
            float2 a[8];
#pragma unroll
            for (int q = 0; q < 8; q++) a[q] = (q < 4) ? L.P[lane + 64 * q] : make_float2(0.0f, 0.0f);
            fft512_wave(a, L.scr[0], lane, twA, twB);
            const uint32_t kb = (lane >> 3) + 8 * (lane & 7);
            float bv = -1.0f; uint32_t bk = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                float m = cm2(a[t]);
                L.m2[kb + 64 * t] = m;
                if (m > bv) { bv = m; bk = kb + 64 * t; }
            }
            wave_argmax(bv, bk);
            if (lane == 0) { L.f[3] = bv; L.u[0] = bk; }
        }
        __syncthreads();
        float dphi;
        {
            const uint32_t i0 = L.u[0];
            float v0 = sqrtf(L.f[3]);
            float vneg = sqrtf(L.m2[(i0 + FX_NFFT - 1) & (FX_NFFT - 1)]);
            float vpos = sqrtf(L.m2[(i0 + 1) & (FX_NFFT - 1)]);
            float qa = 0.5f * (vpos + vneg) - v0, qb = 0.5f * (vpos - vneg);
            float idx = qa == 0.0f ? 0.0f : -qb / (2.0f * qa);
            float index = (float)i0 + idx;
            dphi = (i0 > FX_NFFT / 2 ? index - (float)FX_NFFT : index) * (6.28318531f / (float)FX_NFFT);
        }
        const uint32_t mix_dl = rad2u32(dphi);
        float phi;
        {
            float2 term = make_float2(0.0f, 0.0f);
            if (tid < FX_S_LEN) term = derot(L.P[tid], mix_dl * (uint32_t)tid, sc);
            float2 metric = block_csum256(term, L, lane, wave);
            phi = atan2c(metric.y, metric.x);
        }

        WSTAMP(2);
        FxFrameHead fr;
        fr.start = a0; fr.offset = boff; fr.rxy = peak; fr.tau = tau; fr.gamma = gamma; fr.dphi = dphi; fr.phi = phi;
        fr.seek_pos = span_pos; fr.seek_floor = span_floor; fr.det_pos = pos;
        fr.pfb = 0; fr.mfc0 = 0; fr.mix_th = rad2u32(phi); fr.mix_dl = mix_dl; fr.mf_scale = 0.0f;
        fr.pilot_dphi = fr.pilot_phi = fr.pilot_gain = 0.0f; fr.pll_th = 0; fr.pll_f = 0.0f; fr.flags = (floor_ <= a0 ? FX_FLAG_FLOOR_CLEAR : 0u) | span_flags;
        fr.pay_len = fr.ms = fr.check = fr.fec0 = fr.fec1 = fr.pay_sym_len = 0;
#pragma unroll
        for (int j = 0; j < FX_HDR_DEC; j++) fr.header[j] = 0;

        if (MODE == FX_MODE_DETECT) {
            // Speculative walker not yet locked: a weak peak may be a false alarm the sequential chain
            // never sees (its hop grid differs).  Ignore it and keep the grid; lock on a strong one.
            if (!locked && !(peak > 0.7f)) {
                __syncthreads();
                if (lo) { L.win[tid] = nw; L.win[FX_HOP + tid] = make_float2(0.0f, 0.0f); }
                x2_0 = x2_1; pos += FX_HOP; fresh = false;
                __syncthreads();
                continue;
            }
            locked = true; fr.flags |= FX_FLAG_EXACT;
            // back to SEEK with the second half of the aligned window as overlap
            fr.next = a0 + FX_NFFT;
            if (tid == 0) static_cast<FxFrameHead &>(frames[job.frame_base + nfr]) = fr;
            nfr++;
            span_pos = a0 + FX_NFFT; span_floor = floor_; span_flags = FX_FLAG_SPAN_EXACT;
            __syncthreads();
            float2 w = lo ? L.win[FX_HOP + tid] : make_float2(0.0f, 0.0f);
            __syncthreads();
            if (lo) L.win[tid] = w;
            x2_0 = block_sum256(cm2(w), L, lane, wave);
            pos = a0 + FX_NFFT; fresh = false;
            __syncthreads();
            continue;
        }

        // ------------------------------------------------------------ flex_rx: preamble + header span
        if (tau > 0.0f) { fr.pfb = (unsigned)(tau * (float)FX_NPFB) % FX_NPFB; fr.mfc0 = 0; }
        else { fr.pfb = (unsigned)((1.0f + tau) * (float)FX_NPFB) % FX_NPFB; fr.mfc0 = 1; }
        fr.mf_scale = 0.5f / gamma;
        constexpr int dly = EQ ? FX_EQ_DELAY : 0;                      // the equaliser moves every symbol instant 3 symbols later
        const int nh = (int)sym_sample(FX_SYM0_PAY + dly - 1, fr.mfc0);  // sample of the last header symbol
        if (a0 + nh + 1 > n) { exit_code = FX_EXIT_NEED_DATA; break; }
        {
            constexpr int NLD = (640 + WALK_THREADS - 1) / WALK_THREADS;        // (nh < 640: the preamble + header span)
            float2 ld[NLD];
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int m = tid + WALK_THREADS * k; ld[k] = m <= nh ? xv(xs, a0 + m, floor_) : make_float2(0.0f, 0.0f); }
#pragma unroll
            for (int k = 0; k < NLD; k++) { const int m = tid + WALK_THREADS * k; if (m <= nh) L.v[m] = derot(ld[k], fr.mix_th + mix_dl * (uint32_t)m, sc); }
        }
        if (tid < FX_MF_TAPS) L.taps[tid] = T->proto[fr.pfb + FX_NPFB * tid];
        __syncthreads();
        if (!EQ) {
            if (tid < FX_HDR_SYM) {
                const int nc = (int)sym_sample(FX_SYM0_HDR + tid, fr.mfc0);
                float ar = 0.0f, ai = 0.0f;
#pragma unroll 4
                for (int t = 0; t < FX_MF_TAPS; t++) {
                    const float2 w = L.v[nc - t]; const float h = L.taps[t];
                    ar = fmaf(h, w.x, ar); ai = fmaf(h, w.y, ai);
                }
                L.hdr[tid] = make_float2(ar * fr.mf_scale, ai * fr.mf_scale);
            }
        } else {
            // Equaliser stage (liquid: FLEXFRAMESYNC_ENABLE_EQ): the matched filter is evaluated at every sample, a 13-tap
            // eqlms at 2 samples/symbol follows it, trained on the 64 p/n symbols (normalised LMS, mu = 0.05) and frozen
            // from the header on.  mfo[] (matched-filter outputs of the whole span) borrows the coarse scan's window.
            float2 *mfo = L.cw;
            for (int m = tid; m <= nh; m += WALK_THREADS) {
                float ar = 0.0f, ai = 0.0f;
#pragma unroll 4
                for (int t = 0; t < FX_MF_TAPS; t++) {
                    const float2 w = (m - t >= 0) ? L.v[m - t] : make_float2(0.0f, 0.0f); const float h = L.taps[t];
                    ar = fmaf(h, w.x, ar); ai = fmaf(h, w.y, ai);
                }
                mfo[m] = make_float2(ar * fr.mf_scale, ai * fr.mf_scale);
            }
            __syncthreads();
            if (wave == 0) {                                               // lane i < 13 owns tap i; sums are 16-lane xor butterflies
                float2 w = lane < FX_EQ_TAPS ? make_float2(T->eq0[lane], 0.0f) : make_float2(0.0f, 0.0f);
                for (int c = 2 * FX_M + dly; c < FX_SYM0_HDR + dly; c++) {
                    const int nc = (int)sym_sample(c, fr.mfc0);
                    const float2 r = lane < FX_EQ_TAPS ? mfo[nc - (FX_EQ_TAPS - 1) + lane] : make_float2(0.0f, 0.0f);
                    float2 y = cmulc(r, w); float x2 = cm2(r);
#pragma unroll
                    for (int mk = 1; mk < 16; mk <<= 1) { y.x += __shfl_xor(y.x, mk, 64); y.y += __shfl_xor(y.y, mk, 64); x2 += __shfl_xor(x2, mk, 64); }
                    if (x2 > 0.0f) {
                        const float2 d = T->pn[c - 2 * FX_M - dly];
                        const float2 e = make_float2(d.x - y.x, d.y - y.y);
                        const float g = FX_EQ_MU / x2;
                        const float2 cc = cmulc(r, e);
                        w.x = fmaf(g, cc.x, w.x); w.y = fmaf(g, cc.y, w.y);
                    }
                }
                if (lane < 16) L.eqw[lane] = w;
            }
            __syncthreads();
            if (tid < FX_HDR_SYM) {
                const int nc = (int)sym_sample(FX_SYM0_HDR + dly + tid, fr.mfc0);
                L.hdr[tid] = eq_sum16(mfo + nc - (FX_EQ_TAPS - 1), L.eqw);
            }
        }
        __syncthreads();
        // pilot sync: 32-point DFT of the 15 de-rotated pilots
        if (tid < FX_HDR_PILOTS) L.pb[tid] = cmulc(L.hdr[FX_PILOT_SPACING * tid], T->pilots[tid]);
        __syncthreads();
        if (tid < 32) {
            float ar = 0.0f, ai = 0.0f;
            for (int p = 0; p < FX_HDR_PILOTS; p++) {
                const float2 w = T->tw[16 * ((tid * p) & 31)], b = L.pb[p];
                ar = fmaf(b.x, w.x, ar); ar = fmaf(-b.y, w.y, ar);
                ai = fmaf(b.x, w.y, ai); ai = fmaf(b.y, w.x, ai);
            }
            L.m2[tid] = fmaf(ar, ar, ai * ai);
        }
        __syncthreads();
        float pdphi, pphi, pgain;
        {
            float best = -1.0f; unsigned i0 = 0;
            for (unsigned k = 0; k < 32; k++) { float m = L.m2[k]; if (m > best) { best = m; i0 = k; } }
            float y0 = sqrtf(L.m2[i0]), yneg = sqrtf(L.m2[(i0 + 31) & 31]), ypos = sqrtf(L.m2[(i0 + 1) & 31]);
            float qa = 0.5f * (ypos + yneg) - y0, qb = 0.5f * (ypos - yneg);
            float idx = qa == 0.0f ? 0.0f : -qb / (2.0f * qa);
            float index = i0 < 16 ? (float)i0 : (float)i0 - 32.0f;
            pdphi = (index + idx) * (6.28318531f / 512.0f);
        }
        const uint32_t pdl = rad2u32(pdphi);
        {
            float mr = 0.0f, mi = 0.0f;
            for (unsigned p = 0; p < FX_HDR_PILOTS; p++) {
                float2 t = derot(L.pb[p], pdl * (FX_PILOT_SPACING * p), sc);
                mr += t.x; mi += t.y;
            }
            pphi = atan2c(mi, mr);
            pgain = sqrtf(fmaf(mr, mr, mi * mi)) / (float)FX_HDR_PILOTS;
        }
        const uint32_t pph = rad2u32(pphi);
        const float pg = 1.0f / pgain;
        if (tid < FX_HDR_SYM && (tid % FX_PILOT_SPACING) != 0) {
            float2 y = derot(L.hdr[tid], pph + pdl * (uint32_t)tid, sc);
            y.x *= pg; y.y *= pg;
            const int nn = tid - 1 - tid / FX_PILOT_SPACING;           // data index (pilots removed)
            L.hs[nn] = (uint8_t)((y.x > 0.0f ? 0u : 1u) | (y.y > 0.0f ? 0u : 2u));
        }
        __syncthreads();
        if (tid < FX_HDR_ENC)
            L.b0[tid] = (uint8_t)((L.hs[4 * tid] << 6) | (L.hs[4 * tid + 1] << 4) | (L.hs[4 * tid + 2] << 2) | L.hs[4 * tid + 3]);
        __syncthreads();
        decode_header_bytes(L, T, tid);
        if (tid == 0) {
            const uint8_t *hd = L.b1;
            int ok = (int)L.u[1];
            unsigned pay_len = 0, ms = 0, check = 0, fec0 = 0, fec1 = 0, nsym = 0;
            if (ok) {
                const uint8_t *h = hd + FX_HDR_USER;
                pay_len = ((unsigned)h[1] << 8) | h[2]; ms = h[3];
                check = (h[4] >> 5) & 7; fec0 = h[4] & 0x1f; fec1 = h[5] & 0x1f;
                const unsigned bps = modem_bps(ms);
                if (h[0] != FX_PROTOCOL || bps == 0 || check == FX_CRC_UNKNOWN || check > FX_CRC_32 ||
                    !fec_supported(fec0) || !fec_supported(fec1)) ok = 0;
                else {
                    unsigned bits = 8 * fec_enc_len(fec1, fec_enc_len(fec0, pay_len + crc_len(check)));
                    nsym = (bits + bps - 1) / bps;
                }
            }
            L.u[1] = (uint32_t)ok; L.u[2] = pay_len; L.u[3] = ms; L.u[4] = check; L.u[5] = fec0; L.u[6] = fec1; L.u[7] = nsym;
        }
        __syncthreads();
        const bool hv = L.u[1] != 0;
        fr.pilot_dphi = pdphi; fr.pilot_phi = pphi; fr.pilot_gain = pgain;
        fr.pll_f = pdphi; fr.pll_th = pph + pdl * (uint32_t)FX_HDR_SYM;
#pragma unroll
        for (int j = 0; j < FX_HDR_DEC; j++) fr.header[j] = L.b1[j];
        int64_t last_c = FX_SYM0_PAY + dly - 1;
        if (hv) {
            fr.flags |= FX_FLAG_HEADER_VALID;
            fr.pay_len = L.u[2]; fr.ms = L.u[3]; fr.check = L.u[4]; fr.fec0 = L.u[5]; fr.fec1 = L.u[6]; fr.pay_sym_len = L.u[7];
            last_c += fr.pay_sym_len;
        }
        fr.next = a0 + sym_sample(last_c, fr.mfc0) + 1;
        WSTAMP(3);
        if (!locked && !hv) {
            // speculative walker, header did not check out: most likely a false alarm on payload data.
            // Do not skip the 618 samples a real invalid frame would consume (a true preamble may sit
            // there); resume seeking on the same grid.  Nothing before the lock is ever spliced.
            if (tid == 0) static_cast<FxFrameHead &>(frames[job.frame_base + nfr]) = fr;
            nfr++;
            __syncthreads();
            if (lo) L.win[tid] = nw;
            x2_0 = x2_1; pos += FX_HOP; fresh = false;
            __syncthreads();
            continue;
        }
        locked = true; fr.flags |= FX_FLAG_EXACT;
        bool incomplete = fr.next > n;
        if (incomplete) fr.flags |= FX_FLAG_INCOMPLETE;
        if (tid == 0) {
            if (!incomplete && emit_runs(fr.det_pos, job.frame_base + nfr)) fr.flags |= FX_FLAG_SPAN_BAD;
            static_cast<FxFrameHead &>(frames[job.frame_base + nfr]) = fr;
        }
        if (EQ && tid < FX_EQ_TAPS) frames[job.frame_base + nfr].eq[tid] = L.eqw[tid];
        nfr++;
        if (incomplete) { exit_code = FX_EXIT_PAYLOAD; break; }
        // synchroniser reset: fresh detector right after the frame's last symbol
        pos = fr.next; floor_ = fr.next; fresh = true; x2_0 = 0.0f;
        span_pos = pos; span_floor = floor_; span_flags = FX_FLAG_SEEK_FRESH | (may_skip ? 0u : FX_FLAG_SPAN_EXACT);
        __syncthreads();
        if (lo) L.win[tid] = make_float2(0.0f, 0.0f);
        __syncthreads();
    }

    if (tid == 0) {
        FxWalkResult r;
        r.n_frames = nfr; r.exit_code = exit_code; r.pos = pos; r.floor = floor_; r.fresh = fresh ? 1u : 0u;
        r.has_handoff = has_handoff; r.handoff_start = ho_start; r.handoff_offset = ho_off; r.hops = hops;
        r.handoff_rxy = ho_rxy; r.hops_cheap = hops_cheap;
        r.tail_pos = span_pos; r.tail_floor = span_floor; r.handoff_pos = ho_pos; r.handoff_clear = ho_clear;
        // the seek in progress: a frame cut short by the end of data is walked again by the next block, from this very state
        r.tail_flags = span_flags;
        if (emit_runs(has_handoff ? ho_pos : pos, 0x80000000u | job_index)) r.tail_flags |= FX_FLAG_SPAN_BAD;
#ifdef FX_STAMPS
        for (int i = 0; i < 4; i++) r.stamp[i] = wst_[i];
#else
        for (int i = 0; i < 4; i++) r.stamp[i] = 0;
#endif
        *result = r;
        atomicAdd(&hdr->hops, hops); atomicAdd(&hdr->hops_cheap, hops_cheap); atomicAdd(&hdr->walk_jobs_run, 1u);
    }
}

// the walker of a mode and its LDS layout: the detector-only mode has its own (detect_run, a wave per hop)
template <int MODE, int WW> struct WalkLdsSel { typedef WalkLdsT<WW> type; };
template <int WW> struct WalkLdsSel<FX_MODE_DETECT, WW> { typedef DetLdsT<WW> type; };
template <int MODE, int WW, bool EQ, bool EXT = false>
__device__ __forceinline__ void walk_any(const FxWalkJob &job, uint32_t job_index, FxWalkResult *result, FxFrame *frames, FxVerifyRun *runs, uint32_t run_cap,
                                         FxBlockHdr *hdr, const FxTables *T, typename WalkLdsSel<MODE, WW>::type &L, const float2 (&twA)[7], const float2 (&twB)[7],
                                         const FxWalkJob *all_jobs = nullptr, const FxWalkResult *all_results = nullptr, uint32_t n_jobs_total = 0)
{
    if constexpr (MODE == FX_MODE_DETECT) detect_run<WW, EXT>(job, job_index, result, frames, hdr, T, L, twA, twB, all_jobs, all_results, n_jobs_total);
    else walk_run<MODE, WW, EQ, EXT>(job, job_index, result, frames, runs, run_cap, hdr, T, L, twA, twB, all_jobs, all_results, n_jobs_total);
}

template <int MODE, int WW, bool EQ, bool EXT = false>
__global__ __launch_bounds__(64 * WW, MODE == FX_MODE_DETECT ? FX_DETECT_OCC : FX_FLEX_OCC) FX_WALK_VGPR_ATTR
void fx_walk_kernel(const FxWalkJob *jobs, const uint32_t *job_list, FxWalkResult *results, FxFrame *frames, FxVerifyRun *runs, uint32_t run_cap,
                    FxBlockHdr *hdr, const FxTables *T, uint32_t n_jobs_total, const uint32_t *n_list, uint32_t list_cap)
{
    __shared__ typename WalkLdsSel<MODE, WW>::type L;
    const int tid = threadIdx.x, lane = tid & 63;
    if constexpr (EXT) {
        // the repair round enqueued with the block (fx_host.cpp:enqueue_back): how many segments are queued is only known on the
        // device -- usually none, and the workgroups leave at once; else they stride over the list
        if (n_list && blockIdx.x >= min(*n_list, list_cap)) return;
    }
    float2 twA[7], twB[7];
#pragma unroll
    for (int r = 1; r < 8; r++) { twA[r - 1] = T->tw[lane * r]; twB[r - 1] = T->tw[8 * (lane & 7) * r]; }
    for (int i = tid; i < FX_NFFT; i += 64 * WW) L.S[i] = T->S[i];
    if constexpr (EXT) {
        if (n_list) {
            const uint32_t nreq = min(*n_list, list_cap);
            for (uint32_t i = blockIdx.x; i < nreq; i += gridDim.x) {
                const uint32_t ji = job_list[i] & 0x7fffffffu;
                const FxWalkJob job = jobs[ji];
                walk_any<MODE, WW, EQ, EXT>(job, ji, results + ji, frames, runs, run_cap, hdr, T, L, twA, twB, jobs, results, n_jobs_total);
                __syncthreads();
            }
            return;
        }
    }
    const uint32_t ji = job_list[blockIdx.x] & 0x7fffffffu;
    const FxWalkJob job = jobs[ji];
    walk_any<MODE, WW, EQ, EXT>(job, ji, results + ji, frames, runs, run_cap, hdr, T, L, twA, twB, jobs, results, n_jobs_total);
}

// (the equaliser stage is a compile-time variant of the flex_rx walker: the default instance carries none of its code)
// ext: the walks of a repair round (they may carry on into the segments behind theirs); n_list (ext only): the number of queued
// segments is read from there on the device and `njobs` workgroups stride over them (list_cap bounds the count)
extern "C" hipError_t fx_launch_walk(unsigned mode, int eq, unsigned njobs, hipStream_t st, const FxWalkJob *jobs, const uint32_t *job_list, FxWalkResult *results,
                                     FxFrame *frames, FxVerifyRun *runs, uint32_t run_cap, FxBlockHdr *hdr, const FxTables *T, int ext, uint32_t n_jobs_total,
                                     const uint32_t *n_list, uint32_t list_cap)
{
    if (njobs == 0) return hipSuccess;
#define FX_WALK_LAUNCH(M, W, E, X) hipLaunchKernelGGL((fx_walk_kernel<M, W, E, X>), dim3(njobs), dim3(64 * W), 0, st, jobs, job_list, results, frames, runs, run_cap, hdr, T, n_jobs_total, n_list, list_cap)
    if (mode == FX_MODE_DETECT) { if (ext) FX_WALK_LAUNCH(FX_MODE_DETECT, FX_DETECT_WAVES, false, true); else FX_WALK_LAUNCH(FX_MODE_DETECT, FX_DETECT_WAVES, false, false); }
    else if (ext) { if (eq) FX_WALK_LAUNCH(FX_MODE_FLEXRX, FX_FLEX_WAVES, true, true); else FX_WALK_LAUNCH(FX_MODE_FLEXRX, FX_FLEX_WAVES, false, true); }
    else if (eq) FX_WALK_LAUNCH(FX_MODE_FLEXRX, FX_FLEX_WAVES, true, false);
    else FX_WALK_LAUNCH(FX_MODE_FLEXRX, FX_FLEX_WAVES, false, false);
#undef FX_WALK_LAUNCH
    return hipGetLastError();
}

// ===================================================================== seek verification
// A run of consecutive hops of the exact detector, all expected to come up empty (the hops a locked walker skipped
// on the strength of its coarse scan).  Hops are independent given (pos, floor), so the runs -- emitted by the walkers
// themselves, counted in the block header -- spread over the whole chip; a workgroup takes runs grid-stride and slides its
// window exactly like the walker does.  A hop that does fire marks the span's owner (a frame-table slot, or a job's
// tail span) FX_FLAG_SPAN_BAD; fx_chain_kernel then walks that span again with the exact detector on every hop.
template <int WW>
__global__ __launch_bounds__(64 * WW, FX_VERIFY_OCC)
void fx_seekverify_kernel(const FxVerifyRun *runs, uint32_t run_cap, const FxWalkJob *jobs, FxWalkResult *results, FxFrame *frames, FxBlockHdr *hdr,
                          const FxTables *T, uint32_t phase)
{
    // phase 0: the runs [0, runs_done) -- those of the speculative walkers of a block of continuing streams, verified while
    // the block still waits for its predecessor's state; phase 1: the rest (runs_done is 0 when the block has no such wait)
    constexpr int WALK_THREADS = 64 * WW;
    __shared__ SeekLdsT<WW> L;
    const uint32_t split = min(hdr->runs_done, run_cap);
    const uint32_t run0 = phase ? split : 0u, nruns = phase ? min(hdr->n_runs, run_cap) : split;
    if (run0 + blockIdx.x >= nruns) return;
    const int tid_in = threadIdx.x;
    float2 twA[7], twB[7];
    {
        const int tid = tid_in, lane = tid & 63;
#pragma unroll
        for (int r = 1; r < 8; r++) { twA[r - 1] = T->tw[lane * r]; twB[r - 1] = T->tw[8 * (lane & 7) * r]; }
        for (int i = tid; i < FX_NFFT; i += WALK_THREADS) L.S[i] = T->S[i];
    }
    const float s2sum = T->s2sum;
    for (uint32_t ri = run0 + blockIdx.x; ri < nruns; ri += gridDim.x) {
        int tid = tid_in; asm volatile("" : "+v"(tid));        // (indices are computed where they are used, not kept: see walk_run)
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const bool lo = tid < HALF;
        const FxVerifyRun run = runs[ri];
        const FxWalkJob &jb = jobs[run.job];
        const XSrc xs = make_xsrc(jb.x, jb.xa_end, jb.n);
        const float threshold = jb.threshold;
        int64_t pos = run.pos;
        __syncthreads();
        float2 w = lo ? xv(xs, pos - FX_HOP + tid, run.floor) : make_float2(0.0f, 0.0f);
        if (lo) L.win[tid] = w;
        float x2_0 = block_sum256(cm2(w), L, lane, wave);
        bool fired = false;
        for (uint32_t h = 0; h < run.nhops; h++) {
            const float2 nw = lo ? xv(xs, pos + tid, run.floor) : make_float2(0.0f, 0.0f);
            if (lo) L.win[FX_HOP + tid] = nw;
            const float x2_1 = block_sum256(cm2(nw), L, lane, wave);
            uint32_t bidx; int boff; float peak;
            if (seek_sweep(L, x2_0, x2_1, threshold, s2sum, twA, twB, lane, wave, bidx, boff, peak)) { fired = true; break; }
            __syncthreads();
            if (lo) L.win[tid] = nw;
            x2_0 = x2_1; pos += FX_HOP;
            __syncthreads();
        }
        if (fired && tid == 0) {
            if (run.owner & 0x80000000u) atomicOr(&results[run.owner & 0x7fffffffu].tail_flags, (uint32_t)FX_FLAG_SPAN_BAD);
            else atomicOr(&frames[run.owner].flags, (uint32_t)FX_FLAG_SPAN_BAD);
            atomicAdd(&hdr->verify_failures, 1u);
        }
    }
}

extern "C" hipError_t fx_launch_seekverify(unsigned grid, hipStream_t st, const FxVerifyRun *runs, uint32_t run_cap, const FxWalkJob *jobs, FxWalkResult *results,
                                           FxFrame *frames, FxBlockHdr *hdr, const FxTables *T, uint32_t phase)
{
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(fx_seekverify_kernel<FX_VERIFY_WAVES>, dim3(grid), dim3(64 * FX_VERIFY_WAVES), 0, st, runs, run_cap, jobs, results, frames, hdr, T, phase);
    return hipGetLastError();
}

// ===================================================================== chain: stitch, repair, resume state, carried tail
// One workgroup per stream turns the walkers' lists into the list the sequential synchroniser would have produced
// (DESIGN.md section 2.1), entirely on the device:
//   * a segment's walker ends with its hand-off target (start, CFO bin); if the next segment's speculative list holds
//     that very frame (and no zero-floor masked what either of them read), the lists are spliced there;
//   * otherwise -- and wherever fx_seekverify_kernel found the exact detector firing on a hop a walker had skipped, or a
//     frame table filled up -- this workgroup walks the stretch itself, from the true state, exact detector on every hop
//     (walk_run: it IS a walker), and carries on from that walk's result;
//   * the chain's frames go, compacted, to the stream's region of the chain table; the resume state and the unconsumed
//     tail go to where the next block's true walker will look for them.
// Fast path (no repair needed anywhere, at most CHAIN_MAXJ segments): hand-off look-ups for all segments in parallel,
// a pointer chase through LDS, parallel compaction.  Anything else takes the general sequential loop below it.
#define CHAIN_MAXJ 1024
#define CHAIN_NONE 4095u

struct ChainLds {
    uint32_t lk[CHAIN_MAXJ + 1];        // successor segment (11 bits) | index of the hand-off target in its list (12 bits, CHAIN_NONE: not there)
    uint32_t info[CHAIN_MAXJ + 1];      // frames | exact frames << 12 | exit code << 24 | has hand-off << 27 | tail span bad << 28 | too many frames << 29
    uint32_t P[CHAIN_MAXJ + 1];         // frames the chain holds up to and including this segment's (minus the first segment's own)
    uint32_t accA[CHAIN_MAXJ + 1], accB[CHAIN_MAXJ + 1];
    uint16_t skip[CHAIN_MAXJ + 1], m[CHAIN_MAXJ + 1], pred[CHAIN_MAXJ + 1], jumpA[CHAIN_MAXJ + 1], jumpB[CHAIN_MAXJ + 1];
    uint8_t  markA[CHAIN_MAXJ + 1], markB[CHAIN_MAXJ + 1];
    uint32_t sh[8];
};

__device__ __forceinline__ void wg_sync_global() { __threadfence(); __syncthreads(); __threadfence(); }

__device__ __forceinline__ void copy_frame(FxFrame *dst, const FxFrame *src)
{
    // (frame tables are 256-byte records at 256-byte boundaries: sixteen 16-byte pieces, all loads in flight before the first store)
    const uint4 *a = reinterpret_cast<const uint4 *>(src); uint4 *b = reinterpret_cast<uint4 *>(dst);
    static_assert(sizeof(FxFrame) == 256, "sixteen 16-byte pieces");
    const uint4 v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3], v4 = a[4], v5 = a[5], v6 = a[6], v7 = a[7];
    const uint4 v8 = a[8], v9 = a[9], v10 = a[10], v11 = a[11], v12 = a[12], v13 = a[13], v14 = a[14], v15 = a[15];
    b[0] = v0; b[1] = v1; b[2] = v2; b[3] = v3; b[4] = v4; b[5] = v5; b[6] = v6; b[7] = v7;
    b[8] = v8; b[9] = v9; b[10] = v10; b[11] = v11; b[12] = v12; b[13] = v13; b[14] = v14; b[15] = v15;
}

// Fast path of the chain (nothing to repair, at most CHAIN_MAXJ segments): returns false when the general path is needed.
//   A. every segment, in parallel: its own summary and the look-up of its hand-off target in the list it points at;
//   B. which segments are on the chain, and how many frames the chain holds before each: the segments form a linked list
//      (first -> successor -> ...), ranked by pointer doubling in LDS -- log2(segments) rounds instead of a pointer chase;
//   C. compaction, one thread per segment on the chain.
template <int NT>
//
// Segments walked again from a true state (FxWalkJob.pad_ = 1, their start / floor / fresh rewritten: see "repair rounds" in
// fx_host.cpp) are entered plainly -- from their first frame, nothing spliced -- when the predecessor's hand-off hop state is
// that very state.  With req_list set, a segment on the chain whose hand-off target the next list does not hold is not only
// reported: the next segment's job is rewritten to start from the true state and queued for such a walk, and the chain is
// followed on through it as if the walk had already happened (its tail, and so its own hand-off, rarely changes), so that
// one round finds all the misses of a stream, not just the first.
__device__ __forceinline__ bool chain_fast_path(const FxStreamDesc &sd, const FxWalkJob *jobs, const FxWalkResult *results, const FxFrame *frames, FxFrame *out,
                                                ChainLds &C, uint32_t &cnt, int64_t &fin_pos, int64_t &fin_floor, bool &fin_fresh,
                                                FxWalkJob *jobs_rw = nullptr, uint32_t *req_list = nullptr, FxBlockHdr *hdr_rw = nullptr)
{
    const int tid = threadIdx.x;
    const uint32_t first = sd.first_job, nj = sd.n_jobs, T = nj;           // T: the list's end marker
    if (nj > CHAIN_MAXJ) return false;
    if (tid < 8) C.sh[tid] = 0;
    const unsigned long long tph_ = __builtin_readcyclecounter();      // phase clocks of stream 0 (fxrx_debug_chain_stamps): [4] look-ups, [5] list ranking
    __syncthreads();
    // A.
    for (uint32_t j = tid; j < nj; j += NT) {
        const FxWalkResult &R = results[first + j];
        const FxFrame *F = frames + jobs[first + j].frame_base;
        uint32_t nf = R.n_frames; const uint32_t ex = R.exit_code;
        if (ex == FX_EXIT_PAYLOAD && nf > 0) nf--;
        uint32_t cntE = 0; bool badspan = (R.tail_flags & FX_FLAG_SPAN_BAD) != 0;
        for (uint32_t i = 0; i < nf; i++) {
            const uint32_t fl = F[i].flags;
            cntE += (fl & FX_FLAG_EXACT) ? 1u : 0u;
            badspan = badspan || ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_SPAN_BAD));
        }
        // (repair rounds: a segment in which a skipped hop fires is walked again as it was, only with the exact detector on every
        // hop -- whether or not the chain turns out to pass through the span in question)
        uint32_t requeued = 0;
        if (badspan && req_list && !jobs[first + j].no_skip) {
            jobs_rw[first + j].no_skip = 1;
            req_list[atomicAdd(&hdr_rw->n_repair_req, 1u)] = (first + j) | 0x80000000u;   // (bit 31: for the host's per-stream bookkeeping)
            requeued = 1; C.sh[0] = 1;
        }
        C.info[j] = (nf & 4095u) | ((cntE & 4095u) << 12) | (ex << 24) | ((R.has_handoff ? 1u : 0u) << 27) | (((R.tail_flags & FX_FLAG_SPAN_BAD) ? 1u : 0u) << 28) |
                    ((nf >= CHAIN_NONE ? 1u : 0u) << 29) | (requeued << 30);
        uint32_t lk = CHAIN_NONE << 11, skipE = 0;
        if (R.has_handoff && ex == FX_EXIT_STOP && j + 1 < nj) {
            uint32_t nxt = j + 1;
            while (nxt + 1 < nj && R.handoff_start >= jobs[first + nxt].stop + FX_HOP) nxt++;
            const FxWalkResult &RN = results[first + nxt]; const FxFrame *FN = frames + jobs[first + nxt].frame_base;
            uint32_t found = CHAIN_NONE, eb = 0;
            uint32_t nfn = RN.n_frames, nfe = nfn; if (RN.exit_code == FX_EXIT_PAYLOAD && nfe > 0) nfe--;
            const FxWalkJob &JN = jobs[first + nxt];
            uint32_t plain = 0;
            if (JN.pad_ && R.pos == JN.start && R.floor == JN.floor && (R.fresh != 0) == (JN.fresh != 0)) { found = 0; plain = 1; }
            // A frame is a function of (start, CFO bin) alone only if no sample it reads was masked by a zero-floor: splice
            // only when both floors lie at or below the start; else the segment is walked from the true state.
            for (uint32_t i = 0; !plain && R.handoff_clear && i < nfn && i < CHAIN_NONE; i++) {
                const uint32_t fl = FN[i].flags;
                if ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_FLOOR_CLEAR) && FN[i].start == R.handoff_start && FN[i].offset == R.handoff_offset) { found = i; skipE = eb; break; }
                if ((fl & FX_FLAG_EXACT) && i < nfe) eb++;
            }
            lk = nxt | (found << 11) | (plain << 23);
        }
        C.lk[j] = lk; C.skip[j] = (uint16_t)skipE; C.m[j] = 0xFFFFu; C.pred[j] = 0xFFFFu;
    }
    __syncthreads();
    if (hdr_rw && blockIdx.x == 0 && tid == 0) hdr_rw->stamp[4] = (uint32_t)(__builtin_readcyclecounter() - tph_);
    // B. list ranking.  Edge j -> nxt carries the number of frames nxt contributes when entered from j.
    for (uint32_t j = tid; j <= nj; j += NT) {
        uint32_t jmp = T, w = 0;
        if (j < nj) {
            const uint32_t inf = C.info[j], ex = (inf >> 24) & 7u, lk = C.lk[j];
            const bool term = j + 1 == nj || ex != FX_EXIT_STOP || !((inf >> 27) & 1u);
            const bool hit = ((lk >> 11) & 4095u) != CHAIN_NONE;
            if (!term && (hit || req_list)) { jmp = lk & 2047u; w = hit ? ((C.info[jmp] >> 12) & 4095u) - C.skip[j] : 0u; }
            // (a speculative walker that never locked hands nothing over; it is only ever reached through a miss, and the walk
            // queued for it will carry on into the segments behind it: look for further misses there in the same round)
            if (req_list && j + 1 < nj && ex == FX_EXIT_STOP && !((inf >> 27) & 1u) && jobs[first + j].handoff) { jmp = j + 1; w = 0; }
        }
        C.jumpA[j] = (uint16_t)jmp; C.accA[j] = w; C.markA[j] = j == 0 ? 1 : 0; C.markB[j] = j == 0 ? 1 : 0; C.P[j] = 0;
    }
    __syncthreads();
    {
        uint16_t *jr = C.jumpA, *jw = C.jumpB; uint32_t *ar = C.accA, *aw = C.accB; uint8_t *mr = C.markA, *mw = C.markB;
        for (uint32_t span = 1; span < nj; span <<= 1) {
            for (uint32_t j = tid; j <= nj; j += NT) {
                const uint32_t k = jr[j];
                if (mr[j] && k != T) { mw[k] = 1; C.P[k] = C.P[j] + ar[j]; }       // nodes at distance [span, 2 span) from the head
                jw[j] = jr[k]; aw[j] = ar[j] + ar[k];
            }
            __syncthreads();
            for (uint32_t j = tid; j <= nj; j += NT) mr[j] = mw[j];
            __syncthreads();
            uint16_t *tj = jr; jr = jw; jw = tj; uint32_t *ta = ar; ar = aw; aw = ta;
        }
        // every segment on the chain tells its successor how it is entered
        for (uint32_t j = tid; j < nj; j += NT) {
            if (!mr[j]) continue;
            const uint32_t inf = C.info[j], ex = (inf >> 24) & 7u, lk = C.lk[j], found = (lk >> 11) & 4095u;
            const bool term = j + 1 == nj || ex != FX_EXIT_STOP || !((inf >> 27) & 1u);
            if (!term && found != CHAIN_NONE) { const uint32_t k = lk & 2047u; C.m[k] = (uint16_t)(found | (((lk >> 23) & 1u) ? 0u : 0x8000u)); C.pred[k] = (uint16_t)j; }
            if (j == 0) C.m[0] = 0;
        }
        __syncthreads();
        // problems, the chain's end, frame offsets
        const uint32_t c0 = (C.info[0] >> 12) & 4095u;
        for (uint32_t j = tid; j < nj; j += NT) {
            if (!mr[j]) continue;
            const uint32_t inf = C.info[j], nf = inf & 4095u, ex = (inf >> 24) & 7u, cm = C.m[j], m = cm & 0x7FFFu, lk = C.lk[j];
            const bool spliced = (cm & 0x8000u) != 0, nothing = spliced && nf <= m;
            const bool term = j + 1 == nj || ex != FX_EXIT_STOP || !((inf >> 27) & 1u);
            bool problem = ((inf >> 29) & 1u) || ex == FX_EXIT_TABLE_FULL || ex == FX_EXIT_INVALID;
            if (!nothing && ((inf >> 28) & 1u) && !((inf >> 30) & 1u)) problem = true;  // a skipped hop of its tail seek fires (and the segment is not queued for it)
            if (problem) C.sh[5] = 1;                                                  // (nothing a walk from a hand-off state mends)
            if ((inf >> 30) & 1u) problem = true;
            if (req_list && j + 1 < nj && ex == FX_EXIT_STOP && !((inf >> 27) & 1u) && jobs[first + j].handoff) problem = true;   // (passed through, see above)
            if (!term && ((lk >> 11) & 4095u) == CHAIN_NONE) {                         // target not in the next list: repair
                problem = true;
                if (req_list) {
                    const uint32_t k = lk & 2047u; const FxWalkResult &R = results[first + j];
                    FxWalkJob &J = jobs_rw[first + k];
                    J.start = R.pos; J.floor = R.floor; J.fresh = R.fresh ? 1u : 0u; J.prelock = 0; J.no_skip = 1; J.state_in = nullptr; J.pad_ = 1;
                    req_list[atomicAdd(&hdr_rw->n_repair_req, 1u)] = first + k;
                }
            }
            if (problem) C.sh[0] = 1;
            const uint32_t pj = C.pred[j];
            const uint32_t contrib = j == 0 ? c0 : ((inf >> 12) & 4095u) - (pj == 0xFFFFu ? 0u : C.skip[pj]);
            const uint32_t upto = c0 + C.P[j];                                         // frames up to and including this segment's
            C.accA[j] = upto - contrib;                                                // (accA is free now: offset of its first frame)
            if (term || problem) { C.sh[1] = j; C.sh[2] = nothing ? 1u : 0u; C.sh[3] = upto; }
        }
        __syncthreads();
        if (tid == 0 && !C.sh[0] && C.sh[3] > sd.chain_cap) { C.sh[0] = 1; C.sh[5] = 1; }
        __syncthreads();
    }
    if (hdr_rw && blockIdx.x == 0 && tid == 0) hdr_rw->stamp[5] = (uint32_t)(__builtin_readcyclecounter() - tph_);
    // C. compaction, a thread per segment on the chain.  (Tried: the segments' threads only list the chain's frames and the waves copy
    // them, a 256-byte record per coalesced load / store pair -- three times slower: four waves keep sixteen records in flight, 256
    // threads with sixteen 16-byte loads each keep thousands.)
    if (!C.sh[0]) {
        for (uint32_t j = tid; j < nj; j += NT) {
            const uint32_t cm = C.m[j];
            if (cm == 0xFFFFu) continue;
            const uint32_t m = cm & 0x7FFFu, nf = C.info[j] & 4095u; const bool spliced = (cm & 0x8000u) != 0;
            const FxFrame *F = frames + jobs[first + j].frame_base;
            uint32_t o = C.accA[j];
            for (uint32_t i = m; i < nf; i++) {
                const uint32_t fl = F[i].flags;
                if (!(fl & FX_FLAG_EXACT)) continue;
                const bool own = !(spliced && i == m);
                if (own && (fl & FX_FLAG_SPAN_BAD)) { C.sh[0] = 1; C.sh[5] = 1; }
                copy_frame(out + o, F + i);
                if (!own) out[o].rxy = results[first + C.pred[j]].handoff_rxy;          // coarse peak as the true chain saw it
                o++;
            }
        }
    }
    __syncthreads();
    const bool ok = !C.sh[0];
    if (ok) {
        const uint32_t ej = C.sh[1];
        const FxWalkResult &R = C.sh[2] ? results[first + C.pred[ej]] : results[first + ej];
        fin_pos = R.pos; fin_floor = R.floor; fin_fresh = R.fresh != 0; cnt = C.sh[3];
    }
    __syncthreads();
    return ok;
}

// resume state, carried tail, frame count of a stream's chain
template <int NT>
__device__ __forceinline__ void chain_finish(const FxStreamDesc &sd, uint32_t s, const FxStreamState &st_in, uint32_t cnt, int64_t fin_pos, int64_t fin_floor,
                                             bool fin_fresh, uint32_t *chain_count, FxBlockHdr *hdr)
{
    const int tid = threadIdx.x;
    const XSrc xs = make_xsrc(sd.x, sd.xa_end, sd.n);
    if (cnt > sd.chain_cap) { if (tid == 0) atomicOr(&hdr->flags, (uint32_t)FX_BLK_CHAIN_FULL); cnt = sd.chain_cap; }
    int64_t keep_from = fin_fresh ? fin_pos : fin_pos - FX_HOP;
    keep_from = max(-st_in.carry_len, min(keep_from, sd.n));
    const int64_t keep = sd.n - keep_from;
    const bool overflow = keep > sd.carry_cap;
    if (!overflow)
        for (int64_t i = tid; i < keep; i += NT) sd.carry_out_end[i - keep] = xld(xs, keep_from + i);
    if (tid == 0) {
        FxStreamState so;
        so.pos = fin_pos - sd.n; so.floor = max(fin_floor - sd.n, -keep); so.carry_len = keep; so.fresh = fin_fresh ? 1u : 0u;
        so.invalid = overflow ? 1u : 0u; so.overflow = overflow ? 1u : 0u; so.pad_ = 0;
        *sd.state_out = so; *sd.state_out_host = so;
        chain_count[s] = cnt;
        if (overflow) atomicOr(&hdr->flags, (uint32_t)FX_BLK_CARRY_OVERFLOW);
    }
}

__device__ __forceinline__ bool chain_state_in(const FxStreamDesc &sd, uint32_t s, FxStreamState &st_in, uint32_t *chain_count, FxBlockHdr *hdr)
{
    st_in.pos = 0; st_in.floor = 0; st_in.carry_len = 0; st_in.fresh = 1; st_in.invalid = 0; st_in.overflow = 0; st_in.pad_ = 0;
    if (sd.state_in) st_in = *sd.state_in;
    if (!st_in.invalid) return true;
    if (threadIdx.x == 0) {                                     // nothing to build on: say so downstream, the host replays
        FxStreamState so = st_in; so.invalid = 1; so.overflow = 0;
        *sd.state_out = so; *sd.state_out_host = so; chain_count[s] = 0;
        atomicOr(&hdr->flags, (uint32_t)FX_BLK_INVALID);
    }
    return false;
}

// The kernel every block runs: the fast path only, small enough (256 threads, 45 KB of LDS, few registers) to find room on
// a chip busy with the walkers and payload kernels of the other blocks in flight.  A stream that needs a repair is left
// without a chain and without a state (FX_BLK_NEEDS_REPAIR): fxrx_collect then runs fx_chain_kernel -- the full-size one,
// which can walk -- for the block and enqueues the blocks behind it again.
#define CHAINFAST_THREADS 256
// In-chain repair round (fx_host.cpp:enqueue_back): pass 1 is this kernel with the request list -- a stream that only has
// hand-off misses / fired skipped hops gets its segments queued and is left PENDING (stat[s] = 2) instead of being flagged;
// the queued segments are walked (fx_walk_kernel<..., EXT>, count read on the device); pass 2 stitches the pending streams
// again, without a request list, and flags what is still not settled (FX_BLK_NEEDS_REPAIR: fxrx_collect takes over, as it
// does for everything the fast path cannot do at all).  pass 0: the call of fxrx_collect's own repair rounds.
extern "C" __global__ __launch_bounds__(CHAINFAST_THREADS)
void fx_chainfast_kernel(const FxStreamDesc *streams, const FxWalkJob *jobs, const FxWalkResult *results, const FxFrame *frames, FxFrame *chain,
                         uint32_t *chain_count, FxBlockHdr *hdr, uint32_t force_repair, FxWalkJob *jobs_rw, uint32_t *req_list, uint32_t *stat, uint32_t pass)
{
    __shared__ ChainLds C;
    const uint32_t s = blockIdx.x;
    if (pass == 2u && stat[s] != 2u) return;                    // settled in pass 1, or beyond a repair round
    const FxStreamDesc sd = streams[s];
    FxStreamState st_in;
    if (!chain_state_in(sd, s, st_in, chain_count, hdr)) { if (pass == 1u && threadIdx.x == 0) stat[s] = 3u; return; }
    uint32_t cnt = 0; int64_t fin_pos = 0, fin_floor = 0; bool fin_fresh = true;
    const unsigned long long t0_ = __builtin_readcyclecounter();
    if (threadIdx.x == 0) C.sh[5] = 0;
    __syncthreads();
    const bool ok = !force_repair && chain_fast_path<CHAINFAST_THREADS>(sd, jobs, results, frames, chain + sd.chain_base, C, cnt, fin_pos, fin_floor, fin_fresh,
                                                                        pass == 2u ? nullptr : jobs_rw, pass == 2u ? nullptr : req_list, hdr);
    if (s == 0 && threadIdx.x == 0) { hdr->stamp[0] = (uint32_t)(__builtin_readcyclecounter() - t0_); }
    if (!ok) {
        if (threadIdx.x == 0) {
            // (FX_BLK_NEEDS_SLOW: not -- or not only -- hand-off misses: the full-size chain kernel has to go through it)
            const bool slow = force_repair || sd.n_jobs > CHAIN_MAXJ || C.sh[5];
            if (pass == 1u && !slow) stat[s] = 2u;               // its segments are queued: walked and stitched again within the chain
            else {
                FxStreamState so = st_in; so.invalid = 1; so.overflow = 0;
                *sd.state_out = so; *sd.state_out_host = so; chain_count[s] = 0;
                atomicOr(&hdr->flags, (uint32_t)(FX_BLK_NEEDS_REPAIR | (slow ? FX_BLK_NEEDS_SLOW : 0)));
                if (pass) stat[s] = 3u;
            }
        }
        return;
    }
    if (pass && threadIdx.x == 0) stat[s] = 1u;
    chain_finish<CHAINFAST_THREADS>(sd, s, st_in, cnt, fin_pos, fin_floor, fin_fresh, chain_count, hdr);
    if (s == 0 && threadIdx.x == 0) hdr->stamp[3] = (uint32_t)(__builtin_readcyclecounter() - t0_);
}

extern "C" hipError_t fx_launch_chainfast(unsigned nstreams, hipStream_t st, const FxStreamDesc *streams, const FxWalkJob *jobs, const FxWalkResult *results,
                                          const FxFrame *frames, FxFrame *chain, uint32_t *chain_count, FxBlockHdr *hdr, uint32_t force_repair,
                                          FxWalkJob *jobs_rw, uint32_t *req_list, uint32_t *stat, uint32_t pass)
{
    hipLaunchKernelGGL(fx_chainfast_kernel, dim3(nstreams), dim3(CHAINFAST_THREADS), 0, st, streams, jobs, results, frames, chain, chain_count, hdr, force_repair,
                       jobs_rw, req_list, stat, pass);
    return hipGetLastError();
}

// The full-size chain kernel: same fast path, and behind it the general, sequential one that can walk.
template <int MODE, int WW, bool EQ>
__global__ __launch_bounds__(64 * WW, 1)
void fx_chain_kernel(const FxStreamDesc *streams, const FxWalkJob *jobs, uint32_t n_jobs_total, FxWalkResult *results, FxFrame *frames,
                     FxFrame *chain, uint32_t *chain_count, FxVerifyRun *runs, uint32_t run_cap, FxBlockHdr *hdr, uint32_t force_slow,
                     const FxTables *T)
{
    constexpr int NT = 64 * WW;
    __shared__ typename WalkLdsSel<MODE, WW>::type L;
    __shared__ ChainLds C;
    const uint32_t s = blockIdx.x;
    const FxStreamDesc sd = streams[s];
    const int tid = threadIdx.x, lane = tid & 63;
    FxFrame *out = chain + sd.chain_base;
    FxStreamState st_in;
    if (!chain_state_in(sd, s, st_in, chain_count, hdr)) return;
    float2 twA[7], twB[7];
#pragma unroll
    for (int r = 1; r < 8; r++) { twA[r - 1] = T->tw[lane * r]; twB[r - 1] = T->tw[8 * (lane & 7) * r]; }
    for (int i = tid; i < FX_NFFT; i += NT) L.S[i] = T->S[i];

    const uint32_t first = sd.first_job, end = first + sd.n_jobs;
    uint32_t cnt = 0;
    int64_t fin_pos = 0, fin_floor = 0; bool fin_fresh = true;
    bool done = !force_slow && chain_fast_path<NT>(sd, jobs, results, frames, out, C, cnt, fin_pos, fin_floor, fin_fresh);

    // ------------------------------------------------------------------------------------------ general path
    if (!done) {
        uint32_t cur = first, m = 0;
        const FxWalkResult *Rp = results + cur; const FxFrame *F = frames + jobs[cur].frame_base;
        FxWalkResult *rep_res = results + n_jobs_total + s;
        float splice_rxy = -1.0f; bool spliced = false;
        int64_t tpos = 0, tfloor = 0; bool tfresh = true;
        cnt = 0;
        // a walk this workgroup does itself (one call site: the walker is a big piece of code)
        bool need_walk = false; uint32_t wj = 0; int64_t wst = 0, wfl = 0; bool wfr = false;
        auto repair = [&](uint32_t jidx, int64_t st, int64_t fl, bool fr) { need_walk = true; wj = jidx; wst = st; wfl = fl; wfr = fr; };
        for (;;) {
            if (need_walk) {
                FxWalkJob j = jobs[wj];
                j.start = wst; j.floor = wfl; j.fresh = wfr ? 1u : 0u; j.prelock = 0; j.no_skip = 1; j.state_in = nullptr;
                j.frame_base = sd.repair_base; j.max_frames = sd.repair_cap;
                __syncthreads();
                walk_any<MODE, WW, EQ>(j, wj, rep_res, frames, runs, run_cap, hdr, T, L, twA, twB);
                wg_sync_global();
                if (tid == 0) atomicAdd(&hdr->repairs, 1u);
                Rp = rep_res; F = frames + sd.repair_base; m = 0; need_walk = false;
            }
            const FxWalkResult R = *Rp;
            uint32_t nf = R.n_frames;
            if (R.exit_code == FX_EXIT_PAYLOAD && nf > 0) nf--;            // incomplete frame: the next block takes it
            int bad = -1;
            for (uint32_t i = m; i < nf; i++) {
                const uint32_t fl = F[i].flags;
                if (!(fl & FX_FLAG_EXACT)) continue;                       // tentative pre-lock entries of a speculative walk
                const bool own = !(i == m && splice_rxy >= 0.0f);          // (a spliced frame's seek is the hand-off's)
                if (own && (fl & FX_FLAG_SPAN_BAD)) { bad = (int)i; break; }
                if (cnt < sd.chain_cap) {
                    const uint32_t *a = reinterpret_cast<const uint32_t *>(F + i); uint32_t *b = reinterpret_cast<uint32_t *>(out + cnt);
                    if (tid < (int)(sizeof(FxFrame) / 4))               // coarse peak of a spliced frame: as the true chain saw it
                        b[tid] = (!own && tid == (int)(offsetof(FxFrameHead, rxy) / 4)) ? __builtin_bit_cast(uint32_t, splice_rxy) : a[tid];
                }
                cnt++;
            }
            if (bad >= 0) {                                                // a skipped hop of that frame's seek fires: walk it exactly
                const int64_t sp = F[bad].seek_pos, sf = F[bad].seek_floor; const bool sfr = (F[bad].flags & FX_FLAG_SEEK_FRESH) != 0;
                repair(cur, sp, sf, sfr); spliced = false; splice_rxy = -1.0f; continue;
            }
            // the seek in progress when this walker stopped (a spliced-in walker that contributed nothing is not on the chain)
            if (!(spliced && nf <= m) && (R.tail_flags & FX_FLAG_SPAN_BAD)) {
                repair(cur, R.tail_pos, R.tail_floor, (R.tail_flags & FX_FLAG_SEEK_FRESH) != 0); spliced = false; splice_rxy = -1.0f; continue;
            }
            splice_rxy = -1.0f;
            const bool last = (cur + 1 == end);
            if (R.exit_code == FX_EXIT_TABLE_FULL) { repair(cur, R.pos, R.floor, R.fresh != 0); spliced = false; continue; }   // same segment, where the table filled up
            if (last || R.exit_code != FX_EXIT_STOP || !R.has_handoff) {
                if (spliced && nf <= m) { fin_pos = tpos; fin_floor = tfloor; fin_fresh = tfresh; }   // its own hop state is speculative: resume from the true chain's hand-off hop
                else { fin_pos = R.pos; fin_floor = R.floor; fin_fresh = R.fresh != 0; }
                break;
            }
            spliced = false;
            // hand-off: look the target up in the next segment's speculative list
            // (segments the true walker crossed without a detection cannot hold the target: skip them)
            uint32_t nxt = cur + 1;
            while (nxt + 1 < end && R.handoff_start >= jobs[nxt].stop + FX_HOP) nxt++;
            const FxWalkResult *RNp = results + nxt; const FxFrame *FN = frames + jobs[nxt].frame_base;
            const uint32_t nfn = RNp->n_frames;
            uint32_t found = 0xFFFFFFFFu;
            // A frame is a function of (start, CFO bin) alone only if no sample it reads was masked by a zero-floor: splice
            // only when both floors lie at or below the start; else walk the segment from the true state.
            for (uint32_t i = 0; R.handoff_clear && i < nfn; i++) {
                const uint32_t fl = FN[i].flags;
                if ((fl & FX_FLAG_EXACT) && (fl & FX_FLAG_FLOOR_CLEAR) && FN[i].start == R.handoff_start && FN[i].offset == R.handoff_offset) { found = i; break; }
            }
            if (found != 0xFFFFFFFFu) {
                splice_rxy = R.handoff_rxy; spliced = true; tpos = R.pos; tfloor = R.floor; tfresh = R.fresh != 0;
                cur = nxt; m = found; Rp = RNp; F = FN; continue;
            }
            repair(nxt, R.pos, R.floor, R.fresh != 0);
            cur = nxt;
        }
        __syncthreads();
    }

    chain_finish<NT>(sd, s, st_in, cnt, fin_pos, fin_floor, fin_fresh, chain_count, hdr);
}

extern "C" hipError_t fx_launch_chain(unsigned mode, int eq, unsigned nstreams, hipStream_t st, const FxStreamDesc *streams, const FxWalkJob *jobs, uint32_t n_jobs_total,
                                      FxWalkResult *results, FxFrame *frames, FxFrame *chain, uint32_t *chain_count, FxVerifyRun *runs, uint32_t run_cap,
                                      FxBlockHdr *hdr, uint32_t force_slow, const FxTables *T)
{
    if (mode == FX_MODE_DETECT)
        hipLaunchKernelGGL((fx_chain_kernel<FX_MODE_DETECT, FX_DETECT_WAVES, false>), dim3(nstreams), dim3(64 * FX_DETECT_WAVES), 0, st, streams, jobs, n_jobs_total, results,
                           frames, chain, chain_count, runs, run_cap, hdr, force_slow, T);
    else if (eq)
        hipLaunchKernelGGL((fx_chain_kernel<FX_MODE_FLEXRX, FX_FLEX_WAVES, true>), dim3(nstreams), dim3(64 * FX_FLEX_WAVES), 0, st, streams, jobs, n_jobs_total, results,
                           frames, chain, chain_count, runs, run_cap, hdr, force_slow, T);
    else
        hipLaunchKernelGGL((fx_chain_kernel<FX_MODE_FLEXRX, FX_FLEX_WAVES, false>), dim3(nstreams), dim3(64 * FX_FLEX_WAVES), 0, st, streams, jobs, n_jobs_total, results,
                           frames, chain, chain_count, runs, run_cap, hdr, force_slow, T);
    return hipGetLastError();
}

// ===================================================================== plan: chain frames -> payload jobs, work lists, result records
// One workgroup lays the payload stage out on the device: arena offsets by prefix sums over the chain frames of all streams
// (stream order, then position -- the order results are reported in), the matched-filter work items, the PLL lists (one per
// modulation class, each padded to whole waves) and the two decode lists (with / without a Reed-Solomon stage).  It also
// writes one FxOutRec per frame straight into pinned host memory and mirrors the block header there.
#define PLAN_THREADS 256
#define PLAN_NV 7

__device__ __forceinline__ unsigned pll_class(unsigned ms)
{
    switch (ms) {
    case FX_MODEM_PSK2: return 0; case FX_MODEM_PSK4: return 1; case FX_MODEM_PSK8: return 2; case FX_MODEM_PSK16: return 3;
    case FX_MODEM_DPSK2: return 4; case FX_MODEM_DPSK4: return 5; case FX_MODEM_DPSK8: return 6; case FX_MODEM_ASK4: return 7;
    case FX_MODEM_QAM16: return 8; case FX_MODEM_QAM32: return 9; case FX_MODEM_QAM64: return 10; default: return 11;   // QPSK
    }
}

// exclusive scan of PLAN_NV values per thread over the workgroup, in one go (two barriers); v[] -> prefixes, tot[] -> totals
__device__ __forceinline__ void plan_scan(uint32_t (&v)[PLAN_NV], uint32_t (&tot)[PLAN_NV], uint32_t (*ws)[PLAN_NV])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t inc[PLAN_NV];
#pragma unroll
    for (int q = 0; q < PLAN_NV; q++) {
        inc[q] = v[q];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)inc[q], d, 64); if (lane >= d) inc[q] += o; }
    }
    __syncthreads();
    if (lane == 63) for (int q = 0; q < PLAN_NV; q++) ws[wave][q] = inc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PLAN_NV; q++) {
        uint32_t wb = 0, t = 0;
#pragma unroll
        for (int w = 0; w < PLAN_THREADS / 64; w++) { const uint32_t x = ws[w][q]; if (w < wave) wb += x; t += x; }
        tot[q] = t; v[q] = wb + inc[q] - v[q];
    }
}

// hdr: the block's walk-phase counters (zeroed again at the end, for the slot's next block); hdr_pay: what the payload kernels
// read; hdr_host: the host's copy.
//
// Two kernels, each a handful of workgroups over contiguous ranges of the frames (one workgroup took 2 ms for the 38 k frames
// and 317 k trellis work items of config 4):
//   fx_plan_kernel       sizes -> arena offsets (prefix sums), jobs, records, matched-filter items, counts per list.  The
//                        prefix across workgroups is a decoupled look-back: a workgroup publishes the totals of its range,
//                        then adds up what the workgroups before it have published (their totals, or their inclusive prefix
//                        once they have one).  It only ever waits for lower-numbered workgroups, which were dispatched first.
//   fx_planlists_kernel  list slots (the counts are complete: kernel boundary), padding, and -- by the workgroup that
//                        finishes last -- the block header, its mirrors, and the zeroing of counters and workspace.
// Workspace (uint32, zero between blocks): [0] finish ticket | [16, 60) counts and fill cursors | [128, 384) look-back flags |
// [384, 2432) range totals | [2432, 4480) inclusive prefixes
#define PLAN_MAXG 256
enum { PW_TICKET = 0, PW_CLS_CNT = 16, PW_DEC_CNT = 28, PW_VBC_CNT = 31, PW_CLS_FILL = 38, PW_DEC_FILL = 50, PW_VBC_FILL = 53,
       PW_FLAG = 128, PW_AGG = 384, PW_INC = 384 + 8 * PLAN_MAXG, PW_WORDS = 384 + 16 * PLAN_MAXG };

// Wave-aggregated atomic add: the lanes of a wave that add to the same counter counters[key] do so with ONE atomic (38 k
// frames of one modulation and one code would otherwise queue up on three addresses).  Returns, per active lane, the
// counter's value before its own contribution, as if the lanes had added one after the other in lane order.  Called by all
// lanes of the wave (inactive ones pass active = false).
__device__ __forceinline__ uint32_t wave_agg_add(uint32_t *counters, uint32_t key, uint32_t inc, bool active)
{
    const int lane = threadIdx.x & 63;
    unsigned long long rem = __ballot(active);
    uint32_t result = 0;
    while (rem) {
        const int leader = __ffsll((long long)rem) - 1;
        const uint32_t k = (uint32_t)__shfl((int)key, leader, 64);
        const bool mine = active && key == k;
        const unsigned long long same = __ballot(mine);
        uint32_t incl = mine ? inc : 0u;                                      // inclusive prefix over the lanes of this key
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counters[k], total);
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (mine) result = base + incl - inc;
        rem &= ~same;
    }
    return result;
}

struct PlanFrame {                      // what both passes of fx_plan_kernel derive from a chain frame
    const FxFrame *fp; uint32_t sidx, bps, k, l0, l1, nblk, vnb; bool valid, batch;
};
__device__ __forceinline__ PlanFrame plan_frame(uint32_t g, bool live, const FxStreamDesc *streams, uint32_t nstreams, const uint32_t *stream_base, const FxFrame *chain,
                                                uint32_t detect, uint32_t vb_blk, uint32_t (&v)[PLAN_NV])
{
    PlanFrame f; f.fp = chain; f.sidx = 0; f.bps = f.k = f.l0 = f.l1 = f.nblk = f.vnb = 0; f.valid = f.batch = false;
#pragma unroll
    for (int q = 0; q < PLAN_NV; q++) v[q] = 0;
    if (!live) return f;
    uint32_t lo = 0, hi = nstreams;                                          // stream_base[lo] <= g < stream_base[hi]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (stream_base[mid] <= g) lo = mid; else hi = mid; }
    f.sidx = lo;
    const FxFrame *fp = chain + streams[lo].chain_base + (g - stream_base[lo]);
    f.fp = fp;
    f.valid = !detect && (fp->flags & FX_FLAG_HEADER_VALID);
    if (f.valid) {
        const uint32_t nsym = fp->pay_sym_len, plen = fp->pay_len;
        f.bps = modem_bps(fp->ms);
        f.k = plen + crc_len(fp->check); f.l0 = fec_enc_len(fp->fec0, f.k); f.l1 = fec_enc_len(fp->fec1, f.l0);
        f.nblk = (nsym + 1023u) / 1024u;
        v[0] = (nsym + 7u) & ~7u;                                           // 8-symbol granules: 64-byte block I/O in the PLL kernel
        v[1] = (max(f.l1, f.k) + 8u + 15u) & ~15u;
        v[2] = ((8u * max(f.l0, f.k) + 6u + 63u) & ~63u) + 64u;              // whole 64-step chunks, lane-major
        v[3] = (plen + 15u) & ~15u;
        v[4] = f.nblk; v[5] = 1u;
        // batch Viterbi path: hard decisions, fec0 a K = 7 convolutional code, fec1 neither convolutional nor Reed-Solomon
        f.batch = vb_blk && conv_p(fp->fec0) && !conv_p(fp->fec1) && fp->fec1 != FX_FEC_RS_M8;
        if (f.batch) { f.vnb = (8u * f.k + 6u + vb_blk - 1u) / vb_blk; v[6] = f.vnb; }
    }
    return f;
}
// the frames a workgroup of a plan kernel handles: [g0, g1), whole chunks of PLAN_THREADS
__device__ __forceinline__ void plan_range(uint32_t N, uint32_t &g0, uint32_t &g1)
{
    const uint32_t per = ((N + gridDim.x - 1) / gridDim.x + PLAN_THREADS - 1) / PLAN_THREADS * PLAN_THREADS;
    g0 = min(N, blockIdx.x * per); g1 = min(N, g0 + per);
}
__device__ __forceinline__ uint32_t plan_stream_bases(const uint32_t *chain_count, uint32_t nstreams, uint32_t *stream_base, uint32_t (*ws)[PLAN_NV])
{
    // frames before each stream (every workgroup computes and writes the same values)
    const int tid = threadIdx.x;
    uint32_t run = 0;
    for (uint32_t s0 = 0; s0 < nstreams; s0 += PLAN_THREADS) {
        const uint32_t sidx = s0 + tid;
        uint32_t v[PLAN_NV] = { sidx < nstreams ? chain_count[sidx] : 0u, 0, 0, 0, 0, 0, 0 }, tot[PLAN_NV];
        plan_scan(v, tot, ws);
        if (sidx < nstreams) stream_base[sidx] = run + v[0];
        run += tot[0];
        __syncthreads();
    }
    if (tid == 0) stream_base[nstreams] = run;
    __threadfence(); __syncthreads(); __threadfence();
    return run;
}

extern "C" __global__ __launch_bounds__(PLAN_THREADS)
void fx_plan_kernel(const FxStreamDesc *streams, uint32_t nstreams, uint32_t detect, uint32_t eq, uint32_t vb_blk, const FxFrame *chain, const uint32_t *chain_count,
                    uint32_t *stream_base, FxPayJob *pjobs, FxOutRec *recs, uint32_t *mf_job, uint32_t *mf_c0, uint32_t mf_cap, uint32_t vb_cap, uint32_t *pw)
{
    // vb_blk: trellis steps per block of the batch Viterbi path (0: path off, e.g. soft decisions)
    __shared__ uint32_t ws[PLAN_THREADS / 64][PLAN_NV];
    __shared__ uint32_t base_s[PLAN_NV];
    const int tid = threadIdx.x;
    const uint32_t N = plan_stream_bases(chain_count, nstreams, stream_base, ws);
    uint32_t g0, g1; plan_range(N, g0, g1);
    const uint32_t wg = blockIdx.x;
    // 1. totals of the range
    uint32_t run[PLAN_NV] = { 0, 0, 0, 0, 0, 0, 0 };
    for (uint32_t c0 = g0; c0 < g1; c0 += PLAN_THREADS) {
        uint32_t v[PLAN_NV], tot[PLAN_NV];
        (void)plan_frame(c0 + tid, c0 + tid < g1, streams, nstreams, stream_base, chain, detect, vb_blk, v);
        plan_scan(v, tot, ws);
#pragma unroll
        for (int q = 0; q < PLAN_NV; q++) run[q] += tot[q];
        __syncthreads();
    }
    // 2. what lies before the range
    if (tid == 0) {
        uint32_t base[PLAN_NV] = { 0, 0, 0, 0, 0, 0, 0 };
        if (wg > 0) {
            for (int q = 0; q < PLAN_NV; q++) pw[PW_AGG + 8 * wg + q] = run[q];
            __threadfence();
            __hip_atomic_store(&pw[PW_FLAG + wg], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            for (int p = (int)wg - 1; p >= 0; p--) {
                uint32_t f;
                do { f = __hip_atomic_load(&pw[PW_FLAG + p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); if (!f) __builtin_amdgcn_s_sleep(2); } while (!f);
                const volatile uint32_t *src = pw + (f == 2u ? PW_INC : PW_AGG) + 8 * p;
                for (int q = 0; q < PLAN_NV; q++) base[q] += src[q];
                if (f == 2u) break;
            }
        }
        for (int q = 0; q < PLAN_NV; q++) { pw[PW_INC + 8 * wg + q] = base[q] + run[q]; base_s[q] = base[q]; }
        __threadfence();
        __hip_atomic_store(&pw[PW_FLAG + wg], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // 3. jobs, records, matched-filter items, counts
    uint32_t sym_run = base_s[0], byte_run = base_s[1], dw_run = base_s[2], out_run = base_s[3], mf_run = base_s[4], vb_run = base_s[6];
    for (uint32_t c0 = g0; c0 < g1; c0 += PLAN_THREADS) {
        const uint32_t g = c0 + tid;
        const bool live = g < g1;
        uint32_t v[PLAN_NV], tot[PLAN_NV];
        PlanFrame f = plan_frame(g, live, streams, nstreams, stream_base, chain, detect, vb_blk, v);
        const FxFrame *fp = f.fp;
        plan_scan(v, tot, ws);
        // (work items beyond the arena, padding of the seven code classes included: the frame goes the wave-per-frame way, and
        // so does every frame behind it)
        if (f.batch && (uint64_t)vb_run + v[6] + f.vnb + 7u * 128u > vb_cap) { f.batch = false; f.vnb = 0; }
        const uint32_t sym_off = sym_run + v[0], byte_off = byte_run + v[1], dw_off = dw_run + v[2], out_off = out_run + v[3], mf_off = mf_run + v[4];
        sym_run += tot[0]; byte_run += tot[1]; dw_run += tot[2]; out_run += tot[3]; mf_run += tot[4]; vb_run += tot[6];
        if (live) {
            const FxStreamDesc &sd = streams[f.sidx];
            FxPayJob j;
            j.x = sd.x; j.xa_end = sd.xa_end; j.start = fp->start; j.mix_th = fp->mix_th; j.mix_dl = fp->mix_dl; j.mf_scale = fp->mf_scale;
            j.pfb = fp->pfb; j.mfc0 = fp->mfc0; j.pll_th = fp->pll_th; j.pll_f = fp->pll_f; j.ms = fp->ms; j.bps = f.bps;
            j.nsym = f.valid ? fp->pay_sym_len : 0u; j.sym_off = sym_off;
            j.pay_len = fp->pay_len; j.check = fp->check; j.fec0 = fp->fec0; j.fec1 = fp->fec1; j.k = f.k; j.l0 = f.l0; j.l1 = f.l1;
            j.byte_off = byte_off; j.dw_off = dw_off; j.out_off = out_off; j.pad_ = f.valid ? 1u : 0u;
            j.eq = eq; j.chain_idx = (uint32_t)(fp - chain); j.vb_off = 0; j.vb_nblk = f.vnb;
            pjobs[g] = j;
            // the record goes to pinned host memory: assemble it in registers, send it as eight 16-byte stores
            union { FxOutRec r; uint4 q[sizeof(FxOutRec) / 16]; } u;
            u.r.start = sd.abs_base + fp->start; u.r.stream = f.sidx; u.r.offset = fp->offset;
            u.r.rxy = fp->rxy; u.r.tau = fp->tau; u.r.gamma = fp->gamma; u.r.dphi = fp->dphi; u.r.phi = fp->phi; u.r.pfb = fp->pfb;
            u.r.pilot_dphi = fp->pilot_dphi; u.r.pilot_phi = fp->pilot_phi; u.r.pilot_gain = fp->pilot_gain;
            u.r.flags = fp->flags & FX_FLAG_HEADER_VALID;
            u.r.pay_len = f.valid ? fp->pay_len : 0u; u.r.ms = fp->ms; u.r.check = fp->check; u.r.fec0 = fp->fec0; u.r.fec1 = fp->fec1;
            u.r.nsym = f.valid ? fp->pay_sym_len : 0u; u.r.bps = f.bps; u.r.sym_off = sym_off; u.r.out_off = out_off;
            u.r.evm_sum = 0.0f; u.r.payload_valid = 0; u.r.status = 0;
            const uint32_t *hw = reinterpret_cast<const uint32_t *>(fp->header); uint32_t *rw = reinterpret_cast<uint32_t *>(u.r.header);
#pragma unroll
            for (int i = 0; i < FX_HDR_DEC / 4; i++) rw[i] = hw[i];
            u.r.byte_off = byte_off;
            uint4 *dst = reinterpret_cast<uint4 *>(recs + g);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(FxOutRec) / 16); i++) dst[i] = u.q[i];
            if (f.valid) {
                for (uint32_t c = 0; c < f.nblk; c++) if (mf_off + c < mf_cap) { mf_job[mf_off + c] = g; mf_c0[mf_off + c] = c * 1024u; }
            }
        }
        {   // counts per list (wave-aggregated: one atomic per wave and list)
            const bool val = live && f.valid;
            const uint32_t ms = val ? fp->ms : 0u, fec0 = val ? fp->fec0 : 0u, fec1 = val ? fp->fec1 : 0u;
            (void)wave_agg_add(pw + PW_CLS_CNT, pll_class(ms), 1u, val);
            (void)wave_agg_add(pw + PW_DEC_CNT, f.batch ? 2u : ((fec0 == FX_FEC_RS_M8 || fec1 == FX_FEC_RS_M8) ? 1u : 0u), 1u, val);
            (void)wave_agg_add(pw + PW_VBC_CNT, val && f.batch ? (uint32_t)conv_p(fec0) - 1u : 0u, f.vnb, val && f.batch);
        }
        __syncthreads();
    }
}

extern "C" __global__ __launch_bounds__(PLAN_THREADS)
void fx_planlists_kernel(uint32_t nstreams, uint32_t vb_blk, uint32_t *stream_base, FxPayJob *pjobs, uint32_t mf_cap, uint32_t *pll_list, uint32_t *dec_list,
                         uint32_t list_cap, uint32_t *vb_items, uint32_t vb_cap, FxBlockHdr *hdr, FxBlockHdr *hdr_pay, FxBlockHdr *hdr_host, uint32_t *pw)
{
    __shared__ uint32_t cls_base[FX_PLL_CLASSES + 1], vbc_base[8], last;
    const int tid = threadIdx.x;
    const uint32_t N = stream_base[nstreams];
    if (tid == 0) {
        uint32_t b = 0;
        for (int c = 0; c < FX_PLL_CLASSES; c++) { cls_base[c] = b; b += (pw[PW_CLS_CNT + c] + 63u) & ~63u; }   // whole waves
        cls_base[FX_PLL_CLASSES] = b;
        uint32_t vb = 0;
        for (int c = 0; c < 7; c++) { vbc_base[c] = vb; vb += (pw[PW_VBC_CNT + c] + 127u) & ~127u; }            // whole forward-pass waves: 128 slots
        vbc_base[7] = vb;
    }
    __syncthreads();
    uint32_t g0, g1; plan_range(N, g0, g1);
    for (uint32_t c0 = g0; c0 < g1; c0 += PLAN_THREADS) {                     // (whole chunks: the slot cursors are advanced wave by wave)
        const uint32_t g = c0 + tid;
        const bool live = g < g1;
        uint32_t ms = 0, fec0 = 0, fec1 = 0, vnb = 0; bool val = false;
        if (live) { const FxPayJob &j = pjobs[g]; val = j.pad_ != 0; ms = j.ms; fec0 = j.fec0; fec1 = j.fec1; vnb = j.vb_nblk; }
        const unsigned c = pll_class(ms);
        const uint32_t pp = cls_base[c] + wave_agg_add(pw + PW_CLS_FILL, c, 1u, val);
        if (val && pp < list_cap) pll_list[pp] = g;
        const uint32_t rs = vnb ? 2u : ((fec0 == FX_FEC_RS_M8 || fec1 == FX_FEC_RS_M8) ? 1u : 0u);
        const uint32_t dp = wave_agg_add(pw + PW_DEC_FILL, rs, 1u, val);
        if (val && dp < list_cap) dec_list[(size_t)rs * list_cap + dp] = g;
        const bool bt = val && vnb;                                             // its trellis blocks, among those of the same code
        const uint32_t vc = bt ? (uint32_t)conv_p(fec0) - 1u : 0u;
        const uint32_t at = vbc_base[vc] + wave_agg_add(pw + PW_VBC_FILL, vc, vnb, bt);
        if (bt) {
            for (uint32_t b = 0; b < vnb; b++) if (at + b < vb_cap) { vb_items[at + b] = g; vb_items[vb_cap + at + b] = b; }
            pjobs[g].vb_off = at;
        }
    }
    if (blockIdx.x == 0) {                                                      // padding slots of every class
        for (int c = 0; c < FX_PLL_CLASSES; c++)
            for (uint32_t i = cls_base[c] + pw[PW_CLS_CNT + c] + tid; i < cls_base[c + 1] && i < list_cap; i += PLAN_THREADS) pll_list[i] = 0xFFFFFFFFu;
        for (int c = 0; c < 7; c++)
            for (uint32_t i = vbc_base[c] + pw[PW_VBC_CNT + c] + tid; i < vbc_base[c + 1] && i < vb_cap; i += PLAN_THREADS) vb_items[i] = 0xFFFFFFFFu;
    }
    // the workgroup that finishes last: block header for the payload kernels and for the host; counters, workspace and the
    // walk-phase header are zeroed for the slot's next block
    __threadfence(); __syncthreads();
    if (tid == 0) last = atomicAdd(&pw[PW_TICKET], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (tid == 0) {
        const volatile uint32_t *tot = pw + PW_INC + 8 * (gridDim.x - 1);
        FxBlockHdr h = *hdr;
        h.n_frames = N; h.n_pjobs = tot[5]; h.n_mfblk = min(tot[4], mf_cap);
        h.n_dec_plain = pw[PW_DEC_CNT + 0]; h.n_dec_rs = pw[PW_DEC_CNT + 1]; h.n_dec_batch = pw[PW_DEC_CNT + 2];
        h.n_vb_items = min(vbc_base[7], vb_cap); h.vb_blk = vb_blk; h.vb_want = tot[6] + 7u * 128u; h.n_vb_fallback = 0;
        for (int c = 0; c < FX_PLL_CLASSES; c++) { h.pll_cnt[c] = pw[PW_CLS_CNT + c]; h.pll_base[c] = cls_base[c]; }
        h.pll_base[FX_PLL_CLASSES] = cls_base[FX_PLL_CLASSES];
        h.sym_total = tot[0]; h.byte_total = tot[1]; h.dw_total = tot[2]; h.out_total = tot[3];
        h.done = 1;
        *hdr_pay = h;
        *hdr_host = h;
    }
    __syncthreads();
    {
        uint32_t *z = reinterpret_cast<uint32_t *>(hdr);
        for (int i = tid; i < (int)(sizeof(FxBlockHdr) / 4); i += PLAN_THREADS) z[i] = 0u;
        for (int i = tid; i < PW_WORDS; i += PLAN_THREADS) pw[i] = 0u;
    }
}

extern "C" hipError_t fx_launch_plan(hipStream_t st, unsigned grid, const FxStreamDesc *streams, uint32_t nstreams, uint32_t detect, uint32_t eq, uint32_t vb_blk, const FxFrame *chain,
                                     const uint32_t *chain_count, uint32_t *stream_base, FxPayJob *pjobs, FxOutRec *recs, uint32_t *mf_job, uint32_t *mf_c0, uint32_t mf_cap,
                                     uint32_t *pll_list, uint32_t *dec_list, uint32_t list_cap, uint32_t *vb_items, uint32_t vb_cap, FxBlockHdr *hdr, FxBlockHdr *hdr_pay,
                                     FxBlockHdr *hdr_host, uint32_t *plan_ws)
{
    const unsigned g = grid < 1u ? 1u : (grid > PLAN_MAXG ? PLAN_MAXG : grid);
    hipLaunchKernelGGL(fx_plan_kernel, dim3(g), dim3(PLAN_THREADS), 0, st, streams, nstreams, detect, eq, vb_blk, chain, chain_count, stream_base, pjobs, recs, mf_job, mf_c0,
                       mf_cap, vb_cap, plan_ws);
    hipLaunchKernelGGL(fx_planlists_kernel, dim3(g), dim3(PLAN_THREADS), 0, st, nstreams, vb_blk, stream_base, pjobs, mf_cap, pll_list, dec_list, list_cap, vb_items, vb_cap,
                       hdr, hdr_pay, hdr_host, plan_ws);
    return hipGetLastError();
}
extern "C" unsigned fx_plan_ws_words(void) { return PW_WORDS; }

// ===================================================================== payload: mix + polyphase MF
#define PMF_THREADS 256
#define PMF_SYMS    1024          // symbols per workgroup
#define PMF_SPAN    (2 * PMF_SYMS + FX_MF_TAPS)

// Work items (frame, first symbol) come from fx_plan_kernel; their number is only known on the device, so the grid is
// sized from the host's estimate and strides over the list.  EQ: the equaliser stage is on -- the matched filter is
// evaluated at every sample of the item's span and the frame's 13 trained taps combine 13 of its outputs per symbol.
template <bool EQ>
__global__ __launch_bounds__(PMF_THREADS)
void fx_paymf_kernel(const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr, const FxFrame *chain, float2 *sym_raw,
                     const FxTables *T)
{
    constexpr int LEAD = EQ ? FX_EQ_TAPS - 1 : 0;
    __shared__ float2 v[PMF_SPAN + LEAD + 4];
    __shared__ float2 u[EQ ? 2 * PMF_SYMS + FX_EQ_TAPS : 1];
    __shared__ float2 eqw[16];
    __shared__ float taps[FX_MF_TAPS];
    const uint32_t nitems = hdr->n_mfblk;
    const int tid = threadIdx.x;
    const float2 *sc = T->sc;
    for (uint32_t bi = blockIdx.x; bi < nitems; bi += gridDim.x) {
        const FxPayJob job = jobs[blk_job[bi]];
        const uint32_t c0 = blk_c0[bi];                            // first payload symbol of this item
        const uint32_t ns = min((uint32_t)PMF_SYMS, job.nsym - c0);
        const int64_t sym0 = (int64_t)FX_SYM0_PAY + (EQ ? FX_EQ_DELAY : 0);
        const int64_t nlo = sym_sample(sym0 + c0, job.mfc0) - LEAD - (FX_MF_TAPS - 1);
        const int64_t nhi = sym_sample(sym0 + c0 + ns - 1, job.mfc0);
        const int span = (int)(nhi - nlo + 1);
        const XSrc xs = make_xsrc(job.x, job.xa_end, 0);
        __syncthreads();
        if (tid < FX_MF_TAPS) taps[tid] = T->proto[job.pfb + FX_NPFB * tid];
        if (EQ && tid < 16) eqw[tid] = tid < FX_EQ_TAPS ? chain[job.chain_idx].eq[tid] : make_float2(0.0f, 0.0f);
        for (int m = tid; m < span; m += PMF_THREADS) {
            const int64_t nn = nlo + m;                            // (nn >= 0: a payload symbol is hundreds of samples into the frame)
            v[m] = derot(xld(xs, job.start + nn), job.mix_th + job.mix_dl * (uint32_t)nn, sc);
        }
        __syncthreads();
        if (EQ) {
            const int nu = span - (FX_MF_TAPS - 1);                // matched-filter outputs at samples nlo + 27 ... nhi
            for (int m = tid; m < nu; m += PMF_THREADS) {
                float ar = 0.0f, ai = 0.0f;
#pragma unroll 7
                for (int t = 0; t < FX_MF_TAPS; t++) {
                    const float2 w = v[m + (FX_MF_TAPS - 1) - t]; const float h = taps[t];
                    ar = fmaf(h, w.x, ar); ai = fmaf(h, w.y, ai);
                }
                u[m] = make_float2(ar * job.mf_scale, ai * job.mf_scale);
            }
            __syncthreads();
            for (uint32_t i = tid; i < ns; i += PMF_THREADS) {
                const int nc = (int)(sym_sample(sym0 + c0 + i, job.mfc0) - nlo) - (FX_MF_TAPS - 1);   // index of the symbol's sample in u[]
                sym_raw[(size_t)job.sym_off + c0 + i] = eq_sum16(u + nc - (FX_EQ_TAPS - 1), eqw);
            }
        } else {
            for (uint32_t i = tid; i < ns; i += PMF_THREADS) {
                const int nc = (int)(sym_sample(sym0 + c0 + i, job.mfc0) - nlo);
                float ar = 0.0f, ai = 0.0f;
#pragma unroll 7
                for (int t = 0; t < FX_MF_TAPS; t++) {
                    const float2 w = v[nc - t]; const float h = taps[t];
                    ar = fmaf(h, w.x, ar); ai = fmaf(h, w.y, ai);
                }
                sym_raw[(size_t)job.sym_off + c0 + i] = make_float2(ar * job.mf_scale, ai * job.mf_scale);
            }
        }
    }
}

// The matched filter proper (no equaliser): four consecutive symbols per thread.  A symbol is 28 taps over samples two apart from
// its neighbour's, so four symbols share all but six of their samples: 34 LDS reads instead of 112, and the kernel -- which was
// bound by exactly those reads (56 two-way-conflicted ds_read_b64 per symbol) -- becomes a stream of the IQ through the mixer.
// Layout that makes the register tile conflict-free: with the item's first symbol at span sample 27, symbol i is
//     y[i] = sum_k h[2k] vo[13 + i - k] + h[2k+1] ve[13 + i - k],     ve[j] = v[2j], vo[j] = v[2j+1]
// (summed over the taps in ascending order, as everywhere); thread t, symbols 4t..4t+3, needs ve/vo[4t .. 4t+16].  Both are
// kept de-interleaved by four -- ve[j] at VE[j & 3][j >> 2] -- so that every load of the tile is "phase array p, entry
// t + const": consecutive lanes, consecutive 8-byte words.  The eight phase arrays sit PMF4_STRIDE entries apart, chosen so
// that the 16 lanes of a staging store (8 arrays x 2 entries) fall into 32 different banks.
#define PMF4_STRIDE 274
extern "C" __global__ __launch_bounds__(PMF_THREADS)
void fx_paymf4_kernel(const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr, float2 *sym_raw, const FxTables *T)
{
    __shared__ float2 ph[8 * PMF4_STRIDE];                          // [parity * 4 + phase][entry]
    __shared__ __attribute__((aligned(16))) float taps[FX_MF_TAPS];
    static_assert(FX_MF_TAPS == 28 && PMF_SYMS == 4 * PMF_THREADS, "tile below: 28 taps, four symbols per thread");
    static_assert(PMF4_STRIDE >= (PMF_SPAN + 8) / 8 + 5, "phase arrays hold the span plus the slack phantom symbols read");
    const uint32_t nitems = hdr->n_mfblk;
    const int tid = threadIdx.x;
    const float2 *sc = T->sc;
    for (uint32_t bi = blockIdx.x; bi < nitems; bi += gridDim.x) {
        const FxPayJob job = jobs[blk_job[bi]];
        const uint32_t c0 = blk_c0[bi];                            // first payload symbol of this item
        const uint32_t ns = min((uint32_t)PMF_SYMS, job.nsym - c0);
        const int64_t nlo = sym_sample((int64_t)FX_SYM0_PAY + c0, job.mfc0) - (FX_MF_TAPS - 1);     // span sample 0; the item's symbol i at span sample 27 + 2 i
        const int span = 2 * (int)ns + (FX_MF_TAPS - 2);
        const float2 *px = job.x, *pa = job.xa_end;
        const int64_t p0 = job.start + nlo;                       // stream index of span sample 0 (>= floor: a payload symbol is hundreds of samples into the frame)
        const uint32_t th0 = job.mix_th + job.mix_dl * (uint32_t)nlo, dl = job.mix_dl;
        __syncthreads();
        if (tid < FX_MF_TAPS) taps[tid] = T->proto[job.pfb + FX_NPFB * tid];
        // the span (at most 2074 samples: nine per thread), every load issued before the first one is needed -- a rolled loop of
        // load / wait / mix / store left one 8-byte load per thread in flight and the kernel waiting on memory latency 2/3 of its time
        constexpr int NLD = (PMF_SPAN + PMF_THREADS - 1) / PMF_THREADS;
        float2 raw[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int m = tid + PMF_THREADS * k;
            const int64_t p = p0 + m;
            raw[k] = m < span ? (p < 0 ? pa[p] : px[p]) : make_float2(0.0f, 0.0f);
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int m = tid + PMF_THREADS * k;
            const int j = m >> 1;
            if (m < span) ph[((m & 1) * 4 + (j & 3)) * PMF4_STRIDE + (j >> 2)] = derot(raw[k], th0 + dl * (uint32_t)m, sc);
        }
        __syncthreads();
        if (4u * (uint32_t)tid < ns) {
            float2 e[17], o[17];
#pragma unroll
            for (int j = 0; j < 17; j++) {
                e[j] = ph[(j & 3) * PMF4_STRIDE + tid + (j >> 2)];
                o[j] = ph[(4 + (j & 3)) * PMF4_STRIDE + tid + (j >> 2)];
            }
            float h[FX_MF_TAPS];
#pragma unroll
            for (int q = 0; q < FX_MF_TAPS / 4; q++) { const float4 t4 = reinterpret_cast<const float4 *>(taps)[q]; h[4 * q] = t4.x; h[4 * q + 1] = t4.y; h[4 * q + 2] = t4.z; h[4 * q + 3] = t4.w; }
            float yr[4], yi[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float ar = 0.0f, ai = 0.0f;
#pragma unroll
                for (int k = 0; k < 14; k++) {                      // taps 2k (odd-sample array), 2k + 1 (even-sample array): ascending tap order
                    ar = fmaf(h[2 * k], o[13 + r - k].x, ar); ai = fmaf(h[2 * k], o[13 + r - k].y, ai);
                    ar = fmaf(h[2 * k + 1], e[13 + r - k].x, ar); ai = fmaf(h[2 * k + 1], e[13 + r - k].y, ai);
                }
                yr[r] = ar * job.mf_scale; yi[r] = ai * job.mf_scale;
            }
            // (sym_off is a multiple of 8 symbols, c0 of 1024: 32-byte aligned; symbols beyond ns land in the padding of the frame's granule)
            float4 *dst = reinterpret_cast<float4 *>(sym_raw + (size_t)job.sym_off + c0 + 4u * (uint32_t)tid);
            dst[0] = make_float4(yr[0], yi[0], yr[1], yi[1]);
            dst[1] = make_float4(yr[2], yi[2], yr[3], yi[3]);
        }
    }
}

extern "C" hipError_t fx_launch_paymf(unsigned grid, int eq, hipStream_t st, const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr,
                                      const FxFrame *chain, float2 *sym_raw, const FxTables *T)
{
    if (grid == 0) return hipSuccess;
    if (eq) hipLaunchKernelGGL(fx_paymf_kernel<true>, dim3(grid), dim3(PMF_THREADS), 0, st, jobs, blk_job, blk_c0, hdr, chain, sym_raw, T);
    else hipLaunchKernelGGL(fx_paymf4_kernel, dim3(grid), dim3(PMF_THREADS), 0, st, jobs, blk_job, blk_c0, hdr, sym_raw, T);
    return hipGetLastError();
}

// ===================================================================== payload: PLL + hard demod (lane per frame)
// The decision-directed loop is a strict symbol-to-symbol recurrence, so a frame is one lane.  What the
// lane can do is keep the recurrence's critical path short: raw symbols are fetched eight at a time
// (one block ahead, 64 contiguous bytes per lane), results leave as 16-byte stores, the sin/cos table sits
// in LDS and the four PSK2/PSK4 constellation points in registers.
#define PLL_THREADS 64
#define PLL_MAX_WAVES 4       // waves per workgroup is a launch-time placement knob (one wave per SIMD of a CU at 4)
#define PLL_BLK     8

__device__ __forceinline__ void pll_load8(const float4 *p, float4 b[4]) { b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = p[3]; }

// One instantiation per modulation scheme: the demodulator is resolved at compile time, so the loop body is
// straight-line code (the host launches one grid per scheme present in the batch, over an index list).
template <int MS>
__device__ __forceinline__ unsigned pll_demod(float2 r, unsigned &prev, const float2 *sc, float2 &xh, float &pe)
{
    if constexpr (MS == FX_MODEM_PSK2 || MS == FX_MODEM_PSK4) {
        // decision regions are the axes' half-planes / quadrants about them; with the exact axis points the rotation
        // r conj(xhat) is a swap / sign flip (bit-identical to the general fma form)
        unsigned idx; float pr, pim;
        if constexpr (MS == FX_MODEM_PSK2) {
            const bool neg = !(r.x > 0.0f);
            idx = neg ? 1u : 0u;
            xh = make_float2(neg ? -1.0f : 1.0f, 0.0f);
            pr = neg ? -r.x : r.x; pim = neg ? -r.y : r.y;
        } else {
            const bool horiz = fabsf(r.x) >= fabsf(r.y);
            const bool neg = horiz ? !(r.x > 0.0f) : !(r.y > 0.0f);
            idx = (horiz ? 0u : 1u) + (neg ? 2u : 0u);
            const float sg = neg ? -1.0f : 1.0f;
            xh = horiz ? make_float2(sg, 0.0f) : make_float2(0.0f, sg);
            // horiz: r * conj(+-1) = +-(x, y);  vertical: r * conj(+-j) = +-(y, -x)
            const float a = horiz ? r.x : r.y, b = horiz ? r.y : -r.x;
            pr = neg ? -a : a; pim = neg ? -b : b;
        }
        (void)pr;
        pe = pim;                                                   // imag(r conj(xhat))
        return gray_enc(idx);
    } else {
        return modem_demod((unsigned)MS, modem_bps((unsigned)MS), r, prev, sc, xh, pe);
    }
}

// One block of eight symbols.  TAIL: the frame's last block, where only `live` symbols count (the others are padding of
// the 8-symbol granule: rotated and stored, but the loop state and the EVM sum stay frozen).
template <int MS, bool TAIL>
__device__ __forceinline__ void pll_block8(const float4 (&cur)[4], unsigned live, uint32_t &th, float &fq, float &evm, unsigned &prev, const float2 *sc,
                                           float4 *out, uint2 *hd)
{
    float rr[2 * PLL_BLK]; unsigned h0 = 0, h1 = 0;
    // the carrier phasor: from the table at the head of every block of eight, turned by the phase increment in between
    // (the table look-up is off the symbol-to-symbol recurrence)
    float wc, ws; sincos_u32(th, sc, wc, ws);
#pragma unroll
    for (int k = 0; k < PLL_BLK; k++) {
        const float4 v4 = cur[k >> 1];
        const float2 y = (k & 1) ? make_float2(v4.z, v4.w) : make_float2(v4.x, v4.y);
        float2 r = make_float2(fmaf(y.x, wc, y.y * ws), fmaf(y.y, wc, -(y.x * ws))), xh; float pe;
        unsigned pv = prev;
        const unsigned s = pll_demod<MS>(r, pv, sc, xh, pe);
        if (!TAIL || (unsigned)k < live) {
            float dr = r.x - xh.x, di = r.y - xh.y;
            evm += fmaf(dr, dr, di * di);
            fq = fmaf(pe, 68356.5248f, fq);                                // alpha = 1e-4 (x 2^32/2pi): frequency
            const float t = phase_step(fmaf(pe, 6835652.5f, fq));          // beta = 1e-2 (x 2^32/2pi): phase, then advance; whole phase units
            th += (uint32_t)(int)t;
            float cd, sd; sincos_small(t, cd, sd);
            const float c2 = fmaf(wc, cd, -(ws * sd)), s2 = fmaf(ws, cd, wc * sd);
            wc = c2; ws = s2;
            prev = pv;
        }
        rr[2 * k] = r.x; rr[2 * k + 1] = r.y;
        if (k < 4) h0 |= s << (8 * k); else h1 |= s << (8 * (k - 4));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = make_float4(rr[4 * k], rr[4 * k + 1], rr[4 * k + 2], rr[4 * k + 3]);
    *hd = make_uint2(h0, h1);
}

template <int MS>
__device__ __forceinline__ void pll_frame(uint32_t f, const FxPayJob *jobs, const float2 *sc, const float2 *sym_raw, float2 *framesyms, uint8_t *hard, FxOutRec *recs)
{
    if (f == 0xFFFFFFFFu) return;                                                  // padding of the class's last wave
    const FxPayJob job = jobs[f];
    const float4 *in = reinterpret_cast<const float4 *>(sym_raw + job.sym_off);      // sym_off is a multiple of 8
    float4 *out = reinterpret_cast<float4 *>(framesyms + job.sym_off);
    uint2 *hd = reinterpret_cast<uint2 *>(hard + job.sym_off);
    // loop state: phase (2^32 = one turn), frequency in the same units per symbol, EVM accumulator
    uint32_t th = job.pll_th; float fq = job.pll_f * 683565248.0f, evm = 0.0f; unsigned prev = 0;
    const unsigned nsym = job.nsym;
    const uint32_t nfull = nsym / PLL_BLK, rem = nsym % PLL_BLK;
    float4 cur[4], nxt[4];
    if (nsym) pll_load8(in, cur);
    for (uint32_t b = 0; b < nfull; b++) {
        if ((b + 1) * PLL_BLK < nsym) pll_load8(in + 4 * (b + 1), nxt);
        pll_block8<MS, false>(cur, PLL_BLK, th, fq, evm, prev, sc, out + 4 * b, hd + b);
#pragma unroll
        for (int k = 0; k < 4; k++) cur[k] = nxt[k];
    }
    if (rem) pll_block8<MS, true>(cur, rem, th, fq, evm, prev, sc, out + 4 * nfull, hd + nfull);
    recs[f].evm_sum = evm;                                                          // straight into the host's result record
}

// One kernel for all modulation schemes: fx_plan_kernel lists the frames per scheme, each list padded to whole waves, so
// a wave runs exactly one instantiation of the loop (the demodulator is resolved at compile time, straight-line code).
// The number of waves is only known on the device: the grid strides over them.
__global__ __launch_bounds__(PLL_THREADS * PLL_MAX_WAVES)
void fx_paypll_kernel(const FxPayJob *jobs, const uint32_t *pll_list, const FxBlockHdr *hdr, const float2 *sym_raw,
                      float2 *framesyms, uint8_t *hard, FxOutRec *recs, const FxTables *T)
{
    const uint32_t nw = hdr->pll_base[FX_PLL_CLASSES] >> 6;
    const uint32_t wpg = blockDim.x >> 6, w0 = blockIdx.x * wpg + (threadIdx.x >> 6);
    if (blockIdx.x * wpg >= nw) return;
    // a frame's PLL is one long dependent chain: when it shares a SIMD with walker / decoder waves of other
    // blocks in flight, let it win the issue arbitration
    __builtin_amdgcn_s_setprio(3);
    __shared__ float2 sc[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) sc[i] = T->sc[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (uint32_t w = w0; w < nw; w += gridDim.x * wpg) {
        uint32_t cls = 0;
        while (cls + 1 < FX_PLL_CLASSES && (w << 6) >= hdr->pll_base[cls + 1]) cls++;
        cls = __builtin_amdgcn_readfirstlane(cls);
        const uint32_t f = pll_list[(w << 6) + lane];
#define FX_PLL_CASE(C, M) case C: pll_frame<M>(f, jobs, sc, sym_raw, framesyms, hard, recs); break;
        switch (cls) {
            FX_PLL_CASE(0, FX_MODEM_PSK2) FX_PLL_CASE(1, FX_MODEM_PSK4) FX_PLL_CASE(2, FX_MODEM_PSK8) FX_PLL_CASE(3, FX_MODEM_PSK16)
            FX_PLL_CASE(4, FX_MODEM_DPSK2) FX_PLL_CASE(5, FX_MODEM_DPSK4) FX_PLL_CASE(6, FX_MODEM_DPSK8) FX_PLL_CASE(7, FX_MODEM_ASK4)
            FX_PLL_CASE(8, FX_MODEM_QAM16) FX_PLL_CASE(9, FX_MODEM_QAM32) FX_PLL_CASE(10, FX_MODEM_QAM64)
            default: pll_frame<FX_MODEM_QPSK>(f, jobs, sc, sym_raw, framesyms, hard, recs); break;
        }
#undef FX_PLL_CASE
    }
}

// host-side launcher
extern "C" hipError_t fx_launch_paypll(unsigned grid_waves, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs, const uint32_t *pll_list, const FxBlockHdr *hdr,
                                       const float2 *sym_raw, float2 *framesyms, uint8_t *hard, FxOutRec *recs, const FxTables *T)
{
    if (grid_waves == 0) return hipSuccess;
    const unsigned w = waves_per_wg < 1u ? 1u : (waves_per_wg > PLL_MAX_WAVES ? PLL_MAX_WAVES : waves_per_wg);
    hipLaunchKernelGGL(fx_paypll_kernel, dim3((grid_waves + w - 1) / w), dim3(PLL_THREADS * w), 0, st, jobs, pll_list, hdr, sym_raw, framesyms, hard, recs, T);
    return hipGetLastError();
}

// ===================================================================== payload: packet decode (wave per frame)
#define DEC_THREADS 64

__device__ __forceinline__ unsigned getbit(const uint8_t *b, uint32_t i) { return (b[i >> 3] >> (7 - (i & 7))) & 1u; }

// gather-permute `nbytes` bytes bit by bit: dst bit i = src bit perm[i] (frame generator: tables built by the host)
__device__ __forceinline__ void permute_bits(const uint8_t *src, uint8_t *dst, const uint32_t *perm, uint32_t nbytes, int lane)
{
    for (uint32_t j = lane; j < nbytes; j += DEC_THREADS) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) v = (v << 1) | getbit(src, perm[8 * j + b]);
        dst[j] = (uint8_t)v;
    }
}

// The packetizer's block interleaver, undone in place by one wave.  The interleaver is four passes of byte-pair swaps
// under the bit masks ff / 0f / 55 / 33; a pass pairs even byte 2i with odd byte 2j+1, where j runs through
// m Np + nn (m = 0..M-1 fastest, nn starting at n/3 and wrapping mod Np) skipping values >= n/2.  Within a pass no byte is
// touched twice, so the lanes take 64 candidates at a time and rank the valid ones with a ballot; de-interleaving applies
// the passes in reverse order.  No table, hence nothing for the host to prepare when a new packet length shows up.
__device__ __forceinline__ void deinterleave_wave(uint8_t *x, uint32_t n, int lane)
{
    if (n < 2) return;
    const uint32_t M = 1u + (uint32_t)floorf(sqrtf((float)n));
    uint32_t N = n / M; while (n >= M * N) N++;
    const uint32_t n2 = n / 2, nn0 = n / 3, step_q = 64u / M, step_m = 64u % M;
    for (int p = 3; p >= 0; p--) {
        const uint32_t Np = N + (p == 0 ? 0u : (p == 1 ? 2u : (p == 2 ? 4u : 8u)));
        const uint32_t mk = p == 0 ? 0xffu : (p == 1 ? 0x0fu : (p == 2 ? 0x55u : 0x33u));
        uint32_t m = (uint32_t)lane % M, q = (uint32_t)lane / M, i_base = 0;
        const uint32_t max_iter = (M * (Np + 1u)) / 64u + 2u;                     // every (m, nn) pair at most once: cannot be reached
        for (uint32_t it = 0; i_base < n2 && it < max_iter; it++) {
            const uint32_t nn = q == 0 ? nn0 : (nn0 + q) % Np;
            const uint32_t j = m * Np + nn;
            const bool valid = j < n2;
            const unsigned long long mask = __ballot(valid);
            const uint32_t i = i_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (valid && i < n2) {
                const uint32_t a = 2u * i, b = 2u * j + 1u;
                const uint32_t va = x[a], vb = x[b];
                x[a] = (uint8_t)((va & ~mk) | (vb & mk)); x[b] = (uint8_t)((va & mk) | (vb & ~mk));
            }
            i_base += (uint32_t)__popcll(mask);
            m += step_m; q += step_q;
            if (m >= M) { m -= M; q++; }
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
    }
}

// the same swaps on an array of soft values, eight bytes per packet byte (MSB first): a masked swap of two 64-bit words
__device__ __forceinline__ void deinterleave_soft_wave(uint8_t *xs, uint32_t n, int lane)
{
    if (n < 2) return;
    unsigned long long *x = reinterpret_cast<unsigned long long *>(xs);
    const uint32_t M = 1u + (uint32_t)floorf(sqrtf((float)n));
    uint32_t N = n / M; while (n >= M * N) N++;
    const uint32_t n2 = n / 2, nn0 = n / 3, step_q = 64u / M, step_m = 64u % M;
    for (int p = 3; p >= 0; p--) {
        const uint32_t Np = N + (p == 0 ? 0u : (p == 1 ? 2u : (p == 2 ? 4u : 8u)));
        // bit mask ff / 0f / 55 / 33 -> the bytes (soft values) of the word that change places
        const unsigned long long mk = p == 0 ? 0xFFFFFFFFFFFFFFFFull : (p == 1 ? 0xFFFFFFFF00000000ull : (p == 2 ? 0xFF00FF00FF00FF00ull : 0xFFFF0000FFFF0000ull));
        uint32_t m = (uint32_t)lane % M, q = (uint32_t)lane / M, i_base = 0;
        const uint32_t max_iter = (M * (Np + 1u)) / 64u + 2u;
        for (uint32_t it = 0; i_base < n2 && it < max_iter; it++) {
            const uint32_t nn = q == 0 ? nn0 : (nn0 + q) % Np;
            const uint32_t j = m * Np + nn;
            const bool valid = j < n2;
            const unsigned long long mask = __ballot(valid);
            const uint32_t i = i_base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            if (valid && i < n2) {
                const uint32_t a = 2u * i, b = 2u * j + 1u;
                const unsigned long long va = x[a], vb = x[b];
                x[a] = (va & ~mk) | (vb & mk); x[b] = (va & mk) | (vb & ~mk);
            }
            i_base += (uint32_t)__popcll(mask);
            m += step_m; q += step_q;
            if (m >= M) { m -= M; q++; }
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
    }
}
// hard decisions of n packet bytes from their soft values (value > 127 = 1)
__device__ __forceinline__ void soft_to_hard_wave(const uint8_t *soft, uint8_t *out, uint32_t n, int lane)
{
    for (uint32_t j = lane; j < n; j += DEC_THREADS) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long *>(soft + 8 * (size_t)j);
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) v = (v << 1) | ((((unsigned)(w >> (8 * b)) & 255u) > 127u) ? 1u : 0u);
        out[j] = (uint8_t)v;
    }
}

// SECDED with nd data bytes per block behind one parity byte (nd = 2, 4, 8), Hsiao columns `col`
__device__ __forceinline__ void secded_decode(const uint8_t *col, uint32_t nd, uint32_t n, const uint8_t *enc, uint8_t *dec, int lane)
{
    const uint32_t nblk = (n + nd - 1) / nd;
    for (uint32_t blk = lane; blk < nblk; blk += DEC_THREADS) {
        const uint32_t nb = min(nd, n - nd * blk);
        uint8_t d[8]; uint8_t par = 0;
        for (uint32_t j = 0; j < 8; j++) d[j] = j < nb ? enc[(nd + 1) * blk + 1 + j] : 0;
        for (uint32_t j = 0; j < 8 * nd; j++) if (d[j >> 3] & (0x80u >> (j & 7))) par ^= col[j];
        const uint8_t syn = (uint8_t)(enc[(nd + 1) * blk] ^ par);
        if (syn != 0 && __popc((unsigned)syn) != 1)
            for (uint32_t j = 0; j < 8 * nd; j++) if (col[j] == syn) { d[j >> 3] ^= (uint8_t)(0x80u >> (j & 7)); break; }
        for (uint32_t j = 0; j < nb; j++) dec[nd * blk + j] = d[j];
    }
}

// Reed-Solomon RS(255,223) block of N = dl + 32 bytes, corrected in place in `r` (a lane-private copy); syndromes,
// Berlekamp-Massey, Chien search, Forney -- the same steps, in the same order, as the CPU side (more than 16 byte
// errors: left as received)
__device__ void rs_decode_block(uint8_t *r, unsigned N, const FxTables *T)
{
    const uint8_t *ex = T->rsexp, *lg = T->rslog;
    auto gm = [&](uint8_t a, uint8_t b) -> uint8_t { return (a && b) ? ex[lg[a] + lg[b]] : (uint8_t)0; };
    auto gd = [&](uint8_t a, uint8_t b) -> uint8_t { return a ? ex[lg[a] + 255 - lg[b]] : (uint8_t)0; };
    uint8_t S[32]; unsigned nz = 0;
    for (unsigned i = 0; i < 32; i++) {
        const uint8_t a = ex[i + 1]; uint8_t sy = 0;
        for (unsigned j = 0; j < N; j++) sy = (uint8_t)(gm(sy, a) ^ r[j]);
        S[i] = sy; nz |= sy;
    }
    if (!nz) return;
    uint8_t Lm[33], B[33], Tm[33];
    for (int i = 0; i < 33; i++) { Lm[i] = 0; B[i] = 0; }
    Lm[0] = B[0] = 1;
    unsigned L = 0, m = 1; uint8_t b = 1;
    for (unsigned k = 0; k < 32; k++) {
        uint8_t d = S[k];
        for (unsigned i = 1; i <= L; i++) d ^= gm(Lm[i], S[k - i]);
        if (d == 0) { m++; continue; }
        for (int i = 0; i < 33; i++) Tm[i] = Lm[i];
        const uint8_t coef = gd(d, b);
        for (unsigned i = 0; i + m <= 32; i++) Lm[i + m] ^= gm(coef, B[i]);
        if (2 * L <= k) { L = k + 1 - L; for (int i = 0; i < 33; i++) B[i] = Tm[i]; b = d; m = 1; } else m++;
    }
    if (L > 16) return;
    uint8_t Om[32];
    for (unsigned i = 0; i < 32; i++) { uint8_t v = 0; for (unsigned j = 0; j <= i && j <= L; j++) v ^= gm(Lm[j], S[i - j]); Om[i] = v; }
    unsigned pos[16], np = 0; uint8_t val[16];
    for (unsigned j = 0; j < N; j++) {
        const unsigned e = (N - 1 - j) % 255, inv = (255 - e) % 255;
        uint8_t v = 0;
        for (unsigned i = 0; i <= L; i++) v ^= gm(Lm[i], ex[(inv * i) % 255]);
        if (v) continue;
        if (np == 16) return;
        uint8_t num = 0, den = 0;
        for (unsigned i = 0; i < 32; i++) num ^= gm(Om[i], ex[(inv * i) % 255]);
        for (unsigned i = 1; i <= L; i += 2) den ^= gm(Lm[i], ex[(inv * (i - 1)) % 255]);
        if (den == 0) return;
        pos[np] = j; val[np] = gd(num, den); np++;
    }
    if (np != L) return;
    for (unsigned i = 0; i < np; i++) r[pos[i]] ^= val[i];
}

__device__ __forceinline__ unsigned take_bits(const uint8_t *b, uint32_t nbytes, uint32_t pos, unsigned nbits)
{
    unsigned v = 0;
    for (unsigned i = 0; i < nbits; i++) { const uint32_t q = pos + i; v = (v << 1) | (q < 8 * nbytes ? getbit(b, q) : 0u); }
    return v;
}

template <bool WITH_RS>
__device__ __forceinline__ void block_fec_decode(unsigned fs, uint32_t n, const uint8_t *enc, uint8_t *dec, const FxTables *T, int lane)
{
    unsigned bk, bn;
    if (fs == FX_FEC_HAMMING84) {
        for (uint32_t j = lane; j < n; j += DEC_THREADS)
            dec[j] = (uint8_t)((T->h84dec[enc[2 * j]] << 4) | T->h84dec[enc[2 * j + 1]]);
    } else if (fs == FX_FEC_SECDED7264) {
        secded_decode(T->sdcol, 8, n, enc, dec, lane);
    } else if (fs == FX_FEC_SECDED3932) {
        secded_decode(T->sd39col, 4, n, enc, dec, lane);
    } else if (fs == FX_FEC_SECDED2216) {
        secded_decode(T->sd22col, 2, n, enc, dec, lane);
    } else if (WITH_RS && fs == FX_FEC_RS_M8) {
        // one lane per 255-byte block; the block is corrected in place in the (mutable) encoded buffer
        uint32_t nb = (n + 222) / 223; if (nb == 0) nb = 1;
        const uint32_t dl = (n + nb - 1) / nb;
        for (uint32_t blk = lane; blk < nb; blk += DEC_THREADS) {
            uint8_t *r = const_cast<uint8_t *>(enc) + (size_t)blk * (dl + 32);
            rs_decode_block(r, dl + 32, T);
            const uint32_t off = blk * dl, len = off < n ? min(n - off, dl) : 0u;
            for (uint32_t j = 0; j < len; j++) dec[off + j] = r[j];
        }
    } else if (blk_spec(fs, bk, bn)) {
        // bit-packed codewords: decode whole output bytes per lane (lcm(k,8)/k codewords make lcm(k,8)/8 bytes)
        const uint32_t el = fec_enc_len(fs, n);
        const unsigned grp_cw = bk == 4 ? 2u : (bk == 8 ? 1u : 2u);          // codewords per group
        const unsigned grp_by = bk == 4 ? 1u : (bk == 8 ? 1u : 3u);          // output bytes per group
        const uint32_t ngrp = (n + grp_by - 1) / grp_by;
        for (uint32_t g = lane; g < ngrp; g += DEC_THREADS) {
            unsigned acc = 0;
            for (unsigned c = 0; c < grp_cw; c++) {
                const uint32_t j = g * grp_cw + c;
                unsigned r = take_bits(enc, el, j * bn, bn), d;
                if (fs == FX_FEC_HAMMING74) d = T->h74dec[r];
                else if (fs == FX_FEC_HAMMING128) d = T->h128dec[r];
                else {
                    const unsigned syn = (T->golenc[(r >> 12) & 0xfff] ^ r) & 0xfff;
                    const uint32_t e = T->golerr[syn];
                    if (e != 0xFFFFFFFFu) r ^= e;
                    d = (r >> 12) & 0xfff;
                }
                acc = (acc << bk) | d;
            }
            for (unsigned b = 0; b < grp_by; b++) {
                const uint32_t o = g * grp_by + b;
                if (o < n) dec[o] = (uint8_t)(acc >> (8 * (grp_by - 1 - b)));
            }
        }
    } else {
        for (uint32_t j = lane; j < n; j += DEC_THREADS) dec[j] = enc[j];
    }
}

// K=7 (0x6d, 0x4f) hard-decision Viterbi, one lane per state, exchange-free of LDS.
//
// A shift-register trellis is a perfect shuffle: state p feeds rotl(p,1) (input bit = p's old MSB) and
// its butterfly partner is p ^ 32.  Keeping state s of time t in lane rotr(s, t mod 6) makes every
// successor land in the lane its predecessor came from, so one add-compare-select step needs exactly one
// XOR-lane exchange, with masks cycling 32,16,8,4,2,1 -- permlane32_swap, permlane16_swap and DPP row
// operations, no ds_bpermute.  Metrics, tie rule (equal -> predecessor with MSB 0) and traceback are the
// plain algorithm's; only the lane numbering rotates.
//   enc: coded bits (punctured stream), dec: n bytes, dw: T = 8n+6 decision words of scratch.
// One add-compare-select step at rotation phase PH.  Metrics are kept doubled (P = 2*metric, even) so that a
// parity tag can break ties and carry the decision.  Every lane forms two keys from its own metric,
//   key of its own branch          K = P + 2*bm + hi            (hi = MSB of the state this lane holds), and
//   key of the branch to the partner's successor   X = P + 2*(nbits - bm) + hi
// (both generator polynomials have their end taps set, so the cross branch expects the complement), ships X to
// the butterfly partner and keeps min(K, X of the partner).  The keys never tie, the smaller wins with "equal
// metrics -> predecessor with MSB 0", and its LSB -- the MSB of the state it came from -- is the decision bit.
//
// Exchange: phases 2-5 are DPP row operations folded into the min.  For phases 0 / 1 (xor 32 / xor 16) the lanes
// with hi = 1 form their two keys in swapped roles (A = X, B = K; their expected code bits are stored
// complemented for that), so that one permlane swap of (A, B) leaves min(A', B') = the wanted minimum in every
// lane -- no copies, no select.
//   tab: !PUNCT: ta[PH] / tb[PH] hold the four possible key increments (byte r = received pair ra | rb << 1);
//        PUNCT:  ta[PH] = expected code bits (bit-doubled, role-adjusted), tb[PH] = 2 * hi.
template <int PH>
__device__ __forceinline__ uint32_t acs_exchange_min(uint32_t A, uint32_t B)
{
    if constexpr (PH == 0) { auto r = __builtin_amdgcn_permlane32_swap(A, B, false, false); return min(r[0], r[1]); }
    else if constexpr (PH == 1) { auto r = __builtin_amdgcn_permlane16_swap(A, B, false, false); return min(r[0], r[1]); }
    else if constexpr (PH == 2) return min(A, (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)B, 0x140, 0xf, 0xf, false), 0x141, 0xf, 0xf, false));
    else if constexpr (PH == 3) return min(A, (uint32_t)__builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp((int)B, 0x141, 0xf, 0xf, false), 0x1B, 0xf, 0xf, false));
    else if constexpr (PH == 4) return min(A, (uint32_t)__builtin_amdgcn_mov_dpp((int)B, 0x4E, 0xf, 0xf, false));
    else return min(A, (uint32_t)__builtin_amdgcn_mov_dpp((int)B, 0xB1, 0xf, 0xf, false));
}

// VM: 0 hard decisions, rate 1/2 (table-driven increments); 1 hard decisions, punctured; 2 soft decisions (any puncturing)
template <int PH, int VM>
__device__ __forceinline__ void acs_step(uint32_t &P, unsigned cu, const unsigned (&ta)[6], const unsigned (&tb)[6], int lane, unsigned &hist)
{
    // cu: this step's received code word, already in a scalar register -- VM 1: R4 | H4<<4 | (2*nbits)<<8; VM 0: 8*r;
    // VM 2: soft value A | soft value B << 8 | (present ? 255 : 0) << 16 | likewise B << 24
    uint32_t A, B;
    if constexpr (VM == 2) {
        // cost of an expected bit e against a soft value v: e ? 255 - v : v = v ^ (e ? 255 : 0); ta[PH] holds the lane's two
        // expected bits as byte masks (role-adjusted), one v_sad_u8 adds the two byte costs
        const unsigned hi = ((unsigned)lane >> (5 - PH)) & 1u;
        const unsigned t = ((cu & 0xffffu) ^ ta[PH]) & (cu >> 16);
        const unsigned a = 2u * __builtin_amdgcn_sad_u8(t, 0u, 0u) + hi;
        const unsigned b = 2u * (((cu >> 16) & 255u) + (cu >> 24)) - a;               // 2 * 255 * bits present - a
        A = P + a;
        B = P + b + tb[PH];
    } else if constexpr (VM == 1) {
        const unsigned hi = ((unsigned)lane >> (5 - PH)) & 1u;
        const unsigned t = (ta[PH] ^ (cu & 15u)) & ((cu >> 4) & 15u);
        const unsigned a = __popc(t) + hi;                                             // 2 * bm (role-adjusted) + hi
        const unsigned b = ((cu >> 8) & 15u) - a;                                      // off the metric's critical path
        A = P + a;
        B = P + b + tb[PH];
    } else {
        A = P + __builtin_amdgcn_ubfe(ta[PH], cu, 8u);
        B = P + __builtin_amdgcn_ubfe(tb[PH], cu, 8u);
    }
    const uint32_t m = acs_exchange_min<PH>(A, B);
    hist = (hist << 1) | (m & 1u);
    P = m & ~1u;
}

// enc: VM 0 / 1: the coded bits, packed; VM 2: one soft byte per coded bit
template <int VM>
__device__ __forceinline__ void viterbi27(int p, uint32_t n, const uint8_t *enc, uint8_t *dec, unsigned long long *dw, uint8_t *scratch, int lane, uint32_t *stamp_fwd)
{
#ifdef FX_STAMPS
    unsigned long long tv0_ = __builtin_readcyclecounter();
#endif
    const uint32_t Tn = 8 * n + 6;
    // per rotation phase: this lane's state, its MSB (hi), the bit-doubled expected code bits of its own branch
    // (A in bits 0-1, B in bits 2-3), role-adjusted for the swap phases -- see acs_step
    unsigned ta[6], tb[6];
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const unsigned st = (((unsigned)lane << ph) | ((unsigned)lane >> (6 - ph))) & 63u;   // state held at phase ph
        const unsigned hi = st >> 5;                                                          // this lane is the MSB-1 side
        const unsigned sr = ((st << 1) | hi) & 0x7f;                                          // own branch: (st) -> rotl(st,1)
        unsigned e = ((__popc(sr & 0x6d) & 1) ? 3u : 0u) | ((__popc(sr & 0x4f) & 1) ? 12u : 0u);
        if (ph < 2 && hi) e ^= 15u;                                                           // swapped roles: A = cross key
        if (VM == 2) { ta[ph] = ((e & 3u) ? 0xffu : 0u) | ((e & 12u) ? 0xff00u : 0u); tb[ph] = 2u * hi; }
        else if (VM == 1) { ta[ph] = e; tb[ph] = 2u * hi; }
        else {
            unsigned wa = 0, wb = 0;
#pragma unroll
            for (unsigned r = 0; r < 4; r++) {
                const unsigned R4 = ((r & 1u) ? 3u : 0u) | ((r & 2u) ? 12u : 0u);
                const unsigned a = __popc(e ^ R4) + hi;
                wa |= a << (8 * r); wb |= (4u + 2u * hi - a) << (8 * r);
            }
            ta[ph] = wa; tb[ph] = wb;
        }
    }
    unsigned pa, pb;                                       // puncturing rows (A, B) as bit masks over the column
    switch (p) {
    case 2: pa = 0x3; pb = 0x1; break;                 // 11 / 10
    case 3: pa = 0x3; pb = 0x5; break;                 // 110 / 101
    case 4: pa = 0xf; pb = 0x1; break;                 // 1111 / 1000
    case 5: pa = 0xb; pb = 0x15; break;                // 11010 / 10101
    case 6: pa = 0x17; pb = 0x29; break;               // 111010 / 100101
    case 7: pa = 0x2f; pb = 0x51; break;               // 1111010 / 1000101
    default: pa = 0x1; pb = 0x1; break;
    }
    // received code word of step t: R4 | H4 << 4 | (2 * bits present) << 8
    auto prep = [&](uint32_t t) -> unsigned {
        if (t >= Tn) return 0u;
        const unsigned col = t % (unsigned)p;
        const unsigned hasA = (pa >> col) & 1, hasB = (pb >> col) & 1;
        uint32_t nb = (p == 1) ? 2 * t : t + (t + (unsigned)p - 1) / (unsigned)p;       // coded bits before step t
        unsigned ra = 0, rb = 0;
        if (VM == 2) {
            if (hasA) { ra = enc[nb]; nb++; }
            if (hasB) { rb = enc[nb]; }
            return ra | (rb << 8) | (hasA * 0xff0000u) | (hasB * 0xff000000u);
        }
        if (hasA) { ra = getbit(enc, nb); nb++; }
        if (hasB) { rb = getbit(enc, nb); }
        if (VM == 0) return 8u * (ra | (rb << 1));
        return (ra * 3u) | (rb * 12u) | ((hasA * 3u | hasB * 12u) << 4) | ((2u * (hasA + hasB)) << 8);
    };
    uint32_t P = (lane == 0) ? 0u : (1u << 25);            // state 0 sits in lane 0 at every phase
    unsigned code_next = prep(lane);
    for (uint32_t t0 = 0; t0 < Tn; t0 += 64) {
        const unsigned code = code_next;
        code_next = prep(t0 + 64 + lane);                  // next chunk's loads fly while this chunk computes
        const uint32_t nstep = min(64u, Tn - t0);
        unsigned hist[2] = { 0u, 0u };
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t ub = 32u * h, ue = min(nstep, ub + 32u);
            uint32_t u = ub;
            // the code words travel lane -> scalar register; fetch a whole group of six before its first step, so that the
            // VALU -> SGPR -> operand latency of the fetch is off the metric chain
            auto cw = [&](uint32_t uu) -> unsigned { return (unsigned)__builtin_amdgcn_readlane((int)code, (int)uu); };
            while (u < ue) {
                switch ((t0 + u) % 6) {
                case 0:
                    if (u + 6 <= ue) {
                        const unsigned c0 = cw(u), c1 = cw(u + 1), c2 = cw(u + 2), c3 = cw(u + 3), c4 = cw(u + 4), c5 = cw(u + 5);
                        acs_step<0, VM>(P, c0, ta, tb, lane, hist[h]); acs_step<1, VM>(P, c1, ta, tb, lane, hist[h]);
                        acs_step<2, VM>(P, c2, ta, tb, lane, hist[h]); acs_step<3, VM>(P, c3, ta, tb, lane, hist[h]);
                        acs_step<4, VM>(P, c4, ta, tb, lane, hist[h]); acs_step<5, VM>(P, c5, ta, tb, lane, hist[h]);
                        u += 6;
                    } else { acs_step<0, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; }
                    break;
                case 1: acs_step<1, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; break;
                case 2: acs_step<2, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; break;
                case 3: acs_step<3, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; break;
                case 4: acs_step<4, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; break;
                default: acs_step<5, VM>(P, cw(u), ta, tb, lane, hist[h]); u++; break;
                }
            }
        }
        // lane-major history: word [chunk][lane] holds this lane's decisions, newest step in bit 0 of each half
        dw[t0 + lane] = ((unsigned long long)hist[1] << 32) | hist[0];
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
#ifdef FX_STAMPS
    if (lane == 0 && stamp_fwd) *stamp_fwd = (uint32_t)(__builtin_readcyclecounter() - tv0_);
#endif
    // ---- traceback, parallel over 64-step chunks with exact verification ----
    // A serial traceback is Tn dependent steps.  Instead every lane traces whole chunks: chunk c is first entered
    // from a *guess* S_c of its end state, obtained by tracing chunk c+1 back from state 0 (survivor paths merge
    // within a few constraint lengths, so the guess is almost always right), and yields the state B_c at its
    // start.  The chain is then verified: the true end state of chunk c is B_{c+1} (0 for the last chunk); any
    // chunk whose guess differs is re-traced from the true state, until nothing changes.  At that fixed point
    // every chunk was entered from the state the serial traceback would have had -- same bits, bit for bit.
    {
        const uint32_t nchunk = (Tn + 63) / 64, Lc = nchunk - 1;
        const uint32_t *dw32 = reinterpret_cast<const uint32_t *>(dw);
        uint8_t *Sarr = scratch, *Barr = scratch + nchunk;          // per-chunk end-state guess / start state
        constexpr int NCH = 4;                                      // chains per lane, interleaved to overlap load latency
        // one pass of up to NCH chunks per lane: st[] in/out, bits out
        auto trace = [&](const uint32_t (&cidx)[NCH], const bool (&act)[NCH], unsigned (&st)[NCH], uint32_t (&bits)[NCH][2]) {
            unsigned r[NCH]; uint32_t ns[NCH];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                const uint32_t t0 = 64u * cidx[j];
                ns[j] = act[j] ? min(64u, Tn - t0) : 0u;
                r[j] = (t0 + ns[j]) % 6u;
                bits[j][0] = bits[j][1] = 0u;
            }
#pragma unroll
            for (int h = 1; h >= 0; h--) {
                for (int uh = 31; uh >= 0; uh--) {
                    uint32_t word[NCH]; bool on[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; j++) {
                        on[j] = (uint32_t)(32 * h + uh) < ns[j];
                        const unsigned ln = ((st[j] >> r[j]) | (st[j] << (6 - r[j]))) & 63u;
                        word[j] = on[j] ? dw32[((size_t)cidx[j] * 64 + ln) * 2 + h] : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < NCH; j++) {
                        if (on[j]) {
                            const uint32_t nh = min(ns[j] - 32u * h, 32u);
                            bits[j][h] |= (st[j] & 1u) << uh;
                            st[j] = (st[j] >> 1) | (((word[j] >> (nh - 1 - uh)) & 1u) << 5);
                            r[j] = r[j] ? r[j] - 1 : 5;
                        }
                    }
                }
            }
        };
        auto emit = [&](uint32_t c, const uint32_t (&b)[2]) {
            // steps 64c .. 64c+63 are output bytes 8c .. 8c+7, MSB first (the buffer has 8 bytes of slack)
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                lo |= (__brev((b[0] >> (8 * k)) & 0xffu) >> 24) << (8 * k);
                hi |= (__brev((b[1] >> (8 * k)) & 0xffu) >> 24) << (8 * k);
            }
            *reinterpret_cast<uint2 *>(dec + 8 * (size_t)c) = make_uint2(lo, hi);
        };
        // pass 1: guess + trace
        for (uint32_t base = 0; base < nchunk; base += 64 * NCH) {
            uint32_t cidx[NCH], cnext[NCH], bits[NCH][2]; bool act[NCH], warm[NCH]; unsigned st[NCH];
#pragma unroll
            for (int j = 0; j < NCH; j++) {
                cidx[j] = base + 64 * j + lane; act[j] = cidx[j] < nchunk;
                cnext[j] = cidx[j] + 1; warm[j] = act[j] && cidx[j] < Lc; st[j] = 0;
            }
            trace(cnext, warm, st, bits);                               // warm-up through chunk c+1 from state 0
            unsigned S[NCH];
#pragma unroll
            for (int j = 0; j < NCH; j++) S[j] = st[j];
            trace(cidx, act, st, bits);
#pragma unroll
            for (int j = 0; j < NCH; j++) if (act[j]) { Sarr[cidx[j]] = (uint8_t)S[j]; Barr[cidx[j]] = (uint8_t)st[j]; emit(cidx[j], bits[j]); }
        }
        // pass 2: verify the chain, re-trace what was entered from a wrong state, repeat to the fixed point
        for (;;) {
            __threadfence_block(); __builtin_amdgcn_wave_barrier();
            bool changed = false;
            for (uint32_t base = 0; base < nchunk; base += 64 * NCH) {
                uint32_t cidx[NCH], bits[NCH][2]; bool redo[NCH]; unsigned st[NCH]; bool any = false;
#pragma unroll
                for (int j = 0; j < NCH; j++) {
                    cidx[j] = base + 64 * j + lane;
                    redo[j] = false; st[j] = 0;
                    if (cidx[j] < nchunk) {
                        const unsigned need = cidx[j] == Lc ? 0u : Barr[cidx[j] + 1];
                        if (Sarr[cidx[j]] != need) { redo[j] = true; st[j] = need; any = true; }
                    }
                }
                if (__any(any)) {
                    unsigned S[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; j++) S[j] = st[j];
                    trace(cidx, redo, st, bits);
#pragma unroll
                    for (int j = 0; j < NCH; j++) if (redo[j]) { Sarr[cidx[j]] = (uint8_t)S[j]; Barr[cidx[j]] = (uint8_t)st[j]; emit(cidx[j], bits[j]); }
                    changed = true;
                }
            }
            if (!__any(changed)) break;
        }
    }
}

// CRC over n bytes by one wavefront.  The register update is linear over GF(2): every lane runs the bitwise
// CRC over its own contiguous piece from a zero register (lane 0: from the all-ones preset), and the pieces are
// combined Horner-style, acc <- M_s acc ^ reg_i, where M_s advances a register through s zero bytes; lane j
// holds column j of M_s.  Same key as the serial reflected algorithm, ~60x fewer dependent steps.
__device__ __forceinline__ uint32_t crc_wave(uint32_t poly_rev, uint32_t mask, const uint8_t *msg, uint32_t n, int lane)
{
    const uint32_t s = n / 64, first = n - 63 * s;            // lane 0 takes the first `first` bytes, the others s each
    const uint32_t start = lane == 0 ? 0u : first + (uint32_t)(lane - 1) * s, len = lane == 0 ? first : s;
    uint32_t reg = lane == 0 ? mask : 0u;
    for (uint32_t i = 0; i < len; i++) {
        reg ^= msg[start + i];
#pragma unroll
        for (int j = 0; j < 8; j++) reg = (reg >> 1) ^ (poly_rev & (0u - (reg & 1u)));
    }
    uint32_t acc = (uint32_t)__builtin_amdgcn_readlane((int)reg, 0);
    if (s) {
        uint32_t col = 1u << (lane & 31);
        for (uint32_t i = 0; i < 8 * s; i++) col = (col >> 1) ^ (poly_rev & (0u - (col & 1u)));
        for (int i = 1; i < 64; i++) {
            uint32_t v = (lane < 32 && ((acc >> lane) & 1u)) ? col : 0u;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) v ^= (uint32_t)__shfl_xor((int)v, m, 64);
            acc = (uint32_t)__builtin_amdgcn_readlane((int)v, 0) ^ (uint32_t)__builtin_amdgcn_readlane((int)reg, i);
        }
    }
    return (~acc) & mask;
}

#ifdef FX_STAMPS
#define FX_STAMP(i) do { unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0 && res) res[jf].stamp[i] = (uint32_t)(t_ - t_prev); t_prev = t_; } while (0)
#define FX_STAMP_INIT unsigned long long t_prev = __builtin_readcyclecounter()
#else
#define FX_STAMP(i) do { } while (0)
#define FX_STAMP_INIT do { } while (0)
#endif

// WITH_RS: the Reed-Solomon decoder keeps ~150 bytes of per-lane state; frames that use it go through their own
// instance so that every other frame keeps the lean (39-VGPR, 8 waves/SIMD) one.  job_idx lists this grid's frames.
//
// Waves never talk to each other (no LDS, no barrier), so the workgroup size is only a placement knob: wide
// workgroups keep a block's decode waves together on few CUs instead of sprinkling one wave over every CU, which
// matters to the walker of the next block (its workgroups need a whole, empty register file each).
#define DEC_MAX_WAVES 8
// de-whiten, CRC, copy out (the payload goes straight into the host's result arena); A: the k decoded bytes
__device__ __forceinline__ void dec_tail(const FxPayJob &job, uint32_t jf, uint8_t *A, int lane, uint8_t *out, FxOutRec *recs, uint32_t status)
{
    const uint8_t mask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 };
    for (uint32_t j = lane; j < job.k; j += DEC_THREADS) A[j] ^= mask[j & 3];
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; 4 * j < job.pay_len; j += DEC_THREADS) {                // out_off is a multiple of 16
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) if (4 * j + b < job.pay_len) w |= (uint32_t)A[4 * j + b] << (8 * b);
        reinterpret_cast<uint32_t *>(out + job.out_off)[j] = w;
    }
    const uint32_t cl = job.k - job.pay_len;
    uint32_t rx = 0, key = 0;
    for (uint32_t i = 0; i < cl; i++) rx = (rx << 8) | A[job.pay_len + i];
    switch (job.check) {
    case FX_CRC_CHECKSUM: {
        uint32_t sm = 0;
        for (uint32_t i = lane; i < job.pay_len; i += DEC_THREADS) sm += A[i];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sm += (uint32_t)__shfl_xor((int)sm, m, 64);
        key = (~sm + 1u) & 0xff; break; }
    case FX_CRC_8:  key = crc_wave(0xE0u, 0xFFu, A, job.pay_len, lane); break;
    case FX_CRC_16: key = crc_wave(0xA001u, 0xFFFFu, A, job.pay_len, lane); break;
    case FX_CRC_24: key = crc_wave(0xD3B6BAu, 0xFFFFFFu, A, job.pay_len, lane); break;
    case FX_CRC_32: key = crc_wave(0xEDB88320u, 0xFFFFFFFFu, A, job.pay_len, lane); break;
    default: key = 0; break;
    }
    if (lane == 0) {
        recs[jf].payload_valid = (key == rx) ? 1u : 0u;
        recs[jf].status = status;
    }
}

// one frame, one wave.  SOFT: decoding from per-bit soft values (fx_softdemod_kernel wrote them, 8 per packet byte):
// a convolutional stage decodes from soft values as long as nothing before it took hard decisions -- the stage nearest the
// channel (fec1), and fec0 too when fec1 is FEC_NONE; every other stage takes hard decisions (value > 127).
// de-interleave n bytes at buf through the wave's LDS buffer X when they fit (see fx_vbpre_kernel: the passes are dependent byte
// swaps, slow through global memory), else in place
#define DEC_LDS 4608
__device__ __forceinline__ void deinterleave_staged(uint8_t *buf, uint32_t n, uint8_t *X, int lane)
{
    if (!X || n + 16u > DEC_LDS) { deinterleave_wave(buf, n, lane); return; }
    const uint32_t n16 = (n + 15u) / 16u;                                       // (buffers are 16-byte aligned with slack behind the packet)
    for (uint32_t j = lane; j < n16; j += DEC_THREADS) reinterpret_cast<uint4 *>(X)[j] = reinterpret_cast<const uint4 *>(buf)[j];
    __builtin_amdgcn_wave_barrier();
    deinterleave_wave(X, n, lane);
    for (uint32_t j = lane; j < n16; j += DEC_THREADS) reinterpret_cast<uint4 *>(buf)[j] = reinterpret_cast<const uint4 *>(X)[j];
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
}

template <bool WITH_RS, bool SOFT>
__device__ __forceinline__ void dec_frame(uint32_t ji, int lane, const FxPayJob *jobs, const uint32_t *job_idx, const uint8_t *hard, uint8_t *bufA,
                                          uint8_t *bufB, uint8_t *soft_arena, unsigned long long *dw_arena, uint8_t *out, FxOutRec *recs, FxPayResult *res,
                                          const FxTables *T, uint8_t *X)
{
    const uint32_t jf = job_idx[ji];
    FxPayJob job = jobs[jf];
    // one wave per frame: pin the loop bounds into SGPRs so that every loop below is scalar-controlled
    job.l0 = __builtin_amdgcn_readfirstlane(job.l0); job.l1 = __builtin_amdgcn_readfirstlane(job.l1);
    job.k = __builtin_amdgcn_readfirstlane(job.k); job.pay_len = __builtin_amdgcn_readfirstlane(job.pay_len);
    job.fec0 = __builtin_amdgcn_readfirstlane(job.fec0); job.fec1 = __builtin_amdgcn_readfirstlane(job.fec1);
    job.bps = __builtin_amdgcn_readfirstlane(job.bps); job.check = __builtin_amdgcn_readfirstlane(job.check);
    uint8_t *A = bufA + job.byte_off, *B = bufB + job.byte_off;
    const uint8_t *hs = hard + job.sym_off;
    const unsigned bps = job.bps;
    FX_STAMP_INIT;
    uint32_t status = 0;
    const int pc1 = conv_p(job.fec1), pc0 = conv_p(job.fec0);
    if constexpr (SOFT) {
        uint8_t *S = soft_arena + 8 * (size_t)job.byte_off;
        deinterleave_soft_wave(S, job.l1, lane);
        FX_STAMP(1);
        bool still_soft = false;
        if (pc1) viterbi27<2>(pc1, job.l0, S, B, dw_arena + job.dw_off, A, lane, nullptr);
        else if (job.fec1 == FX_FEC_NONE) still_soft = true;
        else {
            soft_to_hard_wave(S, A, job.l1, lane);
            __threadfence_block(); __builtin_amdgcn_wave_barrier();
            block_fec_decode<WITH_RS>(job.fec1, job.l0, A, B, T, lane);
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        FX_STAMP(2);
        if (still_soft) {
            deinterleave_soft_wave(S, job.l0, lane);
            FX_STAMP(3);
            if (pc0) viterbi27<2>(pc0, job.k, S, A, dw_arena + job.dw_off, B, lane, res ? &res[jf].stamp[6] : nullptr);
            else {
                soft_to_hard_wave(S, B, job.l0, lane);
                __threadfence_block(); __builtin_amdgcn_wave_barrier();
                block_fec_decode<WITH_RS>(job.fec0, job.k, B, A, T, lane);
            }
        } else {
            deinterleave_staged(B, job.l0, X, lane);
            FX_STAMP(3);
            if (pc0 == 1) viterbi27<0>(1, job.k, B, A, dw_arena + job.dw_off, B, lane, res ? &res[jf].stamp[6] : nullptr);
            else if (pc0) viterbi27<1>(pc0, job.k, B, A, dw_arena + job.dw_off, B, lane, res ? &res[jf].stamp[6] : nullptr);
            else block_fec_decode<WITH_RS>(job.fec0, job.k, B, A, T, lane);
        }
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        FX_STAMP(4);
    } else {
    // 1. hard symbols -> packet bytes (MSB first)
    for (uint32_t j = lane; j < job.l1; j += DEC_THREADS) {
        unsigned v = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint32_t k = 8 * j + b, sidx = k / bps, sb = bps - 1 - (k % bps);
            v = (v << 1) | ((hs[sidx] >> sb) & 1u);
        }
        A[j] = (uint8_t)v;
    }
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    FX_STAMP(0);
    // 2. outer plan (fec1): de-interleave l1 bytes in place, decode -> l0 bytes
    deinterleave_staged(A, job.l1, X, lane);
    FX_STAMP(1);
    if (pc1 == 1) viterbi27<0>(1, job.l0, A, B, dw_arena + job.dw_off, A, lane, nullptr);
    else if (pc1) viterbi27<1>(pc1, job.l0, A, B, dw_arena + job.dw_off, A, lane, nullptr);
    else block_fec_decode<WITH_RS>(job.fec1, job.l0, A, B, T, lane);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    FX_STAMP(2);
    // 3. inner plan (fec0): de-interleave l0 bytes in place, decode -> k bytes
    deinterleave_staged(B, job.l0, X, lane);
    FX_STAMP(3);
    if (pc0 == 1) viterbi27<0>(1, job.k, B, A, dw_arena + job.dw_off, B, lane, res ? &res[jf].stamp[6] : nullptr);
    else if (pc0) viterbi27<1>(pc0, job.k, B, A, dw_arena + job.dw_off, B, lane, res ? &res[jf].stamp[6] : nullptr);
    else block_fec_decode<WITH_RS>(job.fec0, job.k, B, A, T, lane);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    FX_STAMP(4);
    }
    dec_tail(job, jf, A, lane, out, recs, status);
    FX_STAMP(5);
}

template <bool WITH_RS, bool SOFT>
__global__ __launch_bounds__(WITH_RS ? DEC_THREADS : DEC_THREADS * DEC_MAX_WAVES)
void fx_paydec_kernel(const FxPayJob *jobs, const uint32_t *job_idx, const FxBlockHdr *hdr, uint32_t first_wave, const uint8_t *hard, uint8_t *bufA,
                      uint8_t *bufB, uint8_t *soft_arena, unsigned long long *dw_arena, uint8_t *out, FxOutRec *recs, FxPayResult *res, const FxTables *T,
                      FxBlockHdr *fallback_host)
{
    // The list's length is known on the device only.  The lean instance has no loop around its body (a grid-stride loop
    // doubles the register footprint): the host launches it over the list's capacity -- a grid sized from the previous block
    // plus a second, normally idle one for the rest -- and surplus waves leave at once.  The Reed-Solomon instance strides.
    // (fallback_host set: the list is that of the frames the batch Viterbi path handed back)
    __shared__ __attribute__((aligned(16))) uint8_t Xs[WITH_RS ? 16 : DEC_LDS];   // (the lean instance, one wave per workgroup: staging for the de-interleaver)
    uint8_t *X = (!WITH_RS && blockDim.x == 64) ? Xs : nullptr;
    const uint32_t njobs = WITH_RS ? hdr->n_dec_rs : (fallback_host ? hdr->n_vb_fallback : hdr->n_dec_plain);
    const uint32_t wpg = blockDim.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t ji0 = __builtin_amdgcn_readfirstlane(first_wave + blockIdx.x * wpg + (threadIdx.x >> 6));
    if (ji0 >= njobs) return;
    __builtin_amdgcn_s_setprio(2);
    if constexpr (WITH_RS) {
        for (uint32_t ji = ji0; ji < njobs; ji += gridDim.x * wpg) dec_frame<true, SOFT>(ji, lane, jobs, job_idx, hard, bufA, bufB, soft_arena, dw_arena, out, recs, res, T, X);
    } else {
        dec_frame<false, SOFT>(ji0, lane, jobs, job_idx, hard, bufA, bufB, soft_arena, dw_arena, out, recs, res, T, X);
    }
}

extern "C" hipError_t fx_launch_paydec(int with_rs, int soft, unsigned first_wave, unsigned grid_waves, unsigned waves_per_wg, hipStream_t st, const FxPayJob *jobs,
                                       const uint32_t *job_idx, const FxBlockHdr *hdr, const uint8_t *hard, uint8_t *bufA, uint8_t *bufB, uint8_t *soft_arena,
                                       unsigned long long *dw_arena, uint8_t *out, FxOutRec *recs, FxPayResult *res, const FxTables *T, FxBlockHdr *fallback_host)
{
    if (grid_waves == 0) return hipSuccess;
    const unsigned w = with_rs ? 1u : (waves_per_wg < 1u ? 1u : (waves_per_wg > DEC_MAX_WAVES ? DEC_MAX_WAVES : waves_per_wg));
    const dim3 grid((grid_waves + w - 1) / w), block(DEC_THREADS * w);
#define FX_DEC_LAUNCH(RS, SF) hipLaunchKernelGGL((fx_paydec_kernel<RS, SF>), grid, block, 0, st, jobs, job_idx, hdr, first_wave, hard, bufA, bufB, soft_arena, dw_arena, out, recs, res, T, fallback_host)
    if (with_rs) { if (soft) FX_DEC_LAUNCH(true, true); else FX_DEC_LAUNCH(true, false); }
    else { if (soft) FX_DEC_LAUNCH(false, true); else FX_DEC_LAUNCH(false, false); }
#undef FX_DEC_LAUNCH
    return hipGetLastError();
}

// ===================================================================== payload: batch Viterbi (lane per trellis block)
// The wave-per-frame decoder above keeps one trellis state per lane: every add-compare-select step pays a lane exchange and
// a dependent chain, ~13 instructions for one trellis step of one frame.  Here a LANE runs the whole 64-state trellis of a
// stretch of one frame -- 64 path metrics in registers, 32 butterflies of straight-line VALU code per step, no lane exchange
// -- so one instruction serves 64 stretches: ~5 instructions per state and step, and nothing in a step depends on anything
// but the previous step.
//
// Enough lanes come from cutting every frame's trellis into blocks of `blk` steps that run in parallel.  A block other than
// the first does not know its start metrics; it runs vb_warm(p) (64 to 128) steps of warm-up from all-equal metrics first.  Survivor
// paths merge within a few constraint lengths, after which the metric DIFFERENCES -- all that add-compare-select decisions
// depend on -- are the true ones.  That is not assumed but verified: every block records its metric differences at the start
// of its region and at its end; fx_vbpost_kernel checks block b's start against block b-1's end and, on a mismatch, runs
// block b again from the true differences.  Decisions are then exactly the sequential decoder's, for any block size.
//
// Frames that take this path (fx_plan_kernel): hard decisions, fec0 a K = 7 convolutional code (any puncturing), fec1 not.
//   fx_vbpre_kernel   wave per frame   symbols -> packet bytes, de-interleave, fec1, de-interleave: the coded bits
//   fx_vbfwd_kernel   lane per block   forward pass, one 64-bit decision word per trellis step
//   fx_vbpost_kernel  wave per frame   verification (+ repair), chunk-parallel traceback, de-whitening, CRC, payload out
__device__ __forceinline__ constexpr unsigned vb_expect(unsigned j)      // expected code bits (A | B << 1) of the branch j -> 2j
{
    const unsigned sr = (j << 1) & 0x7fu;
    return ((unsigned)__builtin_popcount(sr & 0x6du) & 1u) | (((unsigned)__builtin_popcount(sr & 0x4fu) & 1u) << 1);
}

struct VbState {
    uint32_t P[64];            // path metrics
};

// metric differences to the smallest, as bytes (a K = 7 trellis keeps them within a dozen once every state is reachable)
__device__ __forceinline__ void vb_save_vec(const uint32_t (&P)[64], uint8_t *dst)
{
    uint32_t mn = P[0];
#pragma unroll
    for (int i = 1; i < 64; i++) mn = min(mn, P[i]);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) w |= min(P[4 * i + b] - mn, 255u) << (8 * b);
        d[i] = w;
    }
}

// One butterfly, fully unrolled by the caller: predecessors j and j + 32, successors 2j and 2j + 1.  Keys are (metric << 1)
// + 2 cost (+ 1 for the predecessor with MSB 1): the minimum is the survivor, "equal -> predecessor with MSB 0", and its
// LSB is the decision bit, shifted into the history word by one v_alignbit.
template <int J>
__device__ __forceinline__ void vb_butterfly(const uint32_t (&P)[64], uint32_t (&N)[64], const uint32_t (&c2)[4], const uint32_t (&c2p)[4], uint32_t &hist)
{
    constexpr unsigned e = vb_expect(J), ne = 3u - e;
    const uint32_t a0 = (P[J] << 1) + c2[e],  a1 = (P[J + 32] << 1) + c2p[ne];
    const uint32_t b0 = (P[J] << 1) + c2[ne], b1 = (P[J + 32] << 1) + c2p[e];
    const uint32_t m0 = min(a0, a1), m1 = min(b0, b1);
    hist = __builtin_amdgcn_alignbit(m0, hist, 1); N[2 * J] = m0 >> 1;
    hist = __builtin_amdgcn_alignbit(m1, hist, 1); N[2 * J + 1] = m1 >> 1;
}
template <int J0>
__device__ __forceinline__ void vb_half(const uint32_t (&P)[64], uint32_t (&N)[64], const uint32_t (&c2)[4], const uint32_t (&c2p)[4], uint32_t &hist)
{
    vb_butterfly<J0 + 0>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 1>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 2>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 3>(P, N, c2, c2p, hist);
    vb_butterfly<J0 + 4>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 5>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 6>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 7>(P, N, c2, c2p, hist);
    vb_butterfly<J0 + 8>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 9>(P, N, c2, c2p, hist);  vb_butterfly<J0 + 10>(P, N, c2, c2p, hist); vb_butterfly<J0 + 11>(P, N, c2, c2p, hist);
    vb_butterfly<J0 + 12>(P, N, c2, c2p, hist); vb_butterfly<J0 + 13>(P, N, c2, c2p, hist); vb_butterfly<J0 + 14>(P, N, c2, c2p, hist); vb_butterfly<J0 + 15>(P, N, c2, c2p, hist);
}

// puncturing rows (A, B) of the K = 7 codes as bit masks over the column
__device__ __forceinline__ void vb_punct(int p, unsigned &pa, unsigned &pb)
{
    switch (p) {
    case 2: pa = 0x3; pb = 0x1; break;   case 3: pa = 0x3; pb = 0x5; break;   case 4: pa = 0xf; pb = 0x1; break;
    case 5: pa = 0xb; pb = 0x15; break;  case 6: pa = 0x17; pb = 0x29; break; case 7: pa = 0x2f; pb = 0x51; break;
    default: pa = 0x1; pb = 0x1; break;
    }
}

// One trellis step of one lane: P -> N.  Returns the 64 decision bits.
__device__ __forceinline__ unsigned long long vb_step(const uint32_t (&P)[64], uint32_t (&N)[64], unsigned pa, unsigned pb, unsigned up, uint32_t &col, uint32_t &nb,
                                                      uint32_t w0, uint32_t w1, uint32_t wbase, bool adv)
{
    const unsigned hasA = (pa >> col) & 1u, hasB = (pb >> col) & 1u;
    const unsigned long long w64 = ((unsigned long long)w0 << 32) | w1;
    const unsigned top2 = (unsigned)((w64 << ((nb - wbase) & 63u)) >> 62);
    const unsigned ra = top2 >> 1, rb = hasA ? (top2 & 1u) : (top2 >> 1);
    // 2 x cost of the expected pair e = A | B << 1 against what was received (punctured positions cost nothing)
    const uint32_t a0 = (hasA & ra) << 1, a1 = (hasA & (ra ^ 1u)) << 1, b0 = (hasB & rb) << 1, b1 = (hasB & (rb ^ 1u)) << 1;
    const uint32_t c2[4] = { a0 + b0, a1 + b0, a0 + b1, a1 + b1 };
    const uint32_t c2p[4] = { c2[0] | 1u, c2[1] | 1u, c2[2] | 1u, c2[3] | 1u };
    uint32_t h0 = 0, h1 = 0;
    vb_half<0>(P, N, c2, c2p, h0);
    vb_half<16>(P, N, c2, c2p, h1);
    if (adv) { nb += hasA + hasB; col = col + 1u == up ? 0u : col + 1u; }
    return ((unsigned long long)h1 << 32) | h0;
}

// Decision words are kept step-major within the 64 work items of a forward-pass wave: item slot i = 64 w + l keeps the word
// of its region step u at dwv[(w * blk + u) * 64 + l].  A wave's store of one step is then one contiguous 512-byte piece,
// and the traceback wave (same items, same lanes, walking the steps downwards together) reads it back the same way.
__device__ __forceinline__ unsigned long long *vb_slab(unsigned long long *dwv, uint32_t slot, uint32_t blk)
{
    return dwv + (size_t)(slot >> 6) * blk * 64u + (slot & 63u);
}

// The forward pass of one lane over a stretch of the trellis of a frame (coded bits at enc: packed, MSB first, punctured
// with period p).  The loop is the wave's: nsteps iterations (even), the lane's region [t_reg, t1) starting at iteration
// reg_at; before that a lane either warms up (init 0: all-equal metrics at t_reg - reg_at) or idles (init 1: the encoder's
// start state, init 2: metric differences read from init_vec).  Decisions of region step u go to dwl[64 u] (vb_slab).
__device__ __forceinline__ void vb_forward(const uint8_t *enc, int p, uint32_t t_reg, uint32_t t1, int init, const uint8_t *init_vec,
                                           unsigned long long *dwl, uint8_t *start_vec, uint8_t *end_vec, uint32_t nsteps, uint32_t reg_at, bool lane_on)
{
    uint32_t P[64], N[64];
#pragma unroll
    for (int i = 0; i < 64; i++) P[i] = 0u;
    unsigned pa, pb; vb_punct(p, pa, pb);
    const unsigned up = (unsigned)p;
    const uint32_t ts = init == 0 ? t_reg - reg_at : t_reg;               // first step this lane really runs
    uint32_t col = ts % up;
    uint32_t nb = (p == 1) ? 2u * ts : ts + (ts + up - 1u) / up;          // coded bits before step ts
    const uint32_t *enc32 = reinterpret_cast<const uint32_t *>(enc);      // (byte_off is a multiple of 16)
    uint32_t w0 = 0, w1 = 0, wbase = 0;
    for (uint32_t u = 0; u < nsteps; u += 2) {
        if (u == reg_at) {
            if (init == 1) {
#pragma unroll
                for (int i = 1; i < 64; i++) P[i] = 1u << 24;
                P[0] = 0u;
            } else if (init == 2) {
#pragma unroll
                for (int i = 0; i < 64; i++) P[i] = init_vec[i];
            }
            if (init != 1 && lane_on && start_vec) vb_save_vec(P, start_vec);
        }
        if ((u & 15u) == 0u) {                                             // refill the 64-bit window of coded bits
            const uint32_t idx = nb >> 5;
            const uint32_t x0 = lane_on ? enc32[idx] : 0u, x1 = lane_on ? enc32[idx + 1] : 0u;
            w0 = __builtin_bswap32(x0); w1 = __builtin_bswap32(x1); wbase = idx << 5;
        }
        const bool run = u >= reg_at || init == 0;                          // (a lane that idles keeps its stream position)
        const unsigned long long d0 = vb_step(P, N, pa, pb, up, col, nb, w0, w1, wbase, run);
        const unsigned long long d1 = vb_step(N, P, pa, pb, up, col, nb, w0, w1, wbase, run);
        // (regions are an even number of steps long within the slab: step ur + 1 is inside it whenever ur is)
        const uint32_t ur = u - reg_at, t = t_reg + ur;
        if (lane_on && u >= reg_at && t < t1) { __builtin_nontemporal_store(d0, dwl + (size_t)ur * 64u); __builtin_nontemporal_store(d1, dwl + (size_t)(ur + 1u) * 64u); }
    }
    if (lane_on && end_vec) vb_save_vec(P, end_vec);
}

// per work item, after the forward pass: [7:0] the end state its traceback started from, [15:8] the state it arrived at
#define VB_ST_REP 0x20000u        // ... were not: the block has been run again from the true ones
static_assert(FX_VB_WARM <= 128, "trellis blocks are at least 128 steps: a block's warm-up must fit into the block before it");
// warm-up steps by puncturing period: the rate-1/2 code merges fastest; the high-rate punctured codes (5/6, 6/7, 7/8) carry
// little redundancy per step and take longest (at marginal SNR a 96-step warm-up left every sixth of their blocks to be run again)
__device__ __forceinline__ uint32_t vb_warm(int p) { return p == 1 ? 64u : (p >= 5 ? 128u : (uint32_t)FX_VB_WARM); }
#ifndef FX_VB_TWARM
#define FX_VB_TWARM 64            // traceback warm-up: steps of the next block traced from state 0 to guess the block's end state
#endif

// ---- the forward pass proper: TWO trellis blocks per lane, their doubled metrics side by side in the 16-bit halves of a
// register.  v_pk_add_u16 / v_pk_min_u16 then serve both: per butterfly and pair of blocks 4 adds, 2 mins, 2 masks (the
// survivors' parity bits out of the next step's metrics) and 2 x 2 operations that collect the decision bits -- 6
// instructions per butterfly and trellis block instead of 10.  Halves cannot overflow: metric differences stay below 256,
// a block and its warm-up are at most 4192 steps of at most 2 each, doubled: < 2^15.
// A lane's two work items are slots 128 w + l (low halves) and 128 w + 64 + l (high halves): both keep the step-major
// layout of their own 64-slot slab, so every store is still one contiguous 512-byte piece per half.
typedef unsigned short vb_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t vb_pk_add(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, (vb_u16x2)(__builtin_bit_cast(vb_u16x2, a) + __builtin_bit_cast(vb_u16x2, b)));
}
__device__ __forceinline__ uint32_t vb_pk_min(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(vb_u16x2, a), __builtin_bit_cast(vb_u16x2, b)));
}

// One butterfly of both blocks: predecessors J and J + 32, successors 2J and 2J + 1.  Keys are doubled metrics + 2 cost,
// + 1 for the predecessor with MSB 1; their minimum is the survivor ("equal -> predecessor with MSB 0") and its LSB the
// decision bit.  The history word of the successors' group of 16 states takes state 2J + 1 first, then 2J: groups are run
// through from their highest butterfly down, so that state 16 g + i ends up at bit i of either half.
template <int J>
__device__ __forceinline__ void vb2_butterfly(const uint32_t (&Q)[64], uint32_t (&N)[64], const uint32_t (&cc)[4], const uint32_t (&ccp)[4], uint32_t &hist)
{
    constexpr unsigned e = vb_expect(J), ne = 3u - e;
    const uint32_t m0 = vb_pk_min(vb_pk_add(Q[J], cc[e]),  vb_pk_add(Q[J + 32], ccp[ne]));
    const uint32_t m1 = vb_pk_min(vb_pk_add(Q[J], cc[ne]), vb_pk_add(Q[J + 32], ccp[e]));
    hist = (hist << 1) | (m1 & 0x00010001u); N[2 * J + 1] = m1 & 0xFFFEFFFEu;
    hist = (hist << 1) | (m0 & 0x00010001u); N[2 * J] = m0 & 0xFFFEFFFEu;
}
template <int G>
__device__ __forceinline__ uint32_t vb2_group(const uint32_t (&Q)[64], uint32_t (&N)[64], const uint32_t (&cc)[4], const uint32_t (&ccp)[4])
{
    uint32_t h = 0;
    vb2_butterfly<8 * G + 7>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 6>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 5>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 4>(Q, N, cc, ccp, h);
    vb2_butterfly<8 * G + 3>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 2>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 1>(Q, N, cc, ccp, h); vb2_butterfly<8 * G + 0>(Q, N, cc, ccp, h);
    return h;
}

// the coded-bit stream of one trellis block: position, puncturing column, a 64-bit window
struct VbStream { uint32_t col, nb, w0, w1, wbase; };
// 2 x cost of the four expected pairs e = A | B << 1 against what was received at this step (punctured positions cost nothing)
__device__ __forceinline__ void vb_costs(VbStream &s, unsigned pa, unsigned pb, unsigned up, bool adv, uint32_t (&c2)[4])
{
    const unsigned hasA = (pa >> s.col) & 1u, hasB = (pb >> s.col) & 1u;
    const unsigned long long w64 = ((unsigned long long)s.w0 << 32) | s.w1;
    const unsigned top2 = (unsigned)((w64 << ((s.nb - s.wbase) & 63u)) >> 62);
    const unsigned ra = top2 >> 1, rb = hasA ? (top2 & 1u) : (top2 >> 1);
    const uint32_t a0 = (hasA & ra) << 1, a1 = (hasA & (ra ^ 1u)) << 1, b0 = (hasB & rb) << 1, b1 = (hasB & (rb ^ 1u)) << 1;
    c2[0] = a0 + b0; c2[1] = a1 + b0; c2[2] = a0 + b1; c2[3] = a1 + b1;
    if (adv) { s.nb += hasA + hasB; s.col = s.col + 1u == up ? 0u : s.col + 1u; }
}
// One trellis step of both blocks: Q -> N; the decision words of the low-half and of the high-half block.
__device__ __forceinline__ void vb2_step(const uint32_t (&Q)[64], uint32_t (&N)[64], unsigned pa, unsigned pb, unsigned up, VbStream &sa, VbStream &sb, bool adv_a, bool adv_b,
                                         unsigned long long &da, unsigned long long &db)
{
    uint32_t ca[4], cb[4], cc[4], ccp[4];
    vb_costs(sa, pa, pb, up, adv_a, ca); vb_costs(sb, pa, pb, up, adv_b, cb);
#pragma unroll
    for (int i = 0; i < 4; i++) { cc[i] = ca[i] | (cb[i] << 16); ccp[i] = cc[i] | 0x00010001u; }
    const uint32_t h0 = vb2_group<0>(Q, N, cc, ccp), h1 = vb2_group<1>(Q, N, cc, ccp), h2 = vb2_group<2>(Q, N, cc, ccp), h3 = vb2_group<3>(Q, N, cc, ccp);
    // low halves -> the low-half block's word (states 0..63 at bits 0..63), high halves -> the other's
    const uint32_t al = __builtin_amdgcn_perm(h1, h0, 0x05040100u), ah = __builtin_amdgcn_perm(h3, h2, 0x05040100u);
    const uint32_t bl = __builtin_amdgcn_perm(h1, h0, 0x07060302u), bh = __builtin_amdgcn_perm(h3, h2, 0x07060302u);
    da = ((unsigned long long)ah << 32) | al; db = ((unsigned long long)bh << 32) | bl;
}

// The traceback of the block BEFORE this one does not know its end state.  Survivors merge within a few constraint lengths, so
// tracing this block's first FX_VB_TWARM steps back from state 0 arrives, almost always, at the state the true path passes
// through at this block's first step: the guess.  It is made here, by the lane that has just written those decision words
// (they come back from the L2, not from HBM -- fx_vbtrace_kernel used to read them a second time for this), and left in bits
// 24..29 of the item's status word; fx_vbfinish_kernel verifies every guess against the state the traceback really arrives at.
__device__ __forceinline__ void vb_trace16(const unsigned long long *dwl, uint32_t u0, uint32_t lim, unsigned &st, uint32_t &bits16);
__device__ __forceinline__ uint32_t vb_make_guess(const unsigned long long *dwl, uint32_t len)
{
    const uint32_t wl = min((uint32_t)FX_VB_TWARM, len);
    unsigned S = 0;
    for (int grp = FX_VB_TWARM / 16 - 1; grp >= 0; grp--) {
        if (16u * (uint32_t)grp >= wl) continue;
        uint32_t unused = 0;
        vb_trace16(dwl, 16u * (uint32_t)grp, wl, S, unused);
    }
    return (S & 63u) << 24;
}

// metric differences of one half to its smallest, as bytes (see vb_save_vec; the halves hold doubled metrics)
__device__ __forceinline__ void vb2_save_vec(const uint32_t (&Q)[64], int half, uint8_t *dst)
{
    uint32_t mn = 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 64; i++) mn = min(mn, (Q[i] >> (16 * half)) & 0xFFFFu);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t w = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) w |= min((((Q[4 * i + b] >> (16 * half)) & 0xFFFFu) - mn) >> 1, 255u) << (8 * b);
        d[i] = w;
    }
}

struct VbHalf {                       // one of a lane's two work items
    const uint8_t *enc; uint32_t t_reg, t1; bool on, first; unsigned long long *dwl; uint8_t *vec;
};

// Forward pass of a lane's two trellis blocks (same puncturing code p): vb_warm(p) steps of warm-up from all-equal
// metrics (a frame's first block idles instead and starts from the encoder's state), then blk region steps whose decisions
// go to the blocks' slabs; start and end metric differences to the blocks' vector slots.
// Coded bits reach the lanes through LDS: every 128 steps a lane fetches the 64 bytes (16-byte aligned) that hold its block's next
// <= 256 coded bits and parks them in its own 17-dword row (odd stride: no bank conflicts); the 64-bit window the steps eat from is
// refilled from there every 16 steps.  (Refilling it straight from global memory -- two dwords per lane, 64 lanes 36 to 100 bytes
// apart, back in the same cache lines sixteen steps later -- cost 35 x the coded bits in HBM fetches on config 4: the lines did not
// survive in the L2 between visits.)
#define VB_ROW 17
__device__ __forceinline__ void vb_stage(const uint8_t *enc, uint32_t nb, bool on, uint32_t *row, uint32_t &lbase)
{
    const uint32_t o = (nb >> 3) & ~15u;
    const uint4 *src = reinterpret_cast<const uint4 *>(enc + o);
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0, q2 = q0, q3 = q0;
    if (on) { q0 = src[0]; q1 = src[1]; q2 = src[2]; q3 = src[3]; }
    row[0] = q0.x; row[1] = q0.y; row[2] = q0.z; row[3] = q0.w; row[4] = q1.x; row[5] = q1.y; row[6] = q1.z; row[7] = q1.w;
    row[8] = q2.x; row[9] = q2.y; row[10] = q2.z; row[11] = q2.w; row[12] = q3.x; row[13] = q3.y; row[14] = q3.z; row[15] = q3.w;
    lbase = o << 3;
}
__device__ __forceinline__ void vb2_forward(int p, uint32_t blk, const VbHalf &A, const VbHalf &B, uint32_t *lds)
{
    uint32_t Q[64], N[64];
#pragma unroll
    for (int i = 0; i < 64; i++) Q[i] = 0u;
    unsigned pa, pb; vb_punct(p, pa, pb);
    const unsigned up = (unsigned)p;
    const uint32_t warm = vb_warm(p);
    VbStream sa, sb;
    {
        const uint32_t tsa = A.first ? A.t_reg : A.t_reg - warm, tsb = B.first ? B.t_reg : B.t_reg - warm;
        sa.col = tsa % up; sa.nb = (p == 1) ? 2u * tsa : tsa + (tsa + up - 1u) / up; sa.w0 = sa.w1 = sa.wbase = 0;
        sb.col = tsb % up; sb.nb = (p == 1) ? 2u * tsb : tsb + (tsb + up - 1u) / up; sb.w0 = sb.w1 = sb.wbase = 0;
    }
    uint32_t *rowa = lds + (threadIdx.x & 63) * VB_ROW, *rowb = lds + (64 + (threadIdx.x & 63)) * VB_ROW;
    uint32_t la = 0, lb = 0;
    const uint32_t nsteps = warm + blk;
    for (uint32_t u = 0; u < nsteps; u += 2) {
        if (u == warm) {
            // the encoder's start state for a first block (every other state out of reach: beaten by anything real within
            // six steps); the record of the start differences for the others
            if (A.first) {
#pragma unroll
                for (int i = 0; i < 64; i++) Q[i] = (Q[i] & 0xFFFF0000u) | (i ? 1024u : 0u);
            } else if (A.on) vb2_save_vec(Q, 0, A.vec);
            if (B.first) {
#pragma unroll
                for (int i = 0; i < 64; i++) Q[i] = (Q[i] & 0x0000FFFFu) | (i ? 1024u << 16 : 0u);
            } else if (B.on) vb2_save_vec(Q, 1, B.vec);
        }
        if ((u & 127u) == 0u) { vb_stage(A.enc, sa.nb, A.on, rowa, la); vb_stage(B.enc, sb.nb, B.on, rowb, lb); }
        if ((u & 15u) == 0u) {                                             // refill the 64-bit windows of coded bits (own row: no barrier needed)
            const uint32_t ia = (sa.nb - la) >> 5, ib = (sb.nb - lb) >> 5;
            sa.w0 = __builtin_bswap32(rowa[ia]); sa.w1 = __builtin_bswap32(rowa[ia + 1]); sa.wbase = la + (ia << 5);
            sb.w0 = __builtin_bswap32(rowb[ib]); sb.w1 = __builtin_bswap32(rowb[ib + 1]); sb.wbase = lb + (ib << 5);
        }
        const bool in_reg = u >= warm;
        const bool run_a = in_reg || !A.first, run_b = in_reg || !B.first;     // (a block that idles keeps its stream position)
        unsigned long long a0, b0, a1, b1;
        vb2_step(Q, N, pa, pb, up, sa, sb, run_a, run_b, a0, b0);
        vb2_step(N, Q, pa, pb, up, sa, sb, run_a, run_b, a1, b1);
        // (regions are an even number of steps long within the slab: step ur + 1 is inside it whenever ur is)
        const uint32_t ur = u - warm;
        // decision words stream through -- written once, read once by the traceback: non-temporal, so that they do not flood the L2 --
        // except the first FX_VB_TWARM steps of a block, which this lane reads back at the end for the traceback's guess
        if (in_reg && ur < (uint32_t)FX_VB_TWARM) {
            if (A.on && A.t_reg + ur < A.t1) { A.dwl[(size_t)ur * 64u] = a0; A.dwl[(size_t)(ur + 1u) * 64u] = a1; }
            if (B.on && B.t_reg + ur < B.t1) { B.dwl[(size_t)ur * 64u] = b0; B.dwl[(size_t)(ur + 1u) * 64u] = b1; }
        } else {
            if (A.on && in_reg && A.t_reg + ur < A.t1) { __builtin_nontemporal_store(a0, A.dwl + (size_t)ur * 64u); __builtin_nontemporal_store(a1, A.dwl + (size_t)(ur + 1u) * 64u); }
            if (B.on && in_reg && B.t_reg + ur < B.t1) { __builtin_nontemporal_store(b0, B.dwl + (size_t)ur * 64u); __builtin_nontemporal_store(b1, B.dwl + (size_t)(ur + 1u) * 64u); }
        }
    }
    if (A.on) vb2_save_vec(Q, 0, A.vec + 64);
    if (B.on) vb2_save_vec(Q, 1, B.vec + 64);
}

// ---- forward pass: one lane per (frame, trellis block) work item; the items of a wave share their puncturing code ----
// vb_items: [0, cap) the frame of every item slot (0xFFFFFFFF: padding of a code class's last wave), [cap, 2 cap) its block
struct VbItem { uint32_t g, b, t_reg, t1, Tn, nblk; bool on; };
__device__ __forceinline__ VbItem vb_item(const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap, uint32_t slot, uint32_t blk)
{
    VbItem it;
    const uint32_t item = vb_items[slot];
    it.on = item != 0xFFFFFFFFu;
    it.g = it.on ? item : 0u; it.b = it.on ? vb_items[item_cap + slot] : 0u;
    const FxPayJob &job = jobs[it.g];
    it.Tn = 8u * job.k + 6u; it.nblk = job.vb_nblk;
    it.t_reg = it.b * blk; it.t1 = min(it.Tn, it.t_reg + blk);
    return it;
}

extern "C" __global__ __launch_bounds__(64, 3)
void fx_vbfwd_kernel(const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap, const FxBlockHdr *hdr, uint32_t first_item, const uint8_t *bufB,
                     unsigned long long *dwv, uint8_t *vec_arena, uint32_t *vb_st)
{
    // (128 item slots per wave: the low-half blocks are slots it0 .. it0 + 63, the high-half blocks the 64 behind them;
    // code classes are padded to 128 slots, so a wave's items share their puncturing code)
    const uint32_t nitems = hdr->n_vb_items, blk = hdr->vb_blk;
    const uint32_t it0 = first_item + blockIdx.x * 128u;
    if (it0 >= nitems) return;
    const uint32_t slot_a = it0 + threadIdx.x, slot_b = slot_a + 64u;
    const VbItem ia = vb_item(jobs, vb_items, item_cap, slot_a, blk);
    VbItem ib = ia; ib.on = false;
    if (slot_b < nitems) ib = vb_item(jobs, vb_items, item_cap, slot_b, blk);
    const unsigned long long live_a = __ballot(ia.on), live_b = __ballot(ib.on);
    if (!live_a && !live_b) return;
    const FxPayJob &ja = jobs[ia.g], &jb = jobs[ib.g];
    // (a padding slot has no frame of its own: the wave's code class is read from a live lane)
    const int p = live_a ? conv_p((unsigned)__shfl((int)ja.fec0, __ffsll((long long)live_a) - 1, 64))
                         : conv_p((unsigned)__shfl((int)jb.fec0, __ffsll((long long)live_b) - 1, 64));
    VbHalf A, B;
    A.enc = bufB + ja.byte_off; A.on = ia.on; A.first = !ia.on || ia.b == 0; A.t_reg = A.first && !ia.on ? 0u : ia.t_reg; A.t1 = ia.t1;
    A.dwl = vb_slab(dwv, slot_a, blk); A.vec = vec_arena + (size_t)slot_a * 128u;
    B.enc = bufB + jb.byte_off; B.on = ib.on; B.first = !ib.on || ib.b == 0; B.t_reg = B.first && !ib.on ? 0u : ib.t_reg; B.t1 = ib.t1;
    B.dwl = vb_slab(dwv, ib.on ? slot_b : slot_a, blk); B.vec = vec_arena + (size_t)(ib.on ? slot_b : slot_a) * 128u;
    __shared__ uint32_t rows[128 * VB_ROW];
    vb2_forward(p, blk, A, B, rows);
    if (ia.on) vb_st[slot_a] = vb_make_guess(A.dwl, ia.t1 - ia.t_reg);
    if (ib.on) vb_st[slot_b] = vb_make_guess(B.dwl, ib.t1 - ib.t_reg);
}

// The same pass, one work item per lane (32-bit metrics): twice the waves of the packed kernel at about 1.6 x its instructions
// per block.  Faster when the block's work items do not even fill the chip once (one block in flight: latency), slower as soon
// as they do.
extern "C" __global__ __launch_bounds__(64)
void fx_vbfwd1_kernel(const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap, const FxBlockHdr *hdr, uint32_t first_item, const uint8_t *bufB,
                      unsigned long long *dwv, uint8_t *vec_arena, uint32_t *vb_st)
{
    const uint32_t nitems = hdr->n_vb_items, blk = hdr->vb_blk;
    const uint32_t it0 = first_item + blockIdx.x * 64u;
    if (it0 >= nitems) return;
    const uint32_t slot = it0 + threadIdx.x;
    const VbItem it = vb_item(jobs, vb_items, item_cap, slot, blk);
    const unsigned long long live = __ballot(it.on);
    if (!live) return;
    const FxPayJob &job = jobs[it.g];
    const int p = conv_p((unsigned)__shfl((int)job.fec0, __ffsll((long long)live) - 1, 64));
    uint8_t *vec = vec_arena + (size_t)slot * 128u;
    const bool first = !it.on || it.b == 0;
    vb_forward(bufB + job.byte_off, p, it.t_reg, it.t1, first ? 1 : 0, nullptr, vb_slab(dwv, slot, blk), first ? nullptr : vec, vec + 64,
               vb_warm(p) + blk, vb_warm(p), it.on);
    if (it.on) vb_st[slot] = vb_make_guess(vb_slab(dwv, slot, blk), it.t1 - it.t_reg);
}

__device__ __forceinline__ bool vb_same64(const uint8_t *a, const uint8_t *b)
{
    const uint4 *pa = reinterpret_cast<const uint4 *>(a), *pb = reinterpret_cast<const uint4 *>(b);
    bool same = true;
#pragma unroll
    for (int i = 0; i < 4; i++) { const uint4 x = pa[i], y = pb[i]; same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w; }
    return same;
}

// ---- hand-over check: a block whose warm-up did not arrive at the true metric differences -- its start record differs from the end
// record of the block before it -- runs again from them (same lanes as the forward pass; a wave without such a block leaves at
// once).  An optimisation only, launched while blocks keep needing it (fx_host.cpp): results never rest on it -- fx_vbfinish_kernel
// compares, for every block of a frame, the FINAL start record with the FINAL end record of its predecessor and hands the frame to
// the wave-per-frame decoder if any pair differs (a block that ran again here and ended differently leaves such a pair behind it).
extern "C" __global__ __launch_bounds__(64)
void fx_vbfix_kernel(const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap, const FxBlockHdr *hdr, uint32_t first_item, const uint8_t *bufB,
                     unsigned long long *dwv, uint8_t *vec_arena, uint32_t *vb_st, uint32_t dbg)
{
    const uint32_t nitems = hdr->n_vb_items, blk = hdr->vb_blk;
    const uint32_t it0 = first_item + blockIdx.x * 64u;
    if (it0 >= nitems) return;
    const uint32_t slot = it0 + threadIdx.x;
    const VbItem it = vb_item(jobs, vb_items, item_cap, slot, blk);
    uint8_t *vec = vec_arena + (size_t)slot * 128u;
    // (dbg bit 1, tests only: every other failing block is left as it is, for the fallback path to be exercised)
    const bool bad = it.on && it.b > 0 && !vb_same64(vec, vec - 64) && !((dbg & 2u) && (slot & 1u));
    const unsigned long long fails = __ballot(bad);
    if (!fails) return;
    const FxPayJob &job = jobs[it.g];
    const int p = conv_p((unsigned)__shfl((int)job.fec0, __ffsll((long long)fails) - 1, 64));
    unsigned long long *dwl = vb_slab(dwv, slot, blk);
    vb_forward(bufB + job.byte_off, p, it.t_reg, it.t1, 2, bad ? vec - 64 : vec, dwl, vec, vec + 64, blk, 0u, bad);
    if (bad) vb_st[slot] = VB_ST_REP | vb_make_guess(dwl, it.t1 - it.t_reg);       // (fx_vbfinish_kernel counts these; the guess: from the new decisions)
}

// sixteen traceback steps u0 + 15 .. u0 of one lane (those below lim only): the words first, then the chain through them
__device__ __forceinline__ void vb_trace16(const unsigned long long *dwl, uint32_t u0, uint32_t lim, unsigned &st, uint32_t &bits16)
{
    unsigned long long q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) q[i] = (u0 + (uint32_t)i < lim) ? dwl[(size_t)(u0 + (uint32_t)i) * 64u] : 0ull;
#pragma unroll
    for (int i = 15; i >= 0; i--) {
        if (u0 + (uint32_t)i < lim) {
            bits16 |= (st & 1u) << i;
            st = (st >> 1) | ((unsigned)((q[i] >> st) & 1ull) << 5);
        }
    }
}
// steps 64c .. 64c+63 of a frame are output bytes 8c .. 8c+7, MSB first
__device__ __forceinline__ void vb_emit(uint8_t *dec, uint32_t c, uint32_t b0, uint32_t b1)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lo |= (__brev((b0 >> (8 * k)) & 0xffu) >> 24) << (8 * k);
        hi |= (__brev((b1 >> (8 * k)) & 0xffu) >> 24) << (8 * k);
    }
    *reinterpret_cast<uint2 *>(dec + 8 * (size_t)c) = make_uint2(lo, hi);
}
// the 64-step chunk c of a region of len steps, entered in state st: bits out, state at the chunk's start back
__device__ __forceinline__ unsigned vb_trace_chunk(const unsigned long long *dwl, uint32_t c, uint32_t len, unsigned st, uint32_t &b0, uint32_t &b1)
{
    unsigned long long bits = 0;
#pragma unroll 1
    for (int grp = 3; grp >= 0; grp--) {                                   // (rolled: sixteen words in flight, not sixty-four)
        uint32_t b16 = 0;
        vb_trace16(dwl, 64u * c + 16u * (uint32_t)grp, len, st, b16);
        bits |= (unsigned long long)b16 << (16 * grp);
    }
    b0 = (uint32_t)bits; b1 = (uint32_t)(bits >> 32);
    return st;
}

// ---- traceback: the same lane per work item, all lanes of a wave walking down their blocks together ----
// A block other than the last one of its frame does not know its end state: it traces the first FX_VB_TWARM steps of the
// next block from state 0 (the survivors of all states merge within a few constraint lengths) and starts from where that
// arrives.  fx_vbfinish_kernel checks that guess against the state the next block's traceback really arrived at.
extern "C" __global__ __launch_bounds__(64)
void fx_vbtrace_kernel(const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap, const FxBlockHdr *hdr, uint32_t first_item, uint8_t *bufA,
                       unsigned long long *dwv, uint32_t *vb_st, uint32_t dbg)
{
    const uint32_t nitems = hdr->n_vb_items, blk = hdr->vb_blk;
    const uint32_t it0 = first_item + blockIdx.x * 64u;
    if (it0 >= nitems) return;
    const uint32_t slot = it0 + threadIdx.x;
    const VbItem it = vb_item(jobs, vb_items, item_cap, slot, blk);
    if (!__ballot(it.on)) return;
    const FxPayJob &job = jobs[it.g];
    const uint32_t len = it.on ? it.t1 - it.t_reg : 0u;
    uint32_t old = it.on ? vb_st[slot] : 0u;
    uint32_t flags = 0u;
    const unsigned long long *dwl = vb_slab(dwv, slot, blk);
    flags = old & VB_ST_REP;
    // the end state: 0 behind the flushed tail; elsewhere the guess the next block's forward pass left (its first FX_VB_TWARM
    // steps traced back from state 0)
    const bool has_next = it.on && it.b + 1u < it.nblk;
    // (dbg bit 0, tests only: the guess is state 0 -- wrong 63 times in 64, for the re-trace path to be exercised)
    const unsigned S = (has_next && !(dbg & 1u)) ? ((vb_st[slot + 1u] >> 24) & 63u) : 0u;
    uint8_t *A = bufA + job.byte_off + it.t_reg / 8u;
    unsigned st = S;
    for (int c = (int)(blk / 64u) - 1; c >= 0; c--) {
        if (!__any(64u * (uint32_t)c < len)) continue;
        uint32_t b0, b1;
        st = vb_trace_chunk(dwl, (uint32_t)c, len, st, b0, b1);
        if (64u * (uint32_t)c < len) vb_emit(A, (uint32_t)c, b0, b1);
    }
    if (it.on) vb_st[slot] = S | (st << 8) | flags | (old & 0x3F000000u);       // (the guess bits stay: the item before this one may still be reading them)
}

// One block's traceback again, by a whole wave (its end-state guess was wrong): parallel over 64-step chunks with exact
// verification -- chunk c is first entered from a guess of its end state (chunk c+1 traced back from state 0), then the
// chain of chunk start states is checked and chunks entered from a wrong state are traced again, to the fixed point.
// Returns the state at the block's first step.
__device__ __forceinline__ unsigned vb_retrace_block(const unsigned long long *dwl, uint32_t len, unsigned end_state, uint8_t *dec, uint8_t *scratch, int lane)
{
    const uint32_t nchunk = (len + 63) / 64, Lc = nchunk - 1;             // (at most 64 chunks: blk <= 4096)
    uint8_t *Sarr = scratch, *Barr = scratch + nchunk;                     // (2 len / 64 bytes: the frame's byte buffer holds more than len / 8)
    const uint32_t c = (uint32_t)lane;
    if (c < nchunk) {
        uint32_t b0, b1; unsigned S = end_state;
        if (c < Lc) S = vb_trace_chunk(dwl, c + 1, len, 0u, b0, b1);
        const unsigned Bst = vb_trace_chunk(dwl, c, len, S, b0, b1);
        Sarr[c] = (uint8_t)S; Barr[c] = (uint8_t)Bst; vb_emit(dec, c, b0, b1);
    }
    for (;;) {
        __threadfence_block(); __builtin_amdgcn_wave_barrier();
        bool redo = false; unsigned need = 0;
        if (c < nchunk) { need = c == Lc ? end_state : Barr[c + 1]; redo = Sarr[c] != need; }
        if (!__any(redo)) break;
        __builtin_amdgcn_wave_barrier();
        if (redo) {
            uint32_t b0, b1;
            const unsigned Bst = vb_trace_chunk(dwl, c, len, need, b0, b1);
            Sarr[c] = (uint8_t)need; Barr[c] = (uint8_t)Bst; vb_emit(dec, c, b0, b1);
        }
    }
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    return Barr[0];
}

// ---- front part, one wave per frame: symbols -> packet bytes, de-interleave, fec1 (not convolutional), de-interleave ----
// The de-interleaver is four passes of dependent byte swaps: through global memory every one of its ~17 rounds per pass is
// a memory round trip (the kernel took 113 us for 2-KB packets, nearly all of it waiting).  Packets that fit are therefore
// packed and de-interleaved in LDS and written out once; longer ones take the same steps in place in global memory.
#define VBPRE_LDS 4608           // (packets up to ~4.5 KB -- 2 KB payloads at rate 1/2 -- stay in LDS; at 12 KB a CU held only 13 of these one-wave workgroups)
__device__ __forceinline__ void vb_pack_bytes(const uint8_t *hs, unsigned bps, uint32_t nbytes, uint8_t *dst, int lane)
{
    // (bit 8 j of the packet is bit bps - 1 - (8 j mod bps) of symbol 8 j / bps: one division per byte, then the bits are walked)
    for (uint32_t j = lane; j < nbytes; j += DEC_THREADS) {
        uint32_t sidx = (8u * j) / bps; int sb = (int)(bps - 1u - (8u * j - sidx * bps));
        unsigned v = 0, cur = hs[sidx];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            v = (v << 1) | ((cur >> sb) & 1u);
            if (--sb < 0 && b < 7) { sb = (int)bps - 1; cur = hs[++sidx]; }
        }
        dst[j] = (uint8_t)v;
    }
}
extern "C" __global__ __launch_bounds__(DEC_THREADS)
void fx_vbpre_kernel(const FxPayJob *jobs, const uint32_t *job_idx, const FxBlockHdr *hdr, uint32_t first_wave, const uint8_t *hard, uint8_t *bufA, uint8_t *bufB,
                     const FxTables *T)
{
    __shared__ __attribute__((aligned(16))) uint8_t X[VBPRE_LDS];
    const uint32_t njobs = hdr->n_dec_batch;
    const uint32_t ji = first_wave + blockIdx.x;
    if (ji >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t jf = job_idx[ji];
    FxPayJob job = jobs[jf];
    job.l0 = __builtin_amdgcn_readfirstlane(job.l0); job.l1 = __builtin_amdgcn_readfirstlane(job.l1);
    job.fec1 = __builtin_amdgcn_readfirstlane(job.fec1); job.bps = __builtin_amdgcn_readfirstlane(job.bps);
    uint8_t *A = bufA + job.byte_off, *B = bufB + job.byte_off;
    const uint8_t *hs = hard + job.sym_off;
    const bool in_lds = job.l1 + 32u <= VBPRE_LDS && job.l0 + 32u <= VBPRE_LDS;
    if (in_lds) {
        vb_pack_bytes(hs, job.bps, job.l1, X, lane);
        __builtin_amdgcn_wave_barrier();
        deinterleave_wave(X, job.l1, lane);
        if (job.fec1 == FX_FEC_NONE) {
            // (no outer code: the packet bytes are the inner code's already -- l0 = l1 -- and stay where they are)
            deinterleave_wave(X, job.l0, lane);
        } else {
            for (uint32_t j = lane; j < job.l1; j += DEC_THREADS) A[j] = X[j];
            __threadfence_block(); __builtin_amdgcn_wave_barrier();
            block_fec_decode<false>(job.fec1, job.l0, A, B, T, lane);
            __threadfence_block(); __builtin_amdgcn_wave_barrier();
            for (uint32_t j = lane; j < job.l0; j += DEC_THREADS) X[j] = B[j];
            __builtin_amdgcn_wave_barrier();
            deinterleave_wave(X, job.l0, lane);
        }
        // out in 16-byte pieces (byte_off is a multiple of 16; the buffer has that much slack behind the packet), with the
        // eight defined bytes behind the coded bits that the forward pass's window reads run into
        if (lane < 16) X[job.l0 + lane] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t n16 = (job.l0 + 8u + 15u) / 16u;
        for (uint32_t j = lane; j < n16; j += DEC_THREADS) reinterpret_cast<uint4 *>(B)[j] = reinterpret_cast<const uint4 *>(X)[j];
        return;
    }
    vb_pack_bytes(hs, job.bps, job.l1, A, lane);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    deinterleave_wave(A, job.l1, lane);
    block_fec_decode<false>(job.fec1, job.l0, A, B, T, lane);
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    deinterleave_wave(B, job.l0, lane);
    // (the window reads of the forward pass run up to 8 bytes past the coded bits: keep them defined)
    if (lane < 8) B[job.l0 + lane] = 0;
}

// ---- back part, one wave per frame: the chain of traceback states across the frame's blocks, then the frame's tail ----
extern "C" __global__ __launch_bounds__(64)
void fx_vbfinish_kernel(const FxPayJob *jobs, const uint32_t *job_idx, FxBlockHdr *hdr, uint32_t first_wave, uint8_t *bufA, uint8_t *bufB,
                        unsigned long long *dwv, const uint8_t *vec_arena, const uint32_t *vb_st, uint32_t *fb_list, uint32_t list_cap, uint8_t *out, FxOutRec *recs,
                        FxBlockHdr *hdr_host)
{
    const uint32_t njobs = hdr->n_dec_batch, blk = hdr->vb_blk;
    const uint32_t ji = first_wave + blockIdx.x;
    if (ji >= njobs) return;
    const int lane = threadIdx.x & 63;
    const uint32_t jf = job_idx[ji];
    FxPayJob job = jobs[jf];
    job.k = __builtin_amdgcn_readfirstlane(job.k); job.pay_len = __builtin_amdgcn_readfirstlane(job.pay_len);
    job.check = __builtin_amdgcn_readfirstlane(job.check); job.vb_nblk = __builtin_amdgcn_readfirstlane(job.vb_nblk);
    job.vb_off = __builtin_amdgcn_readfirstlane(job.vb_off);
    uint8_t *A = bufA + job.byte_off, *B = bufB + job.byte_off;
    const uint32_t Tn = 8u * job.k + 6u, at = job.vb_off, nblk = job.vb_nblk;
    bool bad = false, mism = false; uint32_t rep = 0;
    for (uint32_t base = 0; base < nblk; base += 64) {
        const uint32_t b = base + (uint32_t)lane;
        const uint32_t v = b < nblk ? vb_st[at + b] : 0u, vn = b + 1u < nblk ? vb_st[at + b + 1u] : 0u;
        // every hand-over, on the records as they stand now: block b started from what block b - 1 ended with
        if (b > 0 && b < nblk) { const uint8_t *vec = vec_arena + (size_t)(at + b) * 128u; const bool same = vb_same64(vec, vec - 64); bad = bad || !same; }
        rep += (uint32_t)__popcll(__ballot((v & VB_ST_REP) != 0u));
        mism = mism || (b + 1u < nblk && (v & 0xffu) != ((vn >> 8) & 0xffu));
    }
    if (__any(bad)) {                                                       // an unverified hand-over: the frame is decoded the other way
        // (the host's copy only learns THAT frames were handed back -- a plain store, whoever gets there --; fxrx_collect then fetches the
        // count if the fallback launch of the chain, sized from the last block, may not have covered it)
        if (lane == 0) { const uint32_t s = atomicAdd(&hdr->n_vb_fallback, 1u); if (s < list_cap) fb_list[s] = jf; hdr_host->vb_ticket = 1u; }
        return;
    }
    if (__any(mism)) {                                                      // a wrong end-state guess: from the last block downwards
        unsigned need = (vb_st[at + nblk - 1u] >> 8) & 0xffu;
        for (int b = (int)nblk - 2; b >= 0; b--) {
            const uint32_t v = vb_st[at + (uint32_t)b];
            if ((v & 0xffu) != need) need = vb_retrace_block(vb_slab(dwv, at + (uint32_t)b, blk), blk, need, A + (size_t)b * (blk / 8u), B, lane);
            else need = (v >> 8) & 0xffu;
        }
    }
    (void)Tn;
    __threadfence_block(); __builtin_amdgcn_wave_barrier();
    dec_tail(job, jf, A, lane, out, recs, rep << 8);
}

extern "C" hipError_t fx_launch_vbpre(unsigned first_wave, unsigned n_waves, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx, const FxBlockHdr *hdr,
                                      const uint8_t *hard, uint8_t *bufA, uint8_t *bufB, const FxTables *T)
{
    if (n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL(fx_vbpre_kernel, dim3(n_waves), dim3(DEC_THREADS), 0, st, jobs, job_idx, hdr, first_wave, hard, bufA, bufB, T);
    return hipGetLastError();
}
// forward pass, [hand-over check,] traceback: the lane-per-work-item kernels, over the same item slots
extern "C" hipError_t fx_launch_vbitems(unsigned first_item, unsigned n_items, hipStream_t st, const FxPayJob *jobs, const uint32_t *vb_items, uint32_t item_cap,
                                        const FxBlockHdr *hdr, uint8_t *bufA, const uint8_t *bufB, unsigned long long *dwv, uint8_t *vec_arena, uint32_t *vb_st, uint32_t dbg, int packed,
                                        int with_fix)
{
    if (n_items == 0) return hipSuccess;
    const dim3 grid((n_items + 63) / 64), block(64);
    if (!packed) hipLaunchKernelGGL(fx_vbfwd1_kernel, grid, block, 0, st, jobs, vb_items, item_cap, hdr, first_item, bufB, dwv, vec_arena, vb_st);
    else hipLaunchKernelGGL(fx_vbfwd_kernel, dim3((n_items + 127) / 128), block, 0, st, jobs, vb_items, item_cap, hdr, first_item, bufB, dwv, vec_arena, vb_st);
    if (with_fix) hipLaunchKernelGGL(fx_vbfix_kernel, grid, block, 0, st, jobs, vb_items, item_cap, hdr, first_item, bufB, dwv, vec_arena, vb_st, dbg);
    hipLaunchKernelGGL(fx_vbtrace_kernel, grid, block, 0, st, jobs, vb_items, item_cap, hdr, first_item, bufA, dwv, vb_st, dbg);
    return hipGetLastError();
}
extern "C" hipError_t fx_launch_vbfinish(unsigned first_wave, unsigned n_waves, hipStream_t st, const FxPayJob *jobs, const uint32_t *job_idx, FxBlockHdr *hdr,
                                         uint8_t *bufA, uint8_t *bufB, unsigned long long *dwv, const uint8_t *vec_arena, const uint32_t *vb_st, uint32_t *fb_list, uint32_t list_cap,
                                         uint8_t *out, FxOutRec *recs, FxBlockHdr *hdr_host)
{
    if (n_waves == 0) return hipSuccess;
    hipLaunchKernelGGL(fx_vbfinish_kernel, dim3(n_waves), dim3(64), 0, st, jobs, job_idx, hdr, first_wave, bufA, bufB, dwv, vec_arena, vb_st, fb_list, list_cap, out, recs, hdr_host);
    return hipGetLastError();
}

// ---- payload symbols to the host (fxrx_config.want_framesyms): as many as the block holds -- a count only the device knows,
// where a hipMemcpyAsync would have to move the arena's upper bound -- in 16-byte stores straight into pinned host memory
extern "C" __global__ __launch_bounds__(256)
void fx_symcopy_kernel(const FxBlockHdr *hdr, const float2 *sym, float2 *host)
{
    const size_t n16 = ((size_t)hdr->sym_total * sizeof(float2) + 15) / 16;
    const uint4 *src = reinterpret_cast<const uint4 *>(sym); uint4 *dst = reinterpret_cast<uint4 *>(host);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// ---- uploads as kernels.  A block's samples in page-locked host memory (the drop-in's ring, a caller's pinned buffer) and the
// block's own descriptors are read over the bus by the shader cores: a launch costs the host 3 us, where hipMemcpyAsync held the
// calling thread for 0.2 - 0.6 ms per 8-MB block (more with more blocks in flight) and 30 - 50 us per small copy -- with 2^20-sample
// blocks of one continuing stream the host was what limited the rate.  8-byte pieces (a float2), eight in flight per thread.
extern "C" __global__ __launch_bounds__(256)
void fx_upload_kernel(const unsigned long long *src, unsigned long long *dst, size_t n8)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n8; i += 8 * stride) {
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; k++) dst[i + k * stride] = v[k];
    }
    for (; i < n8; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
extern "C" hipError_t fx_launch_upload(hipStream_t st, const void *src, void *dst, size_t bytes, unsigned n_cus)
{
    const size_t n8 = (bytes + 7) / 8;
    if (n8 == 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<size_t>((n8 + 256 * 8 - 1) / (256 * 8), 4u * (size_t)n_cus);
    hipLaunchKernelGGL(fx_upload_kernel, dim3(grid ? grid : 1u), dim3(256), 0, st, reinterpret_cast<const unsigned long long *>(src), reinterpret_cast<unsigned long long *>(dst), n8);
    return hipGetLastError();
}
extern "C" __global__ void fx_copy_u32_kernel(const uint32_t *src, uint32_t *dst) { *dst = *src; }
extern "C" hipError_t fx_launch_copy_u32(hipStream_t st, const uint32_t *src, uint32_t *dst)
{
    hipLaunchKernelGGL(fx_copy_u32_kernel, dim3(1), dim3(1), 0, st, src, dst);
    return hipGetLastError();
}

extern "C" hipError_t fx_launch_symcopy(unsigned grid, hipStream_t st, const FxBlockHdr *hdr, const float2 *sym, float2 *host)
{
    hipLaunchKernelGGL(fx_symcopy_kernel, dim3(grid), dim3(256), 0, st, hdr, sym, host);
    return hipGetLastError();
}

// ===================================================================== payload: soft decisions (optional)
// One thread per payload symbol: the carrier-recovered symbol (fx_paypll_kernel left it in framesyms) -> bps soft values,
// 0 = surely 0 ... 255 = surely 1, MSB of the symbol first, written at the bit's position in the packet (8 soft values per
// packet byte).  soft = clamp(rint(127 + 16 gamma (d0 - d1))), gamma = 1.2 M, d0 / d1 = squared distance to the nearest point
// whose label has a 0 / a 1 at that bit: exhaustive over the M phases for PSK, per axis for ASK / QAM; differential PSK: the
// hard symbol's bits as 0 / 255.  Work items are the matched filter's (frame, 1024 symbols).
__device__ __forceinline__ uint8_t soft_byte(float d0, float d1, float gamma16)
{
    float t = rintf(fmaf(d0 - d1, gamma16, 127.0f));
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (uint8_t)t;
}
__device__ __forceinline__ void soft_axis(float v, unsigned nb, float al, float gamma16, uint8_t *soft)
{
    const unsigned L = 1u << nb;
    float d0[3] = { 1e30f, 1e30f, 1e30f }, d1[3] = { 1e30f, 1e30f, 1e30f };
    for (unsigned i = 0; i < L; i++) {
        const float dx = v - (2.0f * (float)i - (float)(L - 1)) * al, d = dx * dx;
        const unsigned g = gray_enc(i);
#pragma unroll
        for (unsigned b = 0; b < 3; b++) if (b < nb) { if ((g >> (nb - 1 - b)) & 1u) d1[b] = fminf(d1[b], d); else d0[b] = fminf(d0[b], d); }
    }
#pragma unroll
    for (unsigned b = 0; b < 3; b++) if (b < nb) soft[b] = soft_byte(d0[b], d1[b], gamma16);
}

extern "C" __global__ __launch_bounds__(256)
void fx_softdemod_kernel(const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr, const float2 *framesyms,
                         const uint8_t *hard, uint8_t *soft_arena, const FxTables *T)
{
    const uint32_t nitems = hdr->n_mfblk;
    const float2 *sc = T->sc;
    for (uint32_t bi = blockIdx.x; bi < nitems; bi += gridDim.x) {
        const FxPayJob job = jobs[blk_job[bi]];
        const uint32_t c0 = blk_c0[bi], ns = min(1024u, job.nsym - c0);
        const unsigned ms = job.ms, bps = job.bps;
        const float gamma16 = 1.2f * (float)(1u << bps) * 16.0f;
        uint8_t *S = soft_arena + 8 * (size_t)job.byte_off;
        const uint32_t nbits = 8u * job.l1;
        for (uint32_t i = threadIdx.x; i < ns; i += blockDim.x) {
            const uint32_t c = c0 + i;
            const float2 r = framesyms[(size_t)job.sym_off + c];
            uint8_t sb[8];
            switch (ms) {
            case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8: {
                const unsigned hsym = hard[(size_t)job.sym_off + c];
                for (unsigned b = 0; b < bps; b++) sb[b] = ((hsym >> (bps - 1 - b)) & 1u) ? 255 : 0;
                break; }
            case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16: {
                float d0[4] = { 1e30f, 1e30f, 1e30f, 1e30f }, d1[4] = { 1e30f, 1e30f, 1e30f, 1e30f };
                for (unsigned k = 0; k < (1u << bps); k++) {
                    const float2 p = psk_point(k, bps, sc);
                    const float dx = r.x - p.x, dy = r.y - p.y, d = fmaf(dx, dx, dy * dy);
                    const unsigned g = gray_enc(k);
#pragma unroll
                    for (unsigned b = 0; b < 4; b++) if (b < bps) { if ((g >> (bps - 1 - b)) & 1u) d1[b] = fminf(d1[b], d); else d0[b] = fminf(d0[b], d); }
                }
#pragma unroll
                for (unsigned b = 0; b < 4; b++) if (b < bps) sb[b] = soft_byte(d0[b], d1[b], gamma16);
                break; }
            case FX_MODEM_ASK4: soft_axis(r.x, 2, 0.447213595f, gamma16, sb); break;
            case FX_MODEM_QPSK: soft_axis(-r.y, 1, 0.70710678118654752f, gamma16, sb); soft_axis(-r.x, 1, 0.70710678118654752f, gamma16, sb + 1); break;
            default: {
                unsigned mi, mq; float al;
                if (ms == FX_MODEM_QAM16) { mi = 2; mq = 2; al = 0.316227766f; }
                else if (ms == FX_MODEM_QAM32) { mi = 3; mq = 2; al = 0.196116135f; }
                else { mi = 3; mq = 3; al = 0.154303350f; }
                soft_axis(r.x, mi, al, gamma16, sb); soft_axis(r.y, mq, al, gamma16, sb + mi);
                break; }
            }
            for (unsigned b = 0; b < bps; b++) { const uint32_t q = c * bps + b; if (q < nbits) S[q] = sb[b]; }
        }
    }
}

extern "C" hipError_t fx_launch_softdemod(unsigned grid, hipStream_t st, const FxPayJob *jobs, const uint32_t *blk_job, const uint32_t *blk_c0, const FxBlockHdr *hdr,
                                          const float2 *framesyms, const uint8_t *hard, uint8_t *soft_arena, const FxTables *T)
{
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(fx_softdemod_kernel, dim3(grid), dim3(256), 0, st, jobs, blk_job, blk_c0, hdr, framesyms, hard, soft_arena, T);
    return hipGetLastError();
}

// ===================================================================== frame generator (flex_tx counterpart)
// One thread per symbol n of a frame: constellation point of the symbol (preamble from the table, header = pilots +
// QPSK of the encoded header packet, payload from its symbol indices -- the indices come from fx_txenc_kernel), then
// the two output samples y[2n+i] = sum_t h[i+2t] x[n-t], t ascending,
// exactly as the host generator and the oracle accumulate them.  Symbols are staged through LDS (256 + 14 per tile).
#define TX_TILE 256
__device__ __forceinline__ float2 tx_point(unsigned ms, unsigned v, const float2 *sc)
{
    const unsigned bps = modem_bps(ms);
    switch (ms) {
    case FX_MODEM_QPSK: { const float h = 0.70710678118654752f; return make_float2((v & 1) ? -h : h, (v & 2) ? -h : h); }
    case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16:
    case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8:          // v is the phase index (Gray-decoded, accumulated)
        if (bps <= 2) {
            const unsigned q = bps == 1 ? 2u * (v & 1u) : (v & 3u);
            return make_float2(q == 0 ? 1.0f : (q == 2 ? -1.0f : 0.0f), q == 1 ? 1.0f : (q == 3 ? -1.0f : 0.0f));
        }
        return sc[(v << (32 - bps)) >> 22];
    case FX_MODEM_ASK4: return make_float2((2.0f * (float)v - 3.0f) * 0.447213595f, 0.0f);   // v Gray-decoded
    default: {                                                                                  // v = si << mq | sq, Gray-decoded
        unsigned mi, mq; float al;
        if (ms == FX_MODEM_QAM16) { mi = 2; mq = 2; al = 0.316227766f; }
        else if (ms == FX_MODEM_QAM32) { mi = 3; mq = 2; al = 0.196116135f; }
        else { mi = 3; mq = 3; al = 0.154303350f; }
        const unsigned si = v >> mq, sq = v & ((1u << mq) - 1u);
        return make_float2((2.0f * (float)si - (float)((1u << mi) - 1u)) * al, (2.0f * (float)sq - (float)((1u << mq) - 1u)) * al);
    }
    }
}

extern "C" __global__ __launch_bounds__(TX_TILE)
void fx_txgen_kernel(const FxTxJob *jobs, const uint32_t *tile_job, const uint32_t *tile_n0, const uint8_t *head_idx, const uint8_t *pay_idx,
                     const FxTxTables *T, float2 *out)
{
    const float2 *sc = T->sc;
    __shared__ float2 xs[TX_TILE + 14];
    __shared__ float h[32];
    const FxTxJob job = jobs[tile_job[blockIdx.x]];
    const uint32_t n0 = tile_n0[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < 29) h[tid] = job.taps[tid];
    const uint32_t nhead = FX_PN_LEN + FX_HDR_SYM;
    for (int i = tid; i < TX_TILE + 14; i += TX_TILE) {
        const int64_t n = (int64_t)n0 - 14 + i;
        float2 v = make_float2(0.0f, 0.0f);
        if (n >= 0 && n < (int64_t)FX_PN_LEN) v = T->pn[n];
        else if (n >= (int64_t)FX_PN_LEN && n < (int64_t)nhead) {                // header: a pilot every 16th symbol, QPSK in between
            const uint32_t hi = (uint32_t)n - FX_PN_LEN;
            if (hi % FX_PILOT_SPACING == 0) v = T->pilots[hi / FX_PILOT_SPACING];
            else v = tx_point(FX_MODEM_QPSK, head_idx[(size_t)job.head_off + hi - 1u - hi / FX_PILOT_SPACING], sc);
        }
        else if (n >= (int64_t)nhead && n < (int64_t)(nhead + job.npay)) v = tx_point(job.ms, pay_idx[(size_t)job.idx_off + (size_t)(n - nhead)], sc);
        xs[i] = v;
    }
    __syncthreads();
    const uint32_t n = n0 + tid;
    if (n >= job.nsym) return;
    float2 y[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        float ar = 0.0f, ai = 0.0f;
#pragma unroll
        for (int t = 0; t < 15; t++) {
            const int hi = i + 2 * t;
            if (hi > 28 || (uint32_t)t > n) continue;
            const float2 w = xs[tid + 14 - t];
            ar = fmaf(h[hi], w.x, ar); ai = fmaf(h[hi], w.y, ai);
        }
        y[i] = make_float2(ar, ai);
    }
    float4 *o = reinterpret_cast<float4 *>(out + job.out_off + 2ull * n);       // out_off is even: 16-byte aligned
    if ((job.out_off & 1ull) == 0) *o = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
    else { out[job.out_off + 2ull * n] = y[0]; out[job.out_off + 2ull * n + 1] = y[1]; }
}

extern "C" hipError_t fx_launch_txgen(unsigned ntiles, hipStream_t st, const FxTxJob *jobs, const uint32_t *tile_job, const uint32_t *tile_n0,
                                      const uint8_t *head_idx, const uint8_t *pay_idx, const FxTxTables *T, float2 *out)
{
    hipLaunchKernelGGL(fx_txgen_kernel, dim3(ntiles), dim3(TX_TILE), 0, st, jobs, tile_job, tile_n0, head_idx, pay_idx, T, out);
    return hipGetLastError();
}

// ----- synthetic channel (test / bench signal source): carrier offset and phase by the closed-form 32-bit NCO the
// receiver uses, gain, and white Gaussian noise from a counter-based generator -- Philox-4x32-10 keyed by the stream's seed,
// counter = sample pair index, Box-Muller on its four words -- so that every sample of every stream is reproducible from
// (seed, index) alone, on any grid.  One thread per pair of samples (16-byte accesses); grid.y = stream.
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t h0 = __umulhi(0xD2511F53u, c.x), l0 = 0xD2511F53u * c.x, h1 = __umulhi(0xCD9E8D57u, c.z), l1 = 0xCD9E8D57u * c.z;
        c = make_uint4(h1 ^ c.y ^ k.x, l1, h0 ^ c.w ^ k.y, l0);
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}
__device__ __forceinline__ float2 gauss_pair(uint32_t a, uint32_t b)
{
    const float u1 = ((float)(a >> 8) + 1.0f) * 5.9604645e-8f;            // (0, 1]
    const float u2 = (float)(b >> 8) * 5.9604645e-8f;                     // [0, 1)
    const float r = sqrtf(-2.0f * logf(u1));
    float sn, cs; sincosf(6.28318531f * u2, &sn, &cs);
    return make_float2(r * cs, r * sn);
}
extern "C" __global__ __launch_bounds__(256)
void fx_channel_kernel(float2 *x, unsigned long long n_per_stream, const FxChannel *chs, const FxTxTables *T)
{
    const FxChannel ch = chs[blockIdx.y];
    float4 *xs = reinterpret_cast<float4 *>(x + (size_t)blockIdx.y * n_per_stream);     // (n_per_stream is even: 16-byte aligned)
    const unsigned long long npair = n_per_stream / 2;
    const uint2 key = make_uint2((uint32_t)ch.seed, (uint32_t)(ch.seed >> 32));
    for (unsigned long long p = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; p < npair; p += (unsigned long long)gridDim.x * blockDim.x) {
        float4 v = xs[p];
        const uint4 rnd = philox4x32_10(make_uint4((uint32_t)p, (uint32_t)(p >> 32), 0u, 0u), key);
        const float2 w0 = gauss_pair(rnd.x, rnd.y), w1 = gauss_pair(rnd.z, rnd.w);
        float c0, s0, c1, s1;
        const uint32_t th = ch.th0 + ch.dl * (uint32_t)(2ull * p);
        sincos_u32(th, T->sc, c0, s0); sincos_u32(th + ch.dl, T->sc, c1, s1);
        float4 y;
        y.x = fmaf(ch.sigma, w0.x, ch.gain * (v.x * c0 - v.y * s0)); y.y = fmaf(ch.sigma, w0.y, ch.gain * (v.x * s0 + v.y * c0));
        y.z = fmaf(ch.sigma, w1.x, ch.gain * (v.z * c1 - v.w * s1)); y.w = fmaf(ch.sigma, w1.y, ch.gain * (v.z * s1 + v.w * c1));
        xs[p] = y;
    }
}
extern "C" hipError_t fx_launch_channel(hipStream_t st, float2 *x, unsigned n_streams, unsigned long long n_per_stream, const FxChannel *chs, const FxTxTables *T)
{
    if (n_streams == 0 || n_per_stream == 0) return hipSuccess;
    const unsigned long long npair = n_per_stream / 2;
    const unsigned gx = (unsigned)std::min<unsigned long long>((npair + 255) / 256, 4096ull);
    hipLaunchKernelGGL(fx_channel_kernel, dim3(gx, n_streams), dim3(256), 0, st, x, n_per_stream, chs, T);
    return hipGetLastError();
}

// ----- packet encoder: CRC, whitening, two code stages (convolutional with puncturing, Hamming, Golay, SECDED,
// Reed-Solomon), two interleavers, bit packing, Gray / DPSK index arithmetic -- what fx_codec.hpp::packet_encode +
// FrameGen::payload_indices do on the host, one wave per frame.
__device__ __forceinline__ unsigned tx_gray_dec(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }

// one code stage: n bytes at src -> fec_enc_len(fs, n) bytes at dst (src has >= 8 zero bytes of slack behind it)
__device__ __forceinline__ void tx_fec_encode(unsigned fs, uint32_t n, const uint8_t *src, uint8_t *dst, int lane, const FxTxTables *T)
{
    const uint32_t el = fec_enc_len(fs, n);
    const int p = conv_p(fs);
    unsigned bk, bn;
    if (p) {                                     // K=7 (0x6d, 0x4f): every coded bit is a parity of a 7-bit window; a byte per lane
        unsigned pa, pb;
        switch (p) {
        case 2: pa = 0x3; pb = 0x1; break;   case 3: pa = 0x3; pb = 0x5; break;   case 4: pa = 0xf; pb = 0x1; break;
        case 5: pa = 0xb; pb = 0x15; break;  case 6: pa = 0x17; pb = 0x29; break; case 7: pa = 0x2f; pb = 0x51; break;
        default: pa = 1; pb = 1; break;
        }
        unsigned ncol = 0, slot_col[8], slot_gen[8];        // kept bits of one puncturing period: slot -> (column, generator)
        for (int c = 0; c < p; c++) {
            if ((pa >> c) & 1) { slot_col[ncol] = (unsigned)c; slot_gen[ncol] = 0x6d; ncol++; }
            if ((pb >> c) & 1) { slot_col[ncol] = (unsigned)c; slot_gen[ncol] = 0x4f; ncol++; }
        }
        const uint32_t Tn = 8 * n + 6;
        const uint32_t nbits = p == 1 ? 2 * Tn : Tn + (Tn + (uint32_t)p - 1) / (uint32_t)p;
        for (uint32_t j = lane; j < el; j += DEC_THREADS) {
            unsigned v = 0;
            for (int b = 0; b < 8; b++) {
                const uint32_t q = 8 * j + b;
                unsigned bit = 0;
                if (q < nbits) {
                    const uint32_t g = q / ncol, r = q % ncol;
                    const uint32_t t = g * (uint32_t)p + slot_col[r];
                    unsigned sr = 0;                        // input bits t-6 .. t, bit t in the LSB; outside the message: 0
                    for (int i = 6; i >= 0; i--) { const int64_t tt = (int64_t)t - i; sr = (sr << 1) | ((tt >= 0 && tt < (int64_t)(8 * n)) ? getbit(src, (uint32_t)tt) : 0u); }
                    bit = __popc(sr & slot_gen[r]) & 1u;
                }
                v = (v << 1) | bit;
            }
            dst[j] = (uint8_t)v;
        }
    } else if (blk_spec(fs, bk, bn)) {           // bit-packed block codes: k-bit blocks -> n-bit codewords back to back
        const uint32_t nb = (8 * n + bk - 1) / bk;
        for (uint32_t j = lane; j < el; j += DEC_THREADS) {
            unsigned v = 0;
            for (int b = 0; b < 8; b++) {
                const uint32_t q = 8 * j + b, cwi = q / bn, pos = q % bn;
                unsigned bit = 0;
                if (cwi < nb) {
                    unsigned d = 0;
                    for (unsigned i = 0; i < bk; i++) { const uint32_t qq = cwi * bk + i; d = (d << 1) | (qq < 8 * n ? getbit(src, qq) : 0u); }
                    const uint32_t cw = fs == FX_FEC_HAMMING74 ? T->h74enc[d] : (fs == FX_FEC_HAMMING128 ? T->h128enc[d] : T->golenc[d]);
                    bit = (cw >> (bn - 1 - pos)) & 1u;
                }
                v = (v << 1) | bit;
            }
            dst[j] = (uint8_t)v;
        }
    } else if (fs == FX_FEC_HAMMING84) {
        for (uint32_t i = lane; i < n; i += DEC_THREADS) { dst[2 * i] = T->h84enc[src[i] >> 4]; dst[2 * i + 1] = T->h84enc[src[i] & 15]; }
    } else if (fs == FX_FEC_SECDED7264 || fs == FX_FEC_SECDED2216 || fs == FX_FEC_SECDED3932) {
        const uint32_t nd = fs == FX_FEC_SECDED7264 ? 8u : (fs == FX_FEC_SECDED2216 ? 2u : 4u);
        const uint8_t *col = fs == FX_FEC_SECDED7264 ? T->sdcol : (fs == FX_FEC_SECDED2216 ? T->sd22col : T->sd39col);
        const uint32_t nblk = (n + nd - 1) / nd;
        for (uint32_t blk = lane; blk < nblk; blk += DEC_THREADS) {
            const uint32_t nbytes = min(nd, n - nd * blk);
            uint8_t par = 0;
            for (uint32_t j = 0; j < 8 * nbytes; j++) if (src[nd * blk + (j >> 3)] & (0x80u >> (j & 7))) par ^= col[j];
            dst[(nd + 1) * blk] = par;
            for (uint32_t j = 0; j < nbytes; j++) dst[(nd + 1) * blk + 1 + j] = src[nd * blk + j];
        }
    } else if (fs == FX_FEC_RS_M8) {             // RS(255,223): equal shortened blocks, a lane per block
        const uint32_t nb = max(1u, (n + 222u) / 223u), dl = (n + nb - 1) / nb;
        for (uint32_t b = lane; b < nb; b += DEC_THREADS) {
            uint8_t par[32];
#pragma unroll
            for (int k = 0; k < 32; k++) par[k] = 0;
            const uint32_t off = b * dl, len = off < n ? min(n - off, dl) : 0u;
            uint8_t *o = dst + b * (dl + 32);
            for (uint32_t i = 0; i < dl; i++) {
                const uint8_t di = i < len ? src[off + i] : (uint8_t)0;
                o[i] = di;
                const uint8_t fb = (uint8_t)(di ^ par[31]);
                const unsigned lf = T->rslog[fb];
#pragma unroll
                for (int k = 31; k > 0; k--) par[k] = (uint8_t)(par[k - 1] ^ ((fb && T->rsgen[k]) ? T->rsexp[lf + T->rslog[T->rsgen[k]]] : 0));
                par[0] = (fb && T->rsgen[0]) ? T->rsexp[lf + T->rslog[T->rsgen[0]]] : (uint8_t)0;
            }
#pragma unroll
            for (int k = 0; k < 32; k++) o[dl + k] = par[31 - k];
        }
    } else {
        for (uint32_t j = lane; j < n; j += DEC_THREADS) dst[j] = src[j];
    }
}

extern "C" __global__ __launch_bounds__(DEC_THREADS)
void fx_txenc_kernel(const FxTxEncJob *jobs, const uint8_t *pay, const uint32_t *perm_arena, uint8_t *bufA, uint8_t *bufB, uint8_t *pay_idx,
                     const FxTxTables *T)
{
    const FxTxEncJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    uint8_t *A = bufA + job.buf_off, *B = bufB + job.buf_off;
    auto sync = [&]() { __threadfence_block(); __builtin_amdgcn_wave_barrier(); };
    // message + CRC, whitened
    for (uint32_t j = lane; j < job.n; j += DEC_THREADS) A[j] = pay[job.pay_off + j];
    sync();
    uint32_t key = 0;
    switch (job.check) {
    case FX_CRC_CHECKSUM: {
        uint32_t sm = 0;
        for (uint32_t j = lane; j < job.n; j += DEC_THREADS) sm += A[j];
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sm += (uint32_t)__shfl_xor((int)sm, m, 64);
        key = (~sm + 1u) & 0xff; break; }
    case FX_CRC_8:  key = crc_wave(0xE0u, 0xFFu, A, job.n, lane); break;
    case FX_CRC_16: key = crc_wave(0xA001u, 0xFFFFu, A, job.n, lane); break;
    case FX_CRC_24: key = crc_wave(0xD3B6BAu, 0xFFFFFFu, A, job.n, lane); break;
    case FX_CRC_32: key = crc_wave(0xEDB88320u, 0xFFFFFFFFu, A, job.n, lane); break;
    default: key = 0; break;
    }
    const uint32_t cl = job.k - job.n;
    if ((uint32_t)lane < cl) A[job.n + cl - 1 - lane] = (uint8_t)(key >> (8 * lane));
    sync();
    { const uint8_t mask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 }; for (uint32_t j = lane; j < job.k; j += DEC_THREADS) A[j] ^= mask[j & 3]; }
    if (lane < 8) A[job.k + lane] = 0;                                          // slack behind the message reads as zero
    sync();
    tx_fec_encode(job.fec0, job.k, A, B, lane, T);
    sync();
    permute_bits(B, A, perm_arena + job.perm0_off, job.l0, lane);               // interleaver of the first stage
    if (lane < 8) A[job.l0 + lane] = 0;
    sync();
    tx_fec_encode(job.fec1, job.l0, A, B, lane, T);
    sync();
    permute_bits(B, A, perm_arena + job.perm1_off, job.l1, lane);               // interleaver of the second stage
    if (lane < 8) A[job.l1 + lane] = 0;
    sync();
    // bits -> modem words -> constellation indices (Gray decoding; DPSK: running sum of the decoded words mod M)
    const unsigned bps = modem_bps(job.ms), M1 = (1u << bps) - 1u;
    const bool dpsk = job.ms == FX_MODEM_DPSK2 || job.ms == FX_MODEM_DPSK4 || job.ms == FX_MODEM_DPSK8;
    unsigned carry = 0;
    for (uint32_t j0 = 0; j0 < job.npay; j0 += DEC_THREADS) {
        const uint32_t j = j0 + lane;
        unsigned w = 0;
        if (j < job.npay) for (unsigned b = 0; b < bps; b++) { const uint32_t q = j * bps + b; w = (w << 1) | (q < 8 * job.l1 ? getbit(A, q) : 0u); }
        unsigned v;
        switch (job.ms) {
        case FX_MODEM_QPSK: v = w; break;
        case FX_MODEM_PSK2: case FX_MODEM_PSK4: case FX_MODEM_PSK8: case FX_MODEM_PSK16: case FX_MODEM_ASK4:
        case FX_MODEM_DPSK2: case FX_MODEM_DPSK4: case FX_MODEM_DPSK8: v = tx_gray_dec(w); break;
        default: { const unsigned mq = job.ms == FX_MODEM_QAM64 ? 3u : 2u; v = (tx_gray_dec(w >> mq) << mq) | tx_gray_dec(w & ((1u << mq) - 1u)); }
        }
        if (dpsk) {
            unsigned s_ = v;                                                    // inclusive scan over the 64 symbols of this pass
#pragma unroll
            for (int m = 1; m < 64; m <<= 1) { const unsigned o = (unsigned)__shfl_up((int)s_, m, 64); if (lane >= m) s_ += o; }
            v = (s_ + carry) & M1;
            carry = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
        }
        if (j < job.npay) pay_idx[job.idx_off + j] = (uint8_t)v;
    }
}

extern "C" hipError_t fx_launch_txenc(unsigned njobs, hipStream_t st, const FxTxEncJob *jobs, const uint8_t *pay, const uint32_t *perm_arena,
                                      uint8_t *bufA, uint8_t *bufB, uint8_t *pay_idx, const FxTxTables *T)
{
    hipLaunchKernelGGL(fx_txenc_kernel, dim3(njobs), dim3(DEC_THREADS), 0, st, jobs, pay, perm_arena, bufA, bufB, pay_idx, T);
    return hipGetLastError();
}
