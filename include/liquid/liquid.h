/*
 * liquid/liquid.h -- shim that satisfies `#include <liquid/liquid.h>` of gr::liquiddsp's three block implementations
 * (/root/reference/lib/flex_rx_impl.h:25, /root/reference/lib/frame_detector_cc_impl.h:25,
 * /root/reference/lib/flex_tx_impl.h:25) when they are built against libfxrx.so instead of libliquid.
 *
 * Only what those files use is declared: the flexframesync / qdetector_cccf / msequence / flexframegen entry points
 * (include/fxrx.h, layer 1) and the LIQUID_* constants of their switch statements
 * (/root/reference/lib/flex_rx_impl.cc:77-171, /root/reference/lib/flex_tx_impl.cc:52,79-178,
 * /root/reference/lib/frame_detector_cc_impl.cc:54).
 *
 * Complex samples: liquid declares `liquid_float_complex` as `float complex` in C and as `std::complex<float>` when
 * <complex> was included first (C++); gr_complex is std::complex<float>.  The shim makes fxrx.h use that very type in
 * its prototypes and in framesyncstats_s, so the reference's calls -- `flexframesync_execute(d_fs, in, d_inbuf_len)` with a
 * gr_complex*, `qdetector_cccf_execute(d_detector, in[i])` with a gr_complex by value, `info->_frame_symbols =
 * _stats.framesyms` -- compile as written.  All three spellings are two packed floats and are passed identically
 * (x86-64 SysV: one SSE eightbyte), so the binary interface is the one libfxrx.so exports.
 * tests/cpp/test_reference_callsites.cpp compiles and runs those call sequences through this header.
 */
#ifndef FXRX_LIQUID_SHIM_H
#define FXRX_LIQUID_SHIM_H

#ifdef __cplusplus
#include <complex>
typedef std::complex<float> liquid_float_complex;
#else
#include <complex.h>
typedef float _Complex liquid_float_complex;
#endif

#define FXRX_COMPLEX_TYPE liquid_float_complex
#include "../fxrx.h"

/* values as stored in the flexframe header [RECALLED liquid.h v1.3.x]; the same numbers as csrc/fx_common.h */
typedef enum {
    LIQUID_CRC_UNKNOWN = 0, LIQUID_CRC_NONE, LIQUID_CRC_CHECKSUM, LIQUID_CRC_8, LIQUID_CRC_16, LIQUID_CRC_24, LIQUID_CRC_32
} crc_scheme;
typedef enum {
    LIQUID_FEC_UNKNOWN = 0, LIQUID_FEC_NONE = 1, LIQUID_FEC_REP3 = 2, LIQUID_FEC_REP5 = 3, LIQUID_FEC_HAMMING74 = 4,
    LIQUID_FEC_HAMMING84 = 5, LIQUID_FEC_HAMMING128 = 6, LIQUID_FEC_GOLAY2412 = 7, LIQUID_FEC_SECDED2216 = 8,
    LIQUID_FEC_SECDED3932 = 9, LIQUID_FEC_SECDED7264 = 10, LIQUID_FEC_CONV_V27 = 11, LIQUID_FEC_CONV_V29 = 12,
    LIQUID_FEC_CONV_V39 = 13, LIQUID_FEC_CONV_V615 = 14, LIQUID_FEC_CONV_V27P23 = 15, LIQUID_FEC_CONV_V27P34 = 16,
    LIQUID_FEC_CONV_V27P45 = 17, LIQUID_FEC_CONV_V27P56 = 18, LIQUID_FEC_CONV_V27P67 = 19, LIQUID_FEC_CONV_V27P78 = 20,
    LIQUID_FEC_RS_M8 = 27
} fec_scheme;
typedef enum {
    LIQUID_MODEM_UNKNOWN = 0, LIQUID_MODEM_PSK2 = 1, LIQUID_MODEM_PSK4 = 2, LIQUID_MODEM_PSK8 = 3, LIQUID_MODEM_PSK16 = 4,
    LIQUID_MODEM_DPSK2 = 9, LIQUID_MODEM_DPSK4 = 10, LIQUID_MODEM_DPSK8 = 11, LIQUID_MODEM_ASK4 = 18,
    LIQUID_MODEM_QAM16 = 27, LIQUID_MODEM_QAM32 = 28, LIQUID_MODEM_QAM64 = 29, LIQUID_MODEM_QPSK = 40
} modulation_scheme;

#endif
