/* TEST INFRASTRUCTURE -- CPU oracle ("sdro") for the sdrbase/dsp RX hot path.
 *
 * A from-scratch restatement, in plain C, of what the reference computes on the path named by
 * BASELINE.json / SURVEY.md §8.  It is the checker for the HIP engine: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (sdrangel_amd/, libsdrx.so) never links, imports or calls anything in oracle/.
 *
 * Pinning: every function here is checked bit-for-bit against the reference's own classes
 * compiled from /root/reference (oracle/ref_shim.cpp -> oracle/_ref/libsdrref.so) by
 * tests/test_oracle_vs_ref.py (runs where the reference is present) and against the golden
 * vectors in tests/golden/ that were generated from that same compiled reference
 * (tests/golden/make_golden.py).  The reference itself ships no tests or vectors.
 */
#ifndef SDRO_H
#define SDRO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* fcpos coding = the device plugins' m_fcPos: 0 infradyne, 1 supradyne, 2 centre. */
enum { SDRO_FC_INF = 0, SDRO_FC_SUP = 1, SDRO_FC_CEN = 2 };
/* channelizer stage modes = DownChannelizer::FilterStage::Mode (downchannelizer.h:76-80). */
enum { SDRO_MODE_CENTER = 0, SDRO_MODE_LOWER = 1, SDRO_MODE_UPPER = 2 };

/* ---- Decimators<qint32,qint16,16,InputBits> (decimators.h:279-341) ---- */
typedef struct sdro_decim sdro_decim;
sdro_decim* sdro_decim_new(int log2_decim, int fcpos, int input_bits);
void        sdro_decim_free(sdro_decim*);
void        sdro_decim_reset(sdro_decim*);
/* the next call runs decimate{2^log2}_{fcpos} on the SAME six stage states (one Decimators object, several variants) */
void        sdro_decim_switch(sdro_decim*, int log2_decim, int fcpos);
/* iq: interleaved int16 I,Q; n_int16 = number of int16 (the reference's `len`).  Whole groups
 * only, tail dropped (decimators.h:3492).  Returns #complex outputs written to out_iq. */
int32_t     sdro_decim_process(sdro_decim*, const int16_t* iq, int32_t n_int16, int16_t* out_iq);
/* DecimatorsU<qint32, quint8, 16, 8, Shift> (decimatorsu.h:175-216): unsigned 8-bit I/Q, value = buf - Shift */
sdro_decim* sdro_decimu_new(int log2_decim, int fcpos, int shift);
int32_t     sdro_decimu_process(sdro_decim*, const uint8_t* iq, int32_t n_uint8, int16_t* out_iq);
/* #int16 consumed per loop iteration of the reference function (its `pos +=` stride). */
int32_t     sdro_decim_group_int16(int log2_decim, int fcpos);

/* ---- DownChannelizer (downchannelizer.cpp:50-91,157-189,250-287) ---- */
/* float bisection; writes up to 32 modes; returns n_stages. */
int32_t sdro_chan_plan(int32_t in_rate, int32_t req_rate, int32_t req_fc,
                       uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs);
typedef struct sdro_chain sdro_chain;
sdro_chain* sdro_chain_new(int32_t n_stages, const uint8_t* modes);
void        sdro_chain_free(sdro_chain*);
void        sdro_chain_reset(sdro_chain*);
int64_t     sdro_chain_feed(sdro_chain*, const int16_t* iq, int64_t n_cplx, int16_t* out_iq);

/* ---- float back-end: NCO, Interpolator, g_fft, fftfilt, PhaseDiscriminators ---- */
void    sdro_nco_table(float* tbl4096);                                  /* nco.cpp:30-39 */
int32_t sdro_nco_inc(float freq, float rate);                            /* nco.cpp:48-52 */

typedef struct sdro_backend sdro_backend;
/* NCO(setFreq(nco_freq,in_rate)) -> Interpolator::create(phase_steps,in_rate,cutoff,tpp) ->
 * decimate with distance += in_rate/out_rate  (nfmdemod.cpp:150-160, 453-476). */
sdro_backend* sdro_backend_new(float nco_freq, float in_rate, float out_rate,
                               int32_t phase_steps, float cutoff, float taps_per_phase);
void    sdro_backend_free(sdro_backend*);
int64_t sdro_backend_feed(sdro_backend*, const int16_t* iq, int64_t n_cplx, float* out_iq);
int32_t sdro_backend_ntaps(const sdro_backend*);                         /* taps per phase */
const float* sdro_backend_taps(const sdro_backend*);                     /* [phase][ntaps] */

void    sdro_gfft(float* iq, int32_t n, int32_t inverse);                /* gfft.h:3308-3330 */

typedef struct sdro_fftfilt sdro_fftfilt;
/* f1 >= 0: fftfilt(f1, f2, len) (fftfilt.cpp:77-84,108-146); f1 < 0: fftfilt(f2, len) = the DSB low pass (:86-93,149-170) */
sdro_fftfilt* sdro_fftfilt_new(float f1, float f2, int32_t len);
void    sdro_fftfilt_free(sdro_fftfilt*);
const float* sdro_fftfilt_filter(const sdro_fftfilt*);                   /* len complex */
/* fftfilt(fin, len) + create_asym_filter(fopp, fin) (fftfilt.cpp:172-225; ATV demod) */
sdro_fftfilt* sdro_fftfilt_new_asym(float fopp, float fin, int32_t len);
const float* sdro_fftfilt_filter_opp(const sdro_fftfilt*);               /* len complex */
/* mode 0 runFilt, 1 runSSB usb, 2 runSSB lsb, 3 runDSB (fftfilt.cpp:261-361), 4 runAsym usb, 5 runAsym lsb (:363-402) */
int64_t sdro_fftfilt_run(sdro_fftfilt*, int32_t mode, const float* in_iq, int64_t n, float* out_iq);

/* kind 0: phaseDiscriminatorDelta (phasediscri.h:61-78); 1: phaseDiscriminator (:50-55) */
void    sdro_discri(int32_t kind, float fm_scaling, const float* in_iq, int64_t n, float* out);

/* Lowpass<Real> (kind 0: create(ntaps, rate, f1)) / Bandpass<Real> (kind 1: create(ntaps, rate, f1, f2)), lowpass.h / bandpass.h */
typedef struct sdro_fir sdro_fir;
sdro_fir* sdro_fir_new(int32_t kind, int32_t ntaps, double rate, double f1, double f2);
void    sdro_fir_free(sdro_fir*);
int32_t sdro_fir_taps(const sdro_fir*, float* out);                      /* ntaps/2 + 1 folded taps */
void    sdro_fir_run(sdro_fir*, const float* in, int64_t n, float* out); /* streaming: state carried */

/* ---- DC offset correction of the device stream: DSPDeviceSourceEngine::iqCorrections(.., false)
 * (dspdevicesourceengine.cpp:175-181,255-259; MovingAverageUtil<int32_t,int64_t,1024>) ---- */
typedef struct sdro_dccorr sdro_dccorr;
sdro_dccorr* sdro_dccorr_new(void);
void    sdro_dccorr_free(sdro_dccorr*);
void    sdro_dccorr_process(sdro_dccorr*, const int16_t* iq, int64_t n_cplx, int16_t* out_iq);

/* DSPDeviceSourceEngine::iqCorrections(begin, end, imbalanceCorrection = true), float flavour (IMBALANCE_INT undefined):
 * dspdevicesourceengine.cpp:175-181, 217-253.  Strict IEEE, scalar (SURVEY finding 6). */
typedef struct sdro_iqimb sdro_iqimb;
sdro_iqimb* sdro_iqimb_new(void);
void    sdro_iqimb_free(sdro_iqimb*);
void    sdro_iqimb_process(sdro_iqimb*, const int16_t* iq, int64_t n_cplx, int16_t* out_iq);

/* ---- float half-band decimators: DecimatorsFI / FF / IF over IntHalfbandFilterEOF<64> (oracle/sdro_fdecim.c) ----
 * in_kind 0: float I/Q, 1: int16 I/Q (DecimatorsIF<qint16,input_bits>); out_kind 0: int16 Sample (FI), 1: float (FF, IF).
 * n_elems = the reference's nbIAndQ; returns #complex outputs (whole groups only, tail dropped). */
/* 24-bit sample build (SDR_RX_SAMPLE_24BIT): Decimators<qint32,qint16,24,{8,12,16}> and the DownChannelizer stage chain on
 * Sample{qint32,qint32}; objects are the same structs as the 16-bit flavour (free with sdro_decim_free / sdro_chain_free) */
sdro_decim* sdro_decim24_new(int log2, int fcpos, int bits);
int32_t sdro_decim24_process(sdro_decim*, const int16_t* iq, int32_t n_int16, int32_t* out_iq);
sdro_chain* sdro_chain24_new(int32_t n_stages, const uint8_t* modes);
int64_t sdro_chain24_feed(sdro_chain*, const int32_t* iq, int64_t n_cplx, int32_t* out_iq);

/* audio-rate tails of the NFM / SSB demodulators (oracle/sdro_audio.c) */
typedef struct sdro_nfmtail sdro_nfmtail;
sdro_nfmtail* sdro_nfmtail_new(int32_t audio_rate, float fm_scaling, float squelch_level, int32_t squelch_gate, float volume, float af_bandwidth);
void    sdro_nfmtail_free(sdro_nfmtail*);
void    sdro_nfmtail_process(sdro_nfmtail*, const float* ci, int64_t n, int16_t* audio);
typedef struct sdro_ssbtail sdro_ssbtail;
sdro_ssbtail* sdro_ssbtail_new(int32_t agc_active, int32_t agc_nb_samples, double agc_threshold, int32_t agc_threshold_enable,
                               int32_t agc_gate, int32_t agc_clamping, float volume);
void    sdro_ssbtail_free(sdro_ssbtail*);
void    sdro_ssbtail_process(sdro_ssbtail*, const float* sideband, int64_t n, int16_t* audio);

/* IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h), Order 2..8 */
typedef struct sdro_iir sdro_iir;
sdro_iir* sdro_iir_new(int32_t order, const float* a, const float* b);
void    sdro_iir_free(sdro_iir*);
void    sdro_iir_run(sdro_iir*, const float* in, int64_t n, float* out);

typedef struct sdro_fdecim sdro_fdecim;
sdro_fdecim* sdro_fdecim_new(int log2_decim, int fcpos, int in_kind, int out_kind, int input_bits);
void    sdro_fdecim_free(sdro_fdecim*);
void    sdro_fdecim_reset(sdro_fdecim*);
void    sdro_fdecim_switch(sdro_fdecim*, int log2_decim, int fcpos);   /* next call: another decimateK_x on the same six filters */
int32_t sdro_fdecim_process(sdro_fdecim*, const void* in, int32_t n_elems, void* out);
int32_t sdro_fdecim_group(int log2_decim, int fcpos);      /* elements per loop iteration of the reference function */

#ifdef __cplusplus
}
#endif
#endif
