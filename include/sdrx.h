/* sdrx -- MI355X (gfx950) engine for SDRangel's sdrbase/dsp RX hot path.
 *
 * C ABI of libsdrx.so.  Plain pointers and sizes only.  Every entry point names the reference
 * interface it replaces (paths relative to the lainy/sdrangel tree, v4.0.6).
 *
 * Conventions
 *   - a complex sample is the reference's `Sample` {int16 re; int16 im} packed in 4 bytes
 *     (sdrbase/dsp/dsptypes.h:44-65); buffers of them are "iq" (interleaved I,Q int16).
 *   - every function returns 0 on success or a negative code (SDRX_E*); nothing throws across
 *     the ABI.  sdrx_last_error() gives the text of the calling thread's last failure.
 *   - a handle is single-threaded (caller serialises, like one Decimators member per device
 *     thread in the reference); different handles are independent and may sit on different GPUs.
 *   - `*_dev` variants take device pointers and are asynchronous on the handle's HIP stream;
 *     the host-pointer variants copy in/out and return when the result is in the caller's buffer.
 *   - the library has NO CPU fallback: without a usable HIP device every create call fails with
 *     SDRX_ENODEV.
 */
#ifndef SDRX_H
#define SDRX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SDRX_OK        0
#define SDRX_EINVAL   -1   /* bad argument */
#define SDRX_ENODEV   -2   /* no HIP device / device index out of range */
#define SDRX_EHIP     -3   /* a HIP runtime call failed (text in sdrx_last_error) */
#define SDRX_ENOMEM   -4
#define SDRX_ESTATE   -5   /* call not valid in the handle's current state */

/* fcPos of the device plugins (limesdrinputthread.cpp:103-135): which decimateK_* is called */
#define SDRX_FC_INF 0      /* decimateK_inf */
#define SDRX_FC_SUP 1      /* decimateK_sup */
#define SDRX_FC_CEN 2      /* decimateK_cen */

/* DownChannelizer::FilterStage::Mode (sdrbase/dsp/downchannelizer.h:76-80) */
#define SDRX_MODE_CENTER 0
#define SDRX_MODE_LOWER  1
#define SDRX_MODE_UPPER  2

const char* sdrx_version(void);
const char* sdrx_last_error(void);
int         sdrx_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Decimators<qint32, qint16, 16, InputBits>  (sdrbase/dsp/decimators.h:279-341)
 * One handle == one `m_decimators` member used with ONE (log2, fcpos) -- the reference keeps the
 * six half-band states inside the object (decimators.h:326-340); so does the handle.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_decim sdrx_decim_t;

/* log2_decim 0..6, fcpos SDRX_FC_*, input_bits 8|12|16 (decimation_shifts<16,InputBits>,
 * decimators.h:25-185). */
int sdrx_decim_create(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int input_bits);
/* DecimatorsU<qint32, quint8, 16, 8, Shift> (sdrbase/dsp/decimatorsu.h:175-216; RTL-SDR thread,
 * plugins/samplesource/rtlsdr/rtlsdrthread.h:55 uses Shift = 127): unsigned 8-bit I/Q, value = byte - shift,
 * decimation_shifts<16,8>.  Same cascades, strides and tail drop as Decimators. */
int sdrx_decim_create_u8(sdrx_decim_t** out, int device, int log2_decim, int fcpos, int shift);
int sdrx_decim_process_u8(sdrx_decim_t* h, const uint8_t* iq, int32_t n_uint8, int16_t* out_iq, int32_t* n_out_cplx);
/* d_iq must be 8-byte aligned */
int sdrx_decim_process_dev_u8(sdrx_decim_t* h, const uint8_t* d_iq, int64_t n_uint8, int16_t* d_out_iq, int64_t* n_out_cplx);
int sdrx_decim_destroy(sdrx_decim_t* h);
/* zero filter state == a freshly constructed Decimators object */
int sdrx_decim_reset(sdrx_decim_t* h);

/* Replaces  m_decimators.decimateK_{inf,sup,cen}(&it, buf, len)  (decimators.h:463-3885):
 * `iq`/`n_int16` are the reference's `buf`/`len`; whole groups only, a trailing partial group
 * is dropped and NOT carried (decimators.h:3492); state is carried across calls.
 * out_iq must hold n_int16/2 >> log2 complex samples; *n_out_cplx = how far `it` advanced. */
int sdrx_decim_process(sdrx_decim_t* h, const int16_t* iq, int32_t n_int16,
                       int16_t* out_iq, int32_t* n_out_cplx);

/* Same contract on device-resident buffers, asynchronous on the handle's stream.
 * d_iq must be 16-byte aligned. */
int sdrx_decim_process_dev(sdrx_decim_t* h, const int16_t* d_iq, int64_t n_int16,
                           int16_t* d_out_iq, int64_t* n_out_cplx);
/* MANY device streams, ONE launch.  The reference runs one Decimators object per device thread
 * (plugins/samplesource/limesdrinput/limesdrinputthread.cpp:103-135, filesource/testsource likewise); with many
 * concurrent device sets each of their blocks (32 768 samples for LimeSDR) is far too small to fill a GPU on its own.
 * handles[i] consumes n_elems[i] input elements at d_iq[i] (int16 for sdrx_decim_create handles, bytes for
 * sdrx_decim_create_u8 ones; same whole-group / tail-drop rule per stream, each stream's own carried state) into
 * d_out_iq[i]; n_out_cplx[i] (optional) receives how far that stream's `it` advanced.  All handles must have been created
 * with the same (log2, fcpos, input flavour) on the same device and must be distinct; everything is queued on
 * handles[0]'s stream, which the other handles join (as by sdrx_decim_set_stream) the first time.  More than 64
 * handles are served by consecutive launches of 64. */
int sdrx_decim_process_dev_batch(sdrx_decim_t* const* handles, int32_t n_handles, const void* const* d_iq,
                                 const int64_t* n_elems, int16_t* const* d_out_iq, int64_t* n_out_cplx);
/* Pinned, double-buffered HOST path (SURVEY 8b "Ownership": the async variant + sync).  The device thread's receive
 * buffer IS a slot of a pinned ring the handle owns, so a block travels host -> HBM by DMA while the previous blocks are
 * being decimated and their outputs travel back (limesdrinputthread.cpp:77-135: LMS_RecvStream(buf) -> decimate -> FIFO):
 *     void* in = sdrx_decim_ring_acquire(h);            next free slot (NULL + last_error when the ring is full)
 *     ... fill `in` with up to slot_elems elements ...
 *     sdrx_decim_ring_submit(h, n_elems);               returns at once; same whole-group / tail-drop rule per block
 *     sdrx_decim_ring_retire(h, &out, &n_out_cplx);     oldest submitted block: waits for it; `out` (pinned) stays valid
 *                                                        until that slot is acquired again
 * Blocks are retired in submission order.  `flush_slots` full blocks are coalesced into ONE copy + ONE launch (consecutive
 * blocks of a stream are consecutive samples, and a full slot is a whole number of groups, so the result is identical to
 * separate calls): 1 = every block at once (lowest latency), 16 = a LimeSDR-sized 32 768-sample block rate that is not
 * bound by launch latency.  slot_elems: int16 per slot (bytes for the u8 flavour), a whole number of groups, bytes % 16 == 0. */
int sdrx_decim_ring_create(sdrx_decim_t* h, int32_t slot_elems, int32_t n_slots, int32_t flush_slots);
int sdrx_decim_ring_destroy(sdrx_decim_t* h);
void* sdrx_decim_ring_acquire(sdrx_decim_t* h);
int sdrx_decim_ring_submit(sdrx_decim_t* h, int32_t n_elems);
int sdrx_decim_ring_retire(sdrx_decim_t* h, const int16_t** out_iq, int32_t* n_out_cplx);
int sdrx_decim_sync(sdrx_decim_t* h);
/* run on a caller-owned hipStream_t; NULL = the handle's own (non-blocking) stream.  The HIP default stream has the
 * handle value 0 and is therefore NOT selectable: work queued on it (PyTorch's default stream) is not ordered against the
 * handle's stream -- synchronise, or hand over a real stream object. */
int sdrx_decim_set_stream(sdrx_decim_t* h, void* hip_stream);
/* #int16 consumed per loop iteration of the matching reference function (its `pos +=`) */
int sdrx_decim_group_int16(int log2_decim, int fcpos);
/* One reference Decimators object runs every decimateK_x on the SAME six half-band filters (m_decimator2 .. m_decimator64,
 * decimators.h:326-333): a device thread that changes log2Decim or fcPos at run time continues on whatever each stage saw
 * last.  A handle here is one variant; `sdrx_decim_stages_t` is the object's shared filter set.  On a change of variant:
 *     sdrx_decim_save_stages(old_handle, stages);      stages 1..log2(old) := what old_handle's filters hold now
 *     sdrx_decim_load_stages(new_handle, stages);      new_handle continues from them (its own history is forgotten)
 * and the outputs equal the reference object's, bit for bit (include/sdrx/dsp.hpp does this inside sdrx::Decimators).
 * Cost: a few milliseconds per change (a one-lane walk over 4096 samples on the device); nothing on the steady path. */
typedef struct sdrx_decim_stages sdrx_decim_stages_t;
int sdrx_decim_stages_create(sdrx_decim_stages_t** s, int device);
int sdrx_decim_stages_destroy(sdrx_decim_stages_t* s);
int sdrx_decim_save_stages(sdrx_decim_t* h, sdrx_decim_stages_t* s);
int sdrx_decim_load_stages(sdrx_decim_t* h, const sdrx_decim_stages_t* s);

/* checkpoint of the carried state (the last `sdrx_decim_state_bytes()` bytes of consumed input;
 * the six ring buffers of the reference are a pure function of it) */
int64_t sdrx_decim_state_bytes(const sdrx_decim_t* h);
int sdrx_decim_get_state(sdrx_decim_t* h, void* host_buf);
int sdrx_decim_set_state(sdrx_decim_t* h, const void* host_buf);
/* HIP-event timing of the chain kernel itself, on the stream it is launched on: when enabled every
 * process call brackets its main kernel with two events; get_timing synchronises the stream and
 * returns the summed kernel time and launch count since the last reset. */
int sdrx_decim_set_timing(sdrx_decim_t* h, int enabled);
int sdrx_decim_get_timing(sdrx_decim_t* h, double* total_ms, int64_t* launches, int reset);
/* name + launch geometry of the kernel the last process call launched (for profiling/bench) */
int sdrx_decim_last_launch(const sdrx_decim_t* h, char* kernel_name, int name_cap,
                           int* grid, int* block, int* lds_bytes);

/* ------------------------------------------------------------------------------------------
 * DownChannelizer bank  (sdrbase/dsp/downchannelizer.{h,cpp}) -- N channels fed from ONE device
 * stream.  Replaces N x { ThreadedBasebandSampleSink::feed -> DownChannelizer::feed }
 * (threadedbasebandsamplesink.cpp:114-119, downchannelizer.cpp:50-91): the input is read once and
 * every distinct prefix of the channels' half-band chains is evaluated once.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_chan_bank sdrx_chan_bank_t;

/* Runs DownChannelizer::applyConfiguration's float bisection (downchannelizer.cpp:157-189,
 * 250-287) per channel: req_rate[c]/req_fc[c] are the DSPConfigureChannelizer arguments. */
int sdrx_chan_bank_create(sdrx_chan_bank_t** out, int device, int32_t in_rate, int32_t n_ch,
                          const int32_t* req_rate, const int32_t* req_fc);
int sdrx_chan_bank_destroy(sdrx_chan_bank_t* h);
/* per-channel result of the bisection == what MsgChannelizerNotification reports
 * (downchannelizer.cpp:184-187); modes[] gets n_stages entries (cap 32). */
int sdrx_chan_bank_info(const sdrx_chan_bank_t* h, int32_t ch, int32_t* n_stages, uint8_t* modes,
                        int32_t* out_rate, int32_t* residual_ofs);
/* the bisection alone, no device needed (host logic) */
int sdrx_chan_plan(int32_t in_rate, int32_t req_rate, int32_t req_fc,
                   uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs);
/* DSPConfigureChannelizer for one channel: chain rebuilt with ZERO history
 * (downchannelizer.cpp:167-171 frees and recreates the stages). */
int sdrx_chan_bank_reconfigure(sdrx_chan_bank_t* h, int32_t ch, int32_t req_rate, int32_t req_fc);
/* A new DownChannelizer next to the running ones (a demod plugin added to the device set,
 * sdrbase/device/devicesourceapi.h:47-50 addThreadedSink): starts from zero history with the next feed; the existing
 * channels keep their histories and queued output.  *channel = its index. */
int sdrx_chan_bank_add_channel(sdrx_chan_bank_t* h, int32_t req_rate, int32_t req_fc, int32_t* channel);
/* removeThreadedSink: channel `ch` stops producing and its queued output is dropped; the index stays reserved
 * (it can be revived with sdrx_chan_bank_reconfigure). */
int sdrx_chan_bank_remove_channel(sdrx_chan_bank_t* h, int32_t ch);
/* number of independently planned stage tries the bank currently evaluates per feed (1 after create / reset; a
 * reconfigured or added channel runs in a trie of its own, and tries without a live channel are retired) */
/* checkpoint of the bank's filter state: every stream's history and sample count (the reference's per-stage rings are a
 * pure function of them); queued, unread output is not part of it.  A state fits only a bank with the same channels
 * configured in the same order; set_state checks the shape, drops what is queued and continues the saved timeline. */
int64_t sdrx_chan_bank_state_bytes(const sdrx_chan_bank_t* b);
int sdrx_chan_bank_get_state(sdrx_chan_bank_t* b, void* host_buf);
int sdrx_chan_bank_set_state(sdrx_chan_bank_t* b, const void* host_buf);
int32_t sdrx_chan_bank_group_count(const sdrx_chan_bank_t* h);
int sdrx_chan_bank_reset(sdrx_chan_bank_t* h);

/* Replaces DownChannelizer::feed(begin, end, positiveOnly) for every channel of the bank.  Any
 * n_cplx; decimation phase is carried across calls (no drop).  Outputs accumulate in per-channel
 * device queues until read. */
int sdrx_chan_bank_feed(sdrx_chan_bank_t* h, const int16_t* iq, int64_t n_cplx);
int sdrx_chan_bank_feed_dev(sdrx_chan_bank_t* h, const int16_t* d_iq, int64_t n_cplx);
/* number of complex outputs of channel ch waiting to be read */
int64_t sdrx_chan_bank_available(sdrx_chan_bank_t* h, int32_t ch);
/* == the m_sampleBuffer handed to m_sampleSink->feed (downchannelizer.cpp:87): copies up to cap
 * complex samples of channel ch to host memory and removes them; returns the count (<0: error) */
int64_t sdrx_chan_bank_read(sdrx_chan_bank_t* h, int32_t ch, int16_t* out_iq, int64_t cap);
/* readCommit-style: discard up to n queued samples of channel ch without copying (n < 0: all) */
int64_t sdrx_chan_bank_skip(sdrx_chan_bank_t* h, int32_t ch, int64_t n);
/* device-side view of what the last feed produced for channel ch (valid until the next feed) */
int sdrx_chan_bank_last_dev(sdrx_chan_bank_t* h, int32_t ch, const int16_t** d_out_iq, int64_t* n_cplx);
int sdrx_chan_bank_sync(sdrx_chan_bank_t* h);
int sdrx_chan_bank_set_stream(sdrx_chan_bank_t* h, void* hip_stream);
/* the hipStream_t the bank's kernels run on (its own stream unless set_stream gave it another) */
int sdrx_chan_bank_get_stream(sdrx_chan_bank_t* h, void** hip_stream);
/* as sdrx_decim_set_timing: brackets each feed's tree_kernel launches (all passes) */
int sdrx_chan_bank_set_timing(sdrx_chan_bank_t* h, int enabled);
int sdrx_chan_bank_get_timing(sdrx_chan_bank_t* h, double* total_ms, int64_t* feeds, int reset);
int sdrx_chan_bank_last_launch(const sdrx_chan_bank_t* h, char* kernel_name, int name_cap,
                               int* grid, int* block, int* lds_bytes);

/* ------------------------------------------------------------------------------------------
 * Channel back-end bank: what every channelrx demod does with the DownChannelizer output before its
 * audio-rate tail (plugins/channelrx/demodnfm/nfmdemod.cpp:150-163, demodssb/ssbdemod.cpp:158-172):
 *     Complex c(re, im); c *= m_nco.nextIQ();                         NCO (sdrbase/dsp/nco.cpp:30-64)
 *     if (m_interpolator.decimate(&dist, c, &ci)) { ... dist += step } Interpolator (interpolator.h:23-36)
 *     n = filter->runSSB(ci, &sideband, usb) | runFilt(...)            fftfilt (fftfilt.cpp:261-325), g_fft
 *     demod = m_phaseDiscri.phaseDiscriminatorDelta(...)               phasediscri.h:50-78
 * One handle holds N channels; every feed produces that feed's outputs (like the demod's feed()
 * running to completion), read them before the next feed.  float32, <= 1 ulp of the strict-IEEE
 * scalar reference build (SURVEY.md finding 6).
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_backend sdrx_backend_t;
typedef struct sdrx_backend_cfg {
    int32_t in_rate;         /* channelizer output rate: m_nco.setFreq(nco_freq, in_rate) */
    int32_t nco_freq;        /* the demods pass -frequencyOffset of MsgChannelizerNotification */
    int32_t out_rate;        /* audio / demod rate; distance step = (Real) in_rate / (Real) out_rate; <= in_rate */
    float   interp_cutoff;   /* m_interpolator.create(16, in_rate, interp_cutoff, taps_per_phase) */
    float   taps_per_phase;  /* 4.5 (default, NFM) or 2.0 (SSB) */
    int32_t filt_mode;       /* 0 none, 1 runFilt, 2 runSSB usb, 3 runSSB lsb, 4 runDSB  (getDC = true),
                              * 5 runAsym usb, 6 runAsym lsb: fftfilt(f2, 2048) + create_asym_filter(fopp = f1, fin = f2) (atvdemod.cpp:262,647) */
    float   f1, f2;          /* modes 1-3: fftfilt(f1, f2, 1024); mode 4: fftfilt(f2, 2048) (DSBFilter, ssbdemod.cpp:92); normalised to the OUTPUT rate */
    int32_t discri;          /* 0 none, 1 phaseDiscriminatorDelta (NFM; bit-identical to the strict-IEEE reference),
                              * 2 phaseDiscriminator (UDPSrc): std::arg = atan2f.  The device evaluates a double atan2 rounded once
                              * to float: <= 3 ulp from glibc's atan2f (2 measured, tests/test_backend_gpu.py), i.e. outside the
                              * 1 ulp of the other float stages -- parity with a given reference binary depends on that box's libm. */
    float   fm_scaling;      /* setFMScaling */
} sdrx_backend_cfg;
int sdrx_backend_create(sdrx_backend_t** out, int device, int32_t n_ch, const sdrx_backend_cfg* cfg);
int sdrx_backend_destroy(sdrx_backend_t* h);
/* iq[c] / n_per_ch[c]: channel c's new samples (what DownChannelizer handed to m_sampleSink->feed) */
int sdrx_backend_feed(sdrx_backend_t* h, const int16_t* const* iq, const int64_t* n_per_ch);
int sdrx_backend_feed_dev(sdrx_backend_t* h, const int16_t* const* d_iq, const int64_t* n_per_ch);
/* cfg 4 hand-over without a host round trip: channel c takes what the bank's last feed produced for its channel c
 * (sdrx_chan_bank_last_dev), ordered on the device -- the back-end's readers wait for the bank's stream, and the bank's
 * stream waits until they have consumed the samples before anything queued on it later (its next feed) may run.
 * The distance schedule, which needs the counts only, overlaps the bank's kernels. */
int sdrx_backend_feed_bank(sdrx_backend_t* h, sdrx_chan_bank_t* bank);
/* outputs of the last feed for channel ch: complex (re,im pairs) unless a discriminator is on;
 * returns the number of FLOATS written (<0: error) */
int64_t sdrx_backend_read(sdrx_backend_t* h, int32_t ch, float* out, int64_t cap_floats);
/* design products, for inspection: polyphase taps [16][ntaps], filter spectrum (2048 complex slots; 1024 used
 * unless filt_mode 4), NCO increment */
int sdrx_backend_get_design(sdrx_backend_t* h, int32_t ch, int32_t* ntaps_per_phase, float* taps, int32_t taps_cap,
                            float* filter_iq, int32_t* nco_inc);
int sdrx_backend_sync(sdrx_backend_t* h);

/* ------------------------------------------------------------------------------------------
 * Audio-rate tail of the NFM and SSB demodulators -- what follows the resampler / fftfilt in
 * NFMDemod::feed (plugins/channelrx/demodnfm/nfmdemod.cpp:150-300; m_deltaSquelch, m_ctcssOn, m_audioMute off) and
 * SSBDemod::feed (plugins/channelrx/demodssb/ssbdemod.cpp:181-250; mono): discriminator + power squelch + gate delay
 * line + 301-tap Bandpass for NFM, MagAGC (sdrbase/dsp/agc.cpp:96-175) + delay line + step value for SSB, down to the
 * qint16 the demod writes to both channels of m_audioBuffer.  Input per channel: the complex float stream the demod body
 * sees (sdrx_backend_* with discri = 0: resampler output for NFM, fftfilt sideband for SSB).  One output per input.
 * Serial state machines: one lane per channel, N channels per handle.  m_prevArg of the reference's PhaseDiscriminators
 * is uninitialised (phasediscri.h:139); it starts at 0 here.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_audiotail sdrx_audiotail_t;
typedef struct sdrx_audiotail_cfg {
    int32_t kind;                 /* 0 NFM, 1 SSB */
    int32_t audio_rate;           /* m_audioSampleRate (48000) */
    float   volume;               /* m_settings.m_volume (NFM) / m_volume (SSB) */
    /* NFM */
    float   fm_scaling;           /* m_phaseDiscri.setFMScaling(): (float) audioRate / (2 * fmDeviation) */
    float   squelch_level;        /* m_squelchLevel: linear power */
    int32_t squelch_gate;         /* m_squelchGate, samples */
    float   af_bandwidth;         /* m_bandpass.create(301, rate, 300.0, af_bandwidth) */
    /* SSB: MagAGC(12000, agcTarget, 1e-2) after resize(n, n / 2, agcTarget), setStepDownDelay(n) (ssbdemod.cpp:411-414) */
    int32_t agc_active;           /* settings.m_agc; off: agcVal = 10.0 */
    int32_t agc_nb_samples;       /* (audioRate / 1000) * (1 << agcTimeLog2) */
    int32_t agc_threshold_enable; /* setThresholdEnable */
    int32_t agc_gate;             /* setGate, samples */
    int32_t agc_clamping;         /* setClamping; clampMax = SDR_RX_SCALED / 100 */
    double  agc_threshold;        /* setThreshold: powerFromdB(dB) * 32768^2 */
} sdrx_audiotail_cfg;
int sdrx_audiotail_create(sdrx_audiotail_t** h, int device, int32_t n_ch, const sdrx_audiotail_cfg* cfg);
int sdrx_audiotail_destroy(sdrx_audiotail_t* h);
int sdrx_audiotail_reset(sdrx_audiotail_t* h);
/* in[c]: n[c] complex floats (re, im); audio[c]: n[c] qint16 (the value written to .l and .r) */
int sdrx_audiotail_feed(sdrx_audiotail_t* h, const float* const* in, const int64_t* n, int16_t* const* audio);
int sdrx_audiotail_feed_dev(sdrx_audiotail_t* h, const float* const* d_in, const int64_t* n, int16_t* const* d_audio);
int sdrx_audiotail_sync(sdrx_audiotail_t* h);

/* ------------------------------------------------------------------------------------------
 * The 24-bit sample build of the integer half-band path (the reference compiled with SDR_RX_SAMPLE_24BIT: dsptypes.h:24-34
 * FixReal = qint32 and an 8-byte Sample; decimators.h:326-333, downchannelizer.h:78-81 IntHalfbandFilterEO<qint64,qint64,N>;
 * decimation_shifts<24,InputBits>, decimators.h:62-185).  Samples in and out of these calls are {int32 re, int32 im}.
 *   sdrx_decim24_*      Decimators<qint32, qint16, 24, {8,12,16}>::decimate{1..64}_{cen,inf,sup}: same call contract as
 *                       sdrx_decim_process (whole groups, dropped tail, carried state); out_iq holds 2 x int32 per sample
 *   sdrx_chan24_bank_*  N DownChannelizers on a 24-bit stream: any feed length, carried phase, final `/= (1 << n)`;
 *                       sdrx_chan24_bank_read returns what the LAST feed produced for the channel
 * Exact (incl. the build's 32-bit wrap of the centre tap, inthalfbandfiltereo.h:818-827), plain 64-bit arithmetic, not tuned.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_decim24 sdrx_decim24_t;
int sdrx_decim24_create(sdrx_decim24_t** h, int device, int log2_decim, int fcpos, int input_bits);
int sdrx_decim24_destroy(sdrx_decim24_t* h);
int sdrx_decim24_reset(sdrx_decim24_t* h);
int sdrx_decim24_process(sdrx_decim24_t* h, const int16_t* iq, int32_t n_int16, int32_t* out_iq, int32_t* n_out_cplx);
/* device pointers, asynchronous on the handle's stream (sdrx_decim24_sync waits): d_iq = n_cplx int16 pairs, d_out = room for
 * (n_cplx >> log2) + 1 samples of 8 bytes; *n_out_cplx = samples this call produced (the phase carries) */
int sdrx_decim24_process_dev(sdrx_decim24_t* h, const void* d_iq, int64_t n_cplx, void* d_out, int64_t* n_out_cplx);
int sdrx_decim24_sync(sdrx_decim24_t* h);
typedef struct sdrx_chan24_bank sdrx_chan24_bank_t;
int sdrx_chan24_bank_create(sdrx_chan24_bank_t** h, int device, int32_t in_rate, int32_t n_ch, const int32_t* req_rate, const int32_t* req_fc);
int sdrx_chan24_bank_destroy(sdrx_chan24_bank_t* h);
int sdrx_chan24_bank_reset(sdrx_chan24_bank_t* h);
int sdrx_chan24_bank_info(const sdrx_chan24_bank_t* h, int32_t ch, int32_t* n_stages, uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs);
int sdrx_chan24_bank_feed(sdrx_chan24_bank_t* h, const int32_t* iq, int64_t n_cplx);
int64_t sdrx_chan24_bank_read(sdrx_chan24_bank_t* h, int32_t ch, int32_t* out_iq, int64_t cap);
/* device pointers: feed n_cplx {int32,int32} from HBM, then look at each channel's output where it lies (valid until the
 * next feed); asynchronous on the bank's stream, sdrx_chan24_bank_sync waits */
int sdrx_chan24_bank_feed_dev(sdrx_chan24_bank_t* h, const void* d_iq, int64_t n_cplx);
int sdrx_chan24_bank_out_dev(sdrx_chan24_bank_t* h, int32_t ch, const void** d_out, int64_t* n_cplx);
int sdrx_chan24_bank_sync(sdrx_chan24_bank_t* h);

/* ------------------------------------------------------------------------------------------
 * IIRFilter<float, Order> (sdrbase/dsp/iirfilter.h; FilterMbe's low/high-pass pair, filtermbe.h:76-77): N recursive
 * filters, one per channel, state carried across feeds.  `a` / `b` are the constructor's arguments in the reference's
 * meaning: order 2 = the specialisation (y = b0 s + b1 x0 + b2 x1 + a1 y0 + a2 y1); other orders = the generic template,
 * including its swapped coefficient copy (iirfilter.h:78-81).  Serial along time: one lane per channel.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_iir sdrx_iir_t;
typedef struct sdrx_iir_cfg { int32_t order; float a[9]; float b[9]; } sdrx_iir_cfg;   /* order 2..8, order + 1 coefficients each */
int sdrx_iir_create(sdrx_iir_t** h, int device, int32_t n_ch, const sdrx_iir_cfg* cfg);
int sdrx_iir_destroy(sdrx_iir_t* h);
int sdrx_iir_reset(sdrx_iir_t* h);
int sdrx_iir_feed(sdrx_iir_t* h, const float* const* in, const int64_t* n, float* const* out);

/* ------------------------------------------------------------------------------------------
 * Lowpass<Real> / Bandpass<Real> (sdrbase/dsp/lowpass.h:11-105, bandpass.h:11-128): the symmetric-folded real
 * FIRs of the demods' audio tail (NFM: m_lowpass.create(301, rate, 250.0), m_bandpass.create(301, rate, 300.0, bw),
 * nfmdemod.cpp:88,428-429; filter() per audio sample :239,279), N channels per handle, state carried across feeds.
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_firbank sdrx_firbank_t;
typedef struct sdrx_fir_cfg {
    int32_t kind;            /* 0: Lowpass::create(ntaps, sample_rate, f1); 1: Bandpass::create(ntaps, sample_rate, f1, f2) */
    int32_t ntaps;           /* made odd like the reference does */
    float   sample_rate, f1, f2;
} sdrx_fir_cfg;
int sdrx_firbank_create(sdrx_firbank_t** out, int device, int32_t n_ch, const sdrx_fir_cfg* cfg);
int sdrx_firbank_destroy(sdrx_firbank_t* h);
/* == filter(sample) for every sample of in[c]; out[c] gets n_per_ch[c] floats */
int sdrx_firbank_feed(sdrx_firbank_t* h, const float* const* in, const int64_t* n_per_ch, float* const* out);
/* the ntaps/2 + 1 folded taps (m_taps); returns their count */
int sdrx_firbank_get_taps(const sdrx_firbank_t* h, int32_t ch, float* taps, int32_t cap);

/* ------------------------------------------------------------------------------------------
 * SampleSinkFifo (sdrbase/dsp/samplesinkfifo.{h,cpp}) -- host ring of `Sample`, same
 * write / readBegin / readCommit contract, minus the Qt signal (a callback instead of dataReady()).
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_fifo sdrx_fifo_t;
typedef void (*sdrx_fifo_data_ready_cb)(void* user);
int      sdrx_fifo_create(sdrx_fifo_t** out, uint32_t size_samples);          /* SampleSinkFifo(int) + setSize */
int      sdrx_fifo_destroy(sdrx_fifo_t* f);
int      sdrx_fifo_set_size(sdrx_fifo_t* f, uint32_t size_samples);           /* setSize: also empties */
uint32_t sdrx_fifo_size(sdrx_fifo_t* f);
uint32_t sdrx_fifo_fill(sdrx_fifo_t* f);
void     sdrx_fifo_on_data_ready(sdrx_fifo_t* f, sdrx_fifo_data_ready_cb cb, void* user);
/* write(const quint8* data, uint count) (samplesinkfifo.cpp:70-111): count in BYTES, returns samples written */
uint32_t sdrx_fifo_write_bytes(sdrx_fifo_t* f, const uint8_t* data, uint32_t count_bytes);
/* write(begin, end) (samplesinkfifo.cpp:113-153): count in samples */
uint32_t sdrx_fifo_write(sdrx_fifo_t* f, const int16_t* iq, uint32_t count_samples);
/* read(begin, end) (samplesinkfifo.cpp:155-191) */
uint32_t sdrx_fifo_read(sdrx_fifo_t* f, int16_t* iq, uint32_t count_samples);
/* readBegin / readCommit (samplesinkfifo.cpp:193-231): two spans as offsets into the ring */
uint32_t sdrx_fifo_read_begin(sdrx_fifo_t* f, uint32_t count, const int16_t** part1, uint32_t* n1,
                              const int16_t** part2, uint32_t* n2);
uint32_t sdrx_fifo_read_commit(sdrx_fifo_t* f, uint32_t count);
/* samples dropped by overflowing writes since creation (the reference only logs them) */
uint64_t sdrx_fifo_dropped(sdrx_fifo_t* f);

/* ------------------------------------------------------------------------------------------
 * .sdriq record files (sdrbase/dsp/filerecord.cpp:129-148; read by the FileSource plugin,
 * plugins/samplesource/filesource/filesourceinput.cpp / filesourcethread.cpp:170-229).
 * Layout: qint32 sampleRate | quint64 centerFrequency | time_t startTimeStamp | quint32 sampleSize, written
 * field by field = 24 bytes, then raw `Sample`s.  readHeader() treats any sampleSize other than 16/24 as 16.
 * Reference quirk kept visible: on end-of-file FileSourceThread::tick() rewinds to sizeof(FileRecord::Header),
 * which is 32 with struct padding, i.e. loop playback skips the first two samples of the file.
 * ------------------------------------------------------------------------------------------ */
#define SDRX_SDRIQ_HEADER_BYTES 24
#define SDRX_SDRIQ_LOOP_OFFSET  32
typedef struct sdrx_sdriq_header {
    int32_t  sample_rate;
    uint64_t center_frequency;
    int64_t  start_timestamp;
    uint32_t sample_size;       /* 16 or 24 after parsing */
} sdrx_sdriq_header;
int sdrx_sdriq_parse_header(const uint8_t* bytes, uint64_t n_bytes, sdrx_sdriq_header* out);
int sdrx_sdriq_write_header(uint8_t* bytes24, const sdrx_sdriq_header* hdr);

/* ---- float half-band decimators ----
 * DecimatorsFI (sdrbase/dsp/decimatorsfi.h:29-57: float I/Q in, int16 Sample out; the AirspyHF thread's member,
 * plugins/samplesource/airspyhf/airspyhfthread.h), DecimatorsFF (decimatorsff.h: float in, float FSample out) and
 * DecimatorsIF<qint16,InputBits> (decimatorsif.h:52-79: int16 in, float out), all over IntHalfbandFilterEOF<64>
 * (inthalfbandfiltereof.h).  One handle = one decimateK_{inf,sup,cen} method of one object: (log2_decim, fcpos).
 *   in_kind  0 float I/Q            1 int16 I/Q, scaled by decimation_scale<input_bits> (8|12|16) at the output
 *   out_kind 0 int16 Sample = (int16)(v * SDR_RX_SCALED), truncation (float input only)      1 float re, im
 * n_elems = the reference's nbIAndQ (floats or int16s); whole groups only, tail dropped; filter state carried.
 * Results are bit-identical to the reference built without -ffast-math (same operation order, no FMA). */
typedef struct sdrx_fdecim sdrx_fdecim_t;
#define SDRX_FD_IN_F32  0
#define SDRX_FD_IN_I16  1
#define SDRX_FD_OUT_I16 0
#define SDRX_FD_OUT_F32 1
int sdrx_fdecim_create(sdrx_fdecim_t** out, int device, int log2_decim, int fcpos, int in_kind, int out_kind, int input_bits);
int sdrx_fdecim_destroy(sdrx_fdecim_t* h);
int sdrx_fdecim_reset(sdrx_fdecim_t* h);
/* host pointers; blocking.  *n_out_cplx = advance of the reference's output iterator */
int sdrx_fdecim_process(sdrx_fdecim_t* h, const void* in, int32_t n_elems, void* out, int32_t* n_out_cplx);
/* device pointers (d_in 16-byte, d_out 8-byte aligned); asynchronous on the handle's stream */
int sdrx_fdecim_process_dev(sdrx_fdecim_t* h, const void* d_in, int64_t n_elems, void* d_out, int64_t* n_out_cplx);
int sdrx_fdecim_sync(sdrx_fdecim_t* h);
int sdrx_fdecim_set_stream(sdrx_fdecim_t* h, void* hip_stream);
/* input elements per loop iteration of the reference method (its `pos +=` stride) */
int32_t sdrx_fdecim_group(int log2_decim, int fcpos);
/* checkpoint of the carried state (the cascade's filter rings) */
int64_t sdrx_fdecim_state_bytes(const sdrx_fdecim_t* h);
int sdrx_fdecim_get_state(sdrx_fdecim_t* h, void* host_buf);
int sdrx_fdecim_set_state(sdrx_fdecim_t* h, const void* host_buf);
/* the six IntHalfbandFilterEOF members that all decimateK_x of one DecimatorsFI / FF / IF object share (cascade stage s is
 * member s in every variant): same protocol as sdrx_decim_save_stages / _load_stages.  The float handles carry their filters'
 * rings explicitly, so both calls are plain device copies. */
typedef struct sdrx_fdecim_stages sdrx_fdecim_stages_t;
int sdrx_fdecim_stages_create(sdrx_fdecim_stages_t** s, int device);
int sdrx_fdecim_stages_destroy(sdrx_fdecim_stages_t* s);
int sdrx_fdecim_save_stages(sdrx_fdecim_t* h, sdrx_fdecim_stages_t* s);
int sdrx_fdecim_load_stages(sdrx_fdecim_t* h, const sdrx_fdecim_stages_t* s);
int sdrx_fdecim_set_timing(sdrx_fdecim_t* h, int enabled);
int sdrx_fdecim_get_timing(sdrx_fdecim_t* h, double* total_ms, int64_t* launches, int reset);
int sdrx_fdecim_last_launch(const sdrx_fdecim_t* h, char* kernel_name, int name_cap, int* grid, int* block, int* lds_bytes);

/* ---- DC offset correction of the device stream ----
 * What DSPDeviceSourceEngine::work does to every FIFO span before the sinks see it when m_dcOffsetCorrection is set
 * (dspdevicesourceengine.cpp:339-343,375-379 -> iqCorrections(begin, end, false), :175-181,255-259):
 * re -= (int32) m_iBeta, im -= (int32) m_qBeta with MovingAverageUtil<int32_t,int64_t,1024> averages (total of the last
 * 1024 samples / 1024, truncating).  State (the last 1023 samples) is carried across calls; reset == fresh engine.
 * The I/Q imbalance branch (:183-253: float/double averages, a division and a sqrt per sample, serial) is not offered. */
typedef struct sdrx_dccorr sdrx_dccorr_t;
int sdrx_dccorr_create(sdrx_dccorr_t** out, int device);
int sdrx_dccorr_destroy(sdrx_dccorr_t* h);
int sdrx_dccorr_reset(sdrx_dccorr_t* h);
/* in place on a host span, like the reference (blocking) */
int sdrx_dccorr_process(sdrx_dccorr_t* h, int16_t* iq, int64_t n_cplx);
/* device buffers, asynchronous on the handle's stream; d_out_iq must not alias d_iq */
int sdrx_dccorr_process_dev(sdrx_dccorr_t* h, const int16_t* d_iq, int16_t* d_out_iq, int64_t n_cplx);

/* I/Q imbalance correction of the device stream: DSPDeviceSourceEngine::iqCorrections(begin, end, imbalanceCorrection = true)
 * (dspdevicesourceengine.cpp:175-181, 217-253, float flavour: IMBALANCE_INT is not defined), i.e. DC removal + phase and
 * amplitude imbalance estimated by 128-deep float/double moving averages, in the reference's statement order.  The
 * recurrence is serial per stream, so ONE handle serves `n_streams` device streams side by side (one lane each).
 * Buffers are rewritten in place like the reference rewrites the FIFO span.  State carries across calls; reset = freshly
 * constructed engine members. */
typedef struct sdrx_iqimb sdrx_iqimb_t;
int sdrx_iqimb_create(sdrx_iqimb_t** h, int device, int32_t n_streams);
int sdrx_iqimb_destroy(sdrx_iqimb_t* h);
int sdrx_iqimb_reset(sdrx_iqimb_t* h);
int sdrx_iqimb_process(sdrx_iqimb_t* h, int16_t* const* iq, const int64_t* n_cplx);
int sdrx_iqimb_process_dev(sdrx_iqimb_t* h, const int16_t* const* d_iq, int16_t* const* d_out_iq, const int64_t* n_cplx);
int sdrx_iqimb_sync(sdrx_iqimb_t* h);
int sdrx_iqimb_set_stream(sdrx_iqimb_t* h, void* hip_stream);
int sdrx_dccorr_sync(sdrx_dccorr_t* h);
int sdrx_dccorr_set_stream(sdrx_dccorr_t* h, void* hip_stream);

/* ---- diagnostics ----
 * SURVEY 8(d) quotes the HBM roofline twice: the datasheet's 8 TB/s and what a read-only streaming kernel
 * (sum of int32 over n_bytes, best of reps launches) reaches on this box.  Not part of the sample path. */
int sdrx_measure_hbm_read(int device, uint64_t n_bytes, int32_t reps, double* gb_per_s);

/* ------------------------------------------------------------------------------------------
 * Fan-out of one staged source stream to several GPUs by peer copies (xGMI on an 8 x MI355X node): SURVEY 8e's optional
 * staging path -- "one stream per GPU, xGMI only for fan-out, no collective on the per-sample path".  A stream uploaded (or
 * decimated) once on `src_device` is copied point-to-point into one buffer per destination GPU, each on its own stream;
 * the consumers (e.g. one channelizer bank per GPU over the same 61.44 MS/s stream) read their local copy.
 *   send      asynchronous; the copies start when what `producer_stream` (on src_device; NULL = everything queued on the
 *             legacy stream) holds so far is done
 *   buffer    destination i's device pointer (on dst_devices[i]); valid contents after wait / stream_wait
 *   wait      host waits for destination i;  stream_wait: a consumer stream on that GPU waits instead (device-ordered)
 * ------------------------------------------------------------------------------------------ */
typedef struct sdrx_fanout sdrx_fanout_t;
int sdrx_fanout_create(sdrx_fanout_t** f, int src_device, int32_t n_dst, const int32_t* dst_devices, int64_t max_bytes);
int sdrx_fanout_destroy(sdrx_fanout_t* f);
int sdrx_fanout_send(sdrx_fanout_t* f, const void* d_src, int64_t bytes, void* producer_stream);
void* sdrx_fanout_buffer(sdrx_fanout_t* f, int32_t i);
int sdrx_fanout_wait(sdrx_fanout_t* f, int32_t i);
int sdrx_fanout_stream_wait(sdrx_fanout_t* f, int32_t i, void* consumer_stream);

#ifdef __cplusplus
}
#endif
#endif /* SDRX_H */
