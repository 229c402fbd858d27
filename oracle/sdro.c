/* TEST INFRASTRUCTURE -- see sdro.h.  Integer half-band path of the oracle.
 *
 * Restated from the *behaviour* of the reference (closed forms of SURVEY.md Appendix A), not
 * from its ring-buffer code: each stage keeps a plain 64-deep delay line of its (rotated)
 * inputs and evaluates
 *
 *     M = index of the newest (odd) input,  N = order, P = N/4, S = hbShift = 12
 *     acc = sum_{i<P} c[i] * ( x[M-2i] + x[M-(N-2)+2i] )  +  ( x[M-(N/2-1)] << (S-1) )
 *     y   = acc >> (S-1)          (int32 wrap-around arithmetic, arithmetic shift)
 *
 * which is what IntHalfbandFilterEO::storeSample32/advancePointer/doFIR
 * (inthalfbandfiltereo.h:769-790, 832-870) produce for every second input.
 */
#include "sdro.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* hbfiltertraits.cpp:136-154 / :85-99 -- (int32)(c * 4096), truncation toward zero */
static const int32_t HB64[16] = { -1, 2, -5, 8, -12, 17, -25, 35, -47, 64, -86, 117, -164, 244, -424, 1300 };
static const int32_t HB48[12] = { -4, 7, -12, 19, -31, 48, -71, 103, -152, 236, -419, 1299 };
#define HB_SHIFT 12

typedef struct {
    int order;            /* 64 (Decimators) or 48 (DownChannelizer) */
    const int32_t* c;
    int mode;             /* 0 centre, 1 lower/inf (x j^(n+1)), 2 upper/sup (x (-j)^(n+1)) */
    int narrow;           /* 1: Sample (int16) flavour -- int16 negation and int16 store */
    int wide;             /* 1: SDR_RX_SAMPLE_24BIT build: IntHalfbandFilterEO<qint64,qint64,N> -- int64 accumulators, int32 samples */
    uint32_t n;           /* inputs seen since construction (rotation phase = n & 3, parity = n & 1) */
    int32_t re[64], im[64];
} hb_stage;

static void hb_init(hb_stage* s, int order, int mode, int narrow)
{
    memset(s, 0, sizeof *s);            /* wide = 0: callers of the 24-bit flavour set it afterwards */
    s->order = order;
    s->c = order == 64 ? HB64 : HB48;
    s->mode = mode;
    s->narrow = narrow;
}

static inline int32_t neg(const hb_stage* s, int32_t v)
{
    /* (FixReal) -sample->imag()  (inthalfbandfiltereo.h:164): negate in int, cast to int16 */
    return s->narrow ? (int32_t)(int16_t)(uint16_t)(0u - (uint32_t)v) : (int32_t)(0u - (uint32_t)v);
}

/* push one input; returns 1 and overwrites *re,*im when an output is produced */
static inline int hb_push(hb_stage* s, int32_t* re, int32_t* im)
{
    int32_t xr = *re, xi = *im;
    const uint32_t ph = s->n & 3u;
    if (s->mode == 1) {          /* j^(n+1): (-y,x) (-x,-y) (y,-x) (x,y)   (:626-658, 158-206) */
        switch (ph) {
        case 0: { int32_t t = xr; xr = neg(s, xi); xi = t; break; }
        case 1: xr = neg(s, xr); xi = neg(s, xi); break;
        case 2: { int32_t t = xr; xr = xi; xi = neg(s, t); break; }
        default: break;
        }
    } else if (s->mode == 2) {   /* (-j)^(n+1): (y,-x) (-x,-y) (-y,x) (x,y)  (:660-692, 357-405) */
        switch (ph) {
        case 0: { int32_t t = xr; xr = xi; xi = neg(s, t); break; }
        case 1: xr = neg(s, xr); xi = neg(s, xi); break;
        case 2: { int32_t t = xr; xr = neg(s, xi); xi = t; break; }
        default: break;
        }
    }
    const uint32_t M = s->n & 63u;
    s->re[M] = xr; s->im[M] = xi;
    const int odd = (int)(s->n & 1u);
    s->n++;
    if (!odd) return 0;

    const int N = s->order, P = N / 4;
    if (s->wide) {
        /* EOStorageType = AccuType = qint64 (decimators.h:326-333, downchannelizer.h:78-81): the pair sums and products are
         * int64; the centre tap is `((int32_t) x) << 11` -- an INT shift, i.e. it wraps at 32 bits -- added to the int64 sum
         * (inthalfbandfiltereo.h:818-827, 858-867); the result `acc >> 11` is narrowed to the int32 it is stored in. */
        int64_t wr = 0, wi = 0;
        for (int i = 0; i < P; i++) {
            const uint32_t a = (M - 2u * (uint32_t)i) & 63u;
            const uint32_t b = (M - (uint32_t)(N - 2) + 2u * (uint32_t)i) & 63u;
            wr += ((int64_t)s->re[a] + (int64_t)s->re[b]) * (int64_t)s->c[i];
            wi += ((int64_t)s->im[a] + (int64_t)s->im[b]) * (int64_t)s->c[i];
        }
        const uint32_t mc = (M - (uint32_t)(N / 2 - 1)) & 63u;
        wr += (int64_t)(int32_t)((uint32_t)s->re[mc] << (HB_SHIFT - 1));
        wi += (int64_t)(int32_t)((uint32_t)s->im[mc] << (HB_SHIFT - 1));
        *re = (int32_t)(uint32_t)(uint64_t)(wr >> (HB_SHIFT - 1));
        *im = (int32_t)(uint32_t)(uint64_t)(wi >> (HB_SHIFT - 1));
        return 1;
    }
    uint32_t ar = 0, ai = 0;
    for (int i = 0; i < P; i++) {
        const uint32_t a = (M - 2u * (uint32_t)i) & 63u;
        const uint32_t b = (M - (uint32_t)(N - 2) + 2u * (uint32_t)i) & 63u;
        ar += ((uint32_t)s->re[a] + (uint32_t)s->re[b]) * (uint32_t)s->c[i];
        ai += ((uint32_t)s->im[a] + (uint32_t)s->im[b]) * (uint32_t)s->c[i];
    }
    const uint32_t m = (M - (uint32_t)(N / 2 - 1)) & 63u;
    ar += (uint32_t)s->re[m] << (HB_SHIFT - 1);
    ai += (uint32_t)s->im[m] << (HB_SHIFT - 1);
    int32_t yr = (int32_t)ar >> (HB_SHIFT - 1);
    int32_t yi = (int32_t)ai >> (HB_SHIFT - 1);
    if (s->narrow) { yr = (int16_t)yr; yi = (int16_t)yi; }   /* Sample::setReal(FixReal) (:828-829) */
    *re = yr; *im = yi;
    return 1;
}

/* ------------------------------------------------------------------ Decimators */
struct sdro_decim {
    int log2, fcpos, bits, ushift;
    int pre, post, group;
    hb_stage st[6];
};

/* decimation_shifts<16,InputBits> (decimators.h:25-185) */
static void shifts(int bits, int log2, int* pre, int* post)
{
    static const int pre12[7]  = { 4, 3, 2, 1, 0, 0, 0 }, post12[7] = { 0, 0, 0, 0, 0, 1, 2 };
    static const int pre8[7]   = { 8, 7, 6, 5, 4, 3, 2 };
    if (bits == 12) { *pre = pre12[log2]; *post = post12[log2]; }
    else if (bits == 8) { *pre = pre8[log2]; *post = 0; }
    else { *pre = 0; *post = log2; }                       /* <16,16> */
}

int32_t sdro_decim_group_int16(int log2, int fcpos)
{
    /* the `pos +=` strides of decimateK_{inf,sup,cen} (decimators.h:348 ... 3492) */
    if (log2 == 0) return 2;
    if (log2 <= 2) return 4 << log2;                        /* 8, 16 for every mode */
    return fcpos == SDRO_FC_CEN ? (2 << log2) : (4 << log2);
}

void sdro_decim_reset(sdro_decim* d)
{
    const int L = d->log2;
    for (int s = 0; s < L; s++) {
        int mode = 0;
        if (d->fcpos != SDRO_FC_CEN) {
            /* inf: Inf,Sup,...,Sup,Cen ; sup: Sup,Inf,...,Inf,Cen ; L=1: single ; L=2: pair
             * (call pattern of decimators.h:463-2584) */
            const int first = d->fcpos == SDRO_FC_INF ? 1 : 2, other = 3 - first;
            if (s == 0) mode = first;
            else if (L >= 3 && s == L - 1) mode = 0;
            else mode = other;
        }
        hb_init(&d->st[s], 64, mode, 0);
    }
}

sdro_decim* sdro_decim_new(int log2, int fcpos, int bits)
{
    if (log2 < 0 || log2 > 6 || fcpos < 0 || fcpos > 2 || (bits != 8 && bits != 12 && bits != 16)) return 0;
    sdro_decim* d = (sdro_decim*)calloc(1, sizeof *d);
    d->log2 = log2; d->fcpos = fcpos; d->bits = bits;
    shifts(bits, log2, &d->pre, &d->post);
    d->group = sdro_decim_group_int16(log2, fcpos);
    sdro_decim_reset(d);
    return d;
}

void sdro_decim_free(sdro_decim* d) { free(d); }

/* One Decimators object serves every decimateK_x with the SAME six filters (m_decimator2 .. m_decimator64,
 * decimators.h:326-333): stage s of any cascade is member s.  Calling another variant on the object therefore starts from
 * whatever each of its stages saw last (ring contents stay; the rotation pattern of myDecimateInf/Sup is positional inside a
 * call, so it restarts).  Stages beyond the old cascade keep what they held before. */
void sdro_decim_switch(sdro_decim* d, int log2, int fcpos)
{
    for (int s = 0; s < 6; s++) {
        hb_stage* st = &d->st[s];
        if (!st->order) { hb_init(st, 64, 0, 0); continue; }
        int32_t re[64], im[64];
        for (int k = 0; k < 64; k++) { re[k] = st->re[(st->n + (uint32_t)k) & 63u]; im[k] = st->im[(st->n + (uint32_t)k) & 63u]; }
        memcpy(st->re, re, sizeof re); memcpy(st->im, im, sizeof im);
        st->n = 0;                                  /* oldest first: the next store goes to slot 0, phase 0 */
    }
    d->log2 = log2; d->fcpos = fcpos;
    shifts(d->bits, log2, &d->pre, &d->post);
    d->group = sdro_decim_group_int16(log2, fcpos);
    for (int s = 0; s < log2; s++) {
        int mode = 0;
        if (fcpos != SDRO_FC_CEN) {
            const int first = fcpos == SDRO_FC_INF ? 1 : 2, other = 3 - first;
            mode = s == 0 ? first : (log2 >= 3 && s == log2 - 1) ? 0 : other;
        }
        d->st[s].mode = mode;
    }
}

int32_t sdro_decim_process(sdro_decim* d, const int16_t* iq, int32_t n_int16, int16_t* out)
{
    if (n_int16 < d->group) return 0;
    const int32_t n_cplx = (n_int16 / d->group) * (d->group / 2);
    int32_t n_out = 0;
    for (int32_t i = 0; i < n_cplx; i++) {
        int32_t re = (int32_t)((uint32_t)(int32_t)iq[2*i]   << d->pre);
        int32_t im = (int32_t)((uint32_t)(int32_t)iq[2*i+1] << d->pre);
        int s = 0;
        for (; s < d->log2; s++)
            if (!hb_push(&d->st[s], &re, &im)) break;
        if (s == d->log2) {
            out[2*n_out]   = (int16_t)(re >> d->post);
            out[2*n_out+1] = (int16_t)(im >> d->post);
            n_out++;
        }
    }
    return n_out;
}

/* DecimatorsU (decimatorsu.h:218-3251): same cascades and strides, input (buf[pos] - Shift) << pre */
sdro_decim* sdro_decimu_new(int log2, int fcpos, int shift)
{
    sdro_decim* d = sdro_decim_new(log2, fcpos, 8);
    if (d) d->ushift = shift;
    return d;
}

int32_t sdro_decimu_process(sdro_decim* d, const uint8_t* iq, int32_t n_u8, int16_t* out)
{
    if (n_u8 < d->group) return 0;
    const int32_t n_cplx = (n_u8 / d->group) * (d->group / 2);
    int32_t n_out = 0;
    for (int32_t i = 0; i < n_cplx; i++) {
        int32_t re = (int32_t)((uint32_t)((int32_t)iq[2*i]   - d->ushift) << d->pre);
        int32_t im = (int32_t)((uint32_t)((int32_t)iq[2*i+1] - d->ushift) << d->pre);
        int s = 0;
        for (; s < d->log2; s++)
            if (!hb_push(&d->st[s], &re, &im)) break;
        if (s == d->log2) {
            out[2*n_out]   = (int16_t)(re >> d->post);
            out[2*n_out+1] = (int16_t)(im >> d->post);
            n_out++;
        }
    }
    return n_out;
}

/* ------------------------------------------------------------------ 24-bit sample build (SDR_RX_SAMPLE_24BIT)
 * Decimators<qint32, qint16, 24, InputBits> (decimators.h): same cascades, strides and stage modes, the six filters are
 * IntHalfbandFilterEO<qint64,qint64,64>, shifts are decimation_shifts<24,InputBits> (:62-185), the Sample is {qint32, qint32}. */
static void shifts24(int bits, int log2, int* pre, int* post)
{
    /* <24,16>: pre 8 - log2;  <24,12>: pre 12 - log2;  <24,8>: pre 16 - log2;  post 0 */
    *pre = (bits == 16 ? 8 : bits == 12 ? 12 : 16) - log2; *post = 0;
}

sdro_decim* sdro_decim24_new(int log2, int fcpos, int bits)
{
    sdro_decim* d = sdro_decim_new(log2, fcpos, bits);
    if (!d) return 0;
    shifts24(bits, log2, &d->pre, &d->post);
    for (int s = 0; s < log2; s++) d->st[s].wide = 1;
    return d;
}

int32_t sdro_decim24_process(sdro_decim* d, const int16_t* iq, int32_t n_int16, int32_t* out)
{
    if (n_int16 < d->group) return 0;
    const int32_t n_cplx = (n_int16 / d->group) * (d->group / 2);
    int32_t n_out = 0;
    for (int32_t i = 0; i < n_cplx; i++) {
        int32_t re = (int32_t)((uint32_t)(int32_t)iq[2*i]   << d->pre);
        int32_t im = (int32_t)((uint32_t)(int32_t)iq[2*i+1] << d->pre);
        int s = 0;
        for (; s < d->log2; s++)
            if (!hb_push(&d->st[s], &re, &im)) break;
        if (s == d->log2) {
            out[2*n_out]   = re >> d->post;
            out[2*n_out+1] = im >> d->post;
            n_out++;
        }
    }
    return n_out;
}

/* ------------------------------------------------------------------ DownChannelizer */
static int contains(float ss, float se, float cs, float ce)
{
    /* signalContainsChannel (downchannelizer.cpp:240-248) */
    if (se <= ss) return 0;
    if (ce <= cs) return 0;
    return ss <= cs && se >= ce;
}

int32_t sdro_chan_plan(int32_t in_rate, int32_t req_rate, int32_t req_fc,
                       uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs)
{
    /* applyConfiguration (downchannelizer.cpp:157-189): ints first, then int -> float (Real) */
    if (in_rate == 0) { *out_rate = 0; *residual_ofs = 0; return 0; }
    float s = (float)(in_rate / -2), e = (float)(in_rate / 2);
    float cs = (float)(req_fc - req_rate / 2), ce = (float)(req_fc + req_rate / 2);
    int n = 0;
    for (;;) {
        /* createFilterChain (:250-287).  `x / 2.0` promotes to double, `x / 2.0f` stays float;
         * every argument is rounded to float when passed (Real parameters). */
        const float bw = e - s;
        const float rot = bw / 4;
        const float lo_end = (float)((double)s + (double)bw / 2.0 - 0.0);   /* safetyMargin = 0 */
        const float up_start = e - bw / 2.0f + 0.0f;
        if (n < 32 && contains(s + 0.0f, lo_end, cs, ce)) {
            modes[n++] = SDRO_MODE_LOWER;
            e = (float)((double)s + (double)bw / 2.0);
            continue;
        }
        if (n < 32 && contains(up_start, e - 0.0f, cs, ce)) {
            modes[n++] = SDRO_MODE_UPPER;
            s = e - bw / 2.0f;
            continue;
        }
        if (n < 32 && contains(s + rot + 0.0f, e - rot - 0.0f, cs, ce)) {
            modes[n++] = SDRO_MODE_CENTER;
            const float ns = s + rot, ne = e - rot;
            s = ns; e = ne;
            continue;
        }
        const float ofs = (float)((((double)(ce - cs)) / 2.0 + (double)cs) - (((double)(e - s)) / 2.0 + (double)s));
        *residual_ofs = (int32_t)ofs;                      /* Real -> int m_currentCenterFrequency */
        break;
    }
    *out_rate = in_rate / (1 << n);
    return n;
}

struct sdro_chain {
    int n;
    uint8_t modes[32];
    hb_stage st[32];
};

void sdro_chain_reset(sdro_chain* c)
{
    for (int i = 0; i < c->n; i++) hb_init(&c->st[i], 48, c->modes[i], 1);
}

sdro_chain* sdro_chain_new(int32_t n_stages, const uint8_t* modes)
{
    if (n_stages < 0 || n_stages > 32) return 0;
    sdro_chain* c = (sdro_chain*)calloc(1, sizeof *c);
    c->n = n_stages;
    if (n_stages) memcpy(c->modes, modes, (size_t)n_stages);
    sdro_chain_reset(c);
    return c;
}

void sdro_chain_free(sdro_chain* c) { free(c); }

int64_t sdro_chain_feed(sdro_chain* c, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    /* DownChannelizer::feed (downchannelizer.cpp:50-91) */
    if (c->n == 0) { memcpy(out, iq, (size_t)n_cplx * 4); return n_cplx; }   /* pass-through (:57-60) */
    const int32_t div = 1 << c->n;
    int64_t n_out = 0;
    for (int64_t i = 0; i < n_cplx; i++) {
        int32_t re = iq[2*i], im = iq[2*i+1];
        int s = 0;
        for (; s < c->n; s++)
            if (!hb_push(&c->st[s], &re, &im)) break;
        if (s == c->n) {
            out[2*n_out]   = (int16_t)(re / div);          /* s.m_real /= (1<<n): C division, toward zero */
            out[2*n_out+1] = (int16_t)(im / div);
            n_out++;
        }
    }
    return n_out;
}

/* DownChannelizer of the 24-bit build: IntHalfbandFilterEO<qint64,qint64,48> stages (downchannelizer.h:78-81) on
 * Sample{qint32, qint32}: int32 negation in the rotations, int32 store, the same final `/= (1 << n)` */
sdro_chain* sdro_chain24_new(int32_t n_stages, const uint8_t* modes)
{
    sdro_chain* c = sdro_chain_new(n_stages, modes);
    if (c) for (int i = 0; i < c->n; i++) { c->st[i].narrow = 0; c->st[i].wide = 1; }
    return c;
}

int64_t sdro_chain24_feed(sdro_chain* c, const int32_t* iq, int64_t n_cplx, int32_t* out)
{
    if (c->n == 0) { memcpy(out, iq, (size_t)n_cplx * 8); return n_cplx; }
    const int32_t div = 1 << c->n;
    int64_t n_out = 0;
    for (int64_t i = 0; i < n_cplx; i++) {
        int32_t re = iq[2*i], im = iq[2*i+1];
        int s = 0;
        for (; s < c->n; s++)
            if (!hb_push(&c->st[s], &re, &im)) break;
        if (s == c->n) { out[2*n_out] = re / div; out[2*n_out+1] = im / div; n_out++; }
    }
    return n_out;
}

/* ------------------------------------------------------------------ DC offset correction of the device stream
 * DSPDeviceSourceEngine::iqCorrections(begin, end, imbalanceCorrection = false) (dspdevicesourceengine.cpp:175-181,255-259),
 * run by work() on every FIFO span before the sinks see it when m_dcOffsetCorrection is set:
 *     m_iBeta(re); m_qBeta(im);   re -= (int32) m_iBeta;   im -= (int32) m_qBeta;
 * m_iBeta = MovingAverageUtil<int32_t, int64_t, 1024> (util/movingaverage.h): running total of the last 1024 samples
 * (fewer while filling up), read back as total / 1024 -- C++ division, truncating, by 1024 even while filling up.
 * Restated: avg[n] = trunc(sum(x[max(0, n-1023) .. n]) / 1024), y[n] = (int16)(x[n] - avg[n]); state = last 1023 inputs. */
struct sdro_dccorr { int16_t hist[2][1023]; };

sdro_dccorr* sdro_dccorr_new(void) { return (sdro_dccorr*)calloc(1, sizeof(sdro_dccorr)); }
void sdro_dccorr_free(sdro_dccorr* d) { free(d); }
void sdro_dccorr_process(sdro_dccorr* d, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    for (int comp = 0; comp < 2; comp++) {
        int64_t total = 0;                                  /* the 1023 inputs in front of x[0] (zeros at stream start) */
        for (int i = 0; i < 1023; i++) total += d->hist[comp][i];
        for (int64_t n = 0; n < n_cplx; n++) {
            total += iq[2 * n + comp];                      /* window = the last 1024 inputs including x[n] */
            out[2 * n + comp] = (int16_t)(iq[2 * n + comp] - (int32_t)(total / 1024));
            total -= n >= 1023 ? iq[2 * (n - 1023) + comp] : d->hist[comp][n];      /* the oldest one leaves */
        }
        int16_t nh[1023];                                   /* new history = last 1023 inputs */
        for (int i = 0; i < 1023; i++) {
            const int64_t src = n_cplx - 1023 + i;
            nh[i] = src >= 0 ? iq[2 * src + comp] : d->hist[comp][i + n_cplx];
        }
        memcpy(d->hist[comp], nh, sizeof nh);
    }
}

/* ------------------------------------------------------------------ DC + I/Q imbalance correction of the device stream
 * DSPDeviceSourceEngine::iqCorrections(begin, end, true) with IMBALANCE_INT undefined (dspdevicesourceengine.cpp:175-181,
 * 217-253; members dspdevicesourceengine.h:106-107,120-125).  Per sample, in this order:
 *   m_iBeta(re); m_qBeta(im)                                        MovingAverageUtil<int32,int64,1024> (as sdro_dccorr)
 *   xi = (re - (int32)m_iBeta) / 32768.f ; xq likewise              float
 *   m_avgII(xi*xi); m_avgIQ(xi*xq)                                  MovingAverageUtil<float,double,128>
 *   if (avgII != 0) m_avgPhi(avgIQ / avgII)                         MovingAverageUtil<double,double,128>, asDouble() = total / 128
 *   yq = xq - avgPhi * xi                                           double arithmetic, rounded to float on assignment
 *   m_avgII2(xi*xi); m_avgQQ2(yq*yq)
 *   if (avgQQ2 != 0) m_avgAmp(sqrt(avgII2 / avgQQ2))
 *   zq = avgAmp * yq                                                double, rounded to float
 *   re' = (qint16)(xi * 32768.f) ; im' = (qint16)(zq * 32768.f)     float -> int (truncation) -> low 16 bits
 * MovingAverageUtil<T,Total,N>::operator()(s) (util/movingaverage.h:42-56): while filling  total += s;  afterwards
 * total += s - oldest  with the subtraction in T (float for the four power averages, double for phi / amp). */
typedef struct { float s[128]; int n; unsigned idx; double total; } mavg_fd;
typedef struct { double s[128]; int n; unsigned idx; double total; } mavg_dd;
typedef struct { int32_t s[1024]; int n; unsigned idx; int64_t total; } mavg_i;

static void mavg_fd_put(mavg_fd* m, float v)
{
    if (m->n < 128) { m->s[m->n++] = v; m->total += v; }
    else { const float d = v - m->s[m->idx]; m->total += d; m->s[m->idx] = v; m->idx = (m->idx + 1) % 128; }
}
static void mavg_dd_put(mavg_dd* m, double v)
{
    if (m->n < 128) { m->s[m->n++] = v; m->total += v; }
    else { const double d = v - m->s[m->idx]; m->total += d; m->s[m->idx] = v; m->idx = (m->idx + 1) % 128; }
}
static void mavg_i_put(mavg_i* m, int32_t v)
{
    if (m->n < 1024) { m->s[m->n++] = v; m->total += v; }
    else { m->total += v - m->s[m->idx]; m->s[m->idx] = v; m->idx = (m->idx + 1) % 1024; }
}

struct sdro_iqimb { mavg_i iBeta, qBeta; mavg_fd II, IQ, II2, QQ2; mavg_dd Phi, Amp; };

sdro_iqimb* sdro_iqimb_new(void) { return (sdro_iqimb*)calloc(1, sizeof(sdro_iqimb)); }
void sdro_iqimb_free(sdro_iqimb* d) { free(d); }
void sdro_iqimb_process(sdro_iqimb* d, const int16_t* iq, int64_t n_cplx, int16_t* out)
{
    for (int64_t n = 0; n < n_cplx; n++) {
        const int re = iq[2 * n], im = iq[2 * n + 1];
        mavg_i_put(&d->iBeta, re); mavg_i_put(&d->qBeta, im);
        const float xi = (float)(re - (int32_t)(d->iBeta.total / 1024)) / 32768.0f;
        const float xq = (float)(im - (int32_t)(d->qBeta.total / 1024)) / 32768.0f;
        mavg_fd_put(&d->II, xi * xi); mavg_fd_put(&d->IQ, xi * xq);
        if (d->II.total / 128 != 0) mavg_dd_put(&d->Phi, (d->IQ.total / 128) / (d->II.total / 128));
        const float yq = (float)((double)xq - (d->Phi.total / 128) * (double)xi);
        mavg_fd_put(&d->II2, xi * xi); mavg_fd_put(&d->QQ2, yq * yq);
        if (d->QQ2.total / 128 != 0) mavg_dd_put(&d->Amp, sqrt((d->II2.total / 128) / (d->QQ2.total / 128)));
        const float zq = (float)((d->Amp.total / 128) * (double)yq);
        out[2 * n]     = (int16_t)(int32_t)(xi * 32768.0f);
        out[2 * n + 1] = (int16_t)(int32_t)(zq * 32768.0f);
    }
}
